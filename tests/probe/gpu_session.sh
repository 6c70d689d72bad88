mkdir -p gpurun_out
python tests/probe/range_floor.py > gpurun_out/r05k_range_floor.log 2>&1
cat gpurun_out/r05k_range_floor.log | grep -v amdgpu
