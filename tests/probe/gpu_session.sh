mkdir -p gpurun_out
T=r05p
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/${T}_gpu_suite.log 2>&1
python tests/probe/small_batch.py > gpurun_out/${T}_small_batch.log 2>&1
DEDF_NODE_SPLIT=0 python tests/probe/small_batch.py > gpurun_out/${T}_small_batch_nosplit.log 2>&1
tail -8 gpurun_out/${T}_gpu_suite.log; grep lmax gpurun_out/${T}_small_batch.log; echo nosplit; grep lmax gpurun_out/${T}_small_batch_nosplit.log
