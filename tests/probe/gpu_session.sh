mkdir -p gpurun_out
T=r04r
python tests/probe/fps_time.py 2>&1 | grep "fps n" > gpurun_out/${T}_fps_time.log; cat gpurun_out/${T}_fps_time.log
python -m pytest tests/test_graph.py -m gpu -q -x 2>&1 | tail -2
