mkdir -p gpurun_out
T=r05t
python -m pytest tests/test_unet.py tests/test_lmax3.py tests/test_keypoint_extractor.py tests/test_config5.py tests/test_agent.py -m gpu -q --durations=5 > gpurun_out/${T}_unet_suite.log 2>&1
python tests/probe/unet_time.py > gpurun_out/${T}_unet_time.log 2>&1
DEDF_SO2_UNET=0 python tests/probe/unet_time.py > gpurun_out/${T}_unet_time_general.log 2>&1
python tests/probe/unet_time.py >> gpurun_out/${T}_unet_time.log 2>&1
DEDF_SO2_UNET=0 python tests/probe/unet_time.py >> gpurun_out/${T}_unet_time_general.log 2>&1
tail -6 gpurun_out/${T}_unet_suite.log; grep -v amdgpu gpurun_out/${T}_unet_time.log; echo general; grep -v amdgpu gpurun_out/${T}_unet_time_general.log
