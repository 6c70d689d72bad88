# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
T=r04z
( DEDF_STRESS_LMAX3=1 python tests/stress_parity.py 90 81; DEDF_STRESS_LMAX3=1 python tests/stress_parity.py 30 82 sample; DEDF_STRESS_LMAX3=1 python tests/stress_parity.py 30 83 ebm; python tests/stress_parity.py 30 84 half ) > gpurun_out/${T}_stress.log 2>&1
grep "ALL OK\|FAIL\|Traceback" gpurun_out/${T}_stress.log
python tests/stress_extractors.py 20 85 > gpurun_out/${T}_stress_extractors.log 2>&1; tail -3 gpurun_out/${T}_stress_extractors.log
