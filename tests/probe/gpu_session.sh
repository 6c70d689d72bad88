# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
( DEDF_STRESS_LMAX3=1 python tests/stress_parity.py 90 61; DEDF_STRESS_LMAX3=1 python tests/stress_parity.py 30 62 sample; DEDF_STRESS_LMAX3=1 python tests/stress_parity.py 30 63 ebm ) > gpurun_out/r03v_stress.log 2>&1
grep "ALL OK\|FAIL\|Traceback" gpurun_out/r03v_stress.log
python tests/stress_extractors.py > gpurun_out/r03v_stress_extractors.log 2>&1; tail -3 gpurun_out/r03v_stress_extractors.log
