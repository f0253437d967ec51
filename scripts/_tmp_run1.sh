timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_sgm.py tests/test_gpu_bm.py -x -q -m gpu > gpurun_out/pytest_tmp.txt 2>&1
echo "rc=$?"; tail -15 gpurun_out/pytest_tmp.txt | cut -c1-300
