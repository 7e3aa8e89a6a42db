set -e
python -m pytest tests/test_fullpel_gpu.py tests/test_me_full_gpu.py tests/test_subpel_gpu.py tests/test_golden.py -m gpu -x -q > gpurun_out/r03_t2.txt 2>&1 || { tail -30 gpurun_out/r03_t2.txt; exit 1; }
tail -2 gpurun_out/r03_t2.txt
python tools/kernel_times.py > gpurun_out/r03_times_b.txt 2>&1
cat gpurun_out/r03_times_b.txt
