set -e
V=$PWD/svt-av1-1_amd/variants
python -m pytest tests/test_subpel_gpu.py tests/test_hme_gpu.py tests/test_me_full_gpu.py tests/test_golden.py tests/test_me_4k_gpu.py -m gpu -x -q > gpurun_out/r03_tests_g.txt 2>&1 || { tail -30 gpurun_out/r03_tests_g.txt; exit 1; }
tail -2 gpurun_out/r03_tests_g.txt
{
python tools/kernel_times.py hme sub85 sub209
SVTAV1_HIP_LIB=$V/libsvtav1_hip_stamps.so python tools/subpel_stamps_probe.py
} > gpurun_out/r03_times_g.txt 2>&1
cat gpurun_out/r03_times_g.txt
