set -e
V=$PWD/svt-av1-1_amd/variants
python -m pytest tests/test_subpel_gpu.py -m gpu -x -q > gpurun_out/r03_tests_i.txt 2>&1 || { tail -30 gpurun_out/r03_tests_i.txt; exit 1; }
tail -2 gpurun_out/r03_tests_i.txt
{
python tools/kernel_times.py sub85 sub209
SVTAV1_HIP_LIB=$V/libsvtav1_hip_stamps.so python tools/subpel_stamps_probe.py
} > gpurun_out/r03_times_i.txt 2>&1
cat gpurun_out/r03_times_i.txt
