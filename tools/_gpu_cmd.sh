set -e
V=$PWD/svt-av1-1_amd/variants
python -m pytest tests/test_hme_gpu.py -m gpu -x -q > gpurun_out/r03_tests_j.txt 2>&1 || { tail -30 gpurun_out/r03_tests_j.txt; exit 1; }
tail -2 gpurun_out/r03_tests_j.txt
{
python tools/kernel_times.py hme fp85 fp209
SVTAV1_HIP_LIB=$V/libsvtav1_hip_hmenoprio.so python tools/kernel_times.py hme
SVTAV1_HIP_LIB=$V/libsvtav1_hip_fp85prio.so python tools/kernel_times.py fp85
SVTAV1_HIP_LIB=$V/libsvtav1_hip_fp209prio.so python tools/kernel_times.py fp209
python bench.py --no-legs --no-cpu-baseline
} > gpurun_out/r03_times_j.txt 2>&1
cat gpurun_out/r03_times_j.txt
