set -e
V=$PWD/svt-av1-1_amd/variants
for i in 1 2 3; do
python bench.py --no-legs --no-cpu-baseline | cut -c60-140
SVTAV1_HIP_LIB=$V/libsvtav1_hip_hmenoprio.so python bench.py --no-legs --no-cpu-baseline | cut -c60-140
done
