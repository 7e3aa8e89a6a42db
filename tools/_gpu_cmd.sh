set -e
for v in "" nopu noplanes; do
  if [ -n "$v" ]; then export SVTAV1_HIP_LIB=$PWD/svt-av1-1_amd/variants/libsvtav1_hip_$v.so; fi
  python tools/kernel_times.py sub85 sub209 >> gpurun_out/r03_times_c.txt 2>&1
done
cat gpurun_out/r03_times_c.txt
