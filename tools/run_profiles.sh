#!/bin/bash
# GPU-box script: the round's measured artefacts -- full GPU test suite, bench line (with CPU baseline), rocprofv3 kernel
# stats of the same command, whole-picture ME probe.  (HBM traffic counters: tools/run_pmc_traffic.sh.)
set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q > $R/gpurun_out/gpu_tests.log 2>&1 || { tail -20 $R/gpurun_out/gpu_tests.log; exit 1; }
tail -2 $R/gpurun_out/gpu_tests.log
timeout -k 10 400 python3 $R/bench.py > $R/gpurun_out/bench_full.json 2> $R/gpurun_out/bench_full.err
cat $R/gpurun_out/bench_full.json
timeout -k 10 200 python3 $R/tools/me_picture_probe.py 10 batch > $R/gpurun_out/me_picture_probe.txt 2>&1 || true
timeout -k 10 200 python3 $R/tools/me_picture_probe.py 10 batch209 >> $R/gpurun_out/me_picture_probe.txt 2>&1 || true
grep 1080p $R/gpurun_out/me_picture_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
cut -c1-140 $R/gpurun_out/prof_bench/bench_kernel_stats.csv
