#!/bin/bash
# GPU-box script: transform / quant parity, per-size transform timing, kernel trace.
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_tq_gpu.py -x -q > gpurun_out/tq_tests.log 2>&1 || { tail -30 gpurun_out/tq_tests.log; exit 1; }
tail -3 gpurun_out/tq_tests.log
timeout -k 10 300 python tools/tq_probe.py > gpurun_out/tq_probe.txt 2>&1
cat gpurun_out/tq_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tq -o tq -- python3 $GRAFT_REPO_ROOT/tools/tq_probe.py --iters 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_tq.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_tq -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -50
