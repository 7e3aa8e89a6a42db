#!/bin/bash
# GPU box: golden check of the 209-PU chain, then its timing and per-kernel split
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_golden.py -q -m gpu > gpurun_out/golden209.log 2>&1
tail -2 gpurun_out/golden209.log
timeout -k 10 200 python tools/me_picture_probe.py 5 batch209 | tee gpurun_out/me209_probe.txt
timeout -k 10 200 python tools/me_picture_probe.py 5 batch | tee -a gpurun_out/me209_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof209" -- python3 "$GRAFT_REPO_ROOT/tools/me_picture_probe.py" 3 batch209 > "$GRAFT_REPO_ROOT/gpurun_out/prof209.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/prof209 -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/me209_kernel_stats.csv
cat gpurun_out/me209_kernel_stats.csv
