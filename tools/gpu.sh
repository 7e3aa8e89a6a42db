#!/bin/bash
# build the product first; refuse to spend GPU time on a stale .so
set -e
cd /root/repo
if ! make -C svt-av1-1_amd -j8 > /tmp/build.log 2>&1; then grep -E "error" /tmp/build.log | head; echo "BUILD FAILED"; exit 1; fi
grep -E "warning" /tmp/build.log | head -5 || true
exec /usr/local/graft/bin/gpurun "$@"
