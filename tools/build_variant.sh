#!/bin/bash
# A/B tooling: builds svt-av1-1_amd/variants/libsvtav1_hip_<name>.so = the product library with ONE source recompiled under extra flags.
# usage: tools/build_variant.sh <name> <csrc/file.hip> [hipcc flags...]      (then: SVTAV1_HIP_LIB=... python tools/kernel_times.py)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../svt-av1-1_amd"
mkdir -p variants build
obj=variants/$(basename "$src" .hip)_$name.o
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -c "$src" -o "$obj" "$@"
others=$(for f in csrc/*.hip; do b=build/$(basename "$f" .hip).o; [ "$f" != "$src" ] && echo "$b"; done)
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libsvtav1_hip_$name.so $obj $others -L/opt/rocm/lib -lrccl
rm -f "$obj"
echo "built svt-av1-1_amd/variants/libsvtav1_hip_$name.so"
