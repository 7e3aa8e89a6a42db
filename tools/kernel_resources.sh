#!/bin/bash
# Resource metadata of every kernel in a built object: VGPRs, spills, scratch, LDS, occupancy -- from the code object's notes.
# usage: tools/kernel_resources.sh svt-av1-1_amd/csrc/me_fullpel209.hip [extra hipcc flags]
set -e
src=$1; shift
out=/tmp/kres_$$
mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$src" -o $out/o.o -save-temps=obj "$@" 2>/dev/null
asm=$(ls $out/*gfx950*.s | head -1)
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|agpr_count):" "$asm" | paste - - - - - - - - | sed 's/\s\+/ /g'
echo "scratch instructions: $(grep -c 'scratch_' "$asm")"
cp "$asm" /tmp/last_kernel.s
rm -rf $out
