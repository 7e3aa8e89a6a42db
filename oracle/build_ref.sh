#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds the *reference's own* kernel sources, where they lie under
# /root/reference, into oracle/_ref/ (git-ignored, but shipped to the GPU box with the snapshot).
#
# What is built and why it is legitimate:
#   * Only reference .c files are compiled, straight from /root/reference (nothing is copied into
#     the repo).  No stand-in is written for anything the image lacks: the reference's 9 NASM files
#     cannot be assembled here (no nasm/yasm), so every object that needs one of their 36 symbols on
#     a path we drive is simply NOT used.  Unused functions that mention a NASM symbol are dropped by
#     --gc-sections (roots = the exported list below), so the library has no unresolved symbols.
#   * libsvtref_kernels.so : leaf kernels of the ME path (both asm_type rows):
#       C_DEFAULT/EbComputeSAD_C.c, ASM_SSE2/EbMeSadCalculation_Intrinsic_SSE2.c,
#       ASM_SSE4_1/EbComputeSAD_Intrinsic_SSE4_1.c, ASM_AVX2/EbComputeSAD_Intrinsic_AVX2.c,
#       ASM_SSE4_1/EbPictureOperators_Intrinsic_SSE4_1.c (wrapped SSD),
#       ASM_SSSE3/EbAvcStyleMcp_Intrinsic_SSSE3.c (4-tap half-pel filters),
#       ASM_SSE2/EbCombinedAveragingSAD_Intrinsic_SSE2.c, ASM_AVX2/EbCombinedAveragingSAD_Intrinsic_AVX2.c
#   The repo's oracle (oracle/*.c) is validated against these in tests/test_oracle_vs_ref.py, and
#   tests/golden/ fixtures are generated from them by tests/golden/make_golden.py.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${SVT_REFERENCE_ROOT:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/Source/Lib/Codec" ]; then
  echo "build_ref: $REF not present (GPU box) - using prebuilt oracle/_ref if any" >&2
  exit 0
fi
mkdir -p "$OUT/obj"
S="$REF/Source"
INC="-I$S/API -I$S/Lib/Codec -I$S/Lib/C_DEFAULT -I$S/Lib/ASM_SSE2 -I$S/Lib/ASM_SSSE3 -I$S/Lib/ASM_SSE4_1 -I$S/Lib/ASM_AVX2"
CFLAGS="-O2 -std=c99 -w -mavx2 -fPIC -ffunction-sections -fdata-sections"
FILES=(
  Lib/C_DEFAULT/EbComputeSAD_C.c
  Lib/ASM_SSE2/EbMeSadCalculation_Intrinsic_SSE2.c
  Lib/ASM_SSE2/EbCombinedAveragingSAD_Intrinsic_SSE2.c
  Lib/ASM_SSE4_1/EbComputeSAD_Intrinsic_SSE4_1.c
  Lib/ASM_SSE4_1/EbPictureOperators_Intrinsic_SSE4_1.c
  Lib/ASM_SSSE3/EbAvcStyleMcp_Intrinsic_SSSE3.c
  Lib/ASM_AVX2/EbComputeSAD_Intrinsic_AVX2.c
  Lib/ASM_AVX2/EbCombinedAveragingSAD_Intrinsic_AVX2.c
)
OBJS=()
for f in "${FILES[@]}"; do
  o="$OUT/obj/$(basename "$f" .c).o"
  if [ ! -f "$o" ] || [ "$S/$f" -nt "$o" ]; then
    gcc $CFLAGS $INC -c "$S/$f" -o "$o"
  fi
  OBJS+=("$o")
done
# our own drivers that replay the call sequence of the reference's *static* L2 functions over the
# reference kernels (they contain no reference code, only calls into it)
for f in "$HERE"/ref_fullpel_driver.c; do
  o="$OUT/obj/$(basename "$f" .c).o"
  gcc -O2 -std=gnu99 -Wall -mavx2 -fPIC -ffunction-sections -c "$f" -o "$o"
  OBJS+=("$o")
done
# exported roots (everything else is local and garbage-collected)
cat > "$OUT/obj/kernels.map" <<'EOF'
{
  global:
    SadLoopKernel; FastLoop_NxMSadKernel; CombinedAveragingSAD;
    SadLoopKernel_SSE4_1_INTRIN; SadLoopKernel_SSE4_1_HmeL0_INTRIN;
    SadLoopKernel_AVX2_INTRIN; SadLoopKernel_AVX2_HmeL0_INTRIN;
    GetEightHorizontalSearchPointResults_8x8_16x16_PU_SSE41_INTRIN;
    GetEightHorizontalSearchPointResults_32x32_64x64_PU_SSE41_INTRIN;
    GetEightHorizontalSearchPointResults_8x8_16x16_PU_AVX2_INTRIN;
    GetEightHorizontalSearchPointResults_32x32_64x64_PU_AVX2_INTRIN;
    SadCalculation_8x8_16x16_SSE2_INTRIN; SadCalculation_32x32_64x64_SSE2_INTRIN;
    InitializeBuffer_32bits_SSE2_INTRIN;
    ExtSadCalculation_8x8_16x16_SSE4_INTRIN; ExtSadCalculation_32x32_64x64_SSE4_INTRIN;
    SpatialFullDistortionKernel4x4_SSSE3_INTRIN; SpatialFullDistortionKernel8x8_SSSE3_INTRIN;
    SpatialFullDistortionKernel16MxN_SSSE3_INTRIN;
    AvcStyleLumaInterpolationFilterHorizontal_SSSE3_INTRIN;
    AvcStyleLumaInterpolationFilterVertical_SSSE3_INTRIN;
    Compute4xMSad_AVX2_INTRIN; Compute8xMSad_AVX2_INTRIN; Compute16xMSad_AVX2_INTRIN;
    Compute24xMSad_AVX2_INTRIN; Compute32xMSad_AVX2_INTRIN; Compute48xMSad_AVX2_INTRIN;
    Compute64xMSad_AVX2_INTRIN;
    CombinedAveraging8xMSAD_AVX2_INTRIN; CombinedAveraging16xMSAD_AVX2_INTRIN;
    CombinedAveraging24xMSAD_AVX2_INTRIN; CombinedAveraging32xMSAD_AVX2_INTRIN;
    CombinedAveraging48xMSAD_AVX2_INTRIN; CombinedAveraging64xMSAD_AVX2_INTRIN;
    CombinedAveraging4xMSAD_SSE2_INTRIN;
    ref_*;
  local: *;
};
EOF
gcc -shared -o "$OUT/libsvtref_kernels.so" "${OBJS[@]}" \
    -Wl,--gc-sections -Wl,--version-script="$OUT/obj/kernels.map" -Wl,-z,defs -lc
echo "built $OUT/libsvtref_kernels.so"

# ---------------------------------------------------------------------------------------------
# libsvtref_me.so : the reference's MotionEstimateLcu (Codec/EbMotionEstimation.c:6152) with everything it
# reaches, driven by oracle/ref_me_lcu_driver.c.  All 144 reference .c files are compiled where they lie;
# --gc-sections keeps only what MotionEstimateLcu / MeContextCtor reach.  Two symbols that exist only in the
# reference's NASM sources stay UNRESOLVED (no stand-ins are written): Log2f_SSE2 and
# PictureCopyKernel_SSE2.  They are call targets only, so the library loads with lazy binding and works
# as long as neither is reached: the driver refuses use_subpel_flag=1 (PU_HalfPelRefinement calls Log2f).
mkdir -p "$OUT/obj_all"
export OUT CFLAGS INC
ls "$S"/Lib/{C_DEFAULT,ASM_SSE2,ASM_SSSE3,ASM_SSE4_1,ASM_AVX2,Codec}/*.c | xargs -P "$(nproc)" -I{} sh -c \
  'o="$OUT/obj_all/$(basename {} .c).o"; if [ ! -f "$o" ] || [ {} -nt "$o" ]; then gcc $CFLAGS $INC -c {} -o "$o"; fi'
gcc -O2 -std=gnu99 -w -mavx2 -fPIC -ffunction-sections $INC -c "$HERE/ref_me_lcu_driver.c" -o "$OUT/obj/ref_me_lcu_driver.o"
gcc -O2 -std=gnu99 -Wall -fPIC -ffunction-sections -c "$HERE/ref_fullpel209_driver.c" -o "$OUT/obj/ref_fullpel209_driver.o"
# leaf kernels of the sub-pel refinement called through the reference's own function-pointer tables (headers where they lie)
gcc -O2 -std=gnu99 -w -mavx2 -fPIC -ffunction-sections $INC -c "$HERE/ref_subpel_leaf_driver.c" -o "$OUT/obj/ref_subpel_leaf_driver.o"
# the reference's quantiser tables (av1_build_quantizer) and scan orders (av1_scan_orders) in the ABI's row layout
gcc -O2 -std=gnu99 -w -mavx2 -fPIC -ffunction-sections $INC -c "$HERE/ref_quant_tables_driver.c" -o "$OUT/obj/ref_quant_tables_driver.o"
# the reference's 8-bit single-reference convolutions (av1_convolve_{2d,x,y,2d_copy}_sr_c) driven like av1_inter_prediction
gcc -O2 -std=gnu99 -w -mavx2 -fPIC -ffunction-sections $INC -c "$HERE/ref_convolve_driver.c" -o "$OUT/obj/ref_convolve_driver.o"
# the reference's open-loop intra search (OpenLoopIntraSearchLcu) and its neighbour / prediction building blocks
gcc -O2 -std=gnu99 -w -mavx2 -fPIC -ffunction-sections $INC -c "$HERE/ref_ois_driver.c" -o "$OUT/obj/ref_ois_driver.o"
# the reference's bi-prediction search for chosen quarter-pel vectors (BiPredictionSearch and everything under it)
gcc -O2 -std=gnu99 -w -mavx2 -fPIC -ffunction-sections $INC -c "$HERE/ref_bipred_driver.c" -o "$OUT/obj/ref_bipred_driver.o"
printf '{ global: ref_me_lcu_run; ref_interp_region; ref_fullpel_search_209pu; ref_halfpel_ssd_leaf; ref_halfpel_sad_leaf; ref_quarterpel_ssd_leaf; ref_quarterpel_sad_leaf; generate_padding; generate_padding16_bit; Decimation2D; ref_build_quantizer_rows; ref_scan_order; ref_av1_convolve_sr; ref_av1_convolve_compound; ref_av1_highbd_convolve_sr; ref_av1_highbd_convolve_compound; ref_ois_predict; ref_ois_search_picture; ref_bipred_search; local: *; };\n' > "$OUT/obj/me.map"
gcc -shared -o "$OUT/libsvtref_me.so" "$OUT"/obj_all/*.o "$OUT/obj/ref_me_lcu_driver.o" "$OUT/obj/ref_fullpel209_driver.o" "$OUT/obj/ref_subpel_leaf_driver.o" "$OUT/obj/ref_quant_tables_driver.o" "$OUT/obj/ref_convolve_driver.o" "$OUT/obj/ref_ois_driver.o" "$OUT/obj/ref_bipred_driver.o" \
    -Wl,--gc-sections -Wl,--version-script="$OUT/obj/me.map" -lm -lpthread
echo "built $OUT/libsvtref_me.so (unresolved by design: $(nm -D "$OUT/libsvtref_me.so" | awk '$1=="U" && $2 !~ /@/ {printf "%s ", $2}'))"

# ---------------------------------------------------------------------------------------------
# libsvtref_subpel.so : the reference's sub-pel refinement (HalfPelSearch_LCU + the static QuarterPelSearch_LCU) run with
# fractionalSearchMethod = SUB_SAD_SEARCH / FULL_SAD_SEARCH, where Log2f_SSE2 is never evaluated (oracle/ref_subpel_search_driver.c).
# The driver's translation unit IS the reference's EbMotionEstimation.c (included where it lies, to reach the static function), so it is
# linked in place of obj_all/EbMotionEstimation.o; everything else as above (same objects, no stand-ins, lazy binding).
gcc $CFLAGS $INC -std=gnu99 -c "$HERE/ref_subpel_search_driver.c" -o "$OUT/obj/ref_subpel_search_driver.o"
printf '{ global: ref_subpel_search; local: *; };\n' > "$OUT/obj/subpel.map"
gcc -shared -o "$OUT/libsvtref_subpel.so" $(ls "$OUT"/obj_all/*.o | grep -v '/EbMotionEstimation\.o$') "$OUT/obj/ref_subpel_search_driver.o" \
    -Wl,--gc-sections -Wl,--version-script="$OUT/obj/subpel.map" -lm -lpthread
echo "built $OUT/libsvtref_subpel.so (unresolved by design: $(nm -D "$OUT/libsvtref_subpel.so" | awk '$1=="U" && $2 !~ /@/ {printf "%s ", $2}'))"

# ---------------------------------------------------------------------------------------------
# libsvtref_tq.so : the reference's C transform and quantisation entry points (parity targets of the
# transform/quant kernels): Av1TransformTwoD_NxN_c, av1_fwd_txfm2d_WxH_c, Av1InverseTransformTwoD_NxN_c,
# av1_inv_txfm2d_add_WxH_c (Codec/EbTransforms.c) and aom_quantize_b*_c_II / aom_highbd_quantize_b*_c
# (Codec/EbFullLoop.c).  Same objects, same no-stand-in rule; lazy binding for anything NASM-only.
printf '{ global: Av1TransformTwoD_*_c; av1_fwd_txfm2d_*_c; Av1InverseTransformTwoD_*_c; av1_inv_txfm2d_add_*_c; aom_quantize_b*_c_II; aom_highbd_quantize_b*_c; av1_cospi_arr_data; av1_sinpi_arr_data; HandleTransform*_c; EnergyComputation; FullDistortionKernel32Bits; ResidualKernel_c; local: *; };\n' > "$OUT/obj/tq.map"
gcc -shared -o "$OUT/libsvtref_tq.so" "$OUT"/obj_all/*.o \
    -Wl,--gc-sections -Wl,--version-script="$OUT/obj/tq.map" -lm -lpthread
echo "built $OUT/libsvtref_tq.so (unresolved: $(nm -D "$OUT/libsvtref_tq.so" | awk '$1=="U" && $2 !~ /@/ {printf "%s ", $2}'))"

# ---------------------------------------------------------------------------------------------
# libsvtref_bench.so : batch loops over the reference's C transform / quantisation chain and single-reference convolutions for
# bench.py's per-leg cpu_baseline (oracle/ref_bench_driver.c: calls only, no reference code).  Same objects, same no-stand-in rule.
gcc -O2 -std=gnu99 -w -mavx2 -fPIC -ffunction-sections $INC -c "$HERE/ref_bench_driver.c" -o "$OUT/obj/ref_bench_driver.o"
printf '{ global: ref_bench_tq_chain; ref_bench_convolve; local: *; };\n' > "$OUT/obj/bench.map"
gcc -shared -o "$OUT/libsvtref_bench.so" "$OUT"/obj_all/*.o "$OUT/obj/ref_bench_driver.o" "$OUT/obj/ref_convolve_driver.o" \
    -Wl,--gc-sections -Wl,--version-script="$OUT/obj/bench.map" -lm -lpthread
echo "built $OUT/libsvtref_bench.so (unresolved: $(nm -D "$OUT/libsvtref_bench.so" | awk '$1=="U" && $2 !~ /@/ {printf "%s ", $2}'))"
