/*
 * oracle/ref_subpel_leaf_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Calls the leaf kernels of the REFERENCE's sub-pel refinement THROUGH THE REFERENCE'S OWN FUNCTION-POINTER TABLES, compiled
 * from /root/reference into oracle/_ref/libsvtref_me.so by oracle/build_ref.sh:
 *   SpatialFullDistortionKernel_funcPtrArray[asm_type][k]   (Codec/EbPictureOperators.h:532-559)
 *   NxMSadKernel_funcPtrArray[asm_type][k]                  (Codec/EbComputeSAD.h:126-152)
 *   NxMSadAveragingKernel_funcPtrArray[asm_type][k]         (Codec/EbComputeSAD.h:154-180)
 *   CombinedAveragingSSD                                    (Codec/EbMotionEstimation.c:2792-2817)
 * with the argument order PU_HalfPelRefinement / PU_QuarterPelRefinementOnTheFly use (Codec/EbMotionEstimation.c:1912-1943,
 * :2914-2929).  Contains no reference code, only the table lookups and calls.
 *
 * The table index is an ARGUMENT: the reference computes it as `Log2f(pu_width) - 2` (SSD) and `pu_width >> 3` (SAD);
 * Log2f is Log2f_SSE2 (Codec/EbDefinitions.h:1822), which exists only in a NASM file this image cannot assemble, and no
 * stand-in is written for it -- the test computes floor(log2(width)) itself and says so.
 */
#include <stdint.h>

#include "EbDefinitions.h"
#include "EbComputeSAD.h"
#include "EbPictureOperators.h"

uint32_t CombinedAveragingSSD(uint8_t *src, uint32_t src_stride, uint8_t *ref1, uint32_t ref1Stride, uint8_t *ref2, uint32_t ref2Stride,
                              uint32_t height, uint32_t width);

/* *pBestSsd = (uint32_t)SpatialFullDistortionKernel_funcPtrArray[asm_type][ssd_index](src, stride, ref, stride, pu_width, pu_height) */
uint32_t ref_halfpel_ssd_leaf(int asm_type, int ssd_index, uint8_t *src, uint32_t src_stride, uint8_t *rec, uint32_t rec_stride,
                              uint32_t pu_width, uint32_t pu_height)
{
    return (uint32_t)SpatialFullDistortionKernel_funcPtrArray[asm_type][ssd_index](src, src_stride, rec, rec_stride, pu_width, pu_height);
}

/* *pBestSad = (uint32_t)NxMSadKernel_funcPtrArray[asm_type][sad_index](src, stride, ref, stride, pu_height, pu_width) */
uint32_t ref_halfpel_sad_leaf(int asm_type, int sad_index, uint8_t *src, uint32_t src_stride, uint8_t *rec, uint32_t rec_stride,
                              uint32_t pu_width, uint32_t pu_height)
{
    return (uint32_t)NxMSadKernel_funcPtrArray[asm_type][sad_index](src, src_stride, rec, rec_stride, pu_height, pu_width);
}

/* dist = CombinedAveragingSSD(src, 64, buf1, stride1, buf2, stride2, pu_height, pu_width) */
uint32_t ref_quarterpel_ssd_leaf(uint8_t *src, uint32_t src_stride, uint8_t *r1, uint32_t r1_stride, uint8_t *r2, uint32_t r2_stride,
                                 uint32_t pu_width, uint32_t pu_height)
{
    return CombinedAveragingSSD(src, src_stride, r1, r1_stride, r2, r2_stride, pu_height, pu_width);
}

/* *pBestSad = (uint32_t)NxMSadAveragingKernel_funcPtrArray[asm_type][sad_index](src, 64, buf1, s1, buf2, s2, pu_height, pu_width) */
uint32_t ref_quarterpel_sad_leaf(int asm_type, int sad_index, uint8_t *src, uint32_t src_stride, uint8_t *r1, uint32_t r1_stride, uint8_t *r2,
                                 uint32_t r2_stride, uint32_t pu_width, uint32_t pu_height)
{
    return (uint32_t)NxMSadAveragingKernel_funcPtrArray[asm_type][sad_index](src, src_stride, r1, r1_stride, r2, r2_stride, pu_height, pu_width);
}
