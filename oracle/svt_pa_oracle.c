/*
 * oracle/svt_pa_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see svt_me_oracle.h).
 *
 * Plain-C restatement of the picture-analysis producers of the ME inputs and of the reference-picture border padding
 * (paths under Source/Lib/Codec of the reference):
 *   generate_padding        EbMcp.c:173-215    horizontal replication of every picture row, then whole padded rows copied up / down
 *   generate_padding16_bit  EbMcp.c:220-262    the same on 16-bit samples
 *   Decimation2D            EbPictureAnalysisProcess.c:100-125   point sampling, every `step`-th sample of every `step`-th row
 *   DecimateInputPicture    EbPictureAnalysisProcess.c:4885-4936 Decimation2D into the decimated plane's interior (written at
 *                           origin_x + origin_x * stride -- x used for y; paddings are square) + generate_padding
 * The statement order of the reference is kept (two dependent passes) so that the device kernel's single-pass gather is checked
 * against the literal algorithm.  PINNED against the reference's own generate_padding / generate_padding16_bit / Decimation2D
 * (oracle/_ref/libsvtref_me.so) in tests/test_pa_vs_ref.py.
 */
#include <stdint.h>
#include <string.h>

void orc_generate_padding(uint8_t *pic, uint32_t stride, uint32_t width, uint32_t height, uint32_t pad_w, uint32_t pad_h)
{
    uint8_t *row = pic + pad_w + (size_t)pad_h * stride;
    for (uint32_t y = 0; y < height; y++, row += stride) { /* EbMcp.c:188-196 */
        memset(row - pad_w, row[0], pad_w);
        memset(row + width, row[width - 1], pad_w);
    }
    uint8_t *top = pic + (size_t)pad_h * stride, *bot = pic + (size_t)(pad_h + height - 1) * stride;
    for (uint32_t k = 1; k <= pad_h; k++) { /* :198-212: whole rows of `stride` bytes */
        memcpy(top - (size_t)k * stride, top, stride);
        memcpy(bot + (size_t)k * stride, bot, stride);
    }
}

/* all quantities in SAMPLES (the reference's 16-bit variant is called with byte quantities: stride << 1, width << 1, ...) */
void orc_generate_padding16(uint16_t *pic, uint32_t stride, uint32_t width, uint32_t height, uint32_t pad_w, uint32_t pad_h)
{
    uint16_t *row = pic + pad_w + (size_t)pad_h * stride;
    for (uint32_t y = 0; y < height; y++, row += stride) {
        for (uint32_t x = 1; x <= pad_w; x++) row[-(int)x] = row[0];
        for (uint32_t x = 0; x < pad_w; x++) row[width + x] = row[width - 1];
    }
    uint16_t *top = pic + (size_t)pad_h * stride, *bot = pic + (size_t)(pad_h + height - 1) * stride;
    for (uint32_t k = 1; k <= pad_h; k++) {
        memcpy(top - (size_t)k * stride, top, sizeof(uint16_t) * stride);
        memcpy(bot + (size_t)k * stride, bot, sizeof(uint16_t) * stride);
    }
}

void orc_decimation_2d(const uint8_t *in, uint32_t in_stride, uint32_t in_w, uint32_t in_h, uint8_t *out, uint32_t out_stride, uint32_t step)
{
    for (uint32_t y = 0; y < in_h; y += step) { /* EbPictureAnalysisProcess.c:114-122 */
        for (uint32_t x = 0; x < in_w; x += step) out[x >> (step >> 1)] = in[x];
        in += (size_t)in_stride << (step >> 1);
        out += out_stride;
    }
}

/* What Picture Analysis leaves in an EbPaReferenceObject_t: pads the full plane (origin 68), fills + pads the quarter (origin 32)
 * and sixteenth (origin 16) planes.  `full` must hold the width x height picture at (68,68). */
void orc_pa_derive_planes(uint8_t *full, uint32_t full_stride, uint32_t width, uint32_t height, uint8_t *quarter, uint32_t quarter_stride,
                          uint8_t *sixteenth, uint32_t sixteenth_stride)
{
    orc_generate_padding(full, full_stride, width, height, 68, 68);
    const uint8_t *pic = full + 68 + (size_t)68 * full_stride;
    if (quarter) {
        orc_decimation_2d(pic, full_stride, width, height, quarter + 32 + (size_t)32 * quarter_stride, quarter_stride, 2);
        orc_generate_padding(quarter, quarter_stride, width >> 1, height >> 1, 32, 32);
    }
    if (sixteenth) {
        orc_decimation_2d(pic, full_stride, width, height, sixteenth + 16 + (size_t)16 * sixteenth_stride, sixteenth_stride, 4);
        orc_generate_padding(sixteenth, sixteenth_stride, width >> 2, height >> 2, 16, 16);
    }
}
