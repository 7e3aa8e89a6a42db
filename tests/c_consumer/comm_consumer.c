/*
 * tests/c_consumer/comm_consumer.c -- the multi-GPU entries of include/svtav1_hip.h driven from C99, the way the reference's C host
 * would: device planes from the HIP runtime's C API, everything else through the library.
 *
 *   comm_consumer plan  <width> <height> <origin> <sample_bytes> <world>     (no device needed)
 *        prints, for every rank, its SB share, its slab rows and its transfer list (svthip_shard_range, svthip_recon_slab_rows,
 *        svthip_recon_exchange_plan) -- tests/test_comm_consumer.py checks them on the CPU;
 *   comm_consumer exchange <in.bin> <out.bin>                                (GPU box, world 1)
 *        in : u32 width, height, origin, sample_bytes, then the padded Y, Cb, Cr planes (interior valid, borders junk)
 *        out: the three planes after svthip_comm_create(world 1) -> svthip_recon_exchange_dev -> svthip_synchronize,
 *             then 3 x 28 x 85 x 24 bytes gathered by svthip_me_gather_results_dev from the pattern the program generated.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "svtav1_hip.h"

#define CHECK(call)                                                                                                   \
    do {                                                                                                              \
        int32_t rc_ = (call);                                                                                         \
        if (rc_ != SVTHIP_OK) {                                                                                       \
            fprintf(stderr, "comm_consumer: %s -> 0x%08x: %s / %s\n", #call, (unsigned)rc_, svthip_comm_last_error(), svthip_last_error()); \
            exit(3);                                                                                                  \
        }                                                                                                             \
    } while (0)
#define HIPCHECK(call)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "comm_consumer: %s -> %s\n", #call, hipGetErrorString(e_));       \
            exit(3);                                                                          \
        }                                                                                     \
    } while (0)

static void fill_picture(svthip_recon_picture *p, uint32_t w, uint32_t h, uint32_t origin, uint32_t sample_bytes)
{
    memset(p, 0, sizeof(*p));
    p->stride_y = w + 2 * origin;
    p->stride_cb = p->stride_cr = (w >> 1) + origin; /* (w >> 1) + 2 * (origin >> 1) */
    p->width = (uint16_t)w;
    p->height = (uint16_t)h;
    p->origin_x = p->origin_y = (uint16_t)origin;
    p->sample_bytes = (uint8_t)sample_bytes;
}

static int do_plan(int argc, char **argv)
{
    if (argc != 7) return 2;
    const uint32_t w = (uint32_t)atoi(argv[2]), h = (uint32_t)atoi(argv[3]), origin = (uint32_t)atoi(argv[4]), sb = (uint32_t)atoi(argv[5]);
    const int32_t world = atoi(argv[6]);
    svthip_recon_picture pic;
    fill_picture(&pic, w, h, origin, sb);
    pic.y = pic.cb = pic.cr = (void *)(uintptr_t)16; /* the plan only asks whether chroma is present */
    const uint32_t n_sb = ((w + 63) / 64) * ((h + 63) / 64);
    for (int32_t r = 0; r < world; r++) {
        uint32_t f, c, y0, n, nx = 0;
        svthip_shard_range(n_sb, world, r, &f, &c);
        svthip_recon_slab_rows(h, world, r, &y0, &n);
        printf("rank %d sb %u %u rows %u %u\n", r, f, c, y0, n);
        svthip_xfer plan[3 * 2 * 64];
        CHECK(svthip_recon_exchange_plan(&pic, world, r, plan, 3 * 2 * 64, &nx));
        for (uint32_t i = 0; i < nx; i++)
            printf("xfer %d %d %u %u %llu %llu\n", r, plan[i].peer, plan[i].plane, plan[i].send, (unsigned long long)plan[i].offset,
                   (unsigned long long)plan[i].bytes);
    }
    return 0;
}

static int do_exchange(int argc, char **argv)
{
    if (argc != 4) return 2;
    FILE *fi = fopen(argv[2], "rb");
    if (!fi) { perror(argv[2]); return 2; }
    uint32_t hdr[4];
    if (fread(hdr, 4, 4, fi) != 4) return 2;
    svthip_recon_picture pic;
    fill_picture(&pic, hdr[0], hdr[1], hdr[2], hdr[3]);
    const size_t es = hdr[3];
    const size_t bytes[3] = {(size_t)pic.stride_y * (pic.height + 2u * pic.origin_y) * es,
                             (size_t)pic.stride_cb * ((pic.height >> 1) + 2u * (pic.origin_y >> 1)) * es,
                             (size_t)pic.stride_cr * ((pic.height >> 1) + 2u * (pic.origin_y >> 1)) * es};
    void *host[3], *dev[3];
    for (int i = 0; i < 3; i++) {
        host[i] = malloc(bytes[i]);
        if (fread(host[i], 1, bytes[i], fi) != bytes[i]) { fprintf(stderr, "comm_consumer: short read\n"); return 2; }
    }
    fclose(fi);

    svthip_ctx *ctx = NULL;
    svthip_comm *comm = NULL;
    CHECK(svthip_create(0, &ctx));
    CHECK(svthip_comm_create(ctx, NULL, 0, 1, &comm)); /* one rank: no id, RCCL is never entered */
    if (svthip_comm_rank(comm) != 0 || svthip_comm_world(comm) != 1) return 4;
    for (int i = 0; i < 3; i++) {
        HIPCHECK(hipMalloc(&dev[i], bytes[i]));
        HIPCHECK(hipMemcpy(dev[i], host[i], bytes[i], hipMemcpyHostToDevice));
    }
    pic.y = dev[0]; pic.cb = dev[1]; pic.cr = dev[2];
    CHECK(svthip_recon_exchange_dev(comm, &pic, NULL));
    CHECK(svthip_synchronize(ctx));
    FILE *fo = fopen(argv[3], "wb");
    if (!fo) { perror(argv[3]); return 2; }
    for (int i = 0; i < 3; i++) {
        HIPCHECK(hipMemcpy(host[i], dev[i], bytes[i], hipMemcpyDeviceToHost));
        fwrite(host[i], 1, bytes[i], fo);
    }
    /* a malformed picture is refused before anything is queued */
    {
        svthip_recon_picture bad = pic;
        bad.stride_y = pic.width;
        if (svthip_recon_exchange_dev(comm, &bad, NULL) != SVTHIP_ERR_BAD_PARAMETER) { fprintf(stderr, "comm_consumer: bad stride accepted\n"); return 5; }
        bad = pic;
        bad.cr = NULL;
        if (svthip_recon_exchange_dev(comm, &bad, NULL) != SVTHIP_ERR_BAD_PARAMETER) { fprintf(stderr, "comm_consumer: cb without cr accepted\n"); return 5; }
    }

    /* ME results of 3 jobs x 28 SBs x 85 PUs: world 1 = this rank's rows are all rows */
    const uint32_t n_jobs = 3, n_sb = 28, rec = 85 * 24;
    const size_t gb = (size_t)n_jobs * n_sb * rec;
    uint8_t *pat = (uint8_t *)malloc(gb), *back = (uint8_t *)malloc(gb);
    for (size_t i = 0; i < gb; i++) pat[i] = (uint8_t)(i * 2654435761u >> 13);
    void *d_local, *d_full;
    HIPCHECK(hipMalloc(&d_local, gb));
    HIPCHECK(hipMalloc(&d_full, gb));
    HIPCHECK(hipMemcpy(d_local, pat, gb, hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(d_full, 0, gb));
    CHECK(svthip_me_gather_results_dev(comm, d_local, d_full, n_jobs, n_sb, rec, NULL));
    CHECK(svthip_synchronize(ctx));
    HIPCHECK(hipMemcpy(back, d_full, gb, hipMemcpyDeviceToHost));
    if (memcmp(pat, back, gb)) { fprintf(stderr, "comm_consumer: gathered rows differ\n"); return 6; }
    fwrite(back, 1, gb, fo);
    fclose(fo);

    svthip_comm_destroy(comm);
    for (int i = 0; i < 3; i++) { HIPCHECK(hipFree(dev[i])); free(host[i]); }
    HIPCHECK(hipFree(d_local));
    HIPCHECK(hipFree(d_full));
    free(pat); free(back);
    svthip_destroy(ctx);
    printf("comm_consumer ok\n");
    return 0;
}

int main(int argc, char **argv)
{
    int rc = 2;
    if (argc >= 2 && !strcmp(argv[1], "plan")) rc = do_plan(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "exchange")) rc = do_exchange(argc, argv);
    if (rc == 2) fprintf(stderr, "usage: comm_consumer plan w h origin sample_bytes world | comm_consumer exchange in.bin out.bin\n");
    return rc;
}
