/*
 * predict_c.c -- a plain C client of the C-ABI (include/opal_hip.h): no Python, no torch.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/predict_c.c -o examples/predict_c \
 *       -Lopal_amd -lopal_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../opal_amd' -Wl,-rpath,/opt/rocm/lib
 *   examples/predict_c patterns.txt k weights.f32 K bits reads.txt out_classes.txt [reverse]
 *
 * reads.txt: one sequence per line (what fasta_reader yields); weights.f32: 2^bits little-endian floats.
 * Does what `opal.py predict` does around the `vw -t` process (reference opal.py:449-469) with device buffers
 * allocated by hipMalloc: pack -> copy to HBM -> opal_predict -> copy class ids back -> one id per line.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "opal_hip.h"

#define CHECK_OPAL(call, ctx)                                                              \
    do {                                                                                   \
        int rc__ = (call);                                                                 \
        if (rc__ != OPAL_OK) {                                                             \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, opal_last_error(ctx));    \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)
#define CHECK_HIP(call)                                                                    \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e__));             \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static char *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 1);
    if (buf && fread(buf, 1, (size_t)sz, f) != (size_t)sz) { free(buf); buf = NULL; }
    fclose(f);
    if (buf) { buf[sz] = 0; *n = (size_t)sz; }
    return buf;
}

int main(int argc, char **argv)
{
    if (argc < 8) {
        fprintf(stderr, "usage: %s patterns.txt k weights.f32 K bits reads.txt out.txt [reverse]\n", argv[0]);
        return 2;
    }
    const int k = atoi(argv[2]), K = atoi(argv[4]), bits = atoi(argv[5]);
    const int reverse = argc > 8 ? atoi(argv[8]) : 1;

    /* patterns.txt: "<P> <t>" then P rows of t positions (util/fasta2skm.py:120-134) */
    size_t pn = 0;
    char *ptxt = slurp(argv[1], &pn);
    if (!ptxt) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    int P = 0, t = 0, consumed = 0;
    if (sscanf(ptxt, "%d %d%n", &P, &t, &consumed) != 2 || P < 1 || t < 1) { fprintf(stderr, "bad patterns header\n"); return 1; }
    int32_t *pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)P * (size_t)t);
    char *cur = ptxt + consumed;
    for (int i = 0; i < P * t; i++) pos[i] = (int32_t)strtol(cur, &cur, 10);

    /* weights */
    size_t wn = 0;
    float *w = (float *)slurp(argv[3], &wn);
    if (!w || wn != ((size_t)1 << bits) * sizeof(float)) { fprintf(stderr, "weights file must hold 2^bits floats\n"); return 1; }

    /* reads: one per line -> concatenated bytes + offsets */
    size_t rn = 0;
    char *rtxt = slurp(argv[6], &rn);
    if (!rtxt) { fprintf(stderr, "cannot read %s\n", argv[6]); return 1; }
    uint64_t n = 0;
    for (size_t i = 0; i < rn; i++) n += rtxt[i] == '\n';
    uint64_t *seq_off = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
    uint8_t *blob = (uint8_t *)malloc(rn + 1);
    uint64_t o = 0, r = 0;
    seq_off[0] = 0;
    for (size_t i = 0; i < rn; i++) {
        if (rtxt[i] == '\n') seq_off[++r] = o;
        else blob[o++] = (uint8_t)rtxt[i];
    }

    uint64_t *frag_off = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
    uint32_t *frag_len = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    uint64_t nbytes = 0;
    CHECK_OPAL(opal_pack_reads(blob, seq_off, n, NULL, 0, frag_off, frag_len, &nbytes), NULL);
    uint8_t *packed = (uint8_t *)malloc(nbytes + 1);
    CHECK_OPAL(opal_pack_reads(blob, seq_off, n, packed, nbytes, frag_off, frag_len, &nbytes), NULL);

    opal_ctx *ctx = NULL;
    CHECK_OPAL(opal_ctx_create(0, NULL, &ctx), NULL);
    CHECK_OPAL(opal_set_patterns(ctx, pos, P, t, k), ctx);
    CHECK_OPAL(opal_load_weights(ctx, w, (uint64_t)1 << bits, K, bits), ctx);

    uint8_t *d_reads = NULL;
    uint64_t *d_off = NULL;
    uint32_t *d_len = NULL;
    int32_t *d_cls = NULL;
    CHECK_HIP(hipMalloc((void **)&d_reads, nbytes + 1));
    CHECK_HIP(hipMalloc((void **)&d_off, sizeof(uint64_t) * (n + 1)));
    CHECK_HIP(hipMalloc((void **)&d_len, sizeof(uint32_t) * (n + 1)));
    CHECK_HIP(hipMalloc((void **)&d_cls, sizeof(int32_t) * (n + 1)));
    CHECK_HIP(hipMemcpy(d_reads, packed, nbytes, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_off, frag_off, sizeof(uint64_t) * n, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_len, frag_len, sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    CHECK_OPAL(opal_predict(ctx, d_reads, nbytes, d_off, d_len, n, reverse, d_cls, NULL), ctx);
    CHECK_OPAL(opal_sync(ctx), ctx);
    int32_t *cls = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    CHECK_HIP(hipMemcpy(cls, d_cls, sizeof(int32_t) * n, hipMemcpyDeviceToHost));

    FILE *out = fopen(argv[7], "w");
    if (!out) { fprintf(stderr, "cannot write %s\n", argv[7]); return 1; }
    for (uint64_t i = 0; i < n; i++) fprintf(out, "%d\n", cls[i]);   /* what `vw -p` writes */
    fclose(out);
    printf("%llu fragments scored with %s\n", (unsigned long long)n, opal_predict_kernel_name(ctx));

    hipFree(d_reads); hipFree(d_off); hipFree(d_len); hipFree(d_cls);
    opal_ctx_destroy(ctx);
    free(cls); free(packed); free(frag_len); free(frag_off); free(blob); free(seq_off); free(rtxt); free(w); free(pos); free(ptxt);
    return 0;
}
