// Internal declarations shared by the host-side and device-side halves of libopal_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "opal_hip.h"

struct opal_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    // patterns (util/fasta2skm.py:120-134)
    int k = 0, P = 0, t = 0;
    uint8_t *d_pos = nullptr;  // P*t positions, each < k <= 256
    uint32_t *d_fast = nullptr; // per-pattern records of the emit fast path (FASTPAT_DW dwords each), nullptr when the geometry has none
    bool fast_canonP = false;   // 3 <= P <= 9, pattern 0 a run of 16, the others gathered: emit_canonP_kernel<P>
    int fast_pair = 0;          // the gathered rows can be read two letters to a word (block_addr_pair): 1, or 2 = with a spare word
    bool fast_canon2 = false;   // P == 2, pattern 0 a run of 16, pattern 1 gathered: the emit kernel's straight-line step
    int p0_run = -1;           // first position of pattern 0 when it is a run of consecutive positions, else -1
    // weights (VW regressor, compact `w` components)
    float *d_w = nullptr;      // 2^bits + pad floats: first K(+3) floats mirrored at the end so rows never wrap
    uint64_t w_alloc_floats = 0;
    bool w_owned = true;       // false: d_w is the caller's buffer (opal_adopt_weights_device), never freed or grown here
    float *d_g2 = nullptr;     // trainer: AdaGrad accumulators, 2^bits floats (canonical slots only)
    uint64_t g2_alloc_floats = 0;
    bool g2_valid = false;
    uint64_t n_slots = 0;
    int K = 0, bits = 0;
    // VW switches
    int hash_all = 0, add_constant = 1, clip = 1;
    float min_label = -1.0f, max_label = 1.0f;
    // tuning (opal_set_tuning): predict row-load policy -1 = by table size, 0 = default loads, 1 = non-temporal; emit draw size 0 = default
    int tune_nt = -1;
    uint64_t tune_qchunk = 0;
    uint64_t train_group = 0;     // examples per group of the deterministic trainer, 0 = default (8192)
    void *dtrain = nullptr;       // buffers of the deterministic trainer (opal_dtrain.hip), grown on demand
    int tune_emit_kt = 0;         // experiment: 2 / 3 = emit_canon2_kernel with per-base term arrays for pattern 0
    bool tune_emit_fast = true;   // false: opal_features uses the generic kernel only (tests compare the two)
    // scratch for the offsets scan
    unsigned long long *d_queue = nullptr;   // work-queue counter of the predict kernel (allocated with the ctx)
    uint64_t *d_block_sums = nullptr;
    unsigned long long *d_static_sums = nullptr;   // this device's copy of the static scratch (hipGetSymbolAddress; never freed)
    uint64_t block_sums_cap = 0;
    // scratch of the FASTA ingest (grown on demand)
    void *ing_tiles = nullptr, *ing_rec = nullptr;
    uint64_t ing_tiles_cap = 0, ing_rec_cap = 0;
    std::string err;
};

namespace opal {
int fail(opal_ctx *ctx, int code, const std::string &msg);
// the deterministic trainer (opal_dtrain.hip)
void dtrain_free(opal_ctx *ctx);
uint32_t dtrain_pshift(int K, int bits);
int train_batch_groups(opal_ctx *ctx, const uint8_t *d_reads, uint64_t reads_bytes, const uint64_t *d_frag_off, const uint32_t *d_frag_len,
                       const int32_t *d_labels, uint64_t n, int reverse, float eta, float l1, float l2, int32_t *d_out_pred);
// forgets a table the ctx does not own, so that the allocation paths below start from "no table"
inline void drop_adopted_table(opal_ctx *ctx)
{
    if (!ctx->w_owned) { ctx->d_w = nullptr; ctx->w_alloc_floats = 0; ctx->w_owned = true; }
}
}  // namespace opal

#define OPAL_HIP_CHECK(ctx, call)                                                                              \
    do {                                                                                                       \
        hipError_t e__ = (call);                                                                               \
        if (e__ != hipSuccess)                                                                                 \
            return opal::fail((ctx), OPAL_EHIP, std::string(#call) + ": " + hipGetErrorString(e__));            \
    } while (0)
