// Native one-against-all trainer: replaces the `vw --oaa K -b bits --save_resume` process that the reference
// feeds at opal.py:307-327,363-367.  One wave owns one training example (= one fragment);
// per example and class c (labels y_c = +1 for the example's class, -1 otherwise; squared loss; every feature
// value is 1):
//   sweep 1  p_c  = sum_f w[slot(f,c)]                     (phase S of the predict kernel, same order)
//            err_c = y_c - clip(p_c, [min_label,max_label]);  s = eta / sqrt(#features)
//   sweep 2  G[slot(f,c)] += 4 err_c^2;  ppu_c += 1/sqrt(G[slot(f,c)])        (AdaGrad accumulators, VW --adaptive)
//            u_c = s*ppu_c < 1e-6 ? 2 err_c s : err_c (1 - exp(-2 s ppu_c)) / ppu_c   (VW --invariant, squared loss)
//   sweep 3  w[slot(f,c)] += u_c / sqrt(G[slot(f,c)])      (then optional l2 decay / l1 soft-threshold)
// slot(f,c) = (h_f*K + c) mod 2^b, features in the reference's order, the constant feature last.
// This is a VW-like rule, not VW's exact one (VW is absent: parity unpinned); oracle/opal_oracle.c restates it
// sequentially and tests compare the two in `sequential` mode (one wave, examples in order).
// In the default mode all waves run concurrently (Hogwild): k-mer rows are updated with plain read-modify-write,
// the constant row (shared by every example) with float atomics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "opal_device.h"
#include "opal_internal.h"

namespace {

// 1/sqrt(x) as one v_rsq_f32 (1 ulp) instead of a correctly rounded sqrt + divide (~60 instructions per element):
// the per-weight learning rate does not need the last bit, and the sweeps were ALU-bound without it.
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }

struct TrainArgs {
    float eta, l1, l2, min_label, max_label;
    int sequential;
    int phases;   // diagnostic: bit0 = run sweep 2, bit1 = run sweep 3 (always 3 outside profiling experiments)
};

// AL: K is a multiple of VEC (every lane wraps its own address); otherwise contiguous rows / rows that reach the table's end
// are told apart (RowCross), see opal_dtrain.hip.
template <int VEC, int NV, int U, bool AL>
__global__ __launch_bounds__(BLOCK) void train_kernel(Batch B, Geom G, float *__restrict__ w, float *__restrict__ g2,
                                                       uint32_t mask, int K, uint32_t const_idx,
                                                       const int32_t *__restrict__ labels, TrainArgs ta,
                                                       int32_t *__restrict__ out_pred, unsigned long long *__restrict__ queue)
{
    using V = typename VecOf<VEC>::type;
    constexpr int CH = WAVE * VEC * NV;
    __shared__ __attribute__((aligned(16))) uint8_t s_seq[WPB][2 * SEQB];
    __shared__ __attribute__((aligned(16))) uint32_t s_hb[WPB][HB + HB_PAD];
    __shared__ uint32_t s_tab[KTAB];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    fill_kmer4_table(s_tab);
    uint32_t *hb = s_hb[wv];
    if (ta.sequential && ((uint64_t)blockIdx.x * WPB + wv) != 0) return;    // one wave, examples in order
    const uint32_t Ku = (uint32_t)K;

    // concurrent mode: examples are drawn one at a time from the work queue (an example takes 0.2-4 ms, so the
    // atomic is free and a coarser hand-out only lengthens the tail); sequential mode: 0, 1, 2, ...
    for (uint64_t frag = ta.sequential ? 0 : next_chunk(queue, lane, 1); frag < B.n;
         frag = ta.sequential ? frag + 1 : next_chunk(queue, lane, 1)) {
        const uint32_t label = (uint32_t)(labels[frag] - 1);   // VW class ids are 1-based
        float best = 0.0f;
        uint32_t best_c = 0xffffffffu;
        for (uint32_t c0 = 0; c0 < Ku; c0 += CH) {
            uint32_t loff[NV], cls[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) {
                cls[j] = c0 + (uint32_t)(j * WAVE + lane) * VEC;
                loff[j] = min(cls[j], (Ku - 1u) & ~(uint32_t)(VEC - 1));   // lanes past the row re-read its last vector (which may be partial)
            }
            // ---- sweep 1: scores (identical to predict)
            V acc[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) acc[j] = V{};
            uint32_t nfeat = 0;
            walk_fragment(B, G, frag, mask, true, const_idx, s_seq[wv], hb, s_tab, lane, [&](uint32_t cnt) {
                if (lane < HB_PAD) hb[cnt + lane] = const_idx;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // contiguous rows: the predict kernel's pipeline on unwrapped addresses; a flush that holds a row which passes
                // the table's end goes slot by slot (the mirror behind the table may be stale while training)
                if constexpr (AL) stream_rows<VEC, NV, U, 1, false>(w, hb, cnt, Ku, mask, loff, acc);
                else if (!any_row_crosses(hb, cnt, Ku, mask, lane)) stream_rows<VEC, NV, U, 2, false>(w, hb, cnt, Ku, mask, loff, acc);
                else stream_rows_wrapped<VEC, NV>(w, hb, cnt, Ku, mask, cls, acc);
                nfeat += cnt;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            });
            float err[NV][VEC], gsq[NV][VEC], ppu[NV][VEC], upd[NV][VEC];
#pragma unroll
            for (int j = 0; j < NV; j++)
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    float p = vget<VEC>(acc[j], e);
                    if (p != p) p = 0.0f;
                    else if (p > ta.max_label) p = ta.max_label;
                    else if (p < ta.min_label) p = ta.min_label;
                    const uint32_t c = cls[j] + e;
                    if (c < Ku && (best_c == 0xffffffffu || p > best)) { best = p; best_c = c; }
                    const float y = (c == label) ? 1.0f : -1.0f;
                    err[j][e] = c < Ku ? y - p : 0.0f;          // slots past the row (K not a multiple of VEC) belong to no class
                    gsq[j][e] = 4.0f * err[j][e] * err[j][e];
                    ppu[j][e] = 0.0f;
                }
            // classes whose clipped prediction already equals the label (err == 0: typically most of the K-1 negatives
            // of a trained model, saturated at -1) change neither G nor w: their lanes skip sweeps 2 and 3 entirely
            const bool reg_on = (ta.l1 != 0.0f) || (ta.l2 != 0.0f);
            bool act[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) {
                bool any = reg_on;
#pragma unroll
                for (int e = 0; e < VEC; e++) any = any || (err[j][e] != 0.0f);
                act[j] = any && (cls[j] < Ku);
            }
            const float s = ta.eta / sqrtf((float)nfeat);
            // Sweeps 2 and 3 keep U2 rows in flight when the rows of a group cannot touch the same slot (VW's layout
            // lets rows overlap: |row_a - row_b| mod 2^b < K); otherwise, and for the constant row in concurrent
            // mode (atomics), the group is processed row by row so that the result equals the sequential rule.
            constexpr int U2 = (NV <= 2) ? 4 : (NV == 4 ? 2 : 1);
            const uint32_t thr = (mask - Ku + 1u) & mask;                       // 2^b - K
            const bool tiny = (uint64_t)Ku * 2ull > (uint64_t)mask + 1ull;      // every pair of rows overlaps
            const RowCross crosses(Ku, mask);                                   // does a row pass the table's end?
            auto independent = [&](const uint32_t (&row)[U2], const uint32_t (&idx)[U2]) -> bool {
                bool ok = !tiny;
#pragma unroll
                for (int x = 0; x < U2; x++) {
                    ok = ok && (ta.sequential || idx[x] != const_idx) && (AL || !crosses(row[x]));   // (rows of a group are contiguous)
#pragma unroll
                    for (int y = x + 1; y < U2; y++) {
                        const uint32_t d = (row[x] - row[y]) & mask;
                        ok = ok && d >= Ku && d <= thr;
                    }
                }
                return ok;
            };
            // Branch-free loads: a load under `if (act[j])` makes the compiler's wait-count pass assume the load may
            // still be pending when its registers are reused, and it then puts `s_waitcnt vmcnt(0)` in front of every
            // load of the group (seen in the ISA: the U2 rows were fetched one after the other).  So the grouped paths
            // load unconditionally (lanes without work read slot 0, an always-cached line), compute on every lane and
            // predicate only the stores.  The values computed by idle lanes are never stored: their err is 0, so
            // their update u is 0, and lanes past K never pass `act`.
            // ---- sweep 2: adaptive accumulators + prediction-per-update
            auto adapt_one = [&](uint32_t idx) {
                const uint32_t row = (idx * Ku) & mask;
                const bool shared_row = (idx == const_idx) && !ta.sequential;
                if (shared_row) {   // every example touches the constant row: returning atomics, no lost adds
#pragma unroll
                    for (int j = 0; j < NV; j++) {
                        if (act[j]) {
#pragma unroll
                            for (int e = 0; e < VEC; e++) {
                                if (cls[j] + (uint32_t)e >= Ku) continue;
                                const float gv = atomicAdd(g2 + (size_t)((row + cls[j] + (uint32_t)e) & mask), gsq[j][e]) + gsq[j][e];
                                ppu[j][e] += (gv > 0.0f) ? rsq(gv) : 0.0f;
                            }
                        }
                    }
                    return;
                }
                float gv[NV][VEC];
                const bool cross = !AL && crosses(row);
#pragma unroll
                for (int j = 0; j < NV; j++) row_vload<VEC, AL>(g2, row, cls[j], Ku, mask, cross, act[j], gv[j]);
#pragma unroll
                for (int j = 0; j < NV; j++) {
#pragma unroll
                    for (int e = 0; e < VEC; e++) {
                        gv[j][e] += gsq[j][e];
                        ppu[j][e] += (gv[j][e] > 0.0f) ? rsq(gv[j][e]) : 0.0f;
                    }
                    if (act[j]) row_vstore<VEC, AL>(g2, row, cls[j], Ku, mask, cross, gv[j]);
                }
            };
            if (ta.phases & 1)
            walk_fragment(B, G, frag, mask, true, const_idx, s_seq[wv], hb, s_tab, lane, [&](uint32_t cnt) {
                uint32_t f = 0;
                if constexpr (U2 > 1) {
                    for (; f + U2 <= cnt; f += U2) {
                        uint32_t idx[U2], row[U2];
#pragma unroll
                        for (int u = 0; u < U2; u++) {
                            idx[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb[f + u]);
                            row[u] = (idx[u] * Ku) & mask;
                        }
                        if (independent(row, idx)) {
                            float gv[U2][NV][VEC];
#pragma unroll
                            for (int u = 0; u < U2; u++) {
#pragma unroll
                                for (int j = 0; j < NV; j++)   // idle lanes read slot 0 (see the note on branch-free loads)
                                    row_vload<VEC, AL>(g2, row[u], cls[j], Ku, mask, false, act[j], gv[u][j]);
                            }
#pragma unroll
                            for (int u = 0; u < U2; u++)
#pragma unroll
                                for (int j = 0; j < NV; j++) {
#pragma unroll
                                    for (int e = 0; e < VEC; e++) {
                                        gv[u][j][e] += gsq[j][e];
                                        ppu[j][e] += (gv[u][j][e] > 0.0f) ? rsq(gv[u][j][e]) : 0.0f;
                                    }
                                    if (act[j]) row_vstore<VEC, AL>(g2, row[u], cls[j], Ku, mask, false, gv[u][j]);
                                }
                        } else {
#pragma unroll
                            for (int u = 0; u < U2; u++) adapt_one(idx[u]);
                        }
                    }
                }
                for (; f < cnt; f++) adapt_one((uint32_t)__builtin_amdgcn_readfirstlane((int)hb[f]));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            });
#pragma unroll
            for (int j = 0; j < NV; j++)
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    const float x = s * ppu[j][e];
                    float u;
                    if (err[j][e] == 0.0f || ppu[j][e] == 0.0f) u = 0.0f;
                    else if (x < 1e-6f) u = 2.0f * err[j][e] * s;
                    else u = err[j][e] * (1.0f - expf(-2.0f * x)) / ppu[j][e];
                    upd[j][e] = u;
                }
            // ---- sweep 3: weight update
            const float decay = 1.0f - s * ta.l2;
            const float shrink = s * ta.l1;
            const bool reg = (ta.l1 != 0.0f) || (ta.l2 != 0.0f);
            auto finish_w = [&](float wold, float d) -> float {
                float nw = wold + d;
                if (reg) {
                    nw *= decay;
                    nw = nw > shrink ? nw - shrink : (nw < -shrink ? nw + shrink : 0.0f);
                }
                return nw;
            };
            auto update_one = [&](uint32_t idx) {
                const uint32_t row = (idx * Ku) & mask;
                const bool shared_row = (idx == const_idx) && !ta.sequential;
                if (shared_row) {   // every example touches the constant row: no plain read-modify-write here, or updates get lost
#pragma unroll
                    for (int j = 0; j < NV; j++) {
                        if (act[j]) {
#pragma unroll
                            for (int e = 0; e < VEC; e++) {
                                if (cls[j] + (uint32_t)e >= Ku) continue;
                                const size_t slot = (size_t)((row + cls[j] + (uint32_t)e) & mask);
                                const float gve = g2[slot];
                                const float d = (gve > 0.0f) ? upd[j][e] * rsq(gve) : 0.0f;
                                if (!reg) {
                                    atomicAdd(w + slot, d);
                                } else {   // decay and soft threshold are not additive: compare-and-swap until this example's step lands
                                    unsigned int *pw = reinterpret_cast<unsigned int *>(w + slot);
                                    unsigned int seen = *pw, assumed;
                                    do {
                                        assumed = seen;
                                        seen = atomicCAS(pw, assumed, __float_as_uint(finish_w(__uint_as_float(assumed), d)));
                                    } while (seen != assumed);
                                }
                            }
                        }
                    }
                    return;
                }
                float gv[NV][VEC], wo[NV][VEC];
                const bool cross = !AL && crosses(row);
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    row_vload<VEC, AL>(g2, row, cls[j], Ku, mask, cross, act[j], gv[j]);
                    row_vload<VEC, AL>(w, row, cls[j], Ku, mask, cross, act[j], wo[j]);
                }
#pragma unroll
                for (int j = 0; j < NV; j++) {
#pragma unroll
                    for (int e = 0; e < VEC; e++) {
                        const float d = (gv[j][e] > 0.0f) ? upd[j][e] * rsq(gv[j][e]) : 0.0f;
                        wo[j][e] = finish_w(wo[j][e], d);
                    }
                    if (act[j]) row_vstore<VEC, AL>(w, row, cls[j], Ku, mask, cross, wo[j]);
                }
            };
            if (ta.phases & 2)
            walk_fragment(B, G, frag, mask, true, const_idx, s_seq[wv], hb, s_tab, lane, [&](uint32_t cnt) {
                uint32_t f = 0;
                if constexpr (U2 > 1) {
                    for (; f + U2 <= cnt; f += U2) {
                        uint32_t idx[U2], row[U2];
#pragma unroll
                        for (int u = 0; u < U2; u++) {
                            idx[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb[f + u]);
                            row[u] = (idx[u] * Ku) & mask;
                        }
                        if (independent(row, idx)) {
                            float gv[U2][NV][VEC], wo[U2][NV][VEC];
#pragma unroll
                            for (int u = 0; u < U2; u++) {
#pragma unroll
                                for (int j = 0; j < NV; j++) {
                                    row_vload<VEC, AL>(g2, row[u], cls[j], Ku, mask, false, act[j], gv[u][j]);
                                    row_vload<VEC, AL>(w, row[u], cls[j], Ku, mask, false, act[j], wo[u][j]);
                                }
                            }
#pragma unroll
                            for (int u = 0; u < U2; u++)
#pragma unroll
                                for (int j = 0; j < NV; j++) {
#pragma unroll
                                    for (int e = 0; e < VEC; e++) {
                                        const float d = (gv[u][j][e] > 0.0f) ? upd[j][e] * rsq(gv[u][j][e]) : 0.0f;
                                        wo[u][j][e] = finish_w(wo[u][j][e], d);
                                    }
                                    if (act[j]) row_vstore<VEC, AL>(w, row[u], cls[j], Ku, mask, false, wo[u][j]);
                                }
                        } else {
#pragma unroll
                            for (int u = 0; u < U2; u++) update_one(idx[u]);
                        }
                    }
                }
                for (; f < cnt; f++) update_one((uint32_t)__builtin_amdgcn_readfirstlane((int)hb[f]));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            });
        }
        if (out_pred) {
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                const float ob = __shfl_xor(best, d, WAVE);
                const uint32_t oc = (uint32_t)__shfl_xor((int)best_c, d, WAVE);
                const bool take = (oc != 0xffffffffu) && (best_c == 0xffffffffu || ob > best || (ob == best && oc < best_c));
                if (take) { best = ob; best_c = oc; }
            }
            if (lane == 0) out_pred[frag] = (int32_t)(best_c + 1u);
        }
    }
}

struct TVariant {
    int vec, nv;
    bool al;
    const void *fn;
};
#define OPAL_TV(VEC, NV, U, AL) {VEC, NV, AL, (const void *)&train_kernel<VEC, NV, U, AL>}
const TVariant kTrainVariants[] = {
    OPAL_TV(1, 1, 8, true),   // K = 1, 3
    OPAL_TV(2, 1, 8, true),   // K = 2
    OPAL_TV(4, 1, 8, true), OPAL_TV(4, 2, 8, true), OPAL_TV(4, 4, 4, true), OPAL_TV(4, 8, 2, true),         // K % 4 == 0
    OPAL_TV(4, 1, 8, false), OPAL_TV(4, 2, 8, false), OPAL_TV(4, 4, 4, false), OPAL_TV(4, 8, 2, false),     // any other K >= 5
};

const TVariant &pick_train_variant(int K)
{
    const int vec = K >= 4 ? 4 : (K % 2 == 0) ? 2 : 1;   // 16-byte vectors for every K >= 4 (dword-aligned, see row_vload)
    const int need = (K + WAVE * vec - 1) / (WAVE * vec);
    const int nv = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
    for (const TVariant &v : kTrainVariants)
        if (v.vec == vec && v.nv == nv && v.al == (K % vec == 0)) return v;
    return kTrainVariants[0];
}

__global__ void mirror_head_kernel2(float *w, uint64_t n_slots, uint32_t pad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pad) w[n_slots + i] = w[i % n_slots];
}

__global__ void scatter_weights_kernel(float *__restrict__ w, const uint32_t *__restrict__ idx, const float *__restrict__ val,
                                       uint64_t nnz, uint64_t n_slots)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride)
        if (idx[i] < n_slots) w[idx[i]] = val[i];
}

int refresh_mirror(opal_ctx *ctx)
{
    const uint32_t pad = (uint32_t)ctx->K + 4;
    hipLaunchKernelGGL(mirror_head_kernel2, dim3((pad + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_w, ctx->n_slots, pad);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    return OPAL_OK;
}

int alloc_floats(opal_ctx *ctx, float **p, uint64_t *cap, uint64_t need)
{
    if (*cap < need) {
        if (*p) { OPAL_HIP_CHECK(ctx, hipFree(*p)); *p = nullptr; *cap = 0; }
        hipError_t e = hipMalloc((void **)p, need * sizeof(float));
        if (e != hipSuccess) return opal::fail(ctx, OPAL_ENOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
        *cap = need;
    }
    return OPAL_OK;
}

// dst[idx[i]] = val[i] for nnz host pairs (dst: n_slots floats on the device, already zeroed on the ctx stream); synchronises.
int scatter_pairs(opal_ctx *ctx, float *dst, const uint32_t *idx, const float *val, uint64_t nnz, const char *who)
{
    if (!nnz) return OPAL_OK;
    struct Staging {   // released on every path
        uint32_t *idx = nullptr;
        float *val = nullptr;
        ~Staging() { if (idx) (void)hipFree(idx); if (val) (void)hipFree(val); }
    } stg;
    hipError_t e = hipMalloc((void **)&stg.idx, nnz * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&stg.val, nnz * sizeof(float));
    if (e != hipSuccess) return opal::fail(ctx, OPAL_ENOMEM, std::string(who) + ": hipMalloc failed: " + hipGetErrorString(e));
    e = hipMemcpyAsync(stg.idx, idx, nnz * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(stg.val, val, nnz * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        const int grid = (int)std::min<uint64_t>((nnz + 255) / 256, 65536);
        hipLaunchKernelGGL(scatter_weights_kernel, dim3(grid), dim3(256), 0, ctx->stream, dst, stg.idx, stg.val, nnz, ctx->n_slots);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return opal::fail(ctx, OPAL_EHIP, std::string(who) + ": " + hipGetErrorString(e));
    return OPAL_OK;
}

}  // namespace

extern "C" int opal_train_begin(opal_ctx *ctx, int K, int bits, int resume)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_train_begin: null ctx");
    if (bits < 1 || bits > 32) return opal::fail(ctx, OPAL_EINVAL, "opal_train_begin: need 1 <= bits <= 32");
    if (K < 1 || K > (1 << 20)) return opal::fail(ctx, OPAL_EINVAL, "opal_train_begin: need 1 <= K <= 2^20");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n_slots = 1ull << bits;
    const uint64_t need = n_slots + (uint64_t)K + 4;
    const bool keep = resume && ctx->d_w && ctx->K == K && ctx->bits == bits;
    if (resume && !keep) return opal::fail(ctx, OPAL_ESTATE, "opal_train_begin: resume needs weights with the same K and bits loaded");
    if (!keep) opal::drop_adopted_table(ctx);   // an adopted table is trained in place when resuming, otherwise left alone
    int rc = alloc_floats(ctx, &ctx->d_w, &ctx->w_alloc_floats, need);
    if (rc) return rc;
    rc = alloc_floats(ctx, &ctx->d_g2, &ctx->g2_alloc_floats, n_slots);
    if (rc) return rc;
    ctx->n_slots = n_slots; ctx->K = K; ctx->bits = bits;
    if (!keep) {
        OPAL_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_w, 0, need * sizeof(float), ctx->stream));
        OPAL_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_g2, 0, n_slots * sizeof(float), ctx->stream));
    } else if (!ctx->g2_valid) {
        OPAL_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_g2, 0, n_slots * sizeof(float), ctx->stream));
    }
    ctx->g2_valid = true;
    return OPAL_OK;
}

extern "C" int opal_train_batch(opal_ctx *ctx, const uint8_t *d_reads, uint64_t reads_bytes, const uint64_t *d_frag_off,
                                const uint32_t *d_frag_len, const int32_t *d_labels, uint64_t n, int reverse, float eta,
                                float l1, float l2, int sequential, int32_t *d_out_pred)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_train_batch: null ctx");
    if (ctx->P == 0) return opal::fail(ctx, OPAL_ESTATE, "opal_train_batch: opal_set_patterns has not been called");
    if (!ctx->d_w || !ctx->d_g2 || !ctx->g2_valid) return opal::fail(ctx, OPAL_ESTATE, "opal_train_batch: call opal_train_begin first");
    if (n && (!d_reads || !d_frag_off || !d_frag_len || !d_labels)) return opal::fail(ctx, OPAL_EINVAL, "opal_train_batch: null batch pointer");
    if (!(eta > 0.0f) || l1 < 0.0f || l2 < 0.0f) return opal::fail(ctx, OPAL_EINVAL, "opal_train_batch: need eta > 0, l1 >= 0, l2 >= 0");
    if (sequential < 0 || sequential > 2) return opal::fail(ctx, OPAL_EINVAL, "opal_train_batch: mode is 0 (concurrent), 1 (one wave) or 2 (deterministic groups)");
    if (n == 0) return OPAL_OK;
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (sequential == 2) return opal::train_batch_groups(ctx, d_reads, reads_bytes, d_frag_off, d_frag_len, d_labels, n, reverse, eta, l1, l2, d_out_pred);
    Batch B{d_reads, reads_bytes, d_frag_off, d_frag_len, n};
    Geom G{ctx->d_pos, ctx->k, ctx->P, ctx->t, (ctx->t + 3) & ~3, reverse != 0, ctx->hash_all, ctx->p0_run};
    const uint32_t mask = ctx->bits == 32 ? 0xffffffffu : ((1u << ctx->bits) - 1u);
    const uint32_t const_idx = OPAL_VW_CONSTANT & mask;
    const TVariant &v = pick_train_variant(ctx->K);
    int per_cu = 0;
    OPAL_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v.fn, BLOCK, 0));
    if (per_cu < 1) per_cu = 1;
    const uint64_t want = (n + WPB - 1) / WPB;
    int grid = sequential ? 1 : (int)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)ctx->num_cus * per_cu));
    float *w = ctx->d_w, *g2 = ctx->d_g2;
    int K = ctx->K;
    int phases = 3;
#ifdef OPAL_EXPERIMENTS   // profiling builds only: the shipped library reads no environment variable
    if (const char *e = getenv("OPAL_TRAIN_PHASES")) phases = atoi(e);
#endif
    TrainArgs ta{eta, l1, l2, ctx->min_label, ctx->max_label, sequential != 0, phases};
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_queue, 0, sizeof(unsigned long long), ctx->stream));
    void *args[] = {&B, &G, &w, &g2, (void *)&mask, &K, (void *)&const_idx, &d_labels, &ta, &d_out_pred, &ctx->d_queue};
    OPAL_HIP_CHECK(ctx, hipLaunchKernel(v.fn, dim3(grid), dim3(BLOCK), args, 0, ctx->stream));
    // keep the wrap padding coherent so that opal_predict can be called between batches
    return refresh_mirror(ctx);
}

extern "C" int opal_get_weights(opal_ctx *ctx, float *dst, int dst_is_device)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_get_weights: null ctx");
    if (!ctx->d_w || ctx->K == 0) return opal::fail(ctx, OPAL_ESTATE, "opal_get_weights: no weights");
    if (!dst) return opal::fail(ctx, OPAL_EINVAL, "opal_get_weights: null dst");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    OPAL_HIP_CHECK(ctx, hipMemcpyAsync(dst, ctx->d_w, ctx->n_slots * sizeof(float),
                                       dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return OPAL_OK;
}

extern "C" int opal_load_weights_sparse(opal_ctx *ctx, const uint32_t *idx, const float *val, uint64_t nnz, int K, int bits)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_load_weights_sparse: null ctx");
    if (nnz && (!idx || !val)) return opal::fail(ctx, OPAL_EINVAL, "opal_load_weights_sparse: null arrays");
    if (bits < 1 || bits > 32) return opal::fail(ctx, OPAL_EINVAL, "opal_load_weights_sparse: need 1 <= bits <= 32");
    if (K < 1 || K > (1 << 20)) return opal::fail(ctx, OPAL_EINVAL, "opal_load_weights_sparse: need 1 <= K <= 2^20");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n_slots = 1ull << bits;
    const uint64_t need = n_slots + (uint64_t)K + 4;
    opal::drop_adopted_table(ctx);
    int rc = alloc_floats(ctx, &ctx->d_w, &ctx->w_alloc_floats, need);
    if (rc) return rc;
    ctx->n_slots = n_slots; ctx->K = K; ctx->bits = bits; ctx->g2_valid = false;
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_w, 0, need * sizeof(float), ctx->stream));
    rc = scatter_pairs(ctx, ctx->d_w, idx, val, nnz, "opal_load_weights_sparse");
    if (rc) return rc;
    rc = refresh_mirror(ctx);
    if (rc) return rc;
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return OPAL_OK;
}

extern "C" int opal_get_accumulators(opal_ctx *ctx, float *dst, int dst_is_device)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_get_accumulators: null ctx");
    if (!ctx->d_g2 || !ctx->g2_valid || ctx->K == 0) return opal::fail(ctx, OPAL_ESTATE, "opal_get_accumulators: no training state (call opal_train_begin first)");
    if (!dst) return opal::fail(ctx, OPAL_EINVAL, "opal_get_accumulators: null dst");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    OPAL_HIP_CHECK(ctx, hipMemcpyAsync(dst, ctx->d_g2, ctx->n_slots * sizeof(float),
                                       dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return OPAL_OK;
}

extern "C" int opal_load_accumulators_sparse(opal_ctx *ctx, const uint32_t *idx, const float *val, uint64_t nnz)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_load_accumulators_sparse: null ctx");
    if (!ctx->d_w || ctx->K == 0) return opal::fail(ctx, OPAL_ESTATE, "opal_load_accumulators_sparse: load the weights first");
    if (nnz && (!idx || !val)) return opal::fail(ctx, OPAL_EINVAL, "opal_load_accumulators_sparse: null arrays");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    int rc = alloc_floats(ctx, &ctx->d_g2, &ctx->g2_alloc_floats, ctx->n_slots);
    if (rc) return rc;
    ctx->g2_valid = false;
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_g2, 0, ctx->n_slots * sizeof(float), ctx->stream));
    rc = scatter_pairs(ctx, ctx->d_g2, idx, val, nnz, "opal_load_accumulators_sparse");
    if (rc) return rc;
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->g2_valid = true;     // opal_train_begin(resume=1) keeps these accumulators
    return OPAL_OK;
}

// ------------------------------------------------------------------------------------------ evaluation counts
// Replaces the per-line string compare + pandas groupby of opal.evaluate_predictions (reference opal.py:111-131):
// per reference class, how many fragments there are and how many were predicted correctly.  Classes are dense
// integer ids (the host maps taxid strings through vw-dico.txt, reference opal.py:76-88).  Block-private LDS
// histograms, one global atomic per non-empty bucket per block.
namespace {
constexpr int EVAL_LDS_CLASSES = 8192;
__global__ __launch_bounds__(256) void eval_counts_kernel(const int32_t *__restrict__ pred, const int32_t *__restrict__ ref,
                                                           uint64_t n, uint32_t n_classes,
                                                           unsigned long long *__restrict__ correct,
                                                           unsigned long long *__restrict__ total)
{
    __shared__ uint32_t s_tot[EVAL_LDS_CLASSES];
    __shared__ uint32_t s_cor[EVAL_LDS_CLASSES];
    const bool use_lds = n_classes <= EVAL_LDS_CLASSES;
    if (use_lds) {
        for (uint32_t i = threadIdx.x; i < n_classes; i += blockDim.x) { s_tot[i] = 0; s_cor[i] = 0; }
        __syncthreads();
    }
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t r = (uint32_t)ref[i];
        if (r >= n_classes) continue;                 // host guarantees 0 <= ref < n_classes; never index outside
        const bool ok = pred[i] == ref[i];
        if (use_lds) {
            atomicAdd(&s_tot[r], 1u);
            if (ok) atomicAdd(&s_cor[r], 1u);
        } else {
            atomicAdd(&total[r], 1ull);
            if (ok) atomicAdd(&correct[r], 1ull);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_classes; i += blockDim.x) {
            if (s_tot[i]) atomicAdd(&total[i], (unsigned long long)s_tot[i]);
            if (s_cor[i]) atomicAdd(&correct[i], (unsigned long long)s_cor[i]);
        }
    }
}
}  // namespace

extern "C" int opal_eval_counts(opal_ctx *ctx, const int32_t *d_pred, const int32_t *d_ref, uint64_t n, uint32_t n_classes,
                                uint64_t *d_correct, uint64_t *d_total)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_eval_counts: null ctx");
    if (n_classes == 0 || n_classes > (1u << 24)) return opal::fail(ctx, OPAL_EINVAL, "opal_eval_counts: need 1 <= n_classes <= 2^24");
    if (!d_correct || !d_total || (n && (!d_pred || !d_ref))) return opal::fail(ctx, OPAL_EINVAL, "opal_eval_counts: null pointer");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(d_correct, 0, (size_t)n_classes * sizeof(uint64_t), ctx->stream));
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(d_total, 0, (size_t)n_classes * sizeof(uint64_t), ctx->stream));
    if (n == 0) return OPAL_OK;
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, (uint64_t)ctx->num_cus * 4));
    hipLaunchKernelGGL(eval_counts_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_pred, d_ref, n, n_classes,
                       (unsigned long long *)d_correct, (unsigned long long *)d_total);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    return OPAL_OK;
}
