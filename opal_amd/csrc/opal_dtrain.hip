// Deterministic concurrent mode of the native one-against-all trainer (opal_train_batch with sequential == 2).
//
// Why not "the sequential rule, run concurrently": VW's stream (opal.py:363-367) is serial through the constant row,
// which every example reads and writes, and VW's row layout makes unrelated rows share slots: at K = 356, b = 31 the
// 549 rows of 16 random examples already overlap somewhere (rows are K slots wide and start anywhere: (549*16)^2 / 2 *
// 2K / 2^b = 13 overlapping pairs), so groups of slot-disjoint examples do not exist at any useful size.  Atomics are
// no way out either: integer or float atomic adds on random rows of an 8 GiB table run at 0.2 TB/s on this GPU, 20x
// below plain read-modify-write (tools/rmw_probe.hip, profiles/r02_rmw_probe), and float adds do not commute bit for bit.
//
// The rule here is group-synchronous and owner-computes; every step has ONE defined order, so two runs give identical
// tables, and oracle/opal_oracle.c (orc_train_groups) restates it on the CPU:
//   the stream is cut into groups of GS consecutive examples; per group
//   A  (one wave per example, tables read-only)   p_c = sum_f w[slot(f,c)] in feature order, constant last;
//      err_c = y_c - clip(p_c);  q_c = 4 err_c^2;  ppu_c = sum_f 1/sqrt(G[slot(f,c)] + q_c);
//      u_c = the invariant squared-loss step of opal_train.hip from (err_c, ppu_c, s = eta / sqrt(#features))
//   S  the group's k-mer rows (row start, example), in (example, feature) order, are stably sorted by PARTITION of the
//      slot space (2^pshift slots each, at least K: a row spills into the next partition at most)
//   B  one wave per partition, even partitions first, then odd ones (so concurrently worked rows never share a slot):
//      its rows in sorted order:  G[slot] += q_c;  w[slot] += u_c / sqrt(G[slot])   (+ l2 decay, l1 threshold)
//   C  the constant row, examples in order:  G += q_c;  w += u_c / sqrt(G)
// Like the concurrent mode it is this project's own VW-like rule (VW absent: parity unpinned); unlike it, an example
// sees the table as it was when its group started (the concurrent mode's waves see whatever 5 000 others have written).
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "opal_device.h"
#include "opal_internal.h"

namespace {

__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }

struct DArgs {
    float eta, l1, l2, min_label, max_label;
};

// w after one step d (then optional l2 decay and l1 soft threshold, as opal_train.hip's finish_w)
__device__ __forceinline__ float step_w(float wold, float d, bool reg, float decay, float shrink)
{
    float nw = wold + d;
    if (reg) {
        nw *= decay;
        nw = nw > shrink ? nw - shrink : (nw < -shrink ? nw + shrink : 0.0f);
    }
    return nw;
}

// ------------------------------------------------------------------------------------------ phase A
// One wave per example of the group.  Writes, per example e: upd[e][K], gsq[e][K], sfac[e] = eta / sqrt(#features), the
// row start of every k-mer feature into keys[] (value = e) at the example's place in the group's (example, feature)
// order, and the progressive prediction.  Reads w and G, writes neither.
// AL: K is a multiple of VEC (a vector never straddles the table's end: every lane wraps its own address, as before round 3);
// otherwise rows are split into contiguous ones (vectors on unwrapped, dword-aligned addresses) and the rare ones that reach
// the table's end (slot by slot), see RowCross.
template <int VEC, int NV, int U, bool AL>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(NV >= 8 ? 2 : 1))) void dtrain_score_kernel(Batch B, Geom G, const float *__restrict__ w, const float *__restrict__ g2,
                                                              uint32_t mask, int K, uint32_t const_idx,
                                                              const int32_t *__restrict__ labels, DArgs ta, uint64_t e0, uint32_t gn,
                                                              const uint64_t *__restrict__ foff, float *__restrict__ upd_out,
                                                              float *__restrict__ gsq_out, float *__restrict__ sfac,
                                                              uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                              int32_t *__restrict__ out_pred, unsigned long long *__restrict__ queue,
                                                              unsigned long long *__restrict__ amask, uint32_t nit)
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
    const uint32_t Ku = (uint32_t)K;
    const bool reg_on = (ta.l1 != 0.0f) || (ta.l2 != 0.0f);
    for (uint64_t e = first_chunk(wv, 1); e < gn; e = next_chunk_after(queue, lane, 1, gn)) {
        const uint64_t frag = e0 + e;
        const uint32_t label = (uint32_t)(labels[frag] - 1);
        const uint32_t kbase = (uint32_t)(foff[frag] - foff[e0]);
        const uint32_t nfe = (uint32_t)(foff[frag + 1] - foff[frag]);      // k-mer features (the constant is not a key)
        float best = 0.0f;
        uint32_t best_c = 0xffffffffu;
        for (uint32_t c0 = 0; c0 < Ku; c0 += CH) {
            uint32_t loff[NV], cls[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) {
                cls[j] = c0 + (uint32_t)(j * WAVE + lane) * VEC;
                loff[j] = min(cls[j], (Ku - 1u) & ~(uint32_t)(VEC - 1));   // lanes past the row re-read its last vector (which may be partial)
            }
            V acc[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) acc[j] = V{};
            uint32_t nfeat = 0;
            walk_fragment(B, G, frag, mask, true, const_idx, s_seq[wv], hb, s_tab, lane, [&](uint32_t cnt) {
                if (lane < HB_PAD) hb[cnt + lane] = const_idx;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (c0 == 0) {   // the rows of this flush: everything but the constant, which is the example's very last entry
                    for (uint32_t i = lane; i < cnt; i += WAVE) {
                        const uint32_t at = nfeat + i;
                        if (at < nfe) { keys[kbase + at] = (hb[i] * Ku) & mask; vals[kbase + at] = (uint32_t)e; }
                    }
                }
                // contiguous rows: the predict kernel's pipeline, unwrapped addresses (a vector may reach up to VEC - 1 slots past its
                // row: masked below); a flush that holds a row which passes the table's end goes slot by slot
                if constexpr (AL) stream_rows<VEC, NV, U, 1, false>(w, hb, cnt, Ku, mask, loff, acc);
                else if (!any_row_crosses(hb, cnt, Ku, mask, lane)) stream_rows<VEC, NV, U, 2, false>(w, hb, cnt, Ku, mask, loff, acc);
                else stream_rows_wrapped<VEC, NV>(w, hb, cnt, Ku, mask, cls, acc);
                nfeat += cnt;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            });
            if (c0 == 0) {
                // a fragment that reaches outside the buffer hashes to nothing but owns nfe keys: mark them void
                for (uint32_t at = (nfeat ? nfeat - 1u : 0u) + lane; at < nfe; at += WAVE) { keys[kbase + at] = 0u; vals[kbase + at] = 0xffffffffu; }
                if (lane == 0) sfac[e] = ta.eta / sqrtf((float)nfeat);
            }
            float err[NV][VEC], gsq[NV][VEC], ppu[NV][VEC];
            bool act[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) {
                bool any = reg_on;
#pragma unroll
                for (int x = 0; x < VEC; x++) {
                    float p = vget<VEC>(acc[j], x);
                    if (p != p) p = 0.0f;
                    else if (p > ta.max_label) p = ta.max_label;
                    else if (p < ta.min_label) p = ta.min_label;
                    const uint32_t c = cls[j] + x;
                    if (c < Ku && (best_c == 0xffffffffu || p > best)) { best = p; best_c = c; }
                    const float y = (c == label) ? 1.0f : -1.0f;
                    err[j][x] = c < Ku ? y - p : 0.0f;         // slots past the row (K not a multiple of VEC) belong to no class
                    gsq[j][x] = 4.0f * err[j][x] * err[j][x];
                    ppu[j][x] = 0.0f;
                    any = any || (err[j][x] != 0.0f);
                }
                act[j] = any && (cls[j] < Ku);
            }
            const float s = ta.eta / sqrtf((float)nfeat);
            // ppu: a read-only sweep over the accumulator rows, 4 rows in flight, loads unconditional (idle lanes read slot 0)
            walk_fragment(B, G, frag, mask, true, const_idx, s_seq[wv], hb, s_tab, lane, [&](uint32_t cnt) {
                constexpr int U2 = (NV <= 2) ? 4 : (NV == 4 ? 2 : 1);
                if (!AL && any_row_crosses(hb, cnt, Ku, mask, lane)) {
                    // a row of this flush passes the table's end (rare on real tables): the whole flush slot by slot, a vector at a time
                    for (uint32_t f = 0; f < cnt; f++) {
                        const uint32_t row = ((uint32_t)__builtin_amdgcn_readfirstlane((int)hb[f]) * Ku) & mask;
#pragma unroll
                        for (int j = 0; j < NV; j++) {
                            float gv[VEC];
                            row_vload<VEC>(g2, row, cls[j], Ku, mask, true, act[j], gv);
#pragma unroll
                            for (int x = 0; x < VEC; x++) {
                                const float t = gv[x] + gsq[j][x];
                                ppu[j][x] += (t > 0.0f) ? rsq(t) : 0.0f;
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    return;
                }
                // contiguous rows: one vector load per lane and vector (it may reach up to VEC - 1 slots past its row: those slots
                // have err = 0 and never leave the lane)
                uint32_t f = 0;
                for (; f + U2 <= cnt; f += U2) {
                    float gv[U2][NV][VEC];
#pragma unroll
                    for (int u = 0; u < U2; u++) {
                        const uint32_t row = ((uint32_t)__builtin_amdgcn_readfirstlane((int)hb[f + u]) * Ku) & mask;
#pragma unroll
                        for (int j = 0; j < NV; j++) row_vload<VEC, AL>(g2, row, cls[j], Ku, mask, false, act[j], gv[u][j]);
                    }
#pragma unroll
                    for (int u = 0; u < U2; u++)
#pragma unroll
                        for (int j = 0; j < NV; j++)
#pragma unroll
                            for (int x = 0; x < VEC; x++) {
                                const float t = gv[u][j][x] + gsq[j][x];
                                ppu[j][x] += (t > 0.0f) ? rsq(t) : 0.0f;
                            }
                }
                for (; f < cnt; f++) {
                    const uint32_t row = ((uint32_t)__builtin_amdgcn_readfirstlane((int)hb[f]) * Ku) & mask;
                    float gv[NV][VEC];
#pragma unroll
                    for (int j = 0; j < NV; j++) row_vload<VEC, AL>(g2, row, cls[j], Ku, mask, false, act[j], gv[j]);
#pragma unroll
                    for (int j = 0; j < NV; j++)
#pragma unroll
                        for (int x = 0; x < VEC; x++) {
                            const float t = gv[j][x] + gsq[j][x];
                            ppu[j][x] += (t > 0.0f) ? rsq(t) : 0.0f;
                        }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            });
#pragma unroll
            for (int j = 0; j < NV; j++) {
                float u[VEC], q[VEC];
#pragma unroll
                for (int x = 0; x < VEC; x++) {
                    const float z = s * ppu[j][x];
                    float v;
                    if (err[j][x] == 0.0f || ppu[j][x] == 0.0f) v = 0.0f;
                    else if (z < 1e-6f) v = 2.0f * err[j][x] * s;
                    else v = err[j][x] * (1.0f - expf(-2.0f * z)) / ppu[j][x];
                    u[x] = v;
                    q[x] = gsq[j][x];
                }
                // which of the example's vectors carry a step: one bit per lane, one word per pass of phase B's row loop (same
                // lane <-> class mapping there), so that phase B fetches the step arrays of live vectors only
                {
                    bool live = reg_on && cls[j] < Ku;
#pragma unroll
                    for (int x = 0; x < VEC; x++) live = live || (u[x] != 0.0f) || (q[x] != 0.0f);
                    const unsigned long long m = __ballot(live);
                    const uint32_t it = c0 / (uint32_t)(WAVE * VEC) + (uint32_t)j;
                    if (lane == 0 && it < nit) amask[e * (uint64_t)nit + it] = m;
                }
                if (cls[j] + (uint32_t)VEC <= Ku) {
                    vstoreu<VEC>(upd_out + e * (uint64_t)Ku + cls[j], u);
                    vstoreu<VEC>(gsq_out + e * (uint64_t)Ku + cls[j], q);
                } else {                                          // the row's last, partial vector
#pragma unroll
                    for (int x = 0; x < VEC; x++)
                        if (cls[j] + (uint32_t)x < Ku) {
                            upd_out[e * (uint64_t)Ku + cls[j] + x] = u[x];
                            gsq_out[e * (uint64_t)Ku + cls[j] + x] = q[x];
                        }
                }
            }
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

// ------------------------------------------------------------------------------------------ partition starts
// pstart[p] = first sorted item of partition p (pstart[NP] = n): every thread looks at one boundary of the sorted keys.
__global__ void dtrain_starts_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t pshift, uint32_t NP, uint32_t *__restrict__ pstart)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const uint32_t lo = i == 0 ? 0u : (keys[i - 1] >> pshift) + 1u;       // partitions lo .. hi start at item i
    const uint32_t hi = i == n ? NP : (keys[i] >> pshift);
    for (uint32_t p = lo; p <= hi && p <= NP; p++) pstart[p] = i;
}

// ------------------------------------------------------------------------------------------ phase B
// One wave per partition of one colour (colour = partition & 1).  Its rows in sorted order, one after the other (the next
// item's step sizes are fetched while the current row is updated): G += q, w += u / sqrt(G).  Rows of two partitions of one
// colour never share a slot (a row spills at most into the NEXT partition), so the waves of a launch never meet.
template <int VEC, bool AL>
__global__ __launch_bounds__(BLOCK) void dtrain_apply_kernel(float *__restrict__ w, float *__restrict__ g2, uint32_t mask, int K,
                                                              const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                              const uint32_t *__restrict__ pstart, uint32_t NP, uint32_t colour,
                                                              const float *__restrict__ upd, const float *__restrict__ gsq,
                                                              const float *__restrict__ sfac, DArgs ta, unsigned long long *__restrict__ queue,
                                                              const unsigned long long *__restrict__ amask, uint32_t nit)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    const uint32_t Ku = (uint32_t)K;
    const bool reg = (ta.l1 != 0.0f) || (ta.l2 != 0.0f);
    const RowCross crosses(Ku, mask);
    const uint32_t nq = (NP - colour + 1u) / 2u;                 // partitions of this colour
    for (uint64_t q = first_chunk(wv, 1); q < nq; q = next_chunk_after(queue, lane, 1, nq)) {
        const uint32_t p = colour + 2u * (uint32_t)q;
        const uint32_t i0 = pstart[p], i1 = pstart[p + 1];
        for (uint32_t i = i0; i < i1; i++) {
            const uint32_t row = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[i]);
            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)vals[i]);
            if (e == 0xffffffffu) continue;                      // void key (fragment outside its buffer)
            const float s = sfac[e];
            const float decay = 1.0f - s * ta.l2, shrink = s * ta.l1;
            const float *ue = upd + (uint64_t)e * Ku, *qe = gsq + (uint64_t)e * Ku;
            const unsigned long long *me = amask + (uint64_t)e * nit;
            // one row: G += q, w += u / sqrt(G) for the classes with a step.  The last vector of an example's steps may reach into the
            // next example's (the arrays are padded): the excess is zeroed.  A row that passes the table's end (wave-uniform, rare)
            // is worked slot by slot; a contiguous one with one vector load and store per table, the row's last vector, when partial,
            // stored slot by slot.
            auto one_row = [&](auto crossing) {
                constexpr bool CROSS = decltype(crossing)::value;
                for (uint32_t it = 0; it < nit; it++) {
                    // classes already at their label: nothing to add (as VW skips zero updates).  Phase A left one bit per vector
                    // (step != 0 or accumulator term != 0, or regularisation on): the step arrays of the others are not even fetched
                    if (!((me[it] >> lane) & 1ull)) continue;
                    const uint32_t c = (uint32_t)lane * VEC + it * (uint32_t)(WAVE * VEC);
                    float u[VEC], qv[VEC];
                    vloadu<VEC>(ue + c, u);
                    vloadu<VEC>(qe + c, qv);
#pragma unroll
                    for (int x = 0; x < VEC; x++)
                        if (c + (uint32_t)x >= Ku) { u[x] = 0.0f; qv[x] = 0.0f; }
                    float gv[VEC], wv[VEC];
                    row_vload<VEC, AL>(g2, row, c, Ku, mask, CROSS, true, gv);
                    row_vload<VEC, AL>(w, row, c, Ku, mask, CROSS, true, wv);
#pragma unroll
                    for (int x = 0; x < VEC; x++) {
                        gv[x] += qv[x];
                        const float d = (gv[x] > 0.0f) ? u[x] * rsq(gv[x]) : 0.0f;
                        wv[x] = step_w(wv[x], d, reg, decay, shrink);
                    }
                    row_vstore<VEC, AL>(g2, row, c, Ku, mask, CROSS, gv);
                    row_vstore<VEC, AL>(w, row, c, Ku, mask, CROSS, wv);
                }
            };
            if (!AL && crosses(row)) one_row(std::true_type{});
            else one_row(std::false_type{});
            // the next row of this partition may be the same row, or overlap it: its loads must see these stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
}

// ------------------------------------------------------------------------------------------ phase C
// The constant row: one lane per class, the group's examples in order, accumulators in registers.  The chain over the
// examples is serial, so what counts is how fast one wave gets (upd, gsq) of the next example: straight from memory the
// loop waited 100 ns per example even with 32 examples' loads in flight (0.83 ms per group of 8192, a tenth of the
// group's time and a fifth in later passes).  Here three waves of the block stream tiles of CT examples x 64 classes
// into LDS, one tile ahead, and the fourth wave runs the chain out of LDS.  Same operations in the same order per class.
// (Loads two tiles ahead, the registers riding through the barrier, changed nothing: the chain wave sets the pace.)
constexpr int CT = 32;                 // examples per tile: 2 buffers x 2 arrays x 32 x 64 floats = 32 KiB of LDS
constexpr int CT_LOADERS = 192;        // threads of waves 1-3
template <bool REG>   // regularisation on (l1 or l2 non-zero): off, a step is one add and the chain half as long
__global__ __launch_bounds__(256) void dtrain_const_kernel(float *__restrict__ w, float *__restrict__ g2, uint32_t mask, int K, uint32_t crow,
                                                            const float *__restrict__ upd, const float *__restrict__ gsq,
                                                            const float *__restrict__ sfac, uint32_t gn, DArgs ta)
{
    __shared__ float s_u[2][CT][WAVE], s_q[2][CT][WAVE], s_s[2][CT];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const uint32_t c0 = blockIdx.x * WAVE, c = c0 + (uint32_t)lane;
    const uint32_t Ku = (uint32_t)K;
    constexpr bool reg = REG;
    const uint32_t ntile = (gn + CT - 1) / CT;
    auto load_tile = [&](uint32_t tile, int buf) {          // waves 1-3: all loads first, then the LDS stores
        constexpr int PER = (CT * WAVE + CT_LOADERS - 1) / CT_LOADERS;
        const uint32_t e0 = tile * CT, t0 = threadIdx.x - WAVE;
        float u[PER], q[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint32_t i = t0 + (uint32_t)k * CT_LOADERS, ex = i / WAVE, cl = i % WAVE;
            const bool on = i < CT * WAVE && e0 + ex < gn && c0 + cl < Ku;
            u[k] = on ? upd[(uint64_t)(e0 + ex) * Ku + c0 + cl] : 0.0f;
            q[k] = on ? gsq[(uint64_t)(e0 + ex) * Ku + c0 + cl] : 0.0f;
        }
        const float sv = (t0 < CT && e0 + t0 < gn) ? sfac[e0 + t0] : 0.0f;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint32_t i = t0 + (uint32_t)k * CT_LOADERS;
            if (i < CT * WAVE) { s_u[buf][i / WAVE][i % WAVE] = u[k]; s_q[buf][i / WAVE][i % WAVE] = q[k]; }
        }
        if (t0 < CT) s_s[buf][t0] = sv;
    };
    const size_t slot = (size_t)((crow + min(c, Ku - 1u)) & mask);
    float Gc = 0.0f, Wc = 0.0f;
    if (wv == 0) { Gc = g2[slot]; Wc = w[slot]; }
    else if (ntile) load_tile(0, 0);
    __syncthreads();
    for (uint32_t tile = 0; tile < ntile; tile++) {
        const int buf = (int)(tile & 1u);
        if (wv != 0) {
            if (tile + 1 < ntile) load_tile(tile + 1, buf ^ 1);
        } else {
            // branch-free, eight examples at a time out of registers: the only serial dependences are one add for G and an
            // add + select for w per example (with a branch per example every step paid the LDS and rsq latencies: 100 ns;
            // walking only the examples that carry a step for these 64 classes, from a per-tile bit mask, doubled the kernel's
            // time in round 3: the fixed loop lets the LDS reads of the next eight run ahead, a data-dependent one does not)
            const uint32_t cnt = min((uint32_t)CT, gn - tile * CT);
            for (uint32_t x0 = 0; x0 < cnt; x0 += 8) {
                float u[8], q[8], sx[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {               // rows past cnt (the group's last tile) were filled with zeros
                    const uint32_t x = min(x0 + (uint32_t)k, (uint32_t)CT - 1u);
                    u[k] = s_u[buf][x][lane]; q[k] = s_q[buf][x][lane]; sx[k] = s_s[buf][x];
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    // a step that is skipped (bitwise on purpose: short-circuit evaluation became a branch per example)
                    const bool off = (x0 + (uint32_t)k >= cnt) | (!reg & (q[k] == 0.0f) & (u[k] == 0.0f));
                    const float Gn = Gc + q[k];
                    const float d = (Gn > 0.0f) ? u[k] * rsq(Gn) : 0.0f;
                    const float Wn = step_w(Wc, d, reg, 1.0f - sx[k] * ta.l2, sx[k] * ta.l1);
                    Gc = Gn;                                 // a skipped step has q = 0: G + 0 is G
                    Wc = off ? Wc : Wn;
                }
            }
        }
        __syncthreads();
    }
    if (wv == 0 && c < Ku) {
        g2[slot] = Gc;
        w[slot] = Wc;
    }
}

__global__ void dtrain_mirror_kernel(float *w, uint64_t n_slots, uint32_t pad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pad) w[n_slots + i] = w[i % n_slots];
}

// strided gather of the group boundaries of the feature offsets: out[g] = foff[min(g * gs, n)]
__global__ void dtrain_bounds_kernel(const uint64_t *__restrict__ foff, uint64_t n, uint64_t gs, uint64_t ng, uint64_t *__restrict__ out)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g <= ng) out[g] = foff[g * gs < n ? g * gs : n];
}

struct SVariant {
    int vec, nv;
    bool al;
    const void *fn;
};
#define OPAL_SV(VEC, NV, U, AL) {VEC, NV, AL, (const void *)&dtrain_score_kernel<VEC, NV, U, AL>}
const SVariant kScoreVariants[] = {
    OPAL_SV(1, 1, 8, true),   // K = 1, 3
    OPAL_SV(2, 1, 8, true),   // K = 2
    OPAL_SV(4, 1, 8, true), OPAL_SV(4, 2, 8, true), OPAL_SV(4, 4, 4, true), OPAL_SV(4, 8, 2, true),         // K % 4 == 0
    OPAL_SV(4, 1, 8, false), OPAL_SV(4, 2, 8, false), OPAL_SV(4, 4, 4, false), OPAL_SV(4, 8, 2, false),     // any other K >= 5
};

struct DTrainState {
    uint64_t *foff = nullptr, *bounds = nullptr;
    uint64_t foff_cap = 0, bounds_cap = 0;
    float *upd = nullptr, *gsq = nullptr, *sfac = nullptr;
    uint64_t upd_cap = 0, gsq_cap = 0, sfac_cap = 0;
    uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *pstart = nullptr;
    uint64_t k_cap[2] = {0, 0}, v_cap[2] = {0, 0}, pstart_cap = 0;
    unsigned long long *amask = nullptr;      // per example of the group, one 64-bit word per 64 vectors of classes: which carry a step
    uint64_t amask_cap = 0;
    void *temp = nullptr;
    size_t temp_cap = 0;
};

template <class T>
int grow(opal_ctx *ctx, T **p, uint64_t *cap, uint64_t need)
{
    if (*cap >= need) return OPAL_OK;
    if (*p) { OPAL_HIP_CHECK(ctx, hipFree(*p)); *p = nullptr; *cap = 0; }
    const hipError_t e = hipMalloc((void **)p, need * sizeof(T));
    if (e != hipSuccess) return opal::fail(ctx, OPAL_ENOMEM, std::string("deterministic trainer: hipMalloc failed: ") + hipGetErrorString(e));
    *cap = need;
    return OPAL_OK;
}

}  // namespace

void opal::dtrain_free(opal_ctx *ctx)
{
    DTrainState *st = static_cast<DTrainState *>(ctx->dtrain);
    if (!st) return;
    void *all[] = {st->foff, st->bounds, st->upd, st->gsq, st->sfac, st->keys[0], st->keys[1], st->vals[0], st->vals[1], st->pstart, st->amask, st->temp};
    for (void *p : all)
        if (p) (void)hipFree(p);
    delete st;
    ctx->dtrain = nullptr;
}

// log2 of the partition size: at least K slots (a row may spill into the next partition only), at most 2^14 partitions
uint32_t opal::dtrain_pshift(int K, int bits)
{
    uint32_t ps = 0;
    while ((1u << ps) < (uint32_t)K) ps++;
    if (bits > 14 && ps < (uint32_t)bits - 14u) ps = (uint32_t)bits - 14u;
    return std::min<uint32_t>(ps, (uint32_t)bits);
}

int opal::train_batch_groups(opal_ctx *ctx, const uint8_t *d_reads, uint64_t reads_bytes, const uint64_t *d_frag_off,
                             const uint32_t *d_frag_len, const int32_t *d_labels, uint64_t n, int reverse, float eta, float l1, float l2,
                             int32_t *d_out_pred)
{
    if ((uint64_t)ctx->K * 2ull > ctx->n_slots)
        return opal::fail(ctx, OPAL_EINVAL, "opal_train_batch: the deterministic mode needs a table of at least 2 K slots");
    if (!ctx->dtrain) ctx->dtrain = new DTrainState;
    DTrainState *st = static_cast<DTrainState *>(ctx->dtrain);
    const uint64_t gs = ctx->train_group ? ctx->train_group : 8192;
    const uint64_t ng = (n + gs - 1) / gs;
    const int K = ctx->K;
    int rc = grow(ctx, &st->foff, &st->foff_cap, n + 1);
    if (rc) return rc;
    rc = grow(ctx, &st->bounds, &st->bounds_cap, ng + 1);
    if (rc) return rc;
    rc = opal_feature_offsets(ctx, d_frag_len, n, reverse, st->foff);
    if (rc) return rc;
    hipLaunchKernelGGL(dtrain_bounds_kernel, dim3((unsigned)((ng + 256) / 256)), dim3(256), 0, ctx->stream, st->foff, n, gs, ng, st->bounds);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    std::vector<uint64_t> bounds(ng + 1);
    OPAL_HIP_CHECK(ctx, hipMemcpyAsync(bounds.data(), st->bounds, (ng + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t max_keys = 1;
    for (uint64_t g = 0; g < ng; g++) max_keys = std::max(max_keys, bounds[g + 1] - bounds[g]);
    if (max_keys >= (1ull << 31)) return opal::fail(ctx, OPAL_EINVAL, "opal_train_batch: a group of examples has 2^31 features or more");
    const uint64_t gmax = std::min<uint64_t>(gs, n);
    rc = grow(ctx, &st->upd, &st->upd_cap, gmax * (uint64_t)K + 4);      // + 4: the last example's last vector may be partial
    if (rc) return rc;
    rc = grow(ctx, &st->gsq, &st->gsq_cap, gmax * (uint64_t)K + 4);
    if (rc) return rc;
    rc = grow(ctx, &st->sfac, &st->sfac_cap, gmax);
    if (rc) return rc;
    const int vec = K >= 4 ? 4 : (K % 2 == 0) ? 2 : 1;                     // 16-byte vectors for every K >= 4 (dword-aligned, see row_vload)
    const uint32_t nit = (uint32_t)((K + WAVE * vec - 1) / (WAVE * vec));  // passes of 64 vectors over a row
    rc = grow(ctx, &st->amask, &st->amask_cap, gmax * (uint64_t)nit);
    if (rc) return rc;
    for (int i = 0; i < 2; i++) {
        rc = grow(ctx, &st->keys[i], &st->k_cap[i], max_keys);
        if (rc) return rc;
        rc = grow(ctx, &st->vals[i], &st->v_cap[i], max_keys);
        if (rc) return rc;
    }
    const uint32_t pshift = opal::dtrain_pshift(K, ctx->bits);
    const uint32_t NP = (uint32_t)(ctx->n_slots >> pshift);
    rc = grow(ctx, &st->pstart, &st->pstart_cap, (uint64_t)NP + 2);
    if (rc) return rc;
    size_t temp_need = 0;
    OPAL_HIP_CHECK(ctx, rocprim::radix_sort_pairs(nullptr, temp_need, st->keys[0], st->keys[1], st->vals[0], st->vals[1], (unsigned int)max_keys,
                                                  pshift, (unsigned int)ctx->bits, ctx->stream));
    if (st->temp_cap < temp_need) {
        if (st->temp) { OPAL_HIP_CHECK(ctx, hipFree(st->temp)); st->temp = nullptr; st->temp_cap = 0; }
        if (hipMalloc(&st->temp, std::max<size_t>(temp_need, 256)) != hipSuccess)
            return opal::fail(ctx, OPAL_ENOMEM, "deterministic trainer: hipMalloc of the sort's workspace failed");
        st->temp_cap = std::max<size_t>(temp_need, 256);
    }

    Batch B{d_reads, reads_bytes, d_frag_off, d_frag_len, n};
    Geom G{ctx->d_pos, ctx->k, ctx->P, ctx->t, (ctx->t + 3) & ~3, reverse != 0, ctx->hash_all, ctx->p0_run};
    const uint32_t mask = ctx->bits == 32 ? 0xffffffffu : ((1u << ctx->bits) - 1u);
    const uint32_t const_idx = OPAL_VW_CONSTANT & mask;
    const uint32_t crow = (uint32_t)(((uint64_t)const_idx * (uint64_t)K) & mask);
    const int need = (int)nit;
    const int nv = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
    const void *score_fn = kScoreVariants[0].fn;
    const bool al = K % vec == 0;
    for (const SVariant &v : kScoreVariants)
        if (v.vec == vec && v.nv == nv && v.al == al) score_fn = v.fn;
    const void *apply_fn = vec == 4 ? (al ? (const void *)&dtrain_apply_kernel<4, true> : (const void *)&dtrain_apply_kernel<4, false>)
                         : vec == 2 ? (const void *)&dtrain_apply_kernel<2, true> : (const void *)&dtrain_apply_kernel<1, true>;
    int per_cu_s = 0, per_cu_a = 0;
    OPAL_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_s, score_fn, BLOCK, 0));
    OPAL_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_a, apply_fn, BLOCK, 0));
    per_cu_s = std::max(per_cu_s, 1);
    per_cu_a = std::max(per_cu_a, 1);
    DArgs ta{eta, l1, l2, ctx->min_label, ctx->max_label};
    float *w = ctx->d_w, *g2 = ctx->d_g2;
    float *upd = st->upd, *gsq = st->gsq, *sfac = st->sfac;
    const uint64_t *foff = st->foff;
    unsigned long long *queue = ctx->d_queue;
    unsigned long long *amask = st->amask;
    const unsigned long long *camask = st->amask;

    for (uint64_t g = 0; g < ng; g++) {
        uint64_t e0 = g * gs;
        uint32_t gn = (uint32_t)std::min<uint64_t>(gs, n - e0);
        const uint32_t nk = (uint32_t)(bounds[g + 1] - bounds[g]);
        uint32_t *keys_in = st->keys[0], *vals_in = st->vals[0], *keys_s = st->keys[1], *vals_s = st->vals[1];
        // A: scores, step sizes, row keys (tables read-only)
        OPAL_HIP_CHECK(ctx, hipMemsetAsync(queue, 0, sizeof(unsigned long long), ctx->stream));
        {
            const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((gn + WPB - 1) / WPB, (uint64_t)ctx->num_cus * per_cu_s));
            void *args[] = {&B, &G, &w, &g2, (void *)&mask, (void *)&K, (void *)&const_idx, &d_labels, &ta, &e0, &gn, &foff, &upd, &gsq, &sfac,
                            &keys_in, &vals_in, &d_out_pred, &queue, &amask, (void *)&nit};
            OPAL_HIP_CHECK(ctx, hipLaunchKernel(score_fn, dim3(grid), dim3(BLOCK), args, 0, ctx->stream));
        }
        if (nk) {
            // S: stable sort of the group's rows by partition; where every partition starts
            size_t tb = st->temp_cap;
            OPAL_HIP_CHECK(ctx, rocprim::radix_sort_pairs(st->temp, tb, keys_in, keys_s, vals_in, vals_s, nk, pshift, (unsigned int)ctx->bits, ctx->stream));
            hipLaunchKernelGGL(dtrain_starts_kernel, dim3((nk + 256) / 256), dim3(256), 0, ctx->stream, keys_s, nk, pshift, NP, st->pstart);
            OPAL_HIP_CHECK(ctx, hipGetLastError());
            // B: even partitions, then odd ones
            for (uint32_t colour = 0; colour < 2 && colour < NP; colour++) {
                const uint32_t nq = (NP - colour + 1u) / 2u;
                OPAL_HIP_CHECK(ctx, hipMemsetAsync(queue, 0, sizeof(unsigned long long), ctx->stream));
                const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((nq + WPB - 1) / WPB, (uint64_t)ctx->num_cus * per_cu_a));
                const uint32_t *ks = keys_s, *vs = vals_s, *ps = st->pstart;
                const float *cu = upd, *cq = gsq, *cs = sfac;
                void *args[] = {&w, &g2, (void *)&mask, (void *)&K, &ks, &vs, &ps, (void *)&NP, &colour, &cu, &cq, &cs, &ta, &queue, &camask, (void *)&nit};
                OPAL_HIP_CHECK(ctx, hipLaunchKernel(apply_fn, dim3(grid), dim3(BLOCK), args, 0, ctx->stream));
            }
        }
        // C: the constant row, examples in order
        if (l1 != 0.0f || l2 != 0.0f)
            hipLaunchKernelGGL(dtrain_const_kernel<true>, dim3((K + 63) / 64), dim3(256), 0, ctx->stream, w, g2, mask, K, crow, (const float *)upd,
                               (const float *)gsq, (const float *)sfac, gn, ta);
        else
            hipLaunchKernelGGL(dtrain_const_kernel<false>, dim3((K + 63) / 64), dim3(256), 0, ctx->stream, w, g2, mask, K, crow, (const float *)upd,
                               (const float *)gsq, (const float *)sfac, gn, ta);
        OPAL_HIP_CHECK(ctx, hipGetLastError());
    }
    const uint32_t pad = (uint32_t)K + 4;
    hipLaunchKernelGGL(dtrain_mirror_kernel, dim3((pad + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_w, ctx->n_slots, pad);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    return OPAL_OK;
}
