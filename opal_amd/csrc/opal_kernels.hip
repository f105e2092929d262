// Device-side half of libopal_hip.so: hand-written HIP kernels for gfx950 (MI355X, wave64) and the C-ABI
// entry points that launch them.  See DESIGN.md for the data layout and the roofline of each kernel.
//
// Work decomposition (all kernels): one 64-lane wave owns one fragment at a time; waves stride over the batch.
//   phase H (hash):   the wave stages the fragment's bytes in LDS (2-bit codes expanded to the ASCII the
//                     reference hashes; the reverse-complement strand is produced on the fly), lanes walk
//                     the sliding windows, gather the t pattern positions and run MurmurHash3_x86_32 over
//                     "<t chars><pattern index>" in registers.  Hashes land in an LDS buffer in the
//                     reference's feature order (window-major, pattern-minor, forward strand then RC).
//   phase S (score):  for every hash in order the wave reads the K-float row (h*K mod 2^b) of the weight
//                     table straight into registers, lanes own classes, and adds it to its accumulators.
//                     fp32 adds happen in feature order per class, exactly like VW's sequential sum.
//                     U rows are kept in flight per wave (software pipeline, no conditionals in the loop).
//   emit mode replaces phase S by a coalesced copy of the LDS hash buffer to HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "opal_device.h"
#include "opal_internal.h"

namespace {

// F_j of opal_feature_offsets for one fragment (length word with the raw flag)
__device__ __forceinline__ uint64_t feature_count(uint32_t len_word, const Geom &G)
{
    const uint32_t len = len_word & ~OPAL_RAW_FLAG;
    return len >= (uint32_t)G.k ? (uint64_t)(len - (uint32_t)G.k + 1u) * (uint64_t)G.P * (G.reverse ? 2u : 1u) : 0ull;
}

// ------------------------------------------------------------------------------------------ emit kernels
// the emit kernel only copies the hash buffer out: a smaller one than the scoring kernels' (the 548 features of a 200-bp
// fragment still fit) halves its LDS per block and lifts the LDS limit on resident waves
constexpr int HB_EMIT = 640;
constexpr uint64_t EMIT_QCHUNK = 32;   // fragments per draw from the work queue: a fragment takes ~10 us here (100x less than in predict) and the
                                       // counter is one address: 8 per draw = 83 M atomics/s held the kernel at 6.14 ms, 16-32 give 5.26, 256 5.65 (tail)
__global__ __launch_bounds__(BLOCK) void emit_idx_kernel(Batch B, Geom G, uint32_t mask, const uint64_t *__restrict__ out_off,
                                                          uint32_t *__restrict__ out_idx, uint64_t out_cap,
                                                          unsigned long long *__restrict__ queue, uint64_t qchunk)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_seq[WPB][2 * SEQB];
    __shared__ __attribute__((aligned(16))) uint32_t s_hb[WPB][HB_EMIT];
    __shared__ uint32_t s_tab[KTAB];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    fill_kmer4_table(s_tab);
    for (uint64_t q0 = next_chunk(queue, lane, qchunk); q0 < B.n; q0 = next_chunk(queue, lane, qchunk))
    for (uint64_t frag = q0; frag < min(q0 + qchunk, B.n); frag++) {
        const uint64_t o0 = out_off[frag];
        if (o0 > out_cap || feature_count(B.len[frag], G) > out_cap - o0) continue;   // never write outside out_idx
        uint32_t *dst = out_idx + o0;
        uint32_t *hb = s_hb[wv];
        walk_fragment<HB_EMIT>(B, G, frag, mask, false, 0u, s_seq[wv], hb, s_tab, lane, [&](uint32_t cnt) {
            if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {   // wave-uniform: 16-byte stores, then the tail
                const uint32_t n4 = cnt / 4u;
                for (uint32_t i = lane; i < n4; i += WAVE)
                    reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(hb)[i];
                for (uint32_t i = n4 * 4u + lane; i < cnt; i += WAVE) dst[i] = hb[i];
            } else {
                for (uint32_t i = lane; i < cnt; i += WAVE) dst[i] = hb[i];
            }
            dst += cnt;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        });
    }
}

// Debug: the t gathered characters of every feature (reference order), t bytes per feature.
__global__ __launch_bounds__(BLOCK) void emit_names_kernel(Batch B, Geom G, const uint64_t *__restrict__ out_off,
                                                            uint8_t *__restrict__ out_names, uint64_t out_cap)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_seq[WPB][SEQB];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    uint8_t *sq = s_seq[wv];
    const uint64_t nwaves = (uint64_t)gridDim.x * WPB;
    const int k = G.k, P = G.P, t = G.t;
    for (uint64_t frag = (uint64_t)blockIdx.x * WPB + wv; frag < B.n; frag += nwaves) {
        const FragRef fr = locate_fragment(B, frag);
        const bool raw = fr.raw;
        const uint32_t len = fr.len;
        if (len < (uint32_t)k) continue;
        const uint8_t *src = fr.src;
        const uint32_t W = len - (uint32_t)k + 1u;
        const uint32_t tw_seq = (uint32_t)(SEQB - k + 1);
        uint64_t f0 = out_off[frag];
        if (f0 > out_cap || feature_count(B.len[frag], G) > out_cap - f0) continue;   // never write outside out_names
        for (int s = 0; s < (G.reverse ? 2 : 1); s++) {
            for (uint32_t w0 = 0; w0 < W; w0 += tw_seq) {
                const uint32_t tw = min(W - w0, tw_seq);
                load_tile(sq, src, len, raw, fr.shift, s, w0, tw + (uint32_t)k - 1u, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (uint32_t i = lane; i < tw * (uint32_t)P; i += WAVE) {
                    const uint32_t wl = i / (uint32_t)P, p = i % (uint32_t)P;
                    uint8_t *o = out_names + (f0 + i) * (uint64_t)t;
                    for (int j = 0; j < t; j++) o[j] = sq[wl + G.pos[p * (uint32_t)G.ts + j]];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                f0 += (uint64_t)tw * P;
            }
        }
    }
}

template <int VEC, int NV, int U, bool NT>
__global__ __launch_bounds__(BLOCK) void predict_kernel(Batch B, Geom G, const float *__restrict__ tab, uint32_t mask, int K,
                                                         int add_constant, uint32_t const_idx, int clip, float min_label,
                                                         float max_label, int32_t *__restrict__ out_class,
                                                         float *__restrict__ out_scores, unsigned long long *__restrict__ queue)
{
    using V = typename VecOf<VEC>::type;
    constexpr int CH = WAVE * VEC * NV;  // classes per chunk
    __shared__ __attribute__((aligned(16))) uint8_t s_seq[WPB][2 * SEQB];
    __shared__ __attribute__((aligned(16))) uint32_t s_hb[WPB][HB + HB_PAD];
    __shared__ uint32_t s_tab[KTAB];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    fill_kmer4_table(s_tab);
    uint32_t *hb = s_hb[wv];
    const uint32_t Ku = (uint32_t)K;
    for (uint64_t q0 = next_chunk(queue, lane); q0 < B.n; q0 = next_chunk(queue, lane))
    for (uint64_t frag = q0; frag < min(q0 + QCHUNK, B.n); frag++) {
        float best = 0.0f;
        uint32_t best_c = 0xffffffffu;
        for (uint32_t c0 = 0; c0 < Ku; c0 += CH) {
            uint32_t loff[NV];
            uint32_t cls[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) {
                cls[j] = c0 + (uint32_t)(j * WAVE + lane) * VEC;
                loff[j] = min(cls[j], Ku - VEC);  // lanes past the row's end re-read its last vector (same cache line)
            }
            V acc[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) acc[j] = V{};
            walk_fragment(B, G, frag, mask, add_constant != 0, const_idx, s_seq[wv], hb, s_tab, lane, [&](uint32_t cnt) {
                // dummy rows for the prefetch overrun: the constant row (always cache-hot)
                if (lane < HB_PAD) hb[cnt + lane] = const_idx;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                stream_rows<VEC, NV, U, false, NT>(tab, hb, cnt, Ku, mask, loff, acc);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            });
            // finalize this chunk: clip (gd.cc finalize_prediction), optional score dump, running first-max
#pragma unroll
            for (int j = 0; j < NV; j++) {
                if (cls[j] < Ku) {
                    float sc[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; e++) {
                        float s = vget<VEC>(acc[j], e);
                        if (clip) {
                            if (s != s) s = 0.0f;
                            else if (s > max_label) s = max_label;
                            else if (s < min_label) s = min_label;
                        }
                        sc[e] = s;
                        if (best_c == 0xffffffffu || s > best) { best = s; best_c = cls[j] + e; }
                    }
                    if (out_scores) {
                        float *o = out_scores + frag * (uint64_t)Ku + cls[j];
#pragma unroll
                        for (int e = 0; e < VEC; e++) o[e] = sc[e];
                    }
                }
            }
        }
        // cross-lane first-max: larger score wins, ties go to the lower class (oaa.cc: strict '>')
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const float ob = __shfl_xor(best, d, WAVE);
            const uint32_t oc = (uint32_t)__shfl_xor((int)best_c, d, WAVE);
            const bool take = (oc != 0xffffffffu) && (best_c == 0xffffffffu || ob > best || (ob == best && oc < best_c));
            if (take) { best = ob; best_c = oc; }
        }
        if (lane == 0) out_class[frag] = (int32_t)(best_c + 1u);
    }
}

// ------------------------------------------------------------------------------------------ offsets scan
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;

__device__ __forceinline__ uint64_t frag_features(uint32_t lw, int k, int P, int strands)
{
    const uint32_t len = lw & ~OPAL_RAW_FLAG;
    return len >= (uint32_t)k ? (uint64_t)(len - (uint32_t)k + 1u) * (uint64_t)(P * strands) : 0ull;
}

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint64_t o = __shfl_up(v, d, WAVE);
        if (lane >= d) v += o;
    }
    return v;
}

// pass 1: per-tile totals; pass 3: exclusive scan inside each tile + tile base.
template <bool WRITE>
__global__ __launch_bounds__(BLOCK) void scan_tiles_kernel(const uint32_t *__restrict__ frag_len, uint64_t n, int k, int P,
                                                            int strands, uint64_t *__restrict__ tile_sums,
                                                            uint64_t *__restrict__ out_off)
{
    __shared__ uint64_t s_w[WPB];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t tsum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = (base + i < n) ? frag_features(frag_len[base + i], k, P, strands) : 0ull;
        tsum += v[i];
    }
    const uint64_t incl = wave_incl_scan(tsum, lane);
    if (lane == WAVE - 1) s_w[wv] = incl;
    __syncthreads();
    uint64_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < WPB; w++) {
        if (w < wv) wbase += s_w[w];
        total += s_w[w];
    }
    if (!WRITE) {
        if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
    } else {
        uint64_t run = tile_sums[blockIdx.x] + wbase + (incl - tsum);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            if (base + i < n) out_off[base + i] = run;
            run += v[i];
        }
    }
}

// pass 2: one block turns tile totals into exclusive tile bases; writes the grand total to out_off[n].
__global__ __launch_bounds__(BLOCK) void scan_sums_kernel(uint64_t *__restrict__ tile_sums, uint64_t ntiles, uint64_t n,
                                                           uint64_t *__restrict__ out_off)
{
    __shared__ uint64_t s_w[WPB];
    __shared__ uint64_t s_carry;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < ntiles; b0 += BLOCK) {
        const uint64_t i = b0 + threadIdx.x;
        const uint64_t v = i < ntiles ? tile_sums[i] : 0ull;
        const uint64_t incl = wave_incl_scan(v, lane);
        if (lane == WAVE - 1) s_w[wv] = incl;
        __syncthreads();
        uint64_t wbase = s_carry;
        for (int w = 0; w < wv; w++) wbase += s_w[w];
        if (i < ntiles) tile_sums[i] = wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == BLOCK - 1) s_carry = wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) out_off[n] = s_carry;
}

// Scratch of opal_feature_offsets_noalloc: the per-tile totals of the offsets scan live in a file-scope device array
// instead of a hipMalloc'ed buffer, for callers that must not allocate (HIP graph capture, serving loops).  Every GPU
// the library runs on holds its own copy (512 KiB) until the process ends.  In host code the NAME `scratch` is only a
// stand-in for the symbol: the device address comes from hipGetSymbolAddress, once per device (static_scratch_for),
// and only that pointer is ever handed to kernels.
constexpr uint64_t NOALLOC_MAX_TILES = 65536;                                  // x 8 B = 512 KiB
constexpr uint64_t NOALLOC_MAX_FRAGMENTS = NOALLOC_MAX_TILES * SCAN_TILE;      // 134 217 728 fragments per call
}  // namespace
static __device__ __align__(256) unsigned long long scratch[NOALLOC_MAX_TILES];   // file scope
namespace {

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
unsigned long long *g_scratch_ptr[MAX_DEVICES] = {};   // per device: that GPU's copy of `scratch`

// the device must be current; called from opal_ctx_create (the library's set-up call), once per device
int static_scratch_for(int device, unsigned long long **out, std::string *err)
{
    if (device < 0 || device >= MAX_DEVICES) { *err = "device id beyond the static scratch table"; return OPAL_EINVAL; }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        unsigned long long *p = nullptr;
        const hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) { *err = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e); return OPAL_EHIP; }
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return OPAL_OK;
}

// ------------------------------------------------------------------------------------------ table padding
__global__ void mirror_head_kernel(float *w, uint64_t n_slots, uint32_t pad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pad) w[n_slots + i] = w[i % n_slots];
}

// ------------------------------------------------------------------------------------------ fragment views
// drawfrag's `seq[pos:pos+l]` (util/drawfrag.py:38-39) without moving a base: fragment i is (offset, length) into the
// batch that holds the genomes; a draw that does not fit its genome becomes an empty fragment.
__global__ __launch_bounds__(BLOCK) void fragment_views_kernel(const uint64_t *__restrict__ g_off, const uint32_t *__restrict__ g_len,
                                                                uint64_t n_genomes, const uint64_t *__restrict__ pos,
                                                                const uint32_t *__restrict__ gen, uint64_t n, uint32_t frag_len,
                                                                uint64_t *__restrict__ frag_off, uint32_t *__restrict__ out_len)
{
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t g = gen[i];
        const uint64_t p = pos[i];
        uint64_t o = 0;
        uint32_t lw = 0;
        if (g < n_genomes) {
            const uint32_t glw = g_len[g];
            const uint64_t gl = glw & ~OPAL_RAW_FLAG;
            if (p <= gl && frag_len <= gl - p) {
                o = g_off[g] + p;                               // bases for packed genomes, bytes for raw ones
                lw = (glw & OPAL_RAW_FLAG) | frag_len;
            }
        }
        frag_off[i] = o;
        out_len[i] = lw;
    }
}

// ------------------------------------------------------------------------------------------ launch helpers
struct Variant {
    int vec, nv, u;
    const void *fn;      // default cache policy: tables that fit the caches
    const void *fn_nt;   // non-temporal row loads: tables far larger than the caches
    const char *name;
};

#define OPAL_V(VEC, NV, U) {VEC, NV, U, (const void *)&predict_kernel<VEC, NV, U, false>, (const void *)&predict_kernel<VEC, NV, U, true>, \
                           "predict_kernel<VEC=" #VEC ",NV=" #NV ",U=" #U ">"}
const Variant kVariants[] = {
    OPAL_V(1, 1, 8), OPAL_V(1, 2, 8), OPAL_V(1, 3, 4), OPAL_V(1, 4, 4), OPAL_V(1, 5, 2), OPAL_V(1, 6, 2), OPAL_V(1, 7, 2), OPAL_V(1, 8, 2),
    OPAL_V(2, 1, 8), OPAL_V(2, 2, 8), OPAL_V(2, 3, 4), OPAL_V(2, 4, 4), OPAL_V(2, 5, 2), OPAL_V(2, 6, 2), OPAL_V(2, 7, 2), OPAL_V(2, 8, 2),
    OPAL_V(4, 1, 8), OPAL_V(4, 2, 8), OPAL_V(4, 3, 4), OPAL_V(4, 4, 4), OPAL_V(4, 5, 2), OPAL_V(4, 6, 2), OPAL_V(4, 7, 2), OPAL_V(4, 8, 2),
};

constexpr uint64_t NT_TABLE_BYTES = 4ull << 30;   // tables from this size on are streamed with non-temporal loads

const Variant &pick_variant(int K)
{
    const int vec = (K % 4 == 0) ? 4 : (K % 2 == 0) ? 2 : 1;
    int nv = (K + WAVE * vec - 1) / (WAVE * vec);
    if (nv > 8) nv = 8;  // larger K: class chunks of 64*vec*8, the fragment is re-hashed per chunk
    for (const Variant &v : kVariants)
        if (v.vec == vec && v.nv == nv) return v;
    return kVariants[0];
}

int grid_for(opal_ctx *ctx, const void *fn, uint64_t n, int *out_grid)
{
    int per_cu = 0;
    OPAL_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, BLOCK, 0));
    if (per_cu < 1) per_cu = 1;
    const uint64_t want = (n + WPB - 1) / WPB;
    const uint64_t cap = (uint64_t)ctx->num_cus * (uint64_t)per_cu;
    *out_grid = (int)std::max<uint64_t>(1, std::min(want, cap));
    return OPAL_OK;
}

int check_batch(opal_ctx *ctx, const char *who, const void *reads, const void *off, const void *len, uint64_t n)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, std::string(who) + ": null ctx");
    if (ctx->P == 0) return opal::fail(ctx, OPAL_ESTATE, std::string(who) + ": opal_set_patterns has not been called");
    if (n && (!reads || !off || !len)) return opal::fail(ctx, OPAL_EINVAL, std::string(who) + ": null batch pointer");
    if (n > (1ull << 40)) return opal::fail(ctx, OPAL_EINVAL, std::string(who) + ": n too large");
    return OPAL_OK;
}

}  // namespace

// ============================================================================================== C-ABI
extern "C" int opal_ctx_create(int device_id, void *stream, opal_ctx **out)
{
    if (!out) return opal::fail(nullptr, OPAL_EINVAL, "opal_ctx_create: null out");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return opal::fail(nullptr, OPAL_EHIP, std::string("opal_ctx_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (device_id < 0 || device_id >= ndev) return opal::fail(nullptr, OPAL_EINVAL, "opal_ctx_create: bad device id");
    opal_ctx *ctx = new (std::nothrow) opal_ctx;
    if (!ctx) return opal::fail(nullptr, OPAL_ENOMEM, "opal_ctx_create: out of memory");
    ctx->device = device_id;
    e = hipSetDevice(device_id);
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device_id);
        if (e == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    }
    // NULL is HIP's default (null) stream, which is also what torch.cuda.current_stream() is unless the caller
    // switched streams: work enqueued here is then ordered with the caller's copies without extra events.
    ctx->stream = (hipStream_t)stream;
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_queue, 256);
    if (e != hipSuccess) {
        std::string m = std::string("opal_ctx_create: ") + hipGetErrorString(e);
        delete ctx;
        return opal::fail(nullptr, OPAL_EHIP, m);
    }
    {   // this device's copy of the static scratch (opal_feature_offsets_noalloc); device_id is current here
        std::string m;
        const int rc = static_scratch_for(device_id, &ctx->d_static_sums, &m);
        if (rc) {
            (void)hipFree(ctx->d_queue);
            delete ctx;
            return opal::fail(nullptr, rc, "opal_ctx_create: " + m);
        }
    }
    *out = ctx;
    return OPAL_OK;
}

extern "C" int opal_ctx_destroy(opal_ctx *ctx)
{
    if (!ctx) return OPAL_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_pos) (void)hipFree(ctx->d_pos);
    if (ctx->d_w) (void)hipFree(ctx->d_w);
    if (ctx->d_g2) (void)hipFree(ctx->d_g2);
    if (ctx->ing_tiles) (void)hipFree(ctx->ing_tiles);
    if (ctx->ing_rec) (void)hipFree(ctx->ing_rec);
    if (ctx->d_block_sums) (void)hipFree(ctx->d_block_sums);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    delete ctx;
    return OPAL_OK;
}

extern "C" int opal_sync(opal_ctx *ctx)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_sync: null ctx");
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return OPAL_OK;
}

extern "C" void *opal_ctx_stream(opal_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int opal_set_patterns(opal_ctx *ctx, const int32_t *pos, int P, int t, int k)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_set_patterns: null ctx");
    if (!pos || P < 1 || P > OPAL_MAX_P || t < 1 || t > OPAL_MAX_T || k < 1 || k > OPAL_MAX_K)
        return opal::fail(ctx, OPAL_EINVAL, "opal_set_patterns: need 1<=P<=64, 1<=t<=64, 1<=k<=256");
    // the reference only asserts k % t == 0 when a pattern file exists (fasta2skm.py:125); the identity
    // pattern used without a file has t == k, which satisfies it too.
    if (k % t != 0) return opal::fail(ctx, OPAL_EINVAL, "opal_set_patterns: k must be a multiple of t");
    const int ts = (t + 3) & ~3;   // device rows are padded to whole dwords (Geom::ts)
    std::vector<uint8_t> h((size_t)P * ts, 0);
    for (int i = 0; i < P * t; i++) {
        if (pos[i] < 0 || pos[i] >= k) return opal::fail(ctx, OPAL_EINVAL, "opal_set_patterns: position outside [0,k)");
        h[(size_t)(i / t) * ts + (i % t)] = (uint8_t)pos[i];
    }
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_pos) { OPAL_HIP_CHECK(ctx, hipFree(ctx->d_pos)); ctx->d_pos = nullptr; }
    OPAL_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_pos, h.size()));
    OPAL_HIP_CHECK(ctx, hipMemcpy(ctx->d_pos, h.data(), h.size(), hipMemcpyHostToDevice));
    ctx->P = P; ctx->t = t; ctx->k = k;
    ctx->p0_run = pos[0];   // pattern 0 as a run of consecutive positions (Opal's first row: ldpc.py:83-85), else -1
    for (int j = 1; j < t; j++)
        if (pos[j] != pos[0] + j) ctx->p0_run = -1;
    return OPAL_OK;
}

static int alloc_table(opal_ctx *ctx, uint64_t n_slots, int K, int bits)
{
    if (bits < 1 || bits > 32 || n_slots != (1ull << bits))
        return opal::fail(ctx, OPAL_EINVAL, "weights: need 1 <= bits <= 32 and n_slots == 2^bits");
    if (K < 1 || K > (1 << 20)) return opal::fail(ctx, OPAL_EINVAL, "weights: need 1 <= K <= 2^20");
    const uint64_t pad = (uint64_t)K + 4;
    const uint64_t need = n_slots + pad;
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->w_alloc_floats < need) {
        if (ctx->d_w) { OPAL_HIP_CHECK(ctx, hipFree(ctx->d_w)); ctx->d_w = nullptr; ctx->w_alloc_floats = 0; }
        hipError_t e = hipMalloc((void **)&ctx->d_w, need * sizeof(float));
        if (e != hipSuccess) return opal::fail(ctx, OPAL_ENOMEM, std::string("weights: hipMalloc failed: ") + hipGetErrorString(e));
        ctx->w_alloc_floats = need;
    }
    ctx->n_slots = n_slots; ctx->K = K; ctx->bits = bits; ctx->g2_valid = false;
    return OPAL_OK;
}

static int finish_table(opal_ctx *ctx)
{
    const uint32_t pad = (uint32_t)ctx->K + 4;
    hipLaunchKernelGGL(mirror_head_kernel, dim3((pad + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_w, ctx->n_slots, pad);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return OPAL_OK;
}

extern "C" int opal_load_weights(opal_ctx *ctx, const float *w, uint64_t n_slots, int K, int bits)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_load_weights: null ctx");
    if (!w) return opal::fail(ctx, OPAL_EINVAL, "opal_load_weights: null table");
    int rc = alloc_table(ctx, n_slots, K, bits);
    if (rc) return rc;
    OPAL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_w, w, n_slots * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return finish_table(ctx);
}

extern "C" int opal_set_weights_device(opal_ctx *ctx, const float *d_w, uint64_t n_slots, int K, int bits)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_set_weights_device: null ctx");
    if (!d_w) return opal::fail(ctx, OPAL_EINVAL, "opal_set_weights_device: null table");
    int rc = alloc_table(ctx, n_slots, K, bits);
    if (rc) return rc;
    OPAL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_w, d_w, n_slots * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    return finish_table(ctx);
}

extern "C" int opal_set_vw_options(opal_ctx *ctx, int hash_all, int add_constant, int clip, float min_label, float max_label)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_set_vw_options: null ctx");
    if (clip && !(min_label <= max_label)) return opal::fail(ctx, OPAL_EINVAL, "opal_set_vw_options: min_label > max_label");
    ctx->hash_all = hash_all != 0; ctx->add_constant = add_constant != 0; ctx->clip = clip != 0;
    ctx->min_label = min_label; ctx->max_label = max_label;
    return OPAL_OK;
}

extern "C" int opal_feature_offsets(opal_ctx *ctx, const uint32_t *d_frag_len, uint64_t n, int reverse, uint64_t *d_out_off)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_feature_offsets: null ctx");
    if (ctx->P == 0) return opal::fail(ctx, OPAL_ESTATE, "opal_feature_offsets: opal_set_patterns has not been called");
    if (!d_out_off || (n && !d_frag_len)) return opal::fail(ctx, OPAL_EINVAL, "opal_feature_offsets: null pointer");
    if (n > (1ull << 40)) return opal::fail(ctx, OPAL_EINVAL, "opal_feature_offsets: n too large");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (ntiles > ctx->block_sums_cap) {
        OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_block_sums) { OPAL_HIP_CHECK(ctx, hipFree(ctx->d_block_sums)); ctx->d_block_sums = nullptr; }
        const uint64_t cap = std::max<uint64_t>(ntiles, 1024);
        OPAL_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_block_sums, cap * sizeof(uint64_t)));
        ctx->block_sums_cap = cap;
    }
    const int strands = reverse ? 2 : 1;
    if (ntiles) {
        hipLaunchKernelGGL(scan_tiles_kernel<false>, dim3((unsigned)ntiles), dim3(BLOCK), 0, ctx->stream, d_frag_len, n, ctx->k,
                           ctx->P, strands, ctx->d_block_sums, d_out_off);
        OPAL_HIP_CHECK(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(BLOCK), 0, ctx->stream, ctx->d_block_sums, ntiles, n, d_out_off);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    if (ntiles) {
        hipLaunchKernelGGL(scan_tiles_kernel<true>, dim3((unsigned)ntiles), dim3(BLOCK), 0, ctx->stream, d_frag_len, n, ctx->k,
                           ctx->P, strands, ctx->d_block_sums, d_out_off);
        OPAL_HIP_CHECK(ctx, hipGetLastError());
    }
    return OPAL_OK;
}

// Same three launches as opal_feature_offsets, with the tile totals in this GPU's static scratch: no hipMalloc, no
// hipFree, no synchronisation.  Pass 1 overwrites every total it later reads, so nothing has to start at zero.
extern "C" int opal_feature_offsets_noalloc(opal_ctx *ctx, const uint32_t *d_frag_len, uint64_t n, int reverse, uint64_t *d_out_off)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_feature_offsets_noalloc: null ctx");
    if (ctx->P == 0) return opal::fail(ctx, OPAL_ESTATE, "opal_feature_offsets_noalloc: opal_set_patterns has not been called");
    if (!d_out_off || (n && !d_frag_len)) return opal::fail(ctx, OPAL_EINVAL, "opal_feature_offsets_noalloc: null pointer");
    if (n > NOALLOC_MAX_FRAGMENTS)
        return opal::fail(ctx, OPAL_ERANGE, "opal_feature_offsets_noalloc: n exceeds the static scratch (at most 134217728 fragments per call); "
                                            "split the batch or use opal_feature_offsets");
    if (!ctx->d_static_sums) return opal::fail(ctx, OPAL_ESTATE, "opal_feature_offsets_noalloc: static scratch not resolved");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    unsigned long long *sums = ctx->d_static_sums;     // the device address, never the symbol's name
    const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    const int strands = reverse ? 2 : 1;
    if (ntiles) {
        hipLaunchKernelGGL(scan_tiles_kernel<false>, dim3((unsigned)ntiles), dim3(BLOCK), 0, ctx->stream, d_frag_len, n, ctx->k,
                           ctx->P, strands, (uint64_t *)sums, d_out_off);
        OPAL_HIP_CHECK(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(BLOCK), 0, ctx->stream, (uint64_t *)sums, ntiles, n, d_out_off);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    if (ntiles) {
        hipLaunchKernelGGL(scan_tiles_kernel<true>, dim3((unsigned)ntiles), dim3(BLOCK), 0, ctx->stream, d_frag_len, n, ctx->k,
                           ctx->P, strands, (uint64_t *)sums, d_out_off);
        OPAL_HIP_CHECK(ctx, hipGetLastError());
    }
    return OPAL_OK;
}

extern "C" int opal_fragment_views(opal_ctx *ctx, const uint64_t *d_genome_off, const uint32_t *d_genome_len, uint64_t n_genomes,
                                   const uint64_t *d_pos, const uint32_t *d_genome_idx, uint64_t n, uint32_t frag_len,
                                   uint64_t *d_frag_off, uint32_t *d_frag_len)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_fragment_views: null ctx");
    if (frag_len >= OPAL_RAW_FLAG) return opal::fail(ctx, OPAL_EINVAL, "opal_fragment_views: need frag_len < 2^31");
    if (n == 0) return OPAL_OK;
    if (!d_genome_off || !d_genome_len || !d_pos || !d_genome_idx || !d_frag_off || !d_frag_len)
        return opal::fail(ctx, OPAL_EINVAL, "opal_fragment_views: null pointer");
    if (n > (1ull << 40)) return opal::fail(ctx, OPAL_EINVAL, "opal_fragment_views: n too large");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((n + BLOCK - 1) / BLOCK, (uint64_t)ctx->num_cus * 16));
    hipLaunchKernelGGL(fragment_views_kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_genome_off, d_genome_len, n_genomes, d_pos,
                       d_genome_idx, n, frag_len, d_frag_off, d_frag_len);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    return OPAL_OK;
}

extern "C" int opal_features(opal_ctx *ctx, const uint8_t *d_reads, uint64_t reads_bytes, const uint64_t *d_frag_off,
                             const uint32_t *d_frag_len, uint64_t n, int reverse, int bits, const uint64_t *d_out_off,
                             uint32_t *d_out_idx, uint64_t out_cap)
{
    int rc = check_batch(ctx, "opal_features", d_reads, d_frag_off, d_frag_len, n);
    if (rc) return rc;
    if (bits < 1 || bits > 32) return opal::fail(ctx, OPAL_EINVAL, "opal_features: need 1 <= bits <= 32");
    if (n && (!d_out_off || !d_out_idx)) return opal::fail(ctx, OPAL_EINVAL, "opal_features: null output pointer");
    if (n == 0) return OPAL_OK;
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Batch B{d_reads, reads_bytes, d_frag_off, d_frag_len, n};
    Geom G{ctx->d_pos, ctx->k, ctx->P, ctx->t, (ctx->t + 3) & ~3, reverse != 0, ctx->hash_all, ctx->p0_run};
    const uint32_t mask = bits == 32 ? 0xffffffffu : ((1u << bits) - 1u);
    int grid = 1;
    rc = grid_for(ctx, (const void *)&emit_idx_kernel, n, &grid);
    if (rc) return rc;
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_queue, 0, sizeof(unsigned long long), ctx->stream));
    uint64_t qchunk = EMIT_QCHUNK;
    if (const char *e = getenv("OPAL_EMIT_QCHUNK")) qchunk = std::max<uint64_t>(1, strtoull(e, nullptr, 10));   // experiments
    hipLaunchKernelGGL(emit_idx_kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, B, G, mask, d_out_off, d_out_idx, out_cap, ctx->d_queue, qchunk);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    return OPAL_OK;
}

extern "C" int opal_feature_names(opal_ctx *ctx, const uint8_t *d_reads, uint64_t reads_bytes, const uint64_t *d_frag_off,
                                  const uint32_t *d_frag_len, uint64_t n, int reverse, const uint64_t *d_out_off,
                                  uint8_t *d_out_names, uint64_t out_cap)
{
    int rc = check_batch(ctx, "opal_feature_names", d_reads, d_frag_off, d_frag_len, n);
    if (rc) return rc;
    if (n && (!d_out_off || !d_out_names)) return opal::fail(ctx, OPAL_EINVAL, "opal_feature_names: null output pointer");
    if (n == 0) return OPAL_OK;
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Batch B{d_reads, reads_bytes, d_frag_off, d_frag_len, n};
    Geom G{ctx->d_pos, ctx->k, ctx->P, ctx->t, (ctx->t + 3) & ~3, reverse != 0, ctx->hash_all, ctx->p0_run};
    int grid = 1;
    rc = grid_for(ctx, (const void *)&emit_names_kernel, n, &grid);
    if (rc) return rc;
    hipLaunchKernelGGL(emit_names_kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, B, G, d_out_off, d_out_names, out_cap);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    return OPAL_OK;
}

extern "C" const char *opal_predict_kernel_name(opal_ctx *ctx)
{
    if (!ctx || ctx->K == 0) return "";
    return pick_variant(ctx->K).name;
}

extern "C" int opal_predict(opal_ctx *ctx, const uint8_t *d_reads, uint64_t reads_bytes, const uint64_t *d_frag_off,
                            const uint32_t *d_frag_len, uint64_t n, int reverse, int32_t *d_out_class, float *d_out_scores)
{
    int rc = check_batch(ctx, "opal_predict", d_reads, d_frag_off, d_frag_len, n);
    if (rc) return rc;
    if (!ctx->d_w || ctx->K == 0) return opal::fail(ctx, OPAL_ESTATE, "opal_predict: no weights loaded");
    if (n && !d_out_class) return opal::fail(ctx, OPAL_EINVAL, "opal_predict: null d_out_class");
    if (n == 0) return OPAL_OK;
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Batch B{d_reads, reads_bytes, d_frag_off, d_frag_len, n};
    Geom G{ctx->d_pos, ctx->k, ctx->P, ctx->t, (ctx->t + 3) & ~3, reverse != 0, ctx->hash_all, ctx->p0_run};
    const uint32_t mask = ctx->bits == 32 ? 0xffffffffu : ((1u << ctx->bits) - 1u);
    const uint32_t const_idx = OPAL_VW_CONSTANT & mask;
    const Variant &v = pick_variant(ctx->K);
    // cache policy of the row loads by table size (see nt_load); OPAL_PREDICT_NT=0/1 overrides it for experiments
    bool nt = ctx->n_slots * sizeof(float) >= NT_TABLE_BYTES;
    if (const char *e = getenv("OPAL_PREDICT_NT")) nt = atoi(e) != 0;
    const void *fn = nt ? v.fn_nt : v.fn;
    int grid = 1;
    rc = grid_for(ctx, fn, n, &grid);
    if (rc) return rc;
    const float *tab = ctx->d_w;
    int K = ctx->K, addc = ctx->add_constant, clip = ctx->clip;
    float minl = ctx->min_label, maxl = ctx->max_label;
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_queue, 0, sizeof(unsigned long long), ctx->stream));
    void *args[] = {&B, &G, &tab, (void *)&mask, &K, &addc, (void *)&const_idx, &clip, &minl, &maxl, &d_out_class, &d_out_scores, &ctx->d_queue};
    OPAL_HIP_CHECK(ctx, hipLaunchKernel(fn, dim3(grid), dim3(BLOCK), args, 0, ctx->stream));
    return OPAL_OK;
}
