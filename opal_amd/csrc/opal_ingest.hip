// GPU-side FASTA ingest: the text of a FASTA/FASTQ-ish file, resident in HBM, is turned
// into the fragment batch layout of include/opal_hip.h without the host ever touching a sequence byte.
// Semantics = fasta_functions.fasta_reader over open(path,'r') (reference util/fasta_functions.py:9-43, universal
// newlines): '>' at a line start opens a record; a line starting with '+' mutes everything up to the next '>';
// other lines are concatenated without their terminator; text before the first '>' is dropped (unless the file has
// no '>' at all: then it is one unnamed record); a record with an empty sequence is kept mid-file and dropped at
// end of file.  Record names are not produced (the predict path never uses them).
//
// The reader is a transducer over bytes with a tiny state (kind of the current line, last '>'/'+' event), so the
// number of sequence bytes in a stretch of text is an affine function of the state it is entered with.  A stretch
// is summarised as (headers, last event, last line kind, cA, cB, cC): cA = bytes on the line the stretch starts in
// (sequence iff that line is a data line of an unmuted record), cB = bytes on its own data lines before its first
// event (sequence iff the record it continues is unmuted), cC = bytes that are sequence whatever came before.
// Summaries compose associatively, so the text is read ONCE (opal_ingest_fasta):
//   1. ing_fused_kernel     a single-pass chained scan.  Persistent blocks take 16 KiB tiles in order; a block analyses its
//                           tile (the bytes and their masks stay in registers), publishes the tile's summary, looks back
//                           over its predecessors' published summaries / states until it knows the reader state and the
//                           sequence position its tile starts at, publishes its own end state, and then packs: the 2-bit
//                           code of every sequence byte goes straight into ONE packed stream of all sequence bytes
//                           (LDS-staged, whole dwords stored), headers record where their sequence starts, bytes outside
//                           ACGT flag their record as raw.  The two dwords a tile may share with its neighbours go to a
//                           side record instead of HBM;
//   2. ing_edges_kernel x2  zeroes those shared dwords, then ORs every tile's share in (one thread per tile)
//   3. ing_records_kernel   per record: length, offset (packed fragments are addressed in BASES, so a record is a
//                           view into the stream and needs no alignment), space for raw records behind the stream
//   4. ing_rawcopy_kernel   only when some record is raw: second pass that copies the bytes of those records
// No intermediate copy of the sequence exists; one host synchronisation returns the counts.
// opal_ingest_scan (sizes only, for callers that want exact buffers first) keeps the two-level scan of tile summaries:
// ing_summary_kernel + ing_scan_local_kernel + ing_scan_kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <type_traits>

#include "opal_device.h"
#include "opal_internal.h"

namespace {

constexpr int CPT = 4;                               // 16-byte chunks per thread (contiguous)
constexpr int ING_THREAD_BYTES = 16 * CPT;
constexpr int ING_TILE = BLOCK * ING_THREAD_BYTES;   // bytes per block
constexpr int ING_LDS_WORDS = ING_TILE / 16 + 2;     // 2-bit codes of one tile, congruent with the output dwords
constexpr uint32_t EVEN = 0x55555555u;               // per-byte flags of a chunk live in the even bits: byte j <-> bit 2j

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): chunk indices stay compile-time constants, so the
// per-chunk arrays live in registers (a loop the compiler leaves rolled would move them to scratch or LDS)
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// ---- summaries ------------------------------------------------------------------------------------------------
// kind: 0 = no line start seen, 1 = header or '+' line, 3 = data line;  ev: 0 = no event seen, 1 = '>', 2 = '+'
struct Sum { uint32_t h, ev, kind, cA, cB, cC; };
__device__ __forceinline__ Sum sum_identity() { return Sum{0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ Sum sum_combine(const Sum &a, const Sum &b)   // a, then b
{
    Sum r;
    r.h = a.h + b.h;
    r.ev = b.ev ? b.ev : a.ev;
    r.kind = b.kind ? b.kind : a.kind;
    // b.cA lies on the line a ends in: unknown to a (-> cA), one of a's data lines (-> like cB of a), or not sequence
    const uint32_t u = b.cB + (a.kind == 3u ? b.cA : 0u);
    r.cA = a.cA + (a.kind == 0u ? b.cA : 0u);
    r.cB = a.cB + (a.ev == 0u ? u : 0u);
    r.cC = a.cC + b.cC + (a.ev == 1u ? u : 0u);
    return r;
}
__device__ __forceinline__ Sum sum_shfl_up(const Sum &v, int d)
{
    Sum r;
    const uint32_t st = (uint32_t)__shfl_up((int)(v.ev | (v.kind << 2)), d, WAVE);
    r.h = (uint32_t)__shfl_up((int)v.h, d, WAVE);
    r.ev = st & 3u;
    r.kind = st >> 2;
    r.cA = (uint32_t)__shfl_up((int)v.cA, d, WAVE);
    r.cB = (uint32_t)__shfl_up((int)v.cB, d, WAVE);
    r.cC = (uint32_t)__shfl_up((int)v.cC, d, WAVE);
    return r;
}

// the reader's concrete state in front of a stretch: headers seen, sequence bytes so far, last event, line kind
struct Pre { uint64_t h, cnt; uint32_t ev, kind; };
__device__ __forceinline__ Pre pre_apply(const Pre &p, const Sum &s)
{
    Pre r;
    const bool open = p.ev != 2u;
    r.cnt = p.cnt + ((open && p.kind == 3u) ? s.cA : 0u) + (open ? s.cB : 0u) + s.cC;
    r.h = p.h + s.h;
    r.ev = s.ev ? s.ev : p.ev;
    r.kind = s.kind ? s.kind : p.kind;
    return r;
}

// Block-wide scan of per-thread summaries (NW waves): returns this thread's exclusive prefix, the block total
// through `total`.  Contains two __syncthreads.
template <int NW = WPB>
__device__ __forceinline__ Sum block_excl_scan_sum(const Sum &v, Sum *s_w, Sum &total)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    Sum incl = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const Sum o = sum_shfl_up(incl, d);
        if (lane >= d) incl = sum_combine(o, incl);
    }
    if (lane == WAVE - 1) s_w[wv] = incl;
    __syncthreads();
    Sum base = sum_identity();
    total = sum_identity();
#pragma unroll
    for (int w = 0; w < NW; w++) {
        if (w < wv) base = sum_combine(base, s_w[w]);
        total = sum_combine(total, s_w[w]);
    }
    Sum excl = sum_shfl_up(incl, 1);
    if (lane == 0) excl = sum_identity();
    __syncthreads();
    return sum_combine(base, excl);
}

// ---- 16 bytes at a time ------------------------------------------------------------------------------------------
// The 2-bit fields in bits 0-1 of every byte of f0..f3 (byte b of f_d belongs to text byte 4d+b) as one word in text
// order: field j in bits 2j, 2j+1.  Or-ing the four words puts field (b,d) at bit 8b+2d; two delta swaps exchange the
// roles of b and d (a 4x4 transpose of 2-bit fields).
__device__ __forceinline__ uint32_t gather_fields(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3)
{
    uint32_t F = f0 | (f1 << 2) | (f2 << 4) | (f3 << 6);
    uint32_t t = ((F >> 12) ^ F) & 0x0000f0f0u;       // exchange address bits 4 and 2
    F ^= t | (t << 12);
    t = ((F >> 6) ^ F) & 0x00cc00ccu;                  // exchange address bits 3 and 1
    F ^= t | (t << 6);
    return F;
}

__device__ __forceinline__ uint32_t chunk_byte(const uint32_t (&x)[4], uint32_t j)
{
    // the four dwords as values first: a select between array elements becomes a select between addresses, which
    // sends the whole array to scratch / LDS
    const uint32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    const uint32_t lo = j < 4u ? x0 : x1, hi = j < 12u ? x2 : x3;
    const uint32_t w = j < 8u ? lo : hi;
    return (w >> ((j & 3u) * 8u)) & 0xffu;
}

// Terminator flags of a chunk: bit 2j = byte j is '\r', bit 2j+1 = byte j is '\n'.
// v_perm_b32 with the DATA as selector is an exact one-byte-value test: selector 12 yields 0x00, selectors >= 13 yield
// 0xff, selectors 0..11 pick a source byte or a source sign (all 0xff here).  x ^ 0x06 sends '\n' (10) and nothing else
// to 12, x ^ 0x01 sends '\r' (13) and nothing else to 12.
__device__ __forceinline__ uint32_t term_fields(const uint32_t (&x)[4])
{
    uint32_t f[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t not_nl = __builtin_amdgcn_perm(0xffffffffu, 0xffffffffu, x[d] ^ 0x06060606u);   // 0x00 at '\n'
        const uint32_t not_cr = __builtin_amdgcn_perm(0xffffffffu, 0xffffffffu, x[d] ^ 0x01010101u);   // 0x00 at '\r'
        f[d] = (~not_nl & 0x02020202u) | (~not_cr & 0x01010101u);
    }
    return gather_fields(f[0], f[1], f[2], f[3]);
}

// The thread's bytes: CPT chunks of 16, bytes past the end of the text read as '\n'; valid[c] = even-bit mask of the
// bytes of chunk c that exist; prev = the byte in front of the thread's first byte (0 at the start of the text).
__device__ __forceinline__ void load_chunks(const uint8_t *__restrict__ text, uint64_t n, uint64_t base, uint32_t (&x)[CPT][4],
                                            uint32_t (&valid)[CPT], uint32_t &prev)
{
    prev = (base > 0 && base <= n) ? text[base - 1] : 0u;
    if (base + ING_THREAD_BYTES <= n && (reinterpret_cast<uintptr_t>(text) & 15u) == 0) {
        static_for<CPT>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            const uint4 v = *reinterpret_cast<const uint4 *>(text + base + 16 * c);
            x[c][0] = v.x; x[c][1] = v.y; x[c][2] = v.z; x[c][3] = v.w;
            valid[c] = EVEN;
        });
    } else {
        static_for<CPT>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            uint32_t nv = 0;
            static_for<4>([&](auto di) {
                constexpr int d = decltype(di)::value;
                uint32_t w = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint64_t p = base + 16 * c + 4 * d + b;
                    const uint32_t ch = p < n ? (uint32_t)text[p] : (uint32_t)'\n';
                    nv += p < n ? 1u : 0u;
                    w |= ch << (8 * b);
                }
                x[c][d] = w;
            });
            valid[c] = nv >= 16u ? EVEN : (((1u << (2u * nv)) - 1u) & EVEN);
        });
    }
}

// Line starts and non-terminator bytes of one chunk (even-bit masks).  nl_in / cr_in: the byte in front of the chunk
// is '\n' / '\r'; on return they describe the chunk's last byte.
__device__ __forceinline__ void chunk_lines(const uint32_t (&x)[4], uint32_t valid, bool first, uint32_t &nl_in, uint32_t &cr_in,
                                            uint32_t &LS, uint32_t &NT)
{
    const uint32_t I = term_fields(x);
    const uint32_t nlo = I & ~EVEN, cre = I & EVEN;
    // a byte starts a line after '\n', or after a '\r' that is not followed by '\n' (universal newlines)
    LS = ((((nlo << 1) | nl_in) | (((cre << 2) | cr_in) & ~(nlo >> 1))) & EVEN) | (first ? 1u : 0u);
    LS &= valid;
    NT = ~((nlo >> 1) | cre) & valid;
    nl_in = I >> 31;
    cr_in = (I >> 30) & 1u;
}

// Symbolic walk of one chunk: which bytes are sequence under which condition on the state the THREAD is entered with.
struct Walk { uint32_t kind, ev, h; };                // running state of the thread (0 = still the incoming one)
struct Masks { uint32_t mA, mB, mC, HD; };           // cA / cB / cC bytes and the '>' line starts of the chunk
__device__ __forceinline__ Masks walk_chunk(const uint32_t (&x)[4], uint32_t LS, uint32_t NT, Walk &w)
{
    Masks m{0, 0, 0, 0};
    const uint32_t pre = NT & (LS ? ((LS & (0u - LS)) - 1u) : 0xffffffffu);   // bytes of the line the chunk starts in
    if (w.kind == 0u) m.mA = pre;
    else if (w.kind == 3u) {
        if (w.ev == 0u) m.mB = pre;
        else if (w.ev == 1u) m.mC = pre;
    }
    uint32_t rem = LS;
    while (rem) {
        const uint32_t bit = rem & (0u - rem);
        rem ^= bit;
        const uint32_t next = rem & (0u - rem);             // 0 when this is the chunk's last line start
        const uint32_t c = chunk_byte(x, (uint32_t)__builtin_ctz(bit) >> 1);
        if (c == '>') { w.h++; w.ev = 1u; w.kind = 1u; m.HD |= bit; }
        else if (c == '+') { w.ev = 2u; w.kind = 1u; }
        else {
            w.kind = 3u;
            const uint32_t s = NT & (next - 1u) & ~(bit - 1u);   // bytes from this line start up to the next one
            if (w.ev == 0u) m.mB |= s;
            else if (w.ev == 1u) m.mC |= s;
        }
    }
    return m;
}

// Everything the two text passes share: loads the thread's bytes, walks them, returns the thread's summary.
__device__ __forceinline__ Sum analyse_thread(const uint8_t *__restrict__ text, uint64_t n, uint64_t base, uint32_t (&x)[CPT][4],
                                              Masks (&m)[CPT])
{
    uint32_t valid[CPT], prev;
    load_chunks(text, n, base, x, valid, prev);
    uint32_t nl_in = prev == '\n' ? 1u : 0u, cr_in = prev == '\r' ? 1u : 0u;
    Walk w{0, 0, 0};
    Sum s = sum_identity();
    static_for<CPT>([&](auto ci) {
        constexpr int c = decltype(ci)::value;
        uint32_t LS, NT;
        chunk_lines(x[c], valid[c], base == 0 && c == 0, nl_in, cr_in, LS, NT);
        m[c] = walk_chunk(x[c], LS, NT, w);
        s.cA += (uint32_t)__builtin_popcount(m[c].mA);
        s.cB += (uint32_t)__builtin_popcount(m[c].mB);
        s.cC += (uint32_t)__builtin_popcount(m[c].mC);
    });
    s.h = w.h; s.ev = w.ev; s.kind = w.kind;
    return s;
}

// ---- pass 1: one summary per tile --------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void ing_summary_kernel(const uint8_t *__restrict__ text, uint64_t n, Sum *__restrict__ tile_sum)
{
    __shared__ Sum s_w[WPB];
    const uint64_t base = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)threadIdx.x * ING_THREAD_BYTES;
    uint32_t x[CPT][4];
    Masks m[CPT];
    Sum v = analyse_thread(text, n, base, x, m);
    // ordered reduction (the combine is not commutative): lane i takes in lanes i+d .. after its own
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        Sum o;
        const uint32_t st = (uint32_t)__shfl_down((int)(v.ev | (v.kind << 2)), d, WAVE);
        o.h = (uint32_t)__shfl_down((int)v.h, d, WAVE);
        o.ev = st & 3u; o.kind = st >> 2;
        o.cA = (uint32_t)__shfl_down((int)v.cA, d, WAVE);
        o.cB = (uint32_t)__shfl_down((int)v.cB, d, WAVE);
        o.cC = (uint32_t)__shfl_down((int)v.cC, d, WAVE);
        if (lane + d < WAVE) v = sum_combine(v, o);
    }
    if (lane == 0) s_w[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        Sum t = s_w[0];
#pragma unroll
        for (int w = 1; w < WPB; w++) t = sum_combine(t, s_w[w]);
        tile_sum[blockIdx.x] = t;
    }
}

// ---- pass 2a: scan of the tile summaries inside segments of 256 tiles (one block each, one tile per thread) -----------
constexpr int SEG_TILES = BLOCK;
__global__ __launch_bounds__(BLOCK) void ing_scan_local_kernel(const Sum *__restrict__ tile_sum, uint64_t ntiles, Sum *__restrict__ tile_excl,
                                                                Sum *__restrict__ seg_sum)
{
    __shared__ Sum s_w[WPB];
    const uint64_t i = (uint64_t)blockIdx.x * SEG_TILES + threadIdx.x;
    const Sum v = i < ntiles ? tile_sum[i] : sum_identity();
    Sum total;
    const Sum excl = block_excl_scan_sum(v, s_w, total);
    if (i < ntiles) tile_excl[i] = excl;                   // what the segment holds in front of tile i
    if (threadIdx.x == 0) seg_sum[blockIdx.x] = total;
}

// ---- pass 2b: the state every segment starts in (one block over the segment totals) ----------------------------------------------------------------------
// fin: [0] headers in the text, [1] sequence bytes T, [2] cursor of the raw area (starts behind the packed stream),
//      [3] raw records, [4] a record is longer than 2^31-1, [5] the last record is empty (dropped), [6] raw area overflow,
//      [7] a byte outside ACGT in front of the first header (makes record 0 raw when the text has no header at all),
//      [8] a look-back gave up waiting (never seen; the exit every spinning wave reaches)
enum { F_H = 0, F_T, F_CURSOR, F_NRAW, F_TOOLONG, F_DROPPED, F_OVERFLOW, F_PREBAD, F_STALL, F_TICKET, F_WORDS = 16 };
constexpr int SC_ITEMS = 2;
constexpr int SC_WAVES = 4;
constexpr int SC_BLOCK = SC_WAVES * WAVE;
// a round covers SC_BLOCK * SC_ITEMS segments of SEG_TILES tiles = 2 GiB of text: the u32 counts of a Sum hold
static_assert((uint64_t)SC_BLOCK * SC_ITEMS * SEG_TILES * ING_TILE <= (1ull << 31), "u32 counts inside one scan round");
// (written for any list of summaries: here tile_sum / ntiles / tile_pre are the segment totals, their count, the segment states)
__global__ __launch_bounds__(SC_BLOCK) void ing_scan_kernel(const Sum *__restrict__ tile_sum, uint64_t ntiles, Pre *__restrict__ tile_pre,
                                                          uint64_t *__restrict__ fin)
{
    __shared__ Sum s_w[SC_WAVES];
    __shared__ Pre s_carry;
    if (threadIdx.x == 0) s_carry = Pre{0, 0, 0, 0};
    __syncthreads();
    for (uint64_t b0 = 0; b0 < ntiles; b0 += (uint64_t)SC_BLOCK * SC_ITEMS) {
        const uint64_t i0 = b0 + (uint64_t)threadIdx.x * SC_ITEMS;
        Sum v[SC_ITEMS], agg = sum_identity();
#pragma unroll
        for (int j = 0; j < SC_ITEMS; j++) {
            v[j] = i0 + j < ntiles ? tile_sum[i0 + j] : sum_identity();
            agg = sum_combine(agg, v[j]);
        }
        Sum total;
        const Sum excl = block_excl_scan_sum<SC_WAVES>(agg, s_w, total);
        const Pre carry = s_carry;
        Pre run = pre_apply(carry, excl);
#pragma unroll
        for (int j = 0; j < SC_ITEMS; j++) {
            if (i0 + j < ntiles) tile_pre[i0 + j] = run;
            run = pre_apply(run, v[j]);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = pre_apply(carry, total);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        fin[F_H] = s_carry.h;
        fin[F_T] = s_carry.cnt;
        fin[F_CURSOR] = (((s_carry.cnt + 3) / 4) + 15) & ~15ull;   // raw records go behind the packed stream
    }
}

// ---- the single pass: analyse, chain, pack ---------------------------------------------------------------------------
// removes the 2-bit fields of v whose field in keep (both bits set for a kept field) is clear
__device__ __forceinline__ uint32_t squeeze_fields(uint32_t v, uint32_t keep)
{
    v &= keep;
    while (keep != 0xffffffffu) {
        const uint32_t b = (uint32_t)__builtin_ctz(~keep);    // lowest hole
        const uint32_t up = keep >> b;
        if (!up) break;                                        // nothing kept above it
        const uint32_t run = (uint32_t)__builtin_ctz(up);
        const uint32_t low = (1u << b) - 1u;
        v = (v & low) | ((v >> run) & ~low);
        keep = (keep & low) | ((keep >> run) & ~low);
    }
    return v;
}


// What a tile publishes for its successors, one 64-bit word at a time (bit 63 = the word is there; the words are zeroed
// before the launch).  Every word validates itself, so relaxed device-scope loads and stores are all the ordering needed.
//   agg     the tile's summary: cA, cB, cC in 15 bits each (<= 16384), h in 14 (a header takes two bytes), ev, kind
//   p0, p1  the reader state BEHIND the tile: (kind, ev, headers so far) and the sequence bytes so far
struct Desc { unsigned long long agg, p0, p1, pad; };
constexpr unsigned long long D_VALID = 1ull << 63;
static_assert(ING_TILE <= 16384, "summary counts of a tile in 15 bits");
__device__ __forceinline__ unsigned long long enc_agg(const Sum &s)
{
    return D_VALID | ((unsigned long long)s.kind << 61) | ((unsigned long long)s.ev << 59) | ((unsigned long long)s.h << 45) |
           ((unsigned long long)s.cC << 30) | ((unsigned long long)s.cB << 15) | (unsigned long long)s.cA;
}
__device__ __forceinline__ Sum dec_agg(unsigned long long v)
{
    Sum s;
    s.cA = (uint32_t)v & 0x7fffu; s.cB = (uint32_t)(v >> 15) & 0x7fffu; s.cC = (uint32_t)(v >> 30) & 0x7fffu;
    s.h = (uint32_t)(v >> 45) & 0x3fffu; s.ev = (uint32_t)(v >> 59) & 3u; s.kind = (uint32_t)(v >> 61) & 3u;
    return s;
}
__device__ __forceinline__ Pre dec_pre(unsigned long long p0, unsigned long long p1)
{
    Pre r;
    r.h = p0 & ((1ull << 59) - 1); r.ev = (uint32_t)(p0 >> 59) & 3u; r.kind = (uint32_t)(p0 >> 61) & 3u;
    r.cnt = p1 & ~D_VALID;
    return r;
}
__device__ __forceinline__ unsigned long long ld_desc(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_desc(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the state a tile starts in = the state behind the tile in front of it (after the kernel: plain loads)
__device__ __forceinline__ Pre tile_entry_state(const Desc *__restrict__ desc, uint64_t t)
{
    return t ? dec_pre(desc[t - 1].p0, desc[t - 1].p1) : Pre{0, 0, 0, 0};
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int l)
{
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, l, WAVE), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), l, WAVE);
    return ((uint64_t)hi << 32) | lo;
}

// Decoupled look-back of one wave: lane i looks at tile t-1-i (then 64 further back, ...).  The nearest tile whose end
// state is published gives the start; the summaries of the tiles between it and t are applied on top, in text order.
// Tiles are tickets drawn in the order blocks start (ing_fused_kernel), so the tile this wave waits for is always being
// worked on by a resident block; as a last line of defence the wave gives up after LOOKBACK_SPINS polls and the call fails
// with OPAL_EHIP instead of hanging.
constexpr uint32_t LOOKBACK_SPINS = 1u << 20;
__device__ __forceinline__ Pre look_back(const Desc *desc, uint64_t t, int lane, uint64_t *fin)
{
    Sum acc = sum_identity();                 // the tiles between the window and t (wave-uniform)
    int64_t base = (int64_t)t - 1;
    uint32_t spins = 0;
    for (;;) {
        const int64_t idx = base - lane;
        unsigned long long a = 0, p0 = D_VALID, p1 = D_VALID;    // in front of the text: the reader's initial state
        if (idx >= 0) {
            p0 = ld_desc(&desc[idx].p0);
            p1 = ld_desc(&desc[idx].p1);
            a = ld_desc(&desc[idx].agg);
        }
        const bool has_p = ((p0 & p1) >> 63) != 0, has_a = (a >> 63) != 0;
        const uint64_t bp = __ballot(has_p), bn = __ballot(!has_p && !has_a);
        const int j = bp ? __builtin_ctzll(bp) : WAVE;           // the nearest tile with a published end state
        const uint64_t between = j == WAVE ? ~0ull : ((1ull << j) - 1ull);
        if (bn & between) {                                      // a tile in between has published nothing yet
            if (++spins > LOOKBACK_SPINS) {
                if (lane == 0) fin[F_STALL] = 1;
                return Pre{0, 0, 0, 0};
            }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        Sum v = lane < j ? dec_agg(a) : sum_identity();
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {                     // lane + d holds the EARLIER tile: it goes first
            Sum o;
            const uint32_t st = (uint32_t)__shfl_down((int)(v.ev | (v.kind << 2)), d, WAVE);
            o.h = (uint32_t)__shfl_down((int)v.h, d, WAVE);
            o.ev = st & 3u; o.kind = st >> 2;
            o.cA = (uint32_t)__shfl_down((int)v.cA, d, WAVE);
            o.cB = (uint32_t)__shfl_down((int)v.cB, d, WAVE);
            o.cC = (uint32_t)__shfl_down((int)v.cC, d, WAVE);
            if (lane + d < WAVE) v = sum_combine(o, v);
        }
        Sum win;
        win.h = (uint32_t)__shfl((int)v.h, 0, WAVE); win.ev = (uint32_t)__shfl((int)v.ev, 0, WAVE); win.kind = (uint32_t)__shfl((int)v.kind, 0, WAVE);
        win.cA = (uint32_t)__shfl((int)v.cA, 0, WAVE); win.cB = (uint32_t)__shfl((int)v.cB, 0, WAVE); win.cC = (uint32_t)__shfl((int)v.cC, 0, WAVE);
        acc = sum_combine(win, acc);
        if (j < WAVE) return pre_apply(dec_pre(shfl64(p0, j), shfl64(p1, j)), acc);
        base -= WAVE;
    }
}

// a tile's share of the (at most two) dwords of the packed stream that it may have in common with its neighbours
struct Edge { uint64_t g0; uint32_t nd, first, last; };

// 2-bit codes of all sequence bytes into the packed stream out32 (base p of the concatenated sequence in bits 2*(p&15) of
// dword p>>4), seq_start[r] for every header, raw_flag[r] for every record that holds a byte outside ACGT.
__global__ __launch_bounds__(BLOCK) void ing_fused_kernel(const uint8_t *__restrict__ text, uint64_t n, uint64_t ntiles, Desc *desc,
                                                           Edge *__restrict__ edges, uint64_t *__restrict__ fin, uint32_t *__restrict__ out32,
                                                           uint64_t cap_words, uint64_t *__restrict__ seq_start,
                                                           uint32_t *__restrict__ raw_flag, uint64_t max_records)
{
    __shared__ Sum s_w[WPB];
    __shared__ uint32_t s_bits[ING_LDS_WORDS];
    __shared__ Pre s_tp;
    __shared__ uint64_t s_ticket;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    {
        // The tile is a ticket drawn when the block starts, not blockIdx.x: HIP does not promise that blocks are dispatched
        // in index order, and a look-back may only wait for tiles whose blocks are already running.  Tickets are handed out
        // in the order blocks start, so every tile below this one belongs to a block that is resident or done.
        if (threadIdx.x == 0) s_ticket = atomicAdd((unsigned long long *)&fin[F_TICKET], 1ull);
        for (int i = threadIdx.x; i < ING_LDS_WORDS; i += BLOCK) s_bits[i] = 0;   // ordered by the scan's barriers
        __syncthreads();
        const uint64_t t = s_ticket;
        const uint64_t base = t * ING_TILE + (uint64_t)threadIdx.x * ING_THREAD_BYTES;
        uint32_t x[CPT][4];
        Masks m[CPT];
        const Sum mine = analyse_thread(text, n, base, x, m);
        Sum total;
        const Sum excl = block_excl_scan_sum(mine, s_w, total);
        if (wv == 0 && lane == 0) st_desc(&desc[t].agg, enc_agg(total));
        // what does not depend on the state the tile starts in, while the predecessors publish: the 2-bit codes and the
        // bytes outside ACGT of every chunk (the text bytes themselves are not needed after this)
        uint32_t codes_c[CPT], bad_c[CPT];
        static_for<CPT>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            uint32_t q[4], nzb[4];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                // A=0x41 C=0x43 G=0x47 T=0x54: (ch >> 1) & 3 gives 0,1,3,2 -> fix the G/T order
                q[d] = (x[c][d] >> 1) & 0x03030303u;
                q[d] ^= (q[d] >> 1) & 0x01010101u;
                const uint32_t diff = x[c][d] ^ __builtin_amdgcn_perm(0u, 0x54474341u, q[d]);     // "ACGT"[code] == byte ?
                nzb[d] = ((((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) >> 7) & 0x01010101u;        // 1 in every byte outside ACGT
            }
            codes_c[c] = gather_fields(q[0], q[1], q[2], q[3]);
            bad_c[c] = gather_fields(nzb[0], nzb[1], nzb[2], nzb[3]);
        });
        if (wv == 0) {
            const Pre tp0 = t ? look_back(desc, t, lane, fin) : Pre{0, 0, 0, 0};
            if (lane == 0) {
                const Pre end = pre_apply(tp0, total);
                st_desc(&desc[t].p0, D_VALID | ((unsigned long long)end.kind << 61) | ((unsigned long long)end.ev << 59) | end.h);
                st_desc(&desc[t].p1, D_VALID | end.cnt);
                s_tp = tp0;
                if (t == ntiles - 1) {
                    fin[F_H] = end.h;
                    fin[F_T] = end.cnt;
                    fin[F_CURSOR] = (((end.cnt + 3) / 4) + 15) & ~15ull;   // raw records go behind the packed stream
                }
            }
        }
        __syncthreads();
        const Pre tp = s_tp;                                                   // the state this tile starts in
        const Pre in = pre_apply(tp, excl);
        const bool open = in.ev != 2u, open_a = open && in.kind == 3u;
        uint64_t cnt = in.cnt, hrun = in.h;
        uint32_t lpos = (uint32_t)(tp.cnt & 15u) + (uint32_t)(in.cnt - tp.cnt);
        bool prebad = false;
        static_for<CPT>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            const uint32_t CM = m[c].mC | (open ? m[c].mB : 0u) | (open_a ? m[c].mA : 0u);
            const uint64_t h_in = hrun;                                   // headers in front of this chunk
            uint32_t hd = m[c].HD;
            while (hd) {
                const uint32_t bit = hd & (0u - hd);
                hd ^= bit;
                const uint64_t r = hrun++;
                if (r < max_records) seq_start[r] = cnt + (uint32_t)__builtin_popcount(CM & (bit - 1u));
            }
            uint32_t codes = codes_c[c], bad = bad_c[c];
            bad &= CM;
            uint64_t last_r = ~0ull;
            while (bad) {
                const uint32_t bit = bad & (0u - bad);
                bad ^= bit;
                const uint64_t hh = h_in + (uint32_t)__builtin_popcount(m[c].HD & (bit - 1u));
                if (hh == 0) { prebad = true; continue; }              // in front of the first header: record 0 only if there is none
                const uint64_t r = hh - 1;
                if (r == last_r || r >= max_records) continue;
                last_r = r;
                if (__atomic_load_n(&raw_flag[r], __ATOMIC_RELAXED) == 0u && atomicExch(&raw_flag[r], 1u) == 0u)
                    atomicAdd((unsigned long long *)&fin[F_NRAW], 1ull);
            }
            const uint32_t nf = (uint32_t)__builtin_popcount(CM);
            if (nf) {
                codes = squeeze_fields(codes, CM | (CM << 1));
                const uint32_t lbit = 2u * lpos, sh = lbit & 31u;
                atomicOr(&s_bits[lbit >> 5], codes << sh);
                if (sh && (codes >> (32u - sh))) atomicOr(&s_bits[(lbit >> 5) + 1], codes >> (32u - sh));
            }
            lpos += nf;
            cnt += nf;
        });
        if (prebad) fin[F_PREBAD] = 1;                                       // plain store of a constant
        __syncthreads();
        const uint64_t tot = pre_apply(tp, total).cnt - tp.cnt;
        const uint64_t g0 = tp.cnt >> 4;
        const uint32_t nd = (uint32_t)(((tp.cnt & 15u) + tot + 15u) >> 4);
        // interior dwords hold bases of this tile only; the first and the last may be shared with neighbouring tiles
        for (uint32_t i = threadIdx.x + 1u; i + 1u < nd; i += BLOCK)
            if (g0 + i < cap_words) out32[g0 + i] = s_bits[i];
        if (threadIdx.x == 0) edges[t] = Edge{g0, nd, nd ? s_bits[0] : 0u, nd > 1u ? s_bits[nd - 1u] : 0u};
    }
}

// The shared dwords: zeroed (ZERO), then every tile's share OR-ed in.  One thread per tile.
template <bool ZERO>
__global__ __launch_bounds__(BLOCK) void ing_edges_kernel(const Edge *__restrict__ edges, uint64_t ntiles, uint32_t *__restrict__ out32,
                                                           uint64_t cap_words)
{
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= ntiles) return;
    const Edge e = edges[t];
    if (e.nd == 0) return;
    const uint64_t gl = e.g0 + e.nd - 1;
    if (ZERO) {
        if (e.g0 < cap_words) out32[e.g0] = 0u;
        if (gl < cap_words) out32[gl] = 0u;
    } else {
        if (e.first && e.g0 < cap_words) atomicOr(&out32[e.g0], e.first);
        if (e.last && gl < cap_words) atomicOr(&out32[gl], e.last);
    }
}

// ---- raw records: the bytes of the records flagged raw to out8[frag_off[r] + i] (a second pass over the text, only when
// such records exist)
__global__ __launch_bounds__(BLOCK) void ing_rawcopy_kernel(const uint8_t *__restrict__ text, uint64_t n, const Desc *__restrict__ desc,
                                                             const uint64_t *__restrict__ fin, const uint64_t *__restrict__ seq_start,
                                                             const uint32_t *__restrict__ raw_flag, uint64_t max_records,
                                                             uint8_t *__restrict__ out8, uint64_t reads_cap,
                                                             const uint64_t *__restrict__ frag_off)
{
    __shared__ Sum s_w[WPB];
    const uint64_t base = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)threadIdx.x * ING_THREAD_BYTES;
    uint32_t x[CPT][4];
    Masks m[CPT];
    const Sum mine = analyse_thread(text, n, base, x, m);
    Sum total;
    const Sum excl = block_excl_scan_sum(mine, s_w, total);
    const Pre in = pre_apply(tile_entry_state(desc, blockIdx.x), excl);
    const bool open = in.ev != 2u, open_a = open && in.kind == 3u;
    const uint64_t H = fin[F_H];
    uint64_t cnt = in.cnt, hrun = in.h;
    static_for<CPT>([&](auto ci) {
        constexpr int c = decltype(ci)::value;
        uint32_t CM = m[c].mC | (open ? m[c].mB : 0u) | (open_a ? m[c].mA : 0u);
        // the chunk's sequence bytes, one record (stretch between headers) at a time
        uint32_t hd = m[c].HD, from = 0xffffffffu;
        uint64_t hcur = hrun;
        while (CM) {
            const uint32_t nb = hd & (0u - hd);                     // next header in the chunk (0: none)
            uint32_t seg = CM & from & (nb - 1u);
            if (seg && (hcur != 0 || H == 0)) {
                const uint64_t r = hcur ? hcur - 1 : 0;
                if (r < max_records && raw_flag[r]) {
                    const uint64_t dst0 = frag_off[r] + (cnt - seq_start[r]);
                    while (seg) {
                        const uint32_t bit = seg & (0u - seg);
                        seg ^= bit;
                        const uint64_t dst = dst0 + (uint32_t)__builtin_popcount(CM & (bit - 1u));
                        if (dst < reads_cap) out8[dst] = (uint8_t)chunk_byte(x[c], (uint32_t)__builtin_ctz(bit) >> 1);
                    }
                }
            }
            if (!nb) break;
            hd ^= nb;
            from = ~(nb - 1u);
            hcur++;
        }
        hrun += (uint32_t)__builtin_popcount(m[c].HD);
        cnt += (uint32_t)__builtin_popcount(CM);
    });
}

// ---- records ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void ing_records_kernel(const uint64_t *__restrict__ seq_start, uint32_t *__restrict__ raw_flag,
                                                             uint64_t *__restrict__ fin, uint64_t max_records, uint64_t reads_cap,
                                                             uint64_t *__restrict__ frag_off, uint32_t *__restrict__ frag_len)
{
    const uint64_t H = fin[F_H], T = fin[F_T];
    if (H > max_records) return;                               // reported by the host
    const uint64_t n_rec = H ? H : (T ? 1 : 0);
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; r < n_rec; r += stride) {
        const uint64_t s = seq_start[r];
        const uint64_t e = (r + 1 < n_rec) ? seq_start[r + 1] : T;
        const uint64_t len = e - s;
        if (H && r == n_rec - 1 && len == 0) { fin[F_DROPPED] = 1; continue; }   // fasta_functions.py:20-22
        if (H == 0 && fin[F_PREBAD]) {                         // a text without any header is one record: all of it counts
            raw_flag[0] = 1u;
            fin[F_NRAW] = 1;
        }
        const bool raw = raw_flag[r] != 0;
        if (len > 0x7fffffffull) fin[F_TOOLONG] = 1;           // plain stores of a constant
        const uint32_t l31 = len > 0x7fffffffull ? 0x7fffffffu : (uint32_t)len;
        frag_len[r] = l31 | (raw ? OPAL_RAW_FLAG : 0u);
        if (raw) {
            const uint64_t o = atomicAdd((unsigned long long *)&fin[F_CURSOR], (unsigned long long)len);
            if (o + len > reads_cap) fin[F_OVERFLOW] = 1;
            frag_off[r] = o;                                   // raw fragments are addressed in bytes
        } else {
            frag_off[r] = s;                                   // packed fragments in bases: a view into the stream
        }
    }
}

int grow(opal_ctx *ctx, void **p, uint64_t *cap, uint64_t need)
{
    if (*cap >= need) return OPAL_OK;
    if (*p) { OPAL_HIP_CHECK(ctx, hipFree(*p)); *p = nullptr; *cap = 0; }
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) return opal::fail(ctx, OPAL_ENOMEM, std::string("opal_ingest_fasta: hipMalloc failed: ") + hipGetErrorString(e));
    *cap = need;
    return OPAL_OK;
}

}  // namespace

// scratch of opal_ingest_scan inside ctx->ing_tiles: [segment states | fin | per-tile prefixes | tile summaries | segment totals]
struct IngScratch {
    Pre *seg_pre;
    uint64_t *fin;
    Sum *tile_excl, *tile_sum, *seg_sum;
    uint64_t nseg, bytes;
};
static IngScratch ing_layout(void *base, uint64_t ntiles)
{
    IngScratch L;
    L.nseg = (ntiles + SEG_TILES - 1) / SEG_TILES;
    L.seg_pre = (Pre *)base;                                   // 24-byte entries, 8-byte aligned
    L.fin = (uint64_t *)(L.seg_pre + L.nseg);
    L.tile_excl = (Sum *)(L.fin + F_WORDS);
    L.tile_sum = L.tile_excl + ntiles;
    L.seg_sum = L.tile_sum + ntiles;
    L.bytes = L.nseg * (sizeof(Pre) + sizeof(Sum)) + F_WORDS * sizeof(uint64_t) + 2 * ntiles * sizeof(Sum) + 256;
    return L;
}
// scratch of opal_ingest_fasta inside ctx->ing_tiles: [fin | tile descriptors | tile edges]
struct FusedScratch {
    uint64_t *fin;
    Desc *desc;
    Edge *edges;
    uint64_t bytes;
};
static FusedScratch fused_layout(void *base, uint64_t ntiles)
{
    FusedScratch L;
    L.fin = (uint64_t *)base;
    L.desc = (Desc *)(L.fin + F_WORDS);
    L.edges = (Edge *)(L.desc + ntiles);
    L.bytes = F_WORDS * sizeof(uint64_t) + ntiles * (sizeof(Desc) + sizeof(Edge)) + 256;
    return L;
}

extern "C" int opal_ingest_scan(opal_ctx *ctx, const uint8_t *d_text, uint64_t nbytes, uint64_t *n_headers_out, uint64_t *seq_bytes_out)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_ingest_scan: null ctx");
    if (!n_headers_out || !seq_bytes_out) return opal::fail(ctx, OPAL_EINVAL, "opal_ingest_scan: null count outputs");
    *n_headers_out = 0;
    *seq_bytes_out = 0;
    if (nbytes == 0) return OPAL_OK;
    if (!d_text) return opal::fail(ctx, OPAL_EINVAL, "opal_ingest_scan: null device pointer");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t ntiles = (nbytes + ING_TILE - 1) / ING_TILE;
    if (ntiles > 0x7fffffffull) return opal::fail(ctx, OPAL_EINVAL, "opal_ingest_scan: text too large for one call (32 TiB)");
    const IngScratch S = ing_layout(nullptr, ntiles);
    int rc = grow(ctx, &ctx->ing_tiles, &ctx->ing_tiles_cap, S.bytes);
    if (rc) return rc;
    const IngScratch L = ing_layout(ctx->ing_tiles, ntiles);
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(L.fin, 0, F_WORDS * sizeof(uint64_t), ctx->stream));
    hipLaunchKernelGGL(ing_summary_kernel, dim3((unsigned)ntiles), dim3(BLOCK), 0, ctx->stream, d_text, nbytes, L.tile_sum);
    hipLaunchKernelGGL(ing_scan_local_kernel, dim3((unsigned)L.nseg), dim3(BLOCK), 0, ctx->stream, L.tile_sum, ntiles, L.tile_excl, L.seg_sum);
    hipLaunchKernelGGL(ing_scan_kernel, dim3(1), dim3(SC_BLOCK), 0, ctx->stream, L.seg_sum, L.nseg, L.seg_pre, L.fin);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    uint64_t host[2];
    OPAL_HIP_CHECK(ctx, hipMemcpyAsync(host, L.fin, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *n_headers_out = host[F_H];
    *seq_bytes_out = host[F_T];
    return OPAL_OK;
}

extern "C" int opal_ingest_fasta(opal_ctx *ctx, const uint8_t *d_text, uint64_t nbytes, uint8_t *d_reads_out, uint64_t reads_cap,
                                 uint64_t *d_frag_off, uint32_t *d_frag_len, uint64_t max_records, uint64_t *n_records_out,
                                 uint64_t *reads_bytes_out)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_ingest_fasta: null ctx");
    if (!n_records_out || !reads_bytes_out) return opal::fail(ctx, OPAL_EINVAL, "opal_ingest_fasta: null count outputs");
    *n_records_out = 0;
    *reads_bytes_out = 0;
    if (nbytes == 0) return OPAL_OK;
    if (!d_text || !d_reads_out || !d_frag_off || !d_frag_len) return opal::fail(ctx, OPAL_EINVAL, "opal_ingest_fasta: null device pointer");
    if (max_records == 0) return opal::fail(ctx, OPAL_EINVAL, "opal_ingest_fasta: max_records is 0");
    if (reinterpret_cast<uintptr_t>(d_reads_out) & 3u) return opal::fail(ctx, OPAL_EINVAL, "opal_ingest_fasta: d_reads_out must be 4-byte aligned");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t ntiles = (nbytes + ING_TILE - 1) / ING_TILE;
    if (ntiles > 0x7fffffffull) return opal::fail(ctx, OPAL_EINVAL, "opal_ingest_fasta: text too large for one call (32 TiB)");
    int rc = grow(ctx, &ctx->ing_rec, &ctx->ing_rec_cap, (max_records + 1) * (sizeof(uint64_t) + sizeof(uint32_t)) + 256);
    if (rc) return rc;
    rc = grow(ctx, &ctx->ing_tiles, &ctx->ing_tiles_cap, fused_layout(nullptr, ntiles).bytes);
    if (rc) return rc;
    const FusedScratch L = fused_layout(ctx->ing_tiles, ntiles);
    uint64_t *fin = L.fin;
    uint64_t *seq_start = (uint64_t *)ctx->ing_rec;
    uint32_t *raw_flag = (uint32_t *)(seq_start + (max_records + 1));
    uint32_t *out32 = reinterpret_cast<uint32_t *>(d_reads_out);
    const uint64_t cap_words = reads_cap / 4;
    // fin and the descriptors start as zero ("nothing published yet")
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(fin, 0, F_WORDS * sizeof(uint64_t) + ntiles * sizeof(Desc), st));
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(seq_start, 0, sizeof(uint64_t), st));   // a file without any '>' is one record that starts at 0
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(raw_flag, 0, max_records * sizeof(uint32_t), st));
    const unsigned fgrid = (unsigned)ntiles;
    hipLaunchKernelGGL(ing_fused_kernel, dim3(fgrid), dim3(BLOCK), 0, st, d_text, nbytes, ntiles, L.desc, L.edges, fin, out32, cap_words,
                       seq_start, raw_flag, max_records);
    const unsigned egrid = (unsigned)((ntiles + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(ing_edges_kernel<true>, dim3(egrid), dim3(BLOCK), 0, st, L.edges, ntiles, out32, cap_words);
    hipLaunchKernelGGL(ing_edges_kernel<false>, dim3(egrid), dim3(BLOCK), 0, st, L.edges, ntiles, out32, cap_words);
    const int rgrid = (int)std::max<uint64_t>(1, std::min<uint64_t>((max_records + BLOCK - 1) / BLOCK, (uint64_t)ctx->num_cus * 8));
    hipLaunchKernelGGL(ing_records_kernel, dim3(rgrid), dim3(BLOCK), 0, st, seq_start, raw_flag, fin, max_records, reads_cap, d_frag_off,
                       d_frag_len);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    uint64_t host[F_WORDS];
    OPAL_HIP_CHECK(ctx, hipMemcpyAsync(host, fin, sizeof(host), hipMemcpyDeviceToHost, st));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (host[F_STALL]) return opal::fail(ctx, OPAL_EHIP, "opal_ingest_fasta: a tile waited for its predecessors in vain (scan chain stalled)");
    const uint64_t H = host[F_H], T = host[F_T];
    // too small a buffer: the counts the caller needs for a second call come back through the two outputs
    if (H > max_records) {
        *n_records_out = H;                // (which records are raw is not known for those beyond max_records: bytes unknown, 0)
        return opal::fail(ctx, OPAL_ERANGE, "opal_ingest_fasta: more records than max_records");
    }
    if (host[F_TOOLONG]) return opal::fail(ctx, OPAL_EINVAL, "opal_ingest_fasta: a record is longer than 2^31-1 bases");
    const uint64_t n_rec = (H ? H : (T ? 1 : 0)) - (host[F_DROPPED] ? 1 : 0);
    *n_records_out = n_rec;
    if (n_rec == 0) return OPAL_OK;
    const uint64_t out_bytes = host[F_NRAW] ? host[F_CURSOR] : ((T + 3) / 4 + 3) & ~3ull;
    if (out_bytes > reads_cap || host[F_OVERFLOW]) {
        *reads_bytes_out = out_bytes;      // (the cursor has run over every raw record: this is what the text needs)
        return opal::fail(ctx, OPAL_ERANGE, "opal_ingest_fasta: d_reads_out too small");
    }
    if (host[F_NRAW]) {
        hipLaunchKernelGGL(ing_rawcopy_kernel, dim3((unsigned)ntiles), dim3(BLOCK), 0, st, d_text, nbytes, L.desc, fin, seq_start, raw_flag,
                           max_records, d_reads_out, reads_cap, d_frag_off);
        OPAL_HIP_CHECK(ctx, hipGetLastError());
        OPAL_HIP_CHECK(ctx, hipStreamSynchronize(st));
    }
    *reads_bytes_out = out_bytes;
    return OPAL_OK;
}
