// GPU-side FASTA ingest: the text of a FASTA/FASTQ-ish file, resident in HBM, is turned
// into the fragment batch layout of include/opal_hip.h without the host ever touching a sequence byte.
// Semantics = fasta_functions.fasta_reader over open(path,'r') (reference util/fasta_functions.py:9-43, universal
// newlines): '>' at a line start opens a record; a line starting with '+' mutes everything up to the next '>';
// other lines are concatenated without their terminator; text before the first '>' is dropped (unless the file has
// no '>' at all: then it is one unnamed record); a record with an empty sequence is kept mid-file and dropped at
// end of file.  Record names are not produced (the predict path never uses them).
//
// Everything is a prefix scan over bytes with a small associative state:
//   state(i) = (# header lines started at or before i, last '>'/'+' event, type of the line byte i belongs to)
//   content(i) = byte i is sequence text  ->  its output position is the exclusive sum of content flags,
// then records are classified (all upper-case ACGT -> 2-bit packed, else raw) by a flat pass over the sequence bytes
// and written out: one wave per short record, the whole grid per long record (genomes).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "opal_device.h"
#include "opal_internal.h"

namespace {

constexpr int ING_ITEMS = 16;                  // bytes per thread
constexpr int ING_TILE = BLOCK * ING_ITEMS;    // bytes per block

struct St {
    uint64_t h;    // header-line starts seen
    uint32_t ev;   // last event: 0 none, 1 '>', 2 '+'
    uint32_t lt;   // type of the current line: 0 unknown (no line start seen), 1 '>', 2 '+', 3 data
};
__device__ __forceinline__ St st_combine(const St &a, const St &b)
{
    St r;
    r.h = a.h + b.h;
    r.ev = b.ev ? b.ev : a.ev;
    r.lt = b.lt ? b.lt : a.lt;
    return r;
}
__device__ __forceinline__ St st_shfl_up(const St &v, int d)
{
    St r;
    r.h = __shfl_up(v.h, d, WAVE);
    r.ev = __shfl_up(v.ev, d, WAVE);
    r.lt = __shfl_up(v.lt, d, WAVE);
    return r;
}
__device__ __forceinline__ bool is_term(uint32_t c) { return c == '\n' || c == '\r'; }

// The contribution of byte i: a line start sets the line type (and an event for '>' / '+').
__device__ __forceinline__ St st_of_byte(uint32_t c, uint32_t prev, bool first)
{
    St s{0, 0, 0};
    const bool line_start = first || prev == '\n' || (prev == '\r' && c != '\n');
    if (line_start) {
        if (c == '>') { s.h = 1; s.ev = 1; s.lt = 1; }
        else if (c == '+') { s.ev = 2; s.lt = 2; }
        else s.lt = 3;
    }
    return s;
}

// Block-wide exclusive scan of per-thread aggregates (256 threads); returns this thread's exclusive prefix and
// the block total through `total`.
__device__ __forceinline__ St block_excl_scan_st(St v, St *s_w, St &total)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    St incl = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const St o = st_shfl_up(incl, d);
        if (lane >= d) incl = st_combine(o, incl);
    }
    if (lane == WAVE - 1) s_w[wv] = incl;
    __syncthreads();
    St base{0, 0, 0};
    total = St{0, 0, 0};
#pragma unroll
    for (int w = 0; w < WPB; w++) {
        if (w < wv) base = st_combine(base, s_w[w]);
        total = st_combine(total, s_w[w]);
    }
    St excl_in_wave = st_shfl_up(incl, 1);
    if (lane == 0) excl_in_wave = St{0, 0, 0};
    __syncthreads();
    return st_combine(base, excl_in_wave);
}

__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t *s_w, uint64_t &total)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint64_t o = __shfl_up(incl, d, WAVE);
        if (lane >= d) incl += o;
    }
    if (lane == WAVE - 1) s_w[wv] = incl;
    __syncthreads();
    uint64_t base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < WPB; w++) {
        if (w < wv) base += s_w[w];
        total += s_w[w];
    }
    __syncthreads();
    return base + incl - v;
}

// the ING_ITEMS bytes of one thread: one 16-byte load when the text pointer allows it (bytes past the end read as '\n')
__device__ __forceinline__ void load_items(const uint8_t *__restrict__ text, uint64_t n, uint64_t base, uint8_t (&c)[ING_ITEMS])
{
    static_assert(ING_ITEMS == 16, "one uint4 per thread");
    if (base + ING_ITEMS <= n && (reinterpret_cast<uintptr_t>(text) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + base);
        __builtin_memcpy(c, &v, ING_ITEMS);
    } else {
#pragma unroll
        for (int i = 0; i < ING_ITEMS; i++) c[i] = base + i < n ? text[base + i] : (uint8_t)'\n';
    }
}

// pass 1: per-tile aggregate state
__global__ __launch_bounds__(BLOCK) void ing_tile_state_kernel(const uint8_t *__restrict__ text, uint64_t n, St *__restrict__ tile_st)
{
    __shared__ St s_w[WPB];
    const uint64_t base = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)threadIdx.x * ING_ITEMS;
    St agg{0, 0, 0};
    uint32_t prev = (base > 0 && base <= n) ? text[base - 1] : 0;
    uint8_t c[ING_ITEMS];
    load_items(text, n, base, c);
#pragma unroll
    for (int i = 0; i < ING_ITEMS; i++) {
        const uint64_t p = base + i;
        if (p < n) {
            agg = st_combine(agg, st_of_byte(c[i], prev, p == 0));
            prev = c[i];
        }
    }
    St total;
    (void)block_excl_scan_st(agg, s_w, total);
    if (threadIdx.x == 0) tile_st[blockIdx.x] = total;
}

// pass 2: exclusive scan of the tile aggregates by one block (SC_ITEMS consecutive tiles per thread and round);
// grand total to *total_out
constexpr int SC_ITEMS = 8;
__global__ __launch_bounds__(BLOCK) void ing_scan_tiles_st_kernel(St *__restrict__ tile_st, uint64_t ntiles, St *__restrict__ total_out)
{
    __shared__ St s_w[WPB];
    __shared__ St s_carry;
    if (threadIdx.x == 0) s_carry = St{0, 0, 0};
    __syncthreads();
    for (uint64_t b0 = 0; b0 < ntiles; b0 += (uint64_t)BLOCK * SC_ITEMS) {
        const uint64_t i0 = b0 + (uint64_t)threadIdx.x * SC_ITEMS;
        St v[SC_ITEMS], agg{0, 0, 0};
#pragma unroll
        for (int j = 0; j < SC_ITEMS; j++) {
            v[j] = i0 + j < ntiles ? tile_st[i0 + j] : St{0, 0, 0};
            agg = st_combine(agg, v[j]);
        }
        St total;
        const St excl = block_excl_scan_st(agg, s_w, total);
        const St carry = s_carry;
        St run = st_combine(carry, excl);
#pragma unroll
        for (int j = 0; j < SC_ITEMS; j++) {
            if (i0 + j < ntiles) tile_st[i0 + j] = run;
            run = st_combine(run, v[j]);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = st_combine(carry, total);
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = s_carry;
}

__global__ __launch_bounds__(BLOCK) void ing_scan_tiles_u64_kernel(uint64_t *__restrict__ tile_v, uint64_t ntiles, uint64_t *__restrict__ total_out)
{
    __shared__ uint64_t s_w[WPB];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < ntiles; b0 += (uint64_t)BLOCK * SC_ITEMS) {
        const uint64_t i0 = b0 + (uint64_t)threadIdx.x * SC_ITEMS;
        uint64_t v[SC_ITEMS], agg = 0;
#pragma unroll
        for (int j = 0; j < SC_ITEMS; j++) {
            v[j] = i0 + j < ntiles ? tile_v[i0 + j] : 0ull;
            agg += v[j];
        }
        uint64_t total;
        const uint64_t excl = block_excl_scan_u64(agg, s_w, total);
        const uint64_t carry = s_carry;
        uint64_t run = carry + excl;
#pragma unroll
        for (int j = 0; j < SC_ITEMS; j++) {
            if (i0 + j < ntiles) tile_v[i0 + j] = run;
            run += v[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = s_carry;
}

// passes 3 and 5 share this body.  WRITE=false: count the content bytes of each tile.
// WRITE=true: compact content bytes to seq[pos] and record where each header's sequence starts.
template <bool WRITE>
__global__ __launch_bounds__(BLOCK) void ing_content_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                             const St *__restrict__ tile_st, const St *__restrict__ total_st,
                                                             uint64_t *__restrict__ tile_cnt, uint8_t *__restrict__ seq,
                                                             uint64_t *__restrict__ seq_start, uint64_t max_records)
{
    __shared__ St s_w[WPB];
    __shared__ uint64_t s_c[WPB];
    const uint64_t base = (uint64_t)blockIdx.x * ING_TILE + (uint64_t)threadIdx.x * ING_ITEMS;
    const bool any_header = total_st->h > 0;
    uint8_t c[ING_ITEMS];
    St agg{0, 0, 0};
    uint32_t prev0 = (base > 0 && base <= n) ? text[base - 1] : 0;
    uint32_t prev = prev0;
    load_items(text, n, base, c);
#pragma unroll
    for (int i = 0; i < ING_ITEMS; i++) {
        const uint64_t p = base + i;
        if (p < n) {
            agg = st_combine(agg, st_of_byte(c[i], prev, p == 0));
            prev = c[i];
        }
    }
    St total;
    St run = st_combine(tile_st[blockIdx.x], block_excl_scan_st(agg, s_w, total));
    // per-byte state + content flags
    uint32_t flags = 0, hdr = 0;
    uint64_t hcount[ING_ITEMS];
    prev = prev0;
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < ING_ITEMS; i++) {
        const uint64_t p = base + i;
        if (p < n) {
            const St b = st_of_byte(c[i], prev, p == 0);
            run = st_combine(run, b);
            prev = c[i];
            // lt==0 only happens before any line start, i.e. never (byte 0 is a line start)
            const bool content = !is_term(c[i]) && run.lt == 3 && run.ev != 2 && (run.h > 0 || !any_header);
            if (content) { flags |= 1u << i; cnt++; }
            if (b.h) hdr |= 1u << i;
        }
        hcount[i] = run.h;
    }
    uint64_t tot;
    const uint64_t excl = block_excl_scan_u64(cnt, s_c, tot);
    if (!WRITE) {
        if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
        return;
    }
    uint64_t pos = tile_cnt[blockIdx.x] + excl;
#pragma unroll
    for (int i = 0; i < ING_ITEMS; i++) {
        if (hdr & (1u << i)) {
            const uint64_t r = hcount[i] - 1;                 // record index of this header
            if (r < max_records) seq_start[r] = pos;
        }
        if (flags & (1u << i)) seq[pos++] = c[i];
    }
}

// Classification is flat over the compacted sequence bytes, so one 4 Mbp genome occupies the whole chip just as
// 4M short reads do: a thread looks at 16 bytes; only a byte outside ACGT (rare) costs a binary search for its
// record and a store of 1 into that record's flag.
constexpr int CL_ITEMS = 16;
__device__ __forceinline__ uint64_t record_of(const uint64_t *__restrict__ seq_start, uint64_t n_rec, uint64_t pos)
{
    // last r with seq_start[r] <= pos (empty records share a start with their successor: the last one holds the byte)
    uint64_t lo = 0, hi = n_rec - 1;
    while (lo < hi) {
        const uint64_t mid = (lo + hi + 1) >> 1;
        if (seq_start[mid] <= pos) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(BLOCK) void ing_classify_kernel(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ seq_start,
                                                              uint64_t n_rec, uint64_t total_content, uint32_t *__restrict__ raw_flag)
{
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK * CL_ITEMS;
    for (uint64_t base = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * CL_ITEMS; base < total_content; base += stride) {
        uint8_t c[CL_ITEMS];
        if (base + CL_ITEMS <= total_content) {
            const uint4 v = *reinterpret_cast<const uint4 *>(seq + base);      // seq comes from hipMalloc, base % 16 == 0
            __builtin_memcpy(c, &v, CL_ITEMS);
        } else {
#pragma unroll
            for (int i = 0; i < CL_ITEMS; i++) c[i] = base + i < total_content ? seq[base + i] : (uint8_t)'A';
        }
        uint32_t bad = 0;
#pragma unroll
        for (int i = 0; i < CL_ITEMS; i++) {
            const uint32_t ch = c[i];
            if (!(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T')) bad |= 1u << i;
        }
        while (bad) {
            const int i = __builtin_ctz(bad);
            bad &= bad - 1;
            raw_flag[record_of(seq_start, n_rec, base + i)] = 1u;   // same value from every writer
        }
    }
}

// per record: length word, output bytes; records longer than LONG_BASES go on the list of ing_pack_long_kernel
constexpr uint64_t LONG_BASES = 1u << 16;
constexpr uint32_t LONG_CAP = 1u << 16;
__global__ __launch_bounds__(BLOCK) void ing_lens_kernel(const uint64_t *__restrict__ seq_start, uint64_t n_rec, uint64_t total_content,
                                                          const uint32_t *__restrict__ raw_flag, uint32_t *__restrict__ frag_len,
                                                          uint64_t *__restrict__ rec_bytes, uint64_t *__restrict__ too_long,
                                                          unsigned int *__restrict__ long_count, uint64_t *__restrict__ long_list)
{
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; r < n_rec; r += stride) {
        const uint64_t s = seq_start[r];
        const uint64_t e = (r + 1 < n_rec) ? seq_start[r + 1] : total_content;
        const uint64_t len = e - s;
        const bool raw = raw_flag[r] != 0;
        if (len > 0x7fffffffull) *too_long = 1;           // reported as OPAL_EINVAL by the host; plain store of a constant
        const uint32_t l31 = len > 0x7fffffffull ? 0x7fffffffu : (uint32_t)len;
        frag_len[r] = l31 | (raw ? OPAL_RAW_FLAG : 0u);
        rec_bytes[r] = raw ? len : (len + 3) / 4;
        if (len > LONG_BASES) {
            const unsigned int at = atomicAdd(long_count, 1u);
            if (at < LONG_CAP) long_list[at] = r;
        }
    }
}

// exclusive scan over records of rec_bytes (tile = 256*8 records), 3 launches like the offsets scan
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = BLOCK * RS_ITEMS;
template <bool WRITE>
__global__ __launch_bounds__(BLOCK) void ing_rec_scan_kernel(const uint64_t *__restrict__ rec_bytes, const uint32_t *__restrict__ frag_len,
                                                              uint64_t n_rec, uint64_t *__restrict__ tile_v,
                                                              uint64_t *__restrict__ out_byte_off, uint64_t *__restrict__ frag_off)
{
    __shared__ uint64_t s_w[WPB];
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE + (uint64_t)threadIdx.x * RS_ITEMS;
    uint64_t v[RS_ITEMS], sum = 0;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        v[i] = base + i < n_rec ? rec_bytes[base + i] : 0ull;
        sum += v[i];
    }
    uint64_t tot;
    const uint64_t excl = block_excl_scan_u64(sum, s_w, tot);
    if (!WRITE) {
        if (threadIdx.x == 0) tile_v[blockIdx.x] = tot;
        return;
    }
    uint64_t run = tile_v[blockIdx.x] + excl;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        if (base + i < n_rec) {
            out_byte_off[base + i] = run;
            const bool raw = (frag_len[base + i] & OPAL_RAW_FLAG) != 0;
            frag_off[base + i] = raw ? run : run * 4;          // packed fragments are addressed in bases
        }
        run += v[i];
    }
}

// output bytes [b0, b1) of one record, strided over by `nthreads` threads of which this is number `tid`
__device__ __forceinline__ void pack_span(const uint8_t *__restrict__ seq, uint64_t s, uint64_t len, bool raw, uint8_t *__restrict__ dst,
                                          uint64_t b0, uint64_t b1, uint64_t tid, uint64_t nthreads)
{
    if (raw) {
        for (uint64_t i = b0 + tid; i < b1; i += nthreads) dst[i] = seq[s + i];
        return;
    }
    // 4 letters per output byte: two aligned dwords + v_alignbyte instead of four byte loads (seq has 16 bytes of
    // slack behind its last byte), then all four codes at once
    const uint32_t *__restrict__ seq4 = reinterpret_cast<const uint32_t *>(seq);
    for (uint64_t b = b0 + tid; b < b1; b += nthreads) {
        const uint64_t a = s + 4 * b;
        const uint32_t x = __builtin_amdgcn_alignbyte(seq4[(a >> 2) + 1], seq4[a >> 2], (uint32_t)(a & 3));
        // A=0x41 C=0x43 G=0x47 T=0x54: (ch >> 1) & 3 gives 0,1,3,2 -> fix the G/T order
        uint32_t q = (x >> 1) & 0x03030303u;
        q ^= (q >> 1) & 0x01010101u;
        const uint64_t left = len - 4 * b;                                    // letters of the record from here on (>= 1)
        if (left < 4) q &= (1u << (8 * (uint32_t)left)) - 1u;
        dst[b] = (uint8_t)((q | (q >> 6) | (q >> 12) | (q >> 18)) & 0xffu);
    }
}

// one wave per record: write the packed (4 bases/byte) or raw copy.  Records on the long list are left to
// ing_pack_long_kernel (unless the list overflowed: then every record is handled here).
__global__ __launch_bounds__(BLOCK) void ing_pack_kernel(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ seq_start,
                                                          const uint32_t *__restrict__ frag_len, const uint64_t *__restrict__ out_byte_off,
                                                          uint64_t n_rec, uint8_t *__restrict__ out, uint64_t out_cap,
                                                          const unsigned int *__restrict__ long_count)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const uint64_t nwaves = (uint64_t)gridDim.x * WPB;
    const bool split_long = *long_count <= LONG_CAP;
    for (uint64_t r = (uint64_t)blockIdx.x * WPB + wv; r < n_rec; r += nwaves) {
        const uint32_t lw = frag_len[r];
        const uint64_t len = lw & ~OPAL_RAW_FLAG;
        if (split_long && len > LONG_BASES) continue;
        const bool raw = (lw & OPAL_RAW_FLAG) != 0;
        const uint64_t o = out_byte_off[r], nb = raw ? len : (len + 3) / 4;
        if (o + nb > out_cap) continue;
        pack_span(seq, seq_start[r], len, raw, out + o, 0, nb, (uint64_t)lane, WAVE);
    }
}

// the long records (genomes), one after the other, each spread over the whole grid
__global__ __launch_bounds__(BLOCK) void ing_pack_long_kernel(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ seq_start,
                                                               const uint32_t *__restrict__ frag_len, const uint64_t *__restrict__ out_byte_off,
                                                               uint8_t *__restrict__ out, uint64_t out_cap,
                                                               const unsigned int *__restrict__ long_count, const uint64_t *__restrict__ long_list)
{
    const unsigned int n_long = *long_count;
    if (n_long > LONG_CAP) return;
    const uint64_t tid = (uint64_t)blockIdx.x * BLOCK + threadIdx.x, nthreads = (uint64_t)gridDim.x * BLOCK;
    for (unsigned int li = 0; li < n_long; li++) {
        const uint64_t r = long_list[li];
        const uint32_t lw = frag_len[r];
        const uint64_t len = lw & ~OPAL_RAW_FLAG;
        const bool raw = (lw & OPAL_RAW_FLAG) != 0;
        const uint64_t o = out_byte_off[r], nb = raw ? len : (len + 3) / 4;
        if (o + nb > out_cap) continue;
        pack_span(seq, seq_start[r], len, raw, out + o, 0, nb, tid, nthreads);
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
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t ntiles = (nbytes + ING_TILE - 1) / ING_TILE;
    // scratch: [tile states | tile counts | totals] , compacted text, record starts, record bytes/offsets
    int rc = grow(ctx, &ctx->ing_tiles, &ctx->ing_tiles_cap, ntiles * (sizeof(St) + sizeof(uint64_t)) + 256);
    if (rc) return rc;
    rc = grow(ctx, &ctx->ing_seq, &ctx->ing_seq_cap, nbytes + 16);   // slack: the packer reads whole dwords
    if (rc) return rc;
    const uint64_t rec_tiles = (max_records + RS_TILE - 1) / RS_TILE;
    rc = grow(ctx, &ctx->ing_rec, &ctx->ing_rec_cap, (max_records + 1) * (3 * sizeof(uint64_t) + sizeof(uint32_t)) +
                                                       (rec_tiles + LONG_CAP) * sizeof(uint64_t) + 256);
    if (rc) return rc;
    St *tile_st = (St *)ctx->ing_tiles;
    uint64_t *tile_cnt = (uint64_t *)(tile_st + ntiles);
    St *total_st = (St *)(tile_cnt + ntiles);                    // 16 bytes
    uint64_t *total_cnt = (uint64_t *)(total_st + 1);
    uint64_t *total_out = total_cnt + 1;
    uint64_t *too_long = total_out + 1;
    uint8_t *seq = (uint8_t *)ctx->ing_seq;
    uint64_t *seq_start = (uint64_t *)ctx->ing_rec;
    uint64_t *rec_bytes = seq_start + (max_records + 1);
    uint64_t *out_byte_off = rec_bytes + (max_records + 1);
    uint64_t *rec_tile_v = out_byte_off + (max_records + 1);
    uint64_t *long_list = rec_tile_v + rec_tiles;
    unsigned int *long_count = (unsigned int *)(long_list + LONG_CAP);      // 2 x u32 (8-byte slot)
    uint32_t *raw_flag = (uint32_t *)(long_count + 2);

    hipLaunchKernelGGL(ing_tile_state_kernel, dim3((unsigned)ntiles), dim3(BLOCK), 0, st, d_text, nbytes, tile_st);
    hipLaunchKernelGGL(ing_scan_tiles_st_kernel, dim3(1), dim3(BLOCK), 0, st, tile_st, ntiles, total_st);
    hipLaunchKernelGGL(ing_content_kernel<false>, dim3((unsigned)ntiles), dim3(BLOCK), 0, st, d_text, nbytes, tile_st, total_st, tile_cnt,
                       seq, seq_start, max_records);
    hipLaunchKernelGGL(ing_scan_tiles_u64_kernel, dim3(1), dim3(BLOCK), 0, st, tile_cnt, ntiles, total_cnt);
    // a file without any '>' is one record that starts at 0
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(seq_start, 0, sizeof(uint64_t), st));
    hipLaunchKernelGGL(ing_content_kernel<true>, dim3((unsigned)ntiles), dim3(BLOCK), 0, st, d_text, nbytes, tile_st, total_st, tile_cnt,
                       seq, seq_start, max_records);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    struct { St s; uint64_t content; } host;
    OPAL_HIP_CHECK(ctx, hipMemcpyAsync(&host, total_st, sizeof(host), hipMemcpyDeviceToHost, st));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(st));
    const uint64_t T = host.content;
    uint64_t n_rec = host.s.h ? host.s.h : (T ? 1 : 0);
    if (n_rec > max_records) return opal::fail(ctx, OPAL_ERANGE, "opal_ingest_fasta: more records than max_records");
    if (n_rec && host.s.h) {
        // the last record is dropped when its sequence is empty (fasta_functions.py:20-22)
        uint64_t last_start = 0;
        OPAL_HIP_CHECK(ctx, hipMemcpyAsync(&last_start, seq_start + (n_rec - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        OPAL_HIP_CHECK(ctx, hipStreamSynchronize(st));
        if (last_start == T) n_rec--;
    }
    *n_records_out = n_rec;
    if (n_rec == 0) return OPAL_OK;
    const int wgrid = (int)std::max<uint64_t>(1, std::min<uint64_t>((n_rec + WPB - 1) / WPB, (uint64_t)ctx->num_cus * 8));
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(too_long, 0, sizeof(uint64_t), st));
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(long_count, 0, 2 * sizeof(unsigned int) + n_rec * sizeof(uint32_t), st));   // count + flags
    const int fgrid = ctx->num_cus * 8;
    if (T) hipLaunchKernelGGL(ing_classify_kernel, dim3(fgrid), dim3(BLOCK), 0, st, seq, seq_start, n_rec, T, raw_flag);
    hipLaunchKernelGGL(ing_lens_kernel, dim3((unsigned)std::min<uint64_t>((n_rec + BLOCK - 1) / BLOCK, (uint64_t)fgrid)), dim3(BLOCK), 0, st,
                       seq_start, n_rec, T, raw_flag, d_frag_len, rec_bytes, too_long, long_count, long_list);
    const uint64_t rt = (n_rec + RS_TILE - 1) / RS_TILE;
    hipLaunchKernelGGL(ing_rec_scan_kernel<false>, dim3((unsigned)rt), dim3(BLOCK), 0, st, rec_bytes, d_frag_len, n_rec, rec_tile_v, out_byte_off, d_frag_off);
    hipLaunchKernelGGL(ing_scan_tiles_u64_kernel, dim3(1), dim3(BLOCK), 0, st, rec_tile_v, rt, total_out);
    hipLaunchKernelGGL(ing_rec_scan_kernel<true>, dim3((unsigned)rt), dim3(BLOCK), 0, st, rec_bytes, d_frag_len, n_rec, rec_tile_v, out_byte_off, d_frag_off);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    uint64_t tail[2] = {0, 0};   // {packed bytes, any record longer than 2^31-1}
    OPAL_HIP_CHECK(ctx, hipMemcpyAsync(tail, total_out, sizeof(tail), hipMemcpyDeviceToHost, st));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(st));
    const uint64_t out_bytes = tail[0];
    if (tail[1]) return opal::fail(ctx, OPAL_EINVAL, "opal_ingest_fasta: a record is longer than 2^31-1 bases");
    if (out_bytes > reads_cap) return opal::fail(ctx, OPAL_ERANGE, "opal_ingest_fasta: d_reads_out too small");
    hipLaunchKernelGGL(ing_pack_kernel, dim3(wgrid), dim3(BLOCK), 0, st, seq, seq_start, d_frag_len, out_byte_off, n_rec, d_reads_out, reads_cap,
                       long_count);
    hipLaunchKernelGGL(ing_pack_long_kernel, dim3(fgrid), dim3(BLOCK), 0, st, seq, seq_start, d_frag_len, out_byte_off, d_reads_out, reads_cap,
                       long_count, long_list);
    OPAL_HIP_CHECK(ctx, hipGetLastError());
    *reads_bytes_out = out_bytes;
    return OPAL_OK;
}
