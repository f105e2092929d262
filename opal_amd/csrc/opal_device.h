// Device code shared by every kernel of libopal_hip.so: fragment staging, MurmurHash3, the fragment walker
// (phase H) and the pipelined weight-row stream (phase S).  Included by opal_kernels.hip and opal_train.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opal_hip.h"

namespace {

constexpr int WAVE = 64;
constexpr int WPB = 4;          // waves per block
constexpr int BLOCK = WAVE * WPB;
constexpr int SEQB = 512;       // bytes of one strand tile held in LDS per wave
constexpr int HB = 1536;        // hash-buffer entries (u32) per wave
constexpr int HB_PAD = 16;      // dummy entries after the live ones (prefetch overrun target)

struct Batch {
    const uint8_t *reads;
    uint64_t reads_bytes;
    const uint64_t *off;
    const uint32_t *len;
    uint64_t n;
};
struct Geom {
    const uint8_t *pos;  // P rows of ts bytes: t positions, padded to a multiple of 4 so a row reads as dwords
    int k, P, t, ts;
    int reverse;
    int hash_all;
    int p0_run;   // >= 0: pattern 0 is the ascending run p0_run, p0_run+1, ... (Opal's first row is 0..t-1); -1 otherwise
    int fast_canon2 = 0;        // emit fast path: P == 2, pattern 0 a run of 16 positions, pattern 1 gathered
    int fast_maxlen = 512;      // emit fast path: longest fragment (bases) the launched fast kernel stages (EF_MAXLEN unless an experiment lowers it)
    const uint32_t *fast = nullptr;   // emit fast path (t == 16, positions ascending inside every block of 4): FASTPAT_DW dwords per pattern, else nullptr
};

// Per-pattern record of the emit fast path (built by opal_set_patterns), FASTPAT_DW dwords:
//   [0..15] 4 * (pos_j - (j & 3)): the byte offset, inside a window's slice of the per-base array Z of emit_fast, of the
//           word that holds the code of pos_j at bits 2+2(j&3)
//   [16]    murmur term of the decimal pattern index xor the name length (wave-uniform tail of every name of the pattern)
//   [17]    0xf when the 16 positions are one run of consecutive positions, else 0
//   [18]    4 * (pos_0 + 1): for a run, the offset of the one word that holds all 16 codes
constexpr int FASTPAT_DW = 52;
constexpr int FASTPAT_PAIR = 20;   // [20..48]: the row's paired form (PairRec in opal_kernels.hip, built by opal_set_patterns)

// ------------------------------------------------------------------------------------------ byte helpers
__device__ __forceinline__ uint32_t rc_byte(uint32_t c)
{
    // str.translate(maketrans('ATGCatgc','TACGTACG')): util/fasta_functions.py:46-50
    const uint32_t u = c | 0x20u;
    uint32_t r = c;
    r = (u == 'a') ? 'T' : r;
    r = (u == 't') ? 'A' : r;
    r = (u == 'g') ? 'C' : r;
    r = (u == 'c') ? 'G' : r;
    return r;
}

// Loads bytes [i0, i0+cnt) of strand `strand` of one fragment into LDS as the ASCII text the reference sees.
// `shift` (0..3) = position of the fragment's first base inside its first packed byte (packed fragments may start
// at any base of a packed genome that lives in the same buffer).
// `codes` (packed fragments only): store the 2-bit code of each base (complemented on the reverse strand) instead
// of its letter, for the table-driven hash of hash_name_tab.
__device__ __forceinline__ void load_tile(uint8_t *sq, const uint8_t *src, uint32_t len, bool raw, uint32_t shift,
                                          int strand, uint32_t i0, uint32_t cnt, int lane, bool codes = false)
{
    for (uint32_t i = lane; i < cnt; i += WAVE) {
        const uint32_t p = i0 + i;
        const uint32_t sp = strand ? (len - 1u - p) : p;
        uint32_t c;
        if (raw) {
            c = src[sp];
        } else {
            const uint32_t q = sp + shift;
            const uint32_t code = (src[q >> 2] >> ((q & 3u) * 2u)) & 3u;
            if (codes) { sq[i] = (uint8_t)(strand ? 3u - code : code); continue; }
            c = (0x54474341u >> (code * 8u)) & 0xffu;  // "ACGT"
        }
        if (strand) c = rc_byte(c);
        sq[i] = (uint8_t)c;
    }
}

// Fast staging of a 2-bit packed fragment: each lane expands 4 consecutive bases (one output dword) with two byte
// loads, a funnel shift and v_perm_b32 lookups into "ACGT" (forward) or "TGCA" (complement, bases taken in
// descending order for the reverse strand).  abs0 = absolute base offset of the fragment inside `reads`.
// Bytes of sq at positions >= len receive don't-care values (no window reads them).
__device__ __forceinline__ void load_strand_packed(uint8_t *sq, const uint8_t *__restrict__ reads, uint64_t reads_bytes,
                                                   uint64_t abs0, uint32_t len, int strand, int lane, bool codes)
{
    uint32_t *sq4 = reinterpret_cast<uint32_t *>(sq);
    const int64_t last_byte = (int64_t)reads_bytes - 1;
    for (uint32_t i = lane; i < (len + 3u) / 4u; i += WAVE) {
        // forward: bases 4i..4i+3 ascending; reverse: output chars 4i..4i+3 = comp of bases len-1-4i .. len-4-4i
        const int64_t a = strand ? (int64_t)abs0 + (int64_t)len - 4 - 4 * (int64_t)i : (int64_t)abs0 + 4 * (int64_t)i;
        const int64_t bi = a >> 2;
        const uint32_t sh = (uint32_t)(a & 3) * 2u;
        const uint32_t lo = reads[min(max(bi, (int64_t)0), last_byte)];
        const uint32_t hi = reads[min(max(bi + 1, (int64_t)0), last_byte)];
        const uint32_t x = ((lo | (hi << 8)) >> sh) & 0xffu;         // 4 codes, base a in bits 1:0
        const uint32_t y = (x & 0x33u) * 0x1001u & 0x00030003u;       // codes 0,2 -> bytes 0,2
        const uint32_t z = ((x >> 2) & 0x33u) * 0x1001u & 0x00030003u; // codes 1,3 -> bytes 0,2
        uint32_t sel = y | (z << 8);                                   // selector byte j = code of base a+j
        uint32_t chars;
        if (strand) {
            sel = __builtin_amdgcn_perm(0u, sel, 0x00010203u);         // reverse the 4 bases
            chars = codes ? (sel ^ 0x03030303u) : __builtin_amdgcn_perm(0u, 0x41434754u, sel);   // "TGCA"[code] = complement
        } else {
            chars = codes ? sel : __builtin_amdgcn_perm(0u, 0x54474341u, sel);                    // "ACGT"[code]
        }
        sq4[i] = chars;
    }
}

// Resolves fragment `frag` of a batch: raw fragments are addressed in BYTES, packed ones in BASES (so that a packed
// fragment can alias any window of a packed genome).  A fragment that would reach outside the buffer gets len 0.
struct FragRef {
    const uint8_t *src;
    uint32_t len, shift;
    bool raw;
};
__device__ __forceinline__ FragRef locate_fragment(const Batch &B, uint64_t frag)
{
    FragRef r;
    const uint32_t lw = B.len[frag];
    r.raw = (lw & OPAL_RAW_FLAG) != 0;
    r.len = lw & ~OPAL_RAW_FLAG;
    const uint64_t off = B.off[frag];
    uint64_t first, last;  // byte range [first, last)
    if (r.raw) { first = off; last = off + r.len; r.shift = 0; }
    else { first = off >> 2; last = (off + r.len + 3) >> 2; r.shift = (uint32_t)(off & 3); }
    if (last > B.reads_bytes || first > last) { r.len = 0; first = 0; }
    r.src = B.reads + first;
    return r;
}

// ------------------------------------------------------------------------------------------ MurmurHash3_x86_32
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_rotateleft32(x, r); }
__device__ __forceinline__ uint32_t mm_k(uint32_t k1)
{
    k1 *= 0xcc9e2d51u;
    k1 = rotl32(k1, 15);
    k1 *= 0x1b873593u;
    return k1;
}
// block step with the block's term mm_k(k1) already known
__device__ __forceinline__ uint32_t mm_hk(uint32_t h, uint32_t km)
{
    h ^= km;
    h = rotl32(h, 13);
    uint32_t h5;
    asm("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(h5) : "v"(h));
    return h5 + 0xe6546b64u;
}
__device__ __forceinline__ uint32_t mm_h(uint32_t h, uint32_t k1)
{
    h ^= mm_k(k1);
    h = rotl32(h, 13);
    // h*5 + c as shift-add: keeps the quarter-rate integer multiplier for the two products that need it
    uint32_t h5;
    asm("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(h5) : "v"(h));
    return h5 + 0xe6546b64u;
}
__device__ __forceinline__ uint32_t mm_fin(uint32_t h, uint32_t len)
{
    h ^= len;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// The murmur tail of a feature name: the decimal pattern index p < 100 behind T gathered characters (T % 4 == 0, so the
// digits start a fresh block): its term and the name's length, both wave-uniform.
__device__ __forceinline__ uint32_t name_tail(int p, int T, uint32_t &len)
{
    if (p < 10) { len = (uint32_t)T + 1u; return mm_k((uint32_t)('0' + p)); }
    len = (uint32_t)T + 2u;
    return mm_k((uint32_t)('0' + p / 10) | ((uint32_t)('0' + p % 10) << 8));
}

// Hash of the feature name "<t gathered chars><decimal p>" (util/fasta2skm.py:140) for the window whose first
// byte in LDS is `wb`.
// Table-driven variant for 2-bit packed fragments (letters are only ACGT): a block of 4 gathered letters has 256
// possible values, so its murmur term mm_k(4 letters) = two quarter-rate 32-bit multiplies + a rotate comes from a
// 256-entry LDS table (one per block) indexed by the 4 gathered 2-bit codes (the emit kernel is VALU-bound: PMC
// shows every SIMD issue cycle taken, profiles/r01_cfg5_emit).  wb points at staged CODES (load_* with codes=true).
constexpr int KTAB = 512;          // [0,256): mm_k(4 letters); [256,512): the hash after the FIRST block (h = 0 going in)
__device__ __forceinline__ void fill_kmer4_table(uint32_t *tab)   // one table per block; every thread calls it once
{
    for (uint32_t i = threadIdx.x; i < 256u; i += BLOCK) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) word |= ((0x54474341u >> (((i >> (2 * j)) & 3u) * 8u)) & 0xffu) << (8 * j);   // "ACGT"[code j]
        const uint32_t km = mm_k(word);
        tab[i] = km;
        tab[256u + i] = mm_hk(0u, km);
    }
    __syncthreads();
}
// murmur finaliser for a hash whose tail term and length are already folded in
__device__ __forceinline__ uint32_t mm_fmix(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
// pq[b] = the 4 positions of block b, one per byte (the pattern row read as dwords; wave-uniform, kept in SGPRs)
template <int T>
__device__ __forceinline__ uint32_t hash_name_tab(const uint8_t *wb, const uint32_t (&pq)[T / 4], int p, const uint32_t *tab)
{
    static_assert(T % 4 == 0, "fast path needs whole blocks");
    uint32_t h = 0;
#pragma unroll
    for (int b = 0; b < T / 4; b++) {
        const uint32_t q = pq[b];
        const uint32_t idx = (uint32_t)wb[q & 0xffu] | ((uint32_t)wb[(q >> 8) & 0xffu] << 2) |
                             ((uint32_t)wb[(q >> 16) & 0xffu] << 4) | ((uint32_t)wb[q >> 24] << 6);
        h = b == 0 ? tab[256u + idx] : mm_hk(h, tab[idx]);
    }
    uint32_t len;
    const uint32_t tail = name_tail(p, T, len);
    return mm_fmix(h ^ (tail ^ len));                  // tail term and length: one wave-uniform word
}

// Raw (text) fragments with t = 16: murmur over the gathered letters in registers, as hash_name_generic computes
// it, plus a 4-bytes-at-a-time test whether the name is all digits (the `--hash strings` rule then applies and the
// caller falls back to hash_name_generic; it never happens for DNA text).
template <int T>
__device__ __forceinline__ uint32_t hash_name_ascii(const uint8_t *wb, const uint32_t (&pq)[T / 4], int p, bool &all_digits)
{
    uint32_t h = 0;
    bool dig = true;
#pragma unroll
    for (int b = 0; b < T / 4; b++) {
        const uint32_t q = pq[b];
        const uint32_t k1 = (uint32_t)wb[q & 0xffu] | ((uint32_t)wb[(q >> 8) & 0xffu] << 8) |
                            ((uint32_t)wb[(q >> 16) & 0xffu] << 16) | ((uint32_t)wb[q >> 24] << 24);
        // '0'..'9' in every byte: high nibble 3, low nibble + 6 does not carry into bit 4
        dig = dig && (k1 & 0xf0f0f0f0u) == 0x30303030u && (((k1 & 0x0f0f0f0fu) + 0x06060606u) & 0x10101010u) == 0u;
        h = mm_h(h, k1);
    }
    uint32_t len;
    h ^= name_tail(p, T, len);
    all_digits = dig;
    return mm_fin(h, len);
}

// A pattern made of consecutive positions needs no gather: cw[i] holds the codes of bases i..i+3 packed into one
// byte (= the table index of the block that starts at base i), so a block costs one ds_read_u8 with an immediate
// offset and one table read.  wbw = cw + window base + first position of the run.
template <int T>
__device__ __forceinline__ uint32_t hash_name_run(const uint8_t *wbw, int p, const uint32_t *tab)
{
    uint32_t h = tab[256u + wbw[0]];
#pragma unroll
    for (int b = 1; b < T / 4; b++) h = mm_hk(h, tab[wbw[4 * b]]);
    return mm_fmix(h ^ (mm_k((uint32_t)('0' + p)) ^ (uint32_t)(T + 1)));
}
// cw[i] = c[i] | c[i+1]<<2 | c[i+2]<<4 | c[i+3]<<6 for the staged code bytes c[0..len) of one strand (both arrays
// SEQB/2 bytes, 4-byte aligned); entries whose block would pass the end of the strand are don't-care.
__device__ __forceinline__ void pack_code_windows(uint8_t *cw, const uint8_t *c, uint32_t len, int lane)
{
    const uint32_t *c4 = reinterpret_cast<const uint32_t *>(c);
    uint32_t *w4 = reinterpret_cast<uint32_t *>(cw);
    constexpr uint32_t LAST = SEQB / 8 - 1;
    auto squeeze = [](uint32_t d) { return (d | (d >> 6) | (d >> 12) | (d >> 18)) & 0xffu; };   // 4 code bytes -> 8 bits
    for (uint32_t m = lane; m < (len + 3u) / 4u; m += WAVE) {
        const uint32_t x = squeeze(c4[m]) | (squeeze(c4[min(m + 1u, LAST)]) << 8);             // codes 4m .. 4m+7
        w4[m] = (x & 0xffu) | (((x >> 2) & 0xffu) << 8) | (((x >> 4) & 0xffu) << 16) | (((x >> 6) & 0xffu) << 24);
    }
}

// Generic: any t, any p < 1000, optional VW `--hash strings` digit rule (all-digit names are numbers).
__device__ __forceinline__ uint32_t hash_name_generic(const uint8_t *wb, const uint8_t *__restrict__ pp, int t, int p,
                                                      bool digit_rule, uint32_t mask)
{
    uint32_t h = 0, k1 = 0;
    uint32_t n = 0;
    bool digits = true;
    uint64_t num = 0;
    auto feed = [&](uint32_t c) {
        k1 |= c << (8u * (n & 3u));
        n++;
        if ((n & 3u) == 0) { h = mm_h(h, k1); k1 = 0; }
        const uint32_t d = c - (uint32_t)'0';
        digits = digits && (d <= 9u);
        num = num * 10ull + d;
    };
    // the row as dwords: wave-uniform, so the position bytes come from the scalar unit (a byte load of the same
    // address is a vector load + vmcnt(0) in front of every character)
    const uint32_t *__restrict__ pw = reinterpret_cast<const uint32_t *>(pp);
    uint32_t q = 0;
    for (int j = 0; j < t; j++) {
        if ((j & 3) == 0) q = (uint32_t)__builtin_amdgcn_readfirstlane((int)pw[j >> 2]);
        feed(wb[q & 0xffu]);
        q >>= 8;
    }
    if (p >= 100) feed('0' + (p / 100) % 10);
    if (p >= 10) feed('0' + (p / 10) % 10);
    feed('0' + p % 10);
    if (n & 3u) h ^= mm_k(k1);
    h = mm_fin(h, n);
    if (digit_rule && digits) return (uint32_t)num & mask;  // hashstring(): ret + seed with seed 0  [VW-unpinned]
    return h & mask;
}

// ------------------------------------------------------------------------------------------ phase H
// Walks one fragment, appending masked hashes to the LDS buffer hb; calls flush(count) whenever the buffer
// is full and once at the end (after the optional constant feature).  flush must consume hb[0..count).
// sq: 2*SEQB bytes of wave-private LDS (staged strands, then the packed code windows of pack_code_windows);
// tab: the block's table of fill_kmer4_table.
template <int HBN = HB, class Flush>
__device__ __forceinline__ void walk_fragment(const Batch &B, const Geom &G, uint64_t frag, uint32_t mask,
                                              bool add_const, uint32_t const_idx, uint8_t *sq, uint32_t *hb,
                                              const uint32_t *tab, int lane, Flush &&flush)
{
    const FragRef fr = locate_fragment(B, frag);  // never reads outside the batch buffer
    const bool raw = fr.raw;
    const uint32_t len = fr.len;
    const uint8_t *src = fr.src;
    const int k = G.k, P = G.P, t = G.t;
    const bool fast16 = (t == 16) && !raw;   // 2-bit packed: table-driven (P <= OPAL_MAX_P = 64: one or two index digits)
    const bool raw16 = (t == 16) && raw;      // text: murmur over letters in registers
    const bool digit_rule = raw && !G.hash_all;
    uint32_t fill = 0;
    const uint32_t S = G.reverse ? 2u : 1u;
    // hashes `cnt` windows whose LDS start is base_of(local index) and appends them at hb[fill + local*P + p]
    auto hash_windows = [&](uint32_t cnt, bool have_cw, auto &&base_of) {
        for (int p = 0; p < P; p++) {
            const uint8_t *pp = G.pos + p * G.ts;
            if (p == 0 && have_cw) {
                const uint8_t *cw = sq + SEQB + G.p0_run;
                for (uint32_t wl = lane; wl < cnt; wl += WAVE)
                    hb[fill + wl * (uint32_t)P] = hash_name_run<16>(cw + base_of(wl), 0, tab) & mask;
            } else if (fast16) {
                // the pattern row as 4 dwords, read once per pattern into SGPRs (left to the compiler this became a
                // vector global load + vmcnt(0) inside the window loop)
                uint32_t pq[4];
#pragma unroll
                for (int b = 0; b < 4; b++)
                    pq[b] = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t *>(pp)[b]);
                for (uint32_t wl = lane; wl < cnt; wl += WAVE)
                    hb[fill + wl * (uint32_t)P + (uint32_t)p] = hash_name_tab<16>(sq + base_of(wl), pq, p, tab) & mask;
            } else if (raw16) {
                uint32_t pq[4];
#pragma unroll
                for (int b = 0; b < 4; b++)
                    pq[b] = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t *>(pp)[b]);
                for (uint32_t wl = lane; wl < cnt; wl += WAVE) {
                    bool all_digits;
                    uint32_t h = hash_name_ascii<16>(sq + base_of(wl), pq, p, all_digits) & mask;
                    if (digit_rule && all_digits) h = hash_name_generic(sq + base_of(wl), pp, t, p, true, mask);
                    hb[fill + wl * (uint32_t)P + (uint32_t)p] = h;
                }
            } else {
                for (uint32_t wl = lane; wl < cnt; wl += WAVE)
                    hb[fill + wl * (uint32_t)P + (uint32_t)p] = hash_name_generic(sq + base_of(wl), pp, t, p, digit_rule, mask);
            }
        }
        // wave-private LDS: every lane's ds_write retires before the next ds_read issues (same wave, in-order
        // LDS queue); the fences only stop the compiler from reordering.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        fill += cnt * (uint32_t)P;
    };
    if (len >= (uint32_t)k) {
        const uint32_t W = len - (uint32_t)k + 1u;
        constexpr uint32_t LP = SEQB / 2;
        if (len <= LP) {
            // whole fragment resident: both strands staged once, windows of the two strands share one flat index
            // (fewer partially filled 64-lane steps: 274 windows -> 5 steps instead of 2 x 3)
            if (raw) {
                load_tile(sq, src, len, raw, fr.shift, 0, 0, len, lane);
                if (S == 2) load_tile(sq + LP, src, len, raw, fr.shift, 1, 0, len, lane);
            } else {
                load_strand_packed(sq, B.reads, B.reads_bytes, B.off[frag], len, 0, lane, fast16);
                if (S == 2) load_strand_packed(sq + LP, B.reads, B.reads_bytes, B.off[frag], len, 1, lane, fast16);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool have_cw = fast16 && G.p0_run >= 0;
            if (have_cw) {
                pack_code_windows(sq + SEQB, sq, len, lane);
                if (S == 2) pack_code_windows(sq + SEQB + LP, sq + LP, len, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            const uint32_t U = S * W;
            uint32_t u0 = 0;
            while (u0 < U) {
                uint32_t room = (HBN - fill) / (uint32_t)P;
                if (room < min(U - u0, (uint32_t)WAVE)) { flush(fill); fill = 0; room = HBN / (uint32_t)P; }
                const uint32_t tu = min(U - u0, room);
                hash_windows(tu, have_cw, [&](uint32_t ul) { const uint32_t u = u0 + ul; return u < W ? u : LP + (u - W); });
                u0 += tu;
            }
        } else {
            const uint32_t tw_seq = (uint32_t)(SEQB - k + 1);
            for (uint32_t s = 0; s < S; s++) {
                uint32_t w0 = 0;
                while (w0 < W) {
                    uint32_t room = (HBN - fill) / (uint32_t)P;
                    if (room < min(W - w0, (uint32_t)WAVE)) { flush(fill); fill = 0; room = HBN / (uint32_t)P; }
                    const uint32_t tw = min(min(W - w0, tw_seq), room);
                    load_tile(sq, src, len, raw, fr.shift, (int)s, w0, tw + (uint32_t)k - 1u, lane, fast16);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    hash_windows(tw, false, [&](uint32_t wl) { return wl; });
                    w0 += tw;
                }
            }
        }
    }
    if (add_const) {
        if (fill == HBN) { flush(fill); fill = 0; }
        if (lane == 0) hb[fill] = const_idx;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        fill++;
    }
    flush(fill);
}

// Work queue shared by the emit, predict and (concurrent-mode) train kernels: fragments are handed out QCHUNK at a time through a counter that
// the entry point zeroes before the launch; a wave that drew short fragments simply draws more, so ragged batches
// keep every wave busy to the end.  One atomic per QCHUNK fragments and wave (one per fragment measurably contends on
// the counter at 8M fragments/s); every wave leaves as soon as the counter passes n.
constexpr uint64_t QCHUNK = 8;
__device__ __forceinline__ uint64_t next_chunk(unsigned long long *queue, int lane, uint64_t chunk = QCHUNK)
{
    unsigned long long v = 0;
    if (lane == 0) v = atomicAdd(queue, (unsigned long long)chunk);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Work distribution of kernels whose work items are short (emit: ~8 us per fragment; the deterministic learner: one example
// or one slot range per draw).  The queue counter is ONE address and same-address atomics retire at ~83 M/s on
// this GPU (12 ns each): a wave's first draw and its last (the one that tells it the queue is empty) were 2 x 8192 atomics =
// ~200 us per launch, most of a 200k-fragment batch.  So: wave w owns chunk w outright (no atomic), the counter hands out
// the chunks after those, and a wave looks at the counter with a plain coherent load first and leaves without an atomic
// when it is already past the end.
__device__ __forceinline__ uint64_t first_chunk(int wv, uint64_t qchunk)
{
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane(wv);   // threadIdx.x / 64 is a VGPR to the compiler: keep the loop scalar
    return ((uint64_t)blockIdx.x * WPB + (uint64_t)w) * qchunk;
}
__device__ __forceinline__ uint64_t next_chunk_after(unsigned long long *queue, int lane, uint64_t qchunk, uint64_t n)
{
    const uint64_t base = (uint64_t)gridDim.x * WPB * qchunk;
    unsigned long long v = 0;
    if (base >= n) return n;   // the waves' own chunks cover the batch: nobody touches the counter
    if (lane == 0) v = atomicAdd(queue, (unsigned long long)qchunk);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return base + (((uint64_t)hi << 32) | lo);
}

// The emit kernels draw 10^5 chunks per launch, and ONE counter serves 83 M draws/s: a launch that finds nothing to do in its
// chunks (the fast kernel over a batch of long reads) still paid 12 ns per chunk, 1 ms at 1.3 M fragments.  Eight counters a
// cache line apart share the chunks behind the waves' own: counter r hands out the r-th eighth, a wave starts at counter
// (its number mod 8) and moves on to the next one when a counter has run out, until all eight have.
// Words of ctx->d_queue: [0] the one counter of the kernels that use next_chunk(), [Q_LEFT] fragments the emit fast kernel
// left to the generic one, [Q_SUBQ_FAST + 16 r], [Q_SUBQ_GENERIC + 16 r] the emit kernels' counters, [Q_BAD] out-of-buffer
// fragments seen since the last opal_sync (outside the range that is cleared per launch).
constexpr int SUBQ = 8, SUBQ_WORDS = 16;
enum { Q_LEFT = 2, Q_SUBQ_FAST = 16, Q_SUBQ_GENERIC = Q_SUBQ_FAST + SUBQ * SUBQ_WORDS, Q_BAD = Q_SUBQ_GENERIC + SUBQ * SUBQ_WORDS, Q_WORDS = 512 };
struct SubQ { uint32_t r, tried; };                       // a wave's cursor over the counters (wave-uniform)
__device__ __forceinline__ SubQ subq_start(int wv)
{
    return SubQ{(uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * WPB + (uint32_t)wv) % SUBQ)), 0u};
}
__device__ __forceinline__ uint64_t next_chunk_subq(unsigned long long *subq, int lane, uint64_t qchunk, uint64_t n, SubQ &st)
{
    const uint64_t base = (uint64_t)gridDim.x * WPB * qchunk;
    if (base >= n) return n;                               // the waves' own chunks cover the batch: nobody touches a counter
    const uint64_t nch = (n - base + qchunk - 1) / qchunk; // chunks behind the waves' own
    const uint64_t per = (nch + SUBQ - 1) / SUBQ;
    while (st.tried < (uint32_t)SUBQ) {
        unsigned long long v = 0;
        if (lane == 0) v = atomicAdd(subq + st.r * SUBQ_WORDS, 1ull);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        const uint64_t c = ((uint64_t)hi << 32) | lo, ch = (uint64_t)st.r * per + c;
        if (c < per && ch < nch) return base + ch * qchunk;
        st.r = (st.r + 1u) % (uint32_t)SUBQ;               // this counter has run out
        st.tried++;
    }
    return n;
}

// ------------------------------------------------------------------------------------------ phase S
template <int VEC> struct VecOf;
template <> struct VecOf<1> { using type = float; };
template <> struct VecOf<2> { using type = float2; };
template <> struct VecOf<4> { using type = float4; };

template <int VEC> __device__ __forceinline__ float vget(const typename VecOf<VEC>::type &v, int e);
template <> __device__ __forceinline__ float vget<1>(const float &v, int) { return v; }
template <> __device__ __forceinline__ float vget<2>(const float2 &v, int e) { return e ? v.y : v.x; }
template <> __device__ __forceinline__ float vget<4>(const float4 &v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

__device__ __forceinline__ void vadd(float &a, const float &b) { a += b; }
__device__ __forceinline__ void vadd(float2 &a, const float2 &b) { a.x += b.x; a.y += b.y; }
__device__ __forceinline__ void vadd(float4 &a, const float4 &b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// streaming (non-temporal) load of VEC floats, for tables far larger than the caches: rows are read once and never
// again.  Measured on the predict kernel with the 8 GiB table: 133.3 -> 122.1 ms at K=356 and 275 -> 253 ms at K=1564
// (the default policy tops out near 6.3 TB/s of HBM traffic, nt reaches 6.9).  Small tables keep the default policy:
// they live in L2 / the Infinity Cache, which nt loads bypass (2^20 slots: 29.6 ms default, 72.6 ms nt).
// The row (h*K) mod 2^b starts at a multiple of gcd(K, 4) floats: for K that is not a multiple of 4 (K = the number of distinct
// taxids, reference opal.py:288: any value) the 16-byte vectors of a row are only 4- or 8-byte aligned.  gfx950 global loads
// take any dword-aligned address (unaligned access mode), so the vector types below declare 4-byte alignment and the
// compiler still emits ONE global_load_dwordx4 / dwordx2 per lane; an aligned address costs nothing extra.
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int VEC, bool NT> __device__ __forceinline__ typename VecOf<VEC>::type row_load(const float *p);
template <> __device__ __forceinline__ float row_load<1, true>(const float *p) { return __builtin_nontemporal_load(p); }
template <> __device__ __forceinline__ float row_load<1, false>(const float *p) { return *p; }
template <> __device__ __forceinline__ float2 row_load<2, true>(const float *p)
{
    const f2u v = __builtin_nontemporal_load(reinterpret_cast<const f2u *>(p));
    return make_float2(v.x, v.y);
}
template <> __device__ __forceinline__ float2 row_load<2, false>(const float *p)
{
    const f2u v = *reinterpret_cast<const f2u *>(p);
    return make_float2(v.x, v.y);
}
template <> __device__ __forceinline__ float4 row_load<4, true>(const float *p)
{
    const f4u v = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <> __device__ __forceinline__ float4 row_load<4, false>(const float *p)
{
    const f4u v = *reinterpret_cast<const f4u *>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// Does the row that starts at slot `row` come within 3 slots of the table's end or pass it (row + K + 3 > 2^b)?  Wave-uniform.
// Rows that do not are "contiguous": every 16-byte vector of theirs, the last, partial one included, lies inside the table,
// and row + class never overflows 32 bits.
struct RowCross {
    uint32_t thr;   // 2^b - K - 3
    bool all;       // K + 3 > 2^b (toy tables of the tests): every row does
    __device__ __forceinline__ RowCross(uint32_t K, uint32_t mask) : thr((mask - K - 2u) & mask), all((uint64_t)K + 3ull > (uint64_t)mask + 1ull) {}
    __device__ __forceinline__ bool operator()(uint32_t row) const { return all || row > thr; }
};

// ---- vectors of the trainers' tables.  A row (h*K) mod 2^b starts at any slot when K is not a multiple of 4, so the 16-byte
// vectors of a row are only dword aligned (loads and stores through f4u / f2u: still one dwordx4 / dwordx2 instruction).
template <int VEC> __device__ __forceinline__ void vloadu(const float *p, float (&o)[VEC])
{
    if constexpr (VEC == 4) { const f4u v = *reinterpret_cast<const f4u *>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    else if constexpr (VEC == 2) { const f2u v = *reinterpret_cast<const f2u *>(p); o[0] = v.x; o[1] = v.y; }
    else o[0] = *p;
}
template <int VEC> __device__ __forceinline__ void vstoreu(float *p, const float (&o)[VEC])
{
    if constexpr (VEC == 4) { f4u v; v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3]; *reinterpret_cast<f4u *>(p) = v; }
    else if constexpr (VEC == 2) { f2u v; v.x = o[0]; v.y = o[1]; *reinterpret_cast<f2u *>(p) = v; }
    else *p = o[0];
}
// The VEC slots of row `row` from class c on, for a lane that has work (`on`; lanes without read slot 0, an always-cached
// line, so that the load itself is unconditional).  `crossing` is wave-uniform (RowCross): the row reaches the table's end
// and every slot is wrapped on its own; otherwise the row is contiguous and ONE vector load serves the lane -- when K is not
// a multiple of VEC the row's last vector reaches up to VEC - 1 slots past the row (still inside the table); what it finds
// there belongs to no class of this row and the callers mask it.
// MASKED (K a multiple of VEC, so a vector never straddles the table's end): the lane wraps its own address, no case split.
template <int VEC, bool MASKED = false>
__device__ __forceinline__ void row_vload(const float *tab, uint32_t row, uint32_t c, uint32_t K, uint32_t mask, bool crossing, bool on,
                                          float (&o)[VEC])
{
    if constexpr (MASKED) {
        vloadu<VEC>(tab + (on ? (size_t)((row + c) & mask) : (size_t)0), o);
    } else if (!crossing) {
        vloadu<VEC>(tab + (size_t)(on ? row + c : 0u), o);      // 32-bit sum: cannot overflow for a contiguous row
    } else {
#pragma unroll
        for (int e = 0; e < VEC; e++) o[e] = tab[on ? (size_t)((row + min(c + (uint32_t)e, K - 1u)) & mask) : (size_t)0];
    }
}
// Stores the lane's slots of the row that belong to a class (c + e < K): one vector store when all VEC do and the row is
// contiguous, slot by slot otherwise (the row's last, partial vector; rows that pass the table's end).
template <int VEC, bool MASKED = false>
__device__ __forceinline__ void row_vstore(float *tab, uint32_t row, uint32_t c, uint32_t K, uint32_t mask, bool crossing, const float (&o)[VEC])
{
    if constexpr (MASKED) {
        vstoreu<VEC>(tab + (size_t)((row + c) & mask), o);
    } else if (!crossing && c + (uint32_t)VEC <= K) {
        vstoreu<VEC>(tab + (size_t)(row + c), o);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; e++)
            if (c + (uint32_t)e < K) tab[(size_t)((row + c + (uint32_t)e) & mask)] = o[e];
    }
}

// Streams rows hb[0..nf) of the padded weight table into acc.  hb[nf .. nf+U) must be readable (dummy rows:
// they are loaded and never consumed).  loff[j] = this lane's float offset inside a row for vector j (already
// clamped into the row, so every load is unconditional).
template <int VEC, int NV, int U, int WRAP = 0, bool NT = false>
__device__ __forceinline__ void stream_rows(const float *__restrict__ tab, const uint32_t *hb, uint32_t nf, uint32_t K,
                                            uint32_t mask, const uint32_t (&loff)[NV],
                                            typename VecOf<VEC>::type (&acc)[NV])
{
    using V = typename VecOf<VEC>::type;
    V r[U][NV];
    // WRAP = 0: the table is padded with a mirror of its head, rows run on into it (predict).
    // WRAP = 1: canonical slots only, every lane wraps its own address (trainers, K a multiple of VEC: the mirror may be stale).
    // WRAP = 2: every row of hb is contiguous (RowCross says so for all of them: the trainers check a flush before they call):
    //           no wrap, and row + lo is a 32-bit sum formed per load (no 64-bit address pair held per vector).
    auto row_of = [&](uint32_t f) -> uint32_t {
        const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb[f]);
        return (idx * K) & mask;  // (h*K) mod 2^b, b <= 32: low 32 bits suffice
    };
    auto ld = [&](uint32_t row, uint32_t lo) -> V {
        if constexpr (WRAP == 1) return row_load<VEC, NT>(tab + (size_t)((row + lo) & mask));
        else if constexpr (WRAP == 2) return row_load<VEC, NT>(tab + (size_t)(row + lo));   // trainer: nt measured 3 % slower
        else return row_load<VEC, NT>(tab + (size_t)row + lo);
    };
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t row = row_of(u);
#pragma unroll
        for (int j = 0; j < NV; j++) r[u][j] = ld(row, loff[j]);
    }
    uint32_t f = 0;
    for (; f + U <= nf; f += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t row = row_of(f + U + u);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                vadd(acc[j], r[u][j]);
                r[u][j] = ld(row, loff[j]);
            }
        }
    }
    const uint32_t rem = nf - f;  // < U, wave-uniform
#pragma unroll
    for (int u = 0; u < U - 1; u++) {
        if ((uint32_t)u < rem) {
#pragma unroll
            for (int j = 0; j < NV; j++) vadd(acc[j], r[u][j]);
        }
    }
}

// The trainers' sweep 1 over a flush that holds a row which passes the table's end (rare on real tables, the rule on the
// tiny tables of the tests): row by row, every slot wrapped on its own, same order of additions per class as stream_rows.
// cls[j] = first class of the lane's vector j (classes >= K re-read the row's last slot and are never used).
template <int VEC, int NV>
__device__ __forceinline__ void stream_rows_wrapped(const float *__restrict__ tab, const uint32_t *hb, uint32_t nf, uint32_t K, uint32_t mask,
                                                    const uint32_t (&cls)[NV], typename VecOf<VEC>::type (&acc)[NV])
{
    float a[NV][VEC];
#pragma unroll
    for (int j = 0; j < NV; j++)
#pragma unroll
        for (int e = 0; e < VEC; e++) a[j][e] = vget<VEC>(acc[j], e);
    for (uint32_t f = 0; f < nf; f++) {
        const uint32_t row = ((uint32_t)__builtin_amdgcn_readfirstlane((int)hb[f]) * K) & mask;
#pragma unroll
        for (int j = 0; j < NV; j++)
#pragma unroll
            for (int e = 0; e < VEC; e++) a[j][e] += tab[(size_t)((row + min(cls[j] + (uint32_t)e, K - 1u)) & mask)];
    }
#pragma unroll
    for (int j = 0; j < NV; j++) {
        if constexpr (VEC == 4) acc[j] = make_float4(a[j][0], a[j][1], a[j][2], a[j][3]);
        else if constexpr (VEC == 2) acc[j] = make_float2(a[j][0], a[j][1]);
        else acc[j] = a[j][0];
    }
}
// true when one of the rows hb[0..nf) passes the table's end (wave-uniform result)
__device__ __forceinline__ bool any_row_crosses(const uint32_t *hb, uint32_t nf, uint32_t K, uint32_t mask, int lane)
{
    const RowCross crosses(K, mask);
    bool x = false;
    for (uint32_t i = lane; i < nf; i += WAVE) x = x || crosses((hb[i] * K) & mask);
    return __ballot(x) != 0ull;
}

}  // namespace
