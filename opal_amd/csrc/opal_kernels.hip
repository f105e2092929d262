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

// ---- emit fast path: 2-bit packed fragments of at most EF_MAXLEN bases, t = 16, positions ascending inside each block of 4.
// The r01 kernel spent its time in VALU issue (96 %) and in the LDS (72 % busy): per hash 16 byte gathers with an address
// add each, 12 shift-ors to build the four table indices, and a hash buffer written and read back.  Here a wave builds,
// per strand, ONE dword per base: Z[i] = the 16 codes i-1 .. i+14 (32 bits of the packed strand from bit 2i-2), so that
//   * the code of base i+j sits at bits 2+2j of Z[i]: the j-th letter of a gathered block is taken from Z[pos_j - j] under
//     the mask 3 << (2+2j), and the four masked words or-ed together ARE the block's byte offset into the 256-entry table
//     (4 VALU instructions per block instead of 7 + 4 address adds);
//   * Z[i+1] holds the codes i .. i+15: a pattern that is a run of 16 consecutive positions (Opal's pattern 0) gets all
//     four table offsets from ONE read (byte b of the word, times 4).
// Lanes own consecutive windows, so every gather is `ds_read_addtid_b32` (address = M0 + 4*lane, no address VGPR, no
// VALU): the wave-uniform part (window step + pattern position) goes through M0 on the scalar unit.  Hashes go straight
// from registers to HBM (lane u writes features u*P .. u*P+P-1: contiguous across the wave), no LDS hash buffer.
constexpr int EF_MAXLEN = 512;           // bases per strand (2 x 250/300-bp reads included; 19 KiB of LDS per block of 4 waves still leaves the 7 blocks per CU the registers allow)
constexpr int EF_ZLEN = EF_MAXLEN + 64;  // Z dwords per strand: lanes past the last window read (and discard) up to 63 entries further
constexpr int EF_PKLEN = 40;             // packed dwords per strand: one zero dword in front (Z[0] starts 2 bits before base 0), 33 of data, slack
                                         // (the Z build reads up to entry 37: whole iterations of 64)
constexpr int EF_WAVE_BYTES = 2 * SEQB + HB_EMIT * 4;   // per-wave LDS of the generic kernel (strands + hash buffer)

// Reads Z[lane + (base + q_i)/4] for wave-uniform byte offsets.  M0 is written one instruction before the read that uses it
// (the ISA asks for one wait state between an SALU write of M0 and an LDS "add-TID" instruction): the add that forms the
// next address is that instruction, so a gather costs two scalar instructions and no vector one.
// M0 is on the clobber lists: the compiler reserves it and merges its own M0 initialisations across code that does not
// visibly define it.  clang warns that clobbers of reserved registers "may not be preserved" (-Winline-asm, switched off
// in the build flags): not being preserved is what the clobber says.
__device__ __forceinline__ uint32_t zread1(uint32_t base, uint32_t q0)
{
    uint32_t x, t;
    asm volatile("s_add_u32 %1, %2, %3\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tds_read_addtid_b32 %0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x), "=&s"(t) : "s"(base), "s"(q0) : "memory", "scc", "m0");
    return x;
}
__device__ __forceinline__ void zread4(uint32_t base, uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t (&x)[4])
{
    uint32_t t0, t1;
    asm volatile("s_add_u32 %4, %6, %7\n\ts_mov_b32 m0, %4\n\t"
                 "s_add_u32 %5, %6, %8\n\tds_read_addtid_b32 %0\n\ts_mov_b32 m0, %5\n\t"
                 "s_add_u32 %4, %6, %9\n\tds_read_addtid_b32 %1\n\ts_mov_b32 m0, %4\n\t"
                 "s_add_u32 %5, %6, %10\n\tds_read_addtid_b32 %2\n\ts_mov_b32 m0, %5\n\t"
                 "s_nop 0\n\tds_read_addtid_b32 %3\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&s"(t0), "=&s"(t1)
                 : "s"(base), "s"(q0), "s"(q1), "s"(q2), "s"(q3) : "memory", "scc", "m0");
}
__device__ __forceinline__ void zread8(uint32_t base, const uint32_t (&q)[8], uint32_t (&x)[8])
{
    uint32_t t0, t1;
    asm volatile("s_add_u32 %8, %10, %11\n\ts_mov_b32 m0, %8\n\t"
                 "s_add_u32 %9, %10, %12\n\tds_read_addtid_b32 %0\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %8, %10, %13\n\tds_read_addtid_b32 %1\n\ts_mov_b32 m0, %8\n\t"
                 "s_add_u32 %9, %10, %14\n\tds_read_addtid_b32 %2\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %8, %10, %15\n\tds_read_addtid_b32 %3\n\ts_mov_b32 m0, %8\n\t"
                 "s_add_u32 %9, %10, %16\n\tds_read_addtid_b32 %4\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %8, %10, %17\n\tds_read_addtid_b32 %5\n\ts_mov_b32 m0, %8\n\t"
                 "s_add_u32 %9, %10, %18\n\tds_read_addtid_b32 %6\n\ts_mov_b32 m0, %9\n\t"
                 "s_nop 0\n\tds_read_addtid_b32 %7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]),
                   "=&s"(t0), "=&s"(t1)
                 : "s"(base), "s"(q[0]), "s"(q[1]), "s"(q[2]), "s"(q[3]), "s"(q[4]), "s"(q[5]), "s"(q[6]), "s"(q[7]) : "memory", "scc", "m0");
}
// zread4 + zread8 in one batch (the KT experiment: pattern 0's four per-base terms with the first half of pattern 1's gathers)
__device__ __forceinline__ void zread12(uint32_t base, const uint32_t (&k)[4], const uint32_t (&q)[8], uint32_t (&kx)[4], uint32_t (&x)[8])
{
    uint32_t t0, t1;
    asm volatile("s_add_u32 %12, %14, %15\n\ts_mov_b32 m0, %12\n\t"
                 "s_add_u32 %13, %14, %16\n\tds_read_addtid_b32 %0\n\ts_mov_b32 m0, %13\n\t"
                 "s_add_u32 %12, %14, %17\n\tds_read_addtid_b32 %1\n\ts_mov_b32 m0, %12\n\t"
                 "s_add_u32 %13, %14, %18\n\tds_read_addtid_b32 %2\n\ts_mov_b32 m0, %13\n\t"
                 "s_add_u32 %12, %14, %19\n\tds_read_addtid_b32 %3\n\ts_mov_b32 m0, %12\n\t"
                 "s_add_u32 %13, %14, %20\n\tds_read_addtid_b32 %4\n\ts_mov_b32 m0, %13\n\t"
                 "s_add_u32 %12, %14, %21\n\tds_read_addtid_b32 %5\n\ts_mov_b32 m0, %12\n\t"
                 "s_add_u32 %13, %14, %22\n\tds_read_addtid_b32 %6\n\ts_mov_b32 m0, %13\n\t"
                 "s_add_u32 %12, %14, %23\n\tds_read_addtid_b32 %7\n\ts_mov_b32 m0, %12\n\t"
                 "s_add_u32 %13, %14, %24\n\tds_read_addtid_b32 %8\n\ts_mov_b32 m0, %13\n\t"
                 "s_add_u32 %12, %14, %25\n\tds_read_addtid_b32 %9\n\ts_mov_b32 m0, %12\n\t"
                 "s_add_u32 %13, %14, %26\n\tds_read_addtid_b32 %10\n\ts_mov_b32 m0, %13\n\t"
                 "s_nop 0\n\tds_read_addtid_b32 %11\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(kx[0]), "=&v"(kx[1]), "=&v"(kx[2]), "=&v"(kx[3]), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]),
                   "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&s"(t0), "=&s"(t1)
                 : "s"(base), "s"(k[0]), "s"(k[1]), "s"(k[2]), "s"(k[3]), "s"(q[0]), "s"(q[1]), "s"(q[2]), "s"(q[3]), "s"(q[4]), "s"(q[5]),
                   "s"(q[6]), "s"(q[7]) : "memory", "scc", "m0");
}
// zread1 + zread8 in one batch (the canonical two-pattern step: pattern 0's one word and the first half of pattern 1's gathers)
__device__ __forceinline__ void zread9(uint32_t base, uint32_t r, const uint32_t (&q)[8], uint32_t &xr, uint32_t (&x)[8])
{
    uint32_t t0, t1;
    asm volatile("s_add_u32 %9, %11, %12\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %10, %11, %13\n\tds_read_addtid_b32 %0\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %9, %11, %14\n\tds_read_addtid_b32 %1\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %10, %11, %15\n\tds_read_addtid_b32 %2\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %9, %11, %16\n\tds_read_addtid_b32 %3\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %10, %11, %17\n\tds_read_addtid_b32 %4\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %9, %11, %18\n\tds_read_addtid_b32 %5\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %10, %11, %19\n\tds_read_addtid_b32 %6\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %9, %11, %20\n\tds_read_addtid_b32 %7\n\ts_mov_b32 m0, %9\n\t"
                 "s_nop 0\n\tds_read_addtid_b32 %8\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(xr), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]),
                   "=&s"(t0), "=&s"(t1)
                 : "s"(base), "s"(r), "s"(q[0]), "s"(q[1]), "s"(q[2]), "s"(q[3]), "s"(q[4]), "s"(q[5]), "s"(q[6]), "s"(q[7]) : "memory", "scc", "m0");
}
// zread1 + nine more in one batch (the paired form: pattern 0's word, a row's eight words and its spare)
__device__ __forceinline__ void zread10(uint32_t base, uint32_t r, const uint32_t (&q)[9], uint32_t &xr, uint32_t (&x)[9])
{
    uint32_t t0, t1;
    asm volatile("s_add_u32 %10, %12, %13\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %11, %12, %14\n\tds_read_addtid_b32 %0\n\ts_mov_b32 m0, %11\n\t"
                 "s_add_u32 %10, %12, %15\n\tds_read_addtid_b32 %1\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %11, %12, %16\n\tds_read_addtid_b32 %2\n\ts_mov_b32 m0, %11\n\t"
                 "s_add_u32 %10, %12, %17\n\tds_read_addtid_b32 %3\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %11, %12, %18\n\tds_read_addtid_b32 %4\n\ts_mov_b32 m0, %11\n\t"
                 "s_add_u32 %10, %12, %19\n\tds_read_addtid_b32 %5\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %11, %12, %20\n\tds_read_addtid_b32 %6\n\ts_mov_b32 m0, %11\n\t"
                 "s_add_u32 %10, %12, %21\n\tds_read_addtid_b32 %7\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %11, %12, %22\n\tds_read_addtid_b32 %8\n\ts_mov_b32 m0, %11\n\t"
                 "s_nop 0\n\tds_read_addtid_b32 %9\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(xr), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]),
                   "=&s"(t0), "=&s"(t1)
                 : "s"(base), "s"(r), "s"(q[0]), "s"(q[1]), "s"(q[2]), "s"(q[3]), "s"(q[4]), "s"(q[5]), "s"(q[6]), "s"(q[7]), "s"(q[8]) : "memory", "scc", "m0");
}
// nine words in one batch (the paired form: a row's eight words and its spare)
__device__ __forceinline__ void zread9w(uint32_t base, const uint32_t (&q)[9], uint32_t (&x)[9])
{
    uint32_t t0, t1;
    asm volatile("s_add_u32 %9, %11, %12\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %10, %11, %13\n\tds_read_addtid_b32 %0\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %9, %11, %14\n\tds_read_addtid_b32 %1\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %10, %11, %15\n\tds_read_addtid_b32 %2\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %9, %11, %16\n\tds_read_addtid_b32 %3\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %10, %11, %17\n\tds_read_addtid_b32 %4\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %9, %11, %18\n\tds_read_addtid_b32 %5\n\ts_mov_b32 m0, %9\n\t"
                 "s_add_u32 %10, %11, %19\n\tds_read_addtid_b32 %6\n\ts_mov_b32 m0, %10\n\t"
                 "s_add_u32 %9, %11, %20\n\tds_read_addtid_b32 %7\n\ts_mov_b32 m0, %9\n\t"
                 "s_nop 0\n\tds_read_addtid_b32 %8\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]),
                   "=&s"(t0), "=&s"(t1)
                 : "s"(base), "s"(q[0]), "s"(q[1]), "s"(q[2]), "s"(q[3]), "s"(q[4]), "s"(q[5]), "s"(q[6]), "s"(q[7]), "s"(q[8]) : "memory", "scc", "m0");
}
// the block's table offset from the four words that hold its letters at bits 2-3, 4-5, 6-7, 8-9 (v_and + 3 v_and_or)
// (left to the compiler the same expression takes 5 instructions: three v_and, one v_and_or, one v_or3)
__device__ __forceinline__ uint32_t and_or(uint32_t x, uint32_t m, uint32_t acc)   // (x & m) | acc, m wave-uniform
{
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(m), "v"(acc));
    return d;
}
__device__ __forceinline__ uint32_t block_addr(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
    return and_or(x3, 0x300u, and_or(x2, 0xc0u, and_or(x1, 0x30u, x0 & 0xcu)));
}
// The same from TWO words: each holds two of the block's letters, the first where it belongs (bits 2-3 / 6-7), the second
// 2 (gap - 1) bits further up (gap = distance of the two positions): a rotate by a wave-uniform amount brings it to bits 4-5 / 8-9.
// A row may have one pair whose letters lie too far apart for one word (1 row in 8 of ldpc.py's has one, 1 in 600 two): its second
// letter comes from the row's SPARE word, read where the letter sits in place; the masks say which (m_a / m_b: 0x30 / 0x300, or 0 for
// the pair the spare serves; m_s: the spare's letter when it belongs to this block, else 0).  Eight (nine with the spare)
// LDS reads per row instead of 16, for 24 (28) VALU instructions per row instead of 16: the kernels were LDS-bound (91 % busy) when
// this was built and sit at 93 % VALU issue / 82 % LDS since.
template <bool SPARE>
__device__ __forceinline__ uint32_t block_addr_pair(uint32_t xa, uint32_t sa, uint32_t ma, uint32_t xb, uint32_t sb, uint32_t mb,
                                                    uint32_t xs, uint32_t ms)
{
    const uint32_t ra = __builtin_amdgcn_alignbit(xa, xa, sa), rb = __builtin_amdgcn_alignbit(xb, xb, sb);
    const uint32_t a = and_or(rb, mb, and_or(xb, 0xc0u, and_or(ra, ma, xa & 0xcu)));
    return SPARE ? and_or(xs, ms, a) : a;
}
// PAIR (template argument of the straight-line kernels): 0 = one letter per word, 1 = two letters per word, every pair of every
// row inside one word (a third of ldpc.py's numHash-8 geometries, 87 % of its numHash-1 ones), 2 = with the spare word (98 % / 99.8 %)
// A gathered row in its paired form (opal_set_patterns): record words [FASTPAT_PAIR ..)
struct PairRec { uint32_t w[9], rot[8], m2[8], ms[4]; };
constexpr int PAIRREC_DW = 29;
// which words of a row's record a kernel variant reads (the others never reach a register): PAIR 0 the 16 word offsets, PAIR 1
// eight words and eight rotates, PAIR 2 the whole PairRec
template <int PAIR>
__device__ constexpr bool pairrec_used(int i) { return PAIR == 2 || (PAIR == 1 ? (i < 8 || (i >= 9 && i < 17)) : i < 16); }
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// One step of 64 windows (flat index u0 + lane over the forward strand's W windows, then the reverse complement's).
// MODE 0: every lane is on the forward strand, 1: on the reverse strand, 2: the step straddles the two (at most one step
// per fragment): every gather is then made for both strands and picked per lane.
template <int MODE>
__device__ __forceinline__ void emit_fast_step(const uint32_t *__restrict__ fast, int P, uint32_t zf, uint32_t zr, uint32_t u0,
                                               uint32_t W, uint32_t U, uint32_t mask, const uint32_t *tab, int lane,
                                               uint32_t *__restrict__ dst, bool pair_store)
{
    const uint32_t u = u0 + (uint32_t)lane;
    const bool live = u < U;
    const bool on_rc = u >= W;
    const uint32_t bf = zf + u0 * 4u;                  // forward: Z_f[u + q]
    const uint32_t br = zr + (u0 - W) * 4u;            // reverse: Z_r[u - W + q] (for lanes before W it points into Z_f's slack: discarded)
    const uint32_t b0 = MODE == 1 ? br : bf;
    auto rd4 = [&](uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t (&x)[4]) {
        if constexpr (MODE != 2) {
            zread4(b0, q0, q1, q2, q3, x);
        } else {
            uint32_t y[4];
            zread4(bf, q0, q1, q2, q3, x);
            zread4(br, q0, q1, q2, q3, y);
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = on_rc ? y[i] : x[i];
        }
    };
    const char *tabc = reinterpret_cast<const char *>(tab);
    uint32_t h_prev = 0;
    for (int p = 0; p < P; p++) {
        // the pattern's record through the scalar cache (constant address space: the kernel never writes it), straight
        // into SGPRs; as a plain global load it became a vector load + a v_readfirstlane per word, per pattern and step
        typedef const uint32_t __attribute__((address_space(4))) *const_u32;
        const const_u32 fp = (const_u32)(uintptr_t)(fast + p * FASTPAT_DW);
        const uint32_t tail = fp[16];
        const uint32_t runs = fp[17];
        uint32_t a[4];
        if (runs == 0xfu) {                            // a run of 16 consecutive positions (Opal's pattern 0): one read
            const uint32_t rq = fp[18];
            uint32_t x = zread1(b0, rq);
            if constexpr (MODE == 2) { const uint32_t y = zread1(br, rq); x = on_rc ? y : x; }
#pragma unroll
            for (int b = 0; b < 4; b++) a[b] = ((x >> (8 * b)) & 0xffu) << 2;
        } else if (runs == 0u && MODE != 2) {          // an LDPC row: 16 gathers in two batches
            uint32_t x[8];
#pragma unroll
            for (int hb2 = 0; hb2 < 2; hb2++) {
                uint32_t q[8];
#pragma unroll
                for (int i = 0; i < 8; i++) q[i] = fp[8 * hb2 + i];
                zread8(b0, q, x);
                a[2 * hb2] = block_addr(x[0], x[1], x[2], x[3]);
                a[2 * hb2 + 1] = block_addr(x[4], x[5], x[6], x[7]);
            }
        } else {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                uint32_t x[4];
                rd4(fp[4 * b], fp[4 * b + 1], fp[4 * b + 2], fp[4 * b + 3], x);
                a[b] = block_addr(x[0], x[1], x[2], x[3]);     // for a run block the four reads hit the same word: still right
            }
        }
        // a[b] has bits 2..9 only.  Lanes past the last window may have picked up stale words: their hashes are never stored.
        uint32_t h = *reinterpret_cast<const uint32_t *>(tabc + 1024 + a[0]);   // the hash after the first block
#pragma unroll
        for (int b = 1; b < 4; b++) h = mm_hk(h, *reinterpret_cast<const uint32_t *>(tabc + a[b]));
        h = mm_fmix(h ^ tail) & mask;
        if (pair_store) {                              // P == 2, 8-byte aligned: one global_store_dwordx2 per lane
            if (p == 0) h_prev = h;
            else if (live) *reinterpret_cast<uint2 *>(dst + 2u * u) = make_uint2(h_prev, h);
        } else if (live) {
            dst[u * (uint32_t)P + (uint32_t)p] = h;
        }
    }
}

// A fragment's packed dwords, fetched ahead of their use: lane l (< EF_PKLEN) holds the two dwords that, funnel-shifted by
// the fragment's offset inside its first dword, give dword l - 1 of the fragment.  The fast kernels issue these loads for the
// NEXT fragment of a draw before they walk the current one, so that the HBM latency of a random 50-byte read (views into
// 1.5 GB of genomes: nothing is cache-resident) is hidden behind a fragment's worth of hashing.
struct EmitFetch { uint32_t lo, hi; };
__device__ __forceinline__ EmitFetch emit_fetch(const Batch &B, uint64_t off, int lane)
{
    const uint32_t *__restrict__ R32 = reinterpret_cast<const uint32_t *>(B.reads);
    const uint64_t nd = B.reads_bytes >> 2;
    const uint64_t a0 = off >> 4;
    EmitFetch f{0u, 0u};
    if (lane > 0 && lane < EF_PKLEN) {
        f.lo = R32[min(a0 + (uint64_t)lane - 1u, nd - 1)];
        f.hi = R32[min(a0 + (uint64_t)lane, nd - 1)];
    }
    return f;
}

// Stages one fragment (2-bit packed, len <= EF_MAXLEN, its dwords inside the buffer, fetched by emit_fetch) as the per-base arrays Z_f, Z_r.
// ZL: Z entries per strand (EF_ZLEN serves fragments of up to EF_MAXLEN bases).  KT (experiment): also KT[i] = the block table's
// term of codes i .. i+3 for every base, behind the packed dwords, so that a run pattern reads its four terms per window with
// add-TID reads instead of four conflicted table reads.
template <int ZL = EF_ZLEN, bool KT = false>
__device__ __forceinline__ void emit_stage(const EmitFetch &pf, const Geom &G, uint64_t off, uint32_t len, uint32_t *zw, int lane,
                                           const uint32_t *tab = nullptr)
{
    uint32_t *zf = zw, *zr = zw + ZL, *pkf = zw + 2 * ZL, *pkr = pkf + EF_PKLEN;
    const uint32_t sh = (uint32_t)(off & 15u) * 2u;
    const uint32_t S = G.reverse ? 2u : 1u;
    // the fragment's bases as packed dwords behind one zero dword: base 0 of the fragment at bit 0 of pkf[1]
    if (lane < EF_PKLEN) pkf[lane] = lane > 0 ? __builtin_amdgcn_alignbit(pf.hi, pf.lo, sh) : 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (S == 2 && lane < EF_PKLEN) {
        // reverse complement, packed: dword m = comp(codes len-1-16m .. len-16-16m): the 32 forward bits at base len-16-16m
        // with the order of their 2-bit groups reversed (v_bfrev + swap inside the pairs) and complemented
        uint32_t v = 0;
        if (lane > 0) {
            const int s = (int)len - 16 * lane;             // m = lane - 1
            uint32_t x = 0;
            if (s >= 0) x = __builtin_amdgcn_alignbit(pkf[(s >> 4) + 2], pkf[(s >> 4) + 1], (uint32_t)(s & 15) * 2u);
            else if (s > -16) x = pkf[1] << (uint32_t)(-s * 2);
            const uint32_t y = __builtin_bitreverse32(x);
            v = ~(((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u));
        }
        pkr[lane] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // Z[i] = 32 bits of the strand from bit 2i-2, for i = lane + 64 it (Z[len] serves run patterns; the entries of the last
    // iteration that lie past it are never used).  64 entries are 4 packed dwords further on, so the funnel shift is the
    // same in every iteration and all LDS addresses are one per-lane base + immediates: one VALU instruction per 64 entries.
    {
        const uint32_t bo = 2u * (uint32_t)lane + 30u, b = bo & 31u;
        const uint32_t *pff = pkf + (bo >> 5), *pr = pkr + (bo >> 5);
        uint32_t *zfl = zf + lane, *zrl = zr + lane;
        uint32_t *kfl = pkr + EF_PKLEN + lane, *krl = kfl + ZL;
        const char *tabc = reinterpret_cast<const char *>(tab);
#pragma unroll
        for (int it = 0; it < (ZL - WAVE) / WAVE + 1; it++) {
            if ((uint32_t)(it * WAVE) > len) break;
            const uint32_t vf = __builtin_amdgcn_alignbit(pff[4 * it + 1], pff[4 * it], b);
            zfl[it * WAVE] = vf;
            if constexpr (KT) kfl[it * WAVE] = *reinterpret_cast<const uint32_t *>(tabc + (vf & 0x3fcu));   // bits 2..9 = codes i .. i+3
            if (S == 2) {
                const uint32_t vr = __builtin_amdgcn_alignbit(pr[4 * it + 1], pr[4 * it], b);
                zrl[it * WAVE] = vr;
                if constexpr (KT) krl[it * WAVE] = *reinterpret_cast<const uint32_t *>(tabc + (vr & 0x3fcu));
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Hashes fragment `frag` (2-bit packed, k <= len <= EF_MAXLEN, its dwords inside the buffer) into dst[0 .. S*W*P).
__device__ __forceinline__ void emit_fast(const EmitFetch &pf, const Geom &G, uint64_t off, uint32_t len, uint32_t mask, uint32_t *zw,
                                          const uint32_t *tab, int lane, uint32_t *__restrict__ dst)
{
    emit_stage(pf, G, off, len, zw, lane);
    uint32_t *zf = zw, *zr = zw + EF_ZLEN;
    const uint32_t S = G.reverse ? 2u : 1u;
    const uint32_t W = len - (uint32_t)G.k + 1u, U = S * W;
    const uint32_t zfa = sgpr((uint32_t)(uintptr_t)zf), zra = sgpr((uint32_t)(uintptr_t)zr);
    const bool pair_store = G.P == 2 && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0;
    for (uint32_t u0 = 0; u0 < U; u0 += WAVE) {
        if (u0 + WAVE <= W || S == 1) emit_fast_step<0>(G.fast, G.P, zfa, zra, u0, W, U, mask, tab, lane, dst, pair_store);
        else if (u0 >= W) emit_fast_step<1>(G.fast, G.P, zfa, zra, u0, W, U, mask, tab, lane, dst, pair_store);
        else emit_fast_step<2>(G.fast, G.P, zfa, zra, u0, W, U, mask, tab, lane, dst, pair_store);
    }
    // the next fragment's staging writes must not overtake this fragment's reads (same wave: LDS ops retire in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Which fragments the fast kernel takes: 2-bit packed, k <= len <= EF_MAXLEN, and every dword the staging needs inside
// the buffer (locate_fragment's rule for the fragment itself; fragments in the buffer's last, partial dword go the generic way).
__device__ __forceinline__ bool emit_fast_takes(const Batch &B, const Geom &G, uint32_t lw, uint64_t off)
{
    // (off <= 4 * reads_bytes first: an offset near 2^64 must not wrap the sums below and slip through; locate_fragment
    // rejects it, so such a fragment goes to the generic kernel and is reported by opal_sync)
    return !(lw & OPAL_RAW_FLAG) && lw >= (uint32_t)G.k && lw <= (uint32_t)G.fast_maxlen && off <= (B.reads_bytes << 2) &&
           ((off + lw + 3) >> 2) <= B.reads_bytes && ((off + lw + 15) >> 4) <= (B.reads_bytes >> 2);
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

// Adds the fragments a block of the fast kernels left over into queue[2]: per block, not per wave (8192 same-address atomics
// are 90 us, which is what a launch over an all-text batch cost for taking nothing).
__device__ __forceinline__ void emit_report_left(uint32_t *s_left, uint32_t left, int lane, unsigned long long *queue)
{
    if (left && lane == 0) atomicAdd(s_left, left);
    __syncthreads();
    if (threadIdx.x == 0 && *s_left) atomicAdd(queue + Q_LEFT, (unsigned long long)*s_left);
}

// ---- The canonical Opal geometry with numHash = 1 (ldpc.py: pattern 0 = a run of 16 positions, pattern 1 = one LDPC row;
// P = 2, output 8-byte aligned) has a kernel of its own: the two pattern records live in SGPRs for the whole kernel, a step
// is straight-line code (no pattern loop, no per-pattern dispatch), and the LDS round trips of the two patterns overlap:
// pattern 0's word travels with the first half of pattern 1's gathers, the first six table reads with the second half.
// MIXED: the one step per fragment whose lanes straddle the two strands reads both and picks per lane.
// PAIR: pattern 1's letters come two to a word (q = a PairRec: see block_addr_pair and opal_set_patterns), so the step's ten
// add-TID reads are one batch and the table reads follow in one go.
template <bool MIXED, bool KT = false, int PAIR = 0>
__device__ __forceinline__ void canon2_step(uint32_t bf, uint32_t br, bool on_rc, uint32_t rq, const uint32_t (&q)[PAIRREC_DW], uint32_t tail0,
                                            uint32_t tail1, uint32_t u, bool live, uint32_t mask, const uint32_t *tab,
                                            uint32_t *__restrict__ dst, uint32_t ktoff = 0u)
{
    const char *tabc = reinterpret_cast<const char *>(tab);
    // (tried: pattern 1's four lookups from a copy of the table in global memory, through the otherwise idle L1, to take
    // them off the LDS: 3.97 -> 4.93 ms; a 64-lane dword gather costs the vector cache far more than the LDS's conflicts)
    // (tried: 32 bank-private copies of the table, entry e of copy c at dword 32 e + c, lane l reading copy l & 31, in blocks
    // of 12-16 waves that share the 32 KB: SQ_LDS_BANK_CONFLICT 6.7e8 -> 0 and SQ_LDS_IDX_ACTIVE -33 % as predicted, but the
    // address shifts and the second table recomputed in registers add 18 % VALU instructions, and at most 6 waves per SIMD
    // stay resident: 3.905 ms against 3.915, i.e. nothing; with 14-16 waves per block 4.5-4.8 ms)
    auto T = [&](uint32_t a) { return *reinterpret_cast<const uint32_t *>(tabc + a); };
    uint32_t x[8], x0;
    const uint32_t qa[8] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]};
    const uint32_t qb[8] = {q[8], q[9], q[10], q[11], q[12], q[13], q[14], q[15]};
    // MIXED: every lane reads through its own address (its strand's array): one v_add per gather instead of reading both
    // strands' arrays with add-TID reads and picking (the LDS is the unit this kernel is bound by, the VALU has slack)
    const uint32_t vb = MIXED ? (on_rc ? br : bf) + 4u * (uint32_t)(threadIdx.x & (WAVE - 1)) : 0u;
    auto vread = [&](uint32_t q) { return *reinterpret_cast<const uint32_t __attribute__((address_space(3))) *>((uintptr_t)(vb + q)); };
    uint32_t a0[4], a1[4];
    uint32_t t00, t01, t02, t03;
    uint32_t xp[9];
    if constexpr (PAIR != 0) {
        const uint32_t qw[9] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]};
        const uint32_t q8[8] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]};
        xp[8] = 0u;
        if constexpr (MIXED) {
            x0 = vread(rq);
#pragma unroll
            for (int i = 0; i < (PAIR == 2 ? 9 : 8); i++) xp[i] = vread(qw[i]);
        } else if constexpr (PAIR == 2) {
            zread10(bf, rq, qw, x0, xp);
        } else {
            uint32_t x8[8];
            zread9(bf, rq, q8, x0, x8);
#pragma unroll
            for (int i = 0; i < 8; i++) xp[i] = x8[i];
        }
#pragma unroll
        for (int b = 0; b < 4; b++) a0[b] = ((x0 >> (8 * b)) & 0xffu) << 2;
    } else if constexpr (KT) {
        // pattern 0's terms are per-base: window w's block b is base w + r + 4b: KT[w + r + 4b] (rq = 4 (r + 1))
        const uint32_t kq[4] = {ktoff + rq - 4u, ktoff + rq + 12u, ktoff + rq + 28u, ktoff + rq + 44u};
        uint32_t kx[4];
        if constexpr (MIXED) {
#pragma unroll
            for (int i = 0; i < 4; i++) kx[i] = vread(kq[i]);
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = vread(qa[i]);
        } else {
            zread12(bf, kq, qa, kx, x);
        }
        t00 = mm_hk(0u, kx[0]); t01 = kx[1]; t02 = kx[2]; t03 = kx[3];
    } else {
        if constexpr (MIXED) {
            x0 = vread(rq);
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = vread(qa[i]);
        } else {
            zread9(bf, rq, qa, x0, x);
        }
#pragma unroll
        for (int b = 0; b < 4; b++) a0[b] = ((x0 >> (8 * b)) & 0xffu) << 2;
    }
    if constexpr (PAIR != 0) {
        static_assert(!KT, "the per-base term experiment has no paired form");
        // q: [0..8] the nine words, [9..16] rotates, [17..24] second-letter masks, [25..28] spare masks
#pragma unroll
        for (int b = 0; b < 4; b++)
            a1[b] = block_addr_pair<PAIR == 2>(xp[2 * b], q[9 + 2 * b], PAIR == 1 ? 0x30u : q[17 + 2 * b], xp[2 * b + 1], q[10 + 2 * b],
                                               PAIR == 1 ? 0x300u : q[18 + 2 * b], xp[8], q[25 + b]);   // (PAIR == 1: every pair inside its word, the masks are constants)
        t00 = T(1024u + a0[0]); t01 = T(a0[1]); t02 = T(a0[2]); t03 = T(a0[3]);
        const uint32_t t10 = T(1024u + a1[0]), t11 = T(a1[1]), t12 = T(a1[2]), t13 = T(a1[3]);
        const uint32_t h0 = mm_fmix(mm_hk(mm_hk(mm_hk(t00, t01), t02), t03) ^ tail0) & mask;
        const uint32_t h1 = mm_fmix(mm_hk(mm_hk(mm_hk(t10, t11), t12), t13) ^ tail1) & mask;
        // (byte offset formed in 32 bits: the store then takes the wave-uniform base from SGPRs and one VGPR offset, no 64-bit add)
        if (live) *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(dst) + (uint32_t)(u * 8u)) = make_uint2(h0, h1);
        return;
    }
    a1[0] = block_addr(x[0], x[1], x[2], x[3]);
    a1[1] = block_addr(x[4], x[5], x[6], x[7]);
    if constexpr (!KT) { t00 = T(1024u + a0[0]); t01 = T(a0[1]); t02 = T(a0[2]); t03 = T(a0[3]); }
    const uint32_t t10 = T(1024u + a1[0]), t11 = T(a1[1]);
    if constexpr (MIXED) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = vread(qb[i]);
    } else {
        zread8(bf, qb, x);
    }
    a1[2] = block_addr(x[0], x[1], x[2], x[3]);
    a1[3] = block_addr(x[4], x[5], x[6], x[7]);
    const uint32_t t12 = T(a1[2]), t13 = T(a1[3]);
    const uint32_t h0 = mm_fmix(mm_hk(mm_hk(mm_hk(t00, t01), t02), t03) ^ tail0) & mask;
    const uint32_t h1 = mm_fmix(mm_hk(mm_hk(mm_hk(t10, t11), t12), t13) ^ tail1) & mask;
    if (live) *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(dst) + (uint32_t)(u * 8u)) = make_uint2(h0, h1);
}

template <bool KT, int ZL, int PAIR = 0>
__global__ __launch_bounds__(BLOCK) void emit_canon2_kernel(Batch B, Geom G, uint32_t mask, const uint64_t *__restrict__ out_off,
                                                             uint32_t *__restrict__ out_idx, uint64_t out_cap,
                                                             unsigned long long *__restrict__ queue, uint64_t qchunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_wave[WPB][(KT ? 4 : 2) * ZL + 2 * EF_PKLEN];
    constexpr uint32_t KTOFF = (2u * ZL + 2u * EF_PKLEN) * 4u;     // KT_f - Z_f = KT_r - Z_r, in bytes
    __shared__ uint32_t s_tab[KTAB];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    __shared__ uint32_t s_left;   // fragments this block leaves to the generic kernel: ONE atomic per block at the end
    if (threadIdx.x == 0) s_left = 0;
    fill_kmer4_table(s_tab);
    typedef const uint32_t __attribute__((address_space(4))) *const_u32;
    const const_u32 f0 = (const_u32)(uintptr_t)G.fast, f1 = (const_u32)(uintptr_t)(G.fast + FASTPAT_DW);
    uint32_t q[PAIRREC_DW];
#pragma unroll
    for (int i = 0; i < PAIRREC_DW; i++) q[i] = pairrec_used<PAIR>(i) ? f1[(PAIR != 0 ? FASTPAT_PAIR : 0) + i] : 0u;
    const uint32_t rq = f0[18], tail0 = f0[16], tail1 = f1[16];
    uint32_t *zw = s_wave[wv];
    const uint32_t zfa = sgpr((uint32_t)(uintptr_t)zw), zra = zfa + ZL * 4u;
    const uint32_t S = G.reverse ? 2u : 1u;
    uint32_t left = 0;
    SubQ sq = subq_start(wv);
    for (uint64_t q0 = first_chunk(wv, qchunk); q0 < B.n; q0 = next_chunk_subq(queue + Q_SUBQ_FAST, lane, qchunk, B.n, sq)) {
        const uint64_t mine = q0 + (uint64_t)lane;
        uint64_t o0 = 0, off = 0;
        uint32_t lw = 0;
        bool take = false, mine_left = false;
        if ((uint64_t)lane < qchunk && mine < B.n) {
            o0 = out_off[mine];
            lw = B.len[mine];
            off = B.off[mine];
            const bool fits = !(o0 > out_cap || feature_count(lw, G) > out_cap - o0);   // never write outside out_idx
            take = fits && emit_fast_takes(B, G, lw, off);
            mine_left = fits && !take;
        }
        left += (uint32_t)__popcll(__ballot(mine_left));   // wave-uniform
        uint64_t todo = __ballot(take);
        EmitFetch nxt{0u, 0u};
        if (todo) nxt = emit_fetch(B, readlane64(off, __builtin_ctzll(todo)), lane);
        while (todo) {
            const int i = __builtin_ctzll(todo);
            todo &= todo - 1;
            const EmitFetch cur = nxt;
            if (todo) nxt = emit_fetch(B, readlane64(off, __builtin_ctzll(todo)), lane);   // in flight while fragment i is hashed
            uint32_t *dst = out_idx + readlane64(o0, i);
            const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)lw, i);
            emit_stage<ZL, KT>(cur, G, readlane64(off, i), len, zw, lane, s_tab);
            const uint32_t W = len - (uint32_t)G.k + 1u, U = S * W;
            for (uint32_t u0 = 0; u0 < U; u0 += WAVE) {
                const uint32_t u = u0 + (uint32_t)lane;
                const uint32_t bf = zfa + u0 * 4u, br = zra + (u0 - W) * 4u;
                // the fragment's last step under EXEC = its live lanes only (18 of 64 at 200 bases): the LDS serves 32 lanes a
                // pass and the table reads collide less with fewer lanes
                if (u0 + WAVE > U && u >= U) continue;
                if (u0 + WAVE <= W || S == 1) canon2_step<false, KT, PAIR>(bf, 0u, false, rq, q, tail0, tail1, u, true, mask, s_tab, dst, KTOFF);
                else if (u0 >= W) canon2_step<false, KT, PAIR>(br, 0u, false, rq, q, tail0, tail1, u, true, mask, s_tab, dst, KTOFF);
                else canon2_step<true, KT, PAIR>(bf, br, u >= W, rq, q, tail0, tail1, u, true, mask, s_tab, dst, KTOFF);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next fragment's staging stays behind these reads
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    emit_report_left(&s_left, left, lane, queue);
}

// (Tried in round 3, after the paired letters had made the kernel VALU-bound: the draw's fragments hashed as ONE stream of windows --
// two fragments staged in two LDS slots of 320 entries per strand, and when fewer than 64 windows of fragment A are left the step's
// other lanes start fragment B's forward strand through the per-lane-address step, with a per-lane output pointer: 4.3 steps per
// 200-base fragment instead of 5, the last of them no longer spent on 18 live lanes.  Correct (all parity tests), and worth 2 %:
// cfg5 25.4 -> 24.9 ms, 200 bases 6.33 -> 6.40e11 hashes/s, 300 bases -2 %, 100 bases -11 % (no chaining below 64 windows per strand,
// more bookkeeping per fragment): the second per-lane-address step per fragment, 6 instead of 7 waves per SIMD and the scalar state of
// two fragments eat what the saved step gives.  Removed.  One lesson kept: fetched-ahead registers must be redefined at ONE place of
// a loop; redefined on several paths they are copied on every back edge, which waits for the loads in flight at every step.)
// ---- The canonical Opal geometry with numHash = 2 ... 8 (ldpc.py: pattern 0 a run of 16 positions, patterns 1 .. P-1 LDPC rows;
// the reference CLI's default is --num-hash 8, P = 9: opal.py:534-536) gets the same treatment as numHash = 1: a step is
// straight-line code over all P patterns (P is a template argument: no pattern loop, no per-pattern dispatch), the P hashes of a
// window stay in registers and leave as ONE run of P consecutive dwords per lane (dwordx4 / dwordx2 / dword stores on dword
// alignment) instead of P stores 4 P bytes apart, and the records of the LDPC rows stream through the scalar cache one pattern
// ahead (17 SGPRs each: all P - 1 of them do not fit the scalar register file, so they are re-read every step; the scalar loads
// travel with the previous pattern's gathers).  Pattern 0's one word rides with the first eight gathers of pattern 1 (zread9).
typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u2u __attribute__((ext_vector_type(2), aligned(4)));
struct PatRec { uint32_t q[PAIRREC_DW]; uint32_t tail; };   // q: 16 word offsets, or a PairRec (PAIR)
template <int PAIR>
__device__ __forceinline__ PatRec load_patrec(uint64_t fast_base, int p)
{
    // the base goes through an empty asm so that the loads below cannot be hoisted out of the step (and all records held at once)
    asm volatile("" : "+s"(fast_base));
    typedef const uint32_t __attribute__((address_space(4))) *const_u32;
    const const_u32 fp = (const_u32)(uintptr_t)(fast_base + (uint64_t)p * (FASTPAT_DW * 4u));
    PatRec r;
#pragma unroll
    for (int i = 0; i < PAIRREC_DW; i++) r.q[i] = pairrec_used<PAIR>(i) ? fp[(PAIR != 0 ? FASTPAT_PAIR : 0) + i] : 0u;
    r.tail = fp[16];
    return r;
}
template <int P, bool MIXED, int PAIR>
__device__ __forceinline__ void canonP_step(uint32_t bf, uint32_t br, bool on_rc, uint64_t fast_base, uint32_t rq, uint32_t tail0, uint32_t u,
                                            bool live, uint32_t mask, const uint32_t *tab, uint32_t *__restrict__ dst)
{
    const char *tabc = reinterpret_cast<const char *>(tab);
    auto T = [&](uint32_t a) { return *reinterpret_cast<const uint32_t *>(tabc + a); };
    // MIXED: per-lane addresses into the lane's own strand (see canon2_step)
    const uint32_t vb = MIXED ? (on_rc ? br : bf) + 4u * (uint32_t)(threadIdx.x & (WAVE - 1)) : 0u;
    auto vread = [&](uint32_t q) { return *reinterpret_cast<const uint32_t __attribute__((address_space(3))) *>((uintptr_t)(vb + q)); };
    auto rd8 = [&](const uint32_t (&q)[8], uint32_t (&x)[8]) {
        if constexpr (MIXED) {
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = vread(q[i]);
        } else {
            zread8(bf, q, x);
        }
    };
    uint32_t h[P];
    PatRec rec = load_patrec<PAIR>(fast_base, 1);
#pragma unroll
    for (int p = 1; p < P; p++) {
        PatRec nxt = rec;
        if (p + 1 < P) nxt = load_patrec<PAIR>(fast_base, p + 1);  // in flight during this pattern's gathers
        if constexpr (PAIR != 0) {
            // two letters to a word: eight reads (PAIR == 2: and the spare) are all of the row
            constexpr int NW = PAIR == 2 ? 9 : 8;
            const uint32_t qw[9] = {rec.q[0], rec.q[1], rec.q[2], rec.q[3], rec.q[4], rec.q[5], rec.q[6], rec.q[7], rec.q[8]};
            const uint32_t q8[8] = {rec.q[0], rec.q[1], rec.q[2], rec.q[3], rec.q[4], rec.q[5], rec.q[6], rec.q[7]};
            uint32_t xp[9], x8[8];
            xp[8] = 0u;
            uint32_t x0 = 0u;
            if constexpr (MIXED) {
                if (p == 1) x0 = vread(rq);
#pragma unroll
                for (int i = 0; i < NW; i++) xp[i] = vread(qw[i]);
            } else if constexpr (PAIR == 2) {
                if (p == 1) zread10(bf, rq, qw, x0, xp);           // pattern 0's word travels with pattern 1's
                else zread9w(bf, qw, xp);
            } else {
                if (p == 1) zread9(bf, rq, q8, x0, x8);
                else zread8(bf, q8, x8);
#pragma unroll
                for (int i = 0; i < 8; i++) xp[i] = x8[i];
            }
            if (p == 1) {
                uint32_t a0[4];
#pragma unroll
                for (int b = 0; b < 4; b++) a0[b] = ((x0 >> (8 * b)) & 0xffu) << 2;
                h[0] = mm_fmix(mm_hk(mm_hk(mm_hk(T(1024u + a0[0]), T(a0[1])), T(a0[2])), T(a0[3])) ^ tail0) & mask;
            }
            uint32_t a[4];
#pragma unroll
            for (int b = 0; b < 4; b++)
                a[b] = block_addr_pair<PAIR == 2>(xp[2 * b], rec.q[9 + 2 * b], PAIR == 1 ? 0x30u : rec.q[17 + 2 * b], xp[2 * b + 1], rec.q[10 + 2 * b],
                                                  PAIR == 1 ? 0x300u : rec.q[18 + 2 * b], xp[8], rec.q[25 + b]);
            const uint32_t t0 = T(1024u + a[0]), t1 = T(a[1]), t2 = T(a[2]), t3 = T(a[3]);
            h[p] = mm_fmix(mm_hk(mm_hk(mm_hk(t0, t1), t2), t3) ^ rec.tail) & mask;
        } else {
        const uint32_t qa[8] = {rec.q[0], rec.q[1], rec.q[2], rec.q[3], rec.q[4], rec.q[5], rec.q[6], rec.q[7]};
        const uint32_t qb[8] = {rec.q[8], rec.q[9], rec.q[10], rec.q[11], rec.q[12], rec.q[13], rec.q[14], rec.q[15]};
        uint32_t x[8];
        if (p == 1) {                                              // pattern 0's word travels with pattern 1's first half
            uint32_t x0;
            if constexpr (MIXED) {
                x0 = vread(rq);
                rd8(qa, x);
            } else {
                zread9(bf, rq, qa, x0, x);
            }
            uint32_t a0[4];
#pragma unroll
            for (int b = 0; b < 4; b++) a0[b] = ((x0 >> (8 * b)) & 0xffu) << 2;
            h[0] = mm_fmix(mm_hk(mm_hk(mm_hk(T(1024u + a0[0]), T(a0[1])), T(a0[2])), T(a0[3])) ^ tail0) & mask;
        } else {
            rd8(qa, x);
        }
        const uint32_t t0 = T(1024u + block_addr(x[0], x[1], x[2], x[3])), t1 = T(block_addr(x[4], x[5], x[6], x[7]));
        rd8(qb, x);
        const uint32_t t2 = T(block_addr(x[0], x[1], x[2], x[3])), t3 = T(block_addr(x[4], x[5], x[6], x[7]));
        h[p] = mm_fmix(mm_hk(mm_hk(mm_hk(t0, t1), t2), t3) ^ rec.tail) & mask;
        }
        rec = nxt;
    }
    if (live) {                                                    // features u*P .. u*P+P-1: one run of P dwords
        uint32_t *o = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(dst) + (uint32_t)(u * (uint32_t)(4 * P)));   // (32-bit byte offset: SGPR base + one VGPR)
        int i = 0;
#pragma unroll
        for (; i + 4 <= P; i += 4) { u4u v; v.x = h[i]; v.y = h[i + 1]; v.z = h[i + 2]; v.w = h[i + 3]; *reinterpret_cast<u4u *>(o + i) = v; }
        if (P - i >= 2) { u2u v; v.x = h[i]; v.y = h[i + 1]; *reinterpret_cast<u2u *>(o + i) = v; i += 2; }
        if (P - i == 1) o[i] = h[i];
    }
}

// (tried: every step through per-lane addresses -- compiler-scheduled ds_read_b32, one v_add per gather, no M0 traffic, finer
// waits -- instead of the add-TID batches: 4-13 % slower at numHash 4 and 8)
template <int P, int PAIR>
__global__ __launch_bounds__(BLOCK) void emit_canonP_kernel(Batch B, Geom G, uint32_t mask, const uint64_t *__restrict__ out_off,
                                                             uint32_t *__restrict__ out_idx, uint64_t out_cap,
                                                             unsigned long long *__restrict__ queue, uint64_t qchunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_wave[WPB][2 * EF_ZLEN + 2 * EF_PKLEN];
    __shared__ uint32_t s_tab[KTAB];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    __shared__ uint32_t s_left;   // fragments this block leaves to the generic kernel: ONE atomic per block at the end
    if (threadIdx.x == 0) s_left = 0;
    fill_kmer4_table(s_tab);
    typedef const uint32_t __attribute__((address_space(4))) *const_u32;
    const const_u32 f0 = (const_u32)(uintptr_t)G.fast;
    const uint32_t rq = f0[18], tail0 = f0[16];
    const uint64_t fast_base = (uint64_t)(uintptr_t)G.fast;
    uint32_t *zw = s_wave[wv];
    const uint32_t zfa = sgpr((uint32_t)(uintptr_t)zw), zra = zfa + EF_ZLEN * 4u;
    const uint32_t S = G.reverse ? 2u : 1u;
    uint32_t left = 0;
    SubQ sq = subq_start(wv);
    for (uint64_t q0 = first_chunk(wv, qchunk); q0 < B.n; q0 = next_chunk_subq(queue + Q_SUBQ_FAST, lane, qchunk, B.n, sq)) {
        const uint64_t mine = q0 + (uint64_t)lane;
        uint64_t o0 = 0, off = 0;
        uint32_t lw = 0;
        bool take = false, mine_left = false;
        if ((uint64_t)lane < qchunk && mine < B.n) {
            o0 = out_off[mine];
            lw = B.len[mine];
            off = B.off[mine];
            const bool fits = !(o0 > out_cap || feature_count(lw, G) > out_cap - o0);   // never write outside out_idx
            take = fits && emit_fast_takes(B, G, lw, off);
            mine_left = fits && !take;
        }
        left += (uint32_t)__popcll(__ballot(mine_left));   // wave-uniform
        uint64_t todo = __ballot(take);
        EmitFetch nxt{0u, 0u};
        if (todo) nxt = emit_fetch(B, readlane64(off, __builtin_ctzll(todo)), lane);
        while (todo) {
            const int i = __builtin_ctzll(todo);
            todo &= todo - 1;
            const EmitFetch cur = nxt;
            if (todo) nxt = emit_fetch(B, readlane64(off, __builtin_ctzll(todo)), lane);   // in flight while fragment i is hashed
            uint32_t *dst = out_idx + readlane64(o0, i);
            const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)lw, i);
            emit_stage(cur, G, readlane64(off, i), len, zw, lane);
            const uint32_t W = len - (uint32_t)G.k + 1u, U = S * W;
            for (uint32_t u0 = 0; u0 < U; u0 += WAVE) {
                const uint32_t u = u0 + (uint32_t)lane;
                const uint32_t bf = zfa + u0 * 4u, br = zra + (u0 - W) * 4u;
                if (u0 + WAVE > U && u >= U) continue;         // the last step under EXEC = its live lanes (see emit_canon2_kernel)
                if (u0 + WAVE <= W || S == 1) canonP_step<P, false, PAIR>(bf, 0u, false, fast_base, rq, tail0, u, true, mask, s_tab, dst);
                else if (u0 >= W) canonP_step<P, false, PAIR>(br, 0u, false, fast_base, rq, tail0, u, true, mask, s_tab, dst);
                else canonP_step<P, true, PAIR>(bf, br, u >= W, fast_base, rq, tail0, u, true, mask, s_tab, dst);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next fragment's staging stays behind these reads
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    emit_report_left(&s_left, left, lane, queue);
}

// The two emit kernels share one work loop: a wave draws qchunk (<= 64) consecutive fragments, every lane looks at one of
// them (inside out_cap? the fast kernel's kind or the generic kernel's?), and the wave then walks the selected ones.
// FAST = true: emit_fast for the fragments emit_fast_takes() accepts; counts the others in queue[2].
// FAST = false: the generic walker for every fragment (fast_ran == 0) or for those the fast kernel left (fast_ran != 0;
// it returns at once when that kernel left none).
// amdgpu_waves_per_eu(8, 8): the generic walker fits 64 VGPRs without spilling when told to (left alone the compiler takes 77
// and loses two waves per SIMD: the text path fell from 327 M to 229 M fragments/s)
template <bool FAST>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FAST ? 7 : 8, 8))) void emit_idx_kernel(Batch B, Geom G, uint32_t mask, const uint64_t *__restrict__ out_off,
                                                          uint32_t *__restrict__ out_idx, uint64_t out_cap,
                                                          unsigned long long *__restrict__ queue, uint64_t qchunk, int fast_ran)
{
    constexpr int WAVE_BYTES = FAST ? (2 * EF_ZLEN + 2 * EF_PKLEN) * 4 : EF_WAVE_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t s_wave[WPB][WAVE_BYTES];
    __shared__ uint32_t s_tab[KTAB];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    if (!FAST && fast_ran && __builtin_nontemporal_load(queue + Q_LEFT) == 0ull) return;   // block-uniform: nothing was left over
    __shared__ uint32_t s_left;
    if (FAST && threadIdx.x == 0) s_left = 0;
    fill_kmer4_table(s_tab);
    uint32_t left = 0;
    SubQ sq = subq_start(wv);
    for (uint64_t q0 = first_chunk(wv, qchunk); q0 < B.n; q0 = next_chunk_subq(queue + (FAST ? Q_SUBQ_FAST : Q_SUBQ_GENERIC), lane, qchunk, B.n, sq)) {
        const uint64_t mine = q0 + (uint64_t)lane;
        uint64_t o0 = 0, off = 0;
        uint32_t lw = 0;
        bool take = false, mine_left = false;
        if ((uint64_t)lane < qchunk && mine < B.n) {
            o0 = out_off[mine];
            lw = B.len[mine];
            off = B.off[mine];
            const bool fits = !(o0 > out_cap || feature_count(lw, G) > out_cap - o0);   // never write outside out_idx
            const bool fast_kind = emit_fast_takes(B, G, lw, off);
            take = fits && (FAST ? fast_kind : !(fast_ran && fast_kind));
            mine_left = FAST && fits && !fast_kind;
        }
        if (FAST) left += (uint32_t)__popcll(__ballot(mine_left));   // wave-uniform
        if constexpr (FAST) {
            uint64_t todo = __ballot(take);
            EmitFetch nxt{0u, 0u};
            if (todo) nxt = emit_fetch(B, readlane64(off, __builtin_ctzll(todo)), lane);
            while (todo) {
                const int i = __builtin_ctzll(todo);
                todo &= todo - 1;
                const EmitFetch cur = nxt;
                if (todo) nxt = emit_fetch(B, readlane64(off, __builtin_ctzll(todo)), lane);   // in flight while fragment i is hashed
                emit_fast(cur, G, readlane64(off, i), (uint32_t)__builtin_amdgcn_readlane((int)lw, i), mask,
                          reinterpret_cast<uint32_t *>(s_wave[wv]), s_tab, lane, out_idx + readlane64(o0, i));
            }
        } else {
            // nothing but the mask stays live across the walker (it needs every register it can get: 61 VGPRs, 8 waves per
            // SIMD; holding the lanes' offsets and lengths for v_readlane cost 15 more and a quarter of the text path's rate)
            for (uint64_t todo = __ballot(take); todo; todo &= todo - 1) {
                const uint64_t frag = q0 + (uint64_t)__builtin_ctzll(todo);
                uint32_t *dst = out_idx + out_off[frag];
                // a fragment whose (offset, length) reaches outside d_reads has no bases to hash, yet owns a range of the
                // output (the offsets come from the lengths alone): fill it with 0xffffffff and count it for opal_sync
                if (locate_fragment(B, frag).len == 0) {
                    const uint64_t cnt = feature_count(B.len[frag], G);
                    for (uint64_t j = lane; j < cnt; j += WAVE) dst[j] = 0xffffffffu;
                    if (cnt && lane == 0) atomicAdd(queue + Q_BAD, 1ull);
                    continue;
                }
                uint32_t *hb = reinterpret_cast<uint32_t *>(s_wave[wv] + 2 * SEQB);
                walk_fragment<HB_EMIT>(B, G, frag, mask, false, 0u, s_wave[wv], hb, s_tab, lane, [&](uint32_t cnt) {
                    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {   // wave-uniform: 16-byte stores, then the tail
                        const uint32_t n4 = cnt / 4u;
                        for (uint32_t j = lane; j < n4; j += WAVE)
                            reinterpret_cast<uint4 *>(dst)[j] = reinterpret_cast<const uint4 *>(hb)[j];
                        for (uint32_t j = n4 * 4u + lane; j < cnt; j += WAVE) dst[j] = hb[j];
                    } else {
                        for (uint32_t j = lane; j < cnt; j += WAVE) dst[j] = hb[j];
                    }
                    dst += cnt;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                });
            }
        }
    }
    if constexpr (FAST) emit_report_left(&s_left, left, lane, queue);
}

// Debug: the t gathered characters of every feature (reference order), t bytes per feature.
__global__ __launch_bounds__(BLOCK) void emit_names_kernel(Batch B, Geom G, const uint64_t *__restrict__ out_off,
                                                            uint8_t *__restrict__ out_names, uint64_t out_cap,
                                                            unsigned long long *__restrict__ queue)
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
        if (len == 0 && feature_count(B.len[frag], G) > 0) {   // reaches outside d_reads: NUL names, counted for opal_sync
            const uint64_t o = out_off[frag], cnt = feature_count(B.len[frag], G);
            if (o <= out_cap && cnt <= out_cap - o) {
                for (uint64_t j = lane; j < cnt * (uint64_t)t; j += WAVE) out_names[o * (uint64_t)t + j] = 0;
                if (lane == 0) atomicAdd(queue + Q_BAD, 1ull);
            }
            continue;
        }
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
                // lanes past the row's end re-read its last vector (same cache line); when K is not a multiple of VEC the last
                // live lane does too: its vector starts at K - VEC and overlaps its neighbour's classes (same sums, same order)
                loff[j] = min(cls[j], Ku - VEC);
            }
            V acc[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) acc[j] = V{};
            walk_fragment(B, G, frag, mask, add_constant != 0, const_idx, s_seq[wv], hb, s_tab, lane, [&](uint32_t cnt) {
                // dummy rows for the prefetch overrun: the constant row (always cache-hot)
                if (lane < HB_PAD) hb[cnt + lane] = const_idx;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                stream_rows<VEC, NV, U, 0, NT>(tab, hb, cnt, Ku, mask, loff, acc);
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
                        if (best_c == 0xffffffffu || s > best) { best = s; best_c = loff[j] + e; }
                    }
                    if (out_scores) {
                        float *o = out_scores + frag * (uint64_t)Ku + loff[j];
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
    OPAL_V(1, 1, 8),   // K = 1
    OPAL_V(2, 1, 8),   // K = 2, 3
    OPAL_V(4, 1, 8), OPAL_V(4, 2, 8), OPAL_V(4, 3, 4), OPAL_V(4, 4, 4), OPAL_V(4, 5, 2), OPAL_V(4, 6, 2), OPAL_V(4, 7, 2), OPAL_V(4, 8, 2),
};

constexpr uint64_t NT_TABLE_BYTES = 4ull << 30;   // tables from this size on are streamed with non-temporal loads

const Variant &pick_variant(int K)
{
    // 16-byte row loads for every K >= 4 whatever K mod 4 is (dword-aligned dwordx4 loads, see row_load): with 4-byte loads
    // for odd K the kernel ran at 77 % instead of 86 % of the HBM peak at K = 1563 against 1564 (scripts/oddk_bench.py)
    const int vec = K >= 4 ? 4 : K >= 2 ? 2 : 1;
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
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_queue, Q_WORDS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ctx->d_queue, 0, Q_WORDS * sizeof(unsigned long long));
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
    if (ctx->d_fast) (void)hipFree(ctx->d_fast);
    if (ctx->d_w && ctx->w_owned) (void)hipFree(ctx->d_w);
    if (ctx->d_g2) (void)hipFree(ctx->d_g2);
    if (ctx->ing_tiles) (void)hipFree(ctx->ing_tiles);
    if (ctx->ing_rec) (void)hipFree(ctx->ing_rec);
    if (ctx->d_block_sums) (void)hipFree(ctx->d_block_sums);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    opal::dtrain_free(ctx);
    delete ctx;
    return OPAL_OK;
}

extern "C" int opal_sync(opal_ctx *ctx)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_sync: null ctx");
    // fragments of opal_features / opal_feature_names calls that reached outside their d_reads (queue word Q_BAD): read and
    // cleared on the ctx's own stream, so that a ctx on a stream of its own never serialises against the null stream
    unsigned long long bad = 0;
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    OPAL_HIP_CHECK(ctx, hipMemcpyAsync(&bad, ctx->d_queue + Q_BAD, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (bad) {
        OPAL_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_queue + Q_BAD, 0, sizeof(bad), ctx->stream));
        OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return opal::fail(ctx, OPAL_EINVAL, std::to_string(bad) + " fragment(s) of the emit calls since the last opal_sync reach outside d_reads "
                                            "(offset + length beyond reads_bytes): their features were written as 0xffffffff / NUL names");
    }
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
    // emit fast path (see emit_fast): t == 16 and, inside every block of 4, position j at least j (true for ascending rows)
    if (ctx->d_fast) { OPAL_HIP_CHECK(ctx, hipFree(ctx->d_fast)); ctx->d_fast = nullptr; }
    bool fast_ok = t == 16, canon2 = false, pair_ok = true, spare_used = false;
    std::vector<uint32_t> fp((size_t)P * FASTPAT_DW, 0u);
    auto host_mm_k = [](uint32_t k1) { k1 *= 0xcc9e2d51u; k1 = (k1 << 15) | (k1 >> 17); return k1 * 0x1b873593u; };
    for (int p = 0; p < P && fast_ok; p++) {
        uint32_t *r = &fp[(size_t)p * FASTPAT_DW];
        uint32_t runs = 0;
        for (int b = 0; b < 4; b++) {
            bool run = true;
            for (int j = 0; j < 4; j++) {
                const int q = pos[p * t + 4 * b + j] - j;
                if (q < 0) { fast_ok = false; break; }
                r[4 * b + j] = (uint32_t)(4 * q);
                run = run && pos[p * t + 4 * b + j] == pos[p * t + 4 * b] + j;
            }
            if (run && b > 0 && pos[p * t + 4 * b] != pos[p * t + 4 * b - 4] + 4) run = false;   // blocks must also follow each other
            if (run) runs |= 1u << b;
        }
        if (runs != 0xfu) runs = 0;     // the kernel only distinguishes "one run of 16" from "gather everything"
        if (p == 0) canon2 = runs == 0xfu;
        else canon2 = canon2 && runs == 0;
        const uint32_t digits = p < 10 ? (uint32_t)('0' + p) : ((uint32_t)('0' + p / 10) | ((uint32_t)('0' + p % 10) << 8));
        r[16] = host_mm_k(digits) ^ (uint32_t)(t + (p < 10 ? 1 : 2));
        r[17] = runs;
        r[18] = (uint32_t)(4 * (pos[p * t] + 1));
        // the paired form (block_addr_pair; record words FASTPAT_PAIR + [0..8] words, [9..16] rotates, [17..24] second-letter masks,
        // [25..28] spare masks): word 2b + m serves letters 4b + 2m and 4b + 2m + 1 when the second lies 1 .. 14 - 2m positions
        // behind the first (still inside the word's 16 bases); ONE pair of the row may fail that: the spare word serves its second letter
        uint32_t *pr = r + FASTPAT_PAIR;
        int failed = 0;
        pr[8] = 0;                                                // no spare needed: any word of the window (its mask is 0)
        for (int b = 0; b < 4; b++)
            for (int m = 0; m < 2; m++) {
                const int a = p * t + 4 * b + 2 * m, gap = pos[a + 1] - pos[a], w = 2 * b + m;
                const bool ok = gap >= 1 && 2 * m + gap <= 14;
                pr[w] = r[4 * b + 2 * m];
                pr[9 + w] = ok ? (uint32_t)(2 * (gap - 1)) : 0u;
                pr[17 + w] = ok ? (m ? 0x300u : 0x30u) : 0u;
                if (!ok && failed++ == 0) {
                    pr[8] = r[4 * b + 2 * m + 1];                 // the second letter in place: bits 4-5 (m = 0) or 8-9 (m = 1)
                    pr[25 + b] = m ? 0x300u : 0x30u;
                }
            }
        if (failed > 1) pair_ok = false;
        if (failed == 1) spare_used = true;
    }
    if (fast_ok) {
        OPAL_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_fast, fp.size() * sizeof(uint32_t)));
        OPAL_HIP_CHECK(ctx, hipMemcpy(ctx->d_fast, fp.data(), fp.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    ctx->fast_canon2 = fast_ok && P == 2 && canon2;
    ctx->fast_canonP = fast_ok && P >= 3 && P <= 9 && canon2;   // a run followed by P - 1 gathered rows: emit_canonP_kernel<P>
    ctx->fast_pair = !(fast_ok && canon2 && pair_ok) ? 0 : spare_used ? 2 : 1;   // the gathered rows' letters two to a word (2: a row needs its spare word)
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
    opal::drop_adopted_table(ctx);
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

extern "C" int opal_adopt_weights_device(opal_ctx *ctx, float *d_w, uint64_t n_slots, uint64_t capacity_floats, int K, int bits)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_adopt_weights_device: null ctx");
    if (!d_w) return opal::fail(ctx, OPAL_EINVAL, "opal_adopt_weights_device: null table");
    if (bits < 1 || bits > 32 || n_slots != (1ull << bits))
        return opal::fail(ctx, OPAL_EINVAL, "opal_adopt_weights_device: need 1 <= bits <= 32 and n_slots == 2^bits");
    if (K < 1 || K > (1 << 20)) return opal::fail(ctx, OPAL_EINVAL, "opal_adopt_weights_device: need 1 <= K <= 2^20");
    if (capacity_floats < n_slots + (uint64_t)K + 4)
        return opal::fail(ctx, OPAL_EINVAL, "opal_adopt_weights_device: the buffer needs room for n_slots + K + 4 floats (the mirror of the table's head)");
    if ((reinterpret_cast<uintptr_t>(d_w) & 15u) != 0) return opal::fail(ctx, OPAL_EINVAL, "opal_adopt_weights_device: the table must be 16-byte aligned");
    OPAL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    OPAL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_w && ctx->w_owned) OPAL_HIP_CHECK(ctx, hipFree(ctx->d_w));
    ctx->d_w = d_w; ctx->w_alloc_floats = capacity_floats; ctx->w_owned = false;
    ctx->n_slots = n_slots; ctx->K = K; ctx->bits = bits; ctx->g2_valid = false;
    return finish_table(ctx);
}

extern "C" int opal_set_tuning(opal_ctx *ctx, int predict_nt, uint64_t emit_qchunk, int emit_fast, uint64_t train_group)
{
    if (!ctx) return opal::fail(nullptr, OPAL_EINVAL, "opal_set_tuning: null ctx");
    if (predict_nt < -1 || predict_nt > 1) return opal::fail(ctx, OPAL_EINVAL, "opal_set_tuning: predict_nt is -1 (by table size), 0 or 1");
    ctx->tune_nt = predict_nt;
    ctx->tune_qchunk = emit_qchunk;
    ctx->tune_emit_fast = emit_fast != 0;
    ctx->tune_emit_kt = emit_fast >= 2 ? emit_fast : 0;
    ctx->train_group = train_group;
    return OPAL_OK;
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
    // fragments per draw: EMIT_QCHUNK; a batch that the grid's waves cover with one draw each (n <= 32 x #waves, 262k
    // fragments on a full grid) is dealt out evenly instead, one draw of ceil(n / #waves) per wave and no atomics at all
    // (with them a 200k-fragment launch cost 16k same-address atomics = 200 us, as much as the work)
    const uint64_t qmax = std::min<uint64_t>(ctx->tune_qchunk ? ctx->tune_qchunk : EMIT_QCHUNK, WAVE);
    auto draw_for = [&](int grid_blocks) -> uint64_t {
        const uint64_t waves = (uint64_t)grid_blocks * WPB;
        if (ctx->tune_qchunk || n > qmax * waves) return qmax;
        return (n + waves - 1) / waves;
    };
    // the emit kernels' counters and the leftover count start at zero (opal_device.h: words of d_queue); Q_BAD lies behind them
    OPAL_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_queue, 0, Q_BAD * sizeof(unsigned long long), ctx->stream));
    // the fast kernel reads the batch as dwords: it needs a 4-byte aligned buffer of at least two of them
    const int fast = ctx->d_fast && ctx->tune_emit_fast && (reinterpret_cast<uintptr_t>(d_reads) & 3u) == 0 && reads_bytes >= 8;
    int grid = 1;
    if (fast) {
        G.fast = ctx->d_fast;
        G.fast_canon2 = ctx->fast_canon2;
        static_assert(EF_MAXLEN == 512, "Geom::fast_maxlen's default");
        if (ctx->tune_emit_kt == 2 && ctx->fast_canon2 && (reinterpret_cast<uintptr_t>(d_out_idx) & 7u) == 0) G.fast_maxlen = 256;
        // P == 2 makes every fragment's feature count even: with an 8-byte aligned d_out_idx every fragment's range is too
        if (ctx->fast_canon2 && (reinterpret_cast<uintptr_t>(d_out_idx) & 7u) == 0) {
            // tune_emit_kt (experiment, opal_set_tuning emit_fast = 2 / 3): per-base term arrays for pattern 0, for fragments of up
            // to 256 bases (2) or EF_MAXLEN (3); what such a kernel does not take goes to the generic kernel
            const void *fn2 = ctx->tune_emit_kt == 2 ? (const void *)&emit_canon2_kernel<true, 320>
                            : ctx->tune_emit_kt == 3 ? (const void *)&emit_canon2_kernel<true, EF_ZLEN>
                            : (ctx->fast_pair == 1 && ctx->tune_emit_kt != 4) ? (const void *)&emit_canon2_kernel<false, EF_ZLEN, 1>
                            : (ctx->fast_pair == 2 && ctx->tune_emit_kt != 4) ? (const void *)&emit_canon2_kernel<false, EF_ZLEN, 2>
                                                                              : (const void *)&emit_canon2_kernel<false, EF_ZLEN>;
            rc = grid_for(ctx, fn2, n, &grid);
            if (rc) return rc;
            uint64_t qc = draw_for(grid);
            void *args[] = {&B, &G, (void *)&mask, &d_out_off, &d_out_idx, &out_cap, &ctx->d_queue, &qc};
            OPAL_HIP_CHECK(ctx, hipLaunchKernel(fn2, dim3(grid), dim3(BLOCK), args, 0, ctx->stream));
        } else if (ctx->fast_canonP) {
            const int pair = ctx->tune_emit_kt == 4 ? 0 : ctx->fast_pair;
#define OPAL_CP(N) (pair == 1 ? (const void *)&emit_canonP_kernel<N, 1> : pair == 2 ? (const void *)&emit_canonP_kernel<N, 2> : (const void *)&emit_canonP_kernel<N, 0>)
            const void *fnP = ctx->P == 3 ? OPAL_CP(3) : ctx->P == 4 ? OPAL_CP(4) : ctx->P == 5 ? OPAL_CP(5) : ctx->P == 6 ? OPAL_CP(6)
                            : ctx->P == 7 ? OPAL_CP(7) : ctx->P == 8 ? OPAL_CP(8) : OPAL_CP(9);
#undef OPAL_CP
            rc = grid_for(ctx, fnP, n, &grid);
            if (rc) return rc;
            uint64_t qc = draw_for(grid);
            void *args[] = {&B, &G, (void *)&mask, &d_out_off, &d_out_idx, &out_cap, &ctx->d_queue, &qc};
            OPAL_HIP_CHECK(ctx, hipLaunchKernel(fnP, dim3(grid), dim3(BLOCK), args, 0, ctx->stream));
        } else {
            rc = grid_for(ctx, (const void *)&emit_idx_kernel<true>, n, &grid);
            if (rc) return rc;
            hipLaunchKernelGGL(emit_idx_kernel<true>, dim3(grid), dim3(BLOCK), 0, ctx->stream, B, G, mask, d_out_off, d_out_idx, out_cap,
                               ctx->d_queue, draw_for(grid), 1);
        }
        OPAL_HIP_CHECK(ctx, hipGetLastError());
    }
    // every fragment the fast kernel does not take (text, long reads, other geometries); returns at once when there is none
    rc = grid_for(ctx, (const void *)&emit_idx_kernel<false>, n, &grid);
    if (rc) return rc;
    hipLaunchKernelGGL(emit_idx_kernel<false>, dim3(grid), dim3(BLOCK), 0, ctx->stream, B, G, mask, d_out_off, d_out_idx, out_cap,
                       ctx->d_queue, draw_for(grid), fast);
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
    hipLaunchKernelGGL(emit_names_kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, B, G, d_out_off, d_out_names, out_cap, ctx->d_queue);
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
    // cache policy of the row loads by table size (see row_load) unless opal_set_tuning forces one
    const bool nt = ctx->tune_nt < 0 ? ctx->n_slots * sizeof(float) >= NT_TABLE_BYTES : ctx->tune_nt != 0;
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
