// Host-side half of libopal_hip.so: error plumbing, read packing, and the CPython-exact fragment sampler.
// Nothing here touches the GPU.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "opal_internal.h"

namespace opal {
static thread_local std::string g_err;
int fail(opal_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    g_err = msg;
    return code;
}
}  // namespace opal

// OPAL_SRC_HASH: sha256 (first 16 hex digits) over the library's sources and headers, passed by __graft_entry__.build();
// tests compare it with the tree so that a library built from other sources than the ones beside it is noticed
#ifndef OPAL_SRC_HASH
#define OPAL_SRC_HASH "unknown"
#endif
extern "C" const char *opal_version(void) { return "opal_hip 0.2 (gfx950) src " OPAL_SRC_HASH; }

extern "C" const char *opal_last_error(const opal_ctx *ctx) { return ctx ? ctx->err.c_str() : opal::g_err.c_str(); }

// ------------------------------------------------------------------------------------------------ packing
// Fragment batch layout: see include/opal_hip.h.  A sequence made only of upper-case A/C/G/T is 2-bit packed;
// anything else (lower case, N, IUPAC, digits, '\r' ...) is kept as raw bytes because the reference treats
// sequences as plain text (util/fasta2skm.py:137-141 gathers characters, not bases).
static inline int base_code(uint8_t c)
{
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
    }
}

extern "C" int opal_pack_reads(const uint8_t *seqs, const uint64_t *seq_off, uint64_t n, uint8_t *out, uint64_t out_cap,
                               uint64_t *frag_off, uint32_t *frag_len, uint64_t *out_bytes)
{
    if ((!seqs && n) || !seq_off) return opal::fail(nullptr, OPAL_EINVAL, "opal_pack_reads: null input");
    uint64_t o = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (seq_off[i + 1] < seq_off[i]) return opal::fail(nullptr, OPAL_EINVAL, "opal_pack_reads: seq_off not monotone");
        const uint64_t len = seq_off[i + 1] - seq_off[i];
        if (len >= OPAL_RAW_FLAG) return opal::fail(nullptr, OPAL_EINVAL, "opal_pack_reads: sequence longer than 2^31-1");
        const uint8_t *s = seqs + seq_off[i];
        bool acgt = true;
        for (uint64_t j = 0; j < len; j++)
            if (base_code(s[j]) < 0) { acgt = false; break; }
        const uint64_t nbytes = acgt ? (len + 3) / 4 : len;
        if (frag_off) frag_off[i] = acgt ? o * 4 : o;   // packed fragments are addressed in bases, raw ones in bytes
        if (frag_len) frag_len[i] = (uint32_t)len | (acgt ? 0u : OPAL_RAW_FLAG);
        if (out) {
            if (o + nbytes > out_cap) return opal::fail(nullptr, OPAL_ERANGE, "opal_pack_reads: output buffer too small");
            if (acgt) {
                for (uint64_t b = 0; b < nbytes; b++) {
                    uint32_t v = 0;
                    for (uint64_t j = 0; j < 4 && 4 * b + j < len; j++) v |= (uint32_t)base_code(s[4 * b + j]) << (2 * j);
                    out[o + b] = (uint8_t)v;
                }
            } else if (len) {
                memcpy(out + o, s, len);
            }
        }
        o += nbytes;
    }
    if (out_bytes) *out_bytes = o;
    return OPAL_OK;
}

// ------------------------------------------------------------------------------------------------ CPython random
// MT19937 exactly as CPython seeds and consumes it: random.seed(int) -> init_by_array(limbs of |seed|),
// randint(0,n-1) -> _randbelow_with_getrandbits(n) = getrandbits(bit_length(n)) with rejection,
// getrandbits(k<=32) = genrand >> (32-k), k>32 = little-endian words with the top one shifted.
#if defined(__HIP_DEVICE_COMPILE__)
#define OPAL_HOST_CLONES                      // hipcc also parses this file for the device, where there is nothing to clone
#else
#define OPAL_HOST_CLONES __attribute__((target_clones("avx2", "default")))   // picked at load time by the host CPU
#endif
struct opal_rng {
    static constexpr int N = 624, M = 397;
    uint32_t s[N];
    int idx = N;

    void init_genrand(uint32_t seed)
    {
        s[0] = seed;
        for (int i = 1; i < N; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
        idx = N;
    }
    void init_by_array(const uint32_t *key, int klen)
    {
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int c = (N > klen ? N : klen); c > 0; c--) {
            s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i >= N) { s[0] = s[N - 1]; i = 1; }
            if (++j >= klen) j = 0;
        }
        for (int c = N - 1; c > 0; c--) {
            s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= N) { s[0] = s[N - 1]; i = 1; }
        }
        s[0] = 0x80000000u;
    }
    // One twist of the whole state followed by the tempering of all 624 outputs into out[]: both loops are free of
    // loop-carried dependencies inside their three segments (i+M reads the previous state for i < N-M and the new one
    // after), so the compiler vectorises them; the consumers below then only index out[].
    uint32_t out[N];
    OPAL_HOST_CLONES void refill()
    {
        auto tw = [](uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((0u - (v & 1u)) & 0x9908b0dfu); };
        for (int i = 0; i < N - M; i++) s[i] = s[i + M] ^ tw(s[i], s[i + 1]);
        for (int i = N - M; i < 2 * (N - M); i++) s[i] = s[i + M - N] ^ tw(s[i], s[i + 1]);
        for (int i = 2 * (N - M); i < N - 1; i++) s[i] = s[i + M - N] ^ tw(s[i], s[i + 1]);
        s[N - 1] = s[M - 1] ^ tw(s[N - 1], s[0]);
        for (int i = 0; i < N; i++) {
            uint32_t y = s[i];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out[i] = y;
        }
        idx = 0;
    }
    uint32_t next()
    {
        if (idx >= N) refill();
        return out[idx++];
    }
    uint64_t getrandbits(int k)
    {
        if (k <= 32) return next() >> (32 - k);
        uint64_t lo = next();
        uint64_t hi = next();
        if (k < 64) hi >>= (64 - k);
        return lo | (hi << 32);
    }
    uint64_t randbelow(uint64_t n)
    {
        const int k = 64 - __builtin_clzll(n);
        uint64_t r = getrandbits(k);
        while (r >= n) r = getrandbits(k);
        return r;
    }
};

extern "C" int opal_rng_create(opal_rng **out)
{
    if (!out) return opal::fail(nullptr, OPAL_EINVAL, "opal_rng_create: null out");
    opal_rng *r = new (std::nothrow) opal_rng;
    if (!r) return opal::fail(nullptr, OPAL_ENOMEM, "opal_rng_create: out of memory");
    r->init_genrand(5489u);
    *out = r;
    return OPAL_OK;
}
extern "C" int opal_rng_destroy(opal_rng *rng) { delete rng; return OPAL_OK; }

extern "C" int opal_rng_seed(opal_rng *rng, const uint32_t *limbs, int n_limbs)
{
    if (!rng || !limbs || n_limbs < 1) return opal::fail(nullptr, OPAL_EINVAL, "opal_rng_seed: bad arguments");
    rng->init_by_array(limbs, n_limbs);
    return OPAL_OK;
}

extern "C" int opal_rng_randbelow(opal_rng *rng, uint64_t n, uint64_t *out)
{
    if (!rng || !out || n == 0) return opal::fail(nullptr, OPAL_EINVAL, "opal_rng_randbelow: bad arguments");
    *out = rng->randbelow(n);
    return OPAL_OK;
}

extern "C" int opal_sample_positions(opal_rng *rng, const uint64_t *genome_len, uint64_t n_genomes, uint64_t frag_len,
                                     double coverage, uint64_t *out_pos, uint32_t *out_genome, uint64_t cap, uint64_t *n_out)
{
    if (!rng || (!genome_len && n_genomes) || !n_out || frag_len == 0)
        return opal::fail(nullptr, OPAL_EINVAL, "opal_sample_positions: bad arguments");
    if (cap && (!out_pos || !out_genome)) return opal::fail(nullptr, OPAL_EINVAL, "opal_sample_positions: null outputs");
    uint64_t n = 0;
    for (uint64_t g = 0; g < n_genomes; g++) {
        const uint64_t L = genome_len[g];
        if (L < frag_len) continue;                       // drawfrag.py:32-33, RNG untouched
        const double want = coverage * (double)L;         // drawfrag.py:31
        const uint64_t range = L - frag_len + 1;          // random.randint(0, len-k), drawfrag.py:38
        const int kbits = 64 - __builtin_clzll(range);
        // the loop below runs `while coverage < c*len` (drawfrag.py:36) with coverage += frag_len per draw and the
        // safety break `try_num > 10*len` (drawfrag.py:48-49): the number of draws is known in advance
        uint64_t want_n = 0;                              // = number of t >= 0 with (double)(t*frag_len) < want
        if (want > 0.0) {
            const double q = want / (double)frag_len;
            if (q > (double)(10 * L + 1)) want_n = 10 * L + 1;
            else {
                want_n = (uint64_t)q;
                while ((double)(want_n * frag_len) < want) want_n++;
                while (want_n > 0 && !((double)((want_n - 1) * frag_len) < want)) want_n--;
                if (want_n > 10 * L + 1) want_n = 10 * L + 1;
            }
        }
        if (kbits <= 32 && n + want_n <= cap) {
            // _randbelow: getrandbits(kbits) = genrand >> (32-kbits), redrawn while >= range.  Branch-free over the
            // tempered block: every output is stored, the write index only advances past accepted ones.
            const int shift = 32 - kbits;
            uint64_t got = 0;
            while (got < want_n) {
                if (rng->idx >= opal_rng::N) rng->refill();
                int i = rng->idx;
                const uint64_t room = want_n - got;
                // at most `room` outputs can be accepted from min(room, rest of the block) candidates
                const int end = (int)std::min<uint64_t>((uint64_t)opal_rng::N, (uint64_t)i + room);
                uint64_t w = n + got;
                for (; i < end; i++) {
                    const uint64_t r = rng->out[i] >> shift;
                    out_pos[w] = r;                       // slot w <= n + want_n - 1 < cap (overwritten if rejected)
                    out_genome[w] = (uint32_t)g;
                    w += r < range ? 1u : 0u;
                }
                got = w - n;
                rng->idx = i;
            }
            n += want_n;
        } else {
            for (uint64_t t = 0; t < want_n; t++) {
                const uint64_t p = rng->randbelow(range);
                if (n < cap) { out_pos[n] = p; out_genome[n] = (uint32_t)g; }
                n++;
            }
        }
    }
    *n_out = n;
    if (n > cap) return opal::fail(nullptr, OPAL_ERANGE, "opal_sample_positions: more fragments than cap");
    return OPAL_OK;
}

extern "C" int opal_rng_shuffle(opal_rng *rng, uint64_t *perm, uint64_t n)
{
    if (!rng || (!perm && n)) return opal::fail(nullptr, OPAL_EINVAL, "opal_rng_shuffle: bad arguments");
    // Lib/random.py shuffle: for i in reversed(range(1, n)): j = _randbelow(i+1); swap.  The draws do not depend on
    // the array, so a block of them is made first and its targets prefetched (the swaps of a 10^7-entry array are
    // cache misses), then the swaps run in the reference's order.
    constexpr uint64_t B = 64;
    uint64_t jbuf[B];
    uint64_t i = n;
    while (i > 1) {
        const uint64_t m = std::min<uint64_t>(B, i - 1);
        for (uint64_t t = 0; t < m; t++) {
            jbuf[t] = rng->randbelow(i - t);              // step i-1-t draws from [0, i-1-t]
            __builtin_prefetch(&perm[jbuf[t]], 1);
        }
        for (uint64_t t = 0; t < m; t++) {
            const uint64_t a = i - 1 - t, j = jbuf[t];
            const uint64_t tmp = perm[a];
            perm[a] = perm[j];
            perm[j] = tmp;
        }
        i -= m;
    }
    return OPAL_OK;
}
