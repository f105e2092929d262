// Host-side entry points of the C-ABI (opal_host.cpp) under AddressSanitizer + UBSan: built and run by
// tests/test_host_sanitizers.py with plain g++ (no GPU, no HIP runtime calls on these paths).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "opal_hip.h"

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static int pack_case(const std::vector<std::string> &seqs)
{
    std::string cat;
    std::vector<uint64_t> off{0};
    for (auto &s : seqs) { cat += s; off.push_back(cat.size()); }
    const uint64_t n = seqs.size();
    uint64_t need = 0;
    // exact-size heap buffers so the sanitizer sees any byte written past what the library reported
    std::vector<uint64_t> foff(n);
    std::vector<uint32_t> flen(n);
    REQUIRE(opal_pack_reads((const uint8_t *)cat.data(), off.data(), n, nullptr, 0, foff.data(), flen.data(), &need) == OPAL_OK);
    uint8_t *out = (uint8_t *)malloc(need ? need : 1);
    uint64_t wrote = 0;
    REQUIRE(opal_pack_reads((const uint8_t *)cat.data(), off.data(), n, out, need, foff.data(), flen.data(), &wrote) == OPAL_OK);
    REQUIRE(wrote == need);
    // unpack and compare
    for (uint64_t i = 0; i < n; i++) {
        const bool raw = (flen[i] & OPAL_RAW_FLAG) != 0;
        const uint32_t len = flen[i] & ~OPAL_RAW_FLAG;
        REQUIRE(len == seqs[i].size());
        for (uint32_t j = 0; j < len; j++) {
            char c;
            if (raw) c = (char)out[foff[i] + j];
            else { const uint64_t b = foff[i] + j; c = "ACGT"[(out[b >> 2] >> ((b & 3) * 2)) & 3]; }
            REQUIRE(c == seqs[i][j]);
        }
    }
    if (need > 0) {   // one byte short must be refused, not overrun
        uint8_t *small = (uint8_t *)malloc(need - 1 ? need - 1 : 1);
        REQUIRE(opal_pack_reads((const uint8_t *)cat.data(), off.data(), n, small, need - 1, foff.data(), flen.data(), &wrote) != OPAL_OK);
        free(small);
    }
    free(out);
    return 0;
}

int main()
{
    REQUIRE(opal_version() != nullptr);
    REQUIRE(pack_case({}) == 0);
    REQUIRE(pack_case({""}) == 0);
    REQUIRE(pack_case({"A", "", "ACGT", "ACGTA", "ACGTN", "acgt", std::string(1001, 'G'), "T"}) == 0);
    REQUIRE(pack_case({std::string(3, 'C'), std::string(7, 'N'), std::string(64, 'A')}) == 0);
    {   // null / inconsistent arguments come back as errors
        uint64_t need = 0;
        REQUIRE(opal_pack_reads(nullptr, nullptr, 1, nullptr, 0, nullptr, nullptr, &need) != OPAL_OK);
        REQUIRE(opal_last_error(nullptr)[0] != 0);
    }
    opal_rng *rng = nullptr;
    REQUIRE(opal_rng_create(&rng) == OPAL_OK);
    const uint32_t seed = 42;
    REQUIRE(opal_rng_seed(rng, &seed, 1) == OPAL_OK);
    uint64_t v = 0;
    REQUIRE(opal_rng_randbelow(rng, 10, &v) == OPAL_OK && v < 10);
    REQUIRE(opal_rng_randbelow(rng, 1, &v) == OPAL_OK && v == 0);
    REQUIRE(opal_rng_randbelow(rng, (1ull << 40) + 3, &v) == OPAL_OK && v < (1ull << 40) + 3);
    REQUIRE(opal_rng_randbelow(rng, 0, &v) != OPAL_OK);
    const uint32_t big[3] = {0xffffffffu, 0x12345678u, 7u};
    REQUIRE(opal_rng_seed(rng, big, 3) == OPAL_OK);
    REQUIRE(opal_rng_seed(rng, big, 0) != OPAL_OK);
    {   // sampler: exact-capacity buffers, then one short
        const std::vector<uint64_t> glen{1000, 150, 199, 200, 201, 5000};
        uint64_t n_out = 0;
        REQUIRE(opal_rng_seed(rng, &seed, 1) == OPAL_OK);
        (void)opal_sample_positions(rng, glen.data(), glen.size(), 200, 1.0, nullptr, nullptr, 0, &n_out);
        REQUIRE(n_out == 5 + 1 + 2 + 25);
        std::vector<uint64_t> pos(n_out);
        std::vector<uint32_t> gi(n_out);
        REQUIRE(opal_rng_seed(rng, &seed, 1) == OPAL_OK);
        uint64_t n2 = 0;
        REQUIRE(opal_sample_positions(rng, glen.data(), glen.size(), 200, 1.0, pos.data(), gi.data(), n_out, &n2) == OPAL_OK);
        REQUIRE(n2 == n_out);
        for (uint64_t i = 0; i < n2; i++) REQUIRE(gi[i] < glen.size() && pos[i] + 200 <= glen[gi[i]]);
        std::vector<uint64_t> pos_s(n_out - 1);
        std::vector<uint32_t> gi_s(n_out - 1);
        REQUIRE(opal_rng_seed(rng, &seed, 1) == OPAL_OK);
        REQUIRE(opal_sample_positions(rng, glen.data(), glen.size(), 200, 1.0, pos_s.data(), gi_s.data(), n_out - 1, &n2) == OPAL_ERANGE);
        REQUIRE(n2 == n_out);
        REQUIRE(opal_sample_positions(rng, glen.data(), glen.size(), 0, 1.0, pos.data(), gi.data(), n_out, &n2) != OPAL_OK);
    }
    {
        std::vector<uint64_t> perm(1000);
        for (uint64_t i = 0; i < perm.size(); i++) perm[i] = i;
        REQUIRE(opal_rng_shuffle(rng, perm.data(), perm.size()) == OPAL_OK);
        std::vector<char> seen(perm.size(), 0);
        for (uint64_t x : perm) { REQUIRE(x < perm.size() && !seen[x]); seen[x] = 1; }
        REQUIRE(opal_rng_shuffle(rng, perm.data(), 0) == OPAL_OK);
        REQUIRE(opal_rng_shuffle(rng, perm.data(), 1) == OPAL_OK);
    }
    REQUIRE(opal_rng_destroy(rng) == OPAL_OK);
    REQUIRE(opal_rng_destroy(nullptr) == OPAL_OK);
    printf("host sanitizer driver: ok\n");
    return 0;
}
