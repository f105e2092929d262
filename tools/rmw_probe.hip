// Hardware probe (not part of the product): what rate do random K-float rows of an 8 GiB table sustain for
// read-only, write-only and read-modify-write access, as a function of rows in flight per wave and waves per CU?
// Used to decide whether the learner's sweeps 2/3 (opal_train.hip) are bound by HBM or by latency/occupancy.
//   hipcc --offload-arch=gfx950 -O3 -o tools/rmw_probe tools/rmw_probe.hip && tools/rmw_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

// MODE 0 read, 1 write, 2 rmw (one table), 3 read table A + rmw table B (the shape of sweep 3),
// 5 = read with non-temporal loads (what the predict kernel uses), 6 = rmw with nt loads and nt stores, 7 = rmw with
// plain loads and nt stores,
// 4 = rmw that also rewrites (unchanged) the bytes of the row's first and last 128-B line that lie outside the
// row, so that every line it dirties is dirtied completely
template <int MODE, int U>
__global__ __launch_bounds__(256) void probe(float *__restrict__ a, float *__restrict__ b, uint32_t mask, uint32_t K, uint32_t rows_per_wave,
                                             float *__restrict__ sink)
{
    extern __shared__ uint8_t pad[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + threadIdx.x / 64;
    const uint32_t nv = (K / 4 + 63) / 64;          // float4 per lane per row (K % 4 == 0)
    float4 acc = make_float4(0, 0, 0, 0);
    for (uint32_t r0 = 0; r0 < rows_per_wave; r0 += U) {
        uint32_t row[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            row[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)((mix(wave * 0x9e3779b9u + r0 + u) * K) & mask));
        if (MODE == 4) {
            float pv[U];
            size_t pa[U];
            bool on[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t head = row[u] & 31u, end = row[u] + K, tail = (32u - (end & 31u)) & 31u;
                on[u] = lane < 32 ? (uint32_t)lane < head : (uint32_t)(lane - 32) < tail;
                pa[u] = lane < 32 ? (size_t)(row[u] - head + lane) : (size_t)end + (lane - 32);
                if (on[u]) pv[u] = a[pa[u]];
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (on[u]) a[pa[u]] = pv[u];
        }
        for (uint32_t j = 0; j < nv; j++) {
            const uint32_t c = min((j * 64 + lane) * 4, K - 4);
            float4 va[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4) va[u] = *reinterpret_cast<const float4 *>(a + (size_t)row[u] + c);
                if (MODE == 7) va[u] = *reinterpret_cast<const float4 *>(a + (size_t)row[u] + c);
                if (MODE == 5 || MODE == 6) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(a + (size_t)row[u] + c));
                    va[u] = make_float4(t.x, t.y, t.z, t.w);
                }
                if (MODE == 3) vb[u] = *reinterpret_cast<const float4 *>(b + (size_t)row[u] + c);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (MODE == 0 || MODE == 5) { acc.x += va[u].x; acc.y += va[u].y; acc.z += va[u].z; acc.w += va[u].w; }
                if (MODE == 1) *reinterpret_cast<float4 *>(a + (size_t)row[u] + c) = make_float4(1.f, 2.f, 3.f, (float)r0);
                if (MODE == 2 || MODE == 4) { va[u].x += 1.f; va[u].y += 1.f; va[u].z += 1.f; va[u].w += 1.f; *reinterpret_cast<float4 *>(a + (size_t)row[u] + c) = va[u]; }
                if (MODE == 6 || MODE == 7) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    f4 t; t.x = va[u].x + 1.f; t.y = va[u].y + 1.f; t.z = va[u].z + 1.f; t.w = va[u].w + 1.f;
                    __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(a + (size_t)row[u] + c));
                }
                if (MODE == 8) {   // integer atomic adds without return (deterministic accumulation: order does not matter)
                    int *ia = reinterpret_cast<int *>(a + (size_t)row[u] + c);
                    atomicAdd(ia, 1); atomicAdd(ia + 1, 2); atomicAdd(ia + 2, 3); atomicAdd(ia + 3, (int)r0);
                }
                if (MODE == 9) {   // float atomic adds without return
                    float *fa = a + (size_t)row[u] + c;
                    atomicAdd(fa, 1.f); atomicAdd(fa + 1, 2.f); atomicAdd(fa + 2, 3.f); atomicAdd(fa + 3, 4.f);
                }
                if (MODE == 3) { vb[u].x += va[u].x; vb[u].y += va[u].y; vb[u].z += va[u].z; vb[u].w += va[u].w; *reinterpret_cast<float4 *>(b + (size_t)row[u] + c) = vb[u]; }
            }
        }
    }
    if ((MODE == 0 || MODE == 5) && acc.x + acc.y + acc.z + acc.w == 1234.5678f) sink[0] = acc.x;
    if (pad && threadIdx.x == 100000) pad[0] = 1;
}

template <int MODE, int U>
static void run(float *a, float *b, float *sink, uint32_t mask, uint32_t K, int blocks_per_cu, int cus)
{
    // occupancy is set through dynamic LDS: 160 KB per CU / blocks_per_cu
    const size_t lds = (160 * 1024) / blocks_per_cu - 512;
    CK(hipFuncSetAttribute((const void *)&probe<MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t rows_per_wave = 2048;
    const int grid = cus * blocks_per_cu * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<MODE, U>), dim3(grid), dim3(256), lds, 0, a, b, mask, K, rows_per_wave / 4, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<MODE, U>), dim3(grid), dim3(256), lds, 0, a, b, mask, K, rows_per_wave, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double rows = (double)grid * 4 * rows_per_wave;
    const double units = (MODE == 0 || MODE == 1 || MODE == 5) ? 1 : (MODE == 3 ? 3 : 2);   // atomics: a read and a write at the L2
    const double bytes = rows * K * 4.0 * units;
    printf("mode=%s U=%d waves/CU=%2d K=%u  %.2f ms  %.2f TB/s (%.0f row-units)\n",
           MODE == 0 ? "read " : MODE == 1 ? "write" : MODE == 2 ? "rmw  " : MODE == 3 ? "r+rmw" : MODE == 4 ? "rmw+p" : MODE == 5 ? "rd nt" : MODE == 6 ? "rmwNN" : MODE == 7 ? "rmw-N" : MODE == 8 ? "atomI" : "atomF", U, blocks_per_cu * 4, K, ms, bytes / ms / 1e9, units);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char **argv)
{
    const uint32_t K = argc > 1 ? (uint32_t)atoi(argv[1]) : 356;
    const int bits = argc > 2 ? atoi(argv[2]) : 31;
    if (K % 4 || K < 4 || K > 2048 || bits < 20 || bits > 31) { fprintf(stderr, "usage: rmw_probe [K%%4==0 <= 2048] [bits 20..31]\n"); return 2; }
    const uint32_t mask = (uint32_t)((1ull << bits) - 1);
    const size_t n = (1ull << bits) + K + 64;
    float *a, *b, *sink;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&sink, 256));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs\n", p.name, cus);
    for (int bpc : {2, 4, 8}) {
        run<0, 4>(a, b, sink, mask, K, bpc, cus);
        run<0, 8>(a, b, sink, mask, K, bpc, cus);
        run<5, 8>(a, b, sink, mask, K, bpc, cus);
        run<1, 4>(a, b, sink, mask, K, bpc, cus);
        run<2, 1>(a, b, sink, mask, K, bpc, cus);
        run<2, 4>(a, b, sink, mask, K, bpc, cus);
        run<2, 8>(a, b, sink, mask, K, bpc, cus);
        run<4, 4>(a, b, sink, mask, K, bpc, cus);
        run<6, 4>(a, b, sink, mask, K, bpc, cus);
        run<7, 4>(a, b, sink, mask, K, bpc, cus);
        run<8, 1>(a, b, sink, mask, K, bpc, cus);
        run<8, 4>(a, b, sink, mask, K, bpc, cus);
        run<9, 4>(a, b, sink, mask, K, bpc, cus);
        run<3, 2>(a, b, sink, mask, K, bpc, cus);
        run<3, 4>(a, b, sink, mask, K, bpc, cus);
        run<3, 8>(a, b, sink, mask, K, bpc, cus);
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(sink));
    return 0;
}
