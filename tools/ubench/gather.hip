// Micro-benchmark: what does one wave-wide SCATTERED load instruction cost a CU, as a function of the bytes per lane,
// the number of active lanes and their placement?  (decides table entry widths / probe batching in tk_flat_kernel)
//   hipcc --offload-arch=gfx950 -O3 -o gather tools/ubench/gather.hip && ./gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int W>
struct Vec;
template <> struct Vec<4> { typedef uint32_t T; };
template <> struct Vec<8> { typedef uint2 T; };
template <> struct Vec<16> { typedef uint4 T; };

__device__ inline uint32_t fold(uint32_t x) { return x; }
__device__ inline uint32_t fold(uint2 x) { return x.x ^ x.y; }
__device__ inline uint32_t fold(uint4 x) { return x.x ^ x.y ^ x.z ^ x.w; }

// every wave: `iters` rounds of ILP independent gathers; lane active iff (active_mask >> lane) & 1
template <int W, int ILP>
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t* tab, uint32_t mask_entries, uint64_t active, int iters, uint32_t* out) {
    typedef typename Vec<W>::T T;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    if ((active >> lane) & 1) {
        for (int i = 0; i < iters; ++i) {
            T v[ILP];
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                h = h * 1664525u + 1013904223u;
                const uint32_t idx = (h >> 8) & mask_entries;
                v[k] = *(const T*)(tab + (size_t)idx * W);
            }
#pragma unroll
            for (int k = 0; k < ILP; ++k) acc ^= fold(v[k]);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int W, int ILP>
static double run(const uint8_t* tab, size_t tab_bytes, uint64_t active, uint32_t* out, int blocks, int iters) {
    const uint32_t mask_entries = (uint32_t)(tab_bytes / W) - 1;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    gather_kernel<W, ILP><<<blocks, 256>>>(tab, mask_entries, active, iters, out);
    hipEventRecord(a);
    gather_kernel<W, ILP><<<blocks, 256>>>(tab, mask_entries, active, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    const size_t sizes[2] = {4u << 20, 64u << 20};
    uint8_t* tab;
    uint32_t* out;
    hipMalloc(&tab, sizes[1]);
    hipMemset(tab, 1, sizes[1]);
    hipMalloc(&out, 64);
    const int blocks = cus * 8, iters = 256;      // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
    struct { const char* name; uint64_t m; } masks[] = {
        {"64 lanes", ~0ull}, {"32 low", 0xFFFFFFFFull}, {"32 even", 0x5555555555555555ull}, {"16 (every 4th)", 0x1111111111111111ull},
        {"16 low", 0xFFFFull}, {"8 (every 8th)", 0x0101010101010101ull}, {"1 lane", 1ull}};
    printf("CUs %d, clock %.2f GHz; cycles per wave-instruction per CU (all 32 waves of a CU issuing)\n", cus, ghz);
    for (int s = 0; s < 2; ++s) {
        printf("table %zu MB\n", sizes[s] >> 20);
        for (auto& mk : masks) {
            const double n_instr_per_cu = (double)blocks * 4 * iters * 4 / cus;   // ILP 4
            double t4 = run<4, 4>(tab, sizes[s], mk.m, out, blocks, iters);
            double t8 = run<8, 4>(tab, sizes[s], mk.m, out, blocks, iters);
            double t16 = run<16, 4>(tab, sizes[s], mk.m, out, blocks, iters);
            printf("  %-16s  4B %7.1f   8B %7.1f   16B %7.1f\n", mk.name, t4 * 1e-3 * ghz * 1e9 / n_instr_per_cu,
                   t8 * 1e-3 * ghz * 1e9 / n_instr_per_cu, t16 * 1e-3 * ghz * 1e9 / n_instr_per_cu);
        }
    }
    // dependent chain (ILP 1) for latency-bound comparison at 8 waves/SIMD
    {
        const double n_instr_per_cu = (double)blocks * 4 * iters / cus;
        double t = run<16, 1>(tab, sizes[0], ~0ull, out, blocks, iters);
        printf("ILP 1, 16B, 64 lanes, 4 MB: %.1f cycles per wave-instruction per CU\n", t * 1e-3 * ghz * 1e9 / n_instr_per_cu);
    }
    return 0;
}
