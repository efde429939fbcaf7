// Micro-benchmark 2: scattered 16-byte loads -- cache-policy variants and table sizes (L1-resident .. beyond L2),
// and the same gather from LDS.   hipcc --offload-arch=gfx950 -O3 -o gather2 tools/ubench/gather2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

enum { PLAIN = 0, NT = 1, SC0 = 2, SC1 = 3, SC0SC1 = 4, SC0SC1NT = 5, NVARIANT = 6 };
static const char* vname[NVARIANT] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt"};

template <int V>
__device__ inline uint4 ld16(const uint8_t* p) {
    uint4 v;
    if (V == PLAIN) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (V == NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (V == SC0) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (V == SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (V == SC0SC1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (V == SC0SC1NT) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int V>
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t* tab, uint32_t mask_entries, int iters, uint32_t* out) {
    uint32_t h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            v[k] = ld16<V>(tab + (size_t)((h >> 8) & mask_entries) * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the same from LDS: a 64 KB table per workgroup of 1024 threads
__global__ __launch_bounds__(1024) void lds_gather_kernel(const uint8_t* tab, int iters, uint32_t* out) {
    __shared__ uint4 t[4096];
    for (int i = threadIdx.x; i < 4096; i += 1024) t[i] = ((const uint4*)tab)[i];
    __syncthreads();
    uint32_t h = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            uint4 v = t[(h >> 8) & 4095];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int V>
static double run(const uint8_t* tab, size_t bytes, uint32_t* out, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    gather_kernel<V><<<blocks, 256>>>(tab, (uint32_t)(bytes / 16) - 1, iters, out);
    hipEventRecord(a);
    gather_kernel<V><<<blocks, 256>>>(tab, (uint32_t)(bytes / 16) - 1, iters, out);
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
    uint8_t* tab;
    uint32_t* out;
    hipMalloc(&tab, 64u << 20);
    hipMemset(tab, 1, 64u << 20);
    hipMalloc(&out, 64);
    const int blocks = cus * 8, iters = 256;
    const double lanes_per_cu = (double)blocks * 256 * iters * 4 / cus;
    const size_t sizes[] = {16u << 10, 256u << 10, 1u << 20, 4u << 20, 16u << 20, 64u << 20};
    printf("cycles per ACTIVE LANE per CU, 16-byte scattered loads, 32 waves per CU, %d CUs, %.2f GHz\n", cus, ghz);
    printf("%-10s", "table");
    for (int v = 0; v < NVARIANT; ++v) printf(" %11s", vname[v]);
    printf("\n");
    for (size_t s : sizes) {
        double t[NVARIANT];
        t[0] = run<PLAIN>(tab, s, out, blocks, iters); t[1] = run<NT>(tab, s, out, blocks, iters); t[2] = run<SC0>(tab, s, out, blocks, iters);
        t[3] = run<SC1>(tab, s, out, blocks, iters); t[4] = run<SC0SC1>(tab, s, out, blocks, iters); t[5] = run<SC0SC1NT>(tab, s, out, blocks, iters);
        printf("%6zu KB ", s >> 10);
        for (int v = 0; v < NVARIANT; ++v) printf(" %11.2f", t[v] * 1e-3 * ghz * 1e9 / lanes_per_cu);
        printf("\n");
    }
    {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        lds_gather_kernel<<<cus * 2, 1024>>>(tab, iters, out);
        hipEventRecord(a);
        lds_gather_kernel<<<cus * 2, 1024>>>(tab, iters, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("LDS 64 KB table, 16-byte random reads: %.2f cycles per lane per CU (incl. the table fill)\n",
               ms * 1e-3 * ghz * 1e9 / ((double)cus * 2 * 1024 * iters * 4 / cus));
    }
    return 0;
}
