// Micro-benchmark 3: issue cost of the VALU instructions tk_flat_kernel leans on, in cycles per wave-instruction per
// SIMD (4 = full rate).   hipcc --offload-arch=gfx950 -O3 -o valu tools/ubench/valu.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                                         \
    __global__ __launch_bounds__(256) void name(uint32_t* out, int iters, uint32_t s) {           \
        uint32_t a = threadIdx.x * 2654435761u + s, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 99u; \
        uint32_t e = a * 3u, f = b * 5u, g = c * 7u, h = d * 9u;                                   \
        uint64_t A = a, B = b, C = c, D = d;                                                       \
        for (int i = 0; i < iters; ++i) { REP8(body) }                                             \
        if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)(A ^ B ^ C ^ D)) == 0x12345u) out[0] = a;   \
    }

KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mul24, asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 1, %4\n v_lshl_add_u64 %1, %1, 1, %4\n v_lshl_add_u64 %2, %2, 1, %4\n v_lshl_add_u64 %3, %3, 1, %4" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(A));)
KERNEL(k_lshlrev_b64, asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D));)
KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %0, 13\n v_alignbit_b32 %1, %1, %1, 13\n v_alignbit_b32 %2, %2, %2, 13\n v_alignbit_b32 %3, %3, %3, 13" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_bitop3, asm volatile("v_bitop3_b32 %0, %0, %4, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %2 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %0 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_dpp_add, asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_wave_shr, asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_bcnt, asm volatile("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mbcnt, asm volatile("v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_hi_u32_b32 %1, %4, %1\n v_mbcnt_lo_u32_b32 %2, %4, %2\n v_mbcnt_hi_u32_b32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_cmp_cndmask, asm volatile("v_cmp_gt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");)
KERNEL(k_readlane, asm volatile("v_readlane_b32 s20, %0, 3\n v_add_u32 %1, %1, s20\n v_readlane_b32 s21, %2, 5\n v_add_u32 %3, %3, s21" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");)
KERNEL(k_ffbl, asm volatile("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %4, %1\n v_perm_b32 %1, %1, %4, %2\n v_perm_b32 %2, %2, %4, %3\n v_perm_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_sdwa_xor, asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %2, %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %3, %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
KERNEL(k_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(e), "v"(f) : "vcc");)

typedef void (*kern_t)(uint32_t*, int, uint32_t);
static void run(const char* name, kern_t k, int waves_per_simd, int cus, double ghz, uint32_t* out, int per_rep) {
    const int iters = 4096;
    const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = one per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double n = (double)iters * 8 * per_rep * waves_per_simd;   // wave-instructions per SIMD
    printf("  %-16s %d waves/SIMD: %6.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, ms * 1e-3 * ghz * 1e9 / n);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    uint32_t* out;
    hipMalloc(&out, 64);
    printf("clock %.2f GHz (nominal; the counts below assume it)\n", ghz);
    for (int w = 1; w <= 4; w *= 4) {
        run("v_add_u32", k_add, w, cus, ghz, out, 4);
        run("v_mul_lo_u32", k_mul_lo, w, cus, ghz, out, 4);
        run("v_mul_u32_u24", k_mul24, w, cus, ghz, out, 4);
        run("v_mad_u32_u24", k_mad24, w, cus, ghz, out, 4);
        run("v_mad_u64_u32", k_mad_u64_u32, w, cus, ghz, out, 4);
        run("v_lshl_add_u64", k_lshl_add_u64, w, cus, ghz, out, 4);
        run("v_lshlrev_b64", k_lshlrev_b64, w, cus, ghz, out, 4);
        run("v_alignbit_b32", k_alignbit, w, cus, ghz, out, 4);
        run("v_bitop3_b32", k_bitop3, w, cus, ghz, out, 4);
        run("v_perm_b32", k_perm, w, cus, ghz, out, 4);
        run("v_xor sdwa", k_sdwa_xor, w, cus, ghz, out, 4);
        run("v_add dpp row_shr", k_dpp_add, w, cus, ghz, out, 4);
        run("v_mov dpp wave_shr", k_wave_shr, w, cus, ghz, out, 4);
        run("v_bcnt", k_bcnt, w, cus, ghz, out, 4);
        run("v_mbcnt", k_mbcnt, w, cus, ghz, out, 4);
        run("v_ffbl", k_ffbl, w, cus, ghz, out, 4);
        run("cmp+cndmask", k_cmp_cndmask, w, cus, ghz, out, 4);
        run("readlane+add", k_readlane, w, cus, ghz, out, 4);
    }
    return 0;
}
