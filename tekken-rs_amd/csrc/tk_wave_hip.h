// tk_wave_hip.h -- the wave primitives of tk_encode_impl.h on gfx950 (wave64).
#ifndef TK_WAVE_HIP_H
#define TK_WAVE_HIP_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TK_DEV __device__ __forceinline__

TK_DEV int wv_lane() { return (int)(threadIdx.x & 63u); }

TK_DEV uint64_t wv_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }  // the compare mask itself, no VGPR round trip

// value of `v` in lane `src` (0..63); every lane of the wave must execute it
TK_DEV uint32_t wv_shfl(uint32_t v, int src) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v);
}

// value of `v` in the first active lane, as a wave-uniform (scalar) value
TK_DEV uint32_t wv_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
TK_DEV uint64_t wv_first64(uint64_t v) {
    return ((uint64_t)wv_first((uint32_t)(v >> 32)) << 32) | (uint64_t)wv_first((uint32_t)v);
}

// value of `v` in lane `l`, l wave-uniform (v_readlane_b32: no LDS crossbar, result is scalar)
TK_DEV uint32_t wv_readlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// minimum of `v` over all 64 lanes as a scalar: DPP row shifts inside the 4 rows of 16 lanes, then the
// two row broadcasts; lane 63 ends up with the minimum (the classic GFX9 wave reduction, VALU only)
TK_DEV uint32_t wv_min_u32(uint32_t v) {
    uint32_t x = v, y;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x111, 0xF, 0xF, false); x = y < x ? y : x;  // row_shr:1
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x112, 0xF, 0xF, false); x = y < x ? y : x;  // row_shr:2
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x114, 0xF, 0xF, false); x = y < x ? y : x;  // row_shr:4
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x118, 0xF, 0xF, false); x = y < x ? y : x;  // row_shr:8
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x142, 0xA, 0xF, false); x = y < x ? y : x;  // row_bcast:15
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x143, 0xC, 0xF, false); x = y < x ? y : x;  // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// lane i receives lane i+1's value, lane 63 receives 0   (DPP wave_shl:1)
TK_DEV uint32_t wv_up1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false);
}

// lane i receives lane i-1's value, lane 0 receives 0     (DPP wave_shr:1)
TK_DEV uint32_t wv_dn1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);
}

// orders this wave's earlier global stores before its later loads: the stores must have been
// acknowledged (s_waitcnt vmcnt(0)) before a lane reads what another lane of the wave wrote
TK_DEV void wv_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// executed by ONE lane (inside a lane-predicated block whose result feeds no wave primitive)
TK_DEV uint64_t wv_brev64(uint64_t x) { return __builtin_bitreverse64(x); }  // s_brev_b64 on uniform values

TK_DEV uint32_t wv_atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
TK_DEV uint32_t wv_atomic_exch(uint32_t* p, uint32_t v) { return atomicExch(p, v); }
TK_DEV uint32_t wv_atomic_max(uint32_t* p, uint32_t v) { return atomicMax(p, v); }
// a word other CUs (other XCDs: other L2s) update with device-scope atomics during this kernel: read it where they land
TK_DEV uint32_t wv_load_coherent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// executed by ALL 64 lanes in uniform control flow: *p += 64*v, lane i receives old + i*v
TK_DEV uint32_t wv_atomic_add_all(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }

// ---- primitives of the flat (chunk-per-wave) path, tk_flat_impl.h ----
// lane-predicate from a wave-uniform 64-bit mask (v_cndmask with the SGPR pair as the condition)
TK_DEV bool wv_inverse_ballot(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// v_perm_b32: result byte i = byte sel[i] of the 8 bytes {hi:lo} (0..3 = lo, 4..7 = hi)
TK_DEV uint32_t wv_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// this wave's LDS slice: earlier ds writes of any lane are visible to later ds reads of every lane
TK_DEV void wv_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
TK_DEV void wv_lds_or(uint32_t* p, uint32_t v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

TK_DEV void wv_lds_and64(uint64_t* p, uint64_t v) {
    __hip_atomic_fetch_and(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// workgroup level (tk_long_impl.h: 16 waves merge one long piece): the thread's index in the block, the block barrier
TK_DEV uint32_t wv_tid() { return threadIdx.x; }
TK_DEV void wv_block_sync() { __syncthreads(); }

// 16 text bytes at p (any alignment) as 4 little-endian dwords
TK_DEV void wv_load16(const uint8_t* p, uint32_t* x) {
    typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_u;
    const u32x4_u v = *reinterpret_cast<const u32x4_u*>(p);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
}

// four consecutive words to an address that is only word-aligned (one global_store_dwordx4 per lane)
TK_DEV void wv_store16(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    typedef uint32_t __attribute__((ext_vector_type(4), aligned(4))) u32x4_w;
    u32x4_w v;
    v.x = a; v.y = b; v.z = c; v.w = d;
    *reinterpret_cast<u32x4_w*>(p) = v;
}

// inclusive prefix sum over the 64 lanes: DPP row shifts inside the rows of 16, then the two row broadcasts
// (lanes without a source read 0) -- VALU only, no LDS crossbar
TK_DEV uint32_t wv_scan_incl_u32(uint32_t v) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return x;
}

// bytes sh..sh+3 of the 8 bytes {hi:lo} (v_alignbyte_b32), sh in 0..3
TK_DEV uint32_t wv_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }

// a device address kept as an integer (e.g. in LDS) back as a pointer: say that it is GLOBAL memory, or the loads through it
// become flat_load (generic address space: slower path, counts against the LDS counter as well)
TK_DEV const uint8_t* wv_global_ptr(uint64_t addr) {
    typedef const uint8_t __attribute__((address_space(1))) * gptr_t;
    return (const uint8_t*)reinterpret_cast<gptr_t>(addr);
}

// The kernel's argument block (the struct the kernel takes by value as its FIRST parameter), opaque to the optimiser from here on: a
// field read through the result is a scalar load from the kernarg segment that stays where it is written.  Read through the
// parameter itself, every field is loaded at the kernel's entry and lives in scalar registers from there on -- a long kernel then
// spills them to VGPR lanes (v_writelane / v_readlane: VALU instructions of the dear kind) where a reload costs no VALU issue at all.
#define WV_KARGS(T, a) (*wv_kargs_fresh<T>())
template <class T>
TK_DEV const T __attribute__((address_space(4))) * wv_kargs_fresh() {
    typedef const T __attribute__((address_space(4))) * kptr_t;
    kptr_t p = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// "this value is needed HERE": keeps the compiler from sinking the load that produces it behind a later branch (it
// otherwise turns independent loads into a chain of conditional ones -- each a full memory round trip)
#define WV_PIN(x) asm volatile("" : "+v"(x))

#endif
