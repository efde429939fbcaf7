// tk_wave_emu.h -- CPU emulation of one 64-lane wavefront with fibers (ucontext).
//
// TEST INFRASTRUCTURE ONLY.  It exists so that the CPU test-suite (no GPU in the build
// container) can execute the very source of the device algorithm,
// tekken-rs_amd/csrc/tk_encode_impl.h, lane by lane, and compare it with the oracle.  It is
// never linked into libtekken_hip.so and is not reachable from the product API.
//
// Model: 64 fibers, one per lane.  A wave primitive (ballot / shuffle / ...) deposits the
// lane's operand and yields to the scheduler; once every live lane has arrived the scheduler
// publishes a snapshot and resumes the lanes.  The scheduler asserts that all lanes arrive at
// the SAME primitive, which is exactly the discipline real hardware needs (wave operations
// must be reached in wave-uniform control flow).
#ifndef TK_WAVE_EMU_H
#define TK_WAVE_EMU_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define TK_DEV static inline
#define TK_DEV_NOINLINE static

namespace tkemu {

enum Op { OP_NONE = 0, OP_BALLOT, OP_SHFL, OP_UP1, OP_DN1, OP_SYNC, OP_FIRST, OP_ATOMIC, OP_MIN, OP_BARRIER };

struct Wave {
    ucontext_t sched;
    ucontext_t ctx[64];
    std::vector<char> stacks[64];
    bool done[64];
    int cur = 0;
    int op[64];
    uint32_t dep_u32[64];
    uint8_t dep_pred[64];
    uint32_t snap_u32[64];
    uint64_t snap_ballot = 0;
    std::function<void(int)> body;
    uint64_t n_ops = 0;
};

extern Wave* g_wave;

inline void yield_op(int op) {
    Wave* w = g_wave;
    w->op[w->cur] = op;
    swapcontext(&w->ctx[w->cur], &w->sched);
}

inline void fiber_entry(int lane) {
    Wave* w = g_wave;
    w->body(lane);
    w->done[lane] = true;
    w->op[lane] = OP_NONE;
    swapcontext(&w->ctx[lane], &w->sched);
}

inline void run_wave(const std::function<void(int)>& body) {
    static Wave* wave = nullptr;
    if (!wave) {
        wave = new Wave();
        for (int l = 0; l < 64; ++l) wave->stacks[l].resize(512 * 1024);
    }
    Wave* w = wave;
    g_wave = w;
    w->body = body;
    for (int l = 0; l < 64; ++l) {
        w->done[l] = false;
        w->op[l] = OP_NONE;
        getcontext(&w->ctx[l]);
        w->ctx[l].uc_stack.ss_sp = w->stacks[l].data();
        w->ctx[l].uc_stack.ss_size = w->stacks[l].size();
        w->ctx[l].uc_link = nullptr;
        makecontext(&w->ctx[l], (void (*)())fiber_entry, 1, l);
    }
    for (;;) {
        int live = 0;
        for (int l = 0; l < 64; ++l) {
            if (w->done[l]) continue;
            w->cur = l;
            swapcontext(&w->sched, &w->ctx[l]);
            if (!w->done[l]) ++live;
        }
        if (live == 0) break;
        // every live lane is parked at a primitive: they must agree on which one
        int op = OP_NONE;
        for (int l = 0; l < 64; ++l) {
            if (w->done[l]) continue;
            if (op == OP_NONE) op = w->op[l];
            else if (op != w->op[l]) {
                fprintf(stderr, "tkemu: lanes diverged at a wave primitive (lane %d op %d vs %d)\n", l, w->op[l], op);
                abort();
            }
        }
        if (live != 64) {
            // a lane left the kernel while others still execute wave primitives
            fprintf(stderr, "tkemu: %d lanes exited early while others wait at op %d\n", 64 - live, op);
            abort();
        }
        uint64_t b = 0;
        for (int l = 0; l < 64; ++l) {
            w->snap_u32[l] = w->dep_u32[l];
            if (w->dep_pred[l]) b |= 1ull << l;
        }
        w->snap_ballot = b;
        ++w->n_ops;
    }
}

// ------------------------------------------------------------------------------------------
// A WORKGROUP of n waves (tk_long_impl.h: 16 waves merge one long piece): every wave is a Wave as above with its own 64
// fibers; the waves take turns, each running until its lanes park at the next primitive.  A wave whose lanes park at the
// workgroup barrier (wv_block_sync) waits until every wave of the block is there.  Between two barriers the waves run one
// after the other, every wave's lanes one after the other: code that is correct on the device -- no cross-wave or
// cross-lane race between barriers / wave primitives -- computes the same here, and every access goes through the host's
// sanitizers.  A wave that returns while others wait at a barrier is an error, like on the device.
// ------------------------------------------------------------------------------------------
struct Block {
    std::vector<Wave*> waves;
    int cur_wave = 0;
    uint64_t n_barriers = 0;
};
extern Block* g_block;   // non-null while run_block executes

inline void run_block(int n_waves, const std::function<void(int)>& body, size_t stack_bytes = 256 * 1024) {
    static Block* block = nullptr;
    if (!block) block = new Block();
    while ((int)block->waves.size() < n_waves) {
        Wave* w = new Wave();
        for (int l = 0; l < 64; ++l) w->stacks[l].resize(stack_bytes);
        block->waves.push_back(w);
    }
    Block* b = block;
    g_block = b;
    b->n_barriers = 0;
    std::vector<int> state(n_waves, 0);   // 0 running, 1 at the barrier, 2 done
    for (int v = 0; v < n_waves; ++v) {
        Wave* w = b->waves[v];
        w->body = body;
        w->n_ops = 0;
        for (int l = 0; l < 64; ++l) {
            w->done[l] = false;
            w->op[l] = OP_NONE;
            getcontext(&w->ctx[l]);
            w->ctx[l].uc_stack.ss_sp = w->stacks[l].data();
            w->ctx[l].uc_stack.ss_size = w->stacks[l].size();
            w->ctx[l].uc_link = nullptr;
            makecontext(&w->ctx[l], (void (*)())fiber_entry, 1, l);
        }
    }
    for (;;) {
        bool any_running = false;
        for (int v = 0; v < n_waves; ++v) {
            if (state[v] != 0) continue;
            any_running = true;
            Wave* w = b->waves[v];
            g_wave = w;
            b->cur_wave = v;
            int live = 0;
            for (int l = 0; l < 64; ++l) {
                if (w->done[l]) continue;
                w->cur = l;
                swapcontext(&w->sched, &w->ctx[l]);
                if (!w->done[l]) ++live;
            }
            if (live == 0) { state[v] = 2; continue; }
            int op = OP_NONE;
            for (int l = 0; l < 64; ++l) {
                if (w->done[l]) continue;
                if (op == OP_NONE) op = w->op[l];
                else if (op != w->op[l]) {
                    fprintf(stderr, "tkemu: wave %d: lanes diverged at a primitive (lane %d op %d vs %d)\n", v, l, w->op[l], op);
                    abort();
                }
            }
            if (live != 64) {
                fprintf(stderr, "tkemu: wave %d: %d lanes exited early while others wait at op %d\n", v, 64 - live, op);
                abort();
            }
            if (op == OP_BARRIER) { state[v] = 1; continue; }
            uint64_t bm = 0;
            for (int l = 0; l < 64; ++l) {
                w->snap_u32[l] = w->dep_u32[l];
                if (w->dep_pred[l]) bm |= 1ull << l;
            }
            w->snap_ballot = bm;
            ++w->n_ops;
        }
        if (any_running) continue;
        int at_barrier = 0, done = 0;
        for (int v = 0; v < n_waves; ++v) { at_barrier += state[v] == 1; done += state[v] == 2; }
        if (at_barrier == 0) break;
        if (done != 0) {
            fprintf(stderr, "tkemu: %d waves returned while %d wait at the workgroup barrier\n", done, at_barrier);
            abort();
        }
        ++b->n_barriers;
        for (int v = 0; v < n_waves; ++v) state[v] = 0;
    }
    g_block = nullptr;
}

}  // namespace tkemu

TK_DEV int wv_lane() { return tkemu::g_wave->cur; }
TK_DEV uint32_t wv_tid() { return tkemu::g_block ? (uint32_t)(tkemu::g_block->cur_wave * 64 + tkemu::g_wave->cur) : (uint32_t)tkemu::g_wave->cur; }
TK_DEV void wv_block_sync() { tkemu::yield_op(tkemu::g_block ? tkemu::OP_BARRIER : tkemu::OP_SYNC); }

TK_DEV uint64_t wv_ballot(bool p) {
    tkemu::Wave* w = tkemu::g_wave;
    w->dep_pred[w->cur] = p ? 1 : 0;
    tkemu::yield_op(tkemu::OP_BALLOT);
    return tkemu::g_wave->snap_ballot;
}

TK_DEV uint32_t wv_shfl(uint32_t v, int src) {
    tkemu::Wave* w = tkemu::g_wave;
    if (src < 0 || src > 63) {
        fprintf(stderr, "tkemu: wv_shfl source lane %d out of range (lane %d)\n", src, w->cur);
        abort();
    }
    w->dep_u32[w->cur] = v;
    tkemu::yield_op(tkemu::OP_SHFL);
    return tkemu::g_wave->snap_u32[src];
}

// all lanes are active in the emulator: the first active lane is lane 0
TK_DEV uint32_t wv_first(uint32_t v) {
    tkemu::Wave* w = tkemu::g_wave;
    w->dep_u32[w->cur] = v;
    tkemu::yield_op(tkemu::OP_FIRST);
    return tkemu::g_wave->snap_u32[0];
}
TK_DEV uint64_t wv_first64(uint64_t v) {
    return ((uint64_t)wv_first((uint32_t)(v >> 32)) << 32) | (uint64_t)wv_first((uint32_t)v);
}

TK_DEV uint32_t wv_readlane(uint32_t v, int l) { return wv_shfl(v, l); }

TK_DEV uint32_t wv_min_u32(uint32_t v) {
    tkemu::Wave* w = tkemu::g_wave;
    w->dep_u32[w->cur] = v;
    tkemu::yield_op(tkemu::OP_MIN);
    uint32_t m = 0xFFFFFFFFu;
    for (int l = 0; l < 64; ++l) m = tkemu::g_wave->snap_u32[l] < m ? tkemu::g_wave->snap_u32[l] : m;
    return m;
}

TK_DEV uint32_t wv_up1(uint32_t v) {
    tkemu::Wave* w = tkemu::g_wave;
    int lane = w->cur;
    w->dep_u32[lane] = v;
    tkemu::yield_op(tkemu::OP_UP1);
    return lane < 63 ? tkemu::g_wave->snap_u32[lane + 1] : 0u;
}

TK_DEV uint32_t wv_dn1(uint32_t v) {
    tkemu::Wave* w = tkemu::g_wave;
    int lane = w->cur;
    w->dep_u32[lane] = v;
    tkemu::yield_op(tkemu::OP_DN1);
    return lane > 0 ? tkemu::g_wave->snap_u32[lane - 1] : 0u;
}

TK_DEV uint64_t wv_brev64(uint64_t x) {
    uint64_t r = 0;
    for (int i = 0; i < 64; ++i) r |= ((x >> i) & 1ull) << (63 - i);
    return r;
}

TK_DEV void wv_sync() { tkemu::yield_op(tkemu::OP_SYNC); }

TK_DEV uint32_t wv_atomic_add(uint32_t* p, uint32_t v) {
    uint32_t old = *p;
    *p = old + v;
    return old;
}

TK_DEV uint32_t wv_atomic_exch(uint32_t* p, uint32_t v) {
    uint32_t old = *p;
    *p = v;
    return old;
}

TK_DEV uint32_t wv_load_coherent(const uint32_t* p) { return *p; }
TK_DEV uint32_t wv_atomic_max(uint32_t* p, uint32_t v) {
    uint32_t old = *p;
    if (v > old) *p = v;
    return old;
}

TK_DEV uint32_t wv_atomic_add_all(uint32_t* p, uint32_t v) {
    tkemu::Wave* w = tkemu::g_wave;
    int lane = w->cur;
    if (lane == 0) {  // lanes run in order 0..63 between two primitives
        w->dep_u32[0] = *p;
        *p += 64u * v;
    }
    tkemu::yield_op(tkemu::OP_ATOMIC);
    return tkemu::g_wave->snap_u32[0] + (uint32_t)lane * v;
}

// ---- primitives of the flat (chunk-per-wave) path ----
TK_DEV bool wv_inverse_ballot(uint64_t m) { return (m >> tkemu::g_wave->cur) & 1ull; }

TK_DEV uint32_t wv_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = (sel >> (8 * i)) & 0xFFu;
        if (s > 7) { fprintf(stderr, "tkemu: wv_perm selector %u not modelled\n", s); abort(); }
        r |= (uint32_t)((v >> (8 * s)) & 0xFFull) << (8 * i);
    }
    return r;
}

TK_DEV void wv_lds_sync() { tkemu::yield_op(tkemu::OP_SYNC); }
TK_DEV void wv_lds_or(uint32_t* p, uint32_t v) { *p |= v; }
TK_DEV void wv_lds_and64(uint64_t* p, uint64_t v) { *p &= v; }

TK_DEV void wv_load16(const uint8_t* p, uint32_t* x) { memcpy(x, p, 16); }
#define WV_KARGS(T, a) (a)
TK_DEV void wv_store16(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }

TK_DEV uint32_t wv_scan_incl_u32(uint32_t v) {
    tkemu::Wave* w = tkemu::g_wave;
    const int lane = w->cur;
    w->dep_u32[lane] = v;
    tkemu::yield_op(tkemu::OP_SHFL);
    uint32_t s = 0;
    for (int l = 0; l <= lane; ++l) s += tkemu::g_wave->snap_u32[l];
    return s;
}

TK_DEV uint32_t wv_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3u)));
}

#define WV_PIN(x) ((void)(x))
TK_DEV const uint8_t* wv_global_ptr(uint64_t addr) { return reinterpret_cast<const uint8_t*>((uintptr_t)addr); }

#endif
