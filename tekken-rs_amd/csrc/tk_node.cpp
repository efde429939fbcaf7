// tk_node.cpp -- node-level C ABI (include/tekken_hip.h, tk_node_*): every GPU of one node behind ONE call.
//
// BASELINE north_star: "a batch of documents shards ... across the 8xMI355X node with a single RCCL gather of token-id
// buffers over xGMI at the end"; SURVEY section 8b: ctx_create(.., device_ids[], n_devices, ..).  The reference has no
// counterpart (one process, one thread, one CPU: src/tekkenizer.rs:378-405 is a pure function of one &str) -- documents
// are independent, so the path shards with no data-path collective:
//
//   * one process, one tk_ctx (tables, stream, workspace) per device, one host thread per device and call;
//   * the batch is cut into contiguous runs of whole documents with balanced BYTES (not counts: C5 has 16 B .. 32 KiB);
//   * every device tokenizes its run (tk_encode_batch_device: ids stay in HBM);
//   * ONE exchange: the host knows every run's id count (single process: no size collective needed), then inside one
//     ncclGroupStart / ncclGroupEnd the root posts a ncclRecv per peer and every peer one ncclSend -- direct
//     peer -> root transfers, each on its own xGMI link (never a ring: xGMI is point to point, a ring is bound by one
//     link).  The ids travel in the 18-bit wire format when the vocabulary allows it (tk_pack_ids18_device:
//     2.25 bytes per id), the root unpacks every peer's payload into its slice of the output as it arrives and
//     rebases the per-document offsets -- ids land in document order on the root device, one copy to the host.
//
// RCCL is opened with dlopen when a node of more than one device is created: a single-GPU caller never loads it, and in
// a process that already holds a RCCL (PyTorch bundles one) that copy is reused.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tekken_hip.h"
#include "tk_engine.h"
#include "tk_kernels.h"

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool open(std::string& err) {
        if (h) return true;
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);           // a copy this process already holds (PyTorch's)
        for (const char* n : {"librccl.so.1", "librccl.so"})
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) { err = std::string("RCCL is not available: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(h, "ncclSend");
        Recv = (decltype(Recv))dlsym(h, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            err = "RCCL library lacks a needed symbol";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;
thread_local std::string g_node_tls_err;

struct DBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {       // the caller has made the buffer's device current
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// pinned host staging (the run-relative document offsets of a device's shard): kept across calls, grown on demand
struct HBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// One host thread per device for the life of the node (the first form started and joined a std::thread per device and
// call): a call hands every worker its job and waits for all of them.
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has = false, quit = false;
    void loop() {
        for (;;) {
            std::function<void()> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return has || quit; });
                if (quit && !has) return;
                j = std::move(job);
            }
            j();
            {
                std::lock_guard<std::mutex> lk(m);
                has = false;
            }
            cv.notify_all();
        }
    }
    void start() { th = std::thread([this] { loop(); }); }
    void post(std::function<void()> j) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            has = true;
        }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !has; });
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

struct Shard {
    uint64_t d0 = 0, d1 = 0, b0 = 0, n_bytes = 0;   // documents [d0, d1), bytes from b0
    uint64_t n_ids = 0;
    void* d_ids = nullptr;                           // context-owned, valid until the context's next call
    void* d_oo = nullptr;
    int rc = TK_OK;
    std::string err;
};

}  // namespace

struct tk_node {
    int n = 0;
    std::vector<int> devs;
    std::vector<tk_ctx*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;              // empty when the node runs without RCCL (one device)
    std::vector<DBuf> in_bytes, in_offs, packed;
    std::vector<HBuf> rel;                     // pinned: run-relative document offsets going up
    std::vector<Worker*> workers;              // one per device (n > 1)
    std::vector<DBuf> stage;                   // on the root: the peers' payloads as they arrive
    DBuf all_ids, all_offs;                    // on the root: the gathered result
    bool wire18 = false;                       // every id fits 18 bits
    bool loopback = false;                     // TK_NODE_FORCE_RCCL: also the root's own run travels through send / recv
    bool d2d = false;                          // test builds only (TK_NODE_TEST_TRANSPORT): device-to-device copies stand in for ncclSend / ncclRecv
    std::vector<hipEvent_t> ev_peer;           // d2d: the peer's payload is ready (recorded on its stream, awaited by the root's)
    std::vector<uint64_t> last_shard_bytes, last_shard_ids;
    std::mutex mu;
    std::string err;
    float last_gather_ms = 0.f, last_kernels_ms = 0.f;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

#define NODE_HIP(nd, call)                                                                          \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (nd)->err = std::string(#call) + ": " + hipGetErrorString(_e);                         \
            return TK_ERR_RUNTIME;                                                                 \
        }                                                                                          \
    } while (0)
#define NODE_NCCL(nd, call)                                                                         \
    do {                                                                                           \
        ncclResult_t _r = (call);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            (nd)->err = std::string(#call) + ": " + g_rccl.GetErrorString(_r);                     \
            return TK_ERR_RUNTIME;                                                                 \
        }                                                                                          \
    } while (0)

extern "C" const char* tk_node_last_error(const tk_node* nd) { return nd ? nd->err.c_str() : g_node_tls_err.c_str(); }

extern "C" void tk_node_destroy(tk_node* nd) {
    if (!nd) return;
    for (Worker* w : nd->workers) { w->stop(); delete w; }
    nd->workers.clear();
    for (HBuf& h : nd->rel) h.release();
    for (int i = 0; i < (int)nd->comm.size(); ++i)
        if (nd->comm[i]) (void)g_rccl.CommDestroy(nd->comm[i]);
    for (hipEvent_t e : nd->ev_peer)
        if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < nd->n; ++i) {
        if (i < (int)nd->devs.size()) (void)hipSetDevice(nd->devs[i]);
        if (i < (int)nd->in_bytes.size()) { nd->in_bytes[i].release(); nd->in_offs[i].release(); nd->packed[i].release(); nd->stage[i].release(); }
        if (i == 0) {
            nd->all_ids.release(); nd->all_offs.release();
            if (nd->ev0) (void)hipEventDestroy(nd->ev0);
            if (nd->ev1) (void)hipEventDestroy(nd->ev1);
        }
        if (i < (int)nd->stream.size() && nd->stream[i]) (void)hipStreamDestroy(nd->stream[i]);
        if (i < (int)nd->ctx.size() && nd->ctx[i]) tk_ctx_destroy(nd->ctx[i]);
    }
    delete nd;
}

extern "C" int tk_node_create(const uint8_t* token_bytes, const uint32_t* token_offsets, uint32_t n_ranks, uint32_t num_special_tokens,
                              uint32_t bos_id, uint32_t eos_id, const int* device_ids, int n_devices, tk_node** out_node) {
    if (!out_node) { g_node_tls_err = "out_node is NULL"; return TK_ERR_INVALID_ARG; }
    *out_node = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) { g_node_tls_err = "n_devices must be 1..64"; return TK_ERR_INVALID_ARG; }
    bool d2d = false;
#ifdef TK_NODE_TEST_TRANSPORT
    // Test builds only (`make ablate`; never in the shipped library): TK_NODE_TRANSPORT=d2d runs the WHOLE N > 1 branch -- shards,
    // worker threads, 18-bit packing, staging, unpacking, offset rebase -- with stream-ordered device-to-device copies standing
    // in for ncclSend / ncclRecv, and lets device ids repeat: N contexts on ONE GPU.  What it cannot show is the link itself.
    if (const char* tr = getenv("TK_NODE_TRANSPORT")) d2d = strcmp(tr, "d2d") == 0;
#endif
    for (int i = 0; i < n_devices && !d2d; ++i)
        for (int j = 0; j < i; ++j)
            if (device_ids[i] == device_ids[j]) {
                g_node_tls_err = "device " + std::to_string(device_ids[i]) + " is listed twice: one context per GPU";
                return TK_ERR_INVALID_ARG;
            }
    tk_node* nd = new tk_node();
    nd->d2d = d2d;
    nd->n = n_devices;
    nd->devs.assign(device_ids, device_ids + n_devices);
    nd->ctx.assign(n_devices, nullptr);
    nd->stream.assign(n_devices, nullptr);
    nd->in_bytes.resize(n_devices); nd->in_offs.resize(n_devices); nd->packed.resize(n_devices); nd->stage.resize(n_devices);
    nd->rel.resize(n_devices);
    auto fail = [&](int code, const std::string& msg) {
        g_node_tls_err = msg;
        tk_node_destroy(nd);
        return code;
    };
    for (int i = 0; i < n_devices; ++i) {
        // (tables are replicated: every device holds its own copy, built from the same rank table)
        int rc = tk_ctx_create(token_bytes, token_offsets, n_ranks, num_special_tokens, bos_id, eos_id, device_ids[i], &nd->ctx[i]);
        if (rc != TK_OK) return fail(rc, std::string("device ") + std::to_string(device_ids[i]) + ": " + tk_last_error(nullptr));
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreateWithFlags(&nd->stream[i], hipStreamNonBlocking) != hipSuccess)
            return fail(TK_ERR_RUNTIME, "hipStreamCreate failed on device " + std::to_string(device_ids[i]));
    }
    nd->wire18 = (uint64_t)n_ranks + num_special_tokens <= (1ull << 18);
    nd->loopback = getenv("TK_NODE_FORCE_RCCL") != nullptr;
    if (hipSetDevice(device_ids[0]) != hipSuccess || hipEventCreate(&nd->ev0) != hipSuccess || hipEventCreate(&nd->ev1) != hipSuccess)
        return fail(TK_ERR_RUNTIME, "hipEventCreate failed");
    if (nd->d2d) {
        nd->ev_peer.assign(n_devices, nullptr);
        for (int i = 0; i < n_devices; ++i)
            if (hipSetDevice(device_ids[i]) != hipSuccess || hipEventCreateWithFlags(&nd->ev_peer[i], hipEventDisableTiming) != hipSuccess)
                return fail(TK_ERR_RUNTIME, "hipEventCreate failed");
    } else if (n_devices > 1 || nd->loopback) {
        std::lock_guard<std::mutex> lock(g_rccl_mu);
        std::string e;
        if (!g_rccl.open(e)) return fail(TK_ERR_RUNTIME, e);
        nd->comm.assign(n_devices, nullptr);
        ncclResult_t r = g_rccl.CommInitAll(nd->comm.data(), n_devices, nd->devs.data());
        if (r != ncclSuccess) {
            nd->comm.clear();
            return fail(TK_ERR_RUNTIME, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
        }
    }
    if (n_devices > 1)
        for (int i = 0; i < n_devices; ++i) {
            nd->workers.push_back(new Worker());
            nd->workers.back()->start();
        }
    *out_node = nd;
    return TK_OK;
}

// contiguous document ranges with balanced BYTES: n + 1 cut points (the rule of tekken-rs_amd/parallel.py shard_by_bytes)
static std::vector<uint64_t> shard_by_bytes(const uint64_t* offs, uint64_t n_docs, int n) {
    std::vector<uint64_t> cuts(1, 0);
    const uint64_t total = offs[n_docs] - offs[0];
    for (int r = 1; r < n; ++r) {
        const uint64_t target = offs[0] + (uint64_t)((unsigned __int128)total * (uint64_t)r / (uint64_t)n);
        uint64_t lo = 0, hi = n_docs + 1;                      // first index with offs[i] >= target
        while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (offs[mid] < target) lo = mid + 1; else hi = mid; }
        uint64_t c = lo;
        if (c < cuts.back()) c = cuts.back();
        if (c > n_docs) c = n_docs;
        cuts.push_back(c);
    }
    cuts.push_back(n_docs);
    return cuts;
}

// ids_dst == NULL: the result goes into pinned blocks of the library's pool (tk_free_result); otherwise into the caller's
// buffers (ids_cap entries, n_docs + 1 offsets)
static int node_encode(tk_node* nd, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs, int add_bos, int add_eos,
                       uint32_t* ids_dst, uint64_t ids_cap, uint64_t* offs_dst, uint32_t** ids_res, uint64_t** offs_res, uint64_t* n_ids_res) {
    if (doc_offsets[0] != 0) { nd->err = "doc_offsets[0] must be 0"; return TK_ERR_INVALID_ARG; }
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_offsets[d + 1] < doc_offsets[d]) { nd->err = "doc_offsets must be non-decreasing"; return TK_ERR_INVALID_ARG; }
    const int n = nd->n;
    const std::vector<uint64_t> cuts = shard_by_bytes(doc_offsets, n_docs, n);
    std::vector<Shard> sh(n);

    // ---- every device: its run up, tokenized, and (peers) packed for the wire -- one host thread per device ----
    auto work = [&](int k) {
        Shard& s = sh[k];
        s.d0 = cuts[k]; s.d1 = cuts[k + 1];
        s.b0 = doc_offsets[s.d0]; s.n_bytes = doc_offsets[s.d1] - s.b0;
        const uint64_t nd_k = s.d1 - s.d0;
        auto hip_fail = [&](const char* what, hipError_t e) { s.rc = TK_ERR_RUNTIME; s.err = std::string(what) + ": " + hipGetErrorString(e); };
        hipError_t e = hipSetDevice(nd->devs[k]);
        if (e != hipSuccess) return hip_fail("hipSetDevice", e);
        if ((e = nd->in_bytes[k].reserve(s.n_bytes + 64)) != hipSuccess || (e = nd->in_offs[k].reserve((nd_k + 1) * 8)) != hipSuccess)
            return hip_fail("hipMalloc", e);
        // (the run-relative offsets in pinned staging that lives as long as the node: both copies are asynchronous DMAs when the
        // caller's text is pinned too -- tk_host_alloc --, and the encode call below is the one wait)
        if ((e = nd->rel[k].reserve((nd_k + 1) * 8)) != hipSuccess) return hip_fail("hipHostMalloc", e);
        uint64_t* rel = (uint64_t*)nd->rel[k].p;
        for (uint64_t d = 0; d <= nd_k; ++d) rel[d] = doc_offsets[s.d0 + d] - s.b0;
        hipStream_t st = nd->stream[k];
        if (s.n_bytes && (e = hipMemcpyAsync(nd->in_bytes[k].p, bytes + s.b0, s.n_bytes, hipMemcpyHostToDevice, st)) != hipSuccess)
            return hip_fail("hipMemcpyAsync", e);
        if ((e = hipMemcpyAsync(nd->in_offs[k].p, rel, (nd_k + 1) * 8, hipMemcpyHostToDevice, st)) != hipSuccess)
            return hip_fail("hipMemcpyAsync", e);
        s.rc = tk_encode_batch_device(nd->ctx[k], nd->in_bytes[k].p, nd->in_offs[k].p, nd_k, s.n_bytes, add_bos, add_eos, st, &s.d_ids,
                                      &s.d_oo, &s.n_ids);
        if (s.rc != TK_OK) { s.err = tk_last_error(nd->ctx[k]); return; }
        const bool travels = k != 0 || nd->loopback;
        if (travels && nd->wire18 && s.n_ids) {
            if ((e = nd->packed[k].reserve(tk_ids18_bytes(s.n_ids))) != hipSuccess) return hip_fail("hipMalloc", e);
            s.rc = tk_pack_ids18_device(nd->ctx[k], s.d_ids, s.n_ids, nd->packed[k].p, st);
            if (s.rc != TK_OK) s.err = tk_last_error(nd->ctx[k]);
        }
    };
    if (n == 1) {
        work(0);
    } else {
        for (int k = 0; k < n; ++k) nd->workers[k]->post([&work, k] { work(k); });
        for (int k = 0; k < n; ++k) nd->workers[k]->wait();
    }
    for (int k = 0; k < n; ++k)
        if (sh[k].rc != TK_OK) {
            // a device failed: copies from the caller's buffer and from the pinned offset staging may still be queued on the other
            // devices' streams -- nothing returns to the caller, who may free or reuse that memory, before they have drained
            for (int j = 0; j < n; ++j) {
                (void)hipSetDevice(nd->devs[j]);
                (void)hipStreamSynchronize(nd->stream[j]);
            }
            (void)hipSetDevice(nd->devs[0]);
            nd->err = "device " + std::to_string(nd->devs[k]) + ": " + sh[k].err;
            return sh[k].rc;
        }
    nd->last_shard_bytes.assign(n, 0); nd->last_shard_ids.assign(n, 0);
    for (int k = 0; k < n; ++k) { nd->last_shard_bytes[k] = sh[k].n_bytes; nd->last_shard_ids[k] = sh[k].n_ids; }

    // ---- the one exchange: every run's ids (+ per-document offsets) to the root device, document order ----
    uint64_t total = 0;
    std::vector<uint64_t> base(n + 1, 0);
    for (int k = 0; k < n; ++k) { base[k] = total; total += sh[k].n_ids; }
    base[n] = total;
    NODE_HIP(nd, hipSetDevice(nd->devs[0]));
    NODE_HIP(nd, nd->all_ids.reserve((total ? total : 1) * 4));
    NODE_HIP(nd, nd->all_offs.reserve((n_docs + 1) * 8));
    uint32_t* ids = (uint32_t*)nd->all_ids.p;
    uint64_t* offs = (uint64_t*)nd->all_offs.p;
    hipStream_t rs = nd->stream[0];
    NODE_HIP(nd, hipEventRecord(nd->ev0, rs));
    NODE_HIP(nd, hipMemsetAsync(offs, 0, 8, rs));
    const bool use_rccl = !nd->comm.empty() || nd->d2d;
    if (!(use_rccl && nd->loopback)) {
        // the root's own run: device-to-device inside the root
        if (sh[0].n_ids) NODE_HIP(nd, hipMemcpyAsync(ids, sh[0].d_ids, sh[0].n_ids * 4, hipMemcpyDeviceToDevice, rs));
        if (sh[0].d1 > sh[0].d0)
            NODE_HIP(nd, hipMemcpyAsync(offs + 1, (const uint64_t*)sh[0].d_oo + 1, (sh[0].d1 - sh[0].d0) * 8, hipMemcpyDeviceToDevice, rs));
    }
    if (use_rccl) {
        for (int k = nd->loopback ? 0 : 1; k < n; ++k)
            if (nd->wire18 && sh[k].n_ids) NODE_HIP(nd, nd->stage[k].reserve(tk_ids18_bytes(sh[k].n_ids)));
        // one transfer peer k -> root: ncclSend on the peer's communicator and stream + ncclRecv on the root's (all of them inside
        // ONE group: seven links side by side), or -- test transport -- a device-to-device copy on the root's stream behind an
        // event on the peer's
        auto xfer = [&](int k, const void* src, void* dst, size_t count, ncclDataType_t ty, size_t elem) -> int {
            if (nd->d2d) {
                NODE_HIP(nd, hipMemcpyAsync(dst, src, count * elem, hipMemcpyDeviceToDevice, rs));
                return TK_OK;
            }
            NODE_NCCL(nd, g_rccl.Send(src, count, ty, 0, nd->comm[k], nd->stream[k]));
            NODE_NCCL(nd, g_rccl.Recv(dst, count, ty, k, nd->comm[0], rs));
            return TK_OK;
        };
        if (nd->d2d) {
            for (int k = nd->loopback ? 0 : 1; k < n; ++k) {
                NODE_HIP(nd, hipSetDevice(nd->devs[k]));
                NODE_HIP(nd, hipEventRecord(nd->ev_peer[k], nd->stream[k]));
                NODE_HIP(nd, hipSetDevice(nd->devs[0]));
                NODE_HIP(nd, hipStreamWaitEvent(rs, nd->ev_peer[k], 0));
            }
        } else {
            NODE_NCCL(nd, g_rccl.GroupStart());
        }
        auto post = [&]() -> int {                                 // (a failure in here must still close the group)
            for (int k = nd->loopback ? 0 : 1; k < n; ++k) {
                const Shard& s = sh[k];
                const uint64_t nd_k = s.d1 - s.d0;
                int rc = TK_OK;
                if (s.n_ids) {
                    if (nd->wire18) rc = xfer(k, nd->packed[k].p, nd->stage[k].p, tk_ids18_bytes(s.n_ids), ncclUint8, 1);
                    else rc = xfer(k, s.d_ids, ids + base[k], s.n_ids, ncclUint32, 4);
                    if (rc != TK_OK) return rc;
                }
                if (nd_k && (rc = xfer(k, (const uint64_t*)s.d_oo + 1, offs + s.d0 + 1, nd_k, ncclUint64, 8)) != TK_OK) return rc;
            }
            return TK_OK;
        };
        const int prc = post();
        const ncclResult_t ge = nd->d2d ? ncclSuccess : g_rccl.GroupEnd();
        if (prc != TK_OK) return prc;
        if (ge != ncclSuccess) { nd->err = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge); return TK_ERR_RUNTIME; }
        for (int k = nd->loopback ? 0 : 1; k < n; ++k) {
            const Shard& s = sh[k];
            if (nd->wire18 && s.n_ids) {
                int rc = tk_unpack_ids18_device(nd->ctx[0], nd->stage[k].p, s.n_ids, ids + base[k], rs);
                if (rc != TK_OK) { nd->err = tk_last_error(nd->ctx[0]); return rc; }
            }
        }
    }
    // run-relative id offsets -> batch offsets
    for (int k = 1; k < n; ++k)
        if (base[k] && sh[k].d1 > sh[k].d0) NODE_HIP(nd, tk_launch_add_u64(offs + sh[k].d0 + 1, sh[k].d1 - sh[k].d0, base[k], rs));
    NODE_HIP(nd, hipEventRecord(nd->ev1, rs));

    // ---- down to the host ----
    *n_ids_res = total;
    if (ids_dst && total > ids_cap) { nd->err = "ids_out is too small for " + std::to_string(total) + " ids"; return TK_ERR_INVALID_ARG; }
    uint32_t* h_ids = ids_dst ? ids_dst : (uint32_t*)tk_pinned_get((total ? total : 1) * 4);
    uint64_t* h_offs = ids_dst ? offs_dst : (uint64_t*)tk_pinned_get((n_docs + 1) * 8);
    if (!h_ids || !h_offs) {
        if (!ids_dst) { tk_pinned_put(h_ids); tk_pinned_put(h_offs); }
        nd->err = "hipHostMalloc failed";
        return TK_ERR_RUNTIME;
    }
    hipError_t e = hipSuccess;
    if (total) e = hipMemcpyAsync(h_ids, ids, total * 4, hipMemcpyDeviceToHost, rs);
    if (e == hipSuccess) e = hipMemcpyAsync(h_offs, offs, (n_docs + 1) * 8, hipMemcpyDeviceToHost, rs);
    if (e == hipSuccess) e = hipStreamSynchronize(rs);
    for (int k = 1; k < n && e == hipSuccess; ++k) {              // the peers' sends are complete (their buffers are reused next call)
        (void)hipSetDevice(nd->devs[k]);
        e = hipStreamSynchronize(nd->stream[k]);
    }
    (void)hipSetDevice(nd->devs[0]);
    if (e != hipSuccess) {
        if (!ids_dst) { tk_pinned_put(h_ids); tk_pinned_put(h_offs); }
        nd->err = std::string("gather / result copy failed: ") + hipGetErrorString(e);
        return TK_ERR_RUNTIME;
    }
    (void)hipEventElapsedTime(&nd->last_gather_ms, nd->ev0, nd->ev1);
    float kmax = 0.f;
    for (int k = 0; k < n; ++k) {
        float p = 0.f;
        (void)tk_last_timing(nd->ctx[k], &p, nullptr);
        if (p > kmax) kmax = p;
    }
    nd->last_kernels_ms = kmax;
    *ids_res = h_ids;
    *offs_res = h_offs;
    return TK_OK;
}

extern "C" int tk_node_encode_batch(tk_node* nd, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs, int add_bos,
                                    int add_eos, tk_result* out) {
    if (!nd) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(nd->mu);
    if (!doc_offsets || !out || (!bytes && doc_offsets[n_docs])) { nd->err = "null argument"; return TK_ERR_INVALID_ARG; }
    memset(out, 0, sizeof(*out));
    uint32_t* h_ids = nullptr;
    uint64_t* h_offs = nullptr;
    uint64_t total = 0;
    int rc = node_encode(nd, bytes, doc_offsets, n_docs, add_bos, add_eos, nullptr, 0, nullptr, &h_ids, &h_offs, &total);
    if (rc != TK_OK) return rc;
    out->ids = h_ids;
    out->offsets = h_offs;
    out->n_ids = total;
    out->n_docs = n_docs;
    return TK_OK;
}

extern "C" int tk_node_encode_batch_pinned(tk_node* nd, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs, int add_bos,
                                           int add_eos, uint32_t* ids_out, uint64_t ids_capacity, uint64_t* offsets_out, uint64_t* n_ids_out) {
    if (!nd) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(nd->mu);
    if (!doc_offsets || !ids_out || !offsets_out || !n_ids_out || (!bytes && doc_offsets[n_docs])) { nd->err = "null argument"; return TK_ERR_INVALID_ARG; }
    uint32_t* h_ids = nullptr;
    uint64_t* h_offs = nullptr;
    *n_ids_out = 0;
    return node_encode(nd, bytes, doc_offsets, n_docs, add_bos, add_eos, ids_out, ids_capacity, offsets_out, &h_ids, &h_offs, n_ids_out);
}

extern "C" int tk_node_last_timing(const tk_node* nd, float* kernels_ms_max, float* gather_ms) {
    if (!nd) return TK_ERR_INVALID_ARG;
    if (kernels_ms_max) *kernels_ms_max = nd->last_kernels_ms;
    if (gather_ms) *gather_ms = nd->last_gather_ms;
    return TK_OK;
}

extern "C" int tk_node_n_devices(const tk_node* nd) { return nd ? nd->n : 0; }

extern "C" int tk_node_last_shards(const tk_node* nd, uint64_t* shard_bytes, uint64_t* shard_ids, int cap) {
    if (!nd) return TK_ERR_INVALID_ARG;
    for (int k = 0; k < cap && k < (int)nd->last_shard_bytes.size(); ++k) {
        if (shard_bytes) shard_bytes[k] = nd->last_shard_bytes[k];
        if (shard_ids) shard_ids[k] = nd->last_shard_ids[k];
    }
    return (int)nd->last_shard_bytes.size();
}
