//! The C ABI of tekken-rs_amd/libtekken_hip.so, one declaration per entry of include/tekken_hip.h that a Rust host needs.
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct TkCtx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct TkNode {
    _p: [u8; 0],
}
#[repr(C)]
pub struct TkResult {
    pub ids: *mut u32,
    pub offsets: *mut u64,
    pub n_ids: u64,
    pub n_docs: u64,
}
#[repr(C)]
pub struct TkTextResult {
    pub bytes: *mut u8,
    pub offsets: *mut u64,
    pub n_bytes: u64,
    pub n_docs: u64,
}

pub const TK_OK: c_int = 0;
pub const TK_ERR_INVALID_CONFIG: c_int = -1;
pub const TK_ERR_RUNTIME: c_int = -2;
pub const TK_ERR_INVALID_UTF8: c_int = -3;
pub const TK_ERR_NO_DEVICE: c_int = -4;
pub const TK_ERR_INVALID_ARG: c_int = -5;
pub const TK_ERR_TOKEN_NOT_FOUND: c_int = -9;
pub const TK_ERR_SPECIAL_POLICY: c_int = -10;

extern "C" {
    // engine level: replaces CoreBPE::new / CoreBPE::encode (src/tekkenizer.rs:125, :384-386)
    pub fn tk_ctx_create(token_bytes: *const u8, token_offsets: *const u32, n_ranks: u32, num_special_tokens: u32, bos_id: u32,
                         eos_id: u32, device_id: c_int, out_ctx: *mut *mut TkCtx) -> c_int;
    pub fn tk_ctx_destroy(ctx: *mut TkCtx);
    pub fn tk_last_error(ctx: *const TkCtx) -> *const c_char;
    pub fn tk_encode_batch(ctx: *mut TkCtx, bytes: *const u8, doc_offsets: *const u64, n_docs: u64, add_bos: c_int, add_eos: c_int,
                           validate_utf8: c_int, out: *mut TkResult) -> c_int;
    pub fn tk_free_result(r: *mut TkResult);
    // one document, caller-owned output (the reference's own call shape; no allocation per call)
    pub fn tk_encode_one(ctx: *mut TkCtx, text: *const u8, len: u64, add_bos: c_int, add_eos: c_int, ids_out: *mut u32,
                         ids_capacity: u64, n_ids: *mut u64) -> c_int;
    pub fn tk_encode_batch_device(ctx: *mut TkCtx, d_bytes: *const c_void, d_doc_offsets: *const c_void, n_docs: u64, n_bytes: u64,
                                  add_bos: c_int, add_eos: c_int, hip_stream: *mut c_void, d_ids: *mut *mut c_void,
                                  d_out_offsets: *mut *mut c_void, n_ids: *mut u64) -> c_int;
    // the same with the checks tk_encode_batch makes for host callers, on the device: TK_CHECK_OFFSETS = 1, TK_CHECK_UTF8 = 2
    pub fn tk_encode_batch_device_ex(ctx: *mut TkCtx, d_bytes: *const c_void, d_doc_offsets: *const c_void, n_docs: u64, n_bytes: u64,
                                     add_bos: c_int, add_eos: c_int, checks: c_int, hip_stream: *mut c_void, d_ids: *mut *mut c_void,
                                     d_out_offsets: *mut *mut c_void, n_ids: *mut u64) -> c_int;
    // memo of merged pieces (round 4): a device table {unknown piece of 2..16 bytes -> its <= 4 ids}; never changes an id
    pub fn tk_ctx_set_memo(ctx: *mut TkCtx, log2_entries: c_int, policy: c_int) -> c_int;
    pub fn tk_ctx_memo_clear(ctx: *mut TkCtx) -> c_int;
    pub fn tk_memo_stats(ctx: *const TkCtx, lookups_last: *mut u64, hits_last: *mut u64, lookups_total: *mut u64, hits_total: *mut u64,
                         active_last: *mut c_int) -> c_int;
    pub fn tk_host_alloc(bytes: usize) -> *mut c_void;
    pub fn tk_host_free(p: *mut c_void);
    pub fn tk_encode_batch_pipelined(ctx: *mut TkCtx, bytes: *const u8, doc_offsets: *const u64, n_docs: u64, add_bos: c_int,
                                     add_eos: c_int, slice_bytes: u64, ids_out: *mut u32, ids_capacity: u64, offsets_out: *mut u64,
                                     n_ids: *mut u64) -> c_int;
    pub fn tk_ctx_set_special_tokens(ctx: *mut TkCtx, strings_blob: *const u8, string_offsets: *const u32, n: u32) -> c_int;
    pub fn tk_decode_batch(ctx: *mut TkCtx, ids: *const u32, id_offsets: *const u64, n_docs: u64, policy: c_int,
                           out: *mut TkTextResult, bad_doc: *mut u64) -> c_int;
    pub fn tk_free_text_result(r: *mut TkTextResult);
    pub fn tk_ctx_set_pattern(ctx: *mut TkCtx, mode: c_int) -> c_int;

    // node level: every GPU of the node behind one call (north star: documents sharded across the GPUs, one RCCL
    // gather of the id buffers over xGMI); SURVEY section 8b `ctx_create(.., device_ids[], n_devices, ..)`
    pub fn tk_node_create(token_bytes: *const u8, token_offsets: *const u32, n_ranks: u32, num_special_tokens: u32, bos_id: u32,
                          eos_id: u32, device_ids: *const c_int, n_devices: c_int, out_node: *mut *mut TkNode) -> c_int;
    pub fn tk_node_destroy(node: *mut TkNode);
    pub fn tk_node_last_error(node: *const TkNode) -> *const c_char;
    pub fn tk_node_encode_batch(node: *mut TkNode, bytes: *const u8, doc_offsets: *const u64, n_docs: u64, add_bos: c_int,
                                add_eos: c_int, out: *mut TkResult) -> c_int;
    // caller-owned host buffers (tk_host_alloc: pinned -- nothing allocated, pinned or copied on the host per call)
    pub fn tk_node_encode_batch_pinned(node: *mut TkNode, bytes: *const u8, doc_offsets: *const u64, n_docs: u64, add_bos: c_int,
                                       add_eos: c_int, ids_out: *mut u32, ids_capacity: u64, offsets_out: *mut u64, n_ids_out: *mut u64) -> c_int;
    // how the last batch was cut: text bytes and ids of every device's run
    pub fn tk_node_last_shards(node: *const TkNode, shard_bytes: *mut u64, shard_ids: *mut u64, cap: c_int) -> c_int;
}
