//! Safe wrappers over ffi.rs.  `HipEngine` = one GPU (one `tk_ctx`), `HipNode` = several GPUs of one node.
use crate::ffi::*;
use std::os::raw::c_int;

/// What the shim hands back; tekken-rs maps it onto its `TokenizerError` (src/errors.rs:23-59):
/// InvalidConfig -> `TokenizerError::InvalidConfig` (:45-46), TokenNotFound -> `TokenNotFound` (:49-50),
/// SpecialTokenPolicy -> `SpecialTokenPolicy` (:53-54), everything else -> `Tokenizers(msg)` (:37-38).
#[derive(Debug)]
pub enum HipError {
    InvalidConfig(String),
    TokenNotFound(String),
    SpecialTokenPolicy(String),
    Tokenizers(String),
}

impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        match self {
            HipError::InvalidConfig(m) | HipError::TokenNotFound(m) | HipError::SpecialTokenPolicy(m) | HipError::Tokenizers(m) => f.write_str(m),
        }
    }
}
impl std::error::Error for HipError {}

fn map_err(rc: c_int, msg: *const std::os::raw::c_char) -> HipError {
    let msg = if msg.is_null() { String::new() } else { unsafe { std::ffi::CStr::from_ptr(msg) }.to_string_lossy().into_owned() };
    match rc {
        TK_ERR_INVALID_CONFIG | TK_ERR_INVALID_ARG => HipError::InvalidConfig(msg),
        TK_ERR_TOKEN_NOT_FOUND => HipError::TokenNotFound(msg),
        TK_ERR_SPECIAL_POLICY => HipError::SpecialTokenPolicy(msg),
        _ => HipError::Tokenizers(msg), // runtime / no device / invalid UTF-8
    }
}

fn pack_ranks(ranks: &[Vec<u8>]) -> (Vec<u8>, Vec<u32>) {
    let mut blob = Vec::with_capacity(ranks.iter().map(|t| t.len()).sum());
    let mut offs = Vec::with_capacity(ranks.len() + 1);
    offs.push(0u32);
    for t in ranks {
        blob.extend_from_slice(t);
        offs.push(blob.len() as u32);
    }
    (blob, offs)
}

fn pack_docs(docs: &[&str]) -> (Vec<u8>, Vec<u64>) {
    let mut bytes = Vec::with_capacity(docs.iter().map(|d| d.len()).sum());
    let mut offs = Vec::with_capacity(docs.len() + 1);
    offs.push(0u64);
    for d in docs {
        bytes.extend_from_slice(d.as_bytes());
        offs.push(bytes.len() as u64);
    }
    (bytes, offs)
}

unsafe fn take(res: &mut TkResult, n_docs: usize) -> Vec<Vec<u32>> {
    let ids = std::slice::from_raw_parts(res.ids, res.n_ids as usize);
    let o = std::slice::from_raw_parts(res.offsets, n_docs + 1);
    let out = (0..n_docs).map(|d| ids[o[d] as usize..o[d + 1] as usize].to_vec()).collect();
    tk_free_result(res);
    out
}

pub struct HipEngine {
    ctx: *mut TkCtx,
}
unsafe impl Send for HipEngine {}
unsafe impl Sync for HipEngine {} // the context serialises its calls internally (one stream + mutex)

impl HipEngine {
    /// `ranks[i]` = token bytes of rank i, i.e. the inverse of the map `reload_mergeable_ranks` builds (src/tekkenizer.rs:776-816).
    pub fn new(ranks: &[Vec<u8>], num_special: u32, bos: u32, eos: u32, device: i32) -> Result<Self, HipError> {
        let (blob, offs) = pack_ranks(ranks);
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { tk_ctx_create(blob.as_ptr(), offs.as_ptr(), ranks.len() as u32, num_special, bos, eos, device, &mut ctx) };
        if rc != TK_OK {
            return Err(map_err(rc, unsafe { tk_last_error(std::ptr::null()) }));
        }
        Ok(Self { ctx })
    }

    /// Memo of merged pieces: `log2_entries` 0 switches it off, 10..26 sizes the table (32-byte entries); `always` = no adaptive pause.
    /// The table can change how long a call takes, never an id (an entry is the exact key and the exact merge result).
    pub fn set_memo(&self, log2_entries: i32, always: bool) -> Result<(), HipError> {
        let rc = unsafe { tk_ctx_set_memo(self.ctx, log2_entries as c_int, always as c_int) };
        if rc != TK_OK {
            return Err(map_err(rc, unsafe { tk_last_error(self.ctx) }));
        }
        Ok(())
    }

    /// `Tekkenizer::encode` for ONE `&str` (the reference's own signature): no allocation inside the library, the ids land in
    /// a Vec sized for the worst case (one id per byte + BOS + EOS).
    pub fn encode(&self, text: &str, add_bos: bool, add_eos: bool) -> Result<Vec<u32>, HipError> {
        let mut ids: Vec<u32> = Vec::with_capacity(text.len() + 2);
        let mut n = 0u64;
        let rc = unsafe {
            tk_encode_one(self.ctx, text.as_ptr(), text.len() as u64, add_bos as c_int, add_eos as c_int, ids.as_mut_ptr(),
                          ids.capacity() as u64, &mut n)
        };
        if rc != TK_OK {
            return Err(map_err(rc, unsafe { tk_last_error(self.ctx) }));
        }
        unsafe { ids.set_len(n as usize) };
        Ok(ids)
    }

    pub fn encode_batch(&self, docs: &[&str], add_bos: bool, add_eos: bool) -> Result<Vec<Vec<u32>>, HipError> {
        let (bytes, offs) = pack_docs(docs);
        let mut res = TkResult { ids: std::ptr::null_mut(), offsets: std::ptr::null_mut(), n_ids: 0, n_docs: 0 };
        // &str is valid UTF-8 by construction => validate_utf8 = 0
        let rc = unsafe { tk_encode_batch(self.ctx, bytes.as_ptr(), offs.as_ptr(), docs.len() as u64, add_bos as c_int, add_eos as c_int, 0, &mut res) };
        if rc != TK_OK {
            return Err(map_err(rc, unsafe { tk_last_error(self.ctx) }));
        }
        Ok(unsafe { take(&mut res, docs.len()) })
    }
}
impl Drop for HipEngine {
    fn drop(&mut self) {
        unsafe { tk_ctx_destroy(self.ctx) }
    }
}

/// All GPUs of one node behind one call: the batch is cut into contiguous runs of whole documents balanced by BYTES, every GPU
/// tokenizes its run, the id buffers are gathered on the first device with direct peer -> root RCCL transfers (18 bits per id
/// on the wire) and come back in document order.
pub struct HipNode {
    node: *mut TkNode,
}
unsafe impl Send for HipNode {}
unsafe impl Sync for HipNode {}

impl HipNode {
    pub fn new(ranks: &[Vec<u8>], num_special: u32, bos: u32, eos: u32, devices: &[i32]) -> Result<Self, HipError> {
        let (blob, offs) = pack_ranks(ranks);
        let mut node = std::ptr::null_mut();
        let rc = unsafe {
            tk_node_create(blob.as_ptr(), offs.as_ptr(), ranks.len() as u32, num_special, bos, eos, devices.as_ptr(), devices.len() as c_int, &mut node)
        };
        if rc != TK_OK {
            return Err(map_err(rc, unsafe { tk_node_last_error(std::ptr::null()) }));
        }
        Ok(Self { node })
    }

    pub fn encode_batch(&self, docs: &[&str], add_bos: bool, add_eos: bool) -> Result<Vec<Vec<u32>>, HipError> {
        let (bytes, offs) = pack_docs(docs);
        let mut res = TkResult { ids: std::ptr::null_mut(), offsets: std::ptr::null_mut(), n_ids: 0, n_docs: 0 };
        let rc = unsafe { tk_node_encode_batch(self.node, bytes.as_ptr(), offs.as_ptr(), docs.len() as u64, add_bos as c_int, add_eos as c_int, &mut res) };
        if rc != TK_OK {
            return Err(map_err(rc, unsafe { tk_node_last_error(self.node) }));
        }
        Ok(unsafe { take(&mut res, docs.len()) })
    }

    /// The batch form for a host that encodes batch after batch: packed text + offsets in, ids + id offsets out, all four in
    /// buffers the caller keeps (pinned: `tk_host_alloc`) -- returns the number of ids written.
    pub fn encode_batch_into(&self, bytes: &[u8], doc_offsets: &[u64], add_bos: bool, add_eos: bool, ids_out: &mut [u32],
                             offsets_out: &mut [u64]) -> Result<usize, HipError> {
        assert!(!doc_offsets.is_empty() && offsets_out.len() >= doc_offsets.len());
        let mut n: u64 = 0;
        let rc = unsafe {
            tk_node_encode_batch_pinned(self.node, bytes.as_ptr(), doc_offsets.as_ptr(), (doc_offsets.len() - 1) as u64, add_bos as c_int,
                                        add_eos as c_int, ids_out.as_mut_ptr(), ids_out.len() as u64, offsets_out.as_mut_ptr(), &mut n)
        };
        if rc != TK_OK {
            return Err(map_err(rc, unsafe { tk_node_last_error(self.node) }));
        }
        Ok(n as usize)
    }
}
impl Drop for HipNode {
    fn drop(&mut self) {
        unsafe { tk_node_destroy(self.node) }
    }
}
