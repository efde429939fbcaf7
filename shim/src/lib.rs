//! tekken-hip: `extern "C"` declarations of include/tekken_hip.h and the safe wrappers tekken-rs calls.
//!
//! Where it plugs into the reference (tekken-rs):
//!   * `Tekkenizer::new` (src/tekkenizer.rs:122-126): next to `CoreBPE::new(..)` build `hip::HipEngine::new(&ranks_by_rank,
//!     num_special_tokens, bos, eos, device)` from the same validated rank table;
//!   * `Tekkenizer::encode` (src/tekkenizer.rs:378-405): replace `self.tekkenizer.encode(text, &HashSet::new())`, the id
//!     shift (:390-392) and the BOS / EOS insertion (:394-402) by `self.hip.encode(text, bos, eos)` -- ids come back final.
//!     Keep `self.bos_id()?` / `self.eos_id()?` in front so that `TokenNotFound` is raised exactly where it was;
//!   * new `Tekkenizer::encode_batch(&self, &[&str], bool, bool) -> Result<Vec<Vec<u32>>>` (no reference equivalent);
//!   * `hip::HipNode` for all GPUs of a node behind the same call (documents sharded whole, one RCCL gather).
//! Not compiled in the build image.
pub mod ffi;
pub mod hip;

pub use hip::{HipEngine, HipError, HipNode};
