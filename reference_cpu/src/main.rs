//! reference_cpu: upstream `Tekkenizer::encode` (tekken-rs src/tekkenizer.rs:378-405) over the bench corpora.
//! See README.md.  Not compiled in the build image.
mod corpus;

use std::sync::Arc;
use std::time::Instant;
use tekken::Tekkenizer;

fn fnv1a_bytes(h: &mut u64, bytes: &[u8]) {
    for &b in bytes {
        *h ^= b as u64;
        *h = h.wrapping_mul(1099511628211);
    }
}

fn fnv1a_ids(h: &mut u64, ids: &[u32]) {
    for &v in ids {
        fnv1a_bytes(h, &v.to_le_bytes());
    }
}

struct Args {
    vocab: String,
    kind: String,
    docs: u64,
    doc_len: u64,
    first_doc: u64,
    seed: Option<u64>,
    threads: usize,
    passes: usize,
    add_bos: bool,
    add_eos: bool,
}

fn parse_args() -> Args {
    let mut a = Args { vocab: String::new(), kind: "ascii".into(), docs: 1_000_000, doc_len: 512, first_doc: 0, seed: None,
                       threads: 1, passes: 1, add_bos: true, add_eos: true };
    let v: Vec<String> = std::env::args().collect();
    let mut i = 1;
    while i < v.len() {
        let val = |i: usize| v.get(i + 1).cloned().unwrap_or_else(|| panic!("{} needs a value", v[i]));
        match v[i].as_str() {
            "--vocab" => { a.vocab = val(i); i += 1; }
            "--kind" => { a.kind = val(i); i += 1; }
            "--docs" => { a.docs = val(i).parse().unwrap(); i += 1; }
            "--doc-len" => { a.doc_len = val(i).parse().unwrap(); i += 1; }
            "--first-doc" => { a.first_doc = val(i).parse().unwrap(); i += 1; }
            "--seed" => { let s = val(i); a.seed = Some(u64::from_str_radix(s.trim_start_matches("0x"), 16).unwrap()); i += 1; }
            "--threads" => { a.threads = val(i).parse().unwrap(); i += 1; }
            "--passes" => { a.passes = val(i).parse().unwrap(); i += 1; }
            "--no-bos" => a.add_bos = false,
            "--no-eos" => a.add_eos = false,
            other => panic!("unknown argument {other}"),
        }
        i += 1;
    }
    if a.vocab.is_empty() {
        a.vocab = std::env::var("TEKKEN_JSON").expect("--vocab or TEKKEN_JSON");
    }
    a
}

fn main() {
    let a = parse_args();
    let kind = match a.kind.as_str() { "ascii" => 0, "mixed" => 1, "zipf" => 2, k => panic!("kind {k}") };
    // bench.py's seed convention: BASE_SEED + 1 for the single-GPU run of every kind
    let seed = a.seed.unwrap_or(corpus::BASE_SEED + 1);
    let t0 = Instant::now();
    let (data, offs) = corpus::generate(kind, seed, a.first_doc, a.docs, a.doc_len);
    let mut hc = 1469598103934665603u64;
    fnv1a_bytes(&mut hc, &data);
    eprintln!("corpus: {} docs, {} bytes, {:.2} s, corpus_fnv1a {:016x}", a.docs, data.len(), t0.elapsed().as_secs_f64(), hc);

    let tok = Arc::new(Tekkenizer::from_file(&a.vocab).expect("from_file"));      // src/tekkenizer.rs:222-248
    let docs: Vec<&str> = (0..a.docs as usize)
        .map(|d| std::str::from_utf8(&data[offs[d] as usize..offs[d + 1] as usize]).expect("generator emits valid UTF-8"))
        .collect();

    // single thread (the north star's baseline), ids fingerprinted in document order
    let mut best = f64::MAX;
    let mut h = 0u64;
    let mut n_ids = 0u64;
    for _ in 0..a.passes.max(1) {
        h = 1469598103934665603u64;
        n_ids = 0;
        let t = Instant::now();
        for d in &docs {
            let ids = tok.encode(d, a.add_bos, a.add_eos).expect("encode");     // src/tekkenizer.rs:378-405
            n_ids += ids.len() as u64;
            fnv1a_ids(&mut h, &ids);
        }
        best = best.min(t.elapsed().as_secs_f64());
    }
    let mbs = data.len() as f64 / 1e6 / best;

    // N threads: one Tekkenizer shared (&self is Sync, tests/test_tokenizer_output.rs:5-12), documents striped
    let mut mbs_nt = 0.0;
    if a.threads > 1 {
        let t = Instant::now();
        std::thread::scope(|s| {
            for w in 0..a.threads {
                let tok = Arc::clone(&tok);
                let docs = &docs;
                let (bos, eos, nt) = (a.add_bos, a.add_eos, a.threads);
                s.spawn(move || {
                    let mut n = 0usize;
                    let mut d = w;
                    while d < docs.len() {
                        n += tok.encode(docs[d], bos, eos).expect("encode").len();
                        d += nt;
                    }
                    std::hint::black_box(n);
                });
            }
        });
        mbs_nt = data.len() as f64 / 1e6 / t.elapsed().as_secs_f64();
    }
    println!(
        "{{\"reference\": \"tekken-rs Tekkenizer::encode (tiktoken-rs CoreBPE)\", \"kind\": \"{}\", \"docs\": {}, \"input_bytes\": {}, \
         \"ids_total\": {}, \"MBps_1_thread\": {:.1}, \"threads\": {}, \"MBps_n_threads\": {:.1}, \"fnv1a_ids\": \"{:016x}\", \
         \"corpus_fnv1a\": \"{:016x}\", \"add_bos\": {}, \"add_eos\": {}}}",
        a.kind, a.docs, data.len(), n_ids, mbs, a.threads, mbs_nt, h, hc, a.add_bos, a.add_eos
    );
}
