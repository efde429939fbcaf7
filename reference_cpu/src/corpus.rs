//! The seeded corpus generator of tools/corpus_gen.c, statement for statement (same PRNG: xoshiro256** seeded per
//! document with splitmix64; same draws in the same order), so that this binary and bench.py tokenize the SAME bytes.
//! `corpus_fnv1a` (main.rs) against `python tools/corpus_check.py` proves it before any id is compared.

pub const BASE_SEED: u64 = 0x7E44E2;
const N_WORDS: usize = 4096;
const MAX_WORD: usize = 16;

pub struct Rng {
    s: [u64; 4],
}

fn splitmix64(x: &mut u64) -> u64 {
    *x = x.wrapping_add(0x9E3779B97F4A7C15);
    let mut z = *x;
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^ (z >> 31)
}

impl Rng {
    pub fn new(seed: u64, idx: u64) -> Rng {
        let mut x = seed ^ idx.wrapping_mul(0xD1342543DE82EF95).wrapping_add(0x2545F4914F6CDD1D);
        let mut s = [0u64; 4];
        for v in s.iter_mut() {
            *v = splitmix64(&mut x);
        }
        Rng { s }
    }
    fn next(&mut self) -> u64 {
        let s = &mut self.s;
        let result = s[1].wrapping_mul(5).rotate_left(7).wrapping_mul(9);
        let t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = s[3].rotate_left(45);
        result
    }
    fn below(&mut self, n: u32) -> u32 {
        (((self.next() >> 32) * n as u64) >> 32) as u32
    }
    fn unit(&mut self) -> f64 {
        (self.next() >> 11) as f64 * (1.0 / 9007199254740992.0)
    }
}

pub struct Words {
    words: Vec<Vec<u8>>,
    cdf: Vec<f64>,
}

const ONSET: [&str; 36] = ["", "b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t", "v", "w", "y", "z", "th",
                           "sh", "ch", "st", "tr", "pr", "br", "cl", "gr", "pl", "fr", "wh", "qu", "sp", "sl", "dr"];
const NUCLEUS: [&str; 13] = ["a", "e", "i", "o", "u", "ea", "ou", "io", "ai", "ee", "oo", "ie", "au"];
const CODA: [&str; 21] = ["", "", "", "n", "r", "s", "t", "l", "d", "m", "ng", "nt", "st", "ck", "ll", "rd", "ss", "nd", "ly", "er", "ed"];
const COMMON: [&str; 100] = ["the", "of", "and", "to", "a", "in", "is", "that", "it", "was", "for", "on", "are", "as", "with", "his",
                             "they", "at", "be", "this", "have", "from", "or", "one", "had", "by", "word", "but", "not", "what",
                             "all", "were", "we", "when", "your", "can", "said", "there", "use", "an", "each", "which", "she", "do",
                             "how", "their", "if", "will", "up", "other", "about", "out", "many", "then", "them", "these", "so",
                             "some", "her", "would", "make", "like", "him", "into", "time", "has", "look", "two", "more", "write",
                             "go", "see", "number", "no", "way", "could", "people", "my", "than", "first", "water", "been", "call",
                             "who", "oil", "its", "now", "find", "long", "down", "day", "did", "get", "come", "made", "may", "part"];
const SYMS: &[u8] = b"!@#$%^&*()_+-={}[]|\\:;\"'<>,.?/";
const CONTR: [&str; 7] = ["'s", "'t", "'re", "'ve", "'m", "'ll", "'d"];

impl Words {
    pub fn new() -> Words {
        let mut r = Rng::new(0x7E44E2, 0xC0FFEE);
        let mut words: Vec<Vec<u8>> = COMMON.iter().map(|w| w.as_bytes().to_vec()).collect();
        while words.len() < N_WORDS {
            let n = words.len();
            let mut w: Vec<u8> = Vec::new();
            let mut syl = 1 + r.below(3) as i32;
            if n > 1024 {
                syl += r.below(2) as i32;
            }
            for _ in 0..syl {
                let a = ONSET[r.below(ONSET.len() as u32) as usize];
                let b = NUCLEUS[r.below(NUCLEUS.len() as u32) as usize];
                let c = CODA[r.below(CODA.len() as u32) as usize];
                w.extend_from_slice(a.as_bytes());
                w.extend_from_slice(b.as_bytes());
                w.extend_from_slice(c.as_bytes());
            }
            if w.len() < 2 || w.len() >= MAX_WORD {
                continue;
            }
            if words.iter().any(|k| *k == w) {
                continue;
            }
            words.push(w);
        }
        let mut tot = 0.0f64;
        for i in 0..N_WORDS {
            tot += 1.0 / (i + 1) as f64;
        }
        let mut acc = 0.0f64;
        let mut cdf = vec![0.0f64; N_WORDS];
        for i in 0..N_WORDS {
            acc += 1.0 / (i + 1) as f64 / tot;
            cdf[i] = acc;
        }
        cdf[N_WORDS - 1] = 1.0;
        Words { words, cdf }
    }
    fn zipf_word(&self, r: &mut Rng) -> usize {
        let u = r.unit();
        let (mut lo, mut hi) = (0usize, N_WORDS - 1);
        while lo < hi {
            let mid = (lo + hi) / 2;
            if self.cdf[mid] < u {
                lo = mid + 1;
            } else {
                hi = mid;
            }
        }
        lo
    }
}

/// bounded writer over one document's bytes
struct Wr<'a> {
    p: &'a mut [u8],
    n: usize,
}

impl<'a> Wr<'a> {
    fn cap(&self) -> usize {
        self.p.len()
    }
    fn put(&mut self, s: &[u8]) {
        for &b in s {
            if self.n < self.p.len() {
                self.p[self.n] = b;
                self.n += 1;
            }
        }
    }
    fn putc(&mut self, c: u8) {
        if self.n < self.p.len() {
            self.p[self.n] = c;
            self.n += 1;
        }
    }
    /// writes cp only if it fits entirely (never splits a code point)
    fn put_cp(&mut self, cp: u32) -> bool {
        let mut b = [0u8; 4];
        let l = if cp < 0x80 {
            b[0] = cp as u8;
            1
        } else if cp < 0x800 {
            b[0] = 0xC0 | (cp >> 6) as u8;
            b[1] = 0x80 | (cp & 0x3F) as u8;
            2
        } else if cp < 0x10000 {
            b[0] = 0xE0 | (cp >> 12) as u8;
            b[1] = 0x80 | ((cp >> 6) & 0x3F) as u8;
            b[2] = 0x80 | (cp & 0x3F) as u8;
            3
        } else {
            b[0] = 0xF0 | (cp >> 18) as u8;
            b[1] = 0x80 | ((cp >> 12) & 0x3F) as u8;
            b[2] = 0x80 | ((cp >> 6) & 0x3F) as u8;
            b[3] = 0x80 | (cp & 0x3F) as u8;
            4
        };
        if self.n + l > self.p.len() {
            return false;
        }
        self.p[self.n..self.n + l].copy_from_slice(&b[..l]);
        self.n += l;
        true
    }
}

fn ascii_word(ws: &Words, r: &mut Rng, w: &mut Wr) {
    let wi = ws.zipf_word(r);
    let c = r.below(100);
    let mut buf = ws.words[wi].clone();
    if c < 2 {
        for b in buf.iter_mut() {
            *b -= 32;
        }
    } else if c < 12 {
        buf[0] -= 32;
    }
    w.put(&buf);
}

fn ascii_sep(r: &mut Rng, w: &mut Wr) {
    let c = r.below(100);
    if c < 82 {
        w.putc(b' ');
    } else if c < 87 {
        w.put(b", ");
    } else if c < 92 {
        w.put(b". ");
    } else if c < 95 {
        w.putc(b'\n');
    } else if c < 96 {
        w.put(b"\n\n");
    } else if c < 98 {
        w.putc(b' ');
        let nd = 1 + r.below(6);
        for _ in 0..nd {
            let d = r.below(10);
            w.putc(b'0' + d as u8);
        }
        w.putc(b' ');
    } else if c < 99 {
        w.putc(b' ');
        let ns = 1 + r.below(8);
        for _ in 0..ns {
            let k = r.below(SYMS.len() as u32);
            w.putc(SYMS[k as usize]);
        }
        w.putc(b' ');
    } else {
        let s = CONTR[r.below(7) as usize];
        w.put(s.as_bytes());
        w.putc(b' ');
    }
}

fn gen_ascii(ws: &Words, r: &mut Rng, out: &mut [u8]) {
    let mut w = Wr { p: out, n: 0 };
    while w.n < w.cap() {
        ascii_word(ws, r, &mut w);
        ascii_sep(r, &mut w);
    }
}

fn pick_range(r: &mut Rng, lo: u32, hi: u32) -> u32 {
    lo + r.below(hi - lo + 1)
}

fn gen_mixed(ws: &Words, r: &mut Rng, out: &mut [u8]) {
    let mut w = Wr { p: out, n: 0 };
    let mut wt = [0u32; 5];
    wt[0] = 30 + r.below(40);
    wt[1] = r.below(25);
    wt[2] = r.below(25);
    wt[3] = r.below(25);
    wt[4] = r.below(12);
    let tot: u32 = wt.iter().sum();
    let mut guard = 0;
    while w.n < w.cap() && guard < 8 {
        let before = w.n;
        let mut c = r.below(tot);
        if c < wt[0] {
            ascii_word(ws, r, &mut w);
        } else if { c -= wt[0]; c < wt[1] } {
            let wi = ws.zipf_word(r);
            let word = ws.words[wi].clone();
            for &ch in word.iter() {
                if (ch == b'a' || ch == b'e' || ch == b'o' || ch == b'u' || ch == b'n') && r.below(3) == 0 {
                    const ACC: [u32; 13] = [0xE0, 0xE1, 0xE4, 0xE8, 0xE9, 0xEA, 0xF1, 0xF3, 0xF6, 0xFC, 0xFA, 0xC9, 0xD6];
                    let k = r.below(13) as usize;
                    if !w.put_cp(ACC[k]) {
                        break;
                    }
                } else {
                    w.putc(ch);
                }
            }
        } else if { c -= wt[1]; c < wt[2] } {
            let script = r.below(3);
            let n = 2 + r.below(9);
            for i in 0..n {
                let mut cp = if script == 0 { pick_range(r, 0x0430, 0x044F) } else if script == 1 { pick_range(r, 0x03B1, 0x03C9) } else { pick_range(r, 0x0627, 0x063A) };
                if i == 0 && script == 0 && r.below(5) == 0 {
                    cp -= 0x20;
                }
                if !w.put_cp(cp) {
                    break;
                }
            }
        } else if { c -= wt[2]; c < wt[3] } {
            let script = r.below(3);
            let n = 1 + r.below(10);
            for _ in 0..n {
                let cp = if script == 0 { pick_range(r, 0x4E00, 0x9FA5) } else if script == 1 { pick_range(r, 0xAC00, 0xD7A3) } else { pick_range(r, 0x0E01, 0x0E2E) };
                if !w.put_cp(cp) {
                    break;
                }
            }
        } else {
            let n = 1 + r.below(3);
            for _ in 0..n {
                let cp = if r.below(2) != 0 { pick_range(r, 0x1F600, 0x1F64F) } else { pick_range(r, 0x2200, 0x22FF) };
                if !w.put_cp(cp) {
                    break;
                }
            }
        }
        // separator
        let s = r.below(100);
        if s < 70 {
            w.putc(b' ');
        } else if s < 75 {
            w.put(b", ");
        } else if s < 79 {
            w.put(b". ");
        } else if s < 82 {
            w.putc(b'\n');
        } else if s < 84 {
            w.put(b"\r\n");
        } else if s < 86 {
            w.put_cp(0x00A0);
        } else if s < 87 {
            w.put_cp(0x2028);
        } else if s < 89 {
            w.put_cp(0x3000);
        } else if s < 90 {
            w.put_cp(0x3002);
        } else if s < 91 {
            w.put(b"\t");
        } else if s < 94 {
            w.putc(b' ');
            let nd = 1 + r.below(5);
            let kind = r.below(4);
            for _ in 0..nd {
                let cp = if kind == 0 {
                    pick_range(r, 0x0660, 0x0669)
                } else if kind == 1 {
                    pick_range(r, 0xFF10, 0xFF19)
                } else if kind == 2 {
                    b'0' as u32 + r.below(10)
                } else if r.below(3) == 0 {
                    0x00B2
                } else if r.below(2) != 0 {
                    0x2167
                } else {
                    0x00BD
                };
                if !w.put_cp(cp) {
                    break;
                }
            }
            w.putc(b' ');
        } else if s < 96 {
            let k = r.below(9);
            if k == 7 {
                w.putc(b'\'');
                w.put_cp(0x017F);
            } else if k == 8 {
                w.put(b"'S");
            } else {
                w.put(CONTR[k as usize].as_bytes());
            }
            w.putc(b' ');
        } else if s < 98 {
            w.putc(b' ');
            let ns = 1 + r.below(4);
            for _ in 0..ns {
                let k = r.below(SYMS.len() as u32);
                w.putc(SYMS[k as usize]);
            }
            if r.below(3) == 0 {
                w.putc(b'\n');
            }
        } else {
            w.put_cp(0x0301);
            w.putc(b' ');
        }
        guard = if w.n == before { guard + 1 } else { 0 };
    }
    while w.n < w.cap() {
        let n = w.n;
        w.p[n] = b' ';
        w.n += 1;
    }
}

fn zipf_len(r: &mut Rng) -> u64 {
    let a = 16.0f64.powf(-0.2);
    let b = 32768.0f64.powf(-0.2);
    let u = r.unit();
    let x = (a - u * (a - b)).powf(-5.0);
    let l = x as u64;
    l.clamp(16, 32768)
}

fn gen_zipf_doc(ws: &Words, r: &mut Rng, out: &mut [u8]) {
    let special = r.below(1000);
    if special == 0 {
        for o in out.iter_mut() {
            *o = b'a' + r.below(26) as u8;
        }
    } else if special == 1 {
        for o in out.iter_mut() {
            let c = r.below(40);
            *o = if c == 0 { b'\n' } else if c == 1 { b'\t' } else { b' ' };
        }
    } else {
        gen_ascii(ws, r, out);
    }
}

/// kind: 0 ascii, 1 mixed, 2 zipf.  -> (packed bytes, offsets[n_docs + 1])
pub fn generate(kind: i32, seed: u64, first_doc: u64, n_docs: u64, doc_len: u64) -> (Vec<u8>, Vec<u64>) {
    let ws = Words::new();
    let mut offs = vec![0u64; n_docs as usize + 1];
    for d in 0..n_docs {
        let len = if kind != 2 { doc_len } else { zipf_len(&mut Rng::new(seed, first_doc + d)) };
        offs[d as usize + 1] = offs[d as usize] + len;
    }
    let mut data = vec![0u8; offs[n_docs as usize] as usize];
    for d in 0..n_docs {
        let mut r = Rng::new(seed, first_doc + d);
        let out = &mut data[offs[d as usize] as usize..offs[d as usize + 1] as usize];
        match kind {
            0 => gen_ascii(&ws, &mut r, out),
            1 => gen_mixed(&ws, &mut r, out),
            _ => {
                let _ = zipf_len(&mut r);
                gen_zipf_doc(&ws, &mut r, out)
            }
        }
    }
    (data, offs)
}
