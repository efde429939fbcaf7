"""Load-time measurement for SURVEY section 8 row f-2 (the reference's tests/test_full_vocab_profile.rs and
test_detailed_profile.rs time exactly this): Tekkenizer.from_file on the bench-size tekken.json, with and without the
side-file cache (TK_TABLE_CACHE_DIR).  Prints one JSON line.   python tools/load_time.py [--device 0|-1]"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import synth_vocab as sv
    tk = importlib.import_module("tekken-rs_amd")
    path = sv.ensure_default()
    if a.device >= 0:
        import torch
        torch.zeros(1, device="cuda:%d" % a.device)   # device start-up is not load time

    def once():
        t0 = time.perf_counter()
        t = tk.Tekkenizer.from_file(path, device=a.device)
        dt = time.perf_counter() - t0
        if a.device >= 0:
            assert t.encode("hello world 123", True, True)
        t.close()
        return dt

    os.environ.pop("TK_TABLE_CACHE_DIR", None)
    once()
    plain = sorted(once() for _ in range(a.reps))
    with tempfile.TemporaryDirectory() as d:
        os.environ["TK_TABLE_CACHE_DIR"] = d
        fill = once()
        cached = sorted(once() for _ in range(a.reps))
        size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    print(json.dumps({"what": "Tekkenizer.from_file", "file_bytes": os.path.getsize(path), "device": a.device,
                      "uncached_s_median": round(plain[len(plain) // 2], 4), "cache_fill_s": round(fill, 4),
                      "cached_s_median": round(cached[len(cached) // 2], 4), "cache_bytes": size,
                      "speedup": round(plain[len(plain) // 2] / cached[len(cached) // 2], 1)}))


if __name__ == "__main__":
    main()
