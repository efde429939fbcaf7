#!/usr/bin/env python3
"""GPU probe of the memo of merged pieces: fresh seeded batches through one context, per call the device times and the memo's
look-ups / hits.  python tools/memo_probe.py [--vocab-fit heldout] [--kind mixed --doc-len 2048] [--docs N] [--batches B] [--repeat]"""
import argparse, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, corpus, synth_vocab as sv
tk = importlib.import_module("tekken-rs_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--vocab-fit", default="same"); ap.add_argument("--kind", default="ascii"); ap.add_argument("--doc-len", type=int, default=512)
ap.add_argument("--docs", type=int, default=1000000); ap.add_argument("--batches", type=int, default=4); ap.add_argument("--repeat", action="store_true")
ap.add_argument("--ablate", default="", help="development build only: comma list of TK_DEBUG_ABLATE values, applied one per extra pass over the LAST batch")
ap.add_argument("--log2", type=int, default=24); ap.add_argument("--policy", type=int, default=1); ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
vp = sv.ensure_heldout() if a.vocab_fit == "heldout" else sv.ensure_default()
tokz = tk.Tekkenizer.from_file(vp, device=0); eng = tokz.engine()
eng.set_memo(a.log2, a.policy)
stream = torch.cuda.current_stream().cuda_stream
for b in range(a.batches):
    data, offs = corpus.generate(a.kind, a.docs, a.doc_len, seed=corpus.BASE_SEED + 1 + (0 if a.repeat else 100 * b), threads=64)
    d_b = torch.from_numpy(data).cuda(); d_o = torch.from_numpy(offs.astype(np.int64)).cuda()
    for rep in range(a.reps):
        eng.encode_batch_device_views(d_b.data_ptr(), d_o.data_ptr(), a.docs, int(offs[-1]), True, True, stream)
        t = eng.last_timing(); m = eng.memo_stats()
        print(json.dumps({"batch": b, "rep": rep, "pipeline_ms": round(t["pipeline_ms"], 3), "flat_ms": round(t["encode_kernel_ms"], 3), "merge_ms": round(t["merge_ms"], 3),
                          "lookups": m["lookups_last"], "hits": m["hits_last"], "rate": round(m["hits_last"] / max(1, m["lookups_last"]), 4), "active": m["active_last"]}), flush=True)

for v in [x for x in a.ablate.split(",") if x]:
    os.environ["TK_DEBUG_ABLATE"] = v
    for rep in range(a.reps):
        eng.encode_batch_device_views(d_b.data_ptr(), d_o.data_ptr(), a.docs, int(offs[-1]), True, True, stream)
        t = eng.last_timing(); m = eng.memo_stats()
        print(json.dumps({"ablate": int(v), "rep": rep, "pipeline_ms": round(t["pipeline_ms"], 3), "merge_ms": round(t["merge_ms"], 3), "hits": m["hits_last"]}), flush=True)
