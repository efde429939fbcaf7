import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus, synth_vocab as sv
tk = importlib.import_module("tekken-rs_amd")
toks, ns, bos, eos = sv.load_tokens(sv.ensure_default())
eng = tk.Engine(toks, ns, bos, eos, device=0)
data, offs = corpus.generate("ascii", 1000000, 512, seed=corpus.BASE_SEED + 1)
import ctypes
out = np.zeros(len(data), np.uint8)
for _ in range(3):
    rc = tk.lib().tk_split_batch(eng._h, data.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), len(offs) - 1, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    assert rc == 0
print("starts", int(out.sum()))
