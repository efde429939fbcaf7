"""Sanitizers where they can run: the CPU build (GPU AddressSanitizer / XNACK runs are not available on the pool).

* `test_device_source_under_asan_ubsan`: tests/emu/libtk_emu_asan.so is the VERY device source (csrc/tk_flat_impl.h,
  tk_encode_impl.h, tk_long_impl.h: every LDS index, queue record, table probe, and the workgroup merges' scratch) compiled for the CPU wave emulator with
  -fsanitize=address,undefined, and oracle/libtk_oracle_asan.so the oracle; a subset of the emulator / oracle suites is
  re-run against them in a child interpreter (libasan preloaded).  TK_SANITIZE_FULL=1 runs all of test_flat_path.py,
  test_kernel_emu.py, test_oracle_golden.py and test_merge_golden.py that way (seven to ten minutes here; last run clean).
* `test_loaders_fuzzed_under_asan_ubsan`: tests/fuzz/fuzz_loaders -- the hand-written JSON / base64 reader and the two
  cache side-file loaders of the host side, fed truncated / bit-flipped / spliced inputs.  A damaged side file must
  never change what a load returns (both files end with a checksum of their payload).
"""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not p or not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("no libasan.so next to this gcc")
    return p


def test_device_source_under_asan_ubsan():
    asan = _libasan()
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu"), "libtk_emu_asan.so"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libtk_oracle_asan.so"])
    env = dict(os.environ, TK_TEST_SANITIZE="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    if os.environ.get("TK_SANITIZE_FULL"):
        sel = ["tests/test_flat_path.py", "tests/test_kernel_emu.py", "tests/test_oracle_golden.py", "tests/test_merge_golden.py",
               "tests/test_long_merge_emu.py"]
        extra = []
    else:
        # the flat chunk kernel + both merge kernels on UTF-8 / run-heavy / dense-piece streams and on the adversarial merge
        # vocabularies; the per-document kernels against the oracle; the oracle against its golden vectors
        sel = ["tests/test_flat_path.py", "tests/test_kernel_emu.py", "tests/test_oracle_golden.py", "tests/test_merge_golden.py",
               "tests/test_long_merge_emu.py"]
        extra = ["-k", "test_block_merges_on_repetitive or test_emu_cut_decomposition or test_emu_flat_utf8 or test_emu_flat_runs_and_misses or test_emu_flat_dense_pieces or test_emu_flat_baseline_shapes "
                       "or test_emu_memo_of_merged_pieces or test_emu_flat_handback_and_mixed or test_emu_flat_small_alphabet_packed or test_emu_flat_every_ascii_byte_pair "
                       "or test_emu_long_single_piece or test_emu_split_only or test_emu_small_vocab_known_answer "
                       "or test_emu_reference_vectors_on_consistent_vocab or test_split_matches_independent_engine "
                       "or test_small_vocab_known_answer or test_reference_vectors_on_consistent_vocab or test_encode_properties "
                       "or test_batch_equals_single or test_oracle_matches_independent_merge_vectors"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + sel + extra,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail


def test_loaders_fuzzed_under_asan_ubsan(tmp_path):
    _libasan()
    sys.path.insert(0, HERE)
    from test_host_tokenizer import model
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "fuzz"), "fuzz_loaders"])
    toks = [bytes([i]) for i in range(256)] + [b"hello", b"world", b"he", b"ll", b"wor", b" the", "é".encode(), "中文".encode(),
                                               b"a" * 17, b"tokenizer tokens"]
    m = model(toks, specials=("<unk>", "<s>", "</s>", "[INST]", "é☃", "[AUDIO]", "[BEGIN_AUDIO]"))
    m["config"]["pattern"] = "[^\\r\\n\\p{L}\\p{N}]?\\p{L}+"
    m["audio"] = {"sampling_rate": 16000, "frame_rate": 12.5, "audio_encoding_config": {"num_mel_bins": 128, "hop_length": 160, "window_size": 400},
                  "chunk_length_s": None}
    seed = tmp_path / "seed.json"
    seed.write_text(json.dumps(m, ensure_ascii=False))
    for s in (1, 2):
        r = subprocess.run([os.path.join(HERE, "fuzz", "fuzz_loaders"), str(seed), str(tmp_path / ("scratch%d" % s)), "2500", str(s)],
                           capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"))
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert "json: 2500 mutants" in r.stdout and "tables side file" in r.stdout
