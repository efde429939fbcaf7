"""pytest configuration: markers, import paths and shared fixtures.

`-m "not gpu"`  : oracle vs golden vectors, split-rule model, the device source on the CPU wave
                  emulator, host-side loader / decode logic, C-ABI symbol check.   (no GPU needed)
`-m gpu`        : parity tests proper -- the HIP path through the C ABI vs the oracle.
"""
import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"),
          os.path.join(ROOT, "tests", "emu")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tk():
    """The product package (directory name has a hyphen)."""
    return importlib.import_module("tekken-rs_amd")


@pytest.fixture(scope="session")
def golden():
    d = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(d, "split_vectors.json")) as f:
        split = json.load(f)
    with open(os.path.join(d, "reference_vectors.json")) as f:
        ref = json.load(f)
    return {"split": split, "ref": ref}


@pytest.fixture(scope="session")
def small_vocab():
    """The construction of reference tests/test_small_vocab.rs:11-67: 256 bytes + hello + world."""
    toks = [bytes([i]) for i in range(256)] + [b"hello", b"world"]
    return {"tokens": toks, "num_special": 10, "bos": 1, "eos": 2}


@pytest.fixture(scope="session")
def test_vocab():
    """A small trained vocabulary (fast to build) used by the CPU-side differential tests."""
    import helpers
    return helpers.small_trained_vocab()


@pytest.fixture(scope="session")
def bench_vocab():
    """The 130072-rank synthetic vocabulary the bench uses (same size class as the reference's asset)."""
    import synth_vocab as sv
    toks, ns, bos, eos = sv.load_tokens(sv.ensure_default())
    return {"tokens": toks, "num_special": ns, "bos": bos, "eos": eos, "path": sv.default_path()}
