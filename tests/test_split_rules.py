"""The LOCAL piece-start rules (tools/split_rules_model.py, what the HIP kernel implements with
ballots) against the oracle and the golden vectors, including the sliding-window commit logic."""
import itertools

import helpers
import split_rules_model as M
import tk_oracle


def test_rules_on_golden(golden):
    for c in golden["split"]["cases"]:
        doc = c["text"].encode("utf-8")
        got, _ = M.rule_split(doc)
        assert got == c["starts"], c["text"]


def test_rules_exhaustive_small_alphabet():
    alpha = ["a", "s", "1", "'", "!", " ", "\n", "\t"]
    for n in range(0, 6):
        for tup in itertools.product(alpha, repeat=n):
            doc = "".join(tup).encode()
            assert M.rule_split(doc)[0] == tk_oracle.split(doc), doc


def test_window_commit_logic():
    for doc in helpers.random_unicode_docs(1500, seed=3, max_len=60):
        exp = tk_oracle.split(doc)
        for w in (8, 13, 64):
            got, _ = M.window_split(doc, w)
            assert got == exp, (doc, w)
