#!/usr/bin/env python3
"""Which code points does the split's class table (generated from Python `regex`, tools/gen_unicode_tables.py) classify differently
from Unicode 13.0 (python's unicodedata in this image)?  The reference's regex-syntax crate is unpinned (Cargo.toml:40 pulls it in
through tiktoken-rs / fancy-regex), so ITS Unicode version is whatever the maintainer's lock file had: the classes L / N / White_Space
of a code point assigned after that version are the one place where the reference's split and ours may differ without either being
wrong (trap T11).  This writes the list (ranges) to tests/golden/unicode_drift.json; tests/test_oracle_golden.py
asserts that the bench corpora's alphabets and every golden text stay clear of it.

  python tools/unicode_drift.py [--check]"""
import json
import os
import sys
import unicodedata

import regex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "unicode_drift.json")
# the first code point (one of them) that each Unicode version added as a letter / number: the newest one `regex` knows dates its tables
PROBES = [("13.0", 0x10E80), ("14.0", 0x1E290), ("15.0", 0x11F04), ("15.1", 0x2EBF0), ("16.0", 0x10D4A), ("17.0", 0x16EA0)]


def regex_unicode_version():
    rl = regex.compile(r"[\p{L}\p{N}]")
    v = "< 13.0"
    for name, cp in PROBES:
        if rl.match(chr(cp)):
            v = name
    return v


def cls13(cp):
    c = unicodedata.category(chr(cp))
    if c[0] == "L":
        return 1
    if c[0] == "N":
        return 2
    return 0


def main():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_unicode_tables as g
    cls = g.classify_all()
    # White_Space (the Rust regex crate's \s) has not changed since Unicode 6.3: the table's set must be exactly these 25
    ws = [9, 10, 11, 12, 13, 32, 0x85, 0xA0, 0x1680] + list(range(0x2000, 0x200B)) + [0x2028, 0x2029, 0x202F, 0x205F, 0x3000]
    got_ws = [cp for cp in range(0x110000) if cls[cp] == 3]
    assert got_ws == sorted(ws), "the table's white space is not Unicode's White_Space property"
    ranges = []
    for cp in range(0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        now = cls[cp] if cls[cp] != 3 else 0
        old = cls13(cp)
        if now != old:
            if ranges and ranges[-1][1] == cp - 1 and ranges[-1][2] == old and ranges[-1][3] == now:
                ranges[-1][1] = cp
            else:
                ranges.append([cp, cp, old, now])
    res = {"generator": "tools/unicode_drift.py", "regex_module": regex.__version__, "regex_unicode_version": regex_unicode_version(),
           "unicodedata_version": unicodedata.unidata_version, "classes": "0 = other, 1 = L, 2 = N (White_Space is identical: 25 code points)",
           "code_points": sum(r[1] - r[0] + 1 for r in ranges), "ranges_lo_hi_class13_classNow": ranges}
    if "--check" in sys.argv:
        old = json.load(open(OUT))
        assert old["ranges_lo_hi_class13_classNow"] == ranges, "tests/golden/unicode_drift.json is stale"
        print("unicode_drift.json is current: %d code points in %d ranges" % (res["code_points"], len(ranges)))
        return
    with open(OUT, "w") as f:
        json.dump(res, f, separators=(",", ":"))
    print("regex %s = Unicode %s; unicodedata %s; %d code points differ (%d ranges)" % (res["regex_module"], res["regex_unicode_version"],
          res["unicodedata_version"], res["code_points"], len(ranges)))


if __name__ == "__main__":
    main()
