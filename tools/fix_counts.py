#!/usr/bin/env python3
"""Rewrites the two test counts DESIGN.md quotes (profiles/r0N_quoted.json: test_counts) to what pytest collects now.
tests/test_docs_quote_profiles.py is the check; this is the pen (run after adding or removing tests)."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_docs_quote_profiles as T
qpath = T.QUOTED_PATH if hasattr(T, "QUOTED_PATH") else os.path.join(ROOT, "profiles", "r06_quoted.json")
Q = json.load(open(qpath))
for c in Q.get("test_counts", []):
    have = T._collected(c["marker"])
    old = c["quote"]
    nums = list(re.finditer(r"\d+", old))
    m = nums[c.get("which", 0)]
    new = old[:m.start()] + str(have) + old[m.end():]
    if new != old:
        doc = os.path.join(ROOT, c["doc"])
        text = open(doc).read()
        assert old in text, (c["doc"], old)
        open(doc, "w").write(text.replace(old, new))
        c["quote"] = new
        print(f"{c['doc']}: {old!r} -> {new!r}")
json.dump(Q, open(qpath, "w"), indent=1)
