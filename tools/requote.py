#!/usr/bin/env python3
"""After a re-measurement: rewrite the figures DESIGN.md / profiles/README.md quote (profiles/r06_quoted.json) so that they are the figures
of the committed files again, keeping each number's format.  tests/test_docs_quote_profiles.py is the check; this is the pen.
Entries whose text holds digit groups separated by spaces (strip_spaces) are reported, not rewritten."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_docs_quote_profiles import resolve
QP = os.path.join(ROOT, "profiles", "r06_quoted.json")
Q = json.load(open(QP))
groups = {}
for q in Q["quotes"]:
    groups.setdefault((q["doc"], q["quote"]), []).append(q)
docs = {}
manual = []
for (doc, quote), qs in groups.items():
    text = docs.setdefault(doc, open(os.path.join(ROOT, doc)).read())
    if quote not in text:
        manual.append((doc, quote, "text not found")); continue
    if any(q.get("strip_spaces") for q in qs):
        for q in qs:
            have = float(resolve(json.load(open(os.path.join(ROOT, q["file"]))), q["path"])) * q.get("scale", 1.0)
            manual.append((doc, quote, f"which {q.get('which', 0)} -> {have:.6g}"))
        continue
    sci = any(q.get("sci") for q in qs)
    pat = r"-?\d+(?:\.\d+)?(?:e-?\d+)?" if sci else r"-?\d+(?:\.\d+)?"
    spans = [m.span() for m in re.finditer(pat, quote)]
    new = quote
    repl = {}
    for q in qs:
        have = float(resolve(json.load(open(os.path.join(ROOT, q["file"]))), q["path"])) * q.get("scale", 1.0)
        if q.get("transform") == "spread_percent":
            have = (have - 1.0) * 100.0
        a, b = spans[q.get("which", 0)]
        old = quote[a:b]
        said = float(old)
        tol = 0.06 if q.get("transform") == "spread_percent" else q.get("rel_tol", 0.01) * abs(have) + q.get("abs_tol", 0.0)
        if abs(said - have) <= tol * 0.5:
            continue                                              # comfortably inside: leave the text alone
        if "e" in old:
            mant = old.split("e")[0]
            dec = len(mant.split(".")[1]) if "." in mant else 0
            s = f"{have:.{dec}e}"
            m, e = s.split("e")
            s = m + "e" + str(int(e))
        else:
            dec = len(old.split(".")[1]) if "." in old else 0
            s = f"{have:.{dec}f}"
        repl[(a, b)] = s
    for (a, b), s in sorted(repl.items(), reverse=True):
        new = new[:a] + s + new[b:]
    if new != quote:
        assert text.count(quote) >= 1
        docs[doc] = text.replace(quote, new)
        for q in qs:
            q["quote"] = new
        print(f"{doc}: {quote!r} -> {new!r}")
for doc, text in docs.items():
    open(os.path.join(ROOT, doc), "w").write(text)
json.dump(Q, open(QP, "w"), indent=1)
for m in manual:
    print("MANUAL:", m)
