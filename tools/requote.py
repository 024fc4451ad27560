#!/usr/bin/env python3
"""After a re-measurement: rewrite the figures DESIGN.md quotes (profiles/r06_quoted.json) so that they are the figures of the committed
files again, keeping each number's format.  tests/test_docs_quote_profiles.py is the check; this is the pen.
Handles both kinds of entries: `which` (index of the number inside the quoted text) and `find` (the exact number text inside it; a `find` such as
"0.61 at" keeps its trailing words).  Entries with digit groups separated by spaces (`strip_spaces`) are reported, not rewritten.
Every entry that shares one quoted text is applied to that text together; the entries' `quote` / `find` are updated with the document."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_docs_quote_profiles import resolve
QP = os.path.join(ROOT, "profiles", "r06_quoted.json")
Q = json.load(open(QP))


def fmt(old: str, have: float) -> str:
    if "e" in old:
        mant = old.split("e")[0]
        dec = len(mant.split(".")[1]) if "." in mant else 0
        m, e = f"{have:.{dec}e}".split("e")
        return m + "e" + str(int(e))
    dec = len(old.split(".")[1]) if "." in old else 0
    return f"{have:.{dec}f}"


groups = {}
for q in Q["quotes"]:
    groups.setdefault((q["doc"], q["quote"]), []).append(q)
docs, manual = {}, []
for (doc, quote), qs in groups.items():
    text = docs.setdefault(doc, open(os.path.join(ROOT, doc)).read())
    if quote not in text:
        manual.append((doc, quote, "text not found"))
        continue
    edits = []                                                    # (start, end, new text, entry)
    for q in qs:
        have = float(resolve(json.load(open(os.path.join(ROOT, q["file"]))), q["path"])) * q.get("scale", 1.0)
        pat = r"\d+(?:\.\d+)?(?:e-?\d+)?" if q.get("sci") else r"\d+(?:\.\d+)?"
        if q.get("strip_spaces"):
            manual.append((doc, quote, f"{q['path']} -> {have:.6g} (digit groups: by hand)"))
            continue
        if "find" in q:
            at = quote.find(q["find"])
            m = re.search(pat, q["find"])
            if at < 0 or not m:
                manual.append((doc, quote, f"find {q['find']!r} not in the quote"))
                continue
            a, b = at + m.start(), at + m.end()
        else:
            spans = [mm.span() for mm in re.finditer(r"-?" + pat, quote)]
            a, b = spans[q.get("which", 0)]
        old = quote[a:b]
        said = float(old)
        tol = q.get("rel_tol", 0.01) * abs(have) + q.get("abs_tol", 0.0)
        if abs(said - have) <= 0.5 * tol:
            continue                                              # comfortably inside: leave the text alone
        edits.append((a, b, fmt(old.lstrip("-"), abs(have)) if old.startswith("-") and have >= 0 else fmt(old, have), q))
    if not edits:
        continue
    new = quote
    for a, b, s, q in sorted(edits, key=lambda e: e[0], reverse=True):
        if "find" in q:
            q["_newfind"] = q["find"].replace(quote[a:b], s, 1)
        new = new[:a] + s + new[b:]
    if new != quote:
        docs[doc] = docs[doc].replace(quote, new)
        for q in qs:
            q["quote"] = new
            if "_newfind" in q:
                q["find"] = q.pop("_newfind")
        print(f"{doc}: {quote!r}\n   -> {new!r}")
for q in Q["quotes"]:
    q.pop("_newfind", None)
for doc, text in docs.items():
    open(os.path.join(ROOT, doc), "w").write(text)
json.dump(Q, open(QP, "w"), indent=1)
for m in manual:
    print("MANUAL:", m)
