#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's assembly.  usage: tools/isa_mix.py <file.s> <mangled-name-substring> [--dump out.s]
(assembly: hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -S csrc/dmel_fwd.hip -o /tmp/fwd.s)"""
import collections
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".section") or lines[i].strip().startswith(".end_amdhsa_kernel") or (i > start and lines[i].startswith("_ZN")))
body = [l.strip() for l in lines[start + 1:end]]
ins = [l for l in body if l and not l.startswith((";", ".")) and not l.split(";")[0].strip().endswith(":")]
if "--dump" in sys.argv:
    open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(lines[start:end]))
c = collections.Counter()
for l in ins:
    op = l.split()[0]
    k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
         "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "other")
    c[k] += 1
print(len(ins), "instructions (static):", dict(c))
ops = collections.Counter(l.split()[0] for l in ins if l.startswith("v_") and not l.startswith("v_mfma"))
for k, v in ops.most_common(60):
    print(f"  {k:30s}{v}")
