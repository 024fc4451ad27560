#!/usr/bin/env python3
"""Static instruction mix of the fused forward's training kernels, before / after: tools/isa_mix_report.py <dir with r4_<N>.s and r5_<N>.s> <out.json>
(assembly: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDMEL_ONLY_NFFT=<N> -x hip --cuda-device-only -S csrc/dmel_fwd.hip; "r4" = the file at
the round-4 commit 11d74c4).  Static counts of the kernel's text -- loops are counted once, both sides of a branch are counted -- so they sit
beside, not in place of, the executed counts of profiles/r05_pmc_sq_c2.json (SQ_INSTS_VALU per wave)."""
import collections, json, re, sys

def mix(path, key):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and re.search(key, l.split(":")[0]) and l.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith(".section") or lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].startswith("_ZN"))
    ins = [l.strip() for l in lines[start + 1:end]]
    ins = [l for l in ins if l and not l.startswith((";", ".")) and not l.split(";")[0].strip().endswith(":")]
    c = collections.Counter()
    ops = collections.Counter()
    for l in ins:
        op = l.split()[0]
        k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
             "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "other")
        c[k] += 1
        if k == "valu":
            ops[op] += 1
    groups = collections.Counter()
    for op, n in ops.items():
        g = ("packed fma / mul / add" if op.startswith("v_pk_") else "moves" if op.startswith(("v_mov", "v_accvgpr", "v_swap")) else
             "selects" if op.startswith("v_cndmask") else "sign flips / bit ops" if op.startswith(("v_xor", "v_and", "v_or", "v_bfe", "v_lshl", "v_lshr", "v_ashr", "v_bfi", "v_perm")) else
             "scalar fma / mul / add" if op.startswith(("v_fma", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac", "v_mac")) else
             "integer address arithmetic" if op.startswith(("v_add_u32", "v_add_co", "v_addc", "v_mad_u", "v_mul_lo", "v_mul_hi", "v_sub_u32", "v_mad_i", "v_add3", "v_lshl_add", "v_add_lshl", "v_sub_co", "v_subrev")) else
             "compares" if op.startswith("v_cmp") else "transcendentals" if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")) else
             "cross-lane (dpp / readlane / permlane)" if ("dpp" in op or op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane"))) else "other")
        groups[g] += n
    return {"instructions": len(ins), "by_unit": dict(c), "valu_by_kind": dict(groups.most_common()), "valu_top_opcodes": dict(ops.most_common(12))}

d, out = sys.argv[1], sys.argv[2]
res = {"_how": __doc__}
KEYS = {"r4": {1024: r"dmel_fwd_kernelILi1024ELi0ELi1E", 2048: r"dmel_fwd_kernelILi2048ELi0ELi1E", 4096: r"dmel_fwd_kernelILi4096ELi0ELi1E"},
        "r5": {1024: r"dmel_fwd_kernelILi1024ELi5ELi1E", 2048: r"dmel_fwd_kernelILi2048ELi5ELi1E", 4096: r"dmel_fwd_kernelILi4096ELi0ELi1E"}}
for n in (1024, 2048, 4096):
    res[str(n)] = {"round4 (kTrain)": mix(f"{d}/r4_{n}.s", KEYS["r4"][n]), "round5 (kTrainW at 1024 / 2048, kTrain + DIT at 4096)": mix(f"{d}/r5_{n}.s", KEYS["r5"][n])}
json.dump(res, open(out, "w"), indent=1)
for n in (1024, 2048, 4096):
    a, b = res[str(n)].values()
    print(n, a["by_unit"], "->", b["by_unit"])
