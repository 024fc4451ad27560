#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of one profiling session (tools/profile_session.sh -> gpurun_out/prof_<tag>/...) into the small files
committed under profiles/.   usage: python tools/summarize_profile.py gpurun_out/prof_r04 r04
Every part of the session is optional: what is missing is skipped (and said so)."""
import csv
import glob
import hashlib
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r04")
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
out = os.path.join(ROOT, "profiles")
FWD_C2 = "dmel_fwd_kernel<1024, 5"          # the training-mode forward at n_fft 1024 (kTrainW since round 5; "<1024, 0" before)


def hits(pattern):
    """the NEWEST match only: a session directory that was merged from two runs holds one rocprofv3 output per run"""
    h = glob.glob(os.path.join(src, pattern), recursive=True)
    return [max(h, key=os.path.getmtime)] if h else []


def short(name):
    return name.split("(")[0].replace("void ", "")


def last_json(path):
    if not os.path.exists(path):
        return None
    lines = [ln for ln in open(path) if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def dump(name, obj):
    json.dump(obj, open(os.path.join(out, name), "w"), indent=1)
    print("wrote", name)


def trace_rows(sub):
    h = hits(f"{sub}/**/*_kernel_trace.csv")
    return list(csv.DictReader(open(h[0]))) if len(h) == 1 else None


def kernel_summary(rows, only_dmel=True, min_calls=1, by_grid=False):
    lst = []
    names = sorted({r["Kernel_Name"] for r in rows if (not only_dmel or "dmel" in r["Kernel_Name"])})
    for k in names:
        rs = [r for r in rows if r["Kernel_Name"] == k]
        grids = sorted({r.get("Grid_Size", r.get("Grid_Size_X")) for r in rs}, key=lambda v: int(v)) if by_grid else [None]
        for g in grids:
            rg = rs if g is None else [r for r in rs if r.get("Grid_Size", r.get("Grid_Size_X")) == g]
            if len(rg) < min_calls:
                continue
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rg]
            r0 = rg[0]
            lst.append(dict(kernel=short(k)[:110], calls=len(d), avg_ns=round(statistics.mean(d), 1), median_ns=statistics.median(d), min_ns=min(d),
                            max_ns=max(d), vgpr=r0.get("VGPR_Count"), accum_vgpr=r0.get("Accum_VGPR_Count"), lds=r0.get("LDS_Block_Size"),
                            scratch=r0.get("Scratch_Size"), wg=r0.get("Workgroup_Size", r0.get("Workgroup_Size_X")),
                            grid=r0.get("Grid_Size", r0.get("Grid_Size_X"))))
    return lst


def counters(sub, kernel_sub):
    """{counter: [values per dispatch]} of the kernels whose name contains kernel_sub"""
    h = hits(f"{sub}/**/*_counter_collection.csv")
    if len(h) != 1:
        return {}
    acc = {}
    for r in csv.DictReader(open(h[0])):
        if kernel_sub in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return acc


def sq_digest(med):
    waves = med.get("SQ_WAVES", 0.0) or 1.0
    d = {"medians_per_launch": med,
         "per_wave": {k.replace("SQ_INSTS_", "insts_"): round(v / waves, 1) for k, v in med.items() if k.startswith("SQ_INSTS_")}}
    if "SQ_WAVE_CYCLES" in med:
        d["shares_of_wave_cycles"] = {k: round(med[k] / med["SQ_WAVE_CYCLES"], 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in med}
    if "SQ_LDS_IDX_ACTIVE" in med:
        d["lds_bank_conflict_share"] = round(med.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, med["SQ_LDS_IDX_ACTIVE"]), 4)
    return d


def valu_figures(med, avg_kernel_us, clk=None):
    """roofline.valu_busy / valu_insts_per_wave of bench.py.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1024 SIMDs; the kernel's
    length IN CYCLES is SQ_BUSY_CYCLES / 32 (the counter sums the busy cycles of the 32 shader engines: 8 XCDs x 4) from the SAME --pmc pass (`clk`),
    not wall time x a nominal 2.4 GHz: under dense vector issue the chip clocks lower (NOTEBOOK R5.1), so round 5's figure understated how busy
    the pipe is.  GRBM_GUI_ACTIVE / 8 of the same pass is kept beside it: MI355X_MICROARCH.md's clock estimate, which reads high on
    dispatches this short (3.6 GHz at config 2's 18 us, 2.1-2.4 GHz on the 0.1-0.4 ms shapes)."""
    out_ = {}
    if med.get("SQ_WAVES") and "SQ_INSTS_VALU" in med:
        out_["valu_insts_per_wave"] = round(med["SQ_INSTS_VALU"] / med["SQ_WAVES"], 1)
    if clk and clk.get("SQ_BUSY_CYCLES") and clk.get("SQ_ACTIVE_INST_VALU"):
        cyc = clk["SQ_BUSY_CYCLES"] / 32.0
        out_["valu_busy"] = round(clk["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc, 4)
        out_["kernel_cycles"] = round(cyc, 1)
        if avg_kernel_us:
            out_["effective_clock_ghz"] = round(cyc / (avg_kernel_us * 1e3), 3)
            out_["valu_busy_wall_2p4ghz"] = round(clk["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (avg_kernel_us * 2400.0), 4)
        if clk.get("GRBM_GUI_ACTIVE"):
            out_["grbm_gui_active_per_xcd"] = round(clk["GRBM_GUI_ACTIVE"] / 8.0, 1)
        out_["valu_busy_note"] = ("SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 shader engines, both from ONE --pmc pass "
                                  "(tools/profile_session.sh: clk); effective_clock_ghz = those cycles over the kernel-trace average of the same session; "
                                  "valu_busy_wall_2p4ghz = round 5's definition (wall x 2.4 GHz)")
    elif "SQ_ACTIVE_INST_VALU" in med and avg_kernel_us:
        out_["valu_busy"] = round(med["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (avg_kernel_us * 2400.0), 4)
        out_["valu_busy_note"] = "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles at 2.4 GHz, kernel-trace average of the same session) -- no clk pass in this session"
    return out_


sha = hashlib.sha256(open(os.path.join(ROOT, "differentiable-mel-spectrogram_amd", "csrc", "dmel_fwd.hip"), "rb").read()).hexdigest()[:16]
hbm = {"_how": "rocprofv3 --pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE (tools/profile_session.sh); medians over the dispatches.  gfx950 correction "
               "(MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of streamed bytes -> doubled, calibrated in the config-2 run on dmel_dot_kernel, which "
               "reads exactly 2 x 4 194 304 B with 16-byte loads.  WRITE_SIZE is taken as it is.  Working sets below the 256 MiB Infinity Cache: fabric-side "
               "bytes, not necessarily DRAM bytes.  bench.py quotes a number only while csrc/dmel_fwd.hip still hashes to kernel_source_sha16.",
       "kernel_source_sha16": sha}

# ---- config 2: bench line, kernel trace, timeline, counters ---------------------------------------------------------------
bench = last_json(os.path.join(src, "bench_c2.json"))
rows = trace_rows("kt")
if rows:
    st = hits("kt/**/*_kernel_stats.csv")
    if len(st) == 1:
        shutil.copy(st[0], os.path.join(out, f"{tag}_kernel_stats_c2.csv"))
    dump(f"{tag}_kernel_trace_summary_c2.json", kernel_summary(rows, only_dmel=False, min_calls=20))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))

    def tight(i):      # forward, dot and the optimizer's kernels back to back: a graph replay, not the eagerly issued trial
        if i + 3 >= len(rows) or FWD_C2 not in rows[i]["Kernel_Name"] or "dmel_dot" not in rows[i + 1]["Kernel_Name"]:
            return False
        if "multi_tensor_apply" not in rows[i + 2]["Kernel_Name"]:
            return False
        return int(rows[i + 2]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) < 12000
    idx = [i for i in range(len(rows)) if tight(i)]
    if idx:
        mid = idx[len(idx) * 3 // 4]
        seq, prev = [], None
        for r in rows[mid:mid + 12]:
            s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            seq.append(dict(kernel=short(r["Kernel_Name"])[:80], dur_us=round((e0 - s0) / 1e3, 2), gap_before_us=None if prev is None else round((s0 - prev) / 1e3, 2)))
            prev = e0
        dump(f"{tag}_step_timeline_c2.json", seq)

fetch, write = counters("fetch", FWD_C2), counters("write", FWD_C2)
dfetch = counters("fetch", "dmel_dot_kernel")
factor = 2.0
if fetch and write and dfetch:
    dot_known = 2 * 4 * 256 * 128 * 32
    f_med, w_med, d_med = statistics.median(fetch["FETCH_SIZE"]), statistics.median(write["WRITE_SIZE"]), statistics.median(dfetch["FETCH_SIZE"])
    factor = dot_known / (d_med * 1024)
    traffic = int(round(f_med * 1024 * 2 + w_med * 1024))
    hbm["c2"] = {"dmel_fwd_kernel_FETCH_SIZE_KiB_raw": f_med, "dmel_fwd_kernel_WRITE_SIZE_KiB": w_med, "dmel_fwd_kernel_bytes_per_launch": traffic,
                 "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"] if bench else None,
                 "ratio_to_algorithmic": round(traffic / bench["roofline"]["algorithmic_bytes_per_launch"], 4) if bench else None,
                 "dispatches": len(fetch["FETCH_SIZE"]), "dmel_dot_kernel_FETCH_SIZE_KiB_raw": d_med, "dmel_dot_kernel_bytes_known": dot_known,
                 "fetch_calibration_factor": round(factor, 3)}
    if bench:
        bench["roofline"]["traffic"] = traffic
        bench["roofline"]["traffic_source"] = f"profiles/hbm_traffic.json (rocprofv3 --pmc, dmel_fwd.hip sha16 {sha})"
sq = {}
for sub in ("sq1", "sq2"):
    for k, v in counters(sub, FWD_C2).items():
        sq[k] = statistics.median(v)
if sq:
    avg_c2 = None
    if rows:
        ks = [k for k in kernel_summary(rows, only_dmel=False, min_calls=20) if FWD_C2 in k["kernel"]]
        avg_c2 = ks[0]["avg_ns"] / 1e3 if ks else None
    clk_c2 = {k: statistics.median(v) for k, v in counters("clk", FWD_C2).items()}
    hbm.setdefault("c2", {}).update(valu_figures(sq, avg_c2, clk_c2))
    d = sq_digest(sq)
    d["_how"] = ("rocprofv3 --pmc, two separate passes (tools/profile_session.sh: sq1, sq2) around `bench.py --steps 30 --warmup 5 --mode eager`; medians over the "
                 "dispatches of dmel_fwd_kernel<1024, train>; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles")
    dump(f"{tag}_pmc_sq_c2.json", d)
# (the bench line is written at the end: the counters of the other shapes are filled into it too)

# ---- the other shapes ------------------------------------------------------------------------------------------------
ALG = {"c3": 4 * (32 * 160000 + 2 * 32 * 128 * 313), "c5": 4 * (32 * 220500 + 2 * 32 * 128 * 501), "c4": 4 * (2048 * 16000 + 2 * 2048 * 128 * 32),
       "esc_n4096": 4 * (32 * 40000 + 2 * 32 * 64 * 501), "esc_n8192": 4 * (32 * 40000 + 2 * 32 * 64 * 501)}
shapes = {}
ktime = {}
kt_path = os.path.join(src, "ktime.txt")
if os.path.exists(kt_path):
    for ln in open(kt_path):
        p = ln.split()
        if len(p) >= 4 and p[0].startswith("libdmel"):
            ktime[f"{p[1]}_{p[2]}"] = float(p[3])
for cfg in ("c3", "c5", "c4", "esc_n4096", "esc_n8192"):
    rws = trace_rows(f"kt_{cfg}")
    if not rws:
        continue
    ent = {"kernels": kernel_summary(rws), "forward_train_us_unprofiled": ktime.get(f"{cfg}_train")}
    f_, w_ = counters(f"fetch_{cfg}", "dmel_fwd_kernel"), counters(f"write_{cfg}", "dmel_fwd_kernel")
    if f_ and w_:
        fm, wm = statistics.median(f_["FETCH_SIZE"]), statistics.median(w_["WRITE_SIZE"])
        tr = int(round(fm * 1024 * 2 + wm * 1024))
        ent["traffic"] = {"FETCH_SIZE_KiB_raw": fm, "WRITE_SIZE_KiB": wm, "dmel_fwd_kernel_bytes_per_launch": tr, "algorithmic_bytes_per_launch": ALG[cfg],
                          "ratio_to_algorithmic": round(tr / ALG[cfg], 4), "dispatches": len(f_["FETCH_SIZE"])}
        pf, pw = counters(f"fetch_{cfg}", "dmel_prep_kernel"), counters(f"write_{cfg}", "dmel_prep_kernel")
        if pf and pw:
            ent["traffic"]["dmel_prep_kernel_bytes_per_launch"] = int(round(statistics.median(pf["FETCH_SIZE"]) * 2048 + statistics.median(pw["WRITE_SIZE"]) * 1024))
        hbm[cfg] = ent["traffic"]
    s_ = {k: statistics.median(v) for k, v in counters(f"sq_{cfg}", "dmel_fwd_kernel").items()}
    s2_ = {k: statistics.median(v) for k, v in counters(f"sq2_{cfg}", "dmel_fwd_kernel").items()}
    s_.update(s2_)
    if s_:
        ent["sq"] = sq_digest(s_)
        fk = [k for k in ent["kernels"] if "dmel_fwd_kernel" in k["kernel"]]
        clk_ = {k: statistics.median(v) for k, v in counters(f"clk_{cfg}", "dmel_fwd_kernel").items()}
        hbm.setdefault(cfg, {}).update(valu_figures(s_, fk[0]["avg_ns"] / 1e3 if fk else None, clk_))
        if "traffic" in ent:
            ent["traffic"] = hbm[cfg]
    shapes[cfg] = ent
if shapes:
    shapes["_how"] = ("tools/profile_session.sh `shapes`: per shape a kernel trace of a train of forward launches (tools/ktime.py <cfg> train) and three counter passes "
                      "(FETCH_SIZE, WRITE_SIZE, SQ) of the same command; c4 = BASELINE config 4's global batch (2048 x 16000) on one GPU; esc_n4096 / esc_n8192 = the "
                      "reference's ESC-50 shape (32 x 40000 @ 8 kHz, hop 80, 64 mels) at lambd 400 / 700")
    dump(f"{tag}_shapes.json", shapes)
if len(hbm) > 2:
    dump("hbm_traffic.json", hbm)
if bench:
    # what bench.py quotes from hbm_traffic.json when it runs AFTER this session (the line of this session ran before the file existed for this source)
    def fill(rl, q):
        for k_ in ("valu_busy", "valu_insts_per_wave", "kernel_cycles", "effective_clock_ghz", "valu_busy_note"):
            if q.get(k_) is not None:
                rl[k_] = q[k_]                                   # (this session's counters, whatever an older hbm_traffic.json said when the line ran)
        if q.get("dmel_fwd_kernel_bytes_per_launch") is not None:
            rl["traffic"] = q["dmel_fwd_kernel_bytes_per_launch"]
    fill(bench["roofline"], hbm.get("c2", {}))
    oc = bench.get("other_configs", {})
    for name, key in (("c3", "c3"), ("c5", "c5"), ("c4_on_one_gpu", "c4")):
        if isinstance(oc.get(name), dict) and "roofline" in oc[name]:
            fill(oc[name]["roofline"], hbm.get(key, {}))
    for name, key in (("esc50_x0.3", "esc_n4096"), ("esc50_lambd700", "esc_n8192")):
        e = oc.get("reference_experiment_shapes", {}).get(name)
        if isinstance(e, dict) and "roofline" in e:
            fill(e["roofline"], hbm.get(key, {}))
    dump(f"{tag}_bench_c2.json", bench)
for nm, fn in (("reference_shapes", "reference_shapes.json"), ("batch_sweep", "batch_sweep.json")):
    d = last_json(os.path.join(src, fn))
    if d:
        dump(f"{tag}_{nm}.json", {"_how": f"tools/{'time_reference_shapes' if nm == 'reference_shapes' else 'batch_sweep'}.py on the final build of the round", "result": d})

# ---- reducers ---------------------------------------------------------------------------------------------------------
red = {}
def red_entry(d):
    return {"value_frames_per_s": d["value"], "ms_per_step": d["ms_per_step"], "n_ranks": d["n_gpus"], "reducer": d["config"].get("reducer"),
            "rccl_ranks_seen": d["config"].get("rccl_ranks_seen"),
            "issued": d["module_step"]["issued"], "regions": d["module_step"].get("timed_regions")}
for nm in ("bench_1rank_plain", "bench_1rank_rccl", "bench_1rank_mailbox", "bench_4ranks_one_gpu_mailbox", "bench_8ranks_one_gpu_mailbox"):
    d = last_json(os.path.join(src, nm + ".json"))
    if d:
        red[nm] = red_entry(d)
two = [last_json(os.path.join(src, f"bench_2ranks_one_gpu_mailbox_{i}.json")) for i in range(1, 6)]
two = [d for d in two if d]
if two:
    vals = [d["value"] for d in two]
    red["bench_2ranks_one_gpu_mailbox"] = {"runs": len(two), "value_frames_per_s": vals, "ms_per_step": [d["ms_per_step"] for d in two],
                                           "median_frames_per_s": statistics.median(vals), "min_frames_per_s": min(vals), "max_frames_per_s": max(vals),
                                           "spread_max_over_min": round(max(vals) / min(vals), 3), "issued": [d["module_step"]["issued"] for d in two]}
if red:
    red["_how"] = ("bench.py --steps 200 --warmup 20 --no-other-configs: plain; DMEL_BENCH_FORCE_DIST=1 (one rank, the reducer still runs: one-rank RCCL communicator / "
                   "mailbox addressed to itself); DMEL_BENCH_SHARE_GPU=1 --gpus 2 / 4 --reducer mailbox (processes on ONE GPU exchanging through IPC-mapped inboxes: the "
                   "code path of N GPUs, not their speed); the two-rank run repeated five times in fresh processes; ms_per_step is the median of 11 regions")
    dump(f"{tag}_reducers.json", red)

# ---- optional gradients, trainable filterbank ---------------------------------------------------------------------------
rws = trace_rows("kt_f2")
if rws:
    dump(f"{tag}_kernel_trace_summary_f2_backward_extras.json", kernel_summary(rws, by_grid=True))
    d = last_json(os.path.join(src, "kt_f2.log"))
    if d:
        dump(f"{tag}_f2_backward_extras_timings.json", d)
tf = {}
d = last_json(os.path.join(src, "learnable_fb.json"))
if d:
    tf["step_us_graph_k1"] = d
d = last_json(os.path.join(src, "fbgrad.json"))
if d:
    tf["dmel_backward_fb_us"] = d
rws = trace_rows("kt_lfb")
if rws:
    tf["kernels_profiled"] = kernel_summary(rws, only_dmel=False, min_calls=50, by_grid=True)
for kern, key in (("dmel_fwd_kernel<1024, 4", "dmel_fwd_kernel<1024,kTrainH>"), ("dmel_fwd_kernel<1024, 0", "dmel_fwd_kernel<1024,kTrain>"), ("dmel_fwd_kernel<1024, 5", "dmel_fwd_kernel<1024,kTrainW>"),
                  ("dmel_fbgrad_lds_kernel<true, false, true>", "dmel_fbgrad_lds_kernel<bf16x3>"), ("dmel_fbgrad_lds_kernel<true, false, false>", "dmel_fbgrad_lds_kernel<fp32>")):
    c = {k: statistics.median(v) for k, v in counters("sq_lfb", kern).items()}
    if c:
        tf.setdefault("counters", {})[key] = c
if tf:
    tf["_how"] = ("tools/time_learnable_fb.py (lambd and the 513 x 128 filterbank trained together at config 2, HIP-graph replay of the nn.Module step, ONE step per replay: "
                  "includes the gap between two graph launches), tools/time_fbgrad.py (dmel_backward_fb alone, fp32 vs bf16x3), kernel trace and SQ counters of the former")
    dump(f"{tag}_trainable_filterbank.json", tf)
st = os.path.join(src, "stamps_c2.txt")
if os.path.exists(st):
    shutil.copy(st, os.path.join(out, f"{tag}_stamps_c2.txt"))
    print("wrote", f"{tag}_stamps_c2.txt")
if bench:
    print({k: bench[k] for k in ("value", "ms_per_step")}, bench["roofline"]["avg_launch_us"], bench["roofline"]["frac"])
