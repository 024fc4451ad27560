#!/usr/bin/env python3
"""bench.py -- frames/s (forward + backward + optimizer update of lambd) of the DMEL layer on N MI355X: BASELINE.json's metric.

A step = one pass of the hot path over one batch of synthetic waveforms already resident in HBM, driven through the
drop-in boundary, the nn.Module:
    opt.zero_grad();  y = MelSpectrogramLayer(x);  y.backward(G_Y);  [N > 1: all-reduce lambd.grad];  opt.step()
  forward : DC removal -> Gaussian-window STFT -> |.|^2 -> mel contraction -> log(. + 1e-10): ONE fused kernel at this
            clip length (torch.ops.dmel.mel_spectrogram -> dmel_forward_dev), carrying d out / d lambd
  backward: lambd.grad = <G_Y, tangent>: one deterministic fp64 dot kernel (dmel_backward_scratch)
  update  : torch.optim.Adam on lambd (the reference's optimizer, main.py:52; lr small enough that n_fft stays the
            metric's 1024 over the run: the reference's lr_tf = 1.0 would leave the band within ~40 steps)
lambd never leaves the device (the reference reads it to the host per sample, time_frequency.py:39; here the kernels read
it and check their n_fft themselves, include/dmel.h), so nothing in the step waits for the host.  `value` is timed over
that step either queued eagerly or replayed from a HIP graph captured once from the same Python code (whichever is
faster on this host; both are printed under "module_step"); the bare two-launch C-ABI step of round 1 (no autograd, no
optimizer) is kept as the side key "c_abi_kernels".

Workload at every N: BASELINE config 2 per GPU (256 x 16000 @16 kHz, n_fft 1024 (lambd 128), hop 512, 128 mels; config 4
is exactly 8 of these, weak scaling).  N > 1: one process per GPU; run plainly (`python bench.py --gpus N`) the script
spawns the N ranks itself, under torchrun it joins the group it finds in the environment.

Prints ONE JSON line (rank 0).  See DESIGN.md for the roofline accounting.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (B per GPU, L, sample_rate, lambd, hop, n_mels)
    "c1": (4, 16000, 16000, 64.0, 256, 64),
    "c2": (256, 16000, 16000, 128.0, 512, 128),
    "c3": (32, 160000, 16000, 256.0, 512, 128),
    "c5": (32, 220500, 44100, 256.0, 441, 128),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 (matrix), v_mfma_f32_16x16x4_f32
FP32_VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 (vector)
ADAM_LR = 1e-3


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="auto", choices=["auto", "eager", "graph"],
                    help="how the timed steps are issued: eagerly from Python, replayed from a HIP graph captured from the same code, "
                         "or (auto) whichever a short trial finds faster")
    ap.add_argument("--bf16-activations", action="store_true", help="store the log-mel output and read its gradient as bf16 "
                    "(DMEL_FLAG_OUT_BF16); the arithmetic, the tangent and d lambd stay fp32.  Default: fp32, the reference's output type")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config 3 / config 5 side measurements")
    ap.add_argument("--report", default="first", choices=["first", "median"],
                    help="which timed region `value` quotes: the first (W warm-up steps, then K timed: the contract's; default) or the median of all")
    ap.add_argument("--reducer", default="rccl", choices=["rccl", "mailbox"],
                    help="N > 1: how lambd.grad is summed over the ranks.  rccl: ncclAllReduce issued in the step's stream between the dot "
                         "kernel and the update (the default).  mailbox: peer-to-peer granules written by the dot kernel's last workgroup "
                         "(dmel_mailbox_*, include/dmel.h): no launch of its own")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the launcher and the reporting path (gloo, no kernels): "
                    "what tests/test_bench_launcher_cpu.py runs")
    return ap.parse_args(argv)


# ---- launcher: `python bench.py --gpus N` spawns its own ranks -------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args) -> int:
    """Start one fresh process per GPU (nothing in THIS process has touched the GPU: a process that has must never be
    replaced or forked), wait for them, pass rank 0's JSON line through.  Returns the exit code."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DMEL_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    # poll every child: when one exits non-zero the rest are stopped (a rank that died inside RCCL init or a collective would
    # otherwise leave rank 0 waiting for ever), and the whole job has a deadline
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("DMEL_BENCH_TIMEOUT_S", "1800"))
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            failed = 124
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:                                   # exactly the children started here, by pid
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    out0 = chunks[0] if chunks else b""
    codes = [failed if failed is not None else 0] + [p.returncode or 0 for p in procs]
    line = ""
    for ln in (out0 or b"").decode("utf-8", "replace").splitlines():
        if ln.startswith("{"):
            line = ln
    if line:
        print(line, flush=True)
    rc = max((abs(c) for c in codes), default=0)
    if rc == 0 and not line:
        rc = 1
    return rc


# ---- CPU baseline --------------------------------------------------------------------------------------------------
def cpu_baseline(cfg):
    """The reference's algorithm on this host's cores (rank 0, N=1 only): the C oracle and the batched
    torch restatement, each best of 5 on one full batch of the same workload; the faster one is reported."""
    import torch
    from oracle import dmel_oracle as O
    from oracle import torch_restatement as TR
    from dmel_amd import synth
    B, L, sr, lam, hop, M = cfg
    Bs = min(B, 256)
    x = synth.waveforms(Bs, L, seed=0)
    T = L // hop + 1
    g = synth.cotangent((Bs, 1, M, T), seed=1)
    frames = Bs * T
    cores = os.cpu_count() or 1
    res = {}
    O.forward(x[:2], lam, hop, M, sr, apply_log=True)
    best = 1e30
    for _ in range(5):
        t0 = time.perf_counter()
        _, tan = O.forward(x, lam, hop, M, sr, apply_log=True)
        O.backward(g, tan)
        best = min(best, time.perf_counter() - t0)
    res["c_oracle_openmp"] = (frames / best, O.threads())
    xt, gt = torch.from_numpy(x), torch.from_numpy(g)
    fb = TR.melscale_fbanks(TR.n_fft_of(torch.tensor(lam)) // 2 + 1, 0.0, sr // 2, M, sr)
    TR.step(xt[:2], gt[:2], lam, hop, M, sr, log=True, fb=fb)
    best = 1e30
    for _ in range(5):
        t0 = time.perf_counter()
        TR.step(xt, gt, lam, hop, M, sr, log=True, fb=fb)
        best = min(best, time.perf_counter() - t0)
    res["torch_batched"] = (frames / best, torch.get_num_threads())
    kind = max(res, key=lambda k: res[k][0])
    return {"value": round(res[kind][0], 1), "unit": "frames/s", "cores": int(res[kind][1]), "kind": "port",
            "sample": f"one full batch ({Bs} x {L} samples = {frames} frames) fwd+bwd, best of 5; "
                      f"{kind} (other: " + ", ".join(f"{k}={v[0]:.0f} f/s" for k, v in res.items() if k != kind) + ")",
            "host_cpus": cores}


def dry_run(args, rank, world):
    """No GPU: rehearse group set-up, the barrier / max-over-ranks timing and the JSON line (gloo)."""
    import torch
    import torch.distributed as dist
    from dmel_amd import dist as ddist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, L, sr, lam, hop, M = CONFIGS[args.config]
    T = L // hop + 1
    sar = ddist.ScalarAllReduce() if world > 1 else None
    grad = torch.tensor([float(rank + 1)])

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        pass
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if sar is not None:
            grad.fill_(float(rank + 1))
            sar.reduce(grad, 0)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        assert float(grad) == world * (world + 1) / 2
    # the agreed-decision path: the real GraphedStep against a CPU stand-in of the device whose "latest lambd" picture is seen
    # late by a rank-dependent amount, every step containing a real collective (tools/graph_rehearsal.py).  A rank that re-captured
    # at another call than the others would strand them in that collective.
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import graph_rehearsal as GR
    from dmel_amd import GraphedStep, capi

    def allreduce(v):
        tt = torch.tensor([v], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt)
        return float(tt)

    caps, gsteps, colls, uncovered = GR.rehearse(GraphedStep, capi.n_fft, capi.decide_launch, allreduce, rank, world, calls=120, k=2,
                                                 max_ahead=4, eval_at=(60,))
    agreed = True
    if world > 1:
        mine = torch.tensor(caps + [-1] * (64 - len(caps)))
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        agreed = all(torch.equal(e, every[0]) for e in every)
    assert agreed and not uncovered and gsteps == 240, (caps, uncovered, gsteps)
    if rank == 0:
        print(json.dumps({"metric": "spectrogram frames/sec (fwd+bwd)", "value": round(world * B * T * args.steps / max(elapsed, 1e-9), 1),
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / max(1, args.steps), 5), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "none (dry run: launcher and collective only, no kernels)",
                          "config": {"workload": "dry run", "parallelism": f"batch-sharded x{world}",
                                     "collective": "none" if sar is None else ("native dmel_comm" if sar.native else "torch.distributed: " + sar.why),
                                     "graphed_step_rehearsal": {"calls": 120, "steps_per_replay": 2, "capture_calls": caps, "collectives": colls,
                                                                "same_on_every_rank": agreed}}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    in_group = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_group:
        raise SystemExit(spawn_ranks(args))          # before anything here has touched a GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if in_group and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world)

    import numpy as np
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU (use --dry-run for the CPU rehearsal of the launcher)"
    # DMEL_BENCH_SHARE_GPU=1: every rank on device 0 (a rehearsal of the N-rank code path on a one-GPU box: RCCL refuses two ranks
    # on one device, so the group is gloo and the reducer must be the mailbox, whose IPC-mapped inboxes work within one GPU too)
    share = os.environ.get("DMEL_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
        assert args.reducer == "mailbox" or world == 1, "DMEL_BENCH_SHARE_GPU needs --reducer mailbox"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("DMEL_BENCH_FORCE_DIST") == "1":     # the env knob exercises the reducer path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cdev = "cpu" if share else dev                                     # where the tensors of the bench's own collectives live

    from dmel_amd import MelSpectrogramLayer, capi, synth
    from dmel_amd import dist as ddist

    cfg = CONFIGS[args.config]
    B, L, sr, lam, hop, M = cfg
    T = L // hop + 1
    frames_per_rank = B * T
    # every rank owns a different shard of the global batch (seeded by rank); inputs are resident before timing
    x = torch.from_numpy(synth.waveforms(B, L, seed=100 * rank)).to(dev)
    g32 = torch.from_numpy(synth.cotangent((B, 1, M, T), seed=1 + 100 * rank)).to(dev)
    act = torch.bfloat16 if args.bf16_activations else torch.float32
    g = g32.to(act)

    # ---- the step through the boundary ----------------------------------------------------------------------------------
    layer = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=str(dev),
                                optimized=True, log=True, out_dtype=act).to(dev)
    try:
        opt = torch.optim.Adam([layer.lambd], lr=ADAM_LR, fused=True, capturable=True)
        opt_kind = "Adam(fused=True, capturable=True)"
    except Exception:                                                   # noqa: BLE001
        opt = torch.optim.Adam([layer.lambd], lr=ADAM_LR, capturable=True)
        opt_kind = "Adam(capturable=True)"
    sar, mar, reducer = None, None, "none (one rank)"
    if dist is not None:
        if args.reducer == "mailbox":
            mar = ddist.MailboxAllReduce()                              # raises on every rank if any rank cannot set it up
            mar.attach(layer, dev)
            reducer = "peer-to-peer mailbox folded into dmel_dot_kernel (dmel_mailbox_*): no launch of its own"
        else:
            sar = ddist.ScalarAllReduce()
            reducer = "RCCL ncclAllReduce in the step's stream (" + ("native dmel_comm" if sar.native else "torch.distributed: " + sar.why) + ")"
    # proof that `world` ranks joined the reducer the step uses: a vector of ones summed through THAT path (the native RCCL
    # communicator in the step's stream, or the mailbox) -- a rank that fell back, or a group of fewer ranks, shows here
    ranks_seen = 1
    if dist is not None:
        probe = torch.ones(1, device=dev)
        if sar is not None:
            sar.reduce(probe, torch.cuda.current_stream(dev).cuda_stream)
        else:
            mar.reduce(probe, torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        ranks_seen = int(round(float(probe)))
        assert ranks_seen == world, f"the reducer summed {ranks_seen} ranks, the job has {world}"
    lam_param = layer.lambd

    def module_step():
        opt.zero_grad(set_to_none=True)
        layer(x).backward(g)
        if sar is not None:                                             # data-parallel: the update needs the SUM over ranks
            sar.reduce(lam_param.grad, torch.cuda.current_stream(dev).cuda_stream)
        opt.step()

    for _ in range(3):                                                  # tables, allocator, optimizer state, lambd tracking
        module_step()
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def time_loop(fn, warm, steps):
        import gc
        for _ in range(warm):
            fn()
        # (the driver's region is well under a millisecond: a pass of Python's cycle collector inside it would be a tenth of the figure)
        gc_was = gc.isenabled()
        gc.disable()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        el = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # ways of issuing the same step: eagerly from Python, or replayed from a HIP graph captured from that very code
    # (dmel_amd.GraphedStep: watches the lambd the kernels report and re-captures when the n_fft or its guards change);
    # "graph xk" unrolls k steps into one graph, which pays the ~8-14 us between two graph launches once per k steps
    from dmel_amd import GraphedStep
    modes, graph_why = {"eager": (module_step, 1)}, ""
    MAX_AHEAD = 8
    if args.mode in ("auto", "graph"):
        # (a graph launch costs ~8-14 us of idle time on the device between two replays -- profiles/r03_step_timeline_c2.json --
        # whatever the graph holds: the more steps one replay carries, the less of it each step pays)
        ks = [1] + [k for k in (4, 10, 20) if args.steps % k == 0]
        for k in ks:
            ok = True
            try:
                gs = GraphedStep(module_step, [layer], max_ahead=MAX_AHEAD, steps_per_replay=k)
                for _ in range(MAX_AHEAD + 4):                          # past the first delayed look: the graph without guards is in place
                    gs()
                torch.cuda.synchronize()
            except Exception as e:                                      # noqa: BLE001 -- report and fall back to eager issue
                graph_why = f"{type(e).__name__}: {e}"[:300]
                torch.cuda.synchronize()
                ok = False
            if dist is not None:
                # every rank must time the same set of modes: a capture that failed on one rank only would leave the others
                # replaying a graph with an all-reduce in it that this rank never joins
                flag = torch.tensor([1 if ok else 0], device=cdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if ok and int(flag.item()) == 0:
                    graph_why = graph_why or "graph capture failed on another rank"
                ok = int(flag.item()) == 1
            if not ok:
                break
            modes["graph" if k == 1 else f"graph x{k}"] = (gs, k)

    def run_mode(name, warm, steps):
        fn, k = modes[name]
        return time_loop(fn, (warm + k - 1) // k, steps // k)

    chosen = args.mode
    trial = {}
    if chosen == "graph":
        chosen = "graph" if "graph" in modes else "eager"
    if chosen in ("auto", "graph") and len(modes) > 1:
        # the trial times what the timed region will: exactly --steps steps behind a synchronisation (a replay of many steps pays
        # less launch gap per step, but the first launch after a synchronisation is exposed, which counts when --steps is small)
        ntrial = args.steps
        names = [m for m in modes if (args.mode == "auto" or m != "eager")]
        for name in names:
            trial[name] = min(run_mode(name, 4, ntrial), run_mode(name, 0, ntrial)) / ntrial
        chosen = min(trial, key=trial.get)
        if dist is not None:                                            # every rank must take the same path
            order = list(modes)
            pick = torch.tensor([order.index(chosen)], device=cdev)
            dist.all_reduce(pick, op=dist.ReduceOp.MIN)
            chosen = order[int(pick.item())]
    elif chosen == "auto":
        chosen = "eager"
    # The timed region: W untimed steps, then EXACTLY K timed steps between barrier + synchronize on both sides, MAX over ranks.
    # With the driver's small K such a region is one or two graph launches (well under a millisecond), so it is REPEATED: the
    # first region is the one the contract describes (W warm-up steps before it) and the one reported; REGIONS - 1 more of exactly
    # K steps follow for the side fields (median, min, max).
    REGIONS = 11 if args.steps * 0.05 < 50.0 else 3                       # (long regions, > ~50 ms of steps: three)
    regions = [run_mode(chosen, args.warmup, args.steps)]
    for _ in range(REGIONS - 1):
        regions.append(run_mode(chosen, 0, args.steps))
    # `value` / `ms_per_step` are the FIRST region again -- W untimed steps, then exactly K timed ones, as the contract
    # describes and as rounds 1-3 reported (round 4 reported the median region: a different statistic); --report median restores that.
    # Median, minimum and maximum of all regions are side fields either way.
    elapsed = regions[0] if args.report == "first" else sorted(regions)[len(regions) // 2]
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * frames_per_rank * args.steps / elapsed
    status = layer.lambd_status()
    lam_end = float(layer.lambd.detach())
    assert status["error"] == 0 and capi.n_fft(lam_end) == capi.n_fft(lam), (status, lam_end)
    module_step_info = {"issued": chosen, "optimizer": f"{opt_kind}, lr {ADAM_LR}", "lambd_end": round(lam_end, 4),
                        "timed_regions": {"n": len(regions), "steps_each": args.steps,
                                          "reported": "first region (warm-up + K steps: the contract's)" if args.report == "first" else "median region",
                                          "ms_per_step_min": round(1e3 * min(regions) / args.steps, 5),
                                          "ms_per_step_median": round(1e3 * sorted(regions)[len(regions) // 2] / args.steps, 5),
                                          "ms_per_step_max": round(1e3 * max(regions) / args.steps, 5),
                                          "first_region_ms_per_step": round(1e3 * regions[0] / args.steps, 5)},
                        "trial_ms_per_step": {k: round(1e3 * v, 4) for k, v in trial.items()},
                        "guards_last_call": status["guards"], "graph_unavailable": graph_why or None,
                        "graph_captures": {m: modes[m][0].captures for m in modes if m != "eager"}}
    # the driver's default --steps gives a timed region of well under a millisecond: the same mode over at least 50 graph replays
    # (>= 200 steps) is reported next to it (`steps` / `value` stay what was asked for)
    k_chosen = modes[chosen][1]
    long_steps = max(200, 50 * k_chosen)
    if args.steps < long_steps:
        module_step_info["long_run_ms_per_step"] = round(1e3 * run_mode(chosen, 8, long_steps) / long_steps, 5)
        module_step_info["long_run_steps"] = long_steps
    if dist is None:
        for m in modes:                                                 # the other ways of issuing the same step, for the record
            if m != chosen:
                module_step_info[m.replace(" ", "_") + "_ms_per_step"] = round(1e3 * run_mode(m, 8, 80) / 80, 4)
        module_step_info[chosen.replace(" ", "_") + "_ms_per_step"] = round(ms_per_step, 4)

    # ---- what a training loop would see (train.py:25-49: a NEW batch every step), side figures, never `value` ---------------
    # loader_fed: INTEGRATION.md's loop -- one graph replay per step (k = 1), each preceded by a device-side copy of the next batch
    # into the graph's static input (`x_static.copy_(batch)`); the batches come from a pool of POOL resident ones (> 256 MiB in
    # total, so neither L2 nor the 256 MiB Infinity Cache holds the next one: the copy reads HBM).
    # cold_inputs: the chosen issue mode with every step of a replay reading a DIFFERENT resident batch of that pool (no copy):
    # the forward's first touch of x comes from HBM, not from the cache the previous step left.
    if dist is None and not args.no_other_configs:
        try:
            POOL = 24 if B * L * 4 * 24 > (256 << 20) else max(24, (300 << 20) // (B * L * 4) + 1)
            pool = [torch.from_numpy(synth.waveforms(B, L, seed=1000 + i)).to(dev) for i in range(POOL)]
            layer4 = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=str(dev),
                                         optimized=True, log=True, out_dtype=act).to(dev)
            opt4 = torch.optim.Adam([layer4.lambd], lr=ADAM_LR, fused=True, capturable=True)
            x_static = pool[0].clone()

            def module_step4():
                opt4.zero_grad(set_to_none=True)
                layer4(x_static).backward(g)
                opt4.step()

            for _ in range(3):
                module_step4()
            torch.cuda.synchronize()
            gs4 = GraphedStep(module_step4, [layer4], max_ahead=MAX_AHEAD, steps_per_replay=1) if chosen != "eager" else module_step4
            it = [0]

            def fed_step():
                x_static.copy_(pool[it[0] % POOL], non_blocking=True)
                it[0] += 1
                gs4()

            for _ in range(MAX_AHEAD + 4):
                fed_step()
            torch.cuda.synchronize()
            nfed = 240
            el4 = sorted(time_loop(fed_step, 8 if r == 0 else 0, nfed) for r in range(5))[2]
            # the copy alone (same pool rotation), to separate it from the step
            def copy_only():
                x_static.copy_(pool[it[0] % POOL], non_blocking=True)
                it[0] += 1
            elc = sorted(time_loop(copy_only, 8 if r == 0 else 0, nfed) for r in range(3))[1]
            assert layer4.lambd_status()["error"] == 0
            module_step_info["loader_fed"] = {"ms_per_step": round(1e3 * el4 / nfed, 5), "frames_per_s": round(frames_per_rank * nfed / el4, 1),
                                              "copy_alone_ms": round(1e3 * elc / nfed, 5), "steps": nfed, "pool_batches": POOL,
                                              "pool_mib": round(POOL * B * L * 4 / 2**20, 1), "median_of": 5,
                                              "issued": "graph (one step per replay) + x_static.copy_(next batch) per step" if chosen != "eager" else "eager + copy",
                                              "note": "INTEGRATION.md's training loop (train.py:25-49): a new batch every step through a device-side "
                                                      "copy into the graph's static input; side figure, never `value`"}
            del gs4
            # loader_fed_slots (round 5): GraphedStep(inputs=[x], steps_per_replay=10) -- ten static slots per set, feed() copies the next
            # batch on a side stream while the previous replay runs, one replay per ten batches; the same pool rotation
            for by_address in ((False, True) if chosen != "eager" else ()):
                KS = 10
                layer6 = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=str(dev),
                                             optimized=True, log=True, out_dtype=act).to(dev)
                opt6s = torch.optim.Adam([layer6.lambd], lr=ADAM_LR, fused=True, capturable=True)

                def module_step6(xb):
                    opt6s.zero_grad(set_to_none=True)
                    layer6(xb).backward(g)
                    opt6s.step()

                for _ in range(3):
                    module_step6(pool[0])
                torch.cuda.synchronize()
                gs6 = GraphedStep(module_step6, [layer6], max_ahead=MAX_AHEAD, steps_per_replay=KS, inputs=[pool[0]], zero_copy=[True] if by_address else None)
                it6 = [0]

                def fed_slot():
                    gs6.feed(pool[it6[0] % POOL])
                    it6[0] += 1

                for _ in range((MAX_AHEAD + 4) * KS):
                    fed_slot()
                torch.cuda.synchronize()
                nfs = 24 * KS
                el6 = sorted(time_loop(fed_slot, KS if r == 0 else 0, nfs) for r in range(5))[2]
                assert layer6.lambd_status()["error"] == 0
                if by_address:
                    module_step_info["loader_fed_by_address"] = {
                        "ms_per_step": round(1e3 * el6 / nfs, 5), "frames_per_s": round(frames_per_rank * nfs / el6, 1),
                        "steps_per_replay": KS, "steps": nfs, "pool_batches": POOL, "median_of": 5,
                        "issued": "GraphedStep(inputs=[x], zero_copy=[True], steps_per_replay=10).feed(batch): no copy of the batch -- its address goes into a "
                                  "pointer cell the fused forward reads (DMEL_FLAG_X_INDIRECT), ten addresses per replay in one 80-byte copy inside the graph",
                        "note": "for batches that are already device tensors (train.py:33 `inputs.to(device)` makes one per step); side figure, never `value`"}
                else:
                    module_step_info["loader_fed_slots"] = {
                        "ms_per_step": round(1e3 * el6 / nfs, 5), "frames_per_s": round(frames_per_rank * nfs / el6, 1),
                        "steps_per_replay": KS, "steps": nfs, "pool_batches": POOL, "median_of": 5,
                        "issued": "GraphedStep(inputs=[x], steps_per_replay=10).feed(batch): copies on a side stream under the previous replay, one replay per ten batches",
                        "note": "the loop INTEGRATION.md documents since round 5 (train.py:25-49: a new batch every step); side figure, never `value`"}
                gs6.close()
                del gs6, layer6, opt6s
            # cold inputs, no copy: k steps per replay, step j of a replay reads pool[j]
            kc = k_chosen if chosen != "eager" else 1
            kc = max(kc, 20) if chosen != "eager" else 1
            jt = [0]

            def module_step5():
                opt4.zero_grad(set_to_none=True)
                layer4(pool[jt[0] % POOL]).backward(g)
                jt[0] += 1
                opt4.step()

            if chosen != "eager":
                jt[0] = 0
                fn5 = GraphedStep(module_step5, [layer4], max_ahead=MAX_AHEAD, steps_per_replay=kc)
                for _ in range(MAX_AHEAD + 4):
                    fn5()
            else:
                fn5 = module_step5
            torch.cuda.synchronize()
            n5 = max(240, 12 * kc)
            el5 = sorted(time_loop(fn5, 8 if r == 0 else 0, n5 // kc) for r in range(5))[2]
            assert layer4.lambd_status()["error"] == 0
            module_step_info["cold_inputs"] = {"ms_per_step": round(1e3 * el5 / (n5 // kc * kc), 5),
                                               "frames_per_s": round(frames_per_rank * (n5 // kc * kc) / el5, 1),
                                               "steps_per_replay": kc, "distinct_batches_per_replay": min(kc, POOL), "median_of": 5,
                                               "note": "every step of a replay reads a different resident batch (pool > 256 MiB): no step finds its input "
                                                       "in L2 / Infinity Cache; side figure, never `value`"}
            del fn5, opt4, layer4, pool
        except Exception as e:                                          # noqa: BLE001
            module_step_info["loader_fed"] = module_step_info.get("loader_fed") or {"error": f"{type(e).__name__}: {e}"[:300]}
            module_step_info.setdefault("cold_inputs", {"error": f"{type(e).__name__}: {e}"[:300]})
        torch.cuda.empty_cache()

    # ---- the N > 1 code path on this one GPU (what DMEL_BENCH_FORCE_DIST=1 runs): a world-1 RCCL group, the ncclAllReduce of
    # lambd.grad captured inside the step.  The driver's scaling run starts at N = 1 on the plain path; this figure shows that the
    # path the other N take costs the same step when there is nobody to talk to (within 3 %: DMEL_BENCH_STRICT=1 asserts it).
    if dist is None and not args.no_other_configs:
        try:
            import torch.distributed as tdist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ["MASTER_PORT"] = str(_free_port())
            # RCCL prints a version banner through C stdio when its first communicator comes up: stdout carries the JSON line only
            sys.stdout.flush()
            _flush_c_stdio()
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            try:
                sar1 = ddist.ScalarAllReduce()
                layer6 = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=str(dev),
                                             optimized=True, log=True, out_dtype=act).to(dev)
                opt6 = torch.optim.Adam([layer6.lambd], lr=ADAM_LR, fused=True, capturable=True)

                def module_step6():
                    opt6.zero_grad(set_to_none=True)
                    layer6(x).backward(g)
                    sar1.reduce(layer6.lambd.grad, torch.cuda.current_stream(dev).cuda_stream)
                    opt6.step()

                for _ in range(3):
                    module_step6()
                torch.cuda.synchronize()
                fn6 = module_step6
                if chosen != "eager":
                    fn6 = GraphedStep(module_step6, [layer6], max_ahead=MAX_AHEAD, steps_per_replay=k_chosen)
                    for _ in range(MAX_AHEAD + 4):
                        fn6()
                    torch.cuda.synchronize()
                n6 = max(200, 50 * k_chosen)
                el6 = sorted(time_loop(fn6, 8 if r == 0 else 0, n6 // k_chosen) for r in range(5))[2]
                ref = sorted(run_mode(chosen, 8 if r == 0 else 0, n6) for r in range(5))[2]         # the plain path, same length, now
                ratio = (el6 / (n6 // k_chosen * k_chosen)) / (ref / n6)
                module_step_info["dist_path_world1"] = {"ms_per_step": round(1e3 * el6 / (n6 // k_chosen * k_chosen), 5),
                                                        "plain_ms_per_step_same_length": round(1e3 * ref / n6, 5), "ratio": round(ratio, 4),
                                                        "within_3pct": bool(abs(ratio - 1.0) <= 0.03), "steps": n6, "median_of": 5,
                                                        "reducer": "RCCL ncclAllReduce in the step's stream (" + ("native dmel_comm" if sar1.native else "torch.distributed: " + sar1.why) + "), world 1",
                                                        "note": "the code path every N > 1 run takes, with nobody to talk to; side figure"}
                del fn6, opt6, layer6
                sar1.close()
            finally:
                tdist.destroy_process_group()
                _flush_c_stdio()
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
            if os.environ.get("DMEL_BENCH_STRICT") == "1":
                assert module_step_info["dist_path_world1"]["within_3pct"], module_step_info["dist_path_world1"]
        except AssertionError:
            raise
        except Exception as e:                                          # noqa: BLE001
            module_step_info["dist_path_world1"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- side figure, never `value`: the same step with the opt-in dmel_amd.LambdAdam (torch.optim.Adam's update of lambd as
    # one launch instead of torch's two), issued the way the headline was
    if dist is None and not args.no_other_configs:
        try:
            from dmel_amd import LambdAdam
            layer2 = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=str(dev),
                                         optimized=True, log=True, out_dtype=act).to(dev)
            opt2 = LambdAdam([layer2.lambd], lr=ADAM_LR)

            def module_step2():
                opt2.zero_grad(set_to_none=True)
                layer2(x).backward(g)
                opt2.step()

            for _ in range(3):
                module_step2()
            torch.cuda.synchronize()
            fn2 = module_step2
            if chosen != "eager":
                fn2 = GraphedStep(module_step2, [layer2], max_ahead=MAX_AHEAD, steps_per_replay=k_chosen)
                for _ in range(MAX_AHEAD + 4):
                    fn2()
                torch.cuda.synchronize()
            n2 = max(200, 50 * k_chosen)
            el2 = min(time_loop(fn2, 8, n2 // k_chosen), time_loop(fn2, 0, n2 // k_chosen))
            st2 = layer2.lambd_status()
            assert st2["error"] == 0
            module_step_info["with_lambd_adam"] = {"ms_per_step": round(1e3 * el2 / n2, 5), "frames_per_s": round(frames_per_rank * n2 / el2, 1),
                                                   "steps": n2, "issued": chosen, "lambd_end": round(float(layer2.lambd.detach()), 4),
                                                   "note": "opt-in dmel_amd.LambdAdam (dmel_adam_step: one launch) in place of torch.optim.Adam; "
                                                           "a side figure, never `value`"}
            del fn2, opt2, layer2
        except Exception as e:                                          # noqa: BLE001
            module_step_info["with_lambd_adam"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        # ... and with that update applied by the workgroup that finishes the backward's dot product (LambdAdam(fused_into_backward=layer),
        # dmel_plan_attach_adam): the step is two launches -- fused forward, dot
        try:
            from dmel_amd import LambdAdam
            layer2f = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=str(dev),
                                          optimized=True, log=True, out_dtype=act).to(dev)
            layer2f(x)
            opt2f = LambdAdam([layer2f.lambd], lr=ADAM_LR, fused_into_backward=layer2f)

            def module_step2f():
                opt2f.zero_grad(set_to_none=True)
                layer2f(x).backward(g)
                opt2f.step()

            for _ in range(3):
                module_step2f()
            torch.cuda.synchronize()
            fn2f = module_step2f
            if chosen != "eager":
                fn2f = GraphedStep(module_step2f, [layer2f], max_ahead=MAX_AHEAD, steps_per_replay=k_chosen)
                for _ in range(MAX_AHEAD + 4):
                    fn2f()
                torch.cuda.synchronize()
            n2f = max(200, 50 * k_chosen)
            el2f = min(time_loop(fn2f, 8, n2f // k_chosen), time_loop(fn2f, 0, n2f // k_chosen))
            assert layer2f.lambd_status()["error"] == 0 and float(opt2f.state[layer2f.lambd]["step"]) > 0
            module_step_info["with_lambd_adam_fused"] = {"ms_per_step": round(1e3 * el2f / n2f, 5), "frames_per_s": round(frames_per_rank * n2f / el2f, 1),
                                                         "steps": n2f, "issued": chosen, "lambd_end": round(float(layer2f.lambd.detach()), 4),
                                                         "note": "opt-in LambdAdam(fused_into_backward=layer): Adam's update of lambd applied in the tail of the "
                                                                 "backward's dot kernel (two launches per step); a side figure, never `value`"}
            del fn2f, opt2f, layer2f
        except Exception as e:                                          # noqa: BLE001
            module_step_info["with_lambd_adam_fused"] = {"error": f"{type(e).__name__}: {e}"[:200]}

    # ---- side figure, never `value`: BASELINE config 2 read literally ("bf16 activations / fp32 grad"): the log-mel output stored
    # as bf16 and its gradient read as bf16 (arithmetic, tangent and d lambd stay fp32), issued the way the headline was
    if dist is None and not args.no_other_configs and not args.bf16_activations:
        try:
            layer3 = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=str(dev),
                                         optimized=True, log=True, out_dtype=torch.bfloat16).to(dev)
            opt3 = torch.optim.Adam([layer3.lambd], lr=ADAM_LR, fused=True, capturable=True)
            g16 = g32.to(torch.bfloat16)

            def module_step3():
                opt3.zero_grad(set_to_none=True)
                layer3(x).backward(g16)
                opt3.step()

            for _ in range(3):
                module_step3()
            torch.cuda.synchronize()
            fn3 = module_step3
            if chosen != "eager":
                fn3 = GraphedStep(module_step3, [layer3], max_ahead=MAX_AHEAD, steps_per_replay=k_chosen)
                for _ in range(MAX_AHEAD + 4):
                    fn3()
                torch.cuda.synchronize()
            n3 = max(200, 50 * k_chosen)
            el3 = min(time_loop(fn3, 8, n3 // k_chosen), time_loop(fn3, 0, n3 // k_chosen))
            assert layer3.lambd_status()["error"] == 0
            module_step_info["with_bf16_activations"] = {"ms_per_step": round(1e3 * el3 / n3, 5), "frames_per_s": round(frames_per_rank * n3 / el3, 1),
                                                         "steps": n3, "issued": chosen,
                                                         "note": "output and its gradient as bf16 (BASELINE config 2's wording), arithmetic / tangent / "
                                                                 "d lambd fp32; a side figure: `value` is measured with fp32 activations, the reference's"}
            del fn3, opt3, layer3
        except Exception as e:                                          # noqa: BLE001
            module_step_info["with_bf16_activations"] = {"error": f"{type(e).__name__}: {e}"[:200]}

    # ---- the bare kernels through the C ABI (round 1's headline): fused forward + dot, lambd by value, no autograd, no update -
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    out = torch.empty((B, 1, M, T), dtype=act, device=dev)
    tan = torch.empty((B, 1, M, T), dtype=torch.float32, device=dev)
    dl = torch.zeros(1, dtype=torch.float32, device=dev)
    count = out.numel()
    cur = torch.cuda.current_stream(dev)

    def step_kernels():
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, cur.cuda_stream,
                     extra_flags=capi.DMEL_FLAG_OUT_BF16 if args.bf16_activations else 0)
        plan.backward(g.data_ptr(), tan.data_ptr(), count, dl.data_ptr(), cur.cuda_stream, grad_bf16=args.bf16_activations)

    step_kernels()
    torch.cuda.synchronize()
    info = plan.info()
    n_k = max(20, min(200, args.steps))
    for _ in range(10):
        step_kernels()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_k):
        step_kernels()
    torch.cuda.synchronize()
    kern_ms = 1e3 * (time.perf_counter() - t0) / n_k

    # ---- dominant kernel: average launch duration from HIP events on the launch stream ------------------------------------
    # An event pair around ONE launch also times the launch packets around it (1.5-4 us, varying from box to box).  The
    # duration is therefore taken from trains of launches between two HIP events on the launch stream: a train of whole steps
    # (forward + dot, so the forward sees the cache state the dot kernel leaves) minus a train of the dot launches alone.
    # This is what rocprofv3 reports as the dispatch duration plus the sub-microsecond gap between dependent dispatches.
    def kernel_times(pl, xx, gg, oo, tt, dd, lam_v, n_elems, batch, bf16):
        pl.set_profiling(True)
        for _ in range(20):
            pl.forward(xx.data_ptr(), batch, lam_v, oo.data_ptr(), tt.data_ptr(), True, 1e-10, cur.cuda_stream,
                       extra_flags=capi.DMEL_FLAG_OUT_BF16 if bf16 else 0)
            pl.backward(gg.data_ptr(), tt.data_ptr(), n_elems, dd.data_ptr(), cur.cuda_stream, grad_bf16=bf16)
        torch.cuda.synchronize()
        prof = pl.get_profile()
        pl.set_profiling(False)
        pair = 1e3 * prof["fwd_ms"] / max(1, prof["fwd_launches"])
        prep = 1e3 * prof["prep_ms"] / max(1, prof["prep_launches"])
        bwd = 1e3 * prof["bwd_ms"] / max(1, prof["bwd_launches"])
        ntrain = 100
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record(cur)
        for _ in range(ntrain):
            pl.forward(xx.data_ptr(), batch, lam_v, oo.data_ptr(), tt.data_ptr(), True, 1e-10, cur.cuda_stream,
                       extra_flags=capi.DMEL_FLAG_OUT_BF16 if bf16 else 0)
            pl.backward(gg.data_ptr(), tt.data_ptr(), n_elems, dd.data_ptr(), cur.cuda_stream, grad_bf16=bf16)
        ev[1].record(cur)
        ev[2].record(cur)
        for _ in range(ntrain):
            pl.backward(gg.data_ptr(), tt.data_ptr(), n_elems, dd.data_ptr(), cur.cuda_stream, grad_bf16=bf16)
        ev[3].record(cur)
        torch.cuda.synchronize()
        step_us = 1e3 * ev[0].elapsed_time(ev[1]) / ntrain
        dot_us = 1e3 * ev[2].elapsed_time(ev[3]) / ntrain
        fwd = step_us - dot_us - (prep if prep > 0 else 0.0)   # long clips: every forward also launched the partial-sum kernel
        return fwd, pair, prep, bwd, step_us

    fwd_us, fwd_pair_us, prep_us, bwd_us, _ = kernel_times(plan, x, g, out, tan, dl, lam, count, B, args.bf16_activations)
    # algorithmic bytes of ONE launch of the fused forward kernel: read x once + write out and tangent once
    alg_bytes = 4 * (B * L + B * M * T) + (2 if args.bf16_activations else 4) * B * M * T
    achieved = alg_bytes / (fwd_us * 1e-6) / 1e9
    # HBM-side bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read from inside the
    # process); the committed measurement is quoted only if it was taken on THIS kernel source, otherwise traffic is null
    quoted, traffic_src = profile_quotes()
    traffic = quoted.get(args.config, {}).get("dmel_fwd_kernel_bytes_per_launch")
    # the contraction stage on the matrix cores: executed fp32 MFMA flops of one launch = non-zero 4x16 filterbank blocks
    # x 16-row tiles x 2048 flop per v_mfma_f32_16x16x4_f32, against the dense fp32 matrix peak
    mfma_flops = mfma_flops_of(info, B, T)
    dense_flops = 2.0 * 2.0 * B * T * (info["n_fft"] // 2 + 1) * M                      # 2 F M per frame, power and tangent rows
    mfma = {"contraction": {0: "16x16x4 fp32 tiles over the non-zero 4x16 blocks", 1: "wave-local 4x4x1 fp32 (kTrainW)", 2: "dense bf16x3 (kTrainH)"}.get(info.get("contraction"), "none"),
            "executed_tflops": round(mfma_flops / (fwd_us * 1e-6) / 1e12, 2), "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
            "frac": round(mfma_flops / (fwd_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "executed_flops_per_launch": mfma_flops, "dense_equivalent_flops_per_launch": dense_flops,
            "note": "banded HTK bank: only its non-zero band is multiplied (the dense product of SURVEY 8(d) would be dense_equivalent_flops); the whole kernel is the "
                    "denominator and it is not MFMA-bound -- north_star's >= 50 % MFMA utilisation is unmet by design on this path (DESIGN 4.2)"}
    roofline = roofline_of(info, B, T, alg_bytes, fwd_us, mfma_flops)
    # what binds: the vector pipe's instruction stream -- SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles) and the
    # vector instructions per wave, from the same PMC session as `traffic` (same source-sha rule)
    qc = quoted.get(args.config, {})
    roofline.update({"valu_busy": qc.get("valu_busy"), "valu_insts_per_wave": qc.get("valu_insts_per_wave"), "kernel_cycles": qc.get("kernel_cycles"),
                     "effective_clock_ghz": qc.get("effective_clock_ghz"), "valu_busy_note": qc.get("valu_busy_note")})
    # SURVEY 8(d)'s strict forward bytes (read x + write Y: the tangent this launch also writes is the backward's operand there)
    roofline["frac_section8d"] = round((4 * B * L + (2 if args.bf16_activations else 4) * B * M * T) / (fwd_us * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)
    roofline["frac_note"] = ("frac: x + out + tangent (every tensor this launch touches, once) over the launch's live-measured duration and 8 TB/s; "
                             "frac_section8d: SURVEY 8(d)'s forward bytes only (x + out); hbm_frac_cold: the same launch on a batch no cache holds; "
                             "steady_state_frac: per-launch bytes of config 4's batch on this GPU (8 rounds of resident workgroups: launch, first touch and "
                             "drain paid once) over its un-profiled train time")
    roofline.update({"traffic": traffic, "traffic_source": traffic_src, "avg_launch_us_single_event_pair": round(fwd_pair_us, 2),
                     "other_kernels_us": {"prep": round(prep_us, 2), "backward_dot": round(bwd_us, 2)}, "mfma_stage": mfma})

    par = f"batch-sharded x{world}"
    if dist is not None:
        par += ", one all-reduce (SUM) of lambd.grad per step before the update: " + reducer
        if mar is not None:
            mar.check()
    result = {
        "metric": "spectrogram frames/sec (fwd+bwd)", "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "activations": "bf16 (output and its gradient; arithmetic, tangent and d lambd fp32)" if args.bf16_activations else "f32",
        "config": {"workload": f"BASELINE config 2 per GPU: batch {B} x {L} samples @ {sr} Hz, n_fft {info['n_fft']} "
                               f"(lambd {lam}), hop {hop}, n_mels {M}, log fused; step = nn.Module forward + backward to lambd.grad + Adam update of lambd"
                   if args.config == "c2" else f"{args.config}: batch {B} x {L} @ {sr}, lambd {lam}, hop {hop}, n_mels {M}",
                   "global_batch": B * world, "frames_per_step": frames_per_rank * world, "parallelism": par, "reducer": reducer,
                   "rccl_ranks_seen": ranks_seen,
                   "step_contents": "forward + backward to lambd.grad + torch.optim.Adam(fused, capturable) update of lambd (two torch kernels for the one scalar); "
                                    "SURVEY 8(d)'s metric asks for forward + backward only: that is c_abi_kernels (no autograd, no optimizer)",
                   "launch": ("eager from Python: torch.ops.dmel.mel_spectrogram + autograd + optimizer.step()" if chosen == "eager" else
                              f"HIP graph captured from the nn.Module step (dmel_amd.GraphedStep), {modes[chosen][1]} step(s) per replay") +
                             "; lambd stays on the device, no host synchronisation inside the timed region"},
        "module_step": module_step_info,
        "c_abi_kernels": {"frames_per_s": round(frames_per_rank / (kern_ms * 1e-3), 1), "ms_per_step": round(kern_ms, 5),
                          "note": "fused forward + dot through include/dmel.h, lambd by value, no autograd, no optimizer (round 1's headline)"},
        "roofline": roofline,
        "kernel_info": info,
    }

    if rank == 0 and world == 1 and not args.no_other_configs and args.config == "c2":
        result["other_configs"] = other_configs(torch, capi, synth, dev, kernel_times)
        # `frac` prices the forward with its input resident in cache (the timed loop re-reads one batch); the same kernel on a batch no
        # cache holds (other_configs.c2_cold: a pool > 256 MiB) beside it
        cold = result["other_configs"].get("c2_cold", {}).get("fused_forward_us_cold")
        if cold:
            roofline["hbm_frac_cold"] = round(roofline["algorithmic_bytes_per_launch"] / (cold * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)
            roofline["avg_launch_us_cold"] = cold
        r4 = result["other_configs"].get("c4_on_one_gpu", {})
        if r4.get("roofline", {}).get("frac") is not None:
            roofline["steady_state_frac"] = r4["roofline"]["frac"]
            roofline["steady_state_ns_per_frame"] = r4.get("ns_per_frame_forward")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        _flush_c_stdio()            # RCCL prints a version banner through C stdio: keep the JSON line the last one
        print(json.dumps(result), flush=True)


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:               # noqa: BLE001
        pass


def mfma_flops_of(info, B, T):
    """executed fp32 MFMA flops of ONE launch of the fused forward, by the contraction it ran (dmel_plan_info.contraction, ABI 5):
    0: non-zero 4x16 filterbank blocks x 16-row tiles x 2048 flop per v_mfma_f32_16x16x4_f32;
    1 (kTrainW, the HTK bank at n_fft 1024 / 2048): every wave issues wl_steps v_mfma_f32_4x4x1_16b_f32 per tile -- 16 blocks of (4 x 1)(1 x 4):
       512 flop per instruction whether or not a block's rows all carry frames -- x 8 waves per workgroup x the grid;
    otherwise (spectrogram modes, direct DFT, global-memory transform, the bf16x3 dense contraction: priced on another pipe): 0."""
    if info.get("kernel_path") != 0:
        return 0.0
    kind = info.get("contraction", 0)
    if kind == 1:
        return 512.0 * info["wl_steps"] * 8 * info["grid_fwd"]
    if kind == 0:
        return 2048.0 * info["fb_blocks"] * mfma_row_tiles(info, B, T)
    return 0.0


def mfma_row_tiles(info, B, T):
    """16-row tiles (8 frames x (power, tangent)) the contraction of one launch multiplies every non-zero filterbank block with"""
    fpt = info["frames_per_tile"]
    return B * ((T + fpt - 1) // fpt) * max(1, fpt // 8)


def roofline_of(info, B, T, alg_bytes, fwd_us, mfma_flops):
    """The dominant kernel against the roof that binds it.  Three floors for one launch of the fused forward: its algorithmic bytes
    at 8 TB/s; the flops of its transforms -- ONE complex n_fft-point FFT per frame when training (frame + tangent packed),
    5 n_fft log2(n_fft) flops by the usual convention -- at the fp32 vector peak; the executed fp32 MFMA flops at the fp32
    matrix peak.  `bound` names the largest floor, `frac` is that floor over the measured launch; all three fractions are printed."""
    import math
    n = info["n_fft"]
    frames = B * T
    fft_flops = 5.0 * n * math.log2(n) * frames if n >= 2 else 0.0
    floors = {"hbm": alg_bytes / (HBM_PEAK_GBS * 1e9), "valu": fft_flops / (FP32_VALU_PEAK_TFLOPS * 1e12),
              "mfma": mfma_flops / (FP32_MFMA_PEAK_TFLOPS * 1e12)}
    bound = max(floors, key=floors.get)
    t = fwd_us * 1e-6
    r = {"bound": bound, "kernel": f"dmel_fwd_kernel<{n},train>"}
    if bound == "hbm":
        r.update({"achieved": round(alg_bytes / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s"})
    else:
        fl = fft_flops if bound == "valu" else mfma_flops
        r.update({"achieved": round(fl / t / 1e12, 2), "peak": FP32_VALU_PEAK_TFLOPS if bound == "valu" else FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"})
    r["frac"] = round(floors[bound] / t, 4)
    r["fracs"] = {k: round(v / t, 4) for k, v in floors.items()}
    r["floors_us"] = {k: round(1e6 * v, 2) for k, v in floors.items()}
    r["algorithmic_bytes_per_launch"] = alg_bytes
    r["fft_flops_per_launch"] = fft_flops
    r["avg_launch_us"] = round(fwd_us, 2)
    return r


def profile_quotes():
    """({config: {dmel_fwd_kernel_bytes_per_launch, valu_busy, valu_insts_per_wave, ...}}, source note) from profiles/hbm_traffic.json -- HBM-side bytes
    and SQ counters come from rocprofv3 PMC passes (they cannot be read from inside the process); the committed measurement is quoted
    only if it was taken on THIS kernel source, otherwise nothing is"""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(tpath):
        return {}, None
    try:
        import hashlib
        tj = json.load(open(tpath))
        src = open(os.path.join(ROOT, "differentiable-mel-spectrogram_amd", "csrc", "dmel_fwd.hip"), "rb").read()
        sha = hashlib.sha256(src).hexdigest()[:16]
        if tj.get("kernel_source_sha16") == sha:
            return {k: v for k, v in tj.items() if isinstance(v, dict)}, f"profiles/hbm_traffic.json (rocprofv3 --pmc, dmel_fwd.hip sha16 {sha})"
        return {}, f"profiles/hbm_traffic.json was measured on another build of dmel_fwd.hip (have {sha}): not quoted"
    except Exception:                                                   # noqa: BLE001
        return {}, None


def other_configs(torch, capi, synth, dev, kernel_times):
    """BASELINE configs 3 and 5 (both n_fft 2048) and the shapes of the reference's own experiments, not bench lines of their own:
    kernel times of the forward + dot through the C ABI and the same roofline accounting, plus config 5's front end isolated
    inside a real MelConvNet training step."""
    res = {}
    quoted, _ = profile_quotes()

    def one(name, B, L, sr, lam, hop, M):
        T = L // hop + 1
        x = torch.from_numpy(synth.waveforms(B, L, seed=7)).to(dev)
        g = torch.from_numpy(synth.cotangent((B, 1, M, T), seed=8)).to(dev)
        out, tan, dl = torch.empty((B, 1, M, T), device=dev), torch.empty((B, 1, M, T), device=dev), torch.zeros(1, device=dev)
        plan = capi.Plan(L, hop, M, sr, max_batch=B)
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        info = plan.info()
        fwd_us, _, prep_us, bwd_us, step_us = kernel_times(plan, x, g, out, tan, dl, lam, out.numel(), B, False)
        alg = 4 * (B * L + 2 * B * M * T)
        mflops = mfma_flops_of(info, B, T)
        rl = roofline_of(info, B, T, alg, fwd_us, mflops)
        rl["frac_section8d"] = round(4.0 * (B * L + B * M * T) / (fwd_us * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)
        q = quoted.get({"c4_on_one_gpu": "c4", "esc50_x0.3": "esc_n4096", "esc50_lambd700": "esc_n8192"}.get(name, name), {})
        rl["traffic"] = q.get("dmel_fwd_kernel_bytes_per_launch")
        rl["valu_busy"], rl["valu_insts_per_wave"] = q.get("valu_busy"), q.get("valu_insts_per_wave")
        rl["kernel_cycles"], rl["effective_clock_ghz"] = q.get("kernel_cycles"), q.get("effective_clock_ghz")
        return {"workload": f"batch {B} x {L} @ {sr} Hz, n_fft {info['n_fft']} (lambd {round(lam, 1)}), hop {hop}, n_mels {M}",
                "frames_per_step": B * T, "step_us": round(step_us, 2), "frames_per_s": round(B * T / (step_us * 1e-6), 1),
                "kernels_us": {"prep_partial_sums": round(prep_us, 2), "fused_forward": round(fwd_us, 2), "backward_dot": round(bwd_us, 2)},
                "roofline": rl,
                "lds_bytes": info["lds_bytes"], "grid": info["grid_fwd"]}

    for name in ("c3", "c5"):
        res[name] = one(name, *CONFIGS[name])
    # BASELINE config 4's GLOBAL batch (2048 x 16000 = 65 536 frames) on ONE GPU: the many-round regime of the same kernel (a launch
    # of 8 rounds of resident workgroups: later rounds hide the launch / first-touch / drain of a one-round launch)
    try:
        B4 = 8 * CONFIGS["c2"][0]
        r4 = one("c4_on_one_gpu", B4, *CONFIGS["c2"][1:])
        r4["ns_per_frame_forward"] = round(1e3 * r4["kernels_us"]["fused_forward"] / r4["frames_per_step"], 4)
        r4["ns_per_frame_step"] = round(1e3 * r4["step_us"] / r4["frames_per_step"], 4)
        res["c4_on_one_gpu"] = r4
    except Exception as e:                                              # noqa: BLE001
        res["c4_on_one_gpu"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # the "mel params" of north_star: lambd AND the (513 x 128) filterbank trained together at config 2 (learnable_fb=True): the dense
    # contraction in the forward, the filterbank gradient's GEMM, both optimizers -- per-step time of the HIP-graph replayed nn.Module
    # step for the exact fp32-MFMA path (with and without the saved spectrogram) and the split-bf16 path (DMEL_FLAG_MFMA_BF16X3)
    try:
        res["trainable_filterbank_c2"] = trainable_filterbank(torch, synth, dev)
    except Exception as e:                                              # noqa: BLE001
        res["trainable_filterbank_c2"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # config 2 with COLD inputs: the same trains of launches, every forward reading another resident batch of a pool > 256 MiB
    # (beyond L2 and the Infinity Cache), next to the warm figure (ONE batch replayed: it lives in cache)
    try:
        res["c2_cold"] = c2_cold(torch, capi, synth, dev)
    except Exception as e:                                              # noqa: BLE001
        res["c2_cold"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # the shapes of the reference's own experiments (search_spaces.py:4-33 ESC-50, :36-66 Audio-MNIST: 8 kHz, hop 80, 64 mels,
    # init_lambd = 8000 x / 6 for x = 0.01, 0.035, 0.3 -> n_fft 128, 512, 4096) and one lambd the run may drift to (n_fft 8192)
    ref = {}
    for name, (B, L, lam) in {"esc50_x0.01": (32, 40000, 8000 * 0.01 / 6), "esc50_x0.035": (32, 40000, 8000 * 0.035 / 6),
                              "esc50_x0.3": (32, 40000, 8000 * 0.3 / 6), "esc50_lambd700": (32, 40000, 700.0),
                              "audio_mnist_x0.3": (64, 8000, 8000 * 0.3 / 6)}.items():
        try:
            ref[name] = one(name, B, L, 8000, lam, 80, 64)
        except Exception as e:                                          # noqa: BLE001
            ref[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    res["reference_experiment_shapes"] = ref
    try:
        res["c5"]["train_step"] = c5_train_step(torch, synth, dev)
    except Exception as e:                                              # noqa: BLE001
        res["c5"]["train_step"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # the paper's own model: MelPANNsNet / Cnn6 (models.py:138-166, panns.py:135-202) at the ESC-50 shape of search_spaces.py:4-33
    try:
        res["c5_panns"] = {"train_step": panns_train_step(torch, synth, dev)}
    except Exception as e:                                              # noqa: BLE001
        res["c5_panns"] = {"train_step": {"error": f"{type(e).__name__}: {e}"[:300]}}
    return res


def trainable_filterbank(torch, synth, dev, k=10, replays=30):
    """lambd and the filterbank matrix trained together (models.py:42-48 made a parameter) at BASELINE config 2."""
    import dmel_amd
    from dmel_amd import GraphedStep, MelSpectrogramLayer
    B, L, sr, lam, hop, M = CONFIGS["c2"]
    T = L // hop + 1
    x = torch.from_numpy(synth.waveforms(B, L, seed=0)).to(dev)
    g = torch.from_numpy(synth.cotangent((B, 1, M, T), seed=1)).to(dev)
    out = {}
    F = 2 ** (int(6 * lam) - 1).bit_length() // 2 + 1
    flops = 2.0 * B * T * F * M
    for name, kw, native in (("fp32_mfma_recompute", dict(save_spec=False), False), ("fp32_mfma", {}, False), ("bf16x3", dict(mfma="bf16x3"), False),
                             ("bf16x3_LambdAdam", dict(mfma="bf16x3"), True)):
        layer = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, device=str(dev), optimized=True,
                                    log=True, learnable_fb=True, **kw).to(dev)
        # lr 0: the same kernels and the same work, parameters that stay put (any lr > 0 moves the 65 000 zero entries of the HTK matrix
        # together, and after a few hundred steps the lowest band -- ONE non-zero bin -- of the quietest frames turns negative under the
        # log: what torch does with a leaf filterbank too, but not what a timing loop should run into)
        opt = (dmel_amd.LambdAdam(layer.parameters(), lr=0.0) if native
               else torch.optim.Adam(layer.parameters(), lr=0.0, fused=True, capturable=True))

        def step():
            opt.zero_grad(set_to_none=True)
            layer(x).backward(g)
            opt.step()

        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
        except Exception as e:                                          # noqa: BLE001
            raise RuntimeError(f"{name}: {e}") from e
        gs = GraphedStep(step, [layer], max_ahead=4, steps_per_replay=k)
        for _ in range(8):
            gs()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(replays):
                gs()
            e1.record()
            torch.cuda.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1) / (replays * k))
        us = sorted(ts)[1]
        out[name] = {"step_us": round(us, 2), "frames_per_s": round(B * T / (us * 1e-6), 1),
                     "optimizer": "dmel_amd.LambdAdam (one launch per parameter)" if native else "torch.optim.Adam(fused, capturable) on both parameters"}
        gs.close()
        del gs, opt, layer
    out["note"] = (f"HIP-graph replay of the nn.Module step ({k} steps per replay, median of 3 trains); dense contraction 2 x {flops / 1e9:.3f} GFLOP per step "
                   "(forward: power + tangent rows; gradient GEMM); default is fp32_mfma (exact); bf16x3 is opt-in (mfma='bf16x3')")
    return out


def c2_cold(torch, capi, synth, dev, pool_n=24, ntrain=120):
    """Forward-kernel time at config 2 when x comes from HBM: trains of (forward(x_i) + dot) minus trains of the dot alone, x_i rotating
    over `pool_n` resident batches (24 x 16.4 MB = 393 MB > the 256 MiB Infinity Cache), against the same trains on ONE batch."""
    B, L, sr, lam, hop, M = CONFIGS["c2"]
    T = L // hop + 1
    cur = torch.cuda.current_stream(dev)
    pool = [torch.from_numpy(synth.waveforms(B, L, seed=2000 + i)).to(dev) for i in range(pool_n)]
    g = torch.from_numpy(synth.cotangent((B, 1, M, T), seed=9)).to(dev)
    out, tan, dl = torch.empty((B, 1, M, T), device=dev), torch.empty((B, 1, M, T), device=dev), torch.zeros(1, device=dev)
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    n = out.numel()

    def train(xs, with_fwd=True):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(cur)
        for i in range(ntrain):
            if with_fwd:
                plan.forward(xs[i % len(xs)].data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, cur.cuda_stream)
            plan.backward(g.data_ptr(), tan.data_ptr(), n, dl.data_ptr(), cur.cuda_stream)
        ev1.record(cur)
        torch.cuda.synchronize()
        return 1e3 * ev0.elapsed_time(ev1) / ntrain

    for _ in range(2):
        train(pool)
    dot = sorted(train(pool, False) for _ in range(3))[1]
    cold = sorted(train(pool) for _ in range(5))[2]
    warm = sorted(train(pool[:1]) for _ in range(5))[2]
    alg = 4 * (B * L + 2 * B * M * T)
    return {"pool_batches": pool_n, "pool_mib": round(pool_n * B * L * 4 / 2**20, 1),
            "fused_forward_us_cold": round(cold - dot, 2), "fused_forward_us_warm": round(warm - dot, 2), "backward_dot_us": round(dot, 2),
            "step_us_cold": round(cold, 2), "step_us_warm": round(warm, 2),
            "frames_per_s_cold": round(B * T / (cold * 1e-6), 1), "frames_per_s_warm": round(B * T / (warm * 1e-6), 1),
            "hbm_frac_cold": round(alg / ((cold - dot) * 1e-6) / (HBM_PEAK_GBS * 1e9), 4),
            "note": "trains of forward + dot launches through the C ABI (median of 5); cold: x rotates over the pool so every forward reads "
                    "its 16.4 MB from HBM; warm: one batch replayed (what the headline's resident batch sees)"}


def c5_train_step(torch, synth, dev, steps=8):
    """config 5: ESC-50-shaped clips through MelConvNet (models.py:105-136), CrossEntropy, Adam with the two learning-rate
    groups of main.py:36-53; the front end's kernels isolated with the library's HIP-event profiling."""
    from dmel_amd import nets
    B, L, sr, lam, hop, M = CONFIGS["c5"]
    ncls = 50
    torch.manual_seed(0)
    net = nets.MelConvNet(ncls, torch.tensor(lam), str(dev), M, sr, L, hop_length=hop, optimized=True, energy_normalize=True).to(dev)
    opt = nets.make_optimizer(net, lr_model=1e-4, lr_tf=1.0)
    loss_fn = torch.nn.CrossEntropyLoss()
    x = torch.from_numpy(synth.waveforms(B, L, seed=0)).to(dev)
    y = (torch.arange(B, device=dev) * 7) % ncls

    def step():
        opt.zero_grad(set_to_none=True)
        logits, _ = net(x)
        loss = loss_fn(logits, y)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    plan = net.spectrogram_layer._plan_for(torch.device(dev))
    plan.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pr = plan.get_profile()
    plan.set_profiling(False)
    front_ms = (pr["prep_ms"] + pr["fwd_ms"] + pr["bwd_ms"]) / steps
    return {"step_ms": round(1e3 * dt, 3), "frontend_ms": round(front_ms, 4), "frontend_share": round(front_ms / (1e3 * dt), 4),
            "frontend_frames_per_s": round(B * (L // hop + 1) / (front_ms * 1e-3), 1),
            "launches_per_step": {"prep": pr["prep_launches"] / steps, "forward": pr["fwd_launches"] / steps, "backward": pr["bwd_launches"] / steps},
            "note": "MelConvNet + CrossEntropy + Adam (lr_model 1e-4, lr_tf 1.0), batch 32; front end = HIP-event time of its kernels "
                    "(each event pair also brackets the launch packets)"}


def panns_train_step(torch, synth, dev, steps=8):
    """The model the paper trains on ESC-50 (search_spaces.py:7: MelPANNsNet = DMEL layer -> log -> Cnn6, models.py:138-166,
    panns.py:135-202) at its own shape: 32 clips of 40 000 samples at 8 kHz, hop 80, 64 mel bands, init_lambd = 8000 x 0.3 / 6 = 400
    (n_fft 4096), CrossEntropy, Adam with the two learning-rate groups of main.py:36-53; the front end's kernels isolated with the
    library's HIP-event profiling, as in c5.train_step."""
    from dmel_amd import nets, panns
    B, L, sr, lam, hop, M, ncls = 32, 40000, 8000, 400.0, 80, 64, 50
    torch.manual_seed(0)
    net = panns.MelPANNsNet(ncls, torch.tensor(lam), str(dev), M, sr, L, hop_length=hop, optimized=True, energy_normalize=True).to(dev)
    opt = nets.make_optimizer(net, lr_model=1e-4, lr_tf=1.0)
    loss_fn = torch.nn.CrossEntropyLoss()
    x = torch.from_numpy(synth.waveforms(B, L, seed=0)).to(dev)
    y = (torch.arange(B, device=dev) * 7) % ncls

    def step():
        opt.zero_grad(set_to_none=True)
        logits, _ = net(x)
        loss_fn(logits, y).backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    plan = net.spectrogram_layer._plan_for(torch.device(dev))
    plan.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pr = plan.get_profile()
    plan.set_profiling(False)
    front_ms = (pr["prep_ms"] + pr["fwd_ms"] + pr["bwd_ms"]) / steps
    T = L // hop + 1
    return {"step_ms": round(1e3 * dt, 3), "frontend_ms": round(front_ms, 4), "frontend_share": round(front_ms / (1e3 * dt), 4),
            "frontend_frames_per_s": round(B * T / (front_ms * 1e-3), 1), "n_fft": int(plan.info()["n_fft"]),
            "parameters": int(sum(p.numel() for p in net.parameters())),
            "launches_per_step": {"prep": pr["prep_launches"] / steps, "forward": pr["fwd_launches"] / steps, "backward": pr["bwd_launches"] / steps},
            "note": "MelPANNsNet (DMEL layer + log + Cnn6, 4.6 M parameters) + CrossEntropy + Adam (lr_model 1e-4, lr_tf 1.0), batch 32 x 40000 @ 8 kHz, "
                    "hop 80, 64 mels, lambd 400 (n_fft 4096); eager steps; front end = HIP-event time of its kernels"}


if __name__ == "__main__":
    main()
