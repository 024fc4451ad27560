#!/usr/bin/env python3
"""bench.py -- frames/s (forward + backward) of the DMEL layer on N MI355X, BASELINE.json's metric.

A step = one pass of the hot path over one batch of synthetic waveforms already resident in HBM:
  forward : DC removal -> Gaussian-window STFT -> |.|^2 -> mel contraction -> log(. + 1e-10)
            (dmel_forward: ONE fused kernel at this clip length, carrying d out / d lambd)
  backward: lambd.grad = <grad_out, tangent>   (dmel_backward: one deterministic fp64 dot kernel)
  N > 1   : + one all-reduce (RCCL) of the scalar gradient, batch sharded over ranks (weak scaling)
Workload at every N: BASELINE config 2 per GPU (256 x 16000 @16 kHz, n_fft 1024 (lambd 128),
hop 512, 128 mels; config 4 is exactly 8 of these).  The step is driven through the C ABI
(include/dmel.h): two kernel launches (fused forward, dot) queued eagerly on the current stream, the
host runs ahead of the device.  `--graph` replays the step from a HIP graph instead (slower here), and
the nn.Module path (autograd, one host read of lambd per step) is reported beside it as "module_path".

Prints ONE JSON line (rank 0).  See DESIGN.md for the roofline accounting.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

CONFIGS = {
    # name: (B per GPU, L, sample_rate, lambd, hop, n_mels)
    "c1": (4, 16000, 16000, 64.0, 256, 64),
    "c2": (256, 16000, 16000, 128.0, 512, 128),
    "c3": (32, 160000, 16000, 256.0, 512, 128),
    "c5": (32, 220500, 44100, 256.0, 441, 128),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 (matrix), v_mfma_f32_16x16x4_f32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--graph", action="store_true", help="replay each step from a HIP graph instead of launching it eagerly "
                    "(measured slower here: ~5 us of per-replay overhead against a ~35 us step)")
    ap.add_argument("--bf16-activations", action="store_true", help="store the log-mel output and read its gradient as bf16 "
                    "(DMEL_FLAG_OUT_BF16); the arithmetic, the tangent and d lambd stay fp32.  Default: fp32, the reference's output type")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-module-path", action="store_true")
    return ap.parse_args()


def cpu_baseline(cfg):
    """The reference's algorithm on this host's cores (rank 0, N=1 only): the C oracle and the batched
    torch restatement, each best of 5 on one full batch of the same workload; the faster one is reported."""
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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("DMEL_BENCH_FORCE_DIST") == "1":     # the env knob exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from dmel_amd import MelSpectrogramLayer, capi, synth
    from dmel_amd import dist as ddist

    cfg = CONFIGS[args.config]
    B, L, sr, lam, hop, M = cfg
    T = L // hop + 1
    frames_per_rank = B * T
    # every rank owns a different shard of the global batch (seeded by rank); inputs are resident before timing
    x = torch.from_numpy(synth.waveforms(B, L, seed=100 * rank)).to(dev)
    g = torch.from_numpy(synth.cotangent((B, 1, M, T), seed=1 + 100 * rank)).to(dev)
    act = torch.bfloat16 if args.bf16_activations else torch.float32
    g = g.to(act)
    out = torch.empty((B, 1, M, T), dtype=act, device=dev)
    tan = torch.empty((B, 1, M, T), dtype=torch.float32, device=dev)
    RING = 32  # gradient buffers: the all-reduce of step k may still be in flight while steps k+1 .. k+15 run
    dl = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in range(RING)]
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    count = out.numel()

    def step_kernels(stream_ptr, k):
        plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, stream_ptr,
                     extra_flags=capi.DMEL_FLAG_OUT_BF16 if args.bf16_activations else 0)
        plan.backward(g.data_ptr(), tan.data_ptr(), count, dl[k % RING].data_ptr(), stream_ptr, grad_bf16=args.bf16_activations)

    cur = torch.cuda.current_stream(dev)
    step_kernels(cur.cuda_stream, 0)          # builds the per-n_fft tables (hipMalloc) outside any capture
    torch.cuda.synchronize()
    info = plan.info()

    graphs = None
    if args.graph:
        graphs = []
        for k in range(RING):                 # one graph per gradient buffer
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                step_kernels(torch.cuda.current_stream(dev).cuda_stream, k)
            graphs.append(gr)

    # one all-reduce of the scalar gradient per step, natively over RCCL on its own stream (dmel_comm_*),
    # falling back to torch.distributed (stream-ordered, no overlap) if the native communicator is unavailable
    sar = ddist.ScalarAllReduce() if dist is not None else None
    main_stream = torch.cuda.current_stream(dev).cuda_stream
    tick = [None] * 64
    nstep = [0]

    def one_step(_k):
        k = nstep[0]
        nstep[0] += 1
        i = k % RING
        # Buffer i was last reduced by the collective of step k-32.  Ordering the compute stream after a collective
        # costs a barrier packet, so it is done once per 16 steps, on the collective of step k-17: the
        # communicator's stream runs them in order, hence everything up to k-17 is then complete, which covers the
        # buffers steps k .. k+15 overwrite.
        if sar is not None and k % 16 == 0 and k >= 17 and tick[(k - 17) % 64] is not None:
            sar.wait(tick[(k - 17) % 64], main_stream)
        if graphs is not None:
            graphs[i].replay()
        else:
            step_kernels(main_stream, k)
        if sar is not None:
            tick[k % 64] = sar.reduce_async(dl[i], main_stream)

    def drain():
        k = nstep[0]
        if sar is not None and k >= 1 and tick[(k - 1) % 64] is not None:
            sar.wait(tick[(k - 1) % 64], main_stream)      # the last collective (they complete in order)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        one_step(k)
    drain()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(k)
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * frames_per_rank * args.steps / elapsed

    # ---- per-kernel device time with HIP events around each launch (eager pass, same stream) --------
    plan.set_profiling(True)
    nprof = max(20, min(args.steps, 200))
    for k in range(nprof):
        step_kernels(cur.cuda_stream, k)
    torch.cuda.synchronize()
    prof = plan.get_profile()
    plan.set_profiling(False)
    fwd_pair_us = 1e3 * prof["fwd_ms"] / max(1, prof["fwd_launches"])
    prep_us = 1e3 * prof["prep_ms"] / max(1, prof["prep_launches"])
    bwd_us = 1e3 * prof["bwd_ms"] / max(1, prof["bwd_launches"])
    # An event pair around ONE launch also times the launch packets around it (1.5-4 us, varying from box to box).  The
    # dominant kernel's average launch duration is therefore taken from trains of launches between two HIP events on the
    # launch stream: a train of whole steps (forward + dot, the timed region's own mix, so the forward sees the cache state
    # the dot kernel leaves) minus a train of the dot launches alone.  This is what rocprofv3 reports as the dispatch
    # duration plus the sub-microsecond gap between dependent dispatches.
    ntrain = 100
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.cuda.stream(cur):
        ev[0].record(cur)
        for k in range(ntrain):
            step_kernels(cur.cuda_stream, k)
        ev[1].record(cur)
        ev[2].record(cur)
        for k in range(ntrain):
            plan.backward(g.data_ptr(), tan.data_ptr(), count, dl[k % RING].data_ptr(), cur.cuda_stream, grad_bf16=args.bf16_activations)
        ev[3].record(cur)
    torch.cuda.synchronize()
    fwd_us = 1e3 * (ev[0].elapsed_time(ev[1]) - ev[2].elapsed_time(ev[3])) / ntrain
    if prep_us > 0:
        fwd_us -= prep_us          # long clips: every forward also launched the partial-sum kernel
    # algorithmic bytes of ONE launch of the fused forward kernel:
    # read x once + write out and tangent once, fp32
    alg_bytes = 4 * (B * L + B * M * T) + (2 if args.bf16_activations else 4) * B * M * T
    achieved = alg_bytes / (fwd_us * 1e-6) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.config, {}).get("dmel_fwd_kernel_bytes_per_launch")
        except Exception:
            traffic = None
    # the contraction stage on the matrix cores: executed fp32 MFMA flops of one launch = non-zero 4x16 filterbank blocks
    # x 16-row tiles x 2048 flop per v_mfma_f32_16x16x4_f32 (SQ_INSTS_MFMA in profiles/r01_pmc_sq_c2.json counts the same
    # instructions), against the dense fp32 matrix peak
    mfma_flops = 2048.0 * info["fb_blocks"] * (B * ((T + info["frames_per_tile"] - 1) // info["frames_per_tile"])) if info["kernel_path"] == 0 else 0.0
    mfma = {"executed_tflops": round(mfma_flops / (fwd_us * 1e-6) / 1e12, 2), "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
            "frac": round(mfma_flops / (fwd_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "note": "banded filterbank: only the non-zero blocks are multiplied (dense would be fb_blocks_dense); the kernel is not MFMA-bound"}
    roofline = {"bound": "hbm", "kernel": f"dmel_fwd_kernel<{info['n_fft']},train>", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": round(fwd_us, 2),
                "avg_launch_us_single_event_pair": round(fwd_pair_us, 2),
                "other_kernels_us": {"prep": round(prep_us, 2), "backward_dot": round(bwd_us, 2)}, "mfma_stage": mfma}

    result = {
        "metric": "spectrogram frames/sec (fwd+bwd)", "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "activations": "bf16 (output and its gradient; arithmetic, tangent and d lambd fp32)" if args.bf16_activations else "f32",
        "config": {"workload": f"BASELINE config 2 per GPU: batch {B} x {L} samples @ {sr} Hz, n_fft {info['n_fft']} "
                               f"(lambd {lam}), hop {hop}, n_mels {M}, log fused, fwd + backward to lambd.grad"
                   if args.config == "c2" else f"{args.config}: batch {B} x {L} @ {sr}, lambd {lam}, hop {hop}, n_mels {M}",
                   "global_batch": B * world, "frames_per_step": frames_per_rank * world,
                   "parallelism": f"batch-sharded x{world}" + (f", one all-reduce of d lambd per step (RCCL, " + ("native dmel_comm on its own stream, up to 17 in flight" if sar.native else "torch.distributed fallback: " + sar.why) + ")" if sar is not None else ""),
                   "launch": "hip-graph replay" if args.graph else "eager, 2 launches per step (fused forward, dot)"},
        "roofline": roofline,
        "kernel_info": info,
    }

    if rank == 0 and world == 1:
        if not args.no_module_path:
            layer = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop,
                                        device=str(dev), optimized=True, log=True).to(dev)
            n_mod = max(10, min(args.steps, 100))
            for i in range(n_mod + 5):
                if i == 5:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                layer.lambd.grad = None
                (layer(x) * g).sum().backward()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / n_mod
            result["module_path"] = {"frames_per_s": round(frames_per_rank / dt, 1), "ms_per_step": round(1e3 * dt, 4),
                                     "note": "nn.Module + autograd, eager, includes the loss (mul+sum) and one host read of lambd per step"}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
