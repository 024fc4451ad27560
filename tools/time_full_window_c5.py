import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dmel_amd
from dmel_amd import capi, synth
B, L, lam, hop, M, sr = 32, 220500, 9000.0, 441, 128, 44100
T = L // hop + 1
x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
out = torch.empty((B, 1, M, T), device="cuda"); tan = torch.empty_like(out); g = torch.randn_like(out)
dl = torch.zeros(1, device="cuda")
plan = capi.Plan(L, hop, M, sr, max_batch=B)
s = torch.cuda.current_stream().cuda_stream
def step():
    plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), True, 1e-10, s, extra_flags=capi.DMEL_FLAG_FULL_WINDOW)
    plan.backward(g.data_ptr(), tan.data_ptr(), out.numel(), dl.data_ptr(), s)
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps(dict(shape=f"{B}x{L}", n_fft=2 * L, kernel_path=plan.info()["kernel_path"], s_per_step=round(dt, 3), frames_per_s=round(B * T / dt), finite=bool(torch.isfinite(out).all()))))
