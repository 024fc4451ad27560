import sys, torch, time
sys.path.insert(0, '/root/repo')
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import dmel_amd
from dmel_amd import MelSpectrogramLayer, synth
for (B, L, lam) in ((64, 8000, 400.0), (32, 40000, 400.0), (64, 8192, 400.0)):
    x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
    lay = MelSpectrogramLayer(torch.tensor(lam), n_mels=64, n_points=L, sample_rate=8000, hop_length=80, device="cuda:0", optimized=False, log=True).to("cuda:0")
    y = lay(x); y.sum().backward(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        y = lay(x); y.sum().backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(B, L, "n_fft", 2 * L, "path", lay.plan_info()["kernel_path"], f"{dt*1e3:.2f} ms/step", f"{B*(L//80+1)/dt/1e6:.2f} M frames/s")
