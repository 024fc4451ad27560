import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import dmel_amd
from dmel_amd import MelSpectrogramLayer, synth
B, L, sr, lam, hop, M = 256, 16000, 16000, 128.0, 512, 128
x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
g = torch.randn(B, 1, M, L // hop + 1, device="cuda")
layer = MelSpectrogramLayer(torch.tensor(lam), n_mels=M, n_points=L, sample_rate=sr, hop_length=hop, optimized=True, log=True).cuda()
def t(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
with torch.no_grad():
    print("forward no_grad", t(lambda: layer(x)))
print("forward grad", t(lambda: layer(x)))
def fb():
    y = layer(x); y.backward(g)
print("fwd + y.backward(g)", t(fb))
def fb2():
    y = layer(x); (y * g).sum().backward()
print("fwd + (y*g).sum().backward()", t(fb2))
opt = torch.optim.SGD([layer.lambd], lr=1e-9)
def fb3():
    opt.zero_grad(set_to_none=True); y = layer(x); y.backward(g); opt.step()
print("fwd + backward + sgd step (lambd changes -> host read)", t(fb3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): fb()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


# baseline: a do-nothing custom Function through the same engine path, and a pure-torch op of similar shape
class _Noop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lam, buf):
        return buf

    @staticmethod
    def backward(ctx, g_):
        return torch.zeros((), device=g_.device), None


lam_t = torch.tensor(1.0, device="cuda", requires_grad=True)
buf = torch.empty_like(g)
def noop():
    y = _Noop.apply(lam_t, buf); y.backward(g)
print("do-nothing Function fwd + backward", t(noop))
def puretorch():
    y = buf * lam_t; y.backward(g)
print("pure torch (buf * lam).backward(g)", t(puretorch))
