"""GPU tests of the opt-in optimizer for the layer's parameter (dmel_adam_step, dmel_amd.LambdAdam) against torch.optim.Adam."""
import numpy as np
import pytest
import torch

import cases as C

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("kw", [dict(lr=1e-3), dict(lr=1.0, betas=(0.8, 0.99), eps=1e-6), dict(lr=0.05, weight_decay=0.01),
                                dict(lr=0.02, maximize=True)])
def test_lambd_adam_follows_torch_adam(kw):
    """the same gradients through both optimizers, 300 steps: parameters and moments stay within a few ulps"""
    from dmel_amd import LambdAdam
    gen = torch.Generator(device="cpu").manual_seed(3)
    for shape in ((), (5,), (3, 7), (513, 128)):                                  # the last: many workgroups, the ticket word
        p0 = torch.randn(shape, generator=gen) * 10.0 + 100.0
        a = torch.nn.Parameter(p0.clone().to(DEV))
        b = torch.nn.Parameter(p0.clone().to(DEV))
        ours, ref = LambdAdam([a], **kw), torch.optim.Adam([b], **kw)
        for step in range(300):
            g = (torch.randn(shape, generator=gen) * (1.0 + 0.01 * step)).to(DEV)
            a.grad = g.clone(); b.grad = g.clone()
            ours.step(); ref.step()
        torch.cuda.synchronize()
        assert torch.allclose(a.detach(), b.detach(), rtol=2e-6, atol=1e-6), (shape, float((a - b).abs().max()))
        ma, mb = ours.state[a]["exp_avg"], ref.state[b]["exp_avg"]
        assert torch.allclose(ma, mb, rtol=1e-5, atol=2e-6 * float(mb.abs().max()))           # (a moment near zero is a difference of large terms)
        va, vb = ours.state[a]["exp_avg_sq"], ref.state[b]["exp_avg_sq"]
        assert torch.allclose(va, vb, rtol=1e-5, atol=1e-6 * float(vb.abs().max()))
        assert float(ours.state[a]["step"]) == 300.0


def test_lambd_adam_trains_the_layer_inside_a_hip_graph():
    """the whole step -- forward, backward to lambd.grad, the one-launch update -- captured once and replayed: lambd ends where
    torch's capturable Adam puts it"""
    from dmel_amd import LambdAdam, MelSpectrogramLayer
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)

    def mk():
        return MelSpectrogramLayer(torch.tensor(float(case["lambd"])), n_mels=case["n_mels"], n_points=case["L"], sample_rate=case["sr"],
                                   hop_length=case["hop"], device=DEV, optimized=True, log=True).to(DEV)

    la, lb = mk(), mk()
    oa = LambdAdam([la.lambd], lr=1e-3)
    ob = torch.optim.Adam([lb.lambd], lr=1e-3, capturable=True)

    def step(lay, o):
        o.zero_grad(set_to_none=True)
        (lay(x) * g).sum().backward()
        o.step()

    step(la, oa); step(lb, ob)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step(la, oa)
    for _ in range(20):
        gr.replay()
        step(lb, ob)
    torch.cuda.synchronize()
    va, vb = float(la.lambd.detach()), float(lb.lambd.detach())
    assert abs(va - vb) <= 2e-6 * abs(vb) and va != float(case["lambd"])
    assert float(oa.state[la.lambd]["step"]) == 21.0                        # one eager step + 20 replays (the capture itself executes nothing)


def test_lambd_adam_rejects_what_it_is_not_for():
    from dmel_amd import LambdAdam, capi
    with pytest.raises(ValueError):
        LambdAdam([torch.nn.Parameter(torch.zeros(4))])                       # a CPU parameter
    with pytest.raises(ValueError):
        LambdAdam([torch.nn.Parameter(torch.zeros(8, device=DEV, dtype=torch.float64))])
    with pytest.raises(ValueError):
        LambdAdam([torch.nn.Parameter(torch.zeros(4, device=DEV))], betas=(1.0, 0.9))
    t = torch.zeros(4, device=DEV)
    with pytest.raises(RuntimeError):
        capi.adam_step(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 0, 0, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, False, 0)
    big = torch.zeros(5000, device=DEV)
    with pytest.raises(RuntimeError, match="ticket"):
        capi.adam_step(big.data_ptr(), big.data_ptr(), big.data_ptr(), big.data_ptr(), t.data_ptr(), 0, 5000, 1e-3, 0.9, 0.999, 1e-8, 0.0, False, 0)


def test_lambd_adam_state_dict_round_trip_through_the_cpu():
    """save -> torch.load(map_location='cpu') -> load_state_dict -> step (the usual resume): the step count and the moments are back on
    the parameter's device as fp32 before a pointer of them reaches the kernel (a host pointer went to dmel_adam_kernel),
    and the resumed run continues exactly where an uninterrupted one is"""
    import io
    from dmel_amd import LambdAdam
    gen = torch.Generator(device="cpu").manual_seed(5)
    for shape in ((), (513, 128)):
        p0 = torch.randn(shape, generator=gen) + 50.0
        grads = [torch.randn(shape, generator=gen).to(DEV) for _ in range(12)]
        a = torch.nn.Parameter(p0.clone().to(DEV))
        ref = torch.nn.Parameter(p0.clone().to(DEV))
        oa, oref = LambdAdam([a], lr=0.01), LambdAdam([ref], lr=0.01)
        for g in grads[:6]:
            a.grad = g.clone(); ref.grad = g.clone()
            oa.step(); oref.step()
        buf = io.BytesIO()
        torch.save({"opt": oa.state_dict(), "p": a.detach().cpu()}, buf)
        buf.seek(0)
        ck = torch.load(buf, map_location="cpu")
        assert "ticket" not in next(iter(ck["opt"]["state"].values()))                  # the ticket word is private, not checkpointed
        b = torch.nn.Parameter(ck["p"].to(DEV))
        ob = LambdAdam([b], lr=0.01)
        ob.load_state_dict(ck["opt"])
        for g in grads[6:]:
            b.grad = g.clone(); ref.grad = g.clone()
            ob.step(); oref.step()
        torch.cuda.synchronize()
        st = ob.state[b]
        assert st["step"].device.type == "cuda" and st["step"].dtype == torch.float32 and float(st["step"]) == 12.0
        assert st["exp_avg"].device.type == "cuda" and st["exp_avg_sq"].device.type == "cuda"
        assert torch.equal(b.detach(), ref.detach()), shape
    # a state whose tensors were left on the host by hand (no capturable policy involved) is repaired too
    c = torch.nn.Parameter(torch.tensor(3.0, device=DEV))
    oc = LambdAdam([c], lr=0.1)
    c.grad = torch.tensor(1.0, device=DEV); oc.step()
    for k in ("step", "exp_avg", "exp_avg_sq"):
        oc.state[c][k] = oc.state[c][k].cpu()
    oc.state[c]["ticket"] = torch.zeros((), dtype=torch.float32)                       # what a round-3 checkpoint carried
    c.grad = torch.tensor(1.0, device=DEV); oc.step()
    torch.cuda.synchronize()
    assert float(oc.state[c]["step"]) == 2.0 and oc.state[c]["step"].is_cuda and "ticket" not in oc.state[c]


def test_adam_fused_into_the_backward_is_lambd_adam_bit_for_bit():
    """LambdAdam(fused_into_backward=layer): the workgroup that finishes the backward's dot product applies the update itself
    (dmel_plan_attach_adam) -- the same arithmetic as dmel_adam_step on the gradient it has just written, so lambd, both moments and
    the step count follow the one-launch optimizer bit for bit, eagerly and replayed from a HIP graph; step() launches nothing."""
    from dmel_amd import LambdAdam, MelSpectrogramLayer
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)

    def mk():
        return MelSpectrogramLayer(torch.tensor(float(case["lambd"])), n_mels=case["n_mels"], n_points=case["L"], sample_rate=case["sr"],
                                   hop_length=case["hop"], device=DEV, optimized=True, log=True).to(DEV)

    la, lb = mk(), mk()
    la(x); lb(x)                                               # the plans exist from the first forward on
    oa = LambdAdam([la.lambd], lr=0.05, weight_decay=0.01, fused_into_backward=la)
    ob = LambdAdam([lb.lambd], lr=0.05, weight_decay=0.01)

    def step(lay, o):
        o.zero_grad(set_to_none=True)
        (lay(x) * g).sum().backward()
        o.step()

    hist = []
    for _ in range(3):
        step(la, oa); step(lb, ob)
        hist.append((la.lambd.detach().clone(), lb.lambd.detach().clone()))
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step(la, oa)
    for _ in range(10):
        gr.replay()
        step(lb, ob)
        hist.append((la.lambd.detach().clone(), lb.lambd.detach().clone()))
    torch.cuda.synchronize()
    for a, b in hist:
        assert torch.equal(a, b), (float(a), float(b))
    assert float(la.lambd.detach()) != float(case["lambd"])
    for key in ("exp_avg", "exp_avg_sq", "step"):
        assert torch.equal(oa.state[la.lambd][key], ob.state[lb.lambd][key]), key
    # the gradient is still delivered (AccumulateGrad runs behind the op), and detaching gives the plain backward back
    assert la.lambd.grad is not None and torch.isfinite(la.lambd.grad).all()
    oa.detach()
    before = la.lambd.detach().clone()
    (la(x) * g).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(la.lambd.detach(), before)
    with pytest.raises(ValueError):
        LambdAdam([lb.lambd], lr=0.05, fused_into_backward=la)           # not that layer's parameter


def test_fused_adam_is_attached_only_between_zero_grad_and_step():
    """the plan used to carry raw pointers to lambd and its Adam state from the first zero_grad() until an explicit
    detach().  Now: a backward outside the zero_grad() ... step() bracket (an evaluation pass) is a plain backward; an optimizer that is
    dropped inside the bracket takes its attachment with it; the plan references the state tensors while it carries their addresses."""
    import gc
    from dmel_amd import LambdAdam, MelSpectrogramLayer
    case = C.BY_NAME["g1_c1"]
    x = torch.from_numpy(C.make_input(case)).to(DEV)
    g = torch.from_numpy(C.make_cotangent(case)).to(DEV)
    lay = MelSpectrogramLayer(torch.tensor(float(case["lambd"])), n_mels=case["n_mels"], n_points=case["L"], sample_rate=case["sr"],
                              hop_length=case["hop"], device=DEV, optimized=True, log=True).to(DEV)
    lay(x)
    plan = next(iter(lay._plans.values()))
    opt = LambdAdam([lay.lambd], lr=0.05, fused_into_backward=lay)
    opt.zero_grad()
    assert plan._adam_keep is not None and plan._adam_keep[0] is lay.lambd
    (lay(x) * g).sum().backward()
    opt.step()
    torch.cuda.synchronize()
    after_step = lay.lambd.detach().clone()
    assert float(after_step) != float(case["lambd"]) and plan._adam_keep is None
    # an evaluation backward behind the step: lambd stays, the gradient is delivered
    lay.lambd.grad = None
    (lay(x) * g).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(lay.lambd.detach(), after_step) and lay.lambd.grad is not None
    with pytest.raises(RuntimeError):
        opt.step()                                               # no zero_grad() in front of that backward: nothing was applied, and step() says so
    # dropped inside the bracket: the finalizer detaches
    opt.zero_grad()
    assert plan._adam_keep is not None
    del opt
    gc.collect()
    assert plan._adam_keep is None
    (lay(x) * g).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(lay.lambd.detach(), after_step)
    # another optimizer takes over
    opt2 = torch.optim.Adam([lay.lambd], lr=0.05, fused=True, capturable=True)
    opt2.zero_grad()
    (lay(x) * g).sum().backward()
    opt2.step()
    torch.cuda.synchronize()
    assert not torch.equal(lay.lambd.detach(), after_step)
