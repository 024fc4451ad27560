"""The batched torch restatement (cpu_baseline leg) agrees with the C oracle and the reference goldens."""
import numpy as np
import pytest
import torch

import cases as C
from oracle import dmel_oracle as O
from oracle import torch_restatement as TR


@pytest.mark.parametrize("name", ["g1_c1", "g2_c2", "g6_normwin", "g6_neglambd", "g6_n256_ragged", "g5_n128"])
def test_restatement_matches_golden_and_oracle(name):
    case = C.BY_NAME[name]
    gold = C.load(case)
    x = torch.from_numpy(C.make_input(case).astype(np.float32))
    g = torch.from_numpy(C.make_cotangent(case))
    lam = torch.tensor(float(case["lambd"]), requires_grad=True)
    mel = TR.forward(x, lam, case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"], case["normalize_window"])
    assert TR.n_fft_of(lam) == int(gold["n_fft"])
    np.testing.assert_allclose(mel.detach().numpy(), gold["mel"], rtol=1e-4, atol=1e-6 * float(gold["mel"].max()))
    (dl,) = torch.autograd.grad((torch.log(mel + 1e-10) * g).sum(), lam)
    assert abs(float(dl) - float(gold["dlam_log"])) <= 1e-4 * abs(float(gold["dlam_log"])) + 1e-7
    o, t = O.forward(x.numpy(), case["lambd"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"],
                     case["normalize_window"], apply_log=True)
    assert abs(O.backward(g.numpy(), t) - float(dl)) <= 1e-4 * abs(float(dl)) + 1e-7
