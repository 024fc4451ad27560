"""The caller nets keep the reference's checkpoint contract: state_dict keys and shapes
captured from the reference's models.py:58-136 (tests/golden/net_state_keys.json)."""
import json
import os

import torch


def test_state_dict_contract_matches_reference():
    from dmel_amd import nets
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "net_state_keys.json")))
    for name, keys in gold.items():
        net = getattr(nets, name)(10, torch.tensor(8000 * 0.035 / 6), "cpu", 64, 8000, 8000, hop_length=80, optimized=True,
                                  energy_normalize=True)
        got = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert got == keys, name
        assert net.size == (64, 101) and net.energy_normalize is True


def test_two_lr_groups():
    from dmel_amd import nets
    net = nets.MelLinearNet(3, torch.tensor(10.0), "cpu", 8, 8000, 800, hop_length=80, optimized=True)
    opt = nets.make_optimizer(net, lr_model=1e-4, lr_tf=1.0, name="sgd")
    lrs = {id(g["params"][0]): g["lr"] for g in opt.param_groups}
    assert lrs[id(net.spectrogram_layer.lambd)] == 1.0
    assert all(v == 1e-4 for k, v in lrs.items() if k != id(net.spectrogram_layer.lambd))
