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


# ---- MelPANNsNet / Cnn6 --------------------------------------------------------------------------
def test_panns_state_dict_contract_matches_reference():
    """keys + shapes captured from the reference's MelPANNsNet (models.py:138-166, panns.py:135-202)."""
    import cases as C
    from dmel_amd import panns
    cfg = C.PANNS_CFG
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "panns_state_keys.json")))
    net = panns.MelPANNsNet(cfg["n_classes"], torch.tensor(cfg["lambd"]), "cpu", cfg["n_mels"], cfg["sr"], cfg["L"],
                            hop_length=cfg["hop"], optimized=True, energy_normalize=True)
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == gold
    assert [n for n, _ in net.named_parameters()][0] == "spectrogram_layer.lambd"


def test_cnn6_matches_reference_outputs_on_cpu():
    """The CNN half (stock torch ops) fed with the log-mel the reference's own run produced (g9_panns.npz: s_log) returns
    the reference's clipwise outputs: same weights through cases.fill_state, eval mode."""
    import numpy as np
    import cases as C
    from dmel_amd import panns
    cfg = C.PANNS_CFG
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_panns.npz"))
    net = panns.MelPANNsNet(cfg["n_classes"], torch.tensor(cfg["lambd"]), "cpu", cfg["n_mels"], cfg["sr"], cfg["L"],
                            hop_length=cfg["hop"], optimized=True, energy_normalize=True)
    C.fill_state(net, seed=cfg["seed"])
    net.eval()
    with torch.no_grad():
        y = net.spectrogram_model(torch.from_numpy(gold["s_log"]).transpose(2, 3))
    assert y.shape == (cfg["B"], cfg["n_classes"])
    assert float((y - torch.from_numpy(gold["clipwise_log"])).abs().max()) <= 1e-6


def test_iid_axis_mask_zeroes_one_band_per_example():
    from dmel_amd.panns import _IidAxisMask
    torch.manual_seed(0)
    x = torch.ones(64, 1, 50, 30)
    for axis, param in ((2, 8), (3, 64)):
        y = _IidAxisMask(param, axis)(x)
        other = 3 if axis == 2 else 2
        prof = y[:, 0].amin(dim=other - 1)                     # (B, size along axis): 0 inside the band
        assert ((prof == 0) | (prof == 1)).all()
        widths = (prof == 0).sum(dim=1)
        assert int(widths.max()) < min(param, x.shape[axis]) + 1 and int(widths.max()) > 0
        for row in prof:                                       # the zeros are contiguous
            idx = torch.nonzero(row == 0).flatten()
            assert idx.numel() == 0 or int(idx[-1] - idx[0]) + 1 == idx.numel()
        assert not torch.equal(prof[0], prof[1]) or not torch.equal(prof[1], prof[2])     # iid across examples


def test_load_cnn6_checkpoint_remaps_keys(tmp_path):
    """utils.py:15-36: every key of checkpoint['model'] gets the prefix 'spectrogram_model.'; strict=False."""
    import pytest
    from dmel_amd import panns
    donor = panns.Cnn6(527, 64)
    state = {k: v.clone() for k, v in donor.state_dict().items() if not k.startswith("fc_esc50")}
    state["fc_audioset.weight"] = torch.zeros(527, 512)          # the pretrained head has no counterpart
    state["fc_audioset.bias"] = torch.zeros(527)
    path = tmp_path / "Cnn6.pth"
    torch.save({"model": state}, path)
    net = panns.MelPANNsNet(50, torch.tensor(40.0), "cpu", 64, 8000, 8000, hop_length=80, optimized=True)
    res = panns.load_cnn6_checkpoint(net, str(path))
    assert sorted(res.unexpected_keys) == ["spectrogram_model.fc_audioset.bias", "spectrogram_model.fc_audioset.weight"]
    assert sorted(res.missing_keys) == ["spectrogram_layer.lambd", "spectrogram_model.fc_esc50.bias", "spectrogram_model.fc_esc50.weight"]
    assert torch.equal(net.spectrogram_model.conv_block3.conv1.weight, donor.conv_block3.conv1.weight)
    with pytest.raises(FileNotFoundError):
        panns.load_cnn6_checkpoint(net, str(tmp_path / "missing.pth"))


# ---- f1: the reference's own logits (tests/golden/g11_nets.npz, make_golden.run_nets) ----------------------------------
def test_net_heads_reproduce_reference_logits_on_cpu(monkeypatch):
    """The heads of MelConvNet / MelLinearNet / MelMlpNet (stock torch ops on CPU) fed with the `s` the reference's own run produced give the
    reference's logits: same closed-form weights (cases.fill_state), dropout replaced by the identity as in the capture."""
    import numpy as np
    import torch.nn.functional as F
    import cases as C
    from dmel_amd import nets
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_nets.npz"))
    monkeypatch.setattr(F, "dropout", lambda x, *a, **k: x)
    case = C.BY_NAME["g1_c1"]
    for key, cls, en in (("conv_g1_log", "MelConvNet", True), ("conv_g1_lin", "MelConvNet", False), ("linear_g1_log", "MelLinearNet", True),
                         ("mlp_g1_log", "MelMlpNet", True), ("mlp_g1_lin", "MelMlpNet", False)):
        net = getattr(nets, cls)(C.NET_CLASSES, torch.tensor(float(case["lambd"])), "cpu", case["n_mels"], case["sr"], case["L"],
                                 hop_length=case["hop"], optimized=True, energy_normalize=en)
        C.fill_state(net, seed=C.NET_SEED)
        s = torch.from_numpy(gold[key + "_s"])
        net._features = lambda x, s=s: s                      # the front end needs the GPU: feed the reference's spectrogram
        with torch.no_grad():
            logits, s_out = net(torch.zeros(case["B"], case["L"]))
        ref = torch.from_numpy(gold[key + "_logits"])
        assert logits.shape == ref.shape and s_out is s
        assert float((logits - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6, key
