#!/usr/bin/env python3
"""G8: torchaudio.functional.melscale_fbanks tables for the six (n_freqs, n_mels, sample_rate) of the survey.

torchaudio (pinned 0.13.1, requirements.txt:8 of the reference; call site models.py:42-48) is in neither this image nor the reference tree, so
the committed ``g8_fbanks.npz`` was written by the STAND-IN of make_golden.py (torchaudio's published ``melscale_fbanks`` +
``_create_triangular_filterbank``, htk / norm=None, restated op for op in fp32 torch) and says so in its ``source`` entry: SURVEY row a6 stays
"formula-faithful, parity unpinned".  On any machine that HAS torchaudio this script pins it in one run:

    python tests/golden/make_g8_fbanks.py            # compares the real torchaudio with the committed tables (bit for bit) and, if they agree,
                                                     # rewrites g8_fbanks.npz with source = "torchaudio <version>"
    python tests/golden/make_g8_fbanks.py --standin  # regenerate from the stand-in (what this image can do)

tests/test_capi_host.py::test_filterbank_tables_g8 holds dmel_mel_fbanks_host (and the oracle) to these tables."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
SHAPES = [(257, 64, 16000), (513, 128, 16000), (1025, 128, 16000), (1025, 128, 44100), (65, 64, 8000), (2049, 64, 8000)]
PATH = os.path.join(HERE, "g8_fbanks.npz")


def tables(fn):
    out = {}
    for F, M, sr in SHAPES:
        fb = fn(n_freqs=F, f_min=0.0, f_max=sr // 2, n_mels=M, sample_rate=sr)          # models.py:42-48 with f_min = 0, f_max = sr // 2 (models.py:25)
        out[f"fb_{F}_{M}_{sr}"] = fb.detach().numpy().astype(np.float32)
    return out


def main(argv):
    from make_golden import _melscale_fbanks
    if "--standin" in argv:
        np.savez_compressed(PATH, source=np.array("stand-in (make_golden._melscale_fbanks): torchaudio absent"), **tables(_melscale_fbanks))
        print("wrote", PATH, "from the stand-in")
        return 0
    try:
        import torchaudio
    except ImportError:
        print("torchaudio is not importable here: nothing to pin (use --standin to regenerate the committed tables)")
        return 2
    real = tables(torchaudio.functional.melscale_fbanks)
    have = np.load(PATH)
    worst = 0.0
    for k, v in real.items():
        d = float(np.abs(v - have[k]).max())
        worst = max(worst, d)
        print(f"{k}: max |torchaudio - committed| = {d:.3e}  bit-identical: {np.array_equal(v, have[k])}")
    if worst == 0.0:
        np.savez_compressed(PATH, source=np.array(f"torchaudio {torchaudio.__version__} (bit-identical to the stand-in)"), **real)
        print("pinned: rewrote", PATH)
        return 0
    print("the stand-in and torchaudio DIFFER: a6 is not what the reference computes; fix dmel_mel_fbanks_host / the oracle against these tables")
    np.savez_compressed(PATH.replace(".npz", "_torchaudio.npz"), source=np.array(f"torchaudio {torchaudio.__version__}"), **real)
    return 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))
