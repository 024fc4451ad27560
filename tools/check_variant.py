#!/usr/bin/env python3
"""Quick correctness check of a library variant (DMEL_LIB) against the oracle on g2_c2 (n_fft 1024) and g3_c3 (2048): max rel
error of mel and of the tangent, both kernels modes.  For A/B work on the GPU box before the full test suite."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import dmel_amd
from dmel_amd import capi
import cases as C
from oracle import dmel_oracle as O
for name in sys.argv[1:] or ["g2_c2", "g1_c1"]:
    case = C.BY_NAME[name]
    x_np = C.make_input(case).astype(np.float32)
    x = torch.from_numpy(x_np).cuda()
    plan = capi.Plan(case["L"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"], case["normalize_window"])
    out = torch.empty(C.out_shape(case), device="cuda"); tan = torch.empty_like(out)
    s = torch.cuda.current_stream().cuda_stream
    plan.forward(x.data_ptr(), case["B"], case["lambd"], out.data_ptr(), tan.data_ptr(), True, 1e-10, s)
    out2 = torch.empty_like(out)
    plan.forward(x.data_ptr(), case["B"], case["lambd"], out2.data_ptr(), None, True, 1e-10, s)
    torch.cuda.synchronize()
    o_ref, t_ref = O.forward(x_np, case["lambd"], case["hop"], case["n_mels"], case["sr"], case["f_min"], case["f_max"], case["normalize_window"], apply_log=True)
    eo = float(np.abs(out.cpu().numpy() - o_ref).max()); ei = float(np.abs(out2.cpu().numpy() - o_ref).max())
    et = float(np.abs(tan.cpu().numpy() - t_ref).max() / np.abs(t_ref).max())
    print(os.path.basename(os.environ.get("DMEL_LIB", "")), name, f"log-mel abs err train {eo:.2e} infer {ei:.2e}  tangent rel {et:.2e}", "OK" if max(eo, ei, et) < 1e-4 else "FAIL")
