import os, sys, subprocess, numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch, dmel_amd
    from dmel_amd import capi, synth
    B,L,sr,lam,hop,M = (8,16000,16000,128.0,512,128) if os.environ.get("CMP_CFG","c2")=="c2" else ((2,160000,16000,256.0,512,128) if os.environ["CMP_CFG"]=="c3" else (2,40000,8000,400.0,80,64))
    T = L//hop+1
    x = torch.from_numpy(synth.waveforms(B, L, seed=0)).cuda()
    out = torch.zeros((B,1,M,T), device="cuda"); tan = torch.zeros_like(out)
    plan = capi.Plan(L, hop, M, sr, max_batch=B)
    plan.forward(x.data_ptr(), B, lam, out.data_ptr(), tan.data_ptr(), False, 1e-10, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.save(sys.argv[1], np.stack([out.cpu().numpy(), tan.cpu().numpy()]))
else:
    e = dict(os.environ); e["DMEL_WLC"]="0"
    subprocess.check_call([sys.executable, __file__, "/tmp/o_old.npy"], env=e)
    subprocess.check_call([sys.executable, __file__, "/tmp/o_new.npy"])
    a, b = np.load("/tmp/o_old.npy"), np.load("/tmp/o_new.npy")
    d = np.abs(a-b) > 1e-4*np.abs(a).max()
    print("mismatches", d.sum(), "of", d.size)
    idx = np.argwhere(d)
    print("planes", np.unique(idx[:,0]), "clips", np.unique(idx[:,1]), "mels", np.unique(idx[:,3]), "frames", np.unique(idx[:,4]))
    print(idx[:10]); print(a[d][:10], b[d][:10])
    if len(sys.argv) == 1:
        r = np.abs(a - b) / (np.abs(a) + 1e-6 * np.abs(a).max())
        print("median rel err by frame:", np.round(np.median(r[0], axis=(0, 1, 2)), 4))
        print("median rel err by mel (every 8th):", np.round(np.median(r[0], axis=(0, 1, 3))[::8], 4))
        print("median rel err by clip:", np.round(np.median(r[0], axis=(1, 2, 3)), 4))
