import os, sys, numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "distantspeechrecognition-mirror_amd"))
import torch
import dsr._capi as dsr
from oracle import oracle
dsr.load(); dev = torch.device("cuda:0")
hg = np.load(os.path.join(ROOT, "tests", "golden", "proto_M256-m4-r1.npy")); h, g = hg
for (U, N) in ((2, 160000), (2, 20000), (2, 20003)):
    x = (torch.randn((U, 8, N), device=dev) * 3000)
    ana = dsr.FilterBank(h, 256, 4, 1, False, 0)
    X = ana.analysis(x).cpu().numpy()
    xr = x[1, 5].cpu().numpy()
    want = oracle.analysis_bank(xr, h, 256, 4, 1, 0)[:, :129]
    got = X[1, 5]
    err = np.abs(got - want).max(axis=1) / np.sqrt(np.mean(np.abs(want) ** 2))
    bad = np.nonzero(err > 1e-3)[0]
    print(U, N, got.shape, want.shape, "max rel err %.3g" % err.max(), "bad frames:", bad[:10], len(bad))
