import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "distantspeechrecognition-mirror_amd"))
import torch
import dsr._capi as dsr
dsr.load(); dev = torch.device("cuda:0")
hg = np.load(os.path.join(ROOT, "tests", "golden", "proto_M256-m4-r1.npy")); h, g = hg
N = 160000
torch.manual_seed(0)
ana = dsr.FilterBank(h, 256, 4, 1, False, 0)
for U in (33, 128, 300):
    x = (torch.randn((U, 8, N), device=dev) * 3000)
    X = ana.analysis(x)
    worst = 0.0; nbad = 0
    for u in range(U):
        Xs = ana.analysis(x[u:u + 1].contiguous())
        d = (X[u] - Xs[0]).abs().max().item()
        worst = max(worst, d); nbad += d > 1.0
    print("U=%d blocks=%d: worst diff %.3g, utterances off: %d" % (U, U * 8, worst, nbad), flush=True)
