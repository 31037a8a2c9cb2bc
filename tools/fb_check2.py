import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "distantspeechrecognition-mirror_amd"))
import torch
import dsr._capi as dsr
dsr.load(); dev = torch.device("cuda:0")
hg = np.load(os.path.join(ROOT, "tests", "golden", "proto_M256-m4-r1.npy")); h, g = hg
U, N = 1000, 160000
torch.manual_seed(0)
x = (torch.randn((U, 8, N), device=dev) * 3000)
ana = dsr.FilterBank(h, 256, 4, 1, False, 0)
X = ana.analysis(x)
for u in (0, 1, 499, 998, 999):
    Xs = ana.analysis(x[u:u + 1].contiguous())
    d = (X[u] - Xs[0]).abs().max().item(); ref = Xs.abs().max().item()
    print("utt", u, "max diff big-batch vs single", d, "scale", ref, flush=True)
os.environ["DSR_FB_WAVE"] = "1"
Xw = ana.analysis(x[:4].contiguous())
print("vs wave kernel:", (X[:4] - Xw).abs().max().item(), Xw.abs().max().item())
