#!/usr/bin/env python3
"""per-repetition timing of the analysis kernel (looking for launch-to-launch variation)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "distantspeechrecognition-mirror_amd"))
import torch
import dsr._capi as dsr
dsr.load(); dev = torch.device("cuda:0")
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hg = np.load(os.path.join(ROOT, "tests", "golden", "proto_M256-m4-r1.npy")); h, g = hg
x = torch.randn((U, 8, 160000), device=dev) * 3000
ana = dsr.FilterBank(h, 256, 4, 1, False, 0)
ns = torch.full((U,), 160000, dtype=torch.int32, device=dev)
X = ana.analysis(x, ns); torch.cuda.synchronize()
ts = []
for i in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); X = ana.analysis(x, ns); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("U=%d per-rep ms:" % U, " ".join("%.2f" % t for t in ts))
