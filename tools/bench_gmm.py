#!/usr/bin/env python3
"""GMM micro benchmark (BASELINE config 3): T frames x 39-dim against 256 codebooks x 16 Gaussians (G = 4096).
Reports TFLOP/s on the algorithmic 4*D*G flop per frame for the exact VALU kernel (mode 0) and the MFMA kernel (mode 2)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "distantspeechrecognition-mirror_amd"))
import torch
import dsr._capi as dsr
from tests import synth
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=1 << 20); ap.add_argument("--K", type=int, default=256); ap.add_argument("--R", type=int, default=16)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--no-argmin", action="store_true"); ap.add_argument("--modes", type=str, default="0,2"); a = ap.parse_args()
dsr.load(); dev = torch.device("cuda:0")
m = synth.gmm_model(a.K, a.R, 39, seed=12); gm = dsr.Gmm(**m)
g = torch.Generator(device=dev); g.manual_seed(11)
x = torch.randn((a.frames, 39), generator=g, device=dev)
res = {}
for mode in [int(v) for v in a.modes.split(',')]:
    for _ in range(2): sc, am = gm.score(x, mode=mode, want_argmin=not a.no_argmin)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): sc, am = gm.score(x, mode=mode, want_argmin=not a.no_argmin)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / a.reps
    res[mode] = (sc, am)
    print("mode %d: %.3f ms  %.1f TFLOP/s algorithmic (4*39*%d flop/frame)" % (mode, ms, a.frames * 4.0 * 39 * a.K * a.R / ms / 1e9, a.K * a.R), flush=True)
if len(res) < 2 or a.no_argmin: sys.exit(0)
d = (res[0][0] - res[2][0]).abs(); rel = (d / res[0][0].abs().clamp_min(1.0)).max().item()
agree = (res[0][1] == res[2][1]).float().mean().item()
print("mode 2 vs mode 0: max rel score diff %.3g, argmin agreement %.6f" % (rel, agree))
