#!/usr/bin/env python3
"""SubbandGSCRLS micro benchmark: B utterances x 8 ch x T frames, M = 256 (129 bins); reports ms and the algorithmic rate
((C + 1) x 8 bytes per (frame, bin): the snapshot in, the output out -- the adaptation state stays on the device)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "distantspeechrecognition-mirror_amd"))
import torch
import dsr._capi as dsr
from tests import synth
ap = argparse.ArgumentParser(); ap.add_argument("--utts", type=int, default=256); ap.add_argument("--frames", type=int, default=1257); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dsr.load(); dev = torch.device("cuda:0")
M, Cn = 256, 8; F = M // 2 + 1
mp = synth.linear_array(Cn); d = dsr.calcDelaysPolar2(np.float32(0.5), np.float32(1.57), mp)
bf = dsr.Beamformer(M, Cn); bf.calcGSCWeights(16000.0, d); bf.select("gsc"); bf.rlsConfig(0.9, 0.001); bf.initPrecisionMatrix(0.01)
X = torch.view_as_complex(torch.randn((a.utts, Cn, a.frames, F, 2), device=dev))
Y, wa = bf.gsc_rls(X); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps): Y, wa = bf.gsc_rls(X)
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / a.reps
print("gsc_rls: %.2f ms for %d utt x %d frames x %d bins; %.1f GB/s algorithmic ((C+1) x 8 B per frame-bin), %.2f us per (utt, frame)" %
      (ms, a.utts, a.frames, F, a.utts * a.frames * F * (Cn + 1) * 8 / ms / 1e6, ms * 1e3 / (a.utts * a.frames)))
