#!/usr/bin/env python3
"""WPE micro benchmark: B utterances x N frames, M = 256 (129 bins), prediction taps lowerN..upperN, 2 iterations.
Reports ms and the algorithmic rate on 16 B per (channel, frame, bin): the snapshot in, the dereverberated snapshot out."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "distantspeechrecognition-mirror_amd"))
import torch
import dsr._capi as dsr
ap = argparse.ArgumentParser(); ap.add_argument("--utts", type=int, default=256); ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--chan", type=int, default=4); ap.add_argument("--lower", type=int, default=2); ap.add_argument("--upper", type=int, default=9); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dsr.load(); dev = torch.device("cuda:0")
M = 256; F = M // 2 + 1
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / a.reps
Y1 = torch.view_as_complex(torch.randn((a.utts, a.frames, F, 2), device=dev))
ms = timeit(lambda: dsr.wpe_single(Y1, M, a.lower, a.upper, 2))
print("wpe_single: %.2f ms for %d utt x %d frames x %d bins, %d taps; %.1f GB/s algorithmic" % (ms, a.utts, a.frames, F, a.upper - a.lower + 1, a.utts * a.frames * F * 16 / ms / 1e6), flush=True)
u2 = max(1, a.utts // 8)
Yc = torch.view_as_complex(torch.randn((u2, a.chan, a.frames, F, 2), device=dev))
ms = timeit(lambda: dsr.wpe_multi(Yc, M, a.lower, a.upper, 2))
print("wpe_multi : %.2f ms for %d utt x %d ch x %d frames x %d bins, %d stacked taps; %.1f GB/s algorithmic" % (ms, u2, a.chan, a.frames, F, a.chan * (a.upper - a.lower + 1), u2 * a.chan * a.frames * F * 16 / ms / 1e6))
