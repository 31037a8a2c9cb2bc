#!/usr/bin/env python3
"""How even are the eight XCD queues of the time-sliced decode?  Placements per utterance (the decoder's own count) summed by u mod 8, and what a dealing by
the first segment's pace would give if an utterance's share of the work in its first segment predicted its share of the whole (greedy: next utterance to the least loaded queue)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "distantspeechrecognition-mirror_amd"))
import torch
import dsr._capi as dsr
from tests import synth
dsr.load(); dev = torch.device("cuda:0")
U, T = 1000, 1000
arcs, fin = synth.random_wfst(50000, 1024, seed=21, outdeg=4, eps_frac=0.1, out_frac=0.05, nWords=5000, nFinal=50)
g = dsr.Wfst()
for x in arcs: g.add_arc(*x)
for s, c in fin: g.add_final(s, c)
gen = torch.Generator(device=dev); gen.manual_seed(5)
gm = dsr.Gmm(**synth.gmm_model(1024, 4, 39, seed=12))
f = torch.randn((U, T + 16, 39), generator=gen, device=dev)
f = torch.nn.functional.avg_pool1d(f.transpose(1, 2), 9, 1).transpose(1, 2)[:, :T].contiguous() * 3.0
sc = gm.score(f.reshape(-1, 39), mode=0, want_argmin=False)[0].reshape(U, T, 1024)
dec = dsr.Decoder(beam=53.79, lmScale=12.0, maxActive=65536); dec.set(g)
full = np.array([o["placements"] for o in dec.decode_batch(sc, maxPath=16)], float)
first = np.array([o["placements"] for o in dec.decode_batch(sc[:, :125].contiguous(), maxPath=16)], float)
stat = np.array([full[q::8].sum() for q in range(8)])
print("static u mod 8: queue loads max/mean %.4f  (std/mean of an utterance %.3f)" % (stat.max() / stat.mean(), full.std() / full.mean()))
print("correlation of an utterance's first 125 frames with its total: %.3f" % np.corrcoef(first, full)[0, 1])
load = np.zeros(8); est = np.zeros(8)
for u in range(U):                      # utterances in order, each to the queue that is least loaded by the FIRST-segment measure (what pacing by the first segment does)
    q = int(np.argmin(est)); est[q] += first[u]; load[q] += full[u]
print("dealt by first-segment pace: queue loads max/mean %.4f" % (load.max() / load.mean()))
