#!/usr/bin/env python3
"""Regenerates tests/golden/* from the reference's DATA files (run in the authoring container only).

Inputs (read-only, data not code):
  /root/reference/btk/tools/filterbank/Headset1.wav                 mono 16 kHz PCM16
  /root/reference/btk/examples/prototypes/Nyquist/*.m               Nyquist(M) analysis+synthesis prototypes
  /root/reference/asr/test/Lexicon.txt                              4-symbol lexicon
Outputs:
  Headset1_16k_s16.npy      int16 samples of the wav (SampleFeature::read with norm==0 keeps int16 scale)
  proto_M*.npy              float64 [2][m*M]: row 0 analysis h, row 1 synthesis g
                            (text layout read by btk/src/superdirectiveBeamformer.cc:23-72)
  Lexicon.txt               copied verbatim (data)
  linpack_csvdc.npz         seeded 8x8 / 4x4 complex matrices and the singular values + pseudo-inverse
                            that the reference's own LINPACK csvdc (oracle/_ref) produces for them
"""
import os, sys, wave, shutil
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "..", ".."))

w = wave.open(f"{REF}/btk/tools/filterbank/Headset1.wav")
assert w.getnchannels() == 1 and w.getsampwidth() == 2 and w.getframerate() == 16000
x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
np.save(f"{HERE}/Headset1_16k_s16.npy", x)

for name in ("M=256-m=4-r=1", "M=512-m=2-r=2", "M=512-m=2-r=3"):
    v = np.array(open(f"{REF}/btk/examples/prototypes/Nyquist/{name}.m").read().split(), dtype=np.float64)
    np.save(f"{HERE}/proto_{name.replace('=', '')}.npy", v.reshape(2, -1))

shutil.copy(f"{REF}/asr/test/Lexicon.txt", f"{HERE}/Lexicon.txt")

# --- LINPACK csvdc goldens through oracle/_ref (the reference's own sources) ---
import ctypes as C
from oracle import oracle as O
O.build_ref()
L = O.ref_linpack()
rng = np.random.default_rng(20240607)
mats, svals, pinvs = [], [], []
for n in (8, 8, 8, 4, 2):
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    if len(mats) == 1:   # a diffuse-noise-like Hermitian, badly conditioned matrix
        d = np.abs(np.subtract.outer(np.arange(n), np.arange(n))) * 41.0
        A = np.sinc(2 * 16000 * 3 / (256 * 343740.0) * d) / 1.01
        np.fill_diagonal(A, 1.0); A = A.astype(np.complex128)
    a = np.asfortranarray(A.astype(np.complex64))
    s = np.zeros(2 * n, np.complex64); e = np.zeros(2 * n, np.complex64)
    u = np.zeros((n, n), np.complex64, order="F"); v = np.zeros((n, n), np.complex64, order="F")
    acopy = a.copy(order="F")
    info = L.ref_csvdc(acopy.ctypes.data_as(C.c_void_p), n, n, n, s.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                       u.ctypes.data_as(C.c_void_p), n, v.ctypes.data_as(C.c_void_p), n, 11)
    assert info == 0
    # pinv assembled exactly as beamformer.cc:284-302
    sinv = np.where(np.abs(s[:n]) < 1e-8, 0, 1.0 / s[:n]).astype(np.complex64)
    P = np.zeros((n, n), np.complex64)
    for i in range(n):
        for j in range(n):
            acc = np.complex64(0)
            for k in range(n):
                acc = np.complex64(acc + v[j, k] * sinv[k] * np.conj(u[i, k]))
            P[j, i] = acc
    mats.append(A); svals.append(s[:n].copy()); pinvs.append(P)
np.savez(f"{HERE}/linpack_csvdc.npz", **{f"A{i}": m for i, m in enumerate(mats)},
         **{f"s{i}": m for i, m in enumerate(svals)}, **{f"P{i}": m for i, m in enumerate(pinvs)})
print("fixtures written to", HERE)
