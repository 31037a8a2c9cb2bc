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
  linpack_csvdc.npz         seeded complex matrices (2x2 .. 64x64: random, diffuse-field coherence matrices of 8/16/64-microphone
                            arrays, rank deficient, zero) and the singular values, U, V and pseudo-inverse that the reference's own
                            LINPACK csvdc (oracle/_ref) produces for them; six rectangular matrices (2x8 .. 8x2, 2x64) the same way
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
from oracle import oracle as O
O.build_ref()
rng = np.random.default_rng(20240607)


def diffuse(n, f, pitch, M=256, fs=16000.0, c=343740.0, planar=False, mu=0.01):
    """coherence matrix of a diffuse field as setDiffuseNoiseModel + divideAllNonDiagonalElements build it (beamformer.cc:2486-2553)"""
    if planar:
        k = int(round(np.sqrt(n))); pos = np.array([(i * pitch, j * pitch) for i in range(k) for j in range(k)], float)
    else:
        pos = np.stack([np.arange(n) * pitch, np.zeros(n)], 1)
    d = np.sqrt(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1))
    A = np.sinc(2 * fs * f / (M * c) * d) / (1.0 + mu)
    np.fill_diagonal(A, 1.0)
    return A.astype(np.complex128)


cases = [("rand8", rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))),
         ("diffuse8_f3", diffuse(8, 3, 41.0)),
         ("rand8b", rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))),
         ("rand4", rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))),
         ("rand2", rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))),
         # round 2: the array sizes of BASELINE configs[4] (64 channels) and what lies between
         ("rand16", rng.standard_normal((16, 16)) + 1j * rng.standard_normal((16, 16))),
         ("diffuse16_f1", diffuse(16, 1, 20.0)),
         ("diffuse64_f1", diffuse(64, 1, 20.0, planar=True)),
         ("diffuse64_f40", diffuse(64, 40, 20.0, planar=True)),
         ("diffuse64_f128", diffuse(64, 128, 20.0, planar=True)),
         ("rand64", rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))),
         ("rank6of8", None), ("hermitian33", None), ("zero3", np.zeros((3, 3), complex))]
B = rng.standard_normal((8, 6)) + 1j * rng.standard_normal((8, 6)); cases[11] = ("rank6of8", B @ B.conj().T)
H = rng.standard_normal((33, 33)) + 1j * rng.standard_normal((33, 33)); cases[12] = ("hermitian33", H @ H.conj().T / 33)
out = {}
for i, (name, A) in enumerate(cases):
    n = A.shape[0]
    info, s, u, v = O.ref_csvdc(A)
    # pinv assembled exactly as beamformer.cc:284-302 (complex<float> accumulation, k ascending)
    thr = np.float32(1e-8)
    sinv = np.array([np.complex64(0) if np.abs(x) < thr else np.complex64(1) / x for x in s], np.complex64)
    P = np.zeros((n, n), np.complex64)
    for a_ in range(n):
        for b_ in range(n):
            acc = np.complex64(0)
            for k in range(n):
                acc = np.complex64(acc + np.complex64(np.complex64(v[b_, k] * sinv[k]) * np.conj(u[a_, k])))
            P[b_, a_] = acc
    out["A%d" % i] = A; out["s%d" % i] = s; out["P%d" % i] = P; out["U%d" % i] = np.ascontiguousarray(u); out["V%d" % i] = np.ascontiguousarray(v)
    out["info%d" % i] = np.int32(info)
# rectangular matrices (round 2): scaling() of SubbandMMI hands an nSource x chanN demixing matrix to the pseudo-inverse (beamformer.cc:1862).  For rows <
# columns the shipped assembly loop (:283-297) runs over singular values and left vectors csvdc never produced; the golden pseudo-inverse sums the
# min(rows, cols) terms that exist (same complex<float> accumulation, k ascending).
rect = [("wide2x8", 2, 8), ("wide3x8", 3, 8), ("wide2x5", 2, 5), ("tall8x2", 8, 2), ("tall6x3", 6, 3), ("wide2x64", 2, 64)]
for i, (name, n, p_) in enumerate(rect):
    A = rng.standard_normal((n, p_)) + 1j * rng.standard_normal((n, p_))
    info, s, u, v = O.ref_csvdc(A)
    thr = np.float32(1e-7); K = min(n, p_)
    sinv = np.array([np.complex64(0) if np.abs(x) < thr else np.complex64(1) / x for x in s], np.complex64)
    P = np.zeros((p_, n), np.complex64)
    for a_ in range(n):
        for b_ in range(p_):
            acc = np.complex64(0)
            for k in range(K):
                acc = np.complex64(acc + np.complex64(np.complex64(v[b_, k] * sinv[k]) * np.conj(u[a_, k])))
            P[b_, a_] = acc
    out["RA%d" % i] = A; out["Rs%d" % i] = s; out["RP%d" % i] = P; out["RU%d" % i] = np.ascontiguousarray(u); out["RV%d" % i] = np.ascontiguousarray(v)
    out["Rinfo%d" % i] = np.int32(info)
out["rnames"] = np.array([c[0] for c in rect])
out["names"] = np.array([c[0] for c in cases])
np.savez_compressed(f"{HERE}/linpack_csvdc.npz", **out)
print("fixtures written to", HERE)
