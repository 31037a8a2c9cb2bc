"""Seeded synthetic inputs shared by tests, smoke() and bench.py (shapes of BASELINE.md / SURVEY.md 8d)."""
import numpy as np


def linear_array(C=8, spacing_mm=41.0):
    mp = np.zeros((C, 3), np.float64)
    mp[:, 0] = (np.arange(C) - (C - 1) / 2.0) * spacing_mm
    return mp


def array_signal(nsamp, C=8, seed=7, az_deg=30.0, fs=16000.0, sigma=3000.0, noise=300.0):
    """White Gaussian source, low-passed, far-field fractional delays on a linear array + sensor noise."""
    rng = np.random.default_rng(seed)
    src = rng.standard_normal(nsamp + 64) * sigma
    k = np.hanning(9); k /= k.sum()
    src = np.convolve(src, k, mode="same")
    mp = linear_array(C)
    tau = mp[:, 0] * np.cos(np.deg2rad(az_deg)) / 343740.0 * fs          # samples
    S = np.fft.rfft(src)
    f = np.arange(len(S)) / float(len(src))
    out = np.zeros((C, nsamp), np.float32)
    for c in range(C):
        d = np.fft.irfft(S * np.exp(-2j * np.pi * f * tau[c]), len(src))
        out[c] = (d[32:32 + nsamp] + rng.standard_normal(nsamp) * noise).astype(np.float32)
    return out


def gmm_model(K, R, D, seed=12):
    rng = np.random.default_rng(seed)
    G = K * R
    mean = rng.standard_normal((G, D)).astype(np.float32)
    ivar = (1.0 / rng.uniform(0.5, 2.0, (G, D))).astype(np.float32)
    det = (-np.log(ivar.astype(np.float64)).sum(1)).astype(np.float32)
    w = rng.dirichlet(np.ones(R), K).reshape(-1)
    val = (-np.log(w)).astype(np.float32)
    return dict(refN=np.full(K, R, np.int32), mean=mean, ivar=ivar, det=det, val=val)


def random_wfst(S, nDist, seed=21, outdeg=4, eps_frac=0.1, out_frac=0.05, nWords=5000, nFinal=50, ties=False):
    """Arc list [(s1,s2,in,out,cost)] + finals [(s,cost)].  Epsilon arcs only go 'forward' (dst>src) so there is
    no epsilon cycle; state 0 is the source of the first arc (= initial state)."""
    rng = np.random.default_rng(seed)
    arcs = []
    for s in range(S):
        n = max(1, int(rng.geometric(1.0 / outdeg)))
        n = min(n, 4 * outdeg)
        for _ in range(n):
            eps = rng.random() < eps_frac and s < S - 1
            dst = int(rng.integers(s + 1, S)) if eps else int(rng.integers(0, S))
            i = 0 if eps else int(rng.integers(1, nDist + 1))
            o = int(rng.integers(1, nWords + 1)) if rng.random() < out_frac else 0
            cost = float(np.float32(rng.integers(0, 6))) if ties else float(np.float32(rng.uniform(0, 5)))
            arcs.append((s, dst, i, o, cost))
    fin = [(int(s), float(np.float32(rng.uniform(0, 1)))) for s in rng.choice(np.arange(1, S), min(nFinal, S - 1), replace=False)]
    return arcs, fin
