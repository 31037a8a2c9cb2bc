"""CPU tests (-m "not gpu"): the oracle against everything the reference holds for this path.

The reference ships no golden outputs, so the oracle is pinned by
  * the perfect-reconstruction property of the reference's own Nyquist(M) prototypes on its own
    Headset1.wav (tests/golden/, copied data files) -- pins analysis/synthesis index conventions,
  * the reference's in-tree LINPACK csvdc (built from the reference sources into oracle/_ref, results
    committed as tests/golden/linpack_csvdc.npz) -- pins the complex<float> pseudo-inverse,
  * byte-level file-format round trips and closed-form properties.
"""
import os

import numpy as np
import pytest

from tests import synth
from tests.conftest import GOLDEN


# ------------------------------------------------------------------ filter banks
@pytest.mark.parametrize("pname,tol", [("M256-m4-r1", 3e-3), ("M512-m2-r2", 2e-5), ("M512-m2-r3", 2e-7)])
@pytest.mark.parametrize("dct", [1, 2])
def test_filterbank_perfect_reconstruction(oracle, headset, protos, pname, tol, dct):
    """btk/tools/filterbank/testNyquistFilterBankDesign.py:45-66: analysis -> synthesis, output scaled by D,
    reproduces the input with zero delay when the delays are compensated (types 1 and 2)."""
    M, m, r, h, g = protos[pname]
    D = M >> r
    x = headset[:24000]
    X = oracle.analysis_bank(x, h, M, m, r, dct)
    assert X.shape[0] == oracle.analysis_num_frames(len(x), M, m, r, dct)
    y = oracle.synthesis_bank(X, g, M, m, r, dct) * D
    n = min(len(x), len(y))
    e = y[2000:n - 2000] - x[2000:n - 2000]
    assert np.sqrt(np.mean(e ** 2)) / np.sqrt(np.mean(x ** 2)) < tol


def test_filterbank_frame_counts(oracle, protos):
    M, m, r, h, g = protos["M256-m4-r1"]
    # type 0: ceil(N/D) + 2m-1 ; type 1: + mR-1 ; type 2: - laN + mR-1   (modulated.cc:279-296, 461-516)
    assert oracle.analysis_num_frames(160000, 256, 4, 1, 0) == 1250 + 7
    assert oracle.analysis_num_frames(160001, 256, 4, 1, 0) == 1251 + 7
    assert oracle.analysis_num_frames(160000, 256, 4, 1, 1) == 1250 + 7
    assert oracle.analysis_num_frames(160000, 256, 4, 1, 2) == 1250 - 3 + 7
    assert oracle.analysis_num_frames(100, 256, 4, 1, 2) == 0         # fewer blocks than the look-ahead
    X = oracle.analysis_bank(np.ones(1000, np.float32), h, M, m, r, 0)
    assert np.abs(X[:, 1:128] - np.conj(X[:, :128:-1])).max() < 1e-9   # real input: Hermitian spectrum
    # synthesis emits T - processingDelay blocks
    y = oracle.synthesis_bank(X, g, M, m, r, 0)
    assert len(y) == (X.shape[0] - 7) * 128


def test_analysis_matches_closed_form(oracle, protos):
    """X_t[f] = sum_l h[l] x[n_t - l] e^{+2 pi j f l / M}, n_t = (t+1) D - 1."""
    M, m, r, h, g = protos["M512-m2-r2"]
    D = M >> r
    rng = np.random.default_rng(0)
    x = rng.standard_normal(3000).astype(np.float32)
    X = oracle.analysis_bank(x, h, M, m, r, 0)
    xp = np.concatenate([np.zeros(m * M), x.astype(np.float64), np.zeros(40 * D)])
    for t in (0, 3, 11, X.shape[0] - 1):
        nt = (t + 1) * D - 1 + m * M
        seg = xp[nt - np.arange(m * M)]
        u = (h * seg).reshape(m, M).sum(0)
        ref = np.fft.ifft(u) * M
        assert np.abs(X[t] - ref).max() < 1e-9 * (1 + np.abs(ref).max())


def test_normal_fft_bank(oracle):
    x = np.random.default_rng(1).standard_normal(2000).astype(np.float32)
    X = oracle.normal_fft_bank(x, 256, 1, winType=1)
    assert X.shape == (2000 // 128 + 1 + 1, 256)


# ------------------------------------------------------------------ beamformer
def _csvdc_cases():
    z = np.load(os.path.join(GOLDEN, "linpack_csvdc.npz"))
    return z, [str(n) for n in z["names"]]


def test_pseudoinverse_pinned_by_linpack(oracle):
    """The oracle's csvdc restatement (orc_svd.c) against the outputs of the reference's own csvdc (built from the reference sources by
    oracle/Makefile:_ref, outputs committed by tests/golden/make_fixtures.py): singular values, U, V and the pseudo-inverse assembled as
    beamformer.cc:284-302 -- BIT FOR BIT, on 2x2 .. 64x64 matrices incl. the diffuse-field coherence matrices of 8/16/64-microphone arrays."""
    z, names = _csvdc_cases()
    assert len(names) >= 14 and "diffuse64_f1" in names
    for i, name in enumerate(names):
        A = z["A%d" % i]
        info, s, u, v = oracle.csvdc(A)
        assert info == int(z["info%d" % i]), name
        assert np.array_equal(s.view(np.float32), z["s%d" % i].view(np.float32)), name
        assert np.array_equal(np.ascontiguousarray(u).view(np.float32), z["U%d" % i].view(np.float32)), name
        assert np.array_equal(np.ascontiguousarray(v).view(np.float32), z["V%d" % i].view(np.float32)), name
        Po, ok = oracle.pseudoinverse(A)
        assert np.array_equal(Po.astype(np.complex64).view(np.float32), z["P%d" % i].view(np.float32)), name
        assert ok == bool(np.all(np.abs(z["s%d" % i]) >= np.float32(1e-8)) and info == 0), name
        # and the routine is an SVD: singular values agree with LAPACK's on the same complex64 matrix
        sl = np.linalg.svd(A.astype(np.complex64).astype(np.complex128), compute_uv=False)
        assert np.abs(np.sort(s.real)[::-1] - sl).max() <= 2e-5 * max(sl[0], 1e-30), name


def test_rectangular_csvdc_pinned_by_linpack(oracle):
    """the same pin for rectangular matrices (2x8 .. 8x2, 2x64): what scaling() of SubbandMMI's nSource x chanN demixing matrix needs
    (beamformer.cc:1455-1490, 1862) -- singular values, U, V bit for bit; the pseudo-inverse with the min(rows, cols) terms csvdc provides"""
    z = np.load(os.path.join(GOLDEN, "linpack_csvdc.npz"))
    names = [str(n) for n in z["rnames"]]
    assert len(names) >= 6
    L = oracle.lib()
    for i, name in enumerate(names):
        A = z["RA%d" % i]; n, p = A.shape
        info, s, u, v = oracle.csvdc(A)
        assert info == int(z["Rinfo%d" % i]), name
        assert np.array_equal(s.view(np.float32), z["Rs%d" % i].view(np.float32)), name
        assert np.array_equal(np.ascontiguousarray(u).view(np.float32), z["RU%d" % i].view(np.float32)), name
        assert np.array_equal(np.ascontiguousarray(v).view(np.float32), z["RV%d" % i].view(np.float32)), name
        P = np.zeros((p, n), np.complex128)
        L.orc_pseudoinverse_mn(oracle._p(np.ascontiguousarray(A, np.complex128)), n, p, oracle._p(P), oracle.C.c_float(1e-7))
        assert np.array_equal(P.astype(np.complex64).view(np.float32), z["RP%d" % i].view(np.float32)), name
        # Moore-Penrose on the side that has full rank
        E = A @ P if n <= p else P @ A
        assert np.abs(E - np.eye(min(n, p))).max() < 2e-5, name


def test_linpack_ref_live(oracle):
    """When the reference sources are present (authoring container) run csvdc itself again, on fresh matrices: same bits."""
    if oracle.build_ref() is None:
        pytest.skip("/root/reference not present")
    rng = np.random.default_rng(99)
    for n, p in ((6, 6), (1, 1), (5, 9), (9, 5), (17, 17), (40, 40)):
        for k in range(3):
            A = rng.standard_normal((n, p)) + 1j * rng.standard_normal((n, p))
            if k == 2 and n == p and n > 2:
                B = rng.standard_normal((n, n - 2)); A = (B @ B.T).astype(np.complex128)         # rank deficient, real symmetric
            ir, sr, ur, vr = oracle.ref_csvdc(A)
            io, so, uo, vo = oracle.csvdc(A)
            assert ir == io
            assert np.array_equal(sr.view(np.float32), so.view(np.float32)), (n, p, k)
            assert np.array_equal(np.ascontiguousarray(ur).view(np.float32), np.ascontiguousarray(uo).view(np.float32)), (n, p, k)
            assert np.array_equal(np.ascontiguousarray(vr).view(np.float32), np.ascontiguousarray(vo).view(np.float32)), (n, p, k)


def test_mvdr_weights_properties(oracle):
    M, Cn = 256, 8
    mp = synth.linear_array(Cn)
    delays = oracle.calc_delays_polar2(np.float32(0.5), np.float32(np.pi / 2), mp)
    wq = oracle.calc_mainlobe(16000.0, delays, M)
    assert np.allclose(wq[0], 1.0 / Cn) and np.allclose(wq[M - 5], np.conj(wq[5]))
    R = oracle.diffuse_noise_model(mp, M, 16000.0, 343740.0, mu=0.01)
    assert np.allclose(R[:, np.arange(Cn), np.arange(Cn)], 1.0) and np.allclose(R, np.conj(np.swapaxes(R, 1, 2)))
    w = oracle.mvdr_weights(wq, R)
    assert np.all(w[0] == 1.0)                                             # beamformer.cc:2413-2415
    resp = np.einsum("fc,fc->f", np.conj(w[1:]), wq[1:M // 2 + 1])
    assert np.abs(resp - 1.0 / Cn).max() < 1e-4                            # w = invR d / (C d^H invR d)
    # MVDR has lower diffuse-noise output power than delay-and-sum at every bin
    pm = np.einsum("fc,fcd,fd->f", np.conj(w[1:]), R[1:], w[1:]).real
    pd = np.einsum("fc,fcd,fd->f", np.conj(wq[1:129]), R[1:], wq[1:129]).real
    assert np.all(pm <= pd * (1 + 1e-4))


def test_blocking_matrix(oracle):
    rng = np.random.default_rng(2)
    d = rng.standard_normal(8) + 1j * rng.standard_normal(8)
    B, ok = oracle.blocking_matrix(d)
    assert ok and np.abs(np.conj(B.T) @ B - np.eye(7)).max() < 1e-12
    # columns are orthogonal to conj(d):  B^H conj(d) = 0  (P = I - conj(d) d^T / |d|^2)
    assert np.abs(np.conj(B.T) @ np.conj(d)).max() < 1e-12


# ------------------------------------------------------------------ MFCC chain
def test_mfcc_config1_shapes(oracle, headset):
    """BASELINE config 1: Headset1.wav -> 39-dim stream; frame count of SampleFeature(320,160,padZeros=False)."""
    lda = (np.random.default_rng(1234).standard_normal((39, 195)) / np.sqrt(195)).astype(np.float32)
    f = oracle.mfcc_chain(headset, oracle.mfcc_cfg(lda=lda))
    assert f.shape == (841, 39) and np.isfinite(f).all()
    c = oracle.mfcc_chain(headset, oracle.mfcc_cfg(), stage=2)
    assert np.abs(c.mean(0)).max() < 1e-3                                   # batch CMN
    s = oracle.mfcc_chain(headset, oracle.mfcc_cfg(), stage=0)             # no LDA: spliced 195-dim
    assert s.shape == (841, 195)
    assert np.array_equal(s[0, :13], s[0, 13 * 7:13 * 8]) and np.array_equal(s[100, 13 * 7:13 * 8], c[100])
    assert np.array_equal(s[-1, -13:], c[-1]) and np.array_equal(s[0, 13 * 8:13 * 9], c[1])


def test_mfcc_operators_against_numpy(oracle, headset):
    x = headset[8000:8000 + 4000]
    blk = oracle.sample_blocks(x, 320, 160, False)
    assert blk.shape[0] == 23 and np.array_equal(blk[3], x[480:800])
    assert oracle.sample_blocks(x, 320, 160, True).shape[0] == 25
    pw = oracle.mfcc_chain(x, oracle.mfcc_cfg(), stage=4)
    # pre-emphasis carries its prior across the overlapping blocks (feature.cc:1164-1167)
    t = 5
    b = blk[t].astype(np.float64); prior = np.concatenate([[blk[t - 1][-1]], b[:-1]])
    pre = (b - 0.95 * prior).astype(np.float32)
    ham = ((0.54 - 0.46 * np.cos(2 * np.pi * np.arange(320) / 319.0)) * pre).astype(np.float32)
    ref = np.abs(np.fft.rfft(ham.astype(np.float64), 512)) ** 2
    assert np.abs(pw[t] - ref).max() / ref.max() < 1e-6
    rows = oracle.melbank(257, 16000.0, 0.0, 0.0, 30, 1)
    assert len(rows) == 30 and rows[0][0] == 0 and all(len(c) > 0 for _, c in rows)
    assert rows[0][1][-1] < 0                                               # v1 quirk: last tap lies past the right edge
    assert rows[-1][0] + len(rows[-1][1]) <= 257
    r2 = oracle.melbank(257, 16000.0, 0.0, 0.0, 30, 2)
    assert not np.array_equal(rows[10][1], r2[10][1])                       # v1 evaluates the triangle one bin late
    Cm = oracle.cosine_matrix(13, 30, 1)
    assert np.allclose(Cm[0], 1.0) and np.allclose(Cm[1, 0], np.cos(np.pi * 0.5 / 30))


def test_vtln_identity_and_warp(oracle):
    pw = np.abs(np.random.default_rng(3).standard_normal((4, 257))) + 1.0
    out = oracle.vtln(pw, 1.0, 1.0, 1)
    assert np.abs(out - pw).max() / pw.max() < 1e-9
    w = oracle.vtln(pw, 1.1, 0.8, 1)
    assert np.isfinite(w).all() and abs(w.sum() / pw.sum() - 1.0) < 0.05   # interval integration conserves mass
    assert np.isfinite(oracle.vtln(pw, 0.9, 0.8, 2)).all()


def test_blockconv_is_plain_framing(oracle):
    x = np.arange(128 * 20, dtype=np.float32)
    out = oracle.blockconv(x.reshape(20, 128), 320, 160)
    assert out.shape[0] == (len(x) - 320) // 160 + 1
    for t in (0, 1, 7, out.shape[0] - 1):
        assert np.array_equal(out[t], x[160 * t:160 * t + 320])


def test_cmn_and_adjacent(oracle):
    x = np.random.default_rng(4).standard_normal((50, 13)).astype(np.float32)
    y, mean, var = oracle.cmn_batch(x, 3.0)
    assert np.abs(mean - x.mean(0)).max() < 1e-5 and np.abs(var - x.var(0)).max() < 1e-4
    assert np.abs(y - (x - mean) / (3.0 * np.sqrt(var))).max() < 1e-5
    r = oracle.cmn_runon(x, 0.0)
    m = np.zeros(13, np.float32)
    for t in range(3):
        m = (np.float32(0.98) * m + (1.0 - np.float32(0.98)) * x[t]).astype(np.float32)
    assert np.abs(r[2] - (x[2] - m)).max() < 1e-6
    a = oracle.adjacent(x, 5)
    assert a.shape == (50, 143) and np.array_equal(a[0, :13], x[0]) and np.array_equal(a[49, -13:], x[49])
    assert np.array_equal(a[20, 5 * 13:6 * 13], x[20]) and np.array_equal(a[20, :13], x[15])
    assert oracle.adjacent(x[:3], 5).shape[0] == 0                           # cannot be primed


# ------------------------------------------------------------------ GMM
def test_gmm_against_float64(oracle):
    m = synth.gmm_model(8, 16, 39, seed=5)
    cb = oracle.Codebooks(m["refN"], m["mean"], m["ivar"], m["det"])
    x = np.random.default_rng(6).standard_normal((200, 39)).astype(np.float32)
    sc, am = oracle.gmm_score_opt(cb, m["val"], x)
    d = ((m["mean"][None].astype(np.float64) - x[:, None]) ** 2 * m["ivar"][None]).sum(-1) + 39 * np.log(2 * np.pi) + m["det"][None]
    d = d.reshape(200, 8, 16)
    assert np.array_equal(d.argmin(-1), am)
    ref = 0.5 * d.min(-1) + m["val"].reshape(8, 16)[np.arange(8)[None], d.argmin(-1)]
    assert np.abs(sc - ref).max() / np.abs(ref).max() < 1e-5
    sa = oracle.gmm_score_all(cb, m["val"], x)
    lse = -np.log(np.exp(-(0.5 * d + m["val"].reshape(1, 8, 16))).sum(-1))
    assert np.abs(sa - lse).max() / np.abs(lse).max() < 1e-5
    assert np.all(sa <= sc + 1e-4)                                          # full mixture is never worse than its best term


def test_gmm_files_byte_exact(oracle, tmp_path):
    m = synth.gmm_model(3, 4, 5, seed=7)
    cb = oracle.Codebooks(m["refN"], m["mean"], m["ivar"], m["det"])
    p = str(tmp_path / "cb.bin")
    assert oracle.cbset_save(cb, ["a", "bb", "ccc"], p) == 0
    raw = open(p, "rb").read()
    assert raw[:4] == (64207531).to_bytes(4, "big") and raw[8:12] == (3).to_bytes(4, "big")   # big endian magic, count
    assert raw[12:16] == b"\x00\x01a\x00"                                                      # int16 length + bytes + NUL
    assert raw[-4:] == (123456789).to_bytes(4, "big")
    cb2, names = oracle.cbset_load(p)
    assert names == ["a", "bb", "ccc"] and np.array_equal(cb2.mean, cb.mean) and np.array_equal(cb2.det, cb.det)


# ------------------------------------------------------------------ WFST + decoder
def _brute_force_best(ex, scores, lmScale, finals_only=True):
    """Exhaustive Viterbi in float64 over (frame, node) for graphs WITHOUT epsilon arcs."""
    n = len(ex["nodeState"]); T = scores.shape[0]
    best = np.full(n, np.inf); best[0] = 0.0
    for t in range(T):
        nb = np.full(n, np.inf)
        for s in range(n):
            if not np.isfinite(best[s]):
                continue
            for a in range(ex["arcOff"][s], ex["arcOff"][s + 1]):
                v = best[s] + lmScale * ex["arcCost"][a] + scores[t, ex["arcIn"][a] - 1]
                if v < nb[ex["arcDst"][a]]:
                    nb[ex["arcDst"][a]] = v
        best = nb
    fin = best + np.where(ex["nodeFinal"] == 1, lmScale * ex["nodeCost"], np.inf)
    return fin.min()


def test_decoder_matches_exhaustive_viterbi(oracle):
    arcs, fin = synth.random_wfst(60, 6, seed=3, eps_frac=0.0, nFinal=10)
    g = oracle.Wfst()
    for a in arcs:
        g.add_arc(*a)
    for s, c in fin:
        g.add_final(s, c)
    sc = np.random.default_rng(8).uniform(0, 10, (25, 6)).astype(np.float32)
    r = g.decode(sc, beam=1e9, lmScale=2.0)
    assert r["rc"] == 0 and r["reachedFinal"]
    ref = _brute_force_best(g.export(), sc.astype(np.float64), 2.0)
    assert abs(r["score"] - ref) / ref < 1e-5
    assert len(r["arcs"]) == 25 and np.array_equal(r["arcFrames"], np.arange(25))


def test_wfst_container_semantics(oracle, tmp_path):
    g = oracle.Wfst()
    g.add_arc(5, 6, 1, 0, 1.0); g.add_arc(5, 7, 2, 9, 2.0); g.add_arc(6, 5, 0, 0, 0.5); g.add_arc(7, 7, 0, 0, 0.0)
    g.add_final(7, 0.25)
    ex = g.export()
    assert ex["nodeState"][0] == 5                                           # source of the first arc = initial node
    assert list(ex["arcDst"][ex["arcOff"][0]:ex["arcOff"][1]]) == [2, 1]     # prepended: reverse file order
    assert len(ex["arcDst"]) == 3                                            # eps:eps self loop dropped by the text reader
    t, b = str(tmp_path / "g.txt"), str(tmp_path / "g.bin")
    assert g.write(t, binary=False) == 0 and g.write(b, binary=True) == 0
    raw = open(b, "rb").read()
    assert raw[:4] == (6).to_bytes(4, "big") and raw[-4:] == (2147483647).to_bytes(4, "big")
    for path, binary in ((t, False), (b, True)):
        g2 = oracle.Wfst(); assert g2.read(path, binary) == 0
        e2 = g2.export()
        assert sorted(zip(e2["nodeState"][e2["arcDst"]], e2["arcIn"], e2["arcOut"])) == sorted(zip(ex["nodeState"][ex["arcDst"]], ex["arcIn"], ex["arcOut"]))
    assert g.add_final(7, 0.0) != 0                                          # "Automaton already has final node"


def test_decoder_edge_cases(oracle):
    arcs, fin = synth.random_wfst(100, 8, seed=1)
    g = oracle.Wfst()
    for a in arcs:
        g.add_arc(*a)
    for s, c in fin:
        g.add_final(s, c)
    assert g.decode(np.zeros((0, 8), np.float32))["rc"] == -8                # no frames: the exception escapes decode()
    sc = np.random.default_rng(2).uniform(0, 10, (30, 8)).astype(np.float32)
    r1 = g.decode(sc, beam=1e9); r2 = g.decode(sc, beam=5.0)
    assert r2["score"] >= r1["score"] and r2["activeHypos"] < r1["activeHypos"]
    r3 = g.decode(sc, beam=1e9, lmPenalty=1.0)
    assert r3["score"] >= r1["score"]
    # tokens are floats: the stored total equals float(ac)+float(lm) in double
    assert r1["score"] == float(np.float32(r1["ac"])) + float(np.float32(r1["lm"]))


# ------------------------------------------------------------------ PerfectReconstructionFFT banks, LPC envelopes
@pytest.mark.parametrize("M,m,r", [(16, 2, 0), (16, 3, 1), (8, 2, 2)])
def test_pr_banks_match_closed_forms(oracle, M, m, r):
    """The ring-buffer restatement of modulated.cc:686-970 against the closed forms the device kernels use."""
    rng = np.random.default_rng(3 + M + m)
    M2, N, D, R, pd = 2 * M, 2 * M * m, M >> r, 1 << r, 2 * m - 1
    h = np.sin(np.pi * (np.arange(N) + 0.5) / N)
    x = rng.standard_normal(300).astype(np.float32)
    X = oracle.pr_analysis_bank(x, h, M, m, r)
    assert X.shape == ((len(x) + D - 1) // D + pd, M2)
    xp = np.concatenate([np.zeros(8 * N), x.astype(np.float64), np.zeros(8 * N)]); off = 8 * N
    w = np.exp(-1j * np.pi * np.arange(M2) / M2)
    for t in range(X.shape[0]):
        nt = (t + 1) * D - 1
        u = np.zeros(M2)
        for k in range(m):
            u += (-1) ** k * h[np.arange(M2) + M2 * k] * xp[off + nt - np.arange(M2) - (r + 2) * k * D]
        ref = np.fft.ifft(w * u)
        assert np.abs(X[t] - ref).max() < 1e-9 * (1 + np.abs(ref).max())
    g = rng.standard_normal(N); T = 20
    Y = rng.standard_normal((T, M2)) + 1j * rng.standard_normal((T, M2))
    out = oracle.pr_synthesis_bank(Y, g, M, m, r)
    V = np.real(np.fft.fft(Y, axis=1) * np.exp(1j * np.pi * np.arange(M2) / M2))
    conv = {}
    for t in range(pd, T):
        c = np.zeros(M2); flip = 1 if m % 2 == 1 else -1
        for k in range(m):
            tv = t - (r + 2) * k
            if tv >= 0:
                c += flip * g[np.arange(M2) + M2 * (m - k - 1)] * V[tv]
            flip *= -1
        conv[t] = c
    ref = np.zeros((T - pd, D))
    for b in range(T - pd):
        for s in range(2 * R):
            tt = b + pd - (2 * R - s - 1)
            if tt in conv:
                ref[b, ::-1] += conv[tt][s * D:(s + 1) * D] / R
    assert out.shape == ref.shape and np.abs(out - ref).max() < 1e-6 * np.abs(ref).max()


def test_lpc_envelopes_against_numpy(oracle):
    """lpc.cc / lpc.h restatement: Levinson-Durbin solves the normal equations; the envelopes follow from the coefficients."""
    rng = np.random.default_rng(5)
    dim, order = 320, 12
    e = rng.standard_normal(dim + 64); y = np.zeros(dim + 64)
    for n in range(2, dim + 64):
        y[n] = 1.2 * y[n - 1] - 0.7 * y[n - 2] + e[n]
    x = (y[64:] * np.hamming(dim)).astype(np.float32)
    import ctypes as C
    L = oracle.lib()
    A = np.zeros(order + 1, np.float32); E = np.zeros(order + 1, np.float32)
    L.orc_lpc_warp_autocorr(x.ctypes.data_as(C.c_void_p), dim, order, C.c_float(0.0), A.ctypes.data_as(C.c_void_p), E.ctypes.data_as(C.c_void_p))
    r = np.array([np.dot(x[:dim - i].astype(np.float64), x[i:].astype(np.float64)) for i in range(order + 1)])
    Rm = np.array([[r[abs(i - j)] for j in range(order)] for i in range(order)])
    a = np.linalg.solve(Rm, r[1:])
    assert np.allclose(-A[1:], a, rtol=2e-3, atol=2e-4)               # fp32 recursion vs fp64 solve
    assert abs(E[0] - r[0]) < 1e-5 * r[0]
    H = np.abs(np.fft.rfft(A.astype(np.float64), 512)) ** 2
    lpc = oracle.lpc_feature(x[None], order, 0.0, 0, 1)[0]
    assert np.allclose(lpc, 2 * E[0] / (H[:161] * dim), rtol=1e-5)
    # MVDR envelope from its definition (lpc.h:153-192)
    mv = oracle.lpc_feature(x[None], order, 0.0, 0, 0)[0]
    pc = np.zeros(2 * order + 1)
    for i in range(order + 1):
        pc[order + i] = -sum((order + 1 - i - 2 * ii) * float(A[ii]) * float(A[ii + i]) for ii in range(order - i + 1))
        pc[order - i] = pc[order + i]
    P = np.abs(np.fft.rfft(pc, 512)) ** 2
    assert np.allclose(mv, E[0] / np.sqrt(P[:161]), rtol=1e-4)
    # Burg: same resonance, A[0] = 1
    Ab = np.zeros(order + 1, np.float32); Eb = np.zeros(order + 1, np.float32)
    L.orc_lpc_burg_autocorr(x.ctypes.data_as(C.c_void_p), dim, order, Ab.ctypes.data_as(C.c_void_p), Eb.ctypes.data_as(C.c_void_p))
    assert Ab[0] == 1.0 and abs(Ab[1] - A[1]) < 0.05 and abs(Ab[2] - A[2]) < 0.05
    with pytest.raises(ValueError):
        oracle.lpc_feature(x[None], 161, 0.0, 0, 0)


def test_zelinski_postfilter_against_numpy(oracle):
    """postfilter.cc:56-221,428-493 restated vs a direct numpy recursion (alpha = 0 for the first two frames, minFrames, clamps)."""
    rng = np.random.default_rng(2)
    Cn, T, F, alpha, minFrames = 5, 12, 7, 0.6, 2
    wq = np.exp(-1j * rng.uniform(0, 6, (F, Cn))) / Cn
    X = rng.standard_normal((Cn, T, F)) + 1j * rng.standard_normal((Cn, T, F))
    Y = np.einsum("fc,ctf->tf", np.conj(wq), X)
    for ptype in (1, 2):
        out, w = oracle.zelinski_postfilter(X, Y, wq, alpha, ptype, minFrames)
        phi = np.zeros((F, Cn, Cn), complex)
        for t in range(T):
            a = alpha if t - 1 > 0 else 0.0
            ta = np.conj(wq).T[:, None, :] * X[:, t:t + 1, :]          # [C][1][F]
            ta = ta[:, 0, :]
            for f in range(F):
                outer = np.outer(ta[:, f], np.conj(ta[:, f]))
                phi[f] = a * phi[f] + (1 - a) * outer if a > 0 else outer
                iu = np.triu_indices(Cn, 1)
                s = phi[f][iu].sum()
                num = max(s.real, 0.0) if (ptype & 1) and (t - 1 >= minFrames) else abs(s)
                W = min(max(num / np.trace(phi[f]).real * 2.0 / (Cn - 1.0), 1e-4), 1.0)
                assert abs(W - w[t, f]) < 1e-12
                ref = Y[t, f] if t - 1 < minFrames else W * Y[t, f]
                assert abs(out[t, f] - ref) < 1e-12
    with pytest.raises(ValueError):
        oracle.zelinski_postfilter(X[:1], Y, wq[:, :1])


def test_wpe_single_against_numpy(oracle):
    """dereverberation.cc:93-226 restated vs numpy normal equations (first iteration: theta_n = max(|y_n|, 1e-3)^2)."""
    rng = np.random.default_rng(9)
    N, M, lowerN, upperN, loadDb = 80, 8, 2, 6, -20.0
    P = upperN - lowerN + 1
    Y = rng.standard_normal((N, M)) + 1j * rng.standard_normal((N, M))
    out, gn = oracle.wpe_single(Y, lowerN, upperN, 1, loadDb, 0.0, 16000.0)
    for b in (0, 3, 7):
        y = Y[:, b]
        th = np.maximum(np.abs(y), 1e-3) ** 2
        R = np.zeros((P, P), complex); r = np.zeros(P, complex)
        for n in range(lowerN, N):
            lag = np.array([y[n - lowerN - l] if n - lowerN - l >= 0 else 0.0 for l in range(P)])
            R += np.outer(lag, np.conj(lag)) / th[n]; r += np.conj(y[n]) * lag / th[n]
        d = np.abs(np.diag(R)); R[np.diag_indices(P)] = d + d.max() * 10 ** (loadDb / 10)
        g = np.linalg.solve(R, r)
        assert np.abs(g - gn[b]).max() < 1e-10 * max(1.0, np.abs(g).max())
        pred = np.array([sum(np.conj(g[l]) * (y[n - lowerN - l] if n - lowerN - l >= 0 else 0.0) for l in range(P)) if n >= lowerN else 0.0 for n in range(N)])
        assert np.abs(out[:, b] - (y - pred)).max() < 1e-10


def test_lefkimmiatis_postfilter_against_numpy(oracle):
    """postfilter.cc:981-1176 restated vs a direct numpy evaluation: pseudo-inverse with the absolute singular-value floor,
    Lambda = d^H pinv(R) d, the coherence-based noise estimate and the weight rule on both sides of fbinX1."""
    rng = np.random.default_rng(4)
    Cn, T, F, alpha, fb1, thr = 4, 9, 6, 0.7, 3, 0.99
    X = rng.standard_normal((Cn, T, F)) + 1j * rng.standard_normal((Cn, T, F))
    d = np.exp(1j * rng.uniform(0, 6, (F, Cn))) / Cn
    Y = np.einsum("fc,ctf->tf", np.conj(d), X)
    R = np.zeros((F, Cn, Cn), complex)
    for f in range(F):
        A = rng.standard_normal((Cn, Cn)) + 1j * rng.standard_normal((Cn, Cn))
        R[f] = np.eye(Cn) + 0.15 * (A + A.conj().T)
    R[1] = np.ones((Cn, Cn))                                   # rank one: three singular values fall below the floor
    lam = oracle.lefkimmiatis_lambda(R, d, 1e-4)
    for f in range(F):
        if f == 1:                                             # a dropped singular value: the reference falls back to the identity
            ref = np.vdot(d[f], d[f])
        else:
            ref = np.conj(d[f]) @ np.linalg.pinv(R[f]) @ d[f]
        assert abs(lam[f] - ref) <= 2e-5 * max(1.0, abs(ref)), f  # single-precision SVD in the reference's pseudoinverse
    out, wp = oracle.lefkimmiatis_postfilter(X, Y, d, R, lam, alpha, 2, 1, thr, fb1)
    csd = np.zeros((F, Cn, Cn), complex)
    for t in range(T):
        a = alpha if t - 1 > 0 else 0.0
        for f in range(F):
            ta = np.conj(d[f]) * X[:, t, f]
            new = np.outer(ta, np.conj(ta))
            csd[f] = new if a == 0.0 else a * csd[f] + (1 - a) * new
            iu = np.triu_indices(Cn, 1)
            psd = np.real(np.diag(csd[f]))
            Rc = R[f][iu].copy(); Rc[np.real(Rc) > thr] = thr
            hs = 0.5 * (psd[iu[0]] + psd[iu[1]])
            Rm = R[f][iu].copy(); clipm = (np.real(Rm) > thr) & (np.imag(Rm) <= 0.0); Rm[clipm] = thr      # McCowan's clip (postfilter.cc:789-826)
            phi_ss = 2.0 * abs(np.sum((csd[f][iu] - Rm * hs) / (1.0 - Rm))) / (Cn * (Cn - 1))
            phi_vv = 2.0 * abs(np.sum((hs - csd[f][iu]) / (1.0 - Rc))) / (Cn * (Cn - 1))
            W = phi_ss / (phi_ss + (phi_vv if f < fb1 else phi_vv / abs(lam[f])))
            W = min(1.0, max(1e-4, W))
            assert abs(wp[t, f] - W) <= 1e-9 * max(W, 1e-4), (t, f)
            exp = Y[t, f] * W if t - 1 >= 1 else Y[t, f]
            assert abs(out[t, f] - exp) <= 1e-9 * abs(exp)


def test_wpe_multi_oracle_properties(oracle):
    """dereverberation.cc:281-586 restated: with one channel it is the single-channel operator; with two, the filters solve the
    loaded normal equations built from the stacked lags (direct numpy evaluation of one iteration), and filterChan picks the filter."""
    rng = np.random.default_rng(8)
    N, M, lowerN, upperN, loadDb = 50, 6, 2, 4, -20.0
    P = upperN - lowerN + 1
    Y1 = rng.standard_normal((1, N, M)) + 1j * rng.standard_normal((1, N, M))
    o1, g1 = oracle.wpe_multi(Y1, lowerN, upperN, 2, loadDb, 0.0, 16000.0)
    os_, gs = oracle.wpe_single(Y1[0], lowerN, upperN, 2, loadDb, 0.0, 16000.0)
    assert np.array_equal(o1[0], os_) and np.array_equal(g1[0], gs)
    Cn = 2
    Y = rng.standard_normal((Cn, N, M)) + 1j * rng.standard_normal((Cn, N, M))
    out, gn = oracle.wpe_multi(Y, lowerN, upperN, 1, loadDb, 0.0, 16000.0)
    b = 1
    lags = np.zeros((N, Cn * P), complex)
    for n in range(lowerN, N):
        for c in range(Cn):
            for l in range(P):
                ix = n - lowerN - l
                lags[n, c * P + l] = Y[c, ix, b] if ix >= 0 else 0.0
    for c in range(Cn):
        th = np.maximum(np.abs(Y[c, :, b]), 1e-3) ** 2              # first iteration: filters are zero
        R = np.zeros((Cn * P, Cn * P), complex); r = np.zeros(Cn * P, complex)
        for n in range(lowerN, N):
            R += np.outer(lags[n], np.conj(lags[n])) / th[n]; r += np.conj(Y[c, n, b]) * lags[n] / th[n]
        d = np.abs(np.diag(R)); R[np.diag_indices(Cn * P)] = d + d.max() * 10 ** (loadDb / 10)
        g = np.linalg.solve(R, r)
        assert np.abs(gn[c, b] - g).max() <= 1e-9 * np.abs(g).max()
        pred = lags @ np.conj(g)
        exp = Y[c, :, b] - np.where(np.arange(N) >= lowerN, pred, 0.0)
        assert np.abs(out[c, :, b] - exp).max() <= 1e-9
    out0, gn0 = oracle.wpe_multi(Y, lowerN, upperN, 1, loadDb, 0.0, 16000.0, filterChan=0)
    assert np.array_equal(gn0, gn) and np.array_equal(out0[0], out[0]) and not np.array_equal(out0[1], out[1])
    pred = lags @ np.conj(gn[0, b])
    assert np.abs(out0[1, :, b] - (Y[1, :, b] - np.where(np.arange(N) >= lowerN, pred, 0.0))).max() <= 1e-9


def test_gsc_rls_against_numpy(oracle):
    """beamformer.cc:1627-1698 restated vs numpy matrix algebra: Z = B^H X, g = P Z / (mu + Z^H P Z), P <- (P - g Z^H P) / mu,
    wa <- (I - sigma2 P) wa + g conj(Y), threshold constraint; the frame's output uses the weights before the update."""
    rng = np.random.default_rng(6)
    Cn, T, M, mu, s2, s2i, alpha = 4, 25, 8, 0.9, 0.01, 0.05, 0.2
    F, n = M // 2 + 1, Cn - 1
    d = oracle.calc_delays_polar2(np.float32(0.7), np.float32(1.2), synth.linear_array(Cn)); wq = oracle.calc_mainlobe(16000.0, d, M)
    B = np.array([oracle.blocking_matrix(wq[f])[0] for f in range(F)])
    X = rng.standard_normal((Cn, T, M)) + 1j * rng.standard_normal((Cn, T, M))
    Y, waF = oracle.gsc_rls(X, wq, B, mu, s2, s2i, alpha, 2, True, False)
    muf, s2f, s2if, af = float(np.float32(mu)), float(np.float32(s2)), float(np.float32(s2i)), float(np.float32(alpha))
    P = [np.eye(n, dtype=complex) * float(np.float32(1.0) / np.float32(s2i)) for _ in range(F)]; wa = np.zeros((F, n), complex)
    for t in range(T):
        assert abs(Y[t, 0] - np.vdot(wq[0], X[:, t, 0])) < 1e-12
        for f in range(1, F):
            x = X[:, t, f]
            y = np.vdot(wq[f] - B[f] @ wa[f], x)
            assert abs(Y[t, f] - y) <= 1e-9 * max(1.0, abs(y)), (t, f)
            Z = B[f].conj().T @ x
            PZ = P[f] @ Z; PH = P[f].conj().T @ Z
            g = (PZ / muf) / (np.vdot(PH, Z) / muf + 1.0)
            P[f] = (P[f] - np.outer(g, PH.conj())) / muf
            w2 = (np.eye(n) - s2f * P[f]) @ wa[f] + g * np.conj(y)
            nr = np.linalg.norm(w2)
            if nr * nr >= af:
                w2 = w2 * (af / nr)
            wa[f] = w2
    assert np.abs(waF[1:] - wa[1:]).max() <= 1e-9 * max(1.0, np.abs(wa).max())
    Yoff, _ = oracle.gsc_rls(X, wq, B, mu, s2, s2i, alpha, 2, False, False)
    assert np.abs(Yoff - oracle.gsc_apply(X, wq, B, np.zeros((F, n), complex))).max() < 1e-12
