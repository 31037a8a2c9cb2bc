"""BASELINE configs[4]: 64-channel array, MVDR + post-filter + dereverberation on long streams handed over in blocks with carried state.

The reference operators are frame-by-frame streams, so "a 10-minute stream" and "one utterance" are the same computation for them.  The
batched device path processes a stream in blocks and carries what the reference keeps in its objects between frames: the analysis bank's
m*M samples (modulated.h:79-163), the synthesis bank's R*m subband frames (modulated.cc:586-664), the post-filter's spectral densities
(postfilter.cc:428-497), the RLS precision matrices and active weights (beamformer.cc:1552-1700), the WPE filters (dereverberation.cc:258-277).
Every test here compares the BLOCK-WISE device result with the ONE-SHOT oracle over the whole stream."""
import numpy as np
import pytest

from tests import synth
from tests.conftest import load_proto

pytestmark = pytest.mark.gpu


def planar_array(k=8, pitch_mm=20.0):
    """8 x 8 planar array, 20 mm pitch"""
    mp = np.zeros((k * k, 3), np.float64)
    g = (np.arange(k) - (k - 1) / 2.0) * pitch_mm
    mp[:, 0] = np.repeat(g, k); mp[:, 1] = np.tile(g, k)
    return mp


def planar_signal(nsamp, mp, seed, az=0.6, el=1.1, sigma=3000.0, noise=300.0, fs=16000.0):
    """far-field low-passed white source on the array + independent sensor noise, int16-ranged fp32"""
    rng = np.random.default_rng(seed)
    src = rng.standard_normal(nsamp + 64) * sigma
    k = np.hanning(9); k /= k.sum(); src = np.convolve(src, k, mode="same")
    dirv = -np.array([np.sin(el) * np.cos(az), np.sin(el) * np.sin(az), np.cos(el)])
    tau = (mp @ dirv) / 343740.0 * fs
    S = np.fft.rfft(src); f = np.arange(len(S)) / float(len(src))
    out = np.zeros((mp.shape[0], nsamp), np.float32)
    for c in range(mp.shape[0]):
        d = np.fft.irfft(S * np.exp(-2j * np.pi * f * tau[c]), len(src))
        out[c] = (d[32:32 + nsamp] + rng.standard_normal(nsamp) * noise).astype(np.float32)
    return out


def _full(a, M):
    """[...][F] unique bins -> [...][M] with the conjugate mirror (what the reference's vectors hold)"""
    F = M // 2 + 1
    f = np.zeros(a.shape[:-1] + (M,), np.complex128); f[..., :F] = a; f[..., F:] = np.conj(a[..., 1:F - 1][..., ::-1]); return f


@pytest.mark.parametrize("dct", [0, 2])
def test_64ch_blockwise_front_end_matches_the_one_shot_oracle(dsr, oracle, cuda, dct):
    """analysis (64 ch) -> MVDR (diffuse model, csvdc pseudo-inverse) -> Zelinski post-filter -> single-channel WPE -> synthesis, in four blocks
    of unequal length (the last one ragged), against the oracle run once over the whole stream."""
    import torch
    M, m, r = 256, 4, 1
    D = M >> r; F = M // 2 + 1
    h, g = load_proto("M256-m4-r1")
    Cn = 64; mp = planar_array()
    blocks = [40 * D, 24 * D, 33 * D, 17 * D + 57]                       # samples per block; the last one is not a multiple of D
    N = sum(blocks)
    x = np.stack([planar_signal(N, mp, seed=100 + u) for u in range(2)])   # [U][C][N]
    U = x.shape[0]
    delays = dsr.calcDelaysPolar2(np.float32(0.6), np.float32(1.1), mp)
    bf = dsr.Beamformer(M, Cn); bf.calcArrayManifoldVectors(16000.0, delays); bf.setDiffuseNoiseModel(mp, 16000.0, 343740.0)
    bf.divideAllNonDiagonalElements(0.01); bf.calcMVDRWeights(16000.0, 1e-8); bf.select("mvdr")
    W = bf.get(1); wq = bf.get(0)
    # weights: the reference's arithmetic on the oracle side (csvdc restated, pinned by the reference's own routine)
    Ro = oracle.diffuse_noise_model(mp, M, 16000.0, 343740.0, mu=0.01)
    Wo = oracle.mvdr_weights(oracle.calc_mainlobe(16000.0, delays, M), Ro, 1e-8)
    assert np.abs(W - Wo).max() <= 1e-12 * np.abs(Wo).max()
    resp = np.einsum("fc,fc->f", np.conj(W[1:]), wq[1:F]); assert np.abs(resp - 1.0 / Cn).max() < 2e-3     # distortionless (fp32 SVD of 64 x 64)

    ana = dsr.FilterBank(h, M, m, r, False, dct); syn = dsr.FilterBank(g, M, m, r, True, dct)
    sa = dsr.FilterBankState(ana, U, Cn); ss = dsr.FilterBankState(syn, U)
    pf = dsr.ZelinskiPostFilter(M, Cn, wq[:F], alpha=0.6, type=2, minFrames=0); pf.carry(True)
    lowerN, upperN = 2, 5
    xd = torch.from_numpy(x).to(cuda)
    ys, Ys, Zs, Vs = [], [], [], []
    gn = torch.zeros((U, F, upperN - lowerN + 1), dtype=torch.complex128, device=cuda)
    off = 0
    for bi, nb in enumerate(blocks):
        last = bi == len(blocks) - 1
        X = sa.analysis_block(xd[:, :, off:off + nb].contiguous(), last=last); off += nb
        Y = bf.apply(X)
        Z = pf.apply(X, Y)
        V, gn = dsr.wpe_single(Z, M, lowerN, upperN, 2, -20.0, 0.0, 16000.0, gn=gn)    # per block; the filters carry over like across reset()
        y = ss.synthesis_block(V)
        Ys.append(Y.cpu().numpy()); Zs.append(Z.cpu().numpy()); Vs.append(V.cpu().numpy()); ys.append(y.cpu().numpy())
    Yb, Zb, Vb, yb = (np.concatenate(a, axis=1) for a in (Ys, Zs, Vs, ys))
    for u in range(U):
        Xo = np.stack([oracle.analysis_bank(x[u, c], h, M, m, r, dct) for c in range(Cn)])           # [C][T][M]
        T = Xo.shape[1]
        assert Yb.shape[1] == T                                                                      # same frame count as one stream
        Yo = oracle.beamform_apply(Xo, Wo)
        rms = np.sqrt(np.mean(np.abs(Yo[:, :F]) ** 2))
        assert np.abs(Yb[u] - Yo[:, :F]).max() < 2e-5 * rms * np.sqrt(M)
        Zo, _ = oracle.zelinski_postfilter(Xo[:, :, :F], Yo[:, :F], wq[:F], 0.6, 2, 0)
        assert np.abs(Zb[u] - Zo).max() < 2e-5 * rms * np.sqrt(M)
        # WPE block by block on the oracle's own post-filter output, filters carried as the reference's reset() does
        t0 = 0; gno = None; Vo = []
        for a in Zs:
            nT = a.shape[1]; v, gno = oracle.wpe_single(_full(Zo[t0:t0 + nT], M), lowerN, upperN, 2, -20.0, 0.0, 16000.0, gnInit=gno); Vo.append(v); t0 += nT
        Vo = np.concatenate(Vo)
        assert np.abs(Vb[u] - Vo[:, :F]).max() < 1e-4 * rms * np.sqrt(M)
        yo = oracle.synthesis_bank(Vo, g, M, m, r, dct)
        assert yb.shape[1] == len(yo)
        assert np.abs(yb[u] - yo).max() < 1e-4 * np.sqrt(np.mean(yo.astype(np.float64) ** 2)) * 4 + 1e-3
    # the carried state makes the blocks one stream: a one-shot device run over the whole stream gives the same subband frames
    X1 = ana.analysis(xd); Y1 = bf.apply(X1).cpu().numpy()
    assert np.abs(Y1 - Yb).max() <= 1e-6 * np.abs(Y1).max()


@pytest.mark.parametrize("M,m,r,name", [(512, 2, 2, "M512-m2-r2"), (512, 2, 3, "M512-m2-r3")])
def test_blockwise_filterbanks_other_designs(dsr, oracle, cuda, M, m, r, name):
    """the wave-per-frame and generic kernels, decimation R = 4 and 8 (synthesis history R*m - 1 = 7 / 15 frames), delayCompensationType 0/1/2"""
    import torch
    h, g = load_proto(name)
    D = M >> r; F = M // 2 + 1
    rng = np.random.default_rng(5 + r)
    blocks = [48 * D, 40 * D, 9 * D + 3]
    N = sum(blocks)
    x = (rng.standard_normal((2, 1, N)) * 1000).astype(np.float32)
    xd = torch.from_numpy(x).to(cuda)
    for dct in (0, 1, 2):
        ana = dsr.FilterBank(h, M, m, r, False, dct); syn = dsr.FilterBank(g, M, m, r, True, dct)
        sa = dsr.FilterBankState(ana, 2, 1); ss = dsr.FilterBankState(syn, 2)
        Xs, ys = [], []; off = 0
        for bi, nb in enumerate(blocks):
            X = sa.analysis_block(xd[:, :, off:off + nb].contiguous(), last=bi == len(blocks) - 1); off += nb
            Xs.append(X.cpu().numpy()); ys.append(ss.synthesis_block(X[:, 0].contiguous()).cpu().numpy())
        Xb = np.concatenate(Xs, axis=2); yb = np.concatenate(ys, axis=1)
        for u in range(2):
            Xo = oracle.analysis_bank(x[u, 0], h, M, m, r, dct)
            assert Xb.shape[2] == Xo.shape[0]
            rms = np.sqrt(np.mean(np.abs(Xo) ** 2))
            assert np.abs(Xb[u, 0] - Xo[:, :F]).max() < 2e-5 * rms * np.sqrt(M)
            yo = oracle.synthesis_bank(Xo, g, M, m, r, dct)
            assert yb.shape[1] == len(yo)
            assert np.abs(yb[u] - yo).max() < 2e-5 * np.sqrt(np.mean(yo.astype(np.float64) ** 2)) * np.sqrt(M)
        # a second stream on the same state objects after reset()
        sa.reset(); ss.reset()
        X = sa.analysis_block(xd[:, :, :blocks[0]].contiguous(), last=True).cpu().numpy()
        assert np.abs(X[0, 0] - oracle.analysis_bank(x[0, 0, :blocks[0]], h, M, m, r, dct)[:, :F]).max() < 2e-5 * rms * np.sqrt(M)


@pytest.mark.parametrize("kind,Cn", [("zelinski", 64), ("mccowan", 64), ("lefkimmiatis", 64), ("zelinski", 20), ("mccowan", 33)])
def test_large_array_postfilters(dsr, oracle, cuda, kind, Cn):
    """the wave-per-bin post-filter kernel (16 < C <= 64) against the oracle, one shot and in two blocks with carried densities"""
    import torch
    rng = np.random.default_rng(300 + Cn)
    U, T, M = 2, 30, 32
    F = M // 2 + 1
    mp = planar_array() if Cn == 64 else synth.linear_array(Cn, 15.0)
    wq = (np.exp(-1j * rng.uniform(0, 6, (F, Cn))) / Cn).astype(np.complex128)
    s = rng.standard_normal((U, T, F)) + 1j * rng.standard_normal((U, T, F))
    X = np.stack([s * np.conj(wq[:, c]) * Cn + 0.8 * (rng.standard_normal((U, T, F)) + 1j * rng.standard_normal((U, T, F))) for c in range(Cn)], axis=1).astype(np.complex64)
    Y = np.einsum("fc,uctf->utf", np.conj(wq), X.astype(np.complex128)).astype(np.complex64)
    alpha, ptype, minFrames = 0.7, 2, 1
    if kind == "zelinski":
        pf = dsr.ZelinskiPostFilter(M, Cn, wq, alpha=alpha, type=ptype, minFrames=minFrames)
        ref = lambda u: oracle.zelinski_postfilter(X[u].astype(np.complex128), Y[u].astype(np.complex128), wq, alpha, ptype, minFrames)
    else:
        R = oracle.pf_diffuse_noise_model(mp, M, 16000.0); R[:, np.eye(Cn, dtype=bool)] += np.float32(0.05)
        if kind == "mccowan":
            pf = dsr.McCowanPostFilter(M, Cn, wq, alpha=alpha, type=ptype, minFrames=minFrames, threshold=0.99)
            ref = lambda u: oracle.mccowan_postfilter(X[u].astype(np.complex128), Y[u].astype(np.complex128), wq, R, alpha, ptype, minFrames, 0.99)
        else:
            pf = dsr.LefkimmiatisPostFilter(M, Cn, wq, minSV=1e-8, fbinX1=3, alpha=alpha, type=ptype, minFrames=minFrames, threshold=0.99)
            lam = oracle.lefkimmiatis_lambda(R, wq, 1e-8)
            ref = lambda u: oracle.lefkimmiatis_postfilter(X[u].astype(np.complex128), Y[u].astype(np.complex128), wq, R, lam, alpha, ptype, minFrames, 0.99, 3)
        pf.setDiffuseNoiseModel(mp, 16000.0); pf.setAllLevelsOfDiagonalLoading(0.05)
    Xd, Yd = torch.from_numpy(X).to(cuda), torch.from_numpy(Y).to(cuda)
    nf = torch.tensor([T, T - 7], dtype=torch.int32, device=cuda)
    got, w = pf.apply(Xd, Yd, nframes=nf, want_weights=True)
    got, w = got.cpu().numpy(), w.cpu().numpy()
    for u in range(U):
        Tu = T if u == 0 else T - 7
        wo, ww = ref(u)
        np.testing.assert_allclose(w[u][:Tu], ww[:Tu], rtol=1e-6)
        assert np.abs(got[u][:Tu] - wo[:Tu]).max() <= 1e-6 * np.abs(wo).max()
        assert np.all(got[u][Tu:] == 0)
    assert w[0].min() < 0.95
    # two blocks with carried densities == one shot
    pf.carry(True)
    a, wa = pf.apply(Xd[:, :, :11].contiguous(), Yd[:, :11].contiguous(), want_weights=True)
    b, wb = pf.apply(Xd[:, :, 11:].contiguous(), Yd[:, 11:].contiguous(), want_weights=True)
    wcat = torch.cat([wa, wb], dim=1).cpu().numpy()
    for u in range(U):
        np.testing.assert_allclose(wcat[u], ref(u)[1], rtol=1e-6)
    pf.resetState()                                                        # new streams: the densities start from scratch again
    c, wc = pf.apply(Xd[:, :, :11].contiguous(), Yd[:, :11].contiguous(), want_weights=True)
    assert torch.equal(wc, wa)


@pytest.mark.parametrize("Cn", [8, 5, 20])
def test_small_array_postfilter_kernels_carry(dsr, oracle, cuda, Cn):
    """the thread-per-bin kernels (densities in registers for C in {2,3,4,6,8}, in the state array otherwise): blocks with carried state == one shot"""
    import torch
    rng = np.random.default_rng(400 + Cn)
    U, T, M = 3, 26, 32
    F = M // 2 + 1
    mp = synth.linear_array(Cn, 30.0)
    wq = (np.exp(-1j * rng.uniform(0, 6, (F, Cn))) / Cn).astype(np.complex128)
    X = (rng.standard_normal((U, Cn, T, F)) + 1j * rng.standard_normal((U, Cn, T, F))).astype(np.complex64)
    Y = np.einsum("fc,uctf->utf", np.conj(wq), X.astype(np.complex128)).astype(np.complex64)
    Xd, Yd = torch.from_numpy(X).to(cuda), torch.from_numpy(Y).to(cuda)
    R = oracle.pf_diffuse_noise_model(mp, M, 16000.0); R[:, np.eye(Cn, dtype=bool)] += np.float32(0.1)
    for kind in ("zelinski", "mccowan"):
        if kind == "zelinski":
            pf = dsr.ZelinskiPostFilter(M, Cn, wq, alpha=0.6, type=1, minFrames=2)
            ref = lambda u: oracle.zelinski_postfilter(X[u].astype(np.complex128), Y[u].astype(np.complex128), wq, 0.6, 1, 2)
        else:
            pf = dsr.McCowanPostFilter(M, Cn, wq, alpha=0.6, type=1, minFrames=2, threshold=0.99)
            pf.setDiffuseNoiseModel(mp, 16000.0); pf.setAllLevelsOfDiagonalLoading(0.1)
            ref = lambda u: oracle.mccowan_postfilter(X[u].astype(np.complex128), Y[u].astype(np.complex128), wq, R, 0.6, 1, 2, 0.99)
        pf.carry(True)
        outs = []
        for lo, hi in ((0, 1), (1, 9), (9, T)):                              # a first block of a single frame: the start-up alpha = 0 spans two blocks
            o, _ = pf.apply(Xd[:, :, lo:hi].contiguous(), Yd[:, lo:hi].contiguous(), want_weights=True); outs.append(o)
        got = torch.cat(outs, dim=1).cpu().numpy()
        for u in range(U):
            wo, _ = ref(u)
            assert np.abs(got[u] - wo).max() <= 1e-6 * np.abs(wo).max(), kind


@pytest.mark.parametrize("Cn,qc", [(8, 0), (5, 2), (20, 0), (64, 0)])
def test_gsc_rls_ragged_and_carried(dsr, oracle, cuda, Cn, qc):
    """a ragged batch stops adapting at each utterance's own last frame (beamformer.cc:1552-1612) -- final active weights and output
    against the oracle run per utterance on its own length; and the state carried over two blocks == the oracle run once over both
    (the reference keeps adapting across reset()).  C = 20 / 64: precision matrix in memory (the large-array path)."""
    import torch
    rng = np.random.default_rng(500 + Cn)
    U, T, M = 2, (18 if Cn == 64 else 40), 16
    F = M // 2 + 1
    mp = planar_array() if Cn == 64 else synth.linear_array(Cn, 25.0)
    delays = dsr.calcDelaysPolar2(np.float32(0.4), np.float32(1.2), mp)
    bf = dsr.Beamformer(M, Cn); bf.calcGSCWeights(16000.0, delays); bf.select("gsc"); bf.rlsConfig(0.95, 0.01)
    bf.initPrecisionMatrix(0.01)
    if qc:
        bf.setQuadraticConstraint(0.4, qc)
    wq = bf.get(0); B = bf.get(3)[:F]
    s = rng.standard_normal((U, T, F)) + 1j * rng.standard_normal((U, T, F))
    X = np.stack([s * np.conj(wq[:F, c]) * Cn + 0.7 * (rng.standard_normal((U, T, F)) + 1j * rng.standard_normal((U, T, F))) for c in range(Cn)], axis=1).astype(np.complex64)
    lens = [T, T - 11]
    Xz = X.copy(); Xz[1, :, lens[1]:] = 0                                   # the padded tail of the shorter utterance
    nf = torch.tensor(lens, dtype=torch.int32, device=cuda)
    Y, wa = bf.gsc_rls(torch.from_numpy(Xz).to(cuda), nframes=nf)
    Y, wa = Y.cpu().numpy(), wa.cpu().numpy()
    tolw = 1e-8 if Cn <= 8 else 1e-6
    for u in range(U):
        Yo, wao = oracle.gsc_rls(_full(X[u][:, :lens[u]], M), wq, B, 0.95, 0.01, 0.01, 0.4, qc, True, False)
        np.testing.assert_allclose(wa[u][1:], wao[1:], rtol=tolw, atol=1e-11)
        assert np.abs(Y[u][:lens[u]] - Yo[:, :F]).max() <= 4e-6 * np.abs(Yo).max()
        assert np.all(Y[u][lens[u]:] == 0)
    # two blocks, state carried
    bf.rlsCarry(True); bf.initPrecisionMatrix(0.01)
    cut = T // 3
    Ya, _ = bf.gsc_rls(torch.from_numpy(np.ascontiguousarray(X[:, :, :cut])).to(cuda))
    Yb, wb = bf.gsc_rls(torch.from_numpy(np.ascontiguousarray(X[:, :, cut:])).to(cuda))
    Yc = torch.cat([Ya, Yb], dim=1).cpu().numpy(); wb = wb.cpu().numpy()
    for u in range(U):
        Yo, wao = oracle.gsc_rls(_full(X[u], M), wq, B, 0.95, 0.01, 0.01, 0.4, qc, True, False)
        np.testing.assert_allclose(wb[u][1:], wao[1:], rtol=tolw, atol=1e-11)
        assert np.abs(Yc[u] - Yo[:, :F]).max() <= 4e-6 * np.abs(Yo).max()
    bf.rlsResetState()                                                     # fresh streams start from P0 and zero weights again
    Yd, _ = bf.gsc_rls(torch.from_numpy(np.ascontiguousarray(X[:, :, :cut])).to(cuda))
    assert torch.equal(Yd, Ya)


def test_ten_minute_stream_in_sixty_blocks(dsr, oracle, cuda):
    """BASELINE configs[4] at its own length: a 10-minute 64-channel stream handed over in sixty 10-second blocks (the last one ragged).  Errors of the
    carried state -- filter-bank history, post-filter densities, WPE filters -- that need many blocks to show would show here:
      * analysis -> MVDR -> Zelinski -> synthesis block by block = the same operators run ONCE over the whole stream on the device;
      * the first block against the oracle (analysis, MVDR weights through the restated csvdc, Zelinski);
      * WPE of a late block (the 58th) against the oracle started from the filters the device carried into that block."""
    import torch
    from bench_streams import planar_array, planar_block
    M, m, r = 256, 4, 1
    D = M >> r; F = M // 2 + 1
    h, g = load_proto("M256-m4-r1")
    mp = planar_array(); Cn = mp.shape[0]
    nblk, nb = 60, 1250 * D
    N = nblk * nb + 57
    x = planar_block(torch, cuda, 1, mp, N, seed=4242)                             # [1][64][N] on the device (2.5 GB)
    delays = dsr.calcDelaysPolar2(np.float32(0.6), np.float32(1.1), mp)
    bf = dsr.Beamformer(M, Cn); bf.calcArrayManifoldVectors(16000.0, delays); bf.setDiffuseNoiseModel(mp, 16000.0, 343740.0)
    bf.divideAllNonDiagonalElements(0.01); bf.calcMVDRWeights(16000.0, 1e-8); bf.select("mvdr")
    W = bf.get(1); wq = bf.get(0)
    ana = dsr.FilterBank(h, M, m, r, False, 0); syn = dsr.FilterBank(g, M, m, r, True, 0)
    sa = dsr.FilterBankState(ana, 1, Cn); ss = dsr.FilterBankState(syn, 1); sw = dsr.FilterBankState(syn, 1)
    pf = dsr.ZelinskiPostFilter(M, Cn, wq[:F], alpha=0.6, type=2, minFrames=0); pf.carry(True)
    lowerN, upperN = 2, 5
    gn = torch.zeros((1, F, upperN - lowerN + 1), dtype=torch.complex128, device=cuda)
    Ys, Zs, ys, yw = [], [], [], []
    probe = {}
    for b in range(nblk):
        lo = b * nb; hi = N if b == nblk - 1 else lo + nb
        X = sa.analysis_block(x[:, :, lo:hi].contiguous(), last=(b == nblk - 1))
        Y = bf.apply(X); Z = pf.apply(X, Y)
        if b == 57:
            probe["gn"] = gn.clone(); probe["Z"] = Z.clone()
        V, gn = dsr.wpe_single(Z, M, lowerN, upperN, 2, -20.0, 0.0, 16000.0, gn=gn)
        if b == 57:
            probe["V"] = V.clone()
        if b == 0:
            probe["X0"] = X.cpu().numpy(); probe["Y0"] = Y.cpu().numpy(); probe["Z0"] = Z.cpu().numpy()
        Ys.append(Y); Zs.append(Z); ys.append(ss.synthesis_block(Z)); yw.append(sw.synthesis_block(V))
        del X
    Yb = torch.cat(Ys, 1); Zb = torch.cat(Zs, 1); yb = torch.cat(ys, 1); ywb = torch.cat(yw, 1)
    assert bool(torch.isfinite(ywb).all()) and float(ywb.abs().max()) > 0.0        # the dereverberated stream stays finite over all sixty blocks
    # ---- one run over the whole stream
    X1 = ana.analysis(x); assert X1.shape[2] == Yb.shape[1]
    Y1 = bf.apply(X1)
    pf1 = dsr.ZelinskiPostFilter(M, Cn, wq[:F], alpha=0.6, type=2, minFrames=0)
    Z1 = pf1.apply(X1, Y1); del X1
    y1 = syn.synthesis_run(Z1)
    assert float((Y1 - Yb).abs().max()) <= 1e-6 * float(Y1.abs().max())
    assert float((Z1 - Zb).abs().max()) <= 1e-5 * float(Z1.abs().max())
    assert y1.shape == yb.shape and float((y1 - yb).abs().max()) <= 1e-5 * float(y1.abs().max())
    # ---- the first block against the oracle
    x0 = x[0, :, :nb].cpu().numpy()
    Ro = oracle.diffuse_noise_model(mp, M, 16000.0, 343740.0, mu=0.01)
    Wo = oracle.mvdr_weights(oracle.calc_mainlobe(16000.0, delays, M), Ro, 1e-8)
    assert np.abs(W - Wo).max() <= 1e-12 * np.abs(Wo).max()
    # (the block's frames reach no further than its own samples: the oracle run on the block alone gives those frames; its zero-input tail frames are dropped)
    Xo = np.stack([oracle.analysis_bank(x0[c], h, M, m, r, 0) for c in range(Cn)])[:, :probe["X0"].shape[2]]
    rms = np.sqrt(np.mean(np.abs(Xo[:, :, :F]) ** 2))
    assert np.abs(probe["X0"][0] - Xo[:, :, :F]).max() < 2e-5 * rms * np.sqrt(M)
    Yo = oracle.beamform_apply(Xo, Wo)
    rmsY = np.sqrt(np.mean(np.abs(Yo[:, :F]) ** 2))
    assert np.abs(probe["Y0"][0] - Yo[:, :F]).max() < 2e-5 * rmsY * np.sqrt(M)
    Zo, _ = oracle.zelinski_postfilter(Xo[:, :, :F], Yo[:, :F], wq[:F], 0.6, 2, 0)
    assert np.abs(probe["Z0"][0] - Zo).max() < 2e-5 * rmsY * np.sqrt(M)
    # ---- WPE of block 58, from the filters the device carried into it
    Zp = probe["Z"][0].cpu().numpy().astype(np.complex128)
    gn0 = np.ascontiguousarray(_full(probe["gn"][0].cpu().numpy().T, M).T)               # [F][P] -> [M][P]: the mirrored bins carry the conjugate filters
    Vo, _ = oracle.wpe_single(_full(Zp, M), lowerN, upperN, 2, -20.0, 0.0, 16000.0, gnInit=gn0)
    assert np.abs(probe["V"][0].cpu().numpy() - Vo[:, :F]).max() < 1e-4 * np.sqrt(np.mean(np.abs(Zp) ** 2)) * np.sqrt(M)
