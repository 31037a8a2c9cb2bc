"""bench_streams.py -- BASELINE.json configs[4] behind `bench.py --workload config5`: 64-channel (8 x 8 planar, 20 mm) array, long-form
streams handed over in 10-second blocks with carried state, analysis bank -> MVDR (diffuse-noise model) -> Zelinski post-filter ->
single-channel WPE -> synthesis bank (SURVEY.md 8d "Config 5"; reference: btk/modulated/modulated.cc:360-674, btk/beamformer/beamformer.cc:2392-2635,
btk/postfilter/postfilter.cc:428-497, btk/dereverberation/dereverberation.cc:61-270).

A step = one 10-second block of every stream of the batch (all operators keep what the reference keeps in its stream objects between frames:
filter-bank history, post-filter densities, WPE filters); K steps = K x 10 s of every stream.  value = stream-hours of enhanced audio per wall
second.  The dominant kernel's roofline uses SURVEY 8d's algorithmic bytes; its launch time comes from HIP events on the launch stream."""
import json
import os
import time

import numpy as np

HBM_PEAK_GBS = 8000.0


def planar_array(k=8, pitch_mm=20.0):
    mp = np.zeros((k * k, 3), np.float64)
    g = (np.arange(k) - (k - 1) / 2.0) * pitch_mm
    mp[:, 0] = np.repeat(g, k); mp[:, 1] = np.tile(g, k)
    return mp


def planar_block(torch, dev, U, mp, nsamp, seed, az=0.6, el=1.1, sigma=3000.0, noise=300.0, fs=16000.0):
    """One block of every stream, generated on the GPU: far-field low-passed white source on the array + independent sensor noise,
    int16-ranged fp32 [U][C][nsamp] (the generator of tests/test_gpu_config5.py, batched)."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    Cn = mp.shape[0]; n = nsamp + 64
    dirv = -np.array([np.sin(el) * np.cos(az), np.sin(el) * np.sin(az), np.cos(el)])
    tau = torch.from_numpy((mp @ dirv) / 343740.0 * fs).to(dev)
    f = torch.arange(n // 2 + 1, device=dev, dtype=torch.float64) / n
    k = torch.hann_window(9, periodic=False, device=dev, dtype=torch.float32); k = k / k.sum()
    src = torch.randn((U, 1, n), generator=g, device=dev) * sigma
    src = torch.nn.functional.conv1d(src, k.view(1, 1, -1), padding=4)[:, 0]
    S = torch.fft.rfft(src.double())
    x = torch.empty((U, Cn, nsamp), dtype=torch.float32, device=dev)
    for c in range(Cn):
        d = torch.fft.irfft(S * torch.exp(-2j * np.pi * f * tau[c]), n=n)
        x[:, c] = d[:, 32:32 + nsamp].float() + torch.randn((U, nsamp), generator=g, device=dev) * noise
    return x


class Chain:
    """the operators of one batch of streams with their carried state"""

    def __init__(self, dsr, torch, dev, U, h, g, M=256, m=4, r=1, lowerN=2, upperN=5, iters=2):
        self.dsr, self.torch, self.U, self.M = dsr, torch, U, M
        self.mp = planar_array(); self.Cn = Cn = self.mp.shape[0]; F = M // 2 + 1
        delays = dsr.calcDelaysPolar2(np.float32(0.6), np.float32(1.1), self.mp)
        self.bf = bf = dsr.Beamformer(M, Cn); bf.calcArrayManifoldVectors(16000.0, delays); bf.setDiffuseNoiseModel(self.mp, 16000.0, 343740.0)
        bf.divideAllNonDiagonalElements(0.01); bf.calcMVDRWeights(16000.0, 1e-8); bf.select("mvdr")
        self.wq = bf.get(0); self.W = bf.get(1)
        self.ana = dsr.FilterBank(h, M, m, r, False, 0); self.syn = dsr.FilterBank(g, M, m, r, True, 0)
        self.sa = dsr.FilterBankState(self.ana, U, Cn); self.ss = dsr.FilterBankState(self.syn, U)
        self.pf = dsr.ZelinskiPostFilter(M, Cn, self.wq[:F], alpha=0.6, type=2, minFrames=0); self.pf.carry(True)
        self.lowerN, self.upperN, self.iters = lowerN, upperN, iters
        self.fuse_pf = not os.environ.get("DSR_BENCH_NO_PF_FUSE")     # beamformer + post-filter as one pass over the snapshots (dsr_zelinski_apply_bf)
        self.gn = torch.zeros((U, F, upperN - lowerN + 1), dtype=torch.complex128, device=dev)
        self.ev = None

    def block(self, x, last=False, timed=False):
        """one block of every stream: [U][C][N] samples -> [U][N'] enhanced samples; timed: HIP-event intervals per stage (ms)"""
        torch, dsr = self.torch, self.dsr
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if timed else None
        if timed: ev[0].record()
        X = self.sa.analysis_block(x, last=last)
        if timed: ev[1].record()
        if self.fuse_pf:                                         # the post-filter behind its beamformer: one pass over the snapshots for both (setBeamformer)
            if timed: ev[2].record()
            Z = self.pf.apply_bf(self.bf, X)
        else:
            Y = self.bf.apply(X)
            if timed: ev[2].record()
            Z = self.pf.apply(X, Y)
        if timed: ev[3].record()
        V, self.gn = dsr.wpe_single(Z, self.M, self.lowerN, self.upperN, self.iters, -20.0, 0.0, 16000.0, gn=self.gn)
        if timed: ev[4].record()
        y = self.ss.synthesis_block(V)
        if timed: ev[5].record()
        self.ev = ev
        return y, X.shape[2]

    def stage_ms(self):
        return [self.ev[i].elapsed_time(self.ev[i + 1]) for i in range(5)]


def cpu_sample(mdl, x1, h, g, M, m, r, lowerN, upperN, iters):
    """the oracle on ONE stream's block (64 channels): the same chain, one core"""
    from oracle import oracle as O
    try:
        O.lib(native=True)
    except Exception:
        O.lib(native=False)
    F = M // 2 + 1
    t0 = time.time()
    Xo = np.stack([O.analysis_bank(x1[c], h, M, m, r, 0) for c in range(x1.shape[0])])
    Yo = O.beamform_apply(Xo, mdl.W)
    Zo, _ = O.zelinski_postfilter(Xo[:, :, :F], Yo[:, :F], mdl.wq[:F], 0.6, 2, 0)
    Zf = np.zeros((Zo.shape[0], M), np.complex128); Zf[:, :F] = Zo; Zf[:, F:] = np.conj(Zo[:, 1:F - 1][:, ::-1])
    Vo, _ = O.wpe_single(Zf, lowerN, upperN, iters, -20.0, 0.0, 16000.0)
    yo = O.synthesis_bank(Vo, g, M, m, r, 0)
    return time.time() - t0, yo


def run(args, ROOT):
    import torch
    import torch.distributed as dist
    import dsr._capi as dsr
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); lrank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lrank))
    torch.cuda.set_device(lrank); dev = torch.device("cuda", lrank)
    dsr.load()
    hg = np.load(os.path.join(ROOT, "tests", "golden", "proto_M256-m4-r1.npy")); h, g = hg[0], hg[1]
    M, m, r = 256, 4, 1; D = M >> r
    U = args.streams; nsamp = int(args.secs * 16000); nsamp -= nsamp % D           # whole hops per block: a block boundary is a frame boundary
    ch = Chain(dsr, torch, dev, U, h, g, M, m, r)
    # a few distinct blocks of input, cycled (generating 10 minutes x 64 channels per stream is set-up cost, not the measurement)
    xs = [planar_block(torch, dev, U, ch.mp, nsamp, seed=100 + 17 * rank + b) for b in range(3)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for w in range(args.warmup):
        ch.block(xs[w % len(xs)])
    sync(); t0 = time.time()
    stage = np.zeros(5); T = 0
    for k in range(args.steps):
        y, T = ch.block(xs[k % len(xs)], timed=(k >= args.steps - 8))            # stage table: event intervals of the last blocks
        if k >= args.steps - 8:
            torch.cuda.current_stream().synchronize(); stage += np.array(ch.stage_ms())
    sync(); dt = time.time() - t0
    finite = bool(torch.isfinite(y).all())
    tm = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    dt = float(tm.item())
    if rank != 0:
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return None
    nst = min(8, args.steps); stage_ms = (stage / nst).tolist()
    Cn, F = ch.Cn, M // 2 + 1
    names = ["analysis", "beamform", "postfilter", "wpe", "synthesis"]
    # SURVEY 8d bytes per stream-frame: analysis C x (512 in + 1032 out); beamformer (C + 1) rows; Zelinski (C + 2) rows (snapshots + beamformed in, filtered out);
    # single-channel WPE: a row in, a row out; synthesis a row in, D samples out
    alg = dict(analysis=U * T * Cn * 1544.0, beamform=U * T * (Cn + 1) * F * 8.0, postfilter=U * T * (Cn + 2) * F * 8.0, wpe=U * T * 2 * F * 8.0,
               synthesis=U * T * (F * 8.0 + D * 4.0))
    dom = int(np.argmax(stage_ms)); dn = names[dom]
    roof = dict(kernel=dict(analysis="k_analysis_q256", beamform="k_bf_apply", postfilter="k_zel_pairs", wpe="k_wpe", synthesis="k_synthesis")[dn],
                bound="hbm", achieved=alg[dn] / (stage_ms[dom] / 1000.0) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", traffic=None, launch_ms=stage_ms[dom])
    roof["frac"] = roof["achieved"] / roof["peak"]
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r03_config5_traffic.json")))
        if tj.get("streams") == U and tj.get("secs") == args.secs:
            roof["traffic"] = tj["kernels"][roof["kernel"]]["bytes_per_launch"]; roof["traffic_note"] = "PMC FETCH_SIZE + WRITE_SIZE (profiles/r03_config5_traffic.json)"
    except (OSError, ValueError, KeyError):
        pass
    if ch.fuse_pf:
        # the beamformer's sum is formed in the post-filter's pass over the snapshots: one stage, (C + 1) rows in (snapshots + nothing else), one row out -- SURVEY 8d's
        # bytes of the two stages minus the row they no longer hand over through memory twice
        alg["postfilter"] = U * T * (Cn + 1) * F * 8.0; names = ["analysis", "postfilter", "wpe", "synthesis"]; stage_ms = [stage_ms[0], stage_ms[2], stage_ms[3], stage_ms[4]]
        dom = int(np.argmax(stage_ms)); dn = names[dom]
        roof.update(kernel=dict(analysis="k_analysis_q256", postfilter="k_zel_pairs", wpe="k_wpe", synthesis="k_synthesis")[dn], achieved=alg[dn] / (stage_ms[dom] / 1000.0) / 1e9, launch_ms=stage_ms[dom])
        roof["frac"] = roof["achieved"] / roof["peak"]
    stages = {nm: dict(ms=round(stage_ms[i], 3), bound="hbm", achieved=round(alg[nm] / (stage_ms[i] / 1000.0) / 1e9, 1), unit="GB/s",
                       frac_of_hbm_peak=round(alg[nm] / (stage_ms[i] / 1000.0) / 1e9 / HBM_PEAK_GBS, 3)) for i, nm in enumerate(names)}
    cpu = None
    if not args.no_cpu and world == 1:
        x1 = xs[(args.steps - 1) % len(xs)][0].cpu().numpy()
        cdt, yo = cpu_sample(ch, x1, h, g, M, m, r, ch.lowerN, ch.upperN, ch.iters)
        cpu = dict(value=(nsamp / 16000.0) / 3600.0 / cdt, unit="audio_hours_per_sec", cores=1, kind="port",
                   sample="one 10-s block of one 64-channel stream through the same chain from cold state, oracle C restatement, %.1f s of CPU time" % cdt,
                   xRT=(nsamp / 16000.0) / cdt)
    audio_s = world * U * (nsamp / 16000.0)
    line = dict(metric="decoded_audio_hours_per_sec", value=audio_s / 3600.0 / (dt / args.steps), unit="audio_hours/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                ms_per_step=1000.0 * dt / args.steps, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload="BASELINE configs[4]: %d streams/GPU x %d blocks of %.2f s x 64 ch (8 x 8 planar, 20 mm), carried state: analysis M=256 m=4 r=1 -> MVDR "
                                     "(diffuse model, mu 0.01) -> Zelinski post-filter (alpha 0.6; behind its beamformer: one pass over the snapshots for both) -> single-channel WPE (taps %d..%d, %d iterations) -> synthesis; "
                                     "no decode in this configuration" % (U, args.steps, nsamp / 16000.0, ch.lowerN, ch.upperN, ch.iters),
                            streams_per_gpu=U, stream_minutes=args.steps * nsamp / 16000.0 / 60.0, frames_per_block=int(T), xRT=audio_s / (dt / args.steps),
                            output_finite=finite, parallelism="stream-sharded x%d, no exchange" % world),
                roofline=roof, stages=stages, cpu_baseline=cpu)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    return line
