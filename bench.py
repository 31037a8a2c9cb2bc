#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X: decoded audio hours per wall-second of the full
8-ch analysis filterbank -> MVDR -> synthesis -> MFCC -> 4k-Gaussian GMM -> WFST Viterbi pipe (BASELINE.json
configs[3]: 1k utterances x 10 s x 8 ch, M=256 m=4 r=1, 39-dim features, 1024 distributions x 4 Gaussians,
WFST 50k states / ~200k arcs, lmScale 12, beam tuned to ~5k active tokens).

One process per GPU; every rank decodes its own shard of utterances (weak scaling: --utts per GPU); the only
exchange is the gather of the 1-best word sequences to rank 0 over RCCL.  Rank 0 prints ONE JSON line.
Steps are independent batches and are pipelined over two pipe objects on two HIP streams (step k+1's front end under the
tail of step k's decode); all K timed steps are enqueued and collected between the two barriers.  --serial: one after the other.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver supports dmabuf IPC only: RCCL across processes needs this (already exported on the pool)

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "distantspeechrecognition-mirror_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable with a float4 copy)
MFMA_F32_PEAK_TF = 157.3


def make_input(torch, dev, U, C, nsamp, seed):
    """Synthetic far-field array recordings, generated on the GPU: white Gaussian
    source (sigma 3000) low-passed, per-channel fractional delay for a source at 30 deg on a 41 mm linear array,
    plus independent N(0, 300^2) sensor noise.  int16-ranged fp32, layout [utt][chan][sample]."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    x = torch.empty((U, C, nsamp), dtype=torch.float32, device=dev)
    pos = (torch.arange(C, device=dev, dtype=torch.float64) - (C - 1) / 2.0) * 41.0
    tau = pos * np.cos(np.deg2rad(30.0)) / 343740.0 * 16000.0
    n = nsamp + 64
    f = torch.arange(n // 2 + 1, device=dev, dtype=torch.float64) / n
    k = torch.hann_window(9, periodic=False, device=dev, dtype=torch.float32); k = k / k.sum()
    chunk = 25
    for u0 in range(0, U, chunk):
        u1 = min(U, u0 + chunk)
        src = torch.randn((u1 - u0, 1, n), generator=g, device=dev) * 3000.0
        src = torch.nn.functional.conv1d(src, k.view(1, 1, -1), padding=4)[:, 0]
        S = torch.fft.rfft(src.double())
        for c in range(C):
            d = torch.fft.irfft(S * torch.exp(-2j * np.pi * f * tau[c]), n=n)
            x[u0:u1, c] = d[:, 32:32 + nsamp].float() + torch.randn((u1 - u0, nsamp), generator=g, device=dev) * 300.0
    return x


def build_models(dsr, synth, nDist, nStates):
    hg = np.load(os.path.join(ROOT, "tests", "golden", "proto_M256-m4-r1.npy")); h, g = hg[0], hg[1]
    M, m, r, Cn = 256, 4, 1, 8
    ana = dsr.FilterBank(h, M, m, r, False, 0); syn = dsr.FilterBank(g, M, m, r, True, 0)
    mp = synth.linear_array(Cn)
    delays = dsr.calcDelaysPolar2(np.float32(np.deg2rad(30.0)), np.float32(np.pi / 2), mp)
    bf = dsr.Beamformer(M, Cn); bf.calcArrayManifoldVectors(16000.0, delays); bf.setDiffuseNoiseModel(mp, 16000.0, 343740.0)
    bf.divideAllNonDiagonalElements(0.01); bf.calcMVDRWeights(16000.0, 1e-8); bf.select("mvdr")
    lda = (np.random.default_rng(1234).standard_normal((39, 195)) / np.sqrt(195)).astype(np.float32)
    mf = dsr.Mfcc(lda=lda)
    gm_m = synth.gmm_model(nDist, 4096 // nDist, 39, seed=12)
    gm = dsr.Gmm(**gm_m)
    arcs, fin = synth.random_wfst(nStates, nDist, seed=21, outdeg=4, eps_frac=0.1, out_frac=0.05, nWords=5000, nFinal=50)
    gd = dsr.Wfst()
    for a in arcs:
        gd.add_arc(*a)
    for s, c in fin:
        gd.add_final(s, c)
    return dict(h=h, g=g, M=M, m=m, r=r, C=Cn, ana=ana, syn=syn, bf=bf, lda=lda, mf=mf, gm_m=gm_m, gm=gm, arcs=arcs, fin=fin, gd=gd,
                nArcs=len(arcs))


def tune_beam(dsr, torch, mdl, scores, nfr, target=5000.0):
    """Beam (in score units) that gives a mean of ~target active tokens per frame on a few utterances (untimed)."""
    lo, hi, best = 1.0, 4000.0, None
    for _ in range(9):
        beam = float(np.sqrt(lo * hi))
        dec = dsr.Decoder(beam=beam, lmScale=12.0, maxActive=65536, streams=scores.shape[0]); dec.set(mdl["gd"])
        out = dec.decode_batch(scores, nfr, maxPath=16)
        if any(o["status"] not in (0, 5) for o in out):
            hi = beam; continue
        act = np.mean([o["activeHypos"] / max(1, o["frames"] + 1) for o in out])
        best = (beam, act)
        if act > target:
            hi = beam
        else:
            lo = beam
        if abs(act - target) / target < 0.05:
            break
    return best


def cpu_baseline(mdl, x_host, nsamp, beam, max_seconds=12.0, threads=1):
    """The oracle (plain C restatement of the reference, -O3 -march=native build when gcc is there) on a bounded sample of the same
    workload: whole utterances through the same pipe.  threads == 1: the reference's own shape (it is single threaded).  threads > 1:
    utterance-parallel over the host cores -- worker threads inside this process (the oracle is C called through ctypes, which releases
    the GIL; it keeps no static state), so nothing is forked or exec'ed from a process that holds the GPU."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    try:
        O.lib(native=True); nat = True
    except Exception:
        nat = False
    O.lib(native=nat)
    cb = O.Codebooks(mdl["gm_m"]["refN"], mdl["gm_m"]["mean"], mdl["gm_m"]["ivar"], mdl["gm_m"]["det"])
    W = mdl["bf"].get(4); cfg = O.mfcc_cfg(lda=mdl["lda"])
    h, g, M, m, r, Cn = mdl["h"], mdl["g"], mdl["M"], mdl["m"], mdl["r"], mdl["C"]

    def graph():
        go = O.Wfst()
        for a in mdl["arcs"]:
            go.add_arc(*a)
        for s, c in mdl["fin"]:
            go.add_final(s, c)
        return go
    go = graph()                                               # read-only in orc_decode: shared by the worker threads
    t0 = time.time()

    def one(args):
        u, k = args
        if time.time() - t0 > max_seconds:
            return None
        Xc = np.stack([O.analysis_bank(x_host[u, c, :nsamp], h, M, m, r, 0) for c in range(Cn)])
        y = O.synthesis_bank(O.beamform_apply(Xc, W), g, M, m, r, 0)
        f = O.mfcc_chain(y, cfg)
        sc, _ = O.gmm_score_opt(cb, mdl["gm_m"]["val"], f, native=nat)
        return go.decode(sc, beam=beam, lmScale=12.0).get("words")
    n = x_host.shape[0]
    if threads == 1:
        words = []
        for u in range(n):
            w = one((u, 0))
            if w is None:
                break
            words.append(w)
    else:
        # thread k takes utterances k, k + threads, ...
        def lane(k):
            return [(u, one((u, k))) for u in range(k, n, threads)]
        with ThreadPoolExecutor(threads) as ex:
            got = sorted(sum(ex.map(lane, range(threads)), []))
        words = [w for _, w in got if w is not None]
    dt = time.time() - t0
    return len(words), dt, words, nat


def cpu_config1():
    """BASELINE configs[0] (the reference's own CPU-runnable case): the 1-channel recording shipped with the reference (its samples are the
    committed fixture tests/golden/Headset1_16k_s16.npy) through the MFCC chain of SURVEY.md Appendix C.3 on one host core, oracle build."""
    from oracle import oracle as O
    try:
        O.lib(native=True)
    except Exception:
        pass
    x = np.load(os.path.join(ROOT, "tests", "golden", "Headset1_16k_s16.npy")).astype(np.float32)
    lda = (np.random.default_rng(1234).standard_normal((39, 195)) / np.sqrt(195)).astype(np.float32)
    cfg = O.mfcc_cfg(lda=lda)
    O.mfcc_chain(x, cfg)                                       # warm (page in, tables)
    reps, t0 = 0, time.time()
    while time.time() - t0 < 1.5:
        f = O.mfcc_chain(x, cfg); reps += 1
    dt = (time.time() - t0) / reps
    return dict(workload="Headset1.wav (134823 samples, 16 kHz, 1 ch) -> pre-emphasis .. LDA (39-d), 1 core", frames=int(f.shape[0]),
                frames_per_sec=f.shape[0] / dt, xRT=(len(x) / 16000.0) / dt, ms=1000.0 * dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="pipe", choices=["pipe", "config5"], help="pipe: BASELINE configs[3], the full 8-ch pipe with decode (the default, the headline "
                    "metric); config5: BASELINE configs[4], 64-ch MVDR + Zelinski + WPE on long streams in 10-s blocks with carried state (bench_streams.py; --steps = blocks "
                    "per stream, default 60 = 10 minutes)")
    ap.add_argument("--streams", type=int, default=32, help="config5: streams per GPU")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps; the second and third full-batch steps of a fresh process run their first HBM-bound "
                    "kernel 6-8x slower (a power-state transient of the board, same kernel, same data: 4.5 -> 25-32 ms), so the default keeps them out of the timed region")
    ap.add_argument("--utts", type=int, default=1000, help="utterances per GPU per step (weak scaling, the default)")
    ap.add_argument("--total-utts", type=int, default=0, help="strong scaling: this many utterances in all, sharded u -> rank u mod world "
                    "(BASELINE configs[3]: --total-utts 1000); 0 = weak scaling with --utts per GPU")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host cores of the N-core CPU baseline leg (0 = all the process may use, at most 16)")
    ap.add_argument("--secs", type=float, default=10.0, help="seconds of audio per utterance")
    ap.add_argument("--states", type=int, default=50000)
    ap.add_argument("--dists", type=int, default=1024)
    ap.add_argument("--beam", type=float, default=0.0, help="0 = tune to ~5k active tokens")
    ap.add_argument("--gmm-mode", type=int, default=2, help="2: MFMA contraction + candidate search in the accumulator layout + exact re-score inside the rounding bound (argmin = mode 0 on every frame, cost rel <= 2e-6; config.gmm_mode2_vs_mode0 reports the agreement on the full batch); 0: exact VALU kernel")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--strong-total", type=int, default=1000, help="multi-GPU runs: after the weak-scaling region, time this many utterances sharded over the "
                    "ranks as a second region (BASELINE configs[3] as stated) and report it as config.strong; 0 = skip")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed mode-2-vs-mode-0 agreement run")
    ap.add_argument("--pipes", type=int, default=2, help="pipe objects / HIP streams the steps rotate over (default 2; --serial: 1)")
    ap.add_argument("--overlap", action="store_true", help="(default) two pipe objects on two HIP streams: a step is enqueued whole while the step before is still decoding, "
                    "so the front end of step k+1 runs on the CUs the persistent decode workgroups of step k free as its utterances finish (+8 %% throughput; every step "
                    "still does all of its work, and everything is collected before the clock stops)")
    ap.add_argument("--no-fuse", action="store_true", help="analysis bank and beamformer as two kernels with the channel snapshots in HBM between them "
                    "(default: one kernel, dsr_fb_analysis_beamform)")
    ap.add_argument("--serial", action="store_true", help="one pipe, steps strictly one after the other (clean per-kernel event intervals: what the counter passes of "
                    "tools/round_profiles.sh use)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the multi-rank run: nccl (= RCCL, the default) or gloo -- gloo with "
                    "DSR_BENCH_DEVICE=0 rehearses the multi-rank control flow with several ranks on ONE GPU (collectives on host tensors)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 60 if args.workload == "config5" else 5
    if args.warmup is None:
        args.warmup = 2 if args.workload == "config5" else 3
    if args.workload == "config5":
        import bench_streams
        line = bench_streams.run(args, ROOT)
        if line is not None:
            print(json.dumps(line))
        return

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); lrank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DSR_BENCH_DEVICE"):
        lrank = int(os.environ["DSR_BENCH_DEVICE"])                              # rehearsal: every rank on this device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lrank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(lrank)
    dev = torch.device("cuda", lrank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")             # where the (tiny) collectives' tensors live
    import dsr._capi as dsr
    from tests import synth
    dsr.load()

    from dsr.dist import shard_utterances
    strong = args.total_utts > 0
    my_ids = shard_utterances(args.total_utts, world, rank) if strong else list(range(args.utts))
    U, Cn, nsamp = len(my_ids), 8, int(args.secs * 16000)
    if U < 1:
        raise SystemExit("rank %d has no utterance: --total-utts must be at least the number of GPUs" % rank)
    total_utts = args.total_utts if strong else world * U
    mdl = build_models(dsr, synth, args.dists, args.states)
    x = make_input(torch, dev, U, Cn, nsamp, seed=7 + rank)
    ns_host = np.full(U, nsamp, np.int32); ns_dev = torch.from_numpy(ns_host).to(dev)

    # ---- untimed set-up: scores of a few utterances to tune the beam
    probe = dsr.Decoder(beam=1.0, maxActive=1024, streams=1); probe.set(mdl["gd"])
    pp = dsr.Pipe(mdl["ana"], mdl["syn"], mdl["bf"], mdl["mf"], mdl["gm"], probe, gmmMode=args.gmm_mode)
    nprobe = min(4, U)
    pp.run(x[:nprobe].contiguous(), ns_dev[:nprobe].contiguous(), ns_host[:nprobe], maxPath=16, want_paths=False)
    Tm = mdl["mf"].frames(((nsamp + 127) // 128) * 128)
    sc_host = pp.intermediate_host(4)[: nprobe * Tm * args.dists].reshape(nprobe, Tm, args.dists)
    sc_probe = torch.from_numpy(sc_host).to(dev)
    nfr_probe = torch.full((nprobe,), Tm, dtype=torch.int32, device=dev)
    if args.beam > 0:
        beam, act = args.beam, None
    else:
        beam, act = tune_beam(dsr, torch, mdl, sc_probe, nfr_probe)
        if world > 1:   # every rank must use the same beam
            b = torch.tensor([beam], dtype=torch.float64, device=cdev); dist.broadcast(b, 0); beam = float(b.item())
    del pp, probe
    # Steps run one after the other on one pipe (clean per-kernel event intervals).  With --overlap, two pipes on two HIP
    # streams: a step is enqueued whole and collected when the pipe is needed again, so the ragged end of one step's decode
    # (utterances finish at different times) overlaps the front end of the next step; every step still does all of its work
    # and everything is collected before the clock stops.
    maxPath = 2 * Tm + 64
    npipes = 1 if args.serial else max(2, args.pipes)
    pipes, streams = [], []
    for i in range(npipes):
        dec = dsr.Decoder(beam=beam, lmScale=12.0, maxActive=65536); dec.set(mdl["gd"])
        mf_i = mdl["mf"] if i == 0 else dsr.Mfcc(lda=mdl["lda"])          # the MFCC plan owns scratch memory: one per pipe
        pipes.append(dsr.Pipe(mdl["ana"], mdl["syn"], mdl["bf"], mf_i, mdl["gm"], dec, gmmMode=args.gmm_mode, fused=not args.no_fuse))
        streams.append(torch.cuda.Stream(device=dev))
    inflight = [False] * npipes

    def collect(i):
        res, arcs, words = pipes[i].collect(reuse=True); inflight[i] = False          # (the host arrays of a pipe are reused step after step)
        return res, words, pipes[i].stage_ms()

    def gather(r):
        # the path's only exchange: gather the 1-best word sequences on rank 0 (RCCL over xGMI)
        if world > 1:
            from dsr.dist import gather_one_best
            res, words = r[0], r[1]
            gather_one_best(words, np.array([q.nWords for q in res], np.int32), world, rank, cdev, dist)
        return r

    def finish(i):
        return gather(collect(i))

    batch = [x, ns_dev, ns_host]                                             # what a step decodes (the strong-scaling region below swaps in this rank's shard)

    def submit(i):
        with torch.cuda.stream(streams[i]):
            pipes[i].submit(batch[0], batch[1], batch[2], maxPath=maxPath, want_paths=True)
        inflight[i] = True

    def run_steps(n):
        """n steps; returns the collected results in step order.  A pipe object is collected (host wait + copy) before its next step is enqueued;
        the gather of what it held runs after that enqueue: the collective's kernel needs CUs, which the decode of the step in flight holds until
        its tail, and the next step's front end should already be queued for that same tail."""
        done = []
        for k in range(n):
            i = k % npipes
            pend = collect(i) if inflight[i] else None
            submit(i)
            if pend is not None:
                done.append(gather(pend))
        for k in range(n, n + npipes):                                      # drain in submission order
            i = k % npipes
            if inflight[i]:
                done.append(finish(i))
        return done

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s_ in streams:
        s_.wait_stream(torch.cuda.current_stream())
    run_steps(args.warmup)
    # set-up, untimed: every pipe object has to have run once (its workspace and the decoder's scratch are allocated on first
    # use; with W < number of pipes that first use would otherwise fall into the timed region)
    for i in range(min(args.warmup, npipes), npipes):
        submit(i); finish(i)
    sync(); t0 = time.time()
    done = run_steps(args.steps)
    sync(); dt = time.time() - t0
    # stage table: one more step, untimed and alone on the GPU (in the timed region the front end of a step shares the GPU with
    # the end of the previous step's decode, so its event intervals include waiting); the roofline entry of the dominant
    # kernel below uses the event intervals of the timed steps themselves
    serial_ms = None; serial_step_ms = None
    if npipes > 1:
        sync(); ts = time.time(); submit(0); serial_ms = finish(0)[2]; sync(); serial_step_ms = 1000.0 * (time.time() - ts)
    # ---- configs[3] as literally stated ("1k-utterance batch sharded over N GPUs"): a second timed region of the same run.  Every rank decodes the
    # utterances u = rank mod world of a 1000-utterance batch; at world = 1 that is the region above.  speedup = the weak region's step (one GPU,
    # 1000 utterances) over this one.  The decode runs one utterance per workgroup per CU (time-sliced when there are more utterances than workgroups):
    # a shard costs max(1, shard / 256) utterance-durations, so 125 utterances (8 GPUs) cost as much as 250 (4 GPUs).
    strong2 = None
    if world > 1 and not strong and args.strong_total > 0:
        ids = shard_utterances(args.strong_total, world, rank); Us = min(len(ids), U)
        if Us >= 1:
            batch[0], batch[1], batch[2] = x[:Us].contiguous(), ns_dev[:Us].contiguous(), ns_host[:Us]
            run_steps(1)                                                    # untimed: the shapes of the smaller batch
            sync(); ts = time.time(); done2 = run_steps(args.steps); sync(); dts = time.time() - ts
            tm2 = torch.tensor([dts], dtype=torch.float64, device=cdev); dist.all_reduce(tm2, op=dist.ReduceOp.MAX); dts = float(tm2.item())
            bad2 = sum(1 for res, _, _ in done2 for r in res if r.status != 0)
            strong2 = dict(total_utts=args.strong_total, utts_this_rank=Us, ms_per_step=1000.0 * dts / args.steps,
                           value=args.strong_total * args.secs / 3600.0 / (dts / args.steps), unit="audio_hours/s", failed_utts_rank0=bad2,
                           decode_rounds=round(max(1.0, Us / 256.0), 2), note="shard u -> rank u mod world; one workgroup per CU decodes an utterance: a shard of more than 256 utterances is time-sliced and "
                                "costs shard/256 utterance-durations, a smaller one costs one")
            batch[0], batch[1], batch[2] = x, ns_dev, ns_host
    stage = np.zeros(6); placements = 0; active = 0; bad = 0; frames = 0
    for res, words, sms in done:
        stage += np.array(sms)
        if os.environ.get("DSR_BENCH_VERBOSE") and rank == 0:
            print("step stage ms: " + " ".join("%.2f" % v for v in sms), file=sys.stderr, flush=True)
        placements += sum(r.placements for r in res); active += sum(r.activeHypos for r in res)
        frames += sum(r.frames + 1 for r in res); bad += sum(1 for r in res if r.status != 0)
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = 1000.0 * dt / args.steps
    audio_s = total_utts * args.secs
    hours_per_s = audio_s / 3600.0 / (dt / args.steps)

    if rank == 0:
        stage_ms = (stage / args.steps).tolist()
        names = ["analysis", "beamform", "synthesis", "mfcc", "gmm", "viterbi"]
        T_ana = mdl["ana"].frames(nsamp)
        # algorithmic bytes / flops per launch (DESIGN.md "Measurement")
        fusedFE = (not args.no_fuse) and mdl["ana"].analysis_beamform_supported(mdl["bf"])
        alg = {
            # 512 B in + 1032 B out per channel-frame.  Fused with the beamformer the snapshots stay on the chip and 8d's formula for that
            # variant applies: every sample once in, one beamformed row per frame out = C * 512 + 1032 bytes per frame (what the kernel really moves
            # -- tiles re-read the history they share -- is the counter figure next to it, not this one)
            "analysis": ("hbm", U * T_ana * (Cn * 512.0 + 1032.0)) if fusedFE else ("hbm", U * Cn * T_ana * 1544.0),
            "beamform": ("hbm", 0.0) if fusedFE else ("hbm", U * T_ana * (Cn + 1) * 129 * 8.0),
            "synthesis": ("lds", U * T_ana * (129 * 8.0 + 128 * 4.0)),          # bytes quoted for reference: bound by the LDS traffic of its FFT + overlap-add
            "mfcc": ("fp64", U * Tm * (160 * 4.0 + 39 * 4.0)),                  # bytes quoted for reference: bound by its fp64 FFT through LDS (feature.cc is double)
            "gmm": ("mfma" if args.gmm_mode == 2 else "valu", U * Tm * 4.0 * 39 * 4096),
            "viterbi": ("hbm", placements / args.steps * 40.0),                 # 20 B arc + 4 B score + 16 B token per expanded arc
        }
        dom = int(np.argmax(stage_ms)); dn = names[dom]; kind, amount = alg[dn]
        secs = stage_ms[dom] / 1000.0
        if kind in ("hbm", "lds", "fp64"):
            roof = dict(kernel="k_" + dn, bound="hbm", achieved=amount / secs / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        else:
            roof = dict(kernel="k_" + dn, bound="mfma", achieved=amount / secs / 1e12, peak=MFMA_F32_PEAK_TF, unit="TFLOP/s")
        roof["frac"] = roof["achieved"] / roof["peak"]; roof["traffic"] = None
        # HBM bytes per launch from the counter passes of this round (profiles/r01_traffic.json: FETCH_SIZE and WRITE_SIZE, separate
        # rocprofv3 --pmc runs of this same workload); only quoted when the workload is the one they were collected on
        try:
            tj_path = next(pth for pth in (os.path.join(ROOT, "profiles", "r0%d_traffic.json" % r) for r in (3, 2, 1)) if os.path.exists(pth))
            tj = json.load(open(tj_path))
            key = {"viterbi": "k_viterbi", "analysis": "k_analysis_q256", "beamform": "k_bf_apply"}.get(dn)
            # the counters were collected on this workload shape at the beam recorded in the file; the tuned beam of a run moves in its fourth digit
            same = U == 1000 and args.secs == 10.0 and args.states == 50000 and abs(beam - float(tj.get("beam", 53.787))) < 0.01 * beam
            if key in tj["kernels"] and same:
                roof["traffic"] = tj["kernels"][key]["bytes_per_launch"]
                roof["traffic_note"] = "bytes per launch, PMC FETCH_SIZE + WRITE_SIZE (profiles/%s, collected at beam %.3f)" % (os.path.basename(tj_path), float(tj.get("beam", 53.787)))
        except (OSError, ValueError, KeyError):
            pass
        roof["launch_ms"] = stage_ms[dom]
        stages = {}
        tbl_ms = serial_ms if serial_ms is not None else stage_ms
        for i, nm in enumerate(names):
            if nm == "beamform" and fusedFE:
                continue                                                # no such kernel in this mode: the analysis entry covers it
            k, a = alg[nm]; s = tbl_ms[i] / 1000.0
            byteq = k in ("hbm", "lds", "fp64")
            stages[nm] = dict(ms=round(tbl_ms[i], 3), bound=k,
                              achieved=round(a / s / (1e9 if byteq else 1e12), 3) if s > 0 else None,
                              unit=("GB/s" if k == "hbm" else "GB/s of HBM bytes (not the bound)") if byteq else "TFLOP/s")
            if k == "hbm" and s > 0:
                stages[nm]["frac_of_hbm_peak"] = round(a / s / 1e9 / HBM_PEAK_GBS, 3)
            if nm == "analysis" and fusedFE and s > 0:
                stages[nm]["bytes"] = "SURVEY 8d, fused variant: (C * 512 + 1032) B per frame"
                try:                                                    # what the counters saw for this kernel (separate --pmc passes of this workload)
                    kb = tj["kernels"].get("k_analysis_bf_q256")
                    if kb and same:
                        stages[nm]["counter_GBs"] = round(kb["bytes_per_launch"] / s / 1e9, 1); stages[nm]["counter_over_algorithmic"] = round(kb["bytes_per_launch"] / a, 2)
                except NameError:
                    pass
        gmm_check = None
        if args.gmm_mode == 2 and world == 1 and not args.no_verify:
            # the MFMA scoring path against the exact one on this very batch: same decoder, same features, scores from mode 0
            dec0 = dsr.Decoder(beam=beam, lmScale=12.0, maxActive=65536); dec0.set(mdl["gd"])
            p0 = dsr.Pipe(mdl["ana"], mdl["syn"], mdl["bf"], dsr.Mfcc(lda=mdl["lda"]), mdl["gm"], dec0, gmmMode=0, fused=not args.no_fuse)
            r0, _, w0 = p0.run(x, ns_dev, ns_host, maxPath=maxPath, want_paths=True)
            res2, words2, _ = done[-1]
            same1 = sum(1 for u in range(U) if r0[u].nWords == res2[u].nWords and np.array_equal(w0[u, :r0[u].nWords], words2[u, :res2[u].nWords]))
            sdiff = max(abs(r0[u].score - res2[u].score) / max(1.0, abs(r0[u].score)) for u in range(U))
            gmm_check = dict(one_best_agree="%d/%d" % (same1, U), max_rel_score_diff=float(sdiff),
                             note="GMM mode 2 (MFMA, argmin = mode 0 on every frame, cost rel <= 2e-6) against mode 0 (reference order, bit exact) through the same decoder")
            del p0, dec0
        cpu = None
        if not args.no_cpu and world == 1:                               # the CPU baseline is a 1-GPU-run item (rank 0 at N = 1 only)
            try:
                ncores = len(os.sched_getaffinity(0))
            except AttributeError:
                ncores = os.cpu_count() or 1
            nthr = args.cpu_threads if args.cpu_threads > 0 else min(16, ncores)
            ncpu = min(36, U)                                           # ~0.31 s per utterance and core: about 11 s single-threaded
            xh = x[:min(U, max(ncpu, 4 * nthr))].cpu().numpy()         # N-core leg: four utterances per core
            done, cdt, cwords, nat = cpu_baseline(mdl, xh[:ncpu], nsamp, beam, max_seconds=12.0, threads=1)
            cpu = dict(value=done * args.secs / 3600.0 / cdt, unit="audio_hours_per_sec", cores=1, kind="port",
                       sample="%d whole utterances (%.0f s x 8 ch) through the same pipe, oracle C restatement%s, %.1f s of CPU time"
                              % (done, args.secs, " -O3 -march=native" if nat else "", cdt),
                       xRT=done * args.secs / cdt)
            # the same utterances decoded on the GPU give the same word sequences? (front end differs by fp32 rounding)
            agree = sum(1 for u in range(done) if cwords[u] is not None and np.array_equal(cwords[u], words[u, :res[u].nWords]))
            cpu["one_best_agree"] = "%d/%d" % (agree, done)
            # N-core leg: utterance-parallel over the host cores this process may use, one utterance per core
            if nthr > 1 and xh.shape[0] >= 2:
                nd, ndt, _, _ = cpu_baseline(mdl, xh, nsamp, beam, max_seconds=12.0, threads=min(nthr, xh.shape[0]))
                cpu["n_core"] = dict(value=nd * args.secs / 3600.0 / ndt, unit="audio_hours_per_sec", cores=min(nthr, xh.shape[0]), host_cores=ncores,
                                     xRT=nd * args.secs / ndt, sample="%d utterances over %d worker threads (oracle C code, GIL released), %.1f s wall" % (nd, min(nthr, xh.shape[0]), ndt))
            cpu["config1"] = cpu_config1()
        line = dict(metric="decoded_audio_hours_per_sec", value=hours_per_s, unit="audio_hours/s", n_gpus=world, steps=args.steps,
                    warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling="strong" if strong else "weak", vs_baseline=None,
                    dtype="f32", data="synthetic",
                    config=dict(workload="full pipe: %d utt/GPU x %.0f s x 8 ch, analysis M=256 m=4 r=1 -> MVDR -> synthesis -> MFCC(39) -> "
                                         "GMM %d dists x %d Gaussians -> WFST %d states/%d arcs, lmScale 12, beam %.1f"
                                         % (U, args.secs, args.dists, 4096 // args.dists, args.states, mdl["nArcs"], beam),
                                utts_per_gpu=U, total_utts=total_utts,
                                scaling_mode=("strong: --total-utts %d sharded u -> rank u mod %d (BASELINE configs[3])" % (total_utts, world)) if strong
                                else "weak: %d utterances per GPU (at 1 GPU this is BASELINE configs[3]'s 1k-utterance batch)" % U,
                                xRT=audio_s / (dt / args.steps), beam=beam,
                                mean_active_tokens=active / max(1, frames), failed_utts=bad, gmm_mode=args.gmm_mode,
                                parallelism="utterance-sharded x%d, RCCL gather of 1-best" % world,
                                front_end="analysis bank + MVDR in one kernel (channel snapshots never written; stage 'analysis' covers both, bytes = samples in x 23/16 + beamformed rows out)" if fusedFE
                                else "analysis bank and beamformer as two kernels",
                                step_overlap=("two pipe objects on two HIP streams: step k+1 is enqueued while step k decodes, its front end runs on the CUs step k's persistent "
                                              "decode workgroups free as utterances finish; `stages` and serial_step_ms are one extra, untimed step alone on the GPU "
                                              "(their sum exceeds ms_per_step by what the overlap hides); --serial runs the steps one after the other") if npipes > 1 else "none",
                                serial_step_ms=serial_step_ms, strong=strong2, gmm_mode2_vs_mode0=gmm_check),
                    roofline=roof, stages=stages, cpu_baseline=cpu)
        print(json.dumps(line))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
