"""world_size-2 gloo tests of the path's only exchange (1-best gather, lattice gather) and of the utterance sharding."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "distantspeechrecognition-mirror_amd"))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dsr.dist import gather_one_best, shard_utterances
    ids = shard_utterances(7, world, rank)
    rng = np.random.default_rng(100)
    allw = [rng.integers(1, 5000, size=int(rng.integers(0, 9))).astype(np.uint32) for _ in range(7)]
    words = np.zeros((len(ids), 16), np.uint32); nw = np.zeros(len(ids), np.int32)
    for i, u in enumerate(ids):
        nw[i] = len(allw[u]); words[i, :nw[i]] = allw[u]
    got = gather_one_best(words, nw, world, rank, torch.device("cpu"), dist)
    if rank == 0:
        ok = True
        for r in range(world):
            for i, u in enumerate(shard_utterances(7, world, r)):
                ok = ok and got[r][i] == allw[u].astype(np.int64).tolist()
        q.put(ok)
    else:
        q.put(got is None)
    dist.barrier(); dist.destroy_process_group()


def test_gather_one_best_gloo_world2():
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in ps:
        p.join(60)
    assert all(res)


def test_shard_covers_all():
    from dsr.dist import shard_utterances
    for world in (1, 2, 4, 8):
        ids = sorted(sum((shard_utterances(1000, world, r) for r in range(world)), []))
        assert ids == list(range(1000))
        assert max(len(shard_utterances(1000, world, r)) for r in range(world)) - min(len(shard_utterances(1000, world, r)) for r in range(world)) <= 1


def test_single_rank_no_comm():
    from dsr.dist import gather_one_best
    w = np.array([[3, 4, 0], [9, 0, 0]], np.uint32)
    assert gather_one_best(w, [2, 1], 1, 0, torch.device("cpu")) == [[[3, 4], [9]]]


def _fake_lattice_image(rng):
    """a lattice image in the layout of LatticeData::pack (csrc/lattice.cpp): [magic, nNodes, nEdges, finalStatesN] int32, nodeFinal, then the edge arrays"""
    nN, nE = int(rng.integers(2, 40)), int(rng.integers(1, 200))
    hdr = np.array([0x4C415431, nN, nE, 1], np.int32)
    parts = [hdr, rng.integers(0, 2, nN).astype(np.int32)] + [rng.integers(0, nN, nE).astype(np.int32) for _ in range(2)] + \
            [rng.integers(0, 50, nE).astype(np.uint32) for _ in range(2)] + [rng.integers(0, 90, nE).astype(np.int32) for _ in range(2)] + \
            [rng.standard_normal(nE), rng.standard_normal(nE)]
    return np.concatenate([np.ascontiguousarray(p).view(np.uint8) for p in parts])


def _lat_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dsr.dist import gather_lattices, shard_utterances
    rng = np.random.default_rng(7)
    allim = [_fake_lattice_image(rng) for _ in range(9)]
    ids = shard_utterances(9, world, rank) if rank == 0 else shard_utterances(9, world, rank)[:1]      # ragged: rank 1 holds a single utterance
    got = gather_lattices([allim[u] for u in ids], world, rank, torch.device("cpu"), dist)
    if rank == 0:
        import dsr._capi as K
        ok = True
        for r in range(world):
            rid = shard_utterances(9, world, r) if r == 0 else shard_utterances(9, world, r)[:1]
            ok = ok and len(got[r]) == len(rid)
            for i, u in enumerate(rid):
                ok = ok and np.array_equal(got[r][i], allim[u])
                L = K.Lattice.unpack(got[r][i])                            # the C-ABI reads the image back (host side, no GPU needed)
                ok = ok and np.array_equal(L.pack(), allim[u]) and len(L.data["from"]) == int(allim[u][8:12].view(np.int32)[0])
        q.put(ok)
    else:
        q.put(got is None)
    dist.barrier(); dist.destroy_process_group()


def test_gather_lattices_gloo_world2():
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    port = 29950 + (os.getpid() % 400)
    ps = [ctx.Process(target=_lat_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in ps:
        p.join(60)
    assert all(res)


def test_gather_lattices_single_rank():
    from dsr.dist import gather_lattices
    im = [np.arange(5, dtype=np.uint8), np.zeros(0, np.uint8)]
    got = gather_lattices(im, 1, 0, torch.device("cpu"))
    assert len(got) == 1 and np.array_equal(got[0][0], im[0]) and len(got[0][1]) == 0
