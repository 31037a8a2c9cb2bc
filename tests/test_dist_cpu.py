"""world_size-2 gloo test of the path's only exchange (1-best gather) and of the utterance sharding."""
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
