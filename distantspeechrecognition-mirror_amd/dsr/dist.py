"""The path's only exchange step: gather the decoded 1-best word-id sequences of every rank on rank 0.

Utterances are sharded over ranks (one process per GPU, no data-path collective); after a batch each rank holds
ragged int32 word sequences.  One max-length all-gather (1 int per rank) + one padded gather (KBs) move them to
rank 0 -- RCCL over xGMI on the GPUs (backend "nccl"), gloo in the CPU tests.
"""
import numpy as np


def shard_utterances(n_utts, world, rank):
    """Utterance ids of `rank` (round robin: u -> rank u mod world)."""
    return list(range(rank, n_utts, world))


def gather_one_best(words, n_words, world, rank, device, dist=None):
    """words: np.uint32/int32 [U][maxPath], n_words: [U].  Returns on rank 0 a list (per rank) of lists (per local
    utterance) of int word ids; None elsewhere.  With world == 1 no communication happens."""
    import torch
    n_words = np.asarray(n_words, np.int32)
    U = len(n_words)
    if world == 1:
        return [[words[u, :n_words[u]].astype(np.int64).tolist() for u in range(U)]]
    wmax_loc = torch.tensor([int(n_words.max()) if U else 0, U], dtype=torch.int32, device=device)
    allmax = [torch.zeros_like(wmax_loc) for _ in range(world)]
    dist.all_gather(allmax, wmax_loc)
    wmax = max(int(t[0].item()) for t in allmax)
    umax = max(int(t[1].item()) for t in allmax)
    pad = torch.zeros((umax, wmax + 1), dtype=torch.int32, device=device)
    if U:
        pad[:U, 0] = torch.from_numpy(n_words).to(device)
        if wmax:
            pad[:U, 1:] = torch.from_numpy(np.ascontiguousarray(words[:, :wmax]).astype(np.int32)).to(device)
    out = [torch.zeros_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, out, dst=0)
    if rank != 0:
        return None
    res = []
    for r in range(world):
        o = out[r].cpu().numpy(); nu = int(allmax[r][1].item())
        res.append([o[u, 1:1 + o[u, 0]].astype(np.int64).tolist() for u in range(nu)])
    return res


def gather_lattices(images, world, rank, device, dist=None):
    """The lattice half of the exchange step (north star: "gather decoded lattices/1-best").  images: list (per local utterance) of np.uint8
    arrays = dsr.Lattice.pack() (flat image: nodes, edges in creation order, frames, ac/lm doubles).  Returns on rank 0 a list (per rank) of lists
    (per local utterance) of np.uint8 images -- dsr.Lattice.unpack() gives the lattice back -- and None elsewhere.  Two collectives per batch:
    one all-gather of (total bytes, utterance count) and one padded byte gather; lattices are tens of KB per utterance, the xGMI links idle."""
    import torch
    U = len(images)
    if world == 1:
        return [[np.ascontiguousarray(im, np.uint8) for im in images]]
    lens = np.array([len(im) for im in images], np.int64)
    head = torch.tensor([int(lens.sum()), U], dtype=torch.int64, device=device)
    allhead = [torch.zeros_like(head) for _ in range(world)]
    dist.all_gather(allhead, head)
    bmax = max(int(t[0].item()) for t in allhead); umax = max(int(t[1].item()) for t in allhead)
    # per rank: [umax] int64 lengths as bytes, then the images back to back, padded to the longest rank
    buf = np.zeros(8 * umax + bmax, np.uint8)
    buf[:8 * U] = lens.view(np.uint8)
    if U and lens.sum():
        buf[8 * umax:8 * umax + int(lens.sum())] = np.concatenate([np.ascontiguousarray(im, np.uint8) for im in images])
    t = torch.from_numpy(buf).to(device)
    out = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, out, dst=0)
    if rank != 0:
        return None
    res = []
    for r in range(world):
        o = out[r].cpu().numpy(); nu = int(allhead[r][1].item())
        ln = o[:8 * nu].view(np.int64); off = 8 * umax; ims = []
        for u in range(nu):
            ims.append(o[off:off + int(ln[u])].copy()); off += int(ln[u])
        res.append(ims)
    return res
