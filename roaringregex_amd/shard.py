"""Sharding of a '\\n'-delimited corpus over the GPUs of one node.

Lines are independent and the compiled automaton is a few KB, so the path shards with NO exchange step: the
corpus is cut into `world` byte ranges of near-equal size whose boundaries fall right after a '\\n', every rank
scans its own range with its own replica of the tables, and the results are simply concatenated in rank order
(or their counters summed).  torch.distributed is needed only to move the small results, never the text.
"""
import numpy as np


def line_aligned_ranges(data, world):
    """[(start, end)] * world: contiguous, covering data, each start at a line start, sizes within one line of
    len(data)/world.  `data` is anything numpy can view as uint8 (array, memmap)."""
    a = np.asarray(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    n = a.size
    cuts = [0]
    for r in range(1, world):
        target = max(cuts[-1], (n * r) // world)
        if target >= n:
            cuts.append(n)
            continue
        # first line start at or after `target`
        if target == 0 or a[target - 1] == 10:
            cuts.append(target)
            continue
        window = 1 << 16
        pos = target
        found = n
        while pos < n:
            nl = np.flatnonzero(a[pos:pos + window] == 10)
            if nl.size:
                found = pos + int(nl[0]) + 1
                break
            pos += window
        cuts.append(min(found, n))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def line_aligned_cuts(fetch, n, world, window=1 << 16):
    """The same cuts for a corpus that is not in one array: fetch(lo, hi) -> the bytes [lo, hi) as uint8 (a file, a memory map,
    a generator).  Only the neighbourhood of the world - 1 targets is read.  -> [(start, end)] * world."""
    cuts = [0]
    for r in range(1, world):
        target = max(cuts[-1], (n * r) // world)
        if target >= n:
            cuts.append(n)
            continue
        if target == 0 or int(np.asarray(fetch(target - 1, target), dtype=np.uint8)[0]) == 10:
            cuts.append(target)
            continue
        pos, found = target, n
        while pos < n:
            nl = np.flatnonzero(np.asarray(fetch(pos, min(n, pos + window)), dtype=np.uint8) == 10)
            if nl.size:
                found = pos + int(nl[0]) + 1
                break
            pos += window
        cuts.append(min(found, n))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def lines_in(a):
    """Number of strings in a shard (a trailing fragment without '\\n' counts)."""
    a = np.asarray(a, dtype=np.uint8)
    return int((a == 10).sum()) + (1 if a.size and a[-1] != 10 else 0)


def match_sharded(match_fn, data, rank, world, group=None):
    """Run match_fn(shard_bytes) -> accept vector (uint8: a numpy array or a torch tensor, on any device) on this rank's shard
    and all-gather the per-rank vectors (variable length) so that every rank returns the accept vector of the WHOLE corpus
    (numpy).  With world == 1 or no initialised process group this is just match_fn(data).  The only things that travel are
    one line count and one accept vector per rank - never the text."""
    ranges = line_aligned_ranges(data, world)
    s, e = ranges[rank]
    mine = match_fn(np.asarray(data[s:e]))
    is_tensor = hasattr(mine, "is_cuda")
    if world == 1:
        return mine.cpu().numpy() if is_tensor else np.ascontiguousarray(mine, dtype=np.uint8)
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine_t = (mine if is_tensor else torch.from_numpy(np.ascontiguousarray(mine, dtype=np.uint8))).to(dev)
    n_mine = torch.tensor([mine_t.numel()], dtype=torch.int64, device=dev)
    every = [torch.zeros_like(n_mine) for _ in range(world)]
    dist.all_gather(every, n_mine, group=group)
    counts = [int(t.item()) for t in every]
    cap = max(counts + [1])
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    buf[:mine_t.numel()] = mine_t
    gathered = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    return np.concatenate([g[:c].cpu().numpy() for g, c in zip(gathered, counts)])
