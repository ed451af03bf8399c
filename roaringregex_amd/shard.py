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


def lines_in(a):
    """Number of strings in a shard (a trailing fragment without '\\n' counts)."""
    a = np.asarray(a, dtype=np.uint8)
    return int((a == 10).sum()) + (1 if a.size and a[-1] != 10 else 0)


def match_sharded(match_fn, data, rank, world, group=None):
    """Run match_fn(shard_bytes) -> uint8 accept vector on this rank's shard and all-gather the per-rank vectors
    (variable length) so that every rank returns the accept vector of the WHOLE corpus.  With world == 1 or no
    initialised process group this is just match_fn(data)."""
    ranges = line_aligned_ranges(data, world)
    s, e = ranges[rank]
    mine = np.ascontiguousarray(match_fn(np.asarray(data[s:e])), dtype=np.uint8)
    if world == 1:
        return mine
    import torch
    import torch.distributed as dist
    counts = [lines_in(np.asarray(data[a:b])) for a, b in ranges]
    assert counts[rank] == mine.size
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    cap = max(counts + [1])
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    buf[:mine.size] = torch.from_numpy(mine).to(dev)
    gathered = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    return np.concatenate([g[:c].cpu().numpy() for g, c in zip(gathered, counts)])
