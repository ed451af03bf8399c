"""N > 1 path on the CPU: line-aligned shards + rank-order concatenation reproduce the unsharded result.
world_size 2 over gloo (127.0.0.1).  The per-shard matcher here is the ORACLE (this is a CPU test of the
sharding logic; on the GPU the same match_sharded() is driven with RRegex.match_corpus)."""
import os
import socket

import numpy as np
import pytest

from roaringregex_amd.shard import line_aligned_cuts, line_aligned_ranges, lines_in

torch = pytest.importorskip("torch")


def test_ranges_are_line_aligned_and_cover():
    rng = np.random.default_rng(0)
    for n in (0, 1, 5, 1000, 70000):
        a = rng.choice(np.frombuffer(b"ab\n", dtype=np.uint8), size=n, p=[0.48, 0.48, 0.04]) if n else np.empty(0, np.uint8)
        for world in (1, 2, 3, 8):
            r = line_aligned_ranges(a, world)
            assert len(r) == world and r[0][0] == 0 and r[-1][1] == n
            for (s, e), (s2, _) in zip(r, r[1:]):
                assert e == s2 and s <= e
            for s, _ in r:
                assert s == 0 or s == n or a[s - 1] == 10
            assert sum(lines_in(a[s:e]) for s, e in r) == lines_in(a)
            # the same cuts through a byte fetcher (a corpus that is not in one array: bench.py --shard-one-corpus)
            assert line_aligned_cuts(lambda lo, hi: a[lo:hi], n, world, window=97) == r
    # no newline at all: everything stays on rank 0's... first shard that reaches the end
    a = np.frombuffer(b"x" * 1000, dtype=np.uint8)
    r = line_aligned_ranges(a, 4)
    assert sum(e - s for s, e in r) == 1000 and sum(1 for s, e in r if e > s) == 1


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tools"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import synth
    from pyoracle import OracleRegex
    from roaringregex_amd.shard import match_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pattern = r"[A-Za-z0-9._]+@[A-Za-z0-9.]+"
        data = synth.corpus("email", 5, 300_000, threads=1)[:-7]       # unterminated last line on purpose
        o = OracleRegex(pattern)
        whole = match_sharded(lambda shard: o.match_lines(shard), data, rank, world)
        want = o.match_lines(data)
        ok = whole.shape == want.shape and bool((whole == want).all())
        # counters can also be reduced instead of gathered
        t = torch.tensor([int(whole.sum())], dtype=torch.int64)
        dist.all_reduce(t)
        q.put((rank, ok, int(t.item()), int(want.sum()) * world))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_match_equals_unsharded():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, total, want_total in res:
        assert ok, rank
        assert total == want_total
