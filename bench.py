#!/usr/bin/env python3
"""bench.py — GB/s of input scanned by the RoaringRegex hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus N ...          # N > 1 without WORLD_SIZE: starts the N ranks itself (child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (rrx_match_corpus: one kernel launch) over one resident batch: the
256-state-class config of BASELINE.json (configs[2]): the URL regex U2 (226 reference states, BitSet<4> class)
over 8 GiB of synthetic URL log lines per GPU.  The corpus and its newline index are resident in HBM before
the timed region.  N > 1: every rank scans its own 8 GiB shard (weak scaling), no collective on the data path;
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0 with the contract's fields plus "roofline" and "cpu_baseline".
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6400.0  # what a plain coalesced read stream reaches on these boxes (tools/probe/readpeak.hip,
                             # profiles/r01_readpeak_coalesced_read_stream.txt; the guide quotes 6.29 TB/s for a float4 copy)

WORKLOADS = {
    # name: (synth kind, pattern key, default bytes per GPU, BASELINE config it stands for)
    "url": ("url", "U2", 8 << 30, "configs[2]: URL regex U2"),
    "email": ("email", "EMAIL", 1 << 30, "configs[1]: email regex"),
    "arepeat": ("arepeat", "A300", 1 << 30, "configs[3]: a{1,300} (4 GiB over 4 GPUs = 1 GiB per GPU)"),
    "kwlines": ("kwlines", "K1000", 8 << 30, "configs[4](i): k1|...|k1000 over lines k<n> (64 GiB over 8 GPUs = 8 GiB per GPU)"),
    "kwlog": ("kwlog", "K1000C", 8 << 30, "configs[4](ii): .*(k1|...|k1000).* over log lines (8 GiB per GPU)"),
    # "also report a long-line variant (>= 256 B/line) for C2/C3"
    "email_long": ("email_long", "EMAIL", 1 << 30, "configs[1], long-line variant: email regex over lines of 290-600 bytes (mean 437)"),
    "url_long": ("url_long", "U2", 8 << 30, "configs[2], long-line variant: URL regex U2 over lines of 390-740 bytes (mean 575)"),
    # not a BASELINE config: the pattern class only an NFA engine can run (its subset construction has 2^41 sets)
    "nondet": ("ablines", "NONDET", 8 << 30, "extra: (a|b)*a(a|b){40} over lines of a/b (no DFA exists within memory)"),
    "nondet600": ("ablong", "NONDET600", 1 << 30, "extra: (a|b)*a(a|b){600} over lines of 500-900 a/b (604 positions, group-cooperative NFA)"),
    "nondet5000": ("ablong", "NONDET5000", 1 << 28, "extra: (a|b)*a(a|b){5000} over lines of 500-900 a/b (5003 positions, wave-resident NFA)"),
    # automata that do not determinise over text whose live sets are few: the sampled table
    "urltail": ("url", "U2TAIL", 8 << 30, "extra: U2(x|y)*x(x|y){30} over the URL corpus (no DFA within memory: 2^31 sets on x/y text; sampled table)"),
    "urlalt": ("url", "U2ALT", 8 << 30, "extra: (U2)|(x|y)*x(x|y){30} over the URL corpus (no DFA within memory; sampled table)"),
    # search with a product table beyond the LDS (8193 rows): the stripe-wise search kernel's global form (--search)
    "abx": ("ablines", "ABX", 1 << 30, "extra: [ab]*a[ab]{11}x over lines of a/b (search tables beyond the LDS)"),
    # the same automata over SHORT lines (30-120 bytes): only the first block of 2048 positions is ever live - sparse sets
    "short5000": ("ablines", "NONDET5000", 1 << 28, "extra: (a|b)*a(a|b){5000} over lines of 30-120 a/b (sparse live sets)"),
    "short16000": ("ablines", "NONDET16000", 1 << 28, "extra: (a|b)*a(a|b){16000} over lines of 30-120 a/b (16003 positions, sparse live sets)"),
}


ENGINES = {"auto": "ENGINE_AUTO", "nfa": "ENGINE_NFA", "dfa": "ENGINE_DFA", "dfa2": "ENGINE_DFA2", "wave": "ENGINE_NFA_WAVE",
           "dfa-global": "ENGINE_DFA_GLOBAL", "block": "ENGINE_NFA_BLOCK", "sparse": "ENGINE_NFA_SPARSE"}


def patterns():
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        kat = json.load(f)
    u2 = [k["pattern"] for k in kat["kat"] if k["pattern"].startswith("(http|https|ftp)")][0]
    k1000 = kat["big_states"][-1]["pattern"]
    return {"U2": u2, "EMAIL": r"[A-Za-z0-9._]+@[A-Za-z0-9.]+", "A300": "a{1,300}", "K1000": k1000, "K1000C": ".*(" + k1000 + ").*",
            "ABX": "[ab]*a[ab]{11}x", "U2TAIL": u2 + "(x|y)*x(x|y){30}", "U2ALT": "(" + u2 + ")|(x|y)*x(x|y){30}",
            "NONDET": "(a|b)*a(a|b){40}", "NONDET600": "(a|b)*a(a|b){600}", "NONDET5000": "(a|b)*a(a|b){5000}", "NONDET16000": "(a|b)*a(a|b){16000}"}


def traffic_from_profile(workload, nbytes, engine_name):
    """HBM bytes per launch as counted by an EARLIER rocprofv3 PMC run committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE in separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot
    read PMC counters itself, so `roofline.traffic` (= measured in this run) stays null and this figure is reported
    beside it, with the file it comes from, only for the exact configuration that was profiled."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    for row in table:
        if row["workload"] == workload and row["bytes"] == nbytes and row["engine"] == engine_name:
            return {"bytes": int(row["fetch_kb"] * 1024 * 2 + row["write_kb"] * 1024), "file": row["file"], "kernel": row.get("kernel")}
    return None


def launch_ranks(n, argv):
    """`bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks as CHILD processes (one
    torch.distributed.run, which starts one process per GPU) and relay their exit code.  This runs before torch is
    imported and before anything touches the GPU in this process; nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def host_cores():
    """CPU share of this process: the cgroup quota if there is one, else the affinity mask, capped at 16
    (a one-GPU box hands out 16 cores even though it shows all of the host's)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 16))


def cpu_baseline(pattern, host, target_seconds=12.0):
    """The oracle (CPU port of the reference's loop, NFA.cc:86-100) on a bounded sample of the same corpus,
    one oracle instance per thread (the reference is not re-entrant either), all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyoracle import OracleRegex, simd
    cores = host_cores()
    chunk = 1 << 20                                    # corpus chunks end with '\n': line-aligned slices
    # calibrate on 4 MiB, then size the sample for ~target_seconds
    o = OracleRegex(pattern)
    t0 = time.perf_counter()
    o.match_lines(host[:4 * chunk])
    per_core = 4 * chunk / (time.perf_counter() - t0)
    per_thread = int(min(len(host) // cores, max(4 * chunk, per_core * target_seconds)) // chunk * chunk)
    oracles = [OracleRegex(pattern) for _ in range(cores)]
    accepted = [0] * cores

    def work(i):
        accepted[i] = int(oracles[i].match_lines(host[i * per_thread:(i + 1) * per_thread]).sum())

    threads = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    total = per_thread * cores
    return {"value": round(total / dt / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": "first %d MiB of the same corpus, %d threads x %d MiB, oracle/rr_oracle.c (-O2; BitSet<2>/<4> step with %s)"
                      % (total >> 20, cores, per_thread >> 20, "the reference's vector ORs, SSE2 / AVX2 (BitSet.cc:8-21)" if simd() else "scalar words"),
            "single_core_GBs": round(per_core / 1e9, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="url", choices=sorted(WORKLOADS))
    ap.add_argument("--bytes", type=int, default=0, help="bytes per GPU (default: the BASELINE size)")
    ap.add_argument("--engine", default="auto", choices=sorted(ENGINES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stripe", type=int, default=0, help="bytes per GPU lane (0 = chosen from the corpus size)")
    ap.add_argument("--pcie", action="store_true", help="also time the host-buffer entry (upload + index + match + download)")
    ap.add_argument("--search", action="store_true", help="also time rrx_search_corpus (match offsets per line) on the same corpus")
    ap.add_argument("--shard-one-corpus", action="store_true",
                    help="ONE corpus of gpus x bytes, cut after '\\n' into one byte range per rank (roaringregex_amd.shard) - SURVEY 8(e) to the "
                         "letter; default: every rank generates a corpus of its own (the same work per GPU)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # N > 1 and nobody started the ranks for us: do it ourselves, as child processes, before any GPU call here.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launch and the flag disagree" % (args.gpus, world))

    # RRX_BENCH_REHEARSAL=1: all ranks share GPU 0 and talk over gloo — a dry run of the N > 1 control flow on a
    # one-GPU box (numbers from it mean nothing).  RRX_BENCH_REHEARSAL=launcher: no device work at all — only the
    # launcher, the rendezvous and the rank proof run (CPU test of the control flow; "value" is null).
    rehearsal = os.environ.get("RRX_BENCH_REHEARSAL", "")
    if rehearsal == "launcher":
        import torch
        import torch.distributed as dist
        ranks_seen = 1
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            ranks_seen = int(t.item())
            dist.destroy_process_group()
        if ranks_seen != args.gpus:
            raise SystemExit("rank proof failed: %d ranks answered, --gpus %d" % (ranks_seen, args.gpus))
        if rank == 0:
            print(json.dumps({"metric": "GB/s input scanned (whole job)", "value": None, "unit": "GB/s", "n_gpus": world,
                              "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "rehearsal": "launcher-only"}), flush=True)
        return

    import numpy as np
    import torch
    import roaringregex_amd as rr
    import synth

    assert torch.cuda.is_available(), "bench.py needs a MI355X"
    if rehearsal == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        if rehearsal == "1":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # rank proof: an all-reduce of ones over the process group that will also carry the barrier (RCCL on a real run)
        t = torch.ones(1, dtype=torch.int64, device="cpu" if rehearsal == "1" else "cuda")
        dist.all_reduce(t)
        ranks_seen = int(t.item())
        if ranks_seen != args.gpus:
            raise SystemExit("rank proof failed: %d ranks answered, --gpus %d" % (ranks_seen, args.gpus))

    kind, pkey, default_bytes, config_name = WORKLOADS[args.workload]
    nbytes = args.bytes or default_bytes
    pattern = patterns()[pkey]
    regex = rr.RRegex(pattern, getattr(rr, ENGINES[args.engine]), device=local_rank)

    # ---- synthetic shard of this rank (seed differs per rank), generated on the host cores, then resident in HBM
    t0 = time.perf_counter()
    threads = max(1, min(len(os.sched_getaffinity(0)), 64) // max(1, world))
    sharding = "by lines, no collective"
    if args.shard_one_corpus:
        # one corpus of world x nbytes bytes (seed 2), never materialised whole: every rank finds the cuts from the bytes around the
        # world - 1 targets and generates its own range only (the generator's 1 MiB chunks are independent of each other)
        from roaringregex_amd.shard import line_aligned_cuts
        MiB = 1 << 20
        total = world * nbytes

        def fetch(lo, hi):
            first = lo // MiB * MiB
            buf = np.empty((hi - first + MiB - 1) // MiB * MiB, dtype=np.uint8)
            synth.fill_window(kind, 2, first, buf, threads=1)
            return buf[lo - first:hi - first]

        lo, hi = line_aligned_cuts(fetch, total, world)[rank]
        assert lo % MiB == 0, "the generator's chunks end with a newline: a cut at a multiple of a chunk stays there"
        host = np.empty(hi - lo, dtype=np.uint8)
        synth.fill_window(kind, 2, lo, host, threads=threads)
        nbytes = hi - lo
        sharding = "one corpus of %d bytes cut after '\\n' (shard.line_aligned_cuts), rank r scans range r; no collective" % total
    else:
        host = np.empty(nbytes, dtype=np.uint8)
        synth.fill(kind, 2 + 1000 * rank, host, threads=threads)
    gen_s = time.perf_counter() - t0
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    piece = 1 << 30
    for off in range(0, nbytes, piece):
        dev[off:off + piece].copy_(torch.from_numpy(host[off:off + piece]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    corpus = rr.Corpus(dev, stripe=args.stripe)
    torch.cuda.synchronize()
    index_ms = (time.perf_counter() - t0) * 1e3
    nlines = corpus.num_lines
    out = torch.empty((nlines + 31) // 32 + 4, dtype=torch.int32, device="cuda")  # accept bitmap, 1 bit per line

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, like the index above: the first launches upload the tables, load the code object and bring the GPU out of
    # its idle clocks (a cold kernel measures 5-7 % low whatever --warmup says); then the W warm-up steps of the contract
    for _ in range(8):
        regex.match_corpus_bits(corpus, out=out)
    torch.cuda.synchronize()
    # (a stride-2 table that is not replicated is being ordered by a background thread since the first match - tens of ms of
    # host work, DESIGN.md 4.7 - and swapped in when done: the steady state this bench measures begins then)
    t_wait = time.perf_counter()
    while (regex.table_order_pending or regex.sampled_table_pending) and time.perf_counter() - t_wait < 2.0:
        time.sleep(0.01)
    order_wait_ms = (time.perf_counter() - t_wait) * 1e3
    # one-shot ("cold") rate: a corpus nobody has indexed yet — index build + one match, clocks warm, HIP events
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    cold = []
    for _ in range(3):
        e0.record()
        c2 = rr.Corpus(dev, stripe=args.stripe)       # no hint: what a first-time caller of rrx_corpus_create pays
        e1.record()
        regex.match_corpus_bits(c2, out=out)
        e2.record()
        torch.cuda.synchronize()
        cold.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
        del c2
    cold_index_ms, cold_match_ms = min(cold, key=lambda p: p[0] + p[1])
    # the same through the one-shot entry (rrx_match_device): with the stride-2 table the text is read once, the index is
    # a by-product (per-stripe counts + scan + compaction of the lanes' verdict streams)
    # (the entry is synchronous: it returns the line count.  Every call is listed; reported: the MEDIAN of the twelve and
    # the worst call over the median - round 2 reported a mean, and one 10.6 ms call in twelve halved the figure.)
    oneshot_ms = []
    for _ in range(12):
        e0.record()
        _, n1 = regex.match_device_bits(dev, cap_lines=nlines + 64, out=out if out.numel() >= (nlines + 64 + 31) // 32 else None)
        e1.record()
        torch.cuda.synchronize()
        assert n1 == nlines
        oneshot_ms.append(e0.elapsed_time(e1))
    # the same with a bound a first-time caller can know: no string shorter than four bytes on average, i.e. cap_lines = n / 4
    # (the call clears cap_lines / 8 bytes of bitmap: 268 MB for 8 GiB) - the figure above sizes the bitmap from an index built earlier
    apriori_ms = []
    out4 = torch.empty((nbytes // 4 + 31) // 32, dtype=torch.int32, device=dev.device)
    for _ in range(5):
        e0.record()
        _, n1 = regex.match_device_bits(dev, cap_lines=nbytes // 4, out=out4)
        e1.record()
        torch.cuda.synchronize()
        assert n1 == nlines
        apriori_ms.append(e0.elapsed_time(e1))
    del out4
    apriori_ms = sorted(apriori_ms)[2]
    oneshot_all = [round(x, 4) for x in oneshot_ms]
    oneshot_worst = max(oneshot_ms)
    oneshot_ms = sorted(oneshot_ms)[len(oneshot_ms) // 2 - 1]          # lower median of twelve
    for _ in range(args.warmup):
        regex.match_corpus_bits(corpus, out=out)
    barrier()
    # HIP events on the stream the kernel is launched on (torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        regex.match_corpus_bits(corpus, out=out)
        b.record()
    barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    per_rank = [round(nbytes * args.steps / elapsed_local / 1e9, 2)]
    if dist is not None:
        on = "cpu" if rehearsal == "1" else "cuda"
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=on)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        per_rank = [round(nbytes * args.steps / float(x.item()) / 1e9, 2) for x in every]

    accepted = int(regex.match_corpus(corpus).sum(dtype=torch.int64).item())
    # the same launches on the stride-2 table AS NUMBERED (a second regex that is forbidden the order search), so that the record
    # carries what the profiled order is worth on this box: three alternating rounds of ten launches each, after the
    # timed region; rank 0 at N = 1 only
    order_ab = None
    if rank == 0 and world == 1 and regex.table_order is not None:
        plain = rr.RRegex(pattern, getattr(rr, ENGINES[args.engine]), device=local_rank)
        plain.set_background_order(False)
        for _ in range(10):                                # (the GPU idled while `plain` was compiled on the host: both tables warm before the rounds)
            plain.match_corpus_bits(corpus, out=out)
            regex.match_corpus_bits(corpus, out=out)
        torch.cuda.synchronize()
        sums = {"ordered": [], "numbered": []}
        rounds = []
        for _ in range(3):
            for name, r_ in (("ordered", regex), ("numbered", plain)):
                ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for a, b in ev2:
                    a.record()
                    r_.match_corpus_bits(corpus, out=out)
                    b.record()
                torch.cuda.synchronize()
                ms = [a.elapsed_time(b) for a, b in ev2]
                sums[name] += ms
                rounds.append((name, round(sum(ms) / len(ms), 4)))
        assert plain.table_order is None
        order_ab = {k: round(sum(v) / len(v), 4) for k, v in sums.items()}
        order_ab["rounds"] = rounds
        del plain
    if rank == 0:
        avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
        achieved = nbytes / avg_kernel_s / 1e9
        cold_GBs = nbytes / ((cold_index_ms + cold_match_ms) / 1e3) / 1e9
        res = {
            "metric": "GB/s input scanned (whole job), 256-state regex" if args.workload == "url" else "GB/s input scanned (whole job)",
            "value": round(world * nbytes * args.steps / elapsed / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "BASELINE %s (%d reference states, %s class) over %.2f GiB synthetic lines per GPU"
                                   % (config_name, regex.states_n, "BitSet<%d>" % regex.set_class if regex.set_class else "Roaring", nbytes / 2**30),
                       "pattern_states": regex.states_n, "useful_states": regex.useful_states, "engine": regex.engine_name,
                       "table_order_profiled_conflicts_before_after": regex.table_order, "table_order_wait_ms": round(order_wait_ms, 1),
                       "table_order_ab_kernel_ms": order_ab,
                       "sampled_table_states_open_transitions": regex.sampled_table,
                       "sampled_table_escaped_lines_last_launch": regex.sampled_escapes() if regex.sampled_table else None,
                       "bytes_per_gpu": nbytes, "stripe_bytes": corpus.stripe, "lines_per_gpu": nlines, "accepted_rank0": accepted, "sharding": sharding},
            "per_gpu_GBs": round(nbytes * args.steps / elapsed / 1e9, 2),
            "per_rank_GBs": per_rank,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "traffic_from_profile": traffic_from_profile(args.workload, nbytes, regex.engine_name),
                         "frac_of_read_stream_6.4TBs": round(achieved / HBM_ACHIEVABLE_GBS, 4),
                         "kernel_ms_avg": round(avg_kernel_s * 1e3, 4), "kernel_ms_median": round(sorted(kernel_ms)[len(kernel_ms) // 2], 4),
                         "kernel_ms_min": round(min(kernel_ms), 4), "kernel_ms_each_launch": [round(x, 4) for x in kernel_ms],
                         "launches_before_the_timed_region": 8 + 3 + args.warmup,
                         "algorithmic_bytes_per_launch": nbytes},
            # the same corpus met for the first time: newline index + one match (nothing reused); never the headline
            "cold": {"index_ms": round(cold_index_ms, 4), "match_ms": round(cold_match_ms, 4), "GBs": round(cold_GBs, 2),
                     "frac": round(cold_GBs / HBM_PEAK_GBS, 4),
                     "one_shot_ms": round(oneshot_ms, 4), "one_shot_GBs": round(nbytes / oneshot_ms / 1e6, 2),
                     "one_shot_frac": round(nbytes / oneshot_ms / 1e6 / HBM_PEAK_GBS, 4), "one_shot_statistic": "median of 12 calls",
                     "one_shot_worst_over_median": round(oneshot_worst / oneshot_ms, 3), "one_shot_ms_each_call": oneshot_all,
                     "one_shot_bitmap": "sized from the line count of an index built earlier (+64)",
                     "one_shot_cap_n_over_4_ms": round(apriori_ms, 4), "one_shot_cap_n_over_4_frac": round(nbytes / apriori_ms / 1e6 / HBM_PEAK_GBS, 4)},
            "setup": {"generate_s": round(gen_s, 2), "index_ms": round(index_ms, 3)},
        }
        if args.pcie:
            sample = host[:min(nbytes, 4 << 30)]
            regex.match_host(sample[:1 << 20])
            t0 = time.perf_counter()
            regex.match_host(sample)
            res["pcie_inclusive_GBs"] = round(len(sample) / (time.perf_counter() - t0) / 1e9, 3)
        if args.search:
            s, e = regex.search_corpus(corpus)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                s, e = regex.search_corpus(corpus)
            torch.cuda.synchronize()
            res["search"] = {"GBs": round(3 * nbytes / (time.perf_counter() - t0) / 1e9, 2), "lines_with_a_match": int((e >= 0).sum().item()),
                             "mean_match_len": round(float((e - s)[e >= 0].double().mean().item()), 2) if int((e >= 0).sum().item()) else 0.0}
            del s, e
            cnt, first, s, e = regex.search_all(corpus)          # count pass + prefix sum + fill pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cnt, first, s, e = regex.search_all(corpus)
            torch.cuda.synchronize()
            res["search"]["all_matches"] = {"GBs": round(nbytes / (time.perf_counter() - t0) / 1e9, 2), "matches": int(s.numel())}
            total = int(s.numel())
            del cnt, first, s, e
            # the same through the one-call entry (rrx_search_all: one launch, look-back over the chunks' match counts),
            # arrays sized for the result (a caller that guesses too low pays a second call)
            first, s, e = regex.search_all_fused(corpus, cap=total)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                first, s, e = regex.search_all_fused(corpus, cap=total)
            torch.cuda.synchronize()
            res["search"]["all_matches_one_call"] = {"GBs": round(3 * nbytes / (time.perf_counter() - t0) / 1e9, 2), "matches": int(s.numel())}
            del first, s, e
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(pattern, host)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
