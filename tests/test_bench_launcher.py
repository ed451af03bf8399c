"""bench.py --gpus N must start N ranks by itself and prove it.  RRX_BENCH_REHEARSAL=launcher
runs only the launcher, the rendezvous and the rank proof (gloo, no device work), so this runs on a CPU-only box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_starts_two_ranks_and_reports_them():
    p = run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"RRX_BENCH_REHEARSAL": "launcher"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout            # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2


def test_flag_and_launch_must_agree():
    p = run(["--gpus", "2"], {"RRX_BENCH_REHEARSAL": "launcher", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert "disagree" in (p.stderr + p.stdout)
