"""`bench.py --gpus N` typed on its own must start N ranks itself (missing #3): the parent launches them as child
processes before any GPU call and relays rank 0's JSON line.  Run here with --dry-run (gloo, no device work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"] + extra, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # ONE JSON line on stdout, nothing else
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks():
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["gpus_requested"] == 2
    assert out["passes_all_ranks"] == 2.0 and out["max_elapsed"] == 2.0      # SUM and MAX over the ranks in one all-reduce


def test_gpus_1_is_one_process():
    out = _run(["--gpus", "1"])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1


def test_under_a_launcher_no_second_launch():
    """With WORLD_SIZE set (the driver's torch.distributed.run) bench.py is a rank, not a launcher."""
    out = _run(["--gpus", "2"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1
