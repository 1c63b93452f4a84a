"""bench.py's launcher contract, checked without a GPU: a request for N > 1 GPUs must never end in an N = 1 line."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


def test_gpus_2_without_two_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        return                      # a real multi-GPU node: the spawn path itself is exercised by the driver's scaling run
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "needs 2 visible GPUs" in r.stderr
    assert '"metric"' not in r.stdout          # no JSON line at all


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in r.stderr and '"metric"' not in r.stdout
    r = _run(["--gpus", "0"])
    assert r.returncode != 0
