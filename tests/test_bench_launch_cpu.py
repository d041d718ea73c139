"""CPU: bench.py starts its own ranks when it is run bare with --gpus N (`python bench.py --gpus 8` used to run
world 1 silently and print n_gpus 1).  --dry-run exercises the launcher, the rendezvous on 127.0.0.1 and the relay of rank
0's JSON line with two gloo ranks and no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_bare_gpus_2_starts_two_ranks_and_relays_one_json_line():
    p = _run(["--gpus", "2", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["dry_run"] is True


def test_bare_gpus_8_starts_eight_ranks():
    """The driver's full-node launch shape (--gpus 8): launcher, rendezvous on 127.0.0.1, one all-reduce over eight gloo ranks,
    ONE JSON line relayed."""
    p = _run(["--gpus", "8", "--dry-run"], {"OMP_NUM_THREADS": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["ranks_seen"] == 8 and rec["dry_run"] is True


def test_gpu_count_comes_from_sysfs_not_from_hip():
    """the launcher counts GPUs from the KFD topology (no HIP call in the parent); unknown topology = None."""
    sys.path.insert(0, ROOT)
    import bench
    n = bench.gpu_count_without_hip()
    assert n is None or (isinstance(n, int) and n >= 0)
    src = open(os.path.join(ROOT, "bench.py")).read()
    launch = src[src.index("def launch_ranks"):src.index("def dry_run")]
    assert "device_count" not in launch


def test_world_size_mismatch_fails_loudly():
    p = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0
    assert "WORLD_SIZE=1" in (p.stderr + p.stdout)
    assert not any(l.lstrip().startswith("{") for l in p.stdout.splitlines())  # no n_gpus line for a run that did not happen


def test_bare_gpus_1_runs_in_process():
    p = _run(["--gpus", "1", "--dry-run"])
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1
