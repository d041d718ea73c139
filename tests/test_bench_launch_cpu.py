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


def test_visible_devices_mask_narrows_the_gpu_count(tmp_path, monkeypatch):
    """the launcher's GPU count is the KFD topology INTERSECTED with the *_VISIBLE_DEVICES masks (pure env parsing)."""
    sys.path.insert(0, ROOT)
    import glob
    import bench
    files = []
    for i, simd in enumerate((0, 1024, 1024, 1024, 1024)):  # node 0 = the CPU
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {16 if simd == 0 else 0}\nsimd_count {simd}\n")
        files.append(str(d / "properties"))
    monkeypatch.setattr(glob, "glob", lambda pat: files if "kfd" in pat else [])
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.gpu_count_without_hip() == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.gpu_count_without_hip() == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.gpu_count_without_hip() == 1  # HIP's "0,2" now indexes a list of one: only entry 0 survives
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "")
    assert bench.gpu_count_without_hip() == 0


def test_ranks_pin_themselves_to_disjoint_cpu_sets():
    """every rank takes its own CPU slice before its first GPU call -- from LOCAL_RANK / LOCAL_WORLD_SIZE
    alone when somebody else's torch.distributed.run started it (the driver's launch form)."""
    code = ("import sys, os, json; sys.path.insert(0, %r); import bench; r = bench.pin_rank_to_cpus(); "
            "print(json.dumps({'pin': r, 'aff': sorted(os.sched_getaffinity(0))}))" % ROOT)
    n_cpu = len(os.sched_getaffinity(0))
    if n_cpu < 4:
        import pytest
        pytest.skip("needs >= 4 CPUs")
    seen = []
    for local in range(2):
        env = {k: v for k, v in os.environ.items() if k not in ("BRATS_RANK_CPUS", "BRATS_NO_PIN")}
        env.update({"LOCAL_RANK": str(local), "LOCAL_WORLD_SIZE": "2", "WORLD_SIZE": "2", "RANK": str(local)})
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-2000:]
        rec = json.loads(p.stdout.strip().splitlines()[-1])
        assert rec["pin"] is not None and rec["pin"]["n"] == len(rec["aff"]) == n_cpu // 2
        seen.append(set(rec["aff"]))
    assert not (seen[0] & seen[1])
    # world 1: nothing to pin
    env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_RANK", "LOCAL_WORLD_SIZE", "WORLD_SIZE", "RANK", "BRATS_RANK_CPUS")}
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert json.loads(p.stdout.strip().splitlines()[-1])["pin"] is None


def test_graph_leg_runs_in_fresh_ranks_and_reports_a_failure_as_its_exit_code():
    """the graph + RCCL leg is a child torch.distributed.run (fresh rank processes, rank environment of the
    caller scrubbed); success carries its numbers, a failed child its exit code -- never an exception, never a retry.  Exercised
    here through --dry-run (gloo, no GPU)."""
    sys.path.insert(0, ROOT)
    import bench
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_PORT="1", OMP_NUM_THREADS="1")  # a rank's own environment
    ok = bench.graph_ddp_leg(2, ["--gpus", "2", "--dry-run"], env)
    assert ok["rc"] == 0 and ok["n_gpus"] == 2 and "stdout_tail" not in ok
    bad = bench.graph_ddp_leg(2, ["--gpus", "3", "--dry-run"], env)  # (WORLD_SIZE 2 under a --gpus 3 label: the ranks refuse)
    assert bad["rc"] not in (0, None) and "ms_per_step" not in bad
    hung = bench.graph_ddp_leg(2, ["--gpus", "2", "--dry-run"], dict(env, BRATS_GRAPH_LEG_TIMEOUT="0.05"))
    assert hung["rc"] == 124


def test_ranks_take_the_cpus_next_to_their_gpu(tmp_path, monkeypatch):
    """pin_rank_to_cpus on a fake 4-GPU, 2-socket topology: KFD nodes -> PCI address -> local_cpulist; the two ranks of a socket split
    its CPUs in rank order; a KFD node this container may not read (another tenant's GPU) is skipped; a device mask falls back to
    contiguous slices."""
    sys.path.insert(0, ROOT)
    import bench
    kfd, pci = tmp_path / "kfd", tmp_path / "pci"
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 8:
        import pytest
        pytest.skip("needs >= 8 CPUs")
    half = len(cpus) // 2
    socket = [cpus[:half], cpus[half:]]
    (kfd / "0").mkdir(parents=True)
    (kfd / "0" / "properties").write_text("cpu_cores_count 16\nsimd_count 0\n")
    for g in range(4):
        d = kfd / str(g + 1)
        d.mkdir()
        loc = ((0x10 + g) << 8)  # bus 0x10 + g, device 0, function 0
        (d / "properties").write_text(f"simd_count 1024\nlocation_id {loc}\ndomain 0\n")
        dev = pci / f"0000:{0x10 + g:02x}:00.0"
        dev.mkdir(parents=True)
        s = socket[g // 2]
        (dev / "local_cpulist").write_text(f"{s[0]}-{s[-1]}\n" if s == list(range(s[0], s[-1] + 1)) else ",".join(map(str, s)) + "\n")
    monkeypatch.setattr(bench, "KFD_NODES", str(kfd))
    monkeypatch.setattr(bench, "PCI_DEVICES", str(pci))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "BRATS_RANK_CPUS", "BRATS_NO_PIN"):
        monkeypatch.delenv(var, raising=False)
    assert bench.gpu_count_without_hip() == 4
    near = bench.gpu_local_cpus()
    assert near == [socket[0], socket[0], socket[1], socket[1]]
    import torch
    before, threads_before = os.sched_getaffinity(0), torch.get_num_threads()  # (the pin also sets torch's thread count: put it back --
    try:                                                                       #  the oracle's golden tests compare CPU sums to 2e-5)
        got = []
        for local in range(4):
            os.sched_setaffinity(0, before)
            monkeypatch.setenv("LOCAL_RANK", str(local))
            monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
            r = bench.pin_rank_to_cpus()
            if half // 2 >= 4:
                assert r is not None and "NUMA" in r["source"], r
                got.append(sorted(os.sched_getaffinity(0)))
            else:
                assert r is None  # fewer than 4 CPUs per rank: nothing is pinned
        if got:
            assert got[0] + got[1] == socket[0] and got[2] + got[3] == socket[1]
        # a device mask reorders / hides devices: contiguous slices of the affinity set instead
        os.sched_setaffinity(0, before)
        monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
        monkeypatch.setenv("LOCAL_RANK", "1")
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
        r = bench.pin_rank_to_cpus()
        assert r is not None and r["source"].startswith("contiguous") and sorted(os.sched_getaffinity(0)) == cpus[half:]
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(threads_before)
