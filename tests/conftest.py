import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


DDP_GPU_RESULT = os.path.join(ROOT, "gpurun_out", "ddp2_one_gpu.json")
_ddp_proc = None


def deselect_match(item, config):
    """-k is applied by pytest AFTER this hook (it is a modifyitems hook of its own, ordering not guaranteed): evaluate it
    here so that `-k something_else` does not start the two-rank worker."""
    try:
        from _pytest.mark import KeywordMatcher
        from _pytest.mark.expression import Expression
        return Expression.compile(config.getoption("-k")).evaluate(KeywordMatcher.from_item(item))
    except Exception:
        return True


def _gpu_run_selected(config):
    expr = config.getoption("-m") or ""
    return "gpu" in expr and "not gpu" not in expr


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # tests/test_ddp_gpu.py::test_two_ranks_on_one_gpu: the two ranks must be FRESH processes started before this process
    # touches the GPU (a process that has initialised HIP must not fork + exec on the GPU boxes), so they are launched
    # here -- after collection (importing the test modules does not initialise HIP), before the first test runs, and only
    # when that test is actually selected -- and run beside the other tests; the test only waits for their verdict file.
    global _ddp_proc
    if not _gpu_run_selected(config) or _ddp_proc is not None:
        return
    deselect = config.getoption("-k") or ""
    wanted = [it for it in items if "test_two_ranks_on_one_gpu" in it.nodeid and it.get_closest_marker("gpu")]
    if not wanted or (deselect and not any(deselect_match(it, config) for it in wanted)):
        return
    import subprocess
    from bench import gpu_count_without_hip  # sysfs + *_VISIBLE_DEVICES only: this process must not have touched HIP yet
    if not gpu_count_without_hip():
        return
    os.makedirs(os.path.dirname(DDP_GPU_RESULT), exist_ok=True)
    for f in (DDP_GPU_RESULT, DDP_GPU_RESULT + ".err0", DDP_GPU_RESULT + ".err1"):
        if os.path.exists(f):
            os.remove(f)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    log = open(DDP_GPU_RESULT + ".log", "w")
    _ddp_proc = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port),
                                  os.path.join(ROOT, "tests", "_ddp_gpu_worker.py"), DDP_GPU_RESULT],
                                 stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT)


def pytest_unconfigure(config):
    if _ddp_proc is not None and _ddp_proc.poll() is None:
        _ddp_proc.kill()  # (the exact child started above)


@pytest.fixture(scope="session")
def ddp_two_rank_result():
    """(process, verdict path) of the 2-rank run started at configure time, or None when it was not started."""
    return (_ddp_proc, DDP_GPU_RESULT) if _ddp_proc is not None else None


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
