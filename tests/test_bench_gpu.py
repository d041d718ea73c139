"""-m gpu: bench.py's output contract on a small configuration (a child process: the JSON line is what the driver parses)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout  # ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_line_contract_small_configuration():
    r = _bench("--width", "8", "--patch", "32", "--steps", "4", "--warmup", "2", "--no-infer", "--no-cpu-baseline", "--no-parity-leg")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "box"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 4 and r["warmup"] == 2 and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["vs_baseline"] is None and r["data"] == "synthetic" and r["dtype"] == "bf16" and "workload" in r["config"]
    assert abs(r["value"] - 2 / (r["ms_per_step"] * 1e-3)) < 0.02 * r["value"]  # patches/s = batch / step time
    rf, box = r["roofline"], r["box"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "frac_of_box", "traffic", "kernel", "avg_ms"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["frac_of_box"] - rf["achieved"] / box["mfma_TFLOPs"]) < 1e-3
    # the box probe: a sane MI355X (sustained 16-bit matrix rate between 1 and 2.6 PFLOP/s, stream between 3 and 8 TB/s)
    assert 1000.0 < box["mfma_TFLOPs"] < 2600.0 and 3.0 < box["stream_TBps"] < 8.0 and 1000 < box["sclk_MHz"] < 2500


def test_bench_x3_precision_runs_the_split_kernels():
    r = _bench("--width", "8", "--patch", "32", "--steps", "3", "--warmup", "1", "--precision", "x3", "--no-infer", "--no-cpu-baseline",
               "--no-parity-leg")
    assert r["dtype"] == "x3" and r["roofline"]["peak"] < 1000.0  # a third of the 16-bit peak: three MFMAs per product


def test_bench_roofline_stays_on_the_plain_kernel_beside_the_backward_statistics_form():
    """Width 48: the blocks' second input-gradient launches carry a GroupNorm-backward pass (family conv_igemm_bst).  The
    roofline line must name a plain implicit-GEMM launch (whole work = the FLOPs in its numerator) and list the fused form
    beside it; with BRATS_FOLD_BWD_STATS=0 the family is absent."""
    args = ("--width", "48", "--patch", "32", "--steps", "3", "--warmup", "1", "--no-infer", "--no-cpu-baseline", "--no-parity-leg")
    r = _bench(*args)
    rf = r["roofline"]
    assert rf["kernel"].startswith("conv_igemm ") and "conv_igemm_bst" in rf["families"]
    ff = rf["fused_epilogue_form"]
    assert ff["kernel"].startswith("conv_igemm_bst ") and ff["launches"] > 0 and ff["avg_ms"] > 0
    old = os.environ.get("BRATS_FOLD_BWD_STATS")
    os.environ["BRATS_FOLD_BWD_STATS"] = "0"
    try:
        r0 = _bench(*args)
    finally:
        if old is None:
            os.environ.pop("BRATS_FOLD_BWD_STATS")
        else:
            os.environ["BRATS_FOLD_BWD_STATS"] = old
    assert "conv_igemm_bst" not in r0["roofline"]["families"] and "fused_epilogue_form" not in r0["roofline"]


def test_bench_other_stated_configurations_are_on_the_line():
    """configs[2] (ASSP-48 bf16, eager + graph), the reference's fp16 + GradScaler loop and configs[4] (ASSP-64 fp16 + e4m3) ride
    the default line as short legs; here at a 32^3 patch."""
    r = _bench("--width", "48", "--patch", "32", "--steps", "3", "--warmup", "1", "--no-infer", "--no-cpu-baseline", "--no-parity-leg",
               "--other-configs-patch", "32")
    for leg, dtype in (("configs2_per_gpu", "bf16"), ("fp16_mode", "fp16"), ("configs4_per_gpu", "fp16+e4m3 conv (all)")):
        assert leg in r, leg
        assert r[leg]["dtype"] == dtype and r[leg]["ms_per_step"] > 0 and r[leg]["loss"] == r[leg]["loss"]
        assert abs(r[leg]["patches_per_s"] - r[leg]["patches_per_gpu"] / (r[leg]["ms_per_step"] * 1e-3)) < 0.02 * r[leg]["patches_per_s"]
    assert r["configs2_per_gpu"]["as_one_hipgraph"]["ms_per_step"] > 0 and "grad_scale" in r["fp16_mode"]
    # configs[4] is a throughput figure, not a parity configuration, and its label leads with that
    assert r["configs4_per_gpu"]["config"].startswith("does NOT hold parity") and r["configs4_per_gpu"]["holds_parity"] is False
    assert r["metric"].startswith("train patches/sec") and r["dtype"] == "bf16"  # headline fields unchanged


def test_bench_two_ranks_on_one_gpu_exercise_the_ddp_block():
    """`bench.py --gpus 2` with both ranks on this box's one GPU (gloo: RCCL refuses two ranks per device): the launcher, the
    patient sharding, the bucketed gradient all-reduce overlapped with the backward program and the `ddp` block of the JSON line
    (allreduce_ms, exposed_ms, overlap_frac, per-rank step times) run every round, not only on the day an 8-GPU node appears
    (VERDICT r4 item 8).  The numbers mean nothing as a scaling result: two ranks share one GPU and gloo reduces on the host."""
    r = _bench("--gpus", "2", "--width", "8", "--patch", "32", "--steps", "6", "--warmup", "2", "--no-infer", "--no-cpu-baseline",
               "--no-parity-leg", "--no-other-configs", env={"BRATS_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "4"})
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2" and r["scaling"] == "weak"
    assert abs(r["value"] - 4 / (r["ms_per_step"] * 1e-3)) < 0.02 * r["value"]  # whole-job patches/s = 2 ranks x 2 patches / step time
    d = r["ddp"]
    assert d["world_size"] == 2 and d["backend"] == "gloo" and len(d["ms_per_step_by_rank"]) == 2 and d["buckets"] >= 1
    assert d["payload_MB"] > 0 and d["allreduce_ms"] > 0 and d["comm_dtype"] == "float32" and d["graph_captured_collectives"] is False
    assert d["exposed_ms"] is not None and d["exposed_ms"] >= 0 and 0.0 <= d["overlap_frac"] <= 1.0
    assert max(d["ms_per_step_by_rank"]) <= r["ms_per_step"] * 1.05  # the line reports the MAX over ranks


def test_bench_rccl_rehearsal_eager_headline_then_graph_leg_in_fresh_ranks():
    """The shape of the driver's N > 1 line, rehearsed on this box's one GPU over the REAL RCCL backend (BRATS_FORCE_DDP=rccl: a
    world-1 communicator, every bucket's all-reduce a real asynchronous RCCL launch): the eager step is the headline and carries
    the host-enqueue budget and ranks_seen in its `ddp` block; the whole step incl. the collectives as one hipGraph runs afterwards
    in a FRESH rank process started by rank 0 and lands as `graph_ddp` with its exit code (VERDICT r5 item 1 a, b, e)."""
    r = _bench("--width", "8", "--patch", "32", "--steps", "6", "--warmup", "3", "--no-infer", "--no-cpu-baseline", "--no-parity-leg",
               "--no-other-configs", env={"BRATS_FORCE_DDP": "rccl"})
    d = r["ddp"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["ranks_seen"] == 1 and d["graph_captured_collectives"] is False
    assert d["buckets"] >= 1 and d["allreduce_ms"] > 0 and d["exposed_ms"] is not None
    assert d["host_enqueue_ms"] > 0 and r["host_enqueue_ms"] > 0  # (no upper bound: this tiny configuration is host-bound, the burst after the timed region may read above the in-loop average)
    g = r["graph_ddp"]
    assert g["rc"] == 0, g
    assert g["ms_per_step"] > 0 and g["n_gpus"] == 1 and g["ddp"]["graph_captured_collectives"] is True and g["ddp"]["backend"] == "nccl"
    assert g["host_enqueue_ms"] < r["host_enqueue_ms"]  # one graph launch per step instead of a few hundred enqueues
    assert abs(g["loss"] - r["config"]["loss"]) < 2e-2  # same workload and seed (the capture adds two eager warm-up steps)


def test_bench_measures_the_dominant_kernels_hbm_traffic_in_its_own_run():
    """roofline.traffic is measured by the run that prints the line -- two rocprofv3 --pmc child processes
    (FETCH_SIZE, WRITE_SIZE: separate passes) over `bench.py --pmc-leg`, the dominant kernel's launch -- and says so; the committed
    record is only the labelled fallback.  The headline's own shape (width 48, 2 x 128^3), a short run."""
    r = _bench("--steps", "4", "--warmup", "2", "--no-infer", "--no-cpu-baseline", "--no-parity-leg", "--no-other-configs")
    rf = r["roofline"]
    assert rf["kernel"] == "conv_igemm cin=48 cout=48 k=3 dil=1 @2x128x128x128"
    tp = rf["traffic_profiled"]
    assert tp["source"].startswith("live"), tp
    algo = tp["algorithmic_MB"] * 1e6
    assert rf["traffic"] == int((tp["fetch_MB"] + tp["write_MB"]) * 1024 * 1024)
    assert 0.9 * algo < rf["traffic"] < 2.5 * algo, (rf["traffic"], algo)  # every byte at least once, no runaway re-reads
    assert 0.9 * algo / 2 < tp["write_MB"] * 1024 * 1024 < 1.2 * algo / 2   # the output is written once
