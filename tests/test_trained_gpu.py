"""-m gpu: the Dice / logit bars of north_star on TRAINED weights ("next round" item 1).

Every other parity test runs the networks at their initialisation, where the thresholded prediction has little to do with
the target (hard Dice 0.0 ... 0.19 on two of three channels): "Dice within 1e-3" says nothing there.  Here both width-48
networks are first TRAINED on the GPU -- split-precision mode (model.precision = "x3"), GraphedTrainStep, fused Dice, Ranger2020
-- on oracle.synth.tumour_phantom volumes, whose label is a function of the image (intensity offsets of nested ellipsoids
under unit noise), until the logits sit around the decision surface because the network learned the target.  Then, with
those weights and on volumes the training never saw:

  * the CPU oracle (oracle/unet.py, the reference's CPU arithmetic) must itself reach hard Dice >= 0.5 on all three channels
    -- otherwise the test FAILS rather than compare degenerate numbers;
  * f32 and x3 logits within 1e-3 abs of the oracle (one 128^3 patch; the stitched logits of a configs[3] volume,
    4x240x240x155 padded to 160, 18 windows, for EquiUnet);
  * bf16 and fp16 (the reference's own autocast dtype, learning/engine.py:304): hard Dice against the target within 1e-3 of
    the oracle's -- the bar itself, for every precision on every volume of this test, the half-contrast stress patch included
    (round 6: the doubled bar bf16 had there is gone) -- with the margin and the fraction of thresholded voxels that flipped printed;
  * a SWEEP over five more trained weight sets per network (other initialisations, other training volumes) and three fresh
    volumes each -- training-like, half contrast, a third of the contrast -- so that bf16's margin is a measured distribution and
    not one sample: the stated bar (1e-3 on volumes like the ones the network was trained and is benchmarked on) is asserted for
    every set and both 16-bit types, fp16 is asserted at 1e-3 on the stress volumes too, and bf16's stress-volume tail is
    reported (profiles/r06_trained_weights_parity.txt) without a bar of its own;
  * the benchmarked chain (sliding window + 8-flip TTA, Evaluator: learning/engine.py:236-259, src/definer.py:696-697) in
    bf16 / fp16 against the same chain in f32, whose network arithmetic the stitched-logit check has just pinned.
The 320-step training runs reproduce BIT FOR BIT across runs and boxes (every reduction of the step is ordered; checked: two
runs on one box and a third on another print identical loss curves and Dice values), so the margins below are properties of the tree,
not of the lease.  The numbers land in profiles/ through scripts/r5_final.sh (pytest -s)."""
import argparse
import contextlib
import io
import os
import warnings

import pytest
import torch

from oracle import inference as oinf
from oracle import synth, unet

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
LOGIT_ATOL = 1e-3
DICE_ATOL = 1e-3
F8_DICE_ATOL = 5e-2  # the opt-in e4m3 convolution path (3 significand bits) does NOT hold the 1e-3 bar: 6e-4 .. 1.05e-2 seen over the
                     # trees of round 5 (it moves with the trained weights); the number is printed and recorded, the assert is a sanity bound
PATCH = (128, 128, 128)
VOL = (240, 240, 155)
STEPS = {"equiunet": 320, "equiunet_assp_evo": 320}
POOL = 12  # training batches of 2 patches, cycled


def _get(model, seed=0):
    from brats21_amd import get_model
    torch.manual_seed(seed)
    ns = argparse.Namespace(model=model, width=48, norm="group", act="relu", num_classes=3, dropout=0)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return get_model(ns)


_trained = {}


def _train(model, seed=0, steps=None):
    """x3-mode training run on the phantom; returns (module on the GPU in eval mode, CPU state dict, loss curve).  seed = 0 is the
    weight set of rounds 4 / 5; other seeds change the initialisation and the training volumes (the sweep)."""
    key = (model, seed)
    if key in _trained:
        return _trained[key]
    from brats21_amd.engine import GraphedTrainStep, TrainStep
    from brats21_amd.optim import Ranger2020
    m = _get(model, seed).to(DEV).train()
    m.precision = "x3"
    with contextlib.redirect_stdout(io.StringIO()):
        opt = Ranger2020(m.parameters(), lr=2e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-5, weight_decay=1e-5,
                         capturable=True)
    step = GraphedTrainStep(TrainStep(m, opt, criterion=None, amp=False))
    pool = [tuple(a.to(DEV) for a in synth.tumour_phantom(2, PATCH, 5000 + 100 * seed + i)) for i in range(POOL)]
    curve = []
    steps = steps or STEPS[model]
    for it in range(steps):
        x, t = pool[it % POOL]
        loss = step(x, t)
        if it % 20 == 0 or it == steps - 1:
            curve.append((it, float(loss.item())))
    del step, pool
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    m.eval()
    print(f"\n{model}-48 (weight set {seed}) trained {steps} x3 steps on 2x4x128^3 phantoms: loss " + " ".join(f"{i}:{l:.3f}" for i, l in curve))
    if seed != 0:  # (the sweep keeps the CPU weights only)
        m = None
    _trained[key] = (m, sd, curve)
    return _trained[key]


def _oracle(model, sd, x):
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    fwd = unet.equiunet_forward if model == "equiunet" else unet.assp_evo_forward
    with torch.no_grad():
        return fwd(sd, x)[0]


def _report(tag, out, ref, t, d_ref):
    dev = (out - ref).abs()
    d = unet.hard_dice(out, t)
    flips = float(((out > 0) != (ref > 0)).float().mean())
    margin = float((d - d_ref).abs().max())
    print(f"  {tag}: logits max {float(dev.max()):.3e} mean {float(dev.mean()):.3e}; hard Dice {[round(v, 5) for v in d.flatten().tolist()]} "
          f"|dDice| max {margin:.2e} (bar {DICE_ATOL:.0e}); thresholded voxels flipped {flips:.3e}")
    return margin, flips


@pytest.mark.parametrize("model", ["equiunet", "equiunet_assp_evo"])
def test_trained_patch_logits_and_dice_vs_oracle(model):
    """Two volumes the training never saw: one like the training set (contrast 1.0: Dice ~0.98, few voxels near the decision
    surface) and one at HALF the contrast, where the trained network is unsure over a large share of the lesion."""
    m, sd, curve = _train(model)
    assert curve[-1][1] < 0.6 * curve[0][1], curve  # it did learn
    for contrast, seed in ((1.0, 777), (0.5, 778)):
        x, t = synth.tumour_phantom(1, PATCH, seed, contrast=contrast)
        ref = _oracle(model, sd, x)
        d_ref = unet.hard_dice(ref, t)
        near = float((ref.abs() < 0.5).float().mean())
        print(f"\n{model}-48 TRAINED, fresh 4x128^3 phantom at contrast {contrast}: oracle hard Dice (TC, WT, ET) "
              f"{[round(v, 4) for v in d_ref.flatten().tolist()]}, |logits| max {float(ref.abs().max()):.2f}, voxels with |logit| < 0.5: {near:.3e}")
        assert float(d_ref.min()) >= 0.5, f"the oracle's own Dice with the trained weights is degenerate: {d_ref}"
        xd = x.to(DEV)
        res = {}
        with torch.no_grad():
            for prec in ("fp32", "x3", "bf16", "fp16"):
                m.precision = prec
                out = m(xd)
                out = (out[0] if isinstance(out, (tuple, list)) else out).float().cpu()
                res[prec] = _report(prec, out, ref, t, d_ref) + (float((out - ref).abs().max()),)
            # BASELINE.json configs[4]'s arithmetic: fp16 storage, every 3x3x3 convolution on the e4m3 MFMA kernels (scales
            # from the recorded |max| of each tensor)
            m.precision, m.conv_fp8 = "fp16", "fwd"
            out = m(xd)
            out = (out[0] if isinstance(out, (tuple, list)) else out).float().cpu()
            res["fp16+e4m3"] = _report("fp16 + e4m3 convolutions", out, ref, t, d_ref) + (float((out - ref).abs().max()),)
            # e4m3 only at the large levels (128^3 + 64^3; 128^3 alone) -- does a restricted form hold the bar?
            from brats21_amd import ops as _ops
            for min_size in (64, 128):
                old_ms = _ops.set_f8_min_size(min_size)
                try:
                    out = m(xd)
                finally:
                    _ops.set_f8_min_size(old_ms)
                out = (out[0] if isinstance(out, (tuple, list)) else out).float().cpu()
                res[f"fp16+e4m3>={min_size}"] = _report(f"fp16 + e4m3 convolutions at the levels >= {min_size}^3 only", out, ref, t, d_ref) + (float((out - ref).abs().max()),)
            m.conv_fp8 = None
        m.precision = "x3"
        assert res["fp32"][2] < LOGIT_ATOL and res["x3"][2] < LOGIT_ATOL, res
        assert res["fp16+e4m3"][0] <= F8_DICE_ATOL, (contrast, res["fp16+e4m3"])
        for prec in ("fp32", "x3", "bf16", "fp16"):  # the bar itself, everywhere (the half-contrast stress patch included)
            assert res[prec][0] <= DICE_ATOL, (contrast, prec, res[prec])


# the whole sweep (five weight sets per network, ~7 minutes) runs with BRATS_SWEEP_FULL=1 -- scripts/r6_final.sh, whose output is
# profiles/r06_final_trained_weights_parity.txt; the default GPU suite takes the first two weight sets (the same asserts)
SWEEP_SEEDS = (1, 2, 3, 4, 5) if os.environ.get("BRATS_SWEEP_FULL", "0") == "1" else (1, 2)
SWEEP_STEPS = {"equiunet": 560, "equiunet_assp_evo": 320}
SWEEP_VOLUMES = ((1.0, "training-like"), (0.5, "half contrast (stress)"), (0.35, "a third of the contrast (stress)"))


@pytest.mark.parametrize("model", ["equiunet", "equiunet_assp_evo"])
def test_trained_seed_sweep_dice_margin_distribution(model):
    """five more trained weight sets (other initialisations, other training volumes), three fresh volumes each.
    Asserted at the bar (1e-3, no other number in this file): bf16 and fp16 on the training-like volume of every weight set -- the
    volumes north_star's sentence is about -- and fp16 on every stress volume.  bf16 on the stress volumes, where the trained
    network itself is unsure (oracle Dice 0.5 .. 0.9), is REPORTED as a distribution: its tail is the number a user of the bf16
    configuration should know, and fp16 (15.1 vs 14.5 ms / step, bench.py `fp16_mode`) is the configuration that holds the bar
    there.  The table goes to gpurun_out/r06_trained_sweep_<model>.txt for profiles/."""
    from brats21_amd import get_model  # noqa: F401
    rows, lines = [], []
    for seed in SWEEP_SEEDS:
        # (EquiUnet's other initialisations need more than weight set 0's 320 steps before all three classes are learnt)
        _, sd, curve = _train(model, seed, steps=SWEEP_STEPS[model])
        assert curve[-1][1] < 0.6 * curve[0][1], (seed, curve)
        m = _get(model, seed)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        for vi, (contrast, what) in enumerate(SWEEP_VOLUMES):
            x, t = synth.tumour_phantom(1, PATCH, 9000 + 10 * seed + vi, contrast=contrast)
            ref = _oracle(model, sd, x)
            d_ref = unet.hard_dice(ref, t)
            xd = x.to(DEV)
            row = {"seed": seed, "contrast": contrast, "oracle_dice_min": float(d_ref.min())}
            with torch.no_grad():
                for prec in ("bf16", "fp16"):
                    m.precision = prec
                    out = m(xd)
                    out = (out[0] if isinstance(out, (tuple, list)) else out).float().cpu()
                    row[prec] = float((unet.hard_dice(out, t) - d_ref).abs().max())
                    row[prec + "_flips"] = float(((out > 0) != (ref > 0)).float().mean())
            rows.append(row)
            lines.append(f"{model}-48 weight set {seed}, {what:34s}: oracle Dice min {row['oracle_dice_min']:.4f}; |dDice| bf16 {row['bf16']:.2e} "
                         f"(flipped {row['bf16_flips']:.1e}), fp16 {row['fp16']:.2e} (flipped {row['fp16_flips']:.1e})")
            print("  " + lines[-1])
        del m
        torch.cuda.empty_cache()
    def stat(vals):
        v = sorted(vals)
        return f"n {len(v)}, median {v[len(v) // 2]:.2e}, max {v[-1]:.2e}, above the 1e-3 bar: {sum(1 for a in v if a > DICE_ATOL)}"
    # a weight set whose ORACLE segmentation of the training-like volume is degenerate (a class never learnt) says nothing about
    # parity: it is listed, not counted -- and at least four of the five must count
    bad = {r["seed"] for r in rows if r["contrast"] == 1.0 and r["oracle_dice_min"] < 0.5}
    lines.append(f"weight sets not counted (oracle Dice < 0.5 on their training-like volume): {sorted(bad) if bad else 'none'}")
    assert len(bad) <= (1 if len(SWEEP_SEEDS) >= 4 else 0), bad
    rows = [r for r in rows if r["seed"] not in bad]
    like = [r for r in rows if r["contrast"] == 1.0]
    stress = [r for r in rows if r["contrast"] < 1.0]
    summary = [f"{model}-48, {len(SWEEP_SEEDS) - len(bad)} trained weight sets x {len(SWEEP_VOLUMES)} fresh volumes, hard-Dice margin against the CPU oracle:",
               f"  training-like volumes  bf16: {stat([r['bf16'] for r in like])}",
               f"  training-like volumes  fp16: {stat([r['fp16'] for r in like])}",
               f"  stress volumes         bf16: {stat([r['bf16'] for r in stress])}",
               f"  stress volumes         fp16: {stat([r['fp16'] for r in stress])}"]
    print("\n" + "\n".join(summary))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"r06_trained_sweep_{model}.txt"), "w") as f:
        f.write("\n".join(summary + [""] + lines) + "\n")
    for r in like:
        assert r["bf16"] <= DICE_ATOL and r["fp16"] <= DICE_ATOL, r
    for r in stress:
        assert r["fp16"] <= DICE_ATOL, r


def test_trained_config3_sliding_window_flip8_vs_oracle():
    """configs[3] with the trained EquiUnet-48: one phantom volume 4x240x240x155, padded to 160, 128^3 windows, overlap 0.5."""
    from brats21_amd import tta
    from brats21_amd.evaluate import Evaluator, hard_dice_metric, shape_to_divisible
    from brats21_amd.inferers import sliding_window_inference
    m, sd, _ = _train("equiunet")
    m.skip_deep_heads_in_eval = True
    x, t = synth.tumour_phantom(1, VOL, 888)
    xd, td = x.to(DEV), t.to(DEV)
    padded, p_b, p_a = shape_to_divisible(xd, k=8)
    assert tuple(padded.shape[2:]) == (240, 240, 160)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref = oinf.sliding_window_inference(padded.cpu(), PATCH, 1, lambda w: unet.equiunet_forward(sd, w)[0], overlap=0.5)
    crop = (slice(None), slice(None), slice(None), slice(None), slice(p_b[2], 160 - p_a[2]))
    d_ref = unet.hard_dice(ref[crop], t)
    print(f"\nEquiUnet-48 TRAINED, configs[3] phantom volume: oracle stitched hard Dice (TC, WT, ET) {[round(v, 4) for v in d_ref.flatten().tolist()]}")
    assert float(d_ref.min()) >= 0.5, d_ref
    errs = {}
    with torch.no_grad():
        for prec in ("fp32", "x3", "bf16", "fp16"):
            m.precision = prec
            out = sliding_window_inference(padded, PATCH, 3, lambda w: m(w), overlap=0.5).float().cpu()
            errs[prec] = _report(f"stitched, identity TTA, {prec}", out[crop], ref[crop], t, d_ref) + (float((out - ref).abs().max()),)
    assert errs["fp32"][2] < LOGIT_ATOL and errs["x3"][2] < LOGIT_ATOL, errs
    for prec in errs:
        assert errs[prec][0] <= DICE_ATOL, (prec, errs[prec])
    # the benchmarked chain: 8-flip TTA + threshold + background removal, 16-bit against f32
    dice, seg = {}, {}
    for prec, amp, dt in (("fp32", False, None), ("bf16", True, torch.bfloat16), ("fp16", True, torch.float16)):
        m.precision = prec
        ev = Evaluator(m, tta_transforms=tta.flip8(), sliding_window_size=PATCH, sw_batch_size=3, overlap=0.5, k_divisible=8, amp=amp)
        out = ev(xd, target=td)
        seg[prec], dice[prec] = out["seg"].cpu(), hard_dice_metric(out["seg"], td).cpu()
        del ev
    for prec in ("bf16", "fp16"):
        margin = float((dice[prec] - dice["fp32"]).abs().max())
        flips = float((seg[prec] != seg["fp32"]).float().mean())
        print(f"  8-flip chain {prec}: hard Dice {[round(v, 5) for v in dice[prec].flatten().tolist()]} vs f32 "
              f"{[round(v, 5) for v in dice['fp32'].flatten().tolist()]}: |dDice| max {margin:.2e}, segmentation voxels flipped {flips:.3e}")
        assert margin <= DICE_ATOL, (prec, dice)
    m.precision = "x3"
