"""-m gpu: the multi-tensor HIP Ranger2020 (brats21_amd/optim.py, csrc/ranger.hip) against the golden vectors of
the reference's learning/optimizer.py and against the CPU oracle on a real model's parameter set."""
import argparse
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import ranger as orang

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_ranger_matches_reference_golden(golden_dir):
    from _replay import ranger_replay
    from brats21_amd.optim import Ranger2020

    def step_fn(params, lr, kw):
        plist = list(params.values())
        opt = Ranger2020(plist, lr=lr, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-5, **kw)

        def step(grads):
            for n, p in params.items():
                p.grad = None if grads[n] is None else grads[n].to(DEV)
            opt.step()
        return {"step": step, "state": lambda n: opt.state[params[n]]}

    ranger_replay(golden_dir, lambda t: torch.nn.Parameter(t.to(DEV)), step_fn, lambda t: t.detach().cpu().numpy(),
                  rtol=2e-5)  # f32 tolerance: the device's sqrt / divide round differently from the CPU's by <= 1-2 ulp per step


def test_ranger_full_model_vs_oracle_and_state_dict_roundtrip():
    """All EquiUnet-16 parameters, 7 steps (both RAdam branches + one lookahead), vs the oracle tensor by tensor;
    then a state_dict round trip through the CPU (Engine.resume path) must continue identically."""
    from brats21_amd import get_model
    from brats21_amd.optim import Ranger2020
    with contextlib.redirect_stdout(io.StringIO()):
        m = get_model(argparse.Namespace(model="equiunet", width=16, norm="group", act="relu", num_classes=3, dropout=0)).to(DEV)
    params = [p for p in m.parameters()]
    cpu = [p.detach().cpu().clone() for p in params]
    states = [orang.new_state(c) for c in cpu]
    kw = dict(lr=3e-3, alpha=0.5, k=6, betas=(.95, 0.999), eps=1e-5, weight_decay=1e-5, use_gc=True, gc_conv_only=False)
    opt = Ranger2020(params, N_sma_threshhold=5, **kw)
    gen = torch.Generator().manual_seed(3)

    def one_step(o):
        for p, c, st in zip(params, cpu, states):
            g = torch.randn(c.shape, generator=gen) * 0.05
            p.grad = g.to(DEV)
            orang.ranger_step(c, g, st, n_sma_threshold=5, **kw)
        o.step()

    for _ in range(7):
        one_step(opt)
    worst = max(float((p.detach().cpu() - c).abs().max() / c.abs().max().clamp_min(1e-3)) for p, c in zip(params, cpu))
    assert worst < 5e-6, worst
    sd = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in opt.state_dict()["state"][0].items()}
    assert set(sd) == {"step", "exp_avg", "exp_avg_sq", "slow_buffer"} and sd["step"] == 7
    full = opt.state_dict()
    full["state"] = {i: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in s.items()} for i, s in full["state"].items()}
    opt2 = Ranger2020(params, N_sma_threshhold=5, **kw)
    opt2.load_state_dict(full)
    for _ in range(6):  # crosses the lookahead at step 12
        one_step(opt2)
    worst = max(float((p.detach().cpu() - c).abs().max() / c.abs().max().clamp_min(1e-3)) for p, c in zip(params, cpu))
    assert worst < 1e-5, worst
    with pytest.raises(NotImplementedError):
        Ranger2020(params, normloss=True)
    with pytest.raises(ValueError):
        Ranger2020(params, alpha=1.5)


@pytest.mark.parametrize("amp_dtype", [torch.bfloat16, torch.float16])
def test_graphed_train_step_matches_eager(amp_dtype):
    """The whole step (fwd + fused Dice + bwd + capturable Ranger) captured into one hipGraph replays to the same
    parameters as the eager TrainStep (bf16 kernels are deterministic), across the step-5 -> 6 RAdam switch and the
    lookahead sync at step 6.  float16: the reference's autocast + GradScaler loop captured too -- Ranger2020(capturable=True) takes
    the loss scale and the overflow flag as device tensors, so scale(), the inf check, the step and update() are all kernels;
    the eager twin (capturable=False) reads the flag on the host as the reference's loop does."""
    from brats21_amd import get_model
    from brats21_amd.engine import GraphedTrainStep, TrainStep
    from brats21_amd.optim import Ranger2020
    from oracle import synth
    ns = argparse.Namespace(model="equiunet", width=8, norm="group", act="relu", num_classes=3, dropout=0)
    xs = [synth.random_image(1, 4, (16, 16, 16), seed=20 + i).to(DEV) for i in range(8)]
    t = synth.nested_spheres(1, (16, 16, 16)).to(DEV)
    results = []
    for graphed in (False, True):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            m = get_model(ns).to(DEV).train()
        opt = Ranger2020(m.parameters(), lr=1e-2, weight_decay=1e-5, use_gc=True, capturable=graphed)
        step = TrainStep(m, opt, amp=True, amp_dtype=amp_dtype)
        if graphed:
            step = GraphedTrainStep(step, warmup=2)
            # the two warm-up steps run on the first batch: replay the same schedule eagerly below
        losses = []
        seq = [xs[0], xs[0], xs[0]] + xs[1:] if graphed else [xs[0], xs[0], xs[0]] + xs[1:]
        if graphed:
            losses.append(float(step(xs[0], t).detach()))       # 2 eager warm-ups + capture + 1 replay = steps 1..3 on xs[0]
            for x in xs[1:]:
                losses.append(float(step(x, t).detach()))
        else:
            for i, x in enumerate(seq):
                l = float(step(x, t).detach())
                if i >= 2:
                    losses.append(l)
        torch.cuda.synchronize()
        results.append((losses, [p.detach().clone() for p in m.parameters()], opt))
    (l0, p0, o0), (l1, p1, o1) = results
    assert len(l0) == len(l1) == 8
    np.testing.assert_allclose(l0, l1, rtol=1e-5, atol=1e-6)
    worst = max(float((a - b).abs().max()) for a, b in zip(p0, p1))
    assert worst < 1e-5, worst
    assert o1.state_dict()["state"][0]["step"] == 10 == o0.state_dict()["state"][0]["step"]


def test_graphed_train_step_follows_a_learning_rate_schedule():
    """An LR scheduler changes group["lr"] between steps (the reference steps its scheduler once per epoch,
    learning/engine.py:151-155): the captured graph must follow without a re-capture (device-side lr scalar)."""
    from brats21_amd import get_model
    from brats21_amd.engine import GraphedTrainStep, TrainStep
    from brats21_amd.optim import Ranger2020
    from oracle import synth
    ns = argparse.Namespace(model="equiunet", width=8, norm="group", act="relu", num_classes=3, dropout=0)
    x = synth.random_image(1, 4, (16, 16, 16), seed=31).to(DEV)
    t = synth.nested_spheres(1, (16, 16, 16)).to(DEV)
    lrs = [1e-2] * 4 + [4e-3] * 3 + [5e-4] * 3     # steps 1..10; the graph is captured at step 3
    outs = []
    for graphed in (False, True):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            m = get_model(ns).to(DEV).train()
        opt = Ranger2020(m.parameters(), lr=lrs[0], use_gc=True, capturable=graphed)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 1.0)  # (a real scheduler object drives group["lr"])
        step = TrainStep(m, opt, amp=True)
        if graphed:
            step = GraphedTrainStep(step, warmup=2)
        done = 0
        while done < len(lrs):
            for g in opt.param_groups:
                g["lr"] = lrs[done]
            sched.last_epoch += 1
            step(x, t)
            done += 3 if (graphed and done == 0) else 1   # first graphed call = 2 warm-ups + capture replay
        torch.cuda.synchronize()
        outs.append([p.detach().clone() for p in m.parameters()])
    worst = max(float((a - b).abs().max()) for a, b in zip(*outs))
    assert worst < 1e-5, worst
    # and the schedule mattered: a constant-lr run ends elsewhere
    assert lrs[0] != lrs[-1]


def test_graphed_train_step_draws_fresh_dropout_masks_at_every_replay():
    """--dropout p under GraphedTrainStep: the step counter of the mask generator lives on the device and is advanced INSIDE the
    captured step, so every replay of the one hipGraph draws new masks.  With a vanishing learning rate the
    weights do not move: the loss of consecutive replays on the same batch differs only through the masks."""
    import argparse, contextlib, io
    from brats21_amd import get_model, synth
    from brats21_amd.engine import GraphedTrainStep, TrainStep
    from brats21_amd.optim import Ranger2020
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = get_model(argparse.Namespace(model="equiunet", width=16, norm="group", act="relu", num_classes=3, dropout=0.3)).to(dev).train()
        opt = Ranger2020(m.parameters(), lr=1e-12, capturable=True)
    x = synth.random_image(2, 4, (32, 32, 32), seed=5, device=dev)
    t = synth.nested_spheres(2, (32, 32, 32), device=dev)
    step = GraphedTrainStep(TrainStep(m, opt, criterion=None, amp=True), warmup=2)
    losses = [float(step(x, t)) for _ in range(6)]
    c0 = int(m._dropout_state[1])
    losses += [float(step(x, t)) for _ in range(3)]
    assert int(m._dropout_state[1]) == c0 + 3          # the counter moves with the replays
    assert len({round(v, 7) for v in losses[1:]}) >= 6  # ... and so do the masks
    m.eval()
    with torch.no_grad():
        a, b = m(x)[0], m(x)[0]
    assert torch.equal(a, b)


@pytest.mark.parametrize("capturable", [False, True])
@pytest.mark.parametrize("use_gcnorm", [False, True])
def test_ranger_under_gradscaler_without_the_unscale_pass(capturable, use_gcnorm):
    """torch.amp.GradScaler with an optimizer that declares _step_supports_amp_scaling (as torch's fused Adam): the scaler hands
    the loss scale and the overflow flag over as device tensors.  (i) Steps on gradients that still carry the scale 2^16 equal --
    bit for bit -- the steps of a twin optimizer fed the unscaled gradients (the scale is a power of two), both RAdam branches
    and a lookahead step included; (ii) an overflowed step (inf in one gradient) moves nothing: parameters, moments, and -- in
    capturable mode, where the skip is decided on the device -- the device-side step counter; the scale halves; (iii) the next
    clean step continues as if the skipped one had never happened (learning/engine.py:117-122 semantics)."""
    from brats21_amd.optim import Ranger2020
    gen = torch.Generator().manual_seed(11)
    shapes = [(12, 5, 3, 3, 3), (12,), (7, 12, 1, 1, 1), (3000,)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=gen).to(DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    kw = dict(lr=3e-3, alpha=0.5, k=3, betas=(.95, 0.999), eps=1e-5, weight_decay=1e-5, use_gcnorm=use_gcnorm, capturable=capturable)
    with contextlib.redirect_stdout(io.StringIO()):
        oa, ob = Ranger2020(pa, **kw), Ranger2020(pb, **kw)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 16, growth_interval=10 ** 6)
    assert oa._step_supports_amp_scaling
    scaler.scale(torch.zeros(1, device=DEV))  # (the scaler allocates its device scalars at the first scale() call)

    def grads(overflow=False):
        gs = [torch.randn(s, generator=gen) * 0.05 for s in shapes]
        for p, q, g in zip(pa, pb, gs):
            p.grad = (g * 2.0 ** 16).to(DEV)   # what a backward of scaler.scale(loss) leaves
            q.grad = g.to(DEV)
        if overflow:
            pa[2].grad[3, 4] = float("inf")

    def scaled_step():
        scaler.step(oa)   # no scaler.unscale_(): the optimizer reads g / scale itself
        scaler.update()

    for _ in range(7):   # k = 3: lookahead at steps 3 and 6; N_sma crosses its threshold at step 6
        grads()
        scaled_step()
        ob.step()
    for p, q in zip(pa, pb):
        assert torch.equal(p.detach(), q.detach())
    before = [p.detach().clone() for p in pa]
    mom = [oa.state[p]["exp_avg"].clone() for p in pa]
    grads(overflow=True)
    scaled_step()
    assert float(scaler.get_scale()) == 2.0 ** 15
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, pa))
    assert all(torch.equal(a, oa.state[p]["exp_avg"]) for a, p in zip(mom, pa))
    if capturable:
        oa.sync_steps()
    assert oa.state[pa[0]]["step"] == 7, oa.state[pa[0]]["step"]
    # the clean step after the skipped one: the twin simply takes its 8th step (scale now 2^15)
    gs = [torch.randn(s, generator=gen) * 0.05 for s in shapes]
    for p, q, g in zip(pa, pb, gs):
        p.grad, q.grad = (g * 2.0 ** 15).to(DEV), g.to(DEV)
    scaled_step()
    ob.step()
    for p, q in zip(pa, pb):
        assert torch.equal(p.detach(), q.detach())


def test_capturable_ranger_new_plan_after_a_skipped_step_starts_from_the_device_step_count():
    """in capturable mode an overflow-skipped step advances state['step'] on the host but not the device counter.  A
    plan built afterwards (the set of parameters with a gradient changed) must be seeded from the device's count -- otherwise
    RAdam's rectification and the lookahead phase shift by one.  The twin never saw the skipped step."""
    from brats21_amd.optim import Ranger2020
    gen = torch.Generator().manual_seed(5)
    shapes = [(12, 5, 3, 3, 3), (12,), (7, 12, 1, 1, 1), (3000,)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=gen).to(DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    kw = dict(lr=3e-3, alpha=0.5, k=3, betas=(.95, 0.999), eps=1e-5, weight_decay=1e-5, capturable=True)
    with contextlib.redirect_stdout(io.StringIO()):
        oa, ob = Ranger2020(pa, **kw), Ranger2020(pb, **kw)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10, growth_interval=10 ** 6)
    scaler.scale(torch.zeros(1, device=DEV))

    def grads(scale, skip_last=False, overflow=False):
        gs = [torch.randn(s, generator=gen) * 0.05 for s in shapes]
        for i, (p, q, g) in enumerate(zip(pa, pb, gs)):
            p.grad, q.grad = ((g * scale).to(DEV), g.to(DEV)) if not (skip_last and i == 3) else (None, None)
        if overflow:
            pa[0].grad[0, 0, 0, 0, 0] = float("inf")

    for _ in range(4):
        grads(2.0 ** 10)
        scaler.step(oa); scaler.update(); ob.step()
    grads(2.0 ** 10, overflow=True)
    scaler.step(oa); scaler.update()          # skipped on the device; the host count of `oa` is now 5, the device's 4
    assert float(scaler.get_scale()) == 2.0 ** 9
    for _ in range(3):                        # a NEW plan (3 of the 4 parameters): steps 5, 6 (lookahead at 6: k = 3), 7
        grads(2.0 ** 9, skip_last=True)
        scaler.step(oa); scaler.update(); ob.step()
    for i, (p, q) in enumerate(zip(pa, pb)):
        assert torch.equal(p.detach(), q.detach()), i
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    assert [sa[i]["step"] for i in range(3)] == [7, 7, 7] == [sb[i]["step"] for i in range(3)]
