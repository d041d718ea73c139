"""ORACLE (test infrastructure): CPU restatement of the reference's Ranger2020 optimizer
(learning/optimizer.py:62-255: RAdam with the N_sma threshold + gradient centralisation + lookahead),
as a pure function over explicit state.  Pinned by tests/golden/ranger.npz, produced by the reference's
own class run in this container (tests/golden/make_golden.py: ranger_fixture), including use_gcnorm
(learning/optimizer.py:23-36,189-190; off by default).  normloss (:39-59,192-198) is not restated: the reference's own
step() raises there ("a leaf Variable that requires grad is being used in an in-place operation", :198 -- step() runs
without torch.no_grad()), so there is no behaviour to pin."""
import math

import torch


def radam_step_size(step, beta1, beta2, n_sma_threshold):
    """learning/optimizer.py:198-214 -> (N_sma > threshold, step_size)."""
    beta2_t = beta2 ** step
    n_max = 2 / (1 - beta2) - 1
    n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma > n_sma_threshold:
        ss = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) / (
            1 - beta1 ** step)
        return True, ss
    return False, 1.0 / (1 - beta1 ** step)


def new_state(p):
    return {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p), "slow_buffer": p.clone()}


def ranger_step(p, grad, state, lr=1e-3, alpha=0.5, k=6, n_sma_threshold=5, betas=(0.95, 0.999), eps=1e-5,
                weight_decay=0.0, use_gc=True, gc_conv_only=False, use_gcnorm=False):
    """One Ranger2020 update of tensor ``p`` (in place on p and state); returns p."""
    g = grad.clone().float()
    if use_gc and g.dim() > (3 if gc_conv_only else 1):               # :11-20, gc_loc=True (:186-187)
        g = g - g.mean(dim=tuple(range(1, g.dim())), keepdim=True)
    if use_gcnorm and g.numel() > 2:                                    # :23-36 with use_channels=False (:189-190)
        g = g / (g.std() + 1e-8)
    beta1, beta2 = betas
    state["step"] += 1
    state["exp_avg_sq"].mul_(beta2).addcmul_(g, g, value=1 - beta2)     # :192-193
    state["exp_avg"].mul_(beta1).add_(g, alpha=1 - beta1)               # :195-196
    adaptive, step_size = radam_step_size(state["step"], beta1, beta2, n_sma_threshold)
    if adaptive:
        gg = state["exp_avg"] / (state["exp_avg_sq"].sqrt() + eps)       # :217-219
    else:
        gg = state["exp_avg"]                                            # :220-221 (an alias, not a copy)
    if weight_decay != 0:
        gg.add_(p, alpha=weight_decay)                                   # :222-223 (pollutes exp_avg when aliased)
    p.add_(gg, alpha=-step_size * lr)                                    # :228
    if state["step"] % k == 0:                                           # :233-240
        state["slow_buffer"].add_(p - state["slow_buffer"], alpha=alpha)
        p.copy_(state["slow_buffer"])
    return p
