"""Shared replay of the closed-form Ranger2020 schedule (CPU oracle test and GPU optimizer test)."""
import json
import os

import numpy as np

from oracle import synth


def ranger_replay(golden_dir, make_param, step_fn, read, rtol=2e-6):
    """Replays the 13-step closed-form schedule of tests/golden/make_golden.py::ranger_fixture."""
    g = np.load(os.path.join(golden_dir, "ranger.npz"))
    meta = json.loads(str(g["meta"]))
    shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
    for case, kw in meta["cases"].items():
        params = {n: make_param(synth.closed_form("rp." + n, s)) for n, s in shapes.items()}
        ctx = step_fn(params, meta["lr"], kw)
        for step in range(1, meta["steps"] + 1):
            grads = {n: (None if n == "unused" else synth.closed_form(f"rg.{n}.{step}", shapes[n], 0.1 * step)) for n in shapes}
            ctx["step"](grads)
            if step in (5, 6, 13):
                for n in shapes:
                    np.testing.assert_allclose(read(params[n]), g[f"{case}.{step}.{n}"], rtol=rtol, atol=2e-7,
                                               err_msg=f"{case} step {step} {n}")
        for n in shapes:
            if n != "unused":
                st = ctx["state"](n)
                np.testing.assert_allclose(read(st["exp_avg"]), g[f"{case}.exp_avg.{n}"], rtol=rtol, atol=1e-8)
                np.testing.assert_allclose(read(st["exp_avg_sq"]), g[f"{case}.exp_avg_sq.{n}"], rtol=rtol, atol=1e-10)
                np.testing.assert_allclose(read(st["slow_buffer"]), g[f"{case}.slow.{n}"], rtol=rtol, atol=2e-7)
