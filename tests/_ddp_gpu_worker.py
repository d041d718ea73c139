"""Worker of tests/test_ddp_gpu.py::test_two_ranks_on_one_gpu: launched by torch.distributed.run (2 ranks, gloo backend,
both ranks on cuda:0 -- RCCL refuses two ranks on one device) from tests/conftest.py BEFORE the pytest process touches the
GPU.  Drives the REAL backward programs of both models through GradientBuckets.push and checks the averaged gradients
against a manual all-reduce of the bucket-less gradients of an identical replica.  Rank 0 writes a JSON verdict."""
import argparse
import contextlib
import io
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_path = sys.argv[1]
    from brats21_amd import get_model
    from brats21_amd.ddp import GradientBuckets, init_process_group_from_env
    from brats21_amd.engine import TrainStep
    from brats21_amd.optim import Ranger2020
    from oracle import synth, unet
    os.environ["BRATS_DIST_BACKEND"] = "gloo"
    rank, world, _ = init_process_group_from_env()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    res = {"world": world, "cases": {}}
    for name, width, size in (("equiunet", 8, (16, 16, 16)), ("equiunet_assp_evo", 16, (16, 16, 16))):
        ns = argparse.Namespace(model=name, width=width, norm="group", act="relu", num_classes=3, dropout=0)

        def make():
            torch.manual_seed(0)  # identical replicas on every rank
            with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = get_model(ns).to(dev).train()
            m.precision = "fp32"
            return m
        m, ref = make(), make()
        buckets = GradientBuckets(m, bucket_bytes=1 << 16)  # small buckets: several all-reduces in flight during backward
        x = synth.random_image(2, 4, size, seed=100 + rank).to(dev)  # a different shard per rank
        t = synth.nested_spheres(2, size).to(dev)
        worst = 0.0
        for step in range(3):  # step 0 learns the bucket order (reduce in finish()), 1 and 2 push from the backward program
            for mod in (m, ref):
                mod.zero_grad(set_to_none=True)
                out = mod(x)
                unet.deep_supervision_loss(out, t).backward()
            buckets.finish()
            for (k, p), q in zip(m.named_parameters(), ref.parameters()):
                if q.grad is None:
                    assert p.grad is None, k
                    continue
                g = q.grad.clone()
                dist.all_reduce(g)
                g /= world
                err = float((p.grad - g).abs().max())
                worst = max(worst, err / (float(g.abs().max()) + 1e-30))
                assert torch.allclose(p.grad, g, rtol=1e-6, atol=1e-7 * float(g.abs().max()) + 1e-12), (name, step, k, err)
        assert len(buckets._plan) > 1
        # and a few full training steps keep the replicas bit-identical (same averaged gradients -> same Ranger update)
        with contextlib.redirect_stdout(io.StringIO()):
            opt = Ranger2020(m.parameters(), lr=1e-3, use_gc=False)
        ts = TrainStep(m, opt, amp=False, buckets=buckets)
        losses = [float(ts(x, t).detach()) for _ in range(3)]
        flat = torch.cat([p.detach().flatten() for p in m.parameters()])
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert all(torch.equal(both[0], b) for b in both[1:]), "replicas diverged"
        res["cases"][name] = {"buckets": len(buckets._plan), "payload_bytes": buckets.payload_bytes(), "worst_rel_err": worst,
                              "losses_rank%d" % rank: losses}
    dist.barrier()
    if rank == 0:
        res["ok"] = True
        res["finished"] = time.strftime("%Y-%m-%d %H:%M:%S")
        with open(out_path + ".tmp", "w") as f:
            json.dump(res, f)
        os.replace(out_path + ".tmp", out_path)
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:  # noqa: BLE001 -- the verdict file must say why
        import traceback
        if os.environ.get("RANK", "0") == "0" or True:
            with open(sys.argv[1] + ".err%s" % os.environ.get("RANK", "0"), "w") as f:
                f.write(traceback.format_exc())
        raise
