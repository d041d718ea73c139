"""-m "not gpu": the multi-GPU path (brats21_amd.ddp) on CPU with the gloo backend, world_size 2:
bucketed gradient averaging (both the post-backward path and the in-backward push() path used by the
accelerated models), bucket planning, patient sharding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from brats21_amd.ddp import GradientBuckets, shard_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(6, 5)
        self.b = nn.Linear(5, 3)
        self.unused = nn.Parameter(torch.ones(4))  # statically unused, like EvoNorm `v`
        self._grad_sink = None

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, use_push, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _Net()  # identical initial weights on every rank (seeded inside)
    torch.manual_seed(100 + rank)  # ... but a different data shard per rank
    buckets = GradientBuckets(net, bucket_bytes=64)  # tiny buckets -> several all-reduces
    out = []
    for step in range(3):
        x = torch.randn(4, 6)
        net.zero_grad(set_to_none=True)
        loss = net(x).pow(2).mean()
        loss.backward()
        local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        if use_push and step > 0:
            # emulate the accelerated models: the backward program pushes finished gradients (reverse order)
            for idx in reversed(range(len(local))):
                if local[idx] is not None:
                    net._grad_sink(idx, local[idx])
        buckets.finish()
        # reference: explicit all-reduce of the local gradients
        for p, g in zip(net.parameters(), local):
            if g is None:
                assert p.grad is None
                continue
            dist.all_reduce(g)
            torch.testing.assert_close(p.grad, g / world, atol=1e-7, rtol=1e-6)
        out.append(float(loss.detach()))
    assert len(buckets._plan) > 1 and buckets.payload_bytes() == 4 * sum(p.numel() for p in (net.a.weight, net.a.bias, net.b.weight, net.b.bias))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def _accum_worker(rank, world, port, mode, q):
    """Gradient accumulation (the reference's --gradient_accumulation_iter, learning/engine.py:119-130: no zero_grad
    between micro-batches) through push(): the averaged result must be the mean over ranks of the SUM over micro-batches."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _Net()
    torch.manual_seed(200 + rank)
    buckets = GradientBuckets(net, bucket_bytes=64)
    params = list(net.parameters())

    def backward_with_push(x):
        """what the accelerated models do: the backward program pushes each finished gradient, THEN autograd
        accumulates it into p.grad"""
        loss = net(x).pow(2).mean()
        grads = torch.autograd.grad(loss, [p for p in params if p is not net.unused])
        it = iter(grads)
        local = [None if p is net.unused else next(it) for p in params]
        for idx in reversed(range(len(local))):
            if local[idx] is not None:
                net._grad_sink(idx, local[idx])
        for p, g in zip(params, local):
            if g is not None:
                p.grad = g.clone() if p.grad is None else p.grad.add_(g)
        return local

    for outer in range(3):  # outer step 0 learns the bucket order, 1 and 2 use the push path
        net.zero_grad(set_to_none=(mode != "zero_in_place"))
        total = [None] * len(params)
        micro = 1 if mode == "zero_in_place" else 3
        for mb in range(micro):
            x = torch.randn(4, 6)
            if mode == "no_sync" and mb < micro - 1:
                with buckets.no_sync():
                    local = backward_with_push(x)
                    buckets.finish()  # a no-op inside no_sync
            else:
                local = backward_with_push(x)
                if mode == "finish_each":
                    buckets.finish()
            for i, g in enumerate(local):
                if g is not None:
                    total[i] = g.clone() if total[i] is None else total[i] + g
        if mode != "finish_each":
            buckets.finish()
        for p, g in zip(params, total):
            if g is None:
                assert p.grad is None
                continue
            dist.all_reduce(g)
            torch.testing.assert_close(p.grad, g / world, atol=1e-6, rtol=1e-5)
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


@pytest.mark.parametrize("mode", ["finish_once", "finish_each", "no_sync", "zero_in_place"])
def test_gradient_accumulation_gloo_world2(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


@pytest.mark.parametrize("use_push", [False, True])
def test_gradient_buckets_gloo_world2(use_push):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, use_push, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got[0][0] == 0 and got[1][0] == 1 and got[0][1] != got[1][1]  # different shards -> different losses


def _bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _Net()
    torch.manual_seed(300 + rank)
    buckets = GradientBuckets(net, bucket_bytes=64, comm_dtype=torch.bfloat16)
    for step in range(3):
        net.zero_grad(set_to_none=True)
        net(torch.randn(4, 6)).pow(2).mean().backward()
        local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        if step > 0:
            for idx in reversed(range(len(local))):
                if local[idx] is not None:
                    net._grad_sink(idx, local[idx])
        buckets.finish()
        for p, g in zip(net.parameters(), local):
            if g is None:
                continue
            assert p.grad.dtype == torch.float32  # the optimizer still sees f32 gradients
            ref = g.bfloat16()
            dist.all_reduce(ref)  # what a bf16 transport computes: rounded per rank, summed in bf16
            torch.testing.assert_close(p.grad, ref.float() / world, atol=1e-6, rtol=1e-5)
            dist.all_reduce(g)
            torch.testing.assert_close(p.grad, g / world, atol=2e-2 * float(g.abs().max()) / world + 1e-6, rtol=2e-2)
    assert buckets.payload_bytes() == 2 * sum(p.numel() for p in (net.a.weight, net.a.bias, net.b.weight, net.b.bias))
    assert buckets.allreduce_ms(2) >= 0.0
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_bf16_transport_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_patient_sharding():
    world = 8
    shards = [shard_indices(21, r, world, epoch=3) for r in range(world)]
    assert all(len(s) == 3 for s in shards)                        # ceil(21/8) each, ranks stay in step
    assert set(i for s in shards for i in s) == set(range(21))     # every patient seen
    assert shard_indices(21, 0, world, epoch=3) == shards[0]       # deterministic
    assert shard_indices(21, 0, world, epoch=4) != shards[0]       # reshuffled per epoch
    assert shard_indices(5, 1, 2, shuffle=False) == [1, 3, 0]      # wrap-around padding
