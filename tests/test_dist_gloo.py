"""world_size-2 gloo tests (CPU) of the multi-process plumbing of the data-parallel path: the single flat gradient
all-reduce, the SyncBatchNorm statistics exchange and the bench's max-over-ranks timing reduction.  The kernels
themselves need a GPU; here only the torch.distributed call pattern runs, on CPU tensors."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cavp_amd.train import GradArena, allreduce_arena, dist_world
        assert dist_world() == world
        # 1) one flat arena, one collective: local grads are pre-scaled by 1/world (train_step), SUM == mean
        params = [nn.Parameter(torch.zeros(5, 3)), nn.Parameter(torch.zeros(7)), nn.Parameter(torch.zeros(2, 2, 3, 3))]
        frozen = nn.Parameter(torch.zeros(4), requires_grad=False)
        arena = GradArena(params + [frozen], "cpu")
        assert id(frozen) not in arena.views and arena.flat.numel() % 4 == 0
        for i, p in enumerate(params):
            arena.views[id(p)].fill_((rank + 1) * (i + 1) / world)
        allreduce_arena(arena)
        for i, p in enumerate(params):
            expect = sum((r + 1) * (i + 1) for r in range(world)) / world
            assert torch.allclose(arena.views[id(p)], torch.full(p.shape, expect)), (rank, i)
        # 1b) two-piece reduction used to overlap the collective with the backward: parameters flagged "late" are laid out
        #     first, `flat[split:]` (early-final gradients) is reduced asynchronously, `flat[:split]` at the end
        from cavp_amd.train import allreduce_arena_early, allreduce_arena_late
        late = {id(params[1])}
        ar2 = GradArena(params + [frozen], "cpu", late_ids=late)
        assert ar2.params[0] is params[1] and ar2.split == 8 and ar2.flat.numel() == arena.flat.numel()
        for i, p in enumerate(params):
            ar2.views[id(p)].fill_((rank + 1) * (i + 1) / world)
        work = allreduce_arena_early(ar2)
        assert work is not None
        allreduce_arena_late(ar2, work)
        for i, p in enumerate(params):
            expect = sum((r + 1) * (i + 1) for r in range(world)) / world
            assert torch.allclose(ar2.views[id(p)], torch.full(p.shape, expect)), (rank, i, "split")
        for i, p in enumerate(params):
            ar2.views[id(p)].fill_(float(rank + i))
        allreduce_arena_late(ar2, None)          # no early piece started: one collective over everything
        assert torch.allclose(ar2.views[id(params[2])], torch.full(params[2].shape, float(sum(r + 2 for r in range(world)))))
        # 1c) opt-in bf16 wire format (240 MB instead of 479 MB per step over xGMI): each piece is cast to bf16, summed over the
        #     ranks in bf16 and cast back into the f32 arena - one-piece and two-piece (asynchronous early range) forms
        from cavp_amd.train import grad_allreduce_dtype, set_grad_allreduce_dtype
        set_grad_allreduce_dtype(torch.bfloat16)
        try:
            assert grad_allreduce_dtype() == torch.bfloat16
            gen = torch.Generator().manual_seed(100 + rank)
            mine = {i: torch.randn(p.shape, generator=gen) / world for i, p in enumerate(params)}
            for async_two_piece in (False, True):
                for i, p in enumerate(params):
                    ar2.views[id(p)].copy_(mine[i])
                if async_two_piece:
                    allreduce_arena_late(ar2, allreduce_arena_early(ar2))
                else:
                    allreduce_arena(ar2)
                assert ar2.wire.dtype == torch.bfloat16 and ar2.flat.dtype == torch.float32
                for i, p in enumerate(params):
                    parts = []
                    for r in range(world):   # every rank's contribution, regenerated from its seed
                        gr = torch.Generator().manual_seed(100 + r)
                        parts.append({j: torch.randn(q_.shape, generator=gr) / world for j, q_ in enumerate(params)}[i])
                    exact = sum(parts)
                    got = ar2.views[id(p)]
                    # rounding: each contribution to bf16 (2^-9 relative), the sum once more
                    tol = 2.0 ** -7 * sum(t.abs() for t in parts) + 1e-6
                    assert bool(((got - exact).abs() <= tol).all()), (rank, i, async_two_piece, float((got - exact).abs().max()))
                    assert not torch.equal(got, exact) or exact.numel() < 4   # it really went through bf16
        finally:
            set_grad_allreduce_dtype(torch.float32)
        # views alias the flat buffer (p.grad = view => the optimiser sees the reduced values with no copy)
        arena.zero()
        assert float(arena.views[id(params[0])].abs().sum()) == 0.0
        # 2) SyncBatchNorm forward: ONE exchange of (mean, M2) per layer; Chan combine over ranks == full-batch moments
        from cavp_amd.train import gather_bn_moments
        g = torch.Generator().manual_seed(7)
        full = torch.randn(world * 6, 3, generator=g) * 2 + 5
        mine = full[rank * 6:(rank + 1) * 6]
        loc = torch.stack((mine.mean(0), ((mine - mine.mean(0)) ** 2).sum(0)), -1)
        allm = gather_bn_moments(loc)
        assert allm.shape == (world, 3, 2) and torch.equal(allm[rank], loc)
        mean = allm[:, :, 0].mean(0)
        m2 = (allm[:, :, 1] + 6 * (allm[:, :, 0] - mean) ** 2).sum(0)
        assert torch.allclose(mean, full.mean(0), atol=1e-5) and torch.allclose(m2 / full.shape[0], full.var(0, unbiased=False), atol=1e-4)
        # 3) bench.py timing reduction: MAX over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == float(world)
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_process_gloo_collectives():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
