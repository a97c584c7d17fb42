"""world_size-2 and -4 gloo tests (CPU) of the multi-rank plumbing of the path: AllGather_multi forward/backward semantics
(reference trainer/trainer_egoclip.py:25-41) and the scalar-gather form of the MLM/ITM loss reductions, which must give
the same loss and the same local gradients as gathering the logits (reference model.py:411-418)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import types
    from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi
    args = types.SimpleNamespace(world_size=world, rank=rank)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(3, 5, generator=g, requires_grad=True)
    y = AllGather_multi.apply(x, world, args)
    ok = y.shape == (3 * world, 5) and torch.equal(y[3 * rank:3 * rank + 3], x.detach())
    w = torch.arange(3 * world * 5, dtype=torch.float32).reshape(3 * world, 5)
    (y * w).sum().backward()
    ok = ok and torch.equal(x.grad, w[3 * rank:3 * rank + 3])       # local slice only, no reduction

    # scalar-gather CE == logits-gather CE (loss value and local gradient)
    V = 11
    logits = torch.randn(4, V, generator=g, requires_grad=True)
    labels = torch.randint(0, V, (4,), generator=g)
    labels[rank] = -100
    lg_all = AllGather_multi.apply(logits, world, args)
    lb_all = AllGather_multi.apply(labels, world, args)
    ref = torch.nn.functional.cross_entropy(lg_all, lb_all, ignore_index=-100)
    (g_ref,) = torch.autograd.grad(ref, logits)
    s = torch.nn.functional.cross_entropy(logits, labels, ignore_index=-100, reduction='sum')
    cnt = (labels != -100).sum().float()
    tot = AllGather_multi.apply(torch.stack([s, cnt]).reshape(1, 2), world, args)
    mine = tot[:, 0].sum() / tot[:, 1].sum()
    (g_mine,) = torch.autograd.grad(mine, logits)
    ok = ok and torch.allclose(ref, mine, atol=1e-6) and torch.allclose(g_ref, g_mine, atol=1e-7)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_allgather_and_loss_reduction(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + 7 * world + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def _exchange_worker(rank, world, port, q):
    """ExchangeClipsFn (trainer/exchange.py) against an all-gather of the same tokens: forward values, and the gradient each
    owner receives = sum of what the requesters computed on the tokens they fetched."""
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from egovlpv2_amd.trainer.exchange import gather_requests, ExchangeClipsFn
    bsz, rows, d = 4, 3, 5
    ok = True
    # three rounds: both ranks request, only rank 0 requests, nobody requests (a rank that asks for nothing still serves)
    draws = [{0: [0, 5, 2, 7], 1: [4, 1, 6, 1]}, {0: [6, 1, 2, 6], 1: [4, 5, 6, 7]}, {0: [0, 1, 2, 3], 1: [4, 5, 6, 7]}]
    if world > 2:
        # more than one peer: a rank fetches from several owners and an owner serves several requesters (two random rounds), only
        # the last rank requests, nobody requests
        own = {r: list(range(r * bsz, (r + 1) * bsz)) for r in range(world)}
        rand = [{r: torch.randint(0, world * bsz, (bsz,), generator=torch.Generator().manual_seed(100 * k + r)).tolist() for r in range(world)}
                for k in range(2)]
        last = dict(own)
        last[world - 1] = [0, bsz, 2 * bsz + 1, 1]
        draws = rand + [last, own]
    for rnd, dr in enumerate(draws):
        g = torch.Generator().manual_seed(10 * rnd + rank)
        x = torch.randn(bsz * rows, d, generator=g, requires_grad=True)
        table = gather_requests(dr[rank], rank, bsz, world)
        ok = ok and table == [sorted({j for j in dr[r] if not r * bsz <= j < (r + 1) * bsz}) for r in range(world)]
        got = ExchangeClipsFn.apply(x, table, rank, bsz, rows)
        allx = [torch.empty(bsz * rows, d) for _ in range(world)]
        dist.all_gather(allx, x.detach())
        allx = torch.cat(allx, 0)
        want = torch.cat([allx[j * rows:(j + 1) * rows] for j in table[rank]], 0) if table[rank] else torch.empty(0, d)
        ok = ok and got.shape == want.shape and torch.equal(got, want)
        w = torch.randn(got.shape, generator=g)
        (got * w).sum().backward()                     # every rank runs backward (a zero-size gradient still serves the others)
        # reference: gather every requester's weights and add them on the owner's rows
        sizes = [len(t) * rows for t in table]
        ws = []
        for r in range(world):
            buf = w.clone() if r == rank else torch.empty(sizes[r], d)
            if sizes[r]:
                dist.broadcast(buf, src=r)
            ws.append(buf)
        ref = torch.zeros(bsz * rows, d)
        for r in range(world):
            for n, j in enumerate(table[r]):
                if j // bsz == rank:
                    i = j - rank * bsz
                    ref[i * rows:(i + 1) * rows] += ws[r][n * rows:(n + 1) * rows]
        ok = ok and x.grad is not None and torch.allclose(x.grad, ref, atol=1e-6)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_negative_clip_token_exchange(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29950 + 7 * world + (os.getpid() % 300)
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def _wire_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from egovlpv2_amd.trainer.grad_sync import allreduce_bf16_wire
    ok = True
    for n in (1, 7, 4096, 100003):                                    # ragged sizes: the last shard is padded
        locals_ = [torch.randn(n, generator=torch.Generator().manual_seed(7 * n + r)) * (1 + r) for r in range(world)]
        flat = locals_[rank].clone()
        allreduce_bf16_wire(flat)
        # what the option promises: each element = bf16(sum over ranks, in rank order, of bf16(local)), the same bits on every rank
        want = torch.stack([t.to(torch.bfloat16).float() for t in locals_]).sum(0).to(torch.bfloat16).float()
        ok = ok and torch.equal(flat, want)
        exact = torch.stack(locals_).sum(0)
        ok = ok and float((flat - exact).norm() / exact.norm()) < 2 ** -7
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        ok = ok and all(torch.equal(b, flat) for b in both)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_bf16_wire_gradient_sum(world):
    """trainer/grad_sync.py::allreduce_bf16_wire (FlatGradSync(wire='bf16')): bf16 on the links, fp32 accumulation on
    arrival, identical bits on every rank."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29300 + 7 * world + (os.getpid() % 250)
    procs = [ctx.Process(target=_wire_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def _egomcq_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from egovlpv2_amd.trainer.validate import EgoMCQAccumulator
    g = torch.Generator().manual_seed(7)                      # the same stream on every rank: question (step, r) is row step * world + r
    steps, b2 = 6, 5
    vtc = torch.randn(steps * world, b2, generator=g)
    vtm = torch.rand(steps * world, b2, generator=g)
    gt = torch.randint(0, b2, (steps * world,), generator=g)
    ty = torch.randint(1, 3, (steps * world,), generator=g)
    acc = EgoMCQAccumulator()
    for s_ in range(steps):
        i = s_ * world + rank
        acc.add({'vtc': vtc[i:i + 1], 'vtm': vtm[i:i + 1], 'ensemble': vtc[i:i + 1] + vtm[i:i + 1]}, gt[i:i + 1], ty[i:i + 1])
    a = acc.arrays()
    ok = torch.equal(a['gt'], gt) and torch.equal(a['type'], ty) and torch.equal(a['vtm'], vtm) and torch.equal(a['ensemble'], vtc + vtm)
    m = acc.metrics()
    from egovlpv2_amd.model.metric import egomcq_accuracy_metrics_ensemble, egomcq_accuracy_metrics_vtm
    ok = ok and m['egomcq_accuracy_metrics_ensemble'] == egomcq_accuracy_metrics_ensemble(vtc + vtm, gt, ty)
    ok = ok and m['egomcq_accuracy_metrics_vtm'] == egomcq_accuracy_metrics_vtm(vtm, gt, ty)
    ok = ok and set(m['egomcq_accuracy_metrics_vtm']) == {'Inter-video', 'Intra-video'}
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_egomcq_validation_gathers_world2():
    """trainer_egoclip.py:250-291: per batch every rank contributes one question; ground truth, the two score arrays and the question
    types are all-gathered in rank order and the metrics of the concatenation equal those of a single process over all questions."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 300)
    procs = [ctx.Process(target=_egomcq_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_egomcq_accumulator_single_process():
    """without a process group the accumulator keeps the local arrays (the 1-GPU validation run)"""
    sys.path.insert(0, REPO)
    from egovlpv2_amd.trainer.validate import EgoMCQAccumulator
    acc = EgoMCQAccumulator()
    vtc = torch.tensor([[0.1, 0.9, 0.0, 0.0, 0.0], [0.5, 0.1, 0.1, 0.1, 0.1]])
    vtm = torch.tensor([[0.2, 0.1, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.9]])
    acc.add({'vtc': vtc, 'vtm': vtm}, torch.tensor([1, 0]), torch.tensor([1, 2]))
    m = acc.metrics()
    assert m['egomcq_accuracy_metrics_ensemble'] == {'Inter-video': 100.0, 'Intra-video': 0.0}
    assert m['egomcq_accuracy_metrics_vtm'] == {'Inter-video': 0.0, 'Intra-video': 0.0}
