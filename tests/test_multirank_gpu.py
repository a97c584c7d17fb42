"""Two ranks of the HIP path on ONE GPU (gloo rendezvous, both processes on cuda:0): the world_size-2 semantics of
FrozenInTime.forward -- global EgoNCE matrix, scalar-gathered MLM/ITM losses, hard negatives owned by the other rank
(pixels gathered, shared video prefix for own clips) -- against the CPU oracle run under the same process group, and
DDP(static_graph) and flat-buffer gradient averaging over several steps, on the two-stream schedule (`overlap=False` stays available to the worker).  fp32 storage; tolerances 1e-3 on losses, 5e-3 on gradients."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


class _HostGather(torch.autograd.Function):
    """AllGather_multi semantics (trainer_egoclip.py:25-41) staged through the host so that it works on gloo with device
    tensors: forward = concatenation over ranks, backward = the local slice of the incoming gradient."""

    group = None            # gloo group of the host staging (None: the default group of the one-GPU rehearsal)

    @staticmethod
    def forward(ctx, t, n_gpu, args):
        ctx.rank, ctx.b = args.rank, t.shape[0]
        c = t.detach().cpu().contiguous()
        out = [torch.empty_like(c) for _ in range(args.world_size)]
        dist.all_gather(out, c, group=_HostGather.group)
        return torch.cat(out, 0).to(t.device)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.b * ctx.rank: ctx.b * (ctx.rank + 1)], None, None


def _worker(rank, world, port, q, steps, overlap, gsync='ddp', backend='gloo'):
    try:
        if not overlap:
            os.environ['EGV_NO_OVERLAP'] = '1'
        sys.path.insert(0, REPO)
        sys.path.insert(0, os.path.join(REPO, 'tests'))
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        # backend 'gloo': both ranks on cuda:0 (the one-GPU rehearsal; collectives staged through the host).  backend 'nccl': one GPU per
        # rank on RCCL -- the product transport: AllGather_multi itself, DDP / FlatGradSync over xGMI -- with a gloo group beside it
        # for the CPU oracle's gathers
        rccl = backend == 'nccl'
        dev_index = rank if rccl else 0
        host_group = None
        if rccl:
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            torch.cuda.set_device(dev_index)
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev_index))
            host_group = dist.new_group(backend='gloo')
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        _HostGather.group = host_group
        import types
        from helpers import load_golden, oracle_setup
        from oracle import ref_model as O
        from egovlpv2_amd.model.model import FrozenInTime
        from egovlpv2_amd.model.loss import EgoNCE
        from egovlpv2_amd.synthetic import make_batch
        torch.cuda.set_device(dev_index)
        _, cfg, B, L, wseed, _ = load_golden('tiny')
        B = 4                                              # 2 negatives per rank and step
        sd, _, _, _, oc = oracle_setup(cfg, B, L, wseed, 0, requires_grad=True)
        m = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True},
                         {'model': 'roberta-base', 'pretrained': True, 'input': 'text'},
                         path_config=cfg, task_names='EgoNCE_MLM_ITM', compute_dtype=torch.float32)
        m.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True)
        m = m.cuda()
        flat = None
        if gsync == 'ddp':
            from torch.nn.parallel import DistributedDataParallel as DDP
            net = DDP(m, device_ids=[dev_index], static_graph=True, gradient_as_bucket_view=True, find_unused_parameters=False)
        else:                                              # trainer/grad_sync.py: all-reduce of the flat per-block gradient buffers
            from egovlpv2_amd.trainer.grad_sync import FlatGradSync
            net, flat = m, FlatGradSync(m)
        args = types.SimpleNamespace(world_size=world, rank=rank)
        names = [n for n, _ in m.named_parameters()]
        worst = {'loss': 0.0, 'grad': 0.0, 'remote': 0, 'no_remote': 0}
        for step in range(steps):
            data, noun, verb = make_batch(cfg, B, L, 500 + 10 * step + rank)
            dev = {'video': data['video'].cuda(), 'text': {k: v.cuda() for k, v in data['text'].items()},
                   'text_mlm_ids': data['text_mlm_ids'].cuda(), 'text_mlm_labels': data['text_mlm_labels'].cuda()}
            np.random.seed(40 + step + rank)
            torch.manual_seed(40 + step + rank)
            net.zero_grad(set_to_none=True)
            if rccl:
                from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi
            loss, ld, ret = net(dev, noun.cuda(), verb.cuda(), AllGather_multi.apply if rccl else _HostGather.apply, world, args,
                                {'loss': {'type': 'EgoNCE'}}, EgoNCE(), dev_index, task_names='EgoNCE_MLM_ITM')
            if flat is not None:
                flat.backward(loss)
            else:
                loss.backward()
            torch.cuda.synchronize()
            # oracle, same rank, same RNG stream, gathers over the same group
            np.random.seed(40 + step + rank)
            torch.manual_seed(40 + step + rank)
            for v in sd.values():
                v.grad = None
            oloss, old, oret = O.forward_losses(sd, data, noun, verb, oc, 'EgoNCE_MLM_ITM',
                                                world={'rank': rank, 'gather': lambda t: _HostGather.apply(t, world, args)})
            oloss.backward()
            assert [x for x in ret['_itm_neg_log']] == [x for x in oret['_itm_neg_log']], (ret['_itm_neg_log'], oret['_itm_neg_log'])
            lo = rank * B
            rem = [j for (_, kind, j) in ret['_itm_neg_log'] if kind == 'video' and not lo <= j < lo + B]
            worst['remote' if rem else 'no_remote'] += 1
            for k in ('EgoNCE', 'loss_mlm', 'loss_itm', 'loss_total'):
                a, r = float(ld[k]), float(old[k])
                worst['loss'] = max(worst['loss'], abs(a - r) / abs(r))
            # DDP averaged the HIP gradients over ranks; average the oracle's the same way
            for n in names:
                g = sd[n].grad
                g = torch.zeros_like(sd[n]) if g is None else g.detach().clone()
                dist.all_reduce(g, group=host_group)
                g /= world
                a = dict(m.named_parameters())[n].grad.double().cpu().reshape(-1)
                r = g.double().reshape(-1)
                if n.endswith('.key.bias'):
                    continue                                    # true gradient is exactly zero (softmax shift invariance)
                err = (a - r).norm().item() / (r.norm().item() + 1e-6)
                if err > worst['grad']:
                    worst['grad'], worst['grad_at'] = err, f'{n} (step {step})'
                if err > 5e-3:
                    worst.setdefault('bad', []).append((n, step, round(err, 4)))
        q.put((rank, 'ok', worst))
        dist.destroy_process_group()
    except Exception as e:                                        # surface the failure in the parent
        import traceback
        q.put((rank, 'error', traceback.format_exc()[-3000:]))


@pytest.mark.parametrize('overlap,gsync', [(True, 'ddp'), (True, 'flat')])       # (the single-stream DDP variant, 100 s of process start-up, went: the suite's time limit)
def test_world2_full_step_vs_oracle(overlap, gsync):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90) + (100 if overlap else 0) + (200 if gsync == 'flat' else 0)
    steps = 2 if gsync == 'flat' else 1          # (DDP: one step -- the suite's time limit; the flat sync, the default, runs two)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, steps, overlap, gsync)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    for rank, status, info in res:
        assert status == 'ok', info
        assert info['loss'] < 1e-3, info
        assert info['grad'] < 5e-3, info
    # the negative draws must have exercised the other-rank clip path at least once across ranks and steps
    if steps > 1:
        assert sum(info['remote'] for _, _, info in res) > 0, res


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="first contact with RCCL at N > 1: needs two GPUs (the build and test boxes have one)")


@needs_two_gpus
@pytest.mark.parametrize('gsync', ['flat', 'ddp'])
def test_world2_full_step_vs_oracle_on_rccl(gsync):
    """The world-size-2 step of test_world2_full_step_vs_oracle on the PRODUCT transport: one GPU per rank, backend 'nccl' (RCCL over
    xGMI), AllGather_multi for the EgoNCE embeddings, the device request gather + point-to-point token exchange for the other rank's
    hard negatives, FlatGradSync's in-place all-reduces issued from the weight-gradient stream / DDP's buckets -- against the CPU oracle
    under a gloo group of the same ranks.  Skipped on a one-GPU box: the first N > 1 execution on RCCL is this parity test, not the
    scaling bench."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 90) + (200 if gsync == 'flat' else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 2, True, gsync, 'nccl')) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    for rank, status, info in res:
        assert status == 'ok', info
        assert info['loss'] < 1e-3, info
        assert info['grad'] < 5e-3, info
    assert sum(info['remote'] for _, _, info in res) > 0, res


def _run_bench(nproc, extra, env_extra, port):
    """bench.py as the driver launches it (torch.distributed.run for nproc > 1); returns rank 0's JSON line"""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), **env_extra)
    base = ['bench.py', '--gpus', str(nproc), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-gemm-events'] + extra
    if nproc > 1:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
               '--master-port', str(port)] + base
    else:
        cmd = [sys.executable] + base
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                     # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_flat_grad_sync_rehearsal():
    """`bench.py --gpus 2` the way the driver launches it (torch.distributed.run, one process per rank), on ONE GPU under
    EGV_BENCH_REHEARSAL=1 (gloo rendezvous, both ranks on cuda:0), default gradient sync (--grad-sync flat: in-place all-reduce of the
    per-block flat gradient buffers), the full three-loss step with the other rank's hard negatives: the launch path, the
    max-over-ranks timing and the one-JSON-line contract, and -- dropout off -- the losses against 1-rank runs: the MLM loss is the
    label-count-weighted mean of the two ranks' own MLM losses (no cross-rank term but the mean), EgoNCE / ITM see 16 instead of 8
    candidates and must stay finite and in range."""
    common = ['--batch', '4', '--frames', '4', '--drop-rate', '0']
    two = _run_bench(2, common + ['--grad-sync', 'flat'], {'EGV_BENCH_REHEARSAL': '1'}, 29631)
    assert two['n_gpus'] == 2 and two['steps'] == 2 and two['scaling'] == 'weak' and two['value'] > 0
    assert two['config']['global_batch'] == 8
    singles = [_run_bench(1, common + ['--seed-offset', str(r)], {}, 29641 + r) for r in range(2)]
    cnt = two['losses']['mlm_labels_per_rank']
    assert len(cnt) == 2 and [s['losses']['mlm_labels_per_rank'][0] for s in singles] == cnt
    want = sum(s['losses']['loss_mlm'] * c for s, c in zip(singles, cnt)) / sum(cnt)
    assert abs(two['losses']['loss_mlm'] - want) <= 2e-3 * abs(want), (two['losses'], [s['losses'] for s in singles])
    for k in ('EgoNCE', 'loss_itm', 'loss_total'):
        assert np.isfinite(two['losses'][k]) and 0 < two['losses'][k] < 50, two['losses']


@needs_two_gpus
@pytest.mark.parametrize('wire', ['fp32', 'bf16'])
def test_bench_two_ranks_on_rccl(wire):
    """`bench.py --gpus 2` exactly as the driver launches it, one GPU per rank on RCCL (no rehearsal switch): the one-JSON-line contract
    and the MLM loss against the label-count-weighted mean of two 1-rank runs on the same batches, with the fp32 and the bf16 wire
    format of the flat gradient sync.  Skipped on a one-GPU box."""
    common = ['--batch', '4', '--frames', '4', '--drop-rate', '0']
    two = _run_bench(2, common + ['--grad-sync', 'flat', '--grad-wire', wire], {}, 29661 + (7 if wire == 'bf16' else 0))
    assert two['n_gpus'] == 2 and two['steps'] == 2 and two['scaling'] == 'weak' and two['value'] > 0
    assert two['config']['global_batch'] == 8
    singles = [_run_bench(1, common + ['--seed-offset', str(r)], {}, 29671 + r) for r in range(2)]
    cnt = two['losses']['mlm_labels_per_rank']
    want = sum(s['losses']['loss_mlm'] * c for s, c in zip(singles, cnt)) / sum(cnt)
    assert abs(two['losses']['loss_mlm'] - want) <= 2e-3 * abs(want), (two['losses'], [s['losses'] for s in singles])
    for k in ('EgoNCE', 'loss_itm', 'loss_total'):
        assert np.isfinite(two['losses'][k]) and 0 < two['losses'][k] < 50, two['losses']


def _nccl_worker(port, q):
    """ONE rank on the real transport (RCCL: backend 'nccl', world size 1 -- two ranks cannot share a GPU on RCCL)."""
    try:
        sys.path.insert(0, REPO)
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.pop('EGV_EXCHANGE_HOST_TABLE', None)
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        from egovlpv2_amd.trainer import exchange as X
        from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi
        import types
        out = {}
        # (a) the request table over the device all-gather vs the host all-gather, issued from an external low-priority stream
        # between other collectives of the same communicator (as in the step: gathers before, all-reduces after)
        side = torch.cuda.Stream(priority=0)
        bsz = 8
        for trial in range(4):
            ids = [(3 * trial + 5 * i) % (2 * bsz) for i in range(bsz)]          # global ids of a 2-rank job: half of them "remote"
            t = torch.randn(bsz, 16, device='cuda')
            g = AllGather_multi.apply(t, 1, types.SimpleNamespace(world_size=1, rank=0))
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                get = X.start_request_gather(ids, torch.tensor(ids, dtype=torch.int64).cuda(non_blocking=True), 0, bsz, 1)
            flat = torch.randn(1 << 20, device='cuda')
            ref = flat.clone()
            w = dist.all_reduce(flat, async_op=True)
            table = get()
            w.wait()
            torch.cuda.synchronize()
            assert table == X.gather_requests(ids, 0, bsz, 1), (table, ids)
            assert table == [sorted({j for j in ids if j >= bsz})], table
            assert torch.equal(flat, ref) and torch.equal(g, t)
        out['table'] = 'ok'
        # (b) the bf16-wire gradient sum (grad_sync.allreduce_bf16_wire) on RCCL's all-to-all / all-gather, from a side stream as
        # FlatGradSync(wire='bf16') issues it: with one rank the result is the bf16 rounding of the buffer
        from egovlpv2_amd.trainer.grad_sync import allreduce_bf16_wire
        flat = torch.randn((1 << 20) + 3, device='cuda')
        want = flat.to(torch.bfloat16).float()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            allreduce_bf16_wire(flat)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(flat, want)
        out['wire'] = 'ok'
        q.put(('ok', out))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(('error', traceback.format_exc()[-3000:]))


def test_request_gather_on_rccl_matches_host_gather():
    """trainer/exchange.py::start_request_gather on its PRODUCTION branch (device all-gather on RCCL + pinned-memory read-back behind
    an event), from a side stream, interleaved with other collectives of the communicator, against the host all-gather of the gloo
    tests (round-4 advisor finding: the branch had never run)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(29700 + os.getpid() % 90, q))
    p.start()
    status, info = q.get(timeout=600)
    p.join(timeout=60)
    if p.is_alive():
        p.kill()
    assert status == 'ok', info


def test_bench_one_rank_on_rccl_flat_grad_sync():
    """bench.py --force-ddp: the data-parallel step (process group on RCCL, FlatGradSync's in-place all-reduces issued from the
    weight-gradient stream in backward order) on the real communicator at world size 1: the launch path, the collectives' issue order
    on RCCL and the one-JSON-line contract; the losses must be finite and in range (the values themselves are pinned by the parity
    tests; DDP on RCCL is covered by the same option with --grad-sync ddp, not run here for the suite's time limit)."""
    common = ['--batch', '4', '--frames', '4', '--drop-rate', '0']
    r = _run_bench(1, common + ['--force-ddp', '--grad-sync', 'flat'], {}, 29652)
    assert r['n_gpus'] == 1 and r['value'] > 0
    for k in ('EgoNCE', 'loss_mlm', 'loss_itm', 'loss_total'):
        assert np.isfinite(r['losses'][k]) and 0 < r['losses'][k] < 50, r['losses']
