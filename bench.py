#!/usr/bin/env python
"""bench.py -- EgoVLPv2 pre-training hot path on MI355X: video-text pairs/s for FrozenInTime forward+backward.

  python bench.py --gpus 1 --steps K --warmup W                       (one rank)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = zero_grad + forward (EgoNCE + MLM + ITM, three backbone passes) + backward over one synthetic batch of B
pairs per GPU at 16 x 224^2 frames / 32 tokens (BASELINE.json configs[2]; `--workload dual` = configs[1], EgoNCE only),
bf16 storage / fp32 accumulation, weights cast from the fp32 masters inside every timed step, data-parallel gradient all-reduce (flat per-block buffers, or DDP)
over RCCL for N > 1.  Prints ONE JSON line (rank 0) with the contract fields + `roofline` (dominant kernel: the MFMA GEMM,
timed with HIP events on its own stream inside the timed region) + `cpu_baseline` (the CPU oracle timed on the host cores).
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF dense)
N_CUS = 256                    # compute units of one MI355X


def flops_per_pair(cfg, L, workload):
    """Algorithmic matmul FLOPs per video-text pair, forward (SURVEY.md §8d formulas); fwd+bwd = 3x."""
    d, F, N, S, P = cfg.dim, cfg.frames, cfg.n_patches, cfg.seq, cfg.proj_dim
    patch = 2 * F * N * d * (3 * cfg.patch ** 2)
    vblock = 2 * (2 * S * d * 3 * d + 2 * S * d * d) + 4 * d * (F * N * (1 + F) + S) + 4 * d * (F * N * (1 + N) + S) + 4 * S * d * 4 * d
    i2t = 4 * S * d * d + 2 * L * d * 2 * d + 4 * S * L * d
    tlayer = 8 * L * d * d + 4 * L * L * d + 16 * L * d * d
    t2i = 4 * L * d * d + 4 * S * d * d + 4 * L * S * d
    proj = 2 * (2 * d * P + 4 * P * P)
    dual = patch + cfg.depth * vblock + cfg.depth * tlayer + proj
    fused = patch + cfg.depth * vblock + cfg.depth * tlayer + cfg.n_fuse * (i2t + t2i)
    heads = 2 * L * d * d + 2 * L * d * cfg.vocab + 2 * L * d * d + 6 * d * d
    return dual if workload == 'dual' else dual + 2 * fused + heads


def skipped_flops_per_pair(cfg, L, workload, world, tail=True):
    """Forward FLOPs of the reference's algorithm that this build does not execute because they are dead or duplicated:
    the MLM pass's last video block (output discarded); the ITM pass's unfused video prefix for clips
    this rank already pushed through the identical prefix in the MLM pass (all of them at world size 1; on average all but
    the ~B/4 * (W-1)/W hard-negative clips owned by other ranks otherwise); round 5: the rows nobody reads of the LAST block of a
    video pass (EgoNCE tower, ITM stack: only the CLS rows leave it, so attn.proj, the patch queries of the space attention, the
    image-to-text part and the MLP run on the CLS rows alone -- model._video_block_tail) and the second patch embedding of the same
    clips (EgoNCE tower and shared prefix embed once)."""
    d, F, N, S = cfg.dim, cfg.frames, cfg.n_patches, cfg.seq
    # per-row work of a block that only the CLS row needs: attn.proj, the MLP, the patch queries of the space attention
    tail_rows = (S - 1) * (2 * d * d + 4 * d * 4 * d) + 4 * d * (F * N * (1 + N))
    tail_i2t = (S - 1) * (4 * d * d + 4 * L * d)                      # fused: qkv_i2t, proj_i2t, the image-to-text attention of the other rows
    if workload == 'dual':
        return float(tail_rows) if tail else 0.0
    patch = 2 * F * N * d * (3 * cfg.patch ** 2)
    vblock = 2 * (2 * S * d * 3 * d + 2 * S * d * d) + 4 * d * (F * N * (1 + F) + S) + 4 * d * (F * N * (1 + N) + S) + 4 * S * d * 4 * d
    i2t = 4 * S * d * d + 2 * L * d * 2 * d + 4 * S * L * d
    prefix = patch + (cfg.depth - cfg.n_fuse) * vblock
    shared = 1.0 - 0.25 * (world - 1) / world
    tails = (2 * tail_rows + tail_i2t) if tail else 0.0                # EgoNCE tower's last block + the ITM stack's last (fused) block
    return (vblock + i2t) + shared * prefix + tails + patch


def _cpu_sample(frames, L, workload, thread_choices, budget_s):
    """CPU oracle timing (runs in a subprocess): SURVEY.md section 8(d) procedure -- one warm-up step, then >= 2 timed fwd+bwd steps
    at B = 1 on the workload's own shapes.  Thread count: the best of `thread_choices` on a 4-frame probe (one warm-up + one timed
    step each).  If warm-up + 2 timed steps of the full shape do not fit what is left of `budget_s`, the 4-frame shape is the sample
    and the caller scales it by the FLOP ratio.  Returns a dict."""
    from oracle import ref_model as O
    from egovlpv2_amd.synthetic import make_state_dict, make_batch
    from egovlpv2_amd.config import PathConfig, tiny_config
    tasks = 'EgoNCE' if workload == 'dual' else 'EgoNCE_MLM_ITM'
    t_start = time.time()

    def one(c, B, Lx, sd=None):
        if sd is None:
            sd = make_state_dict(c, 0)
            for v in sd.values():
                if v.is_floating_point():
                    v.requires_grad_(True)
        for v in sd.values():
            v.grad = None
        data, noun, verb = make_batch(c, B, Lx, 99)
        t0 = time.time()
        loss, _, _ = O.forward_losses(sd, data, noun, verb, O.make_cfg(**c.as_dict()), tasks)
        loss.backward()
        return time.time() - t0, sd
    torch.set_num_threads(thread_choices[0])
    one(tiny_config(), 2, 16)                       # page the libraries in
    probe_frames = min(4, frames)
    c4 = PathConfig(frames=probe_frames)
    tried, sd4 = {}, None
    for th in thread_choices:
        torch.set_num_threads(th)
        _, sd4 = one(c4, 1, L, sd4)                 # warm-up at this thread count (first touch of the thread pool / buffers)
        tried[th], sd4 = one(c4, 1, L, sd4)
    best = min(tried, key=tried.get)
    torch.set_num_threads(best)
    del sd4
    cf = PathConfig(frames=frames)
    ratio = flops_per_pair(cf, L, workload) / flops_per_pair(c4, L, workload)
    est = tried[best] * ratio
    left = budget_s - (time.time() - t_start)
    if frames > probe_frames and 3.3 * est <= left:
        _, sdf = one(cf, 1, L)
        t1, sdf = one(cf, 1, L, sdf)
        t2, sdf = one(cf, 1, L, sdf)
        return {"frames": frames, "threads": best, "tried": tried, "steps_s": [t1, t2], "seconds": 0.5 * (t1 + t2), "ratio": 1.0}
    t1, sd4 = one(c4, 1, L)
    t2, sd4 = one(c4, 1, L, sd4)
    t3, sd4 = one(c4, 1, L, sd4)
    return {"frames": probe_frames, "threads": best, "tried": tried, "steps_s": [t2, t3], "seconds": 0.5 * (t2 + t3),
            "ratio": ratio if frames > probe_frames else 1.0}


def cpu_baseline(cfg, L, workload, timeout_s=330, budget_s=240):
    """The CPU oracle (oracle/ref_model.py, kind 'port': a restatement of the reference's fp32 CPU path, pinned to the
    reference by tests/golden) timed on this box's host cores on a BOUNDED sample of the same workload, as SURVEY.md 8(d)
    prescribes: the same model and shapes (16 x 224^2 frames, 32 tokens, all three losses) at B = 1, one warm-up step and two
    timed fwd+bwd steps, at the better of 64 / 128 torch threads (probed on the 4-frame shape); when three full-shape steps do
    not fit the budget the 4-frame shape is timed instead and scaled by the FLOP ratio.  Runs in a subprocess with a timeout."""
    import subprocess
    cores = os.cpu_count() or 1
    choices = sorted({t for t in (64, 128) if t <= cores} or {max(1, cores // 2) if cores > 16 else cores})
    code = (f"import sys, json; sys.path.insert(0, {REPO!r}); import bench; "
            f"print(json.dumps(bench._cpu_sample({cfg.frames}, {L}, {workload!r}, {choices!r}, {budget_s})))")
    try:
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=timeout_s, cwd=REPO)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                              # never let the baseline leg take the bench down
        return {"value": None, "unit": "pairs/s", "cores": choices[-1], "kind": "port", "sample": f"failed: {type(e).__name__}"}
    try:
        model_name = [ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')][0]
    except Exception:
        model_name = 'unknown CPU'
    dt, ratio = float(d['seconds']), float(d['ratio'])
    return {"value": round(1.0 / (dt * ratio), 5), "unit": "pairs/s", "cores": int(d['threads']), "kind": "port",
            "sample": f"oracle fp32 fwd+bwd, B=1, {d['frames']}x{cfg.img}^2 frames, {L} tokens, {'EgoNCE' if workload == 'dual' else 'EgoNCE+MLM+ITM'}, "
                      f"1 warm-up + 2 timed steps ({d['steps_s'][0]:.1f} s, {d['steps_s'][1]:.1f} s) at {d['threads']} torch threads of {cores} logical CPUs "
                      f"({model_name})" + (f"; scaled to {cfg.frames} frames by the FLOP ratio {ratio:.2f}" if ratio != 1.0 else ""),
            "seconds_per_step": round(dt, 2), "steps_s": [round(x, 2) for x in d['steps_s']],
            "threads_tried_s_on_4_frames": {str(k): round(float(v), 2) for k, v in d['tried'].items()},
            "flops_ratio_workload_over_sample": round(ratio, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='full', choices=['full', 'dual'])
    ap.add_argument('--arch', default='base16', choices=['base16', 'large14'],
                    help='base16: ViT-B/16 + RoBERTa-base (the measured configs); large14: the configs[4] geometry -- ViT-L/14 TimeSformer '
                         '(24 blocks, d = 1024, 16 heads, 256 patches per frame) + RoBERTa-large width, 12 fused layers -- in bf16 (no fp8 weights)')
    ap.add_argument('--fp8', action='store_true', help='configs[4] "fp8 MFMA weight path": MX-fp8 forward / dgrad GEMMs in the video blocks (FrozenInTime(video_fp8=True))')
    ap.add_argument('--optimizer', action='store_true', help='also time the fused AdamW step + LR schedule (SURVEY.md §8f item 1)')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--text-len', type=int, default=32)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--drop-rate', type=float, default=0.1, help='RoBERTa dropout in the train step (pretrained roberta-base config: 0.1)')
    ap.add_argument('--seed-offset', type=int, default=0, help='added to the per-rank data / RNG seeds (1234 + rank, 1 + rank): a 1-rank run with offset r sees the batch of rank r of a multi-rank run (test aid)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gemm-events', action='store_true')
    ap.add_argument('--force-ddp', action='store_true', help='run the data-parallel path (process group, gradient sync) even at world size 1 (test aid)')
    ap.add_argument('--grad-sync', default='flat', choices=['flat', 'ddp'],
                    help='world > 1: flat = all-reduce of the per-block flat gradient buffers as they complete (trainer/grad_sync.py); '
                         'ddp = torch DistributedDataParallel as in the reference')
    ap.add_argument('--grad-wire', default=None, choices=['fp32', 'bf16'],
                    help='--grad-sync flat: fp32 = in-place all-reduce (default), bf16 = bf16 on the links with fp32 accumulation on arrival '
                         '(grad_sync.allreduce_bf16_wire; half the bytes, two bf16 roundings per gradient element)')
    a = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    # EGV_BENCH_REHEARSAL=1 (test aid, never a measurement): all ranks share GPU 0 and rendezvous over gloo, so that the
    # world_size > 1 code path (DDP, gathers, other-rank negatives, max-over-ranks timing) can be run on a 1-GPU box.
    rehearsal = bool(os.environ.get('EGV_BENCH_REHEARSAL'))
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dist = world > 1 or a.force_ddp
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        if rehearsal:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from egovlpv2_amd import hipops as ops
    from egovlpv2_amd.config import PathConfig
    from egovlpv2_amd.synthetic import make_state_dict, make_batch
    from egovlpv2_amd.model.model import FrozenInTime
    from egovlpv2_amd.model.loss import EgoNCE
    from egovlpv2_amd.trainer.trainer_egoclip import AllGather_multi

    cfg = PathConfig(frames=a.frames, drop_rate=a.drop_rate)     # roberta-base hidden / attention dropout (train mode)
    if a.arch == 'large14':
        cfg = PathConfig(depth=24, n_fuse=12, patch=14, dim=1024, heads=16, frames=a.frames, drop_rate=a.drop_rate)
    tasks = 'EgoNCE' if a.workload == 'dual' else 'EgoNCE_MLM_ITM'
    dtype = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    model = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True},
                         {'model': 'roberta-base', 'pretrained': True, 'input': 'text'}, path_config=cfg,
                         task_names='EgoNCE_MLM_ITM', compute_dtype=dtype, video_fp8=a.fp8)
    model.load_state_dict(make_state_dict(cfg, 0), strict=True)          # same random-init weights on every rank
    model = model.to(dev)
    net = model
    gsync = None
    if use_dist and a.grad_sync == 'ddp':
        from torch.nn.parallel import DistributedDataParallel as DDP
        net = DDP(model, device_ids=[local], static_graph=True, gradient_as_bucket_view=True,
                  find_unused_parameters=False, bucket_cap_mb=64)
    elif use_dist:
        from egovlpv2_amd.trainer.grad_sync import FlatGradSync
        gsync = FlatGradSync(model, wire=a.grad_wire)               # all-reduce of the flat per-block gradient buffers as they complete
    data, noun, verb = make_batch(cfg, a.batch, a.text_len, 1234 + rank + a.seed_offset)
    n_mlm_labels = int(((data['text_mlm_labels'] >= 0) & (data['text_mlm_labels'] < cfg.vocab)).sum())
    data = {'video': data['video'].to(dev), 'text': {k: v.to(dev) for k, v in data['text'].items()},
            'text_mlm_ids': data['text_mlm_ids'].to(dev), 'text_mlm_labels': data['text_mlm_labels'].to(dev)}
    noun, verb = noun.to(dev), verb.to(dev)
    args = types.SimpleNamespace(world_size=world, rank=rank)
    loss_fn = EgoNCE()
    conf = {'loss': {'type': 'EgoNCE'}}
    np.random.seed(1 + rank + a.seed_offset)
    torch.manual_seed(1 + rank + a.seed_offset)

    optimizer = scheduler = None
    if a.optimizer:
        from egovlpv2_amd.set_optim_schedule import set_schedule
        ocfg = {"optimizer": {"type": "AdamW", "args": {"lr": 3e-5, "weight_decay": 0.01, "lr_mult_head": 4, "lr_mult_cross_modal": 4}}}
        optimizer, scheduler = set_schedule(model, ocfg, {"decay_power": "cosine", "end_lr": 1e-7}, 100000, 10000)

    def step():
        ops.invalidate_weight_cache()            # weights are re-cast from the fp32 masters every step, as in training
        for p in model.parameters():
            p.grad = None
        loss, ld, _ = net(data, noun, verb, AllGather_multi.apply, world, args, conf, loss_fn, local, task_names=tasks)
        if gsync is not None:
            gsync.backward(loss)
        else:
            loss.backward()
        if optimizer is not None:
            optimizer.step()
            scheduler.step()
        return ld

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        ld = step()                                 # a failure of the default gradient-sync path fails the benchmark on every rank
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]     # one event per step on the calling stream (no sync): the spread
    marks[0].record()                                                              # of the K step times, reported beside their mean
    t0 = time.perf_counter()
    for i in range(a.steps):
        ld = step()
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    # Per-kernel durations for the roofline object: the same K steps twice more with one HIP event pair around every GEMM,
    # attention and LayerNorm launch, on the stream the kernel is launched on -- once exactly as timed above (text-side and
    # weight-gradient kernels share the CUs with the kernel being timed: "in_step", what rocprofv3 sees in the step) and once
    # single-stream (EGV_NO_OVERLAP=1: "isolated", the kernel alone on the chip).  Neither pass is part of `value`.
    use_events = not a.no_gemm_events
    passes = {}
    if use_events:
        def event_pass(single_stream):
            prev = os.environ.get('EGV_NO_OVERLAP')
            if single_stream:
                os.environ['EGV_NO_OVERLAP'] = '1'
            try:
                step()
                sync()
                ops.prof_reset()
                ops.prof_enable(True)
                for _ in range(a.steps):
                    step()
                sync()
                ops.prof_enable(False)
                return ops.prof_collect()
            finally:
                if single_stream:
                    if prev is None:
                        os.environ.pop('EGV_NO_OVERLAP', None)
                    else:
                        os.environ['EGV_NO_OVERLAP'] = prev
        passes['in_step'] = event_pass(False)
        passes['isolated'] = event_pass(True)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    losses = {k: round(float(v.detach()), 5) for k, v in ld.items()}
    mlm_counts = torch.tensor([n_mlm_labels], dtype=torch.int64, device=dev)
    if use_dist:
        gathered = [torch.zeros_like(mlm_counts) for _ in range(world)]
        dist.all_gather(gathered, mlm_counts)
        mlm_counts = torch.cat(gathered)
    losses['mlm_labels_per_rank'] = [int(x) for x in mlm_counts.tolist()]     # weights of the per-rank MLM means in the global mean

    roof = None
    if use_events and rank == 0:
        kinds = {-1: 'launch declined by a one-pass attention entry point (no kernel ran: the caller took another path)', 0: 'gemm_kernel<bf16,NT> (generic 128x128)', 1: 'gemm_kernel<bf16,NN> (generic)', 2: 'gemm_kernel<bf16,TN> (generic)', 7: 'gemm_skinny_kernel (bf16 NT, <= 16 rows: projection heads)', 9: 'wgrad_small_m_kernel (outer product, <= 16 rows)',
                 4: 'gemm_kernel<f32,NT>', 5: 'gemm_kernel<f32,NN>', 6: 'gemm_kernel<f32,TN>',
                 8: 'gemm_ring_kernel<256x128> (NT fwd+dgrad, DMA ring)', 10: 'gemm_wgrad_ring_kernel (TN wgrad, 256x128 DMA ring)',
                 12: 'gemm_pp_kernel (NT fwd+dgrad, persistent ping-pong 256x256)', 13: 'gemm_ring_kernel<128x128> (NT, text-side grids)',
                 14: 'gemm_wgrad_pp_kernel (TN wgrad, ping-pong 256x256, one launch per gradient)',
                 15: 'gemm_wgrad_group_kernel (TN wgrad, persistent grouped launch: all weight gradients of a SpaceTimeBlock)'}
        pmc_names = {8: 'gemm_ring_kernel<256x128>', 10: 'gemm_wgrad_ring_kernel', 12: 'gemm_pp_kernel', 13: 'gemm_ring_kernel<128x128>',
                     14: 'gemm_wgrad_pp_kernel', 15: 'gemm_wgrad_group_kernel'}

        def aggregate(recs):
            agg = {}
            for fl, ms, kd, by, cu in recs:
                e = agg.setdefault(kd, [0.0, 0.0, 0, 0.0, 0.0])
                e[0] += fl
                e[1] += ms
                e[2] += 1
                e[3] += by
                e[4] += ms * (cu if cu > 0 else N_CUS)      # CU-milliseconds the launches were planned for
            return agg
        agg_in, agg_iso = aggregate(passes['in_step']), aggregate(passes['isolated'])
        if os.environ.get('EGV_BENCH_SHAPES'):           # per (kernel kind, FLOPs) breakdown on stderr: which shapes run slow in-step
            for tag, recs in passes.items():
                by = {}
                for fl, ms, kd, _b, _c in recs:
                    e = by.setdefault((kd, fl), [0.0, 0])
                    e[0] += ms
                    e[1] += 1
                for (kd, fl), (ms, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:40]:
                    print(f"shape[{tag}] kind={kd} gflop={fl / 1e9:9.2f} n/step={n / a.steps:6.1f} ms/step={ms / a.steps:7.3f} "
                          f"avg_us={ms / n * 1e3:8.1f} TF={fl * n / (ms * 1e-3) / 1e12 if ms else 0:7.1f}", file=sys.stderr)
        gemm_kinds = [k for k in agg_in if k < 20]
        dom = max(gemm_kinds, key=lambda k: agg_in[k][1])
        fl, ms, n, alg_bytes, cu_ms = agg_in[dom]
        ach = fl / (ms * 1e-3) / 1e12
        iso = agg_iso.get(dom)
        ach_iso = iso[0] / (iso[1] * 1e-3) / 1e12 if iso and iso[1] else None
        # HBM traffic and MFMA-busy counters of the dominant kernel are NOT measured in this run: they come from rocprofv3 --pmc
        # passes over this script (in-step, separate passes per counter group, gfx950 FETCH_SIZE correction) that were summarised by
        # tools/pmc_instep.py into a committed file; `source` says which
        traffic = mfma_busy = launches_per_step = trace_avg_us = None
        pmc_file = None
        for cand in ('round6_pmc_instep.json', 'round5_pmc_instep.json', 'round4_pmc_instep.json', 'round3_pmc_instep.json', 'round2_pmc_instep.json'):
            if os.path.exists(os.path.join(REPO, 'profiles', cand)):
                pmc_file = cand
                break
        try:
            pm = json.load(open(os.path.join(REPO, 'profiles', pmc_file)))
            ent = pm['kernels'].get(pmc_names.get(dom, ''))
            if ent:
                traffic = {k: ent[k] for k in ('fetch_bytes_per_launch', 'write_bytes_per_launch', 'traffic_bytes_per_launch', 'method') if k in ent}
                traffic['algorithmic_bytes_per_launch'] = round(alg_bytes / n)
                traffic['source'] = f'profiles/{pmc_file} (committed rocprofv3 --pmc passes over this script; not re-measured in this run)'
                mfma_busy = ent.get('mfma_busy_frac')
                trace_avg_us = ent.get('avg_us')
            launches_per_step = pm.get('launches_per_step')
        except Exception:
            pass

        def tf_table(agg):
            return {kinds.get(k, str(k)): {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1) if v[1] else 0.0, "ms_per_step": round(v[1] / a.steps, 2),
                                           "launches_per_step": v[2] // a.steps} for k, v in agg.items() if k < 20}

        def hbm_class(agg, ks):
            by = sum(agg[k][3] for k in ks if k in agg)
            ms_ = sum(agg[k][1] for k in ks if k in agg)
            nl = sum(agg[k][2] for k in ks if k in agg)
            if not ms_:
                return None
            return {"GB/s": round(by / (ms_ * 1e-3) / 1e9, 1), "frac_of_8TB/s": round(by / (ms_ * 1e-3) / 8e12, 4), "ms_per_step": round(ms_ / a.steps, 2),
                    "launches_per_step": nl // a.steps}
        hbm = {"attention (fwd / dQ / dK,dV / one-pass bwd launches, all towers)": {"in_step": hbm_class(agg_in, (20, 21, 22, 23)), "isolated": hbm_class(agg_iso, (20, 21, 22, 23))},
               "layernorm (fwd + bwd launches)": {"in_step": hbm_class(agg_in, (30, 31)), "isolated": hbm_class(agg_iso, (30, 31))},
               "bytes": "algorithmic: every operand and result of a launch once"}
        roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4), "frac_in_step": round(ach / PEAK_BF16_TFLOPS, 4),
                "frac_isolated": round(ach_iso / PEAK_BF16_TFLOPS, 4) if ach_iso else None,
                # the backward's persistent grids are planned for the CUs the grouped weight-gradient launch leaves them (DESIGN.md
                # 3.2): FLOPs over the MFMA peak of the CUs each launch was planned for, time-weighted
                "frac_of_granted_cus_in_step": round(fl / (cu_ms * 1e-3 / N_CUS) / 1e12 / PEAK_BF16_TFLOPS, 4) if cu_ms else None,
                "mean_cus_granted_in_step": round(cu_ms / ms, 1) if ms else None,
                "measured": "HIP events around every launch of the kernel, on its stream, in a repeat of the timed steps: in_step = as timed (two-stream); "
                            "isolated = single-stream (EGV_NO_OVERLAP=1); frac = in_step",
                "traffic": traffic, "mfma_busy_frac_rocprof": mfma_busy,
                "kernel": kinds.get(dom, str(dom)), "launches": n, "avg_launch_ms": round(ms / n, 4),
                # the same kernel's average duration in the committed rocprofv3 kernel trace of this command (two-stream step).  The HIP
                # events of this run bracket dispatch latency as well (a few microseconds per launch), so avg_launch_ms reads higher
                "avg_launch_ms_rocprof": round(trace_avg_us / 1e3, 4) if trace_avg_us else None,
                "frac_in_step_rocprof": round(fl / n / (trace_avg_us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4) if trace_avg_us else None,
                "all_gemm": tf_table(agg_in), "all_gemm_isolated": tf_table(agg_iso),
                "gemm_ms_per_step": round(sum(agg_in[k][1] for k in gemm_kinds) / a.steps, 2),
                "hbm_bound_classes": hbm,
                "launches_per_step_rocprof": launches_per_step,
                "launches_per_step_source": f'profiles/{pmc_file}' if launches_per_step is not None else None}

    if rank == 0:
        pairs = world * a.batch * a.steps
        value = pairs / dt
        fpp = 3.0 * flops_per_pair(cfg, a.text_len, a.workload)
        from egovlpv2_amd import switches as SW
        tail_on = a.dtype == 'bf16' and not a.fp8 and SW.on('EGV_CLS_TAIL') and SW.on('EGV_VIDEO_RES32')
        fpx = fpp - 3.0 * skipped_flops_per_pair(cfg, a.text_len, a.workload, world, tail_on)
        out = {"metric": "video-text pairs/sec/node (EgoClip fwd+bwd, 16x224^2, 32 tok)", "value": round(value, 3),
               "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "step_ms_min_median_max": [round(step_ms[0], 2), round(step_ms[len(step_ms) // 2], 2), round(step_ms[-1], 2)],
               "dtype": (a.dtype + "+mxfp8(video fwd/dgrad GEMMs)") if a.fp8 else a.dtype, "data": "synthetic",
               "config": {"workload": (("configs[2] full fusion EgoNCE+MLM+ITM" if a.arch == 'base16' else "full fusion EgoNCE+MLM+ITM") if a.workload == 'full' else "configs[1] dual encoder EgoNCE")
                          + (f", ViT-B/16 TimeSformer + RoBERTa-base" if a.arch == 'base16' else ", configs[4] geometry: ViT-L/14 TimeSformer + RoBERTa-large width, bf16 weights")
                          + f", B={a.batch}/GPU, {a.frames}x224^2, {a.text_len} tok",
                          "global_batch": world * a.batch, "parallelism": f"dp{world}", "drop_rate": a.drop_rate, "timed": "zero_grad + fwd + bwd (+ gradient all-reduce: " + ((a.grad_sync + ("/bf16 wire" if getattr(gsync, "wire", "") == "bf16" else "")) if use_dist else "none") + "), weight cast included" + (" + fused AdamW step" if a.optimizer else "")},
               # model_tflops: the reference algorithm's matmul FLOPs per pair x pairs/s; executed_tflops leaves
               # out what skipped_flops_per_pair lists: the dead MLM video block, the ITM video prefix shared with the MLM pass, the
               # unread rows of the last block of the EgoNCE tower and of the ITM stack, the second patch embedding
               "model_tflops": round(value * fpp / 1e12, 1), "executed_tflops": round(value * fpx / 1e12, 1),
               "mfma_frac_of_peak": round(value * fpx / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
               "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
               "kernel_launches_per_step": (roof or {}).pop('launches_per_step_rocprof', None),
               "kernel_launches_per_step_source": (roof or {}).pop('launches_per_step_source', None),
               "losses": losses, "roofline": roof}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, a.text_len, a.workload)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
