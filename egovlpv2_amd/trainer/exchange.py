"""Hard-negative clip exchange of the ITM branch across data-parallel ranks.

The reference all-gathers the raw pixels of every rank (model/model.py:429-431: 77 MB per rank at configs[2], 616 MB per
step at 8 GPUs) and pushes every sampled negative through the video prefix again.  Here the ranks exchange

  1. the sampled clip indices (a B-entry int64 vector per rank: one device all-gather on RCCL, read back through pinned memory
     behind an event; on gloo -- the CPU / one-GPU tests -- a host all-gather), and
  2. only the prefix TOKENS of the clips that were actually drawn from another rank (bf16 (S, d) per clip, at most
     ceil(B/2) clips per rank and step), point to point from their owner,

and the gradient of those tokens travels back to the owner in backward, where it joins the gradient of the owner's own use
of the same prefix.  Values are identical to the reference's (the prefix is a function of the pixels and of replicated
parameters only, and the kernels are batch-composition independent); parameter gradients are identical after DDP's
averaging because the owner's contribution replaces the requester's (the sum over ranks is unchanged).  The ITM draw itself
keeps the reference's order of draws (model/model.py:459-468; CPU generators, see FrozenInTime.forward) -- it happens before this
exchange, per rank, on the host.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

_meta_group = [None]


def meta_group():
    """gloo side group for host-resident metadata (created on first use, by every rank at the same point of the step)"""
    if _meta_group[0] is None:
        _meta_group[0] = dist.new_group(backend='gloo') if dist.get_backend() != 'gloo' else dist.group.WORLD
    return _meta_group[0]


def _table(rows, bsz):
    """rows[r] = the clip ids rank r sampled -> table[requester] = sorted remote clip ids it needs (ids are global: owner = id // bsz)"""
    table = []
    for r, ids in enumerate(rows):
        lo = r * bsz
        table.append(sorted({int(j) for j in ids if not lo <= j < lo + bsz}))
    return table


def gather_requests(vid_list, rank, bsz, world):
    """every rank's list of sampled clip ids -> the request table.  One small HOST all-gather (gloo): blocks the calling host thread
    until every rank's host has arrived -- the transport of the gloo tests; on RCCL use start_request_gather."""
    mine = torch.tensor(list(vid_list), dtype=torch.int64)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=meta_group())
    return _table([o.tolist() for o in out], bsz)


def start_request_gather(vid_list, ids_dev, rank, bsz, world):
    """Start the exchange of the sampled clip ids and return a function that yields the request table when it is needed.

    RCCL (backend 'nccl'): the B ids of this rank (`ids_dev`, already on the device) go through ONE device all-gather on the
    current stream, the result is copied to pinned host memory behind it, and the returned function waits for that copy's event --
    no host-side collective, no extra process group, and the wait is over by the time the ITM pass is assembled (the draw happens
    before the MLM pass is enqueued, FrozenInTime.forward).  The collective is issued at the same point of the step by every rank
    (the draw), like every other collective of the step.
    gloo (device tensors cannot travel): the host all-gather of gather_requests, done here.  EGV_EXCHANGE_HOST_TABLE=1 forces it."""
    from .. import switches as SW
    if dist.get_backend() == 'gloo' or SW.on('EGV_EXCHANGE_HOST_TABLE') or ids_dev is None or not ids_dev.is_cuda:
        table = gather_requests(vid_list, rank, bsz, world)
        return lambda: table
    buf = torch.empty(world * bsz, dtype=torch.int64, device=ids_dev.device)
    dist.all_gather_into_tensor(buf, ids_dev.contiguous())
    host = torch.empty(world * bsz, dtype=torch.int64, pin_memory=True)
    host.copy_(buf, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()

    def get():
        ev.synchronize()
        return _table(host.view(world, bsz).tolist(), bsz)
    return get


def _p2p(ops_):
    if ops_:
        for w in dist.batch_isend_irecv(ops_):
            w.wait()


class ExchangeClipsFn(torch.autograd.Function):
    """tokens of the clips this rank requested from other ranks, in the order of table[rank]; every rank calls it (a rank that
    requests nothing still serves).  x: (bsz * rows, d) prefix tokens of this rank's own clips."""

    @staticmethod
    def forward(ctx, x, table, rank, bsz, rows):
        world = len(table)
        host = dist.get_backend() == 'gloo' and x.is_cuda            # gloo moves host memory only (2-ranks-on-1-GPU tests)
        d = x.shape[1]
        mine = table[rank]
        recv = torch.empty(len(mine) * rows, d, dtype=x.dtype, device='cpu' if host else x.device)
        ops_, keep = [], []
        send_plan = {}
        for q in range(world):                                        # what do I owe rank q?
            if q == rank:
                continue
            ids = [j - rank * bsz for j in table[q] if j // bsz == rank]
            send_plan[q] = ids
            if ids:
                # whole clips leave as (rows, d) blocks: one gather over the clip axis (ids: a handful of ints), no per-row index list
                sel = torch.tensor(ids, dtype=torch.int64, device=x.device)
                buf = x.detach().reshape(bsz, rows, d).index_select(0, sel).view(len(ids) * rows, d)
                buf = buf.cpu() if host else buf.contiguous()
                keep.append(buf)
                ops_.append(dist.P2POp(dist.isend, buf, q))
        pos = 0
        recv_plan = {}
        for o in range(world):                                        # what do I get from owner o?
            n = sum(1 for j in mine if j // bsz == o)
            if o != rank and n:
                recv_plan[o] = (pos, n)
                ops_.append(dist.P2POp(dist.irecv, recv[pos * rows:(pos + n) * rows], o))
            pos += n
        _p2p(ops_)
        ctx.plan = (send_plan, recv_plan, rank, bsz, rows, host)
        ctx.xshape, ctx.xdtype, ctx.xdev = x.shape, x.dtype, x.device
        return recv.to(x.device) if host else recv

    @staticmethod
    def backward(ctx, g):
        send_plan, recv_plan, rank, bsz, rows, host = ctx.plan
        d = ctx.xshape[1]
        g = g.contiguous()
        gh = g.cpu() if host else g
        ops_, bufs = [], []
        for o, (pos, n) in recv_plan.items():                         # gradients of what I received go back to the owners
            ops_.append(dist.P2POp(dist.isend, gh[pos * rows:(pos + n) * rows], o))
        for q, ids in send_plan.items():
            if ids:
                b = torch.empty(len(ids) * rows, d, dtype=ctx.xdtype, device='cpu' if host else ctx.xdev)
                bufs.append((ids, b))
                ops_.append(dist.P2POp(dist.irecv, b, q))
        _p2p(ops_)
        gx = torch.zeros(ctx.xshape, dtype=ctx.xdtype, device=ctx.xdev)
        gv = gx.view(bsz, rows, d)
        for ids, b in bufs:                                           # fixed order (rank order; the ids of one requester are distinct): deterministic
            sel = torch.tensor(ids, dtype=torch.int64, device=ctx.xdev)
            gv[sel] += b.to(ctx.xdev).view(len(ids), rows, d)         # one gather-add-scatter per requester, whole clips
        return gx, None, None, None, None


def bytes_per_step(bsz, rows, d, world, itemsize=2):
    """upper bound of what one rank receives per step: ceil(bsz/2) negatives, all remote, one (rows, d) token block each"""
    return ((bsz + 1) // 2) * rows * d * itemsize
