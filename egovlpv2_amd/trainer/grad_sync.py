"""Data-parallel gradient averaging on the flat per-block gradient buffers (the reference wraps the model in
DistributedDataParallel, base/base_trainer.py:267-269).

DDP sees 557 parameters: its reducer copies / scales every gradient into a bucket with one small kernel per tensor (560 launches
and 7 ms per step on one MI355X before any byte moves) and all-reduces 64 MB buckets.  The block executor already delivers the
gradients of a SpaceTimeBlock / RobertaLayer as ONE flat fp32 buffer, complete (summed over the EgoNCE / MLM / ITM uses of the
block) at a known point of backward.  FlatGradSync all-reduces that buffer in place, asynchronously, the moment it is complete --
about 50 RCCL calls of 20-57 MB per step, overlapped with the rest of backward, in the same order on every rank (every rank runs
the same graph) -- and the three dozen tensors outside the blocks (embeddings, heads, projections) after backward.  The average
comes from scaling the loss by 1 / world before backward (exact in floating point for power-of-two worlds), so no kernel touches
the gradients besides the collective itself.

    sync = FlatGradSync(model)                 # instead of DistributedDataParallel(model, ...)
    loss, loss_dict, ret = model(...)
    sync.backward(loss)                        # = (loss / world).backward() + the collectives
    optimizer.step()

Parameters must enter backward with p.grad None (optimizer.zero_grad(set_to_none=True), the torch default): the gradient views
of a flat buffer become p.grad by reference while the collective may still be running on them.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .. import hipops as ops
from .. import switches as SW


_wire_bufs = {}          # (device, world * per) -> (send, recv): staging of allreduce_bf16_wire, reused call after call


def _wire_staging(device, n):
    """Two bf16 staging buffers of n elements.  One pair per size and device for the life of the process: the ~50 calls of a step run
    one after another on ONE stream (FlatGradSync's side stream, or the calling stream on CPU / in tests), so reuse is ordered by that
    stream -- no allocator traffic and no record_stream bookkeeping on a path that runs beside the backward pass."""
    key = (str(device), int(n))
    b = _wire_bufs.get(key)
    if b is None:
        b = _wire_bufs[key] = (torch.zeros(n, dtype=torch.bfloat16, device=device), torch.empty(n, dtype=torch.bfloat16, device=device))
    return b


def allreduce_bf16_wire(flat, group=None):
    """Sum over the ranks of an fp32 buffer, in place, MOVED as bf16 and ACCUMULATED in fp32 on arrival (half the bytes of the fp32
    all-reduce on the per-link-bound xGMI ring): every rank sends shard q of its buffer, rounded to bf16, to rank q (all-to-all); rank q
    adds the `world` copies of its shard in fp32, in rank order, rounds the sum once and all-gathers it.  Every rank ends with the
    same bits (the shard's owner forms the sum; nobody else does), so replicas do not drift apart.  Error: two bf16 roundings per
    element (2^-9 relative each) instead of none -- an OPTION (FlatGradSync(wire='bf16')); fp32 stays the default until a node has
    measured both.  Callers on a GPU must issue every call on the same stream (the staging buffers are reused)."""
    world = dist.get_world_size(group)
    n = flat.numel()
    per = -(-n // world)
    send, recv = _wire_staging(flat.device, world * per)
    send[:n].copy_(flat)
    if world * per > n:
        send[n:].zero_()
    dist.all_to_all_single(recv, send, group=group)
    part = recv.view(world, per).float().sum(0).to(torch.bfloat16)
    dist.all_gather_into_tensor(send, part, group=group)
    flat.copy_(send[:n])
    return flat


class FlatGradSync:
    def __init__(self, model, group=None, wire=None):
        """wire: 'fp32' (default; in-place all-reduce of the flat buffers) or 'bf16' (allreduce_bf16_wire: bf16 on the links, fp32
        accumulation on arrival); None: the switch EGV_SYNC_WIRE."""
        self.model = model
        self.group = group
        self.wire = wire or SW.value('EGV_SYNC_WIRE')
        if self.wire not in ('fp32', 'bf16'):
            raise ValueError(f"FlatGradSync: wire must be 'fp32' or 'bf16', not {self.wire!r}")
        self._side = None
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # EGV_SYNC_FORCE=1 (test aid): issue the collectives even in a one-rank group, to run the RCCL code path on a 1-GPU box
        self.comm = self.world > 1 or (SW.on('EGV_SYNC_FORCE') and dist.is_available() and dist.is_initialized())
        self._works = []
        self._packed = set()
        self._flats = []

    def _on_pack(self, flat, params):
        # called with the stream current on which `flat` is complete (hipops.set_pack_hook): the collective is ordered after it
        self._packed.update(id(p) for p in params)
        self._flats.append((flat, params))
        if self.comm:
            self._reduce(flat)

    # bf16 wire: buffers below this size (the tensors outside the blocks: embeddings' position / type tables, LayerNorms, head biases, gates)
    # keep the exact fp32 all-reduce -- their bytes do not matter on the links and two bf16 roundings do matter on a LayerNorm gain
    WIRE_MIN_NUMEL = 1 << 20

    def _reduce(self, t):
        if self.wire == 'fp32' or t.numel() < self.WIRE_MIN_NUMEL:
            self._works.append(dist.all_reduce(t, group=self.group, async_op=True))
        elif t.is_cuda:
            # cast / exchange / sum / gather on a side stream ordered after the stream the buffer is complete on; the calling stream
            # joins it once, at the end of backward
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                allreduce_bf16_wire(t.view(-1), self.group)
            t.record_stream(self._side)
        else:
            allreduce_bf16_wire(t.view(-1), self.group)

    def backward(self, loss):
        """backward of loss / world with the gradient all-reduces issued as the block buffers complete; returns after every
        collective has been ordered before the calling stream (RCCL) or finished (gloo)"""
        for p in self.model.parameters():
            if p.grad is not None:
                raise RuntimeError("FlatGradSync.backward: gradients must be None on entry (zero_grad(set_to_none=True))")
        ops.set_pack_hook(self._on_pack)
        try:
            (loss * (1.0 / self.world)).backward()
            if self.comm:
                rest = [p.grad for p in self.model.parameters() if p.grad is not None and id(p) not in self._packed]
                for g in rest:
                    if not g.is_contiguous():
                        raise RuntimeError("FlatGradSync: a parameter gradient outside the block buffers is not contiguous")
                    self._reduce(g)
                for w in self._works:
                    w.wait()
                if self._side is not None:
                    torch.cuda.current_stream().wait_stream(self._side)
            # The collectives ran IN PLACE on the flat buffers: that reaches p.grad only if autograd kept the views it was handed
            # (AccumulateGrad steals a gradient by reference when .grad is None, nothing else holds it and no hook is
            # registered).  A copy made instead would hold the un-reduced local values: check, do not assume.
            for flat, params in self._flats:
                lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
                for p in params:
                    g = p.grad
                    if g is not None and not (lo <= g.data_ptr() < hi):
                        raise RuntimeError("FlatGradSync: autograd copied a block gradient instead of keeping the view of the flat "
                                           "buffer (a hook or a live reference on a parameter gradient?): the copy is not reduced")
        finally:
            ops.set_pack_hook(None)
            self._works.clear()
            self._packed.clear()
            self._flats.clear()
