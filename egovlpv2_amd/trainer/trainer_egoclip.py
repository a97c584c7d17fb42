"""AllGather_multi (reference trainer/trainer_egoclip.py:25-41): all_gather with an autograd backward that
keeps only the local slice of the incoming gradient (no reduction).  ``backend='nccl'`` is RCCL on ROCm."""
import torch
import torch.distributed as dist


class AllGather_multi(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tensor, n_gpu, args):
        tensor = tensor.contiguous()
        ctx.rank = args.rank
        ctx.batch_size = tensor.shape[0]
        if args.world_size == 1:
            return tensor.clone()
        out = torch.empty((args.world_size * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
        if tensor.is_cuda and dist.get_backend() == 'gloo':
            # gloo has no device all_gather_into_tensor: stage through the host (rehearsal runs of the multi-rank path on a
            # single GPU; RCCL, the production backend, takes the branch below)
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, tensor.cpu())
            out.copy_(host)
            return out
        dist.all_gather_into_tensor(out, tensor)      # one flat collective instead of world_size buffers + cat
        return out

    @staticmethod
    def backward(ctx, grad_output):
        b, r = ctx.batch_size, ctx.rank
        return grad_output[b * r: b * (r + 1)], None, None
