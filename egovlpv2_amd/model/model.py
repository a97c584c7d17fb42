"""FrozenInTime on MI355X: the reference model API (model/model.py:46-595) over hand-written HIP kernels.

Drop-in surface: ``FrozenInTime(video_params, text_params, projection_dim, load_checkpoint,
projection, load_temporal_fix, config, task_names, norm_layer, embed_dim)``, ``forward(data, n_embeds, v_embeds,
allgather, n_gpu, args, config, loss_egonce, gpu, return_embeds, task_names)``, ``infer``, ``compute_text``,
``compute_video``, module-level ``sim_matrix`` / ``sim_matrix_batch_val`` and identical state-dict names.

Not a port: the module tree only holds the parameters (under the reference's names); the computation is a
sequence of fused HIP ops (egovlpv2_amd/hipops.py -> libegovlp_hip.so):
  * one token matrix (B*S, d) per modality, never permuted: divided space/time attention, the CLS splice and
    both cross-attentions index it in place (csrc/egv_attn.hip);
  * every Linear is an MFMA GEMM with bias / GELU / gate / residual epilogues (csrc/egv_gemm.hip);
  * MLM logits are never all-gathered: per-rank CE partial sums are exchanged instead (same loss, same grads).
Extra keyword arguments (``compute_dtype``, ``path_config``) are extensions; defaults reproduce the reference
architecture (ViT-B/16 TimeSformer + RoBERTa-base, 6 fused layers).
"""
from __future__ import annotations

import copy
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.distributed as dist

from .. import hipops as ops
from .. import switches as SW
from ..config import PathConfig
from ..synthetic import param_shapes
from ..utils.util import state_dict_data_parallel_fix

# EgoNCE_MLM_ITM_Config.yml of the reference (read there from cwd at import time; here a plain default)
DEFAULT_YML = dict(input_image_embed_size=768, vocab_size=50265, mlm_prob=0.15, input_text_embed_size=768,
                   hidden_size=768, num_heads=12, num_layers=12, mlp_ratio=4, drop_rate=0.1, num_fuse_block=6,
                   use_checkpoint=True, decay_power='cosine', end_lr=1e-7, warmup_steps=0.1)
config = dict(DEFAULT_YML)
ROBERTA_BASE_DROPOUT = 0.1     # hidden_dropout_prob = attention_probs_dropout_prob of the pretrained roberta-base config (model.py:68)

F32_MIN = torch.finfo(torch.float32).min


class _Node(nn.Module):
    """Bare container: the module tree exists only to give parameters the reference's dotted names."""


def _register(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool = False):
    parts = dotted.split('.')
    mod = root
    for p in parts[:-1]:
        nxt = mod._modules.get(p)
        if nxt is None:
            nxt = _Node()
            mod.add_module(p, nxt)
        mod = nxt
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


def _init_tensor(name: str, shape, gen: torch.Generator, cfg: PathConfig) -> torch.Tensor:
    """Reference initialisation rules: RoBERTa normal(0,0.02) with zero biases / unit LayerNorm (HF _init_weights,
    roberta.py:737); TimeSformer Linear defaults, time attention qkv = 0 and proj.weight = 1
    (video_transformer.py:96-102), alpha gates 0 (:114, roberta.py:440), model cls_token / temporal_embed 0
    (model.py:150, video_transformer.py:293), pos_embed / video cls_token trunc_normal(0.02) (:319-320), heads
    normal(0, 0.02) (model.py:35-43)."""
    last = name.rsplit('.', 1)[-1]
    if 'alpha_' in name or name in ('cls_token', 'video_model.temporal_embed'):
        return torch.zeros(shape)
    if '.timeattn.' in name:
        if name.endswith('proj.weight'):
            return torch.ones(shape)
        return torch.zeros(shape)
    if last == 'bias' or name == 'mlm_score.bias':
        return torch.zeros(shape)
    if len(shape) == 1:                        # LayerNorm weights
        return torch.ones(shape)
    if name in ('video_model.pos_embed', 'video_model.cls_token'):
        return torch.nn.init.trunc_normal_(torch.empty(shape), std=0.02, generator=gen)
    if name.startswith('video_model.') or name.startswith('txt_proj') or name.startswith('vid_proj'):
        fan_in = int(np.prod(shape[1:]))       # nn.Linear / nn.Conv2d default: kaiming_uniform(a=sqrt(5))
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound
    t = torch.randn(shape, generator=gen) * 0.02
    if name.endswith('word_embeddings.weight') or name.endswith('position_embeddings.weight'):
        t[cfg.pad_id].zero_()                  # nn.Embedding(padding_idx=1)
    return t


def _gather_clips(plan, rows, sources, dtype=None):
    """out clip i = sources[plan[i][0]] clip plan[i][1] of flat (clips * rows, d) matrices: ONE gather over the clip axis when every clip
    comes from the first source (one rank, or no negative drawn from another rank), else one copy per clip"""
    src0 = sources[0]
    d = src0.shape[1]
    dtype = dtype or src0.dtype
    if all(w == 0 for w, _ in plan) and src0.dtype == dtype:
        idx = torch.tensor([j for _, j in plan], dtype=torch.int64).to(src0.device, non_blocking=True)
        return src0.reshape(-1, rows, d).index_select(0, idx).reshape(len(plan) * rows, d)
    out = torch.empty(len(plan) * rows, d, dtype=dtype, device=src0.device)
    for i, (w, j) in enumerate(plan):
        out[i * rows:(i + 1) * rows].copy_(sources[w][j * rows:(j + 1) * rows])
    return out


class SelectClipsFn(torch.autograd.Function):
    """out clip i = sources[plan[i][0]] clip plan[i][1] on flat (clips * rows_per_clip, d) token matrices.  The plan lives on
    the host (the ITM negatives are drawn there), so forward is one copy per clip and backward adds each clip's gradient
    into its source in plan order -- deterministic also when a clip is selected more than once."""

    @staticmethod
    def forward(ctx, plan, rows, *sources):
        d = sources[0].shape[1]
        out = _gather_clips(plan, rows, sources)
        ctx.plan, ctx.rows = plan, rows
        ctx.shapes = [None if s is None else s.shape for s in sources]
        return out

    @staticmethod
    def backward(ctx, g):
        rows = ctx.rows
        grads = [None if shp is None else torch.zeros(shp, dtype=g.dtype, device=g.device) for shp in ctx.shapes]
        for i, (w, j) in enumerate(ctx.plan):
            grads[w][j * rows:(j + 1) * rows] += g[i * rows:(i + 1) * rows]
        return (None, None, *grads)


class FrozenInTime(nn.Module):
    def __init__(self, video_params, text_params, projection_dim=4096, load_checkpoint=None, projection='minimal',
                 load_temporal_fix='bilinear', config=config, task_names='EgoNCE_ITM_MLM', norm_layer=None, embed_dim=768,
                 compute_dtype=torch.bfloat16, path_config: PathConfig | None = None, init_seed: int = 0, text_fp32: bool = False,
                 video_fp8: bool = False, activation_checkpointing: bool | None = None):
        super().__init__()
        self.video_params = video_params
        self.text_params = text_params
        self.load_temporal_fix = load_temporal_fix
        self.config = dict(DEFAULT_YML, **(config or {}))
        self.task_names = task_names
        if not text_params['pretrained']:
            raise NotImplementedError("Huggingface text models require pretrained init.")       # model.py:65-66
        if not text_params['model'].startswith('roberta'):
            raise NotImplementedError(f"{text_params['model']} not implemented")
        if video_params['model'] != 'SpaceTimeTransformer':
            raise NotImplementedError(f"{video_params['model']} not implemented")                # model.py:97
        if projection not in ('minimal',):
            raise NotImplementedError(f"projection={projection!r}")
        if compute_dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("compute_dtype must be torch.bfloat16 or torch.float32")
        if path_config is None:
            path_config = PathConfig(depth=self.config['num_layers'], n_fuse=self.config['num_fuse_block'],
                                     frames=video_params['num_frames'], dim=embed_dim, heads=self.config['num_heads'],
                                     mlp_ratio=self.config['mlp_ratio'], vocab=self.config['vocab_size'],
                                     proj_dim=projection_dim, img=video_params.get('img_size', 224),
                                     drop_rate=ROBERTA_BASE_DROPOUT)
        self.cfg = path_config
        if self.cfg.head_dim != 64:
            raise NotImplementedError("attention kernels are built for head_dim 64")
        self.num_frames = self.cfg.frames
        self.num_fuse_block = self.cfg.n_fuse
        self.num_text_layer = self.cfg.depth
        self.compute_dtype = compute_dtype
        self.text_fp32 = bool(text_fp32) or SW.on('EGV_TEXT_FP32')
        # BASELINE.json configs[4] ("fp8 MFMA weight path"): the forward / data-gradient GEMMs of the video blocks on MX-fp8 (OCP
        # MXFP8 E4M3) weights and activations; weight gradients, attention, LayerNorm, the text tower and the heads stay bf16
        self.video_fp8 = (bool(video_fp8) or SW.on('EGV_VIDEO_FP8')) and compute_dtype == torch.bfloat16
        # bf16 mode: the text tower's residual stream (LayerNorm inputs / outputs, residual sums) stays fp32 between bf16 GEMMs, as under
        # torch.autocast (trainer/trainer_egoclip.py:143); EGV_TEXT_RES32=0 stores it in bf16 like the video tower's
        self.text_res32 = compute_dtype == torch.bfloat16 and SW.on('EGV_TEXT_RES32')
        self.patches_per_frame = self.cfg.n_patches
        # Activation checkpointing of the video blocks (the reference: yml `use_checkpoint`, torch.utils.checkpoint around every block,
        # model.py:239-266,326 -- on by default there, for 40 GB parts).  Here it is an OPTION: at configs[2] every saved activation fits
        # (34.6 GB of 288 GB) and recomputing costs a forward per block, so the default keeps everything.  activation_checkpointing=True,
        # or EGV_ACT_CHECKPOINT=1, or EGV_ACT_CHECKPOINT=yml (then the yml's use_checkpoint decides, as in the reference).
        if activation_checkpointing is None:
            v = SW.value('EGV_ACT_CHECKPOINT')
            activation_checkpointing = v == '1' or (v == 'yml' and bool(self.config.get('use_checkpoint')))
        self.act_checkpoint = bool(activation_checkpointing)

        gen = torch.Generator().manual_seed(init_seed)
        for name, shape in param_shapes(self.cfg, task_names).items():
            if name.endswith('position_ids'):
                _register(self, name, torch.arange(self.cfg.max_pos).expand(1, -1).clone(), buffer=True)
            else:
                _register(self, name, _init_tensor(name, shape, gen, self.cfg))
        self._P = None

        # Pretrained towers (model.py:69 RobertaModel.from_pretrained("roberta-base"); :80-94 the timm ViT-B/16 state dict loaded
        # with strict=False when no full checkpoint is given).  The reference hard-codes its file locations; here they come from
        # text_params['pretrained_path'] / video_params['pretrained_path'] (or EGV_ROBERTA_CHECKPOINT / EGV_VIT_CHECKPOINT).
        # Without a path the towers keep their seeded random init (there is no network for the hub downloads).
        tpath = text_params.get('pretrained_path') or SW.value('EGV_ROBERTA_CHECKPOINT') or None
        if tpath:
            self.load_pretrained_text(tpath)
        vpath = video_params.get('pretrained_path') or SW.value('EGV_VIT_CHECKPOINT') or None
        if vpath and video_params.get('pretrained', True) and load_checkpoint in ["", None]:
            self.load_pretrained_vit(vpath)

        if load_checkpoint not in ["", None]:
            from ..utils.checkpoint import load_checkpoint_file
            checkpoint = load_checkpoint_file(load_checkpoint)
            state_dict = checkpoint['state_dict']
            new_state_dict = state_dict_data_parallel_fix(state_dict, self.state_dict())
            new_state_dict = self._inflate_positional_embeds(new_state_dict)
            self.load_state_dict(new_state_dict, strict=False)

    # ------------------------------------------------------------------ pretrained towers
    @staticmethod
    def _read_state_dict(path):
        if str(path).endswith('.safetensors'):
            from safetensors.torch import load_file
            return load_file(str(path))
        sd = torch.load(path, map_location='cpu', weights_only=True)
        return sd.get('state_dict', sd) if isinstance(sd, dict) and 'state_dict' in sd else sd

    def _load_matching(self, loaded, what):
        """load_state_dict(strict=False) semantics on a name-mapped dict: unknown names are ignored, missing ones keep their init,
        a shape mismatch is an error (as in torch).  Returns the names that were loaded."""
        own = self.state_dict()
        done = []
        with torch.no_grad():
            for k, v in loaded.items():
                if k not in own:
                    continue
                if tuple(own[k].shape) != tuple(v.shape):
                    raise RuntimeError(f"{what}: size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}")
                own[k].copy_(v.to(own[k].dtype))
                done.append(k)
        self._P = None
        return done

    def load_pretrained_text(self, path):
        """model.py:69: HF roberta-base weights into text_model.* (checkpoint names with or without the 'roberta.' prefix; lm_head /
        pooler are not part of the path; the cross-attention layers of the fused blocks stay newly initialised, as they do under
        from_pretrained)"""
        sd = self._read_state_dict(path)
        mapped = {}
        for k, v in sd.items():
            k = k[len('roberta.'):] if k.startswith('roberta.') else k
            if k.startswith(('embeddings.', 'encoder.')) and not k.endswith('position_ids'):
                mapped['text_model.' + k] = v
        done = self._load_matching(mapped, 'load_pretrained_text')
        if not done:
            raise RuntimeError(f"load_pretrained_text: no RoBERTa tensors found in {path}")
        return done

    def load_pretrained_vit(self, path):
        """model.py:80-94: the timm ViT-B/16 state dict (cls_token, pos_embed, patch_embed.proj.*, blocks.i.{norm1,norm2,attn.qkv,
        attn.proj,mlp.fc1,mlp.fc2}.*, norm.*) into video_model.* with strict=False: the classifier head is dropped, temporal
        embedding / temporal attention / norm3 / fusion parameters keep their init"""
        sd = state_dict_data_parallel_fix(self._read_state_dict(path), {k[len('video_model.'):]: 0 for k in self.state_dict() if k.startswith('video_model.')})
        done = self._load_matching({'video_model.' + k: v for k, v in sd.items()}, 'load_pretrained_vit')
        if not done:
            raise RuntimeError(f"load_pretrained_vit: no ViT tensors found in {path}")
        return done

    # ------------------------------------------------------------------ plumbing
    def set_device(self, device):
        self.device = device

    def _pinned(self, key, shape, dtype):
        cache = self.__dict__.setdefault('_pin_cache', {})
        t = cache.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
            cache[key] = t
        return t

    def _overlap(self) -> bool:
        """Two-stream execution (EGV_NO_OVERLAP=1 turns it off).  It composes with DDP: the AccumulateGrad nodes the reducer
        hooks were created on the calling stream, autograd makes that stream wait for the producing stream before each
        accumulation, and the reducer orders a bucket's all-reduce after the calling stream at the time the bucket's last
        hook fires (tests/test_multirank_gpu.py runs both modes against the oracle)."""
        return not SW.on('EGV_NO_OVERLAP') and torch.cuda.is_available()

    def _fork_text(self, fn, uses=(), after=None, kind='text'):
        """Run text-encoder work (latency-bound: a dozen workgroups per kernel) on a second HIP stream so that it overlaps
        the video blocks; autograd replays each backward node on the stream of its forward, so the overlap holds for
        backward too.  The side stream starts after `after` (an event of the calling stream) or, by default, after everything
        queued on the calling stream so far.  `uses`: tensors allocated on the calling stream that fn reads -- they are
        recorded on the side stream, otherwise the caching allocator may recycle them while side-stream kernels (forward
        or backward) are still queued.  Returns (out, join); join() orders the calling stream after fn's work only."""
        if not self._overlap() or not SW.on('EGV_TEXT_STREAM') or (kind == 'tail' and not SW.on('EGV_TAIL_STREAM')):
            def nojoin():
                return None
            nojoin.event = None
            return fn(), nojoin
        aux = kind == 'aux' and SW.on('EGV_MLM_TOP_AUX')
        kind = 'text'                                          # the loss tails follow the text tower on its stream
        main = torch.cuda.current_stream()
        sides = self.__dict__.setdefault('_sides', {})
        # ONE companion stream for the text tower and the loss tails.  (A second one for the ITM pass's text prefix -- which queues
        # behind the MLM pass's fused text layers here -- was measured: + 13 ms per step; every further HIP stream costs the others.)
        if sides.get(kind) is None or sides[kind].device != main.device:
            sides[kind] = ops.companion_stream(main.device, 'text')
            # Text-side parameters get their gradients from nodes that ran on the companion stream, while their AccumulateGrad
            # nodes (kept alive by DDP's reducer) belong to the stream DDP was built on.  The engine orders the two streams before
            # every accumulation -- that is the documented behaviour this design relies on (see _overlap) -- so the per-step
            # warning about the mismatch is noise here.
            warn_off = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
            if warn_off is not None:
                warn_off(False)
        side = sides[kind]
        if aux:
            # kind 'aux': the text stream's own weight-gradient companion -- a fourth stream that exists anyway and is idle at both ends of
            # a step (every further HIP stream would share a hardware queue with one of the four: measured + 9 ms with two queues)
            side = ops.weight_gradient_stream_of(side)
        if after is None:
            side.wait_stream(main)
        else:
            side.wait_event(after)
        for t in uses:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(side)
        with torch.cuda.stream(side):
            out = fn()
            done = torch.cuda.Event()
            done.record(side)

        def join():
            torch.cuda.current_stream().wait_event(done)
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if torch.is_tensor(t):
                    t.record_stream(torch.cuda.current_stream())
        join.event = done                                      # (for a consumer on a THIRD stream: _fork_text(..., after=join.event))
        return out, join

    def _prepare_weights(self):
        """bf16 mode: one launch per step makes the W / W^T compute copies of every Linear weight (hipops.prepare_weights);
        embeddings, 1-D tensors and shapes that are not multiples of 64 keep the lazy per-tensor path."""
        if self.compute_dtype != torch.bfloat16:
            return
        lst = self.__dict__.get('_gemm_weights')
        if lst is None:
            lst = self.__dict__['_gemm_weights'] = [p for n, p in self.named_parameters()
                                                    if p.dim() == 2 and n.endswith('.weight') and 'embeddings' not in n]
            # Linears that read the same input run as one GEMM (csrc/egv_block.cpp): query / key / value of every RoBERTa layer
            # (roberta.py:257-270) and key / value of the text-to-image cross attention (roberta.py:241-242)
            groups = []
            c = self.cfg
            for i in range(c.depth):
                pfx = f'text_model.encoder.layer.{i}'
                sets = [[f'{pfx}.attention.self.{m}' for m in ('query', 'key', 'value')]]
                if i >= c.depth - c.n_fuse:
                    sets.append([f'{pfx}.crossattention_t2i.self.{m}' for m in ('key', 'value')])
                for names in sets:
                    groups.append(([self.p(n + '.weight') for n in names], [self.p(n + '.bias') for n in names]))
            self.__dict__['_gemm_groups'] = groups if SW.on('EGV_MERGE_PROJ') else []
            c = self.cfg
            mx = []
            if self.video_fp8 and c.dim % 128 == 0 and c.dim >= 384:
                for i in range(c.depth):
                    b = f'video_model.blocks.{i}.'
                    mx += [self.p(b + m + '.weight') for m in ('timeattn.qkv', 'timeattn.proj', 'attn.qkv', 'attn.proj', 'mlp.fc1', 'mlp.fc2')]
                    if i >= c.depth - c.n_fuse:
                        mx.append(self.p(b + 'attn.qkv_i2t.weight'))
            self.__dict__['_mx_weights'] = mx
        ops.prepare_weights(lst, self.compute_dtype, groups=self.__dict__['_gemm_groups'], fp8=self.__dict__['_mx_weights'])

    def p(self, name: str) -> torch.Tensor:
        if self._P is None:
            self._P = dict(self.named_parameters())
        return self._P[name]

    def _lin(self, x, prefix, act='none', gate=None, res1=None, res2=None, bias=True):
        return ops.linear(x, self.p(prefix + '.weight'), self.p(prefix + '.bias') if bias else None, act=act, gate=gate,
                          res1=res1, res2=res2)

    def _ln_skip(self, x, prefix, eps):
        return ops.layernorm_skip(x, self.p(prefix + '.weight'), self.p(prefix + '.bias'), eps)

    def _ln(self, x, prefix, eps):
        return ops.layernorm(x, self.p(prefix + '.weight'), self.p(prefix + '.bias'), eps)

    # ------------------------------------------------------------------ video side
    def _patch_tokens(self, video, cls_name):
        """video_transformer.py:353-372 / model.py:211-232: patch embed, CLS concat, pos + temporal embedding."""
        B, Fr = video.shape[0], video.shape[1]
        assert Fr == self.num_frames, (Fr, self.num_frames)                  # video_transformer.py:80
        # The patch embedding (Conv2d as im2col + GEMM) is a function of the pixels and the conv weights alone: the EgoNCE tower
        # (video_model.cls_token) and the shared prefix of the MLM / ITM passes (model-level cls_token) embed the same clips, so one
        # call serves both within a step (the reference embeds twice: model.py:224, video_transformer.py:353) and the conv's weight
        # gradient is formed once, from the sum of both consumers' gradients.  Keyed by the tensor's identity, version and grad mode;
        # only inside one forward() call (direct compute_video / infer calls embed on their own: their graphs are the caller's).
        w, b = self.p('video_model.patch_embed.proj.weight'), self.p('video_model.patch_embed.proj.bias')
        key = (id(video), video._version, video.data_ptr(), torch.is_grad_enabled())
        memo = self.__dict__.get('_pe_memo')                                # a dict only while forward() runs (its passes share a graph)
        emb = memo.get(key) if memo is not None else None
        if emb is None:
            emb = ops.patch_embed(video, w, b, self.compute_dtype)
            if memo is not None:
                memo[key] = emb
        x = ops.patch_tokens(video, w, b, self.p(cls_name), self.p('video_model.pos_embed'), self.p('video_model.temporal_embed'),
                             self.compute_dtype, emb=emb)
        return x.reshape(B * self.cfg.seq, self.cfg.dim)

    def _block_params(self, kind, i, fused):
        """parameter tensors of block i in the order the block-level C entry points expect (include/egovlp_hip.h)"""
        cache = self.__dict__.setdefault('_bp_cache', {})
        key = (kind, i, fused)
        lst = cache.get(key)
        if lst is None:
            if kind == 'video':
                pfx = f'video_model.blocks.{i}'
                names = [f'{pfx}.{m}.{t}' for m in ('timeattn.qkv', 'timeattn.proj', 'attn.qkv', 'attn.proj', 'mlp.fc1', 'mlp.fc2')
                         for t in ('weight', 'bias')]
                names += [f'{pfx}.{m}.{t}' for m in ('norm3', 'norm1', 'norm2') for t in ('weight', 'bias')]
                if fused:
                    a = pfx + '.attn'
                    names += [f'{a}.{m}.{t}' for m in ('qkv_text_i2t', 'qkv_i2t', 'proj_i2t') for t in ('weight', 'bias')]
                    names += [a + '.norm_i2t_i.weight', a + '.norm_i2t_i.bias', a + '.alpha_i2t']
            else:
                pfx = f'text_model.encoder.layer.{i}'
                names = [f'{pfx}.{m}.{t}' for m in ('attention.self.query', 'attention.self.key', 'attention.self.value',
                                                   'attention.output.dense', 'intermediate.dense', 'output.dense') for t in ('weight', 'bias')]
                names += [f'{pfx}.{m}.{t}' for m in ('attention.output.LayerNorm', 'output.LayerNorm') for t in ('weight', 'bias')]
                if fused:
                    ca = pfx + '.crossattention_t2i'
                    names += [f'{ca}.{m}.{t}' for m in ('self.query', 'self.key', 'self.value', 'output.dense') for t in ('weight', 'bias')]
                    names += [pfx + '.alpha_t2i']
            lst = cache[key] = [self.p(n) for n in names]
        return lst

    def _video_block(self, x, i, B, y=None, y_mask=None, L=0, next_block=None):
        """SpaceTimeBlock.forward (video_transformer.py:214-228) on the flat (B*S, d) token matrix: one C-ABI call forward, one
        backward (csrc/egv_block.cpp).  next_block = (index, L_next) of the block that consumes the result next (L_next > 0: a fused
        block over L_next text tokens): its first LayerNorm is folded into this call's output pass (ops.video_block, EGV_LN_FOLD)."""
        c = self.cfg
        next_ln = None
        if next_block is not None and next_block[0] < c.depth:
            pn = self._block_params('video', next_block[0], next_block[1] > 0)
            # norm3 weight / bias of the next block, its text length, and whether it runs as the CLS-only head call (a shorter save layout)
            next_ln = (pn[12], pn[13], next_block[1], bool(len(next_block) > 2 and next_block[2]))
        return ops.video_block(x, self._block_params('video', i, y is not None), B, c.frames, c.n_patches, c.heads, c.dim * c.mlp_ratio,
                               c.eps_video, y=y, y_mask=y_mask, L=L, fp8=bool(self.__dict__.get('_mx_weights')), next_ln=next_ln,
                               recompute=self.act_checkpoint)

    def _tail_ok(self, x):
        """may the LAST block of a video pass run in its CLS-only form (_video_block_tail)?  bf16 storage with the fp32 residual stream,
        no MX-fp8 operands (EGV_CLS_TAIL=0 runs the full block)"""
        return (SW.on('EGV_CLS_TAIL') and x.dtype == torch.bfloat16 and ops.video_res32(x, bool(self.__dict__.get('_mx_weights')))
                and self.cfg.dim % 256 == 0)

    def _video_block_tail(self, x, i, B, y=None, y_mask=None, L=0):
        """SpaceTimeBlock.forward (video_transformer.py:214-228) of a block whose output is only ever read at its CLS rows -- the last
        block of the video tower (forward_features returns x[:, 0], :392-394) and of the fused stack (self.norm(v)[:, 0], model.py:275).
        Every output row of a block depends on ALL rows through the space attention's keys and values, i.e. on norm3, the time attention
        with its projection and residual, norm1 and the K | V part of attn.qkv over all rows (one C call: ops.video_block_head); but
        the space attention's query, attn.proj, the image-to-text part, norm2 and the MLP act row by row, so for the CLS rows they run
        on B rows instead of B*S -- 66 % of the block's matrix work (and the gradients of the same share: they are exactly zero for
        the rows nobody reads) is never issued.  Returns the fp32 CLS rows (B, D), tagged for _video_out_norm."""
        return self._video_tail_rest(self._video_tail_head(x, i, B), i, B, y=y, y_mask=y_mask, L=L)

    def _video_tail_head(self, x, i, B):
        """the all-rows part of the CLS-only last block: (qkv_s [M, 3D], fp32 CLS rows of the block input)"""
        c = self.cfg
        p = self._block_params('video', i, False)
        return ops.video_block_head(x, [p[0], p[1], p[2], p[3], p[4], p[5], p[12], p[13], p[14], p[15]], B, c.frames, c.n_patches,
                                    c.heads, c.dim * c.mlp_ratio, c.eps_video)

    def _video_tail_rest(self, head, i, B, y=None, y_mask=None, L=0):
        """the B-row part of the CLS-only last block: ~45 launches of a few microseconds on the CLS rows (and three times that in
        backward); depends on the head through qkv_s and the CLS rows only, so a caller may issue it later and on another stream"""
        c = self.cfg
        fused = y is not None
        pfx = f'video_model.blocks.{i}'
        D = c.dim
        qkv, xc = head
        ctx = ops.cls_attention(qkv, B, c.seq, c.heads)                                   # the CLS query over all S keys (:129)
        s = self._lin(ctx, pfx + '.attn.proj')                                            # (B, D)
        sr = xc + s.float()                                                               # residual from x, not from the time residual (:222)
        if fused:
            a = pfx + '.attn'                                                             # image-to-text cross attention (:155-185)
            kv = self._lin(y, a + '.qkv_text_i2t')
            q = self._lin(self._ln(s, a + '.norm_i2t_i', c.eps_video), a + '.qkv_i2t')
            o = ops.plain_attention(q, kv[:, :D], kv[:, D:], B, c.heads, 1, L, 0.125, mask=y_mask)
            sr = sr + self.p(a + '.alpha_i2t') * self._lin(o, a + '.proj_i2t').float()
        h2 = ops.CastFn.apply(self._ln(sr, pfx + '.norm2', c.eps_video), self.compute_dtype)
        out = sr + self._lin(self._lin(h2, pfx + '.mlp.fc1', act='gelu'), pfx + '.mlp.fc2').float()
        out._cls_rows32 = True
        return out

    def _cls_rows(self, x, B, rows_per_sample):
        return x.reshape(B, rows_per_sample, -1)[:, 0].contiguous()

    def _video_out_norm(self, x, B, prefix, eps):
        """LayerNorm of the CLS rows of the video stream (video_model.norm, video_transformer.py:392-394; self.norm, model.py:275).
        With the fp32 residual stream (ops.video_block under EGV_VIDEO_RES32) the rows come from the stream's fp32 value and the
        LayerNorm runs in fp32, as under the reference's autocast; its result is the next Linear's operand either way."""
        if getattr(x, '_cls_rows32', False):                                # the CLS-only last block hands over the fp32 CLS rows themselves
            return ops.CastFn.apply(self._ln(x, prefix, eps), self.compute_dtype)
        x32 = ops.stream32(x)
        if x32 is None:
            return self._ln(self._cls_rows(x, B, self.cfg.seq), prefix, eps)
        return ops.CastFn.apply(self._ln(ops.stream_rows(x, x32, B, self.cfg.seq), prefix, eps), self.compute_dtype)

    def _proj_mlp(self, x, prefix):
        """txt_proj / vid_proj (model.py:105-115); cfg.proj_style 'linear': the fine-tune variant's txt ReLU-Linear / vid Linear
        (model_epic_charades.py:116-119)"""
        if self.cfg.proj_style == 'linear':
            return self._lin(torch.relu(x), 'txt_proj.1') if prefix == 'txt_proj' else self._lin(x, 'vid_proj.0')
        x = ops.linear(x, self.p(prefix + '.0.weight'), None, act='relu')
        x = self._lin(x, prefix + '.2', act='relu')
        return self._lin(x, prefix + '.4')

    # ------------------------------------------------------------------ text side
    def _drop_p(self) -> float:
        """hidden / attention-probability dropout of the text tower, train mode only: the pretrained roberta-base value (the
        constructor sets cfg.drop_rate = 0.1; the yml drop_rate does not reach the text tower in the reference, model.py:68 vs
        :127-137)"""
        return float(self.cfg.drop_rate) if self.training else 0.0

    def _drop_seed(self) -> int:
        """One 32-bit seed per dropout site and call.  Model-owned counter stream (per rank), so the default CPU / numpy
        generators -- whose draw order the ITM negative sampling shares with the reference -- are left untouched."""
        st = self.__dict__.setdefault('_drop_state', None)
        if st is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            st = self.__dict__['_drop_state'] = [(0x9E3779B1 * (self.__dict__.get('_drop_base', 0) + 1) + 0x632BE5AB * rank)
                                                 & 0xFFFFFFFF, 0]
        st[1] += 1
        return (st[0] + 0x85EBCA6B * st[1]) & 0xFFFFFFFF

    def seed_dropout(self, seed: int):
        """restart the dropout mask stream (tests / reproducible runs)"""
        self.__dict__['_drop_base'] = int(seed)
        self.__dict__['_drop_state'] = None

    def _text_embeddings(self, input_ids, dtype=None):
        e = ops.text_embed(input_ids, self.p('text_model.embeddings.word_embeddings.weight'),
                           self.p('text_model.embeddings.position_embeddings.weight'),
                           self.p('text_model.embeddings.token_type_embeddings.weight'), self.cfg.pad_id, dtype or self.compute_dtype)
        e = self._ln(e, 'text_model.embeddings.LayerNorm', self.cfg.eps_text)
        p = self._drop_p()
        return ops.dropout_add(e, p, self._drop_seed()) if p > 0 else e                  # roberta.py:203

    @staticmethod
    def _key_mask(attention_mask):
        """additive (B, L) fp32 key mask: (1 - m) * finfo(fp32).min (get_extended_attention_mask, roberta.py:826)."""
        return ((1.0 - attention_mask.to(torch.float32)) * F32_MIN).contiguous()

    def _text_layer(self, hid, mask, i, B, L, enc=None, exact=False):
        """RobertaLayer.forward (roberta.py:444-505); enc = video tokens (B*S, d) for the fused layers.  One C-ABI call forward,
        one backward.  Train mode: dropout on the attention probabilities (roberta.py:313) inside the attention kernels and on
        every dense output before its residual add (:342, :422); one seed per site, drawn here in the sites' order."""
        c = self.cfg
        p = self._drop_p()
        fused = enc is not None
        seeds = [0] * 6
        if p > 0:
            seeds[0], seeds[1] = self._drop_seed(), self._drop_seed()
            if fused:
                seeds[2], seeds[3] = self._drop_seed(), self._drop_seed()
            seeds[4] = self._drop_seed()
        res32 = self.text_res32 and hid.dtype == torch.float32 and not (self.text_fp32 and exact)
        return ops.text_layer(hid, mask, self._block_params('text', i, fused), B, L, c.heads, c.dim * c.mlp_ratio, c.eps_text,
                              enc=enc, S=c.seq if fused else 0, drop_p=p, seeds=seeds, res32=res32)

    def _text_operand(self, t, exact_ok=False):
        """text states as a GEMM / attention operand outside the text layers (projection heads, the video blocks' image-to-text
        keys): the fp32 residual stream is rounded to the compute dtype there, as autocast rounds a Linear's input"""
        if t.dtype == self.compute_dtype or (exact_ok and self.text_fp32):
            return t
        return ops.CastFn.apply(t, self.compute_dtype)

    def _text_dtype(self):
        """Storage type of the TEXT-ONLY tower pass (compute_text / compute_text_tokens: the EgoNCE text embedding).  Option
        text_fp32=True (or EGV_TEXT_FP32=1) runs that pass with fp32 storage and exact-fp32 MFMA inside a bf16 model: its output
        error against the reference's fp32 values drops from 1.3e-2 to 2e-6 and the EgoNCE loss error from 7.5e-4 to 3.0e-4, for
        +5.5 ms per step (the pass is 256 token rows, but its fp32 GEMMs take 60-100 us instead of 17 and make the companion
        stream the long pole of the EgoNCE backward).  The fused passes (MLM / ITM), whose text side exchanges tokens with the bf16
        video side every layer, stay in the compute dtype.  Off by default."""
        return torch.float32 if (self.text_fp32 or self.text_res32) else self.compute_dtype

    # ------------------------------------------------------------------ reference API
    def compute_text(self, text_data):
        """model.py:491-505: RoBERTa last_hidden_state[:, 0] -> txt_proj."""
        self._prepare_weights()
        ids, am = text_data['input_ids'], text_data['attention_mask']
        B, L = ids.shape
        hid = self._text_embeddings(ids, self._text_dtype())
        mask = self._key_mask(am)
        for i in range(self.cfg.depth):
            hid = self._text_layer(hid, mask, i, B, L, exact=True)
        return self._proj_mlp(self._text_operand(self._cls_rows(hid, B, L), exact_ok=True), 'txt_proj')

    def compute_text_tokens(self, text_data):
        """model.py:507-522: all token states -> txt_proj."""
        self._prepare_weights()
        ids, am = text_data['input_ids'], text_data['attention_mask']
        B, L = ids.shape
        hid = self._text_embeddings(ids, self._text_dtype())
        mask = self._key_mask(am)
        for i in range(self.cfg.depth):
            hid = self._text_layer(hid, mask, i, B, L, exact=True)
        return self._proj_mlp(self._text_operand(hid, exact_ok=True), 'txt_proj').reshape(B, L, -1)

    def _video_features(self, video_data, fork_rest=False):
        """fork_rest (forward() only): the B-row part of the CLS-only last block and the final LayerNorm -- ~45 dependent launches of a few
        microseconds, three times that in backward -- run on the text stream's weight-gradient companion instead of the calling stream;
        returns (features, join) then.  Forward: the calling stream goes on with the shared prefix.  Backward: the chain's input is the
        EgoNCE loss's gradient, which exists at the very start of the pass, so the companion has run it long before the calling stream
        reaches the tower (1.0 -> 0.45 ms of calling-stream idle time in front of the tower's backward)."""
        self._prepare_weights()          # (a no-op when the step's copies exist: direct compute_video / Feature_Extraction calls find them too)
        B = video_data.shape[0]
        x = self._patch_tokens(video_data, 'video_model.cls_token')
        last = self.cfg.depth - 1
        for i in range(last):
            x = self._video_block(x, i, B, next_block=(i + 1, 0, i + 1 == last and self._tail_ok(x)))
        if fork_rest and self._tail_ok(x) and SW.on('EGV_TAIL_REST_AUX'):
            head = self._video_tail_head(x, last, B)
            return self._fork_text(lambda: self._video_out_norm(self._video_tail_rest(head, last, B), B, 'video_model.norm', self.cfg.eps_video),
                                   uses=head, kind='aux')
        x = self._video_block_tail(x, last, B) if self._tail_ok(x) else self._video_block(x, last, B)
        feats = self._video_out_norm(x, B, 'video_model.norm', self.cfg.eps_video)
        return (feats, None) if fork_rest else feats

    def compute_video(self, video_data):
        """model.py:524-530: SpaceTimeTransformer.forward_features (video_model.cls_token / video_model.norm) -> vid_proj."""
        return self._proj_mlp(self._video_features(video_data), 'vid_proj')

    def _video_prefix(self, video, next_L=0):
        """model.py:211-243 video side: model-level cls_token, pos/temporal embedding and the depth - n_fuse unfused blocks.
        Depends on the pixels only (no dropout, no text), which is what lets forward() share it between MLM and ITM.
        next_L: text length of the fused stack that consumes the prefix (its first block's LayerNorm rides in the last prefix block)."""
        B = video.shape[0]
        n_plain = self.cfg.depth - self.cfg.n_fuse
        v = self._patch_tokens(video, 'cls_token')
        for i in range(n_plain):
            v = self._video_block(v, i, B, next_block=(i + 1, next_L if i + 1 == n_plain else 0))
        return v

    def _text_prefix(self, input_ids, attention_mask):
        """embeddings + the depth - n_fuse unfused RoBERTa layers (model.py:247-257); returns (hidden, additive key mask)"""
        B, L = input_ids.shape
        mask = self._key_mask(attention_mask)
        t = self._text_embeddings(input_ids, torch.float32 if self.text_res32 else None)
        for i in range(self.cfg.depth - self.cfg.n_fuse):
            t = self._text_layer(t, mask, i, B, L)
        return t, mask

    def _fused_stack(self, video, input_ids, attention_mask, need_video_out=True, video_prefix=None, text_prefix=None, defer_top=False):
        """model.py:211-271 / :295-357: model-level cls_token, unfused prefix, then fused steps where both sides read the
        other modality's state from BEFORE the step.  With need_video_out=False (MLM branch) the last video block, whose
        output the reference computes and discards, is skipped.  video_prefix: the already computed
        output of _video_prefix for these clips (video is then ignored); text_prefix: ((hidden, mask), join) of a
        _text_prefix already forked by the caller.
        Both halves of a fused step depend only on the previous step, so the text layer runs on the side stream while the
        video block runs on the calling stream (events both ways per step).
        defer_top (need_video_out=False only): the LAST fused text layer -- which has no video block beside it -- is not issued; the
        third result is a callable that issues it (on the text stream) and returns its output.  The caller may create it AFTER other
        graph parts: the autograd engine then runs its backward BEFORE theirs."""
        c = self.cfg
        B, L = input_ids.shape
        n_plain = c.depth - c.n_fuse
        if text_prefix is None:
            text_prefix = self._fork_text(lambda: self._text_prefix(input_ids, attention_mask), uses=(input_ids, attention_mask))
        (t, mask), join = text_prefix
        v = self._video_prefix(video, next_L=L) if video_prefix is None else video_prefix
        join()
        overlap = self._overlap()
        for i in range(n_plain, c.depth):
            last = i == c.depth - 1
            ev = None
            if overlap:
                ev = torch.cuda.Event()
                ev.record()                                                   # v (and t, joined above) are ready here
            if last and defer_top and not need_video_out:
                def top(t=t, v=v, ev=ev, i=i):
                    t_top, join_top = self._fork_text(lambda: self._text_layer(t, mask, i, B, L, enc=v), uses=(v, mask, t), after=ev, kind='aux')
                    return t_top, join_top
                return None, t, top
            t_new, join = self._fork_text(lambda: self._text_layer(t, mask, i, B, L, enc=v), uses=(v, mask, t), after=ev)
            tail = last and need_video_out and self._tail_ok(v)             # (the result is read at the CLS rows only: model.py:275)
            to_tail = i + 2 == c.depth and need_video_out and self._tail_ok(v)     # the next block is the stack's last and runs as the CLS-only tail
            nxt = None if (i + 1 == c.depth or (i + 2 == c.depth and not need_video_out)) else (i + 1, 0 if to_tail else L, to_tail)
            if last and not need_video_out:
                v_new = None
            elif tail:
                v_new = self._video_block_tail(v, i, B, y=self._text_operand(t), y_mask=mask, L=L)
            else:
                v_new = self._video_block(v, i, B, y=self._text_operand(t), y_mask=mask, L=L, next_block=nxt)
            join()
            v, t = v_new, t_new
        return (v, t, None) if defer_top else (v, t)

    def infer(self, data, video_only=False, return_embeds=True, task_names=None, ret=None):
        """model.py:189-367.  (The reference's mutable default ``ret={}`` is replaced by a fresh dict.)"""
        ret = {} if ret is None else ret
        self._prepare_weights()
        text_data, video_data = data['text'], data['video']
        if task_names is not None:
            self.task_names = task_names
        c = self.cfg
        if 'EgoNCE' in self.task_names or 'Dual' in self.task_names:                                # 'Dual': model_epic_charades.py:196-203
            text_embeddings, join = self._fork_text(lambda: self.compute_text(text_data),
                                                    uses=(text_data['input_ids'], text_data['attention_mask']))
            video_embeddings = self.compute_video(video_data)
            join()
            if return_embeds:
                ret.update({'text_embeds': text_embeddings, 'video_embeds': video_embeddings})
        if 'ITM' in self.task_names:
            B, L = text_data['input_ids'].shape
            v, t = self._fused_stack(video_data, text_data['input_ids'], text_data['attention_mask'],
                                     video_prefix=data.get('_video_prefix'), text_prefix=data.get('_text_prefix'))
            ret.update({'cross_attn_itm_logits': self._itm_logits(v, t, B, L)})
        if 'MLM' in self.task_names:
            B, L = data['text_mlm_ids'].shape
            logits = self._mlm_logits_padded(video_data, data['text_mlm_ids'], text_data['attention_mask'],
                                             video_prefix=data.get('_video_prefix'), text_prefix=data.get('_text_prefix'))
            ret.update({'cross_attn_mlm_logits': logits.reshape(B, L, -1)[..., :c.vocab]})
            ret['_mlm_logits_padded'] = logits
        return ret

    def _itm_logits(self, v, t, B, L):
        """ITM head (model.py:275-293): norm on the video CLS rows, the two cross-modal transforms and poolers, the 2-way score"""
        c = self.cfg
        vf = self._video_out_norm(v, B, 'norm', c.eps_model_norm)                           # self.norm(v)[:, 0]  (:275)
        tf = self._lin(self._text_operand(self._cls_rows(t, B, L)), 'cross_modal_text_transform')
        vf = self._lin(vf, 'cross_modal_video_transform')
        ct = self._lin(tf, 'cross_modal_text_pooler.dense', act='tanh')
        cv = self._lin(vf, 'cross_modal_video_pooler.dense', act='tanh')
        return self._lin(torch.cat([ct, cv], dim=-1), 'itm_score.fc')

    def _mlm_logits_padded(self, video, mlm_ids, attention_mask, video_prefix=None, text_prefix=None):
        """MLM branch of infer (model.py:346-365): fused stack without its dead last video block, then the head"""
        _, t = self._fused_stack(video, mlm_ids, attention_mask, need_video_out=False, video_prefix=video_prefix,
                                 text_prefix=text_prefix)
        return self._mlm_head(t)

    def _mlm_head(self, t):
        """MLM tail (model.py:360-365, heads.py:38-50); the vocabulary axis is padded to a multiple of 128 so that the
        logits rows stay 16-byte aligned (padded columns are excluded from the CE and get zero gradient)."""
        c = self.cfg
        t = self._lin(self._text_operand(t), 'cross_modal_text_transform')
        t = self._lin(t, 'mlm_score.transform.dense', act='gelu')
        t = self._ln(t, 'mlm_score.transform.LayerNorm', c.eps_mlm)
        return ops.vocab_linear(t, self.p('mlm_score.decoder.weight'), self.p('mlm_score.bias'), c.vocab)

    def _begin_step(self):
        """per-step gradient bookkeeping of the block executor (hipops): forget what an aborted step left behind"""
        ids = self.__dict__.get('_param_ids')
        if ids is None:
            ids = self.__dict__['_param_ids'] = frozenset(id(p) for p in self.parameters())
        ops.begin_step(ids)
        # under DistributedDataParallel the reducer copies every gradient the moment autograd produces it: the block backward
        # calls must then join their weight-gradient stream before they return (hipops.set_defer_wgrad_join)
        DDP = torch.nn.parallel.DistributedDataParallel
        if hasattr(DDP, '_active_ddp_module'):
            defer = DDP._active_ddp_module is None
        else:
            # this torch does not tell whether a DDP forward is active: defer only where no reducer can be listening (no process
            # group at all, or the flat gradient sync of trainer/grad_sync.py, which is ordered behind the launch itself)
            defer = not (dist.is_available() and dist.is_initialized()) or ops._pack_hook[0] is not None
        ops.set_defer_wgrad_join(defer)

    def forward(self, data, n_embeds, v_embeds, allgather, n_gpu, args, config, loss_egonce, gpu, return_embeds=True,
                task_names='EgoNCE_ITM_MLM'):
        """model.py:370-487.  Returns (loss, loss_dict, ret)."""
        self.__dict__['_pe_memo'] = {}
        try:
            return self._forward(data, n_embeds, v_embeds, allgather, n_gpu, args, config, loss_egonce, gpu, return_embeds, task_names)
        finally:
            self.__dict__['_pe_memo'] = None

    def _forward(self, data, n_embeds, v_embeds, allgather, n_gpu, args, config, loss_egonce, gpu, return_embeds, task_names):
        ret, loss_dict = {}, {}
        self._begin_step()
        if 'Feature_Extraction' in task_names:                                                   # :375-377
            return self.compute_video(data['video'])
        world = getattr(args, 'world_size', 1)
        gather = (lambda t: allgather(t, n_gpu, args))
        c = self.cfg
        joins = []                                   # companion-stream work the calling stream has not been ordered after yet
        loss_terms = []
        itm_w = None
        want_itm = 'ITM' in task_names
        if 'EgoNCE' in task_names:                                                               # :380-400
            # infer(task_names='EgoNCE') with its tail on the companion stream: the video tower runs on the calling stream, the text
            # tower on the text stream (as in infer); the two projection heads' tail -- vid_proj, the gathers, the three similarity
            # matrices, the EgoNCE loss and the ITM sampling weights, ~30 launches of a few microseconds -- follows the text tower
            # on the text stream instead of sitting between the EgoNCE tower and the MLM / ITM prefix on the calling stream (0.8 ms
            # forward, 0.5 ms backward per step)
            self._prepare_weights()
            self.task_names = 'EgoNCE'
            text_data = data['text']
            # The unfused RoBERTa layers run three times per step on 256 token rows each (EgoNCE tower, MLM prefix, ITM prefix): every
            # launch is latency-bound, so two passes of M = 256 cost twice what one pass of M = 512 costs.  The EgoNCE tower's first
            # depth - n_fuse layers and the MLM pass's text prefix share weights and both inputs exist now: they run as ONE pass over
            # the concatenated batch (per-sample results unchanged: every kernel is batch-independent; dropout masks are a function
            # of (seed, element) and stay independent per element); the ITM prefix cannot join -- its batch is drawn from the EgoNCE
            # similarities.  EGV_TEXT_BATCH=0 restores the separate passes.
            pair_prefix = ('MLM' in task_names and want_itm and c.depth > c.n_fuse and not SW.on('EGV_NO_PREFIX_SHARING')
                           and not self.text_fp32 and SW.on('EGV_TEXT_BATCH')
                           and data['text_mlm_ids'].shape == text_data['input_ids'].shape)
            txt_mlm_pair = None
            if pair_prefix:
                def text_pair():
                    ids, am = text_data['input_ids'], text_data['attention_mask']
                    B, L = ids.shape
                    mask2 = self._key_mask(torch.cat([am, am]))
                    hid = self._text_embeddings(torch.cat([ids, data['text_mlm_ids']]), self._text_dtype())
                    for i in range(c.depth - c.n_fuse):
                        hid = self._text_layer(hid, mask2, i, 2 * B, L)
                    he, hm, mask = hid[:B * L], hid[B * L:], mask2[:B]
                    for i in range(c.depth - c.n_fuse, c.depth):
                        he = self._text_layer(he, mask, i, B, L, exact=True)
                    return self._proj_mlp(self._text_operand(self._cls_rows(he, B, L), exact_ok=True), 'txt_proj'), hm, mask
                (text_embeds_l, hm_pair, mask_pair), join_txt = self._fork_text(
                    text_pair, uses=(text_data['input_ids'], text_data['attention_mask'], data['text_mlm_ids']))
                txt_mlm_pair = ((hm_pair, mask_pair), join_txt)
            else:
                text_embeds_l, join_txt = self._fork_text(lambda: self.compute_text(text_data),
                                                          uses=(text_data['input_ids'], text_data['attention_mask']))
            feats, join_feats = self._video_features(data['video'], fork_rest=True)
            ev_feats = join_feats.event if join_feats is not None else None      # the features come from the companion stream: the tails wait for it, not for the calling stream
            if join_feats is not None:
                joins.append(join_feats)
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else getattr(args, 'rank', 0)
            bsz = data['video'].size(0)

            def egonce_tail(weights):
                video_embeds_l = self._proj_mlp(feats, 'vid_proj')
                video_embeds = gather(ops.CastFn.apply(video_embeds_l, torch.float32))
                text_embeds = gather(ops.CastFn.apply(text_embeds_l, torch.float32))
                n_all, v_all = gather(n_embeds.float()), gather(v_embeds.float())
                output = sim_matrix(text_embeds, video_embeds)
                if config['loss']['type'] == 'EgoNCE':
                    sim_v = sim_matrix(v_all, v_all)
                    sim_n = sim_matrix(n_all, n_all)
                    loss_e, mask_bool, temp = loss_egonce(output, sim_v, sim_n)
                else:
                    loss_e, mask_bool, temp = loss_egonce(output)
                w_host = ev_w = None
                if weights:
                    # The sampling weights of the ITM branch (:438-447) only depend on the EgoNCE branch: compute them and start the
                    # device->host copy now, so that the copy (and the host-side draw below) overlaps the MLM pass instead of
                    # draining the GPU queue.
                    with torch.no_grad():
                        sl = slice(bsz * rank, bsz * (rank + 1))
                        w_v2t = F.softmax(output[sl] / temp, dim=1).masked_fill(mask_bool[sl], 0)
                        w_t2v = F.softmax(output.t()[sl] / temp, dim=1).masked_fill(mask_bool[sl], 0)
                        w_dev = torch.stack([w_v2t, w_t2v]).float()
                        w_host = self._pinned('itm_w', w_dev.shape, torch.float32)     # cached: pinned allocation stalls the device
                        w_host.copy_(w_dev, non_blocking=True)
                        ev_w = torch.cuda.Event()
                        ev_w.record()
                return (video_embeds_l, output, loss_e), (w_host, ev_w)

            def finish_egonce():
                (video_embeds_l, output, loss_e), _ = tail_state['out']

                def join_egonce():
                    join_txt()
                    tail_state['join']()
                    cur = torch.cuda.current_stream()
                    for t_ in (video_embeds_l, output, loss_e):
                        t_.record_stream(cur)
                joins.append(join_egonce)
                ret.update({'text_embeds': text_embeds_l, 'video_embeds': video_embeds_l, 'sim_v2t': output, 'sim_t2v': output.t()})
                loss_dict['EgoNCE'] = loss_e
                loss_terms.insert(0, (1.0, loss_e))
            tail_state = {}
            loss_dict['EgoNCE'] = None                      # (keeps the reference's key order)
            late_tail = want_itm and SW.on('EGV_EGONCE_TAIL_LATE')
            if late_tail:
                # The ITM draw needs the similarities NOW, but the engine runs backward nodes in reverse creation order and the text
                # stream executes in order: a differentiable tail created here has its backward enqueued behind the backward of both
                # text prefixes (created below), and the EgoNCE tower's first backward block waits for it (1.1 ms of calling-stream
                # idle time).  So: the sampling weights from a no-grad evaluation here, the differentiable tail (the same ~30 small
                # launches, on the text stream) right after the text prefixes of the MLM and ITM passes have been created.
                with torch.no_grad():
                    (_, itm_w), _join_w = self._fork_text(lambda: egonce_tail(True), uses=(feats, n_embeds, v_embeds), kind='tail', after=ev_feats)

                def make_tail():
                    tail_state['out'], tail_state['join'] = self._fork_text(lambda: egonce_tail(False), uses=(feats, n_embeds, v_embeds), kind='tail', after=ev_feats)
                    finish_egonce()
            else:
                tail_state['out'], tail_state['join'] = self._fork_text(lambda: egonce_tail(want_itm), uses=(feats, n_embeds, v_embeds), kind='tail', after=ev_feats)
                itm_w = tail_state['out'][1]
                finish_egonce()

        itm_pre = None
        if 'ITM' in task_names:                                                                  # :426-447
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else getattr(args, 'rank', 0)
            bsz = data['video'].size(0)
            if itm_w is None:
                raise ValueError("task 'ITM' needs the EgoNCE branch in the same call: its hard negatives are drawn from the EgoNCE similarities (model.py:438-447)")
            w_host, ev = itm_w
            # the gathers of the ITM branch (:429-431) are issued here so that the text stream can start on the ITM
            # batch while the calling stream is still busy with the MLM pass
            # (the reference all-gathers the pixels here, :430; with the shared prefix only the prefix tokens of the clips that
            # are actually drawn from another rank travel -- trainer/exchange.py)
            share_px = ('MLM' in task_names and c.depth > c.n_fuse and not SW.on('EGV_NO_PREFIX_SHARING'))
            all_video = data['video'] if (world == 1 or share_px) else gather(data['video'])
            all_text_ids = gather(data['text']['input_ids'])
            all_text_masks = gather(data['text']['attention_mask'])
            ev_in = None
            if torch.cuda.is_available():
                ev_in = torch.cuda.Event()
                ev_in.record()
            itm_pre = (rank, bsz, w_host, ev, all_video, all_text_ids, all_text_masks, ev_in)

        # The unfused video prefix (patch embedding + the first depth - n_fuse blocks under the model-level cls_token) is a
        # function of the pixels alone, and the ITM batch is the MLM batch with some clips swapped for hard negatives
        # (model.py:449-468).  The reference recomputes it in the ITM pass; here it is computed once and the ITM pass gathers
        # clip rows from it (same values, and autograd sums both consumers' gradients exactly as the two passes would).
        share_prefix = ('MLM' in task_names and 'ITM' in task_names and c.depth > c.n_fuse
                        and not SW.on('EGV_NO_PREFIX_SHARING'))
        v_pre = None
        data_mlm = data
        if share_prefix:
            am = data['text']['attention_mask']
            txt_mlm = txt_mlm_pair if ('EgoNCE' in task_names and txt_mlm_pair is not None) else \
                self._fork_text(lambda: self._text_prefix(data['text_mlm_ids'], am), uses=(data['text_mlm_ids'], am))
            v_pre = self._video_prefix(data['video'], next_L=data['text_mlm_ids'].shape[1])   # overlaps the MLM text prefix (its last block also normalises for the MLM pass's first fused block)
            data_mlm = dict(data, _video_prefix=v_pre, _text_prefix=txt_mlm)

        def itm_draw():
            """model.py:438-468: ITM labels and hard negatives, then the ITM batch's text side.  With the shared prefix this runs BEFORE
            the MLM pass is enqueued: the draw needs the EgoNCE branch on the host (one event wait) while the calling stream still
            has the whole unfused video prefix queued, the ITM text prefix then sits on the text stream AHEAD of the MLM pass's
            fused text layers (which wait for the video blocks step by step) -- it is ready when the ITM fused stack starts -- and
            in backward the MLM pass's text side is enqueued before the ITM text prefix, i.e. it runs beside the ITM pass instead
            of holding up the first MLM video block (1.8 + 3.2 ms of calling-stream idle time per step)."""
            rank, bsz, w_host, ev, all_video, all_text_ids, all_text_masks, ev_in = itm_pre
            pos_len = bsz // 2
            itm_labels = torch.cat([torch.ones(pos_len), torch.zeros(bsz - pos_len)])
            itm_labels = itm_labels[torch.randperm(itm_labels.size(0))]
            # ONE device->host copy (started above); negatives are drawn on the host in the reference's ORDER of draws
            # (randperm, then np.random.rand and torch.multinomial per negative, model.py:438,459-468) instead of one sync per
            # sample.  All draws use the CPU generators; the reference's multinomial runs on the CUDA generator when the weights
            # live on a GPU, so equal seeds give equal negatives only against a CPU run of the reference (the golden fixtures).
            ev.synchronize()
            w_cpu = w_host
            vid_idx = torch.arange(bsz) + rank * bsz
            txt_idx = vid_idx.clone()
            neg_log = []
            for idx in range(bsz):
                if itm_labels[idx] == 1:
                    continue
                if np.random.rand() > 0.5:
                    j = torch.multinomial(w_cpu[1, idx] + 1e-9, 1).item()
                    vid_idx[idx] = j
                    neg_log.append((idx, 'video', j))
                else:
                    j = torch.multinomial(w_cpu[0, idx] + 1e-9, 1).item()
                    txt_idx[idx] = j
                    neg_log.append((idx, 'text', j))
            dev = data['video'].device
            # one pinned staging buffer + one async copy for the three small host tensors (pageable .to() copies block)
            stage = self._pinned('itm_idx', (3, bsz), torch.int64)
            stage[0].copy_(vid_idx)
            stage[1].copy_(txt_idx)
            stage[2].copy_(itm_labels.long())
            idx_dev = stage.to(dev, non_blocking=True)
            vid_list = vid_idx.tolist()
            req = None
            if world > 1 and share_prefix:
                # which clips does every rank need from which owner?  The exchange of the drawn ids starts NOW (text stream: idle at
                # this point, and the ids get their own host->device copy there), the table is read in run_itm()
                from ..trainer.exchange import start_request_gather
                req, _ = self._fork_text(lambda: start_request_gather(vid_list, stage[0].to(dev, non_blocking=True), rank, bsz, world), after=ev_in)
            vid_idx, txt_idx, labels_dev = idx_dev[0], idx_dev[1], idx_dev[2]
            if share_prefix:
                def text_itm():            # runs on the text stream: it only needs the gathered ids, not the MLM pass
                    ti = stage[1].to(dev, non_blocking=True)
                    ids, am = all_text_ids.index_select(0, ti), all_text_masks.index_select(0, ti)
                    return self._text_prefix(ids, am) + (ids, am)
                out, join = self._fork_text(text_itm, uses=(all_text_ids, all_text_masks), after=ev_in)
                data_itm = {'text': {'input_ids': out[2], 'attention_mask': out[3]}, '_text_prefix': ((out[0], out[1]), join)}
            else:
                data_itm = {'text': {'input_ids': all_text_ids.index_select(0, txt_idx),
                                     'attention_mask': all_text_masks.index_select(0, txt_idx)}}
            return dict(itm_labels=itm_labels, neg_log=neg_log, vid_idx=vid_idx, vid_list=vid_list, labels_dev=labels_dev, data_itm=data_itm, req=req)

        itm_state = itm_draw() if ('ITM' in task_names and share_prefix and SW.on('EGV_ITM_DRAW_EARLY')) else None
        if itm_state is not None and 'EgoNCE' in task_names and late_tail:
            make_tail()          # after both text prefixes were created, before the fused stacks: see late_tail above

        def run_mlm():                                                                           # :404-422
            # infer(task_names='MLM') with the head, the cross entropy and the loss arithmetic on the text stream: in backward the
            # head and the last fused text layer -- which the first MLM video block's backward has to wait for -- then run beside
            # the ITM pass's backward instead of after it (2.2 ms of calling-stream idle time per step)
            self._prepare_weights()
            self.task_names = 'MLM'
            # EGV_MLM_TOP_LATE (round 6): the pass's top -- its last fused text layer (no video block beside it), the head and the cross
            # entropy, ~1 ms of small launches on the text stream in each direction -- is CREATED after the ITM pass.  The engine runs
            # backward nodes in reverse creation order: the top's backward is then the first thing the text stream runs, beside the
            # B-row chain of the ITM pass's CLS-only last block on the calling stream (1.1 ms with the chip idle), instead of after the ITM
            # pass's text layers, where the first MLM video block's backward waited for it (1.2 ms hole at the ITM -> MLM boundary).
            late = ('ITM' in task_names and not SW.on('EGV_ITM_FIRST') and SW.on('EGV_MLM_TOP_LATE') and self._overlap() and SW.on('EGV_TEXT_STREAM'))
            out = self._fused_stack(data_mlm['video'], data['text_mlm_ids'], data['text']['attention_mask'], need_video_out=False,
                                    video_prefix=data_mlm.get('_video_prefix'), text_prefix=data_mlm.get('_text_prefix'), defer_top=late)
            if late and out[2] is not None:
                top = out[2]

                def finish():
                    t_top, join_top = top()
                    joins.append(join_top)
                    mlm_head(t_top, kind='aux')           # (same stream as the layer: in order behind it, no event)
                mlm_late.append(finish)
                return
            mlm_head(out[1])

        def mlm_head(t_mlm, kind='tail'):
            labels = data['text_mlm_labels'].reshape(-1)

            def mlm_tail():
                logits = self._mlm_head(t_mlm)
                ce_sum = ops.cross_entropy_sum(logits, labels, c.vocab, -100)
                # labels outside [0, vocab) other than the ignore index are skipped by the CE kernels (they cannot index the
                # logits): count exactly the labels that contribute, so that a collator / vocabulary mismatch cannot mis-normalise
                # the loss
                cnt = ((labels >= 0) & (labels < c.vocab)).sum().to(torch.float32)
                # the reference all-gathers the (B*L, 50265) logits (412 MB at W=8) and takes the global mean; gathering the
                # two per-rank scalars gives the identical loss and, through AllGather_multi.backward, identical gradients.
                tot = gather(torch.stack([ce_sum, cnt]).reshape(1, 2))
                return logits, tot[:, 0].sum() / tot[:, 1].sum()
            (logits, loss_mlm), join_mlm = self._fork_text(mlm_tail, uses=(t_mlm, labels), kind=kind)
            joins.append(join_mlm)
            Bm, Lm = data['text_mlm_ids'].shape
            ret.update({'cross_attn_mlm_logits': logits.reshape(Bm, Lm, -1)[..., :c.vocab]})
            loss_dict.update({'loss_mlm': loss_mlm})
            terms['mlm'] = (1.0, loss_mlm)

        def run_itm():                                                                           # :426-483
            rank, bsz, w_host, ev, all_video, all_text_ids, all_text_masks, ev_in = itm_pre
            drawn = itm_state if itm_state is not None else itm_draw()
            itm_labels, neg_log, vid_idx, vid_list, labels_dev, data_itm = (drawn[k] for k in ('itm_labels', 'neg_log', 'vid_idx', 'vid_list',
                                                                                               'labels_dev', 'data_itm'))
            if share_prefix:
                lo = rank * bsz
                remote = sorted({j for j in vid_list if not lo <= j < lo + bsz})
                v_rem = None
                if world > 1:
                    # Clips owned by other ranks: their owners already pushed them through the identical prefix (same pixels,
                    # replicated parameters, batch-independent kernels), so the prefix TOKENS are fetched from the owner and
                    # the token gradients return to it in backward (trainer/exchange.py) -- no pixel all-gather, no second
                    # prefix pass, and every rank runs the same graph every step (DDP static_graph).
                    from ..trainer.exchange import ExchangeClipsFn
                    table = drawn['req']()
                    assert table[rank] == remote
                    v_rem = ExchangeClipsFn.apply(v_pre, table, rank, bsz, c.seq)
                plan = [(0, j - lo) if lo <= j < lo + bsz else (1, remote.index(j)) for j in vid_list]
                data_itm['video'] = None
                pre = SelectClipsFn.apply(plan, c.seq, v_pre, v_rem)
                x32 = ops.stream32(v_pre)
                if x32 is not None and SW.on('EGV_ITM_RES32'):
                    # the fp32 value of the residual stream travels with the gathered clips (this rank's own clips: the fp32 rows of the
                    # shared prefix; clips fetched from another rank arrive as bf16 tokens and start from those), so that the ITM pass
                    # continues the stream the way the MLM pass does -- the reference under autocast keeps it in fp32 throughout
                    with torch.no_grad():
                        pre._res32 = _gather_clips(plan, c.seq, (x32, v_rem), torch.float32)
                data_itm['_video_prefix'] = pre
            else:
                data_itm['video'] = all_video.index_select(0, vid_idx)
            def itm_loss(itm_logits):
                ce_sum = ops.cross_entropy_sum(ops.CastFn.apply(itm_logits, torch.float32).contiguous(), labels_dev.contiguous(), 2, -100)
                tot = gather(torch.stack([ce_sum, torch.full_like(ce_sum, float(bsz))]).reshape(1, 2))
                return tot[:, 0].sum() / tot[:, 1].sum()
            r_itm = self.infer(data_itm, task_names='ITM', ret=ret)
            loss_itm = itm_loss(r_itm['cross_attn_itm_logits'])
            loss_dict.update({'loss_itm': loss_itm})
            terms['itm'] = (2.0, loss_itm)
            ret['_itm_labels'] = itm_labels
            ret['_itm_neg_log'] = neg_log

        # Which fused pass is CREATED last runs FIRST in backward (reverse creation order).  EGV_ITM_FIRST=1 creates the ITM pass before
        # the MLM pass (default: the reference's order).  The losses are added in the reference's order either way.
        terms = {}
        if 'MLM' in task_names:
            loss_dict['loss_mlm'] = None
        if 'ITM' in task_names:
            loss_dict['loss_itm'] = None
        mlm_late = []
        for name, fn in ((('ITM', run_itm), ('MLM', run_mlm)) if SW.on('EGV_ITM_FIRST') else (('MLM', run_mlm), ('ITM', run_itm))):
            if name in task_names:
                fn()
        for fn in mlm_late:
            fn()
        for key in ('mlm', 'itm'):
            if key in terms:
                loss_terms.append(terms[key])

        if 'EgoNCE' in task_names and late_tail and 'out' not in tail_state:
            make_tail()
        # everything the companion streams produced is ordered before the calling stream here; the total in the reference's order
        # of additions (EgoNCE + MLM, + 2 ITM: model.py:420,480)
        for j in joins:
            j()
        loss = None
        for wgt, term in loss_terms:
            term = term if wgt == 1.0 else wgt * term
            loss = term if loss is None else loss + term
        loss_dict.update({'loss_total': loss})
        return loss, loss_dict, ret

    # ------------------------------------------------------------------ checkpoint helpers
    def _inflate_positional_embeds(self, new_state_dict):
        """model.py:532-574: adapt temporal_embed when the checkpoint has a different number of frames."""
        curr_keys = list(self.state_dict().keys())
        k = 'video_model.temporal_embed'
        if k in new_state_dict and k in curr_keys:
            load = new_state_dict[k]
            load_f, curr_f, dim = load.shape[1], self.video_params['num_frames'], load.shape[2]
            if load_f != curr_f:
                if load_f > curr_f:
                    new = load[:, :curr_f, :]
                elif self.load_temporal_fix == 'zeros':
                    new = torch.zeros([load.shape[0], curr_f, dim])
                    new[:, :load_f] = load
                elif self.load_temporal_fix in ['interp', 'bilinear']:
                    mode = 'bilinear' if self.load_temporal_fix == 'bilinear' else 'nearest'
                    kw = dict(align_corners=True) if mode == 'bilinear' else {}
                    new = F.interpolate(load.unsqueeze(0), (curr_f, dim), mode=mode, **kw).squeeze(0)
                else:
                    raise NotImplementedError
                new_state_dict[k] = new
        k = 'video_model.pos_embed'
        if k in new_state_dict and k in curr_keys:
            if new_state_dict[k].shape[1] != self.state_dict()[k].shape[1]:
                raise NotImplementedError('Loading models with different spatial resolution / patch number not yet implemented, sorry.')
        return new_state_dict


def sim_matrix(a, b, eps=1e-8):
    """model.py:576-584 (fp32 HIP kernels: row normalisation + MFMA GEMM)."""
    return ops.sim_matrix_f32(a.float(), b.float(), eps)


def sim_matrix_batch_val(a, b, eps=1e-8):
    """model.py:587-595: batched cosine similarity (validation only)."""
    return torch.stack([sim_matrix(a[i], b[i], eps) for i in range(a.shape[0])])
