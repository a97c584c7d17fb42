"""Shape/config record for the EgoVLPv2 pre-training hot path.

Mirrors the values the reference spreads over ``EgoNCE_MLM_ITM_Config.yml`` (hidden_size, num_heads,
num_layers, mlp_ratio, num_fuse_block, vocab_size, drop_rate), the hard-coded ViT-B/16 choice in
``model/model.py:73-83`` and ``configs/pt/egoclip.json`` (num_frames, projection_dim).  The reference
hard-codes depth 12 / 6 fused blocks (video_transformer.py:302, roberta.py:438); here they are fields
so that BASELINE.json's tiny config and the ViT-L extension are constructible.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict


@dataclass
class PathConfig:
    depth: int = 12            # num_layers (yml) == TimeSformer depth
    n_fuse: int = 6            # num_fuse_block (yml)
    img: int = 224
    patch: int = 16
    frames: int = 16           # video_params.num_frames
    dim: int = 768             # hidden_size / embed_dim
    heads: int = 12
    mlp_ratio: int = 4
    vocab: int = 50265
    max_pos: int = 514
    proj_dim: int = 4096
    proj_style: str = 'mlp'    # 'mlp': Linear-ReLU-Linear-ReLU-Linear heads of the pre-training model (model.py:105-115);
                               # 'linear': txt ReLU-Linear / vid Linear of the fine-tune variant (model_epic_charades.py:116-119)
    pad_id: int = 1
    eps_video: float = 1e-5    # nn.LayerNorm default wins over the eps=1e-6 partial (video_transformer.py:250,279)
    eps_text: float = 1e-5     # roberta-base layer_norm_eps
    eps_model_norm: float = 1e-6   # model.py:154-155
    eps_mlm: float = 1e-12     # BertPredictionHeadTransform with default RobertaConfig (heads.py:41)
    drop_rate: float = 0.0     # RoBERTa hidden / attention-probability dropout in train mode.  In the reference the text tower is
                               # RobertaModel.from_pretrained('roberta-base') (model.py:68), i.e. the pretrained config's fixed 0.1
                               # -- the yml drop_rate only reaches bert_config, which feeds the dropout-free MLMHead
                               # (model.py:127-137).  FrozenInTime(...) therefore sets 0.1 whatever the yml says; 0 here so that
                               # parity cases built from a PathConfig are deterministic.

    @property
    def n_patches(self) -> int:
        return (self.img // self.patch) ** 2

    @property
    def seq(self) -> int:
        return 1 + self.frames * self.n_patches

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads

    def as_dict(self):
        d = asdict(self)
        d.pop('drop_rate')
        return d


def tiny_config(**kw) -> PathConfig:
    """BASELINE.json configs[0]: 2+2 layers, 1 fused, 4 x 112^2 frames."""
    base = dict(depth=2, n_fuse=1, img=112, frames=4)
    base.update(kw)
    return PathConfig(**base)


def base_config(**kw) -> PathConfig:
    """BASELINE.json configs[1]/[2]: ViT-B/16 + RoBERTa-base, 16 x 224^2 frames."""
    return PathConfig(**kw)
