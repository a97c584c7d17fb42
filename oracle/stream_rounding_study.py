"""Test infrastructure (build container or any CPU): how much of the bf16 mode's distance from fp32 comes from STORING the video
residual stream in bf16, and how much that distance scatters from one input to the next.

The oracle's video tower (oracle/ref_model.py, a restatement of video_transformer.py:214-228, :353-394) runs three ways on the same
clips: fp32; under torch.autocast(bf16) -- matmuls in bf16, LayerNorm / softmax / residual sums in fp32, what the reference's trainer does
(trainer_egoclip.py:143); and under autocast with the three residual sums of every block (and the patch tokens) rounded to bf16 --
the storage format of this build's bf16 mode.  Prints the two distances and their ratio per batch seed.
usage: python oracle/stream_rounding_study.py [case] [n_seeds]"""
import os, sys
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from oracle import ref_model as R                                    # noqa: E402
from helpers import load_golden                                      # noqa: E402
from egovlpv2_amd.synthetic import make_state_dict, make_batch      # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm()).item()


def rb(x):
    return x.bfloat16().float()


def block_rounded(x, i, sd, cfg, y=None, y_mask_add=None):
    p = f'video_model.blocks.{i}'
    t = R.divided_attention(R._ln(x, sd, p + '.norm3', cfg.eps_video), sd, p + '.timeattn', cfg, 'time')
    tr = rb(x + t)
    s = R.divided_attention(R._ln(tr, sd, p + '.norm1', cfg.eps_video), sd, p + '.attn', cfg, 'space')
    sr = rb(x + s)
    hdn = R._gelu(R._lin(R._ln(sr, sd, p + '.norm2', cfg.eps_video), sd, p + '.mlp.fc1'))
    return rb(sr + R._lin(hdn, sd, p + '.mlp.fc2'))


def features(sd, video, cfg, rounded):
    x = R.patch_tokens(sd, video, cfg, 'video_model.cls_token')
    if rounded:
        x = rb(x)
    for i in range(cfg.depth):
        x = block_rounded(x, i, sd, cfg) if rounded else R.video_block(x, i, sd, cfg)
    return R._ln(x, sd, 'video_model.norm', cfg.eps_video)[:, 0]


def main(name, nseed):
    g, cfg, B, L, wseed, bseed = load_golden(name)
    sd = {k: v.float() for k, v in make_state_dict(cfg, wseed).items()}
    R._gelu = F.gelu                                                  # nn.GELU's own kernel (one rounding of its bf16 result under autocast)
    rat = []
    for s in range(nseed):
        data, _, _ = make_batch(cfg, B, L, bseed + 1000 * s)
        v = data['video'].float()
        with torch.no_grad():
            f0 = R._proj(features(sd, v, cfg, False), sd, 'vid_proj', cfg)
            with torch.autocast('cpu', dtype=torch.bfloat16):
                f1 = R._proj(features(sd, v, cfg, False), sd, 'vid_proj', cfg).float()
                f2 = R._proj(features(sd, v, cfg, True), sd, 'vid_proj', cfg).float()
        e1, e2 = rel(f1, f0), rel(f2, f0)
        rat.append(e2 / e1)
        print(f"{name} seed {s}: autocast {e1:.4e}   autocast + bf16 stream {e2:.4e}   ratio {e2 / e1:.3f}", flush=True)
    t = torch.tensor(rat)
    print(f"ratio mean {t.mean():.3f}  std {t.std():.3f}  min {t.min():.3f}  max {t.max():.3f}")


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'base_f4', int(sys.argv[2]) if len(sys.argv) > 2 else 6)
