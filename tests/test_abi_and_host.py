"""CPU-side checks: the C-ABI library loads and exports every symbol include/egovlp_hip.h declares, the ctypes mirror of
egv_attn_desc has the C layout, the host model has the reference's state-dict surface, and the product path refuses to
run without a GPU (no silent fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(REPO, 'include', 'egovlp_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(egv_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from egovlpv2_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), s
        assert s in _lib.PROTOTYPES, f"{s} declared in the header but not bound in _lib.py"
    assert _lib.lib.egv_abi_version() == _lib.ABI_VERSION


def test_attn_desc_layout_matches_c(tmp_path):
    """compile a tiny C program against the public header and compare sizeof/offsetof with ctypes"""
    from egovlpv2_amd._lib import AttnDesc
    src = tmp_path / 't.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "egovlp_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                   'sizeof(egv_attn_desc), offsetof(egv_attn_desc, lse), offsetof(egv_attn_desc, q_bs), offsetof(egv_attn_desc, scale),'
                   'offsetof(egv_attn_desc, ws_bytes));return 0;}\n')
    exe = tmp_path / 't'
    subprocess.check_call(['gcc', '-I', os.path.join(REPO, 'include'), str(src), '-o', str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(AttnDesc), AttnDesc.lse.offset, AttnDesc.q_bs.offset, AttnDesc.scale.offset, AttnDesc.ws_bytes.offset]
    assert got == want


def test_block_desc_layouts_match_c(tmp_path):
    """the block descriptors (egv_vblock_desc / egv_tlayer_desc, ABI version 4: flags + MX-fp8 weight pointers, merged projections) and
    the grouped weight-gradient problem record: sizeof and the offsets of the fields added last, C against the ctypes mirrors"""
    from egovlpv2_amd._lib import VBlockDesc, TLayerDesc, WgradProblem
    src = tmp_path / 'b.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "egovlp_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(egv_vblock_desc), offsetof(egv_vblock_desc, flags), offsetof(egv_vblock_desc, wq), offsetof(egv_vblock_desc, wtq_s),'
                   'sizeof(egv_tlayer_desc), offsetof(egv_tlayer_desc, flags), offsetof(egv_tlayer_desc, w_qkv), offsetof(egv_tlayer_desc, b_ckv),'
                   'sizeof(egv_wgrad_problem));return 0;}\n')
    exe = tmp_path / 'b'
    subprocess.check_call(['gcc', '-I', os.path.join(REPO, 'include'), str(src), '-o', str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(VBlockDesc), VBlockDesc.flags.offset, VBlockDesc.wq.offset, VBlockDesc.wtq_s.offset,
            ctypes.sizeof(TLayerDesc), TLayerDesc.flags.offset, TLayerDesc.w_qkv.offset, TLayerDesc.b_ckv.offset, ctypes.sizeof(WgradProblem)]
    assert got == want


def test_state_dict_surface_matches_reference():
    from egovlpv2_amd.model.model import FrozenInTime
    from egovlpv2_amd.config import tiny_config
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'tiny.npz'))
    m = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': 4, 'pretrained': True},
                     {'model': 'roberta-base', 'pretrained': True, 'input': 'text'}, path_config=tiny_config(),
                     task_names='EgoNCE_MLM_ITM')
    assert sorted(k for k, _ in m.named_parameters()) == sorted(str(x) for x in g['param_names'])
    assert 'text_model.embeddings.position_ids' in m.state_dict()
    # reference zero / one initialisation (video_transformer.py:96-102,114; roberta.py:440; model.py:150)
    sd = m.state_dict()
    assert float(sd['video_model.blocks.1.attn.alpha_i2t']) == 0 and float(sd['text_model.encoder.layer.1.alpha_t2i']) == 0
    assert sd['video_model.blocks.0.timeattn.qkv.weight'].abs().sum() == 0
    assert (sd['video_model.blocks.0.timeattn.proj.weight'] == 1).all()
    assert sd['cls_token'].abs().sum() == 0


def test_full_size_parameter_count():
    """381.6 M parameters / 557 tensors for the reference architecture."""
    from egovlpv2_amd.synthetic import param_shapes
    from egovlpv2_amd.config import PathConfig
    shapes = {k: v for k, v in param_shapes(PathConfig(frames=4)).items() if not k.endswith('position_ids')}
    assert len(shapes) == 557
    assert abs(sum(int(np.prod(s)) for s in shapes.values()) / 1e6 - 381.6) < 0.1


def test_product_path_refuses_cpu():
    from egovlpv2_amd.model.model import FrozenInTime
    from egovlpv2_amd.config import tiny_config
    from egovlpv2_amd.synthetic import make_batch
    cfg = tiny_config()
    m = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': 4, 'pretrained': True},
                     {'model': 'roberta-base', 'pretrained': True, 'input': 'text'}, path_config=cfg)
    data, _, _ = make_batch(cfg, 2, 16)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.compute_video(data['video'])


def test_missing_library_fails_loudly(tmp_path):
    code = ("import sys, importlib.util as u\n"
            f"sys.path.insert(0, {REPO!r})\n"
            "import egovlpv2_amd._lib as L\n")
    # simulate a tree without the .so by pointing the loader at an empty directory
    env = dict(os.environ)
    pkg = tmp_path / 'egovlpv2_amd'
    pkg.mkdir()
    for f in ('__init__.py', '_lib.py', 'switches.py'):
        (pkg / f).write_text(open(os.path.join(REPO, 'egovlpv2_amd', f)).read())
    r = subprocess.run([sys.executable, '-c', "import egovlpv2_amd._lib"], cwd=str(tmp_path), env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'no CPU fallback' in r.stderr


def test_data_parallel_fix_and_inflate():
    from egovlpv2_amd.utils.util import state_dict_data_parallel_fix
    a = {'module.x': 1, 'module.y': 2}
    assert list(state_dict_data_parallel_fix(a, {'x': 0, 'y': 0})) == ['x', 'y']
    assert list(state_dict_data_parallel_fix({'x': 1}, {'module.x': 0})) == ['module.x']
    from egovlpv2_amd.model.model import FrozenInTime
    from egovlpv2_amd.config import tiny_config
    from egovlpv2_amd.synthetic import make_state_dict
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'tiny.npz'))
    cfg = tiny_config(frames=int(g['inflate_frames']))
    m = FrozenInTime({'model': 'SpaceTimeTransformer', 'num_frames': cfg.frames, 'pretrained': True},
                     {'model': 'roberta-base', 'pretrained': True, 'input': 'text'}, path_config=cfg)
    sd = make_state_dict(tiny_config(), 0)
    out = m._inflate_positional_embeds({'video_model.temporal_embed': sd['video_model.temporal_embed'].clone()})
    assert np.allclose(out['video_model.temporal_embed'][0, :, :8].numpy(), g['inflate_slice'], atol=1e-6)


def test_switch_tables_are_the_only_readers_of_the_environment_and_defaults_are_pinned():
    """Run-time switches: ONE table per side (csrc/egv_api.cpp: egv_config_dump(); egovlpv2_amd/switches.py), no other reader of the
    environment in the product path, and the defaults -- the configuration that is benchmarked and tested -- equal the committed
    tests/golden/switch_defaults.json (a changed default has to change that file too, i.e. it is a reviewed decision)."""
    import ctypes
    import glob
    import json
    import re
    from egovlpv2_amd import _lib, switches
    want = json.load(open(os.path.join(REPO, 'tests', 'golden', 'switch_defaults.json')))
    for f in glob.glob(os.path.join(REPO, 'egovlpv2_amd', 'csrc', '*.[hc]*')):
        if os.path.basename(f) != 'egv_api.cpp':
            assert 'getenv' not in open(f).read(), f
    for f in glob.glob(os.path.join(REPO, 'egovlpv2_amd', '**', '*.py'), recursive=True):
        if os.path.basename(f) != 'switches.py':
            assert not re.search(r'os\.environ|getenv', open(f).read()), f
    got = {}
    for line in _lib.lib.egv_config_dump().decode().strip().split('\n'):
        name, default, current, doc = line.split('\t')
        got[name] = float(default)
        assert doc.strip(), name
        if name not in os.environ:
            assert float(current) == float(default), name
    assert got == want['c']
    assert switches.defaults() == want['python']
    # every switch the C sources name is in the table with the default the call site states (the library aborts otherwise): the
    # names used in the sources and the table agree
    used = set()
    for f in glob.glob(os.path.join(REPO, 'egovlpv2_amd', 'csrc', '*.[hc]*')):
        used |= set(re.findall(r'egv_cfg_(?:on|int|f64)\("(EGV_[A-Z0-9_]+)"', open(f).read()))
    assert used == set(got), (used ^ set(got))
    used_py = set()
    for f in glob.glob(os.path.join(REPO, 'egovlpv2_amd', '**', '*.py'), recursive=True):
        used_py |= set(re.findall(r"SW\.(?:on|value)\('(EGV_[A-Z0-9_]+)'\)", open(f).read())) | set(re.findall(r"_sw\.value\('(EGV_[A-Z0-9_]+)'\)", open(f).read()))
    assert used_py == set(switches.SWITCHES), (used_py ^ set(switches.SWITCHES))


def test_dropin_registers_the_reference_module_names(tmp_path):
    """with egovlpv2_amd.dropin.install() the reference launcher's own import lines (multinode_train_egoclip.py:23-26)
    bind the HIP implementation -- `import model.model as module_arch`, `import model.loss as module_loss` -- while the rest of the
    reference tree (here a stand-in tree: a `model` package with another module, a trainer module, utils.util) keeps importing from
    the tree, with AllGather_multi / state_dict_data_parallel_fix patched in place.  Runs in a subprocess (sys.modules is global)."""
    import subprocess
    import sys
    import textwrap
    tree = tmp_path / 'reftree'
    (tree / 'model').mkdir(parents=True)
    (tree / 'trainer').mkdir()
    (tree / 'utils').mkdir()
    (tree / 'model' / '__init__.py').write_text('')
    (tree / 'model' / 'model.py').write_text('raise ImportError("the reference model.model must not be imported after install()")\n')
    (tree / 'model' / 'loss.py').write_text('raise ImportError("the reference model.loss must not be imported after install()")\n')
    (tree / 'model' / 'metric.py').write_text('MARK = "reference metric module"\n')
    (tree / 'trainer' / '__init__.py').write_text('from .trainer_egoclip import *\n')
    (tree / 'trainer' / 'trainer_egoclip.py').write_text('class AllGather_multi: pass\nclass Multi_Trainer_dist: pass\n')
    (tree / 'utils' / '__init__.py').write_text('')
    (tree / 'utils' / 'util.py').write_text('def state_dict_data_parallel_fix(a, b): raise RuntimeError("reference")\ndef other(): return 7\n')
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {str(tree)!r}); sys.path.insert(1, {REPO!r})
        import egovlpv2_amd.dropin
        egovlpv2_amd.dropin.install()
        import model.metric as module_metric
        import model.loss as module_loss
        import model.model as module_arch
        from trainer import Multi_Trainer_dist
        import trainer.trainer_egoclip as T
        import utils.util as U
        import egovlpv2_amd.model.model as ours, egovlpv2_amd.model.loss as ours_loss
        import egovlpv2_amd.trainer.trainer_egoclip as ours_T, egovlpv2_amd.utils.util as ours_U
        assert module_arch is ours and module_loss is ours_loss
        assert getattr(module_arch, 'FrozenInTime') is ours.FrozenInTime and module_loss.EgoNCE is ours_loss.EgoNCE
        assert module_metric.MARK == 'reference metric module'
        assert T.AllGather_multi is ours_T.AllGather_multi and T.Multi_Trainer_dist is Multi_Trainer_dist
        assert U.state_dict_data_parallel_fix is ours_U.state_dict_data_parallel_fix and U.other() == 7
        egovlpv2_amd.dropin.uninstall()
        assert 'model.model' not in sys.modules
        print('dropin ok')
    """)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'dropin ok' in r.stdout, r.stderr[-2000:]
