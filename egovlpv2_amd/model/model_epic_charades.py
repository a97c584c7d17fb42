"""Fine-tune variant of the model API (reference model/model_epic_charades.py): the same TimeSformer + RoBERTa towers on the
HIP block executor, 256-d projection heads (txt ReLU-Linear, vid Linear; :116-119) and the `Dual` task -- both towers, cosine
similarity of the gathered embeddings, a ranking loss (:410-444).  EPIC-Kitchens MIR pairs it with
AdaptiveMaxMarginRankingLoss weighted by the batch's `relation` field (configs/ft/epic.json:57-62), Charades-Ego with
NormSoftmaxLoss (configs/ft/charades.json:57-61); both are in model/loss.py here.  drop_path_rate must be 0 (it is in both
reference configs): the block executor has no stochastic depth."""
from __future__ import annotations

import torch

from .. import hipops as ops
from ..config import PathConfig
from .model import FrozenInTime as _PretrainModel, DEFAULT_YML, ROBERTA_BASE_DROPOUT, sim_matrix, sim_matrix_batch_val   # noqa: F401


class FrozenInTime(_PretrainModel):
    def __init__(self, video_params, text_params, projection_dim=4096, load_checkpoint=None, projection='minimal',
                 load_temporal_fix='bilinear', config=None, task_names='Dual', norm_layer=None, embed_dim=768,
                 compute_dtype=torch.bfloat16, path_config: PathConfig | None = None, init_seed: int = 0):
        """Same signature as the reference constructor (:45-56).  projection='minimal' means 256-d heads whatever
        projection_dim says (:116-119)."""
        if float(video_params.get('drop_path_rate', 0.0)) != 0.0:
            raise NotImplementedError("drop_path_rate != 0: stochastic depth is not implemented in the block executor")
        yml = dict(DEFAULT_YML, **(config or {}))
        if path_config is None:
            path_config = PathConfig(depth=yml['num_layers'], n_fuse=yml['num_fuse_block'], frames=video_params['num_frames'],
                                     dim=embed_dim, heads=yml['num_heads'], mlp_ratio=yml['mlp_ratio'], vocab=yml['vocab_size'],
                                     proj_dim=256, proj_style='linear', img=video_params.get('img_size', 224),
                                     drop_rate=ROBERTA_BASE_DROPOUT)
        if path_config.proj_style != 'linear':
            raise ValueError("the fine-tune variant uses proj_style='linear' heads")
        super().__init__(video_params, text_params, projection_dim=path_config.proj_dim, load_checkpoint=load_checkpoint,
                         projection=projection, load_temporal_fix=load_temporal_fix, config=config, task_names=task_names,
                         norm_layer=norm_layer, embed_dim=embed_dim, compute_dtype=compute_dtype, path_config=path_config,
                         init_seed=init_seed)

    def forward(self, data, allgather, n_gpu, args, config, loss_dual, gpu, return_embeds=True, task_names='Dual',
                dataset_name='charades'):
        """model_epic_charades.py:410-444.  Returns (loss, loss_dict, ret)."""
        ret, loss_dict = {}, {}
        loss = None
        self._begin_step()
        if 'Dual' in task_names:
            ret = self.infer(data, task_names='Dual')
            video_embeds = allgather(ops.CastFn.apply(ret['video_embeds'], torch.float32), n_gpu, args)
            text_embeds = allgather(ops.CastFn.apply(ret['text_embeds'], torch.float32), n_gpu, args)
            output = sim_matrix(text_embeds, video_embeds)
            if dataset_name == 'epic':
                w_embeds = allgather(data['relation'].float(), n_gpu, args)
                loss = loss_dual(output, w_embeds)
                ret.update({'sim_v2t': output, 'sim_t2v': output.t(), 'epic_relation': w_embeds})
            elif dataset_name == 'charades':
                loss, temp = loss_dual(output)
                ret.update({'sim_v2t': output, 'sim_t2v': output.t()})
            else:
                raise NameError(dataset_name)
            loss_dict.update({'Dual': loss})
        return loss, loss_dict, ret
