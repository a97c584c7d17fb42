"""Mirror of the reference's ``model`` package for the pre-training hot path
(``model.model`` = module_arch, ``model.loss`` = module_loss in multinode_train_egoclip.py:128,135)."""
