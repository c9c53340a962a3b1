"""The slice of ``BaseSdeGenerativeModel`` the hot path touches (lightning_modules/BaseSdeGenerativeModel.py:13-47):
it owns ``score_model``, builds the SDE and ``sampling_eps``, and restores weights from a Lightning checkpoint
(``state_dict`` keys ``score_model.*``, dim_reduction.py:127-129).  Training hooks are out of scope, so this is a
plain ``torch.nn.Module`` -- pytorch_lightning is not needed to evaluate a trained model.
"""
import torch
import torch.nn as nn

from .. import sde_lib
from ..models import utils as mutils
from . import checkpoint_io, utils


@utils.register_lightning_module(name='base')
class BaseSdeGenerativeModel(nn.Module):
    def __init__(self, config, *args, **kwargs):
        super().__init__()
        self.config = config
        self.score_model = mutils.create_model(config)
        self.data_shape = config.data.shape
        self.default_sampling_shape = [config.training.batch_size] + list(self.data_shape)

    def configure_sde(self, config):
        self.sde, self.sampling_eps = sde_lib.configure_sde(config)

    def load_from_checkpoint(self, checkpoint_path, **kwargs):
        """Lightning ``.ckpt`` = torch.save({'state_dict': {'score_model.<k>': tensor}, 'hyper_parameters': ...}).

        Unlike Lightning's classmethod this restores into the existing instance; the call site
        ``pl_module = pl_module.load_from_checkpoint(path)`` (dim_reduction.py:128) reads the same either way.
        ``None`` is an error, as in the reference (Lightning raises on it), unless the score model has no weights to
        restore (the analytic ``ksphere_exact``) or ``config.model.allow_random_init`` opts in (benchmarks / tests:
        no trained checkpoint ships with the reference).
        """
        if checkpoint_path is None:
            has_weights = any(p.requires_grad for p in self.score_model.parameters())   # trained weights
            if has_weights and not bool(self.config.model.get('allow_random_init', False)):
                raise ValueError(
                    "config.model.checkpoint_path is None: the ID estimate of an untrained score network is "
                    "meaningless.  Pass --checkpoint_path, or set config.model.allow_random_init = True "
                    "(--allow_random_init) to time / test the pipeline on random weights.")
            return self
        ckpt = checkpoint_io.load_checkpoint(checkpoint_path)
        self.score_model.load_state_dict(checkpoint_io.score_model_state_dict(ckpt), strict=True)
        return self
