"""The slice of ``BaseSdeGenerativeModel`` the hot path touches (lightning_modules/BaseSdeGenerativeModel.py:13-47):
it owns ``score_model``, builds the SDE and ``sampling_eps``, and restores weights from a Lightning checkpoint
(``state_dict`` keys ``score_model.*``, dim_reduction.py:127-129).  Training hooks are out of scope, so this is a
plain ``torch.nn.Module`` -- pytorch_lightning is not needed to evaluate a trained model.
"""
import torch
import torch.nn as nn

from .. import sde_lib
from ..models import utils as mutils
from . import utils


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

        ``None`` keeps the freshly initialised weights (benchmarks / tests: no trained checkpoint ships with the
        reference).  Unlike Lightning's classmethod this restores into the existing instance; the call site
        ``pl_module = pl_module.load_from_checkpoint(path)`` (dim_reduction.py:128) reads the same either way.
        """
        if checkpoint_path is None:
            return self
        try:
            ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        except Exception:
            # Lightning checkpoints pickle the ConfigDict under 'hyper_parameters'; only 'state_dict' is needed.
            ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        state = ckpt.get("state_dict", ckpt)
        own = {k[len("score_model."):]: v for k, v in state.items() if k.startswith("score_model.")}
        self.score_model.load_state_dict(own if own else state, strict=True)
        return self
