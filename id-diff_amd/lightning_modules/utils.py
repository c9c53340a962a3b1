"""Evaluation-module registry with the reference's function names (lightning_modules/utils.py:1-27)."""
from ..registry import Registry

_MODULES = Registry("lightning module")
register_lightning_module = _MODULES.register
get_lightning_module_by_name = _MODULES.get


def create_lightning_module(config, checkpoint_path=None):
    """Instantiate ``config.training.lightning_module`` and optionally restore a checkpoint into it."""
    module = get_lightning_module_by_name(config.training.lightning_module)(config)
    return module.load_from_checkpoint(checkpoint_path) if checkpoint_path else module
