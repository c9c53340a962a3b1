"""Module registry (drop-in for /root/reference/lightning_modules/utils.py:1-27)."""
_LIGHTNING_MODULES = {}


def register_lightning_module(cls=None, *, name=None):
    def _register(cls):
        local_name = cls.__name__ if name is None else name
        if local_name in _LIGHTNING_MODULES:
            raise ValueError(f'Already registered model with name: {local_name}')
        _LIGHTNING_MODULES[local_name] = cls
        return cls

    return _register if cls is None else _register(cls)


def get_lightning_module_by_name(name):
    return _LIGHTNING_MODULES[name]


def create_lightning_module(config, checkpoint_path=None):
    module = get_lightning_module_by_name(config.training.lightning_module)(config)
    if checkpoint_path:
        module = module.load_from_checkpoint(checkpoint_path)
    return module
