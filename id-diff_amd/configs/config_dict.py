"""Minimal stand-in for ``ml_collections.ConfigDict`` (not installable here).

The reference builds its configs from ``ml_collections.ConfigDict`` objects
(/root/reference/configs/default.py:5-7) and the hot path reads them with
plain attribute access, ``config.data.get(name, default)``
(lightning_data_modules/KSphereDataset.py:11-18) and -- the quirk that matters --
a *dotted* ``hasattr(config, 'dim_estimation.num_datapoints')``
(dim_reduction.py:144-147), which only works because ConfigDict resolves
dotted keys recursively.  This class reproduces exactly that surface.
"""


class ConfigDict(dict):
    """Attribute dict with recursive dotted-key lookup."""

    def __init__(self, initial=None, **kwargs):
        super().__init__()
        if initial:
            for k, v in dict(initial).items():
                self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    # -- item access -------------------------------------------------------
    def __setitem__(self, key, value):
        if isinstance(key, str) and "." in key:
            head, rest = key.split(".", 1)
            if head not in self:
                dict.__setitem__(self, head, ConfigDict())
            self[head][rest] = value
            return
        if isinstance(value, dict) and not isinstance(value, ConfigDict):
            value = ConfigDict(value)
        dict.__setitem__(self, key, value)

    def __getitem__(self, key):
        if isinstance(key, str) and "." in key:
            head, rest = key.split(".", 1)
            return dict.__getitem__(self, head)[rest]
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except (KeyError, TypeError):
            return False

    def get(self, key, default=None):
        try:
            return self[key]
        except (KeyError, TypeError):
            return default

    # -- attribute access --------------------------------------------------
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        try:
            return self[name]
        except (KeyError, TypeError):
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name) from None

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}

    def copy_and_resolve_references(self):
        return ConfigDict(self.to_dict())
