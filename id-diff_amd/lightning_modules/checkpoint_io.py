"""Reading the authors' artefacts without their packages: Lightning ``.ckpt`` files and pickled configs.

A Lightning checkpoint of the reference is ``torch.save({'state_dict': {'score_model.<k>': Tensor, ...},
'hyper_parameters': {'config': ml_collections.ConfigDict}, 'callbacks': ..., 'optimizer_states': ...})``
(``save_hyperparameters()``, /root/reference/lightning_modules/BaseSdeGenerativeModel.py:17); ``main.py --config x.pkl``
unpickles a bare ``ml_collections.ConfigDict`` (/root/reference/main.py:32-34).  Neither ``ml_collections`` nor
``pytorch_lightning`` is installed on the MI355X image, and ``torch.load(weights_only=True)`` rejects their globals.

The unpickler below never imports anything the file names: tensor-rebuild helpers and a short list of builtin
containers are allowed, EVERY other global resolves to an inert stand-in that only records its constructor arguments
and state.  Nothing a checkpoint contains is executed, and a missing third-party package cannot fail the load.
``to_config`` then turns the stand-ins that look like ``ConfigDict`` / ``FieldReference`` into the local ``ConfigDict``.
"""
import collections
import pickle

import torch

from ..configs.config_dict import ConfigDict


class Foreign:
    """Inert stand-in for a class this build does not ship (ml_collections.ConfigDict, Lightning helpers, ...)."""
    _qualname = "?"

    def __new__(cls, *args, **kwargs):
        obj = object.__new__(cls)
        obj.__dict__["args"], obj.__dict__["kwargs"], obj.__dict__["state"], obj.__dict__["items"] = args, kwargs, None, []
        return obj

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        self.__dict__["state"] = state

    def __setitem__(self, key, value):          # dict-like classes pickled through SETITEMS
        self.__dict__["items"].append((key, value))

    def append(self, value):                    # list-like classes pickled through APPENDS
        self.__dict__["items"].append(value)

    def extend(self, values):
        self.__dict__["items"].extend(values)

    def __repr__(self):
        return f"<foreign {self._qualname}>"


_ALLOWED = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("collections", "defaultdict"): collections.defaultdict,
    ("builtins", "dict"): dict, ("builtins", "list"): list, ("builtins", "tuple"): tuple, ("builtins", "set"): set,
    ("builtins", "frozenset"): frozenset, ("builtins", "int"): int, ("builtins", "float"): float,
    ("builtins", "bool"): bool, ("builtins", "str"): str, ("builtins", "bytes"): bytes, ("builtins", "complex"): complex,
    ("builtins", "slice"): slice, ("builtins", "range"): range,
}
# What torch.save emits for tensors / parameters / dtypes / sizes -- an EXACT (module, name) list, not a module prefix:
# `torch.storage._load_from_bytes` (a nested torch.load without restrictions), `torch._utils._import_dotted_name` and
# `torch.serialization.load` live in the same modules as the rebuild helpers and must stay stand-ins.
_TORCH_ALLOWED = {
    ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"), ("torch.nn.parameter", "Parameter"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
}
_TORCH_STORAGES = {"UntypedStorage", "TypedStorage", "DoubleStorage", "FloatStorage", "HalfStorage", "BFloat16Storage",
                   "LongStorage", "IntStorage", "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage",
                   "ComplexDoubleStorage", "ComplexFloatStorage"}


def _is_torch_global(module, name):
    if (module, name) in _TORCH_ALLOWED:
        return True
    if module == "torch" and (name in _TORCH_STORAGES or isinstance(getattr(torch, name, None), torch.dtype)):
        return True
    return False


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return _ALLOWED[(module, name)]
        if _is_torch_global(module, name):
            return super().find_class(module, name)
        if module == "numpy.core.multiarray" or module == "numpy._core.multiarray" or module == "numpy":
            if name in ("_reconstruct", "ndarray", "dtype", "scalar"):
                return super().find_class(module, name)
        return type(name, (Foreign,), {"_qualname": f"{module}.{name}"})


class _PickleModule:
    """What ``torch.load(pickle_module=...)`` expects: ``Unpickler`` + ``load``."""
    __name__ = "idiff_restricted_pickle"
    Unpickler = _Unpickler

    @staticmethod
    def load(f, **kwargs):
        return _Unpickler(f, **kwargs).load()


def to_config(obj):
    """Stand-ins that carry ml_collections' layout (``_fields`` of a ConfigDict, ``_value`` of a FieldReference) and
    plain containers -> local ``ConfigDict`` / builtins.  Anything else foreign is dropped to ``None``."""
    if isinstance(obj, Foreign):
        state = obj.state if isinstance(obj.state, dict) else {}
        if "_fields" in state:                                   # ml_collections.ConfigDict / FrozenConfigDict
            return ConfigDict({k: to_config(v) for k, v in state["_fields"].items()})
        if "_value" in state:                                    # ml_collections FieldReference
            return to_config(state["_value"])
        if obj.items and all(isinstance(i, tuple) and len(i) == 2 for i in obj.items):   # dict subclass (AttributeDict)
            return ConfigDict({k: to_config(v) for k, v in obj.items})
        if state:
            return ConfigDict({k: to_config(v) for k, v in state.items() if isinstance(k, str)})
        return None
    if isinstance(obj, dict):
        return ConfigDict({k: to_config(v) for k, v in obj.items()}) if all(isinstance(k, str) for k in obj) else \
            {k: to_config(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_config(v) for v in obj)
    return obj


def load_checkpoint(path):
    """The checkpoint dict, tensors on the host.  Plain ``weights_only`` loading is tried first (state-dict-only
    files); a file that names foreign classes goes through the inert-stand-in unpickler.  I/O errors and corrupt
    files propagate as they are."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleModule)


def score_model_state_dict(ckpt):
    """``state_dict`` entries of the score network with the ``score_model.`` prefix removed (dim_reduction.py:127-129
    evaluates the raw, non-EMA weights: the EMA swap is commented out at :131-133)."""
    state = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
    own = {k[len("score_model."):]: v for k, v in state.items() if k.startswith("score_model.")}
    return own if own else dict(state)


def load_config_pickle(path):
    """``main.py --config cfg.pkl`` (/root/reference/main.py:32-34) without ml_collections."""
    with open(path, "rb") as f:
        return to_config(_Unpickler(f).load())
