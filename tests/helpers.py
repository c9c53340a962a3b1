"""Shared helpers for the test-suite (config builders mirroring tests/golden/make_golden.py)."""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import id_diff_amd  # noqa: E402
from id_diff_amd.configs.config_dict import ConfigDict  # noqa: E402


def fcn_config(hidden_nodes=2048, hidden_layers=5, state_size=100):
    c = ConfigDict()
    c.model = ConfigDict(name="fcn", state_size=state_size, hidden_layers=hidden_layers,
                         hidden_nodes=hidden_nodes, dropout=0.0, sigma_min=1e-2, sigma_max=4,
                         num_scales=1000)
    c.training = ConfigDict(sde="vesde", continuous=True, batch_size=500)
    c.data = ConfigDict(shape=[state_size])
    return c


def ncsnpp_config(**over):
    c = ConfigDict()
    c.data = ConfigDict(image_size=32, effective_image_size=32, num_channels=3, centered=False,
                        shape=[3, 32, 32])
    c.training = ConfigDict(continuous=True, sde="vesde", batch_size=128)
    c.model = ConfigDict(
        name="ncsnpp", nf=8, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,),
        dropout=0.1, resamp_with_conv=True, conditional=True, fir=True, fir_kernel=[1, 3, 3, 1],
        skip_rescale=True, resblock_type="biggan", progressive="none", progressive_input="residual",
        progressive_combine="sum", embedding_type="fourier", init_scale=0., fourier_scale=16,
        nonlinearity="swish", normalization="GroupNorm", sigma_min=0.01, sigma_max=50, num_scales=1000,
        scale_by_sigma=True, conv_size=3)
    for k, v in over.items():
        c[k] = v
    return c


def overrides_from_golden(z):
    keys = [str(k) for k in z["override_keys"]]
    vals = [ast.literal_eval(str(v)) for v in z["override_vals"]]
    return dict(zip(keys, vals))


def state_dict_from_golden(z, prefix="sd::"):
    return {k[len(prefix):]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith(prefix)}


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def beatgans_config(**over):
    c = ConfigDict()
    c.data = ConfigDict(image_size=16, effective_image_size=16, num_channels=3, centered=False, shape=[3, 16, 16])
    c.training = ConfigDict(continuous=True, sde="vesde", batch_size=128)
    c.model = ConfigDict(
        name="BeatGANsUNetModel", sigma_min=0.01, sigma_max=50, num_scales=1000, image_size=16, in_channels=3,
        model_channels=32, out_channels=3, num_res_blocks=1, num_input_res_blocks=None, embed_channels=16,
        attention_resolutions=(8,), time_embed_channels=None, dropout=0.1, channel_mult=(1, 1, 2),
        input_channel_mult=None, conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, num_heads=1,
        num_head_channels=-1, num_heads_upsample=-1, resblock_updown=True, use_new_attention_order=False,
        resnet_two_cond=False, resnet_cond_channels=None, resnet_use_zero_module=True, attn_checkpoint=False)
    for k, v in over.items():
        c[k] = v
    return c


def ddpm_config(**over):
    c = ConfigDict()
    c.data = ConfigDict(image_size=16, effective_image_size=16, num_channels=1, centered=False, shape=[1, 16, 16])
    c.training = ConfigDict(continuous=True, sde="vesde", batch_size=128)
    c.model = ConfigDict(name="ddpm", nf=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), dropout=0.1,
                         resamp_with_conv=True, conditional=True, nonlinearity="swish", normalization="GroupNorm",
                         input_channels=1, output_channels=1, sigma_min=0.009, sigma_max=50, num_scales=1000,
                         scale_by_sigma=True, ema_rate=0.999)
    for k, v in over.items():
        c[k] = v
    return c


def fill_from_seed(module, seed, scale=0.05):
    """The weight recipe of tests/golden/make_golden.py::fill_from_seed (wide fixtures store no weights)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in module.state_dict().items():
            if not prm.dtype.is_floating_point:
                continue
            fan = prm[0].numel() if prm.ndim > 1 else 1
            draw = torch.randn(prm.shape, generator=g) * (1.0 / max(fan, 1) ** 0.5 if prm.ndim > 1 else scale)
            if prm.ndim == 1 and name.endswith("weight"):
                draw = 1.0 + draw
            prm.copy_(draw)


def weight_abs_sums(module):
    return np.array([float(v.double().abs().sum()) for v in module.state_dict().values() if v.dtype.is_floating_point])


def replay_conditional_noise(seed, n_levels, n_points, num_batches, batch_shape, rows, keep_levels):
    """The draws the reference's conditional loop consumed (global CPU generator seeded with ``seed``; order level,
    point, batch -- dim_reduction.py:39-79) as {(level, point): [rows, *sample_shape]} for the levels in
    ``keep_levels``; the other levels' draws are generated and dropped to keep the stream aligned."""
    g = torch.Generator().manual_seed(int(seed))
    out = {}
    for level in range(n_levels):
        for point in range(n_points):
            z = torch.randn(num_batches, *batch_shape, generator=g)
            if level in keep_levels:
                out[(level, point)] = z.reshape(num_batches * batch_shape[0], *batch_shape[1:])[:rows].clone()
    return out


def fake_ml_collections():
    """Classes pickling like ml_collections 0.1.0's (instance __dict__ with `_fields`, FieldReference with `_value`)
    and like Lightning's AttributeDict (a dict subclass), under the module names a real checkpoint carries."""
    import types
    mods = {}
    for name in ("ml_collections", "ml_collections.config_dict", "ml_collections.config_dict.config_dict",
                 "pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing"):
        mods[name] = types.ModuleType(name)
    cd = mods["ml_collections.config_dict.config_dict"]

    class FieldReference:
        def __init__(self, value):
            self._value, self._field_type, self._ops, self._required = value, type(value), [], False

    class MLConfigDict:
        def __init__(self, **fields):
            self.__dict__["_fields"] = fields
            self.__dict__["_locked"] = False
            self.__dict__["_type_safe"] = True
            self.__dict__["_convert_dict"] = True

    class AttributeDict(dict):
        pass

    for cls, mod in ((FieldReference, cd), (MLConfigDict, cd), (AttributeDict, mods["pytorch_lightning.utilities.parsing"])):
        cls.__module__ = mod.__name__
        cls.__qualname__ = cls.__name__ = {"MLConfigDict": "ConfigDict"}.get(cls.__name__, cls.__name__)
        setattr(mod, cls.__name__, cls)
    return mods, MLConfigDict, FieldReference, AttributeDict


def to_foreign_config(cfg, MLConfigDict, FieldReference=None):
    """The local ConfigDict as the ml_collections-shaped stand-in classes (nested), for writing artefacts the way the
    authors' stack does; with ``FieldReference`` every float leaf is wrapped in one (ml_collections does that for
    placeholders) so the reader's unwrapping is exercised."""
    out = {}
    for k, v in cfg.items():
        if isinstance(v, ConfigDict):
            out[k] = to_foreign_config(v, MLConfigDict, FieldReference)
        elif FieldReference is not None and isinstance(v, float):
            out[k] = FieldReference(v)
        else:
            out[k] = v
    return MLConfigDict(**out)


def write_lightning_artifacts(ckpt_path, score_model_state, cfg, cfg_pkl_path=None):
    """A checkpoint shaped the way Lightning 1.5 writes the reference's (`state_dict` with `score_model.*` keys,
    `hyper_parameters` = AttributeDict(config=ml_collections.ConfigDict), optimizer / callback entries) and, optionally,
    the pickled bare config that `main.py --config x.pkl` reads (/root/reference/main.py:32-34).  The foreign module names
    exist only while the files are written."""
    import pickle
    mods, MLConfigDict, FieldReference, AttributeDict = fake_ml_collections()
    foreign = to_foreign_config(cfg, MLConfigDict, FieldReference)
    state = {'score_model.' + k: v.detach().cpu().clone() for k, v in score_model_state.items()}
    ckpt = {'epoch': 4, 'global_step': 1000, 'pytorch-lightning_version': '1.5.1', 'state_dict': state,
            'hyper_parameters': AttributeDict(config=foreign), 'callbacks': {}, 'lr_schedulers': [],
            'optimizer_states': [{'state': {0: {'step': 1000, 'exp_avg': torch.zeros(3)}}, 'param_groups': [{'lr': 2e-4}]}]}
    saved = {k: sys.modules.get(k) for k in mods}
    try:
        sys.modules.update(mods)
        os.makedirs(os.path.dirname(os.path.abspath(ckpt_path)), exist_ok=True)
        torch.save(ckpt, ckpt_path)
        if cfg_pkl_path is not None:
            with open(cfg_pkl_path, 'wb') as f:
                pickle.dump(foreign, f)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ckpt_path
