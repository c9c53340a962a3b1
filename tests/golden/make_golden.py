"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Run once, in the build container only (``/root/reference`` does not exist on
the GPU box)::

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (GBATZOLIS/ID-diff) is imported read-only from /root/reference.
Third-party packages it imports at module level but that are not installed in
this image (pytorch_lightning, torchvision, cv2) get in-memory stand-ins that
only provide the base classes / names touched at import time; the reference's
JIT build of its CUDA ops is disabled (CPU tensors never reach it,
op/upfirdn2d.py:146).  Nothing from the reference is copied: the .npz files
hold arrays (inputs, weights under the reference's state_dict keys, outputs)
plus the plain config values that produced them.

The reference ships no tests or golden files of its own, so
these vectors -- outputs of the reference run here, torch 2.10 CPU fp32 -- are
what pins the oracle.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _install_standins():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

        @property
        def dtype(self):  # Lightning's DeviceDtypeModuleMixin; BeatGANsUNET.py:257 reads it
            return next(self.parameters()).dtype

    pl.LightningModule = LightningModule
    pl.LightningDataModule = object
    sys.modules["pytorch_lightning"] = pl

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    class InterpolationMode:
        NEAREST = "nearest"
        BILINEAR = "bilinear"

    class Resize:
        def __init__(self, *a, **k):
            pass

    tvf.InterpolationMode = InterpolationMode
    tvt.Resize = Resize
    tvt.functional = tvf
    tvt.ToTensor = lambda: (lambda img: img)
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.transforms.functional"] = tvf
    for missing in ("cv2",):
        try:
            __import__(missing)
        except Exception:
            sys.modules[missing] = types.ModuleType(missing)

    import torch.utils.cpp_extension as ce
    ce.load = lambda *a, **k: None

    # numpy>=2: np.linalg.qr(Tensor) hands back a Tensor and torch.from_numpy
    # rejects it (lightning_data_modules/KSphereDataset.py:42-43).
    _orig = torch.from_numpy
    torch.from_numpy = lambda a: a if isinstance(a, torch.Tensor) else _orig(a)

    import matplotlib
    matplotlib.use("Agg")


_install_standins()
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REPO, "id-diff_amd", "configs"))
from config_dict import ConfigDict  # noqa: E402  (plain attribute dict, ours)

import sde_lib  # noqa: E402  (reference)
from models import utils as mutils  # noqa: E402
from models import fcn as ref_fcn  # noqa: E402,F401
from models import ncsnpp as ref_ncsnpp  # noqa: E402,F401
import op as ref_op  # noqa: E402
from op.upfirdn2d import upfirdn2d_native  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def sd_arrays(module, prefix="sd::"):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


# --------------------------------------------------------------------------
def gen_upfirdn2d():
    g = torch.Generator().manual_seed(1234)
    cases = []
    fir = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32)
    fir /= fir.sum()
    asym4 = torch.randn(4, 4, generator=g).numpy()
    asym3 = torch.randn(3, 3, generator=g).numpy()
    asym23 = torch.randn(2, 3, generator=g).numpy()
    one = np.array([[1.5]], dtype=np.float32)
    # (N, C, H, W, kernel, up, down, pad0, pad1)
    spec = [
        (2, 3, 8, 8, fir, 1, 2, 1, 1),          # ncsnpp downsample_2d family
        (2, 3, 8, 8, fir, 1, 1, 2, 2),          # FIR before stride-2 conv (conv_downsample_2d)
        (2, 3, 4, 4, fir * 4, 2, 1, 2, 1),      # ncsnpp upsample_2d family
        (1, 2, 7, 5, asym4, 1, 1, 0, 0),        # valid conv, asymmetric kernel -> exposes the flip
        (1, 2, 7, 5, asym3, 2, 1, 1, 1),
        (1, 2, 9, 6, asym3, 1, 2, 0, 1),
        (2, 2, 5, 7, asym4, 2, 2, 2, 1),
        (1, 3, 6, 6, asym4, 3, 2, 1, 2),
        (1, 2, 8, 9, asym3, 1, 1, -1, -2),      # negative pads crop
        (1, 2, 8, 9, asym4, 2, 1, -1, 3),
        (1, 1, 5, 5, asym23, 1, 1, 1, 1),       # non-square kernel
        (3, 1, 4, 4, one, 1, 1, 0, 0),          # 1x1 kernel = scaling
        (1, 1, 1, 1, fir, 2, 1, 2, 1),          # single pixel
        (1, 4, 16, 16, fir, 1, 2, 1, 1),
        (1, 4, 33, 17, fir, 1, 2, 1, 1),        # odd sizes
        (1, 4, 16, 16, fir * 4, 2, 1, 2, 1),
    ]
    out = {}
    for i, (n, c, h, w, k, up, down, p0, p1) in enumerate(spec):
        x = torch.randn(n, c, h, w, generator=g)
        kt = torch.from_numpy(np.ascontiguousarray(k, dtype=np.float32))
        y = ref_op.upfirdn2d(x, kt, up=up, down=down, pad=(p0, p1))
        y2 = upfirdn2d_native(x, kt, up, up, down, down, p0, p1, p0, p1)
        assert torch.equal(y, y2)
        out[f"c{i}::x"] = x.numpy()
        out[f"c{i}::k"] = kt.numpy()
        out[f"c{i}::params"] = np.array([up, down, p0, p1], dtype=np.int64)
        out[f"c{i}::y"] = y.numpy()
    # distinct x / y factors and pads through the native entry point
    x = torch.randn(2, 2, 6, 7, generator=g)
    kt = torch.from_numpy(asym23.copy())
    y = upfirdn2d_native(x, kt, 2, 1, 1, 2, 1, 0, 0, 2)
    out["xy::x"] = x.numpy()
    out["xy::k"] = kt.numpy()
    out["xy::params"] = np.array([2, 1, 1, 2, 1, 0, 0, 2], dtype=np.int64)  # up_x up_y down_x down_y px0 px1 py0 py1
    out["xy::y"] = y.numpy()
    out["n_cases"] = np.array(len(spec))
    save("upfirdn2d.npz", **out)


def gen_fused_act():
    g = torch.Generator().manual_seed(77)
    out = {}
    shapes = [(2, 5, 7, 3), (4, 5), (3, 8, 4), (1, 1, 1, 1)]
    for i, shp in enumerate(shapes):
        x = torch.randn(*shp, generator=g)
        b = torch.randn(shp[1], generator=g)
        y_default = ref_op.fused_leaky_relu(x, b)
        # CPU branch ignores negative_slope (op/fused_act.py:87-94) but honours scale
        y_args = ref_op.fused_leaky_relu(x, b, negative_slope=0.05, scale=1.25)
        out[f"c{i}::x"] = x.numpy()
        out[f"c{i}::b"] = b.numpy()
        out[f"c{i}::y_default"] = y_default.numpy()
        out[f"c{i}::y_slope0.05_scale1.25"] = y_args.numpy()
    m = ref_op.FusedLeakyReLU(5)
    out["module_bias_init"] = m.bias.detach().numpy()
    out["n_cases"] = np.array(len(shapes))
    save("fused_act.npz", **out)


def gen_sde():
    t = torch.tensor([1e-5, 1e-3, 0.1, 0.3, 0.5, 1.0], dtype=torch.float32)
    x = torch.arange(12, dtype=torch.float32).reshape(6, 2)
    out = {"t": t.numpy(), "x": x.numpy()}
    for name, (smin, smax) in {"ve_ksphere": (1e-2, 4), "ve_image": (0.01, 50), "ve_mnist": (0.009, 50)}.items():
        s = sde_lib.VESDE(sigma_min=smin, sigma_max=smax, N=1000)
        mean, std = s.marginal_prob(x, t)
        out[f"{name}::mean"] = mean.numpy()
        out[f"{name}::std"] = std.numpy()
        out[f"{name}::params"] = np.array([smin, smax, 1000], dtype=np.float64)
    s = sde_lib.VPSDE(beta_min=0.1, beta_max=20., N=1000)
    mean, std = s.marginal_prob(x, t)
    out["vp::mean"] = mean.numpy()
    out["vp::std"] = std.numpy()
    out["vp::params"] = np.array([0.1, 20., 1000], dtype=np.float64)
    save("sde.npz", **out)


def fcn_config(hidden_nodes, hidden_layers=5, state_size=100):
    c = ConfigDict()
    c.model = ConfigDict(name="fcn", state_size=state_size, hidden_layers=hidden_layers,
                         hidden_nodes=hidden_nodes, dropout=0.0, sigma_min=1e-2, sigma_max=4,
                         num_scales=1000)
    c.training = ConfigDict(sde="vesde", continuous=True)
    return c


def gen_fcn():
    out = {}
    # tiny model, weights stored
    torch.manual_seed(0)
    cfg = fcn_config(64)
    model = mutils.create_model(cfg)
    sde = sde_lib.VESDE(sigma_min=1e-2, sigma_max=4, N=1000)
    score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 100, generator=g)
    t = torch.tensor([1e-5] * 4 + [0.01, 0.1, 0.3, 1.0])
    with torch.no_grad():
        y = score_fn(x, t)
    out.update(sd_arrays(model))
    out["x"] = x.numpy(); out["t"] = t.numpy(); out["score"] = y.numpy()
    out["cfg"] = np.array([100, 5, 64])
    save("fcn_tiny.npz", **out)

    # full-size model (10dim.py:97-103): weights are NOT stored, they are
    # reproduced from torch.manual_seed(0) + nn.Linear default init
    torch.manual_seed(0)
    cfg = fcn_config(2048)
    model = mutils.create_model(cfg)
    score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
    x = torch.randn(16, 100, generator=g)
    t = torch.full((16,), 1e-5)
    with torch.no_grad():
        y = score_fn(x, t)
    chk = np.array([float(v.double().abs().sum()) for v in model.state_dict().values()])
    save("fcn_full_seed0.npz", x=x.numpy(), t=t.numpy(), score=y.numpy(), weight_abs_sums=chk,
         cfg=np.array([100, 5, 2048]))


def ncsnpp_config(**over):
    c = ConfigDict()
    c.data = ConfigDict(image_size=32, effective_image_size=32, num_channels=3, centered=False,
                        shape=[3, 32, 32])
    c.training = ConfigDict(continuous=True, sde="vesde")
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


NCSNPP_VARIANTS = {
    # the benchmark family at reduced width
    "bench_init0": {},
    "bench_init1": {"model.init_scale": 1.0},
    # everything the other switches reach
    # (fir=False is only usable with biggan blocks and progressive='none': the reference's
    #  non-FIR Upsample passes 'nearest' as scale_factor and raises, layerspp.py:117)
    #  and its FIR Upsample-with-conv slices a tensor with step -1 and raises, up_or_down_sampling.py:126
    #  -> 'ddpm' blocks need resamp_with_conv=False, progressive='residual' never runs)
    "ddpm_outskip": {"model.init_scale": 1.0, "model.resblock_type": "ddpm", "model.fir": True,
                     "model.resamp_with_conv": False,
                     "model.progressive": "output_skip", "model.progressive_input": "input_skip",
                     "model.progressive_combine": "cat", "model.embedding_type": "positional",
                     "model.skip_rescale": False, "model.num_res_blocks": 1},
    "biggan_nofir": {"model.init_scale": 1.0, "model.fir": False, "model.progressive": "none",
                     "model.progressive_input": "input_skip", "model.num_res_blocks": 1,
                     "model.nonlinearity": "elu", "data.centered": True},
    "biggan_outskip_sum": {"model.init_scale": 1.0, "model.progressive": "output_skip",
                           "model.progressive_input": "input_skip", "model.progressive_combine": "sum",
                           "model.num_res_blocks": 1},
}


def gen_ncsnpp():
    for name, over in NCSNPP_VARIANTS.items():
        torch.manual_seed(0)
        cfg = ncsnpp_config(**over)
        model = mutils.create_model(cfg)
        sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=50, N=1000)
        score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
        g = torch.Generator().manual_seed(1)
        x = torch.rand(2, 3, 32, 32, generator=g)
        t = torch.tensor([1e-5, 0.2])
        with torch.no_grad():
            y = score_fn(x, t)
            raw = model.eval()(x, t * 999)
        out = sd_arrays(model)
        out["x"] = x.numpy(); out["t"] = t.numpy(); out["score"] = y.numpy(); out["model_out"] = raw.numpy()
        over_keys = sorted(over)
        out["override_keys"] = np.array(over_keys, dtype="U64")
        out["override_vals"] = np.array([repr(over[k]) for k in over_keys], dtype="U64")
        out["n_modules"] = np.array(len(model.all_modules))
        save(f"ncsnpp_{name}.npz", **out)


def beatgans_config(**over):
    c = ConfigDict()
    c.data = ConfigDict(image_size=16, effective_image_size=16, num_channels=3, centered=False, shape=[3, 16, 16])
    c.training = ConfigDict(continuous=True, sde="vesde")
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


BEATGANS_VARIANTS = {
    "paper_like": {},                                                  # style_gan_BeatGAN.py:29-82 at reduced size
    "plain_resample": {"model.resblock_updown": False, "model.num_res_blocks": 2,
                       "model.resnet_use_zero_module": False},          # Downsample/Upsample with 3x3 convs
}


def gen_beatgans():
    from models import BeatGANsUNET  # noqa: F401  (registers the model)
    for name, over in BEATGANS_VARIANTS.items():
        torch.manual_seed(0)
        cfg = beatgans_config(**over)
        model = mutils.create_model(cfg)
        # the reference zero-initialises every block's last conv, the attention projection and the output conv
        # (BeatGANs_nn.py:73-79); random values there make the fixture exercise every branch
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():
            for prm in model.parameters():
                if float(prm.abs().sum()) == 0.0:
                    prm.copy_(torch.randn(prm.shape, generator=g) * 0.05)
        sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=50, N=1000)
        score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
        gx = torch.Generator().manual_seed(1)
        x = torch.rand(2, 3, 16, 16, generator=gx)
        t = torch.tensor([1e-5, 0.2])
        with torch.no_grad():
            y = score_fn(x, t)
            raw = model.eval()(x, t * 999)
        out = sd_arrays(model)
        out["x"] = x.numpy(); out["t"] = t.numpy(); out["score"] = y.numpy(); out["model_out"] = raw.numpy()
        over_keys = sorted(over)
        out["override_keys"] = np.array(over_keys, dtype="U64")
        out["override_vals"] = np.array([repr(over[k]) for k in over_keys], dtype="U64")
        save(f"beatgans_{name}.npz", **out)


def ddpm_config(**over):
    c = ConfigDict()
    c.data = ConfigDict(image_size=16, effective_image_size=16, num_channels=1, centered=False, shape=[1, 16, 16])
    c.training = ConfigDict(continuous=True, sde="vesde")
    c.model = ConfigDict(name="ddpm", nf=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), dropout=0.1,
                         resamp_with_conv=True, conditional=True, nonlinearity="swish", normalization="GroupNorm",
                         input_channels=1, output_channels=1, sigma_min=0.009, sigma_max=50, num_scales=1000,
                         scale_by_sigma=True, ema_rate=0.999)
    for k, v in over.items():
        c[k] = v
    return c


DDPM_VARIANTS = {"mnist_like": {}, "pool_resample": {"model.resamp_with_conv": False, "model.nonlinearity": "elu",
                                                     "data.centered": True}}


def gen_ddpm():
    """`ddpm` is the model every shipped image config selects (e.g. .../image_data/MNIST/config.py:121)."""
    from models import ddpm  # noqa: F401  (registers the model)
    for name, over in DDPM_VARIANTS.items():
        torch.manual_seed(0)
        cfg = ddpm_config(**over)
        model = mutils.create_model(cfg)
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():   # zero-/1e-10-initialised tensors (init_scale=0.) get real values so every branch counts
            for prm in model.parameters():
                if float(prm.abs().max()) < 1e-6 and prm.ndim > 1:
                    prm.copy_(torch.randn(prm.shape, generator=g) * 0.05)
        sde = sde_lib.VESDE(sigma_min=0.009, sigma_max=50, N=1000)
        score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
        x = torch.rand(2, 1, 16, 16, generator=torch.Generator().manual_seed(1))
        t = torch.tensor([1e-5, 0.2])
        with torch.no_grad():
            y = score_fn(x, t)
            raw = model.eval()(x, t * 999)
        out = sd_arrays(model)
        out["x"] = x.numpy(); out["t"] = t.numpy(); out["score"] = y.numpy(); out["model_out"] = raw.numpy()
        over_keys = sorted(over)
        out["override_keys"] = np.array(over_keys, dtype="U64")
        out["override_vals"] = np.array([repr(over[k]) for k in over_keys], dtype="U64")
        out["n_modules"] = np.array(len(model.all_modules))
        save(f"ddpm_{name}.npz", **out)


def gen_ksphere():
    from lightning_data_modules.KSphereDataset import KSphereDataset
    out = {}
    for k in (10, 50):
        cfg = ConfigDict()
        cfg.data = ConfigDict(data_samples=32, n_spheres=1, ambient_dim=100, manifold_dim=k, noise_std=0.0,
                              embedding_type="random_isometry")
        torch.manual_seed(42)
        ds = KSphereDataset(cfg)
        out[f"k{k}::data"] = ds.data.numpy()
    save("ksphere.npz", **out)


def gen_svd_and_rule():
    import plot_utils  # reference (matplotlib Agg)
    out = {}
    g = torch.Generator().manual_seed(9)
    # (a) geometric spectra with a cliff; exact singular values known by construction
    mats = []
    for i, (m, d, k, cliff) in enumerate([(301, 40, 7, 60.0), (1501, 100, 10, 70.0), (1501, 100, 50, 30.0),
                                           (200, 64, 3, 1e3)]):
        u, _ = torch.linalg.qr(torch.randn(m, d, generator=g, dtype=torch.float64))
        v, _ = torch.linalg.qr(torch.randn(d, d, generator=g, dtype=torch.float64))
        s_true = torch.cat([torch.linspace(3000., 2000., d - k, dtype=torch.float64),
                            torch.linspace(2000. / cliff, 2000. / cliff / 3, k, dtype=torch.float64)])
        a = ((u * s_true) @ v.T + 0.37).float()      # non-zero column means -> centring matters
        mats.append(a)
        centred = a - a.mean(dim=0, keepdim=True)
        _, s, _ = torch.linalg.svd(centred)           # dim_reduction.py:193-197
        s64 = torch.linalg.svdvals(centred.double())
        out[f"m{i}::S"] = a.numpy()
        out[f"m{i}::sv_ref_f32"] = s.numpy()
        out[f"m{i}::sv_f64"] = s64.numpy()
        svd = {"singular_values": [s.tolist()]}
        dims_a = plot_utils.plot_distribution(svd, mode="all")
        _, dims_b = plot_utils.plot_dims(svd)
        assert dims_a == dims_b
        out[f"m{i}::dim"] = np.array(dims_a[0])
        out[f"m{i}::k_true"] = np.array(k)
    out["n_mats"] = np.array(len(mats))
    # (b) rule on hand-made spectra incl. flat tails and a largest gap at index 0 (excluded by the rule)
    spectra = [
        [10., 9., 8., 1., .9, .8],
        [100., 9., 8., 7., 6.5, 1., .5],
        [5., 4., 3.9, 3.8, 0.1],
        [3., 2., 1.],
        list(np.linspace(50, 40, 30)) + list(np.linspace(1, .5, 10)),
    ]
    for i, s in enumerate(spectra):
        svd = {"singular_values": [s]}
        d = plot_utils.plot_distribution(svd, mode="all")
        out[f"r{i}::s"] = np.array(s, dtype=np.float64)
        out[f"r{i}::dim"] = np.array(d[0])
    out["n_rules"] = np.array(len(spectra))
    # (c) 'mean' / 'first' aggregation of extract_sing_vals
    svd = {"singular_values": [spectra[0], [11., 9.5, 8., 2., .9, .1]]}
    out["agg::first"] = np.array(plot_utils.plot_distribution(svd, mode="first"))
    out["agg::mean"] = np.array(plot_utils.plot_distribution(svd, mode="mean"))
    out["agg::all"] = np.array(plot_utils.plot_distribution(svd, mode="all"))
    out["agg::s1"] = np.array(svd["singular_values"][1])
    save("svd_rule.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["upfirdn2d", "fused_act", "sde", "fcn", "ncsnpp", "ksphere", "svd", "beatgans", "ddpm"]
    table = {"upfirdn2d": gen_upfirdn2d, "fused_act": gen_fused_act, "sde": gen_sde, "fcn": gen_fcn,
             "ncsnpp": gen_ncsnpp, "ksphere": gen_ksphere, "svd": gen_svd_and_rule, "beatgans": gen_beatgans, "ddpm": gen_ddpm}
    for w in which:
        table[w]()
