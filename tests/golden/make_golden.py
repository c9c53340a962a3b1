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


def gen_ops_dtypes():
    """The two native ops in the other dtypes of the reference's dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
    op/upfirdn2d_kernel.cu:311, op/fused_bias_act_kernel.cu:79), produced by the reference's CPU branches run in that dtype:
    upfirdn2d_native (op/upfirdn2d.py:159-200) on float64 and on float16 tensors, fused_leaky_relu's CPU branch
    (op/fused_act.py:87-94) likewise.  Inputs are drawn in fp32 and rounded to the dtype, so the fixtures hold exactly
    representable values."""
    g = torch.Generator().manual_seed(4321)
    fir = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float64)
    fir /= fir.sum()
    asym4 = torch.randn(4, 4, generator=g).numpy().astype(np.float64)
    asym3 = torch.randn(3, 3, generator=g).numpy().astype(np.float64)
    spec = [
        (2, 3, 8, 8, fir, 1, 2, 1, 1),          # downsample_2d family
        (2, 3, 8, 8, fir, 1, 1, 2, 2),          # FIR before the stride-2 conv
        (2, 3, 4, 4, fir * 4, 2, 1, 2, 1),      # upsample_2d family
        (1, 2, 7, 5, asym4, 1, 1, 0, 0),        # asymmetric kernel: the flip
        (1, 2, 9, 6, asym3, 1, 2, 0, 1),
        (2, 2, 5, 7, asym4, 2, 2, 2, 1),
        (1, 2, 8, 9, asym3, 1, 1, -1, -2),      # negative pads crop
        (1, 4, 33, 17, fir, 1, 2, 1, 1),        # odd sizes
    ]
    out = {"n_cases": np.array(len(spec))}
    for name, dt in (("f64", torch.float64), ("f16", torch.float16)):
        for i, (n, c, h, w, k, up, down, p0, p1) in enumerate(spec):
            x = torch.randn(n, c, h, w, generator=g).to(dt)
            kt = torch.from_numpy(np.ascontiguousarray(k)).to(dt)
            y = upfirdn2d_native(x, kt, up, up, down, down, p0, p1, p0, p1)
            assert y.dtype == dt and torch.equal(y, ref_op.upfirdn2d(x, kt, up=up, down=down, pad=(p0, p1)))
            out[f"ufd::{name}::c{i}::x"], out[f"ufd::{name}::c{i}::k"], out[f"ufd::{name}::c{i}::y"] = x.numpy(), kt.numpy(), y.numpy()
            out[f"ufd::c{i}::params"] = np.array([up, down, p0, p1], dtype=np.int64)
        for i, shp in enumerate([(2, 5, 7, 3), (4, 8), (3, 8, 4), (1, 1, 1, 1)]):
            x = (torch.randn(*shp, generator=g) * 3).to(dt)
            b = torch.randn(shp[1], generator=g).to(dt)
            y = ref_op.fused_leaky_relu(x, b, negative_slope=0.2, scale=1.25)
            assert y.dtype == dt
            out[f"fba::{name}::c{i}::x"], out[f"fba::{name}::c{i}::b"], out[f"fba::{name}::c{i}::y_scale1.25"] = x.numpy(), b.numpy(), y.numpy()
        out["fba::n_cases"] = np.array(4)
    save("ops_dtypes.npz", **out)


def gen_sde_extra():
    """The two other SDEs configure_sde can build (BaseSdeGenerativeModel.py:33-35, 44-46): subVPSDE takes the VP branch of
    get_score_fn (models/utils.py:238-255); SNRSDE has a branch of its own (:270-277, fixture: gen_snr)."""
    t = torch.tensor([1e-5, 1e-3, 0.1, 0.3, 0.5, 1.0], dtype=torch.float32)
    x = torch.arange(12, dtype=torch.float32).reshape(6, 2)
    out = {"t": t.numpy(), "x": x.numpy()}
    s = sde_lib.subVPSDE(beta_min=0.1, beta_max=20., N=1000)
    mean, std = s.marginal_prob(x, t)
    out["subvp::mean"], out["subvp::std"] = mean.numpy(), std.numpy()
    s = sde_lib.SNRSDE(N=1000)
    mean, std = s.marginal_prob(x, t)
    out["snr::mean"], out["snr::std"] = mean.numpy(), std.numpy()
    save("sde_extra.npz", **out)


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


def fill_from_seed(module, seed, scale=0.05):
    """Overwrite every parameter with draws that a test can regenerate from (seed, state_dict order, shapes): wide
    models are pinned without storing tens of MB of weights.  1-D tensors (biases, norm weights) get 1 + draws for
    '.weight' of a norm layer and plain draws otherwise, through one generator, in state_dict order."""
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


def gen_vp():
    """VP branch of get_score_fn (models/utils.py:238-255): VPSDE beta in [0.1, 20], t = 1e-3 (sampling_eps of VP,
    BaseSdeGenerativeModel.py:44-47) and 0.2, on the weights of ncsnpp_bench_init1 (rebuilt from the same seed and
    checked against the stored ones); plus the perturbation mean/std the driver feeds it (dim_reduction.py:180-182)."""
    torch.manual_seed(0)
    cfg = ncsnpp_config(**NCSNPP_VARIANTS["bench_init1"])
    cfg.training.sde = "vpsde"
    model = mutils.create_model(cfg)
    stored = np.load(os.path.join(HERE, "ncsnpp_bench_init1.npz"))
    for k, v in model.state_dict().items():
        assert np.array_equal(stored["sd::" + k], v.numpy()), k
    sde = sde_lib.VPSDE(beta_min=0.1, beta_max=20., N=1000)
    score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(4, 3, 32, 32, generator=g)
    t = torch.tensor([1e-3, 1e-3, 0.2, 0.7])
    z = torch.randn(4, 3, 32, 32, generator=g)
    mean, std = sde.marginal_prob(x, t)
    perturbed = mean + std[(...,) + (None,) * 3] * z      # dim_reduction.py:180-182
    with torch.no_grad():
        y = score_fn(perturbed, t)
    save("ncsnpp_vp.npz", x=x.numpy(), t=t.numpy(), z=z.numpy(), mean=mean.numpy(), std=std.numpy(),
         perturbed=perturbed.numpy(), score=y.numpy(), weights_of=np.array("ncsnpp_bench_init1.npz"),
         params=np.array([0.1, 20., 1000]))


def gen_snr():
    """SNRSDE branch of the unconditional get_score_fn (models/utils.py:270-277; the SDE itself sde_lib.py:153-186): the
    reference's own score_fn on the weights of ncsnpp_bench_init1 at t = 1e-3 (sampling_eps, BaseSdeGenerativeModel.py:44-47),
    0.2 and 0.7, fed with the driver's perturbation mean + std * z (dim_reduction.py:180-182) from its marginal_prob."""
    torch.manual_seed(0)
    cfg = ncsnpp_config(**NCSNPP_VARIANTS["bench_init1"])
    cfg.training.sde = "snrsde"
    model = mutils.create_model(cfg)
    stored = np.load(os.path.join(HERE, "ncsnpp_bench_init1.npz"))
    for k, v in model.state_dict().items():
        assert np.array_equal(stored["sd::" + k], v.numpy()), k
    sde = sde_lib.SNRSDE(N=1000)
    score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
    g = torch.Generator().manual_seed(12)
    x = torch.rand(4, 3, 32, 32, generator=g)
    t = torch.tensor([1e-3, 1e-3, 0.2, 0.7])
    z = torch.randn(4, 3, 32, 32, generator=g)
    mean, std = sde.marginal_prob(x, t)
    perturbed = mean + std[(...,) + (None,) * 3] * z      # dim_reduction.py:180-182
    with torch.no_grad():
        y = score_fn(perturbed, t)
    save("ncsnpp_snr.npz", x=x.numpy(), t=t.numpy(), z=z.numpy(), mean=mean.numpy(), std=std.numpy(),
         perturbed=perturbed.numpy(), score=y.numpy(), weights_of=np.array("ncsnpp_bench_init1.npz"),
         params=np.array([1000]))


WIDE_SEED = 20260


def gen_wide():
    """Wide branches against reference output: nf = 128 so that GroupNorm's min(ch // 4, 32) caps at
    32 groups and the 3x3 convs are Winograd-eligible (Cin % 8 == 0, Cout % 64 == 0); BeatGANs at model_channels =
    128 so that GroupNorm32 has 4 channels per group.  Weights are NOT stored: ``fill_from_seed`` draws them."""
    torch.manual_seed(0)
    over = {"model.nf": 128, "model.ch_mult": (1,), "model.num_res_blocks": 1, "model.attn_resolutions": (8,),
            "model.init_scale": 1.0, "data.image_size": 8, "data.effective_image_size": 8, "data.shape": [3, 8, 8]}
    cfg = ncsnpp_config(**over)
    model = mutils.create_model(cfg)
    fill_from_seed(model, WIDE_SEED)
    sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=50, N=1000)
    score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
    x = torch.rand(3, 3, 8, 8, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([1e-5, 0.2, 0.9])
    with torch.no_grad():
        y = score_fn(x, t)
        raw = model.eval()(x, t * 999)
    ks = sorted(over)
    chk = np.array([float(v.double().abs().sum()) for v in model.state_dict().values() if v.dtype.is_floating_point])
    save("ncsnpp_wide.npz", x=x.numpy(), t=t.numpy(), score=y.numpy(), model_out=raw.numpy(), weight_abs_sums=chk,
         seed=np.array(WIDE_SEED), n_modules=np.array(len(model.all_modules)),
         override_keys=np.array(ks, dtype="U64"), override_vals=np.array([repr(over[k]) for k in ks], dtype="U64"))

    from models import BeatGANsUNET  # noqa: F401
    torch.manual_seed(0)
    over = {"model.model_channels": 128, "model.channel_mult": (1, 2), "model.embed_channels": 64,
            "model.attention_resolutions": (2,), "data.image_size": 8, "data.effective_image_size": 8,
            "data.shape": [3, 8, 8], "model.image_size": 8}
    cfg = beatgans_config(**over)
    model = mutils.create_model(cfg)
    fill_from_seed(model, WIDE_SEED + 1)
    score_fn = mutils.get_score_fn(sde, model, conditional=False, train=False, continuous=True)
    x = torch.rand(3, 3, 8, 8, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        y = score_fn(x, t)
        raw = model.eval()(x, t * 999)
    ks = sorted(over)
    chk = np.array([float(v.double().abs().sum()) for v in model.state_dict().values() if v.dtype.is_floating_point])
    save("beatgans_wide.npz", x=x.numpy(), t=t.numpy(), score=y.numpy(), model_out=raw.numpy(), weight_abs_sums=chk,
         seed=np.array(WIDE_SEED + 1),
         override_keys=np.array(ks, dtype="U64"), override_vals=np.array([repr(over[k]) for k in ks], dtype="U64"))


COND_SEED = 4242


def gen_conditional():
    """get_conditional_manifold_dimension (dim_reduction.py:12-114) RUN from the reference's own file: the Lightning
    registries it imports (lightning_modules.utils / lightning_data_modules.utils pull in the whole training stack)
    are replaced by two-function stand-ins that hand it a reference ncsnpp, the reference VESDE and a one-batch
    labelled validation loader.  The loop, the batching arithmetic, the label filter, the 12 levels, the SVD and the
    three pickles per level are the reference's code.  Noise = torch's global CPU generator seeded with COND_SEED,
    consumed in the reference's order (level, point, batch) -- a test replays it."""
    import pickle
    import tempfile

    over = {"model.init_scale": 1.0, "model.attn_resolutions": (8,), "data.image_size": 16,
            "data.effective_image_size": 16, "data.shape": [3, 16, 16], "model.num_res_blocks": 1}
    cfg = ncsnpp_config(**over)
    torch.manual_seed(0)
    model = mutils.create_model(cfg)
    B = 60
    g = torch.Generator().manual_seed(3)
    images = torch.rand(B, 3, 16, 16, generator=g)
    labels = torch.arange(B) % 3            # label 1 at items 1, 4, 7, ...

    class Module(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.score_model = model

        def load_from_checkpoint(self, path):
            return self

        def configure_sde(self, config):      # BaseSdeGenerativeModel.py:27-47, VE branch
            self.sde = sde_lib.VESDE(sigma_min=config.model.sigma_min, sigma_max=config.model.sigma_max,
                                     N=config.model.num_scales)
            self.sampling_eps = 1e-5

    class Data:
        def setup(self):
            pass

        def val_dataloader(self):
            return [(images, labels)]

    saved = {k: sys.modules.get(k) for k in ("lightning_modules", "lightning_modules.utils",
                                             "lightning_data_modules", "lightning_data_modules.utils", "dim_reduction")}
    lm, lmu = types.ModuleType("lightning_modules"), types.ModuleType("lightning_modules.utils")
    ld, ldu = types.ModuleType("lightning_data_modules"), types.ModuleType("lightning_data_modules.utils")
    lmu.create_lightning_module = lambda config: Module()
    ldu.create_lightning_datamodule = lambda config: Data()
    lm.utils, ld.utils = lmu, ldu
    sys.modules.update({"lightning_modules": lm, "lightning_modules.utils": lmu,
                        "lightning_data_modules": ld, "lightning_data_modules.utils": ldu})
    sys.modules.pop("dim_reduction", None)
    try:
        import dim_reduction as ref_dim   # /root/reference/dim_reduction.py
        assert ref_dim.__file__.startswith(REF)
        with tempfile.TemporaryDirectory() as tmp:
            cfg.logging = ConfigDict(log_path=tmp, log_name="cond")
            cfg.model.checkpoint_path = None
            cfg.device = "cpu"
            cfg.dim_estimation = ConfigDict(num_datapoints=3)
            torch.manual_seed(COND_SEED)
            ref_dim.get_conditional_manifold_dimension(cfg)
            root = os.path.join(tmp, "cond", "svd")
            levels = sorted(os.listdir(root))
            sv, lab, img = [], [], []
            for lv in levels:
                with open(os.path.join(root, lv, "labels_svd.pkl"), "rb") as f:
                    sv.append(np.array(pickle.load(f)["singular_values"], dtype=np.float64))
                with open(os.path.join(root, lv, "labels.pkl"), "rb") as f:
                    lab.append(np.array(pickle.load(f)["labels"]))
                with open(os.path.join(root, lv, "images.pkl"), "rb") as f:
                    img.append(pickle.load(f)["images"])
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    out = sd_arrays(model)
    ks = sorted(over)
    out.update(val_images=images.numpy(), val_labels=labels.numpy(), level_dirs=np.array(levels, dtype="U16"),
               singular_values=np.stack(sv), labels=np.stack(lab), images_pkl=np.stack(img)[0],
               seed=np.array(COND_SEED), num_datapoints=np.array(3),
               override_keys=np.array(ks, dtype="U64"), override_vals=np.array([repr(over[k]) for k in ks], dtype="U64"))
    save("conditional.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["upfirdn2d", "fused_act", "ops_dtypes", "sde", "sde_extra", "fcn", "ncsnpp", "ksphere", "svd", "beatgans", "ddpm", "vp", "snr", "wide", "conditional"]
    table = {"upfirdn2d": gen_upfirdn2d, "fused_act": gen_fused_act, "ops_dtypes": gen_ops_dtypes, "sde": gen_sde, "fcn": gen_fcn,
             "sde_extra": gen_sde_extra, "ncsnpp": gen_ncsnpp, "ksphere": gen_ksphere, "svd": gen_svd_and_rule, "beatgans": gen_beatgans, "ddpm": gen_ddpm, "vp": gen_vp, "snr": gen_snr, "wide": gen_wide,
             "conditional": gen_conditional}
    for w in which:
        table[w]()
