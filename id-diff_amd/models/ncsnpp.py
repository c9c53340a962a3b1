"""NCSN++ score network executed on MI355X kernels (reference: /root/reference/models/ncsnpp.py:39-388).

Host side only holds parameters and a recorded execution plan; every array operation is a launch from
libidiff_hip.so.  What differs from the reference's eager PyTorch graph, by design for gfx950:

* activations are NHWC ([B, H*W, C]): the channel axis is the contraction axis of every conv / NIN, so the
  implicit-GEMM loader reads 16-byte channel vectors and the pointwise kernels are fully coalesced; NCHW only
  exists at the model boundary (3-channel input / output, padded to 4 channels inside);
* 3x3 / 1x1 convolutions, NIN, Dense and the two attention products all run on one fp32-MFMA implicit-GEMM
  kernel; bias, the per-sample time-embedding bias ``Dense_0(act(temb))[:, :, None, None]``
  (layerspp.py:258-259), the residual add and the 1/sqrt(2) skip rescale (:270-273) are fused into its epilogue;
* the Dense_0 projections of ALL residual blocks are one GEMM (their weights are stacked at load time);
* ``torch.cat([h, hs.pop()], dim=1)`` (ncsnpp.py:324) is never materialised on its own: GroupNorm reads the
  two sources and writes the concatenated normalised tensor the next conv consumes, and the 1x1 shortcut
  conv is split into two K-slices accumulated through the residual epilogue;
* FIR resampling calls the same ``upfirdn2d`` kernel as ``op.upfirdn2d`` with minor = C;
* the V bias of attention is added after P.V (softmax rows sum to one), so V^T can be produced directly
  in the K-contiguous layout the P.V product wants.

``state_dict`` keys are the reference's (``all_modules.<i>.<name>``): the constructor creates the same
module list in the same order, so Lightning checkpoints load after stripping ``score_model.``.
Switch combinations that raise inside the reference itself are refused at construction (see oracle/models.py).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from . import utils
from .base import HipScoreModel

_INV_SQRT2 = float(1.0 / np.sqrt(2.0))


# ------------------------------------------------------------------------------------------------------------
# parameter containers (names = reference attribute names; no forward, the executor below runs them)
# ------------------------------------------------------------------------------------------------------------
def _fan_avg_uniform(shape, scale=1.):
    """DDPM default_init (models/layers.py:54-91): variance_scaling(scale or 1e-10, 'fan_avg', 'uniform')."""
    scale = 1e-10 if scale == 0 else scale
    rf = np.prod(shape) / shape[1] / shape[0]
    variance = scale / ((shape[1] * rf + shape[0] * rf) / 2)
    return (torch.rand(*shape) * 2. - 1.) * np.sqrt(3 * variance)


def _conv(cin, cout, k, init_scale=1., stride=1, padding=None):
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=(k // 2 if padding is None else padding))
    conv.weight.data = _fan_avg_uniform(conv.weight.shape, init_scale)
    nn.init.zeros_(conv.bias)
    return conv


def _dense(cin, cout):
    lin = nn.Linear(cin, cout)
    lin.weight.data = _fan_avg_uniform(lin.weight.shape)
    nn.init.zeros_(lin.bias)
    return lin


def _gn(ch):
    return nn.GroupNorm(num_groups=min(ch // 4, 32), num_channels=ch, eps=1e-6)


class NIN(nn.Module):
    def __init__(self, cin, cout, init_scale=0.1):
        super().__init__()
        self.W = nn.Parameter(_fan_avg_uniform((cin, cout), init_scale))
        self.b = nn.Parameter(torch.zeros(cout))


class GaussianFourierProjection(nn.Module):
    def __init__(self, embedding_size, scale):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)


class FirConv2d(nn.Module):
    """up_or_down_sampling.Conv2d, down=True form (:23-56, :144-178)."""

    def __init__(self, cin, cout, ksize=3):
        super().__init__()
        self.weight = nn.Parameter(_fan_avg_uniform((cout, cin, ksize, ksize)))
        self.bias = nn.Parameter(torch.zeros(cout))


class Downsample(nn.Module):
    def __init__(self, in_ch, out_ch=None, with_conv=False, fir=False):
        super().__init__()
        out_ch = out_ch or in_ch
        self.fir, self.with_conv = fir, with_conv
        if with_conv and not fir:
            self.Conv_0 = _conv(in_ch, out_ch, 3, stride=2, padding=0)
        elif with_conv:
            self.Conv2d_0 = FirConv2d(in_ch, out_ch)


class Upsample(nn.Module):
    def __init__(self, with_conv=False, fir=False):
        super().__init__()
        if not fir or with_conv:
            raise NotImplementedError("the reference's Upsample raises for fir=False (layerspp.py:117) and for "
                                      "fir with conv (up_or_down_sampling.py:126)")


class AttnBlockpp(nn.Module):
    def __init__(self, channels, init_scale=0.):
        super().__init__()
        self.GroupNorm_0 = _gn(channels)
        self.NIN_0 = NIN(channels, channels)
        self.NIN_1 = NIN(channels, channels)
        self.NIN_2 = NIN(channels, channels)
        self.NIN_3 = NIN(channels, channels, init_scale=init_scale)


class ResnetBlockDDPMpp(nn.Module):
    def __init__(self, in_ch, out_ch, temb_dim, dropout, init_scale):
        super().__init__()
        self.GroupNorm_0 = _gn(in_ch)
        self.Conv_0 = _conv(in_ch, out_ch, 3)
        self.Dense_0 = _dense(temb_dim, out_ch)
        self.GroupNorm_1 = _gn(out_ch)
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = _conv(out_ch, out_ch, 3, init_scale=init_scale)
        if in_ch != out_ch:
            self.NIN_0 = NIN(in_ch, out_ch)
        self.in_ch, self.out_ch, self.up, self.down = in_ch, out_ch, False, False


class ResnetBlockBigGANpp(nn.Module):
    def __init__(self, in_ch, out_ch, temb_dim, dropout, init_scale, up=False, down=False):
        super().__init__()
        self.GroupNorm_0 = _gn(in_ch)
        self.Conv_0 = _conv(in_ch, out_ch, 3)
        self.Dense_0 = _dense(temb_dim, out_ch)
        self.GroupNorm_1 = _gn(out_ch)
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = _conv(out_ch, out_ch, 3, init_scale=init_scale)
        if in_ch != out_ch or up or down:
            self.Conv_2 = _conv(in_ch, out_ch, 1)
        self.in_ch, self.out_ch, self.up, self.down = in_ch, out_ch, up, down


class Combine(nn.Module):
    def __init__(self, dim1, dim2, method):
        super().__init__()
        self.Conv_0 = _conv(dim1, dim2, 1)
        self.method = method


_ACT_NAMES = {"swish": "silu", "elu": "elu", "relu": "relu", "lrelu": "lrelu"}


def _pad4(c):
    return (c + 3) // 4 * 4


class _T:
    """An NHWC activation: buffer [B, H*W, C] + geometry (+ the per-tile column sums [B, nsplit, C, 2] its producing
    contraction wrote through epilogue.colstats, which let the consuming GroupNorm skip its statistics pass)."""
    __slots__ = ("buf", "H", "W", "C", "stats", "norm")

    def __init__(self, buf, H, W, C, stats=None, norm=None):
        self.buf, self.H, self.W, self.C, self.stats = buf, H, W, C, stats
        # set on the output of a GroupNorm (and kept through the resamplers): (module, elements per normalised group, absolute gain of
        # what followed, modulated?) -- what HipScoreModel.pairs_admissible needs to decide whether the consumer may run on fp16 pairs
        self.norm = norm


# ------------------------------------------------------------------------------------------------------------
# launches of fewer workgroups than this stay on the F(2x2, 3x3) kernel (tests set it to 1 to send small batches through F(4x4, 3x3))
WINO43_MIN_WORKGROUPS = 512
WINO43_PAIRS_MIN_WORKGROUPS = 256


def _winograd43_pays(B, H, W, cin, cout, normed=False):
    """F(4x4, 3x3) where it is served AND faster than F(2x2, 3x3): a workgroup takes 32 tiles of 4x4 pixels x 64 channels and a
    CU holds one.  With the contraction on the fp32 matrix cores, maps of 4x4 pixels (one tile per sample) or launches of fewer
    than two workgroups per CU stay on the 2x2 form (measured 0.87x there, 1.2-1.33x elsewhere: profiles/r04_wino43_time.txt);
    on fp16 pairs (``normed`` inputs, see _conv) it wins from one workgroup per CU on, 4x4 maps included (127 us against 207,
    209 against 370 at B = 2240, 256 / 512 -> 256 channels)."""
    if normed and _lib.conv2d_winograd43h_ok(B, H, W, cin, cout):
        return ((B * (H // 4) * (W // 4) + 31) // 32) * (cout // 64) >= WINO43_PAIRS_MIN_WORKGROUPS
    if H < 8 or W < 8 or not _lib.conv2d_winograd43_ok(B, H, W, cin, cout):
        return False
    return ((B * (H // 4) * (W // 4) + 31) // 32) * (cout // 64) >= WINO43_MIN_WORKGROUPS


# launches of fewer workgroups than this, and maps narrower than this, stay on the 2-D pair kernel (tests set the first to 1)
WINO1D_MIN_WORKGROUPS = 256
WINO1D_MIN_WIDTH = 4


def _wino1d_pays(B, H, W, cin, cout):
    """The row-wise F(4, 3) pair kernel (csrc/wino1d.hip: twice the matrix work of F(4x4, 3x3) for half the operand traffic) where it is served
    and measured at least as fast as the 2-D pair kernel: at one workgroup (512 pixels x 64 channels) per CU and more -- 1.08-1.23x on 16 x 16
    and 32 x 32 maps, 1.03-1.06x on 8 x 8, 1.00-1.01x on 4 x 4 (profiles/r05_wino1d_probe.txt, B = 2240); 64-pixel rows (config 5) likewise."""
    if W < WINO1D_MIN_WIDTH or not _lib.conv2d_wino1d_ok(B, H, W, cin, cout):
        return False
    return ((B * H * W + 511) // 512) * (cout // 64) >= WINO1D_MIN_WORKGROUPS


@utils.register_model(name='ncsnpp')
class NCSNpp(HipScoreModel):
    def __init__(self, config):
        super().__init__()
        m = config.model
        self.config = config
        self.act_name = _ACT_NAMES[m.nonlinearity.lower()]
        self.nf = nf = m.nf
        ch_mult, nrb = m.ch_mult, m.num_res_blocks
        levels = len(ch_mult)
        res = [config.data.effective_image_size // (2 ** i) for i in range(levels)]
        self.fir, self.fir_kernel = m.fir, list(m.fir_kernel)
        self.centered = config.data.centered
        self.skip_rescale = m.skip_rescale
        self.embedding_type = m.embedding_type.lower()
        self.conditional = m.conditional
        self.resblock_type = resblock = m.resblock_type.lower()
        self.progressive, self.progressive_input = m.progressive.lower(), m.progressive_input.lower()
        prog, prog_in = self.progressive, self.progressive_input
        combine = m.progressive_combine.lower()
        init_scale = m.init_scale
        self.channels = C = config.data.num_channels
        assert prog in ('none', 'output_skip', 'residual') and prog_in in ('none', 'input_skip', 'residual')
        assert self.embedding_type in ('fourier', 'positional')
        if prog == 'residual':
            raise NotImplementedError("progressive='residual' raises inside the reference (up_or_down_sampling.py:126)")
        if not self.conditional:
            raise NotImplementedError("unconditional NCSN++ (no time embedding) is not on the manifold_dimension path")

        mods, plan = [], []

        def add(mod):
            mods.append(mod)
            return len(mods) - 1

        def make_res(cin, cout=None, up=False, down=False):
            cout = cout or cin
            if resblock == 'ddpm':
                return ResnetBlockDDPMpp(cin, cout, nf * 4, m.dropout, init_scale)
            if resblock == 'biggan':
                return ResnetBlockBigGANpp(cin, cout, nf * 4, m.dropout, init_scale, up=up, down=down)
            raise ValueError(f'resblock type {resblock} unrecognized.')

        if self.embedding_type == 'fourier':
            assert config.training.continuous, "Fourier features are only used for continuous training."
            plan.append(("fourier", add(GaussianFourierProjection(nf, m.fourier_scale))))
            embed_dim = 2 * nf
        else:
            plan.append(("positional", None))
            embed_dim = nf
        plan.append(("temb_mlp", add(_dense(embed_dim, nf * 4)), add(_dense(nf * 4, nf * 4))))
        if prog == 'output_skip':
            self.pyramid_upsample = Upsample(with_conv=False, fir=self.fir)
        if prog_in == 'input_skip':
            self.pyramid_downsample = Downsample(None, with_conv=False, fir=self.fir)

        pyr_in_ch = C
        plan.append(("stem", add(_conv(C, nf, 3))))
        skips, ch = [nf], nf
        for lvl in range(levels):
            for _ in range(nrb):
                out = nf * ch_mult[lvl]
                plan.append(("res_push", add(make_res(ch, out)),
                             add(AttnBlockpp(out, init_scale)) if res[lvl] in m.attn_resolutions else None))
                ch = out
                skips.append(ch)
            if lvl != levels - 1:
                if resblock == 'ddpm':
                    step = ["down", add(Downsample(ch, with_conv=m.resamp_with_conv, fir=self.fir)), False]
                else:
                    step = ["down", add(make_res(ch, down=True)), True]
                if prog_in == 'input_skip':
                    step += ["input_skip", add(Combine(pyr_in_ch, ch, combine))]
                    if combine == 'cat':
                        ch *= 2
                elif prog_in == 'residual':
                    step += ["residual", add(Downsample(pyr_in_ch, ch, with_conv=True, fir=self.fir))]
                    pyr_in_ch = ch
                else:
                    step += ["none", None]
                plan.append(tuple(step))
                skips.append(ch)
        ch = skips[-1]
        plan.append(("middle", add(make_res(ch)), add(AttnBlockpp(ch, init_scale)), add(make_res(ch))))
        for lvl in reversed(range(levels)):
            for _ in range(nrb + 1):
                out = nf * ch_mult[lvl]
                plan.append(("res_pop", add(make_res(ch + skips.pop(), out))))
                ch = out
            if res[lvl] in m.attn_resolutions:
                plan.append(("attn", add(AttnBlockpp(ch, init_scale))))
            if prog == 'output_skip':
                plan.append(("out_skip", add(_gn(ch)), add(_conv(ch, C, 3, init_scale=init_scale)), lvl == levels - 1))
            if lvl != 0:
                if resblock == 'ddpm':
                    if m.resamp_with_conv or not self.fir:
                        Upsample(with_conv=m.resamp_with_conv, fir=self.fir)  # raises like the reference would
                    plan.append(("up", add(Upsample(with_conv=False, fir=True)), False))
                else:
                    plan.append(("up", add(make_res(ch, up=True)), True))
        assert not skips
        if prog != 'output_skip':
            plan.append(("head", add(_gn(ch)), add(_conv(ch, C, 3, init_scale=init_scale))))
        else:
            plan.append(("head_pyramid",))
        self.all_modules = nn.ModuleList(mods)
        self._plan = plan

    # -------------------------------------------------------------------------------------------- packing
    @staticmethod
    def _pack_conv(conv, cin_split=None):
        """[Cout, Cin, KH, KW] -> K-contiguous panel [Cout, KH, KW, Cin_pad]; optional split of Cin in two."""
        w = conv.weight.detach().float()
        cout, cin, kh, kw = w.shape
        parts = [w] if cin_split is None else [w[:, :cin_split], w[:, cin_split:]]
        out = []
        for part in parts:
            c = part.shape[1]
            buf = torch.zeros(cout, kh, kw, _pad4(c), device=w.device)
            buf[..., :c] = part.permute(0, 2, 3, 1)
            out.append(buf.contiguous())
        return out if cin_split is not None else out[0]

    @staticmethod
    def _pack_nin(nin, cin_split=None):
        w = nin.W.detach().float().t().contiguous()  # [cout, cin]
        if cin_split is None:
            return w
        return [w[:, :cin_split].contiguous(), w[:, cin_split:].contiguous()]

    def _pack(self):
        dev = self.device
        M = self.all_modules
        pk = {"conv": {}, "nin": {}, "gn": {}, "dense_off": {}}
        # FIR taps (up_or_down_sampling._setup_kernel :181-188): outer product normalised to sum 1; x4 for up
        k = np.asarray(self.fir_kernel, dtype=np.float32)
        k = np.outer(k, k)
        k = k / np.sum(k)
        pk["fir_down"] = torch.tensor(k, device=dev)
        pk["fir_up"] = torch.tensor(k * 4.0, device=dev)
        pk["fir_len"] = int(k.shape[0])
        # absolute gain of the resamplers (bound of |output| / max|input|): the whole tap sum without zero insertion, the largest
        # polyphase tap sum with it
        pk["fir_gain"] = {"down": float(np.abs(k).sum()), "pre_conv": float(np.abs(k).sum()),
                          "up": float(max(np.abs(4.0 * k[py::2, px::2]).sum() for py in (0, 1) for px in (0, 1)))}
        dense_w, dense_b, off = [], [], 0
        for i, mod in enumerate(M):
            if isinstance(mod, (ResnetBlockBigGANpp, ResnetBlockDDPMpp)):
                pk["dense_off"][i] = off
                dense_w.append(mod.Dense_0.weight.detach().float())
                dense_b.append(mod.Dense_0.bias.detach().float())
                off += mod.out_ch
        pk["dense_w"] = torch.cat(dense_w, 0).contiguous()
        pk["dense_b"] = torch.cat(dense_b, 0).contiguous()
        pk["dense_total"] = off
        return pk

    def _conv_w(self, pk, key, conv, cin_split=None):
        if key not in pk["conv"]:
            pk["conv"][key] = (self._pack_conv(conv, cin_split), conv.bias.detach().float().contiguous())
        return pk["conv"][key]

    def _nin_w(self, pk, key, nin, cin_split=None):
        if key not in pk["nin"]:
            pk["nin"][key] = (self._pack_nin(nin, cin_split), nin.b.detach().float().contiguous())
        return pk["nin"][key]

    # -------------------------------------------------------------------------------------------- primitive steps
    def _new(self, B, H, W, C, like):
        return _T(torch.empty(B, H * W, C, device=like.device, dtype=torch.float32), H, W, C)

    def _gn_act(self, x, gn, act, x2=None, mod=None):
        """GroupNorm (+ scale-shift modulation ``mod`` [B, 2*Ctot]) (+activation) of x (or of cat[x, x2])."""
        B = x.buf.shape[0]
        HW = x.H * x.W
        C2 = x2.C if x2 is not None else 0
        G = gn.num_groups
        norm = (gn, ((x.C + C2) // G) * HW, 1.0, mod is not None)
        if x.stats is not None and (x2 is None or x2.stats is not None) and x.C + C2 <= 1024 and B <= 65535:
            # both sources carry the column sums their producing contraction wrote: no pass over the activations, and the
            # statistics are finished inside the apply kernel (one launch per GroupNorm)
            ws2, ns2 = x2.stats if x2 is not None else (None, 0)
            y = self._new(B, x.H, x.W, x.C + C2, x.buf)
            _lib.groupnorm_apply_colstats(x.buf, x.C, x2.buf if x2 is not None else None, C2, B, HW, G, x.stats[0], x.stats[1],
                                          ws2, ns2, gn.eps, gn.weight.detach(), gn.bias.detach(), act, y.buf, mod=mod)
            y.norm = norm
            return y
        stats = torch.empty(B * G * 2, device=x.buf.device, dtype=torch.float32)
        if x.stats is not None and (x2 is None or x2.stats is not None):
            ws2, ns2 = x2.stats if x2 is not None else (None, 0)
            _lib.groupnorm_finalize(x.stats[0], x.stats[1], x.C, ws2, ns2, C2, B, HW, G, gn.eps, stats)
        else:
            nsplit = _lib.groupnorm_nsplit(B, HW, x.C + C2)
            ws = torch.empty(B * nsplit * (x.C + C2) * 2, device=x.buf.device, dtype=torch.float64)
            _lib.groupnorm_stats(x.buf, x.C, x2.buf if x2 is not None else None, C2, B, HW, G, gn.eps, ws, stats)
        y = self._new(B, x.H, x.W, x.C + C2, x.buf)
        _lib.groupnorm_apply(x.buf, x.C, x2.buf if x2 is not None else None, C2, B, HW, G, stats,
                             gn.weight.detach(), gn.bias.detach(), act, y.buf, mod=mod)
        y.norm = norm
        return y

    def _conv(self, x, wt, bias, stride=1, pad=1, pad_hi=None, stats=False, normed=False, **ep):
        """``stats=True``: the output feeds a GroupNorm -> ask the epilogue for its per-tile column sums.
        ``normed=True``: the input is the output of a GroupNorm (+ activation, + FIR resampling), i.e. bounded by
        sqrt(group size) * |gamma| + |beta| -- only then may the F(4x4, 3x3) contraction run on fp16 pairs, whose transformed input
        must stay below 65504 (include/idiff_hip.h), and only if THIS checkpoint's gamma / beta keep that bound inside the range
        (base.HipScoreModel.pairs_admissible, decided once per layer at pack time); any other input takes the fp32 contraction."""
        B = x.buf.shape[0]
        cout, kh, kw, cin = wt.shape
        assert cin == x.C, (cin, x.C)
        if normed:
            assert x.norm is not None, "normed=True on a tensor that is not a GroupNorm's (resampled) output"
            gn, group_elems, gain, modulated = x.norm
            normed = self.pairs_admissible(gn, group_elems, gain=gain, transform=True, modulated=modulated)
        ph = pad if pad_hi is None else pad_hi
        OH = (x.H + pad + ph - kh) // stride + 1
        OW = (x.W + pad + ph - kw) // stride + 1
        y = self._new(B, OH, OW, cout, x.buf)
        if "rows_per_group" not in ep:
            ep["rows_per_group"] = OH * OW
        if (kh, kw, stride, pad, ph) == (3, 3, 1, 1, 1) and ep["rows_per_group"] == OH * OW and normed and _wino1d_pays(B, x.H, x.W, cin, cout):
            # Winograd F(4, 3) along the rows on fp16 pairs: 4.5 multiplications per output, half the operand traffic of the 2-D form
            bank = self._packed.setdefault("wino1d", {})
            key = id(wt)
            if key not in bank:
                bank[key] = (wt, _lib.wino1d_pack(wt, cin, cout))
            if stats:
                ns = _lib.conv2d_wino1d_colstats_split(B, x.H, x.W, cin, cout)
                if ns > 0:
                    y.stats = (torch.empty(B * ns * cout * 2, device=x.buf.device, dtype=torch.float64), ns)
                    ep["colstats"] = y.stats[0]
            _lib.conv2d_wino1d(x.buf, bank[key][1], y.buf, B, x.H, x.W, cin, cout, epilogue=_lib.make_epilogue(bias=bias, **ep))
            return y
        if (kh, kw, stride, pad, ph) == (3, 3, 1, 1, 1) and ep["rows_per_group"] == OH * OW and _winograd43_pays(B, x.H, x.W, cin, cout, normed):
            # Winograd F(4x4, 3x3): 2.25 multiplications per output (F(2x2, 3x3) below: 4, the implicit GEMM: 9)
            bank = self._packed.setdefault("wino43", {})
            pairs = normed and _lib.conv2d_winograd43h_ok(B, x.H, x.W, cin, cout)
            key = (id(wt), pairs)
            if key not in bank:
                bank[key] = (wt, _lib.winograd43_pack(wt, cin, cout, pairs=pairs))
            if stats:
                ns = _lib.conv2d_winograd43_colstats_split(B, x.H, x.W, cin, cout)
                if ns > 0:
                    y.stats = (torch.empty(B * ns * cout * 2, device=x.buf.device, dtype=torch.float64), ns)
                    ep["colstats"] = y.stats[0]
            _lib.conv2d_winograd43(x.buf, bank[key][1], y.buf, B, x.H, x.W, cin, cout, epilogue=_lib.make_epilogue(bias=bias, **ep), pairs=pairs)
            return y
        if (kh, kw, stride, pad, ph) == (3, 3, 1, 1, 1) and _lib.conv2d_winograd_ok(B, x.H, x.W, cin, cout):
            # Winograd F(2x2, 3x3): 2.25x fewer MFMA flops; the transformed filter bank is cached beside the panel
            # (keyed by the kernel form too: the split-precision form is asked for per call, so a switch flipped later or
            # another geometry through this layer picks its own bank instead of inheriting the first one packed)
            bank = self._packed.setdefault("wino", {})
            split = _lib.conv2d_winograd_split_ok(B, x.H, x.W, cin, cout)
            key = (id(wt), split)
            if key not in bank:
                bank[key] = (wt, _lib.winograd_pack(wt, cin, cout, split=split))
            if stats:
                ns = _lib.conv2d_winograd_colstats_split(B, x.H, x.W, cin, cout)
                if ns > 0:
                    y.stats = (torch.empty(B * ns * cout * 2, device=x.buf.device, dtype=torch.float64), ns)
                    ep["colstats"] = y.stats[0]
            _lib.conv2d_winograd(x.buf, bank[key][1], y.buf, B, x.H, x.W, cin, cout,
                                 epilogue=_lib.make_epilogue(bias=bias, **ep), split=split)
            return y
        if stats:
            ns = _lib.conv2d_colstats_split(B, x.H, x.W, cin, cout, kh, kw, stride, pad, pad_hi)
            if ns > 0:
                y.stats = (torch.empty(B * ns * cout * 2, device=x.buf.device, dtype=torch.float64), ns)
                ep["colstats"] = y.stats[0]
        _lib.conv2d_nhwc(x.buf, wt, y.buf, B, x.H, x.W, cin, cout, kh, kw, stride, pad,
                         epilogue=_lib.make_epilogue(bias=bias, **ep), pad_hi=pad_hi)
        return y

    def _pointwise(self, x, w, bias, stats=False, **ep):
        """1x1 conv / NIN on NHWC = plain GEMM over [B*HW, Cin]; w is [Cout, Cin]."""
        B = x.buf.shape[0]
        cout, cin = w.shape
        assert cin == x.C, (cin, x.C)
        y = self._new(B, x.H, x.W, cout, x.buf)
        if stats:
            ns = _lib.gemm_colstats_split(B * x.H * x.W, cout, cin, cin, w.stride(0), x.H * x.W)
            if ns > 0:
                y.stats = (torch.empty(B * ns * cout * 2, device=x.buf.device, dtype=torch.float64), ns)
                ep["colstats"] = y.stats[0]
        _lib.gemm(x.buf.view(-1, cin), w, out=y.buf.view(-1, cout), epilogue=_lib.make_epilogue(bias=bias, **ep))
        return y

    def _pointwise_pairs(self, pk, x, w, bias, act_scale, stats=False, **ep):
        """_pointwise on fp16 pairs for an activation that is NOT a GroupNorm's output but whose scale is known: ``act_scale`` = device
        {s, 1 / s} (the attention output is a convex combination of the rows of v: never beyond v's range, so v's scale serves)."""
        B = x.buf.shape[0]
        cout, cin = w.shape
        M = B * x.H * x.W
        if not _lib.gemm_pairs_ok(M, cout, cin):
            return self._pointwise(x, w, bias, stats=stats, **ep)
        y = self._new(B, x.H, x.W, cout, x.buf)
        if stats:
            ns = _lib.gemm_colstats_split(M, cout, cin, cin, w.stride(0), x.H * x.W)
            if ns > 0:
                y.stats = (torch.empty(B * ns * cout * 2, device=x.buf.device, dtype=torch.float64), ns)
                ep["colstats"] = y.stats[0]
        _lib.gemm_pairs(x.buf.view(-1, cin), w, _lib._pairs_scale_of(pk, w), y.buf.view(-1, cout),
                        epilogue=_lib.make_epilogue(bias=bias, **ep), act_scale=act_scale)
        return y

    def _fir(self, x, pk, mode):
        B = x.buf.shape[0]
        n = pk["fir_len"]
        if mode == "up":      # upsample_2d :195-224
            p = n - 2
            k, up, down, p0, p1 = pk["fir_up"], 2, 1, (p + 1) // 2 + 1, p // 2
        elif mode == "down":  # downsample_2d :227-257
            p = n - 2
            k, up, down, p0, p1 = pk["fir_down"], 1, 2, (p + 1) // 2, p // 2
        else:                 # FIR in front of the stride-2 conv, conv_downsample_2d :171-178 (3x3 conv)
            p = (n - 2) + 2
            k, up, down, p0, p1 = pk["fir_down"], 1, 1, (p + 1) // 2, p // 2
        OH = _lib.upfirdn2d_out_size(x.H, up, down, p0, p1, n)
        OW = _lib.upfirdn2d_out_size(x.W, up, down, p0, p1, n)
        y = self._new(B, OH, OW, x.C, x.buf)
        _lib.upfirdn2d_raw(x.buf, k, y.buf, B, x.H, x.W, x.C, up, up, down, down, p0, p1, p0, p1)
        if x.norm is not None:
            y.norm = (x.norm[0], x.norm[1], x.norm[2] * pk["fir_gain"][mode], x.norm[3])
        return y

    def _box(self, x, up):
        B = x.buf.shape[0]
        y = self._new(B, x.H * 2 if up else x.H // 2, x.W * 2 if up else x.W // 2, x.C, x.buf)
        _lib.resample2x_nhwc(x.buf, y.buf, B, x.H, x.W, x.C, up)
        y.norm = x.norm                                        # nearest x2 / 2x2 mean: never beyond the input's range
        return y

    def _add(self, a, b, scale):
        y = self._new(a.buf.shape[0], a.H, a.W, a.C, a.buf)
        _lib.add_scale(a.buf, b.buf, y.buf, a.buf.numel(), scale)
        return y

    def _cat(self, a, b):
        B = a.buf.shape[0]
        y = self._new(B, a.H, a.W, a.C + b.C, a.buf)
        _lib.concat_cols(a.buf, a.C, b.buf, b.C, y.buf, B * a.H * a.W)
        return y

    # -------------------------------------------------------------------------------------------- blocks
    def _resblock(self, idx, x, temb_all, pk, x2=None):
        """ResnetBlockBigGANpp / ResnetBlockDDPMpp (layerspp.py:166-274) on x or on cat[x, x2]."""
        mod = self.all_modules[idx]
        rs = _INV_SQRT2 if self.skip_rescale else 1.0
        c_in = x.C + (x2.C if x2 is not None else 0)
        assert c_in == mod.in_ch, (c_in, mod.in_ch)
        h = self._gn_act(x, mod.GroupNorm_0, self.act_name, x2)
        if mod.up or mod.down:
            assert x2 is None
            if self.fir:
                h, x = self._fir(h, pk, "up" if mod.up else "down"), self._fir(x, pk, "up" if mod.up else "down")
            else:
                h, x = self._box(h, mod.up), self._box(x, mod.up)
        w0, b0 = self._conv_w(pk, (idx, 0), mod.Conv_0)
        off = pk["dense_off"][idx]
        h = self._conv(h, w0, b0, rowbias=temb_all[:, off:off + mod.out_ch], stats=True, normed=True)
        h = self._gn_act(h, mod.GroupNorm_1, self.act_name)
        # shortcut
        if hasattr(mod, "Conv_2") or hasattr(mod, "NIN_0"):
            # cat[x, x2] with equal row pitch: one two-source contraction; otherwise two K-slices through the residual epilogue
            fused = x2 is not None and x.C == x2.C and x.C % 32 == 0
            split = x.C if (x2 is not None and not fused) else None
            if hasattr(mod, "Conv_2"):
                ws, bs = self._conv_w(pk, (idx, 2), mod.Conv_2, split)
                ws = [w.view(w.shape[0], -1) for w in ws] if split is not None else ws.view(ws.shape[0], -1)
            else:
                ws, bs = self._nin_w(pk, (idx, "nin"), mod.NIN_0, split)
            if x2 is None:
                sc = self._pointwise(x, ws, bs)
            elif fused:
                sc = self._new(x.buf.shape[0], x.H, x.W, ws.shape[0], x.buf)
                _lib.gemm_2src(x.buf.view(-1, x.C), x2.buf.view(-1, x2.C), ws, sc.buf.view(-1, ws.shape[0]),
                               epilogue=_lib.make_epilogue(bias=bs))
            else:
                part = self._pointwise(x, ws[0], bs)
                sc = self._pointwise(x2, ws[1], None, residual=part.buf)
        else:
            assert x2 is None
            sc = x
        w1, b1 = self._conv_w(pk, (idx, 1), mod.Conv_1)
        return self._conv(h, w1, b1, residual=sc.buf, out_scale=rs, stats=True, normed=True)   # next: a GroupNorm_0 / skip

    def _attn(self, idx, x, pk):
        """AttnBlockpp (layerspp.py:62-91)."""
        mod = self.all_modules[idx]
        B, HW, C = x.buf.shape[0], x.H * x.W, x.C
        n = self._gn_act(x, mod.GroupNorm_0, None)
        pairs = self.pairs_admissible(mod.GroupNorm_0, n.norm[1], transform=False)
        if (idx, "qk") not in pk["nin"]:
            wq, bq = self._nin_w(pk, (idx, 0), mod.NIN_0)
            wk, bk = self._nin_w(pk, (idx, 1), mod.NIN_1)
            pk["nin"][(idx, "qk")] = (torch.cat([wq, wk], 0).contiguous(), torch.cat([bq, bk], 0).contiguous())
        wqk, bqk = pk["nin"][(idx, "qk")]
        wv, bv = self._nin_w(pk, (idx, 2), mod.NIN_2)
        w3, b3 = self._nin_w(pk, (idx, 3), mod.NIN_3)
        dev = x.buf.device
        qk = torch.empty(B * HW, 2 * C, device=dev, dtype=torch.float32)
        _lib.gemm_normed(pk, n.buf.view(-1, C), wqk, qk, epilogue=_lib.make_epilogue(bias=bqk), pairs=pairs)    # n: a GroupNorm's output
        # V^T[b] = Wv^T-panel [C, Cin] x n[b]^T -> [C, HW], K-contiguous for the P.V product (bias deferred)
        vt = torch.empty(B, C, HW, device=dev, dtype=torch.float32)
        _lib.gemm_weight_times_normed_t(pk, wv, n.buf, vt, B, HW, C, pairs=pairs)
        mixed = torch.empty(B, HW, C, device=dev, dtype=torch.float32)
        rs = _INV_SQRT2 if self.skip_rescale else 1.0
        if pairs and _lib.attention256_ok(B, HW, C):
            # QK^T -> softmax -> PV in one launch, the logits never written (csrc/attention.hip); the operands' power-of-two scales from
            # the projections' row norms (their input n has unit variance times gamma's scale)
            key = (idx, "attn_scale")
            if key not in pk["nin"]:
                gam = float(torch.sqrt((mod.GroupNorm_0.weight.detach().double() ** 2).mean() + (mod.GroupNorm_0.bias.detach().double() ** 2).mean()))
                pk["nin"][key] = (_lib.pairs_scale_from_rows(wqk, bqk, gam), _lib.pairs_scale_from_rows(wv, bv, gam))
            s_qk, s_v = pk["nin"][key]
            _lib.attention256(qk, vt, mixed, B, C, s_qk, s_v, float(int(C) ** (-0.5)), bias_v=bv)
            return self._pointwise_pairs(pk, _T(mixed, x.H, x.W, C), w3, b3, s_v, residual=x.buf, out_scale=rs, stats=True)
        else:
            logits = torch.empty(B, HW, HW, device=dev, dtype=torch.float32)
            _lib.gemm(qk, qk[:, C:], out=logits, M=HW, N=HW, K=C, lda=2 * C, ldb=2 * C, ldc=HW, batch=B,
                      stride_a=HW * 2 * C, stride_b=HW * 2 * C, stride_c=HW * HW)
            _lib.softmax_rows(logits, logits, B * HW, HW, float(int(C) ** (-0.5)))
            _lib.gemm(logits, vt, out=mixed, M=HW, N=C, K=HW, lda=HW, ldb=HW, ldc=C, batch=B,
                      stride_a=HW * HW, stride_b=C * HW, stride_c=HW * C, epilogue=_lib.make_epilogue(bias=bv))
        return self._pointwise(_T(mixed, x.H, x.W, C), w3, b3, residual=x.buf, out_scale=rs, stats=True)

    def _downsample(self, idx, x, pk, **ep):
        """layerspp.Downsample (:129-163) for every (fir, with_conv) pair."""
        mod = self.all_modules[idx] if isinstance(idx, int) else idx
        if mod.fir:
            if not mod.with_conv:
                return self._fir(x, pk, "down")
            key = (id(mod), "firconv")
            if key not in pk["conv"]:
                pk["conv"][key] = (self._pack_conv(mod.Conv2d_0), mod.Conv2d_0.bias.detach().float().contiguous())
            w, b = pk["conv"][key]
            return self._conv(self._fir(x, pk, "pre_conv"), w, b, stride=2, pad=0, **ep)
        if mod.with_conv:
            w, b = self._conv_w(pk, (id(mod), "conv"), mod.Conv_0)
            return self._conv(x, w, b, stride=2, pad=0, pad_hi=1, **ep)  # F.pad(x, (0, 1, 0, 1)), layerspp.py:153-155
        return self._box(x, False)                                       # avg_pool2d(2)

    # -------------------------------------------------------------------------------------------- forward
    forward_accepts_out = True

    def forward(self, x, time_cond, out_rowscale=None, out=None):
        """``out``: optional contiguous fp32 buffer of B * C * H * W values that receives the NCHW result (the drivers pass the rows
        of S: no copy of the scores afterwards)."""
        x, time_cond = self._check_inputs(x, time_cond)
        if x.ndim != 4 or x.shape[1] != self.channels:
            raise RuntimeError(f"ncsnpp: expected [B, {self.channels}, H, W], got {tuple(x.shape)}")
        pk = self.packed()
        M = self.all_modules
        B, C, H, W = x.shape
        dev = x.device
        cp = _pad4(C)
        rs = _INV_SQRT2 if self.skip_rescale else 1.0
        temb = temb_all = h = pyr_in = pyr = None
        hs = []
        for step in self._plan:
            op = step[0]
            if op == "fourier":
                half = self.nf
                temb = torch.empty(B, 2 * half, device=dev, dtype=torch.float32)
                _lib.fourier_embed(time_cond, M[step[1]].W.detach(), temb, B, half)
            elif op == "positional":
                temb = torch.empty(B, self.nf, device=dev, dtype=torch.float32)
                _lib.positional_embed(time_cond, temb, B, self.nf)
            elif op == "temb_mlp":
                l0, l1 = M[step[1]], M[step[2]]
                t1 = _lib.gemm(temb, l0.weight.detach(), epilogue=_lib.make_epilogue(bias=l0.bias.detach(), act=self.act_name))
                # every block consumes Dense_0(act(temb)): apply act once, project for all blocks in one GEMM
                t2 = _lib.gemm(t1, l1.weight.detach(), epilogue=_lib.make_epilogue(bias=l1.bias.detach(), act=self.act_name))
                temb_all = _lib.gemm(t2, pk["dense_w"], epilogue=_lib.make_epilogue(bias=pk["dense_b"]))
            elif op == "stem":
                xin = _T(torch.empty(B, H * W, cp, device=dev, dtype=torch.float32), H, W, cp)
                if self.centered:
                    _lib.nchw_to_nhwc(x, xin.buf, B, C, H * W, cp)
                else:
                    _lib.nchw_to_nhwc(x, xin.buf, B, C, H * W, cp, 2.0, -1.0)  # 2x - 1, ncsnpp.py:264-266
                pyr_in = xin
                w, b = self._conv_w(pk, (step[1], "stem"), M[step[1]])
                hs = [self._conv(xin, w, b, stats=True)]
            elif op == "res_push":
                h = self._resblock(step[1], hs[-1], temb_all, pk)
                if step[2] is not None:
                    h = self._attn(step[2], h, pk)
                hs.append(h)
            elif op == "down":
                _, i_down, is_res, mode, i_pyr = step
                h = self._resblock(i_down, hs[-1], temb_all, pk) if is_res else self._downsample(i_down, hs[-1], pk)
                if mode == "input_skip":
                    pyr_in = self._downsample(self.pyramid_downsample, pyr_in, pk)
                    comb = M[i_pyr]
                    wc, bc = self._conv_w(pk, (i_pyr, "comb"), comb.Conv_0)
                    wc = wc.view(wc.shape[0], -1)
                    if comb.method == "cat":
                        h = self._cat(self._pointwise(pyr_in, wc, bc), h)
                    elif comb.method == "sum":
                        h = self._pointwise(pyr_in, wc, bc, residual=h.buf)
                    else:
                        raise ValueError(f'Method {comb.method} not recognized.')
                elif mode == "residual":
                    pyr_in = self._downsample(i_pyr, pyr_in, pk, residual=h.buf, out_scale=rs)  # (pyr + h)/sqrt2 fused
                    h = pyr_in
                hs.append(h)
            elif op == "middle":
                h = self._resblock(step[1], hs[-1], temb_all, pk)
                h = self._attn(step[2], h, pk)
                h = self._resblock(step[3], h, temb_all, pk)
            elif op == "res_pop":
                h = self._resblock(step[1], h, temb_all, pk, x2=hs.pop())
            elif op == "attn":
                h = self._attn(step[1], h, pk)
            elif op == "out_skip":
                n = self._gn_act(h, M[step[1]], self.act_name)
                key = (step[2], "oskip")
                if key not in pk["conv"]:
                    conv = M[step[2]]
                    wt = self._pack_conv(conv)                       # [C, 3, 3, ch]
                    wpad = torch.zeros(cp, *wt.shape[1:], device=dev)
                    wpad[:C] = wt
                    bpad = torch.zeros(cp, device=dev)
                    bpad[:C] = conv.bias.detach().float()
                    pk["conv"][key] = (wpad.contiguous(), bpad)
                w, b = pk["conv"][key]
                if step[3]:
                    pyr = self._conv(n, w, b, normed=True)
                else:
                    pyr = self._conv(n, w, b, residual=self._fir(pyr, pk, "up").buf, normed=True)
            elif op == "up":
                h = self._resblock(step[1], h, temb_all, pk) if step[2] else self._fir(h, pk, "up")
            elif op == "up_plain":   # layers.Upsample (models/layers.py:593-604): nearest x2 (+ 3x3 conv)
                h = self._box(h, True)
                if M[step[1]].with_conv:
                    w, b = self._conv_w(pk, (step[1], "upconv"), M[step[1]].Conv_0)
                    h = self._conv(h, w, b)
            elif op == "head":
                n = self._gn_act(h, M[step[1]], self.act_name)
                w, b = self._conv_w(pk, (step[2], "head"), M[step[2]])
                h = self._conv(n, w, b, normed=True)
            elif op == "head_pyramid":
                h = pyr
        assert not hs
        c_out = getattr(self, "out_channels", C)
        out = self._out_buffer(out, B, c_out, H, W, dev)
        _lib.nhwc_to_nchw(h.buf, out, B, c_out, H * W, h.C, out_rowscale)
        return out
