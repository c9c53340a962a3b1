"""Oracle restatement of the score networks on the hot path (CPU, PyTorch fp32).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Parameter names and shapes
match the reference's ``state_dict`` so its checkpoints (and the golden
fixtures) load with ``load_state_dict``.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .ops import upfirdn2d


# ----------------------------------------------------------------------------
# fcn
# ----------------------------------------------------------------------------
class FCN(nn.Module):
    """MLP score net, /root/reference/models/fcn.py:6-40 (2-D input branch :30-35).

    ``mlp`` = Linear(D+1,H), Dropout, ELU, hidden_layers x [Linear(H,H), Dropout,
    ELU], Linear(H,D); time is appended as one more input feature.
    """

    def __init__(self, config):
        super().__init__()
        m = config.model
        widths = [m.state_size + 1] + [m.hidden_nodes] * (m.hidden_layers + 1)
        layers = []
        for a, b in zip(widths[:-1], widths[1:]):
            layers += [nn.Linear(a, b), nn.Dropout(m.dropout), nn.ELU()]
        layers.append(nn.Linear(m.hidden_nodes, m.state_size))
        self.mlp = nn.Sequential(*layers)

    def forward(self, x, t):
        if x.ndim != 2:
            raise NotImplementedError("only [batch, state] inputs are on the hot path")
        return self.mlp(torch.cat([x, t[:, None]], dim=1))


# ----------------------------------------------------------------------------
# ncsnpp building blocks
# ----------------------------------------------------------------------------
def _fan_avg_uniform(shape, scale=1.):
    """DDPM 'default_init': variance_scaling(scale, fan_avg, uniform), in_axis=1, out_axis=0.

    /root/reference/models/layers.py:54-91; scale 0 is replaced by 1e-10.
    """
    scale = 1e-10 if scale == 0 else scale
    rf = np.prod(shape) / shape[1] / shape[0]
    variance = scale / ((shape[1] * rf + shape[0] * rf) / 2)
    return (torch.rand(*shape) * 2. - 1.) * np.sqrt(3 * variance)


def _conv(cin, cout, k, init_scale=1., stride=1, padding=None):
    """ddpm_conv1x1 / ddpm_conv3x3, layers.py:100-105,119-132."""
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


def _fir_taps(k, gain=1.):
    """_setup_kernel, up_or_down_sampling.py:181-188: outer product, normalised to sum 1."""
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k = k / np.sum(k)
    return torch.tensor(k * gain)


def fir_upsample(x, k, factor=2):
    """upsample_2d, up_or_down_sampling.py:195-224 (gain factor**2, pad ((p+1)//2+factor-1, p//2))."""
    taps = _fir_taps(k, gain=factor ** 2)
    p = taps.shape[0] - factor
    return upfirdn2d(x, taps, up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def fir_downsample(x, k, factor=2):
    """downsample_2d, up_or_down_sampling.py:227-257 (pad ((p+1)//2, p//2))."""
    taps = _fir_taps(k)
    p = taps.shape[0] - factor
    return upfirdn2d(x, taps, down=factor, pad=((p + 1) // 2, p // 2))


def box_upsample(x, factor=2):
    """naive_upsample_2d, up_or_down_sampling.py:59-63."""
    return x.repeat_interleave(factor, dim=2).repeat_interleave(factor, dim=3)


def box_downsample(x, factor=2):
    """naive_downsample_2d, up_or_down_sampling.py:66-69."""
    n, c, h, w = x.shape
    return x.reshape(n, c, h // factor, factor, w // factor, factor).mean(dim=(3, 5))


class NIN(nn.Module):
    """1x1 'network in network' as a channel contraction, layers.py:555-564."""

    def __init__(self, cin, cout, init_scale=0.1):
        super().__init__()
        self.W = nn.Parameter(_fan_avg_uniform((cin, cout), init_scale))
        self.b = nn.Parameter(torch.zeros(cout))

    def forward(self, x):
        return torch.einsum("bchw,cd->bdhw", x, self.W) + self.b[None, :, None, None]


class GaussianFourierProjection(nn.Module):
    """layerspp.py:32-41: [sin, cos](2*pi*t*W), W fixed random."""

    def __init__(self, embedding_size, scale):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)

    def forward(self, t):
        proj = t[:, None] * self.W[None, :] * 2 * np.pi
        return torch.cat([proj.sin(), proj.cos()], dim=-1)


def positional_embedding(t, dim, max_positions=10000):
    """get_timestep_embedding, layers.py:524-538."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(max_positions) / (half - 1)))
    arg = t.float()[:, None] * freq[None, :]
    emb = torch.cat([arg.sin(), arg.cos()], dim=1)
    return F.pad(emb, (0, 1)) if dim % 2 == 1 else emb


class FirConv2d(nn.Module):
    """up_or_down_sampling.Conv2d (:23-56) in its down=True form (conv_downsample_2d :144-178):
    FIR with pad ((p+1)//2, p//2), p = (len(k)-factor)+(ksize-1), then stride-2 VALID conv + bias.
    The up=True form raises in the reference (negative-step slice, :126) and is not restated."""

    def __init__(self, cin, cout, ksize, down, resample_kernel):
        super().__init__()
        assert down
        self.weight = nn.Parameter(_fan_avg_uniform((cout, cin, ksize, ksize)))
        self.bias = nn.Parameter(torch.zeros(cout))
        self.k, self.ksize = resample_kernel, ksize

    def forward(self, x):
        taps = _fir_taps(self.k)
        p = (taps.shape[0] - 2) + (self.ksize - 1)
        x = upfirdn2d(x, taps, pad=((p + 1) // 2, p // 2))
        return F.conv2d(x, self.weight, stride=2) + self.bias.reshape(1, -1, 1, 1)


class Downsample(nn.Module):
    """layerspp.Downsample :129-163."""

    def __init__(self, in_ch, out_ch=None, with_conv=False, fir=False, fir_kernel=(1, 3, 3, 1)):
        super().__init__()
        out_ch = out_ch or in_ch
        self.fir, self.with_conv, self.k = fir, with_conv, fir_kernel
        if with_conv and not fir:
            self.Conv_0 = _conv(in_ch, out_ch, 3, stride=2, padding=0)
        elif with_conv:
            self.Conv2d_0 = FirConv2d(in_ch, out_ch, 3, True, fir_kernel)

    def forward(self, x):
        if self.fir:
            return self.Conv2d_0(x) if self.with_conv else fir_downsample(x, self.k)
        if self.with_conv:
            return self.Conv_0(F.pad(x, (0, 1, 0, 1)))
        return F.avg_pool2d(x, 2, stride=2)


class Upsample(nn.Module):
    """layerspp.Upsample :94-126; only the FIR-without-conv form runs in the reference."""

    def __init__(self, in_ch, out_ch=None, with_conv=False, fir=False, fir_kernel=(1, 3, 3, 1)):
        super().__init__()
        if not fir or with_conv:
            raise NotImplementedError("reference Upsample raises for fir=False (layerspp.py:117) "
                                      "and for fir+with_conv (up_or_down_sampling.py:126)")
        self.k = fir_kernel

    def forward(self, x):
        return fir_upsample(x, self.k)


class AttnBlockpp(nn.Module):
    """layerspp.AttnBlockpp :62-91: single-head attention over H*W tokens."""

    def __init__(self, channels, skip_rescale=False, init_scale=0.):
        super().__init__()
        self.GroupNorm_0 = _gn(channels)
        self.NIN_0 = NIN(channels, channels)
        self.NIN_1 = NIN(channels, channels)
        self.NIN_2 = NIN(channels, channels)
        self.NIN_3 = NIN(channels, channels, init_scale=init_scale)
        self.skip_rescale = skip_rescale

    def forward(self, x):
        b, c, h, w = x.shape
        n = self.GroupNorm_0(x)
        q = self.NIN_0(n).reshape(b, c, h * w)
        k = self.NIN_1(n).reshape(b, c, h * w)
        v = self.NIN_2(n).reshape(b, c, h * w)
        logits = torch.einsum("bcq,bck->bqk", q, k) * (int(c) ** (-0.5))
        probs = F.softmax(logits, dim=-1)
        mixed = torch.einsum("bqk,bck->bcq", probs, v).reshape(b, c, h, w)
        out = x + self.NIN_3(mixed)
        return out / np.sqrt(2.) if self.skip_rescale else out


class ResnetBlockDDPMpp(nn.Module):
    """layerspp.ResnetBlockDDPMpp :166-209."""

    def __init__(self, act, in_ch, out_ch=None, temb_dim=None, dropout=0.1, skip_rescale=False, init_scale=0.):
        super().__init__()
        out_ch = out_ch or in_ch
        self.GroupNorm_0 = _gn(in_ch)
        self.Conv_0 = _conv(in_ch, out_ch, 3)
        if temb_dim is not None:
            self.Dense_0 = _dense(temb_dim, out_ch)
        self.GroupNorm_1 = _gn(out_ch)
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = _conv(out_ch, out_ch, 3, init_scale=init_scale)
        if in_ch != out_ch:
            self.NIN_0 = NIN(in_ch, out_ch)
        self.act, self.out_ch, self.skip_rescale = act, out_ch, skip_rescale

    def forward(self, x, temb=None):
        h = self.Conv_0(self.act(self.GroupNorm_0(x)))
        if temb is not None:
            h = h + self.Dense_0(self.act(temb))[:, :, None, None]
        h = self.Conv_1(self.Dropout_0(self.act(self.GroupNorm_1(h))))
        if x.shape[1] != self.out_ch:
            x = self.NIN_0(x)
        return (x + h) / np.sqrt(2.) if self.skip_rescale else x + h


class ResnetBlockBigGANpp(nn.Module):
    """layerspp.ResnetBlockBigGANpp :212-274 (FIR or box resampling of both branches)."""

    def __init__(self, act, in_ch, out_ch=None, temb_dim=None, up=False, down=False, dropout=0.1,
                 fir=False, fir_kernel=(1, 3, 3, 1), skip_rescale=True, init_scale=0.):
        super().__init__()
        out_ch = out_ch or in_ch
        self.GroupNorm_0 = _gn(in_ch)
        self.Conv_0 = _conv(in_ch, out_ch, 3)
        if temb_dim is not None:
            self.Dense_0 = _dense(temb_dim, out_ch)
        self.GroupNorm_1 = _gn(out_ch)
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = _conv(out_ch, out_ch, 3, init_scale=init_scale)
        if in_ch != out_ch or up or down:
            self.Conv_2 = _conv(in_ch, out_ch, 1)
        self.act, self.up, self.down, self.fir, self.k = act, up, down, fir, fir_kernel
        self.in_ch, self.out_ch, self.skip_rescale = in_ch, out_ch, skip_rescale

    def _resample(self, z):
        if self.up:
            return fir_upsample(z, self.k) if self.fir else box_upsample(z)
        if self.down:
            return fir_downsample(z, self.k) if self.fir else box_downsample(z)
        return z

    def forward(self, x, temb=None):
        h = self._resample(self.act(self.GroupNorm_0(x)))
        x = self._resample(x)
        h = self.Conv_0(h)
        if temb is not None:
            h = h + self.Dense_0(self.act(temb))[:, :, None, None]
        h = self.Conv_1(self.Dropout_0(self.act(self.GroupNorm_1(h))))
        if self.in_ch != self.out_ch or self.up or self.down:
            x = self.Conv_2(x)
        return (x + h) / np.sqrt(2.) if self.skip_rescale else x + h


class Combine(nn.Module):
    """layerspp.Combine :44-59."""

    def __init__(self, dim1, dim2, method="cat"):
        super().__init__()
        self.Conv_0 = _conv(dim1, dim2, 1)
        self.method = method

    def forward(self, x, y):
        h = self.Conv_0(x)
        return torch.cat([h, y], dim=1) if self.method == "cat" else h + y


def get_act(config):
    """layers.get_act :29-41."""
    return {"elu": nn.ELU, "relu": nn.ReLU, "swish": nn.SiLU,
            "lrelu": lambda: nn.LeakyReLU(negative_slope=0.2)}[config.model.nonlinearity.lower()]()


# ----------------------------------------------------------------------------
# ncsnpp
# ----------------------------------------------------------------------------
class NCSNpp(nn.Module):
    """NCSN++ U-Net, /root/reference/models/ncsnpp.py:39-388.

    The constructor appends modules to ``all_modules`` in the reference's order
    (that order is the checkpoint key layout) and, at the same time, records a
    list of forward steps, so ``forward`` replays the recorded plan instead of
    re-deriving the module walk.
    """

    def __init__(self, config):
        super().__init__()
        m = config.model
        self.act = act = get_act(config)
        nf, ch_mult, nrb = m.nf, m.ch_mult, m.num_res_blocks
        levels = len(ch_mult)
        res = [config.data.effective_image_size // (2 ** i) for i in range(levels)]
        fir, k = m.fir, m.fir_kernel
        self.centered = config.data.centered
        self.skip_rescale = m.skip_rescale
        self.embedding_type = m.embedding_type.lower()
        self.conditional = m.conditional
        self.nf = nf
        resblock = m.resblock_type.lower()
        prog, prog_in = m.progressive.lower(), m.progressive_input.lower()
        combine = m.progressive_combine.lower()
        init_scale = m.init_scale
        C = config.data.num_channels

        mods, plan = [], []

        def add(mod):
            mods.append(mod)
            return len(mods) - 1

        def make_res(cin, cout=None, up=False, down=False):
            if resblock == "ddpm":
                assert not (up or down)
                return ResnetBlockDDPMpp(act, cin, cout, temb_dim=nf * 4, dropout=m.dropout,
                                         skip_rescale=m.skip_rescale, init_scale=init_scale)
            return ResnetBlockBigGANpp(act, cin, cout, temb_dim=nf * 4, up=up, down=down, dropout=m.dropout,
                                       fir=fir, fir_kernel=k, skip_rescale=m.skip_rescale, init_scale=init_scale)

        def make_attn(ch):
            return AttnBlockpp(ch, skip_rescale=m.skip_rescale, init_scale=init_scale)

        if self.embedding_type == "fourier":
            assert config.training.continuous
            plan.append(("fourier", add(GaussianFourierProjection(nf, m.fourier_scale))))
            embed_dim = 2 * nf
        else:
            plan.append(("positional", None))
            embed_dim = nf
        if m.conditional:
            plan.append(("temb_mlp", add(_dense(embed_dim, nf * 4)), add(_dense(nf * 4, nf * 4))))

        if prog == "output_skip":
            self.pyramid_upsample = Upsample(None, fir=fir, fir_kernel=k, with_conv=False)
        if prog_in == "input_skip":
            self.pyramid_downsample = Downsample(None, fir=fir, fir_kernel=k, with_conv=False)

        pyr_in_ch = C
        plan.append(("stem", add(_conv(C, nf, 3))))
        skips = [nf]
        ch = nf
        for lvl in range(levels):
            for _ in range(nrb):
                out = nf * ch_mult[lvl]
                plan.append(("res_push", add(make_res(ch, out)),
                             add(make_attn(out)) if res[lvl] in m.attn_resolutions else None))
                ch = out
                skips.append(ch)
            if lvl != levels - 1:
                if resblock == "ddpm":
                    i_down = add(Downsample(ch, with_conv=m.resamp_with_conv, fir=fir, fir_kernel=k))
                    step = ["down", i_down, False]
                else:
                    step = ["down", add(make_res(ch, down=True)), True]
                if prog_in == "input_skip":
                    step += ["input_skip", add(Combine(pyr_in_ch, ch, method=combine))]
                    if combine == "cat":
                        ch *= 2
                elif prog_in == "residual":
                    step += ["residual", add(Downsample(pyr_in_ch, ch, with_conv=True, fir=fir, fir_kernel=k))]
                    pyr_in_ch = ch
                else:
                    step += ["none", None]
                plan.append(tuple(step))
                skips.append(ch)

        ch = skips[-1]
        plan.append(("middle", add(make_res(ch)), add(make_attn(ch)), add(make_res(ch))))

        pyr_ch = 0
        for lvl in reversed(range(levels)):
            for _ in range(nrb + 1):
                out = nf * ch_mult[lvl]
                plan.append(("res_pop", add(make_res(ch + skips.pop(), out))))
                ch = out
            if res[lvl] in m.attn_resolutions:
                plan.append(("attn", add(make_attn(ch))))
            if prog != "none":
                if prog == "residual":
                    raise NotImplementedError("progressive='residual' raises in the reference "
                                              "(up_or_down_sampling.py:126)")
                first = lvl == levels - 1
                plan.append(("out_skip", add(_gn(ch)),
                             add(_conv(ch, C, 3, init_scale=init_scale)), first))
                pyr_ch = C
            if lvl != 0:
                if resblock == "ddpm":
                    plan.append(("up", add(Upsample(ch, with_conv=m.resamp_with_conv, fir=fir, fir_kernel=k)), False))
                else:
                    plan.append(("up", add(make_res(ch, up=True)), True))
        assert not skips
        if prog != "output_skip":
            plan.append(("head", add(_gn(ch)), add(_conv(ch, C, 3, init_scale=init_scale))))
        else:
            plan.append(("head_pyramid",))

        self.all_modules = nn.ModuleList(mods)
        self._plan = plan

    def forward(self, x, time_cond):
        M = self.all_modules
        rs = (lambda a, b: (a + b) / np.sqrt(2.)) if self.skip_rescale else (lambda a, b: a + b)
        temb, h, hs, pyr_in, pyr = None, None, [], None, None
        for step in self._plan:
            op = step[0]
            if op == "fourier":
                temb = M[step[1]](time_cond)
            elif op == "positional":
                temb = positional_embedding(time_cond, self.nf)
            elif op == "temb_mlp":
                temb = M[step[2]](self.act(M[step[1]](temb)))
            elif op == "stem":
                if not self.conditional:
                    temb = None
                if not self.centered:
                    x = 2 * x - 1.
                pyr_in = x
                hs = [M[step[1]](x)]
            elif op == "res_push":
                h = M[step[1]](hs[-1], temb)
                if step[2] is not None:
                    h = M[step[2]](h)
                hs.append(h)
            elif op == "down":
                _, i_down, takes_temb, mode, i_pyr = step
                h = M[i_down](hs[-1], temb) if takes_temb else M[i_down](hs[-1])
                if mode == "input_skip":
                    pyr_in = self.pyramid_downsample(pyr_in)
                    h = M[i_pyr](pyr_in, h)
                elif mode == "residual":
                    pyr_in = rs(M[i_pyr](pyr_in), h)
                    h = pyr_in
                hs.append(h)
            elif op == "middle":
                h = M[step[1]](hs[-1], temb)
                h = M[step[2]](h)
                h = M[step[3]](h, temb)
            elif op == "res_pop":
                h = M[step[1]](torch.cat([h, hs.pop()], dim=1), temb)
            elif op == "attn":
                h = M[step[1]](h)
            elif op == "out_skip":
                contrib = M[step[2]](self.act(M[step[1]](h)))
                pyr = contrib if step[3] else self.pyramid_upsample(pyr) + contrib
            elif op == "up":
                h = M[step[1]](h, temb) if step[2] else M[step[1]](h)
            elif op == "head":
                h = M[step[2]](self.act(M[step[1]](h)))
            elif op == "head_pyramid":
                h = pyr
        assert not hs
        return h


# ----------------------------------------------------------------------------
# ddpm (the model every shipped image config selects)
# ----------------------------------------------------------------------------
class _DDPMRes(nn.Module):
    """layers.ResnetBlockDDPM :632-680: GroupNorm(32) blocks, NIN shortcut, plain residual sum."""

    def __init__(self, act, in_ch, out_ch, temb_dim, dropout):
        super().__init__()
        self.GroupNorm_0 = nn.GroupNorm(num_groups=32, num_channels=in_ch, eps=1e-6)
        self.Conv_0 = _conv(in_ch, out_ch, 3)
        self.Dense_0 = _dense(temb_dim, out_ch)
        self.GroupNorm_1 = nn.GroupNorm(num_groups=32, num_channels=out_ch, eps=1e-6)
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = _conv(out_ch, out_ch, 3, init_scale=0.)
        if in_ch != out_ch:
            self.NIN_0 = NIN(in_ch, out_ch)
        self.act, self.in_ch, self.out_ch = act, in_ch, out_ch

    def forward(self, x, temb):
        h = self.Conv_0(self.act(self.GroupNorm_0(x)))
        h = h + self.Dense_0(self.act(temb))[:, :, None, None]
        h = self.Conv_1(self.Dropout_0(self.act(self.GroupNorm_1(h))))
        return (self.NIN_0(x) if self.in_ch != self.out_ch else x) + h


class _DDPMAttn(AttnBlockpp):
    """layers.AttnBlock :567-590 = AttnBlockpp without rescale, 32 groups."""

    def __init__(self, channels):
        super().__init__(channels, skip_rescale=False, init_scale=0.)
        self.GroupNorm_0 = nn.GroupNorm(num_groups=32, num_channels=channels, eps=1e-6)


class _DDPMResample(nn.Module):
    """layers.Upsample :593-604 / layers.Downsample :607-629."""

    def __init__(self, channels, with_conv, up):
        super().__init__()
        self.up, self.with_conv = up, with_conv
        if with_conv:
            self.Conv_0 = _conv(channels, channels, 3) if up else _conv(channels, channels, 3, stride=2, padding=0)

    def forward(self, x):
        if self.up:
            h = F.interpolate(x, scale_factor=2, mode="nearest")
            return self.Conv_0(h) if self.with_conv else h
        return self.Conv_0(F.pad(x, (0, 1, 0, 1))) if self.with_conv else F.avg_pool2d(x, 2, 2)


class DDPM(nn.Module):
    """models/ddpm.py:80-213."""

    def __init__(self, config):
        super().__init__()
        m = config.model
        self.act = act = get_act(config)
        self.nf = nf = m.nf
        self.nrb, self.attn_res = m.num_res_blocks, m.attn_resolutions
        self.levels = levels = len(m.ch_mult)
        res = [config.data.effective_image_size // (2 ** i) for i in range(levels)]
        self.centered = config.data.centered
        assert m.conditional
        mods = [_dense(nf, nf * 4), _dense(nf * 4, nf * 4), _conv(m.input_channels, nf, 3)]
        skips, ch = [nf], nf
        for lvl in range(levels):
            for _ in range(m.num_res_blocks):
                mods.append(_DDPMRes(act, ch, nf * m.ch_mult[lvl], 4 * nf, m.dropout))
                ch = nf * m.ch_mult[lvl]
                if res[lvl] in m.attn_resolutions:
                    mods.append(_DDPMAttn(ch))
                skips.append(ch)
            if lvl != levels - 1:
                mods.append(_DDPMResample(ch, m.resamp_with_conv, up=False))
                skips.append(ch)
        mods += [_DDPMRes(act, ch, ch, 4 * nf, m.dropout), _DDPMAttn(ch), _DDPMRes(act, ch, ch, 4 * nf, m.dropout)]
        for lvl in reversed(range(levels)):
            for _ in range(m.num_res_blocks + 1):
                mods.append(_DDPMRes(act, ch + skips.pop(), nf * m.ch_mult[lvl], 4 * nf, m.dropout))
                ch = nf * m.ch_mult[lvl]
            if res[lvl] in m.attn_resolutions:
                mods.append(_DDPMAttn(ch))
            if lvl != 0:
                mods.append(_DDPMResample(ch, m.resamp_with_conv, up=True))
        mods += [nn.GroupNorm(num_channels=ch, num_groups=32, eps=1e-6), _conv(ch, m.output_channels, 3, init_scale=0.)]
        self.all_modules = nn.ModuleList(mods)

    def forward(self, x, labels):
        M, i = self.all_modules, 0
        temb = M[1](self.act(M[0](positional_embedding(labels, self.nf))))
        h = x if self.centered else 2 * x - 1.
        hs = [M[2](h)]
        i = 3
        for lvl in range(self.levels):
            for _ in range(self.nrb):
                h = M[i](hs[-1], temb); i += 1
                if h.shape[-1] in self.attn_res:
                    h = M[i](h); i += 1
                hs.append(h)
            if lvl != self.levels - 1:
                hs.append(M[i](hs[-1])); i += 1
        h = M[i](hs[-1], temb); h = M[i + 1](h); h = M[i + 2](h, temb); i += 3
        for lvl in reversed(range(self.levels)):
            for _ in range(self.nrb + 1):
                h = M[i](torch.cat([h, hs.pop()], dim=1), temb); i += 1
            if h.shape[-1] in self.attn_res:
                h = M[i](h); i += 1
            if lvl != 0:
                h = M[i](h); i += 1
        assert not hs and i + 2 == len(M)
        return M[i + 1](self.act(M[i](h)))


def create_model(config):
    """models/utils.py:114-120."""
    from .beatgans import BeatGANsUNetModel
    models = {"fcn": FCN, "ncsnpp": NCSNpp, "BeatGANsUNetModel": BeatGANsUNetModel, "ddpm": DDPM}
    return models[config.model.name](config)
