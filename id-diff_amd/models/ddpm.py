"""``ddpm`` score network on the MI355X kernels (reference: /root/reference/models/ddpm.py:80-213 with
models/layers.py:567-680).  It is the model every shipped image config of the reference selects (e.g.
configs/dimension_estimation/paper/image_data/MNIST/config.py:121); SURVEY.md lists it under 8(f) "next".

The architecture is NCSN++ with its extras removed -- sinusoidal embedding, ``ResnetBlockDDPM`` (GroupNorm with 32
groups, NIN shortcut, plain residual sum), ``AttnBlock``, nearest / average-pool or conv resampling, no FIR, no
progressive paths -- so it reuses the NCSN++ executor (``ncsnpp.py``) unchanged: only the module list (= the
checkpoint key layout ``all_modules.<i>.*``) and the recorded plan are built here.
"""
import torch.nn as nn

from . import utils
from .base import HipScoreModel
from .ncsnpp import NCSNpp, AttnBlockpp, Downsample, ResnetBlockDDPMpp, _ACT_NAMES, _conv, _dense


def _gn32(ch):
    return nn.GroupNorm(num_groups=32, num_channels=ch, eps=1e-6)


class _Res(ResnetBlockDDPMpp):
    def __init__(self, in_ch, out_ch, temb_dim, dropout):
        super().__init__(in_ch, out_ch, temb_dim, dropout, init_scale=0.)
        self.GroupNorm_0, self.GroupNorm_1 = _gn32(in_ch), _gn32(out_ch)


class _Attn(AttnBlockpp):
    def __init__(self, channels):
        super().__init__(channels, init_scale=0.)
        self.GroupNorm_0 = _gn32(channels)


class _Upsample(nn.Module):
    def __init__(self, channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.Conv_0 = _conv(channels, channels, 3)


@utils.register_model(name='ddpm')
class DDPM(NCSNpp):
    def __init__(self, config):
        HipScoreModel.__init__(self)
        m = config.model
        self.config = config
        self.act_name = _ACT_NAMES[m.nonlinearity.lower()]
        self.nf = nf = m.nf
        levels = len(m.ch_mult)
        res = [config.data.effective_image_size // (2 ** i) for i in range(levels)]
        self.fir, self.fir_kernel = False, [1, 3, 3, 1]
        self.centered = config.data.centered
        self.skip_rescale = False
        self.embedding_type, self.conditional = 'positional', m.conditional
        self.resblock_type, self.progressive, self.progressive_input = 'ddpm', 'none', 'none'
        self.channels, self.out_channels = m.input_channels, m.output_channels
        if not m.conditional:
            raise NotImplementedError("unconditional ddpm (no time embedding) is not on the manifold_dimension path")
        if nf % 32:
            raise ValueError("ddpm uses GroupNorm with 32 groups: nf must be a multiple of 32")
        mods, plan = [], []

        def add(mod):
            mods.append(mod)
            return len(mods) - 1

        plan.append(("positional", None))
        plan.append(("temb_mlp", add(_dense(nf, nf * 4)), add(_dense(nf * 4, nf * 4))))
        plan.append(("stem", add(_conv(m.input_channels, nf, 3))))
        skips, ch = [nf], nf
        for lvl in range(levels):
            for _ in range(m.num_res_blocks):
                out = nf * m.ch_mult[lvl]
                plan.append(("res_push", add(_Res(ch, out, 4 * nf, m.dropout)),
                             add(_Attn(out)) if res[lvl] in m.attn_resolutions else None))
                ch = out
                skips.append(ch)
            if lvl != levels - 1:
                plan.append(("down", add(Downsample(ch, with_conv=m.resamp_with_conv, fir=False)), False, "none", None))
                skips.append(ch)
        plan.append(("middle", add(_Res(ch, ch, 4 * nf, m.dropout)), add(_Attn(ch)), add(_Res(ch, ch, 4 * nf, m.dropout))))
        for lvl in reversed(range(levels)):
            for _ in range(m.num_res_blocks + 1):
                out = nf * m.ch_mult[lvl]
                plan.append(("res_pop", add(_Res(ch + skips.pop(), out, 4 * nf, m.dropout))))
                ch = out
            if res[lvl] in m.attn_resolutions:
                plan.append(("attn", add(_Attn(ch))))
            if lvl != 0:
                plan.append(("up_plain", add(_Upsample(ch, m.resamp_with_conv))))
        assert not skips
        plan.append(("head", add(_gn32(ch)), add(_conv(ch, m.output_channels, 3, init_scale=0.))))
        self.all_modules = nn.ModuleList(mods)
        self._plan = plan
