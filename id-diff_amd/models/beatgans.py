"""``BeatGANsUNetModel`` executed on MI355X kernels (reference: /root/reference/models/BeatGANsUNET.py:18-285,
BeatGANsblocks.py:80-491, BeatGANs_nn.py:23-125; configuration family of
configs/dimension_estimation/extra_experiments/styleGAN/style_gan_BeatGAN.py:29-82).

Same execution model as ``ncsnpp.py`` (NHWC activations, one fp32-MFMA implicit-GEMM kernel for every
contraction, fused epilogues, two-source GroupNorm instead of a materialised ``th.cat([x, lateral])``).  Specific
to this network:

* the scale-shift time conditioning ``norm(h) * (1 + scale) + shift`` (BeatGANsblocks.py:316-321) is folded into the
  GroupNorm-apply kernel, and ``emb_layers`` (SiLU -> Linear(E, 2*C)) of ALL residual blocks is one stacked GEMM;
* resampling is nearest x2 / 2x2 average (no FIR): ``idiff_resample2x_nhwc_f32``;
* attention uses the 1-D-conv QKV projection; with one head the legacy and the new channel orders coincide
  (q | k | v thirds of the projection), which is the only case the dimension-estimation configs use.

Module nesting (``input_blocks.<k>.<j>...``, ``middle_block``, ``output_blocks``, ``time_embed``, ``out``) reproduces the
reference's ``state_dict`` keys.
"""
import torch
import torch.nn as nn

from .. import _lib
from . import utils
from .base import HipScoreModel
from .ncsnpp import NCSNpp, _T, _pad4


def _normalization(ch):
    return nn.GroupNorm(min(32, ch), ch)     # GroupNorm32 (BeatGANs_nn.py:98-104), default eps 1e-5


def _zero(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class Resample(nn.Module):
    """Upsample / Downsample parameter holder (BeatGANsblocks.py:335-396)."""

    def __init__(self, channels, use_conv, up, out_channels=None):
        super().__init__()
        self.up, self.use_conv = up, use_conv
        out_channels = out_channels or channels
        if use_conv and up:
            self.conv = nn.Conv2d(channels, out_channels, 3, padding=1)
        elif use_conv:
            self.op = nn.Conv2d(channels, out_channels, 3, stride=2, padding=1)


class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, up=False, down=False, has_lateral=False,
                 use_zero_module=True):
        super().__init__()
        out_channels = out_channels or channels
        self.channels, self.out_channels = channels, out_channels
        self.in_layers = nn.Sequential(_normalization(channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.up, self.down, self.has_lateral = up, down, has_lateral
        if up or down:
            self.h_upd = Resample(channels, False, up)
            self.x_upd = Resample(channels, False, up)
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * out_channels))
        conv = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if use_zero_module:
            conv = _zero(conv)
        self.out_layers = nn.Sequential(_normalization(out_channels), nn.SiLU(), nn.Dropout(p=dropout), conv)
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads=1, num_head_channels=-1):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.norm = _normalization(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = _zero(nn.Conv1d(channels, channels, 1))


class Block(nn.Sequential):
    """TimestepEmbedSequential (BeatGANsblocks.py:31-43): container only."""


@utils.register_model(name='BeatGANsUNetModel')
class BeatGANsUNetModel(HipScoreModel):
    # shared NHWC primitives (defined once in ncsnpp.py)
    _new = NCSNpp._new
    _gn_act = NCSNpp._gn_act
    _conv = NCSNpp._conv
    _pointwise = NCSNpp._pointwise
    _pointwise_pairs = NCSNpp._pointwise_pairs
    _box = NCSNpp._box
    _cat = NCSNpp._cat
    _pack_conv = staticmethod(NCSNpp._pack_conv)

    def __init__(self, config):
        super().__init__()
        m = config.model
        self.conf = m
        if m.num_classes is not None or m.resnet_two_cond:
            raise NotImplementedError("class / two-condition variants are not on the manifold_dimension path")
        if m.dims != 2:
            raise NotImplementedError("only 2-D BeatGANs U-Nets are on the manifold_dimension path")
        self.mc = m.model_channels
        self.temb_ch = m.time_embed_channels or m.model_channels
        self.channels = m.in_channels
        E = m.embed_channels
        self.time_embed = nn.Sequential(nn.Linear(self.temb_ch, E), nn.SiLU(), nn.Linear(E, E))
        mults = list(m.channel_mult)
        in_mults = list(m.input_channel_mult or m.channel_mult)
        heads_up = m.num_heads if m.num_heads_upsample == -1 else m.num_heads_upsample
        zero = m.resnet_use_zero_module

        def res(ch, out=None, **kw):
            return ResBlock(ch, E, m.dropout, out_channels=out, use_zero_module=zero, **kw)

        def attn(ch, heads):
            blk = AttentionBlock(ch, heads, m.num_head_channels)
            if blk.num_heads != 1:
                raise NotImplementedError("multi-head BeatGANs attention is not used by the dimension-estimation configs")
            return blk

        ch = input_ch = int(mults[0] * self.mc)
        self.input_blocks = nn.ModuleList([Block(nn.Conv2d(m.in_channels, ch, 3, padding=1))])
        chans = [[] for _ in mults]
        chans[0].append(ch)
        self.input_num_blocks = [0] * len(mults)
        self.input_num_blocks[0] = 1
        self.output_num_blocks = [0] * len(mults)
        resolution = m.image_size
        for level, mult in enumerate(in_mults):
            for _ in range(m.num_input_res_blocks or m.num_res_blocks):
                layers = [res(ch, int(mult * self.mc))]
                ch = int(mult * self.mc)
                if resolution in m.attention_resolutions:
                    layers.append(attn(ch, m.num_heads))
                self.input_blocks.append(Block(*layers))
                chans[level].append(ch)
                self.input_num_blocks[level] += 1
            if level != len(mults) - 1:
                resolution //= 2
                self.input_blocks.append(Block(res(ch, ch, down=True) if m.resblock_updown
                                               else Resample(ch, m.conv_resample, False, ch)))
                chans[level + 1].append(ch)
                self.input_num_blocks[level + 1] += 1
        self.middle_block = Block(res(ch), attn(ch, m.num_heads), res(ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(mults))[::-1]:
            for i in range(m.num_res_blocks + 1):
                ich = chans[level].pop() if chans[level] else 0
                layers = [res(ch + ich, int(self.mc * mult), has_lateral=ich > 0)]
                ch = int(self.mc * mult)
                if resolution in m.attention_resolutions:
                    layers.append(attn(ch, heads_up))
                if level and i == m.num_res_blocks:
                    resolution *= 2
                    layers.append(res(ch, ch, up=True) if m.resblock_updown else Resample(ch, m.conv_resample, True, ch))
                self.output_blocks.append(Block(*layers))
                self.output_num_blocks[level] += 1
        out_conv = nn.Conv2d(input_ch, m.out_channels, 3, padding=1)
        self.out = nn.Sequential(_normalization(ch), nn.SiLU(), _zero(out_conv) if zero else out_conv)

    # -------------------------------------------------------------------------------------------- packing
    def _pack(self):
        pk = {"conv": {}, "lin": {}, "emb_off": {}}
        ws, bs, off = [], [], 0
        for mod in self.modules():
            if isinstance(mod, ResBlock):
                pk["emb_off"][id(mod)] = off
                lin = mod.emb_layers[1]
                ws.append(lin.weight.detach().float())
                bs.append(lin.bias.detach().float())
                off += 2 * mod.out_channels
        pk["emb_w"] = torch.cat(ws, 0).contiguous()
        pk["emb_b"] = torch.cat(bs, 0).contiguous()
        return pk

    def _cw(self, pk, conv, split=None):
        key = (id(conv), split)
        if key not in pk["conv"]:
            pk["conv"][key] = (self._pack_conv(conv, split), conv.bias.detach().float().contiguous())
        return pk["conv"][key]

    # -------------------------------------------------------------------------------------------- blocks
    def _resblock(self, mod, x, emb_all, pk, lateral=None):
        x2 = lateral if mod.has_lateral else None
        assert x.C + (x2.C if x2 is not None else 0) == mod.channels
        h = self._gn_act(x, mod.in_layers[0], "silu", x2)
        if mod.up or mod.down:
            assert x2 is None
            h, x = self._box(h, mod.up), self._box(x, mod.up)
        w0, b0 = self._cw(pk, mod.in_layers[2])
        h = self._conv(h, w0, b0, stats=True, normed=True)
        off = pk["emb_off"][id(mod)]
        h = self._gn_act(h, mod.out_layers[0], "silu", mod=emb_all[:, off:off + 2 * mod.out_channels])
        if isinstance(mod.skip_connection, nn.Identity):
            sc = x if x2 is None else self._cat(x, x2)
        else:
            split = x.C if x2 is not None else None
            ws, bsk = self._cw(pk, mod.skip_connection, split)
            if x2 is None:
                sc = self._pointwise(x, ws.view(ws.shape[0], -1), bsk)
            else:
                part = self._pointwise(x, ws[0].view(ws[0].shape[0], -1), bsk)
                sc = self._pointwise(x2, ws[1].view(ws[1].shape[0], -1), None, residual=part.buf)
        w1, b1 = self._cw(pk, mod.out_layers[3])
        return self._conv(h, w1, b1, residual=sc.buf, stats=True, normed=True)

    def _attn(self, mod, x, pk):
        """AttentionBlock._forward (BeatGANsblocks.py:433-443) with QKVAttentionLegacy (:466-491), one head."""
        B, HW, C = x.buf.shape[0], x.H * x.W, x.C
        n = self._gn_act(x, mod.norm, None)
        pairs = self.pairs_admissible(mod.norm, n.norm[1], transform=False)
        key = (id(mod), "qkv")
        if key not in pk["lin"]:
            w = mod.qkv.weight.detach().float().view(3 * C, C)
            b = mod.qkv.bias.detach().float()
            pk["lin"][key] = (w[:2 * C].contiguous(), b[:2 * C].contiguous(), w[2 * C:].contiguous(), b[2 * C:].contiguous(),
                              mod.proj_out.weight.detach().float().view(C, C).contiguous(),
                              mod.proj_out.bias.detach().float().contiguous())
        wqk, bqk, wv, bv, wo, bo = pk["lin"][key]
        dev = x.buf.device
        qk = torch.empty(B * HW, 2 * C, device=dev, dtype=torch.float32)
        _lib.gemm_normed(pk, n.buf.view(-1, C), wqk, qk, epilogue=_lib.make_epilogue(bias=bqk), pairs=pairs)    # n: a GroupNorm's output
        vt = torch.empty(B, C, HW, device=dev, dtype=torch.float32)
        _lib.gemm_weight_times_normed_t(pk, wv, n.buf, vt, B, HW, C, pairs=pairs)
        mixed = torch.empty(B, HW, C, device=dev, dtype=torch.float32)
        if pairs and _lib.attention256_ok(B, HW, C):
            # one launch, the logits never written (csrc/attention.hip); (q * s) . (k * s) with s = ch^-1/4  ==  q . k * ch^-1/2
            skey = (id(mod), "attn_scale")
            if skey not in pk["lin"]:
                gam = float(torch.sqrt((mod.norm.weight.detach().double() ** 2).mean() + (mod.norm.bias.detach().double() ** 2).mean()))
                pk["lin"][skey] = (_lib.pairs_scale_from_rows(wqk, bqk, gam), _lib.pairs_scale_from_rows(wv, bv, gam))
            s_qk, s_v = pk["lin"][skey]
            _lib.attention256(qk, vt, mixed, B, C, s_qk, s_v, float(C) ** (-0.5), bias_v=bv)
            return self._pointwise_pairs(pk, _T(mixed, x.H, x.W, C), wo, bo, s_v, residual=x.buf, stats=True)
        else:
            logits = torch.empty(B, HW, HW, device=dev, dtype=torch.float32)
            _lib.gemm(qk, qk[:, C:], out=logits, M=HW, N=HW, K=C, lda=2 * C, ldb=2 * C, ldc=HW, batch=B,
                      stride_a=HW * 2 * C, stride_b=HW * 2 * C, stride_c=HW * HW)
            # (q * s) . (k * s) with s = ch^-1/4  ==  q . k * ch^-1/2
            _lib.softmax_rows(logits, logits, B * HW, HW, float(C) ** (-0.5))
            _lib.gemm(logits, vt, out=mixed, M=HW, N=C, K=HW, lda=HW, ldb=HW, ldc=C, batch=B, stride_a=HW * HW,
                      stride_b=C * HW, stride_c=HW * C, epilogue=_lib.make_epilogue(bias=bv))  # V bias after P.V: rows of P sum to 1
        return self._pointwise(_T(mixed, x.H, x.W, C), wo, bo, residual=x.buf, stats=True)

    def _resample(self, mod, x, pk):
        if mod.up:
            x = self._box(x, True)
            if mod.use_conv:
                w, b = self._cw(pk, mod.conv)
                x = self._conv(x, w, b)
            return x
        if mod.use_conv:
            w, b = self._cw(pk, mod.op)
            return self._conv(x, w, b, stride=2, pad=1)
        return self._box(x, False)

    def _run_block(self, block, x, emb_all, pk, lateral=None):
        for layer in block:
            if isinstance(layer, ResBlock):
                x = self._resblock(layer, x, emb_all, pk, lateral)
            elif isinstance(layer, AttentionBlock):
                x = self._attn(layer, x, pk)
            elif isinstance(layer, Resample):
                x = self._resample(layer, x, pk)
            elif isinstance(layer, nn.Conv2d):
                w, b = self._cw(pk, layer)
                x = self._conv(x, w, b)
            else:
                raise TypeError(type(layer))
        return x

    # -------------------------------------------------------------------------------------------- forward
    forward_accepts_out = True

    def forward(self, x, t, out_rowscale=None, out=None):
        x, t = self._check_inputs(x, t)
        if x.ndim != 4 or x.shape[1] != self.channels:
            raise RuntimeError(f"BeatGANsUNetModel: expected [B, {self.channels}, H, W], got {tuple(x.shape)}")
        pk = self.packed()
        B, C, H, W = x.shape
        dev = x.device
        temb = torch.empty(B, self.temb_ch, device=dev, dtype=torch.float32)
        _lib.positional_embed(t, temb, B, self.temb_ch, 10000.0, mode=1)
        l0, l2 = self.time_embed[0], self.time_embed[2]
        e1 = _lib.gemm(temb, l0.weight.detach(), epilogue=_lib.make_epilogue(bias=l0.bias.detach(), act="silu"))
        # every block applies SiLU to the embedding first (emb_layers[0]): do it once, then one stacked projection
        e2 = _lib.gemm(e1, l2.weight.detach(), epilogue=_lib.make_epilogue(bias=l2.bias.detach(), act="silu"))
        emb_all = _lib.gemm(e2, pk["emb_w"], epilogue=_lib.make_epilogue(bias=pk["emb_b"]))
        cp = _pad4(C)
        h = _T(torch.empty(B, H * W, cp, device=dev, dtype=torch.float32), H, W, cp)
        _lib.nchw_to_nhwc(x, h.buf, B, C, H * W, cp)
        hs = [[] for _ in self.input_num_blocks]
        k = 0
        for i, nb in enumerate(self.input_num_blocks):
            for _ in range(nb):
                h = self._run_block(self.input_blocks[k], h, emb_all, pk)
                hs[i].append(h)
                k += 1
        h = self._run_block(self.middle_block, h, emb_all, pk)
        k = 0
        for i, nb in enumerate(self.output_num_blocks):
            for _ in range(nb):
                lateral = hs[-i - 1].pop() if hs[-i - 1] else None
                h = self._run_block(self.output_blocks[k], h, emb_all, pk, lateral)
                k += 1
        n = self._gn_act(h, self.out[0], "silu")
        w, b = self._cw(pk, self.out[2])
        h = self._conv(n, w, b, normed=True)
        out = self._out_buffer(out, B, C, H, W, dev)
        _lib.nhwc_to_nchw(h.buf, out, B, C, H * W, h.C, out_rowscale)
        return out
