"""Oracle restatement of BeatGANsUNetModel (CPU, PyTorch fp32).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Follows /root/reference/models/BeatGANsUNET.py:18-285, BeatGANsblocks.py:80-491, BeatGANs_nn.py:23-125 for the
configuration family of configs/.../styleGAN/style_gan_BeatGAN.py:29-82: scale-shift time conditioning, nearest /
average-pool resampling (no FIR), 1-D-conv QKV attention in the legacy head order, no class conditioning.
Module nesting reproduces the reference's ``state_dict`` keys.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def timestep_embedding(t, dim, max_period=10000):
    """BeatGANs_nn.py:107-125: [cos, sin] of t * exp(-log(max_period) * arange(half) / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def normalization(ch):
    return nn.GroupNorm(min(32, ch), ch)          # GroupNorm32, BeatGANs_nn.py:23-25,98-104 (eps default 1e-5)


def zero_(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class Resample(nn.Module):
    """Upsample / Downsample, BeatGANsblocks.py:335-396."""

    def __init__(self, channels, use_conv, up, out_channels=None):
        super().__init__()
        self.up, self.use_conv = up, use_conv
        out_channels = out_channels or channels
        if use_conv and up:
            self.conv = nn.Conv2d(channels, out_channels, 3, padding=1)
        elif use_conv:
            self.op = nn.Conv2d(channels, out_channels, 3, stride=2, padding=1)

    def forward(self, x):
        if self.up:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            return self.conv(x) if self.use_conv else x
        return self.op(x) if self.use_conv else F.avg_pool2d(x, 2, 2)


class ResBlock(nn.Module):
    """BeatGANsblocks.py:80-255 + apply_conditions :258-332 (single condition = the time embedding)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, up=False, down=False, has_lateral=False,
                 use_zero_module=True):
        super().__init__()
        out_channels = out_channels or channels
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.updown, self.has_lateral = up or down, has_lateral
        if self.updown:
            self.h_upd = Resample(channels, False, up)
            self.x_upd = Resample(channels, False, up)
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * out_channels))
        conv = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if use_zero_module:
            conv = zero_(conv)
        self.out_layers = nn.Sequential(normalization(out_channels), nn.SiLU(), nn.Dropout(p=dropout), conv)
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)

    def forward(self, x, emb, lateral=None):
        if self.has_lateral:
            x = torch.cat([x, lateral], dim=1)
        if self.updown:
            h = self.h_upd(self.in_layers[1](self.in_layers[0](x)))
            x = self.x_upd(x)
            h = self.in_layers[2](h)
        else:
            h = self.in_layers(x)
        scale, shift = torch.chunk(self.emb_layers(emb)[:, :, None, None], 2, dim=1)
        h = self.out_layers[0](h) * (1 + scale) + shift
        h = self.out_layers[3](self.out_layers[2](self.out_layers[1](h)))
        return self.skip_connection(x) + h


class AttentionBlock(nn.Module):
    """BeatGANsblocks.py:399-443 with QKVAttentionLegacy (:466-491) or QKVAttention (:498-526)."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, new_order=False):
        super().__init__()
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.norm = normalization(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = zero_(nn.Conv1d(channels, channels, 1))
        self.new_order = new_order

    def forward(self, x):
        b, c, hh, ww = x.shape
        xf = x.reshape(b, c, -1)
        qkv = self.qkv(self.norm(xf))
        n, length = self.num_heads, xf.shape[-1]
        ch = c // n
        if self.new_order:
            q, k, v = (z.reshape(b * n, ch, length) for z in qkv.chunk(3, dim=1))
        else:
            q, k, v = qkv.reshape(b * n, ch * 3, length).split(ch, dim=1)
        s = 1 / math.sqrt(math.sqrt(ch))
        w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), dim=-1)
        a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, length)
        return (xf + self.proj_out(a)).reshape(b, c, hh, ww)


class Block(nn.Sequential):
    """TimestepEmbedSequential, BeatGANsblocks.py:31-43."""

    def forward(self, x, emb, lateral=None):
        for layer in self:
            x = layer(x, emb, lateral) if isinstance(layer, ResBlock) else layer(x)
        return x


class BeatGANsUNetModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        m = config.model
        if m.num_classes is not None or m.resnet_two_cond:
            raise NotImplementedError("class / two-condition variants are not on the manifold_dimension path")
        self.mc = m.model_channels
        self.temb_ch = m.time_embed_channels or m.model_channels
        E = m.embed_channels
        self.time_embed = nn.Sequential(nn.Linear(self.temb_ch, E), nn.SiLU(), nn.Linear(E, E))
        mults = list(m.channel_mult)
        in_mults = list(m.input_channel_mult or m.channel_mult)
        heads_up = m.num_heads if m.num_heads_upsample == -1 else m.num_heads_upsample
        zero = m.resnet_use_zero_module

        def res(ch, out=None, **kw):
            return ResBlock(ch, E, m.dropout, out_channels=out, use_zero_module=zero, **kw)

        def attn(ch, heads):
            return AttentionBlock(ch, heads, m.num_head_channels, m.use_new_attention_order)

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
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_(out_conv) if zero else out_conv)

    def forward(self, x, t):
        hs = [[] for _ in self.input_num_blocks]
        emb = self.time_embed(timestep_embedding(t, self.temb_ch))
        h, k = x, 0
        for i, nb in enumerate(self.input_num_blocks):
            for _ in range(nb):
                h = self.input_blocks[k](h, emb)
                hs[i].append(h)
                k += 1
        h = self.middle_block(h, emb)
        k = 0
        for i, nb in enumerate(self.output_num_blocks):
            for _ in range(nb):
                lateral = hs[-i - 1].pop() if hs[-i - 1] else None
                h = self.output_blocks[k](h, emb, lateral)
                k += 1
        return self.out(h)
