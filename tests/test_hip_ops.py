"""GPU parity of the HIP kernels against the CPU oracle, through the C-ABI (python -m pytest -m gpu)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import id_diff_amd
from helpers import rel_err
from id_diff_amd import _lib, op
from oracle import ops as oops, dim as odim

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_native_library_is_the_one_in_tree():
    import os
    assert os.path.samefile(_lib.library_path(), os.path.join(os.path.dirname(id_diff_amd.__file__), "csrc", "libidiff_hip.so"))
    assert _lib.lib().idiff_abi_version() == 1


# ------------------------------------------------------------------ upfirdn2d
def test_upfirdn2d_golden(golden):
    z = golden("upfirdn2d.npz")
    for i in range(int(z["n_cases"])):
        up, down, p0, p1 = (int(v) for v in z[f"c{i}::params"])
        y = op.upfirdn2d(torch.from_numpy(z[f"c{i}::x"]).to(DEV), torch.from_numpy(z[f"c{i}::k"]).to(DEV), up=up,
                         down=down, pad=(p0, p1))
        ref = torch.from_numpy(z[f"c{i}::y"])
        assert y.shape == ref.shape, i
        torch.testing.assert_close(y.cpu(), ref, rtol=1e-5, atol=1e-6, msg=f"case {i}")
    from id_diff_amd.op.upfirdn2d import upfirdn2d_xy
    ux, uy, dx, dy, px0, px1, py0, py1 = (int(v) for v in z["xy::params"])
    y = upfirdn2d_xy(torch.from_numpy(z["xy::x"]).to(DEV), torch.from_numpy(z["xy::k"]).to(DEV), ux, uy, dx, dy, px0,
                     px1, py0, py1)
    torch.testing.assert_close(y.cpu(), torch.from_numpy(z["xy::y"]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape,up,down,pad", [
    ((128, 128, 32, 32), 1, 2, (1, 1)),   # ncsnpp downsample_2d at BASELINE size
    ((128, 256, 16, 16), 2, 1, (2, 1)),   # upsample_2d
    ((128, 3, 32, 32), 1, 1, (2, 2)),     # FIR before the stride-2 conv
    ((3, 5, 70, 130), 1, 2, (1, 1)),      # several tiles per plane, ragged edges
    ((2, 3, 37, 129), 2, 3, (3, 0)),
])
def test_upfirdn2d_nchw_vs_oracle(shape, up, down, pad):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(*shape, generator=g)
    k = torch.randn(4, 4, generator=g)
    ref = oops.upfirdn2d(x, k, up=up, down=down, pad=pad)
    y = op.upfirdn2d(x.to(DEV), k.to(DEV), up=up, down=down, pad=pad)
    assert rel_err(y.cpu(), ref) < 1e-6


@pytest.mark.parametrize("shape,pad,ksz", [
    ((128, 256, 16, 16), (1, 1), 4),   # the three down-by-2 shapes of SURVEY 8-a5
    ((128, 256, 8, 8), (1, 1), 4),
    ((5, 7, 32, 32), (1, 1), 4),       # fewer planes than a workgroup takes
    ((3, 33, 16, 24), (1, 1), 4),      # non-square, a ragged last workgroup
    ((2, 9, 16, 16), (2, 2), 4),       # wider frame on both sides
    ((2, 9, 16, 16), (3, 1), 4),       # asymmetric pads (even output still)
    ((2, 9, 12, 16), (0, 0), 2),       # 2 x 2 taps, no padding
    ((2, 5, 16, 16), (1, 2), 3),       # 3 x 3 asymmetric taps
])
def test_upfirdn2d_down2_block_kernel_vs_oracle(shape, pad, ksz):
    """upfirdn2d_planes_down2 (2 x 2 output blocks from zero-framed planes in LDS): every geometry that takes it, asymmetric random
    taps (exposes the flip), against the oracle's restatement of upfirdn2d_native (op/upfirdn2d.py:159-200)."""
    g = torch.Generator().manual_seed(sum(shape) + ksz)
    x = torch.randn(*shape, generator=g)
    k = torch.randn(ksz, ksz, generator=g)
    ref = oops.upfirdn2d(x, k, up=1, down=2, pad=pad)
    y = op.upfirdn2d(x.to(DEV), k.to(DEV), up=1, down=2, pad=pad)
    assert y.shape == ref.shape and rel_err(y.cpu(), ref) < 1e-6
    with _lib.thread_option("IDIFF_UFD_ROWS", 1):                  # the per-pixel form it replaced, same answer
        y2 = op.upfirdn2d(x.to(DEV), k.to(DEV), up=1, down=2, pad=pad)
    assert rel_err(y2.cpu(), ref) < 1e-6


@pytest.mark.parametrize("shape,pad,ksz", [
    ((128, 128, 16, 16), (2, 2), 4),   # the plain-FIR shapes of SURVEY 8-a5 (out = in + 1)
    ((128, 256, 8, 8), (2, 2), 4),
    ((128, 3, 32, 32), (2, 2), 4),
    ((3, 33, 16, 24), (2, 2), 4),      # non-square, ragged last group
    ((2, 9, 16, 16), (1, 2), 4),       # out = in: strips end on the plane's edge
    ((2, 9, 12, 16), (0, 0), 3),       # valid convolution with 3 x 3 taps
    ((2, 5, 16, 16), (3, 0), 2),
    ((1, 1, 4, 4), (2, 2), 4),
])
def test_upfirdn2d_fir_strip_kernel_vs_oracle(shape, pad, ksz):
    """upfirdn2d_planes_fir4 (strips of four outputs from zero-framed planes in LDS) against the oracle's restatement of
    upfirdn2d_native (op/upfirdn2d.py:159-200), asymmetric random taps; and the row-walking kernel it replaced."""
    g = torch.Generator().manual_seed(sum(shape) + ksz + 1)
    x = torch.randn(*shape, generator=g)
    k = torch.randn(ksz, ksz, generator=g)
    ref = oops.upfirdn2d(x, k, up=1, down=1, pad=pad)
    y = op.upfirdn2d(x.to(DEV), k.to(DEV), up=1, down=1, pad=pad)
    assert y.shape == ref.shape and rel_err(y.cpu(), ref) < 1e-6
    with _lib.thread_option("IDIFF_UFD_ROWS", 1):
        y2 = op.upfirdn2d(x.to(DEV), k.to(DEV), up=1, down=1, pad=pad)
    assert rel_err(y2.cpu(), ref) < 1e-6


@pytest.mark.parametrize("C", [4, 8, 128, 3])
@pytest.mark.parametrize("mode", [(1, 2, 1, 1), (2, 1, 2, 1), (1, 1, 2, 2)])
def test_upfirdn2d_nhwc_minor(C, mode):
    """minor = C path (NHWC activations) against the NCHW oracle."""
    up, down, p0, p1 = mode
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, C, 12, 10, generator=g)
    k = torch.randn(4, 4, generator=g)
    ref = oops.upfirdn2d(x, k, up=up, down=down, pad=(p0, p1))
    xh = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    oh, ow = ref.shape[2:]
    out = torch.empty(3, oh, ow, C, device=DEV)
    _lib.upfirdn2d_raw(xh, k.to(DEV), out, 3, 12, 10, C, up, up, down, down, p0, p1, p0, p1)
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref) < 1e-6


def test_upfirdn2d_properties_full_size():
    """Size-independent checks at BASELINE size: linearity and the constant-image gain of the normalised FIR."""
    k = torch.tensor(np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0, dtype=torch.float32, device=DEV)
    a = torch.randn(128, 128, 32, 32, device=DEV)
    b = torch.randn(128, 128, 32, 32, device=DEV)
    f = lambda t: op.upfirdn2d(t, k, down=2, pad=(1, 1))
    torch.testing.assert_close(f(a + 2 * b), f(a) + 2 * f(b), rtol=1e-4, atol=1e-5)
    ones = torch.ones(2, 2, 32, 32, device=DEV)
    assert torch.allclose(f(ones)[:, :, 1:-1, 1:-1], torch.ones(2, 2, 14, 14, device=DEV), atol=1e-6)
    up = op.upfirdn2d(ones, k * 4, up=2, pad=(2, 1))
    assert up.shape == (2, 2, 64, 64) and torch.allclose(up[:, :, 2:-2, 2:-2], torch.ones(2, 2, 60, 60, device=DEV), atol=1e-6)


def test_upfirdn2d_errors():
    with pytest.raises(RuntimeError):
        op.upfirdn2d(torch.zeros(1, 1, 2, 2, device=DEV), torch.ones(5, 5, device=DEV))      # kernel > padded input
    with pytest.raises(RuntimeError):
        op.upfirdn2d(torch.zeros(1, 2, 2, device=DEV), torch.ones(2, 2, device=DEV))          # not 4-D


# ------------------------------------------------------------------ fused_bias_act
def test_fused_leaky_relu_golden(golden):
    z = golden("fused_act.npz")
    for i in range(int(z["n_cases"])):
        x, b = torch.from_numpy(z[f"c{i}::x"]).to(DEV), torch.from_numpy(z[f"c{i}::b"]).to(DEV)
        torch.testing.assert_close(op.fused_leaky_relu(x, b).cpu(), torch.from_numpy(z[f"c{i}::y_default"]), rtol=1e-6, atol=1e-7)
        # GPU branch honours negative_slope (the reference's CUDA kernel does; its CPU branch does not)
        y = op.fused_leaky_relu(x, b, negative_slope=0.05, scale=1.25).cpu()
        ref = oops.fused_bias_act_ref(x.cpu(), b.cpu(), None, 3, 0, 0.05, 1.25)
        torch.testing.assert_close(y, ref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("shape", [(128, 128, 32, 32), (128, 256, 4, 4), (7, 5, 3), (4, 6)])
@pytest.mark.parametrize("act,grad", [(3, 0), (3, 1), (3, 2), (1, 0), (1, 1), (1, 2)])
def test_fused_bias_act_modes(shape, act, grad):
    from id_diff_amd.op.fused_act import fused_bias_act
    g = torch.Generator().manual_seed(3)
    x, b, r = torch.randn(*shape, generator=g), torch.randn(shape[1], generator=g), torch.randn(*shape, generator=g)
    ref = oops.fused_bias_act_ref(x, b, r, act, grad, 0.2, 2 ** 0.5)
    y = fused_bias_act(x.to(DEV), b.to(DEV), r.to(DEV), act, grad, 0.2, 2 ** 0.5)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-6, atol=1e-7)
    y0 = fused_bias_act(x.to(DEV), x.new_empty(0).to(DEV), r.to(DEV), act, grad, 0.2, 1.0)   # empty bias, as backward passes it
    torch.testing.assert_close(y0.cpu(), oops.fused_bias_act_ref(x, None, r, act, grad, 0.2, 1.0), rtol=1e-6, atol=1e-7)


def test_fused_module_and_empty_input():
    m = op.FusedLeakyReLU(6).to(DEV)
    x = torch.randn(2, 6, 5, device=DEV)
    torch.testing.assert_close(m(x).cpu(), F.leaky_relu(x.cpu(), 0.2) * 2 ** 0.5, rtol=1e-6, atol=1e-7)
    assert op.fused_leaky_relu(torch.zeros(0, 6, 5, device=DEV), m.bias.detach()).shape == (0, 6, 5)


# ------------------------------------------------------------------ gemm / conv
@pytest.mark.parametrize("M,N,K", [(500, 2048, 104), (2000, 2048, 2048), (128, 512, 256), (37, 100, 2048),
                                   (1, 7, 5), (300, 65, 101), (4096, 128, 1152)])
@pytest.mark.parametrize("act", [None, "elu", "silu"])
def test_gemm_vs_cpu(M, N, K, act):
    g = torch.Generator().manual_seed(M + N + K)
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    ref = a.double() @ w.double().T + b.double()
    ref = {None: lambda v: v, "elu": F.elu, "silu": F.silu}[act](ref)
    y = _lib.gemm(a.to(DEV), w.to(DEV), epilogue=_lib.make_epilogue(bias=b.to(DEV), act=act))
    assert rel_err(y.cpu(), ref) < 2e-6   # fp32 fmaf chain vs fp64


def test_gemm_epilogue_terms():
    g = torch.Generator().manual_seed(5)
    M, N, K, grp = 96, 40, 64, 32
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    rb, res, rsc = torch.randn(M // grp, N, generator=g), torch.randn(M, N, generator=g), torch.rand(M // grp, generator=g) + .5
    ref = ((a @ w.T + rb.repeat_interleave(grp, 0)) + res) * 0.5 * rsc.repeat_interleave(grp)[:, None]
    y = _lib.gemm(a.to(DEV), w.to(DEV), epilogue=_lib.make_epilogue(rowbias=rb.to(DEV), rows_per_group=grp,
                                                                    residual=res.to(DEV), out_scale=0.5, rowscale=rsc.to(DEV)))
    assert rel_err(y.cpu(), ref) < 2e-6


@pytest.mark.parametrize("M,N,K,grp", [(300, 72, 64, 32), (130, 200, 96, 10), (517, 256, 256, 517)])
def test_gemm_writes_only_its_block(M, N, K, grp):
    """C is a column block of a wider, taller buffer and the residual a column block of another: the epilogue's buffer
    descriptors must drop the rows and columns of the last tiles that lie outside [M, N] (M and N are not multiples of
    the 128-row / 128-column tile) and touch nothing else; row groups that are and are not multiples of 32 rows."""
    g = torch.Generator().manual_seed(M + N)
    ldc, ldr = N + 24, N + 40
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ngrp = (M + grp - 1) // grp
    rb, rsc = torch.randn(ngrp, N, generator=g), torch.rand(ngrp, generator=g) + 0.5
    res = torch.randn(M, ldr, generator=g)
    full = torch.full((M + 140, ldc), float("nan"), device=DEV)
    _lib.gemm(a.to(DEV), w.to(DEV), out=full, M=M, N=N, K=K, lda=K, ldb=K, ldc=ldc,
              epilogue=_lib.make_epilogue(bias=b.to(DEV), rowbias=rb.to(DEV), rows_per_group=grp, act="silu",
                                          residual=res.to(DEV), ld_residual=ldr, out_scale=0.75, rowscale=rsc.to(DEV)))
    rows = torch.arange(M) // grp
    ref = (F.silu(a.double() @ w.double().T + b.double() + rb.double()[rows]) + res.double()[:, :N]) * 0.75 * rsc.double()[rows][:, None]
    got = full.cpu()
    assert rel_err(got[:M, :N], ref) < 2e-6
    assert bool(torch.isnan(got[:M, N:]).all()) and bool(torch.isnan(got[M:]).all())


@pytest.mark.parametrize("M,N,K,wscale", [(32768, 256, 256, 1 / 16), (40000, 512, 256, 3e-3), (33000, 128, 128, 40.0), (65536, 256, 32 * 9, 1.0)])
def test_gemm_pairs_vs_cpu(M, N, K, wscale):
    """idiff_gemm_pairs_f32 (fp16 pairs, three matrix instructions per block): operands as the executor hands them over -- A the
    output of a GroupNorm + SiLU (order one, exact zeros, channels of very different gain), the weight of any magnitude (its
    power-of-two scale comes from idiff_gemm_pairs_scale_f32); every epilogue term; M not a multiple of the tile; against fp64 of
    the same fp32 operands, beside the six-product form on the same inputs."""
    assert _lib.gemm_pairs_ok(M, N, K)
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.logspace(-1.5, 0.5, K)[torch.randperm(K, generator=g)]
    a = F.silu(F.group_norm(a.reshape(M // 8, 8, K).permute(0, 2, 1), 32, torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3))
    a = a.permute(0, 2, 1).reshape(M, K).contiguous()
    a[::7, ::5] = 0.0
    w, b = torch.randn(N, K, generator=g) * wscale, torch.randn(N, generator=g)
    grp = 100
    ngrp = (M + grp - 1) // grp
    rb, rsc = torch.randn(ngrp, N, generator=g), torch.rand(ngrp, generator=g) + 0.5
    res = torch.randn(M, N, generator=g)
    ad, wd = a.to(DEV), w.to(DEV)
    sc = _lib.gemm_pairs_scale(wd)
    s0, s1 = sc.cpu().tolist()
    assert s0 * s1 == 1.0 and 2048 <= float(w.abs().max()) * s0 < 4096 and np.log2(s0) == round(np.log2(s0))
    ep = _lib.make_epilogue(bias=b.to(DEV), rowbias=rb.to(DEV), rows_per_group=grp, act="silu", residual=res.to(DEV), out_scale=0.75,
                            rowscale=rsc.to(DEV))
    out = torch.full((M, N), float("nan"), device=DEV)
    _lib.gemm_pairs(ad, wd, sc, out, epilogue=ep)
    rows = torch.arange(M) // grp
    lin = a.double() @ w.double().T
    ref = (F.silu(lin + b.double() + rb.double()[rows]) + res.double()) * 0.75 * rsc.double()[rows][:, None]
    six = _lib.gemm(ad, wd, epilogue=ep)
    e_pairs, e_six = rel_err(out.cpu(), ref), rel_err(six.cpu(), ref)
    assert e_pairs < 2e-6 and e_pairs < 3 * e_six + 1e-7, (e_pairs, e_six)
    # the contraction alone, where no O(1) epilogue term hides its error
    raw = torch.empty(M, N, device=DEV)
    _lib.gemm_pairs(ad, wd, sc, raw)
    assert rel_err(raw.cpu(), lin) < 5e-7
    assert float((raw.cpu().double() - lin).abs().max()) < 1e-5 * float(lin.abs().max())


def test_gemm_pairs_weight_on_the_left_batched():
    """V^T[b] = Wv n[b]^T of an attention block: the weight is the LEFT operand (broadcast over the batch), the activation the
    right one, one 256 x 256 x 256 contraction per image."""
    g = torch.Generator().manual_seed(8)
    B, HW, C = 70, 256, 256
    assert _lib.gemm_pairs_ok(C, HW, C, B)
    n = F.group_norm(torch.randn(B, C, HW, generator=g) * 3 + 1, 32).permute(0, 2, 1).contiguous()        # [B, HW, C]
    wv = torch.randn(C, C, generator=g) * 0.02
    nd, wd = n.to(DEV), wv.to(DEV)
    vt = torch.full((B, C, HW), float("nan"), device=DEV)
    _lib.gemm_weight_times_normed_t({}, wd, nd, vt, B, HW, C)
    ref = torch.einsum("oc,bpc->bop", wv.double(), n.double())
    assert rel_err(vt.cpu(), ref) < 5e-7
    six = torch.empty_like(vt)
    _lib.gemm(wd, nd, out=six, M=C, N=HW, K=C, lda=C, ldb=C, ldc=HW, batch=B, stride_a=0, stride_b=HW * C, stride_c=C * HW)
    assert rel_err(vt.cpu(), ref) < 3 * rel_err(six.cpu(), ref) + 1e-7


@pytest.mark.parametrize("B,C,qk_gain", [(3, 256, 1.0), (5, 128, 1.0), (2, 256, 6.0), (2, 256, 0.05)])
def test_attention256_vs_fp64(B, C, qk_gain):
    """idiff_attention256_f32 (QK^T -> softmax -> PV in one launch, logits on chip, both contractions on fp16 pairs) against fp64 of the same
    fp32 q, k, v, beside the three-launch form on the six-product GEMM.  Operands as the executors produce them: q | k and V^T projected
    from a GroupNorm's output by weights of different gains (qk_gain 6: peaked softmax rows, logits up to +-60; 0.05: nearly uniform rows),
    the operands' power-of-two scales from the projections' row norms (pairs_scale_from_rows).  Reference: layerspp.py:75-91."""
    assert _lib.attention256_ok(B, 256, C)
    g = torch.Generator().manual_seed(int(100 * qk_gain) + C + B)
    HW = 256
    n = F.group_norm(torch.randn(B, C, HW, generator=g) * 3 + 1, 32, torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2)
    n = n.permute(0, 2, 1).contiguous()                                                   # [B, HW, C]
    wqk = torch.randn(2 * C, C, generator=g) * (qk_gain / C ** 0.5)
    bqk = torch.randn(2 * C, generator=g) * 0.1
    wv, bv = torch.randn(C, C, generator=g) * (0.7 / C ** 0.5), torch.randn(C, generator=g) * 0.1
    qk = (n.reshape(-1, C) @ wqk.T + bqk).contiguous()                                    # fp32 operands, as the projection GEMM leaves them
    vt = torch.einsum("oc,bpc->bop", wv, n).contiguous()                                  # [B, C, HW], bias deferred
    scale = float(C) ** -0.5
    q64, k64 = qk[:, :C].double().reshape(B, HW, C), qk[:, C:].double().reshape(B, HW, C)
    logits = torch.einsum("bic,bjc->bij", q64, k64) * scale
    ref = torch.einsum("bij,bcj->bic", torch.softmax(logits, dim=-1), vt.double()) + bv.double()
    qkd, vtd, bvd = qk.to(DEV), vt.to(DEV), bv.to(DEV)
    gam = float(torch.sqrt((n.double() ** 2).mean()))
    s_qk, s_v = _lib.pairs_scale_from_rows(wqk.to(DEV), bqk.to(DEV), gam), _lib.pairs_scale_from_rows(wv.to(DEV), bvd, gam)
    for sc in (s_qk, s_v):
        a, b_ = sc.cpu().tolist()
        assert a * b_ == 1.0 and np.log2(a) == round(np.log2(a))
    assert 0.5 <= float(qk.double().pow(2).mean().sqrt()) * s_qk[0].item() <= 2.0        # the estimate from the weights alone lands near one
    out = torch.full((B * HW, C), float("nan"), device=DEV)
    _lib.attention256(qkd, vtd, out, B, C, s_qk, s_v, scale, bias_v=bvd)
    # the three-launch form on the same operands
    lg = torch.empty(B, HW, HW, device=DEV)
    _lib.gemm(qkd, qkd[:, C:], out=lg, M=HW, N=HW, K=C, lda=2 * C, ldb=2 * C, ldc=HW, batch=B, stride_a=HW * 2 * C, stride_b=HW * 2 * C, stride_c=HW * HW)
    _lib.softmax_rows(lg, lg, B * HW, HW, scale)
    three = torch.empty(B, HW, C, device=DEV)
    _lib.gemm(lg, vtd, out=three, M=HW, N=C, K=HW, lda=HW, ldb=HW, ldc=C, batch=B, stride_a=HW * HW, stride_b=C * HW, stride_c=HW * C,
              epilogue=_lib.make_epilogue(bias=bvd))
    e_fused, e_three = rel_err(out.cpu().reshape(B, HW, C), ref), rel_err(three.cpu(), ref)
    print(f"attention256 B={B} C={C} gain={qk_gain}: fused {e_fused:.2e}, three launches {e_three:.2e}, max |logit| {float(logits.abs().max()):.1f}")
    # (logits of +-60 kept in fp32 cost both forms ~3e-6: 6e-8 x 60; the bar is the three-launch form's own error)
    assert e_fused < max(2e-6, 1.2 * e_three) and e_fused < 3 * e_three + 2e-7, (e_fused, e_three)
    # without the bias: the contraction's own error, no O(1) term beside it
    _lib.attention256(qkd, vtd, out, B, C, s_qk, s_v, scale)
    assert rel_err(out.cpu().reshape(B, HW, C), ref - bv.double()) < 3e-6


def test_attention256_limits():
    """What the one-launch attention refuses (the callers then take the three-launch form), its switches, and its documented failure:
    operands beyond the fp16 range give NaN, never finite wrong numbers."""
    assert _lib.attention256_ok(7, 256, 256) and _lib.attention256_ok(7, 256, 128)
    assert not _lib.attention256_ok(7, 64, 256) and not _lib.attention256_ok(7, 256, 96) and not _lib.attention256_ok(7, 16, 256)
    for name in ("IDIFF_NO_FUSED_ATTN", "IDIFF_NO_PAIRS", "IDIFF_NO_SPLIT"):
        with _lib.thread_option(name, 1):
            assert not _lib.attention256_ok(7, 256, 256)
    B, C = 2, 256
    g = torch.Generator().manual_seed(0)
    qk, vt = torch.randn(B * 256, 2 * C, generator=g).to(DEV), torch.randn(B, C, 256, generator=g).to(DEV)
    one = torch.tensor([1.0, 1.0], device=DEV)
    out = torch.empty(B * 256, C, device=DEV)
    with pytest.raises(RuntimeError, match="shapes"):
        _lib.attention256(qk[:, :C].contiguous(), vt, out, B, C, one, one, 1 / 16)
    _lib.attention256(qk, vt * 1e6, out, B, C, one, one, 1 / 16)                            # s |v| beyond 65504
    assert not bool(torch.isfinite(out).all())
    _lib.attention256(qk, vt * 1e6, out, B, C, one, torch.tensor([2.0 ** -20, 2.0 ** 20], device=DEV), 1 / 16)   # the same values with their scale
    ref = torch.einsum("bij,bcj->bic", torch.softmax(torch.einsum("bic,bjc->bij", qk[:, :C].double().reshape(B, 256, C).cpu(),
                                                                  qk[:, C:].double().reshape(B, 256, C).cpu()) / 16, dim=-1), (vt * 1e6).double().cpu())
    assert rel_err(out.cpu().reshape(B, 256, C), ref) < 3e-6


@pytest.mark.parametrize("mag", [1e-4, 1.0, 3e3])
def test_gemm_pairs_with_activation_scale_two_sources(mag):
    """Operands that are NOT a GroupNorm's output (the residual stream of a U-Net block and its skip connection) at any magnitude:
    the power-of-two scale that idiff_pairs_act_scale_f32 derives from the producers' column sums brings the root mean square of
    cat[x1, x2] to about one, and the pair form keeps 22 bits relative to that scale -- one-source and two-source contraction."""
    g = torch.Generator().manual_seed(5)
    B, HW, C1, C2, N = 160, 256, 128, 128, 256
    M = B * HW
    x1 = (torch.randn(M, C1, generator=g) * mag).to(DEV)
    x2 = (torch.randn(M, C2, generator=g) * mag * 0.3 + 0.1 * mag).to(DEV)
    w = (torch.randn(N, C1 + C2, generator=g) / 16).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV) * mag

    def colsums(x, C):          # what a producing contraction's epilogue leaves: [B, nsplit, C, 2] fp64 per 128-row tile
        t = x.double().reshape(B, HW // 128, 128, C)
        return (torch.stack([t.sum(2), (t * t).sum(2)], -1).contiguous().reshape(-1), HW // 128)
    st1, st2 = colsums(x1, C1), colsums(x2, C2)
    act = _lib.pairs_act_scale(st1, C1, st2, C2, B, HW)
    s0, s1 = act[:2].cpu().tolist()
    rms = float(torch.cat([x1, x2], 1).double().pow(2).mean().sqrt())
    assert s0 * s1 == 1.0 and np.log2(s0) == round(np.log2(s0)) and 0.70 < rms * s0 < 1.42, (s0, rms)
    wsc = _lib.gemm_pairs_scale(w)
    out = torch.full((M, N), float("nan"), device=DEV)
    _lib.gemm_pairs_2src(x1, x2, act, w, wsc, out, epilogue=_lib.make_epilogue(bias=bias))
    ref = torch.cat([x1, x2], 1).double().cpu() @ w.double().cpu().T + bias.double().cpu()
    six = torch.empty(M, N, device=DEV)
    _lib.gemm_2src(x1, x2, w, six, epilogue=_lib.make_epilogue(bias=bias))
    e3, e6 = rel_err(out.cpu(), ref), rel_err(six.cpu(), ref)
    assert e3 < 5e-7 and e3 < 3 * e6 + 1e-7, (e3, e6)
    # one source
    act1 = _lib.pairs_act_scale(st1, C1, None, 0, B, HW)
    w1 = w[:, :C1].contiguous()
    o1 = torch.empty(M, N, device=DEV)
    _lib.gemm_pairs(x1, w1, _lib.gemm_pairs_scale(w1), o1, act_scale=act1)
    assert rel_err(o1.cpu(), x1.double().cpu() @ w1.double().cpu().T) < 5e-7
    if mag == 1.0:
        zero = (torch.zeros(B * 2 * C1 * 2, dtype=torch.float64, device=DEV), 2)
        assert _lib.pairs_act_scale(zero, C1, None, 0, B, HW)[:2].cpu().tolist() == [1.0, 1.0]      # an all-zero tensor: scale 1


def test_gemm_pairs_limits_and_colstats():
    """Beyond fp16's range the high half is +-inf and the outputs NaN (loud, as the fp16-pair convolution); shapes the form does not
    serve are refused; epilogue.colstats has the 128-row tile layout of idiff_gemm_colstats_split."""
    g = torch.Generator().manual_seed(3)
    M, N, K = 256 * 140, 256, 256
    a = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / 16).to(DEV)
    sc = _lib.gemm_pairs_scale(w)
    out = torch.empty(M, N, device=DEV)
    _lib.gemm_pairs(a * 1e5, w, sc, out)
    assert bool(torch.isnan(out).any())
    _lib.gemm_pairs(a * 300, w, sc, out)
    assert bool(torch.isfinite(out).all())
    assert not _lib.gemm_pairs_ok(1000, 256, 256) and not _lib.gemm_pairs_ok(M, 64, 256) and not _lib.gemm_pairs_ok(M, 256, 30)
    assert _lib.gemm_pairs_ok(256, 256, 256, 64) and not _lib.gemm_pairs_ok(256, 256, 256, 63)
    with pytest.raises(RuntimeError, match="not served"):
        _lib.gemm_pairs(a[:1000], w, sc, out[:1000])
    prev = _lib.set_option("IDIFF_NO_PAIRS", 1)
    try:
        assert not _lib.gemm_pairs_ok(M, N, K)
    finally:
        _lib.set_option("IDIFF_NO_PAIRS", prev)
    HW = 256
    ns = _lib.gemm_colstats_split(M, N, K, K, K, HW)
    assert ns == 2
    cs = torch.full((M // HW * ns * N * 2,), float("nan"), device=DEV, dtype=torch.float64)
    _lib.gemm_pairs(a, w, sc, out, epilogue=_lib.make_epilogue(bias=torch.randn(N, generator=g).to(DEV), rows_per_group=HW, colstats=cs))
    ref = out.double().reshape(M // HW, ns, 128, N)
    np.testing.assert_allclose(cs.view(M // HW, ns, N, 2)[..., 0].cpu().numpy(), ref.sum(2).cpu().numpy(), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(cs.view(M // HW, ns, N, 2)[..., 1].cpu().numpy(), (ref * ref).sum(2).cpu().numpy(), rtol=1e-12, atol=1e-9)


def test_gemm_batched_strided():
    g = torch.Generator().manual_seed(6)
    B, HW, C = 3, 64, 16
    qk = torch.randn(B * HW, 2 * C, generator=g)
    ref = torch.einsum("bqc,bkc->bqk", qk[:, :C].reshape(B, HW, C), qk[:, C:].reshape(B, HW, C))
    d = qk.to(DEV)
    out = torch.empty(B, HW, HW, device=DEV)
    _lib.gemm(d, d[:, C:], out=out, M=HW, N=HW, K=C, lda=2 * C, ldb=2 * C, ldc=HW, batch=B, stride_a=HW * 2 * C,
              stride_b=HW * 2 * C, stride_c=HW * HW)
    assert rel_err(out.cpu(), ref) < 2e-6


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad,pad_hi", [
    (4, 32, 32, 128, 128, 3, 1, 1, None), (2, 16, 16, 256, 256, 3, 1, 1, None), (3, 8, 8, 24, 16, 3, 1, 1, None),
    (2, 34, 34, 8, 16, 3, 2, 0, None),    # stride-2 VALID conv behind the FIR (conv_downsample_2d)
    (2, 9, 7, 8, 12, 3, 2, 0, 1),         # F.pad(0,1,0,1) + stride 2
    (2, 32, 32, 4, 128, 3, 1, 1, None),   # stem (3 -> 4 padded channels)
    (3, 33, 33, 4, 128, 3, 2, 0, None),   # input-pyramid stem behind the FIR: 4 channels, stride 2
    (5, 9, 11, 4, 256, 3, 1, 1, None),    # 4 channels, M and N tails
    (2, 32, 32, 128, 3, 3, 1, 1, None),   # head
    (5, 4, 4, 512, 256, 1, 1, 0, None),   # 1x1
])
def test_conv2d_nhwc_vs_cpu(B, H, W, Cin, Cout, k, stride, pad, pad_hi):
    g = torch.Generator().manual_seed(B * H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    xp = F.pad(x, (pad, pad if pad_hi is None else pad_hi, pad, pad if pad_hi is None else pad_hi))
    ref = F.conv2d(xp.double(), w.double(), b.double(), stride=stride)
    OH, OW = ref.shape[2:]
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wt = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.empty(B, OH, OW, Cout, device=DEV)
    _lib.conv2d_nhwc(xd, wt, out, B, H, W, Cin, Cout, k, k, stride, pad, epilogue=_lib.make_epilogue(bias=b.to(DEV)), pad_hi=pad_hi)
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref) < 2e-6


@pytest.mark.parametrize("B,H,W,Cout", [(3, 32, 32, 3), (2, 5, 28, 3), (1, 2, 2, 1), (4, 16, 9, 4), (2, 64, 64, 2)])
def test_conv2d_narrow_head_vs_cpu(B, H, W, Cout):
    """The 128 -> 3 image heads run on the vector ALUs (conv_narrow.hip) instead of a padded MFMA column: against the fp64
    CPU convolution and against the implicit GEMM (IDIFF_NO_PIPE), with the epilogue terms the score function uses
    (bias, activation, out_scale, the per-sample -1/std); widths that are not multiples of the eight-pixel butterfly."""
    g = torch.Generator().manual_seed(B * W + Cout)
    Cin = 128
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    rsc = -(torch.rand(B, generator=g) + 0.5)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wt = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    out = torch.full((B, H, W, Cout), float("nan"), device=DEV)
    _lib.conv2d_nhwc(xd, wt, out, B, H, W, Cin, Cout, 3, 3, 1, 1, epilogue=_lib.make_epilogue(bias=b.to(DEV)))
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref) < 2e-6
    ep = dict(bias=b.to(DEV), act="silu", out_scale=0.5, rowscale=rsc.to(DEV), rows_per_group=H * W)
    _lib.conv2d_nhwc(xd, wt, out, B, H, W, Cin, Cout, 3, 3, 1, 1, epilogue=_lib.make_epilogue(**ep))
    ref2 = F.silu(ref) * 0.5 * rsc.double()[:, None, None, None]
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref2) < 2e-6
    direct = torch.empty_like(out)
    prev = _lib.set_option("IDIFF_NO_PIPE", 1)
    try:
        _lib.conv2d_nhwc(xd, wt, direct, B, H, W, Cin, Cout, 3, 3, 1, 1, epilogue=_lib.make_epilogue(**ep))
    finally:
        _lib.set_option("IDIFF_NO_PIPE", int(prev))
    assert rel_err(out.cpu(), direct.double().cpu()) < 2e-6


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split"])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 32, 32, 128, 128), (5, 16, 16, 256, 64), (7, 8, 8, 64, 128),
                                              (9, 4, 4, 32, 64), (2, 6, 10, 8, 64), (1, 2, 2, 16, 64), (33, 4, 4, 8, 64),
                                              (2, 16, 16, 512, 256), (130, 2, 2, 48, 64)])
def test_conv2d_winograd_vs_cpu(B, H, W, Cin, Cout, split):
    """F(2x2, 3x3) conv against the fp64 CPU convolution, with every epilogue term (per-sample bias, activation,
    residual, scales); tile counts that do not fill a workgroup, maps smaller than a workgroup's 64 tiles.  Both forms of
    the position-wise contractions -- fp32 matrix cores, and operands cut exactly into three bf16 pieces with six partial
    products on the bf16 matrix cores -- are held to the same 3e-6."""
    if split and Cin % 16:
        pytest.skip("the split-precision kernel takes 16 channels per step")
    g = torch.Generator().manual_seed(B * H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    temb = torch.randn(B, Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    rsc = torch.rand(B, generator=g) + 0.5
    assert _lib.conv2d_winograd_ok(B, H, W, Cin, Cout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wt = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    with _lib.thread_option("IDIFF_WINO_SPLIT", 1):        # the split-precision Winograd kernel is opt-in; asked per call
        assert _lib.conv2d_winograd_split_ok(B, H, W, Cin, Cout) == (Cin % 16 == 0)
    assert not _lib.conv2d_winograd_split_ok(B, H, W, Cin, Cout)
    u = _lib.winograd_pack(wt, Cin, Cout, split=split)
    assert u.numel() == (24 if split else 16) * Cin * Cout
    out = torch.empty(B, H, W, Cout, device=DEV)
    with pytest.raises(RuntimeError, match="filter bank"):  # the bank of the other form is refused, never reinterpreted
        _lib.conv2d_winograd(xd, u, out, B, H, W, Cin, Cout, split=not split)
    _lib.conv2d_winograd(xd, u, out, B, H, W, Cin, Cout, epilogue=_lib.make_epilogue(bias=b.to(DEV)), split=split)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref) < 3e-6
    direct = torch.empty_like(out)
    _lib.conv2d_nhwc(xd, wt, direct, B, H, W, Cin, Cout, 3, 3, 1, 1, epilogue=_lib.make_epilogue(bias=b.to(DEV)))
    assert rel_err(out.cpu(), direct.double().cpu()) < 3e-6
    resd = res.permute(0, 2, 3, 1).contiguous().to(DEV)
    _lib.conv2d_winograd(xd, u, out, B, H, W, Cin, Cout, split=split,
                         epilogue=_lib.make_epilogue(bias=b.to(DEV), rowbias=temb.to(DEV), rows_per_group=H * W, act="silu",
                                                     residual=resd, out_scale=0.7071, rowscale=rsc.to(DEV)))
    ref2 = (F.silu(ref + temb.double()[:, :, None, None]) + res.double()) * 0.7071 * rsc.double()[:, None, None, None]
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref2) < 3e-6


def test_conv2d_winograd_splits_inputs_beyond_one_descriptor():
    """Inputs past the 32-bit buffer descriptor are cut into image ranges on the host (winograd.hip); checked on the
    images around the cut with the per-sample epilogue operands that have to be shifted with it."""
    g = torch.Generator(device=DEV).manual_seed(1)
    B, H, Cin, Cout = 4200, 32, 256, 64                      # x = 4.4 GB
    x = torch.randn(B, H * H, Cin, device=DEV, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device=DEV, generator=g) / (9 * Cin) ** 0.5
    u = _lib.winograd_pack(w, Cin, Cout)
    temb = torch.randn(B, Cout, device=DEV, generator=g)
    rsc = torch.rand(B, device=DEV, generator=g) + 0.5
    out = torch.empty(B, H * H, Cout, device=DEV)
    _lib.conv2d_winograd(x, u, out, B, H, H, Cin, Cout, epilogue=_lib.make_epilogue(rowbias=temb, rows_per_group=H * H, rowscale=rsc))
    for b0 in (0, B // 2 - 1, B // 2, B - 1):
        ref = torch.empty(1, H * H, Cout, device=DEV)
        _lib.conv2d_winograd(x[b0:b0 + 1].contiguous(), u, ref, 1, H, H, Cin, Cout,
                             epilogue=_lib.make_epilogue(rowbias=temb[b0:b0 + 1].contiguous(), rows_per_group=H * H,
                                                         rowscale=rsc[b0:b0 + 1].contiguous()))
        assert torch.equal(out[b0:b0 + 1], ref), b0
    del x, out
    torch.cuda.empty_cache()


def test_conv2d_winograd_row_groups_and_residual_pitch():
    """Epilogue row groups that are not whole images (rows_per_group = one map row) and a residual that is a column
    slice of a wider tensor (ld_residual > Cout): the tail's cold path and its second buffer descriptor."""
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout = 3, 6, 8, 16, 64
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    rowb = torch.randn(B * H, Cout, generator=g)             # one bias row per map row
    rsc = torch.rand(B * H, generator=g) + 0.5
    wide = torch.randn(B, H, W, Cout + 32, generator=g)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    u = _lib.winograd_pack(w.permute(0, 2, 3, 1).contiguous().to(DEV), Cin, Cout)
    wided = wide.to(DEV)
    out = torch.empty(B, H, W, Cout, device=DEV)
    _lib.conv2d_winograd(xd, u, out, B, H, W, Cin, Cout,
                         epilogue=_lib.make_epilogue(bias=b.to(DEV), rowbias=rowb.to(DEV), rows_per_group=W, act="lrelu",
                                                     residual=wided, ld_residual=Cout + 32, out_scale=1.25, rowscale=rsc.to(DEV)))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)        # [B, H, W, Cout]
    ref = F.leaky_relu(ref + rowb.double().view(B, H, 1, Cout), 0.2) + wide.double()[..., :Cout]
    ref = ref * 1.25 * rsc.double().view(B, H, 1, 1)
    assert rel_err(out.cpu(), ref) < 3e-6


def test_conv2d_winograd_splits_outputs_beyond_one_descriptor():
    """The output (and a residual of the same extent) passes 4 GB while the input does not: the host cuts the batch on
    that extent too (the kernel stores through a 32-bit buffer descriptor)."""
    g = torch.Generator(device=DEV).manual_seed(2)
    B, H, Cin, Cout = 4200, 32, 8, 256                       # out = 4.4 GB, x = 0.14 GB
    x = torch.randn(B, H * H, Cin, device=DEV, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device=DEV, generator=g) / (9 * Cin) ** 0.5
    u = _lib.winograd_pack(w, Cin, Cout)
    res = torch.randn(B, H * H, Cout, device=DEV, generator=g)
    out = torch.full((B, H * H, Cout), float("nan"), device=DEV)
    _lib.conv2d_winograd(x, u, out, B, H, H, Cin, Cout, epilogue=_lib.make_epilogue(residual=res, out_scale=0.5))
    for b0 in (0, B // 2 - 1, B // 2, B - 1):
        ref = torch.empty(1, H * H, Cout, device=DEV)
        _lib.conv2d_winograd(x[b0:b0 + 1].contiguous(), u, ref, 1, H, H, Cin, Cout,
                             epilogue=_lib.make_epilogue(residual=res[b0:b0 + 1].contiguous(), out_scale=0.5))
        assert torch.equal(out[b0:b0 + 1], ref), b0
    assert not bool(torch.isnan(out[::97]).any())
    del x, out, res
    torch.cuda.empty_cache()


def test_conv2d_winograd_rejects_what_it_cannot_take():
    assert not _lib.conv2d_winograd_ok(2, 7, 8, 32, 64)      # odd height
    assert not _lib.conv2d_winograd_ok(2, 8, 8, 4, 64)       # Cin % 8
    assert not _lib.conv2d_winograd_ok(2, 8, 8, 32, 3)       # Cout % 64
    x = torch.zeros(2, 7, 8, 32, device=DEV)
    with pytest.raises(RuntimeError, match="not supported"):
        _lib.conv2d_winograd(x, torch.zeros(16 * 32 * 64, device=DEV), torch.zeros(2, 7, 8, 64, device=DEV), 2, 7, 8, 32, 64)


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split"])
@pytest.mark.parametrize("B,H,Cin,Cout", [(3, 32, 64, 128), (6, 16, 128, 64), (5, 8, 64, 128), (11, 4, 32, 64), (64, 4, 8, 64)])
def test_winograd_colstats_feed_groupnorm(B, H, Cin, Cout, split):
    """Column sums from the conv epilogue = a statistics pass over its output; maps of 8x8 and 4x4 (several whole samples per
    workgroup, one slot per sample) with sample counts that leave the last workgroup partly empty."""
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, H * H, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    ns = _lib.conv2d_winograd_colstats_split(B, H, H, Cin, Cout)
    assert ns == max(1, H * H // 128)                   # 32 output tiles (of 2x2 pixels) per workgroup
    cs = torch.empty(B * ns * Cout * 2, device=DEV, dtype=torch.float64)
    out = torch.empty(B, H * H, Cout, device=DEV)
    if split and Cin % 16:
        pytest.skip("the split-precision kernel takes 16 channels per step")
    u = _lib.winograd_pack(w, Cin, Cout, split=split)
    assert u.numel() == (24 if split else 16) * Cin * Cout
    _lib.conv2d_winograd(x, u, out, B, H, H, Cin, Cout, split=split,
                         epilogue=_lib.make_epilogue(bias=bias, act="silu", rows_per_group=H * H, colstats=cs))
    G = 32
    st_a, st_b = torch.empty(B * G * 2, device=DEV), torch.empty(B * G * 2, device=DEV)
    _lib.groupnorm_finalize(cs, ns, Cout, None, 0, 0, B, H * H, G, 1e-6, st_a)
    nsp = _lib.groupnorm_nsplit(B, H * H, Cout)
    ws = torch.empty(B * nsp * Cout * 2, device=DEV, dtype=torch.float64)
    _lib.groupnorm_stats(out, Cout, None, 0, B, H * H, G, 1e-6, ws, st_b)
    torch.testing.assert_close(st_a, st_b, rtol=1e-6, atol=1e-7)
    tot = cs.view(B, ns, Cout, 2).sum(1)
    np.testing.assert_allclose(tot[..., 0].cpu().numpy(), out.double().sum(1).cpu().numpy(), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(tot[..., 1].cpu().numpy(), (out.double() ** 2).sum(1).cpu().numpy(), rtol=1e-12, atol=1e-9)
    assert _lib.conv2d_winograd_colstats_split(64, 2, 2, Cin, Cout) == 0     # one tile per sample: two samples per thread
    assert _lib.conv2d_winograd_colstats_split(4, 6, 10, Cin, Cout) == 0      # 15 tiles per sample


@pytest.mark.parametrize("B,HW,C,C2,G,ns1,ns2", [(5, 256, 128, 0, 32, 2, 0), (3, 64, 256, 128, 32, 1, 1), (4, 16, 64, 64, 32, 1, 1),
                                                  (2, 1024, 128, 128, 32, 8, 4), (3, 64, 24, 0, 6, 1, 0)])
def test_groupnorm_apply_from_column_sums(B, HW, C, C2, G, ns1, ns2):
    """idiff_groupnorm_apply_colstats_f32 = idiff_groupnorm_finalize_f32 + idiff_groupnorm_apply_f32, bit for bit (one and
    two sources, groups that straddle the sources, several row tiles per sample), and both = torch's group_norm."""
    g = torch.Generator().manual_seed(B * HW + C)
    x = torch.randn(B, HW, C, generator=g) * 2 + 0.5
    x2 = torch.randn(B, HW, C2, generator=g) - 0.3 if C2 else None
    gamma, beta = torch.randn(C + C2, generator=g), torch.randn(C + C2, generator=g)

    def colsums(t, ns):       # [B, ns, C, 2]: sums and sums of squares over ns equal row ranges
        parts = t.double().view(B, ns, HW // ns, t.shape[-1])
        return torch.stack([parts.sum(2), (parts ** 2).sum(2)], -1).contiguous().to(DEV)
    ws1 = colsums(x, ns1)
    ws2 = colsums(x2, ns2) if C2 else None
    xd, x2d = x.to(DEV), (x2.to(DEV) if C2 else None)
    ga, be = gamma.to(DEV), beta.to(DEV)
    stats = torch.empty(B * G * 2, device=DEV)
    _lib.groupnorm_finalize(ws1, ns1, C, ws2, ns2, C2, B, HW, G, 1e-6, stats)
    y_two = torch.empty(B, HW, C + C2, device=DEV)
    _lib.groupnorm_apply(xd, C, x2d, C2, B, HW, G, stats, ga, be, "silu", y_two)
    y_one = torch.empty_like(y_two)
    _lib.groupnorm_apply_colstats(xd, C, x2d, C2, B, HW, G, ws1, ns1, ws2, ns2, 1e-6, ga, be, "silu", y_one)
    assert torch.equal(y_one, y_two)
    full = torch.cat([x, x2], -1) if C2 else x
    ref = F.silu(F.group_norm(full.double().permute(0, 2, 1), G, gamma.double(), beta.double(), 1e-6)).permute(0, 2, 1)
    assert rel_err(y_one.cpu(), ref) < 3e-6


@pytest.mark.parametrize("M,K1,K2,N", [(1000, 128, 128, 128), (4100, 256, 256, 64), (77, 32, 32, 200)])
def test_gemm_two_sources(M, K1, K2, N):
    """[A1 | A2] @ W^T with the concatenation never formed (the split shortcut of the up path)."""
    g = torch.Generator().manual_seed(M)
    a1, a2 = torch.randn(M, K1, generator=g), torch.randn(M, K2, generator=g)
    w = torch.randn(N, K1 + K2, generator=g) / (K1 + K2) ** 0.5
    b = torch.randn(N, generator=g)
    out = torch.empty(M, N, device=DEV)
    _lib.gemm_2src(a1.to(DEV), a2.to(DEV), w.to(DEV), out, epilogue=_lib.make_epilogue(bias=b.to(DEV)))
    ref = torch.cat([a1, a2], 1).double() @ w.double().T + b.double()
    assert rel_err(out.cpu(), ref) < 2e-6
    with pytest.raises(RuntimeError, match="multiple of 32"):
        _lib.gemm_2src(torch.zeros(8, 16, device=DEV), torch.zeros(8, 16, device=DEV), torch.zeros(4, 32, device=DEV),
                       torch.zeros(8, 4, device=DEV))


def test_gemm_two_sources_beyond_4gib():
    """Both sources past one buffer descriptor (a 4480-row launch set at 32x32x256 is 4.7 GB): rows are cut on the host."""
    g = torch.Generator(device=DEV).manual_seed(2)
    M, K1, N, grp = 4_300_000, 256, 64, 1000
    a1 = torch.randn(M, K1, device=DEV, generator=g)
    a2 = torch.randn(M, K1, device=DEV, generator=g)
    w = torch.randn(N, 2 * K1, device=DEV, generator=g) / 23
    rb = torch.randn(M // grp, N, device=DEV, generator=g)
    out = torch.empty(M, N, device=DEV)
    _lib.gemm_2src(a1, a2, w, out, epilogue=_lib.make_epilogue(rowbias=rb, rows_per_group=grp))
    rows = torch.tensor([0, 1, M // 2 - 1, M // 2, M // 2 + 1, 3_000_000, M - 1], device=DEV)
    ref = torch.cat([a1[rows], a2[rows]], 1).double().cpu() @ w.double().cpu().T + rb[rows // grp].double().cpu()
    assert rel_err(out[rows].cpu(), ref) < 2e-6
    del a1, a2, out
    torch.cuda.empty_cache()


def test_operands_beyond_4gib_are_split_on_the_host():
    """The fast kernel addresses operands through 32-bit-offset buffer descriptors; larger problems are cut into
    row / image ranges (igemm.hip: shift_epilogue).  Checked on the rows / images around the cut."""
    g = torch.Generator(device=DEV).manual_seed(0)
    M, K, N, grp = 1_200_000, 1024, 64, 1000           # A = 4.9 GB
    a = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / 32
    rb = torch.randn(M // grp, N, device=DEV, generator=g)
    rsc = torch.rand(M // grp, device=DEV, generator=g) + 0.5
    y = _lib.gemm(a, w, epilogue=_lib.make_epilogue(rowbias=rb, rows_per_group=grp, rowscale=rsc))
    rows = torch.tensor([0, 1, 599_999, 600_000, 600_001, 999_999, M - 1], device=DEV)
    ref = (a[rows].double().cpu() @ w.double().cpu().T + rb[rows // grp].double().cpu()) * rsc[rows // grp].double().cpu()[:, None]
    assert rel_err(y[rows].cpu(), ref) < 2e-6
    del a, y
    B, H, Cin, Cout = 2100, 32, 512, 32                # x = 4.4 GB
    x = torch.randn(B, H * H, Cin, device=DEV, generator=g)
    wt = torch.randn(Cout, 3, 3, Cin, device=DEV, generator=g) / 68
    res = torch.randn(B, H * H, Cout, device=DEV, generator=g)
    out = torch.empty(B, H * H, Cout, device=DEV)
    _lib.conv2d_nhwc(x, wt, out, B, H, H, Cin, Cout, 3, 3, 1, 1, epilogue=_lib.make_epilogue(residual=res, out_scale=0.5,
                                                                                             rows_per_group=H * H))
    for b in (0, 1049, 1050, B - 1):
        xi = x[b].reshape(H, H, Cin).permute(2, 0, 1)[None].double().cpu()
        ref = F.conv2d(xi, wt.permute(0, 3, 1, 2).double().cpu(), padding=1)[0].permute(1, 2, 0).reshape(H * H, Cout)
        ref = (ref + res[b].double().cpu()) * 0.5
        assert rel_err(out[b].cpu(), ref) < 2e-6, b


# ------------------------------------------------------------------ norm / pointwise
@pytest.mark.parametrize("B,HW,C,C2,G", [(4, 1024, 128, 0, 32), (3, 256, 256, 128, 32), (2, 64, 8, 0, 2), (2, 16, 24, 0, 6),
                                         (5, 1024, 16, 8, 6)])
@pytest.mark.parametrize("act", [None, "silu"])
def test_groupnorm_vs_cpu(B, HW, C, C2, G, act):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, HW, C, generator=g) * 2 + 0.7
    x2 = torch.randn(B, HW, C2, generator=g) if C2 else None
    gamma, beta = torch.randn(C + C2, generator=g), torch.randn(C + C2, generator=g)
    full = x if x2 is None else torch.cat([x, x2], -1)
    ref = F.group_norm(full.permute(0, 2, 1).double(), G, gamma.double(), beta.double(), eps=1e-6).permute(0, 2, 1)
    if act:
        ref = F.silu(ref)
    nsplit = _lib.groupnorm_nsplit(B, HW, C + C2)
    ws = torch.empty(B * nsplit * (C + C2) * 2, device=DEV, dtype=torch.float64)
    stats = torch.empty(B * G * 2, device=DEV)
    xd, x2d = x.to(DEV), (x2.to(DEV) if x2 is not None else None)
    _lib.groupnorm_stats(xd, C, x2d, C2, B, HW, G, 1e-6, ws, stats)
    y = torch.empty(B, HW, C + C2, device=DEV)
    _lib.groupnorm_apply(xd, C, x2d, C2, B, HW, G, stats, gamma.to(DEV), beta.to(DEV), act, y)
    assert rel_err(y.cpu(), ref) < 2e-6


@pytest.mark.parametrize("rows,cols", [(1000, 256), (64, 16), (10, 1024), (5, 1500), (3, 1), (7, 512), (70001, 64), (33, 260)])
def test_softmax_rows(rows, cols):
    x = torch.randn(rows, cols, generator=torch.Generator().manual_seed(cols)) * 5
    d = x.to(DEV)
    _lib.softmax_rows(d, d, rows, cols, 0.25)
    assert rel_err(d.cpu(), F.softmax(x.double() * 0.25, -1)) < 2e-6


def test_pointwise_and_layout():
    g = torch.Generator().manual_seed(11)
    x = torch.rand(3, 3, 8, 8, generator=g)
    y = torch.empty(3, 64, 4, device=DEV)
    _lib.nchw_to_nhwc(x.to(DEV), y, 3, 3, 64, 4, 2.0, -1.0)
    ref = torch.cat([(2 * x - 1).reshape(3, 3, 64).permute(0, 2, 1), torch.zeros(3, 64, 1)], -1)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-6, atol=1e-7)
    back = torch.empty(3, 3, 8, 8, device=DEV)
    rsc = torch.tensor([1.0, -2.0, 0.5], device=DEV)
    _lib.nhwc_to_nchw(y, back, 3, 3, 64, 4, rsc)
    torch.testing.assert_close(back.cpu(), (2 * x - 1) * rsc.cpu()[:, None, None, None], rtol=1e-6, atol=1e-7)
    t = torch.tensor([1e-5 * 999, 0.2 * 999, 0.9 * 999])
    W = torch.randn(16, generator=g) * 16
    out = torch.empty(3, 32, device=DEV)
    _lib.fourier_embed(t.to(DEV), W.to(DEV), out, 3, 16)
    proj = t[:, None] * W[None, :] * 2 * np.pi
    torch.testing.assert_close(out.cpu(), torch.cat([proj.sin(), proj.cos()], -1), rtol=1e-5, atol=2e-6)
    a = torch.randn(2, 6, 6, 8, generator=g)
    up, dn = torch.empty(2, 12, 12, 8, device=DEV), torch.empty(2, 3, 3, 8, device=DEV)
    _lib.resample2x_nhwc(a.to(DEV), up, 2, 6, 6, 8, 1)
    _lib.resample2x_nhwc(a.to(DEV), dn, 2, 6, 6, 8, 0)
    torch.testing.assert_close(up.cpu(), a.repeat_interleave(2, 1).repeat_interleave(2, 2))
    torch.testing.assert_close(dn.cpu(), F.avg_pool2d(a.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1), rtol=1e-6, atol=1e-7)


def test_perturb_randn_stream():
    """In-kernel Philox noise: N(0, 1) moments, reproducible, independent of how rows are cut into launches."""
    D, rows = 3072, 1024
    x = torch.rand(D, device=DEV)
    std = torch.full((rows,), 0.01, device=DEV)
    out, z = torch.empty(rows, D, device=DEV), torch.empty(rows, D, device=DEV)
    _lib.perturb_randn(x, std, None, out, rows, D, 0, 12345, z)
    zc = z.double().cpu()
    assert abs(float(zc.mean())) < 3e-3 and abs(float(zc.var()) - 1.0) < 5e-3
    assert abs(float((zc ** 4).mean()) - 3.0) < 0.05                      # Gaussian kurtosis
    assert abs(float((zc[:, :-1] * zc[:, 1:]).mean())) < 3e-3            # neighbours uncorrelated
    assert abs(float((zc[:-1] * zc[1:]).mean())) < 3e-3
    torch.testing.assert_close(out.cpu(), (x[None] + 0.01 * z).cpu(), rtol=1e-6, atol=1e-7)
    again = torch.empty_like(out)
    _lib.perturb_randn(x, std, None, again, rows, D, 0, 12345)
    assert torch.equal(out, again)
    # second half generated as its own launch with row0 = 512 equals rows 512.. of the full launch
    half = torch.empty(512, D, device=DEV)
    _lib.perturb_randn(x, std[:512].contiguous(), None, half, 512, D, 512, 12345)
    assert torch.equal(half, out[512:])
    other = torch.empty_like(out)
    _lib.perturb_randn(x, std, None, other, rows, D, 0, 12346)
    assert not torch.equal(other, out)
    coeff = torch.full((rows,), 0.5, device=DEV)
    _lib.perturb_randn(x, std, coeff, other, rows, D, 0, 12345)
    torch.testing.assert_close(other.cpu(), (0.5 * x[None] + 0.01 * z).cpu(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,H,Cin,Cout", [(8, 32, 128, 128), (16, 16, 256, 256), (40, 16, 64, 96), (8, 32, 4, 128)])
def test_fused_colstats_feed_groupnorm(B, H, Cin, Cout):
    """epilogue.colstats: per-tile column sums written by the conv + idiff_groupnorm_finalize_f32 give the same
    GroupNorm statistics as the stand-alone statistics pass over the stored tensor."""
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, H * H, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    ns = _lib.conv2d_colstats_split(B, H, H, Cin, Cout, 3, 3, 1, 1)
    assert ns == H * H // 128
    cs = torch.empty(B * ns * Cout * 2, device=DEV, dtype=torch.float64)
    out = torch.empty(B, H * H, Cout, device=DEV)
    _lib.conv2d_nhwc(x, w, out, B, H, H, Cin, Cout, 3, 3, 1, 1,
                     epilogue=_lib.make_epilogue(bias=bias, act="silu", rows_per_group=H * H, colstats=cs))
    ref = out.double().reshape(B, ns, 128, Cout)
    np.testing.assert_allclose(cs.view(B, ns, Cout, 2)[..., 0].cpu().numpy(), ref.sum(2).cpu().numpy(), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(cs.view(B, ns, Cout, 2)[..., 1].cpu().numpy(), (ref * ref).sum(2).cpu().numpy(), rtol=1e-12, atol=1e-9)
    G = 32
    st_a, st_b = torch.empty(B * G * 2, device=DEV), torch.empty(B * G * 2, device=DEV)
    _lib.groupnorm_finalize(cs, ns, Cout, None, 0, 0, B, H * H, G, 1e-6, st_a)
    nsp = _lib.groupnorm_nsplit(B, H * H, Cout)
    ws = torch.empty(B * nsp * Cout * 2, device=DEV, dtype=torch.float64)
    _lib.groupnorm_stats(out, Cout, None, 0, B, H * H, G, 1e-6, ws, st_b)
    torch.testing.assert_close(st_a, st_b, rtol=1e-6, atol=1e-7)
    # samples smaller than a row tile: not available, the caller keeps the stand-alone pass
    assert _lib.conv2d_colstats_split(4096, 8, 8, Cin, Cout, 3, 3, 1, 1) == 0   # 64-row samples inside 128-row tiles
    assert _lib.conv2d_colstats_split(B, H, H, 12, Cout, 3, 3, 1, 1) == 0    # general kernel (Cin % 32 != 0 and not a 4-channel stem)


@pytest.mark.parametrize("up,down,pad", [(1, 2, (1, 1)), (2, 1, (2, 1)), (1, 1, (2, 2)), (2, 3, (3, 0)), (1, 1, (0, 0))])
def test_upfirdn2d_backward_and_double_backward(up, down, pad):
    """Derivatives run the same HIP kernel (op/upfirdn2d.py:19-142); checked against autograd through the oracle."""
    g = torch.Generator().manual_seed(up * 10 + down)
    x = torch.randn(2, 3, 9, 10, generator=g)
    k = torch.randn(4, 4, generator=g)
    xc = x.clone().requires_grad_(True)
    yc = oops.upfirdn2d(xc, k, up=up, down=down, pad=pad)
    go = torch.randn(yc.shape, generator=g)
    gi_ref, = torch.autograd.grad(yc, xc, go, create_graph=True)
    probe = torch.randn(x.shape, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    yd = op.upfirdn2d(xd, k.to(DEV), up=up, down=down, pad=pad)
    god = go.to(DEV).requires_grad_(True)
    gi, = torch.autograd.grad(yd, xd, god, create_graph=True)
    assert rel_err(gi.detach().cpu(), gi_ref.detach()) < 1e-6
    # double backward: d(gi . probe)/d(grad_output) equals the forward op applied to probe
    gg, = torch.autograd.grad(gi, god, probe.to(DEV))
    assert rel_err(gg.cpu(), oops.upfirdn2d(probe, k, up=up, down=down, pad=pad)) < 1e-6


def test_fused_leaky_relu_backward():
    g = torch.Generator().manual_seed(8)
    x, b = torch.randn(3, 6, 5, 4, generator=g), torch.randn(6, generator=g)
    xc, bc = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yc = F.leaky_relu(xc + bc.view(1, -1, 1, 1), 0.1) * 1.3
    go = torch.randn(yc.shape, generator=g)
    gx_ref, gb_ref = torch.autograd.grad(yc, (xc, bc), go)
    xd, bd = x.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    yd = op.fused_leaky_relu(xd, bd, negative_slope=0.1, scale=1.3)
    gx, gb = torch.autograd.grad(yd, (xd, bd), go.to(DEV))
    torch.testing.assert_close(gx.cpu(), gx_ref, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(gb.cpu(), gb_ref, rtol=1e-5, atol=1e-5)
    m = op.FusedLeakyReLU(6).to(DEV)
    m(xd).sum().backward()
    assert m.bias.grad is not None and m.bias.grad.shape == (6,)


# ---- the other dtypes of the reference's native-op dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF), round 4
@pytest.mark.parametrize("name,dtype", [("f64", torch.float64), ("f16", torch.float16)])
def test_native_ops_other_dtypes_golden(golden, name, dtype):
    """op.upfirdn2d / op.fused_leaky_relu on float64 and float16 tensors against the REFERENCE's CPU branches run in that
    dtype (tests/golden/ops_dtypes.npz).  upfirdn2d: fp64 to rounding; fp16 = fp32 accumulation rounded once, the CPU path's
    arithmetic, equal to the last half ulp (1e-3).  fused_leaky_relu follows the reference's CUDA kernel (alpha and scale
    reach it as `float` and are converted to the tensor's dtype, op/fused_bias_act_kernel.cu:19,80-93): BIT-equal to the numpy
    restatement of that arithmetic (oracle.ops.fused_bias_act_native), and within the rounding of alpha of the CPU branch."""
    z = golden("ops_dtypes.npz")
    for i in range(int(z["n_cases"])):
        up, down, p0, p1 = [int(v) for v in z[f"ufd::c{i}::params"]]
        x, k = torch.from_numpy(z[f"ufd::{name}::c{i}::x"]), torch.from_numpy(z[f"ufd::{name}::c{i}::k"])
        y = op.upfirdn2d(x.to(DEV), k.to(DEV), up=up, down=down, pad=(p0, p1))
        ref = torch.from_numpy(z[f"ufd::{name}::c{i}::y"])
        assert y.dtype == dtype and y.shape == ref.shape
        if dtype == torch.float64:
            torch.testing.assert_close(y.cpu(), ref, rtol=1e-13, atol=1e-14)
        else:
            torch.testing.assert_close(y.cpu().float(), ref.float(), rtol=1e-3, atol=1e-3 * float(ref.float().abs().max()) * 0.5)
    for i in range(int(z["fba::n_cases"])):
        x, b = torch.from_numpy(z[f"fba::{name}::c{i}::x"]), torch.from_numpy(z[f"fba::{name}::c{i}::b"])
        y = op.fused_leaky_relu(x.to(DEV), b.to(DEV), 0.2, 1.25)
        assert y.dtype == dtype
        nat = oops.fused_bias_act_native(x, b, None, 3, 0, 0.2, 1.25)
        assert np.array_equal(y.cpu().numpy(), nat), f"{name} case {i}: not the CUDA kernel's arithmetic"
        tol = 2e-8 if dtype == torch.float64 else 2.5e-3
        np.testing.assert_allclose(y.cpu().numpy().astype(np.float64), z[f"fba::{name}::c{i}::y_scale1.25"].astype(np.float64),
                                   rtol=tol, atol=tol * 1e-3)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16])
@pytest.mark.parametrize("shape,act,grad", [((3, 8, 5, 8), 3, 0), ((3, 8, 5, 8), 3, 1), ((2, 7, 3), 3, 0), ((5, 6), 1, 0), ((3, 8, 16), 3, 2),
                                            ((128, 128, 32, 32), 3, 0)])
def test_fused_bias_act_other_dtypes_every_mode(dtype, shape, act, grad):
    """Every act/grad mode, vector (16 bytes per lane) and scalar geometries, and the SURVEY 8(a7) activation size, bit for
    bit against the numpy restatement of the reference kernel's scalar_t arithmetic."""
    g = torch.Generator().manual_seed(len(shape) * 10 + act + grad)
    x = (torch.randn(*shape, generator=g) * 2).to(dtype)
    b = torch.randn(shape[1], generator=g).to(dtype)
    ref = torch.randn(*shape, generator=g).to(dtype)
    y = _lib.fused_bias_act(x.to(DEV), b.to(DEV), ref.to(DEV) if grad == 1 else None, act, grad, 0.17, 1.3)
    want = oops.fused_bias_act_native(x, b, ref if grad == 1 else None, act, grad, 0.17, 1.3)
    assert y.dtype == dtype and np.array_equal(y.cpu().numpy(), want)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16])
@pytest.mark.parametrize("shape,up,down,pad", [((128, 128, 32, 32), 1, 2, (1, 1)), ((16, 256, 16, 16), 2, 1, (2, 1)),
                                               ((3, 5, 70, 130), 1, 2, (1, 1)), ((2, 3, 37, 129), 2, 3, (3, 0))])
def test_upfirdn2d_other_dtypes_vs_oracle(dtype, shape, up, down, pad):
    """SURVEY 8(a5) shape families and ragged planes in float64 / float16 against the oracle (the reference's CPU recipe in the
    same dtype); wrong dtypes are refused, mixed dtypes of input and FIR kernel follow the input as in the reference."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g).to(dtype)
    k = torch.randn(4, 4, generator=g).to(dtype)
    if dtype == torch.float16 and shape[0] * shape[1] > 4096:
        x = x[:8]                                   # the CPU half convolution of the oracle is slow: a slice is enough
    ref = oops.upfirdn2d(x, k, up=up, down=down, pad=pad)
    y = op.upfirdn2d(x.to(DEV), k.float().to(DEV), up=up, down=down, pad=pad)      # the kernel is converted to the input's dtype
    assert y.dtype == dtype
    if dtype == torch.float64:
        assert rel_err(y.cpu(), ref) < 1e-14
    else:
        torch.testing.assert_close(y.cpu().float(), ref.float(), rtol=2e-3, atol=2e-3 * float(ref.float().abs().max()))
    with pytest.raises(RuntimeError, match="not one of float32"):
        op.upfirdn2d(x.to(DEV).to(torch.bfloat16), k.to(DEV), up=up, down=down, pad=pad)


# ---- Winograd F(4x4, 3x3) (round 4)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 32, 32, 128, 128), (3, 8, 8, 256, 64), (5, 8, 12, 24, 64), (2, 16, 16, 512, 256), (130, 4, 4, 48, 64),
                                            (1, 64, 64, 16, 64), (2, 8, 8, 8, 64), (33, 8, 8, 256, 256)])
def test_conv2d_winograd43_vs_cpu(B, H, W, Cin, Cout):
    """F(4x4, 3x3) conv (points 0, +-1/2, +-2, infinity; fp32 transforms, fp32 MFMA contraction) against the fp64 CPU convolution with
    every epilogue term; tile counts that do not fill a workgroup's 32 tiles, maps of one tile per sample, several output-channel
    tiles, 1 to 64 K steps.  Bar 3e-6, the bar of the 2x2 form: the 4x4 transforms cost a factor ~4 in rounding (measured
    0.4-2.0e-6 here, 1-5e-7 for the 2x2 form), still inside it; against the direct kernel likewise."""
    g = torch.Generator().manual_seed(B * H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    temb = torch.randn(B, Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    rsc = torch.rand(B, generator=g) + 0.5
    assert _lib.conv2d_winograd43_ok(B, H, W, Cin, Cout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wt = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    u = _lib.winograd43_pack(wt, Cin, Cout)
    assert u.numel() == 36 * Cin * Cout
    out = torch.full((B, H, W, Cout), float("nan"), device=DEV)          # every output must be written
    _lib.conv2d_winograd43(xd, u, out, B, H, W, Cin, Cout, epilogue=_lib.make_epilogue(bias=b.to(DEV)))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref) < 3e-6
    # elementwise too: a misplaced store moves single entries by O(1) while the norm moves by 1e-2
    assert float((out.permute(0, 3, 1, 2).cpu().double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    direct = torch.empty_like(out)
    _lib.conv2d_nhwc(xd, wt, direct, B, H, W, Cin, Cout, 3, 3, 1, 1, epilogue=_lib.make_epilogue(bias=b.to(DEV)))
    assert rel_err(out.cpu(), direct.double().cpu()) < 3e-6
    resd = res.permute(0, 2, 3, 1).contiguous().to(DEV)
    _lib.conv2d_winograd43(xd, u, out, B, H, W, Cin, Cout,
                           epilogue=_lib.make_epilogue(bias=b.to(DEV), rowbias=temb.to(DEV), rows_per_group=H * W, act="silu",
                                                       residual=resd, out_scale=0.7071, rowscale=rsc.to(DEV)))
    ref2 = (F.silu(ref + temb.double()[:, :, None, None]) + res.double()) * 0.7071 * rsc.double()[:, None, None, None]
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref2) < 3e-6
    assert float((out.permute(0, 3, 1, 2).cpu().double() - ref2).abs().max()) < 1e-4 * float(ref2.abs().max())


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 32, 32, 128, 128), (3, 8, 8, 256, 64), (5, 8, 12, 80, 64), (2, 16, 16, 512, 256), (130, 4, 4, 48, 64),
                                            (1, 64, 64, 32, 64), (2, 8, 8, 32, 64), (33, 8, 8, 256, 256), (7, 16, 16, 384, 256)])
def test_conv2d_winograd43_pairs_vs_cpu(B, H, W, Cin, Cout):
    """The same convolution with its 36 contractions on the fp16 matrix cores, every fp32 operand as a pair of fp16 values
    (winograd43h_kernel): same bars as the fp32 contraction (3e-6 against the fp64 CPU convolution with every epilogue term, against
    the direct kernel likewise; measured 0.6-1.5e-6, at or below the fp32 contraction's).  Inputs span five decades with exact
    zeros among them: elements far below the typical magnitude have a SUBNORMAL low part, which the matrix core must keep."""
    g = torch.Generator().manual_seed(B * H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    x = x * (torch.rand(B, Cin, H, W, generator=g) < 0.9) * torch.exp(torch.randn(B, Cin, H, W, generator=g).clamp(-6, 2))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    temb = torch.randn(B, Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    rsc = torch.rand(B, generator=g) + 0.5
    assert _lib.conv2d_winograd43h_ok(B, H, W, Cin, Cout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wt = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    u = _lib.winograd43_pack(wt, Cin, Cout, pairs=True)
    assert u.numel() == 36 * Cin * Cout + 4
    descale = float(u[-4])
    assert descale > 0 and np.log2(descale) == round(np.log2(descale))       # a power of two: undone exactly
    out = torch.full((B, H, W, Cout), float("nan"), device=DEV)          # every output must be written
    _lib.conv2d_winograd43(xd, u, out, B, H, W, Cin, Cout, epilogue=_lib.make_epilogue(bias=b.to(DEV)), pairs=True)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref) < 3e-6
    assert float((out.permute(0, 3, 1, 2).cpu().double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    direct = torch.empty_like(out)
    _lib.conv2d_nhwc(xd, wt, direct, B, H, W, Cin, Cout, 3, 3, 1, 1, epilogue=_lib.make_epilogue(bias=b.to(DEV)))
    assert rel_err(out.cpu(), direct.double().cpu()) < 3e-6
    resd = res.permute(0, 2, 3, 1).contiguous().to(DEV)
    ns = _lib.conv2d_winograd43_colstats_split(B, H, W, Cin, Cout)
    cs = torch.full((B * ns * Cout * 2,), float("nan"), device=DEV, dtype=torch.float64) if ns > 0 else None
    _lib.conv2d_winograd43(xd, u, out, B, H, W, Cin, Cout, pairs=True,
                           epilogue=_lib.make_epilogue(bias=b.to(DEV), rowbias=temb.to(DEV), rows_per_group=H * W, act="silu",
                                                       residual=resd, out_scale=0.7071, rowscale=rsc.to(DEV), colstats=cs))
    ref2 = (F.silu(ref + temb.double()[:, :, None, None]) + res.double()) * 0.7071 * rsc.double()[:, None, None, None]
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref2) < 3e-6
    assert float((out.permute(0, 3, 1, 2).cpu().double() - ref2).abs().max()) < 1e-4 * float(ref2.abs().max())
    if ns > 0:                                                            # the column sums are those of the stored outputs
        tot = cs.view(B, ns, Cout, 2).sum(1)
        o64 = out.double().reshape(B, H * W, Cout)
        torch.testing.assert_close(tot[..., 0], o64.sum(1), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(tot[..., 1], (o64 * o64).sum(1), rtol=1e-6, atol=1e-6)


def test_conv2d_winograd43_pairs_limits():
    """What the fp16-pair form refuses (-> the fp32 contraction serves it), its switch, and its documented failure: a transformed
    input beyond the fp16 range gives NaN outputs, never finite wrong ones."""
    assert not _lib.conv2d_winograd43h_ok(2, 8, 8, 24, 64)       # Cin % 16
    assert not _lib.conv2d_winograd43h_ok(2, 8, 8, 16, 64)       # fewer than two K steps
    assert _lib.conv2d_winograd43_ok(2, 8, 8, 24, 64)            # ... which the fp32 form takes
    with _lib.thread_option("IDIFF_NO_WINO43H", 1):
        assert not _lib.conv2d_winograd43h_ok(2, 8, 8, 32, 64)
        assert _lib.conv2d_winograd43_ok(2, 8, 8, 32, 64)
    assert _lib.conv2d_winograd43h_ok(2, 8, 8, 32, 64)
    g = torch.Generator().manual_seed(0)
    wt = (torch.randn(64, 3, 3, 32, generator=g) / 17).to(DEV)
    u = _lib.winograd43_pack(wt, 32, 64, pairs=True)
    with pytest.raises(RuntimeError, match="pairs=True"):         # an fp32 bank handed to the pair kernel
        _lib.conv2d_winograd43(torch.zeros(2, 8, 8, 32, device=DEV), _lib.winograd43_pack(wt, 32, 64), torch.zeros(2, 8, 8, 64, device=DEV),
                               2, 8, 8, 32, 64, pairs=True)
    x = torch.randn(2, 8, 8, 32, generator=g).to(DEV)
    out = torch.empty(2, 8, 8, 64, device=DEV)
    _lib.conv2d_winograd43(x * 300.0, u, out, 2, 8, 8, 32, 64, pairs=True)      # |V| up to ~1.2e4: well inside
    ref = torch.empty_like(out)
    _lib.conv2d_winograd43(x * 300.0, _lib.winograd43_pack(wt, 32, 64), ref, 2, 8, 8, 32, 64)
    assert rel_err(out.cpu(), ref.double().cpu()) < 3e-6
    _lib.conv2d_winograd43(x * 1e5, u, out, 2, 8, 8, 32, 64, pairs=True)        # beyond: loud
    assert bool(torch.isnan(out).any()) and not bool(torch.isfinite(out).all())
    # an all-zero filter packs (no scale to find) and convolves to the bias
    uz = _lib.winograd43_pack(torch.zeros_like(wt), 32, 64, pairs=True)
    bias = torch.randn(64, generator=g).to(DEV)
    _lib.conv2d_winograd43(x, uz, out, 2, 8, 8, 32, 64, epilogue=_lib.make_epilogue(bias=bias), pairs=True)
    torch.testing.assert_close(out, bias.expand(2, 8, 8, 64), rtol=0, atol=0)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 32, 32, 128, 128), (3, 8, 8, 256, 64), (5, 16, 16, 80, 64), (2, 16, 16, 512, 256), (1, 64, 32, 32, 64),
                                            (2, 8, 8, 32, 64), (33, 8, 8, 256, 256), (7, 16, 16, 384, 256), (3, 32, 16, 48, 128), (9, 4, 8, 64, 64),
                                            (2, 128, 8, 32, 64), (2, 64, 64, 128, 64), (1, 16, 64, 32, 128), (3, 8, 64, 64, 64), (130, 4, 4, 48, 64), (37, 4, 4, 256, 128),
                                            (5, 8, 4, 32, 64)])
def test_conv2d_wino1d_vs_cpu(B, H, W, Cin, Cout):
    """The 3x3 convolution by F(4, 3) along the rows, three filter rows summed directly, on fp16 pairs (wino1d_kernel) against the fp64 CPU
    convolution with every epilogue term, the direct kernel and the column sums: blocks of an eighth of an image with a halo row on both sides
    (W = 64), of half an image (W = 32), of several images
    (W = 16, 8: the rows of a neighbouring image must not leak into a sample's first / last row), partial last blocks, non-square maps with
    one or two halo rows per block, 2 to 32 K steps, several output-channel tiles.  Bar 3e-6, that of the 2-D pair form (measured below it:
    one transform instead of two).  Inputs span five decades with exact zeros among them, as in the 2-D form's test."""
    g = torch.Generator().manual_seed(B * H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    x = x * (torch.rand(B, Cin, H, W, generator=g) < 0.9) * torch.exp(torch.randn(B, Cin, H, W, generator=g).clamp(-6, 2))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    temb = torch.randn(B, Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    rsc = torch.rand(B, generator=g) + 0.5
    assert _lib.conv2d_wino1d_ok(B, H, W, Cin, Cout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wt = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    u = _lib.wino1d_pack(wt, Cin, Cout)
    assert u.numel() == 18 * Cin * Cout + 4
    descale = float(u[-4])
    assert descale > 0 and np.log2(descale) == round(np.log2(descale))       # a power of two: undone exactly
    out = torch.full((B, H, W, Cout), float("nan"), device=DEV)          # every output must be written
    _lib.conv2d_wino1d(xd, u, out, B, H, W, Cin, Cout, epilogue=_lib.make_epilogue(bias=b.to(DEV)))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref) < 3e-6
    # elementwise too: a misplaced store or a leaked neighbouring row moves single entries by O(1) while the norm moves by 1e-2
    assert float((out.permute(0, 3, 1, 2).cpu().double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    direct = torch.empty_like(out)
    _lib.conv2d_nhwc(xd, wt, direct, B, H, W, Cin, Cout, 3, 3, 1, 1, epilogue=_lib.make_epilogue(bias=b.to(DEV)))
    assert rel_err(out.cpu(), direct.double().cpu()) < 3e-6
    resd = res.permute(0, 2, 3, 1).contiguous().to(DEV)
    ns = _lib.conv2d_wino1d_colstats_split(B, H, W, Cin, Cout)
    assert ns == max(1, H * W // 512)
    cs = torch.full((B * ns * Cout * 2,), float("nan"), device=DEV, dtype=torch.float64)
    _lib.conv2d_wino1d(xd, u, out, B, H, W, Cin, Cout,
                       epilogue=_lib.make_epilogue(bias=b.to(DEV), rowbias=temb.to(DEV), rows_per_group=H * W, act="silu",
                                                   residual=resd, out_scale=0.7071, rowscale=rsc.to(DEV), colstats=cs))
    ref2 = (F.silu(ref + temb.double()[:, :, None, None]) + res.double()) * 0.7071 * rsc.double()[:, None, None, None]
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref2) < 3e-6
    assert float((out.permute(0, 3, 1, 2).cpu().double() - ref2).abs().max()) < 1e-4 * float(ref2.abs().max())
    tot = cs.view(B, ns, Cout, 2).sum(1)                                  # the column sums are those of the stored outputs
    o64 = out.double().reshape(B, H * W, Cout)
    torch.testing.assert_close(tot[..., 0], o64.sum(1), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(tot[..., 1], (o64 * o64).sum(1), rtol=1e-6, atol=1e-6)


def test_conv2d_wino1d_limits():
    """What the row-wise pair form refuses, its switches, and its documented failure: an input beyond the fp16 range of the transformed
    values gives NaN outputs, never finite wrong ones."""
    assert not _lib.conv2d_wino1d_ok(2, 8, 12, 32, 64)           # W not 4 / 8 / 16 / 32 / 64
    assert not _lib.conv2d_wino1d_ok(2, 8, 8, 24, 64)            # Cin % 16
    assert not _lib.conv2d_wino1d_ok(2, 8, 8, 16, 64)            # fewer than two K steps
    assert not _lib.conv2d_wino1d_ok(2, 24, 32, 32, 64)          # a block of 16 rows would straddle two images
    assert not _lib.conv2d_wino1d_ok(2, 8, 8, 32, 48)            # Cout % 64
    for name in ("IDIFF_NO_WINO1D", "IDIFF_NO_WINO43H", "IDIFF_NO_WINOGRAD"):
        with _lib.thread_option(name, 1):
            assert not _lib.conv2d_wino1d_ok(2, 8, 8, 32, 64)
    assert _lib.conv2d_wino1d_ok(2, 8, 8, 32, 64)
    g = torch.Generator().manual_seed(0)
    wt = (torch.randn(64, 3, 3, 32, generator=g) / 17).to(DEV)
    u = _lib.wino1d_pack(wt, 32, 64)
    with pytest.raises(RuntimeError, match="wino1d_pack"):        # another kernel's bank
        _lib.conv2d_wino1d(torch.zeros(2, 8, 8, 32, device=DEV), _lib.winograd43_pack(wt, 32, 64, pairs=True), torch.zeros(2, 8, 8, 64, device=DEV),
                           2, 8, 8, 32, 64)
    with pytest.raises(RuntimeError, match="not supported"):      # a geometry the kernel does not serve is refused, not computed wrongly
        _lib.conv2d_wino1d(torch.zeros(2, 8, 12, 32, device=DEV), u, torch.zeros(2, 8, 12, 64, device=DEV), 2, 8, 12, 32, 64)
    x = torch.randn(2, 8, 8, 32, generator=g).to(DEV)
    out = torch.empty(2, 8, 8, 64, device=DEV)
    ref = torch.empty_like(out)
    _lib.conv2d_wino1d(x * 3000.0, u, out, 2, 8, 8, 32, 64)       # |V| up to ~6e4 / 5.4: inside, where the 2-D form is already beyond its range
    _lib.conv2d_nhwc(x * 3000.0, wt, ref, 2, 8, 8, 32, 64, 3, 3, 1, 1)
    assert rel_err(out.cpu(), ref.double().cpu()) < 3e-6
    _lib.conv2d_wino1d(x * 1e5, u, out, 2, 8, 8, 32, 64)          # beyond: loud
    assert bool(torch.isnan(out).any()) and not bool(torch.isfinite(out).all())
    uz = _lib.wino1d_pack(torch.zeros_like(wt), 32, 64)           # an all-zero filter packs (no scale to find) and convolves to the bias
    bias = torch.randn(64, generator=g).to(DEV)
    _lib.conv2d_wino1d(x, uz, out, 2, 8, 8, 32, 64, epilogue=_lib.make_epilogue(bias=bias))
    torch.testing.assert_close(out, bias.expand(2, 8, 8, 64), rtol=0, atol=0)


def test_conv2d_winograd43_rejects_what_it_cannot_take():
    assert not _lib.conv2d_winograd43_ok(2, 6, 8, 32, 64)       # height not a multiple of 4
    assert not _lib.conv2d_winograd43_ok(2, 8, 8, 4, 64)        # Cin % 8
    assert not _lib.conv2d_winograd43_ok(2, 8, 8, 32, 3)        # Cout % 64
    with _lib.thread_option("IDIFF_NO_WINO43", 1):
        assert not _lib.conv2d_winograd43_ok(2, 8, 8, 32, 64)
    assert _lib.conv2d_winograd43_ok(2, 8, 8, 32, 64)
    x = torch.zeros(2, 8, 8, 32, device=DEV)
    u = torch.zeros(36 * 32 * 64, device=DEV)
    with pytest.raises(RuntimeError, match="per image"):          # a row group that is not the image
        _lib.conv2d_winograd43(x, u, torch.zeros(2, 8, 8, 64, device=DEV), 2, 8, 8, 32, 64,
                               epilogue=_lib.make_epilogue(rowbias=torch.zeros(4, 64, device=DEV), rows_per_group=32))
    with pytest.raises(RuntimeError, match="filter bank"):
        _lib.conv2d_winograd43(x, torch.zeros(16 * 32 * 64, device=DEV), torch.zeros(2, 8, 8, 64, device=DEV), 2, 8, 8, 32, 64)


@pytest.mark.parametrize("B,H,Cin,Cout", [(3, 32, 64, 128), (6, 16, 128, 64), (5, 8, 64, 128), (11, 4, 32, 64), (64, 4, 8, 64), (70, 8, 16, 64)])
def test_winograd43_colstats_feed_groupnorm(B, H, Cin, Cout):
    """Column sums from the F(4x4,3x3) epilogue = a statistics pass over its output: two workgroups per sample (32x32), two, eight
    and thirty-two whole samples per workgroup (16x16, 8x8, 4x4), sample counts that leave the last workgroup partly empty."""
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, H * H, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    ns = _lib.conv2d_winograd43_colstats_split(B, H, H, Cin, Cout)
    assert ns == max(1, H * H // 512)
    cs = torch.full((B * ns * Cout * 2,), float("nan"), device=DEV, dtype=torch.float64)
    out = torch.empty(B, H * H, Cout, device=DEV)
    u = _lib.winograd43_pack(w, Cin, Cout)
    _lib.conv2d_winograd43(x, u, out, B, H, H, Cin, Cout, epilogue=_lib.make_epilogue(bias=bias, act="silu", rows_per_group=H * H, colstats=cs))
    G = 32
    st_a, st_b = torch.empty(B * G * 2, device=DEV), torch.empty(B * G * 2, device=DEV)
    _lib.groupnorm_finalize(cs, ns, Cout, None, 0, 0, B, H * H, G, 1e-6, st_a)
    nsp = _lib.groupnorm_nsplit(B, H * H, Cout)
    ws = torch.empty(B * nsp * Cout * 2, device=DEV, dtype=torch.float64)
    _lib.groupnorm_stats(out, Cout, None, 0, B, H * H, G, 1e-6, ws, st_b)
    torch.testing.assert_close(st_a, st_b, rtol=1e-6, atol=1e-7)
    tot = cs.view(B, ns, Cout, 2).sum(1)
    torch.testing.assert_close(tot[..., 0], out.double().sum(1), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(tot[..., 1], (out.double() ** 2).sum(1), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,H,Cin,Cout", [(3, 32, 64, 128), (6, 16, 128, 64), (5, 8, 64, 128), (7, 8, 32, 64), (70, 8, 32, 64), (2, 64, 32, 64), (11, 4, 32, 64), (70, 4, 32, 64)])
def test_wino1d_colstats_feed_groupnorm(B, H, Cin, Cout):
    """Column sums from the row-wise pair kernel's epilogue = a statistics pass over its output: two workgroups per sample (32x32), two and
    eight whole samples per workgroup (16x16, 8x8), sample counts that leave the last workgroup partly empty."""
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, H * H, Cin, generator=g).to(DEV)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    ns = _lib.conv2d_wino1d_colstats_split(B, H, H, Cin, Cout)
    assert ns == max(1, H * H // 512)
    cs = torch.full((B * ns * Cout * 2,), float("nan"), device=DEV, dtype=torch.float64)
    out = torch.empty(B, H * H, Cout, device=DEV)
    u = _lib.wino1d_pack(w, Cin, Cout)
    _lib.conv2d_wino1d(x, u, out, B, H, H, Cin, Cout, epilogue=_lib.make_epilogue(bias=bias, act="silu", rows_per_group=H * H, colstats=cs))
    G = 32
    st_a, st_b = torch.empty(B * G * 2, device=DEV), torch.empty(B * G * 2, device=DEV)
    _lib.groupnorm_finalize(cs, ns, Cout, None, 0, 0, B, H * H, G, 1e-6, st_a)
    nsp = _lib.groupnorm_nsplit(B, H * H, Cout)
    ws = torch.empty(B * nsp * Cout * 2, device=DEV, dtype=torch.float64)
    _lib.groupnorm_stats(out, Cout, None, 0, B, H * H, G, 1e-6, ws, st_b)
    torch.testing.assert_close(st_a, st_b, rtol=1e-6, atol=1e-7)
    tot = cs.view(B, ns, Cout, 2).sum(1)
    torch.testing.assert_close(tot[..., 0], out.double().sum(1), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(tot[..., 1], (out.double() ** 2).sum(1), rtol=1e-6, atol=1e-6)
