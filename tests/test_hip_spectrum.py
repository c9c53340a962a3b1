"""GPU parity of the spectrum kernels (centring + fp64 Gram + tridiagonalisation + bisection) and the integer ID.

Tolerance from BASELINE.json north_star: singular values within 1e-4 relative of the reference CPU path, ID exact.
"""
import numpy as np
import pytest
import torch

import id_diff_amd
from id_diff_amd import _lib, plot_utils
from oracle import dim as odim

pytestmark = pytest.mark.gpu
DEV = "cuda"
SV_RTOL = 1e-4


def check(S, ids_exact=True):
    sv = _lib.spectrum(S.to(DEV)).cpu()
    ref32 = odim.spectrum(S)                       # the reference path: fp32 centring + fp32 gesdd
    Sd = S.double()
    exact = torch.linalg.svdvals(Sd - Sd.mean(0, keepdim=True))   # fp64 centring + fp64 SVD of the same input
    n = min(S.shape)
    # 1e-4 relative against the reference path wherever fp32 gesdd itself is meaningful
    # (its own absolute noise floor is ~1e-6 * sigma_max)
    floor = 2e-5 * float(exact[0])
    keep = exact > floor
    np.testing.assert_allclose(sv.numpy()[keep], ref32.numpy()[keep], rtol=SV_RTOL)
    # against the exact answer the HIP path is accurate to its final fp32 rounding
    np.testing.assert_allclose(sv.double().numpy()[keep], exact.numpy()[keep], rtol=5e-7)
    assert sv.shape == (n,) and bool((sv[:-1] >= sv[1:]).all())
    if ids_exact:
        assert plot_utils.estimate_dim(sv.tolist()) == odim.estimate_dim(ref32.tolist())
    return sv


def test_golden_matrices(golden):
    z = golden("svd_rule.npz")
    for i in range(int(z["n_mats"])):
        S = torch.from_numpy(z[f"m{i}::S"])
        sv = check(S)
        np.testing.assert_allclose(sv.numpy(), z[f"m{i}::sv_ref_f32"], rtol=SV_RTOL)
        assert plot_utils.estimate_dim(sv.tolist()) == int(z[f"m{i}::dim"]) == int(z[f"m{i}::k_true"])


@pytest.mark.parametrize("M,D", [(1501, 100), (300, 128), (400, 129), (700, 257), (1100, 1024), (64, 64), (5, 3)])
def test_gaussian_matrices(M, D):
    g = torch.Generator().manual_seed(M * 7 + D)
    S = torch.randn(M, D, generator=g) * 3 + torch.randn(D, generator=g) * 10   # big column means: centring matters
    check(S, ids_exact=False)


@pytest.mark.parametrize("cond", [1e2, 1e3, 1e4])
def test_geometric_spectra(cond):
    g = torch.Generator().manual_seed(int(cond))
    M, D = 900, 200
    u, _ = torch.linalg.qr(torch.randn(M, D, generator=g, dtype=torch.float64))
    v, _ = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))
    s = torch.logspace(0, -np.log10(cond), D, dtype=torch.float64) * 500
    S = ((u * s) @ v.T).float()
    check(S, ids_exact=False)


def test_batched_ksphere_shape():
    """BASELINE config 2: P matrices of 1501 x 100 in one call."""
    g = torch.Generator().manual_seed(2)
    P = 16
    S = torch.randn(P, 1501, 100, generator=g)
    S[:, :, 40:] *= 0.02   # a cliff at index 40
    sv = _lib.spectrum(S.to(DEV)).cpu()
    for p in range(P):
        ref = odim.spectrum(S[p])
        np.testing.assert_allclose(sv[p].numpy(), ref.numpy(), rtol=SV_RTOL)
        assert plot_utils.estimate_dim(sv[p].tolist()) == odim.estimate_dim(ref.tolist()) == 60


@pytest.mark.parametrize("P", [64, 512, 4096])
def test_batched_ksphere_at_config2_sizes(P):
    """SURVEY 8(d) cfg 2: P in {64, 512, 4096} matrices of 1501 x 100 in ONE call (2.4 GB of fp32 at P = 4096, generated on
    the device).  Every matrix: descending, sum sv^2 = ||S - mean||_F^2, ID = planted 60; 32 of them against the
    reference path (fp32 gesdd on the host) at 1e-4."""
    M, D, k = 1501, 100, 60
    g = torch.Generator(device=DEV).manual_seed(P)
    S = torch.randn(P, M, D, device=DEV, generator=g)
    S[:, :, 40:] *= 0.02
    S += 0.5
    sv = _lib.spectrum(S)
    assert sv.shape == (P, D) and bool((sv[:, :-1] >= sv[:, 1:]).all())
    Sd = S[: min(P, 512)].double()                                           # fp64 Frobenius check on up to 512 of them
    fro2 = ((Sd - Sd.mean(1, keepdim=True)) ** 2).sum((1, 2))
    torch.testing.assert_close((sv[: Sd.shape[0]].double() ** 2).sum(1), fro2, rtol=1e-5, atol=0)
    svc = sv.cpu()
    ids = {plot_utils.estimate_dim(row.tolist()) for row in svc}
    assert ids == {k}
    for p in torch.linspace(0, P - 1, 32).long().tolist():
        np.testing.assert_allclose(svc[p].numpy(), odim.spectrum(S[p].cpu()).numpy(), rtol=SV_RTOL)


@pytest.mark.parametrize("cond", [1e5, 1e6])
def test_geometric_spectra_stress(cond):
    """SURVEY 8(d) stress set up to cond 1e6, with the floor of the Gram route written down: the Gram eigenvalues carry
    an absolute error of a few u * lambda_max (u = 2^-53), i.e. a singular value sigma is good to
    |d sigma| <= c u sigma_max^2 / (2 sigma).  At the 1e-4 relative bar that floor sits at sigma ~ 2e-6 sigma_max; below
    it the test asks for the absolute bound instead.  (The reference's own fp32 gesdd is only good to ~3e-7 sigma_max
    absolute, so its values below ~3e-3 sigma_max are no yardstick: the comparison is against an fp64 SVD.)"""
    g = torch.Generator().manual_seed(int(cond))
    M, D = 900, 200
    u, _ = torch.linalg.qr(torch.randn(M, D, generator=g, dtype=torch.float64))
    v, _ = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))
    s = torch.logspace(0, -np.log10(cond), D, dtype=torch.float64) * 500
    S = ((u * s) @ v.T).float()
    sv = _lib.spectrum(S.to(DEV)).cpu().double()
    Sd = S.double()
    exact = torch.linalg.svdvals(Sd - Sd.mean(0, keepdim=True))
    smax = float(exact[0])
    bound = 1e-4 * exact + 40 * 2.0 ** -53 * smax * smax / (2 * exact.clamp_min(1e-300)) + 1e-7 * exact   # + fp32 rounding of sv
    assert bool(((sv - exact).abs() <= bound).all()), float(((sv - exact).abs() / bound).max())
    above = exact > 2e-6 * smax
    np.testing.assert_allclose(sv.numpy()[above], exact.numpy()[above], rtol=SV_RTOL)


def test_image_sized_matrix():
    """BASELINE config 3 size (4480 x 3072): large-D path; properties + parity with the CPU SVD."""
    g = torch.Generator().manual_seed(3)
    M, D, k = 4480, 3072, 96
    S = torch.randn(M, D, generator=g)
    S[:, D - k:] *= 1.0 / 70.0
    S = S @ torch.linalg.qr(torch.randn(D, D, generator=g))[0]       # hide the cliff in a rotation
    S = S + 5.0
    sv, eig = _lib.spectrum(S.to(DEV), return_eig=True)
    sv, eig = sv.cpu(), eig.cpu()
    ref = odim.spectrum(S)
    np.testing.assert_allclose(sv.numpy(), ref.numpy(), rtol=SV_RTOL)
    assert plot_utils.estimate_dim(sv.tolist()) == odim.estimate_dim(ref.tolist()) == k
    # trace identity: sum of Gram eigenvalues == squared Frobenius norm of the centred matrix
    c = S.double() - S.double().mean(0, keepdim=True)
    assert abs(float(eig.sum()) / float((c * c).sum()) - 1.0) < 1e-10


def test_config5_sized_matrix(golden):
    """BASELINE config 5 size (S 16768 x 12288: 64x64x3 images, B = 128): the D > 8192 path (1.2 GB fp64 Gram) against
    an fp64 ``torch.linalg.svdvals`` of the same matrix computed once in the build container
    (tests/golden/make_spectrum_cfg5.py; the matrix is rebuilt from its seed).  Planted cliff: 64 columns at 1/70."""
    from golden.make_spectrum_cfg5 import cfg5_matrix, D, K_PLANTED, M
    z = golden("spectrum_cfg5.npz")
    ref = z["sv_f64"]
    S = cfg5_matrix()
    assert S.shape == (M, D)
    c = S - S.mean(dim=0, keepdim=True)
    fro2 = float((c.double() ** 2).sum())
    assert abs(fro2 / float(z["fro2"]) - 1.0) < 1e-12            # the same matrix as the one the fixture was made from
    del c
    sv, eig = _lib.spectrum(S.to(DEV), return_eig=True)
    sv, eig = sv.cpu().double().numpy(), eig.cpu()
    assert sv.shape == (D,) and bool((sv[:-1] >= sv[1:]).all())
    # every singular value, top to bottom, within the 1e-4 bar of the north star (measured ~1e-7: fp32 rounding of sv)
    np.testing.assert_allclose(sv, ref, rtol=1e-4)
    np.testing.assert_allclose(sv[[0, 1, D - K_PLANTED - 1, D - K_PLANTED, D - 1]],
                               ref[[0, 1, D - K_PLANTED - 1, D - K_PLANTED, D - 1]], rtol=2e-6)
    assert plot_utils.estimate_dim(sv.tolist()) == odim.estimate_dim(ref.tolist()) == K_PLANTED
    # trace identity: the Gram eigenvalues sum to ||S - fp64 mean||_F^2 (the kernel centres in fp64)
    Sd = S.double()
    c64 = Sd - Sd.mean(0, keepdim=True)
    assert abs(float(eig.sum()) / float((c64 * c64).sum()) - 1.0) < 1e-10


def _gram(kind, D, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "gauss":
        S = torch.randn(D + 40, D, generator=g, dtype=torch.float64)
    elif kind == "cliff":
        S = torch.randn(D + 90, D, generator=g, dtype=torch.float64)
        S[:, D - 37:] /= 70
        S = S @ torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
    elif kind == "lowrank":                      # rank 20 < panel width 32: CholeskyQR meets exactly dependent columns
        S = torch.randn(D + 10, 20, generator=g, dtype=torch.float64) @ torch.randn(20, D, generator=g, dtype=torch.float64)
    elif kind == "zero":                         # e.g. the zero-initialised output conv of a fresh BeatGANs U-Net
        S = torch.zeros(D + 10, D, dtype=torch.float64)
    elif kind == "diagonal":                     # every panel below the band is exactly zero
        return torch.diag(torch.linspace(1.0, 50.0, D, dtype=torch.float64))
    return S.T @ S


@pytest.mark.parametrize("kind,D", [("gauss", 129), ("gauss", 257), ("cliff", 517), ("lowrank", 300), ("zero", 260), ("diagonal", 333),
                                    ("gauss", 1024), ("cliff", 1501)])
def test_two_stage_eigensolver(kind, D):
    """idiff_symtridiag_f64 for D > 128 = blocked band reduction (CholeskyQR2 + Householder-reconstruction panels, rank-64
    trailing updates on the fp64 matrix cores) + systolic bulge chasing: stage 1 alone preserves the spectrum to rounding,
    and the three routes to the tridiagonal matrix (two-stage with the systolic chase, with one launch per wavefront, the
    unblocked one-stage sweep) give the same eigenvalues as LAPACK."""
    G = _gram(kind, D, seed=D)
    ref = torch.linalg.eigvalsh(G)
    scale = max(float(ref.abs().max()), 1e-300)
    Gd = G.to(DEV)
    B = _lib.sym_band(Gd.clone()).cpu()
    i, j = torch.meshgrid(torch.arange(D), torch.arange(D), indexing="ij")
    assert float(B[(i - j).abs() > 32].abs().max()) == 0.0 and torch.equal(B, B.T)
    assert float((torch.linalg.eigvalsh(B) - ref).abs().max()) <= 2e-14 * scale
    tol = 5e-14 * scale                                        # the bisection stops at 1e-13 relative
    ev = _lib.sym_eigvals(Gd.clone()).cpu()
    assert float((ev - ref).abs().max()) <= tol
    for opt in ("IDIFF_CHASE_WAVEFRONT", "IDIFF_TRIDIAG_ONESTAGE"):
        if opt == "IDIFF_TRIDIAG_ONESTAGE" and D % 2:
            continue                                           # its streaming form needs an even D
        _lib.set_option(opt, True)
        try:
            other = _lib.sym_eigvals(Gd.clone()).cpu()
        finally:
            _lib.set_option(opt, False)
        assert float((other - ref).abs().max()) <= tol, opt


@pytest.mark.parametrize("D", [4352, 4330])
def test_band_reduction_large_block_forms(D):
    """Blocks of >= 2048 tiles take the pipelined strip kernel for the trailing update (sbr.hip, k11p; D = 4330 does not end
    on a tile boundary: the last tile row, the diagonal and the half first column go through the one-tile kernel), and
    IDIFF_SBR_LOOKAHEAD (opt-in) runs the bulk of the update on a helper stream: the same eigenvalues as LAPACK in the
    default form, and bit for bit the same result with the look-ahead."""
    g = torch.Generator(device="cpu").manual_seed(D)
    S = torch.randn(D + 40, D, generator=g, dtype=torch.float64)
    G = (S.T @ S).to(DEV)
    ref = torch.linalg.eigvalsh(G).cpu()                       # rocSOLVER on the device: seconds at this size on the host
    scale = float(ref.abs().max())
    ev = _lib.sym_eigvals(G.clone())
    assert float((ev.cpu() - ref).abs().max()) <= 5e-14 * scale
    prev = _lib.set_option("IDIFF_SBR_LOOKAHEAD", True)
    try:
        ev2 = _lib.sym_eigvals(G.clone())
        ev3 = _lib.sym_eigvals(G.clone())                     # the helper stream and its events are reused
    finally:
        _lib.set_option("IDIFF_SBR_LOOKAHEAD", int(prev))
    assert torch.equal(ev, ev2) and torch.equal(ev, ev3)


def test_stage_exports():
    g = torch.Generator().manual_seed(4)
    M, D = 257, 70
    S = torch.randn(1, M, D, generator=g)
    d = S.to(DEV)
    mean = torch.empty(1, D, device=DEV, dtype=torch.float64)
    scratch = torch.empty(32 * D, device=DEV, dtype=torch.float64)
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.idiff_colmean_f64(d.data_ptr(), 1, M, D, mean.data_ptr(), scratch.data_ptr(), st) == 0
    np.testing.assert_allclose(mean.cpu().numpy()[0], S[0].double().mean(0).numpy(), rtol=1e-12, atol=1e-14)
    G = torch.empty(1, D, D, device=DEV, dtype=torch.float64)
    assert lib.idiff_centered_gram_f64(d.data_ptr(), mean.data_ptr(), 1, M, D, G.data_ptr(), st) == 0
    c = S[0].double() - S[0].double().mean(0, keepdim=True)
    np.testing.assert_allclose(G.cpu().numpy()[0], (c.T @ c).numpy(), rtol=1e-11, atol=1e-11)
    assert torch.equal(G[0], G[0].T)
    diag, off = torch.empty(1, D, device=DEV, dtype=torch.float64), torch.zeros(1, D, device=DEV, dtype=torch.float64)
    scr = torch.empty(max(1, lib.idiff_symtridiag_scratch_doubles(D)), device=DEV, dtype=torch.float64)
    Gc = G.clone()
    assert lib.idiff_symtridiag_f64(Gc.data_ptr(), 1, D, diag.data_ptr(), off.data_ptr(), scr.data_ptr(), st) == 0
    eig = torch.empty(1, D, device=DEV, dtype=torch.float64)
    assert lib.idiff_tridiag_eigvals_f64(diag.data_ptr(), off.data_ptr(), 1, D, eig.data_ptr(), st) == 0
    ref = torch.linalg.eigvalsh(c.T @ c)
    np.testing.assert_allclose(eig.cpu().numpy()[0], ref.numpy(), rtol=1e-9, atol=1e-9 * float(ref[-1]))


def test_fewer_rows_than_columns():
    """torch.linalg.svd returns min(M, D) values (dim_reduction.py:197 never checks the shape): a short last loader batch
    must not fail mid-run.  Both solver paths (D <= 128 in LDS, two-stage)."""
    for M, D in ((3, 5), (40, 100), (150, 260)):
        S = torch.randn(M, D, generator=torch.Generator().manual_seed(M)) + 2.0
        sv = _lib.spectrum(S.to(DEV)).cpu()
        Sd = S.double()
        exact = torch.linalg.svdvals(Sd - Sd.mean(0, keepdim=True))
        assert sv.shape == (M,)
        # the centred matrix has rank M - 1: its last singular value is rounding noise in every implementation
        np.testing.assert_allclose(sv.numpy()[:-1], exact.numpy()[:-1], rtol=2e-6)
        assert float(sv[-1]) <= 2e-6 * float(exact[0])


def test_errors():
    with pytest.raises(RuntimeError, match="no CPU path"):
        _lib.spectrum(torch.zeros(5, 3))


# ---- row-sharded single point
def test_row_sharded_spectrum_single_rank_and_emulated_halves():
    from id_diff_amd import dim_reduction
    g = torch.Generator().manual_seed(11)
    M, D = 900, 384
    S = (torch.randn(M, D, generator=g) * torch.linspace(0.1, 3.0, D) + 0.7).to(DEV)
    ref = _lib.spectrum(S)
    sv = dim_reduction.row_sharded_spectrum(S, M)
    torch.testing.assert_close(sv, ref, rtol=2e-6, atol=1e-6)
    # what two ranks would each contribute, summed by hand: same spectrum
    a, b = S[:500].contiguous(), S[500:].contiguous()
    mean = (_lib.column_sums(a) + _lib.column_sums(b)) / M
    G = _lib.centered_gram(a, mean) + _lib.centered_gram(b, mean)
    sv2 = _lib.sym_eigvals(G.clone()).clamp_min(0).sqrt().flip(0).float()
    torch.testing.assert_close(sv2, ref, rtol=2e-6, atol=1e-6)
    # the blocked form the pipeline uses: upper-triangle row blocks + one mirror = the full Gram, bit for bit
    for blocks in ((0, 384), (0, 128, 384), (0, 64, 192, 256, 384)):
        Gb = torch.zeros(D, D, device=DEV, dtype=torch.float64)
        for r0, r1 in zip(blocks[:-1], blocks[1:]):
            _lib.centered_gram_rows(a, mean, Gb, r0, r1)
        assert float(torch.tril(Gb, -64).abs().max()) == 0.0           # nothing written below the diagonal tiles
        _lib.symmetrize_upper(Gb)
        assert torch.equal(Gb, _lib.centered_gram(a, mean))
    with pytest.raises(RuntimeError, match="tile-aligned"):
        _lib.centered_gram_rows(a, mean, torch.zeros(D, D, device=DEV, dtype=torch.float64), 10, 100)
    assert float(_lib.column_sums(S[:0]).abs().sum()) == 0.0 and float(_lib.centered_gram(S[:0], mean).abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match="total_rows >= D"):
        dim_reduction.row_sharded_spectrum(S[:100].contiguous(), 100)


@pytest.mark.parametrize("M,D", [(4480, 3072), (1501, 1028), (37, 1024), (2050, 1156)])
def test_large_tile_gram_equals_the_small_tile_kernel(M, D):
    """D >= 1024 takes the 128 x 128-tile Gram kernel (operands converted and centred once at staging, prefetched):
    same reduction order as the 64 x 64 kernel, so the two must agree bit for bit; both against an fp64 matmul; D and M
    that are not multiples of the tile / stage; the row-block entry point on 128-row blocks."""
    g = torch.Generator().manual_seed(M + D)
    S = (torch.randn(M, D, generator=g) * torch.linspace(0.05, 2.0, D) + 0.3).to(DEV)
    mean = _lib.column_sums(S) / M
    G = _lib.centered_gram(S, mean)
    prev = _lib.set_option("IDIFF_GRAM_SMALL_TILES", 1)
    try:
        G_small = _lib.centered_gram(S, mean)
    finally:
        _lib.set_option("IDIFF_GRAM_SMALL_TILES", int(prev))
    assert torch.equal(G, G_small)
    assert torch.equal(G, G.T)
    Sc = S.double() - mean
    ref = Sc.T @ Sc
    assert float((G - ref).abs().max() / ref.abs().max()) < 1e-13
    nb = (D + 127) // 128
    cuts = sorted({0, 128 * (nb // 3), 128 * (2 * nb // 3), D})
    Gb = torch.zeros(D, D, device=DEV, dtype=torch.float64)
    for r0, r1 in zip(cuts[:-1], cuts[1:]):
        _lib.centered_gram_rows(S, mean, Gb, r0, r1)
    assert float(torch.tril(Gb, -128).abs().max()) == 0.0
    _lib.symmetrize_upper(Gb)
    assert torch.equal(Gb, G)


def _rows_gpu_worker(rank, world, port, q):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    import id_diff_amd  # noqa: F401
    from id_diff_amd import _lib as lib, dim_reduction, parallel, sde_lib
    from id_diff_amd.models import utils as mutils
    from id_diff_amd.plot_utils import estimate_dim as plot_utils_dim
    from helpers import fcn_config
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    torch.set_num_threads(8)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)   # two processes on the one card: gloo moves
    dev = torch.device("cuda:0")                                           # the device tensors, RCCL needs a GPU per rank
    cfg = fcn_config()
    torch.manual_seed(0)
    model = mutils.create_model(cfg).to(dev).eval()
    sde, eps = sde_lib.configure_sde(cfg)
    score_fn = mutils.get_score_fn(sde, model)
    builder = dim_reduction.ScoreMatrixBuilder(score_fn, sde, eps, dev)
    x = torch.randn(cfg.data.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
    bs = 100
    rows = dim_reduction.batching(tuple(x.shape), bs)[2]
    with torch.no_grad():
        lo, hi = parallel.my_rows(rows, rank, world)
        S_local = builder.build(x, bs, seed=77, row_range=(lo, hi))
        sv = dim_reduction.row_sharded_spectrum(S_local, rows)
        S_full = builder.build(x, bs, seed=77)
        ok_rows = bool(torch.equal(S_full[lo:hi], S_local))
        ref = lib.spectrum(S_full)
        # ... and against the ORACLE (fp32 CPU SVD as the reference path, and its fp64 yardstick) of the full matrix
        from oracle import dim as od
        ref32, ref64 = od.spectrum(S_full.cpu()), od.spectrum_f64(S_full.cpu())
        keep = ref64 > 2e-5 * ref64[0]
        err32 = float(((sv.cpu() - ref32).abs() / ref32)[keep].max())
        err64 = float(((sv.cpu().double() - ref64).abs() / ref64)[keep].max())
        same_id = plot_utils_dim(sv.tolist()) == od.estimate_dim(ref32.tolist())
        # a wider matrix (D = 384: six row blocks, packed trapezoids of the upper triangle in flight), same seed on both ranks
        g = torch.Generator().manual_seed(19)
        T_cpu = torch.randn(900, 384, generator=g) * torch.linspace(0.05, 2.0, 384) + 0.4
        a, b = parallel.my_rows(900, rank, world)
        sv_t = dim_reduction.row_sharded_spectrum(T_cpu[a:b].contiguous().to(dev), 900, block_rows=64)
        t64 = od.spectrum_f64(T_cpu)
        err_t = float(((sv_t.cpu().double() - t64).abs() / t64).max())
    q.put((rank, ok_rows, float((sv - ref).abs().max() / ref.abs().max()), err32, err64, same_id, err_t))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_pipeline_two_processes_one_gpu():
    """Two ranks (two processes on this card, gloo moving the device tensors) each evaluate half of the rows of one
    point's score matrix; the all-reduced spectrum equals the single-rank one, the rows do not depend on the split, and the
    spectrum holds the north star's 1e-4 against the oracle's CPU SVD of the full matrix (same integer ID)."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rows_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok_rows, err, err32, err64, same_id, err_t in got:
        assert ok_rows, rank
        assert err < 5e-6, (rank, err)
        assert err32 < 1e-4 and err64 < 1e-4 and same_id, (rank, err32, err64, same_id)
        assert err_t < 1e-4, (rank, err_t)


def _batched_gram_and_tridiag(S):
    """(mean, G, diag, offd) of a batch through the raw C entry points (the order idiff_spectrum_f32 runs them in)."""
    P, M, D = S.shape
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    mean = torch.empty(P, D, dtype=torch.float64, device=DEV)
    scratch = torch.empty(P * 32 * D, dtype=torch.float64, device=DEV)
    assert lib.idiff_colmean_f64(S.data_ptr(), P, M, D, mean.data_ptr(), scratch.data_ptr(), st) == 0
    G = torch.empty(P, D, D, dtype=torch.float64, device=DEV)
    assert lib.idiff_centered_gram_f64(S.data_ptr(), mean.data_ptr(), P, M, D, G.data_ptr(), st) == 0
    diag, offd = torch.empty(P, D, dtype=torch.float64, device=DEV), torch.empty(P, D, dtype=torch.float64, device=DEV)
    Gw = G.clone()                                                          # the tridiagonalisation consumes its input
    assert lib.idiff_symtridiag_f64(Gw.data_ptr(), P, D, diag.data_ptr(), offd.data_ptr(), None, st) == 0
    return mean, G, diag, offd


@pytest.mark.parametrize("P,M,D", [(5, 1501, 100), (3, 333, 64), (4, 700, 112), (3, 90, 52), (2, 257, 128), (3, 50, 20), (2, 33, 96),
                                   (2, 40, 31), (3, 9, 3), (2, 6, 2), (2, 5, 1)])
def test_batched_small_kernels_vs_the_forms_they_replace(P, M, D):
    """Config-2 kernels (round 4): gram_small_batched_kernel (one workgroup per matrix, operands centred once) must equal the
    64 x 64-tile kernel BIT FOR BIT (same k-steps in the same order) and an fp64 matmul to rounding; tridiag_reg_kernel (rows in
    registers, a lane pair per row) must give a tridiagonal with the eigenvalues of G (fp64 eigvalsh, 1e-13 of the largest) and
    agree with the LDS-resident form it replaces to the same bar.  Shapes: the config-2 size, widths on both sides of every
    block boundary of the two kernels (64, 112 | 52 -> the old Gram kernel, 128, 96 = six full blocks, 31 / 3 / 2 / 1: one
    partly filled block and the degenerate ends), M not a multiple of the 32-row stage."""
    g = torch.Generator().manual_seed(P * 1000 + M + D)
    S = (torch.randn(P, M, D, generator=g) * torch.linspace(0.2, 3.0, D) + 0.7).to(DEV)
    mean, G, diag, offd = _batched_gram_and_tridiag(S)
    with _lib.thread_option("IDIFF_GRAM_SMALL_TILES", 1), _lib.thread_option("IDIFF_TRIDIAG_ONESTAGE", 1):
        mean_o, G_o, diag_o, offd_o = _batched_gram_and_tridiag(S)
    assert torch.equal(G, G_o) and torch.equal(G, G.transpose(1, 2))
    c = S.double() - mean[:, None, :]
    ref = c.transpose(1, 2) @ c
    assert float((G - ref).abs().max() / ref.abs().max()) < 1e-13
    assert bool(torch.isfinite(diag).all()) and bool(torch.isfinite(offd).all())
    for p in range(P):
        lam = np.linalg.eigvalsh(G[p].cpu().numpy())
        scale = max(abs(lam).max(), 1e-300)
        for d, o in ((diag[p], offd[p]), (diag_o[p], offd_o[p])):
            d, o = d.cpu().numpy(), o.cpu().numpy()
            T = np.diag(d) + np.diag(o[:D - 1], 1) + np.diag(o[:D - 1], -1)
            assert np.abs(np.linalg.eigvalsh(T) - lam).max() < 1e-13 * scale


def test_register_tridiag_on_matrices_that_need_no_reflection():
    """tau = 0 steps (a column that is already tridiagonal) are no-ops with the same barrier count: a diagonal matrix, a
    tridiagonal one and the zero matrix come back unchanged."""
    D = 100
    lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    dvals = torch.linspace(1.0, 7.0, D, dtype=torch.float64)
    evals = torch.linspace(-0.5, 0.5, D - 1, dtype=torch.float64)
    mats = torch.stack([torch.diag(dvals), torch.diag(dvals) + torch.diag(evals, 1) + torch.diag(evals, -1),
                        torch.zeros(D, D, dtype=torch.float64)]).to(DEV)
    diag, offd = torch.empty(3, D, dtype=torch.float64, device=DEV), torch.empty(3, D, dtype=torch.float64, device=DEV)
    assert lib.idiff_symtridiag_f64(mats.clone().data_ptr(), 3, D, diag.data_ptr(), offd.data_ptr(), None, st) == 0
    assert torch.equal(diag[0].cpu(), dvals) and float(offd[0].abs().max()) == 0.0
    assert torch.equal(diag[1].cpu(), dvals) and torch.equal(offd[1, :D - 1].cpu().abs(), evals.abs())
    assert float(diag[2].abs().max()) == 0.0 and float(offd[2].abs().max()) == 0.0
