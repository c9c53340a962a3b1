"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz)."""
import numpy as np
import pytest
import torch

from helpers import (beatgans_config, ddpm_config, fcn_config, fill_from_seed, ncsnpp_config, overrides_from_golden,
                     rel_err, state_dict_from_golden, weight_abs_sums)
from oracle import ops as oops, sde as osde, models as omodels, ksphere as oks, dim as odim


def test_upfirdn2d_matches_reference(golden):
    z = golden("upfirdn2d.npz")
    for i in range(int(z["n_cases"])):
        up, down, p0, p1 = (int(v) for v in z[f"c{i}::params"])
        y = oops.upfirdn2d(torch.from_numpy(z[f"c{i}::x"]), torch.from_numpy(z[f"c{i}::k"]), up=up, down=down,
                           pad=(p0, p1))
        ref = torch.from_numpy(z[f"c{i}::y"])
        assert y.shape == ref.shape, i
        torch.testing.assert_close(y, ref, rtol=1e-6, atol=1e-6)
    ux, uy, dx, dy, px0, px1, py0, py1 = (int(v) for v in z["xy::params"])
    y = oops.upfirdn2d_ref(torch.from_numpy(z["xy::x"]), torch.from_numpy(z["xy::k"]), ux, uy, dx, dy, px0, px1,
                           py0, py1)
    torch.testing.assert_close(y, torch.from_numpy(z["xy::y"]), rtol=1e-6, atol=1e-6)


def test_fused_leaky_relu_matches_reference_cpu_branch(golden):
    z = golden("fused_act.npz")
    for i in range(int(z["n_cases"])):
        x, b = torch.from_numpy(z[f"c{i}::x"]), torch.from_numpy(z[f"c{i}::b"])
        torch.testing.assert_close(oops.fused_leaky_relu(x, b), torch.from_numpy(z[f"c{i}::y_default"]),
                                   rtol=1e-6, atol=1e-7)
        # slope argument is ignored on the reference's CPU branch, scale is honoured
        torch.testing.assert_close(oops.fused_leaky_relu(x, b, negative_slope=0.05, scale=1.25),
                                   torch.from_numpy(z[f"c{i}::y_slope0.05_scale1.25"]), rtol=1e-6, atol=1e-7)


def test_c_restatement_of_native_ops(golden):
    """oracle/native_ops.c (plain C, the CUDA kernels' index algebra) against the reference's outputs."""
    from oracle import c_ops
    z = golden("upfirdn2d.npz")
    for i in range(int(z["n_cases"])):
        up, down, p0, p1 = (int(v) for v in z[f"c{i}::params"])
        x = z[f"c{i}::x"]
        n, c, h, w = x.shape
        y = c_ops.upfirdn2d(x.reshape(n * c, h, w, 1), z[f"c{i}::k"], up, up, down, down, p0, p1, p0, p1)
        np.testing.assert_allclose(y.reshape(z[f"c{i}::y"].shape), z[f"c{i}::y"], rtol=1e-5, atol=1e-6)
    ux, uy, dx, dy, px0, px1, py0, py1 = (int(v) for v in z["xy::params"])
    x = z["xy::x"]
    y = c_ops.upfirdn2d(x.reshape(-1, x.shape[2], x.shape[3], 1), z["xy::k"], ux, uy, dx, dy, px0, px1, py0, py1)
    np.testing.assert_allclose(y.reshape(z["xy::y"].shape), z["xy::y"], rtol=1e-5, atol=1e-6)
    z = golden("fused_act.npz")
    for i in range(int(z["n_cases"])):
        y = c_ops.fused_bias_act(z[f"c{i}::x"], z[f"c{i}::b"], None, 3, 0, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(y, z[f"c{i}::y_default"], rtol=1e-6, atol=1e-7)
    # NHWC (minor = C) agrees with the NCHW view
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 6, 9, 7, generator=g)
    k = torch.randn(4, 4, generator=g)
    ref = oops.upfirdn2d(x, k, up=2, down=1, pad=(2, 1)).permute(0, 2, 3, 1).numpy()
    y = c_ops.upfirdn2d(x.permute(0, 2, 3, 1).numpy(), k.numpy(), 2, 2, 1, 1, 2, 1, 2, 1)
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-6)


def test_sde_marginals(golden):
    z = golden("sde.npz")
    t, x = torch.from_numpy(z["t"]), torch.from_numpy(z["x"])
    for name in ("ve_ksphere", "ve_image", "ve_mnist"):
        smin, smax, n = z[f"{name}::params"]
        mean, std = osde.VESDE(float(smin), float(smax), int(n)).marginal_prob(x, t)
        assert torch.equal(mean, torch.from_numpy(z[f"{name}::mean"]))
        assert torch.equal(std, torch.from_numpy(z[f"{name}::std"]))
    mean, std = osde.VPSDE(0.1, 20., 1000).marginal_prob(x, t)
    assert torch.equal(mean, torch.from_numpy(z["vp::mean"]))
    assert torch.equal(std, torch.from_numpy(z["vp::std"]))


def test_sde_marginals_of_the_two_other_sdes(golden):
    """subVPSDE and SNRSDE (configure_sde builds both, get_score_fn evaluates both): reference outputs."""
    z = golden("sde_extra.npz")
    t, x = torch.from_numpy(z["t"]), torch.from_numpy(z["x"])
    mean, std = osde.subVPSDE(0.1, 20., 1000).marginal_prob(x, t)
    assert torch.equal(mean, torch.from_numpy(z["subvp::mean"])) and torch.equal(std, torch.from_numpy(z["subvp::std"]))
    mean, std = osde.SNRSDE(1000).marginal_prob(x, t)
    assert torch.equal(mean, torch.from_numpy(z["snr::mean"])) and torch.equal(std, torch.from_numpy(z["snr::std"]))
    assert callable(osde.get_score_fn(osde.SNRSDE(1000), None))         # models/utils.py:270-277

    class VVSDE:                                                         # anything else is refused, models/utils.py:279-280
        pass
    with pytest.raises(NotImplementedError, match="SDE class VVSDE not yet supported"):
        osde.get_score_fn(VVSDE(), None)


def test_fcn_score_fn_tiny(golden):
    z = golden("fcn_tiny.npz")
    model = omodels.create_model(fcn_config(hidden_nodes=64))
    model.load_state_dict(state_dict_from_golden(z), strict=True)
    score_fn = osde.get_score_fn(osde.VESDE(1e-2, 4, 1000), model)
    with torch.no_grad():
        y = score_fn(torch.from_numpy(z["x"]), torch.from_numpy(z["t"]))
    assert torch.equal(y, torch.from_numpy(z["score"]))


def test_fcn_full_size_from_seed(golden):
    """Full-size fcn (10dim.py:97-103): weights are rebuilt from torch.manual_seed(0), not stored."""
    z = golden("fcn_full_seed0.npz")
    torch.manual_seed(0)
    model = omodels.create_model(fcn_config(hidden_nodes=2048))
    sums = np.array([float(v.double().abs().sum()) for v in model.state_dict().values()])
    np.testing.assert_allclose(sums, z["weight_abs_sums"], rtol=1e-12)
    score_fn = osde.get_score_fn(osde.VESDE(1e-2, 4, 1000), model)
    with torch.no_grad():
        y = score_fn(torch.from_numpy(z["x"]), torch.from_numpy(z["t"]))
    torch.testing.assert_close(y, torch.from_numpy(z["score"]), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("variant", ["bench_init0", "bench_init1", "ddpm_outskip", "biggan_nofir",
                                     "biggan_outskip_sum"])
def test_ncsnpp_score_fn(golden, variant):
    z = golden(f"ncsnpp_{variant}.npz")
    cfg = ncsnpp_config(**overrides_from_golden(z))
    model = omodels.create_model(cfg)
    assert len(model.all_modules) == int(z["n_modules"])
    model.load_state_dict(state_dict_from_golden(z), strict=True)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    with torch.no_grad():
        raw = model.eval()(x, t * 999)
        y = osde.get_score_fn(osde.VESDE(0.01, 50, 1000), model)(x, t)
    # same ATen kernels on the same host; the einsum contraction order is the only freedom
    assert rel_err(raw, z["model_out"]) < 2e-6
    assert rel_err(y, z["score"]) < 2e-6


@pytest.mark.parametrize("variant", ["paper_like", "plain_resample"])
def test_beatgans_score_fn(golden, variant):
    z = golden(f"beatgans_{variant}.npz")
    model = omodels.create_model(beatgans_config(**overrides_from_golden(z)))
    model.load_state_dict(state_dict_from_golden(z), strict=True)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    with torch.no_grad():
        raw = model.eval()(x, t * 999)
        y = osde.get_score_fn(osde.VESDE(0.01, 50, 1000), model)(x, t)
    assert rel_err(raw, z["model_out"]) < 2e-6
    assert rel_err(y, z["score"]) < 2e-6


@pytest.mark.parametrize("variant", ["mnist_like", "pool_resample"])
def test_ddpm_score_fn(golden, variant):
    z = golden(f"ddpm_{variant}.npz")
    model = omodels.create_model(ddpm_config(**overrides_from_golden(z)))
    assert len(model.all_modules) == int(z["n_modules"])
    model.load_state_dict(state_dict_from_golden(z), strict=True)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    with torch.no_grad():
        raw = model.eval()(x, t * 999)
        y = osde.get_score_fn(osde.VESDE(0.009, 50, 1000), model)(x, t)
    assert rel_err(raw, z["model_out"]) < 2e-6
    assert rel_err(y, z["score"]) < 2e-6


def test_reference_broken_switches_are_reported():
    with pytest.raises(NotImplementedError):
        omodels.create_model(ncsnpp_config(**{"model.progressive": "residual"}))
    with pytest.raises(NotImplementedError):
        omodels.create_model(ncsnpp_config(**{"model.resblock_type": "ddpm"}))  # fir + resamp_with_conv upsample


def test_ksphere_data(golden):
    z = golden("ksphere.npz")
    for k in (10, 50):
        torch.manual_seed(42)
        data = oks.ksphere_data(32, 100, k)
        torch.testing.assert_close(data, torch.from_numpy(z[f"k{k}::data"]), rtol=0, atol=1e-6)
        q = oks.isometry(100, k)
        # the embedded points live on the unit sphere of span(Q)
        torch.testing.assert_close(torch.linalg.norm(data @ q, dim=1), torch.ones(32), rtol=0, atol=1e-5)


def test_spectrum_and_id_rule(golden):
    z = golden("svd_rule.npz")
    for i in range(int(z["n_mats"])):
        s_mat = torch.from_numpy(z[f"m{i}::S"])
        s = odim.spectrum(s_mat)
        ref32, ref64 = z[f"m{i}::sv_ref_f32"], z[f"m{i}::sv_f64"]
        np.testing.assert_allclose(s.numpy(), ref32, rtol=1e-5)
        np.testing.assert_allclose(odim.spectrum_f64(s_mat).numpy(), ref64, rtol=1e-10)
        assert odim.estimate_dim(s.tolist()) == int(z[f"m{i}::dim"]) == int(z[f"m{i}::k_true"])
    for i in range(int(z["n_rules"])):
        assert odim.estimate_dim(z[f"r{i}::s"].tolist()) == int(z[f"r{i}::dim"])
    svd = {"singular_values": [z["r0::s"].tolist(), z["agg::s1"].tolist()]}
    for mode in ("first", "mean", "all"):
        assert odim.estimate_dims(svd, mode) == [int(v) for v in z[f"agg::{mode}"]]


def test_batching_arithmetic():
    # dim_reduction.py:166-171 on the three BASELINE workloads
    assert odim.batching((100,), 500) == (4, 1, 1501)
    assert odim.batching((3, 32, 32), 128) == (36, 0, 4480)
    assert odim.batching((3, 64, 64), 128) == (132, 0, 16768)


@pytest.mark.parametrize("k", [10, 50])
def test_cfg1_exact_score_recovers_id_on_cpu(k):
    """BASELINE config 1 (CPU plumbing): the reference recipe with the exact k-sphere score gives ID = k."""
    torch.manual_seed(42)
    data = oks.ksphere_data(2000, 100, k)
    sde = osde.VESDE(1e-2, 4, 1000)
    model = oks.KSphereExact(100, k, 1e-2, 4)
    score_fn = osde.get_score_fn(sde, model)
    loader = [data[i:i + 500] for i in range(0, 2000, 500)]
    out = odim.get_manifold_dimension(score_fn, sde, 1e-5, loader, 500, num_datapoints=3,
                                      generator=torch.Generator().manual_seed(0))
    assert len(out["singular_values"]) == 2 and len(out["singular_values"][0]) == 100
    assert odim.estimate_dims(out) == [k, k]


def test_vp_score_fn(golden):
    """VP branch of get_score_fn (models/utils.py:238-255) + the VP perturbation the driver feeds it (:180-182)."""
    z = golden("ncsnpp_vp.npz")
    w = golden(str(z["weights_of"]))
    cfg = ncsnpp_config(**overrides_from_golden(w))
    cfg.training.sde = "vpsde"
    cfg.model.beta_min, cfg.model.beta_max = 0.1, 20.
    model = omodels.create_model(cfg)
    model.load_state_dict(state_dict_from_golden(w), strict=True)
    sde, eps = osde.make_sde(cfg)
    assert isinstance(sde, osde.VPSDE) and eps == 1e-3
    x, t, noise = torch.from_numpy(z["x"]), torch.from_numpy(z["t"]), torch.from_numpy(z["z"])
    mean, std = sde.marginal_prob(x, t)
    assert torch.equal(mean, torch.from_numpy(z["mean"])) and torch.equal(std, torch.from_numpy(z["std"]))
    perturbed = mean + std[(...,) + (None,) * 3] * noise
    assert torch.equal(perturbed, torch.from_numpy(z["perturbed"]))
    with torch.no_grad():
        y = osde.get_score_fn(sde, model)(perturbed, t)
    assert rel_err(y, z["score"]) < 2e-6


def test_snr_score_fn(golden):
    """SNR branch of get_score_fn (models/utils.py:270-277) + SNRSDE.marginal_prob (sde_lib.py:175-180) against the
    reference's own score_fn output on the stored nf = 8 NCSN++ weights at t = 1e-3 / 0.2 / 0.7."""
    z = golden("ncsnpp_snr.npz")
    w = golden(str(z["weights_of"]))
    cfg = ncsnpp_config(**overrides_from_golden(w))
    cfg.training.sde = "snrsde"
    model = omodels.create_model(cfg)
    model.load_state_dict(state_dict_from_golden(w), strict=True)
    sde, eps = osde.make_sde(cfg)
    assert isinstance(sde, osde.SNRSDE) and eps == 1e-3
    x, t, noise = torch.from_numpy(z["x"]), torch.from_numpy(z["t"]), torch.from_numpy(z["z"])
    mean, std = sde.marginal_prob(x, t)
    assert torch.equal(mean, torch.from_numpy(z["mean"])) and torch.equal(std, torch.from_numpy(z["std"]))
    perturbed = mean + std[(...,) + (None,) * 3] * noise
    assert torch.equal(perturbed, torch.from_numpy(z["perturbed"]))
    with torch.no_grad():
        y = osde.get_score_fn(sde, model)(perturbed, t)
    assert rel_err(y, z["score"]) < 2e-6


def test_wide_ncsnpp_score_fn(golden):
    """nf = 128: GroupNorm's 32-group cap and Winograd-eligible widths, against reference output (weights from seed)."""
    z = golden("ncsnpp_wide.npz")
    model = omodels.create_model(ncsnpp_config(**overrides_from_golden(z)))
    assert len(model.all_modules) == int(z["n_modules"])
    fill_from_seed(model, int(z["seed"]))
    np.testing.assert_allclose(weight_abs_sums(model), z["weight_abs_sums"], rtol=1e-12)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    with torch.no_grad():
        raw = model.eval()(x, t * 999)
        y = osde.get_score_fn(osde.VESDE(0.01, 50, 1000), model)(x, t)
    assert rel_err(raw, z["model_out"]) < 2e-6
    assert rel_err(y, z["score"]) < 2e-6


def test_wide_beatgans_score_fn(golden):
    z = golden("beatgans_wide.npz")
    model = omodels.create_model(beatgans_config(**overrides_from_golden(z)))
    fill_from_seed(model, int(z["seed"]))
    np.testing.assert_allclose(weight_abs_sums(model), z["weight_abs_sums"], rtol=1e-12)
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    with torch.no_grad():
        raw = model.eval()(x, t * 999)
        y = osde.get_score_fn(osde.VESDE(0.01, 50, 1000), model)(x, t)
    assert rel_err(raw, z["model_out"]) < 2e-6
    assert rel_err(y, z["score"]) < 2e-6


def test_conditional_manifold_dimension(golden):
    """oracle.dim.get_conditional_manifold_dimension against the reference's own function run on the same model, the
    same labelled batch and the same global-RNG noise stream (tests/golden/make_golden.py::gen_conditional)."""
    z = golden("conditional.npz")
    model = omodels.create_model(ncsnpp_config(**overrides_from_golden(z)))
    model.load_state_dict(state_dict_from_golden(z), strict=True)
    sde = osde.VESDE(0.01, 50, 1000)
    loader = [(torch.from_numpy(z["val_images"]), torch.from_numpy(z["val_labels"]))]
    torch.manual_seed(int(z["seed"]))
    out = odim.get_conditional_manifold_dimension(osde.get_score_fn(sde, model), sde, 1e-5, loader,
                                                  num_datapoints=int(z["num_datapoints"]))
    assert ['%.3f' % lv["t"] for lv in out] == [str(d) for d in z["level_dirs"]]
    for i, lv in enumerate(out):
        assert lv["labels"] == z["labels"][i].tolist()
        np.testing.assert_array_equal(lv["images"], z["images_pkl"])
        # same draws, same ATen kernels; gesdd of a matrix that differs in the last bits
        np.testing.assert_allclose(np.array(lv["singular_values"]), z["singular_values"][i], rtol=2e-5)


def test_native_ops_in_float64_and_float16(golden):
    """The oracle's restatements of the two ops in the other dtypes of the reference's dispatch against the reference's own
    CPU branches run in that dtype (tests/golden/ops_dtypes.npz): upfirdn2d bit for bit (the same ATen convolution), the
    fused op's CPU branch bit for bit, and the numpy restatement of the CUDA kernel's arithmetic (alpha / scale rounded to
    fp32 and then to the dtype, every operation rounded) within the difference that rounding of alpha makes: one half ulp
    in float16, 2e-8 relative in float64."""
    z = golden("ops_dtypes.npz")
    for name, tol in (("f64", 2e-8), ("f16", 2.5e-3)):
        for i in range(int(z["n_cases"])):
            up, down, p0, p1 = [int(v) for v in z[f"ufd::c{i}::params"]]
            x, k = torch.from_numpy(z[f"ufd::{name}::c{i}::x"]), torch.from_numpy(z[f"ufd::{name}::c{i}::k"])
            y = oops.upfirdn2d(x, k, up=up, down=down, pad=(p0, p1))
            assert y.dtype == x.dtype and torch.equal(y, torch.from_numpy(z[f"ufd::{name}::c{i}::y"]))
        for i in range(int(z["fba::n_cases"])):
            x, b = torch.from_numpy(z[f"fba::{name}::c{i}::x"]), torch.from_numpy(z[f"fba::{name}::c{i}::b"])
            ref = z[f"fba::{name}::c{i}::y_scale1.25"]
            assert torch.equal(oops.fused_leaky_relu(x, b, 0.2, 1.25), torch.from_numpy(ref))
            nat = oops.fused_bias_act_native(x, b, None, 3, 0, 0.2, 1.25)
            assert nat.dtype == ref.dtype
            np.testing.assert_allclose(nat.astype(np.float64), ref.astype(np.float64), rtol=tol, atol=tol * 1e-3)


def test_philox_restatement_known_answers_and_cfg3_fixture_rows(golden):
    """oracle/philox.py (the device's noise stream restated for the CPU): Random123's known answer for Philox4x32-10 on the zero counter /
    zero key, N(0, 1) moments, independence of how rows are cut; and the full-size config-3 fixture is what the oracle network gives on
    those draws (two of its 192 rows recomputed here)."""
    import numpy as np
    import torch
    from oracle import philox, models as omodels, sde as osde
    from golden.make_cfg3_point import T, cfg3, data_point, oracle_model, point_seed
    w = philox.philox4x32_10(np.zeros(1, np.uint32), np.zeros(1, np.uint32), 0)
    assert [int(v[0]) for v in w] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]      # Random123 kat_vectors: philox4x32 10, zeros
    z = philox.normal_rows(99, 3072, 0, 64)
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    np.testing.assert_array_equal(philox.normal_rows(99, 3072, 10, 5), z[10:15])          # a row's draws do not depend on the cut
    fx = golden("cfg3_point.npz")
    cfg = cfg3()
    model = oracle_model(cfg)
    sde = osde.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    x0 = data_point(1)
    rows = fx["rows"][[0, 100]]
    zz = torch.from_numpy(np.stack([philox.normal_rows(point_seed(1), 3072, int(r), 1)[0] for r in rows])).view(2, 3, 32, 32)
    t = torch.ones(2) * T
    mean, std = sde.marginal_prob(x0.unsqueeze(0).repeat(2, 1, 1, 1), t)
    with torch.no_grad():
        got = osde.get_score_fn(sde, model)(mean + std[:, None, None, None] * zz, t).reshape(2, -1)
    ref = torch.from_numpy(fx["S_rows"][[0, 100]])
    assert float((got.double() - ref.double()).norm() / ref.double().norm()) < 1e-5       # batch shape 2 vs 128: fp32 summation order only
