"""Host-side logic of the product (no GPU): config dict, registries, batching, rule, sharding, drop-in names."""
import numpy as np
import pytest
import torch

import id_diff_amd
from helpers import beatgans_config, ddpm_config, fake_ml_collections, fcn_config, ncsnpp_config
from id_diff_amd import dim_reduction, parallel, plot_utils, sde_lib
from id_diff_amd.configs.config_dict import ConfigDict
from id_diff_amd.configs.utils import read_config
from id_diff_amd.models import utils as mutils
from oracle import dim as odim, models as omodels


def test_config_dict_dotted_access():
    c = ConfigDict()
    c.logging = ConfigDict(svd_points=5)
    c.dim_estimation = ConfigDict()
    assert hasattr(c, 'logging.svd_points') and not hasattr(c, 'dim_estimation.num_datapoints')
    assert c.get('dim_estimation.num_datapoints', 26) == 26
    c.dim_estimation.num_datapoints = 7
    assert dim_reduction._num_datapoints(c) == 7  # takes precedence, dim_reduction.py:144-147
    del c.dim_estimation['num_datapoints']
    assert dim_reduction._num_datapoints(c) == 5
    with pytest.raises(NameError):
        dim_reduction._num_datapoints(ConfigDict())


def test_batching_matches_oracle():
    for shape, b in [((100,), 500), ((3, 32, 32), 128), ((3, 64, 64), 128), ((1, 28, 28), 100), ((3, 32, 32), 127)]:
        assert dim_reduction.batching(shape, b) == odim.batching(shape, b)


def test_rule_matches_golden(golden):
    z = golden("svd_rule.npz")
    for i in range(int(z["n_rules"])):
        assert plot_utils.estimate_dim(z[f"r{i}::s"].tolist()) == int(z[f"r{i}::dim"])
    for i in range(int(z["n_mats"])):
        assert plot_utils.estimate_dim(z[f"m{i}::sv_ref_f32"].tolist()) == int(z[f"m{i}::dim"])
    svd = {"singular_values": [z["r0::s"].tolist(), z["agg::s1"].tolist()]}
    for mode in ("first", "mean", "all"):
        assert plot_utils.plot_distribution(svd, mode) == [int(v) for v in z[f"agg::{mode}"]]
    assert plot_utils.plot_dims(svd)[1] == [int(v) for v in z["agg::all"]]


def test_sde_matches_golden(golden):
    z = golden("sde.npz")
    t, x = torch.from_numpy(z["t"]), torch.from_numpy(z["x"])
    mean, std = sde_lib.VESDE(1e-2, 4, 1000).marginal_prob(x, t)
    assert torch.equal(std, torch.from_numpy(z["ve_ksphere::std"])) and torch.equal(mean, x)
    mean, std = sde_lib.VPSDE(0.1, 20., 1000).marginal_prob(x, t)
    assert torch.equal(mean, torch.from_numpy(z["vp::mean"])) and torch.equal(std, torch.from_numpy(z["vp::std"]))
    cfg = fcn_config()
    sde, eps = sde_lib.configure_sde(cfg)
    assert isinstance(sde, sde_lib.VESDE) and eps == 1e-5
    # the two other SDEs configure_sde builds (BaseSdeGenerativeModel.py:33-35, 44-46), against reference outputs
    e = golden("sde_extra.npz")
    mean, std = sde_lib.subVPSDE(0.1, 20., 1000).marginal_prob(x, t)
    assert torch.equal(mean, torch.from_numpy(e["subvp::mean"])) and torch.equal(std, torch.from_numpy(e["subvp::std"]))
    mean, std = sde_lib.SNRSDE(1000).marginal_prob(x, t)
    assert torch.equal(mean, torch.from_numpy(e["snr::mean"])) and torch.equal(std, torch.from_numpy(e["snr::std"]))
    cfg.training.sde = "subvpsde"; cfg.model.beta_min, cfg.model.beta_max = 0.1, 20.
    sde, eps = sde_lib.configure_sde(cfg)
    assert type(sde) is sde_lib.subVPSDE and eps == 1e-3
    cfg.training.sde = "snrsde"
    sde, eps = sde_lib.configure_sde(cfg)
    assert type(sde) is sde_lib.SNRSDE and eps == 1e-3
    from id_diff_amd.models import utils as mutils
    assert callable(mutils.get_score_fn(sde, None))                     # SNR branch, models/utils.py:270-277 of the reference

    class VVSDE(sde_lib.SDE):                                            # a class outside VP / subVP / VE / SNR: :279-280
        pass
    with pytest.raises(NotImplementedError, match="SDE class VVSDE not yet supported"):
        mutils.get_score_fn(VVSDE(1000), None)


def test_state_dict_keys_match_reference(golden):
    """Same module list / parameter names / shapes as the reference => its checkpoints load."""
    from helpers import overrides_from_golden, state_dict_from_golden
    for variant in ["bench_init0", "ddpm_outskip", "biggan_nofir", "biggan_outskip_sum"]:
        z = golden(f"ncsnpp_{variant}.npz")
        model = mutils.create_model(ncsnpp_config(**overrides_from_golden(z)))
        sd = state_dict_from_golden(z)
        own = model.state_dict()
        assert sorted(own) == sorted(sd)
        assert all(own[k].shape == sd[k].shape for k in sd)
        model.load_state_dict(sd, strict=True)
    z = golden("fcn_tiny.npz")
    model = mutils.create_model(fcn_config(hidden_nodes=64))
    model.load_state_dict(state_dict_from_golden(z), strict=True)
    for variant in ["mnist_like", "pool_resample"]:
        z = golden(f"ddpm_{variant}.npz")
        model = mutils.create_model(ddpm_config(**overrides_from_golden(z)))
        model.load_state_dict(state_dict_from_golden(z), strict=True)
    for variant in ["paper_like", "plain_resample"]:
        z = golden(f"beatgans_{variant}.npz")
        model = mutils.create_model(beatgans_config(**overrides_from_golden(z)))
        model.load_state_dict(state_dict_from_golden(z), strict=True)


def test_benchmark_model_size():
    cfg = read_config('configs/dimension_estimation/paper/image_data/cifar_shaped/ncsnpp.py')
    model = mutils.create_model(cfg)
    assert len(model.all_modules) == 59
    assert sum(p.numel() for p in model.parameters()) == 62758915
    oracle_model = omodels.create_model(cfg)
    assert sorted(oracle_model.state_dict()) == sorted(model.state_dict())


def test_config5_model_size():
    cfg = read_config('configs/dimension_estimation/extra_experiments/styleGAN/style_gan_64d_BeatGAN.py')
    model = mutils.create_model(cfg)
    n = sum(p.numel() for p in model.parameters())
    assert abs(n / 1e6 - 87.5) < 0.5, n     # 87.5 M parameters
    assert sorted(omodels.create_model(cfg).state_dict()) == sorted(model.state_dict())


def test_models_fail_loudly_on_cpu():
    model = mutils.create_model(fcn_config(hidden_nodes=64))
    with pytest.raises(RuntimeError, match="no CPU path"):
        model(torch.zeros(2, 100), torch.zeros(2))


def test_round_robin_sharding_is_a_partition():
    for n in (1, 4, 7, 16):
        for w in (1, 2, 4, 8):
            owned = [parallel.my_points(n, r, w) for r in range(w)]
            assert sorted(sum(owned, [])) == list(range(n))
            assert max(len(o) for o in owned) == (n + w - 1) // w


def test_dropin_names():
    import sys
    names = id_diff_amd.install_dropin()
    from op import upfirdn2d, fused_leaky_relu, FusedLeakyReLU  # noqa: F401  (the reference's import line)
    import dim_reduction as dr
    assert dr.get_manifold_dimension is dim_reduction.get_manifold_dimension
    for n in names:
        sys.modules.pop(n, None)


def test_ksphere_datamodule_matches_golden(golden):
    from id_diff_amd.lightning_data_modules.KSphereDataset import KSphereDataset
    z = golden("ksphere.npz")
    cfg = ConfigDict()
    cfg.data = ConfigDict(data_samples=32, n_spheres=1, ambient_dim=100, manifold_dim=10, noise_std=0.0,
                          embedding_type="random_isometry")
    torch.manual_seed(42)
    torch.testing.assert_close(KSphereDataset(cfg).data, torch.from_numpy(z["k10::data"]), rtol=0, atol=1e-6)


def test_collect_points_stops_one_short():
    loader = [torch.zeros(3, 4), torch.ones(3, 4), torch.full((3, 4), 2.)]
    pts = dim_reduction.collect_points(loader, 5)            # idx+1 >= 5 -> 4 points
    assert len(pts) == 4 and pts[3][0][0].item() == 1.0 and pts[0][1] == 3


def test_bessel_ratio_asymptotics():
    from scipy.special import ive
    for p in (11, 51, 101):
        for kappa in (5e3, 1e4, 1e5):
            approx = 1 - (p - 1) / (2 * kappa) + (p - 1) * (p - 3) / (8 * kappa ** 2) + (p - 1) * (p - 3) / (8 * kappa ** 3)
            exact = ive(p / 2, kappa) / ive(p / 2 - 1, kappa)
            assert abs(approx - exact) < 5e-9, (p, kappa, approx, exact)  # O(p^4 / kappa^4)


def test_plot_spectrum_and_distribution_render():
    """§8-f rank 1: the figures either side of the path (plot_utils.py:111-139, :158-195)."""
    rng = np.random.default_rng(0)
    spectra = [np.sort(rng.random(40))[::-1].tolist() for _ in range(3)]
    svd = {'singular_values': spectra}
    fig = plot_utils.plot_spectrum(svd, mode='all', ground_truth=[10, 12])
    ax = fig.axes[0]
    assert len(ax.lines) == 3 + 2                       # three spectra + two ground-truth markers
    assert list(ax.lines[0].get_xdata()) == [30, 30]    # x = len - ground_truth
    np.testing.assert_allclose(ax.lines[2].get_ydata(), spectra[0])
    import matplotlib.pyplot as plt
    plt.close(fig)
    img = plot_utils.plot_spectrum(svd, return_tensor=True, mode='first')
    assert img.dtype == torch.float32 and img.shape[0] == 3 and img.shape[1] > 100 and 0.0 <= float(img.min()) <= float(img.max()) <= 1.0
    image, dims = plot_utils.plot_distribution(svd, return_tensor=True, mode='all')
    assert image.shape[0] == 3 and dims == [plot_utils.estimate_dim(s) for s in spectra]
    assert plot_utils.plot_distribution(svd, mode='all') == dims


def test_score_spectrum_visualization_callback(monkeypatch, tmp_path):
    """lightning_callbacks/callbacks.py:403-432: every svd_frequency epochs -> two images and the mean dimension."""
    from id_diff_amd.lightning_callbacks import utils as cutils
    cb = cutils.get_callback_by_name('ScoreSpectrumVisualization')(show_evolution=True)
    rng = np.random.default_rng(1)
    svd = {'singular_values': [np.sort(rng.random(30))[::-1].tolist() for _ in range(4)]}
    calls = {}

    def fake_gmd(config, name=None, return_svd=False):
        calls['args'] = (config.model.checkpoint_path, name, return_svd)
        return svd

    monkeypatch.setattr(dim_reduction, "get_manifold_dimension", fake_gmd)

    class Experiment:
        def __init__(self): self.images = []
        def add_image(self, tag, img, step): self.images.append((tag, tuple(img.shape), step))

    class Module:
        def __init__(self, epoch):
            self.config = ConfigDict(logging=ConfigDict(svd_frequency=5, save_svd=False, log_path=str(tmp_path), log_name='run'),
                                     model=ConfigDict(checkpoint_path=None))
            self.current_epoch = epoch
            self.logger = type("L", (), {})()
            self.logger.experiment = Experiment()
            self.logged = {}
        def log(self, key, value, **kw): self.logged[key] = value

    quiet = Module(2)
    cb.on_validation_epoch_end(None, quiet)
    assert not quiet.logger.experiment.images and 'args' not in calls
    due = Module(4)
    cb.on_validation_epoch_end(None, due)
    assert calls['args'] == (str(tmp_path / 'run' / 'checkpoints/best/last.ckpt'), 'svd_4', True)
    tags = [t for t, _, _ in due.logger.experiment.images]
    assert tags == ['score specturm', 'dim_distribution'] and all(s[0] == 3 for _, s, _ in due.logger.experiment.images)
    assert due.logged['dim'] == pytest.approx(np.mean([plot_utils.estimate_dim(s) for s in svd['singular_values']]))
    with pytest.raises(ValueError):
        cutils.register_callback(name='ScoreSpectrumVisualization')(type(cb))


def test_image_folder_datamodule(tmp_path):
    """§8-f rank 4: the 'image' data module (ImageDatasets.py:25-105) on a folder of PNGs, both transform recipes."""
    from PIL import Image
    from id_diff_amd.lightning_data_modules import utils as dutils
    rng = np.random.default_rng(0)
    folder = tmp_path / 'faces'
    folder.mkdir()
    frames = []
    for i in range(10):
        a = rng.integers(0, 256, size=(218, 178, 3), dtype=np.uint8)
        frames.append(a)
        Image.fromarray(a).save(folder / f'{i:03d}.png')
    cfg = ConfigDict(data=ConfigDict(datamodule='image', base_dir=str(tmp_path), dataset='faces', shape=[3, 32, 32], crop=True,
                                     split=[0.8, 0.1, 0.1]),
                     training=ConfigDict(batch_size=4), eval=ConfigDict(batch_size=2))
    dm = dutils.create_lightning_datamodule(cfg)
    dm.setup()
    assert (len(dm.train_data), len(dm.valid_data), len(dm.test_data)) == (8, 1, 1)
    batch = next(iter(dm.train_dataloader()))
    assert batch.shape == (4, 3, 32, 32) and batch.dtype == torch.float32 and -1.0 <= float(batch.min()) and float(batch.max()) <= 1.0
    # the crop recipe, re-derived: centre 108x108 window, bicubic to 32x32, [-1, 1]
    x0 = dm.dataset[0]
    win = frames[0][55:163, 35:143]
    ref = np.asarray(Image.fromarray(win).resize((32, 32), Image.BICUBIC), dtype=np.float32) / 255.0
    np.testing.assert_allclose(x0.permute(1, 2, 0).numpy(), (ref - 0.5) / 0.5, atol=1e-6)
    cfg.data.crop = False
    cfg.data.shape = [3, 109, 89]
    plain = dutils.create_lightning_datamodule(cfg)
    plain.setup()
    y0 = plain.dataset[0]
    assert y0.shape == (3, 109, 89) and 0.0 <= float(y0.min()) and float(y0.max()) <= 1.0
    cfg.data.dataset = 'mnist'
    with pytest.raises(FileNotFoundError, match="no network"):
        dutils.create_lightning_datamodule(cfg).setup()


def test_gan_datamodule_reads_the_reference_files(tmp_path):
    """The 'Gan' data module = GanDataset.py:9-68: `data.data_path` / `data.latent_dim` / `data.style_gan` name the file
    (style_gan_horvat/gan_<d>d_train.npy, or latent_dim_<d>/data.pt), the tensors come back as stored, the split is
    int(s0*l) / int(s1*l) / rest, and a missing file RAISES -- generated images only on the explicit `data.synthetic = True`."""
    from id_diff_amd.lightning_data_modules import utils as dutils
    from id_diff_amd.lightning_data_modules.GanDataset import GanDataset
    rng = np.random.default_rng(0)
    arr = rng.random((21, 3, 8, 8))                                  # float64 on disk: `.float()` as GanDataset.py:20
    (tmp_path / 'style_gan_horvat').mkdir()
    np.save(tmp_path / 'style_gan_horvat' / 'gan_7d_train.npy', arr)
    pt = torch.randn(10, 3, 8, 8)
    (tmp_path / 'latent_dim_7').mkdir()
    torch.save(pt, tmp_path / 'latent_dim_7' / 'data.pt')
    # the key set of the authors' config (configs/.../styleGAN/style_gan_base.py:78-95)
    cfg = ConfigDict(data=ConfigDict(datamodule='Gan', data_path=str(tmp_path), latent_dim=7, style_gan=True, split=[0.95, 0.05, 0.0],
                                     shape=[3, 8, 8], image_size=8, centered=False, num_channels=3),
                     training=ConfigDict(batch_size=4, workers=4), validation=ConfigDict(batch_size=2, workers=4),
                     eval=ConfigDict(batch_size=2, workers=4))
    ds = GanDataset(cfg)
    assert len(ds) == 21 and ds[3].dtype == torch.float32
    np.testing.assert_array_equal(ds[3].numpy(), arr[3].astype(np.float32))
    torch.manual_seed(5)
    dm = dutils.create_lightning_datamodule(cfg)
    assert type(dm).__name__ == 'SyntheticDataModule'                # the reference's class name for this module
    dm.setup()
    assert (len(dm.train_data), len(dm.valid_data), len(dm.test_data)) == (19, 1, 1)   # int(.95*21), int(.05*21), the rest
    torch.manual_seed(5)
    ref_train = torch.utils.data.random_split(ds, [19, 1, 1])[0]     # same generator state -> same permutation as the reference's call
    assert list(dm.train_data.indices) == list(ref_train.indices)
    assert next(iter(dm.train_dataloader())).shape == (4, 3, 8, 8)
    cfg.data.style_gan = False
    ds_pt = GanDataset(cfg)
    assert len(ds_pt) == 10 and torch.equal(ds_pt[2], pt[2])
    cfg.data.latent_dim = 64                                         # style_gan_64d_BeatGAN.py:18 -- the file is not there
    for flag in (True, False):
        cfg.data.style_gan = flag
        with pytest.raises(FileNotFoundError, match="gan_64d_train.npy" if flag else "latent_dim_64/data.pt"):
            dutils.create_lightning_datamodule(cfg).setup()
    del cfg.data['style_gan']
    with pytest.raises(AttributeError, match="style_gan"):
        dutils.create_lightning_datamodule(cfg).setup()
    # this repo's own config 5 asks for generated images explicitly; an authors' config never carries that key
    own = read_config('configs/dimension_estimation/extra_experiments/styleGAN/style_gan_64d_BeatGAN.py')
    assert own.data.synthetic is True and own.data.datamodule == 'Gan'
    own.data.data_samples = 10
    dm = dutils.create_lightning_datamodule(own)
    dm.setup()
    assert dm.dataset[0].shape == (3, 64, 64)
    own.data.synthetic = False                                       # ... and without it the reference's contract holds
    with pytest.raises(FileNotFoundError, match="data_path"):
        dutils.create_lightning_datamodule(own).setup()


def test_lightning_checkpoint_round_trip(tmp_path, monkeypatch):
    """A checkpoint shaped the way Lightning writes the reference's (BaseSdeGenerativeModel.py:17 save_hyperparameters:
    `score_model.*` keys + a pickled ml_collections.ConfigDict) loads with strict=True although neither ml_collections nor
    pytorch_lightning is importable -- and nothing the file names gets imported or executed."""
    import sys
    from id_diff_amd.lightning_modules import checkpoint_io
    from id_diff_amd.lightning_modules.utils import create_lightning_module
    from id_diff_amd.models import utils as mutils
    mods, MLConfigDict, FieldReference, AttributeDict = fake_ml_collections()
    cfg = fcn_config(hidden_nodes=32, hidden_layers=2)
    cfg.training.lightning_module = 'base'
    torch.manual_seed(3)
    donor = mutils.create_model(cfg)
    state = {'score_model.' + k: v.clone() for k, v in donor.state_dict().items()}
    foreign_cfg = MLConfigDict(model=MLConfigDict(name='fcn', sigma_min=FieldReference(0.01), hidden_nodes=32),
                               training=MLConfigDict(sde='vesde', batch_size=500))
    ckpt = {'epoch': 7, 'global_step': 123, 'pytorch-lightning_version': '1.5.1', 'state_dict': state,
            'hyper_parameters': AttributeDict(config=foreign_cfg), 'optimizer_states': [{'state': {0: {'exp_avg': torch.ones(3)}}}]}
    path, cfg_pkl = str(tmp_path / 'last.ckpt'), str(tmp_path / 'config.pkl')
    with monkeypatch.context() as mp:
        for k, v in mods.items():
            mp.setitem(sys.modules, k, v)
        torch.save(ckpt, path)
        import pickle
        with open(cfg_pkl, 'wb') as f:
            pickle.dump(foreign_cfg, f)
    assert 'ml_collections' not in sys.modules and 'pytorch_lightning' not in sys.modules

    torch.manual_seed(4)
    module = create_lightning_module(cfg)
    before = {k: v.clone() for k, v in module.score_model.state_dict().items()}
    module = module.load_from_checkpoint(path)
    after = module.score_model.state_dict()
    assert all(torch.equal(after[k], donor.state_dict()[k]) for k in after) and any(not torch.equal(before[k], after[k]) for k in after)
    assert 'ml_collections' not in sys.modules and 'pytorch_lightning' not in sys.modules      # nothing got imported

    loaded = checkpoint_io.load_checkpoint(path)
    hp = checkpoint_io.to_config(loaded['hyper_parameters'])
    assert hp.config.model.name == 'fcn' and hp.config.model.sigma_min == 0.01 and hp.config.training.batch_size == 500
    assert torch.equal(loaded['optimizer_states'][0]['state'][0]['exp_avg'], torch.ones(3))
    conf = checkpoint_io.load_config_pickle(cfg_pkl)                                         # main.py --config x.pkl
    assert isinstance(conf, ConfigDict) and conf.model.hidden_nodes == 32 and hasattr(conf, 'training.sde')

    # a plain state-dict file (no Lightning wrapper) and strictness
    torch.save(donor.state_dict(), str(tmp_path / 'plain.pt'))
    create_lightning_module(cfg).load_from_checkpoint(str(tmp_path / 'plain.pt'))
    bad = dict(state)
    bad.pop(next(iter(bad)))
    torch.save({'state_dict': bad}, str(tmp_path / 'bad.ckpt'))
    with pytest.raises(RuntimeError, match="Missing key"):
        create_lightning_module(cfg).load_from_checkpoint(str(tmp_path / 'bad.ckpt'))
    with pytest.raises(FileNotFoundError):
        create_lightning_module(cfg).load_from_checkpoint(str(tmp_path / 'nope.ckpt'))

    # a malicious global is never resolved: os.system would run on a permissive unpickler
    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ('touch ' + str(tmp_path / 'pwned'),))
    torch.save({'state_dict': state, 'hyper_parameters': Evil()}, str(tmp_path / 'evil.ckpt'))
    create_lightning_module(cfg).load_from_checkpoint(str(tmp_path / 'evil.ckpt'))
    assert not (tmp_path / 'pwned').exists()

    # ... nor are the loaders that live NEXT to the tensor-rebuild helpers in torch's own modules: a nested,
    # unrestricted torch.load (`torch.storage._load_from_bytes`), `torch._utils._import_dotted_name`, `torch.serialization.load`
    import io
    inner = io.BytesIO()
    torch.save(Evil(), inner)

    class Nested:
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (inner.getvalue(),))

    class Dotted:
        def __reduce__(self):
            return (torch._utils._import_dotted_name, ('os.system',))

    class SerLoad:
        def __reduce__(self):
            return (torch.serialization.load, (str(tmp_path / 'evil.ckpt'),), {'weights_only': False})

    for i, bad_obj in enumerate((Nested(), Dotted(), SerLoad())):
        for proto in (2, 4):
            f_ckpt, f_pkl = str(tmp_path / f'evil{i}_{proto}.ckpt'), str(tmp_path / f'evil{i}_{proto}.pkl')
            torch.save({'state_dict': state, 'hyper_parameters': bad_obj}, f_ckpt, pickle_protocol=proto)
            with open(f_pkl, 'wb') as f:
                pickle.dump({'model': bad_obj}, f, protocol=proto)
            loaded = checkpoint_io.load_checkpoint(f_ckpt)
            assert isinstance(loaded['hyper_parameters'], checkpoint_io.Foreign)
            assert torch.equal(loaded['state_dict']['score_model.mlp.0.weight'], state['score_model.mlp.0.weight'])
            checkpoint_io.load_config_pickle(f_pkl)
            assert not (tmp_path / 'pwned').exists()


def test_missing_checkpoint_is_an_error_unless_opted_in():
    """dim_reduction.py:128 `load_from_checkpoint(None)` raises in the reference; random weights need an explicit opt-in."""
    from id_diff_amd.lightning_modules.utils import create_lightning_module
    cfg = fcn_config(hidden_nodes=16, hidden_layers=1)
    cfg.training.lightning_module = 'base'
    with pytest.raises(ValueError, match="allow_random_init"):
        create_lightning_module(cfg).load_from_checkpoint(None)
    cfg.model.allow_random_init = True
    create_lightning_module(cfg).load_from_checkpoint(None)
    cfg = read_config('configs/dimension_estimation/paper/euclidean_data/ksphere/10dim.py')
    cfg.model.name = 'ksphere_exact'                    # analytic score: nothing to restore
    create_lightning_module(cfg).load_from_checkpoint(None)


def test_nan_spectra_raise_instead_of_yielding_an_id():
    """The eigensolver reports trouble as NaN (never a silently wrong spectrum); the driver turns that into an error."""
    ok = torch.tensor([[3.0, 2.0, 1.0]])
    assert torch.equal(dim_reduction.checked_spectra(ok), ok)
    with pytest.raises(RuntimeError, match="reported a failure"):
        dim_reduction.checked_spectra(torch.tensor([[3.0, float("nan"), 1.0]]))


def test_winograd_form_routing():
    """Which 3x3 kernel form the executor picks (host logic + the C-ABI's geometry answers, no GPU): GroupNorm-fed convolutions go to the
    fp16-pair F(4x4, 3x3) kernel from one workgroup per CU on, 4x4 maps included; other inputs keep the fp32 contraction and its
    stricter profitability rule; geometries the pair kernel refuses (Cin % 16) fall back; the switch turns the pair form off."""
    from id_diff_amd import _lib
    from id_diff_amd.models import ncsnpp as hip_ncsnpp
    pays = hip_ncsnpp._winograd43_pays
    assert pays(2240, 32, 32, 128, 128, normed=True) and pays(2240, 32, 32, 128, 128)
    assert pays(2240, 4, 4, 256, 256, normed=True) and not pays(2240, 4, 4, 256, 256)          # 280 workgroups of one-tile samples
    assert pays(128, 16, 16, 256, 256, normed=True) and not pays(127, 16, 16, 256, 128, normed=True)   # 256 workgroups (one per CU) is the threshold
    assert not pays(64, 8, 8, 256, 256, normed=True) and not pays(64, 8, 8, 256, 256)          # a test-sized batch: F(2x2) or the implicit GEMM
    assert _lib.conv2d_winograd43_ok(4, 8, 8, 24, 64) and not _lib.conv2d_winograd43h_ok(4, 8, 8, 24, 64)   # Cin % 16
    assert pays(8192, 8, 8, 24, 64, normed=True) == pays(8192, 8, 8, 24, 64)                   # refused by the pair kernel: the fp32 rule decides
    with _lib.thread_option("IDIFF_NO_WINO43H", 1):
        assert not pays(2240, 4, 4, 256, 256, normed=True) and pays(2240, 32, 32, 128, 128, normed=True)
    assert _lib.lib().idiff_winograd43h_weight_floats(128, 256) == 36 * 128 * 256 + 4
