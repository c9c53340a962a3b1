"""CIFAR-10-shaped 32x32x3 NCSN++ (BASELINE configs 3-4).

Model keys are the set every shipped image config of the reference carries
(/root/reference/configs/dimension_estimation/paper/image_data/MNIST/config.py:111-143) with ``name='ncsnpp'``,
``ch_mult=(1, 2, 2, 2)`` and 3 channels: nf 128, 4 res-blocks per level, attention at 16x16,
swish, FIR [1,3,3,1], skip_rescale, biggan blocks, progressive none / input residual, Fourier scale 16.
No dataset ships with the reference, so data are synthetic images (lightning_data_modules/SyntheticImages.py).
"""
from ......configs.default import get_default_configs
from ......configs.config_dict import ConfigDict


def get_config():
    config = get_default_configs()
    config.logging = ConfigDict(log_path='logs/cifar_shaped/', log_name='ncsnpp', svd_points=3, save_svd=False)
    training = config.training
    training.batch_size = 128
    training.sde = 'vesde'
    training.continuous = True
    config.validation.batch_size = 128
    config.data = ConfigDict(datamodule='image_synthetic', dataset='synthetic', data_samples=256, latent_dim=64,
                             data_seed=0, split=[0.8, 0.1, 0.1], image_size=32, effective_image_size=32,
                             shape=[3, 32, 32], centered=False, num_channels=3, use_data_mean=False,
                             return_labels=False)
    config.model = ConfigDict(
        checkpoint_path=None, sigma_min=0.01, sigma_max=50, num_scales=1000, beta_min=0.1, beta_max=20.,
        dropout=0.1, embedding_type='fourier', name='ncsnpp', scale_by_sigma=True, ema_rate=0.999,
        normalization='GroupNorm', nonlinearity='swish', nf=128, ch_mult=(1, 2, 2, 2), num_res_blocks=4,
        attn_resolutions=(16,), resamp_with_conv=True, conditional=True, fir=True, fir_kernel=[1, 3, 3, 1],
        skip_rescale=True, resblock_type='biggan', progressive='none', progressive_input='residual',
        progressive_combine='sum', attention_type='ddpm', init_scale=0., fourier_scale=16, conv_size=3)
    return config
