"""MNIST-shaped 32x32x1 ``ddpm`` config (key values of
/root/reference/configs/dimension_estimation/paper/image_data/MNIST/config.py:28-160: batch 128, VE sigma 0.009..50,
nf 128, ch_mult (1,2,2,4), 4 res-blocks, attention at 16).  The MNIST folder dataset (``data.datamodule='image'``) is
not shipped and torchvision is unavailable, so the data module defaults to synthetic 1-channel images."""
from ......configs.default import get_default_configs
from ......configs.config_dict import ConfigDict


def get_config():
    config = get_default_configs()
    config.logging = ConfigDict(log_path='logs/mnist/', log_name='ddpm', svd_points=10, save_svd=False)
    training = config.training
    training.batch_size = 128
    training.sde = 'vesde'
    training.continuous = True
    config.validation.batch_size = 128
    config.data = ConfigDict(datamodule='image_synthetic', dataset='mnist', data_samples=256, latent_dim=32, data_seed=0,
                             split=[0.8, 0.1, 0.1], image_size=32, effective_image_size=32, shape=[1, 32, 32],
                             centered=False, num_channels=1, use_data_mean=False, return_labels=False)
    config.model = ConfigDict(
        checkpoint_path=None, sigma_min=0.009, sigma_max=50, num_scales=1000, beta_min=0.1, beta_max=20., dropout=0.1,
        embedding_type='fourier', name='ddpm', input_channels=1, output_channels=1, scale_by_sigma=True, ema_rate=0.999,
        normalization='GroupNorm', nonlinearity='swish', nf=128, ch_mult=(1, 2, 2, 4), num_res_blocks=4,
        attn_resolutions=(16,), resamp_with_conv=True, conditional=True, fir=True, fir_kernel=[1, 3, 3, 1],
        skip_rescale=True, resblock_type='biggan', progressive='none', progressive_input='residual',
        progressive_combine='sum', attention_type='ddpm', init_scale=0., fourier_scale=16, conv_size=3)
    return config
