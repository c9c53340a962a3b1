"""StyleGAN-64d images, 64x64x3 BeatGANs U-Net (BASELINE config 5).

Key values of /root/reference/configs/dimension_estimation/extra_experiments/styleGAN/style_gan_base.py:22-95,
style_gan_BeatGAN.py:19-82 and style_gan_64d_BeatGAN.py:18-24.  The authors' ``gan_64d_train.npy`` is not in the
repository (GanDataset.py:19), so THIS config sets ``data.synthetic = True`` and the ``Gan`` data module generates 64x64
images from a fixed smooth decoder of 64-dimensional latents.  With the real file: set ``data.synthetic = False`` and
``data.data_path`` (the reference's keys ``data_path`` / ``style_gan`` / ``latent_dim`` are honoured as in GanDataset.py:14-22;
a missing file raises).
"""
from ....default import get_default_configs
from ....config_dict import ConfigDict


def get_config():
    config = get_default_configs()
    latent_dim = 64
    config.logging = ConfigDict(log_path='logs/style_gan/', log_name=f'{latent_dim}_BeatGANsUNetModel_dropout_0.3',
                                svd_points=3, save_svd=False)
    training = config.training
    training.batch_size = 128
    training.sde = 'vesde'
    training.continuous = True
    config.validation.batch_size = 256
    config.data = ConfigDict(datamodule='Gan', dataset='style_gan', synthetic=True, data_path=None, style_gan=True, data_samples=64, latent_dim=latent_dim,
                             data_seed=0, split=[0.8, 0.1, 0.1], image_size=64, effective_image_size=64,
                             shape=[3, 64, 64], centered=False, num_channels=3, use_data_mean=False,
                             return_labels=False)
    config.model = ConfigDict(
        checkpoint_path=None, sigma_min=0.01, sigma_max=50, num_scales=1000, beta_min=0.1, beta_max=20.,
        name='BeatGANsUNetModel', ema_rate=0.9999, image_size=64, in_channels=3, model_channels=128, out_channels=3,
        num_res_blocks=2, num_input_res_blocks=None, embed_channels=latent_dim, attention_resolutions=(16,),
        time_embed_channels=None, dropout=0.3, channel_mult=(1, 1, 2, 3, 4), input_channel_mult=None,
        conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, num_heads=1, num_head_channels=-1,
        num_heads_upsample=-1, resblock_updown=True, use_new_attention_order=False, resnet_two_cond=False,
        resnet_cond_channels=None, resnet_use_zero_module=True, attn_checkpoint=False)
    return config
