"""10-sphere in R^100 with the fcn score model (key values of
/root/reference/configs/dimension_estimation/paper/euclidean_data/ksphere/10dim.py:27-121)."""
from ......configs.default import get_default_configs
from ......configs.config_dict import ConfigDict


def get_config(manifold_dim=10):
    config = get_default_configs()
    config.logging = ConfigDict(log_path='logs/ksphere/', log_name=f'{manifold_dim}-sphere', top_k=5,
                                svd_frequency=50, save_svd=False, svd_points=5)
    training = config.training
    training.batch_size = 500
    training.sde = 'vesde'
    training.continuous = True
    config.validation.batch_size = 500
    config.data = ConfigDict(datamodule='KSphere', create_dataset=False, split=[0.8, 0.1, 0.1], data_samples=50000,
                             use_data_mean=False, n_spheres=1, ambient_dim=100, manifold_dim=manifold_dim,
                             noise_std=0.0, embedding_type='random_isometry', dim=100, num_channels=0, shape=[100])
    config.model = ConfigDict(checkpoint_path=None, sigma_max=4, sigma_min=1e-2, name='fcn', state_size=100,
                              hidden_layers=5, hidden_nodes=2048, dropout=0.0, scale_by_sigma=False, num_scales=1000,
                              ema_rate=0.9999)
    return config
