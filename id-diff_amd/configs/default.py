"""Defaults shared by the dimension-estimation configs (key names of /root/reference/configs/default.py:5-84;
only the groups the manifold_dimension path reads are filled in)."""
import torch

from .config_dict import ConfigDict


def get_default_configs():
    config = ConfigDict()
    config.logging = ConfigDict(log_path=None, log_name=None, top_k=None, every_n_epochs=None)
    config.training = ConfigDict(lightning_module='base', gpus=1, num_nodes=1, workers=0, continuous=True,
                                 likelihood_weighting=True, reduce_mean=False, sde='vesde', batch_size=128)
    config.validation = ConfigDict(batch_size=500, workers=0)
    config.eval = ConfigDict(batch_size=512, workers=0)
    config.seed = 42
    config.device = torch.device('cuda:0') if torch.cuda.is_available() else torch.device('cpu')
    config.dim_estimation = ConfigDict()
    return config
