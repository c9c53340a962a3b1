"""``ScoreSpectrumVisualization`` (reference: lightning_callbacks/callbacks.py:403-432): every ``svd_frequency`` epochs run
``get_manifold_dimension`` on the newest checkpoint and log the spectrum plot, the dimension-distribution plot and the
mean estimated dimension.

The class is duck-typed against the hook Lightning calls (``on_validation_epoch_end(trainer, pl_module)``) and against
what it touches on ``pl_module`` (``config``, ``current_epoch``, ``logger.experiment.add_image``, ``log``), so it plugs
into a ``pytorch_lightning.Trainer`` where that package exists and is testable without it.
"""
import logging
import os
import pickle

import numpy as np

from . import utils
from ..plot_utils import plot_distribution, plot_spectrum


@utils.register_callback(name='ScoreSpectrumVisualization')
class ScoreSpectrumVisualization:
    def __init__(self, show_evolution=False, **_unused):
        self.evolution = False   # the reference hard-codes this off (callbacks.py:407)

    @staticmethod
    def _spectra(config, name):
        from ..dim_reduction import get_manifold_dimension
        if config.logging.save_svd:
            get_manifold_dimension(config=config, name=name, return_svd=False)
            path = os.path.join(config.logging.log_path, config.logging.log_name, 'svd', f'{name}.pkl')
            with open(path, 'rb') as f:
                return pickle.load(f)
        return get_manifold_dimension(config=config, name=name, return_svd=True)

    def on_validation_epoch_end(self, trainer, pl_module):
        config = pl_module.config
        if (pl_module.current_epoch + 1) % config.logging.svd_frequency != 0:
            return
        config.model.checkpoint_path = os.path.join(config.logging.log_path, config.logging.log_name,
                                                    "checkpoints/best/last.ckpt")
        try:
            svd = self._spectra(config, f'svd_{pl_module.current_epoch}')
            image = plot_spectrum(svd, return_tensor=True, mode='all')
            image_distro, dims = plot_distribution(svd, return_tensor=True, mode='all')
            pl_module.logger.experiment.add_image('score specturm', image, pl_module.current_epoch)   # tag spelled as upstream
            pl_module.logger.experiment.add_image('dim_distribution', image_distro, pl_module.current_epoch)
            pl_module.log('dim', float(np.mean(dims)), on_step=False, on_epoch=True, prog_bar=True, logger=True)
        except Exception as e:   # the reference swallows and logs (callbacks.py:429-431)
            logging.warning('Could not create a score spectrum')
            logging.error(e)
