"""Script drop-in for /root/reference/get_dim.py:1-11: read a config, point it at a checkpoint, choose the
number of points, run the estimator, print the IDs.  Arguments replace the reference's hard-coded paths."""
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import id_diff_amd  # noqa: F401

from id_diff_amd.configs.utils import read_config  # noqa: E402
from id_diff_amd.dim_reduction import get_manifold_dimension  # noqa: E402
from id_diff_amd.plot_utils import plot_dims  # noqa: E402

if __name__ == "__main__":
    config_path = sys.argv[1] if len(sys.argv) > 1 else 'configs/dimension_estimation/paper/euclidean_data/ksphere/10dim.py'
    config = read_config(config_path)
    if len(sys.argv) > 2:
        config.model.checkpoint_path = sys.argv[2]
    config.dim_estimation.num_datapoints = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    svd = get_manifold_dimension(config, return_svd=True)
    print(plot_dims(svd)[1])
