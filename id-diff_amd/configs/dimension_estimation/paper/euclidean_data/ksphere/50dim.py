"""50-sphere in R^100 (the reference's 50dim.py differs from 10dim.py in manifold_dim / log_name only)."""
import importlib

_ten = importlib.import_module(__name__.rsplit('.', 1)[0] + '.10dim')


def get_config():
    return _ten.get_config(manifold_dim=50)
