"""Dispatch layer (the two lines of /root/reference/run_lib.py:324-328 that are on the hot path)."""
from . import dim_reduction


def get_manifold_dimension(config, name=None):
    dim_reduction.get_manifold_dimension(config, name)


def get_conditional_manifold_dimension(config, name=None):
    dim_reduction.get_conditional_manifold_dimension(config, name)
