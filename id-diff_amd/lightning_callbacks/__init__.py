"""Callbacks either side of the hot path (reference: lightning_callbacks/): the registry and ScoreSpectrumVisualization."""
from . import utils, callbacks  # noqa: F401
