"""Callback registry with the reference's function names (lightning_callbacks/utils.py:1-21)."""
from ..registry import Registry

_CALLBACKS = Registry("callback")
register_callback = _CALLBACKS.register
get_callback_by_name = _CALLBACKS.get
