"""Config modules with the reference's key names (``get_config()`` per file, loaded by ``configs.utils.read_config``)."""
from .config_dict import ConfigDict  # noqa: F401
