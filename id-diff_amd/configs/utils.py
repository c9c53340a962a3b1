"""``read_config`` (drop-in for /root/reference/configs/utils.py:6-10): import ``configs/<path>.py`` by dotted
name and call its ``get_config()``.  The module is resolved inside this package first, then on sys.path (so a
user's own ``configs/`` tree keeps working)."""
import importlib
import importlib.util
import os
import re


def read_config(config_path):
    rel = re.findall(r'configs/[\w|/ | \.]+.py', config_path)[0][:-3]
    dotted = rel.replace('/', '.')
    pkg = __name__.rsplit('.configs', 1)[0]
    for candidate in (f"{pkg}.{dotted}", dotted):
        try:
            module = importlib.import_module(candidate)
            return module.get_config()
        except ModuleNotFoundError:
            continue
    if os.path.exists(config_path):  # a bare file outside any package
        spec = importlib.util.spec_from_file_location("idiff_user_config", config_path)
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        return module.get_config()
    raise ModuleNotFoundError(f"no config module for {config_path!r}")
