"""CLI drop-in for ``python main.py --config <cfg.py|cfg.pkl> --mode manifold_dimension [--checkpoint_path ...]``
(/root/reference/main.py:17-71; absl is not installable here, argparse accepts the same flags incl. ``--flag=value``).

Run it from the repo root as ``python id-diff_amd/main.py ...`` or, for several GPUs of one node,
``python -m torch.distributed.run --nproc-per-node N id-diff_amd/main.py ...``.
"""
import argparse
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import id_diff_amd  # noqa: F401
    __package__ = "id_diff_amd"

from id_diff_amd import parallel, run_lib  # noqa: E402
from id_diff_amd.configs.utils import read_config  # noqa: E402
from id_diff_amd.lightning_modules.checkpoint_io import load_config_pickle  # noqa: E402

_HOT_MODES = ('manifold_dimension', 'conditional_manifold_dimension')


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--config", required=True, help="Training configuration path (.py or .pkl).")
    ap.add_argument("--checkpoint_path", default=None)
    ap.add_argument("--data_path", default=None)
    ap.add_argument("--log_path", default="./")
    ap.add_argument("--mode", required=True)
    ap.add_argument("--eval_folder", default="eval")
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--log_name", default=None)
    ap.add_argument("--gpus", type=int, default=None,
                    help="(not in the reference) shard the data points over this many GPUs of the node: without a launcher "
                         "this process starts one fresh rank process per GPU itself; under torch.distributed.run it must "
                         "equal WORLD_SIZE")
    ap.add_argument("--allow_random_init", action="store_true",
                    help="run on freshly initialised weights when no checkpoint is given (timing / plumbing only)")
    return ap.parse_args(argv)


def main(argv=None):
    flags = parse(argv)
    if flags.config.endswith('pkl'):
        config = load_config_pickle(flags.config)      # ml_collections pickles load without ml_collections
    elif flags.config.endswith('py'):
        config = read_config(flags.config)
    else:
        raise RuntimeError('Unknown config extension. Provide a path to .py or .pkl file.')
    if flags.checkpoint_path is not None:
        config.model.checkpoint_path = flags.checkpoint_path
    if flags.allow_random_init:
        config.model.allow_random_init = True
    if flags.mode not in _HOT_MODES:
        raise SystemExit(f"mode {flags.mode!r} is outside the scope of id-diff_amd (the MI355X build covers "
                         f"{', '.join(_HOT_MODES)}); use the reference for training / sampling / evaluation")
    if flags.gpus is not None:
        need_devices = not str(getattr(config, "device", "cuda")).startswith("cpu")     # the same answer in the launcher and in the ranks
        if flags.gpus > 1 and not parallel.launched():
            # become the launcher: nothing in this process has touched the GPU; fresh rank processes, never an exec
            rc = parallel.launch_local_ranks(os.path.abspath(__file__), list(sys.argv[1:] if argv is None else argv), flags.gpus,
                                             need_devices=need_devices)
            if rc:
                raise SystemExit(rc)
            return
        if parallel.launched():
            parallel.check_world(flags.gpus, int(os.environ["WORLD_SIZE"]), need_devices=need_devices)
    rank, world, local_rank = parallel.init_from_env()
    if world > 1:
        config.device = f"cuda:{parallel.device_ordinal(local_rank)}"
    if flags.mode == 'manifold_dimension':
        run_lib.get_manifold_dimension(config, name=flags.log_name)
    else:
        run_lib.get_conditional_manifold_dimension(config, name=flags.log_name)


if __name__ == "__main__":
    main()
