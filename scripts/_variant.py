"""Variant (diagnostic / A-B) builds of libidiff_hip for the probe scripts: compiled to libidiff_hip.<name>.so by csrc/build.sh, loaded by a
child process through IDIFF_LIB_VARIANT=<name>.  The product library libidiff_hip.so is never rebuilt, patched or replaced by a
probe, so a probe killed half-way (timeout, lease loss) leaves nothing behind that a later test or bench could load by accident."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "id-diff_amd", "csrc", "build.sh")


def build_variant(name, flags, scratch_limit=None):
    """``flags`` for every translation unit ("" = the sources as committed, still a separate file)."""
    env = dict(os.environ, IDIFF_VARIANT=name, IDIFF_VARIANT_FLAGS=flags)
    if scratch_limit is not None:
        env["IDIFF_SCRATCH_LIMIT"] = str(scratch_limit)      # stamps / timing-only kernels may spill; only ever for a variant
    subprocess.run(["bash", BUILD], check=True, stdout=subprocess.DEVNULL, env=env)
    return os.path.join(ROOT, "id-diff_amd", "csrc", f"libidiff_hip.{name}.so")


def run_child(script, name, *args):
    """Run ``script child ...`` with the variant library selected."""
    return subprocess.run([sys.executable, os.path.abspath(script), "child", *args], check=False,
                          env=dict(os.environ, IDIFF_LIB_VARIANT=name))


def remove_variant(name):
    try:
        os.remove(os.path.join(ROOT, "id-diff_amd", "csrc", f"libidiff_hip.{name}.so"))
    except FileNotFoundError:
        pass
