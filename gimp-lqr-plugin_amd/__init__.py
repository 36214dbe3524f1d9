"""gimp-lqr-plugin_amd: MI355X-native seam-carving engine behind the LqrCarver C ABI.

The product is the shared library ``liblqr-hip.so`` built from

  csrc/k_*.hip         hand-written gfx950 kernels, one translation unit per stage (csrc/lqr_common.h has the map)
  csrc/lqr_shim.hip    the lqrhip_* C-ABI shim that launches them
  host/lqr_carver.c    the LqrCarver API (include/lqr.h) in plain C

This Python package is only the build/load helper used by bench.py, the tests
and __graft_entry__.py: it is loaded under the module name
``gimp_lqr_plugin_amd`` (the directory name has a hyphen, as the repo layout
prescribes).  There is no Python or CPU fallback for the carving path: `load()`
raises if the library is missing, and the library itself refuses to create a
carver without a HIP device.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "liblqr-hip.so")


def build(verbose=False):
    """Cross-compile the engine for gfx950 (works without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", HERE, "-j8"], stdout=out)
    return LIB_PATH


def load():
    """dlopen the engine; RTLD_GLOBAL is not needed, the ABI is bound via ctypes."""
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("liblqr-hip.so is not built: run `make -C %s` "
                           "(or __graft_entry__.build()); there is no fallback path" % HERE)
    return ctypes.CDLL(LIB_PATH)


def shard_indices(n_items, rank, world_size):
    """Image i of a batch goes to rank i mod world_size: independent
    images, no data-path collective."""
    return [i for i in range(n_items) if i % world_size == rank]
