"""Test-side binding: the package's ctypes mirror of the LqrCarver ABI
(gimp-lqr-plugin_amd/binding.py) plus the loaders of the two CHECKERS that only tests may use:
  * the CPU oracle      oracle/liblqr_oracle.so              (prefix "o")
  * a genuine liblqr-1  (prefix "", path from $LQR_REAL_LIB), where one exists.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as _ge

_pkg = _ge._import_package()
import importlib

_b = importlib.import_module("gimp_lqr_plugin_amd.binding")
from gimp_lqr_plugin_amd.binding import *          # noqa: F401,F403  (Api, Carver, engine_api, constants, helpers)
from gimp_lqr_plugin_amd.binding import _apis, _malloc_copy, _libc   # noqa: F401

ORACLE_LIB = os.path.join(ROOT, "oracle", "liblqr_oracle.so")

_engine_api = engine_api


def engine_api():
    """the package's engine_api(), plus the kernels' experiment switches from the environment (soaks and A/B runs of the tests and fuzz
    scripts): LQR_LV_DBG -> lqrhip_band_levels_debug (4 no near copy, 8 an image's slots on different XCDs, 16 timing jitter in an
    LQR_JITTER build), LQR_DPP_DBG -> lqrhip_dp_tile_debug (1 no near copies, 2 round 4's tile numbering, 4 timing jitter)"""
    import ctypes
    api = _engine_api()
    if not getattr(api, "_dbg_env_applied", False):
        api._dbg_env_applied = True
        for env, fn in (("LQR_LV_DBG", "lqrhip_band_levels_debug"), ("LQR_DPP_DBG", "lqrhip_dp_tile_debug")):
            if os.environ.get(env):
                f = getattr(api.lib, fn); f.argtypes = [ctypes.c_int]; f.restype = None
                f(int(os.environ[env]))
    return api


def oracle_api():
    if "oracle" not in _apis:
        _apis["oracle"] = Api(ORACLE_LIB, "o")
    return _apis["oracle"]


def real_liblqr_api():
    """A genuine liblqr-1, if $LQR_REAL_LIB points at one (second oracle)."""
    path = os.environ.get("LQR_REAL_LIB")
    if not path or not os.path.exists(path):
        return None
    api = Api(path, "", LIBLQR_SYMBOLS)
    api.has_ext = False
    return api
