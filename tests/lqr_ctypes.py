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
