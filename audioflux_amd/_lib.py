"""Loader for the in-tree C-ABI library libaudioflux_mi355x.so.

The library is the product; this module only finds and dlopens it.  There is
no Python or CPU implementation to fall back to: if the shared object is
missing the import fails with instructions to build it, and if no MI355X is
visible the library's constructors return a negative status which the wrapper
classes turn into RuntimeError.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AFX_LIB") or os.path.join(_HERE, "lib", "libaudioflux_mi355x.so")
_lib = None


def build(verbose=False):
    """Compile csrc/ (gcc + hipcc --offload-arch=gfx950) into lib/."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libaudioflux_mi355x.so failed")
    return LIB_PATH


def get_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `make -C audioflux_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback implementation")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.afx_last_error.restype = ctypes.c_char_p
        _lib.afx_version.restype = ctypes.c_char_p
    return _lib


def last_error():
    return get_lib().afx_last_error().decode(errors="replace")


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status}: {last_error()}")


def runtime_status():
    """0 when a gfx950 device is usable, negative otherwise."""
    return int(get_lib().afx_runtime_status())
