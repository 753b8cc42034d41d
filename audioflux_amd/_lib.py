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


def _share_torch_hip_runtime():
    """One process must hold ONE HIP runtime.  The PyTorch-ROCm wheel bundles its own
    libamdhip64.so (SONAME libamdhip64.so.7, same as /opt/rocm's, which our library
    links).  If both copies get loaded, whichever initialises second sees "no
    ROCm-capable device" and device pointers could not be shared anyway.  So when torch
    is installed, map its copy first: our DT_NEEDED libamdhip64.so.7 then resolves to
    the already-loaded object by SONAME, and torch finds the same file later.
    AFX_HIP_RUNTIME=system skips this (pure C/ctypes deployments without torch)."""
    if os.environ.get("AFX_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def get_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `make -C audioflux_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback implementation")
        _share_torch_hip_runtime()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.afx_last_error.restype = ctypes.c_char_p
        _lib.afx_version.restype = ctypes.c_char_p
        _lib.afx_error_count.restype = ctypes.c_int
    return _lib


def last_error():
    return get_lib().afx_last_error().decode(errors="replace")


class _Checked:
    """A `void` reference entry point (bftObj_bft, stftObj_stft, cwtObj_cwt, ...) cannot return a
    status; the library records a failure inside one on the calling thread (afx_error_count(),
    include/afx_batch.h).  This proxy samples the counter around the call and raises, so an
    out-of-memory or launch failure never reaches the caller as a zero-filled "result"."""
    __slots__ = ("_fn",)

    def __init__(self, fn):
        object.__setattr__(self, "_fn", fn)

    def __setattr__(self, key, value):  # argtypes / restype go to the ctypes function
        setattr(self._fn, key, value)

    def __getattr__(self, key):
        return getattr(self._fn, key)

    def __call__(self, *args):
        lib = get_lib()
        before = lib.afx_error_count()
        res = self._fn(*args)
        if lib.afx_error_count() != before:
            raise RuntimeError(f"{self._fn.__name__} failed: {last_error()}")
        return res


def checked(fn):
    return fn if isinstance(fn, _Checked) else _Checked(fn)


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status}: {last_error()}")


def runtime_status():
    """0 when a gfx950 device is usable, negative otherwise."""
    return int(get_lib().afx_runtime_status())
