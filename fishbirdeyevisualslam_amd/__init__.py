"""fishbirdeyevisualslam_amd -- MI355X-native front-end + optimiser path behind a C ABI.

The product is libfishbird_hip.so (HIP kernels + C-ABI, include/fishbird.h).  This Python
package is only the loader and thin host wrappers used by tests/ and bench.py.  There is no
CPU fallback: if the library is missing, lib() raises.
"""
import ctypes as _C
import os as _os

_HERE = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.path.join(_HERE, "libfishbird_hip.so")
_LIB = None


class FishbirdError(RuntimeError):
    pass


def lib():
    """The C-ABI library.  Raises if it has not been built (python -m fishbirdeyevisualslam_amd.build)."""
    global _LIB
    if _LIB is None:
        if not _os.path.exists(LIB_PATH):
            raise FishbirdError(
                LIB_PATH + " is missing: build it with `python -m fishbirdeyevisualslam_amd.build` "
                "(there is no CPU fallback)")
        # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Two HIP runtimes in one
        # process cannot both own the GPU, so load torch's first: our DT_NEEDED libamdhip64.so.7
        # then binds to the runtime that is already mapped.  Without torch (plain C/C++ hosts)
        # the library uses /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch-less hosts
            pass
        _LIB = _C.CDLL(LIB_PATH)
        _LIB.fb_last_error.restype = _C.c_char_p
    return _LIB


def check(rc, what=""):
    if rc != 0:
        raise FishbirdError("%s failed (%d): %s" % (what, rc, lib().fb_last_error().decode()))
