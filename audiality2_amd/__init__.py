"""audiality2_amd - MI355X-native voice rendering behind Audiality 2's unit API.

The product is the C-ABI shared library ``liba2amd.so`` (include/a2amd.h),
built from csrc/ by ``build.build_all()``.  This package is only the thin
Python veneer the tests and bench.py use to reach that C ABI; there is no
Python or CPU implementation of the render path behind it, and loading fails
loudly when the HIP library has not been built.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (A2AMD_LIB: another build of the same library, for A/B experiments)
LIB_PATH = os.environ.get("A2AMD_LIB") or os.path.join(_HERE, "liba2amd.so")

_lib = None


def load_library():
    """Return the ctypes handle of liba2amd.so (the HIP backend)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.a2amd_version.restype = ctypes.c_char_p
    return _lib


def open_backend(samplerate=48000, basepitch=None, channels=2, device=0, max_batch=64, stream=None):
    """Open a GPU render context; raises if no HIP device is present."""
    from .replay import Backend
    from .synth import basepitch_for
    if basepitch is None:
        basepitch = basepitch_for(samplerate)
    return Backend(load_library(), "a2amd_", samplerate, basepitch, channels, device, max_batch, stream)
