"""ctypes binding of libvamp_hip.so (include/vamp_hip.h).

There is no fallback: if the shared library has not been built (``__graft_entry__.build()`` or
``python -m vamp_amd.build``) importing a symbol raises, and every compute entry point needs a
GPU.  The signatures below are the single place where the C ABI is spelled out in Python.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import warnings

_HERE = os.path.dirname(os.path.abspath(__file__))
# VAMP_HIP_LIB: developer knob to A/B alternative builds of the same ABI (tools/tier_cost.py)
LIB_PATH = os.environ.get("VAMP_HIP_LIB") or os.path.join(_HERE, "libvamp_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_void_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/vamp_hip.h one to one
SIGNATURES = {
    "vamp_version": (C.c_int, []),
    "vamp_last_error": (C.c_char_p, []),
    "vamp_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vamp_ctx_create": (C.c_int, [c_void_pp, C.c_int, C.c_int, C.c_int]),
    "vamp_ctx_destroy": (C.c_int, [C.c_void_p]),
    "vamp_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vamp_ctx_set_stream_default": (C.c_int, [C.c_void_p]),
    "vamp_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "vamp_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "vamp_comm_library": (C.c_int, [C.c_char_p, C.c_int64]),
    "vamp_ctx_set_packing": (C.c_int, [C.c_void_p, C.c_int]),
    "vamp_set_regions": (C.c_int, [C.c_void_p, C.c_int, c_int64_p, c_double_p, c_double_p, c_double_p, c_int32_p,
                                   C.c_int, C.c_int, C.c_int, c_double_p, c_double_p]),
    "vamp_set_region_ids": (C.c_int, [C.c_void_p, c_int32_p]),
    "vamp_region_class": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vamp_region_ndim": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "vamp_lnprob": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, c_double_p, c_double_p, c_double_p]),
    "vamp_lnprob_all": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p, c_double_p]),
    "vamp_map_all": (C.c_int, [C.c_void_p, c_double_p, C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_double, c_double_p,
                     c_double_p, c_double_p, c_int64_p]),
    "vamp_model": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]),
    "vamp_model_all": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "vamp_line_records": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]),
    "vamp_wofz_re": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p, c_double_p]),
    "vamp_sampler_init": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, C.c_uint64, C.c_double, C.c_int32]),
    "vamp_sampler_set_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_int64_p, c_int64_p]),
    "vamp_sampler_set_shard_parts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_int64_p, c_int64_p]),
    "vamp_comm_unique_id": (C.c_int, [C.c_char_p]),
    "vamp_comm_init_rank": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "vamp_comm_destroy": (C.c_int, [C.c_void_p]),
    "vamp_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vamp_sampler_pack_get": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "vamp_sampler_scatter_put": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "vamp_sampler_run_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, c_double_p]),
    "vamp_sampler_bind_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vamp_sampler_state_ptrs": (C.c_int, [C.c_void_p, c_void_pp, c_void_pp, c_int64_p, c_int64_p]),
    "vamp_sampler_half_step": (C.c_int, [C.c_void_p, C.c_int]),
    "vamp_sampler_half_step_part": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "vamp_sampler_half_step_ext": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, c_int32_p, c_int32_p, c_double_p, c_double_p]),
    "vamp_sampler_run": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, c_double_p, c_double_p, c_int64_p, c_double_p]),
    "vamp_sampler_get_state": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_int64_p, c_int64_p]),
    "vamp_sampler_set_state": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int64]),
    "vamp_kernel_timing": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_int64_p]),
    "vamp_exchange_timing": (C.c_int, [C.c_void_p, c_double_p, c_int64_p]),
}

_lib = None


def bind(path):
    """CDLL + prototypes for ANY implementation of the C ABI.  The product only ever binds
    libvamp_hip.so (``load()``); tests and bench.py's cpu_baseline leg bind the host
    implementation of the same header (oracle/libvamp_cpu.so) explicitly through this."""
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Load libvamp_hip.so and attach the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -c 'import "
            "__graft_entry__ as g; g.build()'`).  vamp_amd has no CPU fallback.")
    _warn_if_loaded_before_torch()
    _lib = bind(LIB_PATH)
    return _lib


loaded_before_torch = False


def _warn_if_loaded_before_torch():
    """INTEGRATION.md "Two ROCm runtimes in one process": a PyTorch wheel carries private copies of
    libamdhip64 / libhsa-runtime64 / librccl.  Loaded AFTER torch, libvamp_hip.so binds to the runtime
    torch already mapped and the process has one runtime; loaded BEFORE, it binds to /opt/rocm/lib and a
    later `import torch` brings a second one, whose torch.cuda then fails to initialise ("No HIP GPUs are
    available").  The library itself keeps working either way (its RCCL is the one beside the libamdhip64
    it is bound to), so this is a diagnosis, not an error."""
    global loaded_before_torch
    if "torch" in sys.modules or os.environ.get("VAMP_NO_IMPORT_ORDER_WARNING"):
        return
    import importlib.util
    try:
        has_torch = importlib.util.find_spec("torch") is not None
    except (ImportError, ValueError):
        has_torch = False
    if has_torch:
        loaded_before_torch = True
        warnings.warn("vamp_amd: libvamp_hip.so is being loaded before torch.  If this process imports torch later it will "
                      "hold two ROCm runtimes and torch.cuda will not initialise: `import torch` first (INTEGRATION.md, "
                      "\"Two ROCm runtimes in one process\"); VAMP_NO_IMPORT_ORDER_WARNING=1 silences this.", RuntimeWarning,
                      stacklevel=3)


class VampError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvamp_hip error {code}: {msg}")
        self.code = code


def check(rc, lib=None):
    if rc != 0:
        raise VampError(rc, (lib or load()).vamp_last_error().decode("utf-8", "replace"))
