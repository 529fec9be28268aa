"""ctypes binding of libicematch.so (C ABI in include/icematch.h).

The library is the product: there is no Python/torch fallback. If it is missing or fails to load, every
entry point raises. torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("ICEMATCH_LIB") or os.path.join(CSRC, "libicematch.so")  # env override: A/B builds

_lib: Optional[C.CDLL] = None


class IcematchError(RuntimeError):
    """A call into libicematch.so returned non-zero (`Context.check`): out of device memory, a HIP error, a guard failure, bad arguments.
    Distinct from other RuntimeErrors (torch's, e.g. an `interpolate` size error out of a bad `resize` option) so that callers can tell a
    library / device failure - never to be swallowed - from a failure of an option (`matching/matchers.py` q7, `sequence.py`)."""

    def __init__(self, what: str, rc: int, message: str):
        super().__init__(f"{what} failed ({rc}): {message}")
        self.what, self.rc = what, rc


class LightGlueConf(C.Structure):
    _fields_ = [("depth_confidence", C.c_double), ("width_confidence", C.c_double),
                ("filter_threshold", C.c_double), ("n_layers", C.c_int), ("pruning_min_kpts", C.c_int)]


class SuperGlueConf(C.Structure):
    _fields_ = [("sinkhorn_iterations", C.c_int), ("match_threshold", C.c_double), ("n_layers", C.c_int)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float

# name -> argtypes; every symbol declared in include/icematch.h
SIGNATURES = {
    "im_version": [],
    "im_ctx_create": [_I, C.POINTER(_P)],
    "im_ctx_destroy": [_P],
    "im_last_error": [_P],
    "im_ctx_reserve": [_P, _I, _I, _I, _I],
    "im_profile_begin": [_P],
    "im_profile_end": [_P, C.c_char_p, C.c_size_t],
    "im_set_tensor": [_P, C.c_char_p, C.c_char_p, _P, C.c_size_t],
    "im_finalize_weights": [_P, C.c_char_p],
    "im_superpoint_forward": [_P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _I, _P, _P, _P, _P, _P],
    "im_superpoint_candidates": [_P, _I, _P, _P],
    "im_lightglue_forward": [_P, _P, _P, _P, _P, C.POINTER(LightGlueConf), _P, _P, _P, _P, _P],
    "im_lightglue_forward_pairs": [_P, _I, _P, _P, _P, _P, C.POINTER(LightGlueConf), _P, _P, _P, _P, _P],
    "im_pack_records": [_P, _I, _P, _P, _P, _P, _I, _P, _P, _P],
    "im_superglue_forward": [_P, _P, _P, _P, _P, _P, C.POINTER(SuperGlueConf), _P, _P, _P, _P],
    "im_pack_record": [_P, _P, _P, _P, _P, _I, _P, _P],
    "im_debug_read": [_P, C.c_char_p, _P, C.c_size_t, _P],
    "im_debug_clock_probe": [_P, _I, _P, _P],
    "im_debug_guard_failures": [],
    "im_debug_guard_selftest": [_P, _P],
    "im_debug_guards_check": [_P, _P],
    "im_pyr_down": [_P, _P, _P, _I, _I, _I, _I, _P],
    "im_pyr_up": [_P, _P, _P, _I, _I, _I, _I, _P],
    "im_gemm_nt": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P],
    "im_ffn_fused": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P],
    "im_conv3x3": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "im_conv3x3_winograd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "im_flash_attn": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "im_nms": [_P, _P, _P, _I, _I, _I, _I, _P],
    "im_select_topk": [_P, _P, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P],
    "im_sample_descriptors": [_P, _P, _I, _I, _I, _P, _P, _P, _P],
    "im_assign_from_sim": [_P, _P, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P],
    "im_log_optimal_transport": [_P, _P, _I, _I, _I, _F, _I, _P, _P],
    "im_merge_tile_matches": [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "im_gather_rows": [_P, _P, _I, _P, _I, _P, _P],
    "im_ransac_fundamental": [_P, _P, _P, _I, _I, C.c_double, C.c_uint, _P, _P, _P, _P],
    "im_ransac_essential": [_P, _P, _P, _I, _I, C.c_double, C.c_uint, _P, _P, _P, _P],
    "im_triangulate_linear": [_P, _P, _P, _P, _P, _I, _P, _P],
}


def build(force: bool = False) -> str:
    """Compile libicematch.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building libicematch.so failed:\n{r.stdout[-4000:]}\n{r.stderr[-4000:]}")
    return LIB_PATH


def load() -> C.CDLL:
    """Load the library (no compute is triggered). Raises if it is absent: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64: it must be the HIP runtime of the process (device pointers and streams
    # are shared with torch), so torch is imported before the library's NEEDED libamdhip64.so.7 is resolved.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(there is no CPU/torch fallback for the matching hot path)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name == "im_last_error" else (None if name == "im_ctx_destroy" else C.c_int)
    _lib = lib
    return lib


def ptr(t) -> int:
    """Device (or host) pointer of a torch tensor / numpy array; None -> NULL."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def stream_ptr(device=None) -> Optional[int]:
    """The current HIP stream of `device` (a torch device or index; None = the process's current device). Callers that hold an
    engine pass its device: the stream must belong to the device the context's buffers live on, whatever device is current."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


class Context:
    """RAII wrapper of `im_ctx`."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = _P()
        rc = self.lib.im_ctx_create(device, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"im_ctx_create(device={device}) failed with {rc} (is a HIP device visible?)")
        self.h = h
        self.device = device

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.im_last_error(self.h)
            raise IcematchError(what, rc, msg.decode() if msg else "")

    def call(self, name: str, *args):
        self.check(getattr(self.lib, name)(self.h, *args), name)

    def close(self):
        if getattr(self, "h", None):
            self.lib.im_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
