"""ctypes binding of the C-ABI in include/wespeaker_amd.h (no torch C++ extension machinery:
raw device pointers + the caller's hipStream_t cross the boundary).

The library is the product: if it is missing or fails to load, everything here raises -- there
is no CPU / eager fallback."""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# WS_LIB_PATH: another build of the same library (tools/ab_bench.sh compares two builds in one gpurun call); unset in
# every shipped path, and a path that does not load raises like the default one
LIB_PATH = os.environ.get("WS_LIB_PATH") or os.path.join(_HERE, "lib", "libwespeaker_amd.so")

_lib = None

ABI_VERSION = 105      # = WS_VERSION of include/wespeaker_amd.h

# name -> (restype, argtypes); also used by the ABI test to check every header symbol is exported
SIGNATURES = {
    "ws_version": (c_int, []),
    "ws_last_error": (c_char_p, []),
    "ws_num_frames": (c_int, [c_int, c_int]),
    "ws_wav_probe": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ws_wav_load_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "ws_frontend_create": (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    "ws_frontend_destroy": (None, [c_void_p]),
    "ws_fbank": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_float, c_int, c_int,
                         c_void_p, c_void_p]),
    "ws_engine_create": (c_int, [c_char_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    "ws_engine_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, c_int, POINTER(c_int64)]),
    "ws_engine_finalize": (c_int, [c_void_p, c_int, c_int]),
    "ws_engine_load": (c_int, [c_char_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    "ws_engine_reserve": (c_int, [c_void_p, c_int, c_int]),
    "ws_engine_max_batch": (c_int, [c_void_p]),
    "ws_engine_max_frames": (c_int, [c_void_p]),
    "ws_engine_destroy": (None, [c_void_p]),
    "ws_engine_embed_dim": (c_int, [c_void_p]),
    "ws_engine_feat_dim": (c_int, [c_void_p]),
    "ws_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ws_extract": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_float,
                           c_int, c_void_p, c_void_p]),
    "ws_fbank_ragged": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int64, c_float, c_int, c_int,
                                c_void_p, c_void_p]),
    "ws_forward_ragged": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ws_extract_ragged": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int64, c_float,
                                  c_int, c_void_p, c_void_p]),
    "ws_resample": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "ws_extract_chunked": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int,
                                   c_void_p, c_void_p]),
    "ws_cmn": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "ws_forward_ragged_cmvn": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ws_cmvn": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ws_frontend_set_cmvn": (c_int, [c_void_p, c_int, c_int]),
    "ws_num_windows": (c_int, [c_int, c_int, c_int]),
    "ws_extract_windows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                   c_int, c_void_p, c_int, c_void_p]),
    "ws_engine_set_precision": (c_int, [c_void_p, c_int]),
    "ws_engine_check_range": (c_int, [c_void_p, c_void_p]),
    "ws_debug_dispatch_log": (c_int, [c_int]),
    "ws_debug_fbank_mode": (c_int, [c_int]),
    "ws_debug_clock_probe": (c_int, [c_void_p, c_int, c_int64, c_void_p]),
    "ws_debug_row_gather": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "ws_debug_dispatch_report": (c_int64, [c_char_p, c_int64]),
    "ws_engine_profile_enable": (c_int, [c_void_p, c_int]),
    "ws_engine_profile_read": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ws_engine_flops": (c_double, [c_void_p, c_int, c_int]),
    "ws_plda_create": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                               POINTER(c_void_p)]),
    "ws_plda_destroy": (None, [c_void_p]),
    "ws_plda_prepare_enroll": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p]),
    "ws_plda_prepare_test": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "ws_plda_transform": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ws_plda_llr_matrix": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                   c_void_p, c_void_p]),
    "ws_plda_llr_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "ws_plda_stats_scratch": (c_int64, [c_int, c_int]),
    "ws_plda_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                              c_void_p, c_void_p, c_int64, c_void_p]),
    "ws_rows_affine": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                               c_void_p]),
    "ws_cos_table_rows": (c_int, [c_int]),
    "ws_cos_table_ld": (c_int, [c_int]),
    "ws_cos_prepare": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ws_cos_pairs": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p,
                             c_void_p]),
    "ws_cos_matrix": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "ws_cohort_stats": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int64,
                                c_void_p, c_void_p, c_void_p]),
    "ws_asnorm_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int64, c_void_p, c_void_p]),
}


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle of libwespeaker_amd.so."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "%s not found: build it with `python -m wespeaker_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        # the signatures above are those of include/wespeaker_amd.h at WS_VERSION: a stale build would take
        # shifted arguments silently
        if handle.ws_version() != ABI_VERSION:
            raise NativeError("%s reports C-ABI version %d, these bindings are for %d: rebuild it "
                              "(python -m wespeaker_amd.build)" % (LIB_PATH, handle.ws_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc < 0:
        msg = lib().ws_last_error()
        raise NativeError("%s failed (code %d): %s" % (what or "wespeaker_amd call", rc,
                                                       msg.decode() if msg else ""))
    return rc


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise NativeError("no MI355X / ROCm device visible: the wespeaker_amd hot path runs only on "
                          "the GPU (there is no CPU fallback)")


def current_stream_ptr(device=None):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device/host pointer of a torch tensor or numpy array as c_void_p."""
    if hasattr(t, "data_ptr"):
        return c_void_p(t.data_ptr())
    return c_void_p(t.ctypes.data)
