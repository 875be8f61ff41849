"""ctypes binding of libpnpx.so (the C ABI declared in include/pnpx.h).

There is NO CPU or PyTorch fallback anywhere in this package: if the HIP library is missing or a call
fails, a RuntimeError is raised.
"""
import ctypes as C
import os
import threading

import torch  # noqa: F401  -- must come first: libpnpx.so has to bind to the HIP runtime PyTorch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNPX_LIB", os.path.join(_HERE, "libpnpx.so"))   # PNPX_LIB: A/B builds of the same ABI

_lib = None
_lock = threading.Lock()

c_float_p = C.POINTER(C.c_float)
c_void_p = C.c_void_p


class PnpxError(RuntimeError):
    pass


PNPX_ERR_RANGE = 6      # include/pnpx.h: half-split range guard tripped (pnpx_ctx_status)


# name -> (restype, argtypes); mirrors include/pnpx.h one to one
_P = c_void_p  # device pointers travel as integers
_SIGNATURES = {
    "pnpx_version": (C.c_char_p, []),
    "pnpx_last_error": (C.c_char_p, []),
    "pnpx_ctx_create": (C.c_int, [C.c_int, C.POINTER(c_void_p)]),
    "pnpx_ctx_destroy": (C.c_int, [c_void_p]),
    "pnpx_ctx_reserve": (C.c_int, [c_void_p, C.c_int, C.c_int, C.c_int]),
    "pnpx_ctx_set_option": (C.c_int, [c_void_p, C.c_char_p, C.c_int]),
    "pnpx_ctx_get_option": (C.c_int, [c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "pnpx_ctx_status": (C.c_int, [c_void_p]),
    "pnpx_ctx_bytes": (C.c_size_t, [c_void_p]),
    "pnpx_unet_num_params": (C.c_size_t, []),
    "pnpx_unet_load": (C.c_int, [c_void_p, c_void_p, C.c_size_t]),
    "pnpx_drunet_num_params": (C.c_size_t, [C.c_int]),
    "pnpx_drunet_load": (C.c_int, [c_void_p, c_void_p, C.c_size_t, C.c_int]),
    "pnpx_unet_denoise": (C.c_int, [c_void_p, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, c_void_p]),
    "pnpx_unet_denoise_backward": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, c_void_p]),
    "pnpx_unet_denoise_train": (C.c_int, [c_void_p, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong), c_void_p]),
    "pnpx_unet_denoise_backward_ticket": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int,
                                                    C.c_ulonglong, c_void_p]),
    "pnpx_policy_num_params": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "pnpx_policy_load": (C.c_int, [c_void_p, c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "pnpx_policy_forward": (C.c_int, [c_void_p, _P, _P, _P, C.c_int, C.c_int, C.c_int, c_void_p]),
    "pnpx_unet_profile": (C.c_int, [c_void_p, _P, _P, _P, C.c_int, C.c_int, C.c_int, c_void_p, C.c_int,
                                    c_float_p, C.POINTER(C.c_double), C.POINTER(C.c_char_p), C.POINTER(C.c_int)]),
    "pnpx_fft2": (C.c_int, [c_void_p, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]),
    "pnpx_cdp_forward": (C.c_int, [c_void_p, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]),
    "pnpx_cdp_backward": (C.c_int, [c_void_p, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]),
    "pnpx_spi_inverse": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, c_void_p]),
    "pnpx_psnr": (C.c_int, [c_void_p, _P, _P, _P, C.c_int, C.c_int, c_void_p]),
    "pnpx_rows_gather": (C.c_int, [c_void_p, C.c_int, C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(C.c_size_t), _P,
                                   C.c_int, c_void_p]),
    "pnpx_rows_scatter": (C.c_int, [c_void_p, C.c_int, C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(C.c_size_t), _P,
                                    C.c_int, c_void_p]),
    "pnpx_live_compact": (C.c_int, [c_void_p, _P, _P, C.c_int, _P, C.POINTER(C.c_int), c_void_p]),
    "pnpx_policy_ob_pack": (C.c_int, [c_void_p, C.c_int, C.POINTER(c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), _P,
                                      C.c_int, C.c_int, C.c_int, _P, c_void_p]),
    "pnpx_csmri_admm": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 + [c_void_p]),
    "pnpx_csmri_admm_train": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 +
                              [_P, C.POINTER(C.c_ulonglong), c_void_p]),
    "pnpx_csmri_admm_backward": (C.c_int, [c_void_p, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P] + [C.c_int] * 4 +
                                 [C.c_ulonglong, c_void_p]),
    "pnpx_csmri_hqs_train": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 +
                             [_P, C.POINTER(C.c_ulonglong), c_void_p]),
    "pnpx_csmri_hqs_backward": (C.c_int, [c_void_p, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P] + [C.c_int] * 4 +
                                [C.c_ulonglong, c_void_p]),
    "pnpx_csmri_hqs": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 + [c_void_p]),
    "pnpx_csmri_pg_train": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 +
                            [_P, C.POINTER(C.c_ulonglong), c_void_p]),
    "pnpx_csmri_pg_backward": (C.c_int, [c_void_p, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P] + [C.c_int] * 4 +
                               [C.c_ulonglong, c_void_p]),
    "pnpx_csmri_pg": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 + [c_void_p]),
    "pnpx_csmri_apg_train": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 +
                             [_P, C.POINTER(C.c_ulonglong), c_void_p]),
    "pnpx_csmri_apg_backward": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P] + [C.c_int] * 4 +
                                [C.c_ulonglong, c_void_p]),
    "pnpx_csmri_apg": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 + [c_void_p]),
    "pnpx_csmri_redadmm_train": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 +
                                 [_P, C.POINTER(C.c_ulonglong), c_void_p]),
    "pnpx_csmri_redadmm_backward": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P] +
                                    [C.c_int] * 4 + [C.c_ulonglong, c_void_p]),
    "pnpx_csmri_redadmm": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 + [c_void_p]),
    "pnpx_pr_iadmm_train": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P, _P] + [C.c_int] * 6 +
                            [_P, C.POINTER(C.c_ulonglong), c_void_p]),
    "pnpx_pr_iadmm_backward": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 +
                               [C.c_ulonglong, c_void_p]),
    "pnpx_pr_iadmm": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P, _P] + [C.c_int] * 6 + [c_void_p]),
    "pnpx_spi_admm_train": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 +
                            [_P, C.POINTER(C.c_ulonglong), c_void_p]),
    "pnpx_spi_admm_backward": (C.c_int, [c_void_p, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P] + [C.c_int] * 4 +
                               [C.c_ulonglong, c_void_p]),
    "pnpx_spi_admm": (C.c_int, [c_void_p, _P, _P, _P, _P, _P, _P] + [C.c_int] * 5 + [c_void_p]),
    "pnpx_radon_det_count": (C.c_int, [C.c_int]),
    "pnpx_radon_forward": (C.c_int, [c_void_p, _P, _P, C.c_int, C.c_int, C.c_int, c_void_p]),
    "pnpx_radon_backprojection": (C.c_int, [c_void_p, _P, _P, C.c_int, C.c_int, C.c_int, c_void_p]),
    "pnpx_ct_iadmm_train": (C.c_int, [c_void_p, _P, _P, _P, C.c_int, C.c_float, _P, _P, _P] + [C.c_int] * 4 +
                            [_P, C.POINTER(C.c_ulonglong), c_void_p]),
    "pnpx_ct_iadmm_backward": (C.c_int, [c_void_p, C.c_int, C.c_float, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P] +
                               [C.c_int] * 3 + [C.c_ulonglong, c_void_p]),
    "pnpx_ct_pg_train": (C.c_int, [c_void_p, _P, _P, _P, C.c_int, C.c_float, _P, _P] + [C.c_int] * 4 +
                         [_P, C.POINTER(C.c_ulonglong), c_void_p]),
    "pnpx_ct_pg_backward": (C.c_int, [c_void_p, C.c_int, C.c_float, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P] +
                            [C.c_int] * 3 + [C.c_ulonglong, c_void_p]),
    "pnpx_ct_iadmm": (C.c_int, [c_void_p, _P, _P, _P, C.c_int, C.c_float, _P, _P, _P] + [C.c_int] * 4 + [c_void_p]),
    "pnpx_ct_pg": (C.c_int, [c_void_p, _P, _P, _P, C.c_int, C.c_float, _P, _P] + [C.c_int] * 4 + [c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """The loaded library; raises loudly when it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise PnpxError(
                        f"{LIB_PATH} not found: the HIP library has not been built "
                        "(run `make -C tfpnp_amd/csrc` or __graft_entry__.build()). There is no CPU fallback.")
                l = C.CDLL(LIB_PATH)
                missing = []
                for name, (res, args) in _SIGNATURES.items():
                    try:
                        fn = getattr(l, name)  # AttributeError if the .so lacks a declared symbol
                    except AttributeError:
                        if "PNPX_LIB" in os.environ:   # A/B run against an older build of the ABI (tools/ only)
                            missing.append(name)
                            continue
                        raise
                    fn.restype = res
                    fn.argtypes = args
                if missing:
                    import warnings
                    warnings.warn(f"PNPX_LIB={LIB_PATH} lacks {len(missing)} symbol(s) of include/pnpx.h: "
                                  + ", ".join(missing) + " -- calls to them will fail with AttributeError")
                _lib = l
    return _lib


def check(status):
    if status != 0:
        msg = lib().pnpx_last_error().decode("utf-8", "replace")
        raise PnpxError(f"pnpx call failed (status {status}): {msg}")
