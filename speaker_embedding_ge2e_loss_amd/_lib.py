"""ctypes binding of libge2e_hip.so (include/ge2e_hip.h).  Plumbing only.

The library is loaded lazily and there is NO fallback: if the shared object is
missing or a symbol is absent, every product entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH

ABI_VERSION = 2

VARIANT_SOFTMAX, VARIANT_CONTRAST = 0, 1
VARIANTS = {"softmax": VARIANT_SOFTMAX, "contrast": VARIANT_CONTRAST}
IMPL_AUTO, IMPL_GENERIC, IMPL_FUSED_F32, IMPL_FUSED_SPLIT, IMPL_TILED, IMPL_TEAM, IMPL_WAVE, IMPL_AUTO_NO_TEAM = 0, 1, 2, 3, 4, 5, 6, 7
IMPLS = {"auto": IMPL_AUTO, "generic": IMPL_GENERIC, "fused_f32": IMPL_FUSED_F32,
         "fused_split": IMPL_FUSED_SPLIT, "tiled": IMPL_TILED, "team": IMPL_TEAM, "wave": IMPL_WAVE,
         # AUTO without the eight-CU team kernel: for a GPU that other streams / processes keep busy (ge2e_hip.h)
         "auto_no_team": IMPL_AUTO_NO_TEAM}
IMPL_NAMES = {v: k for k, v in IMPLS.items()}

_fp = C.c_void_p  # device pointers travel as integers

# symbol -> (restype, argtypes); mirrors include/ge2e_hip.h one to one
PROTOTYPES = {
    "ge2e_abi_version": (C.c_int, []),
    "ge2e_strerror": (C.c_char_p, [C.c_int]),
    "ge2e_resolve_impl": (C.c_int, [C.c_int] * 6),
    "ge2e_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "ge2e_workspace_init": (C.c_int, [_fp, C.c_size_t, _fp]),
    "ge2e_loss_fwd_bwd": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float,
                                    C.c_float, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp,
                                    C.c_size_t, _fp]),
    "ge2e_raw_supported": (C.c_int, [C.c_int] * 3),
    "ge2e_loss_fwd_bwd_raw": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, C.c_float,
                                        C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]),
    "ge2e_cos_sim": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _fp,
                               _fp, C.c_size_t, _fp]),
    "ge2e_cos_sim_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ge2e_cos_sim_centroids": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _fp, _fp]),
    "ge2e_calc_loss": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _fp, _fp, _fp]),
    "ge2e_selftest_split_gemm": (C.c_int, [_fp] * 7),
    "ge2e_selftest_wave_ops": (C.c_int, [_fp] * 3),
    "ge2e_selftest_rows16": (C.c_int, [_fp] * 6),
    "ge2e_selftest_team_bytes": (C.c_size_t, [C.c_int]),
    "ge2e_selftest_team": (C.c_int, [_fp, C.c_size_t, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ge2e_centroids": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ge2e_utterance_centroids": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ge2e_centroids_bwd": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ge2e_cos_sim_bwd_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "ge2e_cos_sim_bwd": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _fp, _fp,
                                   _fp, C.c_size_t, _fp]),
    "ge2e_calc_loss_bwd": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _fp, _fp, _fp, _fp]),
    "ge2e_cos_sim_rows": (C.c_int, [_fp, _fp] + [C.c_int] * 6 + [C.c_float, C.c_float, _fp, _fp]),
    "ge2e_cos_sim_rows_bwd_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "ge2e_cos_sim_rows_bwd": (C.c_int, [_fp, _fp, _fp, _fp] + [C.c_int] * 6 + [C.c_float, C.c_float, _fp, _fp, _fp, C.c_size_t, _fp]),
    "ge2e_calc_loss_rows": (C.c_int, [_fp] + [C.c_int] * 5 + [C.c_float, C.c_int, _fp, _fp, _fp]),
    "ge2e_calc_loss_rows_bwd": (C.c_int, [_fp] + [C.c_int] * 5 + [C.c_float, C.c_int, _fp, _fp, _fp, _fp]),
    "ge2e_normalize_unperm": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp]),
    "ge2e_normalize_unperm_bwd": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp]),
    "ge2e_eer_counts": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp]),
    "ge2e_scale_grads": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    "ge2e_sample_batch": (C.c_int, [_fp, C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ge2e_selftest_team_fallback": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, C.c_float,
                                              C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
    "ge2e_selftest_team_abort_midgrid": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, C.c_float,
                                                   C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
    "ge2e_selftest_team_grid": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, C.c_float,
                                          C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp, C.c_int]),
}

_lib = None


class GE2ELibraryError(RuntimeError):
    pass


def load(path: str | None = None):
    """Load (once) and type the shared library.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("GE2E_HIP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise GE2ELibraryError(
            f"{path} not found: the HIP extension is not built. Run "
            "`python -m speaker_embedding_ge2e_loss_amd.build` (needs hipcc); there is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GE2ELibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ge2e_abi_version() != ABI_VERSION:
        raise GE2ELibraryError(f"ABI mismatch: library {lib.ge2e_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().ge2e_strerror(code).decode()
        raise RuntimeError(f"{what} failed: [{code}] {msg}")
