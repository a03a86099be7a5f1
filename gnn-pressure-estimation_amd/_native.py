"""ctypes binding of include/gatres.h (lib/libgatres_hip.so).

There is NO CPU fallback: if the library is missing it is built with hipcc, and if that fails or a kernel
launch fails the caller gets a RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _build

_LIB: Optional[C.CDLL] = None

DTYPE_F32, DTYPE_BF16 = 0, 1

ERRORS = {-1: "GATRES_E_BADARG (null/misaligned pointer or bad size)",
          -2: "GATRES_E_UNSUPPORTED (width / storage type not supported: nc must be a power of two in [4, 128]; bf16 "
              "needs nc >= 32)",
          -3: "GATRES_E_GRAPH (edge endpoint out of range)"}


class GatresGraph(C.Structure):
    """gatres_graph_t"""
    _fields_ = [("num_nodes", C.c_int32), ("num_edges_gat", C.c_int32), ("num_edges_mean", C.c_int32),
                ("num_segments", C.c_int32),
                ("rowptr", C.c_void_p), ("col", C.c_void_p), ("t_rowptr", C.c_void_p), ("t_eid", C.c_void_p),
                ("t_dst", C.c_void_p), ("m_rowptr", C.c_void_p), ("m_col", C.c_void_p), ("mt_rowptr", C.c_void_p),
                ("mt_dst", C.c_void_p), ("seg_ptr", C.c_void_p), ("max_segment_nodes", C.c_int32),
                ("max_segment_edges_gat", C.c_int32), ("max_segment_edges_mean", C.c_int32), ("flags", C.c_int32),
                ("window", C.c_int32 * 21), ("reserved2", C.c_int32), ("perm", C.c_void_p),
                ("halo", C.c_int32 * 7), ("reserved3", C.c_int32),
                ("part_tables", C.c_void_p), ("part_tables_m", C.c_int32), ("part_tables_stride", C.c_int32)]


MODEL_INFERENCE = 1            # GATRES_MODEL_INFERENCE (gatres_model_t.flags)


class GatresModel(C.Structure):
    """gatres_model_t"""
    _fields_ = [("num_blocks", C.c_int32), ("nc", C.c_int32), ("act_dtype", C.c_int32), ("flags", C.c_int32)]


_P = C.c_void_p
_I32, _I64, _U64, _F32, _F64 = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double
_GP, _MP = C.POINTER(GatresGraph), C.POINTER(GatresModel)

# name -> (restype, argtypes): must list every symbol include/gatres.h declares
SIGNATURES = {
    "gatres_graph_count_host": (C.c_int, [_P, _I64, _I64, C.POINTER(_I64)]),
    "gatres_graph_build_host": (C.c_int, [_P, _I64, _I64] + [_P] * 9),
    "gatres_graph_flags_host": (C.c_int, [_P, _I64, _I64, C.POINTER(_I32)]),
    "gatres_graph_part_tables_host": (C.c_int, [_P] * 10 + [_I32, _I32, _P, _I64, C.POINTER(_I64)]),
    "gatres_graph_segments_host": (C.c_int, [_P, _I64, _I64, _I32, _P] + [C.POINTER(_I32)] * 4),
    "gatres_graph_windows_host": (C.c_int, [_P, _I64, _I64, _P, _I32, _P]),
    "gatres_graph_reorder_host": (C.c_int, [_P, _I64, _I64, _P, _I32, _P]),
    "gatres_lz4_decompress_host": (_I64, [_P, _I64, _P, _I64]),
    "gatres_edge_index_hash": (C.c_int, [_P, _I64, _P, _P]),
    "gatres_permute_f32": (C.c_int, [_P, _P, _P, _I32, _I32, _P]),
    "gatres_gather_u8": (C.c_int, [_P, _P, _P, _I32, _P]),
    "gatres_fused_serialize": (C.c_int, [_P]),
    "gatres_probe_xcd_dispatch": (C.c_int, [_P]),
    "gatres_knobs_reload": (C.c_int, []),
    "gatres_fused_status_offset": (_I64, [_MP, _GP]),
    "gatres_lin0_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _P]),
    "gatres_proj_attn_fwd": (C.c_int, [_P] * 7 + [_I32] * 4 + [_P]),
    "gatres_gat_aggregate_fwd": (C.c_int, [_GP] + [_P] * 6 + [_I32] * 3 + [_P]),
    "gatres_mean_residual_relu_fwd": (C.c_int, [_GP, _P, _P, _P, _I32, _P]),
    "gatres_lin1_fwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _P]),
    "gatres_reduce_slabs": (C.c_int, [_P, _I32, _I64, _I64, _P, _P]),
    "gatres_lin1_bwd": (C.c_int, [_P] * 6 + [_I32, _I64, _I32, _I32, _I32, _P]),
    "gatres_mean_bwd": (C.c_int, [_GP, _P, _P, _I32, _P]),
    "gatres_gat_aggregate_bwd_dst": (C.c_int, [_GP] + [_P] * 7 + [_I32, _I32, _P]),
    "gatres_gat_aggregate_bwd_src": (C.c_int, [_GP] + [_P] * 8 + [_I32, _I32, _P]),
    "gatres_proj_bwd_dx": (C.c_int, [_P] * 5 + [_I32] * 3 + [_P]),
    "gatres_proj_bwd_dw": (C.c_int, [_P] * 3 + [_I32, _I64, _I32, _I32, _I32, _P]),
    "gatres_conv_param_grads": (C.c_int, [_P] * 7 + [_I32, _I64, _I32, _I32, _I32, _P]),
    "gatres_lin0_bwd": (C.c_int, [_P] * 5 + [_I32, _I64, _I32, _I32, _P]),
    "gatres_transpose_conv_weights": (C.c_int, [_P, _P, _I32, _I32, _P]),
    "gatres_mask_generate": (C.c_int, [_P, _I32, _F64, _U64, _P, _P, _P]),
    "gatres_stage_batch_mask": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I32, _F64, _U64, _P, _P, _P]),
    "gatres_stage_rows_mask": (C.c_int, [_P, _P, _I32, _P, _P, _P, _I32, _F64, _U64, _P, _P, _P]),
    "gatres_masked_mse": (C.c_int, [_P] * 5 + [_I32, _P]),
    "gatres_adam_step": (C.c_int, [_P] * 5 + [_I64] + [_F64] * 5 + [_F32, _P]),
    # typed per-op kernels (storage type argument): C ints
    "gatres_t_gat_aggregate_fwd": (C.c_int, [_GP] + [_P] * 6 + [C.c_int] * 4 + [_P]),
    "gatres_t_gat_aggregate_bwd_dst": (C.c_int, [_GP] + [_P] * 7 + [C.c_int] * 3 + [_P]),
    "gatres_t_gat_aggregate_bwd_src": (C.c_int, [_GP] + [_P] * 8 + [C.c_int] * 3 + [_P]),
    "gatres_t_mean_residual_relu_fwd": (C.c_int, [_GP, _P, _P, _P, C.c_int, C.c_int, _P]),
    "gatres_t_mean_bwd": (C.c_int, [_GP, _P, _P, C.c_int, C.c_int, _P]),
    "gatres_t_lin0_fwd": (C.c_int, [_P] * 5 + [C.c_int] * 3 + [_P]),
    "gatres_t_lin0_bwd": (C.c_int, [_P] * 5 + [C.c_int, _I64, C.c_int, C.c_int, C.c_int, _P]),
    "gatres_t_lin1_fwd": (C.c_int, [_P] * 4 + [C.c_int] * 3 + [_P]),
    "gatres_t_lin1_bwd": (C.c_int, [_P] * 6 + [C.c_int, _I64, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "gatres_t_conv_param_grads": (C.c_int, [_P] * 7 + [C.c_int, _I64, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "gatres_t_conv_partials": (C.c_int, [_P] * 3 + [C.c_int, _I64, C.c_int, C.c_int, C.c_int] + [_P] * 7 + [C.c_int] * 4 + [_P]),
    "gatres_convert_conv_weights_bf16": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "gatres_t_proj_attn_fwd": (C.c_int, [_P] * 7 + [C.c_int] * 5 + [_P]),
    "gatres_t_proj_bwd_dx": (C.c_int, [_P] * 5 + [C.c_int] * 4 + [_P]),
    "gatres_t_proj_bwd_dw": (C.c_int, [_P] * 3 + [C.c_int, _I64, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "gatres_blocked_supported": (C.c_int, [_MP, _GP]),
    "gatres_bf16_agg_proj_fwd": (C.c_int, [_GP] + [_P] * 12 + [_I32, _P]),
    "gatres_bf16_mean_proj_fwd": (C.c_int, [_GP] + [_P] * 9 + [_I32, _P]),
    "gatres_bf16_src_dx_bwd": (C.c_int, [_GP] + [_P] * 12 + [_I32, _I32, _P]),
    "gatres_param_count": (_I64, [_I32, _I32]),
    "gatres_saved_floats": (_I64, [_MP, _GP]),
    "gatres_scratch_floats": (_I64, [_MP, _GP]),
    "gatres_num_slabs": (_I32, [_MP, _I32]),
    "gatres_model_forward": (C.c_int, [_MP, _GP] + [_P] * 6 + [_P]),
    "gatres_model_backward": (C.c_int, [_MP, _GP] + [_P] * 8 + [_P]),
    "gatres_model_forward_per_op": (C.c_int, [_MP, _GP] + [_P] * 6 + [_P]),
    "gatres_model_backward_per_op": (C.c_int, [_MP, _GP] + [_P] * 8 + [_P]),
    "gatres_model_backward_per_op_part": (C.c_int, [_MP, _GP] + [_P] * 8 + [_I32, _I32, _I32, _P]),
    "gatres_model_reduce_grads": (C.c_int, [_MP, _GP, _P, _P, _P]),
    "gatres_fused_supported": (C.c_int, [_MP, _GP]),
    "gatres_fused_cus_per_segment": (C.c_int, [_MP, _GP]),
    "gatres_fused_window_kernel": (C.c_int, [_MP, _GP]),
    "gatres_fused_reset_sync": (C.c_int, [_MP, _GP, _P, _P]),
    "gatres_fused_set_stamps": (C.c_int, [_P, _I32]),
    "gatres_fused_prepare_backward": (C.c_int, [_MP, _GP, _P, _P, _P]),
    "gatres_fused_run": (C.c_int, [_MP, _GP] + [_P] * 10 + [_I32, _P]),
    "gatres_fused_param_grads": (C.c_int, [_MP, _GP, _P, _P, _P]),
    "gatres_fused_finish": (C.c_int, [_MP, _GP] + [_P] * 4 + [_I32] + [_P] * 4 + [_F64] * 5 + [_F32, _P]),
    "gatres_fused_finish_hp": (C.c_int, [_MP, _GP] + [_P] * 4 + [_I32] + [_P] * 4 + [_F64] * 5 + [_P, _F32, _P, _I32, _F64, _U64, _P, _P]),
    "gatres_fused_finish_folds": (C.c_int, [_MP, _GP]),
    "gatres_fused_param_grads_finish": (C.c_int, [_MP, _GP] + [_P] * 5 + [_I32] + [_P] * 4 + [_F64] * 5 +
                                        [_P, _F32, _I32, _I32, _P]),
    "gatres_train_step": (C.c_int, [_P, _P]),
    "gatres_version": (C.c_char_p, []),
}


ABI_VERSION = 6                # GATRES_ABI_VERSION of include/gatres.h


def diag_build() -> bool:
    """GATRES_DIAG_LIB=1 selects the diagnostic build of the library (stage stamps, wrong-result switches)."""
    return os.environ.get("GATRES_DIAG_LIB", "") not in ("", "0")


def lib_path() -> str:
    return _build.DIAG_LIB_PATH if diag_build() else _build.LIB_PATH


def _version_of(lib: C.CDLL):
    """(abi, build id) from gatres_version(): "gatres-gfx950 abi<N> build <id>"."""
    fn = lib.gatres_version
    fn.restype, fn.argtypes = C.c_char_p, []
    words = fn().decode().split()
    abi = next((int(w[3:]) for w in words if w.startswith("abi") and w[3:].isdigit()), -1)
    bid = words[words.index("build") + 1] if "build" in words and words.index("build") + 1 < len(words) else ""
    return abi, bid


def load(build_if_missing: bool = True) -> C.CDLL:
    """dlopen the in-tree library and attach the C signatures.  The library must have been built from the sources that lie
    beside it: a missing library, or one whose ABI / build id (``gatres_version()``) differs from ``_build.source_id()``, is
    (re)built with hipcc first -- or refused when ``build_if_missing`` is False.  Nothing else is ever loaded: there is no
    path override and no CPU fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    want = _build.source_id()

    def stale() -> str:
        if not os.path.exists(path):
            return f"{path} is missing"
        abi, bid = _version_of(C.CDLL(path))
        if abi != ABI_VERSION or bid != want:
            return f"{path} is abi{abi} build {bid or '?'}, the sources are abi{ABI_VERSION} build {want}"
        return ""

    why = stale()
    if why:
        if not build_if_missing:
            raise RuntimeError(f"{why}; run `python __graft_entry__.py` (build()) first")
        # One builder at a time: under torchrun every rank finds the stale library at the same moment, and N hipcc runs into the
        # same build/ and lib/ paths would mix objects.  The first rank to take the lock builds; the others wait and
        # find a current library when they get it.
        import fcntl
        os.makedirs(_build.LIB_DIR, exist_ok=True)
        with open(os.path.join(_build.LIB_DIR, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                was_missing = "missing" in why
                # Is the file on disk current NOW (another rank may have rebuilt it while this one waited)?  Not through
                # C.CDLL(path): a process that already mapped the stale file gets glibc's cached handle back and would rebuild
                # again.  The build stamp beside the objects is what build_native wrote for the file it linked.
                stamp = os.path.join(_build.OBJ_DIR + ("_diag" if diag_build() else ""), "BUILD_ID")
                try:
                    with open(stamp) as f:
                        on_disk = f.read().strip()
                except OSError:
                    on_disk = ""
                if not os.path.exists(path) or on_disk != want:
                    _build.build_native(diag=diag_build())
                # (a process that already mapped the stale file keeps it under the same name: dlopen a fresh copy of the new
                #  build -- copied while the lock is held, so no other rank relinks the file under the copy)
                lib = C.CDLL(path) if was_missing else _reopen(path)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
        abi, bid = _version_of(lib)
        if abi != ABI_VERSION or bid != want:
            raise RuntimeError(f"rebuilt {path} still reports abi{abi} build {bid}; expected abi{ABI_VERSION} build {want}")
    else:
        lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here == the library does not export the header's symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def load_unchecked(path: str) -> C.CDLL:
    """DIAGNOSTIC TOOLS ONLY (tests/micro/*.py): make an arbitrary build of the library the process's library -- an older
    build for an A/B measurement, a probe build with in-kernel stamps.  No ABI / build-id check; must be called before
    anything else loads the library.  The product never calls this and no environment variable leads here."""
    global _LIB
    if _LIB is not None:
        raise RuntimeError("the library is already loaded")
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _LIB = lib
    return lib


def _reopen(path: str) -> C.CDLL:
    """dlopen caches by path: after a rebuild in the same process, load the new file through a private copy."""
    import shutil
    import tempfile
    fd, tmp = tempfile.mkstemp(suffix=".so", prefix="libgatres_", dir=os.path.dirname(path))
    os.close(fd)
    shutil.copyfile(path, tmp)
    try:
        return C.CDLL(tmp)
    finally:
        os.unlink(tmp)


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: {ERRORS.get(rc, rc)}")
    raise RuntimeError(f"{what}: HIP error {rc}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def current_stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu_tensor(t: torch.Tensor, name: str, dtype=torch.float32) -> None:
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on a ROCm device (got {t.device}); this engine has no CPU path")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
