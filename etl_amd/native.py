"""ctypes binding of libetl_gfx950.so — the C ABI declared in include/etlg.h.

Loading fails loudly when the library has not been built (run
`python -m etl_amd.build` or `__graft_entry__.build()`); there is no Python or
CPU fallback for the decode path.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ETLG_LIB_PATH") or os.path.join(_HERE, "libetl_gfx950.so")

# every symbol include/etlg.h declares
EXPORTS = [
    "etlg_abi_version", "etlg_err_table", "etlg_type_class_of_oid", "etlg_array_elem_class", "etlg_slot_bytes",
    "etlg_ctx_create", "etlg_create_error", "etlg_ctx_destroy", "etlg_ctx_set_stream", "etlg_ctx_set_worker", "etlg_schema_put",
    "etlg_table_state", "etlg_table_ready", "etlg_ctx_reset_stream_state", "etlg_decode", "etlg_last_error",
    "etlg_batch_view_get", "etlg_batch_sync", "etlg_batch_download", "etlg_batch_header_to_device", "etlg_ctx_fence", "etlg_batch_free", "etlg_ctx_slots", "etlg_ctx_profile",
    "etlg_ctx_profile_read", "etlg_scan_boundaries", "etlg_copy_decode", "etlg_frame_tags",
    "etlg_table_forget", "etlg_table_cache_get", "etlg_host_alloc", "etlg_host_free", "etlg_control_stream", "etlg_shard_plan", "etlg_shard_replay",
    "etlg_batch_columns", "etlg_columns_view_get", "etlg_columns_free",
    "etlg_batch_rowbinary", "etlg_batch_protobuf", "etlg_rowbinary_view_get", "etlg_rowbinary_free", "etlg_batch_size_hints",
    "etlg_batch_finish_cells",
]

_LIB = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not built — run `python -m etl_amd.build` (needs hipcc); "
            "the decode path has no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64. If torch is
    # imported AFTER this library pulled in /opt/rocm's copy, the process ends up with
    # two runtimes and the second one finds no device. Importing torch first makes the
    # dynamic linker bind libetl_gfx950.so to the copy torch already loaded (same SONAME).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    # tests/simt builds the kernel sources against a CPU SIMT emulator to check kernel logic without a GPU; that
    # library is test infrastructure and must never serve a decode outside its own test run
    if hasattr(L, "etlg_simt_marker") and os.environ.get("ETLG_SIMT_RUN") != "1":
        raise NativeLibraryMissing(f"{LIB_PATH} is the SIMT test emulator build, not the gfx950 library; the decode path has no CPU fallback")
    L.etlg_abi_version.restype = C.c_uint32
    L.etlg_err_table.argtypes = [C.c_int32]
    L.etlg_err_table.restype = C.POINTER(abi.ErrDesc)
    L.etlg_type_class_of_oid.argtypes = [C.c_uint32]
    L.etlg_array_elem_class.argtypes = [C.c_uint32]
    L.etlg_slot_bytes.argtypes = [C.c_int32]
    L.etlg_slot_bytes.restype = C.c_uint32
    L.etlg_ctx_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
    L.etlg_create_error.restype = C.c_char_p
    L.etlg_ctx_destroy.argtypes = [C.c_void_p]
    L.etlg_ctx_destroy.restype = None
    L.etlg_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.etlg_ctx_set_worker.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64]
    L.etlg_schema_put.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_char_p, C.c_char_p, C.c_uint32,
                                  C.POINTER(abi.Col)]
    L.etlg_table_state.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_uint64]
    L.etlg_table_ready.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]
    L.etlg_ctx_reset_stream_state.argtypes = [C.c_void_p]
    L.etlg_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32,
                              C.POINTER(C.c_void_p)]
    L.etlg_last_error.argtypes = [C.c_void_p]
    L.etlg_last_error.restype = C.POINTER(abi.Error)
    L.etlg_batch_view_get.argtypes = [C.c_void_p, C.POINTER(abi.BatchView)]
    L.etlg_batch_sync.argtypes = [C.c_void_p, C.c_void_p]
    L.etlg_batch_download.argtypes = [C.c_void_p, C.c_void_p]
    L.etlg_batch_header_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.etlg_ctx_fence.argtypes = [C.c_void_p]
    L.etlg_ctx_fence.restype = C.c_int32
    L.etlg_batch_free.argtypes = [C.c_void_p]
    L.etlg_batch_free.restype = None
    L.etlg_ctx_slots.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(abi.SlotDesc))]
    L.etlg_ctx_profile.argtypes = [C.c_void_p, C.c_int32]
    L.etlg_ctx_profile_read.argtypes = [C.c_void_p, C.POINTER(abi.KernelStat), C.c_uint32, C.POINTER(C.c_uint32)]
    L.etlg_scan_boundaries.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.etlg_scan_boundaries.restype = C.c_int32
    L.etlg_copy_decode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]
    L.etlg_copy_decode.restype = C.c_int32
    L.etlg_frame_tags.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
    L.etlg_frame_tags.restype = C.c_int32
    L.etlg_table_forget.argtypes = [C.c_void_p, C.c_uint32]
    L.etlg_batch_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.etlg_columns_view_get.argtypes = [C.c_void_p, C.c_void_p]
    L.etlg_columns_free.argtypes = [C.c_void_p]
    L.etlg_columns_free.restype = None
    L.etlg_batch_rowbinary.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.etlg_batch_protobuf.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.etlg_rowbinary_view_get.argtypes = [C.c_void_p, C.c_void_p]
    L.etlg_rowbinary_free.argtypes = [C.c_void_p]
    L.etlg_rowbinary_free.restype = None
    L.etlg_batch_size_hints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.etlg_batch_finish_cells.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.etlg_table_cache_get.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    L.etlg_shard_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p]
    L.etlg_shard_plan.restype = C.c_int32
    L.etlg_shard_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.etlg_shard_replay.restype = C.c_int32
    L.etlg_control_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    L.etlg_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.etlg_host_free.argtypes = [C.c_void_p]
    L.etlg_host_free.restype = None
    if L.etlg_abi_version() != abi.ABI_VERSION:
        raise RuntimeError("libetl_gfx950.so ABI version mismatch")
    _LIB = L
    return L
