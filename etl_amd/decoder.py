"""Host-side mirror of the reference interface for the decode hot path.

`Decoder` is what the reference's `ApplyLoop` is to its codec: it owns the
per-stream transaction state and the shared table cache (inside the native
context), is primed with stored schemas / table states the way
`SchemaStore` / `StateStore` are (crates/etl/src/store/schema/base.rs:19-69,
crates/etl/src/store/state/base.rs:25-139), and turns batches of raw
logical-replication messages into the events `Destination::write_events`
would receive (crates/etl/src/destination/base.rs:207). All per-row work runs
in the gfx950 kernels of libetl_gfx950.so; this module only moves pointers.
"""
import ctypes as C

import numpy as np

from . import abi, native
from .view import HostBatch


class EtlError(Exception):
    """EtlError {kind, description, detail} (crates/etl/src/error.rs:27-76)."""

    def __init__(self, kind, code, description, detail, frame_index):
        self.kind, self.code, self.description, self.detail, self.frame_index = kind, code, description, detail, frame_index
        super().__init__(f"{abi.KIND_NAMES.get(kind, kind)}: {description} (frame {frame_index})")


def _ptr(a):
    return a.ctypes.data if a is not None else None


class Batch:
    """A decoded batch (etlg_batch). `host()` copies the arena into numpy arrays."""

    def __init__(self, dec, handle, rc, keep=None):
        self.dec, self.h, self.rc = dec, handle, rc
        self._keep = keep
        self.error = dec.last_error() if rc != abi.OK else None

    def view(self):
        v = abi.BatchView()
        self.dec.L.etlg_batch_view_get(self.h, C.byref(v))
        return v

    def sync(self):
        self.rc = self.dec.L.etlg_batch_sync(self.dec.h, self.h)
        self.error = self.dec.last_error() if self.rc != abi.OK else None
        return self.rc

    def header_to_device(self, dst_ptr):
        """Enqueue the 64-byte result header into device memory (multi-GPU all-gather input)."""
        return self.dec.L.etlg_batch_header_to_device(self.dec.h, self.h, C.c_void_p(dst_ptr))

    def host(self):
        v = self.view()
        if v.on_device:
            rc = self.dec.L.etlg_batch_download(self.dec.h, self.h)
            if rc != abi.OK:
                raise self.dec.last_error()
            v = self.view()
        return HostBatch.from_view(v)

    def columns(self, slot, kinds=("I",), on_device=False, parse_arrays=False, format_json=False):
        """Arrow-layout column buffers of the rows decoded against schema slot `slot`, built on the device
        (etlg_batch_columns). The batch must still be device-resident. parse_arrays: bool / int2 / int4 / int8 / oid array
        columns are parsed on the device into list columns (abi.AK_LIST); a malformed literal raises the reference's error.
        format_json: json / jsonb columns leave as LargeUtf8 of serde_json's `Value::to_string()` (abi.ROWS_FORMAT_JSON)."""
        rk = ((abi.ROWS_INSERT if "I" in kinds else 0) | (abi.ROWS_UPDATE if "U" in kinds else 0) | (abi.ROWS_PARSE_ARRAYS if parse_arrays else 0)
              | (abi.ROWS_FORMAT_JSON if format_json else 0))
        out = C.c_void_p()
        rc = self.dec.L.etlg_batch_columns(self.dec.h, self.h, slot, rk, abi.F_OUTPUT_ON_DEVICE if on_device else 0, C.byref(out))
        if rc != abi.OK or not out:
            raise self.dec.last_error()
        return Columns(self.dec, out)

    def finish_cells(self, what=abi.FINISH_ARRAYS | abi.FINISH_FLOATS):
        """Typed arrays / exact floats in the arena, on the device (etlg_batch_finish_cells). Returns abi.FinishStats."""
        st = abi.FinishStats()
        rc = self.dec.L.etlg_batch_finish_cells(self.dec.h, self.h, what, C.byref(st))
        if rc != abi.OK:
            raise self.dec.last_error()
        return st

    def size_hints(self, model):
        """Event::size_hint per event (np.uint64; bit 63 = abi.SIZE_HINT_INCOMPLETE), computed on the device
        (etlg_batch_size_hints). `model`: abi.SizeModel — the reference's size_of constants."""
        out = np.zeros(int(self.view().n_events), dtype=np.uint64)
        rc = self.dec.L.etlg_batch_size_hints(self.dec.h, self.h, C.byref(model), 0, out.ctypes.data)
        if rc != abi.OK:
            raise self.dec.last_error()
        return out

    def rowbinary(self, slot, nullable_flags, engine=abi.CH_MERGE_TREE, on_device=False):
        """ClickHouse RowBinary rows of schema slot `slot`, encoded on the device (etlg_batch_rowbinary). Raises EtlError for
        the reference's ConversionErrors; `RowBinary.status == abi.RB_NEEDS_HOST` when a cell has no device encoding."""
        nf = np.ascontiguousarray(nullable_flags, dtype=np.uint8)
        out = C.c_void_p()
        rc = self.dec.L.etlg_batch_rowbinary(self.dec.h, self.h, slot, nf.ctypes.data, len(nf), engine,
                                             abi.F_OUTPUT_ON_DEVICE if on_device else 0, C.byref(out))
        if rc != abi.OK or not out:
            raise self.dec.last_error()
        return RowBinary(self.dec, out)

    def protobuf(self, slot, on_device=False):
        """BigQuery protobuf rows (one per Insert event) of schema slot `slot`, encoded on the device (etlg_batch_protobuf)."""
        out = C.c_void_p()
        rc = self.dec.L.etlg_batch_protobuf(self.dec.h, self.h, slot, abi.F_OUTPUT_ON_DEVICE if on_device else 0, C.byref(out))
        if rc != abi.OK or not out:
            raise self.dec.last_error()
        return RowBinary(self.dec, out)

    def close(self):
        if self.h:
            self.dec.L.etlg_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Columns:
    """etlg_columns: `view` = etlg_columns_view; `host_arrays()` wraps host-resident buffers as numpy arrays."""

    def __init__(self, dec, handle):
        self.dec, self.h = dec, handle
        self.view = abi.ColumnsView()
        dec.L.etlg_columns_view_get(handle, C.byref(self.view))

    @property
    def n_rows(self):
        return int(self.view.n_rows)

    def column(self, i):
        return self.view.cols[i]

    def _np(self, ptr, nbytes, dtype):
        if not nbytes:
            return np.zeros(0, dtype=dtype)
        assert not self.view.on_device, "device-resident buffers: read them through abi.device_tensor"
        return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=dtype)

    def host_arrays(self, i):
        """(validity bits u8[], deferred bits u8[], values u8[], offsets i64[] | None) of column i (views into the block)."""
        k, n = self.view.cols[i], self.n_rows
        if k.arrow_kind == abi.AK_NONE:
            return None
        bm = (n + 63) // 64 * 8
        offs = self._np(k.offsets, (n + 1) * 8, np.int64) if k.offsets else None
        return self._np(k.validity, bm, np.uint8), self._np(k.deferred, bm, np.uint8), self._np(k.values, int(k.values_bytes), np.uint8), offs

    def child_validity(self, i):
        k = self.view.cols[i]
        return self._np(k.child_validity, (int(k.child_count) + 63) // 64 * 8, np.uint8)

    def child_offsets(self, i):
        k = self.view.cols[i]
        return self._np(k.child_offsets, (int(k.child_count) + 1) * 8, np.int64) if k.child_offsets else None

    def row_event(self):
        return self._np(self.view.row_event, self.n_rows * 8, np.uint64)

    def close(self):
        if self.h:
            self.dec.L.etlg_columns_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RowBinary:
    """etlg_rowbinary: `view` = etlg_rowbinary_view; the accessors below wrap host-resident buffers."""

    def __init__(self, dec, handle):
        self.dec, self.h = dec, handle
        self.view = abi.RowBinaryView()
        dec.L.etlg_rowbinary_view_get(handle, C.byref(self.view))

    status = property(lambda self: int(self.view.status))
    n_rows = property(lambda self: int(self.view.n_rows))

    def _np(self, ptr, nbytes, dtype):
        if not nbytes or not ptr:
            return np.zeros(0, dtype=dtype)
        assert not self.view.on_device
        return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=dtype)

    def bytes(self):
        return self._np(self.view.bytes, int(self.view.n_bytes), np.uint8)

    def row_offsets(self):
        return self._np(self.view.row_offsets, (self.n_rows + 1) * 8, np.int64)

    def row_event(self):
        return self._np(self.view.row_event, self.n_rows * 8, np.uint64)

    def close(self):
        if self.h:
            self.dec.L.etlg_rowbinary_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Decoder:
    def __init__(self, device=0, stream=None):
        self.L = native.lib()
        h = C.c_void_p()
        rc = self.L.etlg_ctx_create(device, C.byref(h))
        if rc != abi.OK:
            why = (self.L.etlg_create_error() or b"").decode()
            raise RuntimeError(f"etlg_ctx_create failed ({abi.KIND_NAMES.get(rc, rc)}: {why}): an MI355X (gfx950) device is required")
        self.h = h
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "h", None):
            self.L.etlg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- control plane (same calls the oracle wrapper exposes)
    def set_stream(self, cuda_stream_ptr):
        return self.L.etlg_ctx_set_stream(self.h, C.c_void_p(cuda_stream_ptr))

    def set_worker(self, worker=abi.WORKER_APPLY, table_id=0, bootstrap_lsn=0):
        return self.L.etlg_ctx_set_worker(self.h, worker, table_id, bootstrap_lsn)

    def debug_overlapped(self):
        """ASYNC batches enqueued beside their predecessor on the second decode stream (debugging aid, not in etlg.h)."""
        self.L.etlg_ctx_debug_overlapped.restype = C.c_ulonglong
        self.L.etlg_ctx_debug_overlapped.argtypes = [C.c_void_p]
        return int(self.L.etlg_ctx_debug_overlapped(self.h))

    def debug_ctl_ahead(self):
        """Batches whose control pre-pass ran ahead of their decode (ASYNC without NO_CONTROL on a stream with Relation / DDL frames)."""
        self.L.etlg_ctx_debug_ctl_ahead.restype = C.c_ulonglong
        self.L.etlg_ctx_debug_ctl_ahead.argtypes = [C.c_void_p]
        return int(self.L.etlg_ctx_debug_ctl_ahead(self.h))

    def fence(self):
        """The context's stream waits (device side) for every ASYNC batch enqueued so far — decode kernels on the library's two
        decode streams and header copies (etlg_ctx_fence); call before enqueuing a consumer of their arenas / headers on it."""
        return self.L.etlg_ctx_fence(self.h)

    def reset_stream_state(self):
        return self.L.etlg_ctx_reset_stream_state(self.h)

    def schema_put(self, table_id, snapshot_lsn, cols, schema="public", name="t"):
        arr = abi.make_cols(cols)
        return self.L.etlg_schema_put(self.h, table_id, snapshot_lsn, schema.encode(), name.encode(), len(cols), arr)

    def table_state(self, table_id, kind, lsn=0):
        return self.L.etlg_table_state(self.h, table_id, kind, lsn)

    def table_ready(self, table_id, snapshot_lsn, repl_mask, ident_mask):
        r = np.asarray(repl_mask, dtype=np.uint8)
        i = np.asarray(ident_mask, dtype=np.uint8)
        return self.L.etlg_table_ready(self.h, table_id, snapshot_lsn, _ptr(r), _ptr(i), len(r))

    def table_forget(self, table_id):
        """SharedTableCache::remove_table (table_cache.rs:131-145)."""
        return self.L.etlg_table_forget(self.h, table_id)

    def cache_state(self, table_id):
        """SharedTableCache::get (table_cache.rs:99-102): None, or (kind 1 WaitingForRelation | 2 Ready, snapshot id, schema slot)."""
        k, sn, sl = C.c_int32(), C.c_uint64(), C.c_int32()
        if not self.L.etlg_table_cache_get(self.h, table_id, C.byref(k), C.byref(sn), C.byref(sl)):
            return None
        return k.value, sn.value, sl.value

    def last_error(self):
        e = self.L.etlg_last_error(self.h).contents
        return EtlError(e.kind, e.code, (e.description or b"").decode(), (e.detail or b"").decode() or None, e.frame_index)

    # ---- decode
    def decode(self, buf, offsets=None, flags=0):
        """Host buffers: buf = bytes / np.uint8, offsets = optional u32 sidecar."""
        a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint32)
        nfr = 0 if off is None else len(off) - 1
        out = C.c_void_p()
        rc = self.L.etlg_decode(self.h, _ptr(a), a.size, _ptr(off), nfr, flags & ~abi.F_INPUT_ON_DEVICE, C.byref(out))
        if not out:
            raise self.last_error()
        return Batch(self, out, rc, keep=(a, off))

    def host_alloc(self, nbytes):
        """Pinned host memory (etlg_host_alloc) as a writable np.uint8 array; free it with host_free(array)."""
        p = C.c_void_p()
        if self.L.etlg_host_alloc(self.h, nbytes, C.byref(p)) != abi.OK:
            raise self.last_error()
        a = np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=np.uint8)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, a):
        self.L.etlg_host_free(C.c_void_p(self._pinned.pop(a.ctypes.data)))

    def decode_host_ptr(self, buf_ptr, nbytes, offs_ptr, nframes, flags):
        """Host buffers by address (pinned staging buffers of a batcher): with abi.F_ASYNC | abi.F_OUTPUT_ON_DEVICE the upload
        travels on the library's copy stream beside the previous batch's decode; the buffers stay untouched until the batch is synced."""
        out = C.c_void_p()
        rc = self.L.etlg_decode(self.h, C.c_void_p(buf_ptr), nbytes, C.c_void_p(offs_ptr), nframes, flags & ~abi.F_INPUT_ON_DEVICE, C.byref(out))
        if not out:
            raise self.last_error()
        return Batch(self, out, rc)

    def debug_staged(self):
        self.L.etlg_ctx_debug_staged.restype = C.c_ulonglong
        self.L.etlg_ctx_debug_staged.argtypes = [C.c_void_p]
        return int(self.L.etlg_ctx_debug_staged(self.h))

    def decode_device(self, buf_ptr, nbytes, offs_ptr, nframes, flags=abi.F_OUTPUT_ON_DEVICE):
        """Device-resident input (raw device pointers, e.g. torch tensor .data_ptr())."""
        out = C.c_void_p()
        rc = self.L.etlg_decode(self.h, C.c_void_p(buf_ptr), nbytes, C.c_void_p(offs_ptr), nframes,
                                flags | abi.F_INPUT_ON_DEVICE, C.byref(out))
        if not out:
            raise self.last_error()
        return Batch(self, out, rc)

    def copy_decode(self, slot, buf, row_offsets, flags=0):
        """Table-copy rows (COPY text format) from host buffers against schema slot `slot`
        (the value table_ready returned)."""
        a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf)
        off = np.ascontiguousarray(row_offsets, dtype=np.uint32)
        out = C.c_void_p()
        rc = self.L.etlg_copy_decode(self.h, slot, _ptr(a), a.size, _ptr(off), len(off) - 1, flags & ~abi.F_INPUT_ON_DEVICE, C.byref(out))
        if not out:
            raise self.last_error()
        return Batch(self, out, rc, keep=(a, off))

    def copy_decode_device(self, slot, buf_ptr, nbytes, offs_ptr, nrows, flags=abi.F_OUTPUT_ON_DEVICE):
        out = C.c_void_p()
        rc = self.L.etlg_copy_decode(self.h, slot, C.c_void_p(buf_ptr), nbytes, C.c_void_p(offs_ptr), nrows,
                                     flags | abi.F_INPUT_ON_DEVICE, C.byref(out))
        if not out:
            raise self.last_error()
        return Batch(self, out, rc)

    # ---- measurement
    def profile(self, enable=True):
        """HIP-event timing of every kernel launch (etlg_ctx_profile). enable=2: additionally keep ASYNC batches on ONE decode stream,
        so that the durations are those of a kernel running alone (with two streams consecutive kernels overlap)."""
        return self.L.etlg_ctx_profile(self.h, 2 if enable == 2 else 1 if enable else 0)

    def profile_read(self):
        arr = (abi.KernelStat * 24)()
        n = C.c_uint32()
        self.L.etlg_ctx_profile_read(self.h, arr, 24, C.byref(n))
        return {arr[i].name.decode(): (arr[i].launches, arr[i].total_ms) for i in range(n.value)}

    def debug_paths(self):
        """Batches finished per kernel path (debugging aid, not in etlg.h): 'fused' / 'cells' / 'plan' = produced by that
        single-pass kernel, 'multipass' = decoded by the multi-pass kernels directly, 'redone' = a single-pass result thrown
        away and redone by the multi-pass kernels (errors), 'plan_redone' = a fixed-width-plan result redone by the generic
        kernel, 'control' = batches that took the control path (a Relation / DDL frame), 'chain_rerun' = ASYNC batches run
        again because their predecessor failed."""
        out = (C.c_ulonglong * 8)()
        self.L.etlg_ctx_debug_paths8(self.h, out)
        return dict(zip(("fused", "cells", "multipass", "redone", "plan", "plan_redone", "control", "chain_rerun"), [int(x) for x in out]))

    def debug_rows(self):
        """Batches by the row-synchronous kernel (debugging aid, not in etlg.h): 'rows' = produced by k_rows, 'rows_redone' = handed back by it
        (a tile that did not fit its LDS window / image) and decoded again by k_cells / k_fused."""
        out = (C.c_ulonglong * 3)()
        self.L.etlg_ctx_debug_rows(self.h, out)
        return {"rows": int(out[0]), "rows_redone": int(out[1]), "rows_resized": int(out[2])}

    def debug_copy(self):
        """Table-copy batches by path (debugging aid, not in etlg.h): 'direct' = produced by the rows -> arena kernel (k_copy_cells),
        'frames' = decoded through the row -> frame rewrite (a malformed row, rows wider than a tile's window, ETLG_COPY_DIRECT=0)."""
        out = (C.c_ulonglong * 2)()
        self.L.etlg_ctx_debug_copy(self.h, out)
        return {"direct": int(out[0]), "frames": int(out[1])}

    def debug_ring_recleared(self):
        """Result blocks cleared again after a second attempt behind their lap's re-initialisation (debugging aid, not in etlg.h)."""
        out = (C.c_ulonglong * 8)()
        self.L.etlg_ctx_debug_ring(self.h, out)
        return int(out[0])

    def debug_chains_healed(self):
        """ASYNC chains that were finished early because their last batch was marked for a second attempt (debugging aid)."""
        out = (C.c_ulonglong * 8)()
        self.L.etlg_ctx_debug_ring(self.h, out)
        return int(out[1])

    def debug_chains_spared(self):
        """Second attempts of a fixed-width-plan batch that left the carried transaction state its first attempt had published: the batches
        in flight behind it were NOT decoded again (debugging aid)."""
        out = (C.c_ulonglong * 8)()
        self.L.etlg_ctx_debug_ring(self.h, out)
        return int(out[5])

    def debug_scan_chained(self):
        """(batches whose decode was enqueued behind their boundary scan with the frame count read on the device, those of them that
        had to be decoded again with the count in hand) — debugging aid."""
        out = (C.c_ulonglong * 8)()
        self.L.etlg_ctx_debug_ring(self.h, out)
        return int(out[3]), int(out[4])

    def debug_chain_reissued(self):
        """ASYNC batches enqueued again — chained to the new result — behind a batch that was decoded again (debugging aid)."""
        out = (C.c_ulonglong * 8)()
        self.L.etlg_ctx_debug_ring(self.h, out)
        return int(out[2])

    def frame_tags(self, buf, offsets):
        """pgoutput tag of every frame (np.uint8; 0 = malformed), classified on the device (etlg_frame_tags)."""
        import numpy as np
        a = np.ascontiguousarray(buf, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint32)
        out = np.zeros(len(o) - 1, dtype=np.uint8)
        rc = self.L.etlg_frame_tags(self.h, a.ctypes.data, len(a), o.ctypes.data, len(o) - 1, 0, out.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"etlg_frame_tags failed: {rc}")
        return out

    def frame_tags_device(self, buf_ptr, nbytes, offs_ptr, nframes, out_ptr):
        """Device-resident variant: raw device pointers in and out."""
        rc = self.L.etlg_frame_tags(self.h, C.c_void_p(buf_ptr), nbytes, C.c_void_p(offs_ptr), nframes,
                                    abi.F_INPUT_ON_DEVICE | abi.F_OUTPUT_ON_DEVICE, C.c_void_p(out_ptr))
        if rc != 0:
            raise RuntimeError(f"etlg_frame_tags failed: {rc}")

    def shard_plan(self, buf, offsets, n_shards):
        """[(f0, f1)] commit-aligned, byte-balanced frame ranges of a HOST stream (etlg_shard_plan: classified and cut on the device)."""
        a = np.ascontiguousarray(buf, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint32)
        return self._shard_plan(a.ctypes.data, len(a), o.ctypes.data, len(o) - 1, n_shards, 0)

    def shard_plan_device(self, buf_ptr, nbytes, offs_ptr, nframes, n_shards):
        """... of a device-resident stream (raw device pointers)."""
        return self._shard_plan(buf_ptr, nbytes, offs_ptr, nframes, n_shards, abi.F_INPUT_ON_DEVICE)

    def _shard_plan(self, buf_ptr, nbytes, offs_ptr, nframes, n_shards, flags):
        cuts = np.zeros(n_shards + 1, dtype=np.uint64)
        rc = self.L.etlg_shard_plan(self.h, C.c_void_p(buf_ptr), nbytes, C.c_void_p(offs_ptr), nframes, n_shards, flags, cuts.ctypes.data)
        if rc != abi.OK:
            raise self.last_error()
        return [(int(cuts[k]), int(cuts[k + 1])) for k in range(n_shards)]

    def shard_replay(self, buf, offsets):
        """Applies the control stream of an EARLIER shard to this context and leaves it outside any transaction (etlg_shard_replay)."""
        a = np.ascontiguousarray(buf, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint32)
        rc = self.L.etlg_shard_replay(self.h, a.ctypes.data if len(a) else None, len(a), o.ctypes.data, len(o) - 1)
        if rc != abi.OK:
            raise self.last_error()

    def control_stream(self, buf_ptr, nbytes, offs_ptr, nframes, on_device=True):
        """The control stream of a frame range, extracted on the device (etlg_control_stream): (bytes np.uint8, offsets np.uint32,
        tag of the range's last frame). `buf_ptr` / `offs_ptr`: device pointers (or host addresses with on_device=False)."""
        flags = abi.F_INPUT_ON_DEVICE if on_device else 0
        nb, nf, last = C.c_size_t(), C.c_size_t(), C.c_uint32()
        out, offs = np.zeros(1 << 16, dtype=np.uint8), np.zeros(1 << 10, dtype=np.uint32)
        for _ in range(2):
            rc = self.L.etlg_control_stream(self.h, C.c_void_p(buf_ptr), nbytes, C.c_void_p(offs_ptr), nframes, flags, out.ctypes.data, out.size,
                                            offs.ctypes.data, offs.size, C.byref(nb), C.byref(nf), C.byref(last))
            if rc == abi.OK:
                return out[:nb.value].copy(), offs[:nf.value + 1].copy(), int(last.value)
            if rc != abi.InvalidArgument or (nb.value <= out.size and nf.value + 1 <= offs.size):
                raise self.last_error()
            out, offs = np.zeros(nb.value + 64, dtype=np.uint8), np.zeros(nf.value + 2, dtype=np.uint32)
        raise self.last_error()

    def scan_boundaries_device(self, buf_ptr, nbytes, out_ptr, cap):
        """Record-boundary scan of a device-resident stream into a device array of `cap` u32 entries; returns nframes."""
        n = C.c_size_t()
        rc = self.L.etlg_scan_boundaries(self.h, C.c_void_p(buf_ptr), nbytes, abi.F_INPUT_ON_DEVICE | abi.F_OUTPUT_ON_DEVICE,
                                         C.c_void_p(out_ptr), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError(f"etlg_scan_boundaries failed: {rc}")
        return int(n.value)

    def scan_boundaries(self, buf, max_frames=None):
        """Record-boundary scan of a host buffer on the device: np.uint32 offsets (nframes + 1)."""
        import numpy as np
        a = np.ascontiguousarray(buf, dtype=np.uint8)
        cap = (len(a) // 5 + 3) if max_frames is None else max_frames + 1
        out = np.empty(cap, dtype=np.uint32)
        n = C.c_size_t()
        rc = self.L.etlg_scan_boundaries(self.h, a.ctypes.data, len(a), 0, out.ctypes.data, cap, C.byref(n))
        if rc != 0:
            raise RuntimeError(f"etlg_scan_boundaries failed: {rc}")
        return out[:n.value + 1].copy()

    def debug_scan(self):
        """(scans that needed a hinted rerun, scans that fell back to the one-lane walk)."""
        out = (C.c_ulonglong * 2)()
        self.L.etlg_ctx_debug_scan(self.h, out)
        return int(out[0]), int(out[1])
