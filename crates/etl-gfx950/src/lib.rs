//! etl-gfx950 — the Rust side of the MI355X-native decode stage for supabase/etl.
//!
//! ```text
//! ReplicationMessageStream ──bytes──▶ StagingBatcher ──64 MiB + sidecar──▶ GpuDecoder::decode ──arena──▶ materialize::events ──▶ EventBatch
//! ```
//! * [`ffi`]         raw binding of `include/etlg.h` (checked against the header by the etl-gfx950 test suite);
//! * [`batcher`]     frames + offsets sidecar accumulation in a ring of pinned buffers, flush policy (apply.rs:1910-1967);
//! * [`copy`]        table copy: `CopyOutStream` items staged as bytes, `GpuDecoder::copy_decode` → `Vec<TableRow>` (table_copy.rs:70-103);
//! * [`flush`]       `last_received_lsn` / `last_commit_end_lsn` / effective flush LSN for a loop that decodes batches (apply.rs:2039-2051,
//!                   1918-1928, 2000, 885-912);
//! * [`materialize`] arena → `Event` / `TableRow` / `Cell`, DEFERRED cells finished with the reference's own parser;
//!
//! None of this has met a Rust compiler (the etl-gfx950 image has none). What HAS been executed is the twin of the call sequence in C++
//! through the same C ABI — tests/native/shim_twin.cpp, run by tests/test_shim_twin.py on the MI355X against the oracle's events and a
//! model of the reference's LSN rules: `StagingBatcher` (ring of pinned buffers, `push_xlog_data`, `take` / `recycle`), `decode_async` /
//! `finish`, `decode_unstaged`, `FlushTracker`, `InFlight`'s drop order. A change here wants the same change there.
//! * [`GpuDecoder`]  safe wrapper of one context: side inputs mirror `SchemaStore` / `StateStore` / `SharedTableCache`
//!                   (crates/etl/src/store/schema/base.rs:19-69, store/state/base.rs:25-139, replication/table_cache.rs:88-154).
pub mod batcher;
pub mod copy;
pub mod ffi;
pub mod flush;
pub mod materialize;

use std::ffi::{CStr, CString};
use std::ptr;

use etl::error::{ErrorKind, EtlError, EtlResult};
use etl::etl_error;
use etl::event::Event;

use crate::batcher::StagedBatch;
use crate::ffi::*;

/// `etlg_error_kind` → the reference's `ErrorKind` (crates/etl/src/error.rs:85-170). Descriptions are the reference's own
/// static strings (`etlg_err_table`), so `EtlError::kind()` / `description()` read as if the CPU decoder had failed.
fn kind_of(k: i32) -> ErrorKind {
    match k {
        ETLG_ConversionError => ErrorKind::ConversionError,
        ETLG_InvalidData => ErrorKind::InvalidData,
        ETLG_ValidationError => ErrorKind::ValidationError,
        ETLG_InvalidState => ErrorKind::InvalidState,
        ETLG_MissingTableSchema => ErrorKind::MissingTableSchema,
        ETLG_CorruptedTableSchema => ErrorKind::CorruptedTableSchema,
        ETLG_DeserializationError => ErrorKind::DeserializationError,
        ETLG_SourceConnectionFailed => ErrorKind::SourceConnectionFailed,
        ETLG_IoError => ErrorKind::IoError,
        ETLG_UnsupportedValueInDestination => ErrorKind::UnsupportedValueInDestination,
        ETLG_NullValuesNotSupportedInArrayInDestination => ErrorKind::NullValuesNotSupportedInArrayInDestination,
        _ => ErrorKind::Unknown,
    }
}

pub struct GpuDecoder {
    ctx: *mut etlg_ctx,
}

// One context per apply-loop stream, used from one task at a time (`&mut self`), like the loop's own state.
unsafe impl Send for GpuDecoder {}

impl GpuDecoder {
    pub fn new(hip_device: i32) -> EtlResult<Self> {
        let mut ctx = ptr::null_mut();
        let rc = unsafe { etlg_ctx_create(hip_device, &mut ctx) };
        if rc != ETLG_OK {
            let why = unsafe { CStr::from_ptr(etlg_create_error()) }.to_string_lossy().into_owned();
            return Err(etl_error!(ErrorKind::ConfigError, "MI355X decode context could not be created", why));
        }
        debug_assert_eq!(unsafe { etlg_abi_version() }, ETLG_ABI_VERSION);
        Ok(Self { ctx })
    }

    fn last_error(&self) -> EtlError {
        let e = unsafe { &*etlg_last_error(self.ctx) };
        let desc: &'static str = if e.description.is_null() { "GPU decode failed" } else { unsafe { CStr::from_ptr(e.description) }.to_str().unwrap_or("GPU decode failed") };
        let detail = if e.detail.is_null() { format!("frame {}", e.frame_index) } else { format!("{} (frame {})", unsafe { CStr::from_ptr(e.detail) }.to_string_lossy(), e.frame_index) };
        etl_error!(kind_of(e.kind), desc, detail)
    }

    /// `SchemaStore::store_table_schema` mirror: stored columns in attnum order.
    pub fn put_schema(&mut self, table_id: u32, snapshot_lsn: u64, schema: &str, table: &str, cols: &[(String, u32, i32, i32, bool, bool)]) -> EtlResult<()> {
        let names: Vec<CString> = cols.iter().map(|c| CString::new(c.0.as_str()).unwrap()).collect();
        let raw: Vec<etlg_col> = cols
            .iter()
            .zip(&names)
            .map(|(c, n)| etlg_col { name: n.as_ptr(), type_oid: c.1, type_modifier: c.2, attnum: c.3, nullable: c.4 as u8, primary_key: c.5 as u8, _pad: [0; 2] })
            .collect();
        let (s, t) = (CString::new(schema).unwrap(), CString::new(table).unwrap());
        match unsafe { etlg_schema_put(self.ctx, table_id, snapshot_lsn, s.as_ptr(), t.as_ptr(), raw.len() as u32, raw.as_ptr()) } {
            ETLG_OK => Ok(()),
            _ => Err(self.last_error()),
        }
    }

    /// Table replication state as `should_apply_changes` sees it (apply.rs:2836-2867).
    pub fn set_table_state(&mut self, table_id: u32, state_kind: i32, lsn: u64) {
        unsafe { etlg_table_state(self.ctx, table_id, state_kind, lsn) };
    }

    /// `SharedTableCache::note_ready` after a table copy (table_cache.rs:122); returns the schema slot.
    pub fn table_ready(&mut self, table_id: u32, snapshot_lsn: u64, replication_mask: &[u8], identity_mask: &[u8]) -> EtlResult<u32> {
        debug_assert_eq!(replication_mask.len(), identity_mask.len());
        let slot = unsafe { etlg_table_ready(self.ctx, table_id, snapshot_lsn, replication_mask.as_ptr(), identity_mask.as_ptr(), replication_mask.len() as u32) };
        if slot < 0 { Err(self.last_error()) } else { Ok(slot as u32) }
    }

    /// `SharedTableCache::remove_table` (table_cache.rs:131-145).
    pub fn forget_table(&mut self, table_id: u32) {
        unsafe { etlg_table_forget(self.ctx, table_id) };
    }

    pub fn set_worker(&mut self, table_sync_table: Option<u32>, bootstrap_snapshot_lsn: u64) {
        let (kind, id) = match table_sync_table { Some(t) => (ETLG_WORKER_TABLE_SYNC, t), None => (ETLG_WORKER_APPLY, 0) };
        unsafe { etlg_ctx_set_worker(self.ctx, kind, id, bootstrap_snapshot_lsn) };
    }

    /// The raw context, for `StagingBatcher::new` (its buffers are pinned through the context's device).
    pub fn raw(&self) -> *mut etlg_ctx {
        self.ctx
    }

    /// Enqueues one staged batch: `ETLG_F_ASYNC | ETLG_F_OUTPUT_ON_DEVICE`, host input. The library uploads the pinned buffer on its
    /// copy stream beside the decode of the batch issued before (include/etlg.h) and returns at once; the staged buffer travels with
    /// the handle and must not be touched until `finish` gives it back. Keep fewer than 32 batches in flight and finish them in
    /// issue order.
    pub fn decode_async(&mut self, staged: StagedBatch) -> Result<InFlight, (StagedBatch, EtlError)> {
        let mut batch = ptr::null_mut();
        // (ETLG_F_FINISH_CELLS: arrays arrive typed and floats exact — materialize.rs then parses no array text on the host)
        let flags = ETLG_F_ASYNC | ETLG_F_OUTPUT_ON_DEVICE | ETLG_F_FINISH_CELLS | if staged.control_free { ETLG_F_NO_CONTROL } else { 0 };
        let rc = unsafe { etlg_decode(self.ctx, staged.frames_ptr(), staged.len, staged.offsets_ptr(), staged.nframes, flags, &mut batch) };
        if batch.is_null() {
            let _ = rc;
            let e = self.last_error();
            return Err((staged, e));
        }
        Ok(InFlight { ctx: self.ctx, batch, staged: Some(staged) })
    }

    /// Has the batch been decoded? Never blocks... it is `finish` that waits. (The C ABI has no non-blocking probe: a loop that
    /// wants one keeps a batch in flight for one fill period of the next buffer — by then it is done — and calls `finish`.)
    ///
    /// Collects a batch issued by `decode_async`: waits for it (`etlg_batch_sync`, issue order), copies the arena to the host
    /// (`etlg_batch_download`) and materialises the events. On a decode error (fail-fast, apply.rs:2475-2481) the events BEFORE the
    /// failing frame are returned with the error. The staged buffer comes back for `StagingBatcher::recycle`.
    pub fn finish(&mut self, f: InFlight, schemas: &mut dyn materialize::SlotSchemas) -> (Vec<Event>, StagedBatch, EtlResult<()>) {
        let mut f = f;
        let (batch, staged) = f.take();
        let rc = unsafe { etlg_batch_sync(self.ctx, batch) };
        let status = if rc == ETLG_OK { Ok(()) } else { Err(self.last_error()) };
        if unsafe { etlg_batch_download(self.ctx, batch) } != ETLG_OK {
            let e = self.last_error();
            unsafe { etlg_batch_free(batch) };
            return (Vec::new(), staged, Err(e));
        }
        let mut view = std::mem::MaybeUninit::<etlg_batch_view>::uninit();
        unsafe { etlg_batch_view_get(batch, view.as_mut_ptr()) };
        let view = unsafe { view.assume_init() };
        let events = unsafe { materialize::events(&view, schemas) };
        unsafe { etlg_batch_free(batch) };
        match events {
            Ok(ev) => (ev, staged, status),
            Err(e) => (Vec::new(), staged, Err(e)),
        }
    }

    /// One message that does not fit a staging buffer (`StagingBatcher::can_stage` says no: a pgoutput tuple may approach 1 GB), decoded by
    /// itself: re-framed into an ordinary (unpinned) buffer and handed to the library synchronously. Call it with nothing in flight
    /// (dispatch and collect first — the event order holds, and the context carries `remote_final_lsn` / the next ordinal from the
    /// batches before into this one and on into the batches behind). Twin: tests/native/shim_twin.cpp, the `!can_stage` branch.
    pub fn decode_unstaged(&mut self, payload: &[u8], schemas: &mut dyn materialize::SlotSchemas) -> (Vec<Event>, EtlResult<()>) {
        let mut framed = Vec::with_capacity(payload.len() + 5 + 64);
        framed.push(b'd');
        framed.extend_from_slice(&((payload.len() as u32 + 4).to_be_bytes()));
        framed.extend_from_slice(payload);
        let offs = [0u32, framed.len() as u32];
        framed.resize(framed.len() + 64, 0);            // (head room the library's staging copy does not need; kept for symmetry with the ring)
        let mut batch = ptr::null_mut();
        let rc = unsafe { etlg_decode(self.ctx, framed.as_ptr(), offs[1] as usize, offs.as_ptr(), 1, ETLG_F_OUTPUT_ON_DEVICE, &mut batch) };
        if batch.is_null() {
            return (Vec::new(), Err(self.last_error()));
        }
        let status = if rc == ETLG_OK { Ok(()) } else { Err(self.last_error()) };
        if unsafe { etlg_batch_download(self.ctx, batch) } != ETLG_OK {
            let e = self.last_error();
            unsafe { etlg_batch_free(batch) };
            return (Vec::new(), Err(e));
        }
        let mut view = std::mem::MaybeUninit::<etlg_batch_view>::uninit();
        unsafe { etlg_batch_view_get(batch, view.as_mut_ptr()) };
        let view = unsafe { view.assume_init() };
        let events = unsafe { materialize::events(&view, schemas) };
        unsafe { etlg_batch_free(batch) };
        match events {
            Ok(ev) => (ev, status),
            Err(e) => (Vec::new(), Err(e)),
        }
    }

    /// Synchronous form: one staged batch in, its events out (`decode_async` + `finish`).
    pub fn decode(&mut self, staged: StagedBatch, schemas: &mut dyn materialize::SlotSchemas) -> (Vec<Event>, StagedBatch, EtlResult<()>) {
        match self.decode_async(staged) {
            Ok(f) => self.finish(f, schemas),
            Err((staged, e)) => (Vec::new(), staged, Err(e)),
        }
    }
}

/// A batch between `decode_async` and `finish`: the library's handle plus the pinned buffer it is reading.
pub struct InFlight {
    ctx: *mut etlg_ctx,
    batch: *mut etlg_batch,
    staged: Option<StagedBatch>,
}

unsafe impl Send for InFlight {}

impl InFlight {
    pub fn frames(&self) -> &[crate::batcher::FrameMeta] {
        &self.staged.as_ref().expect("batch already collected").meta
    }

    /// Takes the handle and the staged buffer out (for `finish`); what is left drops as a no-op.
    fn take(&mut self) -> (*mut etlg_batch, StagedBatch) {
        (std::mem::replace(&mut self.batch, ptr::null_mut()), self.staged.take().expect("batch already collected"))
    }
}

/// A batch that is dropped without `finish` — an apply loop that bails out with batches still queued (`status?` in gpu_collect) — must
/// not release its pinned buffer while the library's upload or kernels may still be reading it, and must not leave the context's
/// ASYNC chain with an unfinished link: the batch is waited for and freed FIRST, the staged buffer (hipHostFree) goes after it.
/// The decoder has to outlive its in-flight batches (drop the queue before the `GpuDecoder`).
impl Drop for InFlight {
    fn drop(&mut self) {
        if !self.batch.is_null() {
            unsafe {
                let _ = etlg_batch_sync(self.ctx, self.batch);
                etlg_batch_free(self.batch);
            }
            self.batch = ptr::null_mut();
        }
        // self.staged drops here, after the library is done with it
    }
}

/// The reference's own type sizes for `etlg_batch_size_hints` (the device evaluates Event::size_hint, event.rs:295-320, with
/// them): layouts are not ABI-stable, so they are taken from THIS build of the `etl` crate, once.
pub fn reference_size_model() -> etlg_size_model {
    use etl::data::{Cell, TableRow};
    use etl::event::{BeginEvent, CommitEvent, DeleteEvent, InsertEvent, RelationEvent, TruncateEvent, UpdateEvent};
    use etl::schema::ReplicatedTableSchema;
    use std::mem::size_of;
    etlg_size_model {
        begin_event: size_of::<BeginEvent>() as u32,
        commit_event: size_of::<CommitEvent>() as u32,
        insert_event: size_of::<InsertEvent>() as u32,
        update_event: size_of::<UpdateEvent>() as u32,
        delete_event: size_of::<DeleteEvent>() as u32,
        truncate_event: size_of::<TruncateEvent>() as u32,
        relation_event: size_of::<RelationEvent>() as u32,
        replicated_table_schema: size_of::<ReplicatedTableSchema>() as u32,
        table_row: size_of::<TableRow>() as u32,
        cell: size_of::<Cell>() as u32,
        _reserved: [0; 2],
    }
}

impl GpuDecoder {
    /// Device-resident decode + per-event size hints, for a batcher that wants the reference's cut points
    /// (EventBatch::push, apply.rs:656-657; flush rule :1932-1935) without materialising `Event`s first. Entries with
    /// ETLG_SIZE_HINT_INCOMPLETE set hold everything but the parts only the host can size (json / array cells, partial rows).
    pub fn size_hints(&mut self, batch: *mut etlg_batch, n_events: usize) -> EtlResult<Vec<u64>> {
        let model = reference_size_model();
        let mut out = vec![0u64; n_events];
        let rc = unsafe { etlg_batch_size_hints(self.ctx, batch, &model, 0, out.as_mut_ptr()) };
        if rc == ETLG_OK { Ok(out) } else { Err(self.last_error()) }
    }
}

/// The multi-GPU recipe (one `GpuDecoder` per device, one process or task per GPU — the reference itself has a single
/// apply task, apply.rs:1210-1336, so this is an extension a host opts into): `shard_plan` cuts a staged stream behind Commit frames,
/// every rank extracts the control stream of its own range (`control_stream`), the ranks exchange those few hundred bytes, every rank
/// replays the streams of the ranks before it (`shard_replay`, rank order) and then decodes its own range. Python twin with the
/// collectives: etl_amd/shard.py; the C entry points are the ones tests/test_shard_decode.py compares against that model.
impl GpuDecoder {
    /// Commit-aligned, byte-balanced frame ranges `[cuts[k], cuts[k + 1])` of a staged batch (etlg_shard_plan).
    pub fn shard_plan(&mut self, staged: &StagedBatch, n_shards: u32) -> EtlResult<Vec<u64>> {
        let mut cuts = vec![0u64; n_shards as usize + 1];
        let rc = unsafe { etlg_shard_plan(self.ctx, staged.frames_ptr(), staged.len, staged.offsets_ptr(), staged.nframes, n_shards, 0, cuts.as_mut_ptr()) };
        if rc == ETLG_OK { Ok(cuts) } else { Err(self.last_error()) }
    }

    /// The Relation / DDL-message transactions of a frame range of a staged batch, reduced to {Begin, control frames, Commit}
    /// (etlg_control_stream): what this rank broadcasts. Returns (bytes, offsets).
    pub fn control_stream(&mut self, staged: &StagedBatch, f0: usize, f1: usize) -> EtlResult<(Vec<u8>, Vec<u32>)> {
        let offs = unsafe { std::slice::from_raw_parts(staged.offsets_ptr(), staged.nframes + 1) };
        let (b0, b1) = (offs[f0] as usize, offs[f1] as usize);
        let rel: Vec<u32> = offs[f0..=f1].iter().map(|o| o - offs[f0]).collect();
        let (mut bytes, mut out_offs) = (vec![0u8; 1 << 16], vec![0u32; 1 << 10]);
        loop {
            let (mut nb, mut nf, mut last) = (0usize, 0usize, 0u32);
            let rc = unsafe {
                etlg_control_stream(self.ctx, staged.frames_ptr().add(b0), b1 - b0, rel.as_ptr(), f1 - f0, 0, bytes.as_mut_ptr(), bytes.len(),
                                    out_offs.as_mut_ptr(), out_offs.len(), &mut nb, &mut nf, &mut last)
            };
            if rc == ETLG_OK {
                bytes.truncate(nb);
                out_offs.truncate(nf + 1);
                return Ok((bytes, out_offs));
            }
            if nb <= bytes.len() && nf + 1 <= out_offs.len() { return Err(self.last_error()); }
            bytes = vec![0u8; nb + 64];
            out_offs = vec![0u32; nf + 2];
        }
    }

    /// Applies the control stream of an EARLIER rank's range (etlg_shard_replay): same schema slots in the same order on every rank;
    /// leaves the context outside any transaction, where a commit-aligned shard starts.
    pub fn shard_replay(&mut self, bytes: &[u8], offsets: &[u32]) -> EtlResult<()> {
        let nframes = offsets.len().saturating_sub(1);
        let rc = unsafe { etlg_shard_replay(self.ctx, if nframes == 0 { ptr::null() } else { bytes.as_ptr() }, bytes.len(), offsets.as_ptr(), nframes) };
        if rc == ETLG_OK { Ok(()) } else { Err(self.last_error()) }
    }
}

impl Drop for GpuDecoder {
    fn drop(&mut self) {
        unsafe { etlg_ctx_destroy(self.ctx) };
    }
}
