//! Table copy through the GPU stage: the rows of a `CopyOutStream` are staged as bytes and decoded a batch at a time
//! (`etlg_copy_decode`), instead of one `parse_table_row_from_postgres_copy_bytes` call per stream item
//! (crates/etl/src/postgres/stream/table_copy.rs:70-103, crates/etl/src/postgres/codec/table_row.rs:47-254).
//!
//! ```text
//! CopyOutStream ──row bytes──▶ CopyStaging ──rows + offsets (pinned)──▶ GpuDecoder::copy_decode ──arena──▶ materialize::table_rows ──▶ Vec<TableRow>
//! ```
//! The per-row `TableCopyPayloadMetadata` (table_copy.rs:84: the byte length of the text row) stays exact: `CopyStaging` keeps
//! every row's length, and the batch's `payload_bytes[0]` is their sum over the rows that decoded.
use etl::bail;
use etl::data::TableRow;
use etl::error::{ErrorKind, EtlResult};

use crate::batcher::PinnedBuf;
use crate::ffi::*;
use crate::materialize;
use crate::GpuDecoder;

/// Rows of one table-copy batch in page-locked memory: the row payloads back to back exactly as `CopyOutStream` yields them (one
/// COPY text row per item, no `'d'` framing) and `nrows + 1` byte offsets.
pub struct CopyStaging {
    rows: PinnedBuf,
    offsets: PinnedBuf,
    len: usize,
    nrows: usize,
}

impl CopyStaging {
    /// `cap_bytes` of row payload; offsets are sized for rows of at least two bytes (an empty-looking row is still `"\n"` plus a
    /// field).
    pub fn new(decoder: &GpuDecoder, cap_bytes: usize) -> EtlResult<Self> {
        let max_rows = cap_bytes / 2 + 2;
        let mut s = Self { rows: PinnedBuf::new(decoder.raw(), cap_bytes)?, offsets: PinnedBuf::new(decoder.raw(), (max_rows + 1) * 4)?, len: 0, nrows: 0 };
        s.set_offset(0, 0);
        Ok(s)
    }
    fn set_offset(&mut self, i: usize, v: u32) {
        self.offsets.bytes_mut(i * 4, 4).copy_from_slice(&v.to_ne_bytes());
    }
    /// Does the next stream item fit? (If not: `GpuDecoder::copy_decode` what is staged, then `clear`.)
    pub fn fits(&self, row: &[u8]) -> bool {
        self.len + row.len() <= self.rows.capacity() && (self.nrows + 2) * 4 <= self.offsets.capacity() && self.len + row.len() < u32::MAX as usize
    }
    /// Stages one `CopyOutStream` item.
    pub fn push(&mut self, row: &[u8]) -> EtlResult<()> {
        if !self.fits(row) {
            bail!(ErrorKind::InvalidState, "Table-copy staging buffer is full", format!("{} bytes staged, row of {} bytes", self.len, row.len()));
        }
        self.rows.bytes_mut(self.len, row.len()).copy_from_slice(row);
        self.len += row.len();
        self.nrows += 1;
        let (n, l) = (self.nrows, self.len as u32);
        self.set_offset(n, l);
        Ok(())
    }
    pub fn nrows(&self) -> usize {
        self.nrows
    }
    pub fn staged_bytes(&self) -> usize {
        self.len
    }
    /// Byte length of staged row `i`: its `TableCopyPayloadMetadata` (table_copy.rs:84).
    pub fn row_len(&self, i: usize) -> u64 {
        let at = |k: usize| u32::from_ne_bytes(self.offsets.bytes(k * 4, 4).try_into().expect("four bytes")) as u64;
        at(i + 1) - at(i)
    }
    pub fn clear(&mut self) {
        self.len = 0;
        self.nrows = 0;
    }
}

impl GpuDecoder {
    /// Decodes the staged rows against schema slot `schema_slot` (what `table_ready` returned for the table's
    /// `ReplicatedTableSchema`; its replicated columns are the `column_schemas` `TableCopyStream::wrap` was given,
    /// table_copy.rs:64-66). Fail-fast like the stream (table_copy.rs:88-92): on a bad row the rows BEFORE it are returned together
    /// with the reference's error for that row.
    pub fn copy_decode(&mut self, schema_slot: u32, staged: &CopyStaging) -> (Vec<TableRow>, EtlResult<()>) {
        let mut batch = std::ptr::null_mut();
        let rc = unsafe {
            etlg_copy_decode(self.ctx, schema_slot as i32, staged.rows.as_ptr(), staged.len, staged.offsets.as_ptr() as *const u32, staged.nrows, 0, &mut batch)
        };
        if batch.is_null() {
            return (Vec::new(), Err(self.last_error()));
        }
        let status = if rc == ETLG_OK { Ok(()) } else { Err(self.last_error()) };
        let mut view = std::mem::MaybeUninit::<etlg_batch_view>::uninit();
        unsafe { etlg_batch_view_get(batch, view.as_mut_ptr()) };
        let view = unsafe { view.assume_init() };
        let rows = unsafe { materialize::table_rows(&view) };
        unsafe { etlg_batch_free(batch) };
        match rows {
            Ok(r) => (r, status),
            Err(e) => (Vec::new(), Err(e)),
        }
    }

    /// The ASYNC form: enqueues the staged rows (`ETLG_F_ASYNC | ETLG_F_OUTPUT_ON_DEVICE`, host input — the library uploads the pinned
    /// buffers on its copy stream beside the decode of the batch before, include/etlg.h) and returns at once, so that the stream keeps
    /// filling the next `CopyStaging` while this one decodes (two or three in rotation, like the WAL batcher's ring). The staging
    /// travels with the handle and comes back from `copy_finish`. Fewer than 32 in flight, finished in issue order.
    /// Twin: tests/native/shim_twin.cpp `twin_copy_run` (tests/test_shim_twin.py: rows against the oracle's, fail-fast, drop order).
    pub fn copy_decode_async(&mut self, schema_slot: u32, staged: CopyStaging) -> Result<CopyInFlight, (CopyStaging, etl::error::EtlError)> {
        let mut batch = std::ptr::null_mut();
        let rc = unsafe {
            etlg_copy_decode(self.ctx, schema_slot as i32, staged.rows.as_ptr(), staged.len, staged.offsets.as_ptr() as *const u32, staged.nrows,
                             ETLG_F_ASYNC | ETLG_F_OUTPUT_ON_DEVICE, &mut batch)
        };
        if batch.is_null() {
            let _ = rc;
            let e = self.last_error();
            return Err((staged, e));
        }
        Ok(CopyInFlight { ctx: self.ctx, batch, staged: Some(staged) })
    }

    /// Collects a batch issued by `copy_decode_async`: waits for it (`etlg_batch_sync`), copies the arena to the host and materialises
    /// the rows. Fail-fast like the stream (table_copy.rs:88-92): on a bad row the rows BEFORE it come back with the reference's error;
    /// the batches behind it are not affected (table-copy batches share no state).
    pub fn copy_finish(&mut self, f: CopyInFlight) -> (Vec<TableRow>, CopyStaging, EtlResult<()>) {
        let mut f = f;
        let batch = std::mem::replace(&mut f.batch, std::ptr::null_mut());
        let staged = f.staged.take().expect("a batch in flight holds its staging");
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
        let rows = unsafe { materialize::table_rows(&view) };
        unsafe { etlg_batch_free(batch) };
        match rows {
            Ok(r) => (r, staged, status),
            Err(e) => (Vec::new(), staged, Err(e)),
        }
    }
}

/// A table-copy batch in flight. Dropping it without `copy_finish` frees the batch FIRST (the library waits for its kernels and its
/// upload: `etlg_batch_free` finishes a pending batch) and only then the pinned staging — the order `InFlight` keeps for WAL batches
/// (lib.rs; both exercised by tests/native/shim_twin.cpp).
pub struct CopyInFlight {
    ctx: *mut etlg_ctx,
    batch: *mut etlg_batch,
    staged: Option<CopyStaging>,
}

impl Drop for CopyInFlight {
    fn drop(&mut self) {
        let _ = self.ctx;
        if !self.batch.is_null() {
            unsafe { etlg_batch_free(self.batch) };
            self.batch = std::ptr::null_mut();
        }
        self.staged.take();
    }
}
