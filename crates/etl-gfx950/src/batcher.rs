//! Staging batcher: accumulates raw CopyData frames of the logical replication stream into PINNED 64 MiB buffers plus the
//! `u32` offsets sidecar `etlg_decode` takes (the host learns every frame length when it receives the message, so the sidecar
//! costs nothing), and remembers per staged frame what the apply loop still needs once the frame's event comes back.
//!
//! Where it goes in supabase/etl: `ApplyLoop::handle_replication_message_and_flush`
//! (crates/etl/src/replication/apply.rs:1910-1946) pushes ONE decoded event per message into `EventBatch` and asks
//! "size hint reached?" after every push (apply.rs:1932-1935). With the GPU stage the loop pushes the message's BYTES here
//! instead, keeps answering keepalives itself (apply.rs:2053-2073, the batcher never sees them), and hands a buffer to
//! `GpuDecoder::decode_async` when `should_flush()` says so — the byte budget below, a Commit (`end_batch` of
//! apply.rs:2339-2360), or the flush deadline of `set_flush_deadline_if_needed` (apply.rs:1927). The buffers form a ring
//! (two or more): while buffer k is uploaded and decoded (`ETLG_F_ASYNC`: the upload runs on the library's copy stream beside
//! the decode of buffer k-1), the loop fills buffer k+1; a buffer returns to the ring when its batch has been collected
//! (`GpuDecoder::finish`). The resulting `Vec<Event>` goes through `EventBatch::push` unchanged, so the reference's own size
//! hints (`TableRow::new`, data/table_row.rs:28-32) still decide where `write_events` batches are cut.
use std::os::raw::c_void;
use std::time::{Duration, Instant};

use etl::bail;
use etl::error::{ErrorKind, EtlResult};

use crate::ffi::{etlg_ctx, etlg_host_alloc, etlg_host_free, ETLG_OK};

/// What the apply loop remembers about a staged frame (apply.rs:2039-2051: `start_lsn` / `end_lsn` of the XLogData message;
/// `FlushTracker::on_frame` consumes it for `update_last_received_lsn`).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct FrameMeta {
    pub wal_start: u64,
    pub wal_end: u64,
    pub tag: u8,
}

/// `cap` bytes of page-locked host memory from `etlg_host_alloc` (hipHostMalloc): the H2D copy of a staged batch is one DMA,
/// asynchronous to the host.
pub struct PinnedBuf {
    ptr: *mut u8,
    cap: usize,
}

unsafe impl Send for PinnedBuf {}

impl PinnedBuf {
    pub fn new(ctx: *mut etlg_ctx, cap: usize) -> EtlResult<Self> {
        let mut p: *mut c_void = std::ptr::null_mut();
        if unsafe { etlg_host_alloc(ctx, cap, &mut p) } != ETLG_OK || p.is_null() {
            bail!(ErrorKind::ConfigError, "Pinned staging buffer could not be allocated", format!("{cap} bytes"));
        }
        Ok(Self { ptr: p as *mut u8, cap })
    }
    pub fn as_ptr(&self) -> *const u8 {
        self.ptr
    }
    pub fn capacity(&self) -> usize {
        self.cap
    }
    fn slice_mut(&mut self, at: usize, len: usize) -> &mut [u8] {
        debug_assert!(at + len <= self.cap);
        unsafe { std::slice::from_raw_parts_mut(self.ptr.add(at), len) }
    }
    /// (for the crate's other staging buffers: copy.rs)
    pub(crate) fn bytes_mut(&mut self, at: usize, len: usize) -> &mut [u8] {
        self.slice_mut(at, len)
    }
    pub(crate) fn bytes(&self, at: usize, len: usize) -> &[u8] {
        debug_assert!(at + len <= self.cap);
        unsafe { std::slice::from_raw_parts(self.ptr.add(at), len) }
    }
}

impl Drop for PinnedBuf {
    fn drop(&mut self) {
        unsafe { etlg_host_free(self.ptr as *mut c_void) };
    }
}

/// One buffer of the ring: `frames` holds `'d' | be32 len | payload` per message, exactly as on the socket; `offsets` the
/// `nframes + 1` byte offsets (`u32`, pinned as well: they travel with the bytes).
pub struct StagedBatch {
    frames: PinnedBuf,
    offsets: PinnedBuf,
    pub len: usize,
    pub nframes: usize,
    pub meta: Vec<FrameMeta>,
    /// No Relation ('R') / Message ('M') frame was staged: the caller may pass `ETLG_F_NO_CONTROL`.
    pub control_free: bool,
}

impl StagedBatch {
    fn new(ctx: *mut etlg_ctx, cap_bytes: usize) -> EtlResult<Self> {
        // the smallest frame the stream carries is a 26-byte XLogData header + 5 bytes of CopyData framing
        let max_frames = cap_bytes / 31 + 2;
        let mut offsets = PinnedBuf::new(ctx, (max_frames + 1) * 4)?;
        offsets.slice_mut(0, 4).copy_from_slice(&0u32.to_ne_bytes());
        Ok(Self { frames: PinnedBuf::new(ctx, cap_bytes)?, offsets, len: 0, nframes: 0, meta: Vec::with_capacity(cap_bytes / 64 + 1), control_free: true })
    }
    pub fn frames_ptr(&self) -> *const u8 {
        self.frames.as_ptr()
    }
    pub fn offsets_ptr(&self) -> *const u32 {
        self.offsets.as_ptr() as *const u32
    }
    fn room_for(&self, payload_len: usize) -> bool {
        self.len + payload_len + 5 <= self.frames.capacity() && (self.nframes + 2) * 4 <= self.offsets.capacity()
    }
    fn reset(&mut self) {
        self.len = 0;
        self.nframes = 0;
        self.meta.clear();
        self.control_free = true;
    }
}

pub struct StagingBatcher {
    cap_bytes: usize,
    max_fill: Duration,
    cur: StagedBatch,
    free: Vec<StagedBatch>,
    first_frame_at: Option<Instant>,
    open_transaction: bool,
}

impl StagingBatcher {
    /// `ring`: buffers in rotation (at least two: one being filled, one in flight). `cap_bytes`: byte budget of a batch
    /// (64 MiB is what the kernels are tuned on; one `etlg_decode` takes at most 2 GiB on the single-pass kernels, and the
    /// sidecar is `u32`). `max_fill`: the pipeline's `max_batch_fill_duration`.
    pub fn new(ctx: *mut etlg_ctx, ring: usize, cap_bytes: usize, max_fill: Duration) -> EtlResult<Self> {
        if ring < 2 || cap_bytes < 64 || cap_bytes > (1usize << 31) - (1 << 20) {
            bail!(ErrorKind::ConfigError, "Invalid staging ring", format!("{ring} buffers of {cap_bytes} bytes"));
        }
        let cur = StagedBatch::new(ctx, cap_bytes)?;
        let mut free = Vec::with_capacity(ring - 1);
        for _ in 1..ring {
            free.push(StagedBatch::new(ctx, cap_bytes)?);
        }
        Ok(Self { cap_bytes, max_fill, cur, free, first_frame_at: None, open_transaction: false })
    }

    /// Does the payload still fit the buffer being filled? When it does not, the caller takes the buffer (`take`) first.
    pub fn fits(&self, payload: &[u8]) -> bool {
        self.cur.room_for(payload.len())
    }

    /// Could the payload be staged at all — does it fit an EMPTY buffer of the ring? A single XLogData message may be larger than
    /// any staging buffer (a pgoutput tuple can approach 1 GB; the ring holds 64 MiB buffers): such a message never goes through
    /// `push_xlog_data`; the caller drains the GPU stage and hands it to the reference's own per-message path.
    pub fn can_stage(&self, payload: &[u8]) -> bool {
        payload.len() + 5 <= self.cap_bytes
    }

    /// Stages the payload of one CopyData message (`payload[0] == b'w'` for XLogData; keepalives stay with the caller).
    /// Re-creates the 5-byte CopyData header tokio-postgres stripped. An XLogData message is at least its 25-byte header
    /// plus the pgoutput tag (postgres-replication: `XLogDataBody`, call site apply.rs:2037-2051).
    pub fn push_xlog_data(&mut self, payload: &[u8]) -> EtlResult<()> {
        if payload.len() < 26 || payload[0] != b'w' {
            bail!(ErrorKind::InvalidData, "Replication message is not an XLogData message with a pgoutput body", format!("{} bytes", payload.len()));
        }
        if !self.cur.room_for(payload.len()) {
            bail!(ErrorKind::InvalidState, "Staging buffer is full", "take() the staged batch before pushing more frames");
        }
        let be = |b: &[u8]| u64::from_be_bytes(b.try_into().unwrap());
        let tag = payload[25];
        let at = self.cur.len;
        let dst = self.cur.frames.slice_mut(at, payload.len() + 5);
        dst[0] = b'd';
        dst[1..5].copy_from_slice(&((payload.len() as u32 + 4).to_be_bytes()));
        dst[5..].copy_from_slice(payload);
        self.cur.len = at + payload.len() + 5;
        self.cur.nframes += 1;
        let end = self.cur.len as u32;   // < 2^31: the ring's buffers are smaller (new)
        let n = self.cur.nframes;
        self.cur.offsets.slice_mut(n * 4, 4).copy_from_slice(&end.to_ne_bytes());
        self.cur.meta.push(FrameMeta { wal_start: be(&payload[1..9]), wal_end: be(&payload[9..17]), tag });
        match tag {
            b'B' => self.open_transaction = true,
            b'C' => self.open_transaction = false,
            b'R' | b'M' => self.cur.control_free = false,
            _ => {}
        }
        self.first_frame_at.get_or_insert_with(Instant::now);
        Ok(())
    }

    pub fn is_empty(&self) -> bool {
        self.cur.nframes == 0
    }

    pub fn staged_bytes(&self) -> usize {
        self.cur.len
    }

    /// Decode now? Byte budget reached, or the fill deadline passed (apply.rs:1927, 1962-1967). Cutting between two frames
    /// of one transaction is fine: the context carries `remote_final_lsn` and the next ordinal across batches
    /// (apply.rs:942-963), on the device for batches in flight.
    pub fn should_flush(&self) -> bool {
        if self.is_empty() {
            return false;
        }
        self.cur.len + (1 << 16) >= self.cap_bytes || self.first_frame_at.is_some_and(|t| t.elapsed() >= self.max_fill)
    }

    /// A Commit was just staged and the caller wants low latency (the reference ends a batch early on some commits:
    /// `end_batch`, apply.rs:2339-2360).
    pub fn at_transaction_boundary(&self) -> bool {
        !self.open_transaction
    }

    /// A buffer is free to be filled while the taken one is in flight. `None`: every other buffer of the ring is still in
    /// flight — collect the oldest batch (`GpuDecoder::finish`) and `recycle` its buffer first (back-pressure, like the
    /// reference pausing while a flush result is pending, apply.rs:1962-1967).
    pub fn take(&mut self) -> Option<StagedBatch> {
        let next = self.free.pop()?;
        self.first_frame_at = None;
        Some(std::mem::replace(&mut self.cur, next))
    }

    /// Returns a collected batch's buffer to the ring.
    pub fn recycle(&mut self, mut b: StagedBatch) {
        b.reset();
        self.free.push(b);
    }
}

#[cfg(test)]
mod tests {
    // These tests need a device context for the pinned allocations: they run where libetl_gfx950.so finds an MI355X.
    use super::*;
    use crate::ffi::{etlg_ctx_create, etlg_ctx_destroy};

    fn xlog(tag: u8, lsn: u64, body: &[u8]) -> Vec<u8> {
        let mut p = vec![b'w'];
        p.extend_from_slice(&lsn.to_be_bytes());
        p.extend_from_slice(&lsn.to_be_bytes());
        p.extend_from_slice(&0i64.to_be_bytes());
        p.push(tag);
        p.extend_from_slice(body);
        p
    }

    fn with_ctx(f: impl FnOnce(*mut etlg_ctx)) {
        let mut ctx = std::ptr::null_mut();
        assert_eq!(unsafe { etlg_ctx_create(0, &mut ctx) }, ETLG_OK);
        f(ctx);
        unsafe { etlg_ctx_destroy(ctx) };
    }

    #[test]
    fn frames_are_reframed_as_copy_data_with_a_sidecar() {
        with_ctx(|ctx| {
            let mut b = StagingBatcher::new(ctx, 2, 1 << 20, Duration::from_secs(1)).unwrap();
            b.push_xlog_data(&xlog(b'B', 0x10, &[0u8; 20])).unwrap();
            b.push_xlog_data(&xlog(b'I', 0x18, &[0u8; 30])).unwrap();
            b.push_xlog_data(&xlog(b'C', 0x20, &[0u8; 25])).unwrap();
            assert!(b.at_transaction_boundary());
            let s = b.take().unwrap();
            assert_eq!(s.nframes, 3);
            let frames = unsafe { std::slice::from_raw_parts(s.frames_ptr(), s.len) };
            let offs = unsafe { std::slice::from_raw_parts(s.offsets_ptr(), s.nframes + 1) };
            assert_eq!(frames[0], b'd');
            assert_eq!(u32::from_be_bytes(frames[1..5].try_into().unwrap()) as usize + 1, offs[1] as usize);
            assert_eq!(offs[3] as usize, s.len);
            assert_eq!(s.meta[1], FrameMeta { wal_start: 0x18, wal_end: 0x18, tag: b'I' });
            assert!(s.control_free);
            assert!(b.is_empty());
            assert!(b.take().is_none());   // the ring's other buffer is in flight
            b.recycle(s);
        });
    }

    #[test]
    fn relation_frames_clear_the_control_free_hint_and_short_payloads_are_refused() {
        with_ctx(|ctx| {
            let mut b = StagingBatcher::new(ctx, 2, 1 << 20, Duration::from_secs(1)).unwrap();
            b.push_xlog_data(&xlog(b'R', 0x10, &[0u8; 12])).unwrap();
            assert!(!b.take().unwrap().control_free);
            assert!(b.push_xlog_data(&xlog(b'I', 0x10, &[])[..25]).is_err());   // an XLogData header without a pgoutput byte
            assert!(b.push_xlog_data(b"k\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0\0").is_err());
        });
    }
}
