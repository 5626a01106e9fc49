//! Staging batcher: accumulates raw CopyData frames of the logical replication stream into one contiguous buffer plus the
//! `u32` offsets sidecar `etlg_decode` takes (the host learns every frame length when it receives the message, so the sidecar
//! costs nothing), and tracks what the apply loop still needs per message while the bytes wait to be decoded.
//!
//! Where it goes in supabase/etl: `ApplyLoop::handle_replication_message_and_flush`
//! (crates/etl/src/replication/apply.rs:1910-1946) pushes ONE decoded event per message into `EventBatch` and asks
//! "size hint reached?" after every push (apply.rs:1932-1935). With the GPU stage the loop pushes the message's BYTES here
//! instead, keeps answering keepalives itself (apply.rs:2053-2073, the batcher never sees them), and decodes when
//! `should_flush()` says so — the byte budget below, a Commit (`end_batch` of apply.rs:2339-2360), or the flush deadline of
//! `set_flush_deadline_if_needed` (apply.rs:1927). The resulting `Vec<Event>` goes through `EventBatch::push` unchanged, so
//! the reference's own size hints (`TableRow::new`, data/table_row.rs:28-32) still decide where `write_events` batches are cut.
use std::time::{Duration, Instant};

/// What the apply loop remembers about a staged frame (apply.rs:2039-2051: `start_lsn` / `end_lsn` of the XLogData message;
/// used for `update_last_commit_end_lsn` and status updates once the frame's event has been delivered).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct FrameMeta {
    pub wal_start: u64,
    pub wal_end: u64,
    pub tag: u8,
}

/// One staged batch: `frames` holds `'d' | be32 len | payload` per message, exactly as on the socket.
pub struct StagedBatch {
    pub frames: Vec<u8>,
    pub offsets: Vec<u32>,
    pub meta: Vec<FrameMeta>,
    /// No Relation ('R') / Message ('M') frame was staged: the caller may pass `ETLG_F_NO_CONTROL`.
    pub control_free: bool,
}

pub struct StagingBatcher {
    cap_bytes: usize,
    max_fill: Duration,
    cur: StagedBatch,
    first_frame_at: Option<Instant>,
    open_transaction: bool,
}

impl StagingBatcher {
    /// `cap_bytes`: soft byte budget of a batch (64 MiB is what the kernels are tuned on; the hard limit of one
    /// `etlg_decode` is 2 GiB on the single-pass kernels). `max_fill`: the pipeline's `max_batch_fill_duration`.
    pub fn new(cap_bytes: usize, max_fill: Duration) -> Self {
        Self { cap_bytes, max_fill, cur: Self::empty(cap_bytes), first_frame_at: None, open_transaction: false }
    }

    fn empty(cap: usize) -> StagedBatch {
        let mut offsets = Vec::with_capacity(cap / 64 + 2);
        offsets.push(0);
        StagedBatch { frames: Vec::with_capacity(cap + (1 << 20)), offsets, meta: Vec::with_capacity(cap / 64 + 1), control_free: true }
    }

    /// Stages the payload of one CopyData message (`payload[0] == b'w'` for XLogData; keepalives stay with the caller).
    /// Re-creates the 5-byte CopyData header tokio-postgres stripped.
    pub fn push_xlog_data(&mut self, payload: &[u8]) {
        debug_assert!(payload.first() == Some(&b'w') && payload.len() >= 26);
        let be = |b: &[u8]| u64::from_be_bytes(b.try_into().unwrap());
        let tag = payload[25];
        self.cur.frames.push(b'd');
        self.cur.frames.extend_from_slice(&((payload.len() as u32 + 4).to_be_bytes()));
        self.cur.frames.extend_from_slice(payload);
        self.cur.offsets.push(self.cur.frames.len() as u32);
        self.cur.meta.push(FrameMeta { wal_start: be(&payload[1..9]), wal_end: be(&payload[9..17]), tag });
        match tag {
            b'B' => self.open_transaction = true,
            b'C' => self.open_transaction = false,
            b'R' | b'M' => self.cur.control_free = false,
            _ => {}
        }
        self.first_frame_at.get_or_insert_with(Instant::now);
    }

    pub fn is_empty(&self) -> bool {
        self.cur.meta.is_empty()
    }

    pub fn staged_bytes(&self) -> usize {
        self.cur.frames.len()
    }

    /// Decode now? Byte budget reached, or the fill deadline passed (apply.rs:1927, 1962-1967). Cutting between two frames
    /// of one transaction is fine: the context carries `remote_final_lsn` and the next ordinal across batches
    /// (apply.rs:942-963), on the device for batches in flight.
    pub fn should_flush(&self) -> bool {
        if self.is_empty() {
            return false;
        }
        self.cur.frames.len() >= self.cap_bytes || self.first_frame_at.is_some_and(|t| t.elapsed() >= self.max_fill)
    }

    /// A Commit was just staged and the caller wants low latency (the reference ends a batch early on some commits:
    /// `end_batch`, apply.rs:2339-2360).
    pub fn at_transaction_boundary(&self) -> bool {
        !self.open_transaction
    }

    /// Hands the staged batch over and starts a new one.
    pub fn take(&mut self) -> StagedBatch {
        self.first_frame_at = None;
        std::mem::replace(&mut self.cur, Self::empty(self.cap_bytes))
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    fn xlog(tag: u8, lsn: u64, body: &[u8]) -> Vec<u8> {
        let mut p = vec![b'w'];
        p.extend_from_slice(&lsn.to_be_bytes());
        p.extend_from_slice(&lsn.to_be_bytes());
        p.extend_from_slice(&0i64.to_be_bytes());
        p.push(tag);
        p.extend_from_slice(body);
        p
    }

    #[test]
    fn frames_are_reframed_as_copy_data_with_a_sidecar() {
        let mut b = StagingBatcher::new(1 << 20, Duration::from_secs(1));
        b.push_xlog_data(&xlog(b'B', 0x10, &[0u8; 20]));
        b.push_xlog_data(&xlog(b'I', 0x18, &[0u8; 30]));
        b.push_xlog_data(&xlog(b'C', 0x20, &[0u8; 25]));
        assert!(b.at_transaction_boundary());
        let s = b.take();
        assert_eq!(s.offsets.len(), 4);
        assert_eq!(s.frames[0], b'd');
        assert_eq!(u32::from_be_bytes(s.frames[1..5].try_into().unwrap()) as usize + 1, s.offsets[1] as usize);
        assert_eq!(s.meta[1], FrameMeta { wal_start: 0x18, wal_end: 0x18, tag: b'I' });
        assert!(s.control_free);
        assert!(b.is_empty());
    }

    #[test]
    fn relation_frames_clear_the_control_free_hint() {
        let mut b = StagingBatcher::new(1 << 20, Duration::from_secs(1));
        b.push_xlog_data(&xlog(b'R', 0x10, &[0u8; 12]));
        assert!(!b.take().control_free);
    }
}
