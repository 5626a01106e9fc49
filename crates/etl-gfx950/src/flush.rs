//! The apply loop's LSN bookkeeping, reproduced for a loop that decodes whole batches: what `ApplyLoop` updates message by
//! message (crates/etl/src/replication/apply.rs:2037-2051 `update_last_received_lsn`; :1918-1928 `update_last_commit_end_lsn`
//! after every pushed event; :2000 `last_commit_end_lsn.take()` when a batch is dispatched) from the staged frames' metadata
//! and the decoded events.
//!
//! Two moments exist per frame now instead of one:
//!   * RECEIPT — the frame is staged (`on_frame`): `last_received_lsn = max(last_received_lsn, wal_start, wal_end)`, exactly what
//!     the reference does before it looks at the message body (apply.rs:2039-2043). Keepalives never reach the batcher; the loop
//!     keeps calling `on_keepalive` for them (apply.rs:2055-2057).
//!   * DELIVERY — the frame's event comes back from the device and is pushed into `EventBatch` (`on_event`): a Commit's `end_lsn`
//!     becomes `last_commit_end_lsn` if it is larger (apply.rs:833-845), as after `self.state.event_batch.push(..)` (:1918-1923).
//! Status updates keep the reference's rule: the flush LSN reported to Postgres only advances from durable writes
//! (`update_last_flush_lsn`, apply.rs:855-858), never from receipt or decode — staging more frames than have been delivered cannot
//! move it. `is_idle` (apply.rs:885-889) additionally requires that no staged frame is waiting for its event.
use etl::event::Event;

use crate::batcher::FrameMeta;

#[derive(Debug, Default, Clone)]
pub struct FlushTracker {
    last_received_lsn: u64,
    last_flush_lsn: u64,
    last_commit_end_lsn: Option<u64>,
    /// frames staged (in a buffer being filled or in flight) whose events have not been delivered yet
    undelivered_frames: u64,
    /// a Begin has been delivered without its Commit (`handling_transaction`, apply.rs:937-939: `remote_final_lsn.is_some()`)
    in_transaction: bool,
}

impl FlushTracker {
    pub fn new(start_lsn: u64) -> Self {
        Self { last_received_lsn: start_lsn, last_flush_lsn: start_lsn, ..Default::default() }
    }

    /// apply.rs:2039-2043: both ends of the XLogData message advance the received position, on receipt.
    pub fn on_frame(&mut self, m: &FrameMeta) {
        self.last_received_lsn = self.last_received_lsn.max(m.wal_start).max(m.wal_end);
        self.undelivered_frames += 1;
    }

    /// apply.rs:2055-2057.
    pub fn on_keepalive(&mut self, wal_end: u64) {
        self.last_received_lsn = self.last_received_lsn.max(wal_end);
    }

    /// The frames of a collected batch have been turned into events (filtered frames yield none): nothing of it is pending.
    pub fn on_batch_delivered(&mut self, nframes: usize) {
        self.undelivered_frames = self.undelivered_frames.saturating_sub(nframes as u64);
    }

    /// apply.rs:1918-1923, per event pushed into `EventBatch`, in stream order.
    pub fn on_event(&mut self, e: &Event) {
        match e {
            Event::Begin(_) => self.in_transaction = true,
            Event::Commit(c) => {
                self.in_transaction = false;
                let end: u64 = c.end_lsn.into();
                self.last_commit_end_lsn = Some(self.last_commit_end_lsn.map_or(end, |old| old.max(end)));
            }
            _ => {}
        }
    }

    /// apply.rs:2000: the dispatched batch carries the commit end LSN seen so far; `Accepted` results hand it back
    /// (`restore_commit_end_lsn`, apply.rs:1808-1810), `Durable` ones advance the flush position.
    pub fn take_commit_end_lsn(&mut self) -> Option<u64> {
        self.last_commit_end_lsn.take()
    }
    pub fn restore_commit_end_lsn(&mut self, lsn: Option<u64>) {
        if let Some(l) = lsn {
            self.last_commit_end_lsn = Some(self.last_commit_end_lsn.map_or(l, |old| old.max(l)));
        }
    }
    /// apply.rs:855-858 (`process_syncing_tables_after_flush` → durable progress).
    pub fn on_durable(&mut self, commit_end_lsn: u64) {
        self.last_flush_lsn = self.last_flush_lsn.max(commit_end_lsn);
        debug_assert!(self.last_received_lsn >= self.last_flush_lsn);
    }

    pub fn last_received_lsn(&self) -> u64 {
        self.last_received_lsn
    }
    pub fn last_flush_lsn(&self) -> u64 {
        self.last_flush_lsn
    }

    /// apply.rs:885-889 + nothing staged: only then may a status update report the received position as flushed
    /// (`effective_flush_lsn`, apply.rs:906-912).
    pub fn is_idle(&self, has_unresolved_batch_work: bool) -> bool {
        !self.in_transaction && !has_unresolved_batch_work && self.last_commit_end_lsn.is_none() && self.undelivered_frames == 0
    }
    pub fn effective_flush_lsn(&self, has_unresolved_batch_work: bool) -> u64 {
        if self.is_idle(has_unresolved_batch_work) { self.last_received_lsn } else { self.last_flush_lsn }
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    #[test]
    fn receipt_moves_the_received_position_only() {
        let mut t = FlushTracker::new(0x100);
        t.on_frame(&FrameMeta { wal_start: 0x110, wal_end: 0x120, tag: b'B' });
        t.on_frame(&FrameMeta { wal_start: 0x120, wal_end: 0x118, tag: b'I' });
        assert_eq!(t.last_received_lsn(), 0x120);
        assert_eq!(t.last_flush_lsn(), 0x100);
        assert!(!t.is_idle(false));   // staged frames wait for their events
        assert_eq!(t.effective_flush_lsn(false), 0x100);
        t.on_batch_delivered(2);
        t.on_keepalive(0x200);
        assert!(t.is_idle(false));
        assert_eq!(t.effective_flush_lsn(false), 0x200);
    }
}
