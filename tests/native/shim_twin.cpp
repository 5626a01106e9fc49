// TEST INFRASTRUCTURE — an executable twin of the Rust shim (crates/etl-gfx950), which this image cannot compile (no rustc): the
// SAME call sequence through the C ABI, statement for statement, so that the ring rotation, the flush arithmetic and the drop
// order are at least run once against the library and checked (tests/test_shim_twin.py: events against the oracle's, LSN
// bookkeeping against a model of the reference's per-message rules, apply.rs:2039-2051 / 1918-1928 / 2000 / 885-912).
//
//   batcher.rs   PinnedBuf / StagedBatch / StagingBatcher     -> struct PinnedBuf, StagedBatch, StagingBatcher below
//   lib.rs       GpuDecoder::decode_async / finish / decode_unstaged, InFlight (+ its Drop)
//   flush.rs     FlushTracker
//   patches/apply_rs.diff   the seam in handle_replication_message_and_flush: gpu_dispatch / gpu_collect -> twin_run's loop
//   copy.rs      CopyStaging, GpuDecoder::copy_decode_async / copy_finish, CopyInFlight (+ its Drop) -> twin_copy_run
//
// The library is whatever the process has loaded (the path comes from the caller: the product build on a GPU box, the emulator build in
// the CPU suite). Every function is looked up by name: a symbol the shim binds and the library lacks fails the run.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <deque>
#include <string>
#include <vector>

#include "../../include/etlg.h"

namespace {

struct Api {
  int32_t (*host_alloc)(etlg_ctx*, size_t, void**);
  void (*host_free)(void*);
  int32_t (*decode)(etlg_ctx*, const uint8_t*, size_t, const uint32_t*, size_t, uint32_t, etlg_batch**);
  int32_t (*batch_sync)(etlg_ctx*, etlg_batch*);
  int32_t (*batch_download)(etlg_ctx*, etlg_batch*);
  int32_t (*batch_view_get)(const etlg_batch*, etlg_batch_view*);
  void (*batch_free)(etlg_batch*);
  const etlg_error* (*last_error)(const etlg_ctx*);
  int32_t (*copy_decode)(etlg_ctx*, int32_t, const uint8_t*, size_t, const uint32_t*, size_t, uint32_t, etlg_batch**);
  bool load(const char* path) {
    void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
#define SYM(field, name) field = (decltype(field))dlsym(h, name); if (!field) return false;
    SYM(host_alloc, "etlg_host_alloc") SYM(host_free, "etlg_host_free") SYM(decode, "etlg_decode") SYM(batch_sync, "etlg_batch_sync")
    SYM(batch_download, "etlg_batch_download") SYM(batch_view_get, "etlg_batch_view_get") SYM(batch_free, "etlg_batch_free")
    SYM(last_error, "etlg_last_error") SYM(copy_decode, "etlg_copy_decode")
#undef SYM
    return true;
  }
};
Api A;

// ---------------------------------------------------------------- batcher.rs
struct FrameMeta { uint64_t wal_start, wal_end; uint8_t tag; };

struct PinnedBuf {   // batcher.rs: PinnedBuf::new / Drop
  uint8_t* ptr = nullptr; size_t cap = 0;
  bool alloc(etlg_ctx* ctx, size_t n) { void* p = nullptr; if (A.host_alloc(ctx, n, &p) != ETLG_OK || !p) return false; ptr = (uint8_t*)p; cap = n; return true; }
  void release() { if (ptr) A.host_free(ptr); ptr = nullptr; cap = 0; }
};

struct StagedBatch {
  PinnedBuf frames, offsets;
  size_t len = 0, nframes = 0;
  std::vector<FrameMeta> meta;
  bool control_free = true;
  bool create(etlg_ctx* ctx, size_t cap_bytes) {   // StagedBatch::new
    const size_t max_frames = cap_bytes / 31 + 2;
    if (!offsets.alloc(ctx, (max_frames + 1) * 4)) return false;
    const uint32_t z = 0; memcpy(offsets.ptr, &z, 4);
    return frames.alloc(ctx, cap_bytes);
  }
  bool room_for(size_t payload_len) const { return len + payload_len + 5 <= frames.cap && (nframes + 2) * 4 <= offsets.cap; }
  void reset() { len = 0; nframes = 0; meta.clear(); control_free = true; }
  void release() { frames.release(); offsets.release(); }
};

struct StagingBatcher {
  size_t cap_bytes = 0;
  StagedBatch* cur = nullptr;
  std::vector<StagedBatch*> free_;
  bool open_transaction = false;
  bool create(etlg_ctx* ctx, size_t ring, size_t cap) {   // StagingBatcher::new
    if (ring < 2 || cap < 64) return false;
    cap_bytes = cap;
    cur = new StagedBatch();
    if (!cur->create(ctx, cap)) return false;
    for (size_t i = 1; i < ring; i++) { auto* b = new StagedBatch(); if (!b->create(ctx, cap)) return false; free_.push_back(b); }
    return true;
  }
  bool fits(size_t n) const { return cur->room_for(n); }
  bool can_stage(size_t n) const { return n + 5 <= cap_bytes; }
  int push_xlog_data(const uint8_t* payload, size_t n) {   // StagingBatcher::push_xlog_data
    if (n < 26 || payload[0] != 'w') return 1;
    if (!cur->room_for(n)) return 2;
    auto be = [](const uint8_t* b) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = v << 8 | b[i]; return v; };
    const uint8_t tag = payload[25];
    const size_t at = cur->len;
    uint8_t* dst = cur->frames.ptr + at;
    dst[0] = 'd';
    const uint32_t l = (uint32_t)n + 4;
    dst[1] = (uint8_t)(l >> 24); dst[2] = (uint8_t)(l >> 16); dst[3] = (uint8_t)(l >> 8); dst[4] = (uint8_t)l;
    memcpy(dst + 5, payload, n);
    cur->len = at + n + 5;
    cur->nframes += 1;
    const uint32_t end = (uint32_t)cur->len;
    memcpy(cur->offsets.ptr + cur->nframes * 4, &end, 4);
    cur->meta.push_back(FrameMeta{be(payload + 1), be(payload + 9), tag});
    if (tag == 'B') open_transaction = true;
    else if (tag == 'C') open_transaction = false;
    else if (tag == 'R' || tag == 'M') cur->control_free = false;
    return 0;
  }
  bool is_empty() const { return cur->nframes == 0; }
  bool should_flush() const { return !is_empty() && cur->len + (1 << 16) >= cap_bytes; }   // (the fill deadline is the caller's: `cuts`)
  StagedBatch* take() {   // StagingBatcher::take
    if (free_.empty()) return nullptr;
    StagedBatch* next = free_.back(); free_.pop_back();
    StagedBatch* out = cur; cur = next;
    return out;
  }
  void recycle(StagedBatch* b) { b->reset(); free_.push_back(b); }
  void release() { if (cur) { cur->release(); delete cur; } for (auto* b : free_) { b->release(); delete b; } cur = nullptr; free_.clear(); }
};

// ---------------------------------------------------------------- flush.rs
struct FlushTracker {
  uint64_t last_received_lsn = 0, last_flush_lsn = 0;
  bool has_commit_end = false; uint64_t last_commit_end_lsn = 0;
  uint64_t undelivered_frames = 0;
  bool in_transaction = false;
  void on_frame(const FrameMeta& m) { uint64_t v = last_received_lsn; if (m.wal_start > v) v = m.wal_start; if (m.wal_end > v) v = m.wal_end; last_received_lsn = v; undelivered_frames += 1; }
  void on_keepalive(uint64_t wal_end) { if (wal_end > last_received_lsn) last_received_lsn = wal_end; }
  void on_batch_delivered(size_t n) { undelivered_frames = undelivered_frames >= n ? undelivered_frames - n : 0; }
  void on_begin() { in_transaction = true; }
  void on_commit(uint64_t end) { in_transaction = false; last_commit_end_lsn = has_commit_end ? (last_commit_end_lsn > end ? last_commit_end_lsn : end) : end; has_commit_end = true; }
  bool is_idle(bool unresolved) const { return !in_transaction && !unresolved && !has_commit_end && undelivered_frames == 0; }
  uint64_t effective_flush_lsn(bool unresolved) const { return is_idle(unresolved) ? last_received_lsn : last_flush_lsn; }
};

// ---------------------------------------------------------------- lib.rs
struct InFlight { etlg_batch* batch; StagedBatch* staged; };

}  // namespace

extern "C" {

// per delivered batch: the host view (valid during the call), the frames the batch held, its status; the caller copies what it wants
typedef void (*twin_on_batch)(void* user, const etlg_batch_view* view, uint64_t nframes, int32_t rc, int32_t err_code, int64_t err_frame);

struct twin_trace {   // one row per input message, one per delivered batch (kind 0 / 1)
  uint32_t kind; uint32_t index;          // message index | frames delivered so far
  uint64_t last_received_lsn, last_commit_end_lsn, effective_flush_lsn;
  uint32_t has_commit_end, undelivered, in_transaction, in_flight;
};

// Runs the seam of patches/apply_rs.diff over a recorded stream. `stream` / `offs`: CopyData-framed messages as on the socket (XLogData
// and keepalives); the 5-byte CopyData header is stripped per message, as tokio-postgres does before the loop sees a payload.
// `cuts[i] != 0`: a dispatch is forced after message i (the fill deadline / a commit with an exit intent). Returns 0, or the step that failed.
int32_t twin_run(const char* libpath, etlg_ctx* ctx, const uint8_t* stream, const uint32_t* offs, uint32_t nmsgs, uint32_t ring, uint32_t cap_bytes,
                 const uint8_t* cuts, twin_on_batch cb, void* user, twin_trace* trace, uint32_t trace_cap, uint32_t* ntrace, uint64_t start_lsn) {
  if (!A.load(libpath)) return 100;
  StagingBatcher batcher;
  if (!batcher.create(ctx, ring, cap_bytes)) return 101;
  FlushTracker tracker; tracker.last_received_lsn = start_lsn; tracker.last_flush_lsn = start_lsn;   // FlushTracker::new
  std::deque<InFlight> in_flight;
  uint64_t frames_delivered = 0;
  uint32_t nt = 0;
  int32_t fail = 0;
  auto note = [&](uint32_t kind, uint32_t index) {
    if (nt < trace_cap) trace[nt] = twin_trace{kind, index, tracker.last_received_lsn, tracker.last_commit_end_lsn, tracker.effective_flush_lsn(false),
                                               tracker.has_commit_end ? 1u : 0u, (uint32_t)tracker.undelivered_frames, tracker.in_transaction ? 1u : 0u, (uint32_t)in_flight.size()};
    nt++;
  };
  // GpuDecoder::finish + the delivery loop of gpu_collect
  auto deliver = [&](etlg_batch* batch, size_t nframes, int32_t rc_sync) {
    int32_t code = 0; int64_t frame = -1;
    if (rc_sync != ETLG_OK) { const etlg_error* e = A.last_error(ctx); if (e) { code = e->code; frame = e->frame_index; } }
    if (A.batch_download(ctx, batch) != ETLG_OK) { A.batch_free(batch); fail = 110; return; }
    etlg_batch_view v; memset(&v, 0, sizeof(v));
    A.batch_view_get(batch, &v);
    tracker.on_batch_delivered(nframes);
    for (uint64_t i = 0; i < v.n_events; i++) {   // tracker.on_event per event pushed into EventBatch, in stream order
      if (v.ev_kind[i] == 'B') tracker.on_begin();
      else if (v.ev_kind[i] == 'C') { uint64_t end; memcpy(&end, v.fixed + v.ev_body_off[i], 8); tracker.on_commit(end); }
    }
    frames_delivered += nframes;
    if (cb) cb(user, &v, nframes, rc_sync, code, frame);
    A.batch_free(batch);
    if (rc_sync != ETLG_OK) fail = 1;   // `status?`: fail-fast, after the events before the failing frame were delivered
  };
  auto collect = [&](bool wait) {   // gpu_collect
    for (;;) {
      if (in_flight.empty() || (!wait && in_flight.size() < 2)) return;
      InFlight f = in_flight.front(); in_flight.pop_front();
      const size_t nframes = f.staged->meta.size();
      const int32_t rc = A.batch_sync(ctx, f.batch);
      deliver(f.batch, nframes, rc);      // GpuDecoder::finish: sync, download, view, events, free ...
      batcher.recycle(f.staged);          // ... and only then does the staged buffer go back to the ring
      note(1, (uint32_t)frames_delivered);
      if (fail || !wait) return;
    }
  };
  auto dispatch = [&]() {   // gpu_dispatch
    if (batcher.is_empty()) return;
    StagedBatch* staged;
    for (;;) { staged = batcher.take(); if (staged) break; collect(true); if (fail) return; }
    etlg_batch* b = nullptr;
    const uint32_t flags = ETLG_F_ASYNC | ETLG_F_OUTPUT_ON_DEVICE | (staged->control_free ? ETLG_F_NO_CONTROL : 0u);
    (void)A.decode(ctx, staged->frames.ptr, staged->len, (const uint32_t*)staged->offsets.ptr, staged->nframes, flags, &b);   // GpuDecoder::decode_async
    if (!b) { batcher.recycle(staged); fail = 120; return; }
    in_flight.push_back(InFlight{b, staged});
  };
  std::vector<uint8_t> unstaged;
  for (uint32_t i = 0; i < nmsgs && !fail; i++) {
    const uint8_t* payload = stream + offs[i] + 5;
    const size_t n = offs[i + 1] - offs[i] - 5;
    if (n >= 1 && payload[0] == 'k') {   // keepalives stay with the loop (apply.rs:2053-2057)
      uint64_t we = 0; for (int k = 0; k < 8; k++) we = we << 8 | payload[1 + k];
      tracker.on_keepalive(we);
      note(0, i);
      continue;
    }
    if (!batcher.can_stage(n)) {
      // a message larger than a whole staging buffer: everything staged before it is decoded and delivered first, then the message goes
      // through the library by itself, from an unpinned buffer, synchronously (GpuDecoder::decode_unstaged) — the context carries the
      // transaction state across, so the batches behind it continue where it ends
      dispatch(); if (fail) break;
      while (!in_flight.empty() && !fail) collect(true);
      if (fail) break;
      unstaged.resize(n + 5 + 64);
      unstaged[0] = 'd'; const uint32_t l = (uint32_t)n + 4;
      unstaged[1] = (uint8_t)(l >> 24); unstaged[2] = (uint8_t)(l >> 16); unstaged[3] = (uint8_t)(l >> 8); unstaged[4] = (uint8_t)l;
      memcpy(unstaged.data() + 5, payload, n);
      const uint32_t o2[2] = {0u, (uint32_t)n + 5};
      auto be = [](const uint8_t* b) { uint64_t v = 0; for (int k = 0; k < 8; k++) v = v << 8 | b[k]; return v; };
      if (n >= 26) tracker.on_frame(FrameMeta{be(payload + 1), be(payload + 9), payload[25]});
      note(0, i);
      etlg_batch* b = nullptr;
      const int32_t rc = A.decode(ctx, unstaged.data(), n + 5, o2, 1, ETLG_F_OUTPUT_ON_DEVICE, &b);
      if (!b) { fail = 130; break; }
      deliver(b, 1, rc);
      note(1, (uint32_t)frames_delivered);
      continue;
    }
    if (!batcher.fits(n)) { dispatch(); if (fail) break; }
    if (batcher.push_xlog_data(payload, n)) { fail = 140; break; }
    tracker.on_frame(batcher.cur->meta.back());
    note(0, i);
    if (batcher.should_flush() || (cuts && cuts[i])) { dispatch(); if (fail) break; }
    collect(false);
  }
  if (!fail) { dispatch(); }
  while (!in_flight.empty() && fail != 110) {   // the queue is drained: on the happy path by gpu_collect(true), after an error by InFlight's Drop
    if (fail) { InFlight f = in_flight.front(); in_flight.pop_front(); (void)A.batch_sync(ctx, f.batch); A.batch_free(f.batch); batcher.recycle(f.staged); }   // Drop for InFlight: sync, free, THEN the pinned buffer
    else collect(true);
  }
  note(2, (uint32_t)frames_delivered);
  *ntrace = nt;
  batcher.release();      // (PinnedBuf's Drop: only after every batch that read the buffers is gone)
  return fail == 1 ? 0 : fail;   // a decode error is a result (reported through the callback), not a failure of the twin
}

}  // extern "C"

// ---------------------------------------------------------------- copy.rs
namespace {
struct CopyStaging {   // copy.rs: CopyStaging::new / fits / push / clear
  PinnedBuf rows, offsets;
  size_t len = 0, nrows = 0;
  bool create(etlg_ctx* ctx, size_t cap_bytes) {
    const size_t max_rows = cap_bytes / 2 + 2;
    if (!rows.alloc(ctx, cap_bytes) || !offsets.alloc(ctx, (max_rows + 1) * 4)) return false;
    set_offset(0, 0);
    return true;
  }
  void set_offset(size_t i, uint32_t v) { memcpy(offsets.ptr + i * 4, &v, 4); }
  bool fits(size_t n) const { return len + n <= rows.cap && (nrows + 2) * 4 <= offsets.cap && len + n < 0xFFFFFFFFull; }
  bool push(const uint8_t* row, size_t n) {
    if (!fits(n)) return false;
    memcpy(rows.ptr + len, row, n);
    len += n; nrows += 1;
    set_offset(nrows, (uint32_t)len);
    return true;
  }
  void clear() { len = 0; nrows = 0; }
  void release() { rows.release(); offsets.release(); }
};
struct CopyInFlight { etlg_batch* batch; CopyStaging* staged; };
}  // namespace

extern "C" {

// The table-copy seam (patches/table_copy_rs.diff) with the ASYNC pair of copy.rs: the items of a CopyOutStream (`rows` / `offs`: one
// COPY text row per item) are staged into a ring of `ring` pinned CopyStaging buffers of `cap_bytes`; a full one is enqueued
// (copy_decode_async) and the oldest batch in flight is collected (copy_finish) once `ring - 1` are out — the stream never waits for a
// decode it does not need. Fail-fast like the stream (table_copy.rs:88-92): the first bad row ends the run after the rows before it were
// delivered; the batches still in flight are dropped — batch first, then its pinned staging (CopyInFlight's Drop).
// cb: per collected batch, the host view, the rows the batch held, status.
int32_t twin_copy_run(const char* libpath, etlg_ctx* ctx, int32_t schema_slot, const uint8_t* rows, const uint32_t* offs, uint32_t nitems,
                      uint32_t ring, uint32_t cap_bytes, twin_on_batch cb, void* user, uint64_t* rows_delivered) {
  if (!A.load(libpath)) return 100;
  if (ring < 2) return 101;
  std::vector<CopyStaging*> free_;
  for (uint32_t i = 0; i < ring; i++) { auto* st = new CopyStaging(); if (!st->create(ctx, cap_bytes)) return 102; free_.push_back(st); }
  CopyStaging* cur = free_.back(); free_.pop_back();
  std::deque<CopyInFlight> in_flight;
  int32_t fail = 0;
  uint64_t delivered = 0;
  auto finish = [&]() {   // GpuDecoder::copy_finish on the oldest batch
    CopyInFlight f = in_flight.front(); in_flight.pop_front();
    const int32_t rc = A.batch_sync(ctx, f.batch);
    int32_t code = 0; int64_t frame = -1;
    if (rc != ETLG_OK) { const etlg_error* e = A.last_error(ctx); if (e) { code = e->code; frame = e->frame_index; } }
    if (A.batch_download(ctx, f.batch) != ETLG_OK) { A.batch_free(f.batch); f.staged->clear(); free_.push_back(f.staged); fail = 110; return; }
    etlg_batch_view v; memset(&v, 0, sizeof(v));
    A.batch_view_get(f.batch, &v);
    delivered += v.n_events;
    if (cb) cb(user, &v, f.staged->nrows, rc, code, frame);
    A.batch_free(f.batch);
    f.staged->clear(); free_.push_back(f.staged);   // the staging comes back only now
    if (rc != ETLG_OK) fail = 1;
  };
  auto dispatch = [&]() {   // GpuDecoder::copy_decode_async
    if (cur->nrows == 0) return;
    while (free_.empty() && !fail) finish();
    if (fail) return;
    etlg_batch* b = nullptr;
    (void)A.copy_decode(ctx, schema_slot, cur->rows.ptr, cur->len, (const uint32_t*)cur->offsets.ptr, cur->nrows, ETLG_F_ASYNC | ETLG_F_OUTPUT_ON_DEVICE, &b);
    if (!b) { fail = 120; return; }
    in_flight.push_back(CopyInFlight{b, cur});
    cur = free_.back(); free_.pop_back();
  };
  for (uint32_t i = 0; i < nitems && !fail; i++) {
    const uint8_t* row = rows + offs[i];
    const size_t n = offs[i + 1] - offs[i];
    if (!cur->fits(n)) { dispatch(); if (fail) break; }
    if (!cur->push(row, n)) { fail = 140; break; }   // (a row larger than a whole staging buffer: the caller sizes the ring for its rows)
  }
  if (!fail) dispatch();
  while (!in_flight.empty() && fail != 110) {
    if (fail) { CopyInFlight f = in_flight.front(); in_flight.pop_front(); A.batch_free(f.batch); f.staged->clear(); free_.push_back(f.staged); }   // Drop for CopyInFlight: the batch (the library finishes it), THEN the staging
    else finish();
  }
  *rows_delivered = delivered;
  cur->release(); delete cur;
  for (auto* st : free_) { st->release(); delete st; }
  return fail == 1 ? 0 : fail;
}

}  // extern "C"
