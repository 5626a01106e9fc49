// Host side of libetl_gfx950.so: the C ABI of include/etlg.h. One translation unit in four files —
//   host_state.h          context, batch and pool structures, kernel launchers
//   host_control.inc      control plane (schemas, Relation / DDL messages)
//   host.cpp              this file: error helpers, pools, lifecycle, side inputs, scan / tags / control-stream calls, etlg_decode and the
//                         ASYNC chain (decode_tail), batch calls
//   host_handoff.inc      etlg_batch_columns / _rowbinary / _protobuf / _size_hints
//   host_orchestrate.inc  side inputs, outputs, kernel ladder, control pre-pass, finish_batch
#include "host_state.h"

namespace {

int32_t set_error(etlg_ctx* c, int32_t code, int64_t frame, const char* detail = nullptr) {
  c->err.code = code;
  c->err.kind = kErrTable[code].kind;
  c->err.description = kErrTable[code].description;
  c->err_detail = detail ? detail : "";
  c->err.detail = c->err_detail.empty() ? nullptr : c->err_detail.c_str();
  c->err.frame_index = frame;
  return c->err.kind;
}
int32_t lib_error(etlg_ctx* c, int32_t kind, const char* what) {
  c->err.code = 0; c->err.kind = kind; c->err.description = what; c->err.detail = nullptr; c->err.frame_index = -1;
  return kind;
}
void clear_error(etlg_ctx* c) { c->err = etlg_error{}; c->err.frame_index = -1; c->err_detail.clear(); }

#define HIPCHK(ctx, call)                                                        \
  do {                                                                           \
    hipError_t _e = (call);                                                      \
    if (_e != hipSuccess) return lib_error((ctx), ETLG_DeviceError, hipGetErrorString(_e)); \
  } while (0)

#include "host_control.inc"   // control plane: schema slots, wire reader, JSON pull parser, Relation / DDL handlers, device slot table

void side_release(etlg_batch* b) { if (b->side) { b->side->users--; b->side = nullptr; } }
void side_use(etlg_batch* b, SideSet* ss) { if (b->side == ss) return; side_release(b); b->side = ss; ss->users++; }

void launch_raw(etlg_ctx* c, int which, const DecParams& p) {
  if (which == kFused) etlg_k_launch_fused((int)c->fq.blk, &p, &c->fq, c->stream);
  else if (which == kPlan) etlg_k_launch_plan(&p, &c->pq, c->stream);
  else if (which == kPlanPre) etlg_k_launch_plan_pre(&p, &c->pq, c->stream);
  else if (which == kCells) etlg_k_launch_cells(&p, &c->fq, c->stream);
  else if (which == kCopyCells) etlg_k_launch_copy_cells(&p, &c->fq, c->stream);
  else if (which == kRows) etlg_k_launch_rows(&p, &c->fq, c->stream);
  else etlg_k_launch(which, &p, c->stream);
}

void launch(etlg_ctx* c, int which, const DecParams& p) {
  if (c->prof) {
    ProfRec r; r.which = which;
    (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, c->stream);
    launch_raw(c, which, p);
    (void)hipEventRecord(r.b, c->stream);
    c->prof_recs.push_back(r);
  } else {
    launch_raw(c, which, p);
  }
}

void launch_copy(etlg_ctx* c, const CopyJob& j, const DecParams& p) {
  if (!j.nrows) return;
  ProfRec r; r.which = kCopy;
  if (c->prof) { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); (void)hipEventRecord(r.a, c->stream); }
  etlg_k_launch_copy(j.d_rows, j.d_row_offs, j.nrows, j.rows_len, j.ncols, j.rel_id, j.d_out, j.d_out_offs, j.lds, c->copy_lane_per_byte ? 1 : 0, &p, c->stream);
  if (c->prof) { (void)hipEventRecord(r.b, c->stream); c->prof_recs.push_back(r); }
}

// A table-copy batch whose first attempt was (or would have been) the rows -> arena kernel goes on as a batch of Insert frames:
// the rows are rewritten (copy.hip, with the reference's row-level errors) and every parameter that named the rows names the frames.
int32_t copy_use_frames(etlg_ctx* c, etlg_batch* b) {
  CopyJob& j = b->copy;
  // the frame buffers are the context's own: an ASYNC batch that needs them (k_cells cannot take its schema, or a kernel is forced) lets
  // everything enqueued before it finish — an earlier batch may still be reading them
  if (j.async) HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, c->d_copy_out.ensure(j.syn_len + 64)); HIPCHK(c, c->d_copy_out_offs.ensure(((size_t)j.nrows + 1) * 4));
  j.d_out = (uint8_t*)c->d_copy_out.p; j.d_out_offs = (uint32_t*)c->d_copy_out_offs.p;
  j.direct = false;
  DecParams& p = b->params;
  p.in = j.d_out; p.offs = j.d_out_offs; p.in_len = j.syn_len;
  b->len = (size_t)j.syn_len; b->d_in_ptr = j.d_out; b->dev_in = j.d_out; b->user_offs = j.d_out_offs;
  launch_copy(c, j, p);
  return ETLG_OK;
}

// A batch on the frame path that is decoded AGAIN (finish_batch: its rows go through the rewrite once more): the context's frame buffers may
// have been re-allocated since its first attempt — a later, larger ASYNC batch grew them (DevBuf::ensure frees the old allocation) — and
// what they hold now is a later batch's frames. Every name the batch holds for them is refreshed first; the streams are idle here (the
// first attempt and everything queued behind it have been waited for), so growing them again is safe too. (the stale pointers
// were a device use-after-free on the bad-row path of an ASYNC frames-at-enqueue batch.)
int32_t copy_repoint_frames(etlg_ctx* c, etlg_batch* b) {
  CopyJob& j = b->copy;
  if (!j.active || j.direct) return ETLG_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, c->d_copy_out.ensure(j.syn_len + 64)); HIPCHK(c, c->d_copy_out_offs.ensure(((size_t)j.nrows + 1) * 4));
  j.d_out = (uint8_t*)c->d_copy_out.p; j.d_out_offs = (uint32_t*)c->d_copy_out_offs.p;
  DecParams& p = b->params;
  p.in = j.d_out; p.offs = j.d_out_offs; p.in_len = j.syn_len;
  b->d_in_ptr = j.d_out; b->dev_in = j.d_out; b->user_offs = j.d_out_offs;
  return ETLG_OK;
}

// The multi-pass pipeline (also the exact first-error path).
void launch_multipass(etlg_ctx* c, const DecParams& p, bool classify_done) {
  if (!classify_done) { if (p.nframes) launch(c, 0, p); launch(c, 1, p); }
  if (p.nframes) launch(c, 3, p);
  launch(c, 4, p);
  if (p.nframes) launch(c, 5, p);
  launch(c, 6, p);
  if (c->mp_tail) { (void)hipEventRecord(c->mp_tail, c->stream); c->mp_tail_set = true; }   // a pre-pass that runs ahead waits for it (shared scratch)
}

OutSet* take_outset(etlg_ctx* c) {
  if (!c->out_pool.empty()) { OutSet* o = c->out_pool.back(); c->out_pool.pop_back(); return o; }
  return new OutSet();
}

uint32_t max_row_bytes(const etlg_ctx* c) {
  uint32_t m = 16;
  for (auto& s : c->slots) m = std::max(m, 2 * s->desc.row_bytes_full);
  return m;
}

void fill_view_common(etlg_batch* b) {
  etlg_ctx* c = b->ctx;
  b->slot_descs.clear();
  // the slots that existed when the batch's own control frames had been applied — not the ones a LATER batch's control plane has
  // created meanwhile (the control pre-pass runs ahead of the decode: etlg_ctx::ctl_stream)
  const size_t n = std::min(c->slots.size(), b->n_slots_view);
  for (size_t i = 0; i < n; i++) b->slot_descs.push_back(c->slots[i]->desc);
  b->v.n_slots = (uint32_t)b->slot_descs.size();
  b->v.slots = b->slot_descs.data();
}

// Reads the result block, resolves device vs host error, commits or rolls back
// the control-plane state and (for host output) copies the arenas back.
int32_t finish_batch(etlg_ctx* c, etlg_batch* b);
extern "C" int32_t finish_cells(etlg_ctx* c, etlg_batch* b, uint32_t what, etlg_finish_stats* stats);   // (host_handoff.inc)
int32_t drain_pending(etlg_ctx* c);
int32_t download_batch(etlg_ctx* c, etlg_batch* b);
struct BatchGuard {  // an etlg_decode that fails half way returns what the batch took from the context's pools
  etlg_batch* b;
  ~BatchGuard() { if (b) etlg_batch_free(b); }
};
int32_t build_side_inputs(etlg_ctx* c, etlg_batch* b, const std::vector<EpochRec>& eps);
int32_t setup_outputs(etlg_ctx* c, etlg_batch* b);
int32_t setup_scratch(etlg_ctx* c, DecParams& p, int set = 0);
bool plan_wanted(etlg_ctx* c, const etlg_batch* b);
int32_t enqueue_single(etlg_ctx* c, etlg_batch* b, int level);
int32_t standard_path(etlg_ctx* c, etlg_batch* b);
int32_t ctl_begin(etlg_ctx* c, etlg_batch* b, DecParams& p, hipStream_t s, bool ahead);
struct SlowScope {   // ETLG_HOST_TIMES=2: names any of the instrumented calls that takes more than a millisecond
  etlg_ctx* c; const char* what; std::chrono::steady_clock::time_point t0;
  SlowScope(etlg_ctx* c_, const char* w) : c(c_), what(w) { if (c && c->host_times_slow) t0 = std::chrono::steady_clock::now(); }
  ~SlowScope() {
    if (!c || !c->host_times_slow) return;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (us > 1000.0) fprintf(stderr, "etlg host times: %s took %.0f us\n", what, us);
  }
};
static inline void ht_start(etlg_ctx* c) { if (c->host_times) c->host_mark = std::chrono::steady_clock::now(); }
static inline void ht_mark(etlg_ctx* c, int i) {
  if (!c->host_times) return;
  const auto t = std::chrono::steady_clock::now();
  const double us = std::chrono::duration<double, std::micro>(t - c->host_mark).count();
  c->host_us[i] += us; c->host_n[i]++;
  if (c->host_times_slow && us > 500.0) fprintf(stderr, "etlg host times: segment %d took %.0f us\n", i, us);   // ETLG_HOST_TIMES=2: outliers as they happen
  c->host_mark = t;
}
int32_t decode_tail(etlg_ctx* c, etlg_batch* b, size_t nframes, bool async, etlg_batch* prev);
int32_t flush_deferred(etlg_ctx* c);
void check_invariants(etlg_ctx* c, const char* where);
struct InvariantScope { etlg_ctx* c; const char* where; ~InvariantScope() { if (c && c->debug_invariants) check_invariants(c, where); } };
hipError_t sync_decode_streams(etlg_ctx* c);

}  // namespace

// Live contexts: a batch may be freed after its context (garbage-collected bindings do that), in which case
// it must not touch the context's pools.
static std::mutex g_live_mu;
static std::map<const etlg_ctx*, uint64_t> g_live_ctx;  // context -> generation (an address can be reused by a later context)
static uint64_t g_ctx_gen = 0;
static bool ctx_alive(const etlg_ctx* c, uint64_t gen) { std::lock_guard<std::mutex> l(g_live_mu); auto it = g_live_ctx.find(c); return it != g_live_ctx.end() && it->second == gen; }

// ====================================================================== C API
// Pooled blocks of the hand-off calls: hipMalloc / hipFree synchronise the device and cost ~100 us each.
static hipError_t blk_take(etlg_ctx* c, size_t bytes, bool host, void** out, size_t* cap) {
  auto& pool = host ? c->blk_host : c->blk_dev;
  size_t best = (size_t)-1;
  for (size_t i = 0; i < pool.size(); i++)
    if (pool[i].second >= bytes && (best == (size_t)-1 || pool[i].second < pool[best].second)) best = i;
  if (best != (size_t)-1 && pool[best].second <= 4 * bytes + (1u << 20)) {
    *out = pool[best].first; *cap = pool[best].second;
    pool.erase(pool.begin() + (long)best);
    return hipSuccess;
  }
  const size_t want = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
  const hipError_t e = host ? hipHostMalloc(out, want, hipHostMallocDefault) : hipMalloc(out, want);
  if (e == hipSuccess) *cap = want;
  return e;
}
static void blk_give(etlg_ctx* c, uint64_t gen, void* p, size_t cap, bool host) {
  if (!p) return;
  if (c && ctx_alive(c, gen)) { (host ? c->blk_host : c->blk_dev).emplace_back(p, cap); return; }
  if (host) (void)hipHostFree(p); else (void)hipFree(p);
}
static void handoff_release(HandoffBlocks& m) {
  blk_give(m.ctx, m.ctx_gen, m.d_a, m.cap_a, false); blk_give(m.ctx, m.ctx_gen, m.d_b, m.cap_b, false); blk_give(m.ctx, m.ctx_gen, m.d_c, m.cap_c, false);
  blk_give(m.ctx, m.ctx_gen, m.h, m.cap_h, true);
  m.d_a = m.d_b = m.d_c = nullptr; m.h = nullptr;
}
struct ScratchBlk {  // a device block for the duration of one call
  etlg_ctx* c; void* p = nullptr; size_t cap = 0;
  ~ScratchBlk() { if (p) blk_give(c, c->gen, p, cap, false); }
};

// A context spreads its work over up to seven HIP streams; ROCm multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware
// queues (default 4), and two decode streams that share one queue serialise consecutive batches (etl_amd/__init__.py has the
// measurements). The runtime reads the variable when it initialises (first HIP call of the process): a host that loads this
// library before its first HIP call gets a workable default, an explicit setting of the caller's is respected.
__attribute__((constructor)) static void etlg_runtime_defaults(void) { (void)setenv("GPU_MAX_HW_QUEUES", "16", 0); }

extern "C" {

uint32_t etlg_abi_version(void) { return ETLG_ABI_VERSION; }

const etlg_err_desc* etlg_err_table(int32_t code) { return (code >= 0 && code < ETLG_E__COUNT) ? &kErrTable[code] : nullptr; }
int32_t etlg_type_class_of_oid(uint32_t oid) { return type_class(oid); }
int32_t etlg_array_elem_class(uint32_t oid) { for (const auto& a : kArrayOids) if (a.oid == oid) return a.elem; return ETLG_TC_STRING; }
uint32_t etlg_slot_bytes(int32_t cls) { return slot_bytes(cls); }

static char g_create_err[256] = "";
const char* etlg_create_error(void) { return g_create_err; }

int32_t etlg_ctx_create(int32_t hip_device, etlg_ctx** out) {
  if (!out) return ETLG_InvalidArgument;
  *out = nullptr;
  g_create_err[0] = 0;
  auto fail = [&](const char* what, hipError_t e) {
    snprintf(g_create_err, sizeof g_create_err, "%s: %s", what, hipGetErrorString(e));
    return (int32_t)ETLG_DeviceError;
  };
  hipError_t e = hipInit(0);
  if (e != hipSuccess) return fail("hipInit", e);
  int ndev = 0;
  e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess) return fail("hipGetDeviceCount", e);
  if (ndev <= 0 || hip_device < 0 || hip_device >= ndev) { snprintf(g_create_err, sizeof g_create_err, "device %d of %d not available", hip_device, ndev); return ETLG_DeviceError; }
  e = hipSetDevice(hip_device);
  if (e != hipSuccess) return fail("hipSetDevice", e);
  auto* c = new etlg_ctx();
  c->device = hip_device;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return fail("hipStreamCreateWithFlags", e); }
  c->own_stream = true;
  e = hipHostMalloc((void**)&c->h_init, sizeof(DevResult), hipHostMallocDefault);
  if (e != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return fail("hipHostMalloc", e); }
  { DevResult init{}; init.first_err = kNoErr; *c->h_init = init; }
  e = hipHostMalloc((void**)&c->h_init_ring, sizeof(DevResult) * etlg_ctx::kResRing, hipHostMallocDefault);
  if (e != hipSuccess) { snprintf(g_create_err, sizeof g_create_err, "hipHostMalloc: %s", hipGetErrorString(e)); delete c; return ETLG_DeviceError; }
  for (uint32_t i = 0; i < etlg_ctx::kResRing; i++) c->h_init_ring[i] = *c->h_init;
  e = hipMalloc((void**)&c->d_init_ring, sizeof(DevResult) * etlg_ctx::kResRing);
  if (e == hipSuccess) e = hipMemcpy(c->d_init_ring, c->h_init_ring, sizeof(DevResult) * etlg_ctx::kResRing, hipMemcpyHostToDevice);
  if (e != hipSuccess) { snprintf(g_create_err, sizeof g_create_err, "hipMalloc: %s", hipGetErrorString(e)); delete c; return ETLG_DeviceError; }
  // (a kernel whose static + dynamic LDS request exceeds a CU's 160 KB is refused here, not at its first launch)
  if (etlg_k_fused_set_lds() || etlg_k_cells_set_lds() || etlg_k_copy_set_lds() || etlg_k_rows_set_lds()) {
    snprintf(g_create_err, sizeof g_create_err, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) refused a kernel's LDS request");
    delete c; return ETLG_DeviceError;
  }
  { const char* fm = getenv("ETLG_FORCE_MULTIPASS"); c->force_multipass = fm && fm[0] == '1'; }
  { const char* e = getenv("ETLG_RING_H2D"); c->ring_h2d = e && e[0] == '1'; }
  { const char* e = getenv("ETLG_COPY_KERNEL"); c->copy_lane_per_byte = e && e[0] == '1'; }
  { const char* e = getenv("ETLG_COPY_DIRECT"); c->copy_direct = !(e && e[0] == '0'); }
  { const char* ht = getenv("ETLG_HOST_TIMES"); c->host_times = ht && (ht[0] == '1' || ht[0] == '2'); c->host_times_slow = ht && ht[0] == '2'; }
  { const char* sc = getenv("ETLG_CTRL_STAGE_CAP"); c->ctrl_stage_cap_test = sc ? (size_t)atol(sc) : 0; }
  { const char* rp = getenv("ETLG_RB_PARTS"); c->rb_parts_test = rp ? (uint32_t)atoi(rp) : 0; }   // tests: lanes per row in k_rb_rows (1-4) instead of the choice by row count
  { const char* fd = getenv("ETLG_FUSED_DBG"); c->fused_dbg = fd ? (uint32_t)atoi(fd) : 0; }
  { const char* fk = getenv("ETLG_FUSED_KERNEL"); c->fused_kernel = fk ? atoi(fk) : -1; }
  { const char* rm = getenv("ETLG_ROWS"); if (rm) c->rows_mode = atoi(rm); }
  { const char* cr = getenv("ETLG_CHAIN_REISSUE"); if (cr) c->chain_reissue = atoi(cr) != 0; }
  { const char* cs = getenv("ETLG_CHAIN_SPARE"); if (cs) c->chain_spare = atoi(cs) != 0; }
  { const char* pd = getenv("ETLG_PLAN_DELETES"); if (pd) c->plan_deletes = atoi(pd) != 0; }
  { const char* sc = getenv("ETLG_SCAN_CHAIN"); if (sc) c->scan_chain_mode = atoi(sc) != 0; }
  { const char* di = getenv("ETLG_DEBUG_INVARIANTS"); if (di) c->debug_invariants = atoi(di) != 0; }
  clear_error(c);
  (void)etlg_k_plan_set_lds();
  if (const char* pm = getenv("ETLG_PLAN")) c->plan_mode = atoi(pm);
  if (const char* pm = getenv("ETLG_PLAN_PRE")) c->plan_pre = atoi(pm);
  if (const char* pm = getenv("ETLG_CTL_HOLD")) c->ctl_hold_mode = atoi(pm) != 0;
  if (const char* pm = getenv("ETLG_PLAN_MARGIN")) c->plan_margin_pct = (uint32_t)atoi(pm);
  if (const char* pm = getenv("ETLG_PLAN_DBG")) c->plan_dbg = (uint32_t)atoi(pm);
  { const char* e = getenv("ETLG_CTL_OVERLAP"); c->ctl_overlap_mode = !(e && e[0] == '0'); }   // 0: batches on the control path stay on one decode stream
  if (const char* pm = getenv("ETLG_OVERLAP")) c->overlap_mode = atoi(pm);
  if (const char* pm = getenv("ETLG_CTL_ASYNC")) c->ctl_async_mode = atoi(pm);
  { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, hip_device) == hipSuccess && ncu > 0) c->n_cus = ncu; }
  { std::lock_guard<std::mutex> l(g_live_mu); c->gen = ++g_ctx_gen; g_live_ctx[c] = c->gen; }
  *out = c;
  return ETLG_OK;
}

void etlg_ctx_destroy(etlg_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)drain_pending(c);
  { std::lock_guard<std::mutex> l(g_live_mu); g_live_ctx.erase(c); }
  (void)sync_decode_streams(c);
  if (c->host_times) {
    static const char* names[8] = {"pre-pass kernels + result", "control list", "control bytes", "host control plane", "side inputs", "outputs", "enqueue decode", "wait for the batch"};
    for (int i = 0; i < 8; i++) if (c->host_n[i]) fprintf(stderr, "etlg host times: %-26s %8.1f us x %llu\n", names[i], c->host_us[i] / (double)c->host_n[i], (unsigned long long)c->host_n[i]);
  }
  if (c->h_scan) { (void)hipHostFree(c->h_scan); c->h_scan = nullptr; }
  for (DevBuf* b : {&c->d_copy_in, &c->d_copy_offs, &c->d_copy_out, &c->d_copy_out_offs, &c->d_scan, &c->d_in, &c->d_offs, &c->d_tag, &c->d_emit, &c->d_ffixed, &c->d_fheap, &c->d_blk32, &c->d_blk64, &c->d_ctrl, &c->d_ctrl_stage, &c->d_res, &c->d_desc, &c->d_colsel}) b->release();
  for (SideSet* ss : c->side_sets) { ss->dev.release(); if (ss->h) (void)hipHostFree(ss->h); if (ss->ready) (void)hipEventDestroy(ss->ready); delete ss; }
  for (OutSet* o : c->out_pool) { o->release(); delete o; }
  for (DevBuf* o : c->offs_pool) { o->release(); delete o; }
  for (auto& b : c->blk_dev) (void)hipFree(b.first);
  for (auto& b : c->blk_host) (void)hipHostFree(b.first);
  if (c->ctl_stream) (void)hipStreamDestroy(c->ctl_stream);
  if (c->mp_tail) (void)hipEventDestroy(c->mp_tail);
  if (c->h_cnt_init) (void)hipHostFree(c->h_cnt_init);
  if (c->h_hand) (void)hipHostFree(c->h_hand);
  if (c->h_ctl_list) (void)hipHostFree(c->h_ctl_list);
  if (c->h_ctl_stage) (void)hipHostFree(c->h_ctl_stage);
  if (c->ctl_alt.h_ctl_list) (void)hipHostFree(c->ctl_alt.h_ctl_list);
  if (c->ctl_alt.h_ctl_stage) (void)hipHostFree(c->ctl_alt.h_ctl_stage);
  for (DevBuf* b : {&c->ctl_alt.d_tag, &c->ctl_alt.d_emit, &c->ctl_alt.d_ffixed, &c->ctl_alt.d_fheap, &c->ctl_alt.d_blk32, &c->ctl_alt.d_blk64, &c->ctl_alt.d_ctrl, &c->ctl_alt.d_ctrl_stage}) b->release();
  c->d_ctl_res.release();
  if (c->scan_stream) (void)hipStreamDestroy(c->scan_stream);
  if (c->h2d_stream) (void)hipStreamDestroy(c->h2d_stream);
  if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
  if (c->res_stream) (void)hipStreamDestroy(c->res_stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  c->d_pre.release();
  if (c->tail2) (void)hipEventDestroy(c->tail2);
  if (c->fence_ev) (void)hipEventDestroy(c->fence_ev);
  for (auto& r : c->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (auto& pr : c->harena_pool) (void)hipHostFree(pr.first);
  if (c->h_init) (void)hipHostFree(c->h_init);
  if (c->h_poison) (void)hipHostFree(c->h_poison);
  if (c->h_init_ring) (void)hipHostFree(c->h_init_ring);
  if (c->d_init_ring) (void)hipFree(c->d_init_ring);
  for (DevResult* r : c->res_pool) (void)hipHostFree(r);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int32_t etlg_ctx_set_stream(etlg_ctx* c, void* s) {
  if (!c) return ETLG_InvalidArgument;
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  (void)drain_pending(c);
  (void)hipStreamSynchronize(c->stream);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  if (s) { c->stream = (hipStream_t)s; c->own_stream = false; }
  else { if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return ETLG_DeviceError; c->own_stream = true; }
  return ETLG_OK;
}

int32_t etlg_ctx_set_worker(etlg_ctx* c, int32_t kind, uint32_t table_id, uint64_t bootstrap) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c) return ETLG_InvalidArgument;
  (void)drain_pending(c);
  c->worker = kind; c->sync_table = table_id; c->bootstrap = bootstrap; c->side_valid = false; c->side_dirty = true;
  return ETLG_OK;
}

int32_t etlg_schema_put(etlg_ctx* c, uint32_t table_id, uint64_t snapshot, const char* nsp, const char* name, uint32_t ncols, const etlg_col* cols) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c || (ncols && !cols)) return ETLG_InvalidArgument;
  (void)drain_pending(c);
  auto s = std::make_shared<StoredSchema>();
  s->table_id = table_id; s->snapshot = snapshot; s->nsp = nsp ? nsp : ""; s->name = name ? name : "";
  for (uint32_t i = 0; i < ncols; i++) {
    StoredCol sc;
    sc.name = cols[i].name ? cols[i].name : ""; sc.type_oid = cols[i].type_oid; sc.typmod = cols[i].type_modifier;
    sc.attnum = cols[i].attnum; sc.nullable = cols[i].nullable != 0; sc.pk = cols[i].primary_key != 0;
    s->cols.push_back(std::move(sc));
  }
  c->cs.store[table_id][snapshot] = s;
  return ETLG_OK;
}

int32_t etlg_table_state(etlg_ctx* c, uint32_t table_id, int32_t kind, uint64_t lsn) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c) return ETLG_InvalidArgument;
  (void)drain_pending(c);
  if (kind == ETLG_TS_ABSENT) c->states.erase(table_id); else c->states[table_id] = TState{kind, lsn};
  c->side_dirty = true;
  return ETLG_OK;
}

int32_t etlg_table_ready(etlg_ctx* c, uint32_t table_id, uint64_t snapshot, const uint8_t* rmask, const uint8_t* imask, uint32_t n) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c || !rmask || !imask) return -ETLG_InvalidArgument;
  (void)drain_pending(c);
  SchemaPtr sch = get_at_or_before(c->cs, table_id, snapshot);
  if (!sch || sch->cols.size() != n) return -ETLG_MissingTableSchema;
  std::vector<uint8_t> r(rmask, rmask + n), i(imask, imask + n);
  const int32_t slot = make_slot(c, sch, r, i);
  c->cs.cache[table_id] = CacheEntry{2, sch->snapshot, slot};
  c->side_dirty = true;
  return slot;
}

int32_t etlg_table_forget(etlg_ctx* c, uint32_t table_id) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c) return ETLG_InvalidArgument;
  (void)drain_pending(c);
  c->cs.cache.erase(table_id);
  c->side_dirty = true;
  return ETLG_OK;
}

int32_t etlg_table_cache_get(const etlg_ctx* c, uint32_t table_id, int32_t* kind, uint64_t* snapshot, int32_t* slot) {
  if (!c || !kind || !snapshot || !slot) return 0;
  auto it = c->cs.cache.find(table_id);
  if (it == c->cs.cache.end()) return 0;
  *kind = (int32_t)it->second.kind; *snapshot = it->second.snapshot; *slot = it->second.slot;
  return 1;
}

int32_t etlg_ctx_reset_stream_state(etlg_ctx* c) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c) return ETLG_InvalidArgument;
  (void)drain_pending(c);
  c->in_txn = false; c->final_lsn = 0; c->next_ord = 0;
  return ETLG_OK;
}

const etlg_error* etlg_last_error(const etlg_ctx* c) { return c ? &c->err : nullptr; }

int32_t etlg_ctx_slots(const etlg_ctx* c, uint32_t* n, const etlg_slot_desc** slots) {
  if (!c || !n || !slots) return ETLG_InvalidArgument;
  static thread_local std::vector<etlg_slot_desc> tmp;
  tmp.clear();
  for (auto& s : c->slots) tmp.push_back(s->desc);
  *n = (uint32_t)tmp.size(); *slots = tmp.data();
  return ETLG_OK;
}

// debugging aid (not part of etlg.h): per-phase cycle sums of the last finished batch
int32_t etlg_ctx_debug_times(etlg_ctx* c, unsigned long long* out12) {
  if (!c || !out12) return ETLG_InvalidArgument;
  for (int i = 0; i < 12; i++) out12[i] = c->last_dbg[i];
  return ETLG_OK;
}

// debugging aid (not part of etlg.h): batches finished per path
//   [0] k_fused  [1] k_cells  [2] multi-pass directly  [3] single-pass result discarded and redone by the multi-pass kernels
int32_t etlg_ctx_debug_paths(etlg_ctx* c, unsigned long long* out4) {
  if (!c || !out4) return ETLG_InvalidArgument;
  for (int i = 0; i < 4; i++) out4[i] = c->path_n[i];
  return ETLG_OK;
}
//   [4] k_plan  [5] plan result discarded and redone by the generic single-pass kernel  [6] batches that took the control path
//   (a Relation / DDL frame, or a caller without ETLG_F_NO_CONTROL on the multi-pass path)  [7] ASYNC batches re-run because their predecessor failed
// debugging aid (not part of etlg.h): ASYNC batches that were enqueued beside their predecessor on the second decode stream
unsigned long long etlg_ctx_debug_overlapped(etlg_ctx* c) { return c ? c->overlapped : 0; }
unsigned long long etlg_ctx_debug_ctl_ahead(etlg_ctx* c) { return c ? c->ctl_ahead_n : 0; }
unsigned long long etlg_ctx_debug_staged(etlg_ctx* c) { return c ? c->staged_async : 0; }   // ASYNC batches whose host input was staged on the copy stream

int32_t etlg_host_alloc(etlg_ctx* c, size_t bytes, void** out) {
  if (!c || !out || !bytes) return ETLG_InvalidArgument;
  *out = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return ETLG_OK;
}
void etlg_host_free(void* p) { if (p) (void)hipHostFree(p); }   // batches whose control pre-pass ran ahead of their decode
// debugging aid (not part of etlg.h): [0] batches k_rows produced, [1] batches it handed back to k_cells / k_fused
int32_t etlg_ctx_debug_rows(etlg_ctx* c, unsigned long long* out2) {   // (three words)
  if (!c || !out2) return ETLG_InvalidArgument;
  out2[0] = c->rows_n; out2[1] = c->rows_redone; out2[2] = c->rows_resized;
  return ETLG_OK;
}
int32_t etlg_ctx_debug_paths8(etlg_ctx* c, unsigned long long* out8) {
  if (!c || !out8) return ETLG_InvalidArgument;
  for (int i = 0; i < 8; i++) out8[i] = c->path_n[i];
  return ETLG_OK;
}

int32_t etlg_ctx_profile(etlg_ctx* c, int32_t enable) {
  if (!c) return ETLG_InvalidArgument;
  c->prof = enable != 0;
  c->prof_serial = enable == 2;   // 2: time kernels one at a time (ASYNC batches stay on one stream), for per-kernel durations that do not overlap
  if (!enable) { for (int i = 0; i < kProfSlots; i++) { c->prof_ms[i] = 0; c->prof_n[i] = 0; } }
  return ETLG_OK;
}

int32_t etlg_ctx_profile_read(etlg_ctx* c, etlg_kernel_stat* out, uint32_t cap, uint32_t* n) {
  if (!c || !n) return ETLG_InvalidArgument;
  (void)hipStreamSynchronize(c->stream);
  for (auto& r : c->prof_recs) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->prof_ms[r.which] += ms; c->prof_n[r.which]++; }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  c->prof_recs.clear();
  uint32_t k = 0;
  for (int i = 0; i < kProfSlots && k < cap; i++) { out[k].name = i == kPlan ? "k_plan" : i == kPlanPre ? "k_plan_pre" : i == kFused ? "k_fused" : i == kCells ? "k_cells" : i == kBounds ? "k_bounds" : i == kCopy ? "k_copy_frames" : i == kCopyCells ? "k_copy_cells" : i == kRows ? "k_rows" : etlg_k_name(i); out[k].launches = c->prof_n[i]; out[k].total_ms = c->prof_ms[i]; k++; }
  *n = k;
  return ETLG_OK;
}

// Record boundaries on the device (scan.hip): fills c->d_offs with nframes + 1 offsets of the frames
// of `d_in[0, len)` and returns nframes. Optimistic kernel + hint reruns + one-lane fallback.
// One record-boundary scan in flight (scan.hip). scan_launch enqueues a run and returns; scan_collect waits for it, reruns it
// with hints / on one lane when tiles guessed wrong, and returns the frame count.
hipError_t scan_launch(etlg_ctx* c, ScanJob& j, bool sequential) {
  const size_t len = j.len;
  hipStream_t s = j.s;
  const size_t tb = etlg_k_bounds_tile_bytes();
  const size_t ntiles = (len + tb - 1) / tb;
  // scratch: result block | tile summaries | base indexes | the tiles' scratch rows, and the hints behind them. A run writes everything it
  // reads (no descriptor to clear, no memset on the stream); the result travels through pinned host memory.
  if (!c->h_scan) { hipError_t e = hipHostMalloc((void**)&c->h_scan, 64); if (e != hipSuccess) return e; }
  if (ntiles > c->scan_tiles_cap) {  // (re)allocation: the layout is by capacity; only the hints need initialising
    const size_t tcap = ntiles + ntiles / 4 + 64;
    const size_t body = (etlg_k_bounds_scratch_bytes(tcap) + 63) & ~(size_t)63;
    const size_t need = body + ((tcap * 4 + 63) & ~(size_t)63) + 64;
    hipError_t e = c->d_scan.ensure(need); if (e != hipSuccess) return e;
    e = hipMemsetAsync((uint8_t*)c->d_scan.p + body, 0xFF, tcap * 4, s); if (e != hipSuccess) return e;
    c->scan_tiles_cap = tcap; c->scan_half = body;
  }
  uint8_t* base = (uint8_t*)c->d_scan.p;
  hipError_t e = j.offs->ensure((j.cap + 2) * 4); if (e != hipSuccess) return e;
  j.cur = j.res ? (uint8_t*)j.res : base;
  c->h_scan[0] = 0; c->h_scan[1] = 0;
  ProfRec r; r.which = kBounds;
  if (c->prof) { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); (void)hipEventRecord(r.a, s); }
  etlg_k_launch_bounds(j.d_in, len, (uint32_t*)j.offs->p, (uint32_t)std::min<size_t>(j.cap + 2, 0xFFFFFFFFu), base, j.res, (uint32_t*)(base + c->scan_half), c->h_scan,
                       sequential ? 1 : 0, s);
  if (c->prof) { (void)hipEventRecord(r.b, s); c->prof_recs.push_back(r); }
  return hipSuccess;
}

hipError_t scan_begin(etlg_ctx* c, ScanJob& j, const uint8_t* d_in, size_t len, hipStream_t s, DevBuf& offs, uint32_t* res = nullptr) {
  j = ScanJob{};
  j.d_in = d_in; j.len = len; j.s = s; j.offs = &offs; j.res = res;
  j.cap = len / 24 + 1024;  // frames the offsets buffer can take; grown to the worst case (5-byte frames) on demand
  if (len == 0) {
    hipError_t e = offs.ensure(64); if (e != hipSuccess) return e;
    return hipMemsetAsync(offs.p, 0, 4, s);
  }
  return scan_launch(c, j, false);
}

hipError_t scan_collect(etlg_ctx* c, ScanJob& j, size_t* nframes_out) {
  *nframes_out = 0;
  if (j.len == 0) return hipSuccess;
  const size_t len = j.len;
  hipStream_t s = j.s;
  const size_t tb = etlg_k_bounds_tile_bytes();
  const size_t ntiles = (len + tb - 1) / tb;
  bool used_hints = false, hints_set = false, overflow = false;
  hipError_t rc = hipSuccess;
  for (int run = 0;; run++) {
    const bool sequential = run >= 4;
    hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) return e;
    uint32_t nf = c->h_scan[0], flags = 0, nbad = 0;
    if (c->h_scan[1]) {  // the run did not hold: the details are in the device result block
      uint32_t res[16];
      e = hipMemcpy(res, j.cur, 64, hipMemcpyDeviceToHost); if (e != hipSuccess) return e;
      nf = res[0]; flags = res[1]; nbad = res[2];
      if (nbad) hints_set = true;
      if (flags & 4u) overflow = true;   // a tile with more frames than its scratch row holds (malformed input): hints do not help, one lane does
    }
    bool again_same = false;
    if (flags & 2u) {  // offsets buffer too small
      if (j.cap >= len / 5 + 2) { rc = hipErrorOutOfMemory; break; }
      j.cap = len / 5 + 2;
      run--; again_same = true;
    } else if (sequential || (!(flags & 4u) && nbad == 0)) {
      if (used_hints) c->scan_reruns++;
      if (sequential) c->scan_seq++;
      *nframes_out = nf;
      break;
    } else {
      used_hints = true;  // some tiles guessed wrong (their hints are set now), or a spin gave up: run again
    }
    if (overflow && !again_same) run = 3;   // (the next run is the one-lane one)
    e = scan_launch(c, j, again_same ? sequential : run + 1 >= 4); if (e != hipSuccess) return e;
  }
  if (hints_set) { const hipError_t e = hipMemsetAsync((uint8_t*)c->d_scan.p + c->scan_half, 0xFF, ntiles * 4, s); if (e != hipSuccess) return e; }   // rare: leave the hints clean for the next scan
  return rc;
}

// Record boundaries on the device: fills `offs` with nframes + 1 offsets of the frames of `d_in[0, len)` and returns nframes.
hipError_t device_scan(etlg_ctx* c, const uint8_t* d_in, size_t len, size_t* nframes_out, hipStream_t s, DevBuf& offs) {
  ScanJob j;
  hipError_t e = scan_begin(c, j, d_in, len, s, offs); if (e != hipSuccess) return e;
  return scan_collect(c, j, nframes_out);
}


int32_t etlg_scan_boundaries(etlg_ctx* c, const uint8_t* buf, size_t len, uint32_t flags, uint32_t* offsets_out, size_t cap,
                             size_t* nframes_out) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c || !nframes_out || (!offsets_out && cap)) return ETLG_InvalidArgument;
  clear_error(c);
  if (len > 0xFFFFFFFFull - 16) return lib_error(c, ETLG_InvalidArgument, "batch too large (max 4 GiB, 2^30 frames)");
  HIPCHK(c, hipSetDevice(c->device));
  const bool in_dev = flags & ETLG_F_INPUT_ON_DEVICE, out_dev = flags & ETLG_F_OUTPUT_ON_DEVICE;
  const uint8_t* d_in_ptr = buf;
  if (!in_dev) {
    HIPCHK(c, c->d_in.ensure(len + 64));
    if (len) HIPCHK(c, hipMemcpyAsync(c->d_in.p, buf, len, hipMemcpyHostToDevice, c->stream));
    d_in_ptr = (const uint8_t*)c->d_in.p;
  }
  size_t nf = 0;
  HIPCHK(c, device_scan(c, d_in_ptr, len, &nf, c->stream, c->d_offs));
  *nframes_out = nf;
  if (nf + 1 > cap) return lib_error(c, ETLG_InvalidArgument, "offsets_out too small for nframes + 1 entries");
  HIPCHK(c, hipMemcpyAsync(offsets_out, c->d_offs.p, (nf + 1) * 4, out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ETLG_OK;
}

int32_t etlg_frame_tags(etlg_ctx* c, const uint8_t* buf, size_t len, const uint32_t* frame_offsets, size_t nframes, uint32_t flags,
                        uint8_t* tags_out) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c || !frame_offsets || (!tags_out && nframes)) return ETLG_InvalidArgument;
  clear_error(c);
  if (len > 0xFFFFFFFFull - 16 || nframes >= (1u << 30)) return lib_error(c, ETLG_InvalidArgument, "batch too large (max 4 GiB, 2^30 frames)");
  if (!nframes) return ETLG_OK;
  HIPCHK(c, hipSetDevice(c->device));
  { const int32_t rc = drain_pending(c); if (rc != ETLG_OK) return rc; }
  const bool in_dev = flags & ETLG_F_INPUT_ON_DEVICE, out_dev = flags & ETLG_F_OUTPUT_ON_DEVICE;
  hipStream_t s = c->stream;
  DecParams p{};
  p.in = buf; p.offs = frame_offsets;
  if (!in_dev) {
    HIPCHK(c, c->d_in.ensure(len + 64)); HIPCHK(c, c->d_offs.ensure((nframes + 1) * 4));
    if (len) HIPCHK(c, hipMemcpyAsync(c->d_in.p, buf, len, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_offs.p, frame_offsets, (nframes + 1) * 4, hipMemcpyHostToDevice, s));
    p.in = (const uint8_t*)c->d_in.p; p.offs = (const uint32_t*)c->d_offs.p;
  }
  p.nframes = (uint32_t)nframes; p.nblocks = (p.nframes + kBlock - 1) / kBlock; p.in_len = len;
  { const int32_t rc = setup_scratch(c, p); if (rc != ETLG_OK) return rc; }
  launch(c, 0, p);   // k_classify: envelope + tag of every frame
  HIPCHK(c, hipMemcpyAsync(tags_out, p.f_tag, nframes, out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  return ETLG_OK;
}

int32_t etlg_shard_plan(etlg_ctx* c, const uint8_t* buf, size_t len, const uint32_t* frame_offsets, size_t nframes, uint32_t n_shards,
                        uint32_t flags, uint64_t* cuts_out) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c || !frame_offsets || !cuts_out || !n_shards || n_shards > 4096) return ETLG_InvalidArgument;
  clear_error(c);
  if (len > 0xFFFFFFFFull - 16 || nframes >= (1u << 30)) return lib_error(c, ETLG_InvalidArgument, "batch too large (max 4 GiB, 2^30 frames)");
  cuts_out[0] = 0;
  for (uint32_t k = 1; k <= n_shards; k++) cuts_out[k] = nframes;
  if (!nframes) { for (uint32_t k = 1; k < n_shards; k++) cuts_out[k] = 0; return ETLG_OK; }
  if (n_shards == 1) return ETLG_OK;
  HIPCHK(c, hipSetDevice(c->device));
  { const int32_t rc = drain_pending(c); if (rc != ETLG_OK) return rc; }
  const bool in_dev = flags & ETLG_F_INPUT_ON_DEVICE;
  hipStream_t s = c->stream;
  DecParams p{};
  p.in = buf; p.offs = frame_offsets;
  if (!in_dev) {
    HIPCHK(c, c->d_in.ensure(len + 64)); HIPCHK(c, c->d_offs.ensure((nframes + 1) * 4));
    if (len) HIPCHK(c, hipMemcpyAsync(c->d_in.p, buf, len, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_offs.p, frame_offsets, (nframes + 1) * 4, hipMemcpyHostToDevice, s));
    p.in = (const uint8_t*)c->d_in.p; p.offs = (const uint32_t*)c->d_offs.p;
  }
  p.nframes = (uint32_t)nframes; p.nblocks = (p.nframes + kBlock - 1) / kBlock; p.in_len = len;
  { const int32_t rc = setup_scratch(c, p); if (rc != ETLG_OK) return rc; }
  launch(c, 0, p);   // k_classify: the tags stay on the device
  HIPCHK(c, c->d_colsel.ensure((size_t)n_shards * 4 + 64));
  uint32_t* d_cuts = (uint32_t*)c->d_colsel.p;
  etlg_k_shard_cuts(p.f_tag, p.offs, p.nframes, n_shards, d_cuts, s);
  std::vector<uint32_t> h(n_shards - 1);
  HIPCHK(c, hipMemcpyAsync(h.data(), d_cuts, (size_t)(n_shards - 1) * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  for (uint32_t k = 1; k < n_shards; k++) cuts_out[k] = std::max<uint64_t>(h[k - 1], cuts_out[k - 1]);   // (cuts never go back: a shard may be empty)
  return ETLG_OK;
}

int32_t etlg_shard_replay(etlg_ctx* c, const uint8_t* buf, size_t len, const uint32_t* frame_offsets, size_t nframes) {
  if (!c) return ETLG_InvalidArgument;
  int32_t rc = ETLG_OK;
  if (nframes) {
    if (!buf || !frame_offsets) return ETLG_InvalidArgument;
    etlg_batch* b = nullptr;
    rc = etlg_decode(c, buf, len, frame_offsets, nframes, ETLG_F_OUTPUT_ON_DEVICE, &b);   // the control path: default flags, synchronous; the events are dropped
    const etlg_error saved = c->err; const std::string detail = c->err_detail;
    if (b) etlg_batch_free(b);
    if (rc != ETLG_OK) { c->err = saved; c->err_detail = detail; c->err.detail = c->err_detail.empty() ? nullptr : c->err_detail.c_str(); return rc; }
  }
  return etlg_ctx_reset_stream_state(c);   // the shard behind starts outside any transaction, at ordinal 0 (its first frame follows a Commit)
}

int32_t etlg_control_stream(etlg_ctx* c, const uint8_t* buf, size_t len, const uint32_t* frame_offsets, size_t nframes, uint32_t flags,
                            uint8_t* out_bytes, size_t out_cap, uint32_t* out_offsets, size_t out_offsets_cap,
                            size_t* n_bytes, size_t* n_frames, uint32_t* last_tag) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c || !frame_offsets || !n_bytes || !n_frames) return ETLG_InvalidArgument;
  clear_error(c);
  *n_bytes = 0; *n_frames = 0;
  if (last_tag) *last_tag = 0;
  if (len > 0xFFFFFFFFull - 16 || nframes >= (1u << 30)) return lib_error(c, ETLG_InvalidArgument, "batch too large (max 4 GiB, 2^30 frames)");
  if (!nframes) return ETLG_OK;
  HIPCHK(c, hipSetDevice(c->device));
  { const int32_t rc = drain_pending(c); if (rc != ETLG_OK) return rc; }
  const bool in_dev = flags & ETLG_F_INPUT_ON_DEVICE;
  hipStream_t s = c->stream;
  DecParams p{};
  p.in = buf; p.offs = frame_offsets;
  if (!in_dev) {
    HIPCHK(c, c->d_in.ensure(len + 64)); HIPCHK(c, c->d_offs.ensure((nframes + 1) * 4));
    if (len) HIPCHK(c, hipMemcpyAsync(c->d_in.p, buf, len, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_offs.p, frame_offsets, (nframes + 1) * 4, hipMemcpyHostToDevice, s));
    p.in = (const uint8_t*)c->d_in.p; p.offs = (const uint32_t*)c->d_offs.p;
  }
  p.nframes = (uint32_t)nframes; p.nblocks = (p.nframes + kBlock - 1) / kBlock; p.in_len = len;
  { const int32_t rc = setup_scratch(c, p); if (rc != ETLG_OK) return rc; }
  launch(c, 0, p);   // k_classify: envelope + tag of every frame
  constexpr uint32_t kCap = 1u << 16;   // control frames of one range the list holds
  // scratch: hdr (16 words) | list (kCap) | span (2 per frame) | keep frames (<= 3 per frame: the frame, its Begin, its Commit) |
  // lens (3 per frame) | out offsets (3 per frame + 1): 16 + 12 kCap + 1 words
  ScratchBlk blk{c};
  constexpr size_t kCtlWords = 16 + 12 * (size_t)kCap + 1;
  HIPCHK(c, blk_take(c, kCtlWords * 4 + 256, false, &blk.p, &blk.cap));
  uint32_t* d_hdr = (uint32_t*)blk.p;
  uint32_t* d_list = d_hdr + 16; uint32_t* d_span = d_list + kCap; uint32_t* d_keep = d_span + 2 * kCap;
  uint32_t* d_lens = d_keep + 3 * kCap; uint32_t* d_oo = d_lens + 3 * kCap;
  HIPCHK(c, hipMemsetAsync(d_hdr, 0, 8, s));
  etlg_k_ctl_pick(p.f_tag, p.nframes, d_hdr, d_list, kCap, s);
  uint32_t hdr[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(hdr, d_hdr, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  if (last_tag) *last_tag = hdr[1];
  const uint32_t n = hdr[0];
  if (!n) { if (out_offsets && out_offsets_cap) out_offsets[0] = 0; return ETLG_OK; }   // the common case: one kernel pair, one 8-byte copy
  if (n > kCap) return lib_error(c, ETLG_Unsupported, "more than 65536 control frames in one range: extract the control stream of smaller ranges");
  std::vector<uint32_t> list(n), span(2 * (size_t)n);
  HIPCHK(c, hipMemcpy(list.data(), d_list, (size_t)n * 4, hipMemcpyDeviceToHost));
  std::sort(list.begin(), list.end());   // the pick is unordered (atomics); the span kernel does not care, the stream does
  HIPCHK(c, hipMemcpyAsync(d_list, list.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
  etlg_k_ctl_span(p.f_tag, p.nframes, d_list, n, d_span, s);
  HIPCHK(c, hipMemcpyAsync(span.data(), d_span, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  std::vector<uint32_t> keep(list);
  for (uint32_t i = 0; i < n; i++) {
    if (span[2 * i] == 0xFFFFFFFFu) continue;        // not inside a transaction of this range: the frame travels alone
    keep.push_back(span[2 * i]);
    if (span[2 * i + 1] != 0xFFFFFFFFu) keep.push_back(span[2 * i + 1]);
  }
  std::sort(keep.begin(), keep.end());
  keep.erase(std::unique(keep.begin(), keep.end()), keep.end());
  const uint32_t nk = (uint32_t)keep.size();   // <= 3 n
  if ((size_t)nk > 3 * (size_t)kCap) return lib_error(c, ETLG_Unsupported, "control stream of this range holds more frames than its scratch block");
  std::vector<uint32_t> lens(nk), oo((size_t)nk + 1, 0);
  HIPCHK(c, hipMemcpyAsync(d_keep, keep.data(), (size_t)nk * 4, hipMemcpyHostToDevice, s));
  etlg_k_ctl_gather(p.in, p.offs, d_keep, nk, d_lens, nullptr, nullptr, s);
  HIPCHK(c, hipMemcpyAsync(lens.data(), d_lens, (size_t)nk * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  for (uint32_t i = 0; i < nk; i++) oo[i + 1] = oo[i] + lens[i];
  *n_bytes = oo[nk]; *n_frames = nk;
  if (!out_bytes || !out_offsets || out_cap < oo[nk] || out_offsets_cap < (size_t)nk + 1)
    return lib_error(c, ETLG_InvalidArgument, "control stream does not fit the output buffers (n_bytes / n_frames hold what it needs)");
  ScratchBlk stage{c};
  HIPCHK(c, blk_take(c, (size_t)oo[nk] + 64, false, &stage.p, &stage.cap));
  HIPCHK(c, hipMemcpyAsync(d_oo, oo.data(), (size_t)nk * 4, hipMemcpyHostToDevice, s));
  etlg_k_ctl_gather(p.in, p.offs, d_keep, nk, d_lens, d_oo, (uint8_t*)stage.p, s);
  HIPCHK(c, hipMemcpyAsync(out_bytes, stage.p, oo[nk], hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  memcpy(out_offsets, oo.data(), ((size_t)nk + 1) * 4);
  return ETLG_OK;
}

// debugging aid (not part of etlg.h): [0] scans that needed a rerun with hints, [1] scans that fell back to the one-lane walk
int32_t etlg_ctx_debug_scan(etlg_ctx* c, unsigned long long* out2) {
  if (!c || !out2) return ETLG_InvalidArgument;
  out2[0] = c->scan_reruns; out2[1] = c->scan_seq;
  return ETLG_OK;
}

// debugging aid (not part of etlg.h): table-copy batches [0] produced by the rows -> arena kernel, [1] decoded through the row -> frame rewrite
int32_t etlg_ctx_debug_copy(etlg_ctx* c, unsigned long long* out2) {
  if (!c || !out2) return ETLG_InvalidArgument;
  out2[0] = c->copy_n[0]; out2[1] = c->copy_n[1];
  return ETLG_OK;
}

// debugging aid (not part of etlg.h): [0] result blocks cleared again after a second attempt behind their lap's re-initialisation,
// [1] chains finished early because their last batch was marked for a second attempt
int32_t etlg_ctx_debug_ring(etlg_ctx* c, unsigned long long* out1) {
  if (!c || !out1) return ETLG_InvalidArgument;
  out1[0] = c->ring_recleared; out1[1] = c->chain_healed; out1[2] = c->chain_reissued; out1[3] = c->scan_chained_n; out1[4] = c->scan_chain_redone; out1[5] = c->chain_spared;   // (callers pass eight words)
  return ETLG_OK;
}

int32_t etlg_copy_decode(etlg_ctx* c, int32_t schema_slot, const uint8_t* buf, size_t len, const uint32_t* row_offsets, size_t nrows,
                         uint32_t flags, etlg_batch** out) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c || !out || !row_offsets) return ETLG_InvalidArgument;
  *out = nullptr;
  clear_error(c);
  if (schema_slot < 0 || (size_t)schema_slot >= c->slots.size()) return lib_error(c, ETLG_InvalidArgument, "unknown schema slot");
  const SlotHost& sh = *c->slots[(size_t)schema_slot];
  const uint32_t ncols = sh.desc.n_cols;
  const uint64_t per_row = etlg_k_copy_bytes_per_row(ncols);
  const uint64_t syn_len = (uint64_t)len + (uint64_t)nrows * per_row;
  if (syn_len > 0xFFFFFFFFull - 64 || nrows >= (1u << 30)) return lib_error(c, ETLG_InvalidArgument, "batch too large (max 4 GiB, 2^30 frames)");
  HIPCHK(c, hipSetDevice(c->device));
  const bool in_dev = flags & ETLG_F_INPUT_ON_DEVICE;
  hipStream_t s = c->stream;
  CopyJob j;
  j.active = true; j.slot = schema_slot; j.nrows = (uint32_t)nrows; j.ncols = ncols; j.rel_id = sh.desc.table_id; j.rows_len = len;
  j.syn_len = syn_len;
  // first attempt: the rows go straight into the arena (k_copy_cells, cells.hip); a batch with anything unusual in it — a malformed
  // row, rows that do not fit a tile's window — fails there and is decoded again through the row -> frame rewrite below
  j.direct = c->copy_direct && nrows != 0 && len < (1ull << 31) && !c->force_multipass && ncols <= etlg_k_cells_maxc();
  // ASYNC (the reference's caller streams rows continuously, postgres/stream/table_copy.rs:78-99): the batch is enqueued behind the
  // table-copy batches already in flight — they share nothing but the stream: every batch decodes inside a virtual transaction of its
  // own — and etlg_batch_sync finishes it. Only the rows -> arena kernel is enqueued that way; what it leaves to the frame rewrite
  // (the context's shared buffers) is done when the batch is synced.
  j.async = (flags & ETLG_F_ASYNC) && (flags & ETLG_F_OUTPUT_ON_DEVICE) && j.direct;
  // Batches of the OTHER kind still in flight finish first: finish_batch writes the stream state a WAL batch leaves into the context,
  // and the rows below must neither see that state nor be chained to it. (A synchronous call finishes everything, as before.)
  bool drain = !j.async;
  for (const etlg_batch* pb : c->pending) if (!pb->copy.active) drain = true;
  if (drain) { const int32_t rc = drain_pending(c); if (rc != ETLG_OK) return rc; clear_error(c); }
  // (a staging block / event `j` still holds when the call returns — an early return, or a decode that failed before there was a batch
  // to adopt them — goes back to the pools)
  struct StageGuard { etlg_ctx* c; CopyJob& j; ~StageGuard() { if (j.stage_blk) blk_give(c, c->gen, j.stage_blk, j.stage_cap, false); if (j.h2d_done) c->ev_pool.push_back(j.h2d_done); } } stage_guard{c, j};
  if (in_dev) { j.d_rows = buf; j.d_row_offs = row_offsets; }
  else if (j.async) {
    // host rows: uploaded into a device block the batch owns, on the copy stream, beside the decode of the batch before it (the caller
    // keeps buf / row_offsets untouched until the batch is synced, as for every ASYNC batch: include/etlg.h)
    const size_t o_offs = (len + 16 + 255) & ~(size_t)255;
    HIPCHK(c, blk_take(c, o_offs + (nrows + 1) * 4 + 64, false, &j.stage_blk, &j.stage_cap));
    if (!c->h2d_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking));
    if (c->ev_pool.empty()) { hipEvent_t e = nullptr; HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_pool.push_back(e); }
    j.h2d_done = c->ev_pool.back(); c->ev_pool.pop_back();
    if (len) HIPCHK(c, hipMemcpyAsync(j.stage_blk, buf, len, hipMemcpyHostToDevice, c->h2d_stream));
    HIPCHK(c, hipMemcpyAsync((uint8_t*)j.stage_blk + o_offs, row_offsets, (nrows + 1) * 4, hipMemcpyHostToDevice, c->h2d_stream));
    HIPCHK(c, hipEventRecord(j.h2d_done, c->h2d_stream));
    HIPCHK(c, hipStreamWaitEvent(s, j.h2d_done, 0));
    j.d_rows = (const uint8_t*)j.stage_blk; j.d_row_offs = (const uint32_t*)((const uint8_t*)j.stage_blk + o_offs);
    c->staged_async++;
  } else {
    HIPCHK(c, c->d_copy_in.ensure(len + 64)); HIPCHK(c, c->d_copy_offs.ensure((nrows + 1) * 4));
    if (len) HIPCHK(c, hipMemcpyAsync(c->d_copy_in.p, buf, len, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_copy_offs.p, row_offsets, (nrows + 1) * 4, hipMemcpyHostToDevice, s));
    j.d_rows = (const uint8_t*)c->d_copy_in.p; j.d_row_offs = (const uint32_t*)c->d_copy_offs.p;
  }
  if (!j.direct) {
    HIPCHK(c, c->d_copy_out.ensure(syn_len + 64)); HIPCHK(c, c->d_copy_out_offs.ensure((nrows + 1) * 4));
    j.d_out = (uint8_t*)c->d_copy_out.p; j.d_out_offs = (uint32_t*)c->d_copy_out_offs.p;
    if (nrows == 0) HIPCHK(c, hipMemsetAsync(j.d_out_offs, 0, 4, s));
  }
  const uint64_t avg = nrows ? (len + nrows - 1) / nrows : 0;
  j.lds = (uint32_t)std::min<uint64_t>(((256 * avg * 9 / 8 + 1024) + 255) & ~255ull, 150 * 1024);
  { const char* e = getenv("ETLG_COPY_LDS"); if (e) j.lds = (uint32_t)atoi(e); }   // measurement knob: 0 = rows read in place (no staging window, more waves per CU)
  // the rows decode inside a virtual transaction of their own; the context's stream state is left alone (finish_batch keeps it
  // that way for a batch that is finished later)
  const bool sv_in = c->in_txn; const uint64_t sv_lsn = c->final_lsn, sv_ord = c->next_ord;
  c->in_txn = true; c->final_lsn = 0; c->next_ord = 0;
  c->copy = j;
  const uint32_t dflags = (flags & (ETLG_F_OUTPUT_ON_DEVICE | ETLG_F_FINISH_CELLS)) | ETLG_F_INPUT_ON_DEVICE | ETLG_F_NO_CONTROL | (j.async ? (uint32_t)ETLG_F_ASYNC : 0u);
  const int32_t rc = j.direct ? etlg_decode(c, j.d_rows, len, j.d_row_offs, nrows, dflags, out)
                              : etlg_decode(c, j.d_out, (size_t)syn_len, j.d_out_offs, nrows, dflags, out);
  j.stage_blk = c->copy.stage_blk; j.h2d_done = c->copy.h2d_done;   // (nullptr once the batch has adopted them: etlg_decode)
  c->copy = CopyJob{};
  c->in_txn = sv_in; c->final_lsn = sv_lsn; c->next_ord = sv_ord;
  return rc;
}

int32_t etlg_decode(etlg_ctx* c, const uint8_t* buf, size_t len, const uint32_t* frame_offsets, size_t nframes, uint32_t flags, etlg_batch** out) {
  if (!c || !out) return ETLG_InvalidArgument;
  InvariantScope inv_scope{c, "etlg_decode"};
  SlowScope slow_scope_decode(c, "etlg_decode");
  *out = nullptr;
  clear_error(c);
  if (len > 0xFFFFFFFFull - 16 || nframes >= (1u << 30)) return lib_error(c, ETLG_InvalidArgument, "batch too large (max 4 GiB, 2^30 frames)");
  HIPCHK(c, hipSetDevice(c->device));
  // (the batch a call left deferred — a boundary scan in flight, a control pre-pass running ahead — is flushed below, once this call
  // knows whether it is itself a batch of the pipelined control path: such a batch sends its own pre-pass out FIRST)
  bool in_dev = flags & ETLG_F_INPUT_ON_DEVICE;
  const bool out_dev = flags & ETLG_F_OUTPUT_ON_DEVICE;
  const bool no_ctrl = flags & ETLG_F_NO_CONTROL;
  const bool scan = frame_offsets == nullptr;
  // ASYNC table-copy batches still in flight (etlg_copy_decode) finish before a WAL batch is taken: a chained batch would read the state
  // the last of them leaves — the end of a virtual transaction — as its carried transaction state
  if (!c->copy.active) for (const etlg_batch* pb : c->pending) if (pb->copy.active) { const int32_t rc_ = flush_deferred(c); (void)rc_; const int32_t rc = drain_pending(c); if (rc != ETLG_OK) return rc; clear_error(c); break; }
  // Can this call turn out to be one that HOLDS the deferred batch (sends its own control pre-pass out first, see `hold` below)? If not,
  // the deferred batch is flushed right here, before this batch's upload is put in front of the decode streams: its control pass and
  // decode kernels must not queue behind the copy of the batch after it (the upload / decode overlap the staging exists for).
  const bool may_hold = (flags & ETLG_F_ASYNC) && out_dev && !no_ctrl && !scan && nframes && c->ctl_hold_mode && c->ctl_async_mode && c->last_had_ctrl &&
                        c->deferred && c->deferred->defer_ctl;
  if (!may_hold) { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  // ---- ASYNC with HOST input (the staging batcher of a Rust host, crates/etl-gfx950/src/batcher.rs: pinned buffers from
  //      etlg_host_alloc): the bytes and the sidecar are copied into a device block the batch owns, on a copy stream of their own,
  //      and from here on the batch IS a device-input batch — it joins the chain, and its upload runs beside the decode of the batch
  //      before it (double buffering: the caller fills its next buffer meanwhile). The caller keeps the host buffers untouched until
  //      the batch is synced, as for every ASYNC batch (include/etlg.h).
  void* stage_blk = nullptr; size_t stage_cap = 0; hipEvent_t h2d_done = nullptr;
  // (in place before the first HIP call below: an early return gives the block and the event back)
  struct StageGuard { etlg_ctx* c; void*& p; size_t& cap; hipEvent_t& ev; ~StageGuard() { if (p) blk_give(c, c->gen, p, cap, false); if (ev) c->ev_pool.push_back(ev); } } stage_guard{c, stage_blk, stage_cap, h2d_done};
  if ((flags & ETLG_F_ASYNC) && out_dev && !in_dev && !scan && len && nframes && !c->copy.active && !c->force_multipass && len < (1ull << 31)) {
    const size_t o_offs = (len + 16 + 255) & ~(size_t)255;   // the kernels' readers may touch up to 16 bytes past the input
    HIPCHK(c, blk_take(c, o_offs + (nframes + 1) * 4 + 64, false, &stage_blk, &stage_cap));
    if (!c->h2d_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking));
    if (c->ev_pool.empty()) { hipEvent_t e = nullptr; HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_pool.push_back(e); }
    h2d_done = c->ev_pool.back(); c->ev_pool.pop_back();
    HIPCHK(c, hipMemcpyAsync(stage_blk, buf, len, hipMemcpyHostToDevice, c->h2d_stream));
    HIPCHK(c, hipMemcpyAsync((uint8_t*)stage_blk + o_offs, frame_offsets, (nframes + 1) * 4, hipMemcpyHostToDevice, c->h2d_stream));
    HIPCHK(c, hipEventRecord(h2d_done, c->h2d_stream));
    if (!c->stream2) {   // the chain may put this batch on either decode stream: both exist before anything waits
      HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
      HIPCHK(c, hipEventCreateWithFlags(&c->tail2, hipEventDisableTiming));
    }
    // (a call that may hold the deferred batch lets only the control stream wait now — its pre-pass reads the upload; the decode streams
    // get their wait once the held batch has been enqueued on them, below)
    for (hipStream_t w : {may_hold ? (hipStream_t) nullptr : c->stream, may_hold ? (hipStream_t) nullptr : c->stream2, c->ctl_stream, c->scan_stream}) if (w) HIPCHK(c, hipStreamWaitEvent(w, h2d_done, 0));
    buf = (const uint8_t*)stage_blk; frame_offsets = (const uint32_t*)((const uint8_t*)stage_blk + o_offs);
    in_dev = true;
    c->staged_async++;
  }
  // ASYNC batches are chained on the device (DecParams.carry) and may be decoded again when they are synced, so everything
  // they read must still be there then: device-resident input AND sidecar (the context's staging and scan buffers are shared
  // by all batches). Anything else is decoded synchronously; etlg_batch_sync on such a batch returns its stored result.
  // Without the caller's no-control assertion the first attempt is optimistic as long as the stream has not been carrying Relation /
  // DDL frames (a control frame then fails the batch with a hint and finish_batch takes the control path); on a stream that does
  // carry them (last_had_ctrl) the control pre-pass runs ahead on its own stream (ctl_begin) — that needs the sidecar.
  const bool async_ok = (flags & ETLG_F_ASYNC) && out_dev && in_dev && (!c->copy.active || c->copy.async) && !c->force_multipass && len < (1ull << 31);
  // (the pre-pass + host control plane of this batch may only run ahead of batches that are themselves on the control path: each of
  // those holds a snapshot of the control state it started from. An OPTIMISTIC batch still pending — enqueued before the context knew
  // the stream carries Relation / DDL frames — may have to be decoded again when it is synced, against the schemas of ITS position in
  // the stream; a control plane that had run ahead of it had changed them under it: rows decoded against the table's next schema,
  // found by tools/async_fuzz.py. Such a batch waits for the chain to drain first — once per stream, at the transition.)
  bool pending_optimistic = false;
  for (const etlg_batch* pb : c->pending) if (pb->pending && !pb->ctl_async) pending_optimistic = true;
  const bool ctl_ahead = async_ok && !no_ctrl && c->last_had_ctrl && !scan && nframes && c->ctl_async_mode && !pending_optimistic;
  const bool async = async_ok && (no_ctrl || !c->last_had_ctrl || ctl_ahead);
  // The pipelined control path, one step deeper (round 4): when the batch held back by the previous call is on that path too, THIS
  // batch's pre-pass goes out before the held one is flushed — i.e. before the host waits for the held batch's pre-pass, runs its
  // control plane (~120 us of host work per 64 MiB of cfg5) and enqueues its decode. The pre-pass of batch k + 1 then runs while the
  // host works on batch k instead of beside batch k's decode kernel (where its small kernels only get going as that one drains, and
  // the host's wait for them was 130 us per batch). The pre-pass kernels read the input and the transaction state the pre-pass before
  // them left on the device — nothing the host control plane of the held batch changes. Two pre-passes in flight: two sets of buffers.
  const bool hold = ctl_ahead && c->ctl_hold_mode && c->deferred && c->deferred->defer_ctl;
  if (!hold) {
    const int32_t rc_ = flush_deferred(c); (void)rc_;
    if (may_hold && h2d_done) for (hipStream_t w : {c->stream, c->stream2}) if (w) HIPCHK(c, hipStreamWaitEvent(w, h2d_done, 0));
  }
  hipStream_t s = c->stream;
  if (!async) { const int32_t rc = drain_pending(c); if (rc != ETLG_OK) return rc; clear_error(c); }
  // The result blocks live in a ring of kResRing: include/etlg.h asks for fewer than 32 batches in flight, and a caller that keeps more
  // (etlg_copy_decode(ASYNC) queues batches back to back too) must not get a block that a batch in flight — or one being decoded again —
  // still writes: the oldest batches are finished first. They stay the caller's: their sync returns what was found here.
  while (c->pending.size() >= etlg_ctx::kResRing - 2) { const int32_t rc = finish_batch(c, c->pending.front()); (void)rc; }

  // ---- record boundaries: the caller's sidecar, or the device scan (scan.hip)
  const uint32_t* h_offs = frame_offsets;
  const uint8_t* d_in_ptr = buf;  // device address of the input
  if (!in_dev) {
    HIPCHK(c, c->d_in.ensure(len + 64));
    if (len) HIPCHK(c, hipMemcpyAsync(c->d_in.p, buf, len, hipMemcpyHostToDevice, s));
    d_in_ptr = (const uint8_t*)c->d_in.p;
  }
  auto* b = new etlg_batch();
  b->ctx = c; b->ctx_gen = c->gen;
  BatchGuard guard{b};
  b->user_no_ctrl = no_ctrl; b->out_dev = out_dev; b->in_dev = in_dev; b->scan = scan; b->len = len;
  b->finish_what = (flags & ETLG_F_FINISH_CELLS) ? (uint32_t)(ETLG_FINISH_ARRAYS | ETLG_FINISH_FLOATS) : 0u;
  b->stage_blk = stage_blk; b->stage_cap = stage_cap; b->h2d_done = h2d_done;   // the batch owns them from here (etlg_batch_free)
  stage_blk = nullptr; h2d_done = nullptr;
  if (c->copy.active && c->copy.stage_blk) {   // an ASYNC table-copy batch whose rows were staged by etlg_copy_decode: likewise
    b->stage_blk = c->copy.stage_blk; b->stage_cap = c->copy.stage_cap; b->h2d_done = c->copy.h2d_done;
    c->copy.stage_blk = nullptr; c->copy.h2d_done = nullptr;
  }
  b->host_in = in_dev ? nullptr : buf; b->host_offs = (in_dev || scan) ? nullptr : h_offs; b->dev_in = in_dev ? buf : nullptr;
  b->d_in_ptr = d_in_ptr; b->user_offs = frame_offsets;
  if (scan) {
    // Round 6: when the batch will be decoded by the fixed-width plan behind its sidecar pre-pass (the cfg2 shape) and the stream has
    // shown how many frames a batch of this size holds, the decode is enqueued AT ONCE behind the scan: the grids are sized by a bound,
    // the kernels read the count the scan leaves on the device (DecParams.nframes_dev), and the host never waits between a scan and
    // the decode behind it. A scan that did not hold, or a batch beyond the bound, comes back with fused_fail bit 5
    // and takes the round-5 route when it is finished (finish_batch). ASYNC: the scan on its own stream, the decode streams wait for
    // its event; otherwise scan and decode follow each other on the context's stream and the call waits once, at its end.
    const size_t offs_cap = len / 24 + 1024;
    // (the estimate: this batch's bytes at the bytes per frame of the last scanned batch of the stream; the bound: a quarter above it)
    const uint64_t nf_est = c->scan_last_len ? std::max<uint64_t>(1, (uint64_t)((unsigned __int128)len * c->scan_last_nf / c->scan_last_len)) : 0;
    const uint64_t bound = std::min<uint64_t>(offs_cap, nf_est + nf_est / 4 + 4096);
    const bool chain_scan = c->scan_chain_mode && c->scan_last_nf && len && no_ctrl && c->worker == ETLG_WORKER_APPLY && !c->copy.active && !c->prof_serial &&
                            c->plan_mode != 0 && (c->fused_kernel < 0 || c->fused_kernel == 3) && c->n_plan_tabs && c->plan_covers_all && c->plan_pre != 0 &&
                            c->plan_uniform_dw != 0 && !c->plan_skip && c->plan_max_row <= 512 && !c->fused_dbg && c->side_valid && !c->side_dirty && !c->slots_dirty &&
                            c->last_epochs.empty() && bound < (1u << 29) && !(async && !c->pending.empty() && c->pending.back()->force_rerun);
    if (async || chain_scan) {
      if (c->offs_pool.empty()) c->offs_pool.push_back(new DevBuf());
      b->scan_offs = c->offs_pool.back(); c->offs_pool.pop_back();
    }
    if (chain_scan) {
      if (async && !c->scan_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->scan_stream, hipStreamNonBlocking));
      b->scan_s = async ? c->scan_stream : s;
      HIPCHK(c, b->scan_offs->ensure((offs_cap + 2) * 4 + 128));
      b->d_scan_res = (uint32_t*)((uint8_t*)b->scan_offs->p + (((offs_cap + 2) * 4 + 63) & ~(size_t)63));   // (the scan's result words live with the batch: no copy behind the scan)
      HIPCHK(c, scan_begin(c, c->scan_job, d_in_ptr, len, b->scan_s, *b->scan_offs, b->d_scan_res));
      b->nf_est = nf_est;
      if (async) {
        if (c->ev_pool.empty()) { hipEvent_t e = nullptr; HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_pool.push_back(e); }
        hipEvent_t sev = c->ev_pool.back(); c->ev_pool.pop_back();
        HIPCHK(c, hipEventRecord(sev, c->scan_stream));
        if (!c->stream2) {
          HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
          HIPCHK(c, hipEventCreateWithFlags(&c->tail2, hipEventDisableTiming));
        }
        for (hipStream_t w : {c->stream, c->stream2}) HIPCHK(c, hipStreamWaitEvent(w, sev, 0));
        c->ev_pool.push_back(sev);   // (recorded and waited for: the waits keep their own reference to this recording)
      }
      b->scan_chained = true;
      c->scan_chained_n++;
      const int32_t rc = decode_tail(c, b, (size_t)bound, async, (async && !c->pending.empty()) ? c->pending.back() : nullptr);
      if (rc != ETLG_OK) return rc;
      guard.b = nullptr;
      *out = b;
      if (!async) return finish_batch(c, b);
      b->pending = true; b->v.on_device = 1; fill_view_common(b); c->pending.push_back(b);
      return ETLG_OK;
    }
    if (async) {
      // The scan of THIS batch runs on its own stream while the previous batch is still being decoded on the context's, and
      // nobody waits for it here: the call returns with the scan in flight and the NEXT call (or the batch's sync) collects
      // the frame count and enqueues the decode. The input must be complete when the call is made (include/etlg.h).
      if (!c->scan_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->scan_stream, hipStreamNonBlocking));
      HIPCHK(c, scan_begin(c, c->scan_job, d_in_ptr, len, c->scan_stream, *b->scan_offs));
      b->deferred = true; b->pending = true; b->v.on_device = 1;
      c->deferred = b; c->pending.push_back(b);
      guard.b = nullptr;
      *out = b;
      return ETLG_OK;
    }
    HIPCHK(c, device_scan(c, d_in_ptr, len, &nframes, s, c->d_offs));
    if (nframes >= (1u << 30)) return lib_error(c, ETLG_InvalidArgument, "batch too large (max 4 GiB, 2^30 frames)");
  }
  if (ctl_ahead) {
    // the pre-pass goes out now, on the control stream; the host control plane and the decode follow when the next call comes in
    if (!c->ctl_stream) {
      HIPCHK(c, hipStreamCreateWithFlags(&c->ctl_stream, hipStreamNonBlocking));
      HIPCHK(c, hipEventCreateWithFlags(&c->mp_tail, hipEventDisableTiming));
      HIPCHK(c, c->d_ctl_res.ensure(sizeof(DevResult) * etlg_ctx::kCtlRing));
      if (b->h2d_done) HIPCHK(c, hipStreamWaitEvent(c->ctl_stream, b->h2d_done, 0));
    }
    b->nframes_in = nframes;
    b->ctl_async = true;
    DecParams& cp = b->ctl_params;
    cp = DecParams{};
    cp.in = d_in_ptr; cp.offs = frame_offsets; cp.nframes = (uint32_t)nframes; cp.nblocks = ((uint32_t)nframes + kBlock - 1) / kBlock; cp.in_len = len;
    cp.worker_kind = (uint32_t)c->worker; cp.sync_table = c->sync_table; cp.copy_slot = -1; cp.host_err_frame = 0xFFFFFFFFu;
    cp.in_txn = c->in_txn; cp.final_lsn = c->final_lsn; cp.next_ord = c->next_ord;
    cp.flags = 32u;
    etlg_batch* prev = c->pending.empty() ? nullptr : c->pending.back();
    if (prev && prev->pending) {
      if (prev->ctl_async && prev->ctl_params.res) cp.carry = prev->ctl_params.res;   // the pre-passes chain among themselves (control stream order)
      else if (prev->kdone) { HIPCHK(c, hipStreamWaitEvent(c->ctl_stream, prev->kdone, 0)); cp.carry = prev->d_res_blk; }   // an optimistic batch: its decode result
      else { const int32_t rc = drain_pending(c); if (rc != ETLG_OK) return rc; cp.in_txn = c->in_txn; cp.final_lsn = c->final_lsn; cp.next_ord = c->next_ord; }
    }
    b->ctl_set = (int)(c->ctl_seq & 1u);   // (the held batch's pre-pass wrote the other set)
    cp.res = (DevResult*)c->d_ctl_res.p + (c->ctl_seq++ % etlg_ctx::kCtlRing);
    { const int32_t rc = ctl_begin(c, b, cp, c->ctl_stream, true); if (rc != ETLG_OK) { if (hold) { const int32_t rc_ = flush_deferred(c); (void)rc_; } return rc; } }
    if (hold) {   // now the held batch: its pre-pass result, its control plane, its decode
      const int32_t rc_ = flush_deferred(c);
      // (the held batch could not be enqueued: this batch's pre-pass was chained to a control result that will never be the state the
      // decode before it leaves — it starts again from the host's state when it is flushed)
      if (rc_ != ETLG_OK) { b->force_rerun = true; b->ctl_started = false; }
      if (b->h2d_done) for (hipStream_t w : {c->stream, c->stream2}) if (w) HIPCHK(c, hipStreamWaitEvent(w, b->h2d_done, 0));
    }
    b->deferred = true; b->defer_ctl = true; b->pending = true; b->v.on_device = 1;
    c->deferred = b; c->pending.push_back(b);
    c->ctl_ahead_n++;
    guard.b = nullptr;
    *out = b;
    return ETLG_OK;
  }
  // A chain behind a batch that has to be decoded again does not heal by itself: every batch enqueued behind one that is marked for a
  // second attempt starts from that batch's unusable result, refuses to run, and is itself decoded again — synchronously — when it is
  // synced; with the caller keeping its window full that went on for the rest of the stream (one fixed-width batch with an UPDATE in it,
  // and every later batch was a poisoned first attempt plus a synchronous second one). The batches in flight are finished first, once;
  // this batch then starts a fresh chain from the host's exact state.
  if (async && !c->pending.empty() && c->pending.back()->force_rerun) { const int32_t rc = drain_pending(c); if (rc != ETLG_OK) return rc; clear_error(c); c->chain_healed++; }
  {
    const int32_t rc = decode_tail(c, b, nframes, async, (async && !c->pending.empty()) ? c->pending.back() : nullptr);
    if (rc != ETLG_OK) return rc;
  }
  guard.b = nullptr;
  *out = b;
  if (async) { b->pending = true; b->v.on_device = 1; fill_view_common(b); c->pending.push_back(b); return ETLG_OK; }
  return finish_batch(c, b);
}

}  // extern "C"

namespace {

// The decode of a batch whose boundary scan was left in flight by etlg_decode: collect the frame count, enqueue the kernels.
// A failure here belongs to THAT batch (its sync reports it), not to the call that happens to run this.
int32_t flush_deferred(etlg_ctx* c) {
  SlowScope slow_scope_flush_deferred(c, "flush_deferred");
  etlg_batch* b = c ? c->deferred : nullptr;
  if (!b) return ETLG_OK;
  c->deferred = nullptr;
  b->deferred = false;
  size_t nframes = 0;
  int32_t rc = ETLG_OK;
  const bool pre_pass_ahead = b->defer_ctl;
  if (b->defer_ctl) { b->defer_ctl = false; nframes = b->nframes_in; }   // its control pre-pass ran ahead: decode_tail collects it (standard_path -> ctl_finish)
  else {
    const hipError_t e = scan_collect(c, c->scan_job, &nframes);
    if (e != hipSuccess) rc = lib_error(c, ETLG_DeviceError, hipGetErrorString(e));
    else if (nframes >= (1u << 30)) rc = lib_error(c, ETLG_InvalidArgument, "batch too large (max 4 GiB, 2^30 frames)");
    else { c->scan_last_nf = nframes; c->scan_last_len = b->len; }
  }
  if (rc == ETLG_OK) {
    // A batch whose boundary scan was deferred joined the chain when the stream had shown no Relation / DDL frame. If it has by now (a
    // batch before this one was synced in between), this batch takes the control path — which starts from the HOST's carried state and
    // control plane: everything before it is finished first. (It used to run its control pass from the state of the last SYNCED batch
    // with the batches in between still pending: rows decoded against missing table state — found by tools/async_fuzz.py, round 4.)
    if (!pre_pass_ahead && !b->user_no_ctrl && c->last_had_ctrl)
      while (!c->pending.empty() && c->pending.front() != b) (void)finish_batch(c, c->pending.front());
    etlg_batch* prev = nullptr;   // the batch issued just before this one, if it is still in flight
    for (size_t i = 0; i < c->pending.size(); i++) if (c->pending[i] == b && i > 0) prev = c->pending[i - 1];
    if (prev && prev->force_rerun) {   // (as in etlg_decode: a chain behind a batch that is decoded again is finished first, once)
      while (!c->pending.empty() && c->pending.front() != b) (void)finish_batch(c, c->pending.front());
      prev = nullptr;
      c->chain_healed++;
    }
    // This batch was in flight (its scan) when a batch before it was marked for a second attempt, and was marked with it; but its
    // DECODE goes out only now. With nothing in flight in front of it, it starts from the final state: one attempt is enough. (Left
    // marked, it was decoded again at its sync and marked everything behind it in turn: a chain without a sidecar never healed.)
    if (!prev && !pre_pass_ahead) b->force_rerun = false;
    rc = decode_tail(c, b, nframes, true, prev);
  }
  if (rc != ETLG_OK) {  // nothing was enqueued for it: it is finished, with this error
    for (size_t i = 0; i < c->pending.size(); i++) if (c->pending[i] == b) { c->pending.erase(c->pending.begin() + (long)i); break; }
    b->pending = false; b->finished = true; b->rc = rc; b->err = c->err; b->err_detail = c->err_detail;
    b->err.detail = b->err_detail.empty() ? nullptr : b->err_detail.c_str();
    return rc;
  }
  fill_view_common(b);
  return ETLG_OK;
}

// Everything of etlg_decode that needs the frame count: parameters, result block, side inputs, outputs, the first kernel.
int32_t decode_tail(etlg_ctx* c, etlg_batch* b, size_t nframes, bool async, etlg_batch* prev) {
  SlowScope slow_scope_decode_tail(c, "decode_tail");
  hipStream_t s = c->stream;
  b->plan_decided = -1;
  const bool scan = b->scan, in_dev = b->in_dev, no_ctrl = b->user_no_ctrl;
  const size_t len = b->len;
  const uint8_t* d_in_ptr = b->d_in_ptr;
  const uint32_t* frame_offsets = b->user_offs;
  const uint32_t* h_offs = b->user_offs;
  const uint32_t nf = (uint32_t)nframes;
  DecParams& p = b->params;
  p = DecParams{};
  p.in = d_in_ptr;
  if (scan) p.offs = (const uint32_t*)(b->scan_offs ? b->scan_offs->p : c->d_offs.p);
  else if (in_dev) p.offs = frame_offsets;
  else {
    HIPCHK(c, c->d_offs.ensure((nframes + 1) * 4));
    HIPCHK(c, hipMemcpyAsync(c->d_offs.p, h_offs, (nframes + 1) * 4, hipMemcpyHostToDevice, s));
    p.offs = (const uint32_t*)c->d_offs.p;
  }
  p.nframes = nf; p.nblocks = (nf + kBlock - 1) / kBlock; p.in_len = len;
  p.worker_kind = (uint32_t)c->worker; p.sync_table = c->sync_table;
  p.flags = c->fused_dbg & 0xF00u;  // profiling ablations (results are wrong)
  p.copy_slot = -1;
  if (c->copy.active) { p.flags |= 2u; p.copy_slot = c->copy.slot; b->copy = c->copy; }
  p.host_err_frame = 0xFFFFFFFFu;
  // carried transaction state: the host's (every earlier batch is finished), or — behind pending ASYNC batches — whatever the
  // batch issued just before this one leaves in its result block
  p.in_txn = c->in_txn; p.final_lsn = c->final_lsn; p.next_ord = c->next_ord;
  p.carry = (async && prev && !c->copy.active) ? prev->d_res_blk : nullptr;   // (table-copy batches: a virtual transaction each, nothing carried)
  p.nframes_dev = b->scan_chained ? b->d_scan_res : nullptr;

  HIPCHK(c, c->d_res.ensure(sizeof(DevResult) * etlg_ctx::kResRing));
  const uint32_t res_slot = c->res_seq % etlg_ctx::kResRing;
  // ---- two streams. A batch whose first attempt is a single-pass kernel (k_plan, k_fused, k_cells) and whose predecessor in the
  //      chain is one too, and still in flight, is enqueued on the OTHER decode stream and told (flags bit 4) that the state it
  //      starts from arrives late: its kernel does not read the predecessor's result block at its start; the few tiles without a
  //      Begin before them in the batch poll for it (plan.hip plan_late_carry, lookback.hip.h txn_lookback). So the tail of batch k — its last waves, the write-back of its dirty lines, the dispatch of
  //      the next kernel — overlaps the staging and parsing of batch k+1: 63.9 -> 50.1 us per 64 MiB cfg2 batch measured with two
  //      independent chains (profiles/r03_plan_development.json). Ordering kept: k+1 starts after k-1 has completed (look-back
  //      buffers rotate with distance two; k's waves are all dispatched by then, so a tile of k+1 that polls cannot hold a slot k
  //      needs), side inputs unchanged (a change drains the chain), no lap boundary of the result ring. Decided before anything is
  //      enqueued for the batch; everything below then runs on the chosen stream.
  struct StreamSwitch { etlg_ctx* c; hipStream_t saved; ~StreamSwitch() { c->stream = saved; } } sw{c, c->stream};
  SlowScope slow_scope_pre(c, "decode_tail: whole after params");
  bool beside = false;
  const bool first_try_single = p.nframes && !c->force_multipass && len < (1ull << 31) && (no_ctrl || !c->last_had_ctrl) && !b->ctl_async;
  if (async && prev && prev->pending && prev->level <= 1 && prev->used_fused && !prev->force_rerun && c->overlap_mode && first_try_single && c->res_seq != 0 && res_slot >= 2 &&
      c->side_valid && !c->side_dirty && !c->slots_dirty && c->last_epochs.empty() && !b->copy.active && !c->prof_serial) {
    p.flags |= 1u;
    const std::vector<EpochRec> no_eps;
    { const int32_t rc = build_side_inputs(c, b, no_eps); if (rc != ETLG_OK) return rc; }   // the unchanged-inputs path: no stream work
    { const int32_t rc = setup_outputs(c, b); if (rc != ETLG_OK) return rc; }
    b->plan_decided = plan_wanted(c, b) ? 1 : 0;
    beside = true;   // (a look-back buffer that has to grow synchronises both streams: take_descriptors)
  }
  // Batches on the pipelined control path (their pre-pass ran ahead, ctl_async) alternate between the two decode streams as well
  // (round 4): the decode kernel of batch k+1 — enqueued by standard_path below, behind its own control pass, side-input set and
  // outputs, all of which go to the stream chosen here — starts beside the tail of batch k's; it reads the carried transaction state
  // late like every batch that runs beside its predecessor (flags bit 4). cfg5: the decode kernels were a single-stream chain.
  else if (async && prev && prev->pending && prev->ctl_async && b->ctl_async && prev->level <= 1 && prev->used_fused && !prev->force_rerun && c->overlap_mode &&
           c->ctl_overlap_mode && p.nframes && !c->force_multipass && len < (1ull << 31) && c->res_seq != 0 && res_slot >= 2 && !b->copy.active && !c->prof_serial) {
    beside = true;
  }
  if (beside) {
    if (!c->stream2) {
      HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
      HIPCHK(c, hipEventCreateWithFlags(&c->tail2, hipEventDisableTiming));
    }
    b->sidx = prev->sidx ^ 1;
    c->stream = b->sidx ? c->stream2 : sw.saved;
    // the batch before the predecessor must be complete; it is, when it ran on this stream — otherwise wait for its event
    etlg_batch* pp = nullptr;
    for (size_t i = 0; i + 1 < c->pending.size(); i++) if (c->pending[i + 1] == prev) pp = c->pending[i];
    if (pp && pp->pending && pp->kdone && pp->sidx != b->sidx) HIPCHK(c, hipStreamWaitEvent(c->stream, pp->kdone, 0));
    p.flags |= 16u;
    c->overlapped++;
  } else {
    b->sidx = 0;
    if (c->tail2_set) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->tail2, 0)); c->tail2_set = false; }   // join: everything enqueued on stream2 so far
  }
  s = c->stream;
  if (b->side && !(b->side->synced & (1u << b->sidx))) {   // the set was uploaded on the other decode stream: the first batch over here waits for it
    HIPCHK(c, hipStreamWaitEvent(s, b->side->ready, 0));
    b->side->synced |= 1u << b->sidx;
  }
  SlowScope slow_scope_ring(c, "decode_tail: result ring and after");
  {  // result block: next slot of a ring that is re-initialised once per lap. Slot 31 is the carry source of the batch in
     // slot 0, so it is re-initialised one batch later than the others.
    const uint32_t seq = c->res_seq++;
    const uint32_t slot = seq % etlg_ctx::kResRing;
    DevResult* ring = (DevResult*)c->d_res.p;
    // the blocks about to be re-initialised belong to batches whose result copies travel on res_stream: with two decode streams a
    // batch completes within microseconds of its predecessor, so "the copy of the batch before last is long done" no longer holds —
    // wait (on the device) for the copy of the latest batch, which is behind all the others
    { SlowScope sw1(c, "ring: wait event");
    if ((slot == 0 || slot == 1) && prev && prev->pending && prev->done) HIPCHK(c, hipStreamWaitEvent(s, prev->done, 0)); }
    { SlowScope sw2(c, "ring: init copy");
    if (seq == 0) HIPCHK(c, hipMemcpyAsync(ring, c->ring_h2d ? (const void*)c->h_init_ring : (const void*)c->d_init_ring, sizeof(DevResult) * etlg_ctx::kResRing, c->ring_h2d ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    else if (slot == 0) HIPCHK(c, hipMemcpyAsync(ring, c->ring_h2d ? (const void*)c->h_init_ring : (const void*)c->d_init_ring, sizeof(DevResult) * (etlg_ctx::kResRing - 1), c->ring_h2d ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    else if (slot == 1) HIPCHK(c, hipMemcpyAsync(ring + (etlg_ctx::kResRing - 1), c->ring_h2d ? (const void*)c->h_init_ring : (const void*)c->d_init_ring, sizeof(DevResult), c->ring_h2d ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s)); }
    b->d_res_blk = ring + slot;
    b->res_seq_no = seq;
  }
  p.res = b->d_res_blk;
  { SlowScope slow_scope_pool(c, "decode_tail: pinned result block");
  if (c->res_pool.empty()) { DevResult* r = nullptr; HIPCHK(c, hipHostMalloc((void**)&r, sizeof(DevResult), hipHostMallocDefault)); c->res_pool.push_back(r); }
  b->h_res = c->res_pool.back(); c->res_pool.pop_back(); }
  if (b->copy.active && !b->copy.direct) launch_copy(c, b->copy, p);  // rows -> Insert frames (writes p.in / p.offs), row-level errors

  // ---- first attempt. A single-pass kernel runs OPTIMISTICALLY as if the batch held no Relation / DDL frame (they are
  //      <0.1 % of frames and absent from almost every batch): no classify / control-list kernels, no host round trip. A
  //      control frame makes that kernel report ETLG_E_CTRL_HINT and the batch takes the control path then (finish_batch).
  const bool single_pass = nf && !c->force_multipass && len < (1ull << 31);
  // ... unless the batch before this one held control frames and the caller asserts nothing: streams that change schemas often
  // (DDL messages every few hundred transactions) would pay for a wasted kernel on every batch
  if (b->plan_decided >= 0) {   // side inputs and outputs were set up for the stream decision above
    { const int32_t rc = enqueue_single(c, b, b->plan_decided ? 0 : 1); if (rc != ETLG_OK) return rc; }
  } else if (single_pass && (no_ctrl || !c->last_had_ctrl) && !b->ctl_async) {
    p.flags |= 1u;
    const std::vector<EpochRec> no_eps;
    { const int32_t rc = build_side_inputs(c, b, no_eps); if (rc != ETLG_OK) return rc; }
    { const int32_t rc = setup_outputs(c, b); if (rc != ETLG_OK) return rc; }
    { const int32_t rc = enqueue_single(c, b, plan_wanted(c, b) ? 0 : 1); if (rc != ETLG_OK) return rc; }
  } else {
    const int32_t rc = standard_path(c, b);
    if (rc != ETLG_OK) return rc;
  }
  if (b->n_slots_view == ~(size_t)0) b->n_slots_view = c->slots.size();
  SlowScope slow_scope_tail(c, "decode_tail: result copy");
  if (async) {
    // the result block is copied on a second stream: on the context's stream the next batch's kernel follows this one
    // directly (a 200-byte device-to-host copy is a 4 us blit kernel plus two dispatch gaps when it sits between them)
    for (int k = 0; k < 2; k++) {
      if (c->ev_pool.empty()) { hipEvent_t e = nullptr; HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_pool.push_back(e); }
      (k ? b->done : b->kdone) = c->ev_pool.back(); c->ev_pool.pop_back();
    }
    if (!c->res_stream) {   // highest priority: its small copies are blit kernels, and the decode streams keep every wave slot of the chip taken
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      HIPCHK(c, hipStreamCreateWithPriority(&c->res_stream, hipStreamNonBlocking, hi));
    }
    HIPCHK(c, hipEventRecord(b->kdone, s));
    if (b->sidx) { HIPCHK(c, hipEventRecord(c->tail2, s)); c->tail2_set = true; }
    HIPCHK(c, hipStreamWaitEvent(c->res_stream, b->kdone, 0));
    HIPCHK(c, hipMemcpyAsync(b->h_res, b->d_res_blk, sizeof(DevResult), hipMemcpyDeviceToHost, c->res_stream));
    HIPCHK(c, hipEventRecord(b->done, c->res_stream));
  } else {
    HIPCHK(c, hipMemcpyAsync(b->h_res, b->d_res_blk, sizeof(DevResult), hipMemcpyDeviceToHost, s));
  }

  return ETLG_OK;
}

}  // namespace

extern "C" {

int32_t etlg_batch_sync(etlg_ctx* c, etlg_batch* b) {
  if (!c || !b) return ETLG_InvalidArgument;
  InvariantScope inv_scope{c, "etlg_batch_sync"};
  if (b->pending) {
    // batches finish in issue order: the carried transaction state of the context is the state after the LAST finished one
    while (b->pending && !c->pending.empty()) { const int32_t rc = finish_batch(c, c->pending.front()); (void)rc; }
  }
  // the context's last error becomes this batch's (the reference returns the error of the call that hit it)
  c->err = b->err; c->err_detail = b->err_detail; c->err.detail = c->err_detail.empty() ? nullptr : c->err_detail.c_str();
  return b->rc;
}

int32_t etlg_batch_header_to_device(etlg_ctx* c, etlg_batch* b, void* dst) {
  if (!c || !b || !dst) return ETLG_InvalidArgument;
  static_assert(offsetof(DevResult, n_frames) == 56, "header layout");
  if (b->deferred) { const int32_t rc = flush_deferred(c); if (rc != ETLG_OK) return rc; }
  if (b->pending && b->kdone && c->res_stream) {   // in flight: behind its kernels on the stream its result block travels on (etlg_ctx_fence joins)
    HIPCHK(c, hipMemcpyAsync(dst, b->d_res_blk, 64, hipMemcpyDeviceToDevice, c->res_stream));
    c->hdr_in_flight = true;
    return ETLG_OK;
  }
  HIPCHK(c, hipMemcpyAsync(dst, b->d_res_blk, 64, hipMemcpyDeviceToDevice, c->stream));
  return ETLG_OK;
}

int32_t etlg_ctx_fence(etlg_ctx* c) {
  if (!c) return ETLG_InvalidArgument;
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (c->tail2_set) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->tail2, 0)); c->tail2_set = false; }
  if (c->hdr_in_flight && c->res_stream) {
    if (!c->fence_ev) HIPCHK(c, hipEventCreateWithFlags(&c->fence_ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->fence_ev, c->res_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->fence_ev, 0));
    c->hdr_in_flight = false;
  }
  return ETLG_OK;
}

int32_t etlg_batch_download(etlg_ctx* c, etlg_batch* b) {
  if (!c || !b) return ETLG_InvalidArgument;
  if (b->pending) { int32_t rc = etlg_batch_sync(c, b); (void)rc; }
  return download_batch(c, b);
}

int32_t etlg_batch_view_get(const etlg_batch* b, etlg_batch_view* out) {
  if (!b || !out) return ETLG_InvalidArgument;
  *out = b->v;
  return ETLG_OK;
}

void etlg_batch_free(etlg_batch* b) {
  if (!b) return;
  if (b->ctx && ctx_alive(b->ctx, b->ctx_gen)) {
    etlg_ctx* c = b->ctx;
    if (b->pending) {  // freed without a sync: it still has to finish, in order, for the context's carried state to be right
      while (b->pending && !c->pending.empty()) { const int32_t rc = finish_batch(c, c->pending.front()); (void)rc; }
      b->pending = false;
    }
    if (b->h_res) (void)hipStreamSynchronize(c->stream);
    side_release(b);
    if (b->ctl_ev) { if (b->ctl_started && c->ctl_stream) (void)hipStreamSynchronize(c->ctl_stream); c->ev_pool.push_back(b->ctl_ev); }
    if (b->h_ctl) c->res_pool.push_back(b->h_ctl);
    if (b->done) c->ev_pool.push_back(b->done);
    if (b->kdone) c->ev_pool.push_back(b->kdone);
    if (b->dev) c->out_pool.push_back(b->dev);
    if (b->scan_offs) c->offs_pool.push_back(b->scan_offs);
    if (b->h2d_done) c->ev_pool.push_back(b->h2d_done);
    if (b->stage_blk) blk_give(c, c->gen, b->stage_blk, b->stage_cap, false);
    if (b->h_res) c->res_pool.push_back(b->h_res);
    if (b->h_arena) c->harena_pool.emplace_back(b->h_arena, b->h_arena_cap);
  } else {  // the context is gone (its pools with it): release what the batch owns outright
    if (b->dev) { b->dev->release(); delete b->dev; }
    if (b->scan_offs) { b->scan_offs->release(); delete b->scan_offs; }
    if (b->stage_blk) (void)hipFree(b->stage_blk);
    if (b->h2d_done) (void)hipEventDestroy(b->h2d_done);
    if (b->done) (void)hipEventDestroy(b->done);
    if (b->kdone) (void)hipEventDestroy(b->kdone);
    if (b->ctl_ev) (void)hipEventDestroy(b->ctl_ev);
    if (b->h_ctl) (void)hipHostFree(b->h_ctl);
    if (b->h_res) (void)hipHostFree(b->h_res);
    if (b->h_arena) (void)hipHostFree(b->h_arena);
  }
  delete b;
}

#include "host_handoff.inc"   // columnar hand-off: etlg_batch_columns / _rowbinary / _protobuf / _size_hints (columns.hip)

}  // extern "C"

#include "host_orchestrate.inc"   // arena download, side inputs, outputs, kernel ladder, control pre-pass, finish_batch
