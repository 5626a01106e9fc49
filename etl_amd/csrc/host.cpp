// Host side of libetl_gfx950.so: the C ABI of include/etlg.h.
//
// What lives here is the rare, serial control plane the reference runs inside
// its apply loop — stored schemas (SchemaStore), table replication states
// (StateStore), the shared per-table protocol cache and the handling of
// Relation / DDL messages (reference: crates/etl/src/replication/apply.rs:
// 2160-2276, 2363-2440, 3643-3734; crates/etl/src/schema.rs:30-61, 99-129,
// 380-441; crates/etl/src/replication/table_cache.rs:53-154) — plus the
// orchestration of the gfx950 kernels that do all per-row work. There is no
// CPU decode path in this file: without a device etlg_ctx_create fails.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <mutex>
#include <set>
#include <memory>
#include <set>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/etlg.h"
#include "dev_types.h"

using namespace etlg;

extern "C" void etlg_k_launch(int which, const DecParams* p, hipStream_t s);
extern "C" const char* etlg_k_name(int which);
extern "C" void etlg_k_ctl_pick(const uint8_t* tags, uint32_t nframes, uint32_t* hdr, uint32_t* list, uint32_t cap, hipStream_t s);
extern "C" void etlg_k_ctl_span(const uint8_t* tags, uint32_t nframes, const uint32_t* list, uint32_t n, uint32_t* span, hipStream_t s);
extern "C" void etlg_k_ctl_gather(const uint8_t* in, const uint32_t* offs, const uint32_t* frames, uint32_t nkeep, uint32_t* lens, const uint32_t* out_offs, uint8_t* out, hipStream_t s);
extern "C" void etlg_k_launch_fused(int blk, const DecParams* p, const void* q, hipStream_t s);
extern "C" int etlg_k_fused_set_lds(void);
extern "C" void etlg_k_launch_bounds(const uint8_t* in, uint64_t len, uint32_t* offs, uint32_t offs_cap, void* cur, void* clear, uint32_t clear_words,
                                     uint32_t* hints, uint32_t* result, int sequential, hipStream_t s);
extern "C" uint32_t etlg_k_bounds_tile_bytes(void);
extern "C" uint32_t etlg_k_copy_bytes_per_row(uint32_t ncols);
extern "C" int etlg_k_copy_set_lds(void);
extern "C" void etlg_k_launch_copy(const uint8_t* rows, const uint32_t* row_offs, uint32_t nrows, uint64_t rows_len, uint32_t ncols,
                                   uint32_t rel_id, uint8_t* out, uint32_t* out_offs, uint32_t lds_bytes, const DecParams* dec, hipStream_t s);
extern "C" void etlg_k_launch_cells(const DecParams* p, const void* q, hipStream_t s);
extern "C" void etlg_k_launch_plan(const DecParams* p, const void* q, hipStream_t s);
extern "C" int etlg_k_plan_set_lds(void);
extern "C" void etlg_k_col_select(const void* sel, hipStream_t s);
extern "C" void etlg_k_col_fixed(const void* job, hipStream_t s);
extern "C" void etlg_k_scan_lens(const uint32_t* lens, uint64_t n, unsigned long long* blk, int64_t* offsets, hipStream_t s);
extern "C" void etlg_k_col_list(const void* job, unsigned long long* blk, int64_t* offsets, int step, hipStream_t s);
extern "C" void etlg_k_size_hints(const void* job, hipStream_t s);
extern "C" void etlg_k_rowbinary(const void* job, unsigned long long* blk, int64_t* offsets, int step, hipStream_t s);
extern "C" void etlg_k_col_var(const void* job, unsigned long long* blk, int64_t* offsets, int step, hipStream_t s);
extern "C" int etlg_k_cells_set_lds(void);
extern "C" uint32_t etlg_k_cells_table_bytes(uint32_t maxc);
extern "C" uint32_t etlg_k_cells_maxc(void);
extern "C" uint32_t etlg_k_cells_lds_floor(uint32_t maxc);
extern "C" uint32_t etlg_k_cells_static_lds(uint32_t maxc);

constexpr int kFused = 7;  // profiling slot of the fused kernel
constexpr int kCells = 8;  // ... of the column-parallel kernel (cells.hip)
constexpr int kBounds = 9; // ... of the record-boundary scan (scan.hip)
constexpr int kCopy = 10;  // ... of the table-copy row splitter (copy.hip)
constexpr int kPlan = 11;  // ... of the fixed-width plan (plan.hip)
constexpr int kProfSlots = 12;

namespace {

// ------------------------------------------------------------ error table
const etlg_err_desc kErrTable[ETLG_E__COUNT] = {
    {ETLG_OK, ""},
    {ETLG_SourceConnectionFailed, "PostgreSQL connection failed"},
    {ETLG_InvalidState, "Invalid transaction state"},
    {ETLG_ValidationError, "Invalid commit LSN"},
    {ETLG_InvalidState, "Missing shared table state"},
    {ETLG_InvalidState, "Waiting for relation state cannot decode row event"},
    {ETLG_ConversionError, "Tuple data field count does not match schema"},
    {ETLG_ConversionError, "Tuple missing source value for full row image"},
    {ETLG_InvalidData, "Required column missing from tuple"},
    {ETLG_ConversionError, "Binary format not supported in tuple data"},
    {ETLG_ConversionError, "UTF-8 conversion failed"},
    {ETLG_ConversionError, "Old tuple row width does not match schema"},
    {ETLG_ConversionError, "Replica-identity tuple shape does not match schema"},
    {ETLG_ConversionError, "Replica-identity tuple missing key columns"},
    {ETLG_ConversionError, "Replica-identity tuple missing source value"},
    {ETLG_InvalidData, "Invalid boolean value"},
    {ETLG_ConversionError, "Integer parsing failed"},
    {ETLG_ConversionError, "Float parsing failed"},
    {ETLG_ConversionError, "Numeric parsing failed"},
    {ETLG_ConversionError, "Bytea hex string conversion failed"},
    {ETLG_ConversionError, "Datetime parsing failed"},
    {ETLG_InvalidData, "UUID parsing failed"},
    {ETLG_DeserializationError, "JSON deserialization failed"},
    {ETLG_ConversionError, "Array input too short"},
    {ETLG_ConversionError, "Array input missing braces"},
    {ETLG_ConversionError, "Array input has a malformed dimensions prefix"},
    {ETLG_ConversionError, "Multidimensional array input is not supported"},
    {ETLG_ConversionError, "Array input contains an unterminated quote"},
    {ETLG_ConversionError, "Array input contains an unterminated escape"},
    {ETLG_MissingTableSchema, "Table schema not found"},
    {ETLG_CorruptedTableSchema, "Replication stream contains columns missing from the stored table schema"},
    {ETLG_ConversionError, "Failed to parse schema change message"},
    {ETLG_IoError, "I/O operation failed"},
    {ETLG_InvalidState, "Bootstrap table schema snapshot exceeded requested snapshot"},
    {ETLG_InvalidState, "Table schema snapshot mismatch"},
    {ETLG_InvalidArgument, "Control frame found in a batch declared control-free"},
    {ETLG_ConversionError, "Row data not properly terminated"},                                    // table_row.rs:100
    {ETLG_ConversionError, "Postgres COPY row contains more columns than the table schema"},       // table_row.rs:183
    {ETLG_ConversionError, "Postgres COPY row contains fewer columns than the table schema"},      // table_row.rs:239
};

// ---------------------------------------------------------------- type map
// Type::from_oid(..).unwrap_or(TEXT) + the arms of parse_cell_from_postgres_text
// (crates/etl/src/postgres/codec/text.rs:32-153).
struct ArrayOid { uint32_t oid; int32_t elem; };
const ArrayOid kArrayOids[] = {
    {1000, ETLG_TC_BOOL}, {1005, ETLG_TC_I16}, {1007, ETLG_TC_I32}, {1016, ETLG_TC_I64}, {1021, ETLG_TC_F32},
    {1022, ETLG_TC_F64}, {1231, ETLG_TC_NUMERIC}, {1001, ETLG_TC_BYTEA}, {1182, ETLG_TC_DATE}, {1183, ETLG_TC_TIME},
    {1270, ETLG_TC_TIMETZ}, {1115, ETLG_TC_TIMESTAMP}, {1185, ETLG_TC_TIMESTAMPTZ}, {2951, ETLG_TC_UUID},
    {199, ETLG_TC_JSON}, {3807, ETLG_TC_JSON}, {1028, ETLG_TC_U32},
    // generic `_xxx` arrays (ArrayCell::String)
    {143, 0}, {210, 0}, {270, 0}, {272, 0}, {273, 0}, {629, 0}, {651, 0}, {719, 0}, {775, 0}, {791, 0}, {1002, 0},
    {1003, 0}, {1006, 0}, {1008, 0}, {1009, 0}, {1010, 0}, {1011, 0}, {1012, 0}, {1013, 0}, {1014, 0}, {1015, 0},
    {1017, 0}, {1018, 0}, {1019, 0}, {1020, 0}, {1027, 0}, {1034, 0}, {1040, 0}, {1041, 0}, {1187, 0}, {1263, 0},
    {1561, 0}, {1563, 0}, {2201, 0}, {2207, 0}, {2208, 0}, {2209, 0}, {2210, 0}, {2211, 0}, {2949, 0}, {3221, 0},
    {3643, 0}, {3644, 0}, {3645, 0}, {3735, 0}, {3770, 0}, {3905, 0}, {3907, 0}, {3909, 0}, {3911, 0}, {3913, 0},
    {3927, 0}, {4073, 0}, {4090, 0}, {4097, 0}, {4192, 0}, {5039, 0}, {6151, 0}, {6152, 0}, {6153, 0}, {6155, 0},
    {6156, 0}, {6157, 0}};

int32_t type_class(uint32_t oid) {
  switch (oid) {
    case 16: return ETLG_TC_BOOL;
    case 17: return ETLG_TC_BYTEA;
    case 20: return ETLG_TC_I64;
    case 21: return ETLG_TC_I16;
    case 23: return ETLG_TC_I32;
    case 26: return ETLG_TC_U32;
    case 114: case 3802: return ETLG_TC_JSON;
    case 700: return ETLG_TC_F32;
    case 701: return ETLG_TC_F64;
    case 1082: return ETLG_TC_DATE;
    case 1083: return ETLG_TC_TIME;
    case 1114: return ETLG_TC_TIMESTAMP;
    case 1184: return ETLG_TC_TIMESTAMPTZ;
    case 1266: return ETLG_TC_TIMETZ;
    case 1700: return ETLG_TC_NUMERIC;
    case 2950: return ETLG_TC_UUID;
    default: break;
  }
  for (const auto& a : kArrayOids) if (a.oid == oid) return ETLG_TC_ARRAY;
  return ETLG_TC_STRING;
}

uint32_t slot_bytes(int32_t cls) {
  switch (cls) {
    case ETLG_TC_BOOL: case ETLG_TC_I16: case ETLG_TC_I32: case ETLG_TC_U32: return 4;
    case ETLG_TC_TIMESTAMP: case ETLG_TC_TIMESTAMPTZ: case ETLG_TC_TIMETZ: return 12;
    case ETLG_TC_UUID: return 16;
    default: return 8;
  }
}

// ------------------------------------------------------------ control state
struct StoredCol { std::string name; uint32_t type_oid; int32_t typmod; int32_t attnum; bool nullable; bool pk; };
struct StoredSchema { uint32_t table_id; uint64_t snapshot; std::string nsp, name; std::vector<StoredCol> cols; };
using SchemaPtr = std::shared_ptr<const StoredSchema>;

struct SlotHost {  // one ReplicatedTableSchema instance
  etlg_slot_desc desc;
  std::vector<etlg_slot_col> cols;
  int identity_type = 0;   // ReplicatedTableSchema::infer_identity_type (schema.rs:686-721): 0 Missing, 1 PrimaryKey, 2 Full, 3 AlternativeKey
};

struct CacheEntry { uint32_t kind; uint64_t snapshot; int32_t slot; };  // kind: 1 waiting, 2 ready
struct TState { int32_t kind; uint64_t lsn; };

struct ControlState {  // everything a failed batch must be able to roll back
  std::map<uint32_t, std::map<uint64_t, SchemaPtr>> store;
  std::map<uint32_t, CacheEntry> cache;
  size_t n_slots = 0;
};

struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = n + n / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct OutSet {  // device output arrays of one batch
  DevBuf kind, flags, table, slot, start, commit, ord, body, fixed, heap;
  size_t ev_cap = 0;
  void release() { kind.release(); flags.release(); table.release(); slot.release(); start.release(); commit.release(); ord.release(); body.release(); fixed.release(); heap.release(); }
};

struct ProfRec { int which; hipEvent_t a, b; };
struct ScanJob { const uint8_t* d_in = nullptr; size_t len = 0; hipStream_t s = nullptr; DevBuf* offs = nullptr; size_t cap = 0; uint8_t* cur = nullptr; };
struct HostErr { int32_t code = 0; uint32_t rank = 0; };
struct EpochRec { uint32_t table_id; DevEpoch ep; };

// A table-copy batch in flight (etlg_copy_decode): the rows that k_copy_frames turns into Insert frames.
struct CopyJob {
  bool active = false;
  int32_t slot = -1;
  const uint8_t* d_rows = nullptr; const uint32_t* d_row_offs = nullptr;
  uint32_t nrows = 0, ncols = 0, rel_id = 0, lds = 0;
  uint64_t rows_len = 0;
  uint8_t* d_out = nullptr; uint32_t* d_out_offs = nullptr;
};

// One uploaded copy of the side inputs (table states + cache timeline, schema slots + columns, the fixed-width plan's tables):
// ONE device block filled by ONE asynchronous copy from a pinned staging block of its own. A batch keeps the set its kernels
// read (`users`) until it is finished, so a change of the side inputs never has to wait for the batches in flight: it goes
// to a set nobody uses.
struct SideSet {
  DevBuf dev;
  uint8_t* h = nullptr; size_t h_cap = 0;
  size_t o_tables = 0, o_epochs = 0, o_slots = 0, o_cols = 0, o_ptabs = 0, o_pcols = 0;
  uint32_t n_slots = 0, n_cols = 0;
  int users = 0;
  hipEvent_t ready = nullptr;   // recorded behind the upload
  uint32_t synced = 0;          // decode streams (bit = etlg_batch::sidx) that are ordered behind the upload
};

}  // namespace

struct etlg_ctx {
  int device = 0;
  uint64_t gen = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int32_t worker = ETLG_WORKER_APPLY;
  uint32_t sync_table = 0;
  uint64_t bootstrap = 0;
  ControlState cs;
  std::map<uint32_t, TState> states;
  std::vector<std::unique_ptr<SlotHost>> slots;
  bool slots_dirty = true;
  // carried transaction state
  bool in_txn = false; uint64_t final_lsn = 0, next_ord = 0;
  // device scratch (grow-only)
  // look-back descriptors are double buffered: each single-pass launch zeroes the buffer of the next one
  size_t desc_half = 0;          // bytes per buffer
  size_t desc_dirty[4] = {0, 0, 0, 0}; // bytes at the head of each buffer that may be non-zero
  uint32_t desc_cur = 0;
  // Two decode streams: consecutive ASYNC batches of the fixed-width plan alternate between them, so the tail of batch k (its last
  // waves, the write-back, the dispatch gap) overlaps the head of batch k+1 (decode_tail, "two streams"). Everything else runs on
  // `stream`; `stream2` is created on first use.
  hipStream_t stream2 = nullptr;
  hipEvent_t tail2 = nullptr;    // recorded behind the last kernel enqueued on stream2
  bool tail2_set = false;
  hipEvent_t fence_ev = nullptr; // etlg_ctx_fence: recorded behind the header copies on res_stream
  bool hdr_in_flight = false;
  int overlap_mode = 1;          // ETLG_OVERLAP=0: one stream, as in round 2
  bool prof_serial = false;      // etlg_ctx_profile(ctx, 2): kernels timed one at a time (no second stream), for per-kernel durations
  unsigned long long overlapped = 0;   // debugging aid: batches launched beside their predecessor
  // result blocks: a ring re-initialised once per lap with one copy
  static constexpr uint32_t kResRing = 32;
  uint32_t res_seq = 0;
  DevResult* h_init_ring = nullptr;
  DevResult* d_init_ring = nullptr;   // the same, in device memory: the ring is re-initialised with a device-to-device copy (a 40 KB host-to-device
                                      // hipMemcpyAsync made the calling thread wait for everything queued on the stream: 6.7 ms behind 15 cfg5 batches)
  CopyJob copy;        // set while etlg_copy_decode runs etlg_decode over its synthetic frames
  DevBuf d_copy_in, d_copy_offs, d_copy_out, d_copy_out_offs;
  DevBuf d_scan;       // scratch of the record-boundary scan
  uint32_t* h_scan = nullptr;  // pinned: its 4-word result
  size_t scan_half = 0, scan_tiles_cap = 0, scan_dirty[2] = {0, 0}; int scan_cur = 0;  // double-buffered scan descriptors: bytes per buffer, dirty 8-byte words, the one the next run uses
  unsigned long long scan_reruns = 0, scan_seq = 0;  // debugging aid: batches that needed hints / the one-lane walk
  DevBuf d_ctrl_stage;   // bytes of a batch's Relation / DDL frames (k_ctrl_list gathers them)
  // ETLG_HOST_TIMES=1: wall-clock microseconds the host spends between marks of the control path, printed when the context goes
  bool host_times = false, host_times_slow = false; double host_us[12] = {0}; uint64_t host_n[12] = {0};
  size_t ctrl_stage_cap_test = 0;
  std::chrono::steady_clock::time_point host_mark;
  DevBuf d_in, d_offs, d_tag, d_emit, d_ffixed, d_fheap, d_blk32, d_blk64, d_ctrl, d_res, d_desc;
  std::vector<SideSet*> side_sets;   // every set ever built (a handful)
  SideSet* side_cur = nullptr;       // the latest upload: what last_tables / last_epochs / last_live describe
  FusedParams fq{};
  PlanParams pq{};
  uint32_t n_dev_slots = 0, n_dev_cols = 0;
  // the fixed-width plan (plan.hip): eligible tables of the current side inputs, and the back-off after a batch that did not conform
  uint32_t n_plan_tabs = 0, plan_max_row = 16;
  bool plan_covers_all = false;
  int plan_mode = 1;             // ETLG_PLAN=0 switches the plan off
  uint32_t plan_margin_pct = 4;  // ETLG_PLAN_MARGIN: LDS window per tile = 64 average frames + this margin (a tile that does not fit is read in place)
  uint32_t plan_dbg = 0;         // ETLG_PLAN_DBG: bit 0 = no LDS staging (tests of the in-place reader)
  int n_cus = 256;
  uint32_t plan_skip = 0, plan_penalty = 4, plan_streak = 0;
  bool side_dirty = true;            // table states / the shared table cache changed since the side inputs were last built
  bool last_any_sync_done = false;
  bool last_had_ctrl = false;        // the last finished batch took the control path and did hold Relation / DDL frames
  // ASYNC without the caller's no-control assertion on a stream that carries Relation / DDL frames (last_had_ctrl): the control
  // pre-pass of batch k+1 (classify, transaction scan, control list + the frames' bytes to pinned memory) runs on its own stream
  // while batch k is decoded; the host control plane of k+1 then runs — still beside k's kernels — when the next call comes in
  // (flush_deferred), and k+1's decode is enqueued behind k's with the device-side carry. One pre-pass in flight at a time.
  hipStream_t ctl_stream = nullptr;
  DevBuf d_ctl_res;                   // ring of kCtlRing pre-pass result blocks (the pre-passes chain their transaction state through them)
  static constexpr uint32_t kCtlRing = 4, kCtlListCap = 4096, kCtlStageCap = 512u << 10;
  uint32_t ctl_seq = 0;
  CtrlFrame* h_ctl_list = nullptr;    // pinned: the first kCtlListCap entries of the control list ...
  uint8_t* h_ctl_stage = nullptr;     // ... and the first kCtlStageCap gathered bytes, copied behind the pre-pass without asking for their sizes
  hipEvent_t mp_tail = nullptr; bool mp_tail_set = false;   // behind the last multi-pass launch (it shares the per-frame scratch with the pre-pass)
  int ctl_async_mode = 1;             // ETLG_CTL_ASYNC=0: control batches are decoded synchronously, as in round 2
  uint64_t cs_gen = 0;                // bumped by every rollback of the control state
  unsigned long long ctl_ahead_n = 0; // debugging aid: batches whose pre-pass ran ahead
  etlg_batch* deferred = nullptr;     // ASYNC batch whose boundary scan (no sidecar) or control pre-pass is in flight: its decode is enqueued by the next call
  ScanJob scan_job;                   // ... and that scan
  hipStream_t res_stream = nullptr;   // ASYNC batches: their result block travels to the host on this stream, so that no copy sits between two decode kernels
  hipStream_t scan_stream = nullptr;  // ASYNC batches without a sidecar: their boundary scan runs here, beside the previous batch's decode
  bool ring_h2d = false;              // ETLG_RING_H2D=1 (measurement / bisect knob): re-initialise the result ring from the host template
  uint64_t fixed_hint = 0;            // largest fixed-arena bound seen so far, with head room (setup_outputs)
  hipStream_t d2h_stream = nullptr;   // etlg_batch_download / host-output decodes: the arena of a finished batch travels here
  hipStream_t h2d_stream = nullptr;   // ASYNC batches with host input: their bytes + sidecar are uploaded here, beside the previous batch's decode
  unsigned long long staged_async = 0;
  std::vector<DevBuf*> offs_pool;     // ... into an offsets buffer the batch owns
  std::vector<std::pair<void*, size_t>> blk_dev, blk_host;  // hand-off calls (columns / RowBinary / size hints): pooled device and pinned blocks
  DevBuf d_colsel;                   // etlg_batch_columns: block counts of the row selection
  std::vector<etlg_batch*> pending;  // ASYNC batches not finished yet, in issue order
  std::vector<hipEvent_t> ev_pool;   // "result block copied back" events of finished batches
  std::vector<int32_t> last_live;      // slots whose columns d_cols currently holds
  std::vector<DevTable> last_tables;   // what d_tables / d_epochs currently hold
  std::vector<DevEpoch> last_epochs;
  bool side_valid = false;
  bool force_multipass = false;  // ETLG_FORCE_MULTIPASS=1 (tests exercise both paths)
  unsigned long long last_dbg[12] = {0};
  unsigned long long path_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int fused_kernel = -1;         // ETLG_FUSED_KERNEL: 0 k_fused/256, 1 k_fused/64, 2 k_cells, 3 k_plan whenever eligible (default: plan, else by frame size)
  uint32_t fused_dbg = 0;        // ETLG_FUSED_DBG: ablation bits for profiling only (results are wrong)
  std::vector<OutSet*> out_pool;
  DevResult* h_init = nullptr;              // pinned, constant: the cleared result block
  DevResult* h_poison = nullptr;            // pinned, constant: "this batch did not run" (fused_fail bit 3)
  std::vector<DevResult*> res_pool;         // pinned result blocks (one per in-flight batch)
  std::vector<std::pair<uint8_t*, size_t>> harena_pool;  // pinned host arenas, reused by size
  // error
  etlg_error err{};
  std::string err_detail;
  // profiling
  bool prof = false;
  std::vector<ProfRec> prof_recs;
  double prof_ms[kProfSlots] = {0};
  uint64_t prof_n[kProfSlots] = {0};
};

struct etlg_batch {
  etlg_ctx* ctx = nullptr;
  etlg_batch_view v{};
  OutSet* dev = nullptr;  // owned device arrays (OUTPUT_ON_DEVICE) — returned to the pool on free
  // host copy of the arena (etlg_batch_download / host-output decode): one pinned block from the context's pool
  uint8_t* h_arena = nullptr; size_t h_arena_cap = 0;
  std::vector<etlg_slot_desc> slot_descs;
  bool pending = false;  // ASYNC: counts not read back yet
  bool finished = false;
  int32_t rc = 0;        // result of the batch once finished (what etlg_decode / etlg_batch_sync return)
  etlg_error err{}; std::string err_detail;
  int level = 1;         // which kernels produced the result: 0 fixed-width plan, 1 generic single pass, 2 multi-pass
  bool user_no_ctrl = false, ctrl_done = false, out_dev = false, in_dev = false, scan = false, any_sync_done = false;
  size_t len = 0;
  const uint8_t* host_in = nullptr; const uint32_t* host_offs = nullptr; const uint8_t* dev_in = nullptr;
  uint64_t ctx_gen = 0;
  bool deferred = false;        // ASYNC without a sidecar: scan in flight, decode not enqueued yet (etlg_ctx::deferred)
  const uint8_t* d_in_ptr = nullptr; const uint32_t* user_offs = nullptr;
  DevBuf* scan_offs = nullptr;  // ASYNC without a sidecar: the batch's own offsets (from the context's pool)
  int plan_decided = -1;      // decode_tail: -1 not decided yet, 0 / 1 = the first attempt is the generic kernel / the fixed-width plan
  int sidx = 0;               // decode stream the batch's first attempt was enqueued on (0: etlg_ctx::stream, 1: stream2)
  bool force_rerun = false;   // a batch of the chain before this one had to be decoded again: whatever this one produced started from the wrong state
  hipEvent_t kdone = nullptr; // ASYNC: recorded behind the batch's kernels on its decode stream (the result copy waits for it)
  hipEvent_t done = nullptr;  // recorded behind the copy of the result block: syncing a batch waits for IT, not for the whole stream
  DevResult* h_res = nullptr;  // pinned, from the context's pool
  CopyJob copy;            // table-copy batch: the splitter has to run again before a multi-pass redo
  DevResult* d_res_blk = nullptr;  // this batch's result block on the device
  bool used_cells = false; // ... and it was k_cells
  bool used_fused = false; // the fused kernel produced this batch; errors re-run the multi-pass kernels
  DecParams params{};
  SideSet* side = nullptr;   // the side inputs its kernels read (released when the batch is finished)
  size_t n_slots_view = ~(size_t)0;   // schema slots the batch's view lists (fill_view_common)
  // what sync needs to finish the batch
  int32_t host_err_code = 0; int64_t host_err_frame = -1; uint32_t host_err_rank = 0;
  std::vector<EpochRec> eps_saved; // epochs of the batch's own control frames (a multi-pass redo needs the same side inputs)
  std::vector<CtrlFrame> ctrl;     // processed control frames (for rollback replay)
  std::vector<std::vector<uint8_t>> ctrl_raw;  // their bytes, same order (the input may be device-resident or come without a sidecar)
  ControlState snapshot;           // control state before the batch
  bool have_snapshot = false;
  uint64_t snap_gen = 0;           // etlg_ctx::cs_gen when the snapshot was taken
  // pipelined control path (etlg_ctx::ctl_stream)
  bool defer_ctl = false;          // deferred because its control pre-pass is in flight (not a boundary scan)
  bool ctl_started = false;        // pre-pass enqueued ahead (ctl_params / h_ctl / ctl_ev are valid)
  bool ctl_async = false;          // took the pipelined control path: a forced re-run redoes its control pass
  size_t nframes_in = 0;
  DecParams ctl_params{};
  DevResult* h_ctl = nullptr;      // pinned copy of the pre-pass result block
  hipEvent_t ctl_ev = nullptr;     // behind the pre-pass and its copies
  // ASYNC with host input: the bytes and the sidecar travel to a device block of the batch's own on the copy stream
  void* stage_blk = nullptr; size_t stage_cap = 0;
  hipEvent_t h2d_done = nullptr;   // behind the two copies (the decode streams wait for it on the device)
};

struct HandoffBlocks {  // two device blocks (+ one pinned block when downloaded), taken from / returned to the context's pool
  etlg_ctx* ctx = nullptr; uint64_t ctx_gen = 0;
  void* d_a = nullptr; void* d_b = nullptr; void* d_c = nullptr; uint8_t* h = nullptr;
  size_t cap_a = 0, cap_b = 0, cap_c = 0, cap_h = 0;
};
struct etlg_columns {  // etlg_batch_columns
  etlg_columns_view v{};
  std::vector<etlg_column> cols;
  HandoffBlocks m;
};

struct etlg_rowbinary {
  etlg_rowbinary_view v{};
  HandoffBlocks m;
};

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

// ----------------------------------------------------------------- slots
int32_t make_slot(etlg_ctx* c, const SchemaPtr& sch, const std::vector<uint8_t>& repl, const std::vector<uint8_t>& ident) {
  auto s = std::make_unique<SlotHost>();
  uint32_t nid = 0;
  for (size_t i = 0; i < sch->cols.size(); i++) {
    if (repl[i] != 1) continue;
    etlg_slot_col sc{};
    sc.type_oid = sch->cols[i].type_oid;
    sc.stored_index = (uint16_t)i;
    sc.type_class = (uint8_t)type_class(sc.type_oid);
    sc.nullable = sch->cols[i].nullable;
    sc.identity = ident[i] == 1;
    sc.key_index = sc.identity ? (uint16_t)nid++ : (uint16_t)0xFFFF;
    s->cols.push_back(sc);
  }
  {  // identity type over the stored columns, as the reference infers it from the two masks
    bool has = false, m_pk = true, m_full = true;
    for (size_t i = 0; i < sch->cols.size(); i++) {
      const bool r = repl[i] == 1, id = ident[i] == 1;
      has |= id;
      if (id != (r && sch->cols[i].pk)) m_pk = false;
      if (id != r) m_full = false;
    }
    s->identity_type = !has ? 0 : m_pk ? 1 : m_full ? 2 : 3;
  }
  const uint32_t n = (uint32_t)s->cols.size();
  etlg_slot_desc& d = s->desc;
  d = etlg_slot_desc{};
  d.table_id = sch->table_id; d.n_stored = (uint32_t)sch->cols.size(); d.snapshot_lsn = sch->snapshot;
  d.n_cols = n; d.n_ident = nid;
  d.state_bytes_full = 4 * ((n + 15) / 16);
  d.state_bytes_key = 4 * ((nid + 15) / 16);
  uint32_t off = d.state_bytes_full, koff = d.state_bytes_key;
  for (auto& sc : s->cols) {
    const uint32_t sb = slot_bytes(sc.type_class);
    sc.off_full = (uint16_t)off; off += sb;
    if (sc.identity) { sc.off_key = (uint16_t)koff; koff += sb; }
  }
  d.row_bytes_full = off; d.row_bytes_key = koff;
  d.cols = s->cols.data();
  c->slots.push_back(std::move(s));
  c->cs.n_slots = c->slots.size();
  c->slots_dirty = true;
  return (int32_t)c->slots.size() - 1;
}

SchemaPtr get_at_or_before(const ControlState& cs, uint32_t table_id, uint64_t snap) {
  auto it = cs.store.find(table_id);  // store/schema/table.rs:61-71
  if (it == cs.store.end()) return nullptr;
  auto ub = it->second.upper_bound(snap);
  if (ub == it->second.begin()) return nullptr;
  --ub;
  return ub->second;
}

bool should_apply(const etlg_ctx* c, uint32_t table_id, uint64_t final_lsn) {  // apply.rs:2836-2867, 3514-3519
  if (c->worker == ETLG_WORKER_TABLE_SYNC) return c->sync_table == table_id;
  auto it = c->states.find(table_id);
  if (it == c->states.end()) return false;
  if (it->second.kind == ETLG_TS_READY) return true;
  if (it->second.kind == ETLG_TS_SYNC_DONE) return it->second.lsn <= final_lsn;
  return false;
}

// --------------------------------------------------------- tiny byte reader
struct Rd {
  const uint8_t* p; size_t n; size_t i = 0; bool ok = true;
  bool need(size_t k) { if (!ok || n - i < k) { ok = false; return false; } return true; }
  uint8_t u8() { return need(1) ? p[i++] : 0; }
  uint16_t u16() { if (!need(2)) return 0; uint16_t v = (uint16_t)(p[i] << 8 | p[i + 1]); i += 2; return v; }
  uint32_t u32() { if (!need(4)) return 0; uint32_t v = (uint32_t)p[i] << 24 | (uint32_t)p[i + 1] << 16 | (uint32_t)p[i + 2] << 8 | p[i + 3]; i += 4; return v; }
  uint64_t u64() { uint64_t h = u32(); return h << 32 | u32(); }
  std::string_view cstr() {
    if (!ok) return {};
    const void* z = memchr(p + i, 0, n - i);
    if (!z) { ok = false; return {}; }
    size_t len = (const uint8_t*)z - (p + i);
    std::string_view r((const char*)p + i, len);
    i += len + 1;
    return r;
  }
};

bool utf8_ok(std::string_view s) {  // String::from_utf8 in the cstr accessors
  const uint8_t* p = (const uint8_t*)s.data(); size_t n = s.size(), i = 0;
  while (i < n) {
    uint8_t c = p[i];
    if (c < 0x80) { i++; continue; }
    size_t need; uint8_t lo = 0x80, hi = 0xBF;
    if (c >= 0xC2 && c <= 0xDF) need = 1;
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    else return false;
    if (n - i <= need) return false;
    if (p[i + 1] < lo || p[i + 1] > hi) return false;
    for (size_t k = 2; k <= need; k++) if ((p[i + k] & 0xC0) != 0x80) return false;
    i += need + 1;
  }
  return true;
}

// ------------------------------------------------- DDL message (JSON) reader
// SchemaChangeMessage (codec/event.rs:37-56, 96-106, 182-196) with serde's
// struct rules: unknown fields skipped, missing / duplicate / mistyped fields
// are errors, null only for Option. A pull parser: no DOM is built.
struct Json {
  const char* p; const char* e; int depth = 0; bool bad = false;
  void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
  bool eat(char c) { ws(); if (p < e && *p == c) { p++; return true; } return false; }
  bool peek(char c) { ws(); return p < e && *p == c; }
  bool fail() { bad = true; return false; }
  static void utf8_push(std::string& o, unsigned cp) {
    if (cp < 0x80) o += (char)cp;
    else if (cp < 0x800) { o += (char)(0xC0 | cp >> 6); o += (char)(0x80 | (cp & 63)); }
    else if (cp < 0x10000) { o += (char)(0xE0 | cp >> 12); o += (char)(0x80 | ((cp >> 6) & 63)); o += (char)(0x80 | (cp & 63)); }
    else { o += (char)(0xF0 | cp >> 18); o += (char)(0x80 | ((cp >> 12) & 63)); o += (char)(0x80 | ((cp >> 6) & 63)); o += (char)(0x80 | (cp & 63)); }
  }
  bool hex4(unsigned& v) {
    if (e - p < 4) return fail();
    v = 0;
    for (int k = 0; k < 4; k++) {
      char h = *p++; int d = h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : -1;
      if (d < 0) return fail();
      v = v * 16 + (unsigned)d;
    }
    return true;
  }
  bool string(std::string* out) {
    ws();
    if (p >= e || *p != '"') return fail();
    p++;
    while (p < e) {
      unsigned char c = (unsigned char)*p++;
      if (c == '"') return true;
      if (c < 0x20) return fail();
      if (c != '\\') { if (out) *out += (char)c; continue; }
      if (p >= e) return fail();
      char x = *p++;
      char plain = 0;
      switch (x) {
        case '"': plain = '"'; break; case '\\': plain = '\\'; break; case '/': plain = '/'; break;
        case 'b': plain = '\b'; break; case 'f': plain = '\f'; break; case 'n': plain = '\n'; break;
        case 'r': plain = '\r'; break; case 't': plain = '\t'; break;
        case 'u': {
          unsigned v;
          if (!hex4(v)) return false;
          if (v >= 0xDC00 && v <= 0xDFFF) return fail();
          if (v >= 0xD800 && v <= 0xDBFF) {
            if (e - p < 2 || p[0] != '\\' || p[1] != 'u') return fail();
            p += 2;
            unsigned w;
            if (!hex4(w)) return false;
            if (w < 0xDC00 || w > 0xDFFF) return fail();
            v = 0x10000 + ((v - 0xD800) << 10) + (w - 0xDC00);
          }
          if (out) utf8_push(*out, v);
          continue;
        }
        default: return fail();
      }
      if (out) *out += plain;
    }
    return fail();
  }
  // number literal -> integer in [lo, hi]; fractions / exponents are type errors
  bool integer(int64_t lo, uint64_t hi, int64_t& out) {
    ws();
    const char* s = p;
    bool neg = false;
    if (p < e && *p == '-') { neg = true; p++; }
    if (p >= e || *p < '0' || *p > '9') return fail();
    if (*p == '0') p++; else while (p < e && *p >= '0' && *p <= '9') p++;
    if (p < e && (*p == '.' || *p == 'e' || *p == 'E')) return fail();
    unsigned __int128 mag = 0;
    for (const char* q = s + (neg ? 1 : 0); q < p; q++) { mag = mag * 10 + (unsigned)(*q - '0'); if (mag > ((unsigned __int128)1 << 70)) return fail(); }
    if (neg) { if (mag > (unsigned __int128)(-(lo + 1)) + 1) return fail(); out = (int64_t)(-(__int128)mag); }
    else { if (mag > hi) return fail(); out = (int64_t)(uint64_t)mag; }
    return true;
  }
  bool skip_number() {
    if (p < e && *p == '-') p++;
    if (p >= e) return fail();
    if (*p == '0') p++;
    else if (*p >= '1' && *p <= '9') while (p < e && *p >= '0' && *p <= '9') p++;
    else return fail();
    if (p < e && *p == '.') { p++; if (p >= e || *p < '0' || *p > '9') return fail(); while (p < e && *p >= '0' && *p <= '9') p++; }
    if (p < e && (*p == 'e' || *p == 'E')) { p++; if (p < e && (*p == '+' || *p == '-')) p++; if (p >= e || *p < '0' || *p > '9') return fail(); while (p < e && *p >= '0' && *p <= '9') p++; }
    return true;
  }
  bool lit(const char* s) { size_t n = strlen(s); if ((size_t)(e - p) < n || memcmp(p, s, n)) return fail(); p += n; return true; }
  bool skip_value() {
    ws();
    if (p >= e) return fail();
    switch (*p) {
      case '{': {
        if (++depth > 128) return fail();
        p++;
        if (eat('}')) { depth--; return true; }
        for (;;) {
          if (!string(nullptr) || !eat(':') || !skip_value()) return fail();
          if (eat(',')) continue;
          if (eat('}')) { depth--; return true; }
          return fail();
        }
      }
      case '[': {
        if (++depth > 128) return fail();
        p++;
        if (eat(']')) { depth--; return true; }
        for (;;) {
          if (!skip_value()) return false;
          if (eat(',')) continue;
          if (eat(']')) { depth--; return true; }
          return fail();
        }
      }
      case '"': return string(nullptr);
      case 't': return lit("true");
      case 'f': return lit("false");
      case 'n': return lit("null");
      default: return skip_number();
    }
  }
  // iterate an object: calls f(key) for each member; f must consume the value
  template <class F>
  bool object(F f) {
    if (!eat('{')) return fail();
    if (++depth > 128) return fail();
    if (eat('}')) { depth--; return true; }
    for (;;) {
      std::string key;
      if (!string(&key) || !eat(':')) return fail();
      if (!f(key)) return fail();
      if (eat(',')) continue;
      if (eat('}')) { depth--; return true; }
      return fail();
    }
  }
  template <class F>
  bool array(F f) {
    if (!eat('[')) return fail();
    if (++depth > 128) return fail();
    if (eat(']')) { depth--; return true; }
    for (;;) {
      if (!f()) return fail();
      if (eat(',')) continue;
      if (eat(']')) { depth--; return true; }
      return fail();
    }
  }
  bool boolean(bool& b) { ws(); if (p < e && *p == 't') { b = true; return lit("true"); } if (p < e && *p == 'f') { b = false; return lit("false"); } return fail(); }
};

bool parse_ddl(std::string_view content, uint64_t snapshot, std::shared_ptr<StoredSchema>& out) {
  Json j{content.data(), content.data() + content.size()};
  auto sch = std::make_shared<StoredSchema>();
  sch->snapshot = snapshot;
  int seen_tag = 0, seen_nsp = 0, seen_rel = 0, seen_oid = 0, seen_ident = 0, seen_cols = 0;
  std::vector<int32_t> pks;
  bool ok = j.object([&](const std::string& k) {
    if (k == "command_tag") { seen_tag++; return j.string(nullptr); }
    if (k == "nspname") { seen_nsp++; return j.string(&sch->nsp); }
    if (k == "relname") { seen_rel++; return j.string(&sch->name); }
    if (k == "oid") { seen_oid++; int64_t v; if (!j.integer(INT64_MIN, INT64_MAX, v)) return false; sch->table_id = (uint32_t)v; return true; }
    if (k == "identity") {
      seen_ident++;
      int s_pk = 0, s_rid = 0, s_rix = 0;
      bool o = j.object([&](const std::string& k2) {
        if (k2 == "primary_key_attnums") { s_pk++; return j.array([&] { int64_t v; if (!j.integer(INT32_MIN, INT32_MAX, v)) return false; pks.push_back((int32_t)v); return true; }); }
        if (k2 == "relreplident") { s_rid++; return j.string(nullptr); }
        if (k2 == "replica_identity_index_attnums") { s_rix++; return j.array([&] { int64_t v; return j.integer(INT32_MIN, INT32_MAX, v); }); }
        return j.skip_value();
      });
      return o && s_pk == 1 && s_rid == 1 && s_rix == 1;
    }
    if (k == "columns") {
      seen_cols++;
      return j.array([&] {
        StoredCol c{};
        int s_n = 0, s_t = 0, s_m = 0, s_a = 0, s_nn = 0, s_d = 0;
        bool notnull = false;
        bool o = j.object([&](const std::string& k2) {
          int64_t v;
          if (k2 == "attname") { s_n++; return j.string(&c.name); }
          if (k2 == "atttypid") { s_t++; if (!j.integer(0, UINT32_MAX, v)) return false; c.type_oid = (uint32_t)v; return true; }
          if (k2 == "atttypmod") { s_m++; if (!j.integer(INT32_MIN, INT32_MAX, v)) return false; c.typmod = (int32_t)v; return true; }
          if (k2 == "attnum") { s_a++; if (!j.integer(INT32_MIN, INT32_MAX, v)) return false; c.attnum = (int32_t)v; return true; }
          if (k2 == "attnotnull") { s_nn++; return j.boolean(notnull); }
          if (k2 == "default_expression") { s_d++; if (j.peek('n')) return j.lit("null"); return j.string(nullptr); }
          return j.skip_value();
        });
        if (!o || s_n != 1 || s_t != 1 || s_m != 1 || s_a != 1 || s_nn != 1 || s_d > 1) return false;
        c.nullable = !notnull;
        sch->cols.push_back(std::move(c));
        return true;
      });
    }
    return j.skip_value();
  });
  if (!ok || j.bad) return false;
  j.ws();
  if (j.p != j.e) return false;
  if (seen_tag != 1 || seen_nsp != 1 || seen_rel != 1 || seen_oid != 1 || seen_ident != 1 || seen_cols != 1) return false;
  for (auto& c : sch->cols) c.pk = std::find(pks.begin(), pks.end(), c.attnum) != pks.end();
  std::stable_sort(sch->cols.begin(), sch->cols.end(), [](const StoredCol& a, const StoredCol& b) { return a.attnum < b.attnum; });
  out = std::move(sch);
  return true;
}

// ----------------------------------------------------- control-plane frames

// handle_relation_message (apply.rs:2363-2440) for one R frame.
HostErr handle_relation(etlg_ctx* c, const CtrlFrame& cf, const uint8_t* body, size_t n, std::vector<EpochRec>& eps) {
  Rd r{body, n};
  const uint32_t rel_id = r.u32();
  (void)r.cstr(); (void)r.cstr();
  const uint8_t replident = r.u8();
  if (!r.ok || (replident != 'd' && replident != 'n' && replident != 'f' && replident != 'i')) return {ETLG_E_WIRE, RK_WIRE};
  const int16_t nc = (int16_t)r.u16();
  if (!r.ok || nc < 0) return {ETLG_E_WIRE, RK_WIRE};
  struct RC { int8_t flags; std::string_view name; };
  std::vector<RC> rcs;
  for (int k = 0; k < nc; k++) {
    RC x; x.flags = (int8_t)r.u8(); x.name = r.cstr(); (void)r.u32(); (void)r.u32();
    if (!r.ok) return {ETLG_E_WIRE, RK_WIRE};
    rcs.push_back(x);
  }
  if (!cf.in_txn) return {ETLG_E_TXN_STATE, RK_TXN};
  if (!should_apply(c, rel_id, cf.final_lsn)) return {};
  // parse_replicated_column_names / parse_replica_identity_column_names (codec/event.rs:352-396)
  std::set<std::string> repl, ident;
  for (auto& x : rcs) { if (!utf8_ok(x.name)) return {ETLG_E_IO, RK_SCHEMA}; repl.emplace(x.name); }
  for (auto& x : rcs) if (replident == 'f' || (x.flags & 1) == 1) ident.emplace(x.name);
  auto cit = c->cs.cache.find(rel_id);
  const bool used_bootstrap = cit == c->cs.cache.end();
  const uint64_t snap = used_bootstrap ? c->bootstrap : cit->second.snapshot;
  SchemaPtr sch = get_at_or_before(c->cs, rel_id, snap);  // get_table_schema_for_relation, apply.rs:3643-3697
  if (!sch) return {ETLG_E_SCHEMA_NOT_FOUND, RK_SCHEMA};
  if (used_bootstrap) { if (sch->snapshot > snap) return {ETLG_E_BOOTSTRAP_SNAPSHOT, RK_SCHEMA}; }
  else if (sch->snapshot != snap) return {ETLG_E_SNAPSHOT_MISMATCH, RK_SCHEMA};
  // ReplicationMask::try_build / IdentityMask::try_build (schema.rs:30-61, 99-129, 220-227)
  std::set<std::string_view> have;
  for (auto& sc : sch->cols) have.insert(sc.name);
  for (auto& nme : repl) if (!have.count(nme)) return {ETLG_E_UNKNOWN_COLUMNS, RK_SCHEMA};
  for (auto& nme : ident) if (!have.count(nme)) return {ETLG_E_UNKNOWN_COLUMNS, RK_SCHEMA};
  std::vector<uint8_t> rm, im;
  for (auto& sc : sch->cols) { rm.push_back(repl.count(sc.name) ? 1 : 0); im.push_back(ident.count(sc.name) ? 1 : 0); }
  const int32_t slot = make_slot(c, sch, rm, im);
  c->cs.cache[rel_id] = CacheEntry{2, sch->snapshot, slot};  // note_ready
  c->side_dirty = true;
  eps.push_back({rel_id, DevEpoch{cf.frame, 2, slot, 1}});
  return {};
}

// handle_message (apply.rs:2160-2276) for one M frame; `wal_start` = snapshot id.
HostErr handle_ddl(etlg_ctx* c, const CtrlFrame& cf, uint64_t wal_start, const uint8_t* body, size_t n, std::vector<EpochRec>& eps) {
  Rd r{body, n};
  (void)r.u8(); (void)r.u64();
  std::string_view prefix = r.cstr();
  const int32_t len = (int32_t)r.u32();
  if (!r.ok || len < 0 || !r.need((size_t)len)) return {ETLG_E_WIRE, RK_WIRE};
  std::string_view content((const char*)body + r.i, (size_t)len);
  if (!utf8_ok(prefix)) return {ETLG_E_IO, RK_SCHEMA};
  if (prefix != "supabase_etl_ddl") return {};  // codec/event.rs:28
  if (!cf.in_txn) return {ETLG_E_TXN_STATE, RK_TXN};
  if (!utf8_ok(content)) return {ETLG_E_IO, RK_SCHEMA};
  std::shared_ptr<StoredSchema> sch;
  if (!parse_ddl(content, wal_start, sch)) return {ETLG_E_DDL_PARSE, RK_SCHEMA};
  if (!should_apply(c, sch->table_id, cf.final_lsn)) return {};
  const uint32_t tid = sch->table_id;
  c->cs.store[tid][sch->snapshot] = sch;                       // store_table_schema
  c->cs.cache[tid] = CacheEntry{1, wal_start, -1};             // note_waiting_for_relation
  c->side_dirty = true;
  eps.push_back({tid, DevEpoch{cf.frame, 1, -1, 0}});
  return {};
}

// The schema slots as the kernels read them. Slot ids are stable for the life of the context (the arenas name them), but a stream
// that changes schemas often leaves most of them dead: the device table holds only the slots in `live` (what the table cache, this
// batch's epochs or a table copy can reach), addressed by their position in it (DevSlot.host_id names the arena's id). The side
// tables then stay small enough for the LDS copy the single-pass kernels want (k_cells needs it) after thousands of DDL messages.
void build_slots(etlg_ctx* c, const std::vector<int32_t>& live, std::vector<DevSlot>& ds, std::vector<DevCol>& dc) {
  for (int32_t li : live) {   // ascending host ids: the device index of a slot is its position in `live`
    if (li < 0 || (size_t)li >= c->slots.size()) { DevSlot dead{}; dead.cols_base = (uint32_t)dc.size(); dead.host_id = (uint32_t)li; ds.push_back(dead); continue; }
    auto& s = c->slots[(size_t)li];
    DevSlot d{};
    d.host_id = (uint32_t)li;
    d.n_cols = s->desc.n_cols; d.n_ident = s->desc.n_ident; d.row_full = s->desc.row_bytes_full; d.row_key = s->desc.row_bytes_key;
    d.st_full = s->desc.state_bytes_full; d.st_key = s->desc.state_bytes_key; d.cols_base = (uint32_t)dc.size();
    {  // which columns can reach the heap, and for which of them the byte count depends on the text (DevSlot.has_var)
      const bool narrow = s->cols.size() <= 16;
      uint32_t ci = 0;
      for (auto& sc : s->cols) {
        const int32_t k = sc.type_class;
        // fixed-width classes never reach the heap — the temporal ones included: their rare non-ISO shapes are decoded on the device too
        // (chrono_fallback, codec.hip.h). A float reaches it only as a DEFERRED text; numeric / bytea entries are sized from the text
        const bool heap = !(k == ETLG_TC_BOOL || k == ETLG_TC_I16 || k == ETLG_TC_I32 || k == ETLG_TC_I64 || k == ETLG_TC_U32 || k == ETLG_TC_UUID ||
                            k == ETLG_TC_DATE || k == ETLG_TC_TIME || k == ETLG_TC_TIMETZ || k == ETLG_TC_TIMESTAMP || k == ETLG_TC_TIMESTAMPTZ);
        const bool scan = heap && !(k == ETLG_TC_STRING || k == ETLG_TC_JSON || k == ETLG_TC_ARRAY);
        if (!narrow) { if (heap) d.has_var = 0xFFFFFFFFu; }
        else {
          if (heap) d.has_var |= 1u << ci;
          if (scan) d.has_var |= 1u << (16 + ci);
          if (sc.identity) {
            d.ident_mask |= 1u << ci;
            if (sc.key_index < 16) { if (heap) d.key_masks |= 1u << sc.key_index; if (scan) d.key_masks |= 1u << (16 + sc.key_index); }
          }
        }
        ci++;
      }
    }
    for (auto& sc : s->cols) {
      DevCol x{};
      x.cls = sc.type_class; x.nullable = sc.nullable; x.identity = sc.identity; x.off_full = sc.off_full; x.off_key = sc.off_key; x.key_index = sc.key_index;
      dc.push_back(x);
    }
    ds.push_back(d);
  }
}

void side_release(etlg_batch* b) { if (b->side) { b->side->users--; b->side = nullptr; } }
void side_use(etlg_batch* b, SideSet* ss) { if (b->side == ss) return; side_release(b); b->side = ss; ss->users++; }

void launch_raw(etlg_ctx* c, int which, const DecParams& p) {
  if (which == kFused) etlg_k_launch_fused((int)c->fq.blk, &p, &c->fq, c->stream);
  else if (which == kPlan) etlg_k_launch_plan(&p, &c->pq, c->stream);
  else if (which == kCells) etlg_k_launch_cells(&p, &c->fq, c->stream);
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
  etlg_k_launch_copy(j.d_rows, j.d_row_offs, j.nrows, j.rows_len, j.ncols, j.rel_id, j.d_out, j.d_out_offs, j.lds, &p, c->stream);
  if (c->prof) { (void)hipEventRecord(r.b, c->stream); c->prof_recs.push_back(r); }
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
int32_t drain_pending(etlg_ctx* c);
int32_t download_batch(etlg_ctx* c, etlg_batch* b);
struct BatchGuard {  // an etlg_decode that fails half way returns what the batch took from the context's pools
  etlg_batch* b;
  ~BatchGuard() { if (b) etlg_batch_free(b); }
};
int32_t build_side_inputs(etlg_ctx* c, etlg_batch* b, const std::vector<EpochRec>& eps);
int32_t setup_outputs(etlg_ctx* c, etlg_batch* b);
int32_t setup_scratch(etlg_ctx* c, DecParams& p);
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
  (void)etlg_k_fused_set_lds();
  (void)etlg_k_cells_set_lds();
  (void)etlg_k_copy_set_lds();
  { const char* fm = getenv("ETLG_FORCE_MULTIPASS"); c->force_multipass = fm && fm[0] == '1'; }
  { const char* e = getenv("ETLG_RING_H2D"); c->ring_h2d = e && e[0] == '1'; }
  { const char* ht = getenv("ETLG_HOST_TIMES"); c->host_times = ht && (ht[0] == '1' || ht[0] == '2'); c->host_times_slow = ht && ht[0] == '2'; }
  { const char* sc = getenv("ETLG_CTRL_STAGE_CAP"); c->ctrl_stage_cap_test = sc ? (size_t)atol(sc) : 0; }
  { const char* fd = getenv("ETLG_FUSED_DBG"); c->fused_dbg = fd ? (uint32_t)atoi(fd) : 0; }
  { const char* fk = getenv("ETLG_FUSED_KERNEL"); c->fused_kernel = fk ? atoi(fk) : -1; }
  clear_error(c);
  (void)etlg_k_plan_set_lds();
  if (const char* pm = getenv("ETLG_PLAN")) c->plan_mode = atoi(pm);
  if (const char* pm = getenv("ETLG_PLAN_MARGIN")) c->plan_margin_pct = (uint32_t)atoi(pm);
  if (const char* pm = getenv("ETLG_PLAN_DBG")) c->plan_dbg = (uint32_t)atoi(pm);
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
  if (c->h_ctl_list) (void)hipHostFree(c->h_ctl_list);
  if (c->h_ctl_stage) (void)hipHostFree(c->h_ctl_stage);
  c->d_ctl_res.release();
  if (c->scan_stream) (void)hipStreamDestroy(c->scan_stream);
  if (c->h2d_stream) (void)hipStreamDestroy(c->h2d_stream);
  if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
  if (c->res_stream) (void)hipStreamDestroy(c->res_stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
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
  for (int i = 0; i < kProfSlots && k < cap; i++) { out[k].name = i == kPlan ? "k_plan" : i == kFused ? "k_fused" : i == kCells ? "k_cells" : i == kBounds ? "k_bounds" : i == kCopy ? "k_copy_frames" : etlg_k_name(i); out[k].launches = c->prof_n[i]; out[k].total_ms = c->prof_ms[i]; k++; }
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
  const size_t ntiles = (len + tb - 1) / tb, ngroups = (ntiles + 63) / 64;
  // scratch: two descriptor buffers (result block of 64 bytes + look-back words) | hints. Each run zeroes the buffer the next
  // run will use, the result travels through pinned host memory: a scan is ONE kernel on the stream, no memset, no copy.
  const size_t words = 8 + ntiles + ngroups;                       // 8-byte words of one descriptor buffer that this run dirties
  if (!c->h_scan) { hipError_t e = hipHostMalloc((void**)&c->h_scan, 64); if (e != hipSuccess) return e; }
  if (ntiles > c->scan_tiles_cap) {  // (re)allocation: the layout is by capacity, everything is initialised once
    const size_t tcap = ntiles + ntiles / 4 + 64;
    const size_t half = ((8 + tcap + tcap / 64 + 2) * 8 + 63) & ~(size_t)63;
    const size_t need = 2 * half + ((tcap * 4 + 63) & ~(size_t)63) + 64;
    hipError_t e = c->d_scan.ensure(need); if (e != hipSuccess) return e;
    e = hipMemsetAsync(c->d_scan.p, 0, 2 * half, s); if (e != hipSuccess) return e;
    e = hipMemsetAsync((uint8_t*)c->d_scan.p + 2 * half, 0xFF, tcap * 4, s); if (e != hipSuccess) return e;
    c->scan_tiles_cap = tcap; c->scan_half = half; c->scan_cur = 0; c->scan_dirty[0] = c->scan_dirty[1] = 0;
  }
  const size_t half = c->scan_half;
  uint8_t* base = (uint8_t*)c->d_scan.p;
  hipError_t e = j.offs->ensure((j.cap + 2) * 4); if (e != hipSuccess) return e;
  j.cur = base + (size_t)c->scan_cur * half;
  uint8_t* oth = base + (size_t)(c->scan_cur ^ 1) * half;
  c->h_scan[0] = 0; c->h_scan[1] = 0;
  const bool dbg = getenv("ETLG_SCAN_DBG") != nullptr;
  ProfRec r; r.which = kBounds;
  if (c->prof) { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); (void)hipEventRecord(r.a, s); }
  etlg_k_launch_bounds(j.d_in, len, (uint32_t*)j.offs->p, (uint32_t)std::min<size_t>(j.cap + 2, 0xFFFFFFFFu), j.cur, oth, (uint32_t)c->scan_dirty[c->scan_cur ^ 1],
                       (uint32_t*)(base + 2 * half), c->h_scan, (sequential ? 1 : 0) | (dbg ? 2 : 0), s);
  if (c->prof) { (void)hipEventRecord(r.b, s); c->prof_recs.push_back(r); }
  c->scan_dirty[c->scan_cur] = words; c->scan_dirty[c->scan_cur ^ 1] = 0;
  c->scan_cur ^= 1;
  return hipSuccess;
}

hipError_t scan_begin(etlg_ctx* c, ScanJob& j, const uint8_t* d_in, size_t len, hipStream_t s, DevBuf& offs) {
  j = ScanJob{};
  j.d_in = d_in; j.len = len; j.s = s; j.offs = &offs;
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
  const bool dbg = getenv("ETLG_SCAN_DBG") != nullptr;
  bool used_hints = false, hints_set = false;
  hipError_t rc = hipSuccess;
  for (int run = 0;; run++) {
    const bool sequential = run >= 4;
    hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) return e;
    uint32_t nf = c->h_scan[0], flags = 0, nbad = 0;
    if (c->h_scan[1] || dbg) {  // some tile failed: the details are in the device result block
      uint32_t res[16];
      e = hipMemcpy(res, j.cur, 64, hipMemcpyDeviceToHost); if (e != hipSuccess) return e;
      nf = res[0]; flags = res[1]; nbad = res[2];
      if (dbg) { fprintf(stderr, "k_bounds tiles %zu phase cycles/tile:", ntiles); for (int k = 4; k < 11; k++) fprintf(stderr, " %u", (unsigned)(res[k] / std::max<size_t>(1, ntiles / 64))); fprintf(stderr, "\n"); }
      if (nbad) hints_set = true;
    }
    bool again_same = false;
    if (flags & 2u) {  // offsets buffer too small
      if (j.cap >= len / 5 + 2) { rc = hipErrorOutOfMemory; break; }
      j.cap = len / 5 + 2;
      run--; again_same = true;
    } else if (sequential || (!(flags & 1u) && nbad == 0)) {
      if (used_hints) c->scan_reruns++;
      if (sequential) c->scan_seq++;
      *nframes_out = nf;
      break;
    } else {
      used_hints = true;  // some tiles guessed wrong (their hints are set now), or a spin gave up: run again
    }
    e = scan_launch(c, j, again_same ? sequential : run + 1 >= 4); if (e != hipSuccess) return e;
  }
  if (hints_set) { const hipError_t e = hipMemsetAsync((uint8_t*)c->d_scan.p + 2 * c->scan_half, 0xFF, ntiles * 4, s); if (e != hipSuccess) return e; }   // rare: leave the hints clean for the next scan
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
  // scratch: hdr (2 words) | list | span (2 per frame) | keep frames (3 per frame) | lens | out offsets
  ScratchBlk blk{c};
  HIPCHK(c, blk_take(c, (size_t)kCap * 4 * 10 + 256, false, &blk.p, &blk.cap));
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

int32_t etlg_copy_decode(etlg_ctx* c, int32_t schema_slot, const uint8_t* buf, size_t len, const uint32_t* row_offsets, size_t nrows,
                         uint32_t flags, etlg_batch** out) {
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  if (!c || !out || !row_offsets) return ETLG_InvalidArgument;
  *out = nullptr;
  // ASYNC batches still in flight finish first: finish_batch writes the stream state they leave into the context, and the
  // rows below decode inside a virtual transaction that must neither see that state nor be overwritten by it
  { const int32_t rc = drain_pending(c); if (rc != ETLG_OK) return rc; }
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
  if (in_dev) { j.d_rows = buf; j.d_row_offs = row_offsets; }
  else {
    HIPCHK(c, c->d_copy_in.ensure(len + 64)); HIPCHK(c, c->d_copy_offs.ensure((nrows + 1) * 4));
    if (len) HIPCHK(c, hipMemcpyAsync(c->d_copy_in.p, buf, len, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_copy_offs.p, row_offsets, (nrows + 1) * 4, hipMemcpyHostToDevice, s));
    j.d_rows = (const uint8_t*)c->d_copy_in.p; j.d_row_offs = (const uint32_t*)c->d_copy_offs.p;
  }
  HIPCHK(c, c->d_copy_out.ensure(syn_len + 64)); HIPCHK(c, c->d_copy_out_offs.ensure((nrows + 1) * 4));
  j.d_out = (uint8_t*)c->d_copy_out.p; j.d_out_offs = (uint32_t*)c->d_copy_out_offs.p;
  if (nrows == 0) HIPCHK(c, hipMemsetAsync(j.d_out_offs, 0, 4, s));
  const uint64_t avg = nrows ? (len + nrows - 1) / nrows : 0;
  j.lds = (uint32_t)std::min<uint64_t>(((256 * avg * 9 / 8 + 1024) + 255) & ~255ull, 150 * 1024);
  { const char* e = getenv("ETLG_COPY_LDS"); if (e) j.lds = (uint32_t)atoi(e); }   // measurement knob: 0 = rows read in place (no staging window, more waves per CU)
  // the rows decode inside a virtual transaction of their own; the context's stream state is left alone
  const bool sv_in = c->in_txn; const uint64_t sv_lsn = c->final_lsn, sv_ord = c->next_ord;
  c->in_txn = true; c->final_lsn = 0; c->next_ord = 0;
  c->copy = j;
  const int32_t rc = etlg_decode(c, j.d_out, (size_t)syn_len, j.d_out_offs, nrows,
                                 (flags & ETLG_F_OUTPUT_ON_DEVICE) | ETLG_F_INPUT_ON_DEVICE | ETLG_F_NO_CONTROL, out);
  c->copy = CopyJob{};
  c->in_txn = sv_in; c->final_lsn = sv_lsn; c->next_ord = sv_ord;
  if (*out) {
    // TableCopyPayloadMetadata: the bytes of the rows that were decoded
    etlg_batch* b = *out;
    const uint64_t done = b->v.n_frames;
    uint32_t o[2] = {0, 0};
    if (in_dev) {
      (void)hipMemcpy(&o[0], row_offsets, 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(&o[1], row_offsets + done, 4, hipMemcpyDeviceToHost);
    } else { o[0] = row_offsets[0]; o[1] = row_offsets[done]; }
    b->v.payload_bytes[0] = o[1] - o[0]; b->v.payload_bytes[1] = 0; b->v.payload_bytes[2] = 0;
  }
  return rc;
}

int32_t etlg_decode(etlg_ctx* c, const uint8_t* buf, size_t len, const uint32_t* frame_offsets, size_t nframes, uint32_t flags, etlg_batch** out) {
  if (!c || !out) return ETLG_InvalidArgument;
  SlowScope slow_scope_decode(c, "etlg_decode");
  *out = nullptr;
  clear_error(c);
  if (len > 0xFFFFFFFFull - 16 || nframes >= (1u << 30)) return lib_error(c, ETLG_InvalidArgument, "batch too large (max 4 GiB, 2^30 frames)");
  HIPCHK(c, hipSetDevice(c->device));
  { const int32_t rc_ = flush_deferred(c); (void)rc_; }
  bool in_dev = flags & ETLG_F_INPUT_ON_DEVICE;
  const bool out_dev = flags & ETLG_F_OUTPUT_ON_DEVICE;
  const bool no_ctrl = flags & ETLG_F_NO_CONTROL;
  const bool scan = frame_offsets == nullptr;
  // ---- ASYNC with HOST input (the staging batcher of a Rust host, crates/etl-gfx950/src/batcher.rs: pinned buffers from
  //      etlg_host_alloc): the bytes and the sidecar are copied into a device block the batch owns, on a copy stream of their own,
  //      and from here on the batch IS a device-input batch — it joins the chain, and its upload runs beside the decode of the batch
  //      before it (double buffering: the caller fills its next buffer meanwhile). The caller keeps the host buffers untouched until
  //      the batch is synced, as for every ASYNC batch (include/etlg.h).
  void* stage_blk = nullptr; size_t stage_cap = 0; hipEvent_t h2d_done = nullptr;
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
    for (hipStream_t w : {c->stream, c->stream2, c->ctl_stream, c->scan_stream}) if (w) HIPCHK(c, hipStreamWaitEvent(w, h2d_done, 0));
    buf = (const uint8_t*)stage_blk; frame_offsets = (const uint32_t*)((const uint8_t*)stage_blk + o_offs);
    in_dev = true;
    c->staged_async++;
  }
  struct StageGuard { etlg_ctx* c; void*& p; size_t& cap; hipEvent_t& ev; ~StageGuard() { if (p) blk_give(c, c->gen, p, cap, false); if (ev) c->ev_pool.push_back(ev); } } stage_guard{c, stage_blk, stage_cap, h2d_done};
  // ASYNC batches are chained on the device (DecParams.carry) and may be decoded again when they are synced, so everything
  // they read must still be there then: device-resident input AND sidecar (the context's staging and scan buffers are shared
  // by all batches). Anything else is decoded synchronously; etlg_batch_sync on such a batch returns its stored result.
  // Without the caller's no-control assertion the first attempt is optimistic as long as the stream has not been carrying Relation /
  // DDL frames (a control frame then fails the batch with a hint and finish_batch takes the control path); on a stream that does
  // carry them (last_had_ctrl) the control pre-pass runs ahead on its own stream (ctl_begin) — that needs the sidecar.
  const bool async_ok = (flags & ETLG_F_ASYNC) && out_dev && in_dev && !c->copy.active && !c->force_multipass && len < (1ull << 31);
  const bool ctl_ahead = async_ok && !no_ctrl && c->last_had_ctrl && !scan && nframes && c->ctl_async_mode;
  const bool async = async_ok && (no_ctrl || !c->last_had_ctrl || ctl_ahead);
  hipStream_t s = c->stream;
  if (!async) { const int32_t rc = drain_pending(c); if (rc != ETLG_OK) return rc; clear_error(c); }

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
  b->stage_blk = stage_blk; b->stage_cap = stage_cap; b->h2d_done = h2d_done;   // the batch owns them from here (etlg_batch_free)
  stage_blk = nullptr; h2d_done = nullptr;
  b->host_in = in_dev ? nullptr : buf; b->host_offs = (in_dev || scan) ? nullptr : h_offs; b->dev_in = in_dev ? buf : nullptr;
  b->d_in_ptr = d_in_ptr; b->user_offs = frame_offsets;
  if (scan) {
    if (async) {
      // The scan of THIS batch runs on its own stream while the previous batch is still being decoded on the context's, and
      // nobody waits for it here: the call returns with the scan in flight and the NEXT call (or the batch's sync) collects
      // the frame count and enqueues the decode. The input must be complete when the call is made (include/etlg.h).
      if (!c->scan_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->scan_stream, hipStreamNonBlocking));
      if (c->offs_pool.empty()) c->offs_pool.push_back(new DevBuf());
      b->scan_offs = c->offs_pool.back(); c->offs_pool.pop_back();
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
    cp.res = (DevResult*)c->d_ctl_res.p + (c->ctl_seq++ % etlg_ctx::kCtlRing);
    { const int32_t rc = ctl_begin(c, b, cp, c->ctl_stream, true); if (rc != ETLG_OK) return rc; }
    b->deferred = true; b->defer_ctl = true; b->pending = true; b->v.on_device = 1;
    c->deferred = b; c->pending.push_back(b);
    c->ctl_ahead_n++;
    guard.b = nullptr;
    *out = b;
    return ETLG_OK;
  }
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
  if (b->defer_ctl) { b->defer_ctl = false; nframes = b->nframes_in; }   // its control pre-pass ran ahead: decode_tail collects it (standard_path -> ctl_finish)
  else {
    const hipError_t e = scan_collect(c, c->scan_job, &nframes);
    if (e != hipSuccess) rc = lib_error(c, ETLG_DeviceError, hipGetErrorString(e));
    else if (nframes >= (1u << 30)) rc = lib_error(c, ETLG_InvalidArgument, "batch too large (max 4 GiB, 2^30 frames)");
  }
  if (rc == ETLG_OK) {
    etlg_batch* prev = nullptr;   // the batch issued just before this one, if it is still in flight
    for (size_t i = 0; i < c->pending.size(); i++) if (c->pending[i] == b && i > 0) prev = c->pending[i - 1];
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
  p.carry = (async && prev) ? prev->d_res_blk : nullptr;

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
  }
  p.res = b->d_res_blk;
  { SlowScope slow_scope_pool(c, "decode_tail: pinned result block");
  if (c->res_pool.empty()) { DevResult* r = nullptr; HIPCHK(c, hipHostMalloc((void**)&r, sizeof(DevResult), hipHostMallocDefault)); c->res_pool.push_back(r); }
  b->h_res = c->res_pool.back(); c->res_pool.pop_back(); }
  if (b->copy.active) launch_copy(c, b->copy, p);  // rows -> Insert frames (writes p.in / p.offs), row-level errors

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

extern "C++" {
// ---- columnar hand-off (columns.hip)
namespace {
struct ColPlan { uint32_t kind, vbytes; bool var; uint32_t child = 0, child_bytes = 0, elem = 0, fmt = 0; };   // fmt: the kernels' internal kind of a formatted string column (columns.hip AK_*_STR)
ColPlan list_plan(uint32_t elem) {  // array literals the device parses: element classes with a fixed-width value
  switch (elem) {
    case ETLG_TC_BOOL: return {ETLG_AK_LIST, 0, true, ETLG_AK_BOOLEAN, 0, elem};
    case ETLG_TC_I16: case ETLG_TC_I32: return {ETLG_AK_LIST, 0, true, ETLG_AK_INT32, 4, elem};
    case ETLG_TC_I64: case ETLG_TC_U32: return {ETLG_AK_LIST, 0, true, ETLG_AK_INT64, 8, elem};
    case ETLG_TC_F32: return {ETLG_AK_LIST, 0, true, ETLG_AK_FLOAT32, 4, elem};
    case ETLG_TC_F64: return {ETLG_AK_LIST, 0, true, ETLG_AK_FLOAT64, 8, elem};
    case ETLG_TC_DATE: return {ETLG_AK_LIST, 0, true, ETLG_AK_DATE32, 4, elem};
    case ETLG_TC_TIME: return {ETLG_AK_LIST, 0, true, ETLG_AK_TIME64_US, 8, elem};
    case ETLG_TC_TIMESTAMP: return {ETLG_AK_LIST, 0, true, ETLG_AK_TIMESTAMP_US, 8, elem};
    case ETLG_TC_TIMESTAMPTZ: return {ETLG_AK_LIST, 0, true, ETLG_AK_TIMESTAMP_US_UTC, 8, elem};
    case ETLG_TC_UUID: return {ETLG_AK_LIST, 0, true, ETLG_AK_FIXED16, 16, elem};
    case ETLG_TC_STRING: return {ETLG_AK_LIST, 0, true, ETLG_AK_LARGE_UTF8, 0, elem};   // text[], varchar[], and every array type without a dedicated arm
    // ArrayCell::Numeric / TimeTz: lists of Display strings (iceberg/encoding.rs:902-945); ArrayCell::Bytes: lists of the decoded bytes
    case ETLG_TC_NUMERIC: case ETLG_TC_TIMETZ: return {ETLG_AK_LIST, 0, true, ETLG_AK_LARGE_UTF8, 0, elem};
    case ETLG_TC_BYTEA: return {ETLG_AK_LIST, 0, true, ETLG_AK_LARGE_BINARY, 0, elem};
    default: return {ETLG_AK_TEXT_FORM, 0, true};   // json elements: the host's
  }
}
bool var_child(const ColPlan& p) { return p.child == ETLG_AK_LARGE_UTF8 || p.child == ETLG_AK_LARGE_BINARY; }   // list children with offsets of their own
ColPlan col_plan(uint32_t cls) {
  switch (cls) {
    case ETLG_TC_BOOL: return {ETLG_AK_BOOLEAN, 0, false};
    case ETLG_TC_I16: case ETLG_TC_I32: return {ETLG_AK_INT32, 4, false};
    case ETLG_TC_I64: case ETLG_TC_U32: return {ETLG_AK_INT64, 8, false};
    case ETLG_TC_F32: return {ETLG_AK_FLOAT32, 4, false};
    case ETLG_TC_F64: return {ETLG_AK_FLOAT64, 8, false};
    case ETLG_TC_DATE: return {ETLG_AK_DATE32, 4, false};
    case ETLG_TC_TIME: return {ETLG_AK_TIME64_US, 8, false};
    case ETLG_TC_TIMESTAMP: return {ETLG_AK_TIMESTAMP_US, 8, false};
    case ETLG_TC_TIMESTAMPTZ: return {ETLG_AK_TIMESTAMP_US_UTC, 8, false};
    case ETLG_TC_UUID: return {ETLG_AK_FIXED16, 16, false};
    case ETLG_TC_STRING: return {ETLG_AK_LARGE_UTF8, 0, true};
    case ETLG_TC_BYTEA: return {ETLG_AK_LARGE_BINARY, 0, true};
    // Display strings in every sink (cell_to_string, iceberg/encoding.rs:349-352; n.to_string() / t.to_string()): formatted on the device
    case ETLG_TC_NUMERIC: return {ETLG_AK_LARGE_UTF8, 0, true, 0, 0, 0, 14u};
    case ETLG_TC_TIMETZ: return {ETLG_AK_LARGE_UTF8, 0, true, 0, 0, 0, 15u};
    default: return {ETLG_AK_TEXT_FORM, 0, true};
  }
}
}  // namespace
}  // extern "C++"

int32_t etlg_batch_columns(etlg_ctx* c, etlg_batch* b, int32_t slot, uint32_t row_kinds, uint32_t flags, etlg_columns** out) {
  if (!c || !b || !out || b->ctx != c) return ETLG_InvalidArgument;
  *out = nullptr;
  if (b->pending) { const int32_t rc = etlg_batch_sync(c, b); if (rc != ETLG_OK) return rc; }   // an ASYNC batch that ended in a decode error: the caller gets that error (fail-fast, as the reference), not a hand-off of the prefix
  if (!b->v.on_device || !b->dev) return lib_error(c, ETLG_InvalidState, "etlg_batch_columns needs a device-resident batch (ETLG_F_OUTPUT_ON_DEVICE, not downloaded)");
  if (slot < 0 || (size_t)slot >= c->slots.size() || !(row_kinds & 3u)) return ETLG_InvalidArgument;
  const bool parse_arrays = (row_kinds & ETLG_ROWS_PARSE_ARRAYS) != 0;
  const SlotHost& sh = *c->slots[(size_t)slot];
  hipStream_t s = c->stream;
  const etlg_batch_view& bv = b->v;
  auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
  std::unique_ptr<etlg_columns, void (*)(etlg_columns*)> cs(new etlg_columns, etlg_columns_free);
  const uint64_t ne = bv.n_events;
  const uint32_t nblk = (uint32_t)((ne + 255) / 256);
  // ---- 1. which events are rows (count -> scan -> scatter); row_event / row_base are sized for every event
  HIPCHK(c, c->d_colsel.ensure(al((size_t)(nblk + 1) * 4) + 64));
  uint32_t* d_blk = (uint32_t*)c->d_colsel.p;
  cs->m.ctx = c; cs->m.ctx_gen = c->gen;
  ScratchBlk rows_blk{c};
  if (ne) HIPCHK(c, blk_take(c, al(ne * 8) * 2, false, &rows_blk.p, &rows_blk.cap));
  void* d_rows = rows_blk.p;
  uint64_t* d_row_event = (uint64_t*)d_rows;
  uint64_t* d_row_base = (uint64_t*)((uint8_t*)d_rows + al(ne * 8));
  uint32_t n_rows32 = 0;
  if (ne) {
    ColSel q{};
    q.ev_kind = bv.ev_kind; q.ev_flags = bv.ev_flags; q.ev_slot = bv.ev_schema_slot; q.ev_body = bv.ev_body_off;
    q.n_events = ne; q.slot = (uint32_t)slot; q.kinds = row_kinds & 3u;
    q.row_full = sh.desc.row_bytes_full; q.row_key = sh.desc.row_bytes_key;
    q.blk = d_blk; q.nblocks = nblk; q.row_event = d_row_event; q.row_base = d_row_base;
    etlg_k_col_select(&q, s);
    HIPCHK(c, hipMemcpyAsync(&n_rows32, d_blk + nblk, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  const uint64_t n = n_rows32;
  // ---- 2. block A: row_event | per column {validity, deferred, values or (lens, offsets)} | counters
  const uint32_t nc = sh.desc.n_cols;
  const size_t bm = al(((size_t)n + 63) / 64 * 8);
  struct Lay { ColPlan pl; size_t validity, deferred, values, lens, offsets; };
  std::vector<Lay> lay(nc);
  size_t off = al(n * 8);
  for (uint32_t i = 0; i < nc; i++) {
    Lay& l = lay[i];
    l.pl = col_plan(sh.cols[i].type_class);
    if (parse_arrays && sh.cols[i].type_class == ETLG_TC_ARRAY) l.pl = list_plan((uint32_t)etlg_array_elem_class(sh.cols[i].type_oid));
    l.validity = l.deferred = l.values = l.lens = l.offsets = 0;
    if (l.pl.kind == ETLG_AK_NONE) continue;
    l.validity = off; off += bm; l.deferred = off; off += bm;
    if (l.pl.var) { l.offsets = off; off += al((n + 1) * 8); l.lens = off; off += al(n * 4); }
    else { l.values = off; off += l.pl.kind == ETLG_AK_BOOLEAN ? bm : al(n * l.pl.vbytes); }
  }
  const size_t o_cnt = off; off += al((size_t)nc * 32);   // per column: nulls, deferred, child nulls, first list error
  const uint32_t nrb = (uint32_t)((n + 255) / 256);
  const size_t o_scan = off; off += al((size_t)(nrb + 1) * 8);   // one scan scratch: var-len columns run one after another on the stream
  const size_t a_bytes = off + 64;
  HIPCHK(c, blk_take(c, a_bytes, false, &cs->m.d_a, &cs->m.cap_a));
  uint8_t* A = (uint8_t*)cs->m.d_a;
  {
    std::vector<unsigned long long> init((size_t)nc * 4, 0ull);
    for (uint32_t i = 0; i < nc; i++) init[(size_t)i * 4 + 3] = ~0ull;
    if (nc) HIPCHK(c, hipMemcpy(A + o_cnt, init.data(), (size_t)nc * 32, hipMemcpyHostToDevice));
  }
  if (n) HIPCHK(c, hipMemcpyAsync(A, d_row_event, n * 8, hipMemcpyDeviceToDevice, s));
  std::vector<ColJob> jobs(nc);
  std::vector<int64_t> var_total(nc, 0);
  for (uint32_t i = 0; i < nc; i++) {
    const Lay& l = lay[i];
    if (l.pl.kind == ETLG_AK_NONE) continue;
    ColJob& j = jobs[i];
    j = ColJob{};
    j.fixed = bv.fixed; j.heap = bv.heap; j.row_base = d_row_base; j.n_rows = n;
    j.col_index = i; j.off_full = sh.cols[i].off_full; j.cls = sh.cols[i].type_class; j.kind = l.pl.fmt ? l.pl.fmt : l.pl.kind;
    j.validity = (unsigned long long*)(A + l.validity); j.deferred = (unsigned long long*)(A + l.deferred);
    j.null_count = (unsigned long long*)(A + o_cnt + (size_t)i * 32); j.deferred_count = j.null_count + 1;
    j.child_nulls = j.null_count + 2; j.err = j.null_count + 3; j.elem_cls = l.pl.elem;
    if (l.pl.var) {
      j.lens = (uint32_t*)(A + l.lens); j.offsets = (const int64_t*)(A + l.offsets);
      if (n) {
        if (l.pl.kind == ETLG_AK_LIST) { j.kind = l.pl.child; etlg_k_col_list(&j, (unsigned long long*)(A + o_scan), (int64_t*)(A + l.offsets), 0, s); }
        else etlg_k_col_var(&j, (unsigned long long*)(A + o_scan), (int64_t*)(A + l.offsets), 0, s);
        HIPCHK(c, hipMemcpyAsync(&var_total[i], A + l.offsets + n * 8, 8, hipMemcpyDeviceToHost, s));
      } else {
        HIPCHK(c, hipMemsetAsync(A + l.offsets, 0, 8, s));
      }
    } else {
      j.values = A + l.values;
      etlg_k_col_fixed(&j, s);
    }
  }
  std::vector<uint64_t> cnt((size_t)nc * 4, 0);
  if (nc) HIPCHK(c, hipMemcpyAsync(cnt.data(), A + o_cnt, (size_t)nc * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  {  // a malformed array literal: the reference's error, for the first such row in event order
    uint64_t first = ~0ull;
    for (uint32_t i = 0; i < nc; i++) if (lay[i].pl.kind == ETLG_AK_LIST || sh.cols[i].type_class == ETLG_TC_JSON) first = std::min(first, cnt[(size_t)i * 4 + 3]);   // (a json cell that is not one JSON value: the same report)
    if (first != ~0ull) {
      uint64_t ev = 0;
      HIPCHK(c, hipMemcpy(&ev, d_row_event + (first >> 8), 8, hipMemcpyDeviceToHost));
      return set_error(c, (int32_t)(first & 0xFF), (int64_t)ev);
    }
  }
  // ---- 3. block B: the bytes of the var-len columns; list columns: child values (or, for lists of strings, child offsets +
  //      lengths) and child validity
  std::vector<size_t> vb(nc, 0);
  size_t b_bytes = 0;
  std::vector<size_t> cvb(nc, 0), vbytes(nc, 0), clen(nc, 0), cscan(nc, 0);   // list columns: child validity offset; bytes behind `values`; child lens; scan scratch
  bool any_text_list = false;
  for (uint32_t i = 0; i < nc; i++) {
    if (!lay[i].pl.var) continue;
    const size_t tot = (size_t)var_total[i];
    vb[i] = b_bytes;
    if (lay[i].pl.kind == ETLG_AK_LIST) {
      const size_t bits = (tot + 63) / 64 * 8;
      if (var_child(lay[i].pl)) {   // child offsets first (i64), then lengths, scan scratch, validity
        any_text_list = true;
        b_bytes += al((tot + 1) * 8); clen[i] = b_bytes; b_bytes += al(tot * 4); cscan[i] = b_bytes; b_bytes += al((tot / 256 + 2) * 8);
        cvb[i] = b_bytes; b_bytes += al(bits);
      } else {
        vbytes[i] = lay[i].pl.child == ETLG_AK_BOOLEAN ? bits : tot * lay[i].pl.child_bytes;
        b_bytes += al(vbytes[i]); cvb[i] = b_bytes; b_bytes += al(bits);
      }
    } else { vbytes[i] = tot; b_bytes += al(tot); }
  }
  if (b_bytes) HIPCHK(c, blk_take(c, b_bytes + 64, false, &cs->m.d_b, &cs->m.cap_b));
  uint8_t* B = (uint8_t*)cs->m.d_b;
  std::vector<int64_t> text_total(nc, 0);
  for (uint32_t i = 0; i < nc; i++) {
    if (!lay[i].pl.var || var_total[i] <= 0) continue;
    if (lay[i].pl.kind == ETLG_AK_LIST) {
      jobs[i].child_validity = (uint32_t*)(B + cvb[i]);
      if (var_child(lay[i].pl)) {   // pass A: byte length + validity of every element, then their offsets
        jobs[i].values = nullptr; jobs[i].child_lens = (uint32_t*)(B + clen[i]); jobs[i].child_offsets = (const int64_t*)(B + vb[i]);
        HIPCHK(c, hipMemsetAsync(B + cvb[i], 0, al(((size_t)var_total[i] + 63) / 64 * 8), s));
        etlg_k_col_list(&jobs[i], nullptr, nullptr, 1, s);
        etlg_k_scan_lens(jobs[i].child_lens, (uint64_t)var_total[i], (unsigned long long*)(B + cscan[i]), (int64_t*)(B + vb[i]), s);
        HIPCHK(c, hipMemcpyAsync(&text_total[i], B + vb[i] + (size_t)var_total[i] * 8, 8, hipMemcpyDeviceToHost, s));
      } else {
        jobs[i].values = B + vb[i];
        HIPCHK(c, hipMemsetAsync(B + vb[i], 0, cvb[i] - vb[i] + al(((size_t)var_total[i] + 63) / 64 * 8), s));   // bitmaps are OR-ed into
        etlg_k_col_list(&jobs[i], nullptr, nullptr, 1, s);
      }
    } else { jobs[i].values = B + vb[i]; etlg_k_col_var(&jobs[i], nullptr, nullptr, 1, s); }
  }
  // ---- 3b. block C: the element bytes of lists of strings
  std::vector<size_t> vc(nc, 0);
  size_t c_bytes = 0;
  if (any_text_list) {
    HIPCHK(c, hipStreamSynchronize(s));
    for (uint32_t i = 0; i < nc; i++) if (lay[i].pl.kind == ETLG_AK_LIST && var_child(lay[i].pl)) { vc[i] = c_bytes; vbytes[i] = (size_t)text_total[i]; c_bytes += al((size_t)text_total[i]) + 64; }
    if (c_bytes) HIPCHK(c, blk_take(c, c_bytes + 64, false, &cs->m.d_c, &cs->m.cap_c));
    for (uint32_t i = 0; i < nc; i++)
      if (lay[i].pl.kind == ETLG_AK_LIST && var_child(lay[i].pl) && var_total[i] > 0) { jobs[i].values = (uint8_t*)cs->m.d_c + vc[i]; etlg_k_col_list(&jobs[i], nullptr, nullptr, 1, s); }
  }
  uint8_t* Cb = (uint8_t*)cs->m.d_c;
  if (nc) HIPCHK(c, hipMemcpyAsync(cnt.data(), A + o_cnt, (size_t)nc * 32, hipMemcpyDeviceToHost, s));   // again: the child null counts
  // ---- 4. the view (device pointers, or a host copy of the blocks)
  const bool on_dev = (flags & ETLG_F_OUTPUT_ON_DEVICE) != 0;
  const uint8_t* base_a = A; const uint8_t* base_b = B; const uint8_t* base_c = Cb;
  if (!on_dev) {
    HIPCHK(c, blk_take(c, al(o_cnt) + al(b_bytes) + c_bytes + 64, true, (void**)&cs->m.h, &cs->m.cap_h));
    if (o_cnt) HIPCHK(c, hipMemcpyAsync(cs->m.h, A, o_cnt, hipMemcpyDeviceToHost, s));
    if (b_bytes) HIPCHK(c, hipMemcpyAsync(cs->m.h + al(o_cnt), B, b_bytes, hipMemcpyDeviceToHost, s));
    if (c_bytes) HIPCHK(c, hipMemcpyAsync(cs->m.h + al(o_cnt) + al(b_bytes), Cb, c_bytes, hipMemcpyDeviceToHost, s));
    base_a = cs->m.h; base_b = cs->m.h + al(o_cnt); base_c = cs->m.h + al(o_cnt) + al(b_bytes);
  }
  HIPCHK(c, hipStreamSynchronize(s));   // row_base (freed on return) is read by the kernels above
  if (!on_dev) {
    blk_give(c, c->gen, cs->m.d_a, cs->m.cap_a, false); blk_give(c, c->gen, cs->m.d_b, cs->m.cap_b, false); blk_give(c, c->gen, cs->m.d_c, cs->m.cap_c, false);
    cs->m.d_a = cs->m.d_b = cs->m.d_c = nullptr;
  }
  cs->cols.resize(nc);
  for (uint32_t i = 0; i < nc; i++) {
    etlg_column& k = cs->cols[i];
    const Lay& l = lay[i];
    k = etlg_column{};
    k.type_class = sh.cols[i].type_class; k.arrow_kind = l.pl.kind; k.value_bytes = l.pl.vbytes; k.nullable = sh.cols[i].nullable;
    if (l.pl.kind == ETLG_AK_NONE) continue;
    k.null_count = cnt[(size_t)i * 4]; k.deferred_count = cnt[(size_t)i * 4 + 1];
    k.validity = base_a + l.validity; k.deferred = base_a + l.deferred;
    if (l.pl.var) { k.offsets = (const int64_t*)(base_a + l.offsets); k.values = base_b ? base_b + vb[i] : nullptr; k.values_bytes = (uint64_t)vbytes[i]; }
    if (l.pl.kind == ETLG_AK_LIST) {
      k.child_kind = l.pl.child; k.child_count = (uint64_t)var_total[i]; k.child_null_count = cnt[(size_t)i * 4 + 2];
      k.child_validity = base_b ? base_b + cvb[i] : nullptr;
      if (var_child(l.pl)) {   // element texts / bytes: offsets in block B, bytes in block C
        k.child_offsets = base_b ? (const int64_t*)(base_b + vb[i]) : nullptr;
        k.values = base_c ? base_c + vc[i] : nullptr;
      }
    }
    if (!l.pl.var) { k.values = base_a + l.values; k.values_bytes = l.pl.kind == ETLG_AK_BOOLEAN ? ((n + 63) / 64) * 8 : n * l.pl.vbytes; }
  }
  cs->v.n_rows = n; cs->v.n_cols = nc; cs->v.on_device = on_dev ? 1u : 0u; cs->v.cols = cs->cols.data();
  cs->v.row_event = (const uint64_t*)base_a;
  *out = cs.release();
  return ETLG_OK;
}

int32_t etlg_batch_size_hints(etlg_ctx* c, etlg_batch* b, const etlg_size_model* m, uint32_t flags, uint64_t* out) {
  if (!c || !b || !m || b->ctx != c) return ETLG_InvalidArgument;
  if (b->pending) { const int32_t rc = etlg_batch_sync(c, b); if (rc != ETLG_OK) return rc; }   // an ASYNC batch that ended in a decode error: the caller gets that error (fail-fast, as the reference), not a hand-off of the prefix
  if (!b->v.on_device || !b->dev) return lib_error(c, ETLG_InvalidState, "etlg_batch_size_hints needs a device-resident batch (ETLG_F_OUTPUT_ON_DEVICE, not downloaded)");
  const etlg_batch_view& bv = b->v;
  const uint64_t ne = bv.n_events;
  if (!ne) return ETLG_OK;
  if (!out) return ETLG_InvalidArgument;
  hipStream_t s = c->stream;
  std::vector<uint32_t> tab;
  const uint32_t ns = (uint32_t)c->slots.size();
  tab.resize((size_t)ns * 5);
  uint32_t ncols = 0;
  for (uint32_t i = 0; i < ns; i++) {
    const SlotHost& sh = *c->slots[i];
    uint32_t* t = &tab[(size_t)i * 5];
    t[0] = sh.desc.n_cols; t[1] = sh.desc.n_ident; t[2] = sh.desc.row_bytes_full; t[3] = sh.desc.row_bytes_key; t[4] = ncols;
    ncols += sh.desc.n_cols;
  }
  const size_t o_cols = tab.size();
  tab.resize(o_cols + (size_t)ncols * 2);
  for (uint32_t i = 0, k = 0; i < ns; i++)
    for (const etlg_slot_col& col : c->slots[i]->cols) { tab[o_cols + 2 * k] = col.type_class | (col.identity ? 1u << 8 : 0u) | ((uint32_t)col.off_full << 16); tab[o_cols + 2 * k + 1] = col.off_key; k++; }
  const bool on_dev = (flags & ETLG_F_OUTPUT_ON_DEVICE) != 0;
  const size_t tab_bytes = (tab.size() * 4 + 63) & ~(size_t)63;
  ScratchBlk dblk{c};
  HIPCHK(c, blk_take(c, tab_bytes + (on_dev ? 0 : ne * 8) + 64, false, &dblk.p, &dblk.cap));
  void* d = dblk.p;
  HIPCHK(c, hipMemcpyAsync(d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
  HintJob j{};
  j.ev_kind = bv.ev_kind; j.ev_flags = bv.ev_flags; j.ev_table = bv.ev_table_id; j.ev_slot = bv.ev_schema_slot; j.ev_body = bv.ev_body_off;
  j.fixed = bv.fixed; j.heap = bv.heap; j.n_events = ne;
  j.slots = (const uint32_t*)d; j.cols = (const uint32_t*)d + o_cols; j.n_slots = ns;
  j.m_begin = m->begin_event; j.m_commit = m->commit_event; j.m_insert = m->insert_event; j.m_update = m->update_event; j.m_delete = m->delete_event;
  j.m_truncate = m->truncate_event; j.m_relation = m->relation_event; j.m_rts = m->replicated_table_schema; j.m_row = m->table_row; j.m_cell = m->cell;
  j.out = on_dev ? (unsigned long long*)out : (unsigned long long*)((uint8_t*)d + tab_bytes);
  etlg_k_size_hints(&j, s);
  if (!on_dev) HIPCHK(c, hipMemcpyAsync(out, j.out, ne * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));   // the tables are freed on return
  return ETLG_OK;
}

int32_t etlg_columns_view_get(const etlg_columns* cs, etlg_columns_view* out) {
  if (!cs || !out) return ETLG_InvalidArgument;
  *out = cs->v;
  return ETLG_OK;
}

void etlg_columns_free(etlg_columns* cs) {
  if (!cs) return;
  handoff_release(cs->m);
  delete cs;
}

static int32_t handoff_rows(etlg_ctx* c, etlg_batch* b, int32_t slot, const uint8_t* nullable_flags, uint32_t n_flags, int32_t engine,
                            uint32_t flags, uint32_t format, etlg_rowbinary** out);

int32_t etlg_batch_rowbinary(etlg_ctx* c, etlg_batch* b, int32_t slot, const uint8_t* nullable_flags, uint32_t n_flags, int32_t engine,
                             uint32_t flags, etlg_rowbinary** out) {
  if (!c || !b || !out || b->ctx != c || !nullable_flags) return ETLG_InvalidArgument;
  return handoff_rows(c, b, slot, nullable_flags, n_flags, engine, flags, 0u, out);
}

int32_t etlg_batch_protobuf(etlg_ctx* c, etlg_batch* b, int32_t slot, uint32_t flags, etlg_rowbinary** out) {
  if (!c || !b || !out || b->ctx != c) return ETLG_InvalidArgument;
  return handoff_rows(c, b, slot, nullptr, 0u, ETLG_CH_MERGE_TREE, flags, 1u, out);
}

// format 0: ClickHouse RowBinary (Insert / Update / Delete rows + the engine's CDC columns); 1: BigQuery protobuf (Insert rows)
static int32_t handoff_rows(etlg_ctx* c, etlg_batch* b, int32_t slot, const uint8_t* nullable_flags, uint32_t n_flags, int32_t engine,
                            uint32_t flags, uint32_t format, etlg_rowbinary** out) {
  *out = nullptr;
  if (b->pending) { const int32_t rc = etlg_batch_sync(c, b); if (rc != ETLG_OK) return rc; }   // an ASYNC batch that ended in a decode error: the caller gets that error (fail-fast, as the reference), not a hand-off of the prefix
  if (!b->v.on_device || !b->dev) return lib_error(c, ETLG_InvalidState, "etlg_batch_rowbinary needs a device-resident batch (ETLG_F_OUTPUT_ON_DEVICE, not downloaded)");
  if (slot < 0 || (size_t)slot >= c->slots.size() || (engine != ETLG_CH_MERGE_TREE && engine != ETLG_CH_REPLACING_MERGE_TREE)) return ETLG_InvalidArgument;
  const SlotHost& sh = *c->slots[(size_t)slot];
  const uint32_t nc = sh.desc.n_cols;
  if (format == 0 && n_flags != nc + 2) return lib_error(c, ETLG_ConversionError, "ClickHouse RowBinary row width mismatch");
  std::unique_ptr<etlg_rowbinary, void (*)(etlg_rowbinary*)> rb(new etlg_rowbinary, etlg_rowbinary_free);
  const bool on_dev = (flags & ETLG_F_OUTPUT_ON_DEVICE) != 0;
  rb->v.on_device = on_dev ? 1u : 0u; rb->v.host_event = ~0ull;
  std::vector<uint32_t> cols(nc);
  for (uint32_t i = 0; i < nc; i++) {
    const uint32_t cls = sh.cols[i].type_class;
    uint32_t elem = 0;
    bool host_class = col_plan(cls).kind == ETLG_AK_TEXT_FORM;   // json (serde_json's normalised Display): the host writes it. numeric / timetz Display strings are formatted on the device
    if (format == 1 && cls == ETLG_TC_ARRAY) host_class = true;   // packed / repeated array fields + NULL-element validation: the host's
    else if (cls == ETLG_TC_ARRAY) {  // arrays of fixed-width elements are encoded on the device (Array(Nullable(T)))
      elem = (uint32_t)etlg_array_elem_class(sh.cols[i].type_oid);
      const ColPlan lp = list_plan(elem);
      host_class = lp.kind != ETLG_AK_LIST || var_child(lp);
    }
    if (host_class) {
      rb->v.status = ETLG_RB_NEEDS_HOST; rb->v.host_column = i;
      *out = rb.release();
      return ETLG_OK;
    }
    cols[i] = cls | ((nullable_flags && nullable_flags[i]) ? 1u << 8 : 0u) | (elem << 9) | ((uint32_t)sh.cols[i].off_full << 16);
  }
  hipStream_t s = c->stream;
  const etlg_batch_view& bv = b->v;
  auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
  const uint64_t ne = bv.n_events;
  const uint32_t nblk = (uint32_t)((ne + 255) / 256);
  // block S (freed on return): block counts | host-row counter | error word | column words | row_base
  const size_t o_cnt = al((size_t)(nblk + 1) * 4), o_cols = o_cnt + 64, o_base = o_cols + al((size_t)nc * 4 + 4), s_bytes = o_base + al(ne * 8) + 64;
  rb->m.ctx = c; rb->m.ctx_gen = c->gen;
  ScratchBlk sblk{c};
  HIPCHK(c, blk_take(c, s_bytes, false, &sblk.p, &sblk.cap));
  void* d_s = sblk.p;
  uint8_t* S = (uint8_t*)d_s;
  const unsigned long long init[2] = {0ull, ~0ull};
  HIPCHK(c, hipMemcpyAsync(S + o_cnt, init, 16, hipMemcpyHostToDevice, s));
  if (nc) HIPCHK(c, hipMemcpyAsync(S + o_cols, cols.data(), (size_t)nc * 4, hipMemcpyHostToDevice, s));
  // block A: row_event | row_offsets | lens | scan scratch (sized for every event being a row)
  const size_t o_off = al(ne * 8), o_len = o_off + al((ne + 1) * 8), o_scan = o_len + al(ne * 4), a_bytes = o_scan + al((size_t)(nblk + 1) * 8) + 64;
  HIPCHK(c, blk_take(c, a_bytes, false, &rb->m.d_a, &rb->m.cap_a));
  uint8_t* A = (uint8_t*)rb->m.d_a;
  uint32_t n32 = 0;
  unsigned long long cnt[2] = {0, ~0ull};
  if (ne) {
    ColSel q{};
    q.ev_kind = bv.ev_kind; q.ev_flags = bv.ev_flags; q.ev_slot = bv.ev_schema_slot; q.ev_body = bv.ev_body_off;
    // ReplacingMergeTree keys its dedup on the source primary key: the reference refuses Update events of a table whose replica
    // identity is neither PrimaryKey nor Full (clickhouse_update_row -> ensure_clickhouse_key_identity_is_primary_key,
    // clickhouse/core.rs:1359-1427). Such Updates are not encoded here: they are left to the host (n_host_rows), which raises
    // the reference's SourceReplicaIdentityError when it meets the first of them.
    const bool upd_ok = format != 0 || engine != ETLG_CH_REPLACING_MERGE_TREE || sh.identity_type == 1 || sh.identity_type == 2;
    q.n_events = ne; q.slot = (uint32_t)slot; q.kinds = format ? 1u : (upd_ok ? 7u : 5u); q.host_rows = (unsigned long long*)(S + o_cnt);
    q.row_full = sh.desc.row_bytes_full; q.row_key = sh.desc.row_bytes_key;
    q.blk = (uint32_t*)S; q.nblocks = nblk; q.row_event = (uint64_t*)A; q.row_base = (uint64_t*)(S + o_base);
    etlg_k_col_select(&q, s);
    HIPCHK(c, hipMemcpyAsync(&n32, S + (size_t)nblk * 4, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  const uint64_t n = n32;
  RbJob j{};
  j.fixed = bv.fixed; j.heap = bv.heap; j.row_event = (const uint64_t*)A; j.row_base = (const uint64_t*)(S + o_base);
  j.ev_kind = bv.ev_kind; j.ev_commit = bv.ev_commit_lsn; j.ev_ord = bv.ev_tx_ordinal;
  j.n_rows = n; j.n_cols = nc; j.engine = (uint32_t)engine;
  j.cdc_nullable = nullable_flags ? (nullable_flags[nc] ? 1u : 0u) | (nullable_flags[nc + 1] ? 2u : 0u) : 0u;
  j.format = format;
  j.cols = (const uint32_t*)(S + o_cols); j.lens = (uint32_t*)(A + o_len); j.offsets = (const int64_t*)(A + o_off);
  j.err = (unsigned long long*)(S + o_cnt) + 1;
  int64_t total = 0;
  if (n) {
    etlg_k_rowbinary(&j, (unsigned long long*)(A + o_scan), (int64_t*)(A + o_off), 0, s);
    HIPCHK(c, hipMemcpyAsync(&total, A + o_off + n * 8, 8, hipMemcpyDeviceToHost, s));
  } else {
    HIPCHK(c, hipMemsetAsync(A + o_off, 0, 8, s));
  }
  HIPCHK(c, hipMemcpyAsync(cnt, S + o_cnt, 16, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  rb->v.n_host_rows = cnt[0];
  if (cnt[1] != ~0ull) {  // the first row (event order) with a cell that has no encoding
    const uint32_t code = (uint32_t)(cnt[1] & 0xFF), col = (uint32_t)((cnt[1] >> 8) & 0xFFFF);
    uint64_t ev = 0;
    HIPCHK(c, hipMemcpy(&ev, A + ((cnt[1] & ~(1ull << 62)) >> 24) * 8, 8, hipMemcpyDeviceToHost));   // (bit 62: not a date-range error, k_rb_lens)
    if (code == 3) {
      rb->v.status = ETLG_RB_NEEDS_HOST; rb->v.host_event = ev; rb->v.host_column = col;
      blk_give(c, c->gen, rb->m.d_a, rb->m.cap_a, false); rb->m.d_a = nullptr;
      *out = rb.release();
      return ETLG_OK;
    }
    if (code == 4) {  // BigQueryTableRow::try_from_tagged_cells (bigquery/encoding.rs:37-45) around validate_numeric_for_bigquery (validation.rs:20-35)
      const int32_t k = lib_error(c, ETLG_UnsupportedValueInDestination, "Cell validation failed for BigQuery compatibility");
      c->err_detail = "Cell at index " + std::to_string(col) + " failed validation";
      c->err.detail = c->err_detail.c_str();
      c->err.frame_index = (int64_t)ev;
      return k;
    }
    const int32_t k = lib_error(c, ETLG_ConversionError, code == 1 ? "NULL value for non-nullable ClickHouse column" : "Date out of ClickHouse Date32 range");
    c->err.frame_index = (int64_t)ev;
    return k;
  }
  if (total) {
    HIPCHK(c, blk_take(c, (size_t)total + 64, false, &rb->m.d_b, &rb->m.cap_b));
    j.out = (uint8_t*)rb->m.d_b;
    etlg_k_rowbinary(&j, nullptr, nullptr, 1, s);
  }
  const uint8_t* base_a = A; const uint8_t* base_b = (const uint8_t*)rb->m.d_b;
  if (!on_dev) {
    HIPCHK(c, blk_take(c, o_len + al((size_t)total) + 64, true, (void**)&rb->m.h, &rb->m.cap_h));
    HIPCHK(c, hipMemcpyAsync(rb->m.h, A, o_off + (n + 1) * 8, hipMemcpyDeviceToHost, s));
    if (total) HIPCHK(c, hipMemcpyAsync(rb->m.h + o_len, rb->m.d_b, (size_t)total, hipMemcpyDeviceToHost, s));
    base_a = rb->m.h; base_b = rb->m.h + o_len;
  }
  HIPCHK(c, hipStreamSynchronize(s));   // block S is freed on return
  if (!on_dev) { blk_give(c, c->gen, rb->m.d_a, rb->m.cap_a, false); blk_give(c, c->gen, rb->m.d_b, rb->m.cap_b, false); rb->m.d_a = rb->m.d_b = nullptr; }
  rb->v.n_rows = n; rb->v.n_bytes = (uint64_t)total;
  rb->v.row_event = (const uint64_t*)base_a; rb->v.row_offsets = (const int64_t*)(base_a + o_off); rb->v.bytes = total ? base_b : nullptr;
  *out = rb.release();
  return ETLG_OK;
}

int32_t etlg_rowbinary_view_get(const etlg_rowbinary* rb, etlg_rowbinary_view* out) {
  if (!rb || !out) return ETLG_InvalidArgument;
  *out = rb->v;
  return ETLG_OK;
}

void etlg_rowbinary_free(etlg_rowbinary* rb) {
  if (!rb) return;
  handoff_release(rb->m);
  delete rb;
}

}  // extern "C"

namespace {

// Copies a device-resident batch into host vectors and returns its OutSet to the pool.
int32_t download_batch(etlg_ctx* c, etlg_batch* b) {
  OutSet* os = b->dev;
  if (!os) return ETLG_OK;
  // The batch is complete when this runs (its result block has been read on the host), so the copies need no ordering against
  // the decode streams: they travel on a stream of their own and overlap the upload and decode of the batches issued after this
  // one (PCIe is full duplex: a host-to-host pipeline costs max(upload, download) per batch, not their sum).
  if (!c->d2h_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
  hipStream_t s = c->d2h_stream;
  etlg_batch_view& v = b->v;
  const size_t n = (size_t)v.n_events;
  // layout of the pinned block: 8-byte arrays first, then 4-byte, then bytes (every part 64-byte aligned)
  auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
  const size_t o_start = 0, o_commit = o_start + al(n * 8), o_ord = o_commit + al(n * 8), o_body = o_ord + al(n * 8);
  const size_t o_table = o_body + al(n * 8), o_slot = o_table + al(n * 4), o_kind = o_slot + al(n * 4), o_flags = o_kind + al(n);
  const size_t o_fixed = o_flags + al(n), o_heap = o_fixed + al((size_t)v.fixed_bytes), total = o_heap + al((size_t)v.heap_bytes) + 64;
  if (b->h_arena_cap < total) {
    if (b->h_arena) { c->harena_pool.emplace_back(b->h_arena, b->h_arena_cap); b->h_arena = nullptr; b->h_arena_cap = 0; }
    // smallest pooled block that fits, else a new one (rounded up so that similar batches can share it)
    size_t best = (size_t)-1;
    for (size_t i = 0; i < c->harena_pool.size(); i++)
      if (c->harena_pool[i].second >= total && (best == (size_t)-1 || c->harena_pool[i].second < c->harena_pool[best].second)) best = i;
    if (best != (size_t)-1) {
      b->h_arena = c->harena_pool[best].first; b->h_arena_cap = c->harena_pool[best].second;
      c->harena_pool.erase(c->harena_pool.begin() + (long)best);
    } else {
      const size_t cap = (total + (total >> 2) + 4095) & ~(size_t)4095;
      HIPCHK(c, hipHostMalloc((void**)&b->h_arena, cap, hipHostMallocDefault));
      b->h_arena_cap = cap;
    }
  }
  uint8_t* h = b->h_arena;
  if (n) {
    HIPCHK(c, hipMemcpyAsync(h + o_kind, os->kind.p, n, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(h + o_flags, os->flags.p, n, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(h + o_table, os->table.p, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(h + o_slot, os->slot.p, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(h + o_start, os->start.p, n * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(h + o_commit, os->commit.p, n * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(h + o_ord, os->ord.p, n * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(h + o_body, os->body.p, n * 8, hipMemcpyDeviceToHost, s));
  }
  if (v.fixed_bytes) HIPCHK(c, hipMemcpyAsync(h + o_fixed, os->fixed.p, (size_t)v.fixed_bytes, hipMemcpyDeviceToHost, s));
  if (v.heap_bytes) HIPCHK(c, hipMemcpyAsync(h + o_heap, os->heap.p, (size_t)v.heap_bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  c->out_pool.push_back(os);
  b->dev = nullptr;
  v.on_device = 0;
  v.ev_kind = h + o_kind; v.ev_flags = h + o_flags; v.ev_table_id = (const uint32_t*)(h + o_table); v.ev_schema_slot = (const uint32_t*)(h + o_slot);
  v.ev_start_lsn = (const uint64_t*)(h + o_start); v.ev_commit_lsn = (const uint64_t*)(h + o_commit);
  v.ev_tx_ordinal = (const uint64_t*)(h + o_ord); v.ev_body_off = (const uint64_t*)(h + o_body);
  v.fixed = h + o_fixed; v.heap = h + o_heap;
  return ETLG_OK;
}

// ------------------------------------------------------------ decode orchestration
// Per-frame scratch of the multi-pass kernels (context-shared, grow-only): (re)binds the pointers of `p`.
int32_t setup_scratch(etlg_ctx* c, DecParams& p) {
  const uint32_t nf = p.nframes, nblocks = p.nblocks;
  HIPCHK(c, c->d_tag.ensure(nf + 16)); HIPCHK(c, c->d_emit.ensure(nf + 16));
  HIPCHK(c, c->d_ffixed.ensure((size_t)nf * 4 + 16)); HIPCHK(c, c->d_fheap.ensure((size_t)nf * 4 + 16));
  HIPCHK(c, c->d_blk32.ensure((size_t)(nblocks + 1) * 4 * 3 + 64));
  HIPCHK(c, c->d_blk64.ensure((size_t)(nblocks + 1) * 8 * 5 + 64));
  p.f_tag = (uint8_t*)c->d_tag.p; p.f_emit = (uint8_t*)c->d_emit.p;
  p.f_fixed = (uint32_t*)c->d_ffixed.p; p.f_heap = (uint32_t*)c->d_fheap.p;
  p.blk_cnt = (uint32_t*)c->d_blk32.p; p.blk_last = p.blk_cnt + (nblocks + 1); p.blk_ev = p.blk_last + (nblocks + 1);
  p.blk_fixed = (uint64_t*)c->d_blk64.p; p.blk_heap = p.blk_fixed + (nblocks + 1); p.blk_payload = p.blk_heap + (nblocks + 1);
  return ETLG_OK;
}

// Side inputs of one batch: table states + the shared-table-cache timeline (`eps`: the epochs its own Relation / DDL frames
// create) + schema slots + the fixed-width plan's tables. Re-uploaded only when they change; a change behind pending ASYNC
// batches finishes those first (their kernels read the old copy).
int32_t build_side_inputs(etlg_ctx* c, etlg_batch* b, const std::vector<EpochRec>& eps) {
  SlowScope slow_scope_build_side_inputs(c, "build_side_inputs");
  DecParams& p = b->params;
  hipStream_t s = c->stream;
  auto point = [&](SideSet* ss, uint32_t n_tables, uint32_t n_epochs) {
    const uint8_t* d = (const uint8_t*)ss->dev.p;
    p.tables = (const DevTable*)(d + ss->o_tables); p.epochs = (const DevEpoch*)(d + ss->o_epochs); p.n_tables = n_tables; p.n_epochs = n_epochs;
    p.slots = (const DevSlot*)(d + ss->o_slots); p.cols = (const DevCol*)(d + ss->o_cols);
    p.n_slots = ss->n_slots; p.n_cols = ss->n_cols;
    if (p.flags & 2u) p.copy_slot = (int32_t)(std::lower_bound(c->last_live.begin(), c->last_live.end(), b->copy.slot) - c->last_live.begin());   // device index of the caller's slot
    side_use(b, ss);
  };
  if (eps.empty() && c->side_valid && !c->side_dirty && !c->slots_dirty && c->last_epochs.empty() && !b->have_snapshot && !b->copy.active) {
    // nothing the side inputs are built from has changed since the last upload (the common case: one call per batch)
    point(c->side_cur, (uint32_t)c->last_tables.size(), 0u);
    b->any_sync_done = c->last_any_sync_done;
    return ETLG_OK;
  }
  std::map<uint32_t, DevTable> tabs;
  auto get = [&](uint32_t id) -> DevTable& {
    auto it = tabs.find(id);
    if (it == tabs.end()) { DevTable t{}; t.table_id = id; t.init_slot = -1; it = tabs.emplace(id, t).first; }
    return it->second;
  };
  for (auto& kv : c->states) { DevTable& t = get(kv.first); t.state_kind = (uint32_t)kv.second.kind; t.state_lsn = kv.second.lsn; }
  const ControlState& cs0 = b->have_snapshot ? b->snapshot : c->cs;  // cache as of batch start
  for (auto& kv : cs0.cache) { DevTable& t = get(kv.first); t.init_kind = kv.second.kind; t.init_slot = kv.second.slot; }
  for (auto& e : eps) get(e.table_id);
  std::vector<DevTable> tv;
  std::vector<DevEpoch> ev;
  for (auto& kv : tabs) {
    DevTable t = kv.second;
    t.ep_begin = (uint32_t)ev.size();
    for (auto& e : eps) if (e.table_id == t.table_id) ev.push_back(e.ep);  // already in frame order
    t.ep_end = (uint32_t)ev.size();
    tv.push_back(t);
  }
  std::vector<int32_t> live;   // slots a frame of this batch can decode against
  for (auto& t : tv) if (t.init_kind == 2u && t.init_slot >= 0) live.push_back(t.init_slot);
  for (auto& e : ev) if (e.kind == 2u && e.slot >= 0) live.push_back(e.slot);
  if (b->copy.active && b->copy.slot >= 0) live.push_back(b->copy.slot);
  std::sort(live.begin(), live.end());
  live.erase(std::unique(live.begin(), live.end()), live.end());
  auto dev_index = [&](int32_t host_slot) -> int32_t {   // position in `live`; -1 stays -1
    if (host_slot < 0) return host_slot;
    return (int32_t)(std::lower_bound(live.begin(), live.end(), host_slot) - live.begin());
  };
  for (auto& t : tv) if (t.init_kind == 2u) t.init_slot = dev_index(t.init_slot);
  for (auto& e : ev) if (e.kind == 2u) e.slot = dev_index(e.slot);
  const bool same = c->side_valid && !c->slots_dirty && live == c->last_live && tv.size() == c->last_tables.size() && ev.size() == c->last_epochs.size() &&
                    (tv.empty() || !memcmp(tv.data(), c->last_tables.data(), tv.size() * sizeof(DevTable))) &&
                    (ev.empty() || !memcmp(ev.data(), c->last_epochs.data(), ev.size() * sizeof(DevEpoch)));
  if (!same) {  // table states, the cache timeline or the slots changed: a new upload, into a set no unfinished batch reads
    std::vector<DevSlot> ds;
    std::vector<DevCol> dc;
    build_slots(c, live, ds, dc);
    // ---- the fixed-width plan (plan.hip): tables the apply worker owns outright, Ready for the whole batch, whose
    //      replicated columns are all bool / int2 / int4 / int8 / oid
    std::vector<PlanTab> pt;
    std::vector<uint32_t> pc;
    if (ev.empty() && c->worker == ETLG_WORKER_APPLY) {
      for (const DevTable& t : tv) {  // tv is sorted by table id
        if (t.state_kind != ETLG_TS_READY || t.init_kind != 2u || t.init_slot < 0 || (size_t)t.init_slot >= live.size()) continue;
        const int32_t host_slot = live[(size_t)t.init_slot];   // (t.init_slot is the device index by now)
        if (host_slot < 0 || (size_t)host_slot >= c->slots.size()) continue;
        const SlotHost& sh = *c->slots[(size_t)host_slot];
        bool ok = sh.desc.n_cols > 0;
        for (auto& sc : sh.cols) {
          const int32_t k = sc.type_class;
          if (!(k == ETLG_TC_BOOL || k == ETLG_TC_I16 || k == ETLG_TC_I32 || k == ETLG_TC_I64 || k == ETLG_TC_U32)) ok = false;
        }
        if (!ok) continue;
        PlanTab e{};
        e.rel_id = t.table_id; e.slot = (uint32_t)host_slot; e.n_cols = sh.desc.n_cols; e.row_dwords = sh.desc.row_bytes_full / 4;   // (the plan writes the arena's id, it does not index the slot table)
        e.cols_base = (uint32_t)pc.size();
        for (auto& sc : sh.cols) pc.push_back((uint32_t)sc.type_class | ((uint32_t)(sc.nullable ? 1 : 0) << 8) | ((uint32_t)sc.off_full << 16));
        pt.push_back(e);
      }
    }
    SideSet* ss = nullptr;
    side_release(b);   // (a batch that is decoded again lets go of the set its first attempt read)
    for (SideSet* x : c->side_sets) if (x->users == 0) { ss = x; break; }
    if (!ss) { ss = new SideSet(); c->side_sets.push_back(ss); HIPCHK(c, hipEventCreateWithFlags(&ss->ready, hipEventDisableTiming)); }
    auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
    ss->o_tables = 0;
    ss->o_epochs = al(tv.size() * sizeof(DevTable) + 16);
    ss->o_slots = ss->o_epochs + al(ev.size() * sizeof(DevEpoch) + 16);
    ss->o_cols = ss->o_slots + al(ds.size() * sizeof(DevSlot) + 16);
    ss->o_ptabs = ss->o_cols + al(dc.size() * sizeof(DevCol) + 16);
    ss->o_pcols = ss->o_ptabs + al(pt.size() * sizeof(PlanTab) + 16);
    const size_t total = ss->o_pcols + al(pc.size() * 4 + 16);
    if (total > ss->h_cap) {
      if (ss->h) (void)hipHostFree(ss->h);
      ss->h = nullptr; ss->h_cap = 0;
      const size_t want = total + total / 2 + 4096;
      HIPCHK(c, hipHostMalloc((void**)&ss->h, want, hipHostMallocDefault));
      ss->h_cap = want;
    }
    HIPCHK(c, ss->dev.ensure(ss->h_cap));
    if (!tv.empty()) memcpy(ss->h + ss->o_tables, tv.data(), tv.size() * sizeof(DevTable));
    if (!ev.empty()) memcpy(ss->h + ss->o_epochs, ev.data(), ev.size() * sizeof(DevEpoch));
    if (!ds.empty()) memcpy(ss->h + ss->o_slots, ds.data(), ds.size() * sizeof(DevSlot));
    if (!dc.empty()) memcpy(ss->h + ss->o_cols, dc.data(), dc.size() * sizeof(DevCol));
    if (!pt.empty()) memcpy(ss->h + ss->o_ptabs, pt.data(), pt.size() * sizeof(PlanTab));
    if (!pc.empty()) memcpy(ss->h + ss->o_pcols, pc.data(), pc.size() * 4);
    HIPCHK(c, hipMemcpyAsync(ss->dev.p, ss->h, total, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipEventRecord(ss->ready, s));
    ss->synced = 1u << (s == c->stream2 && c->stream2 ? 1 : 0);
    ss->n_slots = (uint32_t)ds.size(); ss->n_cols = (uint32_t)dc.size();
    c->side_cur = ss;
    c->last_tables = tv; c->last_epochs = ev; c->side_valid = true;
    c->slots_dirty = false; c->last_live = live;
    c->n_plan_tabs = (uint32_t)pt.size();
    c->plan_max_row = 16;
    for (auto& e : pt) c->plan_max_row = std::max<uint32_t>(c->plan_max_row, e.row_dwords * 4);
    // every table state the batch can meet must be covered by the plan for it to be worth trying: a table that is not
    // eligible but owned (a TEXT column, say) would fail every batch that carries its rows
    c->plan_covers_all = !pt.empty();
    for (const DevTable& t : tv) if (t.state_kind != ETLG_TS_ABSENT && t.state_kind != ETLG_TS_OTHER) {
      bool found = false;
      for (auto& e : pt) if (e.rel_id == t.table_id) found = true;
      if (!found) c->plan_covers_all = false;
    }
  }
  point(c->side_cur, (uint32_t)tv.size(), (uint32_t)ev.size());
  b->any_sync_done = false;
  for (auto& t : tv) if (t.state_kind == ETLG_TS_SYNC_DONE) b->any_sync_done = true;
  c->last_any_sync_done = b->any_sync_done;
  if (!b->have_snapshot) c->side_dirty = false;   // built from the live control state
  return ETLG_OK;
}

// Output arrays (capacity bounds: one event per frame; rows bounded by the widest slot; truncate bodies by 2x frame bytes;
// heap by 2.5x input). Called again when the control path created wider slots.
int32_t setup_outputs(etlg_ctx* c, etlg_batch* b) {
  SlowScope slow_scope_setup_outputs(c, "setup_outputs");
  DecParams& p = b->params;
  if (!b->dev) b->dev = take_outset(c);
  OutSet* os = b->dev;
  const uint32_t nf = p.nframes;
  const size_t evcap = (size_t)nf + 16;
  const uint64_t fixed_cap = (uint64_t)nf * max_row_bytes(c) + 2 * (uint64_t)b->len + 64;
  const uint64_t heap_cap = std::min<uint64_t>(0xFFFFFFF0ull, (uint64_t)b->len * 5 / 2 + 64);
  HIPCHK(c, os->kind.ensure(evcap)); HIPCHK(c, os->flags.ensure(evcap));
  HIPCHK(c, os->table.ensure(evcap * 4)); HIPCHK(c, os->slot.ensure(evcap * 4));
  HIPCHK(c, os->start.ensure(evcap * 8)); HIPCHK(c, os->commit.ensure(evcap * 8));
  HIPCHK(c, os->ord.ensure(evcap * 8)); HIPCHK(c, os->body.ensure(evcap * 8));
  // The fixed arena's bound follows the widest slot, which a stream with DDL keeps widening (ALTER TABLE ADD COLUMN): grown to the
  // exact bound each time, every arena of the pool was re-allocated again and again (hipFree + hipMalloc synchronise the device:
  // 1.9 ms per cfg5 batch, ETLG_HOST_TIMES). The context remembers the largest bound it has seen with head room, and every arena
  // that has to grow goes there at once.
  if (fixed_cap > c->fixed_hint) c->fixed_hint = fixed_cap + fixed_cap / 2;
  if (os->fixed.cap < fixed_cap) HIPCHK(c, os->fixed.ensure(std::max<uint64_t>(fixed_cap, c->fixed_hint)));
  HIPCHK(c, os->heap.ensure(heap_cap));
  p.ev_kind = (uint8_t*)os->kind.p; p.ev_flags = (uint8_t*)os->flags.p; p.ev_table = (uint32_t*)os->table.p; p.ev_slot = (uint32_t*)os->slot.p;
  p.ev_start = (uint64_t*)os->start.p; p.ev_commit = (uint64_t*)os->commit.p; p.ev_ord = (uint64_t*)os->ord.p; p.ev_body = (uint64_t*)os->body.p;
  p.fixed = (uint8_t*)os->fixed.p; p.heap = (uint8_t*)os->heap.p; p.fixed_cap = fixed_cap; p.heap_cap = heap_cap;
  return ETLG_OK;
}

// Should this batch try the fixed-width plan first?
bool plan_wanted(etlg_ctx* c, const etlg_batch* b) {
  const DecParams& p = b->params;
  if (c->plan_mode == 0 || (c->fused_kernel >= 0 && c->fused_kernel != 3)) return false;  // ETLG_PLAN=0 / a forced generic kernel
  if (!c->n_plan_tabs || !c->plan_covers_all || p.n_epochs || (p.flags & 2u) || c->worker != ETLG_WORKER_APPLY) return false;
  if (p.nframes >= (1u << 29) || p.fixed_cap >= (1ull << 34) || c->fused_dbg) return false;   // descriptor: mark 30 bits, fixed dwords 32 bits
  if (c->plan_max_row > 512) return false;   // 64 rows of the widest table sit in LDS beside the staging window
  if (c->fused_kernel == 3) return true;
  if (c->plan_skip) { c->plan_skip--; return false; }  // backing off after a batch that did not conform
  return true;
}

// Look-back descriptor buffers: four in rotation. The launch of batch k uses buffer k mod 4 and zeroes the head of buffer
// (k + 2) mod 4, so the stream carries no memset between kernels (only when a buffer grows or a larger batch left a tail).
// Distance two, not one: batch k+1 may run BESIDE batch k on the second stream (it uses a buffer batch k-1 cleared, which
// completed before k+1 started), and the buffer k clears was last used by batch k-2, which completed before k started.
hipError_t sync_decode_streams(etlg_ctx* c) {
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess && c->stream2) e = hipStreamSynchronize(c->stream2);
  return e;
}
int32_t take_descriptors(etlg_ctx* c, size_t dbytes, uint8_t** cur_out, uint8_t** oth_out) {
  SlowScope slow_scope_take_descriptors(c, "take_descriptors");
  hipStream_t s = c->stream;
  if (dbytes > c->desc_half) {
    const size_t half = (dbytes * 2 + 4095) & ~(size_t)4095;
    HIPCHK(c, sync_decode_streams(c));   // earlier launches may still be using the old buffer
    HIPCHK(c, c->d_desc.ensure(half * 4));
    HIPCHK(c, hipMemsetAsync(c->d_desc.p, 0, half * 4, s));
    HIPCHK(c, hipStreamSynchronize(s));
    c->desc_half = half; for (size_t& d : c->desc_dirty) d = 0;
  }
  const uint32_t cur = c->desc_cur & 3u, oth = (cur + 2u) & 3u;
  uint8_t* dcur = (uint8_t*)c->d_desc.p + cur * c->desc_half;
  uint8_t* doth = (uint8_t*)c->d_desc.p + oth * c->desc_half;
  if (c->desc_dirty[cur]) { HIPCHK(c, hipMemsetAsync(dcur, 0, c->desc_dirty[cur], s)); c->desc_dirty[cur] = 0; }
  c->desc_dirty[cur] = dbytes;                                   // this launch writes it
  if (c->desc_dirty[oth] <= dbytes) c->desc_dirty[oth] = 0;       // ... and clears that much of the one after next
  c->desc_cur = (cur + 1u) & 3u;
  *cur_out = dcur; *oth_out = doth;
  return ETLG_OK;
}

// Enqueues ONE single-pass kernel over the batch: level 0 = the fixed-width plan (plan.hip), level 1 = the generic fused
// kernel (fused.hip) or, for wide frames, the column-parallel one (cells.hip).
int32_t enqueue_single(etlg_ctx* c, etlg_batch* b, int level) {
  SlowScope slow_scope_enqueue_single(c, "enqueue_single");
  const DecParams& p = b->params;
  const uint32_t nf = p.nframes;
  const uint64_t avg = (b->len + nf - 1) / nf;
  b->level = level;
  if (level == 0) {
    PlanParams& q = c->pq;
    q.ntiles = (nf + 63) / 64;
    uint64_t cap = 64 * avg * (100 + (uint64_t)c->plan_margin_pct) / 100 + 128;
    cap = std::min<uint64_t>((cap + 127) & ~127ull, 48 * 1024);
    q.rows_off = (uint32_t)std::max<uint64_t>(cap, 2048 + 64);        // the staging window (a tile read in place parks 64 x 32-byte bodies there)
    // rows of up to 8 dwords wait for the look-back inside their own frame's staged head; wider tables get a region of 64 rows
    q.lds_bytes = q.rows_off + (c->plan_max_row > 32 ? (uint32_t)(64ull * c->plan_max_row + 64) : 0u);
    q.n_tabs = c->n_plan_tabs; q.tabs = (const PlanTab*)((const uint8_t*)b->side->dev.p + b->side->o_ptabs); q.cols = (const uint32_t*)((const uint8_t*)b->side->dev.p + b->side->o_pcols);
    q.dbg = c->plan_dbg;
    q.max_row_dw = (c->plan_max_row + 3) / 4;
    const size_t per = 2 * ((size_t)q.ntiles + ((size_t)q.ntiles + 63) / 64);   // pairs: {agg, lsn}[ntiles] | {agg, lsn}[ngroups]
    const size_t dbytes = per * 8 + 64;
    uint8_t *dcur, *doth;
    { const int32_t rc = take_descriptors(c, dbytes, &dcur, &doth); if (rc != ETLG_OK) return rc; }
    q.desc = (unsigned long long*)dcur;
    q.d_clear = (unsigned long long*)doth; q.clear_words = (uint32_t)(dbytes / 8);
    launch(c, kPlan, p);
    b->used_fused = true; b->used_cells = false;
    return ETLG_OK;
  }
  FusedParams& q = c->fq;
  const uint64_t side = (uint64_t)p.n_tables * sizeof(DevTable) + (uint64_t)p.n_epochs * sizeof(DevEpoch) +
                        (uint64_t)p.n_slots * sizeof(DevSlot) + (uint64_t)p.n_cols * sizeof(DevCol);
  q.side_bytes = (side <= 32768 && !(c->fused_dbg & 16)) ? (uint32_t)((side + 15) & ~15ull) : 0u;
  // kernel choice: narrow frames -> one lane per frame, 256 frames per tile (k_fused); wide frames ->
  // 64 frames per tile with the waves spread over the columns (k_cells, schemas up to 32 columns: a second instantiation beyond 16)
  uint32_t widest = 1;
  for (int32_t li : c->last_live) widest = std::max<uint32_t>(widest, c->slots[(size_t)li]->desc.n_cols);
  const bool cells_ok = widest <= etlg_k_cells_maxc() && q.side_bytes != 0;  // k_cells keeps the side tables in LDS
  bool any_var = false;   // a table the batch may carry has TEXT / NUMERIC / ... columns: one lane per frame crawls on those
  for (int32_t li : c->last_live) for (auto& sc : c->slots[(size_t)li]->cols) {
    const int32_t k = sc.type_class;
    if (!(k == ETLG_TC_BOOL || k == ETLG_TC_I16 || k == ETLG_TC_I32 || k == ETLG_TC_I64 || k == ETLG_TC_U32 || k == ETLG_TC_UUID)) any_var = true;
  }
  int kernel = (avg <= 192 && !(any_var && cells_ok && avg > 96)) ? 0 : (cells_ok ? 2 : 1);  // 0 fused/256, 1 fused/64, 2 cells
  if (c->fused_kernel >= 0 && c->fused_kernel <= 2) kernel = c->fused_kernel == 2 && !cells_ok ? 1 : c->fused_kernel;
  const bool use_cells = kernel == 2;
  q.blk = kernel == 0 ? 256u : 64u;
  q.maxc = widest;
  // LDS window per tile: the average tile plus a margin; a tile that does not fit reads the input in place
  uint64_t cap = kernel == 1 ? (uint64_t)q.blk * avg * 5 / 4 + 2048 : (uint64_t)q.blk * avg * 9 / 8 + 1024;
  if (const char* lm = getenv("ETLG_LDS_MARGIN_PCT")) cap = (uint64_t)q.blk * avg * (100 + (uint64_t)atoi(lm)) / 100 + 1024;
  cap = (cap + 255) & ~255ull;
  if (use_cells) {
    cap = std::max<uint64_t>(cap + etlg_k_cells_table_bytes(widest), etlg_k_cells_lds_floor(widest));
    // LDS decides how many workgroups share a CU (160 KB; the register file allows four): the window takes whatever the
    // allocation can grow by without losing one, so fewer tiles overflow it
    const uint64_t stat = etlg_k_cells_static_lds(widest);
    const uint64_t wgs = std::max<uint64_t>(1, std::min<uint64_t>(4, (160 * 1024) / (cap + q.side_bytes + stat)));
    cap = std::max<uint64_t>(cap, ((160 * 1024) / wgs - 512 - stat - q.side_bytes) & ~255ull);
  }
  cap = std::min<uint64_t>(cap + q.side_bytes, 150 * 1024);
  q.lds_bytes = (uint32_t)cap;
  q.seq_lookback = (b->any_sync_done || (c->fused_dbg & 32)) ? 1u : 0u;
  q.ntiles = (nf + q.blk - 1) / q.blk;
  q.in_aligned = ((uintptr_t)p.in & 15) == 0;
  q.dbg = c->fused_dbg;
  const size_t ngroups = (q.ntiles + 63) / 64;
  const size_t per = (size_t)q.ntiles + ngroups;  // tile descriptors followed by group descriptors
  const size_t dbytes = per * 8 * 3 + 64;
  uint8_t *dcur, *doth;
  { const int32_t rc = take_descriptors(c, dbytes, &dcur, &doth); if (rc != ETLG_OK) return rc; }
  q.d_txn = (unsigned long long*)dcur; q.d_outa = q.d_txn + per; q.d_outb = q.d_outa + per;
  q.ticket = (uint32_t*)(q.d_outb + per);
  q.d_clear = (unsigned long long*)doth; q.clear_words = (uint32_t)(dbytes / 8);
  launch(c, use_cells ? kCells : kFused, p);
  b->used_fused = true;
  b->used_cells = use_cells;
  return ETLG_OK;
}

// The control pre-pass, device half: classify + transaction scan + compaction of the R / M frames (k_ctrl_list also gathers their
// bytes), then the result block, the head of the list and the head of the gathered bytes to pinned memory WITHOUT asking for their
// sizes first (one round trip instead of three; what does not fit is fetched afterwards). `ahead`: on the control stream, beside the
// decode of the batch before (etlg_decode); otherwise on the context's stream, collected at once (run_control_pass).
int32_t ctl_begin(etlg_ctx* c, etlg_batch* b, DecParams& p, hipStream_t s, bool ahead) {
  SlowScope slow_scope_ctl_begin(c, "ctl_begin");
  struct StreamSwitch { etlg_ctx* c; hipStream_t saved; ~StreamSwitch() { c->stream = saved; } } sw{c, c->stream};
  c->stream = s;
  const uint32_t nf = p.nframes;
  if (!c->h_ctl_list) {
    HIPCHK(c, hipHostMalloc((void**)&c->h_ctl_list, sizeof(CtrlFrame) * etlg_ctx::kCtlListCap, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void**)&c->h_ctl_stage, etlg_ctx::kCtlStageCap, hipHostMallocDefault));
  }
  if (ahead) {
    if (c->mp_tail_set) { HIPCHK(c, hipStreamWaitEvent(s, c->mp_tail, 0)); c->mp_tail_set = false; }   // the multi-pass kernels share the per-frame scratch
    if (c->res_pool.empty()) { DevResult* r = nullptr; HIPCHK(c, hipHostMalloc((void**)&r, sizeof(DevResult), hipHostMallocDefault)); c->res_pool.push_back(r); }
    b->h_ctl = c->res_pool.back(); c->res_pool.pop_back();
    if (c->ev_pool.empty()) { hipEvent_t e = nullptr; HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_pool.push_back(e); }
    b->ctl_ev = c->ev_pool.back(); c->ev_pool.pop_back();
    HIPCHK(c, hipMemcpyAsync(p.res, c->d_init_ring, sizeof(DevResult), hipMemcpyDeviceToDevice, s));
  }
  { const int32_t rc = setup_scratch(c, p); if (rc != ETLG_OK) return rc; }
  launch(c, 0, p);
  launch(c, 1, p);
  HIPCHK(c, c->d_ctrl.ensure((size_t)nf * sizeof(CtrlFrame) + 64));
  p.ctrl = (CtrlFrame*)c->d_ctrl.p; p.ctrl_cap = nf;
  {  // room for the control frames' bytes: a batch rarely carries more than a few hundred KB of them
    size_t want = std::min<size_t>(std::max<size_t>(b->len / 16, 64 << 10), 8 << 20);
    if (c->ctrl_stage_cap_test) want = c->ctrl_stage_cap_test;   // ETLG_CTRL_STAGE_CAP (tests): a staging buffer too small for the batch's control frames
    HIPCHK(c, c->d_ctrl_stage.ensure(want));
    p.ctrl_stage = (uint8_t*)c->d_ctrl_stage.p; p.ctrl_stage_cap = (uint32_t)want;
  }
  launch(c, 2, p);
  HIPCHK(c, hipMemcpyAsync(ahead ? b->h_ctl : b->h_res, p.res, sizeof(DevResult), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(c->h_ctl_list, c->d_ctrl.p, (size_t)std::min<uint32_t>(nf, etlg_ctx::kCtlListCap) * sizeof(CtrlFrame), hipMemcpyDeviceToHost, s));
  if (b->in_dev) HIPCHK(c, hipMemcpyAsync(c->h_ctl_stage, c->d_ctrl_stage.p, std::min<size_t>(p.ctrl_stage_cap, etlg_ctx::kCtlStageCap), hipMemcpyDeviceToHost, s));
  if (ahead) { HIPCHK(c, hipEventRecord(b->ctl_ev, s)); b->ctl_started = true; }
  return ETLG_OK;
}

// The control pre-pass of one batch (device half above, unless it ran ahead), then the host control plane (handle_relation /
// handle_ddl) in frame order. Fills b->ctrl / ctrl_raw, `eps`, the host error of the batch.
int32_t run_control_pass(etlg_ctx* c, etlg_batch* b, std::vector<EpochRec>& eps) {
  SlowScope slow_scope_run_control_pass(c, "run_control_pass");
  hipStream_t s = c->stream;
  const uint32_t nf = b->params.nframes;
  ht_start(c);
  b->snapshot = c->cs; b->have_snapshot = true; b->snap_gen = c->cs_gen;
  b->ctrl.clear(); b->ctrl_raw.clear();
  b->host_err_code = 0; b->host_err_frame = -1; b->host_err_rank = 0;
  b->ctrl_done = true;
  if (!nf) return ETLG_OK;
  const bool ahead = b->ctl_started;
  DecParams& p = ahead ? b->ctl_params : b->params;
  if (ahead) {
    HIPCHK(c, hipEventSynchronize(b->ctl_ev));
    s = c->ctl_stream;   // what did not fit the pinned heads is fetched on the stream the pre-pass ran on
  } else {
    { const int32_t rc = ctl_begin(c, b, p, s, false); if (rc != ETLG_OK) return rc; }
    HIPCHK(c, hipStreamSynchronize(s));
  }
  ht_mark(c, 0);
  const DevResult& cr = ahead ? *b->h_ctl : *b->h_res;
  const uint32_t nctrl = std::min<uint32_t>(cr.n_ctrl, p.ctrl_cap);
  if (!nctrl) return ETLG_OK;
  std::vector<CtrlFrame>& ctrl = b->ctrl;
  ctrl.resize(nctrl);
  const uint32_t nhead = std::min<uint32_t>(nctrl, etlg_ctx::kCtlListCap);
  memcpy(ctrl.data(), c->h_ctl_list, (size_t)nhead * sizeof(CtrlFrame));
  if (nctrl > nhead) HIPCHK(c, hipMemcpy(ctrl.data() + nhead, (const CtrlFrame*)c->d_ctrl.p + nhead, (size_t)(nctrl - nhead) * sizeof(CtrlFrame), hipMemcpyDeviceToHost));
  std::sort(ctrl.begin(), ctrl.end(), [](const CtrlFrame& a, const CtrlFrame& b2) { return a.frame < b2.frame; });
  ht_mark(c, 1);
  // the frames' bytes: already on the host, or the gathered copy k_ctrl_list left in the staging buffer (its head is in pinned
  // memory already); a frame that did not fit there is fetched from the input by itself
  std::vector<uint8_t> stage, extra;
  std::vector<size_t> at(nctrl + 1, 0);
  for (uint32_t i = 0; i < nctrl; i++) at[i + 1] = at[i] + (ctrl[i].stage_off == 0xFFFFFFFFu ? ctrl[i].o1 - ctrl[i].o0 : 0u);
  const uint8_t* staged_bytes = c->h_ctl_stage;
  if (b->in_dev) {
    const uint32_t staged = std::min<uint32_t>(cr.ctrl_bytes, p.ctrl_stage_cap);
    bool wait = false;
    if (staged > etlg_ctx::kCtlStageCap) {   // more gathered bytes than the pinned head holds: one copy of everything
      stage.resize((size_t)staged + 16);
      HIPCHK(c, hipMemcpyAsync(stage.data(), c->d_ctrl_stage.p, staged, hipMemcpyDeviceToHost, s));
      staged_bytes = stage.data(); wait = true;
    }
    extra.resize(at[nctrl] + 16);
    for (uint32_t i = 0; i < nctrl; i++)
      if (ctrl[i].stage_off == 0xFFFFFFFFu && ctrl[i].o1 > ctrl[i].o0) {
        HIPCHK(c, hipMemcpyAsync(extra.data() + at[i], b->dev_in + ctrl[i].o0, ctrl[i].o1 - ctrl[i].o0, hipMemcpyDeviceToHost, s));
        wait = true;
      }
    if (wait) HIPCHK(c, hipStreamSynchronize(s));
  }
  ht_mark(c, 2);
  for (uint32_t i = 0; i < nctrl; i++) {
    const CtrlFrame& cf = ctrl[i];
    const uint8_t* fr = !b->in_dev ? b->host_in + cf.o0 : cf.stage_off != 0xFFFFFFFFu ? staged_bytes + cf.stage_off : extra.data() + at[i];
    const size_t flen = cf.o1 - cf.o0;
    b->ctrl_raw.emplace_back(fr, fr + flen);
    // classify guaranteed 'd' len 'w' hdr tag: body starts at +31
    uint64_t wal_start = 0;
    for (int k = 0; k < 8; k++) wal_start = wal_start << 8 | fr[6 + k];
    HostErr he = cf.tag == 'R' ? handle_relation(c, cf, fr + 31, flen - 31, eps) : handle_ddl(c, cf, wal_start, fr + 31, flen - 31, eps);
    if (he.code) { b->host_err_code = he.code; b->host_err_frame = cf.frame; b->host_err_rank = he.rank; b->params.host_err_frame = cf.frame; break; }
  }
  ht_mark(c, 3);
  return ETLG_OK;
}

// The path that knows about control frames: control pre-pass (unless the caller asserted there are none), side inputs with
// the batch's own epochs, then the generic single-pass kernel or — forced, oversized, or behind a host error — the
// multi-pass kernels. Synchronous callers only (immediate carry).
int32_t standard_path(etlg_ctx* c, etlg_batch* b) {
  SlowScope slow_scope_standard_path(c, "standard_path");
  DecParams& p = b->params;
  const uint32_t nf = p.nframes;
  std::vector<EpochRec> eps;
  p.flags &= ~1u;
  if (b->user_no_ctrl) p.flags |= 1u;
  else { const int32_t rc = run_control_pass(c, b, eps); if (rc != ETLG_OK) return rc; c->path_n[6]++; b->n_slots_view = c->slots.size(); }
  b->eps_saved = eps;
  ht_start(c);
  { const int32_t rc = build_side_inputs(c, b, eps); if (rc != ETLG_OK) return rc; }
  ht_mark(c, 4);
  { const int32_t rc = setup_outputs(c, b); if (rc != ETLG_OK) return rc; }
  ht_mark(c, 5);
  if (nf && !b->host_err_code && !c->force_multipass && b->len < (1ull << 31)) { const int32_t rc = enqueue_single(c, b, 1); ht_mark(c, 6); return rc; }
  if (p.carry) {
    // chained to a batch in flight and in need of the multi-pass kernels (the host control plane failed on one of its frames): they
    // take the carried state from the host. The batch is marked "did not run" — the batches behind it stop at that — and is decoded
    // when it is synced, from the exact state (finish_batch, the forced re-run).
    DevResult poison = *c->h_init; poison.fused_fail = 8u;
    if (!c->h_poison) { HIPCHK(c, hipHostMalloc((void**)&c->h_poison, sizeof(DevResult), hipHostMallocDefault)); *c->h_poison = poison; }
    HIPCHK(c, hipMemcpyAsync(b->d_res_blk, c->h_poison, sizeof(DevResult), hipMemcpyHostToDevice, c->stream));
    b->level = 1; b->used_fused = true; b->used_cells = false;
    return ETLG_OK;
  }
  { const int32_t rc = setup_scratch(c, p); if (rc != ETLG_OK) return rc; }
  launch_multipass(c, p, b->ctrl_done && nf != 0 && !b->ctl_started);
  b->level = 2; b->used_fused = false; b->used_cells = false;
  return ETLG_OK;
}

// Finishes every pending ASYNC batch, oldest first.
int32_t drain_pending(etlg_ctx* c) {
  SlowScope slow_scope_drain_pending(c, "drain_pending");
  while (!c->pending.empty()) { const int32_t rc = finish_batch(c, c->pending.front()); (void)rc; }
  return ETLG_OK;
}

// Finishes one batch: waits for its result block; when the optimistic attempt did not hold, decodes the batch again on
// the next path (plan -> generic single pass -> control path / multi-pass exact error cut); resolves device vs host
// error, commits or rolls back the control-plane state, updates the carried transaction state of the context and (for
// host output) copies the arenas back. Batches finish in issue order.
int32_t finish_batch(etlg_ctx* c, etlg_batch* b) {
  SlowScope slow_scope_finish_batch(c, "finish_batch");
  hipStream_t s = c->stream;
  if (b->deferred) {  // its boundary scan is still in flight: collect it and enqueue the decode first
    const int32_t rc = flush_deferred(c);
    if (rc != ETLG_OK) return rc;   // the batch is finished, with that error
  }
  if (b->pending) {  // must be the oldest pending batch
    if (c->pending.empty() || c->pending.front() != b) return lib_error(c, ETLG_InvalidArgument, "ASYNC batches must be synced in issue order");
    c->pending.erase(c->pending.begin());
  }
  auto fail_hip = [&](hipError_t e) {
    b->pending = false; b->finished = true;
    b->rc = lib_error(c, ETLG_DeviceError, hipGetErrorString(e));
    b->err = c->err; b->err_detail.clear();
    return b->rc;
  };
#define FB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail_hip(e_); } while (0)
#define FB_RC(call) do { const int32_t rc_ = (call); if (rc_ != ETLG_OK) { b->pending = false; b->finished = true; b->rc = rc_; b->err = c->err; return rc_; } } while (0)
  ht_start(c);
  if (b->done) FB_HIP(hipEventSynchronize(b->done));   // batches queued behind this one keep running
  else FB_HIP(hipStreamSynchronize(s));
  ht_mark(c, 7);
  bool redone_mp = false;
  bool forced = b->force_rerun;   // a batch before this one in the chain was decoded again: this one started from a state that was not final
  b->force_rerun = false;
  for (int guard = 0; guard < 8; guard++) {
    const DevResult& r0 = *b->h_res;
    const bool failed = forced || r0.first_err != kNoErr || r0.fused_fail;
    const bool ctrl_hint = r0.first_err != kNoErr && (uint32_t)(r0.first_err & 0xFF) == ETLG_E_CTRL_HINT && !b->user_no_ctrl && !b->ctrl_done;
    if (!failed || (b->level == 2 && !ctrl_hint)) break;
    // ---- decode again. Every earlier batch is finished, so the host's carried state is exact: no device chaining.
    DecParams& p = b->params;
    p.carry = nullptr;
    p.flags &= ~16u;
    p.in_txn = c->in_txn; p.final_lsn = c->final_lsn; p.next_ord = c->next_ord;
    const uint32_t ff = forced ? 8u : r0.fused_fail;
    forced = false;
    if (c->ctl_stream) FB_HIP(hipStreamSynchronize(c->ctl_stream));   // (a pre-pass running ahead shares the scratch and the pinned heads with what follows)
    if (!c->pending.empty()) {   // batches behind this one may be running beside it (second stream) and chained to a result that is being replaced
      FB_HIP(sync_decode_streams(c));
      for (etlg_batch* pb : c->pending) { pb->force_rerun = true; if (pb->deferred) pb->ctl_started = false; }   // (a pre-pass that ran ahead started from a state that was not final)
    }
    FB_HIP(hipMemcpyAsync(b->d_res_blk, c->d_init_ring, sizeof(DevResult), hipMemcpyDeviceToDevice, s));
    if (b->copy.active) launch_copy(c, b->copy, p);
    if (ff & 8u) {  // the batch before this one failed, so this one never ran: same path again, now from the right state
      c->path_n[7]++;
      if (b->ctl_async) {
        // it took the control path: its control pass ran against a history that has changed. Back to the state before the batch —
        // its own snapshot, unless a rollback since then has already discarded everything behind the failed batch — and again.
        if (b->have_snapshot && b->snap_gen == c->cs_gen) { c->cs = b->snapshot; c->slots.resize(b->snapshot.n_slots); c->slots_dirty = true; c->side_dirty = true; }
        b->ctrl_done = false; b->ctl_started = false; b->have_snapshot = false;
        FB_RC(standard_path(c, b));
      } else {
        FB_RC(build_side_inputs(c, b, std::vector<EpochRec>()));
        FB_RC(enqueue_single(c, b, b->level));
      }
    } else if (b->level == 0) {  // the fixed-width plan did not cover the batch: generic kernel, and back off
      c->path_n[5]++;
      c->plan_skip = c->plan_penalty; c->plan_penalty = std::min<uint32_t>(c->plan_penalty * 2, 4096); c->plan_streak = 0;
      FB_RC(build_side_inputs(c, b, std::vector<EpochRec>()));
      FB_RC(enqueue_single(c, b, 1));
    } else if (ctrl_hint && !(ff & 1u)) {
      FB_RC(standard_path(c, b));  // a Relation / DDL frame: the control path
    } else {  // an error (or a look-back give-up): the multi-pass kernels know the exact cut at the failing frame
      FB_RC(build_side_inputs(c, b, b->eps_saved));
      FB_RC(setup_scratch(c, p));
      launch_multipass(c, p, false);
      b->level = 2; b->used_fused = false;
      c->path_n[3]++; redone_mp = true;
    }
    FB_HIP(hipMemcpyAsync(b->h_res, b->d_res_blk, sizeof(DevResult), hipMemcpyDeviceToHost, s));
    FB_HIP(hipStreamSynchronize(s));
  }
  if (b->level == 0) { c->path_n[4]++; if (++c->plan_streak >= 16) c->plan_penalty = 4; }
  else if (b->used_fused) c->path_n[b->used_cells ? 1 : 0]++;
  else if (!redone_mp) c->path_n[2]++;
  DevResult r = *b->h_res;
  if (b->used_fused) for (int k = 0; k < 3; k++) { r.payload[k] = 0; for (int sh = 0; sh < 32; sh++) r.payload[k] += r.pay_shard[sh][k]; }
  for (int i = 0; i < 12; i++) c->last_dbg[i] = r.dbg_t[i];
  c->res_pool.push_back(b->h_res);
  b->h_res = nullptr;
  // ---- first error: device (frame, rank) vs host control plane (frame, rank)
  int32_t code = 0; int64_t frame = -1; uint32_t rank = 0xFF;
  if (r.first_err != kNoErr) { frame = (int64_t)(r.first_err >> 16); rank = (uint32_t)((r.first_err >> 8) & 0xFF); code = (int32_t)(r.first_err & 0xFF); }
  if (b->host_err_code) {
    const int64_t hf = b->host_err_frame;
    if (frame < 0 || hf < frame || (hf == frame && b->host_err_rank < rank)) { frame = hf; rank = b->host_err_rank; code = b->host_err_code; }
  }
  // ---- control-plane state: keep only effects of frames before the failing one
  if (code && b->have_snapshot) {
    bool later = false;
    for (auto& cf : b->ctrl) if ((int64_t)cf.frame >= frame) later = true;
    if (later || b->host_err_code) {
      // roll back, then replay the prefix (rare path; errors end the stream anyway)
      c->cs = b->snapshot;
      c->cs_gen++;
      c->slots.resize(b->snapshot.n_slots);
      c->slots_dirty = true; c->side_dirty = true;
      // Effects of the control frames before `frame` are re-applied from the copies of their bytes kept
      // by the first pass (the input itself may be device-resident, or have come without a sidecar).
      std::vector<EpochRec> eps;
      for (size_t i = 0; i < b->ctrl.size() && i < b->ctrl_raw.size(); i++) {
        const CtrlFrame& cf = b->ctrl[i];
        if ((int64_t)cf.frame >= frame) break;
        const uint8_t* fr = b->ctrl_raw[i].data();
        const size_t flen = b->ctrl_raw[i].size();
        uint64_t wal_start = 0;
        for (int k = 0; k < 8; k++) wal_start = wal_start << 8 | fr[6 + k];
        if (cf.tag == 'R') (void)handle_relation(c, cf, fr + 31, flen - 31, eps); else (void)handle_ddl(c, cf, wal_start, fr + 31, flen - 31, eps);
      }
      b->n_slots_view = c->slots.size();
    }
  }
  if (!b->user_no_ctrl) c->last_had_ctrl = b->ctrl_done && !b->ctrl.empty();
  b->have_snapshot = false;
  b->snapshot = ControlState{};
  b->ctrl_raw.clear();
  b->eps_saved.clear();
  c->in_txn = r.out_in_txn != 0; c->final_lsn = r.out_final_lsn; c->next_ord = r.out_next_ord;

  etlg_batch_view& v = b->v;
  v.n_events = r.n_events; v.n_frames = code ? (uint64_t)frame : r.n_frames;
  v.fixed_bytes = r.fixed_bytes; v.heap_bytes = r.heap_bytes;
  for (int i = 0; i < 3; i++) v.payload_bytes[i] = r.payload[i];
  OutSet* os = b->dev;
  v.on_device = 1;
  v.ev_kind = (const uint8_t*)os->kind.p; v.ev_flags = (const uint8_t*)os->flags.p; v.ev_table_id = (const uint32_t*)os->table.p; v.ev_schema_slot = (const uint32_t*)os->slot.p;
  v.ev_start_lsn = (const uint64_t*)os->start.p; v.ev_commit_lsn = (const uint64_t*)os->commit.p; v.ev_tx_ordinal = (const uint64_t*)os->ord.p; v.ev_body_off = (const uint64_t*)os->body.p;
  v.fixed = (const uint8_t*)os->fixed.p; v.heap = (const uint8_t*)os->heap.p;
  b->pending = false; b->finished = true;
  side_release(b);   // its kernels are done: the set may take the next change of the side inputs
  if (!b->out_dev) FB_RC(download_batch(c, b));
  fill_view_common(b);
  clear_error(c);
  b->rc = code ? set_error(c, code, frame) : (int32_t)ETLG_OK;
  b->err = c->err; b->err_detail = c->err_detail;
  return b->rc;
#undef FB_HIP
#undef FB_RC
}

}  // namespace
