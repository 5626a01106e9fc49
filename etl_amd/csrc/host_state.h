// Context / batch / pool structures of the host side of libetl_gfx950.so and the launchers of the kernels (host.cpp and its parts).
// Part of ONE translation unit: host.cpp includes this header and the three .inc files below it in order; they share the
// context structure and the helpers of the anonymous namespace.
#pragma once
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
extern "C" void etlg_k_shard_cuts(const uint8_t* tags, const uint32_t* offs, uint32_t nframes, uint32_t n_shards, uint32_t* cuts, hipStream_t s);
extern "C" void etlg_k_ctl_span(const uint8_t* tags, uint32_t nframes, const uint32_t* list, uint32_t n, uint32_t* span, hipStream_t s);
extern "C" void etlg_k_ctl_gather(const uint8_t* in, const uint32_t* offs, const uint32_t* frames, uint32_t nkeep, uint32_t* lens, const uint32_t* out_offs, uint8_t* out, hipStream_t s);
extern "C" void etlg_k_launch_fused(int blk, const DecParams* p, const void* q, hipStream_t s);
extern "C" int etlg_k_fused_set_lds(void);
extern "C" void etlg_k_launch_bounds(const uint8_t* in, uint64_t len, uint32_t* offs, uint32_t offs_cap, void* scratch, uint32_t* result, uint32_t* hints, uint32_t* hflag,
                                     int sequential, hipStream_t s);
extern "C" size_t etlg_k_bounds_scratch_bytes(size_t ntiles);
extern "C" uint32_t etlg_k_bounds_tile_bytes(void);
extern "C" uint32_t etlg_k_copy_bytes_per_row(uint32_t ncols);
extern "C" int etlg_k_copy_set_lds(void);
extern "C" void etlg_k_launch_copy(const uint8_t* rows, const uint32_t* row_offs, uint32_t nrows, uint64_t rows_len, uint32_t ncols,
                                   uint32_t rel_id, uint8_t* out, uint32_t* out_offs, uint32_t lds_bytes, int lane_per_byte, const DecParams* dec, hipStream_t s);
extern "C" void etlg_k_launch_cells(const DecParams* p, const void* q, hipStream_t s);
extern "C" void etlg_k_launch_plan(const DecParams* p, const void* q, hipStream_t s);
extern "C" void etlg_k_launch_plan_pre(const DecParams* p, const void* q, hipStream_t s);
extern "C" int etlg_k_plan_set_lds(void);
extern "C" void etlg_k_col_select(const void* sel, hipStream_t s);
extern "C" void etlg_k_col_fixed(const void* job, hipStream_t s);
extern "C" uint32_t etlg_k_col_pack_max(void);
extern "C" void etlg_k_col_fixed_pack(const void* jobs, uint32_t n, hipStream_t s);
extern "C" void etlg_k_col_var_pack(const void* jobs, uint32_t n, unsigned long long* const* blk, int64_t* const* offs, unsigned long long* const* tot, int step, hipStream_t s);
extern "C" void etlg_k_scan_lens(const uint32_t* lens, uint64_t n, unsigned long long* blk, int64_t* offsets, hipStream_t s);
extern "C" void etlg_k_col_list(const void* job, unsigned long long* blk, int64_t* offsets, int step, hipStream_t s);
extern "C" void etlg_k_size_hints(const void* job, hipStream_t s);
extern "C" void etlg_k_rowbinary(const void* job, unsigned long long* blk, int64_t* offsets, unsigned long long* tot, int step, hipStream_t s);
extern "C" void etlg_k_col_var(const void* job, unsigned long long* blk, int64_t* offsets, int step, hipStream_t s);
extern "C" int etlg_k_cells_set_lds(void);
extern "C" uint32_t etlg_k_cells_table_bytes(uint32_t maxc);
extern "C" uint32_t etlg_k_copy_cells_table_bytes(uint32_t maxc);
extern "C" uint32_t etlg_k_copy_cells_lds(uint32_t maxc, uint32_t window);
extern "C" void etlg_k_launch_copy_cells(const DecParams* p, const void* q, hipStream_t s);
extern "C" uint32_t etlg_k_cells_maxc(void);
extern "C" void etlg_k_launch_rows(const DecParams* p, const void* q, hipStream_t s);
extern "C" void etlg_k_finish(const void* job, unsigned long long* blk, int64_t* offsets, int step, hipStream_t st);
extern "C" int etlg_k_rows_set_lds(void);
extern "C" int etlg_k_rows_waves(void);
extern "C" int etlg_k_rows_waves_per_simd(void);
extern "C" int etlg_k_rows_occupancy(uint32_t lds_bytes);
extern "C" uint32_t etlg_k_rows_table_bytes(uint32_t maxh_old, uint32_t maxh, uint32_t maxc, uint32_t cf);
extern "C" uint32_t etlg_k_rows_static_lds(void);
extern "C" uint32_t etlg_k_rows_max_cols(void);
extern "C" uint32_t etlg_k_rows_max_heap_cols(void);
extern "C" uint32_t etlg_k_cells_lds_floor(uint32_t maxc);
extern "C" uint32_t etlg_k_cells_static_lds(uint32_t maxc);
extern "C" uint32_t etlg_k_copy_cells_static_lds(uint32_t maxc);

constexpr int kFused = 7;  // profiling slot of the fused kernel
constexpr int kCells = 8;  // ... of the column-parallel kernel (cells.hip)
constexpr int kBounds = 9; // ... of the record-boundary scan (scan.hip)
constexpr int kCopy = 10;  // ... of the table-copy row splitter (copy.hip)
constexpr int kPlan = 11;  // ... of the fixed-width plan (plan.hip)
constexpr int kCopyCells = 12;  // ... of the table-copy rows -> arena kernel (cells.hip, k_cells<.., COPYK>)
constexpr int kPlanPre = 13;    // ... of the plan's sidecar pre-pass (plan.hip, k_plan_pre)
constexpr int kRows = 14;       // ... of the row-synchronous kernel (rows.hip)
constexpr int kProfSlots = 15;

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
  std::vector<uint8_t> pk;   // per replicated column: a primary-key column of the stored schema (ColumnSchema::primary_key)
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
struct ScanJob { const uint8_t* d_in = nullptr; size_t len = 0; hipStream_t s = nullptr; DevBuf* offs = nullptr; size_t cap = 0; uint8_t* cur = nullptr; uint32_t* res = nullptr; };
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
  bool direct = false;      // first attempt: rows -> arena in one kernel (k_copy_cells); a batch that fails there is decoded again through the frames
  uint64_t syn_len = 0;     // bytes of the Insert frames the rows rewrite to (sizes d_out and the arenas)
  bool async = false;       // etlg_copy_decode with ETLG_F_ASYNC: enqueued only, finished by etlg_batch_sync (rows -> arena kernel only)
  // ASYNC with host input: the rows and their offsets were uploaded into a device block of the batch's own, on the copy stream; the
  // batch adopts both in etlg_decode (whatever is still set afterwards is given back by etlg_copy_decode)
  void* stage_blk = nullptr; size_t stage_cap = 0; hipEvent_t h2d_done = nullptr;
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
  // the plan's sidecar pre-pass (k_plan_pre): tile prefixes of a batch, four buffers in rotation (like the look-back descriptors)
  static constexpr uint32_t kPreBufs = 4;
  DevBuf d_pre; size_t pre_half = 0; uint32_t pre_seq = 0;
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
  bool debug_invariants = false;  // ETLG_DEBUG_INVARIANTS=1: check_invariants (host_orchestrate.inc) at every entry point, abort on a violation
  bool prof_serial = false;      // etlg_ctx_profile(ctx, 2): kernels timed one at a time (no second stream), for per-kernel durations
  unsigned long long overlapped = 0;   // debugging aid: batches launched beside their predecessor
  // result blocks: a ring re-initialised once per lap with one copy
  static constexpr uint32_t kResRing = 32;
  uint64_t scan_last_nf = 0, scan_last_len = 0;   // frames and bytes of the last batch whose boundaries were scanned on the device: the next one's grids are sized by its bytes per frame (+ 25 %)
  bool scan_chain_mode = true;   // ETLG_SCAN_CHAIN=0: a batch without a sidecar always waits for its frame count on the host before its decode is enqueued (round 5)
  uint64_t scan_chained_n = 0, scan_chain_redone = 0;
  uint64_t chain_reissued = 0;   // ASYNC batches enqueued again behind a batch that was decoded again (finish_batch, reissue_successors)
  bool chain_reissue = false;    // ETLG_CHAIN_REISSUE=1: the successors of a batch that was decoded again are enqueued again, chained to its new result
                                 // (reissue_successors). Built and measured in round 6 — it LOST on the delete-in-every-10th-batch leg (548 against 905 GB/s:
                                 // every give-up re-runs the whole window on one stream) — so the default stays each successor decoded again at its own sync
  bool chain_spare = true;       // ETLG_CHAIN_SPARE=0: the batches behind a plan batch that was decoded again are always decoded again (the rule before round 6's last session)
  uint64_t chain_spared = 0;     // second attempts of a plan batch that left the carried state its first attempt had published: the batches behind it stood (finish_batch)
  uint64_t chain_healed = 0;     // ASYNC chains finished early because their last batch was marked for a second attempt (etlg_decode)
  uint64_t ring_recleared = 0;   // result blocks cleared again after a second attempt behind their lap's re-initialisation (finish_batch)
  uint32_t res_seq = 0;
  DevResult* h_init_ring = nullptr;
  DevResult* d_init_ring = nullptr;   // the same, in device memory: the ring is re-initialised with a device-to-device copy (a 40 KB host-to-device
                                      // hipMemcpyAsync made the calling thread wait for everything queued on the stream: 6.7 ms behind 15 cfg5 batches)
  CopyJob copy;        // set while etlg_copy_decode runs etlg_decode over its synthetic frames
  DevBuf d_copy_in, d_copy_offs, d_copy_out, d_copy_out_offs;
  DevBuf d_scan;       // scratch of the record-boundary scan
  uint32_t* h_scan = nullptr;  // pinned: its 4-word result
  size_t scan_half = 0, scan_tiles_cap = 0;  // scan scratch: bytes in front of the hints, tiles it is laid out for
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
  int plan_pre = 1;              // ETLG_PLAN_PRE: 0 = the plan kernel runs its look-back itself; 1 = tile prefixes from the sidecar pre-pass (k_plan_pre), one tile per wave; 2 = ... two tiles per wave
  uint32_t plan_key_max = 0;     // the longest Delete by key a planned table sends in canonical form (no leading zeros / plus signs, 'n' at the other positions of a full-width tuple)
  uint32_t plan_uniform_key_dw = 0, plan_key_below = 0;   // key-row dwords shared by every planned table; the shortest row frame one of them can send (the pre-pass prices a shorter frame as a Delete by key)
  bool plan_deletes = true;      // ETLG_PLAN_DELETES=0: a Delete gives the batch up to the generic kernels, as before round 6's last session
  uint32_t plan_uniform_dw = 0;  // row dwords shared by every planned table (0: they differ — no pre-pass)
  uint32_t plan_margin_pct = 4;  // ETLG_PLAN_MARGIN: LDS window per tile = 64 average frames + this margin (a tile that does not fit is read in place)
  bool ctl_overlap_mode = true;  // ETLG_CTL_OVERLAP
  uint32_t plan_dbg = 0;         // ETLG_PLAN_DBG: bit 0 = no LDS staging (tests of the in-place reader)
  int n_cus = 256;
  uint32_t plan_skip = 0, plan_penalty = 4, plan_streak = 0;
  // k_rows (rows.hip) hands a batch back when a tile does not fit its LDS window / image: the batches behind it skip the kernel for a while
  uint32_t rows_skip = 0, rows_penalty = 4, rows_streak = 0;
  uint64_t rows_win_min = 0;   // bytes a 64-frame tile's window had to hold when k_rows last handed a batch back for that reason
  unsigned long long rows_resized = 0;            // batches k_rows took again with a larger window / image
  int rows_mode = 1;             // ETLG_ROWS: 0 never, 1 wherever k_cells / k_fused-64 would run and the batch is eligible
  unsigned long long rows_n = 0, rows_redone = 0;   // batches k_rows produced / handed back
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
  // A second set of everything a control pre-pass writes (per-frame scratch, the control list, the gathered bytes and their pinned
  // heads): the pre-pass of batch k + 1 goes out BEFORE the host control plane of batch k has read batch k's (etlg_decode), so two
  // are in flight and take turns (etlg_batch::ctl_set).
  struct CtlAlt { DevBuf d_tag, d_emit, d_ffixed, d_fheap, d_blk32, d_blk64, d_ctrl, d_ctrl_stage; CtrlFrame* h_ctl_list = nullptr; uint8_t* h_ctl_stage = nullptr; } ctl_alt;
  bool ctl_hold_mode = true;          // ETLG_CTL_HOLD=0: the held batch is flushed before the next pre-pass goes out (round 3's order)
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
  bool copy_lane_per_byte = false;    // ETLG_COPY_KERNEL=1: the lane-per-byte COPY splitter (copy.hip k_copy_split) instead of the lane-per-row one
  bool ring_h2d = false;              // ETLG_RING_H2D=1 (measurement / bisect knob): re-initialise the result ring from the host template
  uint64_t fixed_hint = 0;            // largest fixed-arena bound seen so far, with head room (setup_outputs)
  hipStream_t d2h_stream = nullptr;   // etlg_batch_download / host-output decodes: the arena of a finished batch travels here
  hipStream_t h2d_stream = nullptr;   // ASYNC batches with host input: their bytes + sidecar are uploaded here, beside the previous batch's decode
  unsigned long long staged_async = 0;
  std::vector<DevBuf*> offs_pool;     // ... into an offsets buffer the batch owns
  std::vector<std::pair<void*, size_t>> blk_dev, blk_host;  // hand-off calls (columns / RowBinary / size hints): pooled device and pinned blocks
  DevBuf d_colsel;                   // etlg_batch_columns: block counts of the row selection
  uint32_t rb_parts_test = 0;        // ETLG_RB_PARTS (tests)
  uint8_t* h_hand = nullptr; size_t h_hand_cap = 0;   // pinned: the row formats' small uploads (initial counters + column words, one copy) and read-backs (row count; totals + counters, one copy each)
  unsigned long long* h_cnt_init = nullptr; size_t h_cnt_init_cols = 0;   // pinned {0, 0, 0, ~0} per column: the hand-off's counters start from it (an asynchronous copy; the content never changes)
  std::vector<etlg_batch*> pending;  // ASYNC batches not finished yet, in issue order
  std::vector<hipEvent_t> ev_pool;   // "result block copied back" events of finished batches
  std::vector<int32_t> last_live;      // slots whose columns d_cols currently holds
  std::vector<DevTable> last_tables;   // what d_tables / d_epochs currently hold
  std::vector<DevEpoch> last_epochs;
  bool side_valid = false;
  bool force_multipass = false;  // ETLG_FORCE_MULTIPASS=1 (tests exercise both paths)
  unsigned long long last_dbg[12] = {0};
  unsigned long long path_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long copy_n[2] = {0, 0};   // table-copy batches produced by k_copy_cells / decoded through the row -> frame rewrite
  bool copy_direct = true;                 // ETLG_COPY_DIRECT=0: always the row -> frame rewrite
  int fused_kernel = -1;         // ETLG_FUSED_KERNEL: 0 k_fused/256, 1 k_fused/64, 2 k_cells, 3 k_plan whenever eligible, 4 k_rows whenever eligible (default: plan, else by frame size)
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
  uint32_t finish_what = 0;     // ETLG_F_FINISH_CELLS: the ETLG_FINISH_* bits finish_batch applies before the batch is handed over
  bool scan_chained = false;    // ... and its decode (the fixed-width plan) was enqueued BEHIND the scan, with the frame count read on the device (DecParams.nframes_dev)
  uint32_t* d_scan_res = nullptr;   // the scan's result words, behind the batch's offsets
  hipStream_t scan_s = nullptr;     // ... the stream that scan runs on (ASYNC: the context's scan stream; otherwise the decode stream itself)
  uint64_t nf_est = 0;              // ... the frame count the stream has shown (params.nframes is the bound the grids are sized by; LDS windows are sized by this)
  int plan_decided = -1;      // decode_tail: -1 not decided yet, 0 / 1 = the first attempt is the generic kernel / the fixed-width plan
  int sidx = 0;               // decode stream the batch's first attempt was enqueued on (0: etlg_ctx::stream, 1: stream2)
  int ctl_set = 0;            // which set of control pre-pass buffers its pre-pass wrote (etlg_ctx::ctl_alt is set 1)
  bool force_rerun = false;   // a batch of the chain before this one had to be decoded again: whatever this one produced started from the wrong state
  hipEvent_t kdone = nullptr; // ASYNC: recorded behind the batch's kernels on its decode stream (the result copy waits for it)
  hipEvent_t done = nullptr;  // recorded behind the copy of the result block: syncing a batch waits for IT, not for the whole stream
  DevResult* h_res = nullptr;  // pinned, from the context's pool
  CopyJob copy;            // table-copy batch: the splitter has to run again before a multi-pass redo
  uint32_t copy_span = 0;  // table-copy batch decoded by k_copy_cells: the bytes of its rows (DevResult.copy_span)
  DevResult* d_res_blk = nullptr;  // this batch's result block on the device
  uint32_t res_seq_no = 0;         // ... and the batch's number in the ring's sequence (finish_batch: has the slot's next re-initialisation been issued?)
  bool used_cells = false; // ... and it was k_cells
  bool used_rows = false;  // ... and it was k_rows
  uint32_t rows_tries = 0; // attempts of k_rows at this batch
  bool no_rows = false;    // k_rows handed this batch back: the second attempt takes k_cells / k_fused
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

