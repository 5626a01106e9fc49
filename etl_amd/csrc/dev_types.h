// Structures shared between the host control plane (host.cpp) and the gfx950
// kernels (kernels.hip). Plain data, no pointers into host memory.
#pragma once
#include <stdint.h>

namespace etlg {

// Frame classes written by k_classify (pgoutput tag, or the XLogData/keepalive
// envelope). 0 = malformed frame (wire-level error).
enum : uint8_t { FT_BAD = 0 };

// One replicated column of a schema slot (mirrors etlg_slot_col).
struct DevCol {
  uint8_t cls;       // etlg_type_class
  uint8_t nullable;
  uint8_t identity;
  uint8_t key_col;   // (k_rows) the column whose key_index is THIS record's index: cell j of a dense key tuple decodes against column key_col of record j
  uint16_t off_full; // byte offset inside a full-layout row
  uint16_t off_key;  // byte offset inside a key-layout row
  uint16_t key_index;
  uint16_t hr;       // (k_rows) heap rank: bits 0..7 among the columns of a full row whose class can reach the heap, bits 8..15 among the identity
                     // columns in key_index order; 0xFF = the class never reaches the heap
};

// A ReplicatedTableSchema instance (reference: crates/etl/src/schema.rs:380-441).
struct DevSlot {
  uint32_t n_cols, n_ident;
  uint32_t row_full, row_key;  // bytes, multiples of 4
  uint32_t st_full, st_key;    // state bytes at the head of a row
  uint32_t cols_base;          // index of the first DevCol
  uint32_t has_var;            // != 0: some column may produce a heap entry (var-len / numeric / deferrable class). For slots of
                               // up to 16 columns, bit k: column k may; bit 16 + k: ... and how many bytes depends on the text
                               // (numeric, float, date / time, bytea), not on its length alone. Wider slots: all ones.
  uint32_t key_masks;          // the same two masks for a dense key tuple: bit j = the identity column with key_index j
  uint32_t ident_mask;         // bit k: column k is an identity column (<= 16 columns)
  uint32_t host_id;            // the schema slot id the arenas name (etlg_slot_desc index). The device table holds only the slots a frame
                               // of the batch can decode against, in ascending id order: DevTable.init_slot / DevEpoch.slot /
                               // DecParams.copy_slot index THIS table, and what reaches ev_slot / Truncate bodies is host_id
};

// Per-table side input for one batch: ownership state
// (apply.rs:2844-2850) + the shared-table-cache timeline
// (table_cache.rs:53-154) as a list of epochs keyed by frame index.
struct DevTable {
  uint32_t table_id;
  uint32_t state_kind;  // etlg_table_state_kind
  uint64_t state_lsn;
  uint32_t init_kind;   // cache entry at batch start: 0 none, 1 WaitingForRelation, 2 Ready
  int32_t init_slot;
  uint32_t ep_begin, ep_end;  // range in the DevEpoch array
};

struct DevEpoch {
  uint32_t frame;   // frame index of the R / M message that caused the change
  uint32_t kind;    // 1 WaitingForRelation, 2 Ready
  int32_t slot;
  uint32_t emit;    // R frames: 1 if the RelationEvent is emitted (owned)
};

// Control frame descriptor handed to the host control plane.
struct CtrlFrame {
  uint32_t frame;
  uint32_t tag;        // 'R' | 'M'
  uint32_t in_txn;
  uint32_t stage_off;  // where k_ctrl_list put a copy of the frame's bytes in DecParams.ctrl_stage; ~0: it did not fit, fetch [o0, o1) of the input
  uint64_t final_lsn;
  uint32_t o0, o1;     // byte range of the frame in the input
};

// Result block (device -> host, one small copy per batch).
struct DevResult {
  unsigned long long first_err;  // min over (frame << 16 | rank << 8 | code); ~0 = none
  uint64_t n_events, fixed_bytes, heap_bytes;
  uint64_t payload[3];
  uint64_t n_frames;             // frames consumed
  uint32_t out_in_txn, n_ctrl;
  uint64_t out_final_lsn, out_next_ord;
  uint32_t fused_fail, ctrl_bytes;  // ctrl_bytes: bytes of R / M frames k_ctrl_list gathered into DecParams.ctrl_stage (4-byte granules). fused_fail: single-pass result not usable: 1 a look-back spin gave up (never expected), 2 the fixed-width plan did not
                                 // cover the batch (plan.hip), 4 a schema too wide for k_cells, 8 the ASYNC predecessor of this batch failed
  unsigned long long dbg_t[12];  // ETLG_FUSED_DBG&8: summed shader-clock cycles per phase (lane 0 of every tile)
  // fused kernel: payload byte counters sharded by tile id so that no single address
  // sees more than ntiles/32 atomics; the host folds them into payload[] after the sync
  unsigned long long pay_shard[32][3];
  // plan.hip: set (release) by the batch's last tile once out_in_txn / out_final_lsn / out_next_ord above are written: a batch that runs
  // beside this one on the second stream (DecParams.flags bit 4) polls it before it reads them
  uint32_t carry_ready;
  uint32_t copy_span;   // k_copy_cells: offs[nrows] - offs[0], the bytes of the rows (TableCopyPayloadMetadata) — travels with the result block instead of two 4-byte copies
};

// Side arguments of the fused single-pass kernel (fused.hip).
struct FusedParams {
  unsigned long long* d_txn;   // [ntiles]  st:2 | seg:30 | mark:32
  unsigned long long* d_outa;  // [ntiles]  st:2 | events:30 | heap dwords:32
  unsigned long long* d_outb;  // [ntiles]  st:2 | fixed dwords:62
  uint32_t* ticket;
  uint32_t ntiles;
  uint32_t lds_bytes;          // staging capacity per tile (dynamic LDS)
  uint32_t in_aligned;         // input pointer is 16-byte aligned
  uint32_t blk;                // frames per tile (256 or 64)
  uint32_t dbg;                // ablation bits (profiling only): 1 no LDS staging, 2 skip writes, 4 skip look-back, 8 phase timers
  uint32_t seq_lookback;       // 1: ownership needs the transaction's final_lsn (a table is in SyncDone state)
  uint32_t side_bytes;         // LDS bytes reserved for a copy of the side-input tables (0 = read them from global)
  uint32_t maxc;               // k_cells: widest schema slot of the batch (columns)
  uint32_t clear_words;        // 64-bit words of d_clear this launch zeroes (the descriptor buffer of the NEXT batch)
  unsigned long long* d_clear;
  uint32_t copy_rel;           // k_copy_cells: the table id the rows' Insert events carry
  uint32_t rows_maxh_old;      // k_rows: most heap-class IDENTITY columns in one slot (rows of the heap-cell table for key / old images)
  uint32_t rows_maxh;          // k_rows: most heap-class columns in one slot (rows for new images)
};

constexpr unsigned long long kNoErr = ~0ull;
constexpr int kBlock = 256;  // frames per workgroup (one lane per frame)

// error stage ranks: for one frame the lowest rank wins, mirroring the order
// in which the reference detects them (wire parse -> transaction state ->
// ownership/schema lookup -> event decode).
enum : uint32_t { RK_WIRE = 0, RK_TXN = 1, RK_SCHEMA = 2, RK_DECODE = 3, RK_COPY_SHAPE = 4 };

struct DecParams {
  const uint8_t* in;
  const uint32_t* offs;  // nframes + 1
  uint32_t nframes;
  uint32_t nblocks;
  uint64_t in_len;
  // carried transaction state (apply.rs:942-963)
  uint32_t in_txn;
  uint32_t worker_kind, sync_table;
  // bits 8-11 (0x100 no row stores, 0x200 no header stores, 0x400 no row decode, 0x800 uniform waves only) are
  // profiling ablations set from ETLG_FUSED_DBG; results are wrong with any of them
  uint32_t flags;        // bit0: NO_CONTROL asserted; bit1: table-copy rows (synthetic Insert frames: no ownership
                         // check, NULL allowed in every column — table_row.rs:199-201); bit4: `carry` belongs to a batch that may still
                         // be running on the other stream: poll its carry_ready before reading it (plan.hip only)
  uint64_t final_lsn, next_ord;
  uint32_t host_err_frame;  // frames >= this are ignored (host control plane failed there)
  uint32_t n_tables;
  uint32_t n_epochs, n_slots, n_cols;  // sizes of the side-input arrays
  int32_t copy_slot;     // flags bit1 (table-copy rows): the schema slot every row decodes against
  // side inputs
  const DevTable* tables;
  const DevEpoch* epochs;
  const DevSlot* slots;
  const DevCol* cols;
  // per-frame scratch
  uint8_t* f_tag;
  uint8_t* f_emit;
  uint32_t* f_fixed;  // bytes
  uint32_t* f_heap;   // bytes
  // per-block aggregates / prefixes
  uint32_t* blk_cnt;      // ordinal consumers per block -> exclusive prefix
  uint32_t* blk_last;     // max over ((idx+1)<<1|isBegin) -> exclusive running max
  uint32_t* blk_ev;       // events per block -> exclusive prefix
  uint64_t* blk_fixed;    // bytes
  uint64_t* blk_heap;
  uint64_t* blk_payload;  // 3 per block
  CtrlFrame* ctrl;        // compacted control frames
  uint32_t ctrl_cap;
  uint32_t ctrl_stage_cap;  // bytes of ctrl_stage
  uint8_t* ctrl_stage;    // the control frames' bytes, gathered so that the host fetches them with one copy
  // outputs
  uint8_t* ev_kind; uint8_t* ev_flags;
  uint32_t* ev_table; uint32_t* ev_slot;
  uint64_t* ev_start; uint64_t* ev_commit; uint64_t* ev_ord; uint64_t* ev_body;
  uint8_t* fixed; uint8_t* heap;
  uint64_t fixed_cap, heap_cap;
  DevResult* res;
  // ETLG_F_ASYNC chains batches on the device: when set, the carried transaction state (in_txn / final_lsn / next_ord above)
  // is read from the result block of the batch issued just before this one on the same stream, not from these host values
  const DevResult* carry;
  // ETLG_F_ASYNC without a sidecar, fixed-width plan (round 6): the record-boundary scan of THIS batch is still in flight when the decode is
  // enqueued behind it. `nframes` above is then an upper bound the grids were sized by; the kernels read the count the scan left on the
  // device — words [0] frames, [1] flags, [2] tiles that guessed wrong (either non-zero: the scan did not hold) — and take it for nframes (plan.hip, plan_frames_from_device)
  const uint32_t* nframes_dev;
};

// The fixed-width decode plan (plan.hip): the eligible tables of a batch, sorted by rel_id, and their columns.
struct PlanTab {
  uint32_t rel_id;
  uint32_t slot;        // schema slot id
  uint32_t n_cols;
  uint32_t row_dwords;  // full-layout row block, dwords
  uint32_t cols_base;   // index of the table's first column word
  uint32_t key_dwords;  // key-layout row block, dwords; 0: the plan does not take this table's Deletes (no identity column)
  uint32_t n_ident;
  uint32_t keys_base;   // index of the table's first key word (n_cols of them, behind the column words)
};
// one word per column: cls | nullable << 8 | off_full << 16
// one key word per column (Deletes by key, round 6): identity | key_index << 8 | off_key (dwords) << 24

struct PlanParams {
  unsigned long long* desc;    // look-back words (plan.hip): desc[ntiles] | gdesc[ceil(ntiles / 64)] | dlsn[ntiles]
  unsigned long long* d_clear; // descriptor buffer of the NEXT batch, zeroed by this launch
  uint32_t clear_words;
  uint32_t ntiles;             // tiles of 64 frames (one wave each)
  uint32_t lds_bytes;          // dynamic LDS per tile: the staging window [0, rows_off) + the tile's image of the fixed arena (64 rows)
  uint32_t rows_off;
  uint32_t n_tabs;
  const PlanTab* tabs;
  const uint32_t* cols;
  uint32_t dbg;                // ETLG_PLAN_DBG. bit 0: no LDS staging (tests: the in-place reader). Profiling ablations, results are WRONG:
                               // bit 1 stop after staging, bit 2 stop after the message heads, bit 3 no cell decode / row stores, bit 4 no event header stores
                               // bit 5: phase clocks into DevResult.dbg_t; bit 9: one tile per wave (k_plan) even when two would do
  uint32_t max_row_dw;         // dwords of the widest planned row (k_plan2 / k_plan3 keep a row of up to 6 / 8 dwords in registers)
  // Tile prefixes from the sidecar pre-pass (k_plan_pre, plan.hip), or null: the decode kernel then runs the look-back itself.
  // pre[2 * tile] = {last Begin / Commit mark before the tile : 30 | fixed-arena dwords before it : 32}, pre[2 * tile + 1] = LSN of the
  // Begin that is open when the tile starts — what plan_resolve2 would hand the tile.
  const unsigned long long* pre;
  unsigned long long* pre_out; // ... as the pre-pass writes it
  uint32_t* pre_ticket;        // the pre-pass's arrival counter (zero between launches)
  uint32_t pre_tag;            // 1..3, changes with every use of a buffer: the status of this launch's group words
  uint32_t pre_row_dw;         // the row dwords every planned table shares (the pre-pass prices a frame without reading it)
  uint32_t pre_key_dw;         // ... and the key-row dwords they share, with
  uint32_t pre_key_below;      // the length below which a frame that is no Begin / Commit is priced as a Delete by key: the shortest row frame any
                               // planned table can send (38 + 6 per NOT NULL column + 1 per nullable one). 0: every such frame is a row (Deletes give the batch up)
  uint32_t pre_key_max;        // the longest Delete by key the pre-pass still recognises: a frame of pre_key_below .. pre_key_max bytes has its tag byte read
                               // (like the frames of a Begin's / Commit's length) and is priced as a key row when that is 'D'
};

// ---- columnar hand-off (columns.hip)
struct ColSel {            // which events are rows of the hand-off
  const uint8_t* ev_kind; const uint8_t* ev_flags; const uint32_t* ev_slot; const uint64_t* ev_body;
  uint64_t n_events;
  uint32_t slot, kinds;    // kinds: bit 0 inserts, bit 1 new rows of non-partial updates, bit 2 full old rows of deletes, bit 3 key rows of deletes
  unsigned long long* host_rows;  // optional: I/U/D events of the slot the selection leaves out (partial updates, key-only deletes)
  uint32_t row_full, row_key;
  uint32_t* blk;           // per-block counts -> exclusive prefix; blk[nblocks] = total
  uint32_t nblocks;
  uint64_t* row_event; uint64_t* row_base;   // outputs of k_col_rows
  // BigQuery rows (pb != 0; bigquery/core.rs:978-1036): an event becomes 0, 1 or 2 rows, and a row's base carries what it is in its
  // top bits (kPbDelete: the sparse DELETE row of an old image; kPbKey: that image has the key layout; kPbSecond: sequence ordinal 1)
  uint32_t pb, n_cols;
  uint32_t identity_pk;    // the slot's replica identity is the primary key
  uint32_t pk_comparable;  // every primary-key column is of a class whose Cell equality is equality of the arena's words / bytes
  const uint8_t* fixed; const uint8_t* heap;
  const uint32_t* cols;    // per replicated column: cls | .. | off_full << 16 (RbJob.cols)
  const uint32_t* kcols;   // identity | source nullable << 1 | primary key << 2 | key_index << 8 | off_key << 16 (RbJob.kcols)
};
constexpr unsigned long long kPbDelete = 1ull << 63, kPbKey = 1ull << 62, kPbSecond = 1ull << 61, kPbBase = (1ull << 61) - 1;

struct ColJob {
  const uint8_t* fixed; const uint8_t* heap; const uint64_t* row_base;
  uint64_t n_rows;
  uint32_t col_index, off_full, cls, kind;
  unsigned long long* validity; unsigned long long* deferred; uint8_t* values;
  unsigned long long* null_count; unsigned long long* deferred_count;
  uint32_t* lens; const int64_t* offsets;   // var-len
  // list columns (array literals parsed on the device): element class, child validity words, child null counter, first error
  uint32_t elem_cls, _pad;
  uint32_t* child_validity; unsigned long long* child_nulls; unsigned long long* err;
  uint32_t* child_lens; const int64_t* child_offsets;   // lists of strings: byte length / start of every element
};

struct HintJob {           // Event::size_hint per event (k_size_hints)
  const uint8_t* ev_kind; const uint8_t* ev_flags; const uint32_t* ev_table; const uint32_t* ev_slot; const uint64_t* ev_body;
  const uint8_t* fixed; const uint8_t* heap;
  uint64_t n_events;
  const uint32_t* slots;   // per slot: n_cols, n_ident, row_full, row_key, cols_base
  const uint32_t* cols;    // per column: cls | identity << 8 | off_full << 16, then off_key
  uint32_t n_slots;
  uint32_t m_begin, m_commit, m_insert, m_update, m_delete, m_truncate, m_relation, m_rts, m_row, m_cell;
  unsigned long long* out;
};

struct FinJob {            // the finish pass (k_fin_count / k_fin_fill, columns.hip)
  const uint8_t* ev_kind; const uint8_t* ev_flags; const uint32_t* ev_slot; const uint64_t* ev_body;
  uint8_t* fixed; uint8_t* heap;
  uint64_t n_events;
  const uint32_t* slots;   // per host slot: n_cols, n_ident, row_full, row_key, cols_base, fin_base, n_fin
  const uint32_t* cols;    // per column: cls | identity << 8 | elem_cls << 16, off_full | off_key << 16, key_index
  const uint32_t* fin;     // the finishable columns of every slot (indexes into the slot's columns)
  uint32_t n_slots, maxfin, what;
  uint32_t* lens; const int64_t* offsets;
  uint32_t* counts;        // elements of every array the count pass sized (the fill pass lays the entry out without another walk)
  uint64_t heap_base;      // where the appended entries start
  unsigned long long* stats;   // [0] deferred cells examined, [1] arrays typed, [2] floats settled, [3] left deferred
};

struct RbJob {             // ClickHouse RowBinary rows (k_rb_rows)
  const uint8_t* fixed; const uint8_t* heap; const uint64_t* row_event; const uint64_t* row_base;
  const uint8_t* ev_kind; const uint64_t* ev_commit; const uint64_t* ev_ord;
  uint64_t n_rows;
  uint32_t n_cols, engine;         // engine: 0 MergeTree, 1 ReplacingMergeTree
  uint32_t cdc_nullable;           // bit 0 / 1: the first / second trailing CDC column is Nullable() in the destination
  uint32_t format;                 // 0 ClickHouse RowBinary, 1 BigQuery protobuf (Insert rows, prost wire format)
  const uint32_t* cols;            // per replicated column: cls | nullable << 8 | off_full << 16
  const uint32_t* kcols;           // per replicated column, for key images: identity | source nullable << 1 | primary key << 2 | key_index << 8 | off_key << 16
  const uint8_t* ev_flags;
  uint32_t* lens; const int64_t* offsets; uint8_t* out;
  unsigned long long* err;         // min over failing cells of (row << 24 | column << 8 | code); ~0 = none
  uint32_t has_json;               // the table has a json / jsonb column: the kernels that carry json_display
  uint32_t qparts, parts;          // the counting pass notes where the row's qparts pieces begin (4; 2 / 1 for narrow tables); the byte pass writes a row with parts lanes (1, 2, 4 <= qparts)
  uint32_t* part_off;              // [(qparts - 1) x n_rows]: where pieces 1 .. of a row begin, in bytes from the row's start
};


}  // namespace etlg
