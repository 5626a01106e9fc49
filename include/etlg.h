/*
 * etlg.h — C ABI of libetl_gfx950.so, the MI355X-native batched pgoutput decode
 * + CDC event-transform stage that drops in upstream of supabase/etl's
 * Destination::write_events.
 *
 * The reference has no FFI for this path (it is 100 % Rust, monomorphised
 * generics).  Each entry point below names the reference code it replaces
 * (paths relative to the supabase/etl checkout):
 *
 *   etlg_ctx_create / _destroy   one per apply-loop stream; owns what
 *                                ApplyLoop keeps in `self.state` + the
 *                                SharedTableCache
 *                                (crates/etl/src/replication/apply.rs:942-963,
 *                                 crates/etl/src/replication/table_cache.rs:88-154)
 *   etlg_ctx_set_worker          WorkerContext::{Apply,TableSync} ownership rule
 *                                (apply.rs:2626-2639, 2836-2867, 3514-3519) and the
 *                                bootstrap snapshot id (apply.rs:2410-2413)
 *   etlg_schema_put              SchemaStore::store_table_schema /
 *                                get_table_schema at-or-before lookup
 *                                (crates/etl/src/store/schema/base.rs:19-69,
 *                                 crates/etl/src/store/schema/table.rs:61-71)
 *   etlg_table_state             StateStore::get_table_state for the
 *                                should_apply_changes filter (apply.rs:2836-2867)
 *   etlg_table_ready             SharedTableCache::note_ready as done by the
 *                                table-copy path before streaming starts
 *                                (table_cache.rs:122)
 *   etlg_decode                  the body of the hot loop for N messages:
 *                                ReplicationMessage::parse +
 *                                LogicalReplicationMessage::parse
 *                                (postgres-replication 0.6.7, call sites
 *                                 apply.rs:2037-2125) → handle_*_message
 *                                (apply.rs:2279-2617) → codec::parse_event_from_*
 *                                (crates/etl/src/postgres/codec/event.rs:303-547)
 *                                → parse_cell_from_postgres_text
 *                                (crates/etl/src/postgres/codec/text.rs:32-153)
 *   etlg_last_error              EtlError {kind, description, detail}
 *                                (crates/etl/src/error.rs:27-76, 85-170)
 *   etlg_batch_*                 the Vec<Event> pushed into EventBatch
 *                                (apply.rs:1918-1928), as a columnar arena
 *
 * Conventions mirrored from the Rust side: callee allocates results, caller
 * frees through the exported free; a context is NOT re-entrant (the apply
 * loop is `&mut self`); decode is fail-fast: the first bad frame stops the
 * batch, all events before it stay valid, and the error carries the
 * reference's ErrorKind + static description string.
 *
 * No torch types, no C++ types: plain pointers and sizes only.
 */
#ifndef ETLG_H
#define ETLG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ETLG_ABI_VERSION 1u

/* ------------------------------------------------------------------ errors */

/* Subset of crates/etl/src/error.rs:85-170 `ErrorKind` reachable from the
 * decode path. Values are ours; names are the reference's. */
typedef enum etlg_error_kind {
  ETLG_OK = 0,
  ETLG_ConversionError = 1,        /* error.rs ConversionError */
  ETLG_InvalidData = 2,            /* error.rs InvalidData */
  ETLG_ValidationError = 3,        /* error.rs ValidationError */
  ETLG_InvalidState = 4,           /* error.rs InvalidState */
  ETLG_MissingTableSchema = 5,     /* error.rs MissingTableSchema */
  ETLG_CorruptedTableSchema = 6,   /* error.rs CorruptedTableSchema */
  ETLG_DeserializationError = 7,   /* error.rs DeserializationError */
  ETLG_SourceConnectionFailed = 8, /* wire-level parse failure: tokio_postgres::Error
                                      without SQLSTATE (error.rs:947) */
  ETLG_IoError = 9,                /* io::Error from cstr accessors (error.rs:564) */
  ETLG_UnsupportedValueInDestination = 10, /* error.rs:136; only from etlg_batch_protobuf (bigquery/validation.rs) */
  ETLG_NullValuesNotSupportedInArrayInDestination = 11, /* error.rs:134; only from etlg_batch_protobuf (reject_nulls, bigquery/validation.rs:127-141) */
  /* library-level (no reference analog) */
  ETLG_InvalidArgument = 100,
  ETLG_DeviceError = 101,
  ETLG_Unsupported = 102
} etlg_error_kind;

typedef struct etlg_error {
  int32_t kind;            /* etlg_error_kind */
  int32_t code;            /* etlg_err_code: identifies the static description */
  const char* description; /* the reference's static description string */
  const char* detail;      /* optional dynamic detail (may be NULL) */
  int64_t frame_index;     /* index of the offending CopyData frame, -1 if n/a */
} etlg_error;

/* One code per distinct (kind, static description) pair produced on this
 * path; the table lives in etlg_err_table() so oracle, kernels and tests
 * share numbering but not code. */
typedef enum etlg_err_code {
  ETLG_E_NONE = 0,
  ETLG_E_WIRE = 1,                /* SourceConnectionFailed / "PostgreSQL connection failed" */
  ETLG_E_TXN_STATE = 2,           /* InvalidState / "Invalid transaction state" */
  ETLG_E_COMMIT_LSN = 3,          /* ValidationError / "Invalid commit LSN" */
  ETLG_E_MISSING_SHARED_STATE = 4,/* InvalidState / "Missing shared table state" */
  ETLG_E_WAITING_RELATION = 5,    /* InvalidState / "Waiting for relation state cannot decode row event" */
  ETLG_E_TUPLE_WIDTH = 6,         /* ConversionError / "Tuple data field count does not match schema" */
  ETLG_E_FULL_ROW_MISSING = 7,    /* ConversionError / "Tuple missing source value for full row image" */
  ETLG_E_REQUIRED_NULL = 8,       /* InvalidData / "Required column missing from tuple" */
  ETLG_E_BINARY_FORMAT = 9,       /* ConversionError / "Binary format not supported in tuple data" */
  ETLG_E_UTF8 = 10,               /* ConversionError / "UTF-8 conversion failed" */
  ETLG_E_OLD_ROW_WIDTH = 11,      /* ConversionError / "Old tuple row width does not match schema" */
  ETLG_E_KEY_SHAPE = 12,          /* ConversionError / "Replica-identity tuple shape does not match schema" */
  ETLG_E_KEY_MISSING_COLS = 13,   /* ConversionError / "Replica-identity tuple missing key columns" */
  ETLG_E_KEY_MISSING_VALUE = 14,  /* ConversionError / "Replica-identity tuple missing source value" */
  ETLG_E_BOOL = 15,               /* InvalidData / "Invalid boolean value" */
  ETLG_E_INT = 16,                /* ConversionError / "Integer parsing failed" */
  ETLG_E_FLOAT = 17,              /* ConversionError / "Float parsing failed" */
  ETLG_E_NUMERIC = 18,            /* ConversionError / "Numeric parsing failed" */
  ETLG_E_BYTEA = 19,              /* ConversionError / "Bytea hex string conversion failed" */
  ETLG_E_DATETIME = 20,           /* ConversionError / "Datetime parsing failed" */
  ETLG_E_UUID = 21,               /* InvalidData / "UUID parsing failed" */
  ETLG_E_JSON = 22,               /* DeserializationError / "JSON deserialization failed" */
  ETLG_E_ARRAY_SHORT = 23,        /* ConversionError / "Array input too short" */
  ETLG_E_ARRAY_BRACES = 24,       /* ConversionError / "Array input missing braces" */
  ETLG_E_ARRAY_DIMS = 25,         /* ConversionError / "Array input has a malformed dimensions prefix" */
  ETLG_E_ARRAY_MULTIDIM = 26,     /* ConversionError / "Multidimensional array input is not supported" */
  ETLG_E_ARRAY_QUOTE = 27,        /* ConversionError / "Array input contains an unterminated quote" */
  ETLG_E_ARRAY_ESCAPE = 28,       /* ConversionError / "Array input contains an unterminated escape" */
  ETLG_E_SCHEMA_NOT_FOUND = 29,   /* MissingTableSchema / "Table schema not found" */
  ETLG_E_UNKNOWN_COLUMNS = 30,    /* CorruptedTableSchema / "Replication stream contains columns missing from the stored table schema" */
  ETLG_E_DDL_PARSE = 31,          /* ConversionError / "Failed to parse schema change message" */
  ETLG_E_IO = 32,                 /* IoError / "I/O operation failed" */
  ETLG_E_BOOTSTRAP_SNAPSHOT = 33, /* InvalidState / "Bootstrap table schema snapshot exceeded requested snapshot" */
  ETLG_E_SNAPSHOT_MISMATCH = 34,  /* InvalidState / "Table schema snapshot mismatch" */
  ETLG_E_CTRL_HINT = 35,          /* InvalidArgument / "Control frame found in a batch declared control-free" */
  /* table-copy rows (crates/etl/src/postgres/codec/table_row.rs:57-254) */
  ETLG_E_COPY_UNTERMINATED = 36,  /* ConversionError / "Row data not properly terminated" */
  ETLG_E_COPY_MORE_COLS = 37,     /* ConversionError / "Postgres COPY row contains more columns than the table schema" */
  ETLG_E_COPY_FEWER_COLS = 38,    /* ConversionError / "Postgres COPY row contains fewer columns than the table schema" */
  ETLG_E__COUNT
} etlg_err_code;

typedef struct etlg_err_desc {
  int32_t kind;
  const char* description;
} etlg_err_desc;

/* Static (kind, description) for a code; NULL if out of range. */
const etlg_err_desc* etlg_err_table(int32_t code);

/* ----------------------------------------------------------------- schemas */

/* Value class a column decodes to; determined by the column's *stored* type
 * OID exactly as parse_cell_from_postgres_text switches on it
 * (crates/etl/src/postgres/codec/text.rs:32-153; unknown OIDs are TEXT,
 * crates/etl-postgres/src/type_utils.rs:9-11). */
typedef enum etlg_type_class {
  ETLG_TC_STRING = 0, /* everything without a dedicated arm -> Cell::String */
  ETLG_TC_BOOL = 1,
  ETLG_TC_I16 = 2,
  ETLG_TC_I32 = 3,
  ETLG_TC_I64 = 4,
  ETLG_TC_U32 = 5, /* oid */
  ETLG_TC_F32 = 6,
  ETLG_TC_F64 = 7,
  ETLG_TC_NUMERIC = 8,
  ETLG_TC_BYTEA = 9,
  ETLG_TC_DATE = 10,
  ETLG_TC_TIME = 11,
  ETLG_TC_TIMETZ = 12,
  ETLG_TC_TIMESTAMP = 13,
  ETLG_TC_TIMESTAMPTZ = 14,
  ETLG_TC_UUID = 15,
  ETLG_TC_JSON = 16,
  ETLG_TC_ARRAY = 17, /* any `_xxx` array type; element class via etlg_array_elem_class */
  ETLG_TC__COUNT
} etlg_type_class;

/* OID -> class; never fails (unknown -> STRING). */
int32_t etlg_type_class_of_oid(uint32_t type_oid);
/* For an array OID: element class (STRING for arrays without a dedicated arm). */
int32_t etlg_array_elem_class(uint32_t array_type_oid);
/* Bytes a column of this class occupies in a row's fixed block (multiple of 4). */
uint32_t etlg_slot_bytes(int32_t type_class);

/* A stored column, in attnum order — what ColumnSchema carries
 * (crates/etl-postgres/src/schema.rs:213). */
typedef struct etlg_col {
  const char* name; /* NUL-terminated UTF-8; masks are built by name (crates/etl/src/schema.rs:30-61) */
  uint32_t type_oid;
  int32_t type_modifier;
  int32_t attnum;       /* ordinal_position */
  uint8_t nullable;     /* !attnotnull */
  uint8_t primary_key;  /* 1 if part of the primary key */
  uint8_t _pad[2];
} etlg_col;

/* Table replication state as seen by should_apply_changes
 * (apply.rs:2844-2850): Ready, SyncDone{lsn}, or anything else. */
typedef enum etlg_table_state_kind {
  ETLG_TS_ABSENT = 0,   /* no state stored -> never owned */
  ETLG_TS_READY = 1,
  ETLG_TS_SYNC_DONE = 2,/* owned iff lsn <= remote_final_lsn */
  ETLG_TS_OTHER = 3     /* Init/DataSync/FinishedCopy/SyncWait/Catchup/Errored */
} etlg_table_state_kind;

typedef enum etlg_worker_kind {
  ETLG_WORKER_APPLY = 0,
  ETLG_WORKER_TABLE_SYNC = 1
} etlg_worker_kind;

/* --------------------------------------------------------------- lifecycle */

typedef struct etlg_ctx etlg_ctx;
typedef struct etlg_batch etlg_batch;

uint32_t etlg_abi_version(void);

/* hip_device >= 0 selects the GPU. There is no CPU backend: creation fails
 * with ETLG_DeviceError when no gfx950 device is available. */
int32_t etlg_ctx_create(int32_t hip_device, etlg_ctx** out);
/* Human-readable reason of the last failed etlg_ctx_create on this thread's process. */
const char* etlg_create_error(void);
void etlg_ctx_destroy(etlg_ctx* ctx);

/* Run all device work of this context on an existing hipStream_t
 * (e.g. torch.cuda.current_stream().cuda_stream). NULL = own stream. */
int32_t etlg_ctx_set_stream(etlg_ctx* ctx, void* hip_stream);

int32_t etlg_ctx_set_worker(etlg_ctx* ctx, int32_t worker_kind,
                            uint32_t table_sync_table_id,
                            uint64_t bootstrap_snapshot_lsn);

int32_t etlg_schema_put(etlg_ctx* ctx, uint32_t table_id, uint64_t snapshot_lsn,
                        const char* schema_name, const char* table_name,
                        uint32_t ncols, const etlg_col* cols);

int32_t etlg_table_state(etlg_ctx* ctx, uint32_t table_id, int32_t state_kind,
                         uint64_t lsn);

/* Pre-populate the shared table cache with a Ready replicated schema built
 * from the stored schema at `snapshot_lsn` (at-or-before) and byte masks over
 * its columns (1 = replicated / identity). Returns the schema slot id (>= 0)
 * or a negative etlg_error_kind. */
int32_t etlg_table_ready(etlg_ctx* ctx, uint32_t table_id, uint64_t snapshot_lsn,
                         const uint8_t* replication_mask,
                         const uint8_t* identity_mask, uint32_t nmask);

/* Drop the shared-table-cache entry of a table (idempotent): SharedTableCache::remove_table
 * (crates/etl/src/replication/table_cache.rs:131-145), as the table-sync worker does before it restarts a table
 * (crates/etl/src/replication/table_sync/mod.rs:233). Rows of that table then fail with "Missing shared table state"
 * until a Relation message or etlg_table_ready installs a schema again. Stored schemas and slot ids are untouched. */
int32_t etlg_table_forget(etlg_ctx* ctx, uint32_t table_id);

/* The shared-table-cache entry of a table (SharedTableCache::get, table_cache.rs:99-102): returns 0 if there is none, else 1
 * with *kind = 1 WaitingForRelation | 2 Ready, *snapshot_lsn = the entry's snapshot id, *schema_slot = the Ready schema (-1 while waiting). */
int32_t etlg_table_cache_get(const etlg_ctx* ctx, uint32_t table_id, int32_t* kind, uint64_t* snapshot_lsn, int32_t* schema_slot);

/* Reset the transaction state (remote_final_lsn = None, next ordinal = 0),
 * as a fresh ApplyLoop does. */
int32_t etlg_ctx_reset_stream_state(etlg_ctx* ctx);

/* ------------------------------------------------------------------ decode */

enum {
  ETLG_F_INPUT_ON_DEVICE = 1u << 0,  /* buf / frame_offsets are device pointers */
  ETLG_F_OUTPUT_ON_DEVICE = 1u << 1, /* do not copy the arenas to the host; view holds device pointers */
  ETLG_F_NO_CONTROL = 1u << 2,       /* caller asserts: no R/M/T frame in this batch (skips the
                                        control-plane round trip; verified on device) */
  ETLG_F_ASYNC = 1u << 3,            /* with OUTPUT_ON_DEVICE | NO_CONTROL: enqueue only, do not synchronize;
                                        counts become valid after etlg_batch_sync. The input must be COMPLETE in
                                        device memory when the call is made: the batch may be decoded on a private
                                        stream beside its predecessor, not behind work enqueued on the context's
                                        stream (etlg_ctx_fence orders the other direction). Sync batches in issue
                                        order and keep fewer than 32 of them in flight per context (their
                                        result blocks live in a ring of 32; a batch that reports an error is
                                        decoded again, on the exact-error path, when it is synced).
                                        With frame_offsets = NULL the record-boundary scan of the batch runs on a
                                        private stream beside the previous batch's decode and the call returns with
                                        that scan in flight (the next call on the context, or the batch's sync,
                                        enqueues the decode): the input must be COMPLETE in device memory when the
                                        call is made (not merely enqueued on the context's stream).
                                        With HOST input (no INPUT_ON_DEVICE) and a sidecar: the bytes and the sidecar are
                                        uploaded into a device block the batch owns, on a private copy stream beside the
                                        decode of the batch before it, and the batch joins the chain like a device-input
                                        one. Truly asynchronous only from pinned memory (etlg_host_alloc); the caller must
                                        not touch buf / frame_offsets until the batch is synced (two staging buffers in
                                        rotation: fill one while the other is in flight). Without a sidecar a host-input
                                        batch is decoded synchronously, as before */
  ETLG_F_FINISH_CELLS = 1u << 4      /* round 6: run etlg_batch_finish_cells(ETLG_FINISH_ARRAYS | ETLG_FINISH_FLOATS) on the batch
                                        before it is handed over (an ASYNC batch: when it is synced; a host-output batch: before
                                        the download): array cells come back TYPED (etlg_array_hdr entries) and no float cell
                                        comes back DEFERRED. Also honoured by etlg_copy_decode */
};

/* Pinned (page-locked) host memory for the staging buffers of a host that feeds etlg_decode with ETLG_F_ASYNC: what the
 * reference's apply loop accumulates per batch (EventBatch, crates/etl/src/replication/apply.rs:1918-1928) becomes two or more
 * 64 MiB buffers of raw CopyData frames + their u32 offsets (crates/etl-gfx950/src/batcher.rs). Freed with etlg_host_free. */
int32_t etlg_host_alloc(etlg_ctx* ctx, size_t bytes, void** out);
void etlg_host_free(void* p);

/* buf = `nframes` concatenated CopyData frames exactly as on the socket:
 *   'd' | Int32-BE length (incl. itself) | payload
 * payload = XLogData 'w' | u64 wal_start | u64 wal_end | i64 ts | pgoutput msg
 *         | keepalive 'k' | u64 wal_end | i64 ts | u8 reply
 * frame_offsets: optional sidecar of nframes+1 byte offsets (the host learns
 * each frame length on receipt); NULL = the device scans record boundaries
 * itself and nframes is ignored. (The sidecar is also what lets fixed-width
 * batches skip every inter-tile dependency: frames are priced by their length
 * before they are read — DESIGN.md 3.0, k_plan_pre. Results do not depend on it.)
 * Returns 0, or the etlg_error_kind of the first failing frame (the batch is
 * still returned and holds every event before it). */
int32_t etlg_decode(etlg_ctx* ctx, const uint8_t* buf, size_t len,
                    const uint32_t* frame_offsets, size_t nframes,
                    uint32_t flags, etlg_batch** out);

const etlg_error* etlg_last_error(const etlg_ctx* ctx);

/* Table-copy rows. Replaces parse_table_row_from_postgres_copy_bytes
 * (crates/etl/src/postgres/codec/table_row.rs:47-254) applied to every item of the
 * TableCopyStream (crates/etl/src/postgres/stream/table_copy.rs:54-79).
 * buf = nrows COPY ... TO STDOUT (text format) row payloads exactly as CopyOutStream
 * yields them (one row per CopyData message, WITHOUT the 'd' framing), concatenated;
 * row_offsets = nrows + 1 byte offsets (required; host or device as buf).
 * schema_slot = the ReplicatedTableSchema to decode against: the value etlg_table_ready
 * returned (its replicated columns are the reference's `column_schemas`).
 * The rows come back in the same arena as a batch of Insert events: kind 'I',
 * table_id / schema_slot of the slot, start_lsn = commit_lsn = 0, tx_ordinal = row index,
 * one full-layout row per event; payload_bytes[0] = bytes of the rows decoded
 * (TableCopyPayloadMetadata). NULL fields ("\N") are NULL cells whatever the column's
 * nullability, as in the reference. Fail-fast: the first bad row ends the batch with the
 * reference's error; rows before it are valid. The stream state of the context
 * (transaction carry) is not touched.
 * flags: ETLG_F_INPUT_ON_DEVICE, ETLG_F_OUTPUT_ON_DEVICE, ETLG_F_ASYNC.
 * ETLG_F_ASYNC (with ETLG_F_OUTPUT_ON_DEVICE): enqueue only — the reference's caller streams rows continuously
 * (crates/etl/src/postgres/stream/table_copy.rs:78-99; the table-sync worker batches them, crates/etl/src/replication/table_sync/copy.rs) and
 * the next batch can be handed over while this one decodes. The call returns with the batch in flight; etlg_batch_sync finishes it
 * (counts, error, payload_bytes become valid then). Batches are synced in issue order, fewer than 32 in flight per context, as for
 * etlg_decode. Table-copy batches are independent of each other (a virtual transaction each): a batch that ends in an error does not
 * touch the batches behind it. Host rows are uploaded into a device block the batch owns, on a copy stream beside the decode of the
 * batch before it — truly asynchronous only from pinned memory (etlg_host_alloc); the caller keeps buf / row_offsets untouched (and
 * device-resident input alive) until the batch is synced. What the rows -> arena kernel leaves to the frame rewrite (a malformed row,
 * rows wider than a tile's window) is redone when the batch is synced. Mixing kinds on one context is allowed and serialises: a WAL
 * batch (etlg_decode) finishes the table-copy batches in flight first, and the other way round. */
int32_t etlg_copy_decode(etlg_ctx* ctx, int32_t schema_slot, const uint8_t* buf, size_t len,
                         const uint32_t* row_offsets, size_t nrows, uint32_t flags, etlg_batch** out);

/* Record-boundary scan on its own: the frame_offsets sidecar of `buf` (the same scan
 * etlg_decode runs when it is given none), e.g. to cut a staged stream into shards.
 * Follows 'd' | Int32-BE length from offset 0; a malformed header turns the rest of
 * the buffer into one last frame (which then fails in etlg_decode as a wire error).
 * Replaces the per-message framing the reference gets from its socket codec
 * (postgres/stream/replication_message.rs:89-230: one CopyData payload per stream item).
 * flags: ETLG_F_INPUT_ON_DEVICE (buf is a device pointer), ETLG_F_OUTPUT_ON_DEVICE
 * (offsets_out is a device pointer). offsets_out receives *nframes_out + 1 entries;
 * cap = entries available.
 * Device memory the scan keeps on the context (grow-only, released with the context): a scratch row of 360 16-bit
 * offsets per 8 KiB tile of the largest input scanned so far plus summaries — about 9 % of that input (5.9 MB for
 * 64 MiB, ~470 MB for an input near 4 GiB) — and, per batch in flight without a sidecar, an offsets buffer of
 * len / 24 + 1 024 entries. A tile with more than 360 frames (malformed input: frames shorter than a keepalive)
 * sends the whole scan to the one-lane kernel: correct, slow. */
int32_t etlg_scan_boundaries(etlg_ctx* ctx, const uint8_t* buf, size_t len, uint32_t flags,
                             uint32_t* offsets_out, size_t cap, size_t* nframes_out);

/* The pgoutput tag of every frame of a staged stream (the same classification etlg_decode runs first): tags_out[i] is
 * 'B' 'C' 'R' 'I' 'U' 'D' 'T' 'M' 'Y' 'O', 'k' for a keepalive, 0 for a malformed frame. It is what a host needs to cut
 * a stream into commit-aligned shards (cuts go after a 'C') and to find the rare Relation / DDL-message frames whose effects
 * every later shard must see (etl_amd/shard.py: plan_shards, control_stream) without reading the stream back.
 * Replaces, as a pre-pass, the message-kind dispatch of the apply loop (crates/etl/src/replication/apply.rs:2087-2125).
 * frame_offsets: nframes + 1 entries (required; e.g. from etlg_scan_boundaries).
 * flags: ETLG_F_INPUT_ON_DEVICE (buf / frame_offsets are device pointers), ETLG_F_OUTPUT_ON_DEVICE (tags_out is one). */
int32_t etlg_frame_tags(etlg_ctx* ctx, const uint8_t* buf, size_t len, const uint32_t* frame_offsets, size_t nframes,
                        uint32_t flags, uint8_t* tags_out);

/* Commit-aligned shard cuts of a staged stream (multi-GPU recipe): frames [0, nframes) as n_shards contiguous
 * ranges balanced by bytes, every interior cut right after a Commit ('C') frame — the transaction state (commit_lsn, next ordinal:
 * crates/etl/src/replication/apply.rs:942-963) is then shard-local. The frames are classified and the cuts found on the device; only
 * the n_shards - 1 cut indexes come back. cuts_out (HOST, n_shards + 1 entries): shard k = frames [cuts_out[k], cuts_out[k + 1]);
 * cuts_out[0] = 0, cuts_out[n_shards] = nframes; a stream with fewer Commits than shards leaves some ranges empty (cuts never go back).
 * The reference has no counterpart: it decodes one ordered stream in one task (apply.rs:1210-1336).
 * flags: ETLG_F_INPUT_ON_DEVICE (buf / frame_offsets are device pointers). n_shards <= 4096. */
int32_t etlg_shard_plan(etlg_ctx* ctx, const uint8_t* buf, size_t len, const uint32_t* frame_offsets, size_t nframes, uint32_t n_shards,
                        uint32_t flags, uint64_t* cuts_out);

/* Step 3 of the recipe: apply the control stream of a shard BEFORE this context's own (etlg_control_stream of that range, as the ranks
 * exchanged it) — its Relation and DDL-message frames update the schema store and the shared table cache exactly as in place
 * (apply.rs:2160-2276, 2363-2440) — drop the events it decodes to, and leave the context outside any transaction at ordinal 0, which is
 * where a commit-aligned shard starts. Call once per earlier shard, in rank order, then decode the own shard. HOST buffers; nframes = 0
 * only resets the transaction state. An error (a Relation frame the reference would reject) is the error of the shard that owns the
 * frame: it is returned here, and that shard's own decode reports it with its frame index. */
int32_t etlg_shard_replay(etlg_ctx* ctx, const uint8_t* buf, size_t len, const uint32_t* frame_offsets, size_t nframes);

/* The control stream of a frame range, extracted on the device (multi-GPU recipe): for every transaction of the
 * range that holds a Relation ('R') or logical-decoding Message ('M') frame, its Begin, those frames in order and its Commit
 * (a control frame outside any transaction of the range travels alone). Decoding that stream on another context has the same
 * effect on the schema store and the shared table cache as decoding the whole range (crates/etl/src/replication/apply.rs:
 * 2160-2276, 2363-2440 are the only writers) — it is what the ranks broadcast before they decode their shards.
 * out_bytes / out_offsets: HOST buffers for the frames and their n_frames + 1 offsets; n_bytes / n_frames are always set (a call
 * whose buffers are too small fails with ETLG_InvalidArgument and says what it needs). last_tag: the pgoutput tag of the range's
 * last frame ('C' when the range ends on a transaction boundary). Ranges without control frames — almost all — cost one
 * classification pass and an 8-byte copy; nothing else of the range leaves the device. */
int32_t etlg_control_stream(etlg_ctx* ctx, const uint8_t* buf, size_t len, const uint32_t* frame_offsets, size_t nframes,
                            uint32_t flags, uint8_t* out_bytes, size_t out_cap, uint32_t* out_offsets, size_t out_offsets_cap,
                            size_t* n_bytes, size_t* n_frames, uint32_t* last_tag);

/* ------------------------------------------------------------ batch (arena) */

/* Event kinds: the pgoutput tag of the message that produced the event. */
enum {
  ETLG_EV_BEGIN = 'B',
  ETLG_EV_COMMIT = 'C',
  ETLG_EV_RELATION = 'R',
  ETLG_EV_INSERT = 'I',
  ETLG_EV_UPDATE = 'U',
  ETLG_EV_DELETE = 'D',
  ETLG_EV_TRUNCATE = 'T'
};

/* ev_flags for U/D: bits 0-1 = old row kind, bit 2 = new row is Partial. */
enum {
  ETLG_OLD_NONE = 0,
  ETLG_OLD_FULL = 1, /* OldTableRow::Full */
  ETLG_OLD_KEY = 2,  /* OldTableRow::Key  */
  ETLG_FLAG_PARTIAL = 4
};

/* 2-bit per-cell state stored at the head of every row block. */
enum {
  ETLG_CELL_VALUE = 0,
  ETLG_CELL_NULL = 1,
  ETLG_CELL_MISSING = 2, /* only inside a Partial new row */
  ETLG_CELL_DEFERRED = 3 /* slot = (heap_off, len) of the source text; the host finishes
                            it with the reference's own parse_cell_from_postgres_text */
};

/* Which cells come back DEFERRED (the same rule is implemented by the oracle's CONTRACT mode):
 *  - json / jsonb and every array type: always;
 *  - date / time / timetz / timestamp / timestamptz: never. Texts in the fixed layout of the reference's fast paths
 *    (codec/time.rs:89-154) and the shapes it hands to chrono (`parse_from_str` with "%Y-%m-%d", "%H:%M:%S%.f",
 *    "%Y-%m-%d %H:%M:%S%.f": one-digit fields, whitespace in front of numbers, signed / long years, leap second
 *    :60 = second 59 with nanos + 10^9, more than nine fraction digits) are both decoded on the device; a text
 *    neither accepts is "Datetime parsing failed";
 *  - float4 / float8: decoded on the device whenever the result is certain: Clinger's exact path (mantissa
 *    <= 2^53, |exponent| <= 22: one IEEE operation), else the Eisel-Lemire algorithm on the first 19
 *    significant digits (as Rust's dec2flt does; a longer mantissa must round the same way for w and
 *    w + 1), plus zero, inf / infinity / nan. The rare text for which that is inconclusive (about 1 in
 *    10^4 of random decimal texts, essentially none of Postgres' own shortest-round-trip output) is
 *    DEFERRED. The rule is etl_amd/csrc/float_fast.h; the oracle evaluates the same header for the
 *    DECISION and glibc strtod / strtof for the value. Malformed text is the reference's
 *    "Float parsing failed". */

/* ---- the finish pass (round 6): typed arrays and exact floats in the arena itself.
 *
 * The reference decodes an array cell to Cell::Array(ArrayCell::*) at decode time (parse_cell_from_postgres_text_array,
 * crates/etl/src/postgres/codec/text.rs:163-312) and every float text with Rust's correctly rounded str::parse (:52-59).
 * etlg_decode leaves array cells — and the rare float text its fast rule cannot decide — ETLG_CELL_DEFERRED (above);
 * etlg_batch_finish_cells settles them on the device, in place:
 *  - ETLG_FINISH_ARRAYS: every DEFERRED cell of an array column whose element class is bool / int2 / int4 / int8 / oid /
 *    float4 / float8 / date / time / timetz / timestamp / timestamptz / uuid / numeric / bytea / text-like
 *    (etlg_array_elem_class) and whose literal the reference accepts becomes ETLG_CELL_VALUE; its slot then holds
 *    (heap_off, bytes) of an etlg_array_hdr entry appended behind the batch's heap (heap_bytes grows; the source text
 *    stays where it was). NULL elements, quoting, escapes, the optional dimensions prefix, the unquoted-NULL rule:
 *    exactly the reference's state machine. A literal the reference REJECTS (unbalanced quotes, a multidimensional
 *    array, an element its type's parser refuses ...), a json / jsonb element, or a numeric / timetz element of more
 *    than 40 characters is left DEFERRED: the host's parse_cell_from_postgres_text then raises the reference's own
 *    error (or finishes the cell), as before.
 *  - ETLG_FINISH_FLOATS: every DEFERRED float4 / float8 cell gets its correctly rounded bits (the conversion Rust's
 *    dec2flt falls back to: 768 decimal digits shifted into place, core::num::dec2flt::slow) and becomes
 *    ETLG_CELL_VALUE; float elements of arrays likewise. With both bits set the only DEFERRED cells left in a batch
 *    are json / jsonb, json arrays and literals the reference rejects.
 * The batch must be device-resident and finished (an ASYNC batch is synced first). Events, row blocks and every
 * heap reference that existed stay where they are. */
enum { ETLG_FINISH_ARRAYS = 1u, ETLG_FINISH_FLOATS = 2u };

/* A typed array in the heap (4-byte aligned, a multiple of 4 bytes long):
 *   etlg_array_hdr
 *   uint32_t validity[(n_elems + 31) / 32]      bit k set: element k is a value (clear: NULL)
 *   elem_bytes != 0:  n_elems slots of elem_bytes bytes, each laid out like a row slot of elem_class (NULL: zeros)
 *   elem_bytes == 0:  uint32_t end[n_elems] — element k is data[end[k-1] .. end[k]) (end[-1] = 0) — then the data,
 *                     zero padded to 4 bytes: String / Bytes elements are their unescaped bytes; a Numeric element is an
 *                     etlg_numeric_hdr + digits, padded to 4 bytes (its end offset includes the padding) */
typedef struct etlg_array_hdr {
  uint32_t n_elems;
  uint8_t elem_class; /* etlg_type_class of the elements */
  uint8_t elem_bytes; /* etlg_slot_bytes(elem_class) for fixed-width elements, 0 for String / Bytes / Numeric */
  uint16_t reserved;  /* 0 */
} etlg_array_hdr;

typedef struct etlg_finish_stats {
  uint64_t deferred_seen;  /* DEFERRED cells of array / float columns the pass looked at */
  uint64_t arrays_typed;
  uint64_t floats_settled;
  uint64_t left_deferred;  /* of deferred_seen: still DEFERRED (the host finishes them) */
  uint64_t heap_bytes_added;
} etlg_finish_stats;

/* what: ETLG_FINISH_* bits. stats may be NULL. Idempotent (a second call finds nothing it can settle). */
int32_t etlg_batch_finish_cells(etlg_ctx* ctx, etlg_batch* batch, uint32_t what, etlg_finish_stats* stats);

/* Numeric heap entry header (followed by ndigits little-endian i16 base-10000
 * digits): mirrors PgNumeric (crates/etl-postgres/src/numeric.rs:75-96). */
enum { ETLG_NUM_VALUE = 0, ETLG_NUM_NAN = 1, ETLG_NUM_PINF = 2, ETLG_NUM_NINF = 3 };
typedef struct etlg_numeric_hdr {
  uint8_t kind; /* ETLG_NUM_* */
  uint8_t sign; /* 0 positive, 1 negative */
  int16_t weight;
  uint16_t scale;
  uint16_t ndigits;
} etlg_numeric_hdr;

/* One replicated column of a schema slot. */
typedef struct etlg_slot_col {
  uint32_t type_oid;
  uint16_t stored_index; /* index into the stored (attnum-ordered) schema */
  uint8_t type_class;    /* etlg_type_class */
  uint8_t nullable;
  uint8_t identity;      /* replicated && identity */
  uint8_t _pad;
  uint16_t off_full;     /* byte offset of the slot inside a full-layout row block */
  uint16_t off_key;      /* byte offset inside a key-layout row block (identity cols only) */
  uint16_t key_index;    /* position among identity columns, 0xFFFF if not identity */
} etlg_slot_col;

/* A ReplicatedTableSchema instance (crates/etl/src/schema.rs:380-441): the
 * stored schema at one snapshot + replication/identity masks. */
typedef struct etlg_slot_desc {
  uint32_t table_id;
  uint32_t n_stored;     /* columns in the stored schema */
  uint64_t snapshot_lsn; /* snapshot id of the stored schema */
  uint32_t n_cols;       /* replicated columns */
  uint32_t n_ident;      /* replicated identity columns */
  uint32_t row_bytes_full; /* state bytes (padded to 4) + slots */
  uint32_t row_bytes_key;
  uint32_t state_bytes_full; /* = 4*ceil(n_cols/16) */
  uint32_t state_bytes_key;  /* = 4*ceil(n_ident/16) */
  const etlg_slot_col* cols; /* n_cols entries, replicated order */
} etlg_slot_desc;

/* Columnar view of a decoded batch. Every array has n_events entries.
 *
 *   kind 'B': table_id = xid;            body = i64 commit timestamp
 *   kind 'C': flags    = commit flags;   body = u64 end_lsn, i64 timestamp
 *   kind 'R': table_id, schema_slot;     no body
 *   kind 'I': table_id, schema_slot;     body = new row (full layout)
 *   kind 'U': flags = old kind|partial;  body = [old row][new row]
 *   kind 'D': flags = old kind;          body = [old row]
 *   kind 'T': flags = options, table_id = n owned tables;
 *                                         body = n x {u32 table_id, u32 schema_slot}
 *
 * Row block (full layout: all replicated columns; key layout: identity
 * columns only): 2-bit cell states, column i at bits 2*(i%4) of byte i/4,
 * padded to 4 bytes; then one slot per column at off_full/off_key.
 * Slot contents, little-endian, zero for NULL/MISSING cells:
 *   BOOL u32 0/1 | I16 i32 | I32 i32 | U32 u32 | I64 i64 | F32 u32 bits (8-byte slot)
 *   F64 u64 bits | DATE i32 days from CE (chrono NaiveDate, 8-byte slot)
 *   TIME u32 secs-of-day, u32 nanos | TIMESTAMP/TIMESTAMPTZ i32 days, u32 secs, u32 nanos
 *   TIMETZ u32 secs, u32 nanos, i32 offset seconds east | UUID 16 bytes
 *   STRING/BYTEA/NUMERIC/JSON/ARRAY and any DEFERRED cell: u32 heap_off, u32 len
 * Heap entries are 4-byte aligned and zero padded; STRING/DEFERRED hold the
 * source text verbatim, BYTEA the decoded bytes, NUMERIC an etlg_numeric_hdr
 * + digits (len = 8 + 2*ndigits). A toast cell resolved from the old row
 * aliases the old row's heap entry (Cell::clone).
 * body_off is in bytes into `fixed`; bodies are contiguous in event order.
 */
typedef struct etlg_batch_view {
  uint64_t n_events;
  uint64_t n_frames;        /* frames consumed (index of the failing frame on error) */
  uint64_t fixed_bytes;
  uint64_t heap_bytes;
  uint64_t payload_bytes[3]; /* insert/update/delete value bytes (codec/event.rs:261-297) */
  const uint8_t* ev_kind;
  const uint8_t* ev_flags;
  const uint32_t* ev_table_id;
  const uint32_t* ev_schema_slot;
  const uint64_t* ev_start_lsn;
  const uint64_t* ev_commit_lsn;
  const uint64_t* ev_tx_ordinal;
  const uint64_t* ev_body_off;
  const uint8_t* fixed;
  const uint8_t* heap;
  uint32_t on_device;       /* 1: the pointers above are device pointers */
  uint32_t n_slots;         /* schema slots known to the context (ids are stable across batches) */
  const etlg_slot_desc* slots; /* host memory */
} etlg_batch_view;

int32_t etlg_batch_view_get(const etlg_batch* batch, etlg_batch_view* out);
/* Wait for an ETLG_F_ASYNC batch and refresh its counts/error. */
int32_t etlg_batch_sync(etlg_ctx* ctx, etlg_batch* batch);
/* Enqueue (no synchronisation) a device-to-device copy of the batch's 64-byte
 * result header into `dst_device_8xu64`:
 *   {first_err key (~0 = none), n_events, fixed_bytes, heap_bytes,
 *    payload insert/update/delete bytes, n_frames}
 * Must be called right after the etlg_decode that produced `batch`. This is the
 * per-shard record the multi-GPU all-gather exchanges (one fixed-size header per
 * rank; rank order == LSN order). For a finished batch the copy is enqueued on the
 * context's stream; for an ETLG_F_ASYNC batch in flight it travels on a private
 * stream behind the batch's kernels (a copy between two decode kernels costs a
 * dispatch gap): call etlg_ctx_fence before work on the context's stream reads it. */
int32_t etlg_batch_header_to_device(etlg_ctx* ctx, etlg_batch* batch, void* dst_device_8xu64);
/* Makes the context's stream wait (on the device; the host does not block) for every
 * ETLG_F_ASYNC batch enqueued so far: their decode kernels — consecutive batches may
 * run on two private streams side by side — and their header copies. Device work the
 * caller enqueues on the context's stream afterwards sees their arenas and headers. */
int32_t etlg_ctx_fence(etlg_ctx* ctx);
/* Copy a device-resident (ETLG_F_OUTPUT_ON_DEVICE) batch into host memory;
 * afterwards the view holds host pointers. */
int32_t etlg_batch_download(etlg_ctx* ctx, etlg_batch* batch);
void etlg_batch_free(etlg_batch* batch);

/* ------------------------------------------------------ columnar hand-off */

/* Arrow-layout column buffers for the rows of ONE schema slot of a decoded batch,
 * built on the device from the arena (no Cell objects, no per-row host work).
 * Replaces rows_to_record_batch / build_array_for_field and the cell_to_*
 * converters of the reference's Arrow sinks
 * (crates/etl-destinations/src/iceberg/encoding.rs:34-84, :150-360). */
typedef enum etlg_arrow_kind {
  ETLG_AK_BOOLEAN = 0,      /* values bit-packed, LSB first */
  ETLG_AK_INT32 = 1,        /* I16, I32 (cell_to_i32, encoding.rs:157) */
  ETLG_AK_INT64 = 2,        /* I64, U32 (cell_to_i64, encoding.rs:164) */
  ETLG_AK_FLOAT32 = 3,
  ETLG_AK_FLOAT64 = 4,
  ETLG_AK_DATE32 = 5,       /* days since 1970-01-01 (encoding.rs:194) */
  ETLG_AK_TIME64_US = 6,    /* encoding.rs:201 */
  ETLG_AK_TIMESTAMP_US = 7, /* encoding.rs:208 */
  ETLG_AK_TIMESTAMP_US_UTC = 8, /* encoding.rs:215 */
  ETLG_AK_FIXED16 = 9,      /* uuid: FixedSizeBinary(16) */
  ETLG_AK_LARGE_UTF8 = 10,  /* i64 offsets + bytes */
  ETLG_AK_LARGE_BINARY = 11,
  ETLG_AK_TEXT_FORM = 12,   /* json / arrays: the cell's heap entry (the source text), i64 offsets + bytes; the host finishes it. A json /
                             * jsonb cell has been checked on the device to be one JSON value under serde_json's rules (codec/text.rs:
                             * 126-134): a malformed one fails the call with ETLG_E_JSON at its event, so the host's parse cannot fail (except behind a list
                             * row handed back DEFERRED, see ETLG_ROWS_PARSE_ARRAYS: then the host meets it in event order).
                             * (numeric and timetz columns are ETLG_AK_LARGE_UTF8 of their Display strings — `n.to_string()`,
                             * `t.to_string()`, cell_to_string encoding.rs:349-352 — formatted on the device) */
  ETLG_AK_LIST = 13,        /* only with ETLG_ROWS_PARSE_ARRAYS: array literals parsed on the device (parse_cell_from_postgres_text_array,
                             * codec/text.rs:228-312): i64 list offsets in `offsets`, child validity in `child_validity`, child values
                             * in `values` (child_kind layout). Elements: bool / int2 / int4 / int8 / oid / float4 / float8 / date / time /
                             * timestamp / timestamptz / uuid (fixed-width children), text and every array type without a dedicated
                             * arm (LargeUtf8 child), numeric / timetz (LargeUtf8 child of the elements' Display strings, as the sinks
                             * write them), bytea (LargeBinary child of the decoded bytes). Var-len children: `child_offsets`.
                             * json[] / jsonb[] (round 6): LargeUtf8 child of `j.to_string()` per element (ArrayCell::Json,
                             * iceberg/encoding.rs:577-585); an element of more than 256 bytes or beyond the json writer's limits
                             * (depth 16, 64 members) hands its row back (`deferred`), one that is not JSON is ETLG_E_JSON */
  ETLG_AK_NONE = 255        /* not handed off - no buffers (no class maps to it today) */
} etlg_arrow_kind;

typedef struct etlg_column {
  uint32_t type_class;      /* etlg_type_class of the replicated column */
  uint32_t arrow_kind;      /* etlg_arrow_kind */
  uint32_t value_bytes;     /* bytes per row in `values` (0: bit-packed or var-len) */
  uint32_t nullable;
  uint64_t null_count;      /* rows whose validity bit is 0 */
  uint64_t deferred_count;  /* rows set in `deferred` */
  const uint8_t* validity;  /* n_rows bits, LSB first, padded to 8 bytes */
  const uint8_t* deferred;  /* same shape: cells the kernels handed back ETLG_CELL_DEFERRED. In a fixed-width
                             * column they are null in `validity`; in a TEXT_FORM column their entry is the source text */
  const uint8_t* values;
  const int64_t* offsets;   /* var-len kinds: n_rows + 1 entries, else NULL */
  uint64_t values_bytes;    /* bytes behind `values` */
  /* ETLG_AK_LIST only */
  uint32_t child_kind;      /* the element column's etlg_arrow_kind (BOOLEAN, INT32, INT64, FLOAT32/64, DATE32, TIME64_US, TIMESTAMP_US[_UTC], FIXED16,
                             * LARGE_UTF8) */
  uint32_t _pad;
  uint64_t child_count;     /* elements of all rows = offsets[n_rows] */
  uint64_t child_null_count;
  const uint8_t* child_validity; /* child_count bits */
  const int64_t* child_offsets;  /* child_kind ETLG_AK_LARGE_UTF8 (text[] and every array without a dedicated element type): child_count + 1
                                  * byte offsets into `values`, the unescaped element texts back to back */
} etlg_column;

typedef struct etlg_columns_view {
  uint64_t n_rows;
  uint32_t n_cols;          /* replicated columns of the slot, slot order */
  uint32_t on_device;       /* 1: buffer pointers are device pointers */
  const etlg_column* cols;  /* host memory */
  const uint64_t* row_event; /* n_rows: index of the event each row came from (event order) */
} etlg_columns_view;

typedef struct etlg_columns etlg_columns;

#define ETLG_ROWS_INSERT 1u  /* Insert rows */
#define ETLG_ROWS_UPDATE 2u  /* the new row of non-partial Updates */
#define ETLG_ROWS_PARSE_ARRAYS 4u /* array columns of a fixed-width element class become ETLG_AK_LIST; a malformed literal
                                   * fails the call with the reference's error (ETLG_E_ARRAY_* / the element's parse error,
                                   * frame_index = the event index), as parse_cell_from_postgres_text does at decode time — unless
                                   * a row the device hands back DEFERRED (an element of more than 40 characters, a float text its
                                   * rule does not settle) precedes the first malformed one: then the call succeeds and the
                                   * malformed rows are handed back too, so that the consumer, finishing the deferred rows in event
                                   * order, meets the first problem first */
#define ETLG_ROWS_FORMAT_JSON 8u  /* json / jsonb columns become ETLG_AK_LARGE_UTF8 of serde_json's `Value::to_string()` — what the sinks
                                   * write (`Cell::Json(j) => j.to_string()`: iceberg/encoding.rs:356, ducklake/encoding.rs:173): compact,
                                   * object members in the byte order of their decoded keys with the last of repeated keys, strings escaped
                                   * again, number literals kept (an unsigned exponent gains '+': codec/text.rs:812-815). A cell beyond what
                                   * a lane does (nesting deeper than 16, an object of more than 64 members) keeps its source text and its
                                   * bit in `deferred`, like every cell of an ETLG_AK_TEXT_FORM column; a cell that is not one JSON value
                                   * fails the call with ETLG_E_JSON as without the flag. etlg_batch_rowbinary / _protobuf always write
                                   * the Display string (and report such a cell as host_event / host_column). */

/* `batch` must be device-resident (decoded with ETLG_F_OUTPUT_ON_DEVICE, not downloaded) and finished
 * (an ETLG_F_ASYNC batch is synced first). flags: ETLG_F_OUTPUT_ON_DEVICE keeps the buffers in HBM, otherwise
 * they are copied to host memory. An unchanged-toast cell cannot occur (partial rows are not selected). */
int32_t etlg_batch_columns(etlg_ctx* ctx, etlg_batch* batch, int32_t schema_slot, uint32_t row_kinds,
                           uint32_t flags, etlg_columns** out);
int32_t etlg_columns_view_get(const etlg_columns* cols, etlg_columns_view* out);
/* Returns the buffers to the context's pool (no device synchronisation): work the caller enqueued on them must be complete. */
void etlg_columns_free(etlg_columns* cols);

/* ClickHouse RowBinary rows for ONE schema slot of a decoded batch, encoded on the device: what
 * cell_to_clickhouse_value + encode_to_row_binary (crates/etl-destinations/src/clickhouse/encoding.rs:58-83,
 * :188-283) and append_cdc_columns (clickhouse/core.rs:96-114) produce for the rows core.rs:1078-1127 collects:
 * Insert -> the row; Update -> the new row; Delete -> the old row, and for a Delete that carries only the key the tombstone row
 * expand_key_row builds (core.rs:1437-1472: the key cells in the primary-key columns, NULL in every other column that is nullable
 * at the source and not an array, default_cell's zero value (:1481-1517) in the rest) when the slot's replica identity is the primary
 * key or Full (ensure_clickhouse_key_identity_is_primary_key, :1404-1427) and no nullable non-key column has a type outside the value
 * codec's table (is_array_type decides NULL or empty array there; only the host knows every array type). Events the reference
 * refuses or the device does not build (a Partial update, key-only Deletes of other slots) are left out and counted in n_host_rows.
 * Columns of class numeric / timetz / time are Display strings in the reference (`n.to_string()`, encoding.rs:66-71): they are
 * formatted on the device (PgNumeric Display, crates/etl-postgres/src/numeric.rs:460-560; PgTimeTz, etl-postgres/src/time.rs:113-117,
 * 210-225); so is a json / jsonb cell (`j.to_string()`, encoding.rs:73: serde_json's Display, see ETLG_ROWS_FORMAT_JSON — a cell that
 * is not one JSON value fails the call with ETLG_E_JSON at its event, the reference's decode error, before any report of the sink's own).
 * Arrays of every element class leave as Array(Nullable(T)) (array_cell_to_clickhouse_values, encoding.rs:86-111): fixed-width values,
 * text-like elements as the unescaped text, numeric / timetz / json elements as their Display strings, bytea elements as bytes_to_hex.
 * A DEFERRED scalar cell / an array literal the device cannot take apart (malformed: the host raises the reference's decode error; a
 * numeric / timetz element of more than 40 characters, a json element of more than 256 bytes) / a json cell beyond json_display's
 * limits in a row makes the call return status ETLG_RB_NEEDS_HOST (no bytes; host_event / host_column name the first such cell). Arrays of fixed-width elements are encoded as Array(Nullable(T)). */
typedef enum etlg_ch_engine {
  ETLG_CH_MERGE_TREE = 0,           /* + cdc_operation String, cdc_lsn UInt64 */
  ETLG_CH_REPLACING_MERGE_TREE = 1  /* + _etl_version UInt128 (commit_lsn << 64 | tx_ordinal), _etl_deleted UInt8 */
} etlg_ch_engine;

#define ETLG_RB_OK 0u
#define ETLG_RB_NEEDS_HOST 3u

typedef struct etlg_rowbinary_view {
  uint64_t n_rows;
  uint64_t n_bytes;
  uint64_t n_host_rows;       /* Insert/Update/Delete events of the slot that are not among the rows */
  uint32_t status;            /* ETLG_RB_OK | ETLG_RB_NEEDS_HOST */
  uint32_t on_device;
  uint64_t host_event;        /* NEEDS_HOST: the first event (and column) the device cannot encode; ~0 = the slot's classes */
  uint32_t host_column;
  uint32_t _pad;
  const uint8_t* bytes;       /* rows back to back, event order */
  const int64_t* row_offsets; /* n_rows + 1 (the client cuts inserts at max_bytes_per_insert, client.rs:566-600) */
  const uint64_t* row_event;  /* n_rows */
} etlg_rowbinary_view;

typedef struct etlg_rowbinary etlg_rowbinary;

/* nullable_flags: one byte per destination column = the slot's replicated columns, then the engine's two CDC
 * columns (nullable_flags_from_clickhouse_columns, clickhouse/core.rs:162-209); n_flags must be n_cols + 2
 * ("ClickHouse RowBinary row width mismatch", encoding.rs:263-274, otherwise). A NULL in a non-nullable column and a
 * date outside 1900-01-01..=2299-12-31 fail the call with ETLG_ConversionError like the reference (etlg_last_error:
 * the description, frame_index = the event index). Same batch requirements as etlg_batch_columns. */
int32_t etlg_batch_rowbinary(etlg_ctx* ctx, etlg_batch* batch, int32_t schema_slot, const uint8_t* nullable_flags,
                             uint32_t n_flags, int32_t engine, uint32_t flags, etlg_rowbinary** out);
/* BigQuery Storage Write rows for ONE schema slot: the protobuf bytes of the BigQueryTableRows the sink builds for the slot's events
 * (crates/etl-destinations/src/bigquery/core.rs:978-1036; cell_encode_prost, bigquery/encoding.rs:120-190: field tag = column position
 * + 1, NULL cells leave nothing), every row closed by _CHANGE_TYPE and _CHANGE_SEQUENCE_NUMBER =
 * "{commit_lsn:016x}/{tx_ordinal:016x}/{ordinal:016x}" (core.rs:1405-1407):
 *   Insert -> one UPSERT row (ordinal 0);
 *   Update -> the new row as UPSERT; when the update changed the primary key — the old image's primary-key cells (key image or full
 *             old row) against the new row's, bigquery_primary_key_changed :1557-1645 — a sparse DELETE row of the OLD key goes first
 *             (ordinal 0) and the UPSERT takes ordinal 1 (:1425-1476): such an event is two consecutive rows with the same row_event;
 *   Delete -> the sparse DELETE row: the old image's primary-key cells under their column tags (bigquery_delete_row :1742-1754).
 * Events the reference refuses are left out and counted in n_host_rows (the host raises the reference's error when it meets the first):
 * a partial update, a delete without an old row, a key image or an update without an old row under a replica identity other than the
 * primary key (:1497-1555). So are updates that carry an old row when a primary-key column is of a class whose Cell equality is not
 * equality of the arena's words / bytes (float4 / float8, numeric, timetz), or a primary-key cell is DEFERRED.
 * numeric / timetz cells are their Display strings, a numeric with more than 38 decimal places fails the
 * call like validate_numeric_for_bigquery (bigquery/validation.rs:20-35: ETLG_UnsupportedValueInDestination, "Cell validation failed
 * for BigQuery compatibility", detail "Cell at index N failed validation", frame_index = the event). A json / jsonb cell is its
 * serde_json Display string (encoding.rs:173-176) behind validate_json_for_bigquery (validation.rs:47-85): an integer literal of the
 * parsed value outside u64 / i64 fails the call the same way; a cell that is not one JSON value fails it with ETLG_E_JSON. Array
 * cells (array_cell_encode_prost, encoding.rs:203-290): bool / int2 / int4 / oid / int8 / float4 / float8 / timestamptz packed; date /
 * time / timestamp / uuid, text-like, numeric, timetz and json elements one string field per element, bytea elements one bytes field;
 * an empty array nothing; a NULL element fails the call with ETLG_NullValuesNotSupportedInArrayInDestination (same description and
 * detail), a numeric element of more than 38 decimal places / a json element's integer outside u64 / i64 with
 * ETLG_UnsupportedValueInDestination (validate_elements, validation.rs:143-188). A literal the device cannot take apart (see
 * etlg_batch_rowbinary), DEFERRED cells and json cells beyond json_display's limits return ETLG_RB_NEEDS_HOST.
 * The result is an etlg_rowbinary (same view; n_rows can exceed the number of events). */
int32_t etlg_batch_protobuf(etlg_ctx* ctx, etlg_batch* batch, int32_t schema_slot, uint32_t flags, etlg_rowbinary** out);
int32_t etlg_rowbinary_view_get(const etlg_rowbinary* rb, etlg_rowbinary_view* out);
void etlg_rowbinary_free(etlg_rowbinary* rb);

/* --------------------------------------------------------------- size hints */

/* Event::size_hint (crates/etl/src/event.rs:295-320) for every event of a batch, computed on the device from the
 * arena: what EventBatch::push adds up to decide batch cut points (replication/apply.rs:656-657, 1932-1935).
 * The estimate is built from `size_of::<T>()` of the reference's own types (data/table_row.rs:248-384), whose
 * layouts are not ABI-stable: the Rust shim fills this model once (crates/etl-gfx950/src/lib.rs).
 * What is tested: the reference holds no test that states a size hint as a number, so there is nothing to pin the FORMULA's
 * restatement to — tests/test_gpu_size_hints.py checks the device against oracle/size_hint.py, a second restatement of the same
 * reading of event.rs / table_row.rs by the same hands. That is self-consistency, not parity with the reference. */
typedef struct etlg_size_model {
  uint32_t begin_event;             /* size_of::<BeginEvent>() ... (event.rs:297-316) */
  uint32_t commit_event;
  uint32_t insert_event;
  uint32_t update_event;
  uint32_t delete_event;
  uint32_t truncate_event;
  uint32_t relation_event;
  uint32_t replicated_table_schema; /* per truncated table (event.rs:311-314) */
  uint32_t table_row;               /* size_of::<TableRow>()  (table_row.rs:249-251) */
  uint32_t cell;                    /* size_of::<Cell>(), times the row's Vec capacity: n_cols (codec/event.rs:567),
                                       n_ident for key rows (:800, :831) */
  uint32_t _reserved[2];
} etlg_size_model;

/* The event has a part only the host can size: a DEFERRED json / array / numeric cell (estimate_json_allocated_bytes /
 * estimate_array_allocated_bytes need the parsed value) or a Partial updated row (its Vec capacities depend on the
 * reserve/append sequence of codec/event.rs:636-660). The low bits hold everything else of the event. */
#define ETLG_SIZE_HINT_INCOMPLETE (1ull << 63)

/* out: n_events u64 entries; device memory when flags has ETLG_F_OUTPUT_ON_DEVICE, else host memory.
 * Same batch requirements as etlg_batch_columns. Per cell (estimate_cell_allocated_bytes, table_row.rs:276-299):
 * String -> len (str::to_owned), Bytes -> len (Vec::with_capacity(hex / 2), codec/hex.rs:21), Numeric::Value ->
 * 2 * ndigits (Vec::with_capacity(retained_groups), numeric.rs:444), everything fixed-width -> 0. */
int32_t etlg_batch_size_hints(etlg_ctx* ctx, etlg_batch* batch, const etlg_size_model* model, uint32_t flags,
                              uint64_t* out);

/* Schema slots known to the context (also reachable from every batch view). */
int32_t etlg_ctx_slots(const etlg_ctx* ctx, uint32_t* n_slots,
                       const etlg_slot_desc** slots);

/* ------------------------------------------------------------- measurement */

/* HIP-event timing of the kernels launched by etlg_decode, each on the stream
 * it is launched on (bench.py's roofline leg). enable = 1: as the chain runs —
 * consecutive ETLG_F_ASYNC batches side by side on two decode streams, so their
 * durations overlap; enable = 2: the chain kept on one stream, i.e. every
 * kernel timed alone; 0: off (resets the counters). */
typedef struct etlg_kernel_stat {
  const char* name;
  uint64_t launches;
  double total_ms;
} etlg_kernel_stat;

int32_t etlg_ctx_profile(etlg_ctx* ctx, int32_t enable);
int32_t etlg_ctx_profile_read(etlg_ctx* ctx, etlg_kernel_stat* out,
                              uint32_t cap, uint32_t* n);

#ifdef __cplusplus
}
#endif
#endif /* ETLG_H */
