//! Raw binding of `include/etlg.h` (ABI version 1). One `extern "C"` item per exported symbol, one `#[repr(C)]` struct per
//! C struct, same field order and types; `tests/test_rust_shim.py` in the etl-gfx950 repository parses this file and the
//! header and fails when they drift apart.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_void};

pub const ETLG_ABI_VERSION: u32 = 1;

#[repr(C)]
pub struct etlg_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct etlg_batch {
    _private: [u8; 0],
}

// ---- errors (etlg_error_kind / etlg_err_code)
pub const ETLG_OK: i32 = 0;
pub const ETLG_ConversionError: i32 = 1;
pub const ETLG_InvalidData: i32 = 2;
pub const ETLG_ValidationError: i32 = 3;
pub const ETLG_InvalidState: i32 = 4;
pub const ETLG_MissingTableSchema: i32 = 5;
pub const ETLG_CorruptedTableSchema: i32 = 6;
pub const ETLG_DeserializationError: i32 = 7;
pub const ETLG_SourceConnectionFailed: i32 = 8;
pub const ETLG_IoError: i32 = 9;
pub const ETLG_UnsupportedValueInDestination: i32 = 10;
pub const ETLG_NullValuesNotSupportedInArrayInDestination: i32 = 11;
pub const ETLG_InvalidArgument: i32 = 100;
pub const ETLG_DeviceError: i32 = 101;
pub const ETLG_Unsupported: i32 = 102;

#[repr(C)]
pub struct etlg_error {
    pub kind: i32,
    pub code: i32,
    pub description: *const c_char,
    pub detail: *const c_char,
    pub frame_index: i64,
}

#[repr(C)]
pub struct etlg_err_desc {
    pub kind: i32,
    pub description: *const c_char,
}

// ---- schemas
pub const ETLG_TC_STRING: u8 = 0;
pub const ETLG_TC_BOOL: u8 = 1;
pub const ETLG_TC_I16: u8 = 2;
pub const ETLG_TC_I32: u8 = 3;
pub const ETLG_TC_I64: u8 = 4;
pub const ETLG_TC_U32: u8 = 5;
pub const ETLG_TC_F32: u8 = 6;
pub const ETLG_TC_F64: u8 = 7;
pub const ETLG_TC_NUMERIC: u8 = 8;
pub const ETLG_TC_BYTEA: u8 = 9;
pub const ETLG_TC_DATE: u8 = 10;
pub const ETLG_TC_TIME: u8 = 11;
pub const ETLG_TC_TIMETZ: u8 = 12;
pub const ETLG_TC_TIMESTAMP: u8 = 13;
pub const ETLG_TC_TIMESTAMPTZ: u8 = 14;
pub const ETLG_TC_UUID: u8 = 15;
pub const ETLG_TC_JSON: u8 = 16;
pub const ETLG_TC_ARRAY: u8 = 17;

#[repr(C)]
pub struct etlg_col {
    pub name: *const c_char,
    pub type_oid: u32,
    pub type_modifier: i32,
    pub attnum: i32,
    pub nullable: u8,
    pub primary_key: u8,
    pub _pad: [u8; 2],
}

pub const ETLG_TS_ABSENT: i32 = 0;
pub const ETLG_TS_READY: i32 = 1;
pub const ETLG_TS_SYNC_DONE: i32 = 2;
pub const ETLG_TS_OTHER: i32 = 3;
pub const ETLG_WORKER_APPLY: i32 = 0;
pub const ETLG_WORKER_TABLE_SYNC: i32 = 1;

// ---- decode flags
pub const ETLG_F_INPUT_ON_DEVICE: u32 = 1 << 0;
pub const ETLG_F_OUTPUT_ON_DEVICE: u32 = 1 << 1;
pub const ETLG_F_NO_CONTROL: u32 = 1 << 2;
pub const ETLG_F_ASYNC: u32 = 1 << 3;
pub const ETLG_F_FINISH_CELLS: u32 = 1 << 4;
pub const ETLG_FINISH_ARRAYS: u32 = 1;
pub const ETLG_FINISH_FLOATS: u32 = 2;

// ---- batch (arena)
pub const ETLG_EV_BEGIN: u8 = b'B';
pub const ETLG_EV_COMMIT: u8 = b'C';
pub const ETLG_EV_RELATION: u8 = b'R';
pub const ETLG_EV_INSERT: u8 = b'I';
pub const ETLG_EV_UPDATE: u8 = b'U';
pub const ETLG_EV_DELETE: u8 = b'D';
pub const ETLG_EV_TRUNCATE: u8 = b'T';
pub const ETLG_OLD_NONE: u8 = 0;
pub const ETLG_OLD_FULL: u8 = 1;
pub const ETLG_OLD_KEY: u8 = 2;
pub const ETLG_FLAG_PARTIAL: u8 = 4;
pub const ETLG_CELL_VALUE: u8 = 0;
pub const ETLG_CELL_NULL: u8 = 1;
pub const ETLG_CELL_MISSING: u8 = 2;
pub const ETLG_CELL_DEFERRED: u8 = 3;
pub const ETLG_NUM_VALUE: u8 = 0;
pub const ETLG_NUM_NAN: u8 = 1;
pub const ETLG_NUM_PINF: u8 = 2;
pub const ETLG_NUM_NINF: u8 = 3;

#[repr(C)]
pub struct etlg_numeric_hdr {
    pub kind: u8,
    pub sign: u8,
    pub weight: i16,
    pub scale: u16,
    pub ndigits: u16,
}

#[repr(C)]
pub struct etlg_slot_col {
    pub type_oid: u32,
    pub stored_index: u16,
    pub type_class: u8,
    pub nullable: u8,
    pub identity: u8,
    pub _pad: u8,
    pub off_full: u16,
    pub off_key: u16,
    pub key_index: u16,
}

#[repr(C)]
pub struct etlg_slot_desc {
    pub table_id: u32,
    pub n_stored: u32,
    pub snapshot_lsn: u64,
    pub n_cols: u32,
    pub n_ident: u32,
    pub row_bytes_full: u32,
    pub row_bytes_key: u32,
    pub state_bytes_full: u32,
    pub state_bytes_key: u32,
    pub cols: *const etlg_slot_col,
}

#[repr(C)]
pub struct etlg_batch_view {
    pub n_events: u64,
    pub n_frames: u64,
    pub fixed_bytes: u64,
    pub heap_bytes: u64,
    pub payload_bytes: [u64; 3],
    pub ev_kind: *const u8,
    pub ev_flags: *const u8,
    pub ev_table_id: *const u32,
    pub ev_schema_slot: *const u32,
    pub ev_start_lsn: *const u64,
    pub ev_commit_lsn: *const u64,
    pub ev_tx_ordinal: *const u64,
    pub ev_body_off: *const u64,
    pub fixed: *const u8,
    pub heap: *const u8,
    pub on_device: u32,
    pub n_slots: u32,
    pub slots: *const etlg_slot_desc,
}

#[repr(C)]
pub struct etlg_column {
    pub type_class: u32,
    pub arrow_kind: u32,
    pub value_bytes: u32,
    pub nullable: u32,
    pub null_count: u64,
    pub deferred_count: u64,
    pub validity: *const u8,
    pub deferred: *const u8,
    pub values: *const u8,
    pub offsets: *const i64,
    pub values_bytes: u64,
    pub child_kind: u32,
    pub _pad: u32,
    pub child_count: u64,
    pub child_null_count: u64,
    pub child_validity: *const u8,
    pub child_offsets: *const i64,
}

#[repr(C)]
pub struct etlg_columns_view {
    pub n_rows: u64,
    pub n_cols: u32,
    pub on_device: u32,
    pub cols: *const etlg_column,
    pub row_event: *const u64,
}

#[repr(C)]
pub struct etlg_rowbinary_view {
    pub n_rows: u64,
    pub n_bytes: u64,
    pub n_host_rows: u64,
    pub status: u32,
    pub on_device: u32,
    pub host_event: u64,
    pub host_column: u32,
    pub _pad: u32,
    pub bytes: *const u8,
    pub row_offsets: *const i64,
    pub row_event: *const u64,
}

#[repr(C)]
pub struct etlg_size_model {
    pub begin_event: u32,
    pub commit_event: u32,
    pub insert_event: u32,
    pub update_event: u32,
    pub delete_event: u32,
    pub truncate_event: u32,
    pub relation_event: u32,
    pub replicated_table_schema: u32,
    pub table_row: u32,
    pub cell: u32,
    pub _reserved: [u32; 2],
}

pub const ETLG_SIZE_HINT_INCOMPLETE: u64 = 1 << 63;

/// A typed array in the heap (include/etlg.h: the finish pass).
#[repr(C)]
pub struct etlg_array_hdr {
    pub n_elems: u32,
    pub elem_class: u8,
    pub elem_bytes: u8,
    pub reserved: u16,
}

#[repr(C)]
pub struct etlg_finish_stats {
    pub deferred_seen: u64,
    pub arrays_typed: u64,
    pub floats_settled: u64,
    pub left_deferred: u64,
    pub heap_bytes_added: u64,
}

#[repr(C)]
pub struct etlg_columns {
    _private: [u8; 0],
}

#[repr(C)]
pub struct etlg_rowbinary {
    _private: [u8; 0],
}

pub const ETLG_ROWS_INSERT: u32 = 1;
pub const ETLG_ROWS_UPDATE: u32 = 2;
pub const ETLG_ROWS_PARSE_ARRAYS: u32 = 4;
pub const ETLG_ROWS_FORMAT_JSON: u32 = 8;
pub const ETLG_CH_MERGE_TREE: i32 = 0;
pub const ETLG_CH_REPLACING_MERGE_TREE: i32 = 1;
pub const ETLG_RB_OK: u32 = 0;
pub const ETLG_RB_NEEDS_HOST: u32 = 3;

#[repr(C)]
pub struct etlg_kernel_stat {
    pub name: *const c_char,
    pub launches: u64,
    pub total_ms: f64,
}

extern "C" {
    pub fn etlg_err_table(code: i32) -> *const etlg_err_desc;
    pub fn etlg_type_class_of_oid(type_oid: u32) -> i32;
    pub fn etlg_array_elem_class(array_type_oid: u32) -> i32;
    pub fn etlg_slot_bytes(type_class: i32) -> u32;

    pub fn etlg_abi_version() -> u32;
    pub fn etlg_ctx_create(hip_device: i32, out: *mut *mut etlg_ctx) -> i32;
    pub fn etlg_create_error() -> *const c_char;
    pub fn etlg_ctx_destroy(ctx: *mut etlg_ctx);
    pub fn etlg_ctx_set_stream(ctx: *mut etlg_ctx, hip_stream: *mut c_void) -> i32;
    pub fn etlg_ctx_set_worker(ctx: *mut etlg_ctx, worker_kind: i32, table_sync_table_id: u32, bootstrap_snapshot_lsn: u64) -> i32;
    pub fn etlg_schema_put(
        ctx: *mut etlg_ctx,
        table_id: u32,
        snapshot_lsn: u64,
        schema_name: *const c_char,
        table_name: *const c_char,
        ncols: u32,
        cols: *const etlg_col,
    ) -> i32;
    pub fn etlg_table_state(ctx: *mut etlg_ctx, table_id: u32, state_kind: i32, lsn: u64) -> i32;
    pub fn etlg_table_ready(
        ctx: *mut etlg_ctx,
        table_id: u32,
        snapshot_lsn: u64,
        replication_mask: *const u8,
        identity_mask: *const u8,
        nmask: u32,
    ) -> i32;
    pub fn etlg_table_forget(ctx: *mut etlg_ctx, table_id: u32) -> i32;
    pub fn etlg_table_cache_get(ctx: *const etlg_ctx, table_id: u32, kind: *mut i32, snapshot_lsn: *mut u64, schema_slot: *mut i32) -> i32;
    pub fn etlg_ctx_reset_stream_state(ctx: *mut etlg_ctx) -> i32;

    pub fn etlg_decode(
        ctx: *mut etlg_ctx,
        buf: *const u8,
        len: usize,
        frame_offsets: *const u32,
        nframes: usize,
        flags: u32,
        out: *mut *mut etlg_batch,
    ) -> i32;
    pub fn etlg_last_error(ctx: *const etlg_ctx) -> *const etlg_error;
    pub fn etlg_copy_decode(
        ctx: *mut etlg_ctx,
        schema_slot: i32,
        buf: *const u8,
        len: usize,
        row_offsets: *const u32,
        nrows: usize,
        flags: u32,
        out: *mut *mut etlg_batch,
    ) -> i32;
    pub fn etlg_scan_boundaries(
        ctx: *mut etlg_ctx,
        buf: *const u8,
        len: usize,
        flags: u32,
        offsets_out: *mut u32,
        cap: usize,
        nframes_out: *mut usize,
    ) -> i32;
    pub fn etlg_frame_tags(
        ctx: *mut etlg_ctx,
        buf: *const u8,
        len: usize,
        frame_offsets: *const u32,
        nframes: usize,
        flags: u32,
        tags_out: *mut u8,
    ) -> i32;

    pub fn etlg_control_stream(
        ctx: *mut etlg_ctx,
        buf: *const u8,
        len: usize,
        frame_offsets: *const u32,
        nframes: usize,
        flags: u32,
        out_bytes: *mut u8,
        out_cap: usize,
        out_offsets: *mut u32,
        out_offsets_cap: usize,
        n_bytes: *mut usize,
        n_frames: *mut usize,
        last_tag: *mut u32,
    ) -> i32;
    pub fn etlg_shard_plan(
        ctx: *mut etlg_ctx,
        buf: *const u8,
        len: usize,
        frame_offsets: *const u32,
        nframes: usize,
        n_shards: u32,
        flags: u32,
        cuts_out: *mut u64,
    ) -> i32;
    pub fn etlg_shard_replay(ctx: *mut etlg_ctx, buf: *const u8, len: usize, frame_offsets: *const u32, nframes: usize) -> i32;
    pub fn etlg_host_alloc(ctx: *mut etlg_ctx, bytes: usize, out: *mut *mut c_void) -> i32;
    pub fn etlg_host_free(p: *mut c_void);
    pub fn etlg_batch_view_get(batch: *const etlg_batch, out: *mut etlg_batch_view) -> i32;
    pub fn etlg_batch_sync(ctx: *mut etlg_ctx, batch: *mut etlg_batch) -> i32;
    pub fn etlg_batch_header_to_device(ctx: *mut etlg_ctx, batch: *mut etlg_batch, dst_device_8xu64: *mut c_void) -> i32;
    pub fn etlg_ctx_fence(ctx: *mut etlg_ctx) -> i32;
    pub fn etlg_batch_download(ctx: *mut etlg_ctx, batch: *mut etlg_batch) -> i32;
    pub fn etlg_batch_free(batch: *mut etlg_batch);
    pub fn etlg_batch_columns(ctx: *mut etlg_ctx, batch: *mut etlg_batch, schema_slot: i32, row_kinds: u32, flags: u32, out: *mut *mut etlg_columns) -> i32;
    pub fn etlg_columns_view_get(cols: *const etlg_columns, out: *mut etlg_columns_view) -> i32;
    pub fn etlg_columns_free(cols: *mut etlg_columns);
    pub fn etlg_batch_rowbinary(
        ctx: *mut etlg_ctx,
        batch: *mut etlg_batch,
        schema_slot: i32,
        nullable_flags: *const u8,
        n_flags: u32,
        engine: i32,
        flags: u32,
        out: *mut *mut etlg_rowbinary,
    ) -> i32;
    pub fn etlg_batch_protobuf(ctx: *mut etlg_ctx, batch: *mut etlg_batch, schema_slot: i32, flags: u32, out: *mut *mut etlg_rowbinary) -> i32;
    pub fn etlg_rowbinary_view_get(rb: *const etlg_rowbinary, out: *mut etlg_rowbinary_view) -> i32;
    pub fn etlg_rowbinary_free(rb: *mut etlg_rowbinary);
    pub fn etlg_batch_size_hints(ctx: *mut etlg_ctx, batch: *mut etlg_batch, model: *const etlg_size_model, flags: u32, out: *mut u64) -> i32;
    pub fn etlg_batch_finish_cells(ctx: *mut etlg_ctx, batch: *mut etlg_batch, what: u32, stats: *mut etlg_finish_stats) -> i32;
    pub fn etlg_ctx_slots(ctx: *const etlg_ctx, n_slots: *mut u32, slots: *mut *const etlg_slot_desc) -> i32;

    pub fn etlg_ctx_profile(ctx: *mut etlg_ctx, enable: i32) -> i32;
    pub fn etlg_ctx_profile_read(ctx: *mut etlg_ctx, out: *mut etlg_kernel_stat, cap: u32, n: *mut u32) -> i32;
}
