//! Arena → `Vec<Event>`: rebuilds the values the apply loop pushes into `EventBatch` (crates/etl/src/replication/apply.rs:
//! 1918-1928) from the columnar batch `etlg_decode` returns (`include/etlg.h`, "batch (arena)"). Every `Cell` variant of
//! crates/etl/src/data/cell.rs:19-57 is covered. With `ETLG_F_FINISH_CELLS` (round 6) arrays arrive TYPED (`array_of` below reads
//! the entry the device wrote: no text is parsed on the host) and no float arrives DEFERRED; what the device still hands back
//! DEFERRED (json / jsonb, json arrays, a literal the reference rejects — and, without the flag, every array and the rare
//! inconclusive float) is finished HERE with the reference's own
//! `parse_cell_from_postgres_text` (crates/etl/src/postgres/codec/text.rs:32-153) on the validated source text the arena
//! carries, so such cells cannot differ from the CPU decoder by construction.
//!
//! `parse_cell_from_postgres_text` is `pub(crate)` upstream: either re-export it (`pub use` behind a feature in
//! crates/etl/src/postgres/codec/mod.rs) or move this file into crates/etl as `postgres/codec/gfx950.rs`.
use std::sync::Arc;

use chrono::{DateTime, NaiveDate, NaiveDateTime, NaiveTime, TimeZone, Utc};
use etl::data::{ArrayCell, Cell, OldTableRow, PartialTableRow, TableRow, UpdatedTableRow};
use etl::error::{ErrorKind, EtlResult};
use etl::event::{BeginEvent, CommitEvent, DeleteEvent, Event, InsertEvent, RelationEvent, TruncateEvent, UpdateEvent};
use etl::postgres::codec::text::parse_cell_from_postgres_text;
use etl::schema::ReplicatedTableSchema;
use etl::{bail, etl_error};
use etl_postgres::numeric::{PgNumeric, Sign};
use etl_postgres::time::PgTimeTz;
use etl_postgres::type_utils::convert_type_oid_to_type;
use tokio_postgres::types::PgLsn;
use uuid::Uuid;

use crate::ffi::*;

/// Resolves a schema slot of the context to the `ReplicatedTableSchema` the events carry. The host builds one per slot
/// from the slot descriptor (table id, snapshot id, replicated + identity columns) and its `SchemaStore`
/// (`ReplicatedTableSchema::from_masks`, crates/etl/src/schema.rs:380-441) the first time a batch names the slot.
pub trait SlotSchemas {
    fn schema_of(&mut self, slot: u32, desc: &etlg_slot_desc) -> EtlResult<ReplicatedTableSchema>;
}

struct Arena<'a> {
    v: &'a etlg_batch_view,
    fixed: &'a [u8],
    heap: &'a [u8],
}

impl<'a> Arena<'a> {
    /// SAFETY: `v` must hold HOST pointers (`on_device == 0`; call `etlg_batch_download` first) that outlive `'a`.
    unsafe fn new(v: &'a etlg_batch_view) -> Self {
        debug_assert_eq!(v.on_device, 0);
        Self {
            v,
            fixed: std::slice::from_raw_parts(v.fixed, v.fixed_bytes as usize),
            heap: std::slice::from_raw_parts(v.heap, v.heap_bytes as usize),
        }
    }
    fn u32_at(&self, off: usize) -> u32 {
        u32::from_le_bytes(self.fixed[off..off + 4].try_into().unwrap())
    }
    fn u64_at(&self, off: usize) -> u64 {
        u64::from_le_bytes(self.fixed[off..off + 8].try_into().unwrap())
    }
    fn heap_ref(&self, slot_off: usize) -> &'a [u8] {
        let (o, n) = (self.u32_at(slot_off) as usize, self.u32_at(slot_off + 4) as usize);
        &self.heap[o..o + n]
    }
}

fn time_of(secs: u32, nanos: u32) -> EtlResult<NaiveTime> {
    NaiveTime::from_num_seconds_from_midnight_opt(secs, nanos)
        .ok_or_else(|| etl_error!(ErrorKind::ConversionError, "Datetime parsing failed", "time of day out of range in decoded arena"))
}

fn date_of(days_from_ce: i32) -> EtlResult<NaiveDate> {
    NaiveDate::from_num_days_from_ce_opt(days_from_ce)
        .ok_or_else(|| etl_error!(ErrorKind::ConversionError, "Datetime parsing failed", "date out of range in decoded arena"))
}

/// One cell: `state` = the 2-bit cell state, `at` = offset of the column's slot inside `fixed`.
fn cell(a: &Arena<'_>, col: &etlg_slot_col, state: u8, at: usize) -> EtlResult<Option<Cell>> {
    match state {
        ETLG_CELL_NULL => return Ok(Some(Cell::Null)),
        ETLG_CELL_MISSING => return Ok(None),
        ETLG_CELL_DEFERRED => {
            // validated UTF-8 on the device (core::str::from_utf8 runs before the type switch, codec/event.rs:976)
            let text = unsafe { std::str::from_utf8_unchecked(a.heap_ref(at)) };
            return parse_cell_from_postgres_text(&convert_type_oid_to_type(col.type_oid), text).map(Some);
        }
        _ => {}
    }
    let c = match col.type_class {
        ETLG_TC_BOOL => Cell::Bool(a.u32_at(at) != 0),
        ETLG_TC_I16 => Cell::I16(a.u32_at(at) as i32 as i16),
        ETLG_TC_I32 => Cell::I32(a.u32_at(at) as i32),
        ETLG_TC_U32 => Cell::U32(a.u32_at(at)),
        ETLG_TC_I64 => Cell::I64(a.u64_at(at) as i64),
        ETLG_TC_F32 => Cell::F32(f32::from_bits(a.u32_at(at))),
        ETLG_TC_F64 => Cell::F64(f64::from_bits(a.u64_at(at))),
        ETLG_TC_DATE => Cell::Date(date_of(a.u32_at(at) as i32)?),
        ETLG_TC_TIME => Cell::Time(time_of(a.u32_at(at), a.u32_at(at + 4))?),
        ETLG_TC_TIMESTAMP => Cell::Timestamp(NaiveDateTime::new(date_of(a.u32_at(at) as i32)?, time_of(a.u32_at(at + 4), a.u32_at(at + 8))?)),
        ETLG_TC_TIMESTAMPTZ => {
            let naive = NaiveDateTime::new(date_of(a.u32_at(at) as i32)?, time_of(a.u32_at(at + 4), a.u32_at(at + 8))?);
            Cell::TimestampTz(Utc.from_utc_datetime(&naive) as DateTime<Utc>)
        }
        ETLG_TC_TIMETZ => {
            let offset = chrono::FixedOffset::east_opt(a.u32_at(at + 8) as i32)
                .ok_or_else(|| etl_error!(ErrorKind::ConversionError, "Datetime parsing failed", "UTC offset out of range in decoded arena"))?;
            Cell::TimeTz(PgTimeTz::new(time_of(a.u32_at(at), a.u32_at(at + 4))?, offset))
        }
        ETLG_TC_UUID => Cell::Uuid(Uuid::from_bytes(a.fixed[at..at + 16].try_into().unwrap())),
        ETLG_TC_BYTEA => Cell::Bytes(a.heap_ref(at).to_vec()),
        ETLG_TC_NUMERIC => {
            let raw = a.heap_ref(at);
            let (kind, sign) = (raw[0], raw[1]);
            let weight = i16::from_le_bytes([raw[2], raw[3]]);
            let scale = u16::from_le_bytes([raw[4], raw[5]]);
            let nd = u16::from_le_bytes([raw[6], raw[7]]) as usize;
            Cell::Numeric(match kind {
                ETLG_NUM_NAN => PgNumeric::NaN,
                ETLG_NUM_PINF => PgNumeric::PositiveInfinity,
                ETLG_NUM_NINF => PgNumeric::NegativeInfinity,
                _ => PgNumeric::Value {
                    sign: if sign == 0 { Sign::Positive } else { Sign::Negative },
                    weight,
                    scale,
                    digits: (0..nd).map(|i| i16::from_le_bytes([raw[8 + 2 * i], raw[9 + 2 * i]])).collect(),
                },
            })
        }
        // an array the finish pass typed on the device (ETLG_F_FINISH_CELLS / etlg_batch_finish_cells): no text is parsed here
        ETLG_TC_ARRAY => Cell::Array(array_of(a.heap_ref(at))?),
        // ETLG_TC_STRING; json never arrives as VALUE (always DEFERRED)
        _ => Cell::String(unsafe { String::from_utf8_unchecked(a.heap_ref(at).to_vec()) }),
    };
    Ok(Some(c))
}

/// A typed array entry (etlg_array_hdr, include/etlg.h) -> ArrayCell (crates/etl/src/data/cell.rs:98-134): header, validity words,
/// then element slots (laid out like row slots of the element class) or end offsets + bytes.
fn array_of(raw: &[u8]) -> EtlResult<ArrayCell> {
    let u32_at = |o: usize| u32::from_le_bytes(raw[o..o + 4].try_into().unwrap());
    let u64_at = |o: usize| u64::from_le_bytes(raw[o..o + 8].try_into().unwrap());
    let n = u32_at(0) as usize;
    let (elem, eb) = (raw[4] as i32, raw[5] as usize);
    let vw = (n + 31) / 32;
    let valid = |k: usize| (u32_at(8 + 4 * (k / 32)) >> (k % 32)) & 1 != 0;
    let at0 = 8 + 4 * vw;
    macro_rules! fixed {
        ($variant:ident, $f:expr) => {{
            let mut v = Vec::with_capacity(n);
            for k in 0..n {
                let o = at0 + k * eb;
                v.push(if valid(k) { Some($f(o)?) } else { None });
            }
            ArrayCell::$variant(v)
        }};
    }
    let ok = |x| -> EtlResult<_> { Ok(x) };
    Ok(match elem {
        ETLG_TC_BOOL => fixed!(Bool, |o| ok(u32_at(o) != 0)),
        ETLG_TC_I16 => fixed!(I16, |o| ok(u32_at(o) as i32 as i16)),
        ETLG_TC_I32 => fixed!(I32, |o| ok(u32_at(o) as i32)),
        ETLG_TC_U32 => fixed!(U32, |o| ok(u32_at(o))),
        ETLG_TC_I64 => fixed!(I64, |o| ok(u64_at(o) as i64)),
        ETLG_TC_F32 => fixed!(F32, |o| ok(f32::from_bits(u32_at(o)))),
        ETLG_TC_F64 => fixed!(F64, |o| ok(f64::from_bits(u64_at(o)))),
        ETLG_TC_DATE => fixed!(Date, |o| date_of(u32_at(o) as i32)),
        ETLG_TC_TIME => fixed!(Time, |o| time_of(u32_at(o), u32_at(o + 4))),
        ETLG_TC_TIMESTAMP => fixed!(Timestamp, |o| Ok::<_, etl::error::EtlError>(NaiveDateTime::new(date_of(u32_at(o) as i32)?, time_of(u32_at(o + 4), u32_at(o + 8))?))),
        ETLG_TC_TIMESTAMPTZ => fixed!(TimestampTz, |o| Ok::<_, etl::error::EtlError>(Utc.from_utc_datetime(&NaiveDateTime::new(date_of(u32_at(o) as i32)?, time_of(u32_at(o + 4), u32_at(o + 8))?)))),
        ETLG_TC_TIMETZ => fixed!(TimeTz, |o| {
            let offset = chrono::FixedOffset::east_opt(u32_at(o + 8) as i32)
                .ok_or_else(|| etl_error!(ErrorKind::ConversionError, "Datetime parsing failed", "UTC offset out of range in decoded arena"))?;
            Ok::<_, etl::error::EtlError>(PgTimeTz::new(time_of(u32_at(o), u32_at(o + 4))?, offset))
        }),
        ETLG_TC_UUID => fixed!(Uuid, |o| ok(Uuid::from_bytes(raw[o..o + 16].try_into().unwrap()))),
        _ => {
            // var-len elements: end offsets, then the data
            let data0 = at0 + 4 * n;
            let piece = |k: usize| {
                let lo = if k == 0 { 0 } else { u32_at(at0 + 4 * (k - 1)) as usize };
                &raw[data0 + lo..data0 + u32_at(at0 + 4 * k) as usize]
            };
            match elem {
                ETLG_TC_BYTEA => ArrayCell::Bytes((0..n).map(|k| valid(k).then(|| piece(k).to_vec())).collect()),
                ETLG_TC_NUMERIC => ArrayCell::Numeric((0..n).map(|k| valid(k).then(|| numeric_of(piece(k)))).collect()),
                _ => ArrayCell::String((0..n).map(|k| valid(k).then(|| unsafe { String::from_utf8_unchecked(piece(k).to_vec()) })).collect()),
            }
        }
    })
}

/// A numeric heap entry (etlg_numeric_hdr + base-10000 digits) -> PgNumeric (crates/etl-postgres/src/numeric.rs:75-96).
fn numeric_of(raw: &[u8]) -> PgNumeric {
    let (kind, sign) = (raw[0], raw[1]);
    let weight = i16::from_le_bytes([raw[2], raw[3]]);
    let scale = u16::from_le_bytes([raw[4], raw[5]]);
    let nd = u16::from_le_bytes([raw[6], raw[7]]) as usize;
    match kind {
        ETLG_NUM_NAN => PgNumeric::NaN,
        ETLG_NUM_PINF => PgNumeric::PositiveInfinity,
        ETLG_NUM_NINF => PgNumeric::NegativeInfinity,
        _ => PgNumeric::Value {
            sign: if sign == 0 { Sign::Positive } else { Sign::Negative },
            weight,
            scale,
            digits: (0..nd).map(|i| i16::from_le_bytes([raw[8 + 2 * i], raw[9 + 2 * i]])).collect(),
        },
    }
}

enum RowImage {
    Full(TableRow),
    Partial(PartialTableRow),
}

/// One row block (full layout: every replicated column; key layout: identity columns only).
fn row(a: &Arena<'_>, desc: &etlg_slot_desc, base: usize, key_layout: bool) -> EtlResult<RowImage> {
    let cols = unsafe { std::slice::from_raw_parts(desc.cols, desc.n_cols as usize) };
    let mut values = Vec::with_capacity(if key_layout { desc.n_ident } else { desc.n_cols } as usize);
    let mut missing = Vec::new();
    let mut i = 0usize; // cell index inside this image
    for col in cols {
        if key_layout && col.identity == 0 {
            continue;
        }
        let state = (a.fixed[base + i / 4] >> (2 * (i % 4))) & 3;
        let at = base + if key_layout { col.off_key } else { col.off_full } as usize;
        match cell(a, col, state, at)? {
            Some(c) => values.push(c),
            None => missing.push(i),
        }
        i += 1;
    }
    if missing.is_empty() {
        Ok(RowImage::Full(TableRow::new(values)))
    } else {
        Ok(RowImage::Partial(PartialTableRow::new(i, TableRow::new(values), missing)))
    }
}

fn full(r: RowImage) -> EtlResult<TableRow> {
    match r {
        RowImage::Full(t) => Ok(t),
        RowImage::Partial(_) => bail!(ErrorKind::InvalidState, "Decoded arena holds a partial image where a full row is required"),
    }
}

/// The rows of a table-copy batch (`etlg_copy_decode`: one Insert-shaped event per COPY row, in row order) — what
/// `TableCopyStream` yields item by item (crates/etl/src/postgres/stream/table_copy.rs:84-92).
///
/// SAFETY: `view` must describe a batch whose arrays are in host memory and stay alive for the call.
pub unsafe fn table_rows(view: &etlg_batch_view) -> EtlResult<Vec<TableRow>> {
    let a = Arena::new(view);
    let n = view.n_events as usize;
    let kind = std::slice::from_raw_parts(view.ev_kind, n);
    let slot = std::slice::from_raw_parts(view.ev_schema_slot, n);
    let body = std::slice::from_raw_parts(view.ev_body_off, n);
    let slots = std::slice::from_raw_parts(view.slots, view.n_slots as usize);
    let mut out = Vec::with_capacity(n);
    for i in 0..n {
        if kind[i] != ETLG_EV_INSERT {
            bail!(ErrorKind::InvalidState, "Table-copy batch holds an event that is not a row", kind[i]);
        }
        out.push(full(row(&a, &slots[slot[i] as usize], body[i] as usize, false)?)?);
    }
    Ok(out)
}

/// The events of one decoded batch, in stream order — what `EventBatch` would have received message by message.
///
/// SAFETY: `view` must describe a batch whose arrays are in host memory and stay alive for the call.
pub unsafe fn events(view: &etlg_batch_view, schemas: &mut dyn SlotSchemas) -> EtlResult<Vec<Event>> {
    let a = Arena::new(view);
    let n = view.n_events as usize;
    let kind = std::slice::from_raw_parts(view.ev_kind, n);
    let flags = std::slice::from_raw_parts(view.ev_flags, n);
    let table = std::slice::from_raw_parts(view.ev_table_id, n);
    let slot = std::slice::from_raw_parts(view.ev_schema_slot, n);
    let start = std::slice::from_raw_parts(view.ev_start_lsn, n);
    let commit = std::slice::from_raw_parts(view.ev_commit_lsn, n);
    let ord = std::slice::from_raw_parts(view.ev_tx_ordinal, n);
    let body = std::slice::from_raw_parts(view.ev_body_off, n);
    let slots = std::slice::from_raw_parts(view.slots, view.n_slots as usize);
    let mut out = Vec::with_capacity(n);
    for i in 0..n {
        let (start_lsn, commit_lsn, tx_ordinal) = (PgLsn::from(start[i]), PgLsn::from(commit[i]), ord[i]);
        let b = body[i] as usize;
        let ev = match kind[i] {
            ETLG_EV_BEGIN => Event::Begin(BeginEvent { start_lsn, commit_lsn, tx_ordinal, timestamp: a.u64_at(b) as i64, xid: table[i] }),
            ETLG_EV_COMMIT => Event::Commit(CommitEvent {
                start_lsn,
                commit_lsn,
                tx_ordinal,
                flags: flags[i] as i8,
                end_lsn: PgLsn::from(a.u64_at(b)),
                timestamp: a.u64_at(b + 8) as i64,
            }),
            ETLG_EV_RELATION => {
                let d = &slots[slot[i] as usize];
                Event::Relation(RelationEvent { start_lsn, commit_lsn, tx_ordinal, replicated_table_schema: schemas.schema_of(slot[i], d)? })
            }
            ETLG_EV_TRUNCATE => {
                let mut truncated_tables = Vec::with_capacity(table[i] as usize);
                for k in 0..table[i] as usize {
                    let s = a.u32_at(b + 8 * k + 4);
                    truncated_tables.push(schemas.schema_of(s, &slots[s as usize])?);
                }
                Event::Truncate(TruncateEvent { start_lsn, commit_lsn, tx_ordinal, options: flags[i] as i8, truncated_tables })
            }
            k @ (ETLG_EV_INSERT | ETLG_EV_UPDATE | ETLG_EV_DELETE) => {
                let d = &slots[slot[i] as usize];
                let replicated_table_schema = schemas.schema_of(slot[i], d)?;
                let old_kind = flags[i] & 3;
                let old_bytes = match old_kind {
                    ETLG_OLD_FULL => d.row_bytes_full,
                    ETLG_OLD_KEY => d.row_bytes_key,
                    _ => 0,
                } as usize;
                let old_table_row = match old_kind {
                    ETLG_OLD_FULL => Some(OldTableRow::Full(full(row(&a, d, b, false)?)?)),
                    ETLG_OLD_KEY => Some(OldTableRow::Key(full(row(&a, d, b, true)?)?)),
                    _ => None,
                };
                match k {
                    ETLG_EV_INSERT => {
                        Event::Insert(InsertEvent { start_lsn, commit_lsn, tx_ordinal, replicated_table_schema, table_row: full(row(&a, d, b, false)?)? })
                    }
                    ETLG_EV_UPDATE => {
                        let updated_table_row = match row(&a, d, b + old_bytes, false)? {
                            RowImage::Full(t) => UpdatedTableRow::Full(t),
                            RowImage::Partial(p) => UpdatedTableRow::Partial(p),
                        };
                        debug_assert_eq!(matches!(updated_table_row, UpdatedTableRow::Partial(_)), flags[i] & ETLG_FLAG_PARTIAL != 0);
                        Event::Update(UpdateEvent { start_lsn, commit_lsn, tx_ordinal, replicated_table_schema, updated_table_row, old_table_row })
                    }
                    _ => Event::Delete(DeleteEvent { start_lsn, commit_lsn, tx_ordinal, replicated_table_schema, old_table_row }),
                }
            }
            other => bail!(ErrorKind::InvalidState, "Unknown event kind in decoded arena", other),
        };
        out.push(ev);
    }
    Ok(out)
}
