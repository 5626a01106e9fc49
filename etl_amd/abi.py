"""ctypes mirror of include/etlg.h (structs, enums, error table).

Pure declarations: no library is loaded here. `etl_amd.native` binds them to
libetl_gfx950.so; the test-only oracle wrapper binds the same view struct to
oracle/liboracle.so so both sides can be compared field by field.
"""
import ctypes as C

ABI_VERSION = 1

# etlg_error_kind
OK = 0
ConversionError = 1
InvalidData = 2
ValidationError = 3
InvalidState = 4
MissingTableSchema = 5
CorruptedTableSchema = 6
DeserializationError = 7
SourceConnectionFailed = 8
IoError = 9
UnsupportedValueInDestination = 10
NullValuesNotSupportedInArrayInDestination = 11
InvalidArgument = 100
DeviceError = 101
Unsupported = 102

KIND_NAMES = {
    OK: "Ok", ConversionError: "ConversionError", InvalidData: "InvalidData",
    ValidationError: "ValidationError", InvalidState: "InvalidState",
    MissingTableSchema: "MissingTableSchema", CorruptedTableSchema: "CorruptedTableSchema",
    DeserializationError: "DeserializationError", SourceConnectionFailed: "SourceConnectionFailed",
    IoError: "IoError", UnsupportedValueInDestination: "UnsupportedValueInDestination", NullValuesNotSupportedInArrayInDestination: "NullValuesNotSupportedInArrayInDestination", InvalidArgument: "InvalidArgument", DeviceError: "DeviceError",
    Unsupported: "Unsupported",
}

# etlg_err_code
(E_NONE, E_WIRE, E_TXN_STATE, E_COMMIT_LSN, E_MISSING_SHARED_STATE, E_WAITING_RELATION,
 E_TUPLE_WIDTH, E_FULL_ROW_MISSING, E_REQUIRED_NULL, E_BINARY_FORMAT, E_UTF8, E_OLD_ROW_WIDTH,
 E_KEY_SHAPE, E_KEY_MISSING_COLS, E_KEY_MISSING_VALUE, E_BOOL, E_INT, E_FLOAT, E_NUMERIC,
 E_BYTEA, E_DATETIME, E_UUID, E_JSON, E_ARRAY_SHORT, E_ARRAY_BRACES, E_ARRAY_DIMS,
 E_ARRAY_MULTIDIM, E_ARRAY_QUOTE, E_ARRAY_ESCAPE, E_SCHEMA_NOT_FOUND, E_UNKNOWN_COLUMNS,
 E_DDL_PARSE, E_IO, E_BOOTSTRAP_SNAPSHOT, E_SNAPSHOT_MISMATCH, E_CTRL_HINT, E_COPY_UNTERMINATED,
 E_COPY_MORE_COLS, E_COPY_FEWER_COLS, E__COUNT) = range(40)

# etlg_type_class
(TC_STRING, TC_BOOL, TC_I16, TC_I32, TC_I64, TC_U32, TC_F32, TC_F64, TC_NUMERIC, TC_BYTEA,
 TC_DATE, TC_TIME, TC_TIMETZ, TC_TIMESTAMP, TC_TIMESTAMPTZ, TC_UUID, TC_JSON, TC_ARRAY) = range(18)

TS_ABSENT, TS_READY, TS_SYNC_DONE, TS_OTHER = range(4)
WORKER_APPLY, WORKER_TABLE_SYNC = 0, 1

F_INPUT_ON_DEVICE = 1 << 0
F_OUTPUT_ON_DEVICE = 1 << 1
F_NO_CONTROL = 1 << 2
F_ASYNC = 1 << 3
F_FINISH_CELLS = 1 << 4
FINISH_ARRAYS, FINISH_FLOATS = 1, 2

OLD_NONE, OLD_FULL, OLD_KEY, FLAG_PARTIAL = 0, 1, 2, 4
CELL_VALUE, CELL_NULL, CELL_MISSING, CELL_DEFERRED = range(4)
NUM_VALUE, NUM_NAN, NUM_PINF, NUM_NINF = range(4)


class Col(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type_oid", C.c_uint32), ("type_modifier", C.c_int32),
                ("attnum", C.c_int32), ("nullable", C.c_uint8), ("primary_key", C.c_uint8),
                ("_pad", C.c_uint8 * 2)]


class Error(C.Structure):
    _fields_ = [("kind", C.c_int32), ("code", C.c_int32), ("description", C.c_char_p),
                ("detail", C.c_char_p), ("frame_index", C.c_int64)]


class ErrDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("description", C.c_char_p)]


class SlotCol(C.Structure):
    _fields_ = [("type_oid", C.c_uint32), ("stored_index", C.c_uint16), ("type_class", C.c_uint8),
                ("nullable", C.c_uint8), ("identity", C.c_uint8), ("_pad", C.c_uint8),
                ("off_full", C.c_uint16), ("off_key", C.c_uint16), ("key_index", C.c_uint16)]


class SlotDesc(C.Structure):
    _fields_ = [("table_id", C.c_uint32), ("n_stored", C.c_uint32), ("snapshot_lsn", C.c_uint64),
                ("n_cols", C.c_uint32), ("n_ident", C.c_uint32), ("row_bytes_full", C.c_uint32),
                ("row_bytes_key", C.c_uint32), ("state_bytes_full", C.c_uint32),
                ("state_bytes_key", C.c_uint32), ("cols", C.POINTER(SlotCol))]


class BatchView(C.Structure):
    _fields_ = [("n_events", C.c_uint64), ("n_frames", C.c_uint64), ("fixed_bytes", C.c_uint64),
                ("heap_bytes", C.c_uint64), ("payload_bytes", C.c_uint64 * 3),
                ("ev_kind", C.c_void_p), ("ev_flags", C.c_void_p), ("ev_table_id", C.c_void_p),
                ("ev_schema_slot", C.c_void_p), ("ev_start_lsn", C.c_void_p),
                ("ev_commit_lsn", C.c_void_p), ("ev_tx_ordinal", C.c_void_p),
                ("ev_body_off", C.c_void_p), ("fixed", C.c_void_p), ("heap", C.c_void_p),
                ("on_device", C.c_uint32), ("n_slots", C.c_uint32), ("slots", C.POINTER(SlotDesc))]


class Column(C.Structure):   # etlg_column
    _fields_ = [("type_class", C.c_uint32), ("arrow_kind", C.c_uint32), ("value_bytes", C.c_uint32), ("nullable", C.c_uint32),
                ("null_count", C.c_uint64), ("deferred_count", C.c_uint64), ("validity", C.c_void_p), ("deferred", C.c_void_p),
                ("values", C.c_void_p), ("offsets", C.c_void_p), ("values_bytes", C.c_uint64),
                ("child_kind", C.c_uint32), ("_pad", C.c_uint32), ("child_count", C.c_uint64), ("child_null_count", C.c_uint64),
                ("child_validity", C.c_void_p), ("child_offsets", C.c_void_p)]


class ColumnsView(C.Structure):   # etlg_columns_view
    _fields_ = [("n_rows", C.c_uint64), ("n_cols", C.c_uint32), ("on_device", C.c_uint32), ("cols", C.POINTER(Column)),
                ("row_event", C.c_void_p)]


class RowBinaryView(C.Structure):   # etlg_rowbinary_view
    _fields_ = [("n_rows", C.c_uint64), ("n_bytes", C.c_uint64), ("n_host_rows", C.c_uint64), ("status", C.c_uint32),
                ("on_device", C.c_uint32), ("host_event", C.c_uint64), ("host_column", C.c_uint32), ("_pad", C.c_uint32),
                ("bytes", C.c_void_p), ("row_offsets", C.c_void_p), ("row_event", C.c_void_p)]


class SizeModel(C.Structure):   # etlg_size_model
    _fields_ = [(n, C.c_uint32) for n in ("begin_event", "commit_event", "insert_event", "update_event", "delete_event", "truncate_event",
                                          "relation_event", "replicated_table_schema", "table_row", "cell")] + [("_reserved", C.c_uint32 * 2)]


class FinishStats(C.Structure):   # etlg_finish_stats
    _fields_ = [(n, C.c_uint64) for n in ("deferred_seen", "arrays_typed", "floats_settled", "left_deferred", "heap_bytes_added")]


SIZE_HINT_INCOMPLETE = 1 << 63
CH_MERGE_TREE, CH_REPLACING_MERGE_TREE = 0, 1
RB_OK, RB_NEEDS_HOST = 0, 3
(AK_BOOLEAN, AK_INT32, AK_INT64, AK_FLOAT32, AK_FLOAT64, AK_DATE32, AK_TIME64_US, AK_TIMESTAMP_US, AK_TIMESTAMP_US_UTC, AK_FIXED16,
 AK_LARGE_UTF8, AK_LARGE_BINARY, AK_TEXT_FORM, AK_LIST) = range(14)
AK_NONE = 255
ROWS_INSERT, ROWS_UPDATE, ROWS_PARSE_ARRAYS, ROWS_FORMAT_JSON = 1, 2, 4, 8


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_uint64), ("total_ms", C.c_double)]


def make_cols(cols):
    """cols: iterable of (name, type_oid, nullable[, primary_key[, typmod]]) in attnum order."""
    arr = (Col * len(cols))()
    keep = []
    for i, c in enumerate(cols):
        name, oid, nullable = c[0], c[1], c[2]
        pk = c[3] if len(c) > 3 else 0
        typmod = c[4] if len(c) > 4 else -1
        b = name.encode() if isinstance(name, str) else name
        keep.append(b)
        arr[i].name = b
        arr[i].type_oid = oid
        arr[i].type_modifier = typmod
        arr[i].attnum = i + 1
        arr[i].nullable = 1 if nullable else 0
        arr[i].primary_key = 1 if pk else 0
    arr._keep = keep
    return arr


class _DevArray:
    """Raw device memory as a __cuda_array_interface__ object (what torch.as_tensor needs to wrap a pointer)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, nbytes, device):
    """A uint8 torch tensor over `nbytes` of device memory at `ptr` (no copy; the owner must outlive it)."""
    import torch
    if not nbytes:
        return torch.zeros(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(_DevArray(ptr, nbytes), device=device)
