"""Known answers of the reference's Arrow builders, transcribed from its in-file tests
(crates/etl-destinations/src/iceberg/encoding.rs, `mod tests`): build_*_array :1142-1396, rows_to_record_batch_* :1399-1560,
build_*_list_array :1677-2050. The reference builds its rows from `Cell` literals; here every cell is the Postgres text that
decodes to that literal (the decode is the oracle's, pinned by the codec KATs), so one table per case goes through
decode -> hand-off and must come out as the listed Arrow values. Cases the decode path cannot produce (a cell of the wrong
variant for its column: "Non-bool cell becomes null") are not transcribable and are left out; `Uuid::new_v4()` is a fixed uuid.

Each case: (reference lines, [(column name, type oid, nullable)], rows of texts (None = NULL), expected {column: (arrow type, values)}).
Arrow type names: the reference's DataType, mapped to pyarrow by tests/test_arrow_kats.py. Timestamps are microseconds."""

NULL = None
UUID_TEXT = "123e4567-e89b-12d3-a456-426614174000"
UUID_BYTES = bytes.fromhex(UUID_TEXT.replace("-", ""))
DAYS_2023_05_15 = 19492            # NaiveDate(2023, 5, 15) - 1970-01-01
MICROS_12_30_45 = 45_045_000_000   # NaiveTime(12, 30, 45) since midnight
MICROS_1E9 = 1_000_000_000_000_000  # DateTime::from_timestamp(1000000000, 0) = 2001-09-09 01:46:40 UTC

SCALAR_KATS = [
    ("1142-1158 build_boolean_array", [("v", 16, True)], [["t"], ["f"], [NULL]], {"v": ("Boolean", [True, False, None])}),
    ("1161-1178 build_i32_array", [("a", 21, True), ("b", 23, True)], [["42", "-123"], [NULL, NULL]],
     {"a": ("Int32", [42, None]), "b": ("Int32", [-123, None])}),                                  # I16 widens (cell_to_i32)
    ("1181-1200 build_i64_array", [("a", 20, True), ("b", 26, True)], [["123456789", "456"], ["-987654321", "4294967295"], [NULL, NULL]],
     {"a": ("Int64", [123456789, -987654321, None]), "b": ("Int64", [456, 4294967295, None])}),   # U32 widens (cell_to_i64; u32::MAX :1012)
    ("1203-1220 build_f32_array", [("v", 700, True)], [["2.5"], ["-1.25"], [NULL]], {"v": ("Float32", [2.5, -1.25, None])}),
    ("1223-1240 build_f64_array", [("v", 701, True)], [["1.23456789"], ["-9.87654321"], [NULL]], {"v": ("Float64", [1.23456789, -9.87654321, None])}),
    ("1243-1260 build_string_array", [("v", 25, True)], [["hello"], [NULL]], {"v": ("Utf8", ["hello", None])}),
    ("1263-1283 build_binary_array", [("v", 17, True)], [["\\x0102030405"], ["\\x"], [NULL]], {"v": ("LargeBinary", [b"\x01\x02\x03\x04\x05", b"", None])}),
    ("1286-1308 build_date32_array", [("v", 1082, True)], [["2023-05-15"], ["1970-01-01"], [NULL]], {"v": ("Date32", [DAYS_2023_05_15, 0, None])}),
    ("1311-1334 build_time64_array", [("v", 1083, True)], [["12:30:45"], ["00:00:00"], [NULL]], {"v": ("Time64(us)", [MICROS_12_30_45, 0, None])}),
    ("1337-1358 build_timestamp_array", [("v", 1114, True)], [["2001-09-09 01:46:40"], [NULL]], {"v": ("Timestamp(us)", [MICROS_1E9, None])}),
    ("1361-1386 build_timestamptz_array", [("v", 1184, True)], [["2001-09-09 01:46:40+00"], [NULL]], {"v": ("Timestamp(us,UTC)", [MICROS_1E9, None])}),
    ("1389-1396 build_uuid_array", [("v", 2950, True)], [[UUID_TEXT], [NULL]], {"v": ("FixedSizeBinary(16)", [UUID_BYTES, None])}),
    ("1399-1436 rows_to_record_batch_simple", [("id", 23, False), ("name", 25, False), ("active", 16, False)],
     [["42", "hello", "t"], ["100", "world", "f"]],
     {"id": ("Int32", [42, 100]), "name": ("Utf8", ["hello", "world"]), "active": ("Boolean", [True, False])}),
    ("1439-1466 rows_to_record_batch_with_nulls", [("id", 23, True), ("name", 25, True)], [["42", NULL], [NULL, "test"]],
     {"id": ("Int32", [42, None]), "name": ("Utf8", [None, "test"])}),
    ("1469-1530 rows_to_record_batch_temporal_types", [("date_col", 1082, False), ("time_col", 1083, False), ("ts_col", 1114, False), ("ts_tz_col", 1184, False)],
     [["2023-05-15", "12:30:45", "2001-09-09 01:46:40", "2001-09-09 01:46:40+00"]],
     {"date_col": ("Date32", [DAYS_2023_05_15]), "time_col": ("Time64(us)", [MICROS_12_30_45]), "ts_col": ("Timestamp(us)", [MICROS_1E9]),
      "ts_tz_col": ("Timestamp(us,UTC)", [MICROS_1E9])}),
    ("1533-1560 rows_to_record_batch_binary_and_uuid", [("data", 17, False), ("uuid", 2950, False)], [["\\x0102030405", UUID_TEXT]],
     {"data": ("LargeBinary", [b"\x01\x02\x03\x04\x05"]), "uuid": ("FixedSizeBinary(16)", [UUID_BYTES])}),
    ("1563-1578 rows_to_record_batch_empty", [("id", 23, False), ("name", 25, False)], [], {"id": ("Int32", []), "name": ("Utf8", [])}),
    # cell_to_string (:349-352): numeric / timetz columns are Utf8 of `to_string()` (numeric list elements "12345", "-6789" at :1947-1983)
    ("349-352 + 1947-1983 numeric / timetz as strings", [("n", 1700, True), ("tz", 1266, True)], [["12345", "12:30:00.123+02"], ["-6789", "12:30:00-07:30"], [NULL, NULL]],
     {"n": ("Utf8", ["12345", "-6789", None]), "tz": ("Utf8", ["12:30:00.123+02", "12:30:00-07:30", None])}),
]

# build_*_list_array: (lines, array type oid, element arrow type, rows of literals, expected lists)
LIST_KATS = [
    ("1677-1722 build_boolean_list_array_fn", 1000, "Boolean", ["{t,f,NULL}", "{t}", "{}", NULL], [[True, False, None], [True], [], None]),
    ("1746-1788 build_int32_list_array_fn (int2)", 1005, "Int32", ["{10,20,NULL}", "{}", NULL], [[10, 20, None], [], None]),
    ("1746-1788 build_int32_list_array_fn (int4)", 1007, "Int32", ["{100,-200}", "{}", NULL], [[100, -200], [], None]),
    ("1791-1837 build_int64_list_array_fn (int8)", 1016, "Int64", ["{123456789,-987654321,NULL}", "{}", NULL], [[123456789, -987654321, None], [], None]),
    ("1791-1837 build_int64_list_array_fn (oid)", 1028, "Int64", ["{456,789}", "{}", NULL], [[456, 789], [], None]),
    ("1840-1882 build_float32_list_array_fn", 1021, "Float32", ["{1.5,-2.75,NULL}", "{3.1415927}", "{}", NULL], [[1.5, -2.75, None], [3.1415927410125732], [], None]),
    ("1885-1931 build_float64_list_array_fn", 1022, "Float64", ["{1.23456789,-9.87654321,NULL}", "{3.141592653589793}", "{}", NULL],
     [[1.23456789, -9.87654321, None], [3.141592653589793], [], None]),
    ("1934-2001 build_string_list_array_fn (text)", 1009, "Utf8", ["{hello,world,NULL}", "{}", NULL], [["hello", "world", None], [], None]),
]
