"""json cells as the reference's sinks write them (`Cell::Json(j) => j.to_string()`), from the reference's own tests. serde_json is a
crates.io dependency that is not under /root/reference, so these are the vectors that pin oracle/json_display.py; transcribed by hand
(the reference is Rust; nothing here was generated)."""

# (source text, Display) — crates/etl/src/postgres/codec/text.rs:810-818 (try_from_str_json_accepts_wide_number_literals)
PINNED = [
    ('{"value":1e309}', '{"value":1e+309}'),
    # crates/etl-destinations/src/iceberg/encoding.rs:1587 + :1611 (json!({"key": "value", "number": 123}) through cell_to_string)
    ('{"key": "value", "number": 123}', '{"key":"value","number":123}'),
    # iceberg/encoding.rs:1952-1990, :2391-2491, :2601 (json!({"key": "value"}))
    ('{"key": "value"}', '{"key":"value"}'),
    # ducklake/encoding.rs:858 (json!({"id": 1}))
    ('{"id": 1}', '{"id":1}'),
]

# restated from serde_json 1.0.149's published source / documentation (see oracle/json_display.py's header): NOT pinned by a reference test
RESTATED = [
    ('{"b":1,"a":2}', '{"a":2,"b":1}'),                         # BTreeMap order
    ('{"k":"v","k":"dup"}', '{"k":"dup"}'),                     # the last of repeated keys
    ('{"b":1,"a":2,"b":3,"a":4}', '{"a":4,"b":3}'),
    ('{"aa":1,"b":2,"a":3}', '{"a":3,"aa":1,"b":2}'),           # bytes, not (length, bytes) as jsonb orders them
    ('{"\\u00e9":1,"z":2,"é":3}', '{"z":2,"é":3}'),             # keys compare decoded: "é" IS "é"
    (' [ 1 , 2.50 , -0 , -0.0 , 1E5 , 1e-7 , 1.5e+3 , 123456789012345678901234567890 ] ', '[1,2.50,0,-0.0,1E+5,1e-7,1.5e+3,123456789012345678901234567890]'),
    ('"a\\"b\\\\c\\/d\\b\\f\\n\\r\\t\\u0001\\u001f\\u007f"', '"a\\"b\\\\c/d\\b\\f\\n\\r\\t\\u0001\\u001f\x7f"'),
    ('"\\u00e9\\uD83D\\uDE00"', '"é😀"'),
    ('"é中😀"', '"é中😀"'),
    ("null", "null"), (" true ", "true"), ("false", "false"), ('""', '""'), ("[]", "[]"), ("{}", "{}"), ("[ ]", "[]"), ("{ }", "{}"),
    ('{"a":{"c":[null,true,false],"b":{}}}', '{"a":{"b":{},"c":[null,true,false]}}'),
    ('[{"z":[{"y":1,"x":[]}],"a":"s"}]', '[{"a":"s","z":[{"x":[],"y":1}]}]'),
]
