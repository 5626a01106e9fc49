"""TEST INFRASTRUCTURE — known answers the reference's own tests hold for the BigQuery row encoder, transcribed by hand with the
file:line they stand at (crates/etl-destinations/src/bigquery/encoding.rs, validation.rs). The reference compares against prost's
output (an un-vendored dependency); where a test pins a FIELD TYPE (int64 varint, packed repeated int64) the expected bytes below are
that field type under the protobuf wire format, written out literally here — not computed by oracle/protobuf.py, which these vectors pin.

Cells are in the form etl_amd.view.HostBatch.materialize() produces (what oracle/protobuf.py works on)."""
import datetime as dt


def _days_ce(y, m, d):
    return dt.date(y, m, d).toordinal()


# encoding.rs:451-480  timestamptz_values_encode_as_epoch_microseconds: Utc 2026-01-02 03:04:05 under tag 1
#   scalar:  prost::encoding::int64::encode(1, &micros)          -> key 0x08 (field 1, wire type 0) | varint(micros)
#   array :  prost::encoding::int64::encode_packed(1, &[micros]) -> key 0x0A (field 1, wire type 2) | varint(len) | varint(micros)
# micros = 1 767 323 045 000 000 = 0x6475EF64CF340; its varint, seven bits at a time, low group first:
TSTZ_CELL = ("TimestampTz", _days_ce(2026, 1, 2), 3 * 3600 + 4 * 60 + 5, 0)
TSTZ_MICROS = 1767323045000000
TSTZ_VARINT = bytes([0xC0, 0xE6, 0xB3, 0xB2, 0xEF, 0xEB, 0x91, 0x03])
TSTZ_SCALAR_BYTES = bytes([0x08]) + TSTZ_VARINT
TSTZ_PACKED_BYTES = bytes([0x0A, len(TSTZ_VARINT)]) + TSTZ_VARINT

# encoding.rs:483-496 (scalar) and :385-404 (array: "Cell at index 0", "Element at index 1"), validation.rs:213-229 (scale 38 passes, 39 fails)
NUMERIC_OVER_SCALE = "0.000000000000000000000000000000000000001"      # 39 decimal places: UnsupportedValueInDestination, "would be rounded by BigQuery"
NUMERIC_AT_SCALE = "0.00000000000000000000000000000000000001"         # 38: accepted
NUMERIC_ARRAY_ROUNDING = ["123.456", NUMERIC_OVER_SCALE, "789.012"]   # the element at index 1 fails

# encoding.rs:343-360, validation.rs:44-93: a JSON integer literal outside i64 / u64 is refused; anything with '.', 'e', 'E' is BigQuery's to judge
JSON_ACCEPTED = ['{"value":1e309}', '{"value":18446744073709551615}', '{"value":-9223372036854775808}', '[1.0,{"a":[2,3e400]}]', '"18446744073709551616"']
JSON_REFUSED = ['{"value":18446744073709551616}', '{"value":-9223372036854775809}', '[1,{"deep":[{"n":99999999999999999999}]}]']

# encoding.rs:372-383: an int4[] with a NULL element -> NullValuesNotSupportedInArrayInDestination, detail "Cell at index 0 failed validation"
ARRAY_WITH_NULLS = [1, None, 3]
# encoding.rs:406-418: {1,2,3} between two strings is accepted
ARRAY_VALID = [1, 2, 3]
