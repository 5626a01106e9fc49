"""TEST INFRASTRUCTURE — CPU restatement of what the reference's sinks write for a json / jsonb cell: serde_json's
`from_str::<Value>` (crates/etl/src/postgres/codec/text.rs:126-134) followed by `Value::to_string()`
(clickhouse/encoding.rs:73, bigquery/encoding.rs:173-176, iceberg/encoding.rs:356, ducklake/encoding.rs:173). For the parity tests of
json_display (etl_amd/csrc/columns.hip). Never imported by the product path.

serde_json 1.0.149 (Cargo.lock) is a crates.io dependency that is NOT under /root/reference; the reference builds it with the features
`arbitrary_precision` + `std` and without `preserve_order` (crates/etl/Cargo.toml:36, crates/etl-destinations/Cargo.toml:121). Its
published behaviour, restated:
  * `Value::Object` is a BTreeMap<String, Value>: members leave sorted by the bytes of their decoded keys; inserting a key again
    replaces the value (the last of repeated keys wins);
  * the compact formatter: no whitespace, `,` and `:` bare; strings escaped with \\" \\\\ \\b \\f \\n \\r \\t and \\u00xx (lowercase hex) for
    the other bytes below 0x20, everything else as it is (non-ASCII raw, 0x7f raw, `/` bare);
  * arbitrary_precision keeps a number as the text the scanner collected. The scanner (de.rs scan_integer / scan_decimal /
    scan_exponent) copies the literal, and writes an exponent's sign always: "1e309" -> "1e+309". An integer literal is first tried
    as u64 (i64 when negative) — parse_any_number's `buf.parse()` — and then printed from the integer: only "-0" -> "0" changes.

PINNED by the reference's own tests (tests/golden/json_display_kats.py): `{"value":1e309}` -> `{"value":1e+309}`
(codec/text.rs:812-815); `json!({"key":"value","number":123}).to_string()` == `{"key":"value","number":123}`
(iceberg/encoding.rs:1587-1611, :1990). UNPINNED (serde_json's source is not here to run): the "-0" rule, `E` kept upper-case, the
escape table, key order by decoded bytes — restated from the crate's published source and documentation."""
import json
import re


class Lit(str):
    """A number literal as arbitrary_precision keeps it."""


def _number(lit):
    if re.fullmatch(r"-?[0-9]+", lit):
        v = int(lit)
        if (lit.startswith("-") and -(1 << 63) <= v) or (not lit.startswith("-") and v < (1 << 64)):
            return Lit(str(v))           # through u64 / i64 and back: "-0" -> "0"
        return Lit(lit)
    return Lit(re.sub(r"([eE])(?![+-])", r"\1+", lit))


def _constant(name):
    raise ValueError(f"not JSON: {name}")


def parse(text):
    """The serde_json::Value of a VALID json text: dicts (sorted when written), lists, str, Lit, True / False / None."""
    if isinstance(text, (bytes, bytearray, memoryview)):
        text = bytes(text).decode("utf-8")
    return json.loads(text, object_pairs_hook=dict, parse_int=_number, parse_float=_number, parse_constant=_constant)


_ESC = {0x22: '\\"', 0x5C: "\\\\", 0x08: "\\b", 0x0C: "\\f", 0x0A: "\\n", 0x0D: "\\r", 0x09: "\\t"}


def _string(s):
    return '"' + "".join(_ESC.get(ord(ch)) or (f"\\u{ord(ch):04x}" if ord(ch) < 0x20 else ch) for ch in s) + '"'


def write(v):
    if v is None:
        return "null"
    if v is True:
        return "true"
    if v is False:
        return "false"
    if isinstance(v, Lit):
        return str(v)
    if isinstance(v, str):
        return _string(v)
    if isinstance(v, list):
        return "[" + ",".join(write(x) for x in v) + "]"
    if isinstance(v, dict):
        return "{" + ",".join(_string(k) + ":" + write(x) for k, x in sorted(v.items(), key=lambda kv: kv[0].encode("utf-8", "surrogatepass"))) + "}"
    raise TypeError(type(v))


def display(text):
    """`serde_json::from_str::<Value>(text).unwrap().to_string()` as UTF-8 bytes."""
    return write(parse(text)).encode("utf-8", "surrogatepass")


def numbers(v):
    """Every number literal of the parsed value (validate_json_for_bigquery walks the parsed tree, bigquery/validation.rs:47-62)."""
    if isinstance(v, Lit):
        yield str(v)
    elif isinstance(v, list):
        for x in v:
            yield from numbers(x)
    elif isinstance(v, dict):
        for x in v.values():
            yield from numbers(x)


def device_limits_ok(text, max_depth=16, max_members=64):
    """Whether json_display takes the cell: containers nested at most `max_depth` deep, objects of at most `max_members` members in
    the SOURCE text (repeated keys count), no (decoded) key starting with serde_json's private token."""
    if isinstance(text, (bytes, bytearray, memoryview)):
        text = bytes(text).decode("utf-8")
    ok = True

    def pairs(ps):
        nonlocal ok
        if len(ps) > max_members or any(k.startswith("$serde_json::private::") for k, _ in ps):
            ok = False
        return ("obj", [v for _, v in ps])

    def depth(v):
        if isinstance(v, tuple):
            return 1 + max([depth(x) for x in v[1]], default=0)
        if isinstance(v, list):
            return 1 + max([depth(x) for x in v], default=0)
        return 0
    v = json.loads(text, object_pairs_hook=pairs, parse_int=str, parse_float=str)
    return ok and depth(v) <= max_depth
