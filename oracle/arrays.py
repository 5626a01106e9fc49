"""TEST INFRASTRUCTURE — array literals with var-len elements for the parity tests of the row formats: the elements of a text-like /
numeric / timetz / bytea array as the sinks see them. Never imported by the product path.

The literal is judged by the C++ oracle (parse_array_text, oracle_codec.hpp — crates/etl/src/postgres/codec/text.rs:228-312): a
malformed one raises NeedsHost here (the device hands such a cell back; the host raises the reference's error). The split below is
the same state machine restated for element BYTES (the C++ oracle's repr quotes strings ambiguously): a backslash takes the next
character literally, an unescaped quote toggles quoting and is dropped, an unquoted comma ends the element, an unquoted and
unescaped-to "null" of any case is NULL (text.rs:262-300)."""
import re

from oracle.rowbinary import NeedsHost

STRING_ARRAY_OIDS = {1009, 1015, 1014, 1002, 1003}      # text[] varchar[] bpchar[] "char"[] name[] (ArrayCell::String)
NUMERIC_A, BYTEA_A, TIMETZ_A = 1231, 1001, 1270
JSON_ARRAY_OIDS = {199, 3807}
VAR_ARRAY_OIDS = STRING_ARRAY_OIDS | {NUMERIC_A, BYTEA_A, TIMETZ_A} | JSON_ARRAY_OIDS
ELEM_MAX = 40                                           # kArrElemMax (columns.hip): a longer numeric / timetz element is handed back
JSON_ELEM_MAX = 256                                     # kJsonElemMax: a longer json[] element is handed back
E_JSON = 22                                             # etlg_err_code ETLG_E_JSON (checked against etl_amd.abi in tests/test_oracle_json_display.py)


class JsonDecodeError(Exception):
    """An element of a json[] literal that is not one JSON value: the reference's decode error (codec/text.rs:136-140), which the device
    raises for the batch — not a hand-back."""


def split_literal(type_oid, text):
    """[None | bytes] — the unescaped bytes of every element of a literal the reference accepts."""
    from oracle import oracle
    if isinstance(text, (bytearray, memoryview)):
        text = bytes(text)
    r = oracle.parse_text_cell(type_oid, text)
    if type_oid in JSON_ARRAY_OIDS and r == f"Err({E_JSON})":
        raise JsonDecodeError(r)
    if not r.startswith("Array["):
        raise NeedsHost(r)
    s = text
    if s.startswith(b"["):
        s = s[s.index(b"=") + 1:]
    body = s[1:-1]
    out, val, quoted, in_q, esc = [], bytearray(), False, False, False
    if not body:
        return out

    def close():
        out.append(None if (not quoted and bytes(val).lower() == b"null") else bytes(val))
    for c in body:
        ch = bytes([c])
        if esc:
            val += ch; esc = False
        elif ch == b'"':
            if not in_q:
                quoted = True
            in_q = not in_q
        elif ch == b"\\":
            esc = True
        elif ch == b"," and not in_q:
            close(); val = bytearray(); quoted = False
        else:
            val += ch
    close()
    return out


def display_list(type_oid, text):
    """numeric[] / timetz[] / bytea[]: the C++ oracle's parse (its repr) as what the sinks write — `n.to_string()` / `t.to_string()`
    (oracle/display.py), the decoded bytes; numeric elements come with their display scale (BigQuery's rule). [(value | None, scale | None)]"""
    from oracle import display as D
    from oracle import oracle
    r = oracle.parse_text_cell(type_oid, text)
    if not r.startswith("Array["):
        raise NeedsHost(r)
    out = []
    pat = r"NULL|Numeric\((NaN|\+Inf|-Inf|Infinity|-Infinity)\)|Numeric\(([+-]),w=(-?\d+),s=(\d+),\[([\d,]*)\]\)|TimeTz\((\d+):(\d+):(\d+)\.(\d+),(-?\d+)\)|Bytes\(([0-9a-f]*)\)"
    for m in re.finditer(pat, r[6:-1]):
        t = m.group(0)
        if t == "NULL":
            out.append((None, None))
        elif t.startswith("Numeric(") and m.group(1):
            out.append(({"NaN": "NaN", "+Inf": "Infinity", "Infinity": "Infinity", "-Inf": "-Infinity", "-Infinity": "-Infinity"}[m.group(1)].encode(), None))
        elif t.startswith("Numeric("):
            digits = tuple(int(x) for x in m.group(5).split(",")) if m.group(5) else ()
            out.append((D.numeric_string(0, 1 if m.group(2) == "-" else 0, int(m.group(3)), int(m.group(4)), digits).encode(), int(m.group(4))))
        elif t.startswith("TimeTz("):
            out.append((D.timetz_string(int(m.group(6)) * 3600 + int(m.group(7)) * 60 + int(m.group(8)), int(m.group(9)), int(m.group(10))).encode(), None))
        else:
            out.append((bytes.fromhex(m.group(11)), None))
    return out


def is_string_array(type_oid, text):
    """An array type without an arm of its own (money[], inet[], ranges ...): ArrayCell::String (codec/text.rs:216-226) — told by the
    C++ oracle's parse (only String / NULL elements; an empty array writes the same bytes under every class)."""
    from oracle import oracle
    r = oracle.parse_text_cell(type_oid, bytes(text))
    return r == "Array[]" or r.startswith("Array[String(") or (r.startswith("Array[NULL") and "String(" in r) or r.replace("NULL", "").strip("Array[],") == "" and r.startswith("Array[")


def elements(type_oid, text):
    """[(bytes | None, numeric scale | None)] of a var-len array cell, NeedsHost where the device hands the cell back."""
    if type_oid in STRING_ARRAY_OIDS or (type_oid not in VAR_ARRAY_OIDS and is_string_array(type_oid, text)):
        return [(e, None) for e in split_literal(type_oid, text)]
    if type_oid in JSON_ARRAY_OIDS:                  # `j.to_string()` per element (oracle/json_display.py), within json_display's limits
        from oracle import json_display as J
        els = split_literal(type_oid, text)
        if any(e is not None and (len(e) > JSON_ELEM_MAX or not J.device_limits_ok(e)) for e in els):
            raise NeedsHost("a json element beyond the device's limits")
        return [(None if e is None else J.display(e), None) for e in els]
    if type_oid in (NUMERIC_A, TIMETZ_A) and any(e is not None and len(e) > ELEM_MAX for e in split_literal(type_oid, text)):
        raise NeedsHost("an element of more than 40 characters")
    return display_list(type_oid, text)
