"""Differential fuzz of single-cell value codecs, emulated (or real) kernels against the oracle: python tools/cell_fuzz.py [batches] [seed]
For every class a batch of Insert frames with one mutated text each is decoded by both; the first error (code, frame) and the arena of the
rows before it must agree, and — for the classes whose hand-off writes Display strings — the RowBinary bytes as well. Exemplars are
mutated character-wise (digits, separators, signs, zone suffixes, exponent forms, brackets, escapes), so most texts sit near the
edges of the grammars (chrono's parse_from_str shapes, serde_json's validity, numeric's and float's forms, array literals)."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi
from etl_amd.decoder import Decoder
from oracle import oracle
from oracle import rowbinary as RB
from tests import pgwire as W
from tests import scenarios as SC

CLASSES = {
    "timestamptz": (1184, ["2024-02-29 12:30:45.123456+02", "1999-12-31 23:59:60+00", "0001-01-01 00:00:00+00:00:00", "2024-1-2 3:4:5+5", "2024-01-02T03:04:05Z",
                           "2024-01-02 03:04:05.1234567-07:30", "2024-01-02 03:04:05 BC", "infinity", "-infinity", "294276-12-31 23:59:59.999999+00"],
                    "0123456789-+:. TZBCinfty"),
    "timestamp": (1114, ["2024-02-29 12:30:45.123456", "1999-12-31 23:59:60", "2024-1-2 3:4:5", "2024-01-02T03:04:05", "0001-01-01 00:00:00.5", "infinity"], "0123456789-+:. TBCinf"),
    "date": (1082, ["2024-02-29", "0001-01-01", "9999-12-31", "2023-02-29", "2024-1-2", "10000-01-01", "infinity", "0044-03-15 BC"], "0123456789- BCinf"),
    "time": (1083, ["12:30:45.123456", "23:59:60", "00:00:00", "24:00:00", "1:2:3", "12:30", "12:30:45.1234567890"], "0123456789:. "),
    "timetz": (1266, ["12:30:45.123456+02", "23:59:59-07:30", "00:00:00+15:59:59", "12:30:45+0530", "12:30:45Z", "24:00:00-00"], "0123456789:.+- Z"),
    "jsonb": (3802, ['{"a":1,"b":[true,false,null],"c":{"d":"e\\u00e9\\n"}}', '[1,2.5e-3,-0,"x"]', '"\\ud83d\\ude00"', '{"a":{"a":{"a":[[[[1]]]]}}}', "1e999", '{"k":"v"}   ', "[]", "nul"],
              '{}[]":,\\u0123456789abcdefADtrenlsu-+.E '),
    "numeric": (1700, ["0", "-12.5", "123456789.000100", "NaN", "1e5", "0.000012", "1234567890123456789012345678901234567890.12345", "-0.00", "1e-40", "Infinity", "+17", " 42.50 ", "1_000.5", ".5", "5."],
                "0123456789.+-eE_ NaInfity"),
    "float8": (701, ["1.5", "-0.25", "1e300", "3.141592653589793", "12345678901234567890123", "nan", "Infinity", "1e-400", "0x10", ".5e1", "9007199254740993"], "0123456789.+-eEnaifty x"),
    "int4[]": (1007, ["{1,2,3}", "{}", "{NULL,-5}", "{{1,2},{3,4}}", '{"1",2}', "[0:2]={1,2,3}", "{ 1 , 2 }"], "0123456789{},\"NUL -[]:= "),
    "text[]": (1009, ['{a,b,"c d"}', '{"x\\"y",NULL,"NULL"}', "{}", '{"\\\\"}', "{a b,c}", '{"é",中}'], '{}",\\NULabc é'),
    "float8[]": (1022, ["{1.5,-0.25,NULL}", "{1e300,nan}", "{}", '{"3.14"}', "{Infinity,-inf}", "{12345678901234567890123}"], '0123456789{},".NULnaifty+-eE '),
    "date[]": (1182, ["{2024-02-29,NULL}", "{0001-01-01}", "{1899-12-31}", "{2300-01-01,2024-1-2}", "{}", '{"2024-01-02"}', "{infinity}"], '0123456789{},"-NULinfty BC'),
    "timestamptz[]": (1185, ['{"2024-01-02 03:04:05+00",NULL}', '{"2024-01-02 03:04:05.123456+05:30"}', "{}", '{"1999-12-31 23:59:60+00"}'], '0123456789{},"-+:. NULZT'),
    "time[]": (1183, ["{12:30:45.123456,NULL}", "{23:59:60}", "{}", "{1:2:3}"], '0123456789{},":.NUL '),
    "uuid[]": (2951, ["{123e4567-e89b-12d3-a456-426614174000,NULL}", "{}", "{123e4567e89b12d3a456426614174000}"], '0123456789abcdefABCDEF{},"-NUL'),
    "bytea": (17, ["\\x0102ff", "\\x", "\\xABcd", "\\x0", "abc", "\\\\000\\\\001", "\\xzz"], "\\x0123456789abcdefABzZ"),
    "uuid": (2950, ["123e4567-e89b-12d3-a456-426614174000", "{123e4567-e89b-12d3-a456-426614174000}", "123e4567e89b12d3a456426614174000", "urn:uuid:123e4567-e89b-12d3-a456-426614174000", "123E4567-E89B-12D3-A456-42661417400"],
             "0123456789abcdefABCDEF-{}urn:id"),
}


def mutate(rng, s, alphabet):
    s = list(s)
    for _ in range(rng.choice([0, 1, 1, 1, 2, 3])):
        k = rng.random()
        pos = rng.randrange(len(s) + 1)
        if k < 0.4 and s:
            s[min(pos, len(s) - 1)] = rng.choice(alphabet)
        elif k < 0.7:
            s.insert(pos, rng.choice(alphabet))
        elif k < 0.9 and s:
            del s[min(pos, len(s) - 1)]
        elif s:
            a = min(pos, len(s) - 1)
            s[a:a] = s[a:a + rng.randrange(1, 4)]
    return "".join(s)


ARRAYS = {1007: ["{1,NULL,3}", "{}", "[1:2]={1,2}", '{"1",2}', "{+5,-0}", "{-2147483648,2147483647}", '{"\\1",null,NuLl}', "[-1:0]={7,8}", "{{1,2},{3,4}}", "{ 1 , 2 }"],
          1016: ["{9223372036854775807,-9223372036854775808}", "{}", "{NULL}", '{"12"}'], 1005: ["{-32768,32767}", "{1,2,3}", "{}"],
          1000: ["{t,f,NULL}", "{}", '{"t"}', "{true}"], 1028: ["{0,4294967295}", "{NULL,7}", "{}"]}


VAR_ARRAYS = {1231: ["{0,-12.5,NULL}", "{123456789.000100,NaN}", "{1e5,0.000012}", "{}", '{"1.5"}', "{Infinity,-inf}", "{1_000.5, 42 }"],
              1001: ['{"\\\\x0102ff",NULL}', '{"\\\\x"}', "{}", '{"\\\\xABcd","\\\\x00"}', "{\\\\x41}"],
              1270: ["{12:30:45.123456+02,NULL}", "{23:59:59-07:30}", "{}", '{"00:00:00+15:59:59"}'],
              1009: ['{a,"b c",NULL,"null",nUlL}', '{"x\\"y","a,b","{}"}', '{é,"\\\\"}', "{ a , b }", '{"",x}', "{abcd,abcde,nulls,null}", "[0:1]={x,y}", "{}"]}


def oracle_list(oid, text):
    r = oracle.parse_text_cell(oid, text)
    if not r.startswith("Array["):
        return r
    if oid == 1009:          # text[]: the repr is String("...") | NULL; unambiguous as long as no text holds `")` (the alphabet has no parenthesis)
        import re
        return [None if m.group(0) == "NULL" else m.group(1) for m in re.finditer(r'NULL(?=,|\])|String\("(.*?)"\)(?=,|\]$)', r[6:], flags=re.S)]
    if oid in VAR_ARRAYS:   # numeric[] / bytea[] / timetz[]: what the sinks hand to Arrow (Display strings, decoded bytes)
        from tests.test_gpu_columns import _oracle_display_list
        return _oracle_display_list(oid, text)
    body = r[6:-1]
    out = []
    for e in ([] if not body else body.split(",")):
        out.append(None if e == "NULL" else (e == "Bool(true)") if e.startswith("Bool(") else int(e[e.index("(") + 1:-1]))
    return out


def fuzz_arrays(rng, batches):
    """Array literals parsed on the device into list columns (etlg_batch_columns with ROWS_PARSE_ARRAYS) against the oracle's
    parse_array_text: values, NULL elements, and for the first malformed literal of a batch the reference's error and its frame."""
    from etl_amd.arrow import columns_to_record_batch
    from etl_amd.decoder import EtlError
    bad = cells = 0
    for bi in range(batches):
        for oid, exemplars in list(ARRAYS.items()) + list(VAR_ARRAYS.items()):
            cols = [("id", SC.INT8, False, 1), ("v", oid, True, 0)]
            texts = [mutate(rng, rng.choice(exemplars), '0123456789{},"NULnul -+[]:=tf\\ .eExabcdINy_') for _ in range(40)]
            s = SC.txn([W.insert(42, [str(i), t]) for i, t in enumerate(texts)])
            buf, offs = np.frombuffer(s.bytes(), dtype=np.uint8), s.offsets
            d = Decoder(0)
            SC.simple_table(cols)(d)
            b = d.decode(buf, offs, flags=abi.F_OUTPUT_ON_DEVICE)
            assert b.rc == 0, b.error          # array cells are DEFERRED text in the arena: decoding never fails on them
            want = [oracle_list(oid, t) for t in texts]
            first_err = next((i for i, w in enumerate(want) if isinstance(w, str)), None)
            ok = True
            import re
            longish = [any(len(seg) > 40 for seg in re.split(r"[{},]", t)) for t in texts]   # may hold an element of more than 40 characters: handed back
            try:
                c = b.columns(0, parse_arrays=True)
                got = columns_to_record_batch(c, names=["id", "v"], on_text="binary").column(1).to_pylist()
                dfr = np.unpackbits(c.host_arrays(1)[1], bitorder="little")[:len(texts)]
                c.close()
                # rows that are not handed back carry the oracle's list; a handed-back row is either one with a long element, or a
                # malformed one behind such a row (include/etlg.h, ETLG_ROWS_PARSE_ARRAYS)
                ok = True
                seen_handback = False
                for i, t in enumerate(texts):
                    if dfr[i]:
                        if not (longish[i] or (seen_handback and isinstance(want[i], str))):
                            ok = False
                        seen_handback = True
                    elif isinstance(want[i], str) or got[i] != want[i]:
                        ok = False
                what = ("values", first_err, [int(x) for x in dfr[:8]])
            except EtlError as e:
                ok = first_err is not None and want[first_err] == "Err(%d)" % e.code and e.frame_index == first_err + 1
                if not ok and first_err is not None and any(longish[:e.frame_index - 1]):
                    ok = True   # (a possible hand-back in front: its own verdict decides, which this harness does not restate)
                what = ("error", e.code, e.frame_index, first_err, want[first_err] if first_err is not None else None)
            if ok and (oid in VAR_ARRAYS or not any(longish)):
                # the same cells as BigQuery rows (packed fields; a NULL element fails the batch; a malformed literal is handed back): the
                # first row that is not plain decides, on both sides
                from oracle import protobuf as PB
                o = oracle.Oracle()
                SC.simple_table(cols)(o)
                hbo = o.decode(buf, offs).host_batch()
                try:
                    prow, _idx, _host = PB.event_rows(hbo.materialize(), 0, cols, "PrimaryKey")
                    want_pb = ("ok", b"".join(prow))
                except PB.NullValuesNotSupportedInArrayInDestination as ex:
                    want_pb = ("null", str(ex))
                except RB.NeedsHost:
                    want_pb = ("host",)
                try:
                    r = b.protobuf(0)
                    got_pb = ("host",) if r.status == abi.RB_NEEDS_HOST else ("ok", r.bytes().tobytes())
                    r.close()
                except EtlError as e:
                    got_pb = ("null", e.detail) if e.kind == abi.NullValuesNotSupportedInArrayInDestination else ("err", e.description)
                if want_pb != got_pb:
                    ok = False
                    what = ("protobuf", want_pb[0], got_pb[0], str(want_pb[1:])[:80], str(got_pb[1:])[:80])
                try:     # and as ClickHouse rows (Array(Nullable(T)); a malformed literal is handed back)
                    rrow, _idx, _host = RB.encode_events(hbo.materialize(), 0, [k.type_class for k in hbo.slots[0].cols], [0, 1, 0, 0], abi.CH_MERGE_TREE)
                    want_rb = ("ok", b"".join(rrow))
                except RB.NeedsHost:
                    want_rb = ("host",)
                except RB.ConversionError as ce:
                    want_rb = ("err", str(ce))
                try:
                    r = b.rowbinary(0, [0, 1, 0, 0])
                    got_rb = ("host",) if r.status == abi.RB_NEEDS_HOST else ("ok", r.bytes().tobytes())
                    r.close()
                except EtlError as e:
                    got_rb = ("err", e.description)
                if ok and want_rb != got_rb:
                    ok = False
                    what = ("rowbinary", want_rb[0], got_rb[0], str(want_rb[1:])[:80], str(got_rb[1:])[:80])
            cells += len(texts)
            if not ok:
                bad += 1
                print("ARRAY MISMATCH oid", oid, what, repr(texts[first_err]) if first_err is not None else "")
                open("/tmp/cell_fuzz_arr_%d_%d.txt" % (oid, bi), "w").write(repr(texts))
            b.close(); d.close()
    print("array batches", batches, "cells", cells, "mismatching batches", bad)
    return bad


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    bad = cells = 0
    for bi in range(batches):
        for name, (oid, exemplars, alphabet) in CLASSES.items():
            cols = [("id", SC.INT8, False, 1), ("v", oid, True, 0)]
            texts = [mutate(rng, rng.choice(exemplars), alphabet) for _ in range(60)]
            s = SC.txn([W.insert(42, [str(i), t]) for i, t in enumerate(texts)])
            buf, offs = np.frombuffer(s.bytes(), dtype=np.uint8), s.offsets
            # the kernel path alternates: the default choice (k_fused for these narrow frames), k_cells, the multi-pass kernels
            os.environ.pop("ETLG_FUSED_KERNEL", None); os.environ.pop("ETLG_FORCE_MULTIPASS", None)
            if bi % 3 == 1:
                os.environ["ETLG_FUSED_KERNEL"] = "2"
            elif bi % 3 == 2:
                os.environ["ETLG_FORCE_MULTIPASS"] = "1"
            o, d = oracle.Oracle(), Decoder(0)
            prime = SC.simple_table(cols)
            prime(o); prime(d)
            rb = o.decode(buf, offs)
            b = d.decode(buf, offs, flags=abi.F_OUTPUT_ON_DEVICE)
            e = b.error
            got = (e.code, e.frame_index) if e else (0, -1)
            want = (rb.err_code, rb.err_frame)
            hb = rb.host_batch()
            got_rb = None
            if want[0] == 0 and got[0] == 0:   # (before the arena is downloaded: the hand-off calls take a device-resident batch)
                try:
                    r = b.rowbinary(0, [0, 1, 0, 0])
                    got_rb = ("host", None) if r.status == abi.RB_NEEDS_HOST else ("ok", r.bytes().tobytes())
                    r.close()
                except Exception as ex:
                    got_rb = ("err", getattr(ex, "description", str(ex)))
            got_hints = None
            if want[0] == 0 and got[0] == 0:   # Event::size_hint per event, computed on the device from the arena
                from oracle import size_hint as SH
                sm = abi.SizeModel()
                for kk, vv in SH.MODEL.items():
                    setattr(sm, kk, vv)
                got_hints = b.size_hints(sm)
            got_cols = None
            if want[0] == 0 and got[0] == 0:   # Arrow-layout columns of the same arena (before it is downloaded)
                try:
                    from etl_amd.arrow import columns_to_record_batch
                    cc = b.columns(0)
                    got_cols = columns_to_record_batch(cc, names=["id", "v"], on_text="binary")
                    cc.close()
                except Exception as ex:
                    got_cols = ("err", getattr(ex, "code", None), getattr(ex, "frame_index", None))
            got_json = None
            if name == "jsonb" and want[0] == 0 and got[0] == 0:   # the json column as serde_json's Display strings (ETLG_ROWS_FORMAT_JSON)
                try:
                    cc = b.columns(0, format_json=True)
                    _v, dfr, vals, oo = cc.host_arrays(1)
                    data = vals.tobytes()
                    got_json = ("ok", [data[oo[k]:oo[k + 1]] for k in range(len(texts))], list(np.unpackbits(dfr, bitorder="little")[:len(texts)]))
                    cc.close()
                except Exception as ex:
                    got_json = ("err", getattr(ex, "code", None), getattr(ex, "frame_index", None))
            got_pb = None
            if want[0] == 0 and got[0] == 0:
                try:
                    r = b.protobuf(0)
                    got_pb = ("host", None) if r.status == abi.RB_NEEDS_HOST else ("ok", r.bytes().tobytes())
                    r.close()
                except Exception as ex:
                    got_pb = ("err", getattr(ex, "detail", None) or getattr(ex, "description", str(ex)))
            diff = hb.diff(b.host())
            ok = got == want and not diff
            if ok and want[0] == 0:
                from oracle import size_hint as SH
                want_hints = np.array([SH.event_hint(e, hb.slots, SH.MODEL) for e in hb.materialize()], dtype=np.uint64)
                if not np.array_equal(want_hints, got_hints):
                    ok = False
                    w = np.flatnonzero(want_hints != got_hints)[:3]
                    diff = ["size_hints", [(int(i), int(want_hints[i]), int(got_hints[i])) for i in w]]
            if ok and want[0] == 0:
                from etl_amd.arrow import rows_to_record_batch
                from tests.test_gpu_columns import _same
                want_cols = rows_to_record_batch(hb, 0, names=["id", "v"], on_text="binary")
                if name == "jsonb":   # json cells are validated by the hand-off call (serde_json's grammar): the first cell that is not one JSON value fails it
                    verdicts = [oracle.parse_text_cell(oid, t) for t in texts]
                    firstbad = next((i for i, v in enumerate(verdicts) if v.startswith("Err(")), None)
                    if firstbad is not None:
                        want_cols = ("err", int(verdicts[firstbad][4:-1]), firstbad + 1)
                if name == "jsonb":
                    from oracle import json_display as JD
                    if firstbad is not None:
                        want_json = want_cols
                    else:
                        lim = [JD.device_limits_ok(t) for t in texts]
                        want_json = ("ok", [JD.display(t) if k else t.encode() for t, k in zip(texts, lim)], [0 if k else 1 for k in lim])
                    if want_json != got_json:
                        ok = False
                        diff = ["json columns", str(want_json)[:300], str(got_json)[:300]]
                if isinstance(want_cols, tuple) or isinstance(got_cols, tuple):
                    if want_cols != got_cols:
                        ok = False
                        diff = ["columns", str(want_cols)[:80], str(got_cols)[:80]]
                else:
                    try:
                        _same(want_cols, got_cols)
                    except AssertionError as ae:
                        ok = False
                        diff = ["columns", str(ae)[:200]]
            if ok and want[0] == 0 and "[" not in name:
                # BigQuery rows of the same arena (cell_encode_prost; numeric scale validation)
                from oracle import protobuf as PB
                try:
                    if name == "jsonb" and firstbad is not None:
                        raise PB.UnsupportedValueInDestination("JSON deserialization failed")   # (the decode error, as the device words it)
                    rows, idx, host = PB.event_rows(hb.materialize(), 0, cols, "PrimaryKey")
                    want_pb = ("ok", b"".join(rows))
                except PB.UnsupportedValueInDestination as ue:
                    want_pb = ("err", str(ue))
                except RB.NeedsHost:
                    want_pb = ("host", None)
                if want_pb != got_pb:
                    ok = False
                    diff = ["protobuf", want_pb[0], got_pb[0] if got_pb else None, (want_pb[1] or b"")[:80], (got_pb[1] or b"")[:80] if got_pb else None]
            if ok and name in ("jsonb", "numeric", "timetz", "time", "timestamptz", "timestamp", "date", "float8", "uuid", "bytea", "int4[]", "float8[]", "date[]", "timestamptz[]", "time[]", "uuid[]") and want[0] == 0:
                # the hand-off of the same arena (Display strings, Date32 range, arrays): RowBinary bytes against the oracle's encoder
                try:
                    if name == "jsonb" and firstbad is not None:
                        raise RB.ConversionError("JSON deserialization failed")
                    rows, idx, host = RB.encode_events(hb.materialize(), 0, [c.type_class for c in hb.slots[0].cols], [0, 1, 0, 0], abi.CH_MERGE_TREE)
                    want_rb = ("ok", b"".join(rows))
                except RB.ConversionError as ce:
                    want_rb = ("err", str(ce))
                except RB.NeedsHost:
                    want_rb = ("host", None)
                if want_rb != got_rb:
                    ok = False
                    diff = ["rowbinary", want_rb[0], got_rb[0], (want_rb[1] or b"")[:60], (got_rb[1] or b"")[:60]]
            cells += len(texts)
            if not ok:
                bad += 1
                bad_i = want[1] if want[1] >= 0 else -1
                print("MISMATCH", name, "want", want, "got", got, "diff", diff[:4], "text", repr(texts[bad_i - 1]) if bad_i > 0 else "")
                open("/tmp/cell_fuzz_%s_%d.txt" % (name.replace("[]", "_a"), bi), "w").write(repr(texts))
            b.close(); d.close()
    print("batches", batches, "cells", cells, "mismatching batches", bad)
    os.environ.pop("ETLG_FUSED_KERNEL", None); os.environ.pop("ETLG_FORCE_MULTIPASS", None)
    bad += fuzz_arrays(rng, batches)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
