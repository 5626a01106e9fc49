"""TEST INFRASTRUCTURE — CPU restatement of the Display impls every sink of the reference goes through for numeric, time
and timetz cells (ClickHouse `n.to_string()` clickhouse/encoding.rs:66-71, BigQuery bigquery/encoding.rs:146-161, Arrow
`cell_to_string` iceberg/encoding.rs:349-352). Never imported by the product path.

  numeric_string  format_numeric_value, crates/etl-postgres/src/numeric.rs:460-560 (line by line, incl. the i16 `weight + 1`)
  timetz_string   PgTimeTz Display, crates/etl-postgres/src/time.rs:113-117 + write_utc_offset :210-225
  time_string     chrono 0.4.44 `%H:%M:%S%.f` (TIME_FORMAT, time.rs:17) and NaiveTime's Display: chrono is not vendored under
                  /root/reference; restated from its published behaviour — `%.f` prints nothing, or exactly 3 / 6 / 9 digits;
                  a leap second (nanos >= 10^9 on second 59) prints as second 60.

Pinned by tests/test_oracle_display.py to the reference's own expectations (numeric.rs:694-696, 735-792, 850-943; time.rs:231-258).
Cells are the tuples of etl_amd.view.HostBatch.materialize(): ("Numeric", kind, sign, weight, scale, digits), ("TimeTz", secs, nanos, offset)."""

NUM_VALUE, NUM_NAN, NUM_PINF, NUM_NINF = 0, 1, 2, 3


def _i16(v):
    v &= 0xFFFF
    return v - 0x10000 if v >= 0x8000 else v


def numeric_string(kind, sign, weight, scale, digits):
    if kind == NUM_NAN:
        return "NaN"
    if kind == NUM_PINF:
        return "Infinity"
    if kind == NUM_NINF:
        return "-Infinity"
    out = []
    if len(digits) == 0:                       # :492-503: zero keeps its display scale
        out.append("0")
        if scale > 0:
            out.append("." + "0" * scale)
        return "".join(out)
    if sign:                                   # :506-508
        out.append("-")
    if weight < 0:                             # :510-512
        out.append("0")
    else:
        for d in range(0, weight + 1):         # :515-532
            g = digits[d] if d < len(digits) else 0
            four = f"{g:04}"
            if d == 0:
                t = four.lstrip("0")
                out.append(t if t else "0")
            else:
                out.append(four)
    if scale > 0:                              # :535-558
        out.append(".")
        remaining = scale
        d = _i16(weight + 1)                   # `weight + 1` on an i16 (a release build wraps)
        while remaining > 0:
            g = digits[d] if 0 <= d < len(digits) else 0
            four = f"{g:04}"
            take = min(4, remaining)
            out.append(four[:take])
            remaining -= take
            d += 1
    return "".join(out)


def time_string(secs, nanos):
    leap = 1 if nanos >= 1_000_000_000 else 0
    nanos -= leap * 1_000_000_000
    h, m, s = secs // 3600, secs // 60 % 60, secs % 60 + leap
    out = f"{h:02}:{m:02}:{s:02}"
    if nanos == 0:
        return out
    if nanos % 1_000_000 == 0:
        return out + f".{nanos // 1_000_000:03}"
    if nanos % 1_000 == 0:
        return out + f".{nanos // 1_000:06}"
    return out + f".{nanos:09}"


def utc_offset_string(seconds):
    sign = "-" if seconds < 0 else "+"
    seconds = abs(seconds)
    hours, minutes, rem = seconds // 3600, seconds % 3600 // 60, seconds % 60
    if rem != 0:
        return f"{sign}{hours:02}:{minutes:02}:{rem:02}"
    if minutes != 0:
        return f"{sign}{hours:02}:{minutes:02}"
    return f"{sign}{hours:02}"


def timetz_string(secs, nanos, offset):
    return time_string(secs, nanos) + utc_offset_string(offset)
