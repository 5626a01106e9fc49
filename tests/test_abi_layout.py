"""The drop-in boundary is plain C: include/etlg.h must compile as C11 on its own, and the ctypes mirror of
its structs (etl_amd/abi.py — the binding a cgo / bindgen / JNI user would write the same way) must agree with
the C compiler on every size and field offset."""
import ctypes as C
import os
import subprocess

from etl_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STRUCTS = {
    "etlg_error": (abi.Error, ["kind", "code", "description", "detail", "frame_index"]),
    "etlg_err_desc": (abi.ErrDesc, ["kind", "description"]),
    "etlg_col": (abi.Col, ["name", "type_oid", "type_modifier", "attnum", "nullable", "primary_key"]),
    "etlg_slot_col": (abi.SlotCol, ["type_oid", "stored_index", "type_class", "nullable", "identity", "off_full", "off_key", "key_index"]),
    "etlg_slot_desc": (abi.SlotDesc, ["table_id", "n_stored", "snapshot_lsn", "n_cols", "n_ident", "row_bytes_full", "row_bytes_key",
                                      "state_bytes_full", "state_bytes_key", "cols"]),
    "etlg_batch_view": (abi.BatchView, ["n_events", "n_frames", "fixed_bytes", "heap_bytes", "payload_bytes", "ev_kind", "ev_flags",
                                        "ev_table_id", "ev_schema_slot", "ev_start_lsn", "ev_commit_lsn", "ev_tx_ordinal", "ev_body_off",
                                        "fixed", "heap", "on_device", "n_slots", "slots"]),
    "etlg_kernel_stat": (abi.KernelStat, ["name", "launches", "total_ms"]),
}


def test_header_is_plain_c_and_layouts_match_ctypes(tmp_path):
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "etlg.h"', 'int main(void) {']
    for name, (_, fields) in STRUCTS.items():
        lines.append(f'  printf("{name} %zu", sizeof({name}));')
        for f in fields:
            lines.append(f'  printf(" %zu", offsetof({name}, {f}));')
        lines.append('  printf("\\n");')
    lines.append('  printf("etlg_numeric_hdr %zu\\n", sizeof(etlg_numeric_hdr));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    # -std=c11 -pedantic-errors: the header alone, no C++-isms, no compiler extensions
    subprocess.check_call(["gcc", "-std=c11", "-pedantic-errors", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split("\n")
    seen = {}
    for line in out:
        if line.strip():
            parts = line.split()
            seen[parts[0]] = [int(x) for x in parts[1:]]
    for name, (cls, fields) in STRUCTS.items():
        want = [C.sizeof(cls)] + [getattr(cls, f).offset for f in fields]
        assert seen[name] == want, (name, seen[name], want)
    assert seen["etlg_numeric_hdr"] == [8]


def test_event_header_columns_are_42_bytes():
    """bench.py prices the event header at 42 bytes per event: one byte each for kind and flags,
    four each for table id and schema slot, eight each for start LSN, commit LSN, ordinal and body offset."""
    import bench
    assert bench.HEADER_BYTES_PER_EVENT == 1 + 1 + 4 + 4 + 8 + 8 + 8 + 8
