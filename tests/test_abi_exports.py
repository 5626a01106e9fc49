"""The C-ABI shared library loads without a GPU and exports every function include/etlg.h declares;
the host-only entry points (version, error table, type map, slot layout rule) agree with the oracle."""
import ctypes as C
import os
import re

from etl_amd import abi, native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "etlg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(etlg_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_are_exported():
    L = native.lib()
    names = declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # the Python binding's own list is the header's list
    assert set(native.EXPORTS) == set(names), sorted(set(names) ^ set(native.EXPORTS))


def test_host_only_entry_points_match_the_oracle():
    from oracle import oracle
    L = native.lib()
    o = oracle.Oracle()
    L.etlg_abi_version.restype = C.c_uint32
    assert L.etlg_abi_version() >= 1

    class Desc(C.Structure):
        _fields_ = [("kind", C.c_int32), ("description", C.c_char_p)]
    L.etlg_err_table.restype = C.POINTER(Desc)
    L.etlg_err_table.argtypes = [C.c_int32]
    o.L.oracle_err_description.restype = C.c_char_p
    for code in range(abi.E__COUNT):
        d = L.etlg_err_table(code).contents
        assert d.kind == o.L.oracle_err_kind(code), code
        assert d.description == o.L.oracle_err_description(code), code
    assert not L.etlg_err_table(abi.E__COUNT)
    L.etlg_type_class_of_oid.restype = C.c_int32
    L.etlg_slot_bytes.restype = C.c_uint32
    for oid in list(range(16, 30)) + [114, 700, 701, 1042, 1043, 1082, 1083, 1114, 1184, 1266, 1700, 2950, 3802,
                                        1000, 1001, 1005, 1007, 1009, 1016, 1021, 1022, 1028, 1115, 1182, 1183, 1185, 1231, 1270, 2951, 199, 3807, 99999]:
        assert L.etlg_type_class_of_oid(oid) == o.L.oracle_class_of_oid(oid), oid
        assert L.etlg_array_elem_class(oid) == o.L.oracle_array_elem_class(oid), oid
    for cls in range(18):
        assert L.etlg_slot_bytes(cls) == o.L.oracle_slot_bytes(cls), cls


def test_one_code_path_per_kernel():
    """The flag variants of round 1 are gone: no per-source feature flags in the build, none of their names in the kernel
    sources ("collapse the variant matrix")."""
    import os
    import re
    from etl_amd import build
    assert build.DEFS == {}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"ETLG_(FIXED_TILE|HOT_FIXES|SCALAR_COLS|EARLY_SPAN|STAGE_WIDE|BLK128|TICKET|ABLATE|DECODE_NOINLINE)\b|if \(false\)")
    csrc = os.path.join(root, "etl_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".cpp")):
            hits = [ln for ln in open(os.path.join(csrc, f)).read().split("\n") if pat.search(ln)]
            assert not hits, (f, hits[:3])
