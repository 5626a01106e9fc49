"""crates/etl-gfx950 is shipped as source (no Rust toolchain in this image — SURVEY.md §8(f)#2). What CAN be checked here is
checked: `src/ffi.rs` against `include/etlg.h`, mechanically —
  * every function the header declares has an `extern "C"` item of the same name, arity and argument / return types;
  * every struct has a `#[repr(C)]` twin with the same fields in the same order and type-compatible widths;
  * every enumerator the shim uses has the header's value;
and that the other files reference only symbols `ffi.rs` defines."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "etlg.h")).read()
CRATE = os.path.join(ROOT, "crates", "etl-gfx950")
FFI = open(os.path.join(CRATE, "src", "ffi.rs")).read()

C2RUST = {
    "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "uint8_t": "u8", "int16_t": "i16", "uint16_t": "u16",
    "size_t": "usize", "double": "f64", "void": "()", "char": "c_char",
}


def _strip_comments(s):
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def _ctype(t):
    t = t.replace("struct ", "").strip()
    const = "const" in t.split("*")[0]
    base = t.replace("const", "").replace("*", "").strip()
    stars = t.count("*")
    r = C2RUST.get(base, base)
    if base == "void" and stars:
        r = "c_void"
    for _ in range(stars):
        r = ("*const " if const else "*mut ") + r
        const = False if stars > 1 else const
    return r


def _c_functions():
    h = _strip_comments(HDR)
    out = {}
    for m in re.finditer(r"^\s*(const\s+)?([A-Za-z_0-9]+)\s*(\*?)\s*(etlg_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.M | re.S):
        ret = ((m.group(1) or "") + m.group(2) + m.group(3)).strip()
        args = []
        raw = " ".join(m.group(5).split())
        if raw != "void":
            for a in raw.split(","):
                a = a.strip()
                name = re.search(r"([A-Za-z_0-9]+)$", a).group(1)
                args.append((name, _ctype(a[:len(a) - len(name)])))
        out[m.group(4)] = (_ctype(ret), args)
    return out


def _rust_functions():
    out = {}
    block = FFI[FFI.index('extern "C" {'):]
    for m in re.finditer(r"pub fn (etlg_[a-z_0-9]+)\s*\((.*?)\)\s*(->\s*([^;]+))?;", block, flags=re.S):
        args = []
        raw = " ".join(m.group(2).split()).rstrip(",").strip()
        if raw:
            for a in raw.split(","):
                n, t = a.strip().split(":", 1)
                args.append((n.strip(), t.strip()))
        out[m.group(1)] = ((m.group(4) or "()").strip(), args)
    return out


def test_every_exported_function_is_bound_with_the_same_signature():
    c, r = _c_functions(), _rust_functions()
    from etl_amd import native
    assert set(c) == set(native.EXPORTS), sorted(set(c) ^ set(native.EXPORTS))
    assert set(c) == set(r), sorted(set(c) ^ set(r))
    for name, (ret, args) in c.items():
        rret, rargs = r[name]
        assert rret == ret, (name, rret, ret)
        assert [a[0] for a in rargs] == [a[0] for a in args], (name, rargs, args)
        for (an, ct), (_, rt) in zip(args, rargs):
            assert rt == ct, (name, an, rt, ct)


def _c_structs():
    h = _strip_comments(HDR)
    out = {}
    for m in re.finditer(r"typedef struct (etlg_[a-z_]+) \{(.*?)\} \1;", h, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(.+?)\s*(\*?)\s*([A-Za-z_0-9]+)(\[(\d+)\])?$", decl)
            t, star, name, _, arr = fm.groups()
            rt = _ctype(t + star)
            if arr:
                rt = f"[{rt}; {arr}]"
            fields.append((name, rt))
        out[m.group(1)] = fields
    return out


def _rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (etlg_[a-z_]+) \{(.*?)\n\}", FFI, flags=re.S):
        fields = []
        for line in m.group(2).split("\n"):
            line = line.strip().rstrip(",")
            fm = re.match(r"(pub )?([A-Za-z_0-9]+): (.+)$", line)
            if fm:
                fields.append((fm.group(2), fm.group(3)))
        out[m.group(1)] = fields
    return out


def test_structs_have_the_same_fields_in_the_same_order():
    c, r = _c_structs(), _rust_structs()
    opaque = {"etlg_ctx", "etlg_batch", "etlg_columns", "etlg_rowbinary"}
    assert set(c) == set(r) - opaque, sorted(set(c) ^ (set(r) - opaque))
    for name, fields in c.items():
        assert r[name] == fields, (name, r[name], fields)


def test_constants_match_the_header():
    h = _strip_comments(HDR)
    vals = {}
    for m in re.finditer(r"\b(ETLG_[A-Za-z0-9_]+)\s*=\s*([^,}\n]+)", h):
        v = m.group(2).strip()
        if v.startswith("'"):
            vals[m.group(1)] = ord(v[1])
        else:
            v = v.replace("u", "")
            try:
                vals[m.group(1)] = int(eval(v))
            except Exception:
                pass
    for m in re.finditer(r"#define (ETLG_[A-Z0-9_]+) (\d+)u", HDR):
        vals[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"#define (ETLG_[A-Z0-9_]+) \((\d+)ull << (\d+)\)", HDR):
        vals[m.group(1)] = int(m.group(2)) << int(m.group(3))
    seen = 0
    for m in re.finditer(r"pub const (ETLG_[A-Za-z0-9_]+): [a-z0-9]+ = ([^;]+);", FFI):
        name, v = m.group(1), m.group(2).strip()
        val = ord(v[2]) if v.startswith("b'") else int(eval(v))
        assert name in vals, name
        assert vals[name] == val, (name, vals[name], val)
        seen += 1
    assert seen > 60


def test_other_sources_use_only_bound_symbols():
    bound = set(re.findall(r"\b(etlg_[a-z_0-9]+|ETLG_[A-Za-z0-9_]+)\b", FFI))
    for f in ("lib.rs", "materialize.rs", "batcher.rs"):
        src = open(os.path.join(CRATE, "src", f)).read()
        src = re.sub(r"//[^\n]*", "", src)
        used = set(re.findall(r"\b(etlg_[a-z_0-9]+|ETLG_[A-Z][A-Za-z0-9_]+)\b", src))
        assert used <= bound, (f, sorted(used - bound))
    assert os.path.exists(os.path.join(CRATE, "Cargo.toml")) and os.path.exists(os.path.join(CRATE, "build.rs"))
