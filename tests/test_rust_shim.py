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
    for f in ("lib.rs", "materialize.rs", "batcher.rs", "flush.rs"):
        src = open(os.path.join(CRATE, "src", f)).read()
        src = re.sub(r"//[^\n]*", "", src)
        used = set(re.findall(r"\b(etlg_[a-z_0-9]+|ETLG_[A-Z][A-Za-z0-9_]+)\b", src))
        assert used <= bound, (f, sorted(used - bound))
    assert os.path.exists(os.path.join(CRATE, "Cargo.toml")) and os.path.exists(os.path.join(CRATE, "build.rs"))


REF = "/root/reference"


def _etl_paths():
    """(crate, module path, item) for every `use etl::...` / `etl::a::b::Item` the shim names."""
    out = set()
    for f in ("lib.rs", "materialize.rs", "batcher.rs", "flush.rs"):
        src = re.sub(r"//[^\n]*", "", open(os.path.join(CRATE, "src", f)).read())
        for m in re.finditer(r"use (etl|etl_postgres)::([a-z_:]+)::\{([^}]+)\};", src):
            for item in m.group(3).split(","):
                item = item.strip().split(" as ")[0]
                if item:
                    out.add((m.group(1), m.group(2), item))
        for m in re.finditer(r"use (etl|etl_postgres)::([a-z_:]+)::([A-Za-z_0-9]+);", src):
            out.add((m.group(1), m.group(2), m.group(3)))
        for m in re.finditer(r"use (etl)::(bail|etl_error);", src):
            out.add((m.group(1), "", m.group(2)))
    return sorted(out)


def test_every_reference_path_the_shim_names_exists():
    """grep-level check against the checkout the shim targets (it cannot be compiled here): each `etl::module::Item` resolves to a
    module file of crates/etl (or crates/etl-postgres) that declares or re-exports an item of that name."""
    import pytest
    if not os.path.isdir(REF):
        pytest.skip("the reference checkout is only present in the build container")
    paths = _etl_paths()
    assert len(paths) >= 15, paths
    for crate, mod, item in paths:
        root = os.path.join(REF, "crates", "etl" if crate == "etl" else "etl-postgres", "src")
        if not mod:   # exported macros
            text = "".join(open(os.path.join(dp, fn)).read() for dp, _, fns in os.walk(root) for fn in fns if fn.endswith(".rs"))
            assert re.search(r"macro_rules!\s+" + item + r"\b", text), (crate, item)
            continue
        rel = mod.replace("::", "/")
        cands = [os.path.join(root, rel + ".rs"), os.path.join(root, rel, "mod.rs")]
        files = [c for c in cands if os.path.exists(c)]
        assert files, (crate, mod, item, "no such module")
        text = open(files[0]).read()
        if os.path.isdir(os.path.join(root, rel)):   # a module directory: the item may live in a child it re-exports
            text += "".join(open(os.path.join(root, rel, fn)).read() for fn in os.listdir(os.path.join(root, rel)) if fn.endswith(".rs"))
        assert re.search(r"\b(struct|enum|trait|type|fn|const|static|mod)\s+" + item + r"\b|pub use [^;]*\b" + item + r"\b", text), (crate, mod, item)


def test_the_patch_quotes_the_reference():
    """patches/*.diff: their context lines (the ones without +/-) are lines of the reference file they patch."""
    import pytest
    if not os.path.isdir(REF):
        pytest.skip("the reference checkout is only present in the build container")
    for name, target in (("apply_rs.diff", "crates/etl/src/replication/apply.rs"), ("table_copy_rs.diff", "crates/etl/src/postgres/stream/table_copy.rs")):
        ref = open(os.path.join(REF, target)).read()
        patch = open(os.path.join(CRATE, "patches", name)).read()
        body = patch[patch.index("--- a/" + target):]
        ctx = [l[1:].strip() for l in body.split("\n") if l.startswith(" ") and l.strip()]
        assert len(ctx) >= 8, name
        for l in ctx:
            assert l in ref, (name, l)
        for hunk in re.findall(r"^@@ (.+)$", body, flags=re.M):
            assert hunk.strip().split("(")[0] in ref, (name, hunk)
