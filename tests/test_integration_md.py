"""INTEGRATION.md's Rust bindings are checked against include/*.h by machine: the image has no rustc, so a binding that drifts from the
header (round 5: v2p_gir_collect documented with two arguments, exported with three) would otherwise only be found by a maintainer's
segfault.  Every `extern "C" { pub fn ... }` declaration and every `#[repr(C)] pub struct` with fields inside the document's
```rust blocks must agree with the C prototype / typedef'd struct of the same name: name, arity, argument order, argument names
(Rust keywords aside) and types.  Reference shapes the bindings stand for: gir.rs:283-299 (the SoA marshaller),
personalized_genome.rs:64-65 (two GIRs per sample)."""
from __future__ import annotations

import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = ["vcf2prot_hip.h", "v2p_frontend.h", "v2p_cohort.h", "v2p_step4a.h", "v2p_step4b.h"]

# ---- C side -------------------------------------------------------------------------------------------------------------------


def _strip_c_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _norm_c_type(t: str) -> str:
    """'const uint64_t *' -> 'const uint64_t*'; 'struct v2p_ctx*' -> 'v2p_ctx*'; 'unsigned int' -> 'unsigned'."""
    t = re.sub(r"\bstruct\s+", "", t)
    t = re.sub(r"\s+", " ", t).strip()
    t = re.sub(r"\s*\*\s*", "*", t)
    t = t.replace("unsigned int", "unsigned")
    return t


def _split_c_decl(decl: str):
    """'const uint64_t* start_pos' -> ('const uint64_t*', 'start_pos'); 'float x[32]' -> ('float[32]', 'x'); 'void' -> None."""
    decl = decl.strip()
    if decl in ("", "void"):
        return None
    arr = ""
    m = re.search(r"\[(\w+)\]\s*$", decl)
    if m:
        arr = "[" + m.group(1) + "]"
        decl = decl[: m.start()].strip()
    m = re.match(r"^(.*?[\s\*])(\w+)$", decl)
    if m and m.group(2) not in ("int", "char", "float", "double", "unsigned", "void") and not m.group(2).endswith("_t"):
        return _norm_c_type(m.group(1)) + arr, m.group(2)
    return _norm_c_type(decl) + arr, None       # unnamed parameter


def parse_c_headers():
    funcs, structs = {}, {}
    for h in HEADERS:
        text = _strip_c_comments(open(os.path.join(ROOT, "include", h)).read())
        text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
        for m in re.finditer(r"typedef\s+struct\s*(\w*)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
            fields = []
            for stmt in m.group(2).split(";"):
                stmt = stmt.strip()
                if not stmt:
                    continue
                # 'uint64_t a, b, c' and 'const uint64_t* p'
                first = _split_c_decl(stmt.split(",")[0])
                assert first and first[1], (h, stmt)
                fields.append((first[1], first[0]))
                base = re.sub(r"[\*\s]+$", "", re.match(r"^(.*?)(\w+)(\[\w+\])?$", stmt.split(",")[0].strip()).group(1))
                for more in stmt.split(",")[1:]:
                    more = more.strip()
                    stars = more.count("*")
                    name = more.replace("*", "").strip()
                    fields.append((name, _norm_c_type(base + "*" * stars)))
            structs[m.group(3)] = fields
        body = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
        for m in re.finditer(r"([\w\s\*]+?)\b(v2p_\w+)\s*\(([^()]*)\)\s*;", body):
            ret = _norm_c_type(m.group(1))
            if ret.startswith("typedef") or not ret:
                continue
            args = [a for a in (_split_c_decl(x) for x in m.group(3).split(",")) if a is not None]
            funcs[m.group(2)] = (ret, args)
    return funcs, structs


# ---- Rust side ----------------------------------------------------------------------------------------------------------------

_PRIM = {"c_int": "int", "c_uint": "unsigned", "c_char": "char", "c_void": "void", "u8": "uint8_t", "u16": "uint16_t", "u32": "uint32_t",
         "u64": "uint64_t", "i8": "int8_t", "i16": "int16_t", "i32": "int32_t", "i64": "int64_t", "f32": "float", "f64": "double",
         "usize": "size_t"}
_RENAMED = {"ref_tape": "ref", "alt_tape": "alt", "type_": "type", "in_": "in"}      # Rust keywords / reserved words


def _camel_key(name: str) -> str:
    return name.replace("_", "").lower()


def rust_type_to_c(t: str, c_struct_names) -> str:
    t = t.strip()
    m = re.match(r"^\[\s*(\w+)\s*;\s*(\w+)\s*\]$", t)
    if m:
        return rust_type_to_c(m.group(1), c_struct_names) + "[" + m.group(2) + "]"
    if t.startswith("*const "):
        inner = rust_type_to_c(t[len("*const "):], c_struct_names)
        # const binds to the pointee: '*const *const T' -> 'const T* const*' never appears in the headers; keep the simple rule
        return ("const " + inner + "*") if not inner.endswith("*") else (inner + " const*")
    if t.startswith("*mut "):
        return rust_type_to_c(t[len("*mut "):], c_struct_names) + "*"
    if t in _PRIM:
        return _PRIM[t]
    key = _camel_key(t)
    for c in c_struct_names:
        if _camel_key(c) == key:
            return c
    raise AssertionError(f"INTEGRATION.md uses a type the headers do not know: {t}")


def rust_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```rust\n(.*?)```", text, flags=re.S)


def _strip_rust_comments(src: str) -> str:
    return re.sub(r"//[^\n]*", " ", src)


def parse_rust():
    fns, structs = [], []
    for blk in rust_blocks():
        src = _strip_rust_comments(blk)
        for ext in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S):
            for m in re.finditer(r"pub\s+fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext.group(1), flags=re.S):
                args = []
                for a in [x for x in re.split(r",(?![^\[]*\])", m.group(2)) if x.strip()]:
                    name, ty = a.split(":", 1)
                    args.append((name.strip(), ty.strip()))
                fns.append((m.group(1), (m.group(3) or "").strip(), args))
        for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\)]*\)\]\s*)?pub\s+struct\s+(\w+)\s*\{(.*?)\}", src, flags=re.S):
            fields = []
            for f in [x for x in re.split(r",(?![^\[]*\])", m.group(2)) if x.strip()]:
                name, ty = f.split(":", 1)
                fields.append((name.replace("pub", "").strip(), ty.strip()))
            structs.append((m.group(1), fields))
    return fns, structs


# ---- the checks ---------------------------------------------------------------------------------------------------------------

C_FUNCS, C_STRUCTS = parse_c_headers()
C_OPAQUE = set()
for _h in HEADERS:
    C_OPAQUE |= set(re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", open(os.path.join(ROOT, "include", _h)).read()))
C_TYPE_NAMES = sorted(set(C_STRUCTS) | C_OPAQUE | {"v2p_ctx"})
RUST_FNS, RUST_STRUCTS = parse_rust()


def test_parsers_see_the_document_and_the_headers():
    assert len(C_FUNCS) > 60 and "v2p_gir_collect" in C_FUNCS and "v2p_decode_run" in C_FUNCS
    assert C_FUNCS["v2p_gir_collect"][1] == [("v2p_ctx*", "ctx"), ("v2p_gir_ticket*", "ticket"), ("int64_t*", "err_row")]
    assert "v2p_txstream" in C_STRUCTS and len(C_STRUCTS["v2p_txstream"]) == 17
    assert ("slice_build_ms", "float[32]") in C_STRUCTS["v2p_oneshot_info"]
    names = {f[0] for f in RUST_FNS}
    assert {"v2p_execute_gir", "v2p_execute_gir_shared", "v2p_gir_submit", "v2p_gir_collect", "v2p_stream_upload",
            "v2p_batch_build_and_execute", "v2p_decode_run"} <= names
    assert {s[0] for s in RUST_STRUCTS} >= {"V2pTxStream", "V2pRouting", "V2pOneshotInfo"}


@pytest.mark.parametrize("fn", RUST_FNS, ids=[f[0] for f in RUST_FNS])
def test_extern_fn_matches_header(fn):
    name, ret, args = fn
    assert name in C_FUNCS, f"INTEGRATION.md binds {name}, which no header under include/ declares"
    c_ret, c_args = C_FUNCS[name]
    want_ret = rust_type_to_c(ret, C_TYPE_NAMES) if ret else "void"
    assert want_ret == c_ret, f"{name}: returns {c_ret} in the header, {ret or '()'} in the document"
    assert len(args) == len(c_args), f"{name}: {len(c_args)} arguments in the header {[a[1] for a in c_args]}, {len(args)} in the document {[a[0] for a in args]}"
    for k, ((r_name, r_ty), (c_ty, c_name)) in enumerate(zip(args, c_args)):
        assert rust_type_to_c(r_ty, C_TYPE_NAMES) == c_ty, f"{name}, argument {k} ({r_name}): header has {c_ty}, document has {r_ty}"
        if c_name is not None:
            assert _RENAMED.get(r_name, r_name) == c_name, f"{name}, argument {k}: header calls it {c_name}, document {r_name}"


_FIELD_STRUCTS = [s for s in RUST_STRUCTS if not (len(s[1]) == 1 and s[1][0][0] == "_p")]       # opaque handles carry one zero-sized field


@pytest.mark.parametrize("st", _FIELD_STRUCTS, ids=[s[0] for s in _FIELD_STRUCTS])
def test_repr_c_struct_matches_header(st):
    name, fields = st
    c_name = next((c for c in C_STRUCTS if _camel_key(c) == _camel_key(name)), None)
    assert c_name, f"#[repr(C)] struct {name} has no typedef'd struct in include/"
    c_fields = C_STRUCTS[c_name]
    assert [f[0] for f in fields] == [f[0] for f in c_fields], f"{name}: field names / order differ from {c_name}"
    for (r_name, r_ty), (_, c_ty) in zip(fields, c_fields):
        assert rust_type_to_c(r_ty, C_TYPE_NAMES) == c_ty, f"{name}.{r_name}: header has {c_ty}, document has {r_ty}"


def test_opaque_handles_name_real_types():
    for name, fields in RUST_STRUCTS:
        if len(fields) == 1 and fields[0][0] == "_p":
            assert any(_camel_key(c) == _camel_key(name) for c in C_TYPE_NAMES), f"opaque handle {name} has no counterpart in include/"


def test_submit_collect_sample_keeps_every_array_alive():
    """vcf2prot_hip.h: every array given to v2p_gir_submit (and `res`) must stay valid until v2p_gir_collect.  The sample worker of
    section 3 parks what it holds in `held`: the tuple must own the Task arrays and both tapes next to `res`."""
    src = next(b for b in rust_blocks() if "fn run_of_haplotypes" in b)
    m = re.search(r"struct Held\s*\{(.*?)\}", src, flags=re.S)
    assert m, "the sample's `Held` struct"
    owned = {f.split(":")[0].strip(): f.split(":")[1].strip() for f in m.group(1).split(",") if ":" in f}
    for v in ("code", "sp", "ln", "sr", "ref_a", "alt_a", "res"):
        assert owned.get(v, "").startswith("Vec<"), f"`{v}` is not owned by what the worker parks: it would drop before its ticket is collected"
    assert re.search(r"let mut held\s*:\s*Option<Held>", src)
    m = re.search(r"held\s*=\s*Some\(Held\s*\{(.*?)\}\)\s*;", src, flags=re.S)
    assert m
    kept = [x.split(":")[0].strip() for x in m.group(1).split(",")]
    for v in ("ticket", "code", "sp", "ln", "sr", "ref_a", "alt_a", "res"):
        assert v in kept, f"`{v}` is dropped before its ticket is collected"
    assert re.search(r"v2p_gir_collect\(\s*SHARED\.0\s*,\s*h\.ticket", src), "collect takes the context first"
    assert "may be dropped when it returns" not in open(os.path.join(ROOT, "INTEGRATION.md")).read()
