"""rust/magics-hip is delivered as source (no Rust toolchain in the image): what CAN be checked here is checked — the
`extern "C"` block of src/sys.rs against include/mgx.h, function by function (name, arity, every parameter and return
type under the C -> Rust mapping of tools/gen_rust_sys.py), the #[repr(C)] structs field by field, and that lib.rs only
calls functions sys.rs declares."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_sys as gen  # noqa: E402


def test_every_header_function_is_declared_with_the_same_signature():
    protos = gen.c_prototypes()
    assert len(protos) >= 80
    want = {n: (("" if r == "void" else gen.rust_type(r)), [gen.rust_type(t) for t, _ in ps]) for n, r, ps in protos}
    have = {n: (r, ps) for n, r, ps in gen.rust_declarations()}
    assert sorted(have) == sorted(want), (sorted(set(want) - set(have)), sorted(set(have) - set(want)))
    for n in want:
        assert have[n] == want[n], (n, have[n], want[n])


def test_sys_rs_is_exactly_what_the_generator_emits():
    s = open(gen.SYS_RS).read()
    assert gen.BEGIN in s and gen.END in s
    assert s[s.index(gen.BEGIN):s.index(gen.END) + len(gen.END)] == gen.generate()


def _c_struct_fields(name):
    hdr = re.sub(r"/\*.*?\*/", "", open(gen.HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*(?:\s*,\s*\*?\s*[A-Za-z_][A-Za-z0-9_]*)*)$", decl)
        ctype, names = m.group(1).strip(), [x.strip() for x in m.group(2).split(",")]
        for nm in names:
            out.append((nm.lstrip("* ").lower(), gen.rust_type(" ".join((ctype + (" *" if nm.startswith("*") else "")).replace("*", " * ").split()))))
    return out


def test_repr_c_structs_match_the_header_field_by_field():
    rs = open(gen.SYS_RS).read()
    for name in ("mgx_params", "mgx_robot_desc", "mgx_env_obstacle", "mgx_env_desc"):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, rs, flags=re.S).group(1)
        have = [(m.group(1), " ".join(m.group(2).split())) for m in re.finditer(r"pub ([a-z_0-9]+):\s*([^,\n]+),", body)]
        assert have == _c_struct_fields(name), (name, have, _c_struct_fields(name))
        assert re.search(r"#\[repr\(C\)\]\s*(#\[derive\([^)]*\)\]\s*)?pub struct %s " % name, rs), name


def test_lib_rs_calls_only_declared_functions():
    lib = open(os.path.join(ROOT, "rust", "magics-hip", "src", "lib.rs")).read()
    used = set(re.findall(r"sys::(mgx_[a-z0-9_]+)\s*\(", lib))
    declared = {n for n, _, _ in gen.rust_declarations()}
    assert used and used <= declared, used - declared
