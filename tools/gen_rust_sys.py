"""include/mgx.h -> the `extern "C"` block of rust/magics-hip/src/sys.rs (what `bindgen include/mgx.h` would emit
for the functions; the structs and constants at the top of sys.rs are written by hand).

    python tools/gen_rust_sys.py            # prints the block
    python tools/gen_rust_sys.py --write    # rewrites the block inside sys.rs (between the BEGIN / END markers)

tests/test_rust_shim.py runs the same parser over both files: every function of the header must be declared in
sys.rs with the same arity and the same (mapped) types, and nothing else."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mgx.h")
SYS_RS = os.path.join(ROOT, "rust", "magics-hip", "src", "sys.rs")
BEGIN, END = "// BEGIN generated from include/mgx.h (tools/gen_rust_sys.py)", "// END generated"

SCALARS = {"int": "c_int", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "uint8_t": "u8", "double": "f64",
           "float": "f32", "char": "c_char", "void": "c_void"}
STRUCTS = {"mgx_world", "mgx_params", "mgx_robot_desc", "mgx_env_desc", "mgx_env_obstacle", "mgx_mvn", "mgx_shard_plan", "mgx_mission_desc", "mgx_mission_run_desc"}
RUST_KEYWORDS = {"type", "ref", "in", "fn", "mod", "use", "where", "self", "move", "box", "loop", "match"}


def c_prototypes(text=None):
    """[(name, return type, [(param type, param name)])] of every mgx_* function the header declares"""
    src = text if text is not None else open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    out = []
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(mgx_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret or "struct" in ret.split():
            continue
        plist = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                arr = re.search(r"\[\s*\d*\s*\]$", p)      # `double mean[4]` decays to a pointer
                if arr:
                    p = p[:arr.start()].strip()
                pm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", p)
                ptype, pname = pm.group(1).strip(), pm.group(2)
                if arr:
                    ptype += " *"
                plist.append((" ".join(ptype.replace("*", " * ").split()), pname))
        out.append((name, " ".join(ret.replace("*", " * ").split()), plist))
    return out


def rust_type(ctype):
    """`const T *` -> *const T, `T **` -> *mut *mut T, `T *const *` -> *const *mut T (a qualifier after a `*` belongs to that pointer)"""
    toks = ctype.split()
    pointee_const, i = False, 0
    while toks[i] == "const":
        pointee_const, i = True, i + 1
    base = toks[i]
    i += 1
    if i < len(toks) and toks[i] == "const":      # `T const *`
        pointee_const, i = True, i + 1
    t = SCALARS.get(base) or (base if base in STRUCTS else None)
    assert t is not None, f"unmapped C type {ctype!r}"
    while i < len(toks):
        assert toks[i] == "*", ctype
        t = f"*{'const' if pointee_const else 'mut'} {t}"
        i += 1
        pointee_const = False
        if i < len(toks) and toks[i] == "const":
            pointee_const, i = True, i + 1
    return t


def rust_decl(name, ret, params):
    args = ", ".join(f"{(p + '_') if p in RUST_KEYWORDS else p.lower()}: {rust_type(t)}" for t, p in params)
    r = "" if ret == "void" else f" -> {rust_type(ret)}"
    return f"    pub fn {name}({args}){r};"


def generate():
    return "\n".join([BEGIN, 'extern "C" {'] + [rust_decl(*p) for p in c_prototypes()] + ["}", END])


def rust_declarations(text=None):
    """[(name, return type or '', [param types])] of the extern block of sys.rs"""
    src = text if text is not None else open(SYS_RS).read()
    out = []
    for m in re.finditer(r"pub fn (mgx_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", src, flags=re.S):
        params = [" ".join(p.split(":", 1)[1].split()) for p in m.group(2).split(",") if ":" in p]
        out.append((m.group(1), (m.group(3) or "").strip(), params))
    return out


if __name__ == "__main__":
    block = generate()
    if "--write" in sys.argv:
        s = open(SYS_RS).read()
        if BEGIN in s:
            s = s[:s.index(BEGIN)] + block + s[s.index(END) + len(END):]
        elif 'extern "C" {' in s:
            s = s[:s.index('extern "C" {')] + block + "\n"
        else:
            s = s + block + "\n"
        open(SYS_RS, "w").write(s)
        print(f"wrote {len(c_prototypes())} declarations to {SYS_RS}")
    else:
        print(block)
