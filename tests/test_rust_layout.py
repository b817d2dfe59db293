"""The Rust side of the boundary (rust/rpt-gpu-sys, rust/rpt.patch) cannot be compiled in this image (no
cargo); this test keeps it honest against include/rpt_gpu.h by parsing both:

  * every `#[repr(C)]` struct of rust/rpt-gpu-sys/src/lib.rs has the C struct's fields, in order, with the
    corresponding type (and therefore the same size and offsets, which test_abi.py checks for the C side);
  * the `extern "C"` block declares every function the header declares, with the same arity and argument /
    return types; constants (error codes, enum values, ABI version) agree;
  * rust/rpt.patch is the output of rust/make_patch.py for the reference checkout when that is present, applies to
    the files it names, and gives a `flatten` to every shape of the closed device set.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rpt_gpu.h")
RUST = os.path.join(ROOT, "rust", "rpt-gpu-sys", "src", "lib.rs")

C_SCALAR = {"double": "f64", "float": "f32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8",
            "int": "c_int", "char": "c_char", "void": "c_void"}


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def c_structs():
    src = strip_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} (\w+);", src, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"(const )?(struct )?(\w+)( ?\*)? ?(.*)$", decl)
            const, ctype, ptr, names = mm.group(1), mm.group(3), mm.group(4), mm.group(5)
            for nm in names.split(","):
                nm = nm.strip()
                arr = re.match(r"(\w+)\[(\w+)\]$", nm)
                base = C_SCALAR.get(ctype, ctype)
                if ptr:
                    rt = ("*const " if const else "*mut ") + base
                    fname = nm
                elif arr:
                    n = arr.group(2)
                    n = {"RPT_K_COUNT": "8"}.get(n, n)
                    rt, fname = "[%s; %s]" % (base, n), arr.group(1)
                else:
                    rt, fname = base, nm
                fields.append((fname, rt))
        out[name] = fields
    return out


def rust_structs():
    src = strip_comments(open(RUST).read())
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n    \}", src, flags=re.S):
        fields = []
        for f in re.finditer(r"pub (\w+): ([^,\n]+),", m.group(2)):
            fields.append((f.group(1), f.group(2).strip()))
        out[m.group(1)] = fields
    return out


def c_functions():
    src = strip_comments(open(HEADER).read())
    src = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\n\s*((?:const )?\w+ ?\*?)\s*(rptgpu_\w+)\s*\((.*?)\)\s*;", src, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        out[name] = (ret, [] if args in ("void", "") else [a.strip() for a in args.split(",")])
    return out


def c_arg_to_rust(arg):
    arg = re.sub(r"\[\w*\]", "*", arg)  # array parameters decay to pointers
    m = re.match(r"(const )?(struct )?(\w+)\s*((?:\*\s*)*)\s*(\w+)?\s*(\*)?$", arg)
    const, ctype, stars, trailing = m.group(1), m.group(3), m.group(4).replace(" ", ""), m.group(6)
    n = len(stars) + (1 if trailing else 0)
    base = C_SCALAR.get(ctype, ctype)
    if n == 0:
        return base
    if n == 1:
        return ("*const " if const else "*mut ") + base
    return "*mut " + ("*const " if const else "*mut ") + base  # out-pointer to a pointer


def rust_functions():
    src = strip_comments(open(RUST).read())
    block = re.search(r'extern "C" \{(.*?)\n    \}', src, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        args = [a.split(":", 1)[1].strip() for a in " ".join(m.group(2).split()).split(",") if ":" in a]
        out[m.group(1)] = ((m.group(3) or "()").strip(), args)
    return out


def test_repr_c_structs_match_the_header():
    c, r = c_structs(), rust_structs()
    assert len(c) >= 12
    for name, fields in c.items():
        assert name in r, "rust/rpt-gpu-sys lacks #[repr(C)] struct %s" % name
        rf = r[name]
        assert [f[0] for f in rf] == [f[0] for f in fields], (name, rf, fields)
        for (fn, ct), (_, rt) in zip(fields, rf):
            assert ct == rt, "%s.%s: header says %s, Rust says %s" % (name, fn, ct, rt)


def test_extern_block_declares_every_entry_point():
    c, r = c_functions(), rust_functions()
    assert len(c) >= 25 and set(c) == set(r), sorted(set(c) ^ set(r))
    for name, (ret, args) in c.items():
        rret, rargs = r[name]
        want_ret = {"int": "c_int", "void": "()", "const char*": "*const c_char", "const char *": "*const c_char"}[ret]
        assert rret == want_ret, (name, ret, rret)
        assert len(args) == len(rargs), (name, args, rargs)
        for a, ra in zip(args, rargs):
            assert c_arg_to_rust(a) == ra, "%s: header argument `%s` vs Rust `%s`" % (name, a, ra)


def test_constants_agree():
    hdr = strip_comments(open(HEADER).read())
    rs = strip_comments(open(RUST).read())
    consts = dict(re.findall(r"\b(RPT(?:GPU)?_[A-Z0-9_]+) = (-?\d+)u?", hdr))
    consts.update(dict(re.findall(r"#define (RPTGPU_[A-Z_]+) (\d+)", hdr)))
    rust = dict(re.findall(r"pub const (RPT(?:GPU)?_[A-Z0-9_]+): \w+ = (-?\d+);", rs))
    assert len(rust) >= 29
    for k, v in rust.items():
        assert consts.get(k) == v, (k, v, consts.get(k))
    for k in ("RPTGPU_ABI_VERSION", "RPTGPU_E_COMM", "RPT_SHAPE_MONOMIAL", "RPT_LIGHT_OBJECT", "RPT_FLAG_PERSISTENT"):
        assert k in rust


def test_patch_covers_the_closed_shape_set_and_is_current():
    patch = open(os.path.join(ROOT, "rust", "rpt.patch")).read()
    files = re.findall(r"^diff -ruN a/(\S+)", patch, flags=re.M)
    for f in ("src/shape.rs", "src/kdtree.rs", "src/renderer.rs", "src/shape/sphere.rs", "src/shape/plane.rs",
              "src/shape/cube.rs", "src/shape/mesh.rs", "src/shape/monomial_surface.rs", "src/rng.rs", "src/gpu.rs",
              "Cargo.toml", "examples/dump_golden.rs"):
        assert f in files, f
    # Sphere, Plane, Cube, Monomial, Transformed<T>, Box / Arc forwarding, KdTree<T> (+ Triangle's MESH override)
    assert patch.count("fn flatten(") >= 9 and patch.count("fn flatten_collection(") == 2
    for kind in ("ShapeDesc::Sphere", "ShapeDesc::Plane", "ShapeDesc::Cube", "ShapeDesc::Monomial", "ShapeDesc::Mesh",
                 "ShapeDesc::Group", "ShapeDesc::Transformed"):
        assert kind in patch, kind
    assert "forbid(unsafe_code)" not in patch  # rpt keeps it: nothing in the patch touches that line
    assert "unsafe" not in "".join(l for l in patch.splitlines() if l.startswith("+") and "//" not in l)
    ref = "/root/reference"
    if os.path.isdir(os.path.join(ref, "src")):
        # the committed patch is what the generator produces, and it applies
        import tempfile
        import shutil
        tmp = tempfile.mkdtemp()
        try:
            shutil.copy(os.path.join(ref, "Cargo.toml"), tmp)
            shutil.copytree(os.path.join(ref, "src"), os.path.join(tmp, "src"))
            os.makedirs(os.path.join(tmp, "examples"))
            r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "rust", "rpt.patch")], cwd=tmp,
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
        finally:
            shutil.rmtree(tmp)


def test_committed_patch_is_what_the_generator_makes_today(tmp_path):
    """rust/rpt.patch regenerated from the reference checkout and rust/rpt_additions/ equals the committed file: an edit
    of gpu.rs / rng.rs / dump_golden.rs (or of the generator) without re-running make_patch.py is caught here, and so is
    a reference whose anchors moved.  (No reference checkout — the GPU box — nothing to regenerate from: skipped.)"""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "src")):
        pytest.skip("no reference checkout here")
    out = str(tmp_path / "rpt.patch")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "rust", "make_patch.py"), ref, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    fresh, committed = open(out).read(), open(os.path.join(ROOT, "rust", "rpt.patch")).read()
    assert fresh == committed, "rust/rpt.patch is stale: run `python rust/make_patch.py /root/reference`"


def test_philox_stream_of_the_patch_is_the_oracles(oracle):
    """rust/rpt_additions/rng.rs restated in Python line by line gives the oracle's draws."""
    src = open(os.path.join(ROOT, "rust", "rpt_additions", "rng.rs")).read()
    assert "0xD251_1F53" in src and "0xCD9E_8D57" in src and "0x9E37_79B9" in src and "0xBB67_AE85" in src
    M = 0xFFFFFFFF

    def philox(ctr, key):
        c0, c1, c2, c3 = ctr
        k0, k1 = key
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
            c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M, p1 & M, ((p0 >> 32) ^ c3 ^ k1) & M, p0 & M
            k0, k1 = (k0 + 0x9E3779B9) & M, (k1 + 0xBB67AE85) & M
        return [c0, c1, c2, c3]

    seed, pixel, sample = 0x0123456789ABCDEF, 1234, (7 << 32) | 5
    for draw in range(6):
        o = philox([pixel, sample & M, sample >> 32, draw >> 1], [seed & M, seed >> 32])
        want = (o[3] << 32 | o[2]) if draw & 1 else (o[1] << 32 | o[0])
        assert oracle.lib().oracle_rng_u64(seed, pixel, sample, draw) == want


def test_struct_sizes_in_the_crates_own_tests_are_the_c_sizes():
    import ctypes as C
    from rpt_amd import _abi
    rs = open(RUST).read()
    found = dict(re.findall(r"assert_eq!\(size_of::<(\w+)>\(\), (\d+)\);", rs))
    assert len(found) >= 12
    for name, size in found.items():
        assert C.sizeof(getattr(_abi, name)) == int(size), name
