"""Mechanical substitute for `rustc` (no Rust toolchain in the image): the `extern "C"` block of bindings/rust/src/lib.rs
against include/zk_amd.h -- every function the shim declares must exist in the header with the same arity, and every
parameter / return type must have the same width, signedness, pointer depth and constness.  Plus structural checks of
the items the reference's own tests need (sumcheck/src/lib.rs:53-122): Clone / PartialEq / Debug on the polynomial
types, SumcheckVerifier::{verify, verify_partial}, SubClaim, Transcript, fft_internal, index_pair, and one cached
context per field (a context per call breaks ProductPoly::new for k >= 2)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")
HEADER = os.path.join(ROOT, "include", "zk_amd.h")

C_BASE = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "size_t": "usize", "char": "c_char",
          "double": "f64", "void": "c_void", "int": "i32"}


def c_prototypes():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(zk_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        params = [] if args.strip() in ("", "void") else [c_type(a) for a in split_args(args)]
        protos[name] = (c_type(ret + " _"), params)
    return protos


def split_args(args):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [a.strip() for a in out]


def c_type(decl):
    """'const uint64_t *name' / 'uint64_t out[4]' / 'zk_mle *const *f' -> canonical ('u64', ['const'])-style tuple:
    (base, [constness of each pointee level from the outside in])."""
    decl = decl.strip()
    array = bool(re.search(r"\[[^\]]*\]\s*$", decl))
    decl = re.sub(r"\[[^\]]*\]\s*$", "", decl).strip()
    toks = re.findall(r"[A-Za-z_]\w*|\*", decl)
    # drop the parameter name: the last identifier, unless the declaration is just a type (function return "_")
    if toks and toks[-1] != "*" and toks[-1] != "const":
        toks = toks[:-1]
    base, const_base, levels, i = None, False, [], 0
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            const_base = True
        elif toks[i] not in ("struct", "enum", "unsigned", "signed"):
            base = toks[i]
        i += 1
    # levels: each '*' optionally followed by 'const' (constness of that pointer object, i.e. of the next-outer pointee)
    quals = [const_base]
    while i < len(toks):
        if toks[i] == "*":
            quals.append(False)
        elif toks[i] == "const":
            quals[-1] = True
        i += 1
    n_ptr = len(quals) - 1
    if array:
        n_ptr += 1
        quals.append(False)
    # pointee constness from the outside in: pointer level j (outermost = n_ptr) points at an object with quals[j-1]
    pointee = [quals[j - 1] for j in range(n_ptr, 0, -1)]
    return (C_BASE.get(base, base), tuple("const" if q else "mut" for q in pointee))


def rust_type(t):
    t = t.strip()
    levels = []
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+", t)
        levels.append(m.group(1))
        t = t[m.end():]
    return (t, tuple(levels))


def rust_externs():
    text = open(SHIM).read()
    text = re.sub(r"//[^\n]*", "", text)
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S).group(1)
    fns = {}
    for m in re.finditer(r"fn\s+(zk_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
        params = [rust_type(a.split(":", 1)[1]) for a in split_args(args) if a.strip()]
        fns[name] = (rust_type(ret), params)
    return fns


def test_every_extern_matches_the_header():
    c, r = c_prototypes(), rust_externs()
    assert len(r) == len(re.findall(r"^    fn zk_", open(SHIM).read(), flags=re.M)) >= 30, sorted(r)
    problems = []
    for name, (rret, rparams) in sorted(r.items()):
        if name not in c:
            problems.append(f"{name}: not declared in include/zk_amd.h")
            continue
        cret, cparams = c[name]
        if rret != cret:
            problems.append(f"{name}: return {rret} vs C {cret}")
        if len(rparams) != len(cparams):
            problems.append(f"{name}: {len(rparams)} parameters vs C {len(cparams)}")
            continue
        for i, (rp, cp) in enumerate(zip(rparams, cparams)):
            if rp != cp:
                problems.append(f"{name}: parameter {i}: {rp} vs C {cp}")
    assert not problems, "\n".join(problems)


def test_parser_self_check():
    """the C-side normaliser on the declaration shapes the header uses"""
    assert c_type("const uint64_t *evals") == ("u64", ("const",))
    assert c_type("uint64_t out[4]") == ("u64", ("mut",))
    assert c_type("const uint64_t sum[4]") == ("u64", ("const",))
    assert c_type("zk_mle **out") == ("zk_mle", ("mut", "mut"))
    assert c_type("const zk_mle *const *factors") == ("zk_mle", ("const", "const"))
    assert c_type("zk_mle *const *factors") == ("zk_mle", ("const", "mut"))
    assert c_type("const char * _") == ("c_char", ("const",))
    assert c_type("int32_t _") == ("i32", ())
    assert rust_type("*const *mut zk_mle") == ("zk_mle", ("const", "mut"))
    assert rust_type("*mut *mut zk_ctx") == ("zk_ctx", ("mut", "mut"))


def test_items_the_reference_tests_need_exist():
    src = open(SHIM).read()
    code = re.sub(r"//[^\n]*", "", src)
    for needle in [
        r"impl<F: GpuField> Clone for MultiLinearPolynomial<F>",
        r"impl<F: GpuField> PartialEq for MultiLinearPolynomial<F>",
        r"impl<F: GpuField> fmt::Debug for MultiLinearPolynomial<F>",
        r"#\[derive\(Clone, Debug, PartialEq\)\]\s*pub struct ProductPoly",
        r"pub struct SumcheckVerifier<F: GpuField>",
        r"pub fn verify\(poly: ProductPoly<F>, proof: SumcheckProof<F>\) -> Result<bool, &'static str>",
        r"pub fn verify_partial\(proof: SumcheckProof<F>\) -> Result<SubClaim<F>, &'static str>",
        r"pub struct SubClaim<F: PrimeField> \{ pub sum: F, pub challenges: Vec<F> \}",
        r"pub struct Transcript",
        r"pub fn sample_field_element<F: GpuField>\(&mut self\) -> F",
        r"pub fn sample_n_field_elements<F: GpuField>\(&mut self, n: usize\) -> Vec<F>",
        r"pub fn fft_internal<F: GpuField>\(values: Vec<F>, omega: F\) -> Vec<F>",
        r"pub fn index_pair\(n_vars: u8, index: u8\) -> impl Iterator<Item = \(usize, usize\)>",
        r"pub const fn mask\(n: u8\) -> usize",                       # pairing_index.rs:24-26 (public in the reference)
        r"pub fn evaluation_slice\(&self\) -> &\[F\]",
        r"pub fn prove\(poly: ProductPoly<F>, sum: F\) -> Result<SumcheckProof<F>, &'static str>",
        r"pub fn prove_partial\(poly: ProductPoly<F>, sum: F\) -> Result<\(SumcheckProof<F>, Vec<F>\), &'static str>",
    ]:
        assert re.search(needle, code), f"missing in the shim: {needle}"


def test_one_cached_context_per_field():
    code = re.sub(r"//[^\n]*", "", open(SHIM).read())
    # zk_ctx_create is called in exactly one place (the cache), contexts are destroyed, and nothing else creates one
    assert len(re.findall(r"\bzk_ctx_create\(", code)) == 2        # the extern declaration + the one call in ctx::<F>()
    assert "thread_local!" in code and "zk_ctx_destroy(self.raw)" in code
    assert re.search(r"fn ctx<F: GpuField>\(\) -> Result<Rc<Ctx>, &'static str>", code)
    # every call site goes through the cache
    assert len(re.findall(r"ctx::<F>\(\)", code)) >= 5


def test_layout_and_abi_invariants_are_in_the_source():
    """`limbs()` casts *const F to *const u64: the shim must assert size 32 / alignment 8 at compile time for every field it
    binds, and check the library's ABI revision (equal to the header's) before the first context is made."""
    code = re.sub(r"//[^\n]*", "", open(SHIM).read())
    fields = re.findall(r"impl GpuField for (\w+::Fr)", code)
    assert len(fields) == 3
    for f in fields:
        pat = (r"const _: \(\) = assert!\(std::mem::size_of::<%s>\(\) == 32 && std::mem::align_of::<%s>\(\) == 8\);"
               % (re.escape(f), re.escape(f)))
        assert re.search(pat, code), f"no compile-time layout assertion for {f}"
    header_ver = int(re.search(r"#define ZK_AMD_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    shim_ver = int(re.search(r"const ZK_AMD_ABI_VERSION: i32 = (\d+);", code).group(1))
    assert shim_ver == header_ver
    ctx_fn = code[code.index("fn ctx<F: GpuField>()"):]
    ctx_fn = ctx_fn[:ctx_fn.index("zk_ctx_create(")]
    assert "zk_abi_version() } != ZK_AMD_ABI_VERSION" in ctx_fn, "ABI version must be checked before zk_ctx_create"


def test_verifier_passes_each_round_at_its_own_length():
    """verifier.rs:55-58 interpolates every round polynomial at its own length: the shim must not reject ragged proofs"""
    code = re.sub(r"//[^\n]*", "", open(SHIM).read())
    body = code[code.index("impl<F: GpuField> SumcheckVerifier<F>"):]
    body = body[:body.index("pub fn fft<F: GpuField>")]
    assert "zk_sumcheck_verify_lengths(" in body and "zk_sumcheck_verify_partial_lengths(" in body
    assert "r.len() != ns" not in body and "ZK_ERR_VERIFY_SUM" not in body


def test_mask_is_exported_with_the_reference_body():
    """pairing_index.rs:24-26 `pub const fn mask(n: u8) -> usize { (1 << n) - 1 }`: the shim exports it with that body, index_pair
    uses it, and the C++ / Python mirrors agree on the reference's doc examples (mask(1) -> 1, mask(3) -> 0b111)."""
    code = re.sub(r"//[^\n]*", "", open(SHIM).read())
    m = re.search(r"pub const fn mask\(n: u8\) -> usize \{\s*\(1 << n\) - 1\s*\}", code)
    assert m, "mask must be `(1 << n) - 1`"
    body = code[code.index("pub fn index_pair("):]
    body = body[:body.index("pub struct MultiLinearPolynomial")]
    assert "mask(pos)" in body
    hpp = open(os.path.join(ROOT, "zk_amd", "host", "zk.hpp")).read()
    assert re.search(r"inline size_t mask\(uint8_t n\)", hpp)
    import zk_amd

    assert [zk_amd.mask(n) for n in (0, 1, 3, 8)] == [0, 1, 0b111, 255]
    assert zk_amd.index_pair(3, 1) == [(0, 2), (1, 3), (4, 6), (5, 7)]       # pairing_index.rs:70-77
    with pytest.raises(zk_amd.ZkError):
        zk_amd.index_pair(0, 0)


def test_index_pair_formula_matches_the_reference_kats():
    """the shim's index_pair body, transliterated, against the literal lists of pairing_index.rs:61-96"""
    def index_pair(n_vars, index):
        pos = n_vars - 1 - index
        return [(((j >> pos) << (pos + 1)) | (j & ((1 << pos) - 1)), (((j >> pos) << (pos + 1)) | (j & ((1 << pos) - 1))) | (1 << pos))
                for j in range(1 << (n_vars - 1))]

    assert index_pair(3, 0) == [(0, 4), (1, 5), (2, 6), (3, 7)]
    assert index_pair(3, 1) == [(0, 2), (1, 3), (4, 6), (5, 7)]
    assert index_pair(3, 2) == [(0, 1), (2, 3), (4, 5), (6, 7)]
    assert index_pair(2, 0) == [(0, 2), (1, 3)]
    assert index_pair(2, 1) == [(0, 1), (2, 3)]
    assert index_pair(1, 0) == [(0, 1)]
    src = re.sub(r"\s+", " ", open(SHIM).read())
    assert "let pos = n_vars - 1 - index;" in src
    assert "let left = ((j >> pos) << (pos + 1)) | (j & mask(pos));" in src   # mask(pos) = (1 << pos) - 1
    assert "(left, left | (1usize << pos))" in src


def test_delimiters_balance():
    """No rustc here: at least every (), [] and {} of the shim closes in order (strings, chars, comments and lifetimes skipped)."""
    src = open(SHIM).read()
    stack, i, n = [], 0, len(src)
    pairs = {")": "(", "]": "[", "}": "{"}
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            i = src.index("\n", i) if "\n" in src[i:] else n
            continue
        if src.startswith("/*", i):
            i = src.index("*/", i) + 2
            continue
        if c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
            i += 1
            continue
        if c == "'":
            # a char literal ('x', '\n') or a lifetime ('static, 'a): a literal closes within four characters
            close = src.find("'", i + 1, i + 5)
            if close != -1 and (close == i + 2 or src[i + 1] == "\\"):
                i = close + 1
            else:
                i += 1
            continue
        if c in "([{":
            stack.append((c, src.count("\n", 0, i) + 1))
        elif c in ")]}":
            assert stack and stack[-1][0] == pairs[c], f"unbalanced {c!r} at line {src.count(chr(10), 0, i) + 1}"
            stack.pop()
        i += 1
    assert not stack, f"unclosed {stack[-1][0]!r} opened at line {stack[-1][1]}"
