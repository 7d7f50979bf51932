"""CPU: julia/hip_backend.jl (the binding a Juqbox.jl maintainer includes; Julia is not in the image) against
include/juqbox_hip.h -- struct layouts (field order, names, types) and the return / argument types of EVERY ccall."""
import os
import re

from conftest import ROOT

CTYPE = {  # C type -> the Julia types a ccall may use for it
    "int": {"Cint", "Int32"}, "int32_t": {"Int32", "Cint"}, "int64_t": {"Int64"}, "double": {"Float64", "Cdouble"},
    "void": {"Cvoid"},
    "const double *": {"Ptr{Float64}"}, "double *": {"Ptr{Float64}"},
    "const int32_t *": {"Ptr{Int32}"}, "int32_t *": {"Ref{Int32}", "Ptr{Int32}"},
    "jq_handle *": {"Ptr{Cvoid}"}, "const jq_handle *": {"Ptr{Cvoid}"}, "jq_handle **": {"Ref{Ptr{Cvoid}}"},
    "const jq_problem *": {"Ref{JQProblem}"}, "jq_timing *": {"Ref{JQTiming}"},
    "const char *": {"Cstring"}, "void *": {"Ptr{Cvoid}"}, "char *": {"Ptr{UInt8}"},
    "const jq_csc *": {"Ref{JQCsc}", "Ptr{JQCsc}"}, "const int64_t *": {"Ptr{Int64}"}, "int64_t *": {"Ref{Int64}", "Ptr{Int64}"},
}


def _norm(t):
    t = re.sub(r"\s+", " ", t.strip())
    return re.sub(r"\s*\*", " *", t).replace("* *", "**")


def header():
    txt = open(os.path.join(ROOT, "include", "juqbox_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    structs = {}
    for name in ("jq_problem", "jq_timing", "jq_csc"):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), txt, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                if re.match(r"^[\w ]+ \w+(\s*,\s*\w+)+$", decl):      # "int64_t m, n"
                    typ, names = decl.split(" ", 1)[0], decl.split(" ", 1)[1]
                    fields.extend((nm.strip(), _norm(typ)) for nm in names.split(","))
                    continue
                m = re.match(r"(.*?)(\w+)$", decl)
                fields.append((m.group(2), _norm(m.group(1))))
        structs[name] = fields
    protos = {}
    for m in re.finditer(r"^([\w ]+?\*?)\s*(jq_\w+)\s*\(([^;]*?)\);", txt, flags=re.M | re.S):
        ret, name, args = _norm(m.group(1)), m.group(2), m.group(3)
        argt = []
        if args.strip() != "void":
            for a in args.split(","):
                mm = re.match(r"(.*?)(\w+)$", a.strip())
                argt.append(_norm(mm.group(1)))
        protos[name] = (ret, argt)
    return structs, protos


def julia():
    txt = open(os.path.join(ROOT, "julia", "hip_backend.jl")).read()
    txt = re.sub(r"#.*", "", txt)
    structs = {}
    for name in ("JQProblem", "JQTiming", "JQCsc"):
        body = re.search(r"struct %s\s*\n(.*?)\nend" % name, txt, flags=re.S).group(1)
        structs[name] = [tuple(x.strip() for x in ln.split("::")) for ln in body.splitlines() if "::" in ln]
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*libjq\),\s*(\w+),\s*\(", txt):
        i, depth = m.end(), 1
        while depth:                       # the argument-type tuple, with nested braces / parentheses
            depth += {"(": 1, ")": -1}.get(txt[i], 0)
            i += 1
        tup = txt[m.end():i - 1]
        parts, cur, d = [], "", 0
        for ch in tup:
            d += {"{": 1, "}": -1}.get(ch, 0)
            if ch == "," and d == 0:
                parts.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur.strip())
        calls.append((m.group(1), m.group(2), parts))
    return structs, calls


def test_struct_layouts_match_the_header():
    hs, _ = header()
    js, _ = julia()
    for cname, jname in (("jq_problem", "JQProblem"), ("jq_timing", "JQTiming"), ("jq_csc", "JQCsc")):
        cf, jf = hs[cname], js[jname]
        assert [n for n, _ in cf] == [n for n, _ in jf], (cname, "field names / order")
        for (n, ct), (_, jt) in zip(cf, jf):
            assert jt in CTYPE[ct], (cname, n, ct, jt)


def test_every_ccall_matches_its_prototype():
    _, protos = header()
    _, calls = julia()
    assert len(calls) >= 20
    for name, ret, args in calls:
        assert name in protos, "ccall of an undeclared symbol: " + name
        cret, cargs = protos[name]
        assert ret in CTYPE[cret], (name, "return", cret, ret)
        assert len(args) == len(cargs), (name, "argument count", cargs, args)
        for k, (ct, jt) in enumerate(zip(cargs, args)):
            assert jt in CTYPE[ct], (name, k, ct, jt)


def test_the_shim_binds_every_hot_path_entry_point():
    _, protos = header()
    _, calls = julia()
    bound = {c[0] for c in calls}
    assert set(protos) - bound == set(), "header entry points the Julia shim does not bind: %s" % sorted(set(protos) - bound)


def test_full_leakage_weights_are_passed_or_refused_never_truncated():
    """Round-3 review: the shim passed diag(params.wmat_real) whatever the weights were -- a use_custom_forbidden problem
    (src/evalobjgrad.jl:214-232) evaluated to other numbers without an error.  Now: full weights go to jq_update_wmat, the one
    case that is no Hermitian weight is refused, and diag(...) is only taken from a Diagonal."""
    raw = open(os.path.join(ROOT, "julia", "hip_backend.jl")).read()
    txt = re.sub(r"#.*", "", raw)
    assert "full_weights(params) = !(params.wmat_real isa Diagonal) || !iszero(params.wmat_imag)" in txt
    body = re.search(r"function push_weights!\(wa::Working_Arrays_HIP, params\)(.*?)\nend", txt, flags=re.S).group(1)
    full, diagonal = body.split("\n    else\n")
    assert ":jq_update_wmat," in full and "Matrix{Float64}(params.wmat_real)" in full and "Matrix{Float64}(params.wmat_imag)" in full
    assert "error(" in full and "Diagonal wmat_imag" in full
    assert ":jq_update_wmat_diag" in diagonal and "diag(params.wmat_real)" in diagonal
    # no other place reads the diagonal of wmat_real, and the weights are pushed before every evaluation
    assert txt.count("diag(params.wmat_real)") == 1
    assert "push_weights!(wa, params)" in re.search(r"function sync!(.*?)\nend", txt, flags=re.S).group(1)
    # the implicit-midpoint type reads params.wmat (Diagonal by its field type, src/evalobjgrad.jl:90) and says so if it is not
    imr = re.search(r"function push_weights!\(wa::Working_Arrays_M_HIP, params\)(.*?)\nend", txt, flags=re.S).group(1)
    assert "params.wmat isa Diagonal || error(" in imr


def test_sparse_operators_are_passed_in_csc_form():
    """use_sparse = true problems hand colptr / rowval / nzval (1-based Int64) over, nothing is densified"""
    txt = re.sub(r"#.*", "", open(os.path.join(ROOT, "julia", "hip_backend.jl")).read())
    assert "JQCsc(A::SparseMatrixCSC{Float64,Int64}) = JQCsc(size(A, 1), size(A, 2), pointer(A.colptr), pointer(A.rowval), pointer(A.nzval))" in txt
    new = re.search(r"function jq_new_handle(.*?)\nend", txt, flags=re.S).group(1)
    assert "sparse ? zeros(1, 1) : Matrix{Float64}(params.Hconst)" in new
    assert "sparse ? pointer(c0) : Ptr{JQCsc}(C_NULL)" in new
    assert ":jq_update_hconst_csc" in txt


def _julia_tokens(txt):
    """hip_backend.jl without comments and string / char literals, as (line, token) pairs: identifiers / keywords and the brackets"""
    out, i, n, line = [], 0, len(txt), 1
    while i < n:
        c = txt[i]
        if c == "\n":
            line += 1
            i += 1
        elif c == "#":
            if txt.startswith("#=", i):
                j = txt.index("=#", i + 2)
                line += txt.count("\n", i, j)
                i = j + 2
            else:
                while i < n and txt[i] != "\n":
                    i += 1
        elif c == '"':
            q = '"""' if txt.startswith('"""', i) else '"'
            j = i + len(q)
            while not txt.startswith(q, j):
                if txt[j] == "\\":
                    j += 1
                if txt[j] == "$" and txt[j + 1] == "(":      # interpolation: skip to its closing parenthesis (no nested strings in this file)
                    depth, j = 1, j + 2
                    while depth:
                        depth += {"(": 1, ")": -1}.get(txt[j], 0)
                        j += 1
                    continue
                j += 1
            line += txt.count("\n", i, j)
            i = j + len(q)
        elif c == "'" and i + 2 < n and (txt[i + 2] == "'" or (txt[i + 1] == "\\" and txt[i + 3] == "'")):      # char literal (not a transpose)
            i += 3 if txt[i + 2] == "'" else 4
        elif c.isalpha() or c == "_" or c == "@":
            j = i + 1
            while j < n and (txt[j].isalnum() or txt[j] in "_!"):
                j += 1
            # (a field access `x.end` or a symbol `:end` is not a keyword)
            out.append((line, txt[i:j] if not (i and txt[i - 1] in ".:") else "ident"))
            i = j
        else:
            if c in "()[]{}":
                out.append((line, c))
            i += 1
    return out


def test_julia_binding_blocks_and_brackets_balance():
    """No Julia in the image, so the binding is never parsed by the real thing (review: the regex checks above would not notice a
    syntax slip outside a ccall).  What CAN be checked without an interpreter: every bracket closes in order, and every block opener
    (function, if, for, while, let, try, begin, do, struct, module, quote, macro -- outside brackets, where `for` / `if` are
    comprehension / generator syntax) has its `end`, with `end` inside [] being an index.  A missing or extra `end` / bracket -- the
    usual result of an edit that was never run -- fails here with the line of the opener that is left over."""
    txt = open(os.path.join(ROOT, "julia", "hip_backend.jl")).read()
    openers = {"function", "if", "for", "while", "let", "try", "begin", "do", "struct", "module", "quote", "macro", "baremodule"}
    pairs = {")": "(", "]": "[", "}": "{"}
    brackets, blocks = [], []
    toks = _julia_tokens(txt)
    for k, (line, t) in enumerate(toks):
        if t in "([{":
            brackets.append((t, line))
        elif t in pairs:
            assert brackets and brackets[-1][0] == pairs[t], "line %d: '%s' closes nothing (open: %s)" % (line, t, brackets[-3:])
            brackets.pop()
        elif t == "end":
            if any(b == "[" for b, _ in brackets):      # a[end]: an index
                continue
            assert blocks, "line %d: `end` without an open block" % line
            blocks.pop()
        elif t in openers:
            if t in ("for", "if") and brackets:          # comprehension / generator inside brackets
                continue
            if t == "struct" and k and toks[k - 1][1] == "mutable":
                pass
            blocks.append((t, line, len(brackets)))
        elif t == "abstract" or t == "primitive":
            blocks.append((t, line, len(brackets)))     # `abstract type ... end`
    assert not brackets, "unclosed brackets: %s" % brackets[:5]
    assert not blocks, "blocks without `end`: %s" % [(t, ln) for t, ln, _ in blocks[:5]]
    # ... and the checker itself notices a slip: drop the last `end` of the file
    broken = txt[:txt.rindex("end")]
    depth = 0
    for _, t in _julia_tokens(broken):
        depth += (t in openers) - (t == "end")
    assert depth != 0
