"""Static checks of the reference-side binding (julia/ClearSkyHIP.jl) against the C ABI.

No Julia toolchain exists in this pipeline, so the `.jl` file cannot be executed; what CAN be checked mechanically is that every
`ccall((:sym, LIB), ret, (types...), args...)` names a symbol the header declares, with the header's return type, arity and C type
of each argument, and passes exactly as many values as it declares types -- the drift a silent edit of either side would cause.
The same comparison is made for the ctypes table of the executed binding (clearsky.jl_amd/_lib.py:SIGNATURES).
"""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "clearsky_hip.h")
JL = os.path.join(ROOT, "julia", "ClearSkyHIP.jl")
INTEGRATION = os.path.join(ROOT, "INTEGRATION.md")


# ---- the header's prototypes in a canonical form: (ret, [arg, ...]) with arg in {"int", "int64", "double", "int*", "double*", ...}

def _canon_c(decl: str) -> str:
    """canonical type of one C parameter or return declaration (names and const dropped; cs_ctx and void are both opaque)"""
    d = re.sub(r"/\*.*?\*/", " ", decl)
    depth = d.count("*")
    d = d.replace("*", " ")
    words = [w for w in d.split() if w != "const"]
    base = words[0]
    if base in ("cs_ctx", "void"):
        base = "void"
    elif base in ("int", "int32_t"):
        base = "int"
    elif base == "int64_t":
        base = "int64"
    elif base == "int16_t":
        base = "int16"
    return base + "*" * depth


HEADER_DEV = os.path.join(ROOT, "include", "clearsky_hip_dev.h")


def header_prototypes(dev=False):
    """prototypes of the product header; dev=True: of the product AND the laboratory header (clearsky_hip_dev.h: same library)"""
    src = open(HEADER).read() + (open(HEADER_DEV).read() if dev else "")
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b((?:const\s+)?(?:int|void|char)\s*\**)\s*(cs_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3).strip()
        args = [] if params in ("", "void") else [_canon_c(p) for p in params.split(",")]
        protos[name] = (_canon_c(ret + " x") if "*" in ret else _canon_c(ret), args)
    return protos


# ---- Julia ccall expressions

_JL_TYPES = {
    "Cint": "int", "Int32": "int", "Int64": "int64", "Float64": "double", "Cdouble": "double", "Cstring": "char*", "Cvoid": "void",
    "Ptr{Float64}": "double*", "Ptr{Cdouble}": "double*", "Ref{Float64}": "double*", "Ptr{Cint}": "int*", "Ref{Cint}": "int*",
    "Ptr{Int32}": "int*", "Ptr{Int16}": "int16*", "Ptr{Int64}": "int64*", "Ref{Int64}": "int64*", "Ptr{Cvoid}": "void*",
    "Ptr{Ptr{Cvoid}}": "void**", "Ref{Ptr{Cvoid}}": "void**", "Ptr{UInt8}": "char*", "Ptr{Ptr{Float64}}": "double**",
}


def _split_top(s: str):
    """split at top-level commas (parentheses, brackets and braces nest)"""
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    tail = "".join(cur).strip()
    if tail:
        out.append(tail)
    return out


def julia_ccalls(text: str):
    """[(symbol, ret, [argtypes], n_values_passed, line)] for every ccall((:sym, LIB), ...) in `text`"""
    text = "\n".join(ln.split("#", 1)[0] if not ln.lstrip().startswith("#") else "" for ln in text.split("\n"))   # drop comments
    calls = []
    for m in re.finditer(r"ccall\(\(:(cs_\w+),\s*\w+\)", text):
        start = m.start() + len("ccall")
        depth, i = 0, start
        while True:
            ch = text[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        parts = _split_top(text[start + 1:i])
        sym = m.group(1)
        ret = parts[1]
        assert parts[2].startswith("(") and parts[2].endswith(")"), (sym, parts[2])
        types = _split_top(parts[2][1:-1])
        calls.append((sym, ret, types, len(parts) - 3, text.count("\n", 0, m.start()) + 1))
    return calls


def _check_calls(calls, protos, where):
    assert calls, f"no ccall found in {where}"
    for sym, ret, types, nvalues, line in calls:
        assert sym in protos, f"{where}:{line}: ccall of {sym}, which include/clearsky_hip.h does not declare"
        cret, cargs = protos[sym]
        assert ret in _JL_TYPES, f"{where}:{line}: {sym}: unknown Julia return type {ret}"
        assert _JL_TYPES[ret] == cret, f"{where}:{line}: {sym} returns {cret} in the header, the ccall says {ret}"
        assert len(types) == len(cargs), f"{where}:{line}: {sym} takes {len(cargs)} arguments in the header, the ccall declares {len(types)}"
        assert nvalues == len(types), f"{where}:{line}: {sym}: {len(types)} argument types but {nvalues} values passed"
        for k, (jt, ct) in enumerate(zip(types, cargs)):
            assert jt in _JL_TYPES, f"{where}:{line}: {sym} argument {k + 1}: unknown Julia type {jt}"
            assert _JL_TYPES[jt] == ct, f"{where}:{line}: {sym} argument {k + 1} is {ct} in the header, the ccall says {jt}"


def test_header_parses_every_symbol():
    from clearsky_jl_amd import SIGNATURES
    protos = header_prototypes(dev=True)
    assert set(protos) == set(SIGNATURES), (sorted(set(protos) ^ set(SIGNATURES)))
    # product and laboratory are separate headers (round 5): nothing is declared twice, and the lab symbols are the lab's
    pub = header_prototypes()
    lab = set(protos) - set(pub)
    assert lab == {"cs_set_tuning", "cs_set_interp_plan", "cs_set_matrix_cores", "cs_set_merge", "cs_column_profile", "cs_column_counts",
                   "cs_column_info", "cs_column_work", "cs_interp_plan", "cs_phco2_plan", "cs_faddeeva_batch", "cs_devfn_batch"}, sorted(lab)
    assert len(pub) <= 48, len(pub)
    assert protos["cs_fluxes_discretized"][1][:4] == ["void*", "int64", "double*", "int"]
    assert protos["cs_balanced_ranges"][1][4] == "double**"
    assert protos["cs_fluxes_discretized_multi"][1][0] == "void**"


def test_julia_ccalls_match_header():
    protos = header_prototypes()
    calls = julia_ccalls(open(JL).read())
    _check_calls(calls, protos, "julia/ClearSkyHIP.jl")
    used = {c[0] for c in calls}
    # the B2 members travel through these: a binding without them cannot carry baked gases, CIA pairs or accelerated absorbers
    for sym in ("cs_fluxes_discretized", "cs_fluxes_discretized_multi", "cs_fluxes_discretized_members", "cs_bake", "cs_table_upload",
                "cs_table_eval", "cs_cia_begin", "cs_cia_band", "cs_accel_upload", "cs_accel_store", "cs_accel_fetch", "cs_column_setup",
                "cs_column_set_tables", "cs_column_set_cia", "cs_column_batch", "cs_shape_batch", "cs_shape_points", "cs_gas_upload",
                "cs_gas_upload_par"):
        assert sym in used, f"julia/ClearSkyHIP.jl no longer binds {sym}"


def test_integration_md_ccalls_match_header():
    protos = header_prototypes()
    text = open(INTEGRATION).read()
    blocks = re.findall(r"```julia\n(.*?)```", text, flags=re.S)
    inline = re.findall(r"`(ccall\(\(:cs_.*?\))`", text, flags=re.S)
    calls = []
    for b in blocks + inline:
        calls += julia_ccalls(b)
    _check_calls(calls, protos, "INTEGRATION.md")


def test_julia_binding_has_no_commented_out_b2():
    """B2 is code, not a comment block: the methods the reference's callers dispatch to exist in the module body."""
    src = open(JL).read()
    body = src[:src.index("end # module")]
    for needle in ("function monochromaticfluxes!(", "function update!(A::AcceleratedAbsorber", "function ClearSky.UnifiedAbsorber(",
                   "struct HIPGas", "struct HIPCIA", "function tableslot!(ctx::Context, g::Gas", "function accelslot!(",
                   "hipcheckpressures(𝒜, P[end], P[1])", "g isa SemiGrayGas", "function ClearSky.opacityerror(g::HIPGas"):
        assert needle in body, needle
    assert src[src.index("end # module"):].strip() == "end # module", "nothing but the module may follow (the B2 addendum used to be a comment)"


def test_julia_binding_gives_the_caller_the_device_band_fluxes():
    """Round 5: radiate! dispatches on the core (fluxes.jl:357-383 takes it positionally) and fills F.F+, F.F-, F.Fnet from
    the ABI's Fup / Fdn instead of the host's serial intF!; fluxpack = :bands passes C_NULL for tau, M+, M- (no 146 MB copy per call);
    hipfluxes / hipnetfluxes stand in for fluxes / netfluxes, whose `core` keyword cannot dispatch."""
    src = open(JL).read()
    for needle in ("import ClearSky: monochromaticfluxes!, radiate!", "fluxpack::Symbol", "function hipcolumn!(F⁺::Vector{Float64}, F⁻::Vector{Float64}",
                   "function radiate!(F::ClearSky.FluxPack, core::HIPDiscretized", "if core.fluxpack == :bands",
                   "hipcolumn!(F⁺, F⁻, nothing, nothing, nothing, core, P, g, T, μ, 𝒻S, 𝒻a, 𝒜; θₛ=θₛ)", "@. F.Fnet = F.F⁺ - F.F⁻",
                   "function hipfluxes(", "function hipnetfluxes(", "pτ = Ta === nothing ? C_NULL : pointer(Ta)", "hipfluxes, hipnetfluxes"):
        assert needle in src, needle
    # all three flux entry points are called with the nullable output pointers (tau, M+, M-) and the band-flux vectors, in the header's order
    calls = [c for c in re.findall(r"ccall\(\(:cs_fluxes_discretized\w*, LIB\).*?\)\)\n", src, flags=re.S)]
    assert len(calls) == 3
    for c in calls:
        assert re.search(r"nstream, pτ, p⁺, p⁻, F⁺, F⁻\)\)\n$", c), c[-120:]
    # monochromaticfluxes! keeps the reference's contract (fills M+, M-, tau in place) on top of the same call
    body = src[src.index("function monochromaticfluxes!("):src.index("function radiate!(")]
    assert "hipcolumn!(Vector{Float64}(undef, np), Vector{Float64}(undef, np), M⁺, M⁻, τ, core" in body


_CT = {C.c_int: "int", C.c_int64: "int64", C.c_double: "double", C.c_char_p: "char*", C.c_void_p: "void*", None: "void",
       C.POINTER(C.c_double): "double*", C.POINTER(C.c_int): "int*", C.POINTER(C.c_int32): "int*", C.POINTER(C.c_int16): "int16*",
       C.POINTER(C.c_int64): "int64*", C.POINTER(C.c_void_p): "void**", C.POINTER(C.POINTER(C.c_double)): "double**"}


def test_ctypes_signatures_match_header():
    from clearsky_jl_amd import SIGNATURES
    protos = header_prototypes(dev=True)
    for name, (res, args) in SIGNATURES.items():
        cret, cargs = protos[name]
        assert _CT[res] == cret, f"{name}: returns {cret} in the header, ctypes says {_CT[res]}"
        assert len(args) == len(cargs), f"{name}: {len(cargs)} arguments in the header, ctypes declares {len(args)}"
        for k, (a, ct) in enumerate(zip(args, cargs)):
            # (a void* in the header is either an opaque handle or a device pointer: ctypes binds both as c_void_p; `double **dF` of
            #  cs_column_flux_ptr is bound as POINTER(c_void_p))
            got = _CT[a]
            ok = got == ct or (ct == "double**" and got == "void**") or (ct == "double*" and got == "void*")
            assert ok, f"{name} argument {k + 1}: {ct} in the header, ctypes says {got}"


@pytest.mark.parametrize("bad,msg", [
    ("check(ccall((:cs_set_interp, LIB), Cint, (Ptr{Cvoid}, Cint), ctx.handle))", "values passed"),
    ("check(ccall((:cs_set_interp, LIB), Cint, (Ptr{Cvoid}, Float64), ctx.handle, 1))", "argument 2"),
    ("check(ccall((:cs_set_interp, LIB), Cint, (Ptr{Cvoid},), ctx.handle))", "takes 2 arguments"),
    ("check(ccall((:cs_no_such_symbol, LIB), Cint, (Ptr{Cvoid},), ctx.handle))", "does not declare"),
])
def test_checker_catches_drift(bad, msg):
    with pytest.raises(AssertionError, match=msg):
        _check_calls(julia_ccalls(bad), header_prototypes(), "synthetic")
