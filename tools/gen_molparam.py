#!/usr/bin/env python3
"""Extract the numeric MOLPARAM table (reference src/hitran/molparam.jl:1-1621) into a JSON data file.

Only numbers and identifiers are extracted (molecule number, formula, per-isotopologue abundance,
molar mass, Qref, Chebyshev-fit metadata and coefficients) -- the table is data produced by the reference's
scripts/molparam.py from hitran.org TIPS and must be used verbatim for parity.
Run in the build container only (needs /root/reference):  python tools/gen_molparam.py
"""
import json, re, sys, pathlib

SRC = pathlib.Path("/root/reference/src/hitran/molparam.jl")
OUT = pathlib.Path(__file__).resolve().parent.parent / "clearsky.jl_amd" / "data" / "molparam.json"

def split_blocks(txt):
    # every entry starts with "  MolParam(" at fixed indentation
    parts = re.split(r"\n  MolParam\(", txt)
    return parts[1:]

def arr(kind, blk, start=0):
    m = re.compile(re.escape(kind) + r"\[(.*?)\]", re.S).search(blk, start)
    return m.group(1), m.end()

def main():
    txt = SRC.read_text()
    tmin = float(re.search(r"const TMIN = ([0-9.eE+-]+)", txt).group(1))
    tmax = float(re.search(r"const TMAX = ([0-9.eE+-]+)", txt).group(1))
    mols = []
    for blk in split_blocks(txt):
        if blk.strip().startswith(")"):  # MolParam() empty placeholder
            mols.append(None); continue
        m = re.search(r"#1, molecule number\s*\n\s*(-?\d+),", blk)
        if not m:
            mols.append(None); continue
        M = int(m.group(1))
        formula = re.search(r'#2, molecule formula\s*\n\s*"([^"]*)"', blk).group(1)
        name = re.search(r'#3, molecule name\s*\n\s*"([^"]*)"', blk).group(1)
        pos = blk.index("#4,")
        I, pos = arr("Int64", blk, pos)
        isoform, pos = arr("String", blk, pos)
        afgl, pos = arr("Int64", blk, pos)
        A, pos = arr("Float64", blk, pos)
        mu, pos = arr("Float64", blk, pos)
        qref, pos = arr("Float64", blk, pos)
        hascheb, pos = arr("Bool", blk, pos)
        ncheb, pos = arr("Int64", blk, pos)
        maxrel, pos = arr("Float64", blk, pos)
        pos = blk.index("Vector{Float64}[", pos) + len("Vector{Float64}[")
        cheb = []
        for mm in re.finditer(r"Float64\[(.*?)\]", blk[pos:], re.S):
            s = mm.group(1).strip()
            cheb.append([float(v) for v in s.split(",")] if s else [])
        nums = lambda s, f: [f(v) for v in s.split(",")] if s.strip() else []
        mols.append(dict(
            M=M, formula=formula, name=name,
            I=nums(I, int), isoform=[v.strip().strip('"') for v in isoform.split(",")] if isoform.strip() else [],
            AFGL=nums(afgl, int), A=nums(A, float), mu=nums(mu, float), Qref=nums(qref, float),
            hascheb=[v.strip() == "true" for v in hascheb.split(",")] if hascheb.strip() else [],
            ncheb=nums(ncheb, int), maxrelerr=nums(maxrel, float), cheb=cheb))
    out = dict(TMIN=tmin, TMAX=tmax, source="markmbaum/ClearSky.jl src/hitran/molparam.jl (data table)",
               molecules=mols)
    OUT.write_text(json.dumps(out, indent=None, separators=(",", ":")))
    n = sum(1 for m in mols if m)
    print("wrote", OUT, "molecules:", n, "slots:", len(mols))

if __name__ == "__main__":
    main()
