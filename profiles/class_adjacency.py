"""Static count of the vector instructions that issue at half rate when the SIMD's previous issue was one of their own kind
(profiles/valu_issue_rate.json: an SGPR / VCC / EXEC operand or result -- the scalar-path geometry, v_cmp, v_cndmask --, integer multiplies, the
division helpers), per phase of k_bounce, from the marked listing.  The penalty is the SIMD's (it shows with eight waves resident), so with waves
out of step it is the class's SHARE that prices it (share^2 of the issues pay); the adjacency in program order is printed as well:

    make -C project3-cuda-path-tracer_amd/csrc marks && python profiles/class_adjacency.py [/tmp/pt_marks/pt_api-hip-amdgcn-amd-amdhsa-gfx950.s]

F = full rate (VGPR operands, inline constants), H = the half-rate class, T = transcendental (rcp / sqrt / rsq: a quarter rate).
Static: every branch of a phase is counted once, whatever its frequency -- a lead, not a measurement."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pt_marks/pt_api-hip-amdgcn-amd-amdhsa-gfx950.s"
text = open(path).read()
NAMES = {"0000": "k_bounce<false,false,false,false>", "1000": "k_bounce<true,false,false,false>"}


def cls(line):
    line = line.split(";")[0].strip()
    if not line or line.startswith(".") or line.endswith(":"):
        return "-"
    op = line.split()[0]
    if not op.startswith("v_"):
        return "x"
    ops = line[len(op):]
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log")):
        return "T"
    if re.search(r"\bs\d+|\bs\[|vcc|exec", ops) or op.startswith(("v_cmp", "v_cndmask", "v_readlane", "v_readfirstlane", "v_mul_lo", "v_mul_hi",
                                                                    "v_mad_u64", "v_div_")):
        return "H"
    return "F"


for key, title in NAMES.items():
    sym = "_ZN3ptk8k_bounceILb%sELb%sELb%sELb%sEEEvNS_10BounceArgsE" % tuple(key)
    a = text.index(sym + ":")
    lines = text[a:text.index("s_endpgm", a)].splitlines()
    v = [c for c in map(cls, lines) if c in "HFT"]
    hh = sum(1 for k in range(1, len(v)) if v[k] == "H" and v[k - 1] == "H")
    cyc = lambda pen: 2.4 * v.count("F") + 2.4 * (v.count("H") - (hh if pen else 0)) + 4.2 * (hh if pen else 0) + 8.1 * v.count("T")
    print("%s: %d vector instructions, %d full rate, %d of the half-rate class (%.0f %%), %d transcendental; %d of the half-rate class follow one of "
          "their own (%.0f %%): %.0f cycles as listed, %.0f if the classes alternated (-%.0f %%)"
          % (title, len(v), v.count("F"), v.count("H"), 100.0 * v.count("H") / len(v), v.count("T"), hh, 100.0 * hh / max(v.count("H"), 1),
             cyc(True), cyc(False), 100.0 * (1 - cyc(False) / cyc(True))))
    marks = [(i, int(m.group(1))) for i, l in enumerate(lines) for m in [re.search(r"PTMARK (\d+)", l)] if m]
    for (i0, m0), (i1, m1) in zip(marks, marks[1:]):
        w = [c for c in map(cls, lines[i0:i1]) if c in "HFT"]
        if len(w) < 20:
            continue
        whh = sum(1 for k in range(1, len(w)) if w[k] == "H" and w[k - 1] == "H")
        print("  mark %2d -> %2d: %4d  F %4d  H %4d  T %3d  H after H %4d   %s" % (m0, m1, len(w), w.count("F"), w.count("H"), w.count("T"), whh, "".join(w)[:100]))
