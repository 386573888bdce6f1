"""Static issue-cycle estimate per phase of k_bounce from the marked listing (make marks), priced with profiles/valu_issue_rate.json's classes:
F (VGPR operands) 2.3 cycles per SIMD, H (an SGPR / VCC operand or result, v_cmp, integer multiplies, division helpers, conversions) 2.3 + share * 1.85
(the penalty falls on an H that follows an H of ANY wave of the SIMD: the class's share of the kernel prices it), P (packed fp32) 4.2, T (rcp / sqrt / rsq) 8.1.
    python profiles/phase_cycles.py [listing] [kernel-name-substring]
Static: every branch of a phase counted once -- a lead, not a measurement."""
import collections, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pt_marks/pt_api-hip-amdgcn-amd-amdhsa-gfx950.s"
want = sys.argv[2] if len(sys.argv) > 2 else "k_bounceILb0ELb0ELb0ELb0ELb1E"
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN3ptk8") and want in l and ": ; @" in l][0]
def cls(t):
    op = t.split()[0]
    ops = t[len(op):].split(";")[0]
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log")): return "T"
    if op.startswith("v_pk_"): return "P"
    if re.search(r"\bs\d+|\bs\[|vcc|exec", ops) or op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_mul_lo", "v_mul_hi", "v_mad_u64", "v_div_", "v_cvt_", "v_lshl_add_u64", "v_mbcnt")): return "H"
    return "F"
seg, cur = collections.OrderedDict(), "entry"
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"): break
    m = re.search(r"PTMARK (\d+)", l)
    if m:
        cur = "mark %s" % m.group(1); k = 2
        while cur in seg: cur = "mark %s (#%d)" % (m.group(1), k); k += 1
        continue
    t = l.strip()
    if not l.startswith("\t") or not t.startswith("v_"): continue
    seg.setdefault(cur, collections.Counter())[cls(t)] += 1
tot = collections.Counter()
for c in seg.values(): tot.update(c)
share = tot["H"] / max(sum(tot.values()), 1)
cost = {"F": 2.3, "H": 2.3 + share * 1.85, "P": 4.2, "T": 8.1}
print("%s: %d vector instructions, H share %.2f -> an H costs %.2f cycles" % (want, sum(tot.values()), share, cost["H"]))
for k, c in seg.items():
    cyc = sum(cost[x] * n for x, n in c.items())
    print("%-16s %4d instr  F %4d H %4d P %3d T %3d  ~%5.0f cycles" % (k, sum(c.values()), c["F"], c["H"], c["P"], c["T"], cyc))
