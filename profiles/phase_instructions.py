"""Static instruction counts per phase of k_bounce, from a build whose probe() calls are marks in the ISA listing:
    make -C project3-cuda-path-tracer_amd/csrc marks
    python profiles/phase_instructions.py /tmp/pt_marks/pt_api-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-name-substring]
Counts are per wave and per pass through the phase, in listing order (the compiler lays the blocks out roughly in source order;
a mark inside a loop body counts one trip).  Marks: 14 tile start, 15 nearest-hit loop, 0/1/2 box transform / slabs / hit, 3 sphere
cull, 5/6/4 sphere transform / roots / hit, 16 shading, 9 a hit, 10 scatter, 11 hemisphere, 12 ball certificate, 13 wall certificate,
17 next tile's loads, 18 compaction, 19 stores, 20 tile done."""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else "k_bounceILb0ELb0ELb0ELb0E"
start = [i for i, l in enumerate(lines) if l.startswith("_ZN3ptk") and want in l and ": ; @" in l][0]
seg, cur = collections.OrderedDict(), "entry"
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"):
        break
    m = re.search(r"PTMARK (\d+)", l)
    if m:
        cur = "after mark %s" % m.group(1)
        k = 2
        while cur in seg:
            cur = "after mark %s (#%d)" % (m.group(1), k); k += 1
        continue
    t = l.strip()
    if not l.startswith("\t") or t.startswith((".", ";")) or not t:
        continue
    op = t.split()[0]
    c = seg.setdefault(cur, collections.Counter())
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"
    c[kind] += 1
    if op in ("v_rcp_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32"):
        c["slow"] += 1
    if op.startswith("v_mov_b32"):
        c["v_mov"] += 1
    if op.startswith("v_cndmask"):
        c["cndmask"] += 1
tot = collections.Counter()
for k, c in seg.items():
    tot.update(c)
    print("%-22s valu %4d (v_mov %3d, cndmask %3d, quarter-rate/division %3d)  salu %4d  lds %3d  vmem %3d" % (k, c["valu"], c["v_mov"], c["cndmask"], c["slow"], c["salu"], c["lds"], c["vmem"]))
print("%-22s valu %4d (v_mov %3d, cndmask %3d, quarter-rate/division %3d)  salu %4d  lds %3d  vmem %3d" % ("total", tot["valu"], tot["v_mov"], tot["cndmask"], tot["slow"], tot["salu"], tot["lds"], tot["vmem"]))
