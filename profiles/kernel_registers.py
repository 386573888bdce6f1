"""Register / scratch / LDS budget of every k_bounce instantiation, from the ISA listing of `make marks` (or `make asm`):
    make -C project3-cuda-path-tracer_amd/csrc marks && python profiles/kernel_registers.py [/tmp/pt_marks/pt_api-hip-amdgcn-amd-amdhsa-gfx950.s]"""
import re, subprocess, sys
f = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pt_marks/pt_api-hip-amdgcn-amd-amdhsa-gfx950.s"
s = open(f).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", s, re.S):
    body = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", body).group(1)
    if "k_bounce" not in name and "k_mesh_walk" not in name: continue
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, body).group(1)
    print("%-62s vgpr %3s sgpr %3s scratch %3s B" % (d.replace("void ptd::", "")[:62], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size")))
