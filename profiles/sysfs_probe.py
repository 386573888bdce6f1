import glob, os, subprocess, torch
p = torch.cuda.get_device_properties(0)
print("props", p.name, getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None), getattr(p, "pci_domain_id", None), getattr(p,"clock_rate",None), getattr(p,"memory_clock_rate",None))
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        v = open(d + "/vendor").read().strip()
    except Exception as e:
        v = repr(e)
    print(d, os.path.realpath(d), v)
    for f in ["pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent", "current_link_speed", "unique_id"]:
        try:
            print("  ", f, open(d + "/" + f).read().strip().replace("\n", " | ")[:200])
        except Exception as e:
            print("  ", f, "ERR", e)
    for h in glob.glob(d + "/hwmon/hwmon*"):
        for f in sorted(os.listdir(h)):
            if f.startswith(("power", "temp", "freq")) and (f.endswith(("_input", "_average", "_cap", "_label", "_cap_max"))):
                try:
                    print("  ", f, open(h + "/" + f).read().strip())
                except Exception as e:
                    print("  ", f, "ERR", e)
for cmd in (["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], ["amd-smi", "metric", "--json"]):
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        print(cmd, r.returncode, r.stdout[:1500], r.stderr[:300])
    except Exception as e:
        print(cmd, "ERR", e)
