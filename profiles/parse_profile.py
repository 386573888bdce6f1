#!/usr/bin/env python3
"""Distil gpurun_out/prof_<tag>/ (rocprofv3 output of profiles/run_profile.sh) into profiles/:

    profiles/<tag>_kernel_stats.csv      verbatim rocprofv3 --kernel-trace --stats summary
    profiles/<tag>_one_iteration.txt     per-dispatch durations of one iteration (per-bounce)
    profiles/<tag>_pmc_summary.json      PMC counters per kernel, per dispatch
    profiles/pmc_traffic.json            HBM bytes / VALU instructions per launch and per iteration of its batch, read by bench.py

HBM traffic per launch = FETCH_SIZE * 1024 * read_factor + WRITE_SIZE * 1024 (both counters are in KiB).
read_factor = 2 on gfx950 for coalesced streams (MI355X_MICROARCH.md, section HBM).  It was CALIBRATED on
this access pattern (4 B per lane SoA reads) with the round-1 builds that still had a separate ray-generation
kernel: its first bounce launch read exactly 44 B x P (factor measured 1.88 / 1.99) and the ray-generation
kernel wrote exactly 44 B x P (WRITE_SIZE factor 0.994 / 1.000) -- profiles/r01_pmc_summary.json,
profiles/r01b_pmc_summary.json.  Builds with the fused first bounce have no launch with a known byte count,
so the calibrated factors are reused.

    python profiles/parse_profile.py <tag> [--pixels 921600]
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def inst(name):
    """instantiation of the bounce kernel: k_bounce<FIRST, MANY, DOF>"""
    import re
    m = re.search(r"k_bounce<([^>]*)>", name)
    if m:
        return "k_bounce<%s>" % m.group(1)
    m = re.search(r"k_bounceILb([01])ELb([01])E(?:Lb([01])E)?(?:Lb([01])E)?", name)
    if m:
        return "k_bounce<%s>" % ", ".join("true" if g == "1" else "false" for g in m.groups() if g is not None)
    return None


def kind(name):
    if "k_bounce" in name:
        return "k_bounce"
    if "k_mesh_walk" in name:      # scenes with meshes: the walks of a bounce's rays, launched right before it (round 5)
        return "k_mesh_walk"
    if "k_generate_rays" in name:
        return "k_generate_rays"
    if "k_commit" in name:
        return "k_commit"
    if "k_to_rgba8" in name:
        return "k_to_rgba8"
    return "other"


def load_pmc(d):
    files = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)   # gpurun MERGES runs: newest
    return list(csv.DictReader(open(files[-1]))) if files else []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--pixels", type=int, default=1280 * 720)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--set-default", action="store_true", help="also write profiles/pmc_traffic.json (what bench.py reads): the headline configuration only")
    ap.add_argument("--config-key", default=None, help="also store the distilled counters under this key of profiles/pmc_configs.json (bench.py --pmc-key: "
                                                       "the `configs` blocks of the default run -- c4, c5_one_gpu, mesh)")
    args = ap.parse_args()
    src = os.path.join(ROOT, "gpurun_out", "prof_" + args.tag)
    out = {}

    stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]
    if stats:
        shutil.copy(stats[0], os.path.join(HERE, args.tag + "_kernel_stats.csv"))
        txt = subprocess.run([sys.executable, os.path.join(HERE, "trace_summary.py"), os.path.join(src, "trace")],
                             capture_output=True, text=True).stdout
        open(os.path.join(HERE, args.tag + "_one_iteration.txt"), "w").write(txt)
        tot = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(stats[0])):
            k = kind(r["Name"])
            if k != "other":     # both k_bounce<true> and k_bounce<false> count as k_bounce
                tot[k][0] += float(r["TotalDurationNs"])
                tot[k][1] += int(r["Calls"])
            if inst(r["Name"]):
                out.setdefault("instantiations", {})[inst(r["Name"])] = {"trace_avg_ns": float(r["AverageNs"]), "trace_calls": int(r["Calls"])}
        for k, (ns, calls) in tot.items():
            out.setdefault(k, {})["trace_avg_ns"] = ns / max(calls, 1)
            out[k]["trace_calls"] = calls

    per_dispatch = collections.defaultdict(dict)   # (pass, dispatch id) -> counters
    for p in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
        rows = load_pmc(os.path.join(src, p))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for r in rows:
            k = kind(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
            if inst(r["Kernel_Name"]):
                ii = out.setdefault("instantiations", {}).setdefault(inst(r["Kernel_Name"]), {})
                acc = ii.setdefault("_pmc_sum", {})
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                cnt = ii.setdefault("_pmc_n", {})
                cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0) + 1
                for col in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size"):
                    if col in r:
                        ii[col] = r[col]
            per_dispatch[(p, int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
            per_dispatch[(p, int(r["Dispatch_Id"]))]["_kind"] = k
        for k in agg:
            if k == "other":
                continue
            for c, v in agg[k].items():
                out.setdefault(k, {}).setdefault("pmc_per_dispatch", {})[c] = v / n[(k, c)]
                out[k].setdefault("pmc_dispatches", {})[c] = n[(k, c)]

    # ---- calibration of FETCH_SIZE / WRITE_SIZE on known byte counts -------------------------------------
    known = 44.0 * args.pixels
    fetch = sorted((d, v) for (p, d), v in per_dispatch.items() if p == "pmc_fetch")
    first_bounce, prev = [], None
    for d, v in fetch:
        if v["_kind"] == "k_bounce" and prev == "k_generate_rays":
            first_bounce.append(v["FETCH_SIZE"] * 1024.0)
        prev = v["_kind"]
    cal = {}
    if first_bounce:
        counted = sum(first_bounce) / len(first_bounce)
        cal["first_bounce_fetch_counted_bytes"] = counted
        cal["first_bounce_known_read_bytes"] = known
        cal["read_factor"] = known / counted
    wr = [v["WRITE_SIZE"] * 1024.0 for (p, d), v in per_dispatch.items() if p == "pmc_write" and v["_kind"] == "k_generate_rays"]
    if wr:
        cal["raygen_write_counted_bytes"] = sum(wr) / len(wr)
        cal["raygen_known_write_bytes"] = known
        cal["write_factor"] = known / (sum(wr) / len(wr))
    # Round 4: the path pools are read with 16- and 12-byte-per-lane loads since round 3, a pattern the factor above
    # was never calibrated on.  bench.py's box calibration runs the library's exclusive scan of 2^26 int32 in the SAME process: its
    # k_scan_reduce launches read exactly 4 n bytes with 16-byte-per-lane loads (int4, pt_compaction.h) and write next to nothing, its
    # k_scan_apply launches read 4 n and write 4 n -- launches of known traffic in the same --pmc pass.
    n_scan = 4.0 * (1 << 26)
    rd = [float(r["Counter_Value"]) * 1024.0 for r in load_pmc(os.path.join(src, "pmc_fetch"))
          if "k_scan_reduce" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    wr16 = [float(r["Counter_Value"]) * 1024.0 for r in load_pmc(os.path.join(src, "pmc_write"))
            if "k_scan_apply" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
    if rd:
        cal["scan_reduce_fetch_counted_bytes"] = sum(rd) / len(rd)
        cal["scan_reduce_known_read_bytes"] = n_scan
        cal["read_factor_16B_per_lane_loads"] = n_scan / (sum(rd) / len(rd))
        cal["scan_reduce_launches"] = len(rd)
    if wr16:
        cal["scan_apply_write_counted_bytes"] = sum(wr16) / len(wr16)
        cal["scan_apply_known_write_bytes"] = n_scan
        cal["write_factor_16B_per_lane_stores"] = n_scan / (sum(wr16) / len(wr16))
    out["calibration"] = cal

    kb = dict(out.get("k_bounce", {}).get("pmc_per_dispatch", {}))
    kw = out.get("k_mesh_walk", {}).get("pmc_per_dispatch", {})
    if kw:
        # a "launch" of a mesh scene is the pair bench.py times together (pt_api.hip: launch_bounce): the walk, then the bounce
        for c, v in kw.items():
            kb[c] = kb.get(c, 0.0) + v
    if "FETCH_SIZE" in kb and "WRITE_SIZE" in kb:
        rf = cal.get("read_factor_16B_per_lane_loads", cal.get("read_factor", 2.0))
        # snap to the two documented regimes (exact, or the 2x under-count of coalesced streams)
        rf_used = 2.0 if rf > 1.5 else 1.0
        traffic = kb["FETCH_SIZE"] * 1024.0 * rf_used + kb["WRITE_SIZE"] * 1024.0
        out["k_bounce"]["hbm_bytes_per_launch"] = traffic
        out["k_bounce"]["read_factor_used"] = rf_used
        pj = {"hbm_bytes_per_launch": round(traffic, 1), "read_factor_used": rf_used,
              "read_factor_calibrated": rf, "write_factor_calibrated": cal.get("write_factor"),
              "source": "profiles/%s_pmc_summary.json" % args.tag}
        if kw:
            pj["launch"] = "k_mesh_walk + k_bounce (counters of the two dispatches added)"
        # which bench configuration the counters belong to: bench.py reports them only for the same one, scaled by the
        # iterations a launch carries (the figures are stored per iteration of a launch's batch)
        ipl = None
        try:
            for line in open(os.path.join(src, "pmc_fetch.log")):
                if line.startswith("{"):
                    b = json.loads(line)
                    pj["workload"] = b["config"]["workload"].split(", ")[0:2]
                    ipl = b["roofline"]["iterations_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        if ipl:
            pj["iterations_per_launch"] = ipl
            pj["hbm_bytes_per_launch_iteration"] = traffic / ipl
            if "SQ_INSTS_VALU" in kb:
                pj["valu_wave_insts_per_launch_iteration"] = kb["SQ_INSTS_VALU"] / ipl
                pj["salu_insts_per_launch_iteration"] = kb.get("SQ_INSTS_SALU", 0.0) / ipl
                pj["lds_bank_conflict_cycles_per_launch"] = kb.get("SQ_LDS_BANK_CONFLICT")
                if kb.get("GRBM_GUI_ACTIVE"):
                    # counter-derived vector-issue utilisation: wave64 vector instructions x the 2 cycles each occupies a SIMD
                    # (MI355X_MICROARCH.md: "issues each VALU instruction over 2 cycles") / (1024 SIMDs x the launch's cycles;
                    # GRBM_GUI_ACTIVE is summed over the 8 XCDs)
                    pj["grbm_gui_active_per_launch"] = kb["GRBM_GUI_ACTIVE"]
                    pj["valu_utilisation_counter_derived"] = kb["SQ_INSTS_VALU"] * 2.0 / (1024.0 * kb["GRBM_GUI_ACTIVE"] / 8.0)
        if args.set_default:
            json.dump(pj, open(os.path.join(HERE, "pmc_traffic.json"), "w"), indent=1)
        if args.config_key:
            path = os.path.join(HERE, "pmc_configs.json")
            allc = json.load(open(path)) if os.path.exists(path) else {}
            allc[args.config_key] = pj
            json.dump(allc, open(path, "w"), indent=1, sort_keys=True)
        out["k_bounce"]["traffic"] = pj
    if "SQ_LDS_BANK_CONFLICT" in kb:
        out["k_bounce"]["lds_bank_conflict_cycles_per_launch"] = kb["SQ_LDS_BANK_CONFLICT"]
        out["k_bounce"]["lds_bank_conflict_fraction"] = kb["SQ_LDS_BANK_CONFLICT"] / max(kb.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)
    if "SQ_WAVE_CYCLES" in kb:
        out["k_bounce"]["wave_wait_fraction"] = kb["SQ_WAIT_ANY"] / kb["SQ_WAVE_CYCLES"]
    for ii in out.get("instantiations", {}).values():
        sums, cnts = ii.pop("_pmc_sum", {}), ii.pop("_pmc_n", {})
        if sums:
            ii["pmc_per_dispatch"] = {c: v / cnts[c] for c, v in sums.items()}
            pd = ii["pmc_per_dispatch"]
            if "SQ_WAVE_CYCLES" in pd and "SQ_WAIT_ANY" in pd:
                ii["wave_wait_fraction"] = pd["SQ_WAIT_ANY"] / pd["SQ_WAVE_CYCLES"]
            if pd.get("GRBM_GUI_ACTIVE") and "SQ_INSTS_VALU" in pd:
                ii["valu_utilisation_counter_derived"] = pd["SQ_INSTS_VALU"] * 2.0 / (1024.0 * pd["GRBM_GUI_ACTIVE"] / 8.0)
            if "FETCH_SIZE" in pd and "WRITE_SIZE" in pd and ii.get("trace_avg_ns"):
                ii["hbm_bytes_per_launch"] = pd["FETCH_SIZE"] * 1024.0 * out.get("k_bounce", {}).get("read_factor_used", 2.0) + pd["WRITE_SIZE"] * 1024.0
                ii["hbm_fraction_of_8TBs"] = ii["hbm_bytes_per_launch"] / (ii["trace_avg_ns"] * 1e-9) / 8e12
    # the bench line of the profiled command (trace pass), for the record
    try:
        for line in open(os.path.join(src, "trace.log")):
            if line.startswith("{"):
                out["bench_line_of_the_trace_pass"] = json.loads(line)
    except (OSError, ValueError):
        pass
    json.dump(out, open(os.path.join(HERE, args.tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
