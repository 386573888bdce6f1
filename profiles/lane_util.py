"""Distil gpurun_out/lane_<tag>/ (profiles/run_lane_util.sh) into one JSON object per kernel instantiation: mean counters per launch and
    lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)   -- the share of the issued vector lane-slots whose lane is active
    python profiles/lane_util.py <tag> [<tag> ...] > profiles/r06_lane_util.json"""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for tag in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    files = []
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "lane_" + tag, "pass*"))):      # (gpurun MERGES runs: the newest file of every pass)
        fs = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
        files += fs[-1:]
    for f in files:
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_bounce|k_mesh_walk|k_commit\w*)(<[^>]*>)?", r["Kernel_Name"])
            if m:
                acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, c in sorted(acc.items()):
        mean = {n: sum(v) / len(v) for n, v in c.items()}
        e = {"launches": len(c.get("SQ_INSTS_VALU", c.get("SQ_WAIT_ANY", []))), "mean_per_launch": {n: round(v, 1) for n, v in sorted(mean.items())}}
        if mean.get("SQ_ACTIVE_INST_VALU"):
            e["lane_utilisation"] = round(mean["SQ_THREAD_CYCLES_VALU"] / (64.0 * mean["SQ_ACTIVE_INST_VALU"]), 4)
            e["active_inst_valu_cycles_per_instruction"] = round(mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_INSTS_VALU"], 3)
        if mean.get("SQ_WAVE_CYCLES") and mean.get("SQ_ACTIVE_INST_VALU"):
            e["valu_share_of_wave_cycles"] = round(mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_WAVE_CYCLES"], 4)
        if mean.get("SQ_WAIT_ANY") and mean.get("SQ_ACTIVE_INST_ANY"):
            tot = mean["SQ_WAIT_ANY"] + mean["SQ_WAIT_INST_ANY"] + mean["SQ_ACTIVE_INST_ANY"]
            e["of_wave_cycles"] = {"parked (s_waitcnt / barrier)": round(mean["SQ_WAIT_ANY"] / tot, 4), "issue stall": round(mean["SQ_WAIT_INST_ANY"] / tot, 4),
                                   "issuing": round(mean["SQ_ACTIVE_INST_ANY"] / tot, 4), "LDS issue stall (part of issue stall)": round(mean.get("SQ_WAIT_INST_LDS", 0) / tot, 4)}
        res[k] = e
    out[tag] = res
print(json.dumps(out, indent=1))
