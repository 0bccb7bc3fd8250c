#!/usr/bin/env python
"""Condense the rocprofv3 CSVs of scripts/collect_profiles.sh into two small files:
<dir>/summary/<tag>_kernel_stats.csv   (per-kernel calls / total / average duration)
<dir>/summary/<tag>_pmc_summary.json   (per-kernel: avg duration, MFMA busy %, executed fp64 MFMA flops,
                                        HBM read/write bytes per launch [FETCH_SIZE doubled, see
                                        /opt/skills/guides/MI355X_MICROARCH.md section HBM], L2 hit rate)
These are what gets copied into profiles/."""
import collections
import csv
import json
import os
import re
import sys

d, tag = sys.argv[1], sys.argv[2]
out = os.path.join(d, "summary")
os.makedirs(out, exist_ok=True)
short = lambda k: re.sub(r"\(anonymous namespace\)::|xivo_hip::|void ", "", k).split("(")[0]


def rows(name):
    fn = os.path.join(d, name)
    return list(csv.DictReader(open(fn))) if os.path.exists(fn) else []


# ---- kernel stats from the plain trace
dur = collections.defaultdict(list)
for r in rows("stats_kernel_trace.csv"):
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in dur.values()) or 1.0
with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Calls", "TotalDuration_us", "AverageDuration_us", "Percentage"])
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), round(sum(v), 1), round(sum(v) / len(v), 2), round(100 * sum(v) / tot, 2)])


def counters(name):
    acc = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in rows(name + "_counter_collection.csv"):
        acc[short(r["Kernel_Name"])][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    return acc


def durations(name):
    dd = {}
    for r in rows(name + "_kernel_trace.csv"):
        dd[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return dd


summary = {}
m, md = counters("mfma"), durations("mfma")
for k, disp in m.items():
    n = len(disp)
    busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for c in disp.values())
    gui = sum(c.get("GRBM_GUI_ACTIVE", 0) for c in disp.values())          # summed over the 8 XCDs
    mops = sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) for c in disp.values())
    mops32 = sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) for c in disp.values())
    us = sum(md.get(i, 0) for i in disp)
    s = summary.setdefault(k, {})
    s["launches_profiled"] = n
    s["avg_duration_us_pmc_run"] = us / n if n else None
    # 1024 SIMDs; GRBM_GUI_ACTIVE is per XCD (x8) -> chip cycles = gui / 8
    s["mfma_busy_pct"] = 100.0 * busy / (gui / 8.0 * 1024) if gui else None
    s["executed_mfma_f64_flops_per_launch"] = mops * 512.0 / n if n else None
    s["executed_mfma_f32_flops_per_launch"] = mops32 * 512.0 / n if n else None
    s["shader_clock_ghz"] = (gui / 8.0) / (us * 1e3) if us else None
for nm, key, scale in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
    for k, disp in counters(nm).items():
        vals = [c.get(key, 0) for c in disp.values()]
        if vals:
            summary.setdefault(k, {})["hbm_%s_bytes_per_launch" % ("read" if nm == "fetch" else "write")] = \
                sum(vals) / len(vals) * 1024.0 * scale
for k, disp in counters("l2").items():
    h = sum(c.get("TCC_HIT_sum", 0) for c in disp.values()); mi = sum(c.get("TCC_MISS_sum", 0) for c in disp.values())
    if h + mi:
        summary.setdefault(k, {})["l2_hit_pct"] = 100.0 * h / (h + mi)
summary["_notes"] = {
    "fetch_correction": "FETCH_SIZE (KiB) x 1024 x 2: gfx950 rocprofv3 reports half the bytes of wide coalesced reads",
    "mfma_busy_pct": "SQ_VALU_MFMA_BUSY_CYCLES / (chip cycles x 1024 SIMDs); 64 busy cycles per v_mfma_f64_16x16x4_f64",
    "command": open(os.path.join(d, "command.txt")).read().strip() if os.path.exists(os.path.join(d, "command.txt")) else None,
}
json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in summary.items() if not k.startswith("_")}, indent=1)[:3000])
