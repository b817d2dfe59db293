#!/usr/bin/env python
"""Turns the rocprofv3 (rocpd sqlite) outputs of scripts/profile.sh into the committed summaries
under profiles/:  <tag>_kernel_stats.csv (= rocprofv3 --stats top-kernels), <tag>_pmc.csv (per-kernel
counter sums and per-launch means), <tag>_traffic.json (HBM bytes per launch, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950), <tag>_summary.md.

    python scripts/summarize_profile.py r01
"""
import csv
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    """'void rpt_strict::rpt_paths<rpt_strict::KdFlat>(rptdev::Scene, ...)' -> 'rpt_paths<KdFlat>';
    the two builds of the persistent path kernel are reported under the bench line's name `rpt_paths`
    (the variant is listed in the notes of profiles/README.md)."""
    import re
    head = name.split("(")[0]
    head = re.sub(r"\b\w+::", "", head).replace("void ", "").strip()
    if head.startswith("rpt_paths<"):
        return "rpt_paths"
    return head


def db(sub):
    p = os.path.join(src, sub, "bench_results.db")
    return sqlite3.connect(p) if os.path.exists(p) else None


rows = []
c = db("trace")
if c:
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    rows = [(short(n), n, calls, tot, avg, pct) for n, calls, tot, avg, pct in cur]
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent", "full_name"])
        for s, n, calls, tot, avg, pct in rows:
            w.writerow([s, calls, "%.3f" % tot, "%.3f" % avg, "%.2f" % pct, n])
    regs = {}
    for name, v, s, scr in c.execute("select name, max(vgpr_count), max(sgpr_count), max(scratch_size) from kernels group by name"):
        regs[short(name)] = (v, s, scr)

pmc = {}  # kernel -> counter -> (sum, launches)
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
    c = db(sub)
    if not c:
        continue
    q = ("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
         "group by kernel_name, counter_name")
    for k, cn, total, n in c.execute(q):
        pmc.setdefault(short(k), {})[cn] = (total, n)

with open(os.path.join(dst, tag + "_pmc.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "sum_over_launches", "launches", "mean_per_launch"])
    for k in sorted(pmc):
        for cn in sorted(pmc[k]):
            total, n = pmc[k][cn]
            w.writerow([k, cn, "%.6g" % total, n, "%.6g" % (total / max(1, n))])

traffic = {}
for k, d in pmc.items():
    if not k.startswith("rpt_"):
        continue
    t = {}
    if "FETCH_SIZE" in d:  # KiB; gfx950 reports half the bytes of wide coalesced reads -> double
        t["fetch_bytes_per_launch_raw"] = d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1] * 1024
        t["fetch_bytes_per_launch_x2"] = 2 * t["fetch_bytes_per_launch_raw"]
    if "WRITE_SIZE" in d:
        t["write_bytes_per_launch"] = d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1] * 1024
    if "fetch_bytes_per_launch_x2" in t and "write_bytes_per_launch" in t:
        t["hbm_bytes_per_launch"] = t["fetch_bytes_per_launch_x2"] + t["write_bytes_per_launch"]
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        h, m = d["TCC_HIT_sum"][0], d["TCC_MISS_sum"][0]
        t["l2_hit_rate"] = h / max(1.0, h + m)
    traffic[k] = t
# the PMC passes render 4 spp at 1920x1080 in ONE rpt_paths launch: per-sample figures let bench.py
# scale the measured traffic to the launch size it actually times
PMC_SAMPLES = 1920 * 1080 * 4
for k, t in traffic.items():
    if k == "rpt_paths" and "hbm_bytes_per_launch" in t:
        t["pmc_samples_per_launch"] = PMC_SAMPLES
        t["hbm_bytes_per_sample"] = t["hbm_bytes_per_launch"] / PMC_SAMPLES
traffic["_note"] = ("PMC passes ran bench.py --steps 1 --warmup 0 --spp 4 (2 spp per pass, 9 depths); "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); "
                    "WRITE_SIZE uncalibrated")
json.dump(traffic, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)

with open(os.path.join(dst, tag + "_summary.md"), "w") as f:
    f.write("# rocprofv3 summary %s\n\n" % tag)
    cmdf = os.path.join(src, "trace_command.txt")
    cmd = open(cmdf).read().strip().replace(ROOT + "/", "").replace("/root/repo/", "") if os.path.exists(cmdf) else "python bench.py --steps 2 --warmup 1 --spp 32 --no-cpu-baseline"
    cmd = " ".join(w.split("/")[-1] if w.endswith("bench.py") else w for w in cmd.split())
    f.write("Command: `rocprofv3 --kernel-trace --stats -- %s` (scripts/profile.sh); PMC passes: "
            "`--pmc <counters> --kernel-trace -- python bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline`.\n\n" % cmd)
    bl = os.path.join(src, "bench_trace.log")
    if os.path.exists(bl):
        for line in open(bl):
            if line.startswith("{") and '"ms_per_step"' in line:
                try:
                    d = json.loads(line)
                    k = d["roofline"]["kernel"]
                    f.write("bench.py's own line from that run: value = %.1f %s, ms_per_step = %.1f, `%s` avg launch = %.3f ms "
                            "(HIP events) — compare with the avg us column below.\n\n"
                            % (d["value"], d["unit"], d["ms_per_step"], k, d["roofline"]["kernels"][k]["avg_ms"]))
                except Exception:
                    pass
    f.write("| kernel | calls | total ms | avg us | % | VGPR | SGPR | scratch B/lane |\n|---|---|---|---|---|---|---|---|\n")
    for s, n, calls, tot, avg, pct in rows:
        v = regs.get(s, ("", "", ""))
        f.write("| %s | %d | %.2f | %.1f | %.1f | %s | %s | %s |\n" % (s, calls, tot / 1e3, avg, pct, v[0], v[1], v[2]))
    f.write("\n## PMC (mean per launch)\n\n| kernel | counter | mean per launch |\n|---|---|---|\n")
    for k in sorted(pmc):
        if k.startswith("rpt_"):
            for cn in sorted(pmc[k]):
                total, n = pmc[k][cn]
                f.write("| %s | %s | %.4g |\n" % (k, cn, total / max(1, n)))
    f.write("\n## HBM traffic per launch (bytes)\n\n```json\n%s\n```\n" % json.dumps(traffic, indent=1))
print(open(os.path.join(dst, tag + "_summary.md")).read())
