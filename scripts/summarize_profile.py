#!/usr/bin/env python
"""Turns the rocprofv3 (rocpd sqlite) outputs of scripts/profile.sh into the committed summaries under profiles/:
<tag>_<scene>_kernel_stats.csv (= rocprofv3 --stats top-kernels), <tag>_<scene>_pmc.csv (per-kernel counter sums and
per-launch means), <tag>_<scene>_pmc.json (per kernel: HBM bytes per sample — FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950, WRITE_SIZE as reported —, VALU busy fraction, active lanes per VALU instruction, wait fraction,
L2 hit rate; read by bench.py for the roofline object) and <tag>_<scene>_summary.md.

    python scripts/summarize_profile.py r02 cornell
"""
import csv
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
scene = sys.argv[2] if len(sys.argv) > 2 else "cornell"
src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, scene))
dst = os.environ.get("RPT_PROFILE_DST") or os.path.join(ROOT, "profiles")  # on the GPU box: a directory under gpurun_out/
os.makedirs(dst, exist_ok=True)
pre = os.path.join(dst, "%s_%s" % (tag, scene))
NUM_SIMD, NUM_SE = 1024, 32  # MI355X: 256 CUs x 4 SIMDs; 8 XCDs x 4 shader engines


def short(name):
    """'void rpt_strict::rpt_paths<rpt_strict::KdFlat>(rptdev::Scene, ...)' -> 'rpt_paths'; template arguments are
    dropped so that the names match bench.py's kernel kinds (the variants are listed in the summary's full names)."""
    head = name.split("(")[0]
    head = re.sub(r"\b\w+::", "", head).replace("void ", "").strip()
    for k in ("rpt_paths", "rpt_tree_trace", "rpt_tree_enter", "rpt_rays_init", "rpt_rays_objects"):
        if head.startswith(k + "<"):
            return k
    return head


def db(sub):
    p = os.path.join(src, sub, "bench_results.db")
    return sqlite3.connect(p) if os.path.exists(p) else None


def bench_line(log):
    p = os.path.join(src, log)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{") and '"ms_per_step"' in line:
                try:
                    return json.loads(line)
                except Exception:
                    pass
    return None


def code_object_resources():
    """VGPRs, spills, scratch and LDS of the library's kernels from the code object's own notes (the compiler's figures:
    scripts/kernel_resources.py reads the same).  rocprofv3's kernels table reports VGPRs in an allocation unit that
    halves them on gfx950 (rpt_paths: 128 for 256), and knows nothing of spills."""
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.environ.get("RPTGPU_LIB") or os.path.join(ROOT, "rpt_amd", "lib", "librptgpu.so")
    res = {}
    try:
        with tempfile.TemporaryDirectory() as d:
            import shutil
            shutil.copy(lib, os.path.join(d, "in.o"))
            subprocess.run([llvm + "/llvm-objdump", "--offloading", os.path.join(d, "in.o")], cwd=d, check=True, stdout=subprocess.DEVNULL)
            for co in [os.path.join(d, f) for f in os.listdir(d) if "amdgcn" in f]:
                t = subprocess.run([llvm + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
                for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", t, re.S):
                    b = m.group(0)
                    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))  # noqa: E731
                    sym = re.search(r"\.name:\s+(\S+)", b).group(1)
                    dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
                    res[dem] = {"vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"), "vspill": g("vgpr_spill_count"), "sspill": g("sgpr_spill_count"),
                                "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size"),
                                "waves_by_vgpr": min(8, 512 // max(8, -(-g("vgpr_count") // 8) * 8))}
    except Exception as e:  # (no LLVM tools on this box: the columns fall back to rocprofv3's)
        print("summarize_profile: no code-object notes (%s)" % e, file=sys.stderr)
    return res


rows, regs = [], {}
notes = code_object_resources()
c = db("trace")
if c:
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    rows = [(short(n), n, calls, tot, avg, pct) for n, calls, tot, avg, pct in cur]
    with open(pre + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent", "full_name"])
        for s, n, calls, tot, avg, pct in rows:
            w.writerow([s, calls, "%.3f" % tot, "%.3f" % avg, "%.2f" % pct, n])
    for name, v, s, scr, lds in c.execute("select name, max(vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size) from kernels group by name"):
        regs[name] = (v, s, scr, lds)

pmc = {}  # kernel -> counter -> (sum, launches)
for sub in ("pmc_a", "pmc_b", "pmc_c"):
    c = db(sub)
    if not c:
        continue
    q = ("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
         "group by kernel_name, counter_name")
    for k, cn, total, n in c.execute(q):
        d = pmc.setdefault(short(k), {})
        t0, n0 = d.get(cn, (0.0, 0))
        d[cn] = (t0 + total, n0 + n)

with open(pre + "_pmc.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "sum_over_launches", "launches", "mean_per_launch"])
    for k in sorted(pmc):
        for cn in sorted(pmc[k]):
            total, n = pmc[k][cn]
            w.writerow([k, cn, "%.6g" % total, n, "%.6g" % (total / max(1, n))])

bl = bench_line("pmc_a.log") or bench_line("pmc_b.log") or bench_line("pmc_c.log")
samples = None
if bl:
    samples = bl["value"] * 1e6 * bl["ms_per_step"] / 1e3 * bl["steps"]
pcmd = open(os.path.join(src, "pmc_command.txt")).read().strip() if os.path.exists(os.path.join(src, "pmc_command.txt")) else ""
pcmd = " ".join(w.split("/")[-1] if w.endswith("bench.py") else w for w in pcmd.split())
out = {"workload": (bl["config"]["workload"] if bl else "") + " [PMC passes: %s]" % pcmd, "samples": samples, "kernels": {},
       "_note": "FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE (KiB) as "
                "reported (uncalibrated); valu_busy = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * SQ_BUSY_CYCLES/32 SEs); lanes_active = "
                "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64); valu_frac = valu_busy * lanes_active / 64; wait_frac = "
                "SQ_WAIT_ANY / SQ_WAVE_CYCLES; vmem_latency_cycles = SQ_INST_LEVEL_VMEM / (SQ_INSTS_VMEM_RD + SQ_INSTS_VMEM_WR)"}
for k, d in pmc.items():
    if not k.startswith("rpt_"):
        continue
    g = lambda cn: d[cn][0] if cn in d else None  # noqa: E731
    t = {"launches": max(v[1] for v in d.values())}
    if g("FETCH_SIZE") is not None:
        t["fetch_bytes_x2"] = 2 * g("FETCH_SIZE") * 1024
    if g("WRITE_SIZE") is not None:
        t["write_bytes"] = g("WRITE_SIZE") * 1024
    if "fetch_bytes_x2" in t and "write_bytes" in t:
        t["hbm_bytes"] = t["fetch_bytes_x2"] + t["write_bytes"]
        if samples:
            t["hbm_bytes_per_sample"] = t["hbm_bytes"] / samples
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        t["l2_hit_rate"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_BUSY_CYCLES"):
        t["valu_busy"] = g("SQ_ACTIVE_INST_VALU") * 4 / (NUM_SIMD * g("SQ_BUSY_CYCLES") / NUM_SE)
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        t["lanes_active"] = g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU")
    if "valu_busy" in t and "lanes_active" in t:
        t["valu_frac"] = t["valu_busy"] * t["lanes_active"] / 64.0
    if g("SQ_WAIT_ANY") is not None and g("SQ_WAVE_CYCLES"):
        t["wait_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_INST_LEVEL_VMEM") and (g("SQ_INSTS_VMEM_RD") or 0) + (g("SQ_INSTS_VMEM_WR") or 0) > 0:
        t["vmem_latency_cycles"] = g("SQ_INST_LEVEL_VMEM") / ((g("SQ_INSTS_VMEM_RD") or 0) + (g("SQ_INSTS_VMEM_WR") or 0))
    out["kernels"][k] = t
# kernel durations inside the PMC runs (for measured GB/s of each kernel: bytes / its own time in THAT run)
c = db("pmc_a")
if c:
    try:
        for name, tot, n in c.execute("select name, total_duration, total_calls from top_kernels"):
            k = short(name)
            if k in out["kernels"]:
                out["kernels"][k]["pmc_run_total_us"] = out["kernels"][k].get("pmc_run_total_us", 0.0) + tot
    except Exception:
        pass
for k, t in out["kernels"].items():
    if "hbm_bytes" in t and t.get("pmc_run_total_us"):
        t["hbm_GBs_in_pmc_run"] = t["hbm_bytes"] / 1e9 / (t["pmc_run_total_us"] / 1e6)
        t["hbm_frac_in_pmc_run"] = t["hbm_GBs_in_pmc_run"] / 8000.0
json.dump(out, open(pre + "_pmc.json", "w"), indent=1)

with open(pre + "_summary.md", "w") as f:
    f.write("# rocprofv3 summary %s %s\n\n" % (tag, scene))
    cmdf = os.path.join(src, "trace_command.txt")
    cmd = open(cmdf).read().strip() if os.path.exists(cmdf) else ""
    cmd = " ".join(w.split("/")[-1] if w.endswith("bench.py") else w for w in cmd.split())
    f.write("Trace: `rocprofv3 --kernel-trace --stats -- %s`; PMC passes (scripts/profile.sh): `--pmc <counters> --kernel-trace -- %s`.\n\n" % (cmd, pcmd))
    d = bench_line("bench_trace.log")
    if d:
        k = d["roofline"]["kernel"]
        f.write("bench.py's own line from the traced run: value = %.1f %s, ms_per_step = %.1f, `%s` avg launch = %.3f ms "
                "(HIP events) — compare with the avg us column below.\n\n"
                % (d["value"], d["unit"], d["ms_per_step"], k, d["roofline"]["kernels"][k]["avg_ms"]))
        json.dump(d, open(pre + "_bench_line.json", "w"), indent=1)
    f.write("| kernel | calls | total ms | avg us | % | VGPR | spilled V / S | waves/SIMD by VGPRs | SGPR | scratch B/lane | static LDS B |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    for s, n, calls, tot, avg, pct in rows:
        key = n if n in notes else next((k for k in notes if k.split("(")[0] == n.split("(")[0]), None)
        if key:  # the library's own kernels: the compiler's figures
            r = notes[key]
            cols = (r["vgpr"], "%d / %d" % (r["vspill"], r["sspill"]), r["waves_by_vgpr"], r["sgpr"], r["scratch"], r["lds"])
        else:    # runtime / rocPRIM kernels: what rocprofv3 reports (its VGPR unit halves the count on gfx950)
            v = regs.get(n, ("", "", "", ""))
            cols = ("%s (rocprofv3)" % v[0] if v[0] != "" else "", "", "", v[1], v[2], v[3])
        f.write("| %s | %d | %.2f | %.1f | %.1f | %s | %s | %s | %s | %s | %s |\n" % ((re.sub(r"\b\w+::", "", n.split("(")[0]).replace("void ", ""), calls, tot / 1e3, avg, pct) + cols))
    f.write("\nVGPR, spills, scratch and LDS of the rpt_* kernels: the code object's notes (llvm-readelf --notes of librptgpu.so), "
            "i.e. the compiler's own report; waves/SIMD by VGPRs = 512 / VGPRs rounded up to 8 (the launch bounds and LDS may allow fewer).\n")
    f.write("\n## Derived per kernel (PMC runs)\n\n| kernel | launches | HBM B/sample | HBM GB/s | of 8 TB/s | VALU busy | lanes/64 | VALU x lanes | wait | L2 hit | VMEM latency |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    fmt = lambda v, p="%.3g": (p % v) if v is not None else ""  # noqa: E731
    for k in sorted(out["kernels"], key=lambda k: -out["kernels"][k].get("pmc_run_total_us", 0)):
        t = out["kernels"][k]
        f.write("| %s | %d | %s | %s | %s | %s | %s | %s | %s | %s | %s |\n" % (
            k, t["launches"], fmt(t.get("hbm_bytes_per_sample")), fmt(t.get("hbm_GBs_in_pmc_run")), fmt(t.get("hbm_frac_in_pmc_run")),
            fmt(t.get("valu_busy")), fmt(t.get("lanes_active")), fmt(t.get("valu_frac")), fmt(t.get("wait_frac")),
            fmt(t.get("l2_hit_rate")), fmt(t.get("vmem_latency_cycles"))))
    f.write("\n" + out["_note"] + "\n")
print(open(pre + "_summary.md").read())
