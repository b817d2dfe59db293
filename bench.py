#!/usr/bin/env python
"""bench.py — Msamples/s of the rpt hot path on MI355X (BASELINE.json metric).

A step = one complete frame of BASELINE configs[1]: examples/cornell.rs geometry, 1920x1080,
8 bounces, 512 samples per pixel (override with --spp), through the C ABI in parity mode (IEEE f64,
no FMA contraction).  The timed region is what `Renderer::sample` covers: ray generation
through the last bounce, the reduce over ranks, and the write of the W*H mean colours into HOST memory
(rank 0); scene construction (kd build + upload) is outside it and reported as `scene_create_ms` /
`wall_clock_per_frame_ms`.  With N GPUs rank r renders the tiles tile_id % N == r of the SAME frame
(strong scaling) and rank 0 gathers the f32 pixels each rank owns, one RCCL send / receive group per step.

    python bench.py                      # 1 GPU: the headline (C2) + the other BASELINE configs + live counters
    python bench.py --gpus N --steps K --warmup W    # N GPUs: bench.py starts its own N ranks (torch.distributed.run,
                                                     # one per GPU); fewer than N visible devices is an error
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # the same under an outer launcher
    python bench.py --scene dragon --spp 16          # another BASELINE config at ITS frame size, nothing else
    python bench.py --scene simple_video             # scene rebuilt per frame (examples/simple_video.rs): frames/s

The default 1-GPU run prints ONE JSON line.  Besides the headline it carries
  * `other_configs`: BASELINE configs[2-4] (dragon-class mesh at its 256 spp, fractal spheres, wine glass — mesh and
    glass-spheres variants) at their own frame sizes and 64 spp per step (a step is then a few passes of the size the
    BASELINE sample counts run in),
    5 steps + 1 warm-up each with median / min / max, each with its own roofline object and CPU baseline (>= 16 spp,
    median of 3 repetitions); their values again as top-level scalars (`c3_msamples`, `c4_msamples`, ...).  At N > 1
    the two configs BASELINE assigns to 8 GPUs (fractal spheres, wine-glass mesh) run through the same sharded path
    at 256 spp per step, and `exchange` says whether the library's RCCL exchange moved frames in this run;
  * `roofline` objects whose counter fields (VALU busy, lanes active per VALU instruction, HBM bytes, L2 requests)
    are measured IN THIS RUN: bench.py re-runs one step of each workload under `rocprofv3 --pmc ... --kernel-trace`
    in a child process (no torch, ~10 s per pass) and reads the counters back.  If rocprofv3 is not usable the
    fields fall back to the newest committed profiles/r*_<scene>_pmc.json and `pmc_source` says so.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
L2_PEAK_GBS = 34500.0    # ibid.: aggregate L2 bandwidth, 8 XCDs
# f64 VALU lane slots per second: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz, halved because an f64 instruction
# occupies the SIMD-32 for twice the cycles of an f32 one (78.6 TFLOP/s f64 FMA = 2 flops x this number)
VALU_F64_PEAK_TLANES = 256 * 4 * 32 * 2.4e9 / 2 / 1e12
NUM_SIMD, NUM_SE = 1024, 32

# SURVEY §8d accounting constants, f64 layout (every float field of the f32 layout doubles)
RAY_IO = 2 * (32 + 16)
INST, NODE, LEAF, REF, TRI, PLANE = 96, 16, 16, 4, 72, 32
SHADE_GEOM, MATERIAL = 72, 64
STATE_RW, FOLD_RW = 2 * 192, 2 * 48
ENV = 4 * 32
FB = 24

# the other BASELINE configs in the default run: (scene, spp per step).  Frame size and bounces are the config's own.
# (C3 at its BASELINE 256 spp — it fits one GPU; the 4K configs at 64 spp per step: BASELINE renders them with 1024 / 4096
# spp, i.e. in passes as large as the device holds, and since round 6 a pass holds more than 16 spp of a 4K frame — a
# 16-spp step would time a pass smaller than the config's own (C5 mesh: 913 Msamples/s at 16 spp per step, 964 at 64;
# profiles/r06_pass_size_ab.txt).  5 steps + 1 warm-up each, median and min / max on the line)
OTHER_CONFIGS = (("dragon", 256), ("fractal_spheres", 64), ("wine_glass", 64), ("glass", 64))
OTHER_STEPS, OTHER_WARMUP = 5, 1
PMC_MAX_SPP = 32  # the counter pass of a secondary config renders at most this many spp (the counters are ratios)
CPU_MIN_SPP, CPU_REPS = 16, 3
PMC_A = "FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
PMC_B = "WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM"
PMC_C = "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"


def algorithmic_bytes(c, env_is_hdri):
    """Per-kernel algorithmic bytes of ONE batch whose reference-algorithm visit counts are `c`
    (the oracle's counters: the counts are defined by the reference traversal on the reference
    tree, so a smarter traversal cannot shrink its own denominator)."""
    ext = (c["closest_rays"] * RAY_IO + c["n_inst"] * INST + c["n_inner"] * NODE + c["n_leaf"] * LEAF
           + c["n_ref"] * REF + c["n_tri"] * TRI + c["n_plane"] * PLANE)
    sha = (c["shadow_rays"] * RAY_IO + c["n_inst_sh"] * INST + c["n_inner_sh"] * NODE + c["n_leaf_sh"] * LEAF
           + c["n_ref_sh"] * REF + c["n_tri_sh"] * TRI + c["n_plane_sh"] * PLANE)
    shade = c["hits"] * (SHADE_GEOM + MATERIAL) + c["segments"] * STATE_RW + (c["misses"] * ENV if env_is_hdri else 0)
    resolve = c["segments"] * FOLD_RW + c["samples"] * FB
    raygen = c["samples"] * RAY_IO // 2
    parts = {"rpt_raygen": raygen, "rpt_extend": ext, "rpt_shade": shade, "rpt_shadow": sha, "rpt_resolve": resolve}
    parts["rpt_paths"] = sum(parts.values())  # the persistent kernel does all of it in one launch
    # the per-tree traversal kernel on its own: what KdTree::intersect reads below the root slab test
    # (kdtree.rs:151-223) for closest-hit and shadow rays together, plus one ray in / one record out per root test
    parts["rpt_tree_trace"] = ((c["n_root"] + c["n_root_sh"]) * RAY_IO
                               + (c["n_inner"] + c["n_inner_sh"]) * NODE + (c["n_leaf"] + c["n_leaf_sh"]) * LEAF
                               + (c["n_ref"] + c["n_ref_sh"]) * REF + (c["n_tri"] + c["n_tri_sh"]) * TRI)
    return parts


def host_cpus():
    """threads this process may actually run on (affinity mask and cgroup quota), and the raw count"""
    raw = os.cpu_count() or 1
    n = raw
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(float(q[0]) / float(q[1]))))
    except Exception:
        pass
    return n, raw


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def short_kernel(name):
    """'void rpt_strict::rpt_paths<rpt_strict::KdFlat>(rptdev::Scene, ...)' -> 'rpt_paths'"""
    import re
    head = name.split("(")[0]
    head = re.sub(r"\b\w+::", "", head).replace("void ", "").strip()
    for k in ("rpt_paths", "rpt_tree_trace", "rpt_tree_enter", "rpt_rays_init", "rpt_rays_objects", "rpt_finish",
              "rpt_tree_general", "rpt_shadow_rays"):
        if head.startswith(k + "<"):
            return k
    return head


# ---------------------------------------------------------------------------------------------- live counters
def pmc_worker(args):
    """Child of live_pmc(): ONE step of the workload through the same C-ABI call the bench times, nothing else
    (no torch: the process is up in a second).  Runs under rocprofv3."""
    import numpy as np
    import rpt_amd
    from rpt_amd import _abi, make_params, scenes
    scene, camera, cfg = scenes.SCENES[args.scene]()
    W, H = args.width or cfg["width"], args.height or cfg["height"]
    B = args.bounces if args.bounces is not None else cfg["max_bounces"]
    spp = args.spp or cfg["num_samples"]
    flag = {"auto": 0, "persistent": _abi.RPT_FLAG_PERSISTENT, "wavefront": _abi.RPT_FLAG_WAVEFRONT}[args.pipeline]
    gpu = rpt_amd.GpuScene(scene, 0)
    out = np.empty(W * H * 3, dtype=np.float32)
    gpu.render_batch_reduce(camera, make_params(W, H, B, spp, seed=0x52505447, flags=flag), root=0, out=out)
    gpu.close()


def derive_counters(d, samples, total_us):
    """per-kernel derived metrics from summed raw counters `d` (name -> value) of the launches of one kernel"""
    g = d.get
    t = {}
    if g("FETCH_SIZE") is not None:
        t["fetch_bytes_x2"] = 2 * g("FETCH_SIZE") * 1024  # gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md)
    if g("WRITE_SIZE") is not None:
        t["write_bytes"] = g("WRITE_SIZE") * 1024
    if "fetch_bytes_x2" in t:
        t["hbm_bytes"] = t["fetch_bytes_x2"] + t.get("write_bytes", 0.0)
        t["hbm_bytes_is"] = "2 x FETCH_SIZE + WRITE_SIZE" if "write_bytes" in t else "2 x FETCH_SIZE (no WRITE_SIZE pass)"
        if samples:
            t["hbm_bytes_per_sample"] = t["hbm_bytes"] / samples
        if total_us:
            t["hbm_GBs"] = t["hbm_bytes"] / 1e9 / (total_us / 1e6)
            t["hbm_frac"] = t["hbm_GBs"] / HBM_PEAK_GBS
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        t["l2_hit_rate"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        req = g("TCC_HIT_sum") + g("TCC_MISS_sum")
        if total_us:  # one TCC request moves at most one 128-B line: an upper bound of the bytes the L2s served
            t["l2_GBs_upper"] = req * 128.0 / 1e9 / (total_us / 1e6)
            t["l2_frac_upper"] = t["l2_GBs_upper"] / L2_PEAK_GBS
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
    return t


def live_pmc(scene, spp, passes, extra=(), timeout=150):
    """One step of `scene` at `spp` under `rocprofv3 --pmc <pass> --kernel-trace`, once per counter pass (FETCH_SIZE
    and WRITE_SIZE do not fit one pass; never combined with sys / hip / hsa tracing).  Returns ({kernel: derived},
    note) — the counters of THIS run on THIS box — or (None, why)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    raw, dur, note = {}, {}, None
    tmp = tempfile.mkdtemp(prefix="rptpmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, counters in enumerate(passes):
            d = os.path.join(tmp, "p%d" % i)
            cmd = [exe, "--pmc"] + counters.split() + ["--kernel-trace", "-d", d, "-o", "w", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-worker", "--scene", scene, "--spp", str(spp)] + list(extra)
            why = None
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
                dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
                if r.returncode != 0 or not dbs:
                    why = "rocprofv3 pass %d failed (rc %d): %s" % (i, r.returncode, r.stdout.decode(errors="replace")[-300:])
            except subprocess.TimeoutExpired:
                why = "rocprofv3 pass %d timed out" % i
            if why:  # the SQ pass is the one the roofline needs; a later pass that fails only costs its own fields
                if i == 0:
                    return None, why
                note = why
                continue
            try:
                c = sqlite3.connect(dbs[0])
                for k, cn, total in c.execute("select kernel_name, counter_name, sum(value) from counters_collection "
                                              "group by kernel_name, counter_name"):
                    d_k = raw.setdefault(short_kernel(k), {})  # template variants of a kernel kind share the short name
                    d_k[cn] = d_k.get(cn, 0.0) + total
                if i == 0:  # the kernels' durations INSIDE the counter run (microseconds), the denominator of its byte rates
                    for name, tot in c.execute("select name, total_duration from top_kernels"):
                        dur[short_kernel(name)] = dur.get(short_kernel(name), 0.0) + tot
                c.close()
            except Exception as e:
                if i == 0:
                    return None, "reading the rocprofv3 database failed: %s" % e
                note = "pass %d: %s" % (i, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"raw": raw, "dur_us": dur, "note": note}, None


# ---------------------------------------------------------------------------------------------- one workload
class Workload:
    """one BASELINE config on this rank: scene on the device, step(), timing, accounting"""

    def __init__(self, name, args, rank, world, local_rank, backend, spp=None, is_headline=True):
        import rpt_amd
        from rpt_amd import _abi, scenes
        self.name, self.args, self.rank, self.world = name, args, rank, world
        self.scene, self.camera, cfg = scenes.SCENES[name]()
        self.W = (args.width if is_headline else None) or cfg["width"]
        self.H = (args.height if is_headline else None) or cfg["height"]
        self.B = args.bounces if (is_headline and args.bounces is not None) else cfg["max_bounces"]
        self.spp = spp or (args.spp if is_headline else None) or cfg["num_samples"]
        self.pipe_flag = {"auto": 0, "persistent": _abi.RPT_FLAG_PERSISTENT, "wavefront": _abi.RPT_FLAG_WAVEFRONT}[args.pipeline]
        t0 = time.perf_counter()
        self.gpu = rpt_amd.GpuScene(self.scene, local_rank)  # flatten + kd build (reference rule) + upload
        self.scene_create_ms = (time.perf_counter() - t0) * 1e3
        self.scene_create_warm_ms = None
        if is_headline:  # the first handle of a process also pays HIP initialisation; what every LATER scene costs:
            self.gpu.close()
            t0 = time.perf_counter()
            self.gpu = rpt_amd.GpuScene(self.scene, local_rank)
            self.scene_create_warm_ms = (time.perf_counter() - t0) * 1e3
        self.step_no = 0

    def params(self, **kw):
        from rpt_amd import _abi, make_params
        base = 0 if self.args.fixed_samples else self.step_no * self.spp
        kw.setdefault("collective", _abi.RPT_COLLECTIVE_REDUCE if getattr(self.args, "collective", "gather") == "reduce" else _abi.RPT_COLLECTIVE_GATHER)
        return make_params(self.W, self.H, self.B, self.spp, seed=0x52505447, sample_index_base=base,
                           flags=_abi.RPT_FLAG_PROFILE_KERNELS | self.pipe_flag, **kw)

    def close(self):
        self.gpu.close()


def committed_n1_line(scene, W, H, B, spp):
    """The newest committed bench line (profiles/rNN_bench_default.json, profiles/rNN_<scene>_bench_line.json) of the same
    workload on ONE GPU, for an N-GPU line to carry along; None when there is none.  Informational: a different build."""
    import glob
    import re
    want = "%s %dx%d, %d bounces, %d spp" % (scene, W, H, B, spp)
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*_%s_bench_line.json" % scene)):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        try:
            text = open(path).read().strip()
            try:
                d = json.loads(text)  # (an indented dump: profiles/rNN_<scene>_bench_line.json)
            except ValueError:
                d = json.loads(text.splitlines()[-1])  # (the bench's own single line after other output)
        except Exception:
            continue
        if not m or d.get("n_gpus") != 1 or not str((d.get("config") or {}).get("workload", "")).startswith(want):
            continue
        if best is None or int(m.group(1)) > best[0]:
            best = (int(m.group(1)), {"ms_per_step": d["ms_per_step"], "value": d["value"], "source": os.path.relpath(path, ROOT)})
    return best[1] if best else None


def accounting(wl, st):
    """§8d algorithmic bytes per kernel kind for what this rank traced (st.samples): visit counts of the REFERENCE
    algorithm from the instrumented oracle on a bounded sample (1 spp on 1/8 of the tiles), scaled linearly"""
    from oracle import oracle_ffi as O
    from rpt_amd import make_params
    osc = O.OracleScene(wl.scene)
    pc = make_params(wl.W, wl.H, wl.B, 1, seed=0x52505447, tile=(32, 8), part=(0, 8))
    _, cnt = osc.render(wl.camera, pc, threads=0, counters=True)
    scale = float(st.samples) / max(1, cnt["samples"])
    bytes_k = {k: v * scale for k, v in algorithmic_bytes(cnt, wl.scene.environment.hdri is not None).items()}
    bytes_k["rpt_tree_enter+sort"] = 0.0
    return bytes_k, osc


def cpu_baseline(wl, osc, budget_s, cpu_spp=None):
    """the oracle (uninstrumented -march=native build) on this box's host cores: BASELINE.md §2 — the config's frame
    size and bounce count at >= 16 spp, CPU_REPS repetitions of about budget_s seconds each, the MEDIAN reported.  A
    whole frame at 16 spp takes minutes on the slower configs, so a repetition renders 1/k of the interleaved 32x8
    tiles (a regular sample of the whole frame: cost per sample as in the full frame), k sized for budget_s."""
    from oracle import oracle_ffi as O
    from rpt_amd import make_params
    ncores, raw = host_cpus()
    L, how = O.baseline_lib(native=True)
    fast = O.OracleScene(wl.scene, L)
    W, H, B = wl.W, wl.H, wl.B
    pcal = make_params(W, H, B, 1, seed=0x52505447, tile=(32, 8), part=(0, 8))  # calibrate on 1 spp of 1/8 of the frame
    t1 = time.perf_counter()
    fast.render(wl.camera, pcal, threads=ncores)
    rate = (W * H / 8.0) / max(1e-6, time.perf_counter() - t1)
    spp = max(CPU_MIN_SPP, cpu_spp or 0)
    whole = W * H * spp / rate                      # seconds for the whole frame at spp
    k = int(min(256, max(1, round(whole / budget_s))))
    pcpu = make_params(W, H, B, spp, seed=0x52505447, tile=(32, 8), part=(0, k))
    n_pix = W * H if k == 1 else int(count_part_pixels(W, H, k))
    rates, walls = [], []
    for rep in range(CPU_REPS):
        t1 = time.perf_counter()
        fast.render(wl.camera, pcpu, threads=ncores)
        dt = time.perf_counter() - t1
        walls.append(dt)
        rates.append(n_pix * spp / dt / 1e6)
    med = sorted(rates)[len(rates) // 2]
    cpu = {"value": med, "unit": "Msamples/s", "cores": ncores, "kind": "port", "repetitions": CPU_REPS,
           "all_repetitions": rates,
           "sample": "%s %dx%d, %d bounces, %d spp%s, median of %d repetitions (%.1f s wall each): C++ restatement of rpt's rayon "
                     "path (oracle/, %s), one task per row claimed dynamically by %d std::threads on %s (%d logical CPUs)"
                     % (wl.name, W, H, B, spp, "" if k == 1 else " on 1/%d of the interleaved 32x8 tiles" % k, CPU_REPS,
                        sorted(walls)[len(walls) // 2], how, ncores, cpu_model(), raw)}
    if osc is not None and budget_s >= 4:  # the instrumented checker build on the same sample, once, for the record
        t1 = time.perf_counter()
        osc.render(wl.camera, pcpu, threads=ncores)
        dti = time.perf_counter() - t1
        cpu["instrumented_checker_build"] = {"value": n_pix * spp / dti / 1e6, "unit": "Msamples/s",
                                             "note": "liboracle.so with visit counters compiled in (x86-64-v3)"}
    return cpu


def count_part_pixels(W, H, k, tile=(32, 8)):
    """pixels of the tiles with tile_id % k == 0"""
    import numpy as np
    tiles_x = (W + tile[0] - 1) // tile[0]
    ys, xs = np.mgrid[0:H, 0:W]
    return int((((ys // tile[1]) * tiles_x + xs // tile[0]) % k == 0).sum())


def kernel_table(wl, st, bytes_k):
    from rpt_amd import _abi
    NK = _abi.RPT_K_COUNT
    names = [wl.gpu.lib.rptgpu_kernel_name(k).decode() for k in range(NK)]
    kern_ms = {names[k]: st.kernel_ms[k] for k in range(NK)}
    kern_n = {names[k]: int(st.kernel_launches[k]) for k in range(NK)}
    names = [k for k in names if kern_n[k] > 0]
    # the kernel the roofline line is about: the one with the most time among the disjoint kinds, or the
    # per-tree traversal when it is the bulk of the queries that contain it (scenes with deep trees)
    disjoint = {k: kern_ms[k] for k in names if k not in ("rpt_tree_trace", "rpt_tree_enter+sort")}
    dominant = max(disjoint, key=disjoint.get)
    if "rpt_tree_trace" in names and kern_ms["rpt_tree_trace"] >= 0.5 * (kern_ms.get("rpt_extend", 0) + kern_ms.get("rpt_shadow", 0)) \
            and dominant in ("rpt_extend", "rpt_shadow"):
        dominant = "rpt_tree_trace"
    kernels = {}
    for k in names:
        ms, n = kern_ms[k], max(1, kern_n[k])
        kernels[k] = {"launches": kern_n[k], "avg_ms": ms / n, "total_ms": ms,
                      "alg_GB_per_launch": bytes_k.get(k, 0.0) / n / 1e9,
                      "accounting_GBs": (bytes_k.get(k, 0.0) / 1e9) / (ms / 1e3) if ms > 0 else None}
    return kernels, dominant, kern_n


def roofline_object(wl, st, kernels, dominant, kern_n, bytes_k, pmc, pmc_note, pmc_spp):
    """What bounds the dominant kernel.  This path is branchy scalar f64 whose scenes live in LDS / L2: the roof is the
    f64 VALU — useful lane slots per second — not HBM.  `frac` = VALU busy x lanes active / 64 from the SQ counters;
    `accounting_frac` is SURVEY §8d's figure (algorithmic bytes of the REFERENCE traversal / time / 8 TB/s), kept for the
    contract: it exceeds 1 wherever this traversal skips work the reference's does (leaf-box filter), so it is an
    accounting number, not a fraction of any roof."""
    rank_samples = float(st.samples)
    acc = kernels[dominant]["accounting_GBs"]
    roof = {"bound": "valu", "kernel": dominant, "unit": "Tlane-slot/s (f64 VALU: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2)",
            "peak": VALU_F64_PEAK_TLANES, "achieved": None, "frac": None, "frac_kind": "valu_lane_slots", "traffic": None,
            "frac_is": "VALU busy x lanes active / 64 of the dominant kernel (SQ counters of this run): the share of the f64 "
                       "lane slots that did work.  NOT bytes / HBM peak — that figure is accounting_frac below",
            "accounting_GBs": acc, "accounting_frac": (acc / HBM_PEAK_GBS) if acc else None,
            "accounting_frac_kind": "survey_8d_reference_bytes_over_hbm_peak (not headroom: served from registers / LDS / L2)",
            "accounting_is": "SURVEY §8d: algorithmic bytes of the REFERENCE traversal x units / launch time / 8 TB/s.  The bytes "
                             "are served from registers, LDS and L2 (hbm_frac is the measured HBM share) and partly never "
                             "touched (leaf-box filter, untraced zero-contribution shadow rays): exceeds 1 when cache-served",
            "alg_bytes_per_sample": bytes_k["rpt_paths"] / rank_samples if rank_samples else None,
            "kernels": kernels}
    k, src = None, None
    if pmc is not None:
        # (RptStats brackets a depth's visibility queries as "rpt_shadow"; the profiler sees the kernel inside: rpt_shadow_rays)
        pname = dominant if pmc["raw"].get(dominant) else {"rpt_shadow": "rpt_shadow_rays"}.get(dominant, dominant)
        raw = pmc["raw"].get(pname)
        if raw:
            W, H = wl.W, wl.H
            k = derive_counters(raw, float(W) * H * pmc_spp, pmc["dur_us"].get(pname))
            if pname != dominant:
                roof["counters_of"] = pname
            src = "live: rocprofv3 --pmc passes of one %d-spp step of this workload, run by this bench.py invocation" % pmc_spp
    if k is None:  # fall back to the committed summary of an earlier profile of the same workload
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc.json" % wl.name)))
        path = wl.args.pmc_json or (cands[-1] if cands else None)
        if path and os.path.exists(path):
            try:
                pj = json.load(open(path))
                k = dict(pj.get("kernels", {}).get(dominant) or {})
                if "hbm_frac_in_pmc_run" in k:
                    k["hbm_frac"], k["hbm_GBs"] = k["hbm_frac_in_pmc_run"], k.get("hbm_GBs_in_pmc_run")
                src = "COMMITTED, not measured in this run: %s (%s)%s" % (os.path.relpath(path, ROOT), pj.get("workload", ""),
                                                                           "; live attempt: " + pmc_note if pmc_note else "")
            except Exception as e:
                roof["pmc_error"] = str(e)
    if k:
        roof["pmc_source"] = src
        for f in ("valu_busy", "lanes_active", "valu_frac", "wait_frac", "l2_hit_rate", "l2_GBs_upper", "l2_frac_upper",
                  "vmem_latency_cycles", "hbm_frac", "hbm_bytes_is"):
            if k.get(f) is not None:
                roof[f] = k[f]
        if k.get("hbm_GBs") is not None:
            roof["hbm_measured_GBs"] = k["hbm_GBs"]
        if k.get("hbm_bytes_per_sample") is not None:  # per launch of the dominant kernel in the timed run, like `achieved`
            roof["traffic"] = k["hbm_bytes_per_sample"] * rank_samples / max(1, kern_n[dominant])
        if k.get("valu_frac") is not None:
            roof["frac"] = min(1.0, k["valu_frac"])
            roof["achieved"] = roof["frac"] * VALU_F64_PEAK_TLANES
        hf, vf, wf = k.get("hbm_frac") or 0.0, k.get("valu_busy") or 0.0, k.get("wait_frac") or 0.0
        roof["bound"] = "hbm" if hf >= max(vf, 0.5) else ("valu" if vf >= 0.6 or vf > wf else "latency")
    elif pmc_note:
        roof["pmc_source"] = "none: " + pmc_note
    return roof


def run_simple_video(args, local_rank):
    """examples/simple_video.rs:10-56: the scene is rebuilt for every frame, so what is timed per frame is scene
    hand-off (flatten + kd build + upload) + render + the frame's arrival in host memory."""
    import numpy as np
    import rpt_amd
    from rpt_amd import _abi, make_params, scenes
    frames = max(1, args.steps if args.steps != 3 else 30)
    out, t_create, t_render = None, 0.0, 0.0
    sc0, cam0, cfg = scenes.simple_video(0)
    W, H = args.width or cfg["width"], args.height or cfg["height"]
    B = args.bounces if args.bounces is not None else cfg["max_bounces"]
    spp = args.spp or cfg["num_samples"]
    out = np.empty(W * H * 3, dtype=np.float32)
    for w in range(args.warmup):  # the first handle of a process pays HIP / module initialisation
        g = rpt_amd.GpuScene(sc0, local_rank)
        g.render_batch_reduce(cam0, make_params(W, H, B, spp), root=0, out=out)
        g.close()
    per_frame = []
    t_all = time.perf_counter()
    for f in range(frames):
        sc, cam, _ = scenes.simple_video(f)
        t0 = time.perf_counter()
        g = rpt_amd.GpuScene(sc, local_rank)
        t1 = time.perf_counter()
        g.render_batch_reduce(cam, make_params(W, H, B, spp), root=0, out=out)
        t2 = time.perf_counter()
        g.close()
        t_create += t1 - t0
        t_render += t2 - t1
        per_frame.append((t1 - t0) * 1e3)
    total = time.perf_counter() - t_all
    print(json.dumps({"metric": "frames/s (scene rebuilt per frame)", "value": frames / total, "unit": "frames/s", "n_gpus": 1,
                      "steps": frames, "warmup": args.warmup, "ms_per_step": total / frames * 1e3, "higher_is_better": True,
                      "dtype": "f64", "data": "synthetic",
                      "config": {"workload": "simple_video %dx%d, %d bounces, %d spp, %d frames, a new scene per frame "
                                             "(examples/simple_video.rs:10-56)" % (W, H, B, spp, frames),
                                 "scene_create_ms_mean": t_create / frames * 1e3, "scene_create_ms_first": per_frame[0],
                                 "scene_create_ms_median": sorted(per_frame)[len(per_frame) // 2],
                                 "render_ms_mean": t_render / frames * 1e3,
                                 "python_scene_build_ms_mean": (total - t_create - t_render) / frames * 1e3}}))


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it (the driver's form): start N ranks of this file under
    torch.distributed.run, one per GPU, and hand their output through.  Fewer than N visible devices is an error, not a
    smaller run (RPT_BENCH_BACKEND=gloo, the tests' stand-in, lets ranks share a device)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if os.environ.get("RPT_BENCH_BACKEND", "nccl") == "nccl" and have < args.gpus:
        print("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing to run a smaller job under the same name"
              % (args.gpus, have), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RPT_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Ranks:
    """this process among the ranks of the job"""

    def __init__(self, rank, world, local_rank, backend, force_lib):
        import torch
        self.rank, self.world, self.local_rank, self.backend, self.force_lib = rank, world, local_rank, backend, force_lib
        self.dev = torch.device("cuda", local_rank)
        self.cdev = self.dev if backend == "nccl" else torch.device("cpu")  # where the control-plane tensors live

    def fence(self):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()


def setup_exchange(rk, wl):
    """How this workload's frames reach rank 0.  The collective lives in the library (rptgpu_comm_init /
    rptgpu_render_batch_reduce: ncclSend / ncclRecv or ncclReduce on the library's stream, then D2H on rank 0), so no torch
    op sits in the timed path.  Only the gloo stand-in used by the tests (several ranks sharing one GPU, which RCCL
    refuses) goes through torch.distributed.  Returns (step, lib_collective, note, host_frame)."""
    import torch
    import torch.distributed as dist
    import rpt_amd
    from rpt_amd import distributed as D
    rank, world, gpu, camera = rk.rank, rk.world, wl.gpu, wl.camera
    host_frame = torch.empty(wl.W * wl.H * 3, dtype=torch.float32).pin_memory()
    host_np = host_frame.numpy()
    lib_collective = world == 1 or rk.backend == "nccl" or rk.force_lib
    note = None
    if lib_collective and world > 1:
        # Step 1, local and cheap: can THIS rank open RCCL from the library at all?  Agreed on by every rank BEFORE anyone
        # enters ncclCommInitRank — a rank that cannot would otherwise leave the others blocked in the rendezvous.
        ok = 1
        try:
            rpt_amd.GpuScene.comm_unique_id()
        except Exception as e:
            ok, note = 0, "%s: %s" % (type(e).__name__, e)
        agreed = torch.tensor([ok], dtype=torch.int32, device=rk.cdev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 1:
            # Step 2: the communicator.  ncclCommInitRank either succeeds or fails on every rank (it is itself a rendezvous)
            try:
                uid = [rpt_amd.GpuScene.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0, device=rk.cdev)
                gpu.comm_init(rank, world, uid[0])
            except Exception as e:
                ok, note = 0, "%s: %s" % (type(e).__name__, e)
            agreed = torch.tensor([ok], dtype=torch.int32, device=rk.cdev)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0:  # say why, and let torch's RCCL do the reduce
            if ok:
                gpu.comm_destroy()
            notes = [None] * world
            dist.all_gather_object(notes, note)
            note = next((x for x in notes if x), "rptgpu_comm_init failed on another rank")
            lib_collective = False
            if rank == 0:
                print("bench: library collective unavailable (%s); reducing through torch.distributed" % note, file=sys.stderr)
    frame = None if lib_collective else torch.zeros(wl.W * wl.H * 3, dtype=torch.float32, device=rk.dev)
    render_part = None if lib_collective else D.gpu_render_part(gpu, camera)

    def step():
        p = wl.params()
        if lib_collective:
            # Renderer::sample on every rank; ends with the colours in host memory on rank 0 (renderer.rs:127-128)
            gpu.render_batch_reduce(camera, p, root=0, out=host_np)
        else:
            D.render_frame_sharded(render_part, p, rank, world, frame, dst=0)
            if rank == 0:
                host_frame.copy_(frame, non_blocking=False)
        wl.step_no += 1

    return step, lib_collective, note, host_frame


def timed_steps(rk, wl, step, steps, warmup):
    """warmup untimed steps, then EXACTLY `steps` steps between two fences (barrier + device synchronize on both sides);
    the MAX over ranks of that time.  Every step is also clocked on its own on this rank (a step is synchronous: it
    returns when the frame is in host memory on rank 0), for the median / min / max beside the mean."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    rk.fence()
    wl.gpu.reset_stats()
    each = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        step()
        each.append((time.perf_counter() - t1) * 1e3)
    rk.fence()
    elapsed = time.perf_counter() - t0
    if rk.world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=rk.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = wl.gpu.stats()
    # what each rank spent where (HIP events inside rptgpu_render_batch_reduce), and the rays it traced
    mine = {"rank": rk.rank, "render_ms_per_step": st.reduce_render_ms / max(1, st.reduce_calls),
            "collective_ms_per_step": st.reduce_collective_ms / max(1, st.reduce_calls),
            "copy_ms_per_step": st.reduce_copy_ms / max(1, st.reduce_calls),
            "library_exchange_calls_completed": int(st.reduce_calls),
            "extend_rays": int(st.extend_rays), "shadow_rays": int(st.shadow_rays), "shadow_rays_traced": int(st.shadow_rays_traced),
            "scene_create_ms": wl.scene_create_ms}
    per_rank = [mine]
    if rk.world > 1:
        per_rank = [None] * rk.world
        dist.all_gather_object(per_rank, mine)
    return elapsed, each, st, per_rank


def step_spread(each, samples_per_step):
    """median / min / max of the per-step clocks, as ms and as Msamples/s"""
    s = sorted(each)
    med = s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])
    return {"ms_median": med, "ms_min": s[0], "ms_max": s[-1],
            "msamples_median": samples_per_step / med / 1e3, "msamples_min": samples_per_step / s[-1] / 1e3,
            "msamples_max": samples_per_step / s[0] / 1e3,
            "is": "each of the timed steps clocked on its own on rank 0 (a step returns when its frame is in host memory)"}


def exchange_evidence(rk, lib_collective, note, per_rank):
    """Has the library's own exchange (ncclSend / ncclRecv or ncclReduce inside rptgpu_render_batch_reduce) moved frames
    between devices IN THIS RUN?  Stated on the line because it has never been possible to try before the first multi-GPU
    run (one GPU per box in every development round)."""
    if rk.world == 1:
        return {"ranks": 1, "library_exchange_ran": False, "why": "one rank: nothing to exchange"}
    calls = [r["library_exchange_calls_completed"] for r in per_rank]
    ran = bool(lib_collective and rk.backend == "nccl" and min(calls) > 0)
    return {"ranks": rk.world, "backend": rk.backend, "library_exchange_ran": ran,
            "calls_completed_per_rank": calls,
            "why": ("rptgpu_render_batch_reduce returned success on every rank for every timed step: RCCL moved the ranks' "
                    "pixels to rank 0 inside the library" if ran else
                    ("the library's communicator could not be set up (%s): torch.distributed reduced the frames" % note if note else
                     "gloo stand-in (ranks share a device): torch.distributed reduced the frames, the library's RCCL path did not run"))}


# the secondary configs an N > 1 run times through the same sharded path: the two BASELINE assigns to 8 GPUs (1024 / 4096
# spp there), at 256 spp per step — a rank owns 1/N of the tiles, and its passes should still be as large as the device
# holds (at 16 spp per step the per-depth fixed costs of a rank's small pass were x1.6-1.8 of the ideal 1/N,
# profiles/r05_emulated_ranks.txt; pass size against throughput: profiles/r06_pass_size_ab.txt).  The committed 1-GPU lines
# of the same 256-spp workloads ride along as n1_reference (profiles/r06_<scene>_bench_line.json)
MULTI_GPU_OTHER_CONFIGS = (("fractal_spheres", 256), ("wine_glass", 256))
MULTI_GPU_OTHER_STEPS, MULTI_GPU_OTHER_WARMUP = 2, 1
SCALAR_KEYS = {"dragon": "c3_msamples", "fractal_spheres": "c4_msamples", "wine_glass": "c5_mesh_msamples", "glass": "c5_glass_spheres_msamples"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel per step (default: the config's own)")
    ap.add_argument("--scene", default=None, help="one workload only (default: cornell = the headline, plus the other configs)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--pipeline", default="auto", choices=["auto", "persistent", "wavefront"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="headline only")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not re-run a step under rocprofv3 for the counter fields")
    ap.add_argument("--cpu-spp", type=int, default=None, help="spp of the CPU-baseline sample (default and minimum: 16)")
    ap.add_argument("--dump-frame", default=None, help="rank 0 saves the last step's reduced f32 frame (.npy)")
    ap.add_argument("--fixed-samples", action="store_true", help="every step renders the same samples (tests)")
    ap.add_argument("--emulate-part-of", type=int, default=0, metavar="N",
                    help="single process: render only the tiles rank 0 would own among N ranks (no collective) and report "
                         "the step time, i.e. the per-rank cost that bounds N-GPU scaling; the JSON line is NOT a bench result")
    ap.add_argument("--collective", default="gather", choices=["gather", "reduce"],
                    help="N > 1: how rptgpu_render_batch_reduce brings the ranks' pixels to rank 0 (RptRenderParams::collective)")
    ap.add_argument("--pmc-json", default=None, help="committed rocprofv3 PMC summary to fall back to (default: profiles/<latest>_<scene>_pmc.json)")
    ap.add_argument("--pmc-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    default_run = args.scene is None
    args.scene = args.scene or "cornell"
    if args.pmc_worker:
        return pmc_worker(args)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # no launcher around us: be our own (one rank per GPU, this file again)

    import numpy as np
    import torch
    import torch.distributed as dist
    import rpt_amd
    from rpt_amd import _abi, make_params, scenes
    if args.scene not in scenes.SCENES:
        ap.error("unknown scene %r (known: %s)" % (args.scene, ", ".join(sorted(scenes.SCENES))))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # never a smaller (or larger) job than the one asked for under the same name
        print("bench.py: --gpus %d does not match the launcher's WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    # RPT_BENCH_BACKEND=gloo lets the N>1 logic be exercised with several ranks on ONE GPU
    # (ranks share device local_rank % device_count); the driver's runs use nccl = RCCL.
    backend = os.environ.get("RPT_BENCH_BACKEND", "nccl")
    # test hooks (tests/test_gpu_parity.py): make ONE rank's library collective unavailable, and try the library
    # collective under the gloo stand-in too, so that the fall-back below runs without eight GPUs
    if os.environ.get("RPT_BENCH_FAIL_COMM_RANK", "") == str(rank):
        os.environ["RPTGPU_FAIL_COMM"] = "1"
    force_lib = os.environ.get("RPT_BENCH_FORCE_LIB_COLLECTIVE", "") == "1"
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > 1 and ndev < world:
        print("bench.py: %d ranks, %d HIP device(s) visible: RCCL needs one device per rank" % (world, ndev), file=sys.stderr)
        sys.exit(2)
    local_rank = local_rank % max(1, ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    rk = Ranks(rank, world, local_rank, backend, force_lib)
    dev = rk.dev
    if args.scene == "simple_video" and world == 1:
        return run_simple_video(args, local_rank)

    torch.cuda.synchronize()
    wl = Workload(args.scene, args, rank, world, local_rank, backend)
    W, H, B, spp, gpu, camera = wl.W, wl.H, wl.B, wl.spp, wl.gpu, wl.camera

    if args.emulate_part_of > 1 and world == 1:
        import ctypes as C
        dframe = torch.zeros(W * H * 3, dtype=torch.float32, device=dev)

        def step():  # what one of N ranks does between the collectives
            p = wl.params(tile=(32, 8), part=(0, args.emulate_part_of))
            cam = camera.lower()
            _abi.check(gpu.lib.rptgpu_render_batch_device(gpu.handle, C.byref(cam), C.byref(p), C.c_void_p(dframe.data_ptr()), 1, None), gpu.handle)
            wl.step_no += 1
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        print(json.dumps({"emulated": "rank 0 of %d" % args.emulate_part_of, "scene": args.scene, "spp": spp, "ms_per_step_of_this_rank": dt,
                          "note": "per-rank render time without the reduce; compare with ms_per_step at N=1 divided by N"}))
        gpu.close()
        return

    step, lib_collective, collective_note, host_frame = setup_exchange(rk, wl)
    elapsed, each, st, per_rank = timed_steps(rk, wl, step, args.steps, args.warmup)
    rays_traced = float(sum(r["extend_rays"] + r["shadow_rays_traced"] for r in per_rank))
    rays_reference = float(sum(r["extend_rays"] + r["shadow_rays"] for r in per_rank))

    if rank == 0 and args.dump_frame:
        np.save(args.dump_frame, host_frame.numpy())
    out = None
    if rank == 0:
        total_samples = float(W) * H * spp * args.steps
        value = total_samples / elapsed / 1e6
        bytes_k, osc = accounting(wl, st)
        kernels, dominant, kern_n = kernel_table(wl, st, bytes_k)
        live = world == 1 and not args.no_live_pmc
        pmc, pmc_note = (None, "live counters are collected at N=1 only" if world > 1 else "--no-live-pmc")
        extra = ["--pipeline", args.pipeline] + (["--width", str(W), "--height", str(H)] if (args.width or args.height) else []) \
            + (["--bounces", str(B)] if args.bounces is not None else [])
        if live:
            gpu.close()  # the child renders the same workload on the same GPU: give the memory back first
            pmc, pmc_note = live_pmc(args.scene, spp, (PMC_A, PMC_B, PMC_C), extra)
        roofline = roofline_object(wl, st, kernels, dominant, kern_n, bytes_k, pmc, pmc_note, spp)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(wl, osc, 6.0, args.cpu_spp)

        out = {
            "metric": "Msamples/s", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d bounces, %d spp per step (BASELINE configs[1]: examples/cornell.rs)"
                                   % (args.scene, W, H, B, spp) if args.scene == "cornell" else
                                   "%s %dx%d, %d bounces, %d spp per step" % (args.scene, W, H, B, spp),
                       "precision_mode": "strict",
                       "pipeline": args.pipeline + ("" if args.pipeline != "auto" else " -> " + ("persistent" if kern_n.get("rpt_paths", 0) else "wavefront")),
                       "partition": "interleaved 32x8 tiles, tile_id %% %d == rank" % world,
                       "launched_by": "bench.py itself (torch.distributed.run, one rank per GPU)" if os.environ.get("RPT_BENCH_SELF_LAUNCHED") else
                                      ("an outer launcher (WORLD_SIZE in the environment)" if world > 1 else "single process"),
                       "collective": (("ncclReduce(sum, f32 framebuffer) to rank 0" if (os.environ.get("RPTGPU_COLLECTIVE") or args.collective) == "reduce" else
                                       "gather of the pixels each rank owns (ncclSend / ncclRecv, W*H*12/N bytes per rank) to rank 0")
                                      + " inside librptgpu (rptgpu_render_batch_reduce)" if lib_collective else ("torch.distributed reduce (the library's communicator could not be set up: %s)" % collective_note if collective_note else "torch.distributed reduce (gloo stand-in)")) if world > 1 else "none",
                       "timed_region": "render (f64 arithmetic) + gather of the f32 means over the ranks + D2H of the f32 frame to pinned "
                                       "host memory on rank 0 (rptgpu_render_batch_reduce; the f64 seam rptgpu_render_batch is what the "
                                       "parity tests compare, the same kernels)",
                       "rays_per_s": rays_traced / elapsed,
                       "rays_are": "rays actually traced, from the kernels' own counters: closest-hit rays + the shadow rays "
                                   "that were traversed (summed over the ranks)",
                       "reference_rays_per_s": rays_reference / elapsed,
                       "reference_rays_are": "the rays the REFERENCE casts for these samples: closest-hit + one shadow ray per hit "
                                             "and non-ambient light, including the ones whose light can only add exactly zero",
                       "scene_create_ms": wl.scene_create_ms,
                       "scene_create_is": "flatten + kd build + upload of the FIRST scene of the process (includes HIP "
                                          "initialisation when nothing else has touched the GPU); scene_create_warm_ms = the "
                                          "same scene created a second time",
                       "scene_create_warm_ms": wl.scene_create_warm_ms,
                       "wall_clock_per_frame_ms": (wl.scene_create_warm_ms or wl.scene_create_ms) + elapsed / args.steps * 1e3},
            "step_spread": step_spread(each, float(W) * H * spp),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "exchange": exchange_evidence(rk, lib_collective, collective_note, per_rank),
            "per_rank": per_rank,
            "per_rank_is": "HIP-event time per step inside rptgpu_render_batch_reduce on each rank: its own tiles / the gather "
                           "(includes waiting for the slowest rank) / on rank 0 the assembly of the frame and its D2H",
        }
        if world > 1:
            # one number for the balance of the partition, and the committed 1-GPU line of the same workload beside it, so
            # that an N-GPU line can be read on its own (the driver computes the scaling efficiency from its own N = 1 run)
            rms = [r["render_ms_per_step"] for r in per_rank]
            out["rank_render_ms"] = {"max": max(rms), "min": min(rms), "max_over_min": max(rms) / max(min(rms), 1e-9),
                                     "ideal_is": "ms_per_step at N = 1 divided by N"}
            out["n1_reference"] = committed_n1_line(args.scene, W, H, B, spp)
            if out["n1_reference"]:
                out["n1_reference"]["speedup_of_this_line"] = out["n1_reference"]["ms_per_step"] / out["ms_per_step"]
    gpu.close()  # (idempotent: at N = 1 with live counters it was closed before the child ran)

    # ---- the other BASELINE configs on the same clock.  N = 1: all four, with counters and CPU baselines.  N > 1: the two
    # that BASELINE assigns to 8 GPUs, through the same sharded path and exchange as the headline
    if default_run and not args.no_other_configs:
        others = []
        if world == 1:
            cfgs, osteps, owarm = OTHER_CONFIGS, OTHER_STEPS, OTHER_WARMUP
        else:
            cfgs, osteps, owarm = MULTI_GPU_OTHER_CONFIGS, MULTI_GPU_OTHER_STEPS, MULTI_GPU_OTHER_WARMUP
        for name, ospp in cfgs:
            t_cfg = time.perf_counter()
            entry = None
            try:
                o = Workload(name, args, rank, world, local_rank, backend, spp=ospp, is_headline=False)
                ostep, olib, onote_x, _ = setup_exchange(rk, o)
                dt, oeach, ost, oper_rank = timed_steps(rk, o, ostep, osteps, owarm)
                if rank == 0:
                    ob, oosc = accounting(o, ost)
                    ok, odom, okn = kernel_table(o, ost, ob)
                o.gpu.close()
                if rank == 0:
                    opmc, onote = (None, "--no-live-pmc" if world == 1 else "live counters are collected at N=1 only") \
                        if (args.no_live_pmc or world > 1) else live_pmc(name, min(ospp, PMC_MAX_SPP), (PMC_A,))
                    oroof = roofline_object(o, ost, ok, odom, okn, ob, opmc, onote, min(ospp, PMC_MAX_SPP))
                    oroof["kernels"] = {k: {"launches": v["launches"], "avg_ms": v["avg_ms"], "total_ms": v["total_ms"]} for k, v in ok.items()}
                    ocpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(o, None, 4.0, args.cpu_spp)
                    rays_t = float(sum(r["extend_rays"] + r["shadow_rays_traced"] for r in oper_rank))
                    rays_r = float(sum(r["extend_rays"] + r["shadow_rays"] for r in oper_rank))
                    entry = {"workload": "%s %dx%d, %d bounces, %d spp per step" % (name, o.W, o.H, o.B, ospp),
                             "value": float(o.W) * o.H * ospp * osteps / dt / 1e6, "unit": "Msamples/s", "n_gpus": world,
                             "ms_per_step": dt / osteps * 1e3, "steps": osteps, "warmup": owarm, "spp": ospp,
                             "step_spread": step_spread(oeach, float(o.W) * o.H * ospp),
                             "scene_create_ms": o.scene_create_ms,
                             "rays_per_s": rays_t / dt, "reference_rays_per_s": rays_r / dt,
                             "roofline": oroof, "cpu_baseline": ocpu,
                             "wall_s_of_this_entry": None}
                    if world > 1:
                        rms = [r["render_ms_per_step"] for r in oper_rank]
                        entry["rank_render_ms"] = {"max": max(rms), "min": min(rms)}
                        entry["exchange"] = exchange_evidence(rk, olib, onote_x, oper_rank)
                        entry["n1_reference"] = committed_n1_line(name, o.W, o.H, o.B, ospp)
            except Exception as e:  # one config must not cost the bench line
                if world > 1:
                    raise  # (with several ranks a failure on one of them cannot be skipped over: the others would wait)
                entry = {"workload": name, "error": "%s: %s" % (type(e).__name__, e)}
            if rank == 0:
                entry["wall_s_of_this_entry"] = time.perf_counter() - t_cfg
                others.append(entry)
        if rank == 0:
            out["other_configs"] = others
            for e in others:  # one top-level scalar per config, so that a reader who keeps only scalars keeps them
                key = SCALAR_KEYS.get(str(e.get("workload", "")).split(" ")[0])
                if key and "value" in e:
                    out[key] = e["value"]
                    out[key + "_spp_per_step"] = e["spp"]
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
