#!/usr/bin/env python
"""bench.py — Msamples/s of the rpt hot path on MI355X (BASELINE.json metric).

A step = one complete frame of BASELINE configs[1]: examples/cornell.rs geometry, 1920x1080,
8 bounces, 512 samples per pixel (override with --spp), through the C ABI in parity mode (IEEE f64,
no FMA contraction).  The timed region is what `Renderer::sample` covers: ray generation
through the last bounce, the reduce over ranks, and the write of the W*H mean colours into HOST memory
(rank 0); scene construction (kd build + upload) is outside it and reported as `scene_create_ms` /
`wall_clock_per_frame_ms`.  With N GPUs rank r renders the tiles tile_id % N == r of the SAME frame
(strong scaling) and the f32 framebuffers are summed to rank 0 by one RCCL reduce per step.

    python bench.py                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --scene dragon --spp 16          # another BASELINE config at ITS frame size
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import rpt_amd  # noqa: E402
from rpt_amd import _abi, make_params, scenes  # noqa: E402
from rpt_amd import distributed as D  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# SURVEY §8d accounting constants, f64 layout (every float field of the f32 layout doubles)
RAY_IO = 2 * (32 + 16)
INST, NODE, LEAF, REF, TRI, PLANE = 96, 16, 16, 4, 72, 32
SHADE_GEOM, MATERIAL = 72, 64
STATE_RW, FOLD_RW = 2 * 192, 2 * 48
ENV = 4 * 32
FB = 24


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel per step (default: the config's own)")
    ap.add_argument("--scene", default="cornell", choices=sorted(scenes.SCENES))
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--pipeline", default="auto", choices=["auto", "persistent", "wavefront"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=None, help="spp of the CPU-baseline sample (default: ~15 s of CPU work)")
    ap.add_argument("--dump-frame", default=None, help="rank 0 saves the last step's reduced f32 frame (.npy)")
    ap.add_argument("--fixed-samples", action="store_true", help="every step renders the same samples (tests)")
    ap.add_argument("--emulate-part-of", type=int, default=0, metavar="N",
                    help="single process: render only the tiles rank 0 would own among N ranks (no collective) and report "
                         "the step time, i.e. the per-rank cost that bounds N-GPU scaling; the JSON line is NOT a bench result")
    ap.add_argument("--pmc-json", default=None, help="rocprofv3 PMC summary of this workload (default: profiles/<latest>_<scene>_pmc.json)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RPT_BENCH_BACKEND=gloo lets the N>1 logic be exercised with several ranks on ONE GPU
    # (ranks share device local_rank % device_count); the driver's runs use nccl = RCCL.
    backend = os.environ.get("RPT_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, "--gpus must match the launcher's world size"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    scene, camera, cfg = scenes.SCENES[args.scene]()
    W = args.width or cfg["width"]
    H = args.height or cfg["height"]
    B = args.bounces if args.bounces is not None else cfg["max_bounces"]
    spp = args.spp or cfg["num_samples"]
    precision = _abi.RPT_PRECISION_F64_STRICT

    pipe_flag = {"auto": 0, "persistent": _abi.RPT_FLAG_PERSISTENT, "wavefront": _abi.RPT_FLAG_WAVEFRONT}[args.pipeline]
    torch.cuda.synchronize()
    t_create = time.perf_counter()
    gpu = rpt_amd.GpuScene(scene, local_rank)  # flatten + kd build (reference rule) + upload
    scene_create_ms = (time.perf_counter() - t_create) * 1e3
    host_frame = torch.empty(W * H * 3, dtype=torch.float32).pin_memory()
    host_np = host_frame.numpy()
    # The collective lives in the library (rptgpu_comm_init / rptgpu_render_batch_reduce: ncclReduce on the library's
    # stream, then D2H on rank 0), so no torch op sits in the timed path.  Only the gloo stand-in used by the tests
    # (several ranks sharing one GPU, which RCCL refuses) goes through torch.distributed.
    lib_collective = world == 1 or backend == "nccl"
    collective_note = None
    if lib_collective and world > 1:
        ok = 1
        try:
            uid = [rpt_amd.GpuScene.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0, device=dev)
            gpu.comm_init(rank, world, uid[0])
        except Exception as e:  # e.g. librccl.so not loadable from the library: say so, and let torch's RCCL do the reduce
            ok, collective_note = 0, "%s: %s" % (type(e).__name__, e)
        agreed = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0:
            if ok:
                gpu.comm_destroy()
            notes = [None] * world
            dist.all_gather_object(notes, collective_note)
            collective_note = next((x for x in notes if x), "rptgpu_comm_init failed on another rank")
            lib_collective = False
            if rank == 0:
                print("bench: library collective unavailable (%s); reducing through torch.distributed" % collective_note,
                      file=sys.stderr)
    frame = None if lib_collective else torch.zeros(W * H * 3, dtype=torch.float32, device=dev)
    render_part = None if lib_collective else D.gpu_render_part(gpu, camera)
    step_no = [0]

    def step():
        p = make_params(W, H, B, spp, seed=0x52505447, sample_index_base=0 if args.fixed_samples else step_no[0] * spp,
                        precision=precision, flags=_abi.RPT_FLAG_PROFILE_KERNELS | pipe_flag)
        if lib_collective:
            # Renderer::sample on every rank; ends with the colours in host memory on rank 0 (renderer.rs:127-128)
            gpu.render_batch_reduce(camera, p, root=0, out=host_np)
        else:
            D.render_frame_sharded(render_part, p, rank, world, frame, dst=0)
            if rank == 0:
                host_frame.copy_(frame, non_blocking=False)
        step_no[0] += 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.emulate_part_of > 1 and world == 1:
        out32 = host_np

        def step():  # noqa: F811 — what one of N ranks does between the collectives
            p = make_params(W, H, B, spp, seed=0x52505447, sample_index_base=step_no[0] * spp, precision=precision,
                            flags=_abi.RPT_FLAG_PROFILE_KERNELS | pipe_flag, tile=(32, 8), part=(0, args.emulate_part_of))
            cam = camera.lower()
            import ctypes as C
            _abi.check(gpu.lib.rptgpu_render_batch_device(gpu.handle, C.byref(cam), C.byref(p), C.c_void_p(dframe.data_ptr()), 1, None), gpu.handle)
            step_no[0] += 1
        dframe = torch.zeros(W * H * 3, dtype=torch.float32, device=dev)
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

    for _ in range(args.warmup):
        step()
    fence()
    gpu.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = gpu.stats()

    if rank == 0 and args.dump_frame:
        np.save(args.dump_frame, host_frame.numpy())
    if rank == 0:
        total_samples = float(W) * H * spp * args.steps
        value = total_samples / elapsed / 1e6
        NK = _abi.RPT_K_COUNT
        names = [gpu.lib.rptgpu_kernel_name(k).decode() for k in range(NK)]
        kern_ms = {names[k]: st.kernel_ms[k] for k in range(NK)}
        kern_n = {names[k]: int(st.kernel_launches[k]) for k in range(NK)}
        names = [k for k in names if kern_n[k] > 0]

        # visit counts of the reference algorithm for this workload, from the instrumented oracle
        # on a bounded sample (1 spp on 1/8 of the tiles); they scale linearly with samples
        from oracle import oracle_ffi as O
        osc = O.OracleScene(scene)
        pc = make_params(W, H, B, 1, seed=0x52505447, tile=(32, 8), part=(0, 8))
        _, cnt = osc.render(camera, pc, threads=0, counters=True)
        rank_samples = float(st.samples)  # what THIS rank traced in the timed region
        scale = rank_samples / max(1, cnt["samples"])
        bytes_k = {k: v * scale for k, v in algorithmic_bytes(cnt, scene.environment.hdri is not None).items()}
        bytes_k["rpt_tree_enter+sort"] = 0.0
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
                          "alg_GB_per_launch": bytes_k[k] / n / 1e9,
                          "achieved_GBs": (bytes_k[k] / 1e9) / (ms / 1e3) if ms > 0 else None}
        ach = kernels[dominant]["achieved_GBs"]
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": None,
                    "alg_bytes_per_sample": bytes_k["rpt_paths"] / rank_samples,
                    "achieved_is": "SURVEY §8d algorithmic bytes of the REFERENCE traversal / launch time (an accounting "
                                   "figure); hbm_measured_GBs is what the fabric counters saw",
                    "kernels": kernels}
        # measured counters of the same workload (rocprofv3 PMC passes, scripts/profile.sh + summarize_profile.py)
        pmc_path = args.pmc_json
        if pmc_path is None:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc.json" % args.scene)))
            pmc_path = cands[-1] if cands else None
        if pmc_path and os.path.exists(pmc_path):
            try:
                pj = json.load(open(pmc_path))
                k = pj.get("kernels", {}).get(dominant)
                if k:
                    n_l = max(1, kern_n[dominant])
                    roofline["pmc_source"] = "%s (%s)" % (os.path.relpath(pmc_path, ROOT), pj.get("workload", ""))
                    if k.get("hbm_bytes_per_sample") is not None:
                        roofline["traffic"] = k["hbm_bytes_per_sample"] * rank_samples / n_l
                        roofline["hbm_measured_GBs"] = roofline["traffic"] / 1e9 / (kernels[dominant]["avg_ms"] / 1e3)
                        roofline["hbm_frac"] = roofline["hbm_measured_GBs"] / HBM_PEAK_GBS
                    for f in ("valu_busy", "lanes_active", "valu_frac", "wait_frac", "l2_hit_rate", "vmem_latency_cycles"):
                        if k.get(f) is not None:
                            roofline[f] = k[f]
                    # what bounds the kernel, from the data: the larger of the HBM fraction and the useful-lane
                    # VALU fraction names the roof; a kernel whose waves wait most of their cycles with neither
                    # near its roof is latency-bound
                    hf, vf, wf = roofline.get("hbm_frac") or 0.0, k.get("valu_busy") or 0.0, k.get("wait_frac") or 0.0
                    roofline["bound"] = "hbm" if hf >= max(vf, 0.5) else ("valu" if vf >= 0.6 or vf > wf else "latency")
            except Exception as e:  # a malformed summary must not cost the bench line
                roofline["pmc_error"] = str(e)

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            ncores, raw = host_cpus()
            L, how = O.baseline_lib(native=True)
            fast = O.OracleScene(scene, L)
            # calibrate on 1 spp of 1/8 of the frame, then size the sample for ~15 s of wall time
            pcal = make_params(W, H, B, 1, seed=0x52505447, tile=(32, 8), part=(0, 8))
            t1 = time.perf_counter()
            fast.render(camera, pcal, threads=ncores)
            rate = (W * H / 8.0) / max(1e-6, time.perf_counter() - t1)
            cpu_spp = args.cpu_spp or int(min(64, max(1, round(rate * 15.0 / (W * H)))))
            pcpu = make_params(W, H, B, cpu_spp, seed=0x52505447)
            t1 = time.perf_counter()
            fast.render(camera, pcpu, threads=ncores)
            dt = time.perf_counter() - t1
            # the instrumented checker build on a quarter of that sample, for the record
            pins = make_params(W, H, B, max(1, cpu_spp // 4), seed=0x52505447)
            t1 = time.perf_counter()
            osc.render(camera, pins, threads=ncores)
            dti = time.perf_counter() - t1
            cpu = {"value": W * H * cpu_spp / dt / 1e6, "unit": "Msamples/s", "cores": ncores, "kind": "port",
                   "sample": "%s %dx%d, %d bounces, %d spp (%.1f s wall): C++ restatement of rpt's rayon path (oracle/, %s), one task "
                             "per row claimed dynamically by %d std::threads on %s (%d logical CPUs)"
                             % (args.scene, W, H, B, cpu_spp, dt, how, ncores, cpu_model(), raw),
                   "instrumented_checker_build": {"value": W * H * pins.iterations / dti / 1e6, "unit": "Msamples/s",
                                                  "note": "liboracle.so with visit counters compiled in (x86-64-v3)"}}

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
                       "collective": ("ncclReduce(sum, f32 framebuffer) to rank 0 inside librptgpu (rptgpu_render_batch_reduce)" if lib_collective else ("torch.distributed reduce (the library's communicator could not be set up: %s)" % collective_note if collective_note else "torch.distributed reduce (gloo stand-in)")) if world > 1 else "none",
                       "timed_region": "render + reduce + D2H of the f32 frame to pinned host memory on rank 0",
                       "rays_per_s": (st.extend_rays + st.shadow_rays) / elapsed * (world if world > 1 else 1),
                       "scene_create_ms": scene_create_ms,
                       "wall_clock_per_frame_ms": scene_create_ms + elapsed / args.steps * 1e3},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    gpu.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
