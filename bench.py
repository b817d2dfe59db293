#!/usr/bin/env python
"""bench.py — Msamples/s of the rpt hot path on MI355X (BASELINE.json metric).

A step = one complete frame of BASELINE configs[1]: examples/cornell.rs geometry, 1920x1080,
8 bounces, 512 samples per pixel (override with --spp), through the C ABI
(rptgpu_render_batch_device) in parity mode (IEEE f64, no FMA contraction).  The frame stays in
HBM inside the timed region; with N GPUs rank r renders the tiles tile_id % N == r of the SAME
frame (strong scaling) and the f32 framebuffers are summed to rank 0 by one RCCL reduce per step.

    python bench.py                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
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
    return parts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel per step (default: the config's 512)")
    ap.add_argument("--scene", default="cornell", choices=sorted(scenes.SCENES))
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--mode", default="strict", choices=["strict", "fast"])
    ap.add_argument("--pipeline", default="auto", choices=["auto", "persistent", "wavefront"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=16)
    ap.add_argument("--dump-frame", default=None, help="rank 0 saves the last step's reduced f32 frame (.npy)")
    ap.add_argument("--fixed-samples", action="store_true", help="every step renders the same samples (tests)")
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
    W = args.width or (cfg["width"] if args.scene == "cornell" else min(cfg["width"], 1920))
    H = args.height or (cfg["height"] if args.scene == "cornell" else min(cfg["height"], 1080))
    B = args.bounces if args.bounces is not None else cfg["max_bounces"]
    spp = args.spp or cfg["num_samples"]
    precision = _abi.RPT_PRECISION_F64_STRICT if args.mode == "strict" else _abi.RPT_PRECISION_F64_FAST

    pipe_flag = {"auto": 0, "persistent": _abi.RPT_FLAG_PERSISTENT, "wavefront": _abi.RPT_FLAG_WAVEFRONT}[args.pipeline]
    gpu = rpt_amd.GpuScene(scene, local_rank)
    frame = torch.zeros(W * H * 3, dtype=torch.float32, device=dev)
    render_part = D.gpu_render_part(gpu, camera)
    step_no = [0]

    def step():
        p = make_params(W, H, B, spp, seed=0x52505447, sample_index_base=0 if args.fixed_samples else step_no[0] * spp,
                        precision=precision, flags=_abi.RPT_FLAG_PROFILE_KERNELS | pipe_flag)
        D.render_frame_sharded(render_part, p, rank, world, frame, dst=0)
        step_no[0] += 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
        np.save(args.dump_frame, frame.cpu().numpy())
    if rank == 0:
        total_samples = float(W) * H * spp * args.steps
        value = total_samples / elapsed / 1e6
        names = [gpu.lib.rptgpu_kernel_name(k).decode() for k in range(6)]
        kern_ms = {names[k]: st.kernel_ms[k] for k in range(6)}
        kern_n = {names[k]: int(st.kernel_launches[k]) for k in range(6)}
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
        dominant = max(kern_ms, key=kern_ms.get)
        kernels = {}
        for k in names:
            ms, n = kern_ms[k], max(1, kern_n[k])
            kernels[k] = {"launches": kern_n[k], "avg_ms": ms / n, "total_ms": ms,
                          "alg_GB_per_launch": bytes_k[k] / n / 1e9,
                          "achieved_GBs": (bytes_k[k] / 1e9) / (ms / 1e3) if ms > 0 else None}
        ach = kernels[dominant]["achieved_GBs"]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get(dominant, {})
                if "hbm_bytes_per_sample" in tj:  # measured per sample (PMC run), scaled to this launch size
                    traffic = tj["hbm_bytes_per_sample"] * rank_samples / max(1, kern_n[dominant])
                else:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic,
                    "alg_bytes_per_sample": bytes_k["rpt_paths"] / rank_samples,
                    "kernels": kernels,
                    "note": "compute/latency-bound f64 scalar work on an L2-resident scene; HBM fraction is low by construction (DESIGN.md)"}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            ncores = os.cpu_count() or 1
            pcpu = make_params(W, H, B, args.cpu_spp, seed=0x52505447)
            t1 = time.perf_counter()
            osc.render(camera, pcpu, threads=ncores)
            dt = time.perf_counter() - t1
            cpu = {"value": W * H * args.cpu_spp / dt / 1e6, "unit": "Msamples/s", "cores": ncores, "kind": "port",
                   "sample": "%s %dx%d, %d bounces, %d spp (%.1f s wall): C++ restatement of rpt's rayon path "
                             "(oracle/, one task per row over %d std::threads)" % (args.scene, W, H, B, args.cpu_spp, dt, ncores)}

        out = {
            "metric": "Msamples/s", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d bounces, %d spp per step (BASELINE configs[1]: examples/cornell.rs)"
                                   % (args.scene, W, H, B, spp) if args.scene == "cornell" else
                                   "%s %dx%d, %d bounces, %d spp per step" % (args.scene, W, H, B, spp),
                       "precision_mode": args.mode, "pipeline": args.pipeline + ("" if args.pipeline != "auto" else " -> " + ("persistent" if kern_n.get("rpt_paths", 0) else "wavefront")), "partition": "interleaved 32x8 tiles, tile_id %% %d == rank" % world,
                       "collective": "RCCL reduce(sum) of the f32 framebuffer to rank 0" if world > 1 else "none",
                       "rays_per_s": (st.extend_rays + st.shadow_rays) / elapsed * (world if world > 1 else 1)},
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
