"""The N>1 path on CPU: two gloo ranks shard the frame with rpt_amd.distributed exactly as
bench.py does on GPUs (same shard_params / reduce_frame), with the CPU oracle standing in for the
renderer; rank 0's reduced frame must equal the unsharded frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_ffi as O
    from rpt_amd import distributed as D
    from rpt_amd import make_params, scenes
    scene, cam, _ = scenes.cornell()
    osc = O.OracleScene(scene)
    params = make_params(80, 45, 3, 2, seed=31)

    def render_part(p, frame):
        frame.copy_(torch.from_numpy(osc.render(cam, p, threads=1).reshape(-1)))

    for dtype in (torch.float64, torch.float32):
        frame = torch.zeros(80 * 45 * 3, dtype=dtype)
        D.render_frame_sharded(render_part, params, rank, world, frame, dst=0)
        dist.barrier()
        # the same frame through the gather (what librptgpu's collective does: only owned pixels travel)
        gframe = torch.full((80 * 45 * 3,), float("nan"), dtype=dtype)
        part = torch.zeros(80 * 45 * 3, dtype=dtype)
        render_part(D.shard_params(params, rank, world), part)
        own = torch.from_numpy(D.owned_pixels(80, 45, rank, world))
        gframe.view(-1, 3)[own] = part.view(-1, 3)[own]
        D.gather_frame(gframe, 80, 45, rank, world, dst=0)
        dist.barrier()
        if rank == 0:
            full = torch.from_numpy(osc.render(cam, params, threads=1).reshape(-1)).to(dtype)
            ok = bool((frame == full).all()) and bool((gframe == full).all())
            own = D.shard_params(params, 0, world)
            own_frame = osc.render(cam, own, threads=1)
            covered = float((own_frame != 0).any(axis=1).mean())
            with open(out_path, "a") as f:
                f.write("%s %d %.4f\n" % (str(dtype), ok, covered))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_frame_equals_full_frame(tmp_path):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    lines = out.read_text().split("\n")[:-1]
    assert len(lines) == 2
    for ln in lines:
        _, ok, covered = ln.split()
        assert ok == "1"
        assert 0.3 < float(covered) < 0.7  # interleaved tiles: about half the pixels per rank


def test_shard_params_partition_is_disjoint_and_complete():
    sys.path.insert(0, ROOT)
    from rpt_amd import distributed as D
    from rpt_amd import make_params
    p = make_params(100, 37, 1, 1)
    W, H = 100, 37
    seen = np.zeros(W * H, dtype=int)
    for r in range(8):
        q = D.shard_params(p, r, 8)
        tw, th = q.tile_width, q.tile_height
        tiles_x = (W + tw - 1) // tw
        ys, xs = np.mgrid[0:H, 0:W]
        tile = (ys // th) * tiles_x + xs // tw
        seen += ((tile % q.part_count) == q.part_index).ravel()
    assert (seen == 1).all()
    assert p.part_count == 1  # the original is untouched
