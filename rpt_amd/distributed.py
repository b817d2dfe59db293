"""Multi-GPU: pixel tiles shard across the ranks of one node, one collective per frame.

The hot path shards naturally (reference src/renderer.rs:118-127: rows are independent; SURVEY
§8e).  Rank r renders the tiles with tile_id % world == r (interleaved 32x8 tiles balance sky
against geometry) and writes 0 elsewhere; because Philox is keyed by (pixel, sample) the
frame does not depend on the partition, and the full frame is the SUM of the ranks' frames — one
`reduce` of the f32 framebuffer to rank 0 (RCCL over xGMI when the backend is "nccl"; the same
code runs over gloo on CPU in tests/test_distributed_cpu.py) or, lighter by the factor `world`, one
GATHER of the pixels each rank owns (`gather_frame`: what librptgpu's rptgpu_render_batch_reduce
does with ncclSend / ncclRecv; the tiles are disjoint, so nothing is added).  No other exchange
exists on this path.
"""
import copy

import torch
import torch.distributed as dist

TILE = (32, 8)


def shard_params(params, rank, world, tile=TILE):
    """The same batch, restricted to this rank's tiles."""
    p = copy.copy(params)
    p.tile_width, p.tile_height = int(tile[0]), int(tile[1])
    p.part_index, p.part_count = int(rank), int(world)
    return p


def owned_pixels(width, height, rank, world, tile=TILE):
    """Pixel indices y*width+x of the tiles with tile_id % world == rank, ascending (numpy int64)."""
    import numpy as np
    tw, th = int(tile[0]), int(tile[1])
    tiles_x = (width + tw - 1) // tw
    ys, xs = np.mgrid[0:height, 0:width]
    tid = (ys // th) * tiles_x + xs // tw
    return np.flatnonzero((tid % world == rank).ravel()) if world > 1 else np.arange(width * height)


def gather_frame(frame, width, height, rank, world, dst=0, tile=TILE):
    """The exchange as a gather: every rank sends only the pixels it owns, `dst` puts them in place.  `frame` holds
    this rank's pixels (anything elsewhere); on `dst` it is complete afterwards.  Exact: no value is added to another."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return frame
    f3 = frame.view(-1, 3)
    mine = torch.from_numpy(owned_pixels(width, height, rank, world, tile)).to(frame.device)
    packed = f3.index_select(0, mine).contiguous()
    if rank == dst:
        for r in range(world):
            if r == dst:
                continue
            idx = torch.from_numpy(owned_pixels(width, height, r, world, tile)).to(frame.device)
            buf = torch.empty((idx.numel(), 3), dtype=frame.dtype, device=frame.device)
            if idx.numel():
                dist.recv(buf, src=r)
                f3.index_copy_(0, idx, buf)
    elif packed.numel():
        dist.send(packed, dst=dst)
    return frame


def reduce_frame(frame, dst=0):
    """Sum the ranks' frames into `dst` (every pixel is non-zero on exactly one rank, so the sum
    is exact).  `frame` is a torch tensor (cuda for nccl/RCCL, cpu for gloo)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame


def render_frame_sharded(render_part, params, rank, world, frame, dst=0):
    """render_part(params_for_this_rank, frame) fills `frame` in place (zeros outside the rank's
    tiles); then the frames are reduced to `dst`.  Returns `frame` (complete on `dst`)."""
    render_part(shard_params(params, rank, world), frame)
    return reduce_frame(frame, dst)


def gpu_render_part(gpu_scene, camera):
    """render_part for a GpuScene writing straight into a CUDA f32/f64 tensor."""

    def run(params, frame):
        assert frame.is_cuda and frame.is_contiguous() and frame.numel() == params.width * params.height * 3
        stream = torch.cuda.current_stream(frame.device).cuda_stream
        gpu_scene.render_batch_device(camera, params, frame.data_ptr(), out_is_f32=(frame.dtype == torch.float32),
                                      stream=stream)

    return run
