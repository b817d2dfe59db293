"""The work-item hand-out of rpt_paths (kernels/paths.inc fetch_item: a wave-level pool refilled by ONE atomic on the global
counter, batches sized by the host and guided towards the end) restated in Python and run as a discrete simulation: whatever
the interleaving of the waves' requests, every item is handed to exactly one lane, a lane is told "no item" only after the
counter has passed the end, and the counter never exceeds what api_render.cpp leaves room for below 2^32 (items + 8 x threads).
The device code is tested through its results (tests/test_gpu_parity.py::test_work_item_pools_are_scheduling_only); this
pins the arithmetic the comment in paths.inc argues about."""
import random

import pytest


class Wave:
    def __init__(self):
        self.next = self.end = self.seen = 0
        self.dead = 0


def host_batch(n_items, nblocks, option=0):
    # kernels/launch.inc launch_paths
    return option if option else min(256, max(n_items // (max(1, nblocks) * 32), 16))


def fetch(wave, counter, n_items, nthreads, batch, n_asking):
    """one call of fetch_item for a wave in which n_asking lanes want an item: the indices they get (None = no item left)"""
    avail = (wave.end - wave.next) & 0xFFFFFFFF
    idx = [wave.next + r for r in range(n_asking)]
    if n_asking > avail:
        left = n_items - min(n_items, wave.seen)
        more = max(n_asking - avail, min(batch, left // (nthreads >> 4)))
        base = counter[0]
        counter[0] += more
        assert counter[0] < 1 << 32
        for r in range(avail, n_asking):
            idx[r] = base + (r - avail)
        wave.next = base + (n_asking - avail)
        wave.end = base + more
        wave.seen = wave.end
    else:
        wave.next += n_asking
    out = [i if i < n_items else None for i in idx]
    wave.dead += sum(1 for i in idx if i >= n_items)
    return out


@pytest.mark.parametrize("n_items,nblocks,option", [(1, 4, 0), (63, 2, 0), (5000, 8, 0), (200000, 64, 0), (200000, 64, 1),
                                                     (200000, 64, 256), (4097, 16, 7), (30000, 2048, 0)])
def test_every_item_goes_to_exactly_one_lane(n_items, nblocks, option):
    rnd = random.Random(n_items * 31 + nblocks)
    nthreads = nblocks * 64
    batch = host_batch(n_items, nblocks, option)
    waves = [Wave() for _ in range(nblocks)]
    lanes_done = [[False] * 64 for _ in range(nblocks)]
    counter = [0]
    seen = bytearray(n_items)
    handed = 0
    live = list(range(nblocks))
    while live:
        w = rnd.choice(live)
        asking = [l for l in range(64) if not lanes_done[w][l] and rnd.random() < 0.3]
        if not asking:
            if all(lanes_done[w]):
                live.remove(w)
            continue
        got = fetch(waves[w], counter, n_items, nthreads, batch, len(asking))
        for lane, i in zip(asking, got):
            if i is None:
                lanes_done[w][lane] = True  # (the kernel: done / exhausted — the lane never asks again)
                assert counter[0] >= n_items
            else:
                assert not seen[i]
                seen[i] = 1
                handed += 1
    assert handed == n_items and all(seen)
    # dead items: every lane asks at most once past the end, plus the wave's last guided claim
    assert max(w.dead for w in waves) <= 64
    assert counter[0] <= n_items + nblocks * (64 + batch)
    assert counter[0] <= n_items + 8 * nthreads  # api_render.cpp's item_limit leaves 8 x the grid's threads (>= this grid's)


def test_batches_shrink_towards_the_end():
    # guided self-scheduling: the last claims are small, so that no wave sits on a big pool while others have nothing left
    n_items, nblocks = 1 << 20, 256
    nthreads, batch = nblocks * 64, host_batch(1 << 20, 256)
    assert batch == 128
    waves = [Wave() for _ in range(nblocks)]
    counter = [0]
    claims = []
    w = 0
    while counter[0] < n_items:
        before = counter[0]
        fetch(waves[w % nblocks], counter, n_items, nthreads, batch, 4)  # every wave in turn, four lanes asking
        if counter[0] != before:
            claims.append((before, counter[0] - before))
        w += 1
    assert claims[0][1] == batch
    tail = [c for b, c in claims if n_items - b < 4 * nblocks * 8]  # fewer than 8 items per (4 x wave) left
    # (a wave guides itself by the counter as of ITS last claim: stale by what the others claimed since, hence up to ~2x)
    assert tail and max(tail) <= 16
