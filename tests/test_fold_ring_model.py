"""The per-lane bookkeeping of rpt_paths (kernels/paths.inc) — the record ring, the fold walker, the queue of parked
environment lookups — restated in Python and run over random path histories.  It pins what the comments there argue:

  * the ring of rpt_fold_ring_slots(B) = 3 B + 2 slots is never overrun: a record or a header is only ever written
    into a slot nothing unfolded lives in, with and without parked lookups (the guard "parked + running records >=
    max_bounces forces a drain" is what keeps that true: switched off in the model, the simulation does overrun);
  * the walker never reads a header that has not been written (it stops at the oldest parked path's header);
  * every path is folded exactly once, level by level from the deepest, and the paths that hold records leave the
    ring in the order they entered it.

The device code is tested through its results (bit-equal frames at bounce limits 1-3 and 12, tests/test_gpu_parity.py);
this is the invariant side of the same logic, at path histories a small frame never produces."""
import random

import pytest

FREE = None


class Lane:
    def __init__(self, B, K, guard=True):
        self.B, self.K, self.guard = B, K, guard
        self.ring = 3 * B + 2
        self.slot = [FREE] * self.ring   # ("rec", path, level) | ("hdr", path)
        self.cb = 0                       # header slot of the running (or next) path: fold_st >> 16
        self.wk = 0                       # levels the walker still owes: fold_st & 0xffff
        self.wD = self.wb = self.wpath = 0
        self.parked = []                  # (path, depth, header slot)
        self.park_hd = None
        self.park_recs = 0
        self.in_path = False
        self.depth = 0
        self.path = -1                    # id of the running path
        self.next_path = 0
        self.folded = []                  # (path, level) in fold order
        self.finished = []                # paths whose sample is complete, in completion order
        self.meta = {}                    # path -> depth at its end

    def wrap(self, p):
        return p - self.ring if p >= self.ring else p

    def write(self, pos, what):
        assert self.slot[pos] is FREE, ("overrun", pos, self.slot[pos], what)
        self.slot[pos] = what

    def path_ended(self, path, depth, b):
        self.meta[path] = depth
        if depth == 0:
            self.finished.append(path)
        elif self.wk == 0:               # walker idle: it takes the path from registers
            self.wD, self.wb, self.wpath, self.wk = depth, b, path, depth
        else:                            # the path waits: header into its slot
            self.write(b, ("hdr", path))

    def advance(self, b, depth):
        self.cb = self.wrap(b + depth + 1)

    def park_push(self):
        b = self.cb
        self.parked.append((self.path, self.depth, b))
        if self.depth:
            if self.park_hd is None:
                self.park_hd = b
            self.park_recs += self.depth
            self.advance(b, self.depth)

    def drain(self):
        for path, depth, b in self.parked:  # the older path first
            self.path_ended(path, depth, b)
        self.parked, self.park_hd, self.park_recs = [], None, 0

    def iteration(self, outcome, others_drain, done=False):
        """outcome: "escape" | "end" | "scatter" (a record, the path goes on; only below the bounce limit)"""
        esc = ending = False
        if not self.in_path and not done:
            self.in_path, self.depth, self.path = True, 0, self.next_path
            self.next_path += 1
        ran = self.in_path
        if ran:
            if outcome == "scatter" and self.depth < self.B:
                self.write(self.wrap(self.cb + 1 + self.depth), ("rec", self.path, self.depth))
                self.depth += 1
            elif outcome == "escape":
                esc = ending = True
            else:
                ending = True
        # ---- the parked lookups (PARK instantiations; K = 0 models the others: lookups on the spot)
        if self.K:
            if esc and len(self.parked) < self.K:
                self.park_push()
                esc = ending = False
                self.in_path = False
            back = bool(self.parked) and (done or ending)
            if self.guard and self.park_recs and self.in_path and not ending:
                back = self.park_recs + self.depth >= self.B
            if back or others_drain:      # (the whole wave drains when any of its lanes must, or enough requests wait)
                self.drain()
            if esc:                       # (the queue was full: it is empty now)
                self.park_push()
                ending = False
                self.in_path = False
        if ending:
            b = self.cb
            self.path_ended(self.path, self.depth, b)
            if self.depth:
                self.advance(b, self.depth)
            self.in_path = False
        # ---- one step of the walker
        if self.wk:
            pos = self.wrap(self.wb + self.wk)
            assert self.slot[pos] == ("rec", self.wpath, self.wk - 1), ("walker reads", pos, self.slot[pos], self.wpath, self.wk - 1)
            self.slot[pos] = FREE
            self.folded.append((self.wpath, self.wk - 1))
            nh = self.wrap(self.wb + self.wD + 1)
            last = self.wk == 1
            more = last and nh != self.cb and nh != self.park_hd
            if not last:
                self.wk -= 1
            else:
                self.finished.append(self.wpath)
                if more:
                    assert self.slot[nh] is not FREE and self.slot[nh][0] == "hdr", ("unwritten header", nh, self.slot[nh])
                    path = self.slot[nh][1]
                    self.slot[nh] = FREE
                    self.wD, self.wb, self.wpath, self.wk = self.meta[path], nh, path, self.meta[path]
                else:
                    self.wk = 0


def run(B, K, seed, p_escape, p_end, p_drain, iterations=4000, guard=True):
    rnd = random.Random(seed)
    lane = Lane(B, K, guard)
    for _ in range(iterations):
        r = rnd.random()
        outcome = "escape" if r < p_escape else ("end" if r < p_escape + p_end else "scatter")
        lane.iteration(outcome, rnd.random() < p_drain)
    for _ in range(8 * B + 8):           # out of work: queues drained, the walker runs dry
        lane.iteration("end", False, done=not lane.in_path)
    assert not lane.parked and lane.wk == 0 and not lane.in_path
    assert all(s is FREE for s in lane.slot)
    return lane


@pytest.mark.parametrize("B", [1, 2, 3, 8, 16])
@pytest.mark.parametrize("K", [0, 1, 4])
def test_ring_is_never_overrun_and_every_path_folds_once_in_order(B, K):
    for seed, (pe, pd, pdr) in enumerate([(0.6, 0.02, 0.05), (0.25, 0.05, 0.0), (0.05, 0.01, 0.3), (0.9, 0.0, 0.0),
                                          (0.1, 0.0, 0.02), (0.02, 0.3, 0.5)]):
        lane = run(B, K, 1000 * B + 10 * K + seed, pe, pd, pdr)
        assert sorted(lane.finished) == list(range(lane.next_path))                    # every path's sample completes, once
        ringed = [p for p in lane.finished if lane.meta[p]]
        assert ringed == sorted(ringed)  # paths with records leave the ring in the order they entered it (a path that
        #                                  ended at depth 0 is its own sample and may overtake them)
        per_path = {}
        for path, level in lane.folded:
            per_path.setdefault(path, []).append(level)
        for path, levels in per_path.items():
            assert levels == list(range(lane.meta[path] - 1, -1, -1))                  # deepest first, each level once
        assert set(per_path) == {p for p, d in lane.meta.items() if d}


def test_the_guard_is_what_keeps_the_parked_ring_in_bounds():
    # long paths that escape after B scatters, queues that are never drained from outside: without the guard a parked
    # path's records and the running path's together outgrow the ring
    with pytest.raises(AssertionError, match="overrun"):
        for seed in range(200):
            run(4, 4, seed, 0.12, 0.0, 0.0, guard=False)
    for seed in range(200):
        run(4, 4, seed, 0.12, 0.0, 0.0, guard=True)
