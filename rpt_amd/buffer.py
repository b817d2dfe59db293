"""`Buffer` / `Filter` (reference src/buffer.rs): the consumer of the hot path's output.
Host-side and off the timed path, exactly as in the reference; vectorised with numpy."""
import numpy as np

from .color import color_bytes


class Filter:
    """`Filter::Box(radius)` (buffer.rs:98-108); default radius 0 is a no-op."""

    def __init__(self, radius=0):
        self.radius = int(radius)

    @staticmethod
    def Box(radius):
        return Filter(radius)


class Buffer:
    def __init__(self, width, height, filter=None):  # Buffer::new, buffer.rs:15-22
        self.width, self.height = int(width), int(height)
        self.filter = filter or Filter()
        self.samples = []  # one (H*W, 3) array per add_samples call (= per-pixel Vec<Color>)

    def add_samples(self, samples):  # buffer.rs:32-40
        s = np.asarray(samples, dtype=np.float64).reshape(-1, 3)
        assert len(s) == self.width * self.height, "Invalid sample dimension"
        self.samples.append(s.copy())

    def _filtered(self):  # get_filtered_color, buffer.rs:75-93
        assert self.samples, "Pixel found with no samples"
        w, h, r = self.width, self.height, self.filter.radius
        total = np.zeros((h * w, 3))
        for s in self.samples:  # iter().sum::<Color>() per pixel: batches added in order
            total = total + s
        total = total.reshape(h, w, 3)
        if r == 0:
            return total / float(len(self.samples))
        acc = np.zeros((h, w, 3))
        cnt = np.zeros((h, w, 1))
        # the reference loops i (x) outer, j (y) inner: same order here so sums round alike
        for dx in range(-r, r + 1):
            for dy in range(-r, r + 1):
                ys0, ys1 = max(0, -dy), min(h, h - dy)
                xs0, xs1 = max(0, -dx), min(w, w - dx)
                acc[ys0:ys1, xs0:xs1] += total[ys0 + dy:ys1 + dy, xs0 + dx:xs1 + dx]
                cnt[ys0:ys1, xs0:xs1] += len(self.samples)
        return acc / cnt

    def image(self):  # buffer.rs:43-56 -> (H, W, 3) uint8
        return color_bytes(self._filtered())

    def variance(self):  # buffer.rs:59-73
        n = len(self.samples)
        stack = np.stack(self.samples)  # (n, HW, 3)
        total = np.zeros_like(stack[0])
        for s in self.samples:
            total = total + s
        mean = total / float(n)
        sq = np.zeros(stack.shape[1])
        for s in self.samples:
            d = s - mean
            sq = sq + ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        with np.errstate(all="ignore"):  # one batch: 0/0 = NaN, as the reference computes
            per_pixel = sq / (float(n) - 1.0)
        variance = 0.0
        for v in per_pixel.tolist():
            variance += v
        return variance / float(len(per_pixel))
