"""The one reference test next to the hot path — `colors_work` (reference src/color.rs:31-38) —
reproduced against the oracle and against the host mirror; plus Buffer semantics
(reference src/buffer.rs:32-93)."""
import numpy as np

import rpt_amd
from rpt_amd import Buffer, Filter, color_bytes, hex_color


def test_colors_work_oracle(oracle):  # color.rs:31-38, the reference's own assertions
    black, white, red = oracle.hex_color(0x000000), oracle.hex_color(0xFFFFFF), oracle.hex_color(0xFF0000)
    assert oracle.color_bytes(black) == [0, 0, 0]
    assert oracle.color_bytes(white) == [255, 255, 255]
    assert oracle.color_bytes(red) == [255, 0, 0]


def test_colors_work_host_mirror():
    assert list(color_bytes(hex_color(0x000000))) == [0, 0, 0]
    assert list(color_bytes(hex_color(0xFFFFFF))) == [255, 255, 255]
    assert list(color_bytes(hex_color(0xFF0000))) == [255, 0, 0]


def test_hex_color_and_color_bytes_agree_with_oracle(oracle):
    for x in (0xAAAAAA, 0xBC0000, 0x00BC00, 0xFFFEFA, 0xB7CA79, 0x264653, 0x6F5D48, 0x123456):
        assert tuple(oracle.hex_color(x)) == hex_color(x)
    rs = np.random.RandomState(1)
    cols = np.concatenate([rs.rand(500, 3) * 1.4 - 0.2, [[np.nan, 2.0, -1.0]], [[1.0, 0.0, 0.5]]])
    ours = color_bytes(cols)
    for c, o in zip(cols, ours):
        assert oracle.color_bytes(c) == list(o)


def test_truncation_not_rounding(oracle):
    # `as u8` truncates (color.rs:20-22): 0.5^(1/2.2)*255 = 186.08 -> 186; a value just below 1 -> 254
    assert oracle.color_bytes([0.5, 0.999, 1.0]) == [186, 254, 255]


def test_buffer_image_and_variance_match_oracle(oracle):
    rs = np.random.RandomState(3)
    w, h = 13, 7
    batches = [rs.rand(w * h, 3) * 1.2 for _ in range(3)]
    for radius in (0, 1, 2):
        b = Buffer(w, h, Filter.Box(radius))
        for s in batches:
            b.add_samples(s)
        assert (b.image() == oracle.buffer_image(w, h, radius, batches)).all()
    b = Buffer(w, h)
    for s in batches:
        b.add_samples(s)
    assert abs(b.variance() - oracle.buffer_variance(w, h, batches)) < 1e-15


def test_buffer_batches_are_weighted_equally():
    # each add_samples call contributes ONE batch mean per pixel regardless of its spp (buffer.rs:32-40,75-93)
    b = Buffer(2, 1)
    b.add_samples([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]])
    b.add_samples([[1.0, 1.0, 1.0], [1.0, 1.0, 1.0]])
    img = b.image()
    assert list(img[0, 0]) == list(color_bytes([0.5, 0.5, 0.5])) and list(img[0, 1]) == [255, 255, 255]
    try:
        b.add_samples([[0.0, 0.0, 0.0]])
        assert False
    except AssertionError:
        pass  # "Invalid sample dimension" (buffer.rs:33-36)
