"""`Color` helpers (reference src/color.rs:10-24).  Host-side, off the timed path."""
import numpy as np

SRGB_GAMMA = 2.2


def hex_color(x):  # color.rs:10-15
    r = ((x >> 16) & 0xFF) / 255.0
    g = ((x >> 8) & 0xFF) / 255.0
    b = (x & 0xFF) / 255.0
    return (r ** SRGB_GAMMA, g ** SRGB_GAMMA, b ** SRGB_GAMMA)


def color_bytes(color):  # color.rs:18-24 — clamp, gamma, `as u8` (truncating, NaN -> 0)
    c = np.asarray(color, dtype=np.float64)
    v = np.power(np.minimum(np.maximum(c, 0.0), 1.0), 1.0 / SRGB_GAMMA) * 255.0
    v = np.where(np.isnan(v), 0.0, v)
    return np.floor(np.clip(v, 0.0, 255.0)).astype(np.uint8)
