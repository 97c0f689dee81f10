"""Seeded synthetic tracking masks (two people drifting across 49 frames of 480x720, overlapping in the middle) shared
by the golden generator and the tests: the masks themselves are regenerated from the seed, only the reference's output
for them is committed."""
import numpy as np


def synthetic_masks(seed, frames=49, height=480, width=720):
    """-> uint8 [2, frames, height, width], values 0 / 255 like the PNG frames the reference reads."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width]
    out = np.zeros((2, frames, height, width), np.uint8)
    for i in range(2):
        cx0, cy0 = rng.uniform(0.2, 0.8) * width, rng.uniform(0.3, 0.7) * height
        vx, vy = rng.uniform(-3, 3), rng.uniform(-1.5, 1.5)
        ax, ay = rng.uniform(0.08, 0.22) * width, rng.uniform(0.15, 0.35) * height
        for t in range(frames):
            cx, cy = cx0 + vx * t, cy0 + vy * t
            wob = 1.0 + 0.1 * np.sin(0.4 * t + i)
            m = ((xx - cx) / (ax * wob)) ** 2 + ((yy - cy) / ay) ** 2 <= 1.0
            if t % 11 == 5 + i:                       # a dropped detection: one empty frame
                m[:] = False
            out[i, t] = m * 255
    # speckle so that sub-cell structure (the 16x16 pixels under one token) matters
    noise = rng.random(out.shape) < 0.02
    out[noise] = 255 - out[noise]
    return out
