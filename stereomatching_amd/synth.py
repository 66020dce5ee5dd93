"""Seeded synthetic stereo pairs (uint8) and PGM I/O.

The reference ships one natural scene at five sizes (/root/reference
test/imgs); it does not travel to the GPU box, so tests and bench.py use
these generated pairs instead: a smooth textured left
image and a right image that is the left one displaced by a piecewise-constant
disparity field plus a little independent noise.
"""
from __future__ import annotations

import numpy as np

# the BASELINE.json configurations: name -> (W, H, D, S, border mode)
CONFIGS = {
    "C1": (384, 288, 16, 5, "toroidal"),
    "C2": (1920, 1080, 64, 7, "toroidal"),
    "C3": (3840, 2160, 128, 9, "toroidal"),
    "C4": (1920, 1080, 64, 7, "toroidal"),  # x64 pairs, sharded
    "C5": (3840, 2160, 256, 11, "ghost"),
    # the reference's own defaults (NUM_SHIFTS 30, square_width 21) at its 4K test size
    "REF4K": (3840, 2160, 30, 21, "toroidal"),
}


def make_pair(w: int, h: int, num_shifts: int, seed: int = 0, kind: str = "scene"):
    """Return (left, right) uint8 arrays of shape (h, w).

    kind: "scene" (textured, realistic edge density), "noise" (white noise:
    nearly every pixel is an edge), "constant" (all scores tie: exercises the
    last-shift-wins rule), "zeros".
    """
    rng = np.random.default_rng(0x5EED0000 + seed)
    if kind == "constant":
        a = np.full((h, w), 97, np.uint8)
        return a, a.copy()
    if kind == "zeros":
        a = np.zeros((h, w), np.uint8)
        return a, a.copy()
    if kind == "noise":
        return (rng.integers(0, 256, (h, w), dtype=np.uint8),
                rng.integers(0, 256, (h, w), dtype=np.uint8))
    if kind != "scene":
        raise ValueError(kind)

    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w), np.float32)
    for _ in range(6):
        fx, fy = rng.uniform(0.005, 0.08, 2)
        ph = rng.uniform(0, 2 * np.pi)
        img += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph)
    # blocky texture gives step edges
    bs = max(4, min(w, h) // 24)
    blocks = rng.uniform(-1.5, 1.5, (h // bs + 2, w // bs + 2)).astype(np.float32)
    img += np.kron(blocks, np.ones((bs, bs), np.float32))[:h, :w]
    img += rng.normal(0, 0.02, (h, w)).astype(np.float32)
    img = (img - img.min()) / (img.max() - img.min() + 1e-9)
    # keep away from black: the edge test is relative to local brightness
    left = np.clip(48.0 + img * 200.0, 0, 255).astype(np.uint8)

    # piecewise-constant disparity field in [0, D)
    ds = max(8, min(w, h) // 6)
    dfield = rng.integers(0, max(1, num_shifts), (h // ds + 2, w // ds + 2))
    disp = np.kron(dfield, np.ones((ds, ds), np.int64))[:h, :w]
    # the reference compares left(x) with right(x + d): right(x + d) = left(x)
    src_x = (np.arange(w)[None, :] - disp) % w
    right = left[np.arange(h)[:, None], src_x].astype(np.int16)
    right += rng.integers(-1, 2, (h, w), dtype=np.int16)
    right = np.clip(right, 0, 255).astype(np.uint8)
    return left, right


def write_pgm(path, img: np.ndarray) -> None:
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h))
        f.write(img.tobytes())


def read_pgm(path) -> np.ndarray:
    with open(path, "rb") as f:
        data = f.read()
    assert data[:2] == b"P5", "binary PGM expected"
    tokens, pos = [], 2
    while len(tokens) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tokens.append(int(data[pos:end]))
        pos = end
    pos += 1
    w, h, maxval = tokens
    assert maxval == 255
    return np.frombuffer(data, np.uint8, w * h, pos).reshape(h, w).copy()
