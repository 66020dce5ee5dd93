#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED compiled reference.

Run in the build container only (needs oracle/_ref, i.e. /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Each fixture holds the uint8 input pair and the exact arrays the reference
produced for it (captured as raw u8 / i32 through oracle/capture_image.c, not
through the lossy PPM view): both edge images, score_best, web-1, web-2,
output, and the match / score_all / scores planes of a few shifts.  The
reference's NUM_SHIFTS is compile-time 30 (src/stereo.c:6), so every fixture
is at D = 30.  Fixtures are data only; no reference source is stored.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from stereomatching_amd.synth import make_pair  # noqa: E402
from tests import oracle  # noqa: E402

# name: (w, h, kind, seed, threshold, square_width, times, lines, mode)
CASES = {
    "tor_64x48_s5":    (64, 48, "scene", 11, 0.15, 5, 32, 10, "toroidal"),
    "tor_80x60_s21":   (80, 60, "scene", 12, 0.15, 21, 32, 10, "toroidal"),
    "tor_50x37_s8":    (50, 37, "scene", 13, 0.10, 8, 5, 3, "toroidal"),
    "tor_96x40_noise": (96, 40, "noise", 14, 0.30, 9, 32, 7, "toroidal"),
    "gh_64x48_s5":     (64, 48, "scene", 15, 0.15, 5, 32, 10, "ghost"),
    "gh_80x60_s21":    (80, 60, "scene", 16, 0.15, 21, 32, 10, "ghost"),
    "gh_50x37_s8":     (50, 37, "scene", 17, 0.10, 8, 5, 3, "ghost"),
    "gh_33x70_s13":    (33, 70, "scene", 18, 0.05, 13, 32, 4, "ghost"),
    "tor_40x33_s33":   (40, 33, "scene", 19, 0.15, 33, 32, 5, "toroidal"),
}
PLANE_SHIFTS = (0, 1, 7, 29)

# the reference's own smallest test pair (test/imgs/1-240x135/{a,b}.png, what test/diff.sh
# runs on) at its default parameters (src/stereo.c:7-10): the fixture stores the DECODED
# uint8 pixels (data of the reference's tests, not source) and the compiled reference's arrays
REF_PAIR = Path("/root/reference/test/imgs/1-240x135")
REF_CASES = {
    "ref_240x135_tor": (0.15, 21, 32, 10, "toroidal"),
    "ref_240x135_gh":  (0.15, 21, 32, 10, "ghost"),
}


def decode_gray_png(path):
    from PIL import Image
    im = Image.open(path)
    assert im.mode == "L", im.mode              # 8-bit gray, as read_image requires
    return np.asarray(im, np.uint8).copy()


def main():
    if not oracle.ref_available():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` where /root/reference exists")
    out_dir = Path(__file__).resolve().parent
    jobs = []
    for name, (w, h, kind, seed, thr, sw, times, lines, mode) in CASES.items():
        left, right = make_pair(w, h, oracle.REF_NUM_SHIFTS, seed=seed, kind=kind)
        jobs.append((name, left, right, thr, sw, times, lines, mode))
    if REF_PAIR.is_dir():
        left, right = decode_gray_png(REF_PAIR / "a.png"), decode_gray_png(REF_PAIR / "b.png")
        for name, (thr, sw, times, lines, mode) in REF_CASES.items():
            jobs.append((name, left, right, thr, sw, times, lines, mode))
    for name, left, right, thr, sw, times, lines, mode in jobs:
        ref = oracle.run_reference(left, right, thr, sw, times, lines, mode)
        keep = {"left": left, "right": right,
                "params": np.array([thr, sw, times, lines, oracle.MODES[mode]], np.float64)}
        for k in ("edges-1", "edges-2", "score_best-0", "web-1", "web-2", "output-0"):
            keep[k] = ref[k]
        for d in PLANE_SHIFTS:
            for k in ("matches", "score_all", "scores"):
                keep[f"{k}-{d}"] = ref[f"{k}-{d}"]
        np.savez_compressed(out_dir / f"{name}.npz", **keep)
        print(name, {k: v.shape for k, v in keep.items() if k in ("left", "web-1")})


if __name__ == "__main__" and "--big" not in sys.argv:
    main()


# ---------------------------------------------------------------------------
# the reference's LARGE test pairs (test/time.sh:6-9 runs all of test/imgs): the decoded pixels
# of a pair are stored once, and of the arrays the compiled reference produces for it at its
# defaults only SHA-256 digests -- enough to pin the 4K tiling of the HIP path at D = 30 /
# S = 21 to the reference itself, bit for bit, without shipping gigabytes.
#     python tests/golden/make_golden.py --big       (minutes: the serial program at 4K)
# ---------------------------------------------------------------------------
BIG_PAIRS = {"ref_1920x1080": Path("/root/reference/test/imgs/4-1920x1080"),
             "ref_3840x2160": Path("/root/reference/test/imgs/5-3840x2160")}
BIG_KEYS = ("edges-1", "edges-2", "score_best-0", "web-1", "web-2", "output-0")


def main_big():
    import hashlib
    import json
    if not oracle.ref_available():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` where /root/reference exists")
    out_dir = Path(__file__).resolve().parent
    digests = {}
    for name, d in BIG_PAIRS.items():
        left, right = decode_gray_png(d / "a.png"), decode_gray_png(d / "b.png")
        np.savez_compressed(out_dir / f"{name}_pair.npz", left=left, right=right)
        for mode in ("toroidal", "ghost"):
            ref = oracle.run_reference(left, right, 0.15, 21, 32, 10, mode, keep=lambda k: k in BIG_KEYS)
            digests[f"{name}:{mode}"] = {
                "params": {"threshold": 0.15, "square_width": 21, "times": 32, "lines": 10, "num_shifts": 30},
                "sha256": {k: hashlib.sha256(np.ascontiguousarray(ref[k]).tobytes()).hexdigest() for k in BIG_KEYS},
                "dtype": {k: str(ref[k].dtype) for k in BIG_KEYS},
                "shape": list(ref["web-1"].shape)}
            print(name, mode, digests[f"{name}:{mode}"]["sha256"]["web-1"][:16], flush=True)
            (out_dir / "ref_big_digests.json").write_text(json.dumps(digests, indent=1) + "\n")


if __name__ == "__main__" and "--big" in sys.argv:
    main_big()
