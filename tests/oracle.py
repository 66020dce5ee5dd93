"""ctypes view of oracle/liboracle.so (the CPU restatement) and of the compiled
reference under oracle/_ref.  Checker only: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product
package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
REF_DIR = ORACLE_DIR / "_ref"
MODES = {"toroidal": 0, "ghost": 1}

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def _build() -> Path:
    so = ORACLE_DIR / "liboracle.so"
    src = ORACLE_DIR / "stereo_oracle.c"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(ORACLE_DIR), "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(str(_build()))
        L.smo_find_all_edges.argtypes = [_u8p, C.c_int, C.c_int, C.c_double, C.c_int, _u8p]
        L.smo_edge_decision.argtypes = [C.c_int, C.c_int, C.c_double]
        L.smo_edge_decision.restype = C.c_int
        L.smo_edge_table.argtypes = [C.c_double, _u8p]
        L.smo_match_plane.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p]
        for f in (L.smo_addup_faithful, L.smo_addup_fast):
            f.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p]
        L.smo_record_score.argtypes = [_u8p, _i32p, C.c_int, C.c_int, _i32p]
        L.smo_hot_path.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, _i32p, _i32p]
        L.smo_cost_hot_path.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, _i32p, _i32p]
        L.smo_fill_web_holes.argtypes = [_i32p, C.c_int, C.c_int, C.c_int]
        L.smo_draw_contour_map.argtypes = [_i32p, C.c_int, C.c_int, C.c_int, _u8p]
        L.smo_draw_contour_map.restype = C.c_int
        L.smo_time.restype = C.c_double
        _lib = L
    return _lib


def _mode(m):
    return MODES[m] if isinstance(m, str) else int(m)


def find_all_edges(gray, threshold=0.15, mode="toroidal"):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    out = np.zeros((h, w), np.uint8)
    lib().smo_find_all_edges(gray, w, h, threshold, _mode(mode), out)
    return out


def edge_decision(sl, sr, threshold):
    return lib().smo_edge_decision(int(sl), int(sr), float(threshold))


def edge_table(threshold):
    out = np.zeros((766, 766), np.uint8)
    lib().smo_edge_table(float(threshold), out)
    return out


def match_plane(le, re, shift, mode="toroidal"):
    h, w = le.shape
    out = np.zeros((h, w), np.uint8)
    lib().smo_match_plane(np.ascontiguousarray(le), np.ascontiguousarray(re), w, h,
                          shift, _mode(mode), out)
    return out


def addup(match, square_width, mode="toroidal", faithful=False):
    h, w = match.shape
    out = np.zeros((h, w), np.int32)
    f = lib().smo_addup_faithful if faithful else lib().smo_addup_fast
    f(np.ascontiguousarray(match), w, h, square_width, _mode(mode), out)
    return out


def record_score(match, total):
    h, w = match.shape
    out = np.zeros((h, w), np.int32)
    lib().smo_record_score(np.ascontiguousarray(match), np.ascontiguousarray(total), w, h, out)
    return out


def hot_path(le, re, num_shifts, square_width, mode="toroidal", faithful=False):
    """-> (best, web) int32 arrays"""
    le = np.ascontiguousarray(le, np.uint8)
    re = np.ascontiguousarray(re, np.uint8)
    h, w = le.shape
    best = np.zeros((h, w), np.int32)
    web = np.zeros((h, w), np.int32)
    lib().smo_hot_path(le, re, w, h, num_shifts, square_width, _mode(mode),
                       int(faithful), best, web)
    return best, web


COSTS = {"sad": 1, "ssd": 2}


def cost_hot_path(left, right, num_shifts, square_width, mode="toroidal", cost="sad"):
    """SAD/SSD mode (the build's own definition; no reference exists) -> (best, web)"""
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    h, w = left.shape
    best = np.zeros((h, w), np.int32)
    web = np.zeros((h, w), np.int32)
    lib().smo_cost_hot_path(left, right, w, h, num_shifts, square_width, _mode(mode),
                            COSTS[cost], best, web)
    return best, web


def fill_web_holes(web, times):
    web = np.ascontiguousarray(web, np.int32).copy()
    h, w = web.shape
    lib().smo_fill_web_holes(web, w, h, times)
    return web


def draw_contour_map(web, lines):
    web = np.ascontiguousarray(web, np.int32)
    h, w = web.shape
    out = np.zeros((h, w), np.uint8)
    rc = lib().smo_draw_contour_map(web, w, h, lines, out)
    if rc:
        raise ZeroDivisionError("contour interval is zero (the reference traps here)")
    return out


def pipeline(left, right, threshold=0.15, num_shifts=30, square_width=21, times=32,
             lines=10, mode="toroidal", faithful=False, step3=True):
    """All stage outputs of the restatement as a dict of arrays."""
    el = find_all_edges(left, threshold, mode)
    er = find_all_edges(right, threshold, mode)
    best, web1 = hot_path(el, er, num_shifts, square_width, mode, faithful)
    res = {"edges-1": el, "edges-2": er, "score_best-0": best, "web-1": web1}
    if step3:
        web2 = fill_web_holes(web1, times)
        res["web-2"] = web2
        res["output-0"] = draw_contour_map(web2, lines)
    return res


# ---------------------------------------------------------------------------
# the compiled reference (only where oracle/_ref was built, i.e. where
# /root/reference exists at build time)
# ---------------------------------------------------------------------------

REF_NUM_SHIFTS = 30  # /root/reference/src/stereo.c:6, compile-time


def ref_available() -> bool:
    return (REF_DIR / "stereomatch-capture").exists()


def _read_raw(path):
    data = Path(path).read_bytes()
    w, h, elem = np.frombuffer(data, np.int32, 3)
    dt = {1: np.uint8, 4: np.int32, 8: np.float64}[int(elem)]
    return np.frombuffer(data, dt, int(w) * int(h), 12).reshape(int(h), int(w)).copy()


def run_reference(left, right, threshold=0.15, square_width=21, times=32, lines=10,
                  mode="toroidal", keep=None, allow_sigfpe=False):
    """Run the UNMODIFIED compiled reference on a uint8 pair and return every
    array it dumps, keyed like its file names ("edges-1", "matches-7",
    "score_all-7", "scores-7", "score_best-0", "web-1", "web-2", "output-0")."""
    from stereomatching_amd.synth import write_pgm

    exe = REF_DIR / ("stereomatch-capture" if mode == "toroidal" else "stereomatch-ghost-capture")
    with tempfile.TemporaryDirectory() as td:
        write_pgm(f"{td}/a.pgm", left)
        write_pgm(f"{td}/b.pgm", right)
        os.mkdir(f"{td}/out")
        env = dict(os.environ, SMO_CAPTURE_DIR=f"{td}/out")
        p = subprocess.run([str(exe), f"{td}/a.pgm", f"{td}/b.pgm", repr(float(threshold)),
                            str(square_width), str(times), str(lines)],
                           env=env, capture_output=True, text=True)
        if p.returncode != 0 and not (allow_sigfpe and p.returncode == -8):
            raise RuntimeError(f"reference exited {p.returncode}: {p.stderr}")
        out = {}
        for f in sorted(Path(f"{td}/out").glob("*.raw")):
            if keep is None or keep(f.stem):
                out[f.stem] = _read_raw(f)
        out["stdout"] = p.stdout
        out["returncode"] = p.returncode
    return out


# ---------------------------------------------------------------------------
# full-size images: the same restatement, run on row bands in threads
# ---------------------------------------------------------------------------

def _bands(h, n_bands):
    edges = np.linspace(0, h, min(n_bands, h) + 1).astype(int)
    return [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def _run_threads(jobs, threads):
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=threads) as ex:     # ctypes releases the GIL
        return list(ex.map(lambda j: j(), jobs))


def hot_path_banded(le, re, num_shifts, square_width, mode="toroidal", n_bands=64, threads=None):
    """hot_path() of a large image, bit for bit, computed band by band: the result at a
    pixel depends only on rows within the window, so each band of rows is taken from an
    oracle run on the band plus a `half`-row halo (toroidal: the halo wraps around the
    image; ghost: the image's first / last rows are real borders of the crop as well)."""
    h, w = le.shape
    half = square_width // 2
    threads = threads or min(32, os.cpu_count() or 1)
    best = np.zeros((h, w), np.int32)
    web = np.zeros((h, w), np.int32)

    def job(y0, y1):
        def run():
            if mode == "toroidal":
                rows = np.arange(y0 - half, y1 + half) % h
                lo = half
            else:
                a, b = max(0, y0 - half), min(h, y1 + half)
                rows = np.arange(a, b)
                lo = y0 - a
            if len(rows) < 2 * half + 1:        # the oracle needs square_width <= height
                raise ValueError("band shorter than the window")
            ob, ow = hot_path(le[rows], re[rows], num_shifts, square_width, mode)
            best[y0:y1] = ob[lo:lo + y1 - y0]
            web[y0:y1] = ow[lo:lo + y1 - y0]
        return run
    _run_threads([job(a, b) for a, b in _bands(h, n_bands)], threads)
    return best, web


def cost_hot_path_banded(left, right, num_shifts, square_width, mode="toroidal", cost="sad", n_bands=64,
                         threads=None):
    """cost_hot_path() of a large image, band by band (as hot_path_banded)"""
    h, w = left.shape
    half = square_width // 2
    threads = threads or min(32, os.cpu_count() or 1)
    best = np.zeros((h, w), np.int32)
    web = np.zeros((h, w), np.int32)

    def job(y0, y1):
        def run():
            if mode == "toroidal":
                rows = np.arange(y0 - half, y1 + half) % h
                lo = half
            else:
                a, b = max(0, y0 - half), min(h, y1 + half)
                rows = np.arange(a, b)
                lo = y0 - a
            ob, ow = cost_hot_path(left[rows], right[rows], num_shifts, square_width, mode, cost)
            best[y0:y1] = ob[lo:lo + y1 - y0]
            web[y0:y1] = ow[lo:lo + y1 - y0]
        return run
    _run_threads([job(a, b) for a, b in _bands(h, n_bands)], threads)
    return best, web


def find_all_edges_banded(gray, threshold=0.15, mode="toroidal", n_bands=32, threads=None):
    h, w = gray.shape
    threads = threads or min(32, os.cpu_count() or 1)
    out = np.zeros((h, w), np.uint8)

    def job(y0, y1):
        def run():
            if mode == "toroidal":
                rows = np.arange(y0 - 1, y1 + 1) % h
                lo = 1
            else:
                a, b = max(0, y0 - 1), min(h, y1 + 1)
                rows = np.arange(a, b)
                lo = y0 - a
            out[y0:y1] = find_all_edges(gray[rows], threshold, mode)[lo:lo + y1 - y0]
        return run
    _run_threads([job(a, b) for a, b in _bands(h, n_bands)], threads)
    return out
