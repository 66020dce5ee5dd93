"""Build the native pieces in-tree.

    python -m stereomatching_amd.build            # HIP library (+ oracle checker)

libstereo_hip.so is built with hipcc for gfx950 only (cross-compiles without a
GPU) and lands next to this file so that it travels with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "libstereo_hip.so"
# the bit-sliced kernel's builds are spread over four translation units so that they
# compile side by side (one unit: ~2 min; four beside the rest: ~50 s on 8 cores)
SOURCES = ["sm_match_bs_ds8.hip", "sm_match_bs.hip", "sm_match_bs_duo8.hip", "sm_match_bs_duo.hip", "sm_match_bs_ds4.hip",
           "sm_api.hip", "sm_match.hip", "sm_cost.hip", "sm_cost_qs.hip", "sm_cost_pc.hip", "sm_cost_ssd.hip", "sm_cost_mfma.hip", "sm_cost_strip.hip", "sm_gather.hip"]
HEADERS = [CSRC / "sm_internal.h", CSRC / "sm_match_bs_kernel.h", CSRC / "sm_cost.h", ROOT / "include" / "stereo_hip.h"]
OBJDIR = PKG / "obj"
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # the edge test must round exactly like the reference's C doubles
    "-ffp-contract=off",
    # no SLP packing: the vectoriser turns pairs of f32 adds of the edge kernel into
    # v_pk_add_f32, which keeps co-resident waves from issuing side by side;
    # measured 18.6 -> 16.6 us for the two 4K images
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-result",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin)")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _jobs() -> int:
    env = os.environ.get("SM_BUILD_JOBS")
    return max(1, int(env)) if env else max(1, min(len(SOURCES), os.cpu_count() or 1))


def _compile_and_link(out: Path, objdir: Path, flags=(), verbose: bool = False, jobs: int | None = None) -> None:
    """hipcc -c every source (side by side, `jobs` at a time) into objdir, then link `out`."""
    import concurrent.futures as cf
    objdir.mkdir(parents=True, exist_ok=True)
    common = [*HIPCC_FLAGS, *flags, f"-I{ROOT / 'include'}", f"-I{CSRC}"]

    # objects of another flag set are not reused.  The stamp holds the flags WITHOUT the include
    # paths (they depend on where the tree is checked out; the Makefile spells them relatively and
    # shares obj/product with this function, writing the same stamp), and is never tracked.
    stamp = objdir / "flags.txt"
    text = " ".join([*HIPCC_FLAGS, *flags])
    if not stamp.exists() or stamp.read_text().strip() != text:
        for old in objdir.glob("*.o"):
            old.unlink()
        stamp.write_text(text + "\n")

    def one(src: str) -> Path:
        obj = objdir / (Path(src).stem + ".o")
        if not _stale(obj, [CSRC / src, *HEADERS]):
            return obj
        cmd = [_hipcc(), *common, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj
    with cf.ThreadPoolExecutor(max_workers=jobs or _jobs()) as ex:
        objs = list(ex.map(one, SOURCES))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(out)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    deps = [CSRC / s for s in SOURCES] + HEADERS
    if force or _stale(LIB, deps):
        _compile_and_link(LIB, OBJDIR / "product", verbose=verbose)
    return LIB


def build_diag(verbose: bool = False, name: str = "stamps", flags=()) -> Path:
    """tools/diag/libstereo_hip_<name>.so: the same sources with -DSM_STAMPS (per-wave
    time stamps); a diagnostic build, loaded only by tools/wave_timeline.py."""
    out = ROOT / "tools" / "diag" / f"libstereo_hip_{name}.so"
    out.parent.mkdir(parents=True, exist_ok=True)
    deps = [CSRC / s for s in SOURCES] + HEADERS
    if _stale(out, deps):
        _compile_and_link(out, OBJDIR / f"diag_{name}", ["-DSM_STAMPS", *flags], verbose)
    return out


def build_variants(variants: dict, verbose: bool = False, jobs: int = 4) -> None:
    """Same-device A/B builds for tools/ab_variants.py: name -> extra hipcc flags, each
    to stereomatching_amd/variants/<name>.so (git-ignored, travels with gpurun)."""
    import concurrent.futures as cf
    vdir = PKG / "variants"
    vdir.mkdir(exist_ok=True)
    for old in vdir.glob("*.so"):
        if old.stem not in variants:
            old.unlink()

    def one(item):
        name, flags = item
        _compile_and_link(vdir / f"{name}.so", OBJDIR / f"variant_{name}", flags, verbose,
                          jobs=max(1, _jobs() // max(1, min(jobs, len(variants)))))
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(one, variants.items()))


def build_oracle(verbose: bool = False) -> None:
    """The CPU checker (oracle/liboracle.so) and, where /root/reference exists,
    the compiled reference under oracle/_ref.  Building the checker is not
    using it: nothing in this package loads either."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "liboracle.so"], stdout=out)
    if Path("/root/reference/src").is_dir():
        subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "ref"], stdout=out)


def build_host(verbose: bool = False) -> None:
    """The C command-line programs (stereopar, stereopar-ghost, ...)."""
    out = None if verbose else subprocess.DEVNULL
    if (ROOT / "Makefile").exists():
        # (shares stereomatching_amd/obj/product with build_hip: the library is not compiled twice)
        subprocess.check_call(["make", "-C", str(ROOT), f"-j{_jobs()}", "build=timing"], stdout=out)


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv, verbose=True)
    build_oracle(verbose=True)
    print("built", LIB)
