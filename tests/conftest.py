import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN_DIR = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # a fresh checkout has no binaries (they are git-ignored): build the HIP library
    # in-tree before any test imports it.  hipcc cross-compiles without a GPU.
    from stereomatching_amd import build
    if not build.LIB.exists():
        build.build_hip(verbose=True)


def golden_cases():
    return sorted(p.stem for p in GOLDEN_DIR.glob("*.npz") if not p.stem.endswith("_pair"))


def big_reference_cases():
    """(pair name, mode) of tests/golden/ref_big_digests.json: the reference's own large test pairs
    (pixels in <name>_pair.npz) with SHA-256 digests of what the compiled reference made of them"""
    import json
    f = GOLDEN_DIR / "ref_big_digests.json"
    if not f.exists():
        return []
    return sorted(tuple(k.split(":")) for k in json.loads(f.read_text()))


def load_big_reference(name, mode):
    import json
    z = np.load(GOLDEN_DIR / f"{name}_pair.npz")
    d = json.loads((GOLDEN_DIR / "ref_big_digests.json").read_text())[f"{name}:{mode}"]
    return z["left"], z["right"], d


def sha256_of(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_golden(name):
    z = np.load(GOLDEN_DIR / f"{name}.npz")
    thr, sw, times, lines, mode = z["params"]
    params = dict(threshold=float(thr), square_width=int(sw), times=int(times),
                  lines=int(lines), mode="ghost" if int(mode) else "toroidal")
    return z, params


@pytest.fixture(scope="session")
def hip():
    """The HIP layer on cuda:0.  Fails (not skips) if the extension is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    from stereomatching_amd import pipeline
    return pipeline
