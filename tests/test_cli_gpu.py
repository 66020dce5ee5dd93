"""GPU test of the process boundary: the reference's own acceptance test
(test/diff.sh) restated -- every PPM the GPU programs write in a debug build
equals, byte for byte, what the serial programs write."""
import os
import subprocess
from pathlib import Path

import pytest

from stereomatching_amd.synth import make_pair, write_pgm
from tests.test_cli_cpu import STDOUT_RE, programs, run  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ser,par,sdir,pdir", [("stereomatch", "stereopar", "ser", "par"),
                                               ("stereomatch-ghost", "stereopar-ghost", "sergh", "pargh")])
@pytest.mark.parametrize("w,h,args,shifts", [
    (120, 67, (0.15, 21, 32, 10), None),          # the reference's defaults, 96 files
    (97, 64, (0.1, 8, 3, 4), "40"),               # STEREO_NUM_SHIFTS extension, 126 files
])
def test_diff_sh_equivalent(programs, tmp_path, ser, par, sdir, pdir, w, h, args, shifts):  # noqa: F811
    left, right = make_pair(w, h, 30, seed=w)
    write_pgm(tmp_path / "a.pgm", left)
    write_pgm(tmp_path / "b.pgm", right)
    os.mkdir(tmp_path / sdir)
    os.mkdir(tmp_path / pdir)
    env = dict(os.environ)
    if shifts:
        env["STEREO_NUM_SHIFTS"] = shifts
    outs = []
    for exe in (programs["debug"][ser], programs["debug"][par]):
        p = subprocess.run([str(exe), "a.pgm", "b.pgm", *map(str, args)], cwd=tmp_path,
                           capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stderr
        assert STDOUT_RE.match(p.stdout), p.stdout
        outs.append(p.stdout)
    names = sorted(f.name for f in (tmp_path / sdir).iterdir())
    assert len(names) == 6 + 3 * int(shifts or 30)
    assert names == sorted(f.name for f in (tmp_path / pdir).iterdir())
    for name in names:
        assert (tmp_path / sdir / name).read_bytes() == (tmp_path / pdir / name).read_bytes(), name


def test_timing_build_runs_and_writes_nothing(programs, tmp_path):  # noqa: F811
    left, right = make_pair(640, 360, 30, seed=3)
    write_pgm(tmp_path / "a.pgm", left)
    write_pgm(tmp_path / "b.pgm", right)
    for prog in ("stereopar", "stereopar-ghost"):
        p = run(programs["timing"][prog], "a.pgm", "b.pgm", cwd=tmp_path)
        assert p.returncode == 0, p.stderr
        m = STDOUT_RE.match(p.stdout)
        assert m and (m.group(1), m.group(2)) == ("640", "360")
        assert p.stdout.split()[14] == m.group(3)          # awk '{print $15}' of test/time.sh
    assert not list(tmp_path.glob("*.ppm"))


def test_zero_contour_interval_is_reported(programs, tmp_path):  # noqa: F811
    # a constant pair gives a constant web: the reference divides by a zero interval
    left, right = make_pair(64, 48, 30, kind="constant")
    write_pgm(tmp_path / "a.pgm", left)
    write_pgm(tmp_path / "b.pgm", right)
    p = run(programs["timing"]["stereopar"], "a.pgm", "b.pgm", 0.15, 5, cwd=tmp_path)
    assert p.returncode == 1 and "contour interval is zero" in p.stderr


@pytest.mark.parametrize("ghost", [False, True])
def test_stereopar_batch_matches_the_oracle(programs, tmp_path, ghost):  # noqa: F811
    """stereopar-batch (C host over the C ABI, all visible devices, pinned async transfers,
    narrow maps): every pair's web map equals the oracle's, in list order."""
    import numpy as np

    from stereomatching_amd.synth import read_pgm
    from tests import oracle
    w, h, d, sw, n_pairs = 200, 120, 64, 7, 7
    mode = "ghost" if ghost else "toroidal"
    lines = []
    pairs = []
    for j in range(n_pairs):
        left, right = make_pair(w, h, d, seed=100 + j)
        write_pgm(tmp_path / f"l{j}.pgm", left)
        write_pgm(tmp_path / f"r{j}.pgm", right)
        lines.append(f"l{j}.pgm r{j}.pgm")
        pairs.append((left, right))
    (tmp_path / "list.txt").write_text("# pairs\n" + "\n".join(lines) + "\n")
    os.mkdir(tmp_path / "out")
    exe = programs["timing"]["stereopar"].parent / "stereopar-batch"
    args = [str(exe), "-n", str(d), "-b", "3", "-o", "out", "-r", "2"] + (["-g"] if ghost else []) + \
           ["list.txt", "0.15", str(sw)]
    p = subprocess.run(args, cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    fields = dict(f.split(" = ") for f in p.stdout.strip().split(", "))
    assert int(fields["pairs"]) == 2 * n_pairs and int(fields["shifts"]) == d
    total = 0
    for j, (left, right) in enumerate(pairs):
        el = oracle.find_all_edges(left, 0.15, mode)
        er = oracle.find_all_edges(right, 0.15, mode)
        _, web = oracle.hot_path(el, er, d, sw, mode)
        got = read_pgm_any(tmp_path / "out" / f"web-{j}.pgm")
        assert np.array_equal(got, web), j
        total += int(web.sum())
    assert int(fields["checksum"]) == 2 * total


def test_stereopar_batch_two_workers_on_one_device(programs, tmp_path):  # noqa: F811
    """`-d 0,0`: the multi-device split of stereopar-batch (pair j -> worker j mod n, a host thread
    pair, a plan and three streams per worker) with BOTH workers on device 0 -- what a one-GPU box
    can run of the n_devices > 1 path: every map equals the oracle's, in list order."""
    import numpy as np
    from tests import oracle
    w, h, d, sw, n_pairs = 160, 96, 30, 9, 9
    lines, pairs = [], []
    for j in range(n_pairs):
        left, right = make_pair(w, h, d, seed=300 + j)
        write_pgm(tmp_path / f"l{j}.pgm", left)
        write_pgm(tmp_path / f"r{j}.pgm", right)
        lines.append(f"l{j}.pgm r{j}.pgm")
        pairs.append((left, right))
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    os.mkdir(tmp_path / "out")
    exe = programs["timing"]["stereopar"].parent / "stereopar-batch"
    p = subprocess.run([str(exe), "-d", "0,0", "-n", str(d), "-b", "2", "-o", "out", "list.txt", "0.15", str(sw)],
                       cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    fields = dict(f.split(" = ") for f in p.stdout.strip().split(", "))
    assert int(fields["pairs"]) == n_pairs and int(fields["devices"]) == 2
    total = 0
    for j, (left, right) in enumerate(pairs):
        el = oracle.find_all_edges(left, 0.15, "toroidal")
        er = oracle.find_all_edges(right, 0.15, "toroidal")
        _, web = oracle.hot_path(el, er, d, sw, "toroidal")
        assert np.array_equal(read_pgm_any(tmp_path / "out" / f"web-{j}.pgm"), web), j
        total += int(web.sum())
    assert int(fields["checksum"]) == total


def test_stereopar_batch_failure_path_ends_cleanly(programs, tmp_path):  # noqa: F811
    """-x N injects a failure when a worker is about to submit its N-th batch, with batches in
    flight: both host threads of every worker must stop, the message goes to stderr and the exit
    code is 1 (the reference's convention) -- no hang, no result line."""
    w, h, d = 96, 64, 16
    lines = []
    for j in range(6):
        left, right = make_pair(w, h, d, seed=400 + j)
        write_pgm(tmp_path / f"l{j}.pgm", left)
        write_pgm(tmp_path / f"r{j}.pgm", right)
        lines.append(f"l{j}.pgm r{j}.pgm")
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    exe = programs["timing"]["stereopar"].parent / "stereopar-batch-testhooks"     # (-DSTEREOPAR_BATCH_TEST_HOOKS)
    product = programs["timing"]["stereopar"].parent / "stereopar-batch"
    p = subprocess.run([str(product), "-x", "7", "list.txt"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and p.stderr.startswith("usage:")      # the product program has no such option
    for devices in ("0", "0,0"):
        p = subprocess.run([str(exe), "-d", devices, "-n", str(d), "-b", "1", "-r", "50", "-x", "7",
                            "list.txt", "0.15", "5"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert p.returncode == 1 and "injected failure" in p.stderr and not p.stdout.strip(), (p.stdout, p.stderr)


def read_pgm_any(path):
    """binary PGM with any maxval <= 255 (stereopar-batch writes maxval = number of shifts)"""
    import numpy as np
    data = Path(path).read_bytes()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P5"
    w, h = map(int, parts[1].split())
    assert int(parts[2]) <= 255
    return np.frombuffer(parts[3], np.uint8, w * h).reshape(h, w).astype(np.int32)
