"""CPU tests of the process boundary: the image.h
implementation (PNG/PGM reader, byte-exact P3 writer, file naming) and the
command-line contract (argv, messages, exit codes, stdout line), checked
against the compiled reference where it is available."""
import os
import re
import struct
import subprocess
import zlib
from pathlib import Path

import numpy as np
import pytest

from stereomatching_amd.synth import make_pair, write_pgm
from tests import oracle

ROOT = Path(__file__).resolve().parent.parent
REF_IMGS = Path("/root/reference/test/imgs")
STDOUT_RE = re.compile(r"^width = (\d+), height = (\d+), t1 = [\d.]+, t2 = [\d.]+, elapsed = ([\d.]+)\n$")


@pytest.fixture(scope="module")
def programs():
    subprocess.check_call(["make", "-C", str(ROOT), "build=debug"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", str(ROOT), "build=timing"], stdout=subprocess.DEVNULL)
    return {b: {n: ROOT / b / n for n in ("stereomatch", "stereomatch-ghost", "stereopar",
                                          "stereopar-ghost")} for b in ("debug", "timing")}


def run(exe, *args, cwd=None):
    return subprocess.run([str(exe), *map(str, args)], cwd=cwd, capture_output=True, text=True)


def write_png(path, img, filters=None, depth=8, chunks=1):
    """Minimal PNG writer (gray) exercising every filter type and split IDATs."""
    h, w = img.shape
    raw = bytearray()
    prev = np.zeros(w * (2 if depth == 16 else 1), np.int32)
    bpp = 2 if depth == 16 else 1
    for y in range(h):
        if depth == 16:
            line = np.stack([img[y], img[y] ^ 0x5A], 1).reshape(-1).astype(np.int32)  # hi, lo bytes
        else:
            line = img[y].astype(np.int32)
        f = (filters[y % len(filters)] if filters else 0)
        a = np.concatenate([np.zeros(bpp, np.int32), line[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if f == 0: enc = line
        elif f == 1: enc = line - a
        elif f == 2: enc = line - prev
        elif f == 3: enc = line - ((a + prev) >> 1)
        else:
            p = a + prev - c
            pa, pb, pc = abs(p - a), abs(p - prev), abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            enc = line - pred
        raw.append(f)
        raw += bytes((enc & 0xFF).astype(np.uint8))
        prev = line
    z = zlib.compress(bytes(raw), 6)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 0, 0, 0, 0))
    out += chunk(b"gAMA", struct.pack(">I", 45455))
    step = max(1, len(z) // chunks)
    for i in range(0, len(z), step):
        out += chunk(b"IDAT", z[i:i + step])
    out += chunk(b"IEND", b"")
    Path(path).write_bytes(out)


def read_ppm(path):
    toks = Path(path).read_text().split()
    assert toks[0] == "P3" and toks[3] == "255"
    w, h = int(toks[1]), int(toks[2])
    v = np.array(toks[4:], np.int64).reshape(h, w, 3)
    assert (v[..., 0] == v[..., 1]).all() and (v[..., 0] == v[..., 2]).all()
    return v[..., 0]


def test_png_and_pgm_readers_agree_with_the_pixels(programs, tmp_path):
    """read_image decodes what was encoded: all five PNG filters, several IDAT
    chunks, 8- and 16-bit, and PGM; checked through the edges-1 dump."""
    left, right = make_pair(61, 43, 30, seed=5)
    os.mkdir(tmp_path / "ser")
    write_pgm(tmp_path / "b.pgm", right)
    want = None
    for name, kw in (("a8.png", dict(filters=[0, 1, 2, 3, 4], chunks=3)),
                     ("a16.png", dict(filters=[4, 3, 1], depth=16)),
                     ("a.pgm", None)):
        if kw is None:
            write_pgm(tmp_path / name, left)
        else:
            write_png(tmp_path / name, left, **kw)
        p = run(programs["debug"]["stereomatch"], name, "b.pgm", 0.15, 5, 0, 3, cwd=tmp_path)
        assert p.returncode == 0, p.stderr
        edges = read_ppm(tmp_path / "ser" / "edges-1.ppm")
        if want is None:
            want = np.where(oracle.find_all_edges(left, 0.15) == 1, 0, 255)
        assert np.array_equal(edges, want), name


@pytest.mark.skipif(not (REF_IMGS.exists() and oracle.ref_available()),
                    reason="needs /root/reference and oracle/_ref")
@pytest.mark.parametrize("variant,subdir", [("stereomatch", "ser"), ("stereomatch-ghost", "sergh")])
def test_cli_dumps_are_byte_identical_to_the_reference(programs, tmp_path, variant, subdir):
    """All 96 PPMs of the reference's debug build on its own test image, byte for byte:
    pins the PNG reader, every stage of the oracle and the P3 writer at once."""
    for side in ("a", "b"):
        (tmp_path / f"{side}.png").write_bytes((REF_IMGS / "1-240x135" / f"{side}.png").read_bytes())
    os.makedirs(tmp_path / "ref")
    os.mkdir(tmp_path / "ref" / subdir)
    os.mkdir(tmp_path / subdir)
    args = ("a.png", "b.png", 0.15, 7, 32, 10)
    r = subprocess.run([str(oracle.REF_DIR / f"{variant}-debug"), *map(str, args)],
                       cwd=tmp_path / "ref", capture_output=True, text=True,
                       env=dict(os.environ))
    # the reference resolves the images relative to its cwd
    if r.returncode != 0:
        r = subprocess.run([str(oracle.REF_DIR / f"{variant}-debug"), "../a.png", "../b.png",
                            *map(str, args[2:])], cwd=tmp_path / "ref", capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    p = run(programs["debug"][variant], *args, cwd=tmp_path)
    assert p.returncode == 0, p.stderr
    assert STDOUT_RE.match(p.stdout) and STDOUT_RE.match(r.stdout)
    ref_files = sorted(f.name for f in (tmp_path / "ref" / subdir).iterdir())
    assert len(ref_files) == 96
    for name in ref_files:
        assert (tmp_path / subdir / name).read_bytes() == (tmp_path / "ref" / subdir / name).read_bytes(), name


def test_stdout_line_and_time_sh_field(programs, tmp_path):
    left, right = make_pair(48, 40, 30, seed=2)
    write_pgm(tmp_path / "a.pgm", left)
    write_pgm(tmp_path / "b.pgm", right)
    p = run(programs["timing"]["stereomatch"], "a.pgm", "b.pgm", 0.15, 5, cwd=tmp_path)
    assert p.returncode == 0
    m = STDOUT_RE.match(p.stdout)
    assert m and m.group(1) == "48" and m.group(2) == "40"
    # test/time.sh takes awk field 15
    assert p.stdout.split()[14] == m.group(3)
    assert not list(tmp_path.glob("*.ppm")), "timing build (NO_WRITES) must not write files"


@pytest.mark.parametrize("prog", ["stereomatch", "stereomatch-ghost", "stereopar", "stereopar-ghost"])
def test_cli_errors_match_the_reference_contract(programs, tmp_path, prog):
    """usage / parse / validation errors are decided before any GPU call, so the GPU
    programs can be checked here too (SURVEY.md section 8b 'Errors')."""
    exe = programs["timing"][prog]
    left, right = make_pair(32, 24, 30, seed=1)
    write_pgm(tmp_path / "a.pgm", left)
    write_pgm(tmp_path / "b.pgm", right)
    write_pgm(tmp_path / "c.pgm", left[:, :20])
    p = run(exe)
    assert p.returncode == 1 and p.stderr.startswith("usage: stereomatch [image 1] [image 2] "
                                                     "[threshold = 0.15] [square_width = 21] "
                                                     "[times = 32] [lines = 10]")
    p = run(exe, "missing.png", "b.pgm", cwd=tmp_path)
    assert p.returncode == 1 and p.stderr.startswith("error reading image missing.png:")
    p = run(exe, "a.pgm", "c.pgm", cwd=tmp_path)
    assert (p.returncode, p.stderr) == (1, "error: the two images must have equal width and height\n")
    for argv, msg in (
            (("abc",), "error: threshold must be a number\n"),
            ((0.15, "x"), "error: square_width must be a number\n"),
            ((0.15, 5, "y"), "error: times must be a number\n"),
            ((0.15, 5, 3, "z"), "error: lines must be a number\n"),
            ((1.5,), "error: threshold must be between 0 and 1\n"),
            ((-0.1,), "error: threshold must be between 0 and 1\n"),
            ((0.15, 25), "error: square width must not be higher than image width/height\n"),
            ((0.15, 33), "error: square width must not be higher than image width/height\n")):
        p = run(exe, "a.pgm", "b.pgm", *argv, cwd=tmp_path)
        assert (p.returncode, p.stderr) == (1, msg), argv
    # RGB input is rejected with the channel count
    rgb = b"\x89PNG\r\n\x1a\n"
    ihdr = struct.pack(">IIBBBBB", 2, 2, 8, 2, 0, 0, 0)
    raw = zlib.compress(b"\x00" + bytes(6) + b"\x00" + bytes(6))
    for tag, data in ((b"IHDR", ihdr), (b"IDAT", raw), (b"IEND", b"")):
        rgb += struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))
    (tmp_path / "rgb.png").write_bytes(rgb)
    p = run(exe, "rgb.png", "b.pgm", cwd=tmp_path)
    assert p.returncode == 1
    assert p.stderr == "error reading image rgb.png: wrong number of channels (3) (image must be grayscale)"


def test_stereopar_batch_argument_errors(programs, tmp_path):  # noqa: F811
    """argv / list validation of the batch host happens before any GPU call"""
    exe = programs["timing"]["stereopar"].parent / "stereopar-batch"
    p = run(exe, cwd=tmp_path)
    assert p.returncode == 1 and p.stderr.startswith("usage: stereopar-batch")
    p = run(exe, "nolist.txt", cwd=tmp_path)
    assert p.returncode == 1 and "error reading pair list nolist.txt:" in p.stderr
    (tmp_path / "empty.txt").write_text("# nothing\n")
    p = run(exe, "empty.txt", cwd=tmp_path)
    assert p.returncode == 1 and "the pair list is empty" in p.stderr
    (tmp_path / "bad.txt").write_text("only-one-name\n")
    p = run(exe, "bad.txt", cwd=tmp_path)
    assert p.returncode == 1 and "does not name two images" in p.stderr
    (tmp_path / "missing.txt").write_text("a.pgm b.pgm\n")
    p = run(exe, "missing.txt", cwd=tmp_path)
    assert p.returncode == 1 and "error reading image a.pgm:" in p.stderr
    p = run(exe, "-b", "0", "missing.txt", cwd=tmp_path)
    assert p.returncode == 1 and "-b must be a positive number" in p.stderr
    p = run(exe, "missing.txt", "2.0", cwd=tmp_path)
    assert p.returncode == 1     # image error comes first, as in the reference's argument order
