"""The image reader of the boundary (stereomatching_amd/host/image.c: its own inflate / PNG / PGM
decoder, standing in for stb_image behind read_image of src/image.h:25) parses untrusted files.
Built here with AddressSanitizer + UndefinedBehaviorSanitizer (CPU only) and fed truncated,
bit-flipped and absurd-dimension files: every one must come back as rc 1 with the reference's
message on stderr (src/image.c:22-31) or as a well-formed image -- never a sanitizer report."""
import struct
import subprocess
import zlib
from pathlib import Path

import numpy as np
import pytest

from stereomatching_amd.synth import make_pair, write_pgm
from tests.test_cli_cpu import write_png

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    out = tmp_path_factory.mktemp("fuzz") / "fuzz_image_reader"
    subprocess.check_call(["gcc", "-std=gnu11", "-g", "-O1", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           f"-I{ROOT / 'include'}", f"-I{ROOT / 'oracle'}",
                           str(ROOT / "tests" / "helpers" / "fuzz_image_reader.c"),
                           str(ROOT / "stereomatching_amd" / "host" / "image.c"), "-o", str(out), "-lm"])
    return out


def run(fuzzer, files):
    p = subprocess.run([str(fuzzer), *map(str, files)], capture_output=True, text=True,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "PATH": "/usr/bin:/bin"})
    assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    return [tuple(map(int, l.split())) for l in p.stdout.splitlines()], p.stderr


def png_chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data))


def test_well_formed_files_still_read(fuzzer, tmp_path):
    img, _ = make_pair(61, 37, 16, seed=5)
    write_png(tmp_path / "a.png", img, chunks=3)
    write_pgm(tmp_path / "a.pgm", img)
    res, _ = run(fuzzer, [tmp_path / "a.png", tmp_path / "a.pgm"])
    assert res == [(0, 61, 37), (0, 61, 37)]


def test_truncated_and_bit_flipped_files(fuzzer, tmp_path):
    rng = np.random.default_rng(7)
    img, _ = make_pair(53, 29, 16, seed=6)
    write_png(tmp_path / "ok.png", img, chunks=2)
    write_pgm(tmp_path / "ok.pgm", img)
    files = []
    for stem in ("ok.png", "ok.pgm"):
        data = (tmp_path / stem).read_bytes()
        for cut in sorted(set(int(c) for c in np.linspace(0, len(data) - 1, 40))):
            f = tmp_path / f"cut{cut}_{stem}"
            f.write_bytes(data[:cut])
            files.append(f)
        for i in range(150):
            b = bytearray(data)
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(0, len(b)))
                b[pos] ^= 1 << int(rng.integers(0, 8))
            f = tmp_path / f"flip{i}_{stem}"
            f.write_bytes(bytes(b))
            files.append(f)
    res, err = run(fuzzer, files)
    assert len(res) == len(files)
    for (rc, w, h), f in zip(res, files):
        assert rc in (0, 1), f
        if rc == 0:                    # a flip that left the file well-formed (or only changed pixels)
            assert (w, h) == (53, 29) or w * h <= 53 * 29 * 64, (f, w, h)
    # a PNG cut anywhere before its IEND chunk (the last 12 bytes; like stb_image this reader does not
    # insist on the trailing CRC) is an error, reported the way the reference reports a failed load
    n_png = len((tmp_path / "ok.png").read_bytes())
    for (rc, _, _), f in zip(res, files):
        if f.name.startswith("cut") and f.name.endswith("_ok.png") and int(f.name[3:].split("_")[0]) < n_png - 12:
            assert rc == 1, f.name
    assert "error reading image" in err


def test_absurd_dimensions_and_bad_streams(fuzzer, tmp_path):
    sig = b"\x89PNG\r\n\x1a\n"
    files = []

    def png(name, w, h, depth=8, ctype=0, idat=b"", interlace=0):
        ihdr = struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)
        (tmp_path / name).write_bytes(sig + png_chunk(b"IHDR", ihdr) + png_chunk(b"IDAT", idat) + png_chunk(b"IEND", b""))
        files.append(tmp_path / name)

    row = zlib.compress(b"\x00" + bytes(16))
    png("huge.png", 0x7fffffff, 0x7fffffff, idat=row)             # 4 Epx: must not be allocated
    png("wide.png", 0x40000000, 4, idat=row)                       # w * h overflows int
    png("zero_w.png", 0, 5, idat=row)
    png("zero_h.png", 5, 0, idat=row)
    png("short_data.png", 64, 64, idat=row)                        # far fewer bytes than 64 x 65
    png("garbage_zlib.png", 16, 1, idat=b"\x78\x9c" + bytes(range(50)))
    png("stored_overrun.png", 16, 1, idat=b"\x78\x01\x01\xff\xff\x00\x00" + bytes(8))   # stored block of 65535
    png("bad_filter.png", 16, 1, idat=zlib.compress(b"\x09" + bytes(16)))
    png("interlaced.png", 16, 1, idat=row, interlace=1)
    png("depth1.png", 16, 1, depth=1, idat=zlib.compress(b"\x00\x00\x00"))
    (tmp_path / "no_iend.png").write_bytes(sig + png_chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 0, 0, 0, 0)))
    files.append(tmp_path / "no_iend.png")
    (tmp_path / "chunk_len.png").write_bytes(sig + struct.pack(">I", 0xfffffff0) + b"IHDR" + bytes(13))
    files.append(tmp_path / "chunk_len.png")
    for name, text in (("huge.pgm", b"P5\n2000000000 2000000000\n255\n"), ("neg.pgm", b"P5\n-4 4\n255\n" + bytes(16)),
                       ("maxval.pgm", b"P5\n4 4\n65535\n" + bytes(32)), ("short.pgm", b"P5\n40 40\n255\n" + bytes(10)),
                       ("comment.pgm", b"P5\n# c\n4 # x\n4\n255\n" + bytes(16)), ("empty.pgm", b""),
                       ("ascii.pgm", b"P2\n2 2\n255\n1 2 3 4\n")):
        (tmp_path / name).write_bytes(text)
        files.append(tmp_path / name)
    res, err = run(fuzzer, files + [tmp_path / "does_not_exist.png"])
    by = {f.name: r for f, r in zip(files, res)}
    for name in ("huge.png", "wide.png", "zero_w.png", "zero_h.png", "short_data.png", "garbage_zlib.png",
                 "stored_overrun.png", "bad_filter.png", "no_iend.png", "chunk_len.png", "huge.pgm", "neg.pgm",
                 "short.pgm", "empty.pgm"):
        assert by[name][0] == 1, (name, by[name])
    assert res[-1][0] == 1 and "does_not_exist.png" in err          # perror-style line, src/image.c:22-25
