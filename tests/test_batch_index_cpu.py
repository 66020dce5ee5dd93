"""stereopar-batch's pair -> device mapping (stereomatching_amd/host/batch_index.h; SURVEY.md 8e:
pair j -> device j mod n_devices) checked on the CPU for n_devices > 1: every pair of the list is
owned by exactly one device, each device walks its own pairs in list order, and with a repeat count
it walks them round and round."""
import ctypes as C
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def idx(tmp_path_factory):
    d = tmp_path_factory.mktemp("bidx")
    (d / "w.c").write_text('#include "batch_index.h"\n'
                           "int pairs_of_rank(int n, int r, int k) { return batch_pairs_of_rank(n, r, k); }\n"
                           "int global_index(int r, long s, int m, int k) { return batch_global_index(r, s, m, k); }\n")
    so = d / "w.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", f"-I{ROOT / 'stereomatching_amd' / 'host'}",
                           str(d / "w.c"), "-o", str(so)])
    lib = C.CDLL(str(so))
    lib.global_index.argtypes = [C.c_int, C.c_long, C.c_int, C.c_int]
    return lib


@pytest.mark.parametrize("n_pairs,n_devices", [(64, 8), (65, 8), (7, 8), (8, 8), (1, 1), (100, 3), (5, 2), (64, 1)])
def test_every_pair_has_exactly_one_owner(idx, n_pairs, n_devices):
    n_devices = min(n_devices, n_pairs)          # main() never uses more devices than pairs
    owner = {}
    for rank in range(n_devices):
        mine = idx.pairs_of_rank(n_pairs, rank, n_devices)
        assert mine == len(range(rank, n_pairs, n_devices))
        seqs = [idx.global_index(rank, s, mine, n_devices) for s in range(mine)]
        assert seqs == list(range(rank, n_pairs, n_devices))       # list order, stride n_devices
        for j in seqs:
            assert j not in owner
            owner[j] = rank
        # repeats walk the same pairs again
        assert [idx.global_index(rank, s, mine, n_devices) for s in range(mine, 2 * mine)] == seqs
    assert sorted(owner) == list(range(n_pairs))
    assert all(owner[j] == j % n_devices for j in owner)
    # shares differ by at most one pair
    shares = [idx.pairs_of_rank(n_pairs, r, n_devices) for r in range(n_devices)]
    assert max(shares) - min(shares) <= 1 and sum(shares) == n_pairs
