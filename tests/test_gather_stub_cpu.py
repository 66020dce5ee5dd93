"""sm_gather_maps / sm_broadcast (csrc/sm_gather.hip) against a HOST stand-in for librccl (tests/rccl_stub.c).

No 8-GPU node has been available in any round, and a one-GPU box can only make a one-rank communicator, which sends
nothing: the grouped ncclSend / ncclRecv sequence of the collection had never executed anywhere (round-4 review).
The stand-in restates RCCL's group semantics on host memory -- operations are recorded between ncclGroupStart and
ncclGroupEnd and paired there, an unpaired one is an error where RCCL would hang -- so the ORDER and PAIRING the
library issues for n = 2, 3, 5 ranks with even, uneven and empty shares is checked here, without a GPU.
Runs in a process of its own: the RCCL library of a process is chosen once (sm_comm_set_rccl_library)."""
import subprocess
import sys
import textwrap
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

SCRIPT = textwrap.dedent(r'''
    import ctypes as C, subprocess, sys, tempfile, os
    import numpy as np
    root = sys.argv[1]
    sys.path.insert(0, root)
    from stereomatching_amd import capi
    lib, check, vp = capi.lib, capi.check, C.c_void_p
    td = tempfile.mkdtemp()
    so = os.path.join(td, "librccl_stub.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(root, "tests", "rccl_stub.c")])
    check(lib.sm_comm_set_rccl_library(so.encode()))
    stub = C.CDLL(so)
    stub.stub_log.restype = C.c_char_p
    rng = np.random.default_rng(5)

    def gather(shares):
        n = len(shares)
        comm = vp()
        check(lib.sm_comm_create((C.c_int * n)(*range(n)), n, C.byref(comm)))
        assert lib.sm_comm_size(comm) == n
        src = [rng.integers(0, 256, max(1, s), dtype=np.uint8) for s in shares]
        dst = np.full(sum(shares) + 8, 0xEE, np.uint8)
        stub.stub_log_reset()
        check(lib.sm_gather_maps(comm, (vp * n)(*[a.ctypes.data for a in src]), (C.c_size_t * n)(*shares),
                                 vp(dst.ctypes.data), None))
        want = np.concatenate([a[:s] for a, s in zip(src, shares)]) if sum(shares) else np.zeros(0, np.uint8)
        assert np.array_equal(dst[:sum(shares)], want), shares
        assert (dst[sum(shares):] == 0xEE).all(), "wrote past the collected maps"
        log = stub.stub_log().decode().split()
        lib.sm_comm_destroy(comm)
        return log

    # two ranks: the root copies its own share, rank 1 sends, the root receives -- one group
    assert gather([100, 60]) == ["start", "send(1>0,60)", "recv(0<1,60)", "end"]
    # three ranks, uneven shares: every sender paired with the root's receive, in rank order, ONE group
    assert gather([7, 300, 41]) == ["start", "send(1>0,300)", "recv(0<1,300)", "send(2>0,41)", "recv(0<2,41)", "end"]
    # ranks with nothing to send take no part; an empty root share moves the first receive to offset 0
    assert gather([0, 50, 0]) == ["start", "send(1>0,50)", "recv(0<1,50)", "end"]
    assert gather([64, 0, 64, 0, 3]) == ["start", "send(2>0,64)", "recv(0<2,64)", "send(4>0,3)", "recv(0<4,3)", "end"]
    assert gather([0, 0]) == ["start", "end"]
    # one rank: no RCCL call at all
    assert gather([33]) == []

    # broadcast: every rank's call inside one group, the root's buffer reaches all of them
    n = 3
    comm = vp()
    check(lib.sm_comm_create((C.c_int * n)(*range(n)), n, C.byref(comm)))
    bufs = [np.full(16, r, np.uint8) for r in range(n)]
    bufs[0][:] = np.arange(16)
    stub.stub_log_reset()
    check(lib.sm_broadcast(comm, (vp * n)(*[b.ctypes.data for b in bufs]), 16, None))
    assert all(np.array_equal(b, np.arange(16)) for b in bufs)
    assert stub.stub_log().decode().split() == ["start", "bcast(0<0,16)", "bcast(1<0,16)", "bcast(2<0,16)", "end"]
    # a group that fails is reported, with RCCL's text, not swallowed
    stub.stub_poison_next_group()
    src = [np.zeros(4, np.uint8) for _ in range(n)]
    dst = np.zeros(12, np.uint8)
    rc = lib.sm_gather_maps(comm, (vp * n)(*[a.ctypes.data for a in src]), (C.c_size_t * n)(4, 4, 4), vp(dst.ctypes.data), None)
    assert rc == capi.SM_ERR_HIP and b"ncclGroupEnd failed" in lib.sm_last_error(), lib.sm_last_error()
    lib.sm_comm_destroy(comm)
    # the library of a process is chosen once
    assert lib.sm_comm_set_rccl_library(b"/nonexistent.so") == capi.SM_ERR_ARG
    print("stub-ok")
''')


def test_gather_maps_group_logic_against_a_host_stand_in_for_rccl():
    p = subprocess.run([sys.executable, "-c", SCRIPT, str(ROOT)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "stub-ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
