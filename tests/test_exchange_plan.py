"""The send / receive schedule of the multi-device fit behind the C signature (cmfrec_amd/csrc/exchange_plan.hpp, issued by
MultiDev::exchange in fit.hip inside one ncclGroup), checked without devices: every block reaches every peer exactly once, every
send has its matching receive with the same count, a device's receives tile its replica without touching its own block, and no
two devices start on the same peer.  (direct placement over xGMI; the hardware run is the driver's SCALE stage.)"""
import ctypes as C

import numpy as np
import pytest

from cmfrec_amd import _lib


def plan(bb, dtype=np.float64):
    lib = _lib.load(dtype)
    lib.cmfrec_hip_exchange_plan.restype = C.c_int
    bb = np.asarray(bb, np.int32)
    D = len(bb) - 1
    n = lib.cmfrec_hip_exchange_plan(C.c_int(D), bb.ctypes.data_as(C.c_void_p), None, C.c_int(0))
    out = np.zeros((max(n, 1), 5), np.int32)
    assert lib.cmfrec_hip_exchange_plan(C.c_int(D), bb.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int(n)) == n
    return [tuple(int(v) for v in r) for r in out[:n]]


CASES = [
    [0, 10, 20],                                  # D = 2, equal blocks
    [0, 7, 7, 30],                                # D = 3 with an empty block
    [0, 5, 9, 30],                                # D = 3, unequal
    [0, 1250000, 2500000, 3750000, 5000000, 6250000, 7500000, 8750000, 10000000],       # config 4's users on 8 devices
    [0, 3, 50, 51, 400, 401, 402, 9000, 9001],    # D = 8, very unequal (nnz-balanced item blocks look like this)
    [0, 100],                                     # D = 1: nothing to do
]


@pytest.mark.parametrize("bb", CASES)
def test_every_send_has_its_receive(bb):
    ops = plan(bb)
    D = len(bb) - 1
    rows = [bb[d + 1] - bb[d] for d in range(D)]
    sends = {(d, e): (first, cnt) for d, e, s, first, cnt in ops if s == 1}
    recvs = {(d, e): (first, cnt) for d, e, s, first, cnt in ops if s == 0}
    assert len(sends) == sum(1 for d, e, s, *_ in ops if s == 1), "one send per ordered pair at most"
    assert len(recvs) == sum(1 for d, e, s, *_ in ops if s == 0)
    for d in range(D):
        for e in range(D):
            if d == e:
                assert (d, e) not in sends and (d, e) not in recvs
                continue
            if rows[d] > 0:
                # d sends its own block to e; e receives it into the rows d owns, same count
                assert sends[(d, e)] == (bb[d], rows[d])
                assert recvs[(e, d)] == (bb[d], rows[d])
            else:
                assert (d, e) not in sends and (e, d) not in recvs
    # the receives of a device tile everything but its own block
    for d in range(D):
        got = np.zeros(bb[-1], np.int8)
        for (dev, peer), (first, cnt) in recvs.items():
            if dev == d:
                got[first:first + cnt] += 1
        exp = np.ones(bb[-1], np.int8)
        exp[bb[d]:bb[d + 1]] = 0
        assert np.array_equal(got, exp)


@pytest.mark.parametrize("bb", CASES[:5])
def test_peers_are_visited_in_rotated_order(bb):
    """device d starts on d + 1, d + 2, ...: at every position of the visiting order the D devices talk to D distinct peers (on an
    xGMI node every pair has its own link, so no link carries two transfers of the same round)"""
    ops = plan(bb)
    D = len(bb) - 1
    order = {d: [] for d in range(D)}
    for d, e, s, first, cnt in ops:
        if e not in order[d]:
            order[d].append(e)
    nonempty = [d for d in range(D) if bb[d + 1] > bb[d]]
    if len(nonempty) == D:
        for pos in range(D - 1):
            assert sorted(order[d][pos] for d in range(D)) == list(range(D))
        for d in range(D):
            assert order[d] == [(d + o) % D for o in range(1, D)]


def test_both_precisions_export_the_plan():
    assert plan([0, 4, 9], np.float32) == plan([0, 4, 9], np.float64)
