"""The launch-program hazard checker's own logic (tce_rvos_amd/hazard.py), on the CPU: interval arithmetic against brute
force, happens-before from recorded events, the access models of the argument blocks, and coverage of the C ABI (every
entry point of include/tce_rvos.h is either modelled or declared launch-free).  The check of the real program runs on the
GPU (tests/test_e2e_gpu.py::test_launch_program_is_race_free)."""
import ctypes as C
import random

import numpy as np

from tce_rvos_amd import _lib, hazard
from tce_rvos_amd.hazard import Recorder, dense, overlap, strided, union


def _bytes(iv):
    s = set()
    for lo, hi in iv.tolist():
        s.update(range(lo, hi))
    return s


def test_interval_sets_match_brute_force():
    rnd = random.Random(0)
    for _ in range(200):
        base = rnd.randrange(1, 50)
        run = rnd.randrange(1, 9)
        dims = [(rnd.randrange(1, 5), rnd.randrange(0, 40)) for _ in range(rnd.randrange(0, 3))]
        iv = strided(base, run, *dims)
        want = set()
        starts = [base]
        for c, st in dims:
            starts = [s + i * st for s in starts for i in range(c)]
        for s in starts:
            want.update(range(s, s + run))
        assert _bytes(iv) == want
        assert (iv[1:, 0] > iv[:-1, 1]).all()  # merged: disjoint, not touching, sorted
        other = strided(rnd.randrange(1, 80), rnd.randrange(1, 9), (rnd.randrange(1, 6), rnd.randrange(1, 30)))
        ov = overlap(iv, other)
        common = _bytes(iv) & _bytes(other)
        assert (ov is None) == (not common)
        if ov is not None:
            assert set(range(*ov)) <= common
    assert overlap(dense(0, 10), dense(100, 4)) is None  # NULL pointer = empty set
    assert len(union(dense(8, 4), dense(12, 4))) == 1


def _rec_pair(edge, same_range=True):
    """stream 1 writes X; stream 2 reads X (or a disjoint Y), with / without a record->wait edge between them."""
    r = Recorder()
    X, Y = dense(0x1000, 256), dense(0x2000, 256)
    r.launch("producer", 1, [], [X], "a")
    if edge:
        ev = object()
        r.record_event(ev, 1)
        r.wait_event(ev, 2)
    r.launch("consumer", 2, [X if same_range else Y], [Y], "b")
    return r.analyse()


def test_unordered_conflict_is_found_and_an_edge_clears_it():
    bad = _rec_pair(edge=False)
    assert not bad.clean and bad.n_conflicts == 1 and bad.conflicts[0][0] == "write/read"
    assert "producer" in str(bad) and "consumer" in str(bad)
    assert _rec_pair(edge=True).clean
    assert _rec_pair(edge=False, same_range=False).clean  # unordered but disjoint
    assert _rec_pair(edge=False).unordered_pairs == 1 and _rec_pair(edge=True).unordered_pairs == 0


def test_happens_before_is_transitive_and_respects_issue_order():
    # main(1) forks side(2), side forks side2(3); side2 joins side, side joins main: main's later launch is ordered after
    # side2's write (transitively); a launch issued on main BEFORE the join is not.
    r = Recorder()
    X = dense(0x4000, 64)
    e1, e2, e3, e4 = object(), object(), object(), object()
    r.launch("m0", 1, [], [dense(0x100, 4)], "")
    r.record_event(e1, 1); r.wait_event(e1, 2)
    r.launch("s0", 2, [], [dense(0x200, 4)], "")
    r.record_event(e2, 2); r.wait_event(e2, 3)
    r.launch("t0 writes X", 3, [], [X], "")
    r.launch("m1 reads X early", 1, [X], [], "")  # not ordered with t0
    r.record_event(e3, 3); r.wait_event(e3, 2)
    r.record_event(e4, 2); r.wait_event(e4, 1)
    r.launch("m2 reads X after the joins", 1, [X], [], "")
    rep = r.analyse()
    assert rep.n_conflicts == 1
    kind, a, b, _ = rep.conflicts[0]
    assert a.name.startswith("t0") and b.name.startswith("m1")


def test_arena_reuse_across_branches_is_a_conflict():
    # branch A still reads a buffer when the main chain releases its arena range and hands the same bytes to a new tensor
    r = Recorder()
    buf = dense(0x9000, 1024)
    e = object()
    r.launch("write buf", 1, [], [buf], "")
    r.record_event(e, 1); r.wait_event(e, 2)
    r.launch("branch reads buf", 2, [buf], [dense(0x20000, 16)], "")
    r.launch("main re-uses the range", 1, [], [dense(0x9100, 64)], "")  # release() -> alloc() gave out [0x9100, 0x9140)
    rep = r.analyse()
    assert rep.n_conflicts == 1 and rep.conflicts[0][0] == "read/write"


def test_host_sync_orders_everything_issued_before_it():
    r = Recorder()
    X = dense(0x1000, 16)
    r.launch("w", 1, [], [X], "")
    r.host_sync()
    r.launch("r", 2, [X], [], "")
    r.launch("r on a stream first seen after the sync", 7, [X], [], "")
    assert r.analyse().clean
    r2 = Recorder()
    r2.launch("w", 1, [], [X], "")
    ev = object()
    r2.record_event(ev, 3)  # an event of ANOTHER stream: waiting for it says nothing about stream 1
    r2.host_sync(r2.events[id(ev)][1])
    r2.launch("r", 2, [X], [], "")
    assert not r2.analyse().clean


def test_every_entry_point_is_modelled_or_declared_launch_free():
    names = set(_lib.SIGNATURES) | set(_lib.DEBUG_SIGNATURES)
    missing = names - set(hazard.MODELS) - hazard.NOT_LAUNCHES
    assert not missing, f"entry points without a hazard access model: {sorted(missing)}"
    assert set(hazard.MODELS) <= set(_lib.SIGNATURES)
    assert not (set(hazard.MODELS) & hazard.NOT_LAUNCHES)


def test_gemm_model_separates_level_slices_of_one_tensor():
    # the four input projections write disjoint level slices of src [T, S, 256] with a frame stride of S*256 floats:
    # bounding boxes overlap, the exact strided sets must not
    T, S, D = 5, 4820, 256
    base = 0x10000000
    def g(start_row, rows):
        a = _lib.GemmArgs()
        a.A, a.W, a.C = 0x5000000, 0x6000000, base + start_row * D * 4
        a.M, a.N, a.K, a.lda, a.ldw, a.ldc = rows, D, D, D, D, D
        a.batch, a.sA, a.sC = T, rows * D, S * D
        return hazard.MODELS["tce_gemm_f32"]((C.byref(a), 0))
    (_, w0), (_, w1) = g(0, 3600), g(3600, 920)
    assert overlap(union(*w0), union(*w1)) is None
    (_, w2) = g(3599, 920)
    assert overlap(union(*w0), union(*w2)) is not None
    assert len(union(*w0)) == T  # one dense run per frame


def test_fewrow_and_copy_models():
    q = _lib.FewRowArgs()
    q.x, q.ldx, q.R, q.K, q.nseg = 0x1000, 256, 40, 256, 2
    q.res, q.ldres = 0x90000, 256
    for i, (out, N, ldo) in enumerate(((0x90000, 256, 256), (0xA0000, 2, 2))):
        sg = q.seg[i]
        sg.W, sg.out, sg.N, sg.ldw, sg.ldo = 0x200000 + i * 0x100000, out, N, 256, ldo
    rd, wr = hazard.MODELS["tce_fewrow_linear_f32"]((C.byref(q), 0))
    W = union(*wr)
    assert _bytes(dense(0x90000, 40 * 256 * 4)) <= _bytes(W) and (W[:, 1] - W[:, 0]).sum() == 40 * 256 * 4 + 40 * 2 * 4
    assert overlap(union(*rd), dense(0x90000, 4)) is not None  # the residual is read
    segs = (_lib.CopySeg * 2)()
    segs[0].src, segs[0].dst, segs[0].rows, segs[0].row_words, segs[0].src_pitch_words = 0x1000, 0x8000, 3, 2, 10
    segs[1].src, segs[1].dst, segs[1].rows, segs[1].row_words, segs[1].src_pitch_words = 0x3000, 0x9000, 1, 5, 5
    rd, wr = hazard.MODELS["tce_copy_segments"]((segs, 2, 0))
    assert _bytes(union(*rd)) == set().union(*[range(0x1000 + r * 40, 0x1000 + r * 40 + 8) for r in range(3)]) | set(range(0x3000, 0x3014))
    assert _bytes(union(*wr)) == set(range(0x8000, 0x8018)) | set(range(0x9000, 0x9014))
