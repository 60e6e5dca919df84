"""CPU: the oracle's kd-tree restatement (oracle/okd.c) against the REFERENCE kdtree.c.

Two pins: (1) the committed golden vectors tests/golden/kd_golden.npz, which are outputs of the
reference library itself (scripts/make_golden.py); (2) when oracle/_ref/libkdtree_ref.so is
present (build container), a live comparison on fresh random data.  Checked: range hit lists
INCLUDING iteration order (reverse discovery, kdtree.c:282/759-777), inclusive radius, 1-NN
winners including exact fp32 ties (kdtree.c:343) and the degenerate tree of a sorted cloud.
"""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "kd_golden.npz")


class OwnKd:
    def __init__(self, oa):
        L = C.CDLL(oa.LIB)
        L.okd_create.restype = C.c_void_p
        L.okd_free.argtypes = [C.c_void_p]
        L.okd_insert2.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.okd_nearest2.restype = C.c_void_p
        L.okd_nearest2.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.okd_nearest_range2.restype = C.c_void_p
        L.okd_nearest_range2.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        for f in ("okd_res_size", "okd_res_end", "okd_res_next", "okd_res_free"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.okd_res_item_data.restype = C.c_void_p
        L.okd_res_item_data.argtypes = [C.c_void_p]
        self.L, self.t = L, L.okd_create()

    def insert(self, xy):
        for i, (x, y) in enumerate(xy):
            self.L.okd_insert2(self.t, float(x), float(y), C.c_void_p(i + 1))

    def nearest(self, x, y):
        r = self.L.okd_nearest2(self.t, float(x), float(y))
        v = self.L.okd_res_item_data(r)
        self.L.okd_res_free(r)
        return int(v) - 1

    def range(self, x, y, rad):
        r = self.L.okd_nearest_range2(self.t, float(x), float(y), float(rad))
        out = []
        assert self.L.okd_res_size(r) >= 0
        while not self.L.okd_res_end(r):
            out.append(int(self.L.okd_res_item_data(r)) - 1)
            self.L.okd_res_next(r)
        self.L.okd_res_free(r)
        return out


@pytest.mark.parametrize("case", ["jitter", "lattice", "sorted"])
def test_okd_matches_reference_golden(oa, case):
    g = np.load(GOLD)
    pts, q, rad = g[f"{case}_pts"], g[f"{case}_q"], g[f"{case}_rad"]
    kd = OwnKd(oa)
    kd.insert(pts)
    nn = np.array([kd.nearest(x, y) for x, y in q], np.int32)
    assert np.array_equal(nn, g[f"{case}_nn"])
    hits, offs = g[f"{case}_hits"], g[f"{case}_offs"]
    for i, ((x, y), r) in enumerate(zip(q, rad)):
        assert kd.range(x, y, r) == hits[offs[i]:offs[i + 1]].tolist(), (case, i)
    if case == "lattice":
        # the golden really contains fp32 distance ties for the nearest neighbour
        d2 = ((pts[None, :, 0] - q[:50, None, 0]) ** 2 + (pts[None, :, 1] - q[:50, None, 1]) ** 2)
        srt = np.sort(d2, axis=1)
        assert (srt[:, 0] == srt[:, 1]).any()


def test_okd_matches_reference_live(oa):
    if not os.path.exists(oa.REF_KD):
        pytest.skip("oracle/_ref/libkdtree_ref.so not built (reference tree absent)")
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_golden", os.path.join(os.path.dirname(__file__), "..", "scripts", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    rng = np.random.default_rng(123)
    pts = rng.normal(0, 3, size=(5000, 2)).astype(np.float32)
    pts[::7] = np.round(pts[::7] * 2) / 2  # duplicates and ties
    ref, own = mg.RefKd(), OwnKd(oa)
    ref.insert(pts)
    own.insert(pts)
    q = rng.normal(0, 3.5, size=(600, 2)).astype(np.float32)
    q[::5] = np.round(q[::5] * 4) / 4
    for x, y in q:
        assert own.nearest(x, y) == ref.nearest(x, y)
        for r in (0.1, 0.3, 0.75):
            assert own.range(x, y, r) == ref.range(x, y, r)
    ref.close()
