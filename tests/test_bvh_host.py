"""The host builder of the BVH of short triangle runs (raytracer_amd/csrc/rt_bvh.hip: build_triangle_bvh_host) on the CPU:
the tree's invariants — what the traversal in rt_kernels.hip relies on — on random and on awkward triangle sets."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import capi

LEAF = 0x80000000


def build(tris):
    tris = np.ascontiguousarray(tris, dtype=np.float32)
    n = len(tris)
    rec = np.zeros((n - 1, 16), dtype=np.uint32)
    fn = capi.hip_lib().rt_debug_bvh_host
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]
    assert fn(tris.ctypes.data_as(C.POINTER(C.c_float)), n, rec.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
    return rec


def tri_box(t):
    p = np.stack([t[0:3], t[0:3] + t[3:6], t[0:3] + t[6:9]])
    return p.min(axis=0), p.max(axis=0)


def check(tris, rec):
    n = len(tris)
    f = rec.view(np.float32)
    seen_tri, seen_rec = np.zeros(n, dtype=int), np.zeros(n - 1, dtype=int)
    max_depth = 0

    def walk(word, depth):
        nonlocal max_depth
        max_depth = max(max_depth, depth)
        if word & LEAF:
            i = int(word & ~np.uint32(LEAF))
            seen_tri[i] += 1
            return tri_box(tris[i])
        r = int(word)
        seen_rec[r] += 1
        lo0, c0, hi0, c1, lo1, hi1 = f[r, 0:3], rec[r, 3], f[r, 4:7], rec[r, 7], f[r, 8:11], f[r, 12:15]
        a = walk(c0, depth + 1)
        b = walk(c1, depth + 1)
        # a child's box holds everything below it (leaf boxes are grown, never shrunk)
        assert np.all(lo0 <= a[0]) and np.all(hi0 >= a[1]), r
        assert np.all(lo1 <= b[0]) and np.all(hi1 >= b[1]), r
        # the slab test's distance slack rides in the spare words: 1e-4 of the box's extents
        assert f[r, 11] == pytest.approx(1e-4 * float(np.sum(hi0 - lo0)), rel=1e-5, abs=1e-30)
        assert f[r, 15] == pytest.approx(1e-4 * float(np.sum(hi1 - lo1)), rel=1e-5, abs=1e-30)
        return np.minimum(lo0, lo1), np.maximum(hi0, hi1)

    walk(np.uint32(0), 1)
    assert np.all(seen_tri == 1) and np.all(seen_rec == 1)       # every triangle in one leaf, every record reached once
    assert max_depth <= 60                                       # the traversal's stack holds 64 entries and more
    return max_depth


@pytest.mark.parametrize("n,seed", [(2, 0), (3, 1), (64, 2), (257, 3), (1024, 4)])
def test_random_triangles(n, seed):
    rng = np.random.default_rng(seed)
    tris = np.concatenate([rng.uniform(-10, 10, (n, 3)), rng.uniform(-0.5, 0.5, (n, 6))], axis=1).astype(np.float32)
    depth = check(tris, build(tris))
    assert depth <= 4 * int(np.ceil(np.log2(n))) + 4


def test_large_and_small_triangles_are_separated_early():
    # C2's shape: two huge floor triangles among many small ones — the surface-area tree peels the large ones off near the
    # root, so that a ray through the small ones does not test the floor's box at every level
    rng = np.random.default_rng(5)
    small = np.concatenate([rng.uniform(-1, 1, (60, 3)), rng.uniform(-0.05, 0.05, (60, 6))], axis=1)
    floor = np.array([[-50, -1, -50, 100, 0, 0, 0, 0, 100], [50, -1, 50, -100, 0, 0, 0, 0, -100]], dtype=np.float64)
    tris = np.concatenate([small[:30], floor, small[30:]]).astype(np.float32)
    rec = build(tris)
    check(tris, rec)

    def depth_of(target):
        stack = [(np.uint32(0), 1)]
        while stack:
            word, d = stack.pop()
            if word & LEAF:
                if int(word & ~np.uint32(LEAF)) == target:
                    return d
            else:
                stack += [(rec[int(word), 3], d + 1), (rec[int(word), 7], d + 1)]
    assert depth_of(30) <= 4 and depth_of(31) <= 4


def test_identical_and_degenerate_triangles_still_give_a_bounded_tree():
    same = np.tile(np.array([[1, 2, 3, 1, 0, 0, 0, 1, 0]], dtype=np.float32), (500, 1))      # every centroid equal
    assert check(same, build(same)) <= 60
    flat = np.zeros((300, 9), dtype=np.float32)                                              # zero-area triangles at the origin
    assert check(flat, build(flat)) <= 60
    line = np.zeros((400, 9), dtype=np.float32)
    line[:, 0] = np.arange(400)                                                              # collinear centroids
    line[:, 3:9] = [0.1, 0, 0, 0, 0.1, 0]
    assert check(line, build(line)) <= 60
