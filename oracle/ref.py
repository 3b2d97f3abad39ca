"""oracle/ref.py -- TEST INFRASTRUCTURE (dev container only).

ctypes wrapper over oracle/_ref/libref_<W>x<H>.so, i.e. the REFERENCE's own hot path compiled where it lies
(oracle/ref_build/build_ref.sh).  Used by tools/make_golden.py and by the not-gpu tests that pin the CPU
restatement (oracle/rt_oracle.cpp) against the reference.  It is absent on the GPU box (no /root/reference there)
and nothing in the product imports it.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("RT_REF_ROOT", "/root/reference/Raytracer")


def available(w=1280, h=720):
    return os.path.isdir(REF_ROOT) and os.path.exists(os.path.join(HERE, "_ref", "libref_%dx%d.so" % (w, h)))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Ref:
    """One resolution per library (the reference #defines SCRWIDTH/SCRHEIGHT, camera.h:4-5)."""
    _cache = {}

    def __new__(cls, w=1280, h=720):
        if (w, h) in cls._cache:
            return cls._cache[(w, h)]
        self = super().__new__(cls)
        self.L = C.CDLL(os.path.join(HERE, "_ref", "libref_%dx%d.so" % (w, h)))
        os.environ["RT_REF_ROOT"] = REF_ROOT
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(1)
        os.dup2(devnull, 1)  # the reference prints while loading (scene.h:78, 89, 224)
        try:
            r = self.L.ref_init(REF_ROOT.encode())
        finally:
            os.dup2(saved, 1)
            os.close(devnull)
            os.close(saved)
        assert r == 0, "ref_init failed: %d" % r
        self.W, self.H = self.L.ref_width(), self.L.ref_height()
        assert (self.W, self.H) == (w, h)
        self.L.ref_bvh_create.restype = C.c_void_p
        self.L.ref_sample_seed.restype = C.c_uint
        c = np.zeros(7, np.int32)
        self.L.ref_counts(_p(c))
        self.counts = dict(zip(("quads", "spheres", "cubes", "planes", "triangles", "nodes_used", "nodes_used_qbvh"), c.tolist()))
        cls._cache[(w, h)] = self
        return self

    def sizes(self):
        s = np.zeros(6, np.int32)
        self.L.ref_sizes(_p(s))
        return dict(zip(("Ray", "BVHNode", "GPUBVHNode", "SIMD_BVH_Node", "Triangle", "mat4"), s.tolist()))

    def scene_arrays(self):
        c = self.counts
        total = c["quads"] + c["spheres"] + c["cubes"] + c["planes"] + c["triangles"]
        nb = c["spheres"] + c["triangles"]
        a = {
            "tri_verts": np.zeros((c["triangles"], 3, 4), np.float32), "sphere_info": np.zeros((c["spheres"], 4), np.float32),
            "sphere_invr": np.zeros(c["spheres"], np.float32), "quad_inv_T": np.zeros((c["quads"], 16), np.float32),
            "quad_size": np.zeros(c["quads"], np.float32), "cube_M": np.zeros((c["cubes"], 16), np.float32),
            "cube_inv_M": np.zeros((c["cubes"], 16), np.float32), "cube_b": np.zeros((c["cubes"], 2, 4), np.float32),
            "primitives": np.zeros((total, 4), np.float32), "albedo": np.zeros((total, 4), np.float32),
            "material_type": np.zeros(total, np.int32), "ior": np.zeros(total, np.float32), "absorption": np.zeros((total, 4), np.float32),
            "light_corners": np.zeros((3, 4), np.float32), "light": np.zeros(5, np.float32),
            "nodes": np.zeros((nb * 2 + 1, 8), np.uint32), "prim_idx": np.zeros(nb, np.int32), "texture": np.zeros((64, 128), np.uint32),
        }
        self.L.ref_scene_arrays(*[_p(a[k]) for k in ("tri_verts", "sphere_info", "sphere_invr", "quad_inv_T", "quad_size", "cube_M", "cube_inv_M",
                                                     "cube_b", "primitives", "albedo", "material_type", "ior", "absorption", "light_corners", "light",
                                                     "nodes", "prim_idx", "texture")])
        a["nodes"] = a["nodes"][:c["nodes_used"]]
        return a

    def camera(self):
        cam = np.zeros(16, np.float32)
        self.L.ref_camera(_p(cam))
        return cam

    def quad_T(self):
        t = np.zeros(16, np.float32)
        self.L.ref_quad_T(_p(t))
        return t

    def seeds(self, n=None):
        n = n or self.W * self.H
        s = np.zeros(n, np.uint32)
        self.L.ref_seeds(_p(s), n)
        return s

    def rng_floats(self, s0, n):
        out = np.zeros(n, np.float32)
        self.L.ref_rng_floats(C.c_uint(s0), n, _p(out))
        return out

    def sample_seed(self, s0, s):
        return self.L.ref_sample_seed(C.c_uint(s0), C.c_uint(s))

    def primary_rays(self):
        n = self.W * self.H
        O, D, rD = (np.zeros((n, 3), np.float32) for _ in range(3))
        self.L.ref_primary_rays(_p(O), _p(D), _p(rD))
        return O, D, rD

    def primary_hits(self, qbvh):
        n = self.W * self.H
        obj, t = np.zeros(n, np.int32), np.zeros(n, np.float32)
        self.L.ref_primary_hits(int(qbvh), _p(obj), _p(t))
        return obj, t

    def find_nearest(self, O, D, tmax, qbvh=False):
        O, D, tmax = (np.ascontiguousarray(x, np.float32) for x in (O, D, tmax))
        n = len(tmax)
        obj, t = np.zeros(n, np.int32), np.zeros(n, np.float32)
        self.L.ref_find_nearest(int(qbvh), n, _p(O), _p(D), _p(tmax), _p(obj), _p(t))
        return obj, t

    def is_occluded(self, O, D, tmax, qbvh=False):
        O, D, tmax = (np.ascontiguousarray(x, np.float32) for x in (O, D, tmax))
        n = len(tmax)
        occ = np.zeros(n, np.int32)
        self.L.ref_is_occluded(int(qbvh), n, _p(O), _p(D), _p(tmax), _p(occ))
        return occ

    def hit_attribs(self, obj, I, wo):
        obj = np.ascontiguousarray(obj, np.int32)
        I, wo = np.ascontiguousarray(I, np.float32), np.ascontiguousarray(wo, np.float32)
        n = len(obj)
        N, alb, mat = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        self.L.ref_hit_attribs(n, _p(obj), _p(I), _p(wo), _p(N), _p(alb), _p(mat))
        return N, alb, mat

    def sample(self, seeds0, s_begin, s_count, acc=None, qbvh=False):
        if acc is None:
            acc = np.zeros((self.H, self.W, 4), np.float32)
        seeds0 = np.ascontiguousarray(seeds0, np.uint32)
        self.L.ref_sample(int(qbvh), _p(seeds0), s_begin, s_count, _p(acc))
        return acc

    def sample_strided(self, seeds0, s_begin, s_count, stride, qbvh=False):
        acc = np.zeros((self.H, self.W, 4), np.float32)
        seeds0 = np.ascontiguousarray(seeds0, np.uint32)
        self.L.ref_sample_strided(int(qbvh), _p(seeds0), s_begin, s_count, stride, _p(acc))
        return acc

    def trace(self, vis, depth=5, qbvh=False, aa=False):
        out = np.zeros((self.H, self.W, 4), np.float32)
        if aa:
            self.L.ref_trace_aa(int(qbvh), depth, _p(out))
        else:
            self.L.ref_trace(int(qbvh), vis, depth, _p(out))
        return out

    def finalize(self, acc, f):
        acc = np.ascontiguousarray(acc, np.float32).reshape(-1, 4)
        out = np.zeros(len(acc), np.uint32)
        self.L.ref_finalize(len(acc), _p(acc), f, _p(out))
        return out

    def postfx(self, pixels, which, amount):
        px = np.ascontiguousarray(pixels, np.uint32).copy()
        h, w = px.shape
        self.L.ref_postfx(_p(px), w, h, which, C.c_float(amount))
        return px

    # ---- unit level
    def _rays(self, O, D, t):
        return [np.ascontiguousarray(x, np.float32) for x in (O, D, t)]

    def tri_intersect(self, O, D, t, v9):
        O, D, t = self._rays(O, D, t)
        v9 = np.ascontiguousarray(v9, np.float32)
        hit, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.ref_tri_intersect(len(t), _p(O), _p(D), _p(t), _p(v9), _p(hit), _p(tout))
        return hit, tout

    def tri_derived(self, v9):
        v9 = np.ascontiguousarray(v9, np.float32).reshape(-1, 9)
        N, c = np.zeros((len(v9), 3), np.float32), np.zeros((len(v9), 3), np.float32)
        self.L.ref_tri_derived(len(v9), _p(v9), _p(N), _p(c))
        return N, c

    def sphere_intersect(self, O, D, t, pr4):
        O, D, t = self._rays(O, D, t)
        pr4 = np.ascontiguousarray(pr4, np.float32)
        hit, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.ref_sphere_intersect(len(t), _p(O), _p(D), _p(t), _p(pr4), _p(hit), _p(tout))
        return hit, tout

    def plane_intersect(self, O, D, t, nd4):
        O, D, t = self._rays(O, D, t)
        nd4 = np.ascontiguousarray(nd4, np.float32)
        hit, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.ref_plane_intersect(len(t), _p(O), _p(D), _p(t), _p(nd4), _p(hit), _p(tout))
        return hit, tout

    def quad_intersect(self, O, D, t):
        O, D, t = self._rays(O, D, t)
        hit, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.ref_quad_intersect(len(t), _p(O), _p(D), _p(t), _p(hit), _p(tout))
        return hit, tout

    def cube_intersect(self, O, D, t):
        O, D, t = self._rays(O, D, t)
        hit, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.ref_cube_intersect(len(t), _p(O), _p(D), _p(t), _p(hit), _p(tout))
        return hit, tout

    def aabb(self, O, D, t, bmin, bmax):
        O, D, t = self._rays(O, D, t)
        bmin, bmax = np.ascontiguousarray(bmin, np.float32), np.ascontiguousarray(bmax, np.float32)
        n = len(t)
        a, b, c = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        self.L.ref_aabb(n, _p(O), _p(D), _p(t), _p(bmin), _p(bmax), _p(a), _p(b), _p(c))
        return a, b, c

    def bvh_create(self, v9, pr4=None):
        v9 = np.ascontiguousarray(v9, np.float32).reshape(-1, 9)
        pr4 = np.zeros((0, 4), np.float32) if pr4 is None else np.ascontiguousarray(pr4, np.float32).reshape(-1, 4)
        h = C.c_void_p(self.L.ref_bvh_create(_p(v9), len(v9), _p(pr4), len(pr4)))
        info = np.zeros(2, np.int32)
        self.L.ref_bvh_info(h, _p(info))
        nodes = np.zeros((int(info[0]), 8), np.uint32)
        prim = np.zeros(len(v9) + len(pr4), np.int32)
        self.L.ref_bvh_nodes(h, _p(nodes), _p(prim))
        return h, nodes, prim, int(info[1])

    def bvh_trace(self, h, mode, O, D, t):
        O, D, t = self._rays(O, D, t)
        idx, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.ref_bvh_trace(h, mode, len(t), _p(O), _p(D), _p(t), _p(idx), _p(tout))
        return idx, tout

    def bvh_occluded(self, h, mode, O, D, t):
        O, D, t = self._rays(O, D, t)
        occ = np.zeros(len(t), np.int32)
        self.L.ref_bvh_occluded(h, mode, len(t), _p(O), _p(D), _p(t), _p(occ))
        return occ

    def bvh_destroy(self, h):
        self.L.ref_bvh_destroy(h)

    def random_point_on_light(self, s0):
        out, s1 = np.zeros(10, np.float32), C.c_uint()
        self.L.ref_random_point_on_light(C.c_uint(s0), _p(out), C.byref(s1))
        return out, s1.value

    def diffuse_reflection(self, s0, N):
        N = np.ascontiguousarray(N, np.float32)
        out, s1 = np.zeros(3, np.float32), C.c_uint()
        self.L.ref_diffuse_reflection(C.c_uint(s0), _p(N), _p(out), C.byref(s1))
        return out, s1.value
