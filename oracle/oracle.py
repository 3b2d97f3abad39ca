"""oracle/oracle.py -- TEST INFRASTRUCTURE: ctypes wrapper over oracle/liboracle.so (oracle/rt_oracle.cpp).

The CPU restatement of the reference's trace loop.  It is the checker: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this module.  Nothing under raytracer_amd/ does.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(HERE, "liboracle.so"))
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_void_p]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_sample_seed.restype = C.c_uint32
        L.orc_max_threads.restype = C.c_int
        L.orc_camera_script.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(*xs):
    return [np.ascontiguousarray(x, np.float32) for x in xs]


class Oracle:
    """CPU trace loop over the same rtmi_scene_desc bytes the HIP library receives."""

    def __init__(self, desc=None):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create(C.byref(desc))) if desc is not None else None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def set_threads(self, n):
        self.L.orc_set_threads(int(n))

    def max_threads(self):
        return self.L.orc_max_threads()

    def primary_rays(self, W, H, cam):
        cam, = _f32(cam)
        O, D, rD = (np.zeros((W * H, 3), np.float32) for _ in range(3))
        self.L.orc_primary_rays(W, H, _p(cam), _p(O), _p(D), _p(rD))
        return O, D, rD

    def set_qbvh(self, on=True):
        """Scene::SetQBVH: the reference's default 4-wide traversal for every query of this oracle; returns whether it is in use"""
        return bool(self.L.orc_set_qbvh(self.h, int(on)))

    def qbvh_nodes(self):
        return int(self.L.orc_qbvh_nodes(self.h))

    def primary_hits(self, W, H, cam, mode=0):
        cam, = _f32(cam)
        obj, t = np.zeros(W * H, np.int32), np.zeros(W * H, np.float32)
        self.L.orc_primary_hits(self.h, mode, W, H, _p(cam), _p(obj), _p(t))
        return obj, t

    def find_nearest(self, O, D, tmax, mode=0):
        O, D, tmax = _f32(O, D, tmax)
        n = len(tmax)
        obj, t = np.zeros(n, np.int32), np.zeros(n, np.float32)
        self.L.orc_find_nearest(self.h, mode, n, _p(O), _p(D), _p(tmax), _p(obj), _p(t))
        return obj, t

    def is_occluded(self, O, D, tmax):
        O, D, tmax = _f32(O, D, tmax)
        occ = np.zeros(len(tmax), np.int32)
        self.L.orc_is_occluded(self.h, len(tmax), _p(O), _p(D), _p(tmax), _p(occ))
        return occ

    def hit_attribs(self, obj, I, wo):
        obj = np.ascontiguousarray(obj, np.int32)
        I, wo = _f32(I, wo)
        n = len(obj)
        N, alb, mat = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        self.L.orc_hit_attribs(self.h, n, _p(obj), _p(I), _p(wo), _p(N), _p(alb), _p(mat))
        return N, alb, mat

    def sample(self, W, H, cam, seeds0, s_begin, s_count, stride=1, acc=None, counters=None):
        cam, = _f32(cam)
        seeds0 = np.ascontiguousarray(seeds0, np.uint32)
        if acc is None:
            acc = np.zeros((H, W, 4), np.float32)
        self.L.orc_sample(self.h, W, H, _p(cam), _p(seeds0), s_begin, s_count, stride, _p(acc), _p(counters) if counters is not None else None)
        return acc

    def sample_rows(self, W, H, cam, seeds0, s_begin, s_count, stride, row_begin, row_step, acc=None, counters=None):
        """the same loop over rows row_begin, row_begin+row_step, ... only (bounded CPU-baseline sample)"""
        cam, = _f32(cam)
        seeds0 = np.ascontiguousarray(seeds0, np.uint32)
        if acc is None:
            acc = np.zeros((H, W, 4), np.float32)
        self.L.orc_sample_rows(self.h, W, H, _p(cam), _p(seeds0), s_begin, s_count, stride, row_begin, row_step, _p(acc),
                               _p(counters) if counters is not None else None)
        return acc

    def trace(self, W, H, cam, mode, depth=5, aa=False):
        cam, = _f32(cam)
        out = np.zeros((H, W, 4), np.float32)
        self.L.orc_trace(self.h, W, H, _p(cam), mode, depth, int(aa), _p(out))
        return out

    def finalize(self, acc, f):
        acc = np.ascontiguousarray(acc, np.float32).reshape(-1, 4)
        out = np.zeros(len(acc), np.uint32)
        self.L.orc_finalize(len(acc), _p(acc), f, _p(out))
        return out

    def postfx(self, pixels, which, amount):
        px = np.ascontiguousarray(pixels, np.uint32).copy()
        h, w = px.shape
        self.L.orc_postfx(_p(px), w, h, which, C.c_float(amount))
        return px

    # ---- unit level
    def tri_intersect(self, O, D, t, v9):
        O, D, t, v9 = _f32(O, D, t, v9)
        hit, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.orc_tri_intersect(len(t), _p(O), _p(D), _p(t), _p(v9), _p(hit), _p(tout))
        return hit, tout

    def tri_derived(self, v9):
        v9, = _f32(v9)
        v9 = v9.reshape(-1, 9)
        N, c = np.zeros((len(v9), 3), np.float32), np.zeros((len(v9), 3), np.float32)
        self.L.orc_tri_derived(len(v9), _p(v9), _p(N), _p(c))
        return N, c

    def sphere_intersect(self, O, D, t, pr4):
        O, D, t, pr4 = _f32(O, D, t, pr4)
        hit, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.orc_sphere_intersect(len(t), _p(O), _p(D), _p(t), _p(pr4), _p(hit), _p(tout))
        return hit, tout

    def plane_intersect(self, O, D, t, nd4):
        O, D, t, nd4 = _f32(O, D, t, nd4)
        hit, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.orc_plane_intersect(len(t), _p(O), _p(D), _p(t), _p(nd4), _p(hit), _p(tout))
        return hit, tout

    def quad_intersect(self, O, D, t):
        O, D, t = _f32(O, D, t)
        hit, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.orc_quad_intersect(self.h, len(t), _p(O), _p(D), _p(t), _p(hit), _p(tout))
        return hit, tout

    def cube_intersect(self, O, D, t):
        O, D, t = _f32(O, D, t)
        hit, tout = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.orc_cube_intersect(self.h, len(t), _p(O), _p(D), _p(t), _p(hit), _p(tout))
        return hit, tout

    def aabb(self, O, D, t, bmin, bmax):
        O, D, t, bmin, bmax = _f32(O, D, t, bmin, bmax)
        a, b = np.zeros(len(t), np.int32), np.zeros(len(t), np.float32)
        self.L.orc_aabb(len(t), _p(O), _p(D), _p(t), _p(bmin), _p(bmax), _p(a), _p(b))
        return a, b

    def seeds(self, n):
        out = np.zeros(n, np.uint32)
        self.L.orc_seeds(_p(out), n)
        return out

    def rng_floats(self, s0, n):
        out = np.zeros(n, np.float32)
        self.L.orc_rng_floats(C.c_uint32(s0), n, _p(out))
        return out

    def sample_seed(self, s0, s):
        return self.L.orc_sample_seed(C.c_uint32(s0), C.c_uint32(s))

    def light_point(self, s0):
        out, s1 = np.zeros(3, np.float32), C.c_uint32()
        self.L.orc_light_point(self.h, C.c_uint32(s0), _p(out), C.byref(s1))
        return out, s1.value

    def diffuse_reflection(self, s0, N):
        N, = _f32(N)
        out, s1 = np.zeros(3, np.float32), C.c_uint32()
        self.L.orc_diffuse_reflection(self.h, C.c_uint32(s0), _p(N), _p(out), C.byref(s1))
        return out, s1.value

    def bvh_build_reference(self, v9, pr4=None):
        v9, = _f32(v9)
        v9 = v9.reshape(-1, 9)
        pr4 = np.zeros((0, 4), np.float32) if pr4 is None else np.ascontiguousarray(pr4, np.float32).reshape(-1, 4)
        n = len(v9) + len(pr4)
        nodes, prim = np.zeros((2 * n + 1, 8), np.uint32), np.zeros(n, np.int32)
        used = self.L.orc_bvh_build_reference(_p(v9), len(v9), _p(pr4), len(pr4), _p(nodes), _p(prim))
        return nodes[:used], prim

    def camera_script(self, W, H, events):
        """events: list of ("down"|"up", key) | ("update",) | ("fov", degrees) | ("aspect", ratio); returns (cams[n_updates, 16],
        moved[n_updates]) -- Camera::Update by Camera::Update (camera.h:95-159, renderer.cpp:368-441)"""
        kinds = {"down": 0, "up": 1, "update": 2, "fov": 3, "aspect": 4}
        ev = np.zeros((len(events), 2), np.int32)
        val = np.zeros(len(events), np.float32)
        for k, e in enumerate(events):
            ev[k, 0] = kinds[e[0]]
            if e[0] in ("down", "up"):
                ev[k, 1] = e[1]
            elif e[0] in ("fov", "aspect"):
                val[k] = e[1]
        n = sum(1 for e in events if e[0] == "update")
        cams, moved = np.zeros((max(n, 1), 16), np.float32), np.zeros(max(n, 1), np.int32)
        got = self.L.orc_camera_script(W, H, len(events), _p(ev), _p(val), _p(cams), _p(moved))
        assert got == n
        return cams[:n], moved[:n]
