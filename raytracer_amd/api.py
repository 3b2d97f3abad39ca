"""ctypes bindings used by the Python harness (tests/, bench.py, __graft_entry__.py).

Two libraries, both built in-tree by raytracer_amd/build.py:
  librtmi.so       the C ABI of include/rtmi.h (HIP kernels).  `Device` wraps one rtmi_ctx.
  librtmi_host.so  the C++ host mirror of the reference API (Scene, BVH, Camera, Renderer).

There is no Python or CPU implementation of the trace loop in this package: every render/trace call goes
through librtmi.so and raises `RtmiError` if the library or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
ASSETS = os.path.join(ROOT, "assets")

# rtmi_mode / rtmi_material (include/rtmi.h)
NORMAL, DISTANCE, ALBEDO, RAYTRACING, PATHTRACING = range(5)
DIFFUSE, MIRROR, GLASS, SUBSTANCE, LIGHT = range(5)
REFERENCE_SAH, BINNED_SAH = 0, 1
NOALLOC_MAX_RAYS = 1 << 18  # RTMI_NOALLOC_MAX_RAYS
# GLFW key codes the reference's Renderer::KeyDown / KeyUp handle (renderer.cpp:368-441)
KEY_A, KEY_D, KEY_E, KEY_Q, KEY_S, KEY_W, KEY_RIGHT, KEY_LEFT, KEY_DOWN, KEY_UP = 65, 68, 69, 81, 83, 87, 262, 263, 264, 265


class RtmiError(RuntimeError):
    pass


class BvhNode(C.Structure):
    _fields_ = [("aabbMinx", C.c_float), ("aabbMiny", C.c_float), ("aabbMinz", C.c_float),
                ("aabbMaxx", C.c_float), ("aabbMaxy", C.c_float), ("aabbMaxz", C.c_float),
                ("leftFirst", C.c_int32), ("primitiveCount", C.c_int32)]


FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int32)
UP = C.POINTER(C.c_uint32)


class SceneDesc(C.Structure):
    _fields_ = [("n_quads", C.c_int32), ("n_spheres", C.c_int32), ("n_cubes", C.c_int32), ("n_planes", C.c_int32), ("n_triangles", C.c_int32),
                ("quad_inv_T", FP), ("quad_size", FP), ("sphere_info", FP), ("sphere_invr", FP),
                ("cube_M", FP), ("cube_inv_M", FP), ("cube_b", FP), ("tri_verts", FP),
                ("primitives", FP), ("albedo", FP), ("material_type", IP), ("ior", FP), ("absorption", FP),
                ("nodes", C.POINTER(BvhNode)), ("n_nodes", C.c_int32), ("prim_idx", IP),
                ("light_corners", C.c_float * 12), ("light_A", C.c_float), ("light_s", C.c_float), ("light_emission", C.c_float * 4),
                ("texture", UP), ("tex_w", C.c_int32), ("tex_h", C.c_int32),
                ("whitted_light_pos", (C.c_float * 4) * 2), ("whitted_light_color", (C.c_float * 4) * 2), ("whitted_intensity", C.c_float)]

    @property
    def total(self):
        return self.n_quads + self.n_spheres + self.n_cubes + self.n_planes + self.n_triangles

    def arrays(self):
        """numpy views (no copies) of every array, keyed by field name."""
        def view(ptr, shape, dtype):
            n = int(np.prod(shape))
            if n == 0 or not ptr:
                return np.zeros(shape, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_byte)), (n * np.dtype(dtype).itemsize,)).view(dtype).reshape(shape)
        t, nb = self.total, self.n_spheres + self.n_triangles
        return {
            "quad_inv_T": view(self.quad_inv_T, (self.n_quads, 16), np.float32), "quad_size": view(self.quad_size, (self.n_quads,), np.float32),
            "sphere_info": view(self.sphere_info, (self.n_spheres, 4), np.float32), "sphere_invr": view(self.sphere_invr, (self.n_spheres,), np.float32),
            "cube_M": view(self.cube_M, (self.n_cubes, 16), np.float32), "cube_inv_M": view(self.cube_inv_M, (self.n_cubes, 16), np.float32),
            "cube_b": view(self.cube_b, (self.n_cubes, 2, 4), np.float32), "tri_verts": view(self.tri_verts, (self.n_triangles, 3, 4), np.float32),
            "primitives": view(self.primitives, (t, 4), np.float32), "albedo": view(self.albedo, (t, 4), np.float32),
            "material_type": view(self.material_type, (t,), np.int32), "ior": view(self.ior, (t,), np.float32), "absorption": view(self.absorption, (t, 4), np.float32),
            "nodes": view(self.nodes, (self.n_nodes, 8), np.uint32), "prim_idx": view(self.prim_idx, (nb,), np.int32),
            "texture": view(self.texture, (self.tex_h, self.tex_w), np.uint32),
            "light_corners": np.array(self.light_corners, np.float32).reshape(3, 4),
            "light": np.array([self.light_A, self.light_s] + list(self.light_emission), np.float32),
            "whitted_light_pos": np.array([list(r) for r in self.whitted_light_pos], np.float32),
            "whitted_light_color": np.array([list(r) for r in self.whitted_light_color], np.float32),
            "whitted_intensity": np.float32(self.whitted_intensity),
        }


class PostFx(C.Structure):
    _fields_ = [("gamma_on", C.c_int32), ("chroma_on", C.c_int32), ("vignette_on", C.c_int32),
                ("gamma", C.c_float), ("chroma_intensity", C.c_float), ("vignette_intensity", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("primary_samples", C.c_uint64), ("extend_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("extend_pairs", C.c_uint64), ("extend_prims", C.c_uint64), ("shadow_pairs", C.c_uint64), ("shadow_prims", C.c_uint64),
                ("bounce_launches", C.c_uint64),
                ("ms_extend", C.c_double), ("ms_shade", C.c_double), ("ms_connect", C.c_double), ("ms_other", C.c_double),
                ("n_extend_launches", C.c_uint64), ("n_shade_launches", C.c_uint64), ("n_connect_launches", C.c_uint64),
                ("extend_walkers", C.c_uint64), ("shadow_walkers", C.c_uint64),
                ("suspended", C.c_uint64), ("max_extend_steps", C.c_uint64), ("max_shadow_steps", C.c_uint64),
                ("ms_walk", C.c_double), ("n_walk_windows", C.c_uint64),
                ("ms_drain", C.c_double), ("n_drains", C.c_uint64), ("ms_extend_union", C.c_double), ("ms_walk_union", C.c_double),
                ("ms_shade_union", C.c_double), ("ms_connect_union", C.c_double),
                ("extend_wide_nodes", C.c_uint64), ("extend_confirms", C.c_uint64), ("shadow_wide_nodes", C.c_uint64), ("shadow_confirms", C.c_uint64),
                ("dropped_paths", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Material(C.Structure):
    _fields_ = [("type", C.c_int), ("color", C.c_float * 3), ("ior", C.c_float), ("absorption", C.c_float * 3), ("emission", C.c_float * 3)]

    def __init__(self, type=DIFFUSE, color=(0.93, 0.93, 0.93), ior=1.0, absorption=(0, 0, 0), emission=(0, 0, 0)):
        super().__init__(type, (C.c_float * 3)(*color), ior, (C.c_float * 3)(*absorption), (C.c_float * 3)(*emission))


_dev = None
_host = None


def _ptr(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def device_lib():
    """librtmi.so; raises RtmiError if it was not built."""
    global _dev
    if _dev is None:
        path = os.path.join(_PKG, os.environ.get("RTMI_LIBRARY", "librtmi.so"))  # (experimental builds beside the product library)
        if not os.path.exists(path):
            raise RtmiError("librtmi.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (there is no fallback path)")
        L = C.CDLL(path)
        L.rtmi_last_error.restype = C.c_char_p
        L.rtmi_version.restype = C.c_char_p
        for name in ("rtmi_create", "rtmi_destroy", "rtmi_set_stream", "rtmi_upload_scene", "rtmi_set_camera", "rtmi_set_seeds", "rtmi_reset_accumulator",
                     "rtmi_set_accumulator", "rtmi_render", "rtmi_trace_primary", "rtmi_trace_rays", "rtmi_finalize", "rtmi_postprocess",
                     "rtmi_read_accumulator", "rtmi_synchronize", "rtmi_flush", "rtmi_set_profiling", "rtmi_get_stats", "rtmi_reset_stats",
                     "rtmi_use_own_stream", "rtmi_set_option", "rtmi_get_option", "rtmi_reduce_accumulator", "rtmi_trace_rays_noalloc",
                     "rtmi_set_whitted_lights", "rtmi_memory_info", "rtmi_intersect_primitive", "rtmi_hit_attributes"):
            getattr(L, name).restype = C.c_int
        L.rtmi_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
        L.rtmi_destroy.argtypes = [C.c_void_p]
        L.rtmi_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.rtmi_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.rtmi_set_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.rtmi_set_seeds.argtypes = [C.c_void_p, C.c_void_p]
        L.rtmi_reset_accumulator.argtypes = [C.c_void_p]
        L.rtmi_set_accumulator.argtypes = [C.c_void_p, C.c_void_p]
        L.rtmi_render.argtypes = [C.c_void_p] + [C.c_int] * 6
        L.rtmi_trace_primary.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rtmi_trace_rays.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rtmi_trace_rays_noalloc.argtypes = L.rtmi_trace_rays.argtypes
        L.rtmi_finalize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(PostFx)]
        L.rtmi_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(PostFx)]
        L.rtmi_read_accumulator.argtypes = [C.c_void_p, C.c_void_p]
        L.rtmi_synchronize.argtypes = [C.c_void_p]
        L.rtmi_flush.argtypes = [C.c_void_p]
        L.rtmi_set_profiling.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rtmi_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.rtmi_reset_stats.argtypes = [C.c_void_p]
        L.rtmi_last_error.argtypes = [C.c_void_p]
        L.rtmi_use_own_stream.argtypes = [C.c_void_p]
        L.rtmi_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
        L.rtmi_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]
        L.rtmi_reduce_accumulator.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rtmi_set_whitted_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.rtmi_intersect_primitive.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.rtmi_hit_attributes.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.rtmi_memory_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _dev = L
    return _dev


def host_lib():
    global _host
    if _host is None:
        path = os.path.join(_PKG, "librtmi_host.so")
        if not os.path.exists(path):
            raise RtmiError("librtmi_host.so is missing: run __graft_entry__.build()")
        device_lib()  # librtmi_host.so links librtmi.so
        L = C.CDLL(path)
        L.rth_scene_create.restype = C.c_void_p
        L.rth_scene_destroy.argtypes = [C.c_void_p]
        L.rth_scene_make_default.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.rth_scene_make_room.argtypes = [C.c_void_p, C.c_float]
        L.rth_scene_add_sphere.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Material)]
        L.rth_scene_add_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.POINTER(Material), C.c_float, C.c_float, C.c_float]
        L.rth_scene_add_torus_knot.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(Material)]
        L.rth_scene_add_plane.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Material)]
        L.rth_scene_add_quad.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.POINTER(Material)]
        L.rth_scene_add_cube.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Material)]
        L.rth_scene_load_obj.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Material), C.c_float, C.c_float, C.c_float]
        L.rth_scene_load_obj_mtl.argtypes = L.rth_scene_load_obj.argtypes
        L.rth_scene_load_texture_raw.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.rth_scene_finalize.argtypes = [C.c_void_p, C.c_int]
        L.rth_scene_build_qbvh.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rth_scene_set_lights.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rth_scene_desc.argtypes = [C.c_void_p]
        L.rth_scene_desc.restype = C.POINTER(SceneDesc)
        L.rth_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        L.rth_quad_T.argtypes = [C.c_void_p, C.c_void_p]
        L.rth_fill_seeds.argtypes = [C.c_void_p, C.c_longlong]
        L.rth_camera_default.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.rth_camera_primary_rays.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rth_bvh_build.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rth_triangle_derived.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rth_torus_knot.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.rth_torus_knot.restype = C.c_longlong
        L.rth_write_obj.argtypes = [C.c_char_p, C.c_void_p, C.c_longlong]
        L.rth_renderer_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.rth_renderer_create.restype = C.c_void_p
        for n in ("rth_renderer_destroy", "rth_renderer_init", "rth_renderer_upload_scene"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.rth_renderer_scene.argtypes = [C.c_void_p]
        L.rth_renderer_scene.restype = C.c_void_p
        L.rth_renderer_tick.argtypes = [C.c_void_p, C.c_float]
        for n in ("rth_renderer_set_mode", "rth_renderer_set_aa", "rth_renderer_set_depth", "rth_renderer_set_spp_per_tick"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_int]
        L.rth_renderer_set_postfx.argtypes = [C.c_void_p, C.POINTER(PostFx), C.c_int]
        L.rth_renderer_key.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rth_camera_script.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rth_renderer_set_fov.argtypes = [C.c_void_p, C.c_int]
        L.rth_renderer_set_aspect.argtypes = [C.c_void_p, C.c_float]
        L.rth_renderer_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.rth_renderer_frames.argtypes = [C.c_void_p]
        L.rth_renderer_frames.restype = C.c_uint
        L.rth_renderer_pixels.argtypes = [C.c_void_p]
        L.rth_renderer_pixels.restype = UP
        L.rth_renderer_accumulator.argtypes = [C.c_void_p]
        L.rth_renderer_accumulator.restype = FP
        L.rth_renderer_ctx.argtypes = [C.c_void_p]
        L.rth_renderer_ctx.restype = C.c_void_p
        L.rth_renderer_error.argtypes = [C.c_void_p]
        L.rth_renderer_error.restype = C.c_char_p
        L.rth_renderer_last_ms.argtypes = [C.c_void_p]
        L.rth_renderer_last_ms.restype = C.c_float
        _host = L
    return _host


# ---------------------------------------------------------------------------------------------------------------
class HostScene:
    """rtmi::Scene (raytracer_amd/host/scene.h): primitives + host-built BVH + marshalled upload arrays."""

    scale = 1.0

    def __init__(self, handle=None, owner=True):
        self.L = host_lib()
        self.h = handle if handle is not None else self.L.rth_scene_create()
        self.owner = owner and handle is None

    def __del__(self):
        if getattr(self, "owner", False) and self.h:
            self.L.rth_scene_destroy(self.h)
            self.h = None

    @classmethod
    def default(cls, builder=REFERENCE_SAH, assets=ASSETS):
        s = cls()
        if s.L.rth_scene_make_default(s.h, assets.encode(), builder) != 0:
            raise RtmiError("default scene: assets not found in %s" % assets)
        return s

    def make_room(self, scale=1.0):
        self.L.rth_scene_make_room(self.h, float(scale))
        return self

    def add_sphere(self, pos, r, mat):
        self.L.rth_scene_add_sphere(self.h, pos[0], pos[1], pos[2], r, C.byref(mat))

    def add_plane(self, normal, d, mat):
        self.L.rth_scene_add_plane(self.h, normal[0], normal[1], normal[2], d, C.byref(mat))

    def add_quad(self, size, T16, mat):
        T16 = np.ascontiguousarray(T16, np.float32).reshape(16)
        self.L.rth_scene_add_quad(self.h, size, _ptr(T16), C.byref(mat))

    def add_cube(self, pos, size, M16, mat):
        pos, size, M16 = (np.ascontiguousarray(x, np.float32) for x in (pos, size, M16))
        self.L.rth_scene_add_cube(self.h, _ptr(pos), _ptr(size), _ptr(M16.reshape(16)), C.byref(mat))

    def add_triangles(self, v9, mat, offset=(0, 0, 0)):
        v9 = np.ascontiguousarray(v9, np.float32).reshape(-1, 9)
        self.L.rth_scene_add_triangles(self.h, _ptr(v9), len(v9), C.byref(mat), *[float(x) for x in offset])

    def add_torus_knot(self, nu, nv, centre, mat, R=0.42, r=0.17, tube=0.075, p=2, q=3):
        self.L.rth_scene_add_torus_knot(self.h, nu, nv, R, r, tube, p, q, centre[0], centre[1], centre[2], C.byref(mat))

    def load_obj(self, path, mat, offset=(0, 0, 0)):
        if self.L.rth_scene_load_obj(self.h, path.encode(), C.byref(mat), *[float(x) for x in offset]) != 0:
            raise RtmiError("cannot load " + path)

    def load_obj_with_materials(self, path, fallback, offset=(0, 0, 0)):
        """LoadObject with the file's usemtl / mtllib materials instead of one Material per file"""
        if self.L.rth_scene_load_obj_mtl(self.h, path.encode(), C.byref(fallback), *[float(x) for x in offset]) != 0:
            raise RtmiError("cannot load " + path)

    def load_logo(self, assets=ASSETS):
        if self.L.rth_scene_load_texture_raw(self.h, os.path.join(assets, "logo_128x64.rgb32").encode(), 128, 64) != 0:
            raise RtmiError("logo texture missing")

    def finalize(self, builder=REFERENCE_SAH):
        self.L.rth_scene_finalize(self.h, builder)
        return self

    def build_qbvh(self):
        """BVH::BuildQBVH (bvh.h:213-249): the reference's 4-wide nodes as an (n, 32) uint32 array (6 x 4 floats of box planes, child[4], count[4])"""
        n = self.L.rth_scene_build_qbvh(self.h, None, 0)
        out = np.zeros((n, 32), np.uint32)
        if n:
            self.L.rth_scene_build_qbvh(self.h, _ptr(out), n)
        return out

    def set_lights(self, intensity, pos=None, color=None, roof_color=None):
        """Scene::SetLightIntensity / SetLightPos / SetLightColor / SetRoofLightColor (scene.h:536-551)"""
        a = [None if v is None else np.ascontiguousarray(v, np.float32) for v in (pos, color, roof_color)]
        self.L.rth_scene_set_lights(self.h, float(intensity), *[None if v is None else _ptr(v) for v in a])

    @property
    def desc(self):
        d = self.L.rth_scene_desc(self.h).contents
        d._owner = self  # the descriptor points into this scene: keep it alive as long as the descriptor is
        return d

    def info(self):
        out = np.zeros(7, np.int32)
        self.L.rth_scene_info(self.h, _ptr(out))
        return dict(zip(("quads", "spheres", "cubes", "planes", "triangles", "nodes_used", "max_depth"), out.tolist()))


KNOT_LENGTH, KNOT_CIRCUMFERENCE = 6.2273, 0.47124  # centre line and tube circumference of the default (2,3) knot at scale 1


def mesh_scale(n_tris_target, target_area=4e-4):
    """Unit scale at which the mesh configs are built.  The reference's triangle test rejects |det| < 1e-4
    (primitive.h:59): a triangle of area A is invisible to rays with 2*A*|cos| < 1e-4, so a mesh of N triangles needs
    a total area of about N * 4e-4 to be seen at all.  Three knots have 8.8 m^2 at the reference room's scale; the
    whole scene (room, light, knots, camera) is therefore built in units `scale` times larger.  1 for small meshes."""
    area1 = 3 * KNOT_LENGTH * KNOT_CIRCUMFERENCE
    return float(max(1, int(np.ceil((n_tris_target * target_area / area1) ** 0.5))))


def mesh_scene(n_tris_target, builder=BINNED_SAH, materials=None, scale=None, tube=0.075):
    """The mesh configs of SURVEY 8d: the reference room (6 planes + light quad) with three torus knots at
    x = -1.5, 0, +1.5, z = 1.5 (the monkeys' places, scene.h:80-82): diffuse red, mirror, glass n=1.52, everything
    multiplied by mesh_scale(n) (see there; use default_camera(w, h, scene.scale) with it).  n_tris_target is the
    total triangle count aimed for; the tessellation is isotropic (segments along : around = length : circumference)."""
    k = mesh_scale(n_tris_target) if scale is None else float(scale)
    s = HostScene().make_room(k)
    s.scale = k
    s.load_logo()
    nv = max(3, int(round((n_tris_target / 6.0 / (KNOT_LENGTH / KNOT_CIRCUMFERENCE)) ** 0.5)))
    nu = max(8, int(round(n_tris_target / 6.0 / nv)))
    mats = materials or [Material(DIFFUSE, (0.93, 0.21, 0.21)), Material(MIRROR, (0.93, 0.93, 0.93)), Material(GLASS, (0.93, 0.93, 0.93), 1.52)]
    for m, x in zip(mats, (-1.5, 0.0, 1.5)):
        s.add_torus_knot(nu, nv, (x * k, 0.0, 1.5 * k), m, 0.42 * k, 0.17 * k, tube * k)  # (tube: 0.075 in every config; thicker for the mesh-dominated profile view)
    return s.finalize(builder)


def mesh_scene_obj(n_tris_target, cache_dir, builder=BINNED_SAH, tube=0.075):
    """mesh_scene's scene through the OBJ path the metric names ("path-traced OBJ scene"): each knot is written as a Wavefront OBJ
    file (rtmi::WriteObj, %.9g: binary32 round trip) unless `cache_dir` already holds it, and loaded with Scene::LoadObject
    (scene.h:222-291: one Material per file, triangles in file order).  Same triangles, hence the same BVH and the same image bits,
    as mesh_scene.  Returns (scene, {"obj_write_s", "obj_ingest_s", "bvh_build_s", "obj_bytes"})."""
    import time
    L = host_lib()
    k = mesh_scale(n_tris_target)
    nv = max(3, int(round((n_tris_target / 6.0 / (KNOT_LENGTH / KNOT_CIRCUMFERENCE)) ** 0.5)))
    nu = max(8, int(round(n_tris_target / 6.0 / nv)))
    os.makedirs(cache_dir, exist_ok=True)
    t_write, paths = 0.0, []
    for i, x in enumerate((-1.5, 0.0, 1.5)):
        path = os.path.join(cache_dir, "knot_%dx%d_s%g_t%g_%d.obj" % (nu, nv, k, tube, i))
        if not os.path.exists(path):
            t0 = time.time()
            n = nu * nv * 2
            v9 = np.zeros((n, 9), np.float32)
            if L.rth_torus_knot(nu, nv, 0.42 * k, 0.17 * k, tube * k, 2, 3, x * k, 0.0, 1.5 * k, _ptr(v9)) != n:
                raise RtmiError("torus knot generator")
            tmp = "%s.tmp%d" % (path, os.getpid())  # (several ranks may start at once: write aside, then rename)
            if L.rth_write_obj(tmp.encode(), _ptr(v9), n) != 0:
                raise RtmiError("cannot write " + tmp)
            os.replace(tmp, path)
            t_write += time.time() - t0
        paths.append(path)
    s = HostScene().make_room(k)
    s.scale = k
    s.load_logo()
    mats = [Material(DIFFUSE, (0.93, 0.21, 0.21)), Material(MIRROR, (0.93, 0.93, 0.93)), Material(GLASS, (0.93, 0.93, 0.93), 1.52)]
    t0 = time.time()
    for path, m in zip(paths, mats):
        s.load_obj(path, m)
    t_ingest = time.time() - t0
    t0 = time.time()
    s.finalize(builder)
    return s, {"obj_write_s": t_write, "obj_ingest_s": t_ingest, "bvh_build_s": time.time() - t0, "obj_bytes": sum(os.path.getsize(p) for p in paths)}


def render_frames(devices, frames, total_spp, rank=0, world=1, comm=None, finalize_spp=None):
    """The C++ frame loop (host/frames.h through rth_render_frames): `frames` frames of total_spp samples, this rank's share, frame f
    on devices[f % len(devices)], one host thread per handle; returns the RGB8 frames (rank 0), in order"""
    L = host_lib()
    W, H = devices[0].W, devices[0].H
    hs = (C.c_void_p * len(devices))(*[d.h for d in devices])
    px = np.zeros((frames, H, W), np.uint32)
    L.rth_render_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    mine = (total_spp - rank + world - 1) // world if total_spp > rank else 0
    r = L.rth_render_frames(hs, len(devices), frames, total_spp, rank, world, comm, finalize_spp if finalize_spp is not None else mine, W, H, _ptr(px))
    if r != 0:
        raise RtmiError("rth_render_frames: status %d (%s)" % (r, "; ".join(d.L.rtmi_last_error(d.h).decode() for d in devices)))
    return px


def seed_table(n):
    out = np.zeros(n, np.uint32)
    host_lib().rth_fill_seeds(_ptr(out), n)
    return out


def default_camera(w, h, scale=1.0):
    """camera.h:20-23 as 4 x float4 (pos, topLeft, topRight, bottomLeft); `scale` moves it with a scaled scene"""
    cam = np.zeros(16, np.float32)
    host_lib().rth_camera_default(w, h, _ptr(cam))
    return cam * np.float32(scale)


def zoom_camera(cam16, zoom):
    """The same eye, the screen rectangle shrunk about its centre by `zoom` (< 1: a narrower field of view)."""
    cam = np.array(cam16, np.float32).reshape(4, 4).copy()
    tl, tr, bl = cam[1, :3].copy(), cam[2, :3].copy(), cam[3, :3].copy()
    centre = (tr + bl) * np.float32(0.5)
    for i, c in ((1, tl), (2, tr), (3, bl)):
        cam[i, :3] = centre + (c - centre) * np.float32(zoom)
    return cam.reshape(16)


class Device:
    """One rtmi_ctx (one GPU)."""

    def __init__(self, width, height, device=0, **options):
        self.L = device_lib()
        self.W, self.H = width, height
        h = C.c_void_p()
        r = self.L.rtmi_create(C.byref(h), device, width, height)
        if r != 0:
            raise RtmiError("rtmi_create failed with status %d (no GPU visible?)" % r)
        self.h = h
        self._keep = None
        for k, v in options.items():  # rtmi_set_option per handle, e.g. Device(w, h, wide=0, carry_cap=1000)
            self.set_option(k, v)

    def close(self):
        if self.h:
            self.L.rtmi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r != 0:
            raise RtmiError("rtmi status %d: %s" % (r, self.L.rtmi_last_error(self.h).decode()))

    def upload(self, desc):
        self._chk(self.L.rtmi_upload_scene(self.h, C.byref(desc)))

    def set_camera(self, cam16):
        cam16 = np.ascontiguousarray(cam16, np.float32)
        self._chk(self.L.rtmi_set_camera(self.h, _ptr(cam16)))

    def set_seeds(self, seeds):
        seeds = np.ascontiguousarray(seeds, np.uint32)
        assert seeds.size == self.W * self.H
        self._chk(self.L.rtmi_set_seeds(self.h, _ptr(seeds)))

    def set_stream(self, stream_ptr):
        """run on this HIP stream (a torch stream's .cuda_stream); 0 is HIP's null stream, not "none" (see use_own_stream)"""
        self._chk(self.L.rtmi_set_stream(self.h, C.c_void_p(stream_ptr)))

    def use_own_stream(self):
        self._chk(self.L.rtmi_use_own_stream(self.h))

    def set_option(self, key, value):
        self._chk(self.L.rtmi_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_longlong()
        self._chk(self.L.rtmi_get_option(self.h, key.encode(), C.byref(v)))
        return v.value

    def reduce_accumulator(self, rccl_comm, root=0):
        """one ncclReduce (root >= 0) / ncclAllReduce (root < 0) of the float4 accumulators on the handle's stream"""
        self._chk(self.L.rtmi_reduce_accumulator(self.h, C.c_void_p(rccl_comm), int(root)))

    def set_accumulator(self, dev_ptr):
        self._chk(self.L.rtmi_set_accumulator(self.h, C.c_void_p(dev_ptr)))

    def reset(self):
        self._chk(self.L.rtmi_reset_accumulator(self.h))

    def render(self, mode=PATHTRACING, sample_begin=0, sample_count=1, sample_stride=1, max_depth=5, aa=0):
        self._chk(self.L.rtmi_render(self.h, mode, sample_begin, sample_count, sample_stride, max_depth, aa))

    def sync(self):
        self._chk(self.L.rtmi_synchronize(self.h))

    def flush(self):
        """non-blocking: the handle's stream waits for everything rendered so far (before using a caller-owned accumulator)"""
        self._chk(self.L.rtmi_flush(self.h))

    def trace_primary(self):
        obj = np.zeros(self.W * self.H, np.int32)
        t = np.zeros(self.W * self.H, np.float32)
        self._chk(self.L.rtmi_trace_primary(self.h, _ptr(obj), _ptr(t)))
        return obj, t

    def trace_rays(self, O, D, tmax, any_hit=False):
        O = np.ascontiguousarray(O, np.float32).reshape(-1, 3)
        D = np.ascontiguousarray(D, np.float32).reshape(-1, 3)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = len(O)
        obj = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        self._chk(self.L.rtmi_trace_rays(self.h, n, _ptr(O), _ptr(D), _ptr(tmax), int(any_hit), _ptr(obj), _ptr(t)))
        return obj, t

    def trace_rays_noalloc(self, O, D, tmax, any_hit=False):
        """rtmi_trace_rays_noalloc: no device allocation, at most NOALLOC_MAX_RAYS rays"""
        O = np.ascontiguousarray(O, np.float32).reshape(-1, 3)
        D = np.ascontiguousarray(D, np.float32).reshape(-1, 3)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = len(O)
        obj = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        self._chk(self.L.rtmi_trace_rays_noalloc(self.h, n, _ptr(O), _ptr(D), _ptr(tmax), int(any_hit), _ptr(obj), _ptr(t)))
        return obj, t

    def accumulator(self):
        out = np.zeros((self.H, self.W, 4), np.float32)
        self._chk(self.L.rtmi_read_accumulator(self.h, _ptr(out)))
        return out

    def finalize(self, frames, fx=None):
        out = np.zeros((self.H, self.W), np.uint32)
        self._chk(self.L.rtmi_finalize(self.h, _ptr(out), frames, C.byref(fx) if fx is not None else None))
        return out

    def postprocess(self, pixels, fx):
        px = np.ascontiguousarray(pixels, np.uint32).copy()
        h, w = px.shape
        self._chk(self.L.rtmi_postprocess(self.h, _ptr(px), w, h, C.byref(fx)))
        return px

    def set_whitted_lights(self, pos, color, intensity):
        """the two point lights of the Whitted modes (2 x float4 positions and colours, intensity) without a scene re-upload"""
        pos, color = np.ascontiguousarray(pos, np.float32).reshape(8), np.ascontiguousarray(color, np.float32).reshape(8)
        self._chk(self.L.rtmi_set_whitted_lights(self.h, _ptr(pos), _ptr(color), float(intensity)))

    def intersect_primitive(self, kind, data, O, D, t_in):
        """<Primitive>::Intersect of one primitive against n rays on the device (rtmi_intersect_primitive): (hit, t)"""
        data = np.ascontiguousarray(data, np.float32).reshape(-1)
        O, D, t_in = (np.ascontiguousarray(x, np.float32) for x in (O, D, t_in))
        n = len(t_in)
        hit, t = np.zeros(n, np.int32), np.zeros(n, np.float32)
        self._chk(self.L.rtmi_intersect_primitive(self.h, int(kind), _ptr(data), n, _ptr(O), _ptr(D), _ptr(t_in), _ptr(hit), _ptr(t)))
        return hit, t

    def hit_attributes(self, obj, I, wo):
        """Scene::GetNormal / GetAlbedo / material type of n hits (rtmi_hit_attributes): (N, albedo, material)"""
        obj = np.ascontiguousarray(obj, np.int32)
        I, wo = np.ascontiguousarray(I, np.float32), np.ascontiguousarray(wo, np.float32)
        n = len(obj)
        N, alb, mat = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        self._chk(self.L.rtmi_hit_attributes(self.h, n, _ptr(obj), _ptr(I), _ptr(wo), _ptr(N), _ptr(alb), _ptr(mat)))
        return N, alb, mat

    def memory_info(self):
        """(free, total) bytes of device memory"""
        f, t = C.c_uint64(), C.c_uint64()
        self._chk(self.L.rtmi_memory_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def set_profiling(self, events=False, count_nodes=False):
        self._chk(self.L.rtmi_set_profiling(self.h, int(events), int(count_nodes)))

    def stats(self):
        s = Stats()
        self._chk(self.L.rtmi_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self._chk(self.L.rtmi_reset_stats(self.h))


class Renderer:
    """rtmi::Renderer (raytracer_amd/host/renderer.h): the reference's Init()/Tick() surface."""

    def __init__(self, width=1280, height=720, device=0):
        self.L = host_lib()
        self.W, self.H = width, height
        self.h = self.L.rth_renderer_create(width, height, device)
        self.scene = HostScene(self.L.rth_renderer_scene(self.h), owner=False)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.rth_renderer_destroy(self.h)
            self.h = None

    def init(self):
        if self.L.rth_renderer_init(self.h) != 0:
            raise RtmiError(self.L.rth_renderer_error(self.h).decode())

    def upload_scene(self):
        if self.L.rth_renderer_upload_scene(self.h) != 0:
            raise RtmiError(self.L.rth_renderer_error(self.h).decode())

    def tick(self, dt=0.0):
        self.L.rth_renderer_tick(self.h, dt)
        err = self.L.rth_renderer_error(self.h)
        if err:
            raise RtmiError(err.decode())

    def set_mode(self, mode):
        self.L.rth_renderer_set_mode(self.h, mode)

    def set_aa(self, aa):
        self.L.rth_renderer_set_aa(self.h, int(aa))

    def set_depth(self, d):
        self.L.rth_renderer_set_depth(self.h, d)

    def set_spp_per_tick(self, n):
        self.L.rth_renderer_set_spp_per_tick(self.h, n)

    def set_postfx(self, fx, on=True):
        self.L.rth_renderer_set_postfx(self.h, C.byref(fx), int(on))

    def key(self, key, down):
        """Renderer::KeyDown / KeyUp with a GLFW key code (renderer.cpp:368-441)"""
        self.L.rth_renderer_key(self.h, int(key), int(bool(down)))

    def set_fov(self, fov):
        self.L.rth_renderer_set_fov(self.h, int(fov))

    def set_aspect(self, aspect):
        self.L.rth_renderer_set_aspect(self.h, float(aspect))

    def camera(self):
        cam = np.zeros(16, np.float32)
        self.L.rth_renderer_camera(self.h, _ptr(cam))
        return cam

    @property
    def frames(self):
        return self.L.rth_renderer_frames(self.h)

    def pixels(self):
        return np.ctypeslib.as_array(self.L.rth_renderer_pixels(self.h), (self.H, self.W)).copy()

    def accumulator(self):
        p = self.L.rth_renderer_accumulator(self.h)
        if not p:
            raise RtmiError(self.L.rth_renderer_error(self.h).decode())
        return np.ctypeslib.as_array(p, (self.H, self.W, 4)).copy()

    def last_ms(self):
        return self.L.rth_renderer_last_ms(self.h)
