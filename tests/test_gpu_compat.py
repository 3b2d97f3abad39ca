"""The source-level drop-in on the GPU: tests/compat/integration_tu.cpp holds INTEGRATION.md sections 1-3 verbatim (checked in
test_abi_and_host.py), compiled against compat/ -- the reference's include names, namespace Tmpl8, its member names -- and linked
with librtmi.so.  Here its Renderer::Init / Tick run the way the template's main loop would run them, and the frames it produces
are compared bit for bit with the library's Python path; Scene::FindNearest / IsOccluded / GetNormal / GetAlbedo / GetMaterial and
<Primitive>::Intersect, called one ray at a time as the reference's CPU loops call them, are compared with the reference's golden
vectors (both traversals: scene.useQBVH on = the reference's default, off = IntersectBVH) and the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracer_amd import api, build
from oracle.oracle import Oracle
from conftest import bits

pytestmark = pytest.mark.gpu
W, H = 1280, 720


def P(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def tu():
    os.environ["RTMI_ASSETS"] = api.ASSETS
    api.device_lib(), api.host_lib()
    if not os.path.exists(build.COMPAT_TU_LIB):
        raise RuntimeError("tests/compat/libcompat_tu.so is not built: run __graft_entry__.build()")
    L = C.CDLL(build.COMPAT_TU_LIB)
    L.tu_seeds.restype = C.POINTER(C.c_uint32)
    L.tu_tick.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.tu_set_light.argtypes = [C.c_float] * 4
    L.tu_scene_queries.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 9
    L.tu_brute_force.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert L.tu_init() == 0 and (L.tu_width(), L.tu_height()) == (W, H)
    return L


@pytest.fixture(scope="module")
def dev(default_scene):
    d = api.Device(W, H)
    d.upload(default_scene.desc)
    d.set_seeds(api.seed_table(W * H))
    d.set_camera(api.default_camera(W, H))
    yield d
    d.close()


def test_init_and_ticks_equal_the_python_path(tu, dev, default_scene):
    assert np.array_equal(np.ctypeslib.as_array(tu.tu_seeds(), (W * H,)), api.seed_table(W * H))  # renderer.cpp:51-55 through compat's RandomUInt
    px = np.zeros((H, W), np.uint32)
    assert tu.tu_tick(api.RAYTRACING, 1, P(px)) == 0  # the reference starts in the Whitted mode (renderer.h:38)
    dev.render(api.RAYTRACING, 0, 1, 1, 5, 0)
    assert np.array_equal(px, dev.finalize(0)) and len(np.unique(px)) > 1000
    assert tu.tu_tick(api.PATHTRACING, 2, P(px)) == 2  # SetVisualizationMode clears, two frames accumulate
    dev.reset()
    dev.render(api.PATHTRACING, 0, 1, 1, 5, 0)  # the same calls as the two ticks: float sums depend on where the reads fall
    dev.finalize(1)
    dev.render(api.PATHTRACING, 1, 1, 1, 5, 0)
    want = dev.finalize(2)
    print("pixels that differ:", int((px != want).sum()))
    assert np.array_equal(px, want) and len(np.unique(px)) > 1000
    # a light setter between two ticks reaches the device without an upload (scene.h:536-551, template.cpp:350-465)
    tu.tu_set_light(0.5, 0.25, -0.5, 40.0)
    assert tu.tu_tick(api.RAYTRACING, 1, P(px)) == 2
    moved = api.HostScene.default()
    moved.set_lights(40.0, pos=(0.5, 0.25, -0.5))
    a = moved.desc
    dev.set_whitted_lights(np.ctypeslib.as_array(a.whitted_light_pos).reshape(8), np.ctypeslib.as_array(a.whitted_light_color).reshape(8), a.whitted_intensity)
    dev.render(api.RAYTRACING, 0, 1, 1, 5, 0)
    want = dev.finalize(0)
    assert np.array_equal(px, want)
    dev.set_whitted_lights(np.ctypeslib.as_array(default_scene.desc.whitted_light_pos).reshape(8),
                           np.ctypeslib.as_array(default_scene.desc.whitted_light_color).reshape(8), default_scene.desc.whitted_intensity)
    dev.render(api.RAYTRACING, 0, 1, 1, 5, 0)
    assert (dev.finalize(0) != want).mean() > 0.05  # and the setter did change the picture


@pytest.mark.parametrize("use_qbvh", [1, 0])
def test_scene_queries_one_ray_at_a_time(tu, default_scene, gold_small, use_qbvh):
    g, n = gold_small, 1500
    O, D, ts = (np.ascontiguousarray(g[k][:n]) for k in ("inc_O", "inc_D", "inc_tshadow"))
    obj, occ, mat = (np.zeros(n, np.int32) for _ in range(3))
    t, N, alb = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    tu.tu_scene_queries(n, use_qbvh, P(O), P(D), P(ts), P(obj), P(t), P(occ), P(N), P(alb), P(mat))
    sfx = "_qbvh" if use_qbvh else ""
    assert np.array_equal(obj, g["inc_obj" + sfx][:n].astype(np.int32)) and np.array_equal(bits(t), bits(g["inc_t" + sfx][:n]))
    assert np.array_equal(occ, g["inc_occ" + sfx][:n].astype(np.int32))
    hit = obj >= 0
    assert hit.sum() > 1000 and 0.02 < occ.mean() < 0.98
    I = O[hit] + t[hit, None] * D[hit]  # Ray::IntersectionPoint, primitive.h:24
    oN, oalb, omat = Oracle(default_scene.desc).hit_attribs(obj[hit], I, D[hit])
    assert np.array_equal(bits(N[hit]), bits(oN)) and np.array_equal(bits(alb[hit]), bits(oalb)) and np.array_equal(mat[hit], omat)
    assert not N[~hit].any() and not alb[~hit].any() and not mat[~hit].any()  # objIdx == -1: float3( 0 ), Material()
    assert len(np.unique(mat[hit])) >= 4


def test_primitive_intersect_loop_equals_the_tree(tu, default_scene):
    """every object of the scene tested with its own Intersect( ray ), no BVH: the brute-force loop the reference had before its
    BVH finds what the tree finds"""
    cam = api.default_camera(W, H)
    orc = Oracle(default_scene.desc)
    Oa, Da, _ = orc.primary_rays(W, H, cam)
    # (not column 640: there D.x == 0, the tree's slab test meets 0 * inf and culls boxes the ray does pass through -- SURVEY A.6)
    for x, y in ((641, 100), (300, 360), (600, 400), (680, 380), (1000, 380), (200, 650), (700, 30)):
        obj, t = C.c_int(-2), C.c_float(0)
        tu.tu_brute_force(x, y, 1 << 30, C.byref(obj), C.byref(t))
        k = y * W + x
        o, tt = orc.find_nearest(Oa.reshape(-1, 3)[k:k + 1], Da.reshape(-1, 3)[k:k + 1], np.full(1, 1e34, np.float32))
        assert obj.value == int(o[0]) and bits(np.float32(t.value).reshape(1))[0] == bits(tt)[0], (x, y, obj.value, o[0])
