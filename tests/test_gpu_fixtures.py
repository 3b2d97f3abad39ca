"""The reference-derived tie and adversarial fixtures through the HIP path (tests/golden/bvh_vectors.npz, unit_vectors.npz).

bvh_vectors: `cow`, `ball` (reference assets) and `mix` (3000 random triangles incl. 100 coincident copies -> exact t ties,
40 spheres, a block of rays with a zero direction component).  The scenes are rebuilt here with the product's REFERENCE_SAH
builder, the node array is asserted equal to the reference's dump, and closest-hit / any-hit queries are compared bit for
bit with the reference's IntersectBVH / IntersectBVHShadowRay results (bvh.h:507-530, 640-688) -- with the 4-wide quantised
nodes and with the reference's own boxes, through the parity-hook kernel (hook_kernel 0), through the render path's
persistent walk (k_trace, 1) and through the end-of-render walker that gives a whole wave to one ray (k_tail_wave's code, 2),
whose (t, leaf position) rule, `tmin <= t` box rule and exact-leaf-box confirmation are what decide the ties.

unit_vectors: per-primitive cases (edge-on and axis-aligned triangles, rays with zero components, origins inside spheres,
axis-aligned planes, finite ray.t) against one-primitive scenes: Triangle/Sphere/Plane/Quad/Cube::Intersect on the device
(primitive.h:50-72, 108-127, 154-158, 210-236, 284-295).

Bar: object ids and distance bit patterns equal.
"""
import numpy as np
import pytest

from raytracer_amd import api
from oracle.oracle import Oracle
from conftest import bits

pytestmark = pytest.mark.gpu


def build_fixture_scene(g, name):
    s = api.HostScene()
    if name == "mix":
        for p in g["mix_pr"]:
            s.add_sphere(p[:3], float(p[3]), api.Material())
    s.add_triangles(g[name + "_v9"], api.Material())
    s.finalize(api.REFERENCE_SAH)
    a = s.desc.arrays()
    assert np.array_equal(a["nodes"][:s.info()["nodes_used"]], g[name + "_nodes"]), name  # the tree the reference built
    assert np.array_equal(a["prim_idx"], g[name + "_prim_idx"]), name
    return s


@pytest.mark.parametrize("name", ["cow", "ball", "mix"])
@pytest.mark.parametrize("wide", [1, 0])
@pytest.mark.parametrize("hook_kernel", [0, 1, 2])
def test_reference_bvh_vectors(gold_bvh, name, wide, hook_kernel):
    g = gold_bvh
    s = build_fixture_scene(g, name)
    d = api.Device(256, 256, wide=wide, hook_kernel=hook_kernel)
    d.upload(s.desc)
    O, D = g[name + "_O"], g[name + "_D"]
    n = len(O)
    obj, t = d.trace_rays(O, D, np.full(n, 1e34, np.float32))
    assert np.array_equal(obj, g[name + "_idx_rec"])
    assert np.array_equal(bits(t), bits(g[name + "_t_rec"]))
    occ, _ = d.trace_rays(O, D, g[name + "_tshadow"], any_hit=True)
    assert np.array_equal(occ, g[name + "_occ"])
    assert (obj >= 0).mean() > 0.1 and 0.02 < occ.mean() < 0.98
    if name == "mix":
        # the coincident copies really tie: rays that hit triangle k < 100 would hit its copy 100 + k at the same distance
        hit_dup = np.isin(obj - len(g["mix_pr"]), np.arange(0, 200))
        assert hit_dup.sum() > 100 and (D[:1500] == 0).any(axis=1).all()
    d.close()


def test_leaf_larger_than_64_primitives():
    """the reference's builder stops splitting when no split lowers the SAH cost (bvh.h:344): 200 coincident triangles stay
    one leaf, which its traversal walks like any other; the device splits it into synthetic pairs with the leaf's own box"""
    rng = np.random.default_rng(11)
    base = rng.uniform(-1, 1, (300, 1, 3))
    v9 = (base + rng.normal(scale=0.08, size=(300, 3, 3))).astype(np.float32).reshape(300, 9)
    big = np.repeat(v9[:1], 200, axis=0)
    v9 = np.concatenate([v9[:150], big, v9[150:]])
    s = api.HostScene()
    s.add_triangles(v9, api.Material())
    s.finalize(api.REFERENCE_SAH)
    nodes = s.desc.arrays()["nodes"][:s.info()["nodes_used"]].view(np.int32)
    assert nodes[2:, 7].max() == 201
    orc = Oracle(s.desc)
    n = 20000
    O = (rng.normal(size=(n, 3)) * 2).astype(np.float32)
    tgt = v9.reshape(-1, 3, 3).mean(1)[rng.integers(0, len(v9), n)] + rng.normal(scale=0.05, size=(n, 3))
    D = tgt - O
    D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    tmax = np.full(n, 1e34, np.float32)
    c_obj, c_t = orc.find_nearest(O, D, tmax)
    ts = (c_t * 0.999).astype(np.float32)
    c_occ = orc.is_occluded(O, D, ts)
    copies = np.r_[0, 150:350]  # triangle 0 and its 200 copies: one leaf of 201
    assert np.isin(c_obj, copies).sum() > 50 and len(np.unique(c_obj[np.isin(c_obj, copies)])) == 1  # the first in leaf order wins every tie
    for wide in (1, 0):
        for hook_kernel in (0, 1, 2):
            d = api.Device(128, 128, wide=wide, hook_kernel=hook_kernel)
            d.upload(s.desc)
            obj, t = d.trace_rays(O, D, tmax)
            assert np.array_equal(obj, c_obj) and np.array_equal(bits(t), bits(c_t)), (wide, hook_kernel)
            assert np.array_equal(d.trace_rays(O, D, ts, any_hit=True)[0], c_occ), (wide, hook_kernel)
            d.close()


def one_by_one(d, scenes, O, D, tin):
    """scene k against ray k: (obj, t) per case"""
    obj, t = np.zeros(len(scenes), np.int32), np.zeros(len(scenes), np.float32)
    for k, s in enumerate(scenes):
        d.upload(s.desc)
        o, tt = d.trace_rays(O[k:k + 1], D[k:k + 1], tin[k:k + 1])
        obj[k], t[k] = o[0], tt[0]
    return obj, t


def test_triangle_and_sphere_unit_vectors(gold_unit):
    """one-primitive BVHs: the root leaf's box test (bvh.h:443-452, incl. 0 * inf = NaN on flat boxes) stands between the
    ray and the primitive, so the device result is compared with the oracle on the same one-primitive scene, and every
    device hit with the reference's Triangle/Sphere::Intersect vector"""
    g = gold_unit
    d = api.Device(64, 64)
    sel = np.r_[0:2000:2, 2000:12000:20]  # the adversarial block (axis-aligned triangles, zero components, finite t) + a sample
    v9, O, D, tin = g["tri_v9"][sel], g["tri_O"][sel], g["tri_D"][sel], g["tri_tin"][sel]
    scenes, want_obj, want_t = [], np.zeros(len(sel), np.int32), np.zeros(len(sel), np.float32)
    for k in range(len(sel)):
        s = api.HostScene()
        s.add_triangles(v9[k:k + 1], api.Material())
        s.finalize(api.REFERENCE_SAH)
        scenes.append(s)
        o, t = Oracle(s.desc).find_nearest(O[k:k + 1], D[k:k + 1], tin[k:k + 1])
        want_obj[k], want_t[k] = o[0], t[0]
    obj, t = one_by_one(d, scenes, O, D, tin)
    assert np.array_equal(obj, want_obj) and np.array_equal(bits(t), bits(want_t))
    hit = obj >= 0
    assert hit.sum() > 250
    assert (g["tri_hit"][sel][hit] >= 0).all() and np.array_equal(bits(t[hit]), bits(g["tri_t"][sel][hit]))
    ref_hit = g["tri_hit"][sel] >= 0
    assert (ref_hit & ~hit).mean() < 0.1  # the few that the root box test rejects (edge-on / NaN slabs)
    sel = np.r_[0:1000:2, 3000:12000:36]  # origins inside the sphere first
    pr, O, D, tin = g["sph_pr"][sel], g["sph_O"][sel], g["sph_D"][sel], g["sph_tin"][sel]
    scenes, want_obj, want_t = [], np.zeros(len(sel), np.int32), np.zeros(len(sel), np.float32)
    for k in range(len(sel)):
        s = api.HostScene()
        s.add_sphere(pr[k, :3], float(pr[k, 3]), api.Material())
        s.finalize(api.REFERENCE_SAH)
        scenes.append(s)
        o, t = Oracle(s.desc).find_nearest(O[k:k + 1], D[k:k + 1], tin[k:k + 1])
        want_obj[k], want_t[k] = o[0], t[0]
    obj, t = one_by_one(d, scenes, O, D, tin)
    assert np.array_equal(obj, want_obj) and np.array_equal(bits(t), bits(want_t))
    hit = obj >= 0
    assert hit.sum() > 150
    assert (g["sph_hit"][sel][hit] >= 0).all() and np.array_equal(bits(t[hit]), bits(g["sph_t"][sel][hit]))
    d.close()


def test_plane_quad_cube_unit_vectors(gold_unit, gold_small):
    """analytic primitives have no box in the way: the device must reproduce the reference's vectors case by case"""
    g = gold_unit
    d = api.Device(64, 64)
    sel = np.r_[0:600:2, 600:12000:30]  # axis-aligned normals / directions first
    nd, O, D, tin = g["pl_nd"][sel], g["pl_O"][sel], g["pl_D"][sel], g["pl_tin"][sel]
    scenes = []
    for k in range(len(sel)):
        s = api.HostScene()
        s.add_plane(nd[k, :3], float(nd[k, 3]), api.Material())
        scenes.append(s.finalize())
    obj, t = one_by_one(d, scenes, O, D, tin)
    assert np.array_equal(obj >= 0, g["pl_hit"][sel] >= 0) and np.array_equal(bits(t), bits(g["pl_t"][sel]))
    assert (obj >= 0).sum() > 100
    # the light quad and the cube of the default scene, alone in a scene (same constructor arguments as scene.h:51-66)
    quad = api.HostScene()
    quad.add_quad(3.0, gold_small["quad_T"], api.Material(api.LIGHT, emission=(2, 2, 2)))
    quad.finalize()
    d.upload(quad.desc)
    obj, t = d.trace_rays(g["quad_O"], g["quad_D"], g["quad_tin"])
    assert np.array_equal(obj >= 0, g["quad_hit"] >= 0) and np.array_equal(bits(t), bits(g["quad_t"])) and (obj >= 0).sum() > 500
    cube = api.HostScene()
    cube.add_cube((0, 0, 0), (0.75, 0.75, 0.75), gold_small["scene_cube_M"], api.Material())
    cube.finalize()
    a = cube.desc.arrays()
    assert np.array_equal(bits(a["cube_inv_M"]), bits(gold_small["scene_cube_inv_M"])) and np.array_equal(bits(a["cube_b"]), bits(gold_small["scene_cube_b"]))
    d.upload(cube.desc)
    obj, t = d.trace_rays(g["cube_O"], g["cube_D"], g["cube_tin"])
    assert np.array_equal(obj >= 0, g["cube_hit"] >= 0) and np.array_equal(bits(t), bits(g["cube_t"])) and (obj >= 0).sum() > 500
    d.close()


def test_per_object_hooks_reproduce_the_reference_vectors(default_scene, gold_unit, gold_small):
    """rtmi_intersect_primitive / rtmi_hit_attributes (what compat/primitive.h and compat/scene.h call): every one of the reference's
    Triangle / Sphere / Plane / Quad / Cube::Intersect vectors with NO box in the way, GetNormal / GetAlbedo / material on 60 000 hits"""
    g, d = gold_unit, api.Device(64, 64)
    step = 40  # one call per primitive: every 40th of the 12 000 random ones, all of the adversarial head
    for kind, key, data in ((0, "tri", g["tri_v9"]), (1, "sph", g["sph_pr"]), (2, "pl", g["pl_nd"])):
        sel = np.r_[0:1000, 1000:12000:step]
        hits = 0
        for k in sel:
            if kind == 1:
                dat = np.r_[data[k, :3], np.float32(data[k, 3]) * np.float32(data[k, 3])].astype(np.float32)  # r2 = r * r (primitive.h:105)
            else:
                dat = data[k]
            hit, t = d.intersect_primitive(kind, dat, g[key + "_O"][k:k + 1], g[key + "_D"][k:k + 1], g[key + "_tin"][k:k + 1])
            assert bool(hit[0]) == bool(g[key + "_hit"][k] >= 0) and bits(t)[0] == bits(g[key + "_t"][k:k + 1])[0], (key, k)
            hits += int(hit[0])
        assert hits > 100, (key, hits)
    quad = np.r_[gold_small["scene_quad_inv_T"][0, :12], gold_small["scene_quad_size"][0]].astype(np.float32)
    hit, t = d.intersect_primitive(3, quad, g["quad_O"], g["quad_D"], g["quad_tin"])
    assert np.array_equal(hit != 0, g["quad_hit"] >= 0) and np.array_equal(bits(t), bits(g["quad_t"])) and hit.sum() > 500
    cube = np.r_[gold_small["scene_cube_inv_M"][0, :12], gold_small["scene_cube_b"][0, 0, :3], gold_small["scene_cube_b"][0, 1, :3]].astype(np.float32)
    hit, t = d.intersect_primitive(4, cube, g["cube_O"], g["cube_D"], g["cube_tin"])
    assert np.array_equal(hit != 0, g["cube_hit"] >= 0) and np.array_equal(bits(t), bits(g["cube_t"])) and hit.sum() > 500
    L = api.device_lib()
    assert L.rtmi_intersect_primitive(d.h, 5, 0, 0, 0, 0, 0, 0, 0) == -1
    assert L.rtmi_hit_attributes(d.h, 1, *[api._ptr(np.zeros(3, np.float32))] * 6) == -3  # no scene yet
    d.upload(default_scene.desc)
    gs = gold_small
    N, alb, mat = d.hit_attributes(gs["att_obj"].astype(np.int32), gs["att_I"], gs["att_wo"])
    assert np.array_equal(bits(N), bits(gs["att_N"])) and np.array_equal(bits(alb), bits(gs["att_albedo"])) and np.array_equal(mat, gs["att_mat"].astype(np.int32))
    assert L.rtmi_hit_attributes(d.h, 1, api._ptr(np.array([99999], np.int32)), *[api._ptr(np.zeros(3, np.float32))] * 5) == -1
    d.close()


def test_config0_whitted_1280x720(default_scene, gold_720):
    """BASELINE.json configs[0] at its own size: the default scene, Whitted primary + shadow rays, 1280x720, 1 spp"""
    W, H = 1280, 720
    d = api.Device(W, H)
    d.upload(default_scene.desc)
    cam = api.default_camera(W, H)
    d.render(api.RAYTRACING, 0, 1, 1, 5, 0)
    img = d.accumulator()
    orc = Oracle(default_scene.desc)
    ref = orc.trace(W, H, cam, api.RAYTRACING, 5)
    e = float(np.sqrt(np.mean((img[..., :3].astype(np.float64) - ref[..., :3]) ** 2)))
    assert e <= 1e-5, e  # bar 1e-3; expf in the glass absorption is the only operation that is not correctly rounded on both sides
    assert np.isfinite(img).all()
    px = d.finalize(0)
    assert (px != orc.finalize(ref, 0).reshape(H, W)).mean() < 1e-4
    d.close()


def test_options_api():
    d = api.Device(32, 32)
    L = api.device_lib()
    assert L.rtmi_set_option(d.h, b"no_such_option", 1) == -1 and b"unknown option" in L.rtmi_last_error(d.h)
    assert L.rtmi_set_option(d.h, b"hook_kernel", 3) == -1 and L.rtmi_set_option(d.h, b"slots", 99) == -1
    assert L.rtmi_set_option(d.h, b"lanes", 3) == -1 and L.rtmi_set_option(d.h, b"veteran_rounds", 0) == -1  # (an option of round 2, removed)
    assert d.get_option("lanes") == 1  # one lane unless the caller asks for two
    for key, v in (("wide", 0), ("overlap", 0), ("batch", 3), ("slots", 6), ("carry_cap", 4096), ("left_cap", 64), ("hook_kernel", 1), ("plane_pairs", 0),
                   ("carry_min", 0), ("lanes", 2), ("preview_reads", 1), ("wave_below", 64), ("tail_paths", 1000), ("last_round_bounces", 128)):
        d.set_option(key, v)
        assert d.get_option(key) == v
    d.use_own_stream()
    d.set_stream(0)  # HIP's null stream
    room = api.HostScene().make_room().finalize()
    d.upload(room.desc)
    d.set_seeds(api.seed_table(32 * 32))
    d.render(api.PATHTRACING, 0, 5, 1)
    acc = d.accumulator()
    ref = Oracle(room.desc).sample(32, 32, api.default_camera(32, 32), api.seed_table(32 * 32), 0, 5)
    assert float(np.sqrt(np.mean((acc.astype(np.float64) - ref) ** 2))) <= 5e-5
    d.close()


@pytest.mark.parametrize("hook_kernel", [0, 1])
def test_plane_pairs_equal_the_plane_loop(default_scene, hook_kernel):
    """The reference's room is three pairs of opposite axis planes; the device tests each pair with one division
    (rt_device.h isect_plane_pair).  Same hits, same distance bits as the generic plane loop (option plane_pairs = 0)
    and as the oracle's Plane::Intersect loop (primitive.h:154-158, bvh.h:498-502), for origins inside the room, exactly
    on a wall, outside it, and for rays with zero direction components."""
    rng = np.random.default_rng(11)
    n = 200000
    O = np.stack([rng.uniform(-3.4, 3.4, n), rng.uniform(-1.4, 2.4, n), rng.uniform(-3.4, 4.4, n)], 1).astype(np.float32)
    D = rng.normal(size=(n, 3)).astype(np.float32)
    D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    O[:3000, 0] = np.float32(-3.0)   # on the plane (1,0,0), d = 3: a = 0
    O[3000:6000, 1] = np.float32(2.0)  # on (0,-1,0), d = 2
    O[6000:9000, 2] = np.float32(0.0)
    D[9000:12000, 0] = 0
    D[12000:15000, 1] = 0
    D[15000:16000, 1:] = 0
    D[16000:17000, 2] = np.float32(-0.0)
    tmax = np.full(n, 1e34, np.float32)
    tmax[100000:] = rng.uniform(0.01, 6.0, n - 100000).astype(np.float32)
    orc = Oracle(default_scene.desc)
    c_obj, c_t = orc.find_nearest(O, D, tmax)
    c_occ = orc.is_occluded(O, D, tmax)
    outside = (np.abs(O[:, 0]) > 3) | (O[:, 1] < -1) | (O[:, 1] > 2)
    assert 0.2 < outside.mean() < 0.8 and ((c_obj >= 4) & (c_obj < 10)).mean() > 0.3  # both branches are exercised; planes are objects 4..9
    res = []
    for pairs in (1, 0):
        d = api.Device(64, 64, plane_pairs=pairs, hook_kernel=hook_kernel)
        d.upload(default_scene.desc)
        obj, t = d.trace_rays(O, D, tmax)
        occ, _ = d.trace_rays(O, D, tmax, any_hit=True)
        assert np.array_equal(obj, c_obj) and np.array_equal(bits(t), bits(c_t)), pairs
        assert np.array_equal(occ, c_occ), pairs
        res.append((obj, t, occ))
        d.close()
    assert all(np.array_equal(bits(a), bits(b)) for a, b in zip(res[0], res[1]))
    # shadow rays that end AT the surface they would hit, and one or a few ulp before and beyond it: the any-hit tests of the kernel that
    # creates a shadow ray leave out divisions whose quotient cannot be below the ray's end (rt_device.h: quotient_reaches) -- the
    # boundary between "occluded" and "not" must stay where the division puts it.  Planes, the light quad, the cube, through the
    # render path's begin_any (hook kernel 1) and the plain walk (0)
    inside = ~outside & (c_obj >= 0) & (c_t < 100)
    Oi, Di, ti = O[inside][:60000], D[inside][:60000], c_t[inside][:60000]
    d = api.Device(64, 64, hook_kernel=hook_kernel)
    d.upload(default_scene.desc)
    for steps in (0, 1, -1, 2, -2, 5, -5, 64, -64):
        ends = ti.copy()
        for _ in range(abs(steps)):
            ends = np.nextafter(ends, np.float32(np.inf if steps > 0 else 0))
        want = orc.is_occluded(Oi, Di, ends)
        assert np.array_equal(d.trace_rays(Oi, Di, ends, any_hit=True)[0], want), steps
        if steps in (0, 1):
            assert (0.0 if steps == 0 else 0.9) <= want.mean() <= (0.1 if steps == 0 else 1.0)  # `t < ray.t` is strict: a ray that ends at the hit is free
    d.close()


def qbvh_oracle(desc):
    o = Oracle(desc)
    assert o.set_qbvh(True)
    return o


@pytest.mark.parametrize("hook_kernel", [0, 1, 2])
def test_reference_default_traversal_option(default_scene, gold_720, gold_bvh, hook_kernel):
    """Option qbvh: results follow the reference's DEFAULT traversal (Scene::useQBVH = true, scene.h:585; IntersectQBVH /
    IntersectQBVHShadowRay, bvh.h:579-758) instead of its binary one.  With the option on, the primary hit map at 1280x720 equals
    the reference-default map on ALL 921 600 rays (the 207 rays of column 640 included, golden qbvh_diff_*); the reference-held
    cow / ball / mix vectors equal their *_qbvh results; fresh rays with zero direction components on box planes equal the
    oracle's QBVH restatement (itself bit-exact against the compiled reference: tests/test_oracle_vs_ref.py); a path-traced image
    matches the oracle's QBVH image.  Through every walker of the library (hook kernels 0, 1, 2)."""
    W, H = 1280, 720
    d = api.Device(W, H, qbvh=1, hook_kernel=hook_kernel)
    d.upload(default_scene.desc)
    d.set_camera(gold_720["camera"])
    if hook_kernel == 0:
        obj, t = d.trace_primary()
        q_obj, q_t = gold_720["hit_obj_bvh"].astype(np.int32).copy(), gold_720["hit_t_bvh"].copy()
        idx = gold_720["qbvh_diff_idx"]
        q_obj[idx], q_t[idx] = gold_720["qbvh_diff_obj"], gold_720["qbvh_diff_t"]
        assert np.array_equal(obj, q_obj) and np.array_equal(bits(t), bits(q_t))
    orc = qbvh_oracle(default_scene.desc)
    rng = np.random.default_rng(23)
    n = 60000
    O = np.stack([rng.uniform(-2.9, 2.9, n), rng.uniform(-0.95, 1.95, n), rng.uniform(-2.9, 3.9, n)], 1).astype(np.float32)
    O[:6000] = np.round(O[:6000] * 4) / 4
    D = rng.normal(size=(n, 3)).astype(np.float32)
    D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    D[:2000, 0] = 0
    D[2000:4000, 1] = 0
    D[4000:6000, 2] = 0
    D[6000:6500, :2] = 0
    tmax = np.full(n, 1e34, np.float32)
    c_obj, c_t = orc.find_nearest(O, D, tmax)
    g_obj, g_t = d.trace_rays(O, D, tmax)
    assert np.array_equal(g_obj, c_obj) and np.array_equal(bits(g_t), bits(c_t))
    b_obj, _ = Oracle(default_scene.desc).find_nearest(O, D, tmax)
    assert (b_obj != c_obj).sum() > 0  # rays on which the reference's two traversals differ are in the set
    ts = (np.minimum(c_t, 50) * rng.uniform(0.2, 1.5, n)).astype(np.float32)
    assert np.array_equal(d.trace_rays(O, D, ts, any_hit=True)[0], orc.is_occluded(O, D, ts))
    d.close()
    # the reference-held tie vectors (coincident triangles, zero components): QBVH visiting order decides ties
    for name in ("cow", "ball", "mix"):
        s = build_fixture_scene(gold_bvh, name)
        d = api.Device(64, 64, qbvh=1, hook_kernel=hook_kernel)
        d.upload(s.desc)
        Of, Df = gold_bvh[name + "_O"], gold_bvh[name + "_D"]
        obj, t = d.trace_rays(Of, Df, np.full(len(Of), 1e34, np.float32))
        assert np.array_equal(obj, gold_bvh[name + "_idx_qbvh"]) and np.array_equal(bits(t), bits(gold_bvh[name + "_t_qbvh"])), name
        if name + "_occ_qbvh" in gold_bvh.files:
            assert np.array_equal(d.trace_rays(Of, Df, gold_bvh[name + "_tshadow"], any_hit=True)[0], gold_bvh[name + "_occ_qbvh"]), name
        d.close()


def test_reference_default_traversal_images(default_scene):
    """option qbvh through the renderer: path-traced and Whitted images at 320x180 against the oracle walking the reference's QBVH
    (the centre column's rays have D.x = 0: SURVEY A.6), and the leaf order it imposes leaves the default contract alone."""
    W, H = 320, 180
    cam, seeds = api.default_camera(W, H), api.seed_table(W * H)
    orc = qbvh_oracle(default_scene.desc)
    d = api.Device(W, H, qbvh=1)
    d.upload(default_scene.desc)
    d.set_seeds(seeds)
    d.set_camera(cam)
    d.render(api.PATHTRACING, 0, 4, 1)
    ref = orc.sample(W, H, cam, seeds, 0, 4)
    assert float(np.sqrt(np.mean((d.accumulator()[..., :3].astype(np.float64) - ref[..., :3]) ** 2))) <= 4e-5
    d.render(api.RAYTRACING, 0, 1, 1, 5, 0)
    wref = orc.trace(W, H, cam, api.RAYTRACING, 5)
    assert float(np.sqrt(np.mean((d.accumulator()[..., :3].astype(np.float64) - wref[..., :3]) ** 2))) <= 1e-5
    obj, t = d.trace_primary()
    o_obj, o_t = orc.primary_hits(W, H, cam)
    assert np.array_equal(obj, o_obj) and np.array_equal(bits(t), bits(o_t))
    d.close()
