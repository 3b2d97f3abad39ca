"""GPU parity at the full sizes of BASELINE.json's other configs (configs[2] is covered by test_gpu_mesh.py):

  configs[1]  one OBJ mesh of ~100k triangles through the OBJ loader, Whitted primary + shadow rays, 1920x1080, 1 spp
  configs[3]  the ~1M-triangle path-traced scene at 3840x2160, sample-split over 2/4/8 "ranks"
  configs[4]  a ~10M-triangle scene at 3840x2160 followed by the postprocessing.h pass (gamma, chroma, vignette)

The reference cannot build these scenes (O(n^2) builder), so the checker is the CPU oracle on the same
bytes (oracle/, pinned bit-exact against the compiled reference on the scenes the reference can build): primary hit
ids and distances bit-exact on every pixel, images on a row subset the oracle finishes in seconds (RMSE bar 1e-3,
asserted at 1e-5), plus size-independent properties (sample split == single render, determinism).
"""
import numpy as np
import pytest

from raytracer_amd import api
from oracle.oracle import Oracle
from conftest import bits

pytestmark = pytest.mark.gpu


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def test_config1_obj_mesh_whitted_full_hd(tmp_path):
    W, H = 1920, 1080
    L = api.host_lib()
    nu, nv = 1626, 31  # isotropic tessellation of the (2,3) knot: 1626 * 31 * 2 = 100 812 triangles
    n = nu * nv * 2
    k = float(np.ceil((n * 4e-4 / (api.KNOT_LENGTH * api.KNOT_CIRCUMFERENCE)) ** 0.5))  # see api.mesh_scale: one knot here
    v9 = np.zeros((n, 9), np.float32)
    assert L.rth_torus_knot(nu, nv, 0.42 * k, 0.17 * k, 0.075 * k, 2, 3, 0.0, 0.0, 1.5 * k, v9.ctypes.data) == n
    path = str(tmp_path / "knot100k.obj")
    assert L.rth_write_obj(path.encode(), v9.ctypes.data, n) == 0
    s = api.HostScene().make_room(k)
    s.load_logo()
    s.load_obj(path, api.Material(api.DIFFUSE, (0.93, 0.21, 0.21)))
    s.finalize(api.BINNED_SAH)
    assert s.info()["triangles"] == n
    cam = api.default_camera(W, H, k)
    orc = Oracle(s.desc)
    d = api.Device(W, H)
    d.upload(s.desc)
    d.set_camera(cam)
    obj, t = d.trace_primary()
    o_obj, o_t = orc.primary_hits(W, H, cam)
    assert np.array_equal(obj, o_obj) and np.array_equal(bits(t), bits(o_t))
    assert (obj >= 7).mean() > 0.02  # the mesh is in view
    # Whitted: primary + shadow rays to the two point lights (renderer.cpp:133-153, scene.h:490-515), 1 spp
    d.render(api.RAYTRACING, 0, 1, 1, 5, 0)
    img = d.accumulator()
    ref = orc.trace(W, H, cam, api.RAYTRACING, 5)
    assert rmse(img[..., :3], ref[..., :3]) <= 1e-5
    assert (d.finalize(0) != orc.finalize(ref, 0).reshape(H, W)).mean() < 1e-4  # RGB8 of two images that agree to 1e-5
    d.close()


@pytest.fixture(scope="module")
def scene_1m():
    return api.mesh_scene(1000000, api.BINNED_SAH)


def test_config3_4k_sample_split(scene_1m):
    W, H = 3840, 2160
    scene = scene_1m
    cam, seeds = api.default_camera(W, H, scene.scale), api.seed_table(W * H)
    orc = Oracle(scene.desc)
    d = api.Device(W, H)
    d.upload(scene.desc)
    d.set_seeds(seeds)
    d.set_camera(cam)
    obj, t = d.trace_primary()
    o_obj, o_t = orc.primary_hits(W, H, cam)
    assert np.array_equal(obj, o_obj) and np.array_equal(bits(t), bits(o_t))
    # 8 samples in one go
    d.render(api.PATHTRACING, 0, 8, 1)
    whole = d.accumulator()
    # the oracle on every 16th row, same 8 samples
    ref = orc.sample_rows(W, H, cam, seeds, 0, 8, 1, 3, 16)
    assert rmse(whole[3::16, :, :3], ref[3::16, :, :3]) <= 8e-5  # bar 1e-3 * 8 on the un-normalised sum
    # the same 8 samples split by index over G "ranks" (rank r renders r, r+G, ...), accumulators summed as the reduce does
    for G in (2, 8):
        total = np.zeros_like(whole, dtype=np.float64)
        for r in range(G):
            d.reset()
            d.render(api.PATHTRACING, r, 8 // G, G)
            total += d.accumulator()
        assert np.allclose(total, whole, rtol=4e-6, atol=1e-6), G
    d.close()


def full_spp_split(scene, W, H, SPP, G):
    """the config's whole sample count: one render of SPP samples against the same samples split by index over G ranks (rank r renders
    r, r + G, ...: raytracer_amd/distributed.py) and summed as the reduce sums them.  No oracle at this size (10^9..10^10 paths): the
    properties are that nothing is lost or dropped, that the split image is the whole image up to float summation order, and that the
    picture is a picture"""
    d = api.Device(W, H)
    d.upload(scene.desc)
    d.set_seeds(api.seed_table(W * H))
    d.set_camera(api.default_camera(W, H, scene.scale))
    d.reset_stats()
    d.render(api.PATHTRACING, 0, SPP, 1)
    whole = d.accumulator()
    st = d.stats()
    assert st["primary_samples"] == SPP * W * H and st["dropped_paths"] == 0
    assert np.isfinite(whole).all() and (whole[..., :3] >= 0).all()
    mean = whole[..., :3].mean() / SPP
    assert 0.05 < mean < 5.0, mean
    total = np.zeros(whole.shape, np.float64)
    for r in range(G):
        d.reset()
        d.render(api.PATHTRACING, r, SPP // G, G)
        total += d.accumulator()
    assert d.stats()["dropped_paths"] == 0
    # float32 accumulation of SPP terms in two different orders: each running sum rounds once per sample, at most 2^-24 of the sum
    # each time, so the two orders differ by less than SPP * 2^-24 of the pixel's sum (half of that is still far above what is seen)
    err = np.abs(total - whole)
    rtol = max(1e-5, 3e-8 * SPP)
    assert (err <= rtol * np.abs(whole) + 1e-4).all(), float((err / (np.abs(whole) + 1e-4)).max())
    px = d.finalize(SPP // G)  # (the last rank's share, as a picture)
    assert len(np.unique(px)) > 10000
    d.close()


def test_config3_full_sample_count_split_over_8_ranks(scene_1m):
    """BASELINE.json configs[3] at its own size: 3840x2160, 256 spp, sample-split over 8 ranks (2.1 G paths, twice)"""
    full_spp_split(scene_1m, 3840, 2160, 256, 8)


def test_config4_10m_triangles_4k_with_postprocessing():
    W, H = 3840, 2160
    scene = api.mesh_scene(10000000, api.BINNED_SAH)
    info = scene.info()
    assert info["triangles"] > 9500000
    cam, seeds = api.default_camera(W, H, scene.scale), api.seed_table(W * H)
    orc = Oracle(scene.desc)
    d = api.Device(W, H)
    d.upload(scene.desc)
    d.set_seeds(seeds)
    d.set_camera(cam)
    obj, t = d.trace_primary()
    o_obj, o_t = orc.primary_hits(W, H, cam)
    assert np.array_equal(obj, o_obj) and np.array_equal(bits(t), bits(o_t))
    assert (obj >= 7).mean() > 0.03
    d.render(api.PATHTRACING, 0, 2, 1)
    acc = d.accumulator()
    ref = orc.sample_rows(W, H, cam, seeds, 0, 2, 1, 5, 32)
    assert rmse(acc[5::32, :, :3], ref[5::32, :, :3]) <= 2e-5
    # determinism at this size
    d.reset()
    d.render(api.PATHTRACING, 0, 2, 1)
    assert np.array_equal(bits(d.accumulator()), bits(acc))
    # finalize + the post-processing pass in app order (template.cpp:310-317): gamma, chromatic aberration, vignette
    fx = api.PostFx(1, 1, 1, 2.2, 5.0, 0.8)
    px = d.finalize(2, fx)
    want = orc.finalize(acc, 2).reshape(H, W)
    want = orc.postfx(orc.postfx(orc.postfx(want, 0, 2.2), 1, 5.0), 2, 0.8)
    mism = np.argwhere(px != want)
    assert all(tuple(m) == (H // 2, W // 2) for m in mism)  # centre pixel: 0/0 in the reference
    d.close()


def test_config4_full_sample_count_split_over_8_ranks():
    """BASELINE.json configs[4] at its own size: ~10 M triangles, 3840x2160, 1024 spp, 8-rank sample split (8.5 G paths, twice)"""
    full_spp_split(api.mesh_scene(10000000, api.BINNED_SAH), 3840, 2160, 1024, 8)
