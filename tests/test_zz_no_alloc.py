"""rtmi_trace_rays_noalloc (the entry point that allocates nothing: rays staged in a fixed device array) against
rtmi_trace_rays, through the C ABI: object ids and distance bit patterns equal, on two calls in a row, the second with a
smaller input -- what the first call left in the array must not leak into the second -- and the size limit is an error."""
import numpy as np
import pytest

from raytracer_amd import api
from conftest import bits

pytestmark = pytest.mark.gpu


def test_no_alloc_entry_matches_trace_rays(default_scene, gold_small):
    g = gold_small
    d = api.Device(64, 64)
    d.upload(default_scene.desc)
    O, D = g["inc_O"], g["inc_D"]
    n = len(O)
    far = np.full(n, 1e34, np.float32)
    rng = np.random.default_rng(5)
    small = rng.permutation(n)[:777]  # a different, smaller set for the second call
    for kernel in (0, 2):
        d.set_option("hook_kernel", kernel)
        for any_hit, tm in ((False, far), (True, g["inc_tshadow"])):
            want_obj, want_t = d.trace_rays(O, D, tm, any_hit)
            free0 = d.memory_info()[0]
            got_obj, got_t = d.trace_rays_noalloc(O, D, tm, any_hit)
            assert d.memory_info()[0] == free0, "the call changed the amount of free device memory"
            assert np.array_equal(got_obj, want_obj) and np.array_equal(bits(got_t), bits(want_t)), (kernel, any_hit)
            want_obj, want_t = d.trace_rays(O[small], D[small], tm[small], any_hit)
            got_obj, got_t = d.trace_rays_noalloc(O[small], D[small], tm[small], any_hit)
            assert np.array_equal(got_obj, want_obj) and np.array_equal(bits(got_t), bits(want_t)), (kernel, any_hit, "second, smaller call")
    # against the reference-derived vectors directly, too
    d.set_option("hook_kernel", 0)
    obj, t = d.trace_rays_noalloc(O, D, far)
    assert np.array_equal(obj, g["inc_obj"].astype(np.int32)) and np.array_equal(bits(t), bits(g["inc_t"]))
    assert (obj >= 0).mean() > 0.5
    # the limit: one ray too many is refused with a message, nothing is traced
    L = api.device_lib()
    m = api.NOALLOC_MAX_RAYS + 1
    big = np.zeros((m, 3), np.float32)
    out_i, out_f = np.zeros(m, np.int32), np.zeros(m, np.float32)
    r = L.rtmi_trace_rays_noalloc(d.h, m, big.ctypes.data, big.ctypes.data, out_f.ctypes.data, 0, out_i.ctypes.data, out_f.ctypes.data)
    assert r == -1 and b"at most" in L.rtmi_last_error(d.h)
    # exactly the limit works
    k = api.NOALLOC_MAX_RAYS
    reps = -(-k // n)
    Ob, Db = np.tile(O, (reps, 1))[:k], np.tile(D, (reps, 1))[:k]
    obj, t = d.trace_rays_noalloc(Ob, Db, np.full(k, 1e34, np.float32))
    assert np.array_equal(obj[:n], g["inc_obj"].astype(np.int32)) and np.array_equal(obj[n:2 * n], obj[:n])
    d.close()
    # a second handle on the same GPU finds the same array (one lookup per device)
    d2 = api.Device(32, 32)
    d2.upload(default_scene.desc)
    obj, t = d2.trace_rays_noalloc(O[:100], D[:100], far[:100])
    assert np.array_equal(obj, g["inc_obj"][:100].astype(np.int32))
    d2.close()
