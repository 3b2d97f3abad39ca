"""A GPU that has little memory to give.  The queues of full-size batches take 19 GB per slot at 1920x1080 (4 slots per lane, two
lanes when asked for): when they do not fit, rtmi_render first gives the second lane up, then halves the batches until they fit,
finishing whatever is in flight or carried with the queues it has (drain_and_free in rtmi.hip).  The image does not depend on
batch size or lanes; finishing the carried paths early folds their exact sums into the float accumulator one time more than
on the empty GPU, as a read between two renders does: same image up to that rounding."""
import numpy as np
import pytest

from raytracer_amd import api

pytestmark = pytest.mark.gpu
W, H, SPP = 1920, 1080, 64
GB = 1 << 30


def _render(capfd=None, **opts):
    scene = api.HostScene.default()
    d = api.Device(W, H, **opts)
    d.upload(scene.desc)
    d.set_seeds(api.seed_table(W * H))
    d.set_camera(api.default_camera(W, H))
    d.render(api.PATHTRACING, 0, 8, 1)  # small batches first, whose late paths are carried into the call that has to resize
    d.render(api.PATHTRACING, 8, SPP - 8, 1)
    img = d.accumulator().copy()
    lanes = d.get_option("lanes")
    d.close()
    return img, lanes


def test_batches_shrink_and_the_second_lane_goes_when_memory_is_short(capfd):
    import torch
    want, lanes = _render(lanes=2)
    assert lanes == 2
    free, total = torch.cuda.mem_get_info()
    if free < 200 * GB:
        pytest.skip("needs a GPU with 200 GB free to compare against")
    # room for one lane's queues (4 x 19 GB + carry buffers) but not for two
    hog = torch.empty(free - 110 * GB, dtype=torch.uint8, device="cuda")
    got, lanes = _render(lanes=2)
    err = capfd.readouterr().err
    assert "rendering on one lane" in err and lanes == 1
    assert np.abs(want[..., :3]).max() > 0 and np.allclose(got, want, rtol=2e-6, atol=1e-6)
    del hog
    torch.cuda.empty_cache()
    # room for neither: batches of 16 samples instead of 32
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(free - 50 * GB, dtype=torch.uint8, device="cuda")
    got, _ = _render(lanes=1)
    err = capfd.readouterr().err
    assert "trying batches of 16" in err
    assert np.abs(want[..., :3]).max() > 0 and np.allclose(got, want, rtol=2e-6, atol=1e-6)
    del hog
    torch.cuda.empty_cache()
