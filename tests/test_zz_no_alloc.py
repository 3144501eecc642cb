"""v4r_render_noalloc -- the render entry point that allocates nothing (its queues live in a static device array of
the library) -- against v4r_render, through the C ABI: bit-identical accumulators on two calls in a row, the second
with a smaller input, and a clear error for an image that does not fit the scratch."""
import numpy as np
import pytest

from v4ray_hip import Renderer, V4RayError, scenes

pytestmark = pytest.mark.gpu


def _pair(name, w, h, **kw):
    s1 = scenes.load(name, w, h, **kw)
    s2 = scenes.load(name, w, h, **kw)
    return (Renderer(s1.renderer_param, s1.camera_param, s1, seed=0xA110C),
            Renderer(s2.renderer_param, s2.camera_param, s2, seed=0xA110C))


@pytest.mark.parametrize("name,size,first,second", [
    ("c4", (192, 108), 40, 7),     # 79 488 triangles: deep tree, stack spill slab in the scratch; 20 736 paths per pass
    ("c2", (256, 144), 24, 5),     # metal / dielectric / Lambertian classes, no lights
    ("c1", (160, 160), 20, 3),     # constant medium (fog) + isotropic class
])
def test_noalloc_render_matches_render_twice(name, size, first, second):
    ref, na = _pair(name, *size, pretransform_triangles=(name == "c4"))
    # first call: `first` passes (several batches of the scratch); second call, right after: fewer passes, continuing the job
    ref.render_passes(first)
    na.render_passes_noalloc(first)
    a, n_a = ref.read_accumulator()
    b, n_b = na.read_accumulator()
    assert n_a == n_b == first
    assert np.array_equal(a, b), f"{name}: first call differs on {np.mean(a != b):.4f} of the values"
    ref.render_passes(second, first_pass=first)
    na.render_passes_noalloc(second, first_pass=first)
    a, n_a = ref.read_accumulator()
    b, n_b = na.read_accumulator()
    assert n_a == n_b == first + second
    assert np.array_equal(a, b), f"{name}: second (smaller) call differs on {np.mean(a != b):.4f} of the values"
    assert np.isfinite(b).all() and b.mean() > 0
    # and a strided job from a cleared accumulator, as a rank of a sample-split job would run it
    ref.clear()
    na.clear()
    ref.render_passes(6, first_pass=1, pass_stride=3)
    na.render_passes_noalloc(6, first_pass=1, pass_stride=3)
    assert np.array_equal(ref.read_accumulator()[0], na.read_accumulator()[0])
    ref.close()
    na.close()


def test_noalloc_interleaves_with_the_allocating_entry_point():
    """Both entry points on one handle, alternating: the accumulator is the one image of passes 0..29."""
    ref, mix = _pair("c3", 128, 128)
    ref.render_passes(30)
    mix.render_passes_noalloc(10)
    mix.render_passes(10, first_pass=10)
    mix.render_passes_noalloc(10, first_pass=20)
    a, n_a = ref.read_accumulator()
    b, n_b = mix.read_accumulator()
    assert n_a == n_b == 30
    assert np.array_equal(a, b)  # k_resolve continues the running per-pixel sum in ascending pass order, whatever the batch sizes
    ref.close()
    mix.close()


def test_noalloc_rejects_an_image_beyond_the_scratch():
    s = scenes.load("c3", 1024, 1024)   # 1 048 576 paths per pass: more than the scratch holds
    r = Renderer(s.renderer_param, s.camera_param, s)
    with pytest.raises(V4RayError, match="exceeds the .* path slots"):
        r.render_passes_noalloc(1)
    r.render_passes(1)                  # the handle is still usable
    assert r.read_accumulator()[1] == 1
    r.close()
