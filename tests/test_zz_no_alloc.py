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


def test_noalloc_calls_of_several_handles_and_threads_do_not_race():
    """Every renderer on a GPU shares the one scratch array.  Calls from different handles -- queued back to back without
    any synchronisation, and made from two threads at once -- run one behind the other on the device (each call's stream
    waits for the event the previous call recorded), so each handle's image is exactly its own v4r_render image."""
    import threading
    refs, nas = [], []
    for name, size in (("c3", (160, 160)), ("c2", (192, 108)), ("c3", (96, 128))):
        ref, na = _pair(name, *size)
        refs.append(ref)
        nas.append(na)
    passes = [24, 20, 32]
    for ref, n in zip(refs, passes):
        ref.render_passes(n)
    # (1) one thread, three handles, three rounds of calls with nothing in between
    for k in range(3):
        for na, n in zip(nas, passes):
            na.render_passes_noalloc(n // 4, first_pass=k * (n // 4))
    # (2) the last quarter of every job from a thread of its own, all at once
    errors = []

    def last_quarter(na, n):
        try:
            na.render_passes_noalloc(n - 3 * (n // 4), first_pass=3 * (n // 4))
        except Exception as e:  # reported below
            errors.append(e)

    threads = [threading.Thread(target=last_quarter, args=(na, n)) for na, n in zip(nas, passes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for ref, na, n in zip(refs, nas, passes):
        a, n_a = ref.read_accumulator()
        b, n_b = na.read_accumulator()
        assert n_a == n_b == n
        assert np.array_equal(a, b), f"differs on {np.mean(a != b):.4f} of the values"
    for r in refs + nas:
        r.close()
