"""Cost-ordered dispatch of the tile kernels (lv_group_order_prepare / k_group_order, linevis_amd/csrc/lv_render.hip): the
64x64-pixel groups of a launch start heaviest-of-the-previous-frame first.  A scheduling matter only -- the reference leaves
the order of its ray-generation workgroups to the Vulkan driver -- so the tests pin the one thing that must hold: whatever the
order, the frame (and the PPLL fragment multiset) is the same, and the order is a permutation that follows the measured cost."""
import numpy as np
import pytest

from common import small_case

RTAO = dict(ambient_occlusion_mode="RTAO (Screen Space)", ambient_occlusion_strength=1.0, ambient_occlusion_iterations=2,
            ambient_occlusion_samples_per_frame=4, depth_cue_strength=0.6)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [11, 2])
def test_every_dispatch_order_renders_the_same_frame(hip_lib, mode):
    c = small_case(width=400, height=264, n_lines=60, pts_per_line=40, line_width=0.012, **RTAO)
    ctx = c.hip_context()
    ctx.set_option("dispatch_order", "as_numbered")
    want = ctx.render(mode)
    order, cost = ctx.dispatch_order()
    assert len(order) == 0
    ctx.set_option("dispatch_order", "cost")
    first = ctx.render(mode)              # no history yet: tile-list order
    order0, cost0 = ctx.dispatch_order()
    n = 7 * 5                             # 64 x 64 groups of 400 x 264
    assert len(order0) == n and np.array_equal(order0, np.arange(n)) and cost0.max() > 0
    second = ctx.render(mode)             # ordered by what the first frame measured
    order1, cost1 = ctx.dispatch_order()
    assert sorted(order1.tolist()) == list(range(n))
    assert np.all(np.diff(cost0[order1].astype(np.int64)) <= 0)          # descending cost, ...
    ties = np.diff(cost0[order1].astype(np.int64)) == 0
    assert np.all(np.diff(order1.astype(np.int64))[ties] > 0)            # ... ties in group order
    assert np.array_equal(first, want) and np.array_equal(second, want)
    # a sub-rectangle is another launch geometry: its order starts afresh and the pixels are the same
    assert np.array_equal(ctx.render(mode, tile=(37, 21, 300, 200)), want[21:221, 37:337])
    o2, _ = ctx.dispatch_order()
    assert np.array_equal(o2, np.arange(len(o2)))
    assert np.array_equal(ctx.render(mode), want)


@pytest.mark.gpu
def test_dispatch_order_with_dilated_ao_tiles_and_the_whole_viewport_pass(hip_lib):
    # EAW halo / SVGF: the RTAO pass has a launch geometry of its own (second order array)
    for extra in (dict(ambient_occlusion_denoiser="Edge-Avoiding À-Trous Wavelet Transform", eaw_denoiser_iterations=2),
                  dict(ambient_occlusion_denoiser="SVGF")):
        c = small_case(width=320, height=200, n_lines=40, pts_per_line=40, line_width=0.012, **dict(RTAO, **extra))
        frames = {}
        for order in ("as_numbered", "cost"):
            ctx = c.hip_context()
            ctx.set_option("dispatch_order", order)
            frames[order] = [ctx.render(11) for _ in range(3)]
        for a, b in zip(frames["as_numbered"], frames["cost"]):
            assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [11, 2])
def test_moving_camera_keeps_both_orders_identical(hip_lib, mode):
    """A camera path: every frame is dispatched in the order the PREVIOUS view measured (a stale predictor by construction) and must
    still equal the frame of the tile-list order, byte for byte; the order itself keeps changing along the path."""
    from linevis_amd import camera
    c = small_case(width=400, height=264, n_lines=60, pts_per_line=40, line_width=0.012, transparent=(mode == 2), **RTAO)
    a, b = c.hip_context(), c.hip_context()
    a.set_option("dispatch_order", "as_numbered")
    b.set_option("dispatch_order", "cost")
    orders = []
    for k in range(6):
        pos = (0.25 * np.sin(0.5 * k), 0.1 * k - 0.2, 0.8 - 0.05 * k)
        view, proj, fovy, near, far = camera.default_camera(c.width, c.height, pos)
        for ctx in (a, b):
            ctx.set_camera(view, proj, fovy, near, far, c.width, c.height)
        assert np.array_equal(a.render(mode), b.render(mode)), k
        orders.append(b.dispatch_order()[0].copy())
    assert any(not np.array_equal(orders[i], orders[i + 1]) for i in range(1, 5))
