"""dsm_upload_images: the batched host->device hand-over (copy stream + batched makeImages kernels) builds exactly the
pyramids of the oracle's makeImages restatement, for float and u8 pixels, tight and row-strided (cropped) sources."""
import numpy as np
import pytest

from direct_stereo_slam_amd import synth as S
from direct_stereo_slam_amd._lib import DsmError
from oracle import oracle as O

from _scenes import hip_tracker, make_scene

pytestmark = pytest.mark.gpu


def _pyramids_equal(trk, slot, pyr):
    for lvl, ref in enumerate(pyr):
        np.testing.assert_array_equal(trk.get_frame(slot, lvl), ref)


def test_batched_upload_matches_oracle_pyramids(ctx):
    """40 images (more than two copy groups) for 20 trackers, both slots, in one call"""
    sc = make_scene("small", seed=21)
    rng = np.random.default_rng(5)
    trackers = [hip_tracker(ctx, sc) for _ in range(20)]
    imgs = [(sc.new_img + rng.normal(0, 3.0, sc.new_img.shape)).astype(np.float32) for _ in range(40)]
    trks = trackers + trackers
    slots = [0] * 20 + [1] * 20
    ctx.upload_images(trks, slots, imgs, np.linspace(0.5, 2.0, 40))
    for i in (0, 7, 19, 20, 33, 39):
        _pyramids_equal(trks[i], slots[i], O.make_images(imgs[i], sc.nl))


def test_batched_upload_then_track_equals_single_uploads(ctx):
    sc = make_scene("small", seed=22)
    a, b = hip_tracker(ctx, sc), hip_tracker(ctx, sc)
    a.upload_image(0, sc.new_img, 1.0)
    a.upload_image(1, sc.right_img, 1.0)
    ctx.upload_images([b, b], [0, 1], [sc.new_img, sc.right_img])
    ra = a.trackNewestCoarse(S.IDENTITY_POSE, (0.0, 0.0), sc.nl - 1)
    rb = b.trackNewestCoarse(S.IDENTITY_POSE, (0.0, 0.0), sc.nl - 1)
    assert ra[0] == rb[0]
    for x, y in zip(ra[1:], rb[1:]):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
    assert a.optimizeScale(1.0, sc.nl - 1) == b.optimizeScale(1.0, sc.nl - 1)


def test_u8_pixels_equal_their_float_image(ctx):
    sc = make_scene("odd", seed=23)
    rng = np.random.default_rng(9)
    trk = hip_tracker(ctx, sc)
    u8 = [rng.integers(0, 256, (sc.h, sc.w), dtype=np.uint8) for _ in range(2)]
    ctx.upload_images([trk, trk], [0, 1], u8)
    for s in (0, 1):
        _pyramids_equal(trk, s, O.make_images(u8[s].astype(np.float32), sc.nl))


def test_row_pitch_applies_the_calibration_crop(ctx):
    """camera image 318x98, working size 308x92 (crop origin (5, 3)): pointer to the crop origin + the camera's pitch"""
    sc = make_scene("small", seed=24)
    rng = np.random.default_rng(11)
    trk = hip_tracker(ctx, sc)
    for dtype in (np.uint8, np.float32):
        cam = [(rng.random((sc.h + 6, sc.w + 10)) * 255).astype(dtype) for _ in range(2)]
        views = [c[3:3 + sc.h, 5:5 + sc.w] for c in cam]
        assert not views[0].flags["C_CONTIGUOUS"]
        ctx.upload_images([trk, trk], [0, 1], views)
        for s in (0, 1):
            _pyramids_equal(trk, s, O.make_images(np.ascontiguousarray(views[s]).astype(np.float32), sc.nl))


def test_pinned_sources_are_fetched_by_the_gpu(ctx):
    """pinned caller buffers take the kernel-copy path (16-, 4- and 1-byte units); same pyramids, buffers free on return"""
    from direct_stereo_slam_amd.tracker import pinned_array

    sc = make_scene("small", seed=26)
    rng = np.random.default_rng(13)
    trk = hip_tracker(ctx, sc)
    for dtype, pad in ((np.float32, 0), (np.uint8, 0), (np.uint8, 4), (np.uint8, 7), (np.float32, 3)):
        cam = [pinned_array((sc.h + 2, sc.w + pad), dtype) for _ in range(2)]
        want = []
        for c in cam:
            c[...] = (rng.random(c.shape) * 255).astype(dtype)
            want.append(O.make_images(np.ascontiguousarray(c[1:1 + sc.h, pad:pad + sc.w]).astype(np.float32), sc.nl))
        ctx.upload_images([trk, trk], [0, 1], [c[1:1 + sc.h, pad:pad + sc.w] for c in cam])
        for c in cam:
            c[...] = 0  # must not affect the pyramids
        for s in (0, 1):
            _pyramids_equal(trk, s, want[s])


def test_batched_upload_argument_errors(ctx):
    small, tiny = make_scene("small", seed=25), make_scene("tiny", seed=25)
    a, b = hip_tracker(ctx, small), hip_tracker(ctx, tiny)
    with pytest.raises((DsmError, ValueError)):  # different geometries in one call
        ctx.upload_images([a, b], [0, 0], [small.new_img, tiny.new_img])
    with pytest.raises(DsmError):  # the same slot twice
        ctx.upload_images([a, a], [0, 0], [small.new_img, small.new_img])
    with pytest.raises(DsmError):
        ctx.upload_images([a], [4], [small.new_img])
    ctx.upload_images([], [], [])  # empty batch: nothing to do
    # the tracker is still usable
    ctx.upload_images([a], [0], [small.new_img])
    _pyramids_equal(a, 0, small.new_p)


def test_double_buffered_async_handover(ctx):
    """DSM_SLOT_NEXT_* + dsm_upload_images_async + dsm_frames_advance: the images of step i+1 travel while step i is
    tracked from the front buffers; every step's pyramids and results equal the synchronous path's"""
    from direct_stereo_slam_amd.tracker import pinned_array

    sc = make_scene("small", seed=27)
    rng = np.random.default_rng(17)
    n = 6
    sync = [hip_tracker(ctx, sc) for _ in range(n)]
    dbl = [hip_tracker(ctx, sc) for _ in range(n)]
    frames = []
    for step in range(4):
        left = [(sc.new_img + rng.normal(0, 2.0 + step, sc.new_img.shape)).astype(np.float32) for _ in range(n)]
        right = [(sc.right_img + rng.normal(0, 2.0 + step, sc.right_img.shape)).astype(np.float32) for _ in range(n)]
        frames.append((left, right))
    pin = [[pinned_array(sc.new_img.shape) for _ in range(2 * n)] for _ in range(2)]  # two sets, alternating
    poses0 = np.tile(S.IDENTITY_POSE, (n, 1))

    def hand_over(step, asynchronous):
        bufs = pin[step & 1] if step % 2 == 0 else None  # even steps from pinned, odd steps from pageable memory
        left, right = frames[step]
        imgs = left + right
        if bufs is not None:
            for b, im in zip(bufs, imgs):
                b[...] = im
            imgs = bufs
        ctx.upload_images(dbl + dbl, [2] * n + [3] * n, imgs, np.full(2 * n, 1.0 + 0.1 * step), asynchronous=asynchronous)

    hand_over(0, False)
    for step in range(4):
        ctx.advance_frames(dbl + dbl, [0] * n + [1] * n)
        if step + 1 < 4:
            hand_over(step + 1, True)  # travels while this step is tracked
        left, right = frames[step]
        for i, t in enumerate(sync):
            t.upload_image(0, left[i], 1.0 + 0.1 * step)
            t.upload_image(1, right[i], 1.0 + 0.1 * step)
        ra = ctx.track_batch(sync, poses0, np.zeros((n, 2)), sc.nl - 1)
        rb = ctx.track_batch(dbl, poses0, np.zeros((n, 2)), sc.nl - 1)
        for x, y in zip(ra, rb):
            np.testing.assert_array_equal(x, y)
        ea, sa = ctx.optimize_scale_batch(sync, np.ones(n), sc.nl - 1)
        eb, sb = ctx.optimize_scale_batch(dbl, np.ones(n), sc.nl - 1)
        np.testing.assert_array_equal(ea, eb)
        np.testing.assert_array_equal(sa, sb)
        _pyramids_equal(dbl[step % n], 0, O.make_images(left[step % n], sc.nl))
        ctx.upload_wait()
    with pytest.raises(DsmError):  # nothing handed over to the back buffers since the last advance
        ctx.advance_frames([dbl[0]], [0])


def test_back_buffer_upload_leaves_the_front_untouched(ctx):
    sc = make_scene("small", seed=28)
    trk = hip_tracker(ctx, sc)
    trk.upload_image(0, sc.new_img, 1.0)
    other = (sc.new_img[::-1]).copy()
    ctx.upload_images([trk], [2], [other])
    _pyramids_equal(trk, 0, sc.new_p)
    ctx.advance_frames([trk], [0])
    _pyramids_equal(trk, 0, O.make_images(other, sc.nl))
    ctx.advance_frames([], [])
