"""GPU (-m gpu): the HIP operator, called through the C ABI, against the CPU oracle.

Bar: integer arrays bit-exact; here the per-point f32 products of the
forward and the accumulated alpha are bit-exact too (same IEEE operation sequence on both sides),
the blended images within 2e-6 of their maximum; gradients within 1e-4 of the tensor's max magnitude (float summation order differs: the reference itself uses unordered
atomics, RAST:674-696)."""
import numpy as np
import pytest
import torch

from oracle import oracle
from taichi_3d_gaussian_splatting_amd.synthetic import CONFIGS, synth, view_pose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import parity_util
    return parity_util


def _fwd_bwd(P, scene, q, t, band=3, hook=True, cfg_kw=None, seed=0):
    got = {}
    cfg_kw = dict(cfg_kw or {})
    partial = cfg_kw.pop("allow_partial_tiles", False)
    cfg = P.Rast.GaussianPointCloudRasterisationConfig(**cfg_kw)
    cfg.allow_partial_tiles = partial                  # class attribute, like the grad factors (RAST:782-786)
    cfg_kw["allow_partial_tiles"] = partial
    module = P.Rast(cfg, backward_valid_point_hook=(lambda x: got.setdefault("hook", x)) if hook else None)
    inp = P.make_input(scene, q, t, band)
    ocfg = oracle.default_config(**{k: (int(v) if isinstance(v, bool) else v) for k, v in cfg_kw.items()})
    f, feat_after = P.run_oracle(scene, q, t, ocfg)
    outs = module(inp)
    P.assert_forward_parity(module, inp, outs, f, feat_after)
    image = outs[0]
    rng = np.random.default_rng(seed)
    target = torch.tensor(rng.uniform(0, 1, image.shape).astype(np.float32), device=image.device)
    g_image = (2.0 * (image.detach() - target))
    image.backward(g_image)
    b = P.assert_backward_parity(module, inp, g_image.cpu().numpy(), f, band,
                                 module.last_backward_extras if hook else None, ocfg)
    return module, inp, f, b, got


def test_cfg1_plumbing_full_parity(P):
    """BASELINE config 1: 1e4 Gaussians, 256x256, SH degree 0."""
    s = synth(**CONFIGS["cfg1_plumbing"])
    q, t = view_pose()
    module, inp, f, b, got = _fwd_bwd(P, s, q, t, band=0)
    assert f.M == 10000 and f.K == 31202
    h = got["hook"]                                   # reference tests :207-282 (shapes) + values
    assert h.grad_point_in_camera.shape == (f.M, 3) and h.grad_pointfeatures_in_camera.shape == (f.M, 56)
    assert h.grad_viewspace.shape == (f.M, 2) and h.magnitude_grad_viewspace.shape == (f.M,)
    assert h.num_overlap_tiles.shape == (f.M,) and h.num_affected_pixels.shape == (f.M,)
    ids = f.point_id_in_camera_list
    assert np.array_equal(h.point_id_in_camera_list.cpu().numpy(), ids)
    assert np.array_equal(h.grad_point_in_camera.cpu().numpy(), inp.point_cloud.grad.cpu().numpy()[ids])
    assert np.array_equal(h.grad_pointfeatures_in_camera.cpu().numpy(), inp.point_cloud_features.grad.cpu().numpy()[ids])
    assert np.array_equal(h.num_overlap_tiles.cpu().numpy(), f.num_overlap_tiles)
    assert np.array_equal(h.point_depth.cpu().numpy(), f.point_in_camera[:, 2])
    assert np.array_equal(h.point_uv_in_camera.cpu().numpy(), f.point_uv)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_small_scenes_pose_mask_objects(P, seed):
    """Non-identity, non-unit pose quaternions, two objects, invalid rows, all SH bands."""
    rng = np.random.default_rng(seed)
    s = synth(3000, 160, 96, 0.08, sh_deg=3, seed=seed)
    s.point_invalid_mask[rng.random(3000) < 0.1] = 1
    s.point_object_id[:] = (rng.random(3000) < 0.3).astype(np.int32)
    q = np.array([[0.02, 0.05, -0.01, 0.99], [-0.03, -0.02, 0.04, 1.02]], np.float32)
    t = np.array([[0.1, -0.05, 0.2], [-0.2, 0.1, -0.1]], np.float32)
    _fwd_bwd(P, s, q, t, band=int(seed % 4), seed=seed)


def test_large_and_degenerate_splats(P):
    """Splats far larger than the image, needle-like ones, opaque stacks that saturate pixels."""
    s = synth(800, 128, 128, 0.5, sh_deg=3, seed=5)
    s.point_cloud_features[:100, 4:7] = np.log(3.0)            # huge
    s.point_cloud_features[100:200, 4] = np.log(2.0)           # needles
    s.point_cloud_features[100:200, 5:7] = np.log(1e-4)
    s.point_cloud_features[:, 7] = 6.0                         # nearly opaque -> saturation + 0.99 clamp
    q, t = view_pose()
    module, inp, f, b, _ = _fwd_bwd(P, s, q, t)
    assert (f.pixel_accumulated_alpha > 0.9998).any()          # the T < 1e-4 stop is exercised


@pytest.mark.parametrize("n,w,h,rows_per_pair", [(3000, 640, 400, 4), (20000, 1536, 1024, 1)])
def test_splats_covering_most_of_the_image(P, n, w, h, rows_per_pair):
    """A few translucent background splats whose 3-sigma box covers (nearly) every tile, among ordinary ones: more than a thousand
    (point, tile) rows each.  k_keygen walks such a splat's pairs with the whole block, k_sum_rows sums its rows with the whole
    block (k_backward.hip: SUM_ROWS_GIANT), one more with its wave (> 32 rows), the rest four lanes per point."""
    s = synth(n, w, h, 0.02, sh_deg=3, seed=21)
    s.point_cloud_features[:6, 4:7] = np.log(2.5)              # six splats far larger than the image
    s.point_cloud_features[:6, 7] = -3.0                       # faint: everything behind them still counts
    s.point_cloud[:6, :2] *= 0.2                               # near the optical axis
    s.point_cloud_features[6:40, 4:7] = np.log(0.3)            # and some that cover a few hundred tiles
    s.point_cloud_features[6:40, 7] = -2.0
    q, t = view_pose()
    module, inp, f, b, _ = _fwd_bwd(P, s, q, t, band=3, hook=True, seed=5)
    rows = np.asarray(f.num_overlap_tiles).astype(np.int64) * rows_per_pair
    assert (rows > 1024).sum() >= 3 and ((rows > 32) & (rows <= 1024)).sum() >= 10, (rows.max(), (rows > 1024).sum())


@pytest.mark.parametrize("case", ["empty", "all_invalid", "behind", "one_point"])
def test_edge_cases(P, case):
    s = synth(64, 64, 48 if False else 64, 0.1, seed=3)
    if case == "empty":
        s.point_cloud = s.point_cloud[:0]; s.point_cloud_features = s.point_cloud_features[:0]
        s.point_invalid_mask = s.point_invalid_mask[:0]; s.point_object_id = s.point_object_id[:0]
    elif case == "all_invalid":
        s.point_invalid_mask[:] = 1
    elif case == "behind":
        s.point_cloud[:, 2] = -s.point_cloud[:, 2]
    elif case == "one_point":
        s.point_invalid_mask[1:] = 1
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t)
    f, feat_after = P.run_oracle(s, q, t)
    outs = module(inp)
    P.assert_forward_parity(module, inp, outs, f, feat_after)
    outs[0].sum().backward()
    if case != "one_point":
        assert f.K == 0 and not outs[0].any() and not inp.point_cloud.grad.any()
    else:
        P.assert_backward_parity(module, inp, np.ones((s.height, s.width, 3), np.float32), f, 3)


def test_rgb_only_and_no_grad(P):
    """inference path: torch.no_grad + rgb_only (benchmark/inference_benchmark.py:110-156, RAST:781)."""
    s = synth(5000, 256, 128, 0.05, seed=7)
    q, t = view_pose(1, 3)
    f, _ = P.run_oracle(s, q, t)
    for rgb_only in (False, True):
        module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig(rgb_only=rgb_only))
        for requires_grad in (False, True):      # parameters that require grad, rendered under no_grad (a trainer's validation pass)
            with torch.no_grad():
                image, depth, count = module(P.make_input(s, q, t, requires_grad=requires_grad))
            assert P.rel_err(image.cpu().numpy(), f.rasterized_image) < P.IMAGE_TOL
            if not rgb_only:
                assert np.array_equal(count.cpu().numpy(), f.pixel_valid_point_count)
        if rgb_only:                                  # with a graph being recorded there is nothing to back-propagate through (RAST:478-484)
            with pytest.raises(RuntimeError, match="rgb_only"):
                module(P.make_input(s, q, t, requires_grad=True))


def test_backward_is_bitwise_reproducible(P):
    """No float atomics in this implementation: two runs give identical bits."""
    s = synth(4000, 128, 128, 0.08, seed=11)
    q, t = view_pose()
    grads = []
    for _ in range(2):
        module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
        inp = P.make_input(s, q, t)
        img = module(inp)[0]
        (img * img).sum().backward()
        grads.append((inp.point_cloud.grad.cpu().numpy().copy(), inp.point_cloud_features.grad.cpu().numpy().copy()))
    assert np.array_equal(grads[0][0].view(np.uint32), grads[1][0].view(np.uint32))
    assert np.array_equal(grads[0][1].view(np.uint32), grads[1][1].view(np.uint32))


def test_dispatch_order_hint_from_an_earlier_frame_changes_nothing(P):
    """A context keeps the last backward's tile order (heaviest first) as the dispatch order of its next forward blend
    (gs_api.hip: order_hint).  Pure scheduling: a frame rendered after a backward at ANOTHER pose (a hint that does not fit), at
    the same pose, or on a fresh context (no hint) gives the same bits, forward and backward; a change of the tile grid drops it."""
    s = synth(30000, 256, 192, 0.05, seed=5)
    q, t = view_pose()
    q2, t2 = view_pose(3, 8)

    def run(module, qq, tt):
        inp = P.make_input(s, qq, tt)
        img, depth, count = module(inp)
        (img * img).sum().backward()
        return [x.detach().cpu().numpy().copy() for x in (img, depth, count, inp.point_cloud.grad, inp.point_cloud_features.grad)]

    fresh = run(P.Rast(P.Rast.GaussianPointCloudRasterisationConfig()), q, t)
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    run(module, q2, t2)                      # leaves the ordering of another view behind
    other = run(module, q, t)                # rendered with that ordering
    same = run(module, q, t)                 # rendered with its own ordering
    small = synth(500, 64, 64, 0.1, seed=6)  # another tile grid on the same context: the hint must not be used
    inp = P.make_input(small, q, t)
    img_small = module(inp)[0]
    (img_small * img_small).sum().backward()
    ref_small = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())(P.make_input(small, q, t, requires_grad=False))[0]
    assert np.array_equal(img_small.detach().cpu().numpy().view(np.uint32), ref_small.cpu().numpy().view(np.uint32))
    back = run(module, q, t)                 # back to the first grid: no stale ordering either
    for got in (other, same, back):
        for a, b in zip(got, fresh):
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)


def test_not_a_number_stays_where_the_reference_puts_it(P):
    """The backward blend lets lanes that take nothing from a splat run the first quadrant's arithmetic on alpha = 0 -- only
    while everything entering it is finite (k_backward.hip: `clean`).  With a NaN colour on one splat, and with a NaN in the
    incoming image gradient, the exec-masked form must take over: NaN then reaches exactly the gradient elements it reaches in
    the reference algorithm (the oracle), and every other element keeps its value."""
    s = synth(3000, 128, 96, 0.08, sh_deg=3, seed=23)
    q, t = view_pose()
    f0, _ = P.run_oracle(s, q, t)
    cnt = P.oracle.backward(f0, np.ones((96, 128, 3), np.float32), 3)["num_affected_pixels"]
    victim = int(f0.point_id_in_camera_list[int(np.argmax(cnt))])               # a splat that certainly contributes somewhere
    for case in ("nan_colour", "nan_gradient"):
        sc = synth(3000, 128, 96, 0.08, sh_deg=3, seed=23)
        g = (np.random.default_rng(24).standard_normal((96, 128, 3)) * 0.1).astype(np.float32)
        if case == "nan_colour":
            sc.point_cloud_features[victim, 8] = np.nan                           # red DC coefficient
        else:
            g[40, 50, 1] = np.nan
        module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
        inp = P.make_input(sc, q, t, 3)
        image = module(inp)[0]
        image.backward(torch.tensor(g, device=image.device))
        f, _ = P.run_oracle(sc, q, t)
        b = P.oracle.backward(f, g, 3)
        for got, ref, name in ((inp.point_cloud.grad.cpu().numpy(), b["grad_pointcloud"], "xyz"),
                               (inp.point_cloud_features.grad.cpu().numpy(), b["grad_pointcloud_features"], "features")):
            assert np.isnan(ref).any(), (case, name)                              # the case does exercise NaN
            assert np.array_equal(np.isnan(got), np.isnan(ref)), (case, name, int(np.isnan(got).sum()), int(np.isnan(ref).sum()))
            ok = ~np.isnan(ref)
            assert np.abs(got[ok] - ref[ok]).max() <= 1e-4 * np.abs(ref[ok]).max(), (case, name)


def test_argument_errors(P):
    s = synth(16, 64, 64, 0.1)
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t, requires_grad=False)
    inp.camera_info.camera_width = 72                                   # RAST:1193
    with pytest.raises(AssertionError):
        module(inp)
    inp = P.make_input(s, q, t, requires_grad=False)
    inp.point_cloud_features = inp.point_cloud_features.double()
    with pytest.raises(TypeError):
        module(inp)
    inp = P.make_input(s, q, t, requires_grad=False)
    inp.point_cloud = inp.point_cloud.cpu()
    with pytest.raises(ValueError):
        module(inp)


def test_training_loop_loss_decreases(P):
    """reference tests :284-351 (test_backward_coverage), shortened: Adam on a 32x32 target."""
    s = synth(256, 32, 32, 0.3, seed=2)
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t)
    pc, feat = inp.point_cloud.detach().requires_grad_(True), inp.point_cloud_features.detach().requires_grad_(True)
    inp.point_cloud, inp.point_cloud_features = pc, feat
    opt = torch.optim.Adam([pc, feat], lr=1e-3)
    target = torch.rand(32, 32, 3, device=pc.device)
    losses = []
    for it in range(150):
        inp.color_max_sh_band = it // 40
        opt.zero_grad()
        img = module(inp)[0]
        loss = ((img - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]


def test_cfg2_truck7k_scale_parity(P):
    """BASELINE config 2: 2.3e5 Gaussians, 976x544, SH degree 3."""
    s = synth(**CONFIGS["cfg2_truck7k"])
    q, t = view_pose()
    _fwd_bwd(P, s, q, t, band=3, hook=True)


def test_sort_keys_wider_than_32_bits(P):
    """A depth code of 31 bits + tile bits forces the 64-bit key path; same order as the oracle's i64 sort."""
    s = synth(3000, 160, 96, 0.08, sh_deg=3, seed=4)
    q, t = view_pose()
    module, inp, f, b, _ = _fwd_bwd(P, s, q, t, cfg_kw=dict(depth_to_sort_key_scale=2.0e8), hook=False)
    assert module.last_backward_extras is not None
    with torch.no_grad():
        module(P.make_input(s, q, t, requires_grad=False))
    assert module.last_frame.sort_key_bits > 32
    assert int(f.sort_key.max() & 0xFFFFFFFF) > 2 ** 30


def test_cfg3_headline_full_parity(P):
    """BASELINE config 3 at full size: 5e5 Gaussians, 1920x1088, SH degree 3, forward + backward."""
    s = synth(**CONFIGS["cfg3_headline"])
    q, t = view_pose()
    module, inp, f, b, _ = _fwd_bwd(P, s, q, t, band=3, hook=True)
    assert f.M == 472215 and f.K == 5999394


def test_cfg3_views_of_the_8_gpu_run(P):
    """Two of the eight poses of BASELINE config 4 (the views ranks 0 and 7 render)."""
    s = synth(**CONFIGS["cfg3_headline"])
    for v in (0, 7):
        q, t = view_pose(v, 8)
        _fwd_bwd(P, s, q, t, band=3, hook=False, seed=v)


def test_cfg5_inference_2e6_forward_parity_and_properties(P):
    """BASELINE config 5: 2e6 Gaussians, 1920x1088, inference (no_grad), all outputs."""
    s = synth(**CONFIGS["cfg5_infer2e6"])
    q, t = view_pose()
    f, feat_after = P.run_oracle(s, q, t)
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t, requires_grad=False)
    with torch.no_grad():
        outs = module(inp)
    P.assert_forward_parity(module, inp, outs, f, feat_after)
    # size-independent properties of the binning, checked on the GPU result itself
    fr = module.last_frame
    keys = fr.export("sort_key")
    vals = fr.export("point_offset_with_sort_key").to(torch.int64)
    assert bool((keys[1:] >= keys[:-1]).all()), "keys sorted"
    ties = keys[1:] == keys[:-1]
    assert bool((vals[1:][ties] > vals[:-1][ties]).all()), "stable: ties keep ascending in-camera offset"
    ts, te = fr.export("tile_points_start").to(torch.int64), fr.export("tile_points_end").to(torch.int64)
    assert int((te - ts).sum()) == fr.n_keys == int(fr.export("num_overlap_tiles").sum())
    tile_of_key = keys >> 32
    nonempty = te > ts
    assert bool((tile_of_key[ts[nonempty]] == torch.nonzero(nonempty).flatten()).all())
    assert bool((tile_of_key[te[nonempty] - 1] == torch.nonzero(nonempty).flatten()).all())
    image, depth, count = outs
    assert float(image.min()) >= 0.0 and float(image.max()) <= 1.0 + 1e-5


def test_cfg5_inference_2e6_rgb_only(P):
    """BASELINE config 5 with rgb_only=True (RAST:409-410, 464-484: only the image is produced), the form the reference's
    inference benchmark can run (benchmark/inference_benchmark.py): the image equals the all-outputs image bit for bit
    (same blend arithmetic, three stores fewer) and the oracle's within the image bar; the binning is identical."""
    s = synth(**CONFIGS["cfg5_infer2e6"])
    q, t = view_pose()
    f, feat_after = P.run_oracle(s, q, t)
    imgs = {}
    for rgb_only in (True, False):
        module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig(rgb_only=rgb_only))
        inp = P.make_input(s, q, t, requires_grad=False)
        with torch.no_grad():
            outs = module(inp)
        P.assert_forward_parity(module, inp, outs, f, feat_after, rgb_only=rgb_only)
        imgs[rgb_only] = outs[0].cpu().numpy()
    assert np.array_equal(imgs[True].view(np.uint32), imgs[False].view(np.uint32))


def test_several_frames_in_flight_and_arena_is_stable(P):
    """Two forwards, then their backwards in reverse order (frames pin separate buffer sets); afterwards the
    context allocates nothing more for repeated steps (grow-only arena, no hipMalloc in steady state)."""
    import ctypes as C
    from taichi_3d_gaussian_splatting_amd import _native
    s1, s2 = synth(3000, 160, 96, 0.08, seed=21), synth(5000, 160, 96, 0.05, seed=22)
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    i1, i2 = P.make_input(s1, q, t), P.make_input(s2, q, t)
    f1, fa1 = P.run_oracle(s1, q, t)
    f2, fa2 = P.run_oracle(s2, q, t)
    o1 = module(i1)
    o2 = module(i2)                                # second forward before the first backward
    g2 = torch.ones_like(o2[0])
    o2[0].backward(g2)
    P.assert_backward_parity(module, i2, g2.cpu().numpy(), f2, 3)
    g1 = torch.ones_like(o1[0])
    o1[0].backward(g1)
    P.assert_backward_parity(module, i1, g1.cpu().numpy(), f1, 3)
    ctx = module._ctx_for(i1.point_cloud.device)
    sizes = []
    for _ in range(4):
        inp = P.make_input(s2, q, t)
        img = module(inp)[0]
        img.sum().backward()
        torch.cuda.synchronize()
        sizes.append(_native.lib().gs_ctx_device_bytes(ctx))
    # a frame lives as long as its graph node (like the reference's saved tensors): o1 and o2 still pin two buffer sets,
    # the loop needs one more on its first pass (the previous iteration's graph dies only when `img` is rebound) and
    # nothing after that
    assert sizes[2:] == sizes[1:-1], sizes


def test_forward_without_backward_releases_its_frame(P):
    """A kept frame whose autograd graph is dropped (no backward) goes back to the pool."""
    import gc
    s = synth(2000, 128, 96, 0.08, seed=23)
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    from taichi_3d_gaussian_splatting_amd import _native
    ctx = None
    sizes = []
    for _ in range(6):
        inp = P.make_input(s, q, t)
        out = module(inp)
        ctx = module._ctx_for(inp.point_cloud.device)
        del out, inp
        module.last_frame = None
        gc.collect()
        torch.cuda.synchronize()
        sizes.append(_native.lib().gs_ctx_device_bytes(ctx))
    assert sizes[-1] == sizes[1], sizes              # the pool stops growing: dropped frames are reused


def test_controller_accumulators_match_reference_update(P):
    """the six accumulators of GaussianPointAdaptiveController.update() (reference
    GaussianPointAdaptiveController.py:130-141) maintained by the backward kernel, against a numpy restatement of
    those lines applied to the oracle's hook quantities, over two views."""
    from taichi_3d_gaussian_splatting_amd import ControllerAccumulators
    s = synth(4000, 160, 96, 0.08, sh_deg=3, seed=31)
    N = 4000
    acc = ControllerAccumulators.zeros(N, P.DEV)
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig(), controller_accumulators=acc)
    ref = dict(num_in_camera=np.zeros(N, np.int64), num_pixels=np.zeros(N, np.int64), vs=np.zeros(N), vs_avg=np.zeros(N),
               pos=np.zeros((N, 3)), pos_norm=np.zeros(N))
    for v in range(2):
        q, t = view_pose(v, 2)
        inp = P.make_input(s, q, t)
        f, _ = P.run_oracle(s, q, t)
        img = module(inp)[0]
        g = 2.0 * (img.detach() - 0.5)
        img.backward(g)
        b = oracle.backward(f, g.cpu().numpy(), 3)
        ids = f.point_id_in_camera_list
        npix = b["num_affected_pixels"].astype(np.int64)
        mag = b["magnitude_grad_viewspace"][ids].astype(np.float64)
        ref["num_in_camera"][ids] += 1                                    # :133
        ref["num_pixels"][ids] += npix                                    # :134
        ref["vs"][ids] += mag                                             # :135-136
        with np.errstate(divide="ignore", invalid="ignore"):
            avg = mag / npix                                              # :137
        avg[np.isnan(avg)] = 0                                            # :138
        ref["vs_avg"][ids] += avg                                         # :139
        gp = b["grad_pointcloud"][ids].astype(np.float64)
        ref["pos"][ids] += gp                                             # :140
        ref["pos_norm"][ids] += np.linalg.norm(gp, axis=1)                # :141
    assert np.array_equal(acc.accumulated_num_in_camera.cpu().numpy(), ref["num_in_camera"])
    assert np.array_equal(acc.accumulated_num_pixels.cpu().numpy(), ref["num_pixels"])
    for got, want in [(acc.accumulated_view_space_position_gradients, ref["vs"]),
                      (acc.accumulated_view_space_position_gradients_avg, ref["vs_avg"]),
                      (acc.accumulated_position_gradients, ref["pos"]),
                      (acc.accumulated_position_gradients_norm, ref["pos_norm"])]:
        assert P.rel_err(got.cpu().numpy().astype(np.float64), want) < P.GRAD_TOL
    acc.reset()
    assert not acc.accumulated_num_in_camera.any() and not acc.accumulated_position_gradients.any()


def test_heavy_tile_lists_and_clustered_scene(P):
    """A scene unlike the uniform generator: 1500 image-filling splats on top of 30k small ones (every tile list is
    thousands of entries long, most pairs come from the wave-cooperative keygen / row-sum paths), a dense cluster,
    points hugging the near plane and a camera translated into the cloud."""
    rng = np.random.default_rng(77)
    s = synth(31500, 640, 368, 0.03, sh_deg=3, seed=77)
    s.point_cloud_features[:1500, 4:7] = np.log(rng.uniform(1.0, 4.0, (1500, 3))).astype(np.float32)   # huge
    s.point_cloud_features[:1500, 7] = rng.uniform(-4.0, -1.0, 1500).astype(np.float32)                # faint, so lists stay long
    s.point_cloud[1500:6500] = (np.array([0.3, -0.2, 3.0]) + rng.normal(0, 0.05, (5000, 3))).astype(np.float32)   # cluster
    s.point_cloud[6500:7500, 2] = rng.uniform(0.75, 0.85, 1000).astype(np.float32)                      # around near_plane = 0.8
    q = np.array([[0.01, -0.02, 0.03, 1.0]], np.float32)
    t = np.array([[0.05, 0.02, 1.0]], np.float32)
    module, inp, f, b, _ = _fwd_bwd(P, s, q, t, band=3, hook=True, seed=5)
    lens = f.tile_points_end - f.tile_points_start
    assert lens.max() > 1500 and f.num_overlap_tiles.max() > 500


@pytest.mark.parametrize("n,width,height,sigma0", [(4000, 250, 203, 0.08), (300, 17, 15, 0.3), (20000, 641, 33, 0.05),
                                                   (1500, 100, 1, 0.1)])
def test_partial_edge_tiles_extension(P, n, width, height, sigma0):
    """Extension: image sizes that are not multiples of 16.  Tile counts round up and the
    pixels of an edge tile that fall outside the image do not exist; everything else is the reference's
    algorithm, so the same bars hold against the oracle run with the same switch."""
    s = synth(n, width, height, sigma0, seed=width)
    q, t = view_pose()
    module, inp, f, b, got = _fwd_bwd(P, s, q, t, cfg_kw={"allow_partial_tiles": True})
    assert f.arrays["tile_points_start"].shape[0] == ((width + 15) // 16) * ((height + 15) // 16)
    assert f.K > 0 and got["hook"].magnitude_grad_viewspace_on_image.shape == (height, width, 2)


def test_partial_tiles_true_1080p_at_the_headline_density(P):
    """5e5 Gaussians at a true 1920x1080 frame (68 tile rows, the last one half outside the image)."""
    c = dict(CONFIGS["cfg3_headline"]); c["H"] = 1080
    s = synth(**c)
    q, t = view_pose()
    module, inp, f, b, got = _fwd_bwd(P, s, q, t, cfg_kw={"allow_partial_tiles": True})
    assert f.H == 1080 and f.arrays["tile_points_start"].shape[0] == 120 * 68


def test_partial_tiles_switch_off_is_the_reference_contract(P):
    """Without the switch the C ABI rejects such a size itself (error -1), as RAST:1193-1194 asserts."""
    from taichi_3d_gaussian_splatting_amd import _native
    s = synth(64, 40, 40, 0.2)
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t, requires_grad=False)
    with pytest.raises(AssertionError):
        module(inp)
    L = _native.lib()
    import ctypes as C
    ctx = _native.shared_ctx(0)
    cfg = module._c_config()
    scene = module._c_scene(inp.point_cloud, inp.point_cloud_features, inp.point_invalid_mask, inp.point_object_id)
    out = torch.empty(5, 40, 40, 3, device=P.DEV)
    cam = _native.GsCamera(inp.q_pointcloud_camera.data_ptr(), inp.t_pointcloud_camera.data_ptr(), 1,
                           inp.camera_info.camera_intrinsics.data_ptr(), 40, 40)
    fo = _native.GsForwardOut(*[out[i].data_ptr() for i in range(5)])
    frame = C.c_void_p()
    rc = L.gs_forward(ctx, C.byref(scene), C.byref(cam), C.byref(cfg), C.byref(fo), 0, C.byref(frame),
                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -1 and b"multiples of 16" in L.gs_last_error()


@pytest.mark.parametrize("n", [1, 3, 1001, 4099])
def test_point_counts_that_are_not_multiples_of_four(P, n):
    """The two gradients share one 59*N allocation; the float4-stored feature rows must stay 16-byte aligned for any N
    (found by tools/parity_soak.py: N = 1016 + 1 failed when the positions came first in that buffer)."""
    s = synth(n, 96, 64, 0.15, seed=n)
    q, t = view_pose()
    module, inp, f, b, got = _fwd_bwd(P, s, q, t)
    assert inp.point_cloud_features.grad.data_ptr() % 16 == 0


def test_backward_through_depth_only_gives_zero_gradients(P):
    """The reference ignores the depth gradient (RAST:1157-1163); with unmaterialised grads the image gradient is None then."""
    s = synth(500, 64, 64, 0.1, seed=3)
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t)
    image, depth, count = module(inp)
    depth.sum().backward()
    assert inp.point_cloud.grad is not None and not inp.point_cloud.grad.any()
    assert not inp.point_cloud_features.grad.any()


@pytest.mark.parametrize("seed", [60163, 60266, 141447])
def test_soak_seeds_with_ill_conditioned_splats(P, seed):
    """The three scenes of the parity soaks (tools/parity_soak.py; narrow, tall images full of huge anisotropic splats; 3 of
    ~9000 random scenes) where the DEFAULT backward and the oracle are more than 1e-4 of the tensor maximum apart (1.05e-4 /
    1.14e-4 on the scale gradient, 1.3e-4 / 1.5e-4 on rotation and scale in the third) -- run in BOTH forms of the kernel.

    Cause: the reference forms d p / d Sigma' per contribution as 0.5 p (Sigma^-1 (d d^T) Sigma^-1) with two f32 matrix
    products (UTIL:343-345) and the oracle follows it; for a long thin conic a*dx and b*dy cancel in Sigma^-1 d, so those
    products carry an absolute error of 2^-24 (|a dx| + |b dy|)^2 against a value (a dx + b dy)^2 that is orders smaller.
    The default k_blend_bwd_tile forms the same matrix as v v^T with v = Sigma^-1 d (three fused multiply-adds instead of 24
    operations in the kernel that dominates the frame) and keeps those digits.

    Proof, not argument: with gs_config.bwd_reference_order = 1 the kernel evaluates UTIL:331-348 in the reference's own
    order and the gap CLOSES -- every column group within 1e-5 of the tensor maximum of the oracle (what is left is the f32
    summation order) -- while the default form shows the documented gap, sits several times closer to a float64 autograd
    restatement than to the oracle, and the oracle sits as far from float64 as from the default form.  The same closure is
    shown on the CPU alone by tests/test_oracle_exp_sensitivity.py::test_dpdcov_order_is_the_soak_gap.
    The default form is a deliberate, documented deviation (include/gs_rasterizer.h: bwd_reference_order; DESIGN.md section 3;
    distribution at configs 2 and 3 in profiles/r03_strict_vs_fast.json): its tensor-level bar in THESE three scenes is 2e-4,
    the per-element bar (parity_util.py) holds without any allowance, and everywhere else in the suite it meets 1e-4."""
    import json
    import os
    # Seed 141447 (18 x 441 pixels, 747 splats in camera) is ill-conditioned all round: even where HIP and oracle agree to 4e-6
    # (positions, opacity) both sit 3e-5 .. 8e-5 from float64, so its float64 bars are wider
    hip_f64_tol, orc_f64_tol, closer = (1e-4, 3e-4, 2.5) if seed == 141447 else (2e-5, 2e-4, 5.0)
    c = P.soak_case(seed)
    s, q, t, partial, rng = c["scene"], c["q"], c["t"], c["partial"], c["rng"]
    unit = dict(grad_color_factor=1.0, grad_high_order_color_factor=1.0, grad_s_factor=1.0, grad_q_factor=1.0, grad_alpha_factor=1.0)
    ocfg = oracle.default_config(allow_partial_tiles=int(partial), **unit)
    f, feat_after = P.run_oracle(s, q, t, ocfg)
    target = None
    report = {"seed": seed, "image": [c["W"], c["H"]], "points_in_camera": int(f.M), "metric": "max |a - b| / max |b| per column group"}
    grads = {}
    for form in ("fast", "reference_order"):
        cfg = P.Rast.GaussianPointCloudRasterisationConfig()
        cfg.allow_partial_tiles = partial
        cfg.backward_reference_order = form == "reference_order"
        for k, v in unit.items():
            setattr(cfg, k, v)
        module = P.Rast(cfg)
        inp = P.make_input(s, q, t, 3)
        outs = module(inp)
        P.assert_forward_parity(module, inp, outs, f, feat_after)
        image = outs[0]
        if target is None:
            target = torch.tensor(rng.uniform(0, 1, image.shape).astype(np.float32), device=image.device)
        g = 2.0 * (image.detach() - target)
        image.backward(g)
        # every element under the per-element bar in both forms; tensor level: 1e-5 in the reference's order, 2e-4 in the fast form
        b = P.assert_backward_parity(module, inp, g.cpu().numpy(), f, 3, None, ocfg, tensor_tol=1e-5 if form == "reference_order" else 2e-4)
        grads[form] = (inp.point_cloud.grad.cpu().numpy(), inp.point_cloud_features.grad.cpu().numpy(), b)
    ref_pc, ref_ft = P.float64_autograd_gradients(s, q, t, f, feat_after, g.cpu().numpy())
    gp, gf, b = grads["fast"]
    sp, sf, _ = grads["reference_order"]
    op, of = b["grad_pointcloud"], b["grad_pointcloud_features"]
    # SH columns are left out of the float64 comparison: the restatement evaluates the colour with the forward's ray origin,
    # the reference's backward with t_pointcloud_camera (RAST:731-732), which differ for the non-unit pose quaternion used here
    for name, a_hip, a_strict, a_orc, a_f64 in [("xyz", gp, sp, op, ref_pc), ("q", gf[:, 0:4], sf[:, 0:4], of[:, 0:4], ref_ft[:, 0:4]),
                                                ("s", gf[:, 4:7], sf[:, 4:7], of[:, 4:7], ref_ft[:, 4:7]),
                                                ("opacity", gf[:, 7:8], sf[:, 7:8], of[:, 7:8], ref_ft[:, 7:8])]:
        r = {"fast_vs_oracle": P.rel_err(a_hip, a_orc), "reference_order_vs_oracle": P.rel_err(a_strict, a_orc),
             "fast_vs_float64": P.rel_err(a_hip, a_f64), "oracle_vs_float64": P.rel_err(a_orc, a_f64)}
        report[name] = r
        assert r["reference_order_vs_oracle"] < 1e-5, (name, r)        # the gap is the operation order of UTIL:343-345 and nothing else
        assert r["fast_vs_float64"] < hip_f64_tol, (name, r)           # the default form against exact arithmetic
        assert r["oracle_vs_float64"] < orc_f64_tol, (name, r)         # the reference's f32 operation order against it
    worst = max(("xyz", "q", "s", "opacity"), key=lambda n: report[n]["fast_vs_oracle"])
    report["widest_gap_in"] = worst
    assert report[worst]["fast_vs_oracle"] > 8e-5                        # the documented gap is really there in the default form
    assert report[worst]["fast_vs_oracle"] > closer * report[worst]["fast_vs_float64"]
    assert report[worst]["oracle_vs_float64"] > 0.7 * report[worst]["fast_vs_oracle"]
    report["per_element_bar_use_vs_oracle"] = {k: v["bar_use_max"] for k, v in b["margins"].items()}
    os.makedirs(os.path.join(P.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(P.ROOT, "gpurun_out", f"soak_seed_{seed}.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report))


def test_reference_order_backward_at_cfg2(P):
    """gs_config.bwd_reference_order on a BASELINE-sized frame (config 2): the same bars as the default form, and the two forms
    agree with each other far inside them."""
    s = synth(**CONFIGS["cfg2_truck7k"])
    q, t = view_pose()
    res = {}
    for strict in (False, True):
        cfg = P.Rast.GaussianPointCloudRasterisationConfig()
        cfg.backward_reference_order = strict
        module = P.Rast(cfg)
        inp = P.make_input(s, q, t, 3)
        image = module(inp)[0]
        g = 2.0 * (image.detach() - 0.5)
        image.backward(g)
        res[strict] = (inp.point_cloud.grad.cpu().numpy(), inp.point_cloud_features.grad.cpu().numpy())
        if strict:
            f, _ = P.run_oracle(s, q, t)
            P.assert_backward_parity(module, inp, g.cpu().numpy(), f, 3)
    assert P.rel_err(res[True][0], res[False][0]) < 1e-5 and P.rel_err(res[True][1], res[False][1]) < 1e-5


def test_predicted_sizing_and_its_redo_change_nothing(P):
    """The forward queues binning, sort and blend on PREDICTED sizes (what the last frame of the context needed + 25 %) and reads
    the frame's counters only after its last launch (gs_api.hip: run_forward_tail).  A frame whose prediction held, a frame
    whose pair count outgrew the prediction (the scene doubles between two frames) and a frame whose depth codes got wider
    (depth_to_sort_key_scale x 64) must all give, bit for bit, what a fresh context gives with exact sizes -- forward outputs,
    every exported intermediate, gradients."""
    import os
    if os.environ.get("GS_PREDICT_SIZES") == "0":
        pytest.skip("the library's diagnostic switch GS_PREDICT_SIZES=0 turns the mechanism under test off")
    q, t = view_pose()
    small, big = synth(6000, 256, 192, 0.05, seed=41), synth(12000, 256, 192, 0.07, seed=42)

    def run(module, scene):
        inp = P.make_input(scene, q, t)
        img, depth, count = module(inp)
        fr = module.last_frame
        exports = {n: fr.export(n).cpu().numpy() for n in ("sort_key", "point_offset_with_sort_key", "tile_points_start", "tile_points_end")}
        (img * img).sum().backward()
        outs = [x.detach().cpu().numpy().copy() for x in (img, depth, count, inp.point_cloud.grad, inp.point_cloud_features.grad)]
        return fr.sizing, outs, exports

    def same(a, b):
        for x, y in zip(a[1], b[1]):
            assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
        for n in a[2]:
            assert np.array_equal(a[2][n], b[2][n]), n

    fresh_small = run(P.Rast(P.Rast.GaussianPointCloudRasterisationConfig()), small)
    fresh_big = run(P.Rast(P.Rast.GaussianPointCloudRasterisationConfig()), big)
    assert fresh_small[0] == "exact" and fresh_big[0] == "exact"          # first frame of a context: nothing to predict from
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    first = run(module, small)
    second = run(module, small)
    third = run(module, small)
    assert (first[0], second[0], third[0]) == ("exact", "predicted", "predicted"), (first[0], second[0], third[0])
    same(second, fresh_small); same(third, fresh_small)
    grown = run(module, big)                                               # twice the points: the pair count leaves the prediction
    assert grown[0] == "redone", grown[0]
    same(grown, fresh_big)
    again = run(module, big)
    same(again, fresh_big)
    # wider depth codes on the same context and image size: the key field of the prediction is too narrow
    wide_cfg = P.Rast.GaussianPointCloudRasterisationConfig(depth_to_sort_key_scale=6400.0)
    fresh_wide = run(P.Rast(wide_cfg), big)
    module.config = wide_cfg
    wide = run(module, big)
    assert wide[0] == "redone", wide[0]
    same(wide, fresh_wide)
    back = run(module, big)                                                # and a prediction made from the wide frame holds
    assert back[0] == "predicted", back[0]
    same(back, fresh_wide)


def test_heavy_tiles_shared_by_four_waves_clustered_scene(P):
    """A heavy-tailed scene (synthetic.synth_clustered: an object that fills a tenth of the image with translucent splats, a
    sparse shell, a few huge floaters): the tiles whose work is twice the mean or more get a whole workgroup in the backward
    blend -- four waves, a quadrant each, their per-splat sums added up in LDS before the pair's one row is stored
    (k_backward.hip: COOP).  Same bars against the oracle as everywhere; and the count of such tiles is not zero here."""
    from taichi_3d_gaussian_splatting_amd.synthetic import synth_clustered
    for (n, w, h), waves in (((40000, 480, 272), "four waves per ordinary tile"), ((150000, 1600, 1024), "one wave per ordinary tile")):
        s = synth_clustered(n, w, h, 0.02, sh_deg=3, seed=3)
        q, t = view_pose(2, 8)
        module, inp, f, b, _ = _fwd_bwd(P, s, q, t, band=3, hook=True, seed=9)
        lens = f.tile_points_end - f.tile_points_start
        assert lens.max() > 8 * lens.mean(), (lens.max(), lens.mean())
        import os
        if any(k in os.environ for k in ("GS_BWD_SPLIT_HEAVY", "GS_BWD_SEGMENTS", "GS_BWD_HEAVY_X2")):
            continue                # the suite is also run under the library's diagnostic switches: the policy below is then not the default one
        assert module.last_frame.heavy_tiles() > 0, waves
        tiles, items = module.last_frame.heavy_tiles(), module.last_frame.heavy_tiles(items=True)
        if w * h < 1000 * 600:      # small image: the ordinary waves do not fill the chip, so the forward cut the long lists and the
            if lens.max() > 2048:   # heavy tiles are walked in 512-entry segments (gs_api.hip: want_cuts)
                assert items > tiles, (waves, tiles, items)
        else:                       # large image: one work item per heavy tile
            assert items == tiles, (waves, tiles, items)


def test_heavy_tiles_in_segments_very_long_lists(P):
    """Lists of several thousand entries on a small image (every tile heavy by any measure, the heaviest far beyond the rest): the
    forward stores each pixel's T and accumulated colour every 512 entries of a long list, and the backward hands a heavy tile out
    as one work item per segment, each starting from the record of its cut (k_backward.hip: BwdCoop.seg) -- including the pixels'
    sum |d uv| (magnitude_grad_viewspace_on_image), which the segments leave as partial sums.  Same bars against the oracle; the
    result must not depend on whether the list was cut (GS_BWD_SEGMENTS=0 is covered by the suite run under that switch)."""
    rng = np.random.default_rng(88)
    s = synth(26000, 208, 160, 0.02, sh_deg=3, seed=88)
    # 6000 translucent splats piled on a corner region: lists of thousands of entries there, a few hundred elsewhere
    s.point_cloud[:6000, 0] = rng.uniform(-0.9, -0.4, 6000).astype(np.float32) * s.point_cloud[:6000, 2] / 1.2
    s.point_cloud[:6000, 1] = rng.uniform(-0.7, -0.3, 6000).astype(np.float32) * s.point_cloud[:6000, 2] / 1.2
    s.point_cloud_features[:6000, 4:7] = np.log(rng.uniform(0.05, 0.15, (6000, 3))).astype(np.float32)
    s.point_cloud_features[:6000, 7] = rng.uniform(-5.0, -2.5, 6000).astype(np.float32)
    q, t = view_pose()
    module, inp, f, b, got = _fwd_bwd(P, s, q, t, band=3, hook=True, seed=11)
    lens = f.tile_points_end - f.tile_points_start
    assert lens.max() > 3000, lens.max()
    fr = module.last_frame
    import os
    if not any(k in os.environ for k in ("GS_BWD_SPLIT_HEAVY", "GS_BWD_SEGMENTS", "GS_BWD_HEAVY_X2")):
        assert fr.heavy_tiles() > 0 and fr.heavy_tiles(items=True) >= fr.heavy_tiles() + 4, (fr.heavy_tiles(), fr.heavy_tiles(items=True))


def test_flag_tags_wrap_round_after_255_backwards(P):
    """The backward's `visited` / `touched` flags are tagged with a per-backward value 1..255 instead of being cleared (gs_api.hip:
    prepare_backward_blend); after 255 backwards the buffer is zeroed and the tags start again.  Same bits before, at and
    after the wrap, with a second scene of another size using the same buffer in between."""
    a, b = synth(1500, 96, 64, 0.1, seed=51), synth(2500, 128, 96, 0.08, seed=52)
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())

    def grads(scene):
        inp = P.make_input(scene, q, t)
        img = module(inp)[0]
        (img * img).sum().backward()
        return inp.point_cloud.grad.cpu().numpy().copy(), inp.point_cloud_features.grad.cpu().numpy().copy()

    ref_a, ref_b = grads(a), grads(b)
    f, _ = P.run_oracle(a, q, t)
    for it in range(3, 520):
        if it % 7 == 0:
            got, ref = grads(b), ref_b
        else:
            got, ref = grads(a), ref_a
        if it in (253, 254, 255, 256, 257, 258, 509, 510, 511, 512) or it % 50 == 0:
            assert np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32)), it
            assert np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32)), it


def test_heavy_set_and_cuts_do_not_depend_on_claim_order(P):
    """Soak seed 400829 (round 3): a small image whose heavy tiles outnumber the cap and whose long lists need more cut records than
    a fixed buffer held -- which tiles were heavy / cut then depended on the order of LDS and global atomics, and with it the last
    bits of the gradients (a heavy tile's sums are added in another order than an ordinary one's).  The heavy set is whole bins of
    the work histogram now and every long list gets its records: the same frame rendered three times on one context (exact sizes,
    then predicted ones, which change every capacity) gives the same bits."""
    c = P.soak_case(400829)
    s, q, t, band, partial = c["scene"], c["q"], c["t"], c["band"], c["partial"]
    cfg = P.Rast.GaussianPointCloudRasterisationConfig()
    cfg.allow_partial_tiles = partial
    module = P.Rast(cfg)
    runs = []
    for _ in range(3):
        inp = P.make_input(s, q, t, band)
        img = module(inp)[0]
        (img * img).sum().backward()
        runs.append((module.last_frame.sizing, img.detach().cpu().numpy(), inp.point_cloud.grad.cpu().numpy(), inp.point_cloud_features.grad.cpu().numpy()))
    import os
    if "GS_PREDICT_SIZES" not in os.environ:
        assert [r[0] for r in runs] == ["exact", "predicted", "predicted"]
    for r in runs[1:]:
        for a, b in zip(r[1:], runs[0][1:]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_contributor_on_the_alpha_threshold_behind_a_cut(P):
    """Soak scene 800305 (192x288, lists of 700 - 2400 entries, 20 heavy tiles walked in segments): one pixel has a contributor
    whose alpha sits within an ulp of 1/255 behind a cut.  The reference decides "contributes" once in its forward and again, from
    another expression, in its backward; they disagree there, so the reference's backward chain differs from its forward chain by
    that splat's factor.  A segment that starts from the forward's transmittance would carry the forward's decision (0.39 % on that
    pixel's contributions in front of the cut, five times the per-element bar); the segments notice the disagreement at their
    front edge and the tile is walked again in one piece (k_backward.hip: BwdCoop::redo, k_blend_bwd_repair)."""
    c = P.soak_case(800305)
    s, q, t = c["scene"], c["q"], c["t"]
    module, inp, f, b, _ = _fwd_bwd(P, s, q, t, band=c["band"], hook=True, cfg_kw=dict(allow_partial_tiles=bool(c["partial"])), seed=3)
    import os
    if not any(k in os.environ for k in ("GS_BWD_SPLIT_HEAVY", "GS_BWD_SEGMENTS", "GS_BWD_HEAVY_X2")):
        fr = module.last_frame
        assert fr.heavy_tiles() > 0 and fr.heavy_tiles(items=True) > fr.heavy_tiles()
    assert max(m["bar_use_max"] for m in b["margins"].values()) < 0.6


def test_second_backward_through_a_frame_with_cut_lists(P):
    """backward(retain_graph=True) twice on a frame whose heavy tiles are walked in segments: the forward's cut records are read-only
    for the backward and the segments' partial sums are rewritten with the same values, so the second pass adds exactly the same
    gradient again (autograd accumulates: 2x, an exact doubling) and the hook's per-pixel magnitudes are the same bits."""
    rng = np.random.default_rng(88)
    s = synth(26000, 208, 160, 0.02, sh_deg=3, seed=88)
    s.point_cloud[:6000, 0] = rng.uniform(-0.9, -0.4, 6000).astype(np.float32) * s.point_cloud[:6000, 2] / 1.2
    s.point_cloud[:6000, 1] = rng.uniform(-0.7, -0.3, 6000).astype(np.float32) * s.point_cloud[:6000, 2] / 1.2
    s.point_cloud_features[:6000, 4:7] = np.log(rng.uniform(0.05, 0.15, (6000, 3))).astype(np.float32)
    s.point_cloud_features[:6000, 7] = rng.uniform(-5.0, -2.5, 6000).astype(np.float32)
    q, t = view_pose()
    mags = []
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig(), backward_valid_point_hook=lambda x: mags.append(x.magnitude_grad_viewspace_on_image.cpu().numpy().copy()))
    inp = P.make_input(s, q, t, 3)
    image = module(inp)[0]
    g = 2.0 * (image.detach() - 0.5)
    image.backward(g, retain_graph=True)
    first = (inp.point_cloud.grad.cpu().numpy().copy(), inp.point_cloud_features.grad.cpu().numpy().copy())
    image.backward(g)
    second = (inp.point_cloud.grad.cpu().numpy(), inp.point_cloud_features.grad.cpu().numpy())
    for a, b in zip(first, second):
        assert np.array_equal((2.0 * a).view(np.uint32), b.view(np.uint32))
    assert len(mags) == 2 and np.array_equal(mags[0].view(np.uint32), mags[1].view(np.uint32))
