"""GPU (-m gpu): the path cut at the projected records and the per-splat sums (gs_project_shard, gs_forward_projected,
gs_backward_projected, gs_backward_shard) against the monolithic gs_forward + gs_backward: bit for bit, with the
whole scene as one shard and with the scene cut into shards whose records are concatenated -- the data path of the
Gaussian-parallel multi-GPU scheme, here on one device.  Plus the handle / stream / argument
hardening of the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

from taichi_3d_gaussian_splatting_amd import CameraInfo, _native
from taichi_3d_gaussian_splatting_amd.stages import StagedRasteriser
from taichi_3d_gaussian_splatting_amd.synthetic import synth, view_pose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import parity_util
    return parity_util


def _bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


def _monolithic(P, s, q, t, band, g_fn):
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t, band)
    image, depth, count = module(inp)
    g = g_fn(image.detach())
    image.backward(g)
    return module, inp, (image, depth, count), g


def _shard_input(P, s, lo, hi, q, t, band):
    import copy
    sub = copy.copy(s)
    sub.point_cloud, sub.point_cloud_features = s.point_cloud[lo:hi], s.point_cloud_features[lo:hi]
    sub.point_invalid_mask, sub.point_object_id = s.point_invalid_mask[lo:hi], s.point_object_id[lo:hi]
    return P.make_input(sub, q, t, band, requires_grad=False)


@pytest.mark.parametrize("n_shards", [1, 3])
def test_staged_path_equals_monolithic_bit_for_bit(P, n_shards):
    s = synth(9001, 320, 192, 0.06, sh_deg=3, seed=51)
    s.point_invalid_mask[np.random.default_rng(2).random(9001) < 0.05] = 1
    q, t = view_pose(1, 4)
    g_fn = lambda img: 2.0 * (img - 0.3)
    module, inp, (image, depth, count), g = _monolithic(P, s, q, t, 2, g_fn)
    ref_pc, ref_ft = inp.point_cloud.grad, inp.point_cloud_features.grad

    st = StagedRasteriser()
    bounds = np.linspace(0, 9001, n_shards + 1).astype(int)
    shards = []
    for k in range(n_shards):
        sinp = _shard_input(P, s, bounds[k], bounds[k + 1], q, t, 2)
        rec, ids, frame = st.project_shard(sinp)
        assert bool((ids[1:] > ids[:-1]).all())
        shards.append((sinp, rec, ids, frame))
    records = torch.cat([sh[1] for sh in shards]).contiguous()            # shard-major = ascending global point id
    assert records.shape[0] == module.last_frame.n_points_in_camera
    outs, rframe = st.forward_projected(records, inp.camera_info)
    assert np.array_equal(_bits(outs.rasterized_image), _bits(image))
    assert np.array_equal(_bits(outs.rasterized_depth), _bits(depth))
    assert np.array_equal(outs.pixel_valid_point_count.cpu().numpy(), count.cpu().numpy())
    assert rframe.n_keys == module.last_frame.n_keys
    sums, mag_img = st.backward_projected(rframe, outs, g_fn(outs.rasterized_image), want_magnitude_image=True)
    assert sums.shape == (records.shape[0], 12)
    off = 0
    for (sinp, rec, ids, frame), lo, hi in zip(shards, bounds[:-1], bounds[1:]):
        m = rec.shape[0]
        gr = st.backward_shard(frame, sinp, sums[off:off + m], want_extras=True)
        off += m
        assert np.array_equal(_bits(gr.grad_pointcloud), _bits(ref_pc[lo:hi]))
        assert np.array_equal(_bits(gr.grad_pointcloud_features), _bits(ref_ft[lo:hi]))
    # a projection-only frame exports the per-point arrays, but not the raster stage
    assert shards[0][3].export("point_uv").shape[1] == 2
    with pytest.raises(RuntimeError):
        shards[0][3].export("sort_key")
    with pytest.raises(RuntimeError):
        rframe.export("point_id_in_camera_list")


def test_staged_path_empty_shard_and_empty_view(P):
    s = synth(300, 64, 64, 0.1, seed=52)
    q, t = view_pose()
    st = StagedRasteriser()
    s.point_invalid_mask[:] = 1                                           # nothing in camera
    inp = P.make_input(s, q, t, 3, requires_grad=False)
    rec, ids, frame = st.project_shard(inp)
    assert rec.shape == (0, 16) and ids.shape == (0,)
    outs, rframe = st.forward_projected(rec, inp.camera_info)
    assert not outs.rasterized_image.any() and not outs.pixel_valid_point_count.any()
    sums, _ = st.backward_projected(rframe, outs, torch.ones_like(outs.rasterized_image))
    gr = st.backward_shard(frame, inp, sums)
    assert not gr.flat.any()


def test_backward_twice_with_retain_graph(P):
    """The reference keeps its saved tensors, so backward(retain_graph=True) followed by another backward works
    (RAST:998-1021); here the frame lives as long as the graph node."""
    s = synth(2000, 128, 96, 0.08, seed=53)
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t, 3)
    image = module(inp)[0]
    loss = (image * image).sum()
    loss.backward(retain_graph=True)
    g1 = inp.point_cloud_features.grad.clone()
    inp.point_cloud.grad = None
    inp.point_cloud_features.grad = None
    loss.backward()
    assert np.array_equal(_bits(inp.point_cloud_features.grad), _bits(g1))


def test_stale_and_foreign_frame_handles_are_rejected(P):
    """Frame handles are generation-tagged tickets: released, recycled and foreign ones give GS_ERR_STATE (-4)."""
    L = _native.lib()
    s = synth(500, 64, 64, 0.1, seed=54)
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t, 3, requires_grad=False)
    with torch.no_grad():
        module(inp)
    old = module.last_frame                       # transient frame of the no_grad call
    old_handle = old.handle
    assert old.export("point_uv").shape[0] == old.n_points_in_camera
    with torch.no_grad():
        module(inp)                               # recycles the transient frame: the old ticket stops resolving
    ctx = module._ctx_for(inp.point_cloud.device)
    info = _native.GsFrameInfo()
    assert L.gs_frame_get_info(ctx, old_handle, C.byref(info)) == -4
    assert L.gs_frame_export_count(ctx, old_handle, 1) == -1
    with pytest.raises(RuntimeError):
        old.export("point_uv")
    # double release through the C ABI
    inp2 = P.make_input(s, q, t, 3)
    module(inp2)
    kept = module.last_frame
    h = kept.handle
    assert L.gs_frame_release(ctx, h) == 0
    assert L.gs_frame_release(ctx, h) == -4 and b"not a live frame" in L.gs_last_error()
    kept._h = None
    # a handle of another context
    other = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    other(P.make_input(s, q, t, 3))
    foreign = other.last_frame.handle
    module(P.make_input(s, q, t, 3))
    # same slot number may be live in `module` too, but the generations differ or it is a different frame: at least garbage never crashes
    assert L.gs_frame_get_info(ctx, C.c_void_p(0xdeadbeef00000001), C.byref(info)) == -4
    assert L.gs_frame_get_info(ctx, C.c_void_p(0), C.byref(info)) == -4
    assert foreign is not None


def test_out_of_range_object_ids_are_reported_not_dereferenced(P):
    """point_object_id outside [0, n_objects) on a valid row: the row is left out on the device and gs_forward fails with
    GS_ERR_INVALID_ARGUMENT (the reference reads past its pose arrays)."""
    s = synth(1000, 64, 64, 0.1, seed=55)
    q, t = view_pose()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    s.point_object_id[10] = 7
    s.point_object_id[20] = -3
    with pytest.raises(RuntimeError, match="point_object_id"):
        module(P.make_input(s, q, t, 3, requires_grad=False))
    s.point_invalid_mask[[10, 20]] = 1                                    # invalid rows are never looked at
    with torch.no_grad():
        image = module(P.make_input(s, q, t, 3, requires_grad=False))[0]
    f, _ = P.run_oracle(_valid_ids(s), q, t)
    assert P.rel_err(image.cpu().numpy(), f.rasterized_image) < P.IMAGE_TOL


def test_projections_begun_back_to_back_equal_the_waiting_call(P):
    """gs_project_shard_begin queues the per-point half and returns; the counts are read when first asked for.  Eight views
    begun back to back (what an owner does in the Gaussian-parallel scheme) give the records, ids and -- through gs_backward_shard --
    the gradients of eight gs_project_shard calls, bit for bit; a bad object id surfaces when the frame is first read; frames
    dropped unread and more begun frames than counter slots are harmless."""
    s = synth(7000, 256, 160, 0.06, sh_deg=3, seed=58)
    s.point_invalid_mask[np.random.default_rng(3).random(7000) < 0.05] = 1
    st = StagedRasteriser()
    poses = [view_pose(v, 8) for v in range(8)]
    # fresh tensors for every call: the projection normalises the quaternions in place (RAST:264-266) and doing that twice is
    # not bitwise idempotent, so like is compared with like
    fresh = lambda: [P.make_input(s, q, t, 3, requires_grad=False) for q, t in poses]
    inputs, inputs_b = fresh(), fresh()
    want = [st.project_shard(i) for i in inputs]                                   # (records, ids, frame), waiting each time
    frames = [st.project_shard_begin(i) for i in inputs_b]
    got = [st.project_shard_finish(f) for f in frames]
    rng = np.random.default_rng(4)
    for (rec_w, ids_w, fr_w), (rec_g, ids_g), fr_g, inp, inp_b in zip(want, got, frames, inputs, inputs_b):
        assert fr_g.n_points_in_camera == fr_w.n_points_in_camera == rec_g.shape[0] > 0
        assert np.array_equal(_bits(rec_g), _bits(rec_w)) and np.array_equal(ids_g.cpu().numpy(), ids_w.cpu().numpy())
        sums = torch.tensor(rng.standard_normal((rec_g.shape[0], 12)).astype(np.float32), device=rec_g.device)
        sums[:, 10] = torch.tensor(rng.integers(0, 5, rec_g.shape[0]).astype(np.int32), device=rec_g.device).view(torch.float32)
        a, b = st.backward_shard(fr_w, inp, sums), st.backward_shard(fr_g, inp_b, sums)
        assert np.array_equal(_bits(a.grad_pointcloud), _bits(b.grad_pointcloud))
        assert np.array_equal(_bits(a.grad_pointcloud_features), _bits(b.grad_pointcloud_features))
    # more frames begun than the context has counter slots (63), most of them never read: the extra ones simply wait at once
    many = [st.project_shard_begin(inputs[k % 8]) for k in range(80)]
    assert many[70].n_points_in_camera == want[70 % 8][2].n_points_in_camera
    assert many[3].n_points_in_camera == want[3][2].n_points_in_camera
    del many
    # an out-of-range object id is reported when the frame is first read, and the context stays usable
    bad = synth(1000, 64, 64, 0.1, seed=55)
    bad.point_object_id[10] = 7
    fr = st.project_shard_begin(P.make_input(bad, *view_pose(), 3, requires_grad=False))
    with pytest.raises(RuntimeError, match="point_object_id"):
        fr.n_points_in_camera
    rec, ids = st.project_shard_finish(st.project_shard_begin(fresh()[0]))
    assert np.array_equal(_bits(rec), _bits(want[0][0]))


def _valid_ids(s):
    import copy
    c = copy.copy(s)
    c.point_object_id = np.where(s.point_invalid_mask == 1, 0, s.point_object_id).astype(np.int32)
    return c


def test_calls_on_another_stream_are_ordered_after_the_previous_stream(P):
    """One ctx, forward on stream A, backward on stream B without any synchronisation by the caller: the library makes B
    wait for what the ctx issued on A (scratch is recycled in stream order), so the result equals the one-stream result."""
    s = synth(20000, 512, 320, 0.05, seed=56)
    q, t = view_pose()
    ref_module, ref_inp, _, g = _monolithic(P, s, q, t, 3, lambda img: 2.0 * (img - 0.5))
    torch.cuda.synchronize()
    module = P.Rast(P.Rast.GaussianPointCloudRasterisationConfig())
    inp = P.make_input(s, q, t, 3)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        image = module(inp)[0]
    with torch.cuda.stream(sb):
        sb.wait_stream(sa)               # the caller orders its OWN tensors (image, g); the ctx's scratch is the library's business
        image.backward(g)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(inp.point_cloud_features.grad), _bits(ref_inp.point_cloud_features.grad))
    # and the loss helpers, which share one process-wide ctx, hop streams freely
    from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction
    lf = LossFunction(LossFunction.LossFunctionConfig(enable_regularization=False))
    a, b = torch.rand(3, 64, 80, device=P.DEV), torch.rand(3, 64, 80, device=P.DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        l1 = lf(a, b)[0]
    with torch.cuda.stream(sb):
        l2 = lf(a, b)[0]
    torch.cuda.synchronize()
    assert float(l1) == float(l2)


def test_loss_function_accepts_a_batch(P):
    """LossFunction.py:21-33 takes (B,C,H,W): batch means of L1 and of size_average SSIM."""
    from taichi_3d_gaussian_splatting_amd.LossFunction import LossFunction
    lf = LossFunction(LossFunction.LossFunctionConfig(enable_regularization=False))
    a, b = torch.rand(2, 3, 48, 64, device=P.DEV), torch.rand(2, 3, 48, 64, device=P.DEV)
    L, L1, LD = lf(a, b)
    singles = [lf(a[i], b[i]) for i in range(2)]
    assert abs(float(L) - 0.5 * (float(singles[0][0]) + float(singles[1][0]))) < 1e-6
    assert abs(float(L1) - float(torch.abs(a - b).mean())) < 1e-6
    assert abs(float(LD) - 0.5 * (float(singles[0][2]) + float(singles[1][2]))) < 1e-6


def test_staged_entry_points_reject_frames_of_the_wrong_kind(P):
    """gs_backward needs a gs_forward frame, gs_backward_projected a frame with the raster stage, gs_backward_shard one with the
    projection stage; a frame that was not kept cannot be back-propagated at all (GS_ERR_STATE = -4 through the C ABI)."""
    L = _native.lib()
    s = synth(800, 64, 64, 0.1, seed=57)
    q, t = view_pose()
    st = StagedRasteriser()
    inp = P.make_input(s, q, t, 3, requires_grad=False)
    rec, ids, pframe = st.project_shard(inp)
    outs, rframe = st.forward_projected(rec, inp.camera_info)
    g = torch.ones_like(outs.rasterized_image)
    with pytest.raises(RuntimeError, match="raster stage"):
        st.backward_projected(pframe, outs, g)                    # a projection-only frame has nothing to blend backwards
    sums, _ = st.backward_projected(rframe, outs, g)
    with pytest.raises(RuntimeError, match="projection stage"):
        st.backward_shard(rframe, inp, sums)                      # and a raster-only frame knows no point cloud
    with pytest.raises(ValueError):
        st.backward_shard(pframe, inp, sums[:-1])                 # wrong number of rows
    rec2, ids2, transient = st.project_shard(inp, keep=False)
    with pytest.raises(RuntimeError, match="not kept"):
        st.backward_shard(transient, inp, sums)
    with pytest.raises(TypeError):
        st.forward_projected(rec.double(), inp.camera_info)
    gr = st.backward_shard(pframe, inp, sums)                     # the right pairing still works afterwards
    assert gr.grad_pointcloud.shape == (800, 3)
