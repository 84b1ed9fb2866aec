#!/usr/bin/env python3
"""bench.py -- frames/s of one forward+backward of the rasterisation operator on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json config 3, run as 1920x1088 -- SURVEY 8d): synth(5e5, 1920, 1088, 0.02, SH deg 3,
seed 0), resident in HBM before the timed region.  A step = forward (projection, binning, sort, blend),
dL/dimage = 2*(image-0.5), backward (blend backward, per-point chain, hook payload -- a no-op hook is
installed like the reference trainer's controller.update).  Every rank renders ONE view per step and, with
N > 1, the ranks sum their point gradients with ONE RCCL all-reduce per step (weak scaling): that is BASELINE
config 4 for every N.  The views are the eight poses of config 4, and they CYCLE: rank r renders pose
(r + step) mod 8, a new pose every iteration as the reference's own harness does
(benchmark/inference_benchmark.py:110-156), warm-up and timed region alike.  Other schedules (--views-per-rank,
--reduce view, --scheme gaussian) are opt-in and say so in their metric string.
Rank 0 prints ONE JSON line.  `roofline` is for the kernel that takes the most time, timed with HIP events
recorded inside libgsrast on the launch stream during the timed region; `cpu_baseline` is the CPU oracle
(the C restatement of the reference algorithm, OpenMP) on the same frame, rank 0, N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KERNEL_MODEL_DOC = "DESIGN.md section 'Kernels and rooflines'"


def kernel_models(N, M, K, P, T, key_bits, evals):
    """Algorithmic bytes / flops ONE launch of each kernel has to move.  bound, amount.
    `evals` = pixel x list-entry evaluations the reference algorithm performs on this frame
    (sum over pixels of last_effective_index - tile_start; a saturated pixel stops there, RAST:424-425, 609-610)."""
    passes = (key_bits + 7) // 8
    return {
        "k_filter": ("hbm", 17 * N + 1 * N),
        "k_project": ("hbm", 1 * N + 4 * N + 4 * M + 236 * M + 16 * M + 64 * M + 8 * M + 4 * M + 4 * M),
        "k_keygen": ("hbm", 28 * M + 4 * M + 8 * K),
        "k_sort_hist": ("hbm", 4 * K),
        "k_sort_scatter": ("hbm", 16 * K),
        "k_blend_fwd": ("mfma", 16.3 * evals),        # FP32 VALU flops; peak = 157.3 TF (= f32 MFMA peak)
        "k_blend_bwd_tile": ("mfma", 48.8 * evals),
        "k_sum_rows": ("hbm", 49 * K + 8 * M + 48 * M),          # 48-B row + 1 flag byte per pair (unvisited rows are skipped in practice)
        "k_bwd_points": ("hbm", 4 * N + 300 * M + 248 * N),
    }, passes


PEAK = {"hbm": (8000.0, "GB/s"), "mfma": (157.3, "TFLOP/s")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="cfg3_headline")
    ap.add_argument("--mode", choices=["fwdbwd", "forward"], default=None,
                    help="forward = inference under torch.no_grad (BASELINE config 5); default: forward for cfg5, fwd+bwd otherwise")
    ap.add_argument("--no-hook", dest="hook", action="store_false", default=True,
                    help="by default a backward_valid_point_hook is installed, as the reference trainer does "
                         "(GaussianPointTrainer.py:82-93), so the backward also produces the whole "
                         "BackwardValidPointHookInput payload; --no-hook measures the operator without one")
    ap.add_argument("--scene", default=None, help="a trained scene file (.parquet in the reference's layout or an INRIA .ply) to "
                    "render instead of the synthetic generator's points; camera and resolution still come from --workload")
    ap.add_argument("--views-per-rank", type=int, default=1,
                    help="views each rank renders per step (gradients accumulated over them, one all-reduce per step); a step then is V "
                         "forward+backward passes and `value` counts views per second.  Default 1 for every N = BASELINE config 4; "
                         "V > 1 is the gradient accumulation a data-parallel trainer uses to amortise the all-reduce -- another workload, "
                         "labelled as such in `metric`")
    ap.add_argument("--fixed-pose", action="store_true",
                    help="render the same pose every step (rank r: pose r) instead of cycling the eight poses of config 4")
    ap.add_argument("--reduce", choices=["step", "view"], default="step",
                    help="view-parallel scheme with N > 1: 'step' = one all-reduce per step on the locally accumulated gradient; "
                         "'view' = one asynchronous all-reduce per view, overlapped with the next view's forward+backward")
    ap.add_argument("--scheme", choices=["view", "gaussian"], default="view",
                    help="N > 1: 'view' = parameters replicated, gradient all-reduce; 'gaussian' = every rank owns 1/N of the Gaussians "
                         "and renders one view, projected records and per-splat sums exchanged by two all-to-alls, no all-reduce")
    ap.add_argument("--rgb-only", action="store_true",
                    help="forward mode only: GaussianPointCloudRasterisationConfig.rgb_only=True (RAST:781, 478-484): only the image is produced")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--breakdown-steps", type=int, default=10)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from taichi_3d_gaussian_splatting_amd import CameraInfo, GaussianPointCloudRasterisation as Rast, _native
    from taichi_3d_gaussian_splatting_amd import distributed as gsd
    from taichi_3d_gaussian_splatting_amd.synthetic import CONFIGS, make_scene, synth, view_pose, workload_args

    # GS_BENCH_REHEARSAL=1: every rank uses cuda:0 and the collective runs over gloo -- only to rehearse the
    # N > 1 code path on a one-GPU box; the numbers of such a run mean nothing.
    rehearsal = os.environ.get("GS_BENCH_REHEARSAL") == "1"
    rank, world, local_rank = gsd.init_from_env("gloo" if rehearsal else "nccl")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    mode = args.mode or ("forward" if args.workload.startswith("cfg5") else "fwdbwd")
    partial_tiles = args.workload == "cfg3_1080p"      # extension: the true 1920x1080 frame (68 tile rows, last one half used)
    cfgw = dict(CONFIGS["cfg3_headline"], H=1080) if partial_tiles else workload_args(args.workload)
    scene = synth(**cfgw) if partial_tiles else make_scene(args.workload)
    if args.scene:
        from taichi_3d_gaussian_splatting_amd import scene_io
        pc_np, ft_np = (scene_io.load_inria_ply if args.scene.endswith(".ply") else scene_io.load_parquet)(args.scene)
        scene.point_cloud, scene.point_cloud_features = pc_np, ft_np
        scene.point_invalid_mask = np.zeros(pc_np.shape[0], np.int8)
        scene.point_object_id = np.zeros(pc_np.shape[0], np.int32)
    V = max(1, args.views_per_rank)
    N_POSES = 8                                   # the eight poses of BASELINE config 4, 2 degrees apart
    q, t = view_pose(rank % N_POSES, N_POSES)
    H, W = scene.height, scene.width
    pc = torch.tensor(scene.point_cloud, device=dev, requires_grad=True)
    feat = torch.tensor(scene.point_cloud_features, device=dev, requires_grad=True)
    inp = Rast.GaussianPointCloudRasterisationInput(
        point_cloud=pc, point_cloud_features=feat,
        point_object_id=torch.tensor(scene.point_object_id, device=dev),
        point_invalid_mask=torch.tensor(scene.point_invalid_mask, device=dev),
        camera_info=CameraInfo(torch.tensor(scene.camera_intrinsics, device=dev), H, W, 0),
        q_pointcloud_camera=torch.tensor(q, device=dev), t_pointcloud_camera=torch.tensor(t, device=dev),
        color_max_sh_band=3)
    hook_calls = []
    if args.rgb_only and mode != "forward":
        raise SystemExit("--rgb-only frames cannot be back-propagated (RAST:478-484): use --mode forward")
    rcfg = Rast.GaussianPointCloudRasterisationConfig(rgb_only=bool(args.rgb_only))
    rcfg.allow_partial_tiles = partial_tiles
    probe_cfg = Rast.GaussianPointCloudRasterisationConfig()
    probe_cfg.allow_partial_tiles = partial_tiles
    module = Rast(rcfg,
                  backward_valid_point_hook=(lambda payload: hook_calls.append(1)) if args.hook else None)
    L = _native.lib()
    names = L.gs_kernel_names().decode().split(",")

    ncoll = []
    minus_one = torch.full((H, W, 3), -1.0, device=dev)
    poses = [tuple(torch.tensor(x, device=dev) for x in view_pose(v, N_POSES)) for v in range(N_POSES)]
    step_no = [0]

    def my_views():
        """The V poses this rank renders in the current step: a new one every view rendered, all ranks of a step on different
        poses (rank r, view j of step k: pose (r + (k * V + j) ) mod 8; fixed: pose (r * V + j) mod 8)."""
        k = 0 if args.fixed_pose else step_no[0]
        return [((rank + k * V + j) if not args.fixed_pose else (rank * V + j)) % N_POSES for j in range(V)]
    reducer = gsd.OverlappedGradientReducer() if (world > 1 and args.reduce == "view" and args.scheme == "view") else None
    gp_backend = None
    gp_stats = {}
    if args.scheme == "gaussian":
        if mode != "fwdbwd" or V != 1:
            raise SystemExit("--scheme gaussian runs forward+backward with one view per rank")
        bounds = gsd.shard_bounds(pc.shape[0], world)
        lo, hi = bounds[rank], bounds[rank + 1]
        shard_inp = Rast.GaussianPointCloudRasterisationInput(
            point_cloud=pc.detach()[lo:hi].contiguous(), point_cloud_features=feat.detach()[lo:hi].contiguous(),
            point_object_id=inp.point_object_id[lo:hi].contiguous(), point_invalid_mask=inp.point_invalid_mask[lo:hi].contiguous(),
            camera_info=inp.camera_info, q_pointcloud_camera=poses[0][0], t_pointcloud_camera=poses[0][1], color_max_sh_band=3)
        gp_backend = gsd.HipStageBackend(shard_inp, poses[:world] if world <= N_POSES else [poses[v % N_POSES] for v in range(world)], rcfg)
        module = gp_backend.st.module                  # the ctx the profiler reads

    def step():
        views = my_views()
        step_no[0] += 1
        if gp_backend is not None:
            if world > 1:
                _, _, _, st = gsd.gaussian_parallel_step(gp_backend, lambda img: torch.add(minus_one, img, alpha=2.0))
            else:                                          # one rank: the same four stages chained, nothing to exchange
                rec, h = gp_backend.project(0)
                img, rh = gp_backend.render(rec)
                sums = gp_backend.backward_render(rh, torch.add(minus_one, img, alpha=2.0))
                gp_backend.backward_project(h, sums)
                st = {"bytes_sent": 0, "collectives": 0}
            gp_stats.update(st)
            return
        if mode == "forward":
            with torch.no_grad():
                for v in views:
                    inp.q_pointcloud_camera, inp.t_pointcloud_camera = poses[v]
                    module(inp)
            return
        pc.grad = None
        feat.grad = None
        for v in views:
            inp.q_pointcloud_camera, inp.t_pointcloud_camera = poses[v]
            image, _, _ = module(inp)
            g = torch.add(minus_one, image.detach(), alpha=2.0)   # dL/dimage of an MSE to mid-grey, 2*(image-0.5), one launch
            image.backward(g)
            if reducer is not None:
                reducer.submit(pc.grad, feat.grad)                # asynchronous: runs beside the next view
                pc.grad = None
                feat.grad = None
        if reducer is not None:
            pc.grad, feat.grad = reducer.finish()
            ncoll.append(V)
        elif world > 1:
            ncoll.append(gsd.all_reduce_point_gradients(pc.grad, feat.grad))

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    ctx = None

    def prof(mask):
        nonlocal ctx
        ctx = module._ctx_for(dev)
        _native.check(L.gs_profile_enable(ctx, C.c_uint64(mask)), "gs_profile_enable")

    def prof_read(reset=True):
        ms = (C.c_double * len(names))()
        cnt = (C.c_int64 * len(names))()
        _native.check(L.gs_profile_read(ctx, ms, cnt, len(names), 1 if reset else 0), "gs_profile_read")
        return {n: (ms[i], cnt[i]) for i, n in enumerate(names) if cnt[i] > 0}

    for _ in range(max(args.warmup, 1)):
        step()
    sync_all()
    # Frame statistics for the byte / flop models: an inference call (all outputs) per pose this rank renders, averaged -- M, K and
    # the evaluation count differ a little from pose to pose, and the kernel times below are averages over the same poses.
    probe = Rast(probe_cfg)
    N, P = pc.shape[0], H * W
    ty, tx = (H + 15) // 16, (W + 15) // 16
    tile_of_pixel = (torch.arange(H, device=dev) // 16)[:, None] * tx + (torch.arange(W, device=dev) // 16)[None, :]
    probe_poses = sorted(set(my_views())) if args.fixed_pose else list(range(N_POSES))
    stats = []
    for v in probe_poses:
        inp.q_pointcloud_camera, inp.t_pointcloud_camera = poses[v]
        with torch.no_grad():
            probe(inp)
        fr = probe.last_frame
        # evaluations the reference algorithm performs: every pixel walks its tile list up to its last effective entry
        last = probe.last_forward_outputs["pixel_offset_of_last_effective_point"].to(torch.int64)
        tstart = fr.export("tile_points_start").to(torch.int64)[tile_of_pixel]
        stats.append((fr.n_points_in_camera, fr.n_keys, int((last - tstart).clamp_(min=0).sum().item()), fr.sort_key_bits))
    T = fr.n_tiles
    M, K, evals = (int(round(sum(x[i] for x in stats) / len(stats))) for i in range(3))
    key_bits = max(x[3] for x in stats)
    sizing_seen = module.last_frame.sizing if getattr(module, "last_frame", None) is not None else None
    sync_all()

    # ---- untimed diagnostic pass: every kernel timed, to find the dominant one ----
    prof((1 << len(names)) - 1)
    args.breakdown_steps = max(args.breakdown_steps, 1)         # the dominant kernel has to be found before the timed region
    for _ in range(args.breakdown_steps):
        step()
    sync_all()
    breakdown = prof_read()
    per_step_ms = {n: v[0] / args.breakdown_steps for n, v in breakdown.items()}
    dominant = max(per_step_ms, key=per_step_ms.get)

    # ---- the timed region: only the dominant kernel carries events ----
    prof(1 << names.index(dominant))
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    dom = prof_read()[dominant]
    prof(0)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        models, passes = kernel_models(N, M, K, P, T, key_bits, evals)
        bound, amount = models.get(dominant, ("hbm", 0))
        avg_ms = dom[0] / max(dom[1], 1)
        peak, unit = PEAK[bound]
        achieved = amount / (avg_ms * 1e-3) / (1e9 if bound == "hbm" else 1e12) if avg_ms > 0 else 0.0
        # Counters come from separate rocprofv3 --pmc passes (profiles/collect_pmc.sh, collect_sq.sh) and are quoted here only when
        # they were taken with THIS build of the kernels (same digest of the kernel sources); otherwise null + the reason.
        digest = _native.source_digest()
        traffic, traffic_note = None, "no profiles/pmc_traffic.json"
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("_source_digest") != digest:
                    traffic_note = f"stale counters: taken with kernel sources {pmc.get('_source_digest')}, this build is {digest}"
                elif dominant not in pmc.get(args.workload, {}):
                    traffic_note = "no counters for this workload / kernel"
                else:
                    traffic = pmc[args.workload][dominant]
                    traffic_note = f"profiles/pmc_traffic.json, kernel sources {digest}: (2*FETCH_SIZE + WRITE_SIZE) KiB per launch, separate passes"
            except Exception as e:
                traffic_note = f"unreadable: {e}"
        # SQ counters of the same kernel: wave-level VALU instructions per launch against the SIMD-cycles of the launch measured
        # live.  A wave64 FP32 instruction occupies its SIMD for 2 cycles at the datasheet rate (157.3 TFLOP/s = 256 CUs x 4 SIMDs
        # x 32 lanes x 2 flop x 2.4 GHz); measured (tools/ubench_valu.hip): 2.5 for plain fma/mul/add, 4.2 for DPP / v_cmp /
        # v_cndmask, 8.2 for v_exp / v_rcp / v_sqrt.
        valu = None
        sq_path = os.path.join(ROOT, "profiles", "sq_counters.json")
        if os.path.exists(sq_path) and avg_ms > 0:
            try:
                sq = json.load(open(sq_path))
                if sq.get("source_digest") != digest:
                    valu = {"note": f"stale counters: taken with kernel sources {sq.get('source_digest')}, this build is {digest}"}
                elif sq.get("workload") == args.workload and dominant in sq.get("kernels", {}):
                    insts = sq["kernels"][dominant]["SQ_INSTS_VALU"]
                    simd_cycles = avg_ms * 1e-3 * 2.4e9 * 1024
                    # issue cost of the kernel's own instruction mix on a perfectly fed SIMD (mixes counted from the ISA)
                    floor = {"k_blend_bwd_tile": 3.0, "k_blend_fwd": 2.6}.get(dominant)
                    valu = {"wave_valu_instructions_per_launch": insts, "simd_cycles_per_launch": round(simd_cycles),
                            "cycles_per_valu_instruction": round(simd_cycles / insts, 3),
                            "issue_floor_cycles_per_instruction": floor,
                            "fraction_of_issue_floor": round(floor * insts / simd_cycles, 4) if floor else None,
                            "source": f"profiles/sq_counters.json (SQ_INSTS_VALU, kernel sources {digest}) / live HIP-event launch time, 2.4 GHz, 1024 SIMDs"}
            except Exception:
                valu = None
        fwd_bytes = 17 * N + 332 * M + 88 * K + 28 * P + 8 * T          # SURVEY 8d byte model
        bwd_bytes = (88 * K + 28 * P + 528 * M + 248 * N) if mode == "fwdbwd" else 0
        ms_per_step = elapsed / args.steps * 1e3
        ms_per_view = ms_per_step / V
        is_cfg4 = V == 1 and args.scheme == "view" and args.reduce == "step"
        sched = ("" if is_cfg4 else
                 (f"; gaussian-parallel schedule" if args.scheme == "gaussian" else
                  f"; {V} views per rank per step, all-reduce per {args.reduce} (NOT BASELINE config 4: gradient accumulation)"))
        gen = "synth_clustered" if "clustered" in args.workload else "synth"
        out = {
            "metric": (("fps fwd+bwd @1920x1080 (run as 1920x1088), 5e5 Gaussians" if args.workload == "cfg3_headline" and mode == "fwdbwd"
                        else f"fps {mode} {args.workload} @{W}x{H}, {N} Gaussians") + sched),
            "value": round(world * V * args.steps / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "ms_per_view": round(ms_per_view, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": ("synthetic" if not args.scene else "file:" + os.path.basename(args.scene)),
            "config": {"workload": f"{args.workload}: {gen}(N={cfgw['N']}, {W}x{H}, sigma0={cfgw['sigma0']}, sh_deg={cfgw['sh_deg']}, seed 0), "
                                   f"{'fwd+bwd' if mode == 'fwdbwd' else 'forward only (torch.no_grad)'}, {V} view(s) per GPU per step, "
                                   + ("the 8 poses of BASELINE config 4 cycled (rank r, step k: pose (r + k) mod 8)" if not args.fixed_pose
                                      else "one fixed pose per rank")
                                   + (f", {('one' if args.reduce == 'step' else str(V))} sum all-reduce of 59*N f32 point gradients per step" if world > 1 and args.scheme == "view" else ""),
                       "points": N, "points_in_camera": M, "sort_pairs": K, "tiles": T, "sort_key_bits": key_bits,
                       "pixel_entry_evaluations": evals,
                       "frame_statistics": f"M, K, evaluations: mean over the {len(stats)} pose(s) rendered",
                       "parallelism": (f"gaussian-parallel x{world}: 1/{world} of the Gaussians per rank, one view per rank, 2 all-to-alls"
                                       if args.scheme == "gaussian" else
                                       f"view-parallel x{world}, {V} view(s) per rank per step, all-reduce per {args.reduce}"),
                       "views_per_rank": V, "views_per_step": world * V, "poses_cycled": 1 if args.fixed_pose else N_POSES,
                       "forward_sizing": sizing_seen,
                       "forward_dispatch_order": ("natural (GS_FWD_ORDER_HINT=0)" if os.environ.get("GS_FWD_ORDER_HINT", "1")[:1] == "0" or mode != "fwdbwd"
                                                  else "tile order left by the previous backward of this context (scheduling only; "
                                                       + ("the same pose every step" if args.fixed_pose and V == 1 else "the previous step's pose, 2 degrees away") + ")"),
                       "collectives_per_step": (gp_stats.get("collectives", 0) if args.scheme == "gaussian" else (ncoll[-1] if ncoll else 0)),
                       "exchange_bytes_sent_per_rank_per_step": (gp_stats.get("bytes_sent") if args.scheme == "gaussian"
                                                                 else (0 if world == 1 else 236 * N * (V if args.reduce == "view" else 1))),
                       "rehearsal": rehearsal,
                       "backward_hook": bool(args.hook), "rgb_only": bool(args.rgb_only)},
            "roofline": {"kernel": dominant, "bound": bound, "bound_kind": ("fp32_vector" if bound == "mfma" else "hbm"),
                         "bound_detail": ("FP32 vector ALU (no MFMA anywhere on this path: there is no dense contraction); the schema's "
                                          "'mfma' slot is used because 157.3 TFLOP/s is both the f32 vector and the f32 MFMA dense peak; "
                                          "`achieved` counts the reference algorithm's flops per evaluation, not issued instructions"
                                          if bound == "mfma" else "HBM bandwidth"),
                         "achieved": round(achieved, 3), "peak": peak, "unit": unit,
                         "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_note, "valu": valu,
                         "avg_launch_ms": round(avg_ms, 4), "launches": dom[1], "model": KERNEL_MODEL_DOC, "kernel_sources": digest},
            "frame_hbm": {"algorithmic_bytes_per_view": fwd_bytes + bwd_bytes,
                          "achieved_GBps": round((fwd_bytes + bwd_bytes) / (ms_per_view * 1e-3) / 1e9, 1),
                          "frac_of_8TBps": round((fwd_bytes + bwd_bytes) / (ms_per_view * 1e-3) / 8e12, 4)},
            "kernels_ms_per_view": {k: round(v / V, 4) for k, v in sorted(per_step_ms.items(), key=lambda kv: -kv[1])},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            reps = []
            for r in range(args.cpu_reps + 1):
                c0 = time.perf_counter()
                ocfg = oracle.default_config(allow_partial_tiles=int(partial_tiles))
                f, _ = oracle.forward(scene.point_cloud, scene.point_cloud_features, scene.point_invalid_mask,
                                      scene.point_object_id, q, t, scene.camera_intrinsics, H, W, ocfg)
                if mode == "fwdbwd":
                    oracle.backward(f, 2.0 * (f.rasterized_image - 0.5), 3, ocfg)
                reps.append(time.perf_counter() - c0)
                f.free()
            best = float(np.median(reps[1:]))
            out["cpu_baseline"] = {"value": round(1.0 / best, 4), "unit": "frames/s", "cores": oracle.num_threads(),
                                   "kind": "port",
                                   "sample": f"the whole {args.workload} frame (pose 0 of 8), {mode}, median of {args.cpu_reps} runs after 1 warm-up "
                                             f"({best:.2f} s/frame); oracle/gs_oracle.c -O2 -fopenmp, host has {os.cpu_count()} logical CPUs"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
