"""bench.py -- LiDAR range-view frames/s (forward+backward) of the MI355X-native rasterizer.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one frame: GaussianRasterizer.forward + .backward with non-zero upstream gradients on
colour, depth and occupancy, inputs already resident in HBM (BASELINE.json metric).
N = 1: the whole scene on one GPU.  N > 1: the SAME scene (strong scaling), Gaussians sharded by
range shell across the ranks: RCCL all-gather of the per-shell transmittance plane, all-gather of the
W x H x 5 partial planes, reduce-scatter of the packed per-Gaussian gradient rows (lidargs_dist.py).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     for the dominant kernel (whichever of the forward blend K7 = 68*R_ref + 24*N bytes and the backward
               blend K8 = 68*R_ref + 24*N + 84*V bytes takes longer; SURVEY.md 8d / DESIGN.md section 5):
               ALGORITHMIC bytes per launch / mean duration measured with HIP events on the op's own stream
               inside the timed region, vs 8 TB/s HBM; `traffic` = PMC bytes from the committed profile.
  cpu_baseline the CPU oracle (oracle/lidargs_oracle.c, 1 thread) on a bounded sample of the same
               workload, timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

STAGE_EVERY = 8           # frames between two frames whose stages are bracketed by HIP events
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(P, V, R_ref, N, T):
    """SURVEY.md 8d compact form; returns (fwd, bwd, K7 forward blend, K8 backward blend) bytes per frame."""
    fwd = 48 * P + 112 * V + 112 * R_ref + 8 * T + 24 * N
    bwd = 112 * P + 192 * V + 68 * R_ref + 24 * N
    blend_fwd = 68 * R_ref + 24 * N               # record gather per reference instance + per-pixel outputs
    blend_bwd = 68 * R_ref + 24 * N + 84 * V      # + raster gradients written once per visible Gaussian
    return fwd, bwd, blend_fwd, blend_bwd


def clock_ramp(step, seconds=0.5, fixed_steps=None):
    """A freshly started process on an idle GPU runs its first few hundred frames 5-8 % slower than steady state (measured:
    0.96 vs 0.90 ms/frame): `seconds` of untimed steps before the W warm-up steps the contract asks for.  With more than one
    rank the step holds collectives, so every rank must run the SAME number of steps: pass `fixed_steps`."""
    if fixed_steps is not None:
        for _ in range(fixed_steps):
            step()
        torch.cuda.synchronize()
        return
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        step()
        torch.cuda.synchronize()


def pmc_traffic(kernel_names):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json), corrected as
    MI355X_MICROARCH.md prescribes (FETCH_SIZE doubled on gfx950); None when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    k = json.load(open(files[-1]))["kernels"]
    tot = 0
    for name in kernel_names:
        name, times = name if isinstance(name, tuple) else (name, 1)
        if name not in k:
            return None
        tot += times * k[name]["hbm_bytes_per_launch_corrected"]
    return tot


def cpu_baseline(kind, P_full, H, W, seed, budget_s=20.0):
    """Oracle (single thread) forward+backward on a P/20 sample of the workload, same image size."""
    import lidargs_scenes as sc
    from oracle import lgo
    lgo.build()
    P = max(1000, P_full // 20)
    scene = sc.make_scene(kind, P, H, seed)
    grads = sc.upstream_grads(H, W, seed)
    frames, t0 = 0, time.perf_counter()
    while True:
        f = lgo.forward(scene["means3D"], scene["colors"], scene["opacities"], scene["scales"], scene["rotations"],
                        scene["viewmatrix"], scene["beams"], W, H, bg=scene["bg"])
        lgo.backward(f, *grads)
        frames += 1
        el = time.perf_counter() - t0
        if el > budget_s * 0.5 or frames >= 8:
            break
    fps = frames / el
    # baseline B2: the reference's numpy per-point projector on one 64x2650 sweep's worth of points
    from oracle import range_view
    pts = np.concatenate([scene["means3D"][:20000], scene["colors"][:20000, :1]], 1)
    t1 = time.perf_counter()
    range_view.points_to_pano(pts, H, W, scene["beams"])
    pps = pts.shape[0] / (time.perf_counter() - t1)
    return {
        "projector_points_per_s": pps,
        "value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{frames} fwd+bwd frames of a {P}-Gaussian (1/20) {kind} scene at {H}x{W}, oracle/lidargs_oracle.c, "
                  f"1 thread of {os.cpu_count()} host cores; linear-in-P estimate for the full workload: {fps * P / P_full:.4f} frames/s",
    }


def bench_surfel(args, sc, kind, P, H, W, seed):
    """BASELINE config 5: the 2DGS laser-surfel variant, single GPU, same metric (fwd+bwd frames/s); not the headline line."""
    import numpy as np
    assert args.gpus == 1, "config 5 is a single-GPU configuration"
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU path"
    import build_hip
    build_hip.build()
    from diff_lidargs_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda", 0)
    scene = sc.make_scene(kind, P, H, seed)
    scene["scales"] = np.ascontiguousarray(scene["scales"][:, :2])
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in scene.items()}
    rast = GaussianRasterizer(GaussianRasterizationSettings(
        image_height=H, image_width=W, bg=st["bg"], scale_modifier=1.0, depth_threshold=0.0, viewmatrix=st["viewmatrix"],
        projmatrix=torch.eye(4, device=dev), sh_degree=1, campos=torch.zeros(3, device=dev), prefiltered=False,
        beam_inclinations=st["beams"], lidar_far=80, lidar_near=0, debug=False))
    leaves = {k: st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    means2D = torch.zeros((P, 4), dtype=torch.float32, device=dev, requires_grad=True)
    rng = np.random.default_rng(seed + 200)
    gc = torch.from_numpy(rng.normal(size=(2, H, W)).astype(np.float32)).to(dev)
    go = torch.from_numpy(rng.normal(size=(7, H, W)).astype(np.float32)).to(dev)

    def step():
        for t in list(leaves.values()) + [means2D]:
            t.grad = None
        color, radii, others, _pix = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                          colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
        torch.autograd.backward([color, others], [gc, go])
        return radii
    clock_ramp(step)
    for _ in range(args.warmup):
        radii = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        radii = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    print(json.dumps({
        "metric": "LiDAR range-view frames/sec (fwd+bwd)", "value": args.steps / elapsed, "unit": "frames/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"cfg5: {P} surfels ({kind} scene, seed {seed}) @ {H}x{W}, fwd+bwd, diff_lidargs_surfel_rasterization "
                               f"(2DGS laser-surfel variant), lidar_far=80 lidar_near=0, bg=0",
                   "visible_surfels": int((radii > 0).sum())},
        "roofline": None, "cpu_baseline": None}))


def bench_decode(args):
    """the fused anchor decode (generate_neural_gaussians), forward + backward, single GPU.
    333 334 anchors x 6 offsets = the 2 M candidate Gaussians of the headline frame; not the headline line."""
    import types
    import numpy as np
    assert args.gpus == 1 and torch.cuda.is_available()
    import build_hip
    build_hip.build()
    from neural_gaussians import generate_neural_gaussians
    import lidargs_scenes as sc
    N, k = 333_334, 6
    p, cam, vis, _rng = sc.make_anchor_model(N, k, 5)
    pc = sc.anchor_model_to_torch(p)
    camera = types.SimpleNamespace(camera_center=torch.from_numpy(cam).cuda(), uid=0)
    vmask = torch.from_numpy(vis).cuda()
    mlps = [getattr(pc, "mlp_" + m) for m in ("opacity", "cov", "color", "raydrop")]
    leaves = [pc._anchor_feat, pc._anchor, pc._offset, pc.get_scaling] + [t for m in mlps for t in m.parameters()]

    def step(fn):
        for t in leaves:
            t.grad = None
        xyz, color, opacity, scaling, rot = fn()[:5]
        (xyz.sum() + color.sum() + opacity.sum() + scaling.sum() + rot.sum()).backward()
        return xyz.shape[0]

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            M = step(fn)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            M = step(fn)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, M

    t_hip, M = timed(lambda: generate_neural_gaussians(camera, pc, vmask, is_training=True), args.steps, args.warmup)
    n_vis = int(vis.sum())
    # algorithmic bytes: inputs once per visible anchor + the bool mask, outputs once, and the same again (+ upstream gradients,
    # dense input gradients) for the backward; the per-anchor activations the weight-gradient GEMMs read are NOT counted
    fwd_b = N + n_vis * (128 + 12 + 12 * k + 24) + n_vis * k * 5 + M * 52
    bwd_b = n_vis * (128 + 12 + 12 * k + 24) + M * 52 + N * (128 + 12 + 12 * k + 24)
    out = {"metric": "anchor decode (generate_neural_gaussians) fwd+bwd per second", "value": 1.0 / t_hip, "unit": "decodes/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_hip * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"decode: {N} anchors x {k} offsets, {n_vis} visible, {M} Gaussians out, feat 32, hidden 32, "
                                  f"add_*_dist on (the reference's default model, arguments/__init__.py:51-79)"},
           "roofline": {"bound": "hbm", "kernel": "k_ng_opacity + k_ng_decode + k_ng_backward + weight-gradient GEMMs (whole step)",
                        "achieved": (fwd_b + bwd_b) / t_hip / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (fwd_b + bwd_b) / t_hip / 1e9 / HBM_PEAK_GBS, "traffic": None}}
    if not args.no_cpu_baseline:
        from oracle import neural_gaussians as ng
        from oracle import neural_gaussians_torch as ngt
        params = {m: (seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias) for m, seq in zip(("opacity", "cov", "color", "raydrop"), mlps)}
        flags = (p["add_opacity_dist"], p["add_cov_dist"], p["add_color_dist"])
        t_eager, _ = timed(lambda: ngt.generate(pc._anchor_feat, pc._anchor, pc._offset, pc.get_scaling, params, camera.camera_center, vmask, flags),
                           max(3, args.steps // 5), 2)
        Ns = N // 20
        ps, cams, viss, rng = sc.make_anchor_model(Ns, k, 5)
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < 10.0:
            f = ng.forward(ps, cams, viss)
            Ms = f["xyz"].shape[0]
            ng.backward(ps, f, *[np.ones(sh, np.float32) for sh in ((Ms, 3), (Ms, 2), (Ms, 1), (Ms, 3), (Ms, 4))])
            reps += 1
        t_cpu = (time.perf_counter() - t0) / reps
        out["cpu_baseline"] = {"value": 1.0 / (t_cpu * 20), "unit": "decodes/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"{reps} fwd+bwd of a {Ns}-anchor (1/20) case with oracle/neural_gaussians.py (numpy, BLAS threads as configured), "
                                         f"scaled linearly to the full size",
                               "framework_ops_same_gpu_ms": t_eager * 1e3}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


def bench_loss(args):
    """the fused per-frame image loss + its gradient at the headline image size (64 x 2650)."""
    import numpy as np
    assert args.gpus == 1 and torch.cuda.is_available()
    import build_hip
    build_hip.build()
    from lidar_loss import image_loss
    H, W = 64, 2650
    rng = np.random.default_rng(3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    image = t(rng.random((2, H, W), dtype=np.float32)).requires_grad_(True)
    depth = t((rng.random((1, H, W), dtype=np.float32) * 70).astype(np.float32)).requires_grad_(True)
    gt = t(np.stack([(rng.random((H, W)) > 0.2).astype(np.float32), rng.random((H, W), dtype=np.float32),
                     np.cumsum(rng.normal(scale=0.004, size=(H, W)), axis=1).astype(np.float32) + 20.0]))

    def step(fn):
        image.grad = None; depth.grad = None
        fn().backward()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            step(fn)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            step(fn)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    t_hip = timed(lambda: image_loss(image, depth, gt, 0.2)["loss"], args.steps, args.warmup)
    N = H * W
    bytes_alg = (6 + 3) * 4 * N          # three inputs planes + three gt planes read once, three gradient planes written once
    out = {"metric": "per-frame image loss + gradient (train.py:150-203) per second", "value": 1.0 / t_hip, "unit": "losses/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_hip * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"loss: L1 intensity + L1 depth + 10 MSE ray-drop + (1 - SSIM 11x11) + masked depth-difference L1 on a {H}x{W} frame, "
                                  f"value and gradient, lambda_dssim 0.2"},
           "roofline": {"bound": "hbm", "kernel": "k_loss_pointwise + 4 separable SSIM passes + k_loss_finish (whole step, launch-latency bound)",
                        "achieved": bytes_alg / t_hip / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_alg / t_hip / 1e9 / HBM_PEAK_GBS,
                        "traffic": None}}
    if not args.no_cpu_baseline:
        from oracle import lidar_loss as ol
        from oracle import lidar_loss_torch as olt
        t_eager = timed(lambda: olt.image_loss(image, depth, gt, 0.2), max(5, args.steps // 3), 3)
        im, dp, g = image.detach().cpu().numpy(), depth.detach().cpu().numpy(), gt.cpu().numpy()
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < 8.0:
            ol.forward_backward(im, dp, g, 0.2); reps += 1
        out["cpu_baseline"] = {"value": reps / (time.perf_counter() - t0), "unit": "losses/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} full-size evaluations of oracle/lidar_loss.py (numpy, one thread)",
                               "framework_ops_same_gpu_ms": t_eager * 1e3}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


def bench_train_step(args):
    """The three fused components chained as a training step would run them: anchor decode (f1) -> rasterizer -> image loss (f2)
    -> backward down to anchor features, offsets and MLP weights.  666 667 anchors of the 2 M street scene x 6 offsets."""
    import types
    import numpy as np
    assert args.gpus == 1 and torch.cuda.is_available()
    import build_hip
    build_hip.build()
    import lidargs_scenes as sc
    from diff_lidargs_rasterization import GaussianRasterizer
    from lidar_loss import image_loss
    from neural_gaussians import generate_neural_gaussians
    kind, P, H, W, seed = sc.BASELINE_CONFIGS["cfg3"]
    N, k = 666_667, 6
    scene = sc.make_scene(kind, N, H, seed)
    p, _cam, _vis, rng = sc.make_anchor_model(N, k, seed)
    p["anchor"] = scene["means3D"].astype(np.float32)
    p["offset"] = (0.5 * rng.normal(size=(N, k, 3))).astype(np.float32)
    p["scaling"] = np.concatenate([np.full((N, 3), 0.3, np.float32), scene["scales"].astype(np.float32) * 2.0], 1)   # sigmoid halves them on average
    pc = sc.anchor_model_to_torch(p)
    camera = types.SimpleNamespace(camera_center=torch.zeros(3).cuda(), uid=0)
    st = {k_: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k_, v in scene.items() if k_ in ("viewmatrix", "beams", "bg")}
    rast = GaussianRasterizer(sc.raster_settings(st, W, H))
    gt = torch.from_numpy(np.stack([(rng.random((H, W)) > 0.2).astype(np.float32), rng.random((H, W), dtype=np.float32),
                                    (rng.random((H, W)) * 60).astype(np.float32)])).cuda()
    mlps = [getattr(pc, "mlp_" + m) for m in ("opacity", "cov", "color", "raydrop")]
    leaves = [pc._anchor_feat, pc._anchor, pc._offset, pc.get_scaling] + [t for m in mlps for t in m.parameters()]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    acc = [0.0, 0.0, 0.0, 0.0]
    info = {}

    def step(record):
        for t in leaves:
            t.grad = None
        ev[0].record()
        xyz, color, opacity, scaling, rot, _no, _m = generate_neural_gaussians(camera, pc, None, is_training=True)
        ev[1].record()
        means2D = torch.zeros((xyz.shape[0], 4), device="cuda", requires_grad=True)
        image, depth, _occ, radii = rast(means3D=xyz, means2D=means2D, opacities=opacity, colors_precomp=color, scales=scaling, rotations=rot)
        ev[2].record()
        loss = image_loss(image, depth, gt, 0.2)["loss"] + 0.01 * scaling.prod(dim=1).mean()
        ev[3].record()
        loss.backward()
        ev[4].record()
        if record:
            torch.cuda.synchronize()
            for q in range(4):
                acc[q] += ev[q].elapsed_time(ev[q + 1])
            info.update(gaussians=int(xyz.shape[0]), visible=int((radii > 0).sum()))

    clock_ramp(lambda: step(False))
    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / args.steps
    for _ in range(5):
        step(True)
    print(json.dumps({
        "metric": "training-step core (decode + rasterize + loss, fwd+bwd) per second", "value": 1.0 / t_step, "unit": "steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_step * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"train_step: {N} anchors x {k} offsets -> {info.get('gaussians')} Gaussians ({info.get('visible')} on screen) @ {H}x{W}; "
                               f"generate_neural_gaussians + GaussianRasterizer + image loss, backward to anchors and MLP weights"},
        "stage_ms": {"decode_fwd": acc[0] / 5, "rasterize_fwd": acc[1] / 5, "loss_fwd+grad": acc[2] / 5, "backward(raster+decode)": acc[3] / 5},
        "roofline": None, "cpu_baseline": None}))


def bench_chamfer(args):
    """nearest-neighbour (chamfer) distances between two 169 600-point clouds (one 64 x 2650 frame each),
    both directions, as PointsMeter evaluates them (utils/lidar_utils.py:261-275)."""
    import numpy as np
    assert args.gpus == 1 and torch.cuda.is_available()
    import build_hip
    build_hip.build()
    import chamfer_3D
    n = m = 64 * 2650
    rng = np.random.default_rng(9)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = (d * rng.uniform(3, 70, size=(n, 1))).astype(np.float32)
    b = (a + rng.normal(scale=0.05, size=a.shape)).astype(np.float32)[rng.permutation(n)]
    ta, tb = torch.from_numpy(a[None]).cuda(), torch.from_numpy(b[None]).cuda()
    d1, d2 = torch.empty(1, n, device="cuda"), torch.empty(1, m, device="cuda")
    i1, i2 = torch.empty(1, n, dtype=torch.int32, device="cuda"), torch.empty(1, m, dtype=torch.int32, device="cuda")
    for _ in range(max(1, args.warmup // 3)):
        chamfer_3D.forward(ta, tb, d1, d2, i1, i2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = max(3, args.steps // 5)
    for _ in range(steps):
        chamfer_3D.forward(ta, tb, d1, d2, i1, i2)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    pairs = 2.0 * n * m
    out = {"metric": "chamfer nearest-neighbour evaluations (both directions) per second", "value": 1.0 / t, "unit": "evaluations/s", "n_gpus": 1,
           "steps": steps, "warmup": max(1, args.warmup // 3), "ms_per_step": t * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"chamfer: two clouds of {n} points (one 64x2650 frame each), squared distance + index of the nearest neighbour, both directions",
                      "chamfer_distance": float(d1.mean() + d2.mean())},
           "roofline": {"bound": "valu", "kernel": "k_chamfer_nn (x2)", "achieved": pairs * 8 / t / 1e12, "peak": 78.6, "unit": "TFLOP/s (fp32 vector, unpacked; 8 flop per point pair as the reference writes it)",
                        "frac": pairs * 8 / t / 1e12 / 78.6, "traffic": None}}
    if not args.no_cpu_baseline:
        from oracle import chamfer3d
        ns = 4000
        t0 = time.perf_counter()
        chamfer3d.nearest(a[:ns], b)
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / (tc * (n / ns) * 2), "unit": "evaluations/s", "cores": 1, "kind": "port",
                               "sample": f"{ns} queries against the full {m}-point cloud with oracle/chamfer3d.py (numpy), scaled to 2 x {n} queries"}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


def _flush_c_stdio():
    """RCCL prints a version banner through C stdio when its first communicator comes up; on a pipe that buffer is only written at
    process exit, i.e. AFTER the JSON line.  Flushing it early keeps the JSON line the last thing on stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.workload == "decode":
        return bench_decode(args)
    if args.workload == "loss":
        return bench_loss(args)
    if args.workload == "train_step":
        return bench_train_step(args)
    if args.workload == "chamfer":
        return bench_chamfer(args)
    import lidargs_scenes as sc
    kind, P, H, W, seed = sc.BASELINE_CONFIGS[args.workload]
    if args.workload == "cfg5":
        return bench_surfel(args, sc, kind, P, H, W, seed)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LIDARGS_BENCH_FORCE_SHELLS=1 runs the range-shell code path (RCCL collectives included) with a world of one: the only way
    # to exercise it end to end on a single-GPU box; never set by the driver
    force_shells = world == 1 and os.environ.get("LIDARGS_BENCH_FORCE_SHELLS", "0") == "1"
    if world > 1 or force_shells:
        import torch.distributed as dist
        if force_shells:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if rank == 0:
        import build_hip
        build_hip.build()               # no-op when the in-tree .so is newer than every source; never a stale library
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    from diff_lidargs_rasterization import GaussianRasterizer, _C
    to_torch = lambda sd, device: {k_: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k_, v in sd.items()}
    make_settings = sc.raster_settings

    scene = sc.make_scene(kind, P, H, seed)
    st = to_torch(scene, dev)
    gc, gd, go = (torch.from_numpy(g).to(dev) for g in sc.upstream_grads(H, W, seed))
    settings = make_settings(st, W, H)
    leaves = {k: st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    means2D = torch.zeros((P, 4), dtype=torch.float32, device=dev, requires_grad=True)

    if world == 1 and not force_shells:
        rast = GaussianRasterizer(settings)

        def step():
            for t in list(leaves.values()) + [means2D]:
                t.grad = None
            color, depth, occ, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                            colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
            torch.autograd.backward([color, depth, occ], [gc, gd, go])
    else:
        import lidargs_dist
        comm = lidargs_dist.TorchDistComm()
        # range-shell edges are a load-balancing choice, not a result: cut once for this (static) scene and view
        import math
        beams = st["beams"]
        tile_rad = (16 * 2 * math.pi / W, 4 * float(beams[-1] - beams[0]) / max(1, H - 1))      # 16 columns x 4 rows per tile
        cut = lambda shares: comm.broadcast(lidargs_dist.shell_edges(st["means3D"], st["viewmatrix"], world, 0, 80, scales=st["scales"],
                                                                     tile_rad=tile_rad, shares=shares), 0)
        rast = lidargs_dist.ShellRasterizer(settings, comm, edges=cut(None))

        def step():
            for t in list(leaves.values()) + [means2D]:
                t.grad = None
            color, depth, occ, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                            colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
            torch.autograd.backward([color, depth, occ], [gc, gd, go])

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    _C.profile_enable(True)             # pre-creates the event pool (one-off cost, outside the timed region)
    if world > 1 or force_shells:
        # Measured load balancing of the (static) cut, outside the timed region: a few frames per round, every rank's own kernel
        # time (its HIP-event stage sums, which do not include waiting for the other ranks) -> thinner shells for the slow ranks.
        import torch.distributed as dist
        shares = [1.0 / world] * world
        for _round in range(4):
            _C.profile_enable(True)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            mine = torch.tensor([sum(v[0] for v in _C.profile_summary().values())], dtype=torch.float64, device=dev)
            times = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(times, mine)
            shares = lidargs_dist.rebalance_shares(shares, [float(t) for t in times], fixed=0.25)
            rast.edges = cut(shares)
        _C.profile_enable(True)
    clock_ramp(step, fixed_steps=300 if (world > 1 or force_shells) else None)
    for _ in range(args.warmup):
        step()
    barrier()
    _flush_c_stdio()                    # every rank: whatever the collectives' bring-up printed goes out now, not at exit
    # stage events live inside the timed region, on every STAGE_EVERY-th frame: recording all twelve of them on every frame costs
    # 55 us of device time per frame (measured: 0.94 ms with, 0.88 ms without), which would be the harness, not the path
    _C.profile_enable(STAGE_EVERY)
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    _C.profile_enable(False)
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    allocs1 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    stages = _C.profile_summary()
    spread = None
    if world == 1 and not force_shells:
        # per-frame distribution (median and p10 / p90), after and outside the contract's timed region: one event per
        # frame on torch's current stream, which is the stream the op launches on
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        ev[0].record()
        for i in range(args.steps):
            step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        per = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)])
        spread = {"p10": float(np.percentile(per, 10)), "median": float(np.median(per)), "p90": float(np.percentile(per, 90))}
    cnt = _C.last_counters()
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        N_pix, T_ref = H * W, H * ((W + 15) // 16)
        fwd_b, bwd_b, k7_b, k8_b = algorithmic_bytes(P, cnt["V"], cnt["R_ref"], N_pix, T_ref)
        ms = lambda name: stages.get(name, (0.0, 0))[0]
        k7_ms = ms("render_pass1") + ms("render_pass2") + ms("render_combine")
        k8_ms = ms("render_bwd")
        # the dominant kernel of the frame: the forward blend (3 launches: pass 1, pass 2, combine) or the backward blend
        if k7_ms >= k8_ms:
            dom, blend_b, blend_ms = "k_render_forward<T-only> x2 + k_render_alive + k_render_forward + k_render_combine (reference K7)", k7_b, k7_ms
            # pass 1 runs as two gated rounds (two launches of the T-only kernel) with k_render_alive between them
            traffic = pmc_traffic([("lg::k_render_forward<true>", 2), "lg::k_render_alive", "lg::k_render_forward<false>",
                                   "lg::k_render_combine"])
        else:
            dom, blend_b, blend_ms = "k_render_backward (reference K8)", k8_b, k8_ms
            traffic = pmc_traffic(["lg::k_render_backward"])
        achieved = blend_b / (blend_ms * 1e-3) / 1e9 if blend_ms > 0 else 0.0
        out = {
            "metric": "LiDAR range-view frames/sec (fwd+bwd)", "value": args.steps / elapsed, "unit": "frames/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {P} Gaussians ({kind} scene, seed {seed}) @ {H}x{W}, fwd+bwd, "
                                   f"lidar_far=80 lidar_near=0, bg=0",
                       "visible_gaussians": cnt["V"], "instances_binned": cnt["instances"], "R_ref_16x1": cnt["R_ref"],
                       "tile_rows": cnt["tile_rows"], "sharding": "single GPU" if world == 1 else f"{world} range shells"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_GBs": (traffic / (blend_ms * 1e-3) / 1e9) if traffic and blend_ms > 0 else None,
                         "note": "achieved = the reference data flow's bytes (SURVEY 8d formula) / measured time, so frac > 1 means the "
                                 "frame beats what that data flow could do at HBM peak; the bytes this kernel really moves are "
                                 "`traffic` (PMC, several times fewer: pruned instances, flagged entries, gated segments) and the "
                                 "kernel is VALU-issue bound (DESIGN.md section 4)",
                         "algorithmic_bytes_per_launch": blend_b, "kernel_ms": blend_ms,
                         "frame_algorithmic_bytes": fwd_b + bwd_b,
                         "frame_achieved_GBs": (fwd_b + bwd_b) / (ms_per_step * 1e-3) / 1e9},
            "stage_ms": {k: round(v[0], 4) for k, v in stages.items()},
            "stage_events": f"HIP events on the op's stream, on every {STAGE_EVERY}th frame of the timed region",
            "frame_ms_spread": spread,
            "hipmalloc_calls_in_timed_region": int(allocs1 - allocs0),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(kind, P, H, W, seed)
        elif world == 1:
            out["cpu_baseline"] = None
        _flush_c_stdio()
        print(json.dumps(out), flush=True)
    if world > 1 or force_shells:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
