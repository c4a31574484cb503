"""bench.py -- LiDAR range-view frames/s (forward+backward) of the MI355X-native rasterizer.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3] [--shard both|shells|wedges] [--graph] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts its own N ranks, one process per GPU, by re-executing
itself under torch.distributed.run on 127.0.0.1; with fewer than N devices it says so ("needs N HIP devices, found M").

One "step" = one frame: GaussianRasterizer.forward + .backward with non-zero upstream gradients on
colour, depth and occupancy, inputs already resident in HBM (BASELINE.json metric).
N = 1: the whole scene on one GPU.  N > 1: the SAME scene (strong scaling) cut BOTH ways, one after the other in the same job
(lidargs_dist.py): range shells -- the north star's cut: RCCL all-gather of the per-shell transmittance plane, all-gather of the
W x H x 5 partial planes, all-to-all of the packed per-Gaussian gradient rows -- and column wedges (every rank renders its own pixel
columns: image all-gather + gradient all-to-all).  `value` is the faster cut's (named in config.sharding), `cuts` holds both.
`--workload cfg1` (a 10 k-Gaussian frame, host-bound when run eagerly) carries the same frame as a HIP-graph replay in `graph_replay`.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     for the longest single launch of the frame (k_render_backward on the headline workload): ALGORITHMIC bytes of
               THAT launch in its own units -- 68 B per instance this frame binned (R') + 24 B per pixel + 84 B per visible
               Gaussian, SURVEY.md 8d's K8 on R' -- / its mean duration from HIP events on the op's own stream inside the timed
               region, vs 8 TB/s HBM (0 < frac <= 1 by construction); `traffic` = PMC bytes and `compute` = VALU issue rate from
               the committed profile OF THE SAME WORKLOAD; `kernels` = the same for every other launch group of the frame;
               `vs_reference_dataflow` = the reference data flow's bytes (on its R_ref 16x1 instances) against our frame time.
  cpu_baseline the CPU oracle (oracle/lidargs_oracle.c) on a bounded sample of the same workload, timed on this box's host
               cores: one thread, all cores (one frame per core), the numpy projector restatement, and K1 as torch-CPU ops.
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
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU issue roofs, MEASURED on the MI355X (tools/micro/valu_rate2.hip -> profiles/r03_valu_rate2.json, wall clock, DVFS included, >= 4
# waves per SIMD): a wave64 instruction issues at one of three rates by opcode class --
#   full     v_add/sub/mul/fma/fmac_f32, v_mov, v_and/or/xor, v_lshrrev, v_add/sub_u32, v_bitop3 with VGPR / constant operands
#   half     the same with an SGPR source, v_max/min/med3, conversions, compares, v_cndmask, DPP forms, v_pk_*_f32, f64, integer multiplies
#   quarter  v_exp/log/rcp/rsq/sqrt/sin/cos_f32, v_permlane32_swap
# (round 2 priced everything at 256 CU x 4 SIMD x 2.4 GHz / 4 clocks = 614 G/s; the guide's "2 clocks per wave64 v_fma_f32" is the
# `full` class only, and a lone wave per SIMD issues one instruction of ANY class per 4-5 clocks).  A launch is priced with the class
# shares of its hot loop (tools/isa_mix.py -> profiles/r03_isa_mix.json): time at the roof = insts x sum_c share_c / rate_c.
VALU_CLASS_RATES = {"full": 1100.0, "half": 580.0, "quarter": 300.0}      # G wave-instructions/s; overwritten from the committed profile below


def _load_valu_rates():
    """Class rates from profiles/r03_valu_rate2.json (median of the class's representative opcodes at 4 waves per SIMD)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*valu_rate2*.json")))
    if not files:
        return None
    try:
        j = json.load(open(files[-1]))
        by = {r["op"]: r["w4"]["ginst_per_s"] for r in j["results"]}
        med = lambda names: float(np.median([by[n] for n in names if n in by]))
        VALU_CLASS_RATES.update(full=med(["v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32", "v_mov_b32", "v_and_b32", "v_add_u32"]),
                                half=med(["v_max_f32", "v_cndmask_sgpr_mask", "v_cmp_gt_f32_vcc", "v_pk_fma_f32", "v_mov_dpp_row_ror", "v_cvt_f32_u32", "v_fma_f32_sgpr"]),
                                quarter=med(["v_exp_f32", "v_rsq_f32", "v_sin_f32"]))
        return os.path.basename(files[-1])
    except Exception:
        return None


VALU_RATES_FROM = _load_valu_rates()


_ISA_MIX = None


def _isa_mix():
    """The committed instruction-mix profile to price VALU issue with: the one stamped with THIS build's id (tools/isa_mix.py, build_hip.build_id())
    when there is one, the newest otherwise -- and which of the two it was (round-5 verdict item 13: round 5 priced its rewritten walks on round 3's mix)."""
    global _ISA_MIX
    if _ISA_MIX is None:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*isa_mix*.json")))
        best, same = None, False
        try:
            import build_hip
            bid = build_hip.build_id()
        except Exception:
            bid = None
        for f in files:
            try:
                j = json.load(open(f))
            except Exception:
                continue
            if best is None or not same:
                if bid is not None and j.get("build_id") == bid:
                    best, same = (j, os.path.basename(f)), True
                elif not same:
                    best = (j, os.path.basename(f))
        _ISA_MIX = (best[0], best[1], same, bid) if best else (None, None, False, bid)
    return _ISA_MIX


def valu_class_shares(kernel):
    """(shares {full, half, quarter}, where from) of a kernel's VALU instructions: its hot loop's mix from the committed
    profiles/*isa_mix*.json (`whole` = the whole kernel, for the one-thread-per-Gaussian streams that have no hot loop)."""
    j, name, _same, _bid = _isa_mix()
    if j is None:
        return None, None
    for k, v in j["kernels"].items():
        if k == kernel or k.startswith(kernel + "<") or kernel.startswith(k):
            m = v.get("whole") if v.get("use") == "whole" else v.get("hot")
            if not m:
                continue
            tot = float(m["full"] + m["half"] + m["quarter"]) or 1.0
            return {c: m[c] / tot for c in ("full", "half", "quarter")}, name
    return None, name


def valu_roof(kernel_names, insts, ms):
    """Class-weighted VALU-issue roofline of a launch (group): {achieved G/s, peak = the mix's mean rate, frac}."""
    shares, src = None, None
    for k in kernel_names:
        shares, src = valu_class_shares(k.replace("lg::", "lg::") if k.startswith("lg::") else "lg::" + k)
        if shares:
            break
    if not shares:
        shares, src = {"full": 0.4, "half": 0.55, "quarter": 0.05}, "no ISA profile: a typical blend mix assumed"
    mean_rate = 1.0 / sum(shares[c] / VALU_CLASS_RATES[c] for c in shares)
    g = insts / (ms * 1e-3) / 1e9
    return {"bound": "valu-issue (class-weighted)", "achieved": g, "peak": mean_rate, "unit": "G wave-instructions/s", "frac": g / mean_rate,
            "valu_insts_per_launch": insts, "class_shares": {c: round(v, 3) for c, v in shares.items()},
            "class_rates": dict(VALU_CLASS_RATES), "rates_from": VALU_RATES_FROM, "shares_from": src,
            "shares_build": (_isa_mix()[0] or {}).get("build_id"), "shares_same_build": bool(_isa_mix()[2])}


def reference_dataflow_bytes(P, V, R_ref, N, T):
    """SURVEY.md 8d compact form, priced on the REFERENCE's 16x1 instances (R_ref): (fwd, bwd, K7, K8) bytes per frame.  Not a
    roofline for our launches (they process R' << R_ref instances); reported as `vs_reference_dataflow`."""
    fwd = 48 * P + 112 * V + 112 * R_ref + 8 * T + 24 * N
    bwd = 112 * P + 192 * V + 68 * R_ref + 24 * N
    return fwd, bwd, 68 * R_ref + 24 * N, 68 * R_ref + 24 * N + 84 * V


def raster_kernel_table(P, V, R, N, stages, surfel=False, taken=None, tile_key_bytes=2, touched=None, bwd_entries=None):
    """ALGORITHMIC bytes of each launch in ITS OWN units: R = the instances this frame binned (R'),
    V = visible Gaussians, N = pixels.  rec = bytes gathered per list entry (id 4 + record 64/80 + row span 4), pix = per-pixel
    planes.  `stage` = the lidargs_profile stage that brackets the launch(es); `launches` = launches inside that stage.
    The FORWARD blends are priced on `taken` = the (16x4 patch, instance) pairs some pixel really takes (the contribution flags pass 1
    writes, counted on the host after the timed region): every correct blend must fetch each of them once, while the
    instances BEHIND a patch's saturation point are never needed -- pricing those (R') made a launch that rightly skips them
    look faster than the memory system (r02_a: cfg4 1.9 'of peak').  Round 5: the BACKWARD blend is priced on `bwd_entries` = the list
    entries it gathers (per patch and segment the flagged entries in front of the last blended one: lidargs_last_counters[9], counted by
    a diagnostic launch of the same selection) -- pass 1's flags restart from T = 1 in every segment and are a superset three times that
    size (round 4: the launch's counter traffic was 0.58 of bytes priced on them) -- and on `touched` = the Gaussians whose packed
    gradient line it can add to; the per-Gaussian backward likewise on the touched ones plus the zero rows of everybody."""
    Rb = taken if taken else R                      # no flags (single-segment frames): every binned instance is walked
    Eb = bwd_entries if (bwd_entries and bwd_entries > 0) else Rb
    Tg = touched if (touched is not None and touched >= 0) else V
    rec = 88 if surfel else 68                                  # 3-D: SURVEY 8d's 68 B/instance (id + 64-B record; the row span rides in it)
    pix_f = 56 if surfel else 24                                # surfel: 2 + 7 output planes, 3 accum planes, 2 count planes
    line = 128 if surfel else 64                                # the packed gradient line the backward blend adds into, per touched Gaussian
    pin = 40 if surfel else 44
    rows = 68                                                   # gradient rows every Gaussian receives (3-D: 3+4+2+1+3+4 floats; surfel: 4+4+3+2+1+2 + the depth)
    kb = tile_key_bytes                                         # 2 when the frame has <= 65536 tiles (every BASELINE size), else 4
    bucketed = 4096 < P <= (4 << 20)                            # binning.hip range_sort_buckets_ok
    t = [
        dict(kernel="k_sf_preprocess" if surfel else "k_preprocess", stage="preprocess", launches=1, bound="hbm",
             bytes=(pin + 37) * P + (88 if surfel else 76) * V,
             units=f"{pin} B in + 37 B (radii, radii_xy, key, id, spans, the touched mark) out per Gaussian, + record / row span / colours per visible one"),
        dict(kernel="range sort of the Gaussians (" + ("one linear-bucket pass: hist + prefix + scatter, + one launch that sorts every bucket in LDS and gathers the span records" if bucketed
                                                        else "hist + prefix + scatter per pass; the last pass gathers the span records") + ")", stage="range_sort",
             launches=4 if bucketed else "3 per pass", bound="hbm", bytes=(36 * P) if bucketed else (4 * 20 * P + 8 * P),
             units=("bucket pass: 4 B key read by the histogram + 4 B key read + 8 B pair written; bucket sort: 8 B pair read, 4 B id + 4 B span record written, 4 B span gathered = 36 B per Gaussian" if bucketed
                    else "per pass 4 B key read by the histogram + 8 B pair read + 8 B pair written per Gaussian (priced at 4 passes), + the span gather of the last one")),
        dict(kernel="span block sums + scan of the block sums (+ the 2-KB totals read-back)", stage="scan+readback", launches=2, bound="hbm", bytes=4 * P,
             units="4 B span record per Gaussian read in range order"),
        dict(kernel="k_emit_instances", stage="emit", launches=1, bound="hbm", bytes=8 * P + (kb + 4) * R,
             units=f"4 B span + 4 B id per Gaussian in, {kb} B tile key + 4 B id per instance out"),
        dict(kernel="tile sort of the instances (hist + prefix + scatter per pass)", stage="tile_bin", launches="3 per pass", bound="hbm", bytes=2 * (3 * kb + 8) * R,
             units=f"2 passes x ({kb} B key read by the histogram + {kb + 4} B pair read + {kb + 4} B pair written) per instance"),
        dict(kernel="k_tile_ranges", stage="ranges", launches=1, bound="hbm", bytes=kb * R + 8 * max(1, N // 64),
             units=f"{kb} B tile key per instance in, 8 B range per tile out"),
        dict(kernel="forward blend group (reference K7): T-only walks + alive + full walk + combine", stage=("render_pass1", "render_pass2", "render_combine"),
             launches="4-7 by plan", bound="hbm", bytes=rec * Rb + pix_f * N, units=f"{rec} B per taken (patch, instance) pair + {pix_f} B per pixel (SURVEY 8d K7 on what the frame takes)"),
        dict(kernel="k_zero_touched", stage="bwd_zero", launches=1, bound="hbm", bytes=(2 + rows) * P + line * Tg,
             units=f"1 B mark in + 1 B list out + the {rows} B of gradient rows zeroed per Gaussian, + the {line}-B packed line cleared per touched one"),
        dict(kernel="k_sf_render_backward" if surfel else "k_render_backward", stage="render_bwd", launches=1, bound="hbm",
             bytes=rec * Eb + pix_f * N + 2 * line * Tg, units=f"{rec} B per list entry the launch gathers + {pix_f} B per pixel + the {line}-B packed line read and written per touched Gaussian (SURVEY 8d K8 on what the frame blends)"),
        dict(kernel="k_sf_gaussian_backward" if surfel else "k_gaussian_backward", stage="gaussian_bwd", launches=1, bound="hbm",
             bytes=(pin + line + rows + 1) * Tg + ((4 + 12 + 4) * P if surfel else 0),
             units=f"per touched Gaussian: {pin} B inputs + the {line}-B packed line in, {rows} B of gradient rows out" + (" ; per surfel: radius + centre in, planar depth out" if surfel else "")),
    ]
    ms = lambda st: sum(stages.get(x, (0.0, 0))[0] for x in (st if isinstance(st, tuple) else (st,)))
    out = []
    for k in t:
        k = dict(k)
        k["ms"] = ms(k["stage"])
        if k["ms"] <= 0:
            continue
        k["achieved_GBs"] = k["bytes"] / (k["ms"] * 1e-3) / 1e9
        k["frac"] = k["achieved_GBs"] / HBM_PEAK_GBS
        k["stage"] = "+".join(k["stage"]) if isinstance(k["stage"], tuple) else k["stage"]
        out.append(k)
    return out


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


def committed_profile(kind, workload):
    """The newest committed profiles/*_pmc_<kind>*.json taken on THIS workload (its "workload" field; files of round 1 carry none
    and were all taken on cfg3).  kind = "traffic" (FETCH_SIZE / WRITE_SIZE passes) or "sq" (SQ_* passes).  None if there is none."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_{kind}*.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("workload", "cfg3") == workload:
            best = (f, j)
    return best


def profile_provenance(kind, workload, basename):
    """Where a looked-up counter value comes from: the profile file, the build (a content hash of csrc/ + the headers: build_hip.build_id,
    stamped by the profile tools) and box it was taken on, and this run's build -- a line priced on a profile of another build says so."""
    prof = committed_profile(kind, workload)
    if prof is None:
        return basename
    f, j = prof
    here = None
    try:
        import build_hip
        here = build_hip.build_id()
    except Exception:
        pass
    return {"file": os.path.basename(f), "profile_build_id": j.get("build_id"), "box": j.get("box"), "this_build_id": here,
            "same_build": (j.get("build_id") == here) if (here and j.get("build_id")) else None}


def pmc_lookup(kind, workload, kernel_names, field):
    """Sum of `field` over (kernel, launches) pairs from the committed profile of this workload; (None, None) if any is missing."""
    prof = committed_profile(kind, workload)
    if prof is None:
        return None, None
    f, j = prof
    tot = 0
    if isinstance(kernel_names, dict):
        # a launch GROUP whose members depend on the frame's segment plan: every kernel of the profile whose name starts with one of
        # `any_of`, each weighted by its launches per frame = launches sampled / launches sampled of `per_frame` (one per frame)
        ref = [v for k, v in j["kernels"].items() if kernel_names["per_frame"] in k]
        if not ref or not ref[0].get("launches_sampled"):
            return None, os.path.basename(f)
        frames = ref[0]["launches_sampled"]
        found = False
        for k, v in j["kernels"].items():
            base = k[5:] if k.startswith("void ") else k
            if any(base.startswith(pre) for pre in kernel_names["any_of"]) and field in v:
                # `weight(name)`: the share of this kernel's launches that belong to the group (kernels two groups share, e.g. the
                # radix sort's digit prefix, used by the range sort and the tile sort alike)
                wgt = kernel_names["weight"](base) if "weight" in kernel_names else 1.0
                if wgt <= 0:
                    continue
                tot += wgt * v[field] * v.get("launches_sampled", frames) / frames
                found = True
        return (tot if found else None), os.path.basename(f)
    for name in kernel_names:
        name, times = name if isinstance(name, tuple) else (name, 1)
        hit = [v for k, v in j["kernels"].items() if k == name or k.startswith(name + "<") or k == "void " + name]
        if not hit or field not in hit[0]:
            return None, os.path.basename(f)
        tot += times * hit[0][field]
    return tot, os.path.basename(f)


def sort_pmc_groups(tiles):
    """PMC lookup groups of the two radix sorts of a frame.  Their kernels share names: hist / scatter launches are told apart by
    their digit width (the range sort's passes are >= 7 bits wide, the tile sort's ceil(log2 tiles) bits split evenly are narrower on
    every BASELINE image size), the digit-prefix launches (no width in the name) are shared out by pass count."""
    import math
    import re
    tb = max(1, math.ceil(math.log2(max(2, tiles))))
    t_passes = (tb + 7) // 8
    t_bits = (tb + t_passes - 1) // t_passes
    r_passes = 1                                           # the bucketed range sort: one digit pass (frames above 4 M Gaussians run more)
    sort_kernels = ["lg::k_radix_hist<", "lg::k_radix_scatter<", "lg::k_radix_digit_prefix<", "lg::k_radix_chunk_prefix", "lg::k_bucket_sort"]

    def weight(for_tile_sort):
        def w(name):
            if "k_bucket_sort" in name:                    # the bucketed range sort's second launch
                return 0.0 if for_tile_sort else 1.0
            m = re.search(r"k_radix_(?:hist|scatter)<(\d+),", name)
            if m is None:                                  # digit / chunk prefix: shared by pass count
                return (t_passes if for_tile_sort else r_passes) / float(t_passes + r_passes)
            narrow = int(m.group(1)) <= t_bits and t_bits < 7
            if t_bits >= 7:
                return 0.5                                 # widths overlap: cannot be told apart by name
            return 1.0 if narrow == for_tile_sort else 0.0
        return w
    rs = {"any_of": sort_kernels, "per_frame": "k_preprocess", "weight": weight(False)}
    return {"range sort of the Gaussians (hist + prefix + scatter per pass; the last pass gathers the span records)": rs,
            "range sort of the Gaussians (one linear-bucket pass: hist + prefix + scatter, + one launch that sorts every bucket in LDS and gathers the span records)": rs,
            "tile sort of the instances (hist + prefix + scatter per pass)":
                {"any_of": sort_kernels, "per_frame": "k_preprocess", "weight": weight(True)},
            "span block sums + scan of the block sums (+ the 2-KB totals read-back)":
                {"any_of": ["lg::k_span_block_sums", "lg::k_scan_partials"], "per_frame": "k_preprocess"}}


def host_cores():
    """(usable cores, description): the smaller of the hardware threads, the affinity mask and the cgroup CPU quota -- the GPU
    boxes show 256 hardware threads but give the container a 16-CPU quota, and 256 busy threads on 16 CPUs time nothing."""
    n = os.cpu_count() or 1
    why = f"{n} hardware threads"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, f"affinity mask of {a} CPUs"
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period))
                if q < n:
                    n, why = q, f"cgroup CPU quota of {q} (of {os.cpu_count()} hardware threads)"
            break
        except Exception:
            continue
    return n, why


def cpu_baseline(kind, P_full, H, W, seed, fwd_only=False, surfel=False, budget_s=24.0):
    """The CPU legs of SURVEY 8d on this box's host cores, on a bounded 1/20 sample of the workload (same image size):
      (1) the oracle (oracle/lidargs_oracle.c | lidargs_surfel_oracle.c), one thread -> `value`;
      (1b) the same on ALL host cores, one independent frame per core (frames are independent, the C port is single-threaded);
      (2) the reference's numpy per-point projector restated (oracle/range_view.py), one core -> projector_points_per_s;
      (3) K1 as vectorised torch-CPU ops with torch.set_num_threads(nproc) (oracle/preprocess_torch.py)."""
    import lidargs_scenes as sc
    from concurrent.futures import ThreadPoolExecutor
    from oracle import lgo, lgo_surfel
    lgo.build()
    P = max(1000, P_full // 20)
    scene = sc.make_scene(kind, P, H, seed)
    if surfel:
        scene["scales"] = np.ascontiguousarray(scene["scales"][:, :2])
        rng = np.random.default_rng(seed + 200)
        grads = (rng.normal(size=(2, H, W)).astype(np.float32), rng.normal(size=(7, H, W)).astype(np.float32))
    else:
        grads = sc.upstream_grads(H, W, seed)

    def frame():
        if surfel:
            f = lgo_surfel.forward(scene["means3D"], scene["colors"], scene["opacities"], scene["scales"], scene["rotations"],
                                   scene["viewmatrix"], scene["beams"], W, H, bg=scene["bg"])
            if not fwd_only:
                lgo_surfel.backward(f, *grads)
        else:
            f = lgo.forward(scene["means3D"], scene["colors"], scene["opacities"], scene["scales"], scene["rotations"],
                            scene["viewmatrix"], scene["beams"], W, H, bg=scene["bg"])
            if not fwd_only:
                lgo.backward(f, *grads)

    frames, t0 = 0, time.perf_counter()
    while True:
        frame()
        frames += 1
        el = time.perf_counter() - t0
        if el > budget_s * 0.25 or frames >= 8:
            break
    fps = frames / el
    # (1b) every host core: `cores` POSIX threads inside the oracle library (oracle/lgo_bench.c), each rendering its own frames
    cores, cores_why = host_cores()
    per_core = 2 if el / frames < 1.0 else 1
    el_all = lgo.bench_frames(scene, W, H, grads, cores, per_core, fwd_only=fwd_only, surfel=surfel)
    fps_all = cores * per_core / el_all
    what = "forward" if fwd_only else "fwd+bwd"
    out = {
        "value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{frames} {what} frames of a {P}-{'surfel' if surfel else 'Gaussian'} (1/20) {kind} scene at {H}x{W}, "
                  f"oracle/{'lidargs_surfel_oracle.c' if surfel else 'lidargs_oracle.c'}, 1 thread of {cores} usable host cores ({cores_why}); "
                  f"linear-in-P estimate for the full workload: {fps * P / P_full:.4f} frames/s",
        "all_cores": {"value": fps_all, "unit": "frames/s", "cores": cores, "cores_note": cores_why,
                      "sample": f"{cores * per_core} independent {what} frames of the same 1/20 scene, {cores} POSIX threads "
                                f"x {per_core} frame(s) each (oracle/lgo_bench.c): {el_all:.1f} s; linear-in-P estimate for the full workload: "
                                f"{fps_all * P / P_full:.3f} frames/s"},
    }
    if not surfel:
        # (2) baseline B2: the reference's numpy per-point projector on one sweep's worth of points
        from oracle import range_view
        pts = np.concatenate([scene["means3D"][:20000], scene["colors"][:20000, :1]], 1)
        t2 = time.perf_counter()
        range_view.points_to_pano(pts, H, W, scene["beams"])
        out["projector_points_per_s"] = pts.shape[0] / (time.perf_counter() - t2)
        # (3) baseline B3: K1 as torch-CPU ops, intra-op threads = the usable host cores (and 32, in case oversubscription hurts);
        #     full Gaussian count when one call is estimated under 4 s, else the 1/20 sample scaled linearly
        from oracle import preprocess_torch as pt
        small = {k: torch.from_numpy(scene[k]) for k in ("means3D", "scales", "rotations", "viewmatrix", "beams")}
        full = None
        old = torch.get_num_threads()
        best = None
        try:
            for nt in sorted({cores, min(cores, 32)}, reverse=True):
                torch.set_num_threads(nt)
                run = lambda d: pt.preprocess(d["means3D"], d["scales"], d["rotations"], d["viewmatrix"], d["beams"], W, H)
                run(small)
                t3 = time.perf_counter(); run(small); t_small = time.perf_counter() - t3
                if t_small * (P_full / P) < 4.0:
                    if full is None:
                        fs = sc.make_scene(kind, P_full, H, seed)
                        full = {k: torch.from_numpy(fs[k]) for k in ("means3D", "scales", "rotations", "viewmatrix", "beams")}
                    run(full)
                    t3, reps = time.perf_counter(), 0
                    while time.perf_counter() - t3 < 2.0 and reps < 10:
                        run(full); reps += 1
                    rec = {"ms": (time.perf_counter() - t3) / reps * 1e3, "threads": nt, "measured_on": f"all {P_full} Gaussians"}
                else:
                    rec = {"ms": t_small * (P_full / P) * 1e3, "threads": nt, "measured_on": f"{P} Gaussians (1/20), scaled linearly"}
                if best is None or rec["ms"] < best["ms"]:
                    best = rec
        finally:
            torch.set_num_threads(old)
        best["what"] = "K1 of the workload's Gaussians as vectorised torch-CPU ops (oracle/preprocess_torch.py), best of intra-op thread counts {nproc, 32}"
        out["torch_cpu_preprocess"] = best
    return out


def roofline_object(table, workload, blend_kernels_pmc, ref_flow=None):
    """The contract's `roofline`: the launch with the longest mean duration (a single kernel, so that the rocprofv3 kernel-trace
    CSV next to the bench line can be checked against it), priced on ITS OWN units; every other launch group in `kernels`."""
    single = [k for k in table if k["launches"] == 1]
    dom = max(single, key=lambda k: k["ms"])
    roof = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["frac"], "algorithmic_bytes_per_launch": dom["bytes"], "kernel_ms": dom["ms"], "units": dom["units"],
            "traffic": None, "traffic_GBs": None, "traffic_profile": None}
    names = blend_kernels_pmc.get(dom["kernel"])
    if names:
        tr, src = pmc_lookup("traffic", workload, names, "hbm_bytes_per_launch_corrected")
        roof["traffic"], roof["traffic_profile"] = tr, profile_provenance("traffic", workload, src)
        if tr:
            roof["traffic_GBs"] = tr / (dom["ms"] * 1e-3) / 1e9
            roof["frac_by_counters"] = roof["traffic_GBs"] / HBM_PEAK_GBS      # counter bytes of the committed profile / THIS run's launch time / peak
            roof["traffic_over_algorithmic"] = tr / dom["bytes"]
        insts, src2 = pmc_lookup("sq", workload, names, "SQ_INSTS_VALU")
        if insts:
            roof["compute"] = valu_roof(names if isinstance(names, list) else names["any_of"], insts, dom["ms"])
            roof["compute"]["profile"] = src2
            roof["compute"]["note"] = ("SQ_INSTS_VALU of the committed profile / this run's launch time against the launch's class-weighted issue "
                                       "roof (measured class rates x the hot loop's class shares).  The nearer roof of the two is the one to read.")
            roof["nearer_roof"] = "valu-issue" if roof["compute"]["frac"] > roof["frac"] else "hbm"
    # every launch (group) the committed profiles of this workload cover: PMC traffic and the class-weighted VALU-issue fraction
    for k in table:
        names = blend_kernels_pmc.get(k["kernel"])
        if names:
            tr, _src = pmc_lookup("traffic", workload, names, "hbm_bytes_per_launch_corrected")
            if tr:
                k["traffic"] = tr
                k["traffic_over_algorithmic"] = tr / k["bytes"]
                k["frac_by_counters"] = tr / (k["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            insts, _src = pmc_lookup("sq", workload, names, "SQ_INSTS_VALU")
            if insts:
                vr = valu_roof(names if isinstance(names, list) else names["any_of"], insts, k["ms"])
                k["valu_insts"] = insts
                k["valu_issue_frac"] = vr["frac"]
                k["valu_issue_peak"] = vr["peak"]
    roof["kernels"] = [{k2: (round(v, 5) if isinstance(v, float) else v) for k2, v in k.items()} for k in table]
    if ref_flow is not None:
        roof["vs_reference_dataflow"] = ref_flow
    return roof


def bench_surfel(args, sc, kind, P, H, W, seed):
    """BASELINE config 5: the 2DGS laser-surfel variant, single GPU, same metric (fwd+bwd frames/s); not the headline line."""
    assert args.gpus == 1, "config 5 is a single-GPU configuration"
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU path"
    import build_hip
    build_hip.build()
    from diff_lidargs_rasterization import _C as base_C
    from diff_lidargs_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda", 0)
    scene = sc.make_scene(kind, P, H, seed, opacity_scale=args.opacity_scale)
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
    info = {}

    def step():
        for t in list(leaves.values()) + [means2D]:
            t.grad = None
        color, radii, others, _pix = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                          colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
        torch.autograd.backward([color, others], [gc, go])
        return radii
    base_C.profile_enable(True)
    clock_ramp(step)
    for _ in range(args.warmup):
        radii = step()
    torch.cuda.synchronize()
    base_C.profile_enable(STAGE_EVERY)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        radii = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    base_C.profile_enable(False)
    stages = base_C.profile_summary()
    V = int((radii > 0).sum())
    base_C.counters_enable(True)            # one more frame, outside the timed region, that ends with the counting launches
    step(); torch.cuda.synchronize()
    cnt = base_C.last_counters()
    base_C.counters_enable(False)
    info["R"] = int(cnt["instances"])
    table = raster_kernel_table(P, V, info["R"], H * W, stages, surfel=True, taken=int(cnt["taken_instances"]), touched=int(cnt.get("touched", -1)))
    pmc_names = {"k_sf_render_backward": ["lg::k_sf_render_backward"], "k_sf_preprocess": ["lg::k_sf_preprocess<false>"],
                 "k_sf_gaussian_backward": ["lg::k_sf_gaussian_backward"], "k_zero_touched": ["lg::k_zero_touched"],
                 "forward blend group (reference K7): T-only walks + alive + full walk + combine":
                     {"any_of": ["lg::k_sf_render_forward<", "lg::k_sf_alive", "lg::k_sf_combine"], "per_frame": "k_sf_combine"},
                 "k_emit_instances": ["lg::k_emit_instances"], "k_tile_ranges": ["lg::k_tile_ranges"]}
    pmc_names.update({k: dict(v, per_frame="k_sf_preprocess") for k, v in sort_pmc_groups(((W + 15) // 16) * ((H + 3) // 4)).items()})
    out = {
        "metric": "LiDAR range-view frames/sec (fwd+bwd)", "value": args.steps / elapsed, "unit": "frames/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"cfg5: {P} surfels ({kind} scene, seed {seed}) @ {H}x{W}, fwd+bwd, diff_lidargs_surfel_rasterization "
                               f"(2DGS laser-surfel variant), lidar_far=80 lidar_near=0, bg=0"
                               + ("" if args.opacity_scale == 1.0 else f", opacities x {args.opacity_scale:g} (semi-transparent, early-training regime)"),
                   "visible_surfels": V, "instances_binned": info.get("R", 0), "patch_instance_pairs_taken": int(cnt["taken_instances"]), "touched_surfels": int(cnt.get("touched", -1)), "tile_rows": 4},
        "roofline": roofline_object(table, "cfg5", pmc_names),
        "stage_ms": {k: round(v[0], 4) for k, v in stages.items()},
        "stage_events": f"HIP events on the op's stream, on every {STAGE_EVERY}th frame of the timed region",
        "clock_ramp": "0.5 s of untimed frames ran before the warm-up steps (first frames of a fresh process run 5-8 % slow)"}
    out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(kind, P, H, W, seed, surfel=True)
    print(json.dumps(out))


def bench_render_fps(args):
    """The reference's own FPS definition (train.py:408-414, :454): per view, between two device synchronisations,
    prefilter_voxel (K2 visible_filter on the anchors, gaussian_renderer/__init__.py:252-257) + render (anchor decode + rasterizer
    FORWARD), under no_grad.  FPS = 1 / mean(per-view seconds).  667 k anchors x 6 offsets of the 2 M-Gaussian street scene."""
    import types
    assert args.gpus == 1 and torch.cuda.is_available()
    import build_hip
    build_hip.build()
    import lidargs_scenes as sc
    from diff_lidargs_rasterization import GaussianRasterizer, _C
    from neural_gaussians import generate_neural_gaussians
    kind, P, H, W, seed = sc.BASELINE_CONFIGS["cfg3"]
    N, k = 666_667, 6
    scene = sc.make_scene(kind, N, H, seed)
    p, _cam, _vis, rng = sc.make_anchor_model(N, k, seed)
    p["anchor"] = scene["means3D"].astype(np.float32)
    p["offset"] = (0.5 * rng.normal(size=(N, k, 3))).astype(np.float32)
    p["scaling"] = np.concatenate([np.full((N, 3), 0.3, np.float32), scene["scales"].astype(np.float32) * 2.0], 1)
    pc = sc.anchor_model_to_torch(p)
    camera = types.SimpleNamespace(camera_center=torch.zeros(3).cuda(), uid=0)
    st = {k_: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k_, v in scene.items() if k_ in ("viewmatrix", "beams", "bg")}
    rast = GaussianRasterizer(sc.raster_settings(st, W, H))
    anchor_rot = torch.from_numpy(scene["rotations"]).cuda()             # pc.get_rotation (unit quaternions per anchor)
    info = {}

    def view():
        with torch.no_grad():
            radii_pure = rast.visible_filter(means3D=pc.get_anchor, scales=pc.get_scaling[:, :3], rotations=anchor_rot, cov3D_precomp=None)
            mask = radii_pure > 0                                        # prefilter_voxel
            xyz, color, opacity, scaling, rot = generate_neural_gaussians(camera, pc, mask, is_training=False)
            means2D = torch.zeros((xyz.shape[0], 4), device="cuda")
            image, depth, occ, radii = rast(means3D=xyz, means2D=means2D, opacities=opacity, colors_precomp=color, scales=scaling, rotations=rot)
        info.update(visible_anchors=mask, gaussians=xyz.shape[0], radii=radii)
        return image

    clock_ramp(view)
    for _ in range(args.warmup):
        view()
    per = []
    for _ in range(args.steps):                                           # the reference's own bracket: sync, t, work, sync, t
        torch.cuda.synchronize(); t0 = time.perf_counter()
        view()
        torch.cuda.synchronize(); per.append(time.perf_counter() - t0)
    mean_s = float(np.mean(per))
    # back-to-back (no per-view synchronisation), for comparison with the other lines
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        view()
    torch.cuda.synchronize()
    piped = (time.perf_counter() - t0) / args.steps
    # stage split of one view, events on the current stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = [0.0, 0.0, 0.0]
    for _ in range(5):
        with torch.no_grad():
            ev[0].record()
            mask = rast.visible_filter(means3D=pc.get_anchor, scales=pc.get_scaling[:, :3], rotations=anchor_rot, cov3D_precomp=None) > 0
            ev[1].record()
            xyz, color, opacity, scaling, rot = generate_neural_gaussians(camera, pc, mask, is_training=False)
            ev[2].record()
            rast(means3D=xyz, means2D=torch.zeros((xyz.shape[0], 4), device="cuda"), opacities=opacity, colors_precomp=color, scales=scaling, rotations=rot)
            ev[3].record()
        torch.cuda.synchronize()
        for q in range(3):
            acc[q] += ev[q].elapsed_time(ev[q + 1]) / 5
    n_vis = int(info["visible_anchors"].sum())
    # K2 is an HBM stream: 12 + 12 + 16 B in (anchor, first three scales of a 6-float row -> the whole 24-B row is fetched), 4 + 8 B out
    k2_bytes = (12 + 24 + 16 + 12) * N
    out = {"metric": "rendered range-view frames/sec (prefilter_voxel + decode + rasterize forward, the reference's FPS, train.py:408-414)",
           "value": 1.0 / mean_s, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": mean_s * 1e3,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"render_fps: {N} anchors x {k} offsets ({n_vis} anchors pass visible_filter) -> {info['gaussians']} Gaussians "
                                  f"({int((info['radii'] > 0).sum())} on screen) @ {H}x{W}, no_grad, device synchronised around every view as the reference times it",
                      "back_to_back_ms_per_view": piped * 1e3},
           "stage_ms": {"visible_filter(K2)": acc[0], "generate_neural_gaussians": acc[1], "rasterize_forward": acc[2]},
           "roofline": {"bound": "hbm", "kernel": "k_preprocess<FILTER> (reference K2 filter_preprocessCUDA, R3/cr/forward.cu:388-497)",
                        "achieved": k2_bytes / (acc[0] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": k2_bytes / (acc[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": k2_bytes, "kernel_ms": acc[0],
                        "units": "64 B per anchor: mean 12 + the 24-B scaling row its three scales live in + quaternion 16 in, radii 4 + radii_xy 8 out",
                        "traffic": None, "note": "stage time from events around the Python call (one launch + two fills); the other two stages have their own lines "
                                                 "(--workload decode, --workload cfg2)"},
           "clock_ramp": "0.5 s of untimed views ran before the warm-up steps"}
    if not args.no_cpu_baseline:
        from oracle import lgo
        lgo.build()
        Ns = N // 20
        sm = sc.make_scene(kind, Ns, H, seed)
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < 5.0 and reps < 20:
            lgo.visible_filter(sm["means3D"], sm["scales"], sm["rotations"], sm["viewmatrix"], sm["beams"], W, H)
            f = lgo.forward(sm["means3D"], sm["colors"], sm["opacities"], sm["scales"], sm["rotations"], sm["viewmatrix"], sm["beams"], W, H, bg=sm["bg"])
            reps += 1
        t = (time.perf_counter() - t0) / reps
        out["cpu_baseline"] = {"value": 1.0 / t, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} x (visible_filter + rasterizer forward) of a {Ns}-Gaussian (1/20) scene with oracle/lidargs_oracle.c, 1 thread of "
                                         f"{os.cpu_count()} host cores (the decode is not in this leg: its CPU port is timed by --workload decode); "
                                         f"linear-in-P estimate for the full workload: {1.0 / (t * 20):.4f} frames/s"}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


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

    upstream = {}

    def step(fn):
        # upstream gradients are INPUTS of the step (the rasterizer's backward hands them over, contiguous): made once per output shape.
        # (Until round 3 the step ended in `(xyz.sum() + ...).backward()`: five reductions, four adds and five copies of stride-0
        # gradients, ~0.15 ms of framework launches that belong to no decode.)
        for t in leaves:
            t.grad = None
        outs = fn()[:5]
        key = tuple(tuple(o.shape) for o in outs)
        if key not in upstream:
            g = torch.Generator(device="cuda").manual_seed(11)
            upstream[key] = [torch.randn(o.shape, device="cuda", generator=g) for o in outs]
        torch.autograd.backward(list(outs), upstream[key])
        return outs[0].shape[0]

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            M = step(fn)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            M = step(fn)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, M

    clock_ramp(lambda: step(lambda: generate_neural_gaussians(camera, pc, vmask, is_training=True)))
    t_hip, M = timed(lambda: generate_neural_gaussians(camera, pc, vmask, is_training=True), args.steps, args.warmup)
    n_vis = int(vis.sum())
    # the backward alone (events on the current stream around .backward(): k_ng_backward_mfma + the partial-sum fold + glue)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t_bwd = 0.0
    for _ in range(10):
        for t in leaves:
            t.grad = None
        outs = generate_neural_gaussians(camera, pc, vmask, is_training=True)[:5]
        ups = upstream[tuple(tuple(o.shape) for o in outs)]
        ev[0].record(); torch.autograd.backward(list(outs), ups); ev[1].record()
        torch.cuda.synchronize()
        t_bwd += ev[0].elapsed_time(ev[1]) / 10 * 1e-3
    # matrix-pipe side of the backward (round 5: two launches of k_ng_backward_t16): 788 v_mfma_f32_16x16x4_f32 per 32-anchor tile (272 in
    # the covariance launch, 516 in the heads'; counters: 16.4 M per step at 666 k anchors), 2048 flop each, against the 157.3 TFLOP/s
    # dense f32 MFMA peak (MI355X_MICROARCH.md)
    mfma_insts = ((n_vis + 31) // 32) * 788
    mfma_flop = mfma_insts * 2048.0
    # algorithmic bytes: inputs once per visible anchor + the bool mask, outputs once, and the same again (+ upstream gradients,
    # dense input gradients) for the backward; the per-anchor activations the weight-gradient GEMMs read are NOT counted
    fwd_b = N + n_vis * (128 + 12 + 12 * k + 24) + n_vis * k * 5 + M * 52
    bwd_b = n_vis * (128 + 12 + 12 * k + 24) + M * 52 + N * (128 + 12 + 12 * k + 24)
    out = {"metric": "anchor decode (generate_neural_gaussians) fwd+bwd per second", "value": 1.0 / t_hip, "unit": "decodes/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_hip * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"decode: {N} anchors x {k} offsets, {n_vis} visible, {M} Gaussians out, feat 32, hidden 32, "
                                  f"add_*_dist on (the reference's default model, arguments/__init__.py:51-79); upstream gradients of the five outputs given"},
           "roofline": {"bound": "hbm", "kernel": "k_ng_opacity + k_ng_decode + k_ng_backward + weight-gradient GEMMs (whole step)",
                        "achieved": (fwd_b + bwd_b) / t_hip / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (fwd_b + bwd_b) / t_hip / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "mfma": {"bound": "mfma", "kernel": "k_ng_backward_t16 x 2 (the backward call: both launches + partial-sum fold + autograd glue)",
                                 "achieved": mfma_flop / t_bwd / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": mfma_flop / t_bwd / 1e12 / 157.3,
                                 "mfma_instructions_per_launch": mfma_insts, "backward_ms": t_bwd * 1e3,
                                 "note": "f32-in / f32-accumulate v_mfma_f32_16x16x4_f32, 2048 flop per instruction (round 4's 32x32x2 form issued "
                                         "1064 x 4096 flop per 64 anchors: 1.35 x the flops for the same result, so fractions across rounds compare by "
                                         "backward_ms, not by frac); the rest of the launches is the tile's memory round trips and the per-anchor "
                                         "stage (DESIGN.md section 7, EXPERIMENTS.md round 5)"}}}
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
        loss = image_loss(image, depth, gt, 0.2, scaling=scaling)["loss"]      # the reference's whole loss, scaling_reg (train.py:174) included, natively
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
    # the step's longest call, the decode's backward (two launches of k_ng_backward_t16), timed by HIP events on the op's stream around its
    # C-ABI call (5 more steps): its matrix-pipe side against the dense f32 MFMA peak (788 v_mfma_f32_16x16x4_f32 per 32-anchor tile, 2048
    # flop each)
    import neural_gaussians as ngmod
    real_call = ngmod._lib.lidargs_ng_backward_mfma
    spans = []

    class _Timed:                                                       # stands in for the ctypes library for five steps; every other name passes through
        def __getattr__(self, name):
            return getattr(ngmod._lib_real, name)

        def lidargs_ng_backward_mfma(self, *a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = real_call(*a); e1.record()
            spans.append((e0, e1))
            return rc

    ngmod._lib_real, ngmod._lib = ngmod._lib, _Timed()
    try:
        for _ in range(5):
            step(False)
        torch.cuda.synchronize()
    finally:
        ngmod._lib = ngmod._lib_real
    t_bwd_kernel = sum(a.elapsed_time(b) for a, b in spans) / max(1, len(spans)) * 1e-3
    mfma_insts = ((N + 31) // 32) * 788
    mfma = {"bound": "mfma", "kernel": "k_ng_backward_t16 x 2 (the step's longest call; HIP events around its C-ABI call inside the step)",
            "achieved": mfma_insts * 2048.0 / t_bwd_kernel / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": mfma_insts * 2048.0 / t_bwd_kernel / 1e12 / 157.3,
            "traffic": None, "ms": t_bwd_kernel * 1e3, "mfma_instructions_per_launch": mfma_insts,
            "note": "f32-in / f32-accumulate 16x16x4 tile products (round 4: 32x32x2 tiles, 1.35 x the flops for the same result, 885 us); the rest of "
                    "the two launches is the tile's memory round trips and the per-anchor stage"}
    print(json.dumps({
        "metric": "training-step core (decode + rasterize + loss, fwd+bwd) per second", "value": 1.0 / t_step, "unit": "steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_step * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"train_step: {N} anchors x {k} offsets -> {info.get('gaussians')} Gaussians ({info.get('visible')} on screen) @ {H}x{W}; "
                               f"generate_neural_gaussians + GaussianRasterizer + image loss, backward to anchors and MLP weights"},
        "stage_ms": {"decode_fwd": acc[0] / 5, "rasterize_fwd": acc[1] / 5, "loss_fwd+grad": acc[2] / 5, "backward(raster+decode)": acc[3] / 5},
        "roofline": mfma, "cpu_baseline": None}))


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
           # round 4: a uniform-grid search in front of the brute force (csrc/chamfer.hip).  What the evaluation has to move is the two
           # clouds in and a distance + index per point out; the ~20 small launches of the two directions (bounding box, cell counts, scan,
           # fill, query, the brute force that leaves at once) are latency-bound far below any roof.
           "roofline": {"bound": "hbm", "kernel": "k_ch_query (x2) + the grid build", "achieved": (2 * 12.0 * (n + m) + 8.0 * (n + m)) / t / 1e9, "peak": 8000.0,
                        "unit": "GB/s", "frac": (2 * 12.0 * (n + m) + 8.0 * (n + m)) / t / 1e9 / 8000.0, "traffic": None,
                        "algorithmic_bytes": "12 B per point and direction in, 8 B per point out",
                        "brute_force_equivalent": {"pairs_per_s": pairs / t, "note": "the reference's kernel evaluates every pair: 2 n m = %.3g per evaluation; LIDARGS_CHAMFER_BRUTE=1 runs that path (15.7 ms)" % pairs}}}
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


def bench_points_meter(args):
    """SURVEY section 8 row f3, second half: PointsMeter.update (utils/lidar_utils.py:253-282) on one 64 x 2650 frame -- range images in,
    chamfer distance and F-score out -- as one native call on device-resident images, beside the reference's dataflow on the same
    box (images to the host, numpy back-projection, clouds back to the device, the chamfer kernel, results read back)."""
    import numpy as np
    assert args.gpus == 1 and torch.cuda.is_available()
    import build_hip
    build_hip.build()
    import chamfer_3D
    import points_meter
    import lidargs_scenes as sc
    from oracle import range_view
    H, W = 64, 2650
    rng = np.random.default_rng(5)
    beams = np.ascontiguousarray(sc.beam_table(H, "waymo"), dtype=np.float32)
    truth = (rng.gamma(2.0, 9.0, size=(H, W)) + 2.0).astype(np.float32); truth[rng.random((H, W)) < 0.2] = 0.0
    pred = (truth * (1.0 + 0.005 * rng.normal(size=(H, W)))).astype(np.float32); pred[rng.random((H, W)) < 0.1] = 0.0
    tp, tt, tb = torch.from_numpy(pred).cuda(), torch.from_numpy(truth).cuda(), torch.from_numpy(beams).cuda()
    for _ in range(max(2, args.warmup // 3)):
        out = points_meter.points_metrics(tp, tt, beam_inclinations=tb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = max(5, args.steps)
    for _ in range(steps):
        out = points_meter.points_metrics(tp, tt, beam_inclinations=tb)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    o = out.cpu().numpy()
    n, m = int(o[4]), int(o[5])
    res = {"metric": "PointsMeter.update evaluations per second (range images -> chamfer distance + F-score)", "value": 1.0 / t, "unit": "evaluations/s", "n_gpus": 1,
           "steps": steps, "warmup": max(2, args.warmup // 3), "ms_per_step": t * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"points_meter: one {H}x{W} frame, {n} predicted and {m} ground-truth returns, threshold 0.05", "chamfer_distance": float(o[0]),
                      "f_score": float(o[1])},
           "roofline": {"bound": "hbm", "kernel": "flags + scan + back-projection + the grid search (x2) + the means: ~30 small launches", "achieved": (8.0 * H * W + 2 * 12.0 * (n + m) + 8.0 * (n + m)) / t / 1e9,
                        "peak": 8000.0, "unit": "GB/s", "frac": (8.0 * H * W + 2 * 12.0 * (n + m) + 8.0 * (n + m)) / t / 1e9 / 8000.0, "traffic": None,
                        "algorithmic_bytes": "the two images in, 12 B per point and direction through the search, 8 B per point of results: latency-bound far below any roof"}}
    if not args.no_cpu_baseline:
        # the reference's dataflow, its chamfer kernel replaced by ours (its CUDA extension cannot run here): what update() costs when the
        # images leave the device
        def reference_dataflow():
            p, q = tp.detach().cpu().numpy(), tt.detach().cpu().numpy()
            c1 = range_view.pano_to_points(p, np.zeros_like(p), beams)[:, :3]; c2 = range_view.pano_to_points(q, np.zeros_like(q), beams)[:, :3]
            x1, x2 = torch.FloatTensor(c1[None, ...]).cuda(), torch.FloatTensor(c2[None, ...]).cuda()
            d1, d2 = torch.empty(1, x1.shape[1], device="cuda"), torch.empty(1, x2.shape[1], device="cuda")
            i1, i2 = torch.empty(1, x1.shape[1], dtype=torch.int32, device="cuda"), torch.empty(1, x2.shape[1], dtype=torch.int32, device="cuda")
            chamfer_3D.forward(x1, x2, d1, d2, i1, i2)
            return float((d1.mean() + d2.mean()).cpu())
        reference_dataflow()
        t0 = time.perf_counter()
        for _ in range(5):
            cd = reference_dataflow()
        tr = (time.perf_counter() - t0) / 5
        res["cpu_baseline"] = {"value": 1.0 / tr, "unit": "evaluations/s", "cores": 1, "kind": "port",
                               "sample": f"5 evaluations of the reference's dataflow on this box: .cpu().numpy(), numpy pano_to_lidar (oracle/range_view.py), upload, the native chamfer kernel, "
                                         f".cpu() -- {tr * 1e3:.2f} ms each (chamfer distance {cd:.6f})"}
    else:
        res["cpu_baseline"] = None
    print(json.dumps(res))


def bench_anchor_growing(args):
    """SURVEY section 8 row f4, second half: GaussianModel.anchor_growing (scene/gaussian_model.py:677-775) at 1.2 M anchors x 6 offsets --
    the three levels (voxel edges 16 / 4 / 1 voxel sizes, thresholds x1 / x2 / x4) as three native calls, beside the reference's own op
    sequence (torch.unique + the chunked candidate-voxel x anchor equality scan + scatter max) run by torch on the same GPU."""
    import numpy as np
    assert args.gpus == 1 and torch.cuda.is_available()
    import build_hip
    build_hip.build()
    import anchor_growing as ag
    import lidargs_scenes as sc
    c = sc.anchor_scene(1_200_000, 6, 21)
    N, k, F = c["N"], c["k"], c["feat"].shape[1]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    anchor, offset, scaling, feat, grads, om = t(c["anchor"]), t(c["offset"]), torch.exp(t(c["scaling"])), t(c["feat"]), t(c["grads"]), t(c["offset_mask"])
    g = torch.Generator(device="cuda").manual_seed(7)
    rands = [torch.rand(N * k, device="cuda", generator=g) for _ in range(3)]
    levels = [(0.0005 * 2 ** i, 0.5 ** (i + 1), c["voxel"] * (16 // 4 ** i)) for i in range(3)]      # scene/gaussian_model.py:682, :688, :703-704 with arguments/__init__.py:55-57, :155

    def step():
        return [ag.grow_level(anchor, offset, scaling, feat, grads, om, rands[i], *levels[i]) for i in range(3)]
    warm = max(2, args.warmup // 5)
    for _ in range(warm):
        res = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = max(5, args.steps // 10)
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    tt = (time.perf_counter() - t0) / steps
    counts = [r[2] for r in res]
    C, V, U = (sum(x[j] for x in counts) for j in range(3))
    # what a level has to move: 9 B per offset through the mask (gradient, mask byte, random draw), 48 B per candidate (its offset row, its anchor's
    # position and scaling row), 12 B per anchor through the probe, F floats per candidate in and per new anchor out, 12 B per new anchor
    alg = 3 * (9.0 * N * k + 12.0 * N) + 48.0 * C + 4.0 * F * C + (12.0 + 4.0 * F) * U
    out = {"metric": "anchor_growing evaluations per second (three levels, 1.2 M anchors x 6 offsets)", "value": 1.0 / tt, "unit": "evaluations/s", "n_gpus": 1,
           "steps": steps, "warmup": warm, "ms_per_step": tt * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i32/u64 keys, f32 quantisation",
           "data": "synthetic",
           "config": {"workload": f"anchor_growing: {N} anchors x {k} offsets, voxel 0.01 m, levels (threshold, keep-probability, voxel edge) = "
                                  + ", ".join("(%.4g, %.3g, %.2f)" % (a, 1 - b, s_) for a, b, s_ in levels),
                      "per_level_candidates_voxels_new": counts},
           "roofline": {"bound": "hbm", "kernel": "k_ag_mark (x3: the 7.2 M-offset mask pass) + the hash-set passes", "achieved": alg / tt / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": alg / tt / 1e9 / 8000.0, "traffic": None,
                        "algorithmic_bytes": "per level 9 B per offset + 12 B per anchor; 48 B + 4 F B per candidate; (12 + 4 F) B per new anchor; two host reads per level (counts size the buffers)"}}
    if not args.no_cpu_baseline:
        # the reference's dataflow on this GPU (kind "port": its op sequence restated with the same torch ops, oracle/anchor_growing_torch.py)
        from oracle import anchor_growing as oag
        from oracle import anchor_growing_torch as agt
        ref = lambda chunked: [agt.grow_level(anchor, offset, scaling, feat, grads, om, rands[i], *levels[i], k, chunked=chunked) for i in range(3)]
        r0 = ref(True); torch.cuda.synchronize()
        t0 = time.perf_counter(); r0 = ref(True); torch.cuda.synchronize(); tr = time.perf_counter() - t0
        same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and tuple(a[2]) == tuple(b[2]) for a, b in zip(res, r0))
        out["framework_ops_same_gpu"] = {"ms": tr * 1e3, "speedup": tr / tt, "identical_outputs": bool(same),
                                         "what": "torch.unique(dim=0) + the reference's chunked (candidate voxel x anchor) equality scan in chunks of 4096 anchors + scatter amax, torch-ROCm on this GPU"}
        # host cores: the numpy oracle on the same three levels
        act = np.exp(c["scaling"]).astype(np.float32)
        t0 = time.perf_counter()
        for i in range(3):
            cand = oag.candidate_mask(c["grads"], c["offset_mask"], rands[i].cpu().numpy(), levels[i][0], levels[i][1], N * k)
            oag.grow_level(c["anchor"], c["offset"], act, c["feat"], cand, levels[i][2], exact_division=False)
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / tc, "unit": "evaluations/s", "cores": 1, "kind": "port",
                               "sample": f"one evaluation (the same three levels, the same inputs) with oracle/anchor_growing.py (numpy: sort-based unique, isin, maximum.at), {tc:.2f} s"}
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


def _host_staged_comm(lidargs_dist):
    """LIDARGS_BENCH_ONE_DEVICE=1 (tests): TorchDistComm over gloo with every tensor staged through the host, so that N ranks can share one GPU."""
    class HostStaged(lidargs_dist.TorchDistComm):
        def _via_host(self, fn, t, *a):
            return fn(t.cpu(), *a).to(t.device)

        def all_gather(self, t): return self._via_host(super().all_gather, t)
        def broadcast(self, t, src=0): t.copy_(self._via_host(super().broadcast, t, src)); return t
        def all_reduce(self, t): t.copy_(self._via_host(super().all_reduce, t)); return t
        def all_reduce_async(self, t): self.all_reduce(t); return lambda: None

        def all_reduce_max_async(self, t):
            h = t.cpu(); self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX, group=self.group); t.copy_(h); return lambda: None

        def all_to_all_rows(self, t, send_counts, recv_counts): return self._via_host(super().all_to_all_rows, t, send_counts, recv_counts)
        def reduce_scatter_rows(self, t): return self._via_host(super().reduce_scatter_rows, t)
    return HostStaged()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves, one process
    per GPU, exactly as the driver's torch.distributed.run command line would, and pass rank 0's JSON line through.  Fails with
    a plain statement when the box has fewer than N devices -- not with a launcher error."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not args.launch_check and not (have >= 1 and os.environ.get("LIDARGS_BENCH_ONE_DEVICE", "0") == "1"):
        raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} HIP devices, found {have}")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))   # dmabuf IPC only on these hosts
    raise SystemExit(subprocess.call(cmd, env=env))


def launch_check(args):
    """--launch-check: the rendezvous of the self-launched (or torchrun-launched) ranks without any GPU work -- gloo, one
    all-reduce, one JSON line from rank 0.  What tests/test_bench_cli_cpu.py runs here, where there is no GPU."""
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "n_gpus": args.gpus, "sum_of_ranks_plus_1": float(t.item())}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3",
                    help="cfg3 (headline) | cfg2 (forward only, as BASELINE.json states it) | cfg4 | cfg5 | render_fps | decode | loss | train_step | chamfer | points_meter | anchor_growing")
    ap.add_argument("--fwd-only", action="store_true", help="time the rasterizer forward alone (default for cfg2)")
    ap.add_argument("--fwd-bwd", action="store_true", help="forward + backward also for cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", default="both", choices=["both", "wedges", "shells"],
                    help="N > 1: range shells (the north star's cut by range: two-phase transmittance exchange, image all-gather, "
                         "gradient all-to-all), column wedges (every rank renders its pixel columns; image all-gather + gradient "
                         "all-to-all), or both timed one after the other in the same job (default): `value` is then the faster cut's, "
                         "named in config.sharding, and `cuts` holds both")
    ap.add_argument("--beams", default="uniform", choices=["uniform", "waymo", "neartie"],
                    help="beam-inclination table of the synthetic scene (lidargs_scenes.beam_table): uniform = SURVEY 8d; waymo = "
                         "non-uniform stand-in for the measured table the Waymo configs read from the dataset json")
    ap.add_argument("--opacity-scale", type=float, default=1.0,
                    help="cfg1-5: the scene's opacities U(0.1, 1) times this factor (lidargs_scenes.make_scene).  1 = the BASELINE.json scene (lists saturate "
                         "within a few entries); 0.3 / 0.1 / 0.03 = the semi-transparent frames training starts in: every list is walked to its end")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--graph", action="store_true",
                    help="single GPU: capture forward + backward of the enqueue-only path (lidargs_forward_enqueue: no host wait) in a HIP graph "
                         "and time its replays -- what a training loop does when the host, not the GPU, bounds a small frame")
    ap.add_argument("--enqueue-only", action="store_true",
                    help="render with GaussianRasterizer.enqueue_only (lidargs_forward_enqueue: no host wait per frame); single GPU only")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    if args.launch_check:
        return launch_check(args)

    if args.workload == "decode":
        return bench_decode(args)
    if args.workload == "loss":
        return bench_loss(args)
    if args.workload == "train_step":
        return bench_train_step(args)
    if args.workload == "chamfer":
        return bench_chamfer(args)
    if args.workload == "points_meter":
        return bench_points_meter(args)
    if args.workload == "anchor_growing":
        return bench_anchor_growing(args)
    if args.workload == "render_fps":
        return bench_render_fps(args)
    import lidargs_scenes as sc
    kind, P, H, W, seed = sc.BASELINE_CONFIGS[args.workload]
    if args.workload == "cfg5":
        return bench_surfel(args, sc, kind, P, H, W, seed)
    fwd_only = (args.fwd_only or args.workload == "cfg2") and not args.fwd_bwd
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU path"
    # LIDARGS_BENCH_ONE_DEVICE=1 (tests only, never set by the driver): every rank renders on device 0 and the collectives go through gloo with
    # host staging -- the N > 1 control flow of this file (self-launch, rebalancing rounds, max over ranks, one JSON line) on a one-GPU box.
    # The numbers of such a run mean nothing (the ranks share a GPU and the exchange crosses the host).
    one_device = world > 1 and os.environ.get("LIDARGS_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LIDARGS_BENCH_FORCE_SHELLS=1 runs the range-shell code path (RCCL collectives included) with a world of one: the only way
    # to exercise it end to end on a single-GPU box; never set by the driver
    force_shells = world == 1 and os.environ.get("LIDARGS_BENCH_FORCE_SHELLS", "0") == "1"
    if world > 1 or force_shells:
        import torch.distributed as dist
        if force_shells:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert not fwd_only, "the sharded path is timed forward + backward"

    if rank == 0:
        import build_hip
        build_hip.build()               # no-op when the in-tree .so is newer than every source; never a stale library
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    from diff_lidargs_rasterization import GaussianRasterizer, _C
    to_torch = lambda sd, device: {k_: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k_, v in sd.items()}
    make_settings = sc.raster_settings

    scene = sc.make_scene(kind, P, H, seed, beams=args.beams, opacity_scale=args.opacity_scale)
    st = to_torch(scene, dev)
    gc, gd, go = (torch.from_numpy(g).to(dev) for g in sc.upstream_grads(H, W, seed))
    settings = make_settings(st, W, H)
    leaves = {k: st[k].clone().requires_grad_(not fwd_only) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    means2D = torch.zeros((P, 4), dtype=torch.float32, device=dev, requires_grad=not fwd_only)

    sharded = world > 1 or force_shells

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(step, rebalance=None):
        """Clock ramp, W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; max over the ranks."""
        _C.profile_enable(True)             # pre-creates the event pool (one-off cost, outside the timed region)
        if rebalance is not None:
            rebalance()
            _C.profile_enable(True)
        clock_ramp(step, fixed_steps=300 if sharded else None)
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
        el = time.perf_counter() - t0
        _C.profile_enable(False)
        if world > 1:
            import torch.distributed as dist
            tmax = torch.tensor([el], dtype=torch.float64, device="cpu" if one_device else dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        allocs = int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0)
        stages = _C.profile_summary()
        _C.counters_enable(True)            # one more frame, outside the timed region, that ends with the counting launches (lidargs_last_counters)
        step(); barrier()
        cnt = _C.last_counters()
        _C.counters_enable(False)
        return dict(elapsed=el, stages=stages, cnt=cnt, allocs=allocs)

    cuts = {}
    if not sharded:
        rast = GaussianRasterizer(settings)
        rast.enqueue_only = bool(args.enqueue_only)

        if fwd_only:
            def step():
                with torch.no_grad():
                    rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], colors_precomp=leaves["colors"],
                         scales=leaves["scales"], rotations=leaves["rotations"])
        else:
            def step():
                for t in list(leaves.values()) + [means2D]:
                    t.grad = None
                color, depth, occ, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                                colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
                torch.autograd.backward([color, depth, occ], [gc, gd, go])
        eager_step = step

        def capture(r):
            """forward + backward of rasterizer `r` (enqueue-only path) as a HIP graph; returns the replay callable"""
            r.enqueue_only = True

            def one():
                for t in list(leaves.values()) + [means2D]:
                    t.grad = None
                color, depth, occ, radii = r(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                             colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
                torch.autograd.backward([color, depth, occ], [gc, gd, go])
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                 # learn capacity / tile height and warm up off the default stream, as capture asks
                for _ in range(4):
                    one()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for t in list(leaves.values()) + [means2D]:
                t.grad = None
            import gc as _gc
            _gc.collect()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                one()
            return graph.replay
        if args.graph:
            assert not fwd_only, "--graph captures forward + backward"
            step = capture(rast)
        res = timed_region(step)
        graph_leg = None
        if args.workload == "cfg1" and not args.graph and not fwd_only:
            # a 10 k-Gaussian frame is ~0.15 ms of device time behind ~0.2-0.35 ms of per-frame host work (Python, ctypes, one host read):
            # the eager number measures the box's CPU.  The same frame as a HIP-graph replay rides along in the line.
            rast_g = GaussianRasterizer(settings)
            replay = capture(rast_g)
            for _ in range(args.warmup):
                replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                replay()
            torch.cuda.synchronize()
            tg = (time.perf_counter() - t0) / args.steps
            assert not rast_g.enqueue_status()["overflow"], "the captured binning capacity was too small"
            graph_leg = {"ms_per_step": tg * 1e3, "value": 1.0 / tg, "unit": "frames/s",
                         "what": "the same forward + backward captured once in a HIP graph (enqueue-only path: no host read) and replayed"}
        if args.graph:
            assert not rast.enqueue_status()["overflow"], "the captured binning capacity was too small"
            _C.profile_enable(True)                       # a replay makes no library calls: the stage events come from eager frames, after the timed region
            rast2 = GaussianRasterizer(settings)
            def eager2():
                for t in list(leaves.values()) + [means2D]:
                    t.grad = None
                color, depth, occ, radii = rast2(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                                 colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
                torch.autograd.backward([color, depth, occ], [gc, gd, go])
            for _ in range(16):
                eager2()
            torch.cuda.synchronize()
            res["stages"] = _C.profile_summary()
            _C.profile_enable(False)
            _C.counters_enable(True)
            eager2(); torch.cuda.synchronize()
            res["cnt"] = _C.last_counters()
            _C.counters_enable(False)
    else:
        import math
        import torch.distributed as dist
        import lidargs_dist
        comm = _host_staged_comm(lidargs_dist) if one_device else lidargs_dist.TorchDistComm()
        # range-shell / wedge edges are a load-balancing choice, not a result: cut once for this (static) scene and view
        beams = st["beams"]
        tile_rad = (16 * 2 * math.pi / W, 4 * float(beams[-1] - beams[0]) / max(1, H - 1))      # 16 columns x 4 rows per tile
        for shard in (("shells", "wedges") if args.shard == "both" else (args.shard,)):
            if shard == "shells":
                cut = lambda shares: comm.broadcast(lidargs_dist.shell_edges(st["means3D"], st["viewmatrix"], world, 0, 80, scales=st["scales"],
                                                                             tile_rad=tile_rad, shares=shares), 0)
                rast = lidargs_dist.ShellRasterizer(settings, comm, edges=cut(None))
            else:
                def cut(shares):
                    e = torch.tensor(lidargs_dist.wedge_edges(st["means3D"], st["viewmatrix"], W, world, scales=st["scales"], shares=shares),
                                     dtype=torch.int32, device=dev)
                    return [int(x) for x in comm.broadcast(e, 0).tolist()]
                rast = lidargs_dist.WedgeRasterizer(settings, comm, edges=cut(None))

            def step(rast=rast):
                for t in list(leaves.values()) + [means2D]:
                    t.grad = None
                color, depth, occ, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                                                colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
                torch.autograd.backward([color, depth, occ], [gc, gd, go])

            def rebalance(rast=rast, cut=cut, step=step):
                # Measured load balancing of the (static) cut, outside the timed region: a few frames per round, every rank's own
                # kernel time (its HIP-event stage sums, which do not include waiting for the other ranks) -> thinner shells /
                # narrower wedges for the slow ranks.
                shares = [1.0 / world] * world
                for _round in range(4):
                    _C.profile_enable(True)
                    for _ in range(3):
                        step()
                    torch.cuda.synchronize()
                    mine = torch.tensor([sum(v[0] for v in _C.profile_summary().values())], dtype=torch.float64, device="cpu" if one_device else dev)
                    times = [torch.zeros_like(mine) for _ in range(world)]
                    dist.all_gather(times, mine)
                    shares = lidargs_dist.rebalance_shares(shares, [float(t) for t in times], fixed=0.25)
                    rast.edges = cut(shares)
            cuts[shard] = timed_region(step, rebalance)
            cuts[shard]["edges"] = [float(e) for e in (rast.edges.tolist() if hasattr(rast.edges, "tolist") else rast.edges)]
        best = min(cuts, key=lambda k: cuts[k]["elapsed"])
        res = cuts[best]
        args.shard = best
    elapsed, stages, cnt = res["elapsed"], res["stages"], res["cnt"]
    allocs0 = 0
    allocs1 = res["allocs"]
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
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        N_pix, T_ref = H * W, H * ((W + 15) // 16)
        what = "forward only" if fwd_only else "fwd+bwd"
        out = {
            "metric": f"LiDAR range-view frames/sec ({what})", "value": args.steps / elapsed, "unit": "frames/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {P} Gaussians ({kind} scene, seed {seed}) @ {H}x{W}, {what}, "
                                   f"lidar_far=80 lidar_near=0, bg=0" + ("" if args.beams == "uniform" else f", beam table '{args.beams}' (non-uniform)")
                                   + ("" if args.opacity_scale == 1.0 else f", opacities x {args.opacity_scale:g} (semi-transparent, early-training regime)"),
                       "visible_gaussians": cnt["V"], "instances_binned": cnt["instances"], "R_ref_16x1": cnt["R_ref"],
                       "patch_instance_pairs_taken": cnt["taken_instances"], "touched_gaussians": cnt.get("touched", -1), "tile_rows": cnt["tile_rows"], "segment_slots": cnt["segments"],
                       "forward": ("HIP-graph replay of forward + backward (enqueue-only path captured once)" if args.graph else
                                   "enqueue-only (lidargs_forward_enqueue, no host wait)" if args.enqueue_only else "lidargs_forward (one 2-KB host read per frame)"),
                       "sharding": "single GPU" if world == 1 and not force_shells else
                                   (f"{world} range shells" if args.shard == "shells" else f"{world} column wedges")},
        }
        if not sharded and graph_leg is not None:
            out["graph_replay"] = graph_leg
        if world == 1 and not force_shells:
            # roofline: every launch (group) priced on what IT processes (R' = the instances this frame binned), the longest single
            # launch on top; the reference data flow's bytes (R_ref 16x1 instances) against our time are kept apart, they are a
            # speed-up statement, not a fraction of any roof
            table = raster_kernel_table(P, cnt["V"], cnt["instances"], N_pix, stages, taken=cnt["taken_instances"], touched=cnt.get("touched", -1),
                                        bwd_entries=cnt.get("backward_entries", -1))
            fwd_b, bwd_b, _k7, _k8 = reference_dataflow_bytes(P, cnt["V"], cnt["R_ref"], N_pix, T_ref)
            ref_bytes = fwd_b if fwd_only else fwd_b + bwd_b
            ref_flow = {"frame_bytes_of_the_reference_dataflow": ref_bytes, "GBs_at_our_frame_time": ref_bytes / (ms_per_step * 1e-3) / 1e9,
                        "note": "SURVEY 8d formula on R_ref (16x1 instances the reference would bin) / our frame time: above the 8000 GB/s "
                                "peak means the frame is faster than that data flow could be at HBM speed; NOT a roofline fraction"}
            pmc_names = {"k_render_backward": ["lg::k_render_backward"], "k_preprocess": ["lg::k_preprocess<false>"],
                         "k_gaussian_backward": ["lg::k_gaussian_backward"], "k_emit_instances": ["lg::k_emit_instances"], "k_zero_touched": ["lg::k_zero_touched"],
                         "k_tile_ranges": ["lg::k_tile_ranges"],
                         "forward blend group (reference K7): T-only walks + alive + full walk + combine":
                             {"any_of": ["lg::k_render_forward<", "lg::k_render_pass2_grouped", "lg::k_render_alive", "lg::k_render_combine", "lg::k_render_fused"],
                              "per_frame": "k_preprocess"}}
            pmc_names.update(sort_pmc_groups(int(cnt["tiles"]) if "tiles" in cnt else T_ref // max(1, int(cnt["tile_rows"]))))
            out["roofline"] = roofline_object(table, args.workload, pmc_names, ref_flow)
            # the frame as a whole against the HBM roof, from the same per-launch units (+ what the table leaves out is small)
            own = sum(k["bytes"] for k in table)
            out["roofline"]["frame"] = {"algorithmic_bytes_of_the_listed_launches": own, "GBs": own / (ms_per_step * 1e-3) / 1e9,
                                        "frac": own / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
        else:
            # sharded: rank 0's own launches on rank 0's own units (its selected Gaussians, the instances it binned, its columns);
            # no PMC profile exists for a rank's sub-frame, so HBM fractions only
            try:
                n_own = N_pix // world if args.shard == "wedges" else N_pix
                table = raster_kernel_table(int(cnt["P"]), int(cnt["V"]), int(cnt["instances"]), n_own, stages, taken=int(cnt["taken_instances"]),
                                            touched=int(cnt.get("touched", -1)), bwd_entries=int(cnt.get("backward_entries", -1)))
                out["roofline"] = roofline_object(table, "sharded-" + args.workload, {})
                out["roofline"]["note"] = f"rank 0 of {world}: {int(cnt['P'])} Gaussians selected, {int(cnt['instances'])} instances binned"
            except Exception as e:      # the headline number must not depend on the diagnostics
                out["roofline"] = {"error": str(e)}
        if sharded:
            # every cut that was timed (same job, same ranks, one after the other); `value` above is the faster one's
            out["cuts"] = {k: {"value": args.steps / v["elapsed"], "ms_per_step": v["elapsed"] / args.steps * 1e3, "edges": v["edges"],
                               "stage_ms_rank0": {a: round(b[0], 4) for a, b in v["stages"].items()},
                               "collectives": ("all_reduce(radii), all_gather(T_pass), all_gather(5 planes), all_to_all(gradient rows)" if k == "shells"
                                               else "all_reduce_max(radii), all_gather(own columns of 4 planes), all_to_all(gradient rows, added by the owner)")}
                           for k, v in cuts.items()}
            out["rccl_ranks"] = world
        out.update({
            "stage_ms": {k: round(v[0], 4) for k, v in stages.items()},
            "stage_events": f"HIP events on the op's stream, on every {STAGE_EVERY}th frame of the timed region",
            "frame_ms_spread": spread,
            "hipmalloc_calls_in_timed_region": int(allocs1 - allocs0),
            "clock_ramp": "0.5 s of untimed frames (300 frames when sharded) ran before the warm-up steps: the first frames of a fresh "
                          "process run 5-8 % slower than steady state",
        })
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(kind, P, H, W, seed, fwd_only=fwd_only)
        elif world == 1:
            out["cpu_baseline"] = None
        _flush_c_stdio()
        print(json.dumps(out), flush=True)
    if world > 1 or force_shells:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
