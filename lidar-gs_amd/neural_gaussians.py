"""`generate_neural_gaussians` of LiDAR-GS (gaussian_renderer/__init__.py:17-119) on the fused HIP anchor decode.

Same signature and return values as the reference function; a checkout switches to it with one import
(`from neural_gaussians import generate_neural_gaussians` in gaussian_renderer/__init__.py).  The ~40
framework ops of the reference become three native calls (include/lidargs_neural_gaussians.h); the eight weight gradients
are the plain GEMMs torch already does well (hipBLASLt), fed by the per-anchor deltas the backward kernel writes.

Model configurations outside the native kernel's scope (feature bank, appearance embedding, colour channels != 2, feature /
hidden width != 32, n_offsets not in {4,5,6,8,10}) raise NotImplementedError: there is no silent framework fallback.
"""
import ctypes as C
import os

import torch

from diff_lidargs_rasterization import _C as _base

_lib = _base._lib
MLP_ORDER = ("opacity", "cov", "color", "raydrop")


class _Model(C.Structure):
    _fields_ = [("n_offsets", C.c_int), ("add_opacity_dist", C.c_int), ("add_cov_dist", C.c_int), ("add_color_dist", C.c_int),
                ("W1", C.c_void_p * 4), ("b1", C.c_void_p * 4), ("W2", C.c_void_p * 4), ("b2", C.c_void_p * 4), ("W2T", C.c_void_p * 4)]


for _n in ("lidargs_ng_forward_select", "lidargs_ng_forward_select_enqueue", "lidargs_ng_forward_decode", "lidargs_ng_backward", "lidargs_ng_backward_mfma",
           "lidargs_ng_backward_partials", "lidargs_ng_training_stats", "lidargs_ng_weight_grad_floats", "lidargs_ng_weight_grad_stage_floats", "lidargs_ng_reduce_weight_grads", "lidargs_ng_transpose_w2"):
    getattr(_lib, _n).restype = C.c_int
_lib.lidargs_ng_scratch_bytes.restype = C.c_size_t


def _transposed_w2(params, k, dev):
    """Transposed second-layer weights, the B operand of Y = H W2^T on the matrix pipe (forward and backward): one launch for the four."""
    out = torch.empty(320 * k, dtype=torch.float32, device=dev)
    ptrs = (C.c_void_p * 4)(*[params[4 * i + 2].data_ptr() for i in range(4)])
    with torch.cuda.device(dev):
        _check(_lib.lidargs_ng_transpose_w2(C.c_int(k), ptrs, _base._ptr(out), _base._stream(dev)), "lidargs_ng_transpose_w2")
    o = (0, k, 8 * k, 9 * k)
    return tuple(out[32 * o[i]:32 * o[i] + 32 * (7 * k if i == 1 else k)].view(32, 7 * k if i == 1 else k) for i in range(4))


# Largest N k x 52-byte output block the decode allocates to skip the wait (0 = always wait).  The block is capacity-sized -- N k rows, of
# which the M selected ones (typically a fifth to a third) are handed out as views -- and autograd keeps it alive until the backward, so
# the default bounds what a step can hold that way at 256 MB (4.9 M candidate rows: 820 k anchors x 6 offsets); larger models take the
# waiting path, whose outputs are exactly M rows (round-3 advisor finding: the default used to be 1 GB).
_CAPACITY_BYTES = int(os.environ.get("LIDARGS_NG_CAPACITY_BYTES", str(256 << 20)))
_PINNED = {}
_PINNED_LOCK = __import__("threading").Lock()


def _pinned_counts(dev):
    """Per (device, thread): two pinned ints the selection's counts land in, and the event in front of the decode launch.  The device
    index is resolved (a bare "cuda" device has none) and the event is created with that device current: an event binds to the device of
    its first record.  The table is shared by the autograd threads: guarded."""
    import threading
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (index, threading.get_ident())
    with _PINNED_LOCK:
        hit = _PINNED.get(key)
        if hit is None:
            with torch.cuda.device(index):
                hit = (torch.zeros(2, dtype=torch.int32).pin_memory(), torch.cuda.Event())
            _PINNED[key] = hit
    return hit


def _f32c(t):
    """t as contiguous float32 -- t itself when it already is (the usual case: twenty tensors per call, three framework calls each otherwise)."""
    if t.dtype == torch.float32 and t.is_contiguous():
        return t
    return t.detach().to(torch.float32).contiguous()


def _check(rc, what):
    if rc < 0:
        _base._raise(rc, what)


def _model_struct(k, flags, params, w2t=None):
    m = _Model()
    m.n_offsets, m.add_opacity_dist, m.add_cov_dist, m.add_color_dist = k, int(flags[0]), int(flags[1]), int(flags[2])
    for i in range(4):
        m.W1[i], m.b1[i], m.W2[i], m.b2[i] = (params[4 * i + j].data_ptr() for j in range(4))
        m.W2T[i] = w2t[i].data_ptr() if w2t is not None else None
    return m


class _Decode(torch.autograd.Function):
    """inputs: anchor_feat [N,32], anchor [N,3], offset [N,k,3], scaling [N,6] (get_scaling), then W1,b1,W2,b2 of the opacity, cov,
    color and raydrop MLPs; cam (3 floats, host), visible_mask (bool[N] or None), flags (3 bools)."""

    @staticmethod
    def forward(ctx, anchor_feat, anchor, offset, scaling, *rest):
        params, (cam, visible_mask, flags) = rest[:16], rest[16:]
        _base._require_device(anchor, "anchor")
        dev = anchor.device
        f32 = _f32c
        anchor_feat, anchor, offset, scaling = f32(anchor_feat), f32(anchor), f32(offset), f32(scaling)
        params = tuple(f32(p) for p in params)
        N, k = int(anchor.shape[0]), int(offset.shape[1])
        w2t = _transposed_w2(params, k, dev)
        model = _model_struct(k, flags, params, w2t)
        nb = int(_lib.lidargs_ng_scratch_bytes(C.c_int(N), C.c_int(k)))
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        neural_opacity = torch.empty(max(N, 1) * k, dtype=torch.float32, device=dev)
        mask = torch.empty(max(N, 1) * k, dtype=torch.uint8, device=dev)
        vis = None if visible_mask is None else visible_mask.to(torch.bool).contiguous().view(torch.uint8)
        counts = (C.c_int * 2)()
        camv = (C.c_float * 3)(*[float(c) for c in cam])
        p = _base._ptr
        with torch.cuda.device(dev):
            stream = _base._stream(dev)
            # everything the decode launch needs that does not depend on (n, M), before the select's host read: the device is idle from
            # that read to the launch, so nothing but the output allocation sits between them (the output views are made behind it)
            dec_head = (C.c_int(N), C.byref(model), p(anchor_feat), p(anchor), p(offset), p(scaling), camv, p(neural_opacity))
            dec_tail = (p(scratch), C.c_size_t(nb), stream)
            cap = N * k                                                # rows of each output array: M itself, or its upper bound
            if 0 < cap * 52 <= _CAPACITY_BYTES:
                # No idle device between the two steps: the decode writes into arrays of N k rows (its rows come from the device-side
                # scan), queued right behind the selection; the host waits for the two counts -- an event in front of the decode --
                # while it runs, and the outputs are the first M rows.
                pinned, ev = _pinned_counts(dev)
                _check(_lib.lidargs_ng_forward_select_enqueue(C.c_int(N), C.byref(model), p(vis), p(anchor_feat), p(anchor), camv, p(neural_opacity),
                                                              p(mask), C.c_void_p(pinned.data_ptr()), p(scratch), C.c_size_t(nb), stream),
                       "lidargs_ng_forward_select_enqueue")
                ev.record(torch.cuda.current_stream(dev))
                out = torch.empty(cap * 13, dtype=torch.float32, device=dev)
                base = out.data_ptr()
                at = lambda floats: C.c_void_p(base + 4 * floats)
                _check(_lib.lidargs_ng_forward_decode(*dec_head, at(0), at(3 * cap), at(5 * cap), at(6 * cap), at(9 * cap), *dec_tail), "lidargs_ng_forward_decode")
                ev.synchronize()
                n, M = int(pinned[0]), int(pinned[1])
            else:
                _check(_lib.lidargs_ng_forward_select(C.c_int(N), C.byref(model), p(vis), p(anchor_feat), p(anchor), camv, p(neural_opacity), p(mask),
                                                      counts, p(scratch), C.c_size_t(nb), stream), "lidargs_ng_forward_select")
                n, M = int(counts[0]), int(counts[1])
                cap = M
                out = torch.empty(M * 13, dtype=torch.float32, device=dev)
                if N:
                    base = out.data_ptr()
                    at = lambda floats: C.c_void_p(base + 4 * floats) if M else None
                    _check(_lib.lidargs_ng_forward_decode(*dec_head, at(0), at(3 * M), at(5 * M), at(6 * M), at(9 * M), *dec_tail), "lidargs_ng_forward_decode")
            xyz, color, opacity = out[:3 * M].view(M, 3), out[3 * cap:3 * cap + 2 * M].view(M, 2), out[5 * cap:5 * cap + M].view(M, 1)
            scal, rot = out[6 * cap:6 * cap + 3 * M].view(M, 3), out[9 * cap:9 * cap + 4 * M].view(M, 4)
        ctx.set_materialize_grads(False)          # an unused output (neural_opacity, the mask) must not cost a zero fill of n k entries
        ctx.save_for_backward(anchor_feat, anchor, offset, scaling, scratch, *params, *w2t)
        ctx.meta = (N, k, n, M, tuple(float(c) for c in cam), tuple(bool(f) for f in flags))
        neural_opacity = neural_opacity[:n * k].view(n * k, 1)
        mask_b = mask[:n * k].view(torch.bool)
        ctx.mark_non_differentiable(mask_b)
        return xyz, color, opacity, scal, rot, neural_opacity, mask_b

    @staticmethod
    def backward(ctx, g_xyz, g_color, g_opacity, g_scaling, g_rot, g_no, _g_mask):
        anchor_feat, anchor, offset, scaling, scratch = ctx.saved_tensors[:5]
        params, w2t = ctx.saved_tensors[5:21], ctx.saved_tensors[21:25]
        N, k, n, M, cam, flags = ctx.meta
        dev = anchor.device
        model = _model_struct(k, flags, params, w2t)
        camv = (C.c_float * 3)(*cam)
        f = lambda g, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if g is None else _f32c(g)
        g_xyz, g_color, g_opacity = f(g_xyz, (M, 3)), f(g_color, (M, 2)), f(g_opacity, (M, 1))
        g_scaling, g_rot = f(g_scaling, (M, 3)), f(g_rot, (M, 4))
        g_no = None if g_no is None else _f32c(g_no)
        dense = torch.empty(N * (32 + 3 + 3 * k + 6), dtype=torch.float32, device=dev)          # every row is written by the kernel
        p = _base._ptr

        def dense_views():
            o = 0
            d_feat = dense[o:o + N * 32].view(N, 32); o += N * 32
            d_anchor = dense[o:o + N * 3].view(N, 3); o += N * 3
            d_offset = dense[o:o + N * 3 * k].view(N, k, 3); o += N * 3 * k
            return d_feat, d_anchor, d_offset, dense[o:o + N * 6].view(N, 6)
        if os.environ.get("LIDARGS_NG_ACT_BUFFERS", "0") == "1":
            # the older formulation: the kernel writes what two library GEMMs reduce (1.4 KB per visible anchor)
            d_feat, d_anchor, d_offset, d_scaling = dense_views()
            act_x = torch.empty((n, 40), dtype=torch.float32, device=dev)
            act_h = torch.empty((n, 132), dtype=torch.float32, device=dev)
            delta1 = torch.empty((n, 128), dtype=torch.float32, device=dev)
            delta2 = torch.empty((n, 10 * k), dtype=torch.float32, device=dev)
            if N:
                with torch.cuda.device(dev):
                    _check(_lib.lidargs_ng_backward(C.c_int(N), C.c_int(n), C.byref(model), p(anchor_feat), p(anchor), p(offset), p(scaling), camv,
                                                    p(g_xyz), p(g_color), p(g_opacity), p(g_scaling), p(g_rot), p(g_no), p(d_feat), p(d_anchor),
                                                    p(d_offset), p(d_scaling), p(act_x), p(act_h), p(delta1), p(delta2), p(scratch),
                                                    C.c_size_t(scratch.numel()), _base._stream(dev)), "lidargs_ng_backward")
            # With n ~ 1e5..1e6 and a, b <= 132 a plain mm picks a one-tile kernel that walks all of n serially (0.5 ms each), so n is
            # split into chunks: a batched GEMM of partial products plus a small sum.
            def tn(a, b):
                rows = a.shape[0]
                if rows == 0:
                    return torch.zeros((a.shape[1], b.shape[1]), dtype=torch.float32, device=dev)
                S = max(1, min(512, rows // 512))
                cut = (rows // S) * S
                out = torch.bmm(a[:cut].view(S, cut // S, a.shape[1]).transpose(1, 2), b[:cut].view(S, cut // S, b.shape[1])).sum(0)
                return out + a[cut:].t() @ b[cut:] if cut < rows else out
            G1, G2 = tn(delta1, act_x), tn(delta2, act_h)              # [128, 40], [10k, 132]
        else:
            # everything matrix-shaped on the matrix pipe, weight gradients included: the persistent waves' partial sums come back,
            # one row per wave
            waves, per_wave = C.c_int(0), C.c_int(0)
            _check(_lib.lidargs_ng_backward_partials(C.c_int(k), C.byref(waves), C.byref(per_wave)), "lidargs_ng_backward_partials")
            partials = torch.empty((waves.value, per_wave.value), dtype=torch.float32, device=dev)
            if N:
                with torch.cuda.device(dev):
                    base = dense.data_ptr()
                    at = lambda floats: C.c_void_p(base + 4 * floats)       # the launch first, the views of `dense` behind it
                    _check(_lib.lidargs_ng_backward_mfma(C.c_int(N), C.byref(model), p(anchor_feat), p(anchor), p(offset), p(scaling), camv,
                                                         p(g_xyz), p(g_color), p(g_opacity), p(g_scaling), p(g_rot), p(g_no), at(0), at(N * 32),
                                                         at(N * 35), at(N * (35 + 3 * k)), p(partials), p(scratch), C.c_size_t(scratch.numel()),
                                                         _base._stream(dev)), "lidargs_ng_backward_mfma")
            d_feat, d_anchor, d_offset, d_scaling = dense_views()
            # sum over the waves and unpack into the sixteen parameter gradients, two small launches (before: a framework reduction over
            # [1024, 10368], four concatenations and four strided copies, ~90 us of the backward)
            dins = (35 + int(flags[0]), 35 + int(flags[1]), 35 + int(flags[2]), 35 + int(flags[2]))
            douts = (k, 7 * k, k, k)
            din_c = (C.c_int * 4)(*dins)
            nfl = int(_lib.lidargs_ng_weight_grad_floats(C.c_int(k), din_c))
            _check(nfl, "lidargs_ng_weight_grad_floats")
            if N:
                nst = int(_lib.lidargs_ng_weight_grad_stage_floats(C.c_int(k), din_c))
                flat = torch.empty(nfl + nst, dtype=torch.float32, device=dev)     # the gradients, then the first stage's row-group sums
                with torch.cuda.device(dev):
                    _check(_lib.lidargs_ng_reduce_weight_grads(C.c_int(k), din_c, C.c_int(waves.value), p(partials), p(flat), p(flat[nfl:]), _base._stream(dev)),
                           "lidargs_ng_reduce_weight_grads")
            else:
                flat = torch.zeros(nfl, dtype=torch.float32, device=dev)
            g_params, o = [], 0
            for i in range(4):
                for shape in ((32, dins[i]), (32,), (douts[i], 32), (douts[i],)):
                    cnt = shape[0] * (shape[1] if len(shape) == 2 else 1)
                    g_params.append(flat[o:o + cnt].view(shape)); o += cnt
            return (d_feat, d_anchor, d_offset, d_scaling, *g_params, None, None, None)
        cols = ((0, k), (k, 8 * k), (8 * k, 9 * k), (9 * k, 10 * k))
        dins = (35 + int(flags[0]), 35 + int(flags[1]), 35 + int(flags[2]), 35 + int(flags[2]))
        g_params = []
        for i in range(4):
            r0, r1 = cols[i]
            g_params += [G1[32 * i:32 * i + 32, :dins[i]], G1[32 * i:32 * i + 32, 36], G2[r0:r1, 32 * i:32 * i + 32], G2[r0:r1, 128]]
        return (d_feat, d_anchor, d_offset, d_scaling, *g_params, None, None, None)


def _linear_pair(seq, what, head):
    """Weights of one of the four MLPs after checking its WHOLE layout: the native decode hard-codes Linear(din, 32) - ReLU -
    Linear(32, dout) followed by `head` (scene/gaussian_model.py:113-142: Tanh for opacity, none for cov, Sigmoid for colour and
    ray-drop).  Anything else -- another activation, a missing head, an extra layer -- would decode and back-propagate silently
    wrong, so it is refused like every other unsupported option."""
    import torch.nn as nn
    mods = list(seq)
    want = [nn.Linear, nn.ReLU, nn.Linear] + ([head] if head is not None else [])
    ok = len(mods) == len(want) and all(type(m) is w for m, w in zip(mods, want))
    if ok:
        ok = mods[0].out_features == 32 and mods[2].in_features == 32 and mods[0].bias is not None and mods[2].bias is not None
    if not ok:
        layout = "-".join(type(m).__name__ for m in mods)
        raise NotImplementedError(f"neural_gaussians: unsupported {what} MLP [{layout}]: the native decode handles exactly "
                                  f"Linear(din,32)-ReLU-Linear(32,dout){'-' + head.__name__ if head is not None else ''}")
    return mods[0].weight, mods[0].bias, mods[2].weight, mods[2].bias


def _camera_center_host(viewpoint_camera):
    """The camera centre as three Python floats (the C ABI takes it by value).  Reading a device tensor waits for everything queued in front
    of it -- at the top of a training step that is the previous step's whole backward, and the host then prepares this step's launches with the
    device idle.  A camera's centre does not change between the iterations that draw it (scene/cameras.py:58): the host copy is kept on the
    camera object and reused while the tensor is the same object at the same version (any in-place write bumps `_version`)."""
    t = viewpoint_camera.camera_center
    key = (id(t), int(getattr(t, "_version", 0)), t.device)
    hit = getattr(viewpoint_camera, "_lidargs_camera_center_host", None)
    if hit is not None and hit[0] == key and hit[1]() is t:
        return hit[2]
    cam = t.detach().to("cpu", torch.float32).reshape(3).tolist()
    try:
        import weakref
        viewpoint_camera._lidargs_camera_center_host = (key, weakref.ref(t), cam)
    except (AttributeError, TypeError):
        pass                                                           # a camera object that takes no attributes: read every time
    return cam


def generate_neural_gaussians(viewpoint_camera, pc, visible_mask=None, is_training=False):
    """Drop-in for gaussian_renderer.generate_neural_gaussians (:17-119): same arguments, same 5- or 7-tuple."""
    if getattr(pc, "use_feat_bank", False):
        raise NotImplementedError("neural_gaussians: use_feat_bank=True is not supported by the native decode")
    if getattr(pc, "appearance_dim", 0) > 0:
        raise NotImplementedError("neural_gaussians: appearance_dim > 0 is not supported by the native decode")
    if getattr(pc, "color_channel", 2) != 2:
        raise NotImplementedError("neural_gaussians: colour channels other than 2 (intensity + ray-drop) are not supported")
    anchor_feat, anchor, offset, scaling = pc._anchor_feat, pc.get_anchor, pc._offset, pc.get_scaling
    if anchor_feat.shape[1] != 32:
        raise NotImplementedError("neural_gaussians: feat_dim must be 32")
    k = int(pc.n_offsets)
    if k not in (4, 5, 6, 8, 10) or offset.shape[1] != k:
        raise NotImplementedError("neural_gaussians: n_offsets must be 4, 5, 6, 8 or 10")
    import torch.nn as nn
    params = (*_linear_pair(pc.get_opacity_mlp, "opacity", nn.Tanh), *_linear_pair(pc.get_cov_mlp, "cov", None),
              *_linear_pair(pc.get_color_mlp, "color", nn.Sigmoid), *_linear_pair(pc.get_raydrop_mlp, "raydrop", nn.Sigmoid))
    flags = (bool(pc.add_opacity_dist), bool(pc.add_cov_dist), bool(pc.add_color_dist))
    cam = _camera_center_host(viewpoint_camera)
    xyz, color, opacity, scal, rot, neural_opacity, mask = _Decode.apply(anchor_feat, anchor, offset, scaling, *params, cam, visible_mask, flags)
    if is_training:
        return xyz, color, opacity, scal, rot, neural_opacity, mask
    return xyz, color, opacity, scal, rot


def training_statis(pc, viewspace_point_tensor, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
    """Drop-in for GaussianModel.training_statis (scene/gaussian_model.py:599-622), as a function of the model: updates
    pc.opacity_accum, pc.anchor_demon, pc.offset_gradient_accum and pc.offset_denom in place with one native call (no masks
    expanded, no boolean indexing, no host read).  Call it as `training_statis(gaussians, ...)` where train.py:243 calls the method."""
    dev = pc.opacity_accum.device
    _base._require_device(pc.opacity_accum, "opacity_accum")
    N, k = int(pc.opacity_accum.shape[0]), int(pc.n_offsets)
    u8 = lambda t: t.to(torch.bool).contiguous().view(torch.uint8)
    f32 = lambda t: t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous()
    for t in (pc.opacity_accum, pc.anchor_demon, pc.offset_gradient_accum, pc.offset_denom):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("training_statis: the accumulators must be contiguous float32 tensors")
    vis, sel, upd = u8(anchor_visible_mask), u8(offset_selection_mask), u8(update_filter)
    op, grad = f32(opacity), f32(viewspace_point_tensor.grad)
    nb = int(_lib.lidargs_ng_scratch_bytes(C.c_int(N), C.c_int(k)))
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    p = _base._ptr
    with torch.no_grad(), torch.cuda.device(dev):
        _check(_lib.lidargs_ng_training_stats(C.c_int(N), C.c_int(k), p(vis), p(sel), p(upd), p(op), p(grad), p(pc.opacity_accum), p(pc.anchor_demon),
                                              p(pc.offset_gradient_accum), p(pc.offset_denom), p(scratch), C.c_size_t(nb), _base._stream(dev)),
               "lidargs_ng_training_stats")
