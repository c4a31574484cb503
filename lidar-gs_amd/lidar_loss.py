"""The per-frame loss of LiDAR-GS training (train.py:150-203) on the fused HIP kernels of include/lidargs_loss.h.

    terms = image_loss(image, depth, gt_image, opt.lambda_dssim, scaling=scaling)      # image [2,H,W], depth [1,H,W], gt_image [3,H,W]
    terms["loss"].backward()                                                            # = the reference's `loss`, scaling_reg included

`scaling` (optional, [M, 3], the anchor decode's output) adds scaling_reg = 0.01 * scaling.prod(dim=1).mean() (train.py:174) natively --
value and gradient from two small launches, with no host synchronisation (the framework's prod backward looks for zeros on the host);
without it `loss` is the image part alone.  `terms` also carries Ll1, depth_loss, ssim_loss, raydrop_loss, grad_loss and scaling_reg
(detached scalars, as train.py logs them).  The loss value and its gradients w.r.t. image, depth and scaling come out of the forward
calls (this is the root of the graph); autograd only scales the stored gradients by the incoming one.  HIP tensors only; no framework
fallback."""
import ctypes as C

import torch

from diff_lidargs_rasterization import _C as _base

_lib = _base._lib
_lib.lidargs_image_loss.restype = C.c_int
_lib.lidargs_loss_scratch_bytes.restype = C.c_size_t
_lib.lidargs_scaling_reg.restype = C.c_int
_lib.lidargs_scaling_reg_scratch_bytes.restype = C.c_size_t


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, depth, gt, lambda_dssim, scaling, scaling_weight):
        _base._require_device(image, "image")
        dev = image.device
        H, W = int(image.shape[-2]), int(image.shape[-1])
        if image.shape[0] != 2 or gt.shape[0] != 3 or depth.numel() != H * W:
            raise RuntimeError("image_loss: expected image [2,H,W], depth [1,H,W], gt_image [3,H,W]")
        f32 = lambda t: t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous()
        img, dep, g = f32(image), f32(depth), f32(gt)
        losses = torch.empty(7, dtype=torch.float32, device=dev)        # loss, Ll1, depth, ssim, raydrop, grad, scaling_reg
        g_image = torch.empty_like(img)
        g_depth = torch.empty_like(dep)
        nb = int(_lib.lidargs_loss_scratch_bytes(C.c_int(H), C.c_int(W)))
        M, nr = 0, 0
        if scaling is not None:
            _base._require_device(scaling, "scaling")
            if scaling.dim() != 2 or scaling.shape[1] != 3:
                raise RuntimeError("image_loss: expected scaling [M,3]")
            M = int(scaling.shape[0])
            nr = int(_lib.lidargs_scaling_reg_scratch_bytes(C.c_int(M)))
        scratch = torch.empty(max(nb, nr), dtype=torch.uint8, device=dev)
        p = _base._ptr
        g_scaling = None
        with torch.cuda.device(dev):
            rc = _lib.lidargs_image_loss(C.c_int(H), C.c_int(W), p(img), p(dep), p(g), C.c_float(float(lambda_dssim)), p(losses), p(g_image),
                                         p(g_depth), p(scratch), C.c_size_t(nb), _base._stream(dev))
            if rc < 0:
                _base._raise(rc, "lidargs_image_loss")
            if scaling is not None:
                sc = f32(scaling)
                g_scaling = torch.empty_like(sc)
                # (the same scratch: the stream runs the loss's launches first)
                rc = _lib.lidargs_scaling_reg(C.c_int(M), p(sc), C.c_float(float(scaling_weight)), p(losses[6:]), p(losses), p(g_scaling), p(scratch),
                                              C.c_size_t(max(nb, nr)), _base._stream(dev))
                if rc < 0:
                    _base._raise(rc, "lidargs_scaling_reg")
            else:
                losses[6] = 0.0
        ctx.save_for_backward(g_image, g_depth, g_scaling if g_scaling is not None else g_depth.new_empty(0))
        ctx.shapes = (image.shape, depth.shape, None if scaling is None else scaling.shape)
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), losses

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        g_image, g_depth, g_scaling = ctx.saved_tensors
        gs = (g_scaling * g_loss).view(ctx.shapes[2]) if ctx.shapes[2] is not None else None
        return (g_image * g_loss).view(ctx.shapes[0]), (g_depth * g_loss).view(ctx.shapes[1]), None, None, gs, None


def image_loss(image, depth, gt_image, lambda_dssim=0.2, scaling=None, scaling_weight=0.01):
    loss, terms = _ImageLoss.apply(image, depth, gt_image, float(lambda_dssim), scaling, float(scaling_weight))
    return dict(loss=loss, Ll1=terms[1], depth_loss=terms[2], ssim_loss=terms[3], raydrop_loss=terms[4], grad_loss=terms[5], scaling_reg=terms[6])
