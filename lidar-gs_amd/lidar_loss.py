"""The per-frame image loss of LiDAR-GS training (train.py:150-203) on the fused HIP kernels of include/lidargs_loss.h.

    terms = image_loss(image, depth, gt_image, opt.lambda_dssim)      # image [2,H,W], depth [1,H,W], gt_image [3,H,W]
    loss = terms["loss"] + 0.01 * scaling.prod(dim=1).mean()          # scaling_reg stays a framework op (per Gaussian)
    loss.backward()

`terms` also carries Ll1, depth_loss, ssim_loss, raydrop_loss and grad_loss (detached scalars, as train.py logs them).
The loss value and its gradient w.r.t. image and depth come out of ONE native call (this is the root of the graph); autograd
only scales the stored gradient by the incoming one.  HIP tensors only; no framework fallback."""
import ctypes as C

import torch

from diff_lidargs_rasterization import _C as _base

_lib = _base._lib
_lib.lidargs_image_loss.restype = C.c_int
_lib.lidargs_loss_scratch_bytes.restype = C.c_size_t


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, depth, gt, lambda_dssim):
        _base._require_device(image, "image")
        dev = image.device
        H, W = int(image.shape[-2]), int(image.shape[-1])
        if image.shape[0] != 2 or gt.shape[0] != 3 or depth.numel() != H * W:
            raise RuntimeError("image_loss: expected image [2,H,W], depth [1,H,W], gt_image [3,H,W]")
        f32 = lambda t: t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous()
        img, dep, g = f32(image), f32(depth), f32(gt)
        losses = torch.empty(6, dtype=torch.float32, device=dev)
        g_image = torch.empty_like(img)
        g_depth = torch.empty_like(dep)
        nb = int(_lib.lidargs_loss_scratch_bytes(C.c_int(H), C.c_int(W)))
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        p = _base._ptr
        with torch.cuda.device(dev):
            rc = _lib.lidargs_image_loss(C.c_int(H), C.c_int(W), p(img), p(dep), p(g), C.c_float(float(lambda_dssim)), p(losses), p(g_image),
                                         p(g_depth), p(scratch), C.c_size_t(nb), _base._stream(dev))
        if rc < 0:
            _base._raise(rc, "lidargs_image_loss")
        ctx.save_for_backward(g_image, g_depth)
        ctx.shapes = (image.shape, depth.shape)
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), losses

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        g_image, g_depth = ctx.saved_tensors
        return (g_image * g_loss).view(ctx.shapes[0]), (g_depth * g_loss).view(ctx.shapes[1]), None, None


def image_loss(image, depth, gt_image, lambda_dssim=0.2):
    loss, terms = _ImageLoss.apply(image, depth, gt_image, float(lambda_dssim))
    return dict(loss=loss, Ll1=terms[1], depth_loss=terms[2], ssim_loss=terms[3], raydrop_loss=terms[4], grad_loss=terms[5])
