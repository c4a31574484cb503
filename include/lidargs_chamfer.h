/*
 * lidargs_chamfer.h -- C ABI of the nearest-neighbour (chamfer) distance between two point clouds.
 *
 * Replaces chamfer_cuda_forward / chamfer_cuda_backward of /root/reference/extern/chamfer3D/chamfer3D.cu:141-226, the kernel
 * behind chamfer_3DDist that utils/lidar_utils.py:261-275 (PointsMeter) uses for the chamfer distance and F-score.
 * xyz1 f32[B*n*3], xyz2 f32[B*m*3]; dist1 f32[B*n] / idx1 i32[B*n]: squared distance to, and index of, the nearest point of
 * cloud 2 (lowest index on ties); dist2 / idx2 likewise for cloud 2 against cloud 1.  scratch: lidargs_chamfer_scratch_bytes(B,n,m)
 * bytes of device memory (the uniform grid of the search -- cell counts, offsets, the targets in cell order -- and the brute-force
 * fallback's per-point 64-bit merge keys; the batches run one after the other through the same area).  The result is the reference's
 * brute-force result bit for bit: the search only skips pairs that cannot be nearest (csrc/chamfer.hip).  Device pointers; returns 0 or a negative
 * LIDARGS_ERR_* code.  The backward ACCUMULATES into grad_xyz1 f32[B*n*3] / grad_xyz2 f32[B*m*3] (the caller zeroes them, as the
 * reference's binding does, dist_chamfer_3D.py:66-69).
 */
#ifndef LIDARGS_CHAMFER_H
#define LIDARGS_CHAMFER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

size_t lidargs_chamfer_scratch_bytes(int B, int n, int m);

int lidargs_chamfer_forward(int B, int n, int m, const float* xyz1, const float* xyz2, float* dist1, float* dist2, int* idx1,
                            int* idx2, char* scratch, size_t scratch_bytes, void* stream);

int lidargs_chamfer_backward(int B, int n, int m, const float* xyz1, const float* xyz2, const float* grad_dist1,
                             const float* grad_dist2, const int* idx1, const int* idx2, float* grad_xyz1, float* grad_xyz2,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif
