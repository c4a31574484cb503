"""CPU restatement (numpy, fp32) of the reference's chamfer3D extension:
/root/reference/extern/chamfer3D/chamfer3D.cu (NmDistanceKernel :8-138, NmDistanceGradKernel :167-195).

TEST INFRASTRUCTURE ONLY.  PARITY STATUS: "parity unpinned" for the CUDA arithmetic (the extension is CUDA, cannot be built here;
the reference ships no test vectors for it).  The algorithm is unambiguous: for every point the squared distance to, and the index
of, its nearest neighbour in the other cloud -- d = dx*dx + dy*dy + dz*dz evaluated in fp32 as written, ties and 512-point batches
resolved towards the LOWEST index (strict `<` inside a batch, strict `>` across batches)."""
import numpy as np

F32 = np.float32


def nearest(a, b, chunk=2048):
    """a [n,3], b [m,3] fp32 -> (dist [n] fp32, idx [n] int32)."""
    a, b = np.ascontiguousarray(a, F32), np.ascontiguousarray(b, F32)
    n = a.shape[0]
    dist, idx = np.empty(n, F32), np.empty(n, np.int32)
    for s in range(0, n, chunk):
        q = a[s:s + chunk]
        dx = b[None, :, 0] - q[:, None, 0]; dy = b[None, :, 1] - q[:, None, 1]; dz = b[None, :, 2] - q[:, None, 2]    # chamfer3D.cu:36-38
        d = ((dx * dx).astype(F32) + (dy * dy).astype(F32)).astype(F32) + (dz * dz).astype(F32)                      # :39, fp32, left to right
        d = d.astype(F32)
        i = np.argmin(d, axis=1)                                          # first minimum = lowest index
        idx[s:s + chunk] = i
        dist[s:s + chunk] = d[np.arange(q.shape[0]), i]
    return dist, idx


def forward(xyz1, xyz2):
    """xyz1 [B,n,3], xyz2 [B,m,3] -> dist1 [B,n], dist2 [B,m], idx1, idx2 (chamfer_cuda_forward :141-166)."""
    r = [(nearest(x1, x2), nearest(x2, x1)) for x1, x2 in zip(xyz1, xyz2)]
    return (np.stack([p[0][0] for p in r]), np.stack([p[1][0] for p in r]), np.stack([p[0][1] for p in r]), np.stack([p[1][1] for p in r]))


def backward(xyz1, xyz2, g1, g2, idx1, idx2):
    """Gradients of sum(g1*dist1) + sum(g2*dist2) (:167-226): g*2*(p - q) to the point, the negative to its neighbour."""
    gx1, gx2 = np.zeros(xyz1.shape, np.float64), np.zeros(xyz2.shape, np.float64)
    for b in range(xyz1.shape[0]):
        d = (xyz1[b] - xyz2[b][idx1[b]]).astype(np.float64) * (2.0 * g1[b].astype(np.float64))[:, None]
        gx1[b] += d
        np.add.at(gx2[b], idx1[b], -d)
        d = (xyz2[b] - xyz1[b][idx2[b]]).astype(np.float64) * (2.0 * g2[b].astype(np.float64))[:, None]
        gx2[b] += d
        np.add.at(gx1[b], idx2[b], -d)
    return gx1.astype(F32), gx2.astype(F32)
