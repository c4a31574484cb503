# numerics of an alternative k_preprocess build against the committed one (used by the MFMA covariance-projection experiment, profiles/r04_mfma_cov_projection_ab.txt): run once per LIDARGS_MFMA_COV setting, compare the .npz files
# bit-equality of radii and the size of the record / gradient difference between the scalar and the MFMA projection (run with LIDARGS_MFMA_COV=0 / 2 in two processes)
import os, sys, numpy as np
sys.path[:0]=["/root/repo","/root/repo/lidar-gs_amd","/root/repo/tests"]
import lidargs_scenes as sc
from util import hip_forward_backward
kind,P,H,W,seed = "street", 200000, 64, 2650, 3
scene = sc.make_scene(kind,P,H,seed); grads = sc.upstream_grads(H,W,seed)
r = hip_forward_backward(scene, W, H, grads)
np.savez(sys.argv[1], **{k: v for k, v in r.items()})
