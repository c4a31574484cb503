"""Generate tests/golden/points_meter_golden.npz by EXECUTING the reference's own code for PointsMeter.update
(utils/lidar_utils.py:253-282) wherever it can run here.

Run in the authoring container only (needs /root/reference):

    python tests/golden/make_points_meter_golden.py

Executed reference code (nothing of its source text is written to this repository; the .npz holds inputs and outputs only):
  pano_to_lidar_with_intensities, pano_to_lidar   utils/lidar_utils.py:171-232   (pulled out of the file's AST, as make_rangeview_golden.py does:
                                                                                   the module itself cannot be imported)
  fscore                                          extern/fscore.py:4-18          (the file is executed as it stands)
  preds / scale, dist1.mean() + dist2.mean()      utils/lidar_utils.py:255-256, :274 -- torch float32, written out here as the two lines they are
NOT executed: chamfer_3DDist (extern/chamfer3D: CUDA).  Its output -- for every point the squared distance to its nearest neighbour in the
other cloud, fp32 `dx*dx + dy*dy + dz*dz` -- comes from oracle/chamfer3d.py; the fixture records that.
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", ".."), os.path.join(HERE, "..", "..", "lidar-gs_amd")]
import lidargs_scenes as sc  # noqa: E402
from oracle import chamfer3d  # noqa: E402

REF = "/root/reference/utils/lidar_utils.py"


def reference_functions():
    tree = ast.parse(open(REF).read(), REF)
    want = ["pano_to_lidar_with_intensities", "pano_to_lidar"]
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert sorted(n.name for n in body) == sorted(want)
    ns = {"np": np}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    fs = {"torch": torch}
    exec(compile(open("/root/reference/extern/fscore.py").read(), "/root/reference/extern/fscore.py", "exec"), fs)
    ns["fscore"] = fs["fscore"]
    return ns


def main():
    ns = reference_functions()
    out = {}
    cases = {"a": (16, 128, "uniform", 1.0, 21), "b": (32, 200, "waymo", 1.0, 22), "c": (16, 96, "uniform", 2.5, 23), "d": (8, 64, "fov", 1.0, 24)}
    for tag, (H, W, table, scale, seed) in cases.items():
        rng = np.random.default_rng(seed)
        truth = rng.gamma(2.0, 9.0, size=(H, W)).astype(np.float32) + 2.0
        truth[rng.random((H, W)) < 0.15] = 0.0                                   # dropped rays
        pred = (truth * (1.0 + 0.01 * rng.normal(size=(H, W))) + 0.05 * rng.normal(size=(H, W))).astype(np.float32)
        pred[rng.random((H, W)) < 0.1] = 0.0
        pred[truth == 0.0] = np.where(rng.random((H, W)) < 0.5, 0.0, 7.0)[truth == 0.0]
        pred = (pred * scale).astype(np.float32); truth = (truth * scale).astype(np.float32)
        if table == "fov":
            beams, kw = None, dict(lidar_K=(2.0, 26.9))
        else:
            beams = np.ascontiguousarray(sc.beam_table(H, table), dtype=np.float32)  # what train.py:354 passes: a float32 array
            kw = dict(beam_inclinations=beams)
        p = (torch.from_numpy(pred)[None] / scale).numpy(); t = (torch.from_numpy(truth)[None] / scale).numpy()      # :255-258
        pred_lidar = ns["pano_to_lidar"](pano=p[0], **kw); gt_lidar = ns["pano_to_lidar"](pano=t[0], **kw)             # :260-265
        x1 = torch.FloatTensor(pred_lidar[None, ...]).numpy(); x2 = torch.FloatTensor(gt_lidar[None, ...]).numpy()     # :268-269
        d1, d2, _, _ = chamfer3d.forward(x1, x2)                                     # chamfer_3DDist's output, by its definition
        d1, d2 = torch.from_numpy(d1), torch.from_numpy(d2)
        cd = d1.mean() + d2.mean()                                                   # :274
        f, pr, rc = ns["fscore"](d1, d2, 0.05)                                       # :275-276
        out.update({f"{tag}_H": H, f"{tag}_W": W, f"{tag}_scale": np.float32(scale), f"{tag}_pred": pred, f"{tag}_truth": truth,
                    f"{tag}_beams": beams if beams is not None else np.zeros(0, np.float32), f"{tag}_fov": np.array(kw.get("lidar_K", (0, 0)), np.float32),
                    f"{tag}_pred_lidar": x1[0], f"{tag}_gt_lidar": x2[0],
                    f"{tag}_cd": np.float32(cd.item()), f"{tag}_fscore": np.float32(f[0].item()), f"{tag}_precision": np.float32(pr[0].item()),
                    f"{tag}_recall": np.float32(rc[0].item())})
        print(tag, H, W, x1.shape[1], x2.shape[1], float(cd), float(f[0]))
    dst = os.path.join(HERE, "points_meter_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
