"""What does torch compute on THIS device for `float32_tensor / python_float`?  (the quotient of scene/gaussian_model.py:706, :709)
    python tools/div_convention.py
Compares it bit for bit with the IEEE quotient, with x * float32(1 / float32(s)) and with x * float32(1 / s in double)."""
import numpy as np
import torch

torch.manual_seed(0)
x = (torch.rand(4_000_000, device="cuda") - 0.5) * 120
xs = x.cpu().numpy()
for s in (0.16, 0.04, 0.01, 0.208, 0.052, 0.013, 0.0625, 0.3):
    q = (x / s).cpu().numpy()
    s32 = np.float32(s)
    ieee = xs / s32
    rec32 = xs * (np.float32(1.0) / s32)
    rec64 = xs * np.float32(1.0 / s)
    dbl = (xs.astype(np.float64) / s).astype(np.float32)
    tt = (x / torch.tensor(s, dtype=torch.float32, device="cuda")).cpu().numpy()
    print("s=%-7g torch==ieee %.6f  ==x*f32(1/f32(s)) %.6f  ==x*f32(1/s) %.6f  ==f32(x/s in f64) %.6f   tensor/tensor==ieee %.6f" % (
        s, (q == ieee).mean(), (q == rec32).mean(), (q == rec64).mean(), (q == dbl).mean(), (tt == ieee).mean()))
    r = lambda a: np.rint(a).astype(np.int32)
    print("          voxels differing from torch's: ieee %d, rec32 %d, rec64 %d" % ((r(q) != r(ieee)).sum(), (r(q) != r(rec32)).sum(), (r(q) != r(rec64)).sum()))
