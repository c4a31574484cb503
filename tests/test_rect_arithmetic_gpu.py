"""The tile rect of a Gaussian -- getRect_lidar, R3/cr/auxiliary.h:80-92 and R2/cr/auxiliary.h:99-112 -- bit for bit on ADVERSARIAL
inputs.  The rect is integer output of fp32 arithmetic: `(int)((p.x - rx) / 16)`, `(int)((p.x + rx + 16 - 1) / 16)`, `round(p.y -+ ry)`.
A random scene puts about one Gaussian in 1e8 within an ulp of one of these boundaries (the `+ 15.f` defect of round 3 survived 20 k
random scenes and was caught by one Gaussian of one frame); here every input sits within +-3 ulps of one, for every
tile edge up to column 4800, every column radius up to 40, and every half-integer row of a 1100-beam image.

The device side is the SAME `__device__` function the two preprocess kernels call (lidargs_common.h rect_lidar / rect_surfel), reached
through the test hook `lidargs_debug_rects` of the C ABI; the other side is the oracle's get_rect_lidar / sf_get_rect."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulps(x, k):
    """x moved by k ulps (k in -3..3), fp32."""
    x = np.asarray(x, np.float32).copy()
    for step in range(1, 4):
        up, down = k >= step, k <= -step
        x[up] = np.nextafter(x[up], np.float32(np.inf)); x[down] = np.nextafter(x[down], np.float32(-np.inf))
    return x


def _adversarial(rng, n, W, H):
    """(p.x, p.y, rx, ry): p.x +- rx on a multiple of 16 or on one minus 15 (the upper bound's boundary), p.y +- ry on a half-integer,
    each then moved by -3..3 ulps."""
    tiles_x = (W + 15) // 16
    rx = rng.integers(1, 41, n).astype(np.int32); ry = rng.integers(1, 41, n).astype(np.int32)
    edge = 16.0 * rng.integers(0, tiles_x + 1, n)
    mode = rng.integers(0, 4, n)
    px = np.where(mode == 0, edge + rx,                         # p.x - rx on a tile edge (xmin truncation)
         np.where(mode == 1, edge + 1 - rx,                     # p.x + rx + 15 on a multiple of 16 (xmax truncation, incl. the tie at 2^k)
         np.where(mode == 2, edge, rng.random(n) * W)))         # the centre on an edge; anywhere
    half = rng.integers(0, H + 1, n) + 0.5
    py = np.where(rng.integers(0, 3, n) == 0, half + ry, np.where(rng.integers(0, 2, n) == 0, half - ry, half))
    px = _ulps(px.astype(np.float32), rng.integers(-3, 4, n)); py = _ulps(py.astype(np.float32), rng.integers(-3, 4, n))
    return np.stack([px, py], 1).astype(np.float32), np.stack([rx, ry], 1).astype(np.int32)


def _device_rects(p, r, tiles_x, tiles_y, surfel):
    import torch
    from diff_lidargs_rasterization import _C as binding
    lib = binding._lib
    lib.lidargs_debug_rects.restype = C.c_int
    tp = torch.from_numpy(p).cuda(); tr = torch.from_numpy(r).cuda()
    out = torch.empty((p.shape[0], 4), dtype=torch.int32, device="cuda")
    rc = lib.lidargs_debug_rects(C.c_int(p.shape[0]), C.c_int(int(surfel)), C.c_void_p(tp.data_ptr()), C.c_void_p(tr.data_ptr()),
                                 C.c_int(tiles_x), C.c_int(tiles_y), C.c_void_p(out.data_ptr()),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, binding._err()
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("surfel", [False, True], ids=["3d", "surfel"])
@pytest.mark.parametrize("W,H", [(31, 16), (2650, 64), (4800, 1100)])
def test_rects_bit_exact_on_every_boundary(W, H, surfel, hip_lib_built):
    from oracle import lgo
    rng = np.random.default_rng(W * 7 + H + int(surfel))
    p, r = _adversarial(rng, 2_000_000, W, H)
    tiles_x = (W + 15) // 16
    dev = _device_rects(p, r, tiles_x, H, surfel)
    ref = lgo.rects(p, r, tiles_x, H, surfel=surfel)
    bad = np.nonzero((dev != ref).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} rects differ; first: p={p[bad[0]]!r} r={r[bad[0]]} device={dev[bad[0]]} oracle={ref[bad[0]]}"


def test_the_tie_that_round_3_missed(hip_lib_built):
    """p.x = 16 - 2 ulps, rx = 1: 16.999998 + 16 ties up to 33, minus 1 is 32, the rect reaches tile 1 (xmax = 2)."""
    from oracle import lgo
    p = np.array([[15.999998092651367, 9.2], [47.99999618530273, 9.2], [111.99999237060547, 9.2]], np.float32)
    r = np.array([[1, 1]] * 3, np.int32)
    ref = lgo.rects(p, r, 300, 16)
    assert list(ref[:, 2]) == [2, 4, 8]
    for surfel in (False, True):
        assert np.array_equal(_device_rects(p, r, 300, 16, surfel)[:, 2], ref[:, 2])
