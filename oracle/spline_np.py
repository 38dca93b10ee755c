"""ORACLE (test infrastructure only -- never imported by the product) -- numpy restatement of the two
interpolators of pypose/function/spline.py.  Pinned by tests/golden/spline_golden.npz, which
tests/golden/make_spline_golden.py records from the real reference."""
import numpy as np

from oracle import lie_np


def chspline(points, interval=0.1):
    """Cubic Hermite spline (spline.py:73-102): samples at j + k*interval plus the closing knot; a sample sitting
    on knot j > 0 is evaluated at the END (t = 1) of segment j-1 (searchsorted on the interior knots, :82)."""
    N = points.shape[-2]
    dt = points.dtype
    steps = np.arange(0, 1, interval).astype(dt)
    knots = np.arange(N).astype(dt)
    times = (knots[:, None] + steps).reshape(-1)[:-(len(steps) - 1)]
    seg = np.searchsorted(knots[1:], times, side="left")
    t = (times - knots[seg]) / (knots[seg + 1] - knots[seg])
    m = points[..., 1:, :] - points[..., :-1, :]                           # :85-86 (unit spacing)
    m = np.concatenate([m[..., :1, :], (m[..., 1:, :] + m[..., :-1, :]) / 2, m[..., -1:, :]], -2)     # :87
    A = np.array([[1, 0, -3, 2], [0, 1, -2, 1], [0, 0, 3, -2], [0, 0, -1, 1]], dtype=dt)
    hh = (A @ np.stack([t ** 0, t, t ** 2, t ** 3])).T.astype(dt)         # :90-96
    out = hh[:, 0:1] * points[..., seg, :] + hh[:, 1:2] * m[..., seg, :]
    out = out + hh[:, 2:3] * points[..., seg + 1, :] + hh[:, 3:4] * m[..., seg + 1, :]
    return out.astype(dt)


def bspline_weights(interval, dtype):
    u = np.arange(0, 1, interval).astype(dtype)
    B = (np.array([[5, 3, -3, 1], [1, 3, 3, -2], [0, 0, 0, 1]], dtype=dtype) / 6).astype(dtype)   # :208-210
    return (B @ np.stack([u ** 0, u, u ** 2, u ** 3])).astype(dtype), B.sum(1).astype(dtype)      # :212, :216


def bspline(data, interval=0.1, extrapolate=False):
    """Cumulative SE3 B-spline (spline.py:194-225) for data [..., N, 7]: per segment i the three relative twists
    xi_j = Log(P_{i+j}^-1 P_{i+j+1}) (:215), A_j = Exp(w_j(u) xi_j) (:216), pose = P_i * ((A_0 A_1) A_2) (:219-221),
    and one closing pose per trajectory from the last segment at u = 1 (:217-218, :221-224)."""
    if extrapolate:
        data = np.concatenate([np.repeat(data[..., :1, :], 2, -2), data, np.repeat(data[..., -1:, :], 2, -2)], -2)   # :196-199
    batch, N = data.shape[:-2], data.shape[-2]
    flat = data.reshape(-1, N, 7)
    w, wend = bspline_weights(interval, data.dtype)
    K = w.shape[1]
    out = np.zeros((flat.shape[0], (N - 3) * K + 1, 7), dtype=data.dtype)

    def mul(a, b):
        return lie_np.se3_mul_fwd(a, b)[0]

    for i in range(N - 3):
        xi = [lie_np.se3_log_fwd(mul(lie_np.se3_inv_fwd(flat[:, i + j])[0], flat[:, i + j + 1]))[0] for j in range(3)]
        for k in range(K):
            A = [lie_np.se3_exp_fwd((xi[j] * w[j, k]).astype(data.dtype))[0] for j in range(3)]
            out[:, i * K + k] = mul(flat[:, i], mul(mul(A[0], A[1]), A[2]))
        if i == N - 4:
            A = [lie_np.se3_exp_fwd((xi[j] * wend[j]).astype(data.dtype))[0] for j in range(3)]
            out[:, -1] = mul(flat[:, i], mul(mul(A[0], A[1]), A[2]))
    return out.reshape(batch + (out.shape[1], 7))
