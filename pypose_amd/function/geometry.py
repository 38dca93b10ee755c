"""Camera geometry either side of the bundle-adjustment path.

Host-side mirror of the five pinhole helpers of pypose/function/geometry.py (cart2homo :8, homo2cart :37,
point2pixel :60, pixel2point :115, reprojerr :171): same names, argument meaning, broadcasting and assertion
messages.  With extrinsics, ``point2pixel`` / ``reprojerr`` are ONE kernel per direction (csrc/reproj.hip): the forward
``pplie_se3_reproj_lin`` returns the residual together with its closed-form Jacobian blocks [d r/d pose | d r/d point],
the backward ``pplie_reproj_vjp`` contracts them -- and the LM optimizers take the blocks as they are instead of
sweeping the autograd graph (``closed_form_blocks`` below, optim/multigraph.py).  Without extrinsics the pinhole
algebra is plain tensor arithmetic on whatever device the inputs live on.
"""
import torch

from ..lietensor import LieTensor
from ..lietensor import operation as _op

_ReprojVjp = _op._make_bwd("ReprojVjp", "reproj_vjp", (18, 2), (7, 3))
_closed_recorders = []          # optim: active recorders of (residual, inputs, closed-form blocks)


class _Reproj(torch.autograd.Function):
    """r = homo2cart(K (X . p)) - pixel for broadcastable (pose [...,7], point [...,3], cam [...,11]); saves J [...,18]."""

    @staticmethod
    def forward(X, p, cam):
        r, J = _op._launch("se3_reproj_lin", (X, p, cam), (7, 3, 11), (2, 18))
        return r, J

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.save_for_backward(*inputs, output[1])
        ctx.mark_non_differentiable(output[1])
        ctx.shapes = [tuple(t.shape) for t in inputs]

    @staticmethod
    def backward(ctx, g, _gJ):
        X, p, cam, J = ctx.saved_tensors
        gX, gp = _ReprojVjp.apply(J, g)
        gcam = None
        if ctx.needs_input_grad[2]:
            # intrinsics / observed pixel (rarely optimised): d r/d K = (d pix/d h) (x) q, d r/d pixel = -I
            q = _op._launch("se3_act_fwd", (X, p), (7, 3), (3,))[0]
            K = cam[..., :9].reshape(cam.shape[:-1] + (3, 3))
            h = (K @ q.unsqueeze(-1)).squeeze(-1)
            hz = h[..., 2:]
            tiny = torch.finfo(h.dtype).tiny
            clamped = hz.abs() < tiny
            den = torch.where(hz < 0, -torch.ones_like(hz), torch.ones_like(hz)) * hz.abs().clamp(min=tiny)
            pix = h[..., :2] / den
            gh = torch.cat([g / den, -(pix * g).sum(-1, keepdim=True) / den * (~clamped)], -1)
            gcam = torch.cat([(gh.unsqueeze(-1) * q.unsqueeze(-2)).reshape(gh.shape[:-1] + (9,)), -g], -1)
        out = []
        for grad, shape in zip((gX, gp, gcam), ctx.shapes):
            out.append(None if grad is None else grad.sum_to_size(shape) if tuple(grad.shape) != shape else grad)
        return tuple(out)

    @staticmethod
    def vmap(info, in_dims, X, p, cam):
        X, p, cam = _op._fold_vmap(in_dims, (X, p, cam))
        return _Reproj.apply(X, p, cam), (0, 0)


def _reproject(points, intrinsics, extrinsics, pixels):
    """(..., N, 2) residual through the fused kernels; the closed-form blocks are offered to an active optimizer trace."""
    X = extrinsics.tensor().unsqueeze(-2)                                           # (..., 1, 7): one pose, N points
    K9 = intrinsics.reshape(intrinsics.shape[:-2] + (1, 9))
    lead = torch.broadcast_shapes(points.shape[:-1], X.shape[:-1], K9.shape[:-1], pixels.shape[:-1] if pixels is not None else ())
    uv = pixels if pixels is not None else points.new_zeros(lead + (2,))
    cam = torch.cat([K9.expand(lead + (9,)).to(points.dtype), uv.expand(lead + (2,))], -1)
    r, J = _Reproj.apply(X, points, cam)
    if _closed_recorders:
        Jm = J.reshape(J.shape[:-1] + (2, 9))
        for rec in _closed_recorders:
            rec.note_closed(r, [(extrinsics, Jm[..., :6]), (points, Jm[..., 6:9])],
                            blockers=[t for t in (intrinsics, pixels) if t is not None and t.requires_grad])
    return r


def _need(cond, msg):
    assert cond, msg


def _is_k(intrinsics):
    return intrinsics.shape[-2:] == (3, 3)


def cart2homo(coordinates):
    """``(*, D) -> (*, D + 1)``: append a one."""
    return torch.cat([coordinates, coordinates.new_ones(coordinates.shape[:-1] + (1,))], dim=-1)


def homo2cart(coordinates):
    """``(*, D + 1) -> (*, D)``: divide by the last component; a zero (of either sign) is replaced by the
    smallest normal number with the sign convention ``pm(0) = +1`` of the reference."""
    w = coordinates[..., -1:]
    tiny = torch.finfo(coordinates.dtype).tiny
    sign = torch.where(w < 0, -torch.ones_like(w), torch.ones_like(w))
    return coordinates[..., :-1] / (sign * w.abs().clamp(min=tiny))


def point2pixel(points, intrinsics, extrinsics=None):
    """Pixels ``(*, N, 2)`` of ``points (*, N, 3)`` seen through ``intrinsics (*, 3, 3)``; with ``extrinsics``
    (SE3 ``(*, 7)``, world -> camera) the points are transformed first."""
    _need(points.size(-1) == 3, "Points shape incorrect")
    _need(_is_k(intrinsics), "Intrinsics shape incorrect.")
    lead = [points.shape[:-2], intrinsics.shape[:-2]]
    if extrinsics is not None:
        _need(isinstance(extrinsics, LieTensor) and extrinsics.shape[-1] == 7, "Type incorrect.")
        lead.append(extrinsics.shape[:-1])
    torch.broadcast_shapes(*lead)                                      # raises on incompatible batch dims
    if extrinsics is not None:
        return _reproject(points, intrinsics, extrinsics, None)
    return homo2cart(points @ intrinsics.mT)


def pixel2point(pixels, depth, intrinsics):
    """Camera-frame points ``(*, N, 3)`` of ``pixels (*, N, 2)`` at ``depth (*, N)``."""
    _need(pixels.size(-1) == 2, "Pixels shape incorrect")
    _need(depth.size(-1) == pixels.size(-2), "Depth shape does not match pixels")
    _need(_is_k(intrinsics), "Intrinsics shape incorrect.")
    focal = torch.stack([intrinsics[..., 0, 0], intrinsics[..., 1, 1]], dim=-1)
    centre = torch.stack([intrinsics[..., 0, 2], intrinsics[..., 1, 2]], dim=-1)
    _need(not torch.any(focal[..., 0] == 0), "fx Cannot contain zero")
    _need(not torch.any(focal[..., 1] == 0), "fy Cannot contain zero")
    xy = (pixels - centre.unsqueeze(-2)) * depth.unsqueeze(-1) / focal.unsqueeze(-2)
    return torch.cat([xy, depth.unsqueeze(-1)], dim=-1)


def reprojerr(points, pixels, intrinsics, extrinsics=None, reduction='none'):
    """Projected minus observed pixels: ``'none'`` -> ``(*, N, 2)``; ``'norm'`` / ``'sum'`` reduce the last dim."""
    torch.broadcast_shapes(points.shape[:-2], pixels.shape[:-2], intrinsics.shape[:-2])
    _need(points.size(-1) == 3 and pixels.size(-1) == 2 and _is_k(intrinsics), "Shape not compatible.")
    _need(reduction in {'norm', 'sum', 'none'}, "Reduction method can only be 'norm'|'sum'|'none'.")
    if extrinsics is not None:
        _need(isinstance(extrinsics, LieTensor) and extrinsics.shape[-1] == 7, "Type incorrect.")
        torch.broadcast_shapes(points.shape[:-2], intrinsics.shape[:-2], extrinsics.shape[:-1])
        err = _reproject(points, intrinsics, extrinsics, pixels)
    else:
        err = point2pixel(points, intrinsics) - pixels
    return {'none': lambda e: e, 'norm': lambda e: e.norm(dim=-1), 'sum': lambda e: e.sum(dim=-1)}[reduction](err)


def _centred(points):
    centre = points.mean(dim=-2, keepdim=True)
    return points - centre, centre


def svdtf(source, target):
    """Rigid transform (SE3) that best maps the point cloud ``source [..., N, 3]`` onto ``target [..., N, 3]`` in the
    least-squares sense, by the SVD of the cross-covariance (Kabsch; reference geometry.py:315-358).  A reflection
    (det = -1) is flipped to the nearest rotation the way the reference does (R -> -R)."""
    from ..lietensor.convert import mat2SE3
    assert source.size(-2) == target.size(-2), {"The number of points N has to be the same for both point clouds."}
    src, c_src = _centred(source)
    tgt, c_tgt = _centred(target)
    U, _, Vh = torch.linalg.svd(torch.einsum('...Na, ...Nb -> ...ab', tgt, src))
    R = U @ Vh
    mirrored = (R.det() + 1).abs() < 1e-6
    R[mirrored] = -R[mirrored]
    t = c_tgt.mT - R @ c_src.mT
    return mat2SE3(torch.cat((R, t), dim=-1), check=False)


def svdstf(source, target, with_scale=True):
    """Similarity transform (Sim3: scale, rotation, translation) that best maps ``source [..., N, 3]`` onto
    ``target [..., N, 3]`` (Umeyama 1991; reference geometry.py:361-433); ``with_scale=False`` fixes the scale to 1."""
    from ..lietensor.convert import mat2Sim3
    assert source.size(-2) == target.size(-2), {"The number of points N has to be the same for both point clouds."}
    assert source.size(-1) == 3, {"The source point dim should be 3"}
    assert target.size(-1) == 3, {"The target point dim should be 3"}
    N, m = source.shape[-2:]
    src, c_src = _centred(source)
    tgt, c_tgt = _centred(target)
    U, D, Vh = torch.linalg.svd(tgt.transpose(-2, -1) @ src / N)
    flip = torch.eye(m, dtype=U.dtype, device=U.device).expand_as(U).clone()      # diag(1, 1, det(U Vh)): proper rotation
    flip[..., -1, -1] = torch.sign(torch.det(U @ Vh))
    if with_scale:
        spread = (src.norm(dim=-1) ** 2).mean(dim=-1, keepdim=True)
        scale = torch.sum(torch.diagonal(flip, dim1=-1, dim2=-2) * D, keepdim=True, dim=-1) / spread
    else:
        scale = torch.ones_like(D[..., 0:1])
    scale = scale.unsqueeze(-1)
    R = U @ flip @ Vh
    t = c_tgt.transpose(-2, -1) - scale * R @ c_src.transpose(-2, -1)
    return mat2Sim3(torch.cat((scale * R, t), dim=-1), check=True)
