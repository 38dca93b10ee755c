"""Geodesic rotation loss (host-side mirror of pypose/module/loss.py:6-83): a pure composition of hot-path ops."""
from torch.nn.modules.loss import _Loss

from ..function.checking import is_lietensor


def geodesic_loss(input, target, reduction='mean'):
    """Angle of the relative rotation ``R_input * R_target^-1`` per element, ``|Log(.)|``.  Any two LieTensors are
    accepted: ``rotation()`` extracts the SO3 part (through ``Exp`` for algebra types) (loss.py:25-38).
    ``reduction``: 'none' | 'mean' | 'sum'."""
    assert is_lietensor(input) and is_lietensor(target), "input should be LieTensor"
    assert reduction in ['none', 'mean', 'sum'], "reduction type not supported"
    delta = input.rotation() * target.rotation().Inv()
    if not delta.ltype.on_manifold:
        delta = delta.Log()
    angle = delta.norm(p='fro', dim=-1)
    if reduction == 'mean':
        return angle.mean()
    if reduction == 'sum':
        return angle.sum()
    return angle


class GeodesicLoss(_Loss):
    """Criterion form of :func:`geodesic_loss` (loss.py:41-83)."""
    __constants__ = ["reduction"]

    def __init__(self, reduction="mean"):
        super().__init__(size_average=None, reduce=None, reduction=reduction)

    def forward(self, input, target):
        return geodesic_loss(input, target, self.reduction)
