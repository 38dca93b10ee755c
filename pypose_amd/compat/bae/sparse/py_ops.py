"""``diagonal_op_(A, op)``: apply ``op`` to the diagonal of a sparse CSR matrix, in place
(pypose/optim/optimizer.py:643 clamps it, :664 scales it by 1 + damping)."""
import torch


def diagonal_op_(A, op):
    crow, col, val = A.crow_indices(), A.col_indices(), A.values()
    n = A.shape[0]
    row = torch.repeat_interleave(torch.arange(n, device=col.device), crow[1:] - crow[:-1])
    on = (row == col).nonzero().reshape(-1)
    d = val[on]
    res = op(d)
    val[on] = d if res is None else res
    return A
