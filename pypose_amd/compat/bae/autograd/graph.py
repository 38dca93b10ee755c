"""``jacobian(R, params)``: one sparse [R.numel(), update-width of the parameter] matrix per parameter
(what pypose/optim/optimizer.py:637-642 concatenates and turns into CSR)."""
import torch

from .function import _TAPE, TrackingTensor


def _update_width(p):
    lt = getattr(p, "ltype", None)
    return int(lt.manifold[0]) if lt is not None else (p.shape[-1] if p.dim() else 1)     # optimizer.py:44-49


def jacobian(R, params):
    params = list(params)
    with torch.enable_grad():
        Rt = R.tensor() if isinstance(R, TrackingTensor) else torch.Tensor.as_subclass(R, torch.Tensor)
    dr = Rt.shape[-1] if Rt.dim() else 1
    E = Rt.numel() // dr
    with torch.enable_grad():
        Rm = Rt.reshape(E, dr)
    events = [(root, idx, out) for root, idx, out in _TAPE if any(root is p for p in params)]
    del _TAPE[:]
    dev, dt = Rt.device, Rt.dtype
    pieces = {id(p): ([], [], []) for p in params}
    if isinstance(R, TrackingTensor) and any(R is p for p in params):       # the residual IS the parameter: identity
        n = R.numel()
        ar = torch.arange(n, device=dev)
        r, c, v = pieces[id(R)]
        r.append(ar), c.append(ar), v.append(torch.ones(n, dtype=dt, device=dev))
    elif events and Rt.requires_grad:
        outs = [out for _, _, out in events]
        with torch.enable_grad():
            for k in range(dr):
                grads = torch.autograd.grad(Rm[:, k].sum(), outs, retain_graph=True, allow_unused=True)
                for (root, idx, out), g in zip(events, grads):
                    if g is None:
                        continue
                    m = _update_width(root)
                    g = torch.Tensor.as_subclass(g, torch.Tensor).reshape(-1, out.shape[-1] if out.dim() else 1)[:, :m]
                    rows_n = g.shape[0]
                    if rows_n != E:                      # not one gathered row per residual row: outside the traced structure
                        raise RuntimeError("bae stand-in: a residual row must depend on exactly one gathered row per gather")
                    idx_k = torch.arange(E, device=dev) if idx is None else idx
                    keep = idx_k >= 0
                    e = torch.arange(E, device=dev)
                    rr = (e * dr + k).unsqueeze(-1).expand(E, m)
                    cc = idx_k.clamp_min(0).unsqueeze(-1) * m + torch.arange(m, device=dev)
                    r, c, v = pieces[id(root)]
                    r.append(rr[keep].reshape(-1)), c.append(cc[keep].reshape(-1)), v.append(g[keep].reshape(-1))
    out = []
    for p in params:
        rows = p.shape[0] if p.dim() >= 2 else p.numel()
        ncol = rows * _update_width(p) if p.dim() >= 2 else p.numel()
        r, c, v = pieces[id(p)]
        if r:
            J = torch.sparse_coo_tensor(torch.stack([torch.cat(r), torch.cat(c)]), torch.cat(v), (E * dr, ncol), dtype=dt, device=dev).coalesce()
        else:
            J = torch.sparse_coo_tensor(torch.zeros((2, 0), dtype=torch.int64, device=dev), torch.zeros(0, dtype=dt, device=dev), (E * dr, ncol))
        out.append(J)
    return out
