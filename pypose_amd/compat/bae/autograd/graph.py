"""``jacobian(R, params)``: one sparse [R.numel(), update-width of the parameter] matrix per parameter
(what pypose/optim/optimizer.py:637-642 concatenates and turns into CSR)."""
import torch

from .function import _TAPE, TrackingTensor


def _update_width(p):
    lt = getattr(p, "ltype", None)
    return int(lt.manifold[0]) if lt is not None else (p.shape[-1] if p.dim() else 1)     # optimizer.py:44-49


# ---- kernel route for the relative-pose residual ----------------------------------------------------------------------------
# The reference's sparse pose-graph models (tests/optim/test_sparse_lm.py:11-13, examples/module/pgo) compute
#     R = (relpose.Inv() @ node1.Inv() @ node2).Log().tensor()          node1 / node2: gathered rows of one tracked parameter
# with the REFERENCE's autograd Functions; the graph they leave behind is  SE3_Log <- SE3_Mul(SE3_Mul(const, SE3_Inv(gather)), gather)
# (plus alias / view / expand nodes).  When jacobian() finds exactly that, the per-edge 6 x 6 blocks come from ONE launch of
# pplie_pgo_linearize (csrc/pgo_fused.hip: closed form, the kernel the fused pose-graph path uses) instead of six backward sweeps
# through ~40 eager kernels each; the residual the kernel recomputes is compared with the traced one before its blocks are used.
_PASS_THROUGH = {"AliasBackward0", "ViewBackward0", "ExpandBackward0", "ReshapeAliasBackward0", "UnsafeViewBackward0", "CloneBackward0"}
route_taken = {"last": None}          # "kernel:pgo" / "autograd": what the latest jacobian() call did (tests)


def _skip(fn):
    while fn is not None and fn.name() in _PASS_THROUGH and len(fn.next_functions) == 1:
        fn = fn.next_functions[0][0]
    return fn


def _same_node(a, b):
    return a is not None and b is not None and (a is b or a == b)


def _match_pgo(Rt, events, E, dr):
    """(event of node1, event of node2, Z^-1 rows) if Rt's history is the relative-pose residual over two gathers, else None"""
    why = route_taken.__setitem__
    if dr != 6 or len(events) < 2:
        why("why", f"dr {dr}, {len(events)} gathers")
        return None
    try:
        log = _skip(Rt.grad_fn)
        if log is None or log.name() != "SE3_LogBackward":
            why("why", f"top node {None if log is None else log.name()}")
            return None
        outer = _skip(log.next_functions[0][0])
        if outer is None or outer.name() != "SE3_MulBackward" or len(outer.next_functions) != 2:
            why("why", f"below Log: {None if outer is None else outer.name()}")
            return None
        inner, leaf2 = _skip(outer.next_functions[0][0]), _skip(outer.next_functions[1][0])
        if inner is None or inner.name() != "SE3_MulBackward" or len(inner.next_functions) != 2 or inner.next_functions[0][0] is not None:
            why("why", f"left factor: {None if inner is None else inner.name()}")
            return None                                   # (the measurement must be a constant: no history on the left factor)
        inv = _skip(inner.next_functions[1][0])
        if inv is None or inv.name() != "SE3_InvBackward":
            why("why", f"inner right factor: {None if inv is None else inv.name()}")
            return None
        leaf1 = _skip(inv.next_functions[0][0])
        bases = [_skip(out.grad_fn) for _, _, out in events]
        # (the tape also holds the gathers of earlier forward passes that nobody differentiated -- the loss evaluations of the
        #  LM loop, optimizer.py:659, 670: the two gathers of THIS residual are found by their graph nodes, latest first)
        order = range(len(bases) - 1, -1, -1)
        k1 = next((k for k in order if _same_node(bases[k], leaf1)), None)
        k2 = next((k for k in order if _same_node(bases[k], leaf2)), None)
        if k1 is None or k2 is None or k1 == k2:
            why("why", f"leaves {leaf1 and leaf1.name()} / {leaf2 and leaf2.name()} vs gathers {[b and b.name() for b in bases]}")
            return None
        saved = inner.saved_tensors                        # SE3_Mul saves its left operand (operation.py:866-869): Z^-1
        Zinv = torch.Tensor.as_subclass(saved[0], torch.Tensor).detach().reshape(-1, 7)
        if Zinv.shape[0] != E or any(torch.Tensor.as_subclass(events[k][2], torch.Tensor).reshape(-1, 7).shape[0] != E for k in (k1, k2)):
            why("why", "broadcast operand")
            return None                                    # (a broadcast operand: rows are not one-to-one)
        return k1, k2, Zinv
    except Exception as ex:
        why("why", "exception " + repr(ex))
        return None


def _pgo_blocks(events, k1, k2, Zinv, Rm):
    """residual-checked blocks d r / d node1, d r / d node2 [E, 6, 6] from pplie_pgo_linearize, or None"""
    import ctypes
    from pypose_amd import _C
    n1 = torch.Tensor.as_subclass(events[k1][2], torch.Tensor).detach().reshape(-1, 7)
    n2 = torch.Tensor.as_subclass(events[k2][2], torch.Tensor).detach().reshape(-1, 7)
    if not n1.is_cuda or n1.dtype not in (torch.float32, torch.float64) or _C._test_backend is not None:
        route_taken["why"] = "not a HIP call"
        return None
    E = n1.shape[0]
    nodes = torch.cat([n1, n2], 0).contiguous()
    e = torch.arange(E, device=n1.device)
    idx = torch.stack([e, e + E], -1).contiguous()
    (Z,) = _C.row_op("se3_inv_fwd", [Zinv.contiguous()], (7,))
    r = torch.empty((E, 6), dtype=n1.dtype, device=n1.device)
    J = torch.empty((E, 2, 6, 6), dtype=n1.dtype, device=n1.device)
    sig = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_void_p]
    fn = _C.library().symbol("pplie_pgo_linearize" + ("_f32" if n1.dtype == torch.float32 else "_f64"), sig)
    with _C._on_device(n1.device):
        _C.check(fn(nodes.data_ptr(), idx.data_ptr(), Z.data_ptr(), r.data_ptr(), J.data_ptr(), E, _C.stream_ptr(n1.device)),
                 "pplie_pgo_linearize")
    # (both sides evaluate the same function in the operands' precision, in different association orders -- the kernel forms
    #  n1^-1 n2 first, the traced chain (Z^-1 n1^-1) n2: the difference scales with eps and the size of the translations)
    eps = 4e-5 if n1.dtype == torch.float32 else 1e-11      # (a guard against a mis-recognised program: gross differences, not ulps)
    ref = Rm.detach()
    scale = torch.stack([ref.abs().max(), nodes[:, :3].abs().max(), Zinv[:, :3].abs().max(), ref.new_ones(())]).max()
    if not bool((r - ref).abs().max() <= eps * scale):
        route_taken["why"] = f"residual check: {float((r - ref).abs().max()):.3e}"
        return None                                        # not the function we took it for: the autograd sweeps decide
    return J[:, 0], J[:, 1]


def _kernel_pieces(Rt, Rm, events, pieces, E, dr, dev):
    """fill `pieces` (row, column, value lists per parameter) from the kernel route; False if it does not apply"""
    m = _match_pgo(Rt, events, E, dr)
    if m is None:
        return False
    k1, k2, Zinv = m
    blocks = _pgo_blocks(events, k1, k2, Zinv, Rm)
    if blocks is None:
        return False
    if any(_update_width(events[k][0]) != 6 for k in (k1, k2)):
        return False
    e = torch.arange(E, device=dev)
    for k, Jk in zip((k1, k2), blocks):
        root, idx, out = events[k]
        mw = 6
        idx_k = e if idx is None else idx
        keep = idx_k >= 0                                   # (rows of a fixed block concatenated in front: no column)
        rr = (e.view(E, 1, 1) * dr + torch.arange(dr, device=dev).view(1, dr, 1)).expand(E, dr, mw)
        cc = (idx_k.clamp_min(0).view(E, 1, 1) * mw + torch.arange(mw, device=dev).view(1, 1, mw)).expand(E, dr, mw)
        r, c, v = pieces[id(root)]
        r.append(rr[keep].reshape(-1)), c.append(cc[keep].reshape(-1)), v.append(Jk[keep].reshape(-1))
    return True


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
    elif events and Rt.requires_grad and _kernel_pieces(Rt, Rm, events, pieces, E, dr, dev):
        route_taken["last"] = "kernel:pgo"
    elif events and Rt.requires_grad:
        route_taken["last"] = "autograd"
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
