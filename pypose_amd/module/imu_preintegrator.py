"""IMU pre-integration (host-side mirror of pypose/module/imu_preintegrator.py:8-465).

Same constructor, buffers, ``forward(dt, gyro, acc, rot=None, gyro_cov=None, acc_cov=None,
init_state=None)`` contract and returned dictionary (``rot``, ``vel``, ``pos``, ``cov``, ``Rij``).

Two execution routes:

* fused (no gradient required, tensors on the GPU): two HIP kernels -- ``pplie_imu_integrate``
  (Exp of the gyro increments, the SO3 product scan, the velocity / position / time prefix sums
  and the state prediction in one pass, one wavefront per sequence) and ``pplie_imu_cov``
  (the 9x9 covariance by its backward recurrence, never materialising [B,F+1,9,9]);
* composed (gradients flow, or host tensors): the same algebra written with LieTensor ops, each
  of which is a HIP kernel with a custom backward.

The covariance follows the reference's CODE (a sum over suffix products, :438-464), not the
textbook recursion in its docstring -- see SURVEY.md Appendix C.
"""
import ctypes

import torch
from torch import nn

from .. import _C
from ..basics import cumprod
from ..lietensor import LieTensor, SO3, identity_SO3, so3, vec2skew
from ..lietensor import lietensor as _lt

_INT_SIG = [ctypes.c_void_p] * 8 + [ctypes.POINTER(ctypes.c_double)] + [ctypes.c_void_p] * 6 + \
           [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
_INTB_SIG = [ctypes.c_void_p] * 7 + [ctypes.POINTER(ctypes.c_double)] + [ctypes.c_void_p] * 6 + \
            [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
_COV2_SIG = [ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
             ctypes.POINTER(ctypes.c_double), ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
_COV_SIG = [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                    ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]


def _sfx(t):
    return "_f32" if t.dtype == torch.float32 else "_f64"


def _qrot(q, p):
    """rotate p by the quaternion q (xyzw), plain torch (host-side sums of the initial-state gradients only)"""
    v, w = q[..., :3], q[..., 3:]
    uv = 2 * torch.linalg.cross(v, p.expand(v.shape))
    return p + w * uv + torch.linalg.cross(v, uv)


def _native_node(mod, dt, gyro, acc, rot, r0, v0, p0):
    """The same node as _ImuIntegrate in C++ (csrc_torch/pplie_autograd.cpp ImuOp: a Python Function's backward runs on the autograd
    engine's device thread behind the GIL, ~190 us of host time per training step around 172 us of kernels) for the training case:
    gradients w.r.t. dt / gyro / acc.  None (-> the Python node) when the initial state needs a gradient, the extension is not
    built, or the operands are not plain contiguous device tensors."""
    from ..lietensor import operation as _op
    nat = _op._native()
    if nat is None or not hasattr(nat, "imu_integrate") or r0.requires_grad or v0.requires_grad or p0.requires_grad \
            or not getattr(mod, 'native_backward', True):
        return None
    ts = (dt, gyro, acc) if rot is None else (dt, gyro, acc, rot)
    if any(type(t) is not torch.Tensor or not t.is_contiguous() for t in ts) or dt.dim() != 3:
        return None
    B, F = dt.shape[:2]
    if rot is not None and rot.shape != (B, F, 4):
        return None
    r0b, v0b, p0b = mod._bcast('r0', r0, B, 4), mod._bcast('v0', v0, B, 3), mod._bcast('p0', p0, B, 3)
    g = mod.__dict__.get('_g_list')
    if g is None or g[0] is not mod.gravity or g[1] != mod.gravity._version:
        g = mod.__dict__['_g_list'] = (mod.gravity, mod.gravity._version, [float(x) for x in mod.gravity.tolist()])
    return nat.imu_integrate(dt, gyro, acc, rot, r0b.detach(), v0b.detach(), p0b.detach(), g[2],
                             _op._kernel_address("imu_integrate", dt.dtype), _op._kernel_address("imu_integrate_bwd", dt.dtype))


class _ImuIntegrate(torch.autograd.Function):
    """``integrate`` + ``predict`` (reference :314-426) as ONE autograd node: forward ``pplie_imu_integrate``, backward
    ``pplie_imu_integrate_bwd`` (one reverse pass per sequence) -- the reference's graph for the same thing is two cumsum
    nodes, Exp / Act / Inv / Mul nodes and the log2(F) rounds of its product scan."""

    @staticmethod
    def forward(ctx, mod, dt, gyro, acc, rot, r0, v0, p0):
        B, F = dt.shape[:2]
        orot, ovel, opos, held = mod._launch_integrate(dt, gyro, acc, rot, r0, v0, p0, None, None)
        ctx.mod, ctx.has_rot = mod, rot is not None
        ctx.save_for_backward(*held[:3], held[3] if rot is not None else dt, held[4], orot, ovel, opos, r0, v0, p0)
        return orot, ovel, opos

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rot, g_vel, g_pos):
        dtc, gy, ac, rk, r0b, orot, ovel, opos, r0, v0, p0 = ctx.saved_tensors
        mod = ctx.mod
        B, F = dtc.shape[:2]
        need = ctx.needs_input_grad          # (mod, dt, gyro, acc, rot, r0, v0, p0)
        c = lambda t: None if t is None else t.contiguous()
        g_rot, g_vel, g_pos = c(g_rot), c(g_vel), c(g_pos)
        o_gyro, o_acc = torch.empty_like(gy), torch.empty_like(ac)
        o_dt = torch.empty_like(dtc) if need[1] else None
        P = lambda t: t.data_ptr() if t is not None else None
        fn = _C.library().symbol("pplie_imu_integrate_bwd" + _sfx(dtc), _INTB_SIG)
        with _C._on_device(dtc.device):
            code = fn(P(dtc), P(gy), P(ac), P(rk) if ctx.has_rot else None, P(orot), P(ovel), P(r0b), mod._gravity_host(),
                      P(g_rot), P(g_vel), P(g_pos), P(o_gyro), P(o_acc), P(o_dt), B, F, _C.stream_ptr(dtc.device))
        _C.check(code, "pplie_imu_integrate_bwd")
        g_r0 = g_v0 = g_p0 = None
        if need[5] or need[6] or need[7]:
            # the initial state enters linearly (vel, pos) or as the first factor of the rotation products: its gradients are
            # sums over the steps of quantities already at hand
            zero3 = lambda: torch.zeros((B, F, 3), dtype=dtc.dtype, device=dtc.device)
            Gr = g_rot[..., :3] if g_rot is not None else zero3()
            Gv = g_vel if g_vel is not None else zero3()
            Gp = g_pos if g_pos is not None else zero3()
            Dt = torch.cumsum(dtc, dim=1)
            v0b, p0b = v0.reshape(-1, 1, 3), p0.reshape(-1, 1, 3)
            if need[7]:
                g_p0 = Gp.sum(1, keepdim=True).sum_to_size(p0b.shape).reshape(p0.shape)
            if need[6]:
                g_v0 = (Gv + Dt * Gp).sum(1, keepdim=True).sum_to_size(v0b.shape).reshape(v0.shape)
            if need[5]:
                t = Gr + torch.linalg.cross(ovel - v0b, Gv) + torch.linalg.cross(opos - p0b - v0b * Dt, Gp)
                if not ctx.has_rot:
                    gvec = mod.gravity.to(dtc.dtype).reshape(1, 1, 3).expand(B, F, 3)
                    t = t + torch.linalg.cross(gvec, _qrot(orot, o_acc))
                t = t.sum(1, keepdim=True)
                t = torch.cat([t, torch.zeros_like(t[..., :1])], dim=-1)
                g_r0 = t.sum_to_size(r0.reshape(-1, 1, 4).shape).reshape(r0.shape)
        return None, o_dt, o_gyro, o_acc, None, g_r0, g_v0, g_p0


class IMUPreintegrator(nn.Module):
    def __init__(self, pos=torch.zeros(3), rot=None, vel=torch.zeros(3), gravity=9.81007,
                 gyro_cov=(3.2e-3) ** 2, acc_cov=(8e-2) ** 2, prop_cov=True, reset=False):
        super().__init__()
        if not reset and not prop_cov:
            raise RuntimeError('"prop_cov" and "reset" cannot be False simultaneously.')
        self.reset, self.prop_cov = reset, prop_cov
        rot = identity_SO3() if rot is None else rot
        if isinstance(acc_cov, float):
            acc_cov = torch.tensor([[acc_cov, acc_cov, acc_cov]])
        if isinstance(gyro_cov, float):
            gyro_cov = torch.tensor([[gyro_cov, gyro_cov, gyro_cov]])
        self.register_buffer('gravity', torch.tensor([0, 0, gravity]), persistent=False)
        self.register_buffer('pos', self._check(pos).clone(), persistent=False)
        self.register_buffer('rot', self._check(rot).clone(), persistent=False)
        self.register_buffer('vel', self._check(vel).clone(), persistent=False)
        self.register_buffer('cov', torch.zeros(1, 9, 9), persistent=False)
        self.register_buffer('gyro_cov', gyro_cov, persistent=False)
        self.register_buffer('acc_cov', acc_cov, persistent=False)
        self.Rij = None      # rotation corresponding to the "zero-state" covariance
        self.fused_backward = True     # False: gradients through the composed LieTensor graph (tests compare the two routes)
        self.native_backward = True    # False: the fused node as a Python autograd.Function instead of the C++ one (same kernels)

    def _check(self, obj):
        if obj is not None:
            if len(obj.shape) == 2:
                obj = obj[None, ...]
            elif len(obj.shape) == 1:
                obj = obj[None, None, ...]
        return obj

    # ------------------------------------------------------------------------------------------
    def forward(self, dt, gyro, acc, rot: SO3 = None, gyro_cov=None, acc_cov=None, init_state=None):
        assert (0 < len(acc.shape) == len(dt.shape) == len(gyro.shape) <= 3)
        acc, gyro, dt, rot = self._check(acc), self._check(gyro), self._check(dt), self._check(rot)
        B = dt.shape[0]
        if init_state is None:
            init_state = {'pos': self.pos, 'rot': self.rot, 'vel': self.vel}
        if self.prop_cov:
            # (the fused kernel broadcasts a [1, 1, 3] covariance by stride; the composed route repeats it as the reference does)
            own_gc, own_ac = gyro_cov is None, acc_cov is None
            gyro_cov = self.gyro_cov if own_gc else gyro_cov
            acc_cov = self.acc_cov if own_ac else acc_cov
            if 'cov' not in init_state or init_state['cov'] is None:
                init_cov = self.cov                      # [1, 9, 9]: broadcast by the consumer (the fused route keeps a copy)
            else:
                init_cov = init_state['cov']
            Rij0 = init_state['Rij'] if 'Rij' in init_state else self.Rij

        fused = self._fused_ok(dt, gyro, acc, rot, init_state)
        if fused and self.prop_cov and torch.is_grad_enabled() and \
                any(t is not None and t.requires_grad for t in (gyro_cov, acc_cov, init_cov)):
            fused = False        # a learnt noise model: the covariance itself carries a gradient (composed route)
        if fused:
            # states: one kernel writing rot / vel / pos only; covariance: a second kernel that re-derives what it needs per
            # step (increment, gravity-free acceleration, Rij) from the raw inputs and the integrated rotations -- no
            # auxiliary [B, F, 4 + 4 + 3] streams between the two (csrc/scan.hip)
            predict, _ = self._fused_integrate(dt, gyro, acc, rot, init_state, None, aux=False)
            if self.prop_cov:
                # Rij_k = Rij0 * Dr_k = (Rij0 * r0^-1) * rot_k  (reference :283-286 with rot_k = r0 * Dr_k, :422)
                Cq = self._rij_offset(init_state['rot'], Rij0, B)
                last = SO3(Cq.unsqueeze(1)) * SO3(torch.Tensor.as_subclass(predict['rot'], torch.Tensor)[:, -1:, :])
                Rij = LieTensor(torch.Tensor.as_subclass(last, torch.Tensor), ltype=init_state['rot'].ltype) \
                    if isinstance(init_state['rot'], LieTensor) else last
                # (the covariance is a function of DETACHED states, reference :288-291; the kernel reads raw pointers)
                cov = {'cov': self._fused_cov2(dt, gyro, acc, rot, predict['rot'], Cq, init_cov, gyro_cov, acc_cov),
                       'Rij': Rij.detach()}
            else:
                cov = {'cov': None}
        else:
            inte_state = self.integrate(dt, gyro, acc, rot=rot, init_rot=init_state['rot'])
            predict = self.predict(init_state, inte_state)
            if self.prop_cov:
                gyro_cov = gyro_cov.repeat([B, 1, 1]) if own_gc else gyro_cov
                acc_cov = acc_cov.repeat([B, 1, 1]) if own_ac else acc_cov
                Rij = Rij0 * inte_state['Dr'] if Rij0 is not None else inte_state['Dr']
                cov_input_state = {'Rij': Rij.detach(), 'Rk': inte_state['w'].detach(),
                                   'Ha': vec2skew(inte_state['a'].detach()), 'dt': dt.detach()}
                cov = self.propagate_cov(cov_input=cov_input_state, init_cov=init_cov.expand(B, 9, 9), gyro_cov=gyro_cov,
                                         acc_cov=acc_cov)
            else:
                cov = {'cov': None}

        if not self.reset:
            self.pos = predict['pos'][..., -1:, :]
            self.rot = predict['rot'][..., -1:, :]
            self.vel = predict['vel'][..., -1:, :]
            self.cov = cov['cov']
            self.Rij = Rij[..., -1:, :]
        return {**predict, **cov}

    # ---- fused route ---------------------------------------------------------------------------
    def _fused_ok(self, dt, gyro, acc, rot, init_state):
        # (attribute reads on LieTensors are __torch_function__ round trips: the initial state is looked at through plain
        #  aliases, and the verdict for the module's own unchanged buffers is remembered)
        plain = torch.Tensor.as_subclass
        ts = [dt, gyro, acc, init_state['pos'], plain(init_state['rot'], torch.Tensor), init_state['vel']]
        if rot is not None:
            ts.append(plain(rot, torch.Tensor))
        if _C._test_backend is not None or not all(t.is_cuda for t in ts):
            return False
        if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
            # differentiable fused route (one node, pplie_imu_integrate_bwd): dt / gyro / acc / initial state may require a
            # gradient; a known-orientation input that does, or a functorch transform, takes the composed route
            if (rot is not None and rot.requires_grad) or torch._C._are_functorch_transforms_active() \
                    or not getattr(self, 'fused_backward', True):
                return False
        return dt.dtype in (torch.float32, torch.float64) and all(t.dtype == dt.dtype for t in ts)

    def _rij_offset(self, rot0, Rij0, B):
        """Rij0 * r0^-1 as [B, 4] (kept while both tensors stay the same objects at the same versions)"""
        raw = torch.Tensor.as_subclass(rot0, torch.Tensor)
        rawj = torch.Tensor.as_subclass(Rij0, torch.Tensor) if Rij0 is not None else None
        key = (rot0, raw._version, Rij0, rawj._version if rawj is not None else -1, B)
        hit = self.__dict__.get('_cq_cache')
        if hit is not None and hit[0][0] is key[0] and hit[0][2] is key[2] and hit[0][1] == key[1] and hit[0][3] == key[3] \
                and hit[0][4] == B:
            return hit[1]
        r0 = rot0 if isinstance(rot0, LieTensor) else SO3(rot0)
        Cq = r0.Inv() if Rij0 is None else Rij0 * r0.Inv()
        Cq = torch.Tensor.as_subclass(Cq, torch.Tensor).expand(B, 1, 4).reshape(B, 4).contiguous()
        self.__dict__['_cq_cache'] = (key, Cq)
        return Cq

    def _bcast(self, slot, t, B, w):
        """``t`` (one row, or B rows) as a contiguous [B, w] buffer.  Broadcast copies are kept per slot for as long as the
        source tensor object and its version stay the same: a module called in a loop with the same initial state launched
        a dozen 5 us expand-copies per forward, a third of the time of the integration itself."""
        raw = torch.Tensor.as_subclass(t, torch.Tensor)
        if raw.numel() == B * w and raw.is_contiguous():
            return raw.view(B, w)
        cache = self.__dict__.setdefault('_bcast_cache', {})
        hit = cache.get(slot)
        if hit is not None and hit[0] is t and hit[1] == raw._version and hit[2].shape[0] == B and hit[2].dtype == raw.dtype \
                and hit[2].device == raw.device:
            return hit[2]
        out = raw.reshape(-1, w).expand(B, w).contiguous() if raw.numel() == w else raw.reshape(B, w).contiguous()
        cache[slot] = (t, raw._version, out)
        return out

    def _isotropic(self, cv):
        """is this (single-row) covariance one value three times?  Read back once per tensor object and version."""
        hit = self.__dict__.get('_iso_cache')
        if hit is None or hit[0] is not cv or hit[1] != cv._version:
            v = cv.reshape(-1).tolist()
            hit = self.__dict__['_iso_cache'] = (cv, cv._version, len(v) == 3 and v[0] == v[1] == v[2])
        return hit[2]

    def _gravity_host(self):
        """the gravity vector as a C array (read back once per value: the buffer lives on the device)"""
        gt = self.gravity
        hit = self.__dict__.get('_g_host')
        if hit is None or hit[0] is not gt or hit[1] != gt._version:
            hit = self.__dict__['_g_host'] = (gt, gt._version, (ctypes.c_double * 3)(*[float(x) for x in gt.tolist()]))
        return hit[2]

    def _fused_cov2(self, dt, gyro, acc, rot, rot_out, Cq, init_cov, gyro_cov, acc_cov):
        """pplie_imu_cov2: covariance from the raw inputs + the integrated rotations (segment-walk kernel)."""
        B, F = dt.shape[:2]
        cov = torch.empty((B, 9, 9), dtype=dt.dtype, device=dt.device)

        def strided(cv):            # [B or 1, F or 1, 3] -> element strides over (b, f)
            cv = cv.to(dt.dtype)
            cv = cv if cv.dim() == 3 else cv.reshape(-1, 1, 3)
            cv = cv.contiguous()
            return cv, (cv.stride(0) if cv.shape[0] > 1 else 0), (cv.stride(1) if cv.shape[1] > 1 else 0)
        gc, gsb, gsf = strided(gyro_cov)
        ac, asb, asf = strided(acc_cov)
        if asb == 0 and asf == 0 and self._isotropic(acc_cov):
            asf = -1          # one isotropic variance for every sequence and step (the default noise model): scalar sums in the kernel
        ic = self._bcast('cov0', init_cov if init_cov.dtype == dt.dtype else init_cov.to(dt.dtype), B, 81)
        ro = torch.Tensor.as_subclass(rot_out, torch.Tensor).contiguous()
        rw = torch.Tensor.as_subclass(rot, torch.Tensor).expand(B, F, 4).contiguous() if rot is not None else ro
        g = self._gravity_host()
        fn = _C.library().symbol("pplie_imu_cov2" + _sfx(dt), _COV2_SIG)
        dtc, gyc, acc_c = dt.contiguous(), gyro.contiguous(), acc.contiguous()     # (held: a temporary's block could be reused
        with _C._on_device(dt.device):                                           #  by the next allocation before the launch)
            code = fn(dtc.data_ptr(), gyc.data_ptr(), acc_c.data_ptr(), ro.data_ptr(), rw.data_ptr(),
                      Cq.data_ptr(), ic.data_ptr(), gc.data_ptr(), gsb, gsf, ac.data_ptr(), asb, asf, g, cov.data_ptr(), B, F,
                      _C.stream_ptr(dt.device))
        _C.check(code, "pplie_imu_cov2")
        return cov

    def _launch_integrate(self, dt, gyro, acc, rot, r0, v0, p0, Rij0, aux):
        """pplie_imu_integrate on plain tensors; returns (rot, vel, pos, contiguous inputs it read [dt, gyro, acc, rot, r0])"""
        B, F = dt.shape[:2]
        dev, dty = dt.device, dt.dtype
        c = lambda t: t.contiguous()
        r0 = self._bcast('r0', r0, B, 4)
        v0 = self._bcast('v0', v0, B, 3)
        p0 = self._bcast('p0', p0, B, 3)
        dtc, gy, ac = c(dt), c(gyro), c(acc)
        rk = c(torch.Tensor.as_subclass(rot, torch.Tensor).expand(B, F, 4)) if rot is not None else None
        q0 = c(torch.Tensor.as_subclass(Rij0, torch.Tensor).expand(B, 1, 4).reshape(B, 4)) if Rij0 is not None else None
        orot = torch.empty((B, F, 4), dtype=dty, device=dev)
        ovel = torch.empty((B, F, 3), dtype=dty, device=dev)
        opos = torch.empty((B, F, 3), dtype=dty, device=dev)
        aux = aux or {}
        g = self._gravity_host()
        P = lambda t: t.data_ptr() if t is not None else None
        fn = _C.library().symbol("pplie_imu_integrate" + _sfx(dt), _INT_SIG)
        with _C._on_device(dev):
            code = fn(P(dtc), P(gy), P(ac), P(rk), P(r0), P(v0), P(p0), P(q0), g, P(orot), P(ovel), P(opos),
                      P(aux.get('Rk')), P(aux.get('Rij')), P(aux.get('a')), B, F, _C.stream_ptr(dev))
        _C.check(code, "pplie_imu_integrate")
        return orot, ovel, opos, (dtc, gy, ac, rk, r0)

    def _fused_integrate(self, dt, gyro, acc, rot, init_state, Rij0, aux=None):
        B, F = dt.shape[:2]
        dev, dty = dt.device, dt.dtype
        want_aux = self.prop_cov if aux is None else aux
        aux = {}
        if want_aux:
            aux = {'Rk': torch.empty((B, F, 4), dtype=dty, device=dev), 'Rij': torch.empty((B, F, 4), dtype=dty, device=dev),
                   'a': torch.empty((B, F, 3), dtype=dty, device=dev)}
        r_in = init_state['rot']
        plain = torch.Tensor.as_subclass
        ins = (dt, gyro, acc, plain(r_in, torch.Tensor), init_state['vel'], init_state['pos'])
        if torch.is_grad_enabled() and any(t.requires_grad for t in ins):
            # training through the pre-integrator: one autograd node, backward = pplie_imu_integrate_bwd
            assert not want_aux
            rk = None if rot is None else plain(rot, torch.Tensor).detach()
            nat = _native_node(self, dt, gyro, acc, rk, ins[3], ins[4], ins[5])
            orot, ovel, opos = nat if nat is not None else _ImuIntegrate.apply(self, dt, gyro, acc, rk, ins[3], ins[4], ins[5])
        else:
            # (r_in itself, not the fresh alias in `ins`: _bcast's cache recognises the caller's tensor OBJECT)
            orot, ovel, opos, _ = self._launch_integrate(dt, gyro, acc, rot, r_in, ins[4], ins[5], Rij0, aux)
        rot_out = _lt._wrap(orot, r_in.ltype if isinstance(r_in, LieTensor) else _lt.SO3_type)
        return {'rot': rot_out, 'vel': ovel, 'pos': opos}, aux

    def _fused_cov(self, dt, aux, init_cov, gyro_cov, acc_cov):
        B, F = dt.shape[:2]
        cov = torch.empty((B, 9, 9), dtype=dt.dtype, device=dt.device)

        def strided(cv):            # [B or 1, F or 1, 3] -> element strides over (b, f)
            cv = cv.to(dt.dtype)
            cv = cv if cv.dim() == 3 else cv.reshape(-1, 1, 3)
            cv = cv.contiguous()
            sb = cv.stride(0) if cv.shape[0] > 1 else 0
            sf = cv.stride(1) if cv.shape[1] > 1 else 0
            return cv, sb, sf
        gc, gsb, gsf = strided(gyro_cov)
        ac, asb, asf = strided(acc_cov)
        ic = init_cov.to(dt.dtype).expand(B, 9, 9).contiguous()
        fn = _C.library().symbol("pplie_imu_cov" + _sfx(dt), _COV_SIG)
        with _C._on_device(dt.device):
            code = fn(dt.contiguous().data_ptr(), aux['Rk'].data_ptr(), aux['Rij'].data_ptr(), aux['a'].data_ptr(),
                      ic.data_ptr(), gc.data_ptr(), gsb, gsf, ac.data_ptr(), asb, asf, cov.data_ptr(), B, F,
                      _C.stream_ptr(dt.device))
        _C.check(code, "pplie_imu_cov")
        return cov

    # ---- composed (differentiable) route: reference :314-465 --------------------------------------
    def integrate(self, dt, gyro, acc, rot: SO3 = None, init_rot: SO3 = None):
        B, F = dt.shape[:2]
        dr = so3(gyro * dt).Exp()
        w = torch.cat([identity_SO3(B, 1, dtype=dt.dtype, device=dt.device), dr], dim=1)
        incre_r = cumprod(w, dim=1, left=False)
        if isinstance(rot, LieTensor):
            a = acc - rot.Inv() @ self.gravity
        else:
            if init_rot is None:
                init_rot = identity_SO3(B, 1, dtype=dt.dtype, device=dt.device)
            a = acc - (init_rot * incre_r)[:, 1:, :].Inv() @ self.gravity
        Ra = incre_r[:, :F, :] @ a
        zeros = torch.zeros(B, 1, 3, dtype=dt.dtype, device=dt.device)
        incre_v = torch.cumsum(torch.cat([zeros, Ra * dt], dim=1), dim=1)
        incre_p = torch.cumsum(torch.cat([zeros, incre_v[:, :F, :] * dt + Ra * 0.5 * dt ** 2], dim=1), dim=1)
        incre_t = torch.cat([torch.zeros(B, 1, 1, dtype=dt.dtype, device=dt.device), torch.cumsum(dt, dim=1)], dim=1)
        return {'a': a, 'Dp': incre_p[:, 1:, :], 'Dv': incre_v[..., 1:, :], 'Dr': incre_r[:, 1:, :],
                'Dt': incre_t[..., 1:, :], 'w': w[:, 1:, :]}

    @classmethod
    def predict(cls, init_state, integrate):
        return {
            'rot': init_state['rot'] * integrate['Dr'],
            'vel': init_state['vel'] + init_state['rot'] * integrate['Dv'],
            'pos': init_state['pos'] + init_state['rot'] * integrate['Dp'] + init_state['vel'] * integrate['Dt'],
        }

    @classmethod
    def propagate_cov(cls, cov_input, init_cov, gyro_cov, acc_cov):
        """cov = sum_k P_k Bc_k P_k^T with P_k = A_k ... A_{F-1}, Bc_0 = init_cov (reference :428-465),
        evaluated by one backward pass P <- A_k P, cov += P Bc_k P^T (no [B,F+1,9,9] scan)."""
        dt = cov_input['dt']
        B, F = dt.shape[:2]
        dev, dty = dt.device, dt.dtype
        Cg, Ca = torch.diag_embed(gyro_cov).to(dty), torch.diag_embed(acc_cov).to(dty)
        Rk, Rij, Ha = cov_input['Rk'].matrix(), cov_input['Rij'].matrix(), cov_input['Ha']
        h = dt.unsqueeze(-1)                                       # [B,F,1,1]
        A = torch.eye(9, device=dev, dtype=dty).repeat([B, F, 1, 1])
        A[..., 0:3, 0:3] = Rk.mT
        A[..., 3:6, 0:3] = -(Rij @ Ha) * h
        A[..., 6:9, 0:3] = -0.5 * (Rij @ Ha) * h ** 2
        A[..., 6:9, 3:6] = torch.eye(3, device=dev, dtype=dty) * h
        Bg = torch.zeros(B, F, 9, 3, device=dev, dtype=dty)
        Ba = torch.zeros(B, F, 9, 3, device=dev, dtype=dty)
        Bg[..., 0:3, 0:3] = cov_input['Rk'].Jr() * h
        Ba[..., 3:6, 0:3] = Rij * h
        Ba[..., 6:9, 0:3] = 0.5 * Rij * h ** 2
        Bc = (Bg @ Cg @ Bg.mT + Ba @ Ca @ Ba.mT) / h              # Bc[:, k] is the reference's B_cov[k+1]
        P = torch.eye(9, device=dev, dtype=dty).repeat([B, 1, 1])
        cov = torch.zeros(B, 9, 9, device=dev, dtype=dty)
        for k in range(F, 0, -1):
            if k < F:
                P = A[:, k] @ P
            cov = cov + P @ Bc[:, k - 1] @ P.mT
        P = A[:, 0] @ P
        cov = cov + P @ init_cov @ P.mT
        return {'cov': cov, 'Rij': cov_input['Rij'][..., -1:, :]}
