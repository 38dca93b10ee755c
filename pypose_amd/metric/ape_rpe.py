"""Absolute / relative pose error of an estimated trajectory against a reference one (host-side mirror of
pypose/metric/ape_rpe.py).

Same functions, arguments, defaults, error messages and result dictionaries as the reference.  The arithmetic is a
composition of hot-path kernels -- ``Inv``, ``Mul`` (:244-248, :686-687), ``matrix``, ``mat2SO3``, ``Log`` -- around
host-side bookkeeping (time-stamp association, pair selection, statistics).
"""
import warnings

import torch

from ..function.geometry import svdstf
from ..lietensor.convert import mat2SO3
from ..lietensor.lietensor import SE3_type, Sim3_type
from ..lietensor.utils import SE3, Sim3, identity_Sim3

_STATISTICS = ['All', 'Max', 'Min', 'Mean', 'Median', 'RMSE', 'SSE', 'STD']


class StampedSE3(object):
    """A trajectory: ascending time stamps (float64) and one SE3 pose per stamp (ape_rpe.py:7-104)."""

    def __init__(self, timestamps=None, poses_SE3=None, dtype=torch.float64):
        assert (poses_SE3 is not None), {"The pose must be not None"}
        assert (poses_SE3.numel() != 0), {"The pose must be not empty"}
        assert len(poses_SE3.lshape) == 1, {"Only one trajectory estimation is support,\
                The shape of the trajectory must be 2"}
        self.poses = poses_SE3.to(dtype)
        count, device = poses_SE3.lshape[0], poses_SE3.device
        if timestamps is None:
            self.timestamps = torch.arange(count, dtype=torch.float64, device=device)
        else:
            self.timestamps = timestamps.type(torch.float64).to(device)
        assert len(self.timestamps.shape) == 1, {"The timestamp should be one array"}
        assert self.timestamps.shape[0] == count, {"timestamps and poses must have same length"}
        assert torch.all(torch.sort(self.timestamps)[0] == self.timestamps), {"timestamps must be accending"}

    def __getitem__(self, index):
        return StampedSE3(self.timestamps[index], self.poses[index], self.poses.dtype)

    def reduce_to_ids(self, ids):
        ids = ids.long().tolist() if isinstance(ids, torch.Tensor) else ids
        self.timestamps, self.poses = self.timestamps[ids], self.poses[ids]

    def align(self, trans):
        """Left-multiply every pose by ``trans`` (SE3, or Sim3: the poses are lifted with unit scale and the
        scale of the product is dropped again)."""
        if trans.ltype is SE3_type:
            self.poses = trans @ self.poses
        elif trans.ltype is Sim3_type:
            data = self.poses.tensor()
            lifted = Sim3(torch.cat((data, torch.ones_like(data[..., :1])), dim=-1))
            self.poses = SE3((trans @ lifted).tensor()[..., :7])

    # -- the trajectory's view of its poses (API of ape_rpe.py:63-104): read accessors forward to the LieTensor, the three
    #    "move" calls replace it in place.  One forwarding rule each instead of a method per name.
    _FORWARDED = ("translation", "rotation")                  # poses.<name>()
    _PROPERTIES = {"num_poses": lambda p: p.shape[0], "first_pose": lambda p: p[0], "dtype": lambda p: p.dtype,
                   "device": lambda p: p.device}

    def __getattr__(self, name):
        if name in StampedSE3._FORWARDED:
            return getattr(self.__dict__["poses"], name)
        prop = StampedSE3._PROPERTIES.get(name)
        if prop is not None:
            return prop(self.__dict__["poses"])
        raise AttributeError(name)

    def _replace(self, moved):
        self.poses = moved

    def type(self, dtype=torch.float64):
        self._replace(self.poses.to(dtype))

    def cuda(self):
        self._replace(self.poses.cuda())

    def cpu(self):
        self._replace(self.poses.cpu())

    @property
    def accumulated_distances(self):
        """Path length travelled up to each pose (0 at the first)."""
        t = self.translation()
        hops = torch.linalg.norm(t[:-1] - t[1:], dim=-1, dtype=t.dtype)
        return torch.cat((torch.zeros(1, dtype=t.dtype, device=hops.device), torch.cumsum(hops, dim=0)))


def matching_time_indices(stamps_1, stamps_2, max_diff=0.01, offset_2=0.0):
    """For every stamp of the first list the nearest stamp of the second (shifted by ``offset_2``); pairs further
    apart than ``max_diff`` are dropped.  Returns two index lists (ape_rpe.py:107-139)."""
    gaps = (stamps_1[..., None] - (stamps_2 + offset_2)[None]).abs()
    nearest_gap, nearest = gaps.min(dim=-1)
    keep = nearest_gap < max_diff
    first = torch.arange(len(stamps_1), device=stamps_1.device)
    return first[keep].tolist(), nearest[keep].tolist()


def associate_traj(rtraj, etraj, max_diff=0.01, offset_2=0.0, threshold=0.3):
    """Pair the poses of the two trajectories by time stamp; the shorter trajectory drives the search
    (ape_rpe.py:142-201).  Returns the matched (reference, estimate) sub-trajectories."""
    est_longer = len(etraj.timestamps) > len(rtraj.timestamps)
    longer, shorter = (etraj, rtraj) if est_longer else (rtraj, etraj)
    i_short, i_long = matching_time_indices(shorter.timestamps, longer.timestamps, max_diff,
                                            offset_2 if est_longer else -offset_2)
    assert len(i_short) == len(i_long), {r"matching_time_indices returned unequal number of indices"}
    matches = len(i_long)
    assert matches != 0, \
        {f"found no matching timestamps between estimation and reference with max time "
         f"diff {{max_diff}} (s) and time offset {{offset_2}} (s)"}
    shorter_m, longer_m = shorter[i_short], longer[i_long]
    if matches < threshold * len(shorter.timestamps):
        warnings.warn("Alert !!!!!!!!!!!!!!!!!!!!!!! \
                       The estimated trajectory has not enough \
                       timestamps within the GT timestamps. \
                       May be not be enough for aligned and not accurate results.", category=Warning, stacklevel=2)
    return (shorter_m, longer_m) if est_longer else (longer_m, shorter_m)


def compute_error(rtraj, etraj, output: str = 'translation', mtype: str = 'ape', otype: str = 'All'):
    """Per-pose error of the chosen kind and its statistics (ape_rpe.py:204-288).

    ape: translation error is ``|t_est - t_ref|``; every other kind looks at ``E = est^-1 ref`` (:244-246).
    rpe: ``E = ref^-1 est`` (:248), translation error is the norm of its translation.
    'rotation' / 'pose': Frobenius distance of the 3x3 / 4x4 matrix of E from the identity; 'radian' / 'degree':
    rotation angle of E."""
    if mtype == 'ape' and output == 'translation':
        error = torch.linalg.norm(etraj.translation() - rtraj.translation(), dim=-1)
    else:
        if mtype == 'ape':
            E = (etraj.poses.Inv() @ rtraj.poses).matrix()
        elif mtype == 'rpe':
            E = (rtraj.poses.Inv() @ etraj.poses).matrix()
        if output == 'translation':
            error = E[..., :3, 3].norm(dim=-1)
        elif output == 'rotation':
            R = E[:, :3, :3]
            error = torch.linalg.norm(R - torch.eye(3, device=E.device, dtype=E.dtype).expand_as(R), dim=(-2, -1))
        elif output == 'pose':
            error = torch.linalg.norm(E - torch.eye(4, device=E.device, dtype=E.dtype).expand_as(E), dim=(-2, -1))
        elif output in ('radian', 'degree'):
            error = mat2SO3(E[:, :3, :3], check=False).Log().norm(dim=-1)
            if output == 'degree':
                error = error.rad2deg()
        else:
            raise ValueError(f"Unknown output type: {output}")
    if otype not in _STATISTICS:
        raise ValueError(f"Unknown output metric type, select one in {_STATISTICS}")
    size = error.abs()
    results = {'Max': torch.max(size), 'Min': torch.min(size), 'Mean': torch.mean(size), 'Median': torch.median(size),
               'RMSE': torch.sqrt(torch.mean(torch.pow(error, 2))), 'SSE': torch.sum(torch.pow(error, 2)),
               'STD': torch.std(size)}
    return results if otype == 'All' else results[otype]


def pairs_by_frames(traj, delta, all=False):
    """Index pairs ``delta`` frames apart: every start frame when ``all``, else consecutive multiples of delta."""
    count, delta = traj.num_poses, int(delta)
    assert delta >= 1, "delta must >= 1"
    if all:
        start = torch.arange(count, device=traj.device, dtype=torch.long)
        inside = start + delta < count
        return start[inside].tolist(), (start + delta)[inside].tolist()
    ids = torch.arange(0, count, delta, device=traj.device, dtype=torch.long)
    return ids[:-1].tolist(), ids[1:].tolist()


def pairs_by_dist(traj, delta, tol=0.0, all=False):
    """Index pairs whose poses lie ``delta`` of path length apart (ape_rpe.py:322-365): with ``all`` the best
    partner of every start pose if it is within ``tol`` of delta, else consecutive poses where the walked distance
    since the last chosen pose first reaches delta."""
    if all:
        begin, end = [], []
        walked = traj.accumulated_distances
        for i in range(walked.size(0) - 1):
            ahead = walked[i + 1:] - walked[i]
            j = torch.argmin(torch.abs(ahead - delta)).item()
            if torch.abs(ahead[j] - delta) > tol:
                continue
            begin.append(i)
            end.append(j + i + 1)
        return begin, end
    chosen, path = [], 0.0
    positions = traj.translation()
    last = positions[0]
    for i, here in enumerate(positions):
        path += float(torch.norm(here - last))
        last = here
        if path >= delta:
            chosen.append(i)
            path = 0.0
    return chosen[:-1], chosen[1:]


def pair_id(traj, delta=1.0, associate: str = 'frame', rtol=0.1, all=False):
    if associate == 'frame':
        id_pairs = pairs_by_frames(traj, int(delta), all)
    elif associate == 'distance':
        id_pairs = pairs_by_dist(traj, delta, delta * rtol, all)
    else:
        raise ValueError(f"unsupported delta unit: {associate}")
    if len(id_pairs) == 0:
        raise ValueError(
            f"delta = {delta} ({associate}) produced an empty index list - try lower values or a less strict tolerance")
    return id_pairs


def _register(rtraj, etraj, align, scale, nposes, origin, points):
    """The transform applied to the estimate before comparing (ape_rpe.py:527-536, :674-683): a least-squares
    similarity / rigid fit of the positions (``align`` / ``scale``), or the one that makes the first poses coincide
    (``origin``), else the identity.  ``points`` reproduces how the reference trims the positions before the fit (its
    slice acts on the coordinate axis)."""
    trans = identity_Sim3(1, dtype=etraj.dtype, device=etraj.device)
    if align or scale:
        nposes = etraj.num_poses if nposes == -1 else nposes
        trans = svdstf(points(etraj.translation(), nposes), points(rtraj.translation(), nposes), scale)
    elif origin:
        trans[..., :7] = (rtraj.first_pose @ etraj.first_pose.Inv()).tensor()
    return trans


def ape(rstamp, rpose, estamp, epose, etype: str = "translation", diff: float = 0.01, offset: float = 0.0,
        align: bool = False, scale: bool = False, nposes: int = -1, origin: bool = False, thresh: float = 0.3,
        otype: str = 'All'):
    """Absolute pose error between a reference and an estimated trajectory (ape_rpe.py:407-536): associate by
    time stamp (``diff``, ``offset``, ``thresh``), optionally register the estimate (``align`` / ``scale`` /
    ``origin``), then compare pose by pose.  ``etype``: 'translation' | 'rotation' | 'pose' | 'radian' | 'degree';
    ``otype``: 'All' (dictionary) or one of Max / Min / Mean / Median / RMSE / SSE / STD."""
    rtraj, etraj = associate_traj(StampedSE3(rstamp, rpose), StampedSE3(estamp, epose), diff, offset, thresh)
    etraj.align(_register(rtraj, etraj, align, scale, nposes, origin, lambda t, n: t[..., :n]))
    return compute_error(rtraj, etraj, etype, mtype='ape', otype=otype)


def rpe(rstamp, rpose, estamp, epose, etype: str = "translation", diff: float = 0.01, offset: float = 0.0,
        align: bool = False, scale: bool = False, nposes: int = -1, origin: bool = False, associate: str = 'frame',
        delta: float = 1.0, rtol: float = 0.1, all: bool = False, thresh: float = 0.3, rpair: bool = False,
        otype: str = 'All'):
    """Relative pose error (ape_rpe.py:539-691): like :func:`ape`, but the motion between pose pairs ``delta``
    apart (``associate``: 'frame' | 'distance', tolerance ``rtol``, ``all`` pairs or consecutive ones, pairs picked
    on the reference trajectory when ``rpair``) is compared instead of the poses themselves."""
    rtraj, etraj = associate_traj(StampedSE3(rstamp, rpose), StampedSE3(estamp, epose), diff, offset, thresh)
    etraj.align(_register(rtraj, etraj, align, scale, nposes, origin, lambda t, n: t[:, :n]))
    src, dst = pair_id(rtraj if rpair else etraj, delta, associate, rtol, all)
    r_from, e_from = rtraj[src], etraj[src]
    r_motion = StampedSE3(r_from.timestamps, r_from.poses.Inv() @ rtraj[dst].poses)
    e_motion = StampedSE3(e_from.timestamps, e_from.poses.Inv() @ etraj[dst].poses)
    return compute_error(r_motion, e_motion, etype, mtype='rpe', otype=otype)
