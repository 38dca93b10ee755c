"""g2o pose-graph files <-> tensors (the wire format either side of the pose-graph path; SURVEY.md
section 8f rank 2; reference loader: examples/module/pgo/pgo_dataset.py:8-60).

    VERTEX_SE3:QUAT id x y z qx qy qz qw
    EDGE_SE3:QUAT   i j x y z qx qy qz qw  i11 i12 ... i66      (21 upper-triangular information entries)

``read_g2o`` parses the whole file with numpy in one pass per record type (the reference builds one torch tensor
per line), keeps file order, and returns the same five tensors the reference's ``G2OPGO`` holds.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from ..lietensor import SE3

_IU = np.triu_indices(6)


def _info_to_mat(tri):
    """[E, 21] upper-triangular rows -> symmetric [E, 6, 6] (info2mat of the reference, vectorised)"""
    mat = np.zeros((tri.shape[0], 6, 6), dtype=np.float64)
    mat[:, _IU[0], _IU[1]] = tri
    mat[:, _IU[1], _IU[0]] = tri
    return mat


def read_g2o(path, device='cpu', dtype=None):
    """dict(ids [N] int64, nodes SE3 [N,7], edges [E,2] int64, poses SE3 [E,7], infos [E,6,6]).
    Values are parsed in float64 and cast to ``dtype`` (default: ``torch.get_default_dtype()``) like the
    reference (pgo_dataset.py:31, 46-50); other record types are ignored."""
    dtype = torch.get_default_dtype() if dtype is None else dtype
    vert, edge = [], []
    with open(path) as f:
        for line in f:
            if line.startswith('VERTEX_SE3:QUAT'):
                vert.append(line[len('VERTEX_SE3:QUAT'):])
            elif line.startswith('EDGE_SE3:QUAT'):
                edge.append(line[len('EDGE_SE3:QUAT'):])
    v = np.array(' '.join(vert).split(), dtype=np.float64).reshape(len(vert), -1) if vert else np.zeros((0, 8))
    e = np.array(' '.join(edge).split(), dtype=np.float64).reshape(len(edge), -1) if edge else np.zeros((0, 30))
    if v.shape[1] != 8:
        raise ValueError(f"{path}: VERTEX_SE3:QUAT records need 8 fields, got {v.shape[1]}")
    if e.shape[1] != 30:
        raise ValueError(f"{path}: EDGE_SE3:QUAT records need 30 fields, got {e.shape[1]}")
    ids = torch.from_numpy(v[:, 0].astype(np.int64))
    edges = torch.from_numpy(e[:, :2].astype(np.int64)).to(device)
    nodes = SE3(torch.from_numpy(v[:, 1:8].copy()).to(dtype).to(device))
    poses = SE3(torch.from_numpy(e[:, 2:9].copy()).to(dtype).to(device))
    infos = torch.from_numpy(_info_to_mat(e[:, 9:])).to(dtype).to(device)
    assert ids.size(0) == nodes.size(0) and edges.size(0) == poses.size(0) == infos.size(0)
    return {"ids": ids, "nodes": nodes, "edges": edges, "poses": poses, "infos": infos}


def write_g2o(path, nodes, edges, poses, infos=None, ids=None):
    """Inverse of :func:`read_g2o` (17 significant digits: a float64 round trip is exact)."""
    nodes = np.asarray(torch.as_tensor(nodes).detach().cpu().double())
    poses = np.asarray(torch.as_tensor(poses).detach().cpu().double())
    edges = np.asarray(torch.as_tensor(edges).cpu())
    E = edges.shape[0]
    infos = np.tile(np.eye(6), (E, 1, 1)) if infos is None else np.asarray(torch.as_tensor(infos).detach().cpu().double())
    ids = np.arange(nodes.shape[0]) if ids is None else np.asarray(torch.as_tensor(ids).cpu())
    fmt = lambda row: ' '.join(repr(float(x)) for x in row)
    with open(path, 'w') as f:
        for i, n in zip(ids, nodes):
            f.write(f"VERTEX_SE3:QUAT {int(i)} {fmt(n)}\n")
        for (i, j), p, m in zip(edges, poses, infos):
            f.write(f"EDGE_SE3:QUAT {int(i)} {int(j)} {fmt(p)} {fmt(m[_IU])}\n")


class G2OPGO(torch.utils.data.Dataset):
    """The reference example's dataset class (pgo_dataset.py:8-60) on top of :func:`read_g2o`; there is no
    download step (no network): ``root/dataname`` must exist."""

    def __init__(self, root, dataname, device='cpu', download=False):
        super().__init__()
        if download:
            raise RuntimeError("G2OPGO: downloading is not supported here; place the .g2o file under `root`")
        self.dtype = torch.get_default_dtype()
        d = read_g2o(os.path.join(root, dataname), device=device, dtype=self.dtype)
        self.ids, self.nodes, self.edges, self.poses, self.infos = d["ids"], d["nodes"], d["edges"], d["poses"], d["infos"]

    def init_value(self):
        return self.nodes.clone()

    def __getitem__(self, i):
        return self.edges[i], self.poses[i], self.infos[i]

    def __len__(self):
        return self.edges.size(0)
