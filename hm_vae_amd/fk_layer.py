"""ForwardKinematicsLayer with the reference's surface (fk_layer.py:11-93), as one fused HIP kernel."""
import json

import numpy as np
import torch

from . import functional as Fn
from . import topology


class ForwardKinematicsLayer(torch.nn.Module):
    """forward(rotations, positions=None): rotations (bs, J, 6) or (bs, J, 3, 3) -> joint xyz (bs, J, 3).

    The reference builds (bs,J,4,4) homogeneous matrices and walks the chain with 23 gather/bmm/copy_ steps; here the
    SMPL-24 tree is compile-time and a frame's walk is split by subtree over the four waves of a workgroup (32-frame LDS
    tiles, globals in registers); any other `parents` (joint 0 the root, parents[j] < j, at most 64 joints -- what the
    reference's loop needs too) runs table-driven kernels with the same results (csrc/fk_kernels.hip).
    """

    def __init__(self, device=None, parents=None, positions=None):
        super().__init__()
        if device is None:
            device = torch.device("cuda")
        if parents is None and positions is None:
            # the reference hard-codes absolute paths on the author's machine here; use the packaged assets
            parents = json.load(open(topology.resolve_asset("./utils/data/joint24_parents.json"), "r"))
            positions = np.load(topology.resolve_asset("./utils/data/skeleton_offsets.npy"))
        self.parent_list = [int(p) for p in parents]
        self.parents = torch.tensor(self.parent_list, dtype=torch.long, device=device)
        self.positions = torch.from_numpy(np.asarray(positions)).float()[None, :, :].to(device)     # 1 x J x 3
        self.device = device
        self.b_idxs = None

    def forward(self, rotations, positions=None):
        if rotations.shape[-1] == 3 and rotations.shape[-2] == 3:
            rot = rotations
        elif rotations.shape[-1] == 6:
            rot = rotations
        else:
            raise ValueError("rotations must be (bs,J,6) or (bs,J,3,3)")
        if self.positions.device != rot.device:
            self.positions = self.positions.to(rot.device)
            self.parents = self.parents.to(rot.device)
        off = self.positions[0] if positions is None else positions.to(rot.device)
        _, pos = Fn.FkFn.apply(rot, off, self.parent_list, False)
        return pos
