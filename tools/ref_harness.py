"""Import harness for the *unmodified* reference (`/root/reference`) on CPU.

Build-container only: this file is used by `tools/gen_golden.py` to produce the
fixtures under `tests/golden/`.  Nothing on the GPU box imports it (the reference
does not exist there).  It follows the recipe in SURVEY.md §8(c):

  1. absent, hot-path-irrelevant packages are replaced by inert mocks,
  2. cwd is moved to the reference root (asset paths there are cwd-relative),
  3. `.cuda()` becomes the identity (no GPU in the build container),
  4. `ForwardKinematicsLayer()` is forced onto CPU with the reference's own assets,
  5. `torch.randn_like` can be overridden to inject / record eps.
"""
import os
import sys
import json
import types
from unittest import mock

import numpy as np
import torch

REF_ROOT = os.environ.get("HMVAE_REFERENCE", "/root/reference")

_ABSENT = ["torchgeometry", "cv2", "torchvision", "torchvision.utils", "tensorboardX", "imageio",
           "torch.utils.tensorboard", "lib", "lib.utils", "lib.utils.eval_utils",
           "matplotlib", "matplotlib.pyplot", "matplotlib.animation", "mpl_toolkits",
           "mpl_toolkits.mplot3d", "mpl_toolkits.mplot3d.axes3d", "joblib"]

_done = False


def install():
    """Make `import seq_two_hier_sa_vae` etc. work from the reference tree. Idempotent."""
    global _done
    if _done:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    for name in _ABSENT:
        if name not in sys.modules:
            try:
                if name.startswith("matplotlib") or name.startswith("mpl_toolkits"):
                    __import__(name)
                    continue
            except Exception:
                pass
            sys.modules[name] = mock.MagicMock(name=name)
    os.chdir(REF_ROOT)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    import fk_layer as ref_fk
    parents = json.load(open(os.path.join(REF_ROOT, "utils/data/joint24_parents.json")))
    offsets = np.load(os.path.join(REF_ROOT, "utils/data/skeleton_offsets.npy"))
    base = ref_fk.ForwardKinematicsLayer

    class CpuFK(base):
        def __init__(self, *a, **k):
            base.__init__(self, device=torch.device("cpu"), parents=parents, positions=offsets)

    ref_fk.ForwardKinematicsLayer = CpuFK
    import seq_two_hier_sa_vae as ref_model
    ref_model.ForwardKinematicsLayer = CpuFK
    import trajectory_pred_model as ref_traj
    ref_traj.ForwardKinematicsLayer = CpuFK
    _done = True


class EpsInjector:
    """Context manager: `torch.randn_like` returns the queued tensors (in call order)."""

    def __init__(self, eps_list):
        self.eps = list(eps_list)
        self.calls = []

    def __enter__(self):
        self._orig = torch.randn_like

        def fake(t, *a, **k):
            e = self.eps.pop(0)
            assert tuple(e.shape) == tuple(t.shape), (e.shape, t.shape)
            self.calls.append(tuple(t.shape))
            return e.to(t.dtype)

        torch.randn_like = fake
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig
        return False
