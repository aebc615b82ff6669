"""Trainer with the reference's surface (trainer_motion_vae.py:15-146, 239-283).

Differences that are deliberate: the optimiser is `HipAdam` (same maths and state_dict format as torch.optim.Adam,
one HIP launch per step), `gen_update` reads the nine loss scalars back with ONE device->host copy instead of nine
`.item()` calls, and under torch.distributed the flat gradient bucket is all-reduced (mean) over RCCL before the
optimiser step (ddp.py) — the reference only had an unused nn.DataParallel switch.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.init as init

from . import ddp
from .optim import HipAdam
from .seq_two_hier_sa_vae import TwoHierSAVAEModel, get_scheduler
from .trajectory_pred_model import TrajectoryModel
from .utils_common import write_loss  # noqa: F401  (re-exported like the reference module does)

_LOSS_NAMES = ("total", "kl", "rec_6d", "rec_rot", "rec_pose", "rec_joint_pos", "rec_root_v", "rec_linear_v",
               "rec_angular_v")


class Trainer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg['model_name'] == "TwoHierSAVAEModel":
            self.model = TwoHierSAVAEModel(cfg)
        elif cfg['model_name'] == "TrajectoryModel":
            self.model = TrajectoryModel(cfg)
        else:
            raise ValueError("unknown model_name %r" % cfg['model_name'])
        self.cfg = cfg
        gen_params = [p for p in self.model.parameters() if p.requires_grad]
        self.gen_opt = HipAdam(gen_params, lr=cfg['lr'], weight_decay=cfg['weight_decay'])
        self.gen_scheduler = get_scheduler(self.gen_opt, cfg)
        self.apply(weights_init(cfg['init']))
        self._host_losses = None

    def _this_model(self):
        return self.model.module if isinstance(self.model, nn.DataParallel) else self.model

    def gen_update(self, data, hp, iterations, multigpus, validation_flag=False):
        model = self._this_model()
        if not validation_flag:
            self.gen_opt.zero_grad(set_to_none=True)
        out = model(data, hp, iterations, multigpus=multigpus, validation_flag=validation_flag)
        is_vae = self.cfg['model_name'] == "TwoHierSAVAEModel"
        kl_list = out[9] if is_vae else []      # TrajectoryModel returns nine values (trainer_motion_vae.py:43-45, 72-74)
        prefix = "loss_val_" if validation_flag else "loss_"
        # the reference takes torch.mean of each (a no-op unless nn.DataParallel gathered one value per replica)
        for name, val in zip(_LOSS_NAMES, out[:9]):
            setattr(self, prefix + name, val if val.numel() == 1 else torch.mean(val))
        for i, k in enumerate(kl_list):
            setattr(self, "loss_hier_kl_%d" % (i + 1), k if k.numel() == 1 else torch.mean(k))
        if not validation_flag:
            if ddp.is_distributed():
                if is_vae:
                    eng = model.engine()
                    n = eng.active_grad_floats(iterations >= hp['iteration_interval'])
                else:
                    eng = model.last_engine
                    n = eng.active_grad_floats()
                ddp.all_reduce_mean_(eng.flat_grad[:n])
            self.gen_opt.step()
            if self.gen_scheduler is not None:
                self.gen_scheduler.step()
        # one device->host transfer for the scalars the reference fetched with nine .item() calls
        eng_losses = getattr(model, "last_losses", None)
        if eng_losses is not None:
            v = eng_losses.tolist()
            if not is_vae:
                return tuple(v[:9])
            return (v[0], v[1], v[2], v[3], v[4], 0.0, 0.0, 0.0, 0.0)
        vals = torch.stack([getattr(self, prefix + n).reshape(()) for n in _LOSS_NAMES]).tolist()
        return tuple(vals)

    def gen_update_graph(self, data, hp, iterations, multigpus=False):
        """gen_update with the step replayed from a hipGraph (graph_step.GraphedStep: forward, backward, gradient
        all-reduce slices and the optimiser as a fixed launch sequence; re-captured at the phase switch).  Same nine floats;
        the `loss_*` attributes hold python floats.  A batch of another size than the captured one takes the eager path."""
        from .graph_step import GraphedStep
        gs = getattr(self, "_graphed", None)
        B = data[0].shape[0]
        if gs is not None and gs.r6.shape[0] != B:
            return self.gen_update(data, hp, iterations, multigpus)
        if gs is None:
            gs = self._graphed = GraphedStep(self, hp, data, iterations)
        else:
            gs.set_batch(data)
            gs.it = int(iterations)
        out = gs.run()
        for name, val in zip(_LOSS_NAMES, out):
            setattr(self, "loss_" + name, val)
        return out

    def resume(self, checkpoint_dir, hp, multigpus):
        this_model = self._this_model()
        last_model_name = get_model_list(checkpoint_dir, "gen")
        state_dict = torch.load(last_model_name, map_location="cpu", weights_only=True)
        model_dict = this_model.state_dict()
        model_dict.update(state_dict['state_dict'])
        this_model.load_state_dict(model_dict)
        iterations = int(last_model_name[-11:-3])
        state_dict = torch.load(os.path.join(checkpoint_dir, 'optimizer.pt'), map_location="cpu", weights_only=True)
        self.gen_opt.load_state_dict(state_dict['gen'])
        self.gen_scheduler = get_scheduler(self.gen_opt, hp, iterations)
        # (not in the reference's file: where the in-kernel noise stream of the reparametrisation stood, so that the resumed
        # run draws fresh noise instead of replaying the stream from its start; the reference's loader ignores extra keys)
        if state_dict.get('noise') and hasattr(this_model, 'engine') and next(this_model.parameters()).is_cuda:
            eng = this_model.engine()
            if hasattr(eng, 'load_noise_state'):
                eng.load_noise_state(state_dict['noise'])
        print('Resume from iteration %d' % iterations)
        return iterations

    def save(self, snapshot_dir, iterations, multigpus):
        if ddp.rank() != 0:
            return
        this_model = self._this_model()
        gen_name = os.path.join(snapshot_dir, 'gen_%08d.pt' % (iterations + 1))
        opt_name = os.path.join(snapshot_dir, 'optimizer.pt')
        torch.save({'state_dict': this_model.state_dict()}, gen_name, _use_new_zipfile_serialization=False)
        opt_state = {'gen': self.gen_opt.state_dict()}
        eng = getattr(this_model, '_engine', None)
        noise = eng.noise_state_dict() if (eng is not None and hasattr(eng, 'noise_state_dict')) else None
        if noise is not None:
            opt_state['noise'] = noise
        torch.save(opt_state, opt_name, _use_new_zipfile_serialization=False)

    def load_ckpt(self, ckpt_name):
        this_model = self._this_model()
        state_dict = torch.load(ckpt_name, map_location="cpu", weights_only=True)
        model_dict = this_model.state_dict()
        model_dict.update(state_dict['state_dict'])
        this_model.load_state_dict(model_dict)

    def test(self, data, hp, iterations, multigpus):
        return self._this_model().test(data, hp, iterations)

    def gen_seq(self, data, hp, iterations):
        return self._this_model().gen_seq(data, hp, iterations)

    def forward(self, *inputs):
        print('Forward function not implemented.')


def get_model_list(dirname, key):
    if not os.path.exists(dirname):
        return None
    found = sorted(os.path.join(dirname, f) for f in os.listdir(dirname)
                   if os.path.isfile(os.path.join(dirname, f)) and key in f and ".pt" in f)
    return found[-1] if found else None


def weights_init(init_type='gaussian'):
    """trainer_motion_vae.py:264-283: touches modules whose class name STARTS with Conv/Linear (so nn.Linear only;
    SkeletonConv keeps its own masked init)."""

    def init_fun(m):
        name = m.__class__.__name__
        if (name.find('Conv') == 0 or name.find('Linear') == 0) and hasattr(m, 'weight'):
            if init_type == 'gaussian':
                init.normal_(m.weight.data, 0.0, 0.02)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=math.sqrt(2))
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=math.sqrt(2))
            elif init_type != 'default':
                raise AssertionError("Unsupported initialization: {}".format(init_type))
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias.data, 0.0)

    return init_fun
