"""HipAdam: torch.optim.Adam semantics and state_dict format, stepped by one HIP kernel launch per step value.

trainer_motion_vae.py:29-31 builds `torch.optim.Adam(params, lr, weight_decay)`; its arithmetic (L2 decay folded
into the gradient, lerp first moment, bias corrections) is restated in csrc/elementwise_kernels.hip::adam_kernel.
Parameters whose `.grad` is None are skipped exactly like torch does (no decay, no moment update, step not
advanced) — the reference relies on that for the never-used latent heads.
"""
import ctypes as C
from collections import defaultdict

import torch

from . import _lib


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._host_step = {}
        self._cache = {}
        self._lr_dev = {}

    def _state_for(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = torch.tensor(0.0)
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            self._host_step[p] = 0
        elif p not in self._host_step:
            self._host_step[p] = int(float(st['step']))
        return st

    def lr_tensor(self, gi, device):
        key = (gi, str(device))
        if key not in self._lr_dev:
            self._lr_dev[key] = [torch.zeros(1, device=device), None]
        return self._lr_dev[key]

    def sync_lr(self):
        """Push the host-side learning rate(s) to their device scalars (call after scheduler.step())."""
        for gi, group in enumerate(self.param_groups):
            for (g2, _), slot in self._lr_dev.items():
                if g2 == gi and slot[1] != group['lr']:
                    slot[0].fill_(group['lr'])
                    slot[1] = group['lr']

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            active = [p for p in group['params'] if p.grad is not None]
            if not active:
                continue
            if not active[0].is_cuda:
                raise _lib.HmvaeError("HipAdam needs parameters on a HIP device; there is no CPU path")
            buckets = defaultdict(list)
            for p in active:
                self._state_for(p)
                buckets[self._host_step[p]].append(p)
            slot = self.lr_tensor(gi, active[0].device)
            if slot[1] != group['lr']:
                slot[0].fill_(group['lr'])
                slot[1] = group['lr']
            b1, b2 = group['betas']
            for stepval, plist in buckets.items():
                ptrs = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr(),
                              self.state[p]['exp_avg_sq'].data_ptr(), p.numel()) for p in plist)
                ent = self._cache.get(ptrs)
                if ent is None:
                    tab = (_lib.AdamTensor * len(plist))()
                    for i, t in enumerate(ptrs):
                        tab[i] = _lib.AdamTensor(*t)
                    ent = dict(tab=torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(plist[0].device),
                               step=torch.full((1,), stepval, dtype=torch.int32, device=plist[0].device),
                               host=stepval, max_numel=max(p.numel() for p in plist), params=list(plist), live=False)
                    self._evict()
                    self._cache[ptrs] = ent
                if ent['host'] == stepval:
                    ent['step'].add_(1)
                else:
                    ent['step'].fill_(stepval + 1)
                ent['host'] = stepval + 1
                _lib.check(lib.hmvae_adam_step(ent['tab'].data_ptr(), len(plist), ent['max_numel'], slot[0].data_ptr(),
                                               ent['step'].data_ptr(), b1, b2, group['eps'], group['weight_decay'],
                                               _lib.stream()), "adam_step")
                for p in plist:
                    self._host_step[p] = stepval + 1
        return None

    MAX_CACHE = 16

    def _evict(self):
        """Make room in the table cache: drop the OLDEST entries that no captured hipGraph addresses.  An entry marked
        live is baked into a graph (its `tab` and `step` device tensors are what the replay reads and counts in): it is
        never dropped, and a cache full of live entries simply grows."""
        if len(self._cache) < self.MAX_CACHE:
            return
        for key in [k for k, e in self._cache.items() if not e.get('live')]:
            if len(self._cache) < self.MAX_CACHE:
                break
            del self._cache[key]

    def advance_host_steps(self, n):
        """A graph replay advances the device step counters but not the host mirror; account for it here."""
        if not n:
            return
        for ent in self._cache.values():
            if ent.get('live'):
                ent['host'] += n
                for p in ent['params']:
                    self._host_step[p] += n

    def resync_device_steps(self):
        """After a graph capture (which records the counter increments without executing them) make the host
        mirror and the device counters agree again, and mark the captured entries as the live ones."""
        for ent in self._cache.values():
            ent['host'] = self._host_step[ent['params'][0]]
            ent['step'].fill_(ent['host'])
            ent['live'] = all(p.grad is not None for p in ent['params'])

    def reset_state(self):
        """Back to "never stepped" without touching a pointer: moments zeroed in place, host and device step counters 0,
        learning rates back to their initial values.  Captured graphs that address this optimiser's tables stay valid --
        this is what lets one refinement context (and its two graphs) serve window after window."""
        for p, st in self.state.items():
            if len(st):
                st['exp_avg'].zero_()
                st['exp_avg_sq'].zero_()
                st['step'] = torch.tensor(0.0)
                self._host_step[p] = 0
        for ent in self._cache.values():
            ent['host'] = 0
            ent['step'].fill_(0)
        for group in self.param_groups:
            if 'initial_lr' in group:
                group['lr'] = group['initial_lr']
        self.sync_lr()

    def state_dict(self):
        for p, n in self._host_step.items():
            if p in self.state and len(self.state[p]):
                self.state[p]['step'] = torch.tensor(float(n))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._host_step, self._cache = {}, {}
        for group in self.param_groups:
            for p in group['params']:
                st = self.state.get(p)
                if st:
                    self._host_step[p] = int(float(st['step']))
                    st['step'] = torch.tensor(float(self._host_step[p]))
