"""hipGraph replay of the training step (torch.cuda.CUDAGraph = hipGraph on ROCm).

The fused engine issues ~170 short kernel launches per step; captured once, a step becomes one graph launch plus
one 32-byte device->host copy of the loss scalars.  Under DDP the step is captured as three or four graphs
([forward + decoder backward] | [encoder backward, coarse | fine] | [optimiser]) with the RCCL all-reduces of the
gradient bucket's slices launched eagerly between them (asynchronous for all but the last slice).
"""
import torch

from . import ddp

# Graph lifetime.  On the HIP runtime that the torch 2.10+rocm7.0 wheel bundles (libamdhip64 7.0.2, the copy every Python
# process here runs on) destroying a hipGraphExec that was captured across several streams and launching / synchronising
# afterwards crashes the process sporadically INSIDE the runtime: SIGSEGV in hip::Graph::UpdateStreams (from hipGraphLaunch)
# or in amd::Event::processCallbacks (the runtime's signal-handler thread), "corrupted size vs. prev_size" /
# "free(): invalid pointer" from glibc.  tools/repro/graph_destroy_torch.py shows it with NO code of this repository
# (torch streams, events and torch.cuda.graph only: 5 crashes in 17 runs of 100 capture / destroy rounds); the same
# program on the image's ROCm 7.2.0 runtime (LD_PRELOAD=/opt/rocm/lib/libamdhip64.so.7) ran 12 of 12 clean, and with the
# graphs kept alive 7 of 7 clean on the bundled runtime; the pure-HIP program beside it (graph_destroy_repro.hip) does not
# trigger it on either runtime, so the trigger involves something torch's capture adds.  Joining only the side streams a
# step forked (engine._join) or pre-creating the events changes nothing.  Rounds and logs: tools/repro/README.md.
# So on a runtime older than 7.2 replaced graphs are PARKED (kept alive until the process ends: a few hundred KB of host
# memory each; training re-captures once, at the phase switch; the refinement loop reuses its graphs across windows) and
# on a newer one they are destroyed normally, after a device synchronisation.  HMVAE_GRAPH_PARK=0/1 overrides.
import os as _os
import warnings as _warnings

_RETIRED = []
MAX_RETIRED = 64
# Parked graphs are bounded: per-item capturers stop capturing at MAX_RETIRED (may_capture), and whatever still retires
# graphs beyond that (a trainer re-built hundreds of times in one process) has the OLDEST ones destroyed once TRIM_AT are
# parked -- deterministic, oldest first, after a device synchronisation, down to MAX_RETIRED.  Destroying is the unsafe
# operation on the 7.0.x runtime (about one crash per 300 destroys in tools/repro), which is why the threshold sits far
# above anything training, the refinement loop or the test suite reach; on runtimes >= 7.2 nothing is ever parked.
TRIM_AT = int(_os.environ.get("HMVAE_GRAPH_TRIM_AT", "256"))
_warned = [False]


def hip_runtime_version():
    """hipRuntimeGetVersion() of the libamdhip64 mapped into this process (0 if it cannot be asked)."""
    import ctypes
    try:
        v = ctypes.c_int(0)
        if ctypes.CDLL("libamdhip64.so.7").hipRuntimeGetVersion(ctypes.byref(v)) == 0:
            return int(v.value)
    except OSError:
        pass
    return 0


def _park_default():
    env = _os.environ.get("HMVAE_GRAPH_PARK")
    if env is not None:
        return env != "0"
    return hip_runtime_version() < 70200000


PARK = None           # decided on first use (the runtime library is mapped by then)


def parking():
    global PARK
    if PARK is None:
        PARK = _park_default()
    return PARK


def retire(*graphs):
    graphs = [g for g in graphs if g is not None]
    if not graphs:
        return
    if parking():
        _RETIRED.extend(graphs)
        if len(_RETIRED) > TRIM_AT:
            torch.cuda.synchronize()
            del _RETIRED[:len(_RETIRED) - MAX_RETIRED]          # oldest first
    else:
        torch.cuda.synchronize()            # nothing of the graph is in flight when its last reference goes


def may_capture():
    """False once MAX_RETIRED graphs are parked: callers that would capture per work item launch eagerly from then on
    (said once, loudly -- replays slow down with many hundred live graphs and the memory is never returned)."""
    ok = (not parking()) or len(_RETIRED) < MAX_RETIRED
    if not ok and not _warned[0]:
        _warned[0] = True
        _warnings.warn("hm_vae_amd: %d replaced hipGraphs are parked (HIP runtime %d destroys them unsafely); further "
                       "per-item captures fall back to eager launches" % (len(_RETIRED), hip_runtime_version()))
    return ok


class GraphedStep:
    def __init__(self, trainer, cfg, data, iterations, use_graph=True, warmup=3):
        self.tr, self.cfg = trainer, cfg
        self.model = trainer._this_model()
        self.it = int(iterations)
        self.use_graph = bool(use_graph)
        dev = next(self.model.parameters()).device
        self.is_vae = cfg['model_name'] == "TwoHierSAVAEModel"
        # static input buffers: the graph reads these addresses on every replay
        # (TrajectoryModel: its encoder input -- joint positions, or the 6D rotations -- and the root velocity)
        self.slots = (0, 1) if self.is_vae else ((3 if cfg['trajectory_input_joint_pos'] else 0), 6)
        self.r6 = data[self.slots[0]].to(device=dev, dtype=torch.float32).contiguous().clone()
        self.rmat = data[self.slots[1]].to(device=dev, dtype=torch.float32).contiguous().clone()
        lst = list(data)
        lst[self.slots[0]], lst[self.slots[1]] = self.r6, self.rmat
        self.data = tuple(lst)
        self.graphs = None
        self.g1b = None
        self.g1c = None
        import os
        # HMVAE_DDP_SPLIT=0: one all-reduce after the whole backward.  HMVAE_FORCE_SPLIT=1: take the multi-graph path
        # on a single GPU too (tests / overhead measurement).
        self.split = os.environ.get("HMVAE_DDP_SPLIT", "1") != "0" and self.is_vae
        self.force_split = os.environ.get("HMVAE_FORCE_SPLIT", "0") == "1"
        # HMVAE_DDP_BUCKETS=3 (default): the encoder's backward is cut once more -- coarse layers | fine layers -- so that
        # only the fine layers' few MB are all-reduced exposed; 2: [decoder | encoder]
        self.buckets = int(os.environ.get("HMVAE_DDP_BUCKETS", "3"))
        # HMVAE_DDP_INGRAPH=1: ONE graph for the whole DDP step -- the slices' all-reduces are captured inside it, issued
        # from a helper stream that waits for exactly the launches that write the slice (no join of the side streams, no
        # graph cut: the two cuts of the multi-graph step cost 3.6 % + 4.6 % of a one-GPU step).  Needs a backend whose
        # collectives can be captured (RCCL); exercised on one GPU with a world of one rank.  Off by default until it has
        # run on a multi-GPU node.
        self.ingraph = os.environ.get("HMVAE_DDP_INGRAPH", "0") == "1" and self.is_vae
        self._comm = None
        self._handles = []
        self._pinned = None
        self.phase2 = None
        self.warmup = warmup

    def set_batch(self, data):
        self.r6.copy_(data[self.slots[0]], non_blocking=True)
        self.rmat.copy_(data[self.slots[1]], non_blocking=True)

    def _engine(self):
        return self.model.engine() if self.is_vae else self.model.engine(self.r6.shape[1])

    def _tuple(self, ls):
        if self.is_vae:
            return (ls[0], ls[1], ls[2], ls[3], ls[4], 0.0, 0.0, 0.0, 0.0)
        return tuple(ls[:9])

    # -- the step body, split at the collectives --
    def _fwd_bwd(self):
        eng = self._engine()
        if self.is_vae:
            self.losses = eng.train_step(self.r6, self.rmat, self.it, eps_list=None, backward=True)
        else:
            self.losses = eng.train_step(self.r6, self.rmat, backward=True)

    def _fwd_bwd_dec(self):             # forward + the decoder's backward: every decoder gradient is final afterwards
        eng = self.model.engine()
        self.losses = eng.train_step(self.r6, self.rmat, self.it, eps_list=None, backward=True, stage="dec")

    def _bwd_enc(self, part="enc"):
        self.model.engine().backward_rest(part)

    def _optim(self):
        self.tr.gen_opt.step()

    # -- the in-graph DDP step --
    def _slice_hook(self, which):
        """Start the all-reduce of one slice of the gradient bucket: the collective is issued from the helper stream
        self._comm, which waits for the main stream and for every side stream forked so far -- i.e. for every launch
        that writes the slice -- while the main stream goes on with the backward pass."""
        import torch.distributed as dist
        eng = self.model.engine()
        n = eng.active_grad_floats(self.phase2)
        lo, hi = {"dec": (0, eng.n_dec_floats), "enc_hi": (eng.n_dec_floats, eng.n_hi_floats),
                  "rest": (eng.n_hi_floats, n)}[which]
        main = torch.cuda.current_stream()
        eng._stream_wait(self._comm, main, ("comm_main", which))
        if eng._side is not None:
            for i in sorted(eng._forked):
                eng._stream_wait(self._comm, eng._side[i], ("comm_side", which, i))
        if hi > lo:
            with torch.cuda.stream(self._comm):
                self._handles.append(dist.all_reduce(eng.flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def _ingraph_body(self):
        import torch.distributed as dist
        eng = self.model.engine()
        self._handles = []
        self.losses = eng.train_step(self.r6, self.rmat, self.it, eps_list=None, backward=True, slice_hook=self._slice_hook)
        self._slice_hook("rest")                       # (the step's final join has happened: only the main stream is waited for)
        for h in self._handles:
            h.wait()                                   # the current stream waits for the collective
        eng._stream_wait(torch.cuda.current_stream(), self._comm, "comm_join")
        eng.flat_grad[:eng.active_grad_floats(self.phase2)].mul_(1.0 / dist.get_world_size())
        self._optim()

    def _reduce(self):
        if ddp.is_distributed():
            eng = self._engine()
            ddp.all_reduce_mean_(eng.flat_grad[:eng.active_grad_floats(self.phase2)])

    def _phase2(self):
        return self.is_vae and self.it >= self.cfg['iteration_interval']

    def _snapshot(self):
        opt = self.tr.gen_opt
        params = [p for g in opt.param_groups for p in g['params']]
        state = {}
        for p in params:
            st = opt.state.get(p)
            if st:
                state[p] = (st['exp_avg'].clone(), st['exp_avg_sq'].clone(), opt._host_step.get(p, 0))
        return [p.detach().clone() for p in params], params, state

    def _restore(self, snap):
        values, params, state = snap
        opt = self.tr.gen_opt
        with torch.no_grad():
            for p, v in zip(params, values):
                p.copy_(v)
            for p in params:
                st = opt.state.get(p)
                if not st:
                    continue
                if p in state:
                    st['exp_avg'].copy_(state[p][0])
                    st['exp_avg_sq'].copy_(state[p][1])
                    opt._host_step[p] = state[p][2]
                else:                                  # created by the warm-up: back to "never stepped"
                    st['exp_avg'].zero_()
                    st['exp_avg_sq'].zero_()
                    opt._host_step[p] = 0
        for ent in opt._cache.values():                # device step counters follow the host mirror again
            ent['host'] = opt._host_step[ent['params'][0]]
            ent['step'].fill_(ent['host'])

    def _capture(self):
        self.phase2 = self._phase2()
        # warm up eagerly on a side stream: creates workspaces, index tables and the Adam state outside the graph
        # (the warm-up steps are real optimiser steps: parameters and Adam state are put back afterwards, so a run with
        # graph replay takes exactly the steps an eager run takes)
        snap = self._snapshot()
        use_ingraph = self.ingraph and ddp.group_ready()
        if use_ingraph and self._comm is None:
            self._comm = torch.cuda.Stream()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                if use_ingraph:
                    self._ingraph_body()
                else:
                    self._fwd_bwd()
                    self._reduce()
                    self._optim()
                self.tr.gen_opt.advance_host_steps(0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._restore(snap)
        if self.graphs is not None:
            retire(*self.graphs)
        retire(self.g1b, self.g1c)
        g1 = torch.cuda.CUDAGraph()
        g2 = None
        self.g1b = self.g1c = None
        if use_ingraph:
            host_before = dict(self.tr.gen_opt._host_step)
            with torch.cuda.graph(g1):
                self._ingraph_body()
            self.tr.gen_opt._host_step = host_before
            self.tr.gen_opt.resync_device_steps()
        elif ddp.is_distributed() or self.force_split:
            if self.split:
                # three graphs: [forward + decoder backward] -> async all-reduce of the decoder gradients ||
                # [encoder backward] -> all-reduce of the rest -> [optimiser]
                with torch.cuda.graph(g1):
                    self._fwd_bwd_dec()
                self.g1b = torch.cuda.CUDAGraph()
                if self.buckets >= 3:
                    with torch.cuda.graph(self.g1b):
                        self._bwd_enc("enc_hi")
                    self.g1c = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.g1c):
                        self._bwd_enc("enc_lo")
                else:
                    with torch.cuda.graph(self.g1b):
                        self._bwd_enc()
            else:
                with torch.cuda.graph(g1):
                    self._fwd_bwd()
            g2 = torch.cuda.CUDAGraph()
            host_before = dict(self.tr.gen_opt._host_step)
            with torch.cuda.graph(g2):
                self._optim()
            self.tr.gen_opt._host_step = host_before        # capture does not execute: undo the host-side count
            self.tr.gen_opt.resync_device_steps()
        else:
            host_before = dict(self.tr.gen_opt._host_step)
            with torch.cuda.graph(g1):
                self._fwd_bwd()
                self._optim()
            self.tr.gen_opt._host_step = host_before
            self.tr.gen_opt.resync_device_steps()
        self.graphs = (g1, g2)

    def __del__(self):
        try:
            if self.graphs is not None:
                retire(*self.graphs)
            retire(self.g1b, self.g1c)
        except Exception:
            pass

    def run(self, lag=False):
        """One training step; returns the nine loss floats like Trainer.gen_update.

        lag=True: the read-back does not stall the GPU -- the losses are copied to pinned host memory asynchronously and
        the call returns the PREVIOUS step's values (the first call returns this step's), so the host is already
        queueing step n+1 while step n runs (a blocking read-back leaves the GPU idle for the ~80 us it takes the host
        to wake up and launch the next graph)."""
        if not self.use_graph:
            out = self.tr.gen_update(self.data, self.cfg, self.it, False)
            self.it += 1
            return out
        if self.graphs is None or self.phase2 != self._phase2():
            self._capture()
        g1, g2 = self.graphs
        g1.replay()
        if g2 is not None:
            if self.g1b is not None:
                eng = self.model.engine()
                ddp.reduce_between_segments(eng.flat_grad, eng.n_dec_floats, eng.n_hi_floats,
                                            eng.active_grad_floats(self.phase2), self.g1b.replay,
                                            self.g1c.replay if self.g1c is not None else None)
            else:
                self._reduce()
            g2.replay()
        opt = self.tr.gen_opt
        opt.advance_host_steps(1)
        if self.tr.gen_scheduler is not None:
            self.tr.gen_scheduler.step()
            opt.sync_lr()
        self.it += 1
        if not lag:
            return self._tuple(self.losses.tolist())       # the step's only host synchronisation
        if self._pinned is None:
            self._pinned = [torch.empty(self.losses.numel(), dtype=torch.float32).pin_memory() for _ in range(2)]
            self._events = [torch.cuda.Event(), torch.cuda.Event()]
            self._n_lag = 0
        k = self._n_lag & 1
        self._pinned[k].copy_(self.losses, non_blocking=True)
        self._events[k].record()
        self._n_lag += 1
        j = k if self._n_lag == 1 else 1 - k    # previous step's slot (this step's on the very first call)
        self._events[j].synchronize()
        return self._tuple(self._pinned[j].tolist())

    def last_losses(self):
        """Blocking read of the most recent step's losses (after a run(lag=True) loop)."""
        return self._tuple(self.losses.tolist())
