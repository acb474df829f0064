"""Train-step runner: optimizer, gradient clipping, cyclic LR / momentum schedule and the
data-parallel wrap — the pieces of the reference's train API on the hot path
(mmdet3d/apis/train.py:180-321 ``train_detector``; mmdet3d/utils/util_distribution.py:38-65
``build_ddp``; config keys configs/gga/gga_kitti_config.py:233-260). mmcv's
``EpochBasedRunner`` / ``OptimizerHook`` / ``CyclicLrUpdaterHook`` /
``CyclicMomentumUpdaterHook`` are third-party (not in the tree): their published behaviour
is restated here (parity unpinned).

One process per GPU; gradients are all-reduced by ``torch.nn.parallel.DistributedDataParallel``
over RCCL (backend name ``nccl`` on ROCm) with ``broadcast_buffers=False`` like the reference.
"""
import collections
import math
import os
import time

import torch
import torch.distributed as dist


def annealing_cos(start, end, factor, weight=1.0):
    cos_out = math.cos(math.pi * factor) + 1
    return end + 0.5 * weight * (start - end) * cos_out


class CyclicSchedule:
    """mmcv Cyclic{Lr,Momentum}UpdaterHook, by iteration: one up phase of ``step_ratio_up`` of
    the cycle from ratio 1 to ``target_ratio[0]``, then down to ``target_ratio[1]``, cosine."""

    def __init__(self, base, max_iters, target_ratio=(10, 1e-4), cyclic_times=1, step_ratio_up=0.4):
        self.base = base
        per_phase = max(1, max_iters // cyclic_times)
        up = int(step_ratio_up * per_phase)
        self.per_phase = per_phase
        self.phases = [(0, up, 1.0, target_ratio[0]), (up, per_phase, target_ratio[0], target_ratio[1])]

    def __call__(self, it):
        cur = it % self.per_phase
        for start, end, r0, r1 in self.phases:
            if start <= cur < end:
                return annealing_cos(self.base * r0, self.base * r1, (cur - start) / (end - start))
        return self.base * self.phases[-1][3]


class StepSchedule:
    """mmcv StepLrUpdaterHook with linear warm-up, by iteration (configs/gga/gga_pdg.py: epochs
    ``step=[32, 44]``, ``warmup_iters=500``, ``warmup_ratio=1/3``): lr = base * gamma^(#steps passed),
    and during warm-up lr * (1 - (1 - it / warmup_iters) * (1 - warmup_ratio))."""

    def __init__(self, base, step, iters_per_epoch=1, gamma=0.1, warmup=None, warmup_iters=0, warmup_ratio=0.1):
        self.base, self.gamma = base, gamma
        self.step = sorted(int(s) * int(iters_per_epoch) for s in ([step] if isinstance(step, int) else step))
        assert warmup in (None, 'linear'), 'only linear warm-up is used by configs/gga'
        self.warmup_iters, self.warmup_ratio = (warmup_iters if warmup else 0), warmup_ratio

    def __call__(self, it):
        lr = self.base * self.gamma ** sum(it >= s for s in self.step)
        if it < self.warmup_iters:
            lr *= 1 - (1 - it / self.warmup_iters) * (1 - self.warmup_ratio)
        return lr


def setup_multi_processes(cfg):
    """The host-thread part of ``mmdet3d/utils/setup_env.py:10-53`` (tools/train.py:127 calls it before the model is
    built): with more than one data-loader worker per GPU, OpenMP / MKL math threads default to ONE per process unless the
    environment says otherwise. Here the limit is also applied to the pools that are already initialised (torch's
    intra-op pool, numpy's BLAS through threadpoolctl): the train step's host side only launches kernels and packs a few
    KB of targets, and on a many-core host the default pools (one thread per core) stall it at their barriers - measured
    on 256-core boxes under load: 36.5 ms per PointPillars step with the limit against 36.8 - 43 ms without."""
    data = cfg.get('data') or {}
    workers = data.get('workers_per_gpu', 1)
    if 'train_dataloader' in data:
        workers = max(data['train_dataloader'].get('workers_per_gpu', 1), workers)
    if workers <= 1:
        return
    limited = False
    for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
        if var not in os.environ:
            os.environ[var] = '1'
            limited = True
    if limited:
        torch.set_num_threads(1)
        try:
            from threadpoolctl import threadpool_limits
            setup_multi_processes._blas_limit = threadpool_limits(limits=1)       # kept alive: the limit stays
        except ImportError:
            pass


_NORM_TYPES = (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm, torch.nn.LayerNorm,
               torch.nn.modules.instancenorm._InstanceNorm)


def paramwise_settings(model, base_lr, base_wd, pw):
    """{parameter name: (lr, weight_decay)} as mmcv 1.4.8 - 1.6.0 ``DefaultOptimizerConstructor.add_params`` assigns them
    (mmcv/runner/optimizer/default_constructor.py; the wheel is un-vendored - restated from the published code, parity
    unpinned): it walks the modules, and for the parameters a module owns directly

    * ``bias`` of anything that is neither a normalisation layer nor part of a DCN module: lr * ``bias_lr_mult``,
      weight_decay * ``bias_decay_mult``;
    * every parameter of a normalisation layer (_BatchNorm, _InstanceNorm, GroupNorm, LayerNorm): base lr,
      weight_decay * ``norm_decay_mult`` (when given);
    * depth-wise convolutions (in_channels == groups): weight_decay * ``dwconv_decay_mult`` (when given);
    * the DIRECT children of a ``DeformConv2d`` / ``ModulatedDeformConv2d`` module (its ``conv_offset``): ``bias_lr_mult`` /
      ``bias_decay_mult`` do not apply and ``dcn_offset_lr_mult`` scales the rate (when given). The flag is recomputed
      per module after its own parameters were handled, so the DCN module's own ``bias`` does take the bias multipliers
      (mmcv's comment: "bias_lr_mult affects all bias parameters except for norm.bias dcn.conv_offset.bias")."""
    from . import dcn
    dcn_types = tuple(t for t in (getattr(dcn, 'ModulatedDeformConv2d', None), getattr(dcn, 'ModulatedDeformConv2dPack', None),
                                  getattr(dcn, 'DeformConv2d', None)) if t is not None)
    bias_lr, bias_wd = pw.get('bias_lr_mult', 1.0), pw.get('bias_decay_mult', 1.0)
    norm_wd, dw_wd = pw.get('norm_decay_mult', 1.0), pw.get('dwconv_decay_mult', 1.0)
    offset_lr = pw.get('dcn_offset_lr_mult', 1.0)
    out = {}

    def walk(module, prefix, is_dcn):
        is_norm = isinstance(module, _NORM_TYPES)
        is_dw = isinstance(module, torch.nn.Conv2d) and module.in_channels == module.groups
        for name, p in module.named_parameters(recurse=False):
            if not p.requires_grad:
                continue
            lr, wd = base_lr, base_wd
            if name == 'bias' and not (is_norm or is_dcn):
                lr = base_lr * bias_lr
            if prefix.find('conv_offset') != -1 and is_dcn and isinstance(module, torch.nn.Conv2d):
                lr = base_lr * offset_lr
            if base_wd is not None:
                if is_norm:
                    wd = base_wd * norm_wd
                elif is_dw:
                    wd = base_wd * dw_wd
                elif name == 'bias' and not is_dcn:
                    wd = base_wd * bias_wd
            out[prefix + name] = (lr, wd)
        for cname, child in module.named_children():
            walk(child, f'{prefix}{cname}.', isinstance(module, dcn_types))

    walk(model, '', False)
    return out


def build_optimizer(model, cfg):
    """AdamW (configs/gga/gga_kitti_config.py:233) or SGD with mmcv's ``paramwise_cfg`` (configs/gga/gga_pdg.py:
    ``bias_lr_mult=2, bias_decay_mult=0``) - see ``paramwise_settings`` for which parameters the multipliers reach."""
    cfg = dict(cfg)
    typ = cfg.pop('type')
    pw = cfg.pop('paramwise_cfg', None) or {}
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    if pw:
        unknown = set(pw) - {'bias_lr_mult', 'bias_decay_mult', 'norm_decay_mult', 'dwconv_decay_mult', 'dcn_offset_lr_mult'}
        if unknown:
            raise KeyError(f'paramwise_cfg keys {sorted(unknown)} are not used by configs/gga')
        # mmcv builds one group per parameter; parameters with equal settings are updated alike, so they share a group
        # here (a few groups instead of ~350: the optimizer's multi-tensor kernels then run once per step, not per parameter)
        settings = paramwise_settings(model, cfg['lr'], cfg.get('weight_decay'), pw)
        groups = {}
        for n, p in named:
            lr, wd = settings[n]
            g = groups.setdefault((lr, wd), dict(params=[], lr=lr))
            if wd is not None:
                g['weight_decay'] = wd
            g['params'].append(p)
        params = list(groups.values())
    else:
        params = [p for _, p in named]
    if typ == 'AdamW':
        opt = torch.optim.AdamW(params, fused=all(p.is_cuda for _, p in named), **cfg)
    elif typ == 'SGD':
        opt = torch.optim.SGD(params, **cfg)
    else:
        raise KeyError(f'optimizer {typ} is not used by configs/gga (AdamW, SGD)')
    # The packed operands of the convolution weights are stale after every step, and torch's fused optimizers do not bump the
    # parameters' version counters: weight_bank registers a GLOBAL optimizer-step hook at import (every optimizer of the
    # process, also one the user built) - make sure it is imported before the first step.
    from . import weight_bank  # noqa: F401
    return opt


def init_dist():
    """torchrun / torch.distributed.run environment -> (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        # RCCL registers as 'nccl' on ROCm; GGA_DIST_BACKEND=gloo is for single-GPU / CPU testing
        backend = os.environ.get('GGA_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend)
    return rank, world, local_rank


def init_random_seed(seed=None, device='cuda'):
    """``mmdet3d/apis/train.py:27-55``: the seed all ranks share - ``seed`` if given, else a random one drawn on rank 0 and
    broadcast (the one collective of the start-up; ``tools/dist_train.sh:19`` passes ``--seed 0``, so the reference never takes
    this branch)."""
    if seed is not None:
        return int(seed)
    import numpy as np
    seed = int(np.random.randint(2 ** 31))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seed
    on = torch.device(device) if (torch.cuda.is_available() and dist.get_backend() == 'nccl') else torch.device('cpu')
    t = torch.tensor(seed if dist.get_rank() == 0 else 0, dtype=torch.int32, device=on)
    dist.broadcast(t, src=0)
    return int(t.item())


def set_random_seed(seed, deterministic=False):
    """``mmdet3d/apis/train.py:57-74``: python's, numpy's and torch's generators (what ``tools/train.py`` calls before it builds
    the model and the datasets: weight initialisation, the database sampler's order, the SRL draws and the loader workers'
    base seed all come from them). ``deterministic``: the framework's convolution library in its deterministic mode - this
    repo's kernels are deterministic as they are."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def build_ddp(model, device, find_unused_parameters=False):
    """MMDistributedDataParallel(device_ids=[LOCAL_RANK], broadcast_buffers=False,
    find_unused_parameters=...) (apis/train.py:222-231)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    # device_ids=None: the module already lives on this rank's device, and DDP must NOT move the
    # keyword inputs — the GGA label / in-box-point tensors are consumed on the host by
    # CenterHead_GGA.pack_targets (moving them to the GPU would cost an H2D and a syncing D2H).
    return DDP(model, broadcast_buffers=False, find_unused_parameters=find_unused_parameters)


PREFETCH_FIRST = os.environ.get('GGA_PREFETCH_FIRST', '1') == '1'


class Runner:
    """Iteration loop of the train step: schedule -> train_step -> backward (DDP all-reduce
    overlaps it) -> clip_grad_norm_ -> AdamW.

    Host reads of device values: none on the PointPillars trunk (pillar counts stay on the device).
    The sparse-conv trunk must know its level sizes on the host (data-dependent allocations, as
    in spconv): with ``step(data, next_data=...)`` that point-only front (voxelize + index
    structures) of the NEXT batch runs on a side stream right after this step's optimizer was
    queued, so those reads wait for a handful of small kernels instead of draining the main
    stream's queue; without ``next_data`` they happen in line."""

    def __init__(self, model, cfg, max_iters, distributed=False, device=None, iters_per_epoch=None):
        self.raw_model = model
        self.device = device or next(model.parameters()).device
        self.model = build_ddp(model, self.device, cfg.get('find_unused_parameters', False)) if distributed else model
        self.optimizer = build_optimizer(model, cfg.optimizer)
        gc = (cfg.get('optimizer_config') or {}).get('grad_clip')
        self.grad_clip = dict(gc) if gc else None
        base_lr = cfg.optimizer['lr']
        base_m = cfg.optimizer['betas'][0] if 'betas' in cfg.optimizer else None
        lrc, mc = dict(cfg.get('lr_config') or {}), dict(cfg.get('momentum_config') or {})
        self.lr_sched = self.mom_sched = None
        if lrc.get('policy') == 'cyclic':
            self.lr_sched = CyclicSchedule(base_lr, max_iters, lrc.get('target_ratio', (10, 1e-4)),
                                           lrc.get('cyclic_times', 1), lrc.get('step_ratio_up', 0.4))
        elif lrc.get('policy') == 'step':
            self.lr_sched = StepSchedule(1.0, lrc['step'], iters_per_epoch or 1, lrc.get('gamma', 0.1), lrc.get('warmup'),
                                         lrc.get('warmup_iters', 0), lrc.get('warmup_ratio', 0.1))
        # per-group base rates (paramwise multipliers) the step schedule scales: kept IN the param groups under mmcv's key
        # ('initial_lr', LrUpdaterHook.before_run), so that they ride along in optimizer.state_dict() - a resumed run must not
        # take the already-decayed 'lr' of the checkpoint for its base
        for g in self.optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        if mc.get('policy') == 'cyclic':
            self.mom_sched = CyclicSchedule(base_m, max_iters, mc.get('target_ratio', (0.85 / 0.95, 1)),
                                            mc.get('cyclic_times', 1), mc.get('step_ratio_up', 0.4))
        self.iter = 0
        self.epoch = 0              # finished epochs (train_epochs / resume)
        self.log_interval = (cfg.get('log_config') or {}).get('interval', 50)
        # Arithmetic of the matrix kernels for the train step (dense_conv.PLANES): the two-fp16-plane form, watched by
        # the range guard - armed for iteration 0 and every `range_check_interval` iterations after it; when an operand of
        # a guarded step is not represented as well as fp32 would (dense_conv.RangeGuard) the run continues on three
        # bf16 planes. `gga_dense_planes` / `gga_range_check_interval` are extension keys of the config (not the
        # reference's); GGA_DENSE_PLANES in the environment pins the arithmetic for every caller.
        # The choice belongs to this Runner (`self.planes`): `step` installs it for the duration of the step and puts the
        # process-wide value back, so evaluation or direct operator calls in the same process keep the library default; once
        # a guard has fallen back anywhere in the process no later Runner goes below three planes again.
        from . import dense_conv
        if dense_conv.PLANES_PINNED:
            self.planes = dense_conv.PLANES
        else:
            self.planes = 3 if dense_conv.FELL_BACK else int(cfg.get('gga_dense_planes', 2))
        self.fell_back = bool(dense_conv.FELL_BACK and not dense_conv.PLANES_PINNED)      # this run left two planes through the guard
        self.range_check_interval = int(os.environ.get('GGA_RANGE_CHECK_INTERVAL', cfg.get('gga_range_check_interval', 500)))
        self.range_reports = []     # (iteration, worst share of lost elements, operands seen) per guarded step
        self._guard_next = True     # the first step of every Runner (and the first after a resume) is guarded, whatever its iteration
        self._gc_frozen = False
        self._side = None           # side stream of the input prefetch
        self._prepared = {}         # id(data dict) -> (PreparedInputs, event)
        self._retired = []          # (PreparedInputs, event after the step that consumed them)
        self._ready = collections.OrderedDict()     # id(data dict) -> (event after its upload, the dict): inputs_ready, newest last
        self._in_flight = collections.deque()       # one event per queued step, oldest first (MAX_STEPS_AHEAD)

    def prefetch(self, data):
        """Run the point-only front of the step that will consume ``data`` now, on the side stream."""
        model = self.raw_model
        if not (hasattr(model, 'prepare_inputs') and model.front_reads_counts and torch.cuda.is_available()):
            return
        pts = data.get('points')
        if pts is None or self._prepared_for(data) is not None:
            return
        if self._side is None:
            # high priority: the front is a handful of small kernels whose counts the host waits for - behind the main stream's
            # queued backward kernels they are scheduled late and the host idles (measured on the sparse trunk: 41 ms of the 62 ms
            # step spent in those waits; tools_dev/host_profile.py second)
            self._side = torch.cuda.Stream(device=self.device, priority=-1)
        # The side stream must see the batch's points complete. A batch announced with `inputs_ready` waits for that event
        # only; any other batch may have been uploaded (non_blocking) on the main stream just before this call, and the side
        # stream then waits for everything queued there - which is the whole previous step when the host runs ahead of the
        # device: the front no longer overlaps the step and the device idles while the host reads the front's counts
        # (measured on the sparse trunk, bs 8: 63.7 against 59.5 ms per step).
        ready = self._ready.get(id(data))
        if ready is not None and ready[1] is data:
            self._side.wait_event(ready[0])
        else:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._side):
            prep = model.prepare_inputs(pts)
            ev = torch.cuda.Event()
            ev.record(self._side)
        # The tensors were allocated on the side stream and are consumed on the main one. Instead of
        # Tensor.record_stream (whose deferred frees made the caching allocator fall back to hipMalloc for
        # most of the ~300 buffers of the next prefetch: 47 ms of host time per step) the prepared inputs are
        # kept alive until an event recorded after the consuming step has completed (`_retired`), so their
        # blocks return to the side stream's pool only when no main-stream kernel can still read them.
        # keyed by the identity of the batch dict AND of its points list, holding a weak reference to neither: an entry is
        # only ever used for the very object it was made from (`_prepared_for`), and entries the next step does not consume
        # are dropped there
        self._prepared[id(data)] = (prep, ev, data, pts)

    READY_KEPT = 64

    def inputs_ready(self, *batches):
        """Declare that the tensors of ``batches`` are (or will be, in stream order) complete on the current stream NOW:
        one event, which ``prefetch`` of such a batch waits for instead of the whole main stream. Call it once after the
        uploads of the batches a loop cycles through (``run`` does), or after every batch's upload (``train_epochs``). The
        announcement is for the tensors as they are now: after writing new points INTO an announced batch, announce it
        again. Only the ``READY_KEPT`` most recent announcements are remembered (an epoch of fresh batches must not pin
        them all); a forgotten batch is still correct - its prefetch waits for the whole stream."""
        if self.device.type != 'cuda' or os.environ.get('GGA_INPUTS_READY') == '0':      # (0: A/B switch - every prefetch waits for the stream)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        for b in batches:
            self._ready.pop(id(b), None)
            self._ready[id(b)] = (ev, b)
        while len(self._ready) > max(self.READY_KEPT, len(batches)):
            self._ready.popitem(last=False)

    def _prepared_for(self, data):
        hit = self._prepared.get(id(data))
        if hit is not None and hit[2] is data and hit[3] is data.get('points'):
            return hit
        return None

    def _call_train_step(self, data):
        if self.model is self.raw_model:
            return self.raw_model.train_step(data)
        losses = self.model(**data)                    # DDP forward
        loss, log_vars = self.raw_model._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    def step(self, data, next_data=None):
        from . import dense_conv
        outer, dense_conv.PLANES = dense_conv.PLANES, self.planes
        try:
            return self._step(data, next_data)
        finally:
            dense_conv.PLANES = outer

    # How far the host may run ahead of the device: at the start of a step at most this many earlier steps are still queued
    # or running. The host side of a step is 12-15 ms (PointPillars) / 26-29 ms (sparse trunk) against 33 / 52 ms on the device,
    # and nothing in a step waits for the main stream - unbounded, the host gets 4-5 steps ahead (until the runtime's queue
    # stops it), and everything a queued step owns stays allocated meanwhile: on the sparse trunk 1.6 GB of prefetched index
    # structures per step (measured, round 5: 9.6 GB active with the lead unbounded, 8.0 GB with three steps). Three steps of
    # lead keep the device's queue full (step times equal within noise: 33.8 / 54.0 against 34.0 / 54.5 ms); the wait is a
    # blocking event wait (no spinning core). 0: unbounded.
    MAX_STEPS_AHEAD = int(os.environ.get('GGA_MAX_STEPS_AHEAD', '3'))

    def _bound_lead(self):
        if self.device.type != 'cuda' or self.MAX_STEPS_AHEAD <= 0:
            return
        while len(self._in_flight) > self.MAX_STEPS_AHEAD:
            self._in_flight.popleft().synchronize()

    def _step(self, data, next_data):
        from . import dense_conv, functional as F
        self._bound_lead()
        dense_conv.AMAX_POOL.next_generation()      # one memset for all of this step's absmax slots
        guarded = (self.planes == 2 and self.range_check_interval > 0 and self.device.type == 'cuda'
                   and (self._guard_next or self.iter % self.range_check_interval == 0))
        self._guard_next = False
        if guarded:
            dense_conv.RANGE_GUARD.arm()
        hit = self._prepared_for(data)
        stale = [(p, e) for k, (p, e, d, _) in self._prepared.items() if d is not data]
        self._prepared.clear()                      # a prefetched batch that is not stepped next is dropped, not kept
        if next_data is not None and PREFETCH_FIRST:
            # the NEXT batch's point-only front goes to the device BEFORE this step's kernels are queued: its few small kernels
            # then run at once and the host's reads of their counts return in a millisecond. Queued after the step (as in
            # rounds 1-2) they sat behind the main stream's backlog and the host idled in those reads for most of the step
            # (tools_dev/host_profile.py second: 41 of 65 ms), leaving the device without work at the start of the next one.
            self.prefetch(next_data)
        self._retired = [(p, e) for p, e in self._retired + stale if not e.query()]
        prep = None
        if hit is not None:
            prep, ev = hit[:2]
            torch.cuda.current_stream(self.device).wait_event(ev)
            data = dict(data, points=prep)
        for g in self.optimizer.param_groups:
            if isinstance(self.lr_sched, StepSchedule):
                g['lr'] = g['initial_lr'] * self.lr_sched(self.iter)
            elif self.lr_sched is not None:
                g['lr'] = self.lr_sched(self.iter)
            if self.mom_sched is not None:
                g['betas'] = (self.mom_sched(self.iter), g['betas'][1])
        with F.deferred_batch_counters():                 # one multi-tensor add for the BatchNorm counters of the pass
            out = self._call_train_step(data)
        self.optimizer.zero_grad(set_to_none=True)
        dense_conv.RANGE_GUARD.phase = 'backward'
        out['loss'].backward()
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_([p for p in self.raw_model.parameters() if p.grad is not None],
                                           **self.grad_clip)
        self.optimizer.step()
        if guarded:
            self._check_range()
        self.iter += 1
        if self.device.type == 'cuda' and self.MAX_STEPS_AHEAD > 0:
            queued = torch.cuda.Event(blocking=True)
            queued.record(torch.cuda.current_stream(self.device))
            self._in_flight.append(queued)
        if prep is not None:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            self._retired.append((prep, done))
        if next_data is not None and not PREFETCH_FIRST:
            self.prefetch(next_data)
        return out

    def _check_range(self):
        """End of a guarded step: one host read of the guard's rows; falls back to three bf16 planes when an operand has
        is over the limit of its pass (``dense_conv.RangeGuard``: forward - lost elements, backward - lost mass)."""
        import warnings
        from . import dense_conv
        rows = dense_conv.RANGE_GUARD.disarm()
        fwd = [r for r in rows if r['phase'] == 'forward']
        bwd = [r for r in rows if r['phase'] != 'forward']
        over = [r for r in rows if r['over']]
        self.range_reports.append(dict(
            iter=self.iter, operands=len(rows), over_limit=len(over),
            forward_worst_share_lost=max((r['share_lost'] for r in fwd), default=0.0),
            backward_worst_mass_lost=max((r['mass_lost'] for r in bwd), default=0.0),
            backward_worst_share_lost=max((r['share_lost'] for r in bwd), default=0.0),
            worst_share_below_2p17=max((r['share_below_2p17'] for r in rows), default=0.0),
            first_over=dict(over[0]) if over else None))
        # data parallel: the ranks switch together (one rank on three planes beside seven on two would be a straggler at
        # every all-reduce and a run nobody can reproduce) - a MAX over the ranks of the one-element flag, in the guarded
        # steps only (all ranks guard the same iterations)
        fall_back = bool(over)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            flag = torch.tensor([float(fall_back)], device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            fall_back = bool(flag.item())
        self.range_reports[-1]['fell_back'] = bool(fall_back and not dense_conv.PLANES_PINNED)
        if fall_back and not dense_conv.PLANES_PINNED:
            self.planes = 3
            self.fell_back = dense_conv.FELL_BACK = True
            w = over[0] if over else None
            where = (f'a {w["phase"]} operand {w["shape"]} of the convolutions has {w["share_lost"]:.2%} of its non-zero elements '
                     f'({w["mass_lost"]:.1e} of its L1 mass) below 2^-30 of its largest magnitude') if w else 'another rank reported an operand over the limit'
            warnings.warn(f'iteration {self.iter}: {where} - continuing on three bf16 planes (fp32 exponent range) instead of two '
                          f'fp16 planes from the next step on (the update of this step, computed on two planes, is kept)')

    def freeze_gc(self):
        """Garbage-collector policy of the iteration loop: after the first iterations have built every long-lived object
        (modules, rule books, workspaces) they are moved to the permanent generation (``gc.freeze``), so the generational
        passes that still run during training only look at the few objects of the current step - a full pass over the
        module tree takes tens of ms and stalls the launch queue. Called by ``run`` after its warm-up iterations and by
        bench.py at the same point: the timed loop and the product loop are the same loop."""
        import gc
        if not self._gc_frozen:
            gc.collect()
            gc.freeze()
            if os.environ.get('GGA_GC_OFF') == '1':       # A/B switch for measurements: no collector at all
                gc.disable()
            self._gc_frozen = True

    GC_FREEZE_AFTER = 3

    def run(self, batches, n_iters, logger=None):
        self.raw_model.train()
        t0 = time.time()
        out = None
        self.inputs_ready(*batches)
        for i in range(n_iters):
            if i == self.GC_FREEZE_AFTER:
                self.freeze_gc()
            out = self.step(batches[i % len(batches)], next_data=batches[(i + 1) % len(batches)])
            if logger and (i + 1) % self.log_interval == 0:
                vals = {k: float(v) for k, v in out['log_vars'].items()}      # the only sync, every N iters
                logger(f'iter {i + 1}/{n_iters} lr {self.optimizer.param_groups[0]["lr"]:.3e} '
                       f'time {(time.time() - t0) / (i + 1):.3f}s  ' + ' '.join(f'{k}={v:.4f}' for k, v in vals.items()))
        return out

    # ---- checkpoints: mmcv's file layout (runner/checkpoint.py ``save_checkpoint`` / ``load_checkpoint``, EpochBasedRunner
    # ``save_checkpoint`` / ``resume``; third-party, restated - parity unpinned): {'meta': {'epoch', 'iter', ...},
    # 'state_dict': CPU tensors under the bare module's names, 'optimizer': optimizer.state_dict()}. What the reference's
    # train entry wires up with ``checkpoint_config`` / ``resume_from`` / ``load_from`` (apis/train.py:275-281,318-322).
    def save_checkpoint(self, out_dir, filename_tmpl='epoch_{}.pth', save_optimizer=True, meta=None, create_symlink=True):
        """``out_dir/epoch_{finished epochs}.pth`` (+ ``latest.pth`` pointing at it); called after an epoch has been
        counted (``train_epochs``) or at any iteration."""
        from collections import OrderedDict
        meta = dict(meta or {})
        # the arithmetic THIS run trains on (Runner.planes; the process-wide dense_conv.PLANES is the library default outside
        # `step`) and whether its range guard has fallen back: resume() restores both
        meta.update(epoch=self.epoch, iter=self.iter, time=time.asctime(), gga_amd_planes=int(self.planes),
                    gga_amd_fell_back=bool(self.fell_back))
        classes = getattr(self.raw_model, 'CLASSES', None)
        if classes is not None:
            meta['CLASSES'] = classes
        ckpt = dict(meta=meta, state_dict=OrderedDict((k, v.detach().cpu()) for k, v in self.raw_model.state_dict().items()))
        if save_optimizer:
            ckpt['optimizer'] = self.optimizer.state_dict()
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, filename_tmpl.format(self.epoch))
        tmp = path + '.tmp'
        torch.save(ckpt, tmp)
        os.replace(tmp, path)                     # a reader never sees a half-written file
        if create_symlink:
            link = os.path.join(out_dir, 'latest.pth')
            if os.path.lexists(link):
                os.remove(link)
            try:
                os.symlink(os.path.basename(path), link)
            except OSError:                       # file systems without symlinks: a copy, as mmcv does
                import shutil
                shutil.copy(path, link)
        return path

    def load_checkpoint(self, filename, map_location='cpu', strict=False):
        """Weights only (``load_from``): the file's ``state_dict`` (a bare state dict is accepted too), a leading
        ``module.`` of a wrapped model's names dropped. -> the loaded file."""
        ckpt = torch.load(filename, map_location=map_location, weights_only=False)
        state = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
        state = {(k[7:] if k.startswith('module.') else k): v for k, v in state.items()}
        missing, unexpected = self.raw_model.load_state_dict(state, strict=strict)
        if missing or unexpected:
            import warnings
            warnings.warn(f'{filename}: missing keys {list(missing)[:8]}, unexpected keys {list(unexpected)[:8]}')
        return ckpt

    def resume(self, checkpoint, resume_optimizer=True, map_location='default'):
        """Continue a run (``resume_from``): weights and buffers, epoch and iteration counters (the schedules are functions
        of the iteration), optimizer state. The next step is the one the saved run would have done next."""
        if map_location == 'default':
            map_location = self.device if self.device.type == 'cuda' else 'cpu'
        ckpt = self.load_checkpoint(checkpoint, map_location=map_location, strict=True)
        self.epoch, self.iter = int(ckpt['meta']['epoch']), int(ckpt['meta']['iter'])
        if resume_optimizer and 'optimizer' in ckpt:
            self.optimizer.load_state_dict(ckpt['optimizer'])
            for g in self.optimizer.param_groups:          # checkpoints written before 'initial_lr' rode along
                g.setdefault('initial_lr', g['lr'])
        # a run whose range guard had fallen back to three planes continues on three planes (its operands are known to exceed
        # the two-plane range); any resumed run is guarded again on its first step
        from . import dense_conv
        if ckpt['meta'].get('gga_amd_fell_back') and not dense_conv.PLANES_PINNED:
            self.planes = 3
            self.fell_back = dense_conv.FELL_BACK = True
        self._guard_next = True
        return ckpt['meta']

    def train_epochs(self, data_loader, max_epochs, work_dir=None, checkpoint_config=None, logger=None, to_inputs=None,
                     after_iter=None):
        """The epoch loop of the reference's ``EpochBasedRunner.run`` with the hooks the GGA configs register: sampler
        re-seeded per epoch (``DistSamplerSeedHook``), one ``step`` per loaded batch - the next batch is fetched and its
        points uploaded while this one is stepped, so its point-only front overlaps the step (``prefetch``) -, a
        checkpoint every ``checkpoint_config.interval`` epochs and after the last one (``CheckpointHook``).

        ``loop_seconds`` accumulates where the loop's host time goes: ``fetch`` (waiting in ``next(loader)``: the data side
        is late), ``inputs`` (unpacking + the points' upload), ``step`` (queueing the step); ``after_iter(runner, n)`` is
        called after every iteration (bench.py resets / reads the counters there)."""
        from .loader import PointUploader, to_step_inputs
        if to_inputs is None:
            on_gpu = self.device.type == 'cuda'
            uploader = PointUploader(self.device) if on_gpu and os.environ.get('GGA_POINT_UPLOADER', '1') == '1' else None
            to_inputs = lambda b: to_step_inputs(b, self.device if on_gpu else None, uploader=uploader)
        else:
            uploader = None
        clock = time.perf_counter
        acc = self.loop_seconds = dict(fetch=0.0, inputs=0.0, step=0.0, iters=0)

        def fetch(it):
            t = clock()
            raw = next(it, None)
            t1 = clock()
            acc['fetch'] += t1 - t
            out = to_inputs(raw) if raw is not None else None
            acc['inputs'] += clock() - t1
            return out
        ck = dict(checkpoint_config or {})
        interval, save_last = int(ck.get('interval', -1)), ck.get('save_last', True)
        out_dir = ck.get('out_dir') or work_dir
        keep = int(ck.get('max_keep_ckpts', -1))
        self.raw_model.train()
        out = None
        while self.epoch < max_epochs:
            sampler = getattr(data_loader, 'sampler', None)
            if hasattr(sampler, 'set_epoch'):
                sampler.set_epoch(self.epoch)
            it = iter(data_loader)
            nxt = fetch(it)
            cur = None
            if nxt is not None:
                self.inputs_ready(nxt)
            t0, n = time.time(), 0
            while nxt is not None:
                cur, nxt = nxt, fetch(it)
                if nxt is not None:
                    self.inputs_ready(nxt)
                if self.iter == self.GC_FREEZE_AFTER:
                    self.freeze_gc()
                t = clock()
                out = self.step(cur, next_data=nxt)
                if uploader is not None:
                    uploader.step_done()                  # the slot of `cur` has its last reader: the next upload into it waits for this step
                acc['step'] += clock() - t
                acc['iters'] += 1
                n += 1
                if after_iter is not None:
                    after_iter(self, n)
                if logger and self.iter % self.log_interval == 0:
                    vals = {k: float(v) for k, v in out['log_vars'].items()}
                    logger(f'epoch {self.epoch + 1} iter {n}/{len(data_loader)} lr {self.optimizer.param_groups[0]["lr"]:.3e} '
                           f'time {(time.time() - t0) / n:.3f}s  ' + ' '.join(f'{k}={v:.4f}' for k, v in vals.items()))
            self.epoch += 1
            last = self.epoch == max_epochs
            if out_dir and ((interval > 0 and self.epoch % interval == 0) or (save_last and last)) and _is_rank0():
                self.save_checkpoint(out_dir, save_optimizer=ck.get('save_optimizer', True))
                if keep > 0:
                    for e in range(self.epoch - keep * max(interval, 1), 0, -max(interval, 1)):
                        old = os.path.join(out_dir, f'epoch_{e}.pth')
                        if os.path.exists(old):
                            os.remove(old)
        return out



def _is_rank0():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def find_latest_checkpoint(path, suffix='pth'):
    """``work_dir/latest.pth`` if present, else the ``epoch_N`` / ``iter_N`` file with the largest N (mmdet ``find_latest_checkpoint``)."""
    import glob
    import re
    if not os.path.isdir(path):
        return None
    latest = os.path.join(path, f'latest.{suffix}')
    if os.path.exists(latest):
        return latest
    best, best_n = None, -1
    for f in glob.glob(os.path.join(path, f'*.{suffix}')):
        m = re.search(r'_(\d+)\.' + suffix + '$', os.path.basename(f))
        if m and int(m.group(1)) > best_n:
            best, best_n = f, int(m.group(1))
    return best


def train_detector(model, dataset, cfg, distributed=False, validate=False, timestamp=None, meta=None, logger=None, device=None,
                   after_iter=None):
    """``mmdet3d/apis/train.py:180-322`` for the GGA configs: loaders from ``cfg.data`` (``samples_per_gpu`` /
    ``workers_per_gpu``, sharded over the ranks when ``distributed``, seeded with ``cfg.seed``), the DDP wrap, optimizer /
    clipping / cyclic schedules (``Runner``), ``checkpoint_config``, ``resume_from`` / ``auto_resume`` / ``load_from``, then
    ``cfg.workflow``'s train epochs. ``validate`` (the KITTI AP evaluation hook) is outside the hot path: refused loudly.
    -> the ``Runner`` after training."""
    from .loader import build_dataloader
    if validate:
        raise NotImplementedError('evaluation hooks (KITTI AP) are out of scope: run with validate=False')
    dataset = dataset[0] if isinstance(dataset, (list, tuple)) else dataset
    data = cfg.data
    if 'imgs_per_gpu' in data:                    # mmdet < 2.0 spelling
        data['samples_per_gpu'] = data['imgs_per_gpu']
    runner_cfg = cfg.get('runner') or dict(type='EpochBasedRunner', max_epochs=cfg.total_epochs)
    if runner_cfg['type'] != 'EpochBasedRunner':
        raise NotImplementedError('configs/gga train with the EpochBasedRunner')
    workflow = cfg.get('workflow', [('train', 1)])
    assert all(mode == 'train' for mode, _ in workflow), 'val epochs in the workflow are not supported'
    gpu_ids = cfg.get('gpu_ids', [0])
    loader = build_dataloader(dataset, data['samples_per_gpu'], data['workers_per_gpu'], num_gpus=len(gpu_ids), dist=distributed,
                              seed=cfg.get('seed'), worker_seed=cfg.get('worker_seed'), runner_type=runner_cfg['type'],
                              persistent_workers=data.get('persistent_workers', False))
    max_epochs = int(runner_cfg['max_epochs'])
    runner = Runner(model, cfg, max_iters=max_epochs * len(loader), distributed=distributed, device=device,
                    iters_per_epoch=len(loader))
    work_dir = cfg.get('work_dir')
    resume_from = cfg.get('resume_from')
    if resume_from is None and cfg.get('auto_resume') and work_dir:
        resume_from = find_latest_checkpoint(work_dir)
    if resume_from:
        runner.resume(resume_from)
    elif cfg.get('load_from'):
        runner.load_checkpoint(cfg.load_from)
    runner.train_epochs(loader, max_epochs, work_dir=work_dir, checkpoint_config=cfg.get('checkpoint_config'), logger=logger,
                        after_iter=after_iter)
    return runner
