"""Flat parameter/gradient arenas and the fused Adam that walks them.

MI355X-first layout: all parameters of one network live in ONE contiguous fp32 buffer (and so do their
gradients, and Adam's two moments).  Consequences:
  * ``optimizer.step()`` is a single HBM-bound kernel over the arena (16 B/param read, 12 B/param written);
  * ``zero_grad()`` is one fill;
  * the data-parallel all-reduce operates on contiguous slices of the gradient arena -- bucket boundaries
    are just offsets, nothing is ever copied in or out of a bucket (see ``dist.py``).
Parameters stay ordinary ``nn.Parameter`` objects (views into the arena), so ``state_dict()`` keys and
tensors are exactly the reference's (``models/base_model.py:46-51`` checkpoint format).
"""
import bisect

import torch

from ._cabi import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class FlatArena(object):
    def __init__(self, params):
        self.params = [p for p in params]
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        # 64-float (256 B) alignment per tensor keeps every view 16 B aligned for float4 streams
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 63) // 64 * 64
        self.total = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.data[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            p._him_direct_grad = True
            p._him_arena_range = (o, o + n)

    def rebind(self):
        """Re-attach .grad views (something set them to None) -- keeps the direct-wgrad contract."""
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + n].view(p.shape)

    def zero_grad(self):
        self.rebind()
        from .ops import join_side_stream
        join_side_stream(self.grad.device)     # no weight-gradient kernel may still be writing
        lib.him_fill(self.grad.data_ptr(), self.total, 0.0, _stream())


class FusedAdam(object):
    """torch.optim.Adam look-alike (``zero_grad/step/param_groups/state_dict``) over a FlatArena.
    Numerics follow torch's single-tensor Adam: lerp first moment, sqrt(v)/sqrt(bc2)+eps, lr/bc1."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8, arena=None):
        """``params``: an iterable of Parameters, or torch-style groups ``[{'params': [...], 'lr': ...}, ...]`` (the
        reference freezes the global generator that way, ``models/pix2pixHD_condImg_model.py:122-130``: one group per
        parameter with lr 0 outside the local enhancer).  All groups share ONE arena in the order given; ``step`` runs
        one kernel per maximal contiguous run of parameters with equal hyper-parameters (one launch in the common case)."""
        params = list(params)
        if params and isinstance(params[0], dict):
            groups = [dict(g, params=list(g['params'])) for g in params]
        else:
            groups = [dict(params=params)]
        flat = [p for g in groups for p in g['params']]
        self.arena = arena if arena is not None else FlatArena(flat)
        if [id(p) for p in self.arena.params] != [id(p) for p in flat]:
            raise ValueError('the arena holds a different parameter list than the optimizer was given')
        self.param_groups = []
        for g in groups:
            self.param_groups.append(dict(params=g['params'], lr=g.get('lr', lr), betas=tuple(g.get('betas', betas)),
                                          eps=g.get('eps', eps)))
        self.exp_avg = torch.zeros_like(self.arena.data)
        self.exp_avg_sq = torch.zeros_like(self.arena.data)
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def _runs(self):
        """[(start, end, lr, b1, b2, eps)] over the arena: contiguous parameters with equal hyper-parameters merged."""
        a = self.arena
        runs, i = [], 0
        for g in self.param_groups:
            n = len(g['params'])
            if n == 0:
                continue
            start = a.offsets[i]
            end = a.offsets[i + n] if i + n < len(a.offsets) else a.total
            hp = (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']))
            if runs and runs[-1][2:] == hp and runs[-1][1] == start:
                runs[-1] = (runs[-1][0], end) + hp
            else:
                runs.append((start, end) + hp)
            i += n
        return runs

    # ---- one optimizer step in pieces (bucket by bucket, as the gradients become final during the backward pass) ----
    def begin_step(self):
        """Open a step that ``step_range`` will carry out piecewise; ``step()`` closes it (whatever is left)."""
        if getattr(self, '_done', None):
            raise RuntimeError('a piecewise optimizer step is still open (aborted after part of the arena was updated): '
                               'step() completes it')
        self.step_count += 1
        self._done = []

    def abort_step(self):
        """An exception between ``begin_step`` and ``step``.  Nothing applied yet: the step is forgotten (count restored,
        the next ``step()`` is a whole step).  Part of the arena already updated: the step STAYS OPEN with its
        coverage -- the next ``step()`` completes the complement under the same step count, so no range is ever updated
        twice by one step and none is skipped; ``begin_step`` refuses to open another step on top of it."""
        done = getattr(self, '_done', None)
        if done is not None and not done:
            self.step_count -= 1
            self._done = None

    def _adam(self, lo, hi):
        a = self.arena
        for (s, e, lr, b1, b2, eps) in self._runs():
            s, e = max(s, lo), min(e, hi)
            if s < e:
                lib.him_adam_step(a.data.data_ptr() + 4 * s, a.grad.data_ptr() + 4 * s, self.exp_avg.data_ptr() + 4 * s,
                                  self.exp_avg_sq.data_ptr() + 4 * s, e - s, lr, b1, b2, eps, self.step_count, _stream())

    def _params_in(self, lo, hi):
        a = self.arena
        i = bisect.bisect_left(a.offsets, lo)
        out = []
        while i < len(a.offsets) and a.offsets[i] < hi:
            out.append(a.params[i])
            i += 1
        return out

    def step_range(self, lo, hi):
        """Adam update + panel rebuild of arena elements [lo, hi) on the current stream.  The caller guarantees that their
        gradients are final (and exchanged) and that nothing still reads these parameters (their layers' data gradients
        have run) -- the data-parallel reducer's bucket trigger (dist.GradReducer.bucket_hook)."""
        if getattr(self, '_done', None) is None:
            raise RuntimeError('step_range outside begin_step() ... step()')
        self._adam(lo, hi)
        self._done.append((lo, hi))
        from .ops import refresh_panels
        refresh_panels(self._params_in(lo, hi))

    def step(self):
        a = self.arena
        from .ops import join_side_stream, refresh_panels
        join_side_stream(a.grad.device)        # wait for the side-stream weight gradients
        done = getattr(self, '_done', None)
        if done is None:
            self.step_count += 1
            todo = [(0, a.total)]
        else:                                  # close a piecewise step: the complement of what step_range has covered
            todo, at = [], 0
            for lo, hi in sorted(done):
                if lo > at:
                    todo.append((at, lo))
                at = max(at, hi)
            if at < a.total:
                todo.append((at, a.total))
            self._done = None
        for lo, hi in todo:
            self._adam(lo, hi)
        # everything that reads the RAW parameters may start here; every cached weight panel carries its own event
        # (ops._panel waits for it), so a consumer need not wait for the whole rebuild pass below
        self.updated = torch.cuda.Event()
        self.updated.record(torch.cuda.current_stream(a.data.device))
        for lo, hi in todo:                    # regrouped weight panels of the conv kernels follow the update
            refresh_panels(self._params_in(lo, hi) if done is not None else a.params)

    def load_moments(self, exp_avgs, exp_avg_sqs, step):
        """Adopt per-parameter Adam moments (e.g. from a torch.optim.Adam) -- used by checkpoint import and by the
        teacher-forced parity tests."""
        a = self.arena
        for p, o, m, v in zip(a.params, a.offsets, exp_avgs, exp_avg_sqs):
            n = p.numel()
            self.exp_avg[o:o + n].copy_(m.reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(v.reshape(-1))
        self.step_count = int(step)

    def state_dict(self):
        """``torch.optim.Adam.state_dict()`` layout (what the reference's ``save_network_dict`` pickles,
        ``models/base_model.py:52-66``): per-parameter ``step / exp_avg / exp_avg_sq`` keyed by the parameter's index,
        plus ``param_groups`` with index lists -- loadable by a ``torch.optim.Adam`` over the same parameter list."""
        a = self.arena
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(a.params, a.offsets)):
                n = p.numel()
                state[i] = dict(step=torch.tensor(float(self.step_count)),
                                exp_avg=self.exp_avg[o:o + n].view(p.shape).cpu().clone(),
                                exp_avg_sq=self.exp_avg_sq[o:o + n].view(p.shape).cpu().clone())
        groups, i = [], 0
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != 'params'}
            for k, v in (('weight_decay', 0), ('amsgrad', False), ('maximize', False), ('foreach', None),
                         ('capturable', False), ('differentiable', False), ('fused', None)):
                d.setdefault(k, v)
            d['params'] = list(range(i, i + len(g['params'])))
            i += len(g['params'])
            groups.append(d)
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        """Accepts a ``torch.optim.Adam`` state dict (any torch era: ``step`` as int or tensor) or this class's round-1
        flat-arena format (``step / exp_avg / exp_avg_sq`` over the padded arena)."""
        if 'state' in sd:
            a = self.arena
            st = sd['state']
            if len(st) == 0:
                self.exp_avg.zero_()
                self.exp_avg_sq.zero_()
                self.step_count = 0
            else:
                # Parameter ORDER comes from the file's own param_groups: torch >= 1.6 keys the state by index, but the
                # reference's era (torch 0.3 / 0.4) keys it by id(p) -- memory addresses, also listed per group -- whose
                # numeric order says nothing about the parameter order.  A parameter without an entry (never received a
                # gradient) gets zero moments.  DEVIATION from torch, warned about below: torch would start such a
                # parameter at step 0 (bias corrections 1-b1, 1-b2 on its first update); this optimizer has ONE fused step
                # count, so its first update uses the corrections of step N+1 -- (1-b1)/sqrt(1-b2) times smaller.
                order = [pid for g in sd['param_groups'] for pid in g['params']]
                if len(order) != len(a.params):
                    raise ValueError('optimizer state lists %d parameters, the arena holds %d' % (len(order), len(a.params)))
                known = set(order)
                unknown = [k for k in st.keys() if k not in known]
                if unknown:
                    raise ValueError('optimizer state has entries no param_group lists: %s' % unknown[:4])
                ms, vs, steps = [], [], set()
                for pid, p in zip(order, a.params):
                    e = st.get(pid)
                    if e is None:
                        ms.append(torch.zeros(p.shape))
                        vs.append(torch.zeros(p.shape))
                        continue
                    if tuple(e['exp_avg'].shape) != tuple(p.shape):
                        raise ValueError('optimizer state %s: shape %s vs parameter %s'
                                         % (pid, tuple(e['exp_avg'].shape), tuple(p.shape)))
                    ms.append(e['exp_avg'])
                    vs.append(e['exp_avg_sq'])
                    steps.add(int(float(e['step'])))
                if len(steps) != 1:
                    raise ValueError('per-parameter step counts differ (%s): one fused step count only' % sorted(steps))
                missing = sum(1 for pid in order if pid not in st)
                if missing:
                    import warnings
                    warnings.warn('%d of %d parameters have no Adam state in the file: they start from zero moments at the '
                                  'shared step count %d (torch would restart them at step 0)'
                                  % (missing, len(order), next(iter(steps))))
                self.load_moments(ms, vs, steps.pop())
            for g, s in zip(self.param_groups, sd['param_groups']):
                for k in ('lr', 'betas', 'eps'):
                    if k in s:
                        g[k] = tuple(s[k]) if k == 'betas' else s[k]
                if s.get('weight_decay', 0) or s.get('amsgrad', False):
                    raise NotImplementedError('weight_decay / amsgrad are not used by the reference and not supported')
            return
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update(s)
