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
        params = list(params)
        self.arena = arena if arena is not None else FlatArena(params)
        self.param_groups = [dict(params=self.arena.params, lr=lr, betas=tuple(betas), eps=eps)]
        self.exp_avg = torch.zeros_like(self.arena.data)
        self.exp_avg_sq = torch.zeros_like(self.arena.data)
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def step(self):
        g = self.param_groups[0]
        self.step_count += 1
        a = self.arena
        from .ops import join_side_stream
        join_side_stream(a.grad.device)        # wait for the side-stream weight gradients
        lib.him_adam_step(a.data.data_ptr(), a.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                          a.total, float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                          self.step_count, _stream())
        from .ops import refresh_panels
        refresh_panels(a.params)               # regrouped weight panels of the conv kernels follow the update

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
        return dict(step=self.step_count, exp_avg=self.exp_avg.cpu(), exp_avg_sq=self.exp_avg_sq.cpu(),
                    param_groups=[{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups])

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update(s)
