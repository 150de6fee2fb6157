"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference's only parallelism is single-process ``nn.DataParallel`` (``models/models.py:21-22``): every
step it re-broadcasts all 183 M generator parameters from GPU 0 and reduce-adds the gradients back onto it.
Here every rank owns a full replica + its own Adam state and the ONLY exchange is an all-reduce(avg) of the
flat gradient arenas -- mathematically the reference's mean over per-replica mean losses for equal shards.

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is bound by ONE link:
G (730 MB) ~ 8.4 ms, D (34 MB) ~ 0.4 ms.  Buckets are contiguous slices of the gradient arena (no copies),
cut in reverse parameter order and launched on a side HIP stream the moment the last wgrad kernel of the
bucket has been enqueued, so the exchange hides under the rest of backward.  Bucket size 64 MB keeps each
collective >> the ~20 us launch latency while leaving >= 10 buckets of G to pipeline.
"""
import os

import torch
import torch.distributed as dist


def init_process_group_from_env(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 0, 1
    rank, local = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    if backend is None:
        backend = os.environ.get('HIM_DDP_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend != 'nccl' and torch.cuda.is_available():
        # logic tests: several gloo ranks may share one GPU (RCCL refuses that); models read LOCAL_RANK
        local = local % torch.cuda.device_count()
        os.environ['LOCAL_RANK'] = str(local)
    if backend == 'nccl':
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def timed_wait(waiter, other, sink=None):
    """``waiter.wait_stream(other)``; with a ``sink`` list, bracketed by two timing events on ``waiter`` whose distance
    is the time that stream sat idle for ``other`` (nothing else lies between them): the EXPOSED part of whatever
    ``other`` was doing (bench.py's ``exposed_comm_ms``)."""
    if sink is None:
        waiter.wait_stream(other)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(waiter)
    waiter.wait_stream(other)
    e1.record(waiter)
    sink.append((e0, e1))


class GradReducer(object):
    """Bucketed all-reduce(avg) of a flat gradient buffer.

    ``flat``    1-D tensor holding all gradients (``FlatArena.grad`` or any CPU/GPU tensor).
    ``ranges``  list of (start, end) element ranges, one per parameter, in FORWARD order.
    """

    def __init__(self, flat, ranges, bucket_bytes=64 << 20, group=None, force=False, comm_stream=None, fake=False,
                 local=False, tail_bytes=8 << 20):
        """``comm_stream``: the stream the collectives are enqueued on (default: a stream of this reducer's own).  The
        trainers pass their optimizer streams -- idle during the backward pass, and the optimizer step that follows the
        exchange runs there anyway -- so that N > 1 ranks run the SAME number of streams / hardware queues as one rank
        (the step sits at the hardware-queue cliff documented in DESIGN.md: one more queue costs 10 ms per step).
        ``fake``: single-GPU stand-in for the exchange (bench.py --fake-comm): every bucket is a device-to-device copy of
        its bytes on the comm stream at the real trigger point -- the scheduling and HBM cost of the exchange without a
        second GPU; gradients are left untouched.
        ``local``: no exchange at all (one rank) -- the reducer only tracks which gradient buckets are final and calls
        ``bucket_hook(b, start, end)`` on the comm stream behind them (the trainers hang the bucket's Adam step there).
        ``tail_bytes`` (round 5): the LAST bucket to become final -- the first parameters of the network, whose weight
        gradients close the backward pass -- is the one collective nothing can hide (the optimizer step waits for it, the
        next generator forward for the optimizer step).  Cut in reverse order, that bucket is whatever is left over, up to
        ``bucket_bytes`` (58 MB of the generator's 730 MB at C2: 0.66 ms on one 153 GB/s ring link).  It is split on a
        parameter boundary so that the final piece holds at most ``tail_bytes`` (C2: stem + the first three down-convolutions,
        6.7 MB = 0.08 ms); the rest of it becomes final ~3 ms earlier."""
        self.flat, self.group = flat, group
        self.fake = bool(fake)
        self.local = bool(local)
        self.bucket_hook = None      # called inside _launch, on the comm stream, behind the bucket's exchange
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.fake or self.local or self.world > 1 or (force and dist.is_initialized())   # force: exercise the path on 1 rank
        self.ranges = list(ranges)
        self.range_to_bucket = {}
        self.buckets = []            # (start, end, n_params) ; bucket 0 = LAST parameters (first ready in backward)
        cap = max(bucket_bytes // 4, 1)
        end = n = 0
        start = None
        for (s, e) in reversed(self.ranges):
            if start is None:
                end, start, n = e, s, 0
            start = s
            n += 1
            self.range_to_bucket[(s, e)] = len(self.buckets)
            if end - start >= cap:
                self.buckets.append((start, end, n))
                start = None
        if start is not None:
            self.buckets.append((start, end, n))
        if tail_bytes and self.buckets and (self.buckets[-1][1] - self.buckets[-1][0]) * 4 > tail_bytes:
            s0, e0, _ = self.buckets[-1]
            inside = [r for r in self.ranges if s0 <= r[0] and r[1] <= e0]          # forward order
            cut = next((i for i, r in enumerate(inside) if (r[1] - s0) * 4 > tail_bytes), len(inside))
            if 0 < cut < len(inside):
                mid = inside[cut][0]
                self.buckets[-1] = (mid, e0, len(inside) - cut)
                self.buckets.append((s0, mid, cut))
                for r in inside[:cut]:
                    self.range_to_bucket[r] = len(self.buckets) - 1
        self.on_gpu = flat.is_cuda
        self.comm_stream = (comm_stream or torch.cuda.Stream(device=flat.device)) if self.on_gpu else None
        self.scratch = torch.empty(min(cap, flat.numel()) + 64, dtype=flat.dtype, device=flat.device) if self.fake else None
        self.pending = [0] * len(self.buckets)
        self.launched = [True] * len(self.buckets)
        self.works = []
        self.armed = False
        self.timing = None           # list of (event, event) pairs while a caller measures the exposed exchange

    def attach(self, params):
        """Bind parameters (carrying ``_him_arena_range``) so the wgrad kernels' completion triggers buckets."""
        self.params = list(params)
        for p in self.params:
            p._him_reducer = self

    def begin(self, contributions=1):
        """``contributions``: how many wgrad writes each parameter receives in the coming backward (the
        discriminator is run twice with live weights inside loss_D)."""
        if not self.active:
            return
        # parameters flagged ``_him_dead_grad`` (conv biases in front of a mean-subtracting norm, nn.run_layers) never
        # receive a weight-gradient launch: they must not hold their bucket back
        live = [0] * len(self.buckets)
        for p in getattr(self, 'params', ()):
            if not getattr(p, '_him_dead_grad', False):
                live[self.range_to_bucket[tuple(p._him_arena_range)]] += 1
        if not getattr(self, 'params', None):
            live = [b[2] for b in self.buckets]
        self.pending = [n * contributions for n in live]
        self.launched = [False] * len(self.buckets)
        self.works = []
        self.armed = True

    def on_param(self, p):
        if not self.armed:
            return
        b = self.range_to_bucket.get(tuple(p._him_arena_range))
        if b is None:
            return
        self.pending[b] -= 1
        if self.pending[b] == 0 and not self.launched[b]:
            self._launch(b)

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        view = self.flat[s:e]
        self.launched[b] = True
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            # a bucket can hold gradients written on two streams (conv weight gradients on the side stream, BatchNorm
            # gamma / beta on the main one): the exchange waits for the triggering stream AND the side stream
            from .ops import wgrad_waitables
            streams, events = wgrad_waitables(self.flat.device)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                for side in streams:
                    self.comm_stream.wait_stream(side)
                for done in events:
                    self.comm_stream.wait_event(done)
                if self.fake:                                    # stand-in: move the bucket's bytes once, on the comm stream
                    n = e - s
                    for o in range(0, n, self.scratch.numel()):
                        m = min(self.scratch.numel(), n - o)
                        self.scratch[:m].copy_(view[o:o + m])
                elif self.local or self.world == 1 and not dist.is_initialized():
                    pass                                         # one rank: the gradients are final as they are
                elif dist.get_backend(self.group) == 'nccl':     # RCCL: averaging collective
                    dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group)
                else:                                            # gloo on device tensors (tests): no AVG op
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                    view.div_(self.world)
                if self.bucket_hook is not None:
                    self.bucket_hook(b, s, e)
        else:
            if not self.local:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                view.div_(self.world)
            if self.bucket_hook is not None:
                self.bucket_hook(b, s, e)

    def finish(self):
        """Launch whatever has not been triggered and make the current stream wait for the exchange."""
        if not self.active or not self.armed:
            return
        if self.on_gpu:
            from .ops import join_side_stream
            join_side_stream(self.flat.device)   # side-stream weight gradients of never-triggered buckets
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        if self.on_gpu:
            timed_wait(torch.cuda.current_stream(), self.comm_stream, self.timing)
        self.armed = False


def broadcast_replica_state(model, src=0):
    """Every rank adopts rank ``src``'s parameters, Adam state and module buffers (BatchNorm running statistics,
    spectral-norm ``u``).  The reference's ``nn.DataParallel`` re-broadcasts the module from GPU 0 on EVERY forward
    (``models/models.py:21-22``); with one replica per process the broadcast is needed exactly once -- afterwards the
    replicas receive identical averaged gradients and identical Adam updates.  Returns the number of floats sent."""
    from .ops import invalidate_panels
    sent = 0
    for tag in ('G', 'D'):
        opt = getattr(model, 'optimizer_' + tag, None)
        if opt is None:
            continue
        for flat in (opt.arena.data, opt.exp_avg, opt.exp_avg_sq):
            dist.broadcast(flat, src=src)
            sent += flat.numel()
        step = torch.tensor([opt.step_count], dtype=torch.int64, device=opt.arena.data.device)
        dist.broadcast(step, src=src)
        opt.step_count = int(step.item())
        invalidate_panels(opt.arena.params)      # cached weight panels were built from the pre-broadcast values
    for name in ('netG', 'netD'):
        net = getattr(model, name, None)
        if net is None:
            continue
        for buf in net.buffers():
            if buf.is_floating_point():
                dist.broadcast(buf, src=src)
                sent += buf.numel()
    return sent


def replica_checksum_equal(model):
    """True when every rank holds bit-identical parameters (sum of the raw int32 views of both arenas)."""
    sums = []
    for tag in ('G', 'D'):
        opt = getattr(model, 'optimizer_' + tag, None)
        if opt is not None:
            sums.append(opt.arena.data.view(torch.int32).to(torch.int64).sum())
    mine = torch.stack(sums)
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


def attach_data_parallel(model, bucket_bytes=64 << 20, force=False, broadcast=True, fake=False, tail_bytes=8 << 20):
    """Give a mask2image / box2mask model per-network reducers (no-op for world size 1 unless ``force``).
    Rank 0's parameters / Adam state / buffers are broadcast first (see ``broadcast_replica_state``).
    BatchNorm layers keep per-rank batch statistics (the reference's DataParallel behaviour); only gradients are averaged."""
    if not fake and (not dist.is_initialized() or (dist.get_world_size() <= 1 and not force)):
        return model
    if broadcast and dist.is_initialized() and dist.get_world_size() > 1:
        broadcast_replica_state(model)
    from . import ops
    for tag in ('G', 'D'):
        opt = getattr(model, 'optimizer_' + tag, None)   # box2mask trainer: optimizer_G is its ``optimizer``
        if opt is None:
            continue
        arena = opt.arena
        # the collectives ride on the network's optimizer stream (see GradReducer): no extra stream for N > 1
        dev = arena.grad.device
        comm = (ops._opt_stream(dev) if tag == 'G' else ops._d_opt_stream(dev)) if arena.grad.is_cuda else None
        red = GradReducer(arena.grad, [p._him_arena_range for p in arena.params], bucket_bytes, force=force,
                          comm_stream=comm, fake=fake, tail_bytes=tail_bytes)
        red.attach(arena.params)
        red.bucket_hook = getattr(model, '_bucket_update_' + tag, None)    # the bucket's Adam step behind its exchange
        setattr(model, 'reducer_' + tag, red)
    return model
