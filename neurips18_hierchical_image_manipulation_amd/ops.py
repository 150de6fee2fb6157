"""torch.autograd bindings of the HIP kernels (the only place the C ABI is called from).

PyTorch supplies device memory, the current HIP stream and the autograd tape; every arithmetic op of
the hot path below is a hand-written gfx950 kernel in libhim_hip.so.  Nothing here runs on the CPU and
nothing falls back: a non-CUDA tensor raises.

Weight gradients: parameters that live in a flat gradient arena (``optim.FlatArena``) carry
``_him_direct_grad = True``; their wgrad kernels accumulate straight into ``param.grad`` (which is a view
of the arena, zeroed by ``zero_grad``) and autograd receives ``None`` -- no per-parameter add pass and no
bucket copies for the RCCL all-reduce.
"""
import contextlib
import ctypes
import os
import threading

import torch

from ._cabi import (lib, HimAlgo, HimConv2d, HimDeconv2d, HimResBlock, ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID,
                    PAD_ZERO, PAD_REFLECT, HimError, ONEHOT_PART_IDS, ONEHOT_PART_DENSE)

ACTS = {'none': ACT_NONE, 'relu': ACT_RELU, 'lrelu': ACT_LRELU, 'tanh': ACT_TANH, 'sigmoid': ACT_SIGMOID}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise HimError('HIP op called with a non-GPU tensor: the product path has no CPU fallback')
        if t.dtype != torch.float32:
            raise HimError('HIP ops are fp32 only, got %s' % t.dtype)
        if not t.is_contiguous():
            raise HimError('HIP ops need contiguous NCHW tensors')


def _p(t):
    return 0 if t is None else t.data_ptr()


def _ws(nbytes, like):
    n = max((int(nbytes) + 3) // 4, 1)
    return torch.empty(n, dtype=torch.float32, device=like.device)


def _direct(p):
    return getattr(p, '_him_direct_grad', False) and p.grad is not None


# Weight gradients that go straight into a gradient arena are not needed by anything until the optimizer / the
# all-reduce, so they run on a SIDE stream next to the data-gradient chain: the two MFMA kernels of a layer fill each
# other's partial last waves (every launch here has an imperfect tile count for 256 CUs).  ``join_side_stream`` makes
# the current stream wait for them (called by FusedAdam.step / zero_grad and the reducer).
_SIDE = {}
from .config import SCHED


def _side_stream(device):
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device=device)
    return s


_OPT = {}


def _opt_stream(device):
    """Stream of the generator's optimizer step (runs next to the discriminator's backward)."""
    s = _OPT.get(device)
    if s is None:
        s = _OPT[device] = torch.cuda.Stream(device=device)
    return s


_VGGS = {}


def _vgg_stream(device):
    """Stream of the VGG(fake) forward / backward (next to the discriminator passes on the fake image)."""
    s = _VGGS.get(device)
    if s is None:
        s = _VGGS[device] = torch.cuda.Stream(device=device)
    return s


_REAL = {}


def _real_stream(device):
    """Stream of everything that depends only on the REAL image (its discriminator pass, its VGG features) and of that
    branch's backward.  Not the weight-gradient stream: the branch's data-gradient chain must not queue behind the
    generator's last weight gradients at the end of the step (r03b trace: 4 ms of idle main stream)."""
    s = _REAL.get(device)
    if s is None:
        s = _REAL[device] = torch.cuda.Stream(device=device)
    return s


_DOPT = {}


def _d_opt_stream(device):
    """Stream of the discriminator's gradient exchange + optimizer step in data-parallel runs (hidden under the NEXT
    step's input encoding and generator forward)."""
    s = _DOPT.get(device)
    if s is None:
        s = _DOPT[device] = torch.cuda.Stream(device=device)
    return s


# Weight-gradient routing: by default every weight gradient goes to the one side stream.  A trainer may route the weight
# gradients ISSUED FROM a given stream to another one for the duration of a backward pass (``route_wgrads``): during
# loss_D.backward() the fake-image branch (data gradients on the main stream) and the real-image branch (data gradients
# on the side stream, where its forward ran) then each get a weight-gradient stream of their own instead of queueing all
# of D's weight gradients behind the generator's on the single side stream.
_WGRAD_ROUTE = {}      # cuda_stream handle of the issuing stream -> torch.cuda.Stream that takes its weight gradients
# device -> {routed stream handle: event recorded behind the LAST weight-gradient launch routed there}.  Joins wait for
# the EVENT, not for the stream: a routed stream (the VGG stream) goes on to carry other work, and neither the optimizers
# nor the reducers should wait for that (an entry whose launches have finished costs a join nothing).
_WGRAD_USED = {}


class route_wgrads(object):
    """Route the weight gradients issued from the given streams; restores the previous routes on exit (nestable)."""

    def __init__(self, routes):
        self.routes = {src.cuda_stream: dst for src, dst in routes.items()}

    def __enter__(self):
        self.saved = {k: _WGRAD_ROUTE.get(k) for k in self.routes}
        _WGRAD_ROUTE.update(self.routes)

    def __exit__(self, *a):
        for k, prev in self.saved.items():
            if prev is None:
                _WGRAD_ROUTE.pop(k, None)
            else:
                _WGRAD_ROUTE[k] = prev
        return False


def wgrad_waitables(device):
    """What a consumer of ``device``'s weight gradients must wait for: the side stream, and the event behind the last
    launch on every stream weight gradients were routed to.  -> (streams, events)"""
    return ([s for dev, s in _SIDE.items() if dev == device], list(_WGRAD_USED.get(device, {}).values()))


def join_side_stream(device=None):
    for dev, s in _SIDE.items():
        if device is None or dev == device:
            torch.cuda.current_stream(dev).wait_stream(s)
    for dev, used in _WGRAD_USED.items():
        if device is None or dev == device:
            cur = torch.cuda.current_stream(dev)
            for ev in used.values():
                cur.wait_event(ev)


class _wgrad_stream(object):
    """Context: run the enclosed launches on the weight-gradient stream (the side stream, or the stream the issuing stream
    is routed to), after everything enqueued so far on the current one."""

    def __init__(self, *tensors, weight=None):
        self.tensors = [t for t in tensors if t is not None]
        self.weight = weight

    def __enter__(self):
        self.on = SCHED.wgrad_stream
        if not self.on:
            return None
        dev = self.tensors[0].device
        main = torch.cuda.current_stream(dev)
        side = _WGRAD_ROUTE.get(main.cuda_stream)
        alt = getattr(self.weight, '_him_wgrad_alt', None)
        if side is None and ((alt == 'tail' and SCHED.g_tail_wgrad_alt) or (alt == 'head' and SCHED.g_head_wgrad_alt)):
            # the LAST weight gradients of a chain-shaped generator's backward (its down-convolutions): on the VGG stream
            # (idle since VGG's backward) next to the ResnetBlock stack's weight-gradient GEMMs still queued on the
            # weight-gradient stream, instead of behind them (Pix2PixHDModel_condImg marks the weights)
            side = _vgg_stream(dev)
        self.routed = side is not None
        if side is None:
            side = _side_stream(dev)
        self.side, self.dev = side, dev
        side.wait_stream(main)
        for t in self.tensors:
            t.record_stream(side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return side

    def __exit__(self, *a):
        if self.on:
            if self.routed:
                ev = torch.cuda.Event()
                ev.record(self.side)
                _WGRAD_USED.setdefault(self.dev, {})[self.side.cuda_stream] = ev
            self.ctx.__exit__(*a)
        return False


# Run-time gradient routing for graphs that are shared between the generator loss and the discriminator loss (the
# fake-image discriminator pass is computed ONCE, see Pix2PixHDModel_condImg.forward): ids of weights whose weight
# gradient / whose layer-input gradient must not be computed during the current backward.
SKIP_WGRAD = set()
SKIP_DGRAD = set()


class _GradSwitch(torch.autograd.Function):
    """Identity whose backward passes the gradient only while ``state['open']`` is true."""

    @staticmethod
    def forward(ctx, x, state):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        ctx.state = state
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if g is None or not ctx.state['open']:
            return None, None
        # everything downstream of this node (the discriminator) has finished its part of this backward pass: a trainer
        # may hang work on that moment (Pix2PixHDModel_condImg: D's Adam step, whose weights nobody reads any more)
        cb = ctx.state.pop('on_open_backward', None)
        if cb is not None:
            cb()
        return g, None


def _wkey(w):
    """Routing key of a weight tensor: the id of the PARAMETER it was derived from (spectral-norm layers hand the conv a
    fresh W / sigma tensor every forward and tag it with ``_him_wkey``), else its own id."""
    return getattr(w, '_him_wkey', id(w))


def grad_switch(x, state):
    return _GradSwitch.apply(x, state)


def _notify(p):
    """Tell the data-parallel reducer (if any) that this parameter's gradient for the step is final."""
    r = getattr(p, '_him_reducer', None)
    if r is not None:
        r.on_param(p)


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
# Kernel selection (include/him.h "Algorithm selection"): the library keeps no global state -- every descriptor carries a
# HimAlgo.  This module keeps ONE current HimAlgo per host thread (``current_algo``): the process default is read from
# the HIM_* environment once at import through ``him_algo_from_env`` (tools/ A/B runs; an empty environment = the
# defaults), ``algo_scope`` / ``set_winograd_min_channels`` change the calling thread's copy.
_ALGO_TLS = threading.local()
_ALGO_DEFAULT = None


def _default_algo():
    global _ALGO_DEFAULT
    if _ALGO_DEFAULT is None:
        a = HimAlgo()
        lib.him_algo_from_env(ctypes.byref(a))
        _ALGO_DEFAULT = a
    return _ALGO_DEFAULT


def current_algo():
    """The calling thread's HimAlgo (a live object: descriptors COPY it when they are built)."""
    a = getattr(_ALGO_TLS, 'algo', None)
    if a is None:
        a = _ALGO_TLS.algo = HimAlgo.from_buffer_copy(_default_algo())
    return a


def resolved_algo(algo=None):
    """dict of the concrete values a HimAlgo selects (zeros replaced by the library's defaults) -- for reports."""
    out = HimAlgo()
    lib.him_algo_resolve(ctypes.byref(algo if algo is not None else current_algo()), ctypes.byref(out))
    return out.as_dict()


@contextlib.contextmanager
def algo_scope(**fields):
    """Temporarily override fields of the calling thread's HimAlgo, e.g. ``algo_scope(wino_min_c=-1)`` (direct form),
    ``algo_scope(ksplit_max=4)``, ``algo_scope(disable=ALGO_NO_FEWCH_MFMA)``."""
    a = current_algo()
    saved = HimAlgo.from_buffer_copy(a)
    try:
        for k, v in fields.items():
            setattr(a, k, v)
        yield a
    finally:
        ctypes.memmove(ctypes.byref(a), ctypes.byref(saved), ctypes.sizeof(HimAlgo))


def _conv_desc(x, w, stride, pad, pad_mode, act, slope, frozen=False):
    B, Cin, H, W = x.shape
    Cout, Cin2, KH, KW = w.shape
    if Cin2 != Cin:
        raise HimError('conv2d: weight expects %d input channels, got %d' % (Cin2, Cin))
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    d = HimConv2d(B, Cin, H, W, Cout, KH, KW, stride, pad, pad_mode, OH, OW, act, slope, current_algo())
    if frozen:
        d.algo.disable |= _ALGO_FROZEN
    return d


_ALGO_FROZEN = 1 << 9


# Weight panels: the MFMA kernels read the weights regrouped (include/him.h "Weight panels").  Weights change once
# per optimizer step, so the panel of every nn.Parameter is cached on the parameter and rebuilt (a) by
# FusedAdam.step() right after the update -- on the optimizer's stream, under the other network's backward -- or (b)
# lazily when the parameter's version counter / storage moved (load_state_dict, torch optimizers).  Code that writes
# weights behind torch's back (raw pointers, ``p.data.copy_``) must call ``invalidate_panels``.
PANEL_FWD, PANEL_BWD_DATA = 0, 1


class _Panel(object):
    __slots__ = ('buf', 'nbytes', 'desc', 'is_deconv', 'kind', 'token', 'event', 'stream_id')


def _panel_token(w):
    return (w._version, w.data_ptr(), getattr(w, '_him_gen', 0))


def _build_panel(w, e):
    fn = lib.him_deconv2d_panel_build if e.is_deconv else lib.him_conv2d_panel_build
    fn(ctypes.byref(e.desc), e.kind, _p(w), _p(e.buf), e.nbytes, _stream())
    e.token = _panel_token(w)
    e.event = torch.cuda.Event()
    e.event.record(torch.cuda.current_stream())
    e.stream_id = _stream()


_PANEL_KEYS = {}     # (descriptor bytes, kind asked for, is_deconv) -> (cache key, panel bytes, kind of the panel that serves it)


def _panel(w, d, kind, is_deconv):
    """Device pointer of the cached panel of parameter ``w`` for descriptor ``d`` (0: use the plain entry point)."""
    if not SCHED.panel_cache or not isinstance(w, torch.nn.Parameter):
        return 0
    cache = w.__dict__.get('_him_panels')
    if cache is None:
        cache = w.__dict__['_him_panels'] = {}
    # every descriptor field the panel LAYOUT can depend on (include/him.h "Weight panels": Winograd eligibility needs
    # pad 1 + planes >= 2x2 + an unchanged plane size for the data gradient; tiny heads switch on the output size)
    # ... and the HimAlgo the panel is built with (Winograd thresholds decide the layout)
    # The layout is the LIBRARY's choice for this descriptor (him_conv2d_panel_layout: implicit GEMM / Winograd F(2x2) /
    # fused Winograd / F(4x4)) and depends on batch and plane size, not on the weight alone: the same frozen VGG weight on
    # the last, smaller batch of an epoch leaves F(4x4) (a 36-position panel read as a 16-position one).
    #  the two size / layout queries are ctypes calls; their answer is a pure function of (descriptor, kind):
    # memoised per descriptor bytes so a cache hit stays pure Python on the launch path
    mk = (bytes(d), kind, is_deconv)
    hit = _PANEL_KEYS.get(mk)
    if hit is None:
        if kind == PANEL_BWD_DATA and not is_deconv and lib.him_conv2d_bwd_data_shares_fwd_panel(ctypes.byref(d)):
            kind = PANEL_FWD        # separate-transform Winograd layers: ONE panel per weight serves both directions
        nbytes = int((lib.him_deconv2d_panel_bytes if is_deconv else lib.him_conv2d_panel_bytes)(ctypes.byref(d), kind))
        if is_deconv:
            key = (kind, d.stride, d.pad, d.out_pad, nbytes, bytes(d.algo))
        else:
            key = (kind, int(lib.him_conv2d_panel_layout(ctypes.byref(d), kind)), nbytes, d.stride, d.pad, d.pad_mode,
                   d.H >= 2 and d.W >= 2, d.OH == d.H and d.OW == d.W,
                   d.B * d.OH * d.OW < 131072, d.B * d.Cin * d.H * d.W < (1 << 29), d.B * d.Cout * d.H * d.W < (1 << 29),
                   bytes(d.algo))
        if len(_PANEL_KEYS) > 4096:
            _PANEL_KEYS.clear()
        hit = _PANEL_KEYS[mk] = (key, nbytes, kind)
    key, nbytes, kind = hit
    e = cache.get(key)
    if e is None:
        e = cache[key] = _Panel()
        e.kind, e.is_deconv = kind, is_deconv
        e.desc = type(d).from_buffer_copy(d)
        e.nbytes = nbytes
        e.buf = torch.empty(e.nbytes // 4, dtype=torch.float32, device=w.device) if e.nbytes else None
        e.token = None
    if e.buf is None:
        return 0
    if e.token != _panel_token(w):
        _build_panel(w, e)
    if e.stream_id != _stream():
        torch.cuda.current_stream().wait_event(e.event)
    return e.buf.data_ptr()


def refresh_panels(params):
    """Rebuild the cached panels of ``params`` on the current stream (called by FusedAdam.step after the update)."""
    for p in params:
        p._him_gen = getattr(p, '_him_gen', 0) + 1
        cache = p.__dict__.get('_him_panels')
        if cache:
            for e in cache.values():
                if e.buf is not None:
                    _build_panel(p, e)


def invalidate_panels(params):
    for p in params:
        p._him_gen = getattr(p, '_him_gen', 0) + 1


def set_winograd_min_channels(c):
    """3x3 stride-1 convs with >= c channels on both sides run as Winograd F(2x2,3x3); c <= 0: EVERY Winograd form off.
    Sets ``current_algo().wino_min_c`` of the calling thread (cached panels are keyed by the HimAlgo they were built
    with); returns the previous setting (0 = the library default, 256)."""
    a = current_algo()
    prev = int(a.wino_min_c)
    a.wino_min_c = -1 if c <= 0 else (0 if int(c) == 256 else int(c))
    return prev if prev != 0 else 256


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, pad_mode, act, slope, premasked=False, gate_dx=False):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        x = x.contiguous()
        _chk(x, w, b)
        d = _conv_desc(x, w, stride, pad, pad_mode, act, slope, frozen=bool(getattr(w, '_him_frozen', False)))
        if premasked and act != ACT_RELU:
            raise HimError('conv2d: only a ReLU gate can be applied by the producers of the output gradient')
        ctx.premasked, ctx.gate_dx = bool(premasked), bool(gate_dx)
        y = torch.empty((d.B, d.Cout, d.OH, d.OW), dtype=torch.float32, device=x.device)
        nb = lib.him_conv2d_fwd_ws(ctypes.byref(d))
        ws = _ws(nb, x)
        pan = _panel(w, d, PANEL_FWD, False)
        keep = None
        if pan and SCHED.keep_wino_input and ctx.needs_input_grad[1] and _direct(w):
            # separate-transform Winograd layer whose weight gradient will be taken: keep the transformed input
            nk = int(lib.him_conv2d_fwd_keep_bytes(ctypes.byref(d)))
            if nk:
                keep = torch.empty(nk // 4, dtype=torch.float32, device=x.device)
        if keep is not None:
            lib.him_conv2d_fwd_panel_keep(ctypes.byref(d), _p(x), pan, _p(b), _p(y), _p(keep), _p(ws), nb, _stream())
        elif pan:
            lib.him_conv2d_fwd_panel(ctypes.byref(d), _p(x), pan, _p(b), _p(y), _p(ws), nb, _stream())
        else:
            lib.him_conv2d_fwd(ctypes.byref(d), _p(x), _p(w), _p(b), _p(y), _p(ws), nb, _stream())
        ctx.d = d
        ctx.x, ctx.w, ctx.b, ctx.keep = x, w, b, keep
        ctx.gslice = getattr(x, '_him_grad_slice', None)
        # the OUTPUT must go through save_for_backward: a plain ctx attribute closes a tensor -> grad_fn -> ctx -> tensor
        # cycle through C++ that no collector sees, and with it the whole upstream graph of every step leaks
        ctx.save_for_backward(y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 10
        d, x, w, b = ctx.d, ctx.x, ctx.w, ctx.b
        dy = dy.contiguous()
        st = _stream()
        if d.act != ACT_NONE and not ctx.premasked:
            dz = torch.empty_like(dy)
            lib.him_act_bwd(_p(ctx.saved_tensors[0]), _p(dy), _p(dz), dy.numel(), d.act, d.slope, st)
        else:
            dz = dy
        dx = dw = db = None
        gs = ctx.gslice
        if ctx.needs_input_grad[0] and _wkey(w) not in SKIP_DGRAD and gs is not None and gs[1] <= 4 < d.Cin:
            # only channels [c0, c0+n) of the input can use a gradient (first PatchGAN conv: the image channels behind
            # 35..70 channels of data): data gradient of the n-channel weight slice (tiny-M kernel), zeros elsewhere
            c0, n = gs
            d2 = HimConv2d.from_buffer_copy(d)
            d2.Cin = n
            dxs = torch.empty((d.B, n, d.H, d.W), dtype=torch.float32, device=x.device)
            wsl = w.detach()[:, c0:c0 + n].contiguous()
            nb = lib.him_conv2d_bwd_data_ws(ctypes.byref(d2))
            ws = _ws(nb, x)
            lib.him_conv2d_bwd_data(ctypes.byref(d2), _p(dz), _p(wsl), _p(dxs), _p(ws), nb, st)
            dx = torch.zeros_like(x)
            lib.him_copy_channels(_p(dxs), n, 0, _p(dx), d.Cin, c0, n, d.B, d.H * d.W, 0, 0, 0, st)
        elif ctx.needs_input_grad[0] and _wkey(w) not in SKIP_DGRAD:
            dx = torch.empty_like(x)
            nb = lib.him_conv2d_bwd_data_ws(ctypes.byref(d))
            ws = _ws(nb, x)
            pan = _panel(w, d, PANEL_BWD_DATA, False)
            if ctx.gate_dx:     # x is a ReLU output whose consumers apply its gate: dx = (x > 0) * dgrad
                lib.him_conv2d_bwd_data_gated(ctypes.byref(d), _p(dz), 0 if pan else _p(w), pan or 0, _p(x), _p(dx),
                                              _p(ws), nb, st)
            elif pan:
                lib.him_conv2d_bwd_data_panel(ctypes.byref(d), _p(dz), pan, _p(dx), _p(ws), nb, st)
            else:
                lib.him_conv2d_bwd_data(ctypes.byref(d), _p(dz), _p(w), _p(dx), _p(ws), nb, st)
        skip_w = _wkey(w) in SKIP_WGRAD
        need_w = ctx.needs_input_grad[1] and not skip_w
        need_b = b is not None and ctx.needs_input_grad[2] and not skip_w
        if need_w or need_b:
            nb = lib.him_conv2d_bwd_weight_ws(ctypes.byref(d))
            if need_w and _direct(w) and (not need_b or _direct(b)):
                with _wgrad_stream(x, dz, ctx.keep, weight=w):
                    ws = _ws(nb, x)
                    if ctx.keep is not None:
                        lib.him_conv2d_bwd_weight_kept(ctypes.byref(d), _p(ctx.keep), _p(dz), _p(w.grad),
                                                       _p(b.grad) if need_b else 0, 1, _p(ws), nb, _stream())
                    else:
                        lib.him_conv2d_bwd_weight(ctypes.byref(d), _p(x), _p(dz), _p(w.grad),
                                                  _p(b.grad) if need_b else 0, 1, _p(ws), nb, _stream())
                    _notify(w)
                    if need_b:
                        _notify(b)
            else:
                ws = _ws(nb, x)
                dw = torch.empty_like(w) if need_w else None
                db = torch.empty_like(b) if need_b else None
                lib.him_conv2d_bwd_weight(ctypes.byref(d), _p(x), _p(dz), _p(dw), _p(db), 0, _p(ws), nb, st)
        ctx.keep = None
        return dx, dw, db, None, None, None, None, None, None, None


class _OneHotConv2d(torch.autograd.Function):
    """conv2d over [one-hot(label) | dense channels] evaluated from the label ids (include/him.h "One-hot stems")."""

    @staticmethod
    def forward(ctx, x, label, n_onehot, w, b, pad, pad_mode, act, slope, stride=1, dense_only=False):
        """``dense_only``: ``x`` holds only the dense channels (B, Cin - n_onehot, H, W) -- or is None when there are none --
        instead of the (B, Cin, H, W) concatenation with the materialised one-hot block (``LabelCond`` inputs)."""
        ctx.set_materialize_grads(False)
        label = label.contiguous()
        if x is not None:
            x = x.contiguous()
        _chk(x, label, w, b)
        d = _ids_conv_desc(label, w, stride, pad, pad_mode, act, slope)
        if dense_only and (0 if x is None else x.shape[1]) != d.Cin - n_onehot:
            raise HimError('one-hot conv: %d dense channels given, the weight expects %d'
                           % (0 if x is None else x.shape[1], d.Cin - n_onehot))
        y = torch.empty((d.B, d.Cout, d.OH, d.OW), dtype=torch.float32, device=label.device)
        nb = lib.him_conv2d_onehot_fwd_ws(ctypes.byref(d), n_onehot)
        ws = _ws(nb, label)
        fn = lib.him_conv2d_onehot_fwd_dense if dense_only else lib.him_conv2d_onehot_fwd
        fn(ctypes.byref(d), _p(label), n_onehot, _p(x), _p(w), _p(b), _p(y), _p(ws), nb, _stream())
        ctx.d, ctx.n_onehot, ctx.dense_only = d, n_onehot, bool(dense_only)
        ctx.x, ctx.label, ctx.w, ctx.b = x, label, w, b
        ctx.save_for_backward(y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 11
        if ctx.needs_input_grad[0]:
            raise HimError('one-hot stem conv has no data gradient (its input is data)')
        d, x, label, w, b = ctx.d, ctx.x, ctx.label, ctx.w, ctx.b
        bw = lib.him_conv2d_onehot_bwd_weight_dense if ctx.dense_only else lib.him_conv2d_onehot_bwd_weight
        dy = dy.contiguous()
        st = _stream()
        if d.act != ACT_NONE:
            dz = torch.empty_like(dy)
            lib.him_act_bwd(_p(ctx.saved_tensors[0]), _p(dy), _p(dz), dy.numel(), d.act, d.slope, st)
        else:
            dz = dy
        dw = db = None
        skip_w = _wkey(w) in SKIP_WGRAD
        need_w = ctx.needs_input_grad[3] and not skip_w
        need_b = b is not None and ctx.needs_input_grad[4] and not skip_w
        if need_w or need_b:
            nb = lib.him_conv2d_onehot_bwd_weight_ws(ctypes.byref(d), ctx.n_onehot)
            if need_w and _direct(w) and (not need_b or _direct(b)):
                # the range the early Adam pieces must leave alone: the weight's slot and, when the bias gradient is live
                # (dead_bias_skip off, BatchNorm in eval mode), the bias slot behind it -- run_bias_grad writes b.grad in
                # the SAME launches as w.grad, i.e. after the two events below (the early step_range(hi, total)
                # read a zeroed / partial b.grad).  A bias slot that is not the weight's neighbour: no split.
                stem_range = tuple(w._him_arena_range) if hasattr(w, '_him_arena_range') else None
                if stem_range is not None and need_b:
                    br = getattr(b, '_him_arena_range', None)
                    if br is not None and 0 <= br[0] - stem_range[1] < 64:
                        stem_range = (stem_range[0], br[1])
                    elif br is not None and 0 <= stem_range[0] - br[1] < 64:
                        stem_range = (br[0], stem_range[1])
                    else:
                        stem_range = None
                if SCHED.adam_split_stem and SCHED.wgrad_stream and stem_range is not None:
                    # A generator stem is the LAST node of a chain-shaped generator's backward pass: what the streams hold
                    # at this moment is every data gradient (current stream) and every OTHER weight gradient (weight-gradient
                    # stream) of the network.  A trainer may start the optimizer step of everything else behind these two
                    # events, next to this layer's own weight gradient (Pix2PixHDModel_condImg.optimize_parameters).
                    dev = label.device
                    cur = torch.cuda.current_stream(dev)
                    side = _WGRAD_ROUTE.get(cur.cuda_stream) or _side_stream(dev)
                    ev_main, ev_side = torch.cuda.Event(), torch.cuda.Event()
                    ev_main.record(cur)
                    ev_side.record(side)
                    _STEM_PRE[dev] = (ev_main, ev_side, stem_range, id(w))
                fork = SCHED.stem_wgrad_fork and SCHED.wgrad_stream and d.Cin > ctx.n_onehot
                if fork:
                    # the dense channels' slice (an MFMA weight gradient) + the bias gradient on THIS stream -- it has just
                    # delivered dz and, for a generator stem, has nothing left to do -- next to the label-id slice (run-length
                    # kernel, LDS-bound, 1.1 ms at C2) on the weight-gradient stream: disjoint elements of w.grad, disjoint
                    # workspace regions, a workspace each (a block of the other stream's pool may still be in use there)
                    ws_d = _ws(nb, label)
                    lib.him_conv2d_onehot_bwd_weight_part(ctypes.byref(d), _p(label), ctx.n_onehot, _p(x), int(ctx.dense_only),
                                                          _p(dz), _p(w.grad), _p(b.grad) if need_b else 0, 1, _p(ws_d), nb,
                                                          ONEHOT_PART_DENSE, st)
                    dense_done = torch.cuda.Event()
                    dense_done.record(torch.cuda.current_stream(label.device))
                with _wgrad_stream(x, dz, label):
                    ws = _ws(nb, label)
                    if fork:
                        lib.him_conv2d_onehot_bwd_weight_part(ctypes.byref(d), _p(label), ctx.n_onehot, _p(x),
                                                              int(ctx.dense_only), _p(dz), _p(w.grad), 0, 1, _p(ws), nb,
                                                              ONEHOT_PART_IDS, _stream())
                        torch.cuda.current_stream(label.device).wait_event(dense_done)   # w.grad is final behind BOTH parts
                    else:
                        bw(ctypes.byref(d), _p(label), ctx.n_onehot, _p(x), _p(dz), _p(w.grad),
                           _p(b.grad) if need_b else 0, 1, _p(ws), nb, _stream())
                    _notify(w)
                    if need_b:
                        _notify(b)
            else:
                ws = _ws(nb, label)
                dw = torch.empty_like(w) if need_w else None
                db = torch.empty_like(b) if need_b else None
                bw(ctypes.byref(d), _p(label), ctx.n_onehot, _p(x), _p(dz), _p(dw), _p(db), 0, _p(ws), nb, st)
        return None, None, None, dw, db, None, None, None, None, None, None


_STEM_PRE = {}      # device -> (event on the data-gradient stream, event on the weight-gradient stream, arena range, id(weight))


def take_stem_pre(device):
    """The events recorded in front of the LAST one-hot stem weight gradient on ``device`` (see _OneHotConv2d.backward), once."""
    return _STEM_PRE.pop(device, None)


def _ids_conv_desc(label, w, stride, pad, pad_mode, act, slope):
    """HimConv2d of a convolution whose input is [one-hot(label) | dense]: the weight names the channel count."""
    B, _, H, W = label.shape
    Cout, Cin, KH, KW = w.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    return HimConv2d(B, Cin, H, W, Cout, KH, KW, stride, pad, pad_mode, OH, OW, act, slope, current_algo())


class LabelCond(object):
    """The reference's ``encode_input`` result [one-hot(label) | dense channels] (pix2pixHD_condImg_model.py:144-174 + the
    cat at :204) kept as what it is made from: the (B,1,H,W) id map and the few dense channels (edges, (1-mask)*image, colour
    embedding).  Consumers that can read ids do -- the generators' stems and (round 5) the first PatchGAN convolution
    (``_OneHotConv2d``), the pooled discriminator scales (``pooled``: 3x3 class counts) -- so the one-hot block (147 MB at
    512x256 bs 8) is neither written nor copied into the discriminator inputs; anything else calls ``full()`` (materialised
    once, cached)."""

    def __init__(self, label, n_onehot, dense=None):
        if dense is not None and dense.shape[1] == 0:
            dense = None
        self.label, self.n_onehot, self.dense = label, int(n_onehot), dense
        self._full = self._pooled = None
        self._made = {}      # 'full' / 'pooled' -> (stream that materialised it, event behind its kernels)

    @property
    def n_dense(self):
        return 0 if self.dense is None else self.dense.shape[1]

    @property
    def shape(self):
        B, _, H, W = self.label.shape
        return torch.Size((B, self.n_onehot + self.n_dense, H, W))

    @property
    def device(self):
        return self.label.device

    is_cuda, requires_grad = True, False

    def dim(self):
        return 4

    def detach(self):
        return self

    def contiguous(self):
        return self

    def record_stream(self, stream):
        for t in (self.label, self.dense, self._full, self._pooled):
            if t is not None:
                t.record_stream(stream)

    def _made_on(self, what, t):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._made[what] = (_stream(), ev)
        return t

    def _cross_stream(self, what, t):
        """A cache hit from a stream other than the one that materialised ``t`` (the first caller is the real-image
        side stream when the discriminator input is not split): order the reader behind the producer and tell the caching
        allocator about the second user -- the model's own record_stream calls ran before the buffer existed."""
        made = self._made.get(what)
        if made is not None and made[0] != _stream():
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(made[1])
            t.record_stream(cur)
        return t

    def full(self):
        """The (B, n_onehot + n_dense, H, W) tensor itself (him_onehot + one channel copy), cached."""
        if self._full is not None:
            return self._cross_stream('full', self._full)
        if self._full is None:
            B, C, H, W = self.shape
            buf = torch.empty((B, C, H, W), dtype=torch.float32, device=self.device)
            st = _stream()
            lib.him_onehot(_p(self.label), _p(buf), B, self.n_onehot, C, 0, H * W, st)
            if self.dense is not None:
                lib.him_copy_channels(_p(self.dense), self.n_dense, 0, _p(buf), C, self.n_onehot, self.n_dense, B, H * W,
                                      0, 0, 0, st)
            self._full = self._made_on('full', mark_onehot(buf, self.label, self.n_onehot))
        return self._full

    def slice(self, c0, n):
        """Channels [c0, c0 + n): a ``LabelCond`` while the slice still starts with the whole one-hot block, a tensor else."""
        if c0 == 0 and n >= self.n_onehot:
            k = n - self.n_onehot
            if k == self.n_dense:
                return self
            return LabelCond(self.label, self.n_onehot, slice_channels(self.dense, 0, k) if k else None)
        if c0 >= self.n_onehot:
            return slice_channels(self.dense, c0 - self.n_onehot, n)
        return slice_channels(self.full(), c0, n)

    def pooled(self):
        """AvgPool2d(3, 2, 1, count_include_pad=False) of the whole thing (reference Discriminator_NET.py:31-32, LocalEnhancer
        Pix2Pix_NET.py:50-53): the one-hot channels as class counts of the 3x3 windows straight from the ids
        (him_onehot_pool3s2, bit-identical to pooling the one-hot), the dense channels through the ordinary pool."""
        if self._pooled is not None:
            return self._cross_stream('pooled', self._pooled)
        if self._pooled is None:
            with torch.no_grad():
                B, C, H, W = self.shape
                OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
                out = torch.empty((B, C, OH, OW), dtype=torch.float32, device=self.device)
                st = _stream()
                lib.him_onehot_pool3s2(_p(self.label), _p(out), B, self.n_onehot, C, 0, H, W, OH, OW, st)
                if self.dense is not None:
                    pd = avgpool3s2(self.dense)
                    lib.him_copy_channels(_p(pd), self.n_dense, 0, _p(out), C, self.n_onehot, self.n_dense, B, OH * OW,
                                          0, 0, 0, st)
                self._pooled = self._made_on('pooled', out)
        return self._pooled


def mark_onehot(x, label, n_onehot):
    """Declare that channels [0, n_onehot) of ``x`` are the one-hot encoding of the id map ``label`` (B,1,H,W): the
    first convolution applied to ``x`` may then be evaluated from the ids (``_OneHotConv2d``)."""
    x._him_onehot = (label, int(n_onehot))
    return x


def conv2d(x, w, b=None, stride=1, pad=0, pad_mode='zero', act='none', slope=0.2, grad_premasked=False, gate_dx=False):
    """act(conv2d(pad(x), w) + b); pad_mode 'reflect' == nn.ReflectionPad2d(pad) + Conv2d(padding=0).

    ReLU chains (VGG) can move every activation backward into the kernel that PRODUCES the gradient: with
    ``grad_premasked`` the caller guarantees that every consumer of this layer's output multiplies the gradient it
    sends back by (output > 0) -- this layer then skips its own ReLU backward pass; ``gate_dx`` makes this layer such a
    consumer for its input (dx = (x > 0) * dgrad, in the data-gradient kernel's epilogue)."""
    if isinstance(x, LabelCond):
        pm = PAD_REFLECT if pad_mode == 'reflect' else PAD_ZERO
        d = _ids_conv_desc(x.label, w, stride, pad, pm, ACTS[act], float(slope))
        if SCHED.onehot_stem and w.shape[1] == x.shape[1] and lib.him_conv2d_onehot_fwd_ws(ctypes.byref(d), x.n_onehot):
            return _OneHotConv2d.apply(x.dense, x.label, x.n_onehot, w, b, pad, pm, ACTS[act], float(slope), int(stride), True)
        x = x.full()
    oh = getattr(x, '_him_onehot', None) if SCHED.onehot_stem else None
    if oh is not None and stride == 1 and not x.requires_grad:
        label, n_onehot = oh
        pm = PAD_REFLECT if pad_mode == 'reflect' else PAD_ZERO
        d = _conv_desc(x, w, 1, pad, pm, ACTS[act], float(slope))
        if lib.him_conv2d_onehot_fwd_ws(ctypes.byref(d), n_onehot):
            return _OneHotConv2d.apply(x, label, n_onehot, w, b, pad, pm, ACTS[act], float(slope))
    return _Conv2d.apply(x, w, b, stride, pad, PAD_REFLECT if pad_mode == 'reflect' else PAD_ZERO, ACTS[act],
                         float(slope), bool(grad_premasked), bool(gate_dx))


class CondImage(object):
    """Discriminator input handed over in two parts: ``cond`` (B,Cc,H,W: data, never differentiated -- one-hot labels,
    edges, the conditioning image) and ``image`` (B,Ci,H,W: the real / generated picture).  The reference concatenates
    them (pix2pixHD_condImg_model.py:176-182) and pools the 41-channel tensor per scale; kept apart, the pooled condition
    is computed once per step for all passes (``cond_pyramid``), and the gradient of the first PatchGAN conv comes back as
    Ci channels instead of a zero-filled (Cc+Ci)-channel tensor that the pooling / cat backward would walk again."""

    def __init__(self, cond, image):
        self.cond, self.image = cond, image

    def cat(self):
        cond = self.cond.full() if isinstance(self.cond, LabelCond) else self.cond
        return cat_channels([cond, self.image])


def cond_pyramid(cond, levels, prefill=0, image_channels=3):
    """[cond, pool(cond), pool(pool(cond)), ...] (AvgPool2d(3, 2, 1, count_include_pad=False)), cached on the tensor.
    ``prefill``: per level, that many (B, Cc + image_channels, H, W) input buffers of the first PatchGAN conv with the
    condition already copied in (``_CondImageConv2d`` takes one per pass and adds only the image channels): the 164 MB
    condition copy of the fake-image pass sat on the main stream between the generator's last conv and the first
    discriminator conv; filled here -- at the start of the step, next to the previous step's Adam pass -- it is off the
    critical path."""
    pyr = getattr(cond, '_him_pyramid', None)
    if pyr is None or len(pyr) < levels:
        with torch.no_grad():
            pyr = [cond.detach()]
            while len(pyr) < levels:
                # a LabelCond's next level comes from the ids (3x3 class counts); from there on the levels are tensors
                pyr.append(pyr[-1].pooled() if isinstance(pyr[-1], LabelCond) else avgpool3s2(pyr[-1]))
            if prefill > 0 and SCHED.d_prefill_cond:
                st = _stream()
                for c in pyr:
                    if isinstance(c, LabelCond):
                        # scale 0 reads the ids: the buffers hold [dense condition channels | image slot] only
                        B, _, H, W = c.shape
                        Cd, bufs = c.n_dense, []
                        for _ in range(prefill):
                            x = torch.empty((B, Cd + image_channels, H, W), dtype=torch.float32, device=c.device)
                            if Cd:
                                lib.him_copy_channels(_p(c.dense), Cd, 0, _p(x), Cd + image_channels, 0, Cd, B, H * W, 0, 0,
                                                      0, st)
                            bufs.append(x)
                        c._him_prefilled = bufs
                        continue
                    B, Cc, H, W = c.shape
                    bufs = []
                    for _ in range(prefill):
                        x = torch.empty((B, Cc + image_channels, H, W), dtype=torch.float32, device=c.device)
                        lib.him_copy_channels(_p(c), Cc, 0, _p(x), Cc + image_channels, 0, Cc, B, H * W, 0, 0, 0, st)
                        bufs.append(x)
                    c._him_prefilled = bufs
        cond._him_pyramid = pyr
    return pyr


class _CondImageConv2d(torch.autograd.Function):
    """act(conv2d([cond | image], w) + b), zero padding; gradient to ``image`` (and the parameters) only."""

    @staticmethod
    def forward(ctx, cond, image, w, b, stride, pad, act, slope):
        ctx.set_materialize_grads(False)
        cond, image = cond.contiguous(), image.contiguous()
        _chk(cond, image, w, b)
        B, Cc, H, W = cond.shape
        Ci = image.shape[1]
        if image.shape[0] != B or tuple(image.shape[2:]) != (H, W):
            raise HimError('cond/image conv: shapes %s and %s do not stack' % (tuple(cond.shape), tuple(image.shape)))
        st = _stream()
        pre = getattr(cond, '_him_prefilled', None)
        if pre and tuple(pre[-1].shape) == (B, Cc + Ci, H, W):
            x = pre.pop()                  # condition channels filled by cond_pyramid at the start of the step
            x.record_stream(torch.cuda.current_stream(x.device))
        else:
            x = torch.empty((B, Cc + Ci, H, W), dtype=torch.float32, device=cond.device)
            lib.him_copy_channels(_p(cond), Cc, 0, _p(x), Cc + Ci, 0, Cc, B, H * W, 0, 0, 0, st)
        lib.him_copy_channels(_p(image), Ci, 0, _p(x), Cc + Ci, Cc, Ci, B, H * W, 0, 0, 0, st)
        d = _conv_desc(x, w, stride, pad, PAD_ZERO, act, slope)
        y = torch.empty((d.B, d.Cout, d.OH, d.OW), dtype=torch.float32, device=x.device)
        nb = lib.him_conv2d_fwd_ws(ctypes.byref(d))
        ws = _ws(nb, x)
        pan = _panel(w, d, PANEL_FWD, False)
        if pan:
            lib.him_conv2d_fwd_panel(ctypes.byref(d), _p(x), pan, _p(b), _p(y), _p(ws), nb, st)
        else:
            lib.him_conv2d_fwd(ctypes.byref(d), _p(x), _p(w), _p(b), _p(y), _p(ws), nb, st)
        ctx.d, ctx.Cc, ctx.Ci = d, Cc, Ci
        ctx.x, ctx.w, ctx.b = x, w, b
        ctx.save_for_backward(y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 8
        d, x, w, b, Cc, Ci = ctx.d, ctx.x, ctx.w, ctx.b, ctx.Cc, ctx.Ci
        dy = dy.contiguous()
        st = _stream()
        if d.act != ACT_NONE:
            dz = torch.empty_like(dy)
            lib.him_act_bwd(_p(ctx.saved_tensors[0]), _p(dy), _p(dz), dy.numel(), d.act, d.slope, st)
        else:
            dz = dy
        dimg = dw = db = None
        if ctx.needs_input_grad[1] and _wkey(w) not in SKIP_DGRAD:
            d2 = HimConv2d.from_buffer_copy(d)
            d2.Cin = Ci
            dimg = torch.empty((d.B, Ci, d.H, d.W), dtype=torch.float32, device=x.device)
            wsl = w.detach()[:, Cc:].contiguous()
            nb = lib.him_conv2d_bwd_data_ws(ctypes.byref(d2))
            ws = _ws(nb, x)
            lib.him_conv2d_bwd_data(ctypes.byref(d2), _p(dz), _p(wsl), _p(dimg), _p(ws), nb, st)
        skip_w = _wkey(w) in SKIP_WGRAD
        need_w = ctx.needs_input_grad[2] and not skip_w
        need_b = b is not None and ctx.needs_input_grad[3] and not skip_w
        if need_w or need_b:
            nb = lib.him_conv2d_bwd_weight_ws(ctypes.byref(d))
            if need_w and _direct(w) and (not need_b or _direct(b)):
                with _wgrad_stream(x, dz, weight=w):
                    ws = _ws(nb, x)
                    lib.him_conv2d_bwd_weight(ctypes.byref(d), _p(x), _p(dz), _p(w.grad),
                                              _p(b.grad) if need_b else 0, 1, _p(ws), nb, _stream())
                    _notify(w)
                    if need_b:
                        _notify(b)
            else:
                ws = _ws(nb, x)
                dw = torch.empty_like(w) if need_w else None
                db = torch.empty_like(b) if need_b else None
                lib.him_conv2d_bwd_weight(ctypes.byref(d), _p(x), _p(dz), _p(dw), _p(db), 0, _p(ws), nb, st)
        return None, dimg, dw, db, None, None, None, None


class _IdsCondImageConv2d(torch.autograd.Function):
    """act(conv2d([one-hot(label) | dense condition | image], w) + b), zero padding, with the one-hot block read from the
    ids (reference: the first nn.Conv2d of every NLayerDiscriminator, Discriminator_NET.py:71-74, on the concatenation of
    pix2pixHD_condImg_model.py:176-186): forward = table lookups + the (dense condition | image) channels as an ordinary
    few-channel conv; weight gradient = run-length sums over the label rows + the few-channel weight gradient; data gradient
    to the image channels only (the 3-channel weight slice, as ``_CondImageConv2d``)."""

    @staticmethod
    def forward(ctx, cond, image, w, b, stride, pad, act, slope):
        ctx.set_materialize_grads(False)
        image = image.contiguous()
        _chk(image, w, b)
        B, _, H, W = cond.shape
        Cd, Ci, NC = cond.n_dense, image.shape[1], cond.n_onehot
        if image.shape[0] != B or tuple(image.shape[2:]) != (H, W):
            raise HimError('cond/image conv: shapes %s and %s do not stack' % (tuple(cond.shape), tuple(image.shape)))
        st = _stream()
        pre = getattr(cond, '_him_prefilled', None)
        if pre and tuple(pre[-1].shape) == (B, Cd + Ci, H, W):
            x = pre.pop()                  # dense condition channels filled by cond_pyramid at the start of the step
            x.record_stream(torch.cuda.current_stream(x.device))
        else:
            x = torch.empty((B, Cd + Ci, H, W), dtype=torch.float32, device=image.device)
            if Cd:
                lib.him_copy_channels(_p(cond.dense), Cd, 0, _p(x), Cd + Ci, 0, Cd, B, H * W, 0, 0, 0, st)
        lib.him_copy_channels(_p(image), Ci, 0, _p(x), Cd + Ci, Cd, Ci, B, H * W, 0, 0, 0, st)
        label = cond.label
        d = _ids_conv_desc(label, w, stride, pad, PAD_ZERO, act, slope)
        y = torch.empty((d.B, d.Cout, d.OH, d.OW), dtype=torch.float32, device=x.device)
        nb = lib.him_conv2d_onehot_fwd_ws(ctypes.byref(d), NC)
        ws = _ws(nb, x)
        lib.him_conv2d_onehot_fwd_dense(ctypes.byref(d), _p(label), NC, _p(x), _p(w), _p(b), _p(y), _p(ws), nb, st)
        ctx.d, ctx.NC, ctx.Cc, ctx.Ci = d, NC, NC + Cd, Ci
        ctx.x, ctx.label, ctx.w, ctx.b = x, label, w, b
        ctx.save_for_backward(y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 8
        d, x, label, w, b, Cc, Ci = ctx.d, ctx.x, ctx.label, ctx.w, ctx.b, ctx.Cc, ctx.Ci
        dy = dy.contiguous()
        st = _stream()
        if d.act != ACT_NONE:
            dz = torch.empty_like(dy)
            lib.him_act_bwd(_p(ctx.saved_tensors[0]), _p(dy), _p(dz), dy.numel(), d.act, d.slope, st)
        else:
            dz = dy
        dimg = dw = db = None
        if ctx.needs_input_grad[1] and _wkey(w) not in SKIP_DGRAD:
            d2 = HimConv2d.from_buffer_copy(d)
            d2.Cin = Ci
            dimg = torch.empty((d.B, Ci, d.H, d.W), dtype=torch.float32, device=x.device)
            wsl = w.detach()[:, Cc:].contiguous()
            nb = lib.him_conv2d_bwd_data_ws(ctypes.byref(d2))
            ws = _ws(nb, x)
            lib.him_conv2d_bwd_data(ctypes.byref(d2), _p(dz), _p(wsl), _p(dimg), _p(ws), nb, st)
        skip_w = _wkey(w) in SKIP_WGRAD
        need_w = ctx.needs_input_grad[2] and not skip_w
        need_b = b is not None and ctx.needs_input_grad[3] and not skip_w
        if need_w or need_b:
            nb = lib.him_conv2d_onehot_bwd_weight_ws(ctypes.byref(d), ctx.NC)
            if need_w and _direct(w) and (not need_b or _direct(b)):
                with _wgrad_stream(x, dz, label):
                    ws = _ws(nb, x)
                    lib.him_conv2d_onehot_bwd_weight_dense(ctypes.byref(d), _p(label), ctx.NC, _p(x), _p(dz), _p(w.grad),
                                                           _p(b.grad) if need_b else 0, 1, _p(ws), nb, _stream())
                    _notify(w)
                    if need_b:
                        _notify(b)
            else:
                ws = _ws(nb, x)
                dw = torch.empty_like(w) if need_w else None
                db = torch.empty_like(b) if need_b else None
                lib.him_conv2d_onehot_bwd_weight_dense(ctypes.byref(d), _p(label), ctx.NC, _p(x), _p(dz), _p(dw), _p(db), 0,
                                                       _p(ws), nb, st)
        return None, dimg, dw, db, None, None, None, None


def cond_image_conv2d(cond, image, w, b=None, stride=1, pad=0, act='none', slope=0.2):
    """The first PatchGAN convolution on a ``CondImage`` pair (see there)."""
    if isinstance(cond, LabelCond):
        d = _ids_conv_desc(cond.label, w, stride, pad, PAD_ZERO, ACTS[act], float(slope))
        if (SCHED.d_from_ids and w.shape[1] == cond.shape[1] + image.shape[1]
                and lib.him_conv2d_onehot_fwd_ws(ctypes.byref(d), cond.n_onehot)):
            return _IdsCondImageConv2d.apply(cond, image, w, b, stride, pad, ACTS[act], float(slope))
        cond = cond.full()
    return _CondImageConv2d.apply(cond, image, w, b, stride, pad, ACTS[act], float(slope))


class _Deconv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, out_pad, act, slope):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        x = x.contiguous()
        _chk(x, w, b)
        B, Cin, H, W = x.shape
        Cin2, Cout, KH, KW = w.shape
        if Cin2 != Cin:
            raise HimError('deconv2d: weight expects %d input channels, got %d' % (Cin2, Cin))
        OH = (H - 1) * stride - 2 * pad + KH + out_pad
        OW = (W - 1) * stride - 2 * pad + KW + out_pad
        d = HimDeconv2d(B, Cin, H, W, Cout, KH, KW, stride, pad, out_pad, OH, OW, act, slope, current_algo())
        y = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device)
        nb = lib.him_deconv2d_fwd_ws(ctypes.byref(d))
        ws = _ws(nb, x)
        pan = _panel(w, d, PANEL_FWD, True)
        if pan:
            lib.him_deconv2d_fwd_panel(ctypes.byref(d), _p(x), pan, _p(b), _p(y), _p(ws), nb, _stream())
        else:
            lib.him_deconv2d_fwd(ctypes.byref(d), _p(x), _p(w), _p(b), _p(y), _p(ws), nb, _stream())
        ctx.d = d
        ctx.x, ctx.w, ctx.b = x, w, b
        ctx.gslice = getattr(x, '_him_grad_slice', None)
        # the OUTPUT must go through save_for_backward: a plain ctx attribute closes a tensor -> grad_fn -> ctx -> tensor
        # cycle through C++ that no collector sees, and with it the whole upstream graph of every step leaks
        ctx.save_for_backward(y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 8
        d, x, w, b = ctx.d, ctx.x, ctx.w, ctx.b
        dy = dy.contiguous()
        st = _stream()
        if d.act != ACT_NONE:
            dz = torch.empty_like(dy)
            lib.him_act_bwd(_p(ctx.saved_tensors[0]), _p(dy), _p(dz), dy.numel(), d.act, d.slope, st)
        else:
            dz = dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            nb = lib.him_deconv2d_bwd_data_ws(ctypes.byref(d))
            ws = _ws(nb, x)
            pan = _panel(w, d, PANEL_BWD_DATA, True)
            if pan:
                lib.him_deconv2d_bwd_data_panel(ctypes.byref(d), _p(dz), pan, _p(dx), _p(ws), nb, st)
            else:
                lib.him_deconv2d_bwd_data(ctypes.byref(d), _p(dz), _p(w), _p(dx), _p(ws), nb, st)
        need_w = ctx.needs_input_grad[1]
        need_b = b is not None and ctx.needs_input_grad[2]
        if need_w or need_b:
            nb = lib.him_deconv2d_bwd_weight_ws(ctypes.byref(d))
            if need_w and _direct(w) and (not need_b or _direct(b)):
                with _wgrad_stream(x, dz, weight=w):
                    ws = _ws(nb, x)
                    lib.him_deconv2d_bwd_weight(ctypes.byref(d), _p(x), _p(dz), _p(w.grad),
                                                _p(b.grad) if need_b else 0, 1, _p(ws), nb, _stream())
                    _notify(w)
                    if need_b:
                        _notify(b)
            else:
                ws = _ws(nb, x)
                dw = torch.empty_like(w) if need_w else None
                db = torch.empty_like(b) if need_b else None
                lib.him_deconv2d_bwd_weight(ctypes.byref(d), _p(x), _p(dz), _p(dw), _p(db), 0, _p(ws), nb, st)
        return dx, dw, db, None, None, None, None, None


def conv_transpose2d(x, w, b=None, stride=2, pad=1, out_pad=1, act='none', slope=0.2):
    return _Deconv2d.apply(x, w, b, stride, pad, out_pad, ACTS[act], float(slope))


# ------------------------------------------------------------------------------------------------
# ResnetBlock with both InstanceNorms fused into the Winograd transforms (include/him.h "ResnetBlock")
# ------------------------------------------------------------------------------------------------
class _ResBlock(torch.autograd.Function):
    """out = x + IN(conv3(refpad(relu(IN(conv3(refpad(x)))))))  (reference models/layer_util.py:333-378) as ONE node:
    7 launches forward (two input transforms -- the second normalises on load --, two batched GEMMs, two output
    transforms that reduce the plane statistics), no normalised intermediate in HBM; backward applies the ReLU gate and
    the InstanceNorm backward inside the data gradient's output transform and adds the skip gradient in the last one.
    The conv biases feed an InstanceNorm (zero true gradient): they take part in the forward only."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, eps):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        _chk(x, w1, b1, w2, b2)
        B, Cn, H, W = x.shape
        d = HimResBlock(B, Cn, H, W, eps, current_algo())
        cd = _conv_desc(x, w1, 1, 1, PAD_REFLECT, ACT_NONE, 0.0)
        p1, p2 = _panel(w1, cd, PANEL_FWD, False), _panel(w2, cd, PANEL_FWD, False)
        if not p1 or not p2:
            raise HimError('fused ResnetBlock needs cached weight panels (nn.Parameter weights)')
        y1, y2, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        st1 = torch.empty(2 * B * Cn, dtype=torch.float32, device=x.device)
        st2 = torch.empty_like(st1)
        nb = lib.him_resblock_ws(ctypes.byref(d))
        ws = _ws(nb, x)
        lib.him_resblock_fwd(ctypes.byref(d), _p(x), p1, _p(b1), p2, _p(b2), _p(y1), _p(st1), _p(y2), _p(st2), _p(out),
                             _p(ws), nb, _stream())
        ctx.d, ctx.cd = d, cd
        ctx.x, ctx.w1, ctx.w2 = x, w1, w2
        ctx.save_for_backward(y1, st1, y2, st2)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 6
        g = g.contiguous()
        d, cd, x, w1, w2 = ctx.d, ctx.cd, ctx.x, ctx.w1, ctx.w2
        y1, st1, y2, st2 = ctx.saved_tensors
        q1, q2 = _panel(w1, cd, PANEL_BWD_DATA, False), _panel(w2, cd, PANEL_BWD_DATA, False)
        dy2, dy1 = torch.empty_like(x), torch.empty_like(x)
        dx = torch.empty_like(x)
        nb = lib.him_resblock_ws(ctypes.byref(d))
        ws = _ws(nb, x)
        lib.him_resblock_bwd_data(ctypes.byref(d), _p(g), _p(y1), _p(st1), _p(y2), _p(st2), q1, q2, _p(dy2), _p(dy1), _p(dx),
                                  _p(ws), nb, _stream())
        dws = [None, None]
        nbw = lib.him_resblock_bwd_weight_ws(ctypes.byref(d))
        for i, (w, src, stat, dy) in enumerate(((w2, y1, st1, dy2), (w1, x, None, dy1))):
            which = 2 - i
            if not ctx.needs_input_grad[1 if which == 1 else 3] or _wkey(w) in SKIP_WGRAD:
                continue
            if _direct(w):
                with _wgrad_stream(src, dy, stat):
                    wsw = _ws(nbw, x)
                    lib.him_resblock_bwd_weight(ctypes.byref(d), which, _p(src), _p(stat), _p(dy), _p(w.grad), 1, _p(wsw), nbw,
                                                _stream())
                    _notify(w)
            else:
                wsw = _ws(nbw, x)
                dw = torch.empty_like(w)
                lib.him_resblock_bwd_weight(ctypes.byref(d), which, _p(src), _p(stat), _p(dy), _p(dw), 0, _p(wsw), nbw, _stream())
                dws[which - 1] = dw
        return (dx if ctx.needs_input_grad[0] else None), dws[0], None, dws[1], None, None


# Off by default since round 3's 64x128 conv tiles: inside the multi-stream step the layer-by-layer block is 0.4 ms
# faster (59.6 vs 60.0 ms, three repetitions each, profiles/r03_tile_shape_ab.txt) although the fused unit launches
# fewer kernels and moves fewer bytes; HIM_RESBLOCK_FUSED=1 turns it on, tests/test_ops_gpu.py keeps it pinned.


def resblock_supported(x, w1, w2):
    """True when ``resnet_block`` can take (x, w1, w2): CUDA fp32, nn.Parameter weights (cached panels) and a shape in
    the separate-transform Winograd range (him_resblock_supported)."""
    if not (SCHED.resblock_fused and SCHED.panel_cache and x.is_cuda and x.dim() == 4 and isinstance(w1, torch.nn.Parameter)
            and isinstance(w2, torch.nn.Parameter)):
        return False
    B, Cn, H, W = x.shape
    if tuple(w1.shape) != (Cn, Cn, 3, 3) or tuple(w2.shape) != (Cn, Cn, 3, 3):
        return False
    return bool(lib.him_resblock_supported(ctypes.byref(HimResBlock(B, Cn, H, W, 1e-5, current_algo()))))


def resnet_block(x, w1, b1, w2, b2, eps=1e-5):
    """x + IN(conv3x3(refpad(relu(IN(conv3x3(refpad(x), w1, b1)))), w2, b2)) with the norms fused into the convolutions'
    Winograd transforms; b1 / b2 (in front of an InstanceNorm: zero true gradient) are used as data."""
    return _ResBlock.apply(x, w1, None if b1 is None else b1.detach(), w2, None if b2 is None else b2.detach(), float(eps))


# ------------------------------------------------------------------------------------------------
# instance norm (+activation, +residual)
# ------------------------------------------------------------------------------------------------
class _InstNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, act, slope, eps):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        _chk(x, residual)
        B, Cn, H, W = x.shape
        planes, hw = B * Cn, H * W
        y = torch.empty_like(x)
        mean = torch.empty(planes, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        lib.him_instnorm_fwd(_p(x), _p(residual), _p(y), _p(mean), _p(rstd), planes, hw, eps, act, slope, _stream())
        ctx.x, ctx.mean, ctx.rstd = x, mean, rstd
        ctx.cfg = (planes, hw, act, slope)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 5
        dy = dy.contiguous()
        planes, hw, act, slope = ctx.cfg
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(ctx.x)
            lib.him_instnorm_bwd(_p(ctx.x), _p(ctx.mean), _p(ctx.rstd), _p(dy), _p(dx), planes, hw, act, slope,
                                 _stream())
        dres = dy if (ctx.has_res and ctx.needs_input_grad[1]) else None
        return dx, dres, None, None, None


class _Conv2dIN(torch.autograd.Function):
    """act(InstanceNorm2d(conv2d(pad(x), w) + b)) [+ residual] through him_conv2d_in_act_fwd (include/him.h): taken for
    the descriptors whose forward is a split-K launch -- the InstanceNorm kernel reads the split-K slabs, the finish pass
    disappears; outputs bit-identical to conv2d + instance_norm.  Backward = InstanceNorm backward, then _Conv2d's."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, pad_mode, eps, act, slope, residual):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        _chk(x, w, b, residual)
        d = _conv_desc(x, w, stride, pad, pad_mode, ACT_NONE, 0.0)
        y = torch.empty((d.B, d.Cout, d.OH, d.OW), dtype=torch.float32, device=x.device)
        z = torch.empty_like(y)
        planes, hw = d.B * d.Cout, d.OH * d.OW
        mean = torch.empty(planes, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        nb = lib.him_conv2d_fwd_ws(ctypes.byref(d))
        ws = _ws(nb, x)
        pan = _panel(w, d, PANEL_FWD, False)
        lib.him_conv2d_in_act_fwd(ctypes.byref(d), _p(x), 0 if pan else _p(w), pan or 0, _p(b), _p(y), _p(residual), _p(z),
                                  _p(mean), _p(rstd), eps, act, slope, _p(ws), nb, _stream())
        # the fields _Conv2d.backward reads
        ctx.d, ctx.x, ctx.w, ctx.b, ctx.keep = d, x, w, b, None
        ctx.premasked, ctx.gate_dx = False, False
        ctx.gslice = getattr(x, '_him_grad_slice', None)
        ctx.cfg = (planes, hw, act, slope)
        ctx.has_res = residual is not None
        ctx.save_for_backward(y, mean, rstd)
        return z

    @staticmethod
    def backward(ctx, dz):
        if dz is None:
            return (None,) * 10
        dz = dz.contiguous()
        y, mean, rstd = ctx.saved_tensors
        planes, hw, act, slope = ctx.cfg
        dy = torch.empty_like(y)
        lib.him_instnorm_bwd(_p(y), _p(mean), _p(rstd), _p(dz), _p(dy), planes, hw, act, slope, _stream())
        dx, dw, db = _Conv2d.backward(ctx, dy)[:3]
        dres = dz if (ctx.has_res and ctx.needs_input_grad[9]) else None
        return dx, dw, db, None, None, None, None, None, None, dres


def conv2d_in_act(x, w, b=None, stride=1, pad=0, pad_mode='zero', eps=1e-5, act='none', slope=0.2, residual=None):
    """act(InstanceNorm2d(affine=False)(conv2d(pad(x), w) + b)) [+ residual]: ONE library call
    (him_conv2d_in_act_fwd) where the convolution is a split-K launch, conv2d + instance_norm otherwise."""
    pm = PAD_REFLECT if pad_mode == 'reflect' else PAD_ZERO
    if (SCHED.conv_in_fused and not isinstance(x, LabelCond) and getattr(x, '_him_onehot', None) is None
            and not getattr(w, '_him_frozen', False)):
        d = _conv_desc(x, w, stride, pad, pm, ACT_NONE, 0.0)
        if lib.him_conv2d_in_act_fused(ctypes.byref(d)):
            return _Conv2dIN.apply(x, w, b, stride, pad, pm, float(eps), ACTS[act], float(slope), residual)
    return instance_norm(conv2d(x, w, b, stride, pad, pad_mode, 'none', slope), residual, act, slope, eps)


def instance_norm(x, residual=None, act='none', slope=0.2, eps=1e-5):
    """act(InstanceNorm2d(affine=False)(x)) [+ residual]."""
    return _InstNorm.apply(x, residual, ACTS[act], float(slope), float(eps))


# ------------------------------------------------------------------------------------------------
# box2mask building blocks: BatchNorm2d, stand-alone activation, bilinear x2, channel log-softmax, mask losses
# ------------------------------------------------------------------------------------------------
class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, run_mean, run_var, training, momentum, eps, act, slope):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        _chk(x, residual, gamma, beta, run_mean, run_var)
        B, Cn, H, W = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(Cn, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        nb = lib.him_batchnorm_ws(Cn)
        ws = _ws(nb, x)
        lib.him_batchnorm_fwd(_p(x), _p(residual), _p(gamma), _p(beta), _p(run_mean), _p(run_var), _p(y), _p(mean),
                              _p(rstd), B, Cn, H * W, eps, momentum, 1 if training else 0, act, slope, _p(ws), nb, _stream())
        ctx.x, ctx.gamma, ctx.beta, ctx.mean, ctx.rstd = x, gamma, beta, mean, rstd
        ctx.cfg = (B, Cn, H * W, bool(training), act, slope)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 11
        dy = dy.contiguous()
        B, Cn, hw, training, act, slope = ctx.cfg
        x, gamma, beta = ctx.x, ctx.gamma, ctx.beta
        need_x = ctx.needs_input_grad[0]
        need_g = gamma is not None and ctx.needs_input_grad[2]
        need_b = beta is not None and ctx.needs_input_grad[3]
        dx = torch.empty_like(x) if need_x else None
        nb = lib.him_batchnorm_ws(Cn)
        ws = _ws(nb, x)
        direct = need_g and need_b and _direct(gamma) and _direct(beta)
        if direct:
            dg, db = gamma.grad, beta.grad
        else:
            dg = torch.empty_like(gamma) if need_g else None
            db = torch.empty_like(beta) if need_b else None
        lib.him_batchnorm_bwd(_p(x), _p(gamma), _p(beta), _p(ctx.mean), _p(ctx.rstd), _p(dy), _p(dx), _p(dg), _p(db), B, Cn,
                              hw, 1 if training else 0, act, slope, 1 if direct else 0, _p(ws), nb, _stream())
        if direct:
            _notify(gamma)
            _notify(beta)
            dg = db = None
        dres = dy if (ctx.has_res and ctx.needs_input_grad[1]) else None
        return dx, dres, dg, db, None, None, None, None, None, None, None


def batch_norm(x, gamma, beta, run_mean, run_var, training=True, momentum=0.1, eps=1e-5, act='none', slope=0.2,
               residual=None):
    """act(BatchNorm2d(x)) [+ residual]; training mode updates run_mean / run_var in place (torch semantics)."""
    return _BatchNorm.apply(x, residual, gamma, beta, run_mean, run_var, bool(training), float(momentum), float(eps),
                            ACTS[act], float(slope))


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        _chk(x)
        y = torch.empty_like(x)
        lib.him_act_fwd(_p(x), _p(y), x.numel(), act, slope, _stream())
        ctx.cfg = (act, slope)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None
        dy = dy.contiguous()
        (y,) = ctx.saved_tensors
        dz = torch.empty_like(dy)
        lib.him_act_bwd(_p(y), _p(dy), _p(dz), dy.numel(), ctx.cfg[0], ctx.cfg[1], _stream())
        return dz, None, None


def activation(x, act, slope=0.2):
    """stand-alone nn.ReLU / LeakyReLU / Tanh / Sigmoid (out of place)."""
    return _Act.apply(x, ACTS[act], float(slope))


class _Upsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, align):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        _chk(x)
        B, Cn, H, W = x.shape
        y = torch.empty((B, Cn, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        lib.him_upsample2_fwd(_p(x), _p(y), B * Cn, H, W, align, _stream())
        ctx.cfg = (B * Cn, H, W, align)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None
        dy = dy.contiguous()
        planes, H, W, align = ctx.cfg
        dx = torch.empty((dy.shape[0], dy.shape[1], H, W), dtype=torch.float32, device=dy.device)
        lib.him_upsample2_bwd(_p(dy), _p(dx), planes, H, W, align, _stream())
        return dx, None


def upsample_bilinear2(x, align_corners=False):
    """nn.Upsample(scale_factor=2, mode='bilinear')."""
    return _Upsample2.apply(x, 1 if align_corners else 0)


class _LogSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        _chk(x)
        B, Cn, H, W = x.shape
        y = torch.empty_like(x)
        lib.him_logsoftmax_fwd(_p(x), _p(y), B, Cn, H * W, _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None
        dy = dy.contiguous()
        (y,) = ctx.saved_tensors
        B, Cn, H, W = y.shape
        dx = torch.empty_like(y)
        lib.him_logsoftmax_bwd(_p(y), _p(dy), _p(dx), B, Cn, H * W, _stream())
        return dx


def log_softmax_channels(x):
    """nn.LogSoftmax(dim=1) on a (B,C,H,W) tensor."""
    return _LogSoftmax.apply(x)


class _GateComb(torch.autograd.Function):
    """(1 - p) * ctx + p * obj, p and obj (B,1,H,W) broadcast over ctx's channels (MaskTwoStreamConv_NET.py:213-221)."""

    @staticmethod
    def forward(ctx_, ctx, p, obj):
        ctx, p, obj = ctx.contiguous(), p.contiguous(), obj.contiguous()
        _chk(ctx, p, obj)
        B, C, H, W = ctx.shape
        if p.shape != (B, 1, H, W) or obj.shape != (B, 1, H, W):
            raise ValueError('gate_comb: gate / object logits must be (B,1,H,W)')
        out = torch.empty_like(ctx)
        lib.him_gate_comb_fwd(_p(ctx), _p(p), _p(obj), _p(out), B, C, H * W, _stream())
        ctx_.save_for_backward(ctx, p, obj)
        return out

    @staticmethod
    def backward(ctx_, dout):
        ctx, p, obj = ctx_.saved_tensors
        dout = dout.contiguous()
        B, C, H, W = ctx.shape
        dctx, dp, dobj = torch.empty_like(ctx), torch.empty_like(p), torch.empty_like(obj)
        lib.him_gate_comb_bwd(_p(ctx), _p(p), _p(obj), _p(dout), _p(dctx), _p(dp), _p(dobj), B, C, H * W, _stream())
        return dctx, dp, dobj


def gate_comb(ctx_logit, gate, obj_logit):
    return _GateComb.apply(ctx_logit, gate, obj_logit)


class _MaskedNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, label, mask):
        ctx.set_materialize_grads(False)
        logp, label, mask = logp.contiguous(), label.contiguous(), mask.contiguous()
        _chk(logp, label, mask)
        B, Cn, H, W = logp.shape
        out2 = torch.empty(2, dtype=torch.float32, device=logp.device)
        nb = lib.him_mask_loss_ws()
        ws = _ws(nb, logp)
        lib.him_masked_nll_fwd(_p(logp), _p(label), _p(mask), _p(out2), B, Cn, H * W, _p(ws), nb, _stream())
        ctx.label, ctx.mask, ctx.count, ctx.shape = label, mask, out2[1:2], (B, Cn, H, W)
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        B, Cn, H, W = ctx.shape
        g = g.contiguous()
        d = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        lib.him_masked_nll_bwd(_p(ctx.label), _p(ctx.mask), _p(g), _p(ctx.count), _p(d), B, Cn, H * W, _stream())
        return d, None, None


def masked_nll(logp, label, mask):
    """MaskReconLoss: NLLLoss2d(ignore_index) of log-probabilities (B,C,H,W) against the id map ``label`` (B,1,H,W or
    B,H,W, ids as floats) with the positions where ``mask`` < 0.5 ignored; mean over the valid positions."""
    return _MaskedNLL.apply(logp, label.detach(), mask.detach())


class _BCEMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, t):
        ctx.set_materialize_grads(False)
        p, t = p.contiguous(), t.contiguous()
        _chk(p, t)
        if p.shape != t.shape:
            raise HimError('bce: shape mismatch %s vs %s' % (tuple(p.shape), tuple(t.shape)))
        out = torch.empty((), dtype=torch.float32, device=p.device)
        nb = lib.him_mask_loss_ws()
        ws = _ws(nb, p)
        lib.him_bce_mean_fwd(_p(p), _p(t), p.numel(), _p(out), _p(ws), nb, _stream())
        ctx.p, ctx.t = p, t
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        g = g.contiguous()
        dp = torch.empty_like(ctx.p)
        lib.him_bce_mean_bwd(_p(ctx.p), _p(ctx.t), ctx.p.numel(), _p(g), _p(dp), _stream())
        return dp, None


def bce_mean(p, t):
    """nn.BCELoss()(p, t.detach())."""
    return _BCEMean.apply(p, t.detach())


class _SpaceBatch(torch.autograd.Function):
    """inverse = 0: (B,C,H,W) -> (B*d*d, C, H/d, W/d) phase images; inverse = 1: back.  A permutation: its adjoint is its
    inverse."""

    @staticmethod
    def forward(ctx, x, d, inverse):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        _chk(x)
        if inverse:
            Bd, Cn, Hd, Wd = x.shape
            B, H, W = Bd // (d * d), Hd * d, Wd * d
            y = torch.empty((B, Cn, H, W), dtype=torch.float32, device=x.device)
        else:
            B, Cn, H, W = x.shape
            y = torch.empty((B * d * d, Cn, H // d, W // d), dtype=torch.float32, device=x.device)
        lib.him_space_to_batch(_p(x), _p(y), B, Cn, H, W, d, inverse, _stream())
        ctx.cfg = (d, inverse)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        d, inverse = ctx.cfg
        return _SpaceBatch.apply(g, d, 0 if inverse else 1), None, None


def dilated_conv3x3(x, w, dilation):
    """nn.Conv2d(k=3, stride=1, padding=dilation, dilation=dilation, bias=False) (reference conv3x3,
    models/layer_util.py:254-256) = the plain pad-1 conv on the dilation^2 phase images of x."""
    d = int(dilation)
    if d == 1:
        return conv2d(x, w, None, 1, 1, 'zero', 'none')
    if x.shape[2] % d or x.shape[3] % d:
        raise HimError('dilated conv: plane %dx%d is not divisible by the dilation %d' % (x.shape[2], x.shape[3], d))
    return _SpaceBatch.apply(conv2d(_SpaceBatch.apply(x, d, 0), w, None, 1, 1, 'zero', 'none'), d, 1)


def lr_control(loss_d_real, loss_d_fake, margin=0.3):
    """(g_lr, d_lr) device scalars in {0., 1.} (reference models/Discriminator_NET.py:190-211), no host read-back."""
    a, b = loss_d_real.detach().reshape(1).contiguous(), loss_d_fake.detach().reshape(1).contiguous()
    _chk(a, b)
    out = torch.empty(2, dtype=torch.float32, device=a.device)
    lib.him_lr_control(_p(a), _p(b), float(margin), _p(out), _stream())
    return out[0], out[1]


# ------------------------------------------------------------------------------------------------
# pooling
# ------------------------------------------------------------------------------------------------
class _AvgPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        x = x.contiguous()
        _chk(x)
        B, Cn, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, Cn, OH, OW), dtype=torch.float32, device=x.device)
        lib.him_avgpool3s2_fwd(_p(x), _p(y), B * Cn, H, W, OH, OW, _stream())
        ctx.shape = (B, Cn, H, W, OH, OW)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None
        B, Cn, H, W, OH, OW = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((B, Cn, H, W), dtype=torch.float32, device=dy.device)
        lib.him_avgpool3s2_bwd(_p(dy), _p(dx), B * Cn, H, W, OH, OW, _stream())
        return dx


def avgpool3s2(x):
    """nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False)."""
    if isinstance(x, LabelCond):
        return x.pooled()
    y = _AvgPool3s2.apply(x)
    gs = getattr(x, '_him_grad_slice', None)
    if gs is not None:
        y._him_grad_slice = gs      # channel-wise op: the same slice is the only one that needs a gradient
    return y


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, relu_gate=False):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        x = x.contiguous()
        _chk(x)
        B, Cn, H, W = x.shape
        y = torch.empty((B, Cn, H // k, W // k), dtype=torch.float32, device=x.device)
        lib.him_maxpool_fwd(_p(x), _p(y), B * Cn, H, W, k, _stream())
        ctx.x, ctx.k, ctx.relu_gate = x, k, bool(relu_gate)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None
        x, k = ctx.x, ctx.k
        B, Cn, H, W = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        (lib.him_maxpool_relu_bwd if ctx.relu_gate else lib.him_maxpool_bwd)(_p(x), _p(dy), _p(dx), B * Cn, H, W, k,
                                                                            _stream())
        return dx, None, None


def maxpool(x, k, relu_gate=False):
    """``relu_gate``: x is the output of a ReLU layer that was told its gradient arrives gated (conv2d's
    ``grad_premasked``): the pool's backward applies (x > 0) to what it routes."""
    return _MaxPool.apply(x, int(k), bool(relu_gate))


# ------------------------------------------------------------------------------------------------
# channel plumbing: cat / slice / mask-multiply / blend
# ------------------------------------------------------------------------------------------------
class _CatMask(torch.autograd.Function):
    """out = cat(f_i(mask) * tensor_i, dim=1); mask (B,1,H,W) or None; mode 0 none, 1 mask, 2 (1-mask) -- one int for every
    tensor or a tuple with one mode per tensor."""

    @staticmethod
    def forward(ctx, mask, mode, *ts):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        ts = [t.contiguous() for t in ts]
        _chk(mask, *ts)
        B, _, H, W = ts[0].shape
        Ctot = sum(t.shape[1] for t in ts)
        out = torch.empty((B, Ctot, H, W), dtype=torch.float32, device=ts[0].device)
        modes = tuple(mode) if isinstance(mode, (tuple, list)) else (mode,) * len(ts)
        if len(modes) != len(ts):
            raise ValueError('cat_channels: %d mask modes for %d tensors' % (len(modes), len(ts)))
        st, c0 = _stream(), 0
        for t, md in zip(ts, modes):
            lib.him_copy_channels(_p(t), t.shape[1], 0, _p(out), Ctot, c0, t.shape[1], B, H * W, _p(mask), md, 0, st)
            c0 += t.shape[1]
        ctx.mask, ctx.modes = mask, modes
        ctx.chs = [t.shape[1] for t in ts]
        return out

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return (None,) * (2 + len(ctx.chs))
        dout = dout.contiguous()
        B, Ctot, H, W = dout.shape
        st, c0, grads = _stream(), 0, []
        for i, ch in enumerate(ctx.chs):
            if ctx.needs_input_grad[2 + i]:
                g = torch.empty((B, ch, H, W), dtype=torch.float32, device=dout.device)
                lib.him_copy_channels(_p(dout), Ctot, c0, _p(g), ch, 0, ch, B, H * W, _p(ctx.mask), ctx.modes[i], 0, st)
                grads.append(g)
            else:
                grads.append(None)
            c0 += ch
        return (None, None) + tuple(grads)


def cat_channels(tensors, mask=None, mask_mode=0):
    """cat(tensors, 1), optionally times the mask (mode 1) or its complement (mode 2); ``mask_mode`` may hold one mode per
    tensor (the 'concat' feature fusion: cat((1-m)*ctx, m*obj))."""
    tensors = [t.full() if isinstance(t, LabelCond) else t for t in tensors]
    if mask is None:
        mask_mode = 0
    elif isinstance(mask_mode, (tuple, list)):
        mask_mode = tuple(int(m) for m in mask_mode)
    else:
        mask_mode = int(mask_mode)
    out = _CatMask.apply(mask, mask_mode, *tensors)
    need = [bool(t.requires_grad) for t in tensors]
    if torch.is_grad_enabled() and any(need) and not all(need):
        # only a channel slice of this tensor can receive a gradient (discriminator input = [data | image]): consumers
        # may restrict their data gradient to it (``_Conv2d.backward``)
        c, lo, hi = 0, None, 0
        for t, n in zip(tensors, need):
            if n:
                lo = c if lo is None else lo
                hi = c + t.shape[1]
            c += t.shape[1]
        out._him_grad_slice = (lo, hi - lo)
    return out


def mul_mask(x, mask):
    """x * mask.repeat(1, C, 1, 1)."""
    return _CatMask.apply(mask, 1, x)


class _Blend(torch.autograd.Function):
    """out = (1-m) * a[:, a0:a0+C] + m * b    (m: (B,1,H,W))."""

    @staticmethod
    def forward(ctx, a, a0, b, m):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        a, b, m = a.contiguous(), b.contiguous(), m.contiguous()
        _chk(a, b, m)
        B, Cn, H, W = b.shape
        out = torch.empty_like(b)
        lib.him_blend(_p(a), a.shape[1], a0, _p(b), Cn, 0, _p(m), _p(out), B, Cn, H * W, _stream())
        ctx.m, ctx.a0, ctx.Ca = m, a0, a.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return None, None, None, None
        dout = dout.contiguous()
        B, Cn, H, W = dout.shape
        st = _stream()
        da = db = None
        if ctx.needs_input_grad[0]:
            if ctx.Ca == Cn:
                da = torch.empty_like(dout)
            else:
                da = torch.zeros((B, ctx.Ca, H, W), dtype=torch.float32, device=dout.device)
            lib.him_copy_channels(_p(dout), Cn, 0, _p(da), ctx.Ca, ctx.a0, Cn, B, H * W, _p(ctx.m), 2, 0, st)
        if ctx.needs_input_grad[2]:
            db = torch.empty_like(dout)
            lib.him_copy_channels(_p(dout), Cn, 0, _p(db), Cn, 0, Cn, B, H * W, _p(ctx.m), 1, 0, st)
        return da, None, db, None


def blend(a, b, m, a0=0):
    if isinstance(a, LabelCond):
        if a0 >= a.n_onehot:        # the channels read are dense ones (the output gate reads the image out of the G input)
            a, a0 = a.dense, a0 - a.n_onehot
        else:
            a = a.full()
    return _Blend.apply(a, int(a0), b, m)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        a, b = a.contiguous(), b.contiguous()
        _chk(a, b)
        out = torch.empty_like(a)
        lib.him_add(_p(a), _p(b), _p(out), a.numel(), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


# ------------------------------------------------------------------------------------------------
# input encoding (no gradients)
# ------------------------------------------------------------------------------------------------
def encode_channels(label, inst, image, mask_in, label_nc, use_edges, extra_after=0, color_emb=None, lazy=False):
    """Builds, in ONE (B, Ctot, H, W) buffer and without any torch.cat,
         [ one-hot(label) | edges(inst)? | (1-mask)*image | emb*mask ? ]
    (reference encode_input + the torch.cat at pix2pixHD_condImg_model.py:204).  Returns (buf, n_label, n_cond).
    ``lazy`` (label_nc > 0): ``buf`` is a ``LabelCond`` -- the id map + the dense channels; the one-hot block is written
    only if a consumer asks for it (``LabelCond.full``)."""
    _chk(label, inst, image, mask_in, color_emb)
    B, _, H, W = image.shape
    hw = H * W
    n_label = (label_nc if label_nc else label.shape[1]) + (1 if use_edges else 0)
    n_cond = 3 + (3 if color_emb is not None else 0)
    Ctot = n_label + n_cond
    if lazy and label_nc:
        Cd = Ctot - label_nc
        dense = torch.empty((B, Cd, H, W), dtype=torch.float32, device=image.device)
        st, c = _stream(), 0
        if use_edges:
            lib.him_edges(_p(inst), _p(dense), B, H, W, Cd, 0, st)
            c = 1
        lib.him_copy_channels(_p(image), 3, 0, _p(dense), Cd, c, 3, B, hw, _p(mask_in), 2, 0, st)
        if color_emb is not None:
            lib.him_tile_embed(_p(color_emb), _p(mask_in), _p(dense), B, Cd, c + 3, hw, st)
        return LabelCond(label.contiguous(), label_nc, dense), n_label, n_cond
    buf = torch.empty((B, Ctot, H, W), dtype=torch.float32, device=image.device)
    st = _stream()
    if label_nc:
        lib.him_onehot(_p(label), _p(buf), B, label_nc, Ctot, 0, hw, st)
        c = label_nc
    else:
        c = label.shape[1]
        lib.him_copy_channels(_p(label), c, 0, _p(buf), Ctot, 0, c, B, hw, 0, 0, 0, st)
    if use_edges:
        lib.him_edges(_p(inst), _p(buf), B, H, W, Ctot, c, st)
        c += 1
    lib.him_copy_channels(_p(image), 3, 0, _p(buf), Ctot, c, 3, B, hw, _p(mask_in), 2, 0, st)
    c += 3
    if color_emb is not None:
        lib.him_tile_embed(_p(color_emb), _p(mask_in), _p(buf), B, Ctot, c, hw, st)
    if label_nc:
        mark_onehot(buf, label, label_nc)
    return buf, n_label, n_cond


def slice_channels(x, c0, n):
    """contiguous copy of x[:, c0:c0+n] (no gradient)."""
    if isinstance(x, LabelCond):
        return x.slice(c0, n)
    _chk(x)
    B, Cn, H, W = x.shape
    out = torch.empty((B, n, H, W), dtype=torch.float32, device=x.device)
    lib.him_copy_channels(_p(x), Cn, c0, _p(out), n, 0, n, B, H * W, 0, 0, 0, _stream())
    oh = getattr(x, '_him_onehot', None)
    if oh is not None and c0 == 0 and n >= oh[1]:      # a slice that still starts with the whole one-hot block
        mark_onehot(out, oh[0], oh[1])
    return out


def widen_u8(t):
    """uint8 id map (any shape) on the device -> float32 ids (what every kernel of the path reads)."""
    if not t.is_cuda or t.dtype != torch.uint8:
        raise HimError('widen_u8 needs a uint8 GPU tensor')
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    lib.him_u8_to_f32(t.data_ptr(), out.data_ptr(), t.numel(), _stream())
    return out


def get_masked_image(image, bbox, cls2fill=0.0):
    """Batched, on-device ``get_masked_image`` (reference data/base_dataset.py:342-357): ``image`` (B,C,H,W), ``bbox``
    (B,4) = (wmin, hmin, wmax, hmax) -> (mask (B,1,H,W), mask*image, (1-mask)*image + mask*cls2fill)."""
    image = image.contiguous()
    bbox = bbox.to(device=image.device, dtype=torch.float32).contiguous()
    _chk(image, bbox)
    B, Cn, H, W = image.shape
    if tuple(bbox.shape) != (B, 4):
        raise HimError('get_masked_image: bbox must be (B, 4), got %s' % (tuple(bbox.shape),))
    mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=image.device)
    obj, ctx = torch.empty_like(image), torch.empty_like(image)
    lib.him_masked_image(_p(image), _p(bbox), _p(mask), _p(obj), _p(ctx), B, Cn, H, W, float(cls2fill), _stream())
    return mask, obj, ctx


def tile_embedding(embedding, mask_in):
    """(B,K) embedding, (B,1,H,W) mask -> (B,K,H,W) = embedding[b,k] * mask[b,0,y,x]  (``encode_global_embedding``,
    models/pix2pixHD_condImgColor_model.py:147-160); him_tile_embed writes three channels per launch (the colour path's K)."""
    _chk(embedding, mask_in)
    B, K = embedding.shape
    H, W = mask_in.shape[2], mask_in.shape[3]
    if K % 3:
        raise ValueError('tile_embedding: K = %d (the colour embedding has 3 channels per group)' % K)
    out = torch.empty((B, K, H, W), dtype=torch.float32, device=embedding.device)
    mask_in = mask_in.contiguous()
    for k in range(0, K, 3):
        emb = embedding[:, k:k + 3].contiguous()
        lib.him_tile_embed(_p(emb), _p(mask_in), _p(out), B, K, k, H * W, _stream())
    return out


def masked_mean_color(image, obj_mask, noise=None):
    _chk(image, obj_mask, noise)
    B, _, H, W = image.shape
    emb = torch.empty((B, 3), dtype=torch.float32, device=image.device)
    lib.him_masked_mean(_p(image), _p(obj_mask), _p(noise), _p(emb), B, H * W, _stream())
    return emb


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        a, b = a.contiguous(), b.contiguous()
        _chk(a, b)
        if a.shape != b.shape:
            raise HimError('l1: shape mismatch %s vs %s' % (tuple(a.shape), tuple(b.shape)))
        out = torch.empty((), dtype=torch.float32, device=a.device)
        nb = lib.him_reduce_ws(a.numel())
        ws = _ws(nb, a)
        lib.him_l1_mean_fwd(_p(a), _p(b), a.numel(), _p(out), _p(ws), nb, _stream())
        ctx.a, ctx.b = a, b
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        a, b = ctx.a, ctx.b
        g = g.contiguous()
        da = torch.empty_like(a)
        lib.him_l1_mean_bwd(_p(a), _p(b), a.numel(), _p(g), _p(da), 0, _stream())
        return da, None


def l1_mean(a, b):
    """nn.L1Loss()(a, b.detach())."""
    return _L1Mean.apply(a, b.detach())


class _L1WeightedSum(torch.autograd.Function):
    """sum_i w_i * mean|a_i - b_i| as ONE autograd node: n reductions into a vector, one tiny weighted sum; the backward
    scales the incoming gradient by the weights once and runs the n L1 backward kernels.  Replaces the chain of
    ``loss = loss + w * l1(...) * lambda`` scalar kernels (3 launches forward + 2 backward per term)."""

    @staticmethod
    def forward(ctx, wvec, gate, *tensors):
        ctx.set_materialize_grads(False)
        ctx.gate = bool(gate)
        n = len(tensors) // 2
        a_s = [t.contiguous() for t in tensors[:n]]
        b_s = [t.contiguous() for t in tensors[n:]]
        vals = torch.empty(n, dtype=torch.float32, device=wvec.device)
        st = _stream()
        for a, b in zip(a_s, b_s):
            _chk(a, b)
            if a.shape != b.shape:
                raise HimError('l1: shape mismatch %s vs %s' % (tuple(a.shape), tuple(b.shape)))
        # all pairs in one reduction launch + one finishing launch (per-pair sums as in him_l1_mean_fwd, bit for bit)
        ctx.pa = (ctypes.c_void_p * n)(*[_p(a) for a in a_s])
        ctx.pb = (ctypes.c_void_p * n)(*[_p(b) for b in b_s])
        ctx.pn = (ctypes.c_size_t * n)(*[a.numel() for a in a_s])
        nb = lib.him_l1_multi_ws(n)
        ws = _ws(nb, a_s[0])
        lib.him_l1_multi_fwd(ctx.pa, ctx.pb, ctx.pn, n, _p(vals), _p(ws), nb, st)
        ctx.a_s, ctx.b_s, ctx.wvec = a_s, b_s, wvec
        return (vals * wvec).sum()

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * (2 + 2 * len(ctx.a_s))
        gw = (g * ctx.wvec).contiguous()
        st = _stream()
        grads = [torch.empty_like(a) if ctx.needs_input_grad[2 + i] else None for i, a in enumerate(ctx.a_s)]
        n = len(grads)
        pda = (ctypes.c_void_p * n)(*[_p(da) for da in grads])
        lib.him_l1_multi_bwd(ctx.pa, ctx.pb, ctx.pn, n, _p(gw), pda, 2 if ctx.gate else 0, st)
        return (None, None) + tuple(grads) + (None,) * len(ctx.b_s)


_WVEC_CACHE = {}


def l1_weighted_sum(pairs, weights, gate_relu=False):
    """sum_i weights[i] * nn.L1Loss()(a_i, b_i.detach()) for pairs = [(a_i, b_i), ...].  ``gate_relu``: every a_i is the
    output of a ReLU layer built with ``grad_premasked`` -- the gradients sent back carry the (a_i > 0) gate."""
    dev = pairs[0][0].device
    key = (dev, tuple(float(w) for w in weights))
    wvec = _WVEC_CACHE.get(key)
    if wvec is None:
        wvec = _WVEC_CACHE[key] = torch.tensor(key[1], dtype=torch.float32, device=dev)
    return _L1WeightedSum.apply(wvec, bool(gate_relu), *([a for a, _ in pairs] + [b.detach() for _, b in pairs]))


class _LinComb(torch.autograd.Function):
    """scale * sum_i w_i * t_i over one-element tensors as ONE node / ONE launch each way (him_lincomb_*): the scalar loss
    arithmetic of the trainer, which as torch expressions was ~45 one-element ATen kernels per step."""

    @staticmethod
    def forward(ctx, weights, scale, *terms):
        ctx.set_materialize_grads(False)
        n = len(terms)
        ts = [t.contiguous() for t in terms]
        _chk(*ts)
        if any(t.numel() != 1 for t in ts):
            raise HimError('lincomb: one-element tensors only')
        ctx.weights, ctx.scale, ctx.shapes = tuple(float(w) for w in weights), float(scale), [t.shape for t in ts]
        out = torch.empty((), dtype=torch.float32, device=ts[0].device)
        ptrs = (ctypes.c_void_p * n)(*[_p(t) for t in ts])
        ws = (ctypes.c_float * n)(*ctx.weights)
        lib.him_lincomb_fwd(ptrs, ws, n, ctx.scale, _p(out), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.weights)
        if g is None:
            return (None,) * (n + 2)
        g = g.contiguous()
        need = ctx.needs_input_grad[2:]
        outs = [torch.empty(s, dtype=torch.float32, device=g.device) if nd else None for s, nd in zip(ctx.shapes, need)]
        ptrs = (ctypes.c_void_p * n)(*[_p(o) for o in outs])
        ws = (ctypes.c_float * n)(*ctx.weights)
        lib.him_lincomb_bwd(_p(g), ws, n, ctx.scale, ptrs, _stream())
        return (None, None) + tuple(outs)


def lincomb(terms, weights=None, scale=1.0):
    """scale * (w_0 t_0 + w_1 t_1 + ...), left to right in fp32, over one-element tensors (default weights: 1)."""
    terms = list(terms)
    if not SCHED.lincomb:          # the same expression as a chain of one-element torch ops (A/B switch)
        ws = list(weights) if weights is not None else [1.0] * len(terms)
        acc = 0
        for w, t in zip(ws, terms):
            acc = acc + (t if w == 1.0 else t * w)
        return acc if scale == 1.0 else acc * scale
    ws = tuple(weights) if weights is not None else (1.0,) * len(terms)
    # him_lincomb_* take at most 8 terms per launch: longer sums (--num_D > 8) fold 8 at a time, left to right -- the
    # running sum enters the next launch as its first term with weight 1 (x * 1.0 is exact: same rounding as one chain)
    while len(terms) > 8:
        head = _LinComb.apply(ws[:8], 1.0, *terms[:8])
        terms, ws = [head] + terms[8:], (1.0,) + ws[8:]
    return _LinComb.apply(ws, scale, *terms)


class _MSEConst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        x = x.contiguous()
        _chk(x)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        nb = lib.him_reduce_ws(x.numel())
        ws = _ws(nb, x)
        lib.him_mse_const_fwd(_p(x), x.numel(), target, _p(out), _p(ws), nb, _stream())
        ctx.x, ctx.t = x, target
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        x = ctx.x
        g = g.contiguous()
        dx = torch.empty_like(x)
        lib.him_mse_const_bwd(_p(x), x.numel(), ctx.t, _p(g), _p(dx), 0, _stream())
        return dx, None


def mse_const(x, target):
    """nn.MSELoss()(x, full_like(x, target))."""
    return _MSEConst.apply(x, float(target))


# ------------------------------------------------------------------------------------------------
# spectral norm
# ------------------------------------------------------------------------------------------------
class _SNSigma(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, u):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        W2 = W.contiguous().view(W.shape[0], -1)
        u = u.detach().clone()      # the layer overwrites its persistent u in place after this call
        _chk(W2, u)
        rows, cols = W2.shape
        v = torch.empty(cols, dtype=torch.float32, device=W.device)
        u_new = torch.empty((1, rows), dtype=torch.float32, device=W.device)
        sigma = torch.empty((1, 1), dtype=torch.float32, device=W.device)
        nb = lib.him_sn_ws(rows, cols)
        ws = _ws(nb, W)
        lib.him_sn_power_iter_fwd(_p(W2), _p(u), rows, cols, _p(v), _p(u_new), _p(sigma), _p(ws), nb, _stream())
        ctx.W, ctx.u, ctx.v = W2, u, v
        ctx.save_for_backward(u_new, sigma)    # outputs: never as plain ctx attributes (uncollectable cycle)
        ctx.shape = W.shape
        ctx.mark_non_differentiable(u_new)
        return sigma, u_new

    @staticmethod
    def backward(ctx, g, _gu):
        if g is None:
            return None, None
        W2 = ctx.W
        rows, cols = W2.shape
        g = g.contiguous()
        dW = torch.empty_like(W2)
        nb = lib.him_sn_ws(rows, cols)
        ws = _ws(nb, W2)
        u_new, sigma = ctx.saved_tensors
        lib.him_sn_power_iter_bwd(_p(W2), _p(ctx.u), _p(ctx.v), _p(u_new), _p(sigma), _p(g), rows, cols,
                                  _p(dW), 0, _p(ws), nb, _stream())
        return dW.view(ctx.shape), None


def sn_max_singular_value(W, u):
    """(sigma (1,1), u' (1,rows)) of models/sn_utils.py:11-25 with Ip = 1; sigma is differentiable in W through
    BOTH normalisations (nothing detached, as in the reference)."""
    return _SNSigma.apply(W, u)


class _DivScalar(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, sigma):
        ctx.set_materialize_grads(False)   # an undefined gradient stays None: backward returns early
        W = W.contiguous()
        sigma = sigma.contiguous()
        _chk(W, sigma)
        out = torch.empty_like(W)
        lib.him_div_scalar_fwd(_p(W), _p(sigma), _p(out), W.numel(), _stream())
        ctx.W, ctx.sigma = W, sigma
        return out

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return None, None
        W, sigma = ctx.W, ctx.sigma
        dout = dout.contiguous()
        dW = torch.empty_like(W)
        dsig = torch.empty_like(sigma)
        ws = _ws(4096, W)
        lib.him_div_scalar_bwd(_p(W), _p(sigma), _p(dout), _p(dW), _p(dsig), W.numel(), 0, _p(ws), 4096, _stream())
        return dW, dsig


def div_scalar(W, sigma):
    return _DivScalar.apply(W, sigma)
