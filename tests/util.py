"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def report(name, got, ref, rtol, atol_scale=1.0):
    """max-abs error relative to max|ref|; returns (ok, message) with the worst index for debugging."""
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, '%s: shape %s vs %s' % (name, tuple(got.shape), tuple(ref.shape))
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs()
    worst = err.max().item()
    ok = bool(np.isfinite(worst)) and worst <= rtol * scale * atol_scale
    idx = np.unravel_index(int(err.argmax()), err.shape) if err.numel() else ()
    msg = '%s: max|err|=%.3e (rel %.3e of max|ref|=%.3e, tol %.1e) worst@%s got=%.6e ref=%.6e' % (
        name, worst, worst / scale, scale, rtol, idx, got[idx].item() if err.numel() else 0,
        ref[idx].item() if err.numel() else 0)
    return ok, msg


def assert_close(name, got, ref, rtol=5e-5):
    ok, msg = report(name, got, ref, rtol)
    assert ok, msg
    return msg


def load_golden(tag):
    return np.load(os.path.join(GOLDEN, tag + '.npz'), allow_pickle=False)
