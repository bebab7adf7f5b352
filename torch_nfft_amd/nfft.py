"""Public API: ``nfft_adjoint`` / ``nfft_forward`` / ``nfft_fastsum`` with the reference's signatures and
autograd behaviour (reference: ``torch_nfft/nfft.py:11-179``).

Adjoint and forward are each other's transposes, so each one's backward is the other
(``nfft.py:22-28, 48-54``); there is no gradient w.r.t. the points.
"""
import torch

from . import ops


class NfftAdjointFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, batch, bandwidth, cutoff, real_output):
        y = ops.nfft_adjoint(pos, x, batch, bandwidth, cutoff, 1 if real_output else 0)
        ctx.save_for_backward(pos, batch)
        ctx.cutoff = cutoff
        ctx.real_input = not x.is_complex()
        return y

    @staticmethod
    def backward(ctx, dy):
        pos, batch = ctx.saved_tensors
        dx = ops.nfft_forward(pos, dy, batch, ctx.cutoff, 1 if ctx.real_input else 0)
        return dx, None, None, None, None, None


def nfft_adjoint(x, pos, batch=None, bandwidth=16, cutoff=3, real_output=False):
    """y[b, k+N/2, ...] ~= sum_{i in point set b} x[i, ...] exp(+2 pi i k.pos[i]),  k in [-N/2, N/2)^d."""
    return NfftAdjointFunction.apply(x, pos, batch, bandwidth, cutoff, real_output)


class NfftForwardFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, batch, cutoff, real_output):
        y = ops.nfft_forward(pos, x, batch, cutoff, 1 if real_output else 0)
        ctx.save_for_backward(pos, batch)
        ctx.cutoff = cutoff
        ctx.bandwidth = x.size(1)
        ctx.real_input = not x.is_complex()
        return y

    @staticmethod
    def backward(ctx, dy):
        pos, batch = ctx.saved_tensors
        dx = ops.nfft_adjoint(pos, dy, batch, ctx.bandwidth, ctx.cutoff, 1 if ctx.real_input else 0)
        return dx, None, None, None, None


def nfft_forward(x, pos, batch=None, cutoff=3, real_output=False):
    """y[i, ...] ~= sum_k x[batch[i], k+N/2, ...] exp(-2 pi i k.pos[i]),  N = x.size(1)."""
    return NfftForwardFunction.apply(x, pos, batch, cutoff, real_output)


class NfftFastsumFunction(torch.autograd.Function):
    """y = K x with the trigonometric kernel matrix K_ij = sum_l coeffs[l] exp(2 pi i l.(source_j - target_i)).
    Linear in x; its transpose swaps sources and targets (reference: nfft.py:62-88)."""

    @staticmethod
    def forward(ctx, x, coeffs, sources, targets, source_batch, target_batch, cutoff):
        # the operator is linear in x only: nothing else may ask for a gradient (reference: nfft.py:67-74)
        for name, t in (("coeffs", coeffs), ("sources", sources), ("targets", targets),
                        ("source_batch", source_batch), ("target_batch", target_batch)):
            if t is not None and t.requires_grad:
                raise AssertionError("nfft_fastsum is differentiable w.r.t. x only, but %s requires grad" % name)
        y = ops.nfft_fastsum(sources, targets, x, coeffs, source_batch, target_batch, cutoff)
        ctx.save_for_backward(sources, targets, coeffs, source_batch, target_batch)
        ctx.cutoff = cutoff
        return y

    @staticmethod
    def backward(ctx, dy):
        sources, targets, coeffs, source_batch, target_batch = ctx.saved_tensors
        dx = ops.nfft_fastsum(targets, sources, dy, coeffs, target_batch, source_batch, ctx.cutoff)
        return dx, None, None, None, None, None, None


def nfft_fastsum(x, coeffs, sources, targets=None, source_batch=None, target_batch=None, /, batch=None, cutoff=3):
    """Fast multiplication with a trigonometric kernel matrix (reference: nfft.py:91-179).

    ``y = nfft_fastsum(x, coeffs, sources[, targets][, source_batch, target_batch][, batch=...][, cutoff=...])``;
    ``coeffs`` is a ``[N]*d`` tensor holding b_l at index ``l + N/2``; a real ``x`` gives a real ``y``."""
    if targets is None:
        targets = sources
        target_batch = source_batch
    if batch is not None:
        source_batch = batch
        target_batch = batch
    return NfftFastsumFunction.apply(x, coeffs, sources, targets, source_batch, target_batch, cutoff)
