"""Public API: ``nfft_adjoint`` / ``nfft_forward`` with the reference's signatures and autograd
behaviour (reference: ``torch_nfft/nfft.py:11-58``).

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
