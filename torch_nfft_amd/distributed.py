"""Batch-sharded multi-GPU NFFT (one process per GPU, RCCL over xGMI through torch.distributed).

The reference is single-GPU (``cudaSetDevice(x.get_device())``, csrc/cuda/core_cuda.cu:165); this is new
API on top of the same operators.  The path shards naturally over the batch axis: every point set of a
batch owns its own grid, FFT and output slab (csrc/cuda/spatial_window_operations.cu:146,
core_cuda.cu:216, 264), and ``batch`` is sorted (docs/source/theory/dataformat.rst:35-37), so the points
of the point sets [b0, b1) are one contiguous row range of ``pos`` / ``x``.  There is no halo and no
reduction; the only exchange is an optional all-gather of the per-rank results when the caller wants
the full result on every rank (``gather=True``).  A call with a single point set is not split (that
would need a distributed FFT): it runs on the rank that owns batch 0.

Rank r of R takes the point sets [floor(r B / R), floor((r+1) B / R)).  ``shard_adjoint`` / ``shard_forward``
compute one rank's share given (rank, world) explicitly -- the sharded calls are these plus the all-gather, and
a single process can run every shard of a simulated world one after the other (tests, one-GPU rehearsal).
"""
import torch
import torch.distributed as dist

from . import nfft as _nfft


def batch_range(batch_size, rank, world):
    """Point sets owned by ``rank``: [b0, b1)."""
    return (rank * batch_size) // world, ((rank + 1) * batch_size) // world


def point_bounds(batch, batch_size, world, n):
    """Row boundaries of every rank's points: list of world+1 ints (batch is sorted)."""
    if batch is None:
        # a single point set lives on rank 0
        return [0] + [n] * world
    firsts = torch.tensor([batch_range(batch_size, r, world)[0] for r in range(world)] + [batch_size],
                          dtype=batch.dtype, device=batch.device)
    return [int(v) for v in torch.searchsorted(batch.contiguous(), firsts).tolist()]


def _batch_size(batch):
    return 1 if batch is None else int(batch[-1].item()) + 1


def _all_gather_rows(local, sizes, group):
    """Concatenate per-rank tensors that differ only in dim 0 (sizes known on every rank)."""
    world = len(sizes)
    if world == 1:
        return local
    smax = max(sizes)
    if smax == 0:
        return local
    tail = tuple(local.shape[1:])
    padded = local
    if local.shape[0] != smax:
        padded = local.new_zeros((smax,) + tail)
        padded[:local.shape[0]] = local
    out = local.new_empty((world * smax,) + tail)
    # complex tensors travel as (re, im) pairs: RCCL has no complex dtype
    if local.is_complex():
        dist.all_gather_into_tensor(torch.view_as_real(out), torch.view_as_real(padded.contiguous()), group=group)
    else:
        dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(s == smax for s in sizes):
        return out
    return torch.cat([out[r * smax:r * smax + sizes[r]] for r in range(world)], dim=0)


def shard_adjoint(x, pos, batch, batch_size, rank, world, bandwidth=16, cutoff=3, real_output=False, local_op=None,
                  bounds=None):
    """The slab ``[B_r, N.., *cols]`` of the adjoint transform that rank ``rank`` of ``world`` owns."""
    op = local_op or _nfft.nfft_adjoint
    if bounds is None:
        bounds = point_bounds(batch, batch_size, world, pos.shape[0])
    b0, b1 = batch_range(batch_size, rank, world)
    i0, i1 = bounds[rank], bounds[rank + 1]
    if b1 > b0:
        lb = None if batch is None else batch[i0:i1] - b0
        if lb is not None and lb.numel() == 0:
            lb = None
        y = op(x[i0:i1], pos[i0:i1], lb, bandwidth=bandwidth, cutoff=cutoff, real_output=real_output)
        if y.shape[0] != b1 - b0:  # trailing empty point sets of the shard
            pad = y.new_zeros((b1 - b0,) + tuple(y.shape[1:]))
            pad[:y.shape[0]] = y
            y = pad
        return y
    d = pos.shape[1]
    return x.new_zeros((0,) + (bandwidth,) * d + tuple(x.shape[1:]),
                       dtype=torch.float32 if real_output else torch.complex64)


def shard_forward(x, pos, batch, batch_size, rank, world, cutoff=3, real_output=False, local_op=None, bounds=None):
    """The rows ``[n_r, *cols]`` of the forward transform that rank ``rank`` of ``world`` owns (``x`` is the full
    ``[B, N.., *cols]`` spectrum; only the rank's own slab is read)."""
    op = local_op or _nfft.nfft_forward
    d = pos.shape[1]
    if bounds is None:
        bounds = point_bounds(batch, batch_size, world, pos.shape[0])
    b0, _ = batch_range(batch_size, rank, world)
    i0, i1 = bounds[rank], bounds[rank + 1]
    if i1 > i0:
        lb = None if batch is None else batch[i0:i1] - b0
        nb = 1 if lb is None else int(lb[-1].item()) + 1  # the shard's last point sets may be empty
        return op(x[b0:b0 + nb], pos[i0:i1], lb, cutoff=cutoff, real_output=real_output)
    return x.new_zeros((0,) + tuple(x.shape[1 + d:]), dtype=torch.float32 if real_output else torch.complex64)


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def nfft_adjoint(x, pos, batch=None, bandwidth=16, cutoff=3, real_output=False, group=None, gather=True,
                 local_op=None):
    """Sharded ``nfft_adjoint``.  ``x``, ``pos``, ``batch`` describe the WHOLE batch and are present on every
    rank; each rank transforms its own point sets.  Returns the full ``[B, N.., *cols]`` spectrum on every
    rank (``gather=True``, one all-gather along dim 0) or this rank's ``[B_r, N.., *cols]`` slab."""
    rank, world = _world(group)
    B = _batch_size(batch)
    y = shard_adjoint(x, pos, batch, B, rank, world, bandwidth, cutoff, real_output, local_op)
    if not gather or world == 1:
        return y
    sizes = [batch_range(B, r, world)[1] - batch_range(B, r, world)[0] for r in range(world)]
    return _all_gather_rows(y, sizes, group)


def nfft_forward(x, pos, batch=None, cutoff=3, real_output=False, group=None, gather=True, local_op=None):
    """Sharded ``nfft_forward``.  ``x`` is the full ``[B, N.., *cols]`` spectrum (each rank only reads its own
    slab); returns all ``[n, *cols]`` rows on every rank (``gather=True``) or this rank's rows."""
    rank, world = _world(group)
    B = _batch_size(batch)
    if x.shape[0] != B:
        raise RuntimeError("Input mismatch")
    bounds = point_bounds(batch, B, world, pos.shape[0])
    y = shard_forward(x, pos, batch, B, rank, world, cutoff, real_output, local_op, bounds)
    if not gather or world == 1:
        return y
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    return _all_gather_rows(y, sizes, group)
