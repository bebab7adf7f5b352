"""Batch-sharded multi-GPU NFFT (one process per GPU, RCCL over xGMI through torch.distributed).

The reference is single-GPU (``cudaSetDevice(x.get_device())``, csrc/cuda/core_cuda.cu:165); this is new
API on top of the same operators.  The path shards naturally over the batch axis: every point set of a
batch owns its own grid, FFT and output slab (csrc/cuda/spatial_window_operations.cu:146,
core_cuda.cu:216, 264), and ``batch`` is sorted (docs/source/theory/dataformat.rst:35-37), so the points
of the point sets [b0, b1) are one contiguous row range of ``pos`` / ``x``.  There is no halo and no
reduction; the only exchange is an optional all-gather of the per-rank results when the caller wants
the full result on every rank (``gather=True``).  A call with a single point set is not split (that
would need a distributed FFT): it runs on the rank that owns point set 0 (the last one, ``batch_range``).

Two input contracts.  Replicated (default): ``x``, ``pos``, ``batch`` describe the WHOLE batch on every rank and each rank
picks its row range.  Shard-local (``inputs_are_local=True``): every rank passes only ITS point sets -- what a data-parallel
caller has -- with ``batch`` numbering them from 0; the layout of the whole call (point sets and points per rank) is
exchanged by one small all-gather of two counts, and only when a gathered result is asked for.

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


def _layout(batch, world, n, batch_size=None):
    """``(B, bounds, last_sets)``: see shard_layout; ``last_sets[r]`` = index of the last non-empty point set among
    rank r's points (what the forward transform of a shard needs to size its slab).  One read-back for everything."""
    if batch is None:
        # a single point set: its points live on the rank that owns point set 0 -- the LAST one (batch_range(1, r, world)
        # is (0, 1) for r = world - 1 and empty for every other rank)
        return 1, [0] * world + [n], [0] * world
    bc = batch.contiguous()
    if batch_size is None:
        B_t = bc[-1:] + 1
    else:
        B_t = torch.tensor([batch_size], dtype=bc.dtype, device=bc.device)
    firsts = (torch.arange(world + 1, dtype=bc.dtype, device=bc.device) * B_t) // world  # first point set of every rank
    idx = torch.searchsorted(bc, firsts)
    lasts = bc[(idx[1:] - 1).clamp(min=0)] if n > 0 else idx[1:]
    packed = torch.cat([B_t, idx, lasts]).tolist()
    return int(packed[0]), [int(v) for v in packed[1:world + 2]], [int(v) for v in packed[world + 2:]]


def shard_layout(batch, world, n, batch_size=None):
    """``(B, bounds)``: the number of point sets and the row boundaries of every rank's points (``world + 1`` ints;
    ``batch`` is sorted).  ONE blocking read-back for both -- ``B = batch[-1] + 1`` is needed to place the boundaries,
    so it is formed on the device and comes back with them -- and none at all when there is no batch vector."""
    return _layout(batch, world, n, batch_size)[:2]


def point_bounds(batch, batch_size, world, n):
    """Row boundaries of every rank's points: list of world+1 ints (batch is sorted)."""
    return shard_layout(batch, world, n, batch_size)[1]


def _all_gather_rows(local, sizes, group):
    """Concatenate per-rank tensors that differ only in dim 0 (sizes known on every rank).  The result is allocated
    once and every rank's rows land in place: equal shards by one ``all_gather_into_tensor``, ragged ones by an
    ``all_gather`` onto row views of the result (RCCL takes uneven sizes) or, on backends that do not (gloo), by one
    broadcast per non-empty shard into its view -- no padded staging copy and no ``cat`` (at C4 either would be another
    34 GB)."""
    world = len(sizes)
    if world == 1:
        return local
    tail = tuple(local.shape[1:])
    out = local.new_empty((sum(sizes),) + tail)
    if out.shape[0] == 0:
        return out
    # complex tensors travel as (re, im) pairs: RCCL has no complex dtype
    wire_out = torch.view_as_real(out) if out.is_complex() else out
    wire_in = local.contiguous()
    wire_in = torch.view_as_real(wire_in) if wire_in.is_complex() else wire_in
    if all(s == sizes[0] for s in sizes):
        dist.all_gather_into_tensor(wire_out, wire_in, group=group)
        return out
    views = list(torch.split(wire_out, sizes, dim=0))  # contiguous row ranges of the result
    if dist.get_backend(group) == "nccl" and all(s > 0 for s in sizes):
        dist.all_gather(views, wire_in, group=group)
        return out
    me = dist.get_rank(group)
    for r, view in enumerate(views):
        if sizes[r] == 0:
            continue
        if r == me:
            view.copy_(wire_in)
        dist.broadcast(view, src=r if group is None else dist.get_global_rank(group, r), group=group)
    return out


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


def shard_forward(x, pos, batch, batch_size, rank, world, cutoff=3, real_output=False, local_op=None, bounds=None,
                  x_is_local=False, last_set=None):
    """The rows ``[n_r, *cols]`` of the forward transform that rank ``rank`` of ``world`` owns.  ``x`` is the full
    ``[B, N.., *cols]`` spectrum (only the rank's own slab is read) or, with ``x_is_local``, that slab itself
    ``[B_r, N.., *cols]`` -- what ``shard_adjoint`` / ``nfft_adjoint(gather=False)`` return."""
    op = local_op or _nfft.nfft_forward
    d = pos.shape[1]
    if bounds is None:
        bounds = point_bounds(batch, batch_size, world, pos.shape[0])
    b0, _ = batch_range(batch_size, rank, world)
    i0, i1 = bounds[rank], bounds[rank + 1]
    if i1 > i0:
        lb = None if batch is None else batch[i0:i1] - b0
        # the shard's last point sets may be empty: its slab ends with the last one that holds points
        if lb is None:
            nb = 1
        elif last_set is not None:
            nb = last_set - b0 + 1
        else:
            nb = int(lb[-1].item()) + 1
        slab = x[:nb] if x_is_local else x[b0:b0 + nb]
        return op(slab, pos[i0:i1], lb, cutoff=cutoff, real_output=real_output)
    return x.new_zeros((0,) + tuple(x.shape[1 + d:]), dtype=torch.float32 if real_output else torch.complex64)


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _local_counts(n_sets, n_points, device, group):
    """(point sets, points) of every rank: one all-gather of two integers per rank."""
    rank, world = _world(group)
    if world == 1:
        return [n_sets], [n_points]
    mine = torch.tensor([n_sets, n_points], dtype=torch.int64, device=device)
    allc = torch.empty((2 * world,), dtype=torch.int64, device=device)  # (flat: gloo takes no other output shape)
    dist.all_gather_into_tensor(allc, mine, group=group)
    allc = allc.view(world, 2).tolist()
    return [int(c[0]) for c in allc], [int(c[1]) for c in allc]


def _local_sets(batch, n, local_batch_size):
    """Number of point sets of a shard-local call: given, or ``batch[-1] + 1`` (one blocking read), 1 without a batch
    vector, 0 for a rank without points."""
    if local_batch_size is not None:
        return int(local_batch_size)
    if n == 0:
        return 0
    return 1 if batch is None else int(batch[-1].item()) + 1


def nfft_adjoint(x, pos, batch=None, bandwidth=16, cutoff=3, real_output=False, group=None, gather=True,
                 local_op=None, inputs_are_local=False, local_batch_size=None):
    """Sharded ``nfft_adjoint``.  ``x``, ``pos``, ``batch`` describe the WHOLE batch and are present on every
    rank; each rank transforms its own point sets.  Returns the full ``[B, N.., *cols]`` spectrum on every
    rank (``gather=True``, one all-gather along dim 0) or this rank's ``[B_r, N.., *cols]`` slab.

    ``inputs_are_local=True``: ``x``, ``pos``, ``batch`` hold THIS rank's point sets only (``batch`` counts them from 0;
    ``local_batch_size`` states their number when the last ones may be empty).  The slab of rank r follows those of the
    ranks before it in the gathered spectrum."""
    rank, world = _world(group)
    if inputs_are_local:
        op = local_op or _nfft.nfft_adjoint
        n_r = pos.shape[0]
        B_r = _local_sets(batch, n_r, local_batch_size)
        if n_r > 0:
            y = op(x, pos, batch, bandwidth=bandwidth, cutoff=cutoff, real_output=real_output)
            if y.shape[0] != B_r:  # trailing empty point sets
                pad = y.new_zeros((B_r,) + tuple(y.shape[1:]))
                pad[:y.shape[0]] = y
                y = pad
        else:
            y = x.new_zeros((B_r,) + (bandwidth,) * pos.shape[1] + tuple(x.shape[1:]),
                            dtype=torch.float32 if real_output else torch.complex64)
        if not gather or world == 1:
            return y
        sets, _ = _local_counts(B_r, n_r, y.device, group)
        return _all_gather_rows(y, sets, group)
    B, bounds = shard_layout(batch, world, pos.shape[0])
    y = shard_adjoint(x, pos, batch, B, rank, world, bandwidth, cutoff, real_output, local_op, bounds)
    if not gather or world == 1:
        return y
    sizes = [batch_range(B, r, world)[1] - batch_range(B, r, world)[0] for r in range(world)]
    return _all_gather_rows(y, sizes, group)


def nfft_forward(x, pos, batch=None, cutoff=3, real_output=False, group=None, gather=True, local_op=None,
                 x_is_local=False, inputs_are_local=False):
    """Sharded ``nfft_forward``.  ``x`` is the full ``[B, N.., *cols]`` spectrum (each rank only reads its own
    slab) or, with ``x_is_local=True``, this rank's slab ``[B_r, N.., *cols]`` -- the output of
    ``nfft_adjoint(..., gather=False)``, so an adjoint -> (spectral work) -> forward pipeline never replicates the
    spectra (34.4 GB per rank at C4).  Returns all ``[n, *cols]`` rows on every rank (``gather=True``) or this rank's
    rows.

    ``inputs_are_local=True``: ``pos``, ``batch`` hold this rank's points only (``batch`` counts its point sets from 0) and
    ``x`` is its slab ``[B_r, N.., *cols]``; the gathered result lists the ranks' rows one rank after the other."""
    rank, world = _world(group)
    if inputs_are_local:
        op = local_op or _nfft.nfft_forward
        n_r, d = pos.shape[0], pos.shape[1]
        if n_r > 0:
            nb = 1 if batch is None else int(batch[-1].item()) + 1
            if x.shape[0] < nb:
                raise RuntimeError("Input mismatch")
            y = op(x[:nb], pos, batch, cutoff=cutoff, real_output=real_output)
        else:
            y = x.new_zeros((0,) + tuple(x.shape[1 + d:]), dtype=torch.float32 if real_output else torch.complex64)
        if not gather or world == 1:
            return y
        _, points = _local_counts(x.shape[0], n_r, y.device, group)
        return _all_gather_rows(y, points, group)
    B, bounds, lasts = _layout(batch, world, pos.shape[0])
    b0, b1 = batch_range(B, rank, world)
    if x.shape[0] != (b1 - b0 if x_is_local else B):
        raise RuntimeError("Input mismatch")
    y = shard_forward(x, pos, batch, B, rank, world, cutoff, real_output, local_op, bounds, x_is_local, lasts[rank])
    if not gather or world == 1:
        return y
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    return _all_gather_rows(y, sizes, group)
