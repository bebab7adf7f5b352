"""world_size-2 tests of the batch-sharded wrapper on CPU (gloo).

The sharding / gather logic is device independent; the local transform is injected (``local_op``) and is
the ORACLE here (tests may use it as the checker's compute) -- the product's own local op is the HIP path
and is covered by the gpu tests.  Compared against the unsharded oracle call.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nfft_ref


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_adjoint(x, pos, batch, bandwidth, cutoff, real_output):
    y = nfft_ref.nfft_adjoint(x.numpy(), pos.numpy(), None if batch is None else batch.numpy(), N=bandwidth, m=cutoff,
                              real_output=real_output)
    return torch.from_numpy(np.ascontiguousarray(y)).to(torch.float32 if real_output else torch.complex64)


def _oracle_forward(x, pos, batch, cutoff, real_output):
    y = nfft_ref.nfft_forward(x.numpy(), pos.numpy(), None if batch is None else batch.numpy(), m=cutoff,
                              real_output=real_output)
    return torch.from_numpy(np.ascontiguousarray(y)).to(torch.float32 if real_output else torch.complex64)


def _problem(B, sizes, d=2, N=8, cols=(2,)):
    rng = np.random.default_rng(5)
    n = sum(sizes)
    pos = torch.from_numpy((rng.random((n, d)) - 0.5).astype(np.float32))
    batch = torch.cat([torch.full((s,), i, dtype=torch.long) for i, s in enumerate(sizes)])
    x = torch.from_numpy(rng.standard_normal((n,) + cols).astype(np.float32))
    xh = torch.from_numpy((rng.standard_normal((B,) + (N,) * d + cols)
                           + 1j * rng.standard_normal((B,) + (N,) * d + cols)).astype(np.complex64))
    return pos, batch, x, xh


def _worker(rank, world, port, sizes, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_nfft_amd import distributed as mod  # imports on CPU: only the kernels need a GPU
        B = len(sizes)
        pos, batch, x, xh = _problem(B, sizes)
        ya = mod.nfft_adjoint(x, pos, batch, bandwidth=8, cutoff=2, local_op=_oracle_adjoint)
        ys = mod.nfft_adjoint(x, pos, batch, bandwidth=8, cutoff=2, gather=False, local_op=_oracle_adjoint)
        yf = mod.nfft_forward(xh, pos, batch, cutoff=2, local_op=_oracle_forward)
        yfr = mod.nfft_forward(xh, pos, batch, cutoff=2, real_output=True, gather=False, local_op=_oracle_forward)
        # sharded-input forward: the rank's own slab of the spectrum only (what gather=False adjoint returns)
        b0, b1 = mod.batch_range(B, rank, world)
        yfl = mod.nfft_forward(xh[b0:b1].clone(), pos, batch, cutoff=2, local_op=_oracle_forward, x_is_local=True)
        ypipe = mod.nfft_forward(ys, pos, batch, cutoff=2, gather=False, local_op=_oracle_forward, x_is_local=True)
        try:
            mod.nfft_forward(xh, pos, batch, cutoff=2, local_op=_oracle_forward, x_is_local=True)
            wrong_shape_rejected = B == b1 - b0  # (a world whose rank owns everything: the full spectrum IS the slab)
        except RuntimeError:
            wrong_shape_rejected = True
        torch.save({"ya": ya, "ys": ys, "yf": yf, "yfr": yfr, "yfl": yfl, "ypipe": ypipe,
                    "rejected": wrong_shape_rejected, "layout": mod.shard_layout(batch, world, pos.shape[0]),
                    "range": mod.batch_range(B, rank, world),
                    "bounds": mod.point_bounds(batch, B, world, pos.shape[0])}, out + ".%d" % rank)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [[30, 50, 20, 40], [25, 0, 35], [40]])
def test_sharded_matches_unsharded(tmp_path, sizes):
    world = 2
    port = _free_port()
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, sizes, out), nprocs=world, join=True)
    B = len(sizes)
    if sizes[-1] == 0:
        pytest.skip("last point set must be non-empty (batch[-1] defines B)")
    pos, batch, x, xh = _problem(B, sizes)
    full_a = _oracle_adjoint(x, pos, batch, 8, 2, False)
    full_f = _oracle_forward(xh, pos, batch, 2, False)
    res = [torch.load(out + ".%d" % r) for r in range(world)]
    for r in range(world):
        assert torch.allclose(res[r]["ya"], full_a, atol=1e-5)
        assert torch.allclose(res[r]["yf"], full_f, atol=1e-5)
        b0, b1 = res[r]["range"]
        assert torch.allclose(res[r]["ys"], full_a[b0:b1], atol=1e-5)
        i0, i1 = res[r]["bounds"][r], res[r]["bounds"][r + 1]
        assert torch.allclose(res[r]["yfr"], full_f[i0:i1].real, atol=1e-5)
        assert torch.allclose(res[r]["yfl"], full_f, atol=1e-5)  # sharded-input forward, rows gathered (ragged shards)
        pipe = _oracle_forward(full_a, pos, batch, 2, False)     # adjoint slab -> forward without replicating it
        assert torch.allclose(res[r]["ypipe"], pipe[i0:i1], atol=1e-4)
        assert res[r]["rejected"]
        assert res[r]["layout"] == (B, res[r]["bounds"])
    # shards tile the batch exactly
    assert res[0]["range"][0] == 0 and res[world - 1]["range"][1] == B
    assert res[0]["range"][1] == res[1]["range"][0]


def _worker_single_set(rank, world, port, n, out):
    """batch=None with world > 1: the one point set (and all its points) belongs to the last rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_nfft_amd import distributed as mod
        pos, _, x, xh = _problem(1, [n])
        ya = mod.nfft_adjoint(x, pos, None, bandwidth=8, cutoff=2, local_op=_oracle_adjoint)
        ys = mod.nfft_adjoint(x, pos, None, bandwidth=8, cutoff=2, gather=False, local_op=_oracle_adjoint)
        yf = mod.nfft_forward(xh, pos, None, cutoff=2, local_op=_oracle_forward)
        yfl = mod.nfft_forward(ys, pos, None, cutoff=2, local_op=_oracle_forward, x_is_local=True)
        torch.save({"ya": ya, "ys": ys, "yf": yf, "yfl": yfl, "layout": mod.shard_layout(None, world, n)}, out + ".%d" % rank)
    finally:
        dist.destroy_process_group()


def test_single_point_set_without_batch_vector(tmp_path):
    world, n = 2, 60
    out = str(tmp_path / "res")
    mp.spawn(_worker_single_set, args=(world, _free_port(), n, out), nprocs=world, join=True)
    pos, _, x, xh = _problem(1, [n])
    full_a = _oracle_adjoint(x, pos, None, 8, 2, False)
    full_f = _oracle_forward(xh, pos, None, 2, False)
    res = [torch.load(out + ".%d" % r) for r in range(world)]
    assert float(full_a.abs().max()) > 0
    for r in range(world):
        assert res[r]["layout"] == (1, [0, 0, n])
        assert torch.allclose(res[r]["ya"], full_a, atol=1e-5)  # (was all zeros: set on the last rank, points on the first)
        assert torch.allclose(res[r]["yf"], full_f, atol=1e-5)
        assert res[r]["ys"].shape[0] == (1 if r == world - 1 else 0)
        assert torch.allclose(res[r]["yfl"], _oracle_forward(full_a, pos, None, 2, False), atol=1e-4)


def _worker_local_inputs(rank, world, port, sizes, out):
    """inputs_are_local: every rank passes only its own point sets, numbered from 0."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_nfft_amd import distributed as mod
        B = len(sizes)
        pos, batch, x, xh = _problem(B, sizes)
        b0, b1 = mod.batch_range(B, rank, world)
        i0, i1 = sum(sizes[:b0]), sum(sizes[:b1])
        lpos, lx = pos[i0:i1].clone(), x[i0:i1].clone()
        lbatch = (batch[i0:i1] - b0).clone()
        ys = mod.nfft_adjoint(lx, lpos, lbatch, bandwidth=8, cutoff=2, gather=False, local_op=_oracle_adjoint,
                              inputs_are_local=True, local_batch_size=b1 - b0)
        ya = mod.nfft_adjoint(lx, lpos, lbatch, bandwidth=8, cutoff=2, local_op=_oracle_adjoint, inputs_are_local=True,
                              local_batch_size=b1 - b0)
        yf = mod.nfft_forward(xh[b0:b1].clone(), lpos, lbatch, cutoff=2, local_op=_oracle_forward, inputs_are_local=True)
        yfs = mod.nfft_forward(ys, lpos, lbatch, cutoff=2, gather=False, local_op=_oracle_forward, inputs_are_local=True)
        torch.save({"ys": ys, "ya": ya, "yf": yf, "yfs": yfs, "range": (b0, b1), "rows": (i0, i1)}, out + ".%d" % rank)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [[30, 50, 20, 40], [25, 0, 35], [0, 0, 45, 15]])
def test_shard_local_inputs(tmp_path, sizes):
    world = 2
    out = str(tmp_path / "res")
    mp.spawn(_worker_local_inputs, args=(world, _free_port(), sizes, out), nprocs=world, join=True)
    B = len(sizes)
    pos, batch, x, xh = _problem(B, sizes)
    full_a = _oracle_adjoint(x, pos, batch, 8, 2, False)
    if full_a.shape[0] < B:  # (trailing empty sets cannot be expressed by the batch vector alone)
        full_a = torch.cat([full_a, full_a.new_zeros((B - full_a.shape[0],) + tuple(full_a.shape[1:]))])
    full_f = _oracle_forward(xh, pos, batch, 2, False)
    res = [torch.load(out + ".%d" % r) for r in range(world)]
    for r in range(world):
        b0, b1 = res[r]["range"]
        i0, i1 = res[r]["rows"]
        assert torch.allclose(res[r]["ys"], full_a[b0:b1], atol=1e-5)
        assert torch.allclose(res[r]["ya"], full_a, atol=1e-5)   # slabs gathered in rank order
        assert torch.allclose(res[r]["yf"], full_f, atol=1e-5)   # rows gathered in rank order (ragged)
        pipe = _oracle_forward(full_a, pos, batch, 2, False)
        assert torch.allclose(res[r]["yfs"], pipe[i0:i1], atol=1e-4)
