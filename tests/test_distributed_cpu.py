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
