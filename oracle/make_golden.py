"""Freeze golden vectors from the reference's own exact transform -- TEST INFRASTRUCTURE ONLY.

Runs ONLY in the build container (where /root/reference is mounted).  It loads
the reference's ground-truth module ``torch_nfft/ndft.py`` stand-alone (pure
torch, CPU; the package ``__init__`` cannot be imported because it loads the
un-built CUDA extension) and stores seeded inputs + the reference's outputs as
small ``.npz`` files under ``tests/golden/``.  Nothing of the reference's source
is copied: the fixtures are data (inputs and expected outputs) only.

Scenarios:
  G1  test/test_adjoint.py shape: d=2, B=3 x 1000 points on the radius-1/4 circle, C=10, N=16
  G2  test/test_forward.py shape: d=2, B=1, n=10, C=1, N=16, real x
  G3  BASELINE config 1: d=1, N=64, n=1000 (adjoint + forward)
  G4  d=3, N=16, n=200, ragged batches 50/120/30, real and complex x
  G5  test/test_grad.py shape: d=2, B=2 x 5 points, C=3, N=16 (adjoint + forward)
  G6  fastsum: d=2, n=200, N=8, coeffs given (ndft_fastsum), real x

Usage:  python -B oracle/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/torch_nfft/ndft.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_ndft", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def circle_points(gen, n, d):
    pos = torch.rand((n, d), generator=gen, dtype=torch.float) - 0.5
    return pos / (4 * torch.linalg.norm(pos, dim=1, keepdim=True))  # test_adjoint.py:25-26


def adjoint_cols(ref, x, pos, batch, N):
    # the reference's tests call ndft_adjoint once per column (test_adjoint.py:38)
    x2 = x.reshape(x.shape[0], -1)
    ys = [ref.ndft_adjoint(x2[:, i], pos, batch, N=N)[..., None] for i in range(x2.shape[1])]
    y = torch.cat(ys, dim=-1)
    return y.reshape(y.shape[:-1] + tuple(x.shape[1:]))


def forward_cols(ref, x, pos, batch, d):
    # one call per trailing column (test_forward.py:40)
    xf = x.reshape(tuple(x.shape[:1 + d]) + (-1,))
    ys = [ref.ndft_forward(xf[..., i], pos, batch)[..., None] for i in range(xf.shape[-1])]
    y = torch.cat(ys, dim=-1)
    return y.reshape((pos.shape[0],) + tuple(x.shape[1 + d:]))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if v is None:
            continue
        out[k] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def main():
    ref = load_ref()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    gen = torch.Generator().manual_seed(20240)

    # G1 -- test_adjoint.py:21-38
    d, B, n, C, N = 2, 3, 1000, 10, 16
    pos = circle_points(gen, n * B, d)
    batch = torch.div(torch.arange(n * B), n, rounding_mode="trunc")
    x = torch.rand((n * B, C), generator=gen, dtype=torch.float)
    save("g1_adjoint_2d_batched", pos=pos, batch=batch, x=x, N=N,
         y_adjoint=adjoint_cols(ref, x, pos, batch, N).to(torch.complex64))

    # G2 -- test_forward.py:21-40
    d, B, n, C, N = 2, 1, 10, 1, 16
    pos = circle_points(gen, n, d)
    xh = torch.rand((B,) + (N,) * d + (C,), generator=gen, dtype=torch.float)
    save("g2_forward_2d", pos=pos, x=xh, N=N, y_forward=forward_cols(ref, xh, pos, None, d))

    # G3 -- BASELINE.json config 1
    d, n, N = 1, 1000, 64
    pos = torch.rand((n, d), generator=gen, dtype=torch.float) - 0.5
    x = torch.rand((n,), generator=gen, dtype=torch.float)
    ya = ref.ndft_adjoint(x, pos, None, N=N)
    xh = torch.randn((1, N), generator=gen, dtype=torch.float) + 1j * torch.randn((1, N), generator=gen)
    xh = xh.to(torch.complex64)
    save("g3_1d_n64", pos=pos, x=x, N=N, y_adjoint=ya, xhat=xh, y_forward=ref.ndft_forward(xh, pos, None))

    # G4 -- 3-D, ragged batches, real + complex
    d, N = 3, 16
    sizes = [50, 120, 30]
    n = sum(sizes)
    pos = torch.rand((n, d), generator=gen, dtype=torch.float) - 0.5
    batch = torch.cat([torch.full((s,), i, dtype=torch.long) for i, s in enumerate(sizes)])
    xr = torch.randn((n, 2), generator=gen, dtype=torch.float)
    xc = (torch.randn((n, 2), generator=gen) + 1j * torch.randn((n, 2), generator=gen)).to(torch.complex64)
    ya_r = adjoint_cols(ref, xr, pos, batch, N)
    ya_c = adjoint_cols(ref, xc, pos, batch, N)
    xh = (torch.randn((3,) + (N,) * d + (2,), generator=gen)
          + 1j * torch.randn((3,) + (N,) * d + (2,), generator=gen)).to(torch.complex64)
    save("g4_3d_ragged", pos=pos, batch=batch, x_real=xr, x_complex=xc, N=N,
         y_adjoint_real=ya_r.to(torch.complex64), y_adjoint_complex=ya_c.to(torch.complex64),
         xhat=xh, y_forward=forward_cols(ref, xh, pos, batch, d))

    # G5 -- test_grad.py shapes
    d, B, n, C, N = 2, 2, 5, 3, 16
    pos = circle_points(gen, n * B, d)
    batch = torch.div(torch.arange(n * B), n, rounding_mode="trunc")
    x = torch.rand((n * B, C), generator=gen, dtype=torch.float)
    xh = torch.rand((B,) + (N,) * d + (C,), generator=gen, dtype=torch.float)
    save("g5_grad_shapes", pos=pos, batch=batch, x=x, xhat=xh, N=N,
         y_adjoint=adjoint_cols(ref, x, pos, batch, N).to(torch.complex64),
         y_forward=forward_cols(ref, xh, pos, batch, d))

    # G6 -- fastsum (ndft_fastsum, ndft.py:48-62) with explicit Gaussian coefficients
    d, n, N, sigma = 2, 200, 8, 0.2
    pos = circle_points(gen, n, d)
    l = torch.arange(-N // 2, N // 2, dtype=torch.float)
    c1 = (np.sqrt(np.pi) * sigma) * torch.exp(-(sigma * np.pi * l) ** 2)
    coeffs = c1[:, None] * c1[None, :]
    x = torch.rand((n, 2), generator=gen, dtype=torch.float)
    ys = torch.cat([ref.ndft_fastsum(x[:, i:i + 1], coeffs, pos, N=N) for i in range(2)], dim=-1)
    save("g6_fastsum_2d", pos=pos, x=x, coeffs=coeffs, N=N, y_fastsum=ys,
         exact_trig=ref.exact_trigonometric_matrix(coeffs, pos),
         exact_gauss=ref.exact_gaussian_matrix(sigma, pos))


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not mounted; golden vectors can only be regenerated in the build container")
    main()
