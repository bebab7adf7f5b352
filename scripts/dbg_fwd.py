import os, sys, subprocess, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1:
    import torch, torch_nfft_amd as tn
    n, N, m = int(os.environ.get("NPTS", 1000000)), 256, 4
    gen = torch.Generator(device="cuda").manual_seed(11)
    pos = torch.rand((n, 3), generator=gen, device="cuda") - 0.5
    xh = torch.zeros((1, N, N, N), dtype=torch.complex64, device="cuda")
    idx = torch.randint(0, N, (8, 3), generator=gen, device="cuda")
    xh[0, idx[:, 0], idx[:, 1], idx[:, 2]] = torch.randn(8, generator=gen, device="cuda") + 1j * torch.randn(8, generator=gen, device="cuda")
    yf = tn.nfft_forward(xh, pos, None, cutoff=m)
    x = torch.rand((n,), generator=gen, device="cuda")
    y = tn.nfft_adjoint(x, pos, None, bandwidth=N, cutoff=m)
    lhs = complex(torch.sum(y * xh.conj())); rhs = complex(torch.sum(x.to(torch.complex64) * yf.conj()))
    print("adjointness rel", abs(lhs - rhs) / abs(lhs))
    np.save(sys.argv[1], yf.cpu().numpy())
else:
    for mode in ("lds", "mfma"):
        env = dict(os.environ, NFFT_HIP_GATHER=mode)
        subprocess.run([sys.executable, __file__, "/tmp/yf_%s.npy" % mode], env=env, check=True)
    a, b = np.load("/tmp/yf_lds.npy"), np.load("/tmp/yf_mfma.npy")
    d = np.abs(a - b)
    print("rel_l2 between gathers", np.linalg.norm(a - b) / np.linalg.norm(a), "max", d.max() / np.abs(a).max(), "n bad", int((d > 1e-4 * np.abs(a).max()).sum()))
    print("mean ratio b/a", np.mean((b / a).real), "sum a", a.sum(), "sum b", b.sum())
