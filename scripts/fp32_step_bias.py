"""Signed error statistics of ONE panel step of the fp32 factorisation, over a batch of small matrices (n = 128: the
diagonal-block kernel alone; n = 256: + trsm + one column update; n = 384 ...).  Reference = fp64 Cholesky of the same
fp32 Gram matrix.  Prints mean SIGNED relative errors (bias) next to the rms."""
import sys
import torch
sys.path.insert(0, ".")
from discontinuum_amd import _lib
from discontinuum_amd.backend import GPPlan
from tests.test_gpu_stages import make_case

dev = torch.device("cuda:0")
model, B = "loadest", 64
for n in (128, 256, 512):
    d = 3
    cases = [make_case(model, d, n, seed=100 + b, perturb=0.1) for b in range(B)]
    X = torch.stack([c[0] for c in cases]).float().to(dev).contiguous()
    nz = torch.stack([c[2] for c in cases]).float().to(dev).contiguous()
    theta = torch.stack([c[3] for c in cases])
    p = GPPlan(model, n, d, dtype=torch.float32, device=dev, lookahead=0, batch=B)
    p.set_inputs(X)
    p.stage_gram(theta, nz)
    torch.cuda.synchronize()
    N = p.N
    ws = p._ws
    # per-site views of A and T
    def site_buf(which, b):
        import ctypes as C
        q, ld = C.c_void_p(), C.c_int64()
        p.lib.dgp_plan_buffer(p._h, which, C.byref(q), C.byref(ld))
        per = int(p.lib.dgp_plan_workspace_bytes(p._h))  # not exact per-site stride; recover from the layout instead
        return q.value
    a0 = site_buf(_lib.BUF_A, 0)
    # site stride: query a 1-site plan's workspace size (layout total, 256-aligned)
    p1 = GPPlan(model, n, d, dtype=torch.float32, device=dev, lookahead=0)
    stride = int(p1.lib.dgp_plan_workspace_bytes(p1._h))
    del p1
    def view(which, b):
        off = site_buf(which, 0) - ws.data_ptr() + b * stride
        return ws[off:off + N * N * 4].view(torch.float32).view(N, N)
    G = torch.stack([view(_lib.BUF_A, b).clone() for b in range(B)]).double()
    G = torch.tril(G) + torch.tril(G, -1).transpose(1, 2)
    L64 = torch.linalg.cholesky(G)
    p.stage_potrf()
    torch.cuda.synchronize()
    L32 = torch.stack([torch.tril(view(_lib.BUF_A, b)) for b in range(B)]).double()
    T32 = torch.stack([torch.tril(view(_lib.BUF_T, b)) for b in range(B)]).double()
    E = L32 - L64
    print(f"n={n}")
    nbk = N // 128
    for k in range(nbk):
        s = slice(k * 128, (k + 1) * 128)
        dg = torch.diagonal(L32[:, s, s], dim1=1, dim2=2) / torch.diagonal(L64[:, s, s], dim1=1, dim2=2) - 1
        blkE, blkL = torch.tril(E[:, s, s], -1), torch.tril(L64[:, s, s], -1)
        line = f"  block {k}: pivots mean {dg.mean():+.2e} rms {dg.pow(2).mean().sqrt():.2e} | in-block L: sum(dL L)/sum(L L) {(blkE * blkL).sum() / (blkL * blkL).sum():+.2e}"
        Xref = torch.linalg.inv(L64[:, s, s])
        dX = torch.tril(T32[:, s, s]) - Xref
        line += f" | X=L_kk^-1: sum(dX X)/sum(X X) {(dX * Xref).sum() / (Xref * Xref).sum():+.2e} rel rms {(dX.norm() / Xref.norm()):.2e}"
        if k + 1 < nbk:
            below = slice((k + 1) * 128, N)
            pe, pl = E[:, below, s], L64[:, below, s]
            line += f" | panel below: sum(dL L)/sum(L L) {(pe * pl).sum() / (pl * pl).sum():+.2e} rel rms {(pe.norm() / pl.norm()):.2e}"
        print(line)
    del p
