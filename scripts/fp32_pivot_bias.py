"""Pattern of the fp32 factor's pivot error: diag(L32) against diag of the fp64 Cholesky of the SAME fp32 Gram matrix.
Mean relative difference by position: along the matrix (accumulation length), inside a 128-block, inside a 16-block.
usage: python scripts/fp32_pivot_bias.py [model n]"""
import sys
import torch
sys.path.insert(0, ".")
from discontinuum_amd import _lib
from discontinuum_amd.backend import GPPlan
from tests.test_gpu_stages import make_case

dev = torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "rating"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = 2 if model == "rating" else 3
X, r, noise, theta = make_case(model, d, n, seed=7, perturb=0.1)
p = GPPlan(model, n, d, dtype=torch.float32, device=dev, lookahead=int(sys.argv[3]) if len(sys.argv) > 3 else 2)
p.set_inputs(X.float().to(dev).contiguous())
nz = noise.float().to(dev).contiguous()
p.stage_gram(theta, nz)
G = p.buffer(_lib.BUF_A)[:n, :n].double()
G = torch.tril(G) + torch.tril(G, -1).T
L64 = torch.linalg.cholesky(G)
p.stage_potrf()
torch.cuda.synchronize()
L32 = torch.tril(p.buffer(_lib.BUF_A)[:n, :n]).double()
rel = (torch.diagonal(L32) / torch.diagonal(L64) - 1).cpu()
print(f"{model} n={n}: mean rel pivot(l_kk) error {rel.mean():+.3e}  std {rel.std():.3e}  -> log-det error {2 * torch.log1p(rel).sum():+.4f}")
k = torch.arange(n)
for name, key, nb in (("eighth of the matrix", k * 8 // n, 8), ("16-block inside the 128-block", (k % 128) // 16, 8), ("row inside the 16-block", k % 16, 16)):
    print(f"  by {name}: " + " ".join(f"{rel[key == b].mean():+.2e}" for b in range(nb)))
# off-diagonal: relative Frobenius error of L per 128-panel, and its mean SIGNED error relative to |L64|
E = (L32 - torch.tril(L64))
for b in (0, n // 256, n // 128 - 2):
    s = slice(b * 128, (b + 1) * 128)
    below = slice((b + 1) * 128, n)
    num, den = E[below, s], L64[below, s]
    print(f"  panel {b}: ||dL||/||L|| {num.norm() / den.norm():.3e}   sum(dL*L)/sum(L*L) {(num * den).sum() / (den * den).sum():+.3e}")
# the diagonal-block inverse the kernel hands to trsm: X L - I
T = p.buffer(_lib.BUF_T)[:n, :n].double()
for b in (0, n // 256):
    s = slice(b * 128, (b + 1) * 128)
    F = torch.tril(T[s, s]) @ L32[s, s] - torch.eye(128, dtype=torch.float64, device=dev)
    print(f"  block {b}: ||X L - I||_max {F.abs().max():.3e}  mean diag {torch.diagonal(F).mean():+.3e}")
