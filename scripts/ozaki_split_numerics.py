"""Numerics of an error-free-style split of the factorisation's fp64 GEMMs onto int8 matrix instructions (Ozaki scheme I):
how many 7-bit slices do the ACTUAL operands of this engine need before the split product is as accurate as the native
fp64 MFMA chain, and how many int8 GEMMs does that cost?  CPU / numpy only (a measurement script, not product code).

Operands: a loadest-gp matrix (SURVEY 8d synthetic site), its Cholesky factor L and L^-1 in fp64;
  bulk update tile   C = L[i, 0:K] L[j, 0:K]^T   (K = 512: one group of four panels; K = 8192: a whole trailing row)
  lauum tile         S = T[0:K, i]^T T[0:K, j]   (T = L^-1)
Each operand row is scaled by a power of two (its largest exponent) and cut into s signed slices of w = 7 bits; slice
products A_p B_q^T are exact in int32 for K <= 2^17; the truncated scheme keeps p + q <= s - 1: s (s + 1) / 2 int8 GEMMs.
Errors are relative to the abs-sum (|A| |B|^T)_ij -- the scale of a dot product's rounding error -- against a long-double
reference; `native` is the k-ordered fp64 fma chain the MFMA implements.
usage: python scripts/ozaki_split_numerics.py [n]   (default 8448: K = 8192 needs n > 8192 + 128)"""
import sys
import numpy as np
sys.path.insert(0, ".")
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8448
X, y, noise, theta = bench.site("loadest", n, 3, 0)
import torch
from oracle import gp_oracle as orc
Kh = (orc.GRAMS["loadest"](torch.tensor(X), torch.tensor(X), torch.tensor(theta)) + torch.diag(torch.tensor(noise))).numpy()
L = np.linalg.cholesky(Kh)
T = np.linalg.inv(L)
W = 7


def slices(A, s):
    """rows of A -> (e, [A_0 .. A_{s-1}]) with A ~ 2^e sum_p A_p 2^{-W (p + 1)}, A_p integer in [-64, 64]"""
    amax = np.abs(A).max(axis=1, keepdims=True)
    e = np.where(amax > 0, np.floor(np.log2(np.maximum(amax, 1e-300))) + 1, 0.0)
    R = A / np.exp2(e)  # |R| < 1, exact scaling
    out = []
    for _ in range(s):
        R = R * 2.0 ** W
        d = np.rint(R)
        out.append(d.astype(np.int64))
        R = R - d  # exact
    return e, out


def split_product(A, B, s):
    ea, As = slices(A, s)
    eb, Bs = slices(B, s)
    C = np.zeros((A.shape[0], B.shape[0]), dtype=np.longdouble)
    ngemm = 0
    for g in range(s):  # p + q = g: equal weight, summed exactly in integers first (as an int32 accumulator would)
        acc = np.zeros((A.shape[0], B.shape[0]), dtype=np.int64)
        for p in range(g + 1):
            acc += As[p] @ Bs[g - p].T
            ngemm += 1
        assert np.abs(acc).max() < 2 ** 31, "int32 accumulator would overflow"
        C += acc.astype(np.longdouble) * np.longdouble(2.0) ** (-W * (g + 2))
    return (C * np.exp2(ea).astype(np.longdouble) * np.exp2(eb).T.astype(np.longdouble)), ngemm


def native_chain(A, B):
    C = np.zeros((A.shape[0], B.shape[0]))
    for k in range(A.shape[1]):  # one rounding per step (numpy has no fma: product rounding adds <= 1/2 ulp of a term)
        C = C + np.outer(A[:, k], B[:, k])
    return C


def report(name, A, B):
    ref = A.astype(np.longdouble) @ B.astype(np.longdouble).T
    scale = np.abs(A) @ np.abs(B).T
    nat = native_chain(A, B)
    e_nat = np.abs(nat - ref).astype(np.float64) / scale
    print(f"{name}: K = {A.shape[1]}, max|C| {np.abs(ref).max():.3e}, abs-sum scale max {scale.max():.3e}")
    print(f"   native fp64 chain        max err / abs-sum {e_nat.max():.2e}   (rms {np.sqrt((e_nat ** 2).mean()):.2e})")
    for s in (6, 7, 8, 9, 10):
        C, ng = split_product(A, B, s)
        err = np.abs(C - ref).astype(np.float64) / scale
        print(f"   {s:2d} slices, {ng:2d} int8 GEMMs  max err / abs-sum {err.max():.2e}   (rms {np.sqrt((err ** 2).mean()):.2e})"
              f"{'   <= native' if err.max() <= e_nat.max() else ''}")


nb = n // 128
i, j = nb - 1, nb - 2
rows_i, rows_j = slice(i * 128, (i + 1) * 128), slice(j * 128, (j + 1) * 128)
for K in (512, 8192):
    if K + 256 > n:
        continue
    k0 = (j * 128 - K) // 128 * 128  # the K columns left of block column j
    report(f"bulk update tile ({i},{j})", L[rows_i, k0:k0 + K], L[rows_j, k0:k0 + K])
    c_i, c_j = slice(128, 256), slice(0, 128)
    report("lauum tile (1,0)", T[256:256 + K, c_i].T.copy(), T[256:256 + K, c_j].T.copy())
