"""The two tile-GEMM cores of the O(n^3) stages -- register-staged `TileGemm` (csrc/dgp_gemm.h) and direct-to-LDS
`DmaGemm` (csrc/dgp_gemm_dma.h, every 128 x 128 tile of the benchmark shapes) -- through `dgp_debug_tile_gemm`:

* BITWISE equal to each other (same accumulator layout, same order of the k-sum) in all four operand layouts and both
  precisions, for k-ranges that are and are not a multiple of the direct-to-LDS ring (3 chunks of 64 bytes of k:
  24 doubles / 48 floats), down to a single k-tile -- every parity statement about the 128-tile kernels rests on this;
* equal to the dense product in float64 (fp64: 1e-13 of the row/column norms; fp32: 2e-6).

Until round 3 this check lived in scripts/gemm_wave.hip only.  Reference: the GEMMs are inside gpytorch's Cholesky /
solves underneath src/discontinuum/engines/gpytorch.py:350-353."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _tile_gemm(lib, dtype, core, a_kc, b_kc, A, B, k, tm, tn, reverse=0, variant=0):
    from discontinuum_amd import _lib

    Cm = torch.full((128 * tm, 128 * tn), float("nan"), dtype=dtype, device=A.device)
    rc = lib.dgp_debug_tile_gemm(_lib.F64 if dtype == torch.float64 else _lib.F32, core, int(a_kc), int(b_kc),
                                 C.c_void_p(A.data_ptr()), A.stride(0), C.c_void_p(B.data_ptr()), B.stride(0), k,
                                 C.c_void_p(Cm.data_ptr()), Cm.stride(0), tm, tn, reverse, variant,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return Cm


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("k", [16, 48, 112, 128, 400, 1024, 2000])
def test_direct_to_lds_core_is_bitwise_the_register_staged_core(dtype, a_kc, b_kc, k, gpu_device):
    from discontinuum_amd import _lib

    lib = _lib.load()
    tm, tn = 3, 2
    g = torch.Generator().manual_seed(1000 * k + 2 * int(a_kc) + int(b_kc))
    # operands embedded in larger arrays (leading dimension > extent, as in the factorisation); k-contiguous: rows x k,
    # row-contiguous: k x rows
    pad = 8
    A = torch.randn((128 * tm, k + pad) if a_kc else (k, 128 * tm + pad), dtype=torch.float64, generator=g).to(dtype).to(gpu_device)
    B = torch.randn((128 * tn, k + pad) if b_kc else (k, 128 * tn + pad), dtype=torch.float64, generator=g).to(dtype).to(gpu_device)
    opA = (A[:, :k] if a_kc else A[:, :128 * tm].T).double()
    opB = (B[:, :k] if b_kc else B[:, :128 * tn].T).double()
    ref = opA @ opB.T
    scale = opA.norm(dim=1)[:, None] * opB.norm(dim=1)[None, :]
    results = []
    for reverse in (0, 1):  # ascending k, and the k-tiles of 16 from the last to the first (K^^-1 = L^-T L^-1's order)
        c_reg = _tile_gemm(lib, dtype, 0, a_kc, b_kc, A, B, k, tm, tn, reverse)
        c_dma = _tile_gemm(lib, dtype, 1, a_kc, b_kc, A, B, k, tm, tn, reverse)
        assert torch.isfinite(c_reg).all() and torch.isfinite(c_dma).all()
        assert torch.equal(c_reg, c_dma), (reverse, (c_reg - c_dma).abs().max().item())
        err = ((c_dma.double() - ref).abs() / scale).max().item()
        assert err < (1e-13 if dtype == torch.float64 else 2e-6), (reverse, err)
        results.append(c_dma)
    if k == 16:
        assert torch.equal(results[0], results[1])  # one k-tile: the two orders coincide
    elif dtype == torch.float32 and k >= 400:
        assert not torch.equal(results[0], results[1])  # ... and otherwise they are different roundings of the same sums


def test_tile_gemm_rejects_bad_arguments(gpu_device):
    from discontinuum_amd import _lib

    lib = _lib.load()
    A = torch.zeros(128, 24, dtype=torch.float64, device=gpu_device)
    Cm = torch.zeros(128, 128, dtype=torch.float64, device=gpu_device)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    assert lib.dgp_debug_tile_gemm(0, 1, 1, 1, p(A), 24, p(A), 24, 20, p(Cm), 128, 1, 1, 0, 0, None) == -1   # k % 16
    assert lib.dgp_debug_tile_gemm(0, 1, 1, 1, p(A), 23, p(A), 24, 16, p(Cm), 128, 1, 1, 0, 0, None) == -1   # ld not 16-byte
    assert lib.dgp_debug_tile_gemm(0, 2, 1, 1, p(A), 24, p(A), 24, 16, p(Cm), 128, 1, 1, 0, 0, None) == -1   # core
    assert lib.dgp_debug_tile_gemm(0, 1, 1, 1, p(A), 24, p(A), 24, 16, p(Cm), 64, 1, 1, 0, 0, None) == -1    # ldc


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("k", [128, 256, 384, 512, 640, 1024, 2048])
def test_interleaved_map_and_zero_work_skipping_are_bitwise_the_plain_core(dtype, k, gpu_device):
    """The direct-to-LDS core with its 16-row / 16-column groups dealt alternately to the wave rows / columns (variant 1), and
    on top of that map the zero-work skipping of `lauum_kernel` / `trtri_level_kernel` (variants 2..5: the last block of 128
    k's unrolled with a compile-time live set per quarter, csrc/dgp_gemm_dma.h) -- on operands that HAVE the triangular
    structure each mode assumes in the block of 128 k's visited last, in the operand layouts and k directions those kernels
    use, for k-ranges whose length before that block is 0, 1 and 2 mod 3 chunks (the peeled start of the ring).  Everything
    computed must be bitwise the plain core's (variant 0): the skipped products are exact zeros."""
    from discontinuum_amd import _lib

    lib = _lib.load()
    tm = tn = 2
    g = torch.Generator().manual_seed(7 + k)
    rnd = lambda *shape: torch.randn(*shape, dtype=torch.float64, generator=g)  # noqa: E731
    tri = torch.tril(torch.ones(128, 128, dtype=torch.float64))

    def run(variant, a_kc, b_kc, opA, opB, reverse):  # opA, opB: (rows, k) dense
        A = (opA if a_kc else opA.T).contiguous().to(dtype).to(gpu_device)
        B = (opB if b_kc else opB.T).contiguous().to(dtype).to(gpu_device)
        return _tile_gemm(lib, dtype, 1, a_kc, b_kc, A, B, k, tm, tn, reverse, variant)

    # (variant, a_kc, b_kc, reverse, which operand is triangular in the last-visited block, how)
    cases = [
        (2, False, False, 1, "A", "le"),   # lauum: A = T[:, bi] read transposed, k downwards, op(i, kk) = 0 for i > kk in block 0
        (4, True, False, 1, "B", "le"),    # inverse W-step: B = T[:, j] read transposed, k downwards
        (3, True, False, 0, "A", "ge"),    # inverse T-step: A = T[i, :] by rows, k upwards, op(i, kk) = 0 for kk > i in the last block
    ]
    for variant, a_kc, b_kc, reverse, which, how in cases:
        opA, opB = rnd(128 * tm, k), rnd(128 * tn, k)
        blk = slice(0, 128) if reverse else slice(k - 128, k)  # the block of k's visited last
        mask = tri.T if how == "le" else tri                  # op(i, kk): zero for i > kk  /  zero for kk > i
        if which == "A":
            for t in range(tm):
                opA[128 * t:128 * t + 128, blk] *= mask
        else:
            for t in range(tn):
                opB[128 * t:128 * t + 128, blk] *= mask
        ref = run(0, a_kc, b_kc, opA, opB, reverse)
        il = run(1, a_kc, b_kc, opA, opB, reverse)
        sk = run(variant, a_kc, b_kc, opA, opB, reverse)
        assert torch.equal(ref, il), (variant, (ref - il).abs().max().item())
        assert torch.equal(ref, sk), (variant, (ref - sk).abs().max().item())
    # variant 5: a diagonal tile of a symmetric product -- sub-tiles with row group >= column group only
    opA = rnd(128 * tm, k)
    ref = run(0, False, False, opA, opA, 1)
    low = run(5, False, False, opA, opA, 1)
    keep = torch.kron(torch.tril(torch.ones(8, 8)), torch.ones(16, 16)).bool().to(gpu_device)
    for t in range(tm):
        r_, l_ = ref[128 * t:128 * t + 128, 128 * t:128 * t + 128], low[128 * t:128 * t + 128, 128 * t:128 * t + 128]
        assert torch.equal(r_[keep], l_[keep])  # (the other sub-tiles are unspecified: zero, or the product)
