// dgp_gemm.h -- LDS-tiled MFMA tile-GEMM core shared by the Cholesky / inverse / predict kernels.
//
// One workgroup (256 threads = 4 waves, arranged 2 x 2) accumulates a BM x BN output tile
//     acc[i][j] += sum_k  opA(i, k) * opB(j, k)
// over `ktiles` k-tiles of depth 16.  Each operand is addressed through a base pointer to the
// tile's (row 0, k 0) element and is either
//     KC  ("k-contiguous"):  op(i, k) = p[i * ld + k]     (a row-major panel, rows = i)
//     IC  ("i-contiguous"):  op(i, k) = p[k * ld + i]     (a row-major matrix read transposed)
// so NT / NN / TN products are the same routine.  Roofline: MFMA (fp64 16x16x4 is 64 cycles per
// 2048 flop per SIMD); the LDS image is laid out so every ds_read_b64 fragment read is
// bank-conflict free:  KC -> [row][17] (odd stride), IC -> [k][rows+16] (stride = 128 B mod 256 B).
#pragma once
#include <type_traits>
#include "dgp_common.h"

namespace dgp {

template <typename T, bool KC, int ROWS, int BK_ = 16>
struct OperandTile {
  // depth of a k-tile.  16 everywhere in the product: 32 (half as many barrier episodes per flop) measured +3 % stand-alone
  // on row-contiguous fp64 operands and on fp32 (scripts/gemm_occ.hip: 68.9 -> 70.9, 133.7 -> 137.6 TFLOP/s), -8 % on
  // k-contiguous fp64 ones, and nothing in situ (lauum 89.4 -> 90.1 ms per 32 sites)
  static constexpr int BK = BK_;
  static constexpr int EPT = ROWS * BK / 256;  // elements staged per thread per k-tile (8 or 4)
  static constexpr int STRIDE = KC ? (BK + 1) : (ROWS + 16);
  static constexpr int ELEMS = KC ? ROWS * (BK + 1) : BK * (ROWS + 16);
  static constexpr int VN = Vec16<T>::N;
  static constexpr int NV = EPT / VN;
  using vec_t = typename Vec16<T>::type;
  static_assert(EPT % VN == 0, "staging vector width");

  // Staging map: vector v (VN consecutive elements = 16 bytes) of thread t starts at (r, c) = coords(t, v).
  //   IC: lane l of a load instruction reads bytes [16 l, 16 l + 16) of one k-row, i.e. a wave covers
  //       whole contiguous lines (measured 60.7 -> 68.5 TFLOP/s fp64 against a strided map);
  //   KC: a tile row is only 16 k's (128 B fp64): each thread reads EPT consecutive k's of one row
  //       (measured 68.8 vs 65.8 TFLOP/s for the line-interleaved alternative).
  static __device__ __forceinline__ void coords(int t, int v, int& r, int& c) {
    if (KC) {
      r = t / (BK / EPT);                      // tile row
      c = (t % (BK / EPT)) * EPT + v * VN;     // k offset
    } else {
      const int idx = v * 256 + t;             // linear vector index, k-row major
      r = idx / (ROWS / VN);                   // k
      c = (idx % (ROWS / VN)) * VN;            // tile-row offset
    }
  }
  static __device__ __forceinline__ void load(const T* __restrict__ p, long ld, T (&reg)[EPT], int t) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int r, c;
      coords(t, v, r, c);
      const vec_t x = *reinterpret_cast<const vec_t*>(p + (long)r * ld + c);
#pragma unroll
      for (int e = 0; e < VN; ++e) reg[v * VN + e] = x[e];
    }
  }
  static __device__ __forceinline__ void store(T* __restrict__ s, const T (&reg)[EPT], int t) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int r, c;
      coords(t, v, r, c);
      T* dst = s + r * STRIDE + c;
#pragma unroll
      for (int e = 0; e < VN; ++e) dst[e] = reg[v * VN + e];
    }
  }
  // fragment element for the 16-row group starting at tile row `row0`, k-step ks (4 k's each)
  static __device__ __forceinline__ T frag(const T* __restrict__ s, int row0, int ks, int lane) {
    if (KC) return s[(row0 + (lane & 15)) * STRIDE + ks * 4 + (lane >> 4)];
    return s[(ks * 4 + (lane >> 4)) * STRIDE + row0 + (lane & 15)];
  }
};

template <typename T, bool A_KC, bool B_KC, int BM = 128, int BN = 128, int BK = 16>
struct TileGemm {
  using OA = OperandTile<T, A_KC, BM, BK>;
  using OB = OperandTile<T, B_KC, BN, BK>;
  using acc_t = typename Mfma<T>::acc_t;
  static constexpr int MI = BM / 32;  // 16-row MFMA tiles per wave in M
  static constexpr int NI = BN / 32;
  static constexpr int SMEM_ELEMS = OA::ELEMS + OB::ELEMS;

  static __device__ __forceinline__ void zero(acc_t (&acc)[MI][NI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = T(0);
  }

  // one k-tile: LDS image from the staged registers, refill them with k-tile `kt + PF`, then the MFMAs
  template <int PF>
  static __device__ __forceinline__ void ktile(const T* __restrict__& A, long lda, const T* __restrict__& B, long ldb,
                                               long stepA, long stepB, int kt, int ktiles, T* __restrict__ sA,
                                               T* __restrict__ sB, T (&ra)[OA::EPT], T (&rb)[OB::EPT],
                                               acc_t (&acc)[MI][NI], int t, int lane, int wm, int wn) {
    __syncthreads();  // everyone is done reading the previous k-tile's LDS image
    OA::store(sA, ra, t);
    OB::store(sB, rb, t);
    __syncthreads();
    if (kt + PF < ktiles) {  // the global loads of a later k-tile fly under this tile's MFMAs
      OA::load(A + PF * stepA, lda, ra, t);
      OB::load(B + PF * stepB, ldb, rb, t);
    }
    A += stepA;
    B += stepB;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      T fa[MI], fb[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi] = OA::frag(sA, wm + mi * 16, ks, lane);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fb[ni] = OB::frag(sB, wn + ni * 16, ks, lane);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Mfma<T>::mma(fa[mi], fb[ni], acc[mi][ni]);
    }
  }

  // PF = number of k-tiles the global loads run ahead of the MFMAs (register-staged).  PF = 2 costs one more
  // set of staging registers and pays where the operands miss L2 (long, unsynchronised k-ranges: lauum / trtri).
  // REV: the k-tiles are visited from the LAST to the first (inside a k-tile of 16 the order stays ascending) -- for
  // products whose terms DECAY along k (K^^-1 = L^-T L^-1: k starts at the diagonal block, where L^-1 is largest): summed
  // large-to-small in fp32, every late product below half an ulp of the running sum is dropped, a systematic loss
  // (measured: tr(S K^) - n = -109 at n = 16384, all of the fp32 plans' gradient error); small-to-large keeps them.
  template <int PF = 1, bool REV = false>
  static __device__ __forceinline__ void run(const T* __restrict__ A, long lda, const T* __restrict__ B, long ldb,
                                             int ktiles, T* __restrict__ smem, acc_t (&acc)[MI][NI]) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = (w >> 1) * (BM / 2), wn = (w & 1) * (BN / 2);
    T* sA = smem;
    T* sB = smem + OA::ELEMS;
    ktiles = ktiles * 16 / BK;  // callers count k in tiles of 16; every k-range is a multiple of 128
    long stepA = A_KC ? BK : BK * lda;
    long stepB = B_KC ? BK : BK * ldb;
    if (REV) {
      A += (long)(ktiles - 1) * stepA;
      B += (long)(ktiles - 1) * stepB;
      stepA = -stepA;
      stepB = -stepB;
    }
    T ra[PF][OA::EPT], rb[PF][OB::EPT];
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (p < ktiles) {
        OA::load(A + p * stepA, lda, ra[p], t);
        OB::load(B + p * stepB, ldb, rb[p], t);
      }
    for (int kt = 0; kt < ktiles; kt += PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p)
        if (kt + p < ktiles)
          ktile<PF>(A, lda, B, ldb, stepA, stepB, kt + p, ktiles, sA, sB, ra[p], rb[p], acc, t, lane, wm, wn);
    }
  }

  // visit every accumulator element this lane owns: f(tile_row, tile_col, value&)
  template <typename F>
  static __device__ __forceinline__ void foreach (acc_t (&acc)[MI][NI], F f) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = (w >> 1) * (BM / 2), wn = (w & 1) * (BN / 2);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T v = acc[mi][ni][r];
          f(wm + mi * 16 + Mfma<T>::crow(lane, r), wn + ni * 16 + (lane & 15), v);
          acc[mi][ni][r] = v;
        }
  }
};

// How a trailing-update tile C -= P_i P_j^T meets its accumulators.
//   fp64: the accumulators START at -C, so the read of C overlaps the first operand loads and the epilogue is
//         store-only:  C_new = -( -C + P_i P_j^T ).
//   fp32: the products are summed FROM ZERO and C joins once at the end.  Started at -C, every one of the pass's K
//         fmaf steps rounds at the magnitude of C while the terms it adds are far smaller: products below half an ulp
//         of C are dropped outright (a Schur-complement diagonal only ever adds squares, so its pivots come out too
//         LARGE -- measured: log-determinant +0.30 at rating n = 16384, growing linearly along the matrix, and the same
//         sign on every matrix tried), and the rest contributes K roundings at |C| per pass instead of one.
// fp32 keeps C in registers from the start of the tile (the loads fly under the first operand loads, as they do in
// the fp64 form) and subtracts the finished sum from it in the epilogue: `keep` has the accumulators' shape.
// G = the tile map the accumulators follow: a TileGemm, or a TileCore (whose map may be the interleaved one, dgp_gemm_dma.h).
// LATE (fp32 only): C is read in the epilogue instead -- the direct-to-LDS core runs at 168 registers per lane, where a
// second accumulator-sized array only survives the k-loop in scratch memory (measured: bulk update 109 -> 101 TFLOP/s).
// STREAM (round 5; the BULK update only): the tile of C is read once and written once per launch and not touched again until
// the next group's launches, 17 GB of traffic later -- its loads and stores carry the non-temporal hint (`nt`), so that the
// read-modify-write does not displace the operand panels that every tile of a row / column shares from L2.  Measured on the
// tile alone (scripts/syrk_persist.hip, 4095 tiles of the direct-to-LDS core, fp64): K = 512 65.3 -> 69.8 TFLOP/s (store only
// 72.8, load only 73.2, no C traffic 75.1: a plain read FOLLOWED by a plain write of the same lines costs far more than
// either alone), K = 256 53.9 -> 59.2.  A cache-policy hint: the values are bitwise the same.  The chain's own column
// updates keep plain accesses (their tiles are re-read by the next panel's update within microseconds).
template <bool STREAM, typename T>
__device__ __forceinline__ T ld_c(const T* p) {
  if constexpr (STREAM) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool STREAM, typename T>
__device__ __forceinline__ void st_c(T* p, T v) {
  if constexpr (STREAM) __builtin_nontemporal_store(v, p);
  else *p = v;
}
// WIDE (round 5; fp64 tiles on the plain accumulator map): the tile of C moves in 16-BYTE accesses, half as many instructions.
// In the accumulator layout of v_mfma_f64_16x16x4 a lane owns ONE column (lane & 15) of rows (lane >> 4) + 4 r, so its natural
// access is 8 bytes; here the lanes of a pair (2 j, 2 j + 1) each fetch BOTH columns of two of the four rows (the even lane rows
// r = 0, 2, the odd lane r = 1, 3 -- every wave instruction still covers whole 128-byte lines) and swap the halves they do not own
// through a DPP quad permute (VALU, no LDS).  Pure data movement: bitwise the same tile.  Measured on the bulk tile alone
// (scripts/syrk_persist.hip): K = 512 64.0 -> 69.7 TFLOP/s (with the non-temporal hint 70.6), K = 256 52.7 -> 62.2 (64.1).
__device__ __forceinline__ double dpp_swap_pair(double v) {  // the value of lane ^ 1
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// Both directions work IN PLACE on the accumulator registers (elements 2 h, 2 h + 1 of an accumulator are adjacent: a 16-byte
// load lands in them directly, a 16-byte store reads them directly) and in two passes -- all loads, then all swaps; all swaps,
// then all stores -- so that no load result needs a temporary: a first version that swapped per load made the compiler hold 32
// loads' worth of temporaries beside the 128 accumulators (99 spilled registers, bulk update 6 % SLOWER in situ).
// A lane's pair (x0, x1) = columns (2 j, 2 j + 1) of one row: the even lane keeps x0 and owes x1 to its partner, the odd lane
// keeps x1 and owes x0; after the swap  acc[2 h] = even ? x0 : partner's x1,  acc[2 h + 1] = even ? partner's x0 : x1.
template <typename G, bool NEGATE>
__device__ __forceinline__ void tile_pair_swap(typename G::acc_t (&acc)[G::MI][G::NI], int p) {
#pragma unroll
  for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double x0 = acc[mi][ni][2 * h], x1 = acc[mi][ni][2 * h + 1];
        const double got = dpp_swap_pair(p ? x0 : x1);
        acc[mi][ni][2 * h] = NEGATE ? -(p ? got : x0) : (p ? got : x0);
        acc[mi][ni][2 * h + 1] = NEGATE ? -(p ? x1 : got) : (p ? x1 : got);
      }
}
template <typename G, bool STREAM, bool NEGATE>
__device__ __forceinline__ void tile_load_wide(typename G::acc_t (&acc)[G::MI][G::NI], const double* __restrict__ C, long ld) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, p = lane & 1;
  const double* base = C + (long)((w >> 1) * (16 * G::MI) + (lane >> 4) + 4 * p) * ld + (w & 1) * (16 * G::NI) + ((lane & 15) & ~1);
#pragma unroll
  for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const dgp_d2* src = reinterpret_cast<const dgp_d2*>(base + (long)(mi * 16 + 8 * h) * ld + ni * 16);
        const dgp_d2 v = STREAM ? __builtin_nontemporal_load(src) : *src;
        acc[mi][ni][2 * h] = v[0];
        acc[mi][ni][2 * h + 1] = v[1];
      }
  tile_pair_swap<G, NEGATE>(acc, p);
}
// (destroys acc: the tile's last use)
template <typename G, bool STREAM, bool NEGATE>
__device__ __forceinline__ void tile_store_wide(typename G::acc_t (&acc)[G::MI][G::NI], double* __restrict__ C, long ld) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, p = lane & 1;
  double* base = C + (long)((w >> 1) * (16 * G::MI) + (lane >> 4) + 4 * p) * ld + (w & 1) * (16 * G::NI) + ((lane & 15) & ~1);
  // the inverse exchange: the even lane owes acc[2 h + 1] (its column of row r = 2 h + 1, which the odd lane stores), the odd lane
  // acc[2 h]; afterwards (acc[2 h], acc[2 h + 1]) = the two columns of the row this lane stores
#pragma unroll
  for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double a0 = acc[mi][ni][2 * h], a1 = acc[mi][ni][2 * h + 1];
        const double got = dpp_swap_pair(p ? a0 : a1);
        acc[mi][ni][2 * h] = NEGATE ? -(p ? got : a0) : (p ? got : a0);
        acc[mi][ni][2 * h + 1] = NEGATE ? -(p ? a1 : got) : (p ? a1 : got);
      }
#pragma unroll
  for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const dgp_d2 v = {acc[mi][ni][2 * h], acc[mi][ni][2 * h + 1]};
        dgp_d2* dst = reinterpret_cast<dgp_d2*>(base + (long)(mi * 16 + 8 * h) * ld + ni * 16);
        if (STREAM) __builtin_nontemporal_store(v, dst);
        else *dst = v;
      }
}
// The tile of C on the plain accumulator map through SCALAR row bases + one 32-bit lane offset (round 5).  `G::foreach` hands out
// (row, column) pairs and the obvious `C[row * ld + col]` makes the compiler keep a 64-bit address per access: with the 128
// accumulators of an fp64 128 x 128 tile live at the same time that cost the bulk update 20 spilled registers around every tile.
// Here the part of the address that depends on the lane -- (crow(lane, 0) * ld + (lane & 15)) elements -- is ONE register, the
// rest (wave tile, 16-row group, register index) is wave-uniform and lives in scalar registers, so every access is the saddr
// form of global_load / global_store with an immediate for the 16-column group.  Same elements, same values; measured on the
// bulk tile alone (scripts/syrk_persist.hip, fp64): K = 512 64.3 -> 68.2 TFLOP/s, K = 256 53.2 -> 58.1, no spill, 166 registers.
// f(ptr, mi, ni, r): ptr = the address of the element that accumulator [mi][ni][r] of this lane stands for.
template <typename T, typename G, typename P, typename F>
__device__ __forceinline__ void tile_rows_foreach(P* __restrict__ C, long ld, F f) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const unsigned loff = (unsigned)(((long)Mfma<T>::crow(lane, 0) * ld + (lane & 15)) * (long)sizeof(T));
  char* Cw = (char*)(const_cast<typename std::remove_const<P>::type*>(C) + (long)((w >> 1) * (16 * G::MI)) * ld + (w & 1) * (16 * G::NI));
  const int rstep = Mfma<T>::crow(0, 1);  // rows between consecutive accumulator registers (fp64: 4, fp32: 1)
#pragma unroll
  for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      char* rp = Cw + (long)(mi * 16 + rstep * r) * ld * (long)sizeof(T);
#pragma unroll
      for (int ni = 0; ni < G::NI; ++ni) f(reinterpret_cast<P*>(rp + ni * 16 * (int)sizeof(T) + loff), mi, ni, r);
    }
}
template <typename T, typename G, bool LATE = false, bool STREAM = false, bool WIDE = false>
__device__ __forceinline__ void trailing_begin(typename G::acc_t (&acc)[G::MI][G::NI], typename G::acc_t (&keep)[G::MI][G::NI],
                                               const T* __restrict__ C, long ld) {
  if (sizeof(T) == 4) {
    if (!LATE) tile_rows_foreach<T, G>(C, ld, [&](const T* q, int mi, int ni, int r) { keep[mi][ni][r] = ld_c<STREAM>(q); });
    G::zero(acc);
  } else {
    if constexpr (WIDE && sizeof(T) == 8) tile_load_wide<G, STREAM, true>(acc, (const double*)C, ld);
    else tile_rows_foreach<T, G>(C, ld, [&](const T* q, int mi, int ni, int r) { acc[mi][ni][r] = -ld_c<STREAM>(q); });
  }
}
template <typename T, typename G, bool LATE = false, bool STREAM = false, bool WIDE = false>
__device__ __forceinline__ void trailing_end(typename G::acc_t (&acc)[G::MI][G::NI], typename G::acc_t (&keep)[G::MI][G::NI],
                                             T* __restrict__ C, long ld) {
  if (sizeof(T) == 4 && LATE) {
    tile_rows_foreach<T, G>(C, ld, [&](T* q, int mi, int ni, int r) { acc[mi][ni][r] = ld_c<STREAM>(q) - acc[mi][ni][r]; });
    tile_rows_foreach<T, G>(C, ld, [&](T* q, int mi, int ni, int r) { st_c<STREAM>(q, (T)acc[mi][ni][r]); });
  } else if (sizeof(T) == 4) {
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = keep[mi][ni][r] - acc[mi][ni][r];
    tile_rows_foreach<T, G>(C, ld, [&](T* q, int mi, int ni, int r) { st_c<STREAM>(q, (T)acc[mi][ni][r]); });
  } else {
    if constexpr (WIDE && sizeof(T) == 8) tile_store_wide<G, STREAM, true>(acc, (double*)C, ld);
    else tile_rows_foreach<T, G>(C, ld, [&](T* q, int mi, int ni, int r) { st_c<STREAM>(q, T(-acc[mi][ni][r])); });
  }
}

}  // namespace dgp
