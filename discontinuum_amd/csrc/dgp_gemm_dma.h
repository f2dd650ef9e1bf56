// dgp_gemm_dma.h -- the 128 x 128 tile-GEMM core with operands moved global -> LDS directly, THREE workgroups per CU.
//
// Same contract as TileGemm<T, A_KC, B_KC, 128, 128> (dgp_gemm.h): 256 threads = 4 waves (2 x 2) accumulate
//     acc[i][j] += sum_k opA(i, k) * opB(j, k),   KC: op(i, k) = p[i ld + k],   IC: op(i, k) = p[k ld + i]
// with the SAME accumulator layout and the SAME order of the k-sum: its results are bitwise those of TileGemm.
//
// Why a second core.  The register-staged core (global -> VGPR -> LDS, two barriers per k-tile) needs 208-246 registers in
// fp64, so two waves share a SIMD, and it runs at 68-70 TFLOP/s alone whatever is done to its barriers or its staging
// (scripts/gemm_wave.hip: wave-private staging, software pipelining and direct-to-LDS loads all end at 70 with two waves per
// SIMD).  The same MFMA loop reaches 76-77 of the 78.6 TFLOP/s peak as soon as a THIRD wave fits every SIMD: 512 / 3 -> at most
// 168 registers per lane and 53 KB of LDS per workgroup.  The fp64 accumulators alone are 128 registers, which leaves no room
// for staging registers -- hence:
//   * global_load_lds_dwordx4: the operands never pass through registers, no ds_write is issued;
//   * ring of 3 CHUNKS of 64 bytes of k (8 doubles / 16 floats; A 128 rows + B 128 rows = 16 KB a chunk, 48 KB a workgroup); every
//     wave issues a quarter of a chunk (2 + 2 instructions of 1 KB), ONE barrier per chunk: it publishes chunk c + 1 and frees the
//     slot of chunk c, into which chunk c + 3 is issued at once -- two chunk-times (64 MFMAs per wave) of prefetch distance;
//   * a load instruction writes its 64 x 16 bytes to LDS in lane order, so the LDS image is chosen through WHICH address a lane
//     fetches.  IC operand: whole k-rows of the tile (fp64: one row of 128 doubles per instruction, fp32: two rows of 128 floats),
//     lane l fetching column (its 16-byte unit) ^ 16 (k & 3) -- the XOR spreads the four k's of a fragment read over the banks and
//     only permutes 64- / 128-byte groups inside the row.  KC operand: 16 rows x 64 bytes per instruction (lane l: row l & 15,
//     unit l >> 4), image [16-row group][unit][row][16 bytes].  Every fragment read (ds_read_b64 / _b32, 64 lanes) is bank-conflict
//     free in all four cases;
//   * base pointers stay in scalar registers (global_load saddr form, written as assembly: the compiler otherwise turns them
//     into 64-bit per-lane induction variables and spills).
// fp32 has half the accumulator registers and would fit the old core at three waves too, but gains the same way (lauum 131 ->
// 142-147 TFLOP/s in situ); trailing updates read C in the epilogue there (dgp_gemm.h::trailing_begin, LATE).
// Measured alone (scripts/gemm_wave.hip, fp64 n = 8192, IC/IC): K = 8192 76.7, K = 512 67.5 TFLOP/s (register-staged core 68.2 /
// 55.9).  Roofline: MFMA (v_mfma_f64_16x16x4_f64: 64 cycles per 2048 flop per SIMD; v_mfma_f32_16x16x4_f32: 32).
#pragma once
#include <type_traits>
#include <utility>
#include "dgp_gemm.h"

namespace dgp {

typedef __attribute__((address_space(3))) void* dgp_lds_ptr;

// Zero-work skipping at the triangular END of a k-range (round 4).  K^^-1 = L^-T L^-1 and the two products of the inverse's level
// recursion all finish their k-range in a 128 x 128 DIAGONAL block of a triangular matrix: half of that block's MFMAs multiply
// structural zeros, and a diagonal output tile of the symmetric product needs its lower half only -- 4.5 % + 4.5 % of lauum's
// executed flops at n = 8192, twice that at n = 4096.  A TriMode names the structure (g / h = 16-row group of operand A /
// 16-column group of operand B, 0..7; q = quarter of the last block's 128 k's, 0..3):
//   TRI_ROW_LE: A = a lower-triangular block read transposed, op(i, k) = T[k][i] = 0 for i > k: group g is live in quarter q
//               iff 16 g <= 32 q + 31  <=>  g < 2 (q + 1)
//   TRI_ROW_GE: A = a lower-triangular block by rows, op(i, k) = T[i][k] = 0 for k > i: live iff 16 g + 15 >= 32 q  <=>  g >= 2 q
//   TRI_COL_LE: B read transposed, like TRI_ROW_LE with h
//   TRI_LOWER : the OUTPUT is a diagonal tile of a symmetric product: sub-tiles with g < h are never needed (EVERY chunk; the
//               last block's own zeros are not exploited on top -- diagonal tiles are 3 % of the tiles)
// Skipped products are exact zeros (or unused outputs): the live results are bitwise those of the full computation.
// Two things make it pay.  (1) Skipping only helps when the waves of a workgroup skip EQUALLY (the workgroup advances at its
// slowest wave's pace; round 3 measured the 64-row-halves map at 82.1 -> 81.7 ms): IL = true deals the 16-row / 16-column
// groups alternately to the two wave rows / columns (g = 2 mi + wr, h = 2 ni + wc), and then "g < 2 (q + 1)" reads "mi <= q"
// for BOTH wave rows: per quarter 4, 3, 2, 1 (or 1, 2, 3, 4) of a wave's four row groups are live -- 10 of 16 MFMAs, the same
// 37.5 % a per-wave mask at 16-k granularity would save -- and the live set is a COMPILE-TIME property of the chunk.  (2) No
// control flow may join around the accumulators: the compiler then keeps second copies of the 128 accumulator registers
// and spills thousands of them (every form tried did: per-MFMA branches, a switch over straight-line variants, one branch
// between a plain and a predicated chunk, three unrolled tails for the three ring phases even with the epilogue inside each).
// So the last block is ONE fully unrolled tail; the ring phase is fixed instead by peeling 0..2 chunks off the FRONT of the
// k-range, where the accumulators are still constants, and starting the ring rotated; a kernel with two modes (lauum's
// diagonal / off-diagonal tiles) passes its epilogue in and each path ends with its own copy.  (Also tried, not shipped:
// MFMAs issued from assembly blocks that carry their own scalar branch -- no spills, but non-deterministic garbage whenever a
// block was actually skipped, with the branch targets, operands and wait states all verified in the disassembly.)
enum TriMode { TRI_NONE = 0, TRI_ROW_LE = 1, TRI_ROW_GE = 2, TRI_COL_LE = 3, TRI_LOWER = 4 };

template <typename T, bool A_KC, bool B_KC, int SLOTS_ = 3, bool IL = false>
struct DmaGemm {
  static constexpr bool F64 = sizeof(T) == 8;
  static constexpr int EPU = 16 / (int)sizeof(T);    // elements per 16-byte unit (what one lane fetches)
  static constexpr int KCH = F64 ? 8 : 16;           // k's per chunk: 64 bytes of a k-contiguous row
  static constexpr int KS = KCH / 4;                 // MFMA k-steps per chunk
  static constexpr int SLOTS = SLOTS_;               // ring depth in chunks (2 = 32 KB measures 1-2 % slower alone: scripts/gemm_wave.hip)
  static constexpr int AREA = 128 * KCH;             // elements of one operand's chunk (8 KB)
  static constexpr int SLOT_ELEMS = 2 * AREA;        // A then B
  static constexpr int SMEM_ELEMS = SLOTS * SLOT_ELEMS;  // 48 KB
  using acc_t = typename Mfma<T>::acc_t;

  template <int N>
  static __device__ __forceinline__ void wait_vm() {  // s_waitcnt vmcnt(N) alone
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x0F70);
    asm volatile("" ::: "memory");
  }
  static __device__ __forceinline__ void dma(unsigned voff, const char* base, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0");
  }

  // One operand: what wave w fetches of a chunk (two instructions of 1 KB: bases p0 / p1, the same lane offset) and where it lands.
  //   KC (either precision): instruction = 16 rows x 64 bytes, lane l: row l & 15, unit l >> 4; wave w: rows 32 w .. 32 w + 31.
  //   IC fp64: instruction = one k-row of 128 doubles; wave w: k-rows w and w + 4 (the same swizzle).
  //   IC fp32: instruction = two k-rows of 128 floats (lane l: row l >> 5); wave w: k-rows 2 w, 2 w + 1 and 2 w + 8, 2 w + 9.
  template <bool KC>
  struct Op {
    static __device__ __forceinline__ unsigned voff(int lane, int w, long ld) {
      if (KC) return (unsigned)((long)(lane & 15) * ld * (long)sizeof(T) + 16 * (lane >> 4));
      if (F64) return (unsigned)(((2 * lane) ^ (16 * (w & 3))) * 8);
      const int k = lane >> 5;
      return (unsigned)(((long)k * ld + ((4 * (lane & 31)) ^ (16 * ((2 * w + k) & 3)))) * 4);
    }
    static __device__ __forceinline__ const char* base0(const T* p, int w, long ld) {
      return (const char*)(KC ? p + (long)(32 * w) * ld : p + (long)((F64 ? 1 : 2) * w) * ld);
    }
    static __device__ __forceinline__ long second(long ld) { return (KC ? 16 * ld : (F64 ? 4 : 8) * ld) * (long)sizeof(T); }  // base1 - base0
    static __device__ __forceinline__ long step(long ld) { return KC ? 64 : KCH * ld * (long)sizeof(T); }                   // bytes per chunk
    static constexpr unsigned LDS0 = KC ? 2048 : 1024;  // x w: LDS byte offset of the wave's first instruction in the operand's area
    static constexpr unsigned LDS1 = KC ? 1024 : 4096;  // of the second one, relative to the first
    // fragment address (element index inside the operand's area) of rows wh + 16 mi + (lane & 15), k = 4 ks + (lane >> 4):
    //   KC: q[mi] + 64 ks  (q[mi] = q[0] + MI_STRIDE x group)        IC: q[mi] + 512 ks
    static constexpr int MI_STRIDE = 16 * 4 * EPU;  // a 16-row group is 4 units x 16 rows x EPU elements
    // wsel = the wave's row (operand A) / column (operand B) index 0 / 1; its mi-th 16-row group is g = 4 wsel + mi (two
    // 64-row halves) or, interleaved, g = 2 mi + wsel
    static __device__ __forceinline__ void frag_base(int lane, int wsel, int (&q)[4]) {
      const int fk = lane >> 4, r = lane & 15;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int g = IL ? 2 * mi + wsel : 4 * wsel + mi;
        if (KC) q[mi] = (F64 ? 32 * (fk >> 1) + 2 * r + (fk & 1) : 4 * r + fk) + MI_STRIDE * g;
        else q[mi] = fk * 128 + 64 * (g >> 2) + (((g & 3) ^ fk) & 3) * 16 + r;
      }
    }
    static __device__ __forceinline__ T frag(const T* __restrict__ area, const int (&q)[4], int mi, int ks) {
      return KC ? area[q[mi] + 64 * ks] : area[q[mi] + 512 * ks];
    }
  };
  using OA = Op<A_KC>;
  using OB = Op<B_KC>;

  // A, B: (row 0, k 0) of the operand tiles with leading dimensions lda, ldb; ktiles >= 1 counts k in units of 16 (every caller's
  // k-range is a multiple of 128).  acc comes in initialised (zero, or -C for a trailing update).  All 256 threads call it together.
  // REV: chunks from the last to the first (TileGemm::run's REV: small-to-large summation of products that decay along k; the
  // order inside a k-tile of 16 stays ascending in both cores, so they remain bitwise equal to each other).
  // TRI: the last 128 k's of the range are a triangular diagonal block / the output is a diagonal tile (TriMode above; needs IL).
  struct NoEpilogue {
    __device__ __forceinline__ void operator()() const {}
  };
  struct NoHook {  // hook(round): once per round of SLOTS chunks of the plain steady loop (dgp_common.h: yield_if_asked)
    __device__ __forceinline__ void operator()(int) const {}
  };
  // epi: with TRI the caller's epilogue (its stores of acc) runs INSIDE each of the unrolled tail variants -- merging the
  // 128 accumulator registers of several control-flow paths after the loop is what the register allocator cannot do without
  // second copies (thousands of spills); with its own copy of the epilogue no path ever joins another.
  template <bool REV = false, int TRI = TRI_NONE, typename Epi = NoEpilogue, typename Hook = NoHook>
  static __device__ __forceinline__ void run(const T* __restrict__ A, long lda, const T* __restrict__ B, long ldb, int ktiles,
                                             T* __restrict__ smem, acc_t (&acc)[4][4], Epi epi = Epi(), Hook hook = Hook()) {
    static_assert(TRI == TRI_NONE || IL, "zero-work skipping needs the interleaved group map");
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned va = OA::voff(lane, w, lda), vb = OB::voff(lane, w, ldb);
    const char* a = OA::base0(A, w, lda);
    const char* b = OB::base0(B, w, ldb);
    const long a2 = OA::second(lda), b2 = OB::second(ldb);
    long sa = OA::step(lda), sb = OB::step(ldb);
    // REV walks the k-TILES of 16 downwards, ascending inside each (TileGemm's order).  fp32: a chunk is a k-tile.  fp64: a
    // k-tile is two chunks -- start at the first chunk of the last tile and step +1, -3, +1, -3 ... chunks.
    long sa_odd = sa, sb_odd = sb;  // step after an odd-numbered issue (fp64 REV: back to the previous tile's first chunk)
    if (REV) {
      const long first = (long)(ktiles * 16 / KCH) - (F64 ? 2 : 1);
      a += first * sa;
      b += first * sb;
      if (F64) {
        sa_odd = -3 * sa;
        sb_odd = -3 * sb;
      } else {
        sa = sa_odd = -sa;
        sb = sb_odd = -sb;
      }
    }
    int issued = 0;
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(dgp_lds_ptr)smem);
    const unsigned la = lds + w * OA::LDS0, lb = lds + AREA * (unsigned)sizeof(T) + w * OB::LDS0;
    auto issue = [&](int slot) {  // slot may be a run-time (wave-uniform) value: it only enters the scalar LDS address in m0
      const unsigned o = (unsigned)slot * (SLOT_ELEMS * (unsigned)sizeof(T));
      dma(va, a, la + o);
      dma(va, a + a2, la + o + OA::LDS1);
      dma(vb, b, lb + o);
      dma(vb, b + b2, lb + o + OB::LDS1);
      if (REV && F64) {
        a += (issued & 1) ? sa_odd : sa;
        b += (issued & 1) ? sb_odd : sb;
        ++issued;
      } else {
        a += sa;
        b += sb;
      }
    };
    int qa[4], qb[4];
    OA::frag_base(lane, w >> 1, qa);
    OB::frag_base(lane, w & 1, qb);
    T f[8];
    auto frags = [&](int slot, int ks) {
      const T* d = smem + slot * SLOT_ELEMS;
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = OA::frag(d, qa, i, ks), f[4 + i] = OB::frag(d + AREA, qb, i, ks);
    };
    auto mma = [&]() {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Mfma<T>::mma(f[mi], f[4 + ni], acc[mi][ni]);
    };
    auto chunk = [&](int slot) {  // the MFMAs of one chunk; its first fragments are already in registers
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        mma();
        if (ks + 1 < KS) frags(slot, ks + 1);
      }
    };
    const int C = ktiles * 16 / KCH;  // chunks (every caller has at least 8; any count >= 1 is handled)
    constexpr int TB = 128 / KCH;  // chunks of the triangular last block (16 / 8)
    // one chunk with a COMPILE-TIME set of live MFMAs: Q = the quarter (32 k's) of the last block the chunk lies in, 0..3, or
    // -1 for a chunk before it.  With the interleaved map the live set is the same for every wave (see TriMode above):
    //   TRI_ROW_LE  mi <= Q      TRI_ROW_GE  mi >= Q      TRI_COL_LE  ni <= Q      TRI_LOWER  ni <= mi (every chunk)
    auto chunk_q = [&](int slot, auto qc) {
      constexpr int Q = decltype(qc)::value;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            constexpr bool all = Q < 0;
            const bool on = TRI == TRI_LOWER ? ni <= mi
                          : all ? true
                          : TRI == TRI_ROW_LE ? mi <= Q
                          : TRI == TRI_ROW_GE ? mi >= Q
                          : ni <= Q;
            if (on) acc[mi][ni] = Mfma<T>::mma(f[mi], f[4 + ni], acc[mi][ni]);
          }
        if (ks + 1 < KS) frags(slot, ks + 1);
      }
    };
    // TRI: the chunks before the last block are PRE + 3 n.  The first PRE (0..2) of them are peeled off in front of the steady
    // loop -- while the accumulators are still the constants they were initialised with, so the branches around the peeled
    // steps merge nothing live -- and the ring starts ROTATED (chunk c in slot (c + 3 - PRE) % 3), so that the steady loop and
    // the unrolled tail both see compile-time slots: ONE tail variant, no bubble.
    const int pre = (TRI == TRI_NONE || TRI == TRI_LOWER) ? 0 : (C - TB) % SLOTS;
    const int p0 = pre == 0 ? 0 : SLOTS - pre;  // slot of chunk 0
    __syncthreads();                  // an earlier use of the ring by this workgroup is over
    issue(p0);
    if (C > 1) issue(p0 + 1 >= SLOTS ? p0 + 1 - SLOTS : p0 + 1);
    if (SLOTS > 2 && C > 2) issue(p0 + 2 >= SLOTS ? p0 + 2 - SLOTS : p0 + 2);
    if (SLOTS > 2 && C > 2) wait_vm<8>();
    else if (C > 1) wait_vm<4>();
    else wait_vm<0>();
    __syncthreads();  // chunk 0 is complete
    if constexpr (TRI == TRI_NONE || TRI == TRI_LOWER) {
      // (TRI_LOWER: every chunk has the same compile-time live set -- the plain loops with that chunk; nothing to peel, so it
      // also serves trailing updates, whose accumulators are live from the start)
      auto chunk_u = [&](int slot) {
        if constexpr (TRI == TRI_LOWER) chunk_q(slot, std::integral_constant<int, -1>());
        else chunk(slot);
      };
      frags(0, 0);
      int c0 = 0, round = 0;
      for (; c0 + 2 * SLOTS <= C; c0 += SLOTS, ++round) {  // steady state, branch-free: every chunk of the round has a chunk SLOTS ahead
        hook(round);
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
          chunk_u(u);
          wait_vm<4 * (SLOTS - 2)>();  // this wave's part of chunk c + 1 has landed (chunk c + 2 may be in flight)
          __syncthreads();             // ... everyone's has; and everyone has chunk c in registers or behind it
          issue(u);                    // chunk c + SLOTS into the slot of chunk c
          frags((u + 1) % SLOTS, 0);
        }
      }
      for (; c0 < C; c0 += SLOTS) {
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
          const int c = c0 + u;
          if (c < C) chunk_u(u);
          wait_vm<0>();
          __syncthreads();
          if (c + SLOTS < C) issue(u);
          if (c + 1 < C) frags((u + 1) % SLOTS, 0);
        }
      }
      epi();
    } else {
      static_assert(SLOTS == 3, "the peeled start assumes a ring of three");
      auto step = [&](auto uc, auto qc) {  // one chunk in its steady-state form (a chunk SLOTS ahead exists)
        constexpr int U = decltype(uc)::value;
        chunk_q(U, qc);
        wait_vm<4 * (SLOTS - 2)>();
        __syncthreads();
        issue(U);
        frags((U + 1) % SLOTS, 0);
      };
      const auto full = std::integral_constant<int, -1>();
      // peeled chunks (every k-range here is at least TB + ... chunks long: a chunk SLOTS ahead always exists)
      if (pre == 2) {
        frags(1, 0);
        step(std::integral_constant<int, 1>(), full);
        step(std::integral_constant<int, 2>(), full);
      } else if (pre == 1) {
        frags(2, 0);
        step(std::integral_constant<int, 2>(), full);
      } else {
        frags(0, 0);
      }
      for (int c0 = pre; c0 < C - TB; c0 += SLOTS) {  // steady state: chunk c0 + u in slot u
        step(std::integral_constant<int, 0>(), full);
        step(std::integral_constant<int, 1>(), full);
        step(std::integral_constant<int, 2>(), full);
      }
      // the last block, fully unrolled: slot and live set of every chunk are compile-time constants
      auto one = [&](auto ic) {
        constexpr int I = decltype(ic)::value, U = I % SLOTS;
        constexpr int Qv = I / (TB / 4);          // quarter of the block in VISITING order
        constexpr int QK = REV ? 3 - Qv : Qv;     // ... as a quarter of its k-range
        chunk_q(U, std::integral_constant<int, QK>());
        if constexpr (I + 2 < TB && SLOTS > 2) wait_vm<4 * (SLOTS - 2)>();  // chunk I + 2 exists and was issued: keep it in flight
        else wait_vm<0>();
        __syncthreads();
        if constexpr (I + SLOTS < TB) issue(U);
        if constexpr (I + 1 < TB) frags((U + 1) % SLOTS, 0);
      };
      [&]<int... Is>(std::integer_sequence<int, Is...>) { (one(std::integral_constant<int, Is>()), ...); }
      (std::make_integer_sequence<int, TB>());
      epi();
    }
  }
};

// Which core a kernel's tile uses: the direct-to-LDS one for 128 x 128 tiles, the register-staged one otherwise.
// OCC is the kernel's __launch_bounds__ occupancy, SMEM_ELEMS its LDS array.
// IL (direct-to-LDS core only): the 16-row / 16-column groups of the tile are dealt alternately to the wave rows / columns,
// which balances zero-work skipping (TriSpec); a kernel that sets it must address its accumulators through K::foreach.
// RING (direct-to-LDS core only): chunks in the operand ring, 3 (48 KB per workgroup) or 2 (32 KB: 1-2 % slower alone, but three
// such workgroups leave 64 KB of a CU's LDS free -- one retirement away from the 94.5 KB of the diagonal-block kernel).
template <typename T, bool A_KC, bool B_KC, int BM = 128, int BN = 128, int PF = 1, bool ALLOW_DMA = true, bool IL_ = false, int RING = 3>
struct TileCore {
  using G = TileGemm<T, A_KC, B_KC, BM, BN>;
  static constexpr bool DMA = ALLOW_DMA && BM == 128 && BN == 128;
  static constexpr bool IL = IL_ && DMA;
  using D = DmaGemm<T, A_KC, B_KC, RING, IL>;
  using acc_t = typename G::acc_t;  // (the members trailing_begin / trailing_end need of a tile map: dgp_gemm.h)
  static constexpr int MI = G::MI, NI = G::NI;
  static __device__ __forceinline__ void zero(acc_t (&acc)[MI][NI]) { G::zero(acc); }
  static constexpr int OCC = DMA ? 3 : 2;
  static constexpr int SMEM_ELEMS = DMA ? D::SMEM_ELEMS : G::SMEM_ELEMS;
  template <bool REV = false>
  static __device__ __forceinline__ void run(const T* __restrict__ A, long lda, const T* __restrict__ B, long ldb, int ktiles,
                                             T* __restrict__ smem, typename G::acc_t (&acc)[G::MI][G::NI]) {
    if constexpr (DMA) D::template run<REV>(A, lda, B, ldb, ktiles, smem, acc);
    else G::template run<PF, REV>(A, lda, B, ldb, ktiles, smem, acc);
  }
  // the same with a per-round hook in the direct-to-LDS core's steady loop (the register-staged core ignores it)
  template <typename Hook>
  static __device__ __forceinline__ void run_hooked(const T* __restrict__ A, long lda, const T* __restrict__ B, long ldb, int ktiles,
                                                    T* __restrict__ smem, typename G::acc_t (&acc)[G::MI][G::NI], Hook hook) {
    if constexpr (DMA) D::template run<false, TRI_NONE, typename D::NoEpilogue, Hook>(A, lda, B, ldb, ktiles, smem, acc, typename D::NoEpilogue(), hook);
    else G::template run<PF, false>(A, lda, B, ldb, ktiles, smem, acc);
  }
  // the k-range ends in a triangular diagonal block (TriMode); the register-staged core computes everything (same results).
  // `epi` = the caller's epilogue (its use of acc): it is the LAST thing this call does (see DmaGemm::run).
  template <bool REV, int TRI, typename Epi>
  static __device__ __forceinline__ void run_tri(const T* __restrict__ A, long lda, const T* __restrict__ B, long ldb, int ktiles,
                                                 T* __restrict__ smem, typename G::acc_t (&acc)[G::MI][G::NI], Epi epi) {
    if constexpr (DMA && IL && TRI != TRI_NONE) {
      D::template run<REV, TRI>(A, lda, B, ldb, ktiles, smem, acc, epi);
    } else {
      run<REV>(A, lda, B, ldb, ktiles, smem, acc);
      epi();
    }
  }
  // visit every accumulator element this lane owns: f(tile_row, tile_col, value&) -- the map that matches run()
  template <typename F>
  static __device__ __forceinline__ void foreach (typename G::acc_t (&acc)[G::MI][G::NI], F f) {
    if constexpr (IL) {
      const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            T v = acc[mi][ni][r];
            f(16 * (2 * mi + wr) + Mfma<T>::crow(lane, r), 16 * (2 * ni + wc) + (lane & 15), v);
            acc[mi][ni][r] = v;
          }
    } else {
      G::foreach (acc, f);
    }
  }
};

}  // namespace dgp
