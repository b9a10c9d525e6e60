// Large-tile implicit GEMM for the MATRIX-RATE-bound products of the path (round 4): the deep 3x3 convolutions whose weights do not fit a
// register file (Cin x 9 x Cout with Cout >= 256, or too few tiles for the persistent weight-stationary kernel) and the wide 1x1
// convolutions of yolo_track.yaml at its own scale (K >= 512, N % 128 == 0).  Conv.forward, ultralytics/nn/modules/conv.py:36-38;
// Bottleneck / C2f, nn/modules/block.py:168-188, 271-283; yolo_track.yaml:15-46.
//
// Why another GEMM.  The tiled `gemm_kernel` (gemm.hip) stages both operands global -> VGPR -> ds_write_b128 -> LDS and gives each
// wave a 64 x 32 output tile.  PMC on the K >= 1152 shapes (profiles/r04_a_pmc_tiled_gemm_deep_convs.txt): matrix pipe 40 % busy,
// 3.7 vector instructions per MFMA, LDS active 2.3x the weight-stationary kernel's for fewer MFMAs -- per 64-deep k-step a 128 x 128
// block moves 32 KB through the ~80 B/clk VGPR -> LDS path and 96 KB of fragment reads for 515 clocks of matrix work: LDS-bound.
// This kernel follows the structure MI355X documents as its fast plain-HIP GEMM (cdna_hip_programming.md section 5, "The 256^2 8-phase
// template"), written here from that description for an implicit-GEMM A operand:
//   * block tile 256 rows x (128 * NJ) columns, BK = 64, 8 waves as 2 (rows) x 4 (columns): a wave owns 128 x (32 * NJ) outputs
//     (NJ = 2: 128 accumulator registers), so a k-step needs 24 fragment reads for 64 MFMAs (the tiled kernel: 12 for 16);
//   * BOTH operands arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write; the LDS image is lane-linear
//     per instruction (8 rows x 128 B), the XOR swizzle ((row >> 1) & 7 on the 16-byte chunk: every ds_read_b128 lane group hits 16
//     distinct slots) is applied to the per-lane SOURCE address and to the fragment reads; pixels outside the image, rows past M and
//     dead prefetches are out-of-range buffer offsets, which the DMA turns into zeros (tools/probes/ldsdma_oob.hip);
//   * a k-tile is cut into four HALF-TILES (H0 = A rows of the waves' upper 64 x NJ*32 quadrants, H1 / H2 = the two B halves, H3 = the
//     other A half) and four PHASES of 16 * NJ / 2 MFMAs, one output quadrant each: P1 reads H0 + H1, P2 reads H2, P3 reads H3, P4
//     reads nothing; every phase issues one half-tile of a LATER k-tile into a region whose last read lies >= 2 phases back, so two
//     LDS buffers carry a prefetch distance of 1.5 k-tiles; two counted `vmcnt` per k-tile (never 0 inside the loop), raw `s_barrier`s;
//   * the two wave rows run one barrier apart (waves w and w + 4 share a SIMD: tools/probes/wave_simd.hip), so one half of the block
//     is in its MFMA section while the other reads fragments and issues DMA;
//   * epilogue: BN / bias + activation (+ residual, loaded in the accumulator layout) in registers -> tile image in the output type
//     in LDS (XOR-swizzled rows) -> whole 16-byte row stores.
// Same k order and the same MFMA (16x16x32) per output element as the tiled kernel: results are BIT-IDENTICAL to it (tested).
#include "common.hpp"

#include <type_traits>

namespace moy {

struct GdParams {
  const void* A; int64_t lda; int64_t a_bytes;
  const void* W; int Kpad;
  int M, N, K;
  int stride, Hin, Win, Hout, Wout, Cin;
  uint32_t cin_magic;               // ceil(2^32 / Cin): tap = umulhi(k0, magic)
  const float* scale; const float* shift; int act;
  const void* R; int64_t ldr;
  void* C; int64_t ldc;
  int tiles_n, nblocks;
  int variant;                      // A/B knob MOY_GD_VARIANT: bit 0 = no stagger between the wave halves, bit 1 = no s_setprio around the MFMA sections
  FastDiv fd_tiles_n, fd_hw, fd_wout;
};

__device__ __forceinline__ void gd_dma16(uint32_t voff, __amdgpu_buffer_rsrc_t rs, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds_dst), "s"(rs)
               : "memory");
}
template <int N>
__device__ __forceinline__ void gd_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void gd_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

template <typename T>
__device__ __forceinline__ f32x4 gd_mfma(f32x4 acc, u32x4 w, u32x4 a);
template <>
__device__ __forceinline__ f32x4 gd_mfma<bf16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 gd_mfma<f16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}

// WR wave rows x WC = 8 / WR wave columns; a wave owns 128 rows x 32 * NJ columns.  (WR, NJ) = (2, 2): 256 x 256 block tile, 128 KB of
// LDS; (4, 2): 512 x 128 (the N = 128 convolutions: the same wave tile, all 160 KB of LDS); (2, 1): 256 x 128 (64-column wave tiles:
// measured slower than the tiled kernel, kept for MOY_GEMM_DMA=2 only).
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <typename T>
__device__ __forceinline__ f32x16 gd_mfma32(f32x16 acc, u32x4 w, u32x4 a);
template <>
__device__ __forceinline__ f32x16 gd_mfma32<bf16_t>(f32x16 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 gd_mfma32<f16_t>(f32x16 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}

template <int WR, int NJ>
struct GdGeom {
  static constexpr int WC = 8 / WR;
  static constexpr int BM = 128 * WR, BN = 32 * NJ * WC, BK = 64;
  static constexpr int GA = WR, GB = WC * NJ / 4;        // DMA instructions per thread and half-tile (A half: 64 WR rows, B half: 16 NJ WC rows)
  static_assert(GB >= 1 && GB * 4 == WC * NJ, "a B half-tile is a whole number of 8-wave rounds");
  static constexpr int RA0 = 0, RA1 = WR * 8192, RB0 = WR * 16384, RB1 = RB0 + WC * NJ * 2048;
  static constexpr int BUFB = WR * 16384 + WC * NJ * 4096;   // one k-tile: A + B
  static constexpr int STAGE = 2 * BUFB;
  static constexpr int CTILE = BM * BN * 2;              // output tile image
  static constexpr int LDS = STAGE > CTILE ? STAGE : CTILE;
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <typename T, int WR, int NJ, int KS, int ACT, int MF = 16>
__device__ __forceinline__ void gd_epilogue(const GdParams& p, f32x4 (&acc)[2][4][2][NJ], unsigned char* smem, int m0, int n0, int wr, int wc,
                                            int r, int q, int tid);

// MF = 16: v_mfma_f32_16x16x32 (the shipped form: bit-identical to the tiled kernel); MF = 32: v_mfma_f32_32x32x16 (VERDICT r3 #1b; A/B knob
// MOY_GD_MFMA32=1, NJ = 2 only): the same fragment reads per k-tile (8 + 4 per quadrant), half the MFMA instructions, another summation
// order inside an instruction -- equal to the tiled kernel up to fp32 rounding, not bit for bit.
template <typename T, int WR, int NJ, int KS, int DIAG = 0, int MF = 16>   // DIAG 1: s_memtime stamps per phase section (MOY_GD_DIAG=1; the build's outputs are garbage by design)
__global__ __launch_bounds__(512, 2) void gemm_dma_kernel(const GdParams p) {
  static_assert(MF == 16 || (MF == 32 && NJ == 2), "the 32x32x16 form needs 32-column quadrants");
  using G = GdGeom<WR, NJ>;
  constexpr int WC = G::WC;
  constexpr uint32_t OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int wr = wave / WC, wc = wave % WC;
  const int grp = wave >> 2;                              // waves w and w + 4 share a SIMD: the two halves of the block run one barrier apart

  int bid = blockIdx.x;
  {   // XCD-aware remap (bijective): the blocks of one XCD take consecutive tiles (neighbouring row tiles share halo rows in L2)
    const int nb = p.nblocks, qd = nb >> 3, rm = nb & 7, x = bid & 7;
    bid = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + (bid >> 3);
  }
  const int tile_m = (int)fdiv(bid, p.fd_tiles_n), tile_n = bid - tile_m * p.tiles_n;
  const int m0 = tile_m * G::BM, n0 = tile_n * G::BN;

  const T* __restrict__ Ag = static_cast<const T*>(p.A);
  const T* __restrict__ Wg = static_cast<const T*>(p.W);

  // ---- descriptors (wave-uniform)
  int b0 = 0;
  int64_t a_base = 0;
  if (KS == 1) {
    a_base = (int64_t)m0 * p.lda;
  } else {
    b0 = (int)fdiv(m0, p.fd_hw);
    a_base = (int64_t)b0 * p.Hin * p.Win * p.lda;
  }
  const int64_t a_left = p.a_bytes - a_base * 2;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag + a_base), 0, (uint32_t)(a_left < 0x7fffffffLL ? a_left : 0x7fffffffLL), 0x00020000);
  const int64_t w_base = (int64_t)n0 * p.Kpad;
  const int64_t w_left = ((int64_t)p.N * p.Kpad - w_base) * 2;
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wg + w_base), 0, (uint32_t)(w_left < 0x7fffffffLL ? w_left : 0x7fffffffLL), 0x00020000);

  // ---- DMA geometry of this thread.  A half-tile is 16 pieces of 1 KB (8 rows x 128 B); wave w issues pieces w and w + 8 (A) or
  // w (+ 8 for NJ = 2) (B).  Lane l of piece pc fills row pc*8 + (l >> 3), 16-byte slot l & 7, from source chunk (l & 7) ^ swz(row),
  // swz(row) = (row >> 1) & 7 = ((pc & 1) << 2) | (l >> 4).
  const int srcchunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
  const int subrow = wave * 8 + (lane >> 3);             // row of the piece pair inside its 64-row group
  // A rows: half h (0 / 1), piece j (< WR = the wave row whose rows these are): tile row j*128 + h*64 + subrow
  uint32_t a_off[2][WR];                                 // byte offset of (row, source chunk) at k-tile 0 / tap (0, 0)
  uint32_t a_taps[2][WR];                                // ksize 3: bit t set <=> tap t of the row's output pixel lies inside the image
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < WR; ++j) {
      const int m = m0 + j * 128 + h * 64 + subrow;
      a_off[h][j] = OOB;
      a_taps[h][j] = 0;
      if (m < p.M) {
        if (KS == 1) {
          a_off[h][j] = (uint32_t)(((int64_t)(m - m0) * p.lda + srcchunk * 8) * 2);
          a_taps[h][j] = 1;
        } else {
          const int hw = p.Hout * p.Wout;
          const int b = (int)fdiv(m, p.fd_hw), rem = m - b * hw;
          const int oy = (int)fdiv(rem, p.fd_wout), ox = rem - oy * p.Wout;
          const int iy0 = oy * p.stride - 1, ix0 = ox * p.stride - 1;
          a_off[h][j] = (uint32_t)(((((int64_t)(b - b0) * p.Hin + iy0) * p.Win + ix0) * p.lda + srcchunk * 8) * 2);   // may wrap below 0: fixed by the tap delta
          uint32_t msk = 0;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int iy = iy0 + t / 3, ix = ix0 + t % 3;
            if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) msk |= 1u << t;
          }
          a_taps[h][j] = msk;
        }
      }
    }
  // B rows: half h, piece j (< GB): local row lr = (wave + 8j)*8 + (lane >> 3); tile column = (lr / (16 NJ)) * 32 NJ + h * 16 NJ + lr % (16 NJ)
  uint32_t b_off[2][G::GB];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < G::GB; ++j) {
      const int lr = (wave + 8 * j) * 8 + (lane >> 3);
      const int col = (lr / (16 * NJ)) * (32 * NJ) + h * (16 * NJ) + lr % (16 * NJ);
      b_off[h][j] = (uint32_t)(((int64_t)col * p.Kpad + srcchunk * 8) * 2);
    }
  const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  const uint32_t piece_lds = wave * 1024;
  const int nk = p.K / G::BK;

  // half-tile hh of k-tile kt -> LDS buffer kt & 1.  hh: 0 = A half 0, 1 = B half 0, 2 = B half 1, 3 = A half 1.
  auto issue_half = [&](int kt, int hh) {
    const bool live = kt < nk;                            // dead prefetches keep the vmcnt bookkeeping uniform: every lane out of range
    const uint32_t dst = lds_base + (kt & 1) * G::BUFB + piece_lds;
    if (hh == 0 || hh == 3) {
      const int h = hh == 3;
      uint32_t delta;
      int tap = 0;
      if (KS == 1) {
        delta = (uint32_t)kt * (G::BK * 2);
      } else {
        const int k0 = kt * G::BK;
        tap = (int)__umulhi((unsigned)k0, p.cin_magic);   // k0 / Cin, wave-uniform (Cin % 64 == 0: a k-tile lies inside one tap)
        const int c0 = k0 - tap * p.Cin;
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        delta = (uint32_t)(((ky * p.Win + kx) * (int)p.lda + c0) * 2);
      }
#pragma unroll
      for (int j = 0; j < WR; ++j) {
        const bool ok = live && ((a_taps[h][j] >> tap) & 1u);
        gd_dma16(ok ? a_off[h][j] + delta : OOB, rsA, dst + (h ? G::RA1 : G::RA0) + j * 8192);
      }
    } else {
      const int h = hh == 2;
      const uint32_t delta = (uint32_t)kt * (G::BK * 2);
#pragma unroll
      for (int j = 0; j < G::GB; ++j) gd_dma16(live ? b_off[h][j] + delta : OOB, rsW, dst + (h ? G::RB1 : G::RB0) + j * 8192);
    }
  };

  // ---- fragment addresses: row (.. + r) of a region, chunk (kp*4 + q) ^ ((r >> 1) & 7); kp = 1 flips bit 6 of the byte offset
  const int fsw = ((q ^ (r >> 1)) & 7) << 4;
  const int a_frag = wr * 8192 + r * 128 + fsw;           // + RA(mh) + i*2048, ^ 64 for the second k panel
  const int b_frag = wc * (16 * NJ) * 128 + r * 128 + fsw;   // + RB(nh) + jj*2048

  f32x4 acc[2][4][2][NJ];                                  // MF 16: [m half][16-row block][n half][16-column block]
  f32x16 acc32[2][2][2];                                  // MF 32: [m half][32-row block][n half]
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[mh][i][nh][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[mh][ib][nh][e] = 0.f;

  // ---- prologue: k-tile 0 complete + the first two half-tiles of k-tile 1 (what the steady-state schedule has in flight at P1)
  issue_half(0, 0); issue_half(0, 1); issue_half(0, 2); issue_half(0, 3);
  issue_half(1, 0); issue_half(1, 1);
  gd_wait_vmcnt<G::GA + G::GB>();
  gd_barrier();
  const bool stagger = !(p.variant & 1), prio = !(p.variant & 2);
  if (stagger && grp == 1) gd_barrier();                  // waves 4-7 run one barrier behind waves 0-3

  // MF 32: lane l holds row l & 31 of a 32-row block and the 16-byte chunk 2 ks + (l >> 5) of k-step ks (16 k each): the chunk XOR
  // ((row >> 1) & 7) leaves ks in bits 5-6 of the byte offset
  const int l31 = lane & 31, h32 = lane >> 5;
  const int fsw32 = ((h32 ^ (l31 >> 1)) & 7) << 4;
  const int a_frag32 = wr * 8192 + l31 * 128 + fsw32;     // + RA(mh) + ib*4096, ^ (ks * 32)
  const int b_frag32 = wc * 32 * 128 + l31 * 128 + fsw32; // + RB(nh)

  u32x4 fa[4][2], fb0[NJ][2], fb1[NJ][2];
  auto read_a = [&](const unsigned char* buf, int mh) {
    if constexpr (MF == 16) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
          fa[i][kp] = *reinterpret_cast<const u32x4*>(buf + (mh ? G::RA1 : G::RA0) + i * 2048 + ((a_frag) ^ (kp * 64)));
    } else {
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          fa[ib * 2 + (ks >> 1)][ks & 1] = *reinterpret_cast<const u32x4*>(buf + (mh ? G::RA1 : G::RA0) + ib * 4096 + ((a_frag32) ^ (ks * 32)));
    }
  };
  auto read_b = [&](const unsigned char* buf, int nh, u32x4 (&fb)[NJ][2]) {
    if constexpr (MF == 16) {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
          fb[jj][kp] = *reinterpret_cast<const u32x4*>(buf + (nh ? G::RB1 : G::RB0) + jj * 2048 + ((b_frag) ^ (kp * 64)));
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        fb[ks >> 1][ks & 1] = *reinterpret_cast<const u32x4*>(buf + (nh ? G::RB1 : G::RB0) + ((b_frag32) ^ (ks * 32)));
    }
  };
  auto mfma_quad = [&](int mh, int nh, u32x4 (&fb)[NJ][2]) {
    if (prio) __builtin_amdgcn_s_setprio(1);
    if constexpr (MF == 16) {
#pragma unroll
      for (int kp = 0; kp < 2; ++kp)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) acc[mh][i][nh][jj] = gd_mfma<T>(acc[mh][i][nh][jj], fb[jj][kp], fa[i][kp]);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
          acc32[mh][ib][nh] = gd_mfma32<T>(acc32[mh][ib][nh], fb[ks >> 1][ks & 1], fa[ib * 2 + (ks >> 1)][ks & 1]);
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
  };

  // DIAG: cycles of wave 0 / wave 4 per section, summed over the phases: [0] fragment reads + DMA issue, [1] vmcnt wait, [2] barrier
  // in front of the MFMA section, [3] MFMA section, [4] barrier behind it
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, tprev = 0;
  auto stamp = [&](int i) {
    if constexpr (DIAG == 1) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      ph[i] += t - tprev;
      tprev = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (DIAG == 1) tprev = __builtin_amdgcn_s_memtime();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* buf = smem + (kt & 1) * G::BUFB;
    // P1: quadrant (m0, n0)
    read_a(buf, 0);
    read_b(buf, 0, fb0);
    issue_half(kt + 1, 2);
    stamp(0);
    gd_wait_vmcnt<G::GA + 2 * G::GB>();                   // retires H2, H3 of THIS k-tile (read in P2 / P3): younger = H0, H1, H2 of kt+1
    stamp(1);
    gd_barrier();
    stamp(2);
    mfma_quad(0, 0, fb0);
    stamp(3);
    gd_barrier();
    stamp(4);
    // P2: quadrant (m0, n1)
    read_b(buf, 1, fb1);
    issue_half(kt + 1, 3);
    stamp(0);
    gd_barrier();
    stamp(2);
    mfma_quad(0, 1, fb1);
    stamp(3);
    gd_barrier();
    stamp(4);
    // P3: quadrant (m1, n1)
    read_a(buf, 1);
    issue_half(kt + 2, 0);
    stamp(0);
    gd_barrier();
    stamp(2);
    mfma_quad(1, 1, fb1);
    stamp(3);
    gd_barrier();
    stamp(4);
    // P4: quadrant (m1, n0)
    issue_half(kt + 2, 1);
    stamp(0);
    gd_wait_vmcnt<2 * G::GA + 2 * G::GB>();               // retires H0, H1 of k-tile kt+1 (read in the next P1)
    stamp(1);
    gd_barrier();
    stamp(2);
    mfma_quad(1, 0, fb0);
    stamp(3);
    gd_barrier();
    stamp(4);
  }
  if (stagger && grp == 0) gd_barrier();                  // barrier counts of the two halves meet again
  gd_wait_vmcnt<0>();                                     // dead prefetches still target this block's LDS
  gd_barrier();
  unsigned long long t_loop_end = 0;
  if constexpr (DIAG == 1) t_loop_end = __builtin_amdgcn_s_memtime();

  if constexpr (MF == 32) {
    // the 32 x 32 accumulator of lane l: column (pixel) l & 31, rows (channels) 8 g + 4 (l >> 5) + e for register 4 g + e -- the same
    // "4 consecutive channels of one pixel" cells as the 16 x 16 form, so the epilogue is shared: acc[mh][i][nh][jj] of lane (r, q) is
    // re-addressed through pixel / channel offsets instead of moved between lanes
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x16 c = acc32[mh][ib][nh];
            acc[mh][ib * 2 + (g >> 1)][nh][g & 1] = f32x4{c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
          }
  }
  switch (p.act) {
    case MOY_ACT_SILU: gd_epilogue<T, WR, NJ, KS, MOY_ACT_SILU, MF>(p, acc, smem, m0, n0, wr, wc, r, q, tid); break;
    case MOY_ACT_RELU: gd_epilogue<T, WR, NJ, KS, MOY_ACT_RELU, MF>(p, acc, smem, m0, n0, wr, wc, r, q, tid); break;
    default: gd_epilogue<T, WR, NJ, KS, MOY_ACT_NONE, MF>(p, acc, smem, m0, n0, wr, wc, r, q, tid); break;
  }
  if constexpr (DIAG == 1) {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 8 && (tid == 0 || tid == 256)) {   // (block 8: a full tile away from the first image's border)
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.C) + (tid ? 8 : 0);
      for (int i = 0; i < 5; ++i) dbg[i] = ph[i];
      dbg[5] = t_end - t_loop_end;                        // epilogue
      dbg[6] = (unsigned long long)nk;
    }
  }
}

template <typename T, int WR, int NJ, int KS, int ACT, int MF>
__device__ __forceinline__ void gd_epilogue(const GdParams& p, f32x4 (&acc)[2][4][2][NJ], unsigned char* smem, int m0, int n0, int wr, int wc,
                                            int r, int q, int tid) {
  using G = GdGeom<WR, NJ>;
  constexpr int ROWB = G::BN * 2, CPR = G::BN / 8;        // bytes / 16-byte chunks per tile row
  const T* __restrict__ Rg = static_cast<const T*>(p.R);
  const bool has_sc = p.scale != nullptr, has_sh = p.shift != nullptr;
  const float* scp = has_sc ? p.scale : reinterpret_cast<const float*>(p.W);
  const float* shp = has_sh ? p.shift : reinterpret_cast<const float*>(p.W);
  const int lane = tid & 63, l31 = lane & 31, h32 = lane >> 5;
  // cell (mh, i, nh, jj) of this lane = 4 consecutive output channels nl .. nl+3 of tile row ml
  //   MF 16: row wr*128 + mh*64 + i*16 + r,                     channels wc*32NJ + nh*16NJ + jj*16 + 4q
  //   MF 32: row wr*128 + mh*64 + (i >> 1)*32 + (lane & 31),    channels wc*64 + nh*32 + 8 ((i & 1)*2 + jj) + 4 (lane >> 5)
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int ip = 0; ip < 2; ++ip) {
        const int nl = MF == 16 ? wc * (32 * NJ) + nh * (16 * NJ) + jj * 16 + q * 4 : wc * 64 + nh * 32 + 8 * (ip * 2 + jj) + 4 * h32;
        f32x4 sc = *reinterpret_cast<const f32x4*>(scp + n0 + nl);
        f32x4 sh = *reinterpret_cast<const f32x4*>(shp + n0 + nl);
        if (!has_sc) sc = f32x4{1.f, 1.f, 1.f, 1.f};
        if (!has_sh) sh = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x2 res[2][2];
        if (Rg) {
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
              const int ml = wr * 128 + mh * 64 + (MF == 16 ? (ih * 2 + ip) * 16 + r : ih * 32 + l31);
              res[mh][ih] = *reinterpret_cast<const u32x2*>(Rg + (int64_t)min(m0 + ml, p.M - 1) * p.ldr + n0 + nl);
            }
        }
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int ih = 0; ih < 2; ++ih) {
            const int i = ih * 2 + ip;
            f32x4 v = acc[mh][i][nh][jj] * sc + sh;
            if (ACT == MOY_ACT_SILU) { v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w); }
            else if (ACT == MOY_ACT_RELU) { v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}); }
            if (Rg) {
              // the activation's last multiply must round to fp32 BEFORE the residual is added (the tiled kernel parks the value in
              // LDS in between): without the pin hipcc contracts x * sigmoid(x) + r into one fma and 1 output in 80 000 moves by an ulp
              asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
              v += f32x4{DT<T>::lo(res[mh][ih].x), DT<T>::hi(res[mh][ih].x), DT<T>::lo(res[mh][ih].y), DT<T>::hi(res[mh][ih].y)};
            }
            const int ml = wr * 128 + mh * 64 + (MF == 16 ? i * 16 + r : ih * 32 + l31);
            // tile image [BM][BN] of T, the 16-byte chunk index XORed with the row: the 16 rows of a lane group fall on 16 slots
            unsigned char* cell = smem + ml * ROWB + ((((nl >> 3) ^ (ml & 15)) & (CPR - 1)) << 4) + (nl & 7) * 2;
            *reinterpret_cast<u32x2*>(cell) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
          }
      }
  __syncthreads();
  // whole rows out: thread -> (row, 16-byte chunk); a wave instruction covers 64 / CPR rows of ROWB contiguous bytes
  constexpr int RPP = 512 / CPR, NPASS = G::BM / RPP;
  const int c = tid % CPR, rr0 = tid / CPR;
  T* __restrict__ Cg = static_cast<T*>(p.C);
  u32x4 vv[NPASS];
#pragma unroll
  for (int k = 0; k < NPASS; ++k) {
    const int row = rr0 + k * RPP;
    vv[k] = *reinterpret_cast<const u32x4*>(smem + row * ROWB + (((c ^ (row & 15)) & (CPR - 1)) << 4));
  }
#pragma unroll
  for (int k = 0; k < NPASS; ++k) {
    const int m = m0 + rr0 + k * RPP;
    if (m < p.M) *reinterpret_cast<u32x4*>(Cg + (int64_t)m * p.ldc + n0 + c * 8) = vv[k];
  }
}

template <typename T, int WR, int NJ, int KS, int DIAG = 0, int MF = 16>
static int gd_launch(GdParams& p, hipStream_t st) {
  using G = GdGeom<WR, NJ>;
  if (plan_only(MOY_KERNEL_DMA)) return MOY_OK;          // moy_gemm_query: the dispatch without the launch (and without touching the attribute statics)
#if MOY_DIAG
  if constexpr (DIAG == 0 && MF == 16 && WR == 2 && NJ == 2) {
    static const int mf32 = knob("MOY_GD_MFMA32", 0);                  // MOY_GD_MFMA32=1: v_mfma_f32_32x32x16 (A/B knob; results equal up to fp32 rounding, not bit for bit)
    if (mf32 == 1) return gd_launch<T, WR, NJ, KS, 0, 32>(p, st);
  }
  if constexpr (DIAG == 0 && MF == 16 && std::is_same<T, bf16_t>::value && NJ == 2 && KS == 3) {
    static const int diag = garbage_mode_env("MOY_GD_DIAG");
    if (diag == 1) return gd_launch<T, WR, NJ, KS, 1>(p, st);
  }
#endif
  auto kern = gemm_dma_kernel<T, WR, NJ, KS, DIAG, MF>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return MOY_ELAUNCH;
    attr_set = true;
  }
  const int tiles_m = (p.M + G::BM - 1) / G::BM;
  p.tiles_n = p.N / G::BN;
  p.nblocks = tiles_m * p.tiles_n;
  p.fd_tiles_n = make_fastdiv((uint32_t)p.tiles_n);
  hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(512), G::LDS, st, p);
  return launch_status();
}

template <typename T, int KS>
static int gd_pick(GdParams& p, int form, hipStream_t st) {
#if MOY_DIAG      // (the forms measured slower than the kernels they would replace: A/B knobs of the lab library)
  if (form == 1) return gd_launch<T, 4, 2, KS>(p, st);    // 512 x 128
  if (form == 2) return gd_launch<T, 2, 1, KS>(p, st);    // 256 x 128
#endif
  return form == 0 ? gd_launch<T, 2, 2, KS>(p, st) : MOY_ENOSYS;    // 256 x 256
}

// Eligibility + dispatch; MOY_ENOSYS = not this kernel's shape (moy_gemm falls through to the tiled kernel).
int gemm_dma_try(const moy_gemm_args* a, hipStream_t st) {
  static const int mode = knob("MOY_GEMM_DMA", 1);                    // MOY_GEMM_DMA: 0 = off, 1 = by the heuristic below (default), 2 = whenever the shape fits
  if (!mode) return MOY_ENOSYS;
  if (a->dtype != MOY_BF16 && a->dtype != MOY_F16) return MOY_ENOSYS;
  if (a->A2 || a->a_rows || a->a_mask || a->ln_g || a->out_f32 || a->c_rows_per_batch || a->pre || a->plane_cols || a->dot_n || !a->C || a->run_levels)
    return MOY_ENOSYS;
  if (a->act != MOY_ACT_NONE && a->act != MOY_ACT_SILU && a->act != MOY_ACT_RELU) return MOY_ENOSYS;
  if ((a->N % 128) || (a->K % 64)) return MOY_ENOSYS;
  if ((a->lda % 8) || (a->ldc % 8) || !aligned16(a->A) || !aligned16(a->C) || !aligned16(a->W)) return MOY_ENOSYS;
  if (a->R && ((a->ldr % 4) || (reinterpret_cast<uintptr_t>(a->R) & 7))) return MOY_ENOSYS;
  if ((a->scale && !aligned16(a->scale)) || (a->shift && !aligned16(a->shift))) return MOY_ENOSYS;
  if (a->ksize == 3 && (a->Cin % 64)) return MOY_ENOSYS;
  // form 0: 256 x 256 tiles (N % 256 == 0); form 1: 512 x 128 (N % 128 == 0); form 2: 256 x 128 (MOY_GEMM_DMA_FORM=2 only)
  static const int force = knob("MOY_GEMM_DMA_FORM", -2);
  // measured (288 frames, same device): 256 x 256 beats the tiled kernel by 19-30 % (128 -> 256 stride 2: 564 -> 467 us, 256 -> 256:
  // 280 -> 206 us); 512 x 128 LOSES to the weight-stationary kernel (128 -> 128 stride 1: 274 vs 234 us) and to the tiled one
  // (stride 2: 330 vs 310 us): its A operand is 4/5 of every k-tile's 80 KB and crosses L2 -> LDS nine times; 256 x 128: 352 vs 301 us.
  // So by default only N % 256 == 0 comes here; the other forms stay as A/B knobs (MOY_GEMM_DMA_FORM=1|2).
  int form = 0;
  if (force >= 1) form = force;
  if (form == 0 && (a->N % 256)) return MOY_ENOSYS;
  const int bm = form == 1 ? 512 : 256, bn = form == 0 ? 256 : 128;
  if (mode == 1) {
    // the structure pays where the product is matrix-rate bound and fills the chip: deep K, at least ~1.5 tiles per CU
    const long tiles = (long)((a->M + bm - 1) / bm) * (a->N / bn);
    if (a->K < 512 || tiles < 384) return MOY_ENOSYS;
  }
  GdParams p{};
  {
    static const int variant = knob("MOY_GD_VARIANT", 0);
    p.variant = variant;
  }
  p.A = a->A; p.lda = a->lda; p.W = a->W; p.Kpad = (a->K + 63) / 64 * 64;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.scale = a->scale; p.shift = a->shift; p.act = a->act; p.R = a->R; p.ldr = a->ldr; p.C = a->C; p.ldc = a->ldc;
  if (a->ksize == 3) {
    p.stride = a->stride; p.Hin = a->Hin; p.Win = a->Win; p.Hout = a->Hout; p.Wout = a->Wout; p.Cin = a->Cin;
    p.fd_hw = make_fastdiv((uint32_t)(a->Hout * a->Wout));
    p.fd_wout = make_fastdiv((uint32_t)a->Wout);
    p.cin_magic = (uint32_t)(((1ull << 32) + (unsigned)a->Cin - 1) / (unsigned)a->Cin);
    const int64_t img = (int64_t)a->Hin * a->Win * a->lda * 2;
    const int64_t span = bm / ((int64_t)a->Hout * a->Wout > 0 ? (int64_t)a->Hout * a->Wout : 1) + 2;   // images one row tile can touch
    if (span * img > 0x7fffffffLL && (int64_t)a->B * img > 0x7fffffffLL) return MOY_ENOSYS;
    p.a_bytes = (((int64_t)a->B * a->Hin * a->Win - 1) * a->lda + a->Cin) * 2;
  } else {
    if ((int64_t)bm * a->lda * 2 > 0x7fffffffLL) return MOY_ENOSYS;
    p.a_bytes = (((int64_t)a->M - 1) * a->lda + a->K) * 2;
  }
  if (a->dtype == MOY_BF16) return a->ksize == 3 ? gd_pick<bf16_t, 3>(p, form, st) : gd_pick<bf16_t, 1>(p, form, st);
  return a->ksize == 3 ? gd_pick<f16_t, 3>(p, form, st) : gd_pick<f16_t, 1>(p, form, st);
}

}  // namespace moy
