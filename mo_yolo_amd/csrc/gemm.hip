// Fused implicit-GEMM for gfx950: 1x1 / 3x3 convolutions and linears with BN/bias, activation,
// residual and LayerNorm epilogues (see moy_gemm in include/moyolo.h).
//
// Structure (wave64, 256 threads = 4 waves):
//   * block tile BM x BN, k advanced PANELS panels of 64 bytes per stage
//     (bf16: 32 k per panel -> one v_mfma_f32_16x16x32_bf16 per 16x16 sub-tile and panel;
//      f32 : 16 k per panel -> four v_mfma_f32_16x16x4_f32, the exact-fp32 matrix op);
//   * global -> registers (16-byte chunks, prefetched one stage ahead) -> LDS, two LDS stages,
//     one barrier per stage; LDS rows are 64 B with the 16-B column XOR-swizzled by the row group
//     so the ds_read_b128 fragment reads are bank-conflict free;
//   * MFMA operands are swapped (weights as A, activations as B) so each lane ends up with four
//     consecutive output channels of one pixel -> 16-byte LDS stores of the accumulator tile;
//   * epilogue: accumulators -> LDS (fp32 tile) -> row-wise pass (scale/shift, act, residual,
//     optional LayerNorm with wave shuffles) -> coalesced vector stores.
//   * blockIdx is remapped so that the column tiles of one row tile run on the same XCD (they
//     re-read the same activation rows from that XCD's L2).
#include <stdlib.h>

#include "common.hpp"

namespace moy {

constexpr int PANELS = 2;

struct GemmParams {
  const void* A;
  const void* A2;
  const int32_t* a_rows;
  const uint8_t* a_mask;
  int mask_period;
  int64_t lda;
  const void* W;
  int M, N, K, Kpad;
  int ksize, stride;
  int Hin, Win, Hout, Wout, Cin, lgC;
  const float* scale;
  const float* shift;
  int act;
  const void* R;
  int64_t ldr;
  const float* ln_g;
  const float* ln_b;
  void* C;
  int64_t ldc;
  int out_f32;
  int c_rpb, c_bstride;
  int tiles_n, nblocks;
};

// 16-B column swizzle: lanes of one ds_read_b128 lane group hit distinct bank quartets.
__device__ __forceinline__ int swz(int row, int q) { return q ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3); }

template <typename T>
__device__ __forceinline__ u32x4 add_chunks(u32x4 a, u32x4 b);
template <>
__device__ __forceinline__ u32x4 add_chunks<float>(u32x4 a, u32x4 b) {
  f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
  return __builtin_bit_cast(u32x4, x + y);
}
template <>
__device__ __forceinline__ u32x4 add_chunks<bf16_t>(u32x4 a, u32x4 b) {
  u32x4 r;
  r.x = pack_bf2(bflo(a.x) + bflo(b.x), bfhi(a.x) + bfhi(b.x));
  r.y = pack_bf2(bflo(a.y) + bflo(b.y), bfhi(a.y) + bfhi(b.y));
  r.z = pack_bf2(bflo(a.z) + bflo(b.z), bfhi(a.z) + bfhi(b.z));
  r.w = pack_bf2(bflo(a.w) + bflo(b.w), bfhi(a.w) + bfhi(b.w));
  return r;
}

template <typename T>
__device__ __forceinline__ void mma_panel(f32x4& acc, u32x4 wfrag, u32x4 afrag);
template <>
__device__ __forceinline__ void mma_panel<bf16_t>(f32x4& acc, u32x4 wfrag, u32x4 afrag) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, afrag),
                                                acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_panel<float>(f32x4& acc, u32x4 wfrag, u32x4 afrag) {
  // lane (r, q) holds k = 4q..4q+3 of its row for both operands; step e consumes k = 4q+e of every
  // lane quad, so the four steps together cover the 16 k of the panel exactly once.
  f32x4 w = __builtin_bit_cast(f32x4, wfrag), a = __builtin_bit_cast(f32x4, afrag);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a.w, acc, 0, 0, 0);
}

template <int BM, int BN>
constexpr int gemm_lds_bytes() {
  constexpr int stage = 2 * (BM + BN) * PANELS * 64;
  constexpr int epi = BM * (BN + 4) * 4;
  return stage > epi ? stage : epi;
}

template <typename T, int BM, int BN, bool LN, int NTHR>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const float* Cs, int m0, int n0, int tid) {
  constexpr int LDC = BN + 4;
  const int lane = tid & 63, wave = tid >> 6;
  const T* __restrict__ Rg = static_cast<const T*>(p.R);
  if (!LN) {
    constexpr int CPR = BN / 4;          // 4-column chunks per row
    constexpr int RSTEP = NTHR / CPR;
    const int cc = tid % CPR, rr0 = tid / CPR;
    const int n = n0 + cc * 4;
    if (n < p.N) {                        // N % 4 == 0 (host-checked)
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
      if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
      for (int rr = rr0; rr < BM; rr += RSTEP) {
        const int m = m0 + rr;
        if (m >= p.M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(Cs + rr * LDC + cc * 4);
        v = v * sc + sh;
        v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
        v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
        if (Rg) v += DT<T>::load4(Rg + (int64_t)m * p.ldr + n);
        const int64_t mo = p.c_rpb ? (int64_t)(m / p.c_rpb) * p.c_bstride + (m % p.c_rpb) : (int64_t)m;
        if (p.out_f32)
          *reinterpret_cast<f32x4*>(static_cast<float*>(p.C) + mo * p.ldc + n) = v;
        else
          DT<T>::store4(static_cast<T*>(p.C) + mo * p.ldc + n, v);
      }
    }
  } else {
    // one wave per row; lane owns columns lane*4 .. +3 (N == BN == 256)
    const int n = lane * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
    if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_g + n);
    const f32x4 be = *reinterpret_cast<const f32x4*>(p.ln_b + n);
    for (int rr = wave; rr < BM; rr += NTHR / 64) {
      const int m = m0 + rr;
      if (m >= p.M) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(Cs + rr * LDC + n);
      v = v * sc + sh;
      v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
      v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
      if (Rg) v += DT<T>::load4(Rg + (int64_t)m * p.ldr + n);
      const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
      const f32x4 d = v - mean;
      const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.0f / 256.0f);
      const float rstd = 1.0f / sqrtf(var + 1e-5f);
      v = d * rstd * g + be;
      const int64_t mo = p.c_rpb ? (int64_t)(m / p.c_rpb) * p.c_bstride + (m % p.c_rpb) : (int64_t)m;
      if (p.out_f32)
        *reinterpret_cast<f32x4*>(static_cast<float*>(p.C) + mo * p.ldc + n) = v;
      else
        DT<T>::store4(static_cast<T*>(p.C) + mo * p.ldc + n, v);
    }
  }
}

template <typename T, int BM, int BN, int WGM, int WGN, bool LN, int KS>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int KPB = DT<T>::KPB;      // elements per 16-B chunk
  constexpr int BKP = 4 * KPB;         // elements per 64-B panel
  constexpr int BK = BKP * PANELS;     // elements per stage
  constexpr int RA = BM / 64;          // A rows staged per thread (per panel)
  constexpr int RB = BN / 64;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MT = TM / 16, NT = TN / 16;
  constexpr int A_BYTES = BM * PANELS * 64, B_BYTES = BN * PANELS * 64;
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(BM % 64 == 0 && BN % 64 == 0, "tile");
  static_assert(!LN || BN == 256, "LayerNorm epilogue needs the whole row in one tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave / WGN, wn = wave % WGN;

  // XCD-aware block remap (bijective for any grid size): blocks that share an XCD (id % 8) get
  // consecutive logical ids, i.e. the column tiles of the same row tile.
  int bid = blockIdx.x;
  {
    const int nb = p.nblocks, qd = nb >> 3, rm = nb & 7, x = bid & 7;
    bid = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + (bid >> 3);
  }
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* __restrict__ Ag = static_cast<const T*>(p.A);
  const T* __restrict__ A2g = static_cast<const T*>(p.A2);
  const T* __restrict__ Wg = static_cast<const T*>(p.W);

  // ---- per-thread staging coordinates (loop invariant)
  const int srow = tid >> 2, sq = tid & 3;
  int64_t a_off[RA];   // element offset of the row start (ksize 1) / of pixel (b, 0, 0) (ksize 3)
  bool a_ok[RA];
  int iy0[RA], ix0[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + srow + j * 64;
    a_ok[j] = m < p.M;
    a_off[j] = 0;
    iy0[j] = ix0[j] = 0;
    if (a_ok[j]) {
      if (KS == 1) {
        if (p.a_mask && p.a_mask[m % p.mask_period] == 0) a_ok[j] = false;
        const int64_t row = p.a_rows ? (int64_t)p.a_rows[m] : (int64_t)m;
        a_off[j] = row * p.lda;
      } else {
        const int hw = p.Hout * p.Wout;
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        iy0[j] = oy * p.stride - 1;
        ix0[j] = ox * p.stride - 1;
        a_off[j] = (int64_t)b * p.Hin * p.Win * p.lda;
      }
    }
  }
  int64_t b_off[RB];
  bool b_ok[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int n = n0 + srow + j * 64;
    b_ok[j] = n < p.N;
    b_off[j] = (int64_t)n * p.Kpad;
  }

  u32x4 areg[PANELS][RA], breg[PANELS][RB];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto load_stage = [&](int kt) {
#pragma unroll
    for (int pn = 0; pn < PANELS; ++pn) {
      const int kc = kt * BK + pn * BKP + sq * KPB;
      if (KS == 1) {
        const bool kin = kc < p.K;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
          u32x4 v = zero4;
          if (a_ok[j] && kin) {
            v = *reinterpret_cast<const u32x4*>(Ag + a_off[j] + kc);
            if (A2g) v = add_chunks<T>(v, *reinterpret_cast<const u32x4*>(A2g + a_off[j] + kc));
          }
          areg[pn][j] = v;
        }
      } else {
        const int tap = kc >> p.lgC, c = kc & (p.Cin - 1);
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;   // tap / 3 for tap < 16
        const bool kin = tap < 9;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
          const int iy = iy0[j] + ky, ix = ix0[j] + kx;
          u32x4 v = zero4;
          if (a_ok[j] && kin && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win)
            v = *reinterpret_cast<const u32x4*>(Ag + a_off[j] + ((int64_t)iy * p.Win + ix) * p.lda + c);
          areg[pn][j] = v;
        }
      }
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        u32x4 v = zero4;
        if (b_ok[j]) v = *reinterpret_cast<const u32x4*>(Wg + b_off[j] + kc);   // kc < Kpad by construction
        breg[pn][j] = v;
      }
    }
  };

  auto store_stage = [&](int buf) {
    unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int pn = 0; pn < PANELS; ++pn) {
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        const int row = srow + j * 64;
        *reinterpret_cast<u32x4*>(As + (pn * BM + row) * 64 + swz(row, sq) * 16) = areg[pn][j];
      }
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int row = srow + j * 64;
        *reinterpret_cast<u32x4*>(Bs + (pn * BN + row) * 64 + swz(row, sq) * 16) = breg[pn][j];
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.Kpad / BK;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_stage(kt + 1);
    const unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
    const unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int pn = 0; pn < PANELS; ++pn) {
      u32x4 af[MT], wf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * TM + i * 16 + r;
        af[i] = *reinterpret_cast<const u32x4*>(As + (pn * BM + row) * 64 + swz(row, q) * 16);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * TN + j * 16 + r;
        wf[j] = *reinterpret_cast<const u32x4*>(Bs + (pn * BN + row) * 64 + swz(row, q) * 16);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mma_panel<T>(acc[i][j], wf[j], af[i]);
    }
    if (kt + 1 < nk) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS fp32 tile [BM][BN+4]
  constexpr int LDC = BN + 4;
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      // D[n_local = q*4 + reg][m_local = r]
      const int ml = wm * TM + i * 16 + r, nl = wn * TN + j * 16 + q * 4;
      *reinterpret_cast<f32x4*>(Cs + ml * LDC + nl) = acc[i][j];
    }
  __syncthreads();

  gemm_epilogue<T, BM, BN, LN, 256>(p, Cs, m0, n0, tid);
}

// =================================================================================================
// LDS-DMA variant of the main loop (global_load_lds_dwordx4, no staging registers):
//   * NST-deep ring of LDS stages; the DMA for stage kt+NST-1 is issued while stage kt is computed,
//     so NST-1 stages (tens of KB per CU) are in flight continuously -- these GEMMs have K of only
//     64..2304 and stream M rows once, so bytes in flight, not MFMA rate, set their speed;
//   * a wave instruction writes 1 KiB of LDS linearly (16 rows x 64 B of one panel); the XOR
//     swizzle is applied on the SOURCE address (lane i fetches the chunk that belongs at its linear
//     slot), and the ds_read side applies the same involution;
//   * out-of-image taps, M / N tails, masked rows and K padding are fetched from a 16-byte zero page;
//   * one raw s_barrier per stage with a counted s_waitcnt vmcnt (never __syncthreads in the loop,
//     which would drain the DMA queue).
// The epilogue is shared with the register-staged kernel.
__device__ __attribute__((aligned(16))) uint32_t g_zero_page[4] = {0u, 0u, 0u, 0u};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int BN, int NST>
constexpr int gemm_dma_lds_bytes() {
  constexpr int stage = NST * (BM + BN) * PANELS * 64;
  constexpr int epi = BM * (BN + 4) * 4;
  return stage > epi ? stage : epi;
}

template <typename T, int BM, int BN, int WGM, int WGN, bool LN, int KS, int NST>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_dma_kernel(const GemmParams p) {
  constexpr int NW = WGM * WGN, NTHR = 64 * NW;
  constexpr int KPB = DT<T>::KPB, BKP = 4 * KPB, BK = BKP * PANELS;
  constexpr int TM = BM / WGM, TN = BN / WGN, MT = TM / 16, NT = TN / 16;
  constexpr int A_BYTES = BM * PANELS * 64, B_BYTES = BN * PANELS * 64, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int RGA = BM / 16, RGB = BN / 16;               // 16-row groups per panel
  constexpr int PA = RGA * PANELS, PB = RGB * PANELS;       // 1-KiB pieces per stage
  constexpr int PPA = (PA + NW - 1) / NW, PPB = (PB + NW - 1) / NW;   // pieces per wave
  constexpr int NPW = PPA + PPB;                            // DMA instructions per wave and stage
  static_assert(PA % NW == 0 && PB % NW == 0, "pieces must divide over the waves");
  static_assert(!LN || BN == 256, "LayerNorm epilogue needs the whole row in one tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave / WGN, wn = wave % WGN;

  int bid = blockIdx.x;
  {
    const int nb = p.nblocks, qd = nb >> 3, rm = nb & 7, x = bid & 7;
    bid = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + (bid >> 3);
  }
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* __restrict__ Ag = static_cast<const T*>(p.A);
  const T* __restrict__ Wg = static_cast<const T*>(p.W);
  const T* zero = reinterpret_cast<const T*>(g_zero_page);

  // ---- per-lane, per-piece source coordinates (loop invariant)
  const int lrow = lane >> 2, lcol = lane & 3;
  int64_t a_off[PPA];
  int iy0[PPA], ix0[PPA], a_q[PPA];
  bool a_ok[PPA];
#pragma unroll
  for (int j = 0; j < PPA; ++j) {
    const int pc = wave + NW * j, rg = pc % RGA;
    const int row = rg * 16 + lrow, m = m0 + row;
    a_q[j] = swz(row, lcol);                 // chunk that lives at this lane's linear LDS slot
    a_ok[j] = m < p.M;
    a_off[j] = 0; iy0[j] = ix0[j] = 0;
    if (a_ok[j]) {
      if (KS == 1) {
        if (p.a_mask && p.a_mask[m % p.mask_period] == 0) a_ok[j] = false;
        const int64_t rowi = p.a_rows ? (int64_t)p.a_rows[m] : (int64_t)m;
        a_off[j] = rowi * p.lda;
      } else {
        const int hw = p.Hout * p.Wout;
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        iy0[j] = oy * p.stride - 1; ix0[j] = ox * p.stride - 1;
        a_off[j] = (int64_t)b * p.Hin * p.Win * p.lda;
      }
    }
  }
  int64_t b_off[PPB];
  int b_q[PPB];
  bool b_ok[PPB];
#pragma unroll
  for (int j = 0; j < PPB; ++j) {
    const int pc = wave + NW * j, rg = pc % RGB;
    const int row = rg * 16 + lrow, n = n0 + row;
    b_q[j] = swz(row, lcol);
    b_ok[j] = n < p.N;
    b_off[j] = (int64_t)n * p.Kpad;
  }

  auto issue_stage = [&](int kt, int buf) {
    unsigned char* As = smem + buf * ST_BYTES;
    unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int j = 0; j < PPA; ++j) {
      const int pc = wave + NW * j, pn = pc / RGA, rg = pc % RGA;
      const int kc = kt * BK + pn * BKP + a_q[j] * KPB;
      const T* src = zero;
      if (KS == 1) {
        if (a_ok[j] && kc < p.K) src = Ag + a_off[j] + kc;
      } else {
        const int tap = kc >> p.lgC, c = kc & (p.Cin - 1);
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        const int iy = iy0[j] + ky, ix = ix0[j] + kx;
        if (a_ok[j] && tap < 9 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win)
          src = Ag + a_off[j] + ((int64_t)iy * p.Win + ix) * p.lda + c;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(As + (pn * BM + rg * 16) * 64), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PPB; ++j) {
      const int pc = wave + NW * j, pn = pc / RGB, rg = pc % RGB;
      const int kc = kt * BK + pn * BKP + b_q[j] * KPB;
      const T* src = b_ok[j] ? Wg + b_off[j] + kc : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(Bs + (pn * BN + rg * 16) * 64), 16, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.Kpad / BK;
  // prologue: NST-1 stages in flight
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue_stage(s, s);
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed; the (up to NST-2) younger stages stay in flight
    const int younger = min(NST - 2, nk - 1 - kt);
    if (NST >= 4 && younger == 2) wait_vmcnt<2 * NPW>();
    else if (NST >= 3 && younger == 1) wait_vmcnt<NPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NST - 1 < nk) {
      int nb = buf + NST - 1;
      if (nb >= NST) nb -= NST;
      issue_stage(kt + NST - 1, nb);
    }
    const unsigned char* As = smem + buf * ST_BYTES;
    const unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int pn = 0; pn < PANELS; ++pn) {
      u32x4 af[MT], wf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * TM + i * 16 + r;
        af[i] = *reinterpret_cast<const u32x4*>(As + (pn * BM + row) * 64 + swz(row, q) * 16);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * TN + j * 16 + r;
        wf[j] = *reinterpret_cast<const u32x4*>(Bs + (pn * BN + row) * 64 + swz(row, q) * 16);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mma_panel<T>(acc[i][j], wf[j], af[i]);
    }
    if (++buf == NST) buf = 0;
  }
  __syncthreads();   // all waves done with the ring before it is reused as the fp32 output tile

  constexpr int LDC = BN + 4;
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ml = wm * TM + i * 16 + r, nl = wn * TN + j * 16 + q * 4;
      *reinterpret_cast<f32x4*>(Cs + ml * LDC + nl) = acc[i][j];
    }
  __syncthreads();
  gemm_epilogue<T, BM, BN, LN, NTHR>(p, Cs, m0, n0, tid);
}

template <typename T, int BM, int BN, int WGM, int WGN, bool LN, int KS, int NST>
static int launch_dma(GemmParams& p, hipStream_t st) {
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.nblocks = tiles_m * p.tiles_n;
  constexpr int lds = gemm_dma_lds_bytes<T, BM, BN, NST>();
  static_assert(lds <= 160 * 1024, "LDS");
  auto kern = gemm_dma_kernel<T, BM, BN, WGM, WGN, LN, KS, NST>;
  static bool attr_set = false;
  if (lds > 65536 && !attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(64 * WGM * WGN), lds, st, p);
  return launch_status();
}

template <typename T, int KS>
static int dispatch_dma(GemmParams& p, bool ln, hipStream_t st) {
  if (ln) return launch_dma<T, 64, 256, 1, 4, true, KS, 3>(p, st);
  const long big = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  if (p.N > 64) {
    if (big >= 384) return launch_dma<T, 128, 128, 2, 2, false, KS, 3>(p, st);
    return launch_dma<T, 64, 128, 2, 2, false, KS, 3>(p, st);
  }
  const long mid = (long)((p.M + 127) / 128);
  if (mid >= 384) return launch_dma<T, 128, 64, 2, 2, false, KS, 3>(p, st);
  return launch_dma<T, 64, 64, 2, 2, false, KS, 3>(p, st);
}

template <typename T, int BM, int BN, int WGM, int WGN, bool LN, int KS>
static int launch_cfg(GemmParams& p, hipStream_t st) {
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.nblocks = tiles_m * p.tiles_n;
  constexpr int lds = gemm_lds_bytes<BM, BN>();
  auto kern = gemm_kernel<T, BM, BN, WGM, WGN, LN, KS>;
  static bool attr_set = false;   // > 64 KiB dynamic LDS needs the opt-in once per kernel symbol
  if (lds > 65536 && !attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
        hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds, st, p);
  return launch_status();
}

template <typename T, int KS>
static int dispatch_tile(GemmParams& p, bool ln, hipStream_t st) {
  if (ln) return launch_cfg<T, 64, 256, 1, 4, true, KS>(p, st);
  // Tile choice: fill >= ~2 blocks per CU when the problem allows it, keep tiles large otherwise.
  const long big = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  if (p.N > 64) {
    if (big >= 384) return launch_cfg<T, 128, 128, 2, 2, false, KS>(p, st);
    return launch_cfg<T, 64, 128, 2, 2, false, KS>(p, st);
  }
  const long mid = (long)((p.M + 127) / 128);
  if (mid >= 384) return launch_cfg<T, 128, 64, 2, 2, false, KS>(p, st);
  return launch_cfg<T, 64, 64, 2, 2, false, KS>(p, st);
}

}  // namespace moy

using namespace moy;

extern "C" int moy_gemm(const moy_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->W || !a->C) return MOY_EINVAL;
  if (a->dtype != MOY_F32 && a->dtype != MOY_BF16) return MOY_EINVAL;
  const int kpb = a->dtype == MOY_BF16 ? 8 : 4;
  const int esz = a->dtype == MOY_BF16 ? 2 : 4;
  const int bk = 4 * kpb * PANELS;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return MOY_EINVAL;
  if (a->N % 4) return MOY_EINVAL;
  if (a->ksize != 1 && a->ksize != 3) return MOY_EINVAL;
  if ((a->lda * esz) % 16 || !aligned16(a->A) || !aligned16(a->W)) return MOY_EINVAL;
  if (a->A2 && (!aligned16(a->A2) || a->ksize != 1)) return MOY_EINVAL;
  const int out_esz = a->out_f32 ? 4 : esz;
  if ((a->ldc * out_esz) % (4 * out_esz) || (reinterpret_cast<uintptr_t>(a->C) % (4 * out_esz))) return MOY_EINVAL;
  if (a->R && ((a->ldr * esz) % (4 * esz) || reinterpret_cast<uintptr_t>(a->R) % (4 * esz))) return MOY_EINVAL;
  if ((a->scale && !aligned16(a->scale)) || (a->shift && !aligned16(a->shift))) return MOY_EINVAL;
  const bool ln = a->ln_g != nullptr;
  if (ln && (!a->ln_b || a->N != 256 || !aligned16(a->ln_g) || !aligned16(a->ln_b))) return MOY_EINVAL;
  if (a->a_mask && (a->mask_period <= 0 || a->ksize != 1)) return MOY_EINVAL;
  if (a->a_rows && a->ksize != 1) return MOY_EINVAL;

  GemmParams p{};
  p.A = a->A; p.A2 = a->A2; p.a_rows = a->a_rows; p.a_mask = a->a_mask; p.mask_period = a->mask_period;
  p.lda = a->lda; p.W = a->W; p.M = a->M; p.N = a->N; p.K = a->K;
  p.Kpad = (a->K + bk - 1) / bk * bk;
  p.ksize = a->ksize; p.stride = a->stride;
  p.scale = a->scale; p.shift = a->shift; p.act = a->act; p.R = a->R; p.ldr = a->ldr;
  p.ln_g = a->ln_g; p.ln_b = a->ln_b; p.C = a->C; p.ldc = a->ldc; p.out_f32 = a->out_f32;
  if (a->c_rows_per_batch < 0 || (a->c_rows_per_batch > 0 && a->c_batch_stride < a->c_rows_per_batch)) return MOY_EINVAL;
  p.c_rpb = a->c_rows_per_batch; p.c_bstride = a->c_batch_stride;
  if (a->ksize == 1) {
    if (a->K % kpb) return MOY_EINVAL;
  } else {
    const int C = a->Cin;
    if (C < kpb || (C & (C - 1)) || a->K != 9 * C) return MOY_EINVAL;
    if (a->stride != 1 && a->stride != 2) return MOY_EINVAL;
    if (a->B <= 0 || a->Hin <= 0 || a->Win <= 0) return MOY_EINVAL;
    if (a->Hout != (a->Hin + 2 - 3) / a->stride + 1 || a->Wout != (a->Win + 2 - 3) / a->stride + 1) return MOY_EINVAL;
    if ((long)a->B * a->Hout * a->Wout != a->M) return MOY_EINVAL;
    p.Hin = a->Hin; p.Win = a->Win; p.Hout = a->Hout; p.Wout = a->Wout; p.Cin = C;
    int lg = 0;
    while ((1 << lg) < C) ++lg;
    p.lgC = lg;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // Main-loop variant: register-staged (default) or LDS-DMA ring (MOY_GEMM_IMPL=dma).  Measured on
  // MI355X (tools/bench_gemm.py, round 1): the DMA ring needs 72-120 KB of LDS per block, i.e. 1-2
  // blocks per CU, and loses 20-50 % to the register-staged kernel at 2-3 blocks per CU on every
  // shape of this path -- these short-K GEMMs hide latency with occupancy, not with ring depth.
  static const bool use_dma = [] { const char* e = getenv("MOY_GEMM_IMPL"); return e && e[0] == 'd'; }();
  if (!a->A2 && use_dma) {   // the A + A2 prologue add needs the register-staged kernel
    if (a->dtype == MOY_BF16)
      return a->ksize == 1 ? dispatch_dma<bf16_t, 1>(p, ln, st) : dispatch_dma<bf16_t, 3>(p, ln, st);
    return a->ksize == 1 ? dispatch_dma<float, 1>(p, ln, st) : dispatch_dma<float, 3>(p, ln, st);
  }
  if (a->dtype == MOY_BF16)
    return a->ksize == 1 ? dispatch_tile<bf16_t, 1>(p, ln, st) : dispatch_tile<bf16_t, 3>(p, ln, st);
  return a->ksize == 1 ? dispatch_tile<float, 1>(p, ln, st) : dispatch_tile<float, 3>(p, ln, st);
}
