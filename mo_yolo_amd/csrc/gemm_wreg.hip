// Weight-stationary GEMM for the token-major linears of the path whose K is 256 (value_proj of all decoder layers in one
// launch, transformer.py:255-257; enc_output, head.py:1036-1040): C[M, N] = act(A[M, 256] . W[N, 256]^T * scale + shift),
// 16-bit types, M in the millions, N a multiple of 256.
//
// Why a second GEMM kernel: in the tiled kernel (gemm.hip) a 128x128x256 tile stages 128 KB through the VGPR->LDS store path
// (~79 B/clk/CU) and re-reads 384 KB of fragments for only 512 MFMAs; with K this short the tile prologue, the LDS traffic and
// the epilogue are not amortised and the value projection ran at 2.2 TB/s / 490 TFLOP/s.  Here
//   * a block owns 256 output columns for its whole life: each of its 4 waves keeps W[64 columns][256 k] in REGISTERS
//     (128 VGPRs as MFMA A-operand fragments) -- weights never touch LDS again;
//   * the block walks row tiles of BM rows; an activation tile is BM x 512 B, brought in by LDS-DMA
//     (global_load_lds_dwordx4: no VGPRs, no ds_write) into a ring of NBUF buffers, DIST = NBUF-1 tiles ahead of the MFMAs,
//     retired by a counted vmcnt that leaves the younger tiles and the output stores in flight; ONE barrier per tile;
//   * the LDS image of a tile row is its 32 16-B chunks with the chunk index XORed by (row & 15): the DMA destination is
//     lane-linear, so the swizzle is applied to the per-lane SOURCE address; every ds_read_b128 fragment read is
//     conflict-free (lanes of a 16-lane group hit 16 distinct chunk slots mod 16);
//   * MFMA operands swapped (weights = A): a lane owns 4 consecutive output channels of one row, as in gemm.hip, and the k
//     order per accumulator is the same 8 panels of 32 -> results are bit-identical to the tiled kernel;
//   * epilogue per WAVE straight from the accumulators: the weight rows are dealt to the MFMA fragments so that a lane ends up
//     with 8 consecutive output channels of a row -> 16-byte stores, 64 contiguous bytes per row and store instruction, no LDS
//     round trip and no block barrier, so one wave's epilogue overlaps the other waves' MFMAs;
//   * grid = 8 XCDs x slots; the N/256 column groups of one row-tile sequence sit on the SAME XCD and advance in lock step,
//     so an activation tile is fetched from HBM once and hits that XCD's L2 for the other groups;
//   * other forms of the same kernel: 8 waves x 32 columns (1x1 convs / input_proj with 256 outputs, K in {128, 256, 384, 512}),
//     4 waves x 32 columns, two blocks per CU (128 outputs, K in {128, 192, 256}), score mode (LayerNorm + narrow head, no
//     feature output), and the SEEDED forms (moy_gemm_args.pre: accumulators start from the nearest-2x rows of a half-resolution
//     fp32 product, fetched one tile ahead by hand-counted asynchronous register loads).
#include "common.hpp"

#include <type_traits>
#include <utility>

namespace moy {

struct WregParams {   // ngroups = N / (columns per block)
  const void* A;
  int64_t lda;
  const void* W;
  const float* scale;
  const float* shift;
  int act;
  void* C;
  int64_t ldc;
  int M, N;
  int plane_cols;       // column planes (moy_gemm_args.plane_cols), 0 = off
  int64_t plane_stride;
  int c_rpb, c_bstride; // output row remap (moy_gemm_args.c_rows_per_batch / c_batch_stride), 0 = off
  FastDiv fd_rpb;
  int ngroups;   // N / 256
  int lanes;     // row-tile sequences per XCD
  int ntiles;    // ceil(M / BM)
  // score mode with ROW RUNS (moy_gemm_args.run_*, round 3): only the rows b * run_period + tok0[l] + y * pitch[l] + x are visited
  // (compact row c -> b = c / run_nv, v = c % run_nv -> level l by the compact starts -> (y, x) by the run length)
  int run_levels, run_period, run_nv, run_mv;          // run_mv = (M / run_period) * run_nv compact rows
  int run_a_period, run_a_off;                         // A row = b * run_a_period + token - run_a_off (== score row when run_a_period == run_period, run_a_off == 0)
  int run_cstart[4], run_len[4], run_tok0[4], run_pitch[4];
  FastDiv fd_nv, fd_len[4];
  // score mode (LN = true): LayerNorm statistics + the narrow head of the normalised row, nothing else is written
  const float* ln_g;
  const float* ln_b;
  const float* dot_w;
  const float* dot_b;
  float* dot_out;
  int dot_n;
  const uint8_t* a_mask;
  int mask_period;
  FastDiv fd_mask;
  // accumulator seed (moy_gemm_args.pre; PRE forms): row m = (b, y, x) of a [B, pre_h, pre_w] raster starts from the fp32 row
  // (b * lowhw + (y/2) * loww + x/2) of pre
  const float* pre;
  int64_t ld_pre;
  int pre_w, pre_hw, pre_loww, pre_lowhw;
  FastDiv fd_pre_hw, fd_pre_w;
};

// LDS-DMA of 16 bytes per lane: LDS[m0 + lane*16 ..] <- *gsrc.  M0 is compiler-reserved: saved and restored in the statement.
// Source = wave-uniform base (SGPR pair) + 32-bit per-lane byte offset.
__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds_dst), "s"(sbase)
               : "memory");
}

// 16-byte buffer load whose completion the CALLER bookkeeps (counted vmcnt): hipcc knows nothing of the LDS-DMA pieces in the
// queue, so for a load it can see it would wait vmcnt(0) at the first use and drain the ring.  The destination is not valid
// until a wait_vmcnt_tie() that covers it has run.
__device__ __forceinline__ void bload16_async(u32x4& dst, uint32_t voff, __amdgpu_buffer_rsrc_t rs) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(dst) : "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void bload16_async_16(u32x4& dst, uint32_t voff, __amdgpu_buffer_rsrc_t rs) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=&v"(dst) : "v"(voff), "s"(rs) : "memory");
}
// counted wait that the four async destinations depend on (the "+v" ties keep every use of them behind it)
template <int N>
__device__ __forceinline__ void wait_vmcnt_tie(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// End-of-tile wait of a DMA ring whose tiles also issue NST stores each: tile it+1 must have landed.  The wave's queue, oldest
// first, is DMA(it+1), stores(it-DIST+1), DMA(it+2), ..., DMA(it+DIST), stores(it): behind the piece that is needed lie DIST-1
// younger DMA groups and one store group per tile ALREADY COMPUTED, i.e. min(DIST, it+1) of them -- not DIST.  Round 5: until
// then the steady-state count (DIST-1)*(IPW+NST)+NST was used from the first tile on; with NST >= IPW it exceeds what is
// outstanding at the end of tile 0 (2 IPW + NST at DIST = 2), so the wait was empty and tile 1 was read on the strength of
// having been requested one tile earlier (the one-off wrong value planes of round 4: DESIGN.md section 4, round 5 item 1).
template <int DIST, int IPW, int NST>
__device__ __forceinline__ void wait_ring_tile(int it) {   // `it` wave-uniform
  if constexpr (DIST >= 2 && NST > 0) {
    if (it == 0) { wait_vmcnt<(DIST - 1) * IPW + NST>(); return; }
  }
  if constexpr (DIST >= 3 && NST > 0) {
    if (it == 1) { wait_vmcnt<(DIST - 1) * IPW + 2 * NST>(); return; }
  }
  if constexpr (DIST >= 4 && NST > 0) {
    if (it == 2) { wait_vmcnt<(DIST - 1) * IPW + 3 * NST>(); return; }
  }
  static_assert(DIST <= 4, "ring depth");
  wait_vmcnt<(DIST - 1) * IPW + DIST * NST>();
}

template <typename T>
__device__ __forceinline__ f32x4 mfma16(f32x4 acc, u32x4 w, u32x4 a);
template <>
__device__ __forceinline__ f32x4 mfma16<bf16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<f16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}

constexpr int WREG_MAXDOT = 4;           // classes of the fused narrow head in score mode
template <int BM, int NBUF, bool LN, int NW = 4, int WC = 64, int K = 256>
constexpr int wreg_lds_bytes() {
  // A ring | scale, shift (score mode reads them from LDS) | (score mode) g*w per class, G/B constants, row partials [BM][6][16]
  return NBUF * BM * K * 2 + NW * WC * 8 + (LN ? WREG_MAXDOT * 1024 + 64 + BM * ((2 + WREG_MAXDOT) * 64 + 16) : 0);
}

// WC = output columns per wave (64: K = 256 only; 32: K up to 512 fits the registers), K = reduction length (multiple of 128).
// ABL (timing-only builds, MOY_WREG_ABL, garbage results): bit 0 no MFMAs, bit 1 output stores dropped, bit 2 no activation DMA past
// the prologue, bit 4 s_memtime stamps per phase at the head of C
template <typename T, int BM, int NBUF, int OCC, bool LN, int NW, int WC, int K, int ABL = 0, bool PRE = false>
__global__ __launch_bounds__(64 * NW, (NW == 8 && OCC == 2) ? 4 : OCC) void gemm_wreg_kernel(const WregParams p) {
  static_assert((NW == 4 || NW == 8) && (!LN || (NW == 4 && WC == 64 && (K == 256 || K == 128))), "score mode: 4 waves x 64 columns, K = 256 or 128");
  static_assert((WC == 32 || WC == 64) && K % 64 == 0 && K <= 512, "column width per wave / reduction length");
  constexpr int BNB = NW * WC;             // output columns per block
  constexpr int MT = BM / 16;              // row sub-tiles per wave (every wave covers all BM rows)
  constexpr int NT = WC / 16;              // column sub-tiles per wave
  constexpr int KB = K * 2;                // bytes per activation row
  constexpr int CPR = K / 8;               // 16-byte chunks per row
  // swizzle: chunk ^ (row & SWZ) must stay inside the row.  K a multiple of 128: groups of 16, every ds_read_b128 conflict-free.
  // K = 192 (24 chunks, the P3 neck C2f's cv2): groups of 8 -- rows r and r + 8 of a fragment then share bank slots (two LDS cycles
  // more per read), which a kernel bound by HBM does not notice
  constexpr int SWZ = CPR % 16 == 0 ? 15 : 7;
  constexpr int KP = K / 32;               // MFMA k panels
  constexpr int TILE_BYTES = BM * KB;
  constexpr int IPW = TILE_BYTES / 1024 / NW;  // DMA instructions per wave and tile
  constexpr int DIST = NBUF - 1;           // tiles in flight ahead of the one being computed
  constexpr int NST = LN ? 0 : MT * NT / 2;   // 16-byte store instructions per wave and tile (score mode: see below)
  static_assert(BM % 16 == 0 && IPW >= 1 && IPW * NW * 1024 == TILE_BYTES && NT % 2 == 0, "tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases and descriptors live in SGPRs
  const int r = lane & 15, q = lane >> 4;

  // block -> (XCD, column group, row-tile lane)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = slot % p.ngroups, ln = slot / p.ngroups;
  if (ln >= p.lanes) return;
  const int t0 = xcd + 8 * ln, tstep = 8 * p.lanes;
  if (t0 >= p.ntiles) return;
  const int n_mine = (p.ntiles - t0 + tstep - 1) / tstep;

  float* ssc = reinterpret_cast<float*>(smem + NBUF * TILE_BYTES);
  float* ssh = ssc + BNB;
  float* gw = ssh + BNB;                   // score mode: ln_g[n] * dot_w[c][n]
  float* GB = gw + WREG_MAXDOT * 256;      // [0..3] sum_n gw[c][n]; [4..7] sum_n ln_b[n] * dot_w[c][n] + dot_b[c]
  constexpr int PROW = (2 + WREG_MAXDOT) * 16 + 4;   // floats per row of P: [quantity][wave * 4 + lane group]; +4: rows 4 banks apart
  float* P = GB + 16;                      // row partials, reduced by wave 0 after one barrier
  uint32_t* mbits = reinterpret_cast<uint32_t*>(P + BM * PROW);   // valid-token bitmask (mask_period bits), dynamic tail of the LDS
  const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  const int nb = grp * BNB;                // first output column of the block
  {
    if (tid < BNB) {
      ssc[tid] = p.scale ? p.scale[nb + tid] : 1.0f;
      ssh[tid] = p.shift ? p.shift[nb + tid] : 0.0f;
    }
  }
  if constexpr (LN) {
    // score_c = sum_n LN(v)_n w_cn + b_c = rstd * (sum_n v_n g_n w_cn - mean * G_c) + B_c: every row quantity is a plain sum over
    // the row, so ONE reduction round (sum v, sum v^2, sum v gw_c) replaces the normalise-then-dot passes
#pragma unroll
    for (int c = 0; c < WREG_MAXDOT; ++c) {
      const float wv = c < p.dot_n ? p.dot_w[c * 256 + tid] : 0.0f;
      const float g = p.ln_g[tid] * wv, bb = p.ln_b[tid] * wv;
      gw[c * 256 + tid] = g;
      const float gs = wave_sum(g), bs = wave_sum(bb);
      if (lane == 0) { P[(c * 2 + 0) * 4 + wave] = gs; P[(c * 2 + 1) * 4 + wave] = bs; }
    }
    if (p.a_mask) {
      const int nwords = (p.mask_period + 31) >> 5;
      for (int w = tid; w < nwords; w += 256) {
        uint32_t bits = 0;
        for (int k = 0; k < 32; ++k) {
          const int idx = w * 32 + k;
          if (idx < p.mask_period && p.a_mask[idx]) bits |= 1u << k;
        }
        mbits[w] = bits;
      }
    }
    __syncthreads();
    if (tid < WREG_MAXDOT) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(P + (tid * 2 + 0) * 4), b = *reinterpret_cast<const f32x4*>(P + (tid * 2 + 1) * 4);
      GB[tid] = a.x + a.y + a.z + a.w;
      GB[4 + tid] = b.x + b.y + b.z + b.w + (tid < p.dot_n ? p.dot_b[tid] : 0.0f);
    }
    // (ordered before the first use by the barrier that ends the prologue)
  }

  // ---- DMA geometry (loop invariant): instruction I = wave*IPW + jj fills LDS bytes [I*1024, +1024) of the tile image, i.e. the
  // 64 16-byte chunks X = I*64 + lane (row X / CPR, slot X % CPR; the slot holds source chunk slot ^ (row & 15)).
  // Source address = scalar tile base + per-lane 32-bit offset; rows past M (last tile only) re-read row M-1, never stored.
  const unsigned char* Ab = static_cast<const unsigned char*>(p.A);
  const uint32_t lds_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_base + wave * IPW * 1024));
  int drow[IPW];                                                   // tile row of this lane's chunk
  uint32_t dcol[IPW];                                              // byte offset of the source chunk inside its row
#pragma unroll
  for (int jj = 0; jj < IPW; ++jj) {
    const int X = (wave * IPW + jj) * 64 + lane;
    drow[jj] = X / CPR;
    dcol[jj] = (uint32_t)(((X % CPR) ^ (drow[jj] & SWZ)) * 16);
  }
  const uint32_t row_bytes = (uint32_t)(p.lda * 2);
  // compact row -> row of A (score mode with row runs): arithmetic only -- a table lookup from global memory would put
  // compiler-visible loads into the vector-memory queue that the counted waits of the DMA ring bookkeep by hand
  auto run_row = [&](int c, bool a_side = false) {
    const int b = (int)fdiv((uint32_t)c, p.fd_nv);
    const int v = c - b * p.run_nv;
    int cs = 0, len = p.run_len[0], tok0 = p.run_tok0[0], pitch = p.run_pitch[0];
    uint32_t mg = p.fd_len[0].magic, shf = p.fd_len[0].shift;
#pragma unroll
    for (int l = 1; l < 4; ++l) {
      const bool in = l < p.run_levels && v >= p.run_cstart[l];
      cs = in ? p.run_cstart[l] : cs; len = in ? p.run_len[l] : len; tok0 = in ? p.run_tok0[l] : tok0; pitch = in ? p.run_pitch[l] : pitch;
      mg = in ? p.fd_len[l].magic : mg; shf = in ? p.fd_len[l].shift : shf;
    }
    const uint32_t vv = (uint32_t)(v - cs);
    const uint32_t y = (uint32_t)(((uint64_t)__umulhi(vv, mg) + vv) >> shf);
    const int tok = tok0 + (int)y * pitch + (int)(vv - y * (uint32_t)len);
    return a_side ? b * p.run_a_period + tok - p.run_a_off : b * p.run_period + tok;
  };
  auto issue_tile = [&](int tile, int buf) {
    const int m0 = min(tile, p.ntiles - 1) * BM;                   // over-fetch tiles past the end re-read the last one
    if (LN && p.run_levels) {
#pragma unroll
      for (int jj = 0; jj < IPW; ++jj) {
        const int row = run_row(min(m0 + drow[jj], p.run_mv - 1), true);
        glds16(Ab, (uint32_t)row * row_bytes + dcol[jj], lds_w + buf * TILE_BYTES + jj * 1024);   // (M * lda * 2 < 4 GiB: checked by the host)
      }
      return;
    }
    const unsigned char* tb = Ab + (int64_t)m0 * p.lda * 2;
    const int last = p.M - 1 - m0;                                 // last valid row of this tile (>= BM-1 except in the tail)
#pragma unroll
    for (int jj = 0; jj < IPW; ++jj) {
      const int row = min(drow[jj], last);
      glds16(tb, (uint32_t)row * row_bytes + dcol[jj], lds_w + buf * TILE_BYTES + jj * 1024);
    }
  };

  // ---- weights of this wave's WC columns -> registers (MFMA A operand: lane (r,q) supplies row r of fragment j, k = pn*32 + q*8 ..+7).
  // Store mode: fragment row r of the pair (2t, 2t+1) is channel 32t + 8(r>>2) + 4(j&1) + (r&3), so that the output lane (r, q) --
  // which owns fragment rows 4q..4q+3 -- holds channels 32t + 8q + 0..3 from fragment 2t and + 4..7 from 2t+1: 8 consecutive ones.
  // (A permutation of which channel sits where; every output is the same sum in the same k order.)  Score mode: channel j*16 + r.
  u32x4 wf[NT][KP];
  {
    const T* Wg = static_cast<const T*>(p.W) + (int64_t)(nb + wave * WC) * K;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int pn = 0; pn < KP; ++pn) {
        const int ch = LN ? j * 16 + r : (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        wf[j][pn] = *reinterpret_cast<const u32x4*>(Wg + ch * K + pn * 32 + q * 8);
      }
  }

  // Per-lane scale / shift of the 8 channels a lane owns in each channel pair-group t (store mode; see the weight load)
  // LEAN (the K = 256 value form on 64-row tiles): 128 weight + 64 accumulator registers leave none for these 32 -- the host admits
  // the form only without a BN scale (a Linear: bias only), and the shift is read from LDS where it is added
  constexpr bool LEAN = !LN && WC == 64 && K == 256 && BM == 64;
  float scr[(LN || LEAN) ? 1 : NT / 2][8], shr[(LN || LEAN) ? 1 : NT / 2][8];
  if constexpr (!LN && !LEAN) {
#pragma unroll
    for (int t = 0; t < NT / 2; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = nb + wave * WC + t * 32 + q * 8 + e;
        scr[t][e] = p.scale ? p.scale[c] : 1.0f;
        shr[t][e] = p.shift ? p.shift[c] : 0.0f;
      }
  }
  // prologue: DIST tiles in flight (tiles past the end are clamped re-reads that nobody consumes: the counts stay uniform)
  const int ncol = nb + wave * WC;             // a wave's columns: inside one plane (plane_cols % 64 == 0) or two planes of 32
  // PRE: the seed rows of a tile are fetched one tile ahead, straight into registers (each lane: the 8 channels it owns of its MT
  // rows = MT * NT 16-byte loads), issued BEFORE that pass's DMA so that one counted wait covers both
  static_assert(!PRE || (!LN && DIST == 2 && MT * NT == 4 && ABL == 0), "seeded form: 8 waves x 32 columns or 4 x 32, 32-row tiles, two tiles in flight");
  u32x4 seed[PRE ? 4 : 1];
  [[maybe_unused]] const auto rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pre), 0, 0x7fffffffu, 0x00020000);   // (size checked by the host)
  [[maybe_unused]] auto issue_seed = [&](int tile, u32x4 (&dst)[PRE ? 4 : 1]) {
    if constexpr (PRE) {
      const int m0 = min(tile, p.ntiles - 1) * BM;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const uint32_t m = (uint32_t)min(m0 + i * 16 + r, p.M - 1);
        const int b = (int)fdiv(m, p.fd_pre_hw), rem = (int)m - b * p.pre_hw;
        const int y = (int)fdiv((uint32_t)rem, p.fd_pre_w), x = rem - y * p.pre_w;
        const uint32_t row = (uint32_t)(b * p.pre_lowhw + (y >> 1) * p.pre_loww + (x >> 1));
#pragma unroll
        for (int t = 0; t < NT / 2; ++t) {
          const uint32_t vo = (row * (uint32_t)p.ld_pre + (uint32_t)(ncol + t * 32 + q * 8)) * 4u;
          bload16_async(dst[(i * (NT / 2) + t) * 2], vo, rsP);
          bload16_async_16(dst[(i * (NT / 2) + t) * 2 + 1], vo, rsP);
        }
      }
    }
  };
  if constexpr (PRE) issue_seed(t0, seed);
#pragma unroll
  for (int d = 0; d < DIST; ++d) issue_tile(t0 + d * tstep, d);
  if constexpr (PRE) wait_vmcnt_tie<(DIST - 1) * IPW>(seed[0], seed[1], seed[2], seed[3]); else wait_vmcnt<(DIST - 1) * IPW>();
  __syncthreads();

  f32x4 acc[MT][NT];
  // ---- epilogue of this wave's BM x WC strip, straight from the accumulators: with the permuted weight rows a lane owns 8
  // CONSECUTIVE channels (32t + 8q ..+7) of row i*16 + r -- 16 bytes, the four lanes q of a row 64 contiguous bytes, the 16 rows of
  // a store instruction 16 runs of 64 B (head planes: ONE run of 1 KB).  No LDS round trip.
  // What round 3 measured here (tools/probes/wreg_ablate.py, stamps of MOY_WREG_ABL=16): the kernel moves 14 GB and the write path
  // of a CU buffers far less than a tile, so a store instruction WAITS AT ISSUE for the memory system (1 300-1 800 of a 4 000-cycle
  // tile).  Anti-phase waves of a SIMD, and the four stores paced through the NEXT tile's MFMAs, both only moved that wait around:
  // the tile period is the drain time of its 48 KB (stores + DMA without any MFMA: 2.65-3.0 ms of the 3.15), i.e. the kernel runs
  // at ~85 % of the rate this device COPIES at (5.25 TB/s), not at the 6.9 TB/s it fills at.
  u32x4 pk[MT][NT / 2];
  // store k = i * (NT/2) + t of the tile whose first row is m0.  Rows past M fall outside the descriptor's range (or get an
  // out-of-range offset) and are dropped by the range check: the NST stores are unconditional (the counted waits rely on it).
  auto store_piece = [&](auto kc, int m0) {
    if constexpr (!LN) {
    constexpr int k = decltype(kc)::value, i = k / (NT / 2), t = k % (NT / 2);
    const bool live = !(ABL & 2);
    const int rr = i * 16 + r;
    if (p.plane_cols == 32 && p.c_rpb) {
      // head planes AND an output row remap (round 4: the value projection of ONE pyramid level straight from that level's own
      // tensor -- rows (b, i) of the level land at token b * S + off + i of the plane; `C` already points at row `off`): per-lane
      // row offsets into a descriptor that spans the plane
      T* cb = static_cast<T*>(p.C) + (int64_t)((ncol >> 5) + t) * p.plane_stride;
      const auto rsC = __builtin_amdgcn_make_buffer_rsrc(cb, 0, 0x7fffffffu, 0x00020000);       // (plane bytes < 1 GiB: checked by the host)
      const int m = m0 + rr;
      const int bq = (int)fdiv((uint32_t)m, p.fd_rpb);
      const uint32_t mo = (uint32_t)(bq * p.c_bstride + (m - bq * p.c_rpb));
      const uint32_t vo = (live && m < p.M) ? mo * 64u + q * 16 : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(pk[i][t], rsC, vo, 0, 0);
    } else if (p.plane_cols == 32) {
      // head planes [plane][row][32]: pair-group t of the wave IS plane 2w + t; a store instruction = 16 rows x 64 B = 1 KB contiguous
      T* cb = static_cast<T*>(p.C) + (int64_t)((ncol >> 5) + t) * p.plane_stride + (int64_t)(live ? m0 : 0) * 32;
      const int64_t left = live ? (int64_t)(p.M - m0) * 64 : 0;
      const auto rsC = __builtin_amdgcn_make_buffer_rsrc(cb, 0, (uint32_t)(left < 0x7fffffffLL ? left : 0x7fffffffLL), 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(pk[i][t], rsC, (uint32_t)rr * 64u + q * 16, 0, 0);
    } else if (p.c_rpb) {
      // output row remap (level-major token scatter of input_proj, head.py:1023-1028): the row offset is per lane; the host
      // guarantees that the whole C buffer is addressable with 31 bits
      const auto rsC = __builtin_amdgcn_make_buffer_rsrc(static_cast<T*>(p.C) + ncol, 0, 0x7fffffffu, 0x00020000);
      const int m = m0 + rr;
      const int bq = (int)fdiv((uint32_t)m, p.fd_rpb);
      const uint32_t mo = (uint32_t)(bq * p.c_bstride + (m - bq * p.c_rpb));
      const uint32_t vo = (live && m < p.M) ? (mo * (uint32_t)p.ldc + t * 32 + q * 8) * 2u : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(pk[i][t], rsC, vo, 0, 0);
    } else {
      const int pl = p.plane_cols ? ncol / p.plane_cols : 0;
      T* cb = static_cast<T*>(p.C) + (int64_t)pl * p.plane_stride + (int64_t)(live ? m0 : 0) * p.ldc + (ncol - pl * p.plane_cols);
      const int64_t left = live ? ((int64_t)(p.M - 1 - m0) * p.ldc + WC) * 2 : 0;
      const auto rsC = __builtin_amdgcn_make_buffer_rsrc(cb, 0, (uint32_t)(left < 0x7fffffffLL ? left : 0x7fffffffLL), 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(pk[i][t], rsC, (uint32_t)(rr * (int)p.ldc + q * 8 + t * 32) * 2u, 0, 0);
    }
    }
  };
  // one piece k = i * (NT/2) + t at a time: scale / shift / activation / pack, then its store -- 4 packed registers live, not NST * 4
  auto convert_store = [&](int m0) {
    if constexpr (!LN) {
    auto stage = [&](auto act_c) {
      constexpr int ACT = decltype(act_c)::value;
      auto piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value, i = k / (NT / 2), t = k % (NT / 2);
        float v[8];
        [[maybe_unused]] f32x4 sl0, sl1;
        if constexpr (LEAN) {
          sl0 = *reinterpret_cast<const f32x4*>(ssh + wave * WC + t * 32 + q * 8);
          sl1 = *reinterpret_cast<const f32x4*>(ssh + wave * WC + t * 32 + q * 8 + 4);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x;
          if constexpr (LEAN) x = acc[i][2 * t + (e >> 2)][e & 3] + (e < 4 ? sl0[e & 3] : sl1[e & 3]);
          else x = acc[i][2 * t + (e >> 2)][e & 3] * scr[t][e] + shr[t][e];
          if constexpr (ACT == MOY_ACT_SILU) x = siluf_(x);
          else if constexpr (ACT == MOY_ACT_RELU) x = fmaxf(x, 0.f);
          else if constexpr (ACT == MOY_ACT_SIGMOID) x = fast_sigmoid(x);
          v[e] = x;
        }
        pk[i][t] = u32x4{DT<T>::pack2(v[0], v[1]), DT<T>::pack2(v[2], v[3]), DT<T>::pack2(v[4], v[5]), DT<T>::pack2(v[6], v[7])};
        store_piece(kc, m0);
      };
      [&]<int... Ks>(std::integer_sequence<int, Ks...>) { (piece(std::integral_constant<int, Ks>{}), ...); }
      (std::make_integer_sequence<int, NST>{});
    };
    switch (p.act) {   // wave-uniform
      case MOY_ACT_SILU: stage(std::integral_constant<int, MOY_ACT_SILU>{}); break;
      case MOY_ACT_RELU: stage(std::integral_constant<int, MOY_ACT_RELU>{}); break;
      case MOY_ACT_SIGMOID: stage(std::integral_constant<int, MOY_ACT_SIGMOID>{}); break;
      default: stage(std::integral_constant<int, MOY_ACT_NONE>{}); break;
    }
    }
  };
  // fragment read: row i*16 + r, chunk (pn*4 + q) ^ r  ==  byte (i*16 + r)*KB + ((pn*64) ^ ((q ^ r) << 4))
  const int rbase = r * KB, xq = ((q ^ r) & SWZ) << 4;
  int buf = 0;
  // ABL bit 4: s_memtime stamps per phase, summed over the tiles of block 0 (waves 0 and 4) and left at the head of C (garbage there)
  uint64_t tph[5] = {0, 0, 0, 0, 0}, tlast = 0;
  auto stamp = [&](int k) {
    if constexpr (ABL & 16) {
      const uint64_t t = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tph[k] += t - tlast;
      tlast = t;
    }
  };
  if constexpr (ABL & 16) { tlast = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  for (int it = 0; it < n_mine; ++it) {
    const int tile = t0 + it * tstep;
    u32x4 seed_next[PRE ? 4 : 1];
    if constexpr (PRE) {
      // acc[i][2t], acc[i][2t+1] = the lane's channels 32t + 8q + 0..3 / + 4..7 of row i*16 + r: the two halves of its seed row
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_bit_cast(f32x4, seed[(i * (NT / 2) + (j >> 1)) * 2 + (j & 1)]);
      issue_seed(tile + tstep, seed_next);
    }
    {
      int nbuf = buf + DIST; if (nbuf >= NBUF) nbuf -= NBUF;
      if constexpr (!(ABL & 4)) issue_tile(tile + DIST * tstep, nbuf);
    }
    stamp(0);
    const unsigned char* As = smem + buf * TILE_BYTES;
    if constexpr (!PRE) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // fragments of panel pn+1 are requested before the MFMAs of panel pn (counted lgkmcnt): with one wave per SIMD nothing
    // else hides the LDS latency
    u32x4 af[2][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[0][i] = *reinterpret_cast<const u32x4*>(As + (rbase + xq + i * 16 * KB));
    auto panel = [&](auto pc) {
      constexpr int pn = decltype(pc)::value;
      if constexpr (pn + 1 < KP) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[(pn + 1) & 1][i] = *reinterpret_cast<const u32x4*>(As + (rbase + (((pn + 1) * 64) ^ xq) + i * 16 * KB));
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs (the scheduler otherwise sinks them back)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if constexpr (ABL & 1) { if (pn == 0) acc[i][j] = __builtin_bit_cast(f32x4, wf[j][i] ^ af[0][i]); }
          else acc[i][j] = mfma16<T>(acc[i][j], wf[j][pn], af[pn & 1][i]);
        }
      __builtin_amdgcn_sched_barrier(0);
    };
    [&]<int... Ps>(std::integer_sequence<int, Ps...>) { (panel(std::integral_constant<int, Ps>{}), ...); }
    (std::make_integer_sequence<int, KP>{});

    stamp(1);
    if constexpr (LN) {
      // ---- score mode: row statistics of v = acc * scale + shift (masked token rows: v = shift, i.e. a zero A row) and the
      // narrow head: every lane leaves its partial sums in LDS, wave 0 adds the 16 partials of each row after ONE barrier
      const int m0 = tile * BM;
      bool mk[MT];
      {
        const int mrem = p.a_mask ? m0 - (int)fdiv((uint32_t)m0, p.fd_mask) * p.mask_period : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          mk[i] = false;
          if (p.a_mask) {
            int idx = mrem + i * 16 + r;
            if (idx >= p.mask_period) idx -= p.mask_period;
            mk[i] = ((mbits[idx >> 5] >> (idx & 31)) & 1u) == 0;
          }
        }
      }
      {
        float s1[MT], s2[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = wave * 64 + j * 16 + q * 4;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(ssc + col), sh = *reinterpret_cast<const f32x4*>(ssh + col);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const f32x4 v = mk[i] ? sh : acc[i][j] * sc + sh;
            acc[i][j] = v;
            s1[i] += (v.x + v.y) + (v.z + v.w);
            s2[i] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          float* pr = P + (i * 16 + r) * PROW + wave * 4 + q;
          pr[0] = s1[i];
          pr[16] = s2[i];
        }
      }
      for (int c = 0; c < p.dot_n; ++c) {     // one class at a time (rolled): the class count does not cost registers
        float dd[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) dd[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4 gv = *reinterpret_cast<const f32x4*>(gw + c * 256 + wave * 64 + j * 16 + q * 4);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const f32x4 v = acc[i][j];
            dd[i] += (v.x * gv.x + v.y * gv.y) + (v.z * gv.z + v.w * gv.w);
          }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) P[(i * 16 + r) * PROW + 32 + c * 16 + wave * 4 + q] = dd[i];
      }
      __syncthreads();
      if (wave == 0) {
        const int row = lane < BM ? lane : 0;
        const float* pr = P + row * PROW;
        auto sum16 = [](const float* x) {   // the 16 partials of a row quantity: 4 waves x 4 lane groups
          const f32x4 a = *reinterpret_cast<const f32x4*>(x), b = *reinterpret_cast<const f32x4*>(x + 4);
          const f32x4 c = *reinterpret_cast<const f32x4*>(x + 8), d = *reinterpret_cast<const f32x4*>(x + 12);
          const f32x4 t = (a + b) + (c + d);
          return (t.x + t.y) + (t.z + t.w);
        };
        const float mean = sum16(pr) * (1.0f / 256.0f);
        const float var = fmaxf(sum16(pr + 16) * (1.0f / 256.0f) - mean * mean, 0.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const int64_t nbytes = (int64_t)p.M * p.dot_n * 4;
        const auto rsD = __builtin_amdgcn_make_buffer_rsrc(p.dot_out, 0, (uint32_t)(nbytes < 0x7fffffffLL ? nbytes : 0x7fffffffLL), 0x00020000);
        uint32_t off = lane < BM ? (uint32_t)((m0 + row) * p.dot_n) * 4u : 0x80000000u;   // rows >= M: out of range, dropped
        if (p.run_levels) off = (lane < BM && m0 + row < p.run_mv) ? (uint32_t)(run_row(m0 + row) * p.dot_n) * 4u : 0x80000000u;
#pragma unroll
        for (int c = 0; c < WREG_MAXDOT; ++c)
          if (c < p.dot_n) {
            const float sdot = rstd * (sum16(pr + 32 + c * 16) - mean * GB[c]) + GB[4 + c];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sdot), rsD, off + c * 4, 0, 0);
          }
      }
      // Wave 0's few score stores are not counted below: its wait then also retires that many of the youngest DMA pieces.
    } else {
      convert_store(tile * BM);
    }
    stamp(2);
    // tile it+1 must have landed; the younger DMA tiles and the stores issued since stay in flight
    if constexpr (PRE) {
      // queue, oldest first: DMA(it+1), stores(it-1), seed(it+1), DMA(it+2), stores(it): the next tile's seed AND patch are needed
      wait_vmcnt_tie<IPW + NST>(seed_next[0], seed_next[1], seed_next[2], seed_next[3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) seed[k] = seed_next[k];
    } else if constexpr (ABL & 4) wait_vmcnt<NST>();
    else wait_ring_tile<DIST, IPW, NST>(it);
    stamp(3);
    __syncthreads();
    stamp(4);
    if (++buf == NBUF) buf = 0;
  }
  if constexpr (ABL & 16) {
    if (blockIdx.x == 0 && (wave == 0 || wave == NW / 2) && lane == 0) {
      uint64_t* d = static_cast<uint64_t*>(p.C) + (wave ? 8 : 0);
      for (int k = 0; k < 5; ++k) d[k] = tph[k];
      d[5] = (uint64_t)n_mine;
    }
  }
  wait_vmcnt<0>();   // clamped over-fetch tiles still target this block's LDS
}

static int wreg_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

template <typename T, int BM, int NBUF, int OCC, bool LN = false, int NW = 4, int WC = 64, int K = 256, int ABL = 0, bool PRE = false>
static int launch_wreg(WregParams& p, hipStream_t st) {
  const int lds = wreg_lds_bytes<BM, NBUF, LN, NW, WC, K>() + (LN && p.a_mask ? ((p.mask_period + 31) / 32) * 4 : 0);
  auto kern = gemm_wreg_kernel<T, BM, NBUF, OCC, LN, NW, WC, K, ABL, PRE>;
  const int slots = cu_limit(wreg_num_cus()) / 8 * OCC;          // resident blocks per XCD
  p.ntiles = ((p.run_levels ? p.run_mv : p.M) + BM - 1) / BM;
  p.ngroups = p.N / (NW * WC);
  p.lanes = slots / p.ngroups;
  if (p.lanes < 1) return MOY_ENOSYS;
  if (plan_only(MOY_KERNEL_WREG)) return MOY_OK;          // moy_gemm_query: the dispatch without the launch (and without touching the attribute statics)
  static int attr_lds = 0;
  if (lds > 65536 && lds > attr_lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return MOY_ELAUNCH;
    attr_lds = lds;
  }
  hipLaunchKernelGGL(kern, dim3(8 * p.lanes * p.ngroups), dim3(64 * NW), lds, st, p);
  return launch_status();
}

// Row-tile height of the 32-columns-per-wave forms: MOY_WREG_BM=32|64|128 overrides the per-shape choice (A/B runs; a height whose
// activation ring does not fit falls back to the next smaller one).
#if MOY_DIAG
static int wreg_bm_env() {
  static const int v = knob("MOY_WREG_BM", 0);
  return v;
}
#endif

// 8 waves x 32 columns = 256 columns per block, any K in {128, 256, 384, 512}: the 1x1 convs / input_proj with 256 outputs
template <typename T>
static int launch_wreg_k(WregParams& p, int K, hipStream_t st) {
#if MOY_DIAG
  const int bm = wreg_bm_env();
  if (K == 128 && bm == 128) return launch_wreg<T, 128, 3, 1, false, 8, 32, 128>(p, st);
  if (K == 128 && bm == 64) return launch_wreg<T, 64, 3, 1, false, 8, 32, 128>(p, st);
  if (K == 256 && bm >= 64) return launch_wreg<T, 64, 3, 1, false, 8, 32, 256>(p, st);
  if (K == 384 && bm >= 64) return launch_wreg<T, 64, 3, 1, false, 8, 32, 384>(p, st);
#endif
  switch (K) {
    case 128: return launch_wreg<T, 32, 3, 1, false, 8, 32, 128>(p, st);
    case 256: return launch_wreg<T, 32, 3, 1, false, 8, 32, 256>(p, st);
    case 384: return launch_wreg<T, 32, 3, 1, false, 8, 32, 384>(p, st);
    case 512: return launch_wreg<T, 32, 3, 1, false, 8, 32, 512>(p, st);
    default: return MOY_ENOSYS;
  }
}

// 4 waves x 32 columns = 128 columns per block, two blocks per CU: the 1x1 convs with 128 outputs (C2f cv1 / cv2 at the P3 level)
template <typename T>
static int launch_wreg_n128(WregParams& p, int K, hipStream_t st) {
#if MOY_DIAG
  const int bm = wreg_bm_env();
  if (K == 128 && bm >= 64) return launch_wreg<T, 64, 3, 2, false, 4, 32, 128>(p, st);
  if (K == 192 && bm >= 64) return launch_wreg<T, 64, 3, 2, false, 4, 32, 192>(p, st);
  if (K == 256 && bm >= 64) return launch_wreg<T, 64, 2, 2, false, 4, 32, 256>(p, st);
#endif
  switch (K) {
    case 128: return launch_wreg<T, 32, 3, 2, false, 4, 32, 128>(p, st);
    case 192: return launch_wreg<T, 32, 3, 2, false, 4, 32, 192>(p, st);
    case 256: return launch_wreg<T, 32, 3, 2, false, 4, 32, 256>(p, st);
    default: return MOY_ENOSYS;
  }
}

// Eligibility + dispatch; MOY_ENOSYS = not this kernel's shape (moy_gemm falls through to the tiled kernel).
int gemm_wreg_try(const moy_gemm_args* a, hipStream_t st) {
  static const int mode = knob("MOY_GEMM_WREG", 1);                    // MOY_GEMM_WREG=0 switches the kernel off (A/B runs, bit-identity test)
  if (!mode) return MOY_ENOSYS;
  if (a->dtype != MOY_BF16 && a->dtype != MOY_F16) return MOY_ENOSYS;
  static const int n128 = knob("MOY_WREG_N128", 1);                    // MOY_WREG_N128=0: N = 128 stays on the tiled kernel (A/B runs)
  const bool is128 = a->N == 128 && n128 && (a->K == 128 || a->K == 192 || a->K == 256);
  if (a->ksize != 1 || (!is128 && (a->N % 256 || a->N / 256 > 16))) return MOY_ENOSYS;
  if (a->K != 128 && a->K != 256 && a->K != 384 && a->K != 512 && !(is128 && a->K == 192)) return MOY_ENOSYS;
  if (a->A2 || a->a_rows || a->R || a->out_f32) return MOY_ENOSYS;
  // score mode: LayerNorm + narrow head with NO feature output (C == NULL; moy_gemm documents it); the normalised rows
  // themselves are the tiled kernel's job
  const bool score = a->ln_g && !a->C;
  if (score) {
    if ((a->K != 256 && a->K != 128) || a->N != 256 || a->dot_n < 1 || a->dot_n > WREG_MAXDOT || a->act != MOY_ACT_NONE || a->c_rows_per_batch) return MOY_ENOSYS;
    if (a->a_mask && (a->mask_period < 64 || a->mask_period > 262144)) return MOY_ENOSYS;
    if (a->run_levels) {
      if (a->a_mask || a->run_levels > 4 || a->run_period <= 0 || (a->M % a->run_period) || (int64_t)a->M * a->lda * 2 > 0xffffffffLL) return MOY_ENOSYS;
      if (a->run_a_period < 0 || (a->run_a_period && a->run_levels != 1)) return MOY_EINVAL;     // a separate A numbering: one level per launch
      for (int l = 0; l < a->run_levels; ++l)
        if (a->run_len[l] <= 0 || a->run_rows[l] <= 0 || a->run_pitch[l] < a->run_len[l] || a->run_tok0[l] < 0 ||
            a->run_tok0[l] + (a->run_rows[l] - 1) * a->run_pitch[l] + a->run_len[l] > a->run_period)
          return MOY_EINVAL;
    }
  } else if (a->ln_g || a->a_mask || a->dot_n || !a->C || a->run_levels) {
    return MOY_ENOSYS;
  }
  static const int preok = knob("MOY_WREG_PRE", 1);                   // MOY_WREG_PRE=0: seeded launches stay on the tiled kernel (A/B runs)
  const bool seeded = a->pre != nullptr;   // (shape and alignment of the seed were validated by moy_gemm)
  if (seeded) {
    const bool form = (is128 && a->K == 128) || (!is128 && a->K == 256);
    const int64_t seed_bytes = (int64_t)(a->M / (a->pre_h * a->pre_w)) * (a->pre_h / 2) * (a->pre_w / 2) * a->ld_pre * 4;
    if (!preok || score || !form || a->plane_cols || a->c_rows_per_batch || seed_bytes > 0x7fffffffLL) return MOY_ENOSYS;
  }
  static const int kgen = knob("MOY_WREG_KGEN", 1);                    // MOY_WREG_KGEN=0: only the K = 256 forms (A/B runs)
  // the value-projection form (8 waves x 64 columns, head planes of 32) also exists for K = 128 and with the output row remap
  // (round 4: one launch per pyramid level, straight from that level's tensor)
  const bool valueform = a->plane_cols == 32 && (a->K == 256 || a->K == 128) && (a->N % 512) == 0 && !is128 && !seeded && !score;
  const bool general = !valueform && !score && (a->K != 256 || a->c_rows_per_batch || is128 || seeded);       // the 32-columns-per-wave forms
  if (general && (!kgen || a->plane_cols)) return MOY_ENOSYS;
  if (a->c_rows_per_batch) {               // per-lane 32-bit byte offsets into the remapped C
    const int64_t rows = ((int64_t)a->M / a->c_rows_per_batch + 1) * a->c_batch_stride;
    if (rows * a->ldc * 2 > 0x7fffffffLL) return MOY_ENOSYS;
    if (a->plane_cols && !valueform) return MOY_ENOSYS;
  }
  // persistent row-tile walk: needs many tiles per block.  The 32-columns-per-wave forms (1x1 convs) from 65536 rows; round 6: the value
  // form (8 waves x 64 columns, N >= 1024: six or more column groups share every activation tile) and the score form from 8192 rows --
  // at a few frames per step (the small-batch leg) they are the two longest launches of the plan and the tiled kernel runs them at
  // half this kernel's rate (value projection of 4 frames: 75 us tiled).  Bit-identical to the tiled kernel either way (tests).
  if (a->M < (((valueform && a->N >= 1024) || score) ? 8192 : 65536)) return MOY_ENOSYS;
  if ((a->lda % 8) || !aligned16(a->A) || !aligned16(a->W)) return MOY_ENOSYS;
  if (!score && ((a->ldc % 8) || !aligned16(a->C))) return MOY_ENOSYS;
  if (a->plane_cols && a->plane_cols != 32 && (a->plane_cols % 64)) return MOY_ENOSYS;
  if (a->plane_cols == 32 && (a->ldc != 32 || a->plane_stride * 2 > 0x3fffffffLL)) return MOY_ENOSYS;
  WregParams p{};
  p.A = a->A; p.lda = a->lda; p.W = a->W; p.scale = a->scale; p.shift = a->shift; p.act = a->act;
  p.C = a->C; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.ngroups = a->N / 256;
  p.plane_cols = a->plane_cols; p.plane_stride = a->plane_stride;
  p.c_rpb = a->c_rows_per_batch; p.c_bstride = a->c_batch_stride; p.fd_rpb = make_fastdiv(p.c_rpb > 0 ? p.c_rpb : 1);
  if (seeded) {
    p.pre = a->pre; p.ld_pre = a->ld_pre; p.pre_w = a->pre_w; p.pre_hw = a->pre_h * a->pre_w;
    p.pre_loww = a->pre_w / 2; p.pre_lowhw = (a->pre_h / 2) * (a->pre_w / 2);
    p.fd_pre_hw = make_fastdiv((uint32_t)p.pre_hw); p.fd_pre_w = make_fastdiv((uint32_t)p.pre_w);
    if (is128) return a->dtype == MOY_BF16 ? launch_wreg<bf16_t, 32, 3, 2, false, 4, 32, 128, 0, true>(p, st) : launch_wreg<f16_t, 32, 3, 2, false, 4, 32, 128, 0, true>(p, st);
    return a->dtype == MOY_BF16 ? launch_wreg<bf16_t, 32, 3, 1, false, 8, 32, 256, 0, true>(p, st) : launch_wreg<f16_t, 32, 3, 1, false, 8, 32, 256, 0, true>(p, st);
  }
  if (is128) return a->dtype == MOY_BF16 ? launch_wreg_n128<bf16_t>(p, a->K, st) : launch_wreg_n128<f16_t>(p, a->K, st);
  if (general) return a->dtype == MOY_BF16 ? launch_wreg_k<bf16_t>(p, a->K, st) : launch_wreg_k<f16_t>(p, a->K, st);
  if (score) {
    p.ln_g = a->ln_g; p.ln_b = a->ln_b; p.dot_w = a->dot_w; p.dot_b = a->dot_b; p.dot_out = a->dot_out; p.dot_n = a->dot_n;
    p.a_mask = a->a_mask; p.mask_period = a->mask_period; p.fd_mask = make_fastdiv(a->mask_period > 0 ? a->mask_period : 1);
    p.run_levels = a->run_levels;
    if (a->run_levels) {
      p.run_period = a->run_period;
      int nv = 0;
      for (int l = 0; l < 4; ++l) {
        const bool on = l < a->run_levels;
        p.run_cstart[l] = nv; p.run_len[l] = on ? a->run_len[l] : 1; p.run_tok0[l] = on ? a->run_tok0[l] : 0; p.run_pitch[l] = on ? a->run_pitch[l] : 1;
        p.fd_len[l] = make_fastdiv((uint32_t)p.run_len[l]);
        if (on) nv += a->run_len[l] * a->run_rows[l];
      }
      p.run_nv = nv; p.fd_nv = make_fastdiv((uint32_t)nv);
      p.run_mv = (a->M / a->run_period) * nv;
      p.run_a_period = a->run_a_period ? a->run_a_period : a->run_period;
      p.run_a_off = a->run_a_period ? a->run_a_off : 0;
    }
    if (a->K == 128) return a->dtype == MOY_BF16 ? launch_wreg<bf16_t, 32, 3, 2, true, 4, 64, 128>(p, st) : launch_wreg<f16_t, 32, 3, 2, true, 4, 64, 128>(p, st);
    return a->dtype == MOY_BF16 ? launch_wreg<bf16_t, 32, 3, 2, true>(p, st) : launch_wreg<f16_t, 32, 3, 2, true>(p, st);
  }
  static const int variant = knob("MOY_WREG_VARIANT", 0);
  // measured on the value projection (M = 1.3 M, N = 1536, bf16; tiled kernel 1951 us): 4 waves x 256 columns, BM 32 / 3 buffers /
  // 2 blocks per CU 1168 us; BM 64 / 3 buffers / 1 block per CU 1254 us; BM 64 / 2 buffers 1274 us; BM 32 / 4 buffers 1458 us (one
  // block per CU fits); 8 waves x 512 columns, BM 32 / 3 buffers, one block per CU: 6 % faster than the first (same device) -- half as
  // many blocks re-fetch an activation tile that has left the L2.  (Those variants left the tree in round 3; MOY_WREG_VARIANT=2
  // keeps the 4-wave form selectable for A/B runs.)
#if MOY_DIAG
  if (a->K == 128 && a->dtype == MOY_BF16) {      // A/B forms of the K = 128 value launch (MOY_WREG_V128; bf16 only)
    static const int v128 = knob("MOY_WREG_V128", 0);
    switch (v128) {
      case 3: return launch_wreg<bf16_t, 32, 5, 1, false, 8, 64, 128>(p, st);     // one block, ring of 5
      case 4: return launch_wreg<bf16_t, 64, 3, 1, false, 8, 64, 128>(p, st);     // 64-row tiles
      case 6: return launch_wreg<bf16_t, 64, 2, 1, false, 8, 64, 128>(p, st);     // 64-row tiles, ring of 2
      case 7: return launch_wreg<bf16_t, 64, 4, 1, false, 8, 64, 128>(p, st);     // 64-row tiles, ring of 4
      case 9: return launch_wreg<bf16_t, 32, 3, 1, false, 8, 64, 128>(p, st);     // 32-row tiles (the form before the measurement)
      case 10: return launch_wreg<bf16_t, 128, 3, 1, false, 8, 32, 128>(p, st);   // 32 columns per wave (one plane), 128-row tiles
      case 11: return launch_wreg<bf16_t, 64, 3, 1, false, 8, 32, 128>(p, st);    // 32 columns per wave, 64-row tiles
      default: break;
    }
  }
  if (a->K == 256 && a->dtype == MOY_BF16) {      // A/B forms of the K = 256 value launches (MOY_WREG_V256; bf16 only)
    static const int v256 = knob("MOY_WREG_V256", 0);
    switch (v256) {
      case 1: return launch_wreg<bf16_t, 64, 3, 1, false, 8, 32, 256>(p, st);     // 32 columns per wave (one plane), 64-row tiles
      case 2: return launch_wreg<bf16_t, 32, 3, 1, false, 8, 32, 256>(p, st);     // 32 columns per wave, 32-row tiles
      // (round 5: the forms that spilled -- two blocks per CU at K = 128 [1, 2, 5], 64-row tiles at K = 256 [3, 4: 0.86-0.89 ms against 0.62] -- left
      //  the tree: scratch loads and stores are vector-memory operations the counted waits of the ring do not know about)
      default: break;
    }
  }
#endif
  // round 4, measured (tools/probes/cu_share.py --value-only, 288 frames of the P3 level, 9.9 GB): 32-row tiles 2.53 ms, two blocks
  // per CU (spills) 5.4, ring of 5 2.53, 64-ROW TILES 1.93 ms = 5.1 TB/s -- a plane's run per tile is 4 KB instead of 2 KB and
  // the barrier / DMA-issue cadence per byte halves
  if (a->K == 128)      // (valueform: N % 512 == 0, head planes)
    return a->dtype == MOY_BF16 ? launch_wreg<bf16_t, 64, 3, 1, false, 8, 64, 128>(p, st) : launch_wreg<f16_t, 64, 3, 1, false, 8, 64, 128>(p, st);
  if (a->dtype == MOY_BF16) {
    if (variant == 2 || a->N % 512) return launch_wreg<bf16_t, 32, 3, 2>(p, st);
#if MOY_DIAG
    // timing-only builds of the value-projection form (tools/probes/wreg_ablate.py): see the kernel's ABL comment
    static const int abl = garbage_mode_env("MOY_WREG_ABL");
    switch (abl) {
      case 1: return launch_wreg<bf16_t, 32, 3, 1, false, 8, 64, 256, 1>(p, st);
      case 2: return launch_wreg<bf16_t, 32, 3, 1, false, 8, 64, 256, 2>(p, st);
      case 3: return launch_wreg<bf16_t, 32, 3, 1, false, 8, 64, 256, 3>(p, st);
      case 4: return launch_wreg<bf16_t, 32, 3, 1, false, 8, 64, 256, 4>(p, st);
      case 5: return launch_wreg<bf16_t, 32, 3, 1, false, 8, 64, 256, 5>(p, st);
      case 6: return launch_wreg<bf16_t, 32, 3, 1, false, 8, 64, 256, 6>(p, st);
      case 16: return launch_wreg<bf16_t, 32, 3, 1, false, 8, 64, 256, 16>(p, st);
      case 22: return launch_wreg<bf16_t, 32, 3, 1, false, 8, 64, 256, 22>(p, st);
      default: break;
    }
#endif
    return launch_wreg<bf16_t, 32, 3, 1, false, 8>(p, st);
  }
  if (a->N % 512) return launch_wreg<f16_t, 32, 3, 2>(p, st);
  return launch_wreg<f16_t, 32, 3, 1, false, 8>(p, st);
}

}  // namespace moy
