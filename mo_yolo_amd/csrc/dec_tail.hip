// The row-wise tail of a decoder layer in ONE launch (MOTRDecoderLayer.forward after the deformable sampling,
// nn/modules/transformer.py:642-652, and the box refinement of the decoder loop, :705-709):
//   e2  = LayerNorm2(samp . Wp^T + bp + e1)                      cross_attn.output_proj + dropout(identity) + norm2
//   e3  = LayerNorm3(relu(e2 . W1^T + b1) . W2^T + b2 + e2)      linear1 / relu / linear2 + norm3   -> layer output
//   box = sigmoid(MLP3(e3) + inverse_sigmoid(ref))               dec_bbox_head[i], refined reference boxes
// As separate launches this was output_proj+LN (27 us), linear1 (35), linear2+LN (35) and the box head (22) per layer at
// 96 frames x 300 rows: latency chains over a problem that does not fill the chip, with e2, the 1024-wide hidden activation and
// e3 making round trips through memory.  Structure = csrc/mlp_head.hip, extended:
//   * a block owns 128 rows for the whole chain; the current activation tile lives in LDS (two 64 KB tiles, XOR-swizzled 512-byte
//     rows); each of the 8 waves keeps its 32 output columns of the current weight chunk in registers (MFMA A operand), requested
//     panel by panel 7/8 of a product ahead, from weights in MFMA-fragment order when the caller says so (round 5: a row-major
//     fragment load is 32 isolated 16-byte pieces per instruction);
//   * no epilogue reads global memory (round 5): the fp32 vectors are staged in LDS once per block, the residual tile sits in the
//     buffer the LayerNorm output replaces in place -- a load issued in an epilogue returned only behind the next product's weights;
//   * the FFN runs in d_ffn/256 chunks: h_c = relu(e2 . W1_c^T) -> LDS tile, acc3 += h_c . W2[:, c]^T, so the 1024-wide hidden
//     activation never exists outside LDS; every intermediate is rounded to the storage type exactly where the separate launches
//     stored it, and every product uses their k order;
//   * LayerNorm: one-pass row statistics (sum, sum of squares) -- lane partials, xor-shuffles over the four lane groups, per-wave
//     partials in LDS, one barrier;
//   * the layer output leaves through its LDS tile in whole 512-byte rows.
#include "common.hpp"

namespace moy {

template <typename T>
__device__ __forceinline__ f32x4 tail_mfma(f32x4 acc, u32x4 w, u32x4 a);
template <>
__device__ __forceinline__ f32x4 tail_mfma<bf16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 tail_mfma<f16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}

__device__ __forceinline__ float tail_inv_sigmoid(float x) {   // nn/modules/utils.py:34-38, eps 1e-5
  x = fminf(fmaxf(x, 0.f), 1.f);
  return logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
}

#ifndef MOY_TAIL_NW
#define MOY_TAIL_NW 8       // waves of a block: 8 (two per SIMD, 32 output columns each) or 4 (one per SIMD, 64 columns, 512 registers)
#endif
#ifndef MOY_TAIL_PD
#define MOY_TAIL_PD 1       // steps of activation fragments requested ahead of the MFMAs that use them
#endif
constexpr int TAIL_BM = 128, TAIL_NW = MOY_TAIL_NW;     // TAIL_BM: rows of a block at bench scale; round 6: 64 / 32-row blocks for small M
#ifndef MOY_TAIL_RS
#define MOY_TAIL_RS 8       // row parts of a product's step walk: 8 = one 16-row fragment per step.  With 4 (two fragments in flight per buffer) the
#endif                      // kernel needed 38 registers more than it has, and one of the spilled values was reloaded INSIDE the FFN loop: a scratch
                            // load is a vector-memory operation, results return in order, so that reload waited for the weight prefetch of the next product

// fp32 vectors of the chain staged in LDS once per block (round 5): [bp | ln2_g | ln2_b | b2 | ln3_g | ln3_b | c0 | b1[0 .. 2048)]
constexpr int TAIL_V_BP = 0, TAIL_V_LN2G = 256, TAIL_V_LN2B = 512, TAIL_V_B2 = 768, TAIL_V_LN3G = 1024, TAIL_V_LN3B = 1280, TAIL_V_C0 = 1536,
              TAIL_V_B1 = 1792, TAIL_MAX_FFN = 2048, TAIL_V_N = 1792 + TAIL_MAX_FFN;
constexpr int tail_lds(int bm) { return 2 * bm * 512 + bm * TAIL_NW * 16 + TAIL_V_N * 4; }
constexpr int TAIL_LDS = tail_lds(TAIL_BM);

// BM_ (round 6): rows of a block.  128 at bench scale; at small M (a few frames per step: the small-batch leg) the chain of a block --
// eleven dependent products, each behind 128 KB of weights -- is the kernel's whole duration, and 10 blocks of 128 rows leave 246
// compute units idle: 32-row blocks quarter the matrix work of a block's chain (every row's arithmetic is independent of the block height:
// the same bits).  What a small block's chain then costs is its WEIGHT STREAM: 1.8 MB per block through one CU's 64 B/clk, fourteen
// dependent products at ~2.6 us each (1 200 rows: 36-39 us).  Measured and dropped (round 6): a second weight set in registers, the next
// product's slice requested a whole product ahead -- 43 us: the request is as long as two products, one product of distance hides none of it.
template <typename T, int ABL = 0, int BM_ = TAIL_BM>       // ABL 1: timing-only build that keeps the FIRST weight chunk for every product (MOY_TAIL_ABL=1; results garbage)
                                         // ABL 2: s_memtime stamps of wave 0 per phase, written over the head of `out` (MOY_TAIL_ABL=2; tools/probes/tail_diag.py)
__global__ __launch_bounds__(64 * TAIL_NW) void decoder_tail_kernel(const moy_decoder_tail_args p) {
  constexpr int BM = BM_, NW = TAIL_NW, NTHR = 64 * NW, MT = BM / 16, NT = 16 / NW, WC = 256 / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* XA = smem;
  unsigned char* XB = smem + BM * 512;
  float* P = reinterpret_cast<float*>(smem + 2 * BM * 512);      // [BM][NW][4] floats: LayerNorm / box-head partials
  const float* V = reinterpret_cast<const float*>(smem + 2 * BM * 512 + BM * NW * 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int ntiles = (p.M + BM - 1) / BM;
  unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  unsigned long long wt[6] = {0, 0, 0, 0, 0, 0};        // ABL 2: every wave's clock through FFN chunk 1 (block 0, its last tile)
  auto stamp = [&](int i) {
    if constexpr (ABL >= 2) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long tt = __builtin_amdgcn_s_memtime();
      ph[i] += tt - tprev;
      tprev = tt;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (ABL >= 2) tprev = __builtin_amdgcn_s_memtime();

  // This wave's 32 output rows of the current [*, pitch] weight matrix (k columns koff .. koff+255) as eight 32-wide panels:
  // wa[0] = panels 0-3, wa[1] = panels 4-7.  Round 2 fetched a whole chunk after the product that used the previous one (two chunks
  // live = spills) and measured 36 % of the kernel waiting for those loads; round 3 re-requested the registers by K HALVES the moment
  // a half had been consumed; round 5 does it panel by panel (below).
  u32x4 wa[2][NT][4];
  struct WSrc { const void* W; int row0, pitch, koff; };
  bool w_loaded = false;

  // Round 5: one 32-wide PANEL at a time.  A product now walks its panels outermost (panel pn over every row part, then pn + 1; every
  // accumulator still sees its panels in the order 0..7, so results are unchanged) and the moment a panel has been consumed its registers
  // are re-requested with that panel of the NEXT product: every request is 7/8 of a product + the epilogue ahead of its use (by halves it
  // was 1/2), and the requests are spread over the product instead of arriving at the texture unit in two bursts of 16 per wave.  The
  // per-wave clocks (MOY_TAIL_ABL=2) had shown the second wave of every SIMD finishing a product at 6.1-7.7 k cycles against 3.1-3.6 k of
  // the first, and 5.0 k with the weight traffic ablated: its weights, served after the first waves', arrived late.
  const uint32_t voff_256 = (uint32_t)((r * 256 + q * 8) * 2), voff_ffn = (uint32_t)((r * p.d_ffn + q * 8) * 2);
  auto req_panel = [&](int pn, const WSrc& w) {
    if constexpr (ABL == 1 || ABL == 3) { if (w_loaded) return; }
    if (!w.W) return;
    // buffer loads: the matrix as the descriptor's base, the lane part of the address one 32-bit register per row pitch (r * pitch + q * 8),
    // everything else wave-uniform in the scalar offset -- no 64-bit per-lane pointers held across the products
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w.W), 0, 0x7fffffff, 0x00020000);
    // w_packed: the matrix in MFMA-fragment order (include/moyolo.h) -- this wave's panel is 2 KB contiguous, 16 bytes per lane.  Row-major,
    // a request is 32 rows x 64 bytes: 37.5 GB/s per CU against 132 (tools/probes/l2_segments.hip), and the weights are 128 KB per product
    const uint32_t voff = p.w_packed ? (uint32_t)lane * 16u : (w.pitch == 256 ? voff_256 : voff_ffn);
    const uint32_t so = p.w_packed ? (uint32_t)((((w.row0 >> 5) + wave) * (w.pitch >> 5) + (w.koff >> 5) + pn) * 2048)
                                   : (uint32_t)(((w.row0 + wave * WC) * w.pitch + w.koff + pn * 32) * 2);
    const uint32_t sj = p.w_packed ? 1024u : (uint32_t)(16 * w.pitch * 2);
#pragma unroll
    for (int j = 0; j < NT; ++j)
      wa[pn >> 2][j][pn & 3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so + (uint32_t)j * sj, 0));
  };
  static_assert(NT == 2 && WC == 32, "the fragment order is defined for 32-row groups of two 16-row halves");
  const int lbase = r * 512 + ((q ^ r) << 4);      // fragment of row i*16 + r, chunk (pn*4 + q) ^ r
  // rows in two halves of 64: 16 fragment registers live instead of 32
  auto gemm_acc = [&](const unsigned char* As, f32x4 (&acc)[MT][NT], const WSrc& next) {
    constexpr int RS = MOY_TAIL_RS < MT ? MOY_TAIL_RS : MT, HM = MT / RS, NSTEP = 8 * RS;   // step s = (panel s / RS, row part s % RS), panel outermost
    constexpr int PD = MOY_TAIL_PD;
    u32x4 af[PD + 1][HM];                                 // the fragments of step s+PD are requested before the MFMAs of step s
    auto frag = [&](int s_, u32x4 (&buf)[HM]) {
      const int pn = s_ / RS, rp = s_ % RS;
#pragma unroll
      for (int i = 0; i < HM; ++i) buf[i] = *reinterpret_cast<const u32x4*>(As + ((lbase ^ (pn * 64)) + (rp * HM + i) * 8192));
    };
#pragma unroll
    for (int s_ = 0; s_ < PD; ++s_) frag(s_, af[s_]);
#pragma unroll
    for (int s_ = 0; s_ < NSTEP; ++s_) {
      const int pn = s_ / RS, rp = s_ % RS, kh = pn >> 2, p4 = pn & 3;
      if (s_ + PD < NSTEP) frag(s_ + PD, af[(s_ + PD) % (PD + 1)]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < HM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[rp * HM + i][j] = tail_mfma<T>(acc[rp * HM + i][j], wa[kh][j][p4], af[s_ % (PD + 1)][i]);
      __builtin_amdgcn_sched_barrier(0);
      if (rp == RS - 1) { req_panel(pn, next); __builtin_amdgcn_sched_barrier(0); }        // panel pn consumed: its registers take the next product's
      if (s_ == NSTEP - 1) { if constexpr (ABL == 1 || ABL == 3) w_loaded = true; }
    }
  };
  auto zero = [&](f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // value (row i*16 + r, columns n .. n+3) of a tile
  auto tile_ptr = [&](unsigned char* X, int row, int n) { return X + row * 512 + (((n >> 3) ^ (row & 15)) << 4) + (n & 7) * 2; };
  auto put4 = [&](unsigned char* X, int row, int n, f32x4 v) {
    *reinterpret_cast<u32x2*>(tile_ptr(X, row, n)) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
  };
  auto get4 = [&](unsigned char* X, int row, int n) {
    const u32x2 w = *reinterpret_cast<const u32x2*>(tile_ptr(X, row, n));
    return f32x4{DT<T>::lo(w.x), DT<T>::hi(w.x), DT<T>::lo(w.y), DT<T>::hi(w.y)};
  };
  // LayerNorm over the 256 columns of every row of v (in place): v <- (v - mean) * rstd * g + b
  auto layer_norm = [&](f32x4 (&v)[MT][NT], const float* g, const float* be) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x4 x = v[i][j];
        s1 += (x.x + x.y) + (x.z + x.w);
        s2 += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
      }
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (q == 0) *reinterpret_cast<float2*>(P + ((i * 16 + r) * NW + wave) * 2) = float2{s1, s2};
    }
    __syncthreads();
    f32x4 gg[NT], bb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * WC + j * 16 + q * 4;
      gg[j] = *reinterpret_cast<const f32x4*>(g + n);             // (LDS: the staged vectors)
      bb[j] = *reinterpret_cast<const f32x4*>(be + n);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float* pr = P + (i * 16 + r) * NW * 2;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w += 2) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(pr + w * 2);
        s1 += t.x + t.z;
        s2 += t.y + t.w;
      }
      const float mean = s1 * (1.0f / 256.0f);
      const float var = fmaxf(s2 * (1.0f / 256.0f) - mean * mean, 0.0f);
      const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
      for (int j = 0; j < NT; ++j) v[i][j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
    }
  };

  // ---- P0: sampling output tile -> XA, residual tile e1 -> XB (e2 overwrites it in place), the fp32 vectors -> V, output_proj
  //      weights -> registers.  Round 5: NO epilogue of the chain reads global memory any more.  Vector-memory results return in issue
  //      order, so a bias / residual / LayerNorm load issued in an epilogue came back only after BOTH weight halves of the NEXT product,
  //      requested just before it (32 x 16-byte loads per lane): the stamps (MOY_TAIL_ABL=2) showed the epilogues at twice the time of
  //      the products they follow.  The weight requests now overlap the epilogues instead of gating them.
  constexpr int NREF = (BM * 4 + NTHR - 1) / NTHR; // (row, output) pairs of the box refinement per thread (blocks of fewer than 128 rows: some threads have none)
  constexpr int NLD = BM * 32 / NTHR;              // 16-byte pieces of a 128 x 256 tile per thread
  if ((int)blockIdx.x >= ntiles) return;
  const int m0 = blockIdx.x * BM;
  // ---- P0: sampling output tile -> XA, residual tile e1 -> XB (e2 overwrites it in place), the fp32 vectors -> V, output_proj weights ->
  //      registers.  NO epilogue of the chain reads global memory (round 5): vector-memory results return in issue order, so a bias /
  //      residual / LayerNorm load issued in an epilogue came back only after the weights of the NEXT product requested just before it;
  //      the stamps (MOY_TAIL_ABL=2) showed the epilogues at twice the time of the products they follow.
  //      (Measured and dropped: the block persistent over its row tiles with the next tile's rows requested into registers during the
  //      box head -- the cold start of a tile is 17-18 k of its 136 k cycles -- : 32-64 more registers live across the tile loop, the
  //      allocator spilled 130-190 values and their reloads queue behind the weight requests: 264 us against 207.)
  float ref_in_r[NREF];
  {
    const T* Xg = static_cast<const T*>(p.samp);
    const T* Eg = static_cast<const T*>(p.e1);
    u32x4 xr[NLD], er[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      const int m = min(m0 + row, p.M - 1);
      xr[k] = *reinterpret_cast<const u32x4*>(Xg + (int64_t)m * p.ld_samp + c * 8);
    }
    {
      const WSrc w0{p.Wp, 0, 256, 0};
#pragma unroll
      for (int pn = 0; pn < 8; ++pn) req_panel(pn, w0);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      const int m = min(m0 + row, p.M - 1);
      er[k] = *reinterpret_cast<const u32x4*>(Eg + (int64_t)m * p.ld_e1 + c * 8);
    }
    {
      float* Vw = const_cast<float*>(V);
#pragma unroll
      for (int v = 0; v < 7; v += NW) {
        const int vv = v + wave;
        const float* src = vv == 0 ? p.bp : vv == 1 ? p.ln2_g : vv == 2 ? p.ln2_b : vv == 3 ? p.b2 : vv == 4 ? p.ln3_g : vv == 5 ? p.ln3_b : p.c0;
        if (vv < 7) *reinterpret_cast<f32x4*>(Vw + vv * 256 + lane * 4) = *reinterpret_cast<const f32x4*>(src + lane * 4);
      }
      for (int i = tid * 4; i < p.d_ffn; i += NTHR * 4) *reinterpret_cast<f32x4*>(Vw + TAIL_V_B1 + i) = *reinterpret_cast<const f32x4*>(p.b1 + i);
    }
#pragma unroll
    for (int k = 0; k < NREF; ++k) {
      const int m = m0 + ((tid + k * NTHR) >> 2);
      ref_in_r[k] = (m < p.M && tid + k * NTHR < BM * 4) ? p.ref_in[(int64_t)m * 4 + (tid & 3)] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      *reinterpret_cast<u32x4*>(XA + row * 512 + ((c ^ (row & 15)) << 4)) = xr[k];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      *reinterpret_cast<u32x4*>(XB + row * 512 + ((c ^ (row & 15)) << 4)) = er[k];
    }
  }
  __syncthreads();
  stamp(0);

  f32x4 acc[MT][NT];
  // ---- P1: e2 = LN2(samp . Wp^T + bp + e1) -> XB
  {
    zero(acc);
    gemm_acc(XA, acc, WSrc{p.W1, 0, 256, 0});         // next: the first FFN chunk, in flight under the second K half and the LayerNorm
    stamp(1);
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int j = 0; j < NT; ++j) {                   // residual rows: from the e1 tile in XB, the very bytes e2 replaces below
      const int n = wave * WC + j * 16 + q * 4;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(V + TAIL_V_BP + n);
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = acc[i][j] + bb + get4(XB, i * 16 + r, n);
    }
    layer_norm(acc, V + TAIL_V_LN2G, V + TAIL_V_LN2B);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) put4(XB, i * 16 + r, wave * WC + j * 16 + q * 4, acc[i][j]);
  }
  stamp(12);
  __syncthreads();
  stamp(2);

  // ---- P2: FFN in chunks of 256 hidden units; acc3 accumulates linear2
  f32x4 acc3[MT][NT];
  zero(acc3);
  const int nchunk = p.d_ffn >> 8;
  auto wstamp = [&](int c, int i) {
    if constexpr (ABL >= 2) {
      if (c == 1) { __builtin_amdgcn_sched_barrier(0); wt[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    }
  };
  for (int c = 0; c < nchunk; ++c) {
    zero(acc);
    wstamp(c, 0);
    gemm_acc(XB, acc, WSrc{p.W2, 0, p.d_ffn, c * 256});
    wstamp(c, 1);   // weights: W1 rows [c*256 + wave*32, +32); next: linear2, this wave's 32 output rows, k slice of chunk c
    stamp(3);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * WC + j * 16 + q * 4;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(V + TAIL_V_B1 + c * 256 + n);
#pragma unroll
      for (int i = 0; i < MT; ++i)
        put4(XA, i * 16 + r, n, __builtin_elementwise_max(acc[i][j] + bb, f32x4{0.f, 0.f, 0.f, 0.f}));
    }
    stamp(13);
    wstamp(c, 2);
    __syncthreads();
    wstamp(c, 3);
    stamp(4);
    gemm_acc(XA, acc3, c + 1 < nchunk ? WSrc{p.W1, (c + 1) * 256, 256, 0}
                                      : (p.qkv ? WSrc{p.Wqkv, 512, 256, 0} : WSrc{p.B0, 0, 256, 0}));   // next: the following chunk; then v of the next layer, or the box head's first layer
    stamp(14);
    wstamp(c, 4);
    __syncthreads();                                 // XA is rewritten by the next chunk (or by e3 below)
    wstamp(c, 5);
    stamp(5);
  }
  // e3 = LN3(acc3 + b2 + e2) -> XA, and out
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = wave * WC + j * 16 + q * 4;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(V + TAIL_V_B2 + n);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc3[i][j] = acc3[i][j] + bb + get4(XB, i * 16 + r, n);
  }
  layer_norm(acc3, V + TAIL_V_LN3G, V + TAIL_V_LN3B);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) put4(XA, i * 16 + r, wave * WC + j * 16 + q * 4, acc3[i][j]);
  __syncthreads();
  stamp(6);
  {
    T* Og = static_cast<T*>(p.out);
    // round 3: optionally also out + query_pos (the q = k operand of the NEXT layer's self-attention, transformer.py:637-638),
    // formed as moy_gemm forms its A2 operand: the next layer's q | k projection is then a plain GEMM over it
    T* Xg = static_cast<T*>(p.out_xp);
    const T* Qg = static_cast<const T*>(p.qpos);
#pragma unroll
    for (int k = 0; k < BM * 32 / NTHR; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      if (m0 + row < p.M) {
        const u32x4 e = *reinterpret_cast<const u32x4*>(XA + row * 512 + ((c ^ (row & 15)) << 4));
        *reinterpret_cast<u32x4*>(Og + (int64_t)(m0 + row) * p.ld_out + c * 8) = e;
        if (Xg || p.qkv) {
          const u32x4 qv = *reinterpret_cast<const u32x4*>(Qg + (int64_t)(m0 + row) * p.ld_qpos + c * 8);
          u32x4 sx;
          sx.x = DT<T>::pack2(DT<T>::lo(e.x) + DT<T>::lo(qv.x), DT<T>::hi(e.x) + DT<T>::hi(qv.x));
          sx.y = DT<T>::pack2(DT<T>::lo(e.y) + DT<T>::lo(qv.y), DT<T>::hi(e.y) + DT<T>::hi(qv.y));
          sx.z = DT<T>::pack2(DT<T>::lo(e.z) + DT<T>::lo(qv.z), DT<T>::hi(e.z) + DT<T>::hi(qv.z));
          sx.w = DT<T>::pack2(DT<T>::lo(e.w) + DT<T>::lo(qv.w), DT<T>::hi(e.w) + DT<T>::hi(qv.w));
          if (Xg) *reinterpret_cast<u32x4*>(Xg + (int64_t)(m0 + row) * p.ld_xp + c * 8) = sx;
          if (p.qkv) *reinterpret_cast<u32x4*>(XB + row * 512 + ((c ^ (row & 15)) << 4)) = sx;     // XB: e2 is no longer needed
        }
      }
    }
  }
  // ---- round 5, optional: q | k | v of the NEXT layer's self-attention (transformer.py:637-640: q = k = (out + query_pos) Wq|k^T + b,
  //      v = out Wv^T + b) while the layer output is still on chip: three more products of the kind above instead of two launches that
  //      read out / out + query_pos back (0.075 ms per layer at 86 400 rows against ~0.03 ms here); the bits are those of moy_gemm
  if (p.qkv) {
    __syncthreads();                                   // out + query_pos complete in XB
    T* Qg = static_cast<T*>(p.qkv);
    auto project = [&](const unsigned char* As, int row0, const WSrc& next) {
      f32x4 bq[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bq[j] = *reinterpret_cast<const f32x4*>(p.bqkv + row0 + wave * WC + j * 16 + q * 4);   // before the product: nothing queued ahead
      zero(acc);
      gemm_acc(As, acc, next);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = row0 + wave * WC + j * 16 + q * 4;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int m = m0 + i * 16 + r;
          const f32x4 v = acc[i][j] + bq[j];
          if (m < p.M) *reinterpret_cast<u32x2*>(Qg + (int64_t)m * p.ld_qkv + n) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
        }
      }
    };
    project(XA, 512, WSrc{p.Wqkv, 0, 256, 0});        // v from out; next: q
    project(XB, 0, WSrc{p.Wqkv, 256, 256, 0});        // q from out + query_pos; next: k
    project(XB, 256, WSrc{p.B0, 0, 256, 0});          // k; next: the box head's first layer
    __syncthreads();                                   // XB is rewritten by the box head
  }

  // ---- P3: box head on e3 (XA): t1 -> XB, t2 in registers, 4 dots per row, refinement
  stamp(7);
  zero(acc);
  gemm_acc(XA, acc, WSrc{p.B1, 0, 256, 0});
  stamp(8);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = wave * WC + j * 16 + q * 4;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(V + TAIL_V_C0 + n);
#pragma unroll
    for (int i = 0; i < MT; ++i)
      put4(XB, i * 16 + r, n, __builtin_elementwise_max(acc[i][j] + bb, f32x4{0.f, 0.f, 0.f, 0.f}));
  }
  __syncthreads();
  stamp(9);
  f32x4 c1v[NT], w2v[NT][4];                       // the last epilogue's vectors: requested before the product (nothing is queued ahead of them)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = wave * WC + j * 16 + q * 4;
    c1v[j] = *reinterpret_cast<const f32x4*>(p.c1 + n);
#pragma unroll
    for (int o = 0; o < 4; ++o) w2v[j][o] = *reinterpret_cast<const f32x4*>(p.w2 + o * 256 + n);
  }
  zero(acc);
  gemm_acc(XB, acc, WSrc{nullptr, 0, 0, 0});
  stamp(10);
  float part[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int o = 0; o < 4; ++o) part[i][o] = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const f32x4 bb = c1v[j];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x4 v = __builtin_elementwise_max(acc[i][j] + bb, f32x4{0.f, 0.f, 0.f, 0.f});
      const uint32_t lo = DT<T>::pack2(v.x, v.y), hi = DT<T>::pack2(v.z, v.w);
      v = f32x4{DT<T>::lo(lo), DT<T>::hi(lo), DT<T>::lo(hi), DT<T>::hi(hi)};
#pragma unroll
      for (int o = 0; o < 4; ++o) part[i][o] += (v.x * w2v[j][o].x + v.y * w2v[j][o].y) + (v.z * w2v[j][o].z + v.w * w2v[j][o].w);
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float v = part[i][o];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      part[i][o] = v;
    }
  if (q == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      *reinterpret_cast<f32x4*>(P + ((i * 16 + r) * NW + wave) * 4) = f32x4{part[i][0], part[i][1], part[i][2], part[i][3]};
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NREF; ++k) {
    const int row = (tid + k * NTHR) >> 2, o = tid & 3, m = m0 + row;
    if (m < p.M && row < BM) {
      const float* pr = P + row * NW * 4 + o;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += pr[w * 4];
      v += p.c2[o];
      p.ref_out[(int64_t)m * 4 + o] = sigmoidf_(v + tail_inv_sigmoid(ref_in_r[k]));
    }
  }
  stamp(11);
  if constexpr (ABL >= 2) {
    if ((blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && tid == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.out) + (blockIdx.x ? 16 : 0);
      for (int i = 0; i < 16; ++i) dbg[i] = ph[i];
    }
    if (blockIdx.x == 0 && lane == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.out) + 64 + wave * 8;
      for (int i = 0; i < 6; ++i) dbg[i] = wt[i];
    }
  }
}

}  // namespace moy

using namespace moy;

extern "C" int moy_decoder_tail(const moy_decoder_tail_args* a, void* stream) {
  if (!a || !a->samp || !a->e1 || !a->Wp || !a->bp || !a->ln2_g || !a->ln2_b || !a->W1 || !a->b1 || !a->W2 || !a->b2 || !a->ln3_g ||
      !a->ln3_b || !a->out || !a->B0 || !a->c0 || !a->B1 || !a->c1 || !a->w2 || !a->c2 || !a->ref_in || !a->ref_out)
    return MOY_EINVAL;
  if (a->M <= 0 || a->d_ffn <= 0 || (a->d_ffn % 256)) return MOY_EINVAL;
  if (a->d_ffn > TAIL_MAX_FFN) return MOY_ENOSYS;                       // linear1's bias is staged in LDS whole (512 threads x 4 floats)
  if (a->dtype != MOY_BF16 && a->dtype != MOY_F16) return MOY_ENOSYS;   // fp32: the separate launches (the parity path)
  if ((a->ld_samp % 8) || (a->ld_e1 % 8) || (a->ld_out % 8) || a->ld_samp < 256 || a->ld_e1 < 256 || a->ld_out < 256) return MOY_EINVAL;
  if (a->qkv && (!a->Wqkv || !a->bqkv || !a->qpos || (a->ld_qkv % 4) || a->ld_qkv < 768 || (a->ld_qpos % 8) || a->ld_qpos < 256 || !aligned16(a->Wqkv) ||
                 !aligned16(a->bqkv) || !aligned16(a->qpos) || (reinterpret_cast<uintptr_t>(a->qkv) & 7)))
    return MOY_EINVAL;
  if (a->out_xp && (!a->qpos || (a->ld_xp % 8) || (a->ld_qpos % 8) || a->ld_xp < 256 || a->ld_qpos < 256 || !aligned16(a->out_xp) || !aligned16(a->qpos)))
    return MOY_EINVAL;
  if (!aligned16(a->samp) || !aligned16(a->out) || !aligned16(a->Wp) || !aligned16(a->W1) || !aligned16(a->W2) || !aligned16(a->B0) ||
      !aligned16(a->B1) || !aligned16(a->w2) || !aligned16(a->bp) || !aligned16(a->b1) || !aligned16(a->b2) || !aligned16(a->c0) ||
      !aligned16(a->c1) || !aligned16(a->ln2_g) || !aligned16(a->ln2_b) || !aligned16(a->ln3_g) || !aligned16(a->ln3_b) ||
      !aligned16(a->e1))
    return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // small M: lower blocks, so that the launch is more than a handful of long chains (see BM_ above): 32 rows below 64 x 192 rows, 64 rows
  // below 128 x 192 rows (at least ~192 blocks before the block grows)
  if (a->M < 128 * 192) {
    const bool b32 = a->M < 64 * 192;
    const int bm = b32 ? 32 : 64, lds = tail_lds(bm);
    static bool attr_small = false;
    if (!attr_small) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_tail_kernel<bf16_t, 0, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, tail_lds(32)) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_tail_kernel<f16_t, 0, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, tail_lds(32)) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_tail_kernel<bf16_t, 0, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, tail_lds(64)) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_tail_kernel<f16_t, 0, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, tail_lds(64)) != hipSuccess)
        return MOY_ELAUNCH;
      attr_small = true;
    }
    const int nb = (a->M + bm - 1) / bm;
    if (a->dtype == MOY_BF16) {
      if (b32) hipLaunchKernelGGL((decoder_tail_kernel<bf16_t, 0, 32>), dim3(nb), dim3(64 * TAIL_NW), lds, st, *a);
      else hipLaunchKernelGGL((decoder_tail_kernel<bf16_t, 0, 64>), dim3(nb), dim3(64 * TAIL_NW), lds, st, *a);
    } else {
      if (b32) hipLaunchKernelGGL((decoder_tail_kernel<f16_t, 0, 32>), dim3(nb), dim3(64 * TAIL_NW), lds, st, *a);
      else hipLaunchKernelGGL((decoder_tail_kernel<f16_t, 0, 64>), dim3(nb), dim3(64 * TAIL_NW), lds, st, *a);
    }
    return launch_status();
  }
  static bool attr_set = false;          // > 64 KiB of dynamic LDS: opt in once per kernel symbol
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_tail_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_tail_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  const int blocks = (a->M + TAIL_BM - 1) / TAIL_BM;
#if MOY_DIAG
  static const int abl = garbage_mode_env("MOY_TAIL_ABL");
  if (abl == 3 && a->dtype == MOY_BF16) {          // stamps + no weight traffic after the first product
    auto k3 = decoder_tail_kernel<bf16_t, 3>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) != hipSuccess) return MOY_ELAUNCH;
    hipLaunchKernelGGL(k3, dim3(blocks), dim3(64 * TAIL_NW), TAIL_LDS, st, *a);
    return launch_status();
  }
  if (abl == 2 && a->dtype == MOY_BF16) {
    auto k2 = decoder_tail_kernel<bf16_t, 2>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) != hipSuccess) return MOY_ELAUNCH;
    hipLaunchKernelGGL(k2, dim3(blocks), dim3(64 * TAIL_NW), TAIL_LDS, st, *a);
    return launch_status();
  }
  if (abl == 1 && a->dtype == MOY_BF16) {
    auto k1 = decoder_tail_kernel<bf16_t, 1>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) != hipSuccess) return MOY_ELAUNCH;
    hipLaunchKernelGGL(k1, dim3(blocks), dim3(64 * TAIL_NW), TAIL_LDS, st, *a);
    return launch_status();
  }
#endif
  if (a->dtype == MOY_BF16)
    hipLaunchKernelGGL((decoder_tail_kernel<bf16_t>), dim3(blocks), dim3(64 * TAIL_NW), TAIL_LDS, st, *a);
  else
    hipLaunchKernelGGL((decoder_tail_kernel<f16_t>), dim3(blocks), dim3(64 * TAIL_NW), TAIL_LDS, st, *a);
  return launch_status();
}
