// The row-wise tail of a decoder layer in ONE launch (MOTRDecoderLayer.forward after the deformable sampling,
// nn/modules/transformer.py:642-652, and the box refinement of the decoder loop, :705-709):
//   e2  = LayerNorm2(samp . Wp^T + bp + e1)                      cross_attn.output_proj + dropout(identity) + norm2
//   e3  = LayerNorm3(relu(e2 . W1^T + b1) . W2^T + b2 + e2)      linear1 / relu / linear2 + norm3   -> layer output
//   box = sigmoid(MLP3(e3) + inverse_sigmoid(ref))               dec_bbox_head[i], refined reference boxes
// As separate launches this was output_proj+LN (27 us), linear1 (35), linear2+LN (35) and the box head (22) per layer at
// 96 frames x 300 rows: latency chains over a problem that does not fill the chip, with e2, the 1024-wide hidden activation and
// e3 making round trips through memory.  Structure = csrc/mlp_head.hip, extended:
//   * a block owns 128 rows for the whole chain; the current activation tile lives in LDS (two 64 KB tiles, XOR-swizzled 512-byte
//     rows); each of the 8 waves keeps its 32 output columns of the current weight chunk in registers (MFMA A operand);
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

constexpr int TAIL_BM = 128, TAIL_NW = 8;
constexpr int TAIL_LDS = 2 * TAIL_BM * 512 + TAIL_BM * TAIL_NW * 16;

template <typename T, int ABL = 0>       // ABL 1: timing-only build that keeps the FIRST weight chunk for every product (MOY_TAIL_ABL=1; results garbage)
__global__ __launch_bounds__(64 * TAIL_NW) void decoder_tail_kernel(const moy_decoder_tail_args p) {
  constexpr int BM = TAIL_BM, NW = TAIL_NW, NTHR = 64 * NW, MT = BM / 16, NT = 2, WC = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* XA = smem;
  unsigned char* XB = smem + BM * 512;
  float* P = reinterpret_cast<float*>(smem + 2 * BM * 512);      // [BM][NW][4] floats: LayerNorm / box-head partials
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * BM;

  // This wave's 32 output rows of the current [*, pitch] weight matrix (k columns koff .. koff+255) as two K HALVES of four
  // 32-wide panels: wa[0] = panels 0-3, wa[1] = panels 4-7.  Round 3: a product walks its K halves OUTERMOST (both row halves of
  // half 0, then both of half 1; every accumulator still sees its panels in the order 0..7, so results are unchanged), and the
  // moment a half has been consumed the SAME registers are re-requested with that half of the NEXT product's weights -- the load
  // is in flight under the remaining MFMAs of this product, its epilogue and the barrier.  Round 2 fetched a whole chunk after
  // the product that used the previous one (two chunks live = spills) and measured 36 % of the kernel waiting for those loads.
  u32x4 wa[2][NT][4];
  struct WSrc { const void* W; int row0, pitch, koff; };
  bool w_loaded = false;
  auto req_half = [&](int h, const WSrc& w) {
    if constexpr (ABL == 1) { if (w_loaded) return; }
    if (!w.W) return;
    const T* Wg = static_cast<const T*>(w.W) + (int64_t)(w.row0 + wave * WC) * w.pitch + w.koff + h * 128;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int pn = 0; pn < 4; ++pn) wa[h][j][pn] = *reinterpret_cast<const u32x4*>(Wg + (int64_t)(j * 16 + r) * w.pitch + pn * 32 + q * 8);
  };
  const int lbase = r * 512 + ((q ^ r) << 4);      // fragment of row i*16 + r, chunk (pn*4 + q) ^ r
  // rows in two halves of 64: 16 fragment registers live instead of 32
  auto gemm_acc = [&](const unsigned char* As, f32x4 (&acc)[MT][NT], const WSrc& next) {
    constexpr int RS = 4, HM = MT / RS, NSTEP = 8 * RS;   // step s = (K half kh, row part rp, panel kh * 4 + s % 4), K half outermost;
    u32x4 af[2][HM];                                      // the fragments of step s+1 are requested before the MFMAs of step s
    auto frag = [&](int s_, u32x4 (&buf)[HM]) {
      const int kh = s_ / (4 * RS), rp = (s_ >> 2) % RS, pn = kh * 4 + (s_ & 3);
#pragma unroll
      for (int i = 0; i < HM; ++i) buf[i] = *reinterpret_cast<const u32x4*>(As + ((lbase ^ (pn * 64)) + (rp * HM + i) * 8192));
    };
    frag(0, af[0]);
#pragma unroll
    for (int s_ = 0; s_ < NSTEP; ++s_) {
      const int kh = s_ / (4 * RS), rp = (s_ >> 2) % RS, p4 = s_ & 3;
      if (s_ + 1 < NSTEP) frag(s_ + 1, af[(s_ + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < HM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[rp * HM + i][j] = tail_mfma<T>(acc[rp * HM + i][j], wa[kh][j][p4], af[s_ & 1][i]);
      __builtin_amdgcn_sched_barrier(0);
      if (s_ == NSTEP / 2 - 1) { req_half(0, next); __builtin_amdgcn_sched_barrier(0); }   // K half 0 consumed: its registers take the next product's
      if (s_ == NSTEP - 1) { req_half(1, next); __builtin_amdgcn_sched_barrier(0); if constexpr (ABL == 1) w_loaded = true; }
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
      gg[j] = *reinterpret_cast<const f32x4*>(g + n);
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

  // ---- P0: sampling output tile -> XA, output_proj weights -> registers
  {
    const T* Xg = static_cast<const T*>(p.samp);
    u32x4 xr[BM * 32 / NTHR];
#pragma unroll
    for (int k = 0; k < BM * 32 / NTHR; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      const int m = min(m0 + row, p.M - 1);
      xr[k] = *reinterpret_cast<const u32x4*>(Xg + (int64_t)m * p.ld_samp + c * 8);
    }
    { const WSrc w0{p.Wp, 0, 256, 0}; req_half(0, w0); req_half(1, w0); }
#pragma unroll
    for (int k = 0; k < BM * 32 / NTHR; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      *reinterpret_cast<u32x4*>(XA + row * 512 + ((c ^ (row & 15)) << 4)) = xr[k];
    }
  }
  __syncthreads();

  f32x4 acc[MT][NT];
  // ---- P1: e2 = LN2(samp . Wp^T + bp + e1) -> XB
  {
    zero(acc);
    gemm_acc(XA, acc, WSrc{p.W1, 0, 256, 0});         // next: the first FFN chunk, in flight under the second K half and the LayerNorm
    __builtin_amdgcn_sched_barrier(0);
    const T* Eg = static_cast<const T*>(p.e1);       // residual rows: 8 bytes per lane and sub-tile
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * WC + j * 16 + q * 4;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bp + n);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = min(m0 + i * 16 + r, p.M - 1);
        const u32x2 rs = *reinterpret_cast<const u32x2*>(Eg + (int64_t)m * p.ld_e1 + n);
        acc[i][j] = acc[i][j] + bb + f32x4{DT<T>::lo(rs.x), DT<T>::hi(rs.x), DT<T>::lo(rs.y), DT<T>::hi(rs.y)};
      }
    }
    layer_norm(acc, p.ln2_g, p.ln2_b);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) put4(XB, i * 16 + r, wave * WC + j * 16 + q * 4, acc[i][j]);
  }
  __syncthreads();

  // ---- P2: FFN in chunks of 256 hidden units; acc3 accumulates linear2
  f32x4 acc3[MT][NT];
  zero(acc3);
  const int nchunk = p.d_ffn >> 8;
  for (int c = 0; c < nchunk; ++c) {
    zero(acc);
    gemm_acc(XB, acc, WSrc{p.W2, 0, p.d_ffn, c * 256});   // weights: W1 rows [c*256 + wave*32, +32); next: linear2, this wave's 32 output rows, k slice of chunk c
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * WC + j * 16 + q * 4;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b1 + c * 256 + n);
#pragma unroll
      for (int i = 0; i < MT; ++i)
        put4(XA, i * 16 + r, n, __builtin_elementwise_max(acc[i][j] + bb, f32x4{0.f, 0.f, 0.f, 0.f}));
    }
    __syncthreads();
    gemm_acc(XA, acc3, c + 1 < nchunk ? WSrc{p.W1, (c + 1) * 256, 256, 0} : WSrc{p.B0, 0, 256, 0});   // next: the following chunk, or the box head's first layer
    __syncthreads();                                 // XA is rewritten by the next chunk (or by e3 below)
  }
  // e3 = LN3(acc3 + b2 + e2) -> XA, and out
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = wave * WC + j * 16 + q * 4;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b2 + n);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc3[i][j] = acc3[i][j] + bb + get4(XB, i * 16 + r, n);
  }
  layer_norm(acc3, p.ln3_g, p.ln3_b);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) put4(XA, i * 16 + r, wave * WC + j * 16 + q * 4, acc3[i][j]);
  __syncthreads();
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
        if (Xg) {
          const u32x4 qv = *reinterpret_cast<const u32x4*>(Qg + (int64_t)(m0 + row) * p.ld_qpos + c * 8);
          u32x4 sx;
          sx.x = DT<T>::pack2(DT<T>::lo(e.x) + DT<T>::lo(qv.x), DT<T>::hi(e.x) + DT<T>::hi(qv.x));
          sx.y = DT<T>::pack2(DT<T>::lo(e.y) + DT<T>::lo(qv.y), DT<T>::hi(e.y) + DT<T>::hi(qv.y));
          sx.z = DT<T>::pack2(DT<T>::lo(e.z) + DT<T>::lo(qv.z), DT<T>::hi(e.z) + DT<T>::hi(qv.z));
          sx.w = DT<T>::pack2(DT<T>::lo(e.w) + DT<T>::lo(qv.w), DT<T>::hi(e.w) + DT<T>::hi(qv.w));
          *reinterpret_cast<u32x4*>(Xg + (int64_t)(m0 + row) * p.ld_xp + c * 8) = sx;
        }
      }
    }
  }

  // ---- P3: box head on e3 (XA): t1 -> XB, t2 in registers, 4 dots per row, refinement
  zero(acc);
  gemm_acc(XA, acc, WSrc{p.B1, 0, 256, 0});
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = wave * WC + j * 16 + q * 4;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.c0 + n);
#pragma unroll
    for (int i = 0; i < MT; ++i)
      put4(XB, i * 16 + r, n, __builtin_elementwise_max(acc[i][j] + bb, f32x4{0.f, 0.f, 0.f, 0.f}));
  }
  __syncthreads();
  zero(acc);
  gemm_acc(XB, acc, WSrc{nullptr, 0, 0, 0});
  float part[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int o = 0; o < 4; ++o) part[i][o] = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = wave * WC + j * 16 + q * 4;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.c1 + n);
    f32x4 w2v[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) w2v[o] = *reinterpret_cast<const f32x4*>(p.w2 + o * 256 + n);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x4 v = __builtin_elementwise_max(acc[i][j] + bb, f32x4{0.f, 0.f, 0.f, 0.f});
      const uint32_t lo = DT<T>::pack2(v.x, v.y), hi = DT<T>::pack2(v.z, v.w);
      v = f32x4{DT<T>::lo(lo), DT<T>::hi(lo), DT<T>::lo(hi), DT<T>::hi(hi)};
#pragma unroll
      for (int o = 0; o < 4; ++o) part[i][o] += (v.x * w2v[o].x + v.y * w2v[o].y) + (v.z * w2v[o].z + v.w * w2v[o].w);
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
  static_assert(BM * 4 == NTHR, "one (row, output) per thread");
  {
    const int row = tid >> 2, o = tid & 3, m = m0 + row;
    if (m < p.M) {
      const float* pr = P + row * NW * 4 + o;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += pr[w * 4];
      v += p.c2[o];
      p.ref_out[(int64_t)m * 4 + o] = sigmoidf_(v + tail_inv_sigmoid(p.ref_in[(int64_t)m * 4 + o]));
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
  if (a->dtype != MOY_BF16 && a->dtype != MOY_F16) return MOY_ENOSYS;   // fp32: the separate launches (the parity path)
  if ((a->ld_samp % 8) || (a->ld_e1 % 4) || (a->ld_out % 8) || a->ld_samp < 256 || a->ld_e1 < 256 || a->ld_out < 256) return MOY_EINVAL;
  if (a->out_xp && (!a->qpos || (a->ld_xp % 8) || (a->ld_qpos % 8) || a->ld_xp < 256 || a->ld_qpos < 256 || !aligned16(a->out_xp) || !aligned16(a->qpos)))
    return MOY_EINVAL;
  if (!aligned16(a->samp) || !aligned16(a->out) || !aligned16(a->Wp) || !aligned16(a->W1) || !aligned16(a->W2) || !aligned16(a->B0) ||
      !aligned16(a->B1) || !aligned16(a->w2) || !aligned16(a->bp) || !aligned16(a->b1) || !aligned16(a->b2) || !aligned16(a->c0) ||
      !aligned16(a->c1) || !aligned16(a->ln2_g) || !aligned16(a->ln2_b) || !aligned16(a->ln3_g) || !aligned16(a->ln3_b) ||
      (reinterpret_cast<uintptr_t>(a->e1) & 7))
    return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static bool attr_set = false;          // > 64 KiB of dynamic LDS: opt in once per kernel symbol
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_tail_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_tail_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  const int blocks = (a->M + TAIL_BM - 1) / TAIL_BM;
  static int abl = -1;
  if (abl < 0) abl = garbage_mode_env("MOY_TAIL_ABL");
  if (abl == 1 && a->dtype == MOY_BF16) {
    auto k1 = decoder_tail_kernel<bf16_t, 1>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) != hipSuccess) return MOY_ELAUNCH;
    hipLaunchKernelGGL(k1, dim3(blocks), dim3(64 * TAIL_NW), TAIL_LDS, st, *a);
    return launch_status();
  }
  if (a->dtype == MOY_BF16)
    hipLaunchKernelGGL((decoder_tail_kernel<bf16_t>), dim3(blocks), dim3(64 * TAIL_NW), TAIL_LDS, st, *a);
  else
    hipLaunchKernelGGL((decoder_tail_kernel<f16_t>), dim3(blocks), dim3(64 * TAIL_NW), TAIL_LDS, st, *a);
  return launch_status();
}
