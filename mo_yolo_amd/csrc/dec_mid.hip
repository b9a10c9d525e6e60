// The row-wise MIDDLE of a decoder layer in ONE launch (MOTRDecoderLayer.forward between the self-attention core and the
// deformable sampling, nn/modules/transformer.py:640-646, and MSDeformAttn.forward's two query-side linears, :262-266):
//   e1    = LayerNorm1(x + attn . Wo^T + bo)                     self_attn.out_proj + dropout(identity) + norm1
//   offaw = (e1 + query_pos) . [W_off | W_aw]^T + [b_off | b_aw]  sampling_offsets | attention_weights, fp32 out
// As two launches (out_proj + LayerNorm 66 us, offsets + weights 60 us per layer at 288 frames x 300 rows, both on the tiled GEMM:
// K = 256 is four k steps, so each is mostly pipeline fill, epilogue and drain) this was 0.75 ms of a 22.3 ms pass.  Structure =
// csrc/dec_tail.hip (a block owns 128 rows for the whole chain, activation tiles in LDS with XOR-swizzled 512-byte rows, each of the 8
// waves holds its 32 output columns of the current weight matrix in registers, K halves re-requested as soon as consumed; round 5: weights in
// MFMA-fragment order when the caller says so, the residual tile / fp32 vectors / query_pos tile requested up front -- no epilogue reads global memory):
//   * P1: attn tile . Wo^T, + bias + residual rows of x, one-pass LayerNorm (lane partials, xor-shuffles, per-wave partials in LDS)
//         -> e1, rounded to the storage type: to LDS (the next product's operand) and, in whole 512-byte rows, to global memory
//         (decoder_tail's residual);
//   * P2: A' = e1 + query_pos formed in LDS exactly as moy_gemm's A2 operand is (element pairs added in fp32, one rounding);
//         the first 256 output columns as a full-width product (8 waves x 32 columns), the remaining n_oa - 256 columns (32 for three
//         levels) ROW-split: every wave takes one 16-row tile against the same 32-column weight group -- a full-width second product
//         would spend 7/8 of its MFMAs on padding;
//   * offsets / weights leave as fp32, 16 bytes per lane (4 consecutive columns of one row).
#include "common.hpp"

namespace moy {

template <typename T>
__device__ __forceinline__ f32x4 mid_mfma(f32x4 acc, u32x4 w, u32x4 a);
template <>
__device__ __forceinline__ f32x4 mid_mfma<bf16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mid_mfma<f16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}

constexpr int MID_BM = 128, MID_NW = 8;
// fp32 vectors staged in LDS once per block (round 5, as dec_tail.hip): [bo | ln_g | ln_b | boa[0 .. 512)]
constexpr int MID_V_BO = 0, MID_V_LNG = 256, MID_V_LNB = 512, MID_V_BOA = 768, MID_V_N = 768 + 512;
constexpr int MID_LDS = 2 * MID_BM * 512 + MID_BM * MID_NW * 8 + MID_V_N * 4;

template <typename T>
__global__ __launch_bounds__(64 * MID_NW) void decoder_mid_kernel(const moy_decoder_mid_args p) {
  constexpr int BM = MID_BM, NW = MID_NW, NTHR = 64 * NW, MT = BM / 16, NT = 2, WC = 32;
  static_assert(MT == NW, "the row-split remainder product gives every wave one 16-row tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* XA = smem;
  unsigned char* XB = smem + BM * 512;
  float* P = reinterpret_cast<float*>(smem + 2 * BM * 512);      // [BM][NW][2] floats: LayerNorm partials
  const float* V = reinterpret_cast<const float*>(smem + 2 * BM * 512 + BM * NW * 8);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * BM;

  // this wave's 32 weight rows (output columns) of a [*, 256] matrix as two K halves of four 32-wide panels (see dec_tail.hip)
  u32x4 wa[2][NT][4];
  auto req_half = [&](int h, const void* W, int row0) {
    if (!W) return;
    if (p.w_packed) {        // MFMA-fragment order (include/moyolo.h): the 32-row group row0 / 32, panels 4h .. 4h+3: 8 KB contiguous
      const unsigned char* Wg = static_cast<const unsigned char*>(W) + ((int64_t)(row0 >> 5) * 8 + h * 4) * 2048 + lane * 16;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int pn = 0; pn < 4; ++pn) wa[h][j][pn] = *reinterpret_cast<const u32x4*>(Wg + pn * 2048 + j * 1024);
      return;
    }
    const T* Wg = static_cast<const T*>(W) + (int64_t)row0 * 256 + h * 128;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int pn = 0; pn < 4; ++pn) wa[h][j][pn] = *reinterpret_cast<const u32x4*>(Wg + (int64_t)(j * 16 + r) * 256 + pn * 32 + q * 8);
  };
  const int lbase = r * 512 + ((q ^ r) << 4);      // fragment of row i*16 + r, chunk (pn*4 + q) ^ r
  // full-width product: acc[i][j] (row tile i, this wave's column tile j) += As . wa^T; K halves outermost, `nextW` re-requested per half
  auto gemm_acc = [&](const unsigned char* As, f32x4 (&acc)[MT][NT], const void* nextW, int next_row0) {
    constexpr int RS = 4, HM = MT / RS, NSTEP = 8 * RS;
    u32x4 af[2][HM];
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
        for (int j = 0; j < NT; ++j) acc[rp * HM + i][j] = mid_mfma<T>(acc[rp * HM + i][j], wa[kh][j][p4], af[s_ & 1][i]);
      __builtin_amdgcn_sched_barrier(0);
      if (s_ == NSTEP / 2 - 1) { req_half(0, nextW, next_row0); __builtin_amdgcn_sched_barrier(0); }
      if (s_ == NSTEP - 1) { req_half(1, nextW, next_row0); __builtin_amdgcn_sched_barrier(0); }
    }
  };
  auto tile_ptr = [&](unsigned char* X, int row, int n) { return X + row * 512 + (((n >> 3) ^ (row & 15)) << 4) + (n & 7) * 2; };
  auto put4 = [&](unsigned char* X, int row, int n, f32x4 v) {
    *reinterpret_cast<u32x2*>(tile_ptr(X, row, n)) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
  };
  auto get4 = [&](unsigned char* X, int row, int n) {
    const u32x2 w = *reinterpret_cast<const u32x2*>(tile_ptr(X, row, n));
    return f32x4{DT<T>::lo(w.x), DT<T>::hi(w.x), DT<T>::lo(w.y), DT<T>::hi(w.y)};
  };

  // ---- P0: attention output tile -> XA, the residual tile x -> XB (e1 replaces it in place), the fp32 vectors -> V, out_proj weights ->
  //      registers; the query_pos tile is requested here too and waits in registers until the LayerNorm is through.  Round 5, as in
  //      dec_tail.hip: no epilogue reads global memory -- vector-memory results return in issue order, so a bias / residual / LayerNorm /
  //      query_pos load issued after a product came back only behind the weights of the NEXT product requested during it.
  constexpr int NLD = BM * 32 / NTHR;
  u32x4 qr[NLD];
  {
    const T* Xg = static_cast<const T*>(p.attn);
    const T* Rg = static_cast<const T*>(p.x);
    const T* Qg = static_cast<const T*>(p.qpos);
    u32x4 xr[NLD], rr[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      const int m = min(m0 + row, p.M - 1);
      xr[k] = *reinterpret_cast<const u32x4*>(Xg + (int64_t)m * p.ld_attn + c * 8);
    }
    req_half(0, p.Wo, wave * WC);
    req_half(1, p.Wo, wave * WC);
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      const int m = min(m0 + row, p.M - 1);
      rr[k] = *reinterpret_cast<const u32x4*>(Rg + (int64_t)m * p.ld_x + c * 8);
    }
    {
      float* Vw = const_cast<float*>(V);
      const float* src = wave == 0 ? p.bo : wave == 1 ? p.ln_g : p.ln_b;
      if (wave < 3) *reinterpret_cast<f32x4*>(Vw + wave * 256 + lane * 4) = *reinterpret_cast<const f32x4*>(src + lane * 4);
      if (tid * 4 < p.n_oa) *reinterpret_cast<f32x4*>(Vw + MID_V_BOA + tid * 4) = *reinterpret_cast<const f32x4*>(p.boa + tid * 4);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      const int m = min(m0 + row, p.M - 1);
      qr[k] = *reinterpret_cast<const u32x4*>(Qg + (int64_t)m * p.ld_qpos + c * 8);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      *reinterpret_cast<u32x4*>(XA + row * 512 + ((c ^ (row & 15)) << 4)) = xr[k];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      *reinterpret_cast<u32x4*>(XB + row * 512 + ((c ^ (row & 15)) << 4)) = rr[k];
    }
  }
  __syncthreads();

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- P1: e1 = LN1(attn . Wo^T + bo + x) -> XB and global
  gemm_acc(XA, acc, p.Woa, wave * WC);               // next: offsets | weights, columns [wave*32, +32) of the first 256
  {
#pragma unroll
    for (int j = 0; j < NT; ++j) {                   // residual rows: from the x tile in XB, the very bytes e1 replaces below
      const int n = wave * WC + j * 16 + q * 4;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(V + MID_V_BO + n);
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = acc[i][j] + bb + get4(XB, i * 16 + r, n);
    }
    // one-pass LayerNorm over the 256 columns of every row (as dec_tail.hip)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x4 x = acc[i][j];
        s1 += (x.x + x.y) + (x.z + x.w);
        s2 += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
      }
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (q == 0) *reinterpret_cast<float2*>(P + ((i * 16 + r) * NW + wave) * 2) = float2{s1, s2};
    }
    __syncthreads();
    f32x4 gg[NT], bb2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * WC + j * 16 + q * 4;
      gg[j] = *reinterpret_cast<const f32x4*>(V + MID_V_LNG + n);
      bb2[j] = *reinterpret_cast<const f32x4*>(V + MID_V_LNB + n);
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
      for (int j = 0; j < NT; ++j) put4(XB, i * 16 + r, wave * WC + j * 16 + q * 4, (acc[i][j] - mean) * rstd * gg[j] + bb2[j]);
    }
  }
  __syncthreads();
  // e1 rows leave in whole 512-byte rows; A' = e1 + query_pos -> XA (the attention tile is no longer needed)
  {
    T* Eg = static_cast<T*>(p.e1);
#pragma unroll
    for (int k = 0; k < BM * 32 / NTHR; ++k) {
      const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
      const u32x4 e = *reinterpret_cast<const u32x4*>(XB + row * 512 + ((c ^ (row & 15)) << 4));
      if (m0 + row < p.M) *reinterpret_cast<u32x4*>(Eg + (int64_t)(m0 + row) * p.ld_e1 + c * 8) = e;
      u32x4 s;
      s.x = DT<T>::pack2(DT<T>::lo(e.x) + DT<T>::lo(qr[k].x), DT<T>::hi(e.x) + DT<T>::hi(qr[k].x));
      s.y = DT<T>::pack2(DT<T>::lo(e.y) + DT<T>::lo(qr[k].y), DT<T>::hi(e.y) + DT<T>::hi(qr[k].y));
      s.z = DT<T>::pack2(DT<T>::lo(e.z) + DT<T>::lo(qr[k].z), DT<T>::hi(e.z) + DT<T>::hi(qr[k].z));
      s.w = DT<T>::pack2(DT<T>::lo(e.w) + DT<T>::lo(qr[k].w), DT<T>::hi(e.w) + DT<T>::hi(qr[k].w));
      *reinterpret_cast<u32x4*>(XA + row * 512 + ((c ^ (row & 15)) << 4)) = s;
    }
  }
  __syncthreads();

  // ---- P2a: the first min(n_oa, 256) offset / weight columns: full-width product
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n_rem = p.n_oa > 256 ? p.n_oa - 256 : 0;            // a multiple of 32 (checked by the host)
  gemm_acc(XA, acc, n_rem ? p.Woa : nullptr, 256);              // next: the first 32-column group of the remainder (same rows in every wave)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = wave * WC + j * 16 + q * 4;
    if (n < p.n_oa) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(V + MID_V_BOA + n);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = m0 + i * 16 + r;
        if (m < p.M) *reinterpret_cast<f32x4*>(p.offaw + (int64_t)m * p.ld_oa + n) = acc[i][j] + bb;
      }
    }
  }
  // ---- P2b: remaining columns in groups of 32, row-split: this wave's 16-row tile against the group's weights
  for (int g = 0; g < n_rem; g += 32) {
    f32x4 a2[NT] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    u32x4 af[8];
#pragma unroll
    for (int pn = 0; pn < 8; ++pn) af[pn] = *reinterpret_cast<const u32x4*>(XA + ((lbase ^ (pn * 64)) + wave * 8192));
#pragma unroll
    for (int pn = 0; pn < 8; ++pn)
#pragma unroll
      for (int j = 0; j < NT; ++j) a2[j] = mid_mfma<T>(a2[j], wa[pn >> 2][j][pn & 3], af[pn]);
    if (g + 32 < n_rem) { req_half(0, p.Woa, 256 + g + 32); req_half(1, p.Woa, 256 + g + 32); }
    const int m = m0 + wave * 16 + r;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = 256 + g + j * 16 + q * 4;
      if (m < p.M) *reinterpret_cast<f32x4*>(p.offaw + (int64_t)m * p.ld_oa + n) = a2[j] + *reinterpret_cast<const f32x4*>(V + MID_V_BOA + n);
    }
  }
}

}  // namespace moy

using namespace moy;

extern "C" int moy_decoder_mid(const moy_decoder_mid_args* a, void* stream) {
  if (!a || !a->attn || !a->x || !a->qpos || !a->Wo || !a->bo || !a->ln_g || !a->ln_b || !a->Woa || !a->boa || !a->e1 || !a->offaw)
    return MOY_EINVAL;
  if (a->M <= 0 || a->n_oa <= 0 || (a->n_oa % 32) || a->n_oa > 512) return MOY_EINVAL;
  if (a->dtype != MOY_BF16 && a->dtype != MOY_F16) return MOY_ENOSYS;   // fp32: the separate launches (the parity path)
  if ((a->ld_attn % 8) || (a->ld_x % 8) || (a->ld_qpos % 8) || (a->ld_e1 % 8) || (a->ld_oa % 4) || a->ld_attn < 256 || a->ld_x < 256 ||
      a->ld_qpos < 256 || a->ld_e1 < 256 || a->ld_oa < a->n_oa)
    return MOY_EINVAL;
  if (!aligned16(a->attn) || !aligned16(a->qpos) || !aligned16(a->e1) || !aligned16(a->Wo) || !aligned16(a->Woa) || !aligned16(a->bo) ||
      !aligned16(a->boa) || !aligned16(a->ln_g) || !aligned16(a->ln_b) || !aligned16(a->offaw) || !aligned16(a->x))
    return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static bool attr_set = false;          // > 64 KiB of dynamic LDS: opt in once per kernel symbol
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_mid_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_mid_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, MID_LDS) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  const int blocks = (a->M + MID_BM - 1) / MID_BM;
  if (a->dtype == MOY_BF16)
    hipLaunchKernelGGL((decoder_mid_kernel<bf16_t>), dim3(blocks), dim3(64 * MID_NW), MID_LDS, st, *a);
  else
    hipLaunchKernelGGL((decoder_mid_kernel<f16_t>), dim3(blocks), dim3(64 * MID_NW), MID_LDS, st, *a);
  return launch_status();
}
