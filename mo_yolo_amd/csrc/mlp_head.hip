// Three-layer box head in ONE launch: y = W2 . relu(W1 . relu(W0 . x + b0) + b1) + b2 (+ the refine / anchor step), the MLP of
// enc_bbox_head / dec_bbox_head[i] (nn/modules/transformer.py:149-161 with num_layers = 3; head.py:1045, transformer.py:709),
// hidden width 256, 4 outputs, 16-bit types.
//
// As three launches (two 256x256 GEMMs of M = frames*300 rows + a one-wave-per-row dot) the chain cost 15 + 15 + 24 us per decoder
// layer: each launch is a latency chain (first loads -> 4 k-stages -> epilogue -> drain) over a problem that does not fill the chip,
// and the two hidden activations make a round trip through memory.  Here a block owns 128 rows for the whole chain (with 32-row
// blocks the 256 KB of weights per block made 230 MB of L2 traffic per call and the kernel took 40 us):
//   * the row tile goes to LDS once (XOR-swizzled 512-byte rows, conflict-free ds_read_b128 fragments, as in gemm_wreg.hip);
//   * each of the 8 waves holds its 32 output columns of the current layer's weights in registers (MFMA A operand), so weights never
//     pass through LDS; the next layer's weights are requested while the current epilogue runs;
//   * relu(acc + b) is rounded to the storage type and written back to LDS as the next layer's row tile (same rounding points and
//     the same k order as the separate launches: the hidden activations are bit-identical);
//   * the 4-output layer is a dot over the row: lane partials -> xor-shuffles over the four lane groups -> per-wave partials in LDS
//     -> every thread finishes one (row, output), including sigmoid(y + inverse_sigmoid(ref)) or the anchor add.
#include "common.hpp"

namespace moy {

template <typename T>
__device__ __forceinline__ f32x4 mlp_mfma(f32x4 acc, u32x4 w, u32x4 a);
template <>
__device__ __forceinline__ f32x4 mlp_mfma<bf16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mlp_mfma<f16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}

__device__ __forceinline__ float mlp_inv_sigmoid(float x) {   // nn/modules/utils.py:34-38, eps 1e-5
  x = fminf(fmaxf(x, 0.f), 1.f);
  return logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
}

struct MlpParams {
  const void* X;
  int64_t ldx;
  const int32_t* x_rows;
  int M;
  const void* W0;
  const float* b0;
  const void* W1;
  const float* b1;
  const float* w2;   // [4][256] fp32
  const float* b2;   // [4]
  int mode;          // as moy_rowdot: 0 plain, 1 sigmoid(y + inverse_sigmoid(aux[m])), 2 y + aux[aux_rows[m]]
  const float* aux;
  const int32_t* aux_rows;
  float* y;          // [M][4]
};

constexpr int MLP_BM = 128, MLP_NW = 8;
constexpr int MLP_LDS = 2 * MLP_BM * 512 + 2 * 1024 + 4 * 1024 + MLP_BM * MLP_NW * 4 * 4;

template <typename T>
__global__ __launch_bounds__(64 * MLP_NW) void mlp_head_kernel(const MlpParams p) {
  constexpr int BM = MLP_BM, NW = MLP_NW, NTHR = 64 * NW, MT = BM / 16, NT = 256 / (NW * 16), WC = NT * 16;   // WC columns per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* XA = smem;
  unsigned char* XB = smem + BM * 512;
  float* sb0 = reinterpret_cast<float*>(smem + 2 * BM * 512);
  float* sb1 = sb0 + 256;
  float* sw2 = sb1 + 256;                  // [4][256]
  float* P = sw2 + 1024;                   // [BM][NW waves][4 outputs]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const T* Xg = static_cast<const T*>(p.X);

  // ---- row tile -> registers (4 x 16 B per thread), first layer's weights -> registers, small vectors -> LDS
  u32x4 xr[BM * 32 / NTHR];
#pragma unroll
  for (int k = 0; k < BM * 32 / NTHR; ++k) {
    const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
    const int m = min(m0 + row, p.M - 1);
    const int64_t src = p.x_rows ? (int64_t)p.x_rows[m] : (int64_t)m;
    xr[k] = *reinterpret_cast<const u32x4*>(Xg + src * p.ldx + c * 8);
  }
  u32x4 wf[NT][8];
  auto load_w = [&](const void* W) {
    const T* Wg = static_cast<const T*>(W) + (int64_t)(wave * WC) * 256;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int pn = 0; pn < 8; ++pn) wf[j][pn] = *reinterpret_cast<const u32x4*>(Wg + (j * 16 + r) * 256 + pn * 32 + q * 8);
  };
  load_w(p.W0);
  if (tid < 256) {
    sb0[tid] = p.b0[tid];
    sb1[tid] = p.b1[tid];
#pragma unroll
    for (int o = 0; o < 4; ++o) sw2[o * 256 + tid] = p.w2[o * 256 + tid];
  }
#pragma unroll
  for (int k = 0; k < BM * 32 / NTHR; ++k) {
    const int id = tid + k * NTHR, row = id >> 5, c = id & 31;
    *reinterpret_cast<u32x4*>(XA + row * 512 + ((c ^ (row & 15)) << 4)) = xr[k];
  }
  __syncthreads();

  const int lbase = r * 512 + ((q ^ r) << 4);      // fragment of row i*16 + r, chunk (pn*4 + q) ^ r
  auto gemm = [&](const unsigned char* As, f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 af[2][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[0][i] = *reinterpret_cast<const u32x4*>(As + (lbase + i * 8192));
#pragma unroll
    for (int pn = 0; pn < 8; ++pn) {
      if (pn + 1 < 8) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[(pn + 1) & 1][i] = *reinterpret_cast<const u32x4*>(As + ((lbase ^ ((pn + 1) * 64)) + i * 8192));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mlp_mfma<T>(acc[i][j], wf[j][pn], af[pn & 1][i]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  f32x4 acc[MT][NT];
  // ---- layer 0: t1 = relu(x . W0^T + b0) -> XB (storage type)
  gemm(XA, acc);
  load_w(p.W1);                                      // in flight under the epilogue below
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = wave * WC + j * 16 + q * 4;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(sb0 + n);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const f32x4 v = __builtin_elementwise_max(acc[i][j] + bb, f32x4{0.f, 0.f, 0.f, 0.f});
      const int row = i * 16 + r;
      *reinterpret_cast<u32x2*>(XB + row * 512 + (((n >> 3) ^ (row & 15)) << 4) + (n & 7) * 2) =
          u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
    }
  }
  __syncthreads();
  // ---- layer 1: t2 = relu(t1 . W1^T + b1), rounded to the storage type as the separate launch stored it
  gemm(XB, acc);
  float part[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int o = 0; o < 4; ++o) part[i][o] = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = wave * WC + j * 16 + q * 4;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(sb1 + n);
    f32x4 w2v[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) w2v[o] = *reinterpret_cast<const f32x4*>(sw2 + o * 256 + n);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x4 v = __builtin_elementwise_max(acc[i][j] + bb, f32x4{0.f, 0.f, 0.f, 0.f});
      const uint32_t lo = DT<T>::pack2(v.x, v.y), hi = DT<T>::pack2(v.z, v.w);
      v = f32x4{DT<T>::lo(lo), DT<T>::hi(lo), DT<T>::lo(hi), DT<T>::hi(hi)};
#pragma unroll
      for (int o = 0; o < 4; ++o) part[i][o] += (v.x * w2v[o].x + v.y * w2v[o].y) + (v.z * w2v[o].z + v.w * w2v[o].w);
    }
  }
  // ---- layer 2: 4 dots per row
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
      v += p.b2[o];
      if (p.mode == 1) v = sigmoidf_(v + mlp_inv_sigmoid(p.aux[(int64_t)m * 4 + o]));
      else if (p.mode == 2) v = v + p.aux[(int64_t)p.aux_rows[m] * 4 + o];
      p.y[(int64_t)m * 4 + o] = v;
    }
  }
}

}  // namespace moy

using namespace moy;

extern "C" int moy_mlp_head(const void* X, int64_t ldx, const int32_t* x_rows, int M, const void* W0, const float* b0,
                            const void* W1, const float* b1, const float* w2, const float* b2, int mode, const float* aux,
                            const int32_t* aux_rows, float* y, int dtype, void* stream) {
  if (!X || !W0 || !b0 || !W1 || !b1 || !w2 || !b2 || !y || M <= 0) return MOY_EINVAL;
  if (mode < 0 || mode > 2 || (mode != 0 && !aux) || (mode == 2 && !aux_rows)) return MOY_EINVAL;
  if (dtype != MOY_BF16 && dtype != MOY_F16) return MOY_ENOSYS;      // fp32: moy_gemm x 2 + moy_rowdot (the parity path)
  if ((ldx % 8) || ldx < 256 || !aligned16(X) || !aligned16(W0) || !aligned16(W1) || !aligned16(w2)) return MOY_EINVAL;
  MlpParams p{X, ldx, x_rows, M, W0, b0, W1, b1, w2, b2, mode, aux, aux_rows, y};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = (M + MLP_BM - 1) / MLP_BM;
  static bool attr_set = false;          // > 64 KiB of dynamic LDS: opt in once per kernel symbol
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_head_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_head_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  if (dtype == MOY_BF16)
    hipLaunchKernelGGL((mlp_head_kernel<bf16_t>), dim3(blocks), dim3(64 * MLP_NW), MLP_LDS, st, p);
  else
    hipLaunchKernelGGL((mlp_head_kernel<f16_t>), dim3(blocks), dim3(64 * MLP_NW), MLP_LDS, st, p);
  return launch_status();
}
