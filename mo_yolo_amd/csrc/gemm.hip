// Fused implicit-GEMM for gfx950: 1x1 / 3x3 convolutions and linears with BN/bias, activation,
// residual and LayerNorm epilogues (see moy_gemm in include/moyolo.h).
//
// Structure (wave64; 256 or 512 threads = 4 or 8 waves per block):
//   * block tile BM x BN, k advanced PANELS panels of 64 bytes per stage
//     (bf16: 32 k per panel -> one v_mfma_f32_16x16x32_bf16 per 16x16 sub-tile and panel;
//      f32 : 16 k per panel -> four v_mfma_f32_16x16x4_f32, the exact-fp32 matrix op);
//   * global -> registers (buffer loads, 16-byte chunks, 8 rows x 128 B per wave instruction, prefetched
//     one stage ahead) -> LDS, two LDS stages,
//     one barrier per stage; LDS rows are 64 B with the 16-B column XOR-swizzled by the row group
//     so the ds_read_b128 fragment reads are bank-conflict free;
//   * MFMA operands are swapped (weights as A, activations as B) so each lane ends up with four
//     consecutive output channels of one pixel -> 16-byte LDS stores of the accumulator tile;
//   * epilogue: scale/shift/activation on the accumulators -> LDS (fp32 tile) -> fully unrolled
//     row-wise pass (residual, optional LayerNorm / narrow head with wave shuffles) -> 16-byte stores.
//   * blockIdx is remapped so that the column tiles of one row tile run on the same XCD (they
//     re-read the same activation rows from that XCD's L2).

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
  int64_t a_bytes;   // bytes addressable from A (bounds of the buffer descriptor)
  const void* W;
  int M, N, K, Kpad;
  int ksize, stride;
  int Hin, Win, Hout, Wout, Cin;
  uint32_t cin_magic;   // ceil(2^32 / Cin): tap = umulhi(kc, magic) is exact for kc < 2^32 / Cin
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
  const float* dot_w;
  const float* dot_b;
  float* dot_out;
  int dot_n;
  int wide_store;   // bf16 output, N % 8 == 0, 16-byte aligned rows: 8 columns per store
  int tiles_n, nblocks;
  FastDiv fd_tiles_n, fd_hw, fd_wout, fd_mask, fd_rpb;
  const float* pre;    // accumulator seed: nearest-2x upsampled rows of a half-resolution product (moy_gemm_args.pre)
  int64_t ld_pre;
  int pre_w, pre_hw, pre_loww, pre_lowhw;
  int a2_cols;         // A2 applies to column tiles below this column (block-uniform)
  int plane_cols;      // 0, or: column n lives in plane n / plane_cols at C + plane * plane_stride (column n % plane_cols)
  int64_t plane_stride;
  FastDiv fd_pre_hw, fd_pre_w;
  // ROW RUNS of a score-only launch (moy_gemm_args.run_*; round 6: also in this kernel, for the fp32 engines' folded head): M counts
  // COMPACT rows; compact row c -> b = c / run_nv, v = c % run_nv -> level l by the compact starts -> (y, x) by the run length ->
  // token = tok0[l] + y * pitch[l] + x; A row = b * run_a_period + token - run_a_off, score row = b * run_period + token
  int run_levels, run_period, run_nv, run_a_period, run_a_off;
  int run_cstart[4], run_len[4], run_tok0[4], run_pitch[4];
  FastDiv fd_nv, fd_len[4];
};

// compact row -> token of its frame and the frame index (see GemmParams.run_*).  Select chains, no dynamic index: indexing a by-value
// kernel-parameter array with a runtime value makes hipcc copy the whole struct to scratch (the first version of this function did,
// and the score launches ran 2.5 x slower per row)
__device__ __forceinline__ void run_token(const GemmParams& p, int c, int& b, int& tok) {
  b = (int)fdiv((uint32_t)c, p.fd_nv);
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
  tok = tok0 + (int)y * pitch + (int)(vv - y * (uint32_t)len);
}

// 16-B column swizzle: lanes of one ds_read_b128 lane group hit distinct bank quartets.
__device__ __forceinline__ int swz(int row, int q) { return q ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3); }

template <typename T>
__device__ __forceinline__ u32x4 add_chunks(u32x4 a, u32x4 b);
template <>
__device__ __forceinline__ u32x4 add_chunks<float>(u32x4 a, u32x4 b) {
  f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
  return __builtin_bit_cast(u32x4, x + y);
}
template <typename T>
__device__ __forceinline__ u32x4 add_chunks16(u32x4 a, u32x4 b) {
  using D = DT<T>;
  u32x4 r;
  r.x = D::pack2(D::lo(a.x) + D::lo(b.x), D::hi(a.x) + D::hi(b.x));
  r.y = D::pack2(D::lo(a.y) + D::lo(b.y), D::hi(a.y) + D::hi(b.y));
  r.z = D::pack2(D::lo(a.z) + D::lo(b.z), D::hi(a.z) + D::hi(b.z));
  r.w = D::pack2(D::lo(a.w) + D::lo(b.w), D::hi(a.w) + D::hi(b.w));
  return r;
}
template <>
__device__ __forceinline__ u32x4 add_chunks<f32x3_t>(u32x4 a, u32x4 b) { return add_chunks<float>(a, b); }
template <>
__device__ __forceinline__ u32x4 add_chunks<bf16_t>(u32x4 a, u32x4 b) { return add_chunks16<bf16_t>(a, b); }
template <>
__device__ __forceinline__ u32x4 add_chunks<f16_t>(u32x4 a, u32x4 b) { return add_chunks16<f16_t>(a, b); }

template <typename T>
__device__ __forceinline__ void mma_panel(f32x4& acc, u32x4 wfrag, u32x4 afrag);
template <>
__device__ __forceinline__ void mma_panel<bf16_t>(f32x4& acc, u32x4 wfrag, u32x4 afrag) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, afrag),
                                                acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_panel<f16_t>(f32x4& acc, u32x4 wfrag, u32x4 afrag) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wfrag), __builtin_bit_cast(f16x8, afrag), acc, 0, 0, 0);
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

template <>
__device__ __forceinline__ void mma_panel<f32x3_t>(f32x4& acc, u32x4 wfrag, u32x4 afrag) {}   // (the split form multiplies in compute_stage)

// MOY_F32X3: a 16-byte chunk of four fp32 values -> their fp16 heads and the fp16 of the scaled remainders (x = hi + lo * 2^-11).
// x - float(hi) is exact in fp32 (hi is within half an fp16 ulp of x), so nothing is lost before the second conversion.
__device__ __forceinline__ void split_f16x3(u32x4 c, u32x2& hi, u32x2& lo) {
  typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
  const f32x4 x = __builtin_bit_cast(f32x4, c);
  const uint32_t h01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{x.x, x.y}, h2_));
  const uint32_t h23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{x.z, x.w}, h2_));
  // lo = fp16(x * 2^11 - hi * 2^11): v_fma_mixlo / mixhi_f16 read the fp16 head in place and round the fp32 fma straight into the low /
  // high half of the result (x * 2^11 - hi * 2^11 is exact in fp32, so this is the ONE rounding of the remainder); 8 vector operations
  // per four values instead of 16 (convert back, subtract, scale, convert)
  const f32x4 xs = x * 2048.0f;
  const float m2048 = -2048.0f;
  uint32_t l01, l23;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h01), "v"(m2048), "v"(xs.x));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h01), "v"(m2048), "v"(xs.y));
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h23), "v"(m2048), "v"(xs.z));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h23), "v"(m2048), "v"(xs.w));
  hi = u32x2{h01, h23};
  lo = u32x2{l01, l23};
}

template <int BM, int BN>
constexpr int gemm_lds_bytes() {
  constexpr int stage = 2 * (BM + BN) * PANELS * 64;
  constexpr int epi = BM * (BN + 4) * 4;
  return stage > epi ? stage : epi;
}

// ---- epilogue, part 1 (registers): v = act(acc * scale + shift), then the fp32 tile goes to LDS.
// Each lane owns 4 consecutive output channels per 16x16 sub-tile, so scale/shift are NT float4
// loads per lane, waited for once in straight-line code.
template <int ACT>
__device__ __forceinline__ f32x4 act4(f32x4 v) {
  if (ACT == MOY_ACT_SILU) { v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w); }
  else if (ACT == MOY_ACT_RELU) { v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}); }
  else if (ACT == MOY_ACT_SIGMOID) { v.x = fast_sigmoid(v.x); v.y = fast_sigmoid(v.y); v.z = fast_sigmoid(v.z); v.w = fast_sigmoid(v.w); }
  return v;
}

template <typename T, bool HALF, int ACT, int BN, int TM, int TN, int MT, int NT>
__device__ __forceinline__ void stage_acc_act(const GemmParams& p, f32x4 (&acc)[MT][NT], float* Cs, int n0, int wm, int wn, int r,
                                              int q) {
  constexpr int LDC = BN + 4;
  // All scale/shift loads are issued back to back from always-valid addresses (clamped column; a
  // missing vector reads the weight buffer and is replaced by 1 / 0 afterwards).  With the loads
  // under `if (p.scale)` branches hipcc waited for each one separately: 2*NT dependent L2 round
  // trips per tile in front of the output pass.
  const bool has_sc = p.scale != nullptr, has_sh = p.shift != nullptr;
  const float* scp = has_sc ? p.scale : static_cast<const float*>(p.W);
  const float* shp = has_sh ? p.shift : static_cast<const float*>(p.W);
  f32x4 sc[NT], sh[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * TN + j * 16 + q * 4;
    const int ncl = n < p.N ? n : 0;
    sc[j] = *reinterpret_cast<const f32x4*>(scp + ncl);
    sh[j] = *reinterpret_cast<const f32x4*>(shp + ncl);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (!has_sc) sc[j] = f32x4{1.f, 1.f, 1.f, 1.f};
    if (!has_sh) sh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nl = wn * TN + j * 16 + q * 4;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int ml = wm * TM + i * 16 + r;     // D[n_local = q*4 + reg][m_local = r]
      const f32x4 v = act4<ACT>(acc[i][j] * sc[j] + sh[j]);
      if constexpr (HALF) {
        // the tile goes to LDS already rounded to the output type: half the ds_write traffic (the VGPR->LDS path, ~80 B/clk
        // per CU, is the scarcest resource of the kernel) and no conversion in the output pass.  Row stride BN + 4 halves:
        // the 16 rows of a ds_write_b64 lane group fall on 16 distinct bank pairs.
        if constexpr (!is_f32<T>::value) {
          unsigned char* ch = reinterpret_cast<unsigned char*>(Cs) + ((size_t)ml * (BN + 4) + nl) * 2;
          *reinterpret_cast<u32x2*>(ch) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
        }
      } else {
        *reinterpret_cast<f32x4*>(Cs + ml * LDC + nl) = v;
      }
    }
  }
}

template <typename T, bool HALF, int BN, int TM, int TN, int MT, int NT>
__device__ __forceinline__ void stage_acc(const GemmParams& p, f32x4 (&acc)[MT][NT], float* Cs, int n0, int wm, int wn, int r, int q) {
  switch (p.act) {   // wave-uniform: one branch, not one per element
    case MOY_ACT_SILU: stage_acc_act<T, HALF, MOY_ACT_SILU, BN, TM, TN, MT, NT>(p, acc, Cs, n0, wm, wn, r, q); break;
    case MOY_ACT_RELU: stage_acc_act<T, HALF, MOY_ACT_RELU, BN, TM, TN, MT, NT>(p, acc, Cs, n0, wm, wn, r, q); break;
    case MOY_ACT_SIGMOID: stage_acc_act<T, HALF, MOY_ACT_SIGMOID, BN, TM, TN, MT, NT>(p, acc, Cs, n0, wm, wn, r, q); break;
    default: stage_acc_act<T, HALF, MOY_ACT_NONE, BN, TM, TN, MT, NT>(p, acc, Cs, n0, wm, wn, r, q); break;
  }
}

// element offset of output column n inside C (column planes: moy_gemm_args.plane_cols)
__device__ __forceinline__ int64_t col_off(const GemmParams& p, int n) {
  if (!p.plane_cols) return n;
  const int pl = n / p.plane_cols;
  return (int64_t)pl * p.plane_stride + (n - pl * p.plane_cols);
}

// ---- epilogue, part 2 (after the barrier): row-wise pass over the LDS tile: residual add,
// optional LayerNorm (wave shuffles), coalesced vector stores.  The pass is FULLY UNROLLED with all
// residual loads issued up front: CDNA4's vmcnt counts stores too, so a wait placed inside a rolled
// store loop (as hipcc did for the first version: s_waitcnt vmcnt(0) per iteration) serialises every
// store behind the HBM write latency -- that alone was ~60 % of the 1x1-GEMM time.
template <typename T, int BM, int BN, bool LN, int NTHR>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const float* Cs, int m0, int n0, int tid) {
  constexpr int LDC = BN + 4;
  const int lane = tid & 63, wave = tid >> 6;
  const T* __restrict__ Rg = static_cast<const T*>(p.R);
  constexpr int CPR = LN ? 64 : BN / 4;             // 4-column chunks per row handled per pass
  constexpr int RSTEP = LN ? NTHR / 64 : NTHR / CPR; // rows advanced per pass
  constexpr int NPASS = BM / RSTEP;
  const int cc = LN ? lane : tid % CPR;
  const int rr0 = LN ? wave : tid / CPR;
  const int n = n0 + cc * 4;
  const bool col_ok = n < p.N;                      // N % 4 == 0 (host-checked)
  const int nc = col_ok ? n : 0;                    // clamped column: loads below are always in range
  f32x4 res[NPASS];
  if (Rg) {   // straight-line, branch-free loads (clamped row), so all NPASS are in flight together
    const T* rp = Rg + nc;
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
      const int m = min(m0 + rr0 + k * RSTEP, p.M - 1);
      res[k] = DT<T>::load4(rp + (int64_t)m * p.ldr);
    }
  }
  f32x4 g = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
  f32x4 dw[8];
  if (LN) {
    g = *reinterpret_cast<const f32x4*>(p.ln_g + nc);
    be = *reinterpret_cast<const f32x4*>(p.ln_b + nc);
#pragma unroll
    for (int c = 0; c < 8; ++c) dw[c] = c < p.dot_n ? *reinterpret_cast<const f32x4*>(p.dot_w + c * 256 + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int out_esz = p.out_f32 ? 4 : (int)sizeof(T);
  unsigned char* cbase = static_cast<unsigned char*>(p.C) + col_off(p, n) * out_esz;
#pragma unroll
  for (int k = 0; k < NPASS; ++k) {
    const int rr = rr0 + k * RSTEP, m = m0 + rr;
    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + rr * LDC + cc * 4);
    if (Rg) v += res[k];
    if (LN) {   // one wave per row; lane owns columns lane*4 .. +3 (N == BN == 256)
      const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
      const f32x4 d = v - mean;
      const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.0f / 256.0f);
      v = d * (1.0f / sqrtf(var + 1e-5f)) * g + be;
      if (p.dot_n) {   // fused narrow head on the normalised row (wave-uniform branch)
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < p.dot_n) {
            const float sdot = wave_sum(v.x * dw[c].x + v.y * dw[c].y + v.z * dw[c].z + v.w * dw[c].w);
            if (lane == 0 && m < p.M) {
              int64_t mr = m;
              if (p.run_levels) { int rb_, tok_; run_token(p, m, rb_, tok_); mr = (int64_t)rb_ * p.run_period + tok_; }
              p.dot_out[mr * p.dot_n + c] = sdot + p.dot_b[c];
            }
          }
      }
    }
    if (col_ok && m < p.M && p.C) {   // C == NULL: score-only launch (LayerNorm + narrow head, rows not stored)
      int64_t mo = m;
      if (p.c_rpb) { const int bq = (int)fdiv(m, p.fd_rpb); mo = (int64_t)bq * p.c_bstride + (m - bq * p.c_rpb); }   // wave-uniform rare path
      unsigned char* cp = cbase + mo * p.ldc * out_esz;
      if (p.out_f32)
        *reinterpret_cast<f32x4*>(cp) = v;
      else
        DT<T>::store4(reinterpret_cast<T*>(cp), v);
    }
  }
}

// bf16 output without LayerNorm: 8 columns per thread -> one 16-byte store (half the store
// instructions of the 4-column form; 8-byte-per-lane stores run at ~0.6x the 16-byte rate).
template <typename T, int BM, int BN, int NTHR>
__device__ __forceinline__ void gemm_epilogue_bf16x8(const GemmParams& p, const float* Cs, int m0, int n0, int tid) {
  constexpr int LDC = BN + 4;
  constexpr int CPR = BN / 8, RSTEP = NTHR / CPR, NPASS = BM / RSTEP;
  static_assert(BM % RSTEP == 0, "tile vs threads");
  const T* __restrict__ Rg = static_cast<const T*>(p.R);
  const int cc = tid % CPR, rr0 = tid / CPR;
  const int n = n0 + cc * 8;
  const bool col_ok = n < p.N;                       // N % 8 == 0 on this path (host-checked)
  const int nc = col_ok ? n : 0;
  u32x4 res[NPASS];
  if (Rg) {
    const T* rp = Rg + nc;
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
      const int m = min(m0 + rr0 + k * RSTEP, p.M - 1);
      res[k] = *reinterpret_cast<const u32x4*>(rp + (int64_t)m * p.ldr);
    }
  }
  T* cbase = static_cast<T*>(p.C) + col_off(p, n);
#pragma unroll
  for (int k = 0; k < NPASS; ++k) {
    const int rr = rr0 + k * RSTEP, m = m0 + rr;
    f32x4 v0 = *reinterpret_cast<const f32x4*>(Cs + rr * LDC + cc * 8);
    f32x4 v1 = *reinterpret_cast<const f32x4*>(Cs + rr * LDC + cc * 8 + 4);
    if (Rg) {
      v0 += f32x4{DT<T>::lo(res[k].x), DT<T>::hi(res[k].x), DT<T>::lo(res[k].y), DT<T>::hi(res[k].y)};
      v1 += f32x4{DT<T>::lo(res[k].z), DT<T>::hi(res[k].z), DT<T>::lo(res[k].w), DT<T>::hi(res[k].w)};
    }
    if (col_ok && m < p.M) {
      int64_t mo = m;
      if (p.c_rpb) { const int bq = (int)fdiv(m, p.fd_rpb); mo = (int64_t)bq * p.c_bstride + (m - bq * p.c_rpb); }
      *reinterpret_cast<u32x4*>(cbase + mo * p.ldc) =
          u32x4{DT<T>::pack2(v0.x, v0.y), DT<T>::pack2(v0.z, v0.w), DT<T>::pack2(v1.x, v1.y), DT<T>::pack2(v1.z, v1.w)};
    }
  }
}

// 16-bit output, no residual, no LayerNorm: the LDS tile is already in the output type (stage_acc HALF): two 8-byte LDS
// reads (rows are 8- but not 16-byte aligned) -> one 16-byte store.
template <typename T, int BM, int BN, int NTHR>
__device__ __forceinline__ void gemm_epilogue_half(const GemmParams& p, const float* Cs, int m0, int n0, int tid) {
  constexpr int CPR = BN / 8, RSTEP = NTHR / CPR, NPASS = BM / RSTEP;
  static_assert(BM % RSTEP == 0, "tile vs threads");
  const int cc = tid % CPR, rr0 = tid / CPR;
  const int n = n0 + cc * 8;
  const bool col_ok = n < p.N;                       // N % 8 == 0 on this path (host-checked)
  const unsigned char* ch = reinterpret_cast<const unsigned char*>(Cs);
  T* cbase = static_cast<T*>(p.C) + col_off(p, n);
#pragma unroll
  for (int k = 0; k < NPASS; ++k) {
    const int rr = rr0 + k * RSTEP, m = m0 + rr;
    const u32x2 lo = *reinterpret_cast<const u32x2*>(ch + ((size_t)rr * (BN + 4) + cc * 8) * 2);
    const u32x2 hi = *reinterpret_cast<const u32x2*>(ch + ((size_t)rr * (BN + 4) + cc * 8 + 4) * 2);
    if (col_ok && m < p.M) {
      int64_t mo = m;
      if (p.c_rpb) { const int bq = (int)fdiv(m, p.fd_rpb); mo = (int64_t)bq * p.c_bstride + (m - bq * p.c_rpb); }
      *reinterpret_cast<u32x4*>(cbase + mo * p.ldc) = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
  }
}

// NSET = 2: global loads run two stages ahead of the MFMAs (K of more than two stages); NSET = 1: one
// stage ahead.  A compile-time choice: with both paths in one kernel hipcc's register allocation and
// wait placement degrade for both (measured).
template <typename T, int BM, int BN, int WGM, int WGN, bool LN, int KS, int NSET>
// (second launch-bounds argument = waves per SIMD.  Lab build only: the eight-wave split-fp16 forms with one stage in flight ask for four,
//  i.e. two blocks per CU -- measured slower, launch_cfg)
__global__ __launch_bounds__(64 * WGM * WGN, (MOY_DIAG && std::is_same<T, f32x3_t>::value && WGM * WGN == 8 && NSET == 1 && !LN && BN == 128 && BM == 128) ? 4
                                             : ((std::is_same<T, f32x3_t>::value && WGM * WGN == 4 && (BM / WGM) * (BN / WGN) == 4096) ? 2 : 1))   // (two four-wave blocks of 64 x 64 wave tiles per CU: <= 256 registers)
void gemm_kernel(const GemmParams p) {
  // X3 (MOY_F32X3): the tensors are fp32 and staged exactly as in the fp32 kernel (32 k per stage: 128 bytes of a row), but a stage's
  // LDS image is TWO fp16 panels of 64 bytes per row -- panel 0 the heads, panel 1 the scaled remainders of the same 32 k -- and a stage
  // is ONE k step of v_mfma_f32_16x16x32_f16, three products per output sub-tile (hi.hi into acc; hi.lo, lo.hi into accx)
  constexpr bool X3 = std::is_same<T, f32x3_t>::value;
  constexpr int KPB = DT<T>::KPB;      // elements per 16-B chunk
  constexpr int BKP = 4 * KPB;         // elements per 64-B panel
  constexpr int BK = BKP * PANELS;     // elements per stage
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MT = TM / 16, NT = TN / 16;
  // Panel 1 of each operand stores row R in the slot of row R ^ 1: the 8 lanes of one ds_write_b128 group are the two
  // panels of ONE row (4 x 16 B each) and would otherwise fall on the same 16 of the 32 store banks (BM*64 B = 0 mod
  // 128 B) -- every staging store was a 2-way conflict (SQ_LDS_BANK_CONFLICT = 29 % of SQ_LDS_IDX_ACTIVE,
  // profiles/r01_e_pmc_gemm.txt).  The fragment reads stay conflict-free: ^1 permutes rows inside a quad.
  // Not for the 128x64 tile: there the two extra address registers push hipcc from 77 to 90 VGPRs, past the 84 that
  // three resident blocks need, and the kernel loses 12 % instead of gaining 5-8 % (same-device A/B, tools/ab_gemm.sh).
  constexpr int SKEW = (BM == 128 && BN == 64) ? 0 : 1;
  constexpr int A_BYTES = BM * PANELS * 64, B_BYTES = BN * PANELS * 64;
  constexpr int NW = WGM * WGN, NTHR = 64 * NW;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(BM % 64 == 0 && BN % 32 == 0, "tile");
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
  const int tile_m = (int)fdiv(bid, p.fd_tiles_n), tile_n = bid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* __restrict__ Ag = static_cast<const T*>(p.A);
  const T* __restrict__ A2g = static_cast<const T*>(p.A2);
  const T* __restrict__ Wg = static_cast<const T*>(p.W);

  // ---- per-thread staging coordinates (loop invariant).  A wave-instruction covers 8 rows x 128 B
  // (both 64-B panels of a row).  All global reads are buffer loads: a wave-uniform descriptor plus a
  // 32-bit per-lane byte offset computed ONCE; rows beyond M / N, masked rows, taps outside the image
  // and the K tail use an out-of-range offset that the hardware range check turns into zeros.  The
  // k advance of a 1x1 conv / linear is the scalar `soffset`, so its main loop issues no address VALU
  // at all (the first version spent ~35 VALU ops per 16-B chunk on 64-bit addressing + predicates
  // and was VALU-bound: profiles/r01_b_pmc_gemm.txt).
  static_assert(PANELS == 2, "staging map assumes two 64-B panels per stage");
  constexpr int RPP = NTHR / 8;                 // rows staged per pass (8 threads per 128-B row)
  constexpr int RA2 = BM / RPP, RB2 = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile vs thread count");
  constexpr int ESZ = 16 / KPB;
  constexpr uint32_t OOB = 0x80000000u;        // >= num_records of every descriptor below
  const int srow = tid >> 3, spn = (tid >> 2) & 1, sq = tid & 3;
  const int kc0 = spn * BKP + sq * KPB;        // this thread's k element inside a stage

  // descriptors (built from kernel arguments / blockIdx only => provably wave-uniform)
  int b0 = 0;
  int64_t a_base = 0;                          // element offset of the descriptor base inside A
  if (KS == 1) {
    if (!p.a_rows && !(LN && p.run_levels)) a_base = (int64_t)m0 * p.lda;
  } else {
    b0 = (int)fdiv(m0, p.fd_hw);
    a_base = (int64_t)b0 * p.Hin * p.Win * p.lda;
  }
  const int64_t a_left = p.a_bytes - a_base * ESZ;
  const uint32_t a_rec = (uint32_t)(a_left < 0x7fffffffLL ? a_left : 0x7fffffffLL);
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag + a_base), 0, a_rec, 0x00020000);
  const auto rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((A2g ? A2g : Ag) + a_base), 0, a_rec, 0x00020000);
  // X3: W arrives PRE-SPLIT -- fp16 [2][N][Kpad], plane 0 the heads, plane 1 the scaled remainders (moyolo.h, MOY_F32X3) -- and is
  // staged like a 16-bit operand: panel = plane, 8 k per 16-byte chunk, no vector arithmetic on the way
  constexpr int WESZ = X3 ? 2 : ESZ;
  const int64_t w_base = (int64_t)n0 * p.Kpad;
  const int64_t w_plane = (int64_t)p.N * p.Kpad * WESZ;                         // bytes of one plane (X3) / of the matrix
  const int64_t w_left = (X3 ? 2 : 1) * w_plane - w_base * WESZ;
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(Wg) + w_base * WESZ), 0,
                                                     (uint32_t)(w_left < 0x7fffffffLL ? w_left : 0x7fffffffLL), 0x00020000);

  uint32_t a_voff[RA2];    // byte offset of (row, kc0) for ksize 1; of pixel (iy0, ix0) channel 0 for ksize 3
  uint32_t a_taps[RA2];    // ksize 3: bit t set <=> tap t of this output pixel lies inside the image
#pragma unroll
  for (int j = 0; j < RA2; ++j) {
    const int m = m0 + srow + j * RPP;
    a_voff[j] = OOB;
    a_taps[j] = 0;
    if (m < p.M) {
      if (KS == 1) {
        if (LN && p.run_levels) {                         // score pass over row runs: compact row -> A row (descriptor base = A itself)
          int rb_, tok_;
          run_token(p, m, rb_, tok_);
          a_voff[j] = (uint32_t)(((int64_t)(rb_ * p.run_a_period + tok_ - p.run_a_off) * p.lda + kc0) * ESZ);
        } else {
        const int arow = p.a_rows ? p.a_rows[m] : m;     // the mask belongs to the A rows (tokens), gathered or not
        const bool masked = p.a_mask && p.a_mask[arow - (int)fdiv(arow, p.fd_mask) * p.mask_period] == 0;
        const int64_t row = p.a_rows ? (int64_t)arow : (int64_t)(m - m0);
        if (!masked) a_voff[j] = (uint32_t)((row * p.lda + kc0) * ESZ);
        }
      } else {
        const int hw = p.Hout * p.Wout;
        const int b = (int)fdiv(m, p.fd_hw), rem = m - b * hw;
        const int oy = (int)fdiv(rem, p.fd_wout), ox = rem - oy * p.Wout;
        const int iy0 = oy * p.stride - 1, ix0 = ox * p.stride - 1;
        a_voff[j] = (uint32_t)((((int64_t)(b - b0) * p.Hin + iy0) * p.Win + ix0) * p.lda * ESZ);   // may wrap below 0: fixed by the tap delta
        uint32_t msk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = iy0 + t / 3, ix = ix0 + t % 3;
          if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) msk |= 1u << t;
        }
        a_taps[j] = msk;
      }
    }
  }
  uint32_t b_voff[RB2];
#pragma unroll
  for (int j = 0; j < RB2; ++j) {
    const int n = n0 + srow + j * RPP;
    if constexpr (X3) b_voff[j] = n < p.N ? (uint32_t)(((int64_t)(srow + j * RPP) * p.Kpad + sq * 8) * 2 + spn * w_plane) : OOB;
    else b_voff[j] = n < p.N ? (uint32_t)(((int64_t)(srow + j * RPP) * p.Kpad + kc0) * ESZ) : OOB;
  }

  u32x4 areg[NSET][RA2], breg[NSET][RB2];

  auto load_stage = [&](int kt, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    const int soff = kt * BK * ESZ;            // wave-uniform k advance in bytes
    if (KS == 1) {
      const bool ktail = kt * BK + kc0 >= p.K; // only the zero-padded tail of K (Kpad > K)
#pragma unroll
      for (int j = 0; j < RA2; ++j) areg[SET][j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, ktail ? OOB : a_voff[j], soff, 0);
      if (A2g && n0 < p.a2_cols) {              // prologue add (q = k = x + pos): second stream, same offsets; block-uniform
        u32x4 t2[RA2];
#pragma unroll
        for (int j = 0; j < RA2; ++j) t2[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA2, ktail ? OOB : a_voff[j], soff, 0);
#pragma unroll
        for (int j = 0; j < RA2; ++j) areg[SET][j] = add_chunks<T>(areg[SET][j], t2[j]);
      }
    } else {
      // (round 6, measured and dropped: with Cin a multiple of the stage's k extent the tap and its byte delta are wave-uniform -- scalar unit, ~12
      //  vector instructions fewer per stage and thread; f32x3 5.26-5.29 k against 5.28 k, exact fp32 +0.4 %, the small-batch leg equal: the
      //  vector issue port is not what paces these loops, and the second code path cost the 256 x 64 form 12 bytes of scratch)
      const int kc = kt * BK + kc0;
      const int tap = (int)__umulhi((unsigned)kc, p.cin_magic), c = kc - tap * p.Cin;   // kc / Cin, kc % Cin (any Cin)
      const int ky = (tap * 11) >> 5, kx = tap - ky * 3;   // tap / 3 for tap < 16
      const uint32_t delta = (uint32_t)(((ky * p.Win + kx) * (int)p.lda + c) * ESZ);
#pragma unroll
      for (int j = 0; j < RA2; ++j) {
        const uint32_t vo = ((a_taps[j] >> tap) & 1u) ? a_voff[j] + delta : OOB;   // tap >= 9 never set
        areg[SET][j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0);
      }
    }
    const int soff_w = X3 ? kt * BK * 2 : soff;            // (X3: 32 fp16 per plane row and stage)
#pragma unroll
    for (int j = 0; j < RB2; ++j) breg[SET][j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, b_voff[j], soff_w, 0);
  };

  auto store_stage = [&](int buf, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* Bs = As + A_BYTES;
    if constexpr (X3) {
      // this thread's chunk = k (spn*16 + sq*4 .. +3) of its row: 8 bytes of fp16 heads at k*2 inside the row's 64-byte head panel
      // (16-byte column k/8 = spn*2 + (sq >> 1), swizzled as the 16-bit kernels swizzle it; half (sq & 1)), same place in the lo panel
      const int c16 = spn * 2 + (sq >> 1), hb = (sq & 1) * 8;
#pragma unroll
      for (int j = 0; j < RA2; ++j) {
        const int row = srow + j * RPP;
        u32x2 hi, lo;
        split_f16x3(areg[SET][j], hi, lo);
        *reinterpret_cast<u32x2*>(As + row * 64 + swz(row, c16) * 16 + hb) = hi;
        *reinterpret_cast<u32x2*>(As + (BM + (row ^ SKEW)) * 64 + swz(row, c16) * 16 + hb) = lo;
      }
#pragma unroll
      for (int j = 0; j < RB2; ++j) {          // (already split: the 16-bit kernels' store, panel = plane)
        const int row = srow + j * RPP;
        *reinterpret_cast<u32x4*>(Bs + (spn * BN + (row ^ (spn * SKEW))) * 64 + swz(row, sq) * 16) = breg[SET][j];
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < RA2; ++j) {
      const int row = srow + j * RPP;
      *reinterpret_cast<u32x4*>(As + (spn * BM + (row ^ (spn * SKEW))) * 64 + swz(row, sq) * 16) = areg[SET][j];
    }
#pragma unroll
    for (int j = 0; j < RB2; ++j) {
      const int row = srow + j * RPP;
      *reinterpret_cast<u32x4*>(Bs + (spn * BN + (row ^ (spn * SKEW))) * 64 + swz(row, sq) * 16) = breg[SET][j];
    }
  };

  f32x4 acc[MT][NT];
  f32x4 accx[X3 ? MT : 1][X3 ? NT : 1];        // X3: the cross products hi.lo + lo.hi, in units of 2^-11
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (X3) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  if (p.pre) {   // wave-uniform; the loads are in flight under the first staging loads
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * TM + i * 16 + r;
      if (m < p.M) {
        const int b = (int)fdiv(m, p.fd_pre_hw), rem = m - b * p.pre_hw;
        const int y = (int)fdiv(rem, p.fd_pre_w), x = rem - y * p.pre_w;
        const float* pr = p.pre + ((int64_t)b * p.pre_lowhw + (y >> 1) * p.pre_loww + (x >> 1)) * p.ld_pre;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + wn * TN + j * 16 + q * 4;
          if (n < p.N) acc[i][j] = *reinterpret_cast<const f32x4*>(pr + n);
        }
      }
    }
  }

  auto compute_stage = [&](int buf) {
    const unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
    const unsigned char* Bs = As + A_BYTES;
    if constexpr (X3) {
      // Round 6: the row sub-tiles in GROUPS of two -- the fragments of a group (heads and remainders of two 16-row tiles: 16 registers)
      // are read just before its twelve products instead of all MT tiles' up front (32 registers at MT = 4); with one stage in flight
      // the 128 x 128 eight-wave form then fits 128 registers, i.e. two resident blocks per CU (measured slower than one block with two
      // stages in flight: launch_cfg).  Inside a group the three sweeps keep two products into
      // ONE accumulator four instructions apart (a dependent v_mfma_f32_16x16x32 chain issues every ~53 cycles instead of 16:
      // DESIGN.md round 3, item 11).  Same products in the same k order per accumulator: the same bits.
      constexpr int GI = MT >= 2 ? 2 : 1;
      u32x4 wh[NT], wl[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * TN + j * 16 + r;
        wh[j] = *reinterpret_cast<const u32x4*>(Bs + row * 64 + swz(row, q) * 16);
        wl[j] = *reinterpret_cast<const u32x4*>(Bs + (BN + (row ^ SKEW)) * 64 + swz(row, q) * 16);
      }
#pragma unroll
      for (int i0 = 0; i0 < MT; i0 += GI) {
        u32x4 ah[GI], al[GI];
#pragma unroll
        for (int g = 0; g < GI; ++g) {
          const int row = wm * TM + (i0 + g) * 16 + r;
          ah[g] = *reinterpret_cast<const u32x4*>(As + row * 64 + swz(row, q) * 16);
          al[g] = *reinterpret_cast<const u32x4*>(As + (BM + (row ^ SKEW)) * 64 + swz(row, q) * 16);
        }
#pragma unroll
        for (int g = 0; g < GI; ++g)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma_panel<f16_t>(acc[i0 + g][j], wh[j], ah[g]);
#pragma unroll
        for (int g = 0; g < GI; ++g)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma_panel<f16_t>(accx[i0 + g][j], wh[j], al[g]);
#pragma unroll
        for (int g = 0; g < GI; ++g)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma_panel<f16_t>(accx[i0 + g][j], wl[j], ah[g]);
      }
      return;
    }
#pragma unroll
    for (int pn = 0; pn < PANELS; ++pn) {
      u32x4 af[MT], wf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * TM + i * 16 + r;
        af[i] = *reinterpret_cast<const u32x4*>(As + (pn * BM + (row ^ (pn * SKEW))) * 64 + swz(row, q) * 16);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * TN + j * 16 + r;
        wf[j] = *reinterpret_cast<const u32x4*>(Bs + (pn * BN + (row ^ (pn * SKEW))) * 64 + swz(row, q) * 16);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mma_panel<T>(acc[i][j], wf[j], af[i]);
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, NSET - 1>;
  // (round 6, measured and dropped: hipcc sinks the stage's global loads below its products and hoists the LDS store of the other register
  //  set above them, so the loads fly for half a stage on average, not two; pinning the source order with sched_barrier -- loads first,
  //  optionally the store last -- was SLOWER: f32x3 4.85 -> 4.63 / 4.75 k, exact fp32 3.17 -> 3.15 / 3.08 k frames/s.  The k loop of the
  //  fp32-tensor forms is paced by the LDS pipe, not by the global round trip: dispatch_tile, the 256-row tiles.)
  const int nk = p.Kpad / BK;
  if constexpr (NSET == 1) {
    load_stage(0, S0{});
    store_stage(0, S0{});
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) load_stage(kt + 1, S0{});
      compute_stage(buf);
      if (kt + 1 < nk) store_stage(buf ^ 1, S0{});
      __syncthreads();
    }
  } else if constexpr (NSET >= 3) {
    // Round 6 (lab build: measured slower, launch_cfg): NSET stages of global loads in flight (NSET even: LDS buffer of stage s is s & 1,
    // register set s % NSET).
    static_assert(NSET % 2 == 0, "stage s lives in LDS buffer s & 1 and in register set s % NSET");
    auto for_sets = [&](auto&& f) { [&]<int... U>(std::integer_sequence<int, U...>) { (f(std::integral_constant<int, U>{}), ...); }(std::make_integer_sequence<int, NSET>{}); };
    for_sets([&](auto uc) { constexpr int u = decltype(uc)::value; if (u < nk) load_stage(u, uc); });
    store_stage(0, S0{});
    __syncthreads();
    int kt = 0;
    // steady state: every prefetch of the body is a real stage (kt + u + NSET < nk for every u), no branch around a load
    for (; kt + 2 * NSET <= nk; kt += NSET) {
      for_sets([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        load_stage(kt + u + NSET, uc);                 // set u was stored to LDS as stage kt + u
        compute_stage(u & 1);
        store_stage((u + 1) & 1, std::integral_constant<int, (u + 1) % NSET>{});
        __syncthreads();
      });
    }
    // tail: fewer than 2 NSET stages left, the same body with every step guarded (wave-uniform conditions)
    for (; kt < nk; kt += NSET) {
      for_sets([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if (kt + u < nk) {
          if (kt + u + NSET < nk) load_stage(kt + u + NSET, uc);
          compute_stage(u & 1);
          if (kt + u + 1 < nk) store_stage((u + 1) & 1, std::integral_constant<int, (u + 1) % NSET>{});
          __syncthreads();
        }
      });
    }
  } else {
    load_stage(0, S0{});
    if (nk > 1) load_stage(1, S1{});
    store_stage(0, S0{});
    __syncthreads();
    int kt = 0;
    // steady state: straight-line body (no branch around the prefetch), so the compiler's counted
    // vmcnt leaves the newest stage in flight while the older one is written to LDS
    for (; kt + 3 < nk; kt += 2) {
      load_stage(kt + 2, S0{});                      // set 0 was stored to LDS as stage kt
      compute_stage(0);
      store_stage(1, S1{});
      __syncthreads();
      load_stage(kt + 3, S1{});
      compute_stage(1);
      store_stage(0, S0{});
      __syncthreads();
    }
    // tail: 1..3 stages left; LDS buffer 0 holds stage kt, set 1 holds stage kt+1 (if any)
    if (kt + 2 < nk) load_stage(kt + 2, S0{});
    compute_stage(0);
    if (kt + 1 < nk) {
      store_stage(1, S1{});
      __syncthreads();
      compute_stage(1);
      if (kt + 2 < nk) {
        store_stage(0, S0{});
        __syncthreads();
        compute_stage(0);
      }
    }
    __syncthreads();
  }

  if constexpr (X3) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] += accx[i][j] * (1.0f / 2048.0f);
  }
  // ---- epilogue: accumulators -> LDS tile [BM][BN+4] (fp32, or the output type when nothing else is added)
  float* Cs = reinterpret_cast<float*>(smem);
  if constexpr (!LN && !is_f32<T>::value) {
    if (p.wide_store) {
      if (!p.R) {
        stage_acc<T, true, BN, TM, TN, MT, NT>(p, acc, Cs, n0, wm, wn, r, q);
        __syncthreads();
        gemm_epilogue_half<T, BM, BN, NTHR>(p, Cs, m0, n0, tid);
      } else {
        stage_acc<T, false, BN, TM, TN, MT, NT>(p, acc, Cs, n0, wm, wn, r, q);
        __syncthreads();
        gemm_epilogue_bf16x8<T, BM, BN, NTHR>(p, Cs, m0, n0, tid);
      }
      return;
    }
  }
  stage_acc<T, false, BN, TM, TN, MT, NT>(p, acc, Cs, n0, wm, wn, r, q);
  __syncthreads();
  gemm_epilogue<T, BM, BN, LN, NTHR>(p, Cs, m0, n0, tid);
}

template <typename T, int BM, int BN, int WGM, int WGN, bool LN, int KS, int NSET>
static int launch_inst(GemmParams& p, hipStream_t st) {
  if (plan_only(MOY_KERNEL_TILED)) return MOY_OK;          // moy_gemm_query: the dispatch without the launch (and without touching the attribute statics)
  constexpr int lds = gemm_lds_bytes<BM, BN>();
  auto kern = gemm_kernel<T, BM, BN, WGM, WGN, LN, KS, NSET>;
  static bool attr_set = false;   // > 64 KiB dynamic LDS needs the opt-in once per kernel symbol
  if (lds > 65536 && !attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
        hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(64 * WGM * WGN), lds, st, p);
  return launch_status();
}

template <typename T, int BM, int BN, int WGM, int WGN, bool LN, int KS>
static int launch_cfg(GemmParams& p, hipStream_t st) {
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.nblocks = tiles_m * p.tiles_n;
  p.fd_tiles_n = make_fastdiv(p.tiles_n);
  constexpr int bk = 4 * DT<T>::KPB * PANELS;
  // (the 256 x 128 split form: 128 accumulator registers -- a second stage in flight does not fit 256: 152-200 bytes of scratch)
  if constexpr (std::is_same<T, f32x3_t>::value && (BM / WGM) * (BN / WGN) == 4096) return launch_inst<T, BM, BN, WGM, WGN, LN, KS, 1>(p, st);
  else {
#if MOY_DIAG
  if constexpr (std::is_same<T, f32x3_t>::value && WGM * WGN == 8 && !LN) {
    // round 6, both MEASURED SLOWER than two stages in flight on one resident block (same device, interleaved, whole f32x3 bench):
    //   MOY_X3_NSET=1: one stage in flight, 128 registers, TWO blocks per CU            4.69-4.70 k against 4.90-4.92 k frames/s
    //   MOY_X3_NSET=4: four stages in flight (198 / 253 registers)                       4.61-4.62 k against 4.92-4.93 k
    // (per launch: deep-K convolutions 0.92-0.97 x, K = 288 / 576 forms 1.3-1.4 x) -- the split kernel's stage is bound neither by
    // occupancy nor by the global round trip; what is left is its LDS path (8-byte staging stores of the split halves, 12 fragment
    // reads and one barrier per 24 products)
    static const int x3nset = knob("MOY_X3_NSET", 2);
    if (x3nset == 1) return launch_inst<T, BM, BN, WGM, WGN, LN, KS, 1>(p, st);
    if (x3nset == 4 && p.Kpad / bk > 4) return launch_inst<T, BM, BN, WGM, WGN, LN, KS, 4>(p, st);
  }
#endif
  // (round 6, measured and dropped: four stages in flight for the SMALL launches of the 16-bit forms -- at most one block per compute unit,
  //  the small-batch leg -- 3.33-3.34 k against 3.37 k frames/s at four frames per step: their k loop is not paced by the global round trip)
  // prefetch distance 2 pays from three k-stages on and for tiles at least 64 columns wide (measured)
  if (BN >= 64 && p.Kpad / bk > 2) return launch_inst<T, BM, BN, WGM, WGN, LN, KS, 2>(p, st);
  return launch_inst<T, BM, BN, WGM, WGN, LN, KS, 1>(p, st);
  }
}

// ------------------------------------------------------------------------------------------------
// Direct 3x3 stride-1 convolution for 16-bit types (the bottleneck convs of the C2f blocks: Cin = Cout in {32, 64, 128}).
// The implicit GEMM above re-reads every activation row once per tap: 9 x (BM x Cin) bytes through buffer loads and
// ds_write_b128 (the ~80 B/clk VGPR->LDS path) per output tile.  Here the (TH+2) x 18 pixel halo patch of a TH x 16 output
// tile goes to LDS ONCE and the nine taps are nine shifted views of it; only the (small, L2-resident) weight tap tiles stream
// through a two-stage LDS ring.  Staged bytes per 256 output pixels at C = 64: 41 KB patch + 9 x 8 KB weights vs 440 KB.
//   * patch / weight rows are [row][C] with the 16-B chunk index XORed by a function of the row, chosen by exhaustive search
//     so that every ds_read_b128 lane group is conflict-free for ANY starting pixel (hence any tap shift):
//     C = 32: (row >> 1) & 3;  C = 64: row & 7;  C = 128: (row & 7) << 1;
//   * pixels outside the image are out-of-range buffer offsets -> zeros (= the conv's zero padding), no predicates;
//   * MFMA operands swapped as above (weights as A): a lane owns 4 consecutive output channels of one pixel.
template <int C>
__device__ __forceinline__ int conv_swz(int row) {
  return C == 32 ? ((row >> 1) & 3) : (C == 64 ? (row & 7) : ((row & 7) << 1));
}

template <int C, int BN, int TH>
constexpr int conv_stage_bytes() { return (TH + 2) * 18 * C * 2 + (C == 32 ? 9 : 2) * BN * C * 2; }

template <typename T, int C, int BN, int TH, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void conv3x3_direct_kernel(const GemmParams p) {
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int TW = 16, PH = TH + 2, PW = TW + 2, NPIX = PH * PW;
  constexpr int CPP = C / 8;                      // 16-B chunks per pixel / weight row
  constexpr int NW = WGM * WGN, NTHR = 64 * NW;
  constexpr int MT = TH / WGM, NT = BN / 16 / WGN, TM = MT * 16, TN = NT * 16;
  constexpr int BM = TH * TW;
  constexpr int PATCH_BYTES = NPIX * C * 2, WTAP_BYTES = BN * C * 2;
  constexpr int NPL = (NPIX * CPP + NTHR - 1) / NTHR;     // patch chunks per thread
  constexpr int NWL = (BN * CPP + NTHR - 1) / NTHR;       // weight-tap chunks per thread
  constexpr uint32_t OOB = 0x80000000u;
  constexpr bool WRES = C == 32;                  // all nine weight taps resident (18 KB): no ring, no barrier per tap
  static_assert(TH % WGM == 0 && (BN / 16) % WGN == 0, "tile vs waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* patch = smem;
  unsigned char* wbuf = smem + PATCH_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave / WGN, wn = wave % WGN;

  int bid = blockIdx.x;
  {
    const int nb = p.nblocks, qd = nb >> 3, rm = nb & 7, x = bid & 7;
    bid = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + (bid >> 3);
  }
  // logical id -> (image b, tile row, tile column, channel tile), channel tile fastest (they share the patch in L2)
  const int tiles_x = (p.Wout + TW - 1) / TW, tiles_y = (p.Hout + TH - 1) / TH;
  int t = bid;
  const int tn = t % p.tiles_n; t /= p.tiles_n;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;

  const T* __restrict__ Ag = static_cast<const T*>(p.A);
  const T* __restrict__ Wg = static_cast<const T*>(p.W);
  const int64_t img_elems = (int64_t)p.Hin * p.Win * p.lda;
  const int64_t a_left = p.a_bytes - (int64_t)b * img_elems * 2;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag + (int64_t)b * img_elems), 0,
                                                     (uint32_t)(a_left < 0x7fffffffLL ? a_left : 0x7fffffffLL), 0x00020000);
  const int64_t w_left = ((int64_t)p.N - n0) * p.Kpad * 2;
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wg + (int64_t)n0 * p.Kpad), 0,
                                                     (uint32_t)(w_left < 0x7fffffffLL ? w_left : 0x7fffffffLL), 0x00020000);

  // ---- the halo patch: all loads first, then the LDS stores
  u32x4 preg[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const int i = tid + k * NTHR;
    const int pix = i / CPP, ch = i - pix * CPP;
    const int py = pix / PW, px = pix - py * PW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool ok = i < NPIX * CPP && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
    const uint32_t vo = ok ? (uint32_t)((((int64_t)iy * p.Win + ix) * p.lda + ch * 8) * 2) : OOB;
    preg[k] = __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0);
  }
  uint32_t w_voff[NWL];
  int w_lds[NWL];
#pragma unroll
  for (int k = 0; k < NWL; ++k) {
    const int i = tid + k * NTHR;
    const int row = i / CPP, ch = i - row * CPP;
    const bool ok = i < BN * CPP && n0 + row < p.N;
    w_voff[k] = ok ? (uint32_t)(((int64_t)row * p.Kpad + ch * 8) * 2) : OOB;
    w_lds[k] = i < BN * CPP ? row * C * 2 + ((ch ^ conv_swz<C>(row)) * 16) : -1;
  }
  // Weight taps travel global -> registers -> LDS ring; the registers of tap t + PFD are requested while tap t is computed
  // (one tap is only (C/32)*MT*NT MFMAs per wave, shorter than an L2 round trip: with distance 1 every tap stalled on it).
  // The tap loop is fully unrolled so the register sets rotate statically.  C = 32: all nine taps resident, no ring.
  constexpr int PFD = WRES ? 9 : (C == 64 ? 4 : 3);
  u32x4 wreg[9][NWL];
  auto load_w = [&](auto tap_c) {
    constexpr int tap = decltype(tap_c)::value;
#pragma unroll
    for (int k = 0; k < NWL; ++k) wreg[tap][k] = __builtin_amdgcn_raw_buffer_load_b128(rsW, w_voff[k], tap * C * 2, 0);
  };
  auto store_w = [&](auto tap_c) {
    constexpr int tap = decltype(tap_c)::value;
    constexpr int buf = WRES ? tap : (tap & 1);
#pragma unroll
    for (int k = 0; k < NWL; ++k)
      if (w_lds[k] >= 0) *reinterpret_cast<u32x4*>(wbuf + buf * WTAP_BYTES + w_lds[k]) = wreg[tap][k];
  };
  auto for_taps = [&](auto&& f, auto lo_c, auto hi_c) {       // static loop over taps [lo, hi)
    constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, lo + I>{}), ...); }
    (std::make_integer_sequence<int, (hi > lo ? hi - lo : 0)>{});
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I9 = std::integral_constant<int, 9>;
  using IP = std::integral_constant<int, (PFD < 9 ? PFD : 9)>;
  for_taps(load_w, I0{}, IP{});
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const int i = tid + k * NTHR;
    if (i < NPIX * CPP) {
      const int pix = i / CPP, ch = i - pix * CPP;
      *reinterpret_cast<u32x4*>(patch + pix * C * 2 + ((ch ^ conv_swz<C>(pix)) * 16)) = preg[k];
    }
  }
  store_w(I0{});
  if constexpr (WRES) for_taps(store_w, I1{}, I9{});
  __syncthreads();

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto tap_body = [&](auto tap_c) {
    constexpr int tap = decltype(tap_c)::value;
    if constexpr (!WRES && tap + PFD < 9) load_w(std::integral_constant<int, tap + PFD>{});
    constexpr int ky = tap / 3, kx = tap % 3;
    const unsigned char* wb = wbuf + (WRES ? tap : (tap & 1)) * WTAP_BYTES;
#pragma unroll
    for (int cc = 0; cc < C / 32; ++cc) {
      u32x4 af[MT], wf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int pix = (wm * MT + i + ky) * PW + kx + r;
        af[i] = *reinterpret_cast<const u32x4*>(patch + pix * C * 2 + (((cc * 4 + q) ^ conv_swz<C>(pix)) * 16));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * TN + j * 16 + r;
        wf[j] = *reinterpret_cast<const u32x4*>(wb + row * C * 2 + (((cc * 4 + q) ^ conv_swz<C>(row)) * 16));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mma_panel<T>(acc[i][j], wf[j], af[i]);
    }
    if constexpr (!WRES) {
      if constexpr (tap + 1 < 9) store_w(std::integral_constant<int, tap + 1>{});
      __syncthreads();
    }
  };
  for_taps(tap_body, I0{}, I9{});
  if constexpr (WRES) __syncthreads();

  // ---- epilogue: BN/bias + activation in registers, tile through LDS (output type, or fp32 when a residual is added),
  // rows of the tile are the TH x 16 pixels in raster order
  float* Cs = reinterpret_cast<float*>(smem);
  const T* __restrict__ Rg = static_cast<const T*>(p.R);
  constexpr int CPR = BN / 8, RSTEP = NTHR / CPR, NPASS = BM / RSTEP;
  static_assert(BM % RSTEP == 0, "tile vs threads");
  const int cc8 = tid % CPR, rr0 = tid / CPR;
  const int n = n0 + cc8 * 8;
  T* cbase = static_cast<T*>(p.C) + n;
  auto out_row = [&](int rr, int64_t& m) {          // tile row -> output row index; false outside the image
    const int oy = y0 + rr / TW, ox = x0 + (rr % TW);
    m = ((int64_t)b * p.Hout + oy) * p.Wout + ox;
    return oy < p.Hout && ox < p.Wout && n < p.N;
  };
  if (!Rg) {
    stage_acc<T, true, BN, TM, TN, MT, NT>(p, acc, Cs, n0, wm, wn, r, q);
    __syncthreads();
    const unsigned char* ch = reinterpret_cast<const unsigned char*>(Cs);
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
      const int rr = rr0 + k * RSTEP;
      const u32x2 lo = *reinterpret_cast<const u32x2*>(ch + ((size_t)rr * (BN + 4) + cc8 * 8) * 2);
      const u32x2 hi = *reinterpret_cast<const u32x2*>(ch + ((size_t)rr * (BN + 4) + cc8 * 8 + 4) * 2);
      int64_t m;
      if (out_row(rr, m)) *reinterpret_cast<u32x4*>(cbase + m * p.ldc) = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
  } else {
    constexpr int LDC = BN + 4;
    stage_acc<T, false, BN, TM, TN, MT, NT>(p, acc, Cs, n0, wm, wn, r, q);
    u32x4 res[NPASS];
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {               // clamped, branch-free: all residual loads in flight together
      const int rr = rr0 + k * RSTEP;
      const int oy = min(y0 + rr / TW, p.Hout - 1), ox = min(x0 + (rr % TW), p.Wout - 1);
      const int64_t m = ((int64_t)b * p.Hout + oy) * p.Wout + ox;
      res[k] = *reinterpret_cast<const u32x4*>(Rg + m * p.ldr + (n < p.N ? n : 0));
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
      const int rr = rr0 + k * RSTEP;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(Cs + rr * LDC + cc8 * 8);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(Cs + rr * LDC + cc8 * 8 + 4);
      v0 += f32x4{DT<T>::lo(res[k].x), DT<T>::hi(res[k].x), DT<T>::lo(res[k].y), DT<T>::hi(res[k].y)};
      v1 += f32x4{DT<T>::lo(res[k].z), DT<T>::hi(res[k].z), DT<T>::lo(res[k].w), DT<T>::hi(res[k].w)};
      int64_t m;
      if (out_row(rr, m))
        *reinterpret_cast<u32x4*>(cbase + m * p.ldc) =
            u32x4{DT<T>::pack2(v0.x, v0.y), DT<T>::pack2(v0.z, v0.w), DT<T>::pack2(v1.x, v1.y), DT<T>::pack2(v1.z, v1.w)};
    }
  }
}

template <typename T, int C, int BN, int TH, int WGM, int WGN>
static int launch_conv_direct(GemmParams& p, int B, hipStream_t st) {
  if (plan_only(MOY_KERNEL_CONV_DIRECT)) return MOY_OK;          // moy_gemm_query: the dispatch without the launch (and without touching the attribute statics)
  constexpr int BM = TH * 16;
  constexpr int stage = conv_stage_bytes<C, BN, TH>();
  constexpr int epi_h = BM * (BN + 4) * 2, epi_f = BM * (BN + 4) * 4;
  const int epi = p.R ? epi_f : epi_h;
  const int lds = stage > epi ? stage : epi;
  auto kern = conv3x3_direct_kernel<T, C, BN, TH, WGM, WGN>;
  static bool attr_set = false;
  if (!attr_set) {
    constexpr int mx = stage > epi_f ? stage : epi_f;
    if (mx > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  p.tiles_n = (p.N + BN - 1) / BN;
  p.nblocks = B * ((p.Hout + TH - 1) / TH) * ((p.Wout + 15) / 16) * p.tiles_n;
  hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(64 * WGM * WGN), lds, st, p);
  return launch_status();
}

// Measured negative (round 1): a persistent tile loop that requests the next tile's halo patch into registers under the last four
// taps.  The 44-84 extra live VGPRs (and the hoisted LDS addresses of the unrolled taps) cost the second resident block per CU:
// C = 32 149 -> 200 us, C = 64 114 -> 150 us (TH 8: 131 us), C = 128 unchanged.  Occupancy hides the prologue better than the
// prefetch does; a BN = 128 tile for C = 128 (patch staged once for all channels) measured equal.
// Direct path for the shapes it wins on; MOY_ENOSYS = not eligible (the caller falls back to the implicit GEMM).
template <typename T>
static int try_conv_direct(GemmParams& p, int B, bool ln, hipStream_t st) {
  if constexpr (sizeof(T) != 2) {
    return MOY_ENOSYS;
  } else {
    static const int mode = knob("MOY_CONV_DIRECT", 1);                   // MOY_CONV_DIRECT=0 switches the path off (A/B runs)
    if (!mode || ln || p.ksize != 3 || p.stride != 1 || !p.wide_store || p.c_rpb || p.out_f32) return MOY_ENOSYS;
    if (p.Cin != p.K / 9 || (p.lda % 8) || (p.N % 32)) return MOY_ENOSYS;
    if ((int64_t)p.Hin * p.Win * p.lda * 2 > 0x7fffffffLL) return MOY_ENOSYS;
    // tile utilisation: partial tiles at the image border compute (and stage) for nothing
    auto util = [&](int th) {
      const long tiles = (long)((p.Hout + th - 1) / th) * ((p.Wout + 15) / 16);
      return (double)p.Hout * p.Wout / (double)(tiles * th * 16);
    };
    const bool big = util(16) >= 0.85;
    if (!big && util(8) < 0.75) return MOY_ENOSYS;
    // every wave covers all BN channels of its pixel rows (NT = BN/16) and 4 (TH 16) or 2 (TH 8) rows of 16 pixels: the
    // larger register tile halves the LDS fragment reads per MFMA, which bound the first version (1 ds_read_b128 per MFMA)
    if (p.Cin == 32 && p.N == 32) return big ? launch_conv_direct<T, 32, 32, 16, 4, 1>(p, B, st) : launch_conv_direct<T, 32, 32, 8, 4, 1>(p, B, st);
    static const int th64 = knob("MOY_TH64", 16);
    if (p.Cin == 64 && p.N % 64 == 0) return (big && th64 == 16) ? launch_conv_direct<T, 64, 64, 16, 4, 1>(p, B, st) : launch_conv_direct<T, 64, 64, 8, 4, 1>(p, B, st);
#if MOY_DIAG
    static const int bn128 = knob("MOY_BN128", 0);
    if (bn128 && p.Cin == 128 && p.N % 128 == 0 && big) return launch_conv_direct<T, 128, 128, 16, 4, 2>(p, B, st);
    if (bn128 == 2 && p.Cin == 128 && p.N % 128 == 0) return launch_conv_direct<T, 128, 128, 8, 4, 2>(p, B, st);
#endif
    if (p.Cin == 128 && p.N % 64 == 0) return big ? launch_conv_direct<T, 128, 64, 16, 4, 1>(p, B, st) : launch_conv_direct<T, 128, 64, 8, 4, 1>(p, B, st);
    return MOY_ENOSYS;
  }
}

template <typename T, int KS>
static int dispatch_tile(GemmParams& p, bool ln, hipStream_t st) {
  if constexpr (std::is_same<T, f32x3_t>::value) {
    // (round 6: the LayerNorm form of the split kind on 128-row tiles for launches of many rows -- the score pass over the valid tokens: 64 x 64
    //  wave tiles, the 128 KB of split weights of a k pass fetched once per 128 rows instead of once per 64)
    static const int x3ln = knob("MOY_X3_LN128", 1);
    if (ln && x3ln && p.M >= 128 * 512) return launch_cfg<T, 128, 256, 2, 4, true, KS>(p, st);
  }
  if (ln) return launch_cfg<T, 64, 256, 2, 4, true, KS>(p, st);
#if MOY_DIAG
  static const int force = knob("MOY_TILE", 0);   // MOY_TILE: tuning knob (tools/bench_gemm.py); 0 = the measured heuristic below
  if (force == 1) return launch_cfg<T, 128, 128, 2, 4, false, KS>(p, st);
  if (force == 2) return launch_cfg<T, 64, 128, 2, 2, false, KS>(p, st);
  if (force == 3) return launch_cfg<T, 128, 64, 4, 2, false, KS>(p, st);
  if (force == 4) return launch_cfg<T, 64, 64, 2, 2, false, KS>(p, st);
#endif
  if constexpr (std::is_same<T, f32x3_t>::value) {
    // Round 6: the split form is paced by its LDS path (counters, profiles/r06_c_x3_gemm_counters.txt: at 128 x 128 on eight waves the
    // fragment reads + staging stores of a stage are 128 KB = 1000 cycles of the 128 B/clk pipe against 768 cycles of matrix work per
    // SIMD, at 128 x 64 88 KB against 384; three products per fragment pair do not help while a wave's register tile is 64 x 32 or
    // 32 x 32).  Wave tiles of 64 x 64 (128 accumulator registers, one stage in flight -- a second register set spills), same device,
    // interleaved, tools/bench_gemm.py BG_DT=f32x3 at 96 frames, times relative to the round-5 choice:
    //   N > 64, 1x1 and 3x3 stride 2:  128 x 128 on FOUR waves, two blocks per CU      0.86 / 0.90-0.93
    //   N > 64, 3x3 stride 1:          256 x 128 on eight waves                         0.95-1.0 (four waves: 1.03)
    //   N = 64, 1x1 and 3x3 stride 2:  256 x 64 on four waves (each all 64 columns)     0.91 / 0.96-0.98
    //   N = 64, 3x3 stride 1:          stays 128 x 64 on eight waves (three resident blocks; 256 x 64 on four or eight waves 1.04-1.18,
    //                                  128 x 64 on four waves 1.0-1.04)
    // (Measured, timing only: all three products into ONE accumulator -- the register set of the cross terms gone, so the 64 x 64 wave tiles
    //  take a second stage in flight, 220-224 registers -- is worth 5.18 -> 5.35 k frames/s on the whole f32x3 bench; a real single-accumulator
    //  split needs the remainders at true scale and ~1.7 x the rounding of this form: not built.)
    static const int x3tile = knob("MOY_X3_TILE", 1);
    const long rows256 = (p.M + 255) / 256;
    if (x3tile && p.N > 64 && rows256 * ((p.N + 127) / 128) >= 256) {
      if (KS == 3 && p.stride == 1) return launch_cfg<T, 256, 128, 4, 2, false, KS>(p, st);
      return launch_cfg<T, 128, 128, 2, 2, false, KS>(p, st);
    }
    if (x3tile && p.N > 32 && p.N <= 64 && rows256 >= 256 && !(KS == 3 && p.stride == 1)) return launch_cfg<T, 256, 64, 4, 1, false, KS>(p, st);
#if MOY_DIAG
    if (x3tile == 7 && p.N <= 32 && rows256 >= 256) return launch_cfg<T, 256, 32, 4, 1, false, KS>(p, st);     // 64 x 32 per wave
#endif
  }
  // Tile choice (measured, tools/bench_gemm.py): 8-wave blocks for the large tiles; fill >= ~2 blocks
  // per CU when the problem allows it, keep tiles large otherwise.
  const long big = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  if (p.N > 64) {
    if (big >= 384) return launch_cfg<T, 128, 128, 2, 4, false, KS>(p, st);
    return launch_cfg<T, 64, 128, 2, 2, false, KS>(p, st);
  }
  const long mid = (long)((p.M + 127) / 128);
  if (p.N <= 32 && mid >= 384) return launch_cfg<T, 128, 32, 4, 1, false, KS>(p, st);   // no half-empty 64-wide tile
  if (mid >= 384) return launch_cfg<T, 128, 64, 4, 2, false, KS>(p, st);
  return launch_cfg<T, 64, 64, 2, 2, false, KS>(p, st);
}

}  // namespace moy

using namespace moy;

namespace moy {
int& cu_limit_slot() {
  static thread_local int v = 0;
  return v;
}
}  // namespace moy

namespace moy {
int& plan_only_slot() {
  static thread_local int v = 0;
  return v;
}
}  // namespace moy

extern "C" int moy_gemm_query(const moy_gemm_args* a, int* kernel) {
  int& slot = moy::plan_only_slot();
  slot = -1;
  const int rc = moy_gemm(a, nullptr);
  if (kernel) *kernel = (rc == MOY_OK && slot > 0) ? slot : 0;
  slot = 0;
  return rc;
}

extern "C" int moy_set_cu_limit(int n_cus) {
  int& v = moy::cu_limit_slot();
  const int prev = v;
  v = n_cus > 0 ? n_cus : 0;
  return prev;
}

extern "C" int moy_gemm(const moy_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->W) return MOY_EINVAL;
  if (!a->C && !(a->ln_g && a->dot_n > 0 && a->dot_out)) return MOY_EINVAL;   // rows may be dropped only when the fused head is the output
  if (a->dtype != MOY_F32 && a->dtype != MOY_BF16 && a->dtype != MOY_F16 && a->dtype != MOY_F32X3) return MOY_EINVAL;
  const bool is32 = a->dtype == MOY_F32 || a->dtype == MOY_F32X3;      // fp32 tensors (MOY_F32X3: split-fp16 matrix arithmetic)
  const int kpb = is32 ? 4 : 8;
  const int esz = is32 ? 4 : 2;
  const int bk = 4 * kpb * PANELS;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return MOY_EINVAL;
  if (a->N % 4) return MOY_EINVAL;
  if (a->ksize != 1 && a->ksize != 3) return MOY_EINVAL;
  if ((a->lda * esz) % 16 || !aligned16(a->A) || !aligned16(a->W)) return MOY_EINVAL;
  if (a->A2 && (!aligned16(a->A2) || a->ksize != 1)) return MOY_EINVAL;
  if (a->a2_cols < 0 || (a->a2_cols % 256) || (a->a2_cols && !a->A2)) return MOY_EINVAL;
  const int out_esz = a->out_f32 ? 4 : esz;
  if ((a->ldc * out_esz) % (4 * out_esz) || (reinterpret_cast<uintptr_t>(a->C) % (4 * out_esz))) return MOY_EINVAL;
  if (a->R && ((a->ldr * esz) % (4 * esz) || reinterpret_cast<uintptr_t>(a->R) % (4 * esz))) return MOY_EINVAL;
  if ((a->scale && !aligned16(a->scale)) || (a->shift && !aligned16(a->shift))) return MOY_EINVAL;
  const bool ln = a->ln_g != nullptr;
  if (ln && (!a->ln_b || a->N != 256 || !aligned16(a->ln_g) || !aligned16(a->ln_b))) return MOY_EINVAL;
  if (a->a_mask && (a->mask_period <= 0 || a->ksize != 1)) return MOY_EINVAL;
  if (a->a_rows && a->ksize != 1) return MOY_EINVAL;

  GemmParams p{};
  {
    // bytes reachable from A: the descriptor bound.  ksize 1: the last row read (gathered rows may be
    // anywhere below a_rows_bound) ; ksize 3: the whole [B, Hin, Win] image stack.
    const int64_t rows = a->ksize == 3 ? (int64_t)a->B * a->Hin * a->Win : (a->a_rows ? a->a_rows_bound : (int64_t)a->M);
    if (a->a_rows && a->a_rows_bound <= 0) return MOY_EINVAL;
    p.a_bytes = ((rows - 1) * a->lda + (a->ksize == 3 ? a->Cin : a->K)) * esz;
    if (a->a_rows && p.a_bytes > 0x7fffffffLL) return MOY_ENOSYS;    // gathered rows need one 2 GiB window
    if (a->ksize == 3) {   // a 128-row tile may span several images: all of them must fit one 2 GiB window
      const int64_t span = 128 / ((int64_t)a->Hout * a->Wout > 0 ? (int64_t)a->Hout * a->Wout : 1) + 2;
      const int64_t img = (int64_t)a->Hin * a->Win * a->lda * esz;
      if (span * img > 0x7fffffffLL && (int64_t)a->B * img > 0x7fffffffLL) return MOY_ENOSYS;
    }
    if (a->ksize == 1 && !a->a_rows && (int64_t)256 * a->lda * esz > 0x7fffffffLL) return MOY_ENOSYS;
  }
  p.A = a->A; p.A2 = a->A2; p.a_rows = a->a_rows; p.a_mask = a->a_mask; p.mask_period = a->mask_period;
  p.lda = a->lda; p.W = a->W; p.M = a->M; p.N = a->N; p.K = a->K;
  p.Kpad = (a->K + bk - 1) / bk * bk;
  p.ksize = a->ksize; p.stride = a->stride;
  p.scale = a->scale; p.shift = a->shift; p.act = a->act; p.R = a->R; p.ldr = a->ldr;
  p.ln_g = a->ln_g; p.ln_b = a->ln_b; p.C = a->C; p.ldc = a->ldc; p.out_f32 = a->out_f32;
  if (a->c_rows_per_batch < 0 || (a->c_rows_per_batch > 0 && a->c_batch_stride < a->c_rows_per_batch)) return MOY_EINVAL;
  p.c_rpb = a->c_rows_per_batch; p.c_bstride = a->c_batch_stride;
  p.fd_rpb = make_fastdiv(p.c_rpb);
  p.fd_mask = make_fastdiv(a->mask_period > 0 ? a->mask_period : 1);
  if (a->dot_n < 0 || a->dot_n > 8) return MOY_EINVAL;
  if (a->dot_n && (!ln || !a->dot_w || !a->dot_b || !a->dot_out || !aligned16(a->dot_w))) return MOY_EINVAL;
  p.dot_w = a->dot_w; p.dot_b = a->dot_b; p.dot_out = a->dot_out; p.dot_n = a->dot_n;
  if (a->pre) {
    if (a->ksize != 1 || a->pre_h <= 0 || a->pre_w <= 0 || (a->pre_h & 1) || (a->pre_w & 1) || a->M % (a->pre_h * a->pre_w)) return MOY_EINVAL;
    if (!aligned16(a->pre) || (a->ld_pre % 4) || a->ld_pre < a->N) return MOY_EINVAL;
    p.pre = a->pre; p.ld_pre = a->ld_pre; p.pre_w = a->pre_w; p.pre_hw = a->pre_h * a->pre_w;
    p.pre_loww = a->pre_w / 2; p.pre_lowhw = (a->pre_h / 2) * (a->pre_w / 2);
    p.fd_pre_hw = make_fastdiv((uint32_t)p.pre_hw); p.fd_pre_w = make_fastdiv((uint32_t)p.pre_w);
  }
  p.a2_cols = a->a2_cols ? a->a2_cols : 0x7fffffff;
  if (a->plane_cols < 0 || (a->plane_cols % 32) || (a->plane_cols && (a->ksize != 1 || a->plane_stride <= 0))) return MOY_EINVAL;
  p.plane_cols = a->plane_cols; p.plane_stride = a->plane_stride;
  p.wide_store = !is32 && !a->out_f32 && !ln && a->N % 8 == 0 && (a->ldc % 8) == 0 && aligned16(a->C) &&
                 (!a->R || ((a->ldr % 8) == 0 && aligned16(a->R)));
  if (a->ksize == 1) {
    if (a->K % kpb) return MOY_EINVAL;
  } else {
    const int C = a->Cin;
    if (C < kpb || (C % kpb) || a->K != 9 * C) return MOY_EINVAL;
    if (a->stride != 1 && a->stride != 2) return MOY_EINVAL;
    if (a->B <= 0 || a->Hin <= 0 || a->Win <= 0) return MOY_EINVAL;
    if (a->Hout != (a->Hin + 2 - 3) / a->stride + 1 || a->Wout != (a->Win + 2 - 3) / a->stride + 1) return MOY_EINVAL;
    if ((long)a->B * a->Hout * a->Wout != a->M) return MOY_EINVAL;
    p.Hin = a->Hin; p.Win = a->Win; p.Hout = a->Hout; p.Wout = a->Wout; p.Cin = C;
    p.cin_magic = (uint32_t)(((1ull << 32) + (unsigned)C - 1) / (unsigned)C);
    p.fd_hw = make_fastdiv((uint32_t)(a->Hout * a->Wout));
    p.fd_wout = make_fastdiv((uint32_t)a->Wout);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // Variants that were built and measured slower on every shape of this path (round 1, see DESIGN.md
  // section 4): an LDS-DMA ring (global_load_lds, 1-2 blocks/CU), weights-resident-in-LDS with
  // activations straight to registers (16 rows x 64 B request shape), prefetch distance 2, a persistent
  // tile loop, a direct-from-register epilogue.  They are not kept in the tree.
  if (a->post_W && (a->ksize != 3 || is32 || ln || a->post_n <= 0)) return MOY_ENOSYS;   // (never ignored)
  if (a->ksize == 1 && !is32) {
    const int rc = gemm_wreg_try(a, st);
    if (rc != MOY_ENOSYS) return rc;
  }
  if (a->run_levels) {
    // row runs outside the weight-stationary score kernel (round 6): the tiled kernel takes them for the same launch kind -- LayerNorm +
    // narrow head, rows not stored, no mask / gather / second operand -- with fp32 tensors (the folded head of the fp32 engines); a 16-bit
    // launch the weight-stationary kernel declined stays MOY_ENOSYS (its caller falls back to a_mask, as before)
    if (!is32 || !ln || a->C || a->a_mask || a->a_rows || a->A2 || a->pre || a->ksize != 1 || a->dot_n < 1) return MOY_ENOSYS;
    if (a->run_levels > 4 || a->run_period <= 0 || (a->M % a->run_period)) return MOY_EINVAL;
    if (a->run_a_period < 0 || (a->run_a_period && a->run_levels != 1)) return MOY_EINVAL;
    int nv = 0;
    for (int l = 0; l < 4; ++l) {
      const bool on = l < a->run_levels;
      if (on && (a->run_len[l] <= 0 || a->run_rows[l] <= 0 || a->run_pitch[l] < a->run_len[l] || a->run_tok0[l] < 0 ||
                 a->run_tok0[l] + (a->run_rows[l] - 1) * a->run_pitch[l] + a->run_len[l] > a->run_period))
        return MOY_EINVAL;
      p.run_cstart[l] = nv; p.run_len[l] = on ? a->run_len[l] : 1; p.run_tok0[l] = on ? a->run_tok0[l] : 0; p.run_pitch[l] = on ? a->run_pitch[l] : 1;
      p.fd_len[l] = make_fastdiv((uint32_t)p.run_len[l]);
      if (on) nv += a->run_len[l] * a->run_rows[l];
    }
    const int nb_ = a->M / a->run_period;
    p.run_levels = a->run_levels; p.run_period = a->run_period; p.run_nv = nv; p.fd_nv = make_fastdiv((uint32_t)nv);
    p.run_a_period = a->run_a_period ? a->run_a_period : a->run_period;
    p.run_a_off = a->run_a_period ? a->run_a_off : 0;
    // A rows reach b * run_a_period + token - run_a_off: the descriptor spans the frames' A rows from row 0, inside one 2 GiB window
    const int64_t a_rows_max = (int64_t)(nb_ - 1) * p.run_a_period + (a->run_period - p.run_a_off);
    p.a_bytes = ((a_rows_max > 0 ? a_rows_max : 1) * a->lda) * esz;
    if (p.a_bytes > 0x7fffffffLL) return MOY_ENOSYS;
    p.M = nb_ * nv;                          // the launch walks the COMPACT rows
  }
  if (a->ksize == 3 && !is32 && !ln) {
    const int rc = conv_ws_try(a, st);
    if (rc != MOY_ENOSYS) return rc;
  }
  if (a->post_W) return MOY_ENOSYS;          // the folded 1x1 consumer exists in the stride-2 weight-stationary kernel only: the caller launches it separately
  if (!is32 && !ln) {          // deep K, N % 128 == 0, enough tiles: 256-row tiles through an LDS-DMA pipeline (gemm_dma.hip)
    const int rc = gemm_dma_try(a, st);
    if (rc != MOY_ENOSYS) return rc;
  }
  if (a->ksize == 3 && !is32) {
    const int rc = a->dtype == MOY_BF16 ? try_conv_direct<bf16_t>(p, a->B, ln, st) : try_conv_direct<f16_t>(p, a->B, ln, st);
    if (rc != MOY_ENOSYS) return rc;
  }
  if (a->dtype == MOY_BF16)
    return a->ksize == 1 ? dispatch_tile<bf16_t, 1>(p, ln, st) : dispatch_tile<bf16_t, 3>(p, ln, st);
  if (a->dtype == MOY_F16)
    return a->ksize == 1 ? dispatch_tile<f16_t, 1>(p, ln, st) : dispatch_tile<f16_t, 3>(p, ln, st);
  if (a->dtype == MOY_F32X3)
    return a->ksize == 1 ? dispatch_tile<f32x3_t, 1>(p, ln, st) : dispatch_tile<f32x3_t, 3>(p, ln, st);
  return a->ksize == 1 ? dispatch_tile<float, 1>(p, ln, st) : dispatch_tile<float, 3>(p, ln, st);
}
