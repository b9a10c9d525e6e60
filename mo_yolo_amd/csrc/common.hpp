// Shared device helpers for the gfx950 kernels of libmoyolo.so (wave64 everywhere).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/moyolo.h"

namespace moy {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct bf16_t {
  uint16_t v;
};
struct f16_t {
  uint16_t v;
};
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN preserving) on gfx950
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
// two casts at once: ONE v_cvt_pk_bf16_f32 (two scalar casts cost cvt, cvt, shift, or -- in every bf16 epilogue)
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{lo, hi}, bf16x2_));
}
__device__ __forceinline__ float bflo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

template <typename T>
struct DT;  // elements per 16-byte chunk, load/store of 4 consecutive elements as fp32
template <>
struct DT<float> {
  static constexpr int KPB = 4;
  static constexpr int code = MOY_F32;
  static __device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
  static __device__ __forceinline__ float load1(const float* p) { return *p; }
  static __device__ __forceinline__ void store1(float* p, float v) { *p = v; }
};
template <>
struct DT<double> {   // operator-API entries only (the reference's native op dispatches fp32/fp64: ms_deform_attn_cuda.cu:64)
  static constexpr int KPB = 2;
  static __device__ __forceinline__ double load1(const double* p) { return *p; }
  static __device__ __forceinline__ void store1(double* p, double v) { *p = v; }
};
// MOY_F32X3 (moy_gemm): fp32 in memory, split-fp16 matrix arithmetic -- a tag type, laid out and loaded like float
struct f32x3_t {
  float v;
};
template <>
struct DT<f32x3_t> {
  static constexpr int KPB = 4;
  static constexpr int code = MOY_F32X3;
  static __device__ __forceinline__ f32x4 load4(const f32x3_t* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store4(f32x3_t* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
  static __device__ __forceinline__ float load1(const f32x3_t* p) { return p->v; }
  static __device__ __forceinline__ void store1(f32x3_t* p, float v) { p->v = v; }
};
template <typename T>
struct is_f32 { static constexpr bool value = false; };
template <>
struct is_f32<float> { static constexpr bool value = true; };
template <>
struct is_f32<f32x3_t> { static constexpr bool value = true; };
template <typename T>
struct AccOf { typedef float type; };
template <>
struct AccOf<double> { typedef double type; };
template <>
struct DT<f16_t> {
  static constexpr int KPB = 8;
  static constexpr int code = MOY_F16;
  // The empty asm pins the fp32 value: without it hipcc may fold a preceding fp32 multiply into v_fma_mixlo_f16 (ONE rounding
  // to fp16) in one kernel and keep mul + cvt (two roundings) in another -- the same epilogue then differs by an ulp between
  // kernels (seen: SiLU epilogue, tiled vs weight-stationary GEMM).
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    asm volatile("" : "+v"(lo), "+v"(hi));
    typedef _Float16 h2_ __attribute__((ext_vector_type(2)));      // both casts at once: ONE v_cvt_pk_f16_f32 (RNE), not cvt, cvt, pack
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{lo, hi}, h2_));
  }
  static __device__ __forceinline__ float lo(uint32_t w) { return h2f((uint16_t)(w & 0xffffu)); }
  static __device__ __forceinline__ float hi(uint32_t w) { return h2f((uint16_t)(w >> 16)); }
  // acc + half(w) * g in ONE instruction (v_fma_mix_f32 reads the fp16 half itself; one rounding, as convert + v_fma_f32: the
  // conversion is exact) -- the gather kernels spent a v_cvt_f32_f16 (4.3 cycles, tools/probes/valu_rates.hip) per value on it
  static __device__ __forceinline__ float fma_lo(uint32_t w, float g, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(g), "v"(acc));
    return d;
  }
  static __device__ __forceinline__ float fma_hi(uint32_t w, float g, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(g), "v"(acc));
    return d;
  }
  // the same with a wave-uniform weight in an SGPR (msda_raw.hip: weights come from v_readlane)
  static __device__ __forceinline__ float fma_lo_s(uint32_t w, float g, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "s"(g), "v"(acc));
    return d;
  }
  static __device__ __forceinline__ float fma_hi_s(uint32_t w, float g, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "s"(g), "v"(acc));
    return d;
  }
  static __device__ __forceinline__ f32x4 load4(const f16_t* p) {
    u32x2 w = *reinterpret_cast<const u32x2*>(p);
    return f32x4{lo(w.x), hi(w.x), lo(w.y), hi(w.y)};
  }
  static __device__ __forceinline__ void store4(f16_t* p, f32x4 v) {
    *reinterpret_cast<u32x2*>(p) = u32x2{pack2(v.x, v.y), pack2(v.z, v.w)};
  }
  static __device__ __forceinline__ float load1(const f16_t* p) { return h2f(p->v); }
  static __device__ __forceinline__ void store1(f16_t* p, float v) { p->v = f2h(v); }
};
template <>
struct DT<bf16_t> {
  static constexpr int KPB = 8;
  static constexpr int code = MOY_BF16;
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf2(lo, hi); }
  static __device__ __forceinline__ float lo(uint32_t w) { return bflo(w); }
  static __device__ __forceinline__ float fma_lo(uint32_t w, float g, float acc) { return __builtin_fmaf(bflo(w), g, acc); }
  static __device__ __forceinline__ float fma_hi(uint32_t w, float g, float acc) { return __builtin_fmaf(bfhi(w), g, acc); }
  static __device__ __forceinline__ float fma_lo_s(uint32_t w, float g, float acc) { return __builtin_fmaf(bflo(w), g, acc); }
  static __device__ __forceinline__ float fma_hi_s(uint32_t w, float g, float acc) { return __builtin_fmaf(bfhi(w), g, acc); }
  static __device__ __forceinline__ float hi(uint32_t w) { return bfhi(w); }
  static __device__ __forceinline__ f32x4 load4(const bf16_t* p) {
    u32x2 w = *reinterpret_cast<const u32x2*>(p);
    return f32x4{bflo(w.x), bfhi(w.x), bflo(w.y), bfhi(w.y)};
  }
  static __device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
    *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
  }
  static __device__ __forceinline__ float load1(const bf16_t* p) { return bf2f(p->v); }
  static __device__ __forceinline__ void store1(bf16_t* p, float v) { p->v = f2bf(v); }
};

// Wave-wide reductions on the VALU (DPP), result broadcast through an SGPR.  `__shfl_xor` compiles to ds_bpermute_b32: six
// dependent trips through the LDS crossbar per reduction, which made the LayerNorm epilogue (3 reductions per row) LDS-bound.
// quad_perm [1,0,3,2] / [2,3,0,1] -> quad sums; row_shr:4, row_shr:8 -> the row total in lanes 12..15 of each row of 16;
// row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3) -> the wave total in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xb1, 0xf>(0.f, v);
  v += dpp_f<0x4e, 0xf>(0.f, v);
  v += dpp_f<0x114, 0xf>(0.f, v);
  v += dpp_f<0x118, 0xf>(0.f, v);
  v += dpp_f<0x142, 0xa>(0.f, v);
  v += dpp_f<0x143, 0xc>(0.f, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xb1, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x4e, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x114, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x118, 0xf>(v, v));
  v = fmaxf(v, dpp_f<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_f<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// accurate forms (feed thresholds / final outputs)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// epilogue forms: v_exp_f32 + v_rcp_f32 (about 1 ulp each, 5 VALU ops instead of ~40): the GEMM
// epilogues were VALU-bound on the libm expf + IEEE divide (profiles/r01_b_pmc_gemm.txt)
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float siluf_(float x) { return x * fast_sigmoid(x); }

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case MOY_ACT_SILU: return siluf_(v);
    case MOY_ACT_RELU: return fmaxf(v, 0.0f);
    case MOY_ACT_SIGMOID: return fast_sigmoid(v);
    default: return v;
  }
}

// Division by a launch-time constant: q = (n * M) >> (32 + s) with M = ceil(2^(32+s) / d), s = ceil(log2 d), exact for
// every 32-bit n (Granlund & Montgomery); M = 2^32 + magic.  A 32-bit divide by a runtime value costs ~35 VALU ops and
// the tile set-up (pixel coordinates of every staged row) did five of them before the first load could issue.
struct FastDiv {
  uint32_t magic, shift, d;
};
static FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{0u, 0u, d ? d : 1u};
  d = f.d;
  while ((1ull << f.shift) < d) ++f.shift;
  f.magic = (uint32_t)((((1ull << f.shift) - d) << 32) / d + 1);   // ceil(2^(32+s)/d) - 2^32
  if (d == 1) f.magic = 0;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return (uint32_t)(((uint64_t)__umulhi(n, f.magic) + n) >> f.shift);
}

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MOY_OK : MOY_ELAUNCH;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// moy_gemm_query (round 5): the dispatch of moy_gemm WITHOUT the launch.  While the calling thread's slot is non-zero every launch
// site of moy_gemm's kernels records its kernel family there and returns MOY_OK instead of launching -- the validation and the
// eligibility rules are the ones a real call runs, so a planner can ask "which kernel, if any" on the host.
int& plan_only_slot();
inline bool plan_only(int kernel) {
  int& s = plan_only_slot();
  if (!s) return false;
  s = kernel;
  return true;
}

// gemm_wreg.hip: weight-stationary kernel for K == 256 (MOY_ENOSYS when the shape is not its own)
int gemm_wreg_try(const moy_gemm_args* a, hipStream_t st);
// gemm_dma.hip: 256-row tiles, both operands by LDS-DMA, for the matrix-rate-bound products (MOY_ENOSYS when the shape is not its own)
int gemm_dma_try(const moy_gemm_args* a, hipStream_t st);
// conv_ws.hip: persistent weight-stationary direct 3x3 convolution (MOY_ENOSYS when the shape is not its own)
int conv_ws_try(const moy_gemm_args* a, hipStream_t st);

// ---- Build kinds (round 6).  The PRODUCT library (libmoyolo.so, MOY_DIAG == 0) reads no environment variable and holds no timing-only
// kernel: `knob()` is a compile-time constant there, every A/B dispatch folds to the shipped choice and the lab-only template instances
// sit behind `#if MOY_DIAG`.  The LAB library (libmoyolo_diag.so: `python -m mo_yolo_amd.build --diag`, selected by MOYOLO_LIB) is the
// same sources with -DMOY_DIAG=1: `knob("MOY_X", d)` reads the variable ONCE at the call site's first use (callers keep it in a
// function-local static), the timing-only / stamped instantiations exist.  A host sees no hidden state through include/moyolo.h: the
// one real option of the library is moy_set_cu_limit().
#ifndef MOY_DIAG
#define MOY_DIAG 0
#endif
#if MOY_DIAG
inline int knob(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// Timing-only / diagnostic modes (MOY_*_ABL, MOY_*_DIAG) produce GARBAGE results by design: they exist for tools/bench_gemm.py
// and tools/probes/.  A variable left set would corrupt outputs with rc = 0, so the first use says so loudly on stderr.
inline int garbage_mode_env(const char* name) {
  const int v = knob(name, 0);
  if (v)
    fprintf(stderr, "\n*** libmoyolo_diag: %s=%d selects a TIMING-ONLY / DIAGNOSTIC kernel build: its RESULTS ARE GARBAGE by design. "
                    "Unset it for any run whose outputs matter. ***\n\n", name, v);
  return v;
}
#else
constexpr int knob(const char*, int dflt) { return dflt; }
constexpr int garbage_mode_env(const char*) { return 0; }
#endif

// moy_set_cu_limit(n) (per host thread; the lab library also reads MOY_CU_LIMIT=<n> at every launch): the persistent kernels (gemm_wreg,
// conv_ws) size their grids for n compute units instead of the device's.  Results do not depend on it (the row-tile walk is the
// same); used by the engine's forked value projection and by the CU-partition measurements of tools/probes/cu_share.py.
int& cu_limit_slot();
inline int cu_limit(int n) {
  int v = cu_limit_slot();
  if (v <= 0) v = knob("MOY_CU_LIMIT", 0);
  v = v / 8 * 8;
  if (v >= 8 && v < n) n = v;
  return n;
}

}  // namespace moy
