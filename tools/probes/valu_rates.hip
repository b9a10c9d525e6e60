// Probe (not product code): issue cost of the vector instructions the epilogues are made of, in cycles per wave64 instruction and
// SIMD, with 1 and 2 resident waves per SIMD: v_exp_f32, v_rcp_f32, v_fma_f32, v_pk_fma_f32, v_pk_fma_f16, v_pk_mul_f32,
// v_cvt_pk_bf16_f32, v_cvt_pkrtz_f16_f32, v_exp_f16, v_rcp_f16.  16 independent registers per instruction kind, 64 back to back.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rates.hip -o tools/probes/valu_rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(512, 1) void spin(float* out, int iters, float seed) {
  extern __shared__ unsigned char smem[];
  float v[16];
  float w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = seed + threadIdx.x * 1e-3f + i; w[i] = v[i] * 0.5f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        if (KIND == 0) { asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(v[i]), "+v"(v[i + 1])); }
        if (KIND == 1) { asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1" : "+v"(v[i]), "+v"(v[i + 1])); }
        if (KIND == 2) { asm volatile("v_fma_f32 %0, %0, %2, %2\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(v[i]), "+v"(v[i + 1]) : "v"(w[0])); }
        if (KIND == 3) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&v[i])) : "v"(*reinterpret_cast<double*>(&w[i]))); }
        if (KIND == 4) { asm volatile("v_pk_fma_f16 %0, %0, %2, %2\n\tv_pk_fma_f16 %1, %1, %2, %2" : "+v"(v[i]), "+v"(v[i + 1]) : "v"(w[0])); }
        if (KIND == 5) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[i])) : "v"(*reinterpret_cast<double*>(&w[i]))); }
        if (KIND == 6) { asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %1, %1, %0" : "+v"(v[i]), "+v"(v[i + 1])); }
        if (KIND == 7) { asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n\tv_cvt_pkrtz_f16_f32 %1, %1, %0" : "+v"(v[i]), "+v"(v[i + 1])); }
        if (KIND == 8) { asm volatile("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1" : "+v"(v[i]), "+v"(v[i + 1])); }
        if (KIND == 9) { asm volatile("v_rcp_f16 %0, %0\n\tv_rcp_f16 %1, %1" : "+v"(v[i]), "+v"(v[i + 1])); }
        if (KIND == 10) { asm volatile("v_cvt_f32_f16 %0, %0\n\tv_cvt_f32_f16 %1, %1" : "+v"(v[i]), "+v"(v[i + 1])); }
        if (KIND == 11) { asm volatile("v_pk_add_f16 %0, %0, %2\n\tv_pk_mul_f16 %1, %1, %2" : "+v"(v[i]), "+v"(v[i + 1]) : "v"(w[0])); }
        if (KIND == 12) { asm volatile("v_add_f32 %0, 1.0, %0\n\tv_mul_f32 %1, %1, %2" : "+v"(v[i]), "+v"(v[i + 1]) : "v"(w[0])); }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, int per_pair) {
  float* out;
  (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto k = spin<KIND>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  for (int waves : {4, 8}) {
    const int iters = 4000, blocks = 256 * 4;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 98304, 0, out, iters, 1.0f);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr_per_simd = (double)blocks / 256 * waves / 4 * iters * 4 * 8 * per_pair;
    printf("%-22s waves/CU %d  %7.3f ms  %5.2f ns per instruction and SIMD = %5.1f cycles at 2.4 GHz\n", name, waves, ms, ms * 1e6 / instr_per_simd,
           ms * 1e6 / instr_per_simd * 2.4);
  }
  (void)hipFree(out);
}

int main() {
  run<2>("v_fma_f32", 2);
  run<12>("v_add_f32 / v_mul_f32", 2);
  run<3>("v_pk_fma_f32", 1);
  run<5>("v_pk_mul_f32", 1);
  run<4>("v_pk_fma_f16", 2);
  run<11>("v_pk_add/mul_f16", 2);
  run<0>("v_exp_f32", 2);
  run<1>("v_rcp_f32", 2);
  run<8>("v_exp_f16", 2);
  run<9>("v_rcp_f16", 2);
  run<6>("v_cvt_pk_bf16_f32", 2);
  run<7>("v_cvt_pkrtz_f16_f32", 2);
  run<10>("v_cvt_f32_f16", 2);
  return 0;
}
