// Probe (not product code): what v_mfma_f32_16x16x32_bf16 rate does this device SUSTAIN, and how does it depend on the number of
// resident waves per SIMD and on the number of independent accumulators a wave cycles through?  No memory traffic at all.
// Also: the same loop with an LDS fragment read (ds_read_b128, conflict-free) in front of every MFMA pair, as the weight-stationary
// GEMM issues them (one operand from registers, the other from LDS).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o tools/probes/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;

template <int NACC, int LDS>
__global__ __launch_bounds__(512, 1) void spin(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u};
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (LDS) {
    for (int i = threadIdx.x; i < 16384 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u;
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (LDS && (i & 3) == 0) b = *reinterpret_cast<const u32x4*>(smem + ((it * NACC + i) * 1024 & 16383) + lane * 16);   // one read per FOUR MFMAs (NT = 4)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
    }
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] == 12345.678f) out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC, int LDS>
static void run(int waves_per_cu, const char* name) {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * 8;
  auto k = spin<NACC, LDS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves_per_cu), 98304, 0, out, iters);   // 96 KB of LDS: ONE block per CU
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * waves_per_cu * iters * NACC * 16 * 16 * 32 * 2;
  // cycles per MFMA per SIMD at 2.4 GHz: a SIMD executes blocks/256 * waves/4 wave-loops one after another (or interleaved)
  const double mf_per_simd = (double)blocks / 256 * waves_per_cu / 4 * iters * NACC;
  printf("%-34s waves/CU %2d  %8.3f ms  %7.1f TFLOP/s  %5.1f ns/MFMA/SIMD = %5.1f cycles at 2.4 GHz\n", name, waves_per_cu, ms, flops / ms * 1e-9,
         ms * 1e6 / mf_per_simd, ms * 1e6 / mf_per_simd * 2.4);
  hipFree(out);
}

int main() {
  for (int w : {4, 8}) {
    run<1, 0>(w, "1 accumulator (dependent chain)");
    run<2, 0>(w, "2 accumulators");
    run<4, 0>(w, "4 accumulators");
    run<8, 0>(w, "8 accumulators");
    run<8, 1>(w, "8 accumulators + ds_read_b128 / 4");
  }
  // long run: do the clocks hold?  (~2 s of back-to-back MFMAs)
  for (int rep = 0; rep < 3; ++rep) run<8, 0>(8, "8 accumulators (repeat)");
  return 0;
}
