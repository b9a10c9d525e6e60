// Probe (not product code): which SIMD does wave w of a 512-thread (and 256-thread) block land on?  HW_REG_HW_ID (gfx9 layout):
// wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13].
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/wave_simd.hip -o tools/probes/wave_simd.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void who(unsigned* out) {
  extern __shared__ unsigned char smem[];
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}

int main() {
  for (int nw : {8, 4}) {
    const int blocks = 512;
    unsigned* d;
    (void)hipMalloc(&d, blocks * nw * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(who), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipLaunchKernelGGL(who, dim3(blocks), dim3(64 * nw), nw == 8 ? 98304 : 40000, 0, d);
    std::vector<unsigned> h(blocks * nw);
    (void)hipMemcpy(h.data(), d, blocks * nw * 4, hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < nw; ++w) hist[w][(h[b * nw + w] >> 4) & 3]++;
    printf("%d waves per block: SIMD histogram per wave index over %d blocks\n", nw, blocks);
    for (int w = 0; w < nw; ++w) printf("  wave %d: simd0 %4d simd1 %4d simd2 %4d simd3 %4d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("  first blocks (simd of waves 0..%d): ", nw - 1);
    for (int b = 0; b < 6; ++b) { for (int w = 0; w < nw; ++w) printf("%u", (h[b * nw + w] >> 4) & 3); printf(" "); }
    printf("\n");
    (void)hipFree(d);
  }
  return 0;
}
