// Probe: global stores by request shape (every block its own region, as kernel outputs are):
//   mode 0: 16 B per lane straight from the 16x16 accumulator layout -- lane (r = lane & 15, q = lane >> 4) writes columns 4q..4q+3 (fp32) of
//           row r, rows `pitch` bytes apart: adjacent lanes in different rows (dec_mid's offsets | weights output, pitch 1152)
//   mode 1: the same bytes, 8 adjacent lanes per row (128-B runs, 8 rows per instruction)
//   mode 2: 1 KB contiguous per instruction
//   mode 3: 8 B per lane from the accumulator layout (4 x 16-bit), rows 512 B apart (msda_raw's output rows)
// hipcc --offload-arch=gfx950 -O3 -o tools/probes/store_shapes.bin tools/probes/store_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned char* __restrict__ out, size_t region, int iters, int pitch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* base = out + (size_t)blockIdx.x * region;
  const u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
  for (int it = 0; it < iters; ++it) {
    // a "tile" = 128 rows x `pitch` bytes; wave w owns 128-byte column group w of it (mode 0/1) or a contiguous 16 KB slab (mode 2)
    unsigned char* tile = base + (size_t)(it % 4) * 128 * pitch;
#pragma unroll
    for (int i = 0; i < 8; ++i) {             // 8 x 16 rows
      if constexpr (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4*>(tile + (size_t)(i * 16 + (lane & 15)) * pitch + wave * 128 + j * 64 + (lane >> 4) * 16) = v;
      } else if constexpr (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4*>(tile + (size_t)(i * 16 + j * 8 + (lane >> 3)) * pitch + wave * 128 + (lane & 7) * 16) = v;
      } else if constexpr (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4*>(tile + (size_t)wave * 16384 + (i * 2 + j) * 1024 + lane * 16) = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x2*>(tile + (size_t)(i * 16 + (lane & 15)) * pitch + wave * 128 + j * 32 + (lane >> 4) * 8) = u32x2{v.x, v.y};
      }
    }
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200, pitch = argc > 2 ? atoi(argv[2]) : 1152;
  const size_t region = (size_t)4 * 128 * pitch + 8 * 16384 * 4;
  unsigned char* out;
  if (hipMalloc(&out, region * 256) != hipSuccess) return 1;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](int mode, const char* name) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      switch (mode) {
        case 0: k<0><<<256, 512>>>(out, region, iters, pitch); break;
        case 1: k<1><<<256, 512>>>(out, region, iters, pitch); break;
        case 2: k<2><<<256, 512>>>(out, region, iters, pitch); break;
        default: k<3><<<256, 512>>>(out, region, iters, pitch); break;
      }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 256.0 * 8 * iters * 16384;
      if (rep) printf("%-66s %7.1f GB/s per CU  %6.2f TB/s chip  (%.3f ms)\n", name, bytes / 256 / ms / 1e6, bytes / ms / 1e9, ms);
    }
  };
  printf("256 blocks x 8 waves, 16 KB per wave and step, row pitch %d B\n", pitch);
  run(0, "16 B/lane from the accumulator layout (adjacent lanes = adjacent rows)");
  run(1, "16 B/lane, 8 adjacent lanes per row (128-B runs)");
  run(2, "16 B/lane, 1 KB contiguous");
  run(3, "8 B/lane from the accumulator layout");
  return 0;
}
