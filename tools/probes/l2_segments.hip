// Probe: L2-resident weight streaming, every CU reading the SAME table (as the decoder tail's weights), by request shape:
//   mode 0: 16 B per lane, 4 lanes contiguous (64-B segments in 16 rows of 512 B)   -- the MFMA A-operand pattern of dec_tail / dec_mid
//   mode 1: 16 B per lane, 8 lanes contiguous (128-B segments in 8 rows)
//   mode 2: 16 B per lane, 64 lanes contiguous (1 KB)                                -- a fragment-ordered (pre-packed) weight image
//   mode 3: 4 B per lane, 16 lanes contiguous (64-B segments, 4 per instruction)     -- the tap loads of msda_raw_mfma_kernel
//   mode 4: 8 B per lane, 16 lanes contiguous (128-B segments, 4 per instruction)
//   mode 5 / 6: 16 B per lane, 256-B segments x 4 rows / 512-B rows x 2 (lanes of a row NOT adjacent: lane & 3 / lane & 1 picks the row)
// hipcc --offload-arch=gfx950 -O3 -o tools/probes/l2_segments.bin tools/probes/l2_segments.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned char* __restrict__ tab, int tab_bytes, int iters, uint32_t* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t acc = 0;
  // every wave walks the table in 16 KB slices (its "weight rows"): slice = (it * 8 + wave) mod nslices
  const int nsl = tab_bytes / 16384;
  for (int it = 0; it < iters; ++it) {
    const unsigned char* sl = tab + (size_t)((it * 8 + wave) % nsl) * 16384;
    if constexpr (MODE <= 2 || MODE >= 5) {
      u32x4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int off;
        if (MODE == 0) off = ((i >> 3) * 16 + (lane & 15)) * 512 + (i & 7) * 64 + (lane >> 4) * 16;          // rows of 512 B, 64-B pieces
        else if (MODE == 1) off = ((i >> 2) * 8 + (lane & 7)) * 512 + (i & 3) * 128 + (lane >> 3) * 16;     // 128-B pieces
        else if (MODE == 5) off = ((i >> 1) * 4 + (lane & 3)) * 512 + (i & 1) * 256 + (lane >> 2) * 16;      // 256-B pieces x 4 rows
        else if (MODE == 6) off = (i * 2 + (lane & 1)) * 512 + (lane >> 1) * 16;                            // 512-B rows x 2
        else if (MODE == 7) off = i * 1024 + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);      // 1 KB contiguous, 16-B chunks XOR-permuted inside each 128-B line
        else if (MODE == 8) off = i * 1024 + (lane >> 4) * 256 + (((lane & 15) ^ ((i * 4 + (lane >> 4)) & 15)) * 16);   // ... inside each 256-B row
        else off = i * 1024 + lane * 16;
        v[i] = *reinterpret_cast<const u32x4*>(sl + off);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    } else if constexpr (MODE == 3) {
      uint32_t v[64];
#pragma unroll
      for (int i = 0; i < 64; ++i) v[i] = *reinterpret_cast<const uint32_t*>(sl + ((i * 4 + (lane >> 4)) * 64 + (lane & 15) * 4));
#pragma unroll
      for (int i = 0; i < 64; ++i) acc += v[i];
    } else {
      u32x2 v[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = *reinterpret_cast<const u32x2*>(sl + ((i * 4 + (lane >> 4)) * 128 + (lane & 15) * 8));
#pragma unroll
      for (int i = 0; i < 32; ++i) acc += v[i].x ^ v[i].y;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int tab_bytes = (argc > 1 ? atoi(argv[1]) : 1024) * 1024, iters = argc > 2 ? atoi(argv[2]) : 400;
  unsigned char* tab; uint32_t* sink;
  hipMalloc(&tab, tab_bytes); hipMalloc(&sink, 4);
  hipMemset(tab, 1, tab_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int mode, const char* name) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: k<0><<<256, 512>>>(tab, tab_bytes, iters, sink); break;
        case 1: k<1><<<256, 512>>>(tab, tab_bytes, iters, sink); break;
        case 2: k<2><<<256, 512>>>(tab, tab_bytes, iters, sink); break;
        case 3: k<3><<<256, 512>>>(tab, tab_bytes, iters, sink); break;
        case 4: k<4><<<256, 512>>>(tab, tab_bytes, iters, sink); break;
        case 5: k<5><<<256, 512>>>(tab, tab_bytes, iters, sink); break;
        case 6: k<6><<<256, 512>>>(tab, tab_bytes, iters, sink); break;
        case 7: k<7><<<256, 512>>>(tab, tab_bytes, iters, sink); break;
        default: k<8><<<256, 512>>>(tab, tab_bytes, iters, sink); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 256.0 * 8 * iters * 16384;
      if (rep) printf("%-58s %7.1f GB/s per CU  %6.2f TB/s chip  (%.3f ms)\n", name, bytes / 256 / ms / 1e6, bytes / ms / 1e9, ms);
    }
  };
  printf("table %d KB shared by 256 blocks x 8 waves, 16 KB per wave and step in flight\n", tab_bytes / 1024);
  run(0, "16 B/lane, 64-B segments x 16 rows (MFMA A operand)");
  run(1, "16 B/lane, 128-B segments x 8 rows");
  run(2, "16 B/lane, 1 KB contiguous (fragment-ordered image)");
  run(3, "4 B/lane, 4 x 64-B segments (tap dwords)");
  run(4, "8 B/lane, 4 x 128-B segments");
  run(5, "16 B/lane, 256-B segments x 4 rows");
  run(6, "16 B/lane, 512-B rows x 2");
  run(7, "16 B/lane, 1 KB contiguous, chunks XOR-permuted per 128-B line");
  run(8, "16 B/lane, 1 KB contiguous, chunks XOR-permuted per 256-B row");
  return 0;
}
