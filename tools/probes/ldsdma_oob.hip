// Probe (not product code): does `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS for lanes whose offset fails the
// descriptor's range check?  The direct convolution relies on it for the zero padding of the halo patch.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/ldsdma_oob.hip -o /tmp/ldsdma_oob && /tmp/ldsdma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(unsigned voff, const __amdgpu_buffer_rsrc_t rs, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs), "s"(soff) : "memory");
}

__global__ void probe(const unsigned char* src, int nbytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048 / 4; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;   // poison
  __syncthreads();
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, nbytes, 0x00020000);
  // even lanes read in range (lane*16, permuted), odd lanes out of range
  const unsigned voff = (lane & 1) ? 0x80000000u : (unsigned)((63 - lane) * 16);
  const unsigned base = (unsigned)reinterpret_cast<uintptr_t>(smem);
  dma16(voff, rs, 0, base);
  dma16(voff, rs, 1024, base + 1024);          // soffset moves the in-range lanes by 1 KiB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 2048 / 4; i += 64) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}

int main() {
  const int n = 4096;
  std::vector<unsigned> h(n / 4);
  for (int i = 0; i < n / 4; ++i) h[i] = 0x10000000u + i;
  unsigned char* d; unsigned* o;
  hipMalloc(&d, n); hipMalloc(&o, 2048);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 2048, 0, d, n, o);
  std::vector<unsigned> r(512);
  hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
  int bad_in = 0, zero_oob = 0, poison_oob = 0, other_oob = 0;
  for (int k = 0; k < 2; ++k)
    for (int lane = 0; lane < 64; ++lane)
      for (int w = 0; w < 4; ++w) {
        const unsigned v = r[k * 256 + lane * 4 + w];
        if (lane & 1) { if (v == 0) ++zero_oob; else if (v == 0xdeadbeefu) ++poison_oob; else ++other_oob; }
        else if (v != 0x10000000u + k * 256 + (63 - lane) * 4 + w) ++bad_in;
      }
  printf("ldsdma probe: in-range mismatches %d; out-of-range words: zero %d, untouched %d, other %d\n", bad_in, zero_oob, poison_oob, other_oob);
  return (bad_in == 0 && zero_oob == 256) ? 0 : 1;
}
