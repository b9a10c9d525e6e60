// Fused preprocess + stem + first down-sampling conv:  uint8 BGR frames -> SiLU(BN(conv3x3 s2, 3 -> 32)) -> SiLU(BN(conv3x3 s2, 32 -> 64))
// (predictor.py:117-134 preprocess; layers 0 and 1 of yolo_track.yaml:17-18 over conv.py:36-38) in ONE persistent kernel.
//
// Why: as two launches the pair was 10 % of the step (stem 1.33 ms, 0.34 of its floor, VALU-bound on per-byte gathers; layer 1
// 1.02-1.41 ms) and moved the 32-channel half-resolution tensor -- the largest activation of the network, 10.6 MB per frame --
// to HBM and back (6.1 GB per 288 frames).  Fused, that tensor only ever exists as the halo patch of one output tile in LDS:
//   * a block owns TH x 16 output pixels of layer 1; their (2 TH + 1) x 33 stem pixels need a (4 TH + 3) x 67 pixel window of
//     the frame: 7 KB of uint8, brought in by aligned dword loads one tile ahead (registers), borders by byte;
//   * stem on the matrix cores: K = 27 padded to 32, with the k order chosen so that a lane's 8-element slice is 8 CONSECUTIVE
//     BYTES of one window row (tap row ky = q, bytes (kx, c_bgr) 0..7; slice 3 = the ninth byte of the three rows):
//     3 dword LDS reads + 2 v_alignbyte per fragment instead of 8 byte gathers.  The bytes enter the MFMA as the IEEE halfs
//     1024 + x -- bit pattern 0x6400 | x, exact, made by ONE v_perm_b32 per pair of bytes (no integer -> float conversion at all);
//     the stem's weights are halfs for both engine types and the constant 1024 * sum_k w[n][k] (summed once per block from the
//     weight fragments) is taken out again in the fp32 BN shift; the 1/255 of the preprocess is folded into the fp32 BN scale
//     (closer to the fp32 reference than bf16(u8/255));
//   * BN + SiLU, zero outside the stem image (layer 1's padding), straight into layer 1's patch image: parity-de-interleaved
//     rows, XOR-swizzled chunks, exactly the layout conv_s2_kernel DMA's from HBM;
//   * layer 1 = conv_s2_kernel's core (weights in registers, row reuse), BN + SiLU, tile through LDS, whole-line stores.
// 60 KB of LDS and < 128 VGPRs: two blocks per CU, so one block's VALU-heavy stem phase runs beside the other's MFMA / store phase.
#include "common.hpp"

#include <type_traits>

namespace moy {

struct StemL1Params {
  const uint8_t* in;           // [B, H, W, 3] BGR
  int B, H, W;                 // frame size; stem output Hs x Ws = H/2 x W/2, layer-1 output Ho x Wo = H/4 x W/4
  const void* w0;              // IEEE half [32][32] (both engine types): stem weights, k order of this kernel (host: ops.stem_weights_fused)
  const float* sc0; const float* sh0;
  const void* w1; int Kpad1;   // T [64][Kpad1], k = (ky*3+kx)*32 + c
  const float* sc1; const float* sh1;
  void* out; int64_t ldc;
  int tiles_x, tiles_img, ntiles, per_xcd, bpx;
  FastDiv fd_timg, fd_tx;
};

template <typename T>
__device__ __forceinline__ f32x4 sl1_mfma(f32x4 acc, u32x4 w, u32x4 a);
template <>
__device__ __forceinline__ f32x4 sl1_mfma<bf16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 sl1_mfma<f16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}

constexpr int SL1_TH = 8;

template <typename T, int DIAG = 0>     // DIAG = 1: s_memtime stamps of wave 0 (diagnostic build, MOY_SL1_DIAG=1; output garbage at the head of `out`)
                                        // DIAG = 2 / 3 / 4: stamps + timing-only ablations of the stem phase (results garbage): 2 = no SiLU (BN only),
                                        // 3 = no fragment build (the window dwords go to the MFMA as they are), 4 = both
__global__ __launch_bounds__(512, 4) void stem_l1_kernel(const StemL1Params p) {
  constexpr int TH = SL1_TH, C1 = 32, N1 = 64;
  constexpr int PW = 33, PH = 2 * TH + 1, NPIX = PH * PW;              // layer-1 patch of stem pixels
  constexpr int NFRAG = (NPIX + 15) / 16, FPW = (NFRAG + 7) / 8;       // stem MFMA fragments, per wave
  constexpr int WR = 4 * TH + 3, ROW_DW = 52, ROWB = ROW_DW * 4;        // uint8 window: rows, dwords per row (67 px * 3 B + 3 B phase)
  constexpr int NLD = (WR * ROW_DW + 511) / 512;
  constexpr int WIN_B = ((WR * ROWB + 15) / 16) * 16, PATCH_B = ((NPIX * C1 * 2 + 1023) / 1024) * 1024;
  constexpr int WN = 4, WM = 2, MT = TH / WM, NCP = N1 / 8, NPASS = TH * 16 * NCP / 512;
  constexpr int PHASE = 3;                                              // byte phase of a window row inside its first dword (W % 4 == 0)
  static_assert(MT == 4 && NPASS == 2, "tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* win = smem;
  unsigned char* patch = smem + WIN_B;
  unsigned char* stg = patch + PATCH_B;
  float* wsum = reinterpret_cast<float*>(stg + TH * 16 * N1 * 2);   // [32]: sum_k w0[n][k] (the bias the 1024 + x encoding adds per unit of weight)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int wn = wave % WN, wm = wave / WN;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t_first = xcd * p.per_xcd + slot;
  const int t_limit = min((xcd + 1) * p.per_xcd, p.ntiles);
  if (t_first >= t_limit) return;
  const int n_mine = (t_limit - t_first + p.bpx - 1) / p.bpx;
  const int Hs = p.H >> 1, Ws = p.W >> 1, Ho = p.H >> 2, Wo = p.W >> 2;

  struct Tile { int b, y0, x0; };
  auto tile_of = [&](int it) {
    Tile t;
    const int id = min(t_first + it * p.bpx, p.ntiles - 1);
    t.b = (int)fdiv((uint32_t)id, p.fd_timg);
    const int rem = id - t.b * p.tiles_img;
    const int ty = (int)fdiv((uint32_t)rem, p.fd_tx);
    t.y0 = ty * TH;
    t.x0 = (rem - ty * p.tiles_x) * 16;
    return t;
  };
  // window of tile t -> registers: dword i = tid + k*512 of the [WR][ROW_DW] image.  Round 5: one buffer load per dword, the offset a
  // per-thread constant (window row, dword) + three tile terms -- the first version did this in 64-bit arithmetic with a byte-wise
  // branch per dword, 231 vector instructions per tile and thread: 22 % of the kernel's vector work for 7 KB of input.
  //   byte (4 x0 - 3) * 3 = 12 x0 - 9 of an image row is where the window row starts; its aligned dword is 12 x0 - 12 (PHASE = 3;
  //   image rows are dword aligned: W % 4 == 0).  Rows above / below the image: out-of-range offset (zeros = the stem's padding).
  //   Left of the image (x0 == 0, dwords 0-2): the bytes in front of the row belong to the previous row, never to the padding: zeroed.
  //   Right of the image: only stem pixels outside the stem image read them, and those are zeroed when the patch is written.
  uint32_t wreg[NLD];
  int w_rr[NLD], w_off[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = tid + k * 512;
    w_rr[k] = i / ROW_DW;
    const int dw = i - w_rr[k] * ROW_DW;
    w_off[k] = w_rr[k] < WR ? w_rr[k] * p.W * 3 + dw * 4 - 12 : -1;     // -1: past the window
    if (dw < 3) w_rr[k] |= 0x10000;                                      // bit 16: in front of the row when x0 == 0
  }
  const uint32_t img_bytes = (uint32_t)p.H * p.W * 3;
  auto load_window = [&](const Tile& t) {
    const auto rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.in + (int64_t)t.b * img_bytes), 0, img_bytes, 0x00020000);
    const int tile_off = ((4 * t.y0 - 3) * p.W + 4 * t.x0) * 3;          // byte offset of window row 0, image column 4 x0
    const int iy0 = 4 * t.y0 - 3;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int iy = iy0 + (w_rr[k] & 0xffff);
      const bool ok = w_off[k] != -1 && (unsigned)iy < (unsigned)p.H && !(t.x0 == 0 && (w_rr[k] & 0x10000));
      wreg[k] = __builtin_amdgcn_raw_buffer_load_b32(rsI, ok ? (uint32_t)(tile_off + w_off[k]) : 0x80000000u, 0, 0);
    }
  };
  auto store_window = [&]() {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 512;
      if (i < WR * ROW_DW) reinterpret_cast<uint32_t*>(win)[i] = wreg[k];
    }
  };

  // ---- weights -> registers
  u32x4 w0f[2];
  {
    const f16_t* W0 = static_cast<const f16_t*>(p.w0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      w0f[j] = *reinterpret_cast<const u32x4*>(W0 + (j * 16 + r) * 32 + q * 8);
      float sm = 0.f;
      const uint32_t ww[4] = {w0f[j].x, w0f[j].y, w0f[j].z, w0f[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) sm += DT<f16_t>::lo(ww[e]) + DT<f16_t>::hi(ww[e]);
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      if (wave == 0 && q == 0) wsum[j * 16 + r] = sm;
    }
  }
  u32x4 w1f[9];
  f32x4 sc1, sh1;
  {
    const T* W1 = static_cast<const T*>(p.w1);
    const int n = wn * 16;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w1f[tap] = *reinterpret_cast<const u32x4*>(W1 + (int64_t)(n + r) * p.Kpad1 + tap * C1 + q * 8);
    sc1 = *reinterpret_cast<const f32x4*>(p.sc1 + n + q * 4);
    sh1 = *reinterpret_cast<const f32x4*>(p.sh1 + n + q * 4);
  }
  T* __restrict__ Og = static_cast<T*>(p.out);
  const int64_t img_c = (int64_t)Ho * Wo * p.ldc;
  const int s_px = tid / NCP, s_c = tid % NCP;
  const int s_ty = s_px >> 4, s_tx = s_px & 15;
  const int s_lds = s_px * (N1 * 2) + ((s_c ^ (s_px & (NCP - 1))) * 16);
  const int s_rel = ((s_ty * Wo + s_tx) * (int)p.ldc + s_c * 8) * 2;
  constexpr int PASS_ROWS = 512 / NCP / 16, PASS_LDS = (512 / NCP) * N1 * 2;
  const int s_pass_rel = PASS_ROWS * Wo * (int)p.ldc * 2;
  constexpr uint32_t OOB = 0x80000000u;

  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  auto stamp = [&](int i) {
    if constexpr (DIAG) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long tt = __builtin_amdgcn_s_memtime();
      ph[i] += tt - tprev;
      tprev = tt;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  load_window(tile_of(0));
  const int pix0 = 2 * wm * MT * PW + r;
  if constexpr (DIAG) tprev = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n_mine; ++it) {
    const Tile t = tile_of(it);
    store_window();
    stamp(0);
    __syncthreads();                                   // W: window of tile `it` visible (and the staging of tile it-1 read out)
    stamp(1);
    if (it + 1 < n_mine) load_window(tile_of(it + 1)); // in flight under the whole tile

    // ---- stem: fragments of 16 consecutive patch pixels, K = 32 (27), 32 channels
    f32x4 sc0[2], sh0[2];                               // (re-read per tile from the cache: 16 registers the layer-1 phase needs)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      sc0[j] = *reinterpret_cast<const f32x4*>(p.sc0 + j * 16 + q * 4) * (1.0f / 255.0f);     // the preprocess' /255 (predictor.py:133)
      sh0[j] = *reinterpret_cast<const f32x4*>(p.sh0 + j * 16 + q * 4) - *reinterpret_cast<const f32x4*>(wsum + j * 16 + q * 4) * sc0[j] * 1024.0f;
    }
    int pix = wave * 16 + r, row = pix / PW, s33 = pix - row * PW;
#pragma unroll 1
    for (int k = 0; k < FPW; ++k, pix += 128, row += 128 / PW, s33 += 128 % PW) {   // (rolled: not worth 5x the registers)
      if (wave + 8 * k >= NFRAG) break;                 // wave-uniform
      if (s33 >= PW) { s33 -= PW; ++row; }              // (row, s33) = divmod(pix, PW), carried
      const int col = s33 < 17 ? 2 * s33 : 2 * (s33 - 17) + 1;
      // window coordinates of the stem pixel's 3x3 input patch: rows 2 row .. +2, bytes 6 col .. +8 (+ the row phase)
      const int rsel = min(2 * row + min(q, 2), WR - 1);
      // (24-bit multiplies: v_mul_lo_u32 / v_mad_u64_u32 issue at a quarter of the rate, and this loop is bound by vector issue)
      const int c6 = __mul24(col, 6);
      const int p0 = __mul24(rsel, ROWB) + PHASE + c6;
      const uint32_t* wp = reinterpret_cast<const uint32_t*>(win + (p0 & ~3));
      const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2];
      const int sh = p0 & 3;
      uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
      if (q == 3) {                                     // slice 3: the ninth byte of the three rows, then zeros
        const int pb = __mul24(min(2 * row, WR - 3), ROWB) + PHASE + c6 + 8;
        lo = (uint32_t)win[pb] | ((uint32_t)win[pb + ROWB] << 8) | ((uint32_t)win[pb + 2 * ROWB] << 16);
        hi = 0;
      }
      // bytes -> the halfs 1024 + x: [b, 0x64] per 16-bit slot, one v_perm_b32 per pair (selector byte 4 = a byte of the constant)
      u32x4 af = {__builtin_amdgcn_perm(0x64646464u, lo, 0x04010400u), __builtin_amdgcn_perm(0x64646464u, lo, 0x04030402u),
                  __builtin_amdgcn_perm(0x64646464u, hi, 0x04010400u), __builtin_amdgcn_perm(0x64646464u, hi, 0x04030402u)};
      if constexpr (DIAG == 3 || DIAG == 4) af = u32x4{d0, d1, d2, d0};      // ablation: one aligned read of three dwords, nothing built
      const int sy = 2 * t.y0 - 1 + row, sx = 2 * t.x0 - 1 + col;
      const bool inside = pix < NPIX && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;   // outside: layer 1's zero padding
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 v = sl1_mfma<f16_t>(f32x4{0.f, 0.f, 0.f, 0.f}, w0f[j], af) * sc0[j] + sh0[j];
        if constexpr (DIAG != 2 && DIAG != 4) { v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w); }
        u32x2 o = {DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
        if (!inside) o = u32x2{0u, 0u};
        if (pix < NPIX)
          *reinterpret_cast<u32x2*>(patch + pix * (C1 * 2) + (((2 * j + (q >> 1)) ^ ((pix >> 1) & 3)) * 16) + (q & 1) * 8) = o;
      }
    }
    stamp(2);
    __syncthreads();                                   // P: patch complete
    stamp(3);

    // ---- layer 1 on the patch (conv_s2_kernel's core: C 32, N 64, rows 2y+ky of the de-interleaved image)
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int pixv = pix0;
    asm volatile("" : "+v"(pixv));
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      constexpr int KOFF[3] = {0, 17, 1};
      constexpr int RG = 2;                              // output rows per fragment-row group (2 RG + 1 fragments live)
#pragma unroll
      for (int g = 0; g < MT / RG; ++g) {
        u32x4 a[2 * RG + 1];
#pragma unroll
        for (int y = 0; y < 2 * RG + 1; ++y) {
          const int pix = pixv + (2 * g * RG + y) * PW + KOFF[kx];
          a[y] = *reinterpret_cast<const u32x4*>(patch + pix * (C1 * 2) + ((q ^ ((pix >> 1) & 3)) * 16));
        }
#pragma unroll
        for (int y = 0; y < RG; ++y)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) acc[g * RG + y] = sl1_mfma<T>(acc[g * RG + y], w1f[ky * 3 + kx], a[2 * y + ky]);
      }
    }
    {
      const int ch = wn * 16 + q * 4;
#pragma unroll
      for (int y = 0; y < MT; ++y) {
        f32x4 v = acc[y] * sc1 + sh1;
        v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w);
        const int opx = (wm * MT + y) * 16 + r;
        *reinterpret_cast<u32x2*>(stg + opx * (N1 * 2) + ((((ch >> 3) ^ (opx & (NCP - 1))) * 16) + ((ch >> 2) & 1) * 8)) =
            u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
      }
    }
    stamp(4);
    __syncthreads();                                   // S: staging complete, patch free
    stamp(5);
    {
      const auto rsC = __builtin_amdgcn_make_buffer_rsrc(Og + (int64_t)t.b * img_c, 0, (uint32_t)(img_c * 2), 0x00020000);
      const int off_c = (t.y0 * Wo + t.x0) * (int)p.ldc * 2 + s_rel;
      const bool xok = t.x0 + s_tx < Wo;
      u32x4 vv[NPASS];
#pragma unroll
      for (int k = 0; k < NPASS; ++k) vv[k] = *reinterpret_cast<const u32x4*>(stg + s_lds + k * PASS_LDS);
#pragma unroll
      for (int k = 0; k < NPASS; ++k) {
        const bool ok = xok && t.y0 + s_ty + k * PASS_ROWS < Ho;
        __builtin_amdgcn_raw_buffer_store_b128(vv[k], rsC, ok ? (uint32_t)(off_c + k * s_pass_rel) : OOB, 0, 0);
      }
    }
    stamp(6);
  }
  if constexpr (DIAG) {
    if (blockIdx.x == 0 && tid == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.out);
      for (int i = 0; i < 7; ++i) dbg[i] = ph[i];
      dbg[7] = (unsigned long long)n_mine;
    }
  }
}

static int sl1_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

template <typename T>
static int launch_stem_l1(StemL1Params& p, hipStream_t st) {
  constexpr int TH = SL1_TH;
  constexpr int WIN_B = (((4 * TH + 3) * 52 * 4 + 15) / 16) * 16, PATCH_B = (((2 * TH + 1) * 33 * 64 + 1023) / 1024) * 1024, STG_B = TH * 16 * 128;
  constexpr int LDS = WIN_B + PATCH_B + STG_B + 128;
#if MOY_DIAG
  static const int diag = garbage_mode_env("MOY_SL1_DIAG");
  auto kern = diag == 4 ? stem_l1_kernel<T, 4> : diag == 3 ? stem_l1_kernel<T, 3> : diag == 2 ? stem_l1_kernel<T, 2> : diag ? stem_l1_kernel<T, 1> : stem_l1_kernel<T, 0>;
#else
  auto kern = stem_l1_kernel<T, 0>;
#endif
  static bool attr_set = false;
  if (!attr_set) {
    if (LDS > 65536 && (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_l1_kernel<T, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess ||
                        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess))
      return MOY_ELAUNCH;
    attr_set = true;
  }
  const int Ho = p.H / 4, Wo = p.W / 4;
  p.tiles_x = (Wo + 15) / 16;
  p.tiles_img = p.tiles_x * ((Ho + TH - 1) / TH);
  p.ntiles = p.B * p.tiles_img;
  p.fd_timg = make_fastdiv((uint32_t)p.tiles_img);
  p.fd_tx = make_fastdiv((uint32_t)p.tiles_x);
  p.per_xcd = (p.ntiles + 7) / 8;
  p.bpx = cu_limit(sl1_num_cus()) / 8 * 2;           // two resident blocks per CU
  if (p.bpx > p.per_xcd) p.bpx = p.per_xcd;
  if (p.bpx < 1) p.bpx = 1;
  hipLaunchKernelGGL(kern, dim3(8 * p.bpx), dim3(512), LDS, st, p);
  return launch_status();
}

}  // namespace moy

using namespace moy;

extern "C" int moy_stem_l1_fused(const void* in_u8, int B, int H, int W, const void* w0, const float* scale0, const float* shift0,
                                 const void* w1, const float* scale1, const float* shift1, void* out, int64_t ldc, int dtype,
                                 void* stream) {
  if (!in_u8 || !w0 || !scale0 || !shift0 || !w1 || !scale1 || !shift1 || !out || B <= 0 || H <= 0 || W <= 0) return MOY_EINVAL;
  if ((H % 4) || (W % 4)) return MOY_EINVAL;                         // two stride-2 stages; W % 4 also fixes the byte phase of the window rows
  if (dtype != MOY_BF16 && dtype != MOY_F16) return MOY_ENOSYS;      // (fp32: moy_stem_conv + moy_gemm, the parity path)
  if (ldc < 64 || (ldc % 8) || !aligned16(out) || !aligned16(w0) || !aligned16(w1) || !aligned16(scale0) || !aligned16(shift0) ||
      !aligned16(scale1) || !aligned16(shift1) || (reinterpret_cast<uintptr_t>(in_u8) % 4))
    return MOY_EINVAL;
  if ((int64_t)(H / 4) * (W / 4) * ldc * 2 > 0x3fffffffLL) return MOY_ENOSYS;
  StemL1Params p{};
  p.in = static_cast<const uint8_t*>(in_u8); p.B = B; p.H = H; p.W = W;
  p.w0 = w0; p.sc0 = scale0; p.sh0 = shift0; p.w1 = w1; p.Kpad1 = 320; p.sc1 = scale1; p.sh1 = shift1; p.out = out; p.ldc = ldc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == MOY_BF16 ? launch_stem_l1<bf16_t>(p, st) : launch_stem_l1<f16_t>(p, st);
}
