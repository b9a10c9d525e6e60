// A whole C2f block in ONE persistent kernel: the first (largest-M) C2f of the backbone, 64 -> [32 | 32] -> 64 channels, n = 1,
// shortcut (yolo_track.yaml:19 over ultralytics/nn/modules/block.py:219-240 C2f, :271-283 Bottleneck, conv.py:36-38 Conv):
//
//     y0 | y1 = SiLU(BN(cv1 . x))                      1x1, 64 -> 64
//     z       = SiLU(BN(m.cv1 * y1))                   3x3, 32 -> 32
//     y2      = y1 + SiLU(BN(m.cv2 * z))               3x3, 32 -> 32
//     out     = SiLU(BN(cv2 . [y0 | y1 | y2]))         1x1, 96 -> 64
//
// Why: as four launches this block was 2.66 ms of a 25.7 ms pass (per-launch table of the plan before this kernel), every one
// of them bound by HBM at the largest pixel count of the network (11.9 M pixels per 288 frames): 448 channel-values moved per pixel
// where the block's input and output are 128.  Fused, y0, y1, z and y2 only exist as the tile images of one block in LDS:
//   * a block owns TH x TW (8 x <= 30) output pixels; z is needed on (TH+2) x 32 pixels, y1 on (TH+4) x 34 (conv halos, recomputed
//     by the neighbours: 1.7x on cv1, 1.33x on the first 3x3 -- cheap next to the traffic they replace);
//   * cv1 reads its B operands STRAIGHT from global memory (a lane's fragment slice is 16 contiguous bytes of one pixel), one
//     tile ahead, into registers; everything else is LDS -> MFMA -> LDS;
//   * all four weight sets stay in registers for the whole launch (116 VGPRs per wave), the BN tables in LDS;
//   * pixels outside the image are ZERO in y1 and z (the zero padding of the two 3x3 convs), not SiLU(BN(0));
//   * every intermediate is rounded to T exactly where the separate launches store it, the taps accumulate in the same order
//     (kx outer, ky inner): the result equals the four-launch path up to the fp32 summation order inside one MFMA.
// LDS images of the 32-channel tensors are CHUNK-PLANAR: plane c (channels 8c .. 8c+7) holds 16 bytes per pixel, pixels in patch
// order.  An MFMA B-fragment read (lane (r, q): chunk q of pixel P + r) is then 256 contiguous bytes per 16-lane group -- conflict
// free without an XOR swizzle -- and its address is ONE per-lane base register plus an immediate for every (row, column fragment,
// tap column): the XOR-swizzled pixel-major images of the first version spent 5 VALU operations per fragment address, more than
// the SiLU epilogues.  The output tile is pixel-major [pixel][64 ch] with chunk ^= pixel & 7 (whole-line 16-byte stores).
// 126 KB of LDS, one block of 8 waves per CU.
#include "common.hpp"

#include <type_traits>

namespace moy {

struct C2fParams {
  const void* X; int64_t ldx;                                    // T [B*H*W][>= 64]
  const void* W1; int kp1; const float* sc1; const float* sh1;  // cv1   T [64][kp1]
  const void* Wa; const float* sca; const float* sha;            // m.cv1 T [32][kpb], k = (ky*3+kx)*32 + c
  const void* Wb; const float* scb; const float* shb; int kpb;   // m.cv2
  const void* W2; int kp2; const float* sc2; const float* sh2;  // cv2   T [64][kp2], k = [y0 | y1 | y2]
  void* Out; int64_t ldo;
  int B, H, Wd, TW;
  int tiles_x, tiles_img, ntiles, per_xcd, bpx;
  FastDiv fd_timg, fd_tx;
};

template <typename T>
__device__ __forceinline__ f32x4 c2f_mfma(f32x4 acc, u32x4 w, u32x4 a);
template <>
__device__ __forceinline__ f32x4 c2f_mfma<bf16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 c2f_mfma<f16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}

constexpr int C2F_TH = 8, C2F_PW = 34, C2F_TWMAX = 30;
constexpr int C2F_PL1 = (C2F_TH + 4) * C2F_PW * 16, C2F_PLZ = (C2F_TH + 2) * C2F_PW * 16, C2F_PL2 = C2F_TH * C2F_PW * 16;   // plane sizes
constexpr int C2F_SGB = C2F_TH * 32 * 128;
constexpr int C2F_LDS = 8 * C2F_PL1 + 4 * C2F_PLZ + 4 * C2F_PL2 + C2F_SGB + 384 * 4;

__device__ __forceinline__ f32x4 silu4(f32x4 v) { return f32x4{siluf_(v.x), siluf_(v.y), siluf_(v.z), siluf_(v.w)}; }

template <typename T, int DIAG = 0>     // DIAG = 1: s_memtime stamps of wave 0 (MOY_C2F_DIAG=1; garbage at the head of the output)
__global__ __launch_bounds__(512, 2) void c2f_fused_kernel(const C2fParams p) {
  constexpr int TH = C2F_TH, PW = C2F_PW, PH = TH + 4, NP1 = PH * PW;
  constexpr int NF1 = (NP1 + 15) / 16, FPW1 = (NF1 + 7) / 8;           // cv1 fragments (16 patch pixels each), per wave
  constexpr uint32_t OOB = 0x80000000u;
  static_assert(sizeof(T) == 2, "16-bit types");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PL1 = C2F_PL1, PLZ = C2F_PLZ, PL2 = C2F_PL2;
  unsigned char* Y1 = smem;                         // 4 planes x [PH][PW] pixels: y1 with a halo of 2
  unsigned char* Y0 = Y1 + 4 * PL1;                 // same pixel indexing; only the tile pixels are written / read
  unsigned char* Zs = Y0 + 4 * PL1;                 // 4 planes x [TH+2][PW] pixels: z with a halo of 1 (columns 32, 33: never written)
  unsigned char* Y2 = Zs + 4 * PLZ;                 // 4 planes x [TH][PW] pixels
  unsigned char* Sg = Y2 + 4 * PL2;                 // [TH][32] pixels x 64 ch: output tile
  float* bn = reinterpret_cast<float*>(Sg + C2F_SGB);      // sc1 sh1 [64] | sca sha [32] | scb shb [32] | sc2 sh2 [64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t_first = xcd * p.per_xcd + slot;
  const int t_limit = min((xcd + 1) * p.per_xcd, p.ntiles);
  if (t_first >= t_limit) return;
  const int n_mine = (t_limit - t_first + p.bpx - 1) / p.bpx;

  if (tid < 384) {
    float v;
    if (tid < 64) v = p.sc1[tid];
    else if (tid < 128) v = p.sh1[tid - 64];
    else if (tid < 160) v = p.sca[tid - 128];
    else if (tid < 192) v = p.sha[tid - 160];
    else if (tid < 224) v = p.scb[tid - 192];
    else if (tid < 256) v = p.shb[tid - 224];
    else if (tid < 320) v = p.sc2[tid - 256];
    else v = p.sh2[tid - 320];
    bn[tid] = v;
  }

  struct Tile { int b, y0, x0; };
  auto tile_of = [&](int it) {
    Tile t;
    const int id = min(t_first + it * p.bpx, p.ntiles - 1);
    t.b = (int)fdiv((uint32_t)id, p.fd_timg);
    const int rem = id - t.b * p.tiles_img;
    const int ty = (int)fdiv((uint32_t)rem, p.fd_tx);
    t.y0 = ty * TH;
    t.x0 = (rem - ty * p.tiles_x) * p.TW;
    return t;
  };

  // ---- weights -> registers (MFMA A operand: lane (r, q) holds W[n + r][k .. k + 7], k = slice*32 + q*8)
  const T* W1 = static_cast<const T*>(p.W1);
  const T* Wa = static_cast<const T*>(p.Wa);
  const T* Wb = static_cast<const T*>(p.Wb);
  const T* W2 = static_cast<const T*>(p.W2);
  const int wn = wave & 1, wm = wave >> 1;          // the 3x3 convs: 16-channel half, row group
  const int nf = wave & 3, gh = wave >> 2;          // cv2: 16-channel quarter, fragment parity
  u32x4 w1f[4][2], waf[9], wbf[9], w2f[3];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int s = 0; s < 2; ++s) w1f[n][s] = *reinterpret_cast<const u32x4*>(W1 + (int64_t)(n * 16 + r) * p.kp1 + s * 32 + q * 8);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    waf[tap] = *reinterpret_cast<const u32x4*>(Wa + (int64_t)(wn * 16 + r) * p.kpb + tap * 32 + q * 8);
    wbf[tap] = *reinterpret_cast<const u32x4*>(Wb + (int64_t)(wn * 16 + r) * p.kpb + tap * 32 + q * 8);
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) w2f[s] = *reinterpret_cast<const u32x4*>(W2 + (int64_t)(nf * 16 + r) * p.kp2 + s * 32 + q * 8);

  const T* __restrict__ Xg = static_cast<const T*>(p.X);
  T* __restrict__ Og = static_cast<T*>(p.Out);
  const int64_t img_x = (int64_t)p.H * p.Wd * p.ldx, img_o = (int64_t)p.H * p.Wd * p.ldo;

  // cv1's B operands of a tile: fragment f = wave + 8 i covers patch pixels f*16 .. +15; lane (r, q): 16 bytes at channel q*8 of
  // each 32-channel half.  Pixels outside the image: out-of-range offsets (zeros; the epilogue zeroes y1 there anyway).
  u32x4 xr[FPW1][2];
  auto load_x = [&](const Tile& t) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Xg + (int64_t)t.b * img_x), 0, (uint32_t)(img_x * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < FPW1; ++i) {
      const int p1 = (wave + 8 * i) * 16 + r;
      const int row = p1 / PW, col = p1 - row * PW;
      const int iy = t.y0 - 2 + row, ix = t.x0 - 2 + col;
      const bool ok = p1 < NP1 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
      const uint32_t off = (uint32_t)(((iy * p.Wd + ix) * (int)p.ldx + q * 8) * 2);
      xr[i][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0));
      xr[i][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 64 : OOB, 0, 0));
    }
  };

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

  // One 3x3 conv phase of this wave: NR output rows from row r0, both 16-pixel column fragments, 16 output channels (half wn).
  // src: [rows][PW] pixel image; a fragment row read for tap column kx serves the three tap rows of up to NR output rows.
  auto conv3 = [&](auto nr_tag, const unsigned char* src, int plane, const u32x4 (&wf)[9], int r0, f32x4 (&acc)[3][2]) {
    constexpr int NR = decltype(nr_tag)::value;
    const unsigned char* base = src + q * plane + (r0 * PW + r) * 16;       // every fragment address = base + immediate
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
#pragma unroll
      for (int y = 0; y < NR; ++y) acc[y][cf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        u32x4 a[NR + 2];
#pragma unroll
        for (int y = 0; y < NR + 2; ++y) a[y] = *reinterpret_cast<const u32x4*>(base + (y * PW + cf * 16 + kx) * 16);
#pragma unroll
        for (int y = 0; y < NR; ++y)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) acc[y][cf] = c2f_mfma<T>(acc[y][cf], wf[ky * 3 + kx], a[y + ky]);
      }
    }
  };
  // first 3x3: z = SiLU(BN(.)) on rows r0 .. r0+NR-1 of the z image, zero outside the frame
  auto phase_z = [&](auto nr_tag, const Tile& t, int r0) {
    constexpr int NR = decltype(nr_tag)::value;
    f32x4 acc[3][2];
    conv3(nr_tag, Y1, PL1, waf, r0, acc);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(bn + 128 + wn * 16 + q * 4), sh = *reinterpret_cast<const f32x4*>(bn + 160 + wn * 16 + q * 4);
    unsigned char* zb = Zs + (wn * 2 + (q >> 1)) * PLZ + (q & 1) * 8 + (r0 * PW + r) * 16;
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
      const bool xin = (unsigned)(t.x0 - 1 + cf * 16 + r) < (unsigned)p.Wd;
#pragma unroll
      for (int y = 0; y < NR; ++y) {
        const bool inside = xin && (unsigned)(t.y0 - 1 + r0 + y) < (unsigned)p.H;
        const f32x4 v = silu4(acc[y][cf] * sc + sh);
        u32x2 o = {DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
        if (!inside) o = u32x2{0u, 0u};
        *reinterpret_cast<u32x2*>(zb + (y * PW + cf * 16) * 16) = o;
      }
    }
  };

  load_x(tile_of(0));
  __syncthreads();                                   // BN tables
  if constexpr (DIAG) tprev = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n_mine; ++it) {
    const Tile t = tile_of(it);

    // ---- cv1 on the (TH+4) x PW patch: y1 (all pixels, zero outside the frame) and y0 (tile pixels only)
    {
      unsigned char* wb1 = Y1 + (q >> 1) * PL1 + (q & 1) * 8 + (wave * 16 + r) * 16;
      unsigned char* wb0 = wb1 + 4 * PL1;
#pragma unroll
      for (int i = 0; i < FPW1; ++i) {
        if (wave + 8 * i >= NF1) continue;              // wave-uniform
        const int p1 = (wave + 8 * i) * 16 + r;
        const int row = p1 / PW, col = p1 - row * PW;
        const bool inside = (unsigned)(t.y0 - 2 + row) < (unsigned)p.H && (unsigned)(t.x0 - 2 + col) < (unsigned)p.Wd;
        const bool interior = (unsigned)(row - 2) < (unsigned)TH && col >= 2;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          f32x4 acc = c2f_mfma<T>(f32x4{0.f, 0.f, 0.f, 0.f}, w1f[n][0], xr[i][0]);
          acc = c2f_mfma<T>(acc, w1f[n][1], xr[i][1]);
          const f32x4 sc = *reinterpret_cast<const f32x4*>(bn + n * 16 + q * 4), sh = *reinterpret_cast<const f32x4*>(bn + 64 + n * 16 + q * 4);
          const f32x4 v = silu4(acc * sc + sh);
          u32x2 o = {DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
          if (n < 2) {
            if (interior) *reinterpret_cast<u32x2*>(wb0 + n * 2 * PL1 + i * 128 * 16) = o;
          } else {
            if (!inside) o = u32x2{0u, 0u};
            if (p1 < NP1) *reinterpret_cast<u32x2*>(wb1 + (n - 2) * 2 * PL1 + i * 128 * 16) = o;
          }
        }
      }
    }
    if (it + 1 < n_mine) load_x(tile_of(it + 1));     // in flight under the three phases below
    stamp(0);
    __syncthreads();                                   // y0, y1 complete
    stamp(1);

    // ---- m.cv1: z rows {0-2, 3-5, 6-7, 8-9} for row groups wm = 0..3
    if (wm < 2) phase_z(std::integral_constant<int, 3>{}, t, 3 * wm);
    else phase_z(std::integral_constant<int, 2>{}, t, 2 * wm + 2);
    stamp(2);
    __syncthreads();                                   // z complete
    stamp(3);

    // ---- m.cv2 + shortcut: y2 = y1 + SiLU(BN(.)), output rows 2 wm, 2 wm + 1
    {
      f32x4 acc[3][2];
      conv3(std::integral_constant<int, 2>{}, Zs, PLZ, wbf, 2 * wm, acc);
      const f32x4 sc = *reinterpret_cast<const f32x4*>(bn + 192 + wn * 16 + q * 4), sh = *reinterpret_cast<const f32x4*>(bn + 224 + wn * 16 + q * 4);
      const int cell = (wn * 2 + (q >> 1));
      const unsigned char* rb = Y1 + cell * PL1 + (q & 1) * 8 + ((2 * wm + 2) * PW + r + 2) * 16;
      unsigned char* ob = Y2 + cell * PL2 + (q & 1) * 8 + (2 * wm * PW + r) * 16;
#pragma unroll
      for (int cf = 0; cf < 2; ++cf)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          f32x4 v = silu4(acc[y][cf] * sc + sh);
          const u32x2 res = *reinterpret_cast<const u32x2*>(rb + (y * PW + cf * 16) * 16);
          v += f32x4{DT<T>::lo(res.x), DT<T>::hi(res.x), DT<T>::lo(res.y), DT<T>::hi(res.y)};
          *reinterpret_cast<u32x2*>(ob + (y * PW + cf * 16) * 16) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
        }
    }
    stamp(4);
    __syncthreads();                                   // y2 complete
    stamp(5);

    // ---- cv2 over [y0 | y1 | y2] of the tile pixels: fragment (row i, column half gh); this wave: channel quarter nf
    {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(bn + 256 + nf * 16 + q * 4), sh = *reinterpret_cast<const f32x4*>(bn + 320 + nf * 16 + q * 4);
      const unsigned char* b1 = Y1 + q * PL1 + (2 * PW + gh * 16 + r + 2) * 16;
      const unsigned char* b0 = b1 + 4 * PL1;
      const unsigned char* b2 = Y2 + q * PL2 + (gh * 16 + r) * 16;
      const int pix0 = gh * 16 + r;
      unsigned char* sb = Sg + pix0 * 128 + (((nf * 2 + (q >> 1)) ^ (pix0 & 7)) * 16) + (q & 1) * 8;
#pragma unroll
      for (int i = 0; i < TH; ++i) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(b0 + i * PW * 16);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(b1 + i * PW * 16);
        const u32x4 a2 = *reinterpret_cast<const u32x4*>(b2 + i * PW * 16);
        f32x4 acc = c2f_mfma<T>(f32x4{0.f, 0.f, 0.f, 0.f}, w2f[0], a0);
        acc = c2f_mfma<T>(acc, w2f[1], a1);
        acc = c2f_mfma<T>(acc, w2f[2], a2);
        const f32x4 v = silu4(acc * sc + sh);
        *reinterpret_cast<u32x2*>(sb + i * 32 * 128) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
      }
    }
    stamp(6);
    __syncthreads();                                   // output tile complete (and y0 / y1 / y2 free for the next tile's cv1)
    {
      const auto rsO = __builtin_amdgcn_make_buffer_rsrc(Og + (int64_t)t.b * img_o, 0, (uint32_t)(img_o * 2), 0x00020000);
      u32x4 vv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int pix = k * 64 + (tid >> 3), c = tid & 7;
        vv[k] = *reinterpret_cast<const u32x4*>(Sg + pix * 128 + ((c ^ (pix & 7)) * 16));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int pix = k * 64 + (tid >> 3), c = tid & 7;
        const int orow = pix >> 5, oc = pix & 31;
        const bool ok = oc < p.TW && t.x0 + oc < p.Wd && t.y0 + orow < p.H;
        const uint32_t off = (uint32_t)((((t.y0 + orow) * p.Wd + t.x0 + oc) * (int)p.ldo + c * 8) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(vv[k], rsO, ok ? off : OOB, 0, 0);
      }
    }
    stamp(7);
  }
  if constexpr (DIAG) {
    if (blockIdx.x == 0 && tid == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.Out);
      for (int i = 0; i < 8; ++i) dbg[i] = ph[i];
      dbg[8] = (unsigned long long)n_mine;
    }
  }
}

static int c2f_num_cus() {
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
static int launch_c2f(C2fParams& p, hipStream_t st) {
  static int diag = -1;
  if (diag < 0) { const char* e = getenv("MOY_C2F_DIAG"); diag = e ? atoi(e) : 0; }
  auto kern = diag ? c2f_fused_kernel<T, 1> : c2f_fused_kernel<T, 0>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(c2f_fused_kernel<T, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, C2F_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(c2f_fused_kernel<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, C2F_LDS) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  const int nx = (p.Wd + C2F_TWMAX - 1) / C2F_TWMAX;
  p.TW = (p.Wd + nx - 1) / nx;                       // equal tile widths <= 30 (W = 272: 10 tiles of 28, not 9 of 30 + one of 2)
  p.tiles_x = (p.Wd + p.TW - 1) / p.TW;
  p.tiles_img = p.tiles_x * ((p.H + C2F_TH - 1) / C2F_TH);
  p.ntiles = p.B * p.tiles_img;
  p.fd_timg = make_fastdiv((uint32_t)p.tiles_img);
  p.fd_tx = make_fastdiv((uint32_t)p.tiles_x);
  p.per_xcd = (p.ntiles + 7) / 8;
  p.bpx = c2f_num_cus() / 8;
  if (p.bpx > p.per_xcd) p.bpx = p.per_xcd;
  if (p.bpx < 1) p.bpx = 1;
  hipLaunchKernelGGL(kern, dim3(8 * p.bpx), dim3(512), C2F_LDS, st, p);
  return launch_status();
}

}  // namespace moy

using namespace moy;

extern "C" int moy_c2f_fused(const moy_c2f_args* a, void* stream) {
  if (!a || !a->x || !a->out || !a->w_cv1 || !a->w_m1 || !a->w_m2 || !a->w_cv2 || a->B <= 0 || a->H <= 0 || a->W <= 0) return MOY_EINVAL;
  const float* tabs[8] = {a->scale_cv1, a->shift_cv1, a->scale_m1, a->shift_m1, a->scale_m2, a->shift_m2, a->scale_cv2, a->shift_cv2};
  for (const float* t : tabs)
    if (!t) return MOY_EINVAL;
  if (a->dtype != MOY_BF16 && a->dtype != MOY_F16) return MOY_ENOSYS;      // (fp32: the four moy_gemm launches, the parity path)
  if (a->ldx < 64 || (a->ldx % 8) || a->ldo < 64 || (a->ldo % 8) || a->kp_cv1 < 64 || (a->kp_cv1 % 8) || a->kp_m < 288 || (a->kp_m % 8) ||
      a->kp_cv2 < 96 || (a->kp_cv2 % 8))
    return MOY_EINVAL;
  if (!aligned16(a->x) || !aligned16(a->out) || !aligned16(a->w_cv1) || !aligned16(a->w_m1) || !aligned16(a->w_m2) || !aligned16(a->w_cv2))
    return MOY_EINVAL;
  if ((int64_t)a->H * a->W * a->ldx * 2 > 0x3fffffffLL || (int64_t)a->H * a->W * a->ldo * 2 > 0x3fffffffLL) return MOY_ENOSYS;
  C2fParams p{};
  p.X = a->x; p.ldx = a->ldx;
  p.W1 = a->w_cv1; p.kp1 = a->kp_cv1; p.sc1 = a->scale_cv1; p.sh1 = a->shift_cv1;
  p.Wa = a->w_m1; p.sca = a->scale_m1; p.sha = a->shift_m1;
  p.Wb = a->w_m2; p.scb = a->scale_m2; p.shb = a->shift_m2; p.kpb = a->kp_m;
  p.W2 = a->w_cv2; p.kp2 = a->kp_cv2; p.sc2 = a->scale_cv2; p.sh2 = a->shift_cv2;
  p.Out = a->out; p.ldo = a->ldo;
  p.B = a->B; p.H = a->H; p.Wd = a->W;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return a->dtype == MOY_BF16 ? launch_c2f<bf16_t>(p, st) : launch_c2f<f16_t>(p, st);
}
