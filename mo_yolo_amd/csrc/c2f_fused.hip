// A whole C2f block in ONE persistent kernel: the first (largest-M) C2f of the backbone, 64 -> [32 | 32] -> 64 channels, n = 1,
// shortcut (yolo_track.yaml:19 over ultralytics/nn/modules/block.py:219-240 C2f, :271-283 Bottleneck, conv.py:36-38 Conv):
//
//     y0 | y1 = SiLU(BN(cv1 . x))                      1x1, 64 -> 64
//     z       = SiLU(BN(m.cv1 * y1))                   3x3, 32 -> 32
//     y2      = y1 + SiLU(BN(m.cv2 * z))               3x3, 32 -> 32
//     out     = SiLU(BN(cv2 . [y0 | y1 | y2]))         1x1, 96 -> 64
//
// Why: as four launches this block was 2.66 ms of a 25.7 ms pass (per-launch table of the plan before this kernel),
// every one of them bound by HBM at the largest pixel count of the network (11.9 M pixels per 288 frames): 448 channel-values moved
// per pixel where the block's input and output are 128.  Fused, y0, y1, z and y2 only exist as images of one block in LDS:
//   * a block walks a STRIP -- a column of 8 x (<= 30)-pixel tiles of one image -- from top to bottom and keeps y0, y1 and z in ROW
//     RINGS (image row a in slot a & 15): the rows two neighbouring tiles share (4 of y1's 12, 2 of z's 10) are computed once, so a
//     tile adds 8 new rows to each ring; a pseudo tile above the first one fills the rings.  Column halos (y1 on 34, z on 32 columns)
//     are recomputed by the horizontal neighbours.  SiLU work per output pixel 224 values instead of 272 with independent tiles --
//     the kernel is bound by its SiLU epilogues (2 transcendentals per value), not by bytes or the matrix pipe;
//   * work units: every block gets the same number of whole strips, the strips left over are cut into row ranges (one per block);
//   * cv1 reads its B operands STRAIGHT from global memory (a lane's fragment slice is 16 contiguous bytes of one pixel), one
//     tile ahead, into registers (17 fragments per tile: two per wave, the 17th split over four waves); everything else is
//     LDS -> MFMA -> LDS;
//   * all four weight sets stay in registers for the whole launch (116 VGPRs per wave), the BN tables in LDS;
//   * pixels outside the image are ZERO in y1 and z (the zero padding of the two 3x3 convs), not SiLU(BN(0));
//   * every intermediate is rounded to T exactly where the separate launches store it, the taps accumulate in the same order
//     (kx outer, ky inner): the result equals the four-launch path up to the fp32 summation order inside one MFMA.
// LDS images of the 32-channel tensors are CHUNK-PLANAR: plane c (channels 8c .. 8c+7) holds 16 bytes per pixel, pixels in patch
// order.  An MFMA B-fragment read (lane (r, q): chunk q of pixel P + r) is then 256 contiguous bytes per 16-lane group -- conflict
// free without an XOR swizzle -- and its address is ONE per-lane base register plus an immediate for every (row, column fragment,
// tap column) (+ one addition per ring row): the XOR-swizzled pixel-major images of the first version spent 5 VALU operations per
// fragment address, more than the SiLU epilogues.  The output tile is pixel-major [pixel][64 ch] with chunk ^= pixel & 7 (whole-line
// 16-byte stores).  156 KB of LDS, one block of 8 waves per CU.
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
  int tiles_x, nstrips, bpx;        // bpx: blocks per XCD
  int nfull, rem, parts;            // whole strips per block; strips left over; row ranges each of those is cut into
  FastDiv fd_tx;
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

constexpr int C2F_TH = 8, C2F_PW = 34, C2F_TWMAX = 30, C2F_RING = 16;
constexpr int C2F_PLR = C2F_RING * C2F_PW * 16, C2F_PL2 = C2F_TH * C2F_PW * 16;   // plane sizes: row-ring images, tile-local image
constexpr int C2F_SGB = C2F_TH * 32 * 128;
constexpr int C2F_LDS = 12 * C2F_PLR + 4 * C2F_PL2 + C2F_SGB + 384 * 4;

__device__ __forceinline__ f32x4 silu4(f32x4 v) { return f32x4{siluf_(v.x), siluf_(v.y), siluf_(v.z), siluf_(v.w)}; }

template <typename T, int DIAG = 0>     // DIAG = 1: s_memtime stamps of wave 0 (MOY_C2F_DIAG=1; garbage at the head of the output)
__global__ __launch_bounds__(512, 2) void c2f_fused_kernel(const C2fParams p) {
  constexpr int TH = C2F_TH, PW = C2F_PW, RING = C2F_RING, ROWB = PW * 16;
  constexpr int NPN = TH * PW;                                          // y0 / y1 pixels new per tile: 8 rows x 34
  constexpr int NF1 = NPN / 16, FPW1 = 3;                               // 17 fragments: two per wave + the 17th split over four waves
  static_assert(NF1 == 17, "cv1 fragment dealing");
  constexpr uint32_t OOB = 0x80000000u;
  static_assert(sizeof(T) == 2 && NPN % 16 == 0, "16-bit types; whole fragments");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PLR = C2F_PLR, PL2 = C2F_PL2;
  unsigned char* Y1 = smem;                         // 4 planes x [RING rows][PW]: y1, image row a in ring slot a & 15
  unsigned char* Y0 = Y1 + 4 * PLR;                 // same indexing (columns >= 2 of image rows only)
  unsigned char* Zs = Y0 + 4 * PLR;                 // same indexing, z column zc = image column x0 - 1 + zc (columns 32, 33: never written)
  unsigned char* Y2 = Zs + 4 * PLR;                 // 4 planes x [TH][PW] pixels of the tile
  unsigned char* Sg = Y2 + 4 * PL2;                 // [TH][32] pixels x 64 ch: output tile
  float* bn = reinterpret_cast<float*>(Sg + C2F_SGB);      // sc1 sh1 [64] | sca sha [32] | scb shb [32] | sc2 sh2 [64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  // ---- this block's work units.  A strip = one column of tiles of one image, walked top to bottom (a pseudo tile above its first
  // tile fills the rings).  Every block takes `nfull` whole strips -- XCD x owns the strip range [x, x + 1) * nfull * bpx, its
  // blocks consecutive strips: horizontal neighbours share halo columns through that L2 -- and the `rem` strips left over are cut
  // into `parts` row ranges, one per block, so that no block walks a whole extra strip while the others idle.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nty = (p.H + TH - 1) / TH;
  const int lin = xcd * p.bpx + slot;
  const bool has_part = lin < p.rem * p.parts;
  const int n_units = p.nfull + (has_part ? 1 : 0);
  if (n_units == 0) return;
  struct Unit { int sid, kb, ke; };                 // strip, tiles [kb, ke)
  auto unit_of = [&](int u) {
    Unit un;
    if (u < p.nfull) {
      un.sid = xcd * (p.nfull * p.bpx) + u * p.bpx + slot; un.kb = 0; un.ke = nty;
    } else {
      const int sp = lin / p.parts, part = lin - sp * p.parts;
      un.sid = p.nfull * 8 * p.bpx + sp;
      un.kb = part * nty / p.parts; un.ke = (part + 1) * nty / p.parts;
    }
    return un;
  };

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

  struct Tile { int b, y0, x0; };                   // y0 = (kb - 1) * TH: the pseudo tile that starts a unit
  auto tile_of = [&](int sid, int k) {
    Tile t;
    const int id = min(sid, p.nstrips - 1);
    t.b = (int)fdiv((uint32_t)id, p.fd_tx);
    t.x0 = (id - t.b * p.tiles_x) * p.TW;
    t.y0 = k * TH;
    return t;
  };

  // ---- weights -> registers (MFMA A operand: lane (r, q) holds W[n + r][k .. k + 7], k = slice*32 + q*8)
  const T* W1 = static_cast<const T*>(p.W1);
  const T* Wa = static_cast<const T*>(p.Wa);
  const T* Wb = static_cast<const T*>(p.Wb);
  const T* W2 = static_cast<const T*>(p.W2);
  const int wn = wave & 1, wm = wave >> 1;          // the 3x3 convs: 16-channel half, row pair
  const int nf = wave & 3, gh = wave >> 2;          // cv2: 16-channel quarter, column half
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

  // cv1's B operands of a tile = the 8 NEW y0 | y1 rows it adds to the ring (image rows y0 + 2 .. y0 + 9, 34 columns): fragment
  // f = wave + 8 i covers pixels f*16 .. +15 of them; lane (r, q): 16 bytes at channel q*8 of each 32-channel half.  Pixels outside
  // the image: out-of-range offsets (zeros; the epilogue zeroes y1 there anyway).
  u32x4 xr[FPW1][2];
  auto load_x = [&](const Tile& t) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Xg + (int64_t)t.b * img_x), 0, (uint32_t)(img_x * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < FPW1; ++i) {
      const int f = i < 2 ? wave + 8 * i : 16;          // the 17th fragment is shared by waves 0-3, one 16-channel quarter each
      const int pp = f * 16 + r;
      const int jj = pp / PW, col = pp - jj * PW;
      const int iy = t.y0 + 2 + jj, ix = t.x0 - 2 + col;
      const bool ok = (i < 2 || wave < 4) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
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
  auto slot_of = [&](int a) { return (a + 4 * RING) & (RING - 1); };       // ring slot of image row a (a >= -4 RING)

  // One 3x3 conv phase of this wave: two output rows, both 16-pixel column fragments, 16 output channels (half wn).  src: a ring
  // image; the four source rows are image rows a0 .. a0 + 3; a fragment row read for tap column kx serves the three tap rows of
  // both output rows.  Every address = (lane base + row slot offset: 4 additions per phase) + immediate.
  auto conv3 = [&](const unsigned char* src, const u32x4 (&wf)[9], int a0, f32x4 (&acc)[2][2]) {
    const int lb = q * PLR + r * 16;
    const unsigned char* rowb[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) rowb[y] = src + (lb + slot_of(a0 + y) * ROWB);
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[y][cf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        u32x4 a[4];
#pragma unroll
        for (int y = 0; y < 4; ++y) a[y] = *reinterpret_cast<const u32x4*>(rowb[y] + (cf * 16 + kx) * 16);
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) acc[y][cf] = c2f_mfma<T>(acc[y][cf], wf[ky * 3 + kx], a[y + ky]);
      }
    }
  };

  Unit un = unit_of(0);
  int u = 0, k = un.kb - 1;
  load_x(tile_of(un.sid, k));
  __syncthreads();                                   // BN tables
  if constexpr (DIAG) tprev = __builtin_amdgcn_s_memtime();
  int n_iter = 0;
  for (;;) {
    const Tile t = tile_of(un.sid, k);
    const bool pseudo = k < un.kb;
    Unit nun = un;
    int nu = u, kn = k + 1;
    if (kn == un.ke) {
      ++nu;
      if (nu < n_units) { nun = unit_of(nu); kn = nun.kb - 1; }
    }
    const bool last = nu >= n_units;
    ++n_iter;

    // ---- cv1 on the 8 new rows (image rows y0 + 2 .. y0 + 9): y1 (zero outside the frame) and y0 into their ring slots
    {
      const int plane_off = (q >> 1) * PLR + (q & 1) * 8;
#pragma unroll
      for (int i = 0; i < FPW1; ++i) {
        if (i == 2 && wave >= 4) continue;              // wave-uniform
        const int pp = (i < 2 ? wave + 8 * i : 16) * 16 + r;
        const int jj = pp / PW, col = pp - jj * PW;
        const int arow = t.y0 + 2 + jj;
        const bool inside = (unsigned)arow < (unsigned)p.H && (unsigned)(t.x0 - 2 + col) < (unsigned)p.Wd;
        const bool interior = col >= 2 && (unsigned)arow < (unsigned)p.H;
        unsigned char* w1p = Y1 + plane_off + (slot_of(arow) * PW + col) * 16;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if (i == 2 && n != wave) continue;            // wave-uniform
          f32x4 acc = c2f_mfma<T>(f32x4{0.f, 0.f, 0.f, 0.f}, w1f[n][0], xr[i][0]);
          acc = c2f_mfma<T>(acc, w1f[n][1], xr[i][1]);
          const f32x4 sc = *reinterpret_cast<const f32x4*>(bn + n * 16 + q * 4), sh = *reinterpret_cast<const f32x4*>(bn + 64 + n * 16 + q * 4);
          const f32x4 v = silu4(acc * sc + sh);
          u32x2 o = {DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
          if (n < 2) {
            if (interior) *reinterpret_cast<u32x2*>(w1p + 4 * PLR + n * 2 * PLR) = o;
          } else {
            if (!inside) o = u32x2{0u, 0u};
            *reinterpret_cast<u32x2*>(w1p + (n - 2) * 2 * PLR) = o;
          }
        }
      }
    }
    if (!last) load_x(tile_of(nun.sid, kn));           // in flight under the phases below
    stamp(0);
    __syncthreads();                                   // y0, y1 rows complete
    stamp(1);

    // ---- m.cv1: the 8 new z rows (image rows y0 + 1 .. y0 + 8; the first two of the tile's ten were made by the tile above)
    {
      f32x4 acc[2][2];
      const int az = t.y0 + 1 + 2 * wm;              // first of this wave's two z rows; sources: y1 rows az - 1 .. az + 2
      conv3(Y1, waf, az - 1, acc);
      const f32x4 sc = *reinterpret_cast<const f32x4*>(bn + 128 + wn * 16 + q * 4), sh = *reinterpret_cast<const f32x4*>(bn + 160 + wn * 16 + q * 4);
      const int zl = (wn * 2 + (q >> 1)) * PLR + (q & 1) * 8 + r * 16;
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        unsigned char* zb = Zs + (zl + slot_of(az + y) * ROWB);
        const bool yin = (unsigned)(az + y) < (unsigned)p.H;
#pragma unroll
        for (int cf = 0; cf < 2; ++cf) {
          const bool inside = yin && (unsigned)(t.x0 - 1 + cf * 16 + r) < (unsigned)p.Wd;
          const f32x4 v = silu4(acc[y][cf] * sc + sh);
          u32x2 o = {DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
          if (!inside) o = u32x2{0u, 0u};
          *reinterpret_cast<u32x2*>(zb + cf * 256) = o;
        }
      }
    }
    stamp(2);
    __syncthreads();                                   // z complete
    stamp(3);

    if (!pseudo) {                                     // (the pseudo tile only feeds the rings)
      // ---- m.cv2 + shortcut: y2 = y1 + SiLU(BN(.)), output rows 2 wm, 2 wm + 1 of the tile
      {
        f32x4 acc[2][2];
        const int ao = t.y0 + 2 * wm;                // image row of the first output row; sources: z rows ao - 1 .. ao + 2
        conv3(Zs, wbf, ao - 1, acc);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(bn + 192 + wn * 16 + q * 4), sh = *reinterpret_cast<const f32x4*>(bn + 224 + wn * 16 + q * 4);
        const int cell = (wn * 2 + (q >> 1));
        unsigned char* ob = Y2 + cell * PL2 + (q & 1) * 8 + (2 * wm * PW + r) * 16;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          const unsigned char* rb = Y1 + cell * PLR + (q & 1) * 8 + (slot_of(ao + y) * PW + r + 2) * 16;
#pragma unroll
          for (int cf = 0; cf < 2; ++cf) {
            f32x4 v = silu4(acc[y][cf] * sc + sh);
            const u32x2 res = *reinterpret_cast<const u32x2*>(rb + cf * 256);
            v += f32x4{DT<T>::lo(res.x), DT<T>::hi(res.x), DT<T>::lo(res.y), DT<T>::hi(res.y)};
            *reinterpret_cast<u32x2*>(ob + (y * PW + cf * 16) * 16) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
          }
        }
      }
      stamp(4);
      __syncthreads();                                 // y2 complete
      stamp(5);

      // ---- cv2 over [y0 | y1 | y2] of the tile pixels: fragment (row i, column half gh); this wave: channel quarter nf
      {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(bn + 256 + nf * 16 + q * 4), sh = *reinterpret_cast<const f32x4*>(bn + 320 + nf * 16 + q * 4);
        const int l1 = q * PLR + (gh * 16 + r + 2) * 16;
        const unsigned char* b2 = Y2 + q * PL2 + (gh * 16 + r) * 16;
        const int pix0 = gh * 16 + r;
        unsigned char* sb = Sg + pix0 * 128 + (((nf * 2 + (q >> 1)) ^ (pix0 & 7)) * 16) + (q & 1) * 8;
#pragma unroll
        for (int i = 0; i < TH; ++i) {
          const unsigned char* b1 = Y1 + (l1 + slot_of(t.y0 + i) * ROWB);
          const u32x4 a0 = *reinterpret_cast<const u32x4*>(b1 + 4 * PLR);
          const u32x4 a1 = *reinterpret_cast<const u32x4*>(b1);
          const u32x4 a2 = *reinterpret_cast<const u32x4*>(b2 + i * ROWB);
          f32x4 acc = c2f_mfma<T>(f32x4{0.f, 0.f, 0.f, 0.f}, w2f[0], a0);
          acc = c2f_mfma<T>(acc, w2f[1], a1);
          acc = c2f_mfma<T>(acc, w2f[2], a2);
          const f32x4 v = silu4(acc * sc + sh);
          *reinterpret_cast<u32x2*>(sb + i * 32 * 128) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
        }
      }
      stamp(6);
      __syncthreads();                                 // output tile complete
      {
        const auto rsO = __builtin_amdgcn_make_buffer_rsrc(Og + (int64_t)t.b * img_o, 0, (uint32_t)(img_o * 2), 0x00020000);
        u32x4 vv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int pix = kk * 64 + (tid >> 3), c = tid & 7;
          vv[kk] = *reinterpret_cast<const u32x4*>(Sg + pix * 128 + ((c ^ (pix & 7)) * 16));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int pix = kk * 64 + (tid >> 3), c = tid & 7;
          const int orow = pix >> 5, oc = pix & 31;
          const bool ok = oc < p.TW && t.x0 + oc < p.Wd && t.y0 + orow < p.H;
          const uint32_t off = (uint32_t)((((t.y0 + orow) * p.Wd + t.x0 + oc) * (int)p.ldo + c * 8) * 2);
          __builtin_amdgcn_raw_buffer_store_b128(vv[kk], rsO, ok ? off : OOB, 0, 0);
        }
      }
      stamp(7);
    }
    if (last) break;
    un = nun; u = nu; k = kn;
  }
  if constexpr (DIAG) {
    if (blockIdx.x == 0 && tid == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.Out);
      for (int i = 0; i < 8; ++i) dbg[i] = ph[i];
      dbg[8] = (unsigned long long)n_iter;
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
#if MOY_DIAG
  static const int diag = garbage_mode_env("MOY_C2F_DIAG");
  auto kern = diag ? c2f_fused_kernel<T, 1> : c2f_fused_kernel<T, 0>;
#else
  auto kern = c2f_fused_kernel<T, 0>;
#endif
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(c2f_fused_kernel<T, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, C2F_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C2F_LDS) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  const int nx = (p.Wd + C2F_TWMAX - 1) / C2F_TWMAX;
  p.TW = (p.Wd + nx - 1) / nx;                       // equal tile widths <= 30 (W = 272: 10 tiles of 28, not 9 of 30 + one of 2)
  p.tiles_x = (p.Wd + p.TW - 1) / p.TW;
  p.nstrips = p.B * p.tiles_x;
  p.fd_tx = make_fastdiv((uint32_t)p.tiles_x);
  p.bpx = cu_limit(c2f_num_cus()) / 8;
  if (p.bpx < 1) p.bpx = 1;
  const int nb = 8 * p.bpx, nty = (p.H + C2F_TH - 1) / C2F_TH;
  p.nfull = p.nstrips / nb;
  p.rem = p.nstrips - p.nfull * nb;
  p.parts = p.rem ? nb / p.rem : 1;
  if (p.parts > nty) p.parts = nty;                  // at least one tile per part
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
