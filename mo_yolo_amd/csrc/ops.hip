// Non-GEMM kernels of the tracking path for gfx950 (wave64): stem conv with fused preprocess,
// SPPF pooling, nearest upsample, narrow row-dot heads, query top-k, positional embedding,
// self-attention core, deformable-attention sampling, ID assignment + predictor rows.
// All HBM-side accesses are 8/16-byte vectors over channels-last rows; reductions and softmax use
// wave shuffles; nothing here allocates or synchronises.
#include "common.hpp"

namespace moy {

// ------------------------------------------------------------------------------------------------
// Stem: preprocess (uint8 BGR HWC -> RGB /255, predictor.py:125-133) fused into the first
// 3x3 stride-2 conv + BN + SiLU (conv.py:36-38).  One thread = one output pixel, all COUT channels;
// the 27 x COUT weights live in LDS and are read as wave-uniform broadcasts.
template <typename T, int COUT, int FMT>
__global__ __launch_bounds__(256) void stem_kernel(const void* __restrict__ in, int B, int H, int W,
                                                   const float* __restrict__ w, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, T* __restrict__ out, int64_t ldc) {
  __shared__ float ws[27 * COUT];
  __shared__ float ss[2 * COUT];
  for (int i = threadIdx.x; i < 27 * COUT; i += 256) ws[i] = w[i];
  for (int i = threadIdx.x; i < COUT; i += 256) { ss[i] = scale[i]; ss[COUT + i] = shift[i]; }
  __syncthreads();
  const int Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * Ho * Wo;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= total) return;
  const int b = (int)(pix / ((long)Ho * Wo));
  const int rem = (int)(pix - (long)b * Ho * Wo);
  const int oy = rem / Wo, ox = rem - oy * Wo;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
#pragma unroll 1
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 + ky - 1;
#pragma unroll 1
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 + kx - 1;
      float v[3] = {0.f, 0.f, 0.f};   // RGB
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
        if (FMT == 0) {
          const uint8_t* p = static_cast<const uint8_t*>(in) + (((long)b * H + iy) * W + ix) * 3;
          v[0] = (float)p[2] / 255.0f; v[1] = (float)p[1] / 255.0f; v[2] = (float)p[0] / 255.0f;
        } else {
          const float* p = static_cast<const float*>(in) + ((long)b * 3 * H + iy) * W + ix;
          v[0] = p[0]; v[1] = p[(long)H * W]; v[2] = p[2L * H * W];
        }
      }
#pragma unroll 1
      for (int ci = 0; ci < 3; ++ci) {
        const float* wr = ws + ((ky * 3 + kx) * 3 + ci) * COUT;
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = fmaf(v[ci], wr[c], acc[c]);
      }
    }
  }
  T* o = out + pix * ldc;
#pragma unroll
  for (int c = 0; c < COUT; c += 4) {
    f32x4 y;
    y.x = siluf_(acc[c] * ss[c] + ss[COUT + c]);
    y.y = siluf_(acc[c + 1] * ss[c + 1] + ss[COUT + c + 1]);
    y.z = siluf_(acc[c + 2] * ss[c + 2] + ss[COUT + c + 2]);
    y.w = siluf_(acc[c + 3] * ss[c + 3] + ss[COUT + c + 3]);
    DT<T>::store4(o + c, y);
  }
}

// ------------------------------------------------------------------------------------------------
// Stem on the matrix cores (bf16 path, uint8 input): out[pixel, 32] = W[32, 27] . patch[27] as
// v_mfma_f32_16x16x32_bf16 with K padded 27 -> 32.  A block owns an 8 x 32 tile of output pixels;
// its (17 x 65 x 3)-byte input window is brought into LDS with aligned dword loads, each lane then
// gathers the 8 taps of its K-slice, converts (u8 -> x/255 -> bf16) and feeds the MFMA; BN + SiLU on
// the accumulators (4 consecutive channels per lane), 8-byte stores.  ~15x fewer VALU ops per pixel
// than the scalar-FMA kernel, which was VALU-bound (0.5 ms of a 9 ms step).
constexpr int STEM_TH = 8, STEM_TW = 32;

template <int COUT>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const uint8_t* __restrict__ in, int B, int H, int W,
                                                        const bf16_t* __restrict__ wpad,   // [COUT][32] bf16, k = (ky*3+kx)*3 + c_rgb
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        bf16_t* __restrict__ out, int64_t ldc) {
  constexpr int NTC = COUT / 16;                       // channel tiles of 16
  constexpr int IN_ROWS = 2 * STEM_TH + 1;
  constexpr int ROW_DW = ((2 * STEM_TW + 1) * 3 + 3 + 3) / 4 + 1;     // window bytes + misalignment, in dwords
  constexpr int ROWB = ROW_DW * 4;
  __shared__ uint32_t tile[IN_ROWS][ROW_DW];
  __shared__ int mis_s[IN_ROWS];                       // byte phase of each window row inside its first dword
  const int Ho = H >> 1, Wo = W >> 1;
  const int tiles_x = (Wo + STEM_TW - 1) / STEM_TW, tiles_y = (Ho + STEM_TH - 1) / STEM_TH;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int oy0 = ty * STEM_TH, ox0 = tx * STEM_TW;
  const int iy_base = 2 * oy0 - 1, ix_base = 2 * ox0 - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long img = (long)b * H * W * 3;
  // ---- input window -> LDS with aligned dword loads; bytes outside the image row read as 0 (= padding)
  for (int i = tid; i < IN_ROWS * ROW_DW; i += 256) {
    const int rr = i / ROW_DW, dw = i - rr * ROW_DW;
    const int iy = iy_base + rr;
    uint32_t v = 0;
    const long row0 = img + (long)iy * W * 3;          // first byte of image row iy
    const long start = row0 + (long)ix_base * 3;       // first byte of the window (may precede the row)
    if (dw == 0) mis_s[rr] = (int)(start & 3L);
    if ((unsigned)iy < (unsigned)H) {
      const long addr = (start & ~3L) + dw * 4;
      const long hi = row0 + (long)W * 3;
      if (addr >= row0 && addr + 4 <= hi) {
        v = *reinterpret_cast<const uint32_t*>(in + addr);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (addr + k >= row0 && addr + k < hi) v |= (uint32_t)in[addr + k] << (8 * k);
      }
    }
    tile[rr][dw] = v;
  }
  __syncthreads();
  const uint8_t* tb = reinterpret_cast<const uint8_t*>(&tile[0][0]);
  const int r = lane & 15, q = lane >> 4;
  u32x4 wf[NTC];
  f32x4 sc[NTC], sh[NTC];
  // Channel -> MFMA row assignment: tiles are taken in pairs and row rho of tile j is channel
  // (j/2)*32 + (rho/4)*8 + (j%2)*4 + rho%4, so a lane ends up with 8 CONSECUTIVE channels of its pixel in the accumulators
  // of a tile pair -> one 16-byte store; a wave instruction then writes 16 pixels x 64 B = 1 KiB contiguous (the plain
  // assignment gave 8-byte stores in 32-B pieces).  A single tile (COUT 16) keeps the plain assignment.
  constexpr bool PAIR = (NTC % 2) == 0;
  auto chan = [&](int j, int rho) { return PAIR ? (j / 2) * 32 + (rho / 4) * 8 + (j % 2) * 4 + (rho % 4) : j * 16 + rho; };
#pragma unroll
  for (int j = 0; j < NTC; ++j) {
    wf[j] = *reinterpret_cast<const u32x4*>(wpad + chan(j, r) * 32 + q * 8);      // lane: MFMA row r of tile j, k-slice q
    sc[j] = *reinterpret_cast<const f32x4*>(scale + chan(j, q * 4));
    sh[j] = *reinterpret_cast<const f32x4*>(shift + chan(j, q * 4));
  }
  // this lane's k-slice: k = 8q+e -> tap (ky, kx), channel c_rgb; the stored byte is BGR (2 - c_rgb)
  int kky[8], kcol[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = q * 8 + e, tap = k / 3, c = k - tap * 3;
    kky[e] = k < 27 ? tap / 3 : -1;
    kcol[e] = (tap % 3) * 3 + (2 - c);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {                        // a wave: 2 output rows x 32 columns = 4 groups of 16 pixels
    const int orow = wave * 2 + (g >> 1), ocol = (g & 1) * 16 + r;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = 0.f;
      if (kky[e] >= 0) {
        const int rr = 2 * orow + kky[e];
        v = (float)tb[rr * ROWB + mis_s[rr] + 6 * ocol + kcol[e]] * (1.0f / 255.0f);
      }
      x[e] = v;
    }
    const u32x4 af = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7])};
    f32x4 y[NTC];
#pragma unroll
    for (int j = 0; j < NTC; ++j) {                    // D[channel q*4+reg][pixel r]
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]), __builtin_bit_cast(bf16x8, af), acc, 0, 0, 0);
      y[j] = acc * sc[j] + sh[j];
      y[j].x = siluf_(y[j].x); y[j].y = siluf_(y[j].y); y[j].z = siluf_(y[j].z); y[j].w = siluf_(y[j].w);
    }
    const int oy = oy0 + orow, ox = ox0 + ocol;
    if (oy < Ho && ox < Wo) {
      bf16_t* o = out + (((long)b * Ho + oy) * Wo + ox) * ldc;
      if constexpr (PAIR) {
#pragma unroll
        for (int j = 0; j < NTC; j += 2)
          *reinterpret_cast<u32x4*>(o + chan(j, q * 4)) = u32x4{pack_bf2(y[j].x, y[j].y), pack_bf2(y[j].z, y[j].w),
                                                               pack_bf2(y[j + 1].x, y[j + 1].y), pack_bf2(y[j + 1].z, y[j + 1].w)};
      } else {
#pragma unroll
        for (int j = 0; j < NTC; ++j) DT<bf16_t>::store4(o + j * 16 + q * 4, y[j]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Round 6: the stem of the split-fp16 engine (MOY_F32X3): fp32 output, uint8 input.  A pixel byte IS an fp16 value (0..255, exact), so
// only the weights need the split -- w = hi + lo * 2^-11, both fp16, [2][COUT][32] -- and a 16 x 16 output sub-tile is TWO products
// (bytes . hi into acc, bytes . lo into accx); y = SiLU((acc + accx * 2^-11) * (scale / 255) + shift) in fp32, i.e. the fp32 stem's
// function with the weights carried to 22 bits and the division by 255 applied after the sum instead of per pixel.  Same window /
// tile / channel assignment as stem_mfma_kernel; 16-byte stores of four consecutive channels.  Replaces the scalar-FMA fp32 stem
// (0.92 ms at 96 frames, fp32-FMA bound) in the f32x3 plan only -- the exact fp32 engine keeps u8 / 255 in fp32 as the reference has it.
constexpr int STEM_X3_TPB = 8;      // output tiles a block walks; the input windows of ALL of them are requested up front (4 dwords per thread each)

template <int COUT>
__global__ __launch_bounds__(256) void stem_x3_kernel(const uint8_t* __restrict__ in, int B, int H, int W,
                                                      const f16_t* __restrict__ wsplit,  // [2][COUT][32] fp16 in the k order below
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      float* __restrict__ out, int64_t ldc) {
  constexpr int NTC = COUT / 16;
  constexpr int IN_ROWS = 2 * STEM_TH + 1;
  constexpr int ROW_DW = ((2 * STEM_TW + 1) * 3 + 3 + 3) / 4 + 1;
  constexpr int ROWB = ROW_DW * 4;
  constexpr int NLD = (IN_ROWS * ROW_DW + 255) / 256;
  __shared__ uint32_t tile[IN_ROWS][ROW_DW];
  const int Ho = H >> 1, Wo = W >> 1;
  const int tiles_x = (Wo + STEM_TW - 1) / STEM_TW, tiles_y = (Ho + STEM_TH - 1) / STEM_TH;
  const int ntiles = B * tiles_y * tiles_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const long total = (long)B * H * W * 3;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(in), 0, (uint32_t)total, 0x00020000);     // (host: W % 4 == 0, total < 4 GiB)
  // The window of a tile: 17 rows x 51 dwords from the aligned dword at or below its first byte (rows are whole dwords, so every row of
  // a window has the same byte phase: (2 ox0 - 1) * 3 mod 4).  ALL of a thread's dwords are requested before the first is used
  // (range-checked buffer loads, no branch between them), and the requests of all eight tiles of a block are issued before the first is computed:
  // measured at 96 frames (lab build, tools/probes/stem_x3_time.py), one tile per block: 722 us; without the stores 293, without the
  // loads 388 -- the reads of a block wait behind the write stream, so they must be in flight a tile ahead.  Rows outside the image =
  // out-of-range offset = 0 = the padding; the bytes of a dword that precede the image row (left padding) are masked.
  auto coords = [&](int t, int& b, int& oy0, int& ox0) {
    const int tx = t % tiles_x; t /= tiles_x;
    oy0 = (t % tiles_y) * STEM_TH; ox0 = tx * STEM_TW; b = t / tiles_y;
  };
  uint32_t v[STEM_X3_TPB][NLD];
  auto request = [&](int t, auto nc) {
    constexpr int n_ = decltype(nc)::value;
    int b, oy0, ox0;
    coords(t, b, oy0, ox0);
    const long img = (long)b * H * W * 3;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 256;
      const int rr = i / ROW_DW, dw = i - rr * ROW_DW;
      const int iy = 2 * oy0 - 1 + rr;
      const long start = img + (long)iy * W * 3 + (long)(2 * ox0 - 1) * 3;
      const long addr = (start & ~3L) + dw * 4;
      const bool ok = t < ntiles && i < IN_ROWS * ROW_DW && (unsigned)iy < (unsigned)H && addr >= 0;
      v[n_][k] = __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (uint32_t)addr : 0xffffffffu, 0, 0);
    }
  };
  auto deposit = [&](int t, auto nc) {
    constexpr int n_ = decltype(nc)::value;
    int b, oy0, ox0;
    coords(t, b, oy0, ox0);
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 256;
      const int dw = i % ROW_DW;
      // left edge (ox0 == 0): the window starts 3 bytes before the row, i.e. byte 1 of the dword below it: that dword's bytes 1..3 are the padding
      const uint32_t m = (ox0 == 0 && dw == 0) ? 0u : 0xffffffffu;
      if (i < IN_ROWS * ROW_DW) (&tile[0][0])[i] = v[n_][k] & m;
    }
  };
  u32x4 wh[NTC], wl[NTC];
  f32x4 sc[NTC], sh[NTC];
#pragma unroll
  for (int j = 0; j < NTC; ++j) {                       // lane: MFMA row r of tile j (channel j*16 + r), k-slice q
    wh[j] = *reinterpret_cast<const u32x4*>(wsplit + (j * 16 + r) * 32 + q * 8);
    wl[j] = *reinterpret_cast<const u32x4*>(wsplit + (COUT + j * 16 + r) * 32 + q * 8);
    sc[j] = *reinterpret_cast<const f32x4*>(scale + j * 16 + q * 4) * (1.0f / 255.0f);
    sh[j] = *reinterpret_cast<const f32x4*>(shift + j * 16 + q * 4);
  }
  const uint8_t* tb = reinterpret_cast<const uint8_t*>(&tile[0][0]);
  const uint32_t* tw = &tile[0][0];
  const int t0 = blockIdx.x * STEM_X3_TPB;
  auto for_tiles = [&](auto&& f) { [&]<int... U>(std::integer_sequence<int, U...>) { (f(std::integral_constant<int, U>{}), ...); }(std::make_integer_sequence<int, STEM_X3_TPB>{}); };
  for_tiles([&](auto nc) { request(t0 + decltype(nc)::value, nc); });      // every window of the block in flight at once (beyond the last tile: nothing is read)
  for_tiles([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    const int t = t0 + n;
    if (t >= ntiles) return;                                               // block-uniform
    int b, oy0, ox0;
    coords(t, b, oy0, ox0);
    deposit(t, nc);
    __syncthreads();
    const int mis = (int)(((long)(2 * ox0 - 1) * 3) & 3L);   // byte phase of the window inside its first dword (row pitch and image base: whole dwords)
    // k order of a lane's slice (= moy_stem_l1_fused's, ops.stem_weights_x3): q < 3: the first EIGHT consecutive bytes of window row
    // ky = q (kx = e / 3, c_bgr = e % 3); q = 3: the ninth byte (kx 2, red) of the three rows, then zeros -- three dword reads + two
    // v_alignbyte (or three byte reads) per fragment instead of eight byte gathers
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int orow = wave * 2 + (g >> 1), ocol = (g & 1) * 16 + r;
      uint32_t lo, hi;
      if (q < 3) {
        const int off = (2 * orow + q) * ROWB + mis + 6 * ocol;
        const uint32_t d0 = tw[off >> 2], d1 = tw[(off >> 2) + 1], d2 = tw[(off >> 2) + 2];
        lo = __builtin_amdgcn_alignbyte(d1, d0, off & 3);
        hi = __builtin_amdgcn_alignbyte(d2, d1, off & 3);
      } else {
        uint32_t b3[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) b3[e] = tb[(2 * orow + e) * ROWB + mis + 6 * ocol + 8];
        lo = b3[0] | (b3[1] << 8) | (b3[2] << 16);
        hi = 0;
      }
      // the bytes as fp16: 0x6400 | x is 1024 + x, minus 1024 is exact (one v_perm_b32 + one v_pk_add_f16 per pair)
      typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
      auto pair = [&](uint32_t w4, uint32_t sel) {
        const h2_ big = __builtin_bit_cast(h2_, __builtin_amdgcn_perm(0x64646464u, w4, sel));
        return __builtin_bit_cast(uint32_t, big - h2_{(_Float16)1024.0f, (_Float16)1024.0f});
      };
      const u32x4 af = {pair(lo, 0x04010400u), pair(lo, 0x04030402u), pair(hi, 0x04010400u), pair(hi, 0x04030402u)};
      const int oy = oy0 + orow, ox = ox0 + ocol;
      float* o = out + (((long)b * Ho + oy) * Wo + ox) * ldc;
#pragma unroll
      for (int j = 0; j < NTC; ++j) {                    // D[channel q*4+reg][pixel r]
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accx = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[j]), __builtin_bit_cast(f16x8, af), acc, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[j]), __builtin_bit_cast(f16x8, af), accx, 0, 0, 0);
        f32x4 y = (acc + accx * (1.0f / 2048.0f)) * sc[j] + sh[j];
        y.x = siluf_(y.x); y.y = siluf_(y.y); y.z = siluf_(y.z); y.w = siluf_(y.w);
        if (oy < Ho && ox < Wo) *reinterpret_cast<f32x4*>(o + j * 16 + q * 4) = y;
      }
    }
    __syncthreads();                                      // every wave has read the window before the next one overwrites it
  });
}

// ------------------------------------------------------------------------------------------------
// SPPF: y1 = pool5(x), y2 = pool5(y1), y3 = pool5(y2) (stride 1, pad 2, -inf padding), exactly the
// cascade of block.py:129-134.  One block = one (frame, 16-byte channel chunk): the H x W plane of
// that chunk lives in LDS and each 5x5 pool is a separable row pass + column pass (5 + 5 reads per
// pixel instead of the 169 taps of a direct 13x13 window).
template <typename T>
__device__ __forceinline__ u32x4 max_chunk(u32x4 a, u32x4 b);
template <>
__device__ __forceinline__ u32x4 max_chunk<float>(u32x4 a, u32x4 b) {
  return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(f32x4, a), __builtin_bit_cast(f32x4, b)));
}
// 16-bit planes are pooled as PACKED values, one VALU operation per pair: fp16 with v_pk_max_f16; bf16 (no packed bf16 maximum on
// gfx950) as order-preserving int16 keys -- key = bits ^ 0x7fff for negative values (an involution) -- with v_pk_max_i16: encoded once
// at the load, decoded once per store.  (Unpacking to fp32, fmaxf and repacking cost 28 operations per 16-byte chunk and tap: the
// kernel was bound by them.)
typedef short s16x2_ __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_key2(uint32_t w) { return w ^ (((w >> 15) & 0x00010001u) * 0x7fffu); }
template <typename T>
__device__ __forceinline__ u32x4 pool_encode(u32x4 v) { return v; }
template <>
__device__ __forceinline__ u32x4 pool_encode<bf16_t>(u32x4 v) { return u32x4{bf16_key2(v.x), bf16_key2(v.y), bf16_key2(v.z), bf16_key2(v.w)}; }
template <typename T>
__device__ __forceinline__ u32x4 pool_decode(u32x4 v) { return pool_encode<T>(v); }
template <>
__device__ __forceinline__ u32x4 max_chunk<bf16_t>(u32x4 a, u32x4 b) {     // on keys
  auto mx = [](uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_, x), __builtin_bit_cast(s16x2_, y)));
  };
  return u32x4{mx(a.x, b.x), mx(a.y, b.y), mx(a.z, b.z), mx(a.w, b.w)};
}
template <>
__device__ __forceinline__ u32x4 max_chunk<f16_t>(u32x4 a, u32x4 b) {
  auto mx = [](uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h16x2_, x), __builtin_bit_cast(h16x2_, y)));
  };
  return u32x4{mx(a.x, b.x), mx(a.y, b.y), mx(a.z, b.z), mx(a.w, b.w)};
}

// One block = one frame x G adjacent 16-byte channel chunks (G = 4 when the channel count allows): consecutive threads take the G
// chunks of one pixel, so every global access of a wave is 16 runs of 64 contiguous bytes instead of 64 scattered 16-byte pieces --
// with one chunk per block the kernel spent its time on 12 M separate 16-byte requests (the lesson of the deformable gather).
template <typename T>
__global__ __launch_bounds__(512) void sppf_pool_kernel(const T* __restrict__ x, int64_t ldx, int H, int W, int C, int G,
                                                        T* __restrict__ y1, T* __restrict__ y2, T* __restrict__ y3,
                                                        int64_t ldy) {
  constexpr int KPB = DT<T>::KPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  const int npx = H * W, nit = npx * G, tid = threadIdx.x;
  u32x4* P = reinterpret_cast<u32x4*>(dyn);            // [G][H*W] current planes
  u32x4* Q = P + nit;                                  // [G][H*W] row-pass result
  const int gpr = C / (KPB * G);                       // chunk groups per pixel
  const int b = blockIdx.x / gpr, cg = blockIdx.x % gpr;
  const int gsh = G == 4 ? 2 : (G == 2 ? 1 : 0);
  const long pix0 = (long)b * npx;
  const int c0 = cg * G * KPB;
  for (int it = tid; it < nit; it += 512) {
    const int px = it >> gsh, g = it & (G - 1);
    P[g * npx + px] = pool_encode<T>(*reinterpret_cast<const u32x4*>(x + (pix0 + px) * ldx + c0 + g * KPB));
  }
  __syncthreads();
  T* outs[3] = {y1, y2, y3};
#pragma unroll 1
  for (int s = 0; s < 3; ++s) {
    for (int it = tid; it < nit; it += 512) {          // row pass
      const int px = it >> gsh, i = (it & (G - 1)) * npx + px;
      const int yy = px / W, xx = px - yy * W;
      u32x4 m = P[i];
#pragma unroll
      for (int d = -2; d <= 2; ++d)
        if (d != 0 && (unsigned)(xx + d) < (unsigned)W) m = max_chunk<T>(m, P[i + d]);
      Q[i] = m;
    }
    __syncthreads();
    for (int it = tid; it < nit; it += 512) {          // column pass + store
      const int px = it >> gsh, g = it & (G - 1), i = g * npx + px;
      const int yy = px / W;
      u32x4 m = Q[i];
#pragma unroll
      for (int d = -2; d <= 2; ++d)
        if (d != 0 && (unsigned)(yy + d) < (unsigned)H) m = max_chunk<T>(m, Q[i + d * W]);
      P[i] = m;
      *reinterpret_cast<u32x4*>(outs[s] + (pix0 + px) * ldy + c0 + g * KPB) = pool_decode<T>(m);
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* __restrict__ x, int64_t ldx, int B, int H, int W, int C,
                                                         T* __restrict__ y, int64_t ldy) {
  constexpr int KPB = DT<T>::KPB;
  const int cpr = C / KPB;
  const long total = (long)B * 4 * H * W * cpr;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int cc = (int)(t % cpr);
  const long pix = t / cpr;
  const int Wo = 2 * W, Ho = 2 * H;
  const int b = (int)(pix / ((long)Ho * Wo));
  const int rem = (int)(pix - (long)b * Ho * Wo);
  const int oy = rem / Wo, ox = rem - oy * Wo;
  const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((long)b * H + (oy >> 1)) * W + (ox >> 1)) * ldx + cc * KPB);
  *reinterpret_cast<u32x4*>(y + pix * ldy + cc * KPB) = v;
}

// ------------------------------------------------------------------------------------------------
// Narrow heads: one wave per row, N <= 8 dot products reduced with shuffles.
__device__ __forceinline__ float inv_sigmoid(float x) {   // nn/modules/utils.py:34-38, eps 1e-5
  x = fminf(fmaxf(x, 0.f), 1.f);
  return logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
}

template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ X, int64_t ldx, const int32_t* __restrict__ x_rows,
                                                     int M, int K, const float* __restrict__ Wt,
                                                     const float* __restrict__ bias, int N, int mode,
                                                     const float* __restrict__ aux, const int32_t* __restrict__ aux_rows,
                                                     float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long row = x_rows ? (long)x_rows[m] : (long)m;
  const T* xr = X + row * ldx;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const f32x4 xv = DT<T>::load4(xr + k);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < N) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(Wt + (long)j * K + k);
        acc[j] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
      }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < N) acc[j] = wave_sum(acc[j]);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < N) {
        float v = acc[j] + (bias ? bias[j] : 0.f);
        if (mode == 1) v = sigmoidf_(v + inv_sigmoid(aux[(long)m * N + j]));
        else if (mode == 2) v = v + aux[(long)aux_rows[m] * N + j];
        y[(long)m * N + j] = v;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Query selection: exact radix select of the nq largest scores + bitonic sort of the selection.
__device__ __forceinline__ uint32_t order_key(float f) {   // monotone: larger float -> larger key
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float score_max(const float* __restrict__ s, int nc) {
  float v = s[0];
  for (int c = 1; c < nc; ++c) v = fmaxf(v, s[c]);   // torch .max(-1): NaN-free fixtures
  return v;
}

constexpr int TOPK_THREADS = 1024;

__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const float* __restrict__ scores, int S, int nc, int nq, int P2,
                                                            const uint8_t* __restrict__ valid,
                                                            int32_t* __restrict__ idx_local, int32_t* __restrict__ idx_global,
                                                            int32_t* __restrict__ n_masked) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh_prefix, sh_remaining, sh_cnt, sh_masked, wtot[4];
  __shared__ uint32_t scan[TOPK_THREADS];
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(dyn);   // [P2]

  const int b = blockIdx.x, tid = threadIdx.x;
  const float* sc = scores + (long)b * S * nc;

  if (tid == 0) { sh_prefix = 0; sh_remaining = nq; sh_cnt = 0; sh_masked = 0; }
  uint32_t mask = 0;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = sh_prefix;
    for (int i = tid; i < S; i += TOPK_THREADS) {
      const uint32_t u = order_key(score_max(sc + (long)i * nc, nc));
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1u);
    }
    __syncthreads();
    // bucket d = the highest one whose cumulative count (from bucket 255 down) reaches the number still needed: a 256-wide scan by
    // four waves (one thread walking the buckets cost 256 dependent LDS reads per pass -- half of the kernel's time)
    uint32_t cnt = 0, incl = 0;
    const uint32_t rem = sh_remaining;
    if (tid < 256) {
      cnt = hist[255 - tid];                 // position tid <-> bucket 255 - tid
      incl = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
        if ((tid & 63) >= off) incl += up;
      }
      if ((tid & 63) == 63) wtot[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < 256) {
      for (int w = 0; w < (tid >> 6); ++w) incl += wtot[w];
      const uint32_t excl = incl - cnt;
      if (incl >= rem && excl < rem) {       // exactly one position: the counts of the matching elements sum to >= rem
        sh_remaining = rem - excl;           // still needed from bucket d
        sh_prefix = prefix | ((uint32_t)(255 - tid) << shift);
      }
    }
    mask |= 255u << shift;
    __syncthreads();
  }
  const uint32_t thr = sh_prefix;        // key of the nq-th largest value
  const uint32_t need_eq = sh_remaining; // how many elements equal to thr are taken (lowest index first)

  // ties: ordered ranks over contiguous index ranges
  const int per = (S + TOPK_THREADS - 1) / TOPK_THREADS;
  const int i0 = tid * per, i1 = min(S, i0 + per);
  uint32_t my_eq = 0;
  for (int i = i0; i < i1; ++i) my_eq += order_key(score_max(sc + (long)i * nc, nc)) == thr;
  scan[tid] = my_eq;
  __syncthreads();
  for (int off = 1; off < TOPK_THREADS; off <<= 1) {   // Hillis-Steele inclusive scan
    const uint32_t v = tid >= off ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  uint32_t eq_rank = scan[tid] - my_eq;
  for (int i = tid; i < P2; i += TOPK_THREADS) keys[i] = ~0ull;   // padding sorts last
  __syncthreads();
  for (int i = i0; i < i1; ++i) {
    const uint32_t u = order_key(score_max(sc + (long)i * nc, nc));
    bool take = u > thr;
    if (u == thr) { take = eq_rank < need_eq; ++eq_rank; }
    if (take) {
      const uint32_t pos = atomicAdd(&sh_cnt, 1u);
      keys[pos] = ((unsigned long long)(~u) << 32) | (uint32_t)i;   // ascending = value desc, index asc
    }
  }
  __syncthreads();
  // bitonic sort of P2 64-bit keys
  for (int k = 2; k <= P2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P2; i += TOPK_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], c = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < nq; i += TOPK_THREADS) {
    const int id = (int)(keys[i] & 0xffffffffu);
    idx_local[(long)b * nq + i] = id;
    idx_global[(long)b * nq + i] = b * S + id;
    if (valid && valid[id] == 0) atomicAdd(&sh_masked, 1u);
  }
  __syncthreads();
  if (tid == 0 && n_masked) n_masked[b] = (int)sh_masked;
}

// ------------------------------------------------------------------------------------------------
// pos2posemb (transformer.py:183-190): out[m, c*64 + 2i] = sin(p / t_i), [.. + 2i+1] = cos(p / t_i),
// p = pos[m, c] * 2*pi, t_i = 10000^(2i/64).
template <typename T>
__global__ __launch_bounds__(256) void posemb_kernel(const float* __restrict__ pos, int M, T* __restrict__ out, int64_t ldo) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;   // one thread per (m, c, i): 128 per row
  if (t >= (long)M * 128) return;
  const int m = (int)(t >> 7), ci = (int)(t & 127), c = ci >> 5, i = ci & 31;
  T* o = out + (long)m * ldo + c * 64 + 2 * i;
  if constexpr (sizeof(T) == 2) {
    // 16-bit output: hardware exp2 / sin / cos (v_sin_f32 takes revolutions: p / t_i / 2 pi = pos / t_i, |.| <= 1) -- their
    // ~1e-6 absolute error is three orders below an ulp of T; the pair leaves as one 4-byte store
    const float rev = pos[(long)m * 4 + c] * __builtin_amdgcn_exp2f(-(float)i * 0.41524101186092029f);   // 10000^(-2i/64)
    *reinterpret_cast<uint32_t*>(o) = DT<T>::pack2(__builtin_amdgcn_sinf(rev), __builtin_amdgcn_cosf(rev));
  } else {
    const float p = pos[(long)m * 4 + c] * 6.283185307179586f;
    const float dim_t = powf(10000.0f, (float)(2 * i) / 64.0f);
    const float a = p / dim_t;
    DT<T>::store1(o, sinf(a));
    DT<T>::store1(o + 1, cosf(a));
  }
}

// ------------------------------------------------------------------------------------------------
// Self-attention core, head dim 32, L <= 512 keys, on the matrix cores.
// Block = (b, head, slice of 16-query tiles); K ([key][32], 64-B panels, swizzled like the GEMM
// tiles) and V transposed ([d][key]) of the (b, head) pair are staged in LDS once per block.
// Per wave and 16-query tile:
//   S^T = K . Q^T   (MFMA, keys on the accumulator rows, the query on the lane -> each lane owns
//                    4 keys per 16-key tile of ONE query)
//   softmax over keys: in-lane over the tiles + two xor-shuffles (lanes r, r+16, r+32, r+48)
//   O^T = V^T . P^T (MFMA; P goes from the accumulator layout straight into the B operand: the
//                    key order inside a k-step is permuted identically for V^T's fragment)
// bf16: v_mfma_f32_16x16x32_bf16 (P rounded to bf16); f32: v_mfma_f32_16x16x4_f32 (exact fp32).
__device__ __forceinline__ int swz16(int row, int q) { return q ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3); }

template <typename T>
__device__ __forceinline__ void mma16(f32x4& acc, u32x4 afrag, u32x4 bfrag);
template <>
__device__ __forceinline__ void mma16<bf16_t>(f32x4& acc, u32x4 afrag, u32x4 bfrag) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afrag), __builtin_bit_cast(bf16x8, bfrag), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<f16_t>(f32x4& acc, u32x4 afrag, u32x4 bfrag) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, afrag), __builtin_bit_cast(f16x8, bfrag), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<float>(f32x4& acc, u32x4 afrag, u32x4 bfrag) {
  const f32x4 a = __builtin_bit_cast(f32x4, afrag), b = __builtin_bit_cast(f32x4, bfrag);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
}

// Latency structure (the first version took 58 us per layer for ~3 us of MFMA work: a rolled staging loop with one
// load -> wait -> LDS-write round trip per iteration, a wave-uniform branch with a full s_waitcnt around every key tile,
// Q loaded inside the tile loop): every loop below has a compile-time trip count -- the LDS image always holds NKT*16
// keys (zero rows past L, masked to -inf), so the key loops are branch-free and the compiler keeps the fragment reads in
// flight under the MFMAs; all K/V staging loads and the wave's Q fragments are issued before anything is waited for.
template <typename T, int NKT, bool MASKED>   // NKT (even): 16-key tiles held in LDS and in the score registers
// (launch bound: two blocks per CU = a budget of 256 registers per wave.  With the default budget of 512 hipcc put the MFMA results -- the
//  80 score registers -- into AccVGPRs and fetched every one of them with a v_accvgpr_read before the softmax could touch it: 440 extra
//  vector instructions per block beside 400 v_exp_f32, in a kernel that is bound by exactly that vector work (round 5).  The fp32 form
//  with 32 key tiles needs 254 registers: it keeps the wide budget.)
__global__ __launch_bounds__(256, (sizeof(T) == 4 && NKT > 20) ? 1 : 2) void mha_kernel(const T* __restrict__ qkv, int64_t ld, int L, int nh, int E,
                                                  T* __restrict__ out, int64_t ldo, int tiles_per_block,
                                                  const int32_t* __restrict__ n_prefix, int split) {
  constexpr int KPB = DT<T>::KPB;
  constexpr int NPD = 32 / (4 * KPB);             // 64-B panels per K row: bf16 1, f32 2
  constexpr int ESZ = 16 / KPB;
  constexpr bool BF = KPB == 8;
  constexpr int Lp = NKT * 16;                    // keys in the LDS image
  constexpr int vstride = Lp * ESZ + 16;          // bytes per V^T row (pad breaks the power-of-two stride)
  constexpr int CPK = 4 * NPD;                    // 16-B chunks per key row
  constexpr int NST = (Lp * CPK + 255) / 256;     // staging chunks per thread
  constexpr int MAXQT = (NKT + 3) / 4;            // 16-query tiles per wave (tiles_per_block <= NKT)
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  unsigned char* Ks = dyn;                         // [NPD][Lp][64 B]
  unsigned char* Vt = dyn + NPD * Lp * 64;         // [32][vstride]
  float* kbias = reinterpret_cast<float*>(Vt + 32 * vstride);   // [Lp]: 0 for a live key, -inf otherwise
  const int b = blockIdx.x / nh, h = blockIdx.x % nh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q4 = lane >> 4;
  const T* base = qkv + (long)b * L * ld;
  // key mask of the temporal mode: key j takes part iff j < n_prefix[b] (live track slots) or j >= split (detect queries).
  // The mask enters as the MFMA's C operand (score = -inf + q.k for a dead key or a zero row past L): no VALU in the loops.
  const int npre = MASKED ? n_prefix[b] : L;
  for (int j = tid; j < Lp; j += 256) {
    const bool lv = MASKED ? (j < L && (j < npre || j >= split)) : j < L;
    kbias[j] = lv ? 0.0f : -INFINITY;
  }
  const int t_begin = blockIdx.y * tiles_per_block;
  const int nqt = (L + 15) >> 4;
  const int t_end = min(nqt, t_begin + tiles_per_block);

  // ---- all global loads first: K / V chunks of this thread, Q fragments of this wave's tiles
  u32x4 kreg[NST], vreg[NST];
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int i = tid + k * 256;
    const int key = min(i / CPK, L - 1), ch = i % CPK;       // clamped: always a valid address; rows >= L are zeroed below
    kreg[k] = *reinterpret_cast<const u32x4*>(base + (long)key * ld + E + h * 32 + ch * KPB);
    vreg[k] = *reinterpret_cast<const u32x4*>(base + (long)key * ld + 2 * E + h * 32 + ch * KPB);
  }
  u32x4 qf[MAXQT][NPD];
#pragma unroll
  for (int t = 0; t < MAXQT; ++t) {
    const int qld = min((t_begin + wave + 4 * t) * 16 + r, L - 1);
#pragma unroll
    for (int pd = 0; pd < NPD; ++pd) qf[t][pd] = *reinterpret_cast<const u32x4*>(base + (long)qld * ld + h * 32 + (pd * 4 + q4) * KPB);
  }
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int i = tid + k * 256;
    const int key = i / CPK, ch = i % CPK;
    if (i < Lp * CPK) {
      u32x4 kv = kreg[k], vv = vreg[k];
      if (key >= L) { kv = u32x4{0u, 0u, 0u, 0u}; vv = kv; }
      const int pd = ch >> 2, c4 = ch & 3;
      *reinterpret_cast<u32x4*>(Ks + (pd * Lp + key) * 64 + swz16(key, c4) * 16) = kv;
      const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
      if constexpr (BF) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint16_t hv = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
          *reinterpret_cast<uint16_t*>(Vt + (ch * 8 + e) * vstride + key * 2) = hv;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<uint32_t*>(Vt + (ch * 4 + e) * vstride + key * 4) = w[e];
      }
    }
  }
  __syncthreads();
  const float scaling = 0.17677669529663687f;     // 32^-0.5 (torch MHA scales q before QK^T)
  constexpr float L2E = 1.4426950408889634f;
  const float cexp = scaling * L2E;               // softmax(x * scaling) = exp2((x - max) * cexp) / sum
#pragma unroll
  for (int t = 0; t < MAXQT; ++t) {
    const int qt = t_begin + wave + 4 * t;
    if (qt >= t_end) break;                        // wave-uniform
    const int query = qt * 16 + r;
    f32x4 s[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 acc = *reinterpret_cast<const f32x4*>(kbias + kt * 16 + q4 * 4);
      const int krow = kt * 16 + r;
#pragma unroll
      for (int pd = 0; pd < NPD; ++pd) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + (pd * Lp + krow) * 64 + swz16(krow, q4) * 16);
        mma16<T>(acc, kf, qf[t][pd]);              // D[key = q4*4+reg][query = r]
      }
      s[kt] = acc;                                 // raw q.k (the 32^-0.5 scaling is folded into the exponent below)
      mx = fmaxf(fmaxf(mx, acc.x), acc.y);         // v_max3_f32
      mx = fmaxf(fmaxf(mx, acc.z), acc.w);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (mx == -INFINITY) mx = 0.f;                  // no live key (empty track memory): all weights 0, output 0
    float sum = 0.f;
    const float nmx = -mx * cexp;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 e;
      // one fma + v_exp_f32 (1 ulp) per score: libm expf cost ~40 VALU ops per score and dominated the first version
      e.x = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt].x, cexp, nmx)); e.y = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt].y, cexp, nmx));
      e.z = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt].z, cexp, nmx)); e.w = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt].w, cexp, nmx));
      s[kt] = e;                                  // exp(-inf) = 0 for masked keys and the zero rows past L
      sum += (e.x + e.y) + (e.z + e.w);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if constexpr (BF) {
#pragma unroll
      for (int kb = 0; kb < NKT / 2; ++kb) {
        const f32x4 p0 = s[2 * kb], p1 = s[2 * kb + 1];
        const u32x4 pf = {DT<T>::pack2(p0.x, p0.y), DT<T>::pack2(p0.z, p0.w), DT<T>::pack2(p1.x, p1.y), DT<T>::pack2(p1.z, p1.w)};
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
          const unsigned char* vr = Vt + (dh * 16 + r) * vstride + (kb * 32 + q4 * 4) * 2;
          const u32x2 v0 = *reinterpret_cast<const u32x2*>(vr), v1 = *reinterpret_cast<const u32x2*>(vr + 32);
          mma16<T>(o[dh], u32x4{v0.x, v0.y, v1.x, v1.y}, pf);   // D[d = q4*4+reg][query = r]
        }
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const u32x4 pf = __builtin_bit_cast(u32x4, s[kt]);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
          const u32x4 vf = *reinterpret_cast<const u32x4*>(Vt + (dh * 16 + r) * vstride + (kt * 16 + q4 * 4) * 4);
          mma16<T>(o[dh], vf, pf);
        }
      }
    }
    if (query < L) {
      const float inv = sum > 0.f ? 1.0f / sum : 0.f;
      T* op = out + ((long)b * L + query) * ldo + h * 32 + q4 * 4;
      DT<T>::store4(op, o[0] * inv);
      DT<T>::store4(op + 16, o[1] * inv);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Deformable attention, decoder form (M = 8 heads x D = 32, P = 4, L <= 4): one wave per query;
// lane = head*8 + sub owns channels sub*4..+3 of its head.  Softmax over the L*P logits and the
// sampling locations are recomputed per lane (12 exps), taps are 8/16-byte loads of contiguous
// channels, accumulation in fp32.
struct LevelInfo {
  int H[4], W[4], start[4];
};

template <typename T>
__global__ __launch_bounds__(256) void msda_fused_kernel(const T* __restrict__ value, int64_t ldv, int64_t head_stride, int S, LevelInfo lv, int L,
                                                         const float* __restrict__ offaw, int64_t ld_oa,
                                                         const float* __restrict__ ref, int Lq, int nrows,
                                                         T* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int b = row / Lq;
  const int m = lane >> 3, sub = lane & 7;
  const int LP = L * 4;
  const float* oa = offaw + (long)row * ld_oa;
  const float* offp = oa + m * LP * 2;
  const float* awp = oa + 8 * LP * 2 + m * LP;
  float logit[16], offx[16], offy[16];
#pragma unroll
  for (int i = 0; i < 16; i += 4)
    if (i < LP) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(awp + i);
      logit[i] = a.x; logit[i + 1] = a.y; logit[i + 2] = a.z; logit[i + 3] = a.w;
      const f32x4 o0 = *reinterpret_cast<const f32x4*>(offp + 2 * i), o1 = *reinterpret_cast<const f32x4*>(offp + 2 * i + 4);
      offx[i] = o0.x; offy[i] = o0.y; offx[i + 1] = o0.z; offy[i + 1] = o0.w;
      offx[i + 2] = o1.x; offy[i + 2] = o1.y; offx[i + 3] = o1.z; offy[i + 3] = o1.w;
    }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < LP) mx = fmaxf(mx, logit[i]);
  float den = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < LP) { logit[i] = expf(logit[i] - mx); den += logit[i]; }
  const float inv_den = 1.0f / den;
  const f32x4 rb = *reinterpret_cast<const f32x4*>(ref + (long)row * 4);
  const T* vb = value + (long)b * S * ldv + m * head_stride + sub * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < 4; ++l)
    if (l < L) {
      const int H = lv.H[l], W = lv.W[l];
      const T* vl = vb + (long)lv.start[l] * ldv;
#pragma unroll
      for (int pnt = 0; pnt < 4; ++pnt) {
        const int i = l * 4 + pnt;
        // loc = ref_xy + off / n_points * ref_wh * 0.5   (transformer.py:280-282)
        const float lx = rb.x + offx[i] / 4.0f * rb.z * 0.5f;
        const float ly = rb.y + offy[i] / 4.0f * rb.w * 0.5f;
        const float x = lx * W - 0.5f, y = ly * H - 0.5f;
        const float aw = logit[i] * inv_den;
        const float xf = floorf(x), yf = floorf(y);
        const float fx = x - xf, fy = y - yf;
        const int x0 = (int)xf, y0 = (int)yf;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
          const float wgt = ((t & 1) ? fx : 1.f - fx) * ((t >> 1) ? fy : 1.f - fy) * aw;
          if ((unsigned)xi < (unsigned)W && (unsigned)yi < (unsigned)H)
            acc += DT<T>::load4(vl + ((long)yi * W + xi) * ldv) * wgt;
        }
      }
    }
  DT<T>::store4(out + (long)row * ldo + m * 32 + sub * 4, acc);
}

// Head-plane form of the kernel above for 16-bit values (value[(b*S + s)*32 + m*head_stride + d]: the 32 channels of a head are
// 64 contiguous bytes and the tokens x0, x0 + 1 of a bilinear sample are 128 CONTIGUOUS bytes).  The gather is bound by the number
// of random accesses, not by bytes, so the two x-corners of a sample row are fetched by ONE instruction:
// lane = head*8 + sub, sub = corner_x*4 + channel octet -- 16 bytes per lane, 128 contiguous bytes per head: 24 loads of 16 bytes
// per lane instead of 48 of 8, half as many requests.  Every lane accumulates its own corner's share of its 8 channels (the
// sum over corners is linear); the two corner lanes are added once at the end (one xor-4 shuffle per channel).  Addresses are
// 32-bit offsets into a per-frame buffer descriptor (head_stride * 8 * 2 bytes < 2 GiB, checked by the host); a corner outside
// the level is an out-of-range offset (the hardware returns zeros): no branch around any load, all 24 in flight at once.
template <typename T, int MODE = 1>     // taps in flight per wave: 0 all levels (209 VGPRs, 2 waves/SIMD), 1 one level (114, 4 waves/SIMD: the default), 2 two levels; MOY_MSDA_PLANES=2 / 1 / 3
__global__ __launch_bounds__(256) void msda_planes_kernel(const T* __restrict__ value, int64_t head_stride, int S, LevelInfo lv, int L,
                                                          const float* __restrict__ offaw, int64_t ld_oa,
                                                          const float* __restrict__ ref, int Lq, int nrows,
                                                          T* __restrict__ out, int64_t ldo) {
  static_assert(sizeof(T) == 2, "16-bit values");
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int b = __builtin_amdgcn_readfirstlane(row / Lq);
  const int m = lane >> 3, sub = lane & 7, cx = sub >> 2, oct = sub & 3;
  const int LP = L * 4;
  const float* oa = offaw + (long)row * ld_oa;
  const float* offp = oa + m * LP * 2;
  const float* awp = oa + 8 * LP * 2 + m * LP;
  float logit[16], offx[16], offy[16];
#pragma unroll
  for (int i = 0; i < 16; i += 4)
    if (i < LP) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(awp + i);
      logit[i] = a.x; logit[i + 1] = a.y; logit[i + 2] = a.z; logit[i + 3] = a.w;
      const f32x4 o0 = *reinterpret_cast<const f32x4*>(offp + 2 * i), o1 = *reinterpret_cast<const f32x4*>(offp + 2 * i + 4);
      offx[i] = o0.x; offy[i] = o0.y; offx[i + 1] = o0.z; offy[i + 1] = o0.w;
      offx[i + 2] = o1.x; offy[i + 2] = o1.y; offx[i + 3] = o1.z; offy[i + 3] = o1.w;
    }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < LP) mx = fmaxf(mx, logit[i]);
  float den = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < LP) { logit[i] = __builtin_amdgcn_exp2f((logit[i] - mx) * 1.4426950408889634f); den += logit[i]; }   // (v_exp_f32 / v_rcp_f32: an ulp
  const float inv_den = __builtin_amdgcn_rcpf(den);                                                             // of fp32 on weights that multiply 16-bit values)
  const f32x4 rb = *reinterpret_cast<const f32x4*>(ref + (long)row * 4);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(value + (int64_t)b * S * 32), 0, 0x80000000u, 0x00020000);
  const uint32_t lane_off = (uint32_t)((m * head_stride + oct * 8) * 2);
  constexpr uint32_t OOB = 0x80000000u;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto level_taps = [&](int l, u32x4 (&tap)[4][2], float (&wgt)[4][2]) {
    const int H = lv.H[l], W = lv.W[l];
#pragma unroll
    for (int pnt = 0; pnt < 4; ++pnt) {
      const int i = l * 4 + pnt;
      // loc = ref_xy + off / n_points * ref_wh * 0.5   (transformer.py:280-282)
      const float lx = rb.x + offx[i] / 4.0f * rb.z * 0.5f;
      const float ly = rb.y + offy[i] / 4.0f * rb.w * 0.5f;
      const float x = lx * W - 0.5f, y = ly * H - 0.5f;
      const float aw = logit[i] * inv_den;
      const float xf = floorf(x), yf = floorf(y);
      const float fx = x - xf, fy = y - yf;
      const int xi = (int)xf + cx, y0 = (int)yf;
      const float wx = (cx ? fx : 1.f - fx) * aw;
      const bool xin = (unsigned)xi < (unsigned)W;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int yi = y0 + t;
        const bool ok = xin && (unsigned)yi < (unsigned)H;
        const uint32_t off = lane_off + (uint32_t)((lv.start[l] + yi * W + xi) * 64);
        tap[pnt][t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0));
        wgt[pnt][t] = wx * (t ? fy : 1.f - fy);
      }
    }
  };
  auto level_acc = [&](const u32x4 (&tap)[4][2], const float (&wgt)[4][2]) {
#pragma unroll
    for (int pnt = 0; pnt < 4; ++pnt)
#pragma unroll
      for (int t = 0; t < 2; ++t) {                      // same order of the four corners per sample as the kernel above
        const u32x4 w = tap[pnt][t];
        const float g = wgt[pnt][t];
        acc[0] = DT<T>::fma_lo(w.x, g, acc[0]); acc[1] = DT<T>::fma_hi(w.x, g, acc[1]); acc[2] = DT<T>::fma_lo(w.y, g, acc[2]); acc[3] = DT<T>::fma_hi(w.y, g, acc[3]);
        acc[4] = DT<T>::fma_lo(w.z, g, acc[4]); acc[5] = DT<T>::fma_hi(w.z, g, acc[5]); acc[6] = DT<T>::fma_lo(w.w, g, acc[6]); acc[7] = DT<T>::fma_hi(w.w, g, acc[7]);
      }
  };
  if constexpr (MODE == 0) {                              // all levels' taps in flight at once
    u32x4 tap[4][4][2];
    float wgt[4][4][2];
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (l < L) level_taps(l, tap[l], wgt[l]);
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (l < L) level_acc(tap[l], wgt[l]);
  } else if constexpr (MODE == 1) {                       // level by level
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (l < L) {
        u32x4 tap[4][2];
        float wgt[4][2];
        level_taps(l, tap, wgt);
        level_acc(tap, wgt);
      }
  } else {                                                // two levels in flight: level l+1's taps issued before level l is summed
    u32x4 tap[2][4][2];
    float wgt[2][4][2];
    level_taps(0, tap[0], wgt[0]);
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (l < L) {
        if (l + 1 < L) level_taps(l + 1, tap[(l + 1) & 1], wgt[(l + 1) & 1]);
        level_acc(tap[l & 1], wgt[l & 1]);
      }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], 4);
  if (cx == 0)
    *reinterpret_cast<u32x4*>(out + (long)row * ldo + m * 32 + oct * 8) =
        u32x4{DT<T>::pack2(acc[0], acc[1]), DT<T>::pack2(acc[2], acc[3]), DT<T>::pack2(acc[4], acc[5]), DT<T>::pack2(acc[6], acc[7])};
}

// ------------------------------------------------------------------------------------------------
// Stretch resize of uint8 HWC frames = cv2.resize(img, (Wd, Hd), interpolation=cv2.INTER_LINEAR), the only thing
// LetterBox does on the tracking path (scaleFill branch, data/augment.py:573-576, called from
// MOTRtrack/predict.py:96-105).  cv2 is a third-party dependency absent from the reference tree (requirements.txt:
// opencv-python>=4.6.0); this restates its published 8-bit algorithm (imgproc/resize.cpp: HResizeLinear<uchar,int,short,2048>
// + VResizeLinear<uchar,...,FixedPtCast<int,uchar,22>>): 11-bit fixed-point taps, horizontal pass in int32, vertical pass
// ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2; an exact 2x2 shrink is rerouted to INTER_AREA ((sum+2)>>2).
// One thread makes 4 output pixels (12 bytes = 3 dword stores); HBM-bound: reads the source once, writes the target once.
__device__ __forceinline__ void lin_tap(int d, double scale, int n_src, int& s0, int& s1, int& a0, int& a1, bool is_x) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (is_x) {                                         // resize.cpp: x taps are clamped with the weight forced to 0
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
    s0 = s;
    s1 = min(s + 1, n_src - 1);
  } else {                                            // y taps keep their weights, rows are clipped
    s0 = min(max(s, 0), n_src - 1);
    s1 = min(max(s + 1, 0), n_src - 1);
  }
  a0 = max(-32768, min(32767, __float2int_rn((1.f - f) * 2048.f)));
  a1 = max(-32768, min(32767, __float2int_rn(f * 2048.f)));
}

__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int B, int Hs, int Ws, long src_row_bytes,
                                                               long src_img_bytes, uint8_t* __restrict__ dst, int Hd, int Wd,
                                                               double scale_x, double scale_y, int area2) {
  const int wq = Wd >> 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)B * Hd * wq) return;
  const int xq = (int)(t % wq);
  const int dy = (int)((t / wq) % Hd);
  const int b = (int)(t / ((long)wq * Hd));
  const uint8_t* img = src + (long)b * src_img_bytes;
  uint32_t out[3] = {0u, 0u, 0u};
  if (area2) {
    const uint8_t* r0 = img + (long)(2 * dy) * src_row_bytes;
    const uint8_t* r1 = r0 + src_row_bytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sx = (xq * 4 + i) * 2 * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const uint32_t v = ((uint32_t)r0[sx + c] + r0[sx + 3 + c] + r1[sx + c] + r1[sx + 3 + c] + 2u) >> 2;
        const int byte = i * 3 + c;
        out[byte >> 2] |= v << ((byte & 3) * 8);
      }
    }
  } else {
    int y0, y1, b0, b1;
    lin_tap(dy, scale_y, Hs, y0, y1, b0, b1, false);
    const uint8_t* r0 = img + (long)y0 * src_row_bytes;
    const uint8_t* r1 = img + (long)y1 * src_row_bytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int x0, x1, a0, a1;
      lin_tap(xq * 4 + i, scale_x, Ws, x0, x1, a0, a1, true);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = (int)r0[x0 * 3 + c] * a0 + (int)r0[x1 * 3 + c] * a1;
        const int h1 = (int)r1[x0 * 3 + c] * a0 + (int)r1[x1 * 3 + c] * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        const int byte = i * 3 + c;
        out[byte >> 2] |= (uint32_t)min(max(v, 0), 255) << ((byte & 3) * 8);
      }
    }
  }
  uint32_t* o = reinterpret_cast<uint32_t*>(dst + (((long)b * Hd + dy) * Wd + xq * 4) * 3);
  o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
}

// ------------------------------------------------------------------------------------------------
// Pairwise IoU of pixel x0y0x1y1 boxes, the similarity the validator feeds to HOTA
// (TrackValidator._calculate_box_ious, models/MOTRtrack/val.py:517-553, box_format 'x0y0x1y1', do_ioa False):
// intersection zeroed where either area or the union is <= eps, union forced to 1 there.  T frames padded to
// [T, n, 4] x [T, K, 4] with optional per-frame counts; pairs beyond a frame's counts are written as 0.
__global__ __launch_bounds__(256) void box_iou_kernel(const float* __restrict__ a, const float* __restrict__ b, int T, int n, int K,
                                                      const int32_t* __restrict__ na, const int32_t* __restrict__ nb,
                                                      float* __restrict__ out) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)T * n * K) return;
  const int j = (int)(t % K);
  const int i = (int)((t / K) % n);
  const int f = (int)(t / ((long)K * n));
  float r = 0.f;
  if (i < (na ? na[f] : n) && j < (nb ? nb[f] : K)) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(a + ((long)f * n + i) * 4);
    const f32x4 q = *reinterpret_cast<const f32x4*>(b + ((long)f * K + j) * 4);
    constexpr float EPS = 2.220446049250313e-16f;      // np.finfo('float').eps as the reference compares it
    float inter = fmaxf(fminf(p.z, q.z) - fmaxf(p.x, q.x), 0.f) * fmaxf(fminf(p.w, q.w) - fmaxf(p.y, q.y), 0.f);
    const float a1 = (p.z - p.x) * (p.w - p.y), a2 = (q.z - q.x) * (q.w - q.y);
    float uni = a1 + a2 - inter;
    if (a1 <= EPS || a2 <= EPS || uni <= EPS) inter = 0.f;
    if (uni <= EPS) uni = 1.f;
    r = inter / uni;
  }
  out[t] = r;
}

// Generic operator form (reference plugin API): one thread per output scalar, d fastest.
template <typename T>
__global__ __launch_bounds__(256) void msda_generic_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                           const int64_t* __restrict__ lstart, const T* __restrict__ loc,
                                                           const T* __restrict__ aw, int N, int S, int M, int D, int L,
                                                           int Lq, int P, T* __restrict__ out) {
  typedef typename AccOf<T>::type A;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)N * Lq * M * D;
  if (t >= total) return;
  const int d = (int)(t % D);
  long r = t / D;
  const int m = (int)(r % M); r /= M;
  const int q = (int)(r % Lq);
  const int n = (int)(r / Lq);
  const T* lp = loc + ((((long)n * Lq + q) * M + m) * L) * P * 2;
  const T* ap = aw + ((((long)n * Lq + q) * M + m) * L) * P;
  A acc = 0;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const T* vl = value + (((long)n * S + lstart[l]) * M + m) * D + d;
    for (int p = 0; p < P; ++p) {
      const A x = (A)DT<T>::load1(lp + (l * P + p) * 2) * W - (A)0.5;
      const A y = (A)DT<T>::load1(lp + (l * P + p) * 2 + 1) * H - (A)0.5;
      const A w = DT<T>::load1(ap + l * P + p);
      const A xf = __builtin_elementwise_floor(x), yf = __builtin_elementwise_floor(y);
      const A fx = x - xf, fy = y - yf;
      const int x0 = (int)xf, y0 = (int)yf;
      A s = 0;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int xi = x0 + (tt & 1), yi = y0 + (tt >> 1);
        if ((unsigned)xi < (unsigned)W && (unsigned)yi < (unsigned)H)
          s += ((tt & 1) ? fx : 1 - fx) * ((tt >> 1) ? fy : 1 - fy) *
               (A)DT<T>::load1(vl + ((long)yi * W + xi) * (long)M * D);
      }
      acc += w * s;
    }
  }
  DT<T>::store1(out + t, acc);
}

// Operator backward (ms_deform_attn_backward, ms_deform_attn.h:42-62; arithmetic of ms_deform_im2col_cuda.cuh:301-400
// `ms_deform_attn_col2im_bilinear`): a group of G lanes owns one sample (n, q, m, l, p) and strides the D channels, so
// grad_sampling_loc / grad_attn_weight are a shuffle reduction and a plain store (the reference needs shared-memory
// reductions or atomics for them); grad_value is scattered with hardware float/double atomics.
template <typename T>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lstart, const T* __restrict__ loc,
                                                       const T* __restrict__ aw, const T* __restrict__ gout, int N, int S, int M,
                                                       int D, int L, int Lq, int P, int G, T* __restrict__ gvalue,
                                                       T* __restrict__ gloc, T* __restrict__ gaw) {
  const long total = (long)N * Lq * M * L * P;
  const long sid = (long)blockIdx.x * (256 / G) + threadIdx.x / G;
  const int lane = threadIdx.x % G;
  const bool live = sid < total;                       // dead groups still take part in the shuffles
  long r = live ? sid : total - 1;
  const long smp = r;
  r /= P;
  const int l = (int)(r % L); r /= L;
  const int m = (int)(r % M); r /= M;
  const int q = (int)(r % Lq);
  const int n = (int)(r / Lq);
  const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
  const T x = loc[smp * 2] * W - (T)0.5, y = loc[smp * 2 + 1] * H - (T)0.5;
  const T w = aw[smp];
  T gw = 0, gh = 0, ga = 0;
  if (live && y > -1 && x > -1 && y < H && x < W) {   // the reference's sample gate (cuh:339, :285 forward)
    const T xf = __builtin_elementwise_floor(x), yf = __builtin_elementwise_floor(y);
    const T lx = x - xf, ly = y - yf, hx = 1 - lx, hy = 1 - ly;
    const int x0 = (int)xf, y0 = (int)yf;
    const bool okx0 = x0 >= 0, okx1 = x0 + 1 < W, oky0 = y0 >= 0, oky1 = y0 + 1 < H;
    const long cs = (long)M * D;                        // stride between neighbouring cells of one head
    const long base = (((long)n * S + lstart[l]) * M + m) * D;
    const long i00 = base + ((long)y0 * W + x0) * cs, i01 = i00 + cs, i10 = i00 + (long)W * cs, i11 = i10 + cs;
    const T* go = gout + (((long)n * Lq + q) * M + m) * D;
    for (int d = lane; d < D; d += G) {
      const T g = go[d], tg = g * w;
      T v00 = 0, v01 = 0, v10 = 0, v11 = 0;
      if (oky0 && okx0) { v00 = value[i00 + d]; unsafeAtomicAdd(gvalue + i00 + d, hy * hx * tg); }
      if (oky0 && okx1) { v01 = value[i01 + d]; unsafeAtomicAdd(gvalue + i01 + d, hy * lx * tg); }
      if (oky1 && okx0) { v10 = value[i10 + d]; unsafeAtomicAdd(gvalue + i10 + d, ly * hx * tg); }
      if (oky1 && okx1) { v11 = value[i11 + d]; unsafeAtomicAdd(gvalue + i11 + d, ly * lx * tg); }
      ga += g * (hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11));
      gw += tg * (hy * (v01 - v00) + ly * (v11 - v10));
      gh += tg * (hx * (v10 - v00) + lx * (v11 - v01));
    }
  }
  for (int o = G >> 1; o > 0; o >>= 1) {
    gw += __shfl_xor(gw, o, 64);
    gh += __shfl_xor(gh, o, 64);
    ga += __shfl_xor(ga, o, 64);
  }
  if (live && lane == 0) {
    gloc[smp * 2] = gw * W;
    gloc[smp * 2 + 1] = gh * H;
    gaw[smp] = ga;
  }
}

// ------------------------------------------------------------------------------------------------
// ID assignment (per-frame reset semantics) + predictor rows.
__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int tid, int nthreads, int* total) {
  const int lane = tid & 63, wave = tid >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  const int nw = nthreads >> 6;
  for (int w = 0; w < nw; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(1024) void assign_post_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                           int nq, int nc, float score_thresh, float conf, float img_w,
                                                           float img_h, float* __restrict__ y, float* __restrict__ scores,
                                                           int64_t* __restrict__ obj_idxes, float* __restrict__ rows,
                                                           int64_t* __restrict__ track_id, int32_t* __restrict__ n_rows,
                                                           int32_t* __restrict__ n_ids) {
  __shared__ int wsum[16];
  const int b = blockIdx.x, i = threadIdx.x;
  const bool in = i < nq;
  float score = 0.f, cx = 0.f, cy = 0.f, w = 0.f, h = 0.f;
  int cls = 0;
  if (in) {
    const float* lg = logits + ((long)b * nq + i) * nc;
    const float* bx = boxes + ((long)b * nq + i) * 4;
    float* yr = y + ((long)b * nq + i) * (4 + nc);
    cx = bx[0]; cy = bx[1]; w = bx[2]; h = bx[3];
    yr[0] = cx; yr[1] = cy; yr[2] = w; yr[3] = h;
    float best_l = -INFINITY;
    score = -INFINITY;
    for (int c = 0; c < nc; ++c) {
      const float pr = sigmoidf_(lg[c]);
      yr[4 + c] = pr;
      score = fmaxf(score, pr);
      if (lg[c] > best_l) { best_l = lg[c]; cls = c; }
    }
    scores[(long)b * nq + i] = score;
  }
  const int born = in && score >= score_thresh;
  int K;
  const int id = block_excl_scan(born, wsum, i, 1024, &K);
  if (in) obj_idxes[(long)b * nq + i] = born ? (int64_t)id : (int64_t)-1;
  // predictor rows: active branch if K > 0 else detection fallback (predict.py:43-94)
  const bool cand = in && (K > 0 ? born : true);
  const int keep = cand && score > conf;
  int nkeep;
  const int pos = block_excl_scan(keep, wsum, i, 1024, &nkeep);
  if (keep) {
    float* r = rows + ((long)b * nq + pos) * 6;
    // ops.xywh2xyxy then scale (predict.py:58-72)
    r[0] = (cx - w / 2) * img_w; r[1] = (cy - h / 2) * img_h;
    r[2] = (cx + w / 2) * img_w; r[3] = (cy + h / 2) * img_h;
    r[4] = score; r[5] = (float)cls;
  }
  if (K > 0 && born) track_id[(long)b * nq + id] = (int64_t)id;
  if (i == 0) { n_rows[b] = nkeep; n_ids[b] = K > 0 ? K : -1; }
}


// ------------------------------------------------------------------------------------------------
// Temporal mode (spec in DESIGN.md §7): every sequence b owns a fixed-size query memory of NMAX track
// slots in HBM (content embedding, position embedding, reference box, id, miss counter; the first n_trk[b] slots are live).
// A frame's decoder runs on Lq = NMAX + nq rows per sequence: [track slots | this frame's detect queries], the order of
// MYDecoder._get_decoder_input's concatenations (nn/modules/head.py:1055-1064).
template <typename T>
__global__ __launch_bounds__(256) void temporal_assemble_kernel(const T* __restrict__ trk_embed, const T* __restrict__ trk_qpos,
                                                                const float* __restrict__ trk_ref, const int32_t* __restrict__ n_trk,
                                                                const T* __restrict__ det_embed, int64_t ld_de,
                                                                const T* __restrict__ det_qpos, int64_t ld_dq,
                                                                const float* __restrict__ det_ref, int B, int NMAX, int nq,
                                                                T* __restrict__ embed, int64_t ld_e, T* __restrict__ qpos, int64_t ld_q,
                                                                float* __restrict__ ref_logit, float* __restrict__ ref_sig) {
  constexpr int KPB = DT<T>::KPB, CH = 256 / KPB;        // 16-B chunks per 256-wide row
  const int Lq = NMAX + nq;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)B * Lq * CH) return;
  const int c = (int)(t % CH) * KPB;
  const long row = t / CH;
  const int b = (int)(row / Lq), i = (int)(row % Lq);
  u32x4 e = {0u, 0u, 0u, 0u}, qp = e;
  f32x4 rl = {0.f, 0.f, 0.f, 0.f};
  if (i < NMAX) {
    if (i < n_trk[b]) {                                  // dead slots stay zero: finite keys / values for the masked rows
      const long sr = (long)b * NMAX + i;
      e = *reinterpret_cast<const u32x4*>(trk_embed + sr * 256 + c);
      qp = *reinterpret_cast<const u32x4*>(trk_qpos + sr * 256 + c);
      rl = *reinterpret_cast<const f32x4*>(trk_ref + sr * 4);
    }
  } else {
    const long dr = (long)b * nq + (i - NMAX);
    e = *reinterpret_cast<const u32x4*>(det_embed + dr * ld_de + c);
    qp = *reinterpret_cast<const u32x4*>(det_qpos + dr * ld_dq + c);
    rl = *reinterpret_cast<const f32x4*>(det_ref + dr * 4);
  }
  *reinterpret_cast<u32x4*>(embed + row * ld_e + c) = e;
  *reinterpret_cast<u32x4*>(qpos + row * ld_q + c) = qp;
  if (c == 0) {
    *reinterpret_cast<f32x4*>(ref_logit + row * 4) = rl;
    *reinterpret_cast<f32x4*>(ref_sig + row * 4) = f32x4{sigmoidf_(rl.x), sigmoidf_(rl.y), sigmoidf_(rl.z), sigmoidf_(rl.w)};
  }
}

// ID lifecycle of one frame, one block per sequence: RuntimeTrackerBase.update's loop (nn/modules/head.py:1232-1243) on the
// rows in query order with CARRIED ids / miss counters -- birth: id == -1 and score >= score_thresh -> next id of the
// sequence; miss: id >= 0 and score < filter_thresh -> counter + 1, dropped at miss_tol (no reset of the counter, as
// shipped) -- then the compaction of the surviving + newborn rows into the memory slots (stable, query order) and the
// predictor rows of TrackPredictor.postprocess (predict.py:43-94).  The id counter only moves at births (upstream
// MOTR/models/motr.py:303-325); the shipped re-derivation of max_obj_id from a filtered copy (head.py:1278-1281) can hand
// out an id twice and is deliberately not reproduced.
__global__ __launch_bounds__(1024) void temporal_assign_kernel(const float* __restrict__ logits, const float* __restrict__ boxes, int NMAX,
                                                               int nq, int nc, const int64_t* __restrict__ trk_id,
                                                               const int32_t* __restrict__ trk_dis, const int32_t* __restrict__ n_trk,
                                                               int64_t* __restrict__ max_obj_id, float score_thresh, float filter_thresh,
                                                               int miss_tol, float conf, float img_w, float img_h, float* __restrict__ y,
                                                               float* __restrict__ scores, int64_t* __restrict__ obj_idxes,
                                                               int32_t* __restrict__ dis_out, int32_t* __restrict__ sel_rows,
                                                               int32_t* __restrict__ n_new, int32_t* __restrict__ n_overflow,
                                                               float* __restrict__ rows, int64_t* __restrict__ track_id,
                                                               int32_t* __restrict__ n_rows, int32_t* __restrict__ n_ids) {
  __shared__ int wsum[16];
  const int b = blockIdx.x, i = threadIdx.x;
  const int Lq = NMAX + nq;
  const bool in = i < Lq;
  const int nt = n_trk[b];
  const bool valid = in && (i < nt || i >= NMAX);
  float score = 0.f, cx = 0.f, cy = 0.f, w = 0.f, h = 0.f;
  int cls = 0;
  if (in) {
    const float* lg = logits + ((long)b * Lq + i) * nc;
    const float* bx = boxes + ((long)b * Lq + i) * 4;
    float* yr = y + ((long)b * Lq + i) * (4 + nc);
    cx = bx[0]; cy = bx[1]; w = bx[2]; h = bx[3];
    yr[0] = cx; yr[1] = cy; yr[2] = w; yr[3] = h;
    float best_l = -INFINITY;
    score = -INFINITY;
    for (int c = 0; c < nc; ++c) {
      const float pr = sigmoidf_(lg[c]);
      yr[4 + c] = pr;
      score = fmaxf(score, pr);
      if (lg[c] > best_l) { best_l = lg[c]; cls = c; }
    }
    scores[(long)b * Lq + i] = score;
  }
  int64_t id = -1;
  int dis = 0;
  if (valid && i < NMAX) { id = trk_id[(long)b * NMAX + i]; dis = trk_dis[(long)b * NMAX + i]; }
  const int born = valid && id == -1 && score >= score_thresh;
  int nborn;
  const int rank = block_excl_scan(born, wsum, i, 1024, &nborn);
  const int64_t base_id = max_obj_id[b];
  if (born) {
    id = base_id + rank;
  } else if (valid && id >= 0 && score < filter_thresh) {
    dis += 1;
    if (dis >= miss_tol) id = -1;
  }
  __syncthreads();                                        // every thread has read max_obj_id[b]
  if (i == 0) max_obj_id[b] = base_id + nborn;
  const int active = valid && id >= 0;
  if (in) { obj_idxes[(long)b * Lq + i] = id; dis_out[(long)b * Lq + i] = dis; }
  int K;
  const int slot = block_excl_scan(active, wsum, i, 1024, &K);
  // memory slots of the next frame: live rows in query order; empty slots point at the first detect row (always finite)
  if (active && slot < NMAX) sel_rows[(long)b * NMAX + slot] = b * Lq + i;
  if (i < NMAX && i >= min(K, NMAX)) sel_rows[(long)b * NMAX + i] = b * Lq + NMAX;
  // predictor rows: active branch if K > 0 else detection fallback over the frame's rows (predict.py:43-94)
  const bool cand = K > 0 ? active : valid;
  const int keep = cand && score > conf;
  int nkeep;
  const int pos = block_excl_scan(keep, wsum, i, 1024, &nkeep);
  if (keep) {
    float* r = rows + ((long)b * Lq + pos) * 6;
    r[0] = (cx - w / 2) * img_w; r[1] = (cy - h / 2) * img_h;
    r[2] = (cx + w / 2) * img_w; r[3] = (cy + h / 2) * img_h;
    r[4] = score; r[5] = (float)cls;
  }
  if (active) track_id[(long)b * Lq + slot] = id;
  if (i == 0) {
    n_new[b] = min(K, NMAX);
    n_overflow[b] = max(K - NMAX, 0);
    n_rows[b] = nkeep;
    n_ids[b] = K > 0 ? K : -1;
  }
}

// Commit of a frame into the query memory: ids, miss counters, reference boxes (QIM: ref_pts = inverse_sigmoid(pred_boxes),
// MOTR/models/qim.py:299, eps 1e-5 of MOTR/util/misc.py:532-536) of the selected rows; n_trk = n_new.
__global__ __launch_bounds__(256) void temporal_commit_kernel(const int32_t* __restrict__ sel_rows, const int32_t* __restrict__ n_new,
                                                              const int64_t* __restrict__ obj_idxes, const int32_t* __restrict__ dis_out,
                                                              const float* __restrict__ boxes, int B, int NMAX,
                                                              int64_t* __restrict__ trk_id, int32_t* __restrict__ trk_dis,
                                                              float* __restrict__ trk_ref, int32_t* __restrict__ n_trk) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= B * NMAX) return;
  const int b = t / NMAX, s = t % NMAX;
  const bool live = s < n_new[b];
  const int row = sel_rows[t];
  trk_id[t] = live ? obj_idxes[row] : (int64_t)-1;
  trk_dis[t] = live ? dis_out[row] : 0;
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const f32x4 bx = *reinterpret_cast<const f32x4*>(boxes + (long)row * 4);
    auto inv = [](float x) { x = fminf(fmaxf(x, 0.f), 1.f); return logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f)); };
    r = f32x4{inv(bx.x), inv(bx.y), inv(bx.z), inv(bx.w)};
  }
  *reinterpret_cast<f32x4*>(trk_ref + (long)t * 4) = r;
  if (s == 0) n_trk[b] = n_new[b];
}

// ------------------------------------------------------------------------------------------------
// Side state of the shipped path on device: copy-filter of RuntimeTrackerBase.update + FSQM.
constexpr int FSQM_SLOTS = 300, FSQM_DIM = 256;

// head.py:1173-1196 in fp32 without fused multiply-adds (the reference evaluates each op separately)
__device__ __forceinline__ bool iou_gt_08(const float* a, const float* b) {
  if (fabsf(__fsub_rn(a[0], b[0])) > __fmul_rn(0.5f, fminf(a[0], b[0]))) return false;
  if (fabsf(__fsub_rn(a[1], b[1])) > __fmul_rn(0.5f, fminf(a[1], b[1]))) return false;
  const float ix1 = fmaxf(a[0], b[0]), iy1 = fmaxf(a[1], b[1]);
  const float ix2 = fminf(__fadd_rn(a[0], a[2]), __fadd_rn(b[0], b[2]));
  const float iy2 = fminf(__fadd_rn(a[1], a[3]), __fadd_rn(b[1], b[3]));
  const float inter = __fmul_rn(fmaxf(0.f, __fsub_rn(ix2, ix1)), fmaxf(0.f, __fsub_rn(iy2, iy1)));
  const float uni = __fsub_rn(__fadd_rn(__fmul_rn(a[2], a[3]), __fmul_rn(b[2], b[3])), inter);
  return __fdiv_rn(inter, uni) > 0.8f;
}

template <typename T>
__global__ __launch_bounds__(1024) void track_state_kernel(const float* __restrict__ scores, const float* __restrict__ boxes,
                                                           const int64_t* __restrict__ obj, const T* __restrict__ hs, int64_t ld_hs,
                                                           int B, int nq, int32_t* __restrict__ copy_rows,
                                                           int64_t* __restrict__ copy_ids, int32_t* __restrict__ n_copy,
                                                           float* __restrict__ mem, float* __restrict__ conf, int64_t* __restrict__ ids,
                                                           float* __restrict__ fboxes, int32_t* __restrict__ low,
                                                           int32_t* __restrict__ pool, int pool_cap, int32_t* __restrict__ pool_hc) {
  __shared__ int wsum[16];
  __shared__ int act[1024];        // active rows (query order)
  __shared__ int keep[1024];
  __shared__ int kept_row[1024];   // rows of the copy
  __shared__ int kept_id[1024];
  __shared__ int inj_slot[1024];   // slot chosen for the v-th injected query (-1: none)
  __shared__ float bx[1024][4];
  __shared__ int sh_nvalid;
  const int i = threadIdx.x;
  for (int b = 0; b < B; ++b) {
    const float* sc = scores + (long)b * nq;
    const float* bb = boxes + (long)b * nq * 4;
    const int64_t* ob = obj + (long)b * nq;
    // ---- (1) copy half of RuntimeTrackerBase.update
    const int is_act = i < nq && ob[i] >= 0;
    int K;
    const int apos = block_excl_scan(is_act, wsum, i, 1024, &K);
    if (is_act) { act[apos] = i; bx[apos][0] = bb[i * 4]; bx[apos][1] = bb[i * 4 + 1]; bx[apos][2] = bb[i * 4 + 2]; bx[apos][3] = bb[i * 4 + 3]; }
    if (i < 1024) keep[i] = 1;
    __syncthreads();
    for (int a = 0; a < K; ++a) {          // greedy suppression, head.py:1159-1169
      if (keep[a] && i > a && i < K && keep[i] && iou_gt_08(bx[a], bx[i])) keep[i] = 0;
      __syncthreads();
    }
    const int is_kept = i < K && keep[i];
    int Kc;
    const int kpos = block_excl_scan(is_kept, wsum, i, 1024, &Kc);
    // renumber ids above max_obj_id_pre = 0 (head.py:1268-1275): k-th such row (in order) -> k + 1
    const int idv = is_kept ? (int)ob[act[i]] : 0;
    int dummy;
    const int rpos = block_excl_scan(is_kept && idv > 0, wsum, i, 1024, &dummy);
    if (is_kept) {
      const int nid = idv > 0 ? rpos + 1 : idv;
      kept_row[kpos] = act[i];
      kept_id[kpos] = nid;
      copy_rows[(long)b * nq + kpos] = act[i];
      copy_ids[(long)b * nq + kpos] = nid;
    }
    if (i == 0) n_copy[b] = K > 0 ? Kc : 0;
    __syncthreads();
    // ---- (2a) FSQM.update_confidence over the nq-row track_queries, indexed BY ID (fsqm.py:127-132)
    if (i < nq) {
      const int64_t id = ob[i];
      if (id >= 0 && id < FSQM_SLOTS) {
        conf[id] = sc[i];
        fboxes[id * 4] = bb[i * 4]; fboxes[id * 4 + 1] = bb[i * 4 + 1]; fboxes[id * 4 + 2] = bb[i * 4 + 2]; fboxes[id * 4 + 3] = bb[i * 4 + 3];
        low[id] = 0;
      }
    }
    __syncthreads();
    // ---- (2b) inject_new_queries over the copy (detect_queries): control flow by one lane, as shipped
    const int nd = K > 0 ? Kc : 0;   // no active row: update() returns the full Instances; its scores can exceed 0.7 only if active
    if (i == 0) {
      int nv = 0, cursor = 0, head = pool_hc[0], count = pool_hc[1];
      for (int v = 0; v < nd; ++v) {
        const int row = kept_row[v];
        if (!(sc[row] > 0.7f)) continue;
        while (cursor < FSQM_SLOTS && ids[cursor] != -1) ++cursor;
        if (cursor >= FSQM_SLOTS) break;                    // memory full (fsqm.py:77-79)
        int pid = -1;
        if (count > 0) { pid = pool[head]; head = (head + 1) % pool_cap; --count; } else pool_hc[2] = 1;   // pop(0) of an empty list would raise
        ids[cursor] = pid;
        conf[cursor] = sc[row];
        fboxes[cursor * 4] = bb[row * 4]; fboxes[cursor * 4 + 1] = bb[row * 4 + 1];
        fboxes[cursor * 4 + 2] = bb[row * 4 + 2]; fboxes[cursor * 4 + 3] = bb[row * 4 + 3];
        low[cursor] = 0;
        inj_slot[nv] = cursor;
        act[nv] = row;                                      // reuse: source row of the nv-th injection
        ++nv;
      }
      pool_hc[0] = head; pool_hc[1] = count;
      sh_nvalid = nv;
    }
    __syncthreads();
    for (int v = 0; v < sh_nvalid; ++v)                     // embeddings, in injection order (later wins)
      if (i < FSQM_DIM) mem[(long)inj_slot[v] * FSQM_DIM + i] = DT<T>::load1(hs + ((long)b * nq + act[v]) * ld_hs + i);
    __syncthreads();
    // ---- (2c) remove_inactive_queries (fsqm.py:102-115), pool appends in slot order
    int freed = 0;
    if (i < FSQM_SLOTS && conf[i] < 0.3f) {
      const int l = low[i] + 1;
      low[i] = l;
      freed = l >= 3;
    }
    int nfree;
    const int fpos = block_excl_scan(freed, wsum, i, 1024, &nfree);
    if (freed) {
      const int head = pool_hc[0], count = pool_hc[1];
      if (count + fpos < pool_cap) pool[(head + count + fpos) % pool_cap] = (int)ids[i]; else pool_hc[2] = 1;
      conf[i] = 0.f; ids[i] = -1; low[i] = 0;
      fboxes[i * 4] = fboxes[i * 4 + 1] = fboxes[i * 4 + 2] = fboxes[i * 4 + 3] = 0.f;
    }
    __syncthreads();
    if (i < 1024) keep[i] = freed;                          // reuse: freed-slot flags
    __syncthreads();
    for (int s = 0; s < FSQM_SLOTS; ++s)
      if (keep[s] && i < FSQM_DIM) mem[(long)s * FSQM_DIM + i] = 0.f;
    if (i == 0) pool_hc[1] = min(pool_cap, pool_hc[1] + nfree);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void fsqm_reset_kernel(float* mem, float* conf, int64_t* ids, float* fboxes, int32_t* low,
                                                         int32_t* pool, int pool_cap, int32_t* pool_hc) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < FSQM_SLOTS * FSQM_DIM) mem[t] = 0.f;
  if (t < FSQM_SLOTS) { conf[t] = 0.f; ids[t] = -1; low[t] = 0; pool[t] = t; fboxes[t * 4] = fboxes[t * 4 + 1] = fboxes[t * 4 + 2] = fboxes[t * 4 + 3] = 0.f; }
  if (t == 0) { pool_hc[0] = 0; pool_hc[1] = FSQM_SLOTS; pool_hc[2] = 0; }
}


// ------------------------------------------------------------------------------------------------
// Config C1: Detect decode + NMS.
template <typename T>
__global__ __launch_bounds__(256) void detect_decode_kernel(const T* __restrict__ box, int64_t ld_box, const T* __restrict__ cls,
                                                            int64_t ld_cls, int B, int h, int w, int nc, float stride, int a_off,
                                                            int A, float* __restrict__ y) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int hw = h * w;
  if (t >= (long)B * hw) return;
  const int b = (int)(t / hw), a = (int)(t - (long)b * hw);
  const T* bp = box + t * ld_box;
  float d[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {            // DFL: expectation of softmax over 16 bins
    float v[16], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
      const f32x4 q4 = DT<T>::load4(bp + s * 16 + k);
      v[k] = q4.x; v[k + 1] = q4.y; v[k + 2] = q4.z; v[k + 3] = q4.w;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) mx = fmaxf(mx, v[k]);
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const float e = expf(v[k] - mx); den += e; num += e * (float)k; }
    d[s] = num / den;
  }
  const float ax = (float)(a % w) + 0.5f, ay = (float)(a / w) + 0.5f;
  const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
  float* yb = y + (long)b * (4 + nc) * A + a_off + a;
  yb[0] = (x1 + x2) / 2 * stride;
  yb[(long)A] = (y1 + y2) / 2 * stride;
  yb[2L * A] = (x2 - x1) * stride;
  yb[3L * A] = (y2 - y1) * stride;
  const T* cp = cls + t * ld_cls;
  for (int c = 0; c < nc; ++c) yb[(long)(4 + c) * A] = sigmoidf_(DT<T>::load1(cp + c));
}

constexpr int NMS_THREADS = 1024;

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* __restrict__ y, int nc, int A, int P2, float conf_thres,
                                                          float iou_thres, int max_det, float max_wh, float gain, float pad_x,
                                                          float pad_y, float clip_w, float clip_h, float* __restrict__ rows,
                                                          int32_t* __restrict__ n_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(dyn);     // [P2] (score desc, anchor asc)
  unsigned char* dead = dyn + (size_t)P2 * 8;                                // [P2]
  __shared__ int sh_n, sh_kept, sh_cur_alive;
  __shared__ float cur[5];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* yb = y + (long)b * (4 + nc) * A;
  if (tid == 0) { sh_n = 0; sh_kept = 0; }
  for (int i = tid; i < P2; i += NMS_THREADS) { keys[i] = ~0ull; dead[i] = 0; }
  __syncthreads();
  // candidates: best class score > conf (ops.py:203, :232-233)
  for (int a = tid; a < A; a += NMS_THREADS) {
    float best = -INFINITY;
    for (int c = 0; c < nc; ++c) best = fmaxf(best, yb[(long)(4 + c) * A + a]);
    if (best > conf_thres) {
      const int pos = atomicAdd(&sh_n, 1);
      keys[pos] = ((unsigned long long)(~order_key(best)) << 32) | (unsigned)a;
    }
  }
  __syncthreads();
  const int n = sh_n;
  for (int k = 2; k <= P2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P2; i += NMS_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long u = keys[i], v = keys[ixj];
          if ((u > v) == ((i & k) == 0)) { keys[i] = v; keys[ixj] = u; }
        }
      }
      __syncthreads();
    }
  auto load_box = [&](int a, float* o) {          // xyxy + class offset (ops.py:206, :262-263), score, class
    const float cx = yb[a], cy = yb[(long)A + a], w = yb[2L * A + a], h = yb[3L * A + a];
    float best = -INFINITY; int bc = 0;
    for (int c = 0; c < nc; ++c) { const float v = yb[(long)(4 + c) * A + a]; if (v > best) { best = v; bc = c; } }
    o[0] = cx - w / 2; o[1] = cy - h / 2; o[2] = cx + w / 2; o[3] = cy + h / 2; o[4] = best; o[5] = (float)bc;
  };
  for (int i = 0; i < n; ++i) {
    if (tid == 0) {
      sh_cur_alive = !dead[i] && sh_kept < max_det;
      if (sh_cur_alive) {
        float o[6];
        const int a = (int)(keys[i] & 0xffffffffu);
        load_box(a, o);
        const float off = o[5] * max_wh;
        cur[0] = o[0] + off; cur[1] = o[1] + off; cur[2] = o[2] + off; cur[3] = o[3] + off;
        cur[4] = (cur[2] - cur[0]) * (cur[3] - cur[1]);
        float* r = rows + ((long)b * max_det + sh_kept) * 6;
        float bx0 = (o[0] - pad_x) / gain, by0 = (o[1] - pad_y) / gain, bx1 = (o[2] - pad_x) / gain, by1 = (o[3] - pad_y) / gain;
        if (clip_w > 0.f) { bx0 = fminf(fmaxf(bx0, 0.f), clip_w); bx1 = fminf(fmaxf(bx1, 0.f), clip_w);
                            by0 = fminf(fmaxf(by0, 0.f), clip_h); by1 = fminf(fmaxf(by1, 0.f), clip_h); }
        r[0] = bx0; r[1] = by0; r[2] = bx1; r[3] = by1; r[4] = o[4]; r[5] = o[5];
        ++sh_kept;
      }
    }
    __syncthreads();
    if (sh_kept >= max_det && !sh_cur_alive) break;      // uniform: written by lane 0 before the barrier
    if (sh_cur_alive) {
      for (int j = i + 1 + tid; j < n; j += NMS_THREADS)
        if (!dead[j]) {
          float o[6];
          load_box((int)(keys[j] & 0xffffffffu), o);
          const float off = o[5] * max_wh;
          const float x0 = o[0] + off, y0 = o[1] + off, x1 = o[2] + off, y1 = o[3] + off;
          const float iw = fmaxf(fminf(cur[2], x1) - fmaxf(cur[0], x0), 0.f), ih = fmaxf(fminf(cur[3], y1) - fmaxf(cur[1], y0), 0.f);
          const float inter = iw * ih;
          if (inter / (cur[4] + (x1 - x0) * (y1 - y0) - inter) > iou_thres) dead[j] = 1;
        }
    }
    __syncthreads();
  }
  if (tid == 0) n_rows[b] = sh_kept;
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, int64_t lds_, int M, int N4, T* __restrict__ dst,
                                                   int64_t ldd) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)M * N4) return;
  const int m = (int)(t / N4), c = (int)(t % N4) * 4;
  DT<T>::store4(dst + (long)m * ldd + c, *reinterpret_cast<const f32x4*>(src + (long)m * lds_ + c));
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, int64_t lds_, const int32_t* __restrict__ rows,
                                                          int M, int NC, T* __restrict__ dst, int64_t ldd) {
  constexpr int KPB = DT<T>::KPB;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)M * NC) return;
  const int m = (int)(t / NC), c = (int)(t % NC) * KPB;
  *reinterpret_cast<u32x4*>(dst + (long)m * ldd + c) = *reinterpret_cast<const u32x4*>(src + (long)rows[m] * lds_ + c);
}

// Round 4 (input_proj folded into its consumers): the selected tokens of a frame come from any pyramid level.  level_rows: token ->
// its level and its row in that level's own [B, h*w, C] tensor (0 = a dummy row for the other levels' gathers).  level_select: of the
// per-level fp32 products of a row keep the one of the row's level, add that level's BN shift, round once; rows of masked tokens are
// zero (valid_mask * feats, head.py:1039).
struct LevelTable { int n; int off[5]; int hw[4]; };
__global__ __launch_bounds__(256) void level_rows_kernel(const int32_t* __restrict__ tok_local, int B, int nq, LevelTable lv,
                                                         int32_t* __restrict__ rows, int32_t* __restrict__ level) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= B * nq) return;
  const int b = m / nq, tok = tok_local[m];
  int l = 0;
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (j < lv.n && tok >= lv.off[j]) l = j;
  level[m] = l;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < lv.n) rows[(long)j * B * nq + m] = j == l ? b * lv.hw[j] + tok - lv.off[j] : 0;
}
template <typename T>
__global__ __launch_bounds__(256) void level_select_kernel(const float* __restrict__ G, int64_t level_stride, int64_t ldg,
                                                           const int32_t* __restrict__ level, const float* __restrict__ shift,
                                                           const int32_t* __restrict__ tok_local, const uint8_t* __restrict__ valid,
                                                           int M, int NC, T* __restrict__ dst, int64_t ldd) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)M * NC) return;
  const int m = (int)(t / NC), c = (int)(t % NC) * 4;
  const int l = level[m];
  f32x4 v = *reinterpret_cast<const f32x4*>(G + l * level_stride + (long)m * ldg + c) + *reinterpret_cast<const f32x4*>(shift + l * 256 + c);
  if (valid && !valid[tok_local[m]]) v = f32x4{0.f, 0.f, 0.f, 0.f};
  DT<T>::store4(dst + (long)m * ldd + c, v);
}

__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ in, int n, float* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n) out[t] = sigmoidf_(in[t]);
}

// Round 6: processing ORDER of a frame's decoder queries for the deformable gather.  The queries of a frame arrive in top-k SCORE order,
// i.e. spatially random, so the 2 x 2 windows of the queries one block gathers for share nothing.  perm[b][i] = the query to process
// i-th: the frame's queries sorted by the Morton code of the P3 cell of their reference point (ties: query index).  Only the order in
// which the gather kernels WALK the queries changes -- every output row stays where it is and holds the same bits.
// One block per frame; bitonic sort of (code << 10 | index) keys in LDS (Lq <= 1024).
__device__ __forceinline__ uint32_t morton_spread8(uint32_t v) {     // 8 bits -> the even bits of 16
  v = (v | (v << 4)) & 0x0f0fu;
  v = (v | (v << 2)) & 0x3333u;
  v = (v | (v << 1)) & 0x5555u;
  return v;
}
__global__ __launch_bounds__(512) void query_order_kernel(const float* __restrict__ ref, int Lq, int H0, int W0, int P2,
                                                          int32_t* __restrict__ perm) {
  __shared__ uint32_t key[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < P2; i += 512) {
    uint32_t k = 0xffffffffu;
    if (i < Lq) {
      const float cx = ref[((long)b * Lq + i) * 4 + 0], cy = ref[((long)b * Lq + i) * 4 + 1];
      // (a NaN / out-of-range centre lands in cell 0 or the last cell: any order is a valid order)
      const int xi = min(max((int)(cx * (float)W0), 0), min(W0, 256) - 1), yi = min(max((int)(cy * (float)H0), 0), min(H0, 256) - 1);
      k = ((morton_spread8((uint32_t)xi) | (morton_spread8((uint32_t)yi) << 1)) << 10) | (uint32_t)i;
    }
    key[i] = k;
  }
  __syncthreads();
  for (int k2 = 2; k2 <= P2; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P2; i += 512) {
        const int l = i ^ j;
        if (l > i) {
          const uint32_t a = key[i], c = key[l];
          const bool up = (i & k2) == 0;
          if ((a > c) == up) { key[i] = c; key[l] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < Lq; i += 512) perm[(long)b * Lq + i] = (int32_t)(key[i] & 1023u);
}

inline unsigned nblk(long total, int per = 256) { return (unsigned)((total + per - 1) / per); }

}  // namespace moy

using namespace moy;

// ------------------------------------------------------------------------------------------------
// Upstream MSDeformAttn (MOTR/models/ops/modules/ms_deform_attn.py:83-121), the part between its linears and the native op:
// softmax (or sigmoid) of the attention logits over the L*P samples of a head and the sampling locations
//   2-d reference points: loc = ref[l] + offset / (W_l, H_l);   4-d boxes: loc = ref[l].xy + offset / P * ref[l].wh * 0.5.
// One thread per (query row, head); offsets and logits are two column ranges of one fp32 GEMM output row.
struct PrepLevels { int h[8], w[8]; };
template <typename T>
__global__ __launch_bounds__(256) void msda_prep_kernel(const float* __restrict__ offaw, int64_t ld, int col_off, int col_aw,
                                                        const float* __restrict__ ref, int refdim, int rows, int M, int L, int P,
                                                        PrepLevels lv, int sigmoid_attn, T* __restrict__ loc, T* __restrict__ aw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * M) return;
  const int row = i / M, m = i - row * M;
  const int LP = L * P;
  const float* lg = offaw + (int64_t)row * ld + col_aw + m * LP;
  const float* of = offaw + (int64_t)row * ld + col_off + m * LP * 2;
  float mx = -3.4e38f;
  for (int k = 0; k < LP; ++k) mx = fmaxf(mx, lg[k]);
  float den = 0.f;
  for (int k = 0; k < LP; ++k) den += expf(lg[k] - mx);
  typename AccOf<T>::type* dummy = nullptr; (void)dummy;
  for (int l = 0; l < L; ++l) {
    const float* rf = ref + ((int64_t)row * L + l) * refdim;
    for (int p = 0; p < P; ++p) {
      const int k = l * P + p;
      const float a = sigmoid_attn ? sigmoidf_(lg[k]) : expf(lg[k] - mx) / den;
      float x, y;
      if (refdim == 2) {
        x = rf[0] + of[2 * k] / (float)lv.w[l];
        y = rf[1] + of[2 * k + 1] / (float)lv.h[l];
      } else {
        x = rf[0] + of[2 * k] / (float)P * rf[2] * 0.5f;
        y = rf[1] + of[2 * k + 1] / (float)P * rf[3] * 0.5f;
      }
      const int64_t o = (int64_t)i * LP + k;
      DT<T>::store1(aw + o, a);
      DT<T>::store1(loc + 2 * o, x);
      DT<T>::store1(loc + 2 * o + 1, y);
    }
  }
}

// value.masked_fill_(input_padding_mask[..., None], 0) (ms_deform_attn.py:95-96): rows with mask != 0 become zero
template <typename T>
__global__ __launch_bounds__(256) void mask_rows_kernel(T* __restrict__ x, int64_t ld, int M, int NC, const uint8_t* __restrict__ mask) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * NC) return;
  const int row = (int)(i / NC), c = (int)(i - (long)row * NC);
  if (mask[row]) *reinterpret_cast<u32x4*>(x + (int64_t)row * ld + c * DT<T>::KPB) = u32x4{0u, 0u, 0u, 0u};
}

#define MOY_DISPATCH_T(dtype, ...)                          \
  if ((dtype) == MOY_F32) { using T = float; __VA_ARGS__ }  \
  else if ((dtype) == MOY_BF16) { using T = bf16_t; __VA_ARGS__ } \
  else if ((dtype) == MOY_F16) { using T = f16_t; __VA_ARGS__ } \
  else return MOY_EINVAL;

extern "C" int moy_version(void) { return 100; }

extern "C" const char* moy_strerror(int code) {
  switch (code) {
    case MOY_OK: return "ok";
    case MOY_EINVAL: return "invalid shape or argument";
    case MOY_ENOSYS: return "combination not implemented";
    case MOY_ELAUNCH: return "HIP launch error";
    default: return "unknown error";
  }
}

template <typename T, int FMT>
static int stem_launch(const void* in, int B, int H, int W, const float* w, const float* scale, const float* shift,
                       int Cout, void* out, int64_t ldc, hipStream_t st) {
  const long total = (long)B * (H / 2) * (W / 2);
  T* o = static_cast<T*>(out);
  switch (Cout) {
    case 8: hipLaunchKernelGGL((stem_kernel<T, 8, FMT>), dim3(nblk(total)), dim3(256), 0, st, in, B, H, W, w, scale, shift, o, ldc); break;
    case 16: hipLaunchKernelGGL((stem_kernel<T, 16, FMT>), dim3(nblk(total)), dim3(256), 0, st, in, B, H, W, w, scale, shift, o, ldc); break;
    case 32: hipLaunchKernelGGL((stem_kernel<T, 32, FMT>), dim3(nblk(total)), dim3(256), 0, st, in, B, H, W, w, scale, shift, o, ldc); break;
    case 64: hipLaunchKernelGGL((stem_kernel<T, 64, FMT>), dim3(nblk(total)), dim3(256), 0, st, in, B, H, W, w, scale, shift, o, ldc); break;
    default: return MOY_ENOSYS;
  }
  return launch_status();
}

extern "C" int moy_stem_conv(const void* in, int in_fmt, int B, int H, int W, const float* w, const float* scale,
                             const float* shift, int Cout, void* out, int64_t ldc, int dtype, void* stream) {
  if (!in || !w || !scale || !shift || !out || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return MOY_EINVAL;
  if (ldc < Cout || (ldc % 4) || (in_fmt != 0 && in_fmt != 1)) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    if (reinterpret_cast<uintptr_t>(out) % (4 * sizeof(T))) return MOY_EINVAL;
    return in_fmt == 0 ? stem_launch<T, 0>(in, B, H, W, w, scale, shift, Cout, out, ldc, st)
                       : stem_launch<T, 1>(in, B, H, W, w, scale, shift, Cout, out, ldc, st);
  })
}

extern "C" int moy_stem_conv_mfma(const void* in_u8, int B, int H, int W, const void* wpad, const float* scale, const float* shift,
                                  int Cout, void* out, int64_t ldc, void* stream) {
  if (!in_u8 || !wpad || !scale || !shift || !out || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return MOY_EINVAL;
  if (ldc < Cout || (ldc % 4) || !aligned16(wpad) || !aligned16(scale) || !aligned16(shift) || reinterpret_cast<uintptr_t>(out) % 8 ||
      reinterpret_cast<uintptr_t>(in_u8) % 4)
    return MOY_EINVAL;
  if ((long)B * H * W * 3 > 0x7fffffffffffL) return MOY_EINVAL;
  if (Cout >= 32 && ((ldc % 8) || !aligned16(out))) return MOY_EINVAL;     // 16-byte pixel stores
  const int Ho = H / 2, Wo = W / 2;
  const unsigned blocks = (unsigned)B * ((Ho + STEM_TH - 1) / STEM_TH) * ((Wo + STEM_TW - 1) / STEM_TW);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint8_t* in = static_cast<const uint8_t*>(in_u8);
  const bf16_t* w = static_cast<const bf16_t*>(wpad);
  bf16_t* o = static_cast<bf16_t*>(out);
  switch (Cout) {
    case 16: hipLaunchKernelGGL((stem_mfma_kernel<16>), dim3(blocks), dim3(256), 0, st, in, B, H, W, w, scale, shift, o, ldc); break;
    case 32: hipLaunchKernelGGL((stem_mfma_kernel<32>), dim3(blocks), dim3(256), 0, st, in, B, H, W, w, scale, shift, o, ldc); break;
    case 64: hipLaunchKernelGGL((stem_mfma_kernel<64>), dim3(blocks), dim3(256), 0, st, in, B, H, W, w, scale, shift, o, ldc); break;
    default: return MOY_ENOSYS;
  }
  return launch_status();
}

extern "C" int moy_stem_conv_x3(const void* in_u8, int B, int H, int W, const void* wsplit, const float* scale, const float* shift,
                                int Cout, void* out, int64_t ldc, void* stream) {
  if (!in_u8 || !wsplit || !scale || !shift || !out || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return MOY_EINVAL;
  if (ldc < Cout || (ldc % 4) || !aligned16(wsplit) || !aligned16(scale) || !aligned16(shift) || !aligned16(out) ||
      reinterpret_cast<uintptr_t>(in_u8) % 4)
    return MOY_EINVAL;
  // (the window loads are range-checked dwords at 32-bit offsets: rows of whole dwords, frames below 4 GiB)
  if ((W % 4) || (long)B * H * W * 3 >= 0xfffffffcL) return MOY_ENOSYS;
  const int Ho = H / 2, Wo = W / 2;
  const unsigned tiles = (unsigned)B * ((Ho + STEM_TH - 1) / STEM_TH) * ((Wo + STEM_TW - 1) / STEM_TW);
  const unsigned blocks = (tiles + STEM_X3_TPB - 1) / STEM_X3_TPB;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uint8_t* in = static_cast<const uint8_t*>(in_u8);
  const f16_t* w = static_cast<const f16_t*>(wsplit);
  float* o = static_cast<float*>(out);
  switch (Cout) {
    case 16: hipLaunchKernelGGL((stem_x3_kernel<16>), dim3(blocks), dim3(256), 0, st, in, B, H, W, w, scale, shift, o, ldc); break;
    case 32: hipLaunchKernelGGL((stem_x3_kernel<32>), dim3(blocks), dim3(256), 0, st, in, B, H, W, w, scale, shift, o, ldc); break;
    case 64: hipLaunchKernelGGL((stem_x3_kernel<64>), dim3(blocks), dim3(256), 0, st, in, B, H, W, w, scale, shift, o, ldc); break;
    default: return MOY_ENOSYS;
  }
  return launch_status();
}

extern "C" int moy_sppf_pool(const void* x, int64_t ldx, int B, int H, int W, int C, void* y1, void* y2, void* y3,
                             int64_t ldy, int dtype, void* stream) {
  if (!x || !y1 || !y2 || !y3 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (ldy % 8)) return MOY_EINVAL;
  if (!aligned16(x) || !aligned16(y1) || !aligned16(y2) || !aligned16(y3)) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    const int cpr = C / DT<T>::KPB;                      // 16-byte chunks per pixel
    int G = (cpr % 4 == 0) ? 4 : ((cpr % 2 == 0) ? 2 : 1);
    static const int gmax = knob("MOY_SPPF_G", 4);                                // MOY_SPPF_G: cap of the group size (A/B runs)
    while (G > 1 && (G > gmax || (size_t)H * W * 32 * G > 160 * 1024)) G >>= 1;
    const size_t lds = (size_t)H * W * 32 * G;
    if (lds > 160 * 1024) return MOY_ENOSYS;   // the planes of one chunk must fit in LDS (P5 level: 19x34 .. 34x60)
    auto kern = sppf_pool_kernel<T>;
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return MOY_ELAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(B * (cpr / G)), dim3(512), lds, st, static_cast<const T*>(x), ldx, H, W, C, G,
                       static_cast<T*>(y1), static_cast<T*>(y2), static_cast<T*>(y3), ldy);
    return launch_status();
  })
}

extern "C" int moy_upsample2x(const void* x, int64_t ldx, int B, int H, int W, int C, void* y, int64_t ldy, int dtype,
                              void* stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (ldy % 8)) return MOY_EINVAL;
  if (!aligned16(x) || !aligned16(y)) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    const long total = (long)B * 4 * H * W * (C / DT<T>::KPB);
    hipLaunchKernelGGL((upsample2x_kernel<T>), dim3(nblk(total)), dim3(256), 0, st, static_cast<const T*>(x), ldx, B, H, W, C,
                       static_cast<T*>(y), ldy);
    return launch_status();
  })
}

extern "C" int moy_rowdot(const void* X, int64_t ldx, const int32_t* x_rows, int M, int K, const float* Wt,
                          const float* bias, int N, int mode, const float* aux, const int32_t* aux_rows, float* y, int dtype,
                          void* stream) {
  if (!X || !Wt || !y || M <= 0 || K <= 0 || (K % 4) || N <= 0 || N > 8 || (ldx % 4)) return MOY_EINVAL;
  if (mode < 0 || mode > 2 || (mode != 0 && (!aux || N != 4)) || (mode == 2 && !aux_rows)) return MOY_EINVAL;
  if (!aligned16(Wt)) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((rowdot_kernel<T>), dim3((M + 3) / 4), dim3(256), 0, st, static_cast<const T*>(X), ldx, x_rows, M, K, Wt,
                       bias, N, mode, aux, aux_rows, y);
    return launch_status();
  })
}

extern "C" int moy_topk(const float* scores, int B, int S, int nc, int nq, const uint8_t* valid, int32_t* idx_local,
                        int32_t* idx_global, int32_t* n_masked, void* stream) {
  if (!scores || !idx_local || !idx_global || B <= 0 || S <= 0 || nc <= 0 || nq <= 0 || nq > S || nq > 4096) return MOY_EINVAL;
  int P2 = 1;
  while (P2 < nq) P2 <<= 1;
  hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(TOPK_THREADS), P2 * 8, static_cast<hipStream_t>(stream), scores, S, nc, nq, P2,
                     valid, idx_local, idx_global, n_masked);
  return launch_status();
}

extern "C" int moy_pos2posemb(const float* pos, int M, void* out, int64_t ldo, int dtype, void* stream) {
  if (!pos || !out || M <= 0 || ldo < 256) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((posemb_kernel<T>), dim3(nblk((long)M * 128)), dim3(256), 0, st, pos, M, static_cast<T*>(out), ldo);
    return launch_status();
  })
}

template <typename T, int NKT, bool MASKED>
static int mha_launch(const void* qkv, int64_t ld_qkv, int B, int L, int nh, int E, void* out, int64_t ldo, const int32_t* n_prefix,
                      int split, hipStream_t st) {
  constexpr int NPD = 32 / (4 * DT<T>::KPB), ESZ = 16 / DT<T>::KPB;
  constexpr int Lp = NKT * 16;
  const size_t lds = (size_t)NPD * Lp * 64 + 32 * ((size_t)Lp * ESZ + 16) + (size_t)Lp * 4;
  auto kern = mha_kernel<T, NKT, MASKED>;
  static bool attr_set = false;   // opt in to > 64 KiB of LDS once per kernel symbol
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  // enough blocks to cover the chip: split the 16-query tiles of a (b, head) pair over grid.y
  const int nqt = (L + 15) / 16;
  int splits = (768 + B * nh - 1) / (B * nh);
  splits = splits < 1 ? 1 : (splits > (nqt + 3) / 4 ? (nqt + 3) / 4 : splits);
  const int tpb = (nqt + splits - 1) / splits;
  hipLaunchKernelGGL(kern, dim3(B * nh, (nqt + tpb - 1) / tpb), dim3(256), lds, st, static_cast<const T*>(qkv), ld_qkv, L, nh, E,
                     static_cast<T*>(out), ldo, tpb, n_prefix, split);
  return launch_status();
}

template <typename T, int NKT>
static int mha_launch_m(const void* qkv, int64_t ld_qkv, int B, int L, int nh, int E, void* out, int64_t ldo, const int32_t* n_prefix,
                        int split, hipStream_t st) {
  return n_prefix ? mha_launch<T, NKT, true>(qkv, ld_qkv, B, L, nh, E, out, ldo, n_prefix, split, st)
                  : mha_launch<T, NKT, false>(qkv, ld_qkv, B, L, nh, E, out, ldo, n_prefix, split, st);
}

static int mha_host(const void* qkv, int64_t ld_qkv, int B, int L, int nh, int E, const int32_t* n_prefix, int split, void* out,
                    int64_t ldo, int dtype, void* stream) {
  if (!qkv || !out || B <= 0 || L <= 0 || nh <= 0 || E != nh * 32 || ld_qkv < 3 * E || (ld_qkv % 8) || (ldo % 4)) return MOY_EINVAL;
  if (!aligned16(qkv) || reinterpret_cast<uintptr_t>(out) % 8) return MOY_EINVAL;
  if (split < 0 || split > L) return MOY_EINVAL;
  if (L > 512) return MOY_ENOSYS;    // score registers: 32 key tiles per lane
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    if (L <= 128) return mha_launch_m<T, 8>(qkv, ld_qkv, B, L, nh, E, out, ldo, n_prefix, split, st);
    if (L <= 320) return mha_launch_m<T, 20>(qkv, ld_qkv, B, L, nh, E, out, ldo, n_prefix, split, st);
    return mha_launch_m<T, 32>(qkv, ld_qkv, B, L, nh, E, out, ldo, n_prefix, split, st);
  })
}

extern "C" int moy_mha_core_masked(const void* qkv, int64_t ld_qkv, int B, int L, int nh, int E, const int32_t* n_prefix, int split,
                                   void* out, int64_t ldo, int dtype, void* stream) {
  if (!n_prefix) return MOY_EINVAL;
  return mha_host(qkv, ld_qkv, B, L, nh, E, n_prefix, split, out, ldo, dtype, stream);
}

extern "C" int moy_mha_core(const void* qkv, int64_t ld_qkv, int B, int L, int nh, int E, void* out, int64_t ldo, int dtype,
                            void* stream) {
  return mha_host(qkv, ld_qkv, B, L, nh, E, nullptr, L, out, ldo, dtype, stream);
}

// The paired-corner kernel when the layout allows it (16-bit head planes); MOY_ENOSYS = use msda_fused_kernel.
template <typename T>
static int try_msda_planes(const void* value, int64_t ldv, int64_t head_stride, int S, const LevelInfo& lv, int L, const float* offaw, int64_t ld_oa,
                           const float* ref, int Lq, int nrows, void* out, int64_t ldo, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    static const int planes = knob("MOY_MSDA_PLANES", 1);                  // MOY_MSDA_PLANES=0: the one-corner-per-load kernel on every layout (A/B runs)
    if (planes && ldv == 32 && head_stride * 8 * 2 <= 0x7fffffffLL && (int64_t)S * 64 <= 0x0fffffffLL && aligned16(value) && aligned16(out) &&
        (ldo % 8) == 0 && (head_stride % 8) == 0) {
      if (planes == 3)
        hipLaunchKernelGGL((msda_planes_kernel<T, 2>), dim3((nrows + 3) / 4), dim3(256), 0, st, static_cast<const T*>(value), head_stride, S, lv, L,
                           offaw, ld_oa, ref, Lq, nrows, static_cast<T*>(out), ldo);
      else if (planes == 2)
        hipLaunchKernelGGL((msda_planes_kernel<T, 0>), dim3((nrows + 3) / 4), dim3(256), 0, st, static_cast<const T*>(value), head_stride, S, lv, L,
                           offaw, ld_oa, ref, Lq, nrows, static_cast<T*>(out), ldo);
      else
        hipLaunchKernelGGL((msda_planes_kernel<T, 1>), dim3((nrows + 3) / 4), dim3(256), 0, st, static_cast<const T*>(value), head_stride, S, lv, L,
                           offaw, ld_oa, ref, Lq, nrows, static_cast<T*>(out), ldo);
      return launch_status();
    }
  }
  return MOY_ENOSYS;
}

extern "C" int moy_msda_fused(const void* value, int64_t ldv, int64_t head_stride, int B, int S, const int32_t* shapes_hw, int L, const float* offaw,
                              int64_t ld_oa, const float* ref, int Lq, void* out, int64_t ldo, int dtype, void* stream) {
  if (!value || !shapes_hw || !offaw || !ref || !out || B <= 0 || S <= 0 || L <= 0 || L > 4 || Lq <= 0) return MOY_EINVAL;
  if (head_stride < 32 || (head_stride % 4) || ldv < (head_stride == 32 ? 256 : 32) || (ldv % 4) || ld_oa < 8 * L * 4 * 3 || (ld_oa % 4) || ldo < 256 || (ldo % 4) || !aligned16(ref)) return MOY_EINVAL;
  LevelInfo lv{};
  int s = 0;
  for (int l = 0; l < L; ++l) {
    lv.H[l] = shapes_hw[2 * l]; lv.W[l] = shapes_hw[2 * l + 1]; lv.start[l] = s;
    if (lv.H[l] <= 0 || lv.W[l] <= 0) return MOY_EINVAL;
    s += lv.H[l] * lv.W[l];
  }
  if (s != S) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nrows = B * Lq;
  MOY_DISPATCH_T(dtype, {
    if (reinterpret_cast<uintptr_t>(value) % (4 * sizeof(T)) || !aligned16(offaw) ||
        reinterpret_cast<uintptr_t>(out) % (4 * sizeof(T)))
      return MOY_EINVAL;
    {
      const int rc = try_msda_planes<T>(value, ldv, head_stride, S, lv, L, offaw, ld_oa, ref, Lq, nrows, out, ldo, st);
      if (rc != MOY_ENOSYS) return rc;
    }
    hipLaunchKernelGGL((msda_fused_kernel<T>), dim3((nrows + 3) / 4), dim3(256), 0, st, static_cast<const T*>(value), ldv, head_stride, S, lv, L,
                       offaw, ld_oa, ref, Lq, nrows, static_cast<T*>(out), ldo);
    return launch_status();
  })
}

template <typename T>
static int msda_generic(const void* value, const int64_t* shapes, const int64_t* lstart, const void* loc, const void* aw, int N,
                        int S, int M, int D, int L, int Lq, int P, void* out, void* stream) {
  if (!value || !shapes || !lstart || !loc || !aw || !out) return MOY_EINVAL;
  if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return MOY_EINVAL;
  const long total = (long)N * Lq * M * D;
  hipLaunchKernelGGL((msda_generic_kernel<T>), dim3(nblk(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const T*>(value), shapes, lstart, static_cast<const T*>(loc), static_cast<const T*>(aw), N, S, M,
                     D, L, Lq, P, static_cast<T*>(out));
  return launch_status();
}

extern "C" int moy_msda_fwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                const float* sampling_loc, const float* attn_weight, int N, int S, int M, int D, int L, int Lq,
                                int P, float* out, void* stream) {
  return msda_generic<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, stream);
}

extern "C" int moy_msda_fwd_bf16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                 const void* sampling_loc, const void* attn_weight, int N, int S, int M, int D, int L, int Lq,
                                 int P, void* out, void* stream) {
  return msda_generic<bf16_t>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, stream);
}

extern "C" int moy_msda_fwd_f16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                const void* sampling_loc, const void* attn_weight, int N, int S, int M, int D, int L, int Lq,
                                int P, void* out, void* stream) {
  return msda_generic<f16_t>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, stream);
}

extern "C" int moy_msda_fwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                const double* sampling_loc, const double* attn_weight, int N, int S, int M, int D, int L, int Lq,
                                int P, double* out, void* stream) {
  return msda_generic<double>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out, stream);
}

template <typename T>
static int msda_bwd(const T* value, const int64_t* shapes, const int64_t* lstart, const T* loc, const T* aw, const T* gout, int N, int S,
                    int M, int D, int L, int Lq, int P, T* gvalue, T* gloc, T* gaw, void* stream) {
  if (!value || !shapes || !lstart || !loc || !aw || !gout || !gvalue || !gloc || !gaw) return MOY_EINVAL;
  if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // grad_value accumulates: the reference allocates it with zeros_like (ms_deform_attn_cuda.cu:107); here the callee clears it
  if (hipMemsetAsync(gvalue, 0, sizeof(T) * (size_t)N * S * M * D, st) != hipSuccess) return MOY_ELAUNCH;
  int G = 1;
  while (G < D && G < 64) G <<= 1;
  const long total = (long)N * Lq * M * L * P;
  const long per_block = 256 / G;
  hipLaunchKernelGGL((msda_bwd_kernel<T>), dim3((unsigned)((total + per_block - 1) / per_block)), dim3(256), 0, st, value, shapes, lstart,
                     loc, aw, gout, N, S, M, D, L, Lq, P, G, gvalue, gloc, gaw);
  return launch_status();
}

extern "C" int moy_msda_bwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                const float* sampling_loc, const float* attn_weight, const float* grad_output, int N, int S, int M,
                                int D, int L, int Lq, int P, float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                                void* stream) {
  return msda_bwd<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, N, S, M, D, L, Lq, P,
                         grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

extern "C" int moy_msda_bwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                const double* sampling_loc, const double* attn_weight, const double* grad_output, int N, int S, int M,
                                int D, int L, int Lq, int P, double* grad_value, double* grad_sampling_loc, double* grad_attn_weight,
                                void* stream) {
  return msda_bwd<double>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, N, S, M, D, L, Lq, P,
                          grad_value, grad_sampling_loc, grad_attn_weight, stream);
}

extern "C" int moy_resize_linear_u8(const uint8_t* src, int B, int Hs, int Ws, int64_t src_row_bytes, int64_t src_img_bytes,
                                    uint8_t* dst, int Hd, int Wd, void* stream) {
  if (!src || !dst || B <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || (Wd & 3)) return MOY_EINVAL;
  if (src_row_bytes < (int64_t)Ws * 3 || src_img_bytes < src_row_bytes * Hs) return MOY_EINVAL;
  if (reinterpret_cast<uintptr_t>(dst) & 3) return MOY_EINVAL;
  // cv::resize: inv_scale = dsize / ssize (double), scale = 1 / inv_scale
  const double scale_x = 1.0 / ((double)Wd / (double)Ws), scale_y = 1.0 / ((double)Hd / (double)Hs);
  const int area2 = (Ws == 2 * Wd && Hs == 2 * Hd) ? 1 : 0;      // INTER_LINEAR -> INTER_AREA for the exact 2x shrink
  const long total = (long)B * Hd * (Wd >> 2);
  hipLaunchKernelGGL(resize_linear_u8_kernel, dim3(nblk(total)), dim3(256), 0, static_cast<hipStream_t>(stream), src, B, Hs, Ws,
                     (long)src_row_bytes, (long)src_img_bytes, dst, Hd, Wd, scale_x, scale_y, area2);
  return launch_status();
}

extern "C" int moy_box_iou(const float* a, const float* b, int T, int n, int K, const int32_t* na, const int32_t* nb, float* out,
                           void* stream) {
  if (!a || !b || !out || T <= 0 || n <= 0 || K <= 0) return MOY_EINVAL;
  if (!aligned16(a) || !aligned16(b)) return MOY_EINVAL;
  hipLaunchKernelGGL(box_iou_kernel, dim3(nblk((long)T * n * K)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, T, n, K, na, nb,
                     out);
  return launch_status();
}

extern "C" int moy_assign_post(const float* logits, const float* boxes, int B, int nq, int nc, float score_thresh, float conf,
                               float img_w, float img_h, float* y, float* scores, int64_t* obj_idxes, float* rows,
                               int64_t* track_id, int32_t* n_rows, int32_t* n_ids, void* stream) {
  if (!logits || !boxes || !y || !scores || !obj_idxes || !rows || !track_id || !n_rows || !n_ids) return MOY_EINVAL;
  if (B <= 0 || nq <= 0 || nq > 1024 || nc <= 0) return MOY_EINVAL;
  hipLaunchKernelGGL(assign_post_kernel, dim3(B), dim3(1024), 0, static_cast<hipStream_t>(stream), logits, boxes, nq, nc,
                     score_thresh, conf, img_w, img_h, y, scores, obj_idxes, rows, track_id, n_rows, n_ids);
  return launch_status();
}

extern "C" int moy_temporal_assemble(const void* trk_embed, const void* trk_qpos, const float* trk_ref, const int32_t* n_trk,
                                     const void* det_embed, int64_t ld_de, const void* det_qpos, int64_t ld_dq, const float* det_ref,
                                     int B, int n_max, int nq, void* embed, int64_t ld_e, void* qpos, int64_t ld_q, float* ref_logit,
                                     float* ref_sig, int dtype, void* stream) {
  if (!trk_embed || !trk_qpos || !trk_ref || !n_trk || !det_embed || !det_qpos || !det_ref || !embed || !qpos || !ref_logit || !ref_sig)
    return MOY_EINVAL;
  if (B <= 0 || n_max <= 0 || nq <= 0 || ld_de < 256 || ld_dq < 256 || ld_e < 256 || ld_q < 256) return MOY_EINVAL;
  if ((ld_de % 8) || (ld_dq % 8) || (ld_e % 8) || (ld_q % 8)) return MOY_EINVAL;
  if (!aligned16(trk_embed) || !aligned16(trk_qpos) || !aligned16(trk_ref) || !aligned16(det_embed) || !aligned16(det_qpos) ||
      !aligned16(det_ref) || !aligned16(embed) || !aligned16(qpos) || !aligned16(ref_logit) || !aligned16(ref_sig))
    return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    const long total = (long)B * (n_max + nq) * (256 / DT<T>::KPB);
    hipLaunchKernelGGL((temporal_assemble_kernel<T>), dim3(nblk(total)), dim3(256), 0, st, static_cast<const T*>(trk_embed),
                       static_cast<const T*>(trk_qpos), trk_ref, n_trk, static_cast<const T*>(det_embed), ld_de,
                       static_cast<const T*>(det_qpos), ld_dq, det_ref, B, n_max, nq, static_cast<T*>(embed), ld_e, static_cast<T*>(qpos),
                       ld_q, ref_logit, ref_sig);
    return launch_status();
  })
}

extern "C" int moy_temporal_assign(const float* logits, const float* boxes, int B, int n_max, int nq, int nc, const int64_t* trk_id,
                                   const int32_t* trk_dis, const int32_t* n_trk, int64_t* max_obj_id, float score_thresh,
                                   float filter_thresh, int miss_tol, float conf, float img_w, float img_h, float* y, float* scores,
                                   int64_t* obj_idxes, int32_t* dis_out, int32_t* sel_rows, int32_t* n_new, int32_t* n_overflow,
                                   float* rows, int64_t* track_id, int32_t* n_rows, int32_t* n_ids, void* stream) {
  if (!logits || !boxes || !trk_id || !trk_dis || !n_trk || !max_obj_id || !y || !scores || !obj_idxes || !dis_out || !sel_rows ||
      !n_new || !n_overflow || !rows || !track_id || !n_rows || !n_ids)
    return MOY_EINVAL;
  if (B <= 0 || n_max <= 0 || nq <= 0 || n_max + nq > 1024 || nc <= 0 || miss_tol <= 0) return MOY_EINVAL;
  hipLaunchKernelGGL(temporal_assign_kernel, dim3(B), dim3(1024), 0, static_cast<hipStream_t>(stream), logits, boxes, n_max, nq, nc,
                     trk_id, trk_dis, n_trk, max_obj_id, score_thresh, filter_thresh, miss_tol, conf, img_w, img_h, y, scores, obj_idxes,
                     dis_out, sel_rows, n_new, n_overflow, rows, track_id, n_rows, n_ids);
  return launch_status();
}

extern "C" int moy_temporal_commit(const int32_t* sel_rows, const int32_t* n_new, const int64_t* obj_idxes, const int32_t* dis_out,
                                   const float* boxes, int B, int n_max, int64_t* trk_id, int32_t* trk_dis, float* trk_ref,
                                   int32_t* n_trk, void* stream) {
  if (!sel_rows || !n_new || !obj_idxes || !dis_out || !boxes || !trk_id || !trk_dis || !trk_ref || !n_trk) return MOY_EINVAL;
  if (B <= 0 || n_max <= 0 || !aligned16(boxes) || !aligned16(trk_ref)) return MOY_EINVAL;
  hipLaunchKernelGGL(temporal_commit_kernel, dim3(nblk((long)B * n_max)), dim3(256), 0, static_cast<hipStream_t>(stream), sel_rows, n_new,
                     obj_idxes, dis_out, boxes, B, n_max, trk_id, trk_dis, trk_ref, n_trk);
  return launch_status();
}

extern "C" int moy_track_state_update(const float* scores, const float* boxes, const int64_t* obj_idxes, const void* hs, int64_t ld_hs,
                                      int B, int nq, int32_t* copy_rows, int64_t* copy_ids, int32_t* n_copy, float* mem, float* conf,
                                      int64_t* ids, float* fboxes, int32_t* low, int32_t* pool, int32_t pool_cap, int32_t* pool_hc,
                                      int dtype, void* stream) {
  if (!scores || !boxes || !obj_idxes || !hs || !copy_rows || !copy_ids || !n_copy || !mem || !conf || !ids || !fboxes || !low ||
      !pool || !pool_hc)
    return MOY_EINVAL;
  if (B <= 0 || nq <= 0 || nq > 1024 || ld_hs < FSQM_DIM || pool_cap < FSQM_SLOTS) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((track_state_kernel<T>), dim3(1), dim3(1024), 0, st, scores, boxes, obj_idxes, static_cast<const T*>(hs), ld_hs, B,
                       nq, copy_rows, copy_ids, n_copy, mem, conf, ids, fboxes, low, pool, pool_cap, pool_hc);
    return launch_status();
  })
}

extern "C" int moy_fsqm_reset(float* mem, float* conf, int64_t* ids, float* fboxes, int32_t* low, int32_t* pool, int32_t pool_cap,
                              int32_t* pool_hc, void* stream) {
  if (!mem || !conf || !ids || !fboxes || !low || !pool || !pool_hc || pool_cap < FSQM_SLOTS) return MOY_EINVAL;
  hipLaunchKernelGGL(fsqm_reset_kernel, dim3(nblk(FSQM_SLOTS * FSQM_DIM)), dim3(256), 0, static_cast<hipStream_t>(stream), mem, conf,
                     ids, fboxes, low, pool, pool_cap, pool_hc);
  return launch_status();
}

extern "C" int moy_detect_decode(const void* box, int64_t ld_box, const void* cls, int64_t ld_cls, int B, int h, int w, int nc,
                                 float stride, int a_off, int A, float* y, int dtype, void* stream) {
  if (!box || !cls || !y || B <= 0 || h <= 0 || w <= 0 || nc <= 0 || a_off < 0 || a_off + h * w > A) return MOY_EINVAL;
  if (ld_box < 64 || (ld_box % 4) || ld_cls < nc) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    if (reinterpret_cast<uintptr_t>(box) % (4 * sizeof(T))) return MOY_EINVAL;
    hipLaunchKernelGGL((detect_decode_kernel<T>), dim3(nblk((long)B * h * w)), dim3(256), 0, st, static_cast<const T*>(box), ld_box,
                       static_cast<const T*>(cls), ld_cls, B, h, w, nc, stride, a_off, A, y);
    return launch_status();
  })
}

extern "C" int moy_nms(const float* y, int B, int nc, int A, float conf_thres, float iou_thres, int max_det, float max_wh, float gain,
                       float pad_x, float pad_y, float clip_w, float clip_h, float* rows, int32_t* n_rows, void* stream) {
  if (!y || !rows || !n_rows || B <= 0 || nc <= 0 || A <= 0 || A > 16384 || max_det <= 0 || !(gain > 0.f)) return MOY_EINVAL;
  int P2 = 1;
  while (P2 < A) P2 <<= 1;
  const size_t lds = (size_t)P2 * 9;
  static bool attr_set = false;   // the kernel also has a few static __shared__ words: ask for what the largest A needs
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 9) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(NMS_THREADS), lds, static_cast<hipStream_t>(stream), y, nc, A, P2, conf_thres, iou_thres,
                     max_det, max_wh, gain, pad_x, pad_y, clip_w, clip_h, rows, n_rows);
  return launch_status();
}

extern "C" int moy_cast_f32_to(const float* src, int64_t lds_, int M, int N, void* dst, int64_t ldd, int dtype, void* stream) {
  if (!src || !dst || M <= 0 || N <= 0 || (N % 4) || (lds_ % 4) || (ldd % 4) || !aligned16(src)) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((cast_kernel<T>), dim3(nblk((long)M * (N / 4))), dim3(256), 0, st, src, lds_, M, N / 4, static_cast<T*>(dst), ldd);
    return launch_status();
  })
}

extern "C" int moy_gather_rows(const void* src, int64_t lds_, const int32_t* rows, int M, int N, void* dst, int64_t ldd, int dtype,
                               void* stream) {
  if (!src || !rows || !dst || M <= 0 || N <= 0 || !aligned16(src) || !aligned16(dst)) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    constexpr int KPB = DT<T>::KPB;                     // whole 16-byte chunks: 8 columns of a 16-bit type, 4 of fp32
    if ((N % KPB) || (lds_ % KPB) || (ldd % KPB)) return MOY_EINVAL;
    const int NC = N / KPB;
    hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(nblk((long)M * NC)), dim3(256), 0, st, static_cast<const T*>(src), lds_, rows,
                       M, NC, static_cast<T*>(dst), ldd);
    return launch_status();
  })
}

extern "C" int moy_level_rows(const int32_t* tok_local, int B, int nq, int n_levels, const int32_t* level_hw, int32_t* rows,
                              int32_t* level, void* stream) {
  if (!tok_local || !level_hw || !rows || !level || B <= 0 || nq <= 0 || n_levels < 1 || n_levels > 4) return MOY_EINVAL;
  LevelTable lv{};
  lv.n = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    if (level_hw[l] <= 0) return MOY_EINVAL;
    lv.hw[l] = level_hw[l];
    lv.off[l + 1] = lv.off[l] + level_hw[l];
  }
  hipLaunchKernelGGL(level_rows_kernel, dim3(nblk((long)B * nq)), dim3(256), 0, static_cast<hipStream_t>(stream), tok_local, B, nq, lv, rows,
                     level);
  return launch_status();
}

extern "C" int moy_level_select(const float* G, int64_t level_stride, int64_t ldg, const int32_t* level, const float* shift,
                                const int32_t* tok_local, const uint8_t* valid, int M, int N, void* dst, int64_t ldd, int dtype,
                                void* stream) {
  if (!G || !level || !shift || !dst || M <= 0 || N != 256 || (ldg % 4) || (ldd % 4) || !aligned16(G) || !aligned16(shift)) return MOY_EINVAL;
  if (valid && !tok_local) return MOY_EINVAL;
  if (dtype != MOY_BF16 && dtype != MOY_F16 && dtype != MOY_F32 && dtype != MOY_F32X3) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == MOY_F32 || dtype == MOY_F32X3)            // (round 6: the fp32 engines' folded head)
    hipLaunchKernelGGL((level_select_kernel<float>), dim3(nblk((long)M * (N / 4))), dim3(256), 0, st, G, level_stride, ldg, level, shift,
                       tok_local, valid, M, N / 4, static_cast<float*>(dst), ldd);
  else if (dtype == MOY_BF16)
    hipLaunchKernelGGL((level_select_kernel<bf16_t>), dim3(nblk((long)M * (N / 4))), dim3(256), 0, st, G, level_stride, ldg, level, shift,
                       tok_local, valid, M, N / 4, static_cast<bf16_t*>(dst), ldd);
  else
    hipLaunchKernelGGL((level_select_kernel<f16_t>), dim3(nblk((long)M * (N / 4))), dim3(256), 0, st, G, level_stride, ldg, level, shift,
                       tok_local, valid, M, N / 4, static_cast<f16_t*>(dst), ldd);
  return launch_status();
}

extern "C" int moy_query_order(const float* ref, int B, int Lq, int H0, int W0, int32_t* perm, void* stream) {
  if (!ref || !perm || B <= 0 || Lq <= 0 || Lq > 1024 || H0 <= 0 || W0 <= 0) return MOY_EINVAL;
  int P2 = 1;
  while (P2 < Lq) P2 <<= 1;
  hipLaunchKernelGGL(query_order_kernel, dim3(B), dim3(512), 0, static_cast<hipStream_t>(stream), ref, Lq, H0, W0, P2, perm);
  return launch_status();
}

extern "C" int moy_sigmoid_f32(const float* in, int n, float* out, void* stream) {
  if (!in || !out || n <= 0) return MOY_EINVAL;
  hipLaunchKernelGGL(sigmoid_kernel, dim3(nblk(n)), dim3(256), 0, static_cast<hipStream_t>(stream), in, n, out);
  return launch_status();
}

extern "C" int moy_msda_prep(const float* offaw, int64_t ld, int col_off, int col_aw, const float* ref, int refdim, int rows,
                             int n_heads, int n_levels, int n_points, const int32_t* shapes_hw, int sigmoid_attn, void* loc,
                             void* aw, int dtype, void* stream) {
  if (!offaw || !ref || !loc || !aw || !shapes_hw || rows <= 0 || n_heads <= 0 || n_levels <= 0 || n_levels > 8 || n_points <= 0)
    return MOY_EINVAL;
  if (refdim != 2 && refdim != 4) return MOY_EINVAL;
  if (dtype != MOY_F32 && dtype != MOY_BF16) return MOY_ENOSYS;    // the operator entries that consume loc / aw: fp32, bf16 (fp64: moy_msda_fwd_f64 takes caller-made operands)
  PrepLevels lv{};
  for (int l = 0; l < n_levels; ++l) { lv.h[l] = shapes_hw[2 * l]; lv.w[l] = shapes_hw[2 * l + 1]; if (lv.h[l] <= 0 || lv.w[l] <= 0) return MOY_EINVAL; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)(((long)rows * n_heads + 255) / 256);
  if (dtype == MOY_F32)
    hipLaunchKernelGGL((msda_prep_kernel<float>), dim3(blocks), dim3(256), 0, st, offaw, ld, col_off, col_aw, ref, refdim, rows, n_heads,
                       n_levels, n_points, lv, sigmoid_attn, static_cast<float*>(loc), static_cast<float*>(aw));
  else
    hipLaunchKernelGGL((msda_prep_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, offaw, ld, col_off, col_aw, ref, refdim, rows, n_heads,
                       n_levels, n_points, lv, sigmoid_attn, static_cast<bf16_t*>(loc), static_cast<bf16_t*>(aw));
  return launch_status();
}

extern "C" int moy_mask_rows(void* x, int64_t ld, int M, int N, const uint8_t* mask, int dtype, void* stream) {
  if (!x || !mask || M <= 0 || N <= 0 || !aligned16(x)) return MOY_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MOY_DISPATCH_T(dtype, {
    constexpr int KPB = DT<T>::KPB;
    if ((N % KPB) || (ld % KPB)) return MOY_EINVAL;
    hipLaunchKernelGGL((mask_rows_kernel<T>), dim3(nblk((long)M * (N / KPB))), dim3(256), 0, st, static_cast<T*>(x), ld, M, N / KPB, mask);
    return launch_status();
  })
}

