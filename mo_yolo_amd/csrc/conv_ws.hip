// Weight-stationary direct 3x3 convolution (stride 1, pad 1, Cin == Cout in {32, 64, 128}) + BN + SiLU (+ residual) for 16-bit
// types: the Bottleneck convs of the C2f blocks (ultralytics/nn/modules/block.py:271-283 over conv.py:36-38).
//
// Why a third convolution kernel.  The tile-per-block direct kernel (gemm.hip: conv3x3_direct_kernel) re-stages the weight taps
// through LDS for every 128-256 output pixels (one barrier per tap), loads its halo patch with nothing else in flight and ran at
// 0.25-0.4 of its floor (profiles/r01_*).  Here a block lives for the whole launch (one per CU, 8 waves):
//   * the weights never touch LDS: wave (wn, wm) keeps W[its 16*NTw output channels][9 taps][C] in REGISTERS as MFMA A-operand
//     fragments (C = 128: 144 VGPRs, C = 64: 72, C = 32: 36) -- no weight traffic and no barrier inside a tile;
//   * halo patches ((TH+2) x 18 pixels x C) and, for the shortcut, the residual tile arrive by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`: no VGPRs, no ds_write) into a ring of NBUF buffer sets, DIST = NBUF-1 tiles ahead of the
//     MFMAs, retired by a counted vmcnt that leaves the younger pieces and the output stores in flight.  Pixels outside the image
//     are out-of-range buffer offsets: the DMA writes zeros (= the conv's zero padding; probed: tools/probes/ldsdma_oob.hip);
//   * the LDS image of a patch pixel is its C/8 16-byte chunks with the chunk index XORed by a function of the pixel (the
//     conflict-free maps of gemm.hip); the DMA destination is lane-linear, so the XOR is applied to the per-lane SOURCE address;
//   * a fragment row (16 pixels x 32 channels) read for tap column kx serves the three tap rows ky of up to three output rows:
//     (MT+2) ds_read_b128 per 3*MT MFMAs instead of one read per MFMA;
//   * epilogue: BN + SiLU on the accumulators -> tile in the output type in LDS (for the shortcut: added in place to the
//     DMA'd residual tile, one rounding) -> whole-line 16-byte stores through a per-image descriptor (pixels past the image edge
//     are out-of-range offsets, so the store count per wave is constant and the vmcnt bookkeeping exact);
//   * tiles are dealt so that the blocks of one XCD work on 32 consecutive tiles at a time: halo pixels shared by neighbouring
//     tiles are served by that XCD's L2.
#include "common.hpp"

#include <type_traits>
#include <utility>

#ifndef MOY_CWS_HOIST
#define MOY_CWS_HOIST 1
#endif
#ifndef MOY_CWS_PREGEOM
#define MOY_CWS_PREGEOM 0         // measured (C = 128, 288 frames): the loop top shrinks 1 900 -> 560 cycles per tile, the kernel 244 -> 241 us at
#endif                            // 38x68 and 96 -> 101 us at 19x34: the cycles reappear as barrier waits (the younger wave of each SIMD is the critical path)
#ifndef MOY_CWS_FRAGBASE
#define MOY_CWS_FRAGBASE 1        // round 4: fragment addresses = one of 8 per-lane bases (^ a constant) + an immediate offset, see conv_ws_kernel
#endif
#ifndef MOY_CWS_HOIST_128
#define MOY_CWS_HOIST_128 0       // measured: 255 VGPRs, 247.3 vs 247.1 us -- nothing (hipcc re-derives the values anyway); 81 spills with the residual
#endif

namespace moy {

struct ConvWsParams {
  const void* A; int64_t lda;
  const void* W; int Kpad;
  const float* scale; const float* shift;
  const void* R; int64_t ldr;
  void* C; int64_t ldc;
  int H, Wd;                        // image size (stride 1: input == output)
  int tiles_x, tiles_img, ntiles;
  int per_xcd, bpx;                 // tiles per XCD chunk, blocks per XCD
  FastDiv fd_timg, fd_tx;
  int prio;                         // ping-pong form: raise the wave priority during its MFMA halves (MOY_CWS_PRIO, default 1)
  // GRP forms (round 4): G images side by side on one VIRTUAL row, one zero column between neighbours (= each image's padding), so
  // that 16-column tiles are cut from G * (Wd + 1) - 1 columns instead of from Wd: 68 columns are 5 tiles alone (85 % of a tile
  // row used) and 13 tiles per three images (95 %).  A tile batch index is then a GROUP; virtual column xv -> image g = xv / (Wd + 1)
  // of the group, column xv - g * (Wd + 1) (== Wd: the zero column); the descriptor of a group spans its images.
  int Wv, G, nimg;
  FastDiv fd_w1;
  int delta_a, delta_r, delta_c;    // bytes from virtual column xv of image 0 to the same pixel of image g, per g: image bytes - (Wd + 1) pixels
};

template <int C>
__device__ __forceinline__ int cws_swz(int pix) {   // same maps as conv_swz in gemm.hip
  return C == 32 ? ((pix >> 1) & 3) : (C == 64 ? (pix & 7) : ((pix & 7) << 1));
}

// LDS-DMA of 16 bytes per lane: LDS[lds_dst + lane*16 ..] <- buffer[voff] (zeros when voff fails the range check).
__device__ __forceinline__ void cws_dma16(uint32_t voff, __amdgpu_buffer_rsrc_t rs, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds_dst), "s"(rs)
               : "memory");
}

template <int N>
__device__ __forceinline__ void cws_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// End-of-tile wait of the patch ring (see wait_ring_tile in gemm_wreg.hip): behind the pieces of tile it+1 lie DIST-1 younger piece
// groups and one group of NPASS stores per tile ALREADY COMPUTED -- min(DIST, it+1) of them.  Round 5: the steady-state count used
// from the first tile on left the NPASS youngest pieces of tile 1 unwaited.
template <int DIST, int IPW, int NPASS>
__device__ __forceinline__ void cws_wait_ring_tile(int it) {   // `it` wave-uniform
  if constexpr (DIST >= 2) {
    if (it == 0) { cws_wait_vmcnt<(DIST - 1) * IPW + NPASS>(); return; }
  }
  if constexpr (DIST >= 3) {
    if (it == 1) { cws_wait_vmcnt<(DIST - 1) * IPW + 2 * NPASS>(); return; }
  }
  static_assert(DIST <= 3, "ring depth");
  cws_wait_vmcnt<(DIST - 1) * IPW + DIST * NPASS>();
}

template <typename T>
__device__ __forceinline__ f32x4 cws_mfma(f32x4 acc, u32x4 w, u32x4 a);
template <>
__device__ __forceinline__ f32x4 cws_mfma<bf16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 cws_mfma<f16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}

template <int C, int N, int TH, int NBUF, bool RES>
struct CwsGeom {
  static constexpr int PH = TH + 2, PW = 18, NPIX = PH * PW;
  static constexpr int CPP = C / 8, NCP = N / 8;            // 16-byte chunks per input / output pixel
  static constexpr int PATCH_PIECES = (NPIX * CPP + 63) / 64;
  static constexpr int TPX = TH * 16;
  static constexpr int RES_PIECES = RES ? TPX * NCP / 64 : 0;
  static constexpr int PIECES = PATCH_PIECES + RES_PIECES;
  static constexpr int IPW = (PIECES + 7) / 8;              // DMA instructions per wave and tile (dummies pad the last round)
  static constexpr int SETB = PIECES * 1024;
  static constexpr int STGB = TPX * N * 2;
  static constexpr int LDS = NBUF * SETB + (RES ? 0 : STGB) + 1024;
  static constexpr int NPASS = TPX * NCP / 512;             // 16-byte stores per thread and tile
};

template <typename T, int C, int N, int TH, int WN, int NBUF, bool RES, int OCC, bool SPREAD, int ABL = 0, bool GRP = false>   // ABL: timing-only ablation builds (tools/bench_gemm.py)
__global__ __launch_bounds__(512, 2 * OCC) void conv_ws_kernel(const ConvWsParams p) {
  using G = CwsGeom<C, N, TH, NBUF, RES>;
  constexpr int PW = G::PW, CPP = G::CPP, NCP = G::NCP, IPW = G::IPW, NPASS = G::NPASS;
  constexpr int WM = 8 / WN, MT = TH / WM, NT = N / 16 / WN, KC = C / 32;
  constexpr int DIST = NBUF - 1;
  constexpr int RG = MT < 4 ? MT : 4;      // output rows per fragment-row group
  constexpr uint32_t OOB = 0x80000000u;
  // round 3: at C <= 64 the kernel has ~100 free registers, so the per-piece DMA geometry (a dozen VALU operations per piece and
  // tile, in the phase where nothing hides them) lives in loop-invariant registers again: 270 -> 248 us at C = 64, 437 -> 413 at
  // C = 32 (288 frames, same-device A/B of two builds); 152 -> 225-256 VGPRs, no spills
  constexpr bool HOIST_GEOM = MOY_CWS_HOIST && (C <= 64 || (MOY_CWS_HOIST_128 && !RES));   // (C = 128 with the residual: 81 spilled VGPRs)
  static_assert(TH % WM == 0 && (N / 16) % WN == 0 && G::TPX * NCP % 512 == 0, "tile vs waves");
  static_assert(sizeof(T) == 2, "16-bit types");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int wn = wave % WN, wm = wave / WN;

  // ---- this block's tiles: XCD x owns the tile range [x * per_xcd, (x + 1) * per_xcd); its blocks take consecutive tiles
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t_first = xcd * p.per_xcd + slot;
  const int t_limit = min((xcd + 1) * p.per_xcd, p.ntiles);
  if (t_first >= t_limit) return;
  const int n_mine = (t_limit - t_first + p.bpx - 1) / p.bpx;
  // static priority for the younger wave of every SIMD (MI355X_MICROARCH.md, "two waves per SIMD", item 4): with equal priority the
  // arbiter serves the older wave first and waves 4-7 become the critical path of every tile (MOY_CWS_PRIO bit 1, A/B knob)
  if ((p.prio & 2) && wave >= 4) __builtin_amdgcn_s_setprio(1);

  const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  const uint32_t scratch = lds_base + NBUF * G::SETB + (RES ? 0 : G::STGB);

  // ---- DMA geometry: piece pc = wave + 8k fills LDS bytes [pc*1024, +1024) of a buffer set.
  // Patch pieces: chunk X = pc*64 + lane -> patch pixel X / CPP, LDS slot X % CPP <- source chunk slot ^ swz(pixel).
  // Residual pieces: chunk X -> output pixel X / NCP of the tile, slot X % NCP <- source chunk slot ^ (pixel & (NCP-1)).
  // Recomputed per tile (a dozen VALU operations per piece in the shadow of the MFMAs) instead of held in 2*IPW registers.
  auto piece_geom = [&](int pc, int& dy, int& dx, int& rel) {   // offsets relative to the tile origin; dy = -4096: never valid
    dy = -4096; dx = 0; rel = 0;
    int lv = lane;
    if constexpr (!HOIST_GEOM) asm volatile("" : "+v"(lv));   // opaque: keeps the per-piece geometry out of loop-invariant registers
    if (pc < G::PATCH_PIECES) {
      const int X = pc * 64 + lv, pix = X / CPP, sl = X % CPP;
      const int py = pix / PW, px = pix - py * PW;
      if (pix < G::NPIX) dy = py - 1;
      dx = px - 1;
      rel = ((dy * p.Wd + dx) * (int)p.lda + ((sl ^ cws_swz<C>(pix)) * 8)) * 2;
    } else if (RES && pc < G::PIECES) {
      const int X = (pc - G::PATCH_PIECES) * 64 + lv, opx = X / NCP, sl = X % NCP;
      dy = opx >> 4; dx = opx & 15;
      rel = ((dy * p.Wd + dx) * (int)p.ldr + ((sl ^ (opx & (NCP - 1))) * 8)) * 2;
    }
  };

  const T* __restrict__ Ag = static_cast<const T*>(p.A);
  const T* __restrict__ Rg = static_cast<const T*>(p.R);
  T* __restrict__ Cg = static_cast<T*>(p.C);
  const int64_t img_a = (int64_t)p.H * p.Wd * p.lda, img_r = (int64_t)p.H * p.Wd * p.ldr, img_c = (int64_t)p.H * p.Wd * p.ldc;

  auto tile_coords = [&](int t, int& b, int& y0, int& x0) {
    b = (int)fdiv((uint32_t)t, p.fd_timg);
    const int rem = t - b * p.tiles_img;
    const int ty = (int)fdiv((uint32_t)rem, p.fd_tx);
    y0 = ty * TH;
    x0 = (rem - ty * p.tiles_x) * 16;
  };

  // virtual column -> (valid, byte correction) of a pixel of the group (GRP) or plain range test
  auto col_ok = [&](int xx, int delta, int& corr) {
    corr = 0;
    if constexpr (GRP) {
      if ((unsigned)xx >= (unsigned)p.Wv) return false;
      const int g = (int)fdiv((uint32_t)xx, p.fd_w1);
      corr = g * delta;
      return xx - g * (p.Wd + 1) < p.Wd;
    } else {
      return (unsigned)xx < (unsigned)p.Wd;
    }
  };
  // DMA of one tile = scalar set-up (descriptors, tile origin) + IPW independent pieces, so the pieces can be issued all at
  // once (prologue) or one by one between the MFMA groups of the tile being computed (SPREAD).
  struct TileDma {
    __amdgpu_buffer_rsrc_t rsA, rsR;
    int y0, x0, off_a, off_r;
    uint32_t dst;
    bool live;
  };
  auto tile_setup = [&](int it, int set) {
    TileDma d;
    d.live = it < n_mine;                                       // tiles past the end: every lane out of range (uniform counts)
    int b;
    tile_coords(min(t_first + it * p.bpx, p.ntiles - 1), b, d.y0, d.x0);
    // (GRP: b is a group; its descriptor spans the group's images -- the last group may hold fewer, what lies past them reads as zero)
    const int b0 = GRP ? b * p.G : b, nim = GRP ? min(p.G, p.nimg - b0) : 1;
    d.rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag + (int64_t)b0 * img_a), 0, (uint32_t)(img_a * 2 * nim), 0x00020000);
    d.rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((RES ? Rg : Ag) + (int64_t)b0 * (RES ? img_r : img_a)), 0,
                                              (uint32_t)((RES ? img_r : img_a) * 2 * nim), 0x00020000);
    const int org = d.y0 * p.Wd + d.x0;
    d.off_a = org * (int)p.lda * 2;
    d.off_r = org * (int)p.ldr * 2;
    d.dst = lds_base + set * G::SETB;
    return d;
  };
  auto issue_piece = [&](const TileDma& d, int k) {
    const int pc = wave + 8 * k;                                // wave-uniform
    const bool is_res = RES && pc >= G::PATCH_PIECES;
    int dy, dx, rel;
    piece_geom(pc, dy, dx, rel);
    const int yy = d.y0 + dy, xx = d.x0 + dx;
    int corr;
    const bool cok = col_ok(xx, is_res ? p.delta_r : p.delta_a, corr);
    const bool ok = d.live && (unsigned)yy < (unsigned)p.H && cok;
    const uint32_t voff = ok ? (uint32_t)((is_res ? d.off_r : d.off_a) + rel + corr) : OOB;
    const uint32_t dst = pc < G::PIECES ? d.dst + pc * 1024 : scratch;
    if (ABL == 2) return;
    if (is_res) cws_dma16(voff, d.rsR, dst);
    else cws_dma16(voff, d.rsA, dst);
  };
  auto issue_tile = [&](int it, int set) {
    const TileDma d = tile_setup(it, set);
#pragma unroll
    for (int k = 0; k < IPW; ++k) issue_piece(d, k);
  };
  // PREGEOM (round 3, where HOIST_GEOM has no registers: C = 128): the lane offsets of the NEXT tile's pieces are formed in the
  // shadow of this tile's MFMAs (the matrix pipe is the bound there, the vector slots are free) and wait in IPW registers; the loop
  // top then only issues -- it used to spend ~20 vector operations per piece with nothing to hide them
  constexpr bool PREGEOM = MOY_CWS_PREGEOM && !HOIST_GEOM && !SPREAD;
  uint32_t vo[IPW];
  auto comp_vo = [&](const TileDma& d) {
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
      const int pc = wave + 8 * k;
      const bool is_res = RES && pc >= G::PATCH_PIECES;
      int dy, dx, rel;
      piece_geom(pc, dy, dx, rel);
      const int yy = d.y0 + dy, xx = d.x0 + dx;
      int corr;
      const bool cok = col_ok(xx, is_res ? p.delta_r : p.delta_a, corr);
      const bool ok = d.live && (unsigned)yy < (unsigned)p.H && cok;
      vo[k] = ok ? (uint32_t)((is_res ? d.off_r : d.off_a) + rel + corr) : OOB;
      asm volatile("" : "+v"(vo[k]));      // materialised HERE (the scheduler would otherwise sink the arithmetic to its use at the loop top)
    }
  };
  auto issue_pre = [&](const TileDma& d) {
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
      const int pc = wave + 8 * k;
      const bool is_res = RES && pc >= G::PATCH_PIECES;
      const uint32_t dst = pc < G::PIECES ? d.dst + pc * 1024 : scratch;
      if (ABL == 2) continue;
      if (is_res) cws_dma16(vo[k], d.rsR, dst);
      else cws_dma16(vo[k], d.rsA, dst);
    }
  };

  // ---- weights of this wave -> registers (MFMA A operand: lane (r, q) holds W[n = .. + r][k = tap*C + cc*32 + q*8 .. +7])
  u32x4 wf[NT][9][KC];
  f32x4 sc[NT], sh[NT];
  {
    const T* Wg = static_cast<const T*>(p.W);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = (wn * NT + j) * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cc = 0; cc < KC; ++cc)
          wf[j][tap][cc] = *reinterpret_cast<const u32x4*>(Wg + (int64_t)(n + r) * p.Kpad + tap * C + cc * 32 + q * 8);
      sc[j] = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n + q * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
      sh[j] = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // ---- store-pass geometry of this thread: chunk X = k*512 + tid of the [pixel][N] tile image (pass k advances 512/NCP pixels,
  // a multiple of 16: the tile column and the swizzle term stay, the tile row advances by 32/NCP)
  const int s_px = tid / NCP, s_c = tid % NCP;
  const int s_ty = s_px >> 4, s_tx = s_px & 15;
  const int s_lds = s_px * (N * 2) + ((s_c ^ (s_px & (NCP - 1))) * 16);
  const int s_rel = ((s_ty * p.Wd + s_tx) * (int)p.ldc + s_c * 8) * 2;
  constexpr int PASS_ROWS = 512 / NCP / 16, PASS_LDS = (512 / NCP) * N * 2;
  const int s_pass_rel = PASS_ROWS * p.Wd * (int)p.ldc * 2;

  // prologue: DIST tiles in flight
#pragma unroll
  for (int d = 0; d < DIST; ++d) issue_tile(d, d);
  cws_wait_vmcnt<(DIST - 1) * IPW>();
  __syncthreads();

  const int pix0 = wm * MT * PW + r;      // patch pixel of this lane for output row 0 of the wave, tap column 0
  // Round 4: fragment read addresses without per-read arithmetic.  The pixel of a read is pix0 + D with D = (row)*PW + kx a compile-time
  // constant, its swizzle term depends on the pixel only through (pix & 7) = (r + e) & 7 with e = D & 7 (wm*MT*PW is a multiple of 8), and
  // (cc*4 + q) ^ s = (cc*4) ^ (q ^ s): so address = (F[e] ^ cc*64) + D*C*2 with EIGHT per-lane registers F[e] = set base + pix0*C*2 +
  // ((q ^ swz(r + e)) << 4) -- one v_xor per read for cc != 0, none for cc = 0, D in the instruction's offset field.  The lock-step loop
  // spent 464 integer instructions per tile on these addresses beside its 288 MFMAs.
  static_assert((MT * PW) % 8 == 0, "the swizzle phase of a wave's first row must not depend on the wave");
  // Measured (288 frames, same device, two interleaved rounds): C = 128 253 -> 237 us (-6.5 %), C = 64 254 -> 263 us (+3.7 %: there the
  // compiler already shared most of the address arithmetic across the unrolled reads and the eight extra registers cost more) -- so C = 128 only.
  constexpr bool FRAGBASE = MOY_CWS_FRAGBASE && C == 128;
  uint32_t fbase[8];
  if constexpr (FRAGBASE) {
#pragma unroll
    for (int e = 0; e < 8; ++e) fbase[e] = (uint32_t)(pix0 * (C * 2) + (((q ^ cws_swz<C>(r + e)) & (CPP - 1)) << 4));
  }
  int set = 0;
  // ABL == 5: diagnostic build, s_memtime stamps at the phase boundaries of wave 0 (sums over the block's tiles go to the start
  // of the output tensor of block 0: the build's outputs are garbage by design)
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  auto stamp = [&](int i) {
    if constexpr (ABL == 5) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      ph[i] += t - tprev;
      tprev = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (ABL == 5) tprev = __builtin_amdgcn_s_memtime();
  if constexpr (PREGEOM) comp_vo(tile_setup(DIST, 0));
  for (int it = 0; it < n_mine; ++it) {
    int nset = set + DIST; if (nset >= NBUF) nset -= NBUF;
    const TileDma nd = tile_setup(it + DIST, nset);
    if constexpr (PREGEOM) {
      issue_pre(nd);
    } else if constexpr (!SPREAD) {
#pragma unroll
      for (int k = 0; k < IPW; ++k) issue_piece(nd, k);
    }
    stamp(0);                              // loop top: tile set-up + DMA issue (unless SPREAD)
    const unsigned char* patch = smem + set * G::SETB;
    int pixv = pix0;
    asm volatile("" : "+v"(pixv));       // opaque per tile: the 3*(MT+2)*KC fragment addresses are formed where they are used, not hoisted into registers
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // output rows in groups of RG: RG + 2 fragment rows are live at a time
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int cc = 0; cc < KC; ++cc) {
#pragma unroll
        for (int g = 0; g < MT / RG; ++g) {
          if constexpr (SPREAD) {          // the next tile's pieces, spread over the NG MFMA groups of this tile
            constexpr int NG = 3 * KC * (MT / RG);
            const int gi = (kx * KC + cc) * (MT / RG) + g;
#pragma unroll
            for (int k = 0; k < IPW; ++k)
              if (k * NG / IPW == gi) issue_piece(nd, k);
          }
          if (ABL == 3 && (kx | cc)) continue;
          if constexpr (PREGEOM) {
            if (kx == 1 && cc == 0 && g == 0) comp_vo(tile_setup(it + 1 + DIST, 0));     // (only y0 / x0 / offsets / live of the set-up are used)
          }
          u32x4 a[RG + 2];
#pragma unroll
          for (int y = 0; y < RG + 2; ++y) {
            if constexpr (FRAGBASE) {
              const int D = (g * RG + y) * PW + kx;            // compile-time after unrolling
              a[y] = *reinterpret_cast<const u32x4*>(smem + ((fbase[D & 7] ^ (uint32_t)(cc * 64)) + (uint32_t)(D * (C * 2))));
            } else {
              const int pix = pixv + (g * RG + y) * PW + kx;
              a[y] = *reinterpret_cast<const u32x4*>(patch + pix * (C * 2) + (((cc * 4 + q) ^ cws_swz<C>(pix)) * 16));
            }
          }
#pragma unroll
          for (int y = 0; y < RG; ++y)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
              for (int j = 0; j < NT; ++j)
                acc[g * RG + y][j] = cws_mfma<T>(acc[g * RG + y][j], wf[j][ky * 3 + kx][cc], a[y + ky]);
        }
      }
    }

    stamp(1);                              // MFMA main loop
    // ---- epilogue: BN + SiLU (+ residual, in place) -> tile image [pixel][N] in the output type
    unsigned char* stg = RES ? smem + set * G::SETB + G::PATCH_PIECES * 1024 : smem + NBUF * G::SETB;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ch = (wn * NT + j) * 16 + q * 4;
#pragma unroll
      for (int y = 0; y < MT; ++y) {
        f32x4 v = acc[y][j] * sc[j] + sh[j];
        if (ABL != 1) { v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w); }
        const int opx = (wm * MT + y) * 16 + r;
        unsigned char* cell = stg + opx * (N * 2) + ((((ch >> 3) ^ (opx & (NCP - 1))) * 16) + ((ch >> 2) & 1) * 8);
        if (RES) {
          const u32x2 res = *reinterpret_cast<const u32x2*>(cell);
          v += f32x4{DT<T>::lo(res.x), DT<T>::hi(res.x), DT<T>::lo(res.y), DT<T>::hi(res.y)};
        }
        *reinterpret_cast<u32x2*>(cell) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
      }
    }
    stamp(2);                              // BN + SiLU + staging writes
    __syncthreads();
    stamp(3);                              // barrier
    {
      int b, y0, x0;
      tile_coords(t_first + it * p.bpx, b, y0, x0);
      const int b0 = GRP ? b * p.G : b, nim = GRP ? min(p.G, p.nimg - b0) : 1;
      const auto rsC = __builtin_amdgcn_make_buffer_rsrc(Cg + (int64_t)b0 * img_c, 0, (uint32_t)(img_c * 2 * nim), 0x00020000);
      int corr_c;
      const bool xok = col_ok(x0 + s_tx, p.delta_c, corr_c);
      const int off_c = (y0 * p.Wd + x0) * (int)p.ldc * 2 + s_rel + corr_c;
      u32x4 vv[NPASS];
#pragma unroll
      for (int k = 0; k < NPASS; ++k) vv[k] = *reinterpret_cast<const u32x4*>(stg + s_lds + k * PASS_LDS);
#pragma unroll
      for (int k = 0; k < NPASS; ++k) {
        const bool ok = ABL != 4 && xok && y0 + s_ty + k * PASS_ROWS < p.H;
        __builtin_amdgcn_raw_buffer_store_b128(vv[k], rsC, ok ? (uint32_t)(off_c + k * s_pass_rel) : OOB, 0, 0);
      }
    }
    // tile it+1 must have landed; the younger DMA pieces and the stores issued since stay in flight
    stamp(4);                              // store pass
    cws_wait_ring_tile<DIST, IPW, NPASS>(it);
    stamp(5);                              // wait for the next tile's pieces
    __syncthreads();
    stamp(6);                              // barrier
    if (++set == NBUF) set = 0;
    if constexpr (FRAGBASE) {
      const uint32_t step = set ? (uint32_t)G::SETB : (uint32_t)(-(NBUF - 1) * G::SETB);   // wave-uniform: the bases follow the buffer set
#pragma unroll
      for (int e = 0; e < 8; ++e) fbase[e] += step;
    }
  }
  if constexpr (ABL == 5) {
    if (blockIdx.x == 0 && tid == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.C);
      for (int i = 0; i < 7; ++i) dbg[i] = ph[i];
      dbg[7] = (unsigned long long)n_mine;
    }
  }
  cws_wait_vmcnt<0>();   // over-fetch pieces still target this block's LDS
}

static int cws_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined form of the kernel above.  Measured on the lock-step kernel (s_memtime stamps, MOY_CWS_ABL=5): per tile
// the matrix pipe is busy ~9 200 cycles, but every wave also spends ~2 000 cycles in the BN + SiLU epilogue, ~1 900 issuing the
// next tile's DMA pieces and ~900 in the store pass and barriers while the pipe idles -- 13 700 cycles per tile at C = 128.
// Here a wave's work is cut into half-tiles (RG output rows): the epilogue of half h-1 (VALU, LDS) and the DMA pieces of the
// next tile are interleaved, fragment by fragment, between the MFMA groups of half h, in ONE instruction stream (an MFMA
// holds the vector issue for 8 of its 16 cycles, the epilogue fits in the rest); the two accumulator halves alternate, so no
// extra registers are needed.  Staging is double buffered by tile parity (it lives in the buffer set, beside the patch), the
// store pass of tile t-1 sits between the two halves of tile t.  Per tile: barrier A (staging of t-1 complete / residual of
// t landed), barrier B (patch of t+1 landed, patch of t free).
template <typename T, int C, int N, int TH, int WN, bool RES, int OCC, int STAG = 0>
__global__ __launch_bounds__(512, 2 * OCC) void conv_ws_pipe_kernel(const ConvWsParams p) {
  constexpr int PW = 18, PH = TH + 2, NPIX = PH * PW, CPP = C / 8, NCP = N / 8, TPX = TH * 16;
  constexpr int PATCH_PIECES = (NPIX * CPP + 63) / 64, RES_PIECES = RES ? TPX * NCP / 64 : 0;
  constexpr int IPW0 = (PATCH_PIECES + 7) / 8, IPW1 = RES_PIECES / 8;
  constexpr int STGB = TPX * N * 2, SETB = PATCH_PIECES * 1024 + STGB;
  constexpr int NPASS = TPX * NCP / 512;
  constexpr int WM = 8 / WN, MT = TH / WM, NT = N / 16 / WN, KC = C / 32;
  constexpr int RG = MT / 2, NG = 3 * KC;                  // rows per half, MFMA groups per half
  constexpr uint32_t OOB = 0x80000000u;
  static_assert(MT % 2 == 0 && TH % WM == 0 && (N / 16) % WN == 0 && TPX * NCP % 512 == 0 && RES_PIECES % 8 == 0, "tile vs waves");
  static_assert(RG * NT <= NG && IPW0 <= NG, "the epilogue fragments / DMA pieces of a half must fit between its MFMA groups");
  static_assert((2 * SETB + 1024) * OCC <= 160 * 1024, "LDS budget");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int wn = wave % WN, wm = wave / WN;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t_first = xcd * p.per_xcd + slot;
  const int t_limit = min((xcd + 1) * p.per_xcd, p.ntiles);
  if (t_first >= t_limit) return;
  const int n_mine = (t_limit - t_first + p.bpx - 1) / p.bpx;

  const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  const uint32_t scratch = lds_base + 2 * SETB;
  const T* __restrict__ Ag = static_cast<const T*>(p.A);
  const T* __restrict__ Rg = static_cast<const T*>(p.R);
  T* __restrict__ Cg = static_cast<T*>(p.C);
  const int64_t img_a = (int64_t)p.H * p.Wd * p.lda, img_r = (int64_t)p.H * p.Wd * p.ldr, img_c = (int64_t)p.H * p.Wd * p.ldc;

  struct Tile { int b, y0, x0; bool live; };
  auto tile_of = [&](int it) {
    Tile t;
    t.live = it < n_mine;
    const int id = min(t_first + it * p.bpx, p.ntiles - 1);
    t.b = (int)fdiv((uint32_t)id, p.fd_timg);
    const int rem = id - t.b * p.tiles_img;
    const int ty = (int)fdiv((uint32_t)rem, p.fd_tx);
    t.y0 = ty * TH;
    t.x0 = (rem - ty * p.tiles_x) * 16;
    return t;
  };
  // piece k of the patch of tile t -> LDS set `set` (pieces beyond the patch image: zeros into the scratch KiB, keeps the counts uniform)
  auto patch_piece = [&](const Tile& t, int set, int k) {
    const int pc = wave + 8 * k;
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int X = pc * 64 + lv, pix = X / CPP, sl = X % CPP;
    const int py = pix / PW, px = pix - py * PW;
    const int yy = t.y0 + py - 1, xx = t.x0 + px - 1;
    const bool ok = t.live && pc < PATCH_PIECES && pix < NPIX && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.Wd;
    const uint32_t voff = ok ? (uint32_t)(((yy * p.Wd + xx) * (int)p.lda + ((sl ^ cws_swz<C>(pix)) * 8)) * 2) : OOB;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag + (int64_t)t.b * img_a), 0, (uint32_t)(img_a * 2), 0x00020000);
    cws_dma16(voff, rs, pc < PATCH_PIECES ? lds_base + set * SETB + pc * 1024 : scratch);
  };
  auto res_piece = [&](const Tile& t, int set, int k) {
    const int pc = wave + 8 * k;
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int X = pc * 64 + lv, opx = X / NCP, sl = X % NCP;
    const int yy = t.y0 + (opx >> 4), xx = t.x0 + (opx & 15);
    const bool ok = t.live && yy < p.H && xx < p.Wd;
    const uint32_t voff = ok ? (uint32_t)(((yy * p.Wd + xx) * (int)p.ldr + ((sl ^ (opx & (NCP - 1))) * 8)) * 2) : OOB;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((RES ? Rg : Ag) + (int64_t)t.b * (RES ? img_r : img_a)), 0,
                                                      (uint32_t)((RES ? img_r : img_a) * 2), 0x00020000);
    cws_dma16(voff, rs, lds_base + set * SETB + PATCH_PIECES * 1024 + pc * 1024);
  };

  // weights / BN of this wave -> registers
  u32x4 wf[NT][9][KC];
  f32x4 sc[NT], sh[NT];
  {
    const T* Wg = static_cast<const T*>(p.W);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = (wn * NT + j) * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cc = 0; cc < KC; ++cc)
          wf[j][tap][cc] = *reinterpret_cast<const u32x4*>(Wg + (int64_t)(n + r) * p.Kpad + tap * C + cc * 32 + q * 8);
      sc[j] = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n + q * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
      sh[j] = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const int s_px = tid / NCP, s_c = tid % NCP;
  const int s_ty = s_px >> 4, s_tx = s_px & 15;
  const int s_lds = s_px * (N * 2) + ((s_c ^ (s_px & (NCP - 1))) * 16);
  const int s_rel = ((s_ty * p.Wd + s_tx) * (int)p.ldc + s_c * 8) * 2;
  constexpr int PASS_ROWS = 512 / NCP / 16, PASS_LDS = (512 / NCP) * N * 2;
  const int s_pass_rel = PASS_ROWS * p.Wd * (int)p.ldc * 2;

  // one fragment (4 channels x 16 pixels per lane group) of a finished half -> staging image of its tile
  auto epi_frag = [&](const f32x4& a, int half, int y, int j, unsigned char* stg) {
    const int ch = (wn * NT + j) * 16 + q * 4;
    f32x4 v = a * sc[j] + sh[j];
    v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w);
    const int opx = (wm * MT + half * RG + y) * 16 + r;
    unsigned char* cell = stg + opx * (N * 2) + ((((ch >> 3) ^ (opx & (NCP - 1))) * 16) + ((ch >> 2) & 1) * 8);
    if (RES) {
      const u32x2 res = *reinterpret_cast<const u32x2*>(cell);
      v += f32x4{DT<T>::lo(res.x), DT<T>::hi(res.x), DT<T>::lo(res.y), DT<T>::hi(res.y)};
    }
    *reinterpret_cast<u32x2*>(cell) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
  };
  // the MFMA groups of one half; between(gi) runs before group gi (compile-time gi)
  const int pix0 = wm * MT * PW + r;
  auto half_mfma = [&](auto half_c, const unsigned char* patch, f32x4 (&acc)[RG][NT], auto&& between) {
    constexpr int half = decltype(half_c)::value;
    int pixv = pix0 + half * RG * PW;
    asm volatile("" : "+v"(pixv));
#pragma unroll
    for (int y = 0; y < RG; ++y)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    [&]<int... GI>(std::integer_sequence<int, GI...>) {
      ([&] {
        constexpr int kx = GI / KC, cc = GI % KC;
        between(std::integral_constant<int, GI>{});
        u32x4 a[RG + 2];
#pragma unroll
        for (int y = 0; y < RG + 2; ++y) {
          const int pix = pixv + y * PW + kx;
          a[y] = *reinterpret_cast<const u32x4*>(patch + pix * (C * 2) + (((cc * 4 + q) ^ cws_swz<C>(pix)) * 16));
        }
#pragma unroll
        for (int y = 0; y < RG; ++y)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[y][j] = cws_mfma<T>(acc[y][j], wf[j][ky * 3 + kx][cc], a[y + ky]);
      }(), ...);
    }(std::make_integer_sequence<int, NG>{});
  };
  auto store_pass = [&](const Tile& t, const unsigned char* stg, bool valid) {
    const auto rsC = __builtin_amdgcn_make_buffer_rsrc(Cg + (int64_t)t.b * img_c, 0, (uint32_t)(img_c * 2), 0x00020000);
    const int off_c = (t.y0 * p.Wd + t.x0) * (int)p.ldc * 2 + s_rel;
    const bool xok = valid && t.x0 + s_tx < p.Wd;
    u32x4 vv[NPASS];
#pragma unroll
    for (int k = 0; k < NPASS; ++k) vv[k] = *reinterpret_cast<const u32x4*>(stg + s_lds + k * PASS_LDS);
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
      const bool ok = xok && t.y0 + s_ty + k * PASS_ROWS < p.H;
      __builtin_amdgcn_raw_buffer_store_b128(vv[k], rsC, ok ? (uint32_t)(off_c + k * s_pass_rel) : OOB, 0, 0);
    }
  };

  // The two waves of a SIMD (w and w + 4) run the same stream: without an offset both sit in an epilogue block (VALU) at the same
  // time and the matrix pipe idles, then both want it.  Waves 4-7 start every barrier interval STAG x 64 cycles late, so one
  // wave's VALU block lies beside its partner's MFMA group (MI355X_MICROARCH.md, "two waves that run the same program: try a stagger").
  auto stagger = [&]() {
    if constexpr (STAG > 0) {
      if (wave >= 4) __builtin_amdgcn_s_sleep(STAG);
    }
  };
  // prologue: the first patch
  {
    const Tile t0 = tile_of(0);
#pragma unroll
    for (int k = 0; k < IPW0; ++k) patch_piece(t0, 0, k);
    cws_wait_vmcnt<0>();
    __syncthreads();
  }
  f32x4 accA[RG][NT], accB[RG][NT];
#pragma unroll
  for (int y = 0; y < RG; ++y)
#pragma unroll
    for (int j = 0; j < NT; ++j) accB[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  Tile prev = tile_of(0);
  for (int it = 0; it < n_mine; ++it) {
    const int set = it & 1, pset = set ^ 1;
    const Tile cur = tile_of(it), nxt = tile_of(it + 1);
    unsigned char* patch = smem + set * SETB;
    unsigned char* stg_cur = patch + PATCH_PIECES * 1024;
    unsigned char* stg_prev = smem + pset * SETB + PATCH_PIECES * 1024;
    if constexpr (RES) {
#pragma unroll
      for (int k = 0; k < IPW1; ++k) res_piece(cur, set, k);      // needed by the epilogue of this tile's first half (second half-time)
    }
    // ---- first half of tile `it`; between its groups: the epilogue of the previous tile's second half, the next tile's patch
    half_mfma(std::integral_constant<int, 0>{}, patch, accA, [&](auto gi_c) {
      constexpr int gi = decltype(gi_c)::value;
      if constexpr (gi < IPW0) patch_piece(nxt, pset, gi);
      if constexpr (gi < RG * NT) {
        if (it > 0) epi_frag(accB[gi / NT][gi % NT], 1, gi / NT, gi % NT, stg_prev);
      }
    });
    if constexpr (RES) cws_wait_vmcnt<IPW0>();                    // this tile's residual pieces (older than the IPW0 patch pieces)
    __syncthreads();                                               // A: staging of tile it-1 complete, residual of tile it visible
    stagger();
    store_pass(prev, stg_prev, it > 0);
    // ---- second half; between its groups: the epilogue of the first half
    half_mfma(std::integral_constant<int, 1>{}, patch, accB, [&](auto gi_c) {
      constexpr int gi = decltype(gi_c)::value;
      if constexpr (gi < RG * NT) epi_frag(accA[gi / NT][gi % NT], 0, gi / NT, gi % NT, stg_cur);
    });
    cws_wait_vmcnt<NPASS>();                                       // the next patch has landed; the stores stay in flight
    __syncthreads();                                               // B
    stagger();
    prev = cur;
  }
  // drain: the last tile's second half
  {
    unsigned char* stg_last = smem + ((n_mine - 1) & 1) * SETB + PATCH_PIECES * 1024;
#pragma unroll
    for (int f = 0; f < RG * NT; ++f) epi_frag(accB[f / NT][f % NT], 1, f / NT, f % NT, stg_last);
    __syncthreads();
    store_pass(prev, stg_last, true);
  }
  cws_wait_vmcnt<0>();
}

template <typename T, int C, int N, int TH, int WN, bool RES, int OCC = 1, int STAG = 0>
static int launch_conv_ws_pipe(ConvWsParams& p, int B, hipStream_t st) {
  if (plan_only(MOY_KERNEL_CONV_WS)) return MOY_OK;          // moy_gemm_query: the dispatch without the launch (and without touching the attribute statics)
  constexpr int PATCH_PIECES = ((TH + 2) * 18 * (C / 8) + 63) / 64;
  constexpr int LDS = 2 * (PATCH_PIECES * 1024 + TH * 16 * N * 2) + 1024;
  if constexpr (STAG == 0 && OCC == 1) {
    static const int stag = knob("MOY_CWS_STAG", 0);                  // MOY_CWS_STAG: A/B knob, delay of waves 4-7 after every barrier in units of 64 cycles
    if (stag == 2) return launch_conv_ws_pipe<T, C, N, TH, WN, RES, OCC, 2>(p, B, st);
    if (stag == 4) return launch_conv_ws_pipe<T, C, N, TH, WN, RES, OCC, 4>(p, B, st);
    if (stag == 8) return launch_conv_ws_pipe<T, C, N, TH, WN, RES, OCC, 8>(p, B, st);
  }
  auto kern = conv_ws_pipe_kernel<T, C, N, TH, WN, RES, OCC, STAG>;
  static bool attr_set = false;
  if (!attr_set) {
    if (LDS > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  p.tiles_x = (p.Wd + 15) / 16;
  const int tiles_y = (p.H + TH - 1) / TH;
  p.tiles_img = p.tiles_x * tiles_y;
  p.ntiles = B * p.tiles_img;
  p.fd_timg = make_fastdiv((uint32_t)p.tiles_img);
  p.fd_tx = make_fastdiv((uint32_t)p.tiles_x);
  p.per_xcd = (p.ntiles + 7) / 8;
  p.bpx = cu_limit(cws_num_cus()) / 8 * OCC;
  if (p.bpx < 1) p.bpx = 1;
  if (p.bpx > p.per_xcd) p.bpx = p.per_xcd;
  hipLaunchKernelGGL(kern, dim3(8 * p.bpx), dim3(512), LDS, st, p);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------
// PING-PONG form (round 3).  What the stamps of the lock-step kernel show (per C = 128 tile and wave: matrix pipe 9 200 cycles,
// BN + SiLU 2 000, DMA issue 1 900, stores + barriers 900 -- 13 700 in all) is two waves per SIMD running the SAME program in
// lock step: both sit in their MFMA loop together (each MFMA takes 32 cycles instead of 16: they share the pipe), then both
// sit in the epilogue / DMA issue / store pass together while the matrix pipe idles.  The software-pipelined form above could not
// undo that (its two waves still interleave at instruction granularity and the block's barriers re-align them every tile).
// Here the two waves of a SIMD -- wave w (group A: waves 0-3) and wave w + 4 (group B) -- are put in ANTI-PHASE by construction.
// A tile is cut into four barrier intervals; in every interval one group runs an MFMA half (RG output rows of its tile share,
// alone on the pipe: 16 cycles per MFMA) while the other does the non-matrix work in the vector slots the MFMAs leave free:
//
//     interval   group A (waves 0-3)                                    group B (waves 4-7)
//        0       MFMA half 1 of tile t                                  epilogue half 2 of tile t-1; DMA: patch pieces of t+1
//        1       epilogue half 1 of t; DMA: patch pieces of t+1;        MFMA half 1 of tile t
//                stores of the half-2 rows of tile t-1
//        2       MFMA half 2 of tile t                                  epilogue half 1 of t; DMA: residual pieces of t+1
//        3       epilogue half 2 of t; stores of the half-1 rows of t   MFMA half 2 of tile t
//
// A wave holds ONE accumulator half at a time (RG x NT fragments instead of MT x NT).  Two buffer sets (patch + staging /
// residual image) alternate by tile parity.  Hazards: the patch of t+1 goes where the patch of t-1 lay (last read in interval 3
// of t-1); the residual of t+1 goes into the staging image of t-1, whose last rows leave in interval 1 of t -- hence in interval 2.
// Group A's vector-memory stream per tile is [KA patch pieces, NPS stores, NPS stores] and it retires the pieces with a counted
// vmcnt(2 NPS) that leaves the stores in flight; group B's is [KB patch pieces, KR residual pieces], retired by vmcnt(0) at the
// end of interval 3 (issued one or three intervals earlier).  Results are bit-identical to conv_ws_kernel (same taps in the same
// order into the same accumulators; tests/test_gpu_ops.py).
template <typename T, int C, int N, int TH, int WN, bool RES, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv_ws_pp_kernel(const ConvWsParams p) {
  constexpr int PW = 18, PH = TH + 2, NPIX = PH * PW, CPP = C / 8, NCP = N / 8, TPX = TH * 16;
  constexpr int PATCH_PIECES = (NPIX * CPP + 63) / 64, RES_PIECES = RES ? TPX * NCP / 64 : 0;
  constexpr int STGB = TPX * N * 2, SETB = PATCH_PIECES * 1024 + STGB;
  constexpr int WM = 8 / WN, MT = TH / WM, NT = N / 16 / WN, KC = C / 32;
  constexpr int RG = MT / 2;                               // rows per half
  constexpr int KA = (PATCH_PIECES + 7) / 8;               // patch pieces per wave of group A (interval 1): pieces [0, 4 KA)
  constexpr int KB = (PATCH_PIECES - 4 * KA + 3) / 4;      // ... of group B (interval 0): pieces [4 KA, 4 KA + 4 KB)
  constexpr int KR = RES_PIECES / 4;                       // residual pieces per wave of group B (interval 2)
  constexpr int SETPX = WM * RG * 16;                      // pixels of a store set (the half-h rows of every wave)
  constexpr int NPS = SETPX * NCP / 256;                   // 16-byte stores per thread of group A and store set
  constexpr uint32_t OOB = 0x80000000u;
  static_assert(MT % 2 == 0 && TH % WM == 0 && (N / 16) % WN == 0 && RES_PIECES % 4 == 0 && (SETPX * NCP) % 256 == 0, "tile vs waves");
  static_assert(4 * KA + 4 * KB >= PATCH_PIECES && KB >= 0, "piece split");
  static_assert(2 * SETB + 1024 <= 160 * 1024, "LDS budget");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int wn = wave % WN, wm = wave / WN;
  const bool grpA = wave < 4;
  const bool PRIO = p.prio != 0;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t_first = xcd * p.per_xcd + slot;
  const int t_limit = min((xcd + 1) * p.per_xcd, p.ntiles);
  if (t_first >= t_limit) return;
  const int n_mine = (t_limit - t_first + p.bpx - 1) / p.bpx;

  const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  const uint32_t scratch = lds_base + 2 * SETB;
  const T* __restrict__ Ag = static_cast<const T*>(p.A);
  const T* __restrict__ Rg = static_cast<const T*>(p.R);
  T* __restrict__ Cg = static_cast<T*>(p.C);
  const int64_t img_a = (int64_t)p.H * p.Wd * p.lda, img_r = (int64_t)p.H * p.Wd * p.ldr, img_c = (int64_t)p.H * p.Wd * p.ldc;

  struct Tile { int b, y0, x0; bool live; };
  auto tile_of = [&](int it) {
    Tile t;
    t.live = it < n_mine;
    const int id = min(t_first + it * p.bpx, p.ntiles - 1);
    t.b = (int)fdiv((uint32_t)id, p.fd_timg);
    const int rem = id - t.b * p.tiles_img;
    const int ty = (int)fdiv((uint32_t)rem, p.fd_tx);
    t.y0 = ty * TH;
    t.x0 = (rem - ty * p.tiles_x) * 16;
    return t;
  };
  // patch piece pc of tile t -> LDS set `set` (pieces beyond the patch image: zeros into the scratch KiB, keeps the counts uniform)
  auto patch_piece = [&](const Tile& t, int set, int pc) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int X = pc * 64 + lv, pix = X / CPP, sl = X % CPP;
    const int py = pix / PW, px = pix - py * PW;
    const int yy = t.y0 + py - 1, xx = t.x0 + px - 1;
    const bool ok = t.live && pc < PATCH_PIECES && pix < NPIX && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.Wd;
    const uint32_t voff = ok ? (uint32_t)(((yy * p.Wd + xx) * (int)p.lda + ((sl ^ cws_swz<C>(pix)) * 8)) * 2) : OOB;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag + (int64_t)t.b * img_a), 0, (uint32_t)(img_a * 2), 0x00020000);
    if (ABL == 2) return;
    cws_dma16(voff, rs, pc < PATCH_PIECES ? lds_base + set * SETB + pc * 1024 : scratch);
  };
  auto res_piece = [&](const Tile& t, int set, int pc) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int X = pc * 64 + lv, opx = X / NCP, sl = X % NCP;
    const int yy = t.y0 + (opx >> 4), xx = t.x0 + (opx & 15);
    const bool ok = t.live && yy < p.H && xx < p.Wd;
    const uint32_t voff = ok ? (uint32_t)(((yy * p.Wd + xx) * (int)p.ldr + ((sl ^ (opx & (NCP - 1))) * 8)) * 2) : OOB;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((RES ? Rg : Ag) + (int64_t)t.b * (RES ? img_r : img_a)), 0,
                                                      (uint32_t)((RES ? img_r : img_a) * 2), 0x00020000);
    if (ABL == 2) return;
    cws_dma16(voff, rs, lds_base + set * SETB + PATCH_PIECES * 1024 + pc * 1024);
  };

  // weights / BN of this wave -> registers
  u32x4 wf[NT][9][KC];
  f32x4 sc[NT], sh[NT];
  {
    const T* Wg = static_cast<const T*>(p.W);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = (wn * NT + j) * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cc = 0; cc < KC; ++cc)
          wf[j][tap][cc] = *reinterpret_cast<const u32x4*>(Wg + (int64_t)(n + r) * p.Kpad + tap * C + cc * 32 + q * 8);
      sc[j] = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n + q * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
      sh[j] = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // epilogue of one half: BN + SiLU (+ residual, in place) of RG x NT fragments -> staging image [pixel][N] of the tile
  auto epi_half = [&](const f32x4 (&acc)[RG][NT], int half, unsigned char* stg) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ch = (wn * NT + j) * 16 + q * 4;
#pragma unroll
      for (int y = 0; y < RG; ++y) {
        f32x4 v = acc[y][j] * sc[j] + sh[j];
        if (ABL != 1) { v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w); }
        const int opx = (wm * MT + half * RG + y) * 16 + r;
        unsigned char* cell = stg + opx * (N * 2) + ((((ch >> 3) ^ (opx & (NCP - 1))) * 16) + ((ch >> 2) & 1) * 8);
        if (RES) {
          const u32x2 res = *reinterpret_cast<const u32x2*>(cell);
          v += f32x4{DT<T>::lo(res.x), DT<T>::hi(res.x), DT<T>::lo(res.y), DT<T>::hi(res.y)};
        }
        *reinterpret_cast<u32x2*>(cell) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
      }
    }
  };
  // The MFMA groups of one half (same tap order as conv_ws_kernel: kx, cc outer; ky inner).
  // Fragment addresses without per-read arithmetic: the pixel of a read is P + c (P = this lane's first pixel, c = a compile-time
  // constant), its swizzle term depends on (P + c) & 7 = ((P & 7) + (c & 7)) & 7 only, and the k-block cc enters as an XOR of
  // address bits that nothing else touches.  So 8 per-lane registers A[j] (j = c & 7), formed once per half, give every address
  // as (A[c & 7] ^ (cc << 6)) + an immediate offset: at most one VALU operation per ds_read_b128 (the lock-step kernel's code
  // spends 2-3 plus two hoisted registers per fragment row).  The reads of group g + 1 are issued before the MFMAs of group g.
  const int pix0 = wm * MT * PW + r;
  auto half_mfma = [&](auto half_c, uint32_t patch_off, f32x4 (&acc)[RG][NT]) {
    constexpr int half = decltype(half_c)::value;
    int pv = pix0;
    asm volatile("" : "+v"(pv));
    uint32_t A[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) A[j] = patch_off + (uint32_t)pv * (C * 2) + (uint32_t)((q ^ cws_swz<C>((pv & 7) + j)) * 16);
#pragma unroll
    for (int y = 0; y < RG; ++y)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto rd = [&](auto g_c, u32x4 (&buf)[RG + 2]) {
      constexpr int g = decltype(g_c)::value, kx = g / KC, cc = g % KC;
#pragma unroll
      for (int y = 0; y < RG + 2; ++y) {
        const int c = half * RG * PW + y * PW + kx;         // compile-time after unrolling
        buf[y] = *reinterpret_cast<const u32x4*>(smem + ((A[c & 7] ^ (uint32_t)(cc * 64)) + (uint32_t)(c * (C * 2))));
      }
    };
    u32x4 a[2][RG + 2];
    rd(std::integral_constant<int, 0>{}, a[0]);
    // The partner wave of this SIMD is in its vector phase: with equal priority the arbiter hands it the issue port by age and the
    // MFMAs of this wave go out at HALF rate (stamps: 4 400 cycles for the 144 MFMAs of a half).  Raised priority for the matrix
    // phase: an MFMA issues whenever the pipe is free, the partner's vector work fills the 8 cycles in 16 that an MFMA leaves.
    if (PRIO) __builtin_amdgcn_s_setprio(2);
    [&]<int... GI>(std::integer_sequence<int, GI...>) {
      ([&] {
        constexpr int kx = GI / KC, cc = GI % KC;
        if constexpr (GI + 1 < 3 * KC) rd(std::integral_constant<int, GI + 1>{}, a[(GI + 1) & 1]);
        // (pinning this order with sched_barriers and keeping A[] opaque was measured: 35 spilled VGPRs at C = 128, 302 us instead of 237)
        if (ABL == 3 && GI > 0) return;
        // ky OUTSIDE the rows: consecutive MFMAs go to DIFFERENT accumulators.  A lone wave that issues three dependent MFMAs in
        // a row (rows outside, as the lock-step kernel orders them -- there the partner's MFMAs fall in between) gets one MFMA per
        // 32 cycles instead of 16 (stamps: 4 300 cycles for the 144 MFMAs of a half).  Every accumulator still sees its taps in the
        // same order: results unchanged.
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int y = 0; y < RG; ++y)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[y][j] = cws_mfma<T>(acc[y][j], wf[j][ky * 3 + kx][cc], a[GI & 1][y + ky]);
      }(), ...);
    }(std::make_integer_sequence<int, 3 * KC>{});
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  // group A's store pass over store set h (the half-h rows of every wave): thread ta = tid (0..255), chunk X = k * 256 + ta
  auto store_set = [&](const Tile& t, const unsigned char* stg, int h, bool valid) {
    const auto rsC = __builtin_amdgcn_make_buffer_rsrc(Cg + (int64_t)t.b * img_c, 0, (uint32_t)(img_c * 2), 0x00020000);
    u32x4 vv[NPS];
    auto geom = [&](int k, int& opx, int& row, int& x, int& c) {
      const int X = k * 256 + tid, ps = X / NCP;
      c = X % NCP;
      const int rs_ = ps >> 4;
      x = ps & 15;
      row = (rs_ / RG) * MT + h * RG + (rs_ % RG);
      opx = row * 16 + x;
    };
#pragma unroll
    for (int k = 0; k < NPS; ++k) {
      int opx, row, x, c;
      geom(k, opx, row, x, c);
      vv[k] = *reinterpret_cast<const u32x4*>(stg + opx * (N * 2) + ((c ^ (opx & (NCP - 1))) * 16));
    }
#pragma unroll
    for (int k = 0; k < NPS; ++k) {
      int opx, row, x, c;
      geom(k, opx, row, x, c);
      const bool ok = valid && ABL != 4 && t.y0 + row < p.H && t.x0 + x < p.Wd;
      const int goff = (((t.y0 + row) * p.Wd + t.x0 + x) * (int)p.ldc + c * 8) * 2;
      __builtin_amdgcn_raw_buffer_store_b128(vv[k], rsC, ok ? (uint32_t)goff : OOB, 0, 0);
    }
  };

  // prologue: patch and residual of the first tile, by all eight waves
  {
    const Tile t0 = tile_of(0);
#pragma unroll
    for (int k = 0; k < (PATCH_PIECES + 7) / 8; ++k) patch_piece(t0, 0, wave + 8 * k);
    if constexpr (RES) {
#pragma unroll
      for (int k = 0; k < RES_PIECES / 8; ++k) res_piece(t0, 0, wave + 8 * k);
    }
    cws_wait_vmcnt<0>();
    __syncthreads();
  }

  f32x4 acc[RG][NT];
#pragma unroll
  for (int y = 0; y < RG; ++y)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[y][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  Tile prev = tile_of(0);
  // ABL == 5: diagnostic build, s_memtime stamps of wave 0 (group A) and wave 4 (group B): cycles of WORK in each of the four
  // intervals (up to its barrier) and cycles spent AT the four barriers; sums over the block's tiles go to the start of the output
  // tensor of block 0 (the build's outputs are garbage by design)
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  auto stamp = [&](int i) {
    if constexpr (ABL == 5) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      ph[i] += t - tprev;
      tprev = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (ABL == 5) tprev = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n_mine; ++it) {
    const int set = it & 1, nset = set ^ 1;
    const Tile cur = tile_of(it), nxt = tile_of(it + 1);
    const uint32_t patch_off = (uint32_t)(set * SETB);
    unsigned char* stg_cur = smem + set * SETB + PATCH_PIECES * 1024;
    unsigned char* stg_prev = smem + nset * SETB + PATCH_PIECES * 1024;
    // ---- interval 0
    if (grpA) {
      half_mfma(std::integral_constant<int, 0>{}, patch_off, acc);
    } else {
      if (it > 0) epi_half(acc, 1, stg_prev);
#pragma unroll
      for (int k = 0; k < KB; ++k) patch_piece(nxt, nset, 4 * KA + (wave - 4) + 4 * k);
    }
    stamp(0);
    __syncthreads();
    stamp(4);
    // ---- interval 1
    if (grpA) {
      epi_half(acc, 0, stg_cur);
#pragma unroll
      for (int k = 0; k < KA; ++k) patch_piece(nxt, nset, wave + 4 * k);
      store_set(prev, stg_prev, 1, it > 0);
    } else {
      half_mfma(std::integral_constant<int, 0>{}, patch_off, acc);
    }
    stamp(1);
    __syncthreads();
    stamp(5);
    // ---- interval 2
    if (grpA) {
      half_mfma(std::integral_constant<int, 1>{}, patch_off, acc);
    } else {
      epi_half(acc, 0, stg_cur);
      if constexpr (RES) {
#pragma unroll
        for (int k = 0; k < KR; ++k) res_piece(nxt, nset, (wave - 4) + 4 * k);
      }
    }
    stamp(2);
    __syncthreads();
    stamp(6);
    // ---- interval 3
    if (grpA) {
      epi_half(acc, 1, stg_cur);
      store_set(cur, stg_cur, 0, true);
      cws_wait_vmcnt<2 * NPS>();                                   // this wave's patch pieces of tile it+1; the stores stay in flight
    } else {
      half_mfma(std::integral_constant<int, 1>{}, patch_off, acc);
      cws_wait_vmcnt<0>();                                         // this wave's patch + residual pieces of tile it+1
    }
    stamp(3);
    __syncthreads();
    stamp(7);
    prev = cur;
  }
  if constexpr (ABL == 5) {
    if (blockIdx.x == 0 && (tid == 0 || tid == 256)) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.C) + (tid ? 16 : 0);
      for (int i = 0; i < 8; ++i) dbg[i] = ph[i];
      dbg[8] = (unsigned long long)n_mine;
    }
  }
  // drain: group B's second half of the last tile, then its rows
  {
    unsigned char* stg_last = smem + ((n_mine - 1) & 1) * SETB + PATCH_PIECES * 1024;
    if (!grpA) epi_half(acc, 1, stg_last);
    __syncthreads();
    if (grpA) store_set(prev, stg_last, 1, true);
  }
  cws_wait_vmcnt<0>();
}

template <typename T, int C, int N, int TH, int WN, bool RES, int ABL = 0>
static int launch_conv_ws_pp(ConvWsParams& p, int B, hipStream_t st) {
  if (plan_only(MOY_KERNEL_CONV_WS)) return MOY_OK;          // moy_gemm_query: the dispatch without the launch (and without touching the attribute statics)
  constexpr int PATCH_PIECES = ((TH + 2) * 18 * (C / 8) + 63) / 64;
  constexpr int LDS = 2 * (PATCH_PIECES * 1024 + TH * 16 * N * 2) + 1024;
#if MOY_DIAG
  if constexpr (ABL == 0 && std::is_same<T, bf16_t>::value && !RES) {
    static const int abl = garbage_mode_env("MOY_CWS_ABL");                   // MOY_CWS_ABL=1..4: timing-only builds (no SiLU / no DMA / one MFMA group / no stores)
    if (abl == 1) return launch_conv_ws_pp<T, C, N, TH, WN, RES, 1>(p, B, st);
    if (abl == 2) return launch_conv_ws_pp<T, C, N, TH, WN, RES, 2>(p, B, st);
    if (abl == 3) return launch_conv_ws_pp<T, C, N, TH, WN, RES, 3>(p, B, st);
    if (abl == 4) return launch_conv_ws_pp<T, C, N, TH, WN, RES, 4>(p, B, st);
    if (abl == 5) return launch_conv_ws_pp<T, C, N, TH, WN, RES, 5>(p, B, st);
  }
#endif
  auto kern = conv_ws_pp_kernel<T, C, N, TH, WN, RES, ABL>;
  {
    static const int prio = knob("MOY_CWS_PRIO", 1);
    p.prio = prio;
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (LDS > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  p.tiles_x = (p.Wd + 15) / 16;
  const int tiles_y = (p.H + TH - 1) / TH;
  p.tiles_img = p.tiles_x * tiles_y;
  p.ntiles = B * p.tiles_img;
  p.fd_timg = make_fastdiv((uint32_t)p.tiles_img);
  p.fd_tx = make_fastdiv((uint32_t)p.tiles_x);
  p.per_xcd = (p.ntiles + 7) / 8;
  p.bpx = cu_limit(cws_num_cus()) / 8;
  if (p.bpx < 1) p.bpx = 1;
  if (p.bpx > p.per_xcd) p.bpx = p.per_xcd;
  hipLaunchKernelGGL(kern, dim3(8 * p.bpx), dim3(512), LDS, st, p);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------
// Stride-2 member of the family (the down-sampling convs of the backbone / neck, yolo_track.yaml:18-24,35,41: Cin -> Cout, 3x3,
// stride 2, pad 1).  Same structure as conv_ws_kernel; what changes is the patch:
//   * an output tile of TH x 16 pixels needs (2 TH + 1) x 33 input pixels; a patch row is stored PARITY-DE-INTERLEAVED --
//     the 17 even columns first, then the 16 odd ones -- so that the 16 input pixels a fragment needs for tap column kx
//     (columns 2 ox + kx) are 16 CONSECUTIVE patch pixels (even plane from 0, odd plane from 0, even plane from 1) and the
//     conflict-free XOR maps of the stride-1 kernels apply unchanged (the implicit-GEMM path it replaces spent 12 VALU
//     operations per MFMA on addressing and 2 bank-conflict cycles per LDS instruction, profiles/r01_e_pmc_gemm_conv_wreg.txt);
//   * input row 2 y + 2 serves tap row 2 of output row y and tap row 0 of output row y + 1: 2 RG + 1 fragment reads per 3 RG MFMAs.
struct ConvS2Params {
  const void* A; int64_t lda;
  const void* W; int Kpad;
  const float* scale; const float* shift;
  void* C; int64_t ldc;
  int Hin, Win, Hout, Wout;
  int tiles_x, tiles_img, ntiles, per_xcd, bpx;
  FastDiv fd_timg, fd_tx;
  // POST forms (round 4, moy_gemm_args.post_*): the conv's only consumer is a 1x1 conv (C2f.cv1 behind a down-sampling Conv,
  // yolo_track.yaml:19-20 over block.py:225-235), applied to the finished tile while it sits in LDS: C = SiLU(BN2(tile . W2^T))
  const void* W2; int Kpad2;
  const float* scale2; const float* shift2;
};

//   POST: the tile, rounded to T exactly where the unfused plan stored it, is the B operand of a second product with the N x N
//   weights of the consumer (16 output columns per wave in registers, k ascending in 32-wide panels as every other kernel of
//   the library sums it): BN + SiLU again into a SECOND staging tile, which the unchanged store code writes out.  The conv's own
//   output never reaches HBM.
template <typename T, int C, int N, int TH, int WN, int NBUF, bool POST = false>
__global__ __launch_bounds__(512, 2) void conv_s2_kernel(const ConvS2Params p) {
  constexpr int PW = 33, PH = 2 * TH + 1, NPIX = PH * PW, CPP = C / 8, NCP = N / 8, TPX = TH * 16;
  constexpr int PIECES = (NPIX * CPP + 63) / 64, IPW = (PIECES + 7) / 8;
  constexpr int SETB = PIECES * 1024, STGB = TPX * N * 2;
  constexpr int NPASS = TPX * NCP / 512;
  constexpr int WM = 8 / WN, MT = TH / WM, NT = N / 16 / WN, KC = C / 32;
  constexpr int DIST = NBUF - 1;
  constexpr int RG = MT < 4 ? MT : 4;
  constexpr uint32_t OOB = 0x80000000u;
  static_assert(TH % WM == 0 && (N / 16) % WN == 0 && TPX * NCP % 512 == 0 && MT % RG == 0, "tile vs waves");
  static_assert(NBUF * SETB + (POST ? 2 : 1) * STGB + 1024 <= 160 * 1024, "LDS budget");
  static_assert(!POST || (WM == 1 && N % 32 == 0), "POST: every wave sees all rows of the tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int wn = wave % WN, wm = wave / WN;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t_first = xcd * p.per_xcd + slot;
  const int t_limit = min((xcd + 1) * p.per_xcd, p.ntiles);
  if (t_first >= t_limit) return;
  const int n_mine = (t_limit - t_first + p.bpx - 1) / p.bpx;

  const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  const uint32_t scratch = lds_base + NBUF * SETB + (POST ? 2 : 1) * STGB;
  const T* __restrict__ Ag = static_cast<const T*>(p.A);
  T* __restrict__ Cg = static_cast<T*>(p.C);
  const int64_t img_a = (int64_t)p.Hin * p.Win * p.lda, img_c = (int64_t)p.Hout * p.Wout * p.ldc;

  struct Tile { int b, y0, x0; bool live; };
  auto tile_of = [&](int it) {
    Tile t;
    t.live = it < n_mine;
    const int id = min(t_first + it * p.bpx, p.ntiles - 1);
    t.b = (int)fdiv((uint32_t)id, p.fd_timg);
    const int rem = id - t.b * p.tiles_img;
    const int ty = (int)fdiv((uint32_t)rem, p.fd_tx);
    t.y0 = ty * TH;                        // output coordinates
    t.x0 = (rem - ty * p.tiles_x) * 16;
    return t;
  };
  auto issue_tile = [&](int it, int set) {
    const Tile t = tile_of(it);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Ag + (int64_t)t.b * img_a), 0, (uint32_t)(img_a * 2), 0x00020000);
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
      const int pc = wave + 8 * k;
      int lv = lane;
      if constexpr (!MOY_CWS_HOIST) asm volatile("" : "+v"(lv));   // (round 3: the geometry may live in loop-invariant registers: 86 / 144 VGPRs were in use)
      const int X = pc * 64 + lv, pix = X / CPP, sl = X % CPP;
      const int row = pix / PW, s33 = pix - row * PW;
      const int col = s33 < 17 ? 2 * s33 : 2 * (s33 - 17) + 1;      // patch column of this de-interleaved slot
      const int yy = 2 * t.y0 - 1 + row, xx = 2 * t.x0 - 1 + col;
      const bool ok = t.live && pc < PIECES && pix < NPIX && (unsigned)yy < (unsigned)p.Hin && (unsigned)xx < (unsigned)p.Win;
      const uint32_t voff = ok ? (uint32_t)(((yy * p.Win + xx) * (int)p.lda + ((sl ^ cws_swz<C>(pix)) * 8)) * 2) : OOB;
      cws_dma16(voff, rs, pc < PIECES ? lds_base + set * SETB + pc * 1024 : scratch);
    }
  };

  u32x4 wf[NT][9][KC];
  f32x4 sc[NT], sh[NT];
  {
    const T* Wg = static_cast<const T*>(p.W);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = (wn * NT + j) * 16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cc = 0; cc < KC; ++cc)
          wf[j][tap][cc] = *reinterpret_cast<const u32x4*>(Wg + (int64_t)(n + r) * p.Kpad + tap * C + cc * 32 + q * 8);
      sc[j] = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n + q * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
      sh[j] = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  constexpr int KP2 = N / 32;
  [[maybe_unused]] u32x4 w2f[POST ? NT : 1][POST ? KP2 : 1];
  [[maybe_unused]] f32x4 sc2[POST ? NT : 1], sh2[POST ? NT : 1];
  if constexpr (POST) {
    const T* W2g = static_cast<const T*>(p.W2);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = (wn * NT + j) * 16;
#pragma unroll
      for (int cc = 0; cc < KP2; ++cc) w2f[j][cc] = *reinterpret_cast<const u32x4*>(W2g + (int64_t)(n + r) * p.Kpad2 + cc * 32 + q * 8);
      sc2[j] = p.scale2 ? *reinterpret_cast<const f32x4*>(p.scale2 + n + q * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
      sh2[j] = p.shift2 ? *reinterpret_cast<const f32x4*>(p.shift2 + n + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const int s_px = tid / NCP, s_c = tid % NCP;
  const int s_ty = s_px >> 4, s_tx = s_px & 15;
  const int s_lds = s_px * (N * 2) + ((s_c ^ (s_px & (NCP - 1))) * 16);
  const int s_rel = ((s_ty * p.Wout + s_tx) * (int)p.ldc + s_c * 8) * 2;
  constexpr int PASS_ROWS = 512 / NCP / 16, PASS_LDS = (512 / NCP) * N * 2;
  const int s_pass_rel = PASS_ROWS * p.Wout * (int)p.ldc * 2;

#pragma unroll
  for (int d = 0; d < DIST; ++d) issue_tile(d, d);
  cws_wait_vmcnt<(DIST - 1) * IPW>();
  __syncthreads();

  const int pix0 = 2 * wm * MT * PW + r;   // patch pixel of this lane: input row of output row 0 of the wave, even plane, tap column 0
  int set = 0;
  for (int it = 0; it < n_mine; ++it) {
    {
      int nset = set + DIST; if (nset >= NBUF) nset -= NBUF;
      issue_tile(it + DIST, nset);
    }
    const unsigned char* patch = smem + set * SETB;
    int pixv = pix0;
    asm volatile("" : "+v"(pixv));
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      constexpr int KOFF[3] = {0, 17, 1};                 // even plane, odd plane, even plane shifted by one
#pragma unroll
      for (int cc = 0; cc < KC; ++cc) {
#pragma unroll
        for (int g = 0; g < MT / RG; ++g) {
          u32x4 a[2 * RG + 1];
#pragma unroll
          for (int y = 0; y < 2 * RG + 1; ++y) {
            const int pix = pixv + (2 * g * RG + y) * PW + KOFF[kx];
            a[y] = *reinterpret_cast<const u32x4*>(patch + pix * (C * 2) + (((cc * 4 + q) ^ cws_swz<C>(pix)) * 16));
          }
#pragma unroll
          for (int y = 0; y < RG; ++y)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
              for (int j = 0; j < NT; ++j)
                acc[g * RG + y][j] = cws_mfma<T>(acc[g * RG + y][j], wf[j][ky * 3 + kx][cc], a[2 * y + ky]);
        }
      }
    }
    unsigned char* stg = smem + NBUF * SETB;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ch = (wn * NT + j) * 16 + q * 4;
#pragma unroll
      for (int y = 0; y < MT; ++y) {
        f32x4 v = acc[y][j] * sc[j] + sh[j];
        v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w);
        const int opx = (wm * MT + y) * 16 + r;
        unsigned char* cell = stg + opx * (N * 2) + ((((ch >> 3) ^ (opx & (NCP - 1))) * 16) + ((ch >> 2) & 1) * 8);
        *reinterpret_cast<u32x2*>(cell) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
      }
    }
    __syncthreads();
    const unsigned char* stg_out = stg;
    if constexpr (POST) {
      // second product on the finished tile: rows = its TPX pixels, k = its N channels (chunk (cc*4 + q) ^ (pixel & 15) of the row)
      unsigned char* stg2 = stg + STGB;
      f32x4 acc2[MT][NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < KP2; ++cc) {
        u32x4 a2[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int opx = i * 16 + r;
          a2[i] = *reinterpret_cast<const u32x4*>(stg + opx * (N * 2) + (((cc * 4 + q) ^ (opx & (NCP - 1))) * 16));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc2[i][j] = cws_mfma<T>(acc2[i][j], w2f[j][cc], a2[i]);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int ch = (wn * NT + j) * 16 + q * 4;
#pragma unroll
        for (int y = 0; y < MT; ++y) {
          f32x4 v = acc2[y][j] * sc2[j] + sh2[j];
          v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w);
          const int opx = y * 16 + r;
          unsigned char* cell = stg2 + opx * (N * 2) + ((((ch >> 3) ^ (opx & (NCP - 1))) * 16) + ((ch >> 2) & 1) * 8);
          *reinterpret_cast<u32x2*>(cell) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
        }
      }
      __syncthreads();
      stg_out = stg2;
    }
    {
      const Tile t = tile_of(it);
      const auto rsC = __builtin_amdgcn_make_buffer_rsrc(Cg + (int64_t)t.b * img_c, 0, (uint32_t)(img_c * 2), 0x00020000);
      const int off_c = (t.y0 * p.Wout + t.x0) * (int)p.ldc * 2 + s_rel;
      const bool xok = t.x0 + s_tx < p.Wout;
      u32x4 vv[NPASS];
#pragma unroll
      for (int k = 0; k < NPASS; ++k) vv[k] = *reinterpret_cast<const u32x4*>(stg_out + s_lds + k * PASS_LDS);
#pragma unroll
      for (int k = 0; k < NPASS; ++k) {
        const bool ok = xok && t.y0 + s_ty + k * PASS_ROWS < p.Hout;
        __builtin_amdgcn_raw_buffer_store_b128(vv[k], rsC, ok ? (uint32_t)(off_c + k * s_pass_rel) : OOB, 0, 0);
      }
    }
    cws_wait_ring_tile<DIST, IPW, NPASS>(it);
    __syncthreads();
    if (++set == NBUF) set = 0;
  }
  cws_wait_vmcnt<0>();
}

template <typename T, int C, int N, int TH, int WN, int NBUF, bool POST = false>
static int launch_conv_s2(ConvS2Params& p, int B, hipStream_t st) {
  if (plan_only(MOY_KERNEL_CONV_S2)) return MOY_OK;          // moy_gemm_query: the dispatch without the launch (and without touching the attribute statics)
  constexpr int PIECES = ((2 * TH + 1) * 33 * (C / 8) + 63) / 64;
  constexpr int LDS = NBUF * PIECES * 1024 + (POST ? 2 : 1) * TH * 16 * N * 2 + 1024;
  auto kern = conv_s2_kernel<T, C, N, TH, WN, NBUF, POST>;
  static bool attr_set = false;
  if (!attr_set) {
    if (LDS > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  p.tiles_x = (p.Wout + 15) / 16;
  const int tiles_y = (p.Hout + TH - 1) / TH;
  p.tiles_img = p.tiles_x * tiles_y;
  p.ntiles = B * p.tiles_img;
  p.fd_timg = make_fastdiv((uint32_t)p.tiles_img);
  p.fd_tx = make_fastdiv((uint32_t)p.tiles_x);
  p.per_xcd = (p.ntiles + 7) / 8;
  p.bpx = cu_limit(cws_num_cus()) / 8;
  if (p.bpx < 1) p.bpx = 1;
  if (p.bpx > p.per_xcd) p.bpx = p.per_xcd;
  hipLaunchKernelGGL(kern, dim3(8 * p.bpx), dim3(512), LDS, st, p);
  return launch_status();
}

// Images per virtual row (GRP forms): the smallest group that saves at least 4 % of the tile columns, 1 = plain tiling.  The group's
// descriptors must stay inside 31 bits.
static int cws_group(int W, int64_t img_bytes_max) {
  static const int on = knob("MOY_CWS_GROUP", 1);                      // MOY_CWS_GROUP=0: plain tiling (A/B runs, bit-identity test)
  if (!on) return 1;
  const double base = (double)((W + 15) / 16);
  int best = 1;
  double best_t = base;
  for (int g = 2; g <= 8; ++g) {
    if (img_bytes_max * g > 0x3fffffffLL) break;
    const double t = (double)((g * (W + 1) - 1 + 15) / 16) / g;
    if (t < best_t * 0.98 && t <= base * 0.96) { best = g; best_t = t; }
  }
  return best;
}

template <typename T, int C, int N, int TH, int WN, int NBUF, bool RES, int OCC = 1, bool SPREAD = false, int ABL = 0, bool GRP = false>
static int launch_conv_ws(ConvWsParams& p, int B, hipStream_t st) {
  if (plan_only(MOY_KERNEL_CONV_WS)) return MOY_OK;          // moy_gemm_query: the dispatch without the launch (and without touching the attribute statics)
  using G = CwsGeom<C, N, TH, NBUF, RES>;
  static_assert(G::LDS * OCC <= 160 * 1024, "LDS budget");
  if constexpr (ABL == 0 && !GRP && C == 128 && !SPREAD && OCC == 1) {
    const int64_t ldmax = p.lda > p.ldc ? (p.lda > p.ldr ? p.lda : p.ldr) : (p.ldc > p.ldr ? p.ldc : p.ldr);
    const int g = cws_group(p.Wd, (int64_t)p.H * p.Wd * ldmax * 2);
    if (g > 1 && B >= g) {
      p.G = g; p.nimg = B; p.Wv = g * (p.Wd + 1) - 1;
      p.fd_w1 = make_fastdiv((uint32_t)(p.Wd + 1));
      const int px = p.H * p.Wd - (p.Wd + 1);
      p.delta_a = px * (int)p.lda * 2; p.delta_r = px * (int)p.ldr * 2; p.delta_c = px * (int)p.ldc * 2;
      return launch_conv_ws<T, C, N, TH, WN, NBUF, RES, OCC, SPREAD, 0, true>(p, B, st);
    }
  }
#if MOY_DIAG
  if constexpr (ABL == 0 && !GRP && std::is_same<T, bf16_t>::value && !RES) {
    static const int abl = garbage_mode_env("MOY_CWS_ABL");                   // MOY_CWS_ABL=1..4: timing-only builds (no SiLU / no DMA / one MFMA group / no stores)
    if (abl == 1) return launch_conv_ws<T, C, N, TH, WN, NBUF, RES, OCC, SPREAD, 1>(p, B, st);
    if (abl == 2) return launch_conv_ws<T, C, N, TH, WN, NBUF, RES, OCC, SPREAD, 2>(p, B, st);
    if (abl == 3) return launch_conv_ws<T, C, N, TH, WN, NBUF, RES, OCC, SPREAD, 3>(p, B, st);
    if (abl == 4) return launch_conv_ws<T, C, N, TH, WN, NBUF, RES, OCC, SPREAD, 4>(p, B, st);
    if (abl == 5) return launch_conv_ws<T, C, N, TH, WN, NBUF, RES, OCC, SPREAD, 5>(p, B, st);
  }
#endif
  auto kern = conv_ws_kernel<T, C, N, TH, WN, NBUF, RES, OCC, SPREAD, ABL, GRP>;
  {
    static const int prio = knob("MOY_CWS_PRIO", 1);
    p.prio = prio;
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (G::LDS > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess)
      return MOY_ELAUNCH;
    attr_set = true;
  }
  p.tiles_x = ((GRP ? p.Wv : p.Wd) + 15) / 16;
  const int tiles_y = (p.H + TH - 1) / TH;
  p.tiles_img = p.tiles_x * tiles_y;
  p.ntiles = (GRP ? (B + p.G - 1) / p.G : B) * p.tiles_img;
  p.fd_timg = make_fastdiv((uint32_t)p.tiles_img);
  p.fd_tx = make_fastdiv((uint32_t)p.tiles_x);
  p.per_xcd = (p.ntiles + 7) / 8;
  p.bpx = cu_limit(cws_num_cus()) / 8 * OCC;         // resident blocks per XCD
  if (p.bpx < 1) p.bpx = 1;
  if (p.bpx > p.per_xcd) p.bpx = p.per_xcd;
  hipLaunchKernelGGL(kern, dim3(8 * p.bpx), dim3(512), G::LDS, st, p);
  return launch_status();
}

#if MOY_DIAG
static int cws_variant() {                 // MOY_CWS_VARIANT: A/B knob (bit 0: spread the DMA pieces, bit 1: C = 32 with two blocks per CU)
  static const int v = knob("MOY_CWS_VARIANT", 0);
  return v;
}
#endif

template <typename T>
static int conv_ws_dispatch(ConvWsParams& p, int B, int C, bool res, hipStream_t st) {
#if MOY_DIAG
  const int v = cws_variant();
  const bool sp = v & 1, occ2 = v & 2;
  if (v & 8) {                             // ping-pong form (round 3)
    if (C == 32) return res ? launch_conv_ws_pp<T, 32, 32, 16, 2, true>(p, B, st) : launch_conv_ws_pp<T, 32, 32, 16, 2, false>(p, B, st);
    if (C == 64) return res ? launch_conv_ws_pp<T, 64, 64, 16, 4, true>(p, B, st) : launch_conv_ws_pp<T, 64, 64, 16, 4, false>(p, B, st);
    if (C == 128) return res ? launch_conv_ws_pp<T, 128, 128, 8, 8, true>(p, B, st) : launch_conv_ws_pp<T, 128, 128, 8, 8, false>(p, B, st);
  }
  if (v & 4) {                             // software-pipelined form
    if (C == 32) {
      if (occ2) return res ? launch_conv_ws_pipe<T, 32, 32, 16, 2, true, 2>(p, B, st) : launch_conv_ws_pipe<T, 32, 32, 16, 2, false, 2>(p, B, st);
      return res ? launch_conv_ws_pipe<T, 32, 32, 16, 2, true>(p, B, st) : launch_conv_ws_pipe<T, 32, 32, 16, 2, false>(p, B, st);
    }
    if (C == 64) return res ? launch_conv_ws_pipe<T, 64, 64, 16, 4, true>(p, B, st) : launch_conv_ws_pipe<T, 64, 64, 16, 4, false>(p, B, st);
    if (C == 128) return res ? launch_conv_ws_pipe<T, 128, 128, 8, 8, true>(p, B, st) : launch_conv_ws_pipe<T, 128, 128, 8, 8, false>(p, B, st);
  }
#endif
  if (C == 32) {
#if MOY_DIAG
    if (occ2) return res ? launch_conv_ws<T, 32, 32, 16, 2, 2, true, 2>(p, B, st) : launch_conv_ws<T, 32, 32, 16, 2, 2, false, 2>(p, B, st);
    if (sp) return res ? launch_conv_ws<T, 32, 32, 16, 2, 3, true, 1, true>(p, B, st) : launch_conv_ws<T, 32, 32, 16, 2, 3, false, 1, true>(p, B, st);
#endif
    return res ? launch_conv_ws<T, 32, 32, 16, 2, 3, true>(p, B, st) : launch_conv_ws<T, 32, 32, 16, 2, 3, false>(p, B, st);
  }
  if (C == 64) {
    // (round 5: the two-column-group form MOY_CWS_WN64=2 -- measured slower, DESIGN.md round 4 item 4 -- spilled 15-90 registers: scratch
    //  traffic is vector-memory traffic the counted waits of the patch ring do not know about; it left the tree)
#if MOY_DIAG
    if (sp) return res ? launch_conv_ws<T, 64, 64, 16, 4, 2, true, 1, true>(p, B, st) : launch_conv_ws<T, 64, 64, 16, 4, 3, false, 1, true>(p, B, st);
#endif
    return res ? launch_conv_ws<T, 64, 64, 16, 4, 2, true>(p, B, st) : launch_conv_ws<T, 64, 64, 16, 4, 3, false>(p, B, st);
  }
  if (C == 128) {
#if MOY_DIAG
    if (sp) return res ? launch_conv_ws<T, 128, 128, 8, 8, 2, true, 1, true>(p, B, st) : launch_conv_ws<T, 128, 128, 8, 8, 2, false, 1, true>(p, B, st);
#endif
    return res ? launch_conv_ws<T, 128, 128, 8, 8, 2, true>(p, B, st) : launch_conv_ws<T, 128, 128, 8, 8, 2, false>(p, B, st);
  }
  return MOY_ENOSYS;
}

// Eligibility + dispatch; MOY_ENOSYS = not this kernel's shape (moy_gemm falls through to the other convolution paths).
int conv_ws_try(const moy_gemm_args* a, hipStream_t st) {
  static const int mode = knob("MOY_CONV_WS", 1);                    // MOY_CONV_WS: 0 = off, 1 = on for launches with enough tiles (default), 2 = whenever the shape fits
  if (!mode) return MOY_ENOSYS;
  if (a->dtype != MOY_BF16 && a->dtype != MOY_F16) return MOY_ENOSYS;
  if (a->ksize != 3 || a->act != MOY_ACT_SILU) return MOY_ENOSYS;
  if (a->stride == 2) {
    static const int s2 = knob("MOY_CONV_S2", 1);                    // MOY_CONV_S2=0 switches the stride-2 kernel off (A/B runs)
    const int C = a->Cin, N = a->N;
    const bool l1 = C == 32 && N == 64, l3 = C == 64 && N == 128;
    if (!s2 || !(l1 || l3) || a->K != 9 * C || a->R) return MOY_ENOSYS;
    // a 1x1 consumer folded into the launch (post_*): the 64 -> 128 form only, N x N weights, SiLU
    if (a->post_W && (!l3 || a->post_n != N || a->post_act != MOY_ACT_SILU || !aligned16(a->post_W) ||
                      (a->post_scale && !aligned16(a->post_scale)) || (a->post_shift && !aligned16(a->post_shift))))
      return MOY_ENOSYS;
    if (a->ln_g || a->out_f32 || a->c_rows_per_batch || a->pre || a->plane_cols || a->dot_n || !a->C) return MOY_ENOSYS;
    if ((a->lda % 8) || (a->ldc % 8) || !aligned16(a->A) || !aligned16(a->C) || !aligned16(a->W)) return MOY_ENOSYS;
    if ((a->scale && !aligned16(a->scale)) || (a->shift && !aligned16(a->shift))) return MOY_ENOSYS;
    if ((int64_t)a->Hin * a->Win * a->lda * 2 > 0x3fffffffLL || (int64_t)a->Hout * a->Wout * a->ldc * 2 > 0x3fffffffLL) return MOY_ENOSYS;
    const int TH = l1 ? 8 : 4;
    const long tiles = (long)a->B * ((a->Hout + TH - 1) / TH) * ((a->Wout + 15) / 16);
    const double util = (double)a->Hout * a->Wout * a->B / (double)(tiles * TH * 16);
    if (mode == 1 && (tiles < 2 * cws_num_cus() || util < 0.7)) return MOY_ENOSYS;
    ConvS2Params q{};
    q.A = a->A; q.lda = a->lda; q.W = a->W; q.Kpad = (a->K + 63) / 64 * 64; q.scale = a->scale; q.shift = a->shift;
    q.C = a->C; q.ldc = a->ldc; q.Hin = a->Hin; q.Win = a->Win; q.Hout = a->Hout; q.Wout = a->Wout;
    if (a->post_W) {
      q.W2 = a->post_W; q.Kpad2 = (N + 63) / 64 * 64; q.scale2 = a->post_scale; q.shift2 = a->post_shift;
      return a->dtype == MOY_BF16 ? launch_conv_s2<bf16_t, 64, 128, 4, 8, 2, true>(q, a->B, st) : launch_conv_s2<f16_t, 64, 128, 4, 8, 2, true>(q, a->B, st);
    }
    if (a->dtype == MOY_BF16) return l1 ? launch_conv_s2<bf16_t, 32, 64, 8, 4, 3>(q, a->B, st) : launch_conv_s2<bf16_t, 64, 128, 4, 8, 2>(q, a->B, st);
    return l1 ? launch_conv_s2<f16_t, 32, 64, 8, 4, 3>(q, a->B, st) : launch_conv_s2<f16_t, 64, 128, 4, 8, 2>(q, a->B, st);
  }
  if (a->stride != 1 || a->post_W) return MOY_ENOSYS;
  const int C = a->Cin;
  if ((C != 32 && C != 64 && C != 128) || a->N != C || a->K != 9 * C) return MOY_ENOSYS;
  if (a->ln_g || a->out_f32 || a->c_rows_per_batch || a->pre || a->plane_cols || a->dot_n || !a->C) return MOY_ENOSYS;
  if ((a->lda % 8) || (a->ldc % 8) || !aligned16(a->A) || !aligned16(a->C) || !aligned16(a->W)) return MOY_ENOSYS;
  if (a->R && ((a->ldr % 8) || !aligned16(a->R))) return MOY_ENOSYS;
  if ((a->scale && !aligned16(a->scale)) || (a->shift && !aligned16(a->shift))) return MOY_ENOSYS;
  const int64_t ldmax = a->lda > a->ldc ? (a->lda > a->ldr ? a->lda : a->ldr) : (a->ldc > a->ldr ? a->ldc : a->ldr);
  if ((int64_t)a->Hin * a->Win * ldmax * 2 > 0x3fffffffLL) return MOY_ENOSYS;   // per-image descriptors, 32-bit lane offsets
  const int TH = C == 128 ? 8 : 16;
  // (utilisation of the PLAIN tiling on purpose.  Round 6 measured the 34-pixel-wide level -- 0.56 of every tile used, 0.76 with the tile
  //  columns cut from a virtual row of five images -- on this kernel with the grouped tiling: 0.080 / 0.092 ms (plain / shortcut form)
  //  against the tiled kernel's 0.080 / 0.079 at 288 frames: no gain, the launch stays with the tiled kernel)
  const long tiles = (long)a->B * ((a->Hin + TH - 1) / TH) * ((a->Win + 15) / 16);
  const double util = (double)a->Hin * a->Win * a->B / (double)(tiles * TH * 16);
  if (mode == 1 && (tiles < 2 * cws_num_cus() || util < 0.7)) return MOY_ENOSYS;
  ConvWsParams p{};
  p.A = a->A; p.lda = a->lda; p.W = a->W;
  p.Kpad = (a->K + 63) / 64 * 64;
  p.scale = a->scale; p.shift = a->shift; p.R = a->R; p.ldr = a->R ? a->ldr : 0; p.C = a->C; p.ldc = a->ldc;
  p.H = a->Hin; p.Wd = a->Win;
  return a->dtype == MOY_BF16 ? conv_ws_dispatch<bf16_t>(p, a->B, C, a->R != nullptr, st)
                              : conv_ws_dispatch<f16_t>(p, a->B, C, a->R != nullptr, st);
}

}  // namespace moy
