// Deformable-attention sampling with the FIRST pyramid level gathered RAW and projected afterwards (round 5).
//
// MSDeformAttn.forward (nn/modules/transformer.py:246-287) projects all S tokens with `value_proj` (:255-257) and then samples
// 300 x 8 x 12 points of the projected maps (nn/modules/utils.py:41-78).  On the P3 level (76 x 136 tokens at 1088x608) a layer's
// 38.4 k taps touch at most 46 % of the 82.7 k head-cells it projected, and that launch was the longest of the plan (9.9 GB per 288
// frames, 9.1 GB of them writes).  Both maps are linear and the bilinear sum is linear, so for level 0
//
//     sum_p a_p . bilinear(W x + c)(loc_p)  =  W_h . ( sum_p a_p . bilinear(x)(loc_p) )  +  c_h . sum_p a_p . (in-range corner weights)
//
// with x the level's own 128-channel tensor, W = value_proj o BN o input_proj composed on the host (the fold of round 4) and c its
// bias (zero padding applies to the PROJECTED map, bias included: a corner outside the level contributes nothing, hence the weight
// sum beside c).  The value planes of level 0 are never formed.
//
// One block = 16 queries (4 waves x 4).  Per query a wave
//   * computes softmax / sampling locations with lane = head*8 + sub as `msda_planes_kernel` does and gathers levels 1.. from the
//     head planes exactly as that kernel (two x-corners per 16-byte lane load) -> fp32 partial row in LDS;
//   * gathers level 0 raw: for every (head, point) FOUR dword loads, one per corner -- 64 lanes x 4 bytes = the corner's 128 channels,
//     the address a wave-uniform SCALAR offset (v_readlane of the head's lane) + lane*4, no vector address arithmetic; a lane owns two
//     channels and sums its 16 taps per head in fp32 with scalar weights (no cross-lane reduction); corners outside the level are
//     handled by clamping the 2x2 window into the level and re-dealing the weights to the slots that survive;
//   * leaves g[head][128] (rounded to T: the MFMA operand) and the in-range weight sums in LDS.
// Then wave w multiplies heads 2w, 2w+1: W_h[32 x 128] (A operand, from L2) x g[16 queries] (B operand) on v_mfma_f32_16x16x32, adds
// c_h . s and the partial of the other levels, and stores the 16 x 64 output values it owns.
#include "common.hpp"

#include <type_traits>
#include <utility>

namespace moy {

struct LevelInfoR {
  int H[4], W[4], start[4];      // start[l] (l >= 1): first token of level l in the planes' own numbering (level 0 is not in them)
};

struct MsdaRawParams {
  const void* x0;        // level 0, raw: [B, H0, W0, >= 128] channels-last, pixel pitch ld0 elements
  int64_t ld0;
  const void* wc;        // composed weights of this layer: T [256][128] (wc_packed: in MFMA-fragment order, include/moyolo.h)
  int wc_packed;
  const float* bc;       // composed bias fp32 [256]
  const void* planes;    // head planes of levels 1..: T [8][B * S1][32] (head_stride elements between heads)
  int64_t head_stride;
  int S1;                // tokens per frame in the planes
  LevelInfoR lv;
  int L;
  const float* offaw;
  int64_t ld_oa;
  const float* ref;
  int Lq, nrows;
  int ngroups;           // ceil(nrows / queries per block)
  void* out;
  int64_t ldo;
  int B;
  const int32_t* perm;   // optional [B * Lq]: position (b, i) of the walk -> query perm[b * Lq + i] of frame b (moy_query_order); NULL: identity
};

template <typename T>
__device__ __forceinline__ f32x4 mr_mfma(f32x4 acc, u32x4 w, u32x4 a);
template <>
__device__ __forceinline__ f32x4 mr_mfma<bf16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mr_mfma<f16_t>(f32x4 acc, u32x4 w, u32x4 a) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
}

#ifndef MOY_MR_DEPTH
#define MOY_MR_DEPTH 1
#endif
constexpr int MR_QB = 16;                    // queries per block
constexpr int MR_GP = 8 * 128 * 2 + 16;      // bytes per query row of g (+16: the 16 rows of a fragment read start 4 banks apart)
constexpr int MR_PP = 256 * 4 + 16;          // bytes per query row of the other levels' partial (fp32)

template <typename T>
__global__ __launch_bounds__(256, 3) void msda_raw_kernel(const MsdaRawParams p) {
  static_assert(sizeof(T) == 2, "16-bit values");
  __shared__ __attribute__((aligned(16))) unsigned char sG[MR_QB * MR_GP];
  __shared__ __attribute__((aligned(16))) unsigned char sP[MR_QB * MR_PP];
  __shared__ float sS[MR_QB * 8];
  __shared__ int sRow[MR_QB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware walk: workgroup i runs on XCD i % 8, and every XCD has its own L2 -- XCD x takes the x-th eighth of the query groups,
  // in order, so that the few frames its resident blocks work on (a level-0 map is 2.6 MB at 1088x608) are re-read from ITS L2
  const int nblk = (int)gridDim.x, chunk = (nblk + 7) >> 3;
  const int lblk = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (lblk >= p.ngroups) return;
  const int row0 = lblk * MR_QB;
  const int m = lane >> 3, sub = lane & 7, cx = sub >> 2, oct = sub & 3;
  const int L = p.L, LP = L * 4;
  const T* planes = static_cast<const T*>(p.planes);
  const T* x0 = static_cast<const T*>(p.x0);
  constexpr uint32_t OOB = 0x80000000u;
  const int H0 = p.lv.H[0], W0 = p.lv.W[0];
  const uint32_t pix_pitch = (uint32_t)(p.ld0 * 2), row_pitch = (uint32_t)(W0 * p.ld0 * 2);

  for (int j = 0; j < 4; ++j) {
    const int ql = wave * 4 + j;                                   // query of the block
    const int pos = min(row0 + ql, p.nrows - 1);                   // positions past the end recompute the last one (never stored)
    const int b = __builtin_amdgcn_readfirstlane(pos / p.Lq);
    const int row = p.perm ? b * p.Lq + __builtin_amdgcn_readfirstlane(p.perm[pos]) : pos;      // the walk visits the frame's queries in perm order
    if (lane == 0) sRow[ql] = row;
    const float* oa = p.offaw + (long)row * p.ld_oa;
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
      if (i < LP) { logit[i] = __builtin_amdgcn_exp2f((logit[i] - mx) * 1.4426950408889634f); den += logit[i]; }
    const float inv_den = __builtin_amdgcn_rcpf(den);
    const f32x4 rb = *reinterpret_cast<const f32x4*>(p.ref + (long)row * 4);

    // ---- level 0, raw: per point the clamped 2x2 window (byte offset of its first pixel inside the frame) and the four slot weights
    uint32_t base0[4];
    float w00[4], w01[4], w10[4], w11[4];
    float sw = 0.f;
#pragma unroll
    for (int pnt = 0; pnt < 4; ++pnt) {
      // loc = ref_xy + off / n_points * ref_wh * 0.5   (transformer.py:280-282)
      const float lx = rb.x + offx[pnt] / 4.0f * rb.z * 0.5f;
      const float ly = rb.y + offy[pnt] / 4.0f * rb.w * 0.5f;
      const float x = lx * W0 - 0.5f, y = ly * H0 - 0.5f;
      const float aw = logit[pnt] * inv_den;
      const float xf = floorf(x), yf = floorf(y);
      const float fx = x - xf, fy = y - yf;
      // (a sample far outside the level: clamp before the int conversion)
      const int xi = (int)fminf(fmaxf(xf, -2.0f), (float)W0 + 1.0f), yi = (int)fminf(fmaxf(yf, -2.0f), (float)H0 + 1.0f);
      const float ax0 = (unsigned)xi < (unsigned)W0 ? 1.f - fx : 0.f, ax1 = (unsigned)(xi + 1) < (unsigned)W0 ? fx : 0.f;
      const float ay0 = (unsigned)yi < (unsigned)H0 ? 1.f - fy : 0.f, ay1 = (unsigned)(yi + 1) < (unsigned)H0 ? fy : 0.f;
      const int xb = min(max(xi, 0), W0 - 2), yb = min(max(yi, 0), H0 - 2);
      // slot s of the clamped window holds pixel xb + s: which real corner (if any) is that
      const float sx0 = xi == xb ? ax0 : (xi + 1 == xb ? ax1 : 0.f), sx1 = xi == xb ? ax1 : (xi == xb + 1 ? ax0 : 0.f);
      const float sy0 = yi == yb ? ay0 : (yi + 1 == yb ? ay1 : 0.f), sy1 = yi == yb ? ay1 : (yi == yb + 1 ? ay0 : 0.f);
      base0[pnt] = (uint32_t)(yb * W0 + xb) * pix_pitch;
      w00[pnt] = aw * sx0 * sy0; w01[pnt] = aw * sx1 * sy0; w10[pnt] = aw * sx0 * sy1; w11[pnt] = aw * sx1 * sy1;
      sw += aw * (ax0 + ax1) * (ay0 + ay1);
    }
    if (sub == 0) sS[ql * 8 + m] = sw;

    // ---- levels 1..: head planes, as msda_planes_kernel (one level's taps in flight)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(planes + (int64_t)b * p.S1 * 32), 0, 0x80000000u, 0x00020000);
      const uint32_t lane_off = (uint32_t)((m * p.head_stride + oct * 8) * 2);
#pragma unroll
      for (int l = 1; l < 4; ++l)
        if (l < L) {
          const int H = p.lv.H[l], W = p.lv.W[l];
          u32x4 tap[4][2];
          float wgt[4][2];
#pragma unroll
          for (int pnt = 0; pnt < 4; ++pnt) {
            const int i = l * 4 + pnt;
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
              const uint32_t off = lane_off + (uint32_t)((p.lv.start[l] + yi * W + xi) * 64);
              tap[pnt][t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0));
              wgt[pnt][t] = wx * (t ? fy : 1.f - fy);
            }
          }
#pragma unroll
          for (int pnt = 0; pnt < 4; ++pnt)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const u32x4 w = tap[pnt][t];
              const float g = wgt[pnt][t];
              acc[0] = DT<T>::fma_lo(w.x, g, acc[0]); acc[1] = DT<T>::fma_hi(w.x, g, acc[1]); acc[2] = DT<T>::fma_lo(w.y, g, acc[2]); acc[3] = DT<T>::fma_hi(w.y, g, acc[3]);
              acc[4] = DT<T>::fma_lo(w.z, g, acc[4]); acc[5] = DT<T>::fma_hi(w.z, g, acc[5]); acc[6] = DT<T>::fma_lo(w.w, g, acc[6]); acc[7] = DT<T>::fma_hi(w.w, g, acc[7]);
            }
        }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], 4);
      if (cx == 0) {
        float* dst = reinterpret_cast<float*>(sP + ql * MR_PP) + m * 32 + oct * 8;
        *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
      }
    }

    // ---- level 0 gather: head by head, four corners x four points = 16 dword loads per head, scalar offsets and weights
    {
      const int64_t frame = (int64_t)H0 * W0 * p.ld0;
      const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x0 + (int64_t)b * frame), 0, (uint32_t)(frame * 2), 0x00020000);
      const uint32_t voff = (uint32_t)lane * 4u;
      uint32_t* gdst = reinterpret_cast<uint32_t*>(sG + ql * MR_GP) + lane;
      // two heads' taps in flight: head h+1 is requested before head h is summed (32 dword loads outstanding per wave)
      constexpr int DEPTH = MOY_MR_DEPTH;                           // heads whose taps are in flight beside the one being summed
      uint32_t tp[DEPTH + 1][4][4];
      auto issue = [&](auto hc, uint32_t (&t)[4][4]) {
        constexpr int h = decltype(hc)::value;
#pragma unroll
        for (int pnt = 0; pnt < 4; ++pnt) {
          const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)base0[pnt], h * 8);
          t[pnt][0] = __builtin_amdgcn_raw_buffer_load_b32(rs0, voff, sb, 0);
          t[pnt][1] = __builtin_amdgcn_raw_buffer_load_b32(rs0, voff, sb + pix_pitch, 0);
          t[pnt][2] = __builtin_amdgcn_raw_buffer_load_b32(rs0, voff, sb + row_pitch, 0);
          t[pnt][3] = __builtin_amdgcn_raw_buffer_load_b32(rs0, voff, sb + row_pitch + pix_pitch, 0);
        }
      };
      auto consume = [&](auto hc, const uint32_t (&t)[4][4]) {
        constexpr int h = decltype(hc)::value;
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int pnt = 0; pnt < 4; ++pnt) {
          const float ws[4] = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w00[pnt]), h * 8)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w01[pnt]), h * 8)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w10[pnt]), h * 8)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w11[pnt]), h * 8))};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            g0 = DT<T>::fma_lo_s(t[pnt][c], ws[c], g0);
            g1 = DT<T>::fma_hi_s(t[pnt][c], ws[c], g1);
          }
        }
        gdst[h * 64] = DT<T>::pack2(g0, g1);                        // channels 2*lane, 2*lane + 1 of head h's gathered vector
      };
      [&]<int... Hs>(std::integer_sequence<int, Hs...>) { (issue(std::integral_constant<int, Hs>{}, tp[Hs]), ...); }
      (std::make_integer_sequence<int, DEPTH>{});
      [&]<int... Hs>(std::integer_sequence<int, Hs...>) {
        (([&] {
           if constexpr (Hs + DEPTH < 8) issue(std::integral_constant<int, Hs + DEPTH>{}, tp[(Hs + DEPTH) % (DEPTH + 1)]);
           __builtin_amdgcn_sched_barrier(0);                       // keep the later heads' requests ahead of this head's sums
           consume(std::integral_constant<int, Hs>{}, tp[Hs % (DEPTH + 1)]);
         }()), ...);
      }(std::make_integer_sequence<int, 8>{});
    }
  }
  __syncthreads();

  // ---- projection: wave w owns heads 2w, 2w+1.  A = W_h rows (out channel t*16 + r), B = g of the 16 queries (column r = query r)
  const int r = lane & 15, q4 = lane >> 4;
  const T* wc = static_cast<const T*>(p.wc);
  T* out = static_cast<T*>(p.out);
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int h = wave * 2 + hh;
    u32x4 wa[2][4], gb[4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pn = 0; pn < 4; ++pn)
        wa[t][pn] = p.wc_packed ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(wc) + ((h * 4 + pn) * 2 + t) * 1024 + lane * 16)   // fragment order: 8 KB contiguous per head
                                : *reinterpret_cast<const u32x4*>(wc + (h * 32 + t * 16 + r) * 128 + pn * 32 + q4 * 8);
#pragma unroll
    for (int pn = 0; pn < 4; ++pn) gb[pn] = *reinterpret_cast<const u32x4*>(sG + r * MR_GP + h * 256 + pn * 64 + q4 * 16);
    f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int pn = 0; pn < 4; ++pn)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc2[t] = mr_mfma<T>(acc2[t], wa[t][pn], gb[pn]);
    const float s = sS[r * 8 + h];
    const int row = sRow[r];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ch = h * 32 + t * 16 + q4 * 4;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bc + ch);
      const f32x4 part = *reinterpret_cast<const f32x4*>(sP + r * MR_PP + ch * 4);
      const f32x4 v = acc2[t] + bias * s + part;
      if (row0 + r < p.nrows) *reinterpret_cast<u32x2*>(out + (int64_t)row * p.ldo + ch) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same operator with the TAP SUMS ON THE MATRIX CORES (round 5, second form; three levels, every level at least 2 x 2).
//
// The form above is bound by the vector ALU: every tap dword is unpacked and multiplied by its weight with three to four vector
// instructions per lane (1 527 per query).  A bilinear sum is a [1 x taps] x [taps x channels] product, and the 16 x 16 x 32 MFMA has a
// shape for it that needs NO data movement between the load and the product:
//
//   B operand = the tap data as loaded.  Lane (n = lane & 15, kg = lane >> 4) loads ONE DWORD (channels 2n, 2n+1 of a 32-channel chunk) of
//     each of the FOUR CORNERS of sampling point kg: its four result registers are k = kg*8 .. kg*8+7 = (corner 0..3) x (even, odd channel)
//     of column n -- the B fragment, untouched.
//   A operand = the 16 weights of the (head, level): row 0 carries w[tap] at the even k of its tap and 0 at the odd k, row 1 the reverse,
//     rows 2 / 3 the same with the REMAINDERS w - T(w) (the weights enter the matrix core in T: two terms keep 16 / 22 bits of them); rows
//     4-15 are zero.  Lanes n < 4 read their four weights from a per-wave LDS table with one ds_read_b128 and mask them with a lane constant.
//   D: lanes 0-15 hold, for channel pair n, {even, odd} x {head, remainder} sums: two adds, one pack.
//
// One MFMA = one head x one level x 32 channels: level 0 takes 4 per head (128 channels), levels 1 / 2 one each, accumulated into the
// same D.  The table (corner weights and the byte offset of the clamped 2 x 2 window of every (head, point)) is written by the wave itself
// with lane = head*8 + sub owning point `sub` (levels 0 / 1) and point 8 + sub (level 2): softmax, location and window arithmetic happen
// ONCE per point instead of once per head lane, and there are no v_readlane.  Per query ~420 vector instructions instead of 1 527.
// A block is 16 queries = 8 waves x 2; wave w projects head w.
#ifndef MOY_MRM_DEPTH
#define MOY_MRM_DEPTH 1
#endif
#ifndef MOY_MRM_AUX
#define MOY_MRM_AUX 0        // cache-policy bits of the tap loads (measured: DESIGN.md)
#endif
constexpr int MRM_NW = 8;                    // waves per block
constexpr int MRM_ENT = 96;                  // (head, point) entries per query: 8 x 12

// X4 (round 6): level 0 by 16-BYTE loads.  The counters of round 6 (profiles/r06_gather_order_*) show the kernel bound by the texture
// addresser / L1 pipeline -- busy 85 % of the kernel's cycles at ~8 cycles per wave instruction -- not by misses (a Morton walk order
// raised the L1 hit rate from 72.9 to 76.9 % and changed nothing), so what counts is the NUMBER of vector-memory instructions.  Lane
// (n, kg) loads channels 8n .. 8n+7 of each of the four corners of point kg: 16 lanes = one whole 256-byte pixel, FOUR instructions per
// head instead of sixteen.  The B fragment of product m is then dword m of the four corner registers -- (corner) x (even, odd) of channel
// pair 4n + m, still no instruction between the load and the product -- and D's columns are the channel pairs 4n + m: the lane's four
// results are four consecutive dwords of g (one 16-byte LDS store).  The same products on the same values: bit-identical.
template <typename T, bool X4 = true>
__global__ __launch_bounds__(64 * MRM_NW, 2) void msda_raw_mfma_kernel(const MsdaRawParams p) {
  static_assert(sizeof(T) == 2, "16-bit values");
  __shared__ __attribute__((aligned(16))) unsigned char sG[MR_QB * MR_GP];
  __shared__ __attribute__((aligned(16))) unsigned char sP[MR_QB * MR_PP];
  __shared__ __attribute__((aligned(16))) uint32_t sT[MRM_NW * MRM_ENT * 8];     // per wave: [entry][T(w) x 4 | remainder x 4], each 16-bit value in both halves
  __shared__ uint32_t sB[MRM_NW * MRM_ENT];                                       // per wave: byte offset of the entry's window
  __shared__ float sS[MR_QB * 8];
  __shared__ int sRow[MR_QB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nblk = (int)gridDim.x, chunk = (nblk + 7) >> 3;
  const int lblk = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (lblk >= p.ngroups) return;
  const int row0 = lblk * MR_QB;
  const int m = lane >> 3, sub = lane & 7;           // table building: head m, point sub
  const int n = lane & 15, kg = lane >> 4;           // matrix layout: column / row n, k group kg
  const T* planes = static_cast<const T*>(p.planes);
  const T* x0 = static_cast<const T*>(p.x0);
  const int H0 = p.lv.H[0], W0 = p.lv.W[0], H1 = p.lv.H[1], W1 = p.lv.W[1], H2 = p.lv.H[2], W2 = p.lv.W[2];
  const uint32_t pix_pitch = (uint32_t)(p.ld0 * 2), row_pitch = (uint32_t)(W0 * p.ld0 * 2);
  const uint32_t hs2 = (uint32_t)(p.head_stride * 2);
  const uint32_t amask = n < 4 ? ((n & 1) ? 0xffff0000u : 0x0000ffffu) : 0u;
  uint32_t* myT = sT + wave * MRM_ENT * 8;
  uint32_t* myB = sB + wave * MRM_ENT;
  const uint32_t* aT = myT + kg * 8 + ((n & 2) ? 4 : 0);      // + entry*8 of the unit's first point
  const uint32_t* aB = myB + kg;
  constexpr float LOG2E = 1.4426950408889634f;

  for (int j = 0; j < 2; ++j) {
    const int ql = wave * 2 + j;
    const int pos = min(row0 + ql, p.nrows - 1);
    const int b = __builtin_amdgcn_readfirstlane(pos / p.Lq);
    const int row = p.perm ? b * p.Lq + __builtin_amdgcn_readfirstlane(p.perm[pos]) : pos;      // the walk visits the frame's queries in perm order
    if (lane == 0) sRow[ql] = row;
    const float* oa = p.offaw + (long)row * p.ld_oa;
    const float* offp = oa + m * 24;
    const float* awp = oa + 192 + m * 12;
    // softmax statistics of the head's 12 logits (every lane of the head)
    float mx, inv_den;
    {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(awp), a1 = *reinterpret_cast<const f32x4*>(awp + 4), a2 = *reinterpret_cast<const f32x4*>(awp + 8);
      mx = fmaxf(fmaxf(fmaxf(a0.x, a0.y), fmaxf(a0.z, a0.w)), fmaxf(fmaxf(fmaxf(a1.x, a1.y), fmaxf(a1.z, a1.w)), fmaxf(fmaxf(a2.x, a2.y), fmaxf(a2.z, a2.w))));
      float den = 0.f;
      const float lg[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
#pragma unroll
      for (int i = 0; i < 12; ++i) den += __builtin_amdgcn_exp2f((lg[i] - mx) * LOG2E);     // same order as the first form
      inv_den = __builtin_amdgcn_rcpf(den);
    }
    const f32x4 rb = *reinterpret_cast<const f32x4*>(p.ref + (long)row * 4);
    // one sampling point: clamped 2 x 2 window (slot s holds pixel xb + s) with the weights re-dealt to the slots that survive;
    // returns the point's in-range weight sum
    auto point = [&](int i, int Hl, int Wl, uint32_t tok0, uint32_t tok_pitch, bool store) -> float {
      const float2 o = *reinterpret_cast<const float2*>(offp + 2 * i);
      const float aw = __builtin_amdgcn_exp2f((awp[i] - mx) * LOG2E) * inv_den;
      // loc = ref_xy + off / n_points * ref_wh * 0.5   (transformer.py:280-282)
      const float lx = rb.x + o.x / 4.0f * rb.z * 0.5f;
      const float ly = rb.y + o.y / 4.0f * rb.w * 0.5f;
      const float x = lx * Wl - 0.5f, y = ly * Hl - 0.5f;
      const float xf = floorf(x), yf = floorf(y);
      const float fx = x - xf, fy = y - yf;
      const int xi = (int)fminf(fmaxf(xf, -2.0f), (float)Wl + 1.0f), yi = (int)fminf(fmaxf(yf, -2.0f), (float)Hl + 1.0f);
      const float ax0 = (unsigned)xi < (unsigned)Wl ? 1.f - fx : 0.f, ax1 = (unsigned)(xi + 1) < (unsigned)Wl ? fx : 0.f;
      const float ay0 = (unsigned)yi < (unsigned)Hl ? 1.f - fy : 0.f, ay1 = (unsigned)(yi + 1) < (unsigned)Hl ? fy : 0.f;
      const int xb = min(max(xi, 0), Wl - 2), yb = min(max(yi, 0), Hl - 2);
      const float sx0 = xi == xb ? ax0 : (xi + 1 == xb ? ax1 : 0.f), sx1 = xi == xb ? ax1 : (xi == xb + 1 ? ax0 : 0.f);
      const float sy0 = yi == yb ? ay0 : (yi + 1 == yb ? ay1 : 0.f), sy1 = yi == yb ? ay1 : (yi == yb + 1 ? ay0 : 0.f);
      const float ws[4] = {aw * sx0 * sy0, aw * sx1 * sy0, aw * sx0 * sy1, aw * sx1 * sy1};
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        hi[c] = DT<T>::pack2(ws[c], ws[c]);
        const float rem = ws[c] - DT<T>::lo(hi[c]);
        lo[c] = DT<T>::pack2(rem, rem);
      }
      if (store) {
        const int e = m * 12 + i;
        *reinterpret_cast<u32x4*>(myT + e * 8) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(myT + e * 8 + 4) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        myB[e] = (tok0 + (uint32_t)(yb * Wl + xb)) * tok_pitch;
      }
      return aw * (ax0 + ax1) * (ay0 + ay1);
    };
    {
      const bool l1 = sub >= 4;                                                    // points 0-3: level 0, 4-7: level 1
      float sw = point(sub, l1 ? H1 : H0, l1 ? W1 : W0, l1 ? (uint32_t)p.lv.start[1] : 0u, l1 ? 64u : pix_pitch, true);
      sw = l1 ? 0.f : sw;
      sw += __shfl_xor(sw, 1); sw += __shfl_xor(sw, 2);
      if (sub == 0) sS[ql * 8 + m] = sw;
      point(8 + (sub & 3), H2, W2, (uint32_t)p.lv.start[2], 64u, sub < 4);          // level 2
    }

    // ---- the gather: 12 units of 16 dword loads + 4 MFMAs; unit u + 1 is requested before unit u is multiplied
    //   units 0-7: level 0 of head u (one A operand, four 32-channel chunks);  units 8-11: levels 1 / 2 of heads 2(u-8), 2(u-8)+1
    const int64_t frame = (int64_t)H0 * W0 * p.ld0;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x0 + (int64_t)b * frame), 0, (uint32_t)(frame * 2), 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(planes + (int64_t)b * p.S1 * 32), 0, 0x80000000u, 0x00020000);
    const uint32_t nb = (uint32_t)n * 4u;
    uint32_t* gdst = reinterpret_cast<uint32_t*>(sG + ql * MR_GP);
    float* pdst = reinterpret_cast<float*>(sP + ql * MR_PP);
    constexpr int DEPTH = MOY_MRM_DEPTH;                            // units whose taps are in flight beside the one being multiplied
    uint32_t tp[DEPTH + 1][16];
    auto issue = [&](auto uc, uint32_t (&t)[16]) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u < 8 && X4) {
        const uint32_t voff = aB[u * 12] + (uint32_t)n * 16u;
        const u32x4 c0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, 0, MOY_MRM_AUX));
        const u32x4 c1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, pix_pitch, MOY_MRM_AUX));
        const u32x4 c2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, row_pitch, MOY_MRM_AUX));
        const u32x4 c3 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, row_pitch + pix_pitch, MOY_MRM_AUX));
        t[0] = c0.x; t[1] = c0.y; t[2] = c0.z; t[3] = c0.w;  t[4] = c1.x; t[5] = c1.y; t[6] = c1.z; t[7] = c1.w;        // t[corner * 4 + m]
        t[8] = c2.x; t[9] = c2.y; t[10] = c2.z; t[11] = c2.w;  t[12] = c3.x; t[13] = c3.y; t[14] = c3.z; t[15] = c3.w;
      } else if constexpr (u < 8) {
        const uint32_t voff = aB[u * 12] + nb;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          t[cc * 4 + 0] = __builtin_amdgcn_raw_buffer_load_b32(rs0, voff + cc * 64, 0, MOY_MRM_AUX);
          t[cc * 4 + 1] = __builtin_amdgcn_raw_buffer_load_b32(rs0, voff + cc * 64, pix_pitch, MOY_MRM_AUX);
          t[cc * 4 + 2] = __builtin_amdgcn_raw_buffer_load_b32(rs0, voff + cc * 64, row_pitch, MOY_MRM_AUX);
          t[cc * 4 + 3] = __builtin_amdgcn_raw_buffer_load_b32(rs0, voff + cc * 64, row_pitch + pix_pitch, MOY_MRM_AUX);
        }
      } else {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int l = 1; l < 3; ++l) {
            constexpr int h0 = (u - 8) * 2;
            const uint32_t voff = aB[(h0 + hh) * 12 + l * 4] + nb;
            const uint32_t sh = (uint32_t)(h0 + hh) * hs2, rp = (uint32_t)(l == 1 ? W1 : W2) * 64u;
            uint32_t* tt = t + (hh * 2 + l - 1) * 4;
            tt[0] = __builtin_amdgcn_raw_buffer_load_b32(rs1, voff, sh, MOY_MRM_AUX);
            tt[1] = __builtin_amdgcn_raw_buffer_load_b32(rs1, voff, sh + 64u, MOY_MRM_AUX);
            tt[2] = __builtin_amdgcn_raw_buffer_load_b32(rs1, voff, sh + rp, MOY_MRM_AUX);
            tt[3] = __builtin_amdgcn_raw_buffer_load_b32(rs1, voff, sh + rp + 64u, MOY_MRM_AUX);
          }
      }
    };
    auto weights = [&](int e) {                                    // the A operand of entry e .. e+3 (the four points of a head and level)
      const u32x4 w = *reinterpret_cast<const u32x4*>(aT + e * 8);
      return u32x4{w.x & amask, w.y & amask, w.z & amask, w.w & amask};
    };
    auto consume = [&](auto uc, const uint32_t (&t)[16]) {
      constexpr int u = decltype(uc)::value;
      const f32x4 z{0.f, 0.f, 0.f, 0.f};
      if constexpr (u < 8 && X4) {
        const u32x4 a = weights(u * 12);
        f32x4 d[4];
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) d[mm] = mr_mfma<T>(z, a, u32x4{t[mm], t[4 + mm], t[8 + mm], t[12 + mm]});
        if (lane < 16)        // channel pairs 4 * lane .. 4 * lane + 3 of head u
          *reinterpret_cast<u32x4*>(gdst + u * 64 + lane * 4) = u32x4{DT<T>::pack2(d[0].x + d[0].z, d[0].y + d[0].w), DT<T>::pack2(d[1].x + d[1].z, d[1].y + d[1].w),
                                                                      DT<T>::pack2(d[2].x + d[2].z, d[2].y + d[2].w), DT<T>::pack2(d[3].x + d[3].z, d[3].y + d[3].w)};
      } else if constexpr (u < 8) {
        const u32x4 a = weights(u * 12);
        f32x4 d[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) d[cc] = mr_mfma<T>(z, a, u32x4{t[cc * 4], t[cc * 4 + 1], t[cc * 4 + 2], t[cc * 4 + 3]});
        if (lane < 16) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) gdst[u * 64 + cc * 16 + lane] = DT<T>::pack2(d[cc].x + d[cc].z, d[cc].y + d[cc].w);
        }
      } else {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          constexpr int h0 = (u - 8) * 2;
          const u32x4 a1 = weights((h0 + hh) * 12 + 4), a2 = weights((h0 + hh) * 12 + 8);
          f32x4 d = mr_mfma<T>(z, a1, u32x4{t[hh * 8], t[hh * 8 + 1], t[hh * 8 + 2], t[hh * 8 + 3]});
          d = mr_mfma<T>(d, a2, u32x4{t[hh * 8 + 4], t[hh * 8 + 5], t[hh * 8 + 6], t[hh * 8 + 7]});
          if (lane < 16) *reinterpret_cast<float2*>(pdst + (h0 + hh) * 32 + lane * 2) = float2{d.x + d.z, d.y + d.w};
        }
      }
    };
    [&]<int... Us>(std::integer_sequence<int, Us...>) { (issue(std::integral_constant<int, Us>{}, tp[Us]), ...); }
    (std::make_integer_sequence<int, DEPTH>{});
    [&]<int... Us>(std::integer_sequence<int, Us...>) {
      (([&] {
         if constexpr (Us + DEPTH < 12) issue(std::integral_constant<int, Us + DEPTH>{}, tp[(Us + DEPTH) % (DEPTH + 1)]);
         __builtin_amdgcn_sched_barrier(0);
         consume(std::integral_constant<int, Us>{}, tp[Us % (DEPTH + 1)]);
       }()), ...);
    }(std::make_integer_sequence<int, 12>{});
  }
  __syncthreads();

  // ---- projection: wave w owns head w.  A = W_h rows (out channel t*16 + r), B = g of the 16 queries (column r = query r)
  const int r = lane & 15, q4 = lane >> 4;
  const T* wc = static_cast<const T*>(p.wc);
  T* out = static_cast<T*>(p.out);
  {
    const int h = wave;
    u32x4 wa[2][4], gb[4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pn = 0; pn < 4; ++pn)
        wa[t][pn] = p.wc_packed ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(wc) + ((h * 4 + pn) * 2 + t) * 1024 + lane * 16)   // fragment order: 8 KB contiguous per head
                                : *reinterpret_cast<const u32x4*>(wc + (h * 32 + t * 16 + r) * 128 + pn * 32 + q4 * 8);
#pragma unroll
    for (int pn = 0; pn < 4; ++pn) gb[pn] = *reinterpret_cast<const u32x4*>(sG + r * MR_GP + h * 256 + pn * 64 + q4 * 16);
    f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int pn = 0; pn < 4; ++pn)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc2[t] = mr_mfma<T>(acc2[t], wa[t][pn], gb[pn]);
    const float s = sS[r * 8 + h];
    const int row = sRow[r];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ch = h * 32 + t * 16 + q4 * 4;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bc + ch);
      const f32x4 part = *reinterpret_cast<const f32x4*>(sP + r * MR_PP + ch * 4);
      const f32x4 v = acc2[t] + bias * s + part;
      if (row0 + r < p.nrows) *reinterpret_cast<u32x2*>(out + (int64_t)row * p.ldo + ch) = u32x2{DT<T>::pack2(v.x, v.y), DT<T>::pack2(v.z, v.w)};
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same operator for the fp32 engines (round 6: MOY_F32, and MOY_F32X3 whose tensors are fp32 too): level 0 gathered raw from the
// fp32 P3 tensor (a lane owns two channels of a corner: one 8-byte load, 512 contiguous bytes per wave instruction), levels 1.. from
// fp32 head planes, every sum in fp32, and the per-head projection on the EXACT fp32 matrix instruction (v_mfma_f32_16x16x4_f32) --
// at 2 x 256 x 128 flops per query the projection is 1 % of the engine's arithmetic, so the split-fp16 engine takes the exact form too.
// A block is 8 queries (4 waves x 2): the gathered vectors of a block are 8 x 8 heads x 128 fp32 = 32 KB of LDS; the projection's
// B operand has 16 columns, the upper eight repeat the lower ones and are not stored.
constexpr int MRF_QB = 8;
constexpr int MRF_GP = 8 * 128 * 4 + 16;     // bytes per query row of g (fp32; +16: fragment rows start 4 banks apart)

__global__ __launch_bounds__(256, 2) void msda_raw_f32_kernel(const MsdaRawParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char sG[MRF_QB * MRF_GP];
  __shared__ __attribute__((aligned(16))) unsigned char sP[MRF_QB * MR_PP];
  __shared__ float sS[MRF_QB * 8];
  __shared__ int sRow[MRF_QB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nblk = (int)gridDim.x, chunk = (nblk + 7) >> 3;
  const int lblk = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);          // XCD-aware walk, as above
  if (lblk >= p.ngroups) return;
  const int row0 = lblk * MRF_QB;
  const int m = lane >> 3, sub = lane & 7, cx = sub >> 2, oct = sub & 3;
  const int L = p.L, LP = L * 4;
  const float* planes = static_cast<const float*>(p.planes);
  const float* x0 = static_cast<const float*>(p.x0);
  constexpr uint32_t OOB = 0x80000000u;
  const int H0 = p.lv.H[0], W0 = p.lv.W[0];
  const uint32_t pix_pitch = (uint32_t)(p.ld0 * 4), row_pitch = (uint32_t)(W0 * p.ld0 * 4);

  for (int j = 0; j < 2; ++j) {
    const int ql = wave * 2 + j;
    const int pos = min(row0 + ql, p.nrows - 1);                   // positions past the end recompute the last one (never stored)
    const int b = __builtin_amdgcn_readfirstlane(pos / p.Lq);
    const int row = p.perm ? b * p.Lq + __builtin_amdgcn_readfirstlane(p.perm[pos]) : pos;
    if (lane == 0) sRow[ql] = row;
    const float* oa = p.offaw + (long)row * p.ld_oa;
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
      if (i < LP) { logit[i] = expf(logit[i] - mx); den += logit[i]; }       // (accurate exponential: this is the exact class)
    const float inv_den = 1.0f / den;
    const f32x4 rb = *reinterpret_cast<const f32x4*>(p.ref + (long)row * 4);

    // ---- level 0, raw: per point the clamped 2x2 window and the four slot weights (as the 16-bit form)
    uint32_t base0[4];
    float w00[4], w01[4], w10[4], w11[4];
    float sw = 0.f;
#pragma unroll
    for (int pnt = 0; pnt < 4; ++pnt) {
      const float lx = rb.x + offx[pnt] / 4.0f * rb.z * 0.5f;      // loc = ref_xy + off / n_points * ref_wh * 0.5   (transformer.py:280-282)
      const float ly = rb.y + offy[pnt] / 4.0f * rb.w * 0.5f;
      const float x = lx * W0 - 0.5f, y = ly * H0 - 0.5f;
      const float aw = logit[pnt] * inv_den;
      const float xf = floorf(x), yf = floorf(y);
      const float fx = x - xf, fy = y - yf;
      const int xi = (int)fminf(fmaxf(xf, -2.0f), (float)W0 + 1.0f), yi = (int)fminf(fmaxf(yf, -2.0f), (float)H0 + 1.0f);
      const float ax0 = (unsigned)xi < (unsigned)W0 ? 1.f - fx : 0.f, ax1 = (unsigned)(xi + 1) < (unsigned)W0 ? fx : 0.f;
      const float ay0 = (unsigned)yi < (unsigned)H0 ? 1.f - fy : 0.f, ay1 = (unsigned)(yi + 1) < (unsigned)H0 ? fy : 0.f;
      const int xb = min(max(xi, 0), W0 - 2), yb = min(max(yi, 0), H0 - 2);
      const float sx0 = xi == xb ? ax0 : (xi + 1 == xb ? ax1 : 0.f), sx1 = xi == xb ? ax1 : (xi == xb + 1 ? ax0 : 0.f);
      const float sy0 = yi == yb ? ay0 : (yi + 1 == yb ? ay1 : 0.f), sy1 = yi == yb ? ay1 : (yi == yb + 1 ? ay0 : 0.f);
      base0[pnt] = (uint32_t)(yb * W0 + xb) * pix_pitch;
      w00[pnt] = aw * sx0 * sy0; w01[pnt] = aw * sx1 * sy0; w10[pnt] = aw * sx0 * sy1; w11[pnt] = aw * sx1 * sy1;
      sw += aw * (ax0 + ax1) * (ay0 + ay1);
    }
    if (sub == 0) sS[ql * 8 + m] = sw;

    // ---- levels 1..: fp32 head planes; lane (m, cx, oct) reads the 8 channels oct*8.. of its x-corner: two 16-byte loads per tap row
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
      const int64_t ext = (7 * p.head_stride + (int64_t)(p.B - b) * p.S1 * 32) * 4;       // bytes addressable from this frame's first token
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + (int64_t)b * p.S1 * 32), 0,
                                                        (uint32_t)(ext < 0x7fffffffLL ? ext : 0x7fffffffLL), 0x00020000);
      const uint32_t lane_off = (uint32_t)((m * p.head_stride + oct * 8) * 4);
#pragma unroll
      for (int l = 1; l < 4; ++l)
        if (l < L) {
          const int H = p.lv.H[l], W = p.lv.W[l];
          // two points (8 sixteen-byte loads per lane) in flight at a time: all four would cost 64 registers and spill
#pragma unroll
          for (int ph = 0; ph < 4; ph += 2) {
            f32x4 tap[2][2][2];
            float wgt[2][2];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
              const int i = l * 4 + ph + pp;
              const float lx = rb.x + offx[i] / 4.0f * rb.z * 0.5f;
              const float ly = rb.y + offy[i] / 4.0f * rb.w * 0.5f;
              const float x = lx * W - 0.5f, y = ly * H - 0.5f;
              const float aw = logit[i] * inv_den;
              const float xf = floorf(x), yf = floorf(y);
              const float fx = x - xf, fy = y - yf;
              const int xi = (int)fminf(fmaxf(xf, -2.0f), (float)W + 1.0f) + cx, y0 = (int)fminf(fmaxf(yf, -2.0f), (float)H + 1.0f);
              const float wx = (cx ? fx : 1.f - fx) * aw;
              const bool xin = (unsigned)xi < (unsigned)W;
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                const int yi = y0 + t;
                const bool ok = xin && (unsigned)yi < (unsigned)H;
                const uint32_t off = lane_off + (uint32_t)((p.lv.start[l] + yi * W + xi) * 128);
                tap[pp][t][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0));
                tap[pp][t][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : OOB, 0, 0));
                wgt[pp][t] = wx * (t ? fy : 1.f - fy);
              }
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                const float g = wgt[pp][t];
                const f32x4 u = tap[pp][t][0], v = tap[pp][t][1];
                acc[0] = __builtin_fmaf(u.x, g, acc[0]); acc[1] = __builtin_fmaf(u.y, g, acc[1]); acc[2] = __builtin_fmaf(u.z, g, acc[2]); acc[3] = __builtin_fmaf(u.w, g, acc[3]);
                acc[4] = __builtin_fmaf(v.x, g, acc[4]); acc[5] = __builtin_fmaf(v.y, g, acc[5]); acc[6] = __builtin_fmaf(v.z, g, acc[6]); acc[7] = __builtin_fmaf(v.w, g, acc[7]);
              }
          }
        }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], 4);
      if (cx == 0) {
        float* dst = reinterpret_cast<float*>(sP + ql * MR_PP) + m * 32 + oct * 8;
        *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
      }
    }

    // ---- level 0 gather: head by head, four corners x four points = 16 eight-byte loads per head, scalar offsets and weights
    {
      // (plain global loads, scalar frame base + 32-bit offset: the clamped window never leaves the level, so no range check is
      //  needed -- and hipcc 7.2 lowers __builtin_amdgcn_raw_buffer_load_b64 to a ONE-dword load whose value it then uses for both
      //  halves: tools/probes/raw_f32_debug.py showed every odd channel equal to its even neighbour)
      const int64_t frame = (int64_t)H0 * W0 * p.ld0;
      const unsigned char* fb = reinterpret_cast<const unsigned char*>(x0 + (int64_t)b * frame);
      const uint32_t voff = (uint32_t)lane * 8u;
      float* gdst = reinterpret_cast<float*>(sG + ql * MRF_GP) + lane * 2;
      float2 tp[2][4][4];
      auto issue = [&](auto hc, float2 (&t)[4][4]) {
        constexpr int h = decltype(hc)::value;
#pragma unroll
        for (int pnt = 0; pnt < 4; ++pnt) {
          const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)base0[pnt], h * 8);
          const unsigned char* wb = fb + sb;                            // wave-uniform: the window's first pixel
          t[pnt][0] = *reinterpret_cast<const float2*>(wb + voff);
          t[pnt][1] = *reinterpret_cast<const float2*>(wb + pix_pitch + voff);
          t[pnt][2] = *reinterpret_cast<const float2*>(wb + row_pitch + voff);
          t[pnt][3] = *reinterpret_cast<const float2*>(wb + row_pitch + pix_pitch + voff);
        }
      };
      auto consume = [&](auto hc, const float2 (&t)[4][4]) {
        constexpr int h = decltype(hc)::value;
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int pnt = 0; pnt < 4; ++pnt) {
          const float ws[4] = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w00[pnt]), h * 8)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w01[pnt]), h * 8)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w10[pnt]), h * 8)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w11[pnt]), h * 8))};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            g0 = __builtin_fmaf(t[pnt][c].x, ws[c], g0);
            g1 = __builtin_fmaf(t[pnt][c].y, ws[c], g1);
          }
        }
        *reinterpret_cast<float2*>(gdst + h * 128) = float2{g0, g1};          // channels 2*lane, 2*lane + 1 of head h's gathered vector
      };
      issue(std::integral_constant<int, 0>{}, tp[0]);
      [&]<int... Hs>(std::integer_sequence<int, Hs...>) {
        (([&] {
           if constexpr (Hs + 1 < 8) issue(std::integral_constant<int, Hs + 1>{}, tp[(Hs + 1) & 1]);
           __builtin_amdgcn_sched_barrier(0);                       // keep the next head's requests ahead of this head's sums
           consume(std::integral_constant<int, Hs>{}, tp[Hs & 1]);
         }()), ...);
      }(std::make_integer_sequence<int, 8>{});
    }
  }
  __syncthreads();

  // ---- projection on the exact fp32 matrix instruction: wave w owns heads 2w, 2w+1.  A = W_h rows (out channel t*16 + r), B = g of the
  // queries (column r = query r & 7).  A lane holds k = pn*16 + q4*4 .. +3 of its row for BOTH operands; step e of a panel consumes
  // k = q4*4 + e of every lane quad, so the four steps cover the panel's 16 k exactly once (gemm.hip: mma_panel<float>).
  const int r = lane & 15, q4 = lane >> 4;
  const float* wc = static_cast<const float*>(p.wc);
  float* out = static_cast<float*>(p.out);
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int h = wave * 2 + hh;
    f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int pn = 0; pn < 8; ++pn) {
      const f32x4 gb = *reinterpret_cast<const f32x4*>(sG + (r & 7) * MRF_GP + h * 512 + (pn * 16 + q4 * 4) * 4);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 wa = *reinterpret_cast<const f32x4*>(wc + (h * 32 + t * 16 + r) * 128 + pn * 16 + q4 * 4);
        acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, gb.x, acc2[t], 0, 0, 0);
        acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, gb.y, acc2[t], 0, 0, 0);
        acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, gb.z, acc2[t], 0, 0, 0);
        acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, gb.w, acc2[t], 0, 0, 0);
      }
    }
    const float s = sS[(r & 7) * 8 + h];
    const int row = sRow[r & 7];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ch = h * 32 + t * 16 + q4 * 4;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bc + ch);
      const f32x4 part = *reinterpret_cast<const f32x4*>(sP + (r & 7) * MR_PP + ch * 4);
      const f32x4 v = acc2[t] + bias * s + part;
      if (r < MRF_QB && row0 + r < p.nrows) *reinterpret_cast<f32x4*>(out + (int64_t)row * p.ldo + ch) = v;
    }
  }
}

}  // namespace moy

using namespace moy;

extern "C" int moy_msda_raw0(const moy_msda_raw_args* a, void* stream) {
  if (!a || !a->x0 || !a->wc || !a->bc || !a->offaw || !a->ref || !a->out || !a->shapes_hw) return MOY_EINVAL;
  const bool is32 = a->dtype == MOY_F32 || a->dtype == MOY_F32X3;              // fp32 tensors (round 6; MOY_F32X3: the same kernel, exact fp32 arithmetic)
  if (a->dtype != MOY_BF16 && a->dtype != MOY_F16 && !is32) return MOY_EINVAL;
  const int esz = is32 ? 4 : 2;
  if (a->B <= 0 || a->Lq <= 0 || a->L < 1 || a->L > 4) return MOY_EINVAL;
  if (a->L > 1 && (!a->planes || a->S1 <= 0)) return MOY_EINVAL;
  MsdaRawParams p{};
  int s = 0;
  for (int l = 0; l < a->L; ++l) {
    p.lv.H[l] = a->shapes_hw[2 * l]; p.lv.W[l] = a->shapes_hw[2 * l + 1];
    if (p.lv.H[l] <= 0 || p.lv.W[l] <= 0) return MOY_EINVAL;
    if (l >= 1) { p.lv.start[l] = s; s += p.lv.H[l] * p.lv.W[l]; }
  }
  if (a->L > 1 && s != a->S1) return MOY_EINVAL;
  if (p.lv.H[0] < 2 || p.lv.W[0] < 2) return MOY_ENOSYS;                        // the clamped 2x2 window needs two rows and two columns
  if (a->ld0 < 128 || (a->ld0 % 2) || (reinterpret_cast<uintptr_t>(a->x0) % (is32 ? 8 : 4))) return MOY_EINVAL;
  if ((int64_t)p.lv.H[0] * p.lv.W[0] * a->ld0 * esz > 0x7fffffffLL) return MOY_ENOSYS;      // one frame of level 0 per buffer descriptor
  // head planes: 8 planes of head_stride elements, each holding B * S1 tokens of 32 channels (a stride shorter than that
  // would let a frame's taps read another head's or frame's tokens without any error)
  if (a->L > 1 && (a->head_stride < (int64_t)a->B * a->S1 * 32 || (a->head_stride % 8) || a->head_stride * 8 * esz > 0x7fffffffLL || !aligned16(a->planes)))
    return MOY_EINVAL;
  if (a->ld_oa < 8 * a->L * 4 * 3 || (a->ld_oa % 4) || a->ldo < 256 || (a->ldo % 4) || !aligned16(a->ref) || !aligned16(a->offaw)) return MOY_EINVAL;
  if (!aligned16(a->wc) || !aligned16(a->bc) || (reinterpret_cast<uintptr_t>(a->out) % (is32 ? 16 : 8))) return MOY_EINVAL;
  if (is32 && a->wc_packed) return MOY_EINVAL;                                   // the fp32 form reads wc row-major
  p.wc_packed = a->wc_packed;
  p.x0 = a->x0; p.ld0 = a->ld0; p.wc = a->wc; p.bc = a->bc; p.planes = a->planes; p.head_stride = a->head_stride; p.S1 = a->S1;
  p.L = a->L; p.offaw = a->offaw; p.ld_oa = a->ld_oa; p.ref = a->ref; p.Lq = a->Lq; p.nrows = a->B * a->Lq; p.out = a->out; p.ldo = a->ldo;
  p.B = a->B; p.perm = a->perm;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (is32) {
    p.ngroups = (p.nrows + MRF_QB - 1) / MRF_QB;
    hipLaunchKernelGGL(msda_raw_f32_kernel, dim3((p.ngroups + 7) / 8 * 8), dim3(256), 0, st, p);
    return launch_status();
  }
  p.ngroups = (p.nrows + MR_QB - 1) / MR_QB;
  const int nblk = (p.ngroups + 7) / 8 * 8;          // a multiple of the 8 XCDs: XCD x walks the x-th eighth of the groups
  // the tap sums on the matrix cores: three levels, every one with a 2 x 2 window inside it (MOY_MR_MFMA=0: the vector-ALU form)
  static const int mfma = knob("MOY_MR_MFMA", 1);
  bool small = false;
  for (int l = 0; l < a->L; ++l) small |= p.lv.H[l] < 2 || p.lv.W[l] < 2;
  if (mfma && a->L == 3 && !small) {
    // level 0 by 16-byte loads needs 16-byte aligned pixels (a channel slice of a wider tensor at an odd 8-channel offset keeps dword loads)
    const bool x4ok = (a->ld0 % 8) == 0 && aligned16(a->x0);
#if MOY_DIAG
    static const int x4 = knob("MOY_MRM_X4", 1);                        // A/B: 0 = level 0 by dword loads (the round-5 form)
    if (!x4 || !x4ok) {
#else
    if (!x4ok) {
#endif
      if (a->dtype == MOY_BF16) hipLaunchKernelGGL((msda_raw_mfma_kernel<bf16_t, false>), dim3(nblk), dim3(64 * MRM_NW), 0, st, p);
      else hipLaunchKernelGGL((msda_raw_mfma_kernel<f16_t, false>), dim3(nblk), dim3(64 * MRM_NW), 0, st, p);
      return launch_status();
    }
    if (a->dtype == MOY_BF16) hipLaunchKernelGGL((msda_raw_mfma_kernel<bf16_t, true>), dim3(nblk), dim3(64 * MRM_NW), 0, st, p);
    else hipLaunchKernelGGL((msda_raw_mfma_kernel<f16_t, true>), dim3(nblk), dim3(64 * MRM_NW), 0, st, p);
    return launch_status();
  }
  if (a->dtype == MOY_BF16) hipLaunchKernelGGL((msda_raw_kernel<bf16_t>), dim3(nblk), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((msda_raw_kernel<f16_t>), dim3(nblk), dim3(256), 0, st, p);
  return launch_status();
}
