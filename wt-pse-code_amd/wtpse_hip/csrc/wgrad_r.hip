// 3x3 convolution weight gradient in the x3 arithmetic (conv_x3.hip) with REGISTER-RESIDENT operands: no LDS image, no barrier
// and no LDS store in the main loop.  Replaces autograd's weight gradient of the nn.Conv2d dispatches of the reference hot path
// (algorithms.py:882-888,926-933,404-424; shape_networks.py:182-239) on every map that is a multiple of 32 pixels wide.
//
//   dW[co][ci][ky][kx] = sum_{b,y,x} dY[b,co,y,x] * A(X)[b,ci,y+ky-1,x+kx-1]      A = the fused BatchNorm-apply / ReLU on load
//
// GEMM with K = pixels on v_mfma_f32_16x16x32_bf16: a k-step is 32 consecutive pixels of one image row.  The matrix instruction
// wants, per lane, 8 consecutive k of one row (A: a dY channel) or column (B: an X channel) — and in NCHW 8 consecutive pixels of
// a channel are 32 contiguous bytes.  So every lane loads its own fragments straight from global memory (two 16-byte loads),
// applies the prologue, splits the 8 values into three bf16 terms (x = x0 + x1 + x2, round-to-nearest-even each, as conv_x3.hip)
// and feeds the matrix cores from registers:
//   * lane l = (c16 = l & 15, g = l >> 4): channel c16 of the fragment, pixels x0 + 8g .. x0 + 8g + 7 of the row;
//   * the horizontal taps kx = 0 / 2 need the same row shifted by one pixel: the packed bf16 pairs are re-paired with
//     v_alignbit and the pixel that crosses an 8-pixel group comes from the neighbouring lane group (ds_bpermute, lane +-16);
//     the pixel beyond the 32-pixel strip is loaded by the lane groups at the strip's ends (one dword per channel and row);
//   * the vertical taps pair the X row r with the dY rows r+1, r, r-1: the split dY rows are kept in a 3-slot ring of registers
//     (rows unrolled by three, so that the slots are static).
// A wave owns a [16 MF couts] x [16 NF cins] x 9 taps accumulator set (MF x NF x 9 tiles of 16x16) over its share of the pixels;
// the four waves of a workgroup work on different pixels of the same block and meet in LDS at the very end; per-workgroup
// slabs are folded in fp64 in fixed order by wgrad_reduce_k / wgrad_fold4_k (conv.hip): bitwise reproducible, no atomics.
// Bias gradient (layers without BatchNorm): sum of the dY values the A fragments are made from, in the waves of cin block 0.
#include <type_traits>
#include <utility>
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

struct WgradRArgs {
  const float* dy;
  const float* x0;
  const float* x1;
  const float* pro0;
  const float* pro1;
  // AFF: dy is not materialised — it is the second half of a BatchNorm backward, dy = k1[c] * g + k2[c] * y + k3[c] with
  // g = `dy` (the masked incoming gradient), y = `bn_y` (the layer's raw conv output), coef = `bn_coef` [Cout][3]
  const float* bn_y;
  const float* bn_coef;
  float* slab;     // [nslab][Cout][Cin][9]
  float* slab_b;   // [nslab_b][Cout] or null
  // TERMS 2 (two fp16 terms per operand, conv_x3_kernels.h): dY is multiplied by the power of two that brings its largest magnitude
  // (dy_amax: its amax table, common.h, from its producer / wtpse_amax; null: X3_FWD_SCALE) into [2^14, 2^15), X — as loaded, after the
  // prologue — likewise from the bound in x_amax0 / x_amax1 (the tables of x0 / x1: conv_x3_kernels.h; the larger counts; neither:
  // X3_FWD_SCALE); the slabs are scaled back as they are written
  const unsigned* dy_amax;
  const unsigned* x_amax0;
  const unsigned* x_amax1;
  int B, H, W, C0, C1, Cin, Cout;
  int pro_relu;
  int strips;      // W / 32
  int nseg, rseg;  // row segments per (image, strip) and their height
  int units;       // B * strips * nseg
  int wpp;         // waves per (cout block, cin block) pair (multiple of 4)
  int nci;         // cin blocks
  // twin: W == 16 — a 32-pixel k-step is row y of TWO consecutive images side by side (lane groups 0, 1: image b, columns 0-7 / 8-15;
  // groups 2, 3: image b + 1): the same rows, vertical taps and strip-end logic (both strip ends are image edges: zero), and the pixel
  // that crosses from group 1 to group 2 (the seam between the images) is zero as well.  strips = 1, units over image PAIRS.
  int twin;
};

template <int I> using IC = std::integral_constant<int, I>;
template <class F, int... Is> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ unsigned wr_pack_rne(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// (a, b) -> three dwords holding the bf16 pairs (term_i(a), term_i(b)): see split3_pair in conv_x3.hip
// (the compiler fuses the two subtractions of a level into one v_pk_add_f32; keeping them apart — two v_sub_f32 — measured 7 %
// slower over the layer set: the row step is bound by the NUMBER of vector instructions that fit behind its MFMAs)
__device__ __forceinline__ void wr_split3_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = wr_pack_rne(a, b);
  const float ra = a - __builtin_bit_cast(float, p0 << 16);
  const float rb = b - __builtin_bit_cast(float, p0 & 0xFFFF0000u);
  p1 = wr_pack_rne(ra, rb);
  const float sa = ra - __builtin_bit_cast(float, p1 << 16);
  const float sb = rb - __builtin_bit_cast(float, p1 & 0xFFFF0000u);
  p2 = wr_pack_rne(sa, sb);
}

template <int TERMS>
__device__ __forceinline__ f32x4 mfma16x32(u32x4v a, u32x4v b, f32x4 c) {
  if constexpr (TERMS == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// 8 fp32 values -> three fragments (4 dwords each) of bf16 terms
[[maybe_unused]] __device__ __forceinline__ void wr_split8(const f32x4& lo, const f32x4& hi, u32x4v (&t)[3]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    unsigned q0, q1, q2;
    wr_split3_pair(lo[2 * j], lo[2 * j + 1], q0, q1, q2);
    t[0][j] = q0; t[1][j] = q1; t[2][j] = q2;
    wr_split3_pair(hi[2 * j], hi[2 * j + 1], q0, q1, q2);
    t[0][2 + j] = q0; t[1][2 + j] = q1; t[2][2 + j] = q2;
  }
}

// waves per workgroup: the 16 x 16 blocks (under 256 registers, two waves per SIMD) run as ONE 8-wave workgroup per CU whose
// waves walk down eight adjacent strips in step — a whole 256-pixel row of every channel between them; the larger blocks have
// their SIMD to themselves: four waves
template <int MF, int NF> struct WgradRGeom { static constexpr int NW = (MF * NF >= 2) ? 4 : 8; };

// TERMS: 16-bit terms per fp32 operand — 3 (x3: three bf16 terms, six products per multiply), 2 (x2h: two fp16 terms, three products,
// power-of-two operand scaling; the host launches PRO = true, the scale rides in the prologue coefficients) or 1 (bf16 mode) — wtpse_x3_terms
// TWIN: W == 16, two images per 32-pixel step (WgradRArgs::twin) — a template flag: the seam masks and per-lane image offsets cost the
// row step ~10 % when they were run-time selects in every instantiation (the step is bound by its vector-instruction count)
template <int MF, int NF, bool PRO, bool BIAS, bool AFF = false, int TERMS = 3, bool TWIN = false>
__global__ __launch_bounds__((64 * WgradRGeom<MF, NF>::NW), 1) void wgrad_r_k(WgradRArgs a) {
  static_assert(TERMS >= 1 && TERMS <= 3, "three bf16 terms, two fp16 terms or one bf16 term");
  static_assert(TERMS != 2 || (PRO && !AFF), "x2h: the X scale rides in the prologue; the fused-BatchNorm form has no amax of the dY it forms");
  constexpr int NT = 9, NW = WgradRGeom<MF, NF>::NW;
  // (a, b) -> TERMS dwords of packed bf16 pairs
  auto split_pair = [](float v0, float v1, unsigned (&q)[TERMS]) __attribute__((always_inline)) {
    if constexpr (TERMS == 3) wr_split3_pair(v0, v1, q[0], q[1], q[2]);
    else if constexpr (TERMS == 2) split2h_pair(v0, v1, q[0], q[1]);
    else q[0] = wr_pack_rne(v0, v1);
  };
  // (the wave id is wave-uniform, but only readfirstlane tells the compiler: everything derived from it — the unit, the image,
  // the buffer descriptors — then stays in SGPRs instead of being re-derived per lane behind waterfall loops)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g = lane >> 4;
  const int wgpp = a.wpp / NW;                           // workgroups per pair
  const int pair = blockIdx.x / wgpp, wg = blockIdx.x - pair * wgpp;
  const int cin0 = (pair % a.nci) * (16 * NF), cout0 = (pair / a.nci) * (16 * MF);
  const int wv = wg * NW + wave;
  const int u0 = (int)((long long)wv * a.units / a.wpp), u1 = (int)((long long)(wv + 1) * a.units / a.wpp);
  const int HW = a.H * a.W;
  const unsigned W4 = (unsigned)a.W * 4u;
  float sx = 1.f;
  if constexpr (TERMS == 2) {
    sx = X3_FWD_SCALE;
    if (a.x_amax0 || a.x_amax1) {
      unsigned m = a.x_amax0 ? amax_read(a.x_amax0) : 0u;
      if (a.x_amax1) m = max(m, amax_read(a.x_amax1));
      sx = x3_scale_from_amax(m);
    }
  }
  const float sdy = TERMS == 2 ? (a.dy_amax ? x3_scale_from_amax(amax_read(a.dy_amax)) : X3_FWD_SCALE) : 1.f;

  f32x4 acc[MF][NF][NT];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][t][r] = 0.f;
  // bias gradient: per-lane sums in fp64 (a lane adds up hundreds of dY values whose total nearly cancels — the gradient of a bias
  // behind the WT loss — and an fp32 running sum put this tensor 3.3x as far from the fp64 oracle as the reference's own fp32 run at
  // the benchmark's geometry, tests/test_parity_gpu.py [32-10-256]; two v_add_f64 per pixel pair in the HBM-bound 16 x 16 blocks)
  double bsum[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) bsum[m] = 0.0;

  // per cin fragment: which input tensor it reads (a 16-channel fragment never straddles the two halves of a concat),
  // its channel within that tensor, the prologue coefficients of this lane's channel
  bool xfirst[NF];
  int xch[NF];
  float psc[NF], psh[NF], plo[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) {
    const int c = cin0 + 16 * n;                          // wave-uniform
    xfirst[n] = c < a.C0 || a.x1 == nullptr;
    xch[n] = (xfirst[n] ? c : c - a.C0) + c16;
    psc[n] = 1.f; psh[n] = 0.f; plo[n] = -INFINITY;
    if (PRO) {
      const float* pro = xfirst[n] ? a.pro0 : a.pro1;
      if (pro) { psc[n] = pro[2 * xch[n]]; psh[n] = pro[2 * xch[n] + 1]; }
      if (a.pro_relu & (xfirst[n] ? 1 : 2)) plo[n] = 0.f;
      if (TERMS == 2) { psc[n] *= sx; psh[n] *= sx; }     // (exact: a power of two commutes with the rounding of the fma and with the ReLU)
    }
  }
  float ak1[MF], ak2[MF], ak3[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    ak1[m] = 1.f; ak2[m] = 0.f; ak3[m] = 0.f;
    if (AFF) {
      const float* q = a.bn_coef + 3 * (cout0 + 16 * m + c16);
      ak1[m] = q[0]; ak2[m] = q[1]; ak3[m] = q[2];
    }
  }
  const int src_up = (lane + 16) & 63, src_dn = (lane - 16) & 63;
  // twin: nothing crosses the seam between the two images — lane group 2 hands no pixel 0 down to group 1, group 1 no pixel 7 up to group 2
  const unsigned seam_r = (TWIN && g == 2) ? 0u : ~0u, seam_l = (TWIN && g == 1) ? 0u : ~0u;

  for (int u = u0; u < u1; ++u) {
    // unit -> (image, row segment, 32-pixel strip), strips fastest: the four waves of a workgroup then walk down four adjacent
    // strips in step, i.e. read 512 contiguous bytes of every channel row between them (DRAM pages, L2 lines)
    const int strip = u % a.strips;
    const int bs = u / a.strips;
    const int seg = bs % a.nseg, b = (bs / a.nseg) << (TWIN ? 1 : 0);
    const unsigned nimg = (TWIN && b + 1 < a.B) ? 2u : 1u;      // images behind the descriptors (an odd batch ends on a single one)
    const bool lane_live = (unsigned)(TWIN ? g >> 1 : 0) < nimg;     // false: this lane's image is the odd batch's missing twin
    const int x0 = strip * 32;
    const int y0 = seg * a.rseg, y1 = min(a.H, y0 + a.rseg);
    const int rfirst = max(y0 - 1, 0), rlast = min(y1, a.H - 1);
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(a.dy + (size_t)b * a.Cout * HW, nimg * (unsigned)a.Cout * HW * 4u);
    const __amdgpu_buffer_rsrc_t rsb = AFF ? make_rsrc(a.bn_y + (size_t)b * a.Cout * HW, nimg * (unsigned)a.Cout * HW * 4u) : rsy;
    const __amdgpu_buffer_rsrc_t rsx0 = make_rsrc(a.x0 + (size_t)b * a.C0 * HW, nimg * (unsigned)a.C0 * HW * 4u);
    const __amdgpu_buffer_rsrc_t rsx1 = a.x1 ? make_rsrc(a.x1 + (size_t)b * a.C1 * HW, nimg * (unsigned)a.C1 * HW * 4u) : rsx0;
    // byte offsets of this lane's 8 pixels in row 0
    unsigned ybase[MF], xbase[NF], ebase[NF];
    bool evalid;
    {
      // edge pixel of this lane: lane group 0 fetches the pixel right of the strip (x0 + 32), lane group 3 the one left of it
      const int ex = g == 0 ? x0 + 32 : x0 - 1;
      evalid = !TWIN && (g == 0 || g == 3) && ex >= 0 && ex < a.W;
      const unsigned col = TWIN ? 8u * (g & 1) : (unsigned)(x0 + 8 * g);      // this lane's first pixel in its row
      const unsigned img = TWIN ? (unsigned)(g >> 1) : 0u;                      // ... and its image behind the descriptor
      // (an image of the odd batch's missing twin lies beyond num_records: its lanes read zeros)
#pragma unroll
      for (int m = 0; m < MF; ++m) ybase[m] = ((unsigned)(cout0 + 16 * m + c16) * HW + col + img * (unsigned)a.Cout * HW) * 4u;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        xbase[n] = ((unsigned)xch[n] * HW + col + img * (unsigned)(xfirst[n] ? a.C0 : a.C1) * HW) * 4u;
        ebase[n] = evalid ? ((unsigned)xch[n] * HW + ex) * 4u : BUF_OOB;
      }
    }

    // ---- software pipeline over "stages": stage s = (X row s, dY row s + 1).  While the matrix cores work on X row r, the
    // stage r + 1 that has landed in registers is converted (prologue, bf16 split) for the next step, and the loads of stage
    // r + 3 are issued into the raw slot it frees: two stages (8 KB per wave at 16 x 16 channels) are in flight at any time.
    // All ring slots are static: the row loop is unrolled by four.  Rows outside [y0, y1) / the image read as zero
    // (out-of-range buffer offsets), so the products need no edge cases.
    // The fused-BatchNorm form of the 32 x 32 blocks (AFF: two tensors behind the A operand) has no registers for two stages of
    // them: its A side keeps ONE stage in flight, re-issued as soon as the stage's dY pieces are converted (AD = 1).
    constexpr int AD = (AFF && MF * NF >= 4) ? 1 : 2;
    f32x4 rawx[2][NF][2], rawy[AD][MF][2], rawb[AD][AFF ? MF : 1][2];
    float ak3v[AD][MF];           // AFF: k3 of the stage's dY row, 0 for a row outside [y0, y1)
    float rawe[2][NF];
    u32x4v ay[4][MF][TERMS];      // split dY rows: row y in slot (y - rfirst) & 3      [slot][cout fragment][term]
    u32x4v xb[2][NF][TERMS];      // split X row s in slot (s - rfirst) & 1
    unsigned eq[2][NF][TERMS];    // its edge pixel (lane groups 0 and 3)
    // (branch-free: a row outside its range turns into an out-of-range buffer offset by OR-ing the top bit in — every valid
    // offset is below 2^31 — and the row's byte offset rides in the scalar offset operand)
    auto issue_x = [&](auto Sc, int s) {                   // X row s -> raw slot S
      constexpr int S = decltype(Sc)::value;
      const int xv = (int)(s <= rlast);
      const unsigned xo = xv ? 0u : BUF_OOB;
      const unsigned ro = xv ? (unsigned)s * W4 : 0u;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const __amdgpu_buffer_rsrc_t rs = xfirst[n] ? rsx0 : rsx1;
        const unsigned vo = xbase[n] | xo;
        rawx[S][n][0] = buf_load4(rs, vo, ro);
        rawx[S][n][1] = buf_load4(rs, vo + 16u, ro);
        rawe[S][n] = buf_load(rs, ebase[n] | xo, ro);
      }
    };
    auto issue_a = [&](auto Sc, int s) {                   // dY row s + 1 -> raw slot S % AD
      constexpr int S = decltype(Sc)::value % AD;
      const int y = s + 1;
      const int yv = (int)(y >= y0) & (int)(y < y1);      // (bitwise on purpose: no short-circuit branch inside the row loop)
      const unsigned yo = yv ? (unsigned)y * W4 : 0u, ym = yv ? 0u : BUF_OOB;
      if (AFF) {
#pragma unroll
        for (int m = 0; m < MF; ++m) ak3v[S][m] = (yv && lane_live) ? ak3[m] : 0.f;
      }
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const unsigned vo = ybase[m] | ym;
        rawy[S][m][0] = buf_load4(rsy, vo, yo);
        rawy[S][m][1] = buf_load4(rsy, vo + 16u, yo);
        if (AFF) {
          rawb[S][m][0] = buf_load4(rsb, vo, yo);
          rawb[S][m][1] = buf_load4(rsb, vo + 16u, yo);
        }
      }
    };
    // Conversion of stage s in raw slot S (-> X slot S, dY slot AS), cut into NP pieces of about half a dozen vector instructions:
    //   dY pair      one pair of a dY fragment's 8 pixels: [BatchNorm-apply,] split into the 16-bit terms (+ bias sum)
    //   X pair       one pair of an X fragment's 8 pixels: prologue, split;      X edge: the edge pixel of an X fragment
    //   halo         the neighbour exchange of the converted X row (two ds_bpermute per fragment and term) for the NEXT step — issued
    //                here, a dozen chains in front of their use (round 6: issued where they were used, every 54 MFMAs the wave sat out
    //                one to two LDS round trips behind `s_waitcnt lgkmcnt(0)`)
    //   loads        of stage s + 2 (X; and A when AD = 2);   AD = 1 only: "A loads" of stage s + 1 (the slot was just consumed)
    // AD = 2 (order of arrival: a stage's X loads are issued in front of its dY loads, and vmcnt counts in order):
    //   X pairs, X edges, half of the dY pairs, halo, the other dY pairs, loads
    // AD = 1:  dY pairs, A loads, X pairs, X edges, halo, loads
    // The halo piece must sit behind the step's LAST build of shifted rows (fragment NF - 1: chain 9 MF (NF - 1)), whose operands it replaces.
    constexpr int NPA = 4 * MF, NPX = 4 * NF, NP = NPA + NPX + NF + 2 + (AD == 1 ? 1 : 0);
    constexpr int P_HALO = AD == 2 ? NPX + NF + NPA / 2 : NP - 2;
    unsigned nbr[NF][TERMS], nbl[NF][TERMS];      // halo of the X row the next step multiplies: pixel 8 / pixel -1 of this lane's group
    auto cvt_a = [&](auto Qc, auto Sc, auto ASc) {              // dY pair Q = (fragment, pair) of raw slot S -> dY slot AS
      constexpr int Q = decltype(Qc)::value, SA = decltype(Sc)::value % AD, AS = decltype(ASc)::value;
      constexpr int m = Q / 4, q = Q % 4;
      float v0 = rawy[SA][m][q >> 1][2 * (q & 1)], v1 = rawy[SA][m][q >> 1][2 * (q & 1) + 1];
      if (AFF) {   // rows outside [y0, y1) load g = y = 0 and would come out as k3: the row validity rides in ak3v
        v0 = fmaf(ak1[m], v0, fmaf(ak2[m], rawb[SA][m][q >> 1][2 * (q & 1)], ak3v[SA][m]));
        v1 = fmaf(ak1[m], v1, fmaf(ak2[m], rawb[SA][m][q >> 1][2 * (q & 1) + 1], ak3v[SA][m]));
      }
      if (BIAS) bsum[m] += (double)v0 + (double)v1;
      if (TERMS == 2) { v0 *= sdy; v1 *= sdy; }
      unsigned qq[TERMS];
      split_pair(v0, v1, qq);
#pragma unroll
      for (int t = 0; t < TERMS; ++t) ay[AS][m][t][q] = qq[t];
    };
    auto cvt_x = [&](auto Qc, auto Sc) {                        // X pair Q of raw slot S -> X slot S
      constexpr int Q = decltype(Qc)::value, S = decltype(Sc)::value;
      constexpr int n = Q / 4, q = Q % 4;
      float v0 = rawx[S][n][q >> 1][2 * (q & 1)], v1 = rawx[S][n][q >> 1][2 * (q & 1) + 1];
      if (PRO) {
        v0 = fmaxf(fmaf(v0, psc[n], psh[n]), plo[n]);
        v1 = fmaxf(fmaf(v1, psc[n], psh[n]), plo[n]);
        // an odd batch's missing twin image reads zeros, which the prologue turns into act(shift): harmless while its dY is zero
        // too, but a non-finite shift would make 0 * inf of it — the dead lanes' X is zero, explicitly
        if (TWIN && !lane_live) v0 = v1 = 0.f;
      }
      unsigned qq[TERMS];
      split_pair(v0, v1, qq);
#pragma unroll
      for (int t = 0; t < TERMS; ++t) xb[S][n][t][q] = qq[t];
    };
    auto cvt_e = [&](auto Nc, auto Sc) {                        // edge pixel of X fragment n
      constexpr int n = decltype(Nc)::value, S = decltype(Sc)::value;
      float e = rawe[S][n];
      if (PRO) e = evalid ? fmaxf(fmaf(e, psc[n], psh[n]), plo[n]) : 0.f;     // zero padding applies after the activation
      unsigned qq[TERMS];
      split_pair(e, 0.f, qq);
#pragma unroll
      for (int t = 0; t < TERMS; ++t) eq[S][n][t] = qq[t];
    };
    auto halo = [&](auto Sc) {                                  // neighbour exchange of the converted X row in slot S
      constexpr int S = decltype(Sc)::value;
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int t = 0; t < TERMS; ++t) {
          const u32x4v d = xb[S][n][t];
          // pixel 8 of this lane's group = pixel 0 of the next group (lane + 16); for the last group the strip's right edge,
          // which lane group 0 holds.  Pixel -1 = pixel 7 of the previous group (lane - 16); for group 0 the left edge (group 3).
          const unsigned sup_r = (g == 0 ? eq[S][n][t] : d[0]) & seam_r;
          const unsigned sup_l = (g == 3 ? (eq[S][n][t] << 16) : d[3]) & seam_l;
          nbr[n][t] = (unsigned)__builtin_amdgcn_ds_bpermute(src_up * 4, (int)sup_r);
          nbl[n][t] = (unsigned)__builtin_amdgcn_ds_bpermute(src_dn * 4, (int)sup_l);
        }
    };
    auto piece = [&](auto Pc, auto Sc, auto ASc, int s_next) {   // s_next: the stage being converted + 1
      constexpr int P = decltype(Pc)::value;
      if constexpr (P == NP - 1) {
        issue_x(Sc, s_next + 1);
        if (AD == 2) issue_a(Sc, s_next + 1);
      } else if constexpr (P == P_HALO) {
        halo(Sc);
      } else if constexpr (AD == 2) {
        if constexpr (P < NPX) cvt_x(IC<P>{}, Sc);
        else if constexpr (P < NPX + NF) cvt_e(IC<P - NPX>{}, Sc);
        else if constexpr (P < P_HALO) cvt_a(IC<P - NPX - NF>{}, Sc, ASc);
        else cvt_a(IC<P - NPX - NF - 1>{}, Sc, ASc);
      } else {
        if constexpr (P < NPA) cvt_a(IC<P>{}, Sc, ASc);
        else if constexpr (P == NPA) issue_a(Sc, s_next);
        else if constexpr (P < NPA + 1 + NPX) cvt_x(IC<P - NPA - 1>{}, Sc);
        else cvt_e(IC<P - NPA - 1 - NPX>{}, Sc);
      }
    };
    auto convert_a = [&](auto Sc, auto ASc) { static_for<NPA>([&](auto q) { cvt_a(q, Sc, ASc); }); };
    auto convert_x = [&](auto Sc) {
      static_for<NPX>([&](auto q) { cvt_x(q, Sc); });
      static_for<NF>([&](auto n) { cvt_e(n, Sc); });
    };
    // One X row r (k = r - rfirst, J = k & 3): its products with the dY rows r + 1, r, r - 1 — chains of six dependent MFMAs, one
    // per (cin fragment, vertical tap, cout fragment, horizontal tap) — with the conversion pieces of stage r + 1 dealt out
    // behind the chains.  A wave of the 32 x 32 blocks has its SIMD to itself and issues in order, so the ORDER of the stream is
    // the overlap: a 16x16x32 MFMA holds the vector issue for 8 of its 16 cycles, two vector instructions fit behind each.
    constexpr int NC = NF * 9 * MF;
    static_assert((9 * MF * (NF - 1) * NP + NC - 1) / NC <= P_HALO, "the halo piece overwrites the operands of the step's last build of shifted rows");
    auto step = [&](auto Jc, int r) {
      constexpr int J = decltype(Jc)::value, XS = J & 1, CS = (J + 1) & 1, AS = (J + 2) & 3;
      static_for<NF>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        u32x4v xs[3][TERMS];      // [kx][term]: the row shifted by kx - 1 pixels
#pragma unroll
        for (int t = 0; t < TERMS; ++t) {
          const u32x4v d = xb[XS][n][t];
          const unsigned nb_r = nbr[n][t], nb_l = nbl[n][t];      // exchanged by the previous step's halo piece (or in front of the loop)
          xs[1][t] = d;
          xs[2][t][0] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
          xs[2][t][1] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
          xs[2][t][2] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
          xs[2][t][3] = __builtin_amdgcn_alignbit(nb_r, d[3], 16);
          xs[0][t][0] = __builtin_amdgcn_alignbit(d[0], nb_l, 16);
          xs[0][t][1] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
          xs[0][t][2] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
          xs[0][t][3] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<9 * MF>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          constexpr int ky = c / (3 * MF), m = (c / 3) % MF, kx = c % 3;
          constexpr int slot = (J + 1 - ky + 4) & 3;         // dY row r + 1 - ky
          f32x4 v = acc[m][n][ky * 3 + kx];
          if constexpr (TERMS == 3) {      // the six leading cross terms, smallest first (as conv_x3.hip)
            v = mfma16x32<3>(ay[slot][m][0], xs[kx][2], v);
            v = mfma16x32<3>(ay[slot][m][1], xs[kx][1], v);
            v = mfma16x32<3>(ay[slot][m][2], xs[kx][0], v);
          }
          if constexpr (TERMS >= 2) {
            v = mfma16x32<TERMS>(ay[slot][m][0], xs[kx][1], v);
            v = mfma16x32<TERMS>(ay[slot][m][1], xs[kx][0], v);
          }
          v = mfma16x32<TERMS>(ay[slot][m][0], xs[kx][0], v);
          acc[m][n][ky * 3 + kx] = v;
          constexpr int cg = n * 9 * MF + c;                 // chain index within the step
          constexpr int plo_ = (cg * NP + NC - 1) / NC, phi_ = ((cg + 1) * NP + NC - 1) / NC;
          static_for<phi_ - plo_>([&](auto pp) { piece(IC<plo_ + decltype(pp)::value>{}, IC<CS>{}, IC<AS>{}, r + 2); });
#ifndef WGRAD_R_NO_INTERLEAVE
#pragma unroll
          for (int i = 0; i < (TERMS == 3 ? 6 : TERMS == 2 ? 3 : 1); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);      // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);      // 2 VALU
          }
#endif
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    };

#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int t = 0; t < TERMS; ++t) ay[s][m][t] = (u32x4v){0u, 0u, 0u, 0u};
    // stages rfirst - 1 (its dY row only: X row rfirst - 1 lies outside the unit or the image) and rfirst are converted up
    // front; stages rfirst + 1 and rfirst + 2 are in flight when the row loop starts (AD = 1: only rfirst + 1 on the A side)
    if (AD == 2) {
      issue_a(IC<1>{}, rfirst - 1);
      issue_a(IC<0>{}, rfirst);
      issue_x(IC<0>{}, rfirst);
      convert_a(IC<1>{}, IC<0>{});
      convert_a(IC<0>{}, IC<1>{});
      convert_x(IC<0>{});
      halo(IC<0>{});
      issue_a(IC<1>{}, rfirst + 1);
      issue_x(IC<1>{}, rfirst + 1);
      issue_a(IC<0>{}, rfirst + 2);
      issue_x(IC<0>{}, rfirst + 2);
    } else {
      issue_a(IC<0>{}, rfirst - 1);
      issue_x(IC<0>{}, rfirst);
      convert_a(IC<0>{}, IC<0>{});
      issue_a(IC<0>{}, rfirst);
      convert_x(IC<0>{});
      halo(IC<0>{});
      issue_x(IC<1>{}, rfirst + 1);
      convert_a(IC<0>{}, IC<1>{});
      issue_a(IC<0>{}, rfirst + 1);
      issue_x(IC<0>{}, rfirst + 2);
    }
    int r = rfirst;
    for (; r + 3 <= rlast; r += 4) {
      step(std::integral_constant<int, 0>{}, r);
      step(std::integral_constant<int, 1>{}, r + 1);
      step(std::integral_constant<int, 2>{}, r + 2);
      step(std::integral_constant<int, 3>{}, r + 3);
    }
    if (r <= rlast) step(std::integral_constant<int, 0>{}, r);
    if (r + 1 <= rlast) step(std::integral_constant<int, 1>{}, r + 1);
    if (r + 2 <= rlast) step(std::integral_constant<int, 2>{}, r + 2);
  }

  // ---- the four waves' partial sums meet in LDS, one vertical tap (3 x MF x NF tiles) at a time
  __shared__ f32x4 red[NW][3 * MF * NF][64];
  float* out = a.slab + (size_t)wg * a.Cout * a.Cin * NT;      // slab `wg`: every pair writes its own (cout, cin) block of it
  const float unscale = TERMS == 2 ? 1.f / (sx * sdy) : 1.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) red[wave][(m * NF + n) * 3 + kx][lane] = acc[m][n][ky * 3 + kx];
    __syncthreads();
    for (int e = tid; e < 3 * MF * NF * 64; e += 64 * NW) {
      const int l = e & 63, f = e >> 6;
      const int kx = f % 3, mn = f / 3, n = mn % NF, m = mn / NF;
      f32x4 s = (red[0][f][l] + red[1][f][l]) + (red[2][f][l] + red[3][f][l]);
      if (NW == 8) s += (red[4][f][l] + red[5][f][l]) + (red[6][f][l] + red[7][f][l]);
      const int ci = cin0 + 16 * n + (l & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = cout0 + 16 * m + 4 * (l >> 4) + i;
        out[((size_t)co * a.Cin + ci) * NT + ky * 3 + kx] = TERMS == 2 ? s[i] * unscale : s[i];
      }
    }
  }
  if (BIAS) {
    if (cin0 == 0) {
      __syncthreads();
      double* redb = reinterpret_cast<double*>(&red[0][0][0]);
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        double v = bsum[m];
        v += __shfl_xor(v, 16, 64);                       // the four pixel groups of a channel
        v += __shfl_xor(v, 32, 64);
        if (g == 0) redb[(wave * MF + m) * 16 + c16] = v;
      }
      __syncthreads();
      if (tid < 16 * MF) {
        const int m = tid >> 4, c = tid & 15;
        double s = (redb[(0 * MF + m) * 16 + c] + redb[(1 * MF + m) * 16 + c]) +
                   (redb[(2 * MF + m) * 16 + c] + redb[(3 * MF + m) * 16 + c]);
        if (NW == 8) s += (redb[(4 * MF + m) * 16 + c] + redb[(5 * MF + m) * 16 + c]) +
                          (redb[(6 * MF + m) * 16 + c] + redb[(7 * MF + m) * 16 + c]);
        a.slab_b[(size_t)wg * a.Cout + cout0 + 16 * m + c] = (float)s;       // the slabs are folded in fp64 (wgrad_reduce_k)
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct WgradRPlan {
  int mf, nf, nw, pairs, nci, wpp, nseg, rseg, units, strips;
};

// W == 16 (the deepest level's maps): two images side by side per 32-pixel k-step (WgradRArgs::twin).  WTPSE_WGRAD_R_TWIN=0: those
// layers stay on the LDS kernel (conv_wgrad_x3_k).
static const bool g_wgrad_r_twin = [] { const char* e = getenv("WTPSE_WGRAD_R_TWIN"); return !(e && e[0] == '0'); }();
static bool wgrad_r_width_ok(int W) { return W % 32 == 0 || (W == 16 && g_wgrad_r_twin); }

static bool wgrad_r_plan(int B, int H, int W, int Cin, int Cout, WgradRPlan& p) {
  if (!wgrad_r_width_ok(W) || H < 1 || Cin % 16 != 0 || Cout % 16 != 0) return false;
  p.mf = Cout % 32 == 0 ? 2 : 1;
  p.nf = Cin % 32 == 0 ? 2 : 1;
  p.nci = Cin / (16 * p.nf);
  p.pairs = (Cout / (16 * p.mf)) * p.nci;
  p.strips = W == 16 ? 1 : W / 32;
  // waves: one per SIMD for the blocks of two or four fragments (more than 256 registers), two per SIMD for the 16 x 16 blocks
  p.nw = (p.mf * p.nf >= 2) ? 4 : 8;
  // tuning override, read once per process (the slab count the host sizes its scratch with must agree with the launch)
  static const int target_override = [] { const char* e = getenv("WTPSE_WGRAD_R_WAVES"); return e ? atoi(e) : 0; }();
  int target = (p.mf * p.nf >= 2) ? 1024 : 2048;
  if (target_override >= 64 && p.mf * p.nf >= 2) target = target_override;
  int wpp = (target / p.pairs) / p.nw * p.nw;
  if (wpp < p.nw) wpp = p.nw;
  const int cols = (W == 16 ? (B + 1) / 2 : B) * p.strips;    // (image [pair], strip) columns of H rows
  int nseg = 1;
  if (cols < wpp) nseg = ceil_div(wpp, cols);
  if (nseg > H) nseg = H;
  p.rseg = ceil_div(H, nseg);
  p.nseg = ceil_div(H, p.rseg);
  p.units = cols * p.nseg;
  if (wpp > p.units) wpp = (p.units + p.nw - 1) / p.nw * p.nw;   // trailing waves get no unit: they contribute zero slabs
  p.wpp = wpp;
  return true;
}

extern "C" int wtpse_wgrad_r_supported(int Cin, int Cout, int ksize, int C0, int W) {
  if (W == 16 && (Cin % 32 != 0 || Cout % 32 != 0)) return 0;      // TWIN is instantiated for the 32 x 32 blocks only (the deepest level)
  return ksize == 3 && Cin % 16 == 0 && Cout % 16 == 0 && C0 % 16 == 0 && wgrad_r_width_ok(W);
}

// slabs of a wtpse_conv_wgrad_r launch: `slab` holds that many [Cout][Cin][9] partial gradients, `dbias_slab` as many [Cout]
extern "C" int wtpse_wgrad_r_slabs(int B, int H, int W, int Cin, int Cout) {
  WgradRPlan p;
  if (!wgrad_r_plan(B, H, W, Cin, Cout, p)) return 0;
  return p.wpp / p.nw;
}

extern int g_x3_terms;      // conv_x3.hip: wtpse_x3_terms()
extern "C" void wtpse_wgrad_reduce_launch2(const float* slab, int ksplit, int n, float* dw, int accumulate, const float* slab_b,
                                           int n_b, float* db, void* stream);

static int wgrad_r_impl(const float* dy, const float* x0, int C0, const float* x1, int C1, const float* pro0,
                        const float* pro1, int pro_relu, float* slab, float* dbias_slab, int nslab, float* dw,
                        float* dbias, int accumulate, int B, int H, int W, int Cout, const float* bn_y, const float* bn_coef,
                        const unsigned* dy_amax, const unsigned* x_amax0, const unsigned* x_amax1, void* stream) {
  WTPSE_REQUIRE(dy && x0 && slab && dw && B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0);
  WTPSE_REQUIRE((C1 == 0) == (x1 == nullptr));
  WTPSE_REQUIRE((dbias == nullptr) == (dbias_slab == nullptr));
  WTPSE_REQUIRE((bn_y == nullptr) == (bn_coef == nullptr));
  const int Cin = C0 + C1;
  WTPSE_REQUIRE(wtpse_wgrad_r_supported(Cin, Cout, 3, C1 ? C0 : 16, W));
  WgradRPlan p;
  WTPSE_REQUIRE(wgrad_r_plan(B, H, W, Cin, Cout, p));
  WTPSE_REQUIRE(nslab == p.wpp / p.nw);
  WTPSE_REQUIRE(!dbias || p.mf * p.nf == 1);               // bias gradient: the 16 -> 16 layers without BatchNorm (DeepWT)
  WTPSE_REQUIRE(!(dbias && bn_y));
  WTPSE_REQUIRE((long long)(C0 > C1 ? C0 : C1) * H * W * 4 < (1ll << 31) && (long long)Cout * H * W * 4 < (1ll << 31));
  WgradRArgs a;
  a.dy = dy; a.x0 = x0; a.x1 = x1; a.pro0 = pro0; a.pro1 = pro1; a.slab = slab; a.slab_b = dbias_slab;
  a.bn_y = bn_y; a.bn_coef = bn_coef; a.dy_amax = dy_amax; a.x_amax0 = x_amax0; a.x_amax1 = x1 ? x_amax1 : nullptr;
  a.B = B; a.H = H; a.W = W; a.C0 = C0; a.C1 = C1; a.Cin = Cin; a.Cout = Cout; a.pro_relu = pro_relu;
  a.strips = p.strips; a.nseg = p.nseg; a.rseg = p.rseg; a.units = p.units; a.wpp = p.wpp; a.nci = p.nci; a.twin = W == 16 ? 1 : 0;
  const bool pro = pro0 != nullptr || pro1 != nullptr || pro_relu != 0;
  const bool bias = dbias != nullptr, aff = bn_y != nullptr;
  dim3 grid((unsigned)(p.pairs * (p.wpp / p.nw)));
  const dim3 blk((unsigned)(64 * p.nw));
  hipStream_t st = (hipStream_t)stream;
#define WR_LAUNCH(M, N) do { \
    if (aff) { if (pro) hipLaunchKernelGGL((wgrad_r_k<M, N, true, false, true>), grid, blk, 0, st, a); \
               else hipLaunchKernelGGL((wgrad_r_k<M, N, false, false, true>), grid, blk, 0, st, a); } \
    else if (pro) hipLaunchKernelGGL((wgrad_r_k<M, N, true, false>), grid, blk, 0, st, a); \
    else hipLaunchKernelGGL((wgrad_r_k<M, N, false, false>), grid, blk, 0, st, a); } while (0)
  // bf16 mode (wtpse_x3_terms(1), conv_x3.hip): one term per operand in the blocks of the MFMA-bound layers; the 16 x 16 blocks (the
  // 16-channel layers, which keep three terms in their forward pass too) and the fused-BatchNorm form stay on three
#define WR_LAUNCH1(M, N) do { \
    if (pro) hipLaunchKernelGGL((wgrad_r_k<M, N, true, false, false, 1>), grid, blk, 0, st, a); \
    else hipLaunchKernelGGL((wgrad_r_k<M, N, false, false, false, 1>), grid, blk, 0, st, a); } while (0)
  // x2h (wtpse_x3_terms(2)): every block shape, PRO always on (the X scale rides in the prologue coefficients); the fused-BatchNorm
  // form (aff) stays on three bf16 terms (nothing knows the largest magnitude of a dY that is never materialised)
#define WR_LAUNCH2(M, N) hipLaunchKernelGGL((wgrad_r_k<M, N, true, false, false, 2>), grid, blk, 0, st, a)
  if (g_x3_terms == 2 && !aff && (dy_amax || p.mf * p.nf >= 2)) {     // (16 x 16 blocks without an amax table: x3, see wtpse_hip.h)
    if (a.twin) {
      WTPSE_REQUIRE(p.mf == 2 && p.nf == 2 && !bias);
      hipLaunchKernelGGL((wgrad_r_k<2, 2, true, false, false, 2, true>), grid, blk, 0, st, a);
    } else if (p.mf == 2 && p.nf == 2) WR_LAUNCH2(2, 2);
    else if (p.mf == 2) WR_LAUNCH2(2, 1);
    else if (p.nf == 2) WR_LAUNCH2(1, 2);
    else if (bias) hipLaunchKernelGGL((wgrad_r_k<1, 1, true, true, false, 2>), grid, blk, 0, st, a);
    else WR_LAUNCH2(1, 1);
  } else
#undef WR_LAUNCH2
  if (a.twin) {        // 16-pixel-wide maps: 32 x 32 blocks, no bias gradient, dY materialised
    WTPSE_REQUIRE(p.mf == 2 && p.nf == 2 && !bias && !aff);
    if (g_x3_terms == 1) {
      if (pro) hipLaunchKernelGGL((wgrad_r_k<2, 2, true, false, false, 1, true>), grid, blk, 0, st, a);
      else hipLaunchKernelGGL((wgrad_r_k<2, 2, false, false, false, 1, true>), grid, blk, 0, st, a);
    } else {
      if (pro) hipLaunchKernelGGL((wgrad_r_k<2, 2, true, false, false, 3, true>), grid, blk, 0, st, a);
      else hipLaunchKernelGGL((wgrad_r_k<2, 2, false, false, false, 3, true>), grid, blk, 0, st, a);
    }
  } else
  if (g_x3_terms == 1 && !aff && p.mf * p.nf >= 2) {
    if (p.mf == 2 && p.nf == 2) WR_LAUNCH1(2, 2);
    else if (p.mf == 2) WR_LAUNCH1(2, 1);
    else WR_LAUNCH1(1, 2);
  } else
  if (p.mf == 2 && p.nf == 2) WR_LAUNCH(2, 2);
  else if (p.mf == 2) WR_LAUNCH(2, 1);
  else if (p.nf == 2) WR_LAUNCH(1, 2);
  else if (bias) {
    if (pro) hipLaunchKernelGGL((wgrad_r_k<1, 1, true, true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((wgrad_r_k<1, 1, false, true>), grid, blk, 0, st, a);
  } else WR_LAUNCH(1, 1);
#undef WR_LAUNCH
#undef WR_LAUNCH1
  int rc = wtpse_status();
  if (rc) return rc;
  wtpse_wgrad_reduce_launch2(slab, nslab, Cout * Cin * 9, dw, accumulate, dbias_slab, Cout, dbias, stream);
  return wtpse_status();
}

// Same contract as wtpse_conv_wgrad (include/wtpse_hip.h), 3x3 only; requires wtpse_wgrad_r_supported().
extern "C" int wtpse_conv_wgrad_r(const float* dy, const float* x0, int C0, const float* x1, int C1, const float* pro0,
                                  const float* pro1, int pro_relu, float* slab, float* dbias_slab, int nslab, float* dw,
                                  float* dbias, int accumulate, int B, int H, int W, int Cout, const unsigned* dy_amax,
                                  const unsigned* x_amax0, const unsigned* x_amax1, void* stream) {
  return wgrad_r_impl(dy, x0, C0, x1, C1, pro0, pro1, pro_relu, slab, dbias_slab, nslab, dw, dbias, accumulate, B, H, W, Cout,
                      nullptr, nullptr, dy_amax, x_amax0, x_amax1, stream);
}

// The same with dY given as the un-applied second half of a BatchNorm backward: dY = k1[c] * g + k2[c] * bn_y + k3[c],
// bn_coef [Cout][3] (wtpse_bn_bwd_coef); no bias gradient.
extern "C" int wtpse_conv_wgrad_r_bn(const float* g, const float* bn_y, const float* bn_coef, const float* x0, int C0,
                                     const float* x1, int C1, const float* pro0, const float* pro1, int pro_relu, float* slab,
                                     int nslab, float* dw, int accumulate, int B, int H, int W, int Cout, void* stream) {
  WTPSE_REQUIRE(bn_y && bn_coef);
  return wgrad_r_impl(g, x0, C0, x1, C1, pro0, pro1, pro_relu, slab, nullptr, nslab, dw, nullptr, accumulate, B, H, W, Cout, bn_y,
                      bn_coef, nullptr, nullptr, nullptr, stream);
}
