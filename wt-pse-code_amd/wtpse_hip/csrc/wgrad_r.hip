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
  float* slab;     // [nslab][Cout][Cin][9]
  float* slab_b;   // [nslab_b][Cout] or null
  int B, H, W, C0, C1, Cin, Cout;
  int pro_relu;
  int strips;      // W / 32
  int nseg, rseg;  // row segments per (image, strip) and their height
  int units;       // B * strips * nseg
  int wpp;         // waves per (cout block, cin block) pair (multiple of 4)
  int nci;         // cin blocks
};

__device__ __forceinline__ unsigned wr_pack_rne(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// (a, b) -> three dwords holding the bf16 pairs (term_i(a), term_i(b)): see split3_pair in conv_x3.hip
__device__ __forceinline__ void wr_split3_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = wr_pack_rne(a, b);
  const float ra = a - __builtin_bit_cast(float, p0 << 16);
  const float rb = b - __builtin_bit_cast(float, p0 & 0xFFFF0000u);
  p1 = wr_pack_rne(ra, rb);
  const float sa = ra - __builtin_bit_cast(float, p1 << 16);
  const float sb = rb - __builtin_bit_cast(float, p1 & 0xFFFF0000u);
  p2 = wr_pack_rne(sa, sb);
}

__device__ __forceinline__ f32x4 mfma16x32(u32x4v a, u32x4v b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// 8 fp32 values -> three fragments (4 dwords each) of bf16 terms
__device__ __forceinline__ void wr_split8(const f32x4& lo, const f32x4& hi, u32x4v (&t)[3]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    unsigned q0, q1, q2;
    wr_split3_pair(lo[2 * j], lo[2 * j + 1], q0, q1, q2);
    t[0][j] = q0; t[1][j] = q1; t[2][j] = q2;
    wr_split3_pair(hi[2 * j], hi[2 * j + 1], q0, q1, q2);
    t[0][2 + j] = q0; t[1][2 + j] = q1; t[2][2 + j] = q2;
  }
}

template <int MF, int NF, bool PRO, bool BIAS>
__global__ __launch_bounds__(256, (MF * NF >= 2) ? 1 : 2) void wgrad_r_k(WgradRArgs a) {
  constexpr int NT = 9;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int wgpp = a.wpp >> 2;                           // workgroups per pair
  const int pair = blockIdx.x / wgpp, wg = blockIdx.x - pair * wgpp;
  const int cin0 = (pair % a.nci) * (16 * NF), cout0 = (pair / a.nci) * (16 * MF);
  const int wv = wg * 4 + wave;
  const int u0 = (int)((long long)wv * a.units / a.wpp), u1 = (int)((long long)(wv + 1) * a.units / a.wpp);
  const int HW = a.H * a.W;
  const unsigned W4 = (unsigned)a.W * 4u;

  f32x4 acc[MF][NF][NT];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][t][r] = 0.f;
  float bsum[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) bsum[m] = 0.f;

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
    }
  }
  const int src_up = (lane + 16) & 63, src_dn = (lane - 16) & 63;

  for (int u = u0; u < u1; ++u) {
    // unit -> (image, 32-pixel strip, row segment)
    const int seg = u % a.nseg;
    const int bs = u / a.nseg;
    const int strip = bs % a.strips, b = bs / a.strips;
    const int x0 = strip * 32;
    const int y0 = seg * a.rseg, y1 = min(a.H, y0 + a.rseg);
    const int rfirst = max(y0 - 1, 0), rlast = min(y1, a.H - 1);
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(a.dy + (size_t)b * a.Cout * HW, (unsigned)a.Cout * HW * 4u);
    const __amdgpu_buffer_rsrc_t rsx0 = make_rsrc(a.x0 + (size_t)b * a.C0 * HW, (unsigned)a.C0 * HW * 4u);
    const __amdgpu_buffer_rsrc_t rsx1 = a.x1 ? make_rsrc(a.x1 + (size_t)b * a.C1 * HW, (unsigned)a.C1 * HW * 4u) : rsx0;
    // byte offsets of this lane's 8 pixels in row 0
    unsigned ybase[MF], xbase[NF], ebase[NF];
    bool evalid;
    {
      // edge pixel of this lane: lane group 0 fetches the pixel right of the strip (x0 + 32), lane group 3 the one left of it
      const int ex = g == 0 ? x0 + 32 : x0 - 1;
      evalid = (g == 0 || g == 3) && ex >= 0 && ex < a.W;
#pragma unroll
      for (int m = 0; m < MF; ++m) ybase[m] = ((unsigned)(cout0 + 16 * m + c16) * HW + x0 + 8 * g) * 4u;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        xbase[n] = ((unsigned)xch[n] * HW + x0 + 8 * g) * 4u;
        ebase[n] = ((unsigned)xch[n] * HW + ex) * 4u;
      }
    }

    f32x4 rawx[NF][2], rawy[MF][2];
    float rawe[NF];
    // loads of X row r and dY row r + 1 (rows outside their range read as zero: out-of-range buffer offsets)
    auto issue = [&](int r, bool want_x) {
      const bool xv = want_x && r <= rlast;
      const unsigned ro = (unsigned)r * W4;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const __amdgpu_buffer_rsrc_t rs = xfirst[n] ? rsx0 : rsx1;
        const unsigned vo = xv ? xbase[n] + ro : BUF_OOB;
        rawx[n][0] = buf_load4(rs, vo, 0);
        rawx[n][1] = buf_load4(rs, vo, 16);
        rawe[n] = buf_load(rs, (xv && evalid) ? ebase[n] + ro : BUF_OOB, 0);
      }
      const int y = r + 1;
      const bool yv = y >= y0 && y < y1;
      const unsigned yo = (unsigned)y * W4;
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const unsigned vo = yv ? ybase[m] + yo : BUF_OOB;
        rawy[m][0] = buf_load4(rsy, vo, 0);
        rawy[m][1] = buf_load4(rsy, vo, 16);
      }
    };

    u32x4v ay[3][MF][3];          // ring of split dY rows: [slot][cout fragment][term]
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int t = 0; t < 3; ++t) ay[s][m][t] = (u32x4v){0u, 0u, 0u, 0u};

    // one X row: J = (r - rfirst) mod 3 selects the ring slots statically.  dY row y lives in slot (y - rfirst + 3) mod 3.
    auto step = [&](auto Jc, int r) {
      constexpr int J = decltype(Jc)::value;
      // ---- the dY row that arrived (row r + 1) -> slot (J + 1) % 3
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        if (BIAS) bsum[m] += ((rawy[m][0][0] + rawy[m][0][1]) + (rawy[m][0][2] + rawy[m][0][3])) +
                             ((rawy[m][1][0] + rawy[m][1][1]) + (rawy[m][1][2] + rawy[m][1][3]));
        wr_split8(rawy[m][0], rawy[m][1], ay[(J + 1) % 3][m]);
      }
      // ---- the X row that arrived (row r): prologue, split; edge pixel
      u32x4v xb[NF][3];
      unsigned eq[NF][3];
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        f32x4 lo = rawx[n][0], hi = rawx[n][1];
        float e = rawe[n];
        if (PRO) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lo[j] = fmaxf(fmaf(lo[j], psc[n], psh[n]), plo[n]);
            hi[j] = fmaxf(fmaf(hi[j], psc[n], psh[n]), plo[n]);
          }
          e = evalid ? fmaxf(fmaf(e, psc[n], psh[n]), plo[n]) : 0.f;     // zero padding applies after the activation
        }
        wr_split8(lo, hi, xb[n]);
        wr_split3_pair(e, 0.f, eq[n][0], eq[n][1], eq[n][2]);
      }
      // ---- next rows' loads behind this row's arithmetic
      issue(r + 1, true);
      // ---- products
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        u32x4v xs[3][3];          // [kx][term]: the row shifted by kx - 1 pixels
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const u32x4v d = xb[n][t];
          // pixel 8 of this lane's group = pixel 0 of the next group (lane + 16); for the last group the strip's right edge,
          // which lane group 0 holds.  Pixel -1 = pixel 7 of the previous group (lane - 16); for group 0 the left edge (group 3).
          const unsigned sup_r = g == 0 ? eq[n][t] : d[0];
          const unsigned sup_l = g == 3 ? (eq[n][t] << 16) : d[3];
          const unsigned nb_r = (unsigned)__builtin_amdgcn_ds_bpermute(src_up * 4, (int)sup_r);
          const unsigned nb_l = (unsigned)__builtin_amdgcn_ds_bpermute(src_dn * 4, (int)sup_l);
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
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int y = r + 1 - ky;                      // the dY row this X row meets under vertical tap ky
          if (y < y0 || y >= y1) continue;               // wave-uniform
          const int slot = (J + 1 - ky + 3) % 3;         // folds: J and ky are compile-time
#pragma unroll
          for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              f32x4 c = acc[m][n][ky * 3 + kx];
              // the six leading cross terms, smallest first (as conv_x3.hip)
              c = mfma16x32(ay[slot][m][0], xs[kx][2], c);
              c = mfma16x32(ay[slot][m][1], xs[kx][1], c);
              c = mfma16x32(ay[slot][m][2], xs[kx][0], c);
              c = mfma16x32(ay[slot][m][0], xs[kx][1], c);
              c = mfma16x32(ay[slot][m][1], xs[kx][0], c);
              c = mfma16x32(ay[slot][m][0], xs[kx][0], c);
              acc[m][n][ky * 3 + kx] = c;
            }
        }
      }
    };

    // dY row rfirst (needed by the first X row under ky = 1) arrives with the "previous" row's loads
    issue(rfirst - 1, false);                             // (no X row: rfirst - 1 lies outside the unit or the image)
    {
      // consume the dY row into slot 0
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        if (BIAS) bsum[m] += ((rawy[m][0][0] + rawy[m][0][1]) + (rawy[m][0][2] + rawy[m][0][3])) +
                             ((rawy[m][1][0] + rawy[m][1][1]) + (rawy[m][1][2] + rawy[m][1][3]));
        wr_split8(rawy[m][0], rawy[m][1], ay[0][m]);
      }
      issue(rfirst, true);
    }
    int r = rfirst;
    for (; r + 2 <= rlast; r += 3) {
      step(std::integral_constant<int, 0>{}, r);
      step(std::integral_constant<int, 1>{}, r + 1);
      step(std::integral_constant<int, 2>{}, r + 2);
    }
    if (r <= rlast) step(std::integral_constant<int, 0>{}, r);
    if (r + 1 <= rlast) step(std::integral_constant<int, 1>{}, r + 1);
  }

  // ---- the four waves' partial sums meet in LDS, one vertical tap (3 x MF x NF tiles) at a time
  __shared__ f32x4 red[4][3 * MF * NF][64];
  float* out = a.slab + (size_t)wg * a.Cout * a.Cin * NT;      // slab `wg`: every pair writes its own (cout, cin) block of it
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
    for (int e = tid; e < 3 * MF * NF * 64; e += 256) {
      const int l = e & 63, f = e >> 6;
      const int kx = f % 3, mn = f / 3, n = mn % NF, m = mn / NF;
      const f32x4 s = (red[0][f][l] + red[1][f][l]) + (red[2][f][l] + red[3][f][l]);
      const int ci = cin0 + 16 * n + (l & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = cout0 + 16 * m + 4 * (l >> 4) + i;
        out[((size_t)co * a.Cin + ci) * NT + ky * 3 + kx] = s[i];
      }
    }
  }
  if (BIAS) {
    if (cin0 == 0) {
      __syncthreads();
      float* redb = reinterpret_cast<float*>(&red[0][0][0]);
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        float v = bsum[m];
        v += __shfl_xor(v, 16, 64);                       // the four pixel groups of a channel
        v += __shfl_xor(v, 32, 64);
        if (g == 0) redb[(wave * MF + m) * 16 + c16] = v;
      }
      __syncthreads();
      if (tid < 16 * MF) {
        const int m = tid >> 4, c = tid & 15;
        const float s = (redb[(0 * MF + m) * 16 + c] + redb[(1 * MF + m) * 16 + c]) +
                        (redb[(2 * MF + m) * 16 + c] + redb[(3 * MF + m) * 16 + c]);
        a.slab_b[(size_t)wg * a.Cout + cout0 + 16 * m + c] = s;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct WgradRPlan {
  int mf, nf, pairs, nci, wpp, nseg, rseg, units, strips;
};

static bool wgrad_r_plan(int B, int H, int W, int Cin, int Cout, WgradRPlan& p) {
  if (W % 32 != 0 || H < 1 || Cin % 16 != 0 || Cout % 16 != 0) return false;
  p.mf = Cout % 32 == 0 ? 2 : 1;
  p.nf = Cin % 32 == 0 ? 2 : 1;
  p.nci = Cin / (16 * p.nf);
  p.pairs = (Cout / (16 * p.mf)) * p.nci;
  p.strips = W / 32;
  // waves: one per SIMD for the 32 x 32 blocks (more than 256 registers), two per SIMD otherwise
  const int target = (p.mf * p.nf >= 2) ? 1024 : 2048;
  int wpp = (target / p.pairs) & ~3;
  if (wpp < 4) wpp = 4;
  const int cols = B * p.strips;                          // (image, strip) columns of H rows
  int nseg = 1;
  if (cols < wpp) nseg = ceil_div(wpp, cols);
  if (nseg > H) nseg = H;
  p.rseg = ceil_div(H, nseg);
  p.nseg = ceil_div(H, p.rseg);
  p.units = cols * p.nseg;
  if (wpp > p.units) wpp = (p.units + 3) & ~3;            // trailing waves get no unit: they contribute zero slabs
  p.wpp = wpp;
  return true;
}

extern "C" int wtpse_wgrad_r_supported(int Cin, int Cout, int ksize, int C0, int W) {
  return ksize == 3 && Cin % 16 == 0 && Cout % 16 == 0 && C0 % 16 == 0 && W % 32 == 0;
}

// slabs of a wtpse_conv_wgrad_r launch: `slab` holds that many [Cout][Cin][9] partial gradients, `dbias_slab` as many [Cout]
extern "C" int wtpse_wgrad_r_slabs(int B, int H, int W, int Cin, int Cout) {
  WgradRPlan p;
  if (!wgrad_r_plan(B, H, W, Cin, Cout, p)) return 0;
  return p.wpp / 4;
}

extern "C" void wtpse_wgrad_reduce_launch2(const float* slab, int ksplit, int n, float* dw, int accumulate, const float* slab_b,
                                           int n_b, float* db, void* stream);

// Same contract as wtpse_conv_wgrad (include/wtpse_hip.h), 3x3 only; requires wtpse_wgrad_r_supported().
extern "C" int wtpse_conv_wgrad_r(const float* dy, const float* x0, int C0, const float* x1, int C1, const float* pro0,
                                  const float* pro1, int pro_relu, float* slab, float* dbias_slab, int nslab, float* dw,
                                  float* dbias, int accumulate, int B, int H, int W, int Cout, void* stream) {
  WTPSE_REQUIRE(dy && x0 && slab && dw && B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0);
  WTPSE_REQUIRE((C1 == 0) == (x1 == nullptr));
  WTPSE_REQUIRE((dbias == nullptr) == (dbias_slab == nullptr));
  const int Cin = C0 + C1;
  WTPSE_REQUIRE(wtpse_wgrad_r_supported(Cin, Cout, 3, C1 ? C0 : 16, W));
  WgradRPlan p;
  WTPSE_REQUIRE(wgrad_r_plan(B, H, W, Cin, Cout, p));
  WTPSE_REQUIRE(nslab == p.wpp / 4);
  WTPSE_REQUIRE((long long)(C0 > C1 ? C0 : C1) * H * W * 4 < (1ll << 31) && (long long)Cout * H * W * 4 < (1ll << 31));
  WgradRArgs a;
  a.dy = dy; a.x0 = x0; a.x1 = x1; a.pro0 = pro0; a.pro1 = pro1; a.slab = slab; a.slab_b = dbias_slab;
  a.B = B; a.H = H; a.W = W; a.C0 = C0; a.C1 = C1; a.Cin = Cin; a.Cout = Cout; a.pro_relu = pro_relu;
  a.strips = p.strips; a.nseg = p.nseg; a.rseg = p.rseg; a.units = p.units; a.wpp = p.wpp; a.nci = p.nci;
  const bool pro = pro0 != nullptr || pro1 != nullptr || pro_relu != 0;
  const bool bias = dbias != nullptr;
  dim3 grid((unsigned)(p.pairs * (p.wpp / 4)));
  hipStream_t st = (hipStream_t)stream;
#define WR_LAUNCH(M, N) do { \
    if (pro) { if (bias) hipLaunchKernelGGL((wgrad_r_k<M, N, true, true>), grid, dim3(256), 0, st, a); \
               else hipLaunchKernelGGL((wgrad_r_k<M, N, true, false>), grid, dim3(256), 0, st, a); } \
    else { if (bias) hipLaunchKernelGGL((wgrad_r_k<M, N, false, true>), grid, dim3(256), 0, st, a); \
           else hipLaunchKernelGGL((wgrad_r_k<M, N, false, false>), grid, dim3(256), 0, st, a); } } while (0)
  if (p.mf == 2 && p.nf == 2) WR_LAUNCH(2, 2);
  else if (p.mf == 2) WR_LAUNCH(2, 1);
  else if (p.nf == 2) WR_LAUNCH(1, 2);
  else WR_LAUNCH(1, 1);
#undef WR_LAUNCH
  int rc = wtpse_status();
  if (rc) return rc;
  wtpse_wgrad_reduce_launch2(slab, nslab, Cout * Cin * 9, dw, accumulate, dbias_slab, Cout, dbias, stream);
  return wtpse_status();
}
