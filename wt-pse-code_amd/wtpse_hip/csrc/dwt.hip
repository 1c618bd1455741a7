// Standalone 2-D discrete wavelet transform (Haar / Daubechies-4 by lifting), analysis and synthesis, multi-level.
//
// NOT part of WT-PSE: the reference has no wavelet transform (its "WT" is the whitening transform) and
// nothing here is called from update()/predict().  BASELINE.json's wording names a "2-D DWT analysis/synthesis filter bank"
// with "wavefront shuffles for the separable Haar/Db lifting steps" and a "4-level DWT" HBM stress configuration; SURVEY.md
// §8f-4 allows it only as a standalone HBM-bandwidth micro-benchmark with a self-defined specification: oracle/dwt_cpu.py
// (periodic extension, orthonormal, Mallat layout), against which tests/test_dwt.py checks these kernels.  Parity unpinned.
//
// One launch per level (HBM-bound: reads and writes the level's h x w region once).  A lane owns one COLUMN PAIR
// (x[.., 2c], x[.., 2c+1]: one 8-byte load per row) and walks down RP row pairs, so
//   * the horizontal lifting steps take their neighbours d1[c+1], s1[c-1] from the adjacent lanes with wave shuffles — the
//     64 lanes of a wave cover 62 column pairs plus one halo lane on either side (periodic wrap = index mod), no LDS;
//   * the vertical lifting steps take their neighbours from the lane's own registers (row pairs r-1 .. r+RP, again with the
//     two halo pairs wrapped), so each lane produces the four sub-band values of its column pair for RP row pairs.
// Loads are 8 bytes per lane (512 contiguous bytes per wave and row), stores 4 bytes per lane per sub-band (forward) /
// 8 bytes per lane per row (inverse).
#include "common.h"

namespace {
constexpr float R3 = 1.7320508075688772f;
constexpr float LA = 0.4330127018922193f;     // sqrt3 / 4
constexpr float LB = -0.0669872981077807f;    // (sqrt3 - 2) / 4
constexpr float C1 = 1.9318516525781366f;     // (sqrt3 + 1) / sqrt2
constexpr float C2 = 0.5176380902050415f;     // (sqrt3 - 1) / sqrt2
constexpr float IS2 = 0.7071067811865476f;    // 1 / sqrt2
constexpr int RP = 16;                        // row pairs per wave (plus one halo pair on either side)

struct DwtArgs {
  const float* src;     // forward: the level's input region; inverse: the LL band of the level
  long long src_plane;  // plane stride (floats)
  int src_pitch;
  float* ll;            // forward: where LL goes (next level's input, or the coefficient buffer); inverse: the output region
  long long ll_plane;
  int ll_pitch;
  float* coef;          // coefficient buffer [planes][H][W] (details are written to / read from its quadrants)
  long long coef_plane;
  int coef_pitch;
  int h, w;             // size of the region this level works on (input of the forward level / output of the inverse level)
};

// 1-D forward lifting on (e, o) with neighbours: d1 of the right neighbour, and for s1[n-1] the left neighbour's values.
template <int WV>
__device__ __forceinline__ void fwd_pair_h(float e, float o, float& s, float& d) {   // horizontal: neighbours via shuffles
  if (WV == 0) {
    s = (e + o) * IS2;
    d = (o - e) * IS2;
  } else {
    const float d1 = o - R3 * e;
    const float d1n = __shfl_down(d1, 1, 64);
    const float s1 = e + LA * d1 + LB * d1n;
    const float s1p = __shfl_up(s1, 1, 64);
    s = C1 * s1;
    d = C2 * (d1 + s1p);
  }
}
}  // namespace

template <int WV>
__global__ __launch_bounds__(256) void dwt2_fwd_k(DwtArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int npairs = a.w / 2, nrp = a.h / 2;
  const int c = blockIdx.x * 62 + lane - 1;                       // column pair (lanes 0 and 63 are halo lanes)
  const int cw = ((c % npairs) + npairs) % npairs;                // periodic
  const int r0 = (blockIdx.y * 4 + wave) * RP;
  const float* src = a.src + (size_t)blockIdx.z * a.src_plane;
  // horizontal pass on rows of the pairs r0-1 .. r0+RP
  float sE[RP + 2], dE[RP + 2], sO[RP + 2], dO[RP + 2];
#pragma unroll
  for (int k = 0; k < RP + 2; ++k) {
    const int rp = (((r0 - 1 + k) % nrp) + nrp) % nrp;
    const float2 re = *reinterpret_cast<const float2*>(src + (size_t)(2 * rp) * a.src_pitch + 2 * cw);
    const float2 ro = *reinterpret_cast<const float2*>(src + (size_t)(2 * rp + 1) * a.src_pitch + 2 * cw);
    fwd_pair_h<WV>(re.x, re.y, sE[k], dE[k]);
    fwd_pair_h<WV>(ro.x, ro.y, sO[k], dO[k]);
  }
  // vertical pass (neighbours in registers) and stores
  const bool col_ok = lane >= 1 && lane <= 62 && c < npairs;
  const int h2 = a.h / 2, w2 = a.w / 2;
  float* ll = a.ll + (size_t)blockIdx.z * a.ll_plane;
  float* cf = a.coef + (size_t)blockIdx.z * a.coef_plane;
  float d1s[RP + 2], d1d[RP + 2], s1s[RP + 2], s1d[RP + 2];
  if (WV == 1) {
#pragma unroll
    for (int k = 0; k < RP + 2; ++k) {
      d1s[k] = sO[k] - R3 * sE[k];
      d1d[k] = dO[k] - R3 * dE[k];
    }
#pragma unroll
    for (int k = 0; k < RP + 1; ++k) {
      s1s[k] = sE[k] + LA * d1s[k] + LB * d1s[k + 1];
      s1d[k] = dE[k] + LA * d1d[k] + LB * d1d[k + 1];
    }
  }
#pragma unroll
  for (int k = 1; k <= RP; ++k) {
    const int rp = r0 + k - 1;
    float vll, vlh, vhl, vhh;   // (horizontal band)(vertical band): ll, l-h = low horizontal/high vertical, ...
    if (WV == 0) {
      vll = (sE[k] + sO[k]) * IS2; vlh = (sO[k] - sE[k]) * IS2;
      vhl = (dE[k] + dO[k]) * IS2; vhh = (dO[k] - dE[k]) * IS2;
    } else {
      vll = C1 * s1s[k]; vlh = C2 * (d1s[k] + s1s[k - 1]);
      vhl = C1 * s1d[k]; vhh = C2 * (d1d[k] + s1d[k - 1]);
    }
    if (col_ok && rp < nrp) {
      ll[(size_t)rp * a.ll_pitch + c] = vll;
      cf[(size_t)rp * a.coef_pitch + w2 + c] = vhl;                 // top-right: high horizontal, low vertical
      cf[(size_t)(h2 + rp) * a.coef_pitch + c] = vlh;               // bottom-left: low horizontal, high vertical
      cf[(size_t)(h2 + rp) * a.coef_pitch + w2 + c] = vhh;
    }
  }
}

template <int WV>
__global__ __launch_bounds__(256) void dwt2_inv_k(DwtArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int npairs = a.w / 2, nrp = a.h / 2;
  const int c = blockIdx.x * 62 + lane - 1;
  const int cw = ((c % npairs) + npairs) % npairs;
  const int r0 = (blockIdx.y * 4 + wave) * RP;
  const int h2 = a.h / 2, w2 = a.w / 2;
  const float* llp = a.src + (size_t)blockIdx.z * a.src_plane;
  const float* cf = a.coef + (size_t)blockIdx.z * a.coef_plane;
  float vll[RP + 2], vlh[RP + 2], vhl[RP + 2], vhh[RP + 2];
#pragma unroll
  for (int k = 0; k < RP + 2; ++k) {
    const int rp = (((r0 - 1 + k) % nrp) + nrp) % nrp;
    vll[k] = llp[(size_t)rp * a.src_pitch + cw];
    vhl[k] = cf[(size_t)rp * a.coef_pitch + w2 + cw];
    vlh[k] = cf[(size_t)(h2 + rp) * a.coef_pitch + cw];
    vhh[k] = cf[(size_t)(h2 + rp) * a.coef_pitch + w2 + cw];
  }
  float* out = a.ll + (size_t)blockIdx.z * a.ll_plane;
  const bool col_ok = lane >= 1 && lane <= 62 && c < npairs;
#pragma unroll
  for (int k = 1; k <= RP; ++k) {
    const int rp = r0 + k - 1;
    // vertical inverse: (ll, lh) -> (sE, sO) of the low horizontal band, (hl, hh) -> (dE, dO) of the high one
    float sE, sO, dE, dO;
    if (WV == 0) {
      sE = (vll[k] - vlh[k]) * IS2; sO = (vll[k] + vlh[k]) * IS2;
      dE = (vhl[k] - vhh[k]) * IS2; dO = (vhl[k] + vhh[k]) * IS2;
    } else {
      const float i1 = 1.f / C1, i2 = 1.f / C2;
      {
        const float s1 = vll[k] * i1, s1p = vll[k - 1] * i1;
        const float d1 = vlh[k] * i2 - s1p, d1n = vlh[k + 1] * i2 - s1;
        sE = s1 - LA * d1 - LB * d1n;
        sO = d1 + R3 * sE;
      }
      {
        const float s1 = vhl[k] * i1, s1p = vhl[k - 1] * i1;
        const float d1 = vhh[k] * i2 - s1p, d1n = vhh[k + 1] * i2 - s1;
        dE = s1 - LA * d1 - LB * d1n;
        dO = d1 + R3 * dE;
      }
    }
    // horizontal inverse of the two rows (neighbours: s1 of the left lane, d2 of the right lane)
    float2 re, ro;
    if (WV == 0) {
      re.x = (sE - dE) * IS2; re.y = (sE + dE) * IS2;
      ro.x = (sO - dO) * IS2; ro.y = (sO + dO) * IS2;
    } else {
      const float i1 = 1.f / C1, i2 = 1.f / C2;
      {
        const float s1 = sE * i1, d2 = dE * i2;
        const float d1 = d2 - __shfl_up(s1, 1, 64);
        const float d1n = __shfl_down(d2, 1, 64) - s1;
        re.x = s1 - LA * d1 - LB * d1n;
        re.y = d1 + R3 * re.x;
      }
      {
        const float s1 = sO * i1, d2 = dO * i2;
        const float d1 = d2 - __shfl_up(s1, 1, 64);
        const float d1n = __shfl_down(d2, 1, 64) - s1;
        ro.x = s1 - LA * d1 - LB * d1n;
        ro.y = d1 + R3 * ro.x;
      }
    }
    if (col_ok && rp < nrp) {
      *reinterpret_cast<float2*>(out + (size_t)(2 * rp) * a.ll_pitch + 2 * c) = re;
      *reinterpret_cast<float2*>(out + (size_t)(2 * rp + 1) * a.ll_pitch + 2 * c) = ro;
    }
  }
}

static dim3 dwt_grid(int h, int w, int planes) {
  return dim3((unsigned)ceil_div(w / 2, 62), (unsigned)ceil_div(h / 2, 4 * RP), (unsigned)planes);
}

// See include/wtpse_hip.h.  tmp: 2 * planes * (H/2) * (W/2) floats.
extern "C" int wtpse_dwt2_fwd(const float* x, float* coef, float* tmp, int planes, int H, int W, int wavelet, int levels,
                              void* stream) {
  WTPSE_REQUIRE(x && coef && tmp && planes > 0 && planes <= 65535 && H > 0 && W > 0 && levels >= 1 && levels <= 12);
  WTPSE_REQUIRE((wavelet == 0 || wavelet == 1) && H % (1 << levels) == 0 && W % (1 << levels) == 0 && x != coef);
  hipStream_t st = (hipStream_t)stream;
  const long long tplane = (long long)(H / 2) * (W / 2);
  float* tbuf[2] = {tmp, tmp + (size_t)planes * tplane};
  DwtArgs a;
  a.coef = coef; a.coef_plane = (long long)H * W; a.coef_pitch = W;
  int h = H, w = W;
  for (int lv = 0; lv < levels; ++lv) {
    const bool last = lv == levels - 1;
    if (lv == 0) { a.src = x; a.src_plane = (long long)H * W; a.src_pitch = W; }
    else { a.src = tbuf[(lv - 1) & 1]; a.src_plane = tplane; a.src_pitch = w; }
    if (last) { a.ll = coef; a.ll_plane = (long long)H * W; a.ll_pitch = W; }
    else { a.ll = tbuf[lv & 1]; a.ll_plane = tplane; a.ll_pitch = w / 2; }
    a.h = h; a.w = w;
    if (wavelet == 0) hipLaunchKernelGGL(dwt2_fwd_k<0>, dwt_grid(h, w, planes), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dwt2_fwd_k<1>, dwt_grid(h, w, planes), dim3(256), 0, st, a);
    h /= 2; w /= 2;
  }
  return wtpse_status();
}

extern "C" int wtpse_dwt2_inv(const float* coef, float* x, float* tmp, int planes, int H, int W, int wavelet, int levels,
                              void* stream) {
  WTPSE_REQUIRE(x && coef && tmp && planes > 0 && planes <= 65535 && H > 0 && W > 0 && levels >= 1 && levels <= 12);
  WTPSE_REQUIRE((wavelet == 0 || wavelet == 1) && H % (1 << levels) == 0 && W % (1 << levels) == 0 && x != coef);
  hipStream_t st = (hipStream_t)stream;
  const long long tplane = (long long)(H / 2) * (W / 2);
  float* tbuf[2] = {tmp, tmp + (size_t)planes * tplane};
  DwtArgs a;
  a.coef = const_cast<float*>(coef); a.coef_plane = (long long)H * W; a.coef_pitch = W;
  for (int lv = levels - 1; lv >= 0; --lv) {
    const int h = H >> lv, w = W >> lv;            // output region of this level
    if (lv == levels - 1) { a.src = coef; a.src_plane = (long long)H * W; a.src_pitch = W; }
    else { a.src = tbuf[(lv + 1) & 1]; a.src_plane = tplane; a.src_pitch = w / 2; }
    if (lv == 0) { a.ll = x; a.ll_plane = (long long)H * W; a.ll_pitch = W; }
    else { a.ll = tbuf[lv & 1]; a.ll_plane = tplane; a.ll_pitch = w; }
    a.h = h; a.w = w;
    if (wavelet == 0) hipLaunchKernelGGL(dwt2_inv_k<0>, dwt_grid(h, w, planes), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dwt2_inv_k<1>, dwt_grid(h, w, planes), dim3(256), 0, st, a);
  }
  return wtpse_status();
}
