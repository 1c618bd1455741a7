// The 1x1 heads of the WT-PSE networks as ONE kernel per direction instead of three convolutions:
//   32 -> 32 (ReLU) -> 8 [-> (ReLU) -> nc]      (reference algorithms.py:1006-1012 mu_prior / logvar_prior: three layers;
//                                                 :1199-1200 the segmentation net's `mu`: two layers, 8-channel output)
// As separate convolutions a head reads / writes 113 floats per pixel forward and 266 backward (every layer streams its
// input, output, mask and gradient through HBM at ~3.8 TB/s); chained in registers it is 73 / 105.
//
// Chaining needs no data movement: v_mfma_f32_32x32x2_f32 leaves D[row][px] with the pixel on the lane (l & 31) and the
// row in (register r, half-wave u = l >> 5): row = rho(r,u) = (r & 3) + 8 (r >> 2) + 4u.  The B operand of the next
// MFMA wants B[k][px] with the pixel on the same lane and k = 2*step + u, so register r of the previous result IS the
// B operand of step r if the weights (A operand) are loaded with the permuted k order rho(step, u).
//
// The weight gradients contract over pixels, which sit on lanes: the backward kernel transposes its four operands
// (dh1, x, h1, dh2) through a wave-private LDS tile [32 rows][33] per 32-pixel block and keeps dW1 / dW2 in MFMA
// accumulators over all the blocks a wave owns; per-workgroup slabs are folded in fp64 in a fixed order.
#include "common.h"

__device__ __forceinline__ int rho(int r, int u) { return (r & 3) + 8 * (r >> 2) + 4 * u; }
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
// the inner ReLUs of the x2h heads: a NaN stays a NaN (torch.relu keeps it; v_max_f32 would return 0 and hide a diverged input from the
// reference's isnan scrub of mu, shape_networks.py:490)
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ f32x16 mfma32h(u32x4v a, u32x4v b, f32x16 c) {      // 32x32x16, fp16 operands, fp32 accumulation
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

extern int g_x3_terms;      // conv_x3.hip: wtpse_x3_terms() — 2 (x2h, the default): the heads run on the fp16 matrix cores too

struct HeadArgs {
  const float* x;      // [B][32][HW] head input as stored
  const float* pro;    // [32][2] scale/shift applied on load, or null
  int pro_relu;        // ReLU after the affine
  const float *w1, *b1, *w2, *b2, *w3, *b3;   // [32][32],[32],[8][32],[8],[nc][8],[nc]  (w3 null: two-layer head)
  int nc;
  float* h1;           // [B][32][HW] relu(layer 1), or null (forward without tape)
  float* h2;           // [B][8][HW]  layer 2 (after ReLU for a three-layer head), or null
  float* y;            // [B][nc][HW] three-layer head output
  const float* dy;     // backward: [B][nc][HW] (three layers) or [B][8][HW] (two layers)
  float* dx;           // backward: [B][32][HW] gradient wrt the activated input
  float* slab;         // backward: [gridDim.x][NS] partial weight gradients
  int B, HW, nblk;     // nblk = B*HW/32 pixel blocks
  const unsigned* x_amax;   // x2h kernels: amax table bounding the ACTIVATED input (common.h), or null: the fixed forward scale
  const unsigned* dy_amax;  // x2h backward: amax table of dy (mandatory there)
};

#define HEAD_MAXNC 4

template <bool L3>
__global__ __launch_bounds__(256, 2) void head_fwd_k(HeadArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane >> 5, n = lane & 31;
  const int HW = a.HW, bpi = HW / 32;   // blocks per image
  float a1[16], a2[16], bb1[16], bb2[4], w3r[HEAD_MAXNC][4], bb3[HEAD_MAXNC];
#pragma unroll
  for (int s = 0; s < 16; ++s) a1[s] = a.w1[n * 32 + 2 * s + u];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    a2[r] = n < 8 ? a.w2[n * 32 + rho(r, u)] : 0.f;
    bb1[r] = a.b1[rho(r, u)];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) bb2[r] = a.b2[4 * u + r];
  if (L3) {
#pragma unroll
    for (int j = 0; j < HEAD_MAXNC; ++j) {
      bb3[j] = j < a.nc ? a.b3[j] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) w3r[j][r] = j < a.nc ? a.w3[j * 8 + 4 * u + r] : 0.f;
    }
  }
  // the prologue's coefficients of this lane's 16 input channels, once: read through `a.pro` inside the loop they were re-loaded for every
  // 32-pixel block (the compiler cannot prove that the stores to h1 / h2 / y leave them alone) — 32 loads and their waits per block, as
  // expensive as the block's 32 MFMAs (ablation: 165 -> 113 us without the prologue, 113 without the MFMAs)
  float psc[16], psh[16];
  const bool has_pro = a.pro != nullptr, pro_relu = a.pro_relu != 0;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    psc[s] = has_pro ? a.pro[2 * (2 * s + u)] : 1.f;
    psh[s] = has_pro ? a.pro[2 * (2 * s + u) + 1] : 0.f;
  }
  // the next block's input is fetched while this block runs through its two dependent MFMA chains
  const int stride = gridDim.x * 4;
  float xn[16];
  auto fetch = [&](int blk) {
    const int bq = min(blk, a.nblk - 1);          // past the end: a valid re-read, never used
    const int b = bq / bpi, p = (bq - b * bpi) * 32 + n;
    const float* xb = a.x + (size_t)b * 32 * HW + p;
#pragma unroll
    for (int s = 0; s < 16; ++s) xn[s] = xb[(size_t)(2 * s + u) * HW];
  };
  fetch(blockIdx.x * 4 + wave);
  for (int blk = blockIdx.x * 4 + wave; blk < a.nblk; blk += stride) {
    const int b = blk / bpi, p = (blk - b * bpi) * 32 + n;
    float xv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float v = has_pro ? fmaf(xn[s], psc[s], psh[s]) : xn[s];
      xv[s] = pro_relu ? fmaxf(v, 0.f) : v;
    }
    fetch(blk + stride);
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc1 = mfma32(a1[s], xv[s], acc1);
    float h1v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) h1v[r] = fmaxf(acc1[r] + bb1[r], 0.f);
    if (a.h1) {
      float* hb = a.h1 + (size_t)b * 32 * HW + p;
#pragma unroll
      for (int r = 0; r < 16; ++r) hb[(size_t)rho(r, u) * HW] = h1v[r];
    }
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2 = mfma32(a2[r], h1v[r], acc2);
    float h2v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      h2v[r] = acc2[r] + bb2[r];
      if (L3) h2v[r] = fmaxf(h2v[r], 0.f);
    }
    if (a.h2) {
      float* hb = a.h2 + (size_t)b * 8 * HW + p;
#pragma unroll
      for (int r = 0; r < 4; ++r) hb[(size_t)(4 * u + r) * HW] = h2v[r];
    }
    if (L3) {
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) {
        if (j < a.nc) {
          float part = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) part = fmaf(w3r[j][r], h2v[r], part);
          const float tot = part + __shfl_xor(part, 32, 64);
          if (u == 0) a.y[((size_t)b * a.nc + j) * HW + p] = tot + bb3[j];
        }
      }
    }
  }
}

// ---- x2h form of the forward (round 6).  The fp32-input MFMA runs at 64 cycles per 32x32x2 step: the kernel above spends 2048
// matrix-pipe cycles per 32-pixel block and wave — 110 us per [32,32,256,256] launch whatever it stores, twice the time of its bytes.
// Here both layers run on the fp16 matrix cores at fp32 accuracy exactly as the convolutions do (conv_x3_kernels.h: every operand
// = two fp16 terms of a power-of-two multiple of the value, three products, fp32 accumulation): 12 v_mfma_f32_32x32x16_f16 per block
// (384 cycles).  The chaining survives: a 32x32x16 B operand wants lane (pixel, h) to hold k = 8 h .. 8 h + 7 of a 16-wide K step,
// the previous result leaves lane (pixel, u) with rows rho(r, u) — registers 8 s .. 8 s + 7 of it ARE K step s if the next layer's
// weights take their k in the order rho(8 s + j, h).
// Scales (all powers of two, all exact): x by the bound in its amax table (or the fixed forward scale), W1 / W2 by their own largest
// magnitude, h1 by the bound max_m sum_k |W1[m][k]| * bound(x) + max |b1| — loose by the usual factor of a triangle inequality, which
// costs nothing: an operand keeps all 22 bits down to 2^-18 of the bound, and 2^-40 of the bound absolutely below that.
template <bool L3>
__global__ __launch_bounds__(256, 2) void head_fwd_h_k(HeadArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane >> 5, n = lane & 31;
  const int HW = a.HW, bpi = HW / 32;   // blocks per image
  const unsigned xraw = amax_load(a.x_amax);
  float w1v[2][8], w2v[2][8], bb1[16], bb2[4], w3r[HEAD_MAXNC][4], bb3[HEAD_MAXNC];
  unsigned m1 = 0u, m2 = 0u, mb = 0u;
  float rs1 = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      w1v[s][j] = a.w1[n * 32 + 16 * s + 8 * u + j];                       // layer 1: k = input channel, natural order
      w2v[s][j] = n < 8 ? a.w2[n * 32 + rho(8 * s + j, u)] : 0.f;           // layer 2: k in the order layer 1's result lies in registers
      m1 = max(m1, amax_bits(w1v[s][j]));
      m2 = max(m2, amax_bits(w2v[s][j]));
      rs1 += fabsf(w1v[s][j]);
    }
  rs1 += __shfl_xor(rs1, 32, 64);                                            // sum_k |W1[n][k]| over both k-halves
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    bb1[r] = a.b1[rho(r, u)];
    mb = max(mb, amax_bits(bb1[r]));
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) bb2[r] = a.b2[4 * u + r];
  if (L3) {
#pragma unroll
    for (int j = 0; j < HEAD_MAXNC; ++j) {
      bb3[j] = j < a.nc ? a.b3[j] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) w3r[j][r] = j < a.nc ? a.w3[j * 8 + 4 * u + r] : 0.f;
    }
  }
  const float sw1 = x3_scale_from_amax(wave_umax(m1)), sw2 = x3_scale_from_amax(wave_umax(m2));
  const unsigned xbits = a.x_amax ? amax_reduce(xraw) : 0u;
  const float sx = a.x_amax ? x3_scale_from_amax(xbits) : X3_FWD_SCALE;
  const float xbound = a.x_amax ? __builtin_bit_cast(float, xbits) : 32768.f / X3_FWD_SCALE;
  const float hbound = __builtin_bit_cast(float, wave_umax(amax_bits(rs1))) * xbound + __builtin_bit_cast(float, wave_umax(mb));
  const float sh = x3_scale_from_amax(amax_bits(hbound));
  const float inv1 = 1.f / (sx * sw1), inv2 = 1.f / (sh * sw2);
  u32x4v A1[2][2], A2[2][2];                    // [K step][term]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned q0, q1;
      split2h_pair_c(w1v[s][2 * i] * sw1, w1v[s][2 * i + 1] * sw1, q0, q1);
      A1[s][0][i] = q0; A1[s][1][i] = q1;
      split2h_pair_c(w2v[s][2 * i] * sw2, w2v[s][2 * i + 1] * sw2, q0, q1);
      A2[s][0][i] = q0; A2[s][1][i] = q1;
    }
  // the prologue's coefficients of this lane's 16 input channels (16 s + 8 u + j), times the input scale, once
  float psc[16], psh[16];
  const bool has_pro = a.pro != nullptr;
  const float lo = a.pro_relu ? 0.f : -INFINITY;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 16 * s + 8 * u + j;
      psc[8 * s + j] = (has_pro ? a.pro[2 * c] : 1.f) * sx;
      psh[8 * s + j] = (has_pro ? a.pro[2 * c + 1] : 0.f) * sx;
    }
  const int stride = gridDim.x * 4;
  float xn[16];
  // (per-image buffer descriptors: a lane's channel rows differ by a scalar offset — see head_bwd_h_k)
  const unsigned HW4 = (unsigned)HW * 4u;
  auto fetch = [&](int blk) {
    const int bq = min(blk, a.nblk - 1);          // past the end: a valid re-read, never used
    const int b = __builtin_amdgcn_readfirstlane(bq / bpi);
    const unsigned po = (unsigned)((bq - b * bpi) * 32 + n) * 4u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (size_t)b * 32 * HW, 32u * HW4);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) xn[8 * s + j] = buf_load(rx, po + (unsigned)(8 * u) * HW4, (unsigned)(16 * s + j) * HW4);
  };
  fetch(blockIdx.x * 4 + wave);
  for (int blk = blockIdx.x * 4 + wave; blk < a.nblk; blk += stride) {
    const int b = __builtin_amdgcn_readfirstlane(blk / bpi);
    const unsigned po = (unsigned)((blk - b * bpi) * 32 + n) * 4u;
    u32x4v B1[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = 8 * s + 2 * i;
        unsigned q0, q1;
        split2h_pair_c(fmaxf(fmaf(xn[e], psc[e], psh[e]), lo), fmaxf(fmaf(xn[e + 1], psc[e + 1], psh[e + 1]), lo), q0, q1);
        B1[s][0][i] = q0; B1[s][1][i] = q1;
      }
    fetch(blk + stride);
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {      // the cross terms first (smallest first, as the convolutions)
      acc1 = mfma32h(A1[s][0], B1[s][1], acc1);
      acc1 = mfma32h(A1[s][1], B1[s][0], acc1);
      acc1 = mfma32h(A1[s][0], B1[s][0], acc1);
    }
    float h1v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) h1v[r] = relu_keep_nan(fmaf(acc1[r], inv1, bb1[r]));
    if (a.h1) {
      const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.h1 + (size_t)b * 32 * HW, 32u * HW4);
#pragma unroll
      for (int r = 0; r < 16; ++r) buf_store(ro, po + (unsigned)(4 * u) * HW4, (unsigned)((r & 3) + 8 * (r >> 2)) * HW4, h1v[r]);
    }
    u32x4v B2[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned q0, q1;
        split2h_pair_c(h1v[8 * s + 2 * i] * sh, h1v[8 * s + 2 * i + 1] * sh, q0, q1);
        B2[s][0][i] = q0; B2[s][1][i] = q1;
      }
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      acc2 = mfma32h(A2[s][0], B2[s][1], acc2);
      acc2 = mfma32h(A2[s][1], B2[s][0], acc2);
      acc2 = mfma32h(A2[s][0], B2[s][0], acc2);
    }
    float h2v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      h2v[r] = fmaf(acc2[r], inv2, bb2[r]);
      if (L3) h2v[r] = relu_keep_nan(h2v[r]);
    }
    if (a.h2) {
      const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.h2 + (size_t)b * 8 * HW, 8u * HW4);
#pragma unroll
      for (int r = 0; r < 4; ++r) buf_store(ro, po + (unsigned)(4 * u) * HW4, (unsigned)r * HW4, h2v[r]);
    }
    if (L3) {
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) {
        if (j < a.nc) {
          float part = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) part = fmaf(w3r[j][r], h2v[r], part);
          const float tot = part + __shfl_xor(part, 32, 64);
          if (u == 0) a.y[((size_t)b * a.nc + j) * HW + (po >> 2)] = tot + bb3[j];
        }
      }
    }
  }
}

// slab layout per workgroup: dW1 [32][32] | db1 [32] | dW2 [8][32] | db2 [8] | dW3 [nc][8] | db3 [nc]
__host__ __device__ static inline int head_ns(int nc) { return 1024 + 32 + 256 + 8 + 8 * nc + nc; }

template <bool L3>
__global__ __launch_bounds__(256, 2) void head_bwd_k(HeadArgs a) {
  constexpr int TS = 33;                       // row stride of the transpose tiles: conflict-free column reads
  __shared__ float T[4][4][32 * TS];           // [wave][dh1 | x | h1 | dh2][row][px]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane >> 5, n = lane & 31;
  const int HW = a.HW, bpi = HW / 32;
  float* Tdh1 = T[wave][0];
  float* Tx = T[wave][1];
  float* Th1 = T[wave][2];
  float* Tdh2 = T[wave][3];
  for (int i = lane; i < 32 * TS; i += 64) Tdh2[i] = 0.f;   // rows 8..31 stay zero (dW2 runs as a padded 32-row GEMM)

  float aw1t[16], aw2t[4], w3r[HEAD_MAXNC][4];
#pragma unroll
  for (int r = 0; r < 16; ++r) aw1t[r] = a.w1[rho(r, u) * 32 + n];      // dx  = W1^T dh1: rows c = n, k = rho(r,u)
#pragma unroll
  for (int r = 0; r < 4; ++r) aw2t[r] = a.w2[(4 * u + r) * 32 + n];      // dh1 = W2^T dh2: rows k = n, m = 4u + r
  if (L3) {
#pragma unroll
    for (int j = 0; j < HEAD_MAXNC; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) w3r[j][r] = j < a.nc ? a.w3[j * 8 + 4 * u + r] : 0.f;
  }
  // the prologue's coefficients, once (see head_fwd_k) — through LDS here: this kernel has no 32 registers to spare (it spilled with them)
  __shared__ float2 pro_sh[32];
  const bool has_pro = a.pro != nullptr, pro_relu = a.pro_relu != 0;
  if (threadIdx.x < 32) pro_sh[threadIdx.x] = has_pro ? make_float2(a.pro[2 * threadIdx.x], a.pro[2 * threadIdx.x + 1]) : make_float2(1.f, 0.f);
  __syncthreads();
  f32x16 accW1, accW2;
#pragma unroll
  for (int r = 0; r < 16; ++r) accW1[r] = accW2[r] = 0.f;
  float sb1[16], sb2[4], sw3[HEAD_MAXNC][4], sb3[HEAD_MAXNC];
#pragma unroll
  for (int r = 0; r < 16; ++r) sb1[r] = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) sb2[r] = 0.f;
#pragma unroll
  for (int j = 0; j < HEAD_MAXNC; ++j) {
    sb3[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) sw3[j][r] = 0.f;
  }

  for (int blk = blockIdx.x * 4 + wave; blk < a.nblk; blk += gridDim.x * 4) {
    const int b = blk / bpi, p = (blk - b * bpi) * 32 + n;
    const float* xb = a.x + (size_t)b * 32 * HW + p;
    const float* hb = a.h1 + (size_t)b * 32 * HW + p;
    float xv[16], h1v[16], d2[4];
#pragma unroll
    for (int s = 0; s < 16; ++s) xv[s] = xb[(size_t)(2 * s + u) * HW];
#pragma unroll
    for (int r = 0; r < 16; ++r) h1v[r] = hb[(size_t)rho(r, u) * HW];
    if (L3) {
      float h2v[4], dyv[HEAD_MAXNC];
#pragma unroll
      for (int r = 0; r < 4; ++r) h2v[r] = a.h2[((size_t)b * 8 + 4 * u + r) * HW + p];
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) dyv[j] = j < a.nc ? a.dy[((size_t)b * a.nc + j) * HW + p] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float g = 0.f;
#pragma unroll
        for (int j = 0; j < HEAD_MAXNC; ++j) {
          g = fmaf(w3r[j][r], dyv[j], g);
          sw3[j][r] = fmaf(dyv[j], h2v[r], sw3[j][r]);
        }
        d2[r] = h2v[r] > 0.f ? g : 0.f;
      }
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) sb3[j] += u == 0 ? dyv[j] : 0.f;   // both half-waves hold the same pixel
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) d2[r] = a.dy[((size_t)b * 8 + 4 * u + r) * HW + p];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float2 pc = pro_sh[2 * s + u];
      const float v = has_pro ? fmaf(xv[s], pc.x, pc.y) : xv[s];
      xv[s] = pro_relu ? fmaxf(v, 0.f) : v;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sb2[r] += d2[r];
    // dh1 = (W2^T dh2) * [h1 > 0]
    f32x16 acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc3 = mfma32(aw2t[r], d2[r], acc3);
    float dh1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dh1[r] = h1v[r] > 0.f ? acc3[r] : 0.f;
      sb1[r] += dh1[r];
    }
    // dx = W1^T dh1
    f32x16 acc4;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc4[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc4 = mfma32(aw1t[r], dh1[r], acc4);
    {
      float* db = a.dx + (size_t)b * 32 * HW + p;
#pragma unroll
      for (int r = 0; r < 16; ++r) db[(size_t)rho(r, u) * HW] = acc4[r];
    }
    // weight gradients: pixels from lanes to the K dimension through the wave-private LDS tiles
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      Tdh1[rho(r, u) * TS + n] = dh1[r];
      Th1[rho(r, u) * TS + n] = h1v[r];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) Tx[(2 * s + u) * TS + n] = xv[s];
#pragma unroll
    for (int r = 0; r < 4; ++r) Tdh2[(4 * u + r) * TS + n] = d2[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int col = 2 * t + u;
      accW1 = mfma32(Tdh1[n * TS + col], Tx[n * TS + col], accW1);     // dW1[m][c] += dh1[m][px] x[c][px]
      accW2 = mfma32(Tdh2[n * TS + col], Th1[n * TS + col], accW2);    // dW2[m][k] += dh2[m][px] h1[k][px]
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  // ---- per-lane sums over this lane's pixels -> sums over the 32 lanes of the half-wave
#pragma unroll
  for (int m = 1; m <= 16; m <<= 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sb1[r] += __shfl_xor(sb1[r], m, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) sb2[r] += __shfl_xor(sb2[r], m, 64);
    if (L3) {
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) {
        sb3[j] += __shfl_xor(sb3[j], m, 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) sw3[j][r] += __shfl_xor(sw3[j][r], m, 64);
      }
    }
  }
  // ---- cross-wave reduction through LDS (the transpose tiles are free now), then this workgroup's slab
  const int NS = head_ns(L3 ? a.nc : 0);
  __syncthreads();
  float* red = &T[0][0][0] + wave * 1408;      // 4 x 1408 floats <= sizeof(T); NS <= 1356
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    red[rho(r, u) * 32 + n] = accW1[r];
    if (r < 4) red[1056 + (4 * u + r) * 32 + n] = accW2[r];            // dW2 rows m = rho(r,u) = r + 4u for r < 4
  }
  if (n == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[1024 + rho(r, u)] = sb1[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) red[1312 + 4 * u + r] = sb2[r];
    if (L3) {
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) {
        if (j < a.nc) {
#pragma unroll
          for (int r = 0; r < 4; ++r) red[1320 + j * 8 + 4 * u + r] = sw3[j][r];
          if (u == 0) red[1320 + 8 * a.nc + j] = sb3[j];
        }
      }
    }
  }
  __syncthreads();
  const float* r0 = &T[0][0][0];
  for (int e = threadIdx.x; e < NS; e += 256)
    a.slab[(size_t)blockIdx.x * NS + e] = (r0[e] + r0[1408 + e]) + (r0[2816 + e] + r0[4224 + e]);
}

// ---- x2h form of the backward (round 6): every product on the fp16 matrix cores (27 v_mfma_f32_32x32x16_f16 per block, 864 cycles,
// where the kernel above runs 52 fp32-input MFMAs, 3328 cycles), and NO layer-1 tape: h1 is formed again from the x the kernel reads
// anyway, by head_fwd_h_k's instruction sequence on head_fwd_h_k's operands (the same bits, so the ReLU mask is the forward's).
// Operand scales are powers of two from launch-wide bounds every wave derives for itself: bound(x) and bound(dy) from their amax
// tables, the others through the weights' absolute row sums (h1: W1, b1; dh2: W3; dh1: W2) — see head_fwd_h_k.
// The weight gradients contract over pixels: the four operands go through the wave-private LDS tiles as scaled fp32 values (row
// stride 36 floats: a lane reads 8 consecutive pixels of its row with two 16-byte reads) and are split behind the transpose.
template <bool L3>
__global__ __launch_bounds__(256, 2) void head_bwd_h_k(HeadArgs a) {
  constexpr int TS = 36;
  __shared__ __attribute__((aligned(16))) float T[4][4][32 * TS];           // [wave][dh1 | x | h1 | dh2][row][px]
  __shared__ float2 pro_sh[32];
  __shared__ float4 b1_sh[8], w3_sh[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, u = lane >> 5, n = lane & 31;
  const int HW = a.HW, bpi = HW / 32;
  float* Tdh1 = T[wave][0];
  float* Tx = T[wave][1];
  float* Th1 = T[wave][2];
  float* Tdh2 = T[wave][3];
  for (int i = lane; i < 32 * TS; i += 64) Tdh2[i] = 0.f;   // rows 8..31 stay zero (dW2 runs as a padded 32-row GEMM)
  const unsigned xraw = amax_load(a.x_amax), dyraw = amax_load(a.dy_amax);

  // ---- weights in the four operand layouts, their largest magnitudes and absolute row sums
  float w1v[2][8], w1t[2][8], w2t[4];
  unsigned m1 = 0u, m2 = 0u, mb = 0u;
  float rs1 = 0.f, cs2 = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      w1v[s][j] = a.w1[n * 32 + 16 * s + 8 * u + j];            // layer 1 again: A[m = n][k = 16 s + 8 u + j]
      w1t[s][j] = a.w1[rho(8 * s + j, u) * 32 + n];             // dx = W1^T dh1: A[c = n][k = rho(8 s + j, u)]
      m1 = max(m1, amax_bits(w1v[s][j]));
      rs1 += fabsf(w1v[s][j]);
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w2t[j] = a.w2[(4 * u + j) * 32 + n];                        // dh1 = W2^T dh2: A[k = n][m = 4 u + j] (K index 8 u + j, j < 4)
    m2 = max(m2, amax_bits(w2t[j]));
    cs2 += fabsf(w2t[j]);
  }
  rs1 += __shfl_xor(rs1, 32, 64);                               // sum_k |W1[n][k]|
  cs2 += __shfl_xor(cs2, 32, 64);                               // sum_m |W2[m][n]|
  float c3 = 1.f;                                               // L3: max_r sum_j |W3[j][r]|
  if (L3) {
    c3 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) t += j < a.nc ? fabsf(a.w3[j * 8 + r]) : 0.f;
      c3 = fmaxf(c3, t);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) mb = max(mb, amax_bits(a.b1[rho(r, u)]));
  const bool has_pro = a.pro != nullptr;
  if (threadIdx.x < 32) pro_sh[threadIdx.x] = has_pro ? make_float2(a.pro[2 * threadIdx.x], a.pro[2 * threadIdx.x + 1]) : make_float2(1.f, 0.f);
  if (threadIdx.x < 8) {
    const int bu = threadIdx.x >> 2, bq = threadIdx.x & 3;
    b1_sh[threadIdx.x] = make_float4(a.b1[rho(4 * bq, bu)], a.b1[rho(4 * bq + 1, bu)], a.b1[rho(4 * bq + 2, bu)], a.b1[rho(4 * bq + 3, bu)]);
    w3_sh[threadIdx.x] = (L3 && bq < a.nc) ? make_float4(a.w3[bq * 8 + 4 * bu], a.w3[bq * 8 + 4 * bu + 1], a.w3[bq * 8 + 4 * bu + 2], a.w3[bq * 8 + 4 * bu + 3])
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float sw1 = x3_scale_from_amax(wave_umax(m1)), sw2 = x3_scale_from_amax(wave_umax(m2));
  const unsigned xbits = a.x_amax ? amax_reduce(xraw) : 0u;
  const float sx = a.x_amax ? x3_scale_from_amax(xbits) : X3_FWD_SCALE;
  const float xbound = a.x_amax ? __builtin_bit_cast(float, xbits) : 32768.f / X3_FWD_SCALE;
  const float hbound = __builtin_bit_cast(float, wave_umax(amax_bits(rs1))) * xbound + __builtin_bit_cast(float, wave_umax(mb));
  const float sh = x3_scale_from_amax(amax_bits(hbound));
  const float d2bound = __builtin_bit_cast(float, amax_reduce(dyraw)) * c3;
  const float sd2 = x3_scale_from_amax(amax_bits(d2bound));
  const float d1bound = __builtin_bit_cast(float, wave_umax(amax_bits(cs2))) * d2bound;
  const float sd1 = x3_scale_from_amax(amax_bits(d1bound));
  const float inv1 = 1.f / (sx * sw1), inv3 = 1.f / (sd2 * sw2), inv4 = 1.f / (sd1 * sw1);
  u32x4v A1[2][2], A1T[2][2], A2T[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned q0, q1;
      split2h_pair_c(w1v[s][2 * i] * sw1, w1v[s][2 * i + 1] * sw1, q0, q1);
      A1[s][0][i] = q0; A1[s][1][i] = q1;
      split2h_pair_c(w1t[s][2 * i] * sw1, w1t[s][2 * i + 1] * sw1, q0, q1);
      A1T[s][0][i] = q0; A1T[s][1][i] = q1;
    }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    unsigned q0, q1;
    split2h_pair_c(w2t[2 * i] * sw2, w2t[2 * i + 1] * sw2, q0, q1);
    A2T[0][i] = q0; A2T[1][i] = q1;
    A2T[0][2 + i] = 0u; A2T[1][2 + i] = 0u;
  }
  const float lo = a.pro_relu ? 0.f : -INFINITY;
  __syncthreads();

  f32x16 accW1, accW2;
#pragma unroll
  for (int r = 0; r < 16; ++r) accW1[r] = accW2[r] = 0.f;
  float sb1 = 0.f, sb2 = 0.f;                   // this lane's row (n) of dh1 / dh2, summed over its pixels (scaled by sd1 / sd2)
  float sw3[HEAD_MAXNC][4], sb3[HEAD_MAXNC];
#pragma unroll
  for (int j = 0; j < HEAD_MAXNC; ++j) {
    sb3[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) sw3[j][r] = 0.f;
  }

  // per-image buffer descriptors: a lane's 16 channel rows differ by a scalar offset (the 64-bit addresses of 16 strided loads and 16
  // strided stores held ~60 registers); the next block's operands are fetched while this block runs
  const unsigned HW4 = (unsigned)HW * 4u;
  const int stride = gridDim.x * 4;
  float xn[16], h2n[4], dyn[L3 ? HEAD_MAXNC : 4];
  auto fetch = [&](int blk) {
    const int bq = min(blk, a.nblk - 1);          // past the end: a valid re-read, never used
    const int b = __builtin_amdgcn_readfirstlane(bq / bpi);
    const unsigned po = (unsigned)((bq - b * bpi) * 32 + n) * 4u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (size_t)b * 32 * HW, 32u * HW4);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) xn[8 * s + j] = buf_load(rx, po + (unsigned)(8 * u) * HW4, (unsigned)(16 * s + j) * HW4);
    if (L3) {
      const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.h2 + (size_t)b * 8 * HW, 8u * HW4);
      const __amdgpu_buffer_rsrc_t rd = make_rsrc(a.dy + (size_t)b * a.nc * HW, (unsigned)a.nc * HW4);
#pragma unroll
      for (int r = 0; r < 4; ++r) h2n[r] = buf_load(rh, po + (unsigned)(4 * u) * HW4, (unsigned)r * HW4);
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) dyn[j] = buf_load(rd, po, (unsigned)min(j, a.nc) * HW4);      // j >= nc: out of range, reads 0
    } else {
      const __amdgpu_buffer_rsrc_t rd = make_rsrc(a.dy + (size_t)b * 8 * HW, 8u * HW4);
#pragma unroll
      for (int r = 0; r < 4; ++r) dyn[r] = buf_load(rd, po + (unsigned)(4 * u) * HW4, (unsigned)r * HW4);
    }
  };
  fetch(blockIdx.x * 4 + wave);
  for (int blk = blockIdx.x * 4 + wave; blk < a.nblk; blk += stride) {
    const int b = __builtin_amdgcn_readfirstlane(blk / bpi);
    const unsigned po = (unsigned)((blk - b * bpi) * 32 + n) * 4u;
    float xs[16], d2[4];
#pragma unroll
    for (int e = 0; e < 16; ++e) xs[e] = xn[e];
    if (L3) {
      float h2v[4], dyv[HEAD_MAXNC];
#pragma unroll
      for (int r = 0; r < 4; ++r) h2v[r] = h2n[r];
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) dyv[j] = dyn[j];
      float w3r[HEAD_MAXNC][4];
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) {
        const float4 w = w3_sh[u * 4 + j];
        w3r[j][0] = w.x; w3r[j][1] = w.y; w3r[j][2] = w.z; w3r[j][3] = w.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float g = 0.f;
#pragma unroll
        for (int j = 0; j < HEAD_MAXNC; ++j) {
          g = fmaf(w3r[j][r], dyv[j], g);
          sw3[j][r] = fmaf(dyv[j], h2v[r], sw3[j][r]);
        }
        d2[r] = h2v[r] > 0.f ? g : 0.f;
      }
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) sb3[j] += u == 0 ? dyv[j] : 0.f;   // both half-waves hold the same pixel
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) d2[r] = dyn[r];
    }
    // dh1 before its mask: (W2^T dh2), one K step, rows 4 u + j of dh2 at K index 8 u + j
#pragma unroll
    for (int r = 0; r < 4; ++r) d2[r] *= sd2;
    u32x4v Bd2[2];
    {
      unsigned q0, q1;
      split2h_pair_c(d2[0], d2[1], q0, q1);
      Bd2[0][0] = q0; Bd2[1][0] = q1;
      split2h_pair_c(d2[2], d2[3], q0, q1);
      Bd2[0][1] = q0; Bd2[1][1] = q1;
      Bd2[0][2] = Bd2[0][3] = Bd2[1][2] = Bd2[1][3] = 0u;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Tdh2[(4 * u + r) * TS + n] = d2[r];
    f32x16 acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
    acc3 = mfma32h(A2T[0], Bd2[1], acc3);
    acc3 = mfma32h(A2T[1], Bd2[0], acc3);
    acc3 = mfma32h(A2T[0], Bd2[0], acc3);
    // activated input, times sx (channel c = 16 s + 8 u + j in xs[8 s + j]); h1 = relu(W1 x + b1) as head_fwd_h_k forms it
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    {
      u32x4v B1[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 8 * s + 2 * i;
          const float2 p0 = pro_sh[16 * s + 8 * u + 2 * i], p1 = pro_sh[16 * s + 8 * u + 2 * i + 1];
          const float v0 = fmaxf(fmaf(xs[e], p0.x * sx, p0.y * sx), lo), v1 = fmaxf(fmaf(xs[e + 1], p1.x * sx, p1.y * sx), lo);
          Tx[(16 * s + 8 * u + 2 * i) * TS + n] = v0;
          Tx[(16 * s + 8 * u + 2 * i + 1) * TS + n] = v1;
          unsigned q0, q1;
          split2h_pair_c(v0, v1, q0, q1);
          B1[s][0][i] = q0; B1[s][1][i] = q1;
        }
      fetch(blk + stride);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        acc1 = mfma32h(A1[s][0], B1[s][1], acc1);
        acc1 = mfma32h(A1[s][1], B1[s][0], acc1);
        acc1 = mfma32h(A1[s][0], B1[s][0], acc1);
      }
    }
    // dh1 = (W2^T dh2) * [h1 > 0], times sd1 — in acc3's registers; both go to the transpose tiles as they are formed
    const float k3 = inv3 * sd1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bq = b1_sh[u * 4 + q];
      const float bqv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * q + i;
        const float h = relu_keep_nan(fmaf(acc1[r], inv1, bqv[i]));
        Th1[rho(r, u) * TS + n] = h * sh;
        acc3[r] = h > 0.f ? acc3[r] * k3 : 0.f;
        Tdh1[rho(r, u) * TS + n] = acc3[r];
      }
    }
    // dx = W1^T dh1
    f32x16 acc4;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc4[r] = 0.f;
    {
      u32x4v Bd1[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned q0, q1;
          split2h_pair_c(acc3[8 * s + 2 * i], acc3[8 * s + 2 * i + 1], q0, q1);
          Bd1[s][0][i] = q0; Bd1[s][1][i] = q1;
        }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        acc4 = mfma32h(A1T[s][0], Bd1[s][1], acc4);
        acc4 = mfma32h(A1T[s][1], Bd1[s][0], acc4);
        acc4 = mfma32h(A1T[s][0], Bd1[s][0], acc4);
      }
    }
    {
      const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.dx + (size_t)b * 32 * HW, 32u * HW4);
#pragma unroll
      for (int r = 0; r < 16; ++r) buf_store(ro, po + (unsigned)(4 * u) * HW4, (unsigned)((r & 3) + 8 * (r >> 2)) * HW4, acc4[r] * inv4);
    }
    // weight gradients: pixels from lanes to the K dimension through the wave-private LDS tiles (scaled values); the bias gradients
    // are the row sums of the dh1 / dh2 tiles (one register each instead of sixteen / four per-lane sums)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto operand = [&](const float* tile, int s, u32x4v (&q)[2]) -> float {      // lane (row n, half u): pixels 16 s + 8 u .. + 7 of its row
      const float4* src = reinterpret_cast<const float4*>(tile + n * TS + 16 * s + 8 * u);
      const float4 v0 = src[0], v1 = src[1];
      unsigned q0, q1;
      split2h_pair_c(v0.x, v0.y, q0, q1); q[0][0] = q0; q[1][0] = q1;
      split2h_pair_c(v0.z, v0.w, q0, q1); q[0][1] = q0; q[1][1] = q1;
      split2h_pair_c(v1.x, v1.y, q0, q1); q[0][2] = q0; q[1][2] = q1;
      split2h_pair_c(v1.z, v1.w, q0, q1); q[0][3] = q0; q[1][3] = q1;
      return ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
    };
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4v qa[2], qb[2];
      sb1 += operand(Tdh1, s, qa);
      (void)operand(Tx, s, qb);
      accW1 = mfma32h(qa[0], qb[1], accW1);      // dW1[m][c] += dh1[m][px] x[c][px]
      accW1 = mfma32h(qa[1], qb[0], accW1);
      accW1 = mfma32h(qa[0], qb[0], accW1);
      sb2 += operand(Tdh2, s, qa);
      (void)operand(Th1, s, qb);
      accW2 = mfma32h(qa[0], qb[1], accW2);      // dW2[m][k] += dh2[m][px] h1[k][px]
      accW2 = mfma32h(qa[1], qb[0], accW2);
      accW2 = mfma32h(qa[0], qb[0], accW2);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const float invW1 = 1.f / (sd1 * sx), invW2 = 1.f / (sd2 * sh);

  // ---- per-lane sums over this lane's pixels -> sums over the 32 lanes of the half-wave
  sb1 = (sb1 + __shfl_xor(sb1, 32, 64)) * (1.f / sd1);      // both pixel halves of row n
  sb2 = (sb2 + __shfl_xor(sb2, 32, 64)) * (1.f / sd2);
#pragma unroll
  for (int m = 1; m <= 16; m <<= 1) {
    if (L3) {
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) {
        sb3[j] += __shfl_xor(sb3[j], m, 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) sw3[j][r] += __shfl_xor(sw3[j][r], m, 64);
      }
    }
  }
  // ---- cross-wave reduction through LDS (the transpose tiles are free now), then this workgroup's slab
  const int NS = head_ns(L3 ? a.nc : 0);
  __syncthreads();
  float* red = &T[0][0][0] + wave * 1408;      // 4 x 1408 floats <= sizeof(T); NS <= 1356
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    red[rho(r, u) * 32 + n] = accW1[r] * invW1;
    if (r < 4) red[1056 + (4 * u + r) * 32 + n] = accW2[r] * invW2;      // dW2 rows m = rho(r,u) = r + 4u for r < 4
  }
  if (u == 0) {
    red[1024 + n] = sb1;
    if (n < 8) red[1312 + n] = sb2;
  }
  if (n == 0) {
    if (L3) {
#pragma unroll
      for (int j = 0; j < HEAD_MAXNC; ++j) {
        if (j < a.nc) {
#pragma unroll
          for (int r = 0; r < 4; ++r) red[1320 + j * 8 + 4 * u + r] = sw3[j][r];
          if (u == 0) red[1320 + 8 * a.nc + j] = sb3[j];
        }
      }
    }
  }
  __syncthreads();
  const float* r0 = &T[0][0][0];
  for (int e = threadIdx.x; e < NS; e += 256)
    a.slab[(size_t)blockIdx.x * NS + e] = (r0[e] + r0[1408 + e]) + (r0[2816 + e] + r0[4224 + e]);
}

// out[i] = sum_k slab[k][i] in fp64, fixed order (bitwise reproducible); 32 outputs x 8 k-slices per workgroup
__global__ __launch_bounds__(256) void head_fold_k(const float* __restrict__ slab, int nslab, int n, float* __restrict__ out,
                                                   int accumulate) {
  __shared__ double sh[8][32];
  const int j = threadIdx.x & 31, kq = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + j;
  double s = 0.0;
  if (i < n)
    for (int k = kq; k < nslab; k += 8) s += (double)slab[(size_t)k * n + i];
  sh[kq][j] = s;
  __syncthreads();
  if (kq == 0 && i < n) {
    double t = sh[0][j] + sh[1][j] + sh[2][j] + sh[3][j] + sh[4][j] + sh[5][j] + sh[6][j] + sh[7][j];
    out[i] = accumulate ? out[i] + (float)t : (float)t;
  }
}

static int head_grid(int nblk) {
  int g = (nblk + 3) / 4;
  if (g > 512) g = 512;        // 2 workgroups per CU: every wave keeps its weights (and, backward, its dW accumulators)
  return g < 1 ? 1 : g;
}

extern "C" int wtpse_head_slabs(int B, int HW) { return head_grid(B * (HW / 32)); }

// See include/wtpse_hip.h for the contract.
extern "C" int wtpse_head_fwd(const float* x, const float* pro, int pro_relu, const float* w1, const float* b1, const float* w2,
                              const float* b2, const float* w3, const float* b3, int nc, float* h1, float* h2, float* y,
                              const unsigned* x_amax, int B, int HW, void* stream) {
  WTPSE_REQUIRE(x && w1 && b1 && w2 && b2 && B > 0 && HW > 0 && HW % 32 == 0);
  WTPSE_REQUIRE((w3 == nullptr) == (b3 == nullptr) && (w3 == nullptr) == (y == nullptr));
  WTPSE_REQUIRE(w3 ? (nc >= 1 && nc <= HEAD_MAXNC) : (h2 != nullptr));
  HeadArgs a = {};
  a.x = x; a.pro = pro; a.pro_relu = pro_relu; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.nc = w3 ? nc : 0;
  a.h1 = h1; a.h2 = h2; a.y = y; a.B = B; a.HW = HW; a.nblk = B * (HW / 32); a.x_amax = x_amax;
  dim3 grid((unsigned)head_grid(a.nblk));
  if (g_x3_terms == 2) {
    // (1 / 2 / 3 / 4 workgroups per CU: 103 / 89 / 93 / 90 us per [32,32,256,256] launch with the h2 tape)
    if (w3) hipLaunchKernelGGL(head_fwd_h_k<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(head_fwd_h_k<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  } else if (w3) hipLaunchKernelGGL(head_fwd_k<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(head_fwd_k<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return wtpse_status();
}

extern "C" int wtpse_head_bwd(const float* dy, const float* x, const float* pro, int pro_relu, const float* h1, const float* h2,
                              const float* w1, const float* b1, const float* w2, const float* w3, int nc, float* dx, float* slab,
                              float* dparams, int accumulate, const unsigned* x_amax, const unsigned* dy_amax, int B, int HW,
                              void* stream) {
  const bool x2h = g_x3_terms == 2;            // layer 1 formed again from x: needs b1 and the bound of dy, not the tape
  WTPSE_REQUIRE(dy && x && w1 && w2 && dx && slab && dparams && B > 0 && HW > 0 && HW % 32 == 0);
  WTPSE_REQUIRE(x2h ? (b1 != nullptr && dy_amax != nullptr) : (h1 != nullptr));
  WTPSE_REQUIRE(w3 ? (nc >= 1 && nc <= HEAD_MAXNC && h2 != nullptr) : true);
  HeadArgs a = {};
  a.x = x; a.pro = pro; a.pro_relu = pro_relu; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.w3 = w3; a.nc = w3 ? nc : 0;
  a.x_amax = x_amax; a.dy_amax = dy_amax;
  a.h1 = const_cast<float*>(h1); a.h2 = const_cast<float*>(h2); a.dy = dy; a.dx = dx; a.slab = slab;
  a.B = B; a.HW = HW; a.nblk = B * (HW / 32);
  const int g = head_grid(a.nblk);
  hipStream_t st = (hipStream_t)stream;
  if (x2h) {
    if (w3) hipLaunchKernelGGL(head_bwd_h_k<true>, dim3(g), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(head_bwd_h_k<false>, dim3(g), dim3(256), 0, st, a);
  } else if (w3) hipLaunchKernelGGL(head_bwd_k<true>, dim3(g), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(head_bwd_k<false>, dim3(g), dim3(256), 0, st, a);
  int rc = wtpse_status();
  if (rc) return rc;
  const int ns = head_ns(a.nc);
  hipLaunchKernelGGL(head_fold_k, dim3((ns + 31) / 32), dim3(256), 0, st, slab, g, ns, dparams, accumulate);
  return wtpse_status();
}
