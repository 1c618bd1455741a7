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
                              const float* b2, const float* w3, const float* b3, int nc, float* h1, float* h2, float* y, int B,
                              int HW, void* stream) {
  WTPSE_REQUIRE(x && w1 && b1 && w2 && b2 && B > 0 && HW > 0 && HW % 32 == 0);
  WTPSE_REQUIRE((w3 == nullptr) == (b3 == nullptr) && (w3 == nullptr) == (y == nullptr));
  WTPSE_REQUIRE(w3 ? (nc >= 1 && nc <= HEAD_MAXNC) : (h2 != nullptr));
  HeadArgs a = {};
  a.x = x; a.pro = pro; a.pro_relu = pro_relu; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.nc = w3 ? nc : 0;
  a.h1 = h1; a.h2 = h2; a.y = y; a.B = B; a.HW = HW; a.nblk = B * (HW / 32);
  dim3 grid((unsigned)head_grid(a.nblk));
  if (w3) hipLaunchKernelGGL(head_fwd_k<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(head_fwd_k<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return wtpse_status();
}

extern "C" int wtpse_head_bwd(const float* dy, const float* x, const float* pro, int pro_relu, const float* h1, const float* h2,
                              const float* w1, const float* w2, const float* w3, int nc, float* dx, float* slab, float* dparams,
                              int accumulate, int B, int HW, void* stream) {
  WTPSE_REQUIRE(dy && x && h1 && w1 && w2 && dx && slab && dparams && B > 0 && HW > 0 && HW % 32 == 0);
  WTPSE_REQUIRE(w3 ? (nc >= 1 && nc <= HEAD_MAXNC && h2 != nullptr) : true);
  HeadArgs a = {};
  a.x = x; a.pro = pro; a.pro_relu = pro_relu; a.w1 = w1; a.w2 = w2; a.w3 = w3; a.nc = w3 ? nc : 0;
  a.h1 = const_cast<float*>(h1); a.h2 = const_cast<float*>(h2); a.dy = dy; a.dx = dx; a.slab = slab;
  a.B = B; a.HW = HW; a.nblk = B * (HW / 32);
  const int g = head_grid(a.nblk);
  hipStream_t st = (hipStream_t)stream;
  if (w3) hipLaunchKernelGGL(head_bwd_k<true>, dim3(g), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(head_bwd_k<false>, dim3(g), dim3(256), 0, st, a);
  int rc = wtpse_status();
  if (rc) return rc;
  const int ns = head_ns(a.nc);
  hipLaunchKernelGGL(head_fold_k, dim3((ns + 31) / 32), dim3(256), 0, st, slab, g, ns, dparams, accumulate);
  return wtpse_status();
}
