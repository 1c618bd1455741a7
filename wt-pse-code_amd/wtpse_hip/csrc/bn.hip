// Train-/eval-mode BatchNorm2d pieces (nn.BatchNorm2d, eps 1e-5, momentum 0.1, affine — reference
// algorithms.py:862-864 via ConvD/ConvU/DoubleConv).  The convolution epilogue already produced per-workgroup
// (sum, sum of squares) partials; these kernels turn them into the per-channel (scale, shift) pair that
// consumers apply on load, keep the running statistics, and run the two-reduction backward.
//
//   forward : y = conv output, mean/var over (B,H,W);  z = act(y * scale + shift), scale = gamma*invstd, shift = beta - mean*scale
//   backward: dzh = dz * [z > 0];  dbeta = sum dzh;  dgamma = sum dzh * xhat;  dy = k1*dzh + k2*y + k3
//             k1 = gamma*invstd,  k2 = -gamma*invstd^2*dgamma/N,  k3 = -k1*dbeta/N - k2*mean
#include "common.h"

// one workgroup per channel; partial sums are folded in fp64 in a fixed order.  NT threads: 1024 for the launches with thousands of
// partial rows (the 8192-tile convolutions, the upsampling kernels) — every row is then read in ONE round of eight loads per thread:
// the launch sits between two convolutions of a dependency chain (54 + 40 times per step with its backward twin) and its 9 us were
// four rounds of strided loads that miss this XCD's L2 (the partials were written by workgroups all over the chip)
template <int NT>
__global__ __launch_bounds__(NT) void bn_finalize_k(const float* __restrict__ part, int nblk, int C, double count,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ rmean, float* __restrict__ rvar,
                                                    long long* __restrict__ nbt, float momentum, float eps,
                                                    float* __restrict__ scale_shift, float* __restrict__ save_mean,
                                                    float* __restrict__ save_invstd, unsigned* __restrict__ act_amax) {
  __shared__ double shs[2][NT / 64];
  const int c = blockIdx.x, t = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int k0 = t; k0 < nblk; k0 += NT * 8) {
    float2 p2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + NT * u;
      p2[u] = k < nblk ? *reinterpret_cast<const float2*>(part + ((size_t)k * C + c) * 2) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s1 += p2[u].x;
      s2 += p2[u].y;
    }
  }
  for (int m = 1; m < 64; m <<= 1) {
    s1 += __shfl_xor(s1, m, 64);
    s2 += __shfl_xor(s2, m, 64);
  }
  if ((t & 63) == 0) {
    shs[0][t >> 6] = s1;
    shs[1][t >> 6] = s2;
  }
  __syncthreads();
  if (t == 0) {
    s1 = s2 = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
      s1 += shs[0][w];
      s2 += shs[1][w];
    }
    double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    double invstd = 1.0 / sqrt(var + (double)eps);
    float sc = (float)(gamma[c] * invstd);
    scale_shift[2 * c] = sc;
    scale_shift[2 * c + 1] = (float)(beta[c] - mean * gamma[c] * invstd);
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)invstd;
    if (act_amax) amax_put(act_amax, bn_act_bound(gamma[c], beta[c], count), (unsigned)c);     // common.h: the consumers' x2h input scale
    if (rmean) {
      double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
      rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unbiased);
    }
    if (nbt && c == 0) *nbt += 1;
  }
}

__global__ void bn_eval_coeffs_k(const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, int C,
                                 float* __restrict__ scale_shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float invstd = 1.f / sqrtf(rvar[c] + eps);
  float sc = gamma[c] * invstd;
  scale_shift[2 * c] = sc;
  scale_shift[2 * c + 1] = beta[c] - rmean[c] * sc;
}

// Bound of |scale[c] * y + shift[c]| over a tensor y whose largest magnitude is known (raw_amax: the amax table its producer's epilogue
// or wtpse_amax filled): max_c |scale[c]| * amax + |shift[c]| — the x2h input scale of an activation behind an EVAL-mode BatchNorm
// (running statistics say nothing about this batch) or behind any other per-channel affine map.  One workgroup; writes the whole
// table (it need not be zero on entry).
__global__ __launch_bounds__(256) void act_bound_k(const float* __restrict__ ss, int C, const unsigned* __restrict__ raw_amax,
                                                   unsigned* __restrict__ act_amax) {
  __shared__ float red[4];
  const float ymax = __builtin_bit_cast(float, amax_read(raw_amax));
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, fabsf(ss[2 * c]) * ymax + fabsf(ss[2 * c + 1]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < AMAX_SHARDS) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    act_amax[threadIdx.x * AMAX_STRIDE] = threadIdx.x == 0 ? amax_bits(m) : 0u;
  }
}

// z = act(y*scale[c] + shift[c]); grid = (blocks over HW, B*C)
template <bool VEC>
__global__ __launch_bounds__(256) void affine_act_k(const float* __restrict__ y, const float* __restrict__ ss, int relu,
                                                    int C, int HW, int bpp, float* __restrict__ z) {
  const int bc = blockIdx.x / bpp, blk = blockIdx.x - bc * bpp, c = bc % C;
  const float sc = ss ? ss[2 * c] : 1.f, sh = ss ? ss[2 * c + 1] : 0.f;
  const float lo = relu ? 0.f : -INFINITY;
  const size_t base = (size_t)bc * HW;
  if (VEC) {
    int p = (blk * 256 + threadIdx.x) * 4;
    if (p >= HW) return;
    float4 v = *reinterpret_cast<const float4*>(y + base + p);
    v.x = fmaxf(fmaf(v.x, sc, sh), lo); v.y = fmaxf(fmaf(v.y, sc, sh), lo);
    v.z = fmaxf(fmaf(v.z, sc, sh), lo); v.w = fmaxf(fmaf(v.w, sc, sh), lo);
    *reinterpret_cast<float4*>(z + base + p) = v;
  } else {
    int p = blk * 256 + threadIdx.x;
    if (p >= HW) return;
    z[base + p] = fmaxf(fmaf(y[base + p], sc, sh), lo);
  }
}

// partial[(split*C + c)*2 + {0,1}] = sum over this split's elements of (dzh, dzh*xhat); grid = (C, nsplit).
// A split owns whole (image, segment) units — an image's channel plane in `segs` equal pieces (u = split, split + nsplit, ...): no
// per-element division, float4 streams, two of them in flight per lane.  (Whole images only: 16 channels x 32 images gave 512
// workgroups, two per CU, and 0.41 of the copy rate.)
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_k(const float* __restrict__ dz, const float* __restrict__ y,
                                                       const float* __restrict__ ss, int relu,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       int B, int C, int HW, int segs, float* __restrict__ partial) {
  __shared__ float sh[2][4];
  const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const float sc = ss[2 * c], sf = ss[2 * c + 1], mu = mean[c], is = invstd[c];
  const int SL = HW / segs;                        // elements of a segment (segs > 1: a multiple of 2048)
  float s1 = 0.f, s2 = 0.f;
  auto fold4 = [&](float4 yv, float4 g) __attribute__((always_inline)) {
    if (relu) {
      if (!(fmaf(yv.x, sc, sf) > 0.f)) g.x = 0.f;
      if (!(fmaf(yv.y, sc, sf) > 0.f)) g.y = 0.f;
      if (!(fmaf(yv.z, sc, sf) > 0.f)) g.z = 0.f;
      if (!(fmaf(yv.w, sc, sf) > 0.f)) g.w = 0.f;
    }
    s1 += (g.x + g.y) + (g.z + g.w);
    s2 += (g.x * (yv.x - mu) + g.y * (yv.y - mu)) + (g.z * (yv.z - mu) + g.w * (yv.w - mu));
  };
  for (int u = split; u < B * segs; u += nsplit) {
    const int b = u / segs, sg = u - b * segs;
    const size_t base = ((size_t)b * C + c) * HW + (size_t)sg * SL;
    if (VEC) {
      int p = threadIdx.x * 4;
      for (; p + 1024 < SL; p += 2048) {
        const float4 y0 = *reinterpret_cast<const float4*>(y + base + p), y1 = *reinterpret_cast<const float4*>(y + base + p + 1024);
        const float4 g0 = *reinterpret_cast<const float4*>(dz + base + p), g1 = *reinterpret_cast<const float4*>(dz + base + p + 1024);
        fold4(y0, g0);
        fold4(y1, g1);
      }
      if (p < SL) fold4(*reinterpret_cast<const float4*>(y + base + p), *reinterpret_cast<const float4*>(dz + base + p));
    } else {
      for (int p = threadIdx.x; p < SL; p += 256) {
        float yv = y[base + p], g = dz[base + p];
        if (relu && !(fmaf(yv, sc, sf) > 0.f)) g = 0.f;
        s1 += g;
        s2 += g * (yv - mu);
      }
    }
  }
  s2 *= is;
  s1 = wave_xor_sum(s1, 32);
  s2 = wave_xor_sum(s2, 32);
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = s1;
    sh[1][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[((size_t)split * C + c) * 2] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    partial[((size_t)split * C + c) * 2 + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

// `global_sums` (optional, [C][2]): the same two sums over ALL ranks' elements (synchronised BatchNorm); the dy
// coefficients then use them together with the global `count`, while dgamma/dbeta keep this rank's share (the
// parameter-gradient all-reduce adds the shares up).  `sums_out` (optional) receives this rank's folded sums.
template <int NT>
__global__ __launch_bounds__(NT) void bn_bwd_finalize_k(const float* __restrict__ partial, int nsplit, int C, double count,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int accumulate,
                                                         float* __restrict__ coef, const float* __restrict__ global_sums,
                                                         float* __restrict__ sums_out, int centred_s2) {
  // one workgroup per channel; fp64 fold in a fixed order (lane-strided, butterfly, the waves in order); NT: see bn_finalize_k
  __shared__ double shs[2][NT / 64];
  const int c = blockIdx.x, t = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int k0 = t; k0 < nsplit; k0 += NT * 8) {
    float2 p2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + NT * u;
      p2[u] = k < nsplit ? *reinterpret_cast<const float2*>(partial + ((size_t)k * C + c) * 2) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s1 += p2[u].x;
      s2 += p2[u].y;
    }
  }
  for (int m = 1; m < 64; m <<= 1) {
    s1 += __shfl_xor(s1, m, 64);
    s2 += __shfl_xor(s2, m, 64);
  }
  if ((t & 63) == 0) {
    shs[0][t >> 6] = s1;
    shs[1][t >> 6] = s2;
  }
  __syncthreads();
  s1 = s2 = 0.0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    s1 += shs[0][w];
    s2 += shs[1][w];
  }
  if (centred_s2) s2 *= (double)invstd[c];     // partials of a data-gradient epilogue: sum g * (y - mean), not yet / std
  if (t == 0) {
    if (sums_out) {
      sums_out[2 * c] = (float)s1;
      sums_out[2 * c + 1] = (float)s2;
    }
    if (dbeta) {
      dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
      dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
    }
    if (!coef) return;
    if (global_sums) {
      s1 = global_sums[2 * c];
      s2 = global_sums[2 * c + 1];
    }
    double k1 = (double)gamma[c] * invstd[c];
    double k2 = -(double)gamma[c] * invstd[c] * invstd[c] * s2 / count;
    double k3 = -k1 * s1 / count - k2 * mean[c];
    coef[3 * c] = (float)k1;
    coef[3 * c + 1] = (float)k2;
    coef[3 * c + 2] = (float)k3;
  }
}

// dy = k1 * dz*[z>0] + k2*y + k3 ; one item (plane bc, chunk of 1024 | 256 elements) per workgroup.  amax (optional): the amax table
// of dy (common.h; zero on entry) — dy is the gradient operand of the x2h convolutions that consume it; every wave folds its maximum
// into a shard with one no-return atomic.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_k(const float* __restrict__ dz, const float* __restrict__ y,
                                                      const float* __restrict__ ss, int relu,
                                                      const float* __restrict__ coef, int C, int HW, int bpp,
                                                      float* __restrict__ dy, unsigned* __restrict__ amax) {
  const int bc = blockIdx.x / bpp, blk = blockIdx.x - bc * bpp, c = bc % C;
  const float sc = ss[2 * c], sf = ss[2 * c + 1];
  const float k1 = coef[3 * c], k2 = coef[3 * c + 1], k3 = coef[3 * c + 2];
  const size_t base = (size_t)bc * HW;
  unsigned am = 0u;
  if (VEC) {
    int p = (blk * 256 + threadIdx.x) * 4;
    if (p < HW) {
      float4 g = *reinterpret_cast<const float4*>(dz + base + p);
      float4 v = *reinterpret_cast<const float4*>(y + base + p);
      float4 o;
      o.x = fmaf(k1, (relu && !(fmaf(v.x, sc, sf) > 0.f)) ? 0.f : g.x, fmaf(k2, v.x, k3));
      o.y = fmaf(k1, (relu && !(fmaf(v.y, sc, sf) > 0.f)) ? 0.f : g.y, fmaf(k2, v.y, k3));
      o.z = fmaf(k1, (relu && !(fmaf(v.z, sc, sf) > 0.f)) ? 0.f : g.z, fmaf(k2, v.z, k3));
      o.w = fmaf(k1, (relu && !(fmaf(v.w, sc, sf) > 0.f)) ? 0.f : g.w, fmaf(k2, v.w, k3));
      *reinterpret_cast<float4*>(dy + base + p) = o;
      am = max(max(amax_bits(o.x), amax_bits(o.y)), max(amax_bits(o.z), amax_bits(o.w)));
    }
  } else {
    int p = blk * 256 + threadIdx.x;
    if (p < HW) {
      float g = dz[base + p], v = y[base + p];
      if (relu && !(fmaf(v, sc, sf) > 0.f)) g = 0.f;
      const float o = fmaf(k1, g, fmaf(k2, v, k3));
      dy[base + p] = o;
      am = amax_bits(o);
    }
  }
  if (amax) amax_publish_wave(amax, am, blockIdx.x * 4u + (threadIdx.x >> 6));
}

// Small maps (the 16x16 / 32x32 levels): the three launches above cost 23-30 us of mostly launch latency for a few MB.
// One workgroup per channel does all of it: pass 1 folds (sum dzh, sum dzh*xhat) over the channel's B*HW elements, the
// coefficients are computed in place (same formulas as bn_bwd_finalize_k), pass 2 re-reads the channel (L2 / Infinity
// Cache resident) and writes dy.  HW % 4 == 0, 16-byte aligned tensors.
__global__ __launch_bounds__(1024) void bn_bwd_small_k(const float* __restrict__ dz, const float* __restrict__ y,
                                                       const float* __restrict__ ss, int relu, const float* __restrict__ gamma,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                       float* __restrict__ dy, int B, int C, int HW, unsigned* __restrict__ amax) {
  __shared__ double sh[2][16];
  __shared__ float kc[3];
  const int c = blockIdx.x, t = threadIdx.x;
  const float sc = ss[2 * c], sf = ss[2 * c + 1], mu = mean[c], is = invstd[c];
  const int q = HW / 4, total = B * q;            // float4 pieces of this channel
  float s1 = 0.f, s2 = 0.f;
  for (int e = t; e < total; e += 1024) {
    const int b = e / q, p = (e - b * q) * 4;
    const size_t off = ((size_t)b * C + c) * HW + p;
    const float4 yv = *reinterpret_cast<const float4*>(y + off);
    float4 g = *reinterpret_cast<const float4*>(dz + off);
    if (relu) {
      if (!(fmaf(yv.x, sc, sf) > 0.f)) g.x = 0.f;
      if (!(fmaf(yv.y, sc, sf) > 0.f)) g.y = 0.f;
      if (!(fmaf(yv.z, sc, sf) > 0.f)) g.z = 0.f;
      if (!(fmaf(yv.w, sc, sf) > 0.f)) g.w = 0.f;
    }
    s1 += (g.x + g.y) + (g.z + g.w);
    s2 += (g.x * (yv.x - mu) + g.y * (yv.y - mu)) + (g.z * (yv.z - mu) + g.w * (yv.w - mu));
  }
  double d1 = (double)s1, d2 = (double)s2 * (double)is;
  for (int m = 1; m < 64; m <<= 1) {
    d1 += __shfl_xor(d1, m, 64);
    d2 += __shfl_xor(d2, m, 64);
  }
  if ((t & 63) == 0) {
    sh[0][t >> 6] = d1;
    sh[1][t >> 6] = d2;
  }
  __syncthreads();
  if (t == 0) {
    double a1 = 0.0, a2 = 0.0;
    for (int w = 0; w < 16; ++w) {
      a1 += sh[0][w];
      a2 += sh[1][w];
    }
    dbeta[c] = accumulate ? dbeta[c] + (float)a1 : (float)a1;
    dgamma[c] = accumulate ? dgamma[c] + (float)a2 : (float)a2;
    const double count = (double)B * HW;
    const double k1 = (double)gamma[c] * is;
    const double k2 = -(double)gamma[c] * is * is * a2 / count;
    const double k3 = -k1 * a1 / count - k2 * mu;
    kc[0] = (float)k1;
    kc[1] = (float)k2;
    kc[2] = (float)k3;
  }
  __syncthreads();
  const float k1 = kc[0], k2 = kc[1], k3 = kc[2];
  unsigned am = 0u;
  for (int e = t; e < total; e += 1024) {
    const int b = e / q, p = (e - b * q) * 4;
    const size_t off = ((size_t)b * C + c) * HW + p;
    const float4 v = *reinterpret_cast<const float4*>(y + off);
    const float4 g = *reinterpret_cast<const float4*>(dz + off);
    float4 o;
    o.x = fmaf(k1, (relu && !(fmaf(v.x, sc, sf) > 0.f)) ? 0.f : g.x, fmaf(k2, v.x, k3));
    o.y = fmaf(k1, (relu && !(fmaf(v.y, sc, sf) > 0.f)) ? 0.f : g.y, fmaf(k2, v.y, k3));
    o.z = fmaf(k1, (relu && !(fmaf(v.z, sc, sf) > 0.f)) ? 0.f : g.z, fmaf(k2, v.z, k3));
    o.w = fmaf(k1, (relu && !(fmaf(v.w, sc, sf) > 0.f)) ? 0.f : g.w, fmaf(k2, v.w, k3));
    *reinterpret_cast<float4*>(dy + off) = o;
    am = max(am, max(max(amax_bits(o.x), amax_bits(o.y)), max(amax_bits(o.z), amax_bits(o.w))));
  }
  if (amax) amax_publish_block(amax, am, blockIdx.x);
}

static void bwd_finalize(hipStream_t st, const float* partial, int nsplit, int C, double count, const float* gamma, const float* mean,
                         const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coef, const float* global_sums,
                         float* sums_out, int centred_s2) {
  if (nsplit >= 2048)
    hipLaunchKernelGGL(bn_bwd_finalize_k<1024>, dim3(C), dim3(1024), 0, st, partial, nsplit, C, count, gamma, mean, invstd, dgamma, dbeta,
                       accumulate, coef, global_sums, sums_out, centred_s2);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_k<256>, dim3(C), dim3(256), 0, st, partial, nsplit, C, count, gamma, mean, invstd, dgamma, dbeta,
                       accumulate, coef, global_sums, sums_out, centred_s2);
}

static inline bool vec_ok(int HW, const void* a, const void* b, const void* c) {
  return HW % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

// the apply pass dy = k1 [masked] dz + k2 y + k3 (+ the amax table of dy, zero on entry, or null)
static void launch_apply(const float* dz, const float* y, const float* ss, int relu, const float* coef, float* dy, int B, int C, int HW,
                         unsigned* amax, hipStream_t st) {
  if (vec_ok(HW, dz, y, dy))
    hipLaunchKernelGGL(bn_bwd_apply_k<true>, dim3(ceil_div(HW, 1024) * B * C), dim3(256), 0, st, dz, y, ss, relu, coef, C, HW,
                       ceil_div(HW, 1024), dy, amax);
  else
    hipLaunchKernelGGL(bn_bwd_apply_k<false>, dim3(ceil_div(HW, 256) * B * C), dim3(256), 0, st, dz, y, ss, relu, coef, C, HW,
                       ceil_div(HW, 256), dy, amax);
}

extern "C" int wtpse_bn_finalize(const float* stats_partial, int nblk, int C, long long count, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, long long* num_batches,
                                 float momentum, float eps, float* scale_shift, float* save_mean, float* save_invstd,
                                 unsigned* act_amax, void* stream) {
  WTPSE_REQUIRE(stats_partial && gamma && beta && scale_shift && save_mean && save_invstd && nblk > 0 && C > 0 && count > 0);
  WTPSE_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
  if (nblk >= 2048)
    hipLaunchKernelGGL(bn_finalize_k<1024>, dim3(C), dim3(1024), 0, (hipStream_t)stream, stats_partial, nblk, C, (double)count,
                       gamma, beta, running_mean, running_var, num_batches, momentum, eps, scale_shift, save_mean,
                       save_invstd, act_amax);
  else
    hipLaunchKernelGGL(bn_finalize_k<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, stats_partial, nblk, C, (double)count,
                       gamma, beta, running_mean, running_var, num_batches, momentum, eps, scale_shift, save_mean,
                       save_invstd, act_amax);
  return wtpse_status();
}

extern "C" int wtpse_act_bound(const float* scale_shift, int C, const unsigned* raw_amax, unsigned* act_amax, void* stream) {
  WTPSE_REQUIRE(scale_shift && raw_amax && act_amax && C > 0);
  hipLaunchKernelGGL(act_bound_k, dim3(1), dim3(256), 0, (hipStream_t)stream, scale_shift, C, raw_amax, act_amax);
  return wtpse_status();
}

extern "C" int wtpse_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, int C, float* scale_shift, void* stream) {
  WTPSE_REQUIRE(gamma && beta && running_mean && running_var && scale_shift && C > 0);
  hipLaunchKernelGGL(bn_eval_coeffs_k, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, scale_shift);
  return wtpse_status();
}

extern "C" int wtpse_affine_act(const float* y, const float* scale_shift, int relu, float* z, int B, int C, int HW,
                                void* stream) {
  WTPSE_REQUIRE(y && z && B > 0 && C > 0 && HW > 0);
  hipStream_t st = (hipStream_t)stream;
  if (vec_ok(HW, y, z, nullptr))
    hipLaunchKernelGGL(affine_act_k<true>, dim3(ceil_div(HW, 1024) * B * C), dim3(256), 0, st, y, scale_shift, relu, C, HW, ceil_div(HW, 1024), z);
  else
    hipLaunchKernelGGL(affine_act_k<false>, dim3(ceil_div(HW, 256) * B * C), dim3(256), 0, st, y, scale_shift, relu, C, HW, ceil_div(HW, 256), z);
  return wtpse_status();
}

// segments per image plane: doubled while the launch stays within ~2048 workgroups and a segment stays a multiple of 2048 elements
static int bn_bwd_segs(int B, int C, int HW) {
  int segs = 1;
  const int cap = 2048 / (C > 0 ? C : 1);
  while (B * segs * 2 <= cap && HW % (segs * 2 * 2048) == 0 && HW / (segs * 2) >= 8192) segs *= 2;
  return segs;
}
extern "C" int wtpse_bn_bwd_nsplit(int B, int C, int HW) {
  int ns = 2048 / (C > 0 ? C : 1);   // ~2048 workgroups; a split owns whole (image, segment) units
  const int units = B * bn_bwd_segs(B, C, HW);
  if (ns > units) ns = units;
  if (ns < 1) ns = 1;
  return ns;
}

extern "C" int wtpse_bn_bwd(const float* dz, const float* y, const float* scale_shift, int relu, const float* gamma,
                            const float* save_mean, const float* save_invstd, float* partial, float* coef,
                            float* dgamma, float* dbeta, int accumulate, float* dy, int B, int C, int HW,
                            unsigned* amax, void* stream) {
  WTPSE_REQUIRE(dz && y && scale_shift && gamma && save_mean && save_invstd && partial && coef && dgamma && dbeta && dy);
  WTPSE_REQUIRE(B > 0 && C > 0 && HW > 0);
  hipStream_t st = (hipStream_t)stream;
  // a channel of at most 32k elements with enough channels to occupy the chip: everything in one launch
  if ((long long)B * HW <= 32768 && C >= 96 && vec_ok(HW, dz, y, dy)) {
    hipLaunchKernelGGL(bn_bwd_small_k, dim3(C), dim3(1024), 0, st, dz, y, scale_shift, relu, gamma, save_mean, save_invstd,
                       dgamma, dbeta, accumulate, dy, B, C, HW, amax);
    return wtpse_status();
  }
  const int ns = wtpse_bn_bwd_nsplit(B, C, HW);
  if (vec_ok(HW, dz, y, nullptr))
    hipLaunchKernelGGL(bn_bwd_reduce_k<true>, dim3(C, ns), dim3(256), 0, st, dz, y, scale_shift, relu, save_mean,
                       save_invstd, B, C, HW, bn_bwd_segs(B, C, HW), partial);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_k<false>, dim3(C, ns), dim3(256), 0, st, dz, y, scale_shift, relu, save_mean,
                       save_invstd, B, C, HW, bn_bwd_segs(B, C, HW), partial);
  bwd_finalize(st, partial, ns, C, (double)B * HW, gamma, save_mean,
                     save_invstd, dgamma, dbeta, accumulate, coef, (const float*)nullptr, (float*)nullptr, 0);
  launch_apply(dz, y, scale_shift, relu, coef, dy, B, C, HW, amax, st);
  return wtpse_status();
}

// Synchronised BatchNorm backward in two halves around the caller's all-reduce of `sums` ([C][2]):
//   wtpse_bn_bwd_reduce : this rank's (sum dzh, sum dzh*xhat) -> sums_local
//   wtpse_bn_bwd_apply  : dgamma/dbeta from sums_local, dy from sums_global and the global element count
extern "C" int wtpse_bn_bwd_reduce(const float* dz, const float* y, const float* scale_shift, int relu,
                                   const float* save_mean, const float* save_invstd, float* partial, float* sums_local,
                                   int B, int C, int HW, void* stream) {
  WTPSE_REQUIRE(dz && y && scale_shift && save_mean && save_invstd && partial && sums_local && B > 0 && C > 0 && HW > 0);
  hipStream_t st = (hipStream_t)stream;
  const int ns = wtpse_bn_bwd_nsplit(B, C, HW);
  if (vec_ok(HW, dz, y, nullptr))
    hipLaunchKernelGGL(bn_bwd_reduce_k<true>, dim3(C, ns), dim3(256), 0, st, dz, y, scale_shift, relu, save_mean,
                       save_invstd, B, C, HW, bn_bwd_segs(B, C, HW), partial);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_k<false>, dim3(C, ns), dim3(256), 0, st, dz, y, scale_shift, relu, save_mean,
                       save_invstd, B, C, HW, bn_bwd_segs(B, C, HW), partial);
  bwd_finalize(st, partial, ns, C, 1.0, (const float*)nullptr, save_mean,
                     save_invstd, (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, (const float*)nullptr, sums_local, 0);
  return wtpse_status();
}

extern "C" int wtpse_bn_bwd_apply(const float* dz, const float* y, const float* scale_shift, int relu, const float* gamma,
                                  const float* save_mean, const float* save_invstd, const float* sums_local,
                                  const float* sums_global, long long count_global, float* coef, float* dgamma,
                                  float* dbeta, int accumulate, float* dy, int B, int C, int HW, unsigned* amax, void* stream) {
  WTPSE_REQUIRE(dz && y && scale_shift && gamma && save_mean && save_invstd && sums_local && sums_global && coef && dgamma &&
                dbeta && dy && B > 0 && C > 0 && HW > 0 && count_global > 0);
  hipStream_t st = (hipStream_t)stream;
  // sums_local viewed as a 1-slab partial: [1][C][2]
  bwd_finalize(st, sums_local, 1, C, (double)count_global, gamma, save_mean,
                     save_invstd, dgamma, dbeta, accumulate, coef, sums_global, (float*)nullptr, 0);
  launch_apply(dz, y, scale_shift, relu, coef, dy, B, C, HW, amax, st);
  return wtpse_status();
}

// Second half of a BatchNorm backward whose reductions were formed in the epilogue of the data gradient that produced the
// incoming gradient (wtpse_dgrad_bnb / wtpse_dgrad_x3_bnb): g = dz * [z > 0] already masked, stats_partial[nblk][C][2] =
// per-workgroup (sum g, sum g * (y - mean)).  dgamma / dbeta, then dy = k1 * g + k2 * y + k3 in one elementwise pass.
extern "C" int wtpse_bn_bwd_from_stats(const float* g, const float* y, const float* stats_partial, int nblk, const float* gamma,
                                       const float* save_mean, const float* save_invstd, float* coef, float* dgamma, float* dbeta,
                                       int accumulate, float* dy, int B, int C, int HW, unsigned* amax, void* stream) {
  WTPSE_REQUIRE(g && y && stats_partial && gamma && save_mean && save_invstd && coef && dgamma && dbeta && dy);
  WTPSE_REQUIRE(nblk > 0 && B > 0 && C > 0 && HW > 0);
  hipStream_t st = (hipStream_t)stream;
  bwd_finalize(st, stats_partial, nblk, C, (double)B * HW, gamma, save_mean,
                     save_invstd, dgamma, dbeta, accumulate, coef, (const float*)nullptr, (float*)nullptr, 1);
  // (the scale/shift operand is only read for the ReLU mask: relu = 0 here, the coefficients stand in for it)
  launch_apply(g, y, coef, 0, coef, dy, B, C, HW, amax, st);
  return wtpse_status();
}

// The first launch of wtpse_bn_bwd_from_stats on its own: partials (sum g, sum g (y - mean)) -> coef (k1, k2, k3), dgamma, dbeta (+)=.
// (wtpse_dgrad_bnb_coef uses it for launches with so many workgroups that folding inside the launch costs more than this.)
extern "C" int wtpse_bn_bwd_finalize_coef(const float* stats_partial, int nblk, int C, long long count, const float* gamma,
                                          const float* save_mean, const float* save_invstd, float* coef, float* dgamma,
                                          float* dbeta, int accumulate, void* stream) {
  WTPSE_REQUIRE(stats_partial && gamma && save_mean && save_invstd && coef && dgamma && dbeta && nblk > 0 && C > 0 && count > 0);
  bwd_finalize((hipStream_t)stream, stats_partial, nblk, C, (double)count, gamma,
                     save_mean, save_invstd, dgamma, dbeta, accumulate, coef, (const float*)nullptr, (float*)nullptr, 1);
  return wtpse_status();
}

// dy = k1 * g + k2 * y + k3 with the coefficients a data gradient's tail left in `coef` (wtpse_dgrad_bnb_coef): the whole
// BatchNorm backward that remains once the reductions and their fold happened in the producing launch.
extern "C" int wtpse_bn_bwd_apply_coef(const float* g, const float* y, const float* coef, float* dy, int B, int C, int HW,
                                       unsigned* amax, void* stream) {
  WTPSE_REQUIRE(g && y && coef && dy && B > 0 && C > 0 && HW > 0);
  hipStream_t st = (hipStream_t)stream;
  launch_apply(g, y, coef, 0, coef, dy, B, C, HW, amax, st);
  return wtpse_status();
}
