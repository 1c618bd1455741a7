// Bandwidth-bound pieces of the WT-PSE step that are not convolutions: pooling / upsampling of the U-Net
// blocks (algorithms.py:890,901,929,949), the shape-attention fusion (:1126-1129,1243-1248), the
// reparameterisations (:1068-1075; shape_networks.py:502-510), the caller's losses (Trainer.py:787,842-871;
// shape_networks.py:596-597), Adam (train.py:120-138), the NaN scrub (shape_networks.py:490-506) and a
// Philox normal generator for the sampling noise.  All NCHW fp32.
#include "common.h"

__device__ __forceinline__ float act_in(float v, const float* pro, int c, int relu) {
  if (pro) v = fmaf(v, pro[2 * c], pro[2 * c + 1]);
  return relu ? fmaxf(v, 0.f) : v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float block_sum(float v, float* sh4) {  // 256 threads
  v = wave_xor_sum(v, 32);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

// ------------------------------------------------------------------------------------------------ MaxPool2d(2)
__global__ __launch_bounds__(256) void maxpool2_fwd_k(const float* __restrict__ x, const float* __restrict__ pro, int relu,
                                                      float* __restrict__ out, int C, int H, int W, long long total) {
  const int Ho = H / 2, Wo = W / 2;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int xo = (int)(i % Wo);
  long long r = i / Wo;
  int yo = (int)(r % Ho);
  long long bc = r / Ho;
  int c = (int)(bc % C);
  const float* src = x + (size_t)bc * H * W + (size_t)(2 * yo) * W + 2 * xo;
  float a = act_in(src[0], pro, c, relu), b = act_in(src[1], pro, c, relu);
  float d = act_in(src[W], pro, c, relu), e = act_in(src[W + 1], pro, c, relu);
  out[i] = fmaxf(fmaxf(a, b), fmaxf(d, e));
}

// Plane-per-blockIdx.y variants of the hot pooling / upsampling kernels: 32-bit index arithmetic inside a plane (the
// 64-bit div/mod chains of the flat kernels cost more than the memory traffic) and 16-byte accesses.  Used when the
// row length allows it; the flat kernels above/below remain for odd shapes.  Same expression trees, bitwise equal.
__global__ __launch_bounds__(256) void maxpool2_fwd_v_k(const float* __restrict__ x, const float* __restrict__ pro, int relu,
                                                        float* __restrict__ out, int BC, int C, int H, int W) {
  const int Ho = H / 2, Wo = W / 2, W4 = W / 4;   // W % 4 == 0: a thread reads 2 x 4 inputs, writes 2 outputs
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Ho * W4) return;
  const int k = j % W4, yo = j / W4;
  for (int bc = blockIdx.y; bc < BC; bc += gridDim.y) {
    const int c = bc % C;
    const float* src = x + (size_t)bc * H * W + (size_t)(2 * yo) * W + 4 * k;
    const float4 r0 = *reinterpret_cast<const float4*>(src), r1 = *reinterpret_cast<const float4*>(src + W);
    float2 o;
    o.x = fmaxf(fmaxf(act_in(r0.x, pro, c, relu), act_in(r0.y, pro, c, relu)), fmaxf(act_in(r1.x, pro, c, relu), act_in(r1.y, pro, c, relu)));
    o.y = fmaxf(fmaxf(act_in(r0.z, pro, c, relu), act_in(r0.w, pro, c, relu)), fmaxf(act_in(r1.z, pro, c, relu), act_in(r1.w, pro, c, relu)));
    *reinterpret_cast<float2*>(out + (size_t)bc * Ho * Wo + (size_t)yo * Wo + 2 * k) = o;
  }
}

__device__ __forceinline__ int pool_argmax(float v0, float v1, float v2, float v3) {   // first maximum, NaN wins (as ATen)
  int am = 0;
  float m = v0;
  if (v1 > m || isnan(v1)) { m = v1; am = 1; }
  if (v2 > m || isnan(v2)) { m = v2; am = 2; }
  if (v3 > m || isnan(v3)) { m = v3; am = 3; }
  return am;
}

// stats (optional, with accumulate & 2 and x = the raw output y of a conv + BatchNorm + ReLU layer, pro = its scale/shift): the
// result is the gradient wrt that layer's activated output, masked with its ReLU — so the two reductions of its BatchNorm
// backward (sum g, sum g (y - mean)) are formed here, per workgroup and plane: stats[(b * gridDim.x + blockIdx.x)][C][2]
// (the layout wtpse_bn_bwd_from_stats folds).  The skip connections' gradients of the encoder arrive through this kernel:
// 20 stand-alone reduce passes over full-resolution maps per step otherwise.
__global__ __launch_bounds__(256) void maxpool2_bwd_v_k(const float* __restrict__ x, const float* __restrict__ pro, int relu,
                                                        const float* __restrict__ dout, float* __restrict__ dx, int accumulate,
                                                        int BC, int C, int H, int W, const float* __restrict__ mean,
                                                        float* __restrict__ stats) {
  __shared__ float sh4[4];
  const int Ho = H / 2, Wo = W / 2, W4 = W / 4;   // H even, W % 4 == 0: a thread owns two windows (2 rows x 4 columns)
  const int jj = blockIdx.x * 256 + threadIdx.x;
  const bool valid = jj < Ho * W4;
  if (!valid && !stats) return;
  const int j = valid ? jj : 0;
  const int k = j % W4, yo = j / W4;
  for (int bc = blockIdx.y; bc < BC; bc += gridDim.y) {
    const int c = bc % C;
    const size_t off = (size_t)bc * H * W + (size_t)(2 * yo) * W + 4 * k;
    const float4 r0 = *reinterpret_cast<const float4*>(x + off), r1 = *reinterpret_cast<const float4*>(x + off + W);
    const float2 g = *reinterpret_cast<const float2*>(dout + (size_t)bc * Ho * Wo + (size_t)yo * Wo + 2 * k);
    const int a0 = pool_argmax(act_in(r0.x, pro, c, relu), act_in(r0.y, pro, c, relu), act_in(r1.x, pro, c, relu), act_in(r1.y, pro, c, relu));
    const int a1 = pool_argmax(act_in(r0.z, pro, c, relu), act_in(r0.w, pro, c, relu), act_in(r1.z, pro, c, relu), act_in(r1.w, pro, c, relu));
    float4 d0 = make_float4(a0 == 0 ? g.x : 0.f, a0 == 1 ? g.x : 0.f, a1 == 0 ? g.y : 0.f, a1 == 1 ? g.y : 0.f);
    float4 d1 = make_float4(a0 == 2 ? g.x : 0.f, a0 == 3 ? g.x : 0.f, a1 == 2 ? g.y : 0.f, a1 == 3 ? g.y : 0.f);
    if (accumulate & 1) {
      const float4 p0 = *reinterpret_cast<const float4*>(dx + off), p1 = *reinterpret_cast<const float4*>(dx + off + W);
      d0 = make_float4(p0.x + d0.x, p0.y + d0.y, p0.z + d0.z, p0.w + d0.w);
      d1 = make_float4(p1.x + d1.x, p1.y + d1.y, p1.z + d1.z, p1.w + d1.w);
    }
    if (accumulate & 2) {      // ... * [act(x) > 0]: the ReLU that produced x, fused (x is read here anyway)
      d0.x = act_in(r0.x, pro, c, relu) > 0.f ? d0.x : 0.f; d0.y = act_in(r0.y, pro, c, relu) > 0.f ? d0.y : 0.f;
      d0.z = act_in(r0.z, pro, c, relu) > 0.f ? d0.z : 0.f; d0.w = act_in(r0.w, pro, c, relu) > 0.f ? d0.w : 0.f;
      d1.x = act_in(r1.x, pro, c, relu) > 0.f ? d1.x : 0.f; d1.y = act_in(r1.y, pro, c, relu) > 0.f ? d1.y : 0.f;
      d1.z = act_in(r1.z, pro, c, relu) > 0.f ? d1.z : 0.f; d1.w = act_in(r1.w, pro, c, relu) > 0.f ? d1.w : 0.f;
    }
    if (valid) {
      *reinterpret_cast<float4*>(dx + off) = d0;
      *reinterpret_cast<float4*>(dx + off + W) = d1;
    }
    if (stats) {
      const float mu = mean[c];
      float s1 = 0.f, s2 = 0.f;
      if (valid) {
#pragma clang fp contract(off)
        s1 = ((d0.x + d0.y) + (d0.z + d0.w)) + ((d1.x + d1.y) + (d1.z + d1.w));
        s2 = ((d0.x * (r0.x - mu) + d0.y * (r0.y - mu)) + (d0.z * (r0.z - mu) + d0.w * (r0.w - mu))) +
             ((d1.x * (r1.x - mu) + d1.y * (r1.y - mu)) + (d1.z * (r1.z - mu) + d1.w * (r1.w - mu)));
      }
      s1 = block_sum(s1, sh4);
      s2 = block_sum(s2, sh4);
      if (threadIdx.x == 0) {
        float* dst = stats + (((size_t)(bc / C) * gridDim.x + blockIdx.x) * C + c) * 2;
        dst[0] = s1;
        dst[1] = s2;
      }
    }
  }
}

// dx[b,c,y,x] (+)= dout[b,c,y/2,x/2] if (y,x) is the first maximum of its window (row-major scan, as ATen), else 0
__global__ __launch_bounds__(256) void maxpool2_bwd_k(const float* __restrict__ x, const float* __restrict__ pro, int relu,
                                                      const float* __restrict__ dout, float* __restrict__ dx,
                                                      int accumulate, int C, int H, int W, long long total) {
  const int Ho = H / 2, Wo = W / 2;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int xx = (int)(i % W);
  long long r = i / W;
  int yy = (int)(r % H);
  long long bc = r / H;
  int c = (int)(bc % C);
  float g = 0.f;
  int yo = yy >> 1, xo = xx >> 1;
  if (yo < Ho && xo < Wo) {
    const float* src = x + (size_t)bc * H * W + (size_t)(2 * yo) * W + 2 * xo;
    float v[4] = {act_in(src[0], pro, c, relu), act_in(src[1], pro, c, relu), act_in(src[W], pro, c, relu),
                  act_in(src[W + 1], pro, c, relu)};
    int am = 0;
    float m = v[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (v[k] > m || isnan(v[k])) { m = v[k]; am = k; }
    if (am == (yy & 1) * 2 + (xx & 1)) g = dout[(size_t)bc * Ho * Wo + (size_t)yo * Wo + xo];
  }
  g = (accumulate & 1) ? dx[i] + g : g;
  if ((accumulate & 2) && !(act_in(x[i], pro, c, relu) > 0.f)) g = 0.f;
  dx[i] = g;
}

// ------------------------------------------------------------------------------------------------ bilinear x2 (align_corners=False)
__device__ __forceinline__ void up_src(int d, int n, int& i0, int& i1, float& l1) {
  float s = 0.5f * (d + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void upsample2x_fwd_k(const float* __restrict__ x, const float* __restrict__ pro, int relu,
                                                        float* __restrict__ out, int C, int H, int W, long long total) {
  const int Ho = 2 * H, Wo = 2 * W;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int xo = (int)(i % Wo);
  long long r = i / Wo;
  int yo = (int)(r % Ho);
  long long bc = r / Ho;
  int c = (int)(bc % C);
  int y0, y1, x0, x1;
  float ly, lx;
  up_src(yo, H, y0, y1, ly);
  up_src(xo, W, x0, x1, lx);
  const float* src = x + (size_t)bc * H * W;
  float v00 = act_in(src[y0 * W + x0], pro, c, relu), v01 = act_in(src[y0 * W + x1], pro, c, relu);
  float v10 = act_in(src[y1 * W + x0], pro, c, relu), v11 = act_in(src[y1 * W + x1], pro, c, relu);
  // same association as ATen's upsample_bilinear2d: h0*(w0*a + w1*b) + h1*(w0*c + w1*d)
  out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

__device__ __forceinline__ float up_w(int d, int n, int k) {  // weight of source k in destination d (1-D)
  int i0, i1;
  float l1;
  up_src(d, n, i0, i1, l1);
  return (i0 == k ? 1.f - l1 : 0.f) + (i1 == k ? l1 : 0.f);
}

// 4 consecutive outputs of TWO rows per thread (W even): the output rows 2p + 1 and 2p + 2 interpolate between the same input rows p and
// p + 1 (weights 0.25 / 0.75), so one thread takes the pair — 2 x 4 input loads and the index arithmetic once for two 16-byte stores
// (one row per thread: 3.0 TB/s; p runs over -1 .. H - 1: the first pair holds output row 0 only, the last one row 2H - 1 only).
// Per output bitwise the expression tree of the scalar kernel / ATen (weights from up_src; (1, 0) at the clamped first column and row);
// one plane per blockIdx.y, 32-bit index arithmetic.
// stats (optional): [B * gridDim.x][C][2] per-workgroup (sum, sum of squares) of the OUTPUT, the train-mode BatchNorm
// statistics of a 1x1 conv that was moved in front of the upsampling (see convu_fwd in nn.py)
__global__ __launch_bounds__(256) void upsample2x_fwd4_v_k(const float* __restrict__ x, const float* __restrict__ pro, int relu,
                                                           float* __restrict__ out, float* __restrict__ stats, int BC, int C,
                                                           int H, int W) {
  __shared__ float sh4[4];
  const int Ho = 2 * H, Wo = 2 * W, W2 = W / 2;
  const int jj = blockIdx.x * 256 + threadIdx.x;   // over (H + 1) * (Wo / 4)
  const bool valid = jj < (H + 1) * W2;
  if (!valid && !stats) return;
  const int j = valid ? jj : 0;
  const int k = j % W2, p = j / W2 - 1;
  const int ya = 2 * p + 1, yb = 2 * p + 2;        // this thread's output rows
  const bool va = valid && ya >= 0, vb = valid && yb < Ho;
  int t0, t1;
  float la = 0.f, lb = 0.f;
  if (ya >= 0) up_src(ya, H, t0, t1, la);          // rows (p, min(p + 1, H - 1))
  if (yb < Ho) up_src(yb, H, t0, t1, lb);          // rows (p, p + 1); p = -1 (output row 0): rows (0, min(1, H - 1)) with weight 0 on the second
  const int y0 = max(p, 0), y1 = p < 0 ? min(1, H - 1) : min(p + 1, H - 1);
  const int m = 2 * k;
  const int xm1 = max(m - 1, 0), x1 = min(m + 1, W - 1), x2 = min(m + 2, W - 1);
  for (int bc = blockIdx.y; bc < BC; bc += gridDim.y) {
    const int c = bc % C;
    const float* r0 = x + (size_t)bc * H * W + (size_t)y0 * W;
    const float* r1 = x + (size_t)bc * H * W + (size_t)y1 * W;
    float a[4] = {act_in(r0[xm1], pro, c, relu), act_in(r0[m], pro, c, relu), act_in(r0[x1], pro, c, relu), act_in(r0[x2], pro, c, relu)};
    float b[4] = {act_in(r1[xm1], pro, c, relu), act_in(r1[m], pro, c, relu), act_in(r1[x1], pro, c, relu), act_in(r1[x2], pro, c, relu)};
    float ha[4], hb[4];
    if (m == 0) { ha[0] = 1.f * a[1] + 0.f * a[2]; hb[0] = 1.f * b[1] + 0.f * b[2]; }
    else        { ha[0] = 0.25f * a[0] + 0.75f * a[1]; hb[0] = 0.25f * b[0] + 0.75f * b[1]; }
    ha[1] = 0.75f * a[1] + 0.25f * a[2]; hb[1] = 0.75f * b[1] + 0.25f * b[2];
    ha[2] = 0.25f * a[1] + 0.75f * a[2]; hb[2] = 0.25f * b[1] + 0.75f * b[2];
    ha[3] = 0.75f * a[2] + 0.25f * a[3]; hb[3] = 0.75f * b[2] + 0.25f * b[3];
    float4 oa, ob;
    oa.x = (1.f - la) * ha[0] + la * hb[0];
    oa.y = (1.f - la) * ha[1] + la * hb[1];
    oa.z = (1.f - la) * ha[2] + la * hb[2];
    oa.w = (1.f - la) * ha[3] + la * hb[3];
    ob.x = (1.f - lb) * ha[0] + lb * hb[0];
    ob.y = (1.f - lb) * ha[1] + lb * hb[1];
    ob.z = (1.f - lb) * ha[2] + lb * hb[2];
    ob.w = (1.f - lb) * ha[3] + lb * hb[3];
    float* ob0 = out + (size_t)bc * Ho * Wo + 4 * k;
    if (va) *reinterpret_cast<float4*>(ob0 + (size_t)ya * Wo) = oa;
    if (vb) *reinterpret_cast<float4*>(ob0 + (size_t)yb * Wo) = ob;
    if (stats) {
      float s1 = (va ? (oa.x + oa.y) + (oa.z + oa.w) : 0.f) + (vb ? (ob.x + ob.y) + (ob.z + ob.w) : 0.f);
      float s2 = (va ? (oa.x * oa.x + oa.y * oa.y) + (oa.z * oa.z + oa.w * oa.w) : 0.f) +
                 (vb ? (ob.x * ob.x + ob.y * ob.y) + (ob.z * ob.z + ob.w * ob.w) : 0.f);
      s1 = block_sum(s1, sh4);
      s2 = block_sum(s2, sh4);
      if (threadIdx.x == 0) {
        float* dst = stats + (((size_t)(bc / C) * gridDim.x + blockIdx.x) * C + c) * 2;
        dst[0] = s1;
        dst[1] = s2;
      }
    }
  }
}

// adjoint of the above: gather over the <=4x4 destination pixels that read source (y,x)
__global__ __launch_bounds__(256) void upsample2x_bwd_k(const float* __restrict__ dout, float* __restrict__ dx,
                                                        int accumulate, int H, int W, long long total) {
  const int Ho = 2 * H, Wo = 2 * W;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int xx = (int)(i % W);
  long long r = i / W;
  int yy = (int)(r % H);
  long long bc = r / H;
  const float* src = dout + (size_t)bc * Ho * Wo;
  float g = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 2; ++dy) {
    int yo = 2 * yy + dy;
    if (yo < 0 || yo >= Ho) continue;
    float wy = up_w(yo, H, yy);
    if (wy == 0.f) continue;
    float rowsum = 0.f;
#pragma unroll
    for (int dxo = -1; dxo <= 2; ++dxo) {
      int xo = 2 * xx + dxo;
      if (xo < 0 || xo >= Wo) continue;
      float wx = up_w(xo, W, xx);
      rowsum += wx * src[(size_t)yo * Wo + xo];
    }
    g += wy * rowsum;
  }
  dx[i] = accumulate ? dx[i] + g : g;
}

// plane-per-blockIdx.y form, 4 consecutive dx of one row per thread (W % 4 == 0): the 4 x 10 window of dout comes in
// as 2 x 16-byte + 2 scalar loads per row; per output the same loops and summation order as the scalar kernel
// BN: dout is not materialised — it is the second half of a BatchNorm backward, dout = fmaf(k1, g, fmaf(k2, y, k3)) per channel
// (bn_bwd_apply_k's expression, bn.hip: the same bits) with g = `dout`, y = `bn_y`, (k1, k2, k3) = `bn_coef`[C][3]
template <bool BN>
__global__ __launch_bounds__(256) void upsample2x_bwd_v_k(const float* __restrict__ dout, float* __restrict__ dx, int accumulate,
                                                          int BC, int H, int W, const float* __restrict__ bn_y,
                                                          const float* __restrict__ bn_coef, int C, unsigned* __restrict__ amax) {
  const int Ho = 2 * H, Wo = 2 * W, W4 = W / 4;
  const int j = blockIdx.x * 256 + threadIdx.x;   // over H*(W/4)
  const bool live = j < H * W4;
  const int k = j % W4, yy = j / W4;
  const int xb = 4 * k;            // first dx column; dout columns 2*xb-1 .. 2*xb+8
  unsigned am = 0u;                // largest |dx| this thread wrote (amax table of dx, common.h)
  for (int bc = blockIdx.y; live && bc < BC; bc += gridDim.y) {
    const float* src = dout + (size_t)bc * Ho * Wo;
    float k1 = 1.f, k2 = 0.f, k3 = 0.f;
    if (BN) {
      const int c = bc % C;
      k1 = bn_coef[3 * c]; k2 = bn_coef[3 * c + 1]; k3 = bn_coef[3 * c + 2];
    }
    float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 2; ++dy) {
      const int yo = 2 * yy + dy;
      if (yo < 0 || yo >= Ho) continue;
      const float wy = up_w(yo, H, yy);
      if (wy == 0.f) continue;
      const float* row = src + (size_t)yo * Wo + 2 * xb;
      const float4 q0 = *reinterpret_cast<const float4*>(row), q1 = *reinterpret_cast<const float4*>(row + 4);
      float w[10];                 // w[t] = dout[yo][2*xb - 1 + t]
      w[0] = xb > 0 ? row[-1] : 0.f;
      w[1] = q0.x; w[2] = q0.y; w[3] = q0.z; w[4] = q0.w; w[5] = q1.x; w[6] = q1.y; w[7] = q1.z; w[8] = q1.w;
      w[9] = 2 * xb + 8 < Wo ? row[8] : 0.f;
      if (BN) {
        const float* yrow = bn_y + (size_t)bc * Ho * Wo + (size_t)yo * Wo + 2 * xb;
        const float4 y0 = *reinterpret_cast<const float4*>(yrow), y1 = *reinterpret_cast<const float4*>(yrow + 4);
        const float v[10] = {xb > 0 ? yrow[-1] : 0.f, y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w, 2 * xb + 8 < Wo ? yrow[8] : 0.f};
#pragma unroll
        for (int t = 0; t < 10; ++t) w[t] = fmaf(k1, w[t], fmaf(k2, v[t], k3));
        if (xb == 0) w[0] = 0.f;                    // outside the map: no gradient (not k3)
        if (2 * xb + 8 >= Wo) w[9] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int xx = xb + e;
        float rowsum = 0.f;
#pragma unroll
        for (int dxo = -1; dxo <= 2; ++dxo) {
          const int xo = 2 * xx + dxo;
          if (xo < 0 || xo >= Wo) continue;
          rowsum += up_w(xo, W, xx) * w[2 * e + dxo + 1];
        }
        g[e] += wy * rowsum;
      }
    }
    float* dst = dx + (size_t)bc * H * W + (size_t)yy * W + xb;
    float4 o = make_float4(g[0], g[1], g[2], g[3]);
    if (accumulate) {
      const float4 p = *reinterpret_cast<const float4*>(dst);
      o = make_float4(p.x + o.x, p.y + o.y, p.z + o.z, p.w + o.w);
    }
    *reinterpret_cast<float4*>(dst) = o;
    am = max(am, max(max(amax_bits(o.x), amax_bits(o.y)), max(amax_bits(o.z), amax_bits(o.w))));
  }
  if (amax) amax_publish_wave(amax, am, (blockIdx.x + blockIdx.y * gridDim.x) * 4u + (threadIdx.x >> 6));
}

// F.interpolate(x, size=(Ho,Wo), mode="bilinear") with align_corners=False (Trainer.py:206-209): validation resizes the
// logits to the label size.  src = (dst + 0.5) * in/out - 0.5, clamped at 0; same expression tree as ATen.
__global__ __launch_bounds__(256) void resize_bilinear_k(const float* __restrict__ x, float* __restrict__ out, int H, int W,
                                                         int Ho, int Wo, float sy, float sx, long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int xo = (int)(i % Wo);
  long long r = i / Wo;
  int yo = (int)(r % Ho);
  long long bc = r / Ho;
  float fy = sy * (yo + 0.5f) - 0.5f, fx = sx * (xo + 0.5f) - 0.5f;
  if (fy < 0.f) fy = 0.f;
  if (fx < 0.f) fx = 0.f;
  int y0 = (int)fy, x0 = (int)fx;
  int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  float ly = fy - (float)y0, lx = fx - (float)x0;
  const float* src = x + (size_t)bc * H * W;
  float v00 = src[y0 * W + x0], v01 = src[y0 * W + x1], v10 = src[y1 * W + x0], v11 = src[y1 * W + x1];
  out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

// ------------------------------------------------------------------------------------------------ small elementwise
__global__ __launch_bounds__(256) void relu_mask_k(const float* __restrict__ dz, const float* __restrict__ ref,
                                                   float* __restrict__ dy, int accumulate, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float g = ref[i] > 0.f ? dz[i] : 0.f;
  dy[i] = accumulate ? dy[i] + g : g;
}

__global__ __launch_bounds__(256) void axpy_k(float* __restrict__ dst, const float* __restrict__ src, float alpha, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = fmaf(alpha, src[i], dst[i]);
}

// 16 bytes per lane (n % 4 == 0, 16-byte aligned): the scalar forms above streamed at 4.1 TB/s
__global__ __launch_bounds__(256) void relu_mask_v_k(const float4* __restrict__ dz, const float4* __restrict__ ref,
                                                     float4* __restrict__ dy, int accumulate, long long n4) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 d = dz[i], r = ref[i];
  float4 g = make_float4(r.x > 0.f ? d.x : 0.f, r.y > 0.f ? d.y : 0.f, r.z > 0.f ? d.z : 0.f, r.w > 0.f ? d.w : 0.f);
  if (accumulate) {
    const float4 o = dy[i];
    g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
  }
  dy[i] = g;
}

__global__ __launch_bounds__(256) void axpy_v_k(float4* __restrict__ dst, const float4* __restrict__ src, float alpha, long long n4) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 s4 = src[i];
  float4 d = dst[i];
  d.x = fmaf(alpha, s4.x, d.x); d.y = fmaf(alpha, s4.y, d.y); d.z = fmaf(alpha, s4.z, d.z); d.w = fmaf(alpha, s4.w, d.w);
  dst[i] = d;
}

// out[j] (+)= sum_r partial[r][j]
__global__ __launch_bounds__(256) void reduce_rows_k(const float* __restrict__ partial, int rows, int cols,
                                                     float* __restrict__ out, int accumulate, float scale) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  double s = 0.0;
  for (int r = 0; r < rows; ++r) s += partial[(size_t)r * cols + j];
  float v = (float)s * scale;
  out[j] = accumulate ? out[j] + v : v;
}

// same for a few columns and many rows (the (dw, db) partials of the attention layer: 8192 x 2): one workgroup per
// column, 256 fp64 partial sums folded in a fixed order
__global__ __launch_bounds__(256) void reduce_rows_tall_k(const float* __restrict__ partial, int rows, int cols,
                                                          float* __restrict__ out, int accumulate, float scale) {
  __shared__ double sh[256];
  const int j = blockIdx.x, t = threadIdx.x;
  double s = 0.0;
  for (int r = t; r < rows; r += 256) s += partial[(size_t)r * cols + j];
  sh[t] = s;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if (t < m) sh[t] += sh[t + m];
    __syncthreads();
  }
  if (t == 0) {
    float v = (float)sh[0] * scale;
    out[j] = accumulate ? out[j] + v : v;
  }
}

// ------------------------------------------------------------------------------------------------ shape attention + fusion
// a_pre = w*z + b ; att = sigmoid(a_pre) ; fuse[c] = coef*emb[c] + att*emb[c] ; mask = att > 0.75
__global__ __launch_bounds__(256) void attn_fuse_fwd_k(const float* __restrict__ z, const float* __restrict__ wb,
                                                       const float* __restrict__ emb, float coef, float* __restrict__ att,
                                                       float* __restrict__ att_pre, float* __restrict__ mask,
                                                       float* __restrict__ fuse, int CE, int HW, long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over B*HW
  if (i >= total) return;
  long long b = i / HW;
  int p = (int)(i - b * HW);
  float pre = fmaf(wb[0], z[i], wb[1]);
  float a = sigmoidf_(pre);
  if (att) att[i] = a;
  if (att_pre) att_pre[i] = pre;
  if (mask) mask[i] = a > 0.75f ? 1.f : 0.f;
  for (int c = 0; c < CE; ++c) {
    size_t idx = ((size_t)b * CE + c) * HW + p;
    float e = emb[idx];
    fuse[idx] = coef * e + a * e;
  }
}

// demb = dfuse*(coef+att); datt = sum_c dfuse*emb; dpre = datt*att*(1-att); dz = dpre*w; partial[blk] = (sum dpre*z, sum dpre)
__global__ __launch_bounds__(256) void attn_fuse_bwd_k(const float* __restrict__ dfuse, const float* __restrict__ z,
                                                       const float* __restrict__ emb, const float* __restrict__ att,
                                                       const float* __restrict__ wb, float coef, float* __restrict__ demb,
                                                       float* __restrict__ dz, float* __restrict__ partial, int CE, int HW,
                                                       long long total) {
  __shared__ float sh[4];
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  float dw = 0.f, db = 0.f;
  if (i < total) {
    long long b = i / HW;
    int p = (int)(i - b * HW);
    float a = att[i];
    float datt = 0.f;
    for (int c = 0; c < CE; ++c) {
      size_t idx = ((size_t)b * CE + c) * HW + p;
      float g = dfuse[idx];
      datt = fmaf(g, emb[idx], datt);
      demb[idx] = g * (coef + a);
    }
    float dpre = datt * a * (1.f - a);
    if (dz) dz[i] = dpre * wb[0];
    dw = dpre * z[i];
    db = dpre;
  }
  dw = block_sum(dw, sh);
  db = block_sum(db, sh);
  if (threadIdx.x == 0) {
    partial[(size_t)blockIdx.x * 2] = dw;
    partial[(size_t)blockIdx.x * 2 + 1] = db;
  }
}

// ------------------------------------------------------------------------------------------------ reparameterisation
// teacher (algorithms.py:1068-1075): z = mu + exp(logvar/2)*eps
__global__ __launch_bounds__(256) void reparam_fwd_k(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                     const float* __restrict__ eps, float* __restrict__ z, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) z[i] = fmaf(expf(logvar[i] * 0.5f), eps[i], mu[i]);
}
__global__ __launch_bounds__(256) void reparam_bwd_k(const float* __restrict__ dz, const float* __restrict__ logvar,
                                                     const float* __restrict__ eps, float* __restrict__ dlogvar, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dlogvar[i] = dz[i] * eps[i] * 0.5f * expf(logvar[i] * 0.5f);
}
// student (shape_networks.py:502-510): s = normal(mu, std) ; z = s*std + mu
__global__ __launch_bounds__(256) void reparam_student_k(const float* __restrict__ mu, const float* __restrict__ std_,
                                                         const float* __restrict__ eps, float* __restrict__ z, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    float s = fmaf(std_[i], eps[i], mu[i]);
    z[i] = fmaf(s, std_[i], mu[i]);
  }
}
__global__ __launch_bounds__(256) void exp_half_k(const float* __restrict__ logvar, float* __restrict__ std_, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) std_[i] = expf(logvar[i] * 0.5f);
}

// NaN scrub (shape_networks.py:490-492): if ANY element is NaN, nan_to_num the whole tensor — no host sync:
// pass 1 raises a device flag, pass 2 is a no-op unless the flag is up.
__global__ __launch_bounds__(256) void nan_flag_k(const float* __restrict__ x, long long n, int* __restrict__ flag) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  bool bad = i < n && isnan(x[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ __launch_bounds__(256) void nan_scrub_k(float* __restrict__ x, long long n, const int* __restrict__ flag) {
  if (*flag == 0) return;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  if (isnan(v)) v = 0.f;
  else if (isinf(v)) v = v > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  x[i] = v;
}

// ------------------------------------------------------------------------------------------------ losses
// BCELoss(sigmoid(x), t), mean (Trainer.py:19,787) — per-block partial sums
__global__ __launch_bounds__(256) void bce_sigmoid_fwd_k(const float* __restrict__ x, const float* __restrict__ t, long long n,
                                                         float* __restrict__ partial) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float s = sigmoidf_(x[i]);
    float l1 = fmaxf(logf(s), -100.f), l0 = fmaxf(logf(1.f - s), -100.f);
    acc -= t[i] * l1 + (1.f - t[i]) * l0;
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void bce_sigmoid_bwd_k(const float* __restrict__ x, const float* __restrict__ t,
                                                         const float* g, float w, long long n, float* __restrict__ dx) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = sigmoidf_(x[i]);
  float q = s * (1.f - s);
  float up = (g ? *g : 1.f) * w / (float)n;
  dx[i] = up * (s - t[i]) / fmaxf(q, 1e-12f) * q;
}
// binary_cross_entropy_with_logits(x*m, t, pos_weight) (Trainer.py:868-871)
__global__ __launch_bounds__(256) void bce_logits_pw_fwd_k(const float* __restrict__ x, const float* __restrict__ m,
                                                           const float* __restrict__ t, const float* __restrict__ pw,
                                                           long long n, float* __restrict__ partial) {
  __shared__ float sh[4];
  const float pos = *pw;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float xi = x[i] * m[i], ti = t[i];
    float lw = 1.f + (pos - 1.f) * ti;
    acc += (1.f - ti) * xi + lw * (log1pf(expf(-fabsf(xi))) + fmaxf(-xi, 0.f));
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void bce_logits_pw_bwd_k(const float* __restrict__ x, const float* __restrict__ m,
                                                           const float* __restrict__ t, const float* __restrict__ pw,
                                                           const float* g, float w, long long n, float* __restrict__ dx) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float xi = x[i] * m[i], ti = t[i];
  float lw = 1.f + (*pw - 1.f) * ti;
  float up = (g ? *g : 1.f) * w / (float)n;
  dx[i] = up * ((1.f - ti) + lw * (sigmoidf_(xi) - 1.f)) * m[i];
}
// sums[0] = sum a ; sums[1] = sum a*b  (pos_weight = sums[0]/sums[1], Trainer.py:865)
__global__ __launch_bounds__(256) void sum2_k(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                              float* __restrict__ partial) {
  __shared__ float sh[4];
  float s0 = 0.f, s1 = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    s0 += a[i];
    s1 += a[i] * b[i];
  }
  s0 = block_sum(s0, sh);
  s1 = block_sum(s1, sh);
  if (threadIdx.x == 0) {
    partial[(size_t)blockIdx.x * 2] = s0;
    partial[(size_t)blockIdx.x * 2 + 1] = s1;
  }
}
__global__ void pos_weight_k(const float* __restrict__ sums, float* __restrict__ pw) {
  float v = sums[0] / sums[1];
  *pw = (isinf(v) || isnan(v)) ? 1.f : v;
}
// MSE mean (shape_networks.py:596-597)
__global__ __launch_bounds__(256) void mse_fwd_k(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                 float* __restrict__ partial) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float d = a[i] - b[i];
    acc = fmaf(d, d, acc);
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void mse_bwd_k(const float* __restrict__ a, const float* __restrict__ b, const float* g,
                                                 float w, long long n, float* __restrict__ da) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) da[i] = (g ? *g : 1.f) * w * 2.f * (a[i] - b[i]) / (float)n;
}
// od_pred = sigmoid(logit) > 0.75 ; roi = (image + 1) * od_pred - 1  (Trainer.py:842-853)
__global__ __launch_bounds__(256) void roi_k(const float* __restrict__ image, const float* __restrict__ logit,
                                             float* __restrict__ roi, float* __restrict__ od_pred, int C, int HW,
                                             long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over B*HW
  if (i >= total) return;
  long long b = i / HW;
  int p = (int)(i - b * HW);
  float m = sigmoidf_(logit[i]) > 0.75f ? 1.f : 0.f;
  od_pred[i] = m;
  for (int c = 0; c < C; ++c) {
    size_t idx = ((size_t)b * C + c) * HW + p;
    roi[idx] = (image[idx] + 1.f) * m - 1.f;
  }
}

// ------------------------------------------------------------------------------------------------ Adam (torch.optim.Adam semantics, no weight decay / amsgrad)
// step_dev (optional): device int holding the number of COMPLETED optimiser steps; the bias corrections are then formed
// here from t = step + *step_dev (one thread per block, in double as on the host), so that a captured launch stays valid
// when it is replayed (hipGraph): nothing that changes from step to step is passed by value.
__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                              float bc1, float bc2_sqrt, int step, const int* __restrict__ step_dev,
                                              double b1d, double b2d) {
  if (step_dev) {
    __shared__ float bc[2];
    if (threadIdx.x == 0) {
      const double t = (double)(step + *step_dev);
      bc[0] = (float)(1.0 - pow(b1d, t));
      bc[1] = (float)sqrt(1.0 - pow(b2d, t));
    }
    __syncthreads();
    bc1 = bc[0];
    bc2_sqrt = bc[1];
  }
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float gi = g[i];
  float mi = b1 * m[i] + (1.f - b1) * gi;
  float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

// ------------------------------------------------------------------------------------------------ Philox4x32-10 -> N(0,1)
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
  uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
  uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
// element i of the stream (seed, offset) depends only on (seed, offset + i/4): any sharding of the rows
// over ranks reproduces the single-device stream when each rank passes its global element offset
// offset_dev (optional): device counter added to `offset` (the stream position lives in device memory so that a
// captured launch draws fresh numbers on every replay)
__global__ __launch_bounds__(256) void randn_k(float* __restrict__ out, long long n, unsigned long long seed,
                                               unsigned long long offset, const unsigned long long* __restrict__ offset_dev) {
  if (offset_dev) offset += *offset_dev;
  long long q = (long long)blockIdx.x * 256 + threadIdx.x;  // one Philox block = 4 normals
  long long base = q * 4;
  if (base >= n) return;
  unsigned long long ctr = offset / 4 + (unsigned long long)q;
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float S = 2.3283064365386963e-10f;  // 2^-32
  float u0 = ((float)c0 + 0.5f) * S, u1 = ((float)c1 + 0.5f) * S, u2 = ((float)c2 + 0.5f) * S, u3 = ((float)c3 + 0.5f) * S;
  if (u0 >= 1.f) u0 = 0.99999994f;
  if (u2 >= 1.f) u2 = 0.99999994f;
  float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
  float zv[4] = {r0 * cosf(6.283185307179586f * u1), r0 * sinf(6.283185307179586f * u1),
                 r1 * cosf(6.283185307179586f * u3), r1 * sinf(6.283185307179586f * u3)};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < n) out[base + k] = zv[k];
}

// ================================================================================================ C ABI
#define GRID1(n) dim3((unsigned)(((n) + 255) / 256))
#define ST ((hipStream_t)stream)
#define PLANE_GRID(per_plane, planes) dim3((unsigned)(((per_plane) + 255) / 256), (unsigned)((planes) < 32768 ? (planes) : 32768))
static inline unsigned red_blocks(long long n) {
  long long b = (n + 256 * 16 - 1) / (256 * 16);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (unsigned)b;
}
extern "C" int wtpse_reduce_blocks(long long n) { return (int)red_blocks(n); }

extern "C" int wtpse_maxpool2_fwd(const float* x, const float* pro, int relu, float* out, int B, int C, int H, int W, void* stream) {
  WTPSE_REQUIRE(x && out && B > 0 && C > 0 && H >= 2 && W >= 2);
  long long total = (long long)B * C * (H / 2) * (W / 2);
  if (W % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL(maxpool2_fwd_v_k, PLANE_GRID((H / 2) * (W / 4), B * C), dim3(256), 0, ST, x, pro, relu, out, B * C, C, H, W);
  else
    hipLaunchKernelGGL(maxpool2_fwd_k, GRID1(total), dim3(256), 0, ST, x, pro, relu, out, C, H, W, total);
  return wtpse_status();
}
extern "C" int wtpse_maxpool2_bwd(const float* x, const float* pro, int relu, const float* dout, float* dx, int accumulate,
                                  int B, int C, int H, int W, void* stream) {
  WTPSE_REQUIRE(x && dout && dx && B > 0 && C > 0 && H >= 2 && W >= 2);
  long long total = (long long)B * C * H * W;
  if (W % 4 == 0 && H % 2 == 0 && (((uintptr_t)x | (uintptr_t)dx | (uintptr_t)dout) & 15) == 0)
    hipLaunchKernelGGL(maxpool2_bwd_v_k, PLANE_GRID((H / 2) * (W / 4), B * C), dim3(256), 0, ST, x, pro, relu, dout, dx, accumulate,
                       B * C, C, H, W, (const float*)nullptr, (float*)nullptr);
  else
    hipLaunchKernelGGL(maxpool2_bwd_k, GRID1(total), dim3(256), 0, ST, x, pro, relu, dout, dx, accumulate, C, H, W, total);
  return wtpse_status();
}
extern "C" int wtpse_maxpool2_bwd_stats_blocks(int B, int H, int W) { return B * (((H / 2) * (W / 4) + 255) / 256); }
// wtpse_maxpool2_bwd(accumulate | 2) that also forms the BatchNorm-backward reductions of the layer that produced x (see the kernel)
extern "C" int wtpse_maxpool2_bwd_bnb(const float* x, const float* pro, int relu, const float* dout, float* dx, int accumulate,
                                      const float* mean, float* stats, int B, int C, int H, int W, void* stream) {
  WTPSE_REQUIRE(x && pro && dout && dx && mean && stats && B > 0 && C > 0 && H >= 2 && W >= 4);
  WTPSE_REQUIRE(W % 4 == 0 && H % 2 == 0 && (((uintptr_t)x | (uintptr_t)dx | (uintptr_t)dout) & 15) == 0);
  WTPSE_REQUIRE(B * C < 32768);   // one plane per blockIdx.y: the statistics rows are indexed by gridDim.x
  hipLaunchKernelGGL(maxpool2_bwd_v_k, PLANE_GRID((H / 2) * (W / 4), B * C), dim3(256), 0, ST, x, pro, relu, dout, dx,
                     (accumulate & 1) | 2, B * C, C, H, W, mean, stats);
  return wtpse_status();
}
extern "C" int wtpse_upsample2x_fwd(const float* x, const float* pro, int relu, float* out, int B, int C, int H, int W, void* stream) {
  WTPSE_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0);
  long long total = (long long)B * C * H * W * 4;
  if (W % 2 == 0 && W >= 2 && (((uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL(upsample2x_fwd4_v_k, PLANE_GRID((H + 1) * (W / 2), B * C), dim3(256), 0, ST, x, pro, relu, out, nullptr, B * C, C, H, W);
  else
    hipLaunchKernelGGL(upsample2x_fwd_k, GRID1(total), dim3(256), 0, ST, x, pro, relu, out, C, H, W, total);
  return wtpse_status();
}
extern "C" int wtpse_upsample2x_stats_blocks(int B, int H, int W) { return B * (((H + 1) * (W / 2) + 255) / 256); }
extern "C" int wtpse_upsample2x_fwd_stats(const float* x, float* out, float* stats, int B, int C, int H, int W, void* stream) {
  WTPSE_REQUIRE(x && out && stats && B > 0 && C > 0 && H > 0 && W >= 2 && W % 2 == 0 && (((uintptr_t)out) & 15) == 0);
  WTPSE_REQUIRE(B * C < 32768);   // one plane per blockIdx.y: the statistics rows are indexed by gridDim.x
  hipLaunchKernelGGL(upsample2x_fwd4_v_k, PLANE_GRID((H + 1) * (W / 2), B * C), dim3(256), 0, ST, x, nullptr, 0, out, stats, B * C, C, H, W);
  return wtpse_status();
}
// amax (optional): the amax table (common.h) of dx, ZERO on entry — dx is the dY of the 1x1 convolution in front of the upsampling
// (every wave folds its maximum into a shard with one no-return atomic).
static inline dim3 up_bwd_grid(int per_plane, int planes, bool) { return PLANE_GRID(per_plane, planes); }
extern "C" int wtpse_amax(const float* x, long long n, unsigned* amax_table, void* stream);
extern "C" int wtpse_upsample2x_bwd(const float* dout, float* dx, int accumulate, int B, int C, int H, int W, unsigned* amax, void* stream) {
  WTPSE_REQUIRE(dout && dx && B > 0 && C > 0 && H > 0 && W > 0);
  long long total = (long long)B * C * H * W;
  if (W % 4 == 0 && (((uintptr_t)dout | (uintptr_t)dx) & 15) == 0) {
    hipLaunchKernelGGL(upsample2x_bwd_v_k<false>, up_bwd_grid(H * (W / 4), B * C, amax != nullptr), dim3(256), 0, ST, dout, dx, accumulate,
                       B * C, H, W, nullptr, nullptr, C, amax);
    return wtpse_status();
  }
  hipLaunchKernelGGL(upsample2x_bwd_k, GRID1(total), dim3(256), 0, ST, dout, dx, accumulate, H, W, total);
  const int rc = wtpse_status();
  return (rc || !amax) ? rc : wtpse_amax(dx, total, amax, stream);
}
extern "C" int wtpse_upsample2x_bwd_bn(const float* g, const float* bn_y, const float* bn_coef, float* dx, int B, int C, int H, int W,
                                       unsigned* amax, void* stream) {
  WTPSE_REQUIRE(g && bn_y && bn_coef && dx && B > 0 && C > 0 && H > 0 && W > 0 && W % 4 == 0);
  WTPSE_REQUIRE((((uintptr_t)g | (uintptr_t)bn_y | (uintptr_t)dx) & 15) == 0);
  hipLaunchKernelGGL(upsample2x_bwd_v_k<true>, up_bwd_grid(H * (W / 4), B * C, amax != nullptr), dim3(256), 0, ST, g, dx, 0, B * C, H, W,
                     bn_y, bn_coef, C, amax);
  return wtpse_status();
}
extern "C" int wtpse_resize_bilinear(const float* x, float* out, int B, int C, int H, int W, int Ho, int Wo, void* stream) {
  WTPSE_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0);
  long long total = (long long)B * C * Ho * Wo;
  hipLaunchKernelGGL(resize_bilinear_k, GRID1(total), dim3(256), 0, ST, x, out, H, W, Ho, Wo, (float)H / (float)Ho,
                     (float)W / (float)Wo, total);
  return wtpse_status();
}
extern "C" int wtpse_relu_mask(const float* dz, const float* ref, float* dy, int accumulate, long long n, void* stream) {
  WTPSE_REQUIRE(dz && ref && dy && n > 0);
  if (n % 4 == 0 && (((uintptr_t)dz | (uintptr_t)ref | (uintptr_t)dy) & 15) == 0)
    hipLaunchKernelGGL(relu_mask_v_k, GRID1(n / 4), dim3(256), 0, ST, reinterpret_cast<const float4*>(dz),
                       reinterpret_cast<const float4*>(ref), reinterpret_cast<float4*>(dy), accumulate, n / 4);
  else
    hipLaunchKernelGGL(relu_mask_k, GRID1(n), dim3(256), 0, ST, dz, ref, dy, accumulate, n);
  return wtpse_status();
}
// Counter calibration (tools/pmc_traffic.py): plain streaming copies with a KNOWN byte count in the two access widths the
// kernels use — 16 bytes per lane (the WT-loss, weight-gradient and point-wise kernels) and 4 bytes per lane (the convolutions'
// tile loaders and epilogues) — so that the FETCH_SIZE / WRITE_SIZE factors are derived per width instead of chosen per kernel
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern").
__global__ __launch_bounds__(256) void copy_w16_k(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void copy_w4_k(const float* __restrict__ src, float* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int wtpse_copy_probe(const float* src, float* dst, long long n, int bytes_per_lane, void* stream) {
  WTPSE_REQUIRE(src && dst && n > 0 && n % 4 == 0 && (bytes_per_lane == 4 || bytes_per_lane == 16));
  WTPSE_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
  if (bytes_per_lane == 16)
    hipLaunchKernelGGL(copy_w16_k, dim3(8192), dim3(256), 0, ST, reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n / 4);
  else
    hipLaunchKernelGGL(copy_w4_k, dim3(16384), dim3(256), 0, ST, src, dst, n);
  return wtpse_status();
}
// ---- largest magnitude of a tensor into an amax table (common.h; wtpse_x3_terms == 2: the power-of-two scale of a gradient operand,
// wtpse_hip.h).  NaN patterns sort above inf and make the consumer fall back to scale 1, where the NaN propagates as it would anyway.
__global__ __launch_bounds__(256) void amax_k(const float* __restrict__ x, long long n, unsigned* __restrict__ table) {
  unsigned m = 0u;
  const long long n4 = n >> 2;
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      const uint4 v = x4[i];
      m = max(max(m, v.x & 0x7FFFFFFFu), max(v.y & 0x7FFFFFFFu, max(v.z & 0x7FFFFFFFu, v.w & 0x7FFFFFFFu)));
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
      m = max(m, __float_as_uint(x[i]) & 0x7FFFFFFFu);
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
      m = max(m, __float_as_uint(x[i]) & 0x7FFFFFFFu);
  }
  amax_publish_block(table, m, blockIdx.x);
}
extern "C" int wtpse_amax(const float* x, long long n, unsigned* amax_table, void* stream) {
  WTPSE_REQUIRE(amax_table && n >= 0 && (x || n == 0));
  if (hipMemsetAsync(amax_table, 0, AMAX_WORDS * sizeof(unsigned), ST) != hipSuccess) return wtpse_status();
  if (n > 0) {
    const long long blocks = (n / 4 + 255) / 256 + 1;
    hipLaunchKernelGGL(amax_k, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, ST, x, n, amax_table);
  }
  return wtpse_status();
}
extern "C" int wtpse_axpy(float* dst, const float* src, float alpha, long long n, void* stream) {
  WTPSE_REQUIRE(dst && src && n > 0);
  if (n % 4 == 0 && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0)
    hipLaunchKernelGGL(axpy_v_k, GRID1(n / 4), dim3(256), 0, ST, reinterpret_cast<float4*>(dst), reinterpret_cast<const float4*>(src),
                       alpha, n / 4);
  else
    hipLaunchKernelGGL(axpy_k, GRID1(n), dim3(256), 0, ST, dst, src, alpha, n);
  return wtpse_status();
}
extern "C" int wtpse_reduce_rows(const float* partial, int rows, int cols, float* out, int accumulate, float scale, void* stream) {
  WTPSE_REQUIRE(partial && out && rows > 0 && cols > 0);
  if (cols <= 64 && rows >= 1024)
    hipLaunchKernelGGL(reduce_rows_tall_k, dim3(cols), dim3(256), 0, ST, partial, rows, cols, out, accumulate, scale);
  else
    hipLaunchKernelGGL(reduce_rows_k, GRID1(cols), dim3(256), 0, ST, partial, rows, cols, out, accumulate, scale);
  return wtpse_status();
}
extern "C" int wtpse_zero(void* p, long long nbytes, void* stream) {
  WTPSE_REQUIRE(p && nbytes >= 0);
  hipError_t e = hipMemsetAsync(p, 0, (size_t)nbytes, ST);
  return e == hipSuccess ? WTPSE_OK : (int)e;
}
extern "C" int wtpse_attn_fuse_fwd(const float* z, const float* wb, const float* emb, float coef, float* att, float* att_pre,
                                   float* mask, float* fuse, int B, int CE, int HW, void* stream) {
  WTPSE_REQUIRE(z && wb && emb && fuse && B > 0 && CE > 0 && HW > 0);
  long long total = (long long)B * HW;
  hipLaunchKernelGGL(attn_fuse_fwd_k, GRID1(total), dim3(256), 0, ST, z, wb, emb, coef, att, att_pre, mask, fuse, CE, HW, total);
  return wtpse_status();
}
// partial must hold 2*ceil(B*HW/256) floats; d_wb[2] receives (dw, db)
extern "C" int wtpse_attn_fuse_bwd(const float* dfuse, const float* z, const float* emb, const float* att, const float* wb,
                                   float coef, float* demb, float* dz, float* partial, float* d_wb, int accumulate, int B,
                                   int CE, int HW, void* stream) {
  WTPSE_REQUIRE(dfuse && z && emb && att && wb && demb && partial && d_wb && B > 0 && CE > 0 && HW > 0);
  long long total = (long long)B * HW;
  unsigned nb = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(attn_fuse_bwd_k, dim3(nb), dim3(256), 0, ST, dfuse, z, emb, att, wb, coef, demb, dz, partial, CE, HW, total);
  if (nb >= 1024)
    hipLaunchKernelGGL(reduce_rows_tall_k, dim3(2), dim3(256), 0, ST, partial, (int)nb, 2, d_wb, accumulate, 1.f);
  else
    hipLaunchKernelGGL(reduce_rows_k, dim3(1), dim3(256), 0, ST, partial, (int)nb, 2, d_wb, accumulate, 1.f);
  return wtpse_status();
}
extern "C" int wtpse_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long long n, void* stream) {
  WTPSE_REQUIRE(mu && logvar && eps && z && n > 0);
  hipLaunchKernelGGL(reparam_fwd_k, GRID1(n), dim3(256), 0, ST, mu, logvar, eps, z, n);
  return wtpse_status();
}
extern "C" int wtpse_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dlogvar, long long n, void* stream) {
  WTPSE_REQUIRE(dz && logvar && eps && dlogvar && n > 0);
  hipLaunchKernelGGL(reparam_bwd_k, GRID1(n), dim3(256), 0, ST, dz, logvar, eps, dlogvar, n);
  return wtpse_status();
}
extern "C" int wtpse_exp_half(const float* logvar, float* std_, long long n, void* stream) {
  WTPSE_REQUIRE(logvar && std_ && n > 0);
  hipLaunchKernelGGL(exp_half_k, GRID1(n), dim3(256), 0, ST, logvar, std_, n);
  return wtpse_status();
}
extern "C" int wtpse_reparam_student(const float* mu, const float* std_, const float* eps, float* z, long long n, void* stream) {
  WTPSE_REQUIRE(mu && std_ && eps && z && n > 0);
  hipLaunchKernelGGL(reparam_student_k, GRID1(n), dim3(256), 0, ST, mu, std_, eps, z, n);
  return wtpse_status();
}
// flag: one device int, zeroed here; the tensor is scrubbed only if a NaN was seen
extern "C" int wtpse_nan_scrub(float* x, long long n, int* flag, void* stream) {
  WTPSE_REQUIRE(x && flag && n > 0);
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), ST);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nan_flag_k, GRID1(n), dim3(256), 0, ST, x, n, flag);
  hipLaunchKernelGGL(nan_scrub_k, GRID1(n), dim3(256), 0, ST, x, n, flag);
  return wtpse_status();
}
extern "C" int wtpse_bce_sigmoid_fwd(const float* x, const float* t, long long n, float* partial, float* loss, void* stream) {
  WTPSE_REQUIRE(x && t && partial && loss && n > 0);
  unsigned nb = red_blocks(n);
  hipLaunchKernelGGL(bce_sigmoid_fwd_k, dim3(nb), dim3(256), 0, ST, x, t, n, partial);
  hipLaunchKernelGGL(reduce_rows_k, dim3(1), dim3(256), 0, ST, partial, (int)nb, 1, loss, 0, 1.f / (float)n);
  return wtpse_status();
}
extern "C" int wtpse_bce_sigmoid_bwd(const float* x, const float* t, const float* g, float w, long long n, float* dx, void* stream) {
  WTPSE_REQUIRE(x && t && dx && n > 0);
  hipLaunchKernelGGL(bce_sigmoid_bwd_k, GRID1(n), dim3(256), 0, ST, x, t, g, w, n, dx);
  return wtpse_status();
}
// pos_weight from the global batch: pw = sum(mask)/sum(mask*t), 1 if inf/nan. sums[2], pw[1] device scalars
extern "C" int wtpse_pos_weight(const float* mask, const float* t, long long n, float* partial, float* sums, float* pw, void* stream) {
  WTPSE_REQUIRE(mask && t && partial && sums && pw && n > 0);
  unsigned nb = red_blocks(n);
  hipLaunchKernelGGL(sum2_k, dim3(nb), dim3(256), 0, ST, mask, t, n, partial);
  hipLaunchKernelGGL(reduce_rows_k, dim3(1), dim3(256), 0, ST, partial, (int)nb, 2, sums, 0, 1.f);
  hipLaunchKernelGGL(pos_weight_k, dim3(1), dim3(1), 0, ST, sums, pw);
  return wtpse_status();
}
extern "C" int wtpse_pos_weight_from_sums(const float* sums, float* pw, void* stream) {
  WTPSE_REQUIRE(sums && pw);
  hipLaunchKernelGGL(pos_weight_k, dim3(1), dim3(1), 0, ST, sums, pw);
  return wtpse_status();
}
extern "C" int wtpse_bce_logits_pw_fwd(const float* x, const float* mask, const float* t, const float* pw, long long n,
                                       float* partial, float* loss, void* stream) {
  WTPSE_REQUIRE(x && mask && t && pw && partial && loss && n > 0);
  unsigned nb = red_blocks(n);
  hipLaunchKernelGGL(bce_logits_pw_fwd_k, dim3(nb), dim3(256), 0, ST, x, mask, t, pw, n, partial);
  hipLaunchKernelGGL(reduce_rows_k, dim3(1), dim3(256), 0, ST, partial, (int)nb, 1, loss, 0, 1.f / (float)n);
  return wtpse_status();
}
extern "C" int wtpse_bce_logits_pw_bwd(const float* x, const float* mask, const float* t, const float* pw, const float* g,
                                       float w, long long n, float* dx, void* stream) {
  WTPSE_REQUIRE(x && mask && t && pw && dx && n > 0);
  hipLaunchKernelGGL(bce_logits_pw_bwd_k, GRID1(n), dim3(256), 0, ST, x, mask, t, pw, g, w, n, dx);
  return wtpse_status();
}
extern "C" int wtpse_mse_fwd(const float* a, const float* b, long long n, float* partial, float* loss, void* stream) {
  WTPSE_REQUIRE(a && b && partial && loss && n > 0);
  unsigned nb = red_blocks(n);
  hipLaunchKernelGGL(mse_fwd_k, dim3(nb), dim3(256), 0, ST, a, b, n, partial);
  hipLaunchKernelGGL(reduce_rows_k, dim3(1), dim3(256), 0, ST, partial, (int)nb, 1, loss, 0, 1.f / (float)n);
  return wtpse_status();
}
extern "C" int wtpse_mse_bwd(const float* a, const float* b, const float* g, float w, long long n, float* da, void* stream) {
  WTPSE_REQUIRE(a && b && da && n > 0);
  hipLaunchKernelGGL(mse_bwd_k, GRID1(n), dim3(256), 0, ST, a, b, g, w, n, da);
  return wtpse_status();
}
extern "C" int wtpse_roi(const float* image, const float* logit, float* roi, float* od_pred, int B, int C, int HW, void* stream) {
  WTPSE_REQUIRE(image && logit && roi && od_pred && B > 0 && C > 0 && HW > 0);
  long long total = (long long)B * HW;
  hipLaunchKernelGGL(roi_k, GRID1(total), dim3(256), 0, ST, image, logit, roi, od_pred, C, HW, total);
  return wtpse_status();
}
extern "C" int wtpse_adam(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                          double eps, int step, const int* step_dev, void* stream) {
  WTPSE_REQUIRE(p && g && m && v && n > 0 && step >= 1);
  float bc1 = (float)(1.0 - pow(beta1, (double)step));
  float bc2s = (float)sqrt(1.0 - pow(beta2, (double)step));
  hipLaunchKernelGGL(adam_k, GRID1(n), dim3(256), 0, ST, p, g, m, v, n, (float)lr, (float)beta1, (float)beta2, (float)eps, bc1,
                     bc2s, step, step_dev, beta1, beta2);
  return wtpse_status();
}
extern "C" int wtpse_randn(float* out, long long n, unsigned long long seed, unsigned long long offset,
                           const unsigned long long* offset_dev, void* stream) {
  WTPSE_REQUIRE(out && n > 0 && offset % 4 == 0);
  hipLaunchKernelGGL(randn_k, GRID1((n + 3) / 4), dim3(256), 0, ST, out, n, seed, offset, offset_dev);
  return wtpse_status();
}
__global__ void counter_add_k(void* ctr, long long inc, int is64) {
  if (is64) *(unsigned long long*)ctr += (unsigned long long)inc;
  else *(int*)ctr += (int)inc;
}
extern "C" int wtpse_counter_add(void* counter, long long inc, int is64, void* stream) {
  WTPSE_REQUIRE(counter);
  hipLaunchKernelGGL(counter_add_k, dim3(1), dim3(1), 0, ST, counter, inc, is64);
  return wtpse_status();
}
