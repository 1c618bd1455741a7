// The "WT" (whitening-transform) loss of WT-PSE: compute_whitening_loss + compute_MMD
// (reference algorithms.py:1277-1309, 59-121; shape_networks.py:561-594, 240-309).
//
// Forward   G_b = z_b z_b^T / (HW-1) + eps*I               (z_b is [16, HW]; uncentred)
//           off_b  = sum_{i<j} |G_ij|        ins_off  = mean_b clamp((off_b  - margin)/120, 0)
//           diag_b = sum_i |G_ii - 1|        ins_diag = mean_b clamp((diag_b - margin)/16,  0)
//           v_b = G_b[i<j] (120 values, row-major triu order)    dom = pairwise Gaussian-kernel MMD over domain row blocks
// Backward  dz_b = (dG_b + dG_b^T) z_b / (HW-1)
//
// The only HBM-sized work is reading z (forward) and reading z + writing dz (backward):
// 4.19 MB per image per map forward at 256x256 — 8 flop/B, i.e. purely bandwidth-bound.  The Gram itself is
// done on the matrix cores with A == B == the same register (the 16x16x4 fp32 MFMA takes a [16 ch x 4 pixel]
// fragment for both operands), so the kernel is load -> MFMA with no LDS staging and no shuffles.
#include "common.h"

#define WT_C 16
#define WT_NV 120

// ------------------------------------------------------------------------------------------------ forward: partial Grams
template <bool VEC>
__global__ __launch_bounds__(256) void gram_partial_k(const float* __restrict__ z, int HW, int S, int chunk,
                                                      float* __restrict__ partial) {
  __shared__ float red[4 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / S, s = blockIdx.x - b * S;
  const int c = lane & 15, q = lane >> 4;
  const int p_begin = s * chunk;
  const int p_end = min(HW, p_begin + chunk);
  const float* row = z + ((size_t)b * WT_C + c) * HW;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  // a wave-iteration covers 8 groups of 16 pixels (two 64-pixel spans 256 apart): lane (c, q) loads pixels
  // [g*16 + q*4, +4) of channel c; all 8 float4 loads are issued before the first MFMA
  for (int p0 = p_begin + wave * 64; p0 < p_end; p0 += 512) {
    float v[8][4];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      int p = p0 + (g >> 2) * 256 + (g & 3) * 16 + q * 4;
      if (VEC) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < p_end) t = *reinterpret_cast<const float4*>(row + p);
        v[g][0] = t.x; v[g][1] = t.y; v[g][2] = t.z; v[g][3] = t.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[g][i] = (p + i < p_end) ? row[p + i] : 0.f;
      }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      acc0 = mfma16(v[g][0], v[g][0], acc0);
      acc1 = mfma16(v[g][1], v[g][1], acc1);
      acc2 = mfma16(v[g][2], v[g][2], acc2);
      acc3 = mfma16(v[g][3], v[g][3], acc3);
    }
  }
  f32x4 acc = acc0 + acc1 + acc2 + acc3;
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (q * 4 + r) * 16 + c] = acc[r];
  __syncthreads();
  partial[(size_t)blockIdx.x * 256 + tid] = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
}

__device__ __forceinline__ int triu_index(int i, int j) {  // position of (i<j) in torch.triu_indices(16,16,1) order
  return i * (2 * WT_C - i - 1) / 2 + (j - i - 1);
}


// one workgroup per image: fold the S partials, finish G, emit v and the two L1 sums.  NW waves (4 | 16): per-tile partials
// from a conv epilogue (wtpse_conv_fwd_gram) come in hundreds per image (8 MB per call at B = 32, 256 x 256), and one
// workgroup per image only streams them fast enough with sixteen waves' loads in flight.
// `ticket` (optional): zeroed here for the launch that follows in the stream (mmd_final_k).
template <int NW>
__global__ __launch_bounds__(64 * NW) void gram_finalize_k(const float* __restrict__ partial, int S, int HW, float eps,
                                                           float* __restrict__ gram, float* __restrict__ v,
                                                           float* __restrict__ offdiag, float* __restrict__ diag,
                                                           unsigned* __restrict__ ticket) {
  __shared__ float fold[NW][256];
  __shared__ float shw[2][4];
  const int b = blockIdx.x, t = threadIdx.x;
  if (ticket && b == 0 && t == 0) *ticket = 0u;
  {
    // the waves take every NW-th partial with one 16-byte load per lane (a 1 KB row per wave and load, two rows in flight),
    // then meet in LDS; fixed order: bitwise reproducible
    const int q = t >> 6, l = t & 63;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    int k = q;
    for (; k + NW < S; k += 2 * NW) {
      a0 += *reinterpret_cast<const f32x4*>(partial + ((size_t)b * S + k) * 256 + 4 * l);
      a1 += *reinterpret_cast<const f32x4*>(partial + ((size_t)b * S + k + NW) * 256 + 4 * l);
    }
    for (; k < S; k += NW) a0 += *reinterpret_cast<const f32x4*>(partial + ((size_t)b * S + k) * 256 + 4 * l);
    a0 += a1;
#pragma unroll
    for (int e = 0; e < 4; ++e) fold[q][4 * l + e] = a0[e];
  }
  __syncthreads();
  float so = 0.f, sd = 0.f;
  if (t < 256) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) s += fold[q][t];
    const int i = t >> 4, j = t & 15;
    const float g = s / (float)(HW - 1) + (i == j ? eps : 0.f);
    gram[(size_t)b * 256 + t] = g;
    if (i < j) v[(size_t)b * WT_NV + triu_index(i, j)] = g;
    so = wave_xor_sum(i < j ? fabsf(g) : 0.f, 32);
    sd = wave_xor_sum(i == j ? fabsf(g - 1.f) : 0.f, 32);
    if ((t & 63) == 0) {
      shw[0][t >> 6] = so;
      shw[1][t >> 6] = sd;
    }
  }
  __syncthreads();
  if (t == 0) {
    offdiag[b] = (shw[0][0] + shw[0][1]) + (shw[0][2] + shw[0][3]);
    diag[b] = (shw[1][0] + shw[1][1]) + (shw[1][2] + shw[1][3]);
  }
}

// ------------------------------------------------------------------------------------------------ MMD (compute_MMD.forward)
// One workgroup per row i of v.  With D domains of n rows each and npairs = D(D-1)/2:
//   mmd = (1/npairs) * [ (D-1) * sum_a mean(K_aa) - 2 * sum_{a<b} mean(K_ab) ],  K_ij = exp(-max(|x_i|^2+|x_j|^2-2 x_i.x_j, 1e-30))
//   rowval_i = sum_j c_ij K_ij,  c_ij = (D-1)/(n^2 npairs) within a domain, -1/(n^2 npairs) across domains
//   d mmd / d x_i = (4/(n^2 npairs)) * sum_j s_ij K_ij (x_i - x_j),  s_ij = -(D-1) within, +1 across
// Arithmetic is fp64: the value is a difference of O(1) kernel means that is ~1e-6 at initialisation.
// `fin` (optional): the whole tail in this launch.  The workgroup whose row value arrives last (an agent-scope ticket) also
// forms the three loss values that wt_final_k would: one dependent launch (and its ~5 us of latency) fewer per loss call.
struct WtFinalArgs {
  const float* offdiag;
  const float* diag;
  int B, Bnorm;
  float margin;
  float* losses;
  unsigned* ticket;     // zero when the launch starts (gram_finalize_k); left at zero for the next call
};

__global__ __launch_bounds__(128) void mmd_rows_k(const float* __restrict__ v, int D, int n, double* __restrict__ rowval,
                                                  float* __restrict__ dmmd_dv, WtFinalArgs fin) {
  extern __shared__ double kbuf[];  // [R] signed kernel weights s_ij * K_ij
  __shared__ double shred[2];
  __shared__ int last_s;
  const int R = D * n;
  const int i = blockIdx.x, t = threadIdx.x;
  const float* xi = v + (size_t)i * WT_NV;
  const int di = i / n;
  const double npairs = D > 1 ? 0.5 * D * (D - 1) : 1.0;
  const double inv = 1.0 / ((double)n * n * npairs);
  double ni = 0.0;
  for (int k = 0; k < WT_NV; ++k) ni += (double)xi[k] * xi[k];
  double part = 0.0;
  for (int j = t; j < R; j += 128) {
    const float* xj = v + (size_t)j * WT_NV;
    double nj = 0.0, dot = 0.0;
    for (int k = 0; k < WT_NV; ++k) {
      nj += (double)xj[k] * xj[k];
      dot += (double)xi[k] * xj[k];
    }
    double dist = ni + nj - 2.0 * dot;
    bool clamped = dist < 1e-30;
    double K = exp(-(clamped ? 1e-30 : dist));
    bool same = (j / n) == di;
    part += (same ? (double)(D - 1) : -1.0) * K;
    kbuf[j] = clamped ? 0.0 : (same ? -(double)(D - 1) : 1.0) * K;
  }
  // block reduce (2 waves)
  for (int m = 1; m < 64; m <<= 1) part += __shfl_xor(part, m, 64);
  if ((t & 63) == 0) shred[t >> 6] = part;
  __syncthreads();
  const double rv = D > 1 ? (shred[0] + shred[1]) * inv : 0.0;
  if (t == 0) {
    if (fin.ticket) {
      // publish the row value write-through (an 8-byte agent-scope store), drain it, then take a ticket: the last arriver
      // reads every row value with agent-scope loads (MI355X_MICROARCH.md, inter-workgroup visibility: all-sc1 hand-off)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(rowval) + i, (unsigned long long)__double_as_longlong(rv),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned old = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_s = old == (unsigned)(R - 1);
    } else {
      rowval[i] = rv;
      last_s = 0;
    }
  }
  if (t < WT_NV) {
    double g = 0.0;
    double xik = xi[t];
    for (int j = 0; j < R; ++j) g += kbuf[j] * (xik - (double)v[(size_t)j * WT_NV + t]);
    dmmd_dv[(size_t)i * WT_NV + t] = D > 1 ? (float)(4.0 * inv * g) : 0.f;
  }
  __syncthreads();
  if (!last_s) return;
  // ---- the last row's workgroup: the three loss values (as wt_final_k; offdiag / diag come from the previous launch)
  double a = 0.0, d = 0.0, m = 0.0;
  for (int b = t; b < fin.B; b += 128) {
    a += fmaxf((fin.offdiag[b] - fin.margin) / (float)WT_NV, 0.f);
    d += fmaxf((fin.diag[b] - fin.margin) / (float)WT_C, 0.f);
  }
  for (int r = t; r < R; r += 128)
    m += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(rowval) + r, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
  for (int k = 1; k < 64; k <<= 1) {
    a += __shfl_xor(a, k, 64);
    d += __shfl_xor(d, k, 64);
    m += __shfl_xor(m, k, 64);
  }
  __shared__ double fsh[3][2];
  if ((t & 63) == 0) {
    fsh[0][t >> 6] = a;
    fsh[1][t >> 6] = d;
    fsh[2][t >> 6] = m;
  }
  __syncthreads();
  if (t == 0) {
    fin.losses[0] = (float)((fsh[0][0] + fsh[0][1]) / fin.Bnorm);
    fin.losses[1] = (float)((fsh[1][0] + fsh[1][1]) / fin.Bnorm);
    fin.losses[2] = (float)(fsh[2][0] + fsh[2][1]);
    __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// losses[0] = ins_off, [1] = ins_diag, [2] = dom
// Bnorm: the batch size the two instance means divide by (the global batch under data parallelism)
__global__ __launch_bounds__(256) void wt_final_k(const float* __restrict__ offdiag, const float* __restrict__ diag, int B,
                                                  int Bnorm, float margin, const double* __restrict__ rowval, int R,
                                                  float* __restrict__ losses) {
  __shared__ double sh[3][4];
  const int t = threadIdx.x;
  double a = 0.0, d = 0.0, m = 0.0;
  for (int b = t; b < B; b += 256) {
    a += fmaxf((offdiag[b] - margin) / (float)WT_NV, 0.f);
    d += fmaxf((diag[b] - margin) / (float)WT_C, 0.f);
  }
  for (int i = t; i < R; i += 256) m += rowval[i];
  for (int k = 1; k < 64; k <<= 1) {
    a += __shfl_xor(a, k, 64);
    d += __shfl_xor(d, k, 64);
    m += __shfl_xor(m, k, 64);
  }
  if ((t & 63) == 0) {
    sh[0][t >> 6] = a;
    sh[1][t >> 6] = d;
    sh[2][t >> 6] = m;
  }
  __syncthreads();
  if (t == 0) {
    losses[0] = (float)((sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) / Bnorm);
    losses[1] = (float)((sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) / Bnorm);
    losses[2] = (float)(sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3]);
  }
}

// Fold the per-map losses of one update() the way the reference does (losses: [nmaps][3] = off, diag, dom).
//   mode 0 — WT_PSE.update (algorithms.py:1259-1267): ins = sum_m (off_m + diag_m) / den ; dom = sum_m dom_m / den
//   mode 1 — ShapeVariationalDist_x.update (shape_networks.py:545-554) incl. the accumulator overwrite at :546-548:
//            ins_off = sum_m off_m / den ; ins_diag = 2 * diag_last / den ; ins = ins_off + ins_diag
// out = (ins_total, ins_off, ins_diag, dom); den = len(list) = 3 although only 2 maps are summed.
__global__ void wt_combine_k(const float* __restrict__ losses, int nmaps, float den, int mode, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float off = 0.f, diag = 0.f, dom = 0.f, tot = 0.f;
  for (int m = 0; m < nmaps; ++m) {
    float o = losses[3 * m], d = losses[3 * m + 1];
    off += o;
    diag = mode ? d + d : diag + d;
    tot += o + d;
    dom += losses[3 * m + 2];
  }
  off /= den; diag /= den; dom /= den; tot /= den;
  out[0] = mode ? off + diag : tot;
  out[1] = off; out[2] = diag; out[3] = dom;
}

// ------------------------------------------------------------------------------------------------ backward
// M_b = (dG_b + dG_b^T) / (HW-1) where
//   dG_ij (i<j) = w_off * g_off * sign(G_ij) / (120 B) * [off_b clamp active] + w_dom * g_dom * dmmd_dv[b][ij]
//   dG_ii       = w_diag * g_diag * sign(G_ii - 1) / (16 B) * [diag_b clamp active]
// g_* are device scalars (upstream autograd gradients; no host sync), w_* host-side loss weights.
__global__ __launch_bounds__(256) void wt_dgram_k(const float* __restrict__ gram, const float* __restrict__ offdiag,
                                                  const float* __restrict__ diag, const float* __restrict__ dmmd_dv,
                                                  int B, int R, int HW, float margin, const float* g_off,
                                                  const float* g_diag, const float* g_dom, float w_off, float w_diag,
                                                  float w_dom, float* __restrict__ M) {
  const int b = blockIdx.x, t = threadIdx.x;
  const int i = t >> 4, j = t & 15;
  const int lo = min(i, j), hi = max(i, j);
  const float go = (g_off ? *g_off : 1.f) * w_off, gd = (g_diag ? *g_diag : 1.f) * w_diag,
              gm = (g_dom ? *g_dom : 1.f) * w_dom;
  float val;
  if (i == j) {
    float x = gram[(size_t)b * 256 + t] - 1.f;
    float sgn = (x > 0.f) - (x < 0.f);
    bool active = (diag[b] - margin) / (float)WT_C >= 0.f;
    val = active ? 2.f * gd * sgn / ((float)WT_C * B) : 0.f;  // dG + dG^T doubles the diagonal
  } else {
    float x = gram[(size_t)b * 256 + lo * 16 + hi];
    float sgn = (x > 0.f) - (x < 0.f);
    bool active = (offdiag[b] - margin) / (float)WT_NV >= 0.f;
    val = active ? go * sgn / ((float)WT_NV * B) : 0.f;
    if (b < R) val += gm * dmmd_dv[(size_t)b * WT_NV + triu_index(lo, hi)];
  }
  M[(size_t)b * 256 + t] = val / (float)(HW - 1);
}

// dz[b][c][p] (+)= sum_c' M[b][c][c'] z[b][c'][p] : read z once, write dz once, M through the scalar cache
template <bool VEC>
__global__ __launch_bounds__(256) void gram_bwd_k(const float* __restrict__ z, const float* __restrict__ M, int HW,
                                                  int blocks_per_img, int accumulate, float* __restrict__ dz) {
  const int b = blockIdx.x / blocks_per_img;
  const int blk = blockIdx.x - b * blocks_per_img;
  const float* Mb = M + (size_t)b * 256;
  constexpr int PPT = VEC ? 4 : 1;
  const int p = (blk * 256 + threadIdx.x) * PPT;
  if (p >= HW) return;
  const float* zb = z + (size_t)b * WT_C * HW + p;
  float* db = dz + (size_t)b * WT_C * HW + p;
  float in[WT_C][PPT];
#pragma unroll
  for (int c = 0; c < WT_C; ++c) {
    if (VEC) {
      float4 t = *reinterpret_cast<const float4*>(zb + (size_t)c * HW);
      in[c][0] = t.x; in[c][1] = t.y; in[c][2] = t.z; in[c][3] = t.w;
    } else {
      in[c][0] = zb[(size_t)c * HW];
    }
  }
#pragma unroll
  for (int c = 0; c < WT_C; ++c) {
    float o[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) o[k] = 0.f;
#pragma unroll
    for (int cc = 0; cc < WT_C; ++cc) {
      float m = Mb[c * 16 + cc];
#pragma unroll
      for (int k = 0; k < PPT; ++k) o[k] = fmaf(m, in[cc][k], o[k]);
    }
    if (VEC) {
      float4* dst = reinterpret_cast<float4*>(db + (size_t)c * HW);
      float4 t = make_float4(o[0], o[1], o[2], o[3]);
      if (accumulate & 1) {
        float4 old = *dst;
        if (accumulate & 2) {      // the incoming gradient is wrt relu(z): mask it with [z > 0] (z is in registers)
          old.x = in[c][0] > 0.f ? old.x : 0.f; old.y = in[c][1] > 0.f ? old.y : 0.f;
          old.z = in[c][2] > 0.f ? old.z : 0.f; old.w = in[c][3] > 0.f ? old.w : 0.f;
        }
        t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w;
      }
      *dst = t;
    } else {
      float old = (accumulate & 1) ? db[(size_t)c * HW] : 0.f;
      if ((accumulate & 2) && !(in[c][0] > 0.f)) old = 0.f;
      db[(size_t)c * HW] = old + o[0];
    }
  }
}

// ------------------------------------------------------------------------------------------------ C ABI
// The tail of a forward call in two launches: per-image fold (+ G, v, L1 sums), then the MMD rows with the final sums taken by
// the last row's workgroup.  rowval holds D*n doubles plus one 8-byte ticket word.
static void wt_tail_launch(const float* partial, int S, int B, int HW, float eps, float margin, int domain_num, int per_domain,
                           float* gram, float* v, float* offdiag, float* diag, double* rowval, float* dmmd_dv, float* losses,
                           hipStream_t st) {
  const int R = domain_num * per_domain;
  unsigned* ticket = reinterpret_cast<unsigned*>(rowval + R);
  if (S >= 32)
    hipLaunchKernelGGL(gram_finalize_k<16>, dim3(B), dim3(1024), 0, st, partial, S, HW, eps, gram, v, offdiag, diag, ticket);
  else
    hipLaunchKernelGGL(gram_finalize_k<4>, dim3(B), dim3(256), 0, st, partial, S, HW, eps, gram, v, offdiag, diag, ticket);
  WtFinalArgs fin{offdiag, diag, B, B, margin, losses, ticket};
  hipLaunchKernelGGL(mmd_rows_k, dim3(R), dim3(128), R * sizeof(double), st, v, domain_num, per_domain, rowval, dmmd_dv, fin);
}

extern "C" int wtpse_wt_split(int B, int HW, int* chunk_out) {
  // aim for ~768 workgroups (3 per CU) of at least 2048 pixels, chunks a multiple of 512 pixels (one full
  // wave-iteration per wave): few, fat partials keep the per-image fold short
  int target = 768 / (B > 0 ? B : 1);
  if (target < 1) target = 1;
  int chunk = ceil_div(ceil_div(HW, target), 512) * 512;
  if (chunk < 2048) chunk = 2048;
  int S = ceil_div(HW, chunk);
  if (chunk_out) *chunk_out = chunk;
  return S;
}

extern "C" int wtpse_wt_loss_fwd(const float* z, int B, int C, int HW, float eps, float margin, int domain_num,
                                 int per_domain, float* partial, float* gram, float* v, float* offdiag, float* diag,
                                 double* rowval, float* dmmd_dv, float* losses, void* stream) {
  WTPSE_REQUIRE(z && partial && gram && v && offdiag && diag && rowval && dmmd_dv && losses);
  WTPSE_REQUIRE(C == WT_C && B > 0 && HW > 1 && domain_num >= 1 && per_domain >= 1);
  const int R = domain_num * per_domain;
  WTPSE_REQUIRE(R <= B);
  hipStream_t st = (hipStream_t)stream;
  int chunk;
  const int S = wtpse_wt_split(B, HW, &chunk);
  const bool vec = (HW % 4 == 0) && (((uintptr_t)z & 15) == 0);
  if (vec)
    hipLaunchKernelGGL(gram_partial_k<true>, dim3(B * S), dim3(256), 0, st, z, HW, S, chunk, partial);
  else
    hipLaunchKernelGGL(gram_partial_k<false>, dim3(B * S), dim3(256), 0, st, z, HW, S, chunk, partial);
  wt_tail_launch(partial, S, B, HW, eps, margin, domain_num, per_domain, gram, v, offdiag, diag, rowval, dmmd_dv, losses, st);
  return wtpse_status();
}

// The same loss from per-tile partial Grams produced elsewhere (the epilogue of the conv that wrote z:
// wtpse_conv_fwd_gram): partial [B][S][256].  z itself is not read: 16*HW*4 bytes per image stay in HBM untouched.
extern "C" int wtpse_wt_loss_fwd_partials(const float* partial, int S, int B, int HW, float eps, float margin, int domain_num,
                                          int per_domain, float* gram, float* v, float* offdiag, float* diag, double* rowval,
                                          float* dmmd_dv, float* losses, void* stream) {
  WTPSE_REQUIRE(partial && gram && v && offdiag && diag && rowval && dmmd_dv && losses);
  WTPSE_REQUIRE(S >= 1 && B > 0 && HW > 1 && domain_num >= 1 && per_domain >= 1);
  const int R = domain_num * per_domain;
  WTPSE_REQUIRE(R <= B);
  hipStream_t st = (hipStream_t)stream;
  wt_tail_launch(partial, S, B, HW, eps, margin, domain_num, per_domain, gram, v, offdiag, diag, rowval, dmmd_dv, losses, st);
  return wtpse_status();
}

extern "C" int wtpse_wt_loss_bwd(const float* z, int B, int C, int HW, float margin, int domain_num, int per_domain,
                                 const float* gram, const float* offdiag, const float* diag, const float* dmmd_dv,
                                 const float* g_off, const float* g_diag, const float* g_dom, float w_off, float w_diag,
                                 float w_dom, float* Mws, float* dz, int accumulate, void* stream) {
  WTPSE_REQUIRE(z && gram && offdiag && diag && dmmd_dv && Mws && dz);
  WTPSE_REQUIRE(C == WT_C && B > 0 && HW > 1);
  const int R = domain_num * per_domain;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wt_dgram_k, dim3(B), dim3(256), 0, st, gram, offdiag, diag, dmmd_dv, B, R, HW, margin, g_off, g_diag,
                     g_dom, w_off, w_diag, w_dom, Mws);
  const bool vec = (HW % 4 == 0) && (((uintptr_t)z & 15) == 0) && (((uintptr_t)dz & 15) == 0);
  if (vec) {
    int bpi = ceil_div(HW, 1024);
    hipLaunchKernelGGL(gram_bwd_k<true>, dim3(B * bpi), dim3(256), 0, st, z, Mws, HW, bpi, accumulate, dz);
  } else {
    int bpi = ceil_div(HW, 256);
    hipLaunchKernelGGL(gram_bwd_k<false>, dim3(B * bpi), dim3(256), 0, st, z, Mws, HW, bpi, accumulate, dz);
  }
  return wtpse_status();
}

// The two halves of wtpse_wt_loss_fwd, for the data-parallel path (all-gather of v between them):
//   wtpse_wt_gram_fwd : z -> gram, v, offdiag, diag of this rank's images
//   wtpse_wt_final    : losses[0..1] = this rank's share of the instance means (divide by Bnorm = global batch),
//                       losses[2] = sum(rowval[0..R)) (the MMD of the gathered rows, from wtpse_mmd_fwd)
extern "C" int wtpse_wt_gram_fwd(const float* z, int B, int C, int HW, float eps, float* partial, float* gram, float* v,
                                 float* offdiag, float* diag, void* stream) {
  WTPSE_REQUIRE(z && partial && gram && v && offdiag && diag && C == WT_C && B > 0 && HW > 1);
  hipStream_t st = (hipStream_t)stream;
  int chunk;
  const int S = wtpse_wt_split(B, HW, &chunk);
  const bool vec = (HW % 4 == 0) && (((uintptr_t)z & 15) == 0);
  if (vec)
    hipLaunchKernelGGL(gram_partial_k<true>, dim3(B * S), dim3(256), 0, st, z, HW, S, chunk, partial);
  else
    hipLaunchKernelGGL(gram_partial_k<false>, dim3(B * S), dim3(256), 0, st, z, HW, S, chunk, partial);
  hipLaunchKernelGGL(gram_finalize_k<4>, dim3(B), dim3(256), 0, st, partial, S, HW, eps, gram, v, offdiag, diag, (unsigned*)nullptr);
  return wtpse_status();
}

extern "C" int wtpse_wt_final(const float* offdiag, const float* diag, int B, int Bnorm, float margin, const double* rowval,
                              int R, float* losses, void* stream) {
  WTPSE_REQUIRE(offdiag && diag && rowval && losses && B > 0 && Bnorm >= B && R >= 0);
  hipLaunchKernelGGL(wt_final_k, dim3(1), dim3(256), 0, (hipStream_t)stream, offdiag, diag, B, Bnorm, margin, rowval, R, losses);
  return wtpse_status();
}

extern "C" int wtpse_wt_combine(const float* losses, int nmaps, float den, int mode, float* out, void* stream) {
  WTPSE_REQUIRE(losses && out && nmaps > 0 && den > 0.f);
  hipLaunchKernelGGL(wt_combine_k, dim3(1), dim3(64), 0, (hipStream_t)stream, losses, nmaps, den, mode, out);
  return wtpse_status();
}

// MMD alone on a [R, 120] matrix (the data-parallel path all-gathers v and calls this on the global batch)
extern "C" int wtpse_mmd_fwd(const float* v, int domain_num, int per_domain, double* rowval, float* dmmd_dv, void* stream) {
  WTPSE_REQUIRE(v && rowval && dmmd_dv && domain_num >= 1 && per_domain >= 1);
  const int R = domain_num * per_domain;
  hipLaunchKernelGGL(mmd_rows_k, dim3(R), dim3(128), R * sizeof(double), (hipStream_t)stream, v, domain_num, per_domain,
                     rowval, dmmd_dv, WtFinalArgs{nullptr, nullptr, 0, 0, 0.f, nullptr, nullptr});
  return wtpse_status();
}
