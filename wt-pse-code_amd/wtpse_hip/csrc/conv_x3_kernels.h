// Kernels of the x3 / x2h convolutions (conv_x3.hip holds the host side).  Included by conv_x3_t1/t2/t3.hip, which instantiate the
// kernels for ONE value of TERMS each (three translation units compile side by side), and by conv_x3.hip for the argument struct,
// the split helpers and the packed-weight header.
#pragma once
// 3x3 / 1x1 convolution, fp32 in / fp32 out, on the BF16 matrix cores of gfx950 at fp32 accuracy ("x3" path).
//
// Same contract as conv.hip's conv_fwd_k (forward and data gradient of the nn.Conv2d dispatches of the reference hot path,
// algorithms.py:882-888,926-933 ...; same loader / epilogue fusions), different arithmetic.  The fp32-input MFMA runs at
// 1/16 of the bf16 rate, so every fp32 operand x is split into three bf16 terms x = x0 + x1 + x2 (each the round-to-nearest-even
// bf16 of what the previous terms left; the remainders are exact in fp32) and the product is formed from the six leading
// cross terms
//       a*b ~= a0*b0 + (a0*b1 + a1*b0) + (a0*b2 + a1*b1 + a2*b0)
// each a v_mfma_f32_32x32x16_bf16 with fp32 accumulation: every bf16 x bf16 product is exact in fp32, the dropped terms are
// below 2^-24 |a b| and unbiased, and the accumulated error measures the same as the fp32 MFMA's (tests/test_kernels_gpu.py, CPU
// emulation in DESIGN.md).  Six bf16 MFMAs (6 x 32 cycles) replace eight fp32 MFMAs (8 x 64 cycles) per 32x32x16 block.
//
// GEMM orientation as in conv.hip: D[cout][pixel] += sum_tap W_tap[cout][cin] * X[cin][pixel + tap], one GEMM per tap with
// K = 16 input channels per MFMA.  Operand images in LDS (bf16, 16-byte rows of 8 consecutive k so that one ds_read_b128 is a
// lane's fragment, and consecutive lanes read consecutive 16-byte slots: conflict-free):
//     Xs[term 3][k-half 2][halo position][8 cin]      the 16-channel chunk of the input tile, split on the way in
//     Ws[tap 3][term 3][k-half 2][cout CB][8 cin]     one kernel row of the weights (pre-split by the pack kernel)

#include <stdlib.h>
#include "common.h"
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

struct ConvX3Args {
  const float* in0;
  const float* in1;
  const unsigned short* wx;   // packed split weights, see pack_weights_x3_k
  const float* bias;
  const float* pro0;
  const float* pro1;
  float* out0;
  float* out1;
  float* stats;
  const float* mask;
  // TERMS 2 (two fp16 terms per operand): the input is multiplied by a power of two as it is loaded: the one that brings the largest
  // magnitude of the input AS LOADED (after the prologue) into [2^14, 2^15), read from amax tables (common.h) — in_amax for in0 (a
  // gradient's amax from its producer / wtpse_amax; a forward activation's bound from the BatchNorm finalize, the producer's epilogue
  // or wtpse_act_bound), in_amax1 for in1, the larger of the two counts — or, with neither, in_scale; the result is scaled back in the epilogue
  const unsigned* in_amax;
  const unsigned* in_amax1;
  float in_scale;
  // EPI 0: the amax table (zero on entry) of the STORED output (after bias and output ReLU), or null — what a consumer without a
  // train-mode BatchNorm in between scales by (eval-mode BatchNorm, un-normalised maps)
  unsigned* out_amax;
  // EPI == 2 (BatchNorm backward statistics in a data gradient's epilogue): output channels [bn_c0, bn_c1) are the gradient
  // wrt the activated output of a conv + BatchNorm (+ReLU) layer whose raw conv output is `mask` ([B][bn_c1 - bn_c0][H][W]):
  // they are masked with [mask * scale + shift > 0] (bn_relu) and (sum g, sum g * (y - mean)) partials go to `stats`
  const float* bn_ss;         // [bn_c1 - bn_c0][2]
  const float* bn_mean;       // [bn_c1 - bn_c0]
  int bn_c0, bn_c1, bn_relu;
  BnbTail tail;               // EPI == 2: the BatchNorm-backward coefficients from the last workgroups (common.h), or tickets == null
  BnfTail ftail;              // forward statistics: BatchNorm finalize by the last workgroups (common.h), or tickets == null
  int B, H, W;
  int C0, C1, Cin, CinP;      // CinP: multiple of 16
  int Cout, CoutP, Csplit;    // CoutP: multiple of 32
  int pro_relu, relu_out;
  int tiles_x, tiles_y;
  // XCD-aware workgroup order (launch_x3): dispatch slot L = blockIdx.y * gridDim.x + blockIdx.x goes to XCD L % 8 (round robin);
  // with xcd_tiles = gridDim.x / 8 > 0 XCD q works through the tiles [q * xcd_tiles, (q + 1) * xcd_tiles), the output-channel blocks
  // of a tile in consecutive slots — one L2 then holds a tile's input for all the blocks that read it and for the neighbours that
  // share its halo, instead of every XCD fetching every tile once per block (measured: x3_conv HBM reads 205 -> see DESIGN.md)
  int xcd_tiles;
};

// (tile index, output-channel block) of this workgroup
__device__ __forceinline__ void x3_block_ids(const ConvX3Args& a, int& tile, int& cblk) {
  tile = blockIdx.x;
  cblk = blockIdx.y;
  if (a.xcd_tiles > 0) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x, s = L >> 3;
    cblk = s % (int)gridDim.y;
    tile = (L & 7) * a.xcd_tiles + s / (int)gridDim.y;
  }
}

template <class F, int... I>
__device__ __forceinline__ void x3_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void x3_static_for(F&& f) {
  x3_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- the "x2h" arithmetic (round 5): TWO fp16 terms per fp32 operand, THREE products.
//       x = h0 + h1,  h0 = fp16(x), h1 = fp16(x - h0)          a*b ~= a0*b1 + a1*b0 + a0*b0      (v_mfma_f32_32x32x16_f16)
// fp16 carries 11 significant bits: h0 + h1 reproduces an fp32 operand to 22-24 bits (the remainder x - h0 has at most 13 significant
// bits, h1 keeps 11 of them: the representation error is 0 or +-1 ulp of x, rms 2^-24.5 |x|), the dropped a1*b1 is below 2^-22 |a b|
// and, every term being rounded, unbiased — per product ~2x the rounding error of one fp32 multiply, far below the fp32 accumulation
// error that dominates both arithmetics (measured against fp64 beside the x3 and fp32-MFMA kernels: tests/test_conv_x3_gpu.py).
// Half the MFMAs of x3, two thirds of its LDS image, fragment reads and split instructions.  What fp16 lacks is RANGE (5 exponent
// bits): h1 keeps its 11 bits only while |h0| >= 2^-3 and fp16 overflows at 65504.  So every operand tensor is multiplied by a power
// of two on the way in (exact) and the accumulators by the inverse on the way out:
//   * weights: per layer and direction, from the layer's largest magnitude (pack_weights_x3_k; header in front of the packed block);
//   * forward activations: from a bound of the tensor's largest magnitude (ConvX3Args::in_amax / in_amax1; common.h, bn_act_bound:
//     |gamma| sqrt(N - 1) + |beta| behind a train-mode BatchNorm, the data's amax otherwise) — the precision of round 5's fixed 2^2 on
//     O(1) data, at any scale of the data; without a table (a bare C-ABI caller): X3_FWD_SCALE = 2^2, full precision for
//     2^-5 <= |x| < 2^14, and a value beyond 2^14 overflows to inf and turns the outputs it feeds into NaN (common.h: split2h_pair);
//   * gradients (no scale known a priori): from the tensor's largest magnitude, left by its producer in ConvX3Args::in_amax.
// In two halves (common.h, amax_load / amax_reduce): x3_in_scale_issue() in front of the workgroup's first tile loads, x3_in_scale()
// behind them.
template <int TERMS>
__device__ __forceinline__ unsigned x3_in_scale_issue(const ConvX3Args& a) {
  if constexpr (TERMS != 2) return 0u;
  else return max(amax_load(a.in_amax), amax_load(a.in_amax1));
}
template <int TERMS>
__device__ __forceinline__ float x3_in_scale(const ConvX3Args& a, unsigned issued) {
  if constexpr (TERMS != 2) return 1.f;
  else {
    if (!a.in_amax && !a.in_amax1) return a.in_scale;
    return x3_scale_from_amax(amax_reduce(issued));
  }
}

// (a, b) -> three dwords, each holding the bf16 pair (term_i(a), term_i(b)), i = 0, 1, 2.  Every term is rounded to nearest
// even (v_cvt_pk_bf16_f32) and the remainder formed exactly in fp32, so a = a0 + a1 + a2 up to 2^-25 |a| with terms of
// alternating sign: the dropped cross terms (a1*b2 + a2*b1 + a2*b2 ~ 2^-25 |a b|) are unbiased.  (Truncating splits — mask off
// the low 16 bits — cost the same number of instructions but leave every term with the sign of its operand: the dropped terms
// then bias each product towards zero by ~2^-23, a coherent error that the network amplified 10x more than fp32 rounding.)
__device__ __forceinline__ unsigned pack_rne(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pack_rne(a, b);
  const float ra = a - __builtin_bit_cast(float, p0 << 16);
  const float rb = b - __builtin_bit_cast(float, p0 & 0xFFFF0000u);
  p1 = pack_rne(ra, rb);
  const float sa = ra - __builtin_bit_cast(float, p1 << 16);
  const float sb = rb - __builtin_bit_cast(float, p1 & 0xFFFF0000u);
  p2 = pack_rne(sa, sb);
}

template <int TERMS>
__device__ __forceinline__ void x3_split_pair(float v0, float v1, unsigned (&q)[TERMS]) {
  if constexpr (TERMS == 3) split3_pair(v0, v1, q[0], q[1], q[2]);
  else if constexpr (TERMS == 2) split2h_pair(v0, v1, q[0], q[1]);
  else q[0] = pack_rne(v0, v1);
}
template <int TERMS>
__device__ __forceinline__ f32x16 x3_mfma(u32x4v a, u32x4v b, f32x16 c) {
  if constexpr (TERMS == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// TERMS 2: accumulators back to the operands' own scale (exact: powers of two) before bias / statistics / stores
template <int TERMS, int MT, int NT>
__device__ __forceinline__ void x3_unscale(const ConvX3Args& a, f32x16 (&acc)[MT][NT], float sx) {
  if constexpr (TERMS == 2) {
    const float inv = reinterpret_cast<const float*>(a.wx)[0] / sx;      // 1 / (weight scale x input scale)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] *= inv;
  }
}

// Epilogue shared by the x3 forward kernels (as conv.hip): + bias, ReLU / ReLU mask, branch-free buffer stores, BatchNorm
// (sum, sum^2) partials into row `stats_row`.  `tid` counts within the 256 threads that own the tile; `live` = false drops
// every store (a padding tile).
// EPI: 0 plain, 1 ReLU mask (out = mask > 0 ? value : 0), 2 BatchNorm-backward statistics (ConvX3Args::bn_*): the gradient is
// masked with the ReLU of the layer it flows into and the two reductions of that layer's BatchNorm backward (reference
// algorithms.py:883-889 via autograd) are formed from the accumulators, so bn_bwd_reduce_k never re-reads the two tensors.
// WM = waves along the output channels (conv_x3r_k: 2 — a wave then owns MT blocks of 32 channels x NT column tiles of 32 pixels of a
// (4 / WM)-wave pixel split; conv_x3_k: 1, every wave holds all CB channels of its pixels).
template <int MT, int NT, int TWL, int EPI, bool RED_ALIASES, int WM = 1>
__device__ __forceinline__ void x3_epilogue(const ConvX3Args& a, f32x16 (&acc)[MT][NT], int b, int ty, int tx, int cout0, int tid,
                                            float* red, const float* bias_s, int stats_row, int cblk, bool live) {
  constexpr int PW = 4 / WM;
  constexpr int TW = 1 << TWL, TH = (PW * 32 * NT) / TW;
  constexpr int CBW = 32 * MT, CB = CBW * WM, NACC = 16;
  constexpr bool MASK = EPI == 1, BNB = EPI == 2;
  // wave: index along the pixels; cw0: first channel of this wave's share (both wave-uniform: pinned to scalar registers, or the
  // channel-dependent descriptors / scalar offsets below turn every store into a readfirstlane loop)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((tid >> 6) / WM);
  const int cw0 = WM == 1 ? 0 : __builtin_amdgcn_readfirstlane(((tid >> 6) % WM) * CBW);
  const int r32 = lane & 31, h = lane >> 5;
  const int HW = a.H * a.W;
  if (a.bias) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const float bz = bias_s[cw0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] += bz;
      }
  }
  const bool want_stats = a.stats != nullptr;
  const float* bnp_s = bias_s + CB;                   // EPI 2: [3][CB] (scale | shift | mean) of this block's channels
  if (want_stats && RED_ALIASES) __syncthreads();   // red[PW waves][CB][2] reuses the operand images
  const int C1out = a.Cout - a.Csplit;
  const int Cbn = a.bn_c1 - a.bn_c0;
  int poff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = wave * (32 * NT) + nt * 32 + r32;
    const int gy = ty * TH + (p >> TWL), gx = tx * TW + (p & (TW - 1));
    poff[nt] = (gy < a.H && gx < a.W) ? gy * a.W + gx : -1;
  }
  const __amdgpu_buffer_rsrc_t rs_o0 = make_rsrc(a.out0 + (size_t)b * a.Csplit * HW, (unsigned)a.Csplit * HW * 4u);
  const __amdgpu_buffer_rsrc_t rs_o1 = a.out1 ? make_rsrc(a.out1 + (size_t)b * C1out * HW, (unsigned)C1out * HW * 4u) : rs_o0;
  const __amdgpu_buffer_rsrc_t rs_m = MASK ? make_rsrc(a.mask + (size_t)b * a.Cout * HW, (unsigned)a.Cout * HW * 4u)
                                      : BNB ? make_rsrc(a.mask + (size_t)b * Cbn * HW, (unsigned)Cbn * HW * 4u) : rs_o0;
  const int clane = h * 4;
  unsigned pvo[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) pvo[nt] = (live && poff[nt] >= 0) ? (unsigned)(clane * HW + poff[nt]) * 4u : BUF_OOB;
  constexpr int NSV = NACC * 2;
  const bool full = ty * TH + TH <= a.H && tx * TW + TW <= a.W && cout0 + CB <= a.Cout;
  if (!full) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const bool cvalid = cout0 + cw0 + mt * 32 + (r & 3) + 8 * (r >> 2) + clane < a.Cout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] = (cvalid && poff[nt] >= 0) ? acc[mt][nt][r] : 0.f;
      }
  }
  // The output ReLU, in place and only where a launch has one (a uniform branch): DeepWT's first / third convolution and the fusion
  // conv — never together with statistics or Gram partials (host checks), so everything below takes the accumulators as they are.
  // (Rounds 1-5 clamped every value of every launch with max(v, relu_lo), relu_lo = -inf without a ReLU: an instruction per value — two in
  // the NaN-preserving form — in epilogues that are bound by their instruction count: profiles/NOTES_r06.md.)
  if (a.relu_out) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < NACC; ++r) acc[mt][nt][r] = out_clamp<EPI>(acc[mt][nt][r], 0.f);
  }
  if constexpr (EPI == 0) {
    if (a.out_amax) {       // largest magnitude of what this wave stores (ragged parts are zero by now): one no-return atomic per wave
      unsigned am = 0u;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < NACC; ++r) am = max(am, amax_bits(acc[mt][nt][r]));
      amax_publish_wave(a.out_amax, live ? am : 0u, (unsigned)stats_row * 4u + (unsigned)(tid >> 6));
    }
  }
  const unsigned hw4 = (unsigned)HW * 4u;
  // A launch that folds its own statistics (bnb_tail / bnf_tail) publishes them and takes its tickets BEFORE it stores its output
  // tile: the hand-off drains the workgroup's outstanding stores (s_waitcnt vmcnt(0)), and with 8-16 K output stores in flight
  // that wait cost the 32-channel variants 9-19 % (measured; the statistics never depended on the stores).  The values to store
  // stay in the accumulators (with statistics there is no output ReLU: host check).
  const bool defer = want_stats && (BNB ? a.tail.tickets != nullptr : a.ftail.tickets != nullptr);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float mk[NACC][NT];
    if (MASK) {
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const int cbase = cout0 + cw0 + mt * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mk[r][nt] = buf_load(rs_m, pvo[nt], (unsigned)min(cbase, a.Cout) * hw4);
      }
    }
    if (BNB) {
      // the channels one register holds across the wave lie in one aligned group of 8 and bn_c0 / bn_c1 are multiples of 16:
      // whether a register belongs to the BatchNorm'd tensor is wave-uniform; the others load out of range (0)
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const int cbase = cout0 + cw0 + mt * 32 + (r & 3) + 8 * (r >> 2);
        const bool bn = cbase >= a.bn_c0 && cbase < a.bn_c1;
        const unsigned soff = (unsigned)(bn ? cbase - a.bn_c0 : 0) * hw4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mk[r][nt] = buf_load(rs_m, bn ? pvo[nt] : BUF_OOB, soff);
      }
    }
    float bmu[NACC];
#pragma unroll
    for (int r = 0; r < NACC; ++r) {
      const int cbase = cout0 + cw0 + mt * 32 + (r & 3) + 8 * (r >> 2);
      const bool second = a.out1 != nullptr && cbase >= a.Csplit;
      const __amdgpu_buffer_rsrc_t rs_o = second ? rs_o1 : rs_o0;
      const unsigned soff = (unsigned)(second ? min(cbase, a.Cout) - a.Csplit : min(cbase, a.Csplit)) * hw4;
      float bsc = 0.f, bsh = 1.f;
      bmu[r] = 0.f;
      if (BNB) {
        const int crel = cw0 + mt * 32 + (r & 3) + 8 * (r >> 2) + clane;
        bsc = bnp_s[crel];
        bsh = bnp_s[CB + crel];
        bmu[r] = bnp_s[2 * CB + crel];
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float v = acc[mt][nt][r];
        if (MASK && !(mk[r][nt] > 0.f)) v = 0.f;
        if (BNB) {     // the ReLU decision of the forward pass: fmaf(y, scale, shift) > 0 (channels outside [bn_c0, bn_c1): 0, 1)
          // (opaque to the vectoriser on purpose: with the two pixels' decisions fused into one v_pk_fma_f32 the masks of a few
          // lanes of the upper half-wave came out wrong in ~10 % of the launches — tools/probe/dbg_bnb.py; scalar
          // v_fma_f32 has been bitwise reproducible over thousands of launches)
          float zz = __builtin_fmaf(mk[r][nt], bsc, bsh);
          asm volatile("" : "+v"(zz));
          if (!(zz > 0.f)) v = 0.f;
          acc[mt][nt][r] = v;
        }
        if (!defer) buf_store(rs_o, pvo[nt], soff, v);
      }
    }
    if (want_stats) {
      float sv[NSV];
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma clang fp contract(off)
          if (BNB) {
            const float v = acc[mt][nt][r];
            s1 += v;
            s2 += v * (mk[r][nt] - bmu[r]);
          } else {
            const float v = acc[mt][nt][r];
            s1 += v;
            s2 += v * v;
          }
        }
        sv[r * 2 + 0] = s1;
        sv[r * 2 + 1] = s2;
      }
      // butterfly transpose-reduction over the 32 lanes that hold one channel's pixels (see conv.hip)
#pragma unroll
      for (int st = 0; st < 5; ++st) {
        const int half = NSV >> (st + 1);
        const bool up = (lane >> st) & 1;
#pragma unroll
        for (int i = 0; i < NSV / 2; ++i) {
          if (i < half) {
            // the empty asm makes the two operands opaque values: otherwise the select of two array elements is rewritten
            // into one element with a selected (dynamic) index, and the register array into 32-way compare/select chains
            float lo = sv[i], hi = sv[i + half];
            asm volatile("" : "+v"(lo), "+v"(hi));
            const float keep = up ? hi : lo;
            const float send = up ? lo : hi;
            sv[i] = keep + __shfl_xor(send, 1 << st, 64);
          }
        }
      }
      int idx = 0;
#pragma unroll
      for (int st = 0; st < 5; ++st) idx += ((lane >> st) & 1) * (NSV >> (st + 1));
      const int k = idx & 1, rr = idx >> 1;
      const int crel = cw0 + mt * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
      red[(wave * CB + crel) * 2 + k] = sv[0];
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < CB * 2) {
      const int crel = tid >> 1;
      float s = red[tid];
#pragma unroll
      for (int q = 1; q < PW; ++q) s += red[q * CB * 2 + tid];
      if (BNB) {
        const int c = cout0 + crel;
        if (live && c >= a.bn_c0 && c < a.bn_c1) pub_store(a.stats + ((size_t)stats_row * Cbn + c - a.bn_c0) * 2 + (tid & 1), s);
      } else if (live && cout0 + crel < a.Cout) {
        pub_store(a.stats + ((size_t)stats_row * a.Cout + cout0 + crel) * 2 + (tid & 1), s);
      }
    }
  }
  TailTicket tk;
  tk.old = 0u;
  tk.armed = 0;
  if (defer) {
    if constexpr (BNB) tk = bnb_tail_begin<CB>(a.tail, a.bn_c0, a.bn_c1, cout0, stats_row, cblk, tid);
    else tk = bnf_tail_begin(a.ftail, stats_row, cblk, tid);
  }
  if (defer) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const int cbase = cout0 + cw0 + mt * 32 + (r & 3) + 8 * (r >> 2);
        const bool second = a.out1 != nullptr && cbase >= a.Csplit;
        const __amdgpu_buffer_rsrc_t rs_o = second ? rs_o1 : rs_o0;
        const unsigned soff = (unsigned)(second ? min(cbase, a.Cout) - a.Csplit : min(cbase, a.Csplit)) * hw4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) buf_store(rs_o, pvo[nt], soff, acc[mt][nt][r]);
      }
    if constexpr (BNB)
      bnb_tail<CB>(tk, a.tail, a.stats, a.bn_mean, a.bn_c0, a.bn_c1, cout0, stats_row, cblk, tid,
                   reinterpret_cast<double*>(red), reinterpret_cast<int*>(red + 520));
    else
      bnf_tail<CB>(tk, a.ftail, a.stats, a.Cout, cout0, stats_row, cblk, tid, reinterpret_cast<double*>(red),
                   reinterpret_cast<int*>(red + 520));
  }
}

// NT = 32-pixel column tiles per wave: 2 (256-pixel workgroup tile) or 1 (128 pixels: twice the workgroups for the 16x16
// maps, whose 256-pixel tiles would leave one workgroup per CU with nothing to overlap its loader phases with)
// TERMS = bf16 terms per fp32 operand: 3 (the x3 arithmetic: six products, fp32 accuracy) or 1 (plain bf16 operands, one product,
// fp32 accumulation: the `bf16` mode of BASELINE.json configs[1] — wtpse_x3_terms(), include/wtpse_hip.h; NOT within the 1e-4
// parity bar and never used by the fp32 workloads).  Same tiles, loader and epilogues; the images hold TERMS planes.
template <int KS, int MT, int TWL, int EPI, int NT = 2, int TERMS = 3>
__global__ __launch_bounds__(256, 2) void conv_x3_k(ConvX3Args a) {
  static_assert(TERMS >= 1 && TERMS <= 3, "three bf16 terms (x3), two fp16 terms (x2h) or one bf16 term (bf16 mode)");
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  constexpr int TW = 1 << TWL, TH = (128 * NT) / TW;
  constexpr int PITCH = TW + 2 * PAD, ROWS = TH + 2 * PAD;
  constexpr int PE = PITCH * ROWS;
  constexpr int PEP = (PE + 7) & ~7;
  constexpr int CB = 32 * MT;
  constexpr int NACC = 16;
  constexpr int KC = 16;
  constexpr int XS_U4 = 2 * TERMS * PEP;           // 16-byte slots
  constexpr int WS_U4 = KS * 2 * TERMS * CB;       // one kernel row (KS taps)
  constexpr int NW = (WS_U4 + 255) / 256;          // 16-byte weight loads per thread and kernel row
  constexpr int RED_F = 4 * CB * 2;
  constexpr int MAIN_U4 = (XS_U4 + 2 * WS_U4) > (RED_F + 3) / 4 ? (XS_U4 + 2 * WS_U4) : (RED_F + 3) / 4;
  constexpr int PRO_MAX = MT == 1 ? 256 : 512;                     // input channels (virtual concat, padded) a prologue is staged for
  __shared__ u32x4v smem[MAIN_U4 + CB / 4 + (EPI == 2 ? CB : 0)];
  __shared__ float2 pro_s[PRO_MAX];                // (scale, shift) applied on load; (1, 0) without a prologue, (0, 0) padding
  u32x4v* Xs = smem;
  u32x4v* Ws = smem + XS_U4;
  float* bias_s = reinterpret_cast<float*>(smem + MAIN_U4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  int tile, cblk;
  x3_block_ids(a, tile, cblk);
  int bx = tile;
  const int tx = bx % a.tiles_x;
  bx /= a.tiles_x;
  const int ty = bx % a.tiles_y;
  const int b = bx / a.tiles_y;
  const int cout0 = cblk * CB;
  const int HW = a.H * a.W;
  if (tid < CB) bias_s[tid] = (a.bias && cout0 + tid < a.Cout) ? a.bias[cout0 + tid] : 0.f;
  if (EPI == 2 && tid < CB) {                     // (scale, shift, mean) of the BatchNorm'd output channels; (0, 1, 0) elsewhere
    const int c = cout0 + tid;
    const bool bn = c >= a.bn_c0 && c < a.bn_c1;
    float* q = bias_s + CB + tid;          // three planes [scale | shift | mean] of CB floats
    q[0] = (bn && a.bn_relu) ? a.bn_ss[2 * (c - a.bn_c0)] : 0.f;
    q[CB] = (bn && a.bn_relu) ? a.bn_ss[2 * (c - a.bn_c0) + 1] : 1.f;
    q[2 * CB] = bn ? a.bn_mean[c - a.bn_c0] : 0.f;
  }

  float sx = 1.f;                                  // power of two applied to the input on load (1 unless TERMS 2): set below

  int off[NT];                                    // halo position of this lane's pixel (tap 0,0 corner)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = wave * (32 * NT) + nt * 32 + r32;
    off[nt] = (p >> TWL) * PITCH + (p & (TW - 1));
  }
  // Loader work items: (halo position, k-half) = 8 channels of one position.  The two halves are laid out as
  // [half][positions padded to whole waves], dealt to the waves in blocks of 64, so that every thread gets the same number of
  // items (NIT) and a wave's half is uniform (scalar channel offsets and prologue coefficients).
  constexpr int PB = (PE + 63) / 64;               // 64-position blocks per half
  constexpr int NIT = (2 * PB + 3) / 4;            // items per thread
  int ipos[NIT], ihalf[NIT];
  unsigned voff[NIT];
  bool iin[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int blk = __builtin_amdgcn_readfirstlane(i * 4 + wave);
    ihalf[i] = blk >= PB ? 1 : 0;
    const int p = (blk - ihalf[i] * PB) * 64 + lane;
    ipos[i] = (blk < 2 * PB && p < PE) ? p : -1;
    const int r = p / PITCH, x = p - r * PITCH;
    const int gy = ty * TH + r - PAD, gx = tx * TW + x - PAD;
    iin[i] = ipos[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    voff[i] = iin[i] ? (unsigned)(gy * a.W + gx) * 4u + (unsigned)ihalf[i] * 8u * (unsigned)HW * 4u : BUF_OOB;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < NACC; ++r) acc[mt][nt][r] = 0.f;

  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0 + (size_t)b * a.C0 * HW, (unsigned)a.C0 * HW * 4u);
  const __amdgpu_buffer_rsrc_t rs1 = a.in1 ? make_rsrc(a.in1 + (size_t)b * a.C1 * HW, (unsigned)a.C1 * HW * 4u) : rs0;
  // packed weights: [chunk][cout block of 32][tap][term][half][32][8] bf16 = 16-byte slots [chunk][cb][tap][term*2+half][32]
  const int ncb32 = a.CoutP / 32;
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wx + X3_WHDR, (unsigned)(a.CinP / 16) * ncb32 * TAPS * 6u * 32u * 16u);
  // (per-lane offset + scalar offset: a lane is out of range when voffset >= num_records - soffset, see common.h)

  // Software pipeline over "rows" (one kernel row of one 16-channel chunk = KS taps = KS*MT*NT*6 MFMAs per wave): the global
  // loads of the next row's weights — and, on a chunk's last row, of the next chunk's input tile — are issued before the
  // row's MFMAs and land in registers behind them; the weights go to the other half of a double-buffered LDS slab right
  // after the MFMAs (one barrier per row), the input tile is split and stored once every wave has left the chunk.
  float xv[NIT][8];
  u32x4v wv[NW];
  auto issue_x = [&](int c0) {
    const bool first = c0 < a.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int cbase = first ? c0 : c0 - a.C0;
    const int cn = first ? a.C0 : a.C1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // channels past the end of the tensor are out of the buffer's range and read as zero (their packed weight rows are
      // zero too); min() keeps the scalar offset <= num_records so that the range check cannot wrap
      const unsigned soff = (unsigned)min(cbase + j, cn) * (unsigned)HW * 4u;
#pragma unroll
      for (int i = 0; i < NIT; ++i) xv[i][j] = buf_load(rs, voff[i], soff);
    }
  };
  // Conversion of a loaded chunk — prologue (affine, ReLU; the zero padding applies AFTER it, as in the reference graph), split
  // into three bf16 terms — one channel pair of one item at a time, so that for 3x3 kernels the pieces can sit between the MFMA
  // groups of the chunk's last kernel row (they used to run behind the chunk's barrier, ~1600 cycles per chunk during which
  // the wave issued no MFMA: 9 % of the forward kernel with a prologue, 6 % without).  Branch-free: the coefficients come from
  // LDS ((1, 0) without a prologue), the ReLU is a select on a uniform flag.
  u32x4v tq[NIT][TERMS];
  const bool any_pro = a.pro0 != nullptr || a.pro1 != nullptr || a.pro_relu != 0;     // (TERMS 2 without a prologue: the scale alone, below)
  auto convert_pair = [&](int c0, int i, int j, bool pro) __attribute__((always_inline)) {
    float v0 = xv[i][2 * j], v1 = xv[i][2 * j + 1];
    if (pro) {        // `true` between the MFMA groups (no branch there), any_pro behind a barrier
      // ReLU as ONE v_max against a wave-uniform floor (0 or -inf) instead of v_max + v_cndmask on a flag; without ReLU a NaN leaves
      // the max as -inf, whose split remainder (-inf + inf) is NaN again: the outputs it feeds are NaN either way
      const float lo = ((c0 < a.C0) ? (a.pro_relu & 1) : (a.pro_relu & 2)) ? 0.f : -INFINITY;
      const int cg = min(c0 + ihalf[i] * 8 + 2 * j, PRO_MAX - 2);
      const float2 p0 = pro_s[cg], p1 = pro_s[cg + 1];
      v0 = fmaxf(fmaf(v0, p0.x, p0.y), lo);
      v1 = fmaxf(fmaf(v1, p1.x, p1.y), lo);
      v0 = iin[i] ? v0 : 0.f;
      v1 = iin[i] ? v1 : 0.f;
    } else if (TERMS == 2) {      // data gradients and block inputs: no prologue, out-of-range loads are zeros already
      v0 *= sx;
      v1 *= sx;
    }
    unsigned q[TERMS];
    x3_split_pair<TERMS>(v0, v1, q);
#pragma unroll
    for (int t = 0; t < TERMS; ++t) tq[i][t][j] = q[t];
  };
  auto store_x = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      if (ipos[i] >= 0) {
#pragma unroll
        for (int t = 0; t < TERMS; ++t) Xs[(t * 2 + ihalf[i]) * PEP + ipos[i]] = tq[i][t];
      }
    }
  };
  auto stash_x = [&](int c0) __attribute__((always_inline)) {
    // (compile-time indices: as two `#pragma unroll` loops the 3x3 MT 1 variant indexed xv / tq dynamically, through scratch)
    // (one uniform branch around the whole conversion, not one per pair)
    if (any_pro) x3_static_for<NIT * 4>([&](auto pc) __attribute__((always_inline)) { convert_pair(c0, decltype(pc)::value >> 2, decltype(pc)::value & 3, true); });
    else x3_static_for<NIT * 4>([&](auto pc) __attribute__((always_inline)) { convert_pair(c0, decltype(pc)::value >> 2, decltype(pc)::value & 3, false); });
    store_x();
  };
  // one kernel row of weights: LDS slot s = ((tl * 2 TERMS + q) * CB + co), tl = tap within the row, q = term*2 + half (the packed
  // weights always carry three terms: TERMS = 1 fetches the leading one only)
  unsigned wslot[NW];
#pragma unroll
  for (int it = 0; it < NW; ++it) {
    const int s = tid + 256 * it;
    const int co = s % CB, q6 = (s / CB) % (2 * TERMS), tl = s / (CB * 2 * TERMS);
    const bool ok = s < WS_U4 && cout0 + co < a.CoutP;
    wslot[it] = ok ? (unsigned)(((co >> 5) * (TAPS * 6 * 32) + (tl * 6 + q6) * 32 + (co & 31)) * 16) : BUF_OOB;
  }
  auto issue_w = [&](int chunk, int ky) {
    const unsigned base = ((unsigned)(chunk * ncb32 + cout0 / 32) * (unsigned)(TAPS * 6 * 32) + (unsigned)(ky * KS * 6 * 32)) * 16u;
#pragma unroll
    for (int it = 0; it < NW; ++it) wv[it] = __builtin_bit_cast(u32x4v, buf_load4(rsw, wslot[it], base));
  };
  auto stash_w = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NW; ++it)
      if (NW * 256 == WS_U4 || tid + 256 * it < WS_U4) Ws[buf * WS_U4 + tid + 256 * it] = wv[it];
  };

  const int nchunks = a.CinP / KC;
  // prologue coefficients of this thread's channels: fetched in front of the tile loads (their own latency), scaled and staged below
  float2 praw[PRO_MAX / 256];
#pragma unroll
  for (int q = 0; q < PRO_MAX / 256; ++q) {
    const int c = tid + 256 * q;
    const bool first = c < a.C0;
    const float* pro = first ? a.pro0 : a.pro1;
    const int cl = first ? c : c - a.C0;
    const bool live = c < a.C0 + a.C1;
    praw[q] = !live ? make_float2(0.f, 0.f) : (pro ? make_float2(pro[2 * cl], pro[2 * cl + 1]) : make_float2(1.f, 0.f));
  }
  // the scale is a dependent read of the amax tables (a gradient's, or since round 6 a forward activation's bound): issued in front
  // of the first tile's loads, picked up behind them — its round trip sits neither in front of every workgroup's first load (round 5)
  // nor behind its last one
  const unsigned sx_raw = x3_in_scale_issue<TERMS>(a);
  __builtin_amdgcn_sched_barrier(0);               // (the read stays in FRONT of the tile loads)
  issue_x(0);
  issue_w(0, 0);
  sx = x3_in_scale<TERMS>(a, sx_raw);
#pragma unroll
  for (int q = 0; q < PRO_MAX / 256; ++q)
    if (tid + 256 * q < a.CinP) pro_s[tid + 256 * q] = make_float2(praw[q].x * sx, praw[q].y * sx);
  __syncthreads();                    // pro_s
  stash_x(0);
  stash_w(0);
  __syncthreads();
  int buf = 0;
  // 3x3: the next chunk's tile is loaded at the start of the middle kernel row (in flight behind that row's MFMAs) and converted
  // piecewise between the MFMA groups of the last row; after the chunk's barrier only the LDS stores remain.  1x1 (one row
  // per chunk): loaded in front of the row, converted behind the barrier.  The loads and the conversion also run on the last
  // chunk (out of range: zeros) — no branch inside the MFMA stream.
  constexpr bool PIPE = KS == 3 && MT == 2 && TERMS == 3;     // MT 1: the 36 extra registers cost the third wave per SIMD (measured 210 -> 248 us)
  constexpr int NPIECE = NIT * 4;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const bool last_row = ky == KS - 1;
      const bool more = !last_row || chunk + 1 < nchunks;
      if (more) issue_w(last_row ? chunk + 1 : chunk, last_row ? 0 : ky + 1);
      if (PIPE ? ky == KS - 2 : (last_row && more)) issue_x((chunk + 1) * KC);
      const int c0n = (chunk + 1) * KC;
      int piece = 0;
      auto convert_piece = [&]() __attribute__((always_inline)) {      // one channel pair behind each of the first NPIECE MFMA groups of the last row
        if constexpr (PIPE) if (last_row && piece < NPIECE) {
          convert_pair(c0n, piece >> 2, piece & 3, true);
          // 4 MFMAs (128 cycles of the pipe) : ~20 VALU — one MFMA, then a fifth of the piece
#pragma unroll
          for (int q = 0; q < MT * NT; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (24 + MT * NT - 1) / (MT * NT), 0);
          }
        }
        ++piece;
      };
      // keep the loads in front of the MFMAs (left alone, the scheduler sinks them to their first use behind the row,
      // where their latency is exposed)
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMAs of this kernel row.  The six cross terms of a tap run as six groups of MT*NT independent MFMAs (one per
      // accumulator), smallest terms first; a fragment is re-read for tap tl+1 right behind the group that used it last, so
      // every ds_read has MFMA groups to land behind (left to the scheduler, a tap's 12 reads sit in front of its MFMAs
      // with their latency exposed three times per row).  The weight terms 0 and 1, needed by the first two groups of the
      // next tap and used until the last two of this one, alternate between two register sets.
      const u32x4v* Wb = Ws + buf * WS_U4;
      {
        u32x4v a01[2][MT][2], a2[MT], bfr[NT][3];
        auto rd_a = [&](int tl, int t) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const u32x4v v = Wb[((tl * 2 * TERMS) + t * 2 + h) * CB + mt * 32 + r32];
            if (t == 2) a2[mt] = v; else a01[tl & 1][mt][t] = v;
          }
        };
        auto rd_b = [&](int tl, int t) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bfr[nt][t] = Xs[(t * 2 + h) * PEP + off[nt] + ky * PITCH + tl];
        };
        auto mm = [&](int tl, int ta, int tb) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = x3_mfma<TERMS>(ta == 2 ? a2[mt] : a01[tl & 1][mt][ta], bfr[nt][tb], acc[mt][nt]);
        };
        if constexpr (TERMS == 1) {      // bf16 mode: one product per tap
          rd_a(0, 0); rd_b(0, 0);
#pragma unroll
          for (int tl = 0; tl < KS; ++tl) {
            mm(tl, 0, 0);
            if (tl + 1 < KS) { rd_a(tl + 1, 0); rd_b(tl + 1, 0); }
          }
        } else if constexpr (TERMS == 2) {      // x2h: three products per tap, smallest first; the next tap's fragments behind the first group
          rd_a(0, 0); rd_b(0, 1); rd_a(0, 1); rd_b(0, 0);
#pragma unroll
          for (int tl = 0; tl < KS; ++tl) {
            const bool nx = tl + 1 < KS;
            __builtin_amdgcn_sched_barrier(0);
            mm(tl, 0, 1);
            if (nx) { rd_b(tl + 1, 1); rd_a(tl + 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            mm(tl, 1, 0);
            if (nx) rd_a(tl + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(tl, 0, 0);
            if (nx) rd_b(tl + 1, 0);
          }
        } else {
        rd_a(0, 0); rd_b(0, 2); rd_a(0, 1); rd_b(0, 1); rd_a(0, 2); rd_b(0, 0);
#pragma unroll
        for (int tl = 0; tl < KS; ++tl) {
          const bool nx = tl + 1 < KS;
          __builtin_amdgcn_sched_barrier(0);
          mm(tl, 0, 2);
          if (nx) { rd_b(tl + 1, 2); rd_a(tl + 1, 0); }
          convert_piece();
          __builtin_amdgcn_sched_barrier(0);
          mm(tl, 1, 1);
          if (nx) rd_a(tl + 1, 1);
          convert_piece();
          __builtin_amdgcn_sched_barrier(0);
          mm(tl, 2, 0);
          if (nx) rd_a(tl + 1, 2);
          convert_piece();
          __builtin_amdgcn_sched_barrier(0);
          mm(tl, 0, 1);
          if (nx) rd_b(tl + 1, 1);
          convert_piece();
          __builtin_amdgcn_sched_barrier(0);
          mm(tl, 1, 0);
          convert_piece();
          __builtin_amdgcn_sched_barrier(0);
          mm(tl, 0, 0);
          if (nx) rd_b(tl + 1, 0);
          convert_piece();
        }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) stash_w(buf ^ 1);       // the other half: its readers passed the barrier at the end of the previous row
      if (last_row && more) {
        __syncthreads();                // every wave is done with this chunk's input tile
        if (PIPE) store_x(); else stash_x((chunk + 1) * KC);
      }
      __syncthreads();
      buf ^= 1;
    }
  }
  x3_unscale<TERMS, MT, NT>(a, acc, sx);
  x3_epilogue<MT, NT, TWL, EPI, true>(a, acc, b, ty, tx, cout0, tid, reinterpret_cast<float*>(smem), bias_s, tile, cblk, true);
}

// ------------------------------------------------------------------------------------------------
// conv_x3r_k (round 4): the 3x3 forward / data gradient with REGISTER-FED weights.  Same arithmetic, same tiles, same loader and
// epilogues as conv_x3_k — every accumulator sees the same products in the same order, so the results are bitwise those of
// conv_x3_k (tests/test_conv_x3_gpu.py::test_x3r_equals_x3) — but a different operand supply:
//   * the packed weights ARE in fragment order already ([chunk][32-row block][tap][term][k-half][row][8 k]: lane (h, r32) of a
//     32x32x16 A fragment owns 16 consecutive bytes, a wave 1 KB): every wave loads its A fragments straight from global memory
//     (L2 / L1 resident: 55 KB per 16-channel chunk and 64 output channels) into a ring of three tap slots, two taps ahead of their
//     use.  No weight slab in LDS: 55 of the 88 KB a workgroup stored per chunk, and the two weight-row barriers per chunk, are gone;
//   * the input tile keeps going through LDS (the halo is shared by the workgroup's waves), now DOUBLE-buffered: the next chunk is
//     converted and stored piece by piece between the MFMA groups of the current one, one barrier per chunk (216 MFMAs per wave);
//   * WM = 2 (64 output channels per workgroup): the waves form a 2 x 2 grid — 32 channels x 128 pixels each — instead of 1 x 4
//     (64 channels x 64 pixels): half the weight bytes per MFMA through the vector-memory path (16 B/clk/CU instead of 31), twice the
//     B-fragment reads from LDS (64 of 256 B/clk/CU), which is the cheaper of the two;
//   * MFMAs run pixel-tile-major (six dependent products per accumulator back to back: a chain of v_mfma_f32_32x32x16_bf16 issues
//     at the full rate, MI355X_MICROARCH.md), so only two B fragment sets (current, next) are live.

// TERMS: 3 = three bf16 terms per operand, six products (x3); 1 = one bf16 term, one product (bf16 mode); 2 = two fp16 terms, three products ("x2h": split2h_pair above).
// PLAIN: the launch has no prologue (data gradients, a block's first convolution): the conversion is scale + split only.
template <int WM, int MT, int NT, int TWL, int EPI, int TERMS = 3, bool PLAIN = false>
__global__ __launch_bounds__(256, 2) void conv_x3r_k(ConvX3Args a) {
  static_assert(TERMS >= 1 && TERMS <= 3, "three bf16 terms (x3), two fp16 terms (x2h) or one bf16 term (bf16 mode)");
  constexpr int KS = 3, TAPS = 9, PAD = 1;
  constexpr int PW = 4 / WM;                       // waves along the pixels
  constexpr int TW = 1 << TWL, TH = (PW * 32 * NT) / TW;
  constexpr int PITCH = TW + 2 * PAD, ROWS = TH + 2 * PAD;
  constexpr int PE = PITCH * ROWS;
  constexpr int PEP = (PE + 7) & ~7;
  constexpr int CBW = 32 * MT, CB = CBW * WM;
  constexpr int NACC = 16;
  constexpr int KC = 16;
  constexpr int XS_U4 = 2 * TERMS * PEP;           // 16-byte slots of one input image
  constexpr int MAIN_U4 = 2 * XS_U4;
  constexpr int PRO_MAX = 512;
  static_assert(MAIN_U4 * 4 >= PW * CB * 2 + 4 * CB + 4 && MAIN_U4 * 4 >= 524, "epilogue scratch (statistics, the tails' 256 doubles + flag) aliases the operand images");
  constexpr int DUMP_U4 = MAIN_U4 + CB / 4 + (EPI == 2 ? CB : 0);
  __shared__ u32x4v smem[DUMP_U4 + 64];
  __shared__ float2 pro_s[PRO_MAX];
  u32x4v* Xs = smem;
  float* bias_s = reinterpret_cast<float*>(smem + MAIN_U4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cw = __builtin_amdgcn_readfirstlane(wave % WM), pw = __builtin_amdgcn_readfirstlane(wave / WM);
  const int r32 = lane & 31, h = lane >> 5;
  int tile, cblk;
  x3_block_ids(a, tile, cblk);
  int bx = tile;
  const int tx = bx % a.tiles_x;
  bx /= a.tiles_x;
  const int ty = bx % a.tiles_y;
  const int b = bx / a.tiles_y;
  const int cout0 = cblk * CB;
  const int HW = a.H * a.W;
  float sx = 1.f;                                  // power of two applied to the input on load (1 for TERMS 3): set behind the first loads
  if (tid < CB) bias_s[tid] = (a.bias && cout0 + tid < a.Cout) ? a.bias[cout0 + tid] : 0.f;
  if (EPI == 2 && tid < CB) {
    const int c = cout0 + tid;
    const bool bn = c >= a.bn_c0 && c < a.bn_c1;
    float* q = bias_s + CB + tid;
    q[0] = (bn && a.bn_relu) ? a.bn_ss[2 * (c - a.bn_c0)] : 0.f;
    q[CB] = (bn && a.bn_relu) ? a.bn_ss[2 * (c - a.bn_c0) + 1] : 1.f;
    q[2 * CB] = bn ? a.bn_mean[c - a.bn_c0] : 0.f;
  }

  int off[NT];                                    // halo position of this lane's pixel (tap 0,0 corner)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = pw * (32 * NT) + nt * 32 + r32;
    off[nt] = (p >> TWL) * PITCH + (p & (TW - 1));
  }
  // loader work items exactly as in conv_x3_k: (halo position, k-half) = 8 channels of one position, dealt in whole-wave blocks
  constexpr int PB = (PE + 63) / 64;
  constexpr int NIT = (2 * PB + 3) / 4;
  int ipos[NIT], ihalf[NIT];
  unsigned voff[NIT];
  bool iin[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int blk = __builtin_amdgcn_readfirstlane(i * 4 + wave);
    ihalf[i] = blk >= PB ? 1 : 0;
    const int p = (blk - ihalf[i] * PB) * 64 + lane;
    ipos[i] = (blk < 2 * PB && p < PE) ? p : -1;
    const int r = p / PITCH, x = p - r * PITCH;
    const int gy = ty * TH + r - PAD, gx = tx * TW + x - PAD;
    iin[i] = ipos[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    voff[i] = iin[i] ? (unsigned)(gy * a.W + gx) * 4u + (unsigned)ihalf[i] * 8u * (unsigned)HW * 4u : BUF_OOB;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < NACC; ++r) acc[mt][nt][r] = 0.f;

  const int ncb32 = a.CoutP / 32;
  const int nchunks = a.CinP / KC;
  constexpr unsigned TAP_B = 6u * 32u * 16u;       // bytes of one (chunk, 32-row block, tap): three term slots x two k-halves x 32 rows x 16 B
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wx + X3_WHDR, (unsigned)nchunks * ncb32 * TAPS * TAP_B);
  const unsigned wlane = (unsigned)lane * 16u;     // (k-half h, row r32) = slot h * 32 + r32 = lane
  const int cb32 = cout0 / 32 + cw * MT;

  float xv[NIT][8];
  const float* const xb0 = a.in0 + (size_t)b * a.C0 * HW;
  const float* const xb1 = a.in1 ? a.in1 + (size_t)b * a.C1 * HW : xb0;
  auto issue_x = [&](int c0) __attribute__((always_inline)) {
    const bool first = c0 < a.C0;
    // (a select between two ready-made descriptors came out as a VECTOR value here — every load then sat in a readfirstlane
    // "waterfall" loop; the descriptor is built from a pointer and a size that are pinned to scalar registers instead)
    const unsigned long long pb = (unsigned long long)(first ? xb0 : xb1);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pb), phi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
    const int cbase = __builtin_amdgcn_readfirstlane(first ? c0 : c0 - a.C0);
    const int cn = __builtin_amdgcn_readfirstlane(first ? a.C0 : a.C1);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc((const void*)(((unsigned long long)phi << 32) | plo), (unsigned)cn * HW * 4u);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned soff = (unsigned)min(cbase + j, cn) * (unsigned)HW * 4u;     // past the tensor: out of range, reads 0
#pragma unroll
      for (int i = 0; i < NIT; ++i) xv[i][j] = buf_load(rs, voff[i], soff);
    }
  };
  // one item = 8 channels of one halo position: prologue (affine, ReLU; zero padding AFTER it), split, TERMS 16-byte LDS stores
  u32x4v tq[TERMS];
  auto convert_pair = [&](int c0, int i, int j) __attribute__((always_inline)) {
    float v0 = xv[i][2 * j], v1 = xv[i][2 * j + 1];
    if constexpr (PLAIN) {        // out-of-range loads (padding, channels past the tensor) are zeros already
      if constexpr (TERMS == 2) {
        v0 *= sx;
        v1 *= sx;
      }
    } else {
      // ReLU as ONE v_max against a wave-uniform floor (0 or -inf) instead of v_max + v_cndmask on a flag; without ReLU a NaN leaves
      // the max as -inf, whose split remainder (-inf + inf) is NaN again: the outputs it feeds are NaN either way
      const float lo = ((c0 < a.C0) ? (a.pro_relu & 1) : (a.pro_relu & 2)) ? 0.f : -INFINITY;
      const int cg = min(c0 + ihalf[i] * 8 + 2 * j, PRO_MAX - 2);
      const float2 p0 = pro_s[cg], p1 = pro_s[cg + 1];
      v0 = fmaxf(fmaf(v0, p0.x, p0.y), lo);
      v1 = fmaxf(fmaf(v1, p1.x, p1.y), lo);
      v0 = iin[i] ? v0 : 0.f;
      v1 = iin[i] ? v1 : 0.f;
    }
    unsigned q[TERMS];
    x3_split_pair<TERMS>(v0, v1, q);
#pragma unroll
    for (int t = 0; t < TERMS; ++t) tq[t][j] = q[t];
  };
  // branch-free (the stores sit between MFMA groups): lanes without a position (the tail of a half's last 64-block) store to a
  // per-lane dump slot behind the images
  auto store_item = [&](int i, int xb) __attribute__((always_inline)) {
    const bool ok = ipos[i] >= 0;
    const int base = xb * XS_U4 + ihalf[i] * PEP + ipos[i];
#pragma unroll
    for (int t = 0; t < TERMS; ++t) Xs[ok ? base + t * 2 * PEP : DUMP_U4 + lane] = tq[t];
  };
  // piece p of a chunk's conversion: pair (p & 3) of item (p >> 2); an item's fourth pair is followed by its stores
  auto convert_piece = [&](int c0, int p, int xb) __attribute__((always_inline)) {
    convert_pair(c0, p >> 2, p & 3);
    if ((p & 3) == 3) store_item(p >> 2, xb);
  };

  // A fragments: a ring of three tap slots, loaded two taps ahead of their use.  (A ring of nine — a whole chunk ahead — shrank the
  // waits on these loads from 15 % to 5 % of the launch and left the launch time unchanged, profiles/r04_x3r_ablation.txt.)
  constexpr int RING = 3;
  u32x4v afr[RING][MT][TERMS];
  auto issue_a = [&](int chunk, int tap, int slot) __attribute__((always_inline)) {
    const unsigned base = ((unsigned)(chunk * ncb32 + cb32) * TAPS + (unsigned)tap) * TAP_B;
#pragma unroll
    for (int t = 0; t < TERMS; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        afr[slot][mt][t] = __builtin_bit_cast(u32x4v, buf_load4(rsw, wlane, base + (unsigned)mt * TAPS * TAP_B + (unsigned)t * 2u * 32u * 16u));
  };
  u32x4v bfr[2][TERMS];
  auto rd_b = [&](int xb, int tap, int nt, int set) __attribute__((always_inline)) {
    const u32x4v* X = Xs + xb * XS_U4 + off[nt] + (tap / 3) * PITCH + (tap % 3);
#pragma unroll
    for (int t = 0; t < TERMS; ++t) bfr[set][t] = X[(t * 2 + h) * PEP];
  };
  auto mm = [&](int slot, int nt, int set) __attribute__((always_inline)) {
    // the cross terms, smallest first — per accumulator the order of conv_x3_k
    constexpr int NP = TERMS == 3 ? 6 : TERMS == 2 ? 3 : 1;
    constexpr int TA3[6] = {0, 1, 2, 0, 1, 0}, TB3[6] = {2, 1, 0, 1, 0, 0};
    constexpr int TA2[3] = {0, 1, 0}, TB2[3] = {1, 0, 0};
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int ta = TERMS == 3 ? TA3[q] : TERMS == 2 ? TA2[q % 3] : 0, tb = TERMS == 3 ? TB3[q] : TERMS == 2 ? TB2[q % 3] : 0;
        acc[mt][nt] = x3_mfma<TERMS>(afr[slot][mt][ta], bfr[set][tb], acc[mt][nt]);
      }
  };

  constexpr int NPIECE = NIT * 4;
  constexpr int NGRP = TAPS * NT;                  // MFMA groups (tap, column tile) per chunk
  static_assert(NPIECE <= NGRP, "one conversion piece per MFMA group");
  constexpr int G0 = NGRP - NPIECE;                // the pieces ride behind the last NPIECE groups

  if constexpr (PLAIN) {
    // no prologue (a data gradient, a block's first convolution): the conversion needs the scale only — read as in round 5, in front of
    // everything (the split read below costs the BatchNorm-backward variant 11 registers and with them its third wave per SIMD)
    sx = x3_in_scale<TERMS>(a, x3_in_scale_issue<TERMS>(a));
    issue_x(0);
    issue_a(0, 0, 0);
    issue_a(0, 1, 1);
  } else {
    float2 praw[PRO_MAX / 256];                    // this thread's prologue coefficients: fetched in front of the tile loads
#pragma unroll
    for (int q = 0; q < PRO_MAX / 256; ++q) {
      const int c = tid + 256 * q;
      const bool first = c < a.C0;
      const float* pro = first ? a.pro0 : a.pro1;
      const int cl = first ? c : c - a.C0;
      const bool live = c < a.C0 + a.C1;
      praw[q] = !live ? make_float2(0.f, 0.f) : (pro ? make_float2(pro[2 * cl], pro[2 * cl + 1]) : make_float2(1.f, 0.f));
    }
    const unsigned sx_raw = x3_in_scale_issue<TERMS>(a);      // (the amax tables: read issued in front of the first loads, see conv_x3_k)
    __builtin_amdgcn_sched_barrier(0);
    issue_x(0);
    issue_a(0, 0, 0);
    issue_a(0, 1, 1);
    sx = x3_in_scale<TERMS>(a, sx_raw);
    // (power-of-two scaling commutes with the rounding of the fused multiply-add and with the ReLU: relu(fma(y, s sc, s sh)) = s relu(fma(y, sc, sh)))
#pragma unroll
    for (int q = 0; q < PRO_MAX / 256; ++q)
      if (tid + 256 * q < a.CinP) pro_s[tid + 256 * q] = make_float2(praw[q].x * sx, praw[q].y * sx);
  }
  __syncthreads();                                 // pro_s
  x3_static_for<NPIECE>([&](auto pc) __attribute__((always_inline)) { convert_piece(0, decltype(pc)::value, 0); });
  issue_x(KC);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int xb = chunk & 1;
    const int chn = min(chunk + 1, nchunks - 1);   // fragments prefetched past the last chunk are never used
    const int c0n = (chunk + 1) * KC;
    __builtin_amdgcn_sched_barrier(0);
    rd_b(xb, 0, 0, 0);
    x3_static_for<NGRP>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      constexpr int tap = g / NT, nt = g % NT;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (nt == 0) {
        if constexpr (tap + 2 < TAPS) issue_a(chunk, tap + 2, (tap + 2) % 3);
        else issue_a(chn, tap + 2 - TAPS, (tap + 2) % 3);
      }
      if constexpr (g + 1 < NGRP) rd_b(xb, (g + 1) / NT, (g + 1) % NT, (g + 1) & 1);
      // the next group's fragment reads go out in FRONT of this group's MFMAs (left alone, the scheduler re-uses the registers of
      // the current fragments for them and sinks the reads behind the fourth MFMA: 64 cycles in front of their s_waitcnt)
      __builtin_amdgcn_sched_barrier(0);
      mm(tap % RING, nt, g & 1);
      if constexpr (g >= G0) convert_piece(c0n, g - G0, xb ^ 1);
    });
    __builtin_amdgcn_sched_barrier(0);
    issue_x((chunk + 2) * KC);
    __syncthreads();
  }
  x3_unscale<TERMS, MT, NT>(a, acc, sx);
  x3_epilogue<MT, NT, TWL, EPI, true, WM>(a, acc, b, ty, tx, cout0, tid, reinterpret_cast<float*>(smem), bias_s, tile, cblk, true);
}


// ------------------------------------------------------------------------------------------------
// Launch side of one TERMS value (instantiated in conv_x3_t<TERMS>.hip).
struct X3Launch {
  int ksize;        // 1 | 3
  int mt2;          // 64-channel blocks (x3_mt2, conv_x3.hip)
  int half;         // 64-channel blocks on 128-pixel tiles (x3_half)
  int small;        // 32-channel blocks on 128-pixel tiles (x3_small_tiles)
  int epi;          // 0 plain, 1 ReLU mask, 2 BatchNorm-backward statistics
  int x3r;          // wtpse_x3r_enable(): 0 conv_x3_k everywhere, 1 conv_x3r_k for the 64-channel blocks of 3x3 layers, 2 for every 3x3 layer
  int xcd;          // XCD-aware workgroup order
};

template <int KS, int MT, int EPI, int TERMS>
static int launch_x3(const ConvX3Args& a, const X3Launch& L, hipStream_t st) {
  ConvX3Args args = a;
  const bool narrow = a.W <= 16;
  const bool half = L.half != 0, small = MT == 1 && L.small != 0;
  const int TW = narrow ? 16 : 32, TH = ((small || half) ? 128 : 256) / TW;
  args.tiles_x = ceil_div(a.W, TW);
  args.tiles_y = ceil_div(a.H, TH);
  dim3 grid((unsigned)(a.B * args.tiles_x * args.tiles_y), (unsigned)ceil_div(a.CoutP, 32 * MT));
  args.xcd_tiles = (L.xcd && grid.x % 8 == 0 && (long long)grid.x * grid.y >= 64) ? (int)grid.x / 8 : 0;
  const bool in_launch = tail_in_launch((long long)grid.x * grid.y);     // else: the stand-alone finalize kernel behind the launch
  if (!in_launch) args.tail.tickets = args.ftail.tickets = nullptr;
  if (args.tail.tickets) bnb_tail_geometry(args.tail, (int)grid.x, a.Cout, (double)a.B * a.H * a.W);
  if (args.ftail.tickets) bnf_tail_geometry(args.ftail, (int)grid.x, a.Cout, (double)a.B * a.H * a.W);
  // x3r: 1 = conv_x3r_k where it measured at least as fast (64-channel blocks: +1..10 % on the forward launches, +-1 % on the data
  // gradients), 2 = everywhere (32-channel blocks run 8-20 % SLOWER on it: half the MFMAs per converted input element), 0 = nowhere
  if (KS == 3 && (L.x3r == 2 || (L.x3r == 1 && MT == 2))) {
    if constexpr (KS == 3) {
      const bool plain = !a.pro0 && !a.pro1 && !a.pro_relu;
#define X3R_GO(WM_, NT_, TWL_)                                                                                          \
  do {                                                                                                                  \
    if (plain) hipLaunchKernelGGL((conv_x3r_k<WM_, 1, NT_, TWL_, EPI, TERMS, true>), grid, dim3(256), 0, st, args);     \
    else hipLaunchKernelGGL((conv_x3r_k<WM_, 1, NT_, TWL_, EPI, TERMS, false>), grid, dim3(256), 0, st, args);          \
  } while (0)
      if (half) {
        if constexpr (MT == 2) {
          if (narrow) X3R_GO(2, 2, 4);
          else X3R_GO(2, 2, 5);
        }
      } else if (small) {
        if constexpr (MT == 1) {
          if (narrow) X3R_GO(1, 1, 4);
          else X3R_GO(1, 1, 5);
        }
      } else if (MT == 2) {
        if (narrow) X3R_GO(2, 4, 4);
        else X3R_GO(2, 4, 5);
      } else {
        if (narrow) X3R_GO(1, 2, 4);
        else X3R_GO(1, 2, 5);
      }
#undef X3R_GO
    }
  } else if (small) {
    if constexpr (MT == 1) {
      if (narrow) hipLaunchKernelGGL((conv_x3_k<KS, 1, 4, EPI, 1, TERMS>), grid, dim3(256), 0, st, args);
      else hipLaunchKernelGGL((conv_x3_k<KS, 1, 5, EPI, 1, TERMS>), grid, dim3(256), 0, st, args);
    }
  } else if (narrow)
    hipLaunchKernelGGL((conv_x3_k<KS, MT, 4, EPI, 2, TERMS>), grid, dim3(256), 0, st, args);
  else
    hipLaunchKernelGGL((conv_x3_k<KS, MT, 5, EPI, 2, TERMS>), grid, dim3(256), 0, st, args);
  int rc = wtpse_status();
  if (rc == 0 && !in_launch)
    rc = tail_after_launch(a.tail, a.ftail, a.stats, (int)grid.x, a.Cout, a.bn_c0, a.bn_c1, a.bn_mean, (long long)a.B * a.H * a.W, st);
  return rc;
}

template <int TERMS>
static int x3_dispatch(const ConvX3Args& a, const X3Launch& L, hipStream_t st) {
#define X3E(KS, M) (L.epi == 2 ? launch_x3<KS, M, 2, TERMS>(a, L, st) : L.epi == 1 ? launch_x3<KS, M, 1, TERMS>(a, L, st) : launch_x3<KS, M, 0, TERMS>(a, L, st))
  if (L.ksize == 3) return (L.mt2 || L.half) ? X3E(3, 2) : X3E(3, 1);
  return L.mt2 ? X3E(1, 2) : X3E(1, 1);
#undef X3E
}
