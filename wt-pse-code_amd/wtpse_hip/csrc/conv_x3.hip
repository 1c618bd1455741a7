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

// Phase stamps for tools/probe/x3_stamps.py (never compiled into libwtpse_hip.so): thread 0 of every workgroup records
// s_memtime at the phase boundaries of conv_x3_k.
#ifdef WTPSE_STAMPS
__device__ unsigned long long* g_stamps_x3 = nullptr;
extern "C" int wtpse_probe_set_stamps_x3(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_x3), &p, sizeof(p)); }
#define XSTAMP(i) do { if (g_stamps_x3 && threadIdx.x == 0) g_stamps_x3[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define XSTAMPV(i, v) do { if (g_stamps_x3 && threadIdx.x == 0) g_stamps_x3[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + (i)] = (v); } while (0)
#else
#define XSTAMP(i)
#define XSTAMPV(i, v)
#endif

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
  const float relu_lo = a.relu_out ? 0.f : -INFINITY;
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
        float v = fmaxf(acc[mt][nt][r], relu_lo);
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
            const float v = fmaxf(acc[mt][nt][r], relu_lo);
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
        for (int nt = 0; nt < NT; ++nt) buf_store(rs_o, pvo[nt], soff, fmaxf(acc[mt][nt][r], relu_lo));
      }
    if constexpr (BNB)
      bnb_tail<CB>(tk, a.tail, a.stats, a.bn_mean, a.bn_c0, a.bn_c1, cout0, stats_row, cblk, tid,
                   reinterpret_cast<double*>(red), reinterpret_cast<int*>(red + 4 * CB));
    else
      bnf_tail<CB>(tk, a.ftail, a.stats, a.Cout, cout0, stats_row, cblk, tid, reinterpret_cast<double*>(red),
                   reinterpret_cast<int*>(red + 4 * CB));
  }
}

// NT = 32-pixel column tiles per wave: 2 (256-pixel workgroup tile) or 1 (128 pixels: twice the workgroups for the 16x16
// maps, whose 256-pixel tiles would leave one workgroup per CU with nothing to overlap its loader phases with)
// TERMS = bf16 terms per fp32 operand: 3 (the x3 arithmetic: six products, fp32 accuracy) or 1 (plain bf16 operands, one product,
// fp32 accumulation: the `bf16` mode of BASELINE.json configs[1] — wtpse_x3_terms(), include/wtpse_hip.h; NOT within the 1e-4
// parity bar and never used by the fp32 workloads).  Same tiles, loader and epilogues; the images hold TERMS planes.
template <int KS, int MT, int TWL, int EPI, int NT = 2, int TERMS = 3>
__global__ __launch_bounds__(256, 2) void conv_x3_k(ConvX3Args a) {
  static_assert(TERMS == 3 || TERMS == 1, "three bf16 terms (fp32 accuracy) or one (bf16 mode)");
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

  for (int c = tid; c < a.CinP; c += 256) {
    const bool first = c < a.C0;
    const float* pro = first ? a.pro0 : a.pro1;
    const int cl = first ? c : c - a.C0;
    const bool live = c < a.C0 + a.C1;
    pro_s[c] = !live ? make_float2(0.f, 0.f) : (pro ? make_float2(pro[2 * cl], pro[2 * cl + 1]) : make_float2(1.f, 0.f));
  }

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
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wx, (unsigned)(a.CinP / 16) * ncb32 * TAPS * 6u * 32u * 16u);
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
  const bool any_pro = a.pro0 != nullptr || a.pro1 != nullptr || a.pro_relu != 0;
  auto convert_pair = [&](int c0, int i, int j, bool pro) __attribute__((always_inline)) {
    float v0 = xv[i][2 * j], v1 = xv[i][2 * j + 1];
    if (pro) {        // `true` between the MFMA groups (no branch there), any_pro behind a barrier
      const bool relu = (c0 < a.C0) ? (a.pro_relu & 1) : (a.pro_relu & 2);
      const int cg = min(c0 + ihalf[i] * 8 + 2 * j, PRO_MAX - 2);
      const float2 p0 = pro_s[cg], p1 = pro_s[cg + 1];
      v0 = fmaf(v0, p0.x, p0.y);
      v1 = fmaf(v1, p1.x, p1.y);
      v0 = relu ? fmaxf(v0, 0.f) : v0;
      v1 = relu ? fmaxf(v1, 0.f) : v1;
      v0 = iin[i] ? v0 : 0.f;
      v1 = iin[i] ? v1 : 0.f;
    }
    if constexpr (TERMS == 3) {
      unsigned q0, q1, q2;
      split3_pair(v0, v1, q0, q1, q2);
      tq[i][0][j] = q0;
      tq[i][1][j] = q1;
      tq[i][2][j] = q2;
    } else {
      tq[i][0][j] = pack_rne(v0, v1);
    }
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
    x3_static_for<NIT * 4>([&](auto pc) __attribute__((always_inline)) { convert_pair(c0, decltype(pc)::value >> 2, decltype(pc)::value & 3, any_pro); });
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
  XSTAMP(0);
  XSTAMPV(58, __builtin_amdgcn_s_memrealtime());   // 100 MHz: with slots 0/61 gives the clock the workgroup ran at
  XSTAMPV(1, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32));
  issue_x(0);
  issue_w(0, 0);
  __syncthreads();                    // pro_s
  stash_x(0);
  stash_w(0);
  __syncthreads();
  XSTAMP(2);
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
        bf16x8 a01[2][MT][2], a2[MT], bfr[NT][3];
        auto rd_a = [&](int tl, int t) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, Wb[((tl * 2 * TERMS) + t * 2 + h) * CB + mt * 32 + r32]);
            if (t == 2) a2[mt] = v; else a01[tl & 1][mt][t] = v;
          }
        };
        auto rd_b = [&](int tl, int t) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bfr[nt][t] = __builtin_bit_cast(bf16x8, Xs[(t * 2 + h) * PEP + off[nt] + ky * PITCH + tl]);
        };
        auto mm = [&](int tl, int ta, int tb) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_bf16(ta == 2 ? a2[mt] : a01[tl & 1][mt][ta], bfr[nt][tb], acc[mt][nt]);
        };
        if constexpr (TERMS == 1) {      // bf16 mode: one product per tap
          rd_a(0, 0); rd_b(0, 0);
#pragma unroll
          for (int tl = 0; tl < KS; ++tl) {
            mm(tl, 0, 0);
            if (tl + 1 < KS) { rd_a(tl + 1, 0); rd_b(tl + 1, 0); }
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
      if (chunk < 7) XSTAMP(3 + 7 * chunk + 2 * ky);        // this row's MFMAs issued
      if (more) stash_w(buf ^ 1);       // the other half: its readers passed the barrier at the end of the previous row
      if (last_row && more) {
        __syncthreads();                // every wave is done with this chunk's input tile
        if (chunk < 7) XSTAMP(3 + 7 * chunk + 5);
        if (PIPE) store_x(); else stash_x((chunk + 1) * KC);
      }
      if (chunk < 7 && last_row) XSTAMP(3 + 7 * chunk + 6);
      __syncthreads();
      if (chunk < 7) XSTAMP(3 + 7 * chunk + 2 * ky + 1);    // past the row's barrier
      buf ^= 1;
    }
  }

  XSTAMP(60);
  x3_epilogue<MT, NT, TWL, EPI, true>(a, acc, b, ty, tx, cout0, tid, reinterpret_cast<float*>(smem), bias_s, tile, cblk, true);
  XSTAMP(61);
  XSTAMPV(59, __builtin_amdgcn_s_memrealtime());
  XSTAMPV(62, (unsigned long long)nchunks);
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
#ifdef WTPSE_PROBE
// per workgroup {s_memtime, s_memrealtime} at the start and the end of the main loop (thread 0): the clock the loop ran at
__device__ unsigned long long* g_x3r_clk = nullptr;
__device__ int g_x3r_stagger = 0;
#define RCLK(i) do { if (g_x3r_clk && threadIdx.x == 0) { unsigned long long* q = g_x3r_clk + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 2 * (i); \
    q[0] = __builtin_amdgcn_s_memtime(); q[1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define RCLK(i)
#endif

// ABL (tools/probe/x3r_abl.py only; 0 in the library): pieces of the main loop left out, to price them — 1 conversion VALU, 2 LDS
// stores, 4 weight-fragment loads, 8 input-fragment reads, 16 the chunk barrier, 32 the input tile's global loads.  Results are
// garbage with any bit set.
template <int WM, int MT, int NT, int TWL, int EPI, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_x3r_k(ConvX3Args a) {
  constexpr int KS = 3, TAPS = 9, PAD = 1;
  constexpr int PW = 4 / WM;                       // waves along the pixels
  constexpr int TW = 1 << TWL, TH = (PW * 32 * NT) / TW;
  constexpr int PITCH = TW + 2 * PAD, ROWS = TH + 2 * PAD;
  constexpr int PE = PITCH * ROWS;
  constexpr int PEP = (PE + 7) & ~7;
  constexpr int CBW = 32 * MT, CB = CBW * WM;
  constexpr int NACC = 16;
  constexpr int KC = 16;
  constexpr int XS_U4 = 6 * PEP;                   // 16-byte slots of one input image
  constexpr int MAIN_U4 = 2 * XS_U4;
  constexpr int PRO_MAX = 512;
  static_assert(MAIN_U4 * 4 >= PW * CB * 2 + 4 * CB + 4, "epilogue scratch aliases the operand images");
  constexpr int DUMP_U4 = MAIN_U4 + CB / 4 + (EPI == 2 ? CB : 0);
  __shared__ u32x4v smem[DUMP_U4 + 64];
  __shared__ float2 pro_s[PRO_MAX];
  u32x4v* Xs = smem;
  float* bias_s = reinterpret_cast<float*>(smem + MAIN_U4);

  RCLK(2);
#ifdef WTPSE_PROBE
  if (g_x3r_stagger > 0 && blockIdx.x < 512u) {        // first round only: the workgroup whose LDS allocation does not start at 0 waits
    const unsigned lds_base = __builtin_amdgcn_s_getreg(((8 - 1) << 11) | 6);
    if (lds_base != 0) for (int i = 0; i < g_x3r_stagger; ++i) __builtin_amdgcn_s_sleep(16);     // ~1k cycles per iteration
  }
#endif
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
  if (tid < CB) bias_s[tid] = (a.bias && cout0 + tid < a.Cout) ? a.bias[cout0 + tid] : 0.f;
  if (EPI == 2 && tid < CB) {
    const int c = cout0 + tid;
    const bool bn = c >= a.bn_c0 && c < a.bn_c1;
    float* q = bias_s + CB + tid;
    q[0] = (bn && a.bn_relu) ? a.bn_ss[2 * (c - a.bn_c0)] : 0.f;
    q[CB] = (bn && a.bn_relu) ? a.bn_ss[2 * (c - a.bn_c0) + 1] : 1.f;
    q[2 * CB] = bn ? a.bn_mean[c - a.bn_c0] : 0.f;
  }
  for (int c = tid; c < a.CinP; c += 256) {
    const bool first = c < a.C0;
    const float* pro = first ? a.pro0 : a.pro1;
    const int cl = first ? c : c - a.C0;
    const bool live = c < a.C0 + a.C1;
    pro_s[c] = !live ? make_float2(0.f, 0.f) : (pro ? make_float2(pro[2 * cl], pro[2 * cl + 1]) : make_float2(1.f, 0.f));
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
  constexpr unsigned TAP_B = 6u * 32u * 16u;       // bytes of one (chunk, 32-row block, tap): three terms x two k-halves x 32 rows x 16 B
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wx, (unsigned)nchunks * ncb32 * TAPS * TAP_B);
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
  // one item = 8 channels of one halo position: prologue (affine, ReLU; zero padding AFTER it), split, three 16-byte LDS stores
  u32x4v tq[3];
  auto convert_pair = [&](int c0, int i, int j) __attribute__((always_inline)) {
    float v0 = xv[i][2 * j], v1 = xv[i][2 * j + 1];
    const bool relu = (c0 < a.C0) ? (a.pro_relu & 1) : (a.pro_relu & 2);
    const int cg = min(c0 + ihalf[i] * 8 + 2 * j, PRO_MAX - 2);
    const float2 p0 = pro_s[cg], p1 = pro_s[cg + 1];
    v0 = fmaf(v0, p0.x, p0.y);
    v1 = fmaf(v1, p1.x, p1.y);
    v0 = relu ? fmaxf(v0, 0.f) : v0;
    v1 = relu ? fmaxf(v1, 0.f) : v1;
    v0 = iin[i] ? v0 : 0.f;
    v1 = iin[i] ? v1 : 0.f;
    unsigned q0, q1, q2;
    split3_pair(v0, v1, q0, q1, q2);
    tq[0][j] = q0;
    tq[1][j] = q1;
    tq[2][j] = q2;
  };
  // branch-free (the stores sit between MFMA groups): lanes without a position (the tail of a half's last 64-block) store to a
  // per-lane dump slot behind the images
  auto store_item = [&](int i, int xb) __attribute__((always_inline)) {
    const bool ok = ipos[i] >= 0;
    const int base = xb * XS_U4 + ihalf[i] * PEP + ipos[i];
#pragma unroll
    for (int t = 0; t < 3; ++t) Xs[ok ? base + t * 2 * PEP : DUMP_U4 + lane] = tq[t];
  };
  // piece p of a chunk's conversion: pair (p & 3) of item (p >> 2); an item's fourth pair is followed by its stores
  auto convert_piece = [&](int c0, int p, int xb, bool in_loop) __attribute__((always_inline)) {
    if (!((ABL & 1) && in_loop)) convert_pair(c0, p >> 2, p & 3);
    if ((p & 3) == 3 && !((ABL & 2) && in_loop)) store_item(p >> 2, xb);
  };

  // A fragments: ring of three tap slots
  // tap slots of weight fragments: 3 = two taps ahead.  (ABL 256: 9 slots = a whole chunk ahead — measured: the waits on these loads
  // shrink from 15 % to 5 % of the launch and the launch takes the same time, profiles/r04_x3r_ablation.txt: the chip is at its power
  // limit on this arithmetic, cycles saved come back as a lower clock; 72 registers for nothing)
  constexpr int RING = (ABL & 256) ? TAPS : 3;
  bf16x8 afr[RING][MT][3];
  bf16x8 adummy[3][MT][3];           // ABL 64 only
  auto issue_a1 = [&](int chunk, int tap, int slot, int t) __attribute__((always_inline)) {
    const unsigned base = ((unsigned)(chunk * ncb32 + cb32) * TAPS + (unsigned)tap) * TAP_B;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16x8 v = __builtin_bit_cast(bf16x8, buf_load4(rsw, wlane, base + (unsigned)mt * TAPS * TAP_B + (unsigned)t * 2u * 32u * 16u));
      if constexpr (ABL & 64) adummy[slot][mt][t] = v; else afr[slot][mt][t] = v;
    }
  };
  auto issue_a = [&](int chunk, int tap, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 3; ++t) issue_a1(chunk, tap, slot, t);
  };
  bf16x8 bfr[2][3];
  auto rd_b = [&](int xb, int tap, int nt, int set) __attribute__((always_inline)) {
    const u32x4v* X = Xs + xb * XS_U4 + off[nt] + (tap / 3) * PITCH + (tap % 3);
#pragma unroll
    for (int t = 0; t < 3; ++t) bfr[set][t] = __builtin_bit_cast(bf16x8, X[(t * 2 + h) * PEP]);
  };
  auto mm = [&](int slot, int nt, int set) __attribute__((always_inline)) {
    // the six cross terms, smallest first — per accumulator the order of conv_x3_k
    constexpr int TA[6] = {0, 1, 2, 0, 1, 0}, TB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if constexpr (ABL & 512) {
          // probe only (results are garbage): the same multiply-adds issued as TWO v_mfma_f32_16x16x32_bf16 on the same fragment
          // registers — what the other MFMA shape would make of this operand supply (clock, cycles)
          typedef float f32x4p __attribute__((ext_vector_type(4)));
          f32x16& c = acc[mt][nt];
          f32x4p c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[slot][mt][TA[q]], bfr[set][TB[q]], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[slot][mt][TA[q]], bfr[set][TB[q]], c1, 0, 0, 0);
          c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3]; c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
        } else
        acc[mt][nt] = mfma_bf16(afr[slot][mt][TA[q]], bfr[set][TB[q]], acc[mt][nt]);
      }
  };

  constexpr int NPIECE = NIT * 4;
  constexpr int NGRP = TAPS * NT;                  // MFMA groups (tap, column tile) per chunk
  static_assert(NPIECE <= NGRP, "one conversion piece per MFMA group");
  constexpr int G0 = NGRP - NPIECE;                // the pieces ride behind the last NPIECE groups

  issue_x(0);
  if constexpr (RING == TAPS) {
    x3_static_for<TAPS>([&](auto tc) __attribute__((always_inline)) { issue_a(0, decltype(tc)::value, decltype(tc)::value); });
  } else {
    issue_a(0, 0, 0);
    issue_a(0, 1, 1);
  }
  __syncthreads();                                 // pro_s
  x3_static_for<NPIECE>([&](auto pc) __attribute__((always_inline)) { convert_piece(0, decltype(pc)::value, 0, false); });
  issue_x(KC);
  __syncthreads();
  RCLK(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int xb = chunk & 1;
    const int chn = min(chunk + 1, nchunks - 1);   // fragments prefetched past the last chunk are never used
    const int c0n = (chunk + 1) * KC;
    __builtin_amdgcn_sched_barrier(0);
    rd_b(xb, 0, 0, 0);
    if constexpr (ABL & 8) rd_b(xb, 0, NT - 1, 1);
    x3_static_for<NGRP>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      constexpr int tap = g / NT, nt = g % NT;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (RING == TAPS) {
        // the fragments of tap - 1 have been consumed: their slot takes the NEXT chunk's tap - 1 (nine taps = 216 MFMAs per wave ahead)
        if constexpr (nt == 0 && tap >= 1 && !(ABL & 4)) issue_a(chn, tap - 1, tap - 1);
      } else if constexpr ((ABL & 128) != 0 && NT >= 3) {       // spread: one term per group
        if constexpr (nt < 3) {
          if constexpr (tap + 2 < TAPS) issue_a1(chunk, tap + 2, (tap + 2) % 3, nt);
          else issue_a1(chn, tap + 2 - TAPS, (tap + 2) % 3, nt);
        }
      } else if constexpr (nt == 0 && !(ABL & 4)) {
        if constexpr (tap + 2 < TAPS) issue_a(chunk, tap + 2, (tap + 2) % 3);
        else issue_a(chn, tap + 2 - TAPS, (tap + 2) % 3);
      }
      if constexpr (g + 1 < NGRP && !(ABL & 8)) rd_b(xb, (g + 1) / NT, (g + 1) % NT, (g + 1) & 1);
      // the next group's fragment reads go out in FRONT of this group's six MFMAs (left alone, the scheduler re-uses the registers of
      // the current b1 / b2 for them and sinks the reads behind the fourth MFMA: 64 cycles in front of their s_waitcnt)
      __builtin_amdgcn_sched_barrier(0);
      mm(tap % RING, nt, g & 1);
      if constexpr (g >= G0) convert_piece(c0n, g - G0, xb ^ 1, true);
    });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (RING == TAPS && !(ABL & 4)) issue_a(chn, TAPS - 1, TAPS - 1);
    if constexpr (!(ABL & 32)) issue_x((chunk + 2) * KC);
    if constexpr (!(ABL & 16)) __syncthreads();
  }
  RCLK(1);
  if constexpr (ABL != 0) {       // keep everything the ablations skipped alive
    if (a.B < 0) {
#pragma unroll
      for (int i = 0; i < NIT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[0][0][0] += xv[i][j];
      acc[0][0][1] += __builtin_bit_cast(float, tq[0][0] ^ tq[1][1] ^ tq[2][2]);
      if constexpr (ABL & 64) {
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
          for (int t = 0; t < 3; ++t) acc[0][0][2] += (float)adummy[sl][0][t][0];
      }
    }
  }

  x3_epilogue<MT, NT, TWL, EPI, true, WM>(a, acc, b, ty, tx, cout0, tid, reinterpret_cast<float*>(smem), bias_s, tile, cblk, true);
  RCLK(3);
}

// ------------------------------------------------------------------------------------------------
// Weight packing for the x3 path: OIHW fp32 -> bf16 triples in the kernel's LDS image order.
//   forward : rows = Cout, k = Cin, tap t          element = w[co][ci][t]
//   dgrad   : rows = Cin,  k = Cout, tap T-1-t     element = w[co][ci][T-1-t]   (the data gradient is the forward kernel on dY)
// layout: [k chunk of 16][row block of 32][tap][term 3][k half 2][row 32][8 k] (unsigned short); zero padded.
// desc: n_desc x 8 ints {w_off, Cout, Cin, taps, xf_off, xd_off(-1: none), 0, 0}; x*_off in unsigned shorts.
__global__ __launch_bounds__(256) void pack_weights_x3_k(const float* __restrict__ params, const int* __restrict__ desc,
                                                         unsigned short* __restrict__ packed) {
  const int* d = desc + blockIdx.y * 8;
  const int w_off = d[0], Co = d[1], Ci = d[2], T = d[3];
  const float* w = params + w_off;
  for (int dir = 0; dir < 2; ++dir) {
    const int base = d[4 + dir];
    if (base < 0) continue;
    const int R = dir == 0 ? Co : Ci, K = dir == 0 ? Ci : Co;
    const int RP = (R + 31) & ~31, KP = (K + 15) & ~15;
    const int n = KP * RP * T;                     // (row, k, tap) triples
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
      // e enumerates [chunk][rb][tap][half][row32][k8]
      const int k8 = e & 7, row32 = (e >> 3) & 31, hh = (e >> 8) & 1;
      int q = e >> 9;
      const int t = q % T; q /= T;
      const int rb = q % (RP / 32), chunk = q / (RP / 32);
      const int row = rb * 32 + row32, k = chunk * 16 + hh * 8 + k8;
      float v = 0.f;
      if (row < R && k < K) v = dir == 0 ? w[(row * Ci + k) * T + t] : w[(k * Ci + row) * T + (T - 1 - t)];
      unsigned q0, q1, q2;
      split3_pair(v, 0.f, q0, q1, q2);
      const size_t slot = ((((size_t)chunk * (RP / 32) + rb) * T + t) * 6);
      unsigned short* o = packed + base;
      o[((slot + 0 * 2 + hh) * 32 + row32) * 8 + k8] = (unsigned short)(q0 & 0xFFFFu);
      o[((slot + 1 * 2 + hh) * 32 + row32) * 8 + k8] = (unsigned short)(q1 & 0xFFFFu);
      o[((slot + 2 * 2 + hh) * 32 + row32) * 8 + k8] = (unsigned short)(q2 & 0xFFFFu);
    }
  }
}

extern "C" int wtpse_pack_conv_weights_x3(const float* params, const int* desc, int n_desc, unsigned short* packed, void* stream) {
  WTPSE_REQUIRE(params && desc && packed && n_desc > 0);
  hipLaunchKernelGGL(pack_weights_x3_k, dim3(48, n_desc), dim3(256), 0, (hipStream_t)stream, params, desc, packed);
  return wtpse_status();
}

// 128-pixel tiles: only where the 256-pixel tiling gives under two workgroups per CU on a 16-wide map (the deepest level)
static bool x3_small_tiles(int B, int H, int W, int CoutP, bool mt2) {
  if (W > 16 || mt2) return false;
  return B * ceil_div(H, 16) * ceil_div(W, 16) * (CoutP / 32) < 512;
}

// 3x3 launches take conv_x3r_k (register-fed weights) unless WTPSE_X3R=0 / wtpse_x3r_enable(0): the two kernels give bitwise the
// same results on the same tiles, so the switch is a pure A/B of the operand supply.
static int g_x3r = [] { const char* e = getenv("WTPSE_X3R"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }();
extern "C" int wtpse_x3r_enable(int on) {
  const int was = g_x3r;
  if (on >= 0) g_x3r = on > 2 ? 2 : on;
  return was;
}

#ifdef WTPSE_PROBE
static int g_x3r_abl = 0;
extern "C" int wtpse_probe_x3r_abl(int abl) { g_x3r_abl = abl; return 0; }
extern "C" int wtpse_probe_x3r_clock(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_x3r_clk), &p, sizeof(p)); }
extern "C" int wtpse_probe_x3r_stagger(int n) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_x3r_stagger), &n, sizeof(n)); }
#endif

// bf16 terms per fp32 operand in every x3 kernel of the library (this file and wgrad_r.hip): 3 = fp32 accuracy (default), 1 = the
// bf16 mode (environment WTPSE_X3_TERMS=1 or wtpse_x3_terms(1); include/wtpse_hip.h)
int g_x3_terms = [] { const char* e = getenv("WTPSE_X3_TERMS"); return (e && e[0] == '1') ? 1 : 3; }();
extern "C" int wtpse_x3_terms(int terms) {
  const int was = g_x3_terms;
  if (terms == 1 || terms == 3) g_x3_terms = terms;
  return was;
}

// XCD-aware workgroup order of the x3 convolutions (ConvX3Args::xcd_tiles); WTPSE_X3_XCD=0 / wtpse_x3_xcd(0): dispatch order.
// Same workgroups, same results — a pure A/B of the order.
static int g_x3_xcd = [] { const char* e = getenv("WTPSE_X3_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
extern "C" int wtpse_x3_xcd(int on) {
  const int was = g_x3_xcd;
  if (on >= 0) g_x3_xcd = on ? 1 : 0;
  return was;
}

// half: the 64-channel blocks on 128-pixel tiles (x3_half below) — conv_x3r_k<2, 1, 2, ...>
template <int KS, int MT, int EPI, int TERMS = 3>
static int launch_x3(const ConvX3Args& a, hipStream_t st, bool half = false) {
  ConvX3Args args = a;
  const bool narrow = a.W <= 16;
  const bool small = MT == 1 && x3_small_tiles(a.B, a.H, a.W, a.CoutP, false);
  const int TW = narrow ? 16 : 32, TH = ((small || half) ? 128 : 256) / TW;
  args.tiles_x = ceil_div(a.W, TW);
  args.tiles_y = ceil_div(a.H, TH);
  dim3 grid((unsigned)(a.B * args.tiles_x * args.tiles_y), (unsigned)ceil_div(a.CoutP, 32 * MT));
  args.xcd_tiles = (g_x3_xcd && grid.x % 8 == 0 && (long long)grid.x * grid.y >= 64) ? (int)grid.x / 8 : 0;
  const bool in_launch = tail_in_launch((long long)grid.x * grid.y);     // else: the stand-alone finalize kernel behind the launch
  if (!in_launch) args.tail.tickets = args.ftail.tickets = nullptr;
  if (args.tail.tickets) bnb_tail_geometry(args.tail, (int)grid.x, a.Cout, (double)a.B * a.H * a.W);
  if (args.ftail.tickets) bnf_tail_geometry(args.ftail, (int)grid.x, a.Cout, (double)a.B * a.H * a.W);
  // g_x3r: 1 = conv_x3r_k where it measured at least as fast (64-channel blocks: +1..10 % on the forward launches, +-1 % on the data
  // gradients), 2 = everywhere (32-channel blocks run 8-20 % SLOWER on it: half the MFMAs per converted input element), 0 = nowhere
  if (KS == 3 && TERMS == 3 && (g_x3r == 2 || (g_x3r == 1 && MT == 2))) {
    if constexpr (KS == 3 && TERMS == 3) {
      if (half) {
        if constexpr (MT == 2) {
          if (narrow) hipLaunchKernelGGL((conv_x3r_k<2, 1, 2, 4, EPI>), grid, dim3(256), 0, st, args);
          else hipLaunchKernelGGL((conv_x3r_k<2, 1, 2, 5, EPI>), grid, dim3(256), 0, st, args);
        }
      } else if (small) {
        if constexpr (MT == 1) hipLaunchKernelGGL((conv_x3r_k<1, 1, 1, 4, EPI>), grid, dim3(256), 0, st, args);
      } else if (MT == 2) {
#ifdef WTPSE_PROBE
        if (g_x3r_abl && !narrow && EPI == 0) {
          switch (g_x3r_abl) {
#define ABLCASE(n) case n: hipLaunchKernelGGL((conv_x3r_k<2, 1, 4, 5, 0, n>), grid, dim3(256), 0, st, args); break;
            ABLCASE(1) ABLCASE(2) ABLCASE(3) ABLCASE(4) ABLCASE(8) ABLCASE(16) ABLCASE(32) ABLCASE(35) ABLCASE(39) ABLCASE(47) ABLCASE(63)
            ABLCASE(64) ABLCASE(128) ABLCASE(99) ABLCASE(163) ABLCASE(256) ABLCASE(260) ABLCASE(291) ABLCASE(319) ABLCASE(512) ABLCASE(575)
#undef ABLCASE
            default: return WTPSE_EINVAL;
          }
          return wtpse_status();
        }
#endif
        if (narrow) hipLaunchKernelGGL((conv_x3r_k<2, 1, 4, 4, EPI>), grid, dim3(256), 0, st, args);
        else hipLaunchKernelGGL((conv_x3r_k<2, 1, 4, 5, EPI>), grid, dim3(256), 0, st, args);
      } else {
        if (narrow) hipLaunchKernelGGL((conv_x3r_k<1, 1, 2, 4, EPI>), grid, dim3(256), 0, st, args);
        else hipLaunchKernelGGL((conv_x3r_k<1, 1, 2, 5, EPI>), grid, dim3(256), 0, st, args);
      }
    }
  } else if (small) {
    if constexpr (MT == 1) hipLaunchKernelGGL((conv_x3_k<KS, 1, 4, EPI, 1, TERMS>), grid, dim3(256), 0, st, args);
  } else if (narrow)
    hipLaunchKernelGGL((conv_x3_k<KS, MT, 4, EPI, 2, TERMS>), grid, dim3(256), 0, st, args);
  else
    hipLaunchKernelGGL((conv_x3_k<KS, MT, 5, EPI, 2, TERMS>), grid, dim3(256), 0, st, args);
  int rc = wtpse_status();
  if (rc == 0 && !in_launch)
    rc = tail_after_launch(a.tail, a.ftail, a.stats, (int)grid.x, a.Cout, a.bn_c0, a.bn_c1, a.bn_mean, (long long)a.B * a.H * a.W, st);
  return rc;
}

static bool x3_mt2(int B, int H, int W, int CoutP) {
  const int TW = W <= 16 ? 16 : 32, TH = 256 / TW;
  const int tiles = B * ceil_div(W, TW) * ceil_div(H, TH);
  bool mt2 = (CoutP % 64 == 0) && tiles * (CoutP / 64) >= 512;
  // tuning override 1 | 2, read ONCE per process: the size queries (stats blocks) and the launches must agree on the tiling, also
  // when a recorded launch plan replays the launches later
  static const int mt_override = [] { const char* e = getenv("WTPSE_X3_MT"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0; }();
  if (mt_override == 1) mt2 = false;
  if (mt_override == 2 && CoutP % 64 == 0) mt2 = true;
  return mt2;
}

// Mid-sized launches (3x3, output channels a multiple of 64, too few 256-pixel tiles for the 64-channel blocks to give every CU two
// workgroups): instead of falling back to the 32-channel blocks — which convert every input element for half as many MFMAs — the
// 64-channel blocks of conv_x3r_k on 128-PIXEL tiles (2 x 2 waves of 32 channels x 64 pixels), if that yields at least
// g_x3_half_min workgroups.  WTPSE_X3_HALF=0: off; WTPSE_X3_HALF_MIN: the threshold (default 256 = one per CU: measured on the 16x16
// maps of down4 at B=32, 256 such workgroups beat 512 of the 32-channel blocks on 128-pixel tiles by 11-13 %; forward of down3 /
// up2.conv1 -8..-10 %: profiles/r04_microbench_x3.txt).
static int g_x3_half_min = [] { const char* e = getenv("WTPSE_X3_HALF"); if (e && e[0] == '0') return 0;
                                const char* m = getenv("WTPSE_X3_HALF_MIN"); const int v = m ? atoi(m) : 256; return v > 0 ? v : 256; }();
static bool x3_half(int B, int H, int W, int CoutP, int ksize) {
  if (ksize != 3 || g_x3_half_min == 0 || g_x3r == 0 || g_x3_terms != 3 || CoutP % 64 != 0 || x3_mt2(B, H, W, CoutP)) return false;
  const int TW = W <= 16 ? 16 : 32, TH = 128 / TW;
  return B * ceil_div(W, TW) * ceil_div(H, TH) * (CoutP / 64) >= g_x3_half_min;
}

// workgroups along x of a wtpse_conv_fwd_x3 launch = rows of its `stats` partials
extern "C" int wtpse_conv_x3_stats_blocks(int B, int H, int W, int Cout, int ksize) {
  const int CoutP = (Cout + 31) & ~31;
  const bool small = x3_half(B, H, W, CoutP, ksize) || x3_small_tiles(B, H, W, CoutP, x3_mt2(B, H, W, CoutP));
  const int TW = W <= 16 ? 16 : 32, TH = (small ? 128 : 256) / TW;
  return B * ceil_div(W, TW) * ceil_div(H, TH);
}

static int conv_x3_impl(const float* in0, int C0, const float* in1, int C1, const unsigned short* wpacked,
                        const float* bias, const float* pro0, const float* pro1, int pro_relu, float* out0, float* out1,
                        int Csplit, float* stats, int B, int H, int W, int Cout, int ksize, int relu_out,
                        const float* mask_ref, const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0, int bn_c1,
                        void* stream, BnbTail tail = bnb_tail_none(), BnfTail ftail = bnf_tail_none()) {
  WTPSE_REQUIRE(in0 && wpacked && out0 && B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0);
  WTPSE_REQUIRE(ksize == 1 || ksize == 3);
  WTPSE_REQUIRE((C1 == 0) == (in1 == nullptr));
  WTPSE_REQUIRE(Csplit > 0 && Csplit <= Cout && ((Csplit == Cout) == (out1 == nullptr)));
  WTPSE_REQUIRE(Csplit == Cout || Csplit % 16 == 0);
  WTPSE_REQUIRE(!(stats && relu_out));
  const bool bnb = bn_mean != nullptr;
  WTPSE_REQUIRE(bnb || !(stats && mask_ref));
  WTPSE_REQUIRE(bnb || !(mask_ref && out1));
  WTPSE_REQUIRE(!bnb || (mask_ref && stats && bn_ss && !bias && !relu_out && bn_c0 >= 0 && bn_c0 < bn_c1 && bn_c1 <= Cout &&
                         bn_c0 % 16 == 0 && (bn_c1 % 16 == 0 || bn_c1 == Cout)));
  WTPSE_REQUIRE(C1 == 0 || C0 % 16 == 0);

  ConvX3Args a;
  a.in0 = in0; a.in1 = in1; a.wx = wpacked; a.bias = bias; a.pro0 = pro0; a.pro1 = pro1; a.out0 = out0; a.out1 = out1;
  a.stats = stats; a.mask = mask_ref;
  a.bn_ss = bn_ss; a.bn_mean = bn_mean; a.bn_relu = bn_relu; a.bn_c0 = bnb ? bn_c0 : 0; a.bn_c1 = bnb ? bn_c1 : 0;
  WTPSE_REQUIRE(!tail.tickets || (bnb && tail.partial2 && tail.gamma && tail.invstd && tail.coef && tail.dgamma && tail.dbeta));
  a.tail = tail;
  WTPSE_REQUIRE(!ftail.tickets || (!bnb && stats && ftail.partial2 && ftail.gamma && ftail.beta && ftail.scale_shift && ftail.save_mean &&
                                   ftail.save_invstd && (ftail.rmean == nullptr) == (ftail.rvar == nullptr)));
  a.ftail = ftail;
  a.B = B; a.H = H; a.W = W; a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1; a.CinP = (a.Cin + 15) & ~15;
  a.Cout = Cout; a.CoutP = (Cout + 31) & ~31; a.Csplit = Csplit; a.pro_relu = pro_relu; a.relu_out = relu_out;
  a.tiles_x = a.tiles_y = 0;
  hipStream_t st = (hipStream_t)stream;
  const bool mt2 = x3_mt2(B, H, W, a.CoutP);
  WTPSE_REQUIRE(a.CinP <= (mt2 ? 512 : 256));            // prologue coefficients staged in LDS (conv_x3_k: PRO_MAX)
#define X3T(KS, M, T) (bnb ? launch_x3<KS, M, 2, T>(a, st) : mask_ref ? launch_x3<KS, M, 1, T>(a, st) : launch_x3<KS, M, 0, T>(a, st))
#define X3(KS, M) (g_x3_terms == 1 ? X3T(KS, M, 1) : X3T(KS, M, 3))
  if (ksize == 3 && x3_half(B, H, W, a.CoutP, 3))
    return bnb ? launch_x3<3, 2, 2, 3>(a, st, true) : mask_ref ? launch_x3<3, 2, 1, 3>(a, st, true) : launch_x3<3, 2, 0, 3>(a, st, true);
  if (ksize == 3) return mt2 ? X3(3, 2) : X3(3, 1);
  return mt2 ? X3(1, 2) : X3(1, 1);
#undef X3
#undef X3T
}

// Same contract as wtpse_conv_fwd (include/wtpse_hip.h) with `wpacked` in the x3 layout.  Cout <= 16 runs on a 32-row
// tile with the upper rows idle (zero weights, stores dropped by the range check): the bf16 MFMAs are cheap enough.
extern "C" int wtpse_conv_fwd_x3(const float* in0, int C0, const float* in1, int C1, const unsigned short* wpacked,
                                 const float* bias, const float* pro0, const float* pro1, int pro_relu, float* out0, float* out1,
                                 int Csplit, float* stats, int B, int H, int W, int Cout, int ksize, int relu_out,
                                 const float* mask_ref, void* stream) {
  return conv_x3_impl(in0, C0, in1, C1, wpacked, bias, pro0, pro1, pro_relu, out0, out1, Csplit, stats, B, H, W, Cout, ksize,
                      relu_out, mask_ref, nullptr, nullptr, 0, 0, 0, stream);
}

// Data gradient that also performs the first half of the BatchNorm backward of the layer it flows into (include/wtpse_hip.h).
extern "C" int wtpse_dgrad_x3_bnb(const float* dy, int C, const unsigned short* wpacked, float* out0, float* out1, int Csplit,
                                  const float* bn_y, const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0, int bn_c1,
                                  float* stats, int B, int H, int W, int Cout, int ksize, void* stream) {
  WTPSE_REQUIRE(bn_y && bn_ss && bn_mean && stats);
  return conv_x3_impl(dy, C, nullptr, 0, wpacked, nullptr, nullptr, nullptr, 0, out0, out1, Csplit, stats, B, H, W, Cout, ksize, 0,
                      bn_y, bn_ss, bn_mean, bn_relu, bn_c0, bn_c1, stream);
}

// wtpse_conv_fwd_bnf (conv.hip), x3 layout
extern "C" int wtpse_conv_fwd_x3_ftail(const float* in0, int C0, const float* in1, int C1, const unsigned short* wpacked,
                                       const float* bias, const float* pro0, const float* pro1, int pro_relu, float* out0,
                                       float* stats, const BnfTail* ftail, int B, int H, int W, int Cout, int ksize, void* stream) {
  WTPSE_REQUIRE(ftail && stats);
  return conv_x3_impl(in0, C0, in1, C1, wpacked, bias, pro0, pro1, pro_relu, out0, nullptr, Cout, stats, B, H, W, Cout, ksize, 0,
                      nullptr, nullptr, nullptr, 0, 0, 0, stream, bnb_tail_none(), *ftail);
}

// wtpse_dgrad_bnb_coef (conv.hip), x3 layout
extern "C" int wtpse_dgrad_x3_bnb_tail(const float* dy, int C, const unsigned short* wpacked, float* out0, float* out1, int Csplit,
                                       const float* bn_y, const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0,
                                       int bn_c1, float* stats, const BnbTail* tail, int B, int H, int W, int Cout, int ksize,
                                       void* stream) {
  WTPSE_REQUIRE(bn_y && bn_ss && bn_mean && stats && tail);
  return conv_x3_impl(dy, C, nullptr, 0, wpacked, nullptr, nullptr, nullptr, 0, out0, out1, Csplit, stats, B, H, W, Cout, ksize, 0,
                      bn_y, bn_ss, bn_mean, bn_relu, bn_c0, bn_c1, stream, *tail);
}

// ================================================================================================
// Weight gradient on the bf16 matrix cores (x3 arithmetic):  dW[co][ci][t] = sum_{b,y,x} dY[b,co,y,x] * X[b,ci,y+dy-1,x+dx-1]
// (conv weight gradients of the reference hot path: autograd of algorithms.py:882-888,926-933).  GEMM with K = pixels:
// A = dY[co][pixel], B = X[ci][pixel + tap]; one accumulator tile [32 co][32 ci] per tap.  The bf16 MFMA wants 8
// consecutive k (pixels) per lane and NCHW gives 8 consecutive pixels only at aligned addresses, while the taps shift B by
// -1/0/+1 pixels.  So both operands are kept PIXEL-major in LDS ([8-channel group][pixel][8 ch], bf16 triples, written by
// the same split-on-load loader as the forward kernel) and read with ds_read_b64_tr_b16: a 16-lane group fetches a block of
// 4 pixels x 16 channels and receives it channel-major, i.e. 4 consecutive k of "its" channel — a tap is then just a
// different starting pixel.  Plane strides are = 4 or 12 (mod 16) 16-byte slots: the 4 planes a 32-lane half touches land
// on disjoint banks.
//   Q  (64 co x 64 ci per workgroup): the 4 waves are the quadrants; every wave walks all k-steps of a 64-pixel tile.
//   !Q (32 co x 32 ci):               the 4 waves split the k-steps of a 128-pixel tile; partial sums meet in LDS at the end.
// Accumulators persist over a workgroup's tiles (strided by the k-split); per-workgroup slabs are folded by wgrad_reduce_k.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct WgradX3Args {
  const float* dy;
  const float* x0;
  const float* x1;
  const float* pro0;
  const float* pro1;
  float* slab;    // [ksplit][Cout][Cin][taps]
  int B, H, W, C0, C1, Cin, Cout;
  int pro_relu;
  int tiles_x, tiles_y, ntiles;
  int nci;        // ci blocks
};

__device__ __forceinline__ bf16x8 lds_tr8(const u32x4v* base, int byte_off) {
  // two transposed reads: pixels +0..3 and +4..7 (4 slots = 64 bytes further)
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const char __attribute__((address_space(3)))* p =
      (const char __attribute__((address_space(3)))*)base + byte_off;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 64));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

constexpr int pad_plane(int n) {   // smallest m >= n with m % 16 in {4, 12}
  for (int m = n;; ++m)
    if (m % 16 == 4 || m % 16 == 12) return m;
}

// (ablation hooks of tools/probe/wgrad_abl.py: defined in probe builds only)
#ifdef EXP_W_NOSPLIT
#define WSPLIT(a, b, q0, q1, q2) do { q0 = __builtin_bit_cast(unsigned, a); q1 = __builtin_bit_cast(unsigned, b); q2 = q0 ^ q1; } while (0)
#else
#define WSPLIT(a, b, q0, q1, q2) split3_pair(a, b, q0, q1, q2)
#endif
#ifdef EXP_W_NOSTORE
#define WSTORE(dst, v) asm volatile("" :: "v"(v))
#else
#define WSTORE(dst, v) dst = v
#endif
template <int KS, int TWL, bool Q>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_k(WgradX3Args a) {
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  constexpr int NPX = Q ? 64 : 128;
  constexpr int TW = 1 << TWL, TH = NPX >> TWL;
  constexpr int PITCH = TW + 2 * PAD, ROWS = TH + 2 * PAD;
  constexpr int PE = PITCH * ROWS;
  constexpr int NCG = Q ? 8 : 4;                      // 8-channel groups per image
  constexpr int BLK = NCG * 8;                        // channels per workgroup block (co and ci)
  constexpr int XP = pad_plane(PE), YP = pad_plane(NPX);
  constexpr int XS_U4 = 3 * NCG * XP, YS_U4 = 3 * NCG * YP;
  constexpr int RED_U4 = Q ? 0 : 4 * 1024 / 4;        // !Q: [4 waves][32][32] floats
  constexpr int SM_U4 = (XS_U4 + YS_U4) > RED_U4 ? (XS_U4 + YS_U4) : RED_U4;
  __shared__ u32x4v smem[SM_U4];
  u32x4v* Xs = smem;
  u32x4v* Ys = smem + XS_U4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // one k-slice per XCD (see conv_wgrad_k): its workgroups read the same tiles
  int bx = blockIdx.x, ky = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int k = lin >> 3;
    ky = (lin & 7) + 8 * (k / (int)gridDim.x);
    bx = k % (int)gridDim.x;
  }
  const int cin0 = (bx % a.nci) * BLK, cout0 = (bx / a.nci) * BLK;
  const int HW = a.H * a.W;
  const int wco = Q ? (wave >> 1) : 0, wci = Q ? (wave & 1) : 0;
  const bool any_pro = a.pro0 != nullptr || a.pro1 != nullptr || a.pro_relu != 0;

  // transposed-read lane geometry: group g = lane >> 4 (k half h = g >> 1, channel half g & 1), q = row of the 4-pixel block,
  // p = which 4 of the group's 16 channels this lane addresses
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = g >> 1;
  const int cgl = 2 * (g & 1) + (p >> 1);              // 8-channel group within the wave's 32 channels
  const int a_base = (((wco * 4 + cgl) * YP) + 8 * h + q) * 16 + 8 * (p & 1);
  const int b_base = (((wci * 4 + cgl) * XP) + 8 * h + q) * 16 + 8 * (p & 1);

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // loader items: 64-position blocks [group][block], dealt to the waves round-robin (a wave's group is uniform)
  constexpr int PBX = (PE + 63) / 64, NITX = (NCG * PBX + 3) / 4;
  constexpr int PBY = NPX / 64, NITY = NCG * PBY / 4;
  const int tiles_per_img = a.tiles_x * a.tiles_y;

  for (int tile = ky; tile < a.ntiles; tile += gridDim.y) {
    const int b = tile / tiles_per_img;
    const int trem = tile - b * tiles_per_img;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(a.dy + (size_t)b * a.Cout * HW, (unsigned)a.Cout * HW * 4u);
    const __amdgpu_buffer_rsrc_t rsx0 = make_rsrc(a.x0 + (size_t)b * a.C0 * HW, (unsigned)a.C0 * HW * 4u);
    const __amdgpu_buffer_rsrc_t rsx1 = a.x1 ? make_rsrc(a.x1 + (size_t)b * a.C1 * HW, (unsigned)a.C1 * HW * 4u) : rsx0;
    __syncthreads();   // the previous tile's MFMAs are done with the images
    // ---- X halo tile: (position, 8-channel group) items, two at a time (16 loads in flight per lane; the scheduling
    // barriers keep the compiler from hoisting every item's loads to the top, which spills beside 144 accumulators)
    auto x_item = [&](int i, float (&v)[8], int& cg, int& pos, bool& in, bool& first, int& cb, int& cn) {
      const int blk = __builtin_amdgcn_readfirstlane(i * 4 + wave);
      cg = blk / PBX;
      pos = (blk - cg * PBX) * 64 + lane;
      const int r = pos / PITCH, x = pos - r * PITCH;
      const int gy = ty * TH + r - PAD, gx = tx * TW + x - PAD;
      in = blk < NCG * PBX && pos < PE && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      if (blk >= NCG * PBX) pos = PE;                    // a padding block of the last round: nothing to store
      const unsigned vo = in ? (unsigned)(gy * a.W + gx) * 4u : BUF_OOB;
      const int c = cin0 + min(cg, NCG - 1) * 8;         // first channel of the group; C0 % 8 == 0: never straddles
      first = c < a.C0 || a.x1 == nullptr;
      const __amdgpu_buffer_rsrc_t rs = first ? rsx0 : rsx1;
      cb = first ? c : c - a.C0;
      cn = first ? a.C0 : a.C1;
#ifdef EXP_W_NOXLOAD
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_bit_cast(float, vo + j);
#else
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = buf_load(rs, vo, (unsigned)min(cb + j, cn) * (unsigned)HW * 4u);
#endif
    };
    auto x_finish = [&](float (&v)[8], int cg, int pos, bool in, bool first, int cb, int cn) {
      if (any_pro) {
        const bool relu = first ? (a.pro_relu & 1) : (a.pro_relu & 2);
        const float* pro = first ? a.pro0 : a.pro1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cgi = min(cb + j, cn - 1);
          const float sc = pro ? pro[2 * cgi] : 1.f, sh = pro ? pro[2 * cgi + 1] : 0.f;
          float w = fmaf(v[j], sc, sh);
          if (relu) w = fmaxf(w, 0.f);
          v[j] = in ? w : 0.f;
        }
      }
      if (pos < PE) {
        u32x4v t0, t1, t2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned q0, q1, q2;
          WSPLIT(v[2 * j], v[2 * j + 1], q0, q1, q2);
          t0[j] = q0; t1[j] = q1; t2[j] = q2;
        }
        WSTORE(Xs[(0 * NCG + cg) * XP + pos], t0);
        WSTORE(Xs[(1 * NCG + cg) * XP + pos], t1);
        WSTORE(Xs[(2 * NCG + cg) * XP + pos], t2);
      }
    };
#pragma unroll
    for (int i = 0; i < NITX; i += 2) {
      float v0[8], v1[8];
      int cg0, pos0, cb0, cn0, cg1 = 0, pos1 = PE, cb1 = 0, cn1 = 1;
      bool in0, f0, in1 = false, f1 = true;
      x_item(i, v0, cg0, pos0, in0, f0, cb0, cn0);
      if (i + 1 < NITX) x_item(i + 1, v1, cg1, pos1, in1, f1, cb1, cn1);
      x_finish(v0, cg0, pos0, in0, f0, cb0, cn0);
      if (i + 1 < NITX) x_finish(v1, cg1, pos1, in1, f1, cb1, cn1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- dY tile: (pixel, 8-channel group) items
#pragma unroll
    for (int i = 0; i < NITY; ++i) {
      const int blk = __builtin_amdgcn_readfirstlane(i * 4 + wave);
      const int cg = blk / PBY;
      const int px = (blk - cg * PBY) * 64 + lane;
      const int gy = ty * TH + (px >> TWL), gx = tx * TW + (px & (TW - 1));
      const unsigned vo = (gy < a.H && gx < a.W) ? (unsigned)(gy * a.W + gx) * 4u : BUF_OOB;
      const int c = cout0 + cg * 8;
      float v[8];
#ifdef EXP_W_NOYLOAD
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_bit_cast(float, vo + j);
#else
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = buf_load(rsy, vo, (unsigned)min(c + j, a.Cout) * (unsigned)HW * 4u);
#endif
      u32x4v t0, t1, t2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned q0, q1, q2;
        WSPLIT(v[2 * j], v[2 * j + 1], q0, q1, q2);
        t0[j] = q0; t1[j] = q1; t2[j] = q2;
      }
      WSTORE(Ys[(0 * NCG + cg) * YP + px], t0);
      WSTORE(Ys[(1 * NCG + cg) * YP + px], t1);
      WSTORE(Ys[(2 * NCG + cg) * YP + px], t2);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // ---- MFMAs: k-steps of 16 consecutive pixels of one tile row
    constexpr int NSTEP = Q ? NPX / 16 : NPX / 64;
    // software pipeline over (k-step, tap): the 6 transposed reads of the next tap's B fragments (and, at a step's last
    // tap, of the next step's A fragments) are issued before the 6 MFMAs of the current tap; the scheduling barriers keep
    // the compiler from hoisting reads further ahead (it did, and spilled the fragments beside the 144 accumulators)
    auto step_p0 = [&](int s) { return (Q ? s : wave * NSTEP + s) * 16; };
    auto load_a = [&](int s, bf16x8 (&af)[3]) {
      const int p0 = step_p0(s);
#pragma unroll
      for (int t = 0; t < 3; ++t) af[t] = lds_tr8(Ys, a_base + (t * NCG * YP + p0) * 16);
    };
    auto load_b = [&](int s, int tap, bf16x8 (&bfr)[3]) {
      const int p0 = step_p0(s);
      const int toff = ((p0 >> TWL) + tap / KS) * PITCH + (p0 & (TW - 1)) + tap % KS;
#pragma unroll
      for (int t = 0; t < 3; ++t) bfr[t] = lds_tr8(Xs, b_base + (t * NCG * XP + toff) * 16);
    };
    bf16x8 af[2][3], bfr[2][3];
    load_a(0, af[0]);
    load_b(0, 0, bfr[0]);
#pragma unroll
    for (int idx = 0; idx < NSTEP * TAPS; ++idx) {
      const int s = idx / TAPS, tap = idx % TAPS;
      const int cur = idx & 1, acur = s & 1;
      if (idx + 1 < NSTEP * TAPS) {
        load_b((idx + 1) / TAPS, (idx + 1) % TAPS, bfr[cur ^ 1]);
        if (tap == TAPS - 1) load_a(s + 1, af[acur ^ 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef EXP_W_NOMFMA
      continue;
#endif
      f32x16 c = acc[tap];
      c = mfma_bf16(af[acur][0], bfr[cur][2], c);
      c = mfma_bf16(af[acur][1], bfr[cur][1], c);
      c = mfma_bf16(af[acur][2], bfr[cur][0], c);
      c = mfma_bf16(af[acur][0], bfr[cur][1], c);
      c = mfma_bf16(af[acur][1], bfr[cur][0], c);
      c = mfma_bf16(af[acur][0], bfr[cur][0], c);
      acc[tap] = c;
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- slab[ky][co][ci][t]
  float* slab = a.slab + (size_t)ky * a.Cout * a.Cin * TAPS;
  if constexpr (Q) {
    const int ci = cin0 + wci * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cout0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (co < a.Cout && ci < a.Cin) {
        float* dst = slab + ((size_t)co * a.Cin + ci) * TAPS;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) dst[t] = acc[t][r];
      }
    }
  } else {
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        red[(wave * 32 + co) * 32 + (lane & 31)] = acc[t][r];
      }
      __syncthreads();
      for (int e = tid; e < 1024; e += 256) {
        const int co = e >> 5, ci = e & 31;
        const float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
        if (cout0 + co < a.Cout && cin0 + ci < a.Cin) slab[((size_t)(cout0 + co) * a.Cin + cin0 + ci) * TAPS + t] = v;
      }
    }
  }
}

static bool wgrad_x3_quadrants(int Cin, int Cout) { return Cin % 64 == 0 && Cout % 64 == 0; }

extern "C" int wtpse_wgrad_x3_supported(int Cin, int Cout, int ksize, int C0) {
  return ksize == 3 && Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0 && C0 % 8 == 0;
}

// Pixel tiles of the weight gradient are 16 wide (16x4 / 16x8) on every map: the halo tile of X is then 18x6 = 108 (18x10 = 180)
// positions instead of the 34x4 = 136 (34x6 = 204) of a 32-wide tile — a fifth less to load, split and store in a kernel that
// waits for exactly that (layer set 2022 -> 1864 us; the forward kernel prefers 32x8: its 16x16 form measured 6-8 % slower).
// WTPSE_X3_WGRAD_TW32=1 restores the 32-wide tiles on maps wider than 16 (comparison runs).
static bool wgrad_x3_tw16(int W) {
  if (W <= 16) return true;
  static const bool tw32 = [] { const char* e = getenv("WTPSE_X3_WGRAD_TW32"); return e && e[0] == '1'; }();   // once per process
  return !tw32;
}

extern "C" int wtpse_wgrad_x3_ksplit(int B, int H, int W, int Cin, int Cout) {
  const bool q = wgrad_x3_quadrants(Cin, Cout);
  const int TW = wgrad_x3_tw16(W) ? 16 : 32, TH = (q ? 64 : 128) / TW;
  const int ntiles = B * ceil_div(W, TW) * ceil_div(H, TH);
  const int blk = q ? 64 : 32;
  const int nx = (Cout / blk) * (Cin / blk);
  int target = 512;    // two workgroups per CU
  // tuning override, read once per process and range-checked (the slab buffer is sized from this query)
  static const int wgs_override = [] { const char* e = getenv("WTPSE_X3_WGS"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= 65536) ? v : 0; }();
  if (wgs_override) target = wgs_override;
  int ks = target / nx;
  if (ks < 1) ks = 1;
  if (ks > ntiles) ks = ntiles;
  return ks;
}

extern "C" void wtpse_wgrad_reduce_launch(const float* slab, int ksplit, int n, float* dw, int accumulate, void* stream);

// Same contract as wtpse_conv_wgrad (include/wtpse_hip.h) without the bias gradient; requires wtpse_wgrad_x3_supported().
extern "C" int wtpse_conv_wgrad_x3(const float* dy, const float* x0, int C0, const float* x1, int C1, const float* pro0,
                                   const float* pro1, int pro_relu, float* slab, int ksplit, float* dw, int accumulate, int B,
                                   int H, int W, int Cout, int ksize, void* stream) {
  WTPSE_REQUIRE(dy && x0 && slab && dw && B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0 && ksplit > 0);
  WTPSE_REQUIRE((C1 == 0) == (x1 == nullptr));
  const int Cin = C0 + C1;
  WTPSE_REQUIRE(wtpse_wgrad_x3_supported(Cin, Cout, ksize, C1 ? C0 : 8));
  const bool q = wgrad_x3_quadrants(Cin, Cout);
  WgradX3Args a;
  a.dy = dy; a.x0 = x0; a.x1 = x1; a.pro0 = pro0; a.pro1 = pro1; a.slab = slab;
  a.B = B; a.H = H; a.W = W; a.C0 = C0; a.C1 = C1; a.Cin = Cin; a.Cout = Cout; a.pro_relu = pro_relu;
  const bool narrow = wgrad_x3_tw16(W);
  const int TW = narrow ? 16 : 32, TH = (q ? 64 : 128) / TW;
  a.tiles_x = ceil_div(W, TW);
  a.tiles_y = ceil_div(H, TH);
  a.ntiles = B * a.tiles_x * a.tiles_y;
  WTPSE_REQUIRE(ksplit <= a.ntiles);
  const int blk = q ? 64 : 32;
  a.nci = Cin / blk;
  dim3 grid((unsigned)((Cout / blk) * a.nci), (unsigned)ksplit);
  hipStream_t st = (hipStream_t)stream;
  if (q) {
    if (narrow) hipLaunchKernelGGL((conv_wgrad_x3_k<3, 4, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_wgrad_x3_k<3, 5, true>), grid, dim3(256), 0, st, a);
  } else {
    if (narrow) hipLaunchKernelGGL((conv_wgrad_x3_k<3, 4, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_wgrad_x3_k<3, 5, false>), grid, dim3(256), 0, st, a);
  }
  int rc = wtpse_status();
  if (rc) return rc;
  wtpse_wgrad_reduce_launch(slab, ksplit, Cout * Cin * 9, dw, accumulate, stream);
  return wtpse_status();
}
