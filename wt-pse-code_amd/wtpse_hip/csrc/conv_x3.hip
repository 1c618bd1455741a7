// Host side of the x3 / x2h convolutions (kernels: conv_x3_kernels.h, instantiated per arithmetic in conv_x3_t1/t2/t3.hip): weight
// packing, tiling decisions, the C-ABI entry points — and the LDS-based x3 weight gradient of the 16-pixel-wide maps.
#include "conv_x3_kernels.h"

int x3_dispatch_t1(const ConvX3Args& a, const X3Launch& L, hipStream_t st);
int x3_dispatch_t2(const ConvX3Args& a, const X3Launch& L, hipStream_t st);
int x3_dispatch_t3(const ConvX3Args& a, const X3Launch& L, hipStream_t st);

// ------------------------------------------------------------------------------------------------
// Weight packing for the x3 path: OIHW fp32 -> bf16 triples (or fp16 pairs) in the kernel's LDS image order.
//   forward : rows = Cout, k = Cin, tap t          element = w[co][ci][t]
//   dgrad   : rows = Cin,  k = Cout, tap T-1-t     element = w[co][ci][T-1-t]   (the data gradient is the forward kernel on dY)
// layout: X3_WHDR shorts of header (float {1 / scale, scale, 0, 0, X3_WSLICES partial maxima of |w|}), then [k chunk of 16][row block of 32][tap][term slot 3][k half 2][row 32][8 k]
// (unsigned short); zero padded.  terms 3 / 1: bf16 terms x0, x1, x2, scale 1.  terms 2: fp16 terms h0, h1 of scale * w (third slot
// zero), scale = the power of two that brings the layer's largest |w| into [2^14, 2^15) (pack_scale_x3_k, launched in front).
// desc: n_desc x 8 ints {w_off, Cout, Cin, taps, xf_off, xd_off(-1: none), 0, 0}; x*_off in unsigned shorts.
// X3_WSLICES workgroups per layer, each the largest |w| of its slice -> header floats [4 .. 4 + X3_WSLICES) of the layer's first
// direction; pack_weights_x3_k folds them (one workgroup per layer took as long as the largest layer: 105 us for 590 k weights).
__global__ __launch_bounds__(1024) void pack_scale_x3_k(const float* __restrict__ params, const int* __restrict__ desc,
                                                        unsigned short* __restrict__ packed, int terms) {
  const int* d = desc + blockIdx.y * 8;
  const int n = d[1] * d[2] * d[3];
  const float* w = params + d[0];
  float m = 0.f;
  if (terms == 2)
    for (int e = blockIdx.x * 1024 + threadIdx.x; e < n; e += X3_WSLICES * 1024) m = fmaxf(m, fabsf(w[e]));
  __shared__ float red[16];
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    reinterpret_cast<float*>(packed + (d[4] >= 0 ? d[4] : d[5]))[4 + blockIdx.x] = m;
  }
}

__global__ __launch_bounds__(256) void pack_weights_x3_k(const float* __restrict__ params, const int* __restrict__ desc,
                                                         unsigned short* __restrict__ packed, int terms) {
  const int* d = desc + blockIdx.y * 8;
  const int w_off = d[0], Co = d[1], Ci = d[2], T = d[3];
  const float* w = params + w_off;
  for (int dir = 0; dir < 2; ++dir) {
    const int base = d[4 + dir];
    if (base < 0) continue;
    float sc = 1.f;
    if (terms == 2) {
      const float* part = reinterpret_cast<const float*>(packed + (d[4] >= 0 ? d[4] : d[5])) + 4;
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < X3_WSLICES; ++i) m = fmaxf(m, part[i]);
      sc = x3_scale_from_amax(__builtin_bit_cast(unsigned, m));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      float* hdr = reinterpret_cast<float*>(packed + base);
      hdr[0] = 1.f / sc;
      hdr[1] = sc;
      hdr[2] = hdr[3] = 0.f;
    }
    const int R = dir == 0 ? Co : Ci, K = dir == 0 ? Ci : Co;
    const int RP = (R + 31) & ~31, KP = (K + 15) & ~15;
    const int n = (KP / 8) * RP * T;               // 16-byte slots: (k half, row, tap) of every chunk and row block
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
      // e enumerates [chunk][rb][tap][half][row32]: one thread = the 8 k of one row — a 16-byte store per term slot (element-wise
      // 2-byte stores: 57 us per network and step behind every Adam step)
      const int row32 = e & 31, hh = (e >> 5) & 1;
      int q = e >> 6;
      const int t = q % T; q /= T;
      const int rb = q % (RP / 32), chunk = q / (RP / 32);
      const int row = rb * 32 + row32, k0 = chunk * 16 + hh * 8;
      float v[8];
#pragma unroll
      for (int k8 = 0; k8 < 8; ++k8) {
        const int k = k0 + k8;
        v[k8] = (row < R && k < K) ? (dir == 0 ? w[(row * Ci + k) * T + t] : w[(k * Ci + row) * T + (T - 1 - t)]) : 0.f;
      }
      u32x4v o0, o1, o2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned q0, q1, q2;
        if (terms == 2) {
          split2h_pair(v[2 * j] * sc, v[2 * j + 1] * sc, q0, q1);
          q2 = 0u;
        } else {
          split3_pair(v[2 * j], v[2 * j + 1], q0, q1, q2);
        }
        o0[j] = q0; o1[j] = q1; o2[j] = q2;
      }
      const size_t slot = ((((size_t)chunk * (RP / 32) + rb) * T + t) * 6);
      u32x4v* o = reinterpret_cast<u32x4v*>(packed + base + X3_WHDR);
      o[(slot + 0 * 2 + hh) * 32 + row32] = o0;
      o[(slot + 1 * 2 + hh) * 32 + row32] = o1;
      o[(slot + 2 * 2 + hh) * 32 + row32] = o2;
    }
  }
}

// bf16 / fp16 terms per fp32 operand in every x3 kernel of the library (this file and wgrad_r.hip): 3 = x3 (three bf16 terms, six
// products), 2 = x2h (two fp16 terms, three products, power-of-two operand scaling: the default since round 5), 1 = the bf16 mode
// (one bf16 term, one product: outside the 1e-4 parity bar).  Environment WTPSE_X3_TERMS or wtpse_x3_terms() (include/wtpse_hip.h).
// The packed weights are in the format of the value at the time they were packed: callers re-pack after a change (nn.py does).
int g_x3_terms = [] { const char* e = getenv("WTPSE_X3_TERMS"); return (e && e[0] >= '1' && e[0] <= '3') ? e[0] - '0' : 2; }();
extern "C" int wtpse_x3_terms(int terms) {
  const int was = g_x3_terms;
  if (terms >= 1 && terms <= 3) g_x3_terms = terms;
  return was;
}

extern "C" int wtpse_pack_conv_weights_x3(const float* params, const int* desc, int n_desc, unsigned short* packed, void* stream) {
  WTPSE_REQUIRE(params && desc && packed && n_desc > 0);
  hipLaunchKernelGGL(pack_scale_x3_k, dim3(X3_WSLICES, n_desc), dim3(1024), 0, (hipStream_t)stream, params, desc, packed, g_x3_terms);
  hipLaunchKernelGGL(pack_weights_x3_k, dim3(48, n_desc), dim3(256), 0, (hipStream_t)stream, params, desc, packed, g_x3_terms);
  return wtpse_status();
}

// 128-pixel tiles for the 32-channel blocks: where the 256-pixel tiling gives under two workgroups per CU on a 16-wide map (the deepest
// level) — and, round 6, on the WIDE maps when wtpse_x3_small_wide() is on (32 x 4 tiles): the launches with one or two 16-channel
// chunks (up4.conv3, up4.conv1, down1: HBM-heavy, 8192 / 2048 tiles) have no chunk loop to hide their load -> convert -> multiply ->
// store sequence behind, only the other resident workgroups — twice as many, half as long workgroups (5 per CU by LDS and registers
// instead of 4) keep more loads and stores in flight.  Changes the tiling, hence the rows of `stats`: part of wtpse_tuning_state().
static int g_x3_small_wide = [] { const char* e = getenv("WTPSE_X3_SMALL_WIDE"); return (e && e[0] >= '0' && e[0] <= '1') ? e[0] - '0' : 0; }();
extern "C" int wtpse_x3_small_wide(int on) {
  const int was = g_x3_small_wide;
  if (on >= 0) g_x3_small_wide = on ? 1 : 0;
  return was;
}
static bool x3_small_tiles(int B, int H, int W, int CoutP, bool mt2) {
  if (mt2) return false;
  if (W > 16) return g_x3_small_wide != 0 && W % 32 == 0;
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

// XCD-aware workgroup order of the x3 convolutions (ConvX3Args::xcd_tiles); WTPSE_X3_XCD=0 / wtpse_x3_xcd(0): dispatch order.
// Same workgroups, same results — a pure A/B of the order.
static int g_x3_xcd = [] { const char* e = getenv("WTPSE_X3_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
extern "C" int wtpse_x3_xcd(int on) {
  const int was = g_x3_xcd;
  if (on >= 0) g_x3_xcd = on ? 1 : 0;
  return was;
}

// The run-time switches the tiling of a launch (hence the size of its `stats` / tail buffers) and the format of the packed weights
// depend on, as one word: a launch plan (plan.hip) remembers the word it was recorded under and refuses to replay under another —
// the replayed entry points re-read the switches, and a tiling that changed since the caller sized its buffers would write out of
// bounds.
extern "C" int wtpse_tuning_state(void) { return g_x3_terms | (g_x3r << 4) | (g_x3_xcd << 8) | (g_x3_small_wide << 12); }

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
  if (ksize != 3 || g_x3_half_min == 0 || g_x3r == 0 || CoutP % 64 != 0 || x3_mt2(B, H, W, CoutP)) return false;
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
                        const unsigned* in_amax, void* stream, BnbTail tail = bnb_tail_none(), BnfTail ftail = bnf_tail_none(),
                        const unsigned* in_amax1 = nullptr, unsigned* out_amax = nullptr) {
  WTPSE_REQUIRE(in0 && wpacked && out0 && B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0);
  WTPSE_REQUIRE(!(in_amax1 && !in1) && !(out_amax && (mask_ref || bn_mean)));
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
  a.in_amax = in_amax; a.in_amax1 = in_amax1; a.in_scale = X3_FWD_SCALE; a.out_amax = out_amax;
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
  X3Launch L;
  L.ksize = ksize; L.mt2 = mt2 ? 1 : 0; L.half = x3_half(B, H, W, a.CoutP, ksize) ? 1 : 0;
  L.small = x3_small_tiles(B, H, W, a.CoutP, mt2) ? 1 : 0;
  L.epi = bnb ? 2 : mask_ref ? 1 : 0; L.x3r = g_x3r; L.xcd = g_x3_xcd;
  return g_x3_terms == 1 ? x3_dispatch_t1(a, L, st) : g_x3_terms == 2 ? x3_dispatch_t2(a, L, st) : x3_dispatch_t3(a, L, st);
}

// Same contract as wtpse_conv_fwd (include/wtpse_hip.h) with `wpacked` in the x3 layout.  Cout <= 16 runs on a 32-row
// tile with the upper rows idle (zero weights, stores dropped by the range check): the bf16 MFMAs are cheap enough.
extern "C" int wtpse_conv_fwd_x3(const float* in0, int C0, const float* in1, int C1, const unsigned short* wpacked,
                                 const float* bias, const float* pro0, const float* pro1, int pro_relu, float* out0, float* out1,
                                 int Csplit, float* stats, int B, int H, int W, int Cout, int ksize, int relu_out,
                                 const float* mask_ref, const unsigned* in_amax, const unsigned* in_amax1, unsigned* out_amax,
                                 void* stream) {
  return conv_x3_impl(in0, C0, in1, C1, wpacked, bias, pro0, pro1, pro_relu, out0, out1, Csplit, stats, B, H, W, Cout, ksize,
                      relu_out, mask_ref, nullptr, nullptr, 0, 0, 0, in_amax, stream, bnb_tail_none(), bnf_tail_none(), in_amax1, out_amax);
}

// Data gradient that also performs the first half of the BatchNorm backward of the layer it flows into (include/wtpse_hip.h).
extern "C" int wtpse_dgrad_x3_bnb(const float* dy, int C, const unsigned short* wpacked, float* out0, float* out1, int Csplit,
                                  const float* bn_y, const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0, int bn_c1,
                                  float* stats, int B, int H, int W, int Cout, int ksize, const unsigned* in_amax, void* stream) {
  WTPSE_REQUIRE(bn_y && bn_ss && bn_mean && stats);
  return conv_x3_impl(dy, C, nullptr, 0, wpacked, nullptr, nullptr, nullptr, 0, out0, out1, Csplit, stats, B, H, W, Cout, ksize, 0,
                      bn_y, bn_ss, bn_mean, bn_relu, bn_c0, bn_c1, in_amax, stream);
}

// wtpse_conv_fwd_bnf (conv.hip), x3 layout
extern "C" int wtpse_conv_fwd_x3_ftail(const float* in0, int C0, const float* in1, int C1, const unsigned short* wpacked,
                                       const float* bias, const float* pro0, const float* pro1, int pro_relu, float* out0,
                                       float* stats, const BnfTail* ftail, int B, int H, int W, int Cout, int ksize,
                                       const unsigned* in_amax0, const unsigned* in_amax1, void* stream) {
  WTPSE_REQUIRE(ftail && stats);
  return conv_x3_impl(in0, C0, in1, C1, wpacked, bias, pro0, pro1, pro_relu, out0, nullptr, Cout, stats, B, H, W, Cout, ksize, 0,
                      nullptr, nullptr, nullptr, 0, 0, 0, in_amax0, stream, bnb_tail_none(), *ftail, in_amax1);
}

// wtpse_dgrad_bnb_coef (conv.hip), x3 layout
extern "C" int wtpse_dgrad_x3_bnb_tail(const float* dy, int C, const unsigned short* wpacked, float* out0, float* out1, int Csplit,
                                       const float* bn_y, const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0,
                                       int bn_c1, float* stats, const BnbTail* tail, int B, int H, int W, int Cout, int ksize,
                                       const unsigned* in_amax, void* stream) {
  WTPSE_REQUIRE(bn_y && bn_ss && bn_mean && stats && tail);
  return conv_x3_impl(dy, C, nullptr, 0, wpacked, nullptr, nullptr, nullptr, 0, out0, out1, Csplit, stats, B, H, W, Cout, ksize, 0,
                      bn_y, bn_ss, bn_mean, bn_relu, bn_c0, bn_c1, in_amax, stream, *tail);
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

#define WSPLIT(a, b, q0, q1, q2) split3_pair(a, b, q0, q1, q2)
#define WSTORE(dst, v) dst = v
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
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = buf_load(rs, vo, (unsigned)min(cb + j, cn) * (unsigned)HW * 4u);
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
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = buf_load(rsy, vo, (unsigned)min(c + j, a.Cout) * (unsigned)HW * 4u);
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
