// Implicit-GEMM convolution (3x3 pad 1 / 1x1) on the fp32 matrix cores of gfx950, NCHW fp32.
//
// Replaces the stock nn.Conv2d dispatches of the reference hot path
// (algorithms.py:882-888,926-933,404-424,991,1006-1012,1123,1199-1201 and their duplicates in
// shape_networks.py) — forward, data gradient and weight gradient.
//
// GEMM orientation is chosen for NCHW: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = (cin, tap).
// The MFMA result then has pixels on lanes and output channels in registers, so every store
// instruction writes runs of contiguous pixels of one channel plane.
//   * Cout <= 16           : v_mfma_f32_16x16x4_f32, wave tile 16 cout x 64 pixels   (MODE 0)
//   * Cout multiple of 32  : v_mfma_f32_32x32x2_f32, wave tile 32|64 cout x 64 pixels (MODE 1|2)
//   * Cout, Cin <= 16, 3x3 : MODE 3 (round 3) — the same tile and epilogue as MODE 0, the products in the x3 arithmetic of
//                            conv_x3.hip on v_mfma_f32_16x16x32_bf16: the whole 16-channel input tile is split once into bf16
//                            triples in LDS, the 9 taps x 16 channels are five k = 32 steps (tap pairs), and the layer's 15
//                            weight fragments live in registers for the whole workgroup.  2.4x fewer matrix cycles and 5x
//                            fewer LDS reads than MODE 0: these layers (inc, DeepWT, the teacher's inc) are then bound by HBM.
// A workgroup (4 waves) owns a TH x TW = 256-pixel spatial tile of one image and one cout block;
// input channels are streamed through LDS in chunks (halo tile [KC][TH+2][TW+2] + weight slab
// [KC*taps][CB]).  Several workgroups per CU overlap one another's load and MFMA phases.
//
// Fusions available in the loader/epilogue (all optional):
//   loader   : virtual channel concat of two inputs (torch.cat at algorithms.py:955,1018 never materialises),
//              per-channel affine (BatchNorm apply) + ReLU on the way into LDS
//   epilogue : + bias, ReLU, per-channel (sum, sum of squares) partials for train-mode BatchNorm statistics,
//              channel split of the result into two tensors (data gradient of a concat)
// The data gradient is the same kernel run on dY with tap-flipped, transposed weights (see pack kernel).
#include "common.h"

// Input channels per LDS chunk on the 16-cout path.  These layers (16 -> 16 at full resolution, the 1x1 heads) are bound by
// load latency, not by the matrix pipe: small chunks (4 channels for 3x3, 8 for 1x1 instead of 16) cut LDS and staging
// registers so that more workgroups are resident per CU: inc.conv2 149 -> 122 us forward, 127 -> 105 us data gradient.
#define WTPSE_P16_KC(KS) ((KS) == 3 ? 4 : 8)

struct ConvArgs {
  const float* in0;
  const float* in1;
  const float* wp;     // packed weights [CinP][taps][CoutP]; MODE 3: x3 fragments [5 k-steps][3 terms][64 lanes][8 bf16]
  const float* bias;   // [Cout] or null
  const float* pro0;   // [C0][2] (scale, shift) applied to in0 on load, or null
  const float* pro1;   // [C1][2] for in1, or null
  float* out0;
  float* out1;
  float* stats;        // [gridDim.x][Cout][2] or null
  const float* mask;   // [B][Cout][H][W] or null: out = mask > 0 ? value : 0 (ReLU backward fused into a data gradient)
  const unsigned* in_amax;   // MODE 4 (x2h): the amax table of in0 as loaded (a gradient's amax, a forward activation's bound: common.h), or null: in_scale
  float in_scale;
  unsigned* out_amax;        // EPI 0: the amax table (zero on entry) of the stored output, or null (conv_x3_kernels.h: ConvX3Args::out_amax)
  float* gram;         // [gridDim.x][16][16] or null (16-cout path): the tile's partial Gram  sum_px out[i][px] * out[j][px]
  // EPI == 2 (BatchNorm backward statistics in a data gradient's epilogue, see conv_x3.hip): output channels [bn_c0, bn_c1) are
  // masked with the ReLU of the conv + BatchNorm layer they flow into (raw conv output: `mask`, [B][bn_c1 - bn_c0][H][W]) and
  // (sum g, sum g * (y - mean)) partials go to `stats`
  const float* bn_ss;
  const float* bn_mean;
  int bn_c0, bn_c1, bn_relu;
  BnbTail tail;        // EPI == 2: the BatchNorm-backward coefficients from the last workgroups (common.h), or tickets == null
  BnfTail ftail;       // forward statistics: BatchNorm finalize by the last workgroups (common.h), or tickets == null
  int B, H, W;
  int C0, C1, Cin, CinP;
  int Cout, CoutP, Csplit;
  int pro_relu;        // bit0: ReLU on in0 after the affine, bit1: on in1
  int relu_out;
  int tiles_x, tiles_y;
  int xcd_tiles;       // > 0: tiles per XCD — workgroup b works on tile (b % 8) * xcd_tiles + b / 8 (see conv_fwd_k); 0: tile = b
};


template <bool P16> struct AccT { typedef f32x16 type; };
template <> struct AccT<true> { typedef f32x4 type; };
__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return mfma16(a, b, c); }
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return mfma32(a, b, c); }

template <int PE>
struct PlaneStride {  // smallest S >= PE with S % 32 == 16: the four k-planes a 16x16x4 B-read touches land on disjoint banks
  static constexpr int value = ((PE - 16 + 31) / 32) * 32 + 16;
};

// __launch_bounds__(256, 3): at least 3 workgroups per CU (<= 168 registers per lane, accumulators included)
// MASK is a template parameter, not a run-time test of a.mask: s_waitcnt operands are static, so the waits the mask
// loads need would also be executed (and drain earlier stores) by launches without a mask.
template <int KS, int MODE, int TWL, bool DB, int EPI>
__global__ __launch_bounds__(256, 3) void conv_fwd_k(ConvArgs a) {
  constexpr bool MASK = EPI == 1, BNB = EPI == 2;
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  constexpr int TW = 1 << TWL, TH = 256 / TW;
  constexpr int PITCH = TW + 2 * PAD, ROWS = TH + 2 * PAD;
  constexpr int PE = PITCH * ROWS;
  // every thread stores all of its NPOS tile positions, valid or not (positions past PE land in the plane's padding):
  // the loader has no per-element branch, which matters because beside another wave's MFMA stream each instruction of
  // this phase costs about one MFMA slot (phase stamps of round 1: profiles/r01_conv_phase_stamps.txt)
  constexpr int NPOS = (PE + 255) / 256;
  constexpr int S = PlaneStride<NPOS * 256>::value;
  constexpr bool XP = (MODE == 3 || MODE == 4);    // 16-cout path in the x3 (MODE 3) / x2h (MODE 4) arithmetic (3x3, Cin <= 16)
  constexpr int XT = MODE == 4 ? 2 : 3;            // 16-bit terms per fp32 operand on that path
  constexpr bool P16 = (MODE == 0) || XP;
  static_assert(!XP || KS == 3, "MODE 3 / 4 is the 3x3 path");
  constexpr int MT = P16 ? 1 : MODE;
  constexpr int MB = P16 ? 16 : 32;
  constexpr int CB = MB * MT;
  constexpr int NT = P16 ? 4 : 2;
  constexpr int NB = P16 ? 16 : 32;
  constexpr int KC = P16 ? WTPSE_P16_KC(KS) : 8;
  constexpr int KQ = P16 ? 4 : 2;
  constexpr int NACC = P16 ? 4 : 16;
  constexpr int CB4 = CB / 4;
  constexpr int NW = (KC * TAPS * CB4 + 255) / 256;   // 16-byte weight loads per thread and chunk
  constexpr int PEP = (PE + 7) & ~7;                     // MODE 3: 16-byte slots per (term, k-half) plane of the split tile
  constexpr int XS_SZ = XP ? 2 * XT * PEP * 4 : KC * S;
  constexpr int WS_SZ = XP ? 0 : NW * 1024;            // weight slab [KC*TAPS][CB], padded to whole load rounds
  constexpr int GRAM_SZ = (P16 && KS == 3) ? 4 * 16 * 65 + 4 * 256 : 0;   // wave-private [16 ch][64 px (+1)] tiles + 4 partial Grams
  constexpr int RED_SZ = 4 * CB * 2 > GRAM_SZ ? 4 * CB * 2 : GRAM_SZ;
  constexpr int MAIN_SZ = (XS_SZ + WS_SZ) > RED_SZ ? (XS_SZ + WS_SZ) : RED_SZ;
  __shared__ __attribute__((aligned(16))) float smem[MAIN_SZ + CB + (BNB ? 4 * CB : 0)];
  float* bias_s = smem + MAIN_SZ;   // this block's biases (written here, visible after the first barrier of the chunk loop)
  const float* bnp_s = bias_s + CB; // EPI 2: [3][CB] (scale | shift | mean) of the BatchNorm'd output channels, (0, 1, 0) elsewhere
  float* Xs = smem;
  float* Ws = smem + XS_SZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one: MI355X_MICROARCH.md) and every XCD has its own L2, so
  // with tile = b the eight neighbours of a tile sit on eight different L2s and each fetches the shared halo (and the partial lines
  // at its tile's edges) through the fabric again: the HBM-bound 16-channel kernel read 2.75 x its input (rocprofv3 FETCH_SIZE,
  // calibrated on a copy: profiles/r04_*).  With xcd_tiles set, the workgroups of one XCD walk a contiguous range of tiles — the ~128
  // that are resident together are a block of neighbouring tiles whose halos meet in that XCD's L2.  Placement only: any mapping is correct.
  // (Launches with several output-channel blocks: a tile's blocks in consecutive slots of the same XCD, as in conv_x3.hip.)
  int tile = (int)blockIdx.x, cblk = (int)blockIdx.y;
  if (a.xcd_tiles > 0) {
    const int L = (int)(blockIdx.y * gridDim.x + blockIdx.x), sl = L >> 3;
    cblk = sl % (int)gridDim.y;
    tile = (L & 7) * a.xcd_tiles + sl / (int)gridDim.y;
  }
  int bx = tile;
  const int tx = bx % a.tiles_x;
  bx /= a.tiles_x;
  const int ty = bx % a.tiles_y;
  const int b = bx / a.tiles_y;
  const int cout0 = cblk * CB;
  const int HW = a.H * a.W;
  if (tid < CB) bias_s[tid] = (a.bias && cout0 + tid < a.Cout) ? a.bias[cout0 + tid] : 0.f;
  if (BNB && tid < CB) {
    const int c = cout0 + tid;
    const bool bn = c >= a.bn_c0 && c < a.bn_c1;
    float* q = bias_s + CB + tid;          // three planes [scale | shift | mean] of CB floats
    q[0] = (bn && a.bn_relu) ? a.bn_ss[2 * (c - a.bn_c0)] : 0.f;
    q[CB] = (bn && a.bn_relu) ? a.bn_ss[2 * (c - a.bn_c0) + 1] : 1.f;
    q[2 * CB] = bn ? a.bn_mean[c - a.bn_c0] : 0.f;
  }

  int off[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int p = wave * 64 + nt * NB + (lane & (NB - 1));
    off[nt] = (p >> TWL) * PITCH + (p & (TW - 1));
  }

  typename AccT<P16>::type acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < NACC; ++r) acc[mt][nt][r] = 0.f;

  if constexpr (XP) {
    // ---- MODE 3 / 4: one 16-channel chunk, split once; weights in registers; 16x16x32 MFMAs: six bf16 products per fp32 multiply
    // (x3) or three fp16 products (x2h: operands scaled by powers of two on the way in, the accumulators back on the way out —
    // conv_x3_kernels.h has the full description)
    u32x4* Xq = reinterpret_cast<u32x4*>(smem);           // [term XT][k-half 2][PEP positions] 16-byte rows of 8 channels
    const int g4 = lane >> 4;
    // loader work items: (halo position, k-half) in whole-wave blocks (as conv_x3.hip)
    constexpr int PB = (PE + 63) / 64, NIT = (2 * PB + 3) / 4;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0 + (size_t)b * a.C0 * HW, (unsigned)a.C0 * HW * 4u);
    const bool any_pro = MODE == 4 || a.pro0 != nullptr || a.pro_relu != 0;      // x2h: the input scale rides in the coefficients
    // (x2h) the input's amax table: read issued in front of the tile loads, picked up behind them (common.h, amax_load / amax_reduce)
    const unsigned sx_raw = MODE == 4 ? amax_load(a.in_amax) : 0u;
    __builtin_amdgcn_sched_barrier(0);
    float xv[NIT][8];
    int ipos[NIT], ihalf[NIT];
    bool iin[NIT];
    // prologue coefficients of the two channel halves: fetched (scalar loads, the half of an item is wave-uniform) in front of the
    // tile loads, so that their latency is not paid where they are used (conv_x3.hip: the same pattern cost the forward kernel 9 %;
    // here: 116 -> 106 us with prologue + statistics).  Tried on this path and dropped, no gain either way: 16-byte tile loads (a
    // thread takes four columns of eight channels: 94 vs 88-94 us) and one MFMA per pixel tile and cross term instead of six
    // dependent ones per tile (98 vs 94) — the workgroup's phases (load, convert, multiply, epilogue) add up per SIMD.
    float psc[2][8], psh[2][8];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int cg = min(hh * 8 + j, a.C0 - 1);
        psc[hh][j] = a.pro0 ? a.pro0[2 * cg] : 1.f;
        psh[hh][j] = a.pro0 ? a.pro0[2 * cg + 1] : 0.f;
      }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int blk = __builtin_amdgcn_readfirstlane(i * 4 + wave);
      ihalf[i] = blk >= PB ? 1 : 0;
      const int p = (blk - ihalf[i] * PB) * 64 + lane;
      ipos[i] = (blk < 2 * PB && p < PE) ? p : -1;
      const int r = p / PITCH, x = p - r * PITCH;
      const int gy = ty * TH + r - PAD, gx = tx * TW + x - PAD;
      iin[i] = ipos[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const unsigned vo = iin[i] ? (unsigned)(gy * a.W + gx) * 4u : BUF_OOB;
#pragma unroll
      for (int j = 0; j < 8; ++j)     // channels past C0 are out of the buffer's range and read as zero (their weights are zero too)
        xv[i][j] = buf_load(rs0, vo, (unsigned)min(ihalf[i] * 8 + j, a.C0) * (unsigned)HW * 4u);
    }
    // the layer's weight fragments: a 16-byte header (float {1 / scale, scale}: the x2h weight scale), [k-step 5][term 3] bf16 triples,
    // then [k-step 5][term 2] fp16 pairs of scale * w — one 16-byte row per lane (pre-split by pack_weights_x3p16_k)
    const u32x4* wq = reinterpret_cast<const u32x4*>(a.wp) + 1 + (MODE == 4 ? 15 * 64 : 0);
    u32x4 afr[5][XT];
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5)
#pragma unroll
      for (int t = 0; t < XT; ++t) afr[s5][t] = wq[(s5 * XT + t) * 64 + lane];
    // the x2h input scale: a dependent read of the input's amax table (a gradient's, or since round 6 a forward activation's bound) —
    // behind the tile loads, folded into the prologue coefficients (exact: a power of two)
    const float sx = MODE == 4 ? (a.in_amax ? x3_scale_from_amax(amax_reduce(sx_raw)) : a.in_scale) : 1.f;
    if constexpr (MODE == 4) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          psc[hh][j] *= sx;
          psh[hh][j] *= sx;
        }
    }
    if (any_pro) {   // zero padding applies AFTER the fused affine/ReLU, as in the reference graph
      const float lo = (a.pro_relu & 1) ? 0.f : -INFINITY;     // ReLU = one v_max against a uniform floor (conv_x3_kernels.h: convert_pair)
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int nc = iin[i] ? a.C0 - ihalf[i] * 8 : 0;     // channels of this item that exist (one compare per element below)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float sc = ihalf[i] ? psc[1][j] : psc[0][j], sh = ihalf[i] ? psh[1][j] : psh[0][j];
          const float v = fmaxf(fmaf(xv[i][j], sc, sh), lo);
          xv[i][j] = j < nc ? v : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      if (ipos[i] >= 0) {
        u32x4 tt[XT];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned q0, q1, q2 = 0u;
          if constexpr (XT == 3) wt_split3_pair(xv[i][2 * j], xv[i][2 * j + 1], q0, q1, q2);
          else split2h_pair(xv[i][2 * j], xv[i][2 * j + 1], q0, q1);
          tt[0][j] = q0; tt[1][j] = q1;
          if constexpr (XT == 3) tt[2][j] = q2;
        }
#pragma unroll
        for (int t = 0; t < XT; ++t) Xq[(t * 2 + ihalf[i]) * PEP + ipos[i]] = tt[t];
      }
    }
    __syncthreads();
    // k-step s5 covers taps 2 s5 (lane groups 0, 1) and 2 s5 + 1 (groups 2, 3); the tenth "tap" has zero weights
    int toff[5];
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
      const int t = min(2 * s5 + (g4 >> 1), TAPS - 1);
      toff[s5] = (t / KS) * PITCH + (t % KS);
    }
    const int hsel = (g4 & 1) * PEP;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) {
        u32x4 bfr[XT];
#pragma unroll
        for (int t = 0; t < XT; ++t) bfr[t] = Xq[(t * 2) * PEP + hsel + off[nt] + toff[s5]];
        f32x4 c = acc[0][nt];
        if constexpr (XT == 3) {
          // the six leading cross terms, smallest first (as conv_x3.hip)
          c = wt_mfma16x32(afr[s5][0], bfr[2], c);
          c = wt_mfma16x32(afr[s5][1], bfr[1], c);
          c = wt_mfma16x32(afr[s5][2], bfr[0], c);
          c = wt_mfma16x32(afr[s5][0], bfr[1], c);
          c = wt_mfma16x32(afr[s5][1], bfr[0], c);
          c = wt_mfma16x32(afr[s5][0], bfr[0], c);
        } else {
          c = wt_mfma16x32h(afr[s5][0], bfr[1], c);
          c = wt_mfma16x32h(afr[s5][1], bfr[0], c);
          c = wt_mfma16x32h(afr[s5][0], bfr[0], c);
        }
        acc[0][nt] = c;
      }
    }
    if constexpr (MODE == 4) {      // back to the operands' own scale (exact: powers of two) before bias / statistics / stores
      const float inv = a.wp[0] / sx;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[0][nt] *= inv;
    }
    __syncthreads();      // the epilogue reuses the tile's LDS (Gram staging, statistics)
  } else {
  // halo-tile positions owned by this thread (fixed for the whole kernel): position tid + 256*i of a channel plane;
  // gpos = offset within a global channel plane (-1: zero padding / outside the image / past the tile)
  int gpos[NPOS];
#pragma unroll
  for (int i = 0; i < NPOS; ++i) {
    int p = tid + 256 * i;
    int r = p / PITCH, x = p - r * PITCH;
    int gy = ty * TH + r - PAD, gx = tx * TW + x - PAD;
    gpos[i] = (p < PE && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? gy * a.W + gx : -1;
  }

  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.in0 + (size_t)b * a.C0 * HW, (unsigned)a.C0 * HW * 4u);
  const __amdgpu_buffer_rsrc_t rs1 = a.in1 ? make_rsrc(a.in1 + (size_t)b * a.C1 * HW, (unsigned)a.C1 * HW * 4u) : rs0;
  unsigned voff[NPOS];
#pragma unroll
  for (int i = 0; i < NPOS; ++i) voff[i] = gpos[i] >= 0 ? (unsigned)gpos[i] * 4u : BUF_OOB;
  // packed weights [CinP*TAPS][CoutP]: slab row e4 / CB4, columns cout0 + 4*(e4 % CB4); rows past the last channel
  // (ragged last chunk) and columns past CoutP are out of range and read as zero
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(a.wp, (unsigned)a.CinP * TAPS * (unsigned)a.CoutP * 4u);
  unsigned woff[NW];
#pragma unroll
  for (int it = 0; it < NW; ++it) {
    const int e4 = tid + 256 * it;
    const int row = e4 / CB4, j4 = e4 - row * CB4;
    woff[it] = (row < KC * TAPS && cout0 + j4 * 4 < a.CoutP) ? (unsigned)(row * a.CoutP + cout0 + j4 * 4) * 4u : BUF_OOB;
  }
  const bool any_pro = a.pro0 != nullptr || a.pro1 != nullptr || a.pro_relu != 0;
  // Every global load of a chunk is issued before its first use (KC*NPOS + NW independent loads in flight per lane).
  // DB (register double-buffering, chosen by the host for grids too small to fill the chip with several workgroups
  // per CU): the loads of chunk k+1 are issued right after chunk k has been stashed in LDS and are in flight during
  // chunk k's MFMAs.  With >= 3 resident workgroups per CU other workgroups already cover that latency (measured: +0 %).
  float xv[KC][NPOS];
  f32x4 wv[NW];
  auto issue_loads = [&](int c0) {
    // a chunk never straddles the two inputs (C0 % 16 == 0 is checked on the host); channels past Cin re-read a
    // valid plane (their packed weight rows are zero), so the unrolled load block has no per-channel conditionals
    const bool first = c0 < a.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int cbase = first ? c0 : c0 - a.C0;
    const int cmax = (first ? a.C0 : a.C1) - 1;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const unsigned soff = (unsigned)min(cbase + c, cmax) * (unsigned)HW * 4u;
#pragma unroll
      for (int i = 0; i < NPOS; ++i) xv[c][i] = buf_load(rs, voff[i], soff);
    }
#pragma unroll
    for (int it = 0; it < NW; ++it) wv[it] = buf_load4(rsw, woff[it], (unsigned)(c0 * TAPS) * (unsigned)a.CoutP * 4u);
  };
  auto stash = [&](int c0) {
    const bool first = c0 < a.C0;
    // zero padding applies AFTER the fused affine/ReLU, as in the reference graph
    if (any_pro) {
      const bool relu = first ? (a.pro_relu & 1) : (a.pro_relu & 2);
      const float* pro = first ? a.pro0 : a.pro1;
      const int cbase = first ? c0 : c0 - a.C0;
      const int cmax = (first ? a.C0 : a.C1) - 1;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int cg = min(cbase + c, cmax);
        const float sc = pro ? pro[2 * cg] : 1.f, sh = pro ? pro[2 * cg + 1] : 0.f;
#pragma unroll
        for (int i = 0; i < NPOS; ++i) {
          float v = fmaf(xv[c][i], sc, sh);
          if (relu) v = fmaxf(v, 0.f);
          xv[c][i] = gpos[i] >= 0 ? v : 0.f;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int i = 0; i < NPOS; ++i) Xs[c * S + tid + 256 * i] = xv[c][i];
#pragma unroll
    for (int it = 0; it < NW; ++it) *reinterpret_cast<f32x4*>(Ws + (tid + 256 * it) * 4) = wv[it];
  };
  if (DB) issue_loads(0);
  for (int c0 = 0; c0 < a.CinP; c0 += KC) {
    const int kc = min(KC, a.CinP - c0);
    __syncthreads();   // the previous chunk's MFMAs are done with LDS
    if (!DB) issue_loads(c0);
    stash(c0);
    __syncthreads();
    if (DB && c0 + KC < a.CinP) issue_loads(c0 + KC);
    // ---- MFMA
    const int nq = kc / KQ;
    for (int q = 0; q < nq; ++q) {
      const int cl = q * KQ + (lane / NB);
      const float* xrow = Xs + cl * S;
      const float* wrow = Ws + cl * TAPS * CB + (lane & (MB - 1));
      float av[TAPS][MT], bv[TAPS][NT];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int toff = (t / KS) * PITCH + (t % KS);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[t][mt] = wrow[t * CB + mt * MB];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[t][nt] = xrow[off[nt] + toff];
      }
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma(av[t][mt], bv[t][nt], acc[mt][nt]);
    }
  }
  }
  if (a.bias) {
    float bz[MT][NACC];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NACC; ++r)
        bz[mt][r] = bias_s[P16 ? ((lane >> 4) * 4 + r) : (mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NACC; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] += bz[mt][r];
  }

  // ---- epilogue: bias, ReLU, store (pixels on lanes -> contiguous runs per channel plane), BN partial statistics
  const bool want_stats = a.stats != nullptr;
  if (want_stats) __syncthreads();  // everyone is done with Xs/Ws before they are reused for the reduction
  float* red = smem;                // [4 waves][CB][2]
  const int C1out = a.Cout - a.Csplit;
  // pixel offsets / validity of this lane's NT pixels (same for every channel)
  int poff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int p = wave * 64 + nt * NB + (lane & (NB - 1));
    int gy = ty * TH + (p >> TWL), gx = tx * TW + (p & (TW - 1));
    poff[nt] = (gy < a.H && gx < a.W) ? gy * a.W + gx : -1;
  }
  // vmcnt counts loads and stores in one in-order queue on gfx9 and s_waitcnt operands are static.  A load issued
  // after a store cannot be waited on without waiting for that store's write acknowledgement (1-2 us), and around a
  // store inside an exec-mask branch the compiler must assume the store was skipped, so every wait behind it drains
  // the queue (measured with phase stamps, profiles/r01_conv_phase_stamps.txt: the epilogue took 60k cycles, 21 % of a workgroup's life).
  // Hence: the biases come from LDS (added after the sum, as the reference does), the ReLU-mask loads of an M block are issued before that
  // block's first store, and loads/stores are branch-free buffer operations: the lane's pixel (and the +4 channels of
  // the upper half-wave) sit in a per-lane voffset computed once, the register's first channel in the scalar soffset,
  // and lanes outside the image (BUF_OOB) or channels past the end of the tensor are dropped by the range check
  // (voffset >= num_records - soffset).
  const __amdgpu_buffer_rsrc_t rs_o0 = make_rsrc(a.out0 + (size_t)b * a.Csplit * HW, (unsigned)a.Csplit * HW * 4u);
  const __amdgpu_buffer_rsrc_t rs_o1 = a.out1 ? make_rsrc(a.out1 + (size_t)b * C1out * HW, (unsigned)C1out * HW * 4u) : rs_o0;
  const int Cbn = a.bn_c1 - a.bn_c0;
  const __amdgpu_buffer_rsrc_t rs_m = MASK ? make_rsrc(a.mask + (size_t)b * a.Cout * HW, (unsigned)a.Cout * HW * 4u)
                                      : BNB ? make_rsrc(a.mask + (size_t)b * Cbn * HW, (unsigned)Cbn * HW * 4u) : rs_o0;
  const int clane = P16 ? (lane >> 4) * 4 : (lane >> 5) * 4;   // channel offset of this lane within a register's group
  unsigned pvo[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) pvo[nt] = poff[nt] >= 0 ? (unsigned)(clane * HW + poff[nt]) * 4u : BUF_OOB;
  constexpr int NSV = NACC * 2;
  // Beside other waves' MFMA streams every instruction here costs about one MFMA slot, so the epilogue proper is
  // minimal: per value a max, a store (and three VALU for the statistics); per register a few scalar operations for
  // soffset and the choice of output tensor.  Ragged workgroups (tile not inside the image, cout block not inside the
  // tensor) first zero the accumulators of their invalid (pixel, channel) pairs: the stores of those lanes are dropped
  // by the range check anyway, and zeros do not disturb the statistics.
  const bool full = ty * TH + TH <= a.H && tx * TW + TW <= a.W && cout0 + CB <= a.Cout;
  if (!full) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const bool cvalid = cout0 + (P16 ? r : (mt * 32 + (r & 3) + 8 * (r >> 2))) + clane < a.Cout;
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
  if constexpr (P16 && KS == 3) {
    // Gram of the output tile in the epilogue (the WT loss's G = z z^T, reference algorithms.py:1283): the DeepWT convs that
    // produce z1 / z2 hand the loss their per-tile partial Grams, so that compute_whitening_loss never reads z from HBM
    // again (134 MB per map at B=32, 256x256).  The accumulators hold [channel in registers][pixel on lanes]; the 16x16x4
    // MFMA wants [channel on lanes][4 pixels across lane groups] — one trip through a wave-private LDS tile — and then takes
    // the same register as A and as B (as gram_partial_k does).
    if (a.gram) {
      __syncthreads();   // every wave is done with Xs / Ws
      float* zs = smem + wave * (16 * 65);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < NACC; ++r) zs[((lane >> 4) * 4 + r) * 65 + nt * 16 + (lane & 15)] = acc[0][nt][r];
      __builtin_amdgcn_wave_barrier();
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const float v = zs[(lane & 15) * 65 + 4 * st + (lane >> 4)];
        g = mfma16(v, v, g);
      }
      float* gs = smem + 4 * 16 * 65 + wave * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) gs[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = g[r];
      __syncthreads();
      const float* g0 = smem + 4 * 16 * 65;
      a.gram[(size_t)tile * 256 + tid] = g0[tid] + g0[256 + tid] + g0[512 + tid] + g0[768 + tid];
      __syncthreads();   // before the statistics (if any) reuse smem
    }
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
      amax_publish_wave(a.out_amax, am, (unsigned)tile * 4u + (unsigned)wave);
    }
  }
  const unsigned hw4 = (unsigned)HW * 4u;
  // a launch that folds its own statistics publishes them and takes its tickets BEFORE it stores its output tile (see conv_x3.hip:
  // the hand-off drains the workgroup's outstanding stores); the values to store stay in the accumulators
  const bool defer = want_stats && (BNB ? a.tail.tickets != nullptr : a.ftail.tickets != nullptr);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float mk[NACC][NT];
    if (MASK) {
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const int cbase = cout0 + (P16 ? r : (mt * 32 + (r & 3) + 8 * (r >> 2)));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mk[r][nt] = buf_load(rs_m, pvo[nt], (unsigned)min(cbase, a.Cout) * hw4);
      }
    }
    if (BNB) {
      // whether a register belongs to the BatchNorm'd tensor is wave-uniform (bn_c0 / bn_c1 multiples of 16); the others load
      // out of range (0)
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const int cbase = cout0 + (P16 ? r : (mt * 32 + (r & 3) + 8 * (r >> 2)));
        const bool bn = cbase >= a.bn_c0 && cbase < a.bn_c1;
        const unsigned soff = (unsigned)(bn ? cbase - a.bn_c0 : 0) * hw4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mk[r][nt] = buf_load(rs_m, bn ? pvo[nt] : BUF_OOB, soff);
      }
    }
    float bmu[NACC];
#pragma unroll
    for (int r = 0; r < NACC; ++r) {
      // the channels one register holds across the wave lie in one aligned group of 8 (16 on the 16-wide path) and
      // Csplit is a multiple of 16, so the choice of output tensor is wave-uniform
      const int cbase = cout0 + (P16 ? r : (mt * 32 + (r & 3) + 8 * (r >> 2)));
      const bool second = a.out1 != nullptr && cbase >= a.Csplit;
      const __amdgpu_buffer_rsrc_t rs_o = second ? rs_o1 : rs_o0;
      // min(): soffset stays <= num_records for the zero-padded channels past Cout, so num_records - soffset cannot wrap
      const unsigned soff = (unsigned)(second ? min(cbase, a.Cout) - a.Csplit : min(cbase, a.Csplit)) * hw4;
      float bsc = 0.f, bsh = 1.f;
      bmu[r] = 0.f;
      if (BNB) {
        const int crel = (P16 ? r : (mt * 32 + (r & 3) + 8 * (r >> 2))) + clane;
        bsc = bnp_s[crel];
        bsh = bnp_s[CB + crel];
        bmu[r] = bnp_s[2 * CB + crel];
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float v = acc[mt][nt][r];
        if (MASK && !(mk[r][nt] > 0.f)) v = 0.f;
        if (BNB) {     // the ReLU decision of the forward pass: fmaf(y, scale, shift) > 0
          float zz = __builtin_fmaf(mk[r][nt], bsc, bsh);
          asm volatile("" : "+v"(zz));     // no v_pk_fma_f32 here: see x3_epilogue (conv_x3.hip)
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
#pragma clang fp contract(off)   // square, then add: the partials must not depend on which pairs the compiler fuses
          if (BNB) {
            const float v = acc[mt][nt][r];
            s1 += v;
            s2 += v * (mk[r][nt] - bmu[r]);
          } else {
            const float v = acc[mt][nt][r];   // forward statistics are never combined with a ReLU mask (host check)
            s1 += v;
            s2 += v * v;
          }
        }
        sv[r * 2 + 0] = s1;
        sv[r * 2 + 1] = s2;
      }
      // Butterfly "transpose" reduction of the per-lane statistics sv[r*2 + k] (k = 0: sum, 1: sum of squares over this
      // lane's NT pixels) over the NB lanes that hold one channel's pixels: at step s a lane hands its partner
      // (lane ^ 2^s) the half of the values the partner will own and adds the half it receives, so the value count
      // halves each step: NSV - 1 shuffles instead of NSV * log2(NB).  Afterwards the lane's one remaining value is the
      // total of index  sum_s bit_s(lane) * (NSV >> (s+1)).  (Kept inline: through a lambda taking sv by reference the
      // compiler turned the two-way selects into dynamic indexing of sv, thousands of compare/select pairs.)
      constexpr int LB = P16 ? 4 : 5;               // lane bits spanned by one channel's pixels
      constexpr int HB = P16 ? 3 : 5;               // halving steps = log2(NSV): NSV = 8 | 32
#pragma unroll
      for (int st = 0; st < HB; ++st) {
        const int half = NSV >> (st + 1);
        const bool up = (lane >> st) & 1;
#pragma unroll
        for (int i = 0; i < NSV / 2; ++i) {
          if (i < half) {
            float keep = up ? sv[i + half] : sv[i];
            float send = up ? sv[i] : sv[i + half];
            sv[i] = keep + __shfl_xor(send, 1 << st, 64);
          }
        }
      }
#pragma unroll
      for (int st = HB; st < LB; ++st) sv[0] += __shfl_xor(sv[0], 1 << st, 64);   // leftover lane bit: plain sum
      int idx = 0;
#pragma unroll
      for (int st = 0; st < HB; ++st) idx += ((lane >> st) & 1) * (NSV >> (st + 1));
      const int k = idx & 1, rr = idx >> 1;
      const int crel = P16 ? ((lane >> 4) * 4 + rr) : (mt * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5));
      if (((lane & (NB - 1)) >> HB) == 0) red[(wave * CB + crel) * 2 + k] = sv[0];
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < CB * 2) {
      int crel = tid >> 1;
      const float s = red[tid] + red[CB * 2 + tid] + red[2 * CB * 2 + tid] + red[3 * CB * 2 + tid];
      if (BNB) {
        const int c = cout0 + crel;
        if (c >= a.bn_c0 && c < a.bn_c1) pub_store(a.stats + ((size_t)tile * Cbn + c - a.bn_c0) * 2 + (tid & 1), s);
      } else if (cout0 + crel < a.Cout) {
        pub_store(a.stats + ((size_t)tile * a.Cout + cout0 + crel) * 2 + (tid & 1), s);
      }
    }
  }
  TailTicket tk;
  tk.old = 0u;
  tk.armed = 0;
  if (defer) {
    if constexpr (BNB) tk = bnb_tail_begin<CB>(a.tail, a.bn_c0, a.bn_c1, cout0, tile, cblk, tid);
    else tk = bnf_tail_begin(a.ftail, tile, cblk, tid);
  }
  if (defer) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const int cbase = cout0 + (P16 ? r : (mt * 32 + (r & 3) + 8 * (r >> 2)));
        const bool second = a.out1 != nullptr && cbase >= a.Csplit;
        const __amdgpu_buffer_rsrc_t rs_o = second ? rs_o1 : rs_o0;
        const unsigned soff = (unsigned)(second ? min(cbase, a.Cout) - a.Csplit : min(cbase, a.Csplit)) * hw4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) buf_store(rs_o, pvo[nt], soff, acc[mt][nt][r]);
      }
    if constexpr (BNB)
      bnb_tail<CB>(tk, a.tail, a.stats, a.bn_mean, a.bn_c0, a.bn_c1, cout0, tile, cblk, tid,
                   reinterpret_cast<double*>(red), reinterpret_cast<int*>(red + 520));
    else
      bnf_tail<CB>(tk, a.ftail, a.stats, a.Cout, cout0, tile, cblk, tid, reinterpret_cast<double*>(red),
                   reinterpret_cast<int*>(red + 520));
  }
}

template <int KS, int MODE, bool DB, int EPI>
static int launch_fwd(const ConvArgs& a, hipStream_t st) {
  constexpr int CB = (MODE == 0 || MODE == 3 || MODE == 4) ? 16 : 32 * MODE;
  ConvArgs args = a;
  const bool narrow = a.W <= 16;  // 16x16 tiles for the deepest levels, 8x32 otherwise
  const int TW = narrow ? 16 : 32, TH = 256 / TW;
  args.tiles_x = ceil_div(a.W, TW);
  args.tiles_y = ceil_div(a.H, TH);
  dim3 grid((unsigned)(a.B * args.tiles_x * args.tiles_y), (unsigned)ceil_div(a.CoutP, CB));
  // XCD-aware tile order (WTPSE_C16_XCD=0: dispatch order): first for the HBM-bound 16-channel x3 kernel (MODE 3), then every mode
  static const bool xcd_on = [] { const char* e = getenv("WTPSE_C16_XCD"); return !(e && e[0] == '0'); }();
  args.xcd_tiles = (xcd_on && grid.x % 8 == 0 && (long long)grid.x * grid.y >= 64) ? (int)(grid.x / 8) : 0;
  const bool in_launch = tail_in_launch((long long)grid.x * grid.y);     // else: the stand-alone finalize kernel behind the launch
  if (!in_launch) args.tail.tickets = args.ftail.tickets = nullptr;
  if (args.tail.tickets) bnb_tail_geometry(args.tail, (int)grid.x, a.Cout, (double)a.B * a.H * a.W);
  if (args.ftail.tickets) bnf_tail_geometry(args.ftail, (int)grid.x, a.Cout, (double)a.B * a.H * a.W);
  if (narrow)
    hipLaunchKernelGGL((conv_fwd_k<KS, MODE, 4, DB, EPI>), grid, dim3(256), 0, st, args);
  else
    hipLaunchKernelGGL((conv_fwd_k<KS, MODE, 5, DB, EPI>), grid, dim3(256), 0, st, args);
  int rc = wtpse_status();
  if (rc == 0 && !in_launch)
    rc = tail_after_launch(a.tail, a.ftail, a.stats, (int)grid.x, a.Cout, a.bn_c0, a.bn_c1, a.bn_mean, (long long)a.B * a.H * a.W, st);
  return rc;
}

static int fwd_tiles(int B, int H, int W) {
  const int TW = W <= 16 ? 16 : 32, TH = 256 / TW;
  return B * ceil_div(W, TW) * ceil_div(H, TH);
}

extern "C" int wtpse_conv_stats_blocks(int B, int H, int W) {
  const int TW = W <= 16 ? 16 : 32, TH = 256 / TW;
  return B * ceil_div(W, TW) * ceil_div(H, TH);
}

struct BnbArgs {   // EPI 2 parameters of conv_fwd_impl (all null / 0: none)
  const float* ss;
  const float* mean;
  int relu, c0, c1;
};

static int conv_fwd_impl(const float* in0, int C0, const float* in1, int C1, const float* wpacked, const float* bias,
                         const float* pro0, const float* pro1, int pro_relu, float* out0, float* out1, int Csplit, float* stats,
                         int B, int H, int W, int Cout, int ksize, int relu_out, const float* mask_ref, float* gram,
                         void* stream, BnbArgs bn = BnbArgs{nullptr, nullptr, 0, 0, 0}, BnbTail tail = bnb_tail_none(),
                         BnfTail ftail = bnf_tail_none(), unsigned* out_amax = nullptr) {
  WTPSE_REQUIRE(in0 && wpacked && out0 && B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0);
  WTPSE_REQUIRE(ksize == 1 || ksize == 3);
  WTPSE_REQUIRE((C1 == 0) == (in1 == nullptr));
  WTPSE_REQUIRE(Csplit > 0 && Csplit <= Cout && ((Csplit == Cout) == (out1 == nullptr)));
  WTPSE_REQUIRE(Csplit == Cout || Csplit % 16 == 0);   // the epilogue picks the output tensor per register, not per lane
  WTPSE_REQUIRE(!(stats && relu_out));
  const bool bnb = bn.mean != nullptr;
  WTPSE_REQUIRE(bnb || !(stats && mask_ref));
  WTPSE_REQUIRE(bnb || !(mask_ref && out1));
  WTPSE_REQUIRE(!bnb || (mask_ref && stats && bn.ss && !bias && !relu_out && !gram && bn.c0 >= 0 && bn.c0 < bn.c1 && bn.c1 <= Cout &&
                         bn.c0 % 16 == 0 && (bn.c1 % 16 == 0 || bn.c1 == Cout)));
  WTPSE_REQUIRE(C1 == 0 || C0 % 16 == 0);   // a channel chunk must not straddle the two inputs
  WTPSE_REQUIRE(!(out_amax && (mask_ref || bnb)));
  WTPSE_REQUIRE(!tail.tickets || (bnb && tail.partial2 && tail.gamma && tail.invstd && tail.coef && tail.dgamma && tail.dbeta));
  WTPSE_REQUIRE(!ftail.tickets || (!bnb && stats && !gram && ftail.partial2 && ftail.gamma && ftail.beta && ftail.scale_shift &&
                                   ftail.save_mean && ftail.save_invstd && (ftail.rmean == nullptr) == (ftail.rvar == nullptr)));
  ConvArgs a;
  a.tail = tail;
  a.ftail = ftail;
  a.bn_ss = bn.ss; a.bn_mean = bn.mean; a.bn_relu = bn.relu; a.bn_c0 = bnb ? bn.c0 : 0; a.bn_c1 = bnb ? bn.c1 : 0;
  a.in0 = in0; a.in1 = in1; a.wp = wpacked; a.bias = bias; a.pro0 = pro0; a.pro1 = pro1; a.out0 = out0; a.out1 = out1; a.stats = stats; a.mask = mask_ref; a.gram = gram;
  a.in_amax = nullptr; a.in_scale = 1.f; a.out_amax = out_amax;
  a.B = B; a.H = H; a.W = W; a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1; a.CinP = (a.Cin + 3) & ~3;
  a.Cout = Cout; a.CoutP = (Cout + 15) & ~15; a.Csplit = Csplit; a.pro_relu = pro_relu; a.relu_out = relu_out;
  a.tiles_x = a.tiles_y = 0;
  hipStream_t st = (hipStream_t)stream;
  int mode = Cout <= 16 ? 0 : (Cout % 64 == 0 ? 2 : 1);  // ragged channel counts run on the 32-wide path
  // Grids that cannot give every CU ~3 workgroups (the 16x16 / 32x32 levels): halve the cout block to double the
  // workgroup count, and overlap each workgroup's own loads with its MFMAs (register double-buffering)
  const int tiles = fwd_tiles(B, H, W);
  if (mode == 2 && tiles * (a.CoutP / 64) < 512) mode = 1;
  if (mode == 1 && tiles * ceil_div(a.CoutP, 32) < 384) mode = 0;   // still under two workgroups per CU: 16-cout blocks
  const int cb = mode == 0 ? 16 : 32 * mode;
  const bool db = tiles * ceil_div(a.CoutP, cb) < 768 && a.CinP > (mode == 0 ? WTPSE_P16_KC(ksize) : 8);
#define FWD(KS, M) (bnb ? (db ? launch_fwd<KS, M, true, 2>(a, st) : launch_fwd<KS, M, false, 2>(a, st)) \
                    : mask_ref ? (db ? launch_fwd<KS, M, true, 1>(a, st) : launch_fwd<KS, M, false, 1>(a, st)) \
                               : (db ? launch_fwd<KS, M, true, 0>(a, st) : launch_fwd<KS, M, false, 0>(a, st)))
  if (ksize == 3) {
    if (mode == 0) return FWD(3, 0);
    if (mode == 1) return FWD(3, 1);
    return FWD(3, 2);
  }
  if (mode == 0) return FWD(1, 0);
  if (mode == 1) return FWD(1, 1);
  return FWD(1, 2);
#undef FWD
}

// See include/wtpse_hip.h for the contract.
extern "C" int wtpse_conv_fwd(const float* in0, int C0, const float* in1, int C1, const float* wpacked,
                              const float* bias, const float* pro0, const float* pro1, int pro_relu, float* out0, float* out1,
                              int Csplit, float* stats, int B, int H, int W, int Cout, int ksize, int relu_out,
                              const float* mask_ref, unsigned* out_amax, void* stream) {
  return conv_fwd_impl(in0, C0, in1, C1, wpacked, bias, pro0, pro1, pro_relu, out0, out1, Csplit, stats, B, H, W, Cout, ksize,
                       relu_out, mask_ref, nullptr, stream, BnbArgs{nullptr, nullptr, 0, 0, 0}, bnb_tail_none(), bnf_tail_none(), out_amax);
}

// Data gradient that also performs the first half of the BatchNorm backward of the layer it flows into (include/wtpse_hip.h).
extern "C" int wtpse_dgrad_bnb(const float* dy, int C, const float* wpacked, float* out0, float* out1, int Csplit,
                               const float* bn_y, const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0, int bn_c1,
                               float* stats, int B, int H, int W, int Cout, int ksize, void* stream) {
  WTPSE_REQUIRE(bn_y && bn_ss && bn_mean && stats);
  return conv_fwd_impl(dy, C, nullptr, 0, wpacked, nullptr, nullptr, nullptr, 0, out0, out1, Csplit, stats, B, H, W, Cout, ksize, 0,
                       bn_y, nullptr, stream, BnbArgs{bn_ss, bn_mean, bn_relu, bn_c0, bn_c1});
}

// ---- the 16-channel 3x3 layers in the x3 arithmetic (MODE 3): Cout <= 16, Cin <= 16, one input tensor.  wx16: the layer's
// fragments from wtpse_pack_conv16_x3.  Everything optional: bias, prologue, ReLU, BatchNorm (sum, sum^2) partials `stats`,
// Gram partials `gram_partial` (Cout == 16), ReLU mask `mask_ref`, or — with bn_mean — the BatchNorm-backward epilogue of
// wtpse_dgrad_bnb over all output channels (mask_ref = that layer's raw conv output).
// Arithmetic: wtpse_x3_terms() == 2 -> x2h (MODE 4), unless the input is a GRADIENT (in_is_grad) whose amax table is not given — a
// gradient has no scale known a priori, and an extra pass to find it costs more than this HBM-bound kernel gains: those launches stay
// on x3 (the packed fragments carry both formats).
extern int g_x3_terms;      // conv_x3.hip
static int conv16_x3_impl(const float* in0, int C0, const unsigned short* wx16, const float* bias, const float* pro0,
                          int pro_relu, float* out0, float* stats, float* gram_partial, const float* mask_ref,
                          const float* bn_ss, const float* bn_mean, int bn_relu, int B, int H, int W, int Cout, int relu_out,
                          int in_is_grad, const unsigned* in_amax, void* stream, BnbTail tail = bnb_tail_none(),
                          BnfTail ftail = bnf_tail_none(), unsigned* out_amax = nullptr) {
  WTPSE_REQUIRE(in0 && wx16 && out0 && B > 0 && H > 0 && W > 0 && C0 > 0 && C0 <= 16 && Cout > 0 && Cout <= 16);
  WTPSE_REQUIRE(!(stats && relu_out) && !(gram_partial && (Cout != 16 || relu_out)));
  WTPSE_REQUIRE((((uintptr_t)wx16) & 15) == 0);
  const bool bnb = bn_mean != nullptr;
  WTPSE_REQUIRE(bnb || !(stats && mask_ref));
  WTPSE_REQUIRE(!bnb || (mask_ref && stats && bn_ss && !bias && !relu_out && !gram_partial));
  WTPSE_REQUIRE(!(out_amax && (mask_ref || bnb)));
  WTPSE_REQUIRE(!tail.tickets || (bnb && tail.partial2 && tail.gamma && tail.invstd && tail.coef && tail.dgamma && tail.dbeta));
  WTPSE_REQUIRE(!ftail.tickets || (!bnb && stats && !gram_partial && ftail.partial2 && ftail.gamma && ftail.beta && ftail.scale_shift &&
                                   ftail.save_mean && ftail.save_invstd && (ftail.rmean == nullptr) == (ftail.rvar == nullptr)));
  ConvArgs a;
  a.tail = tail;
  a.ftail = ftail;
  a.in0 = in0; a.in1 = nullptr; a.wp = reinterpret_cast<const float*>(wx16); a.bias = bias; a.pro0 = pro0; a.pro1 = nullptr;
  a.out0 = out0; a.out1 = nullptr; a.stats = stats; a.mask = mask_ref; a.gram = gram_partial;
  a.bn_ss = bn_ss; a.bn_mean = bn_mean; a.bn_relu = bn_relu; a.bn_c0 = 0; a.bn_c1 = bnb ? Cout : 0;
  a.B = B; a.H = H; a.W = W; a.C0 = C0; a.C1 = 0; a.Cin = C0; a.CinP = 16;
  a.Cout = Cout; a.CoutP = 16; a.Csplit = Cout; a.pro_relu = pro_relu; a.relu_out = relu_out;
  a.tiles_x = a.tiles_y = 0;
  a.in_amax = in_amax; a.in_scale = X3_FWD_SCALE; a.out_amax = out_amax;
  hipStream_t st = (hipStream_t)stream;
  if (g_x3_terms == 2 && (!in_is_grad || in_amax)) {
    if (bnb) return launch_fwd<3, 4, false, 2>(a, st);
    if (mask_ref) return launch_fwd<3, 4, false, 1>(a, st);
    return launch_fwd<3, 4, false, 0>(a, st);
  }
  if (bnb) return launch_fwd<3, 3, false, 2>(a, st);
  if (mask_ref) return launch_fwd<3, 3, false, 1>(a, st);
  return launch_fwd<3, 3, false, 0>(a, st);
}

extern "C" int wtpse_conv16_x3(const float* in0, int C0, const unsigned short* wx16, const float* bias, const float* pro0,
                               int pro_relu, float* out0, float* stats, float* gram_partial, const float* mask_ref,
                               const float* bn_ss, const float* bn_mean, int bn_relu, int B, int H, int W, int Cout, int relu_out,
                               int in_is_grad, const unsigned* in_amax, unsigned* out_amax, void* stream) {
  return conv16_x3_impl(in0, C0, wx16, bias, pro0, pro_relu, out0, stats, gram_partial, mask_ref, bn_ss, bn_mean, bn_relu, B, H, W,
                        Cout, relu_out, in_is_grad, in_amax, stream, bnb_tail_none(), bnf_tail_none(), out_amax);
}

// ---- wtpse_dgrad_bnb / wtpse_dgrad_x3_bnb / wtpse_conv16_x3(bn_mean) whose launch ALSO finishes the statistics: the last
// workgroups fold the partials (common.h: bnb_tail) and leave (k1, k2, k3) in `coef`, dgamma / dbeta (+)= in place, so that the
// BatchNorm backward is this launch + wtpse_bn_bwd_apply_coef.  layout: 0 fp32 (`wd`), 1 x3, 2 the 16-channel x3 fragments.
extern "C" int wtpse_dgrad_x3_bnb_tail(const float* dy, int C, const unsigned short* wpacked, float* out0, float* out1, int Csplit,
                                       const float* bn_y, const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0,
                                       int bn_c1, float* stats, const BnbTail* tail, int B, int H, int W, int Cout, int ksize,
                                       const unsigned* in_amax, void* stream);

extern "C" int wtpse_conv_fwd_x3_ftail(const float* in0, int C0, const float* in1, int C1, const unsigned short* wpacked,
                                       const float* bias, const float* pro0, const float* pro1, int pro_relu, float* out0,
                                       float* stats, const BnfTail* ftail, int B, int H, int W, int Cout, int ksize,
                                       const unsigned* in_amax0, const unsigned* in_amax1, void* stream);

// ---- a forward convolution in front of a train-mode BatchNorm whose launch ALSO finishes the statistics (common.h: bnf_tail):
// wtpse_conv_fwd / wtpse_conv_fwd_x3 / wtpse_conv16_x3 with `stats` + wtpse_bn_finalize in one launch.  layout as below.
extern "C" int wtpse_conv_fwd_bnf(const float* in0, int C0, const float* in1, int C1, const void* wpacked, int layout,
                                  const float* bias, const float* pro0, const float* pro1, int pro_relu, float* out0, float* stats,
                                  const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  long long* num_batches, float momentum, float eps, float* scale_shift, float* save_mean,
                                  float* save_invstd, double* partial2, unsigned* tickets, int B, int H, int W, int Cout, int ksize,
                                  const unsigned* in_amax0, const unsigned* in_amax1, unsigned* act_amax, void* stream) {
  WTPSE_REQUIRE(stats && gamma && beta && scale_shift && save_mean && save_invstd && partial2 && tickets);
  WTPSE_REQUIRE(layout >= 0 && layout <= 2);
  BnfTail t = bnf_tail_none();
  t.partial2 = partial2; t.tickets = tickets; t.gamma = gamma; t.beta = beta; t.rmean = running_mean; t.rvar = running_var;
  t.nbt = num_batches; t.momentum = momentum; t.eps = eps; t.scale_shift = scale_shift; t.save_mean = save_mean;
  t.save_invstd = save_invstd; t.act_amax = act_amax;
  if (layout == 1)
    return wtpse_conv_fwd_x3_ftail(in0, C0, in1, C1, static_cast<const unsigned short*>(wpacked), bias, pro0, pro1, pro_relu, out0,
                                   stats, &t, B, H, W, Cout, ksize, in_amax0, in_amax1, stream);
  if (layout == 2) {
    WTPSE_REQUIRE(ksize == 3 && !in1 && C1 == 0 && !pro1);
    return conv16_x3_impl(in0, C0, static_cast<const unsigned short*>(wpacked), bias, pro0, pro_relu, out0, stats, nullptr, nullptr,
                          nullptr, nullptr, 0, B, H, W, Cout, 0, 0, in_amax0, stream, bnb_tail_none(), t);
  }
  return conv_fwd_impl(in0, C0, in1, C1, static_cast<const float*>(wpacked), bias, pro0, pro1, pro_relu, out0, nullptr, Cout, stats,
                       B, H, W, Cout, ksize, 0, nullptr, nullptr, stream, BnbArgs{nullptr, nullptr, 0, 0, 0}, bnb_tail_none(), t);
}

extern "C" int wtpse_bnb_tail_partial2(int nblk, int Cout) { return bnb_tail_groups(nblk) * bnb_tail_ctot(Cout) * 2; }
extern "C" int wtpse_bnb_tail_tickets(int nblk, int Cout) { return bnb_tail_t2off(nblk, Cout) + (Cout + 15) / 16; }

extern "C" int wtpse_dgrad_bnb_coef(const float* dy, int C, const void* wpacked, int layout, float* out0, float* out1, int Csplit,
                                    const float* bn_y, const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0, int bn_c1,
                                    float* stats, const float* gamma, const float* invstd, float* coef, float* dgamma,
                                    float* dbeta, int accumulate, double* partial2, unsigned* tickets, int B, int H, int W,
                                    int Cout, int ksize, const unsigned* in_amax, void* stream) {
  WTPSE_REQUIRE(bn_y && bn_ss && bn_mean && stats && gamma && invstd && coef && dgamma && dbeta && partial2 && tickets);
  WTPSE_REQUIRE(layout >= 0 && layout <= 2);
  BnbTail t = bnb_tail_none();
  t.partial2 = partial2; t.tickets = tickets; t.gamma = gamma; t.invstd = invstd; t.coef = coef; t.dgamma = dgamma; t.dbeta = dbeta;
  t.accumulate = accumulate;
  if (layout == 1)
    return wtpse_dgrad_x3_bnb_tail(dy, C, static_cast<const unsigned short*>(wpacked), out0, out1, Csplit, bn_y, bn_ss, bn_mean,
                                   bn_relu, bn_c0, bn_c1, stats, &t, B, H, W, Cout, ksize, in_amax, stream);
  if (layout == 2) {
    WTPSE_REQUIRE(ksize == 3 && !out1 && Csplit == Cout && bn_c0 == 0 && bn_c1 == Cout);
    return conv16_x3_impl(dy, C, static_cast<const unsigned short*>(wpacked), nullptr, nullptr, 0, out0, stats, nullptr, bn_y, bn_ss,
                          bn_mean, bn_relu, B, H, W, Cout, 0, 1, in_amax, stream, t);
  }
  return conv_fwd_impl(dy, C, nullptr, 0, static_cast<const float*>(wpacked), nullptr, nullptr, nullptr, 0, out0, out1, Csplit, stats,
                       B, H, W, Cout, ksize, 0, bn_y, nullptr, stream, BnbArgs{bn_ss, bn_mean, bn_relu, bn_c0, bn_c1}, t);
}

// Weight fragments of the 16-channel x3 / x2h path, all convs of a network in one launch.  desc: n_desc x 8 ints {w_off, Cout, Cin,
// 9, fwd_off (-1: none), dgrad_off (-1: none), 0, 0}, w_off in floats into `params`, *_off in unsigned shorts into `packed`;
// one direction = 8 shorts of header (float {1 / scale, scale}) + [5 k-steps][3 terms][64 lanes][8] bf16 (x3: 7680 shorts) +
// [5 k-steps][2 terms][64 lanes][8] fp16 of scale * w (x2h: 5120 shorts) = 12808 shorts:  lane = (row = lane & 15, g = lane >> 4),
// tap = 2 s + (g >> 1), k = 8 (g & 1) + j;  forward: rows = Cout, k = Cin, w[row][k][tap];  data gradient: rows = Cin,
// k = Cout, w[k][row][8 - tap];  zero beyond the layer's channels and for the tenth tap.  scale: the power of two that brings the
// layer's largest |w| into [2^14, 2^15) (every workgroup of the layer finds it for itself: at most 2304 weights).
__global__ __launch_bounds__(256) void pack_weights_x3p16_k(const float* __restrict__ params, const int* __restrict__ desc,
                                                            unsigned short* __restrict__ packed) {
  const int* d = desc + blockIdx.y * 8;
  const int w_off = d[0], Co = d[1], Ci = d[2];
  const float* w = params + w_off;
  __shared__ float red[4];
  float m = 0.f;
  for (int e = threadIdx.x; e < Co * Ci * 9; e += 256) m = fmaxf(m, fabsf(w[e]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = x3_scale_from_amax(__builtin_bit_cast(unsigned, m));
  for (int dir = 0; dir < 2; ++dir) {
    const int base = d[4 + dir];
    if (base < 0) continue;
    const int R = dir == 0 ? Co : Ci, K = dir == 0 ? Ci : Co;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      float* hdr = reinterpret_cast<float*>(packed + base);
      hdr[0] = 1.f / sc; hdr[1] = sc; hdr[2] = hdr[3] = 0.f;
    }
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 5 * 64 * 8; e += gridDim.x * 256) {
      const int j = e & 7, lane = (e >> 3) & 63, s5 = e >> 9;
      const int row = lane & 15, g = lane >> 4;
      const int tap = 2 * s5 + (g >> 1), k = 8 * (g & 1) + j;
      float v = 0.f;
      if (tap < 9 && row < R && k < K) v = dir == 0 ? w[(row * Ci + k) * 9 + tap] : w[(k * Ci + row) * 9 + (8 - tap)];
      unsigned q0, q1, q2;
      wt_split3_pair(v, 0.f, q0, q1, q2);
      unsigned short* o = packed + base + 8;
      o[((s5 * 3 + 0) * 64 + lane) * 8 + j] = (unsigned short)(q0 & 0xFFFFu);
      o[((s5 * 3 + 1) * 64 + lane) * 8 + j] = (unsigned short)(q1 & 0xFFFFu);
      o[((s5 * 3 + 2) * 64 + lane) * 8 + j] = (unsigned short)(q2 & 0xFFFFu);
      split2h_pair(v * sc, 0.f, q0, q1);
      unsigned short* oh = o + 5 * 3 * 64 * 8;
      oh[((s5 * 2 + 0) * 64 + lane) * 8 + j] = (unsigned short)(q0 & 0xFFFFu);
      oh[((s5 * 2 + 1) * 64 + lane) * 8 + j] = (unsigned short)(q1 & 0xFFFFu);
    }
  }
}

extern "C" int wtpse_pack_conv16_x3(const float* params, const int* desc, int n_desc, unsigned short* packed, void* stream) {
  WTPSE_REQUIRE(params && desc && packed && n_desc > 0);
  hipLaunchKernelGGL(pack_weights_x3p16_k, dim3(2, n_desc), dim3(256), 0, (hipStream_t)stream, params, desc, packed);
  return wtpse_status();
}

// 3x3 convolution with exactly 16 output channels that also emits the per-tile partial Grams of its output
// (gram_partial: [wtpse_conv_stats_blocks(B,H,W)][256], tile-major per image = the `partial` layout of the WT loss with
// S = tiles per image; see wtpse_wt_loss_fwd_partials).
extern "C" int wtpse_conv_fwd_gram(const float* in0, int C0, const float* wpacked, const float* bias, const float* pro0,
                                   int pro_relu, float* out0, float* gram_partial, int B, int H, int W, int Cout, int relu_out,
                                   unsigned* out_amax, void* stream) {
  WTPSE_REQUIRE(gram_partial && Cout == 16);
  WTPSE_REQUIRE(!relu_out);     // the Gram epilogue works on the accumulators before the ReLU clamp: it describes the stored map only without one
  return conv_fwd_impl(in0, C0, nullptr, 0, wpacked, bias, pro0, nullptr, pro_relu, out0, nullptr, Cout, nullptr, B, H, W, Cout, 3,
                       relu_out, nullptr, gram_partial, stream, BnbArgs{nullptr, nullptr, 0, 0, 0}, bnb_tail_none(), bnf_tail_none(), out_amax);
}

// ------------------------------------------------------------------------------------------------
// Weight packing: OIHW -> forward layout wf[CinP][taps][CoutP] (wf[ci][t][co] = w[co][ci][t]) and
// data-gradient layout wd[CoutP4][taps][CinP16] (wd[co][t][ci] = w[co][ci][taps-1-t]); zero padded.
// One launch packs every conv of a network: `desc` holds 8 ints per conv:
//   {w_off, Cout, Cin, taps, wf_off, wd_off, unused, unused}   (offsets in floats)
__global__ __launch_bounds__(256) void pack_weights_k(const float* __restrict__ params, const int* __restrict__ desc,
                                                      float* __restrict__ packed) {
  const int* d = desc + blockIdx.y * 8;
  const int w_off = d[0], Co = d[1], Ci = d[2], T = d[3], wf_off = d[4], wd_off = d[5];
  const int CiP4 = (Ci + 3) & ~3, CoP16 = (Co + 15) & ~15;
  const int CoP4 = (Co + 3) & ~3, CiP16 = (Ci + 15) & ~15;
  const float* w = params + w_off;
  const int nf = wf_off >= 0 ? CiP4 * T * CoP16 : 0;      // a direction that runs on the x3 kernels has no fp32 layout (-1)
  for (int e = blockIdx.x * 256 + threadIdx.x; e < nf; e += gridDim.x * 256) {
    int co = e % CoP16;
    int t = (e / CoP16) % T;
    int ci = e / (CoP16 * T);
    packed[wf_off + e] = (co < Co && ci < Ci) ? w[(co * Ci + ci) * T + t] : 0.f;
  }
  if (wd_off >= 0) {
    const int nd = CoP4 * T * CiP16;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nd; e += gridDim.x * 256) {
      int ci = e % CiP16;
      int t = (e / CiP16) % T;
      int co = e / (CiP16 * T);
      packed[wd_off + e] = (co < Co && ci < Ci) ? w[(co * Ci + ci) * T + (T - 1 - t)] : 0.f;
    }
  }
}

extern "C" int wtpse_pack_conv_weights(const float* params, const int* desc, int n_desc, float* packed, void* stream) {
  WTPSE_REQUIRE(params && desc && packed && n_desc > 0);
  hipLaunchKernelGGL(pack_weights_k, dim3(16, n_desc), dim3(256), 0, (hipStream_t)stream, params, desc, packed);
  return wtpse_status();
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dW[co][ci][t] = sum_{b,y,x} dY[b,co,y,x] * X[b,ci,y+dy-1,x+dx-1]  (GEMM with K = pixels).
// A workgroup owns (cout block, cin group) and a strided share of the spatial tiles; its 4 waves split each
// tile's 256 pixels and keep their partial dW in MFMA accumulators across tiles; partials are summed through
// LDS and written to slab[ky]; wtpse_wgrad_reduce folds the slabs in a fixed order (bitwise reproducible).
// The N side is a list of (cin, tap) slots, 16|32 per block: slot = nb*NB + j -> cin = slot % cg, tap = slot / cg
// with cg = min(NB, Cin).  This covers both wide layers (cg = NB: one tap per block) and the 1-/3-/8-/16-channel
// inputs of the first layers and heads without padding channels.
struct WgradArgs {
  const float* dy;
  const float* x0;
  const float* x1;
  const float* pro0;
  const float* pro1;
  float* slab;    // [ksplit][Cout][Cin][taps]
  float* dbias;   // [ksplit][Cout] or null
  int B, H, W, C0, C1, Cin, Cout;
  int pro_relu;
  int tiles_x, tiles_y, ntiles;
  int cg, ngroups, nblk;
};

// __launch_bounds__(256, 2): two workgroups per CU (2 waves per SIMD) so that one loads while the other runs MFMAs;
// the 9 x 16 accumulators of the 32-wide path leave ~110 VGPRs for staging
template <int KS, bool P32, int TWL, int NBLK>
__global__ __launch_bounds__(256, 2) void conv_wgrad_k(WgradArgs a) {
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  constexpr int TW = 1 << TWL, TH = 256 / TW;
  constexpr int PITCH = TW + 2 * PAD, ROWS = TH + 2 * PAD;
  constexpr int PE = PITCH * ROWS;
  constexpr int MB = P32 ? 32 : 16;   // couts per block == cin slots per N block
  constexpr int KQ = P32 ? 2 : 4;     // pixels per MFMA
  constexpr int NACC = P32 ? 16 : 4;
  constexpr int MAXNB = NBLK;         // N blocks per workgroup: ceil(cg*TAPS/MB) rounded up to an instantiated count
  // bank-conflict-free strides: 16-path reads (cout|cin) x 2 pixels per 32-lane group -> stride % 32 == 2;
  // 32-path reads 32 channels of one pixel -> odd stride
  constexpr int SA = P32 ? 257 : 258;
  constexpr int SX = P32 ? (PE | 1) : (((PE - 2 + 31) / 32) * 32 + 2);
  constexpr int TILE_SZ = MB * SA + MB * SX;
  constexpr int RED_SZ = P32 ? 4 * 1024 : 4 * MAXNB * 256;
  __shared__ float smem[TILE_SZ > RED_SZ ? TILE_SZ : RED_SZ];
  float* Ys = smem;
  float* Xs = smem + MB * SA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroups are dealt to the 8 XCDs round-robin by linear id.  The gridDim.x workgroups of one k-slice read the same
  // dY / X tiles, so (when the slice count is a multiple of 8) slice ky is given to XCD ky % 8 as a whole: its tiles
  // are then fetched into that XCD's L2 once instead of once per XCD the slice was spread over.
  int bx = blockIdx.x, ky = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int k = lin >> 3;
    ky = (lin & 7) + 8 * (k / (int)gridDim.x);
    bx = k % (int)gridDim.x;
  }
  const int group = bx % a.ngroups;
  const int cout0 = (bx / a.ngroups) * MB;
  const int cin0 = group * a.cg;
  const int HW = a.H * a.W;
  const int j = lane & (MB - 1), kl = lane / MB;
  const bool any_pro = a.pro0 != nullptr || a.pro1 != nullptr || a.pro_relu != 0;

  int boff[MAXNB];
#pragma unroll
  for (int nb = 0; nb < MAXNB; ++nb) {
    int slot = nb * MB + j;
    int ci = slot % a.cg, t = slot / a.cg;
    // slots past the last tap read a valid address and accumulate garbage that is never stored: no branch in the MFMA loop
    boff[nb] = (t < TAPS) ? ci * SX + (t / KS) * PITCH + (t % KS) : 0;
  }
  typename AccT<!P32>::type acc[MAXNB];
#pragma unroll
  for (int nb = 0; nb < MAXNB; ++nb)
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[nb][r] = 0.f;
  double db = 0.0;

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  int it = 0;
  for (int tile = ky; tile < a.ntiles; tile += gridDim.y, ++it) {
    const int b = tile / tiles_per_img;
    const int trem = tile - b * tiles_per_img;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    __syncthreads();
    // ---- tile loads, batched: every global load of a phase is issued before the first LDS store
    // dY tile [MB couts][256 pixels] (this thread: pixel `tid` of every channel), zero outside the image / beyond Cout
    const int gyp = ty * TH + (tid >> TWL), gxp = tx * TW + (tid & (TW - 1));
    const int ypos = (gyp < a.H && gxp < a.W) ? gyp * a.W + gxp : -1;
    // X halo tile [cg cins][ROWS][PITCH] with the same fused affine/ReLU as the forward loader
    constexpr int NPOS = (PE + 255) / 256;
    int lpos[NPOS], gpos[NPOS];
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
      int p = tid + 256 * i;
      int r = p / PITCH, x = p - r * PITCH;
      int gy = ty * TH + r - PAD, gx = tx * TW + x - PAD;
      lpos[i] = p < PE ? p : -1;
      gpos[i] = (p < PE && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? gy * a.W + gx : -1;
    }
#pragma unroll
    for (int g = 0; g < MB / 16; ++g) {
      float yv[16];
      float xv[16][NPOS];
      const __amdgpu_buffer_rsrc_t rsy = make_rsrc(a.dy + (size_t)b * a.Cout * HW, (unsigned)a.Cout * HW * 4u);
      // a cin group never straddles the two inputs (C0 % cg == 0 is checked on the host); channels past the tensor
      // re-read a valid plane: what they accumulate is never stored
      // each 16-channel half of a cin group lies in ONE of the two inputs (C0 % 16 == 0 is checked on the host)
      const int ch0 = cin0 + g * 16;
      const bool xfirst = ch0 < a.C0 || a.x1 == nullptr;
      const int xbase = xfirst ? ch0 : ch0 - a.C0;
      const int xmax = (xfirst ? a.C0 : a.C1) - 1;
      const __amdgpu_buffer_rsrc_t rsx = xfirst ? make_rsrc(a.x0 + (size_t)b * a.C0 * HW, (unsigned)a.C0 * HW * 4u)
                                                : make_rsrc(a.x1 + (size_t)b * a.C1 * HW, (unsigned)a.C1 * HW * 4u);
      const unsigned yoff = ypos >= 0 ? (unsigned)ypos * 4u : BUF_OOB;
      unsigned voff[NPOS];
#pragma unroll
      for (int i = 0; i < NPOS; ++i) voff[i] = gpos[i] >= 0 ? (unsigned)gpos[i] * 4u : BUF_OOB;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        yv[c] = buf_load(rsy, yoff, (unsigned)min(cout0 + g * 16 + c, a.Cout - 1) * (unsigned)HW * 4u);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const unsigned soff = (unsigned)min(xbase + c, xmax) * (unsigned)HW * 4u;
#pragma unroll
        for (int i = 0; i < NPOS; ++i) xv[c][i] = buf_load(rsx, voff[i], soff);
      }
      if (any_pro) {
        const bool relu = xfirst ? (a.pro_relu & 1) : (a.pro_relu & 2);
        const float* pro = xfirst ? a.pro0 : a.pro1;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const int cgl = min(xbase + c, xmax);
          const float sc = pro ? pro[2 * cgl] : 1.f, sh = pro ? pro[2 * cgl + 1] : 0.f;
#pragma unroll
          for (int i = 0; i < NPOS; ++i) {
            float v = fmaf(xv[c][i], sc, sh);
            if (relu) v = fmaxf(v, 0.f);
            xv[c][i] = gpos[i] >= 0 ? v : 0.f;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) Ys[(g * 16 + c) * SA + tid] = yv[c];
#pragma unroll
      for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int i = 0; i < NPOS; ++i)
          if (lpos[i] >= 0) Xs[(g * 16 + c) * SX + lpos[i]] = xv[c][i];
    }
    __syncthreads();
    if (a.dbias && group == 0) {  // bias gradient: plain per-channel sum of the dY tile
      int c = tid / (256 / MB), part = tid % (256 / MB);
      constexpr int PER = MB;     // 256 pixels / (256/MB) threads
      float s = 0.f;
      for (int i = 0; i < PER; ++i) s += Ys[c * SA + part * PER + i];
      db += s;
    }
    // MFMA loop, software pipelined by hand: the LDS reads of step s+1 are in flight while the MFMAs of step s issue
    // back to back (left to the compiler, every MFMA sat behind its own ds_read + s_waitcnt: 172 cycles per MFMA
    // instead of 64, measured with phase stamps in round 1).  sched_barrier keeps the two groups apart.
    constexpr int STEPS = 64 / KQ;
    const float* yrow = Ys + j * SA + wave * 64 + kl;
    auto lds_step = [&](int s, float& av, float (&bv)[MAXNB]) {
      const int pbase = wave * 64 + s * KQ;
      const int rowbase = (pbase >> TWL) * PITCH + (pbase & (TW - 1)) + kl;
      av = yrow[s * KQ];
      if constexpr (KS == 3 && NBLK == 9) {
        // 9 blocks <=> cg == MB: block nb is tap nb of channel j, a compile-time offset from one per-lane address
        const float* xb = Xs + j * SX + rowbase;
#pragma unroll
        for (int nb = 0; nb < MAXNB; ++nb) bv[nb] = xb[(nb / 3) * PITCH + (nb % 3)];
      } else {
#pragma unroll
        for (int nb = 0; nb < MAXNB; ++nb) bv[nb] = Xs[boff[nb] + rowbase];
      }
    };
    float a0, a1, b0[MAXNB], b1[MAXNB];
    lds_step(0, a0, b0);
    for (int s = 0; s < STEPS; s += 2) {
      lds_step(s + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < MAXNB; ++nb) acc[nb] = mfma(a0, b0[nb], acc[nb]);
      __builtin_amdgcn_sched_barrier(0);
      lds_step(s + 2 < STEPS ? s + 2 : s, a0, b0);   // the last iteration re-reads a valid step; the values are unused
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < MAXNB; ++nb) acc[nb] = mfma(a1, b1[nb], acc[nb]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- cross-wave reduction through LDS, then slab[ky][co][ci][t]
  float* slab = a.slab + (size_t)ky * a.Cout * a.Cin * TAPS;
  float* red = smem;
  if constexpr (!P32) {
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < MAXNB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * MAXNB + nb) * 16 + (lane >> 4) * 4 + r) * 16 + j] = acc[nb][r];
    __syncthreads();
    for (int e = tid; e < MAXNB * 256; e += 256) {
      int nb = e >> 8, co = (e >> 4) & 15, jj = e & 15;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[((w * MAXNB + nb) * 16 + co) * 16 + jj];
      int slot = nb * 16 + jj;
      int ci = slot % a.cg, t = slot / a.cg;
      if (t < TAPS && cout0 + co < a.Cout && cin0 + ci < a.Cin)
        slab[((size_t)(cout0 + co) * a.Cin + cin0 + ci) * TAPS + t] = v;
    }
  } else {
#pragma unroll
    for (int nb = 0; nb < MAXNB; ++nb) {
      {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          red[(wave * 32 + co) * 32 + j] = acc[nb][r];
        }
        __syncthreads();
        for (int e = tid; e < 1024; e += 256) {
          int co = e >> 5, jj = e & 31;
          float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
          int slot = nb * 32 + jj;
          int ci = slot % a.cg, t = slot / a.cg;
          if (t < TAPS && cout0 + co < a.Cout && cin0 + ci < a.Cin)
            slab[((size_t)(cout0 + co) * a.Cin + cin0 + ci) * TAPS + t] = v;
        }
      }
    }
  }
  if (a.dbias && group == 0) {
    constexpr int TPC = 256 / MB;  // threads per channel: 16 (P16) or 8 (P32), consecutive lanes
    for (int m = 1; m <= TPC / 2; m <<= 1) db += __shfl_xor(db, m, 64);
    int c = tid / TPC;
    if ((tid % TPC) == 0 && cout0 + c < a.Cout) a.dbias[(size_t)ky * a.Cout + cout0 + c] = (float)db;
  }
}

// out[i] (+)= sum_k slab[k][i]: 32 outputs x 8 k-slices per workgroup, fp64 accumulation (the slabs are partial sums of
// a long, cancellation-prone reduction), fixed order -> bitwise reproducible
// The first nblk_w workgroups fold the weight slabs, the rest (if any) the bias slabs: one launch per convolution.
__global__ __launch_bounds__(256) void wgrad_reduce_k(const float* __restrict__ slab, int ksplit, int n,
                                                      float* __restrict__ out, int accumulate, int nblk_w,
                                                      const float* __restrict__ slab_b, int n_b, float* __restrict__ out_b) {
  __shared__ double sh[8][32];
  const int j = threadIdx.x & 31, kq = threadIdx.x >> 5;
  int blk = blockIdx.x;
  if (blk >= nblk_w) { blk -= nblk_w; slab = slab_b; n = n_b; out = out_b; }
  const int i = blk * 32 + j;
  double s = 0.0;
  if (i < n)
    for (int k = kq; k < ksplit; k += 8) s += (double)slab[(size_t)k * n + i];
  sh[kq][j] = s;
  __syncthreads();
  if (kq == 0 && i < n) {
    double t = sh[0][j] + sh[1][j] + sh[2][j] + sh[3][j] + sh[4][j] + sh[5][j] + sh[6][j] + sh[7][j];
    out[i] = accumulate ? out[i] + (float)t : (float)t;
  }
}

// The same fold with 16-byte loads for n % 4 == 0 (the x3 weight gradient: tens of MB of slabs per layer): a workgroup owns
// 128 outputs, its 8 groups of 32 lanes take every 8th slab (4 rows of 512 contiguous bytes in flight per group), fp64
// accumulation, groups combined in fixed order -> bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_fold4_k(const float* __restrict__ slab, int ksplit, int n, float* __restrict__ out,
                                                     int accumulate) {
  __shared__ double sh[8][32][4];
  const int j = threadIdx.x & 31, kq = threadIdx.x >> 5;
  const int i = (blockIdx.x * 32 + j) * 4;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (i < n) {
    int k = kq;
    for (; k + 24 < ksplit; k += 32) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(slab + (size_t)k * n + i);
      const f32x4 b = *reinterpret_cast<const f32x4*>(slab + (size_t)(k + 8) * n + i);
      const f32x4 c = *reinterpret_cast<const f32x4*>(slab + (size_t)(k + 16) * n + i);
      const f32x4 d = *reinterpret_cast<const f32x4*>(slab + (size_t)(k + 24) * n + i);
      s0 += (double)a[0]; s1 += (double)a[1]; s2 += (double)a[2]; s3 += (double)a[3];
      s0 += (double)b[0]; s1 += (double)b[1]; s2 += (double)b[2]; s3 += (double)b[3];
      s0 += (double)c[0]; s1 += (double)c[1]; s2 += (double)c[2]; s3 += (double)c[3];
      s0 += (double)d[0]; s1 += (double)d[1]; s2 += (double)d[2]; s3 += (double)d[3];
    }
    for (; k < ksplit; k += 8) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(slab + (size_t)k * n + i);
      s0 += (double)a[0]; s1 += (double)a[1]; s2 += (double)a[2]; s3 += (double)a[3];
    }
  }
  sh[kq][j][0] = s0; sh[kq][j][1] = s1; sh[kq][j][2] = s2; sh[kq][j][3] = s3;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int jj = threadIdx.x >> 2, e = threadIdx.x & 3;
    const int o = (blockIdx.x * 32 + jj) * 4 + e;
    if (o < n) {
      const double t = sh[0][jj][e] + sh[1][jj][e] + sh[2][jj][e] + sh[3][jj][e] + sh[4][jj][e] + sh[5][jj][e] + sh[6][jj][e] +
                       sh[7][jj][e];
      out[o] = accumulate ? out[o] + (float)t : (float)t;
    }
  }
}

// fold of the k-split slabs for other translation units (conv_x3.hip)
extern "C" void wtpse_wgrad_reduce_launch(const float* slab, int ksplit, int n, float* dw, int accumulate, void* stream) {
  if (n % 4 == 0)
    hipLaunchKernelGGL(wgrad_fold4_k, dim3(ceil_div(n, 128)), dim3(256), 0, (hipStream_t)stream, slab, ksplit, n, dw, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_k, dim3(ceil_div(n, 32)), dim3(256), 0, (hipStream_t)stream, slab, ksplit, n, dw, accumulate,
                       ceil_div(n, 32), (const float*)nullptr, 0, (float*)nullptr);
}

// the same with the bias-gradient slabs folded in the same launch (wgrad_r.hip)
extern "C" void wtpse_wgrad_reduce_launch2(const float* slab, int ksplit, int n, float* dw, int accumulate, const float* slab_b,
                                           int n_b, float* db, void* stream) {
  if (!db) {
    wtpse_wgrad_reduce_launch(slab, ksplit, n, dw, accumulate, stream);
    return;
  }
  const int nblk_w = ceil_div(n, 32), nblk_b = ceil_div(n_b, 32);
  hipLaunchKernelGGL(wgrad_reduce_k, dim3(nblk_w + nblk_b), dim3(256), 0, (hipStream_t)stream, slab, ksplit, n, dw, accumulate, nblk_w,
                     slab_b, n_b, db);
}

extern "C" int wtpse_wgrad_ksplit(int B, int H, int W, int Cin, int Cout) {
  const int TW = W <= 16 ? 16 : 32, TH = 256 / TW;
  const int ntiles = B * ceil_div(W, TW) * ceil_div(H, TH);
  const bool p32 = Cout > 16;
  const int MB = p32 ? 32 : 16;
  const int cg = Cin < MB ? Cin : MB;
  const int nx = ceil_div(Cout, MB) * ceil_div(Cin, cg);
  int ks = 512 / nx;   // ~2 workgroups per CU (LDS-limited residency of the 32-wide path)
  if (ks < 1) ks = 1;
  if (ks > ntiles) ks = ntiles;
  return ks;
}

extern "C" int wtpse_conv_wgrad(const float* dy, const float* x0, int C0, const float* x1, int C1, const float* pro0,
                                const float* pro1, int pro_relu, float* slab, float* dbias_slab, int ksplit, float* dw, float* dbias,
                                int accumulate, int B, int H, int W, int Cout, int ksize, void* stream) {
  WTPSE_REQUIRE(dy && x0 && slab && dw && B > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0 && ksplit > 0);
  WTPSE_REQUIRE(ksize == 1 || ksize == 3);
  WTPSE_REQUIRE((C1 == 0) == (x1 == nullptr));
  WTPSE_REQUIRE((dbias == nullptr) == (dbias_slab == nullptr));
  const bool p32 = Cout > 16;
  WgradArgs a;
  a.dy = dy; a.x0 = x0; a.x1 = x1; a.pro0 = pro0; a.pro1 = pro1; a.slab = slab; a.dbias = dbias_slab;
  a.B = B; a.H = H; a.W = W; a.C0 = C0; a.C1 = C1; a.Cin = C0 + C1; a.Cout = Cout; a.pro_relu = pro_relu;
  const int MB = p32 ? 32 : 16;
  const int taps = ksize * ksize;
  a.cg = a.Cin < MB ? a.Cin : MB;
  a.ngroups = ceil_div(a.Cin, a.cg);   // a ragged last group re-reads valid planes; its slots are never stored
  WTPSE_REQUIRE(C1 == 0 || C0 % 16 == 0);   // a 16-channel half group must not straddle the two inputs
  a.nblk = ceil_div(a.cg * taps, MB);
  const bool narrow = W <= 16;
  const int TW = narrow ? 16 : 32, TH = 256 / TW;
  a.tiles_x = ceil_div(W, TW);
  a.tiles_y = ceil_div(H, TH);
  a.ntiles = B * a.tiles_x * a.tiles_y;
  WTPSE_REQUIRE(ksplit <= a.ntiles);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(ceil_div(Cout, MB) * a.ngroups), (unsigned)ksplit);
  // instantiated N-block counts: 3x3 -> {1, 2, 5, 9}, 1x1 -> {1}
  const int nb_inst = ksize == 1 ? 1 : (a.nblk <= 1 ? 1 : a.nblk <= 2 ? 2 : a.nblk <= 5 ? 5 : 9);
#define WG_LAUNCH(KS, P, T, N) hipLaunchKernelGGL((conv_wgrad_k<KS, P, T, N>), grid, dim3(256), 0, st, a)
#define WG_TW(KS, P, N) do { if (narrow) WG_LAUNCH(KS, P, 4, N); else WG_LAUNCH(KS, P, 5, N); } while (0)
#define WG_NB(KS, P) do { if (nb_inst == 1) WG_TW(KS, P, 1); else if (nb_inst == 2) WG_TW(KS, P, 2); \
                          else if (nb_inst == 5) WG_TW(KS, P, 5); else WG_TW(KS, P, 9); } while (0)
  if (ksize == 3) {
    if (p32) WG_NB(3, true); else WG_NB(3, false);
  } else {
    if (p32) WG_TW(1, true, 1); else WG_TW(1, false, 1);
  }
#undef WG_NB
#undef WG_TW
#undef WG_LAUNCH
  int rc = wtpse_status();
  if (rc) return rc;
  const int n = Cout * a.Cin * taps;
  const int nblk_w = ceil_div(n, 32), nblk_b = dbias ? ceil_div(Cout, 32) : 0;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3(nblk_w + nblk_b), dim3(256), 0, st, slab, ksplit, n, dw, accumulate, nblk_w,
                     dbias_slab, Cout, dbias);
  return wtpse_status();
}
