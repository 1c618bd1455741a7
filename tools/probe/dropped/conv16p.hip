// The 16-channel 3x3 FORWARD convolution in the x2h arithmetic as a tile-streaming kernel (round 6).
//
// Same layers, same arithmetic, same products in the same order as conv_fwd_k MODE 4 (conv.hip: inc.conv2 / conv3, DeepWT's three
// 16 -> 16 convolutions, the teacher's inc — reference algorithms.py:897-917,1091-1117,398-413): one 16-channel chunk, split once
// into two fp16 terms, the layer's weight fragments in registers, 16x16x32 MFMAs.  What differs is WHO hides a tile's
// load -> convert -> multiply -> store chain.  These launches have no channel-chunk loop to hide it behind; rounds 3-5 left it to the
// four workgroups resident per CU and reached 0.36 of HBM peak (0.49 of the box's copy rate) — and a fifth resident workgroup changes
// nothing (round 6, profiles/NOTES_r06.md: the kernel is not occupancy-bound, its workgroups are chains of latencies with nothing
// of their own in flight).  Here a workgroup works through `tpw` CONSECUTIVE tiles and issues the next tile's loads the moment the
// current tile has left the staging registers: they land behind the current tile's MFMAs, epilogue and stores — zero extra
// registers, one more tile of loads in flight per workgroup at any time.  (The same loop wrapped around conv_fwd_k blew its register
// allocation up — a 300-byte argument block and nine epilogue variants hoisted out of the loop: 168 VGPRs + spills — hence a kernel of
// its own with the forward options only: prologue, bias, output ReLU, BatchNorm (sum, sum^2) partials, Gram partials, amax.)
#include "common.h"

struct Conv16pArgs {
  const float* in;
  const unsigned short* wx16;   // the layer's fragments (wtpse_pack_conv16_x3): 16-byte header, x3 fragments, x2h fragments
  const float* bias;
  const float* pro;             // [C0][2] or null
  float* out;
  float* stats;                 // [tiles][Cout][2] or null
  float* gram;                  // [tiles][256] or null
  const unsigned* in_amax;      // bound of the input as loaded, or null: in_scale
  unsigned* out_amax;           // amax table of the stored output (zero on entry), or null
  float in_scale;
  int B, H, W, C0, Cout;
  int pro_relu, relu_out;
  int tiles_x, tiles_y, tpw;    // gridDim.x * tpw = tiles
};

__global__ __launch_bounds__(256, 3) void conv16p_k(Conv16pArgs a) {
  constexpr int TW = 32, TH = 8, PITCH = TW + 2, ROWS = TH + 2, PE = PITCH * ROWS, PEP = (PE + 7) & ~7;
  constexpr int NT = 4, PB = (PE + 63) / 64, NIT = (2 * PB + 3) / 4;
  constexpr int GRAM_F = 4 * 16 * 65 + 4 * 256;
  constexpr int SM_F = 2 * 2 * PEP * 4 > GRAM_F ? 2 * 2 * PEP * 4 : GRAM_F;
  // (the layer's weight fragments and the scaled prologue coefficients live in LDS, not in 40 vector / 32 scalar registers: with the
  // next tile's 24 staging registers live through the MFMAs and the epilogue the register-resident form spilled 78 + 81 of them)
  __shared__ __attribute__((aligned(16))) float smem[SM_F + 5 * 2 * 64 * 4 + 16 + 32];
  u32x4* Xq = reinterpret_cast<u32x4*>(smem);        // [term 2][k-half 2][PEP positions] 16-byte rows of 8 channels
  u32x4* Wl = reinterpret_cast<u32x4*>(smem + SM_F); // [k-step 5][term 2][64 lanes] x2h fragments of the layer
  float* bias_s = smem + SM_F + 5 * 2 * 64 * 4;
  float2* pro_s = reinterpret_cast<float2*>(bias_s + 16);      // [16] (scale, shift) x input scale

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g4 = lane >> 4;
  const int HW = a.H * a.W;
  const unsigned hw4 = (unsigned)HW * 4u;
  if (tid < 16) bias_s[tid] = (a.bias && tid < a.Cout) ? a.bias[tid] : 0.f;

  const unsigned sx_raw = amax_load(a.in_amax);       // (read issued in front of the tile loads, picked up behind them: common.h)
  __builtin_amdgcn_sched_barrier(0);
  float2 praw = make_float2(1.f, 0.f);              // thread c < 16: the prologue coefficients of input channel c
  if (tid < 16 && a.pro) praw = make_float2(a.pro[2 * min(tid, a.C0 - 1)], a.pro[2 * min(tid, a.C0 - 1) + 1]);
  int ipos[NIT], ihalf[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int blk = __builtin_amdgcn_readfirstlane(i * 4 + wave);
    ihalf[i] = blk >= PB ? 1 : 0;
    const int p = (blk - ihalf[i] * PB) * 64 + lane;
    ipos[i] = (blk < 2 * PB && p < PE) ? p : -1;
  }
  int off[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = wave * 64 + nt * 16 + (lane & 15);
    off[nt] = (p >> 5) * PITCH + (p & 31);
  }
  int toff[5];      // k-step s5 covers taps 2 s5 (lane groups 0, 1) and 2 s5 + 1 (groups 2, 3); the tenth "tap" has zero weights
#pragma unroll
  for (int s5 = 0; s5 < 5; ++s5) {
    const int t = min(2 * s5 + (g4 >> 1), 8);
    toff[s5] = (t / 3) * PITCH + (t % 3);
  }
  const int hsel = (g4 & 1) * PEP;

  float xv[NIT][8];
  bool iin[NIT];
  auto issue_tile = [&](int t) __attribute__((always_inline)) {
    int q = t;
    const int tx = q % a.tiles_x;
    q /= a.tiles_x;
    const int ty = q % a.tiles_y, b = q / a.tiles_y;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.in + (size_t)b * a.C0 * HW, (unsigned)a.C0 * HW * 4u);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int p = max(ipos[i], 0);
      const int r = p / PITCH, x = p - r * PITCH;
      const int gy = ty * TH + r - 1, gx = tx * TW + x - 1;
      iin[i] = ipos[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      // (the channel plane rides in the per-lane offset, not in 24 loop-invariant scalar offsets the compiler would keep live across the
      // tile loop; channels past C0 land beyond num_records and read as zero — their weights are zero too —, and so does BUF_OOB + anything)
      const unsigned vo = (iin[i] ? (unsigned)(gy * a.W + gx) * 4u : BUF_OOB) + (unsigned)(ihalf[i] * 8) * hw4;
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[i][j] = buf_load(rs, vo + (unsigned)j * hw4, 0u);
    }
  };
  // XCD-aware order (as conv_fwd_k): the hardware deals consecutive workgroups to the 8 XCDs in turn; XCD q takes a contiguous range of
  // slots — whole tile rows lie in one slot, and the rows above and below it in the same L2
  int slot = (int)blockIdx.x;
  if ((gridDim.x & 7) == 0 && gridDim.x >= 64) slot = (slot & 7) * ((int)gridDim.x >> 3) + (slot >> 3);
  const int t0 = slot * a.tpw, t1 = t0 + a.tpw;
  issue_tile(t0);
  // the x2h fragments of the layer: [k-step 5][term 2] fp16 pairs of scale * w, one 16-byte row per lane, behind the header + x3 part
  const u32x4* wq = reinterpret_cast<const u32x4*>(a.wx16) + 1 + 15 * 64;
#pragma unroll
  for (int it = 0; it < 3; ++it)
    if (tid + 256 * it < 5 * 2 * 64) Wl[tid + 256 * it] = wq[tid + 256 * it];
  const float sx = a.in_amax ? x3_scale_from_amax(amax_reduce(sx_raw)) : a.in_scale;
  const float inv = reinterpret_cast<const float*>(a.wx16)[0] / sx;
  if (tid < 16) pro_s[tid] = make_float2(praw.x * sx, praw.y * sx);      // (exact: a power of two)
  const bool relu_in = a.pro_relu & 1;
  const float relu_lo = a.relu_out ? 0.f : -INFINITY;
  const int clane = g4 * 4;
  __syncthreads();      // bias_s, pro_s, Wl

#pragma nounroll
  for (int tile = t0; tile < t1; ++tile) {
    // ---- conversion of the tile in the staging registers: prologue (affine, ReLU; zero padding AFTER it, as in the reference graph),
    // scale (folded into the coefficients), split into two fp16 terms, 16-byte LDS rows
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float2 pc = pro_s[ihalf[i] * 8 + j];
        float v = fmaf(xv[i][j], pc.x, pc.y);
        if (relu_in) v = fmaxf(v, 0.f);
        xv[i][j] = (iin[i] && ihalf[i] * 8 + j < a.C0) ? v : 0.f;
      }
      if (ipos[i] >= 0) {
        u32x4 tt[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned q0, q1;
          split2h_pair(xv[i][2 * j], xv[i][2 * j + 1], q0, q1);
          tt[0][j] = q0; tt[1][j] = q1;
        }
        Xq[(0 * 2 + ihalf[i]) * PEP + ipos[i]] = tt[0];
        Xq[(1 * 2 + ihalf[i]) * PEP + ipos[i]] = tt[1];
      }
    }
    // the staging registers are free: the next tile's loads go out now and land behind this tile's MFMAs, epilogue and stores
    if (tile + 1 < t1) issue_tile(tile + 1);
    __syncthreads();
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) {
        const u32x4 b0 = Xq[0 * PEP + hsel + off[nt] + toff[s5]], b1 = Xq[2 * PEP + hsel + off[nt] + toff[s5]];
        const u32x4 a0 = Wl[(s5 * 2 + 0) * 64 + lane], a1 = Wl[(s5 * 2 + 1) * 64 + lane];
        c = wt_mfma16x32h(a0, b1, c);       // the three products, smallest first (as conv_fwd_k MODE 4 / conv_x3_k)
        c = wt_mfma16x32h(a1, b0, c);
        c = wt_mfma16x32h(a0, b0, c);
      }
      acc[nt] = c * inv;
    }
    // ---- epilogue: bias, [Gram partial], stores (pixels on lanes -> contiguous runs per channel plane), [statistics partials], [amax]
    int q = tile;
    const int tx = q % a.tiles_x;
    q /= a.tiles_x;
    const int ty = q % a.tiles_y, b = q / a.tiles_y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float bz = bias_s[clane + r];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt][r] += bz;
    }
    unsigned pvo[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int p = wave * 64 + nt * 16 + (lane & 15);
      const int gy = ty * TH + (p >> 5), gx = tx * TW + (p & 31);
      const bool ok = gy < a.H && gx < a.W;
      pvo[nt] = ok ? (unsigned)(clane * HW + gy * a.W + gx) * 4u : BUF_OOB;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[nt][r] = (ok && clane + r < a.Cout) ? acc[nt][r] : 0.f;      // ragged parts: zero (statistics, Gram, amax)
    }
    __syncthreads();      // every wave is done reading the tile's image: the epilogue reuses LDS, and the next conversion overwrites it
    if (a.gram) {
      // Gram of the output tile (the WT loss's G = z z^T, reference algorithms.py:1283; see conv_fwd_k): accumulators
      // [channel in registers][pixel on lanes] -> one trip through a wave-private LDS tile -> 16x16x4 MFMAs with A == B
      float* zs = smem + wave * (16 * 65);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) zs[(clane + r) * 65 + nt * 16 + (lane & 15)] = acc[nt][r];
      __builtin_amdgcn_wave_barrier();
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const float v = zs[(lane & 15) * 65 + 4 * st + g4];
        g = mfma16(v, v, g);
      }
      float* gs = smem + 4 * 16 * 65 + wave * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) gs[(clane + r) * 16 + (lane & 15)] = g[r];
      __syncthreads();
      const float* g0 = smem + 4 * 16 * 65;
      a.gram[(size_t)tile * 256 + tid] = g0[tid] + g0[256 + tid] + g0[512 + tid] + g0[768 + tid];
      __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(a.out + (size_t)b * a.Cout * HW, (unsigned)a.Cout * HW * 4u);
    unsigned am = 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float v = out_clamp<0>(acc[nt][r], relu_lo);
        am = max(am, amax_bits(v));
        buf_store(rs_o, pvo[nt], (unsigned)min(r, a.Cout) * hw4, v);
      }
    if (a.out_amax) amax_publish_wave(a.out_amax, am, (unsigned)tile * 4u + (unsigned)wave);
    if (a.stats) {
      // (sum, sum^2) of the tile per channel: the butterfly of conv_fwd_k's 16-channel path (8 values -> 3 halving steps + 1 plain),
      // the four waves meet in LDS; one row of partials per tile, as the one-tile-per-workgroup kernel writes them
      float sv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma clang fp contract(off)
          const float v = out_clamp<0>(acc[nt][r], relu_lo);
          s1 += v;
          s2 += v * v;
        }
        sv[r * 2 + 0] = s1;
        sv[r * 2 + 1] = s2;
      }
#pragma unroll
      for (int st = 0; st < 3; ++st) {
        const int half = 8 >> (st + 1);
        const bool up = (lane >> st) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < half) {
            const float keep = up ? sv[i + half] : sv[i];
            const float send = up ? sv[i] : sv[i + half];
            sv[i] = keep + __shfl_xor(send, 1 << st, 64);
          }
        }
      }
      sv[0] += __shfl_xor(sv[0], 8, 64);
      int idx = 0;
#pragma unroll
      for (int st = 0; st < 3; ++st) idx += ((lane >> st) & 1) * (8 >> (st + 1));
      float* red = smem;                // [4 waves][16][2]
      if (((lane & 15) >> 3) == 0) red[(wave * 16 + clane + (idx >> 1)) * 2 + (idx & 1)] = sv[0];
      __syncthreads();
      if (tid < 32 && (tid >> 1) < a.Cout)
        a.stats[((size_t)tile * a.Cout + (tid >> 1)) * 2 + (tid & 1)] = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
      __syncthreads();
    }
  }
}

// tiles a workgroup works through: doubled while every CU keeps >= 8 workgroups' worth of slots filled (2048 workgroups) and the tile
// count divides; WTPSE_C16P_TPW caps it (1: one tile per workgroup)
static int conv16p_tpw(int ntiles) {
  static const int cap = [] { const char* e = getenv("WTPSE_C16P_TPW"); const int v = e ? atoi(e) : 8; return (v >= 1 && v <= 64) ? v : 8; }();
  int tpw = 1;
  while (tpw * 2 <= cap && ntiles % (tpw * 2) == 0 && ntiles / (tpw * 2) >= 1024) tpw *= 2;
  return tpw;
}

// conv.hip: wtpse_conv16_x3's forward launches of the x2h arithmetic without mask / BatchNorm-backward epilogue / in-launch tail
int conv16p_launch(const float* in0, int C0, const unsigned short* wx16, const float* bias, const float* pro0, int pro_relu, float* out0,
                   float* stats, float* gram, int B, int H, int W, int Cout, int relu_out, const unsigned* in_amax, float in_scale,
                   unsigned* out_amax, hipStream_t st) {
  Conv16pArgs a;
  a.in = in0; a.wx16 = wx16; a.bias = bias; a.pro = pro0; a.out = out0; a.stats = stats; a.gram = gram; a.in_amax = in_amax;
  a.out_amax = out_amax; a.in_scale = in_scale; a.B = B; a.H = H; a.W = W; a.C0 = C0; a.Cout = Cout; a.pro_relu = pro_relu;
  a.relu_out = relu_out;
  a.tiles_x = ceil_div(W, 32);
  a.tiles_y = ceil_div(H, 8);
  const int ntiles = B * a.tiles_x * a.tiles_y;
  a.tpw = conv16p_tpw(ntiles);
  hipLaunchKernelGGL(conv16p_k, dim3((unsigned)(ntiles / a.tpw)), dim3(256), 0, st, a);
  return wtpse_status();
}
