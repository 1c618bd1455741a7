// Shared definitions for the WT-PSE gfx950 kernels.  CDNA4 only: wave = 64 lanes, fp32-input MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WTPSE_OK 0
#define WTPSE_EINVAL (-1)
#define WTPSE_ESTATE (-2)     /* wtpse_plan_replay: the library's run-time switches differ from the ones the plan was recorded under */

// Every launcher ends with this: launch errors (bad grid, missing code object) surface as a status code.
static inline int wtpse_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? WTPSE_OK : (int)e;
}

#define WTPSE_REQUIRE(cond) \
  do {                      \
    if (!(cond)) return WTPSE_EINVAL; \
  } while (0)

// D = A*B + C on the matrix cores, exact fp32 (k-ordered fma chain).
//   16x16x4 : lane l holds A[row l&15][k l>>4], B[k l>>4][col l&15]; D reg r -> row (l>>4)*4+r, col l&15
//   32x32x2 : lane l holds A[row l&31][k l>>5], B[k l>>5][col l&31]; D reg r -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_xor_sum(float v, int mask_hi) {
  // butterfly over lane-xor masks 1..mask_hi (mask_hi = 8 -> groups of 16 lanes, 16 -> 32, 32 -> 64)
  for (int m = 1; m <= mask_hi; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Raw buffer resource (stride 0) over `bytes` bytes at `p`: loads whose per-lane byte offset is >= bytes return 0
// in hardware, so zero padding / ragged tiles need no per-element branch; the per-channel plane offset rides in
// the scalar `soffset` operand (measured on gfx950 with tools/probe/bufload.hip: a lane is out of range when
// voffset >= num_records - soffset; keep soffset <= num_records; flags word 0x00020000).  BUF_OOB marks a lane as out of range.
#define BUF_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voffset, unsigned soffset) {   // 16-byte aligned
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voffset, (int)soffset, 0));
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voffset, unsigned soffset) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voffset, (int)soffset, 0));  // b32 = raw bits
}
// Out-of-range lanes (voffset = BUF_OOB) are dropped by the hardware: predicated stores without an exec-mask branch.
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned voffset, unsigned soffset, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)voffset, (int)soffset, 0);
}

// The output clamp of the convolutions' epilogues: max(v, lo), lo = 0 (output ReLU) or -inf.  v_max_f32 returns the OTHER operand when
// one is NaN — a NaN accumulator would be stored as -inf (no ReLU) or 0 (ReLU), where the reference's conv / ReLU store NaN
// (torch.relu(NaN) = NaN) and its `isnan` checks (shape_networks.py:490: the mu scrub) expect to find it.  The FORWARD launches (EPI 0)
// therefore clamp with a compare + select that keeps a NaN; the data gradients' epilogues (EPI 1 / 2) keep the one-instruction form.
template <int EPI>
__device__ __forceinline__ float out_clamp(float v, float lo) {
  if constexpr (EPI == 0) return v < lo ? lo : v;
  else return fmaxf(v, lo);
}

// ---- split-bf16 ("x3") helpers shared by conv.hip's 16-channel path (conv_x3.hip / wgrad_r.hip carry their own copies)
typedef __bf16 wt_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wt_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned wt_pack_rne(float a, float b) {
  wt_bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// (a, b) -> three dwords holding the bf16 pairs (term_i(a), term_i(b)), each term rounded to nearest even, remainders exact
__device__ __forceinline__ void wt_split3_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = wt_pack_rne(a, b);
  const float ra = a - __builtin_bit_cast(float, p0 << 16);
  const float rb = b - __builtin_bit_cast(float, p0 & 0xFFFF0000u);
  p1 = wt_pack_rne(ra, rb);
  const float sa = ra - __builtin_bit_cast(float, p1 << 16);
  const float sb = rb - __builtin_bit_cast(float, p1 & 0xFFFF0000u);
  p2 = wt_pack_rne(sa, sb);
}
__device__ __forceinline__ f32x4 wt_mfma16x32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wt_bf16x8, a), __builtin_bit_cast(wt_bf16x8, b), c, 0, 0, 0);
}

// ---- "x2h" helpers (two fp16 terms per fp32 operand, three products: conv_x3_kernels.h has the full description) shared by the
// convolutions and the register-resident weight gradient
constexpr int X3_WHDR = 32;                // unsigned shorts of header in front of a packed block: float {1 / scale, scale, 0, 0, 12 slice maxima of |w|}
constexpr int X3_WSLICES = 12;
constexpr float X3_FWD_SCALE = 4.f;     // the fallback input scale of a forward activation whose bound nobody supplied (C-ABI callers: in_amax == null)

// The largest magnitude of a gradient tensor travels as an "amax table": AMAX_SHARDS unsigneds (float bits of non-negative values:
// they order like their bit patterns), one per 64-byte line, zero before the tensor's producer runs.  Producers fold a wave's (or a
// workgroup's) maximum into shard (index % AMAX_SHARDS) with one NO-RETURN atomic max — fire and forget: nothing waits for it — and
// consumers take the maximum of the shards.  64 shards: the largest producer launch (65 536 workgroups of four waves in ~140 us) sends
// a shard one atomic per ~35 ns, above the ~12 ns one word takes at the memory side (MI355X_MICROARCH.md, fan-in); on ONE word the
// atomics of a launch would queue for milliseconds.
constexpr int AMAX_SHARDS = 64, AMAX_STRIDE = 16, AMAX_WORDS = AMAX_SHARDS * AMAX_STRIDE;
__device__ __forceinline__ unsigned amax_bits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7FFFFFFFu; }
// Maximum over the 64 lanes of a FULL wave, wave-uniform (in a scalar register).  max is idempotent, so four DPP steps that each
// fold a lane with a partner — xor 1 and xor 2 within a quad (quad_perm), the other quad of an 8-lane half (row_half_mirror), the
// other half of the 16-lane row (row_mirror) — leave every lane of a row with the row's maximum, and four v_readlane + three s_max
// finish it: ~12 cheap instructions where the xor butterfly of rounds 1-5 took six ds_bpermute round trips through the LDS crossbar
// (each behind its own lgkmcnt wait) — in front of EVERY workgroup's first conversion in the x2h kernels, and behind every wave's
// stores in the producers that publish an amax.
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));     // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));     // row_mirror
  const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(r0, r1), max(r2, r3));
}
__device__ __forceinline__ unsigned amax_bits4(f32x4 v) {
  return max(max(amax_bits(v[0]), amax_bits(v[1])), max(amax_bits(v[2]), amax_bits(v[3])));
}
// every thread of the workgroup calls (blockDim.x a multiple of 64, at most 1024); m: the thread's own maximum
__device__ __forceinline__ void amax_publish_block(unsigned* table, unsigned m, unsigned shard_seed) {
  __shared__ unsigned amax_red[16];
  m = wave_umax(m);
  const int nw = (int)(blockDim.x >> 6);
  if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < nw; ++i) m = max(m, amax_red[i]);
    if (m) (void)__hip_atomic_fetch_max(table + (shard_seed % AMAX_SHARDS) * AMAX_STRIDE, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// every lane of a full wave calls (no barrier, no LDS: for streaming kernels that run one small item per workgroup — a persistent,
// one-atomic-per-workgroup form of the BatchNorm-backward apply pass measured 14-29 % slower than this)
__device__ __forceinline__ void amax_publish_wave(unsigned* table, unsigned m, unsigned shard_seed) {
  m = wave_umax(m);
  if ((threadIdx.x & 63) == 0 && m)
    (void)__hip_atomic_fetch_max(table + (shard_seed % AMAX_SHARDS) * AMAX_STRIDE, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// consumer side (any full wave): the maximum over the shards, wave-uniform.  In two halves, so that a kernel can ISSUE the read of the
// table in front of its first tile loads and pick the value up behind them (vmcnt completes in order: the table's round trip then
// hides behind loads the workgroup waits for anyway, instead of standing in front of them or at their end): amax_load() = this lane's
// shard (0 for a null table), amax_reduce() = the maximum over the lanes.
__device__ __forceinline__ unsigned amax_load(const unsigned* table) {
  return table ? table[(threadIdx.x & (AMAX_SHARDS - 1)) * AMAX_STRIDE] : 0u;
}
__device__ __forceinline__ unsigned amax_reduce(unsigned v) {
  static_assert(AMAX_SHARDS == 64, "one shard per lane");
  return wave_umax(v);
}
__device__ __forceinline__ unsigned amax_read(const unsigned* table) {
  unsigned v = table[(threadIdx.x & (AMAX_SHARDS - 1)) * AMAX_STRIDE];
  return amax_reduce(v);
}

// power of two S with amax * S in [2^14, 2^15) (amax: float bits of a non-negative value); 1 for 0 / denormal / inf / nan
__device__ __host__ __forceinline__ float x3_scale_from_amax(unsigned bits) {
  const int e = (int)((bits >> 23) & 0xFFu);
  if (e == 0 || e == 255) return 1.f;
  int se = 268 - e;                         // biased exponent of 2^(14 - (e - 127))
  se = se > 253 ? 253 : se;
  const unsigned sb = (unsigned)se << 23;
  return __builtin_bit_cast(float, sb);
}

// ---- scale of a FORWARD activation (round 6).  An x2h consumer multiplies its input by the power of two that brings a BOUND of the
// tensor's largest magnitude into [2^14, 2^15); the bound travels in an amax table like a gradient's amax does, but nobody has to
// look at the data for it where a train-mode BatchNorm produced the tensor: by Samuelson's inequality every sample of a population
// of N values lies within sqrt(N - 1) (biased) standard deviations of its mean, so |gamma xhat + beta| <= |gamma| sqrt(N - 1) + |beta|
// whatever the data (ReLU only lowers it).  For N = 2 M pixels the bound is 2^10.5 |gamma|: an O(gamma) activation then sits at 2^3.5
// after scaling — full 22-bit precision down to 2^-6.5 |gamma|, absolute error 2^-28.5 |gamma| below, at ANY gamma (the fixed 2^2 of
// round 5 gave exactly this for gamma = 1 and lost relative precision below, NaN above 2^14).  Tensors without a BatchNorm of their own
// batch (DeepWT's maps, the fusion conv, eval-mode BatchNorm) get the amax of the stored data from their producer's epilogue instead.
__device__ __forceinline__ float bn_act_bound(float gamma, float beta, double count) {
  return fabsf(gamma) * (float)sqrt(count > 2.0 ? count - 1.0 : 1.0) + fabsf(beta);
}
// one value into an amax table (zero on entry), from any thread: no-return atomic max on shard (seed % AMAX_SHARDS)
__device__ __forceinline__ void amax_put(unsigned* table, float v, unsigned seed) {
  const unsigned b = amax_bits(v);
  if (b) (void)__hip_atomic_fetch_max(table + (seed % AMAX_SHARDS) * AMAX_STRIDE, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_h_rne(float a, float b) {
  h16x2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
// (a, b) -> two dwords holding the fp16 pairs (h0(a), h0(b)), (h1(a), h1(b)).  No clamp: a value beyond +-65504 (after scaling) becomes
// inf, its remainder -inf, and every product it feeds NaN — an out-of-range operand fails LOUDLY (the caller's NaN check fires, as it does
// for any other divergence) instead of turning into a wrong finite number; a NaN operand stays NaN, as in fp32.  (A saturating form —
// v_med3_f32 per element — swallowed NaNs: v_med3 returns the minimum when an input is NaN.)
// Three instructions per pair: v_cvt_pk_f16_f32, then the remainders with the mixed-precision FMA — fma(h0 [fp16 half of p0], -1, a [fp32])
// rounded to fp16 into the low / high half of p1.  a - h0 is exact in fp32, so this is bit for bit what the compiler's eight-instruction
// rendering of `pack(a - float(h0(a)), b - float(h0(b)))` gave (two cvt_f16, two cvt_f32, cvt_pk, two sub, cvt_pk; it does not select
// v_fma_mix for that by itself): tools/probe/mixsplit_test.hip compares the two over 4 M operands incl. inf / NaN / overflow / denormals.
// NOTE: the hazard recogniser does not look inside inline assembly.  Use this form where the result goes to LDS or is consumed a
// long way off (the convolutions' loaders, the weight gradient's conversion pieces); where a matrix instruction reads the result within
// a few instructions (csrc/head.hip) use split2h_pair_c — there the asm form produced run-to-run differences of a remainder term's size.
__device__ __forceinline__ void split2h_pair_c(float a, float b, unsigned& p0, unsigned& p1) {      // the same values, compiler-scheduled
  const h16x2 v = {(_Float16)a, (_Float16)b};
  p0 = __builtin_bit_cast(unsigned, v);
  p1 = pack_h_rne(a - (float)v[0], b - (float)v[1]);
}
__device__ __forceinline__ void split2h_pair(float a, float b, unsigned& p0, unsigned& p1) {
  p0 = pack_h_rne(a, b);
  unsigned r;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p0), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(p0), "v"(b));
  p1 = r;
}

__device__ __forceinline__ f32x4 wt_mfma16x32h(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// ---- BatchNorm-backward coefficients from the epilogue of the data gradient that produced the statistics (conv.hip, conv_x3.hip:
// EPI 2).  Every workgroup has written its (sum g, sum g (y - mean)) partials to stats[tile][Cbn][2]; instead of a separate
// finalize launch (a 16..256-workgroup kernel that sat between two big launches of a dependency chain, 142 times per step, and
// waited 50 us on average for a free CU beside the weight gradient on the other stream), the partials are folded by whoever
// arrives last: groups of 64 consecutive tiles take a ticket, the last arriver of a group folds the group in index order
// (fp64) into partial2, then the groups of one output-channel block take a second ticket and the last of them folds the
// group sums — again in index order, so the result does not depend on who arrives when — and writes (k1, k2, k3), dgamma,
// dbeta exactly as bn_bwd_finalize_k does.  Nobody waits for anybody: no spinning.  Tickets are zero on entry and are left zero.
struct BnbTail {
  double* partial2;        // [ngroups][ctot][2]
  unsigned* tickets;       // [t2_off + blocks along y]; null: no tail (stats only)
  const float* gamma;      // of the BatchNorm'd channels (index c - bn_c0), as invstd / mean
  const float* invstd;
  float* coef;             // [Cbn][3]
  float* dgamma;
  float* dbeta;
  double count;
  int accumulate, ngroups, ntiles, ctot, t2_off;
};

// Hand-off between workgroups without cache-wide fences (MI355X_MICROARCH.md, inter-workgroup visibility): the producer
// publishes with agent-scope (write-through) stores and drains them (s_waitcnt vmcnt(0)) before it takes its ticket with a
// relaxed agent-scope atomic; the consumer reads with agent-scope loads.  (An agent-scope release fence writes back the whole
// L2 of the XCD: thousands of those beside a kernel that is writing its output are not an option.)
__device__ __forceinline__ void pub_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pub_store(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float pub_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double pub_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void pub_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The two-level fold, in two halves so that the round trip of the first ticket (an L2 atomic: ~1-2 us, during which a workgroup
// that waited for it right away held its CU slot idle — 9-19 % of the short 32-channel launches) hides behind the workgroup's output
// stores: tail_begin() publishes the partials and ISSUES the ticket, the caller stores its tile, tail_fold() picks the ticket up.
// The caller has written this workgroup's partials stats[tile][Cst][2] (channels [c_lo, c_hi) of the launch's output,
// Cst = c_hi - c_lo) with pub_store().
struct TailTicket {
  unsigned old;      // thread 0: what the group's ticket counter held
  int armed;         // uniform: this workgroup takes part
};

__device__ __forceinline__ TailTicket tail_begin(unsigned* tickets, int ngroups, int tile, int y, int tid) {
  TailTicket tk;
  tk.old = 0u;
  tk.armed = 1;
  pub_drain();
  __syncthreads();
  if (tid == 0)
    tk.old = __hip_atomic_fetch_add(tickets + (size_t)y * ngroups + (tile >> 6), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return tk;
}

// -> true in the ONE workgroup of output-channel block `y` that arrives last; there sh[2 * crel + k] holds the fp64 totals of the
// block's channels (threads tid < 2 CB: crel = tid >> 1, k = tid & 1).  sh: 256 doubles of LDS scratch.
// Round 6: both folds run with ALL their loads in flight at once.  The fold is the tail of the launch's critical path — nothing of
// the launch is left to hide it behind — and rounds 3-5 walked a group's 64 partials in eight dependent rounds of eight agent-scope
// loads on 2 CB of the 256 threads (the step-form launches measured 6-7 us longer than the plain ones: tools/probe/instep_gap.py).
// Now the 256 threads form 256 / (2 CB) groups, each group takes a contiguous share of the 64 tiles (or of the group sums) with one
// load per tile, all issued before the first is used; the groups' sums meet in LDS and are added in index order: still a fixed order,
// whoever arrives when.
template <int CB>
__device__ __forceinline__ bool tail_fold(const TailTicket& tk, double* partial2, unsigned* tickets, int ngroups, int ntiles, int ctot,
                                          int t2_off, const float* stats, int c_lo, int c_hi, int cout0, int tile, int y, int tid,
                                          double* sh, int* flag_s) {
  static_assert(CB * 2 <= 256 && 256 % (CB * 2) == 0, "one thread per (channel, sum) and share of the tiles");
  constexpr int NP = 2 * CB;          // (channel, sum) pairs of the block
  constexpr int NQ = 256 / NP;        // thread groups
  constexpr int TPQ = 64 / NQ;        // tiles of a group of 64 per thread group
  const int Cst = c_hi - c_lo;
  const int grp = tile >> 6, g0 = grp << 6;
  const int gsize = min(64, ntiles - g0);
  if (tid == 0) *flag_s = tk.old == (unsigned)(gsize - 1);
  __syncthreads();
  if (!*flag_s) return false;
  const int pr = tid % NP, q = tid / NP;
  const int crel = pr >> 1, k = pr & 1, c = cout0 + crel;
  const bool mine = c >= c_lo && c < c_hi;
  {
    double s = 0.0;
    if (mine) {
      const float* p = stats + ((size_t)g0 * Cst + (c - c_lo)) * 2 + k;
      // (unconditional loads at a clamped index, the tail of a short group dropped afterwards: a load inside a conditional comes
      // out as a branch with its own s_waitcnt vmcnt(0) — 32 serialised round trips, measured slower than rounds 3-5's eight)
      float v[TPQ];
#pragma unroll
      for (int u = 0; u < TPQ; ++u) v[u] = pub_load(p + (size_t)min(q * TPQ + u, gsize - 1) * Cst * 2);
#pragma unroll
      for (int u = 0; u < TPQ; ++u) s += (q * TPQ + u < gsize) ? (double)v[u] : 0.0;
    }
    sh[q * NP + pr] = s;
    __syncthreads();
    if (q == 0 && mine) {
      double t = sh[pr];
#pragma unroll
      for (int qq = 1; qq < NQ; ++qq) t += sh[qq * NP + pr];
      pub_store(partial2 + ((size_t)grp * ctot + c) * 2 + k, t);
    }
    if (tid == 0) __hip_atomic_store(tickets + (size_t)y * ngroups + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  pub_drain();
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(tickets + t2_off + y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag_s = old == (unsigned)(ngroups - 1);
  }
  __syncthreads();
  if (!*flag_s) return false;
  double s = 0.0;
  if (mine) {
    const double* p = partial2 + (size_t)c * 2 + k;
    const int per = (ngroups + NQ - 1) / NQ, i0 = q * per, i1 = min(ngroups, i0 + per);       // a contiguous share of the group sums
    int i = i0;
    for (; i + 8 <= i1; i += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pub_load(p + (size_t)(i + u) * ctot * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < i1; ++i) s += pub_load(p + (size_t)i * ctot * 2);
  }
  __syncthreads();            // (every group is done reading the first fold's sh)
  sh[q * NP + pr] = s;
  __syncthreads();
  if (q == 0) {
    double t = sh[pr];
#pragma unroll
    for (int qq = 1; qq < NQ; ++qq) t += sh[qq * NP + pr];
    sh[pr] = t;
  }
  if (tid == 0) __hip_atomic_store(tickets + t2_off + y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return true;
}

template <int CB>
__device__ __forceinline__ TailTicket bnb_tail_begin(const BnbTail& tl, int bn_c0, int bn_c1, int cout0, int tile, int y, int tid) {
  TailTicket none;
  none.old = 0u;
  none.armed = 0;
  if (tl.tickets == nullptr) return none;
  if (cout0 >= bn_c1 || cout0 + CB <= bn_c0) return none;      // no BatchNorm'd channel in this block (uniform)
  return tail_begin(tl.tickets, tl.ngroups, tile, y, tid);
}

template <int CB>
__device__ __forceinline__ void bnb_tail(const TailTicket& tk, const BnbTail& tl, const float* stats, const float* __restrict__ bn_mean,
                                         int bn_c0, int bn_c1, int cout0, int tile, int y, int tid, double* sh, int* flag_s) {
  if (!tk.armed) return;
  if (!tail_fold<CB>(tk, tl.partial2, tl.tickets, tl.ngroups, tl.ntiles, tl.ctot, tl.t2_off, stats, bn_c0, bn_c1, cout0, tile, y, tid,
                     sh, flag_s))
    return;
  const int c = cout0 + (tid >> 1);
  if (tid < CB * 2 && (tid & 1) == 0 && c >= bn_c0 && c < bn_c1) {
    const int cb = c - bn_c0;
    const double is = (double)tl.invstd[cb], ga = (double)tl.gamma[cb];
    const double s1 = sh[tid], s2 = sh[tid + 1] * is;          // the partials hold sum g (y - mean): not yet / std
    tl.dbeta[cb] = tl.accumulate ? tl.dbeta[cb] + (float)s1 : (float)s1;
    tl.dgamma[cb] = tl.accumulate ? tl.dgamma[cb] + (float)s2 : (float)s2;
    const double k1 = ga * is;
    const double k2 = -ga * is * is * s2 / tl.count;
    const double k3 = -k1 * s1 / tl.count - k2 * (double)bn_mean[cb];
    tl.coef[3 * cb] = (float)k1;
    tl.coef[3 * cb + 1] = (float)k2;
    tl.coef[3 * cb + 2] = (float)k3;
  }
}

// ---- the same for the FORWARD statistics: the convolution that forms the BatchNorm (sum, sum^2) partials of its output
// (stats[tile][Cout][2]) also finishes them — scale/shift, saved mean / invstd, running statistics: what bn_finalize_k does
// (reference: nn.BatchNorm2d in train mode, algorithms.py:883-889) — 198 dependent launches of ~6 us per step otherwise.
struct BnfTail {
  double* partial2;        // [ngroups][ctot][2]
  unsigned* tickets;       // null: no tail
  const float* gamma;
  const float* beta;
  float* rmean;            // running statistics, or null
  float* rvar;
  long long* nbt;          // num_batches_tracked, or null
  float* scale_shift;      // [Cout][2]
  float* save_mean;
  float* save_invstd;
  unsigned* act_amax;      // amax table (zero on entry) that receives the bound of |BatchNorm output| (bn_act_bound), or null
  double count;
  float momentum, eps;
  int ngroups, ntiles, ctot, t2_off;
};

__device__ __forceinline__ TailTicket bnf_tail_begin(const BnfTail& tl, int tile, int y, int tid) {
  TailTicket none;
  none.old = 0u;
  none.armed = 0;
  if (tl.tickets == nullptr) return none;
  return tail_begin(tl.tickets, tl.ngroups, tile, y, tid);
}

template <int CB>
__device__ __forceinline__ void bnf_tail(const TailTicket& tk, const BnfTail& tl, const float* stats, int Cout, int cout0, int tile,
                                         int y, int tid, double* sh, int* flag_s) {
  if (!tk.armed) return;
  if (!tail_fold<CB>(tk, tl.partial2, tl.tickets, tl.ngroups, tl.ntiles, tl.ctot, tl.t2_off, stats, 0, Cout, cout0, tile, y, tid, sh,
                     flag_s))
    return;
  const int c = cout0 + (tid >> 1);
  if (tid < CB * 2 && (tid & 1) == 0 && c < Cout) {
    const double mean = sh[tid] / tl.count;
    double var = sh[tid + 1] / tl.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)tl.eps);
    tl.scale_shift[2 * c] = (float)(tl.gamma[c] * invstd);
    tl.scale_shift[2 * c + 1] = (float)(tl.beta[c] - mean * tl.gamma[c] * invstd);
    tl.save_mean[c] = (float)mean;
    tl.save_invstd[c] = (float)invstd;
    if (tl.act_amax) amax_put(tl.act_amax, bn_act_bound(tl.gamma[c], tl.beta[c], tl.count), (unsigned)c);
    if (tl.rmean) {
      const double unbiased = tl.count > 1.0 ? var * tl.count / (tl.count - 1.0) : var;
      tl.rmean[c] = (float)((1.0 - tl.momentum) * tl.rmean[c] + tl.momentum * mean);
      tl.rvar[c] = (float)((1.0 - tl.momentum) * tl.rvar[c] + tl.momentum * unbiased);
    }
    if (tl.nbt && c == 0) *tl.nbt += 1;
  }
}

// Host side: the caller-provided part of a tail (null tickets: none) completed with the launch's geometry.  partial2 must hold
// wtpse_bnb_tail_partial2(nblk, Cout) doubles, tickets wtpse_bnb_tail_tickets(nblk, Cout) zeroed unsigneds (include/wtpse_hip.h).
static inline int bnb_tail_groups(int ntiles) { return (ntiles + 63) / 64; }
static inline int bnb_tail_ctot(int Cout) { return (Cout + 63) & ~63; }
static inline int bnb_tail_t2off(int ntiles, int Cout) { return ((Cout + 15) / 16) * bnb_tail_groups(ntiles); }
static inline BnbTail bnb_tail_none() {
  BnbTail t;
  t.partial2 = nullptr; t.tickets = nullptr; t.gamma = nullptr; t.invstd = nullptr; t.coef = nullptr; t.dgamma = nullptr; t.dbeta = nullptr;
  t.count = 1.0; t.accumulate = 0; t.ngroups = 0; t.ntiles = 0; t.ctot = 0; t.t2_off = 0;
  return t;
}
static inline void bnb_tail_geometry(BnbTail& t, int ntiles, int Cout, double count) {
  t.ntiles = ntiles; t.ngroups = bnb_tail_groups(ntiles); t.ctot = bnb_tail_ctot(Cout); t.t2_off = bnb_tail_t2off(ntiles, Cout);
  t.count = count;
}
static inline BnfTail bnf_tail_none() {
  BnfTail t;
  t.partial2 = nullptr; t.tickets = nullptr; t.gamma = nullptr; t.beta = nullptr; t.rmean = nullptr; t.rvar = nullptr; t.nbt = nullptr;
  t.scale_shift = nullptr; t.save_mean = nullptr; t.save_invstd = nullptr; t.act_amax = nullptr; t.count = 1.0; t.momentum = 0.f; t.eps = 0.f;
  t.ngroups = 0; t.ntiles = 0; t.ctot = 0; t.t2_off = 0;
  return t;
}
static inline void bnf_tail_geometry(BnfTail& t, int ntiles, int Cout, double count) {
  t.ntiles = ntiles; t.ngroups = bnb_tail_groups(ntiles); t.ctot = bnb_tail_ctot(Cout); t.t2_off = bnb_tail_t2off(ntiles, Cout);
  t.count = count;
}

// Folding the statistics inside the launch makes every workgroup live ~2.5 us longer (its partials must be visible before it takes
// its ticket: store acknowledgement + one L2 atomic round trip).  Measured on the step (back to back, profiles/r03_*): with the
// fold in every launch the convolutions took 2.4 ms more per step than the 346 finalize launches it replaced took (2.2 ms).  A CU
// slot sees nWG / (256 x 2..3) workgroups in a row, so the hand-off wins where that is about one or less; beyond the threshold the
// entry points launch the stand-alone finalize kernel themselves.
#include <cstdlib>
// (Round 6: with the fold's loads all in flight at once — tail_fold — the hand-off is cheaper than the stand-alone kernel up to the
// 8192-workgroup launches of the step as well: 41.54 vs 41.66 ms per step, three alternations on one box; the threshold moves there.)
static inline bool tail_in_launch(long long workgroups) {
  static const long long max_wgs = [] { const char* e = getenv("WTPSE_TAIL_MAX_WGS"); return e ? atoll(e) : 8192ll; }();
  return workgroups <= max_wgs;
}
extern "C" int wtpse_bn_finalize(const float* stats_partial, int nblk, int C, long long count, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, long long* num_batches, float momentum, float eps,
                                 float* scale_shift, float* save_mean, float* save_invstd, unsigned* act_amax, void* stream);
extern "C" int wtpse_bn_bwd_finalize_coef(const float* stats_partial, int nblk, int C, long long count, const float* gamma,
                                          const float* save_mean, const float* save_invstd, float* coef, float* dgamma,
                                          float* dbeta, int accumulate, void* stream);
// after a launch whose tails were switched off for size: the same results from the stand-alone kernels
static inline int tail_after_launch(const BnbTail& tl, const BnfTail& fl, float* stats, int nblk, int Cout, int bn_c0, int bn_c1,
                                    const float* bn_mean, long long count, void* stream) {
  if (tl.tickets)
    return wtpse_bn_bwd_finalize_coef(stats, nblk, bn_c1 - bn_c0, count, tl.gamma, bn_mean, tl.invstd, tl.coef, tl.dgamma, tl.dbeta,
                                      tl.accumulate, stream);
  if (fl.tickets)
    return wtpse_bn_finalize(stats, nblk, Cout, count, fl.gamma, fl.beta, fl.rmean, fl.rvar, fl.nbt, fl.momentum, fl.eps,
                             fl.scale_shift, fl.save_mean, fl.save_invstd, fl.act_amax, stream);
  return 0;
}
