// Shared definitions for the WT-PSE gfx950 kernels.  CDNA4 only: wave = 64 lanes, fp32-input MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WTPSE_OK 0
#define WTPSE_EINVAL (-1)

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
