// Micro-benchmark: do VALU instructions of one wave run in the shadow of another wave's MFMA stream on the same SIMD?
// One workgroup of 512 threads per CU = 2 waves per SIMD.  Waves 0-3 issue NM dependent-free v_mfma_f32_32x32x16_bf16
// (4 accumulators round-robin), waves 4-7 issue NV v_fma_f32 (8 independent chains) or NV v_cvt_pk_bf16_f32 / ds_write_b128.
// Times (s_memtime, wave 0 / wave 4 of each workgroup): MFMA alone, VALU alone, both together.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu.hip -o tools/probe/_build/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // bit 0: MFMA waves work, bit 1: VALU waves work; VK: 0 fma, 1 cvt_pk, 2 ds_write_b128
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* t, int nm, int nv, int vk) {
  __shared__ float4 lds[2048];
  const int wave = threadIdx.x >> 6;
  const bool is_m = wave < 4;
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if (is_m) {
    if (MODE & 1) {
      f32x16 c[4];
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < nm; i += 4) {
        c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[0], 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[1], 0, 0, 0);
        c[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[2], 0, 0, 0);
        c[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[3], 0, 0, 0);
      }
      float s = 0.f;
      for (int i = 0; i < 4; ++i) s += c[i][0];
      t1 = __builtin_amdgcn_s_memtime();
      if (s == 12345.f) out[threadIdx.x] = s;
    }
  } else if (MODE & 2) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    t0 = __builtin_amdgcn_s_memtime();
    if (vk == 0) {
      for (int i = 0; i < nv; i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
    } else if (vk == 1) {
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      for (int i = 0; i < nv; i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bf16x2 p = {(__bf16)v[j], (__bf16)v[(j + 1) & 7]};
          v[j] += __builtin_bit_cast(float, p);
        }
    } else {
      for (int i = 0; i < nv; i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) lds[(threadIdx.x & 255) + 256 * j] = make_float4(v[0], v[1], v[2], v[3]);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.f) out[threadIdx.x] = s + lds[threadIdx.x].x;
  }
  if ((threadIdx.x & 255) == 0) t[blockIdx.x * 2 + (threadIdx.x >> 8)] = t1 - t0;
}

int main() {
  float* out; unsigned long long* t;
  hipMalloc(&out, 4096); hipMalloc(&t, 256 * 2 * 8);
  const int nm = 2048;
  const char* names[3] = {"v_fma_f32", "v_cvt_pk_bf16_f32 + v_add", "ds_write_b128"};
  for (int vk = 0; vk < 3; ++vk) {
    const int nv = vk == 2 ? 1024 : 8192;
    double res[3][2];
    for (int mode = 1; mode <= 3; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, t, nm, nv, vk);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, t, nm, nv, vk);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, t, nm, nv, vk);
      }
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(512);
      hipMemcpy(h.data(), t, 512 * 8, hipMemcpyDeviceToHost);
      std::vector<double> m, v;
      for (int i = 0; i < 256; ++i) { m.push_back((double)h[2 * i]); v.push_back((double)h[2 * i + 1]); }
      std::sort(m.begin(), m.end()); std::sort(v.begin(), v.end());
      res[mode - 1][0] = m[128]; res[mode - 1][1] = v[128];
    }
    printf("%-28s: %d MFMAs alone %.0f cycles (%.1f/MFMA) | %d %s alone %.0f cycles (%.1f each) | together: MFMA wave %.0f (%.1f/MFMA), other wave %.0f (%.1f each)\n",
           names[vk], nm, res[0][0], res[0][0] / nm, nv, vk == 2 ? "stores" : "VALU ops", res[1][1], res[1][1] / nv / (vk == 1 ? 2 : 1), res[2][0], res[2][0] / nm, res[2][1],
           res[2][1] / nv / (vk == 1 ? 2 : 1));
  }
  return 0;
}
