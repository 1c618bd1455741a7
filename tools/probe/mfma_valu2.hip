// Micro-benchmark: VALU instructions in the shadow of the SAME wave's MFMAs.  One wave per SIMD (256-thread workgroups, one per
// CU) issues 2048 v_mfma_f32_32x32x16_bf16 (4 accumulators round-robin), each followed by K independent v_fma_f32 /
// v_cvt_pk_bf16_f32 / ds_write_b128 / ds_read_b128; prints cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu2.hip -o tools/probe/_build/mfma_valu2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(float* out, unsigned long long* t, int nm) {
  __shared__ float4 lds[64 * WAVES * 8];
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  float4 w = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < nm; i += 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[q], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int jj = (q * K + j) & 7;
        if (KIND == 0) v[jj] = fmaf(v[jj], 1.0001f, 0.5f);
        else if (KIND == 1) { bf16x2 p = {(__bf16)v[jj], (__bf16)v[(jj + 1) & 7]}; v[jj] = __builtin_bit_cast(float, p); }
        else if (KIND == 2) { lds[threadIdx.x + 64 * WAVES * jj] = w; }
        else { float4 r = lds[threadIdx.x + 64 * WAVES * jj]; v[jj] += r.x; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += c[i][0];
  for (int i = 0; i < 8; ++i) s += v[i];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (s == 12345.f) out[threadIdx.x] = s + lds[threadIdx.x].x;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int K, int KIND, int WAVES>
double run(float* out, unsigned long long* t) {
  const int nm = 2048;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<K, KIND, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, t, nm);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), t, 256 * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[128] / nm;
}

int main() {
  float* out; unsigned long long* t;
  hipMalloc(&out, 4096); hipMalloc(&t, 256 * 8);
  const char* names[4] = {"v_fma_f32", "v_cvt_pk_bf16_f32", "ds_write_b128", "ds_read_b128 + v_add"};
  printf("cycles per v_mfma_f32_32x32x16_bf16 with K other instructions of the same wave behind each (one wave per SIMD | two waves per SIMD):\n");
#define ROW(KIND) printf("  %-22s K=0 %5.1f | %5.1f   K=1 %5.1f | %5.1f   K=2 %5.1f | %5.1f   K=4 %5.1f | %5.1f   K=6 %5.1f | %5.1f   K=8 %5.1f | %5.1f\n", names[KIND], \
    run<0, KIND, 4>(out, t), run<0, KIND, 8>(out, t), run<1, KIND, 4>(out, t), run<1, KIND, 8>(out, t), run<2, KIND, 4>(out, t), run<2, KIND, 8>(out, t), \
    run<4, KIND, 4>(out, t), run<4, KIND, 8>(out, t), run<6, KIND, 4>(out, t), run<6, KIND, 8>(out, t), run<8, KIND, 4>(out, t), run<8, KIND, 8>(out, t));
  ROW(0) ROW(1) ROW(2) ROW(3)
  return 0;
}
