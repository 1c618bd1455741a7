#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_h_rne(float a, float b) { h16x2 v = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ void split_old(float a, float b, unsigned& p0, unsigned& p1) {
  const h16x2 v = {(_Float16)a, (_Float16)b};
  p0 = __builtin_bit_cast(unsigned, v);
  p1 = pack_h_rne(a - (float)v[0], b - (float)v[1]);
}
__device__ __forceinline__ void split_new(float a, float b, unsigned& p0, unsigned& p1) {
  const h16x2 v = {(_Float16)a, (_Float16)b};
  p0 = __builtin_bit_cast(unsigned, v);
  unsigned r;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p0), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(p0), "v"(b));
  p1 = r;
}
__global__ void k(const float* x, unsigned* o0, unsigned* o1, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 2 + 1 >= n) return;
  float a = x[i * 2], b = x[i * 2 + 1];
  unsigned p0, p1, q0, q1;
  split_old(a, b, p0, p1);
  split_new(a, b, q0, q1);
  o0[i * 2] = p0; o0[i * 2 + 1] = p1; o1[i * 2] = q0; o1[i * 2 + 1] = q1;
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  unsigned s = 12345u;
  for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; unsigned e = 100u + (s >> 8) % 60u; unsigned bits = (s & 0x807FFFFFu) | (e << 23); std::memcpy(&h[i], &bits, 4); }
  h[0] = 0.f; h[1] = -0.f; h[2] = 65504.f; h[3] = 1e-8f; h[4] = INFINITY; h[5] = NAN; h[6] = 70000.f; h[7] = 6e-5f;
  float* dx; unsigned *d0, *d1;
  hipMalloc(&dx, n * 4); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, d0, d1, n);
  std::vector<unsigned> a(n), b(n);
  hipMemcpy(a.data(), d0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n; ++i) if (a[i] != b[i]) { if (bad < 10) printf("diff at %d: x=%g,%g old %08x new %08x\n", i, h[i & ~1], h[i | 1], a[i], b[i]); ++bad; }
  printf("elements %d, differing words %ld\n", n, bad);
  return 0;
}
