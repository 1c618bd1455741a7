// Probe: semantics of raw buffer loads on gfx950 (flags word, range check with voffset / soffset).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* a, float* c, int n, unsigned flags) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, n * 4, flags);
  int l = threadIdx.x;
  c[l] = __builtin_amdgcn_raw_buffer_load_b32(rs, l * 4, 0, 0);                 // plain
  c[64 + l] = __builtin_amdgcn_raw_buffer_load_b32(rs, l * 4, 64 * 4, 0);       // soffset = 64 elements
  c[128 + l] = __builtin_amdgcn_raw_buffer_load_b32(rs, (l & 1) ? 0x80000000 : l * 4, 0, 0);  // OOB marker on odd lanes
  c[192 + l] = __builtin_amdgcn_raw_buffer_load_b32(rs, l * 4, (n - 32) * 4, 0);  // voffset+soffset crosses the end on lanes >= 32
}
int main() {
  const int n = 256;
  float h[n], *a, *c, out[256];
  for (int i = 0; i < n; ++i) h[i] = 1000.f + i;
  hipMalloc(&a, 2 * n * 4); hipMalloc(&c, 256 * 4);
  hipMemset(a, 0, 2 * n * 4);
  hipMemcpy(a, h, n * 4, hipMemcpyHostToDevice);
  float tail[n]; for (int i = 0; i < n; ++i) tail[i] = -7.f;
  hipMemcpy(a + n, tail, n * 4, hipMemcpyHostToDevice);      // memory right behind the buffer
  unsigned flagsv[3] = {0x00020000u, 0x00027000u, 0x0u};
  for (int f = 0; f < 3; ++f) {
    hipMemset(c, 0xff, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, c, n, flagsv[f]);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(out, c, 256 * 4, hipMemcpyDeviceToHost);
    printf("flags %08x err %d | plain[0,1,63] %g %g %g | soff64[0,63] %g %g | oob[0,1,2,3] %g %g %g %g | cross[30,31,32,33] %g %g %g %g\n",
           flagsv[f], (int)e, out[0], out[1], out[63], out[64], out[127], out[128], out[129], out[130], out[131], out[192 + 30],
           out[192 + 31], out[192 + 32], out[192 + 33]);
  }
  return 0;
}
