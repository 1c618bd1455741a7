// Probe: raw buffer load with LDS destination (LDS-DMA) on gfx950: placement (wave-uniform base + lane*4) and what
// out-of-range lanes write.    hipcc --offload-arch=gfx950 -O2 tools/probe/ldsdma.hip -o tools/probe/_build/ldsdma
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k(const float* src, unsigned bytes, float* out) {
  __shared__ float lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)bytes, 0x00020000);
  const int lane = threadIdx.x;
  // lanes 0..59 read element 2*lane (a gather), lanes 60..63 are out of range
  unsigned voff = lane < 60 ? (unsigned)(2 * lane) * 4u : 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 64), 4, (int)voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

int main() {
  float h[256], *d, *o, ho[256];
  for (int i = 0; i < 256; ++i) h[i] = 1000.f + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (unsigned)sizeof(h), o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  printf("lds[60..67] (expect -7 x4 then 1000 1002 1004 1006): ");
  for (int i = 60; i < 68; ++i) printf("%g ", ho[i]);
  printf("\nlds[120..131] (lanes 56..59 valid -> 1112..1118, lanes 60..63 out of range, then untouched -7): ");
  for (int i = 120; i < 132; ++i) printf("%g ", ho[i]);
  printf("\n");
  return 0;
}
