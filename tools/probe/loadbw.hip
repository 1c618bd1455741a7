// Micro-benchmark: how fast can a CU pull NCHW fp32 tiles into registers with the conv loaders' access pattern?
//   lane <-> consecutive pixels of one channel, a batch of NB loads per lane (channels c..c+NB-1, stride H*W*4 bytes),
//   VW = dwords per load (1, 2, 4: the lane then owns VW consecutive pixels), 256-thread workgroups, 2 per CU.
// Prints GB/s for each (VW, NB).  hipcc --offload-arch=gfx950 -O3 tools/probe/loadbw.hip -o tools/probe/_build/loadbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int VW, int NB>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ x, float* __restrict__ out, int C, int HW, int ntiles, int tiles_per_img) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  // a tile = 64*VW pixels x 64 channels; wave w takes channels [16w, 16w+16) in batches of NB
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / tiles_per_img, t = tile - b * tiles_per_img;
    const float* base = x + (size_t)b * C * HW + (size_t)t * 64 * VW + lane * VW;
#pragma unroll
    for (int c0 = 0; c0 < 16; c0 += NB) {
      float v[NB][VW];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const float* p = base + (size_t)(wave * 16 + c0 + j) * HW;
        if constexpr (VW == 1) v[j][0] = __builtin_nontemporal_load(p);
        else if constexpr (VW == 2) { float2 q = *reinterpret_cast<const float2*>(p); v[j][0] = q.x; v[j][1] = q.y; }
        else { float4 q = *reinterpret_cast<const float4*>(p); v[j][0] = q.x; v[j][1] = q.y; v[j][2] = q.z; v[j][3] = q.w; }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int e = 0; e < VW; ++e) acc += v[j][e];
    }
  }
  if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <int VW, int NB>
void run(const float* x, float* out, int B, int C, int HW) {
  const int tiles_per_img = HW / (64 * VW), ntiles = B * tiles_per_img;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<VW, NB>), dim3(512), dim3(256), 0, 0, x, out, C, HW, ntiles, tiles_per_img);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 10.0 * B * 64.0 * HW * 4;
  printf("VW %d dwords/lane, %2d loads in flight per lane: %7.1f us/launch  %7.1f GB/s  (%.2f B/clk/CU at 2.1 GHz)\n", VW, NB, ms * 100, bytes / ms * 1e-6,
         bytes / ms * 1e-6 / 256 / 2.1);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32;     // 32 images = 134 MB (Infinity-Cache resident when re-read); 512 = 2.1 GB (HBM)
  const int C = 64, H = 128, W = 128, HW = H * W;
  printf("tensor [%d,64,128,128] fp32 = %.0f MB, read once per launch\n", B, B * 64.0 * HW * 4e-6);
  float *x, *out;
  hipMalloc(&x, (size_t)B * C * HW * 4);
  hipMalloc(&out, 4096);
  hipMemset(x, 0, (size_t)B * C * HW * 4);
  run<1, 4>(x, out, B, C, HW);  run<1, 8>(x, out, B, C, HW);  run<1, 16>(x, out, B, C, HW);
  run<2, 4>(x, out, B, C, HW);  run<2, 8>(x, out, B, C, HW);  run<2, 16>(x, out, B, C, HW);
  run<4, 2>(x, out, B, C, HW);  run<4, 4>(x, out, B, C, HW);  run<4, 8>(x, out, B, C, HW);
  return 0;
}
