// What the chip sustains on dense bf16 MFMA streams: v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16, same output tile
// per wave (64 x 64 = 4 accumulators of 32x32 or 16 of 16x16), RANDOM or all-zero operands, operands held in registers or
// re-read from LDS for every k-step (ds_read_b128, as the x3 kernels do).  Reports wall TFLOP/s (bf16) and the fp32-equivalent
// rate of the x3 arithmetic (/6), cycles per MFMA and the in-kernel clock (s_memtime / s_memrealtime).  MI355X_MICROARCH.md,
// "DVFS give-back" items 6 and 7: the clock the chip holds under an MFMA stream depends on the data and on the MFMA shape, so
// a kernel's distance from the nominal 2.4 GHz peak is partly not the kernel's.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_peak.hip -o tools/probe/_build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// SHAPE 0: 32x32x16 (2 A frags x 2 B frags per k=16 step, 4 MFMAs of 32 cycles)
// SHAPE 1: 16x16x32 (4 A frags x 4 B frags per k=32 step, 16 MFMAs of 16 cycles)   -> both: 8 fragments per 256 MFMA cycles
template <int SHAPE, bool LDS, int WAVES>
// (a 256-register budget in every variant, so that the accumulators stay in VGPRs: 256 workgroups on 256 CUs still run one per CU)
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void peak_k(const u32x4* __restrict__ src, float* out, unsigned long long* stamps, int iters) {
  __shared__ u32x4 lds[LDS ? 64 * WAVES * 8 : 1];
  const int tid = threadIdx.x;
  constexpr int NF = 8;
  bf16x8 fr[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) fr[i] = __builtin_bit_cast(bf16x8, src[(blockIdx.x * 64 * WAVES + tid) * NF + i]);
  if (LDS) {
#pragma unroll
    for (int i = 0; i < NF; ++i) lds[i * 64 * WAVES + tid] = __builtin_bit_cast(u32x4, fr[i]);
    __syncthreads();
  }
  using acc_t = typename std::conditional<SHAPE == 0, f32x16, f32x4>::type;
  constexpr int NACC = SHAPE == 0 ? 4 : 16, NR = SHAPE == 0 ? 16 : 4;
  acc_t c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < NR; ++r) c[i][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int i = 0; i < NF; ++i) fr[i] = __builtin_bit_cast(bf16x8, lds[i * 64 * WAVES + tid]);
    }
    if constexpr (SHAPE == 0) {
      // two k=16 steps (fragments 0-3 and 4-7): 8 MFMAs = 256 cycles
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            c[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s * 4 + m], fr[s * 4 + 2 + n], c[m * 2 + n], 0, 0, 0);
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          c[m * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[m], fr[4 + n], c[m * 4 + n], 0, 0, 0);
    }
    // keep the accumulators where they are across the back edge (left alone, the register allocator shuffles the sixteen
    // 16x16 tiles through AGPR copies in front of every second MFMA)
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      asm volatile("" : "+v"(c[i]));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < NR; ++r) s += c[i][r];
  out[blockIdx.x * 64 * WAVES + tid] = s;
  if (tid == 64 * WAVES - 64) {   // lane 0 of the youngest wave (the oldest one gets the pipe first and finishes early)
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

template <int SHAPE, bool LDS, int WAVES>
void run(const char* name, const u32x4* src, float* out, unsigned long long* st, bool zeros) {
  const int nblk = 256;   // one workgroup per CU; WAVES = 4: one wave per SIMD, 8: two
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // warm-up: ~1.5 s of back-to-back launches so that the clock has settled
  for (int i = 0; i < 400; ++i) hipLaunchKernelGGL((peak_k<SHAPE, LDS, WAVES>), dim3(nblk), dim3(64 * WAVES), 0, 0, src, out, st, iters);
  hipDeviceSynchronize();
  const int reps = 50;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((peak_k<SHAPE, LDS, WAVES>), dim3(nblk), dim3(64 * WAVES), 0, 0, src, out, st, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nblk * 2);
  hipMemcpy(h.data(), st, nblk * 16, hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int b = 0; b < nblk; ++b) {
    cyc.push_back((double)h[b * 2]);
    clk.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 0.1);   // GHz (s_memrealtime ticks at 100 MHz)
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  const double flop = 2.0 * 64 * 64 * 32 * (double)iters * WAVES * nblk * reps;   // per iteration and wave: 64x64 tile, k = 32
  const double tf = flop / (ms * 1e-3) * 1e-12;
  // per iteration a wave issues 256 cycles' worth of MFMAs (8 x 32 or 16 x 16); two waves per SIMD share the pipe
  printf("%-44s %-6s  %7.1f TF bf16 (x3-equivalent %6.1f TF)  %6.1f cycles per iteration and SIMD (256 = pipe always busy)  clock %.2f GHz\n", name,
         zeros ? "zeros" : "random", tf, tf / 6.0, cyc[nblk / 2] / iters * (WAVES == 8 ? 0.5 : 1.0), clk[nblk / 2]);
  fflush(stdout);
}

int main() {
  const size_t n = (size_t)256 * 512 * 8;   // u32x4 fragments
  std::vector<unsigned> h(n * 4);
  u32x4* src;
  float* out;
  unsigned long long* st;
  hipMalloc(&src, n * 16);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&st, 256 * 16);
  for (int zeros = 0; zeros < 2; ++zeros) {
    srand(1);
    for (size_t i = 0; i < n * 4; ++i) {
      if (zeros) { h[i] = 0; continue; }
      // two bf16 in [-1, 1): random sign, exponent 2^-1..2^-8, random mantissa
      unsigned v = 0;
      for (int k = 0; k < 2; ++k) {
        const unsigned sign = rand() & 1, ex = 119 + (rand() % 8), man = rand() & 0x7F;
        v |= ((sign << 15) | (ex << 7) | man) << (16 * k);
      }
      h[i] = v;
    }
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    run<0, false, 4>("32x32x16, registers, 1 wave/SIMD", src, out, st, zeros);
    run<1, false, 4>("16x16x32, registers, 1 wave/SIMD", src, out, st, zeros);
    run<0, true, 4>("32x32x16, LDS re-read, 1 wave/SIMD", src, out, st, zeros);
    run<1, true, 4>("16x16x32, LDS re-read, 1 wave/SIMD", src, out, st, zeros);
    run<0, true, 8>("32x32x16, LDS re-read, 2 waves/SIMD", src, out, st, zeros);
    run<1, true, 8>("16x16x32, LDS re-read, 2 waves/SIMD", src, out, st, zeros);
  }
  return 0;
}
