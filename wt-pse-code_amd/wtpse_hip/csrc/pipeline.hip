// Device-side training input pipeline: the reference prepares every sample on one host thread with
// Pillow — Compose([Resize(256), RandomScaleCrop(256), Normalize_tf(), ToTensor()]), train.py:58-62 with
// custom_transforms.py:375-391,330-354,139-176,455-499,581-599 — and stacks the batch (Trainer.py:45-55).
// Here the decoded uint8 samples go to the GPU and three kernels do the rest, bit for bit:
//   resample_u8_k : one pass of Pillow's 8-bit separable resampling (src/libImaging/Resample.c
//                   ImagingResampleHorizontal/Vertical_8bpc): 22-bit fixed-point coefficients, +half, >> 22, clip to
//                   8 bits; the coefficient tables (precompute_coeffs + normalize_coeffs_8bpc) are built on the host
//                   (wtpse_hip/input_pipeline.py).  Resize = bicubic pass H then V; the random up-scale = bilinear pass
//                   H then V with per-sample tables that cover only the 256 columns / rows the random crop keeps.
//   input_finish_k: the NEAREST resize + crop of the disc mask as a per-sample index gather, img/127.5 - 1, the mask
//                   thresholds (> 200 background; <= 50 cup — cut from the disc image, custom_transforms.py:488-489),
//                   HWC -> CHW, fp32.
#include "common.h"

#define PRECISION_BITS 22

// in [N][Hin][Win][C], out [N][Hin][Wout][C] (vertical = 0) or in [N][Hin][W][C] -> out [N][Hout][W][C] (vertical = 1).
// bounds [T][L][2] = (first source index, tap count), kk [T][L][ksize]; sample n uses table tab[n] (tab null: table 0).
__global__ __launch_bounds__(256) void resample_u8_k(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                     const int* __restrict__ bounds, const int* __restrict__ kk,
                                                     const int* __restrict__ tab, int ksize, int Hin, int Win, int C, int L,
                                                     int vertical, long long per_sample_out) {
  const int n = blockIdx.y;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= per_sample_out) return;
  const int t = tab ? tab[n] : 0;
  const int Wo = vertical ? Win : L;
  const int c = (int)(e % C);
  const long long r = e / C;
  const int xo = (int)(r % Wo), yo = (int)(r / Wo);
  const int pos = vertical ? yo : xo;                       // position along the resampled axis
  const int* b = bounds + ((size_t)t * L + pos) * 2;
  const int* k = kk + ((size_t)t * L + pos) * ksize;
  const int first = b[0], cnt = b[1];
  const unsigned char* src = in + (size_t)n * Hin * Win * C;
  int ss = 1 << (PRECISION_BITS - 1);
  if (vertical) {
    for (int i = 0; i < cnt; ++i) ss += (int)src[((size_t)(first + i) * Win + xo) * C + c] * k[i];
  } else {
    for (int i = 0; i < cnt; ++i) ss += (int)src[((size_t)yo * Win + first + i) * C + c] * k[i];
  }
  ss >>= PRECISION_BITS;                                    // arithmetic shift, as Pillow's clip8(in >> PRECISION_BITS)
  out[(size_t)n * per_sample_out + e] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}

// img [N][S][S][3] u8 (already scaled + cropped), od [N][S][S] u8 (after Resize): od_out(y, x) looks at
// od[yidx[n][y]][xidx[n][x]] (nearest resize + crop folded into the index tables).
__global__ __launch_bounds__(256) void input_finish_k(const unsigned char* __restrict__ img, const unsigned char* __restrict__ od,
                                                      const int* __restrict__ xidx, const int* __restrict__ yidx,
                                                      float* __restrict__ image, float* __restrict__ od_out,
                                                      float* __restrict__ oc_out, int S) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= S * S) return;
  const int y = p / S, x = p - y * S;
  const unsigned char* px = img + ((size_t)n * S * S + p) * 3;
  float* dst = image + (size_t)n * 3 * S * S + p;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = (float)px[c];
    v = __fdiv_rn(v, 127.5f);        // img /= 127.5 ; img -= 1.0 in float32, two roundings as numpy does
    dst[(size_t)c * S * S] = v - 1.0f;
  }
  const unsigned char m = od[(size_t)n * S * S + (size_t)yidx[n * S + y] * S + xidx[n * S + x]];
  od_out[(size_t)n * S * S + p] = m > 200 ? 0.f : 1.f;
  oc_out[(size_t)n * S * S + p] = m > 50 ? 0.f : 1.f;
}

// See include/wtpse_hip.h for the contract.
extern "C" int wtpse_resample_u8(const unsigned char* in, unsigned char* out, const int* bounds, const int* kk, const int* tab,
                                 int ksize, int N, int Hin, int Win, int C, int L, int vertical, void* stream) {
  WTPSE_REQUIRE(in && out && bounds && kk && ksize > 0 && N > 0 && N < 65536 && Hin > 0 && Win > 0 && C > 0 && L > 0);
  const long long per = vertical ? (long long)L * Win * C : (long long)Hin * L * C;
  hipLaunchKernelGGL(resample_u8_k, dim3((unsigned)((per + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, in, out,
                     bounds, kk, tab, ksize, Hin, Win, C, L, vertical, per);
  return wtpse_status();
}

extern "C" int wtpse_input_finish(const unsigned char* img, const unsigned char* od, const int* xidx, const int* yidx,
                                  float* image, float* od_out, float* oc_out, int N, int S, void* stream) {
  WTPSE_REQUIRE(img && od && xidx && yidx && image && od_out && oc_out && N > 0 && N < 65536 && S > 0);
  hipLaunchKernelGGL(input_finish_k, dim3((unsigned)((S * S + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, img,
                     od, xidx, yidx, image, od_out, oc_out, S);
  return wtpse_status();
}
