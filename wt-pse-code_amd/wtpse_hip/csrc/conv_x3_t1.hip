// conv_x3_k / conv_x3r_k instantiated for TERMS = 1 (conv_x3_kernels.h): one translation unit per arithmetic, compiled side by side.
#include "conv_x3_kernels.h"
int x3_dispatch_t1(const ConvX3Args& a, const X3Launch& L, hipStream_t st) { return x3_dispatch<1>(a, L, st); }
