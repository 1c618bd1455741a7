// conv_x3_k / conv_x3r_k instantiated for TERMS = 3 (conv_x3_kernels.h): one translation unit per arithmetic, compiled side by side.
#include "conv_x3_kernels.h"
int x3_dispatch_t3(const ConvX3Args& a, const X3Launch& L, hipStream_t st) { return x3_dispatch<3>(a, L, st); }
