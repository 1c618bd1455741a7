#!/usr/bin/env python3
"""fp32-MFMA conv kernels vs the split-bf16 ("x3") kernels at the benchmark's layer shapes (HIP events)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402
from wtpse_hip import ops  # noqa: E402
from microbench import SHAPES, timeit, pack, DEV  # noqa: E402


def pack_x3(w):
    co, ci, k, _ = w.shape
    t = k * k
    xf, xd = ops.x3_packed_size(co, ci, t), ops.x3_packed_size(ci, co, t)
    packed = torch.zeros(xf + xd, dtype=torch.int16, device=DEV)
    desc = torch.tensor([0, co, ci, t, 0, xf, 0, 0], dtype=torch.int32, device=DEV)
    ops.lib().call("wtpse_pack_conv_weights_x3", w.reshape(-1).contiguous().data_ptr(), desc.data_ptr(), 1, packed.data_ptr(), ops.stream_ptr())
    return packed, xf


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
  tot = [0.0] * 8
  for c0, c1, co, H, k, name in SHAPES:
      if c0 + c1 < 16:
          continue
      x0 = torch.randn(B, c0, H, H, device=DEV)
      x1 = torch.randn(B, c1, H, H, device=DEV) if c1 else None
      w = torch.randn(co, c0 + c1, k, k, device=DEV) * 0.05
      bias = torch.zeros(co, device=DEV)
      packed, wd_off = pack(w)
      px, xd_off = pack_x3(w)
      dy = torch.randn(B, co, H, H, device=DEV)
      pro0 = torch.rand(c0, 2, device=DEV)
      pro1 = torch.rand(c1, 2, device=DEV) if c1 else None
      flops = 2.0 * (c0 + c1) * co * k * k * H * H * B
      f32, _ = timeit(lambda: ops.conv_fwd(x0, x1, packed.data_ptr(), bias, co, k, pro0, 3, want_stats=True, pro1=pro1), 10)
      f3, _ = timeit(lambda: ops.conv_fwd_x3(x0, x1, px.data_ptr(), bias, co, k, pro0, 3, want_stats=True, pro1=pro1), 10)
      line = "%-12s %3d+%-3d->%-3d k%d @%3d  fwd fp32 %7.1f us %5.1f TF | x3 %7.1f us %5.1f TF (%.2fx)" % (
          name, c0, c1, co, k, H, f32, flops / f32 / 1e6, f3, flops / f3 / 1e6, f32 / f3)
      tot[0] += f32; tot[1] += f3
      if k == 3:        # the LDS-fed-weights kernel of rounds 2-3 (conv_x3_k) beside the default (conv_x3r_k)
          ops.lib().query("wtpse_x3r_enable", 0)
          fo, _ = timeit(lambda: ops.conv_fwd_x3(x0, x1, px.data_ptr(), bias, co, k, pro0, 3, want_stats=True, pro1=pro1), 10)
          ops.lib().query("wtpse_x3r_enable", 2)
          fr, _ = timeit(lambda: ops.conv_fwd_x3(x0, x1, px.data_ptr(), bias, co, k, pro0, 3, want_stats=True, pro1=pro1), 10)
          ops.lib().query("wtpse_x3r_enable", 1)
          line += " [conv_x3_k %7.1f us %5.1f TF, conv_x3r_k %7.1f us %5.1f TF]" % (fo, flops / fo / 1e6, fr, flops / fr / 1e6)
          tot[6] += fo
      if c0 + c1 > 16:
          d32, _ = timeit(lambda: ops.conv_fwd(dy, None, packed.data_ptr() + 4 * wd_off, None, c0 + c1, k, split=(c0 if c1 else None)), 10)
          d3, _ = timeit(lambda: ops.conv_fwd_x3(dy, None, px.data_ptr() + 2 * xd_off, None, c0 + c1, k, split=(c0 if c1 else None)), 10)
          line += " || dgrad fp32 %7.1f us %5.1f TF | x3 %7.1f us %5.1f TF (%.2fx)" % (d32, flops / d32 / 1e6, d3, flops / d3 / 1e6, d32 / d3)
          tot[2] += d32; tot[3] += d3
          if k == 3:
              ops.lib().query("wtpse_x3r_enable", 0)
              do, _ = timeit(lambda: ops.conv_fwd_x3(dy, None, px.data_ptr() + 2 * xd_off, None, c0 + c1, k, split=(c0 if c1 else None)), 10)
              ops.lib().query("wtpse_x3r_enable", 2)
              dr, _ = timeit(lambda: ops.conv_fwd_x3(dy, None, px.data_ptr() + 2 * xd_off, None, c0 + c1, k, split=(c0 if c1 else None)), 10)
              ops.lib().query("wtpse_x3r_enable", 1)
              line += " [conv_x3_k %7.1f us %5.1f TF, conv_x3r_k %7.1f us %5.1f TF]" % (do, flops / do / 1e6, dr, flops / dr / 1e6)
              tot[7] += do
      if ops.wgrad_x3_supported(c0 + c1, co, k, c0 if c1 else 8):
          dw = torch.empty_like(w)
          w32, _ = timeit(lambda: ops.conv_wgrad(dy, x0, x1, k, dw, None, pro0, 3, False, pro1), 10)
          w3, _ = timeit(lambda: ops.conv_wgrad_x3(dy, x0, x1, k, dw, pro0, 3, False, pro1), 10)
          line += " || wgrad fp32 %7.1f us %5.1f TF | x3 %7.1f us %5.1f TF (%.2fx)" % (w32, flops / w32 / 1e6, w3, flops / w3 / 1e6, w32 / w3)
          tot[4] += w32; tot[5] += w3
      print(line, flush=True)
  print("wgrad (supported layers): fp32 %.0f us, x3 %.0f us" % (tot[4], tot[5]))
  print("sum: fwd fp32 %.0f us, x3 %.0f us; dgrad fp32 %.0f us, x3 %.0f us" % tuple(tot[:4]))
  print("3x3 layers on conv_x3_k only (the kernel of rounds 2-3): fwd %.0f us, dgrad %.0f us" % (tot[6], tot[7]))


if __name__ == "__main__":
    main()
