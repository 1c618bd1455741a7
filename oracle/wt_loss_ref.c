/* TEST INFRASTRUCTURE (see oracle/__init__.py) — plain-C restatement, in double precision, of the reference's
 * "WT" loss: compute_whitening_loss + compute_MMD.forward / .mmd / .gaussian_kernel / .my_cdist
 * (reference algorithms.py:1277-1309 and :59-121; duplicates at shape_networks.py:561-594, :240-309).
 * Independent of PyTorch: it checks oracle/wtpse_cpu.py and, through it, the HIP kernels.  Pinned against the
 * reference-generated fixtures in tests/test_oracle_golden.py::test_c_restatement.
 *
 *   z        [B][16][HW] fp32
 *   out[0]   ins_offdiag = mean_b clamp((sum_{i<j} |G_ij| - margin)/120, 0)      (:1289-1291)
 *   out[1]   ins_diag    = mean_b clamp((sum_i |G_ii - 1| - margin)/16, 0)      (:1297-1299)
 *   out[2]   domain      = mean over unordered domain pairs of mmd(x_a, x_b)     (:102-121)
 *   G_b = z_b z_b^T / (HW-1) + eps*I  (uncentred, :1283);  v_b = upper triangle of G_b, row-major (:1305-1306)
 */
#include <math.h>
#include <stdlib.h>

#define C 16
#define NV 120

static double kmean(const double* x, int nx, const double* y, int ny) {
  /* mean_ij exp(-max(|x_i|^2 + |y_j|^2 - 2 x_i.y_j, 1e-30))  — my_cdist + gaussian_kernel, gamma = [1] (:65-80) */
  double s = 0.0;
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ny; ++j) {
      double xn = 0.0, yn = 0.0, dot = 0.0;
      for (int k = 0; k < NV; ++k) {
        xn += x[i * NV + k] * x[i * NV + k];
        yn += y[j * NV + k] * y[j * NV + k];
        dot += x[i * NV + k] * y[j * NV + k];
      }
      double d = xn + yn - 2.0 * dot;
      if (d < 1e-30) d = 1e-30;
      s += exp(-d);
    }
  return s / ((double)nx * ny);
}

int wt_loss_ref(const float* z, int B, int HW, double eps, double margin, int domains, int per_domain, double* out,
                double* gram_out /* [B][256] or NULL */) {
  double* v = (double*)malloc(sizeof(double) * (size_t)B * NV);
  if (!v) return 1;
  double ins_off = 0.0, ins_diag = 0.0;
  for (int b = 0; b < B; ++b) {
    double G[C][C];
    for (int i = 0; i < C; ++i)
      for (int j = i; j < C; ++j) {
        const float* zi = z + ((size_t)b * C + i) * HW;
        const float* zj = z + ((size_t)b * C + j) * HW;
        double s = 0.0;
        for (int p = 0; p < HW; ++p) s += (double)zi[p] * (double)zj[p];
        G[i][j] = G[j][i] = s / (double)(HW - 1) + (i == j ? eps : 0.0);
      }
    double off = 0.0, dg = 0.0;
    int n = 0;
    for (int i = 0; i < C; ++i) {
      dg += fabs(G[i][i] - 1.0);
      for (int j = i + 1; j < C; ++j) {
        off += fabs(G[i][j]);
        v[(size_t)b * NV + n++] = G[i][j];
      }
    }
    double a = (off - margin) / 120.0, d = (dg - margin) / 16.0;
    ins_off += a > 0.0 ? a : 0.0;
    ins_diag += d > 0.0 ? d : 0.0;
    if (gram_out)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) gram_out[(size_t)b * 256 + i * C + j] = G[i][j];
  }
  out[0] = ins_off / B;
  out[1] = ins_diag / B;
  double pen = 0.0;
  for (int a = 0; a < domains; ++a)
    for (int c = a + 1; c < domains; ++c) {
      const double* x = v + (size_t)a * per_domain * NV;
      const double* y = v + (size_t)c * per_domain * NV;
      pen += kmean(x, per_domain, x, per_domain) + kmean(y, per_domain, y, per_domain) - 2.0 * kmean(x, per_domain, y, per_domain);
    }
  if (domains > 1) pen /= (double)domains * (domains - 1) / 2.0;
  out[2] = pen;
  free(v);
  return 0;
}
