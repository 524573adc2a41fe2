/* ORACLE — test infrastructure only; never linked into or called by the product library.
 *
 * Plain-C restatement of the selective scan that MedMamba reaches through
 * mamba_ssm.ops.selective_scan_interface.selective_scan_fn (reference call site
 * MedMamba.py:273-279).  mamba_ssm==1.0.1 (README.md:19) is a CUDA-only third-party package that
 * is not in /root/reference, so the arithmetic follows the reference's only in-tree statement
 * of it, the selective_scan_ref body quoted in temp.py:57-139:
 *   temp.py:61-64   delta = softplus(delta + delta_bias)        (F.softplus: threshold 20)
 *   temp.py:88      deltaA   = exp(delta * A)                    per (b,d,l,n)
 *   temp.py:95-96   deltaB_u = delta * B[b, g(d), n, l] * u      g(d) = d / (dim/G)
 *   temp.py:111-125 x = deltaA*x + deltaB_u ; y = sum_n x*C[b,g(d),n,l]      x(-1) = 0
 *   temp.py:135     out = y + u * D
 * Only the variant MedMamba uses is restated: real A, B/C of shape (batch,G,N,L), z = None,
 * no last-state output.  Parity: "unpinned" by the reference's own tests (it has none); this file
 * is cross-checked against oracle/scan_ref.py (the PyTorch loop) and torch autograd of it through
 * tests/golden/scan_*.npz (tools/gen_golden.py).
 *
 * The backward is the analytic adjoint of that loop (what autograd of temp.py:57-139 computes),
 * carried in double.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

static int g_threads = 0;
void oracle_set_threads(int n) { g_threads = n; }
static int nthreads(void) { return g_threads > 0 ? g_threads : omp_get_max_threads(); }

static inline float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
static inline double softplus_d(double x) { return x > 20.0 ? x : log1p(exp(x)); }

#define IDX4(b, g, n, l, s0, s1, s2) ((long)(b) * (s0) + (long)(g) * (s1) + (long)(n) * (s2) + (l))

/* ---- forward, fp32 arithmetic in the quoted operation order --------------------------- */
static void fwd_f32(const float* u, const float* delta, const float* A, const float* B, const float* C,
                    const float* D, const float* bias, float* out, float* xs, int batch, int dim, int L,
                    int N, int G, long b0, long b1, long b2, long c0, long c1, long c2, int sp, int T) {
  const int H = dim / G, nch = T ? (L + T - 1) / T : 0;
#pragma omp parallel for collapse(2) num_threads(nthreads()) schedule(static)
  for (int b = 0; b < batch; ++b)
    for (int d = 0; d < dim; ++d) {
      float x[256];
      for (int n = 0; n < N; ++n) x[n] = 0.0f;
      const int g = d / H;
      const long row = ((long)b * dim + d) * L;
      for (int l = 0; l < L; ++l) {
        float dl = delta[row + l];
        if (bias) dl = dl + bias[d];
        if (sp) dl = softplus_f(dl);
        const float uu = u[row + l];
        float y = 0.0f;
        for (int n = 0; n < N; ++n) {
          const float dA = expf(dl * A[d * N + n]);
          const float dBu = dl * B[IDX4(b, g, n, l, b0, b1, b2)] * uu;
          x[n] = dA * x[n] + dBu;
          y += x[n] * C[IDX4(b, g, n, l, c0, c1, c2)];
        }
        out[row + l] = D ? y + uu * D[d] : y;
        if (T && ((l + 1) % T == 0 || l == L - 1))
          for (int n = 0; n < N; ++n) xs[(((long)b * dim + d) * nch + l / T) * N + n] = x[n];
      }
    }
}

/* ---- forward, same algorithm in double (the arbiter) ---------------------------------- */
static void fwd_f64(const float* u, const float* delta, const float* A, const float* B, const float* C,
                    const float* D, const float* bias, double* out, double* xs, int batch, int dim, int L,
                    int N, int G, long b0, long b1, long b2, long c0, long c1, long c2, int sp, int T) {
  const int H = dim / G, nch = T ? (L + T - 1) / T : 0;
#pragma omp parallel for collapse(2) num_threads(nthreads()) schedule(static)
  for (int b = 0; b < batch; ++b)
    for (int d = 0; d < dim; ++d) {
      double x[256];
      for (int n = 0; n < N; ++n) x[n] = 0.0;
      const int g = d / H;
      const long row = ((long)b * dim + d) * L;
      for (int l = 0; l < L; ++l) {
        double dl = delta[row + l];
        if (bias) dl = dl + (double)bias[d];
        if (sp) dl = softplus_d(dl);
        const double uu = u[row + l];
        double y = 0.0;
        for (int n = 0; n < N; ++n) {
          const double dA = exp(dl * (double)A[d * N + n]);
          const double dBu = dl * (double)B[IDX4(b, g, n, l, b0, b1, b2)] * uu;
          x[n] = dA * x[n] + dBu;
          y += x[n] * (double)C[IDX4(b, g, n, l, c0, c1, c2)];
        }
        out[row + l] = D ? y + uu * (double)D[d] : y;
        if (T && ((l + 1) % T == 0 || l == L - 1))
          for (int n = 0; n < N; ++n) xs[(((long)b * dim + d) * nch + l / T) * N + n] = x[n];
      }
    }
}

int oracle_scan_fwd(const float* u, const float* delta, const float* A, const float* B, const float* C,
                    const float* D, const float* bias, void* out, void* xs, int batch, int dim, int L, int N,
                    int G, long b0, long b1, long b2, long c0, long c1, long c2, int sp, int f64, int T) {
  if (N > 256 || G <= 0 || dim % G) return -1;
  if (f64)
    fwd_f64(u, delta, A, B, C, D, bias, (double*)out, (double*)xs, batch, dim, L, N, G, b0, b1, b2, c0, c1, c2, sp, T);
  else
    fwd_f32(u, delta, A, B, C, D, bias, (float*)out, (float*)xs, batch, dim, L, N, G, b0, b1, b2, c0, c1, c2, sp, T);
  return 0;
}

/* ---- backward: adjoint of the loop, double accumulation ------------------------------- *
 * gx_t[n]  = C_t[n] g_t + a_{t+1}[n] gx_{t+1}[n]            (adjoint of x_t)
 * dC_t[n] += g_t x_t[n]          dB_t[n] += gx_t[n] dl_t u_t          (summed over the H channels of g)
 * ddl_t    = sum_n gx_t[n] ( x_{t-1}[n] a_t[n] A[n] + B_t[n] u_t )
 * du_t     = sum_n gx_t[n] dl_t B_t[n] + D g_t     dD += g_t u_t
 * dA[n]   += gx_t[n] x_{t-1}[n] a_t[n] dl_t
 * ddelta_t = ddl_t * softplus'(delta_t + bias)   ;  dbias += ddelta_t
 * dB/dC outputs are contiguous (batch,G,N,L).                                                  */
int oracle_scan_bwd(const float* u, const float* delta, const float* A, const float* B, const float* C,
                    const float* D, const float* bias, const float* dout, double* du, double* ddelta,
                    double* dA, double* dB, double* dC, double* dD, double* dbias, int batch, int dim, int L,
                    int N, int G, long b0, long b1, long b2, long c0, long c1, long c2, int sp, int unused) {
  (void)unused;
  if (N > 256 || G <= 0 || dim % G) return -1;
  const int H = dim / G;
  double* dA_part = (double*)calloc((size_t)batch * dim * N, sizeof(double));
  double* dD_part = (double*)calloc((size_t)batch * dim, sizeof(double));
  double* db_part = (double*)calloc((size_t)batch * dim, sizeof(double));
  if (!dA_part || !dD_part || !db_part) return -2;
  int fail = 0;
#pragma omp parallel for collapse(2) num_threads(nthreads()) schedule(dynamic)
  for (int b = 0; b < batch; ++b)
    for (int g = 0; g < G; ++g) {
      double* xs = (double*)malloc(sizeof(double) * (size_t)(L + 1) * N); /* x_{-1} .. x_{L-1} */
      double* dls = (double*)malloc(sizeof(double) * (size_t)L);
      double gx[256];
      if (!xs || !dls) { fail = 1; free(xs); free(dls); continue; }
      for (int h = 0; h < H; ++h) {
        const int d = g * H + h;
        const long row = ((long)b * dim + d) * L;
        for (int n = 0; n < N; ++n) xs[n] = 0.0;
        for (int l = 0; l < L; ++l) {
          double dl = delta[row + l];
          if (bias) dl += (double)bias[d];
          if (sp) dl = softplus_d(dl);
          dls[l] = dl;
          const double uu = u[row + l];
          for (int n = 0; n < N; ++n) {
            const double a = exp(dl * (double)A[d * N + n]);
            xs[(long)(l + 1) * N + n] = a * xs[(long)l * N + n] + dl * (double)B[IDX4(b, g, n, l, b0, b1, b2)] * uu;
          }
        }
        for (int n = 0; n < N; ++n) gx[n] = 0.0;
        double accD = 0.0, accb = 0.0;
        for (int l = L - 1; l >= 0; --l) {
          const double gt = dout[row + l], uu = u[row + l], dl = dls[l];
          double ddl = 0.0, duu = 0.0;
          for (int n = 0; n < N; ++n) {
            const double An = A[d * N + n];
            const double Bn = B[IDX4(b, g, n, l, b0, b1, b2)], Cn = C[IDX4(b, g, n, l, c0, c1, c2)];
            /* gx holds a_{l+1} * gx_{l+1} on entry */
            const double gxt = Cn * gt + gx[n];
            const double a = exp(dl * An);
            const double xprev = xs[(long)l * N + n], xcur = xs[(long)(l + 1) * N + n];
            const long o = (((long)b * G + g) * N + n) * L + l;
            dC[o] += gt * xcur;
            dB[o] += gxt * dl * uu;
            ddl += gxt * (xprev * a * An + Bn * uu);
            duu += gxt * dl * Bn;
            dA_part[((long)b * dim + d) * N + n] += gxt * xprev * a * dl;
            gx[n] = a * gxt;
          }
          if (D) { duu += (double)D[d] * gt; accD += gt * uu; }
          du[row + l] = duu;
          double raw = delta[row + l];
          if (bias) raw += (double)bias[d];
          const double dd = sp ? (raw > 20.0 ? ddl : ddl / (1.0 + exp(-raw))) : ddl;
          ddelta[row + l] = dd;
          accb += dd;
        }
        dD_part[(long)b * dim + d] = accD;
        db_part[(long)b * dim + d] = accb;
      }
      free(xs); free(dls);
    }
  for (int d = 0; d < dim; ++d) {
    double sD = 0.0, sb = 0.0;
    for (int b = 0; b < batch; ++b) { sD += dD_part[(long)b * dim + d]; sb += db_part[(long)b * dim + d]; }
    if (dD) dD[d] = sD;
    if (dbias) dbias[d] = sb;
    for (int n = 0; n < N; ++n) {
      double s = 0.0;
      for (int b = 0; b < batch; ++b) s += dA_part[((long)b * dim + d) * N + n];
      dA[d * N + n] = s;
    }
  }
  free(dA_part); free(dD_part); free(db_part);
  return fail ? -2 : 0;
}
