/*
 * wm_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * Plain-C, float64-accumulating restatement of the two operators on the hot path of
 * akollgaa/weather-model.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (weather_model_amd/) never does.
 *
 * Pinning status
 *   conv2d_*  : PINNED. The reference computes its convolutions by calling
 *               torch.nn.Conv2d (newConvolution.py:129-133, forward calls :149,152,155,158,
 *               backward through loss.backward() :184).  tests/golden/make_golden.py runs
 *               that reference (CustomModel imported from /root/reference) and
 *               torch.nn.functional.conv2d on CPU and stores inputs/outputs/gradients in
 *               tests/golden/; tests/test_oracle.py checks every function below against
 *               those vectors.
 *   gram / solve : PARITY UNPINNED.  The regression path of the reference lives in
 *               tensorflow+keras (multipleRegression.py:273-298, singleRegression.py:74-96),
 *               neither of which exists in this image, and the reference holds no fixture
 *               for it.  The oracle below restates the closed-form minimiser of the exact
 *               loss those scripts minimise (MSE of an affine map = ordinary least squares
 *               with intercept) and is cross-checked against numpy.linalg.lstsq (float64).
 *
 * Conventions (the operator class of every conv call in the reference):
 *   cross-correlation, valid padding, stride 1, dilation 1, groups 1, NCHW contiguous,
 *   weight (Cout,Cin,kH,kW), optional bias (Cout).  Inputs are float32 (what the product
 *   sees), every sum is carried in float64, outputs are float64.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int wm_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* y[n,o,i,j] = b[o] + sum_{c,kh,kw} w[o,c,kh,kw] * x[n,c,i+kh,j+kw]
 * (torch.nn.Conv2d defaults as used at newConvolution.py:129-133). */
void wm_oracle_conv2d_fwd(const float *x, const float *w, const float *b, double *y,
                          int N, int Cin, int H, int W, int Cout, int kH, int kW)
{
    const int OH = H - kH + 1, OW = W - kW + 1;
    if (OH <= 0 || OW <= 0) return;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < Cout; ++o) {
            for (int i = 0; i < OH; ++i) {
                double *yr = y + (((size_t)n * Cout + o) * OH + i) * OW;
                const double bias = b ? (double)b[o] : 0.0;
                for (int j = 0; j < OW; ++j) yr[j] = bias;
                for (int c = 0; c < Cin; ++c)
                    for (int kh = 0; kh < kH; ++kh) {
                        const float *xr = x + (((size_t)n * Cin + c) * H + i + kh) * W;
                        for (int kw = 0; kw < kW; ++kw) {
                            const double wv = w[(((size_t)o * Cin + c) * kH + kh) * kW + kw];
                            for (int j = 0; j < OW; ++j) yr[j] += wv * (double)xr[j + kw];
                        }
                    }
            }
        }
}

/* gx[n,c,p,q] = sum_{o,kh,kw} w[o,c,kh,kw] * gy[n,o,p-kh,q-kw]   (autograd of the above) */
void wm_oracle_conv2d_bwd_data(const float *gy, const float *w, double *gx,
                               int N, int Cin, int H, int W, int Cout, int kH, int kW)
{
    const int OH = H - kH + 1, OW = W - kW + 1;
    memset(gx, 0, sizeof(double) * (size_t)N * Cin * H * W);
    if (OH <= 0 || OW <= 0) return;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < Cin; ++c)
            for (int o = 0; o < Cout; ++o)
                for (int kh = 0; kh < kH; ++kh)
                    for (int kw = 0; kw < kW; ++kw) {
                        const double wv = w[(((size_t)o * Cin + c) * kH + kh) * kW + kw];
                        for (int i = 0; i < OH; ++i) {
                            const float *gr = gy + (((size_t)n * Cout + o) * OH + i) * OW;
                            double *xr = gx + (((size_t)n * Cin + c) * H + i + kh) * W + kw;
                            for (int j = 0; j < OW; ++j) xr[j] += wv * (double)gr[j];
                        }
                    }
}

/* gw[o,c,kh,kw] = sum_{n,i,j} gy[n,o,i,j] * x[n,c,i+kh,j+kw];  gb[o] = sum_{n,i,j} gy[n,o,i,j] */
void wm_oracle_conv2d_bwd_wb(const float *x, const float *gy, double *gw, double *gb,
                             int N, int Cin, int H, int W, int Cout, int kH, int kW)
{
    const int OH = H - kH + 1, OW = W - kW + 1;
    const int nw = Cout * Cin * kH * kW;
    for (int t = 0; t < nw; ++t) gw[t] = 0.0;
    if (gb) for (int o = 0; o < Cout; ++o) gb[o] = 0.0;
    if (OH <= 0 || OW <= 0) return;
    /* one (image, output-row) pair per work item; per-thread float64 partials, summed in
     * thread order afterwards (deterministic for a fixed thread count). */
    const int nt = wm_oracle_threads();
    double *part = (double *)calloc((size_t)nt * (nw + Cout), sizeof(double));
#pragma omp parallel
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double *pw = part + (size_t)tid * (nw + Cout);
        double *pb = pw + nw;
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < OH; ++i)
                for (int o = 0; o < Cout; ++o) {
                    const float *gr = gy + (((size_t)n * Cout + o) * OH + i) * OW;
                    double rb = 0.0;
                    for (int j = 0; j < OW; ++j) rb += (double)gr[j];
                    pb[o] += rb;
                    for (int c = 0; c < Cin; ++c)
                        for (int kh = 0; kh < kH; ++kh) {
                            const float *xr = x + (((size_t)n * Cin + c) * H + i + kh) * W;
                            for (int kw = 0; kw < kW; ++kw) {
                                double r = 0.0;
                                for (int j = 0; j < OW; ++j) r += (double)gr[j] * (double)xr[j + kw];
                                pw[(((size_t)o * Cin + c) * kH + kh) * kW + kw] += r;
                            }
                        }
                }
    }
    for (int t = 0; t < nt; ++t) {
        const double *pw = part + (size_t)t * (nw + Cout);
        for (int q = 0; q < nw; ++q) gw[q] += pw[q];
        if (gb) for (int o = 0; o < Cout; ++o) gb[o] += pw[nw + o];
    }
    free(part);
}

/*
 * float32 "what the reference's library does on a CPU" leg used ONLY by bench.py's cpu_baseline:
 * forward + backward(weight,bias) of the single-channel stencil, all host threads, float32
 * accumulation inside a row, float64 across rows.  Returns nothing; outputs are written so the
 * compiler cannot drop the work.
 */
void wm_oracle_conv2d_fwd_bwd_wb_f32(const float *x, const float *w, const float *b,
                                     const float *gy, float *y, double *gw, double *gb,
                                     int N, int Cin, int H, int W, int Cout, int kH, int kW)
{
    const int OH = H - kH + 1, OW = W - kW + 1;
    if (OH <= 0 || OW <= 0) return;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < Cout; ++o)
            for (int i = 0; i < OH; ++i) {
                float *yr = y + (((size_t)n * Cout + o) * OH + i) * OW;
                const float bias = b ? b[o] : 0.0f;
                for (int j = 0; j < OW; ++j) yr[j] = bias;
                for (int c = 0; c < Cin; ++c)
                    for (int kh = 0; kh < kH; ++kh) {
                        const float *xr = x + (((size_t)n * Cin + c) * H + i + kh) * W;
                        for (int kw = 0; kw < kW; ++kw) {
                            const float wv = w[(((size_t)o * Cin + c) * kH + kh) * kW + kw];
                            for (int j = 0; j < OW; ++j) yr[j] += wv * xr[j + kw];
                        }
                    }
            }
    wm_oracle_conv2d_bwd_wb(x, gy, gw, gb, N, Cin, H, W, Cout, kH, kW);
}

/*
 * Gram of Z = [X | 1 | Y]  (D = F + 1 + P columns), float64, row-major D x D:
 *   acc[a][b] += sum_rows z_a z_b.
 * This is the sufficient statistic of the least-squares problem the reference's linear Dense
 * stacks minimise (multipleRegression.py:273-298 with X/Y rows built at :217-234;
 * singleRegression.py:74-96).  acc is a running sum (caller zeroes it).
 */
void wm_oracle_gram(const float *X, const float *Y, int64_t n, int F, int P, double *acc)
{
    const int D = F + 1 + P;
    const int nt = wm_oracle_threads();
    double *part = (double *)calloc((size_t)nt * D * D, sizeof(double));
#pragma omp parallel
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double *a = part + (size_t)tid * D * D;
        double *z = (double *)malloc(sizeof(double) * D);
#pragma omp for schedule(static)
        for (int64_t r = 0; r < n; ++r) {
            for (int f = 0; f < F; ++f) z[f] = (double)X[r * F + f];
            z[F] = 1.0;
            for (int p = 0; p < P; ++p) z[F + 1 + p] = (double)Y[r * P + p];
            for (int i = 0; i < D; ++i) {
                const double zi = z[i];
                double *ai = a + (size_t)i * D;
                for (int j = i; j < D; ++j) ai[j] += zi * z[j];
            }
        }
        free(z);
    }
    for (int t = 0; t < nt; ++t)
        for (int i = 0; i < D; ++i)
            for (int j = i; j < D; ++j) {
                const double v = part[(size_t)t * D * D + (size_t)i * D + j];
                acc[(size_t)i * D + j] += v;
                if (j != i) acc[(size_t)j * D + i] += v;
            }
    free(part);
}

/* float32 Gram (what a numpy fp32 A.T@A would carry) -- bench.py cpu_baseline only. */
void wm_oracle_gram_f32(const float *X, const float *Y, int64_t n, int F, int P, double *acc)
{
    const int D = F + 1 + P;
    const int nt = wm_oracle_threads();
    double *part = (double *)calloc((size_t)nt * D * D, sizeof(double));
#pragma omp parallel
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double *a = part + (size_t)tid * D * D;
        float *z = (float *)malloc(sizeof(float) * D);
        float *s = (float *)calloc((size_t)D * D, sizeof(float));
        int64_t cnt = 0;
#pragma omp for schedule(static)
        for (int64_t r = 0; r < n; ++r) {
            for (int f = 0; f < F; ++f) z[f] = X[r * F + f];
            z[F] = 1.0f;
            for (int p = 0; p < P; ++p) z[F + 1 + p] = Y[r * P + p];
            for (int i = 0; i < D; ++i) {
                const float zi = z[i];
                float *si = s + (size_t)i * D;
                for (int j = i; j < D; ++j) si[j] += zi * z[j];
            }
            if (++cnt == 1024) {
                for (int t = 0; t < D * D; ++t) { a[t] += (double)s[t]; s[t] = 0.0f; }
                cnt = 0;
            }
        }
        for (int t = 0; t < D * D; ++t) a[t] += (double)s[t];
        free(z);
        free(s);
    }
    for (int t = 0; t < nt; ++t)
        for (int i = 0; i < D; ++i)
            for (int j = i; j < D; ++j) {
                const double v = part[(size_t)t * D * D + (size_t)i * D + j];
                acc[(size_t)i * D + j] += v;
                if (j != i) acc[(size_t)j * D + i] += v;
            }
    free(part);
}

/*
 * Normal-equation solve from the Gram above (float64 Cholesky on the centred system):
 *   beta[(F+1) x P] row-major, intercept in the LAST row (0 when fit_intercept == 0).
 * Returns 0, or -1 if the system is not positive definite.
 */
int wm_oracle_normal_solve(const double *acc, int F, int P, int fit_intercept, double ridge,
                           double *beta)
{
    const int D = F + 1 + P;
    const double n = acc[(size_t)F * D + F];
    double *A = (double *)malloc(sizeof(double) * F * F);
    double *B = (double *)malloc(sizeof(double) * F * P);
    int rc = 0;
    for (int i = 0; i < F; ++i) {
        const double mi = fit_intercept ? acc[(size_t)i * D + F] / n : 0.0;
        for (int j = 0; j < F; ++j) {
            const double sj = acc[(size_t)j * D + F];
            A[i * F + j] = acc[(size_t)i * D + j] - (fit_intercept ? mi * sj : 0.0);
        }
        A[i * F + i] += ridge;
        for (int p = 0; p < P; ++p) {
            const double sy = acc[(size_t)F * D + F + 1 + p];
            B[i * P + p] = acc[(size_t)i * D + F + 1 + p] - (fit_intercept ? mi * sy : 0.0);
        }
    }
    for (int k = 0; k < F && rc == 0; ++k) {
        double d = A[k * F + k];
        for (int t = 0; t < k; ++t) d -= A[k * F + t] * A[k * F + t];
        if (!(d > 0.0)) { rc = -1; break; }
        d = sqrt(d);
        A[k * F + k] = d;
        for (int i = k + 1; i < F; ++i) {
            double s = A[i * F + k];
            for (int t = 0; t < k; ++t) s -= A[i * F + t] * A[k * F + t];
            A[i * F + k] = s / d;
        }
    }
    if (rc == 0) {
        for (int p = 0; p < P; ++p) {
            for (int i = 0; i < F; ++i) {
                double s = B[i * P + p];
                for (int t = 0; t < i; ++t) s -= A[i * F + t] * B[t * P + p];
                B[i * P + p] = s / A[i * F + i];
            }
            for (int i = F - 1; i >= 0; --i) {
                double s = B[i * P + p];
                for (int t = i + 1; t < F; ++t) s -= A[t * F + i] * B[t * P + p];
                B[i * P + p] = s / A[i * F + i];
            }
            double icpt = 0.0;
            if (fit_intercept) {
                icpt = acc[(size_t)F * D + F + 1 + p] / n;
                for (int i = 0; i < F; ++i) icpt -= (acc[(size_t)i * D + F] / n) * B[i * P + p];
            }
            for (int i = 0; i < F; ++i) beta[(size_t)i * P + p] = B[i * P + p];
            beta[(size_t)F * P + p] = icpt;
        }
    }
    free(A);
    free(B);
    return rc;
}

/* ------------------------------------------------------------------ 3-D operators (row a6)
 * Conv3D / MaxPool3D of the reference's convolution.py:164-173 live in tensorflow/keras, absent
 * here: PARITY UNPINNED against the reference's own run.  The restatement follows the published
 * operator definition (valid cross-correlation, stride 1, channels_first) and is pinned against
 * torch.nn.functional.conv3d / max_pool3d on CPU (tests/golden/conv3d_golden.npz), the same
 * operator class with weights in (Cout,Cin,kD,kH,kW) order. */

/* y[b,o,i,j,l] = bias[o] + sum_{c,kd,kh,kw} w[o,c,kd,kh,kw] x[b,c,i+kd,j+kh,l+kw] */
void wm_oracle_conv3d_fwd(const float *x, const float *w, const float *b, double *y, int B,
                          int Cin, int D, int H, int W, int Cout, int kD, int kH, int kW)
{
    const int OD = D - kD + 1, OH = H - kH + 1, OW = W - kW + 1;
    if (OD <= 0 || OH <= 0 || OW <= 0) return;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < B; ++n)
        for (int o = 0; o < Cout; ++o)
            for (int i = 0; i < OD; ++i)
                for (int j = 0; j < OH; ++j) {
                    double *yr = y + ((((size_t)n * Cout + o) * OD + i) * OH + j) * OW;
                    for (int l = 0; l < OW; ++l) yr[l] = b ? (double)b[o] : 0.0;
                    for (int c = 0; c < Cin; ++c)
                        for (int kd = 0; kd < kD; ++kd)
                            for (int kh = 0; kh < kH; ++kh) {
                                const float *xr = x + ((((size_t)n * Cin + c) * D + i + kd) * H + j + kh) * W;
                                const float *wr = w + ((((size_t)o * Cin + c) * kD + kd) * kH + kh) * kW;
                                for (int kw = 0; kw < kW; ++kw) {
                                    const double wv = wr[kw];
                                    for (int l = 0; l < OW; ++l) yr[l] += wv * (double)xr[l + kw];
                                }
                            }
                }
}

/* gx[b,c,p,q,r] = sum_{o,kd,kh,kw} w[o,c,kd,kh,kw] gy[b,o,p-kd,q-kh,r-kw] */
void wm_oracle_conv3d_bwd_data(const float *gy, const float *w, double *gx, int B, int Cin, int D,
                               int H, int W, int Cout, int kD, int kH, int kW)
{
    const int OD = D - kD + 1, OH = H - kH + 1, OW = W - kW + 1;
    memset(gx, 0, sizeof(double) * (size_t)B * Cin * D * H * W);
    if (OD <= 0 || OH <= 0 || OW <= 0) return;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < B; ++n)
        for (int c = 0; c < Cin; ++c)
            for (int o = 0; o < Cout; ++o)
                for (int kd = 0; kd < kD; ++kd)
                    for (int kh = 0; kh < kH; ++kh)
                        for (int kw = 0; kw < kW; ++kw) {
                            const double wv = w[((((size_t)o * Cin + c) * kD + kd) * kH + kh) * kW + kw];
                            for (int i = 0; i < OD; ++i)
                                for (int j = 0; j < OH; ++j) {
                                    const float *gr = gy + ((((size_t)n * Cout + o) * OD + i) * OH + j) * OW;
                                    double *xr = gx + ((((size_t)n * Cin + c) * D + i + kd) * H + j + kh) * W + kw;
                                    for (int l = 0; l < OW; ++l) xr[l] += wv * (double)gr[l];
                                }
                        }
}

/* gw[o,c,kd,kh,kw] = sum_{b,i,j,l} gy[b,o,i,j,l] x[b,c,i+kd,j+kh,l+kw];  gb[o] = sum gy[b,o,...] */
void wm_oracle_conv3d_bwd_wb(const float *x, const float *gy, double *gw, double *gb, int B,
                             int Cin, int D, int H, int W, int Cout, int kD, int kH, int kW)
{
    const int OD = D - kD + 1, OH = H - kH + 1, OW = W - kW + 1;
    const int k3 = kD * kH * kW;
    if (OD <= 0 || OH <= 0 || OW <= 0) return;
#pragma omp parallel for collapse(2) schedule(static)
    for (int o = 0; o < Cout; ++o)
        for (int c = 0; c < Cin; ++c)
            for (int t = 0; t < k3; ++t) {
                const int kw = t % kW, kh = (t / kW) % kH, kd = t / (kW * kH);
                double s = 0.0;
                for (int n = 0; n < B; ++n)
                    for (int i = 0; i < OD; ++i)
                        for (int j = 0; j < OH; ++j) {
                            const float *gr = gy + ((((size_t)n * Cout + o) * OD + i) * OH + j) * OW;
                            const float *xr = x + ((((size_t)n * Cin + c) * D + i + kd) * H + j + kh) * W + kw;
                            for (int l = 0; l < OW; ++l) s += (double)gr[l] * (double)xr[l];
                        }
                gw[((size_t)o * Cin + c) * k3 + t] = s;
            }
    if (gb)
        for (int o = 0; o < Cout; ++o) {
            double s = 0.0;
            for (int n = 0; n < B; ++n) {
                const float *gr = gy + ((size_t)n * Cout + o) * OD * OH * OW;
                for (int l = 0; l < OD * OH * OW; ++l) s += (double)gr[l];
            }
            gb[o] = s;
        }
}

/* 2x2x2 / stride-2 max pool over axes A1..A3 of x (B,A1,A2,A3,C); which = first maximum (0..7,
 * A1-major).  keras.layers.MaxPool3D(pool_size=2, strides=2), padding 'valid' (floor). */
void wm_oracle_maxpool3d_fwd(const float *x, float *y, unsigned char *which, int B, int A1, int A2,
                             int A3, int C)
{
    const int P1 = A1 / 2, P2 = A2 / 2, P3 = A3 / 2;
    for (int n = 0; n < B; ++n)
        for (int p1 = 0; p1 < P1; ++p1)
            for (int p2 = 0; p2 < P2; ++p2)
                for (int p3 = 0; p3 < P3; ++p3)
                    for (int c = 0; c < C; ++c) {
                        float best = 0.0f;
                        int arg = 0;
                        for (int q = 0; q < 8; ++q) {
                            const int a1 = 2 * p1 + (q >> 2), a2 = 2 * p2 + ((q >> 1) & 1),
                                      a3 = 2 * p3 + (q & 1);
                            const float v = x[((((size_t)n * A1 + a1) * A2 + a2) * A3 + a3) * C + c];
                            if (q == 0 || v > best) { best = v; arg = q; }
                        }
                        const size_t o = ((((size_t)n * P1 + p1) * P2 + p2) * P3 + p3) * C + c;
                        y[o] = best;
                        if (which) which[o] = (unsigned char)arg;
                    }
}

/* gradient of the pool: gy routed to the recorded maximum of each window, zero elsewhere */
void wm_oracle_maxpool3d_bwd(const float *gy, const unsigned char *which, float *gx, int B, int A1,
                             int A2, int A3, int C)
{
    const int P1 = A1 / 2, P2 = A2 / 2, P3 = A3 / 2;
    memset(gx, 0, sizeof(float) * (size_t)B * A1 * A2 * A3 * C);
    for (int n = 0; n < B; ++n)
        for (int p1 = 0; p1 < P1; ++p1)
            for (int p2 = 0; p2 < P2; ++p2)
                for (int p3 = 0; p3 < P3; ++p3)
                    for (int c = 0; c < C; ++c) {
                        const size_t o = ((((size_t)n * P1 + p1) * P2 + p2) * P3 + p3) * C + c;
                        const int q = which[o];
                        const int a1 = 2 * p1 + (q >> 2), a2 = 2 * p2 + ((q >> 1) & 1),
                                  a3 = 2 * p3 + (q & 1);
                        gx[((((size_t)n * A1 + a1) * A2 + a2) * A3 + a3) * C + c] = gy[o];
                    }
}
