// gram_internal.h -- the Gram kernels behind wm_gram_f32, one translation unit per family.
//   gram.hip        F = 16 / F = 64 with P = 1 (BASELINE configs 2 and 5), the solve, predict
//   gram_small.hip  F + P <= 15 on the VALU, HBM-bound streaming (the reference's own
//                   multipleRegression shape F = 2, P = 1: multipleRegression.py:35,273-286)
//   gram_quad.hip   F = 4 RX, P = 4 RY for a few (RX, RY) (eight variables at once: F = 16, P = 8):
//                   gram_f16's 4x4x1 MFMA scheme with Y as further operand registers, HBM-bound
//                   (+ gram_quadu_kernel<RX, RY>: padded pieces, any alignment -- F = 2p, P = p for
//                   p = 6 .. 12, one target with up to 32 predictors, ...)
//   gram_mid.hip    F + P + 1 <= 64 otherwise: operands straight from global memory into MFMA
//                   registers, any alignment (the drop-ins' multi-variable fits, F = 2p, P = p)
//   gram_syrk.hip   every other shape: LDS-staged blocked Z'Z on fp32 MFMA (the reference's
//                   singleRegression shape F = 432, P = 216: singleRegression.py:21,33,74-85)
#pragma once

#include "common.h"

// F + P <= 15 and one of the instantiated (F, P) pairs, X 16-byte aligned, Y aligned to its vector
// width.  Workspace: per-workgroup float64 partials.
bool wm_gram_small_supported(const float *X, const float *Y, int F, int P);
size_t wm_gram_small_workspace_bytes(int F, int P);
int wm_gram_small_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                         double *partial, hipStream_t st);

// Any F, P >= 1.  Workspace: per-wave float64 64 x 64 slabs (and 64 constant bytes in front).
size_t wm_gram_syrk_workspace_bytes(long long n, int F, int P);
int wm_gram_syrk_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                        void *workspace, hipStream_t st);

// F + P + 1 <= 64, any alignment.  Workspace: per-wave float64 slabs.
bool wm_gram_mid_supported(int F, int P);
size_t wm_gram_mid_workspace_bytes(int F, int P);
int wm_gram_mid_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                       double *slabs, hipStream_t st);

// (F, P) in {(16,8), (16,4), (8,8), (8,4)}, X aligned to F bytes, Y to P bytes.  Workspace:
// per-workgroup float64 partials.
bool wm_gram_quad_supported(const float *X, const float *Y, int F, int P);
size_t wm_gram_quad_workspace_bytes(int F, int P);
int wm_gram_quad_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                        double *partial, hipStream_t st);

// ceil(F / 4) + ceil(P / 4) <= 9 and P <= 12: rows that are not four equal pieces (or whose pointers
// are not aligned to them), any 4-byte alignment.  Workspace: per-workgroup float64 partials.
bool wm_gram_quadu_supported(int F, int P);
size_t wm_gram_quadu_workspace_bytes(int F, int P);
int wm_gram_quadu_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                         double *partial, hipStream_t st);
