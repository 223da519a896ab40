// solve_internal.h -- the blocked multi-workgroup normal-equation solve (solve_blocked.hip) behind
// wm_normal_solve_f64 for systems beyond the one-wave kernel (F > 24 or P > 4).
#pragma once

#include "common.h"

// A Cholesky pivot must exceed this fraction of the column's own (centred, ridged) diagonal entry;
// below it the column is explained by the ones before it to rounding error, i.e. collinear.  A
// plain `> 0` test lets an exactly duplicated column through whenever rounding leaves 1e-13.
#define WM_PIVOT_RTOL 1e-11

size_t wm_solve_blocked_workspace_bytes(int F, int P);
int wm_solve_blocked_launch(const double *acc, int F, int P, int fit_intercept, double ridge,
                            double *beta, int *status, void *workspace, hipStream_t st);
