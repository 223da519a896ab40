/*
 * wm.h -- C ABI of libwm.so, the MI355X (gfx950) implementation of the weather-model hot path.
 *
 * The reference (akollgaa/weather-model) has no FFI of its own: its hot path is reached through
 * two LIBRARY OPERATORS called from Python --
 *     torch.nn.Conv2d / torch.nn.functional.conv2d   newConvolution.py:129-133,149-158 (+ :184)
 *     keras.layers.Dense(activation='linear') + MSE  multipleRegression.py:273-298,
 *                                                    singleRegression.py:74-96
 * -- so the drop-in boundary is the operator level.  Every entry point below cites the reference
 * call it replaces.  The reference-side binding is a ctypes stub.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless a comment says host.
 *   - every call is asynchronous on the hipStream_t passed as `stream` (NULL = default stream);
 *     nothing synchronises, nothing allocates, nothing is freed.  The caller owns all buffers,
 *     including the workspaces sized by the *_workspace_bytes() queries.
 *   - return 0 on success, a negative WM_ERR_* code otherwise; wm_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - tensors are contiguous float32 NCHW; conv weights are (Cout,Cin,kH,kW) (PyTorch order, so a
 *     reference state_dict is usable as is).  Convolutions are cross-correlations with valid
 *     padding, stride 1, dilation 1, groups 1 -- the only convention the reference uses.
 */
#ifndef WM_H
#define WM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WM_OK               0
#define WM_ERR_INVALID     -1   /* bad argument (null pointer, non-positive size, shape mismatch) */
#define WM_ERR_UNSUPPORTED -2   /* shape outside what the kernels implement */
#define WM_ERR_WORKSPACE   -3   /* workspace missing or too small */
#define WM_ERR_HIP         -4   /* a HIP runtime call or kernel launch failed */
#define WM_ERR_NO_DEVICE   -5   /* no gfx950 device visible */

/* library version (major*10000 + minor*100 + patch) and last error text of this thread */
int         wm_version(void);
const char *wm_last_error(void);
/* Number of compute units of the current device (negative WM_ERR_* on failure). */
int         wm_device_cus(void);

/* ------------------------------------------------------------------ convolution --------- */

/* Replaces torch.nn.functional.conv2d(x, w, b) as called by nn.Conv2d.forward at
 * newConvolution.py:149,152,155,158 (layers defined :129-133) and the operator convention of
 * keras Conv3D(..., strides=1, padding='valid', data_format='channels_first') at
 * convolution.py:164,166.  y[n,o,i,j] = b[o] + sum_{c,kh,kw} w[o,c,kh,kw] x[n,c,i+kh,j+kw].
 * b may be NULL.  relu != 0 fuses the nn.ReLU the reference applies after every conv
 * (newConvolution.py:150,153,156,159).  y is (N,Cout,H-kH+1,W-kW+1). */
int wm_conv2d_fwd_f32(const float *x, const float *w, const float *b, float *y,
                      int N, int Cin, int H, int W, int Cout, int kH, int kW,
                      int relu, void *stream);

/* Replaces the grad_input leg of autograd for that conv (reached from loss.backward(),
 * newConvolution.py:184): gx[n,c,p,q] = sum_{o,kh,kw} w[o,c,kh,kw] gy[n,o,p-kh,q-kw]. */
int wm_conv2d_bwd_data_f32(const float *gy, const float *w, float *gx,
                           int N, int Cin, int H, int W, int Cout, int kH, int kW,
                           void *stream);

/* The same with the backward of the nn.ReLU that FOLLOWS the layer below (newConvolution.py:150,153,
 * 156,159 under loss.backward(), :184) fused in: ymask is that layer's post-ReLU output, shaped like
 * gx, and gx = ymask > 0 ? gx : 0.  wm_relu_mask_f32 is the stand-alone form (in place),
 * wm_relu_mask_to_f32 writes out = ymask > 0 ? g : 0 to a separate buffer. */
int wm_conv2d_bwd_data_relu_f32(const float *gy, const float *w, const float *ymask, float *gx,
                                int N, int Cin, int H, int W, int Cout, int kH, int kW,
                                void *stream);
int wm_relu_mask_f32(float *gx, const float *ymask, int64_t n, void *stream);
int wm_relu_mask_to_f32(const float *g, const float *ymask, float *out, int64_t n, void *stream);

/* Replaces the grad_weight and grad_bias legs (newConvolution.py:184):
 *   gw[o,c,kh,kw] = sum_{n,i,j} gy[n,o,i,j] x[n,c,i+kh,j+kw],  gb[o] = sum_{n,i,j} gy[n,o,i,j].
 * gb may be NULL.  Partial sums go through `workspace` (float64) and are combined in a fixed
 * order, so results are bitwise reproducible run to run on one device. */
size_t wm_conv2d_bwd_wb_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kH, int kW);
int wm_conv2d_bwd_wb_f32(const float *x, const float *gy, float *gw, float *gb,
                         void *workspace, size_t workspace_bytes,
                         int N, int Cin, int H, int W, int Cout, int kH, int kW,
                         void *stream);

/* Deferred finalize of the same gradient legs (newConvolution.py:184-186: loss.backward() followed at
 * once by optimizer.step()): wm_conv2d_bwd_wb_f32 = a partial-sum pass + a small dependent launch that
 * adds the partials in a fixed order.  These three entry points split the two so that the consumer
 * which runs next anyway does the fixed-order sum:
 *   wm_conv2d_bwd_wb_partials_f32   the partial-sum pass alone; `workspace` (sized as above) then
 *                                   holds float64 partials, layout[0] of them per entry, entry-major
 *                                   ([nw+Cout][layout[0]]) when layout[1] == 1, else part-major.
 *                                   Shapes that take the MFMA weight-gradient kernels (both channel
 *                                   counts >= 16) are WM_ERR_UNSUPPORTED: call wm_conv2d_bwd_wb_f32.
 *   wm_partials_finish_f32          the fixed-order sum -> gw (nw floats), gb (nb floats or NULL).
 *   wm_adam_step_partials_f32       the sum fused into the Adam update of the layer's flat parameter
 *                                   vector [w | b] (count = nw + nb; torch.optim.Adam defaults as in
 *                                   wm_adam_step_f32, newConvolution.py:225); grads_out (count floats
 *                                   or NULL) receives the float32 gradient.
 * partials + finish, and partials + adam_step_partials, give bit for bit what wm_conv2d_bwd_wb_f32
 * (followed by wm_adam_step_f32) gives. */
int wm_conv2d_bwd_wb_partials_f32(const float *x, const float *gy, void *workspace,
                                  size_t workspace_bytes, int N, int Cin, int H, int W, int Cout,
                                  int kH, int kW, int *layout, void *stream);
int wm_partials_finish_f32(const void *partials, int nparts, int entry_major, int nw, int nb,
                           float *gw, float *gb, void *stream);
int wm_adam_step_partials_f32(float *params, const void *partials, int nparts, int entry_major,
                              float *m, float *v, int count, float *grads_out, int step, double lr,
                              double beta1, double beta2, double eps, float grad_scale, void *stream);

/* No counterpart in the reference.  The same gradients as wm_conv2d_bwd_wb_f32 for callers that
 * cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc): the float64 partial sums live in a static device array of
 * WM_CONV2D_NOALLOC_SCRATCH_BYTES inside the library (one copy per GPU, held until the process
 * ends) instead of a caller workspace.  Shapes whose partial sums do not fit
 * (wm_conv2d_bwd_wb_workspace_bytes(...) > wm_conv2d_bwd_wb_noalloc_max_bytes()) are rejected
 * with WM_ERR_UNSUPPORTED.  The call allocates and frees nothing.  Results are bitwise those of
 * wm_conv2d_bwd_wb_f32.  All calls on one GPU share that GPU's array: calls that overlap on one
 * GPU (different streams or threads) race -- serialise them.  The device that owns `stream` must be
 * the current device (WM_ERR_INVALID otherwise: each GPU has its own array).  The first call per
 * device and shape makes host-side queries (the array's address, an occupancy query): make it once
 * before a stream capture begins. */
#define WM_CONV2D_NOALLOC_SCRATCH_BYTES (1u << 20)
size_t wm_conv2d_bwd_wb_noalloc_max_bytes(void);
int wm_conv2d_bwd_wb_noalloc_f32(const float *x, const float *gy, float *gw, float *gb,
                                 int N, int Cin, int H, int W, int Cout, int kH, int kW,
                                 void *stream);

/* 3-D counterpart: replaces keras.layers.Conv3D(filters, kernel_size, strides=1, padding='valid',
 * data_format='channels_first') forward (convolution.py:164,166,170,172) and, reached from
 * CustomModel.train_step's tape.gradient (convolution.py:49-54), its three gradients.
 *   y[b,o,i,j,l] = bias[o] + sum_{c,kd,kh,kw} w[o,c,kd,kh,kw] x[b,c,i+kd,j+kh,l+kw]
 * x is (B,Cin,D,H,W), w is (Cout,Cin,kD,kH,kW) (PyTorch order; a Keras kernel (kD,kH,kW,Cin,Cout)
 * is its permutation (4,3,0,1,2)), y is (B,Cout,D-kD+1,H-kH+1,W-kW+1).  Implemented for at most 64
 * channels on either side and tensors below 2^31 elements (WM_ERR_UNSUPPORTED otherwise).
 * Gradients: gx = sum_{o,k} w gy (full correlation), gw = sum_{b,pos} gy x, gb = sum gy; the
 * weight/bias sums go through float64 partials in `workspace`, combined in a fixed order. */
int wm_conv3d_fwd_f32(const float *x, const float *w, const float *b, float *y,
                      int B, int Cin, int D, int H, int W, int Cout, int kD, int kH, int kW,
                      int relu, void *stream);
int wm_conv3d_bwd_data_f32(const float *gy, const float *w, float *gx,
                           int B, int Cin, int D, int H, int W, int Cout, int kD, int kH, int kW,
                           void *stream);
int wm_conv3d_bwd_data_relu_f32(const float *gy, const float *w, const float *ymask, float *gx,
                                int B, int Cin, int D, int H, int W, int Cout, int kD, int kH,
                                int kW, void *stream);   /* gx = ymask > 0 ? gx : 0, as above */
size_t wm_conv3d_bwd_wb_workspace_bytes(int B, int Cin, int D, int H, int W, int Cout,
                                        int kD, int kH, int kW);
int wm_conv3d_bwd_wb_f32(const float *x, const float *gy, float *gw, float *gb,
                         void *workspace, size_t workspace_bytes,
                         int B, int Cin, int D, int H, int W, int Cout, int kD, int kH, int kW,
                         void *stream);
/* Measurement aid, no counterpart in the reference: byte offset inside the workspace at which
 * wm_conv3d_bwd_wb_f32 leaves three uint64 for this shape -- a magic word ("WMCLOCK1"), the shader
 * cycles and the 100 MHz wall-clock ticks its first workgroup ran for (cycles / ticks * 100 = the
 * clock in MHz the chip held under the kernel's MFMA load) -- or 0 when the shape's kernel leaves
 * none.  bench.py prints the figure beside the MFMA roofline. */
size_t wm_conv3d_bwd_wb_clock_offset(int B, int Cin, int D, int H, int W, int Cout,
                                     int kD, int kH, int kW);

/* Replaces keras.layers.MaxPool3D(pool_size=2, strides=2) (convolution.py:165,167,171,173) and its
 * gradient.  x is (B,A1,A2,A3,C) contiguous and the window runs over A1,A2,A3 ("channels_last",
 * the Keras default the reference leaves in place although its convs are channels_first -- so on a
 * (B,32,14,14,14) conv output it pools the (32,14,14) axes; pass C = 1 and fold the
 * channels into B for a channels_first pool).  y is (B,A1/2,A2/2,A3/2,C); `which` (same shape,
 * bytes, may be NULL in forward-only use) records the first maximum of every window, and the
 * backward pass routes gy to it: gx is (B,A1,A2,A3,C), zero elsewhere. */
int wm_maxpool3d_fwd_f32(const float *x, float *y, unsigned char *which,
                         int B, int A1, int A2, int A3, int C, void *stream);
int wm_maxpool3d_bwd_f32(const float *gy, const unsigned char *which, float *gx,
                         int B, int A1, int A2, int A3, int C, void *stream);

/* The reference's two conv branches end in the same pool and are concatenated for the Dense layers
 * (convolution.py:164-174: MaxPool3D on each branch, layers.concatenate, Flatten).  The pair forms
 * run both branches' pools in ONE launch, and sample b of an output (of a gradient) sits at
 * y + b * y_batch_stride: with y1 = y0 + per-sample size and a stride of twice that, the pools
 * write -- and their backward reads -- the concatenated (B, 2 * flat) matrix directly: no
 * concatenation copy forward, no slice copies backward. */
int wm_maxpool3d_fwd_pair_f32(const float *x0, const float *x1, float *y0, float *y1,
                              unsigned char *which0, unsigned char *which1, int B, int A1, int A2,
                              int A3, int C, int64_t y_batch_stride, void *stream);
int wm_maxpool3d_bwd_pair_f32(const float *gy0, const float *gy1, const unsigned char *which0,
                              const unsigned char *which1, float *gx0, float *gx1, int B, int A1,
                              int A2, int A3, int C, int64_t gy_batch_stride, void *stream);

/* Up to any number of device-to-device copies, 8 per launch: `records` is a HOST array of
 * {void *dst; const void *src; int64 bytes} (read during the call; the records travel as launch
 * arguments, so the call is graph-capturable).  Stages a batch (x1, x2, y) into the fixed input
 * buffers of a captured training step (the host side of convolution.py:46-62's train_step) in one
 * launch.  No counterpart in the reference: Keras feeds its step through tf.data. */
int wm_copy_multi(const void *records, int ncopies, void *stream);

/* ------------------------------------------------------------------ least squares ------- */

/* Sufficient statistics of the least-squares problem that the reference's linear Dense stacks
 * minimise by Adam/MSE (multipleRegression.py:273-298, rows built at :217-234;
 * singleRegression.py:74-96).  With Z = [X | 1 | Y] (D = F+1+P columns),
 *   acc[a*D + b] += sum_{r<n} Z[r][a] * Z[r][b]            (float64, row-major D x D)
 * X is n x F row-major float32, Y is n x P row-major float32.  acc is a RUNNING sum: zero it
 * before the first call; several calls (row chunks, or one per GPU followed by an all-reduce)
 * accumulate.  Products are formed in float32 (MFMA), summed in float32 over at most 1024 rows
 * per accumulator and combined in float64 in a fixed order (bitwise reproducible). */
size_t wm_gram_workspace_bytes(int64_t n, int F, int P);
int wm_gram_f32(const float *X, const float *Y, int64_t n, int F, int P, double *acc,
                void *workspace, size_t workspace_bytes, void *stream);

/* Solves the normal equations held in acc for the affine map the reference's model spans:
 *   beta is (F+1) x P row-major float64, row F is the intercept (0 if fit_intercept == 0).
 * ridge is added to the diagonal of the (centred) X'X.  *status (device int, may be NULL) is set
 * to 0, or to 1 if the system was not positive definite (beta is then left NaN): a Cholesky pivot
 * at or below 1e-11 of its column's own centred diagonal entry counts as zero (a column the ones
 * before it explain to rounding error, i.e. collinear features).  F <= 24 with P <= 4 is solved by
 * one wave in registers and needs no workspace; wider systems (the reference's 432 x 216) are
 * factorised in 64-column blocks over many workgroups in the workspace, which must be 16-byte
 * aligned. */
size_t wm_normal_solve_workspace_bytes(int F, int P);
int wm_normal_solve_f64(const double *acc, int F, int P, int fit_intercept, double ridge,
                        double *beta, int *status,
                        void *workspace, size_t workspace_bytes, void *stream);

/* y[r][p] = sum_f X[r][f] beta[f][p] + beta[F][p]  -- replaces model.predict([x1, x2])
 * (multipleRegression.py:451, singleRegression.py:579) for the fitted linear map. */
int wm_linear_predict_f32(const float *X, const double *beta, float *Yhat,
                          int64_t n, int F, int P, void *stream);

/* ------------------------------------------------------------------ fused model --------- */

/* The reference's conv interpolation model (newConvolution.py:109-163), batched over B grid points.
 * params: the 20566 forward-path parameters, flat, in PyTorch parameter order
 *   conv1.weight(16,1,2,1) conv1.bias conv2.weight(32,16,3,1) conv2.bias conv3.weight(32,32,3,1)
 *   conv3.bias conv4.weight conv4.bias linear1.weight(64,192) linear1.bias linear2.weight(6,64)
 *   linear2.bias                                     (newConvolution.py:129-136)
 * x: B x 2 x 6 (rows: start, end; newConvolution.py:100-104); out: B x 6.
 * wm_model_fwd_f32 replaces CustomModel.forward (newConvolution.py:146-163) for every sample. */
int    wm_model_param_count(void);
int    wm_model_fwd_f32(const float *params, const float *x, float *out, int64_t B, void *stream);

/* Forward + backward: replaces model(x) + loss_fn + loss.backward() (newConvolution.py:180-184).
 * Pass y (B x 6) for the fused nn.MSELoss (mean over all 6*B_global outputs; *loss receives this
 * rank's share sum_local (out-y)^2 / (6*B_global)), or dout (B x 6) = dL/dout from autograd; the
 * other must be NULL.  grads: 20566 floats, same layout as params (overwritten, bitwise
 * reproducible).  B_global >= B is the batch the mean runs over (data parallel: all-reduce(sum)
 * grads afterwards).  The workspace (caller-owned, contents undefined afterwards) holds the
 * per-workgroup gradient tiles, their float64 sums and a lane-ordered copy of linear1.weight. */
size_t wm_model_workspace_bytes(int64_t B);
int    wm_model_train_f32(const float *params, const float *x, const float *y, const float *dout,
                          float *out, float *loss, float *grads, void *workspace,
                          size_t workspace_bytes, int64_t B, int64_t B_global, void *stream);

/* wm_model_train_f32 (fused MSE form) fed straight from the resident volume: sample b is the one
 * wm_gather_triplets_f32 would have produced for index idx[b] (or idx0 + b when idx is NULL) --
 * CustomDataset.__getitem__, newConvolution.py:86-106, followed by model(x), loss_fn, backward
 * (:180-184) -- without writing x (B,2,6) and y (B,6) first (144 B per sample each way).  vol is
 * (T,Z,Y,X,6); the other arguments are those of the two calls it replaces. */
int    wm_model_train_gather_f32(const float *params, const float *vol, const int64_t *idx,
                                 int64_t idx0, int T, int Z, int Y, int X, int St, int Sz, int Sy,
                                 int Sx, int time_offset, int d_mid, int d_end, float *out,
                                 float *loss, float *grads, void *workspace,
                                 size_t workspace_bytes, int64_t B, int64_t B_global, void *stream);

/* Multi-tensor Adam for a whole model: replaces self.optimizer.apply_gradients(...) of the Conv3D
 * model's train_step (convolution.py:54; compile(optimizer="adam"), :185).  `table` is a HOST array
 * (read during the call; the records travel to the kernel as launch arguments, 16 per launch)
 * of ntensors records {float *p; const float *g; float *m; float *v; int64 count; int64 first_block}
 * (48 bytes each): a workgroup takes 1024 consecutive elements of one tensor, first_block is the
 * running sum of ceil(count / 1024) and total_blocks its end.  `step` is a DEVICE int holding the number of
 * updates made so far.  advance != 0: the call first moves it on by one (a one-thread launch) and
 * then uses it in the bias corrections -- all on the stream, so the call can be captured in a HIP
 * graph and replayed.  advance == 0 uses the counter as it stands: for a caller that spreads one
 * step over several calls (a layer updated from its gradient hook on a side stream while the rest
 * of the backward pass still runs, the other tensors at the end of the step) -- the first call of
 * the step advances, the others must be ordered after it on the device.
 * keras_eps != 0: keras.optimizers.Adam's update (m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2);
 * var -= m alpha / (sqrt(v) + eps), alpha = lr sqrt(1 - b2^t) / (1 - b1^t)) -- epsilon OUTSIDE the
 * bias correction, as the reference's optimizer has it; 0: torch.optim.Adam's, as wm_adam_step_f32. */
int    wm_adam_multi_f32(const void *table, int ntensors, int64_t total_blocks, int *step,
                         double lr, double beta1, double beta2, double eps, int keras_eps,
                         int advance, void *stream);

/* torch.optim.Adam step with the reference's defaults semantics (newConvolution.py:185,225):
 * no weight decay, no amsgrad; step counts from 1; grads are multiplied by grad_scale first.
 * lr, betas and eps are doubles, as torch holds them: the bias corrections 1 - beta^step and the
 * step size lr / (1 - beta1^step) are formed in double and rounded to float once. */
int    wm_adam_step_f32(float *params, const float *grads, float *m, float *v, int64_t count,
                        int step, double lr, double beta1, double beta2, double eps,
                        float grad_scale, void *stream);

/* ------------------------------------------------------------------ fields (8f next rows) */

/* One pass over a float32 field: out[0..4] = min, max, sum, sum of squares, count of non-NaN
 * elements (float64).  Replaces the per-file float(ds[var].max()/.min()/.mean()/.std()) of
 * data.py:59-60,76-77 and multipleRegression.py:392-393,414-415: mean = sum/count,
 * std = sqrt(sumsq/count - mean^2).  accumulate != 0 merges into the values already in `out`
 * (running statistics over files).  x must be 16-byte aligned.  Fixed-order combine. */
size_t wm_field_stats_workspace_bytes(int64_t n);
int wm_field_stats_f32(const float *x, int64_t n, double *out, int accumulate,
                       void *workspace, size_t workspace_bytes, void *stream);

/* y[i] = (x[i] - offset) / divisor in float32 with a true division: bitwise what numpy computes for
 * normalize ((data - lo) / (hi - lo), newConvolution.py:47-49, convolution.py:80-82,
 * multipleRegression.py:190-192) and standardize ((data - mean) / std, :195-197) on float32 data.
 * x and y 16-byte aligned; y may alias x. */
int wm_affine_scale_f32(const float *x, float *y, int64_t n, float offset, float divisor,
                        void *stream);

/* The sample gather of CustomDataset.__getitem__ / gather_data (newConvolution.py:86-106) for B
 * samples at once.  vol is (T,Z,Y,X,V) float32 (the V variables of a grid point contiguous).
 * Sample index id (idx[b], or idx0 + b when idx is NULL) decodes against shape (St,Sz,Sy,Sx):
 * column = id % Sx, row = id / Sx % Sy, height = id / (Sx Sy) % Sz,
 * time = id / (Sx Sy Sz) % St + time_offset.  xout (B,2,V) = [vol[time], vol[time + d_end]],
 * yout (B,V) = vol[time + d_mid]  (the reference uses d_mid = 3, d_end = 6). */
int wm_gather_triplets_f32(const float *vol, const int64_t *idx, int64_t idx0, float *xout,
                           float *yout, int64_t B, int V, int T, int Z, int Y, int X,
                           int St, int Sz, int Sy, int Sx, int time_offset, int d_mid, int d_end,
                           void *stream);

/* Row construction of the regression generators from a RESIDENT volume, replacing the np.append
 * growth of WeatherDataGenerator.__getitem__ / gather_batches (multipleRegression.py:217-257), the
 * per-sample NetCDF reads of WeatherGenerator (singleRegression.py:251-299) and the random cubes of
 * DataGenerator (convolution.py:118-147).  vol is (files, nvars, Z, Y, X) float32.  Row r takes the
 * (cz,cy,cx) cube of every variable at origin (z0,y0,x0) from files f, f + d_end, f + d_mid:
 *   x1out[r*xstride + .] = cube(vol[f]),  x2out[r*xstride + .] = cube(vol[f + d_end]),
 *   yout[r*p + .] = cube(vol[f + d_mid]),   p = nvars*cz*cy*cx, flattened (var, dz, dy, dx).
 * For the regression rows [x1 | x2] pass x2out = x1out + p and xstride = 2p.
 * origins != NULL: int32 (rows, 4) = (f, z0, y0, x0) per row (device memory).
 * origins == NULL: the reference's region walk (multipleRegression.py:259-270) in closed form,
 * `batch` rows per generator batch, starting from the generator state (index0; r0, r1, r2 = the
 * raw region_index) with period `length` and region moduli (dim_z, dim_y, dim_x) = data_dim[1:]:
 * batch b has file (index0+b) % length * batch + j and origin
 * (R0_b % dim_z, R1_b % dim_y, R2_b % dim_x), R0_b = (index0+b)/length, R1_b = R0_{b-1}/dim_z,
 * R2_b = R1_{b-1}/dim_y for b >= 1. */
int wm_gather_cubes_f32(const float *vol, int files, int nvars, int Z, int Y, int X,
                        int cz, int cy, int cx, int d_mid, int d_end,
                        const int32_t *origins, int64_t rows, int batch, int64_t length,
                        int64_t index0, int64_t r0, int64_t r1, int64_t r2,
                        int dim_z, int dim_y, int dim_x,
                        float *x1out, float *x2out, int64_t xstride, float *yout, void *stream);

/* The same rows from a resident volume in ANY layout: strides[5] (host memory) are the element
 * strides of the (file, variable, z, y, x) axes, all >= 0.  wm_gather_cubes_f32 is this call with the
 * row-major strides of the reference's arrays.  With cube 1 every element of a row comes from its
 * own (file, variable) plane, a 4-byte gather per element in the row-major layout; a volume kept
 * point-major -- (Z, Y, X, files, nvars), strides {nvars, 1, Y*X*files*nvars, X*files*nvars,
 * files*nvars} -- makes the `batch` rows of a generator batch one contiguous read
 * (multipleRegression.py:217-257 slices the same values file by file). */
int wm_gather_cubes_strided_f32(const float *vol, int files, int nvars, int Z, int Y, int X,
                                const int64_t *strides, int cz, int cy, int cx, int d_mid,
                                int d_end, const int32_t *origins, int64_t rows, int batch,
                                int64_t length, int64_t index0, int64_t r0, int64_t r1, int64_t r2,
                                int dim_z, int dim_y, int dim_x,
                                float *x1out, float *x2out, int64_t xstride, float *yout,
                                void *stream);

/* The Gram of the rows that wm_gather_cubes_strided_f32 would build -- [x(t) | x(t + d_end)] ;
 * x(t + d_mid) along the region walk of WeatherDataGenerator (multipleRegression.py:217-270) -- read
 * STRAIGHT from the resident volume: no X, no Y, one pass.  CUBE_SIZE 1 (the reference's setting,
 * multipleRegression.py:35) and 1..4 variables (F = 2 nvars, P = nvars; WM_ERR_UNSUPPORTED
 * otherwise: gather the rows and call wm_gram_f32).  `strides` are the element strides of the
 * volume's (file, variable, z, y, x) axes; the walk arguments are those of wm_gather_cubes_f32.
 * acc is the (F+1+P)^2 float64 accumulator of wm_gram_f32 (same entries, same fixed-order float64
 * combination, bitwise reproducible; a different summation order than gather + wm_gram_f32).
 * wm_linear_metrics_walk_f32 is the matching loss / mae / msle pass (see wm_linear_metrics_f32). */
size_t wm_gram_walk_workspace_bytes(int nvars);
int wm_gram_walk_f32(const float *vol, int files, int nvars, int Z, int Y, int X,
                     const int64_t *strides, int d_mid, int d_end, int64_t rows, int batch,
                     int64_t length, int64_t index0, int64_t r0, int64_t r1, int64_t r2, int dim_z,
                     int dim_y, int dim_x, double *acc, void *workspace, size_t workspace_bytes,
                     void *stream);
int wm_linear_metrics_walk_f32(const float *vol, int files, int nvars, int Z, int Y, int X,
                               const int64_t *strides, int d_mid, int d_end, int64_t rows, int batch,
                               int64_t length, int64_t index0, int64_t r0, int64_t r1, int64_t r2,
                               int dim_z, int dim_y, int dim_x, const double *beta, int want_msle,
                               float msle_eps, double *sums, void *workspace, size_t workspace_bytes,
                               void *stream);

/* The Dense layers of the Conv3D model (convolution.py:175-176: Dense(flat, relu), Dense(prod(
 * input_shape))) and their gradients, replacing keras.layers.Dense / its autodiff for the batch
 * sizes the reference trains with.  x is (M, K), W is (N, K) = (units, inputs), b is (N) or NULL,
 * y is (M, N); relu != 0 applies max(., 0).  N % 4 == 0 and K % 4 == 0 (WM_ERR_UNSUPPORTED
 * otherwise), all tensors 16-byte aligned.
 * wm_dense_bwd_f32: gx (M, K) = gy W, masked where mask_below <= 0 when mask_below != NULL (the
 * post-ReLU activation that fed this layer, i.e. x itself: the ReLU backward of the layer below
 * fused into the store); gW (N, K) = gy^T x; gb (N) = column sums of gy.  gx may be NULL, or gW
 * together with gb, to skip that part; gb is formed by the weight-gradient pass, so gb without gW
 * and a call with neither gx nor gW are rejected (WM_ERR_INVALID).  The workspace holds the split
 * partial sums of gx. */
int wm_dense_fwd_f32(const float *x, const float *W, const float *b, float *y, int M, int N, int K,
                     int relu, void *stream);
size_t wm_dense_bwd_workspace_bytes(int M, int N, int K);
int wm_dense_bwd_f32(const float *x, const float *W, const float *gy, const float *mask_below,
                     float *gx, float *gW, float *gb, void *workspace, size_t workspace_bytes,
                     int M, int N, int K, void *stream);

/* Loss, metrics and loss gradient of the Conv3D model's train step in one pass, replacing
 * keras' compile(loss='mean_squared_error', metrics=['mean_absolute_error',
 * 'mean_squared_logarithmic_error']) as fit() evaluates it (convolution.py:176-180, train_step
 * :46-62).  pred is (B, P); y is (B, P).  broadcast != 0 scores every prediction against every
 * target of the batch -- what the reference's (B, 1, P) targets do to the (B, P) predictions
 * (SURVEY.md 2b); broadcast == 0 pairs row i with row i.  out3 = {mean (y-p)^2, mean |y-p|,
 * mean (log1p(max(p,eps)) - log1p(max(y,eps)))^2} (device memory, 3 floats); grad (may be NULL) is
 * d out3[0] / d pred, shape (B, P).  B <= 4096.  A thread sums its pair terms in float32 over
 * blocks of at most 64 targets (the gradient's target sum in float64 term by term); blocks, threads
 * and workgroups are combined in float64 in a fixed order. */
size_t wm_mse_metrics_workspace_bytes(int64_t P);
int wm_mse_metrics_f32(const float *pred, const float *y, int B, int64_t P, int broadcast,
                       float eps, float *grad, float *out3, void *workspace,
                       size_t workspace_bytes, void *stream);

/* What model.fit / model.evaluate report per epoch for the linear stacks -- the loss (mean squared
 * error) and the compiled metrics mae / msle of the CURRENT coefficients over the rows
 * (multipleRegression.py:292-296: compile(loss=mean_squared_error, weighted_metrics=['mae','msle']);
 * singleRegression.py:90-94; evaluated by fit, :341-350 / :500-509) -- as one streaming pass over X
 * and Y: pred = [X | 1] beta is formed in registers (float64 accumulation as in
 * wm_linear_predict_f32, rounded to float32 like the model's output) and never written.
 *   sums[0] += sum (pred - y)^2      sums[1] += sum |pred - y|
 *   sums[2] += sum (log1p(max(pred, eps)) - log1p(max(y, eps)))^2   (only when want_msle != 0)
 *   sums[3] += n * P
 * sums is a float64[4] on the device; per-workgroup float64 partials go through `workspace` and are
 * combined in a fixed order.  wm_linear_metrics_f32 takes (F + 1) * P' <= 4096 (P' = P rounded up to one of 1, 2, 4, 6, 8, 16,
 * 32) and P <= 32
 * (wm_linear_metrics_supported; WM_ERR_UNSUPPORTED beyond: form the predictions with a GEMM and call
 * wm_residual_sums_f32 on them, count = n * P). */
size_t wm_linear_metrics_workspace_bytes(void);
int wm_linear_metrics_supported(int F, int P);
int wm_linear_metrics_f32(const float *X, const float *Y, const double *beta, int64_t n, int F,
                          int P, int want_msle, float msle_eps, double *sums, void *workspace,
                          size_t workspace_bytes, void *stream);
int wm_residual_sums_f32(const float *pred, const float *Y, int64_t count, int want_msle,
                         float msle_eps, double *sums, void *workspace, size_t workspace_bytes,
                         void *stream);

/* The same sums, and / or model.predict, for WIDE linear maps -- the reference's singleRegression
 * model is Dense(432, linear) -> Dense(216) on 6^3 cubes (singleRegression.py:21,33,74-85; predict
 * :579; loss and metrics :90-94, evaluated by fit :500-509): F % 4 == 0, P % 4 == 0, P <= 224,
 * F <= 8192 (wm_linear_metrics_wide_supported).  pred = [X | 1] beta is formed on the fp32 MFMA
 * (beta rounded to float32, float32 accumulation -- the arithmetic of the reference's own Dense
 * layers) and, straight from the accumulators,
 *   scored against Y into sums[4] as wm_linear_metrics_f32 does, when sums != NULL, and / or
 *   written to Yhat (n x P float32), when Yhat != NULL  (Y may be NULL when sums is).
 * X, Y, Yhat and workspace must be 16-byte aligned.  The workspace holds the per-workgroup float64
 * partials and the float32 coefficient panels packed for the kernel (rebuilt by every call). */
int wm_linear_metrics_wide_supported(int F, int P);
size_t wm_linear_metrics_wide_workspace_bytes(int F, int P);
int wm_linear_metrics_wide_f32(const float *X, const float *Y, const double *beta, int64_t n, int F,
                               int P, int want_msle, float msle_eps, double *sums, float *Yhat,
                               void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ exchange step ------- */

/* The one collective of the data-parallel path (SURVEY.md 8b ABI block, 8e) for hosts without
 * torch.distributed: rows / samples are sharded over one process per GPU, and between wm_gram_f32
 * and wm_normal_solve_f64 (the float64 Gram `acc`) or between wm_model_train_f32 and
 * wm_adam_step_f32 (the flat float32 gradient) every rank sums ONE flat buffer in place.  The
 * reference has no counterpart (its MirroredStrategy, singleRegression.py:488-496, is dead code).
 * RCCL (librccl.so, opened with dlopen at the first call: WM_ERR_UNSUPPORTED if absent) over xGMI;
 * one communicator per process, bound to the device current at wm_comm_init.
 *   wm_comm_unique_id  rank 0 fills WM_COMM_ID_BYTES of HOST memory; ship them to every rank
 *                      (file, socket, environment) -- the library does no rendezvous of its own;
 *   wm_comm_init       collective over the nranks processes; the current device is this rank's GPU;
 *   wm_allreduce_sum   buf (device, count elements of WM_F32 / WM_F64) <- sum over ranks, in place,
 *                      asynchronous on `stream`; one call per step on the whole buffer;
 *   wm_comm_size       ranks of the live communicator, 0 if none;  wm_comm_destroy  frees it. */
#define WM_COMM_ID_BYTES 128
#define WM_F32 0
#define WM_F64 1
int wm_comm_unique_id(void *id128);
int wm_comm_init(int rank, int nranks, const void *unique_id);
int wm_comm_size(void);
int wm_allreduce_sum(void *buf, size_t count, int dtype, void *stream);
int wm_comm_destroy(void);

/* ------------------------------------------------------------------ synthetic data ------ */

/* Counter-based generator: element i of `out` depends only on (seed, offset + i).  Same hash as
 * weather_model_amd/synth.py.  No reference counterpart (the reference reads WRF files). */
int wm_synth_uniform_f32(float *out, int64_t count, uint64_t seed, uint64_t offset,
                         float lo, float hi, void *stream);
/* Rows [row_offset, row_offset+n) of the synthetic regression problem of SURVEY.md 8d:
 * X ~ U(0,1) (dist 0), N(0,1) (dist 1) or U(0,1) from paired draws (dist 2: two 24-bit uniforms per
 * hash, the stream of weather_model_amd/synth.py uniform_paired01 -- half the generator's integer
 * work, used where generation is inside the timed region); Y = X beta + intercept + N(0, noise^2).
 * beta: device, F x P float32. */
int wm_synth_regression_f32(float *X, float *Y, int64_t n, int F, int P, uint64_t seed,
                            uint64_t row_offset, const float *beta, float intercept,
                            float noise, int dist, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WM_H */
