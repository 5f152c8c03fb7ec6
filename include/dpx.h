/*
 * dpx.h -- C ABI of the MI355X (gfx950) proximal-solver backend for Delta-Prox.
 *
 * The reference (princeton-computational-imaging/Delta-Prox @ v2) is pure Python/PyTorch and has
 * no FFI of its own: its ADMM/PGD hot path is the eager torch-op sequences cited next to every
 * entry point below (paths relative to the reference checkout).  This header is the boundary a
 * maintainer would bind instead (INTEGRATION.md shows the ctypes stub): plain device pointers and
 * sizes, a hipStream_t, int status codes; no torch types, no allocation, no exceptions.
 *
 * Conventions
 *   - every image tensor is contiguous NCHW fp32 in device memory, owned by the caller;
 *   - every call is asynchronous on `stream`; workspaces are caller-owned and sized by the
 *     *_bytes() queries; tables (twiddles, OTFs, denominators) are immutable after creation and
 *     may be shared by concurrent calls on different streams;
 *   - return value 0 = ok, <0 = error, text via dpx_last_error() (thread-local);
 *   - per-image scalars (rho, lambda, sigma) are device arrays of length B so that a whole
 *     iteration can be replayed from a hipGraph while the schedule advances.
 *   - spectral tables use an opaque internal half-spectrum layout; always size/produce them
 *     through this API.
 */
#ifndef DPX_H
#define DPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dpx_stream_t; /* hipStream_t */

#define DPX_OK 0
#define DPX_ERR_ARG (-1)
#define DPX_ERR_LAUNCH (-2)
#define DPX_ERR_UNSUPPORTED (-3)

int dpx_version(void);
const char* dpx_last_error(void);
/* optional per-kernel timing with HIP events on the launch stream (used by bench.py's roofline leg):
 * enable, run, then dpx_timing_report writes "kernel_name launches total_ms" lines and resets the log. */
int dpx_timing_enable(int on);
int dpx_timing_report(char* buf, size_t cap);

/* ------------------------------------------------------------------------------------------ */
/* tuning knobs                                                                                */
/* ------------------------------------------------------------------------------------------ */
/* The reference has no counterpart (it has one eager code path per op).  Every dispatcher of this library that chooses between
 * kernel variants or launch geometries reads an integer knob from ONE per-process registry; 0 always means "the library's own
 * rule".  dpx_tune_set takes effect at the next call that reads the knob (no caching), so a test can drive both branches of a
 * dispatcher inside one process; dpx_tune_get reads a knob back; dpx_tune_count / dpx_tune_name enumerate the registry.  The
 * environment variable in the last column only supplies the knob's INITIAL value (read once per process).  Results are
 * numerically equivalent under every setting (bit-identical where the table says so); the knobs trade speed only.
 *
 *   knob                 values                                                         initial value from
 *   cg_fused_max_b       dpx_cg_masked_fft: batches up to this size take the fused        DPX_CG_FUSED_MAX_B (8)
 *                        4-launch CG iteration, larger ones the step-by-step sequence
 *                        (0 = always step by step; at most 32)
 *   cg_split_update      1 = fused CG with its own x / r update launch + flag copy        DPX_CG_SPLIT_UPDATE
 *   cg_unfused           1 = always the step-by-step CG sequence                          DPX_CG_UNFUSED
 *   comm_allgather_ring  1 = ncclAllGather instead of world-1 direct sends                DPX_COMM_ALLGATHER=ring
 *   iter_rows            1 = streaming row kernel, 2 = lock-step ring-buffer kernel,      DPX_ITER_ROWS=seq|lockstep|par
 *                        3 = row-parallel kernel (k_iter_rows_par: the rows of a band side by
 *                        side in one 16-wave workgroup; 256 / 512 / 1024-wide rows)
 *                        (dpx_admm_iter_config overrides); 1 and 3 are bit-identical
 *   iter_par_max_rows    launches of at most this many rows (planes x H) take the row-parallel      DPX_ITER_PAR_MAX_ROWS
 *                        kernel (0 = the library's rule, 8192; < 0 = never)
 *   iter_band, iter_r    bands per plane (streaming) / rows per band (lock-step)          DPX_ITER_BAND, DPX_ITER_R
 *   ds_rpb,              geometry of the one-off fp64 data-spectrum pass                  DPX_DS_RPB,
 *   ds_row_threads,                                                                       DPX_DS_ROW_THREADS,
 *   ds_col_threads                                                                        DPX_DS_COL_THREADS
 *   cg_rows_per_wg,      rows / columns per workgroup of the fused CG matvec's transform      DPX_CG_ROWS_PER_WG,
 *   cg_cols_per_wg       kernels (0 = enough workgroups to cover the chip's 256 CUs)             DPX_CG_COLS_PER_WG
 *   cg_gram_small        fused CG, B <= 8: 1 = the register Gram kernel (default rule),          DPX_CG_GRAM_SMALL
 *                        2 = the 32 x 32 slab kernel of larger batches
 *   cg_no_hint           1 = the fused CG does not look at the stop flag early at the iteration   DPX_CG_NO_HINT
 *                        the previous solve exited at (same result, seven more empty launches)
 *   cg_event_wait        1 = the fused CG waits for its stop flag behind a HIP event per iteration       DPX_CG_EVENT_WAIT
 *                        (rounds 2 - 4; a marker packet, ~5 us of idle stream each) instead of spinning
 *                        on the tag the test kernel stores into host-coherent memory (same result)
 *   pnp_cg_no_fold       1 = dpx_admm_cg_pnp_iter keeps k_zupdate + k_bx_pack_in and k_bx_unpack_out, k_lincomb,  DPX_PNP_CG_NO_FOLD
 *                        k_rhs, k_cgm_start as six launches instead of its head and tail passes (same bits;
 *                        dpx_admm_cg_pnp_iter_folds = 0); 2 = folded, but the head pass is not issued ahead of the
 *                        host's look at the CG's stop flag
 *   unroll_bwd_staged    dpx_admm_unrolled_backward: 0 = the two-kernel backward iteration on          DPX_UNROLL_BWD_STAGED
 *                        power-of-two planes (k_bwd_rows), else the image-domain fused stage; 2 = the
 *                        image-domain fused stage everywhere; 1 = the rhs stage and the z stage of
 *                        neighbouring iterations as two passes (same gradients up to round-off)
 *   unroll_bwd_band      rows per band of k_bwd_rows (0 = two lock-step rounds of a workgroup's rows)     DPX_UNROLL_BWD_BAND
 *   unroll_bwd_par_max_rows  backward launches of at most this many rows (planes x H) take the row-parallel kernel    DPX_UNROLL_BWD_PAR_MAX_ROWS
 *                        (k_bwd_rows_par: the rows of a band side by side in one 16-wave workgroup; 0 = the library's
 *                        rule, 12288; < 0 = never: the lock-step bands of k_bwd_rows); gradients equal to round-off
 *   unroll_bwd_fold_finish  1 = the fused backward stage's last workgroup finishes the iteration's       DPX_UNROLL_BWD_FOLD_FINISH
 *                        reductions instead of a finishing launch (slower: measured, kept for A/B)
 *   ffdnet_presplit      split-f16 inference (dpx_ffdnet_forward_bf16, mode 3): 1 = activations       DPX_FFDNET_PRESPLIT
 *                        travel between the layers as pre-split binary16 operand planes written by
 *                        the producing layer (bit-identical results; measured +-0 .. -5 %: off)
 *   cg_wave_fft          fused CG on 320 x 320 / 384 x 384 planes: 2 = the size-generic transform        DPX_CG_WAVE_FFT
 *                        kernels instead of the one-wave register transforms (same result to round-off)
 *   conv_tile_rows       split-arithmetic 3x3 layers: 8 = 8-row workgroup tiles (two workgroups per CU)      DPX_CONV_TILE_ROWS
 *                        instead of 16-row ones; same results, measured 2 % slower where it could help: off
 *   generic_interleaved  size-generic transforms (planes off the register-radix path): 1 = passes in place   DPX_GENERIC_INTERLEAVED
 *                        on interleaved LDS images (k_cols_il: eight columns per workgroup; k_rows_r2c_il /
 *                        k_rows_c2r_il: a row per one-wave workgroup); 0 = one LDS line per sequence, two buffers
 *                        (k_rows_r2c / k_cols / k_rows_c2r); 2 / 3 = rows / columns only on the new form, 4 = rows
 *                        eight to a 512-thread workgroup (A/B; all settings give bit-identical results)
 *   iter_band_min_rows   shortest band (rows) the streaming row kernel of the two-kernel iteration may     DPX_ITER_BAND_MIN_ROWS
 *                        use when few planes must fill the chip (0 = the library's rule: 4 rows, 2 for launches
 *                        of fewer than 8 planes of 1024-wide rows; bit-identical across band counts)
 */
int dpx_tune_count(void);
const char* dpx_tune_name(int i);                       /* NULL beyond dpx_tune_count() */
int dpx_tune_set(const char* name, int value);          /* DPX_ERR_ARG for a name the registry does not hold */
int dpx_tune_get(const char* name, int* value);

/* The CG solver's three switches as one typed call (cg_fused_max_b <= 32, cg_split_update, cg_unfused); a negative argument
 * leaves that switch unchanged.  Both branches of dpx_cg_masked_fft -- fused (B <= fused_max_b) and step by step -- compute the
 * reference's cg() (linalg/solve/solver_cg.py:56-136) with the same stop rule and exit iteration.                            */
int dpx_cg_config(int fused_max_b, int split_update, int unfused);

/* ------------------------------------------------------------------------------------------ */
/* spectral plans                                                                              */
/* ------------------------------------------------------------------------------------------ */
/* Twiddle tables for H x W planes (fp64-accurate, stored fp32).  Replaces the per-call planning
 * inside torch.fft.fftn / ifftn (reference dprox/linop/conv.py:33-34, proxfn/sum_square.py:150-152). */
size_t dpx_fft_table_bytes(int H, int W);
int dpx_fft_table_init(void* table, int H, int W, dpx_stream_t stream);
/* workspace holding the half spectrum of P = B*C planes */
size_t dpx_spectrum_bytes(int P, int H, int W);

/* ------------------------------------------------------------------------------------------ */
/* PSF -> OTF (setup time)                                                                     */
/* ------------------------------------------------------------------------------------------ */
/* psf2otf: reference dprox/utils/psf2otf.py:11-40 as called by conv._FB (linop/conv.py:23-29):
 * zero-pad, shift the centre floor(k/2) to the origin, DFT over (H, W, C).  Evaluated directly
 * in fp64 on the device (no FFT round-off).  `psf` is device fp64 [kh][kw][kc].
 *   otf  (nullable): complex table used by dpx_fft_conv  (dpx_otf_bytes)
 *   diag (nullable): real |OTF|^2 table = conv.get_diag (linop/conv.py:46-53)  (dpx_diag_bytes);
 *                    diag = (accumulate ? diag : 0) + weight * |OTF|^2                         */
size_t dpx_otf_bytes(int C, int H, int W);
size_t dpx_diag_bytes(int C, int H, int W);
int dpx_psf2otf(const double* psf, int kh, int kw, int kc, int C, int H, int W,
                void* otf, void* diag, float weight, int accumulate, dpx_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Fourier-domain operators                                                                    */
/* ------------------------------------------------------------------------------------------ */
/* y = real(ifft2(OTF * fft2(x)))  (conj_otf=0: conv.forward, linop/conv.py:31-35)
 * y = real(ifft2(conj(OTF) * fft2(x)))  (conj_otf=1: conv.adjoint, linop/conv.py:37-41)         */
int dpx_fft_conv(const float* x, float* y, const void* otf, int conj_otf, int B, int C, int H, int W,
                 const void* table, void* spectrum_ws, dpx_stream_t stream);

/* T proximal-gradient iterations in place (algo/pgd.py:26-54 with f = ||k (*) x - b||^2, g a closed-form prox):
 *     x <- prox_g( x - rho_t[b] (K^T K x - K^T b),  alpha * lam_t[b] )
 * gram_otf = the |OTF|^2 table (dpx_otf_bytes), ktb = K^T b (nullable = 0), rho_tab / lam_tab = [T][B] device arrays
 * (lam_tab nullable = 0), prox in {DPX_PROX_NORM1, DPX_PROX_NONNEG, DPX_PROX_SUMSQ}.  The spectrum stays resident between
 * iterations: one column kernel + one row kernel per iteration (the row kernel finishes the inverse transform, steps,
 * applies the prox, and starts the next forward transform on registers).  Power-of-two planes only: ask dpx_pgd_supported. */
int dpx_pgd_supported(int H, int W, int prox);
int dpx_pgd_run(float* x, const float* ktb, const void* gram_otf, int prox, float alpha, const float* rho_tab,
                const float* lam_tab, int T, int B, int C, int H, int W, const void* table, void* spectrum_ws,
                dpx_stream_t stream);

/* spec (+)= op(OTF) * fft2(b), evaluated once per solve in fp64 and stored as a packed fp32 half spectrum
 * (dpx_spectrum_bytes).  This is the Fourier transform of the constant part of the x-update's
 * right-hand side, sum over Omega of K^T b (proxfn/sum_square.py:126-132, where the reference
 * re-evaluates it with two fp32 FFTs every iteration); otf nullable = identity.                    */
size_t dpx_data_spectrum_ws_bytes(int P, int H, int W);
int dpx_data_spectrum(const float* b, const void* otf, int conj_otf, void* spec_out, int accumulate,
                      int B, int C, int H, int W, void* ws, dpx_stream_t stream);

/* spectral table helpers (setup time).  Tables use an opaque per-(H,W) layout:
 *   dpx_table_to_full / from_full : real diag table <-> full [C][H][W] array (get_diag interop, linop/conv.py:46-53,
 *                                    user-supplied diagonals of BlackBox operators, linop/blackbox.py:74-75)
 *   dpx_denominator_pack          : dd = interleaved (d0 + c0, d1 + c1) for dpx_fourier_solve               */
int dpx_table_to_full(const void* table, float* full, int C, int H, int W, dpx_stream_t stream);
int dpx_table_from_full(const float* full, void* table, int C, int H, int W, dpx_stream_t stream);
/* OTF table from a complex OTF given on the full [C][H][W] grid (Hermitian: the transform of a real kernel image):
 * conv_doe's per-call PSF -> OTF (dprox/linop/conv.py:59-80, psf2otf2), the transform itself being dpx_cfft2.   */
int dpx_otf_from_full(const void* full, void* otf, int C, int H, int W, dpx_stream_t stream);
size_t dpx_denominator_bytes(int C, int H, int W);
int dpx_denominator_pack(const void* d0, float c0, const void* d1, float c1, void* dd, int C, int H, int W,
                         dpx_stream_t stream);

/* x = real(ifft2((fft2(rhs) + spec_add + eps) / (dd.x + rho_b*dd.y + eps)))
 * least_squares.solve_direct, frequency branch -- proxfn/sum_square.py:137-152, with
 *   dd.x = sum over Omega of |OTF|^2 (+ constant diagonals),  dd.y = the same over Psi (dpx_denominator_pack);
 * spec_add (nullable) is a dpx_data_spectrum result: the data part of the right-hand side kept in the
 * Fourier domain, so that rhs only carries rho * sum_i K_i^T (v_i - u_i).
 * c0/c1 of dpx_denominator_pack add the constant diagonals of identity linops
 * (Variable.get_diag, linop/variable.py:47-59, and the `+ rho` of :147-148); rho is device [B]. */
int dpx_fourier_solve(const float* rhs, float* x, const void* spec_add, const void* dd,
                      const float* rho, float eps, int B, int C, int H, int W,
                      const void* table, void* spectrum_ws, dpx_stream_t stream);

/* out = irFFT2[ rFFT2(g) / (d0 + c0 + rho_b (d1 + c1) + eps) ]: the (self-adjoint) linear part of the x-update, i.e.
 * dpx_fourier_solve without data spectrum and without eps in the numerator.  It is the vector-Jacobian product of
 * least_squares.solve_direct w.r.t. its right-hand side (backward of proxfn/sum_square.py:123-156; the reference gets it
 * from PyTorch autograd through fftn / division / ifftn).                                                            */
int dpx_fourier_apply_inv(const float* g, float* out, const void* dd, const float* rho, float eps, int B, int C, int H, int W,
                          const void* table, void* ws, dpx_stream_t stream);

/* complex64 2-D FFT of P planes, optionally centred (ifftshift -> fft2 -> fftshift) and orthonormal: utils.fft2 / ifft2,
 * dprox/utils/misc.py:164-193 (used by CS-MRI style operators mask * fft2(x)).  Out of place, any size.        */
int dpx_cfft2(const void* in, void* out, int inverse, int centred, int ortho, int P, int H, int W, const void* table,
              dpx_stream_t stream);

/* Closed-form CS-MRI data-term update in the (centred) Fourier domain, in place:
 *   z[f] <- (lam_b * z[f] + y[f]) / (1 + lam_b * num_psi)   where mask[f] != 0,      z[f] unchanged elsewhere
 * = the body of csmri._prox between its two transforms (dprox/proxfn/fast/csmri.py:14-25).  z, y: complex64
 * [B, n_per_image]; mask: one byte per element, [B, n_per_image] (mask_images = B) or [1, n_per_image]
 * (mask_images = 1, shared by the batch); lam: [B].                                                          */
int dpx_csmri_update(void* z, const void* y, const unsigned char* mask, int mask_images, const float* lam, float num_psi, int B,
                     long n_per_image, dpx_stream_t stream);

/* weighted_sum_squares._prox (dprox/proxfn/sum_square.py:70-75): out = (ktb + lam_b v) / (diag + lam_b), ktb / diag one image
 * (ktb_images / diag_images = 1) or a batch.                                                                         */
int dpx_wss_prox(const float* v, const float* ktb, int ktb_images, const float* diag, int diag_images, const float* lam, float* out,
                 int B, long n_per_image, dpx_stream_t stream);

/* mul_color (dprox/linop/mul.py:13-43): channel mixing by a spectral response srf [C][C2]:
 *   forward  (transpose = 0): out[n][c2][p] = sum_c  srf[c][c2] x[n][c][p]      (= srf.T @ x), x has C  channels, out C2
 *   adjoint  (transpose = 1): out[n][c][p]  = sum_c2 srf[c][c2] x[n][c2][p]     (= srf   @ x), x has C2 channels, out C   */
int dpx_mul_color(const float* x, const float* srf, float* out, int transpose, int B, int C, int C2, long hw, dpx_stream_t stream);

/* Building blocks of the closed-form single-image super-resolution data term `sisr` (dprox/proxfn/fast/sr.py:45-77):
 *   dpx_upsample_zero : s-fold zero-filling upsampler (sr.py:117-126), planes x h x w -> planes x (h sf) x (w sf)
 *   dpx_cplx_mul      : out[b,i] = (conj_a ? conj(a) : a)[(a_images > 1 ? b : 0), i] * bb[b, i]  (complex64; FBC * F(STy))
 *   dpx_sisr_update   : FR is a [2][B][C][H][W] complex64 buffer: half 0 = FR (input), half 1 receives FX (sr.py:66-72):
 *         FBR  = mean over the sf x sf aliases of FB * FR,   invW = mean over the aliases of |FB|^2
 *         FX   = (FR - conj(FB) * FBR / (invW + I lam_b)) / (I lam_b + 1e-9)
 *     FB holds fb_planes = 1, C or B*C planes.                                                                   */
int dpx_upsample_zero(const float* y, float* out, int sf, long planes, int h, int w, dpx_stream_t stream);
int dpx_cplx_mul(void* out, const void* a, const void* bb, int conj_a, int B, long n_per_image, int a_images, dpx_stream_t stream);
int dpx_sisr_update(void* FR, const void* FB, int fb_planes, const float* lam, float I, int sf, int B, int C, int H, int W,
                    dpx_stream_t stream);

/* out[b, i] = x[b, i] * w[(w_images > 1 ? b : 0), i]: diagonal operators in the image domain -- mosaic's Bayer mask
 * (dprox/linop/subsample.py:17-31) and mul_elementwise (dprox/linop/mul.py:46-73); forward == adjoint.          */
int dpx_mul(const float* x, const float* w, float* out, int B, long n_per_image, int w_images, dpx_stream_t stream);

/* out[b, i] = w[(w_images > 1 ? b : 0), i] * a[b, i] with a, out complex64 and w real fp32: the k-space mask of a
 * subsampled Fourier operator  A x = mask * fft2(x),  A^H y = ifft2(mask * y)  (the CS-MRI forward model the reference
 * writes with eager ops, dprox/proxfn/fast/csmri.py:19-23; config 4's operator through the LinOp plugin surface).   */
int dpx_cplx_scale(void* out, const void* a, const float* w, int B, long n_per_image, int w_images, dpx_stream_t stream);

/* out = sum_i coef[i] * x[i] over n <= 4 operands that are each real (float32) or complex (complex64) arrays of
 * `n_elems` elements; out is complex64 (out_complex = 1) or the REAL PART of the sum as float32 (out_complex = 0).
 * The complex-iterate arithmetic of the CS-MRI solver: `z - u`, `x + u`, `u + x - z` (dprox/contrib/csmri.py:161-169)
 * and deep_prior's `v.real` (dprox/proxfn/pnp/prior.py:79).  x[i] may alias out when both have the same type.        */
int dpx_cplx_lincomb(void* out, int out_complex, int n, const void* const* x, const int* x_complex, const float* coef,
                     long n_elems, dpx_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* spatial operators and fused ADMM steps                                                      */
/* ------------------------------------------------------------------------------------------ */
/* grad(x, dim) forward = x[n+1]-x[n] (circular), adjoint = y[n-1]-y[n]; dim 0 = H, 1 = W.
 * Equal (to fp32 rounding) to the reference's FFT evaluation -- linop/grad.py:8-23 + conv.py:31-41. */
int dpx_grad(const float* x, float* y, int dim, int adjoint, int B, int C, int H, int W, dpx_stream_t stream);

/* out = sum_i coef[i] * (coef_b[i] ? coef_b[i][b] : 1) * x[i]   (n <= 4; x[i] may alias out).
 * The AXPY family of the reference's eager ops: sum.forward (linop/sum.py:13-19), scale
 * (linop/scale.py:20-31), `v - u`, `u + Kx - v` (algo/admm.py:51,57), CG updates
 * (linalg/solve/solver_cg.py:114-129).  n_per_batch = C*H*W.                                   */
int dpx_lincomb(float* out, int n, const float* const* x, const float* coef, const float* const* coef_b,
                int B, long n_per_batch, dpx_stream_t stream);

/* batched dot: out[b] = <x[b], y[b]>  -- bdot, linalg/solve/solver_cg.py:7-22 */
int dpx_bdot(const float* x, const float* y, float* out, int B, long n_per_batch, void* ws, dpx_stream_t stream);
size_t dpx_bdot_ws_bytes(int B, long n_per_batch);
/* B x B Gram matrix of the rows of r ([B, n]) -- for the spectral-norm stop rule of cg()
 * (torch.linalg.norm(ravel(r), 2), solver_cg.py:103-104)                                       */
int dpx_bgram(const float* r, float* out, int B, long n_per_batch, void* ws, dpx_stream_t stream);
/* The same three primitives in float64: the reference's solvers are dtype-generic torch code and its tests solve float64 systems
 * at rtol 1e-8 (tests/linalg/test_linear_solver.py:57-80) -- a float64 right-hand side stays float64 on the device
 * (cg / cg2 / pcg of dprox.linalg.solve, solver_cg.py:56-233).  ws: dpx_bdot_f64_ws_bytes for both dpx_bdot_f64 and dpx_bgram_f64. */
size_t dpx_bdot_f64_ws_bytes(int B, long n_per_batch);
int dpx_bdot_f64(const double* x, const double* y, double* out, int B, long n_per_batch, void* ws, dpx_stream_t stream);
int dpx_bgram_f64(const double* r, double* out, int B, long n_per_batch, void* ws, dpx_stream_t stream);
int dpx_lincomb_f64(double* out, int n, const double* const* x, const double* coef, const double* const* coef_b, int B,
                    long n_per_batch, dpx_stream_t stream);
/* out[0] = max_i |x_i| -- the stop rule of pcg (torch.linalg.vector_norm(r, inf), solver_cg.py:207-228); is_f64 selects the element
 * type of x, out and ws (256 elements)                                                                                          */
int dpx_absmax(const void* x, void* out, long n, int is_f64, void* ws, dpx_stream_t stream);

/* Device-side control of cg() (linalg/solve/solver_cg.py:95-129): the host issues a whole solve without a read-back.
 * `state` (dpx_cg_state_bytes) = float gamma[B], gamma_prev[B], beta[B], pAp[B], tol2[B]; int done, n_done, it, pad.
 *   dpx_cg_init      tol2_i = (rtol * ||b_i||)^2 from bnorm2[i] = <b_i, b_i>; done = 0                        (:96, 1 <= B <= 64)
 *   dpx_cg_test      stop rule :103-104 on the B x B residual Gram matrix (dpx_bgram): lambda_max(G) <= min_i tol2_i, decided
 *                    by an LDL^T factorisation of  tau^2 I - G  in float64; converged -> done = 1, n_done = it (the iterate
 *                    freezes: the kernels below return at once); else gamma = diag G, beta = gamma / gamma_prev (:112), ++it
 *   dpx_cg_direction p = r + beta_b p                                                                          (:111-115)
 *   [caller: Ap = A(p) with its own kernels;  dpx_bdot(p, Ap, state + 3B floats)]                             (:118-122)
 *   dpx_cg_update    alpha_b = gamma_b / pAp_b ; x += alpha p ; r -= alpha Ap                                   (:123-127) */
int dpx_zero(void* p, size_t bytes, dpx_stream_t stream);          /* torch.zeros_like of the solvers' fresh iterates (solver_cg.py:83-85) */
size_t dpx_cg_state_bytes(int B);
int dpx_cg_init(void* state, const float* bnorm2, float rtol, int B, dpx_stream_t stream);
int dpx_cg_test(void* state, const float* gram, int B, dpx_stream_t stream);
int dpx_cg_direction(float* p, const float* r, void* state, int B, long n_per_batch, dpx_stream_t stream);
int dpx_cg_update(float* x, float* r, const float* p, const float* Ap, void* state, int B, long n_per_batch, dpx_stream_t stream);
/* One call = one whole CG solve of config 4's x-update, (A^H A + n_identity rho_b I) x = b with A = mask * fft2 (centred,
 * orthonormal), x0 = 0: least_squares.solve_cg (proxfn/sum_square.py:158-197) over cg() for the subsampled-Fourier data term.
 * Kernels only (dpx_cfft2, mask^2, dpx_cg_*); the host side polls the device's `done` flag two iterations late through a pinned
 * buffer.  Returns the exit iteration (>= 0; max_iters when not converged) or a negative status.  1 <= B <= 64.             */
size_t dpx_cg_masked_fft_ws_bytes(int B, int H, int W, int mask_images);
/* 1 when dpx_cg_masked_fft takes this batch (B <= 64, planes within its LDS-resident transforms); else use dpx_cg_* step by step */
int dpx_cg_masked_fft_supported(int B, int H, int W);
int dpx_cg_masked_fft(float* x, const float* b, const float* mask, int mask_images, const float* rho, float n_identity, float rtol,
                      int max_iters, int B, int H, int W, const void* table, void* ws, dpx_stream_t stream);

/* proximal operators, ProxFn.prox with the scaled/translated wrappers (proxfn/base.py:12-27,55-64):
 *   out = P(v - off, lam_b * alpha) + off      off nullable
 * kinds: norm1 soft-threshold (proxfn/norm.py:6-19), nonneg (proxfn/nonneg.py:10-11),
 *        sum_squares / norm2  v/(1+2 lam) (proxfn/sum_square.py:26-27, norm.py:26-27)          */
#define DPX_PROX_NORM1 0
#define DPX_PROX_NONNEG 1
#define DPX_PROX_SUMSQ 2
#define DPX_PROX_EXTERNAL 3 /* z-update only: v <- K x + u (the denoiser runs next), u untouched */
int dpx_prox(int kind, const float* v, float* out, const float* lam, float alpha, const float* off,
             int B, long n_per_batch, dpx_stream_t stream);

/* Backward of dpx_prox for v = prox(d): gd = J(d)^T g (soft-threshold / nonneg: a mask; sum-squares: a scale) and
 * dlam = d prox / d lam evaluated at d (soft-threshold: -sign(d) on the pass band; nonneg: 0; sum-squares:
 * -2 (d - offset) / (1 + 2 lam)^2), so that grad_lam[b] = alpha * <g_b, dlam_b>.  Same argument meaning as dpx_prox;
 * gd and / or dlam may be NULL.  (Reference: autograd through proxfn/norm.py:6-27, nonneg.py:10-11, sum_square.py:26-27.) */
int dpx_prox_bwd(int kind, const float* d, const float* g, float* gd, float* dlam, const float* lam, float alpha, const float* offset,
                 int B, long n_per_batch, dpx_stream_t stream);

/* one Psi term of the ADMM splitting (algo/admm.py:26-36): linop K_i and prox of g_i */
#define DPX_LIN_IDENTITY 0
#define DPX_LIN_GRAD_H 1
#define DPX_LIN_GRAD_W 2
typedef struct dpx_term {
  int32_t linop;    /* DPX_LIN_* */
  int32_t prox;     /* DPX_PROX_* */
  float alpha;      /* lam multiplier (`alpha * fn`, proxfn/base.py:78-82) */
  int32_t reserved; /* flags: DPX_TERM_NO_DUAL (dpx_admm_iter_rows / dpx_admm_run / dpx_admm_zupdate_rhs; ignored elsewhere) */
  const float* lam; /* device [B] */
  float* v;         /* state v_i [B,C,H,W] */
  float* u;         /* state u_i [B,C,H,W] */
  float* u_out;     /* dpx_admm_iter_rows only: updated u_i (double-buffered, must differ from u) */
} dpx_term;
#define DPX_MAX_TERMS 4
/* half-quadratic splitting on the two-kernel iteration (algo/hqs.py:4-20 = the ADMM iteration with the duals pinned to zero): the
 * contents of u are ignored (finite values required: start from zeros) and the next right-hand side is rho sum_i K_i^T v_i;
 * u / u_out are still read / written (scratch).  Set on every term of the call or on none.                                      */
#define DPX_TERM_NO_DUAL 1
/* the incoming duals u are all zero (the state comes straight from ADMM.initialize, algo/admm.py:61-67): dpx_admm_iter_rows /
 * the FIRST iteration of a dpx_admm_run call need not stream them from HBM (only row 0 of plane 0 of every u must hold zeros).
 * Set on every term of the call or on none.                                                                                 */
#define DPX_TERM_U_ZERO 2
/* ADMM in the update order v, x, u (algo/admm.py:103-120, ADMM_vxu) on the two-kernel iteration: the planes passed as u / u_out carry
 * q_i = -u_i - v_i; a row pass forms t = q + x (the dual update with the fresh x), v = prox(K x + t) (the NEXT iteration's
 * v-update), writes q' = t - v and hands  rho' sum K_i^T (v - t)  to the next x-update.  Set on every term of the call or on none;
 * excludes DPX_TERM_NO_DUAL.                                                                                                 */
#define DPX_TERM_VXU 4

/* rhs = ktb + sum_i rho_b * K_i^T (v_i - u_i)   -- proxfn/sum_square.py:126-135 with
 * b_i = v_i - u_i from algo/admm.py:51.  ktb = sum over Omega of K^T offset (constant per solve). */
/* the same right-hand side for a state straight from ADMM.initialize (v_i = K_i x0, u_i = 0; algo/admm.py:61-67) WITHOUT that state:
 * rhs = rho_b sum_i K_i^T K_i x0, the same differences in the same order; linops[i] in {identity, grad_H, grad_W}                          */
int dpx_admm_rhs_fresh(float* rhs, const float* x0, const float* rho, const int* linops, int nterms, int B, int C, int H, int W,
                       dpx_stream_t stream);
int dpx_admm_rhs(float* rhs, const float* ktb, const float* rho, const dpx_term* terms, int nterms,
                 int B, int C, int H, int W, dpx_stream_t stream);

/* for every term: d = K_i x + u_i ; v_i = prox_i(d) ; u_i = d - v_i   -- algo/admm.py:54-57 */
int dpx_admm_zupdate(const float* x, const dpx_term* terms, int nterms,
                     int B, int C, int H, int W, dpx_stream_t stream);

/* dpx_admm_zupdate of iteration t and dpx_admm_rhs of iteration t + 1 (rho_next, [B]) in ONE pass: 8 instead of 12 plane passes for
 * two gradient terms (the adjoint stencils recompute v - u' at the left / upper neighbour from x and u).  Closed-form proxes only;
 * every dual must be double-buffered (terms[i].u_out != terms[i].u); dual = 1: the right-hand side is formed from the updated duals
 * (ADMM: the caller swaps u / u_out), 0: from the incoming ones (half-quadratic splitting, algo/hqs.py: duals not advanced, u_out is
 * scratch and is not written; with DPX_TERM_NO_DUAL on a term its incoming dual counts as zero and is not fetched).  emit_v = 0: v_i is not stored either -- nothing inside the loop reads it (the next right-hand side is
 * formed here); the caller's LAST z / dual stage is a dpx_admm_zupdate, which writes the final v_i, u_i.  Bit-identical v, u', rhs to the two calls
 * (algo/admm.py:49-59 -- the z / dual update -- followed by proxfn/sum_square.py:126-135 -- the next x-update's offset).           */
int dpx_admm_zupdate_rhs(const float* x, const dpx_term* terms, int nterms, float* rhs, const float* ktb, const float* rho_next,
                         int dual, int emit_v, int B, int C, int H, int W, dpx_stream_t stream);

/* ---- fused backward stages of the iteration (config 5, unrolled training; the reference uses PyTorch autograd through
 * the eager ops of algo/admm.py:49-59).  Per-image scalar gradients are reduced deterministically; `ws` has
 * dpx_admm_bwd_ws_bytes.                                                                                              */
typedef struct dpx_bwd_term {
  int32_t linop, prox;
  float alpha;
  int32_t reserved;
  const float* lam;    /* [B] */
  const float* v;      /* saved forward output v_i = prox(K_i x + u_i)                                  */
  const float* gv;     /* incoming gradient w.r.t. v_i (nullable = 0)                                     */
  const float* gu_new; /* incoming gradient w.r.t. the updated dual u_i' = K_i x + u_i - v_i (nullable)   */
  float* gu;           /* out: gradient w.r.t. the incoming dual u_i                                       */
} dpx_bwd_term;
size_t dpx_admm_bwd_ws_bytes(int B, int C, int H, int W);
/* z/dual stage: gx = sum_i K_i^T g_d_i, terms[i].gu = g_d_i, glam[i*B + b] = d loss / d lam_i[b]               */
int dpx_admm_zupdate_bwd(float* gx, const dpx_bwd_term* terms, int nterms, float* glam, int B, int C, int H, int W, void* ws,
                         dpx_stream_t stream);
/* x stage: grho[b] = -<g_rhs_b, (sum_i K_i^T K_i) x_b> with g_rhs = dpx_fourier_apply_inv(gx)                   */
int dpx_admm_solve_rho_grad(const float* g_rhs, const float* x, const int* linops, int nterms, float* grho, int B, int C, int H, int W,
                            void* ws, dpx_stream_t stream);
/* rhs stage: gv[i] = rho_b K_i g, gu[i] = -gv[i] (either may be NULL), grho[b] = <g_b, rhs_b> / rho_b            */
int dpx_admm_rhs_bwd(const float* g, const float* rhs, const float* rho, const int* linops, int nterms, float* const* gv,
                     float* const* gu, float* grho, int B, int C, int H, int W, void* ws, dpx_stream_t stream);

/* Gradient of the Fourier x-update w.r.t. the OTF of a convolutional data term whose PSF is trained through the solver
 * (end-to-end optics, reference README.md:93-116 with conv_doe, linop/conv.py:81-156; there PyTorch autograd through fft2):
 * G[c,k] (+)= (1/HW) sum_b [ conj(A) Y - 2 Re(A conj X) O ],  A = fft2(g_rhs), X = fft2(x), Y = fft2(offset) (X or Y may be NULL:
 * with X = NULL it is the gradient of the plain product conv_doe.forward / adjoint -- A, Y = fft2 of input and output gradient),
 * O = the OTF; full complex spectra in natural order, unnormalised transforms (dpx_cfft2).  G = dL/dRe O + i dL/dIm O.          */
int dpx_otf_grad(const void* A, const void* X, const void* Y, const void* O, void* G, int B, int C, int H, int W, int accumulate,
                 dpx_stream_t stream);

/* ---- unrolled ADMM with closed-form proxes: T iterations forward / backward without returning to the host language
 * (specialization/unroll.py:14-58 over algo/admm.py:49-59; the per-stage calls above, sequenced on the C side).
 * hist (dpx_admm_unrolled_hist_bytes): per iteration [rhs][x][v_0..v_{n-1}][u_0..u_{n-1}] planes of B*C*H*W floats -- the
 * forward's outputs are its last iteration's x / v_i / u_i planes.  rho_tab [T][B], lam_tabs[i] [T][B] (device).
 * backward: gx / gv_in[i] / gu_in[i] = gradients w.r.t. the final x, v_i, u_i (NULL = 0); out: gv0[i], gu0[i] (initial split /
 * dual variables), grho [T][B], glam [T][n][B], goff[k] = gradient w.r.t. the k-th Omega offset (off_otf[k]: its OTF table,
 * NULL = identity; goff[k] NULL = not wanted).
 * fresh_x0 (nullable; planes on the two-kernel iteration only): the initial state is ADMM.initialize(fresh_x0) untouched (algo/admm.py:61-67:
 * v_i = K_i x0, u_i = 0) and need not have been computed -- the first right-hand side is formed from x0 (dpx_admm_rhs_fresh), the first
 * iteration does not stream the zero duals; v0[i] are not dereferenced, u0[i] must hold zeros in row 0 of plane 0 of every image.       */
size_t dpx_admm_unrolled_hist_bytes(int nterms, int T, int B, int C, int H, int W);
int dpx_admm_unrolled_forward(float* hist, const float* const* v0, const float* const* u0, const int* linops, const int* proxes,
                              const float* alphas, int nterms, const float* rho_tab, const float* const* lam_tabs, int T,
                              const void* spec_add, const void* dd, float eps, int B, int C, int H, int W, const void* table,
                              void* spectrum_ws, const float* fresh_x0, dpx_stream_t stream);
size_t dpx_admm_unrolled_bwd_ws_bytes(int nterms, int B, int C, int H, int W);
int dpx_admm_unrolled_backward(const float* hist, const float* gx, const float* const* gv_in, const float* const* gu_in,
                               float* const* gv0, float* const* gu0, float* grho, float* glam, float* const* goff,
                               const void* const* off_otf, int n_off, const int* linops, const int* proxes, const float* alphas,
                               int nterms, const float* rho_tab, const float* const* lam_tabs, int T, const void* dd, float eps,
                               int B, int C, int H, int W, const void* table, void* spectrum_ws, void* ws, dpx_stream_t stream);
/* bf16 mode of the same training step (BASELINE config 5; specialize(..., method='unroll', dtype='bf16')): the iteration runs in
 * fp32 working planes (`work`, dpx_admm_unrolled_work_bytes_bf16) and what the backward pass needs -- rhs, x, v_i of every
 * iteration -- is kept in a bf16 history, (2 + n) x 2 bytes per pixel and iteration instead of (2 + 2n) x 4 (power-of-two planes: the
 * row kernel of the two-kernel iteration writes the slots itself).  The final state goes to x_out / v_out[i] / u_out[i] (fp32).
 * Backward: as above with ws sized by dpx_admm_unrolled_bwd_ws_bytes_bf16; its stage kernels read the 16-bit slots directly.   */
size_t dpx_admm_unrolled_hist_bytes_bf16(int nterms, int T, int B, int C, int H, int W);
size_t dpx_admm_unrolled_work_bytes_bf16(int nterms, int B, int C, int H, int W);
int dpx_admm_unrolled_forward_bf16(void* hist_bf16, void* work, float* x_out, float* const* v_out, float* const* u_out,
                                   const float* const* v0, const float* const* u0, const int* linops, const int* proxes,
                                   const float* alphas, int nterms, const float* rho_tab, const float* const* lam_tabs, int T,
                                   const void* spec_add, const void* dd, float eps, int B, int C, int H, int W, const void* table,
                                   void* spectrum_ws, const float* fresh_x0, dpx_stream_t stream);
size_t dpx_admm_unrolled_bwd_ws_bytes_bf16(int nterms, int B, int C, int H, int W);
int dpx_admm_unrolled_backward_bf16(const void* hist_bf16, const float* gx, const float* const* gv_in, const float* const* gu_in,
                                    float* const* gv0, float* const* gu0, float* grho, float* glam, float* const* goff,
                                    const void* const* off_otf, int n_off, const int* linops, const int* proxes, const float* alphas,
                                    int nterms, const float* rho_tab, const float* const* lam_tabs, int T, const void* dd, float eps,
                                    int B, int C, int H, int W, const void* table, void* spectrum_ws, void* ws, dpx_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* two-kernel fused ADMM iteration (power-of-two planes)                                       */
/* ------------------------------------------------------------------------------------------ */
/* One iteration of algo/admm.py:49-59 as  cols -> rows :
 *   dpx_admm_iter_cols : spectrum in -> column FFT -> (+ data spectrum) / denominator -> inverse column FFT -> spectrum out
 *   dpx_admm_iter_rows : inverse row FFT (x of this iteration) -> z / dual update of every term (lam of this iteration)
 *                        -> rho_next * sum_i K_i^T (v_i - u_i) -> forward row FFT -> spectrum out (input of the next cols)
 * x and v_i are only written when requested (x_out non-null / emit_v), u_i is double-buffered (terms[i].u -> u_out);
 * rho_next = NULL on the last iteration (no right-hand side is produced); there emit_v = 2 with x_out asks for x ALONE (no z / dual
 * update at all: terms[i].v / u_out are not written).  dpx_rfft_rows seeds the loop with the row
 * transform of the first right-hand side (dpx_admm_rhs).  Spectrum buffers: dpx_spectrum_bytes / 2 each.            */
int dpx_admm_iter_supported(int H, int W, const dpx_term* terms, int nterms);
int dpx_rfft_rows(const float* x, void* spec, int B, int C, int H, int W, const void* table, dpx_stream_t stream);
/* the seed in one pass: spec = row transform of rho_b sum_i K_i^T (v_i - u_i)  (= dpx_admm_rhs with ktb = NULL, then dpx_rfft_rows;
 * algo/admm.py:51 + proxfn/sum_square.py:126-135 for the first x-update of a solve); same support as dpx_admm_iter_rows */
int dpx_admm_seed_rows(void* spec, const float* rho, const dpx_term* terms, int nterms, int B, int C, int H, int W, const void* table,
                       dpx_stream_t stream);
/* the same seed for a state that comes straight from ADMM.initialize (algo/admm.py:61-67: v_i = K_i x0, u_i = 0): v_i - u_i is
 * recomputed from x0 (bit-identical), one image read instead of 2 nterms; terms[i].v / .u are not dereferenced */
int dpx_admm_seed_rows_fresh(void* spec, const float* rho, const float* x0, const dpx_term* terms, int nterms, int B, int C, int H, int W,
                             const void* table, dpx_stream_t stream);
int dpx_admm_iter_cols(const void* spec_in, void* spec_out, const void* spec_add, const void* dd, const float* rho, float eps,
                       int B, int C, int H, int W, const void* table, dpx_stream_t stream);
int dpx_admm_iter_rows(const void* spec_in, void* spec_out, const dpx_term* terms, int nterms, const float* rho_next,
                       float* x_out, int emit_v, int B, int C, int H, int W, const void* table, dpx_stream_t stream);
/* the hot loop of Algorithm.iters (algo/base.py:149-156) for n_iters iterations, entirely on the C side:
 * rho_tab [total_iters][B], lam_tabs[i] [total_iters][B]; returns 0/1 = which of terms[i].u / u_out holds the
 * current u_i afterwards (<0 = error); x / v_i are written by the call's final iteration when emit_last is 1.
 * emit_last = 2: the caller wants x alone (Algorithm.solve returns state[0], algo/base.py:141): when the call ends the solve
 * (it0 + n_iters == total_iters) its final row pass stores x and nothing else -- v_i and the duals stay those of the iteration
 * before; when it does not end the solve, 2 means 1.                                                                             */
/* Fused stages of LinearizedADMM (algo/admm.py:78-100) and PockChambolle (algo/pc.py:6-40) for K_i in {identity, grad_H, grad_W}:
 *   dpx_split_rhs   rhs = ktb + rho_b sum_i K_i^T (x - K_i^T q_i),  q_i = z_i (mode 0, PC: terms[i].v) or (K_i x - v_i) + u_i (mode 1, LADMM)
 *                   -- the right-hand side least_squares.rhs builds from the b_i of pc.py:24-25 / admm.py:84-88 (ktb nullable)
 *   dpx_pc_dual     z_i += r K_i xbar ; z_i -= r prox_i(z_i, r alpha_i), r = terms[i].lam[b]        (pc.py:13-19), in place on terms[i].v */
int dpx_split_rhs(float* rhs, const float* ktb, const float* x, const float* rho, const dpx_term* terms, int nterms, int mode,
                  int B, int C, int H, int W, dpx_stream_t stream);
int dpx_pc_dual(const float* xbar, const dpx_term* terms, int nterms, int B, int C, int H, int W, dpx_stream_t stream);
/* test / tuning hook: rows_mode 0 automatic, 1 streaming row kernel, 2 lock-step row kernel, 3 row-parallel kernel (launches of a few
 * planes; bit-identical to 1); bands_per_plane 0 = automatic */
int dpx_admm_iter_config(int rows_mode, int bands_per_plane);
/* query: bands per plane the streaming row kernel would walk for a launch of `planes` planes of H x W (> 0; the groups of
 * planes x bands always fill whole workgroups -- a partition exists for every plane count), 0 = W is not a row length of that kernel */
int dpx_admm_iter_bands(int planes, int H, int W);
/* tuning hint: the following dpx_admm_* calls are one of `chains` sub-batch chains of a solve that run concurrently on separate
 * streams (the images of a batch never exchange data, algo/admm.py:49-59 acts per image): band lengths are chosen for the planes
 * of all chains together.  1 = a call has the GPU to itself (default).  Results never depend on it.                          */
int dpx_admm_iter_share(int chains);
int dpx_admm_run(void* spec_a, void* spec_b, const void* spec_add, const void* dd, const dpx_term* terms, int nterms,
                 const float* rho_tab, const float* const* lam_tabs, float eps, int it0, int n_iters, int total_iters,
                 float* x_out, int emit_last, int B, int C, int H, int W, const void* table, dpx_stream_t stream);
/* dpx_admm_run for `nchains` sub-batches of one solve at once, each on its own stream with its own spectrum buffers, data spectrum,
 * terms (state views) and [total_iters][B_chain] schedule tables; dd, the problem's tables and the iteration range are shared.
 * The chains' launches are interleaved and their column passes ordered by events, so that the chains advance together.  Returns the
 * dual-buffer parity (equal for all chains), < 0 on error.  (algo/admm.py:49-59 acts per image: splitting the batch changes nothing.) */
#define DPX_MAX_CHAINS 8
typedef struct dpx_chain {
  void* spec_a;              /* holds the chain's seeded spectrum on entry (seed == 0) */
  void* spec_b;
  const void* spec_add;      /* the chain's data spectrum (nullable) */
  const dpx_term* terms;     /* nterms terms over the chain's images */
  const float* rho_tab;      /* [total_iters][B] */
  const float* const* lam_tabs; /* nterms pointers to [total_iters][B] */
  float* x_out;              /* the chain's images of x */
  int32_t B;                 /* images of this chain */
  int32_t seed;              /* 0: spec_a is seeded; 1: seed it first from the terms' v / u (dpx_admm_seed_rows); 2: from seed_x0 (dpx_admm_seed_rows_fresh) */
  dpx_stream_t stream;
  const float* seed_x0;      /* seed == 2: the chain's images of the iterate the state was initialised from */
} dpx_chain;
int dpx_admm_run_chains(const dpx_chain* chains, int nchains, const void* dd, int nterms, float eps, int it0, int n_iters,
                        int total_iters, int emit_last, int C, int H, int W, const void* table);
/* fork / join of the chains' streams (events cached per host thread and device): every to[i] waits for what has been issued on `from` /
 * `into` waits for what has been issued on every from[i]; entries equal to the other side are skipped */
/* 1 if kernels on the two streams overlap, 0 if the streams share a hardware queue (HIP multiplexes streams onto GPU_MAX_HW_QUEUES
 * queues: streams on one queue run strictly one after the other), < 0 on error.  A ~1 ms probe that drains both streams first. */
int dpx_streams_concurrent(dpx_stream_t a, dpx_stream_t b);
int dpx_stream_fork(dpx_stream_t from, const dpx_stream_t* to, int n);
int dpx_stream_join(dpx_stream_t into, const dpx_stream_t* from, int n);

/* ------------------------------------------------------------------------------------------ */
/* FFDNet denoiser (deep_prior z-update)                                                       */
/* ------------------------------------------------------------------------------------------ */
/* FFDNet.forward -- proxfn/pnp/denoisers/models/network_ffdnet.py:54-68 (+ basicblock.py:61-126):
 * replicate-pad to even, pixel-unshuffle(2), append sigma map, nb x conv3x3(+ReLU), pixel-shuffle,
 * crop.  `weights` is the packed blob made by dpx_ffdnet_pack (exact fp32, MFMA f32 path).       */
size_t dpx_ffdnet_packed_bytes(int in_nc, int nc, int nb);
int dpx_ffdnet_pack(void* packed, const float* const* w, const float* const* b, int in_nc, int nc, int nb,
                    dpx_stream_t stream);
size_t dpx_ffdnet_ws_bytes(int B, int in_nc, int nc, int H, int W);
int dpx_ffdnet_forward(const float* x, float* y, const float* sigma, const void* packed,
                       int in_nc, int nc, int nb, int B, int H, int W, void* ws, dpx_stream_t stream);

/* The same network at fp32 accuracy on the bf16 matrix cores (csrc/dpx_conv_bf16.hip): weights and activations are split into
 * three bf16 terms each (exact: 3 x 8 mantissa bits) and six products of order <= 2 are accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16 -- 2.67x the rate of the f32-input matrix instruction at a dropped-term error of 2^-24, the size of one
 * fp32 rounding.  mode = 6: that split (default inference path); mode = 1: plain bf16 operands, fp32 accumulation (bf16 training
 * mode); mode = 3: split-f16 -- x = hi + lo / 2^11 with two binary16 terms, three products on v_mfma_f32_32x32x16_f16 (half the
 * matrix work of mode 6, 1e-7 from float64 on the FFDNet stack), valid while every operand stays inside the binary16 range:
 * dpx_ffdnet_f16_overflow(reset) returns 1 if a mode-3 layer has met |x| > 6e4 since the last reset (synchronises the device).
 * mode = 4: mode 3 with every layer behind the first one that has at most 64 output channels as Winograd F(2x2, 3x3) on the same split-f16
 * arithmetic (k_conv3x3_wino: 16 instead of 36 products per 2 x 2 outputs; transformed weights computed in float64 at pack time, the
 * transforms are fp32 additions; the range trap watches the transformed inputs, |x| < 1.5e4 is always safe).  As accurate as mode 3 on
 * the FFDNet stacks (2e-7 from the f32-input path) and measured SLOWER than it on gfx950: opt-in.  The packed blob
 * of a mode is only valid for that mode; dpx_ffdnet_bf16_packed_bytes covers every mode.
 * Activations travel between the layers as fp32 in the channel-group layout [B][C/8][H][W][8].  nc: multiple of 16.           */
int dpx_ffdnet_f16_overflow(int reset);
size_t dpx_ffdnet_bf16_packed_bytes(int in_nc, int nc, int nb);
int dpx_ffdnet_bf16_pack(void* packed, const float* const* w, const float* const* b, int in_nc, int nc, int nb, int mode,
                         dpx_stream_t stream);
size_t dpx_ffdnet_bf16_ws_bytes(int B, int in_nc, int nc, int H, int W);
int dpx_ffdnet_forward_bf16(const float* x, float* y, const float* sigma, const void* packed, int in_nc, int nc, int nb, int mode,
                            int B, int H, int W, void* ws, dpx_stream_t stream);
/* Reverse mode of the same stack on the split kernels, frozen weights (gradients w.r.t. the image and sigma; the reference lets
 * autograd differentiate network_ffdnet.py:54-68): a forward pass that keeps every layer's output in `acts`
 * (dpx_ffdnet_bf16_acts_bytes; mode 6 or 3), the backward-data layers' weights (dpx_ffdnet_bf16_pack_transposed: flipped /
 * transposed, split planes of `mode`, no bias; dpx_ffdnet_bf16_packed_transposed_bytes) and the backward pass -- the same kernel on
 * those weights with the ReLU masks [a_l > 0] in its epilogue.  Its `mode` (the one packed_T was packed for): 6 = split-bf16, any
 * range; 3 = split-f16 (half the matrix work) on gradients multiplied by a power of two that brings max |gy| into [8, 16) -- gx,
 * d/dsigma (and dW, db below) leave multiplied by its inverse, exact; an operand that leaves the binary16 range on the way sets
 * dpx_ffdnet_f16_overflow (the results are then invalid: run mode 6).  gx / gsigma: either may be NULL.                         */
size_t dpx_ffdnet_bf16_acts_bytes(int B, int in_nc, int nc, int nb, int H, int W);
int dpx_ffdnet_forward_bf16_save(const float* x, float* y, const float* sigma, const void* packed, int in_nc, int nc, int nb, int mode,
                                 int B, int H, int W, void* acts, dpx_stream_t stream);
size_t dpx_ffdnet_bf16_packed_transposed_bytes(int in_nc, int nc, int nb);
int dpx_ffdnet_bf16_pack_transposed(void* packed_T, const float* const* w, int in_nc, int nc, int nb, int mode, dpx_stream_t stream);
size_t dpx_ffdnet_bf16_bwd_ws_bytes(int B, int in_nc, int nc, int H, int W);
int dpx_ffdnet_backward_bf16(const float* gy, float* gx, float* gsigma, const void* packed_T, const void* acts, int in_nc, int nc, int nb,
                             int mode, int B, int H, int W, void* ws, dpx_stream_t stream);
/* ... and with the weight / bias gradients (deep_prior(..., trainable=True), reference proxfn/pnp/prior.py:52-60: the denoiser's
 * parameters join the optimiser): forward and backward-data as above on the split kernels, the weight-gradient GEMM of every layer from
 * the same C8 planes in the pass's arithmetic (dpx_conv3x3_wgrad_c8 below).  gw[l]: [cout_l][cin_l][9] (cin_0 = 4 in_nc + 1),
 * gb[l]: [cout_l]; gw[l] == NULL skips layer l; gx / gsigma may be NULL.                                                      */
size_t dpx_ffdnet_bf16_bwd_w_ws_bytes(int B, int in_nc, int nc, int H, int W);
/* The weight-gradient kernel of that pass on its own: dW[co][ci][dy][dx] = sum_{b,y,x} g[b][co][y][x] a[b][ci][y+dy-1][x+dx-1] (zero outside
 * the image), db[co] = sum g -- what autograd forms for a 3x3 convolution of network_ffdnet.py:54-68 -- from the C8 planes
 * [B][groups][H][W][8] of the layer's output gradient g and input a, K = pixels on the 16-bit matrix instruction at fp32 accuracy
 * (mode 6: split-bf16, any range; 3: split-f16, |g|, |a| < 6e4 -- dpx_ffdnet_f16_overflow -- and g scaled towards 1 .. 16 by the caller).
 * cout <= 8 g_groups, cin <= 8 a_groups, both <= 96, in blocks of 32: equal counts, or one side a single block.  gw [cout][cin][9],
 * gb [cout]; mul (device, nullable): both leave multiplied by *mul.  Partial sums per workgroup + fixed-order reduction: bit-reproducible. */
size_t dpx_conv3x3_wgrad_c8_ws_bytes(int cout, int cin);
int dpx_conv3x3_wgrad_c8(const float* g, const float* a, float* gw, float* gb, int cout, int cin, int g_groups, int a_groups, int mode,
                         const float* mul, int B, int H, int W, void* ws, dpx_stream_t stream);
int dpx_ffdnet_backward_bf16_w(const float* gy, float* gx, float* gsigma, float* const* gw, float* const* gb, const void* packed_T,
                               const void* acts, int in_nc, int nc, int nb, int mode, int B, int H, int W, void* ws, dpx_stream_t stream);

/* One plug-and-play ADMM iteration in one call (algo/admm.py:49-59 with deep_prior(FFDNet) as term `ext`): rhs stage, Fourier
 * solve with the fp64 data spectrum, z / dual stage of the closed-form terms, denoiser (mode 0: f32-input MFMA, packed by
 * dpx_ffdnet_pack; 6 / 1: dpx_ffdnet_bf16_pack) on d = x + u, u = d - v.  sigma: [B] (colour network) or [B C] (gray network per
 * band).  v_new receives the denoised image.  ffd_ws: dpx_ffdnet_ws_bytes / dpx_ffdnet_bf16_ws_bytes of the denoiser's batch.  */
int dpx_admm_pnp_iter(float* x, float* rhs, const dpx_term* terms, int nterms, int ext, float* v_new, const float* rho,
                      const float* sigma, const void* spec_add, const void* dd, float eps, const void* packed, int in_nc, int nc, int nb,
                      int mode, int B, int C, int H, int W, const void* table, void* spectrum_ws, void* ffd_ws, dpx_stream_t stream);

/* ... and with the masked-Fourier CG x-update in place of the Fourier solve (config 4: algo/admm.py:49-59 / 78-100 over
 * least_squares.solve_cg, proxfn/sum_square.py:158-197): rhs = ktb + rho sum_i (v_i - u_i) (ktb: K^T b of the data term, nullable),
 * x = dpx_cg_masked_fft(rhs) (mask / mask_images / n_identity / rtol / max_iters / cg_ws as there), z / dual stage of the closed-form
 * terms, denoiser on d = x + u, u = d - v.  Single-channel images [B, 1, H, W], a gray network (in_nc = 1), sigma [B].  Returns the CG
 * exit iteration (>= 0) or a negative status.
 * Folded head / tail (dpx_admm_cg_pnp_iter_folds(mode, B) = 1: split kernels, fused CG branch, every term on x itself): the z / dual stage
 * and the first layer's input are one pass (even H, W), issued right behind the stop test of the CG iteration the previous solve ended at
 * and predicated on that test, so that the host's look at the flag overlaps with it; the pass behind the last layer also forms u and -- rho_next non-null -- the NEXT iteration's right-hand side, written straight into the CG's start state with
 * x_next (the next call's x, another buffer than this call's) zeroed; that next call passes rhs_ready = 1 and skips its rhs stage.  With
 * rho_next = NULL and rhs_ready = 0 every call stands alone.  cg_hint (>= 0; -1: none): the CG iteration this call's solve is expected to end at
 * -- where the head pass is issued early; without it the exit iteration of the thread's previous solve serves (results never depend on it).                                                                        */
int dpx_admm_cg_pnp_iter(float* x, float* rhs, const float* ktb, const dpx_term* terms, int nterms, int ext, float* v_new, const float* rho,
                         const float* sigma, const float* mask, int mask_images, float n_identity, float rtol, int max_iters,
                         const void* packed, int in_nc, int nc, int nb, int mode, int B, int H, int W, const void* table, void* cg_ws,
                         void* ffd_ws, const float* rho_next, float* x_next, int rhs_ready, int cg_hint, dpx_stream_t stream);
int dpx_admm_cg_pnp_iter_folds(int mode, int B);

/* Generic convolution layers on the same MFMA kernel (residual U-Net denoisers behind deep_prior: DRUNet,
 * dprox/proxfn/pnp/denoisers/models/network_unet.py:67-117, basicblock.py).  NCHW fp32, stride 1.
 *   dpx_conv_pack  : w [cout][cin][taps] (taps = 9: 3x3 pad 1, taps = 1: 1x1), b nullable -> packed blob (dpx_conv_packed_bytes)
 *   dpx_conv2d     : out = conv(in) [+ b] ; relu != 0: ReLU ; res != NULL: out += res (ResBlock skip); cin must be even;
 *                    dilation 1..4 with padding = dilation (3x3 only: IRCNN, models/network_dncnn.py:94-109)
 *   dpx_space_to_depth / dpx_depth_to_space : [B,C,H,W] <-> [B,4C,H/2,W/2], channel c*4 + dy*2 + dx (PixelUnshuffle /
 *     PixelShuffle order): a 2x2 stride-2 convolution is space_to_depth + 1x1 conv, a 2x2 stride-2 transposed
 *     convolution is 1x1 conv + depth_to_space.                                                                       */
size_t dpx_conv_packed_bytes(int cin, int cout, int taps);
int dpx_conv_pack(void* packed, const float* w, const float* b, int cin, int cout, int taps, dpx_stream_t stream);
int dpx_conv2d(const float* in, float* out, const void* packed, const float* res, int relu, int cin, int cout, int taps, int dilation,
               int B, int H, int W, dpx_stream_t stream);
/* weight / bias gradients of one dpx_conv2d layer (autograd of nn.Conv2d / ConvTranspose2d in the reference's trainable
 * denoisers, basicblock.py:61-98 under `specialize(..., 'unroll')` with deep_prior(trainable=True)):
 *   gw[co][ci][tap] = sum_{b,y,x} g[b][co][y][x] * a[b][ci][y + (tap/3 - 1) d][x + (tap%3 - 1) d],  gb[co] = sum g[b][co]
 * g = gradient w.r.t. the layer's pre-activation output, a = the layer's input; either of gw / gb may be NULL.
 * Deterministic (fixed-order two-stage reduction, no atomics).                                                         */
size_t dpx_conv2d_wgrad_ws_bytes(int cin, int cout, int taps, int B, int H, int W);
int dpx_conv2d_wgrad(const float* g, const float* a, float* gw, float* gb, int cin, int cout, int taps, int dilation,
                     int B, int H, int W, void* ws, dpx_stream_t stream);
int dpx_space_to_depth(const float* x, float* y, int B, int C, int H, int W, dpx_stream_t stream);
int dpx_depth_to_space(const float* x, float* y, int B, int C, int H, int W, dpx_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* multi-GPU: RCCL behind the C ABI (one process per GPU)                          */
/* ------------------------------------------------------------------------------------------ */
/* The reference runs on one device (algo/base.py:118); a batch is sharded image-wise here.  Only three collectives exist, none of
 * them inside an iteration: broadcast of shared constants, scatter of a batch held by one rank, all-gather of the results.
 * librccl is loaded lazily; a communicator belongs to the HIP device current at dpx_comm_init.  Byte counts, in-order on `stream`. */
/* dpx_comm_use_library: the shared object with the NCCL / RCCL C API to bind instead of the system's librccl (before the first
 * dpx_comm_unique_id / dpx_comm_init of the process; the environment variable DPX_RCCL_LIB supplies the initial value).        */
int dpx_comm_use_library(const char* path);
int dpx_comm_unique_id(void* out128);                                   /* rank 0, then shipped to the others out of band */
int dpx_comm_init(void** comm, const void* id128, int rank, int world);
int dpx_comm_destroy(void* comm);
int dpx_comm_rank(void* comm);
int dpx_comm_world(void* comm);
int dpx_comm_broadcast(void* comm, void* buf, size_t bytes, int root, dpx_stream_t stream);
int dpx_comm_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, dpx_stream_t stream);
int dpx_comm_scatter(void* comm, const void* send, void* recv, size_t bytes_per_rank, int root, dpx_stream_t stream);

/* U-Net denoiser (UNetDenoiser, dprox/proxfn/pnp/denoisers/wrapper.py:206-221 -> models/unet/unet.py:34-135): every ConvLayer
 * (Conv2d 3x3 pad 1 + bias + LeakyReLU(0.2), unet.py:8-31) is one dpx_conv2d_leaky launch on the matrix-core kernel; between them
 *   dpx_maxpool2            nn.MaxPool2d(2), unet.py:80 (floor; ties -> first maximum, as ATen)         [B,C,H,W] -> [B,C,H/2,W/2]
 *   dpx_copy_channels       to_slice != 0: dst[:, c0:c0+C] = src ; else dst = src[:, c0:c0+C]            (torch.cat / its gradient)
 *   dpx_upsample2_into      nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) + F.pad to (Hd, Wd) (left/top =
 *                           diff // 2, unet.py:103-109) written into channels [c0, c0+C) of the [B,Ctot,Hd,Wd] concat buffer (unet.py:115)
 * and the adjoints used by the backward-data pass (*_bwd; dpx_leaky_relu_bwd: g * (y > 0 ? 1 : slope) from the saved output). */
int dpx_conv2d_leaky(const float* in, float* out, const void* packed, float neg_slope, int cin, int cout, int taps, int B, int H, int W,
                     dpx_stream_t stream);
int dpx_maxpool2(const float* x, float* y, int B, int C, int H, int W, dpx_stream_t stream);
int dpx_maxpool2_bwd(const float* x, const float* gy, float* gx, int B, int C, int H, int W, dpx_stream_t stream);
int dpx_copy_channels(const float* src, float* dst, int to_slice, int B, int C, int H, int W, int Ctot, int c0, dpx_stream_t stream);
int dpx_upsample2_into(const float* src, float* dst, int B, int C, int h, int w, int Ctot, int c0, int Hd, int Wd, dpx_stream_t stream);
int dpx_upsample2_into_bwd(const float* gdst, float* gsrc, int B, int C, int h, int w, int Ctot, int c0, int Hd, int Wd, dpx_stream_t stream);
int dpx_leaky_relu_bwd(const float* y, const float* g, float* gin, long n, float neg_slope, dpx_stream_t stream);

/* Training variants.  dpx_ffdnet_forward_save = dpx_ffdnet_forward that keeps every layer's output in `acts`
 * (dpx_ffdnet_acts_bytes) for the backward pass.  dpx_ffdnet_backward: gradient of the network output w.r.t. its image
 * input (gx, nullable) and w.r.t. the per-image noise level (gsigma[B], nullable) given gy -- the chain of transposed
 * 3x3 convolutions (same MFMA kernel, weights flipped / transposed once by dpx_ffdnet_pack_transposed, the ReLU masks fused into
 * the epilogues), i.e. what PyTorch autograd does for network_ffdnet.py:54-68 in the reference's unrolled / DEQ training.
 * gw / gb (nullable, together): per-layer device pointers [nb] receiving d/dW [cout][cin][3][3] and d/db [cout]
 * (entries may be NULL to skip a layer) -- a pixels-as-K GEMM on the fp32 matrix cores with a deterministic two-stage
 * reduction.                                                                                                           */
size_t dpx_ffdnet_acts_bytes(int B, int in_nc, int nc, int nb, int H, int W);
int dpx_ffdnet_forward_save(const float* x, float* y, const float* sigma, const void* packed, int in_nc, int nc, int nb,
                            int B, int H, int W, void* acts, dpx_stream_t stream);
size_t dpx_ffdnet_packed_transposed_bytes(int in_nc, int nc, int nb);
int dpx_ffdnet_pack_transposed(void* packed_T, const float* const* w, int in_nc, int nc, int nb, dpx_stream_t stream);
size_t dpx_ffdnet_bwd_ws_bytes(int B, int in_nc, int nc, int H, int W);
int dpx_ffdnet_backward(const float* gy, float* gx, float* gsigma, float* const* gw, float* const* gb, const void* packed_T,
                        const void* acts, int in_nc, int nc, int nb, int B, int H, int W, void* ws, dpx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPX_H */
