/*
 * pbbss.h -- C ABI of libpbbss_hip.so: the MI355X (gfx950) engine for the
 * pb_bss cACGMM EM loop and mask-based beamformer extraction.
 *
 * Drop-in boundary.  The reference (fgnt/pb_bss) has no
 * FFI registry: its only native seam is the optional Cython module imported at
 * pb_bss/extraction/beamformer.py:38-56.  Every entry point below replaces the
 * NumPy/LAPACK body of one reference function; the citation names it
 * (paths relative to /root/reference/pb_bss/).
 *
 * Conventions
 *  - All data pointers are DEVICE pointers owned by the caller (PyTorch-ROCm
 *    tensors in the Python host layer).  The library allocates only private
 *    scratch inside the handle and never frees caller memory.
 *  - Row-major, last index fastest.  "c64" = interleaved (re,im) float32,
 *    "c128" = interleaved float64.  B is the flattened "independent" axis
 *    (frequency bins x batch), D sensors, T frames, K classes.
 *  - Arithmetic is IEEE float64 on the device regardless of the storage type
 *    (float32 cannot hold 1e-5 over an EM trajectory).
 *  - Calls are asynchronous on `stream` (a hipStream_t passed as void*).
 *  - Return value: PBBSS_OK or a negative error code; numerical trouble is
 *    reported per problem in caller-provided `status` arrays, never thrown.
 *  - Thread safety: one handle per (device, host thread); no global state.
 */
#ifndef PBBSS_H_
#define PBBSS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBBSS_VERSION 610 /* 0.6.1: pbbss_select_reference_channel, pbbss_apply_beamforming_vector_shared;
                             0.6.0: pbbss_log_pdf_to_affiliation_inline_pa, D = 33 / 34 in pbbss_cacgmm_fit / _predict;
                             0.4.1: pbbss_set_dhtv_probe; 0.4.0: pbbss_split_reset, pbbss_set_spin_limit, pbbss_reference_channel_terms,
                             pbbss_rank_one_approximation, pbbss_matvec (0.3.0: em_opts.precision,
                             mix_opts.sharded, pbbss_comm_info) */

/* ---- error codes --------------------------------------------------------- */
#define PBBSS_OK 0
#define PBBSS_ERR_INVALID_ARG (-1)   /* NULL pointer / bad enum / size <= 0   */
#define PBBSS_ERR_UNSUPPORTED (-2)   /* shape outside the compiled kernels    */
#define PBBSS_ERR_HIP (-3)           /* a HIP runtime call failed             */
#define PBBSS_ERR_LDS_CAPACITY (-4)  /* T too large for the LDS-resident path */
#define PBBSS_ERR_INTERNAL (-5)      /* workspace accounting mismatch (a bug) */

/* ---- per-problem status bits (int32 status arrays) ------------------------ */
#define PBBSS_ST_NONFINITE 1u      /* non-finite covariance / eigenvalues
                                      (reference: assert np.isfinite, cacg.py:127,326,333) */
#define PBBSS_ST_EIG_NOCONV 2u     /* Jacobi did not converge (reference: LinAlgError) */
#define PBBSS_ST_FLOORED 4u        /* >=1 eigenvalue hit the floor (informational) */
#define PBBSS_ST_SLOWPATH 8u       /* in-loop eigen path was taken (informational) */
#define PBBSS_ST_NOT_POSDEF 16u    /* Cholesky of B failed (GEV: LAPACK INFO > N)  */
#define PBBSS_ST_SINGULAR 32u      /* LU met a zero pivot (np.linalg.LinAlgError)  */

typedef struct pbbss_handle_s* pbbss_handle_t;

int pbbss_version(void);
const char* pbbss_error_string(int code);

/* Bind a handle to HIP device `device_id`; queries CU count / LDS size, allocates the small
 * device-side control buffers and a highest-priority side stream (split-bin groups).
 * Calling convention of every entry point below: `stream` is a hipStream_t of THAT device
 * (NULL = the default stream); the library selects the handle's device for the duration of
 * each call and restores the caller's selection; all
 * pointers are device pointers unless stated otherwise; calls only enqueue work (no host
 * synchronisation) unless stated otherwise; one handle must not be used from two host
 * threads at the same time (create one per thread, as pb_bss_amd/_lib.py does).  PBBSS_DEBUG
 * in the environment makes pbbss_create report failing HIP calls on stderr. */
int pbbss_create(pbbss_handle_t* out, int device_id);
/* Shape range.  2 <= D <= 8 sensors: the fused kernels (one wavefront per D x D matrix; the
 * whole cACGMM EM loop in one persistent launch).  9 <= D <= 32: the same entry points
 * (pbbss_cacgmm_fit / _predict, pbbss_cacg_m_step, pbbss_heev_batched, pbbss_psd, pbbss_gev,
 * pbbss_solve, pbbss_mvdr_souden, pbbss_wmwf, pbbss_mvdr, pbbss_ban) run a generic-size path
 * (one workgroup per matrix, matrices in LDS; the EM loop enqueues three kernels per iteration;
 * layout TD only for the fit).  1 <= K <= 6 classes on the fused kernels, 7 <= K <= 19 on the
 * generic-size path at any D.  Watson mixture (pbbss_cwmm_fit): fused kernel for D <= 8, K <= 4,
 * generic-size path up to D = 32, K = 19.  Joint models (pbbss_joint_fit): D <= 32 (the spatial
 * half of 9 <= D <= 32, or of 7..8 classes, on the generic-size kernels; inline permutation
 * alignment for K <= 6), K <= 8 (bound of the spectral kernels).  LCMV: D <= 8. */
int pbbss_destroy(pbbss_handle_t h);

/* ------------------------------------------------------------------------- */
/* a1  normalize_observation                                                  */
/*     distribution/complex_angular_central_gaussian.py:34-55                 */
/*     (+ _unit_norm eps_style='where', distribution/utils.py:223-256)        */
/* y (B,T,D) -> out (B,D,T), unit L2 norm over D; all-zero frames stay zero.  */
/* is_c128: 0 = complex64 in/out, 1 = complex128 in/out.                      */
/* ------------------------------------------------------------------------- */
int pbbss_normalize_observation(pbbss_handle_t h, const void* y, int is_c128,
                                int64_t B, int T, int D, void* out,
                                void* stream);

/* ------------------------------------------------------------------------- */
/* EM options: the keyword arguments of CACGMMTrainer.fit                     */
/*     distribution/cacgmm.py:142-157                                         */
/* ------------------------------------------------------------------------- */
#define PBBSS_COVNORM_NONE 0       /* covariance_norm=False      */
#define PBBSS_COVNORM_EIGENVALUE 1 /* covariance_norm='eigenvalue' */
#define PBBSS_COVNORM_TRACE 2      /* covariance_norm='trace'    */

#define PBBSS_WEIGHT_PER_CLASS_MEAN 0 /* weight_constant_axis=(-1,): mean over frames   */
#define PBBSS_WEIGHT_UNIFORM 1        /* weight_constant_axis=-2: constant 1/K          */
/* pbbss_cacgmm_fit_shared only: weights shared by a group of problems (frequency bins)   */
#define PBBSS_WEIGHT_SHARED_K 2       /* weight_constant_axis=(-3, -1): (group, K)      */
#define PBBSS_WEIGHT_SHARED_KT 3      /* weight_constant_axis=(-3,): (group, K, T)      */

#define PBBSS_LAYOUT_TD 0 /* observation (B,T,D): raw, the kernel unit-normalises */
#define PBBSS_LAYOUT_DT 1 /* observation (B,D,T): already normalised (as _predict/_fit get it) */

typedef struct pbbss_em_opts {
  int32_t iterations;       /* >= 0 EM iterations (M-steps)                     */
  int32_t covariance_norm;  /* PBBSS_COVNORM_*                                   */
  int32_t weight_mode;      /* PBBSS_WEIGHT_*                                    */
  int32_t hermitize;        /* accepted for API parity; the accumulation is     */
                            /* Hermitian by construction (cacg.py:335-336)      */
  int32_t layout;           /* PBBSS_LAYOUT_*                                    */
  int32_t y_is_c128;        /* 0 complex64 observation, 1 complex128             */
  int32_t final_predict;    /* 1: after the last M-step run one E-step with      */
                            /* affiliation_eps = 0 (== model.predict(y))         */
  int32_t force_eig;        /* 1: eigendecompose every iteration (no Cholesky    */
                            /* fast path); for tests                             */
  double affiliation_eps;   /* clip of the posteriors inside the loop (1e-10)    */
  double eigenvalue_floor;  /* relative eigenvalue floor (1e-10)                 */
  int32_t precision;        /* PBBSS_PRECISION_* of the E / M phases (pbbss_cacgmm_fit) */
  int32_t reserved;
} pbbss_em_opts;

/* Arithmetic of the E and M phases of pbbss_cacgmm_fit.                                */
/* F64 (default): float64 throughout, for complex64 and complex128 observations.        */
/* F32: "reference precision" -- what the reference itself computes in for a complex64   */
/* observation with an ndarray initialisation (cacgmm.py:226-227: the initialisation is  */
/* cast to y.real.dtype, every einsum then runs in complex64 / float32): quadratic       */
/* forms, posteriors and covariance sums in packed float32 (csrc/cacgmm_em32.hpp), class */
/* sums, factorisation, eigenvalue floor and the returned model in float64.  complex64   */
/* input, 2 <= D <= 8, K <= 6, frames resident in LDS, no quadratic-form output:         */
/* otherwise PBBSS_ERR_UNSUPPORTED (the caller uses F64, which is a superset in          */
/* accuracy).  K = 5, 6 and source_activity_mask are served since 0.4.1.                 */
#define PBBSS_PRECISION_F64 0
#define PBBSS_PRECISION_F32 1

/* ------------------------------------------------------------------------- */
/* a8  CACGMMTrainer.fit / fit_predict   distribution/cacgmm.py:142-313        */
/*     (loop body: a2 _log_pdf cacg.py:167-203, a3 log_pdf_to_affiliation      */
/*      mixture_model_utils.py:7-55, a5 estimate_mixture_weight :133-203,     */
/*      a6 _fit cacg.py:253-342, a7 from_covariance cacg.py:82-132)            */
/*                                                                             */
/* Initialisation: exactly one of                                              */
/*   gamma0 (B,K,T) f64 affiliations  (cacgmm.py:211-228), or                  */
/*   the model triple in_eigvec c128 (B,K,D,D), in_eigval f64 (B,K,D),         */
/*   in_weight f64 (B,K)  (cacgmm.py:229-234).                                 */
/* Optional: saliency f64 (B,T) or NULL; activity uint8 (B,K,T) or NULL        */
/* (source_activity_mask).                                                     */
/* Outputs (caller-allocated): eigvec c128 (B,K,D,D) columns = eigenvectors,   */
/* eigenvalues ascending as numpy.linalg.eigh; eigval f64 (B,K,D); weight f64  */
/* (B,K); status int32 (B,K) PBBSS_ST_* bits; optional affiliation f64 (B,K,T) */
/* and quadratic_form f64 (B,K,T) written by the final predict (may be NULL).  */
/* The whole EM loop runs on the device in one launch; nothing returns to the  */
/* host between iterations.                                                    */
/* ------------------------------------------------------------------------- */
int pbbss_cacgmm_fit(pbbss_handle_t h, const void* y, int64_t B, int T, int D,
                     int K, const double* gamma0, const void* in_eigvec,
                     const double* in_eigval, const double* in_weight,
                     const double* saliency, const uint8_t* activity,
                     const pbbss_em_opts* opts, void* out_eigvec,
                     double* out_eigval, double* out_weight,
                     int32_t* out_status, double* out_affiliation,
                     double* out_quadratic_form, void* stream);

/* ------------------------------------------------------------------------- */
/* a8' CACGMMTrainer.fit with weight_constant_axis=(-3,) or (-3, -1)             */
/*     distribution/cacgmm.py:59, :142-157; mixture_model_utils.py:184-201       */
/* The mixture weights are estimated over `group` consecutive problems (the      */
/* frequency bins of one utterance): B = n_groups * group.  The whole loop is    */
/* still ONE cooperative launch per batch of co-resident groups; the groups'     */
/* workgroups exchange their masked affiliations (SHARED_KT) or class sums       */
/* (SHARED_K) through device memory once per iteration.                          */
/* opts->weight_mode: PBBSS_WEIGHT_SHARED_K -> in/out weight (B/group, K);       */
/*                    PBBSS_WEIGHT_SHARED_KT -> in/out weight (B/group, K, T).   */
/* Other arguments as pbbss_cacgmm_fit.  2 <= D <= 8, K <= 4, frames resident in */
/* LDS, and `group` workgroups must fit the device at once (768 on MI355X at     */
/* T = 500): otherwise PBBSS_ERR_UNSUPPORTED, and the caller runs the loop step  */
/* by step (pbbss_cacgmm_predict, pbbss_estimate_mixture_weight, pbbss_cacg_fit).*/
/* A hand-off that times out poisons out_status (EIG_NOCONV | NONFINITE) and     */
/* sets the flag pbbss_split_error reports.                                      */
/* ------------------------------------------------------------------------- */
int pbbss_cacgmm_fit_shared(pbbss_handle_t h, const void* y, int64_t B, int T, int D, int K,
                            int64_t group, const double* gamma0, const void* in_eigvec,
                            const double* in_eigval, const double* in_weight,
                            const double* saliency, const uint8_t* activity,
                            const pbbss_em_opts* opts, void* out_eigvec, double* out_eigval,
                            double* out_weight, int32_t* out_status, double* out_affiliation,
                            double* out_quadratic_form, void* stream);

/* ------------------------------------------------------------------------- */
/* a4  CACGMM.predict / _predict   distribution/cacgmm.py:64-95                */
/* One E-step from a model.  weight is addressed with element strides          */
/* (wb, wk, wt) so any reference broadcast shape works: (B,K,1) -> (K,1,0),    */
/* (K,1) -> (0,1,0), (1,K,T) -> (0,T,1).  Outputs f64 (B,K,T); quadratic_form  */
/* and log_pdf may be NULL.                                                    */
/* ------------------------------------------------------------------------- */
int pbbss_cacgmm_predict(pbbss_handle_t h, const void* y, int64_t B, int T,
                         int D, int K, const void* eigvec, const double* eigval,
                         const double* weight, int64_t wb, int64_t wk,
                         int64_t wt, const uint8_t* activity, int layout,
                         int y_is_c128, double affiliation_eps,
                         double* out_affiliation, double* out_quadratic_form,
                         double* out_log_pdf, void* stream);

/* ------------------------------------------------------------------------- */
/* a6+a7  ComplexAngularCentralGaussianTrainer._fit  cacg.py:253-342           */
/* One stand-alone M-step on NORMALISED observations y (B,D,T) (layout DT) or   */
/* raw (B,T,D): covariance = D * sum_t (saliency/q) y y^H / max(sum_t          */
/* saliency, tiny), Hermitian eigendecomposition, eigenvalue normalisation and */
/* floor.  saliency (B,K,T) f64 (the "masked affiliation"), quadratic_form     */
/* (B,K,T) f64.  Also returns the covariance itself if out_cov != NULL.        */
/* ------------------------------------------------------------------------- */
int pbbss_cacg_m_step(pbbss_handle_t h, const void* y, int64_t B, int T, int D,
                      int K, const double* saliency,
                      const double* quadratic_form, int layout, int y_is_c128,
                      int covariance_norm, double eigenvalue_floor,
                      void* out_eigvec, double* out_eigval, void* out_cov,
                      int32_t* out_status, void* stream);

/* ------------------------------------------------------------------------- */
/* Batched Hermitian eigendecomposition (numpy.linalg.eigh as used at          */
/* cacg.py:95 and extraction/beamformer.py:180).  a c128 (N,D,D) -> eigenvalues */
/* ascending f64 (N,D), eigenvectors c128 (N,D,D) in columns.  D <= 32.        */
/* ------------------------------------------------------------------------- */
int pbbss_heev_batched(pbbss_handle_t h, const void* a, int64_t N, int D,
                       double* out_eigval, void* out_eigvec,
                       int32_t* out_status, void* stream);

/* ------------------------------------------------------------------------- */
/* a10  get_power_spectral_density_matrix   extraction/beamformer.py:59-160     */
/* x (B,D,T) c64/c128; mask f64 (B,K,T) (mask_b_stride = K*T) or shared over   */
/* sources; normalize: divide the mask by max(sum_t mask, 1e-10) first.        */
/* mask == NULL: plain sample covariance / T (K must be 1).                    */
/* out c128 (B,K,D,D).                                                         */
/* ------------------------------------------------------------------------- */
int pbbss_psd(pbbss_handle_t h, const void* x, int x_is_c128, int64_t B, int T,
              int D, int K, const double* mask, int normalize, void* out,
              void* stream);

/* ------------------------------------------------------------------------- */
/* a11  get_gev_vector -> _c_get_gev_vector                                    */
/*      extraction/beamformer.py:292-364, cythonized/get_gev_vector.pyx:42-150 */
/* Principal generalised eigenvector of (target, noise), c128 (N,D,D) each,    */
/* normalised so that w^H noise w = 1 (LAPACK zhegvd ITYPE=1).  status[n] != 0  */
/* mirrors INFO != 0: PBBSS_ST_NOT_POSDEF | (minor_order << 8), or              */
/* PBBSS_ST_EIG_NOCONV; the host raises ValueError exactly like the .pyx.      */
/* ------------------------------------------------------------------------- */
int pbbss_gev(pbbss_handle_t h, const void* target, const void* noise,
              int64_t N, int D, void* out_w, int32_t* out_status, void* stream);

/* ------------------------------------------------------------------------- */
/* a11 (use_eig=True)  general, non-Hermitian GEV                              */
/*     extraction/beamformer.py:352-358 -> cythonized/c_eig.pyx:14-123 (zggev) */
/*     extraction/beamformer.py:367-411 (scipy.linalg.eig fallback)            */
/* target, noise (N,D,D) c128, ANY complex matrices with invertible noise (no   */
/* Hermitian / definiteness assumption).  out_w (N,D): UNIT-2-NORM right        */
/* eigenvector of the eigenvalue numpy.argmax would pick (largest real part,    */
/* ties by imaginary part), arbitrary phase -- the reference's normalisation on */
/* this path, not zhegvd's w^H Phi_nn w = 1.  out_lambda (N) c128 or NULL: that  */
/* eigenvalue.  status: PBBSS_ST_SINGULAR (noise exactly singular),             */
/* PBBSS_ST_EIG_NOCONV (the .pyx's "The QZ iteration failed"), PBBSS_ST_NONFINITE. */
/* 1 <= D <= 32.                                                                */
/* ------------------------------------------------------------------------- */
int pbbss_gev_general(pbbss_handle_t h, const void* target, const void* noise,
                      int64_t N, int D, void* out_w, void* out_lambda,
                      int32_t* out_status, void* stream);

/* ------------------------------------------------------------------------- */
/* Batched complex solve  A X = Bm  (numpy.linalg.solve inside stable_solve,   */
/* math/solve.py:20-114; extraction/beamformer.py:250,277,682).  A (N,D,D),    */
/* Bm (N,D,M) c128, M <= D.  LU with partial pivoting.  status PBBSS_ST_SINGULAR */
/* where a pivot is exactly zero (host then takes the lstsq fallback).         */
/* ------------------------------------------------------------------------- */
int pbbss_solve(pbbss_handle_t h, const void* A, const void* Bm, int64_t N,
                int D, int M, void* out_x, int32_t* out_status, void* stream);

/* ------------------------------------------------------------------------- */
/* a12  get_mvdr_vector_souden  extraction/beamformer.py:627-698               */
/*      (+ get_optimal_reference_channel :601-624)                             */
/* phase 1 (this call): G = noise^-1 target; mat = G / max(Re tr G, eps);      */
/* per-matrix SNR numerators/denominators for every candidate reference        */
/* channel: snr_num/snr_den c128 (N,D) = diag(mat^H target mat), diag(mat^H     */
/* noise mat).  The host sums them over frequency (all-reduce when F is        */
/* sharded), takes the argmax and selects column r of out_mat (N,D,D).         */
/* ------------------------------------------------------------------------- */
int pbbss_mvdr_souden(pbbss_handle_t h, const void* target, const void* noise,
                      int64_t N, int D, double eps, void* out_mat,
                      void* out_snr_num, void* out_snr_den,
                      int32_t* out_status, void* stream);

/* ------------------------------------------------------------------------- */
/* N4  get_wmwf_vector  extraction/beamformer.py:701-753                       */
/* Speech-distortion-weighted multichannel Wiener filter, same structure as    */
/* pbbss_mvdr_souden: filter = (noise^-1 target) / (mu + trace) or, with       */
/* frequency_dependent != 0, / sqrt(target[0,0] * trace).  out_mat c128        */
/* (N,D,D); the reference channel is a column of it (per-matrix SNR terms in   */
/* out_snr_num / out_snr_den c128 (N,D), may be NULL).                         */
/* ------------------------------------------------------------------------- */
int pbbss_wmwf(pbbss_handle_t h, const void* target, const void* noise, int64_t N,
               int D, double distortion_weight, int frequency_dependent,
               void* out_mat, void* out_snr_num, void* out_snr_den,
               int32_t* out_status, void* stream);

/* ------------------------------------------------------------------------- */
/* a13  get_mvdr_vector  extraction/beamformer.py:230-260                      */
/* w = noise^-1 h / (h^H noise^-1 h), noise hermitised first.                  */
/* atf c128 (N,D), noise c128 (N,D,D) -> out c128 (N,D).                        */
/* ------------------------------------------------------------------------- */
int pbbss_mvdr(pbbss_handle_t h, const void* atf, const void* noise, int64_t N,
               int D, void* out_w, int32_t* out_status, void* stream);

/* ------------------------------------------------------------------------- */
/* a14  blind_analytic_normalization  extraction/beamformer.py:459-488         */
/* w c128 (N,D), noise c128 (N,D,D) -> out c128 (N,D).                          */
/* ------------------------------------------------------------------------- */
int pbbss_ban(pbbss_handle_t h, const void* w, const void* noise, int64_t N,
              int D, void* out_w, void* stream);

/* ------------------------------------------------------------------------- */
/* a15  apply_beamforming_vector  extraction/beamformer.py:572-583             */
/* out[b,t] = sum_d conj(w[b,d]) x[b,d,t];  w c128 (B,D), x (B,D,T) c64/c128,  */
/* out c128 (B,T).                                                             */
/* ------------------------------------------------------------------------- */
int pbbss_apply_beamforming_vector(pbbss_handle_t h, const void* w,
                                   const void* x, int x_is_c128, int64_t B,
                                   int T, int D, void* out, void* stream);
/* The same with ONE observation shared by several vectors per bin (K beamformers on one   */
/* STFT, the broadcast of the reference's einsum '...a,...at->...t' over a leading axis of    */
/* `vector`): x (x_batch,D,T), w (B,D) and out (B,T) with B a multiple of x_batch; problem b  */
/* reads x[b % x_batch].  No copies of x are made.                                            */
int pbbss_apply_beamforming_vector_shared(pbbss_handle_t h, const void* w, const void* x,
                                          int x_is_c128, int64_t B, int64_t x_batch, int T,
                                          int D, void* out, void* stream);

/* ------------------------------------------------------------------------- */
/* a12, phase 2 on the device: get_optimal_reference_channel                   */
/* extraction/beamformer.py:601-624 and the column select :690-698, for L      */
/* problems (classes / utterances) of F bins each, from the outputs of         */
/* pbbss_mvdr_souden: problem l, bin f is matrix n = l*lead_stride +            */
/* f*bin_stride of that call.  snr_r = sum_f num[f,r] / max(sum_f den[f,r],     */
/* eps) (NumPy's complex ordering in the maximum), ref = first argmax of Re     */
/* snr; out_w c128 (L,F,D) = column ref of every bin's matrix, out_ref (L),     */
/* out_ok (L): 1 if every SNR of the problem is finite (the reference asserts   */
/* it, :619 -- the caller raises when it next synchronises).  Not for sharded   */
/* bins (the sums then need an all-reduce between the phases).                  */
/* ------------------------------------------------------------------------- */
int pbbss_select_reference_channel(pbbss_handle_t h, const void* mat, const void* snr_num,
                                   const void* snr_den, int64_t L, int64_t F, int D,
                                   int64_t lead_stride, int64_t bin_stride, double eps,
                                   void* out_w, int32_t* out_ref, int32_t* out_ok, void* stream);

/* similarity metrics of the permutation solvers (_ScoreMatrix, permutation_alignment.py:380-417) */
#define PBBSS_PA_COS 0
#define PBBSS_PA_MULTIPLY 1
#define PBBSS_PA_EUCLIDEAN 2

/* ------------------------------------------------------------------------- */
/* N1  DHTVPermutationAlignment.calculate_mapping  permutation_alignment.py:295-355 */
/*     (similarity_metric = metric: PBBSS_PA_COS (the reference's default),     */
/*     PBBSS_PA_MULTIPLY or PBBSS_PA_EUCLIDEAN; algorithm 'greedy' (optimal=0)  */
/*     or 'optimal'),                                                            */
/*     score matrix :380-407, assignment :469-589.                              */
/* mask f64 (U,K,F,T) (U utterances); plan int32 (P,3) rows (iterations, start, */
/* end) = DHTVPermutationAlignment.alignment_plan (:204-293), a DEVICE array;    */
/* scratch f64 (U,K,F,T) receives the aligned features (unit-norm for 'cos'); out_mapping    */
/* int32 (U,K,F) is the reverse mapping; status int32 (U) gets                    */
/* PBBSS_ST_NONFINITE where the reference raises 'score matrix is infeasible',   */
/* and PBBSS_ST_EIG_NOCONV where a bounded wait between the workgroups that share */
/* an utterance ran out (they were not co-resident: compute units held by other  */
/* work) -- that utterance's mapping is then invalid; rerun with                  */
/* pbbss_set_dhtv_team(h, 1), the one-workgroup kernel (the Python layer does).   */
/* K <= 8.  One launch runs the whole plan.                                       */
/* ------------------------------------------------------------------------- */
int pbbss_dhtv_calculate_mapping(pbbss_handle_t h, const double* mask, int64_t U,
                                 int K, int F, int T, const int32_t* plan, int P,
                                 int optimal, int metric, double* scratch,
                                 int32_t* out_mapping, int32_t* out_status,
                                 void* stream);

/* ------------------------------------------------------------------------- */
/* N1  OraclePermutationAlignment.calculate_mapping  permutation_alignment.py:703-786 */
/*     GreedyPermutationAlignment.calculate_mapping  permutation_alignment.py:592-701 */
/*     score matrices :380-417 (_ScoreMatrix.cos / multiply / euclidean),          */
/*     assignment :469-589 (_mapping_from_score_matrix).                           */
/* One launch scores every (utterance, bin) of `mask` against the same bin of      */
/* `reference` and assigns the classes.  Both arrays are float64 with contiguous   */
/* frames; *_strides = element strides of (utterance, class, bin) (HOST arrays of  */
/* 3), so the greedy solver can pass mask[:, 1:] / mask[:, :-1] without a copy.    */
/* out_scores f64 (U,F,K,K) [reference class][mask class] or NULL; out_mapping     */
/* int32 (U,K,map_F): column map_col0 + f gets the permutation of bin f (reverse    */
/* mapping, as the reference returns it); out_status int32 (U) gets                */
/* PBBSS_ST_NONFINITE where the reference raises 'score matrix is infeasible'.     */
/* K <= 8.                                                                          */
/* ------------------------------------------------------------------------- */
int pbbss_pa_pairwise_mapping(pbbss_handle_t h, const double* mask, const double* reference,
                              int64_t U, int K, int64_t F, int T, const int64_t* mask_strides,
                              const int64_t* reference_strides, int metric, int optimal,
                              double* out_scores, int32_t* out_mapping, int64_t map_F,
                              int64_t map_col0, int32_t* out_status, void* stream);
/* Greedy solver's recursion (:690-699) in place on mapping int32 (U,K,F): column 0 :=  */
/* identity, then mapping[:, f] = mapping[mapping[:, f-1], f] for f = 1 .. F-1           */
/* (a prefix scan over permutation compositions).                                         */
int pbbss_pa_compose_mapping(pbbss_handle_t h, int32_t* mapping, int64_t U, int K, int64_t F,
                             void* stream);
/* _mapping_from_score_matrix (:469-589): scores f64 (N,K,K) -> out_mapping int32 (K,N); */
/* out_status int32 (1) gets PBBSS_ST_NONFINITE for non-finite scores.                    */
int pbbss_pa_mapping_from_scores(pbbss_handle_t h, const double* scores, int64_t N, int K,
                                 int optimal, int32_t* out_mapping, int32_t* out_status,
                                 void* stream);

/* apply_mapping  permutation_alignment.py:54-104:                              */
/* out[u,k,f,:] = mask[u, mapping[u,k,f], f, :];  mask/out f64 (U,K,F,T).        */
int pbbss_apply_mapping(pbbss_handle_t h, const double* mask,
                        const int32_t* mapping, int64_t U, int K, int F, int T,
                        double* out, void* stream);

/* ------------------------------------------------------------------------- */
/* N2  CWMMTrainer.fit / fit_predict, CWMM.predict   distribution/cwmm.py:76-240, */
/*     :25-52; ComplexWatson.log_pdf / log_norm_1f1  complex_watson.py:73-87,     */
/*     :157-168; ComplexWatsonTrainer._fit + spline  :238-271, :300-315.          */
/* y (B,T,D) complex, raw (the kernel unit-normalises, complex_watson.py:16-29). */
/* Initialisation: gamma0 (B,K,T) f64 affiliations, or a model (in_mode c128     */
/* (B,K,D), in_concentration f64 (B,K), in_weight f64 (B,K)).  iterations may be */
/* 0 with a model (pure predict).  The concentration look-up is the quadratic    */
/* B-spline the reference builds with SciPy (interp1d(kind='quadratic')):        */
/* spline_t (n_coef + 3 knots) and spline_c (n_coef coefficients) are DEVICE     */
/* arrays, valid for eigenvalues in [ev_min, ev_max]; below -> 0, above ->        */
/* max_concentration (the reference's fill_value).  saliency f64 (B,T) or NULL.   */
/* Outputs: mode c128 (B,K,D), concentration f64 (B,K), weight f64 (B,K),        */
/* status int32 (B,K); optional affiliation / log_pdf f64 (B,K,T) from the final  */
/* E-step (final_predict).  2 <= D <= 8 and K <= 4: one fused persistent kernel;   */
/* up to D = 32 sensors / K = 19 classes: generic-size kernels, one launch group   */
/* per iteration (csrc/generic_watson.hip).                                       */
/* ------------------------------------------------------------------------- */
typedef struct pbbss_cwmm_opts {
  int32_t iterations;
  int32_t weight_mode;   /* PBBSS_WEIGHT_* */
  int32_t y_is_c128;
  int32_t final_predict;
  int32_t n_coef;        /* number of B-spline coefficients */
  int32_t group;         /* weight_mode PBBSS_WEIGHT_SHARED_K / _KT (weight_constant_axis (-3, -1) /
                          * (-3,), cwmm.py:217-240): consecutive problems (the bins of an utterance)
                          * that share one weight set; out_weight is then (B / group, K) /
                          * (B / group, K, T).  One cooperative
                          * launch (D <= 8, K <= 4, frames LDS-resident, a group co-resident), else
                          * PBBSS_ERR_UNSUPPORTED and the caller runs the loop step by step.  A grid
                          * barrier that times out poisons every status word (pbbss_split_error).
                          * Other weight modes: ignored (was `reserved`). */
  double ev_min, ev_max, max_concentration;
} pbbss_cwmm_opts;

int pbbss_cwmm_fit(pbbss_handle_t h, const void* y, int64_t B, int T, int D, int K,
                   const double* gamma0, const void* in_mode,
                   const double* in_concentration, const double* in_weight,
                   const double* saliency, const pbbss_cwmm_opts* opts,
                   const double* spline_t, const double* spline_c, void* out_mode,
                   double* out_concentration, double* out_weight,
                   int32_t* out_status, double* out_affiliation,
                   double* out_log_pdf, void* stream);

/* ------------------------------------------------------------------------- */
/* N2/N3  Real-embedding mixture components: von Mises-Fisher and spherical      */
/* Gaussian (distribution/von_mises_fisher.py:33-144, gaussian.py:100-193).      */
/* y (B,N,E) real, row-major, float32 or float64 (y_is_f64); 1 <= E <= 256,      */
/* K <= 8.  `scale` (B,K) is the vMF concentration or the spherical covariance.   */
/* ------------------------------------------------------------------------- */
#define PBBSS_EMBED_VMF 0             /* VonMisesFisher                       */
#define PBBSS_EMBED_GAUSS_SPHERICAL 1 /* SphericalGaussian                    */
#define PBBSS_EMBED_GAUSS_FULL 2      /* Gaussian (pbbss_joint_fit; single fits: pbbss_gauss_full_*) */
#define PBBSS_EMBED_GAUSS_DIAG 3      /* DiagonalGaussian: `scale` (K,E) per-dimension variances,
                                       * B = 1; the log-pdf is evaluated AS THE REFERENCE WRITES
                                       * IT (gaussian.py:76-97 hands the (K,E) precision array to
                                       * einsum as one K x E matrix shared by all classes) */

/* VonMisesFisher.log_pdf (von_mises_fisher.py:62-78; rows are unit-normalised   */
/* first) / SphericalGaussian.log_pdf (gaussian.py:116-137): out (B,K,N) f64.    */
int pbbss_embed_log_pdf(pbbss_handle_t h, const void* y, int y_is_f64, int64_t B,
                        int64_t N, int E, int K, int kind, const double* mean,
                        const double* scale, double* out_log_pdf, void* stream);

/* VonMisesFisherTrainer._fit (von_mises_fisher.py:119-144; normalize != 0 adds   */
/* the row normalisation of .fit, :105-107) / GaussianTrainer._fit with           */
/* covariance_type='spherical' (gaussian.py:152-193).  weights (B,K,N) f64 are    */
/* the per-class saliencies.  out_mean (B,K,E), out_scale (B,K).                  */
/* estimate_mixture_weight (distribution/mixture_model_utils.py:133-203) with reductions    */
/* over independent axes -- the options that couple frequency bins (weight_constant_axis    */
/* containing -3 ...): affiliation (Bo, Bi, K, N) f64, saliency (Bo, Bi, N) or NULL;          */
/* reduce_inner: average over Bi, reduce_n: over N (keepdims) -> out (Bo, Bi', K, N').        */
/* No saliency: mean (:188); saliency: masked sums L1-normalised over the classes with the   */
/* reference's `where(norm == 0, 1e-10)` (:190-201).  Serves the step-wise fit.               */
int pbbss_estimate_mixture_weight(pbbss_handle_t h, const double* affiliation,
                                  const double* saliency, int64_t Bo, int64_t Bi, int K,
                                  int64_t N, int reduce_inner, int reduce_n,
                                  double* out_weight, void* stream);

/* a3 as a stand-alone step: log_pdf_to_affiliation (mixture_model_utils.py:7-55) for the     */
/* step-wise loops of the mixtures whose E-step is not fused with it (weight_constant_axis    */
/* sets beyond the fused loops, frame-varying weights, diagonal covariances).  log_pdf        */
/* (B,K,N) f64; the weight of (b,k,n) is weight[b*wb + k*wk + n*wn] -- a zero stride          */
/* broadcasts, so (B,K,1), (K,1), (B,1,N), (1,K,N) ... arrays are passed as they are;         */
/* activity uint8 (B,K,N) or NULL; affiliation_eps clips without re-normalisation (:50-53).  */
int pbbss_log_pdf_to_affiliation(pbbss_handle_t h, const double* log_pdf, int64_t B, int K,
                                 int64_t N, const double* weight, int64_t wb, int64_t wk,
                                 int64_t wn, const uint8_t* activity, double affiliation_eps,
                                 double* out_affiliation, void* stream);

/* log_pdf_to_affiliation_for_integration_models_with_inline_pa                                */
/* (mixture_model_utils.py:58-130) as a stand-alone step: per bin f the class permutation of   */
/* the spatial log-pdf (F,K,T) that maximises sum_{k,t} softmax_k(lp)(t) lp_k(t),              */
/* lp = spatial[perm] + spectral (itertools.permutations order, first strict maximum wins,     */
/* no weights in the search), then log_pdf_to_affiliation of that sum with the weights         */
/* (strides as above), the activity mask and the clip.  out_permutation (F,K) int32 or NULL.   */
/* K <= 6 (PBBSS_ERR_UNSUPPORTED beyond).  The fused joint fits run the same search in-kernel. */
int pbbss_log_pdf_to_affiliation_inline_pa(pbbss_handle_t h, const double* spatial_log_pdf,
                                           const double* spectral_log_pdf, int64_t F, int K,
                                           int64_t T, const double* weight, int64_t wb,
                                           int64_t wk, int64_t wn, const uint8_t* activity,
                                           double affiliation_eps, double* out_affiliation,
                                           int32_t* out_permutation, void* stream);

int pbbss_embed_fit(pbbss_handle_t h, const void* y, int y_is_f64, int64_t B,
                    int64_t N, int E, int K, int kind, int normalize,
                    const double* weights, double min_concentration,
                    double max_concentration, double* out_mean,
                    double* out_scale, void* stream);

/* ------------------------------------------------------------------------- */
/* N3 (sibling)  Full-covariance Gaussians  distribution/gaussian.py:17-56,        */
/* 152-193 (Gaussian, GaussianTrainer._fit with covariance_type='full').           */
/* y (B,N,E) real row-major, float32 / float64; 1 <= E <= 63.                      */
/* fit: weights (B,K,N) f64 per-class saliencies -> out_mean (B,K,E),              */
/*      out_covariance (B,K,E,E) (weighted scatter about the mean on the FP64      */
/*      matrix pipe).                                                              */
/* log_pdf: mean (B,K,E), covariance (B,K,E,E) -> out (B,K,N) f64, evaluated as    */
/*      the reference writes it (precision Cholesky of sklearn applied as          */
/*      P (y - mean), gaussian.py:46-50); out_status int32 (1) gets                */
/*      PBBSS_ST_NOT_POSDEF where the Cholesky factorisation fails.  K <= 64.       */
/* ------------------------------------------------------------------------- */
int pbbss_gauss_full_fit(pbbss_handle_t h, const void* y, int y_is_f64, int64_t B, int64_t N,
                         int E, int K, const double* weights, double* out_mean,
                         double* out_covariance, void* stream);
int pbbss_gauss_full_log_pdf(pbbss_handle_t h, const void* y, int y_is_f64, int64_t B,
                             int64_t N, int E, int K, const double* mean,
                             const double* covariance, double* out_log_pdf,
                             int32_t* out_status, void* stream);

typedef struct pbbss_mix_opts {
  int32_t iterations;       /* EM iterations; 0 = E-step only with the given model */
  int32_t kind;             /* PBBSS_EMBED_* of the spectral half (joint models)   */
  int32_t weight_mode;      /* vmfmm: PBBSS_WEIGHT_*; joint: PBBSS_JOINT_WEIGHT_*   */
  int32_t embedding_is_f64;
  int32_t obs_is_c128;
  int32_t final_predict;    /* write model.predict(...) to out_affiliation         */
  int32_t inline_pa;        /* inline permutation alignment (mixture_model_utils.py:58) */
  int32_t covariance_norm;  /* PBBSS_COVNORM_* of the cACG half                    */
  double min_concentration, max_concentration;
  double affiliation_eps, eigenvalue_floor;
  double spatial_weight, spectral_weight;
  int32_t sharded;          /* pbbss_joint_fit: the call holds ONE RANK'S block of the frequency  */
                            /* bins; the spectral M-step sums (gaussian.py:152-193,  */
                            /* von_mises_fisher.py:122-144) and bin-constant class weights are   */
                            /* all-reduced over the communicator of pbbss_comm_create, in stream */
                            /* order.  Not for PBBSS_EMBED_GAUSS_FULL.                           */
  int32_t reserved;
} pbbss_mix_opts;

/* GMMTrainer.fit / fit_predict, GMM.predict (distribution/gmm.py:17-171) with               */
/* covariance_type='full' (its default): arguments as pbbss_gmm_fit below, with              */
/* in/out_covariance and fixed_covariance (B,K,E,E); E <= 63.  out_status int32 (1) gets     */
/* PBBSS_ST_NOT_POSDEF if a class covariance stops being positive definite (the reference    */
/* raises sklearn's ValueError from its precision Cholesky, gaussian.py:26).                  */
int pbbss_gmm_full_fit(pbbss_handle_t h, const void* y, int64_t B, int64_t N, int E, int K,
                       const double* gamma0, const double* in_mean,
                       const double* in_covariance, const double* in_weight,
                       const double* saliency, const double* fixed_covariance,
                       const pbbss_mix_opts* opts, double* out_mean, double* out_covariance,
                       double* out_weight, double* out_affiliation, double* out_log_pdf,
                       int32_t* out_status, void* stream);

/* N2  VMFMMTrainer.fit / fit_predict, VMFMM.predict  distribution/vmfmm.py:19-172. */
/* y (B,N,E) real, raw (rows are unit-normalised here, vmfmm.py:76-78).           */
/* gamma0 (B,K,N) affiliations, or (iterations == 0) a model in_mean (B,K,E),      */
/* in_concentration (B,K), in_weight (B,K).  saliency (B,N) or NULL.  Outputs:    */
/* mean (B,K,E), concentration (B,K), weight (B,K); optional affiliation /        */
/* log_pdf (B,K,N) from the final E-step.  All float64.                           */
int pbbss_vmfmm_fit(pbbss_handle_t h, const void* y, int64_t B, int64_t N, int E,
                    int K, const double* gamma0, const double* in_mean,
                    const double* in_concentration, const double* in_weight,
                    const double* saliency, const pbbss_mix_opts* opts,
                    double* out_mean, double* out_concentration,
                    double* out_weight, double* out_affiliation,
                    double* out_log_pdf, void* stream);

/* N3 (sibling)  GMMTrainer.fit / fit_predict, GMM.predict  distribution/gmm.py:17-171 for  */
/* covariance_type = 'spherical' (SphericalGaussian, gaussian.py:100-137, 152-193): the same */
/* loop on the RAW rows of y (B,N,E).  in/out_covariance (B,K) = the scalar variances;        */
/* fixed_covariance (B,K) or NULL (gmm.py:160-167).  Otherwise as pbbss_vmfmm_fit.            */
int pbbss_gmm_fit(pbbss_handle_t h, const void* y, int64_t B, int64_t N, int E, int K,
                  const double* gamma0, const double* in_mean, const double* in_covariance,
                  const double* in_weight, const double* saliency,
                  const double* fixed_covariance, const pbbss_mix_opts* o, double* out_mean,
                  double* out_covariance, double* out_weight, double* out_affiliation,
                  double* out_log_pdf, void* stream);

/* N3  GCACGMMTrainer.fit / GCACGMM.predict  distribution/gcacgmm.py:47-333 and    */
/*     VMFCACGMMTrainer.fit / VMFCACGMM.predict  distribution/vmfcacgmm.py:43-301. */
/* observation (F,T,D) complex raw, embedding (F,T,E) real raw; one mixture over   */
/* all F*T points for the spectral half, one cACG per (f,k).  Class weights by     */
/* weight_constant_axis:                                                          */
#define PBBSS_JOINT_WEIGHT_FK 0       /* (-1,)        -> weight (F,K)          */
#define PBBSS_JOINT_WEIGHT_UNIFORM 1  /* -2 in axis   -> weight (1) = 1/K      */
#define PBBSS_JOINT_WEIGHT_K 2        /* (-3,-1)      -> weight (K)            */
#define PBBSS_JOINT_WEIGHT_KT 3       /* (-3,)        -> weight (K,T)          */
#define PBBSS_JOINT_WEIGHT_CONST 4    /* (-3,-2,-1)   -> weight (1) = 1        */
/* Initialisation: gamma0 (F,K,T), or (iterations == 0) a model (in_eigvec c128    */
/* (F,K,D,D), in_eigval (F,K,D), in_weight (shape above), in_mean (K,E), in_scale). */
/* in_scale / out_scale by opts->kind: VMF concentration (K); GAUSS_SPHERICAL       */
/* variance (K); GAUSS_DIAG per-dimension variances (K,E); GAUSS_FULL covariance    */
/* (K,E,E), E <= 63 (`covariance_type` of GCACGMMTrainer.fit, gcacgmm.py:141,       */
/* :297-303 -> GaussianTrainer._fit, gaussian.py:152-193).  With gamma0, a non-NULL */
/* in_scale is the `fixed_covariance` of gcacgmm.py:305-312.  saliency (F,T) or     */
/* NULL.  Outputs as the inputs' shapes; out_status int32 (F,K), for GAUSS_FULL     */
/* word 0 also carries PBBSS_ST_NOT_POSDEF of the spectral covariances;             */
/* out_affiliation (F,K,T) = model.predict (final_predict).                         */
int pbbss_joint_fit(pbbss_handle_t h, const void* observation,
                    const void* embedding, int64_t F, int T, int D, int E, int K,
                    const double* gamma0, const void* in_eigvec,
                    const double* in_eigval, const double* in_weight,
                    const double* in_mean, const double* in_scale,
                    const double* saliency, const pbbss_mix_opts* opts,
                    void* out_eigvec, double* out_eigval, double* out_weight,
                    double* out_mean, double* out_scale, int32_t* out_status,
                    double* out_affiliation, void* stream);

/* ------------------------------------------------------------------------- */
/* N4  Remaining members of the beamformer family, extraction/beamformer.py.     */
/* All arrays complex128 (interleaved float64) unless noted, 2 <= D <= 8 for the */
/* LCMV solve, D < 30 otherwise.                                                 */
/* ------------------------------------------------------------------------- */
/* get_lcmv_vector (:414-456): atf (K,F,D), response (K) (rounded to complex64 as  */
/* the reference does), noise (F,D,D) -> w (F,D); status[f] = PBBSS_ST_SINGULAR    */
/* where a least-squares fallback was taken (stable_solve, math/solve.py:95-114). */
int pbbss_lcmv(pbbss_handle_t h, const void* atf, const void* response,
               const void* noise, int64_t F, int D, int K, void* out_w,
               int32_t* out_status, void* stream);
/* phase_correction (:517-560): vector viewed as (lead, rest, F, D); the running    */
/* product of the inter-frequency phasors runs along the frequency axis for 2-D    */
/* input (two_d != 0, lead = rest = 1) and along the FIRST axis otherwise, as       */
/* np.cumprod(..., axis=0) does in the reference.  scratch: lead*rest*(F-1) c128.  */
int pbbss_phase_correction(pbbss_handle_t h, const void* vector, int64_t lead,
                           int64_t rest, int F, int D, int two_d, void* scratch,
                           void* out, void* stream);
/* mvdr_snr_postfilter (:502-509): (w^H T w)/(w^H N w) -> out (F) c128.            */
int pbbss_snr_postfilter(pbbss_handle_t h, const void* w, const void* target,
                         const void* noise, int64_t F, int D, void* out,
                         void* stream);
/* get_optimal_reference_channel (:601-624): w_mat, target, noise (F,D,D) c128 ->    */
/* num[f,r] = w_r^H T w_r, den[f,r] = w_r^H N w_r (F,D) c128 for every column r; the  */
/* caller sums over f (all-reduces under bin sharding) and takes the argmax.        */
int pbbss_reference_channel_terms(pbbss_handle_t h, const void* w_mat, const void* target,
                                  const void* noise, int64_t F, int D, void* out_num,
                                  void* out_den, void* stream);
/* rank-one approximation a a^H tr(C) / tr(a a^H) (beamformer_wrapper.py:18-25,     */
/* :61-69): covariance (N,D,D), vector (N,D) c128 -> out (N,D,D).                   */
int pbbss_rank_one_approximation(pbbss_handle_t h, const void* covariance,
                                 const void* vector, int64_t N, int D, void* out,
                                 void* stream);
/* y = M x per problem (the ATF estimate Phi_nn w_gev, beamformer_wrapper.py:28-48): */
/* matrix (N,D,D), vector (N,D) c128 -> out (N,D).                                  */
int pbbss_matvec(pbbss_handle_t h, const void* matrix, const void* vector, int64_t N, int D,
                 void* out, void* stream);
/* distortionless_normalization (:491-499) -> out (F,D).                           */
int pbbss_distortionless_normalization(pbbss_handle_t h, const void* w,
                                       const void* atf, const void* noise,
                                       int64_t F, int D, void* out, void* stream);
/* zero_degree_normalization (:512-514): vector (N,D) -> out (N,D).                */
int pbbss_zero_degree_normalization(pbbss_handle_t h, const void* vector, int64_t N,
                                    int D, int reference_channel, void* out,
                                    void* stream);
/* condition_covariance (:563-569): x (N,D,D) -> out (N,D,D).  out must not alias  */
/* x (PBBSS_ERR_INVALID_ARG when out == x): entries are processed independently.   */
int pbbss_condition_covariance(pbbss_handle_t h, const void* x, int64_t N, int D,
                               double gamma, void* out, void* stream);
/* apply_online_beamforming_vector (:586-598): vector (T,F,D) c128, mix (F,D,T)    */
/* c64/c128 -> out (F,T) c128.                                                     */
int pbbss_apply_online_beamforming_vector(pbbss_handle_t h, const void* vector,
                                          const void* mix, int mix_is_c128,
                                          int64_t F, int T, int D, void* out,
                                          void* stream);

/* ------------------------------------------------------------------------- */
/* STFT edge.  The reference has no transform of its own: */
/* its tests call nara_wpe.utils.stft / istft                                   */
/* (tests/test_distribution/test_spatial_mm.py:4,17-22,                          */
/* pb_bss/transform/griffin_lim_module.py:37); these two entry points replace    */
/* those calls.  The caller passes the windows (device float64): the periodic    */
/* analysis window for pbbss_stft, the biorthogonal synthesis window             */
/* w / sum_m w[n + m shift]^2 for pbbss_istft.                                   */
/* ------------------------------------------------------------------------- */
/* Number of frames of a (faded, padded) signal of num_samples samples. */
int pbbss_stft_num_frames(int64_t num_samples, int size, int shift, int window_length,
                          int fading, int pad);
/* x (C, N) float32 / float64 -> out complex64 / complex128, T = pbbss_stft_num_frames:
 * out_layout 0: (C, T, size/2+1) like the reference's stft;
 * out_layout 1: (size/2+1, T, C), the 'd t f -> f t d' rearrangement the mixture-model
 *               trainers are fed with (test_spatial_mm.py:41), written directly.
 * size: power of two, 4..8192; 1 <= window_length <= size; fading: window_length - shift
 * zeros on both sides; pad: the last partial frame is zero-padded (else dropped). */
int pbbss_stft(pbbss_handle_t h, const void* x, int x_is_f64, int64_t C, int64_t N, int size,
               int shift, int window_length, const double* window, int fading, int pad,
               int out_layout, int out_is_c128, void* out, void* stream);
/* X (C, T, size/2+1) complex64 / complex128 -> out (C, n_out) float64 with
 * n_out = T * shift + window_length - shift - (fading ? 2 (window_length - shift) : 0);
 * frames are added in ascending order (numpy.add.at over the segment view). */
int pbbss_istft(pbbss_handle_t h, const void* X, int x_is_c128, int64_t C, int T, int size,
                int shift, int window_length, const double* synthesis_window, int fading,
                double* out, int64_t n_out, void* stream);

/* ------------------------------------------------------------------------- */
/* Timing hook for bench.py: runs `fit` with HIP events recorded on `stream`   */
/* around the EM kernel launch(es) only and returns the elapsed milliseconds   */
/* of the most recent call (the roofline figure needs the kernel duration on   */
/* the launch stream; torch.cuda.Event only sees torch's current stream).      */
/* ------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------- */
/* (e)  Multi-GPU: frequency bins sharded over the GPUs of one node, ONE        */
/* exchange step -- the all-gather of the posterior masks before permutation    */
/* alignment (permutation_alignment.py:334 needs all bins of an utterance;      */
/* the EM itself needs no collective under weight_constant_axis=(-1,),          */
/* cacgmm.py:151, :204).  One process per GPU; the RCCL communicator lives in   */
/* the handle.  Rank 0 calls pbbss_comm_unique_id (128 bytes), the host carries */
/* the id to the other ranks (MPI, a file, a socket, torch.distributed ...),    */
/* every rank calls pbbss_comm_create.  RCCL is dlopen()ed on first use:        */
/* PBBSS_ERR_UNSUPPORTED where librccl is absent.                               */
/*                                                                             */
/* pbbss_shard_bounds: the contiguous block [start, stop) of `total_bins` owned */
/* by `rank` (sizes differ by at most one: 513 over 8 ranks = 65 + 7 x 64).     */
/* pbbss_allgather_masks: local (outer, stop - start, inner) -> out (outer,     */
/* total_bins, inner) on every rank; elem_bytes 8 (float64) or 4 (float32);     */
/* e.g. masks (U, F_local, K, T): outer = U, inner = K * T.  Enqueued on        */
/* `stream`: pad to the largest block, one ncclAllGather, trim.                 */
/* pbbss_allgather_unpack: the trimming half on its own (gathered = (world,     */
/* outer, ceil(total_bins / world), inner)) for hosts that bring their own      */
/* collective.  The pack / gather buffers belong to the communicator (released  */
/* by pbbss_comm_destroy); collectives of one handle must be enqueued in the     */
/* same order on every rank (RCCL's rule), i.e. from one host thread per handle. */
/* pbbss_comm_info: world size and rank as RCCL itself reports them              */
/* (ncclCommCount / ncclCommUserRank) -- for logs that prove how many GPUs took  */
/* part.  The same communicator carries the all-reduce of sharded joint fits     */
/* (pbbss_mix_opts.sharded).                                                     */
/* ------------------------------------------------------------------------- */
int pbbss_comm_unique_id(void* out_id_128_bytes);
int pbbss_comm_create(pbbss_handle_t h, const void* unique_id, int world_size, int rank);
int pbbss_comm_destroy(pbbss_handle_t h);
int pbbss_comm_info(pbbss_handle_t h, int* out_world_size, int* out_rank);
int pbbss_shard_bounds(int64_t total_bins, int world_size, int rank, int64_t* out_start,
                       int64_t* out_stop);
int pbbss_allgather_masks(pbbss_handle_t h, const void* local, int elem_bytes, int64_t outer,
                          int64_t total_bins, int64_t inner, void* out, void* stream);
int pbbss_allgather_unpack(pbbss_handle_t h, const void* gathered, int elem_bytes,
                           int world_size, int64_t outer, int64_t total_bins, int64_t inner,
                           void* out, void* stream);

int pbbss_set_timing(pbbss_handle_t h, int enable);
/* Tail handling of pbbss_cacgmm_fit (on by default): when B = m * CUs + r with 1 <= r <= 8
 * (e.g. 513 = 2 * 256 + 1 frequency bins) the r remainder problems are run as "split"
 * groups -- several workgroups share one problem's frames and exchange partial sums
 * through L2 -- concurrently on an internal side stream, so that no CU hosts an extra
 * full workgroup.  A bounded inter-workgroup wait that times out (peers not co-resident under
 * heavy contention) never hangs: it sets PBBSS_ST_NONFINITE | PBBSS_ST_EIG_NOCONV in the status
 * words of the launch's split problems (the Python layer then raises like the reference's
 * finiteness assert) and a sticky flag that pbbss_split_error reads synchronously (0 = no
 * wait of this handle ever timed out).  The same switch governs the generic-size path
 * (9 <= D <= 32 or K > 6): there the r <= CUs / 16 remainder bins run as their own chain of
 * launches on the side stream, forked and joined once per fit (no inter-workgroup waits). */
int pbbss_set_split_tail(pbbss_handle_t h, int enable);
/* pbbss_dhtv_calculate_mapping: workgroups that share ONE utterance (used for few utterances,
 * where a single workgroup is bound by one CU's L2 latency): 0 = automatic (default: the
 * frame-slice kernel where it fits -- K <= 5, more than 64 frames, all teams co-resident --
 * else the bin-chunk team kernel / one workgroup per utterance), 1 = always one workgroup per
 * utterance, >= 2 = frame-slice kernel with at most that many workgroups per utterance (64 or
 * 128 frames each), -2..-32 = bin-chunk team kernel of that size (A/B). */
int pbbss_set_dhtv_team(pbbss_handle_t h, int workgroups_per_utterance);
/* pbbss_dhtv_calculate_mapping, frame-slice kernel only: evaluate the first iteration of EVERY
 * plan segment on the input as it stands (all segments at once, P teams per utterance) before
 * the plan proper.  A segment in which no bin asks for a permutation -- and none of whose bins an
 * earlier segment has permuted since -- would break out of the sequential walk of
 * permutation_alignment.py:331-353 unchanged, so the plan kernel skips it; when that holds for
 * every segment the mapping is the identity and the plan kernel returns at once.  Results are
 * identical with and without the probe.  F = 513, T = 500, K = 3 (profiles/r04_i_inline_aligner.txt):
 * probe 0.021 ms; plan 0.33 ms -> 0.22 ms with the ~10-40 flipped bins of the first EM iterations,
 * ~0.1 ms with the 0-4 of a settled EM, ~0.003 ms for aligned masks.  Off by default (a one-shot
 * alignment of freshly fitted masks touches every segment anyway); the inline aligner of
 * CACGMMTrainer.fit (cacgmm.py:260-267) switches it on.
 * flags: bit 0 = the probe; bit 1 = the frame-slice path does NOT write the aligned features
 * into `scratch` (a caller that only wants the mapping saves a K F T float64 store; scratch is
 * still needed as the exchange area and holds unspecified data afterwards).  0..3. */
int pbbss_set_dhtv_probe(pbbss_handle_t h, int flags);
int pbbss_split_error(pbbss_handle_t h, int* out_flag);
/* Consume a reported time-out: waits for the device, then re-zeroes the arrival counters of the
 * split / member protocols and the sticky flag of pbbss_split_error.  The Python layer calls it
 * before it repeats a fit without split groups, so that (a) a LATER genuine NONFINITE |
 * EIG_NOCONV status is not mistaken for another time-out and (b) counters left non-zero by an
 * aborted launch cannot corrupt the next split launch of the handle.  (New in this library; the
 * reference has no inter-process state to reset.) */
int pbbss_split_reset(pbbss_handle_t h);
/* Test knob: the number of polls after which a bounded inter-workgroup wait (split groups,
 * cooperative shared-weight kernel, DHTV teams) gives up; 0 restores the defaults (seconds).
 * A tiny value makes the waits run out although the peers ARE co-resident, i.e. it provokes the
 * real time-out path (status poison + pbbss_split_error, DHTV status) that the Python layer
 * answers with a repeat on a path without inter-workgroup waits.  The DHTV part is a device
 * global: it applies to every handle of the process on this device. */
int pbbss_set_spin_limit(pbbss_handle_t h, unsigned polls);
/* Development aid: device buffer of 64 uint64 receiving per-phase shader-cycle sums
 * of the EM kernel ([wave 0..3][phase 0..7]); only written by library builds made
 * with -DPBBSS_PHASE_PROFILE (`make prof`), ignored otherwise.  NULL disables. */
int pbbss_set_phase_profile(pbbss_handle_t h, void* dev_counters);
int pbbss_last_kernel_ms(pbbss_handle_t h, float* out_ms);
/* Duration of the timed region `lag` launches ago (0 = the most recent one = pbbss_last_kernel_ms;
 * up to 3).  Waits only for THAT launch: a caller that reads lag = 2 after every launch keeps two
 * launches queued behind the running one, whereas reading the most recent one drains the queue and
 * exposes the host's launch latency on the device (~30 us per step, tools/launch_gap.py).  For
 * pbbss_cacgmm_fit the region is the EM kernel itself (events attached to its dispatch). */
int pbbss_kernel_ms_lagged(pbbss_handle_t h, int lag, float* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* PBBSS_H_ */
