// Host-callable launchers of the extraction kernels (beamform.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "em_launch.hpp"

namespace pbbss {
int launch_heev(const double* a, int64_t N, int D, double* val, double* vec, int32_t* st,
                hipStream_t s);
int launch_gev(const double* t, const double* nn, int64_t N, int D, double* w, int32_t* st,
               hipStream_t s);
// gev_general.hip: non-Hermitian generalized eigenproblem (use_eig=True)
int launch_gev_general(const double* t, const double* nn, int64_t N, int D, double* w,
                       double* lambda, int32_t* st, hipStream_t s);
int launch_solve(const double* A, const double* Bm, int64_t N, int D, int M, double* x,
                 int32_t* st, hipStream_t s);
int launch_mvdr_souden(const double* t, const double* nn, int64_t N, int D, double eps, int mode,
                       double* mat, double* num, double* den, int32_t* st, hipStream_t s);
int launch_mvdr(const double* atf, const double* nn, int64_t N, int D, double* w, int32_t* st,
                hipStream_t s);
int launch_ban(const double* w, const double* nn, int64_t N, int D, double* out, hipStream_t s);
int launch_apply(const double* w, const void* x, int x128, int64_t B, int T, int D, double* out,
                 hipStream_t s, int64_t xmod = 0, int64_t b_first = 0);
// reference channel of get_mvdr_vector_souden for L problems (beamformer.py:601-624, :690-698)
int launch_select_reference_channel(const double* mat, const double* num, const double* den,
                                    int64_t L, int64_t F, int D, int64_t lead_stride,
                                    int64_t bin_stride, double eps, double* out_w, int32_t* out_ref,
                                    int32_t* out_ok, hipStream_t s);
int launch_normalize(const void* y, int is128, int64_t B, int T, int D, void* out, hipStream_t s);
int launch_psd(const void* x, int x128, int64_t B, int T, int D, int K, const double* mask,
               int normalize, double* out, const EmLaunchCfg& cfg, hipStream_t s);
// bf_extra.hip
int launch_lcmv(const double* atf, const double* response, const double* noise, int64_t F, int D,
                int K, double* w, int32_t* st, hipStream_t s);
int launch_phase_correction(const double* v, int64_t lead, int64_t rest, int F, int D, int two_d,
                            double* scratch_u, double* out, hipStream_t s);
int launch_refch_terms(const double* wm, const double* tp, const double* nn, int64_t F, int D,
                       double* num, double* den, hipStream_t s);
int launch_rank_one(const double* cov, const double* a, int64_t N, int D, double* out, hipStream_t s);
int launch_matvec(const double* m, const double* x, int64_t N, int D, double* y, hipStream_t s);
int launch_bf_quadratic(int mode, const double* w, const double* m1, const double* m2,
                        const double* atf, int64_t F, int D, double* out, hipStream_t s);
int launch_zero_degree(const double* v, int64_t N, int D, int ref, double* out, hipStream_t s);
int launch_condition_covariance(const double* x, int64_t N, int D, double gamma, double* out,
                                hipStream_t s);
int launch_apply_online(const double* v, const void* mix, int mix_is_c128, int64_t F, int T, int D,
                        double* out, hipStream_t s);
}  // namespace pbbss
