// STFT / inverse STFT launchers (stft.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pbbss.h"

namespace pbbss {

// x (C, N) real -> out (C, T, size/2+1) [layout 0] or (size/2+1, T, C) [layout 1]
int launch_stft(const void* x, int x_is_f64, int64_t C, int64_t N, int size, int shift, int wl,
                const double* window, int fade, int T, int layout, int out_c128, void* out,
                size_t lds_limit, hipStream_t s);

// X (C, T, size/2+1) -> out (C, n_out) float64; frames: (C, T, wl) float64 scratch
int launch_istft(const void* X, int x_is_c128, int64_t C, int T, int size, int shift, int wl,
                 const double* window, int fade, double* frames, double* out, int64_t n_out,
                 size_t lds_limit, hipStream_t s);

}  // namespace pbbss
