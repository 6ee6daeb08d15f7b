// ba_kernels.h -- host-callable launchers of the bundle-adjustment kernels (ba_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ba_types.h"

namespace pvba {
size_t linearize_lds_bytes(const Dims &dm);
size_t dense_lds_bytes(const Dims &dm, int *lds_matrix);
size_t dense_tile_doubles(const Dims &dm);
int tiles_per_thread(const Dims &dm);
int tp_landmark_slots(const Dims &dm, int want); // landmarks per chunk of the large-window landmark role (ba_lin_tp.h); want: the landmarks 256 factors belong to
hipError_t launch_linearize(const View &v, hipStream_t st);
hipError_t launch_reduce(const View &v, hipStream_t st, int phase = 0); // phase: see k_reduce
hipError_t launch_dense(const View &v, hipStream_t st);
hipError_t launch_backsub(const View &v, hipStream_t st);
hipError_t launch_back_reduce(const View &v, double *back_local, hipStream_t st);
hipError_t launch_quality(const View &v, hipStream_t st, int buf_from_ctrl, double *err_sum, double *pack = nullptr);
hipError_t launch_reset(const View &v, const double *fs_init, const double *rho_init, const Ctrl *tmpl, hipStream_t st);
struct GatherArgs {
    const void *src[8];
    uint32_t words[8], off[8]; // 32-bit words: length of segment k, its offset in the destination
};
hipError_t launch_gather(const GatherArgs &a, void *dst, hipStream_t st);
hipError_t launch_prior_prep(const double *S, const double *s, int D, double *Lambda, double *eta, double *ST, hipStream_t st);
} // namespace pvba
