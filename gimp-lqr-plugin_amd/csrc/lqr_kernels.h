// lqr_kernels.h -- every kernel of the engine, declared for the host shim (lqr_shim.hip), which launches them, and included by
// the file that defines each (so that declaration and definition cannot drift apart).  The template kernels are instantiated
// explicitly at the end of their files for exactly the parameter sets the shim launches; a missing one is a link error
// (-Wl,-z,defs in the Makefile).
#pragma once
#include "lqr_common.h"

// k_energy.hip
__global__ void k_wk_init(const DevCarver *cs, int w, int h, int stride, int ch);
__global__ __launch_bounds__(256) void k_wk_init_visible(const DevCarver *cs, int w0, int h, int stride, int ch);
template <int NRG> __global__ void k_emap_full(const DevCarver *cs, DpK p, int w, int h, int stride);
__global__ void k_mask_add(float *plane, int w0, const uint8_t *mask, int channels, int mw, int x0, int y0, int x1, int y1,
                           int nx, int ny, int transposed, int is_rig, int bias_factor);
template <int NRG, int EU_NT> __global__ __launch_bounds__(64) void k_emap_update(const DevCarver *cs, DpK p, int w, int h, int stride, int k, int epoch);
__global__ __launch_bounds__(256) void k_frozen_catchup(const DevCarver *cs, int from, int to, int w_from, int h, int stride);

// k_backtrack.hip
__global__ __launch_bounds__(VPATH_THREADS) void k_vpath(const DevCarver *cs, int w, int h, int stride, int lr, int delta, int log_index, int moved_unit);
template <int DELTA> __global__ __launch_bounds__(VPATH_THREADS) void k_vpath1(const DevCarver *cs, int w, int h, int stride, int lr, int log_index, int moved_unit);

template <int DELTA> __global__ __launch_bounds__(256) void k_vp_maps(const DevCarver *cs, int w, int h, int stride);
template <int DELTA> __global__ __launch_bounds__(VPATH_THREADS) void k_vp_solve(const DevCarver *cs, int w, int h, int stride, int lr, int log_index, int moved_unit);

// k_carve.hip
__global__ __launch_bounds__(256) void k_carve(const DevCarver *cs, int w, int h, int stride, int delta, int move_dp);
template <int NRG> __global__ __launch_bounds__(256) void k_carve_e(const DevCarver *cs, DpK p, int w, int h, int stride, int move_dp, int k, int epoch);

// k_band.hip
template <int PXT, bool UPDATE, int NTH = DP_THREADS> __global__ __launch_bounds__(NTH) void k_dp_sweep(const DevCarver *cs, DpK p, int w, int h, int stride, int lr);
__global__ __launch_bounds__(64) void k_band_update(const DevCarver *cs, DpK p, int w, int h, int stride, int lr);
template <int PXL, int NW, int R, bool LR, bool RIG> __global__ __launch_bounds__(64 * NW) void k_band_update_mw(const DevCarver *cs, DpK p, int w, int h, int stride);
template <int NW, bool LR, bool RIG> __global__ __launch_bounds__(128 * NW) void k_band_update_tw(const DevCarver *cs, DpK p, int w, int h, int stride, int *dev_err);

// k_tiles.hip
template <bool LR, bool RIG> __global__ __launch_bounds__(64) void k_dp_tile(const DevCarver *cs, DpK p, int w, int h, int stride, int y0);
template <int PX, bool LR, bool RIG, bool UPDATE, int DELTA = 1, bool RIGM = false, int HLN = 16>
__global__ __launch_bounds__(64 * DPP_W) void k_dp_tile_p(DevCarver *cs, DpK p, int w, int h, int stride, unsigned long long *exch, int epoch, int *dev_err);

// k_levels.hip
template <bool LR, bool RIG, int DELTA, bool RIGM>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_band_levels(DevCarver *cs, DpK p, int w, int h, int stride, unsigned long long *exch, int epoch, int *dev_err, int P, int n_img);

// k_oneoff.hip
__global__ __launch_bounds__(256) void k_vs_commit(const DevCarver *cs, int w0, int h0, int wc0, int n_seams, int first_level, int finish);
__global__ __launch_bounds__(256) void k_inflate(const InflateDev *jobs, int w0, int w1, int l, int max_level, int *dev_err);
__global__ __launch_bounds__(256) void k_seam_check(const DevCarver *cs, int h, int wc0, int n_seams, int delta, int *dev_err);
__global__ __launch_bounds__(256) void k_vs_rollback(const DevCarver *cs, size_t n, int first_level, int finish_level);
__global__ void k_inject(const DevCarver *cs, int what, int h, int w0, int log_index, int first_level);
__global__ __launch_bounds__(256) void k_compact(const uint8_t *rgb, const int32_t *vs, const float *bias, const float *rig,
                                                  uint8_t *nrgb, float *nbias, float *nrig, int32_t *nvmap, int w0, int w, int ch, int level, int depth);
__global__ __launch_bounds__(256) void k_compact_jobs(const InflateDev *jobs, int w0, int w, int level);
__global__ void k_transpose(const InflateDev *jobs, int w, int h);
__global__ __launch_bounds__(256) void k_mask_line_max(const uint8_t *mask, int channels, int width, int a0, int b0, int line_len, int direction, int *out);
