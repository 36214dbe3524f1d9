/*
 * lqr_hip.h -- thin C-ABI shim between the C host side of the engine
 * (gimp-lqr-plugin_amd/host/lqr_carver.c, which implements include/lqr.h) and
 * the hand-written gfx950 kernels (gimp-lqr-plugin_amd/csrc/lqr_hip.hip).
 *
 * Plain pointers and sizes only; no C++ or torch types.  Each entry point names
 * the liblqr engine stage it replaces and the
 * reference call site that reaches that stage.
 *
 * Everything is batch-native: a LqrHipBatch is n carvers of identical geometry
 * and configuration that advance in lock-step, one grid.y (or grid.z) slice per
 * image.  A single carver is a batch of one.  All work of a batch is enqueued
 * on the batch's HIP stream; only lqrhip_batch_sync and the read-back calls
 * block.  Every function returns 0 on success, LQRHIP_ENOMEM on device OOM and
 * another negative value on any other HIP error; none of them aborts, and no kernel
 * traps: a device-side failure (a persistent grid that was not co-resident, a failed
 * self-check of the session) is recorded in a host-visible word and returned by the next
 * lqrhip_batch_sync / lqrhip_device_sync / lqrhip_inflate as LQRHIP_EFAULT; the host side then rolls
 * the session back (lqrhip_session_rollback) and redoes it on the kernels without spin waits.
 *
 * Threading: like the plug-in's use of liblqr (GTK main loop / PDB run), the library
 * is single-threaded by contract -- the allocation cache, the error word and the
 * profiling records are process globals without locks, and the device is selected
 * (hipSetDevice) for the thread that first calls in.  No environment variable is read
 * except LOCAL_RANK.
 */
#ifndef LQR_HIP_H
#define LQR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define LQRHIP_OK 0
#define LQRHIP_ENOMEM (-2)
#define LQRHIP_EHIP (-1)
#define LQRHIP_EARG (-3)
#define LQRHIP_EFAULT (-4)   /* a kernel of the session gave up or a self-check failed: roll back, redo (lqrhip_session_rollback) */
#define LQRHIP_MAX_DELTA 16

typedef struct LqrHipCarver LqrHipCarver;   /* device-resident planes of one carver */
typedef struct LqrHipBatch LqrHipBatch;     /* n carvers advancing in lock-step     */

/* DP parameters that stay fixed while a visibility map is being built */
typedef struct LqrHipDpParams {
    int delta_x;                                  /* lqr_carver_init, render.c:224 */
    int use_rigidity;                             /* rigidity != 0 */
    float rigidity_map[2 * LQRHIP_MAX_DELTA + 1]; /* [dx + delta_x], host-computed (powf) */
    int nrg_func;                                 /* LqrEnergyFuncBuiltinType, render.c:234 */
    int nrg_radius;                               /* 1 for gradients, 0 for LQR_EF_NULL */
    int w_start;                                  /* bias is divided by this (E4) */
} LqrHipDpParams;

/* -- device / lifetime ----------------------------------------------------- */
/* Selects the HIP device (LOCAL_RANK, else 0) on first use.  Returns the
 * device ordinal or a negative error when no gfx950 device is usable. */
int lqrhip_init(void);
const char *lqrhip_last_error(void);

/* E1 lqr_carver_new (render.c:222,894): upload the interleaved u8 image as the
 * base layout.  The host buffer is not retained. */
LqrHipCarver *lqrhip_carver_create(const unsigned char *rgb, int w, int h, int channels);
void lqrhip_carver_destroy(LqrHipCarver *c);
/* Start over on an existing carver, as lqr_carver_destroy + lqr_carver_new + lqr_carver_init would
 * (render.c:376,222,224), but from an image that is already in HBM: the base layout becomes the
 * w x h x channels interleaved u8 image at `device_rgb` (device-to-device copy on the shim's stream),
 * the visibility map is cleared, masks are dropped; working planes are kept when the geometry is the
 * same.  For batch drivers that keep their inputs device-resident (bench.py). */
int lqrhip_carver_reset(LqrHipCarver *c, const void *device_rgb, int w, int h);
/* wait for the copies lqrhip_carver_reset enqueued */
int lqrhip_reset_sync(void);
/* attached carver (render.c:897): shares the root's visibility map */
int lqrhip_carver_attach(LqrHipCarver *root, LqrHipCarver *aux);
/* E1 lqr_carver_init (render.c:224): allocate the working planes
 * (packed pixels, en, m, back-pointers, seam buffers) for a w x h frame. */
int lqrhip_carver_activate(LqrHipCarver *c);

/* E2 lqr_carver_bias_add_rgb_area / lqr_carver_rigmask_add_rgb_area
 * (io_functions.c:94-95,125-126).  The carver must be flat.  `transposed`
 * says whether the carver frame is the transpose of image orientation. */
int lqrhip_mask_add(LqrHipCarver *c, const unsigned char *mask, int channels, int width, int height,
                    int x_off, int y_off, int transposed, int is_rigmask, int bias_factor);

/* -- batch ------------------------------------------------------------------ */
/* into how many device batches (HIP streams) the host splits a lock-step group of n carvers.  Automatic (set 0): 4 for
 * groups of 32 carvers and more when the process has the hardware queues for them (the HIP runtime's GPU_MAX_HW_QUEUES
 * >= 8 in the environment before HIP initialises; its default of 4 makes the split 30 % slower than one stream, so it is
 * then not made), else 1.  The library sets the variable to 8 itself when it is loaded with the variable unset into a
 * process whose GPU runtime is not up yet.  lqrhip_set_sub_batches(n > 0) pins the number of streams. */
int lqrhip_sub_batches(int n);
/* Images of carved-frame width w that one lock-step batch may hold and still run delta_x = 2 / rigidity-mask carvers on the
 * tiled kernels (0: unknown); larger batches of such carvers are carved group after group (lqrx_carver_resize_batch). */
int lqrhip_general_batch_limit(int w);
/* the same for a given delta_x: 2 .. 4 may also run on k_band_levels (groups of 8 and more), 5 .. 10 on the full-width tiled kernels only */
int lqrhip_general_batch_limit_delta(int w, int delta_x);
void lqrhip_set_sub_batches(int n);
LqrHipBatch *lqrhip_batch_create(LqrHipCarver **carvers, int n);
/* tell a batch how many batches of its group run concurrently on their own streams (0 or 1: alone): kernels whose grid
 * must be co-resident are then never chosen (k_dp_tile_p spins on neighbour tiles) or sized so that ALL the siblings' grids
 * fit together (k_band_levels) */
void lqrhip_batch_set_shared(LqrHipBatch *b, int shared);
void lqrhip_batch_destroy(LqrHipBatch *b);
int lqrhip_batch_sync(LqrHipBatch *b);
/* after a failed call on the batch: drain its stream, discard the device-side error record of the failed call, and have
 * every live batch lay its descriptors and exchange areas out afresh */
void lqrhip_batch_abort(LqrHipBatch *b);
void *lqrhip_batch_stream(LqrHipBatch *b);      /* hipStream_t, for event timing in bench.py */

/* base layout -> working planes: what liblqr's raw[y][x] = y*w+x initialisation means for compacted planes.
 * from_visible = 0: identity map (the carver is flat).  from_visible = 1: the carved frame of a multi-size image, i.e. the
 * pixels of the base layout that carry no level yet, in order (a session redone after a fault; working planes that were lost) */
int lqrhip_wk_init(LqrHipBatch *b, int from_visible);
/* E3+E4 lqr_carver_build_emap: full energy map of the w x h carved frame */
int lqrhip_emap_build(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h);
/* E5 lqr_carver_build_mmap: full cumulative-min DP, tie rule by `leftright` */
int lqrhip_mmap_build(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h, int leftright);
/* One seam of lqr_carver_build_vsmap (reached from lqr_carver_resize,
 * render.c:318,328,529), on a frame that is w wide on entry:
 *   E7 build_vpath (argmin + backtrack) -> seam `log_index` of the session log,
 *   E8 carve (all working planes shift left of the seam, w -> w-1),
 *   E6 update_emap, then either E9 update_mmap (full_rebuild = 0) or E5
 *   build_mmap with the given (already toggled) leftright (full_rebuild = 1).
 * If w-1 == 1 only E7/E8 run (liblqr's finish_vsmap case). */
int lqrhip_seam_step(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h, int log_index,
                     int leftright_pick, int full_rebuild, int leftright_next);
/* reserve the session seam log (n_seams x h ints per carver) before the first seam_step */
int lqrhip_seam_log_reserve(LqrHipBatch *b, int n_seams, int h);
/* E8 update_vsmap for a whole session at once: turn the session's seam log
 * (n_seams seams, carved frame wc0 wide at session start) into visibility
 * levels first_level, first_level+1, ... in the base layout; `finish` applies
 * liblqr's finish_vsmap (last column gets level w0). */
int lqrhip_vs_commit(LqrHipBatch *b, int w0, int h0, int wc0, int n_seams, int first_level, int finish);
/* Session self-check, enqueued behind the last seam step and in front of lqrhip_vs_commit: the seam log of the session (n_seams
 * seams of h rows, carved frame wc0 wide at its start) must hold delta_x-connected seams inside their frames; a violation is
 * reported as LQRHIP_EFAULT by the next lqrhip_batch_sync.  lqrhip_inflate carries the second check (every level of the session
 * exactly once per row) and returns LQRHIP_EFAULT itself, without adopting anything.  lqrhip_set_selfcheck(0) turns both off. */
int lqrhip_session_check(LqrHipBatch *b, int h, int wc0, int n_seams, int delta_x);
void lqrhip_set_selfcheck(int on);
/* whether the host side redoes a session that ended in LQRHIP_EFAULT (default 1); 0: roll back and return LQR_ERROR (tests) */
void lqrhip_set_recovery(int on);
int lqrhip_get_recovery(void);
/* After LQRHIP_EFAULT: drain the batch's stream, drop the error record, remove the session's levels (>= first_level, and
 * finish_vsmap's w0) from the visibility map if they were committed.  The working planes are invalid afterwards. */
int lqrhip_session_rollback(LqrHipBatch *b, int w0, int h0, int first_level, int finish);
/* kernels without spin waits only (k_dp_tile, k_band_update_tw / _mw / k_band_update, k_dp_sweep): set for the redo of a session */
void lqrhip_batch_set_safe(LqrHipBatch *b, int safe);
/* after a spin time-out the whole process stays on those kernels (a shared or partitioned device); 0 re-arms the persistent ones */
void lqrhip_set_no_spin(int on);
int lqrhip_get_no_spin(void);
/* [0] spin time-outs, [1] failed activity predictions, [2] seam-log check failures, [3] level check failures, [4] sessions
 * rolled back, [5] faults injected, [6] sessions carved on the non-spinning kernels after a fault */
int lqrhip_fault_stats(unsigned long long *out8, int reset);
/* Test hook: provoke a fault in the next session(s).  kind 1 spin time-out / 2 failed prediction (the error word is written while
 * the kernels of seam step at_step run), 3 / 4 a seam-log entry out of the frame / disconnected, 5 / 6 a committed level cleared /
 * duplicated (at_step = commits to let pass first: a later sub-batch of a group); times = sessions hit in a row; kind 0 disarms. */
void lqrhip_debug_inject(int kind, int at_step, int times);
/* Test hook: the nth device allocation from now on (0 = the next one) fails once with LQRHIP_ENOMEM; -1 disarms.  What the host side
 * owes its caller then: LQR_NOMEM (the one value src/render.c:42-46 tests for) and a carver that is still consistent. */
void lqrhip_debug_fail_alloc(int nth);
/* The three passes that replace a batch's base planes run in TWO PHASES: the call stages the new planes and runs the pass -- nothing of
 * the carvers changes -- and lqrhip_planes_commit adopts what was staged.  A group commits only after every one of its sub-batches has
 * passed phase one, so that a failed self-check (LQRHIP_EFAULT) or a failed allocation (LQRHIP_ENOMEM) in one sub-batch leaves the
 * whole group where it was.  A staged pass is discarded by the next staging call, lqrhip_session_rollback, _batch_abort, _batch_destroy.
 * E14 lqr_carver_inflate(l) on roots and their attached carvers; carries the session's second self-check */
int lqrhip_inflate(LqrHipBatch *b, int w0, int h0, int l, int max_level);
/* E11 lqr_carver_flatten (render.c:325,636): keep pixels visible at `level` */
int lqrhip_flatten(LqrHipBatch *b, int w0, int h0, int w, int level);
/* E11 lqr_carver_transpose: base planes of a flat carver, w x h -> h x w */
int lqrhip_transpose(LqrHipBatch *b, int w, int h);
int lqrhip_planes_commit(LqrHipBatch *b);
int lqrhip_inflate_commit(LqrHipBatch *b);      /* = lqrhip_planes_commit */

/* -- read-back (blocking) --------------------------------------------------- */
/* E12 scan_line source: the pixels visible at `level`, packed w x h x channels
 * in CARVER orientation (io_functions.c:155-164 then serves rows of it). */
int lqrhip_read_visible(LqrHipCarver *c, int w0, int h0, int w, int level, unsigned char *out);
/* guess_new_size (src/layers_combo.c:275-392): max over lines of the count of mask pixels at or
 * above the threshold; returns the count (>= 0) or a negative error */
int lqrhip_mask_line_max(const unsigned char *mask, int channels, int width, int height, int a0, int b0, int n_lines, int line_len,
                         int direction);
/* same, but into a caller-provided device buffer (no host round trip) */
int lqrhip_read_visible_device(LqrHipCarver *c, int w0, int h0, int w, int level, void *device_out);
int lqrhip_device_sync(void);
/* free / total device memory in bytes (hipMemGetInfo) and the bytes the shim's allocation cache holds */
int lqrhip_mem_info(unsigned long long *free_bytes, unsigned long long *total_bytes, unsigned long long *cached_bytes);
/* Measured HBM ceiling of this device: a 16-byte-per-lane streaming copy of `bytes` bytes, `iters` times;
 * returns read+write GB/s through *gbps (the denominator bench.py reports next to the nominal 8 TB/s). */
int lqrhip_copy_bandwidth(unsigned long long bytes, int iters, double *gbps);
/* the colour ramp of write_vmap_to_layer (src/io_functions.c:249-279) over a
 * w x h visibility map: value = (depth+1-vs)/(depth+1), RGB = value*start + (1-value)*end,
 * alpha = 0.5*(1+value), each channel (guchar)(255*x); vs == 0 -> 0,0,0,0.  Host buffers. */
int lqrhip_vmap_to_rgba(const int *vmap, int w, int h, int depth, const double col_start[3], const double col_end[3],
                        unsigned char *out_rgba);
/* return the cached (freed) device blocks of the shim's allocation cache to the driver */
void lqrhip_pool_trim(void);
/* E12 lqr_vmap_dump (render.c:725): vs of the pixels visible at `level`
 * minus `depth` (0 stays 0), w x h in carver orientation */
int lqrhip_read_vmap(LqrHipCarver *c, int w0, int h0, int w, int level, int depth, int *out);
/* test hooks behind lqrx_carver_get_energy / lqrx_carver_debug_maps */
int lqrhip_read_working(LqrHipCarver *c, int w, int h, float *en, float *m, int *least_dx);

/* kernel-time accounting for bench.py: accumulated HIP-event time (ms) and
 * launch count of the carve kernel (the roofline kernel) since the last reset */
/* 0 off; 1 HIP-event pairs around every kernel of the seam loop; 2 around k_carve only (each pair costs ~10 us
 * of queue time, so the bench's timed region uses 2 and takes the other kernels' times from rocprofv3) */
void lqrhip_prof_enable(int on);
/* how E9 (update_mmap) runs: -1 by batch size (tiled full-width keep-rule sweep up to 8 4K images, the band
 * kernel k_band_update_tw above), 0 band kernel always, 1 tiled sweep whenever its grid fits the device,
 * 2 the per-row-barrier band kernel k_band_update_mw (the default for rows wider than 4200 px), 3 the generic
 * one-wave band kernel k_band_update + k_dp_sweep (what delta_x > 4 runs on) whatever the parameters, 5 k_band_levels
 * (4 was round 4's k_band_tiles, removed in round 6: it now behaves as 0) */
void lqrhip_set_update_mode(int mode);
/* Test hook: threads of the k_dp_sweep<UPDATE> launch behind the band kernels: 256 (default) or 1024 */
void lqrhip_set_sweep_threads(int n);
/* Test hook: the largest group that runs the carve and the energy update as one launch (k_carve_e, delta_x <= 2): 0 = never, 1 = the
 * default (4), n = groups up to n images */
void lqrhip_set_carve_fused(int max_images);
/* E7 form: -1 = the parallel two-kernel backtrack (k_vp_maps / k_vp_solve) for groups of up to par_max images (default 3; 0 keeps the
 * current value) of 1000 rows and more, the one-wave walk k_vpath1 otherwise; 0 = k_vpath1 always; 1 = the parallel form
 * always (delta_x 1 .. 4) */
void lqrhip_set_vpath_mode(int mode, int par_max);
/* Cap on the workgroups of the persistent tiled DP sweep (k_dp_tile_p), whose tiles spin on their neighbours and
 * must all be resident: -1 = the bound derived from the occupancy query at lqrhip_init, n >= 0 = min(n, that bound).
 * Grids above the cap run as k_dp_tile (one launch per 32 rows).  0 forces that path (tests). */
void lqrhip_set_dp_persistent_limit(int workgroups);
/* Geometry of the persistent tiled sweep: 0 = by batch size (3 = 2 px per lane, 32-column tiles with 48-column halos and 48-row blocks
 * while every tile has a compute unit to itself; 2 = 2 px per lane, 64-column tiles, 32-row blocks while twice the tiles fit the
 * residency bound; else 4 px per lane), 2 / 3 / 4 = pinned (tests) */
void lqrhip_set_dp_persistent_px(int px);
void lqrhip_prof_reset(void);
int lqrhip_prof_get(const char *kernel, double *ms_total, long long *launches, double *bytes_total);
/* time during which at least one launch of `kernel` was running (launches of sub-batch streams overlap) */
int lqrhip_prof_get_union(const char *kernel, double *ms_union);
/* Bytes the carves of this process HAD to move (read + write) since the last reset: k_vpath* knows, for the seam it found, how
 * many pixels lie on the side the carve will move, and sums pixels x bytes per pixel (en; m and back pointer unless a full DP
 * follows; the rigidity mask) over images and seams.  The roofline's numerator next to SURVEY 8(d)'s half-row figure. */
int lqrhip_moved_bytes(unsigned long long *bytes, int reset);
/* Test hook: slots (workgroups) per image of k_band_levels (-1 automatic, 0 never, n exactly n). */
void lqrhip_set_band_levels(int slots);
/* k_band_levels' events since the last reset: [0] images stopped by THREE active tiles on one slot (a slot takes two, one per
 * wave; the full-width sweep took over), [1] synchronous (mispredicted or second-tile) loads, [2] tile-levels processed,
 * [3] slot-levels idle, [4] levels in which a slot had two tiles */
int lqrhip_band_levels_stats(unsigned long long *out8, int reset);

#ifdef __cplusplus
}
#endif
#endif /* LQR_HIP_H */
