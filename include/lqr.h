/*
 * lqr.h -- the LqrCarver C ABI as consumed by gimp-lqr-plugin, GLib-free.
 *
 * This header declares exactly the liblqr-1 surface that the plug-in's
 * render() path binds; a maintainer builds the plug-in
 * against this header and links the MI355X engine (liblqr-hip.so) instead of
 * -llqr-1.  Every prototype cites the reference call
 * site(s) it replaces (paths relative to the gimp-lqr-plugin tree).
 *
 * The reference includes <lqr.h> at src/render.c:25, src/io_functions.c:22,
 * src/main.c:28, src/interface_I.h; src/io_functions.h:22-24 hard-errors
 * unless __LQR_H__ is defined, so the guard name is part of the contract.
 *
 * The same header is used to build the CPU oracle (oracle/), with every public
 * symbol renamed olqr_* by oracle/oracle_rename.h so that both libraries can
 * live in one test process.
 */
#ifndef __LQR_H__
#define __LQR_H__

#ifdef __cplusplus
extern "C" {
#endif

/* ---- GLib-free spellings of the GLib typedefs the reference passes ------ */
#ifndef LQR_NO_GLIB_TYPEDEFS
typedef int gint;
typedef unsigned int guint;
typedef unsigned char guchar;
typedef char gchar;
typedef float gfloat;
typedef double gdouble;
typedef int gboolean;
typedef void *gpointer;
#endif
#ifndef TRUE
#define TRUE 1
#endif
#ifndef FALSE
#define FALSE 0
#endif

#define LQR_MAX_NAME_LENGTH (1024)          /* render.c:113-114 buffers */
#define LQR_PROGRESS_MAX_MESSAGE_LENGTH (1024)

/* ---- return values: the plug-in tests only == LQR_NOMEM (render.c:43,45) - */
typedef enum _LqrRetVal {
    LQR_ERROR = 0,      /* generic error            */
    LQR_OK = 1,         /* ok                        */
    LQR_NOMEM = 2,      /* not enough (device) memory */
    LQR_USRCANCEL = 3   /* action cancelled          */
} LqrRetVal;

/* legacy helper macros used at io_functions.c:46,94,125,247 */
#define LQR_CATCH(expr) do { LqrRetVal ret_val__; \
    if ((ret_val__ = (expr)) != LQR_OK) { return ret_val__; } } while (0)
#define LQR_CATCH_F(expr) do { if ((expr) == FALSE) { return LQR_ERROR; } } while (0)
#define LQR_CATCH_MEM(expr) do { if ((expr) == NULL) { return LQR_NOMEM; } } while (0)
#define LQR_TRY_N_N(expr) do { if ((expr) == NULL) { return NULL; } } while (0)
#define CATCH(expr) LQR_CATCH(expr)
#define CATCH_F(expr) LQR_CATCH_F(expr)
#define CATCH_MEM(expr) LQR_CATCH_MEM(expr)
#define TRY_N_N(expr) LQR_TRY_N_N(expr)

/* ---- enums (numeric order is ABI: batch/batch-gimp-lqr.scm passes raw ints) */
typedef enum _LqrResizeOrder {
    LQR_RES_ORDER_HOR = 0,      /* width first  (main.c:78 default) */
    LQR_RES_ORDER_VERT = 1      /* height first */
} LqrResizeOrder;

typedef enum _LqrEnergyFuncBuiltinType {   /* interface.c:2138-2145 */
    LQR_EF_GRAD_NORM = 0,
    LQR_EF_GRAD_SUMABS = 1,
    LQR_EF_GRAD_XABS = 2,                  /* plug-in default, main.c:77 */
    LQR_EF_LUMA_GRAD_NORM = 3,
    LQR_EF_LUMA_GRAD_SUMABS = 4,
    LQR_EF_LUMA_GRAD_XABS = 5,
    LQR_EF_NULL = 6
} LqrEnergyFuncBuiltinType;

/* ---- opaque types ------------------------------------------------------- */
typedef struct _LqrCarver LqrCarver;
typedef struct _LqrCarverList LqrCarverList;
typedef struct _LqrVMap LqrVMap;
typedef struct _LqrVMapList LqrVMapList;
typedef struct _LqrProgress LqrProgress;

typedef LqrRetVal (*LqrProgressFuncInit) (const gchar *init_message);
typedef LqrRetVal (*LqrProgressFuncUpdate) (gdouble percentage);
typedef LqrRetVal (*LqrProgressFuncEnd) (const gchar *end_message);
typedef LqrRetVal (*LqrVMapFunc) (LqrVMap *vmap, gpointer data);

/* ---- lifecycle ---------------------------------------------------------- */
/* render.c:222,894 -- takes ownership of `buffer` (malloc/g_malloc'd,
 * row-major, interleaved, w*h*channels bytes; channels in 1..4 =
 * GRAY, GRAYA, RGB, RGBA as produced by io_functions.c:29-68).  NULL on OOM. */
LqrCarver *lqr_carver_new(guchar *buffer, gint width, gint height, gint channels);
/* render.c:224 -- root carver only; allocates the device-resident DP maps. */
LqrRetVal lqr_carver_init(LqrCarver *r, gint delta_x, gfloat rigidity);
/* render.c:376, interface_I.c:427 -- frees root, attached carvers, vmaps, progress. */
void lqr_carver_destroy(LqrCarver *r);
/* render.c:897 -- aux must have the same w*h; aux replays the root's seams. */
LqrRetVal lqr_carver_attach(LqrCarver *r, LqrCarver *aux);

/* ---- configuration ------------------------------------------------------ */
LqrRetVal lqr_carver_set_energy_function_builtin(LqrCarver *r, LqrEnergyFuncBuiltinType ef_ind); /* render.c:234 */
void lqr_carver_set_resize_order(LqrCarver *r, LqrResizeOrder resize_order);                    /* render.c:235 */
void lqr_carver_set_progress(LqrCarver *r, LqrProgress *p);                                     /* render.c:236 */
void lqr_carver_set_side_switch_frequency(LqrCarver *r, guint switch_frequency);                /* render.c:237 */
LqrRetVal lqr_carver_set_enl_step(LqrCarver *r, gfloat enl_step);                               /* render.c:238 */
gfloat lqr_carver_get_enl_step(LqrCarver *r);                                                   /* render.c:551,658 */
void lqr_carver_set_dump_vmaps(LqrCarver *r);                                                   /* render.c:241 */

/* ---- masks (buffers are consumed during the call; caller frees) ---------- */
/* io_functions.c:94-95 */
LqrRetVal lqr_carver_bias_add_rgb_area(LqrCarver *r, guchar *rgb, gint bias_factor, gint channels,
                                       gint width, gint height, gint x_off, gint y_off);
/* io_functions.c:125-126 */
LqrRetVal lqr_carver_rigmask_add_rgb_area(LqrCarver *r, guchar *rgb, gint channels,
                                          gint width, gint height, gint x_off, gint y_off);

/* ---- run ---------------------------------------------------------------- */
LqrRetVal lqr_carver_resize(LqrCarver *r, gint w1, gint h1);   /* render.c:318,328,529 */
LqrRetVal lqr_carver_flatten(LqrCarver *r);                    /* render.c:325,636     */

/* ---- readout ------------------------------------------------------------ */
/* io_functions.c:155 -- *rgb is engine-owned scratch valid until the next call;
 * returns FALSE (and rewinds) after the last line. */
gboolean lqr_carver_scan_line(LqrCarver *r, gint *n, guchar **rgb);
gboolean lqr_carver_scan_by_row(LqrCarver *r);                 /* io_functions.c:157 */
void lqr_carver_scan_reset(LqrCarver *r);

/* ---- getters ------------------------------------------------------------ */
gint lqr_carver_get_width(LqrCarver *r);          /* not used by the plug-in; exported for symmetry */
gint lqr_carver_get_height(LqrCarver *r);         /* io_functions.c:145,168 */
gint lqr_carver_get_channels(LqrCarver *r);       /* render.c:49 */
gint lqr_carver_get_ref_width(LqrCarver *r);      /* render.c:547,654 */
gint lqr_carver_get_ref_height(LqrCarver *r);     /* render.c:548,655 */
gint lqr_carver_get_orientation(LqrCarver *r);    /* render.c:549,656 */
gint lqr_carver_get_depth(LqrCarver *r);          /* render.c:550,657 */

/* ---- attached-carver list (iteration order = attach order) --------------- */
LqrCarverList *lqr_carver_list_start(LqrCarver *r);            /* render.c:370,496,839,912 */
LqrCarver *lqr_carver_list_current(LqrCarverList *list);       /* render.c:841,914 */
LqrCarverList *lqr_carver_list_next(LqrCarverList *list);

/* ---- visibility ("seam") maps -------------------------------------------- */
LqrVMap *lqr_vmap_dump(LqrCarver *r);                          /* render.c:725; NULL on OOM; caller-owned */
void lqr_vmap_destroy(LqrVMap *vmap);
gint *lqr_vmap_get_data(LqrVMap *vmap);                        /* io_functions.c:218; row-major W0*H0 */
gint lqr_vmap_get_width(LqrVMap *vmap);                        /* io_functions.c:216 */
gint lqr_vmap_get_height(LqrVMap *vmap);                       /* io_functions.c:217 */
gint lqr_vmap_get_depth(LqrVMap *vmap);                        /* io_functions.c:219, render.c:747 */
gint lqr_vmap_get_orientation(LqrVMap *vmap);
LqrVMapList *lqr_vmap_list_start(LqrCarver *r);                /* render.c:344; maps owned by the carver */
LqrVMap *lqr_vmap_list_current(LqrVMapList *list);
LqrVMapList *lqr_vmap_list_next(LqrVMapList *list);
LqrRetVal lqr_vmap_list_foreach(LqrVMapList *list, LqrVMapFunc func, gpointer data);   /* io_functions.c:312 */

/* ---- progress (callbacks fire synchronously on the calling thread) -------- */
LqrProgress *lqr_progress_new(void);                                            /* render.c:770 */
LqrRetVal lqr_progress_set_init(LqrProgress *p, LqrProgressFuncInit init_func);       /* render.c:772 */
LqrRetVal lqr_progress_set_update(LqrProgress *p, LqrProgressFuncUpdate update_func); /* render.c:773 */
LqrRetVal lqr_progress_set_end(LqrProgress *p, LqrProgressFuncEnd end_func);          /* render.c:774 */
LqrRetVal lqr_progress_set_update_step(LqrProgress *p, gfloat update_step);
LqrRetVal lqr_progress_set_init_width_message(LqrProgress *p, const gchar *message);  /* render.c:775 */
LqrRetVal lqr_progress_set_init_height_message(LqrProgress *p, const gchar *message); /* render.c:776-777 */
LqrRetVal lqr_progress_set_end_width_message(LqrProgress *p, const gchar *message);
LqrRetVal lqr_progress_set_end_height_message(LqrProgress *p, const gchar *message);

/* ======================================================================== */
/* Extensions (lqrx_*): not part of liblqr-1.  Test/bench hooks implemented   */
/* identically by the engine and by the oracle, plus the batch entry point.  */
/* ======================================================================== */

/* Energy map (E4) of the carver's current carved frame, row-major
 * lqrx_carver_frame_width() x lqrx_carver_frame_height() floats in CARVER
 * orientation.  Flattens first if the visible width is not the carved width
 * (as liblqr's lqr_carver_get_true_energy does). */
LqrRetVal lqrx_carver_get_energy(LqrCarver *r, gfloat *buffer);
gint lqrx_carver_frame_width(LqrCarver *r);
gint lqrx_carver_frame_height(LqrCarver *r);
/* Debug snapshot of the DP state in the carved frame (w x h, carver
 * orientation): en, m and the back-pointer as dx = parent_x - x.  Any pointer
 * may be NULL.  Only valid on an initialised root carver whose maps exist. */
LqrRetVal lqrx_carver_debug_maps(LqrCarver *r, gfloat *en, gfloat *m, gint *least_dx);
/* With lqrx_set_debug(1), every visibility-map build snapshots (en, m, dx) of
 * the carved frame after its last seam, just before the maps are dropped;
 * lqrx_carver_debug_snapshot copies it out (debug_width x debug_height). */
void lqrx_set_debug(gint on);
gint lqrx_carver_debug_width(LqrCarver *r);
gint lqrx_carver_debug_height(LqrCarver *r);
LqrRetVal lqrx_carver_debug_snapshot(LqrCarver *r, gfloat *en, gfloat *m, gint *least_dx);
/* Visible image at the current size in IMAGE orientation (row-major,
 * get_width*get_height*channels bytes): the same bytes the scan_line loop of
 * io_functions.c:155-164 would assemble, in one call. */
LqrRetVal lqrx_carver_read_image(LqrCarver *r, guchar *out);
/* Same bytes as the scan lines, but written straight into a caller-provided
 * DEVICE buffer (get_width*get_height*channels bytes in CARVER orientation, i.e.
 * image orientation unless lqr_carver_get_orientation() is 1): lets a batch
 * driver hand results to RCCL without a host round trip. */
LqrRetVal lqrx_carver_read_image_device(LqrCarver *r, void *device_ptr);
/* The plug-in's auto-size helper guess_new_size (src/layers_combo.c:275-392): the new size that
 * would remove the discard mask = old size - max over lines of the number of mask pixels whose
 * value (mean colour / 255, times alpha / 255) is >= 0.5 / colour_channels.  `direction` 0 = horizontal
 * (count along rows), 1 = vertical (count along columns); the mask layer is width x height x
 * channels at offset (x_off, y_off) relative to the old_width x old_height image. */
gint lqrx_guess_new_size(const guchar *mask, gint channels, gint width, gint height, gint x_off, gint y_off,
                         gint old_width, gint old_height, gint direction);
/* Carve n independent carvers (same geometry and configuration) in lock-step;
 * equivalent to calling lqr_carver_resize on each.  The engine runs them as one
 * batched launch sequence (the per-frame batch axis). */
LqrRetVal lqrx_carver_resize_batch(LqrCarver **carvers, gint n, gint w1, gint h1);

/* The plug-in's seam-map colour ramp, write_vmap_to_layer's per-pixel loop (src/io_functions.c:249-279),
 * over a dumped map: out_rgba receives get_width*get_height RGBA pixels, row-major.  For vs != 0:
 * value = (double)(depth+1-vs)/(depth+1); R,G,B = (guchar)(255*(value*col_start + (1-value)*col_end));
 * A = (guchar)(255*0.5*(1+value)); vs == 0 -> 0,0,0,0.  col_* are GimpRGB's r,g,b doubles in [0,1]
 * (src/io_functions.c:196,208-209).  The engine runs it as one streaming kernel. */
LqrRetVal lqrx_vmap_to_rgba(LqrVMap *vmap, const gdouble col_start[3], const gdouble col_end[3], guchar *out_rgba);
/* ENGINE ONLY (the oracle returns LQR_ERROR).  Start n carvers over from images that already sit in
 * device memory: equivalent to lqr_carver_destroy + lqr_carver_new(buffer, w, h, channels) +
 * lqr_carver_init(delta_x, rigidity) with the same geometry, channels, delta_x and rigidity, except
 * that the pixels come from device_rgb[i] (w*h*channels interleaved bytes, image orientation) by a
 * device-to-device copy, and that the configuration set through lqr_carver_set_* is kept.  Masks,
 * the visibility map and dumped maps are dropped.  Carvers with attached carvers are refused.
 * This is how a batch driver with HBM-resident inputs (bench.py) reuses one set of carvers. */
LqrRetVal lqrx_carver_reload_device_batch(LqrCarver **carvers, gint n, void *const *device_rgb);

#ifdef __cplusplus
}
#endif

#endif /* __LQR_H__ */
