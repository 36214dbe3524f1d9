/*
 * render_replay.c -- the reference plug-in's render path, replayed in C against
 * include/lqr.h on in-memory layers.
 *
 * TEST PROGRAM.  It is compiled by tests/test_c_replay.py against the public
 * header -- once with the header's built-in GLib-free typedefs, once with
 * -DLQR_NO_GLIB_TYPEDEFS and the small GLib stand-in block below (what a plug-in
 * build that includes <glib.h> first would see) -- and linked either to the
 * engine (liblqr-hip.so) or, through oracle/oracle_rename.h, to the CPU oracle.
 * A prototype that does not match what the reference passes fails to compile here.
 *
 * The call sequence is the one gimp-lqr-plugin issues:
 *   render_init_carver      src/render.c:211-248   (progress, new, init, masks, set_*, attach)
 *   render_noninteractive   src/render.c:318-376   (resize, LqR-back, vmaps, read-out, aux, destroy)
 *   update_bias/set_rigmask src/io_functions.c:70-131
 *   write_carver_to_layer   src/io_functions.c:155-164 (scan_line / scan_by_row)
 *   write_all_vmaps         src/io_functions.c:292-314 (lqr_vmap_list_foreach + callback)
 *   progress_init           src/render.c:761-779
 * with GIMP's drawables replaced by plain buffers.  Only lqr_* (liblqr-1) entry points are used.
 *
 * A second mode replays the INTERACTIVE path on one persistent carver:
 *   render_interactive      src/render.c:465-574   (resize, the getters at :547-551, read-out, aux layers)
 *   render_flatten          src/render.c:576-681   (lqr_carver_flatten, getters, read-out)
 *   render_dump_vmap        src/render.c:683-759   (lqr_vmap_dump on a caller-owned map + lqr_vmap_get_*)
 * driven by --steps=r120x100,d,f,...  (r WxH = render_interactive to that size, f = render_flatten, d = render_dump_vmap).
 *
 * Built with -DREPLAY_DLOPEN the program links to NO carving library: every lqr_* entry point the header declares is
 * resolved with dlsym from the shared object named by --lib=PATH (--prefix=o for the oracle's renamed exports), so the
 * same binary replays the plug-in's sequence on the engine, on the oracle, or on a genuine liblqr-1.so.0 where one exists.
 *
 * usage: render_replay [--lib=PATH [--prefix=P]] [--steps=...] CASE.bin OUT.bin
 *   CASE.bin: int32 header[20] then the image and the mask layers (see read_case)
 *   OUT.bin:  int32 records (see the emit_* helpers), compared with tests/harness.py's results
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef LQR_NO_GLIB_TYPEDEFS
/* what <glib.h> would have provided before <lqr.h> is included */
typedef int gint;
typedef unsigned int guint;
typedef unsigned char guchar;
typedef char gchar;
typedef float gfloat;
typedef double gdouble;
typedef gint gboolean;
typedef void *gpointer;
typedef int gint32;
#else
typedef int gint32;
#endif

#include <lqr.h>

#ifndef __LQR_H__
#error "lqr.h must define __LQR_H__ (src/io_functions.h:22-24)"
#endif

#ifdef REPLAY_DLOPEN
/* every function include/lqr.h declares from liblqr-1, through a table filled by dlsym */
#include <dlfcn.h>
#define LQR_FUNCS(X) \
    X(lqr_carver_attach) \
    X(lqr_carver_bias_add_rgb_area) \
    X(lqr_carver_destroy) \
    X(lqr_carver_flatten) \
    X(lqr_carver_get_channels) \
    X(lqr_carver_get_depth) \
    X(lqr_carver_get_enl_step) \
    X(lqr_carver_get_height) \
    X(lqr_carver_get_orientation) \
    X(lqr_carver_get_ref_height) \
    X(lqr_carver_get_ref_width) \
    X(lqr_carver_get_width) \
    X(lqr_carver_init) \
    X(lqr_carver_list_current) \
    X(lqr_carver_list_next) \
    X(lqr_carver_list_start) \
    X(lqr_carver_new) \
    X(lqr_carver_resize) \
    X(lqr_carver_rigmask_add_rgb_area) \
    X(lqr_carver_scan_by_row) \
    X(lqr_carver_scan_line) \
    X(lqr_carver_scan_reset) \
    X(lqr_carver_set_dump_vmaps) \
    X(lqr_carver_set_energy_function_builtin) \
    X(lqr_carver_set_enl_step) \
    X(lqr_carver_set_progress) \
    X(lqr_carver_set_resize_order) \
    X(lqr_carver_set_side_switch_frequency) \
    X(lqr_progress_new) \
    X(lqr_progress_set_end) \
    X(lqr_progress_set_end_height_message) \
    X(lqr_progress_set_end_width_message) \
    X(lqr_progress_set_init) \
    X(lqr_progress_set_init_height_message) \
    X(lqr_progress_set_init_width_message) \
    X(lqr_progress_set_update) \
    X(lqr_progress_set_update_step) \
    X(lqr_vmap_destroy) \
    X(lqr_vmap_dump) \
    X(lqr_vmap_get_data) \
    X(lqr_vmap_get_depth) \
    X(lqr_vmap_get_height) \
    X(lqr_vmap_get_orientation) \
    X(lqr_vmap_get_width) \
    X(lqr_vmap_list_current) \
    X(lqr_vmap_list_foreach) \
    X(lqr_vmap_list_next) \
    X(lqr_vmap_list_start)
#define X(name) static __typeof__(name) *p_##name;
LQR_FUNCS(X)
#undef X
static int load_lib(const char *path, const char *prefix)
{
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    char sym[128];
    int missing = 0;
    if (!h) { fprintf(stderr, "replay: dlopen(%s): %s\n", path, dlerror()); return 0; }
#define X(name) snprintf(sym, sizeof sym, "%s%s", prefix, #name); *(void **) (&p_##name) = dlsym(h, sym); \
    if (!p_##name) { fprintf(stderr, "replay: %s does not export %s\n", path, sym); missing++; }
    LQR_FUNCS(X)
#undef X
    return missing == 0;
}
#define lqr_carver_attach (*p_lqr_carver_attach)
#define lqr_carver_bias_add_rgb_area (*p_lqr_carver_bias_add_rgb_area)
#define lqr_carver_destroy (*p_lqr_carver_destroy)
#define lqr_carver_flatten (*p_lqr_carver_flatten)
#define lqr_carver_get_channels (*p_lqr_carver_get_channels)
#define lqr_carver_get_depth (*p_lqr_carver_get_depth)
#define lqr_carver_get_enl_step (*p_lqr_carver_get_enl_step)
#define lqr_carver_get_height (*p_lqr_carver_get_height)
#define lqr_carver_get_orientation (*p_lqr_carver_get_orientation)
#define lqr_carver_get_ref_height (*p_lqr_carver_get_ref_height)
#define lqr_carver_get_ref_width (*p_lqr_carver_get_ref_width)
#define lqr_carver_get_width (*p_lqr_carver_get_width)
#define lqr_carver_init (*p_lqr_carver_init)
#define lqr_carver_list_current (*p_lqr_carver_list_current)
#define lqr_carver_list_next (*p_lqr_carver_list_next)
#define lqr_carver_list_start (*p_lqr_carver_list_start)
#define lqr_carver_new (*p_lqr_carver_new)
#define lqr_carver_resize (*p_lqr_carver_resize)
#define lqr_carver_rigmask_add_rgb_area (*p_lqr_carver_rigmask_add_rgb_area)
#define lqr_carver_scan_by_row (*p_lqr_carver_scan_by_row)
#define lqr_carver_scan_line (*p_lqr_carver_scan_line)
#define lqr_carver_scan_reset (*p_lqr_carver_scan_reset)
#define lqr_carver_set_dump_vmaps (*p_lqr_carver_set_dump_vmaps)
#define lqr_carver_set_energy_function_builtin (*p_lqr_carver_set_energy_function_builtin)
#define lqr_carver_set_enl_step (*p_lqr_carver_set_enl_step)
#define lqr_carver_set_progress (*p_lqr_carver_set_progress)
#define lqr_carver_set_resize_order (*p_lqr_carver_set_resize_order)
#define lqr_carver_set_side_switch_frequency (*p_lqr_carver_set_side_switch_frequency)
#define lqr_progress_new (*p_lqr_progress_new)
#define lqr_progress_set_end (*p_lqr_progress_set_end)
#define lqr_progress_set_end_height_message (*p_lqr_progress_set_end_height_message)
#define lqr_progress_set_end_width_message (*p_lqr_progress_set_end_width_message)
#define lqr_progress_set_init (*p_lqr_progress_set_init)
#define lqr_progress_set_init_height_message (*p_lqr_progress_set_init_height_message)
#define lqr_progress_set_init_width_message (*p_lqr_progress_set_init_width_message)
#define lqr_progress_set_update (*p_lqr_progress_set_update)
#define lqr_progress_set_update_step (*p_lqr_progress_set_update_step)
#define lqr_vmap_destroy (*p_lqr_vmap_destroy)
#define lqr_vmap_dump (*p_lqr_vmap_dump)
#define lqr_vmap_get_data (*p_lqr_vmap_get_data)
#define lqr_vmap_get_depth (*p_lqr_vmap_get_depth)
#define lqr_vmap_get_height (*p_lqr_vmap_get_height)
#define lqr_vmap_get_orientation (*p_lqr_vmap_get_orientation)
#define lqr_vmap_get_width (*p_lqr_vmap_get_width)
#define lqr_vmap_list_current (*p_lqr_vmap_list_current)
#define lqr_vmap_list_foreach (*p_lqr_vmap_list_foreach)
#define lqr_vmap_list_next (*p_lqr_vmap_list_next)
#define lqr_vmap_list_start (*p_lqr_vmap_list_start)
#endif

#define MEM_CHECK_N(x) do { if ((x) == NULL) { fprintf(stderr, "replay: out of memory\n"); return NULL; } } while (0)
#define MEM_CHECK1_N(x) do { if ((x) == LQR_NOMEM) { fprintf(stderr, "replay: LQR_NOMEM\n"); return NULL; } } while (0)
#define MEM_CHECK1(x) do { if ((x) == LQR_NOMEM) { fprintf(stderr, "replay: LQR_NOMEM\n"); return FALSE; } } while (0)

typedef struct {            /* an in-memory "layer" */
    gint w, h, bpp;
    guchar *px;
} Layer;

typedef struct {            /* the engine-facing subset of PlugInVals (src/main_common.h:34-60) */
    gint new_width, new_height;
    gint pres_coeff, disc_coeff;
    gfloat rigidity;
    gint delta_x;
    gfloat enl_step;        /* percent */
    gint nrg_func, res_order;
    gint output_seams, resize_aux_layers, scaleback, no_disc_on_enlarge;
} Vals;

static FILE *out;
static void emit(gint v) { fwrite(&v, sizeof v, 1, out); }
static void emit_bytes(const guchar *p, size_t n) { fwrite(p, 1, n, out); }

/* ---- progress callbacks: the plug-in casts gimp_progress_init / _update into these slots
 * (render.c:772-773); here they count the calls, which the Python side compares ------------------ */
static gint n_init, n_update, n_end;
static LqrRetVal my_progress_init(const gchar *message) { (void) message; n_init++; return LQR_OK; }
static LqrRetVal my_progress_update(gdouble percentage) { (void) percentage; n_update++; return LQR_OK; }
static LqrRetVal my_progress_end(const gchar *message) { (void) message; n_end++; return LQR_OK; }

static LqrProgress *progress_init(void)                         /* render.c:761-779 */
{
    LqrProgress *progress = lqr_progress_new();
    MEM_CHECK_N(progress);
    lqr_progress_set_init(progress, (LqrProgressFuncInit) my_progress_init);
    lqr_progress_set_update(progress, (LqrProgressFuncUpdate) my_progress_update);
    lqr_progress_set_end(progress, (LqrProgressFuncEnd) my_progress_end);
    lqr_progress_set_init_width_message(progress, "Resizing width...");
    lqr_progress_set_init_height_message(progress, "Resizing height...");
    return progress;
}

static guchar *rgb_buffer_from_layer(const Layer *l)            /* io_functions.c:29-68: a fresh copy the carver will own */
{
    size_t n = (size_t) l->w * l->h * l->bpp;
    guchar *b = (guchar *) malloc(n ? n : 1);
    if (b) memcpy(b, l->px, n);
    return b;
}

static LqrRetVal update_bias(LqrCarver *carver, const Layer *l, gint bias_factor, gint x_off, gint y_off)   /* io_functions.c:70-100 */
{
    guchar *rgb_buffer;
    if (!l || bias_factor == 0) return LQR_OK;
    CATCH_MEM(rgb_buffer = rgb_buffer_from_layer(l));
    CATCH(lqr_carver_bias_add_rgb_area(carver, rgb_buffer, bias_factor, l->bpp, l->w, l->h, x_off, y_off));
    free(rgb_buffer);                                           /* the callee copied it (:97) */
    return LQR_OK;
}

static LqrRetVal set_rigmask(LqrCarver *carver, const Layer *l, gint x_off, gint y_off)                      /* io_functions.c:102-131 */
{
    guchar *rgb_buffer;
    if (!l) return LQR_OK;
    CATCH_MEM(rgb_buffer = rgb_buffer_from_layer(l));
    CATCH(lqr_carver_rigmask_add_rgb_area(carver, rgb_buffer, l->bpp, l->w, l->h, x_off, y_off));
    free(rgb_buffer);
    return LQR_OK;
}

static LqrCarver *attach_aux_carver(LqrCarver *carver, const Layer *l, gint width, gint height)              /* render.c:881-900 */
{
    guchar *rgb_buffer;
    LqrCarver *aux_carver;
    if (l) {
        rgb_buffer = rgb_buffer_from_layer(l);
        MEM_CHECK_N(rgb_buffer);
        aux_carver = lqr_carver_new(rgb_buffer, width, height, l->bpp);
        MEM_CHECK_N(aux_carver);
        MEM_CHECK1_N(lqr_carver_attach(carver, aux_carver));
    }
    return carver;
}

static gboolean compute_ignore_disc_mask(const Vals *v, gint ow, gint oh, gint nw, gint nh)                  /* render.c:794-821 */
{
    if (!v->no_disc_on_enlarge) return FALSE;
    if (v->res_order == LQR_RES_ORDER_HOR) return (nw > ow) || (nw == ow && nh > oh);
    return (nh > oh) || (nh == oh && nw > ow);
}

/* write_carver_to_layer, io_functions.c:134-182: rows or columns, as the carver says */
static LqrRetVal write_carver_to_layer(LqrCarver *r, Layer *dst)
{
    gint y, k, bpp = lqr_carver_get_channels(r);
    guchar *out_line;
    gint w = dst->w, h = dst->h;
    gint lines = 0;
    while (lqr_carver_scan_line(r, &y, &out_line)) {
        if (lqr_carver_scan_by_row(r)) {
            memcpy(dst->px + (size_t) y * w * bpp, out_line, (size_t) w * bpp);            /* gimp_pixel_rgn_set_row */
        } else {
            for (k = 0; k < h; k++) memcpy(dst->px + ((size_t) k * w + y) * bpp, out_line + (size_t) k * bpp, bpp);   /* _set_col */
        }
        lines++;
    }
    emit(lines);
    return LQR_OK;
}

/* write_vmap_to_layer's accessor part, io_functions.c:216-219; registered through lqr_vmap_list_foreach (:312) */
static LqrRetVal dump_vmap(LqrVMap *vmap, gpointer data)
{
    gint w = lqr_vmap_get_width(vmap), h = lqr_vmap_get_height(vmap), depth = lqr_vmap_get_depth(vmap);
    gint *buffer = lqr_vmap_get_data(vmap);
    gint *count = (gint *) data;
    emit(w); emit(h); emit(depth);
    fwrite(buffer, sizeof(gint), (size_t) w * h, out);
    (*count)++;
    return LQR_OK;
}

static LqrCarver *render_init_carver(const Layer *layer, const Layer *pres, const Layer *disc, const Layer *rigmask, const Vals *vals)
{
    LqrCarver *carver;
    LqrProgress *progress;
    guchar *rgb_buffer;
    gint old_width = layer->w, old_height = layer->h, bpp = layer->bpp, x_off = 0, y_off = 0;
    gfloat rigidity = rigmask ? 3 * vals->rigidity : vals->rigidity;                         /* rigidity_init, render.c:781-792 */
    gboolean ignore_disc_mask = compute_ignore_disc_mask(vals, old_width, old_height, vals->new_width, vals->new_height);

    progress = progress_init();
    MEM_CHECK_N(progress);
    rgb_buffer = rgb_buffer_from_layer(layer);                                              /* render.c:220 */
    MEM_CHECK_N(rgb_buffer);
    carver = lqr_carver_new(rgb_buffer, old_width, old_height, bpp);                        /* :222 */
    MEM_CHECK_N(carver);
    MEM_CHECK1_N(lqr_carver_init(carver, vals->delta_x, rigidity));                         /* :224 */
    MEM_CHECK1_N(update_bias(carver, pres, vals->pres_coeff, x_off, y_off));                /* :225-226 */
    if (!ignore_disc_mask) MEM_CHECK1_N(update_bias(carver, disc, -vals->disc_coeff, x_off, y_off));   /* :227-231 */
    MEM_CHECK1_N(set_rigmask(carver, rigmask, x_off, y_off));                               /* :232-233 */
    lqr_carver_set_energy_function_builtin(carver, vals->nrg_func);                         /* :234: a raw int, as the plug-in passes */
    lqr_carver_set_resize_order(carver, vals->res_order);                                   /* :235 */
    lqr_carver_set_progress(carver, progress);                                              /* :236 */
    lqr_carver_set_side_switch_frequency(carver, 2);                                        /* :237 */
    lqr_carver_set_enl_step(carver, vals->enl_step / 100);                                  /* :238 */
    if (vals->output_seams) lqr_carver_set_dump_vmaps(carver);                              /* :239-242 */
    if (vals->resize_aux_layers) {                                                          /* :243-248 */
        attach_aux_carver(carver, pres, old_width, old_height);
        attach_aux_carver(carver, disc, old_width, old_height);
        attach_aux_carver(carver, rigmask, old_width, old_height);
    }
    return carver;
}

static gboolean write_aux_carver(LqrCarverList **carver_list_p, const Layer *l, gint width, gint height)     /* render.c:902-916 */
{
    LqrCarver *aux_carver;
    LqrCarverList *carver_list = *carver_list_p;
    Layer dst;
    if (!l) return TRUE;
    aux_carver = lqr_carver_list_current(carver_list);
    dst.w = width; dst.h = height; dst.bpp = lqr_carver_get_channels(aux_carver);
    dst.px = (guchar *) calloc((size_t) width * height * dst.bpp + 1, 1);
    if (!dst.px) return FALSE;
    MEM_CHECK1(write_carver_to_layer(aux_carver, &dst));
    emit(dst.w); emit(dst.h); emit(dst.bpp);
    emit_bytes(dst.px, (size_t) dst.w * dst.h * dst.bpp);
    free(dst.px);
    *carver_list_p = lqr_carver_list_next(carver_list);
    return TRUE;
}

static gboolean render_noninteractive(LqrCarver *carver, const Layer *layer, const Layer *pres, const Layer *disc,
                                      const Layer *rigmask, const Vals *vals)
{
    gint old_width = layer->w, old_height = layer->h;
    gint new_width = vals->new_width, new_height = vals->new_height;
    LqrCarverList *carver_list;
    Layer dst;
    gint n_vmaps = 0;
    LqrRetVal ret;

    ret = lqr_carver_resize(carver, new_width, new_height);                                 /* render.c:318 */
    emit((gint) ret);
    MEM_CHECK1(ret);
    if (vals->scaleback) {                                                                  /* SCALEBACK_MODE_LQRBACK, :320-329 */
        MEM_CHECK1(lqr_carver_flatten(carver));
        new_width = old_width;
        new_height = old_height;
        MEM_CHECK1(lqr_carver_resize(carver, new_width, new_height));
    }
    if (vals->output_seams) {                                                               /* :340-346 */
        long pos = ftell(out);
        emit(0);
        MEM_CHECK1(lqr_vmap_list_foreach(lqr_vmap_list_start(carver), dump_vmap, (gpointer) &n_vmaps));
        fseek(out, pos, SEEK_SET); emit(n_vmaps); fseek(out, 0, SEEK_END);
    } else {
        emit(0);
    }
    /* the getters render_interactive reads (:547-551) */
    emit(lqr_carver_get_ref_width(carver)); emit(lqr_carver_get_ref_height(carver));
    emit(lqr_carver_get_orientation(carver)); emit(lqr_carver_get_depth(carver));
    emit(lqr_carver_get_height(carver)); emit(lqr_carver_get_channels(carver));

    dst.w = new_width; dst.h = new_height; dst.bpp = layer->bpp;
    dst.px = (guchar *) calloc((size_t) dst.w * dst.h * dst.bpp + 1, 1);
    if (!dst.px) return FALSE;
    MEM_CHECK1(write_carver_to_layer(carver, &dst));                                        /* :366 */
    emit(dst.w); emit(dst.h); emit(dst.bpp);
    emit_bytes(dst.px, (size_t) dst.w * dst.h * dst.bpp);
    free(dst.px);

    if (vals->resize_aux_layers) {                                                          /* :368-374 */
        carver_list = lqr_carver_list_start(carver);
        if (!write_aux_carver(&carver_list, pres, new_width, new_height)) return FALSE;
        if (!write_aux_carver(&carver_list, disc, new_width, new_height)) return FALSE;
        if (!write_aux_carver(&carver_list, rigmask, new_width, new_height)) return FALSE;
    }
    lqr_carver_destroy(carver);                                                             /* :376 */
    emit(n_init); emit(n_update); emit(n_end);
    return TRUE;
}

/* ---- the interactive path: one persistent carver, driven step by step (render.c:465-759) -------------------------- */
static gboolean emit_state_and_layers(LqrCarver *carver, const Layer *layer, const Layer *pres, const Layer *disc, const Layer *rigmask,
                                      const Vals *vals, gint width, gint height)
{
    LqrCarverList *carver_list;
    Layer dst;
    /* carver_data->ref_w ... ->enl_step, render.c:547-551 and :653-657 */
    emit(lqr_carver_get_ref_width(carver)); emit(lqr_carver_get_ref_height(carver));
    emit(lqr_carver_get_orientation(carver)); emit(lqr_carver_get_depth(carver));
    emit((gint) (lqr_carver_get_enl_step(carver) * 1000 + 0.5f));
    emit(lqr_carver_get_width(carver)); emit(lqr_carver_get_height(carver));
    dst.w = width; dst.h = height; dst.bpp = layer->bpp;
    dst.px = (guchar *) calloc((size_t) dst.w * dst.h * dst.bpp + 1, 1);
    if (!dst.px) return FALSE;
    MEM_CHECK1(write_carver_to_layer(carver, &dst));                                        /* :555, :661 */
    emit(dst.w); emit(dst.h); emit(dst.bpp);
    emit_bytes(dst.px, (size_t) dst.w * dst.h * dst.bpp);
    free(dst.px);
    if (vals->resize_aux_layers) {                                                          /* :557-563, :663-669 */
        carver_list = lqr_carver_list_start(carver);
        if (!write_aux_carver(&carver_list, pres, width, height)) return FALSE;
        if (!write_aux_carver(&carver_list, disc, width, height)) return FALSE;
        if (!write_aux_carver(&carver_list, rigmask, width, height)) return FALSE;
    }
    return TRUE;
}

static gboolean render_interactive(LqrCarver *carver, const Layer *layer, const Layer *pres, const Layer *disc, const Layer *rigmask,
                                   const Vals *vals, gint new_width, gint new_height)        /* render.c:465-574 */
{
    LqrRetVal ret = lqr_carver_resize(carver, new_width, new_height);                       /* :529 */
    emit((gint) ret);
    MEM_CHECK1(ret);
    return emit_state_and_layers(carver, layer, pres, disc, rigmask, vals, new_width, new_height);
}

static gboolean render_flatten(LqrCarver *carver, const Layer *layer, const Layer *pres, const Layer *disc, const Layer *rigmask,
                               const Vals *vals, gint old_width, gint old_height)            /* render.c:576-681 */
{
    LqrRetVal ret = lqr_carver_flatten(carver);                                             /* :636 */
    emit((gint) ret);
    MEM_CHECK1(ret);
    return emit_state_and_layers(carver, layer, pres, disc, rigmask, vals, old_width, old_height);
}

static gboolean render_dump_vmap(LqrCarver *carver)                                         /* render.c:683-759 */
{
    LqrVMap *vmap = lqr_vmap_dump(carver);                                                  /* :722: a map the CALLER owns */
    gint w, h, depth;
    gint *buffer;
    if (!vmap) { fprintf(stderr, "replay: lqr_vmap_dump returned NULL\n"); return FALSE; }
    w = lqr_vmap_get_width(vmap);                                                           /* write_vmap_to_layer, io_functions.c:216-219 */
    h = lqr_vmap_get_height(vmap);
    buffer = lqr_vmap_get_data(vmap);
    depth = lqr_vmap_get_depth(vmap);
    emit(w); emit(h); emit(depth); emit(lqr_vmap_get_orientation(vmap));
    fwrite(buffer, sizeof(gint), (size_t) w * h, out);
    lqr_vmap_destroy(vmap);
    return TRUE;
}

/* --steps=r120x100,d,f,...: the persistent-carver session; the layer's current size follows the steps as the plug-in's
 * drawable does */
static gboolean run_steps(LqrCarver *carver, const Layer *layer, const Layer *pres, const Layer *disc, const Layer *rigmask,
                          const Vals *vals, const char *steps)
{
    gint cur_w = layer->w, cur_h = layer->h, n = 0;
    const char *p = steps;
    long pos = ftell(out);
    emit(0);
    while (*p) {
        if (*p == 'r') {
            gint nw = 0, nh = 0;
            if (sscanf(p, "r%dx%d", &nw, &nh) != 2) { fprintf(stderr, "replay: bad step %s\n", p); return FALSE; }
            emit('r');
            if (!render_interactive(carver, layer, pres, disc, rigmask, vals, nw, nh)) return FALSE;
            cur_w = nw; cur_h = nh;
        } else if (*p == 'f') {
            emit('f');
            if (!render_flatten(carver, layer, pres, disc, rigmask, vals, cur_w, cur_h)) return FALSE;
        } else if (*p == 'd') {
            emit('d');
            if (!render_dump_vmap(carver)) return FALSE;
        } else {
            fprintf(stderr, "replay: bad step %s\n", p);
            return FALSE;
        }
        n++;
        while (*p && *p != ',') p++;
        if (*p == ',') p++;
    }
    fseek(out, pos, SEEK_SET); emit(n); fseek(out, 0, SEEK_END);
    lqr_carver_destroy(carver);
    emit(n_init); emit(n_update); emit(n_end);
    return TRUE;
}

static int read_layer(FILE *f, Layer *l, gint w, gint h, gint bpp)
{
    size_t n = (size_t) w * h * bpp;
    l->w = w; l->h = h; l->bpp = bpp;
    l->px = (guchar *) malloc(n ? n : 1);
    return l->px && fread(l->px, 1, n, f) == n;
}

int main(int argc, char **argv)
{
    gint32 hd[20];
    Layer layer, pres, disc, rig;
    Layer *ppres = NULL, *pdisc = NULL, *prig = NULL;
    Vals v;
    LqrCarver *carver;
    FILE *f;
    float fl[2];
    const char *steps = NULL, *lib = NULL, *prefix = "";
    while (argc > 1 && argv[1][0] == '-' && argv[1][1] == '-') {
        if (!strncmp(argv[1], "--steps=", 8)) steps = argv[1] + 8;
        else if (!strncmp(argv[1], "--lib=", 6)) lib = argv[1] + 6;
        else if (!strncmp(argv[1], "--prefix=", 9)) prefix = argv[1] + 9;
        else { fprintf(stderr, "replay: unknown option %s\n", argv[1]); return 2; }
        argv++; argc--;
    }
    if (argc != 3) { fprintf(stderr, "usage: %s [--lib=PATH [--prefix=P]] [--steps=...] CASE.bin OUT.bin\n", argv[0]); return 2; }
#ifdef REPLAY_DLOPEN
    if (!lib) { fprintf(stderr, "replay: this build resolves the library at run time: --lib=PATH\n"); return 2; }
    if (!load_lib(lib, prefix)) return 3;
#else
    if (lib || prefix[0]) { fprintf(stderr, "replay: --lib needs a -DREPLAY_DLOPEN build\n"); return 2; }
#endif
    f = fopen(argv[1], "rb");
    if (!f || fread(hd, sizeof(gint32), 20, f) != 20 || fread(fl, sizeof(float), 2, f) != 2) { fprintf(stderr, "replay: bad case file\n"); return 2; }
    /* header: w h bpp new_w new_h delta_x nrg_func res_order output_seams resize_aux scaleback no_disc_on_enlarge
     *         pres_coeff disc_coeff has_pres has_disc has_rig mask_bpp - -;  floats: rigidity enl_step(percent) */
    memset(&v, 0, sizeof v);
    v.new_width = hd[3]; v.new_height = hd[4]; v.delta_x = hd[5]; v.nrg_func = hd[6]; v.res_order = hd[7];
    v.output_seams = hd[8]; v.resize_aux_layers = hd[9]; v.scaleback = hd[10]; v.no_disc_on_enlarge = hd[11];
    v.pres_coeff = hd[12]; v.disc_coeff = hd[13];
    v.rigidity = fl[0]; v.enl_step = fl[1];
    if (!read_layer(f, &layer, hd[0], hd[1], hd[2])) return 2;
    if (hd[14]) { if (!read_layer(f, &pres, hd[0], hd[1], hd[17])) return 2; ppres = &pres; }
    if (hd[15]) { if (!read_layer(f, &disc, hd[0], hd[1], hd[17])) return 2; pdisc = &disc; }
    if (hd[16]) { if (!read_layer(f, &rig, hd[0], hd[1], hd[17])) return 2; prig = &rig; }
    fclose(f);
    out = fopen(argv[2], "wb+");
    if (!out) return 2;
    carver = render_init_carver(&layer, ppres, pdisc, prig, &v);
    if (!carver) { fprintf(stderr, "replay: render_init_carver failed\n"); return 1; }
    if (steps) {
        if (!run_steps(carver, &layer, ppres, pdisc, prig, &v, steps)) { fprintf(stderr, "replay: interactive session failed\n"); return 1; }
    } else if (!render_noninteractive(carver, &layer, ppres, pdisc, prig, &v)) { fprintf(stderr, "replay: render failed\n"); return 1; }
    fclose(out);
    return 0;
}
