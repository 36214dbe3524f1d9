/*
 * lqr_carver.c -- host side (plain C) of the MI355X seam-carving engine: the
 * LqrCarver C ABI of include/lqr.h, implemented on top of the device shim of
 * include/lqr_hip.h.  It is what gimp-lqr-plugin's src/render.c and
 * src/io_functions.c bind instead of liblqr-1.
 *
 * This file holds NO pixel arithmetic: it keeps the carver's geometry / level
 * bookkeeping (what liblqr keeps in LqrCarver), decides per seam whether the DP
 * map is rebuilt or band-updated (the side-switch schedule), fires the progress
 * callbacks, and serves scan lines from a packed read-back.  All planes live in
 * HBM; every stage is a kernel launched through lqrhip_*.  There is no CPU
 * fallback: without a gfx950 device lqr_carver_new() returns NULL and says why.
 *
 * Orchestration is written once over a "group" of carvers of identical geometry
 * and configuration advancing in lock-step (lqrx_carver_resize_batch); the
 * plug-in's single carver is a group of one.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

#include "../../include/lqr.h"
#include "../../include/lqr_hip.h"

#define MAXI(a, b) ((a) > (b) ? (a) : (b))
#define MINI(a, b) ((a) < (b) ? (a) : (b))

struct _LqrProgress {
    gfloat update_step;
    LqrProgressFuncInit init;
    LqrProgressFuncUpdate update;
    LqrProgressFuncEnd end;
    gchar init_width_message[LQR_PROGRESS_MAX_MESSAGE_LENGTH];
    gchar end_width_message[LQR_PROGRESS_MAX_MESSAGE_LENGTH];
    gchar init_height_message[LQR_PROGRESS_MAX_MESSAGE_LENGTH];
    gchar end_height_message[LQR_PROGRESS_MAX_MESSAGE_LENGTH];
};

struct _LqrVMap {
    gint *buffer;
    gint width, height, depth, orientation;
};
struct _LqrVMapList {
    LqrVMap *current;
    LqrVMapList *next;
};
struct _LqrCarverList {
    LqrCarver *current;
    LqrCarverList *next;
};

struct _LqrCarver {
    /* geometry, in the carver frame (transposed => frame x runs along image y) */
    int w_start, h_start, w, h, w0, h0;
    int level, max_level;
    int channels;
    int img_w, img_h;           /* the image handed to lqr_carver_new */
    int transposed;
    int active;
    LqrCarver *root;
    LqrCarverList *attached;

    int delta_x;
    float rigidity;
    float rigidity_map[2 * LQRHIP_MAX_DELTA + 1];      /* [dx + delta_x] */
    int has_bias, has_rigmask;
    int nrg_func, nrg_radius;
    int leftright, lr_switch_frequency;
    float enl_step;
    int resize_order;
    int dump_vmaps;
    LqrVMapList *flushed_vs;
    LqrProgress *progress;
    int session_update_step, session_rescale_total, session_rescale_current;

    /* device */
    LqrHipCarver *dev;
    LqrHipBatch *own_batch;     /* batch of one, created lazily (roots only) */
    int wk_valid;               /* working planes describe the carved frame */

    /* read-out cache: visible image at (w, level), carver orientation */
    guchar *ro_image;
    size_t ro_image_len;
    guchar *in_buffer;          /* the caller's pixel buffer (ownership passed at lqr_carver_new), kept as liblqr keeps it */
    size_t in_buffer_len;
    int ro_valid, ro_line;
    guchar *ro_buffer;          /* one line, the pointer scan_line hands out */
    int ro_buffer_len;

    /* debug snapshot */
    float *dbg_en, *dbg_m;
    int *dbg_least, dbg_w, dbg_h;
};

#define MAX_SUB 8
typedef struct {
    LqrCarver **r;
    int n;
    /* the group's carvers are split over nb device batches (one HIP stream each), all advancing seam by seam:
     * the latency-bound chain kernels of one sub-batch run under the bandwidth-bound carve of another */
    LqrHipBatch *b[MAX_SUB];
    int nb;
} Group;

static int g_debug_snapshot = 0;
void lqrx_set_debug(gint on) { g_debug_snapshot = on; }

/* the shim's code behind the last non-OK return: LQRHIP_EFAULT (a kernel gave up, a self-check failed) is what
 * group_build_maps can recover from */
static int g_last_rc = 0;
static LqrRetVal hip_ret(int rc)
{
    if (rc == 0) return LQR_OK;
    g_last_rc = rc;
    fprintf(stderr, "liblqr-hip: device error %d: %s\n", rc, lqrhip_last_error());
    return rc == LQRHIP_ENOMEM ? LQR_NOMEM : LQR_ERROR;
}
/* Recovery (round 6).  The plug-in tests lqr_carver_resize's return value only against LQR_NOMEM (src/render.c:42-46,318) and
 * then writes the carver to the user's layer (:366): a resize must not end in LQR_ERROR because a spin protocol timed out on a
 * busy device.  A session (one visibility-map build) that ends in LQRHIP_EFAULT is rolled back -- the base layout is as before,
 * the host's bookkeeping restored -- and carved again on the kernels without spin waits; only if that fails too does the caller
 * see LQR_ERROR, with the carver as it was before the session.  lqrhip_set_recovery(0) switches the redo off (tests). */
#define HIP_CATCH(expr) LQR_CATCH(hip_ret(expr))

/* ======================= progress ======================================== */
LqrProgress *lqr_progress_new(void)
{
    LqrProgress *p = (LqrProgress *) calloc(1, sizeof *p);
    if (!p) return NULL;
    p->update_step = 0.02f;
    strcpy(p->init_width_message, "Resizing width...");
    strcpy(p->init_height_message, "Resizing height...");
    strcpy(p->end_width_message, "done");
    strcpy(p->end_height_message, "done");
    return p;
}
LqrRetVal lqr_progress_set_init(LqrProgress *p, LqrProgressFuncInit f) { p->init = f; return LQR_OK; }
LqrRetVal lqr_progress_set_update(LqrProgress *p, LqrProgressFuncUpdate f) { p->update = f; return LQR_OK; }
LqrRetVal lqr_progress_set_end(LqrProgress *p, LqrProgressFuncEnd f) { p->end = f; return LQR_OK; }
LqrRetVal lqr_progress_set_update_step(LqrProgress *p, gfloat s) { p->update_step = s; return LQR_OK; }
static LqrRetVal copy_message(gchar *dst, const gchar *src)
{
    if (!src) return LQR_ERROR;
    snprintf(dst, LQR_PROGRESS_MAX_MESSAGE_LENGTH, "%s", src);
    return LQR_OK;
}
LqrRetVal lqr_progress_set_init_width_message(LqrProgress *p, const gchar *m) { return copy_message(p->init_width_message, m); }
LqrRetVal lqr_progress_set_init_height_message(LqrProgress *p, const gchar *m) { return copy_message(p->init_height_message, m); }
LqrRetVal lqr_progress_set_end_width_message(LqrProgress *p, const gchar *m) { return copy_message(p->end_width_message, m); }
LqrRetVal lqr_progress_set_end_height_message(LqrProgress *p, const gchar *m) { return copy_message(p->end_height_message, m); }

/* ======================= lists, vmaps ==================================== */
LqrCarverList *lqr_carver_list_start(LqrCarver *r) { return r->attached; }
LqrCarver *lqr_carver_list_current(LqrCarverList *l) { return l->current; }
LqrCarverList *lqr_carver_list_next(LqrCarverList *l) { return l->next; }
LqrVMapList *lqr_vmap_list_start(LqrCarver *r) { return r->flushed_vs; }
LqrVMap *lqr_vmap_list_current(LqrVMapList *l) { return l->current; }
LqrVMapList *lqr_vmap_list_next(LqrVMapList *l) { return l->next; }
LqrRetVal lqr_vmap_list_foreach(LqrVMapList *list, LqrVMapFunc func, gpointer data)
{
    for (; list; list = list->next) LQR_CATCH(func(list->current, data));
    return LQR_OK;
}
gint *lqr_vmap_get_data(LqrVMap *v) { return v->buffer; }
gint lqr_vmap_get_width(LqrVMap *v) { return v->width; }
gint lqr_vmap_get_height(LqrVMap *v) { return v->height; }
gint lqr_vmap_get_depth(LqrVMap *v) { return v->depth; }
gint lqr_vmap_get_orientation(LqrVMap *v) { return v->orientation; }
void lqr_vmap_destroy(LqrVMap *v) { if (v) { free(v->buffer); free(v); } }

/* ======================= lifecycle ======================================= */
LqrCarver *lqr_carver_new(guchar *buffer, gint width, gint height, gint channels)
{
    LqrCarver *r;
    if (!buffer || width < 1 || height < 1 || channels < 1 || channels > 4) return NULL;
    r = (LqrCarver *) calloc(1, sizeof *r);
    if (!r) return NULL;
    r->dev = lqrhip_carver_create(buffer, width, height, channels);
    if (!r->dev) {
        fprintf(stderr, "liblqr-hip: lqr_carver_new failed: %s\n", lqrhip_last_error());
        free(r);
        return NULL;
    }
    /* ownership passed to the carver (render.c:220-223).  The pixels now live in HBM; the block is kept, as liblqr keeps it,
     * and becomes the read-out buffer: handing 33 MB back to the C library and asking for 31 MB again costs an munmap, an
     * mmap and a page fault per 4 KiB -- more than the transfer itself (measured: 64 x 4K uploads 207 ms with the free, 62 without) */
    r->in_buffer = buffer;
    r->in_buffer_len = (size_t) width * height * channels;
    r->level = r->max_level = 1;
    r->delta_x = 1;
    r->w = r->w0 = r->w_start = width;
    r->h = r->h0 = r->h_start = height;
    r->channels = channels;
    r->img_w = width; r->img_h = height;
    r->nrg_func = LQR_EF_GRAD_XABS;
    r->nrg_radius = 1;
    r->enl_step = 2.0f;
    r->resize_order = LQR_RES_ORDER_HOR;
    r->progress = lqr_progress_new();
    if (!r->progress) { lqrhip_carver_destroy(r->dev); free(r); return NULL; }
    return r;
}

LqrRetVal lqr_carver_init(LqrCarver *r, gint delta_x, gfloat rigidity)
{
    int x;
    if (r->active || delta_x < 0 || delta_x > LQRHIP_MAX_DELTA) return LQR_ERROR;
    r->delta_x = delta_x;
    r->rigidity = rigidity;
    /* rigidity bias ~ |dx|^1.5 summed along the seam (help/en/index.wiki:83); the
     * table is tiny and is computed on the host so that it is bit-identical to C */
    for (x = -delta_x; x <= delta_x; x++)
        r->rigidity_map[x + delta_x] = r->rigidity * powf(fabsf((float) x), 1.5f) / r->h;
    HIP_CATCH(lqrhip_carver_activate(r->dev));
    r->active = 1;
    return LQR_OK;
}

static void carver_free_host(LqrCarver *r)
{
    LqrVMapList *v, *vn;
    for (v = r->flushed_vs; v; v = vn) { vn = v->next; lqr_vmap_destroy(v->current); free(v); }
    free(r->progress);
    free(r->ro_image); free(r->ro_buffer); free(r->in_buffer);
    free(r->dbg_en); free(r->dbg_m); free(r->dbg_least);
    free(r);
}

void lqr_carver_destroy(LqrCarver *r)
{
    LqrCarverList *l, *ln;
    if (!r) return;
    if (r->own_batch) lqrhip_batch_destroy(r->own_batch);
    for (l = r->attached; l; l = ln) {
        ln = l->next;
        lqr_carver_destroy(l->current);
        free(l);
    }
    lqrhip_carver_destroy(r->dev);
    carver_free_host(r);
}

LqrRetVal lqr_carver_attach(LqrCarver *r, LqrCarver *aux)
{
    LqrCarverList *n, *p;
    if (r->w0 != aux->w0 || r->h0 != aux->h0) return LQR_ERROR;
    n = (LqrCarverList *) calloc(1, sizeof *n);
    if (!n) return LQR_NOMEM;
    n->current = aux;
    if (!r->attached) r->attached = n;
    else { for (p = r->attached; p->next; p = p->next); p->next = n; }
    aux->root = r;
    HIP_CATCH(lqrhip_carver_attach(r->dev, aux->dev));
    return LQR_OK;
}

/* ======================= configuration =================================== */
LqrRetVal lqr_carver_set_energy_function_builtin(LqrCarver *r, LqrEnergyFuncBuiltinType ef)
{
    if ((int) ef < LQR_EF_GRAD_NORM || (int) ef > LQR_EF_NULL) return LQR_ERROR;
    r->nrg_func = (int) ef;
    r->nrg_radius = (ef == LQR_EF_NULL) ? 0 : 1;
    return LQR_OK;
}
void lqr_carver_set_resize_order(LqrCarver *r, LqrResizeOrder o) { r->resize_order = (int) o; }
void lqr_carver_set_progress(LqrCarver *r, LqrProgress *p) { free(r->progress); r->progress = p; }
void lqr_carver_set_side_switch_frequency(LqrCarver *r, guint f) { r->lr_switch_frequency = (int) f; }
LqrRetVal lqr_carver_set_enl_step(LqrCarver *r, gfloat s)
{
    if (!(s > 1 && s <= 2)) return LQR_ERROR;
    r->enl_step = s;
    return LQR_OK;
}
gfloat lqr_carver_get_enl_step(LqrCarver *r) { return r->enl_step; }
void lqr_carver_set_dump_vmaps(LqrCarver *r) { r->dump_vmaps = 1; }

/* ======================= getters ========================================= */
gint lqr_carver_get_width(LqrCarver *r) { return r->transposed ? r->h : r->w; }
gint lqr_carver_get_height(LqrCarver *r) { return r->transposed ? r->w : r->h; }
gint lqr_carver_get_channels(LqrCarver *r) { return r->channels; }
gint lqr_carver_get_ref_width(LqrCarver *r) { return r->transposed ? r->h_start : r->w_start; }
gint lqr_carver_get_ref_height(LqrCarver *r) { return r->transposed ? r->w_start : r->h_start; }
gint lqr_carver_get_orientation(LqrCarver *r) { return r->transposed ? 1 : 0; }
gint lqr_carver_get_depth(LqrCarver *r) { return r->w0 - r->w_start; }
gint lqrx_carver_frame_width(LqrCarver *r) { return r->w; }
gint lqrx_carver_frame_height(LqrCarver *r) { return r->h; }

/* ======================= group helpers =================================== */
static void set_width_one(LqrCarver *r, int w1)
{
    r->w = w1;
    r->level = r->w0 - w1 + 1;
    r->ro_valid = 0;
    r->ro_line = 0;
}
static void set_width_tree(LqrCarver *r, int w1)
{
    LqrCarverList *l;
    set_width_one(r, w1);
    for (l = r->attached; l; l = l->next) set_width_tree(l->current, w1);
}

/* delta_x = 2 .. 10 (the whole range of the plug-in's dialog, src/interface.c:47), or a rigidity mask that matters: the carver runs
 * on the tiled kernels' general instantiations, which need the whole group co-resident on ONE stream */
static int is_general(const LqrCarver *c)
{
    return (c->delta_x >= 2 && c->delta_x <= 10) || (c->delta_x == 1 && c->has_rigmask && c->rigidity != 0);
}

static LqrRetVal group_open(Group *g, LqrCarver **rs, int n)
{
    int i, k;
    g->r = rs; g->n = n; g->nb = 0;
    for (k = 0; k < MAX_SUB; k++) g->b[k] = NULL;
    if (n == 1) {
        if (!rs[0]->own_batch) {
            LqrHipCarver *d = rs[0]->dev;
            rs[0]->own_batch = lqrhip_batch_create(&d, 1);
            if (!rs[0]->own_batch) return LQR_NOMEM;
        }
        g->b[0] = rs[0]->own_batch;
        g->nb = 1;
    } else {
        LqrHipCarver **ds = (LqrHipCarver **) malloc((size_t) n * sizeof *ds);
        /* sub-batch streams are for the plain kernels: a shared batch never runs the persistent tiled kernels, and a general
         * group of 32 or more (small images: the limit of lqrx_carver_resize_batch is in tiles, not images) would fall to the
         * one-wave-per-image kernels, ~30x slower */
        int nb = is_general(rs[0]) ? 1 : lqrhip_sub_batches(n);
        if (nb < 1) nb = 1;
        if (nb > MAX_SUB) nb = MAX_SUB;
        if (nb > n) nb = n;
        if (!ds) return LQR_NOMEM;
        for (i = 0; i < n; i++) {
            if (rs[i]->own_batch) { lqrhip_batch_destroy(rs[i]->own_batch); rs[i]->own_batch = NULL; }
            ds[i] = rs[i]->dev;
        }
        for (k = 0; k < nb; k++) {
            int lo = (int) ((long long) n * k / nb), hi = (int) ((long long) n * (k + 1) / nb);
            g->b[k] = lqrhip_batch_create(ds + lo, hi - lo);
            if (!g->b[k]) {     /* give the sub-batches (and their streams) created so far back */
                int q;
                for (q = 0; q < k; q++) { lqrhip_batch_destroy(g->b[q]); g->b[q] = NULL; }
                g->nb = 0;
                free(ds);
                return LQR_NOMEM;
            }
            lqrhip_batch_set_shared(g->b[k], nb > 1 ? nb : 0);       /* how many sub-batches run side by side */
            g->nb = k + 1;
        }
        free(ds);
    }
    return LQR_OK;
}
static void group_close(Group *g)
{
    int k;
    if (g->n > 1)
        for (k = 0; k < g->nb; k++) lqrhip_batch_destroy(g->b[k]);
    g->nb = 0;
}
/* the same device call on every sub-batch */
#define HIP_ALL(g, call)                                                   \
    do {                                                                   \
        int k__;                                                           \
        for (k__ = 0; k__ < (g)->nb; k__++) {                              \
            LqrHipBatch *B = (g)->b[k__];                                  \
            HIP_CATCH(call);                                               \
        }                                                                  \
    } while (0)

#define FOR_TREE(g, i, r, body)                                   \
    for (i = 0; i < (g)->n; i++) {                                \
        LqrCarverList *l__;                                       \
        LqrCarver *r = (g)->r[i];                                 \
        body;                                                     \
        for (l__ = (g)->r[i]->attached; l__; l__ = l__->next) {   \
            r = l__->current;                                     \
            body;                                                 \
        }                                                         \
    }

static void dp_params(const LqrCarver *r, LqrHipDpParams *p)
{
    memset(p, 0, sizeof *p);
    p->delta_x = r->delta_x;
    p->use_rigidity = r->rigidity != 0;
    memcpy(p->rigidity_map, r->rigidity_map, sizeof p->rigidity_map);
    p->nrg_func = r->nrg_func;
    p->nrg_radius = r->nrg_radius;
    p->w_start = r->w_start;
}

/* ======================= flatten / transpose (E11) ======================= */
static LqrRetVal group_flatten(Group *g)
{
    LqrCarver *r0 = g->r[0];
    int i;
    /* a carver that is already flat (nothing hidden, nothing inserted) is its own flattening */
    if (!(r0->w == r0->w0 && r0->level == 1 && r0->max_level == 1 && r0->w_start == r0->w0))
    {
        HIP_ALL(g, lqrhip_flatten(B, r0->w0, r0->h0, r0->w, r0->level));     /* every sub-batch staged ... */
        HIP_ALL(g, lqrhip_planes_commit(B));                                 /* ... before any adopts its flat layout */
    } else if (r0->wk_valid)
        return LQR_OK;
    FOR_TREE(g, i, r, {
        r->w0 = r->w; r->h0 = r->h;
        r->w_start = r->w; r->h_start = r->h;
        r->level = 1; r->max_level = 1;
        r->wk_valid = 0; r->ro_valid = 0; r->ro_line = 0;
    });
    return LQR_OK;
}

static LqrRetVal group_transpose(Group *g)
{
    LqrCarver *r0 = g->r[0];
    int i, x, d;
    if (r0->level > 1 || r0->max_level > 1 || r0->w0 != r0->w) LQR_CATCH(group_flatten(g));
    HIP_ALL(g, lqrhip_transpose(B, r0->w0, r0->h0));
    HIP_ALL(g, lqrhip_planes_commit(B));
    FOR_TREE(g, i, r, {
        d = r->w0; r->w0 = r->h0; r->h0 = d;
        r->w = r->w0; r->h = r->h0;
        r->w_start = r->w0; r->h_start = r->h0;
        r->level = 1; r->max_level = 1;
        if (r->active)      /* the rigidity table is rescaled, not recomputed */
            for (x = -r->delta_x; x <= r->delta_x; x++)
                r->rigidity_map[x + r->delta_x] = r->rigidity_map[x + r->delta_x] * r->w0 / r->h0;
        r->transposed = r->transposed ? 0 : 1;
        r->wk_valid = 0; r->ro_valid = 0; r->ro_line = 0;
    });
    return LQR_OK;
}

LqrRetVal lqr_carver_flatten(LqrCarver *r)
{
    Group g;
    LqrRetVal ret;
    if (r->root) return LQR_ERROR;
    LQR_CATCH(group_open(&g, &r, 1));
    ret = group_flatten(&g);
    group_close(&g);
    return ret;
}

/* ======================= masks (E2) ====================================== */
static LqrRetVal mask_prepare(LqrCarver *r)
{
    if (!r->active || r->root) return LQR_ERROR;
    if (r->w != r->w0 || r->w_start != r->w0 || r->h != r->h0 || r->h_start != r->h0) LQR_CATCH(lqr_carver_flatten(r));
    return LQR_OK;
}

LqrRetVal lqr_carver_bias_add_rgb_area(LqrCarver *r, guchar *rgb, gint bias_factor, gint channels, gint width, gint height,
                                       gint x_off, gint y_off)
{
    LQR_CATCH(mask_prepare(r));
    if (bias_factor == 0) return LQR_OK;
    HIP_CATCH(lqrhip_mask_add(r->dev, rgb, channels, width, height, x_off, y_off, r->transposed, 0, bias_factor));
    r->has_bias = 1;
    r->wk_valid = 0;
    return LQR_OK;
}

LqrRetVal lqr_carver_rigmask_add_rgb_area(LqrCarver *r, guchar *rgb, gint channels, gint width, gint height, gint x_off,
                                          gint y_off)
{
    LQR_CATCH(mask_prepare(r));
    HIP_CATCH(lqrhip_mask_add(r->dev, rgb, channels, width, height, x_off, y_off, r->transposed, 1, 0));
    r->has_rigmask = 1;
    r->wk_valid = 0;
    return LQR_OK;
}

/* ======================= per-seam loop (E10) ============================= */
static LqrRetVal take_debug_snapshot(LqrCarver *r)
{
    size_t n = (size_t) r->w * r->h;
    free(r->dbg_en); free(r->dbg_m); free(r->dbg_least);
    r->dbg_w = r->w; r->dbg_h = r->h;
    r->dbg_en = (float *) malloc(n * sizeof(float));
    r->dbg_m = (float *) malloc(n * sizeof(float));
    r->dbg_least = (int *) malloc(n * sizeof(int));
    if (!r->dbg_en || !r->dbg_m || !r->dbg_least) return LQR_NOMEM;
    HIP_CATCH(lqrhip_read_working(r->dev, r->w, r->h, r->dbg_en, r->dbg_m, r->dbg_least));
    return LQR_OK;
}

static LqrRetVal group_build_vsmap(Group *g, int depth, int *reported)
{
    LqrCarver *r0 = g->r[0];
    LqrHipDpParams p;
    int l, i, lr_switch_interval = 0, n_seams, wc0, first_level, finish = 0, w1;
    if (depth == 0) depth = r0->w_start + 1;
    n_seams = depth - r0->max_level;
    wc0 = r0->w;                         /* = w_start - max_level + 1, the carved frame */
    first_level = 2 * r0->max_level - 1; /* seam l is stored as level l + max_level - 1 */
    /* "frequency" = number of side switches per rescale operation */
    if (r0->lr_switch_frequency) lr_switch_interval = (depth - r0->max_level - 1) / r0->lr_switch_frequency + 1;
    dp_params(r0, &p);
    HIP_ALL(g, lqrhip_seam_log_reserve(B, n_seams, r0->h));

    for (l = r0->max_level; l < depth; l++) {
        int full = 0, lr_pick = r0->leftright, w_before = r0->w;
        if ((l - r0->max_level + r0->session_rescale_current) % r0->session_update_step == 0 && r0->progress &&
            r0->progress->update) {
            /* report completed work, not enqueued work (a session that is being redone does not report twice) */
            HIP_ALL(g, lqrhip_batch_sync(B));
            if (l > *reported) {
                r0->progress->update((gdouble) (l - r0->max_level + r0->session_rescale_current) /
                                     (gdouble) r0->session_rescale_total);
                *reported = l;
            }
        }
        if (w_before - 1 > 1) {
            if (r0->lr_switch_frequency && ((l - r0->max_level + lr_switch_interval / 2) % lr_switch_interval) == 0) {
                for (i = 0; i < g->n; i++) g->r[i]->leftright ^= 1;
                full = 1;
            }
        } else {
            finish = 1;
        }
        HIP_ALL(g, lqrhip_seam_step(B, &p, w_before, r0->h, l - r0->max_level, lr_pick, full, r0->leftright));
        for (i = 0; i < g->n; i++) { g->r[i]->level++; g->r[i]->w--; }
    }

    if (g_debug_snapshot)
        for (i = 0; i < g->n; i++) LQR_CATCH(take_debug_snapshot(g->r[i]));

    /* the seam loop is over: nothing of it is committed to the base layout before its kernels have all ended without a device-side
     * failure and the seam log has passed its self-check */
    HIP_ALL(g, lqrhip_session_check(B, r0->h, wc0, n_seams, r0->delta_x));
    HIP_ALL(g, lqrhip_batch_sync(B));
    HIP_ALL(g, lqrhip_vs_commit(B, r0->w0, r0->h0, wc0, n_seams, first_level, finish));
    /* inflate (E14): every seam of this session is doubled in the base layout */
    HIP_ALL(g, lqrhip_inflate(B, r0->w0, r0->h0, depth - 1, r0->max_level));        /* every sub-batch staged and checked ... */
    HIP_ALL(g, lqrhip_planes_commit(B));                                            /* ... before any adopts its inflated layout */
    w1 = r0->w0 + (depth - 1) - r0->max_level + 1;
    FOR_TREE(g, i, r, {
        r->level = depth; r->max_level = depth;
        r->w0 = w1;
        set_width_one(r, r->w_start);
    });
    return LQR_OK;
}

/* one session: (re)lay the working planes out if needed, energy, DP, the seam loop, commit, inflate */
static LqrRetVal session_run(Group *g, int depth, int redo, int *reported)
{
    LqrCarver *r0 = g->r[0];
    LqrHipDpParams p;
    int i;
    for (i = 0; i < g->n; i++) set_width_one(g->r[i], g->r[i]->w_start - g->r[i]->max_level + 1);    /* the carved frame */
    if (!r0->wk_valid || redo) {
        /* a flat carver: identity; a multi-size image (a redone deeper session, working planes lost in a failed one): the
         * pixels of the base layout without a level, in order */
        const int from_visible = (r0->max_level != 1 || r0->w0 != r0->w_start);
        HIP_ALL(g, lqrhip_wk_init(B, from_visible));
        for (i = 0; i < g->n; i++) g->r[i]->wk_valid = 1;
    }
    dp_params(r0, &p);
    HIP_ALL(g, lqrhip_emap_build(B, &p, r0->w, r0->h));
    HIP_ALL(g, lqrhip_mmap_build(B, &p, r0->w, r0->h, r0->leftright));
    return group_build_vsmap(g, depth, reported);
}

static LqrRetVal group_build_maps(Group *g, int depth)
{
    LqrCarver *r0 = g->r[0];
    LqrRetVal ret;
    int i, k, reported = -1, *snap;
    if (depth <= r0->max_level) return LQR_OK;
    if (!r0->active || r0->root) return LQR_ERROR;
    /* what a session changes on the host before it succeeds: the roots' width, level and side */
    snap = (int *) malloc((size_t) g->n * 3 * sizeof *snap);
    if (!snap) return LQR_NOMEM;
    for (i = 0; i < g->n; i++) { snap[3 * i] = g->r[i]->w; snap[3 * i + 1] = g->r[i]->level; snap[3 * i + 2] = g->r[i]->leftright; }
    g_last_rc = 0;
    ret = session_run(g, depth, 0, &reported);
    for (k = 0; k < 2 && ret != LQR_OK; k++) {
        const int fault = (ret == LQR_ERROR && g_last_rc == LQRHIP_EFAULT);
        const int wc0 = r0->w_start - r0->max_level + 1, n_seams = (depth ? depth : r0->w_start + 1) - r0->max_level;
        /* the carver as it was before the session: the base layout (levels of this session removed if they were committed),
         * the bookkeeping; the working planes are gone */
        /* (also after an allocation failure: the staging of the inflate pass comes after the commit of the session's levels) */
        if (fault || g_last_rc == LQRHIP_ENOMEM)
            for (i = 0; i < g->nb; i++) (void) lqrhip_session_rollback(g->b[i], r0->w0, r0->h0, 2 * r0->max_level - 1, wc0 - n_seams <= 1);
        for (i = 0; i < g->n; i++) {
            LqrCarver *c = g->r[i];
            c->w = snap[3 * i]; c->level = snap[3 * i + 1]; c->leftright = snap[3 * i + 2];
            c->wk_valid = 0; c->ro_valid = 0; c->ro_line = 0;
        }
        if (!fault || !lqrhip_get_recovery() || k == 1) break;
        fprintf(stderr, "liblqr-hip: the session is carved again on the kernels without spin waits\n");
        for (i = 0; i < g->nb; i++) lqrhip_batch_set_safe(g->b[i], 1);
        g_last_rc = 0;
        ret = session_run(g, depth, 1, &reported);
        for (i = 0; i < g->nb; i++) lqrhip_batch_set_safe(g->b[i], 0);
    }
    free(snap);
    return ret;
}

/* ======================= vmaps (E12) ===================================== */
LqrVMap *lqr_vmap_dump(LqrCarver *r)
{
    LqrVMap *v;
    int w = r->w_start, h = r->h, depth = r->w0 - r->w_start, x, y;
    int *frame, *buffer;
    frame = (int *) malloc((size_t) w * h * sizeof(int));
    v = (LqrVMap *) calloc(1, sizeof *v);
    if (!frame || !v) { free(frame); free(v); return NULL; }
    if (lqrhip_read_vmap(r->dev, r->w0, r->h0, w, r->w0 - w + 1, depth, frame) != 0) {
        fprintf(stderr, "liblqr-hip: vmap read-back failed: %s\n", lqrhip_last_error());
        free(frame); free(v);
        return NULL;
    }
    if (r->transposed) {    /* hand the map back in image orientation */
        buffer = (int *) malloc((size_t) w * h * sizeof(int));
        if (!buffer) { free(frame); free(v); return NULL; }
        for (y = 0; y < h; y++)
            for (x = 0; x < w; x++) buffer[(size_t) x * h + y] = frame[(size_t) y * w + x];
        free(frame);
    } else {
        buffer = frame;
    }
    v->buffer = buffer;
    v->width = lqr_carver_get_ref_width(r);
    v->height = lqr_carver_get_ref_height(r);
    v->depth = depth;
    v->orientation = r->transposed;
    return v;
}

static LqrRetVal vmap_internal_dump(LqrCarver *r)
{
    LqrVMap *v = lqr_vmap_dump(r);
    LqrVMapList *n = (LqrVMapList *) calloc(1, sizeof *n), *p;
    if (!v || !n) return LQR_NOMEM;
    n->current = v;
    if (!r->flushed_vs) r->flushed_vs = n;
    else { for (p = r->flushed_vs; p->next; p = p->next); p->next = n; }
    return LQR_OK;
}

/* ======================= resize (E10) ==================================== */
static LqrRetVal group_resize_dir(Group *g, int w1, int want_transposed)
{
    LqrCarver *r = g->r[0];
    int delta, gamma, delta_max, i;
    const gchar *init_msg = want_transposed ? r->progress->init_height_message : r->progress->init_width_message;
    const gchar *end_msg = want_transposed ? r->progress->end_height_message : r->progress->end_width_message;

    if (r->transposed == want_transposed) {
        delta = w1 - r->w_start; gamma = w1 - r->w;
        delta_max = (int) ((r->enl_step - 1) * r->w_start) - 1;
    } else {
        delta = w1 - r->h_start; gamma = w1 - r->h;
        delta_max = (int) ((r->enl_step - 1) * r->h_start) - 1;
    }
    if (delta_max < 1) delta_max = 1;
    if (delta < 0) { delta = -delta; delta_max = delta; }

    for (i = 0; i < g->n; i++) {
        LqrCarver *c = g->r[i];
        c->session_rescale_total = gamma > 0 ? gamma : -gamma;
        c->session_rescale_current = 0;
        c->session_update_step = (int) MAXI(c->session_rescale_total * c->progress->update_step, 1);
    }
    if (r->session_rescale_total && r->progress->init) r->progress->init(init_msg);

    while (gamma) {
        int delta0 = MINI(delta, delta_max), new_w;
        delta -= delta0;
        if (r->transposed != want_transposed) LQR_CATCH(group_transpose(g));
        new_w = MINI(w1, r->w_start + delta_max);
        gamma = w1 - new_w;
        LQR_CATCH(group_build_maps(g, delta0 + 1));
        for (i = 0; i < g->n; i++) {
            set_width_tree(g->r[i], new_w);
            g->r[i]->session_rescale_current = g->r[i]->session_rescale_total - (gamma > 0 ? gamma : -gamma);
            if (g->r[i]->dump_vmaps) LQR_CATCH(vmap_internal_dump(g->r[i]));
        }
        if (new_w < w1) {
            LQR_CATCH(group_flatten(g));
            delta_max = (int) ((r->enl_step - 1) * r->w_start) - 1;
            if (delta_max < 1) delta_max = 1;
        }
    }
    if (r->session_rescale_total && r->progress->end) {
        HIP_ALL(g, lqrhip_batch_sync(B));
        r->progress->end(end_msg);
    }
    return LQR_OK;
}

static int same_config(const LqrCarver *a, const LqrCarver *b)
{
    LqrCarverList *la = a->attached, *lb = b->attached;
    if (a->w0 != b->w0 || a->h0 != b->h0 || a->w != b->w || a->h != b->h || a->w_start != b->w_start ||
        a->h_start != b->h_start || a->level != b->level || a->max_level != b->max_level || a->channels != b->channels ||
        a->transposed != b->transposed || a->active != b->active || a->delta_x != b->delta_x || a->rigidity != b->rigidity ||
        a->has_bias != b->has_bias || a->has_rigmask != b->has_rigmask || a->nrg_func != b->nrg_func ||
        a->leftright != b->leftright || a->lr_switch_frequency != b->lr_switch_frequency || a->enl_step != b->enl_step ||
        a->resize_order != b->resize_order || a->wk_valid != b->wk_valid)
        return 0;
    for (; la && lb; la = la->next, lb = lb->next)
        if (la->current->channels != lb->current->channels) return 0;
    return !la && !lb;
}

static LqrRetVal group_resize(LqrCarver **rs, int n, int w1, int h1)
{
    Group g;
    LqrRetVal ret = LQR_OK;
    int i;
    if (w1 < 1 || h1 < 1) return LQR_ERROR;
    for (i = 0; i < n; i++)
        if (rs[i]->root || !rs[i]->progress) return LQR_ERROR;
    LQR_CATCH(group_open(&g, rs, n));
    if (rs[0]->resize_order == LQR_RES_ORDER_HOR) {
        if ((ret = group_resize_dir(&g, w1, 0)) == LQR_OK) ret = group_resize_dir(&g, h1, 1);
    } else {
        if ((ret = group_resize_dir(&g, h1, 1)) == LQR_OK) ret = group_resize_dir(&g, w1, 0);
    }
    if (ret == LQR_OK) {        /* every sub-batch is synchronised, whatever the others returned */
        int k;
        for (k = 0; k < g.nb; k++) { LqrRetVal rk = hip_ret(lqrhip_batch_sync(g.b[k])); if (ret == LQR_OK) ret = rk; }
    }
    if (ret != LQR_OK) { int k; for (k = 0; k < g.nb; k++) lqrhip_batch_abort(g.b[k]); }      /* nothing of this resize may surface in the next one */
    FOR_TREE(&g, i, r, { r->ro_valid = 0; r->ro_line = 0; });
    group_close(&g);
    return ret;
}

LqrRetVal lqr_carver_resize(LqrCarver *r, gint w1, gint h1) { return group_resize(&r, 1, w1, h1); }

LqrRetVal lqrx_carver_resize_batch(LqrCarver **carvers, gint n, gint w1, gint h1)
{
    int i, same = 1;
    if (n < 1) return LQR_ERROR;
    for (i = 1; i < n; i++) same &= same_config(carvers[0], carvers[i]);
    if (same) {
        /* delta_x = 2 .. 4 and rigidity masks (with rigidity) run on the tiled kernels only while the whole group's tiles are
         * co-resident; a larger group would fall to the one-wave-per-image kernels (~30x slower).  The images are
         * independent, so such a group is carved in consecutive sub-groups that fit. */
        const LqrCarver *c = carvers[0];
        if (is_general(c)) {
            int wmax = c->w_start > c->h_start ? c->w_start : c->h_start, lim;   /* either direction may be carved, shrinking or enlarging */
            if (w1 > wmax) wmax = w1;
            if (h1 > wmax) wmax = h1;
            lim = lqrhip_general_batch_limit_delta(wmax, c->delta_x);
            if (lim >= 1 && n > lim) {
                /* every sub-group is carved whatever the others returned (the images are independent); the first error is reported */
                LqrRetVal first = LQR_OK;
                for (i = 0; i < n; i += lim) {
                    const LqrRetVal r = group_resize(carvers + i, n - i < lim ? n - i : lim, w1, h1);
                    if (first == LQR_OK) first = r;
                }
                return first;
            }
        }
        return group_resize(carvers, n, w1, h1);
    }
    for (i = 0; i < n; i++) LQR_CATCH(lqr_carver_resize(carvers[i], w1, h1));     /* heterogeneous: one by one */
    return LQR_OK;
}

/* ======================= readout (E12) =================================== */
static LqrRetVal fetch_visible(LqrCarver *r)
{
    size_t n = (size_t) r->w * r->h * r->channels;
    if (r->ro_valid) return LQR_OK;
    if (!r->ro_image && r->in_buffer && r->in_buffer_len >= n) {     /* the block the image arrived in */
        r->ro_image = r->in_buffer; r->ro_image_len = r->in_buffer_len;
        r->in_buffer = NULL;
    }
    if (!r->ro_image || r->ro_image_len < n) {       /* kept across read-outs: a fresh block costs a page fault per 4 KiB */
        free(r->ro_image);
        r->ro_image = (guchar *) malloc(n ? n : 1);
        if (!r->ro_image) { r->ro_image_len = 0; return LQR_NOMEM; }
        r->ro_image_len = n;
    }
    HIP_CATCH(lqrhip_read_visible(r->dev, r->w0, r->h0, r->w, r->level, r->ro_image));
    if (r->ro_buffer_len < r->w * r->channels) {
        free(r->ro_buffer);
        r->ro_buffer = (guchar *) malloc((size_t) r->w * r->channels);
        if (!r->ro_buffer) return LQR_NOMEM;
        r->ro_buffer_len = r->w * r->channels;
    }
    r->ro_valid = 1;
    return LQR_OK;
}

void lqr_carver_scan_reset(LqrCarver *r) { r->ro_line = 0; }
gboolean lqr_carver_scan_by_row(LqrCarver *r) { return r->transposed ? FALSE : TRUE; }

/* One packed device->host transfer feeds the whole scan (io_functions.c:155-164
 * calls this once per line); a line is a carver-frame row, i.e. an image column
 * when the carver is transposed. */
gboolean lqr_carver_scan_line(LqrCarver *r, gint *n, guchar **rgb)
{
    if (r->ro_line >= r->h) { r->ro_line = 0; return FALSE; }
    if (fetch_visible(r) != LQR_OK) return FALSE;
    memcpy(r->ro_buffer, r->ro_image + (size_t) r->ro_line * r->w * r->channels, (size_t) r->w * r->channels);
    *n = r->ro_line;
    *rgb = r->ro_buffer;
    r->ro_line++;
    return TRUE;
}

LqrRetVal lqrx_carver_read_image(LqrCarver *r, guchar *out)
{
    int x, y, ch = r->channels;
    if (!r->transposed && !r->ro_valid) {      /* straight into the caller's buffer: no second host copy */
        HIP_CATCH(lqrhip_read_visible(r->dev, r->w0, r->h0, r->w, r->level, out));
        return LQR_OK;
    }
    LQR_CATCH(fetch_visible(r));
    if (!r->transposed) {
        memcpy(out, r->ro_image, (size_t) r->w * r->h * ch);
    } else {
        for (y = 0; y < r->h; y++)
            for (x = 0; x < r->w; x++) memcpy(out + ((size_t) x * r->h + y) * ch, r->ro_image + ((size_t) y * r->w + x) * ch, ch);
    }
    return LQR_OK;
}

LqrRetVal lqrx_carver_read_image_device(LqrCarver *r, void *device_ptr)
{
    HIP_CATCH(lqrhip_read_visible_device(r->dev, r->w0, r->h0, r->w, r->level, device_ptr));
    return LQR_OK;
}

/* ======================= seam-map colour ramp (I5) ======================== */
LqrRetVal lqrx_vmap_to_rgba(LqrVMap *v, const gdouble col_start[3], const gdouble col_end[3], guchar *out_rgba)
{
    if (!v || !v->buffer || !out_rgba) return LQR_ERROR;
    HIP_CATCH(lqrhip_vmap_to_rgba(v->buffer, v->width, v->height, v->depth, col_start, col_end, out_rgba));
    return LQR_OK;
}

/* ======================= reload from device memory ======================= */
LqrRetVal lqrx_carver_reload_device_batch(LqrCarver **rs, gint n, void *const *device_rgb)
{
    int i, x;
    if (n < 1) return LQR_ERROR;
    for (i = 0; i < n; i++)
        if (!rs[i] || rs[i]->root || rs[i]->attached || !device_rgb[i]) return LQR_ERROR;
    for (i = 0; i < n; i++) {
        LqrCarver *r = rs[i];
        LqrVMapList *v, *vn;
        HIP_CATCH(lqrhip_carver_reset(r->dev, device_rgb[i], r->img_w, r->img_h));
        for (v = r->flushed_vs; v; v = vn) { vn = v->next; lqr_vmap_destroy(v->current); free(v); }
        r->flushed_vs = NULL;
        r->level = r->max_level = 1;
        r->w = r->w0 = r->w_start = r->img_w;
        r->h = r->h0 = r->h_start = r->img_h;
        r->transposed = 0;
        r->leftright = 0;
        r->has_bias = r->has_rigmask = 0;
        r->wk_valid = 0;
        r->ro_valid = 0; r->ro_line = 0;
        if (r->active)      /* as lqr_carver_init */
            for (x = -r->delta_x; x <= r->delta_x; x++)
                r->rigidity_map[x + r->delta_x] = r->rigidity * powf(fabsf((float) x), 1.5f) / r->h;
    }
    HIP_CATCH(lqrhip_reset_sync());
    return LQR_OK;
}

/* ======================= auto-size (plug-in side) ======================== */
/* guess_new_size, reference src/layers_combo.c:275-392: the geometry is resolved here, the
 * per-line counting and the max over lines run on the device */
gint lqrx_guess_new_size(const guchar *mask, gint channels, gint width, gint height, gint x_off, gint y_off,
                         gint old_width, gint old_height, gint direction)
{
    int lw = MINI(old_width, width + x_off) - MAXI(0, x_off);
    int lh = MINI(old_height, height + y_off) - MAXI(0, y_off);
    int old_size = direction ? old_height : old_width, m;
    if (direction == 0)     /* lines are mask rows z1 - y_off, z1 in [max(0,y_off), min(old_h, h+y_off)) */
        m = lqrhip_mask_line_max(mask, channels, width, height, MAXI(0, y_off) - y_off, MAXI(0, -x_off),
                                 MINI(old_height, height + y_off) - MAXI(0, y_off), lw, 0);
    else
        m = lqrhip_mask_line_max(mask, channels, width, height, MAXI(0, x_off) - x_off, MAXI(0, -y_off),
                                 MINI(old_width, width + x_off) - MAXI(0, x_off), lh, 1);
    if (m < 0) { fprintf(stderr, "liblqr-hip: guess_new_size failed: %s\n", lqrhip_last_error()); return old_size; }
    return old_size - m;
}

/* ======================= test hooks ====================================== */
LqrRetVal lqrx_carver_get_energy(LqrCarver *r, gfloat *buffer)
{
    Group g;
    LqrHipDpParams p;
    LqrRetVal ret = LQR_OK;
    if (r->root) return LQR_ERROR;
    LQR_CATCH(group_open(&g, &r, 1));
    if (r->w != r->w_start - r->max_level + 1) ret = group_flatten(&g);
    if (ret == LQR_OK && !r->wk_valid) {
        if ((ret = hip_ret(lqrhip_wk_init(g.b[0], r->max_level != 1 || r->w0 != r->w_start))) == LQR_OK) r->wk_valid = 1;
    }
    if (ret == LQR_OK) {
        dp_params(r, &p);
        ret = hip_ret(lqrhip_emap_build(g.b[0], &p, r->w, r->h));
    }
    if (ret == LQR_OK) ret = hip_ret(lqrhip_read_working(r->dev, r->w, r->h, buffer, NULL, NULL));
    group_close(&g);
    return ret;
}

LqrRetVal lqrx_carver_debug_maps(LqrCarver *r, gfloat *en, gfloat *m, gint *least_dx)
{
    if (!r->active || !r->wk_valid) return LQR_ERROR;
    HIP_CATCH(lqrhip_read_working(r->dev, r->w, r->h, en, m, least_dx));
    return LQR_OK;
}
gint lqrx_carver_debug_width(LqrCarver *r) { return r->dbg_w; }
gint lqrx_carver_debug_height(LqrCarver *r) { return r->dbg_h; }
LqrRetVal lqrx_carver_debug_snapshot(LqrCarver *r, gfloat *en, gfloat *m, gint *least_dx)
{
    size_t n = (size_t) r->dbg_w * r->dbg_h;
    if (!r->dbg_m) return LQR_ERROR;
    if (en) memcpy(en, r->dbg_en, n * sizeof(float));
    if (m) memcpy(m, r->dbg_m, n * sizeof(float));
    if (least_dx) memcpy(least_dx, r->dbg_least, n * sizeof(int));
    return LQR_OK;
}
