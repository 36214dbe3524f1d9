// lqr_hip.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the seam-carving
// engine and the C-ABI shim of include/lqr_hip.h that launches them.
//
// The stages replace liblqr-1's CPU engine as reached from the plug-in's
// render path (gimp-lqr-plugin src/render.c:318,328,529 -> lqr_carver_resize):
// energy (E3/E4/E6), cumulative-min DP (E5/E9), seam pick + backtrack (E7),
// carve (E8), visibility map, inflate/flatten/transpose (E11/E14), read-out
// (E12).  Stage numbering: SURVEY.md section 8(a).
//
// Data layout.  liblqr keeps every plane indexed by pixel
// id and moves only an index map; that is hostile to coalescing, so this engine
// keeps two representations:
//   * base layout  (w0 x h0): rgb0 u8*ch, vs i32, bias0/rig0 f32 -- the
//     multi-size image; touched only by the one-off passes;
//   * carved planes (row stride S, h rows): en f32, m f32, least i8 (back pointer
//     as dx), optional rig f32 -- physically COMPACTED: carving a seam moves the
//     shorter side of it by one, and the image's origin with it (FLAG_ORG);
//     pix u32 (packed channels) and bias f32 stay frozen in an older frame.
// All floating point is done with explicitly rounded operations (no FMA
// contraction, IEEE division and sqrt) so that results are bit-identical to the
// C arithmetic of the CPU path.  No MFMA: this is stencil + scan + shift work
// bounded by HBM bandwidth and by the H-step dependency chains.
//
// Order of this file: descriptors and the origin-advanced plane view; energy kernels; generic DP sweep
// (k_dp_sweep) and backtrack (k_vpath, k_vpath1, which also pick the side the carve moves); carve
// (k_carve); energy update (k_emap_update, k_frozen_catchup); the update_mmap kernels -- generic band
// (k_band_update), multi-wave band (k_band_update_mw), the shared DP row (dp_row4), trapezoid-wave band
// (k_band_update_tw), tiled sweeps (k_dp_tile, k_dp_tile_p); visibility map / inflate / compaction /
// transpose; then the host side of the shim (allocation cache, batches, lqrhip_seam_step's per-seam
// sequence, read-out, reset from device memory, copy ceiling, seam-map colour ramp).
#include <hip/hip_runtime.h>
#include <utility>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <map>
#include <deque>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <dirent.h>
#include <unistd.h>

#include "../../include/lqr_hip.h"

// ---------------------------------------------------------------------------
// constants / descriptors
// ---------------------------------------------------------------------------
#define LEAST_INVALID (-128)
#define FLAG_OVF_ROW 0          // first row the band kernel could not handle (h = none)
// The carved planes (en, m, back pointers, rigidity mask) of an image start FLAG_ORG elements into each row:
// a seam is removed by moving whichever side of it is shorter, and moving the left side
// one to the right advances the origin by one.  k_vpath* picks the side for the seam it found and publishes
// {origin before the seam, side, origin after it}; the carve works on physical positions with the first
// two, every other kernel sees rows through pointers advanced by the current origin (gview).
#define FLAG_ORG 3              // origin every kernel but the carve uses
#define FLAG_ORG_PREV 4         // origin of the frame the seam in seam_x was found in (carve)
#define FLAG_SIDE 5             // 0: the part right of the seam moves left; 1: the part left of it moves right
#define FLAG_COUNT 8
#define FLAG_WORDS 64           // size of a carver's flag block
#define DP_THREADS 1024
#define VPATH_THREADS 256
#define BAND_PXL 4
#define BAND_WIN (64 * BAND_PXL)

struct DevCarver {
    // base layout
    uint8_t *rgb0;
    int32_t *vs;
    float *bias0;
    float *rig0;
    // working planes
    uint32_t *pix;
    float *en;
    float *m;
    int8_t *least;
    float *m2;              // second m / back-pointer planes: output of the out-of-place tiled update,
    int8_t *least2;         // swapped with m / least afterwards (by the last tile of k_dp_tile_p<UPDATE>)
    float *bias;
    float *rig;
    int32_t *seam_x;
    int32_t *seam_log;
    int32_t *flags;
};

// Pointers fetched from a descriptor in memory are "generic" to the compiler, which then
// emits flat_load/flat_store (slower issue, and every access also ties up lgkmcnt, which
// defeats software prefetching).  Kernels therefore work on a view whose members are typed
// as address-space-1 (global) pointers, so that they become global_load/global_store.
#define GLOBAL_AS __attribute__((address_space(1)))
typedef GLOBAL_AS uint8_t gu8;
typedef GLOBAL_AS uint32_t gu32;
typedef GLOBAL_AS int32_t gi32;
typedef GLOBAL_AS int8_t gi8;
typedef GLOBAL_AS float gf32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vectors: usable through address-space pointers
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GCarver {
    gu8 *rgb0;
    gi32 *vs;
    gf32 *bias0, *rig0;
    gu32 *pix;
    gf32 *en, *m, *m2;
    gi8 *least, *least2;
    gf32 *bias, *rig;
    gi32 *seam_x, *seam_log, *flags;
};

// physical view: plane pointers as allocated (row y starts at y * stride); what the carve and the one-off
// kernels that lay the planes out use
__device__ __forceinline__ GCarver gview_phys(const DevCarver &d)
{
    GCarver g;
    g.rgb0 = (gu8 *) d.rgb0; g.vs = (gi32 *) d.vs; g.bias0 = (gf32 *) d.bias0; g.rig0 = (gf32 *) d.rig0;
    g.pix = (gu32 *) d.pix; g.en = (gf32 *) d.en; g.m = (gf32 *) d.m; g.least = (gi8 *) d.least;
    g.m2 = (gf32 *) d.m2; g.least2 = (gi8 *) d.least2;
    g.bias = (gf32 *) d.bias; g.rig = (gf32 *) d.rig;
    g.seam_x = (gi32 *) d.seam_x; g.seam_log = (gi32 *) d.seam_log; g.flags = (gi32 *) d.flags;
    return g;
}
// logical view: the carved planes advanced by the image's current origin, so that x = 0 is the first pixel of
// the carved frame in every kernel that indexes by frame coordinates.  The origin is uniform over the rows of an
// image, so rows stay mutually aligned; vector accesses become element-aligned only (measured on gfx950:
// correct, 12-17 % slower than 16-byte aligned ones, scripts/dbg/t_unaligned.hip).  pix / bias stay frozen in
// their own frame (k_emap_update) and are not shifted.
__device__ __forceinline__ GCarver gview(const DevCarver &d)
{
    GCarver g = gview_phys(d);
    const int org = g.flags[FLAG_ORG];
    g.en += org; g.m += org; g.least += org;
    if (g.m2) { g.m2 += org; g.least2 += org; }
    if (g.rig) g.rig += org;
    return g;
}

struct DpK {
    int delta;
    int use_rig;
    float rigmap[2 * LQRHIP_MAX_DELTA + 1];
    int nrg;
    int radius;
    int w_start;
    int ch;
};

static thread_local std::string g_err;
static int g_device = -1;

// Device-side failures (a spin wait that timed out because a persistent grid was not co-resident, a
// prediction that did not hold) never trap: the kernel stores a code in this host-mapped word and
// leaves; the host finds it at its next synchronisation and returns LQRHIP_EHIP (-> LQR_ERROR).
#define DEVERR_TILE_TIMEOUT 1
#define DEVERR_BAND_PREDICTION 2
static int *g_dev_err_host = nullptr;      // hipHostMalloc'ed, mapped
static int *g_dev_err = nullptr;           // its device address
__device__ __forceinline__ void dev_fail(int *flag, int code)
{
    __hip_atomic_store(flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ int dev_failed(int *flag)
{
    return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define HIPCK_VOID(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { g_err = std::string(#expr) + ": " + hipGetErrorString(e__); (void) hipGetLastError(); } } while (0)
#define HIPCK(expr)                                                                   \
    do {                                                                              \
        hipError_t e__ = (expr);                                                      \
        if (e__ != hipSuccess) {                                                      \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e__);               \
            (void) hipGetLastError();                                                 \
            return (e__ == hipErrorOutOfMemory) ? LQRHIP_ENOMEM : LQRHIP_EHIP;        \
        }                                                                             \
    } while (0)

// ---------------------------------------------------------------------------
// exactly-rounded helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ double norm255(uint32_t v) { return __ddiv_rn((double) v, 255.0); }

// brightness / luma of one packed pixel, in double, times alpha (E3).  N255(v) = (double) v / 255.0, correctly rounded:
// norm255 computes it (an FP64 division: ~40 instructions); kernels that need many of them keep the 256 quotients in
// an LDS table filled once per workgroup with that same division (fill_norm255 / Norm255Lut) -- bit-identical values
struct Norm255Div { __device__ __forceinline__ double operator()(uint32_t v) const { return norm255(v); } };
struct Norm255Lut {
    const double *t;
    __device__ __forceinline__ double operator()(uint32_t v) const { return t[v]; }
};
__device__ __forceinline__ void fill_norm255(double *t, int tid, int nthreads)
{
    for (int v = tid; v < 256; v += nthreads) t[v] = norm255((uint32_t) v);
}
template <class N255>
__device__ __forceinline__ double px_bright(uint32_t p, int ch, bool luma, N255 norm255)
{
    double b;
    uint32_t c0 = p & 0xffu, c1 = (p >> 8) & 0xffu, c2 = (p >> 16) & 0xffu, c3 = p >> 24;
    if (ch <= 2) {
        b = norm255(c0);
        if (ch == 2) b = __dmul_rn(b, norm255(c1));
    } else {
        double r = norm255(c0), g = norm255(c1), bl = norm255(c2);
        if (luma)
            b = __dadd_rn(__dadd_rn(__dmul_rn(0.2126, r), __dmul_rn(0.7152, g)), __dmul_rn(0.0722, bl));
        else
            b = __ddiv_rn(__dadd_rn(__dadd_rn(r, g), bl), 3.0);
        if (ch == 4) b = __dmul_rn(b, norm255(c3));
    }
    return b;
}

// gradient energy of carved-frame pixel (x,y) on a w x h frame (E4/compute_e).
// NRG (LqrEnergyFuncBuiltinType) is a template parameter: with a run-time
// energy selector hipcc (ROCm 7.2, gfx950) miscompiled the XABS branch of this
// function (it returned an un-normalised channel value; scripts/dbg/t_energy2.hip
// reproduces it), so every kernel that evaluates the energy is instantiated
// once per energy function and the selector is resolved at launch.
template <int NRG, class BF>
__device__ __forceinline__ float grad_energy_f(BF B, int x, int y, int w, int h)
{
    if (NRG == 6) return 0.0f;
    constexpr int kind = NRG % 3;          // 0 norm, 1 sumabs, 2 xabs
    double gx, gy = 0.0;
    if (kind != 2) {
        if (h == 1) gy = 0.0;
        else if (y == 0) gy = __dsub_rn(B(x, 1), B(x, 0));
        else if (y < h - 1) gy = __dmul_rn(__dsub_rn(B(x, y + 1), B(x, y - 1)), 0.5);
        else gy = __dsub_rn(B(x, y), B(x, y - 1));
    }
    if (w == 1) gx = 0.0;
    else if (x == 0) gx = __dsub_rn(B(1, y), B(0, y));
    else if (x < w - 1) gx = __dmul_rn(__dsub_rn(B(x + 1, y), B(x - 1, y)), 0.5);
    else gx = __dsub_rn(B(x, y), B(x - 1, y));
    double g;
    if (kind == 0) g = __dsqrt_rn(__dadd_rn(__dmul_rn(gx, gx), __dmul_rn(gy, gy)));
    else if (kind == 1) g = __dmul_rn(__dadd_rn(fabs(gx), fabs(gy)), 0.5);
    else g = fabs(gx);
    return __double2float_rn(g);
}

template <int NRG, class N255>
__device__ __forceinline__ float grad_energy(const gu32 *pix, int stride, int x, int y, int w, int h, int ch, N255 n255)
{
    constexpr bool luma = (NRG >= 3);
    return grad_energy_f<NRG>([&](int xx, int yy) { return px_bright(pix[(size_t) yy * stride + xx], ch, luma, n255); }, x, y, w, h);
}

template <int NRG, class N255>
__device__ __forceinline__ float energy_at(const GCarver &c, const DpK &p, int stride, int x, int y, int w, int h, N255 n255)
{
    float e = grad_energy<NRG>(c.pix, stride, x, y, w, h, p.ch, n255);
    if (c.bias) e = __fadd_rn(e, __fdiv_rn(c.bias[(size_t) y * stride + x], (float) p.w_start));
    return e;
}

// resolve the run-time energy selector to a kernel instantiation
#define NRG_DISPATCH(nrg, LAUNCH)                 \
    switch (nrg) {                                \
        case 0: { LAUNCH(0); break; }             \
        case 1: { LAUNCH(1); break; }             \
        case 2: { LAUNCH(2); break; }             \
        case 3: { LAUNCH(3); break; }             \
        case 4: { LAUNCH(4); break; }             \
        case 5: { LAUNCH(5); break; }             \
        default: { LAUNCH(6); break; }            \
    }

// ---------------------------------------------------------------------------
// one-off kernels: working-plane init, full energy map, masks
// ---------------------------------------------------------------------------
__global__ void k_wk_init(const DevCarver *cs, int w, int h, int stride, int ch)
{
    const GCarver c = gview_phys(cs[blockIdx.z]);
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x == 0 && y == 0) { c.flags[FLAG_ORG] = 0; c.flags[FLAG_ORG_PREV] = 0; c.flags[FLAG_SIDE] = 0; }      // planes laid out afresh
    if (x >= stride) return;
    size_t o = (size_t) y * stride + x;
    uint32_t p = 0;
    float b = 0.0f, r = 0.0f;
    if (x < w) {
        const gu8 *s = c.rgb0 + ((size_t) y * w + x) * ch;
        if (ch == 4) p = *(const gu32 *) s;
        else for (int k = 0; k < ch; k++) p |= (uint32_t) s[k] << (8 * k);
        if (c.bias0) b = c.bias0[(size_t) y * w + x];
        if (c.rig0) r = c.rig0[(size_t) y * w + x];
    }
    c.pix[o] = p;
    if (c.bias) c.bias[o] = b;
    if (c.rig) c.rig[o] = r;
}

template <int NRG>
__global__ void k_emap_full(const DevCarver *cs, DpK p, int w, int h, int stride)
{
    __shared__ double s_n255[256];
    fill_norm255(s_n255, threadIdx.x, blockDim.x);
    __syncthreads();
    const GCarver c = gview(cs[blockIdx.z]);
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    c.en[(size_t) y * stride + x] = energy_at<NRG>(c, p, stride, x, y, w, h, Norm255Lut{s_n255});
}

// E2: mask value = mean(colour)/255 * alpha/255 (help/en/index.wiki:48)
__global__ void k_mask_add(float *plane, int w0, const uint8_t *mask, int channels, int mw, int x0, int y0, int x1, int y1,
                           int nx, int ny, int transposed, int is_rig, int bias_factor)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= nx || y >= ny) return;
    const uint8_t *px = mask + ((size_t) (y - y0) * mw + (x - x0)) * channels;
    const bool has_alpha = (channels == 2 || channels >= 4);
    const int cc = channels - (has_alpha ? 1 : 0);
    int sum = 0;
    for (int k = 0; k < cc; k++) sum += px[k];
    int xc = transposed ? y + y1 : x + x1;
    int yc = transposed ? x + x1 : y + y1;
    size_t o = (size_t) yc * w0 + xc;
    if (is_rig) {
        double v = __ddiv_rn((double) sum, (double) (255 * cc));
        if (has_alpha) v = __dmul_rn(v, __ddiv_rn((double) px[channels - 1], 255.0));
        plane[o] = __double2float_rn(v);
    } else {
        double b = __ddiv_rn(__dmul_rn((double) bias_factor, (double) sum), (double) (2 * 255 * cc));
        if (has_alpha) b = __dmul_rn(b, __ddiv_rn((double) px[channels - 1], 255.0));
        plane[o] = __fadd_rn(plane[o], __double2float_rn(b));
    }
}

// ---------------------------------------------------------------------------
// E5 / E9 (full width): cumulative-min DP row sweep, one persistent workgroup
// per image.  The previous row of m lives in LDS (ping-pong), so the only HBM
// traffic is en in, m + back-pointer out (9 B/px), all coalesced.  One
// s_barrier per row.  UPDATE applies liblqr's update_mmap keep-rule to every
// pixel of rows >= flags[FLAG_OVF_ROW]; applied to a superset of liblqr's band
// it leaves identical memory contents (pixels outside the band have unchanged
// inputs, float ops are deterministic).
// ---------------------------------------------------------------------------
template <int PXT, bool UPDATE>
__global__ __launch_bounds__(DP_THREADS) void k_dp_sweep(const DevCarver *cs, DpK p, int w, int h, int stride, int lr)
{
    const GCarver c = gview(cs[blockIdx.x]);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wpad = (w + 3) & ~3;
    float *prev = sm, *cur = sm + wpad;
    const int tid = threadIdx.x;
    int y0 = 0;
    if (UPDATE) {
        y0 = c.flags[FLAG_OVF_ROW];
        if (y0 >= h) return;
    }
    if (y0 == 0) {
        for (int x = tid; x < w; x += DP_THREADS) {
            float e = c.en[x];
            c.m[x] = e;
            prev[x] = e;
        }
        y0 = 1;
    } else {
        for (int x = tid; x < w; x += DP_THREADS) prev[x] = c.m[(size_t) (y0 - 1) * stride + x];
    }
    __syncthreads();

    float e_nx[PXT], mo_nx[PXT], rf_nx[PXT];
    int lo_nx[PXT];
    auto prefetch = [&](int y) {
#pragma unroll
        for (int k = 0; k < PXT; k++) {
            int x = tid + k * DP_THREADS;
            if (x < w && y < h) {
                size_t o = (size_t) y * stride + x;
                e_nx[k] = c.en[o];
                if (UPDATE) { mo_nx[k] = c.m[o]; lo_nx[k] = c.least[o]; }
                if (c.rig) rf_nx[k] = c.rig[o];
            }
        }
    };
    prefetch(y0);
    for (int y = y0; y < h; y++) {
        float e[PXT], mo[PXT], rf[PXT];
        int lo[PXT];
#pragma unroll
        for (int k = 0; k < PXT; k++) { e[k] = e_nx[k]; mo[k] = mo_nx[k]; lo[k] = lo_nx[k]; rf[k] = rf_nx[k]; }
        prefetch(y + 1);
#pragma unroll
        for (int k = 0; k < PXT; k++) {
            int x = tid + k * DP_THREADS;
            if (x < w) {
                const int dlo = max(-x, -p.delta), dhi = min(w - 1 - x, p.delta);
                const float rfact = c.rig ? rf[k] : 1.0f;
                float best = prev[x + dlo];
                if (p.use_rig) best = __fadd_rn(best, __fmul_rn(rfact, p.rigmap[dlo + p.delta]));
                int bdx = dlo;
                for (int dx = dlo + 1; dx <= dhi; dx++) {
                    float cand = prev[x + dx];
                    if (p.use_rig) cand = __fadd_rn(cand, __fmul_rn(rfact, p.rigmap[dx + p.delta]));
                    if (cand < best || (cand == best && lr)) { best = cand; bdx = dx; }
                }
                float nm = __fadd_rn(e[k], best);
                size_t o = (size_t) y * stride + x;
                if (UPDATE) {
                    if (lo[k] == bdx && (double) fabsf(__fsub_rn(mo[k], nm)) < 1e-5) nm = mo[k];
                    else c.m[o] = nm;
                } else {
                    c.m[o] = nm;
                }
                c.least[o] = (int8_t) bdx;
                cur[x] = nm;
            }
        }
        __syncthreads();
        float *t = prev; prev = cur; cur = t;
    }
}

// ---------------------------------------------------------------------------
// E7 build_vpath: argmin of the last row of m with liblqr's tie rule, then the
// backtrack through the back-pointer plane.  One workgroup per image finds the
// argmin; wave 0 then walks the H-step pointer chase entirely in registers:
// rows are taken in chunks of R (R*delta <= 62); lane L holds, for each row of the
// chunk, the 4 back-pointer bytes of columns xa+4L..xa+4L+3 of a 256-column window
// (one coalesced 256-byte load per row), and a chase step is v_readlane + a few
// scalar ops -- no memory or LDS on the dependency chain.  The next chunk starts
// within +-R*delta of this chunk's start column, so its (256-wide) window can be
// loaded into a second register set before this chunk's chase has finished.
// ---------------------------------------------------------------------------
#define VP_ROWS 62
// argmin of row `mrow` (w floats) over a VPATH_THREADS-thread block with liblqr's tie rule: leftmost (lr = 0) / rightmost
// (lr = 1) of equal minima; returns (to every thread) the column, or -1 if no candidate beat liblqr's start value 2^29.
// A thread takes 16 B at a time (4 loads in flight per thread, all issued before the first compare: the row is one
// memory round trip, not fifteen), keeps its own ascending scan, then the block reduces on (value, index) pairs: within
// a wave by DPP-free shuffles, across the four waves through LDS.
__device__ __forceinline__ int row_argmin(const gf32 *mrow, int w, int lr, float *s_val, int *s_idx)
{
    const int tid = threadIdx.x;
    const float INF = __int_as_float(0x7f800000);
    float bv = INF;
    int bi = -1;
    for (int base = 0; base < w; base += 16 * VPATH_THREADS) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = base + 4 * (tid + VPATH_THREADS * i);
            // whole vectors only where they are inside the row (the planes have >= 16 floats of padding, but not initialised)
            if (x + 3 < w) v[i] = *(const GLOBAL_AS f32x4 *) (mrow + x);
            else { v[i] = (f32x4) {INF, INF, INF, INF}; for (int j = 0; j < 4; j++) if (x + j < w) v[i][j] = mrow[x + j]; }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = base + 4 * (tid + VPATH_THREADS * i) + j;
                const float f = v[i][j];
                if (x < w && (f < bv || (f == bv && lr))) { bv = f; bi = x; }
            }
    }
    auto better = [&](float v2, int i2, float v1, int i1) {        // does (v2, i2) replace (v1, i1)?
        if (i2 < 0) return false;
        if (i1 < 0) return true;
        if (v2 < v1) return true;
        if (v2 > v1) return false;
        return lr ? (i2 > i1) : (i2 < i1);
    };
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(bv, o);
        const int i2 = __shfl_xor(bi, o);
        if (better(v2, i2, bv, bi)) { bv = v2; bi = i2; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = bv; s_idx[tid >> 6] = bi; }
    __syncthreads();
    bv = s_val[0]; bi = s_idx[0];
#pragma unroll
    for (int k = 1; k < VPATH_THREADS / 64; k++) if (better(s_val[k], s_idx[k], bv, bi)) { bv = s_val[k]; bi = s_idx[k]; }
    // liblqr starts from m = 2^29: a candidate must beat it (or tie it when lr == 1)
    const float lim = 536870912.0f;
    const bool ok = (bi >= 0) && (bv < lim || (bv == lim && lr));
    return ok ? bi : -1;
}

// Which side of the seam the carve moves (wave 0 of k_vpath*, after the backtrack): the part right of the
// seam holds sum(w - 1 - x), the part left of it sum(x) elements over the rows; the shorter one moves, and
// moving the left part right advances the image's origin by one.  `acc` = this lane's share of sum(x).
__device__ __forceinline__ void publish_side(const GCarver &c, int org, int acc, int w, int h, int lane)
{
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const int side = (2ll * acc < (long long) h * (w - 1)) ? 1 : 0;
    if (lane == 0) { c.flags[FLAG_ORG_PREV] = org; c.flags[FLAG_SIDE] = side; c.flags[FLAG_ORG] = org + side; }
}

__global__ __launch_bounds__(VPATH_THREADS) void k_vpath(const DevCarver *cs, int w, int h, int stride, int lr, int delta,
                                                          int log_index)
{
    const GCarver c = gview(cs[blockIdx.x]);
    const int org = c.flags[FLAG_ORG];           // read before wave 0 publishes the next one
    __shared__ float s_val[VPATH_THREADS / 64];
    __shared__ int s_idx[VPATH_THREADS / 64];
    const int tid = threadIdx.x;

    // ---- argmin over the last row: leftmost (lr=0) / rightmost (lr=1) of equals
    const int xmin = row_argmin(c.m + (size_t) (h - 1) * stride, w, lr, s_val, s_idx);
    if (tid >= 64) return;                       // the chase is one wave
    int x = __builtin_amdgcn_readfirstlane(max(xmin, 0));

    // ---- backtrack
    const int lane = tid;
    gi32 *seam = c.seam_x;
    gi32 *logp = c.seam_log + (size_t) log_index * h;
    const int R = delta > 0 ? min(VP_ROWS, max(1, VP_ROWS / delta)) : VP_ROWS;   // rows per chunk, R*delta <= 62
    uint32_t regs[2][VP_ROWS];
    // window of the chunk whose top row is y_top, for a start column within +-R*delta of cx
    auto window_base = [&](int cx) { return (cx - 126) & ~3; };
    auto load_chunk = [&](int b, int y_top, int xa) {
        const int xl = xa + 4 * lane;
        const bool ok = (xl >= 0) && (xl + 3 < stride);
#pragma unroll
        for (int r = 0; r < VP_ROWS; r++) {
            const int y = max(y_top - r, 0);
            regs[b][r] = ok ? *(const gu32 *) (c.least + (size_t) y * stride + xl) : 0u;
        }
    };
    int y_top = h - 1;
    int xa_cur = window_base(x);
    int acc = 0;
    if (y_top >= 1) load_chunk(0, y_top, xa_cur);
    while (y_top >= 1) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            if (y_top >= 1) {
                const int nrows = min(R, y_top);
                const int xa_next = window_base(x);
                if (y_top - nrows >= 1) load_chunk(b ^ 1, y_top - nrows, xa_next);     // in flight during the chase
                int path = 0;
#pragma unroll
                for (int r = 0; r < VP_ROWS; r++) {
                    if (r < nrows) {
                        path = (lane == r) ? x : path;                                    // lane r <- column at row y_top - r
                        const int o = x - xa_cur;
                        const uint32_t dw = (uint32_t) __builtin_amdgcn_readlane((int) regs[b][r], o >> 2);
                        int d = (int) (int8_t) (dw >> (8 * (o & 3)));
                        d = (d == LEAST_INVALID) ? 0 : d;
                        x += d;
                    }
                }
                if (lane < nrows) { seam[y_top - lane] = path; logp[y_top - lane] = path; acc += path; }
                xa_cur = xa_next;
                y_top -= nrows;
            }
        }
    }
    if (lane == 0) { seam[0] = x; logp[0] = x; acc += x; }
    publish_side(c, org, acc, w, h, lane);
}

// ---------------------------------------------------------------------------
// k_vpath1<DELTA>: k_vpath for delta_x == 1 (described below) and, with 12- / 8- / 4-row chunks, delta_x == 2 / 3 / 4.  The chase is a
// chain of H dependent steps on one wave, so what counts
// is the length of one step and that the back pointers are there when the chase reaches them.  In k_vpath a step
// is v_readlane + 7 scalar instructions (find the lane, pull the dword, extract and sign-extend the byte), ~55 ns.
// Here the rows are taken in chunks of 28:
//   * a 256-column window of back-pointer bytes per row is prefetched THREE chunks ahead (the chunk's start
//     column is then known to within 3 * 28 columns, and it moves at most 28 more inside the chunk; the window
//     leaves 88 columns of margin on each side of the 64 that are used).  One load instruction fetches FOUR rows
//     (16 bytes per lane, 16 lanes per row): a load instruction costs the CU's memory path ~42 cycles whatever its
//     width, and a wave can have only 63 of them outstanding;
//   * when a chunk's turn comes its start column xc is known exactly: the staged rows go through an LDS scratch
//     (row-major, 256 bytes per row: exactly what the loads hold lane by lane) and the 64 columns xc - 32 .. xc + 31
//     come back one per lane, sign-extended (ds_write_b128 x 7, ds_read_i8 x 28, all independent);
//   * rows are then composed in PAIRS, for all 64 columns at once: the two-row displacement of column c is
//     d(r, c) + d(r + 1, c + d(r, c)), one ds_bpermute per pair (independent, pipelined) -- so the chase, the only
//     serial part, has 14 steps per chunk instead of 28.  A step is v_readlane (the lane IS the column) + s_add +
//     a v_writelane that records the path: ~45 cycles with its wait states;
//   * the odd rows' columns are filled in afterwards, all at once (one LDS read).
// No load is guarded or predicated (rows above the image re-read row 1 and their steps are discarded).
// No LEAST_INVALID test: the carve marks a back pointer invalid only next to the seam, inside the interval
// every form of update_mmap recomputes before the next backtrack, so none survives to this point.
// ---------------------------------------------------------------------------
// rows per chunk by delta_x: the path drifts up to ROWS * delta_x columns inside a chunk and must stay within lanes 4 .. 60 of the 64
// spread around its start (<= 28), and the window loaded VP1_AHEAD chunks ahead must still hold those 64 columns
constexpr int vp1_rows(int delta) { return delta == 1 ? 28 : delta == 2 ? 12 : delta == 3 ? 8 : 4; }
#define VP1_AHEAD 3
template <int r>
__device__ __forceinline__ void vp1_step(const int e, int &o, int &path)
{
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(path) : "s"(o), "n"(r));      // lane r <- window offset at row y_top - r
    o += __builtin_amdgcn_readlane(e, o);
}
template <int N, int... Rs>
__device__ __forceinline__ void vp1_chase(const int (&e)[N], int &o, int &path, std::integer_sequence<int, Rs...>)
{
    (vp1_step<Rs>(e[Rs], o, path), ...);
}

template <int DELTA>
__global__ __launch_bounds__(VPATH_THREADS) void k_vpath1(const DevCarver *cs, int w, int h, int stride, int lr, int log_index)
{
    const GCarver c = gview(cs[blockIdx.x]);
    const int org = c.flags[FLAG_ORG];           // read before wave 0 publishes the next one
    __shared__ float s_val[VPATH_THREADS / 64];
    __shared__ int s_idx[VPATH_THREADS / 64];
    constexpr int VP1_ROWS = vp1_rows(DELTA);
    __shared__ __attribute__((aligned(16))) int8_t s_win[VP1_ROWS * 256];     // the current chunk's rows, 256 columns each
    const int tid = threadIdx.x;

    // ---- argmin over the last row: leftmost (lr=0) / rightmost (lr=1) of equals
    const int xmin = row_argmin(c.m + (size_t) (h - 1) * stride, w, lr, s_val, s_idx);
    if (tid >= 64) return;                       // the chase is one wave
    int x = __builtin_amdgcn_readfirstlane(max(xmin, 0));

    // ---- backtrack
    const int lane = tid;
    gi32 *seam = c.seam_x;
    gi32 *logp = c.seam_log + (size_t) log_index * h;
    constexpr int R = VP1_ROWS, NB = VP1_AHEAD + 1, RL = VP1_ROWS / 4;      // RL loads per chunk, four rows each
    // window [base, base + 256) with base in [cx - 135, cx - 120]: the 64 columns around a start column that has moved up to
    // 88 either way since the load are inside
    static_assert(VP1_ROWS % 4 == 0 && VP1_ROWS * DELTA <= 28 && VP1_AHEAD * VP1_ROWS * DELTA + 32 <= 120, "window margin");
    u32x4 regs[NB][RL];                          // ring of packed windows: chunk k lives in regs[k % NB]
    int xa[NB];                                  // their base columns
    auto window_base = [&](int cx) { return (cx - 120) & ~15; };     // multiple of 16: a lane's 16 columns never straddle column 0
    // Nothing is predicated (a select per load cost more instructions than the chase itself): columns outside
    // the plane are clamped into it -- the path never goes there -- and rows above row 1 re-read row 1; the steps
    // taken on those are discarded (see run_chunk).  Uniform row base + 32-bit lane offset: one VALU per load.
    auto load_chunk = [&](int b, int y_top, int cx) {
        const int base = window_base(cx);
        xa[b] = base;
        // lanes 16s .. 16s + 15: row y_top - 4q - s, 16 columns per lane
        const int voff = min(max(base + 16 * (lane & 15), 0), stride - 16);
        const int rsub = (lane >> 4) * stride;
#pragma unroll
        for (int q = 0; q < RL; q++) {
            const int row = max((y_top - 4 * q) * stride - rsub, stride);          // rows above row 1 re-read row 1
            regs[b][q] = *(const GLOBAL_AS u32x4 *) (c.least + (unsigned) (row + voff));
        }
    };
    // one chunk: spread the 64 columns around the start column out over the lanes, chase, record
    int acc = 0;
    auto run_chunk = [&](int b, int y_top) {
        const int relbase = x - 32 - xa[b];                  // window column of lane 0's column
        int e[R];
        // through LDS: what the loads hold lane by lane IS row-major [row][256 columns]; same wave writes and reads, LDS
        // operations of a wave execute in order
#pragma unroll
        for (int q = 0; q < RL; q++) *(u32x4 *) (s_win + q * 1024 + lane * 16) = regs[b][q];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < R; r++) e[r] = s_win[r * 256 + relbase + lane];
        // two-row displacements of every column (the path stays within lanes 4 .. 60, so the wrap-around of the
        // outermost lanes' neighbours never matters)
        int e2[R / 2];
#pragma unroll
        for (int k = 0; k < R / 2; k++) e2[k] = e[2 * k] + __builtin_amdgcn_ds_bpermute((lane + e[2 * k]) << 2, e[2 * k + 1]);
        int o = 32, path = 0;
        vp1_chase<R / 2>(e2, o, path, std::make_integer_sequence<int, R / 2>{});          // lane k <- window offset at row y_top - 2k
        // the odd rows, all at once: lane k looks its even row's displacement up at the column it stands on
        const int odd = path + s_win[min(lane, R / 2 - 1) * 512 + relbase + path];
        __builtin_amdgcn_wave_barrier();
        const int pe = path + x - 32, po = odd + x - 32;     // columns at rows y_top - 2k and y_top - 2k - 1
        const int nrows = min(R, y_top);
        if (2 * lane < nrows) { seam[y_top - 2 * lane] = pe; logp[y_top - 2 * lane] = pe; acc += pe; }
        if (2 * lane + 1 < nrows) { seam[y_top - 2 * lane - 1] = po; logp[y_top - 2 * lane - 1] = po; acc += po; }
        // the column after `nrows` steps: the last chunk may hold fewer real rows than R (the rest re-read row 1)
        x = (nrows == R) ? x + o - 32 : __builtin_amdgcn_readlane((nrows & 1) ? po : pe, nrows >> 1);
    };
    int y_top = h - 1;
    // chunks are issued VP1_AHEAD ahead; the first ones all around the argmin
#pragma unroll
    for (int k = 0; k < VP1_AHEAD; k++) load_chunk(k, y_top - k * R, x);
    while (true) {
#pragma unroll
        for (int k = 0; k < NB; k++) {
            load_chunk((k + VP1_AHEAD) % NB, y_top - VP1_AHEAD * R, x);      // in flight during this and the next two chases
            run_chunk(k, y_top);
            y_top -= R;
            if (y_top < 1) break;
        }
        if (y_top < 1) break;
    }
    if (lane == 0) { seam[0] = x; logp[0] = x; acc += x; }
    publish_side(c, org, acc, w, h, lane);
}

// ---------------------------------------------------------------------------
// E8 carve: remove the seam from every carved plane (en, m, back pointers, rigidity mask), in place, one
// wave per row, 16 B per lane.  The dominant HBM kernel.  Only the SHORTER side of the seam moves
// (DESIGN.md 4.9): k_vpath* compared sum(x) with sum(w - 1 - x) over the seam's rows and published
//   side 0: the part right of the seam moves one to the left, the origin stays;
//   side 1: the part left of it moves one to the right and the image's origin advances by one.
// The side is uniform over an image's rows, so rows stay mutually aligned and every other kernel just sees
// the planes through pointers advanced by the origin.  Seams are delta_x-connected, so a seam's rows differ
// little in x and the per-image choice loses almost nothing against a per-row one; for seams spread over
// the width the mean moved fraction of a row is 1/4 instead of 1/2.
// This kernel works on PHYSICAL positions p = origin + x (rows start 16-byte aligned at p = 0), with aligned
// vector accesses; the element that enters a lane's four from the neighbouring lane comes by DPP.
// The back-pointer plane is re-based on the fly: a stored dx stays valid unless pixel and parent are on
// different sides of the seam (then it changes by one, or becomes LEAST_INVALID if the parent was carved).
// ---------------------------------------------------------------------------
// write-through (sc1) stores: the data reaches memory without a release fence (buffer_wbl2), so a
// drained wave (s_waitcnt vmcnt(0)) can publish a flag that a consumer on another XCD may trust
// (s_nop 1 inside the string: the hazard recogniser does not see the store, and the next VALU instruction may otherwise
// overwrite its data registers before the store has read them -- cdna_hip_programming.md, asm stores)
__device__ __forceinline__ void store_sc1_x4(gu32 *p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }

#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138

// A group = CG chunks of 256 elements of one plane, as loaded, plus the one element beyond the group that
// its edge lane needs.  All planes of a row are LOADED for a group before any of them is stored: a row wave then has
// three planes' loads in flight at once instead of three load -> store round trips one after the other (non-temporal:
// streaming the rows past L2 is worth 6 % of the kernel).  The element that follows (side 0) / precedes (side 1) a
// lane's four is the neighbouring lane's (DPP); only the edge lane of the group fetches it from memory.
// Chunks per group.  Measured at 64 x 4K (one box, us per launch): planes one after the other at 8 waves per SIMD 510-518;
// all planes of a group loaded first with CG = 1 (77 VGPRs, 6 waves) 495, 2 (100, 4) 486, 3 (120, 4) 509, 4 (142 VGPRs,
// 3 waves per SIMD) 470-485, 6 552; CG = 4 squeezed into 128 VGPRs (11 spilled) 531.  Bytes in flight per wave beat
// occupancy: a row's mean moved part (a quarter of 3840) fits one group of 1024.
#ifndef CG
#define CG 4
#endif
#define CGPX (CG * 256)
struct G32 { u32x4 a[CG]; uint32_t edge; };     // 4-byte planes: en, m, rigidity mask
struct G8 { uint32_t a[CG]; uint32_t edge; };   // the back-pointer bytes, 4 px per dword

// ---- side 0: new[p] = old[p + 1] for p in [pv, pend); pend = physical end (exclusive) of the row after the carve.
// Groups run left to right from `base`.
__device__ __forceinline__ void ld_left_u32(const gu32 *row, int base, int pend, int lane, G32 &g)
{
#pragma unroll
    for (int u = 0; u < CG; u++) {
        const int x = base + u * 256 + lane * 4;
        // x <= pend: the chunk that starts at pend holds the old last element, which the lane before needs
        g.a[u] = (x <= pend) ? __builtin_nontemporal_load((const GLOBAL_AS u32x4 *) (row + x)) : (u32x4) {0u, 0u, 0u, 0u};
    }
    g.edge = (lane == 63 && base + CGPX <= pend) ? row[base + CGPX] : 0u;
}
// SC1: write-through stores instead of non-temporal ones (no kernel of this build asks for them: round 3's carve that
// announced its rows did; the switch stays for A/B runs)
template <bool SC1 = false>
__device__ __forceinline__ void st_left_u32(gu32 *row, int base, int pv, int pend, int lane, const G32 &g)
{
#pragma unroll
    for (int u = 0; u < CG; u++) {
        const int x = base + u * 256 + lane * 4;
        const uint32_t first_next = (u < CG - 1) ? (uint32_t) __builtin_amdgcn_readlane((int) g.a[u < CG - 1 ? u + 1 : CG - 1].x, 0) : 0u;
        const uint32_t lane63 = (u < CG - 1) ? first_next : g.edge;
        const uint32_t nx = (uint32_t) __builtin_amdgcn_update_dpp((int) lane63, (int) g.a[u].x, DPP_WAVE_SHL1, 0xf, 0xf, false);
        if (x < pend) {
            u32x4 o;
            o.x = (x >= pv) ? g.a[u].y : g.a[u].x;
            o.y = (x + 1 >= pv) ? g.a[u].z : g.a[u].y;
            o.z = (x + 2 >= pv) ? g.a[u].w : g.a[u].z;
            o.w = (x + 3 >= pv) ? nx : g.a[u].w;
            if (SC1) store_sc1_x4(row + x, o); else __builtin_nontemporal_store(o, (GLOBAL_AS u32x4 *) (row + x));
        }
    }
}

// ---- side 1: new[p] = old[p - 1] for p in (pbeg, pv]; pbeg = the origin before the carve (dead afterwards).
// Groups [gbase, gbase + CGPX) run from the seam towards the origin: a group's stores reach one element past its loads
// on the right, into a group that has been read already.
__device__ __forceinline__ void ld_right_u32(const gu32 *row, int gbase, int pbeg, int pv, int lane, G32 &g)
{
#pragma unroll
    for (int u = 0; u < CG; u++) {
        const int x = gbase + u * 256 + lane * 4;
        // chunks that hold a source (p in [pbeg, pv - 1]) or a destination; x + 3 >= pbeg >= 0 keeps x >= 0
        g.a[u] = (x + 3 >= pbeg && x <= pv) ? __builtin_nontemporal_load((const GLOBAL_AS u32x4 *) (row + x)) : (u32x4) {0u, 0u, 0u, 0u};
    }
    g.edge = (lane == 0 && gbase - 1 >= pbeg) ? row[gbase - 1] : 0u;
}
template <bool SC1 = false>
__device__ __forceinline__ void st_right_u32(gu32 *row, int gbase, int pbeg, int pv, int lane, const G32 &g)
{
#pragma unroll
    for (int u = 0; u < CG; u++) {
        const int x = gbase + u * 256 + lane * 4;
        const uint32_t last_prev = (u > 0) ? (uint32_t) __builtin_amdgcn_readlane((int) g.a[u > 0 ? u - 1 : 0].w, 63) : 0u;
        const uint32_t lane0 = (u > 0) ? last_prev : g.edge;
        const uint32_t pw = (uint32_t) __builtin_amdgcn_update_dpp((int) lane0, (int) g.a[u].w, DPP_WAVE_SHR1, 0xf, 0xf, false);
        if (x <= pv && x + 3 > pbeg) {
            u32x4 o;
            o.x = (x > pbeg && x <= pv) ? pw : g.a[u].x;
            o.y = (x + 1 > pbeg && x + 1 <= pv) ? g.a[u].x : g.a[u].y;
            o.z = (x + 2 > pbeg && x + 2 <= pv) ? g.a[u].y : g.a[u].z;
            o.w = (x + 3 > pbeg && x + 3 <= pv) ? g.a[u].z : g.a[u].w;
            if (SC1) store_sc1_x4(row + x, o); else __builtin_nontemporal_store(o, (GLOBAL_AS u32x4 *) (row + x));
        }
    }
}

// one back pointer of the carved frame: the pixel that lands on new frame column xx came from old column
// xo = xx + right with back pointer dx (parent at old column xo + dx on row y - 1, whose seam pixel was vprev)
__device__ __forceinline__ int rebase_dx(int dx, int xx, int xo, int vprev, int y)
{
    if (y > 0 && dx != LEAST_INVALID) {
        const int q = xo + dx;
        if (q == vprev) dx = LEAST_INVALID;            // parent was the carved pixel
        else dx = q - (q > vprev ? 1 : 0) - xx;
    }
    return dx;
}

// ---- back pointers, side 0.  New frame column xx sits at physical org + xx and takes old column xx + (xx >= v).
// Pixels left of the seam whose parent may lie right of the seam of the row above (xx >= start = min(v, vprev - delta))
// are re-based too; bytes outside [start, wnew) are written back as loaded.
__device__ __forceinline__ void ld_left_8(const gu32 *row32, int base, int pend, int lane, G8 &g)
{
#pragma unroll
    for (int u = 0; u < CG; u++) {
        const int x = base + u * 256 + lane * 4;
        g.a[u] = (x <= pend) ? __builtin_nontemporal_load(row32 + (x >> 2)) : 0u;
    }
    g.edge = (lane == 63 && base + CGPX <= pend) ? row32[(base + CGPX) >> 2] : 0u;
}
template <bool SC1 = false>
__device__ __forceinline__ void st_left_8(gu32 *row32, int base, int org, int start, int v, int vprev, int y, int wnew, int lane, const G8 &g)
{
    const int pend = org + wnew;
#pragma unroll
    for (int u = 0; u < CG; u++) {
        const int x = base + u * 256 + lane * 4;
        const uint32_t first_next = (u < CG - 1) ? (uint32_t) __builtin_amdgcn_readlane((int) g.a[u < CG - 1 ? u + 1 : CG - 1], 0) : 0u;
        const uint32_t lane63 = (u < CG - 1) ? first_next : g.edge;
        const uint32_t nx = (uint32_t) __builtin_amdgcn_update_dpp((int) lane63, (int) g.a[u], DPP_WAVE_SHL1, 0xf, 0xf, false);
        if (x < pend) {
            const uint64_t both = ((uint64_t) nx << 32) | g.a[u];
            uint32_t o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int xx = x + j - org;
                int dx;
                if (xx < start || xx >= wnew) {
                    dx = (int8_t) (g.a[u] >> (8 * j));                  // not part of the job: as loaded
                } else {
                    const bool right = (xx >= v);
                    dx = rebase_dx((int8_t) (both >> (8 * (j + (right ? 1 : 0)))), xx, right ? xx + 1 : xx, vprev, y);
                }
                o |= (uint32_t) (uint8_t) (int8_t) dx << (8 * j);
            }
            if (SC1) __hip_atomic_store(row32 + (x >> 2), o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // write-through
            else __builtin_nontemporal_store(o, row32 + (x >> 2));
        }
    }
}

// ---- back pointers, side 1.  New frame column xx sits at physical org + 1 + xx; pixels left of the seam (xx < v) come
// from physical org + xx (they move), pixels right of it stay where they are and are only re-based while their parent
// may lie left of (or on) the seam of the row above (xx < end_l = max(v, vprev + delta)).  Destination range
// [org + 1, org + 1 + end_l).
__device__ __forceinline__ void ld_right_8(const gu32 *row32, int gbase, int org, int ptop, int lane, G8 &g)
{
#pragma unroll
    for (int u = 0; u < CG; u++) {
        const int x = gbase + u * 256 + lane * 4;
        g.a[u] = (x + 3 >= org && x < ptop) ? __builtin_nontemporal_load(row32 + (x >> 2)) : 0u;      // x + 3 >= org >= 0 keeps x >= 0
    }
    g.edge = (lane == 0 && gbase - 1 >= org) ? row32[(gbase - 4) >> 2] : 0u;
}
template <bool SC1 = false>
__device__ __forceinline__ void st_right_8(gu32 *row32, int gbase, int org, int end_l, int v, int vprev, int y, int lane, const G8 &g)
{
    const int pfirst = org + 1, ptop = org + 1 + end_l;
#pragma unroll
    for (int u = 0; u < CG; u++) {
        const int x = gbase + u * 256 + lane * 4;
        const uint32_t last_prev = (u > 0) ? (uint32_t) __builtin_amdgcn_readlane((int) g.a[u > 0 ? u - 1 : 0], 63) : 0u;
        const uint32_t lane0 = (u > 0) ? last_prev : g.edge;
        const uint32_t pd = (uint32_t) __builtin_amdgcn_update_dpp((int) lane0, (int) g.a[u], DPP_WAVE_SHR1, 0xf, 0xf, false);
        if (x < ptop && x + 3 >= pfirst) {
            const uint64_t both = ((uint64_t) g.a[u] << 8) | (pd >> 24);      // byte k = physical x - 1 + k
            uint32_t o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int xx = x + j - pfirst;
                int dx;
                if (xx < 0 || xx >= end_l) {
                    dx = (int8_t) (g.a[u] >> (8 * j));                  // not part of the job: as loaded
                } else {
                    const bool right = (xx >= v);
                    dx = rebase_dx((int8_t) (both >> (8 * (j + (right ? 1 : 0)))), xx, right ? xx + 1 : xx, vprev, y);
                }
                o |= (uint32_t) (uint8_t) (int8_t) dx << (8 * j);
            }
            if (SC1) __hip_atomic_store(row32 + (x >> 2), o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // write-through
            else __builtin_nontemporal_store(o, row32 + (x >> 2));
        }
    }
}

// one row of the carve, by one wave (the body of k_carve's row loop).  c = the
// PHYSICAL view, org / side = what k_vpath* published for this seam, w = the width before the carve.
template <bool SC1 = false>
__device__ __forceinline__ void carve_row(const GCarver &c, int org, int side, int y, int w, int stride, int delta, int move_dp, int lane)
{
    const int wnew = w - 1;
    const int v = c.seam_x[y];
    const size_t ro = (size_t) y * stride;
    const int vprev = y > 0 ? c.seam_x[y - 1] : 0;
    gu32 *en = (gu32 *) (c.en + ro), *mm = (gu32 *) (c.m + ro), *l32 = (gu32 *) (c.least + ro);
    gu32 *rg = c.rig ? (gu32 *) (c.rig + ro) : (gu32 *) nullptr;
    // pix and bias are NOT moved: they stay in the frame of `frozen epoch` and the energy
    // update maps current coordinates back through the seam log (k_emap_update)
    if (side == 0) {
        const int pv = org + v, pend = org + wnew;
        int start = (y > 0) ? min(v, vprev - delta) : v;       // where the back pointers' job starts (<= v)
        if (start < 0) start = 0;
        for (int base = (org + (move_dp ? start : v)) & ~3; base < pend; base += CGPX) {
            G32 E, M;
            G8 L;
            ld_left_u32(en, base, pend, lane, E);
            if (move_dp) { ld_left_u32(mm, base, pend, lane, M); ld_left_8(l32, base, pend, lane, L); }
            st_left_u32<SC1>(en, base, pv, pend, lane, E);
            if (move_dp) { st_left_u32<SC1>(mm, base, pv, pend, lane, M); st_left_8<SC1>(l32, base, org, start, v, vprev, y, wnew, lane, L); }
        }
        if (rg)
            for (int base = pv & ~3; base < pend; base += CGPX) { G32 R; ld_left_u32(rg, base, pend, lane, R); st_left_u32<SC1>(rg, base, pv, pend, lane, R); }
    } else {
        const int pv = org + v;
        const int end_l = (y > 0) ? min(wnew, max(v, vprev + delta)) : min(wnew, v);      // back pointers' job: new columns [0, end_l)
        const int ptop = org + 1 + max(end_l, 0);
        const int top_u = (pv | 3) + 1;
        for (int top = move_dp ? max(top_u, (ptop + 3) & ~3) : top_u; top > org + 1; top -= CGPX) {
            const int gbase = top - CGPX;
            G32 E, M;
            G8 L;
            ld_right_u32(en, gbase, org, pv, lane, E);
            if (move_dp) { ld_right_u32(mm, gbase, org, pv, lane, M); ld_right_8(l32, gbase, org, ptop, lane, L); }
            st_right_u32<SC1>(en, gbase, org, pv, lane, E);
            if (move_dp) { st_right_u32<SC1>(mm, gbase, org, pv, lane, M); st_right_8<SC1>(l32, gbase, org, max(end_l, 0), v, vprev, y, lane, L); }
        }
        if (rg)
            for (int top = top_u; top > org + 1; top -= CGPX) { G32 R; ld_right_u32(rg, top - CGPX, org, pv, lane, R); st_right_u32<SC1>(rg, top - CGPX, org, pv, lane, R); }
    }
}

__global__ __launch_bounds__(256) void k_carve(const DevCarver *cs, int w, int h, int stride, int delta, int move_dp)
{
    // blockIdx.x = row block (fastest): consecutive workgroups take consecutive rows of one image (2.5 % faster than
    // image-fastest, which round 1 used so that a concurrent band update could follow all images' top rows)
    const GCarver c = gview_phys(cs[blockIdx.y]);
    const int org = c.flags[FLAG_ORG_PREV], side = c.flags[FLAG_SIDE];      // published by k_vpath* for this seam
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) c.flags[FLAG_OVF_ROW] = h;       // k_band_tiles lowers it with atomic mins; the other band kernels overwrite it
    for (int y = blockIdx.x * 4 + (threadIdx.x >> 6); y < h; y += gridDim.x * 4) carve_row(c, org, side, y, w, stride, delta, move_dp, lane);
}

// changed-energy interval of row y after carving (liblqr update_emap), w = new width
__device__ __forceinline__ void nrg_interval(const gi32 *seam, int y, int h, int w, int radius, int &xmin, int &xmax)
{
    int y1a = max(y - radius, 0), y1b = min(y + radius, h - 1);
    int lo = seam[y], hi = seam[y] - 1;
    for (int y1 = y1a; y1 <= y1b; y1++) {
        int x = seam[y1];
        lo = min(lo, x - radius);
        hi = max(hi, x + radius - 1);
    }
    xmin = max(0, lo);
    xmax = min(w - 1, hi);
}

// exclusive scan of 0/1 flags over a 256-thread block; returns rank, total via reference
__device__ __forceinline__ int block_rank_256(bool flag, int *s_wave, int &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long bal = __ballot(flag);
    int r = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();                 // protect s_wave reuse
    if (lane == 0) s_wave[wv] = __popcll(bal);
    __syncthreads();
    int off = 0;
    total = 0;
    for (int i = 0; i < 4; i++) { int t = s_wave[i]; if (i < wv) off += t; total += t; }
    return r + off;
}

// E6 update_emap: recompute en next to the carved seam (w = new width); runs after the carve.
// The packed-pixel (and bias) planes are frozen in the frame they had at seam `epoch`
// of the session; current coordinates are mapped back by undoing seams k..epoch of the
// row (p += (log[j] <= p)), which costs O(k - epoch) per pixel for ~10 pixels per row and
// saves moving 4 (8 with bias) of the 13 bytes per pixel that a carve would otherwise move.
// Brightness samples staged per row: row y is asked for columns [min - 2, max + 1] of the seam over rows
// y-1..y+1 by its own gradient and [min - 1, max] of the seam over rows y-2..y+2 by its neighbours', and the
// seam moves at most delta_x per row: at most max(4*delta_x + 2, 2*delta_x + 4) columns.  EU_NT is a template
// parameter chosen by the launch from delta_x: 12 (delta_x <= 2), 36 (<= 8), 68 (<= 16 = LQRHIP_MAX_DELTA).
#define EU_ROWS 62          // rows per block (+2 halo rows)
#ifndef EU_LOGB
#define EU_LOGB 8            // log entries fetched per round of the walk back to the frozen frame
#endif
template <int NRG, int EU_NT>
__global__ __launch_bounds__(64) void k_emap_update(const DevCarver *cs, DpK p, int w, int h, int stride, int k, int epoch)
{
    const GCarver c = gview(cs[blockIdx.y]);
    __shared__ double bt[64][EU_NT];
    __shared__ float bb[64][EU_NT];
    __shared__ int slo[64];
    __shared__ double s_n255[256];
    const int tid = threadIdx.x;
    fill_norm255(s_n255, tid, 64);
    __syncthreads();
    const int y = blockIdx.x * EU_ROWS + tid - 1;
    const bool row_ok = (y >= 0 && y < h);
    constexpr bool luma = (NRG >= 3);
    int xmin = 0, xmax = -1, lo = 0;
    if (row_ok) {
        nrg_interval(c.seam_x, y, h, w, p.radius, xmin, xmax);
        // samples of this row that rows y-1, y, y+1 will ask for
        int l = xmin - 1, r = xmax + 1;
        if (y > 0) { int a, b; nrg_interval(c.seam_x, y - 1, h, w, p.radius, a, b); if (b >= a) { l = min(l, a); r = max(r, b); } }
        if (y < h - 1) { int a, b; nrg_interval(c.seam_x, y + 1, h, w, p.radius, a, b); if (b >= a) { l = min(l, a); r = max(r, b); } }
        lo = max(l, 0);
        int pos[EU_NT];
#pragma unroll
        for (int i = 0; i < EU_NT; i++) pos[i] = lo + i;
        // undo seams k .. epoch, newest first.  The log entries are loaded eight at a time (unconditionally: indices
        // below `epoch` are clamped and their values replaced by one that moves nothing), so that a row does not wait
        // for one global load per logged seam
        const gi32 *lg = c.seam_log + y;
        for (int j = k; j >= epoch; j -= EU_LOGB) {
            int v[EU_LOGB];
#pragma unroll
            for (int u = 0; u < EU_LOGB; u++) v[u] = lg[(size_t) max(j - u, epoch) * h];
#pragma unroll
            for (int u = 0; u < EU_LOGB; u++) {
                const int vu = (j - u >= epoch) ? v[u] : 0x7fffffff;
#pragma unroll
                for (int i = 0; i < EU_NT; i++) pos[i] += (vu <= pos[i]) ? 1 : 0;
            }
        }
        const int wf = w + (k - epoch) + 1;           // width of the frozen frame
#pragma unroll
        for (int i = 0; i < EU_NT; i++) {
            const bool ok = (lo + i <= min(r, w - 1)) && pos[i] < wf;
            const size_t o = (size_t) y * stride + (ok ? pos[i] : 0);
            bt[tid][i] = ok ? px_bright(c.pix[o], p.ch, luma, Norm255Lut{s_n255}) : 0.0;
            bb[tid][i] = (ok && c.bias) ? c.bias[o] : 0.0f;
        }
    }
    slo[tid] = lo;
    __syncthreads();
    if (!row_ok || tid == 0 || tid == 63) return;
    for (int x = xmin; x <= xmax; x++) {
        float e = grad_energy_f<NRG>([&](int xx, int yy) { const int t = tid + (yy - y); return bt[t][xx - slo[t]]; }, x, y, w, h);
        if (c.bias) e = __fadd_rn(e, __fdiv_rn(bb[tid][x - lo], (float) p.w_start));
        c.en[(size_t) y * stride + x] = e;
    }
}

// bring the frozen planes (pix, bias) forward: remove seams [from, to) of the session log from
// every row; w_from = width of the frame the planes are in.  One block per row, in place.
__global__ __launch_bounds__(256) void k_frozen_catchup(const DevCarver *cs, int from, int to, int w_from, int h, int stride)
{
    const GCarver c = gview(cs[blockIdx.y]);
    extern __shared__ int smc[];
    int *xs = smc;                                  // [to - from]
    uint8_t *rem = (uint8_t *) (smc + (to - from));  // [w_from]
    __shared__ int s_wave[4];
    const int y = blockIdx.x, tid = threadIdx.x, n = to - from;
    for (int i = tid; i < n; i += 256) xs[i] = c.seam_log[(size_t) (from + i) * h + y];
    for (int i = tid; i < w_from; i += 256) rem[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        int pz = xs[i];
        for (int j = i - 1; j >= 0; j--) if (xs[j] <= pz) pz++;
        rem[pz] = 1;
    }
    __syncthreads();
    gu32 *prow = c.pix + (size_t) y * stride;
    gf32 *brow = c.bias ? c.bias + (size_t) y * stride : (gf32 *) nullptr;
    int carry = 0;
    for (int base = 0; base < w_from; base += 256) {
        const int col = base + tid;
        const bool keep = (col < w_from) && !rem[col];
        const uint32_t v = (col < w_from) ? prow[col] : 0u;
        const float bv = (brow && col < w_from) ? brow[col] : 0.0f;
        int total;
        const int rank = carry + block_rank_256(keep, s_wave, total);     // barriers inside: all reads of the chunk are done
        if (keep) { prow[rank] = v; if (brow) brow[rank] = bv; }
        carry += total;
    }
}

// ---------------------------------------------------------------------------
// E9 update_mmap, band form: one wave per image walks the rows; lane L owns
// BAND_PXL consecutive pixels of a BAND_WIN-wide window around the band.  The
// previous row of m stays in LDS; rows are prefetched PF deep into registers so
// the per-row critical path is LDS + VALU only.  If the band ever leaves /
// outgrows the window the kernel records the row in flags[FLAG_OVF_ROW] and the
// full-width sweep (k_dp_sweep<UPDATE>) finishes from there -- same results.
// ---------------------------------------------------------------------------
#define BAND_PF 4
struct BandRow {
    float mo[BAND_PXL];
    float e[BAND_PXL];
    float rf[BAND_PXL];
    uint32_t lo;
};

__global__ __launch_bounds__(64) void k_band_update(const DevCarver *cs, DpK p, int w, int h, int stride, int lr)
{
    const GCarver c = gview(cs[blockIdx.x]);
    __shared__ __attribute__((aligned(16))) float prow[BAND_WIN + 2 * LQRHIP_MAX_DELTA + 8];
    const int lane = threadIdx.x;
    const int delta = p.delta;
    const gi32 *seam = c.seam_x;
    float *pr = prow + LQRHIP_MAX_DELTA + 4;      // pr[-delta .. BAND_WIN+delta) addressable

    int a, b;                                     // current band (liblqr's x_min, x_max)
    {
        int n0, n1;
        nrg_interval(seam, 0, h, w, p.radius, n0, n1);
        a = max(n0, 0); b = min(n1, w - 1);
        for (int x = a + lane; x <= b; x += 64) c.m[x] = c.en[x];      // row 0: m = en
    }
    if (h < 2) { if (lane == 0) c.flags[FLAG_OVF_ROW] = h; return; }

    int y = 1;
    int ovf = h;
    const int wmax_base = max(0, ((w + 3) & ~3) - BAND_WIN);
    while (y < h) {
        // ---- (re)base the window for rows y.. : centre it on the band of row y
        int na, nb;
        {
            int n0, n1;
            nrg_interval(seam, y, h, w, p.radius, n0, n1);
            na = max(min(a, n0) - delta, 0);
            nb = min(max(b, n1) + delta, w - 1);
            // the children of the pixel carved on the row above are always in the band (oracle: spec delta 6)
            na = min(na, max(seam[y - 1] - delta - 1, 0));
            nb = max(nb, min(seam[y - 1] + delta, w - 1));
        }
        if (nb - na + 1 + 2 * delta > BAND_WIN - 8) { ovf = y; break; }
        int centre = (na + nb) >> 1;
        int B = min(max((centre - BAND_WIN / 2) & ~3, 0), wmax_base);
        if (na - delta < B && B > 0) { ovf = y; break; }
        if (nb + delta >= B + BAND_WIN && B + BAND_WIN < w) { ovf = y; break; }
        const int x0 = B + lane * BAND_PXL;
        // previous row of m over the window (+halo) into LDS.  Row y-1 was stored
        // by this wave (or is untouched): make the stores visible first.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        {
            const gf32 *mp = c.m + (size_t) (y - 1) * stride;
            for (int i = lane; i < BAND_WIN + 2 * delta; i += 64) {
                int x = B - delta + i;
                float v = 0.0f;
                if (x >= 0 && x < w) v = __hip_atomic_load((gf32 *) (mp + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pr[i - delta] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();

        BandRow q[BAND_PF];
        auto load_row = [&](BandRow &r, int yy) {
            if (yy < h) {
                size_t o = (size_t) yy * stride + x0;
                f32x4 mv = *(const GLOBAL_AS f32x4 *) (c.m + o);
                f32x4 ev = *(const GLOBAL_AS f32x4 *) (c.en + o);
                r.mo[0] = mv.x; r.mo[1] = mv.y; r.mo[2] = mv.z; r.mo[3] = mv.w;
                r.e[0] = ev.x; r.e[1] = ev.y; r.e[2] = ev.z; r.e[3] = ev.w;
                r.lo = *(const gu32 *) (c.least + o);
                if (c.rig) {
                    f32x4 rv = *(const GLOBAL_AS f32x4 *) (c.rig + o);
                    r.rf[0] = rv.x; r.rf[1] = rv.y; r.rf[2] = rv.z; r.rf[3] = rv.w;
                }
            }
        };
#pragma unroll
        for (int d = 0; d < BAND_PF; d++) load_row(q[d], y + d);

        bool rebase = false;
        while (y < h && !rebase) {
#pragma unroll
            for (int d = 0; d < BAND_PF; d++) {
                if (y < h && !rebase) {
                    int n0, n1;
                    nrg_interval(seam, y, h, w, p.radius, n0, n1);
                    int ra = max(min(a, n0) - delta, 0);
                    int rb = min(max(b, n1) + delta, w - 1);
                    ra = min(ra, max(seam[y - 1] - delta - 1, 0));
                    rb = max(rb, min(seam[y - 1] + delta, w - 1));
                    // the band (plus its parents) must sit inside the window
                    bool fits = (ra - delta >= B || B == 0) && (rb + delta < B + BAND_WIN || B + BAND_WIN >= w);
                    if (!fits) {
                        rebase = true;
                    } else {
                        BandRow &r = q[d];
                        float mc[BAND_PXL];
                        uint32_t lnew = 0;
                        uint32_t nonstop = 0;
#pragma unroll
                        for (int j = 0; j < BAND_PXL; j++) {
                            const int x = x0 + j;
                            const bool inband = (x >= ra && x <= rb);
                            float outm = r.mo[j];
                            int outl = (int8_t) (r.lo >> (8 * j));
                            if (inband) {
                                const int dlo = max(-x, -delta), dhi = min(w - 1 - x, delta);
                                const float rfact = c.rig ? r.rf[j] : 1.0f;
                                const int li = x - B;
                                float best = pr[li + dlo];
                                if (p.use_rig) best = __fadd_rn(best, __fmul_rn(rfact, p.rigmap[dlo + delta]));
                                int bdx = dlo;
                                for (int dx = dlo + 1; dx <= dhi; dx++) {
                                    float cand = pr[li + dx];
                                    if (p.use_rig) cand = __fadd_rn(cand, __fmul_rn(rfact, p.rigmap[dx + delta]));
                                    if (cand < best || (cand == best && lr)) { best = cand; bdx = dx; }
                                }
                                float nm = __fadd_rn(r.e[j], best);
                                bool stop = (outl == bdx) && ((double) fabsf(__fsub_rn(r.mo[j], nm)) < 1e-5);
                                if (!stop) { outm = nm; nonstop |= 1u << j; }
                                outl = bdx;
                            }
                            mc[j] = outm;
                            lnew |= (uint32_t) (uint8_t) (int8_t) outl << (8 * j);
                        }
                        // all lanes have read pr[] for this row: overwrite it with row y
                        __builtin_amdgcn_wave_barrier();
                        *(float4 *) (pr + lane * BAND_PXL) = make_float4(mc[0], mc[1], mc[2], mc[3]);
                        {
                            size_t o = (size_t) y * stride + x0;
                            if (x0 < stride) {
                                { f32x4 t4 = {mc[0], mc[1], mc[2], mc[3]}; *(GLOBAL_AS f32x4 *) (c.m + o) = t4; }
                                *(gu32 *) (c.least + o) = lnew;
                            }
                        }
                        // halo of pr (parents outside the window never matter: see `fits`)
                        // ---- shrink the band: leading run of stops advances a, trailing run pulls b back
                        unsigned long long bal = __ballot(nonstop != 0);
                        if (bal == 0ull) {
                            a = rb + 1; b = ra;
                        } else {
                            int fl = __ffsll((long long) bal) - 1;
                            int ll = 63 - __clzll((long long) bal);
                            uint32_t mf = (uint32_t) __shfl((int) nonstop, fl), ml = (uint32_t) __shfl((int) nonstop, ll);
                            int first = B + fl * BAND_PXL + (__ffs((int) mf) - 1);
                            int last = B + ll * BAND_PXL + (31 - __clz((int) ml));
                            a = first;
                            b = (last == rb) ? rb : last + 1;
                        }
                        __builtin_amdgcn_wave_barrier();
                        load_row(q[d], y + BAND_PF);
                        y++;
                    }
                }
            }
        }
    }
    if (lane == 0) c.flags[FLAG_OVF_ROW] = ovf;
    // band state for the continuation is not needed: the full-width sweep applies the rule everywhere
}

// ---------------------------------------------------------------------------
// E9 update_mmap, band form, delta_x == 1 fast path (the plug-in default).
//
// liblqr walks a band [x_min, x_max] down the image and applies, to every pixel
// of the band, "recompute (best parent, m); keep the stale m if the parent is
// the same and |dm| < 1e-5".  Applied to a pixel whose inputs did not change the
// rule is a no-op, so any superset of the pixels with changed inputs leaves the
// same memory contents.  This kernel therefore needs no global
// band bookkeeping: a 64*PXL-pixel slot is recomputed on row y iff something
// in it (or the pixel beside it) changed on row y-1, or the carve touched it
// (changed energy / re-based parents next to the seam).
//
// One workgroup of NW waves per image; wave v owns slot v of a window of NW
// slots and lane L the PXL consecutive pixels x = B + 64*PXL*v + PXL*L.  The
// 3-neighbour window of the previous row lives in registers (inside a lane
// directly, across lanes by DPP wave shifts, across waves through one 16-byte LDS
// record per wave), so a row costs the active waves ~100 instructions and all
// waves one s_barrier.  Rows are prefetched in batches of R rows into a register
// ping-pong (the next batch is in flight while this one is processed).  The window
// follows the seam: it is re-centred at batch boundaries when the dirty slots or
// the seam come within one slot of its ends; if the dirty region is wider than
// the window the kernel records the row in flags[FLAG_OVF_ROW] and the full-width
// sweep (k_dp_sweep<UPDATE>) finishes from there with identical results.
// ---------------------------------------------------------------------------
template <int PXL> struct PxVec;
template <> struct PxVec<1> { typedef float F __attribute__((ext_vector_type(1))); typedef uint8_t L; };
template <> struct PxVec<2> { typedef float F __attribute__((ext_vector_type(2))); typedef uint16_t L; };
template <> struct PxVec<4> { typedef f32x4 F; typedef uint32_t L; };

struct BandEdge {          // what a wave publishes about the row it just finished
    float first_val;       // m of its first pixel (lane 0)
    float last_val;        // m of its last pixel (lane 63)
    int flags;             // bit0: first pixel changed, bit1: last pixel changed, bit2: anything changed
    int pad;
};

template <int PXL, int NW, int R, bool LR, bool RIG>
__global__ __launch_bounds__(64 * NW) void k_band_update_mw(const DevCarver *cs, DpK p, int w, int h, int stride)
{
    typedef typename PxVec<PXL>::F FV;
    typedef typename PxVec<PXL>::L LV;
    typedef GLOBAL_AS FV GFV;
    typedef GLOBAL_AS LV GLV;
    const GCarver c = gview(cs[blockIdx.x]);
    extern __shared__ int s_touch[];                  // [h] packed (t0 | t1 << 16): pixels the carve touched on row y
    __shared__ __attribute__((aligned(16))) BandEdge s_edge[2][NW + 2];     // [row parity][wave + 1], sentinels at both ends
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float INF = __int_as_float(0x7f800000);
    constexpr int SLOT = 64 * PXL, WIN = SLOT * NW;
    const float rig_l = p.rigmap[0], rig_r = p.rigmap[2];
    const int y_start = 1;

    // pixels of row y whose inputs the carve changed: energy (liblqr's update_emap interval)
    // and parent sets next to the seam; a superset is fine
    for (int y = tid; y < h; y += 64 * NW) {
        const int v0 = c.seam_x[y], vm = c.seam_x[max(y - 1, 0)], vp = c.seam_x[min(y + 1, h - 1)];
        const int t0 = max(min(min(v0, vm), vp) - 2, 0), t1 = min(max(max(v0, vm), vp) + 1, w - 1);
        s_touch[y] = t0 | (t1 << 16);
    }
    if (tid < 2 * (NW + 2)) {
        BandEdge e; e.first_val = INF; e.last_val = INF; e.flags = 0; e.pad = 0;
        s_edge[tid / (NW + 2)][tid % (NW + 2)] = e;
    }
    {   // row 0: m = en on liblqr's interval
        const int v0 = c.seam_x[0], vp = c.seam_x[min(1, h - 1)];
        int lo = v0, hi = v0 - 1;
        if (p.radius) { lo = min(v0, vp) - 1; hi = max(v0, vp); }
        const int a = max(lo, 0), b = min(hi, w - 1);
        for (int x = a + tid; x <= b; x += 64 * NW) c.m[x] = c.en[x];
    }
    __syncthreads();
    if (h < 2) { if (tid == 0) c.flags[FLAG_OVF_ROW] = h; return; }

    const unsigned dummy = (unsigned) h * stride + PXL * tid;      // scratch row for lanes outside the image
    int y = y_start, ovf = h;
    int dirty_lo = -1, dirty_hi = -1;      // dirty slots of the last finished row, window-relative (-1: none)
    int B = 0;
    bool have_window = false;
    while (y < h) {
        // ---- (re)base the window (identical decision in every wave)
        {
            const int t = s_touch[y];
            int lo = t & 0xffff, hi = t >> 16;                       // absolute pixel range that must be inside
            if (have_window && dirty_lo >= 0) { lo = min(lo, B + SLOT * dirty_lo - 1); hi = max(hi, B + SLOT * (dirty_hi + 1)); }
            lo = max(lo, 0); hi = min(hi, w - 1);
            if (hi - lo + 1 > WIN - 2 * SLOT - 2 * (R + 2) - 8 && hi - lo + 1 < w) { ovf = y; break; }
            int nb = (((lo + hi) >> 1) - WIN / 2) & ~3;
            nb = max(0, min(nb, (w - WIN + 3) & ~3));
            B = __builtin_amdgcn_readfirstlane(nb);
            have_window = true;
            // the first and last slot must stay clean for the next R rows (same test as at the batch
            // boundaries below); if even the re-centred window cannot promise that, hand over
            const bool left_ok = (B == 0) || (lo - (R + 2) >= B + SLOT);
            const bool right_ok = (B + WIN >= w) || (hi + (R + 2) < B + WIN - SLOT);
            if (!(left_ok && right_ok)) { ovf = y; break; }
        }
        const int x0 = B + SLOT * wave + PXL * lane;          // first pixel of this lane
        const int sx0 = B + SLOT * wave;                      // first pixel of this wave's slot
        const unsigned lo_off = (unsigned) min(x0, stride - PXL);
        const bool in_img = x0 < w;
        // pixels this lane may recompute: inside the image, and not the first / last pixel of a
        // window that does not end at the image border (their outer neighbour is not in the window;
        // the window is re-centred long before a change can reach them)
        uint32_t okmask = 0;
#pragma unroll
        for (int k = 0; k < PXL; k++) {
            const int x = x0 + k;
            const bool ok = (x < w) && !(B > 0 && x == B) && !(B + WIN < w && x == B + WIN - 1);
            okmask |= ok ? (1u << k) : 0u;
        }

        // previous row: rows < y were stored by this workgroup -> make them visible, then load
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float mp[PXL];
        int par = 0;
        {
            gf32 *mrow = c.m + (size_t) (y - 1) * stride;
#pragma unroll
            for (int k = 0; k < PXL; k++)
                mp[k] = (x0 + k < w) ? __hip_atomic_load(mrow + x0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INF;
            // after a re-base every slot is recomputed once (cheap, and trivially a superset); bit 2
            // (really dirty) stays clear so that the window check below sees only real changes
            if (lane == 0) { s_edge[par][wave + 1].first_val = mp[0]; s_edge[par][wave + 1].flags = 3; }
            if (lane == 63) s_edge[par][wave + 1].last_val = mp[PXL - 1];
        }
        int own_dirty = 1;
        __syncthreads();

        FV q_mo[2][R], q_e[2][R];
        LV q_lo[2][R];
        auto issue = [&](int buf, int ybase) {           // one batch of R rows, unconditional
#pragma unroll
            for (int r = 0; r < R; r++) {
                const unsigned ro = (unsigned) min(ybase + r, h - 1) * (unsigned) stride + lo_off;
                q_mo[buf][r] = *(const GFV *) (c.m + ro);
                q_e[buf][r] = *(const GFV *) (c.en + ro);
                q_lo[buf][r] = *(const GLV *) (c.least + ro);
            }
        };
        issue(0, y);

        bool rebase = false;
        while (y < h && !rebase) {
#pragma unroll
            for (int buf = 0; buf < 2; buf++) {
                if (y < h && !rebase) {
                    // ---- batch boundary: does the window still hold the next R rows?
                    {
                        const BandEdge ef = s_edge[par][1], el = s_edge[par][NW];
                        const int t = s_touch[y];
                        const int t0 = (t & 0xffff) - (R + 2), t1 = (t >> 16) + (R + 2);
                        const bool left_ok = (B == 0) || (!(ef.flags & 4) && t0 >= B + SLOT);
                        const bool right_ok = (B + WIN >= w) || (!(el.flags & 4) && t1 < B + WIN - SLOT);
                        rebase = !(left_ok && right_ok);
                    }
                    if (rebase) {
                        // dirty slot range of the last finished row, for the re-centring
                        int lo = -1, hi = -1;
                        for (int v = 0; v < NW; v++)
                            if (s_edge[par][v + 1].flags & 4) { if (lo < 0) lo = v; hi = v; }
                        dirty_lo = __builtin_amdgcn_readfirstlane(lo);
                        dirty_hi = __builtin_amdgcn_readfirstlane(hi);
                    } else {
                        issue(buf ^ 1, y + R);            // next batch in flight while this one is processed
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            if (y < h) {
                                // what the neighbours published about row y-1
                                const BandEdge eL = s_edge[par][wave], eR = s_edge[par][wave + 2];
                                const int t = s_touch[y];
                                // no short-circuit: all LDS reads of the row are issued together (one round trip)
                                const int touch = (int) ((t & 0xffff) <= sx0 + SLOT - 1) & (int) ((t >> 16) >= sx0);
                                const bool active = (own_dirty | (eL.flags & 2) | (eR.flags & 1) | touch) != 0;
                                float mo[PXL], e[PXL], mc[PXL];
                                const uint32_t lo4 = (uint32_t) q_lo[buf][r];
#pragma unroll
                                for (int k = 0; k < PXL; k++) { mo[k] = q_mo[buf][r][k]; e[k] = q_e[buf][r][k]; }
#pragma unroll
                                for (int k = 0; k < PXL; k++) mc[k] = (x0 + k < w) ? mo[k] : INF;
                                int flags = 0;
                                if (active) {
                                    float left = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(eL.last_val), __float_as_int(mp[PXL - 1]),
                                                                                        DPP_WAVE_SHR1, 0xf, 0xf, false));
                                    float right = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(eR.first_val), __float_as_int(mp[0]),
                                                                                         DPP_WAVE_SHL1, 0xf, 0xf, false));
                                    left = (x0 == 0) ? INF : left;
                                    uint32_t lnew = 0;
                                    unsigned long long any = 0ull, chg_first = 0ull, chg_last = 0ull;
#pragma unroll
                                    for (int k = 0; k < PXL; k++) {
                                        float l = (k == 0) ? left : mp[k > 0 ? k - 1 : 0];
                                        const float cc = mp[k];
                                        float rr = (k == PXL - 1) ? right : mp[k < PXL - 1 ? k + 1 : 0];
                                        if (RIG) { l = __fadd_rn(l, rig_l); rr = __fadd_rn(rr, rig_r); }
                                        // ascending scan with strict < (LR=0: the leftmost minimum wins) or <=
                                        // (LR=1: the rightmost); missing neighbours are +inf.  Written as value
                                        // selects only (no scalar mask arithmetic on the dependency chain).
                                        const float best = fminf(fminf(l, cc), rr);
                                        int bdx;
                                        if (LR) { bdx = (cc == best) ? 0 : -1; bdx = (rr == best) ? 1 : bdx; }
                                        else { bdx = (cc == best) ? 0 : 1; bdx = (l == best) ? -1 : bdx; }
                                        const float nm = __fadd_rn(e[k], best);
                                        const int lo_k = (int) (int8_t) (lo4 >> (8 * k));
                                        // keep rule: same parent and (double) fabsf(d) < 1e-5, i.e. fabsf(d) <= 1e-5f
                                        float d = fabsf(__fsub_rn(mo[k], nm));
                                        d = (lo_k == bdx) ? d : INF;                 // parent changed: never "stop"
                                        d = ((okmask >> k) & 1) ? d : 0.0f;          // pixel not ours to recompute: never changes
                                        const bool ch = d > 1e-5f;
                                        mc[k] = ch ? nm : mc[k];
                                        const int outl = ((okmask >> k) & 1) ? bdx : lo_k;
                                        lnew |= ((uint32_t) outl & 0xffu) << (8 * k);
                                        const unsigned long long bk = __ballot(ch);
                                        any |= bk;
                                        if (k == 0) chg_first = bk;
                                        if (k == PXL - 1) chg_last = bk;
                                    }
                                    flags = (int) (chg_first & 1ull) | ((int) (chg_last >> 63) << 1) | (any ? 4 : 0);
                                    // lanes outside the image write to a scratch row
                                    const unsigned so = in_img ? (unsigned) y * (unsigned) stride + (unsigned) x0 : dummy;
                                    FV tv;
#pragma unroll
                                    for (int k = 0; k < PXL; k++) tv[k] = mc[k];
                                    *(GFV *) (c.m + so) = tv;
                                    *(GLV *) (c.least + so) = (LV) lnew;
                                }
                                own_dirty = flags & 4;
                                par ^= 1;
                                if (lane == 0) { s_edge[par][wave + 1].first_val = mc[0]; s_edge[par][wave + 1].flags = flags; }
                                if (lane == 63) s_edge[par][wave + 1].last_val = mc[PXL - 1];
#pragma unroll
                                for (int k = 0; k < PXL; k++) mp[k] = mc[k];
                                // one barrier per row: LDS only (outstanding global loads/stores keep flying)
                                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                                y++;
                            }
                        }
                    }
                }
            }
        }
    }
    if (tid == 0) c.flags[FLAG_OVF_ROW] = ovf;
}

// ---------------------------------------------------------------------------
// One DP row for a lane's 4 consecutive pixels (shared by the halo kernels below): E5's
// recurrence and, with UPDATE, E9's keep-rule.  mp = the row above (this lane's pixels), left /
// right = its neighbours' adjacent pixels.  Everything on the row's dependency chain is VALU:
// the back pointer is produced directly as a byte in place ((dx & 0xff) << 8k: two selects of
// constants), the four are OR-ed, and "same parent as before" is a byte compare of old ^ new.
// MASK: some of the lane's pixels may lie outside the image (they become +inf).
// left / right come by DPP wave shifts with bound_ctrl: lane 0's left and lane 63's right neighbour read as 0.  In the
// halo kernels those two lanes are the outermost halo columns, whose values are allowed to be wrong from the first row
// of a block on (the error moves inwards one column per row, which is what the halo width pays for), so no register has
// to be preset with +inf for them; the image's own borders are handled by MASK, not by the shift.
// ch[k] (UPDATE): the pixel's (m, back pointer) pair changed.
// ---------------------------------------------------------------------------
template <bool LR, bool RIG, bool UPDATE, bool MASK>
__device__ __forceinline__ void dp_row4(const float (&mp)[4], const float left, const float right, const f32x4 e, const f32x4 mo, const uint32_t lo4,
                                        const bool (&in)[4], const float rig_l, const float rig_r, float (&mc)[4], uint32_t &lnew, bool (&ch)[4])
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float INF = __int_as_float(0x7f800000);
    float best[4];
    uint32_t sel[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float l = (k == 0) ? left : mp[k > 0 ? k - 1 : 0];
        const float cc = mp[k];
        float rr = (k == 3) ? right : mp[k < 3 ? k + 1 : 0];
        if (RIG) { l = __fadd_rn(l, rig_l); rr = __fadd_rn(rr, rig_r); }
        // ascending scan with strict < (LR=0: the leftmost minimum wins) or <= (LR=1: the rightmost)
        best[k] = fminf(fminf(l, cc), rr);
        const uint32_t minus = 0xffu << (8 * k), plus = 0x01u << (8 * k);
        if (LR) { sel[k] = (cc == best[k]) ? 0u : minus; sel[k] = (rr == best[k]) ? plus : sel[k]; }
        else { sel[k] = (cc == best[k]) ? 0u : plus; sel[k] = (l == best[k]) ? minus : sel[k]; }
    }
    // the four sums and the four differences as two packed operations each on the pixel pairs (0, 1) and (2, 3): e and mo
    // sit in aligned register pairs as loaded, so v_pk_add_f32 takes them where they are (left to itself the compiler pairs
    // pixels 1 and 2 and pays four v_mov per row for it).  Individually rounded IEEE adds, as __fadd_rn / __fsub_rn.
    const f32x2 nm01 = (f32x2) {e[0], e[1]} + (f32x2) {best[0], best[1]}, nm23 = (f32x2) {e[2], e[3]} + (f32x2) {best[2], best[3]};
    const float nm[4] = {nm01[0], nm01[1], nm23[0], nm23[1]};
    lnew = (sel[0] | sel[1]) | (sel[2] | sel[3]);
    if (UPDATE) {
        const uint32_t diff = lo4 ^ lnew;
        const f32x2 d01 = (f32x2) {mo[0], mo[1]} - nm01, d23 = (f32x2) {mo[2], mo[3]} - nm23;
        const float dd[4] = {d01[0], d01[1], d23[0], d23[1]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // keep the stale value iff same parent and (double) fabsf(d) < 1e-5, i.e. fabsf(d) <= 1e-5f
            float d = fabsf(dd[k]);
            d = ((diff >> (8 * k)) & 0xffu) ? INF : d;
            ch[k] = d > 1e-5f;
            const float v = ch[k] ? nm[k] : mo[k];
            mc[k] = (!MASK || in[k]) ? v : INF;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) mc[k] = (!MASK || in[k]) ? nm[k] : INF;
    }
}

// dp_row4's arithmetic for PX (2 or 4) consecutive pixels per lane; lo / lnew hold PX back-pointer bytes.  With 2 pixels
// per lane a row is ~33 instructions per wave instead of ~58, and twice as many waves cover the columns.
template <int PX, bool LR, bool RIG, bool UPDATE, bool MASK>
__device__ __forceinline__ void dp_row(const float (&mp)[PX], const float left, const float right, const float (&e)[PX], const float (&mo)[PX],
                                       const uint32_t lo, const bool (&in)[PX], const float rig_l, const float rig_r, float (&mc)[PX],
                                       uint32_t &lnew, bool (&ch)[PX])
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    static_assert(PX % 2 == 0, "pixel pairs");
    const float INF = __int_as_float(0x7f800000);
    float best[PX], nm[PX], dd[PX];
    uint32_t sel[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) {
        float l = (k == 0) ? left : mp[k > 0 ? k - 1 : 0];
        const float cc = mp[k];
        float rr = (k == PX - 1) ? right : mp[k < PX - 1 ? k + 1 : 0];
        if (RIG) { l = __fadd_rn(l, rig_l); rr = __fadd_rn(rr, rig_r); }
        best[k] = fminf(fminf(l, cc), rr);
        const uint32_t minus = 0xffu << (8 * k), plus = 0x01u << (8 * k);
        if (LR) { sel[k] = (cc == best[k]) ? 0u : minus; sel[k] = (rr == best[k]) ? plus : sel[k]; }
        else { sel[k] = (cc == best[k]) ? 0u : plus; sel[k] = (l == best[k]) ? minus : sel[k]; }
    }
    // sums and differences as packed operations on the pixel pairs (2j, 2j + 1), as dp_row4
#pragma unroll
    for (int j = 0; j < PX / 2; j++) {
        const f32x2 s2 = (f32x2) {e[2 * j], e[2 * j + 1]} + (f32x2) {best[2 * j], best[2 * j + 1]};
        nm[2 * j] = s2[0]; nm[2 * j + 1] = s2[1];
        if (UPDATE) {
            const f32x2 d2 = (f32x2) {mo[2 * j], mo[2 * j + 1]} - s2;
            dd[2 * j] = d2[0]; dd[2 * j + 1] = d2[1];
        }
    }
    lnew = sel[0];
#pragma unroll
    for (int k = 1; k < PX; k++) lnew |= sel[k];
    const uint32_t diff = lo ^ lnew;
#pragma unroll
    for (int k = 0; k < PX; k++) {
        float v = nm[k];
        if (UPDATE) {
            // keep the stale value iff same parent and (double) fabsf(d) < 1e-5, i.e. fabsf(d) <= 1e-5f
            float d = fabsf(dd[k]);
            d = ((diff >> (8 * k)) & 0xffu) ? INF : d;
            ch[k] = d > 1e-5f;
            v = ch[k] ? v : mo[k];
        }
        mc[k] = (!MASK || in[k]) ? v : INF;
    }
}
// The same row for delta_x = DELTA (2 * DELTA + 1 candidate parents) and / or with a rigidity mask (RIGM: the rigidity
// term of pixel k is rf[k] * rg[dx + DELTA], liblqr's rigidity_mask * rigidity_map).  nl[i] / nr[i]: the row above at the
// lane's first pixel - 1 - i / last pixel + 1 + i.  The parent is found by liblqr's ascending scan dx = -DELTA .. DELTA
// with strict < (LR = 0: the leftmost minimum wins) or <= (LR = 1: the rightmost), written as compare-and-select;
// candidates outside the image are +inf and never win against the pixel straight above.  As in the delta_x = 1 rows the
// rigidity term of dx = 0 (zero by construction of the table) is not added.
template <int PX, int DELTA, bool LR, bool RIG, bool RIGM, bool UPDATE, bool MASK>
__device__ __forceinline__ void dp_row_g(const float (&mp)[PX], const float (&nl)[DELTA], const float (&nr)[DELTA], const float (&e)[PX],
                                         const float (&mo)[PX], const uint32_t lo, const bool (&in)[PX], const float (&rg)[2 * DELTA + 1],
                                         const float (&rf)[PX], float (&mc)[PX], uint32_t &lnew, bool (&ch)[PX])
{
    static_assert(DELTA <= 2 * PX, "the two neighbouring lanes hold the whole reach");
    const float INF = __int_as_float(0x7f800000);
    float nm[PX];
    lnew = 0;
#pragma unroll
    for (int k = 0; k < PX; k++) {
        float best = 0.0f;
        int bdx = 0;
#pragma unroll
        for (int dx = -DELTA; dx <= DELTA; dx++) {
            const int j = k + dx;
            float v = j < 0 ? nl[j < 0 ? -j - 1 : 0] : j >= PX ? nr[j >= PX ? j - PX : 0] : mp[j >= 0 && j < PX ? j : 0];
            if (RIG && dx != 0) v = __fadd_rn(v, RIGM ? __fmul_rn(rf[k], rg[dx + DELTA]) : rg[dx + DELTA]);
            if (dx == -DELTA) { best = v; bdx = dx; }
            else {
                const bool take = LR ? (v <= best) : (v < best);
                best = take ? v : best;
                bdx = take ? dx : bdx;
            }
        }
        nm[k] = __fadd_rn(e[k], best);
        lnew |= ((uint32_t) bdx & 0xffu) << (8 * k);
    }
    const uint32_t diff = lo ^ lnew;
#pragma unroll
    for (int k = 0; k < PX; k++) {
        float v = nm[k];
        if (UPDATE) {
            float d = fabsf(__fsub_rn(mo[k], v));
            d = ((diff >> (8 * k)) & 0xffu) ? INF : d;
            ch[k] = d > 1e-5f;
            v = ch[k] ? v : mo[k];
        }
        mc[k] = (!MASK || in[k]) ? v : INF;
    }
}
// PX floats / PX back-pointer bytes of one lane, as one load or store
template <int PX> struct LaneVec;
template <> struct LaneVec<2> { typedef float F __attribute__((ext_vector_type(2))); typedef uint16_t L; };
template <> struct LaneVec<4> { typedef f32x4 F; typedef uint32_t L; };

// ---------------------------------------------------------------------------
// E9 update_mmap, band form, "trapezoid waves" (delta_x == 1, no rigidity mask).
//
// k_band_update_mw pays one s_barrier and one LDS exchange per ROW (~0.49 us per row, of which the
// recompute itself is a third).  Here the waves of a window exchange once per BATCH of 16 rows:
// a slot is 256 columns (4 px per lane) of which the middle 224 are its own and 16 on each side
// are halo, recomputed redundantly from the same inputs as the neighbouring slot does -- after r
// rows the outer r halo columns are wrong, the own columns never are.  At a batch boundary every
// slot leaves the last row of its own columns in LDS (s_row) and picks up own + halo from there.
// Each slot is served by two waves that take turns batch by batch (as in k_dp_tile_p): while one
// computes, the other's 48 loads for the next batch are in flight.
//
// In place: a slot's halo columns are its neighbour's own columns, which the neighbour overwrites.
// A wave therefore waits for its prefetched batch BEFORE the barrier that opens that batch; nobody
// stores rows of a batch before that barrier.
//
// As in k_band_update_mw there is no band bookkeeping (section 4.4: any superset of the pixels with
// changed inputs gives liblqr's memory): a slot recomputes a batch iff the pixels changed on the
// row above the batch, or touched by the carve on the batch's rows, are within 16 columns of it.
// The window (NW slots) is re-centred when those pixels come within 16 columns of its ends; if
// they do not fit the kernel records the row in flags[FLAG_OVF_ROW] and the full-width sweep
// finishes from there.
// ---------------------------------------------------------------------------
constexpr int TW_R = 16;                     // rows per batch = halo columns
constexpr int TW_OWN = 256 - 2 * TW_R;       // own columns per slot
// Within an active slot only the lanes near the changes are staged, stored and handed over: pixels further than R
// columns from the changed ones cannot change during a batch; lanes that were not staged compute garbage, which moves
// inwards one column per row, so staging reaches R (kept lanes) + R (rows) + 8 (a kept lane's own four columns, slack).
// A vector-memory instruction costs the CU's memory path ~12 + 0.7 cycles per ACTIVE lane (scripts/dbg/t_ta.hip),
// and that path is this kernel's bound.
constexpr int TW_LANE_MARGIN = 2 * TW_R + 8;

template <int NW, bool LR, bool RIG>
__device__ __forceinline__ void band_update_tw_body(const DevCarver &dc, const DpK &p, int w, int h, int stride, int *dev_err)
{
    constexpr int R = TW_R, OWN = TW_OWN, WIN = NW * OWN, NT = 128 * NW;
    const GCarver c = gview(dc);
    extern __shared__ int s_tw[];                          // [2h]: per row, per batch-starting-at-row touch ranges
    int *s_touch = s_tw, *s_touchR = s_tw + h;
    // m of the last finished row over [B-R, B+WIN+R), double-buffered by batch parity: a slot reads its halo
    // (the neighbours' own columns) at the start of a batch, and a neighbour that is a whole batch faster
    // must not have overwritten them yet
    __shared__ __attribute__((aligned(16))) float s_row[2][WIN + 2 * R];
    __shared__ int s_rec[2][NW][2];                        // [batch parity][slot] {lo, hi}: px changed on that row (lo > hi: none)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wv % NW, par_w = wv / NW;             // the two waves of a slot sit on the same SIMD
    const float INF = __int_as_float(0x7f800000);
    const float rig_l = p.rigmap[0], rig_r = p.rigmap[2];

    // pixels of row y whose inputs the carve changed (see k_band_update_mw), then the union over the
    // 16 rows of a batch starting at y
    for (int y = tid; y < h; y += NT) {
        const int v0 = c.seam_x[y], vm = c.seam_x[max(y - 1, 0)], vp = c.seam_x[min(y + 1, h - 1)];
        const int t0 = max(min(min(v0, vm), vp) - 2, 0), t1 = min(max(max(v0, vm), vp) + 1, w - 1);
        s_touch[y] = t0 | (t1 << 16);
    }
    {   // row 0: m = en on liblqr's interval
        const int v0 = c.seam_x[0], vp = c.seam_x[min(1, h - 1)];
        int lo = v0, hi = v0 - 1;
        if (p.radius) { lo = min(v0, vp) - 1; hi = max(v0, vp); }
        const int a = max(lo, 0), b = min(hi, w - 1);
        for (int x = a + tid; x <= b; x += NT) c.m[x] = c.en[x];
    }
    __syncthreads();
    for (int y = tid; y < h; y += NT) {
        int t0 = 0xffff, t1 = 0;
        for (int r = 0; r < R; r++) {
            const int t = s_touch[min(y + r, h - 1)];
            t0 = min(t0, t & 0xffff); t1 = max(t1, t >> 16);
        }
        s_touchR[y] = t0 | (t1 << 16);
    }
    __syncthreads();
    if (h < 2) { if (tid == 0) c.flags[FLAG_OVF_ROW] = h; return; }

    f32x4 q_e[R], q_mo[R];
    uint32_t q_lo[R];
    int B = 0;
    // full = false: the slot cannot become active in that batch (see the prediction at the issue site);
    // only the row it hands over is needed.  The CU's vector-memory path takes ~16 cycles per 64-lane
    // 16-byte access, so four slots' 3 loads + 2 stores per row (320 cycles) would be the bound.
    // Addresses as uniform plane base + 32-bit lane offset (global_load ... v_off, s[base]): per row one scalar
    // multiply and two VALU adds for the three loads.  With 64-bit per-lane addresses the 48 loads of a batch cost
    // ~2600 cycles of issue (measured), on the SIMD the partner wave is computing on.
    auto issue = [&](int ybase, bool full) {        // full: per LANE (see TW_LANE_MARGIN)
        const int x0 = B + OWN * slot - R + 4 * lane;
        const unsigned lo_off = (unsigned) min(max(x0, 0), stride - 4);
        if (full) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const unsigned row = (unsigned) min(ybase + r, h - 1) * (unsigned) stride;
                const unsigned ro = row + lo_off, ro4 = (row << 2) + (lo_off << 2);
                q_e[r] = *(const GLOBAL_AS f32x4 *) ((const gu8 *) c.en + ro4);
                q_mo[r] = *(const GLOBAL_AS f32x4 *) ((const gu8 *) c.m + ro4);
                q_lo[r] = *(const gu32 *) (c.least + ro);
            }
        } else {
            q_mo[R - 1] = *(const GLOBAL_AS f32x4 *) ((const gu8 *) c.m + ((((unsigned) min(ybase + R - 1, h - 1) * (unsigned) stride) + lo_off) << 2));
        }
    };
    // make the compiler wait for this wave's prefetched batch here
    auto landed = [&]() {
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("" ::"v"(q_e[r]), "v"(q_mo[r]), "v"(q_lo[r]));
    };

    int y = 1, ovf = h, kpar = 0;
    bool loads_full = true;                  // does this wave's staged batch hold all rows (or only the hand-over row)?
    bool lane_staged = true;                 // ... for this lane (a slot stages only the lanes near the changes)
    int dlo = 1 << 30, dhi = -1;             // px changed on the last finished row (absolute x)
    bool have_window = false, force_active = false, just_rebased = false;
    while (y < h) {
        // ---- does the window hold the next batch?  (identical decision in every wave)
        const int t = s_touchR[y];
        int lo = t & 0xffff, hi = t >> 16;
        if (dhi >= dlo) { lo = min(lo, dlo - 1); hi = max(hi, dhi + 1); }
        lo = max(lo, 0); hi = min(hi, w - 1);
        const bool fits = have_window && (B == 0 || lo - R >= B) && (B + WIN >= w || hi + R <= B + WIN - 1);
        bool issue_full = true, lane_all = true;     // lane_all: stage every lane (after a re-centring, or nothing known)
        int plo_l = 0, phi_l = 0;
        int y_issue = -1;                    // batch this wave prefetches at the end of the iteration (one issue site:
                                             // a second one makes the register allocator spill the staging rows)
        if (!fits) {
            if (just_rebased || (hi - lo + 1 + 2 * R + 8 > WIN && WIN < w)) { ovf = y; break; }
            // (placing the changes inside ONE slot's own columns instead of on the middle slots' boundary measured no gain, round 4)
            int nb = (((lo + hi) >> 1) - WIN / 2) & ~3;
            nb = max(0, min(nb, (w - WIN + 3) & ~3));
            B = __builtin_amdgcn_readfirstlane(nb);
            have_window = true;
            // rows < y were stored by this workgroup: make them visible, then reload the row above
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            {
                const gf32 *mrow = c.m + (size_t) (y - 1) * stride;
                for (int i = tid; i < WIN + 2 * R; i += NT) {
                    const int x = B - R + i;
                    s_row[1][i] = (x >= 0 && x < w) ? __hip_atomic_load(mrow + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INF;     // read as [kpar ^ 1]
                }
            }
            kpar = 0;
            force_active = true;             // every slot recomputes once: trivially a superset
            just_rebased = true;
            y_issue = y + par_w * R;
        } else {
        just_rebased = false;
        const int x0 = B + OWN * slot - R + 4 * lane;
        const int own_lo = B + OWN * slot, own_hi = own_lo + OWN - 1;
        const bool own_lane = lane >= R / 4 && lane < 64 - R / 4;
        if (par_w == kpar) {
            const bool active = force_active || (lo - R <= own_hi && hi + R >= own_lo);
            // the prediction below is a superset by construction; should it ever fail, say so instead of
            // computing on rows that were not loaded (the host turns the flag into LQR_ERROR)
            if (active && !loads_full && lane == 0) dev_fail(dev_err, DEVERR_BAND_PREDICTION);
            // Lane granularity of the same superset argument: pixels further than R columns from [lo, hi] cannot change
            // in this batch, so only the lanes touching [lo - R, hi + R] are kept (stored, handed over, counted); for
            // those to be right through R rows their neighbours up to TW_LANE_MARGIN columns out must have been staged.
            const bool lane_valid = force_active || (x0 + 3 >= lo - R && x0 <= hi + R);
            if (active && __any(!force_active && (x0 + 3 >= lo - TW_LANE_MARGIN && x0 <= hi + TW_LANE_MARGIN) && !lane_staged) && lane == 0)
                dev_fail(dev_err, DEVERR_BAND_PREDICTION);
            bool in[4];
#pragma unroll
            for (int k = 0; k < 4; k++) in[k] = (x0 + k >= 0) && (x0 + k < w);
            float mp[4];
            {
                const f32x4 v = *(const f32x4 *) (s_row[kpar ^ 1] + OWN * slot + 4 * lane);
                mp[0] = v[0]; mp[1] = v[1]; mp[2] = v[2]; mp[3] = v[3];
            }
            int rlo = 1 << 30, rhi = -1;
            const int nrows = min(R, h - y);
            // this wave's batch landed an iteration ago; saying so here keeps the compiler from counting
            // vmcnt down through the rows, which would make the later rows wait for the earlier rows' stores
            landed();
            // GUARD: the image ends inside the batch (last batch of a sweep only); MASK: the slot reaches over
            // the image's left or right border
            auto rows = [&](auto guard, auto mask) {
                constexpr bool GUARD = decltype(guard)::value, MASK = decltype(mask)::value;
                const bool own = own_lane && x0 < w && lane_valid;
                // running store offsets (see k_dp_tile_p).  The stores stay conditional here: four slots share this CU's
                // memory path, which is the bound (section 4.5) -- without the condition the row is 100 cycles shorter for
                // the wave and the kernel 3 % slower (the halo lanes' stores are traffic on that path)
                unsigned so = (unsigned) y * (unsigned) stride + (unsigned) x0, so4 = so * 4u;
#pragma unroll
                for (int r = 0; r < R; r++, so += (unsigned) stride, so4 += 4u * (unsigned) stride) {
                    asm volatile("" : "+v"(so), "+v"(so4));
                    if (!GUARD || r < nrows) {
                        float mc[4];
                        uint32_t lnew = 0;
                        bool ch[4];
                        const float left = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[3]), DPP_WAVE_SHR1, 0xf, 0xf, true));
                        const float right = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[0]), DPP_WAVE_SHL1, 0xf, 0xf, true));
                        dp_row4<LR, RIG, true, MASK>(mp, left, right, q_e[r], q_mo[r], q_lo[r], in, rig_l, rig_r, mc, lnew, ch);
                        if (own) {
                            u32x4 tv = {__float_as_uint(mc[0]), __float_as_uint(mc[1]), __float_as_uint(mc[2]), __float_as_uint(mc[3])};
                            *(GLOBAL_AS u32x4 *) ((gu8 *) c.m + so4) = tv;
                            *(gu32 *) (c.least + so) = lnew;
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) mp[k] = mc[k];
                        if (r == R - 1) {
                            // extent of the changes on the batch's last row (own columns)
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const unsigned long long chm = __ballot(ch[k] && (!MASK || in[k]) && own_lane && lane_valid);
                                if (chm) {
                                    const int first = __builtin_ctzll(chm), last = 63 - __builtin_clzll(chm);
                                    rlo = min(rlo, B + OWN * slot - R + 4 * first + k);
                                    rhi = max(rhi, B + OWN * slot - R + 4 * last + k);
                                }
                            }
                        }
                    }
                }
            };
            const bool interior = (x0 - 4 * lane >= 0) && (x0 - 4 * lane + 256 <= w);      // uniform per wave
            if (active) {
                // the computing wave outranks its partner's load issue on the SIMD they share (-3 % on the kernel; no
                // effect in k_dp_tile_p, whose two waves sit on different SIMDs)
                __builtin_amdgcn_s_setprio(2);
                if (nrows == R) { if (interior) rows(std::false_type{}, std::false_type{}); else rows(std::false_type{}, std::true_type{}); }
                else rows(std::true_type{}, std::true_type{});
                __builtin_amdgcn_s_setprio(0);
                if (!lane_valid && nrows == R) {
                    // unchanged by construction (and possibly computed from rows that were not staged): memory has it
#pragma unroll
                    for (int k = 0; k < 4; k++) mp[k] = in[k] ? q_mo[R - 1][k] : INF;
                }
            }
            else if (nrows == R) {
                // nothing can change in this slot during the batch: its last row is what memory holds
#pragma unroll
                for (int k = 0; k < 4; k++) mp[k] = in[k] ? q_mo[R - 1][k] : INF;
            }
            if (nrows == R) {
                // hand the last row over: own columns, and at the window's ends the halo as memory has it
                f32x4 v = {mp[0], mp[1], mp[2], mp[3]};
                const bool edge_halo = (slot == 0 && lane < R / 4) || (slot == NW - 1 && lane >= 64 - R / 4);
                if (edge_halo) {
#pragma unroll
                    for (int k = 0; k < 4; k++) v[k] = in[k] ? q_mo[R - 1][k] : INF;
                }
                if (own_lane || edge_halo) *(f32x4 *) (s_row[kpar] + OWN * slot + 4 * lane) = v;
                if (lane == 0) { s_rec[kpar][slot][0] = rlo; s_rec[kpar][slot][1] = rhi; }
                y_issue = y + 2 * R;
            }
        } else {
            landed();
        }
        // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. the prefetch
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            int a = 1 << 30, b = -1;
#pragma unroll
            for (int v = 0; v < NW; v++) { a = min(a, s_rec[kpar][v][0]); b = max(b, s_rec[kpar][v][1]); }
            dlo = __builtin_amdgcn_readfirstlane(a);
            dhi = __builtin_amdgcn_readfirstlane(b);
        }
        if (y_issue >= 0 && y_issue < h) {
            // can this slot be active in the batch it is about to prefetch (two batches down)?  Changes move
            // one column per row: whatever is dirty now, or touched in the next batch, is at most 2R columns
            // away by then; touches of that batch itself R
            int plo = 1 << 30, phi = -1;
            if (dhi >= dlo) { plo = dlo - 2 * R - 3; phi = dhi + 2 * R + 3; }
            const int t1 = s_touchR[min(y + R, h - 1)], t2 = s_touchR[y_issue];
            plo = min(plo, min((t1 & 0xffff) - 2 * R - 3, (t2 & 0xffff) - R - 3));
            phi = max(phi, max((t1 >> 16) + 2 * R + 3, (t2 >> 16) + R + 3));
            issue_full = (plo <= own_hi && phi >= own_lo);
            lane_all = false;
            plo_l = plo - (TW_LANE_MARGIN - R); phi_l = phi + (TW_LANE_MARGIN - R);
        }
        y += R;
        kpar ^= 1;
        force_active = false;
        }
        if (y_issue >= 0) {
            // per lane: whatever can be within TW_LANE_MARGIN columns of the changes when the batch is computed (plo / phi
            // already contain the R columns of the slot test)
            const int xl = B + OWN * slot - R + 4 * lane;
            lane_staged = issue_full && (lane_all || (xl + 3 >= plo_l && xl <= phi_l));
            issue(y_issue, lane_staged);
            loads_full = issue_full;
        }
        if (just_rebased) {
            landed();                        // before anybody stores rows >= y
            __syncthreads();
        }
    }
    if (tid == 0) c.flags[FLAG_OVF_ROW] = ovf;
}

template <int NW, bool LR, bool RIG>
__global__ __launch_bounds__(128 * NW) void k_band_update_tw(const DevCarver *cs, DpK p, int w, int h, int stride, int *dev_err)
{
    band_update_tw_body<NW, LR, RIG>(cs[blockIdx.x], p, w, h, stride, dev_err);
}

#ifdef LQR_BAND_EXPERIMENTS
#include "band_experiments.inc"
#endif

// ---------------------------------------------------------------------------
// E5 build_mmap, multi-CU form (delta_x == 1, no rigidity mask): trapezoid tiling of the
// dependency cone.  One launch covers DPT_ROWS rows of every image; one WAVE per tile of
// 192 own columns + 32 halo columns on each side (4 px per lane, lanes 8..55 own).  The wave
// loads row y0-1 of m over own+halo, then computes DPT_ROWS rows entirely in registers (DPP
// neighbours, no LDS, no barrier): the halo is recomputed redundantly and goes stale by one
// pixel per row from the outside in, which is exactly what 32 columns allow for 32 rows.  Only
// own columns are stored.  H/32 dependent launches instead of H barriers of one workgroup:
// a 4K sweep takes ~20 x less wall time and uses the whole chip for a batch.
// ---------------------------------------------------------------------------
#define DPT_ROWS 32
#define DPT_OWN 192
template <bool LR, bool RIG>
__global__ __launch_bounds__(64) void k_dp_tile(const DevCarver *cs, DpK p, int w, int h, int stride, int y0)
{
    const GCarver c = gview(cs[blockIdx.y]);
    const int lane = threadIdx.x;
    const float INF = __int_as_float(0x7f800000);
    const float rig_l = p.rigmap[0], rig_r = p.rigmap[2];
    const int own0 = blockIdx.x * DPT_OWN;                  // first own column of the tile
    const int x0 = own0 - 32 + 4 * lane;                    // first pixel of this lane (may be < 0 or >= w)
    const bool own = (lane >= 8 && lane < 56) && x0 < w;
    const int nrows = min(DPT_ROWS, h - y0);
    // clamped load offset: lanes outside the row read something valid and ignore it
    const unsigned lo_off = (unsigned) min(max(x0, 0), stride - 4);
    const bool lane_in = (x0 >= 0);                          // x0 is a multiple of 4: a lane is entirely in or left of the image

    float mp[4];
    if (y0 > 0) {
        const f32x4 v = *(const GLOBAL_AS f32x4 *) (c.m + (size_t) (y0 - 1) * stride + lo_off);
#pragma unroll
        for (int k = 0; k < 4; k++) mp[k] = (lane_in && x0 + k < w) ? v[k] : INF;
    }
    constexpr int R = 8;
    f32x4 q_e[2][R];
    auto issue = [&](int buf, int ybase) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const unsigned ro = (unsigned) min(ybase + r, h - 1) * (unsigned) stride + lo_off;
            q_e[buf][r] = *(const GLOBAL_AS f32x4 *) (c.en + ro);
        }
    };
    issue(0, y0);
#pragma unroll
    for (int b4 = 0; b4 < DPT_ROWS / R; b4++) {
        const int buf = b4 & 1;
        if (b4 * R < nrows) {
            if ((b4 + 1) * R < nrows) issue(buf ^ 1, y0 + (b4 + 1) * R);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int y = y0 + b4 * R + r;
                if (b4 * R + r < nrows) {
                    float mc[4];
                    uint32_t lnew = 0;
                    const f32x4 e = q_e[buf][r];
                    if (y == 0) {
#pragma unroll
                        for (int k = 0; k < 4; k++) mc[k] = (lane_in && x0 + k < w) ? e[k] : INF;     // row 0: m = en
                    } else {
                        float left = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[3]), DPP_WAVE_SHR1, 0xf, 0xf, true));
                        float right = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[0]), DPP_WAVE_SHL1, 0xf, 0xf, true));
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            float l = (k == 0) ? left : mp[k > 0 ? k - 1 : 0];
                            const float cc = mp[k];
                            float rr = (k == 3) ? right : mp[k < 3 ? k + 1 : 0];
                            if (RIG) { l = __fadd_rn(l, rig_l); rr = __fadd_rn(rr, rig_r); }
                            const float best = fminf(fminf(l, cc), rr);
                            int bdx;
                            if (LR) { bdx = (cc == best) ? 0 : -1; bdx = (rr == best) ? 1 : bdx; }
                            else { bdx = (cc == best) ? 0 : 1; bdx = (l == best) ? -1 : bdx; }
                            const float nm = __fadd_rn(e[k], best);
                            mc[k] = (lane_in && x0 + k < w) ? nm : INF;
                            lnew |= ((uint32_t) bdx & 0xffu) << (8 * k);
                        }
                    }
                    {   // no condition on the stores (k_dp_tile_p): lanes that own nothing write to the spare row
                        const unsigned so = own ? (unsigned) y * (unsigned) stride + (unsigned) x0 : (unsigned) h * (unsigned) stride + 4u * lane;
                        f32x4 t = {mc[0], mc[1], mc[2], mc[3]};
                        *(GLOBAL_AS f32x4 *) (c.m + so) = t;
                        *(gu32 *) (c.least + so) = lnew;
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) mp[k] = mc[k];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Persistent form of k_dp_tile for small batches (every tile co-resident): one launch per
// sweep, one workgroup per 128-column tile + 64-px halos, blocks of 64 rows.  Instead of a kernel
// boundary between row blocks a tile waits for its two neighbours only: each publishes "block j
// done" on its own counter after storing the block's last row write-through (sc1), and the halo
// columns of that row are re-read (agent-scope loads) by the neighbours.
//
// The rows of a tile are a serial chain (~0.1 us each) that consumes ~2.3 KB per row; hiding a
// ~2 us memory round trip takes ~20 rows in flight, but one wave can have only 63 memory
// operations outstanding (vmcnt) and each row costs five (three loads, two stores).  So the
// workgroup is two waves that take turns: wave q computes the 16-row batches q, q+2, ... and
// hands the last row over through LDS; while the other computes, its next batch's 48 loads are
// in flight (one batch time ~ 2.7 us of lead).  One s_barrier per batch.
//
// UPDATE = liblqr's update_mmap keep-rule applied to every pixel (a superset of the band,
// section 4.4), reading m / least and writing m2 / least2 (tiles overlap in their halos, so an
// in-place update would let a tile read a neighbour's half-updated (m, least) pair); the host
// tile that finishes last swaps the plane pointers in the device descriptor.
// ---------------------------------------------------------------------------
// PX pixels per lane (4, or 2 when the device has room for twice the tiles: half the instructions per wave and row):
// a tile is 64 * PX columns of which the 16 outer lanes on each side are halo
constexpr int dpp_halo(int px) { return 16 * px; }              // halo columns on each side = rows per block
constexpr int dpp_own(int px) { return 64 * px - 2 * dpp_halo(px); }     // columns a tile owns
constexpr int dpp_ex_tile(int px) { return 2 * 2 * dpp_halo(px); }       // granules a tile publishes: [block parity][side: 0 to the left, 1 to the right][column]
constexpr int dpp_rb(int px, int delta) { return delta >= 3 ? 8 : dpp_halo(px) / delta; }      // rows per block
constexpr int DPP_R = 16;                       // rows per batch
constexpr int DPP_W = 2;                        // waves taking turns
static_assert(dpp_halo(2) % (2 * DPP_R) == 0 && dpp_halo(4) % (2 * DPP_R) == 0, "a block (halo / delta_x rows, delta_x <= 2) is a whole number of batches");
constexpr int DPP_BLK_BITS = 12;                // bits of the block index in a granule's tag
// co-residency bound for the spin waits, set from the occupancy query in lqrhip_init (dpp_resident_workgroups)
static int g_dpp_max_wgs = 0;
static int g_dpp_max_wgs_plain = 0, g_dpp_max_wgs_general = 0;      // ... of the plain / the delta_x = 2, rigidity-mask instantiations
static int g_dpp_max_wgs_px4 = 0;
static int g_dpp_max_wgs_tiles = 0;                                 // ... of k_band_tiles                                   // ... of the plain 4-px instantiations alone (fewer registers than the 2-px ones)


// DELTA = delta_x (1 or 2: errors move DELTA columns per row, so a block is HALO / DELTA rows); RIGM = a rigidity mask
// scales the rigidity term per pixel (one more 4-byte plane read).  The plain instantiations (1, false) use the
// 3-neighbour row above, the others dp_row_g.
#ifdef LQR_TILE_TIMING
__device__ unsigned long long g_tile_dbg[2 * 16];
#define TT(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tdbg[i] += t__ - ttprev; ttprev = t__; } while (0)
#else
#define TT(i) do { } while (0)
#endif
template <int PX, bool LR, bool RIG, bool UPDATE, int DELTA = 1, bool RIGM = false>
__global__ __launch_bounds__(64 * DPP_W) void k_dp_tile_p(DevCarver *cs, DpK p, int w, int h, int stride, unsigned long long *exch, int epoch, int *dev_err)
{
    static_assert(DELTA >= 1 && DELTA <= 4 && DELTA <= 2 * PX && (RIG || !RIGM), "delta_x 1 .. 4; a rigidity mask only matters with rigidity");
    typedef typename LaneVec<PX>::F FV;
    typedef typename LaneVec<PX>::L LV;
    typedef GLOBAL_AS FV GFV;
    typedef GLOBAL_AS LV GLV;
    constexpr int HALO = dpp_halo(PX), OWN = dpp_own(PX), EX_TILE = dpp_ex_tile(PX), TILE = 64 * PX, HL = 16;      // HL: halo lanes per side
    __shared__ FV s_mp[64];                      // the row above the next batch, handed from wave to wave
    __shared__ int s_fail;                       // a neighbour never showed up: both waves leave at the next barrier
    __shared__ volatile int s_polled;            // last block whose halo wave 0 has received
    if (threadIdx.x == 0) { s_fail = 0; s_polled = 0; }
    __syncthreads();
    const GCarver c = gview(cs[blockIdx.y]);
    gf32 *m_out = UPDATE ? c.m2 : c.m;
    gi8 *least_out = UPDATE ? c.least2 : c.least;
    const int ntiles = gridDim.x, tile = blockIdx.x;
    // exchange area of this image: per tile EX_TILE granules ({m bits, tag}, 8 bytes, one store each), then one
    // word that counts finished tiles
    typedef GLOBAL_AS unsigned long long gu64;
    gu64 *ex_img = (gu64 *) exch + (size_t) blockIdx.y * ((size_t) ntiles * EX_TILE + 8);
    gi32 *done_ctr = (gi32 *) (ex_img + (size_t) ntiles * EX_TILE);
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float INF = __int_as_float(0x7f800000);
    const float rig_l = p.rigmap[0], rig_r = p.rigmap[2];
    const int x0 = tile * OWN - HALO + PX * lane;           // first pixel of this lane (may be < 0 or >= w)
    const bool own_lane = lane >= HL && lane < 64 - HL;
    const bool own = own_lane && x0 < w;
    const unsigned lo_off = (unsigned) min(max(x0, 0), stride - PX);
    const bool lane_in = (x0 >= 0);
    bool in[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) in[k] = lane_in && x0 + k < w;

#ifndef DPP_RIGM32
#define DPP_RIGM32 1          // the rigidity-mask instantiation too (256 VGPRs, no spill): config 5 with a mask 46.8 -> 50.2 k
#endif
#ifndef DPP_R2
#define DPP_R2 32
#endif
    // Rows per batch.  The plain 2-px instantiations take a whole 32-row block per turn (the two waves alternate block by
    // block, as the delta_x = 2 instantiations do with their 16-row blocks): half the hand-overs, barriers and loop
    // iterations of 16-row batches for 80 more staging registers (193 VGPRs, no spill; the residency bound is queried per
    // instantiation).  Measured on one box: 4K 20.0 -> 21.85 k, FHD 12.3 -> 13.1 k, config 5 50.9 -> 55.9 k.
    constexpr int R = (PX == 2 && DELTA == 1 && (!RIGM || DPP_RIGM32)) ? DPP_R2 : DELTA >= 3 ? 8 : DPP_R;      // delta_x 3, 4: 8-row blocks (errors move up to 4 columns per row)
    FV q_e[R], q_mo[R], q_rf[RIGM ? R : 1];
    LV q_lo[R];
    constexpr int RB = dpp_rb(PX, DELTA), NBB = RB / R;          // rows, batches per block
    static_assert(RB * DELTA <= HALO && RB % R == 0, "a block's errors stay inside the halo");
    float rg[2 * DELTA + 1];
#pragma unroll
    for (int i = 0; i < 2 * DELTA + 1; i++) rg[i] = p.rigmap[i];
    auto issue = [&](int ybase) {      // uniform plane base + 32-bit lane offset, as in k_band_update_tw
#pragma unroll
        for (int r = 0; r < R; r++) {
            const unsigned row = (unsigned) min(ybase + r, h - 1) * (unsigned) stride;
            const unsigned ro = row + lo_off, ro4 = (row << 2) + (lo_off << 2);
            q_e[r] = *(const GFV *) ((const gu8 *) c.en + ro4);
            if (RIGM) q_rf[r] = *(const GFV *) ((const gu8 *) c.rig + ro4);
            if (UPDATE) {
                q_mo[r] = *(const GFV *) ((const gu8 *) c.m + ro4);
                q_lo[r] = *(const GLV *) (c.least + ro);
            }
        }
    };
    float mp[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) mp[k] = INF;
    // GUARD: the batch may contain row 0 or rows past the image (first and last batch of a sweep);
    // MASK: the tile reaches over the image's left or right border
    auto batch = [&](int ybase, auto guard, auto mask) {
        constexpr bool GUARD = decltype(guard)::value, MASK = decltype(mask)::value;
        // store offsets run down the rows in two VGPRs (elements for the byte plane, bytes for m): one v_add each per
        // row instead of a scalar multiply + two adds; opaque to the compiler so that it keeps them that way.
        // The stores carry no condition: `if (own)` costs s_and_saveexec + s_cbranch_execz + s_or exec per row, which a
        // lone wave pays with ~90 of its ~350 cycles per row (scripts/dbg/t_row.hip: 352 -> 250 cycles).  Lanes that own
        // nothing (halo, beyond the image) write to the spare row below the image instead, their offsets standing still.
        const unsigned inc = own ? (unsigned) stride : 0u, inc4 = inc * 4u;
        unsigned so = own ? (unsigned) ybase * (unsigned) stride + (unsigned) x0 : (unsigned) h * (unsigned) stride + (unsigned) (PX * lane), so4 = so * 4u;
#pragma unroll
        for (int r = 0; r < R; r++, so += inc, so4 += inc4) {
            const int y = ybase + r;
            asm volatile("" : "+v"(so), "+v"(so4));
            if (!GUARD || y < h) {
                float mc[PX], e[PX], mo[PX];
                uint32_t lnew = 0;
#pragma unroll
                for (int k = 0; k < PX; k++) { e[k] = q_e[r][k]; mo[k] = UPDATE ? q_mo[r][k] : 0.0f; }
                if (GUARD && y == 0) {
#pragma unroll
                    for (int k = 0; k < PX; k++) mc[k] = in[k] ? e[k] : INF;     // row 0: m = en
                } else {
                    bool ch[PX];
                    if constexpr (DELTA == 1 && !RIGM) {
                        const float left = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[PX - 1]), DPP_WAVE_SHR1, 0xf, 0xf, true));
                        const float right = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[0]), DPP_WAVE_SHL1, 0xf, 0xf, true));
                        dp_row<PX, LR, RIG, UPDATE, MASK>(mp, left, right, e, mo, UPDATE ? (uint32_t) q_lo[r] : 0u, in, rig_l, rig_r, mc, lnew, ch);
                    } else {
                        // the neighbouring lanes' pixels next to this lane's: DELTA on each side (DELTA <= PX)
                        float nl[DELTA], nr[DELTA], rf[PX];
#pragma unroll
                        for (int i = 0; i < DELTA; i++) {       // pixel i % PX of the lane i / PX + 1 away: one wave shift per lane
                            int a = __builtin_amdgcn_mov_dpp(__float_as_int(mp[PX - 1 - i % PX]), DPP_WAVE_SHR1, 0xf, 0xf, true);
                            int b2 = __builtin_amdgcn_mov_dpp(__float_as_int(mp[i % PX]), DPP_WAVE_SHL1, 0xf, 0xf, true);
                            if (i >= PX) { a = __builtin_amdgcn_mov_dpp(a, DPP_WAVE_SHR1, 0xf, 0xf, true); b2 = __builtin_amdgcn_mov_dpp(b2, DPP_WAVE_SHL1, 0xf, 0xf, true); }
                            nl[i] = __int_as_float(a);
                            nr[i] = __int_as_float(b2);
                        }
#pragma unroll
                        for (int k = 0; k < PX; k++) rf[k] = RIGM ? q_rf[RIGM ? r : 0][k] : 1.0f;
                        dp_row_g<PX, DELTA, LR, RIG, RIGM, UPDATE, MASK>(mp, nl, nr, e, mo, UPDATE ? (uint32_t) q_lo[r] : 0u, in, rg, rf, mc, lnew, ch);
                    }
                }
                {
                    FV t;
#pragma unroll
                    for (int k = 0; k < PX; k++) t[k] = mc[k];
                    *(GFV *) ((gu8 *) m_out + so4) = t;
                    *(GLV *) (least_out + so) = (LV) lnew;
                }
#pragma unroll
                for (int k = 0; k < PX; k++) mp[k] = mc[k];
            }
        }
    };
    // UPDATE: the batch's results stay in the registers that held its inputs (m over q_mo[r], back pointers over q_lo[r]) and
    // are stored after the barrier, in the phase in which this wave used to issue only its next prefetch -- the row loop
    // issues no memory instruction (the two stores and their address updates were 50 of a 2-px row's 194 cycles).  ONE
    // instantiation of the loop (it rewrites the staging registers; a second copy makes the compiler keep two sets of
    // them): a batch that runs past the image computes its surplus rows from re-read copies of the last row and does
    // not store them; pixels outside the image get an energy AND an old value of +inf once per batch, which makes their
    // m +inf on every row (see the call site).
    auto batch_u = [&](int ybase) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            float mc[PX], e[PX], mo[PX];
            uint32_t lnew = 0;
            bool ch[PX];
#pragma unroll
            for (int k = 0; k < PX; k++) { e[k] = q_e[r][k]; mo[k] = q_mo[r][k]; }
            if constexpr (DELTA == 1 && !RIGM) {
                const float left = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[PX - 1]), DPP_WAVE_SHR1, 0xf, 0xf, true));
                const float right = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[0]), DPP_WAVE_SHL1, 0xf, 0xf, true));
                dp_row<PX, LR, RIG, true, false>(mp, left, right, e, mo, (uint32_t) q_lo[r], in, rig_l, rig_r, mc, lnew, ch);
            } else {
                float nl[DELTA], nr[DELTA], rf[PX];
#pragma unroll
                for (int i = 0; i < DELTA; i++) {       // pixel i % PX of the lane i / PX + 1 away: one wave shift per lane
                    int a = __builtin_amdgcn_mov_dpp(__float_as_int(mp[PX - 1 - i % PX]), DPP_WAVE_SHR1, 0xf, 0xf, true);
                    int b2 = __builtin_amdgcn_mov_dpp(__float_as_int(mp[i % PX]), DPP_WAVE_SHL1, 0xf, 0xf, true);
                    if (i >= PX) { a = __builtin_amdgcn_mov_dpp(a, DPP_WAVE_SHR1, 0xf, 0xf, true); b2 = __builtin_amdgcn_mov_dpp(b2, DPP_WAVE_SHL1, 0xf, 0xf, true); }
                    nl[i] = __int_as_float(a);
                    nr[i] = __int_as_float(b2);
                }
#pragma unroll
                for (int k = 0; k < PX; k++) rf[k] = RIGM ? q_rf[RIGM ? r : 0][k] : 1.0f;
                dp_row_g<PX, DELTA, LR, RIG, RIGM, true, false>(mp, nl, nr, e, mo, (uint32_t) q_lo[r], in, rg, rf, mc, lnew, ch);
            }
            if (r == 0 && ybase == 0) {          // row 0: m = en, whatever stood there (update_mmap's first row)
#pragma unroll
                for (int k = 0; k < PX; k++) mc[k] = e[k];
                lnew = 0;
            }
#pragma unroll
            for (int k = 0; k < PX; k++) { mp[k] = mc[k]; q_mo[r][k] = mc[k]; }
            q_lo[r] = (LV) lnew;
        }
    };
    auto store_u = [&](int ybase) {
        const unsigned inc = own ? (unsigned) stride : 0u, inc4 = inc * 4u;
        unsigned so = own ? (unsigned) ybase * (unsigned) stride + (unsigned) x0 : (unsigned) h * (unsigned) stride + (unsigned) (PX * lane), so4 = so * 4u;
        const int nr = min(R, h - ybase);
#pragma unroll
        for (int r = 0; r < R; r++, so += inc, so4 += inc4) {
            if (r < nr) {
                *(GFV *) ((gu8 *) m_out + so4) = q_mo[r];
                *(GLV *) (least_out + so) = q_lo[r];
            }
        }
    };
    const bool interior = (x0 - PX * lane >= 0) && (x0 - PX * lane + TILE <= w);      // uniform: the whole tile window is inside the image

    const int nblk = (h + RB - 1) / RB;
    issue(q * R);
#ifdef LQR_TILE_TIMING
    unsigned long long tdbg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ttprev = __builtin_readcyclecounter();
    const unsigned long long ttstart = ttprev;
#endif
    for (int j = 0; j < nblk; j++) {
        const int y0 = j * RB;
        const int ylast = min(y0 + RB, h) - 1;
#pragma unroll 1
        for (int bb = 0; bb < NBB; bb++) {
            const int yb = y0 + bb * R;
            const bool mine = ((j * NBB + bb) & (DPP_W - 1)) == q && yb < h;          // batches alternate between the two waves
            TT(0);
            if (mine) {
                if (yb > 0) {
                    const FV v = s_mp[lane];
#pragma unroll
                    for (int k = 0; k < PX; k++) mp[k] = v[k];
                }
                if (bb == 0 && j > 0) {
                    // Halo columns of the row above the block: the tile's own values there are contaminated from the
                    // tile edge inwards, the neighbours hold the true ones and published them as data-tagged granules
                    // (one 8-byte {m, tag} per column, each written by ONE write-through store: no flag, no fence, no
                    // drain -- MI355X_MICROARCH.md, hand-off price list).  tag = (launch epoch, block), so nothing has
                    // to be cleared between launches; two slots by block parity, because a neighbour that is a
                    // whole block ahead publishes block j before this tile has read block j - 1.
                    // Every spin is bounded.  A neighbour that never shows up means the grid was not co-resident (the
                    // host sizes it from the occupancy query, but the GPU may be shared): record the failure in the
                    // host-visible error word and stop waiting -- every other tile sees the word in its own spin loop
                    // and leaves too, the host returns LQR_ERROR at its next synchronisation.  Nothing traps.
                    bool any_in = false;
#pragma unroll
                    for (int k = 0; k < PX; k++) any_in |= in[k];
                    const bool need = !own_lane && any_in;
                    const int nb = (lane < 32) ? tile - 1 : tile + 1;                     // left halo <- left neighbour's right-going granules
                    const int col = !need ? 0 : (lane < 32) ? PX * lane : PX * (lane - 64 + HL);        // lanes that need nothing poll a dummy
                    gu64 *src = ex_img + (size_t) (need ? nb : tile) * EX_TILE + (size_t) (((j - 1) & 1) * 2 + (lane < 32 ? 1 : 0)) * HALO + col;
                    const unsigned want = ((unsigned) epoch << DPP_BLK_BITS) | (unsigned) j;
                    unsigned long long g[PX];
                    int spins = 0;
                    bool failed = false;
                    while (true) {
#pragma unroll
                        for (int k = 0; k < PX; k++) g[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        bool ok = true;
#pragma unroll
                        for (int k = 0; k < PX; k++) ok &= ((unsigned) (g[k] >> 32) == want);
                        if (__all(ok || !need)) break;
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                        if ((spins & 1023) == 0 && dev_failed(dev_err)) { failed = true; break; }
                        if (spins > (1 << 22)) { if (lane == 0) dev_fail(dev_err, DEVERR_TILE_TIMEOUT); failed = true; break; }
                    }
                    if (failed) s_fail = 1;
                    if (need) {
#pragma unroll
                        for (int k = 0; k < PX; k++) mp[k] = in[k] ? __uint_as_float((unsigned) g[k]) : INF;
                    }
                    if (lane == 0) s_polled = j;          // the partner's prefetch may start (see the issue site)
                }
                TT(1);
#ifdef LQR_TILE_TIMING
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                TT(2);
                if constexpr (UPDATE) {
                    if (!interior) {
                        // outside the image both the energy and the OLD value become +inf: m = inf + best = inf, and the keep
                        // rule sees |inf - inf| = NaN -> "unchanged" -> keeps the old +inf.  Masking the energy alone is not
                        // enough: the plane's columns beyond the image hold whatever the block held before (the carve's vector
                        // moves shift that along), and a stale NaN there is "unchanged" too -- it stayed, and reached the image's
                        // last column through the min of the row below (found by scripts/fuzz_parity.py under LQRHIP_POISON=r3)
#pragma unroll
                        for (int r = 0; r < R; r++)
#pragma unroll
                            for (int k = 0; k < PX; k++) { q_e[r][k] = in[k] ? q_e[r][k] : INF; q_mo[r][k] = in[k] ? q_mo[r][k] : INF; }
                    }
                    batch_u(yb);
                } else {
                    if (yb > 0 && yb + R <= h) { if (interior) batch(yb, std::false_type{}, std::false_type{}); else batch(yb, std::false_type{}, std::true_type{}); }
                    else batch(yb, std::true_type{}, std::true_type{});
                }
                TT(3);
                {
                    FV v;
#pragma unroll
                    for (int k = 0; k < PX; k++) v[k] = mp[k];
                    s_mp[lane] = v;
                }
                if (yb + R > ylast && j + 1 < nblk) {
                    // publish the block's last row (still in mp): the outer HALO own columns on each side are the
                    // neighbours' halo; lanes 16..31 write the left-going granules, lanes 32..47 the right-going ones
                    if (own_lane) {
                        const int side = lane < 32 ? 0 : 1;
                        gu64 *dst = ex_img + (size_t) tile * EX_TILE + (size_t) ((j & 1) * 2 + side) * HALO + PX * (lane - (side ? 32 : HL));
                        const unsigned long long tag = (unsigned long long) (((unsigned) epoch << DPP_BLK_BITS) | (unsigned) (j + 1)) << 32;
#pragma unroll
                        for (int k = 0; k < PX; k++) __hip_atomic_store(dst + k, tag | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            TT(4);
            // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. every wave's prefetch
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TT(5);
            if (s_fail) return;                  // uniform: written before the barrier, read by both waves after it
            // this wave's next batch, issued AFTER the barrier: the ~50 load instructions (~1500 cycles of issue) then
            // run under the partner's compute instead of in front of it
            if (mine) {
                // The wave that finished the block holds its loads back until the partner has received the neighbours'
                // hand-over: the hand-off's price sits in the CONSUMER CU's memory queue, where ~50 prefetch loads in front of
                // the poll double the wait (measured on the band variant of this kernel: 4200 -> 2200 cycles per block).
                if (bb == NBB - 1 && j + 1 < nblk) {
                    int spins = 0;
                    while (s_polled < j + 1 && !*(volatile int *) &s_fail && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
                }
                TT(6);
                if constexpr (UPDATE) store_u(yb);        // the batch this wave has just computed, before its registers are reloaded
                TT(7);
                issue(yb + DPP_W * R);
                TT(8);
            }
        }
    }
#ifdef LQR_TILE_TIMING
    if (UPDATE && blockIdx.y == 0 && blockIdx.x == gridDim.x / 2 && lane == 0) { for (int i = 0; i < 10; i++) g_tile_dbg[q * 16 + i] = tdbg[i]; g_tile_dbg[q * 16 + 10] = __builtin_readcyclecounter() - ttstart; }
#endif
    if (UPDATE && threadIdx.x == 0) {
        // the update wrote m2 / least2: the tile that finishes last swaps the image's plane pointers in the device
        // descriptor (every tile read the descriptor before it could finish) and re-arms the counter
        if (__hip_atomic_fetch_add(done_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ntiles - 1) {
            DevCarver *d = cs + blockIdx.y;
            float *m = d->m; d->m = d->m2; d->m2 = m;
            int8_t *l = d->least; d->least = d->least2; d->least2 = l;
            __hip_atomic_store(done_ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------
// E9 update_mmap for large batches, one image's band spread over SEVERAL CUs ("band tiles", round 4).
//
// k_band_update_tw walks an image's band on ONE compute unit, and what bounds it is that unit's vector-memory path (one
// instruction per ~35-50 cycles) -- 64 of 256 CUs busy, each saturated.  k_dp_tile_p<UPDATE> spreads an
// image over one CU per 64 columns and runs a row in a third of the time, but out of place and over the FULL width:
// 14 B/px and 60 tiles per 4K image, too much for a batch.  This kernel is that sweep restricted to T tiles around the
// seam, IN PLACE, with the band kernel's activity test per tile and 32-row block:
//   * tile set: T tiles of 64 own columns (+ 32-column halos, 2 px per lane, two turn-taking waves), centred on the middle
//     of the carve-touched columns of all rows; every tile derives the same placement from seam_x.  The touched columns
//     must keep 32 columns away from real columns outside the set, otherwise row 0 is handed to the full-width sweep
//     (flags[FLAG_OVF_ROW], atomic min) and nothing is done here.
//   * in place: a tile stores block j only after its partner wave has received both neighbours' hand-over for block
//     j + 1, i.e. after both neighbours have FINISHED block j -- their inputs for block j (which include this tile's own
//     columns as their halo) were consumed before.  After the last block a "done" hand-over does the same job.  Reading a
//     pixel a neighbour has already updated would be harmless for the pair (m, back pointer) as a whole (the keep rule is
//     idempotent) but not for a torn pair; the ordering excludes both.
//   * activity: a tile is ACTIVE in block j iff a carve-touched pixel of the block's rows lies within 32 columns of its
//     own columns, or one of its own pixels, or one of the 32 outer own pixels of a neighbour (the hand-over granules carry
//     a "changed" bit), changed on the last row of block j - 1: a change travels one column per row, a block is 32 rows.
//     An inactive tile computes and stores nothing; it hands the stored values of its block's last row on.  Inputs are
//     prefetched two blocks ahead only when the tile may be active then (a wrong guess costs a synchronous load, never
//     a result).
//   * edges: lanes beyond the set read the row above a block from memory (nothing changes out there).  The set GROWS on
//     demand: a third of an image's workgroups are RESERVE tiles that wait on a request word; an edge tile asks for one
//     (atomic ticket + request {epoch, side, first block, tile}) as soon as a change enters its outer 32 columns -- from
//     there it cannot pass the outermost column before the block's last row, so a tile that starts with the NEXT block is
//     in time -- or when the seam comes within 64 columns of the edge in the next two blocks.  The woken tile takes the
//     row above its first block from memory (nothing has changed there yet), hands it to the tile that asked, and joins
//     the protocol; it may ask for the next one.  Every tile that ran counts itself in the image's header when it is done;
//     a reserve tile nobody asked for leaves when as many are done as were ever started (base tiles + tickets drawn, the
//     ticket count unchanged around the read: nobody is left who could ask; the asker drains the request store before it
//     publishes anything
//     later).  Only if no reserve is left and a change reaches the outermost own column before a block's last row does the
//     image stop: the block is not stored, its first row goes to flags[FLAG_OVF_ROW] (atomic min), and ABORT granules tell
//     the neighbours, which pass them on and leave.  Every row below the recorded one is then redone by k_dp_sweep<UPDATE>
//     from memory that holds, per pixel, either the old or the final pair -- the same superset argument as for the band
//     kernels' hand-over.  (Rows past the image in its last, partial block are computed from copies and never counted as
//     changes: they were 90 % of the "aborts" of the first version.)
// Grid (base + reserve tiles, images), all co-resident (spin waits, bounded as in k_dp_tile_p); hand-over granules {m, tag} with
// tag = epoch << 13 | changed << 12 | block.
// ---------------------------------------------------------------------------
#define BT_BLK_ABORT 0x7ffu
// [0] images not covered by their tile set, [1] images aborted at an edge, [2] reserve tiles woken, [3] requests that found
// no reserve left (rare events: one atomic each)
__device__ unsigned long long g_bt_stats[8];
#ifdef LQR_BT_TIMING
// per tile slot of image 0 and wave: cycles in [0] receive, [1] compute, [2] rest before the barrier, [3] barrier, [4] wait for the partner's poll,
// [5] stores, [6] prefetch issue, [7] active blocks, [8] whole kernel
__device__ unsigned long long g_bt_time[16][2][10];
#define BTT(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); btt[i] += t__ - btprev; btprev = t__; } while (0)
#else
#define BTT(i) do { } while (0)
#endif
#define BT_STAT(i) do { if (lane == 0) atomicAdd(&g_bt_stats[i], 1ull); } while (0)
constexpr int BT_MAX_BLK = 256;           // blocks of 32 rows: images up to 8192 rows (the tag has 11 bits for the block)
constexpr int BT_T_MAX = 12;              // workgroups per image: base tiles + reserve tiles
constexpr int BT_HDR = 16;                // 8-byte words of an image's header in the exchange area: [0] base tiles finished,
                                          // [1] requests made, [2 ..] the requests {epoch << 32 | side << 31 | start block << 16 | tile}
constexpr int BT_NEVER = 1 << 30;

// One tile of one image from block j0 on.  gt: the tile's place in the image (columns [64 gt, 64 gt + 64)); left_from /
// right_from: first block from which a neighbour tile exists on that side (BT_NEVER: none -- the columns out there are
// read from memory, and watched).
template <bool LR, bool RIG>
__device__ __forceinline__ void band_tile_run(const GCarver &c, const DpK &p, int w, int h, int stride, GLOBAL_AS unsigned long long *hdr,
                                              GLOBAL_AS unsigned long long *ex_img, int epoch, int *dev_err, int n_rsv,
                                              int gt, int j0, int left_from0, int right_from0, const int *s_tlo, const int *s_thi)
{
    constexpr int PX = 2, HALO = 32, OWN = 64, EX_TILE = 2 * 2 * HALO, HL = 16, R = 32, TILE = 128;
    typedef LaneVec<2>::F FV;
    typedef LaneVec<2>::L LV;
    typedef GLOBAL_AS FV GFV;
    typedef GLOBAL_AS LV GLV;
    typedef GLOBAL_AS unsigned long long gu64;
    __shared__ FV s_mp[64];                       // the row above the next block, handed from wave to wave
    __shared__ int s_fail;                        // leave at the next barrier: a neighbour timed out, or the image was aborted
    __shared__ volatile int s_polled;             // last block whose hand-over this workgroup has received
    __shared__ int s_own_chg;                     // an own pixel changed on the last row of the block just finished
    __shared__ volatile int s_nbr_live;           // the hand-over last received says a neighbour was active or changed at its edge
    __shared__ int s_from[2];                     // first block with a left / right neighbour
    __shared__ int s_asked[2];                    // a reserve tile was asked for on that side (or there is none left)
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = (h + R - 1) / R;
    const int ntiles_img = (w + OWN - 1) / OWN;
    // what the carve-touched columns of each block mean for THIS tile, worked out once: bit 0 within reach of its own columns
    // during the block (the tile is active), bit 1 / 2 within 64 columns of its outermost left / right column (a reserve tile
    // may be needed there), bit 3 within three halos (prefetch)
    __shared__ unsigned char s_flag[BT_MAX_BLK + 4];
    {
        const int own_lo_ = gt * OWN, own_hi_ = min(own_lo_ + OWN, w) - 1;
        for (int b = tid; b < nblk + 4; b += 128) {
            unsigned f = 0;
            if (b < nblk) {
                const int lo = s_tlo[b], hi = s_thi[b];
                f |= (lo <= own_hi_ + HALO + 2 && hi >= own_lo_ - HALO - 2) ? 1u : 0u;
                f |= (lo <= own_lo_ + 2 * HALO && hi >= own_lo_ - 2 * HALO) ? 2u : 0u;
                f |= (lo <= own_hi_ + 2 * HALO && hi >= own_hi_ - 2 * HALO) ? 4u : 0u;
                f |= (lo <= own_hi_ + 3 * HALO + 2 && hi >= own_lo_ - 3 * HALO - 2) ? 8u : 0u;
            }
            s_flag[b] = (unsigned char) f;
        }
    }
    if (tid == 0) { s_fail = 0; s_polled = j0; s_own_chg = 0; s_nbr_live = 0; s_from[0] = left_from0; s_from[1] = right_from0; s_asked[0] = s_asked[1] = 0; }
    __syncthreads();
    const float INF = __int_as_float(0x7f800000);
    const float rig_l = p.rigmap[0], rig_r = p.rigmap[2];
    const int x0 = gt * OWN - HALO + PX * lane;               // first pixel of this lane (may be < 0 or >= w)
    const bool own_lane = lane >= HL && lane < 64 - HL;
    const bool own = own_lane && x0 < w;
    const unsigned lo_off = (unsigned) min(max(x0, 0), stride - PX);
    bool in[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) in[k] = x0 + k >= 0 && x0 + k < w;
    const bool any_in = in[0] || in[1];
    const bool interior = (x0 - PX * lane >= 0) && (x0 - PX * lane + TILE <= w);
    const bool real_l = gt > 0, real_r = gt + 1 < ntiles_img;          // real columns beyond this tile on that side

    FV q_e[R], q_mo[R];
    LV q_lo[R];
    auto issue_full = [&](int ybase) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const unsigned row = (unsigned) min(ybase + r, h - 1) * (unsigned) stride;
            const unsigned ro = row + lo_off, ro4 = (row << 2) + (lo_off << 2);
            q_e[r] = *(const GFV *) ((const gu8 *) c.en + ro4);
            q_mo[r] = *(const GFV *) ((const gu8 *) c.m + ro4);
            q_lo[r] = *(const GLV *) (c.least + ro);
        }
    };
    auto issue_last = [&](int ybase) {        // only the row an inactive tile hands on
        const unsigned row = (unsigned) min(ybase + R - 1, h - 1) * (unsigned) stride;
        q_mo[R - 1] = *(const GFV *) ((const gu8 *) c.m + (((row + lo_off)) << 2));
    };
    float mp[PX] = {INF, INF};
    // What changed in the block, per lane.  "Changed" = the stored m of the pixel has other bits than before: that is all a
    // child row can see of its parents (the keep rule of a child looks at its OWN old pair and its parents' m), so it is exactly
    // what has to travel on.  Accumulated as XORs in VGPRs: two v_xor + two v_or per row -- as lane masks in SGPR pairs (`bool`s
    // of the `changed` flags) four accumulators made the register allocator spill 400 SGPRs into the row loop.
    int acc_e0 = 0, acc_e1 = 0;               // pixel 0 / 1 of the lane, rows before the block's last one (a change ON the last
                                              // row reaches the columns beyond in the next block)
    int acc_l0 = 0, acc_l1 = 0;               // ... on the block's last row
    auto batch_u = [&](int ybase) {
        const int nr = min(R, h - ybase);     // (the image's last block computes surplus rows from copies of its last row)
#pragma unroll
        for (int r = 0; r < R; r++) {
            float mc[PX], e[PX], mo[PX];
            uint32_t lnew = 0;
            bool ch[PX];
#pragma unroll
            for (int k = 0; k < PX; k++) { e[k] = q_e[r][k]; mo[k] = q_mo[r][k]; }
            const float left = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[PX - 1]), DPP_WAVE_SHR1, 0xf, 0xf, true));
            const float right = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[0]), DPP_WAVE_SHL1, 0xf, 0xf, true));
            dp_row<PX, LR, RIG, true, false>(mp, left, right, e, mo, (uint32_t) q_lo[r], in, rig_l, rig_r, mc, lnew, ch);
            if (r == 0 && ybase == 0) {          // row 0: m = en, whatever stood there (update_mmap's first row)
#pragma unroll
                for (int k = 0; k < PX; k++) mc[k] = e[k];
                lnew = 0;
            }
            const int msk = (r < nr) ? -1 : 0;
            const int x0b = (__float_as_int(mc[0]) ^ __float_as_int(mo[0])) & msk, x1b = (__float_as_int(mc[1]) ^ __float_as_int(mo[1])) & msk;
            if (r < R - 1) { acc_e0 |= x0b; acc_e1 |= x1b; } else { acc_l0 = x0b; acc_l1 = x1b; }
#pragma unroll
            for (int k = 0; k < PX; k++) { mp[k] = mc[k]; q_mo[r][k] = mc[k]; }
            q_lo[r] = (LV) lnew;
        }
    };
    auto store_u = [&](int ybase) {
        const unsigned inc = own ? (unsigned) stride : 0u, inc4 = inc * 4u;
        unsigned so = own ? (unsigned) ybase * (unsigned) stride + (unsigned) x0 : (unsigned) h * (unsigned) stride + (unsigned) (PX * lane), so4 = so * 4u;
        const int nr = min(R, h - ybase);
#pragma unroll
        for (int r = 0; r < R; r++, so += inc, so4 += inc4) {
            if (r < nr) {
                *(GFV *) ((gu8 *) c.m + so4) = q_mo[r];
                *(GLV *) (c.least + so) = q_lo[r];
            }
        }
    };
    // the hand-over for block j (published by the neighbours after their block j - 1; j == nblk: "done"): the halo lanes
    // take the row above the block from it.  Returns: bit 0 a neighbour's outer pixels changed, bit 1 abort seen, bit 2 time-out,
    // bit 3 a neighbour was active
    auto receive = [&](int j, bool take) -> int {
        const bool halo_lane = !own_lane && any_in;
        const bool side_l = lane < 32;
        // a neighbour that exists from block f on publishes the hand-over for every block >= f (a reserve tile's first act is
        // the hand-over for its first block; base tiles start at block 0, which has none)
        const bool has_nbr = j >= (side_l ? s_from[0] : s_from[1]);
        const bool from_nbr = halo_lane && has_nbr;
        const bool from_mem = halo_lane && !has_nbr && (side_l ? real_l : real_r);
        const int nb = side_l ? gt - 1 : gt + 1;
        const int col = !from_nbr ? 0 : side_l ? PX * lane : PX * (lane - 64 + HL);
        gu64 *src = ex_img + (size_t) (from_nbr ? nb : gt) * EX_TILE + (size_t) (((j - 1) & 1) * 2 + (side_l ? 1 : 0)) * HALO + col;
        const unsigned want = ((unsigned) epoch << 13) | (unsigned) j, abort_tag = ((unsigned) epoch << 13) | BT_BLK_ABORT;
        unsigned long long g[PX];
        int spins = 0, res = 0;
        FV mem = {INF, INF};
        if (take && from_mem) mem = *(const GFV *) ((const gu8 *) c.m + ((((unsigned) (j * R - 1) * (unsigned) stride) + lo_off) << 2));
        while (true) {
#pragma unroll
            for (int k = 0; k < PX; k++) g[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true, ab = false;
#pragma unroll
            for (int k = 0; k < PX; k++) {
                const unsigned t = (unsigned) (g[k] >> 32);
                ab |= (t == abort_tag);
                ok &= ((t & ~0x1800u) == want) || (t == abort_tag);
            }
            if (__any(from_nbr && ab)) { res |= 2; break; }
            if (__all(ok || !from_nbr)) break;
            {
                // not there yet: ONE lane per side watches one granule, backing off, before the full read is tried again --
                // most tiles of a set are inactive and spend their time here; 32 lanes x 2 agent-scope loads per turn from
                // each of them would sit in front of the active tiles' loads
                // (the innermost halo lane of each side: its columns are inside the image whenever the neighbour exists; the
                // outermost ones may lie beyond the image's last column)
                const bool scout = from_nbr && (lane == HL - 1 || lane == 64 - HL);
                int sp = 0;
                while (true) {
                    if (sp < 4) __builtin_amdgcn_s_sleep(1); else if (sp < 32) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(48);
                    const unsigned t = scout ? (unsigned) (__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) : want;
                    if (__all(!scout || (t & ~0x1800u) == want || t == abort_tag)) break;
                    ++sp;
                    if ((sp & 255) == 0 && dev_failed(dev_err)) break;
                    if (sp > (1 << 16)) break;
                }
            }
            ++spins;
            if ((spins & 63) == 0 && dev_failed(dev_err)) { res |= 4; break; }
            if (spins > (1 << 6)) { if (lane == 0) dev_fail(dev_err, DEVERR_TILE_TIMEOUT); res |= 4; break; }      // 64 x ~0.1 s of backed-off polling
        }
        if (!(res & 6)) {
            bool chg = false;
#pragma unroll
            for (int k = 0; k < PX; k++) chg |= ((unsigned) (g[k] >> 32) & 0x1000u) != 0;
            if (__any(from_nbr && chg)) res |= 1;
            if (__any(from_nbr && (((unsigned) (g[0] >> 32) & 0x800u) != 0))) res |= 8;          // a neighbour was active in its last block
            if (take && halo_lane) {
#pragma unroll
                for (int k = 0; k < PX; k++) mp[k] = !in[k] ? INF : from_nbr ? __uint_as_float((unsigned) g[k]) : from_mem ? mem[k] : INF;
            }
        }
        return res;
    };
    auto publish = [&](int j_next, bool abort, bool was_active) {       // the block's last row (in mp) to both neighbours
        if (own_lane) {
            const int side = lane < 32 ? 0 : 1;
            gu64 *dst = ex_img + (size_t) gt * EX_TILE + (size_t) (((j_next - 1) & 1) * 2 + side) * HALO + PX * (lane - (side ? 32 : HL));
#pragma unroll
            for (int k = 0; k < PX; k++) {
                const unsigned tag = ((unsigned) epoch << 13) | (abort ? BT_BLK_ABORT : (((k == 0 ? acc_l0 : acc_l1) != 0 ? 0x1000u : 0u) | (was_active ? 0x800u : 0u) | (unsigned) j_next));
                __hip_atomic_store(dst + k, ((unsigned long long) tag << 32) | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    auto flag = [&](int b) -> unsigned { return s_flag[max(b, 0)]; };          // (blocks past the image: 0)
    // Ask for a reserve tile beyond this one on side s (0 left, 1 right), to start with block jstart.  One wave, uniform.
    auto ask = [&](int s, int jstart) {
        int okv = 0;
        if (lane == 0) {
            const unsigned long long k = __hip_atomic_fetch_add(hdr + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k < (unsigned long long) n_rsv) {
                const unsigned long long word = ((unsigned long long) (unsigned) epoch << 32) | ((unsigned long long) s << 31) | ((unsigned long long) jstart << 16) |
                                                (unsigned long long) (s == 0 ? gt - 1 : gt + 1);
                __hip_atomic_store(hdr + 2 + k, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the request is in memory before anything this tile publishes later
                okv = 1;
                atomicAdd(&g_bt_stats[2], 1ull);
            } else atomicAdd(&g_bt_stats[3], 1ull);
            s_asked[s] = 1;
            if (okv) s_from[s] = jstart;
        }
    };

#ifdef LQR_BT_TIMING
    unsigned long long btt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long btprev = __builtin_readcyclecounter();
    const unsigned long long btstart = btprev;
#endif
    // this wave's first block: the first one >= j0 of its parity
    const int jq = j0 + (((j0 & 1) != q) ? 1 : 0);
    bool staged = jq < nblk && (flag(jq) & 1u);
    if (jq < nblk) { if (staged) issue_full(jq * R); else issue_last(jq * R); }
    if (j0 == 0 && q == 0) {
        // the seam may start within reach of an edge of the set: a reserve tile from the first block on
        if (s_from[0] == BT_NEVER && real_l && ((flag(0) | flag(1)) & 2u)) ask(0, 0);
        if (s_from[1] == BT_NEVER && real_r && ((flag(0) | flag(1)) & 4u)) ask(1, 0);
    }
    __syncthreads();
    for (int j = j0; j < nblk; j++) {
        const int y0 = j * R;
        const bool mine = (j & 1) == q;
        bool act = false, abort = false, nbr_act = false;
        unsigned fnext = 0;
        if (mine) {
            if (j > j0) {
                const FV v = s_mp[lane];
                mp[0] = v[0]; mp[1] = v[1];
            } else if (j > 0) {
                // a reserve tile's first block: nothing has changed in its columns so far, the row above is in memory; its
                // first act is the hand-over of that row to the neighbour that woke it
                const FV v = *(const GFV *) ((const gu8 *) c.m + ((((unsigned) (y0 - 1) * (unsigned) stride) + lo_off) << 2));
                mp[0] = in[0] ? v[0] : INF; mp[1] = in[1] ? v[1] : INF;
                acc_l0 = acc_l1 = 0;
                publish(j, false, false);
            }
            BTT(2);
            int rcv = 0;
            if (j > 0) {
                rcv = receive(j, true);
                if (rcv & 4) s_fail = 1;
                if (lane == 0) { s_nbr_live = (rcv & 9) != 0; s_polled = j; }
            }
            BTT(0);
            abort = (rcv & 2) != 0;
            act = !abort && !(rcv & 4) && ((flag(j) & 1u) || (j > j0 && s_own_chg != 0) || (rcv & 1));
            acc_e0 = acc_e1 = acc_l0 = acc_l1 = 0;
            fnext = flag(j + 1) | flag(j + 2);
            const bool alone_l = real_l && s_from[0] > j, alone_r = real_r && s_from[1] > j;      // nobody beyond this tile during block j
            if (act) {
                if (!staged) issue_full(y0);
                if (!interior) {
                    // outside the image the energy AND the old value become +inf (see k_dp_tile_p)
#pragma unroll
                    for (int r = 0; r < R; r++)
#pragma unroll
                        for (int k = 0; k < PX; k++) { q_e[r][k] = in[k] ? q_e[r][k] : INF; q_mo[r][k] = in[k] ? q_mo[r][k] : INF; }
                }
                BTT(2);
                batch_u(y0);
                BTT(1);
#ifdef LQR_BT_TIMING
                btt[7]++;
#endif
                // grow the set: a change has entered this edge tile's outer 32 columns (it cannot pass the outermost one before
                // the block's last row), or the seam comes within 64 columns of the edge in the next two blocks
                if (j + 1 < nblk) {
                    const bool any_chg = (acc_e0 | acc_e1 | acc_l0 | acc_l1) != 0;
                    const bool zl = __any(any_chg && own_lane && lane < 32), zr = __any(any_chg && own_lane && lane >= 32);
                    if (real_l && s_from[0] == BT_NEVER && !s_asked[0] && (zl || (fnext & 2u))) ask(0, j + 1);
                    if (real_r && s_from[1] == BT_NEVER && !s_asked[1] && (zr || (fnext & 4u))) ask(1, j + 1);
                }
                __builtin_amdgcn_wave_barrier();
                // nobody beyond the outermost own column during this block: it must not have changed before the block's last
                // row, and if it changed ON the last row somebody must be there from the next block on
                const bool last_l = __any(acc_l0 != 0 && lane == HL), last_r = __any(acc_l1 != 0 && lane == 63 - HL);
                const int fl = *(volatile int *) &s_from[0], fr = *(volatile int *) &s_from[1];
                if ((alone_l && (__any(acc_e0 != 0 && lane == HL) || (last_l && j + 1 < nblk && fl > j + 1))) ||
                    (alone_r && (__any(acc_e1 != 0 && lane == 63 - HL) || (last_r && j + 1 < nblk && fr > j + 1)))) {
                    // this block stays unstored, rows from y0 on are the full-width sweep's
                    if (lane == 0) __hip_atomic_fetch_min(c.flags + FLAG_OVF_ROW, y0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    BT_STAT(1);
                    abort = true;
                }
            } else if (!abort) {
                // nothing can change in this block: hand the stored last row on
                const FV v = q_mo[R - 1];
                mp[0] = in[0] ? v[0] : INF; mp[1] = in[1] ? v[1] : INF;
                if (j + 1 < nblk) {      // the seam may still be heading for this edge
                    if (real_l && s_from[0] == BT_NEVER && !s_asked[0] && (fnext & 2u)) ask(0, j + 1);
                    if (real_r && s_from[1] == BT_NEVER && !s_asked[1] && (fnext & 4u)) ask(1, j + 1);
                }
            }
            {
                FV v;
                v[0] = mp[0]; v[1] = mp[1];
                s_mp[lane] = v;
            }
            if (lane == 0) s_own_chg = 0;
            if (__any(own_lane && (acc_l0 | acc_l1) != 0) && lane == 0) s_own_chg = 1;
            if (abort) s_fail = 1;
            publish(j + 1, abort, act);               // j + 1 == nblk: "done" (the neighbours wait for it before their last store)
            nbr_act = (rcv & 8) != 0;
        }
        BTT(2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        BTT(3);
        if (s_fail) return;                      // uniform: written before the barrier
        if (mine) {
            if (j + 1 < nblk) {
                int spins = 0;
                while (s_polled < j + 1 && !*(volatile int *) &s_fail && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
                if (*(volatile int *) &s_fail) continue;          // the partner saw an abort or a time-out: it is at the barrier
            } else if (act) {
                if (receive(nblk, false) & 6) continue;            // (aborted neighbours: their rows are the sweep's anyway)
            }
            BTT(4);
            if (act) store_u(y0);
            BTT(5);
            const int j2 = j + 2;
            if (j2 < nblk) {
                // (the partner has just received the hand-over for block j + 1: what it says about the neighbours' block j is one
                // block fresher than this wave's own knowledge)
                staged = act || nbr_act || s_nbr_live != 0 || (fnext & 8u);      // an active neighbour's band may arrive within two blocks
                if (staged) issue_full(j2 * R); else issue_last(j2 * R);
            }
            BTT(6);
        }
    }
#ifdef LQR_BT_TIMING
    if (blockIdx.y == 0 && lane == 0 && blockIdx.x < 16) { btt[8] = __builtin_readcyclecounter() - btstart; for (int i = 0; i < 10; i++) g_bt_time[blockIdx.x][q][i] = btt[i]; }
#endif
}

template <bool LR, bool RIG>
// (two waves per SIMD, as the residency bound assumes: left alone the max-ilp scheduler spreads the 32-row loop over 262 registers)
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_band_tiles(DevCarver *cs, DpK p, int w, int h, int stride, unsigned long long *exch, int epoch, int *dev_err, int t_base, int hset)
{
    constexpr int OWN = 64, HALO = 32, EX_TILE = 2 * 2 * HALO, R = 32;
    typedef GLOBAL_AS unsigned long long gu64;
    __shared__ int s_tlo[BT_MAX_BLK], s_thi[BT_MAX_BLK];      // per block: columns the carve touched on its rows
    __shared__ int s_smin, s_smax;
    __shared__ unsigned long long s_req;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_rsv = (int) gridDim.x - t_base, slot = (int) blockIdx.x;
    const int nblk = (h + R - 1) / R;
    const int ntiles_img = (w + OWN - 1) / OWN;
    const GCarver c = gview(cs[blockIdx.y]);
    // two sets of image headers, used by alternate launches (hset): this launch clears the other set for the next one
    // (a memset node per seam round cost 6 us + a dependency gap on the stream)
    gu64 *hdr = (gu64 *) exch + ((size_t) hset * gridDim.y + blockIdx.y) * BT_HDR;
    gu64 *ex_img = (gu64 *) exch + (size_t) 2 * gridDim.y * BT_HDR + (size_t) blockIdx.y * ((size_t) ntiles_img * EX_TILE);
    if (slot == 0 && tid < BT_HDR) ((gu64 *) exch + ((size_t) (hset ^ 1) * gridDim.y + blockIdx.y) * BT_HDR)[tid] = 0ull;
    int gt, j0 = 0, lf = BT_NEVER, rf = BT_NEVER;
    if (slot >= t_base) {
        // a reserve tile: wait until an edge tile of this image asks for it, or until nobody is left who could.  Tiles that
        // run are the base tiles and the reserves whose ticket has been drawn; each counts itself in hdr[0] when it is done.
        // With K tickets drawn (hdr[1], monotone) and K unchanged around a read of hdr[0] that says t_base + K tiles are
        // done, every tile that was ever started has ended: no request can follow.  (The first version left when the BASE
        // tiles were done: a request of reserve tile A precedes the hand-over that lets its neighbour go on, but with a
        // second reserve B beyond A the base tiles can be a block ahead of B's last request -- one time-out in 7 000 fuzz
        // cases, on images of few blocks.)
        const int r = slot - t_base;
        if (tid == 0) {
            unsigned long long word = 0;
            int sp = 0;
            while (true) {
                word = __hip_atomic_load(hdr + 2 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned) (word >> 32) == (unsigned) epoch) break;
                const unsigned long long k0 = __hip_atomic_load(hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k0 <= (unsigned long long) r) {         // (else: this tile's ticket is drawn, the word is on its way)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const unsigned long long fin = __hip_atomic_load(hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const unsigned long long k1 = __hip_atomic_load(hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (k1 == k0 && fin >= (unsigned long long) t_base + k0) { word = 0; break; }
                }
                if (sp < 16) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(64);
                if ((++sp & 255) == 0 && dev_failed(dev_err)) { word = 0; break; }
                if (sp > (1 << 22)) { word = 0; break; }
            }
            s_req = word;
        }
        __syncthreads();
        const unsigned long long word = s_req;
        if ((unsigned) (word >> 32) != (unsigned) epoch) return;
        gt = (int) (word & 0xffffu); j0 = (int) ((word >> 16) & 0x7fffu);
        if ((word >> 31) & 1) lf = j0; else rf = j0;          // asked for on the right of a tile: that tile is its left neighbour
    }
    // carve-touched columns per block, and over the whole image (base tiles derive the set's placement from them)
    for (int i = tid; i < nblk; i += 128) { s_tlo[i] = 1 << 30; s_thi[i] = -1; }
    if (tid == 0) { s_smin = 1 << 30; s_smax = -1; }
    __syncthreads();
    {
        int smin = 1 << 30, smax = -1;
        for (int y = tid; y < h; y += 128) {        // pixels of row y whose inputs the carve changed (as k_band_update_tw)
            const int v0 = c.seam_x[y], vm = c.seam_x[max(y - 1, 0)], vp = c.seam_x[min(y + 1, h - 1)];
            const int t0 = max(min(min(v0, vm), vp) - 2, 0), t1 = min(max(max(v0, vm), vp) + 1, w - 1);
            atomicMin(&s_tlo[y / R], t0); atomicMax(&s_thi[y / R], t1);
            smin = min(smin, t0); smax = max(smax, t1);
        }
        for (int o = 32; o > 0; o >>= 1) { smin = min(smin, __shfl_xor(smin, o)); smax = max(smax, __shfl_xor(smax, o)); }
        if (lane == 0) { atomicMin(&s_smin, smin); atomicMax(&s_smax, smax); }
    }
    __syncthreads();
    if (slot < t_base) {
        const int smin = s_smin, smax = s_smax;
        const int tile0 = max(0, min(((smin + smax) >> 1) / OWN - t_base / 2, ntiles_img - t_base));
        gt = tile0 + slot;
        bool run = gt < ntiles_img;                           // (the set may be wider than the image)
        if (run) {
            const int set_lo = tile0 * OWN, set_end = min((tile0 + t_base) * OWN, w);
            const bool covered = (tile0 == 0 || smin >= set_lo + HALO) && (tile0 + t_base >= ntiles_img || smax < set_end - HALO);
            if (!covered) {                                   // uniform over the image's base tiles
                if (slot == 0 && tid == 0) { __hip_atomic_fetch_min(c.flags + FLAG_OVF_ROW, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(&g_bt_stats[0], 1ull); }
                run = false;
            }
        }
        if (run) {
            if (slot > 0) lf = 0;
            if (slot + 1 < t_base && gt + 1 < ntiles_img) rf = 0;
            band_tile_run<LR, RIG>(c, p, w, h, stride, hdr, ex_img, epoch, dev_err, n_rsv, gt, 0, lf, rf, s_tlo, s_thi);
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(hdr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // this base tile is done (or never ran)
    } else {
        band_tile_run<LR, RIG>(c, p, w, h, stride, hdr, ex_img, epoch, dev_err, n_rsv, gt, j0, lf, rf, s_tlo, s_thi);
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(hdr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // this reserve tile is done
    }
}
#ifdef LQR_BT_TIMING
extern "C" int lqrhip_band_tiles_timing(unsigned long long *out) { (void) hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bt_time), sizeof(unsigned long long) * 320) == hipSuccess ? 0 : -1; }
#endif
extern "C" int lqrhip_band_tiles_stats(unsigned long long *out, int reset)
{
    (void) hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bt_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_bt_stats), z, sizeof z); }
    return 0;
}
static int band_tiles_resident(int n_cu)
{
    int per_cu = 1 << 20;
    auto q = [&](auto kern) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 128, 0) != hipSuccess) { (void) hipGetLastError(); n = 0; }
        per_cu = std::min(per_cu, n);
    };
    q(k_band_tiles<false, false>); q(k_band_tiles<false, true>); q(k_band_tiles<true, false>); q(k_band_tiles<true, true>);
    return std::max(0, per_cu - 1) * n_cu;
}

// ---------------------------------------------------------------------------
// visibility map: seam log -> levels in the base layout (E8 update_vsmap for a
// whole session), inflate (E14), flatten / read-out compaction (E11, E12),
// transpose (E11)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vs_commit(const DevCarver *cs, int w0, int h0, int wc0, int n_seams, int first_level,
                                                    int finish)
{
    const GCarver c = gview(cs[blockIdx.y]);
    extern __shared__ int smi[];
    int *xs = smi;                      // [n_seams]
    int *lvl = smi + n_seams;           // [wc0]
    __shared__ int s_wave[4];
    const int y = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < n_seams; k += 256) xs[k] = c.seam_log[(size_t) k * h0 + y];
    for (int i = tid; i < wc0; i += 256) lvl[i] = 0;
    __syncthreads();
    // position of seam k in the session-start frame: undo the earlier removals
    for (int k = tid; k < n_seams; k += 256) {
        int pz = xs[k];
        for (int j = k - 1; j >= 0; j--) if (xs[j] <= pz) pz++;
        lvl[pz] = first_level + k;
    }
    __syncthreads();
    gi32 *vrow = c.vs + (size_t) y * w0;
    int carry = 0;
    for (int base = 0; base < w0; base += 256) {
        int col = base + tid;
        bool z = (col < w0) && (vrow[col] == 0);
        int total;
        int rank = carry + block_rank_256(z, s_wave, total);
        if (z) {
            int l = lvl[rank];
            if (l == 0 && finish) l = w0;       // liblqr finish_vsmap: the last column
            if (l) vrow[col] = l;
        }
        carry += total;
    }
}

// one interleaved pixel of `ch` bytes, base layout: RGBA pixels are dword-aligned (rows start at y * w * 4) and move as
// one 32-bit access instead of four byte accesses
__device__ __forceinline__ void px_copy(uint8_t *dst, const uint8_t *src, int ch)
{
    if (ch == 4) *(uint32_t *) dst = *(const uint32_t *) src;
    else for (int k = 0; k < ch; k++) dst[k] = src[k];
}
__device__ __forceinline__ void px_avg(uint8_t *dst, const uint8_t *a, const uint8_t *b, int ch)       // (a + b) / 2 per channel, as integers
{
    if (ch == 4) {
        const uint32_t x = *(const uint32_t *) a, y = *(const uint32_t *) b;
        *(uint32_t *) dst = (x & y) + (((x ^ y) & 0xfefefefeu) >> 1);          // per byte floor((x + y) / 2), no carries across bytes
    } else {
        for (int k = 0; k < ch; k++) dst[k] = (uint8_t) (((int) a[k] + (int) b[k]) / 2);
    }
}

// E14: one block per row (blockIdx.x) of one carver of the batch (blockIdx.y: every carver and attached carver of the
// batch in ONE launch -- a launch per carver leaves most of the chip idle behind each row's serial rank scan).
// dup(c) = the seam was computed in this session.
struct InflateDev {
    const uint8_t *rgb;
    const int32_t *vs;
    const float *bias, *rig;
    uint8_t *nrgb;
    int32_t *nvs;
    float *nbias, *nrig;
    int ch;
};
__global__ __launch_bounds__(256) void k_inflate(const InflateDev *jobs, int w0, int w1, int l, int max_level)
{
    __shared__ int s_wave[4];
    const InflateDev j = jobs[blockIdx.y];
    const uint8_t *rgb = j.rgb;
    const int32_t *vs = j.vs;
    const float *bias = j.bias, *rig = j.rig;
    uint8_t *nrgb = j.nrgb;
    int32_t *nvs = j.nvs;
    float *nbias = j.nbias, *nrig = j.nrig;
    const int ch = j.ch;
    const int y = blockIdx.x, tid = threadIdx.x;
    const int32_t *vrow = vs + (size_t) y * w0;
    const size_t ri = (size_t) y * w0, ro = (size_t) y * w1;
    int carry = 0;
    for (int base = 0; base < w0; base += 256) {
        int col = base + tid;
        int v = (col < w0) ? vrow[col] : 0;
        bool dup = (col < w0) && v != 0 && v <= l + max_level - 1 && v >= 2 * max_level - 1;
        int total;
        int rank = carry + block_rank_256(dup, s_wave, total);   // dups strictly before col
        if (col < w0) {
            int z = col + rank;
            int left = col > 0 ? col - 1 : col;
            if (dup) {
                px_avg(nrgb + (ro + z) * ch, rgb + (ri + left) * ch, rgb + (ri + col) * ch, ch);
                if (nbias) nbias[ro + z] = __fmul_rn(__fadd_rn(bias[ri + left], bias[ri + col]), 0.5f);
                if (nrig) nrig[ro + z] = __fmul_rn(__fadd_rn(rig[ri + left], rig[ri + col]), 0.5f);
                if (nvs) nvs[ro + z] = l - v + max_level;
                z++;
            }
            px_copy(nrgb + (ro + z) * ch, rgb + (ri + col) * ch, ch);
            if (nbias) nbias[ro + z] = bias[ri + col];
            if (nrig) nrig[ro + z] = rig[ri + col];
            if (nvs) nvs[ro + z] = v ? v + l - max_level + 1 : 0;
        }
        carry += total;
    }
}

// E11/E12: compaction of the pixels visible at `level`; any output may be null
__global__ __launch_bounds__(256) void k_compact(const uint8_t *rgb, const int32_t *vs, const float *bias, const float *rig,
                                                  uint8_t *nrgb, float *nbias, float *nrig, int32_t *nvmap, int w0, int w, int ch,
                                                  int level, int depth)
{
    __shared__ int s_wave[4];
    const int y = blockIdx.x, tid = threadIdx.x;
    const int32_t *vrow = vs + (size_t) y * w0;
    const size_t ri = (size_t) y * w0, ro = (size_t) y * w;
    int carry = 0;
    for (int base = 0; base < w0; base += 256) {
        int col = base + tid;
        int v = (col < w0) ? vrow[col] : 0;
        bool keep = (col < w0) && (v == 0 || v >= level);
        int total;
        int rank = carry + block_rank_256(keep, s_wave, total);
        if (keep && rank < w) {
            if (nrgb) px_copy(nrgb + (ro + rank) * ch, rgb + (ri + col) * ch, ch);
            if (nbias) nbias[ro + rank] = bias[ri + col];
            if (nrig) nrig[ro + rank] = rig[ri + col];
            if (nvmap) nvmap[ro + rank] = v ? v - depth : 0;
        }
        carry += total;
    }
}

// E11 flatten for every carver of a batch in ONE launch (job table as k_inflate: blockIdx.y = job, blockIdx.x = row)
__global__ __launch_bounds__(256) void k_compact_jobs(const InflateDev *jobs, int w0, int w, int level)
{
    __shared__ int s_wave[4];
    const InflateDev j = jobs[blockIdx.y];
    const int y = blockIdx.x, tid = threadIdx.x, ch = j.ch;
    const int32_t *vrow = j.vs + (size_t) y * w0;
    const size_t ri = (size_t) y * w0, ro = (size_t) y * w;
    int carry = 0;
    for (int base = 0; base < w0; base += 256) {
        int col = base + tid;
        int v = (col < w0) ? vrow[col] : 0;
        bool keep = (col < w0) && (v == 0 || v >= level);
        int total;
        int rank = carry + block_rank_256(keep, s_wave, total);
        if (keep && rank < w) {
            px_copy(j.nrgb + (ro + rank) * ch, j.rgb + (ri + col) * ch, ch);
            if (j.nbias) j.nbias[ro + rank] = j.bias[ri + col];
            if (j.nrig) j.nrig[ro + rank] = j.rig[ri + col];
        }
        carry += total;
    }
}

// E11 transpose of every carver of a batch in one launch (blockIdx.z = job); RGBA pixels move as dwords
__global__ void k_transpose(const InflateDev *jobs, int w, int h)
{
    __shared__ uint32_t t32[32][33];
    __shared__ float tb[32][33], tr[32][33];
    const InflateDev j = jobs[blockIdx.z];
    const uint8_t *rgb = j.rgb;
    const float *bias = j.bias, *rig = j.rig;
    uint8_t *nrgb = j.nrgb;
    float *nbias = j.nbias, *nrig = j.nrig;
    const int ch = j.ch;
    int x = blockIdx.x * 32 + threadIdx.x;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int y = blockIdx.y * 32 + i;
        if (x < w && y < h) {
            const size_t o = (size_t) y * w + x;
            uint32_t p = 0;
            if (ch == 4) p = *(const uint32_t *) (rgb + o * 4);
            else for (int k = 0; k < ch; k++) p |= (uint32_t) rgb[o * ch + k] << (8 * k);
            t32[i][threadIdx.x] = p;
            if (bias) tb[i][threadIdx.x] = bias[o];
            if (rig) tr[i][threadIdx.x] = rig[o];
        }
    }
    __syncthreads();
    int oy = blockIdx.y * 32 + threadIdx.x;        // output column index = old y
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int ox = blockIdx.x * 32 + i;              // output row index = old x
        if (ox < w && oy < h) {
            uint32_t p = t32[threadIdx.x][i];
            size_t o = (size_t) ox * h + oy;
            if (ch == 4) *(uint32_t *) (nrgb + o * 4) = p;
            else for (int k = 0; k < ch; k++) nrgb[o * ch + k] = (uint8_t) (p >> (8 * k));
            if (nbias) nbias[o] = tb[threadIdx.x][i];
            if (nrig) nrig[o] = tr[threadIdx.x][i];
        }
    }
}

// auto-size (plug-in's guess_new_size, src/layers_combo.c:275-392): one block per line counts the
// mask pixels at or above the threshold; atomicMax over lines
__global__ __launch_bounds__(256) void k_mask_line_max(const uint8_t *mask, int channels, int width, int a0, int b0, int line_len,
                                                       int direction, int *out)
{
    __shared__ int s_cnt[4];
    const bool has_alpha = (channels == 2 || channels == 4);
    const int c_bpp = channels - (has_alpha ? 1 : 0);
    const int line = blockIdx.x;
    int cnt = 0;
    for (int z2 = threadIdx.x; z2 < line_len; z2 += 256) {
        // direction 0: row a0+line, columns b0+z2;  direction 1: column a0+line, rows b0+z2
        const size_t idx = direction == 0 ? (size_t) (a0 + line) * width + (b0 + z2) : (size_t) (b0 + z2) * width + (a0 + line);
        const uint8_t *px = mask + idx * channels;
        double sum = 0.0;
        for (int c = 0; c < c_bpp; c++) sum = __dadd_rn(sum, (double) px[c]);
        sum = __ddiv_rn(sum, (double) (255 * c_bpp));
        if (has_alpha) sum = __dmul_rn(sum, __ddiv_rn((double) px[channels - 1], 255.0));
        cnt += (sum >= __ddiv_rn(0.5, (double) c_bpp)) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
}

// ===========================================================================
// host side of the shim
// ===========================================================================
struct LqrHipCarver {
    int ch = 0;
    int w0 = 0, h0 = 0;              // base layout dims
    // base planes
    uint8_t *rgb0 = nullptr;
    int32_t *vs = nullptr;           // owned by roots only
    float *bias0 = nullptr, *rig0 = nullptr;
    // working planes
    int active = 0;
    int stride = 0, wk_h = 0;
    uint32_t *pix = nullptr;
    float *en = nullptr, *m = nullptr, *m2 = nullptr, *bias = nullptr, *rig = nullptr;
    int8_t *least = nullptr, *least2 = nullptr;
    int32_t *seam_x = nullptr, *seam_log = nullptr, *flags = nullptr;
    int log_cap = 0, log_h = 0;
    int frozen_epoch = 0;           // pix / bias are in the frame before seam `frozen_epoch` of the session
    LqrHipCarver *root = nullptr;
    std::vector<LqrHipCarver *> aux;
    LqrHipBatch *batch = nullptr;
};

struct LqrHipBatch {
    std::vector<LqrHipCarver *> cs;
    DevCarver *d_desc = nullptr;
    hipStream_t stream = nullptr;
    unsigned long long *exch = nullptr;     // k_dp_tile_p: halo granules per image and tile + finished-tile counters
    size_t exch_elems = 0;
    int exch_ntiles = 0, exch_n = 0, exch_px = 0;      // geometry the exchange area was last laid out for
    int tile_epoch = 0;                     // launches of k_dp_tile_p on this batch (part of the granule tags)
    int bt_launches = 0;                    // launches of k_band_tiles on this batch (which of the two header sets)
    bool dirty = true;
    int shared_n = 1;                       // ... how many batches of the group there are (lqrhip_batch_set_shared)
    bool shared = false;                    // other batches of the same group run concurrently on their own streams:
                                            // no persistent (spin-waiting, co-residency-dependent) kernels
};

struct ProfRec {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double bytes = 0;
};
static int g_prof = 0;                 // 0 off, 1 every kernel, 2 the roofline kernel (k_carve) only
static std::map<std::string, ProfRec> g_profrec;
static hipStream_t g_stream0 = nullptr;

extern "C" const char *lqrhip_last_error(void) { return g_err.c_str(); }

// Workgroups of k_dp_tile_p the device holds at once.  Its tiles spin on their neighbours, so the grid
// must be co-resident: the bound comes from the occupancy query of every instantiation that can be
// launched (the minimum over them), less one workgroup per CU of margin -- the API is known to answer
// one block per CU too many at some SGPR counts (MI355X_MICROARCH.md, residency) -- times the CU count.
// A grid above the bound goes to k_dp_tile (kernel boundaries instead of spin waits).
static int dpp_resident_workgroups(int dev)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 0;
    int per_cu = 1 << 20;
    auto q = [&](auto kern) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 64 * DPP_W, 0) != hipSuccess) { (void) hipGetLastError(); n = 0; }
        per_cu = std::min(per_cu, n);
    };
    q(k_dp_tile_p<4, false, false, false>); q(k_dp_tile_p<4, false, true, false>); q(k_dp_tile_p<4, true, false, false>); q(k_dp_tile_p<4, true, true, false>);
    q(k_dp_tile_p<4, false, false, true>); q(k_dp_tile_p<4, false, true, true>); q(k_dp_tile_p<4, true, false, true>); q(k_dp_tile_p<4, true, true, true>);
    // the 2-px instantiations stage a whole 32-row block (193 VGPRs): their bound is lower, and a grid that is too large for
    // them but fits the 4-px ones must not be sent to k_dp_tile for it
    g_dpp_max_wgs_px4 = std::max(0, per_cu - 1) * prop.multiProcessorCount;
    q(k_dp_tile_p<2, false, false, false>); q(k_dp_tile_p<2, false, true, false>); q(k_dp_tile_p<2, true, false, false>); q(k_dp_tile_p<2, true, true, false>);
    q(k_dp_tile_p<2, false, false, true>); q(k_dp_tile_p<2, false, true, true>); q(k_dp_tile_p<2, true, false, true>); q(k_dp_tile_p<2, true, true, true>);
    g_dpp_max_wgs_plain = std::max(0, per_cu - 1) * prop.multiProcessorCount;
    // the delta_x = 2 / rigidity-mask instantiations (2 px per lane only) hold more registers
    per_cu = 1 << 20;
#define QG(LRV, UPD) q(k_dp_tile_p<2, LRV, true, UPD, 1, true>); q(k_dp_tile_p<2, LRV, false, UPD, 2, false>); q(k_dp_tile_p<2, LRV, true, UPD, 2, false>); q(k_dp_tile_p<2, LRV, true, UPD, 2, true>); \
    q(k_dp_tile_p<2, LRV, false, UPD, 3, false>); q(k_dp_tile_p<2, LRV, true, UPD, 3, false>); q(k_dp_tile_p<2, LRV, true, UPD, 3, true>); \
    q(k_dp_tile_p<2, LRV, false, UPD, 4, false>); q(k_dp_tile_p<2, LRV, true, UPD, 4, false>); q(k_dp_tile_p<2, LRV, true, UPD, 4, true>)
    QG(false, false); QG(false, true); QG(true, false); QG(true, true);
#undef QG
    g_dpp_max_wgs_general = std::max(0, per_cu - 1) * prop.multiProcessorCount;
    g_dpp_max_wgs_tiles = band_tiles_resident(prop.multiProcessorCount);
    return g_dpp_max_wgs_plain;
}

#ifdef LQR_TILE_TIMING
extern "C" int lqrhip_tile_timing(unsigned long long *out) { (void) hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_dbg), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1; }
#endif
// Large lock-step groups are carved on 4 streams, and those need hardware queues of their own: the HIP runtime's
// GPU_MAX_HW_QUEUES, default 4 per process, read ONCE when the runtime initialises (lqrhip_sub_batches below).  A host
// that has never heard of the variable (the plug-in) would silently get one stream and 10 % less.  So when this library
// is loaded into a process that has not brought the GPU runtime up yet -- no descriptor of /dev/kfd is open -- and the
// variable is not set, it is set to 8 here, before the library's own first HIP call initialises the runtime.  A host that
// set it (to anything) keeps its value; a host whose runtime is already up keeps one stream.
static bool kfd_is_open(void)
{
    DIR *d = opendir("/proc/self/fd");
    if (!d) return true;                      // cannot tell: leave the environment alone
    bool open_ = false;
    while (struct dirent *e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        char path[64], link[64];
        snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(path, link, sizeof link - 1);
        if (n <= 0) continue;
        link[n] = 0;
        if (strcmp(link, "/dev/kfd") == 0) { open_ = true; break; }
    }
    closedir(d);
    return open_;
}
__attribute__((constructor)) static void lqrhip_on_load(void)
{
    if (getenv("GPU_MAX_HW_QUEUES") || kfd_is_open()) return;
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

extern "C" int lqrhip_init(void)
{
    if (g_device >= 0) return g_device;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_err = "no HIP device visible: the MI355X engine needs a gfx950 GPU (there is no CPU fallback)";
        (void) hipGetLastError();
        return LQRHIP_EHIP;
    }
    int dev = 0;
    const char *lr = getenv("LOCAL_RANK");
    if (lr) dev = atoi(lr) % n;
    HIPCK(hipSetDevice(dev));
    HIPCK(hipStreamCreateWithFlags(&g_stream0, hipStreamNonBlocking));
    HIPCK(hipHostMalloc((void **) &g_dev_err_host, sizeof(int), hipHostMallocMapped));
    *g_dev_err_host = 0;
    HIPCK(hipHostGetDevicePointer((void **) &g_dev_err, g_dev_err_host, 0));
    g_dpp_max_wgs = dpp_resident_workgroups(dev);
    g_device = dev;
    return dev;
}

// A kernel recorded a failure (dev_fail): report it once, as an error return, and clear the word.  The word is one per
// process and whoever synchronises first finds it -- not necessarily the batch whose kernel failed.  A persistent sweep that
// gave up half way leaves its batch's exchange area (tags, finished-tile counter) and, for an update, the plane pointers
// in the device descriptors in an unknown state, so EVERY live batch is marked for a fresh lay-out of both.
static std::vector<LqrHipBatch *> g_live_batches;
static void invalidate_all_batches(void);
static int check_dev_error(void)
{
    if (!g_dev_err_host || *g_dev_err_host == 0) return 0;
    const int code = *g_dev_err_host;
    *g_dev_err_host = 0;
    invalidate_all_batches();
    g_err = code == DEVERR_TILE_TIMEOUT ? "persistent tiled DP sweep: a neighbour tile never became resident (GPU shared or partitioned?); "
                                          "results of this resize are invalid"
                                        : "band update: activity prediction failed; results of this resize are invalid";
    return LQRHIP_EHIP;
}

// Device allocations go through a small size-class cache: the carve path allocates and frees
// image-sized planes for every inflate / flatten / read-out, and hipMalloc / hipFree (which
// synchronises the device) would otherwise cost more than the kernels between them.  A block is
// only returned to the cache after the stream that used it has been synchronised.
static std::multimap<size_t, void *> g_pool_free;
static std::map<void *, size_t> g_pool_size;
static size_t g_pool_cached = 0;
static const size_t POOL_MAX_CACHED = (size_t) 24 << 30;

// LQRHIP_POISON=<byte> in the environment (debugging aid, scripts/fuzz_parity.py): every block handed out is first filled
// with that byte and the device synchronised, so that a kernel that reads memory nothing wrote yet fails the same way every
// time instead of depending on what the block held before
// LQRHIP_POISON=r1 / r2 / r3: pseudo-random words that look like what a recycled block holds -- floats in [0, 100),
// integers in [0, 2048), arbitrary bits
__global__ void k_poison_random(unsigned *p, size_t n, int mode, int stride, int c0, int c1, int r0, int r1)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        if (stride > 0) {       // LQRHIP_POISON_WINDOW=stride:c0:c1:r0:r1 -- only that window of a plane, zero elsewhere
            const int col = (int) (i % stride), row = (int) (i / stride);
            if (col < c0 || col >= c1 || row < r0 || row >= r1) { p[i] = 0; continue; }
        }
        unsigned hsh = (unsigned) i * 2654435761u + 0x9e3779b9u;
        hsh ^= hsh >> 15; hsh *= 0x85ebca6bu; hsh ^= hsh >> 13; hsh *= 0xc2b2ae35u; hsh ^= hsh >> 16;
        p[i] = mode == 1 ? __float_as_uint((float) (hsh >> 8) * (100.0f / 16777216.0f)) : mode == 2 ? (hsh >> 21) : hsh;
    }
}
static const char *g_alloc_name = "";      // what the allocation is for (LQRHIP_POISON_LOG)
static int g_poison = -2;
static unsigned g_poison32 = 0;
static int pool_poison(void *p, size_t sz)
{
    if (g_poison == -2) {
        const char *e = getenv("LQRHIP_POISON");
        g_poison = -1;
        if (e && e[0] == '0' && e[1] == 'x') { g_poison = 256; g_poison32 = (unsigned) strtoul(e, nullptr, 16); }     // a 32-bit word
        else if (e && e[0] == 'r') g_poison = 256 + atoi(e + 1);
        else if (e && *e) g_poison = atoi(e) & 255;
    }
    if (g_poison < 0) return 0;
    {   // LQRHIP_POISON_RANGE=a:b poisons only the allocations numbered a .. b-1 of the process (LQRHIP_POISON_LOG lists them)
        static long seq = 0, lo = 0, hi = -1;
        static int logit = -1;
        if (logit < 0) {
            logit = getenv("LQRHIP_POISON_LOG") != nullptr;
            const char *r = getenv("LQRHIP_POISON_RANGE");
            if (r) sscanf(r, "%ld:%ld", &lo, &hi);
        }
        const long me = seq++;
        if (logit) fprintf(stderr, "alloc %ld %zu %s\n", me, sz, g_alloc_name);
        if (hi >= 0 && (me < lo || me >= hi)) { HIPCK(hipMemset(p, 0, sz)); HIPCK(hipDeviceSynchronize()); return 0; }
    }
    if (g_poison > 256) {
        static int win[5] = {0, 0, 0, 0, 0};
        static bool once = false;
        if (!once) { once = true; const char *wv = getenv("LQRHIP_POISON_WINDOW"); if (wv) sscanf(wv, "%d:%d:%d:%d:%d", win, win + 1, win + 2, win + 3, win + 4); }
        hipLaunchKernelGGL(k_poison_random, dim3(1024), dim3(256), 0, 0, (unsigned *) p, sz / 4, g_poison - 256, win[0], win[1], win[2], win[3], win[4]);
    }
    else if (g_poison == 256) HIPCK(hipMemsetD32((hipDeviceptr_t) p, (int) g_poison32, sz / 4));
    else HIPCK(hipMemset(p, g_poison, sz));
    HIPCK(hipDeviceSynchronize());
    return 0;
}

static int pool_alloc(void **p, size_t bytes)
{
    const size_t sz = (bytes + ((size_t) 1 << 20) - 1) & ~(((size_t) 1 << 20) - 1);      // 1 MiB classes
    auto it = g_pool_free.find(sz);
    if (it != g_pool_free.end()) {
        *p = it->second;
        g_pool_free.erase(it);
        g_pool_cached -= sz;
        return pool_poison(*p, sz);
    }
    hipError_t e = hipMalloc(p, sz);
    if (e == hipErrorOutOfMemory && !g_pool_free.empty()) {       // give the cache back and retry once
        (void) hipGetLastError();
        for (auto &kv : g_pool_free) { (void) hipFree(kv.second); g_pool_size.erase(kv.second); }
        g_pool_free.clear();
        g_pool_cached = 0;
        e = hipMalloc(p, sz);
    }
    HIPCK(e);
    g_pool_size[*p] = sz;
    return pool_poison(*p, sz);
}
static void pool_free(void *p)
{
    auto it = g_pool_size.find(p);
    if (it == g_pool_size.end()) { (void) hipFree(p); return; }
    if (g_pool_cached + it->second > POOL_MAX_CACHED) {
        (void) hipFree(p);
        g_pool_size.erase(it);
        return;
    }
    g_pool_free.emplace(it->second, p);
    g_pool_cached += it->second;
}

template <typename T>
static int dmalloc_(T **p, size_t n, const char *name)
{
    *p = nullptr;
    g_alloc_name = name;
    return pool_alloc((void **) p, (n ? n : 1) * sizeof(T));
}
#define dmalloc(p, n) dmalloc_((p), (n), #p)
template <typename T>
static void dfree(T *&p)
{
    if (p) pool_free((void *) p);
    p = nullptr;
}

// zero device memory and wait: hipMemset on the null stream is asynchronous for device memory and the
// engine's streams are non-blocking, so a null-stream memset is not ordered with the kernels after it
static hipError_t dzero(void *p, size_t bytes)
{
    hipError_t e = hipMemsetAsync(p, 0, bytes, g_stream0);
    return e != hipSuccess ? e : hipStreamSynchronize(g_stream0);
}

// Host <-> device copies of whole images.  The plug-in hands over and takes back PAGEABLE memory (a g_malloc'ed buffer at
// lqr_carver_new, render.c:222; the scan-line buffer at read-out, io_functions.c:155-164); hipMemcpy on pageable memory
// runs at ~1-2 GB/s here (it pins and unpins as it goes).  These go through a ring of pinned bounce buffers instead, and
// the CPU side of the bounce -- memcpy between the caller's pageable buffer and the ring, page faults of a freshly
// allocated destination included -- is done by a few helper threads in parallel while the DMA engine moves other chunks
// (round 3: one thread, two 8 MB buffers: 8.6 GB/s up, 4.2 GB/s down; a single core's memcpy and its page faults were
// the limit, not PCIe).  The helpers touch host memory only; every HIP call stays on the caller's thread.
// Both functions return with the transfer complete, also on error (the stream is drained before they return).
static const size_t STAGE_BYTES = (size_t) 4 << 20;
static const int STAGE_SLOTS = 16;
static uint8_t *g_stage[STAGE_SLOTS];
static hipEvent_t g_stage_ev[STAGE_SLOTS];
static bool g_stage_ready = false;

struct CopyPool {
    struct Job { void *dst; const void *src; size_t n; };
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<std::pair<int, Job>> q;      // (ticket, job)
    std::vector<char> done;                 // per ticket of the current transfer
    size_t pending = 0;                     // submitted and not finished
    bool stop = false;
    void start(int n)
    {
        for (int i = 0; i < n; i++) th.emplace_back([this] { run(); });
    }
    void run()
    {
        for (;;) {
            std::pair<int, Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [this] { return stop || !q.empty(); });
                if (stop && q.empty()) return;
                j = q.front(); q.pop_front();
            }
            memcpy(j.second.dst, j.second.src, j.second.n);
            {
                std::lock_guard<std::mutex> lk(mu);
                done[j.first] = 1;
                pending--;
            }
            cv_done.notify_all();
        }
    }
    void begin(size_t tickets) { std::lock_guard<std::mutex> lk(mu); done.assign(tickets, 0); }
    void submit(int ticket, void *dst, const void *src, size_t n)
    {
        { std::lock_guard<std::mutex> lk(mu); q.emplace_back(ticket, Job{dst, src, n}); pending++; }
        cv_job.notify_one();
    }
    void drain()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [this] { return pending == 0; });
    }
    void wait(int ticket)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done[ticket] != 0; });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_job.notify_all();
        for (auto &t : th) t.join();
    }
};
static CopyPool *g_copy_pool = nullptr;

static int stage_init(void)
{
    if (g_stage_ready) return 0;
    // all or nothing: a partial set of buffers / events is released again, so a later call starts over
    int made = 0;
    hipError_t e = hipSuccess;
    for (; made < STAGE_SLOTS; made++) {
        g_stage[made] = nullptr; g_stage_ev[made] = nullptr;
        if ((e = hipHostMalloc((void **) &g_stage[made], STAGE_BYTES, hipHostMallocDefault)) != hipSuccess) break;
        if ((e = hipEventCreateWithFlags(&g_stage_ev[made], hipEventDisableTiming)) != hipSuccess) { (void) hipHostFree(g_stage[made]); break; }
    }
    if (made < STAGE_SLOTS) {
        for (int i = 0; i < made; i++) { (void) hipHostFree(g_stage[i]); (void) hipEventDestroy(g_stage_ev[i]); g_stage[i] = nullptr; }
        g_err = std::string("staging buffers: ") + hipGetErrorString(e);
        (void) hipGetLastError();
        return e == hipErrorOutOfMemory ? LQRHIP_ENOMEM : LQRHIP_EHIP;
    }
    if (!g_copy_pool) {
        unsigned hw = std::thread::hardware_concurrency();
        g_copy_pool = new CopyPool();
        g_copy_pool->start(hw >= 16 ? 8 : hw >= 4 ? (int) hw / 2 : 1);
    }
    g_stage_ready = true;
    return 0;
}
static int h2d_staged(void *dst, const void *src, size_t bytes)
{
    int rc = stage_init();
    if (rc) return rc;
    const size_t nchunk = (bytes + STAGE_BYTES - 1) / STAGE_BYTES;
    CopyPool &cp = *g_copy_pool;
    cp.begin(nchunk);
    auto body = [&]() -> int {
        size_t submitted = 0;
        for (size_t i = 0; i < nchunk; i++) {
            // keep the helpers a ring ahead: chunk j goes into slot j % STAGE_SLOTS once the DMA that last read it is done
            for (; submitted < nchunk && submitted < i + STAGE_SLOTS; submitted++) {
                const int slot = (int) (submitted % STAGE_SLOTS);
                HIPCK(hipEventSynchronize(g_stage_ev[slot]));
                const size_t off = submitted * STAGE_BYTES;
                cp.submit((int) submitted, g_stage[slot], (const uint8_t *) src + off, std::min(STAGE_BYTES, bytes - off));
            }
            const int slot = (int) (i % STAGE_SLOTS);
            const size_t off = i * STAGE_BYTES;
            cp.wait((int) i);
            HIPCK(hipMemcpyAsync((uint8_t *) dst + off, g_stage[slot], std::min(STAGE_BYTES, bytes - off), hipMemcpyHostToDevice, g_stream0));
            HIPCK(hipEventRecord(g_stage_ev[slot], g_stream0));
        }
        return 0;
    };
    rc = body();
    cp.drain();         // whatever happened, nobody touches the caller's buffer or the ring after we return
    hipError_t e = hipStreamSynchronize(g_stream0);
    if (!rc && e != hipSuccess) { g_err = std::string("upload: ") + hipGetErrorString(e); rc = LQRHIP_EHIP; }
    return rc;
}
static int d2h_staged(void *dst, const void *src, size_t bytes)
{
    int rc = stage_init();
    if (rc) return rc;
    const size_t nchunk = (bytes + STAGE_BYTES - 1) / STAGE_BYTES;
    CopyPool &cp = *g_copy_pool;
    cp.begin(nchunk);
    size_t copied_out = 0;      // chunks handed to the helpers
    auto hand_over = [&](size_t j) -> int {
        const int slot = (int) (j % STAGE_SLOTS);
        const size_t off = j * STAGE_BYTES;
        HIPCK(hipEventSynchronize(g_stage_ev[slot]));
        cp.submit((int) j, (uint8_t *) dst + off, g_stage[slot], std::min(STAGE_BYTES, bytes - off));
        return 0;
    };
    auto body = [&]() -> int {
        for (size_t i = 0; i < nchunk; i++) {
            if (i >= (size_t) STAGE_SLOTS) {              // slot reuse: the helper must have emptied it
                for (; copied_out <= i - STAGE_SLOTS; copied_out++) { int r = hand_over(copied_out); if (r) return r; }
                cp.wait((int) (i - STAGE_SLOTS));
            }
            const int slot = (int) (i % STAGE_SLOTS);
            const size_t off = i * STAGE_BYTES;
            HIPCK(hipMemcpyAsync(g_stage[slot], (const uint8_t *) src + off, std::min(STAGE_BYTES, bytes - off), hipMemcpyDeviceToHost, g_stream0));
            HIPCK(hipEventRecord(g_stage_ev[slot], g_stream0));
            // hand over whatever has landed already, without waiting for it
            while (copied_out < i && hipEventQuery(g_stage_ev[copied_out % STAGE_SLOTS]) == hipSuccess) { int r = hand_over(copied_out); if (r) return r; copied_out++; }
        }
        for (; copied_out < nchunk; copied_out++) { int r = hand_over(copied_out); if (r) return r; }
        return 0;
    };
    rc = body();
    cp.drain();         // the helpers are done with the caller's buffer
    if (rc) (void) hipStreamSynchronize(g_stream0);
    return rc;
}

static int batch_sync_of(LqrHipCarver *c)
{
    LqrHipCarver *r = c->root ? c->root : c;
    if (r->batch) HIPCK(hipStreamSynchronize(r->batch->stream));
    return 0;
}

extern "C" LqrHipCarver *lqrhip_carver_create(const unsigned char *rgb, int w, int h, int channels)
{
    if (lqrhip_init() < 0) return nullptr;
    LqrHipCarver *c = new LqrHipCarver();
    c->ch = channels; c->w0 = w; c->h0 = h;
    size_t n = (size_t) w * h;
    if (dmalloc(&c->rgb0, n * channels) || dmalloc(&c->vs, n)) { lqrhip_carver_destroy(c); return nullptr; }
    // the visibility map is cleared on the same stream, under the upload: one synchronisation for both
    if (hipMemsetAsync(c->vs, 0, n * sizeof(int32_t), g_stream0) != hipSuccess || h2d_staged(c->rgb0, rgb, n * channels) != 0) {
        g_err = "upload failed";
        lqrhip_carver_destroy(c);
        return nullptr;
    }
    return c;
}

static void free_working(LqrHipCarver *c)
{
    dfree(c->pix); dfree(c->en); dfree(c->m); dfree(c->least); dfree(c->m2); dfree(c->least2); dfree(c->bias); dfree(c->rig);
    dfree(c->seam_x); dfree(c->seam_log); dfree(c->flags);
    c->log_cap = 0;
}

extern "C" void lqrhip_carver_destroy(LqrHipCarver *c)
{
    if (!c) return;
    // its planes may still be in use by kernels on the owning batch's stream or by the shim's own stream (resets, mask
    // uploads, read-outs): wait for those two, not for the device (tearing a batch down was 64 device synchronisations)
    {
        LqrHipCarver *r = c->root ? c->root : c;
        if (r->batch && r->batch->stream) (void) hipStreamSynchronize(r->batch->stream);
        if (c->batch && c->batch != r->batch && c->batch->stream) (void) hipStreamSynchronize(c->batch->stream);
        if (g_stream0) (void) hipStreamSynchronize(g_stream0);
        (void) hipGetLastError();
    }
    dfree(c->rgb0);
    if (!c->root) dfree(c->vs);
    dfree(c->bias0); dfree(c->rig0);
    free_working(c);
    delete c;
}

extern "C" int lqrhip_carver_attach(LqrHipCarver *root, LqrHipCarver *aux)
{
    if (root->w0 != aux->w0 || root->h0 != aux->h0) return LQRHIP_EARG;
    dfree(aux->vs);
    aux->vs = root->vs;
    aux->root = root;
    root->aux.push_back(aux);
    if (root->batch) root->batch->dirty = true;
    return 0;
}

// (re)allocate the working planes for a w x h carved frame
static int ensure_working(LqrHipCarver *c, int w, int h)
{
    int stride = ((w + 16) + 63) & ~63;
    bool need_bias = c->bias0 != nullptr, need_rig = c->rig0 != nullptr;
    if (c->pix && c->stride == stride && c->wk_h == h && (!!c->bias == need_bias) && (!!c->rig == need_rig)) return 0;
    free_working(c);
    c->stride = 0; c->wk_h = 0;
    size_t n = (size_t) stride * (h + 1) + 1024;
    int rc;
    if ((rc = dmalloc(&c->pix, n)) || (rc = dmalloc(&c->en, n)) || (rc = dmalloc(&c->m, n)) || (rc = dmalloc(&c->least, n)) ||
        (rc = dmalloc(&c->seam_x, (size_t) h + 8)) || (rc = dmalloc(&c->flags, (size_t) FLAG_WORDS)) ||
        (need_bias && (rc = dmalloc(&c->bias, n))) || (need_rig && (rc = dmalloc(&c->rig, n)))) {
        free_working(c);            // never leave a half-allocated set behind: a retry must not pass the early-out above
        return rc;
    }
    // all on the shim's stream, one synchronisation
    hipError_t e = hipMemsetAsync(c->least, 0, n, g_stream0);
    if (e == hipSuccess) e = hipMemsetAsync(c->m, 0, n * sizeof(float), g_stream0);
    if (e == hipSuccess) e = hipMemsetAsync(c->en, 0, n * sizeof(float), g_stream0);
    if (e == hipSuccess) e = hipMemsetAsync(c->pix, 0, n * sizeof(uint32_t), g_stream0);
    if (e == hipSuccess) e = hipMemsetAsync(c->flags, 0, (size_t) FLAG_WORDS * sizeof(int32_t), g_stream0);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream0);
    if (e != hipSuccess) { free_working(c); HIPCK(e); }
    c->stride = stride; c->wk_h = h;
    if (c->batch) c->batch->dirty = true;
    return 0;
}

static int ensure_log(LqrHipCarver *c, int n_seams, int h)
{
    if (c->seam_log && c->log_cap >= n_seams && c->log_h == h) return 0;
    dfree(c->seam_log);
    int rc = dmalloc(&c->seam_log, (size_t) n_seams * h);
    if (rc) return rc;
    c->log_cap = n_seams; c->log_h = h;
    if (c->batch) c->batch->dirty = true;
    return 0;
}

extern "C" int lqrhip_carver_activate(LqrHipCarver *c)
{
    c->active = 1;
    // E1 lqr_carver_init allocates the DP maps: do the same here, so that the first resize does
    // not pay for hipMalloc (re-done lazily by lqrhip_wk_init if the geometry changes)
    return ensure_working(c, c->w0, c->h0);
}

extern "C" int lqrhip_mask_add(LqrHipCarver *c, const unsigned char *mask, int channels, int width, int height, int x_off,
                               int y_off, int transposed, int is_rigmask, int bias_factor)
{
    int rc = batch_sync_of(c);
    if (rc) return rc;
    size_t n = (size_t) c->w0 * c->h0;
    float **plane = is_rigmask ? &c->rig0 : &c->bias0;
    if (!*plane) {
        if ((rc = dmalloc(plane, n))) return rc;
        HIPCK(dzero(*plane, n * sizeof(float)));
        if (c->batch) c->batch->dirty = true;
    }
    int wt = transposed ? c->h0 : c->w0, ht = transposed ? c->w0 : c->h0;
    int x0 = x_off < 0 ? x_off : 0, y0 = y_off < 0 ? y_off : 0;
    int x1 = x_off > 0 ? x_off : 0, y1 = y_off > 0 ? y_off : 0;
    int x2 = wt < width + x_off ? wt : width + x_off, y2 = ht < height + y_off ? ht : height + y_off;
    int nx = x2 - x1, ny = y2 - y1;
    if (nx <= 0 || ny <= 0) return 0;
    uint8_t *dmask = nullptr;
    size_t mbytes = (size_t) width * height * channels;
    if ((rc = dmalloc(&dmask, mbytes))) return rc;
    auto run = [&]() -> int {
        int rcu = h2d_staged(dmask, mask, mbytes);
        if (rcu) return rcu;
        dim3 grid((nx + 255) / 256, ny);
        hipLaunchKernelGGL(k_mask_add, grid, dim3(256), 0, g_stream0, *plane, c->w0, dmask, channels, width, x0, y0, x1, y1, nx, ny,
                           transposed, is_rigmask, bias_factor);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(g_stream0));
        return 0;
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);
    dfree(dmask);
    return rc;
}

// ---- batch -----------------------------------------------------------------
// A lock-step group can be split over several HIP streams (sub-batches) that advance seam by seam side by side: a seam
// round is a latency-bound chain (backtrack, energy update, band update: ~0.7 ms at 4K whatever the batch size, on a
// few CUs) followed by the bandwidth-bound carve, so one sub-batch's chain runs under another's carve.  Measured at
// 64 x 4K with 4 streams: +12 % throughput (424k vs 377k Mseams*px/s), but every kernel then shares the chip -- a carve
// launch of 16 images takes 0.20 ms next to the others' kernels (2.6 TB/s algorithmic) instead of 0.13 ms alone -- and
// it needs a hardware queue per stream: with the HIP runtime's default of 4 queues per process (GPU_MAX_HW_QUEUES) the
// streams share queues and the same split is 30 % SLOWER.  lqrhip_set_sub_batches (bench.py --sub-batches) pins the
// number of streams; the default is automatic (lqrhip_sub_batches below).  DESIGN.md 4.11.
static int g_sub_batches = 0;           // 0: automatic (below)
extern "C" void lqrhip_set_sub_batches(int n) { g_sub_batches = n > 0 ? n : 0; }
// Streams a lock-step group of n carvers is split over.  Automatic: 4 for groups of 32 and more WHEN the process has the
// hardware queues for them -- the HIP runtime's GPU_MAX_HW_QUEUES (default 4, read when HIP initialises, shared with every
// other stream of the process) must be 8 or more; with fewer, streams share queues and the split is 30 % slower than
// one stream, so it is not made.
extern "C" int lqrhip_sub_batches(int n)
{
    int nb = g_sub_batches;
    if (nb == 0) {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        nb = (n >= 32 && q && atoi(q) >= 8) ? 4 : 1;
    }
    return n >= 2 * nb ? nb : 1;
}

extern "C" void lqrhip_batch_set_shared(LqrHipBatch *b, int shared) { b->shared = shared != 0; b->shared_n = shared > 1 ? shared : 1; }

extern "C" LqrHipBatch *lqrhip_batch_create(LqrHipCarver **carvers, int n)
{
    if (lqrhip_init() < 0 || n <= 0) return nullptr;
    LqrHipBatch *b = new LqrHipBatch();
    for (int i = 0; i < n; i++) {
        b->cs.push_back(carvers[i]);
        carvers[i]->batch = b;
    }
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **) &b->d_desc, sizeof(DevCarver) * n) != hipSuccess) {
        g_err = "batch_create failed";
        if (b->stream) (void) hipStreamDestroy(b->stream);
        for (auto *c : b->cs) if (c->batch == b) c->batch = nullptr;
        delete b;
        return nullptr;
    }
    g_live_batches.push_back(b);
    return b;
}

static void invalidate_all_batches(void)
{
    for (auto *b : g_live_batches) { b->exch_ntiles = 0; b->dirty = true; }
}

// after a failed resize: drain the stream, drop whatever the kernels recorded (it belongs to the failed call, the next
// resize must not report it), lay everything out afresh
extern "C" void lqrhip_batch_abort(LqrHipBatch *b)
{
    if (!b) return;
    (void) hipStreamSynchronize(b->stream);
    (void) hipGetLastError();
    if (g_dev_err_host) *g_dev_err_host = 0;
    invalidate_all_batches();
}

extern "C" void lqrhip_batch_destroy(LqrHipBatch *b)
{
    if (!b) return;
    g_live_batches.erase(std::remove(g_live_batches.begin(), g_live_batches.end(), b), g_live_batches.end());
    if (b->stream) { (void) hipStreamSynchronize(b->stream); (void) hipStreamDestroy(b->stream); }
    dfree(b->exch);
    for (auto *c : b->cs) if (c->batch == b) c->batch = nullptr;
    if (b->d_desc) (void) hipFree(b->d_desc);
    delete b;
}

extern "C" int lqrhip_batch_sync(LqrHipBatch *b)
{
    HIPCK(hipStreamSynchronize(b->stream));
    return check_dev_error();
}
extern "C" void *lqrhip_batch_stream(LqrHipBatch *b) { return (void *) b->stream; }

static DevCarver make_desc(const LqrHipCarver *c)
{
    DevCarver d;
    d.rgb0 = c->rgb0; d.vs = c->vs; d.bias0 = c->bias0; d.rig0 = c->rig0;
    d.pix = c->pix; d.en = c->en; d.m = c->m; d.least = c->least; d.m2 = c->m2; d.least2 = c->least2; d.bias = c->bias; d.rig = c->rig;
    d.seam_x = c->seam_x; d.seam_log = c->seam_log; d.flags = c->flags;
    return d;
}

static int batch_upload(LqrHipBatch *b)
{
    if (!b->dirty) return 0;
    std::vector<DevCarver> h;
    for (auto *c : b->cs) h.push_back(make_desc(c));
    HIPCK(hipStreamSynchronize(b->stream));
    HIPCK(hipMemcpy(b->d_desc, h.data(), sizeof(DevCarver) * h.size(), hipMemcpyHostToDevice));
    b->dirty = false;
    return 0;
}

static DpK make_dpk(const LqrHipDpParams *p, int ch)
{
    DpK k;
    k.delta = p->delta_x; k.use_rig = p->use_rigidity;
    memcpy(k.rigmap, p->rigidity_map, sizeof k.rigmap);
    k.nrg = p->nrg_func; k.radius = p->nrg_radius; k.w_start = p->w_start; k.ch = ch;
    return k;
}

struct ProfScope {
    ProfRec *rec = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t s;
    ProfScope(const char *name, hipStream_t stream, double bytes) : s(stream)
    {
        // every timed scope costs ~10 us of queue time (two event packets): mode 2 keeps that to the one
        // kernel whose launch time the bench line needs
        if (!g_prof || (g_prof == 2 && strcmp(name, "carve") != 0)) return;
        rec = &g_profrec[name];
        rec->bytes += bytes;
        (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
        (void) hipEventRecord(e0, s);
    }
    ~ProfScope()
    {
        if (!rec) return;
        (void) hipEventRecord(e1, s);
        rec->ev.emplace_back(e0, e1);
    }
};

extern "C" void lqrhip_prof_enable(int on) { g_prof = on; }
static int g_update_mode = -1;
// -1: by batch size (g_tiled_update_px); 0: band kernel (k_band_update_tw); 1: tiled full-width update whenever its
// grid fits; 2: the per-row-barrier band kernel (k_band_update_mw); 3: the generic one-wave band kernel + sweep
// (what delta_x > 2 runs on), whatever the parameters
extern "C" void lqrhip_set_update_mode(int mode) { g_update_mode = mode; }
#ifdef LQR_BAND_EXPERIMENTS
// which trapezoid band kernel update mode 0 (and the engine's own choice for large batches) means:
// 0 k_band_update_tw, 1 k_band_update_td<4 px per lane, 4 slots>, 2 k_band_update_td<2, 8>
static int g_band_kernel = 0;
extern "C" void lqrhip_set_band_kernel(int k) { g_band_kernel = k; }
static int g_band_variant = 0;
extern "C" void lqrhip_set_band_variant(int v) { g_band_variant = v; }
static int g_band_repeat = 1;
extern "C" void lqrhip_set_band_repeat(int n) { g_band_repeat = n; }
#ifdef LQR_BAND_TIMING
extern "C" int lqrhip_ls_hist(unsigned long long *out) { (void) hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ls_hist), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1; }
extern "C" int lqrhip_band_timing(unsigned long long *out) { (void) hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_band_dbg), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -1; }
#endif
#endif
static int g_dpp_limit_override = -1;
static int g_dpp_px_override = 0;       // test hook: 2 or 4 pins the persistent sweep's pixels per lane (0 = by batch size)
extern "C" void lqrhip_set_dp_persistent_px(int px) { g_dpp_px_override = (px == 2 || px == 4) ? px : 0; }
// -1: the occupancy-derived bound (dpp_resident_workgroups); >= 0: at most that many workgroups for the persistent
// tiled sweep -- 0 sends every full DP to k_dp_tile and every incremental update to a band kernel
extern "C" void lqrhip_set_dp_persistent_limit(int workgroups) { g_dpp_limit_override = workgroups; }
extern "C" void lqrhip_prof_reset(void)
{
    for (auto &kv : g_profrec) for (auto &e : kv.second.ev) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
    g_profrec.clear();
}
extern "C" int lqrhip_prof_get(const char *kernel, double *ms_total, long long *launches, double *bytes_total)
{
    auto it = g_profrec.find(kernel);
    *ms_total = 0; *launches = 0; *bytes_total = 0;
    if (it == g_profrec.end()) return 0;
    (void) hipDeviceSynchronize();
    for (auto &e : it->second.ev) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) *ms_total += ms;
    }
    *launches = (long long) it->second.ev.size();
    *bytes_total = it->second.bytes;
    return 0;
}

// Time during which at least one launch of `kernel` was running: with sub-batch streams launches overlap each other and
// other kernels, and bytes / (sum of launch times) would count the overlapped time twice.  Event times are taken relative
// to the first recorded event of the kernel (the GPU's clock is common to all streams).
extern "C" int lqrhip_prof_get_union(const char *kernel, double *ms_union)
{
    *ms_union = 0;
    auto it = g_profrec.find(kernel);
    if (it == g_profrec.end() || it->second.ev.empty()) return 0;
    (void) hipDeviceSynchronize();
    const hipEvent_t base = it->second.ev[0].first;
    std::vector<std::pair<float, float>> iv;
    for (auto &e : it->second.ev) {
        float a = 0, b = 0;
        // an event recorded before `base` on another stream gives a negative time: both orders are tried
        if (hipEventElapsedTime(&a, base, e.first) != hipSuccess) { (void) hipGetLastError(); float t = 0; if (hipEventElapsedTime(&t, e.first, base) == hipSuccess) a = -t; else (void) hipGetLastError(); }
        if (hipEventElapsedTime(&b, base, e.second) != hipSuccess) { (void) hipGetLastError(); float t = 0; if (hipEventElapsedTime(&t, e.second, base) == hipSuccess) b = -t; else (void) hipGetLastError(); }
        iv.emplace_back(a, b);
    }
    std::sort(iv.begin(), iv.end());
    float end = -1e30f;
    for (auto &p : iv) {
        if (p.first > end) { *ms_union += p.second - p.first; end = p.second; }
        else if (p.second > end) { *ms_union += p.second - end; end = p.second; }
    }
    return 0;
}

extern "C" int lqrhip_wk_init(LqrHipBatch *b)
{
    LqrHipCarver *c0 = b->cs[0];
    int w = c0->w0, h = c0->h0, rc;
    for (auto *c : b->cs) {
        if (c->w0 != w || c->h0 != h || c->ch != c0->ch) return LQRHIP_EARG;
        if ((rc = ensure_working(c, w, h))) return rc;
    }
    if ((rc = batch_upload(b))) return rc;
    for (auto *c : b->cs) c->frozen_epoch = 0;
    dim3 grid((c0->stride + 255) / 256, h, (unsigned) b->cs.size());
    hipLaunchKernelGGL(k_wk_init, grid, dim3(256), 0, b->stream, b->d_desc, w, h, c0->stride, c0->ch);
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int lqrhip_emap_build(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h)
{
    int rc;
    if ((rc = batch_upload(b))) return rc;
    LqrHipCarver *c0 = b->cs[0];
    dim3 grid((w + 255) / 256, h, (unsigned) b->cs.size());
    DpK k = make_dpk(p, c0->ch);
#define LAUNCH_EMAP(N) hipLaunchKernelGGL((k_emap_full<N>), grid, dim3(256), 0, b->stream, b->d_desc, k, w, h, c0->stride)
    NRG_DISPATCH(p->nrg_func, LAUNCH_EMAP)
#undef LAUNCH_EMAP
    HIPCK(hipGetLastError());
    return 0;
}


// E5 as H/32 dependent launches of one wave per 192-column tile (any batch size)
static int launch_dp_tiled(LqrHipBatch *b, const DpK &k, int w, int h, int lr)
{
    LqrHipCarver *c0 = b->cs[0];
    const dim3 grid((w + DPT_OWN - 1) / DPT_OWN, (unsigned) b->cs.size());
#define LAUNCH_TILE(LRV, RIGV) hipLaunchKernelGGL((k_dp_tile<LRV, RIGV>), grid, dim3(64), 0, b->stream, b->d_desc, k, w, h, c0->stride, y0)
    for (int y0 = 0; y0 < h; y0 += DPT_ROWS) {
        if (lr) { if (k.use_rig) LAUNCH_TILE(true, true); else LAUNCH_TILE(true, false); }
        else { if (k.use_rig) LAUNCH_TILE(false, true); else LAUNCH_TILE(false, false); }
    }
#undef LAUNCH_TILE
    HIPCK(hipGetLastError());
    return 0;
}

// can the persistent tiled sweep (k_dp_tile_p) take this batch?  Its tiles spin on each other, so the
// whole grid has to be resident at once.
// Pixels per lane of the persistent sweep for this batch: 2 while twice the tiles still fit the residency bound (the row
// chain is then ~33 instructions per wave instead of ~58; measured per 4K seam round, 2 vs 4 px per lane:
// 1 image 0.40 / 0.50 ms, 4: 0.45 / 0.55, 8: 0.58 / 0.62, 12: 0.76 / 0.77), else 4, 0 = not at all.
// `general`: delta_x = 2 and / or a rigidity mask (with rigidity): those instantiations exist for 2 px per lane only
static int dp_persistent_px(const LqrHipBatch *b, int w, bool general = false, int delta = 1)
{
    if (b->shared) return 0;
    const int bound = general ? g_dpp_max_wgs_general : g_dpp_max_wgs;
    const int limit = g_dpp_limit_override >= 0 ? std::min(g_dpp_limit_override, bound) : bound;
    const size_t n = b->cs.size();
    const int hh = b->cs[0]->wk_h;                            // the block index is DPP_BLK_BITS bits of the granule tag
    const int maxblk = (1 << DPP_BLK_BITS) - 1;
    if ((general || g_dpp_px_override != 4) && hh <= maxblk * dpp_rb(2, delta) && (size_t) ((w + dpp_own(2) - 1) / dpp_own(2)) * n <= (size_t) limit) return 2;
    const int limit4 = g_dpp_limit_override >= 0 ? std::min(g_dpp_limit_override, g_dpp_max_wgs_px4) : g_dpp_max_wgs_px4;
    if (!general && g_dpp_px_override != 2 && hh <= maxblk * dpp_halo(4) && (size_t) ((w + dpp_own(4) - 1) / dpp_own(4)) * n <= (size_t) limit4) return 4;
    return 0;
}
static bool dp_persistent_ok(const LqrHipBatch *b, int w) { return dp_persistent_px(b, w) != 0; }
// How many images of frame width `w` (the direction being carved) one lock-step batch may hold and still run delta_x = 2 /
// rigidity-mask carvers on the tiled kernels (k_dp_tile_p's general instantiations: one workgroup per 64 columns per image,
// all co-resident).  Beyond it such a batch would fall to the one-wave-per-image band kernel (~30x slower), so the host
// carves larger batches of such carvers group after group (host/lqr_carver.c, lqrx_carver_resize_batch).  0: no bound known.
extern "C" int lqrhip_general_batch_limit(int w)
{
    if (lqrhip_init() < 0 || w < 1) return 0;
    const int limit = g_dpp_limit_override >= 0 ? std::min(g_dpp_limit_override, g_dpp_max_wgs_general) : g_dpp_max_wgs_general;
    return limit / ((w + dpp_own(2) - 1) / dpp_own(2));
}

// E5 (UPDATE = false) or the full-width form of E9 (UPDATE = true) as one persistent launch
template <bool UPDATE>
static int launch_dp_persistent(LqrHipBatch *b, const DpK &k, int w, int h, int lr)
{
    LqrHipCarver *c0 = b->cs[0];
    const size_t n = b->cs.size();
    bool rigm = false;
    for (auto *c : b->cs) rigm |= (c->rig != nullptr);
    rigm = rigm && k.use_rig;                                  // without rigidity the mask multiplies nothing
    const bool general = k.delta != 1 || rigm;
    if (k.delta < 1 || k.delta > 4) return LQRHIP_EARG;
    const int px = dp_persistent_px(b, w, general, k.delta);
    if (!px) return LQRHIP_EARG;
    const int ntiles = (w + dpp_own(px) - 1) / dpp_own(px);
    int rc;
    const size_t need_elems = ((size_t) ntiles * dpp_ex_tile(px) + 8) * n;
    if (b->exch_elems < need_elems) {
        HIPCK(hipStreamSynchronize(b->stream));
        dfree(b->exch);
        b->exch_elems = 0;
        if ((rc = dmalloc(&b->exch, need_elems))) return rc;
        b->exch_elems = need_elems;
        b->exch_ntiles = 0;
    }
    if (b->exch_ntiles != ntiles || b->exch_n != (int) n || b->exch_px != px) {
        // (re)lay the exchange area out: tags and finished-tile counters start at 0 (afterwards nothing is ever
        // cleared: tags carry the launch epoch, the last tile re-arms the counter)
        HIPCK(hipMemsetAsync(b->exch, 0, need_elems * sizeof(unsigned long long), b->stream));
        b->exch_ntiles = ntiles; b->exch_n = (int) n; b->exch_px = px;
    }
    if (UPDATE) {
        // second planes, allocated on first use -- per carver: a batch may mix carvers that already went
        // through a tiled update on their own with fresh ones
        bool grew = false;
        for (auto *c : b->cs) {
            if (c->m2 && c->least2) continue;
            if (!grew) { HIPCK(hipStreamSynchronize(b->stream)); grew = true; }
            const size_t pe = (size_t) c->stride * (c->wk_h + 1) + 1024;
            const bool fresh_m2 = !c->m2, fresh_l2 = !c->least2;
            if (!c->m2 && (rc = dmalloc(&c->m2, pe))) return rc;
            if (!c->least2 && (rc = dmalloc(&c->least2, pe))) return rc;
            // like the first planes (ensure_working): nothing in them depends on what the block held before
            if (fresh_m2) HIPCK(hipMemsetAsync(c->m2, 0, pe * sizeof(float), b->stream));
            if (fresh_l2) HIPCK(hipMemsetAsync(c->least2, 0, pe, b->stream));
        }
        if (grew) {
            b->dirty = true;
            if ((rc = batch_upload(b))) return rc;
        }
    }
    const int epoch = 1 + ((b->tile_epoch++) % ((1 << (31 - DPP_BLK_BITS)) - 2));          // never 0; above the block index in the 32-bit tag
    const dim3 grid(ntiles, (unsigned) n);
#define LAUNCH_TILE(PXV, LRV, RIGV) hipLaunchKernelGGL((k_dp_tile_p<PXV, LRV, RIGV, UPDATE>), grid, dim3(64 * DPP_W), 0, b->stream, b->d_desc, k, w, h, c0->stride, b->exch, epoch, g_dev_err)
#define LAUNCH_TILE_PX(PXV)                                                                 \
    do {                                                                                    \
        if (lr) { if (k.use_rig) LAUNCH_TILE(PXV, true, true); else LAUNCH_TILE(PXV, true, false); }     \
        else { if (k.use_rig) LAUNCH_TILE(PXV, false, true); else LAUNCH_TILE(PXV, false, false); }      \
    } while (0)
#define LAUNCH_TILE_G(LRV, RIGV, DV, RMV) hipLaunchKernelGGL((k_dp_tile_p<2, LRV, RIGV, UPDATE, DV, RMV>), grid, dim3(64 * DPP_W), 0, b->stream, b->d_desc, k, w, h, c0->stride, b->exch, epoch, g_dev_err)
#define LAUNCH_TILE_G_LR(RIGV, DV, RMV) do { if (lr) LAUNCH_TILE_G(true, RIGV, DV, RMV); else LAUNCH_TILE_G(false, RIGV, DV, RMV); } while (0)
    if (general) {
#define LAUNCH_TILE_G_D(DV) do { if (!k.use_rig) LAUNCH_TILE_G_LR(false, DV, false); else if (!rigm) LAUNCH_TILE_G_LR(true, DV, false); else LAUNCH_TILE_G_LR(true, DV, true); } while (0)
        if (k.delta == 1) LAUNCH_TILE_G_LR(true, 1, true);
        else if (k.delta == 2) LAUNCH_TILE_G_D(2);
        else if (k.delta == 3) LAUNCH_TILE_G_D(3);
        else LAUNCH_TILE_G_D(4);
#undef LAUNCH_TILE_G_D
    }
    else if (px == 2) LAUNCH_TILE_PX(2); else LAUNCH_TILE_PX(4);
#undef LAUNCH_TILE_G_LR
#undef LAUNCH_TILE_G
#undef LAUNCH_TILE_PX
#undef LAUNCH_TILE
    HIPCK(hipGetLastError());
    if (UPDATE)       // the kernel's last tile swapped the pointers in the device descriptors: mirror it
        for (auto *c : b->cs) { std::swap(c->m, c->m2); std::swap(c->least, c->least2); }
    return 0;
}

template <bool UPDATE>
static int launch_dp(LqrHipBatch *b, const DpK &k, int w, int h, int lr)
{
    LqrHipCarver *c0 = b->cs[0];
    if (!UPDATE) {
        bool rigm = false;
        for (auto *c : b->cs) rigm |= (c->rig != nullptr);
        rigm = rigm && k.use_rig;
        if (k.delta == 1 && !rigm) return dp_persistent_ok(b, w) ? launch_dp_persistent<false>(b, k, w, h, lr) : launch_dp_tiled(b, k, w, h, lr);
        if (k.delta >= 1 && k.delta <= 4 && dp_persistent_px(b, w, true, k.delta)) return launch_dp_persistent<false>(b, k, w, h, lr);
    }
    int pxt = (w + DP_THREADS - 1) / DP_THREADS;
    size_t lds = (size_t) 2 * ((w + 3) & ~3) * sizeof(float);
    dim3 grid((unsigned) b->cs.size()), block(DP_THREADS);
#define LAUNCH_DP(P)                                                                                                  \
    do {                                                                                                              \
        if (lds > 64 * 1024)                                                                                          \
            HIPCK(hipFuncSetAttribute((const void *) k_dp_sweep<P, UPDATE>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int) lds));                                                                    \
        hipLaunchKernelGGL((k_dp_sweep<P, UPDATE>), grid, block, lds, b->stream, b->d_desc, k, w, h, c0->stride, lr); \
    } while (0)
    if (pxt <= 1) LAUNCH_DP(1);
    else if (pxt <= 2) LAUNCH_DP(2);
    else if (pxt <= 4) LAUNCH_DP(4);
    else if (pxt <= 8) LAUNCH_DP(8);
    else if (pxt <= 16) LAUNCH_DP(16);
    else { g_err = "image wider than 16384 px is not supported"; return LQRHIP_EARG; }
#undef LAUNCH_DP
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int lqrhip_mmap_build(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h, int leftright)
{
    int rc;
    if ((rc = batch_upload(b))) return rc;
    if (p->delta_x > LQRHIP_MAX_DELTA) return LQRHIP_EARG;
    ProfScope ps("dp_sweep", b->stream, 9.0 * w * h * b->cs.size());
    return launch_dp<false>(b, make_dpk(p, b->cs[0]->ch), w, h, leftright);
}

#ifndef FROZEN_LAG_MAX
#define FROZEN_LAG_MAX 128      // seams the frozen planes may lag behind before they are compacted
#endif

// remove seams [epoch, to) from the frozen planes of every carver of the batch
static int frozen_catchup(LqrHipBatch *b, int to, int w_at_to, int h)
{
    LqrHipCarver *c0 = b->cs[0];
    const int from = c0->frozen_epoch;
    if (to <= from) return 0;
    const int w_from = w_at_to + (to - from);
    size_t lds = (size_t) (to - from) * sizeof(int) + (size_t) w_from + 16;
    hipLaunchKernelGGL(k_frozen_catchup, dim3(h, (unsigned) b->cs.size()), dim3(256), lds, b->stream, b->d_desc, from, to, w_from, h,
                       c0->stride);
    HIPCK(hipGetLastError());
    for (auto *c : b->cs) c->frozen_epoch = to;
    return 0;
}

// batches up to this many pixels use the tiled full-width update (measured break-even with the band kernel at 4K, Mseams*px/s
// tiled / band: 7 images 118 k / 97 k, 8: 130 / 109, 9: 119 / 121, 12: 130 / 139+, 16: 158 / 175+)
// Tiles per image for k_band_tiles (0: not usable here).  All of a group's sub-batches run side by side, each with a
// persistent grid of its own: together they must fit the residency bound of the 2-px tiled instantiations (same register
// budget: the occupancy query below covers k_band_tiles).
static int g_band_tiles = -1;            // -1: automatic; 0: never; n: force n tiles per image (tests)
extern "C" void lqrhip_set_band_tiles(int t) { g_band_tiles = t; }
static int band_tiles_T(const LqrHipBatch *b, int h)
{
    if (g_band_tiles == 0 || (h + 31) / 32 > BT_MAX_BLK) return 0;
    const int limit = g_dpp_limit_override >= 0 ? std::min(g_dpp_limit_override, g_dpp_max_wgs_tiles) : g_dpp_max_wgs_tiles;
    const int per_batch = limit / std::max(b->shared_n, 1);
    int T = std::min(BT_T_MAX, per_batch / (int) std::max<size_t>(b->cs.size(), 1));
    if (g_band_tiles > 0) T = std::min(T, g_band_tiles);
    return T >= (g_band_tiles > 0 ? 1 : 8) ? T : 0;
}
// T workgroups per image: two thirds of them base tiles around the seam, the rest reserve tiles that an edge tile wakes
// when the band comes near the edge of the set (lqrhip_set_band_tiles_reserve pins the number of reserves for tests)
static int g_band_tiles_rsv = -1;
extern "C" void lqrhip_set_band_tiles_reserve(int n) { g_band_tiles_rsv = n; }
static int launch_band_tiles(LqrHipBatch *b, const DpK &k, int w, int h, int lr, int T)
{
    LqrHipCarver *c0 = b->cs[0];
    const size_t n = b->cs.size();
    int rc;
    int n_rsv = g_band_tiles_rsv >= 0 ? std::min(g_band_tiles_rsv, T - 1) : T / 3;
    n_rsv = std::max(0, std::min(n_rsv, BT_HDR - 2));
    const int t_base = T - n_rsv;
    const int ntiles_img = (w + 63) / 64;
    const size_t need_elems = ((size_t) ntiles_img * dpp_ex_tile(2) + 2 * BT_HDR) * n;
    if (b->exch_elems < need_elems) {
        HIPCK(hipStreamSynchronize(b->stream));
        dfree(b->exch);
        b->exch_elems = 0;
        if ((rc = dmalloc(&b->exch, need_elems))) return rc;
        b->exch_elems = need_elems;
        b->exch_ntiles = 0;
    }
    if (b->exch_ntiles != ntiles_img || b->exch_n != (int) n || b->exch_px != 102) {       // 102: this kernel's layout and tags
        HIPCK(hipMemsetAsync(b->exch, 0, need_elems * sizeof(unsigned long long), b->stream));
        b->exch_ntiles = ntiles_img; b->exch_n = (int) n; b->exch_px = 102;
    }
    // (the images' headers -- tiles finished, tickets drawn, requests -- start every launch at zero: there are two sets, a launch
    // uses one and clears the other for the next launch; granules and request words carry the epoch)
    const int hset = (b->bt_launches++) & 1;
    const int epoch = 1 + ((b->tile_epoch++) % ((1 << 19) - 2));           // never 0; 19 bits above changed / active bits and block index
    const dim3 grid(T, (unsigned) n);
#define LAUNCH_BT(LRV, RIGV) hipLaunchKernelGGL((k_band_tiles<LRV, RIGV>), grid, dim3(128), 0, b->stream, b->d_desc, k, w, h, c0->stride, b->exch, epoch, g_dev_err, t_base, hset)
    if (lr) { if (k.use_rig) LAUNCH_BT(true, true); else LAUNCH_BT(true, false); }
    else { if (k.use_rig) LAUNCH_BT(false, true); else LAUNCH_BT(false, false); }
#undef LAUNCH_BT
    HIPCK(hipGetLastError());
    return 0;
}
static const long long g_tiled_update_px = 8LL * 3840 * 2160;

// One seam of a lock-step batch: k_vpath* (pick + backtrack, publishes the side to move) -> k_carve ->
// k_emap_update -> one form of update_mmap (or the full DP after a side switch), all on the batch's stream.
static int seam_step_impl(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h, int log_index, int leftright_pick,
                          int full_rebuild, int leftright_next);
// LQRHIP_DUMP=<prefix> (debugging aid): after every seam step the first image's seam, flags and DP planes go to
// <prefix>_<call>.bin -- header {w, h, stride, log_index, FLAG_COUNT}, flags, seam_x[h], m[stride * h], least[stride * h]
extern "C" int lqrhip_seam_step(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h, int log_index, int leftright_pick,
                                int full_rebuild, int leftright_next)
{
    const int rc = seam_step_impl(b, p, w, h, log_index, leftright_pick, full_rebuild, leftright_next);
    static const char *dump = getenv("LQRHIP_DUMP");
    if (rc == 0 && dump) {
        static int call = 0;
        LqrHipCarver *c = b->cs[0];
        HIPCK(hipStreamSynchronize(b->stream));
        const size_t np = (size_t) c->stride * h;
        std::vector<int> hdr = {w, h, c->stride, log_index, FLAG_COUNT}, flags(FLAG_COUNT), seam(h);
        std::vector<float> m(np);
        std::vector<int8_t> least(np);
        HIPCK(hipMemcpy(flags.data(), c->flags, FLAG_COUNT * sizeof(int), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(seam.data(), c->seam_x, (size_t) h * sizeof(int), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(m.data(), c->m, np * sizeof(float), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(least.data(), c->least, np, hipMemcpyDeviceToHost));
        char path[512];
        snprintf(path, sizeof path, "%s_%04d.bin", dump, call++);
        if (FILE *f = fopen(path, "wb")) {
            fwrite(hdr.data(), sizeof(int), hdr.size(), f); fwrite(flags.data(), sizeof(int), flags.size(), f);
            fwrite(seam.data(), sizeof(int), seam.size(), f); fwrite(m.data(), sizeof(float), np, f); fwrite(least.data(), 1, np, f);
            fclose(f);
        }
    }
    return rc;
}

static int seam_step_impl(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h, int log_index, int leftright_pick,
                          int full_rebuild, int leftright_next)
{
    int rc;
    LqrHipCarver *c0 = b->cs[0];
    for (auto *c : b->cs)
        if (log_index >= c->log_cap) return LQRHIP_EARG;
    if ((rc = batch_upload(b))) return rc;
    const unsigned n = (unsigned) b->cs.size();
    DpK k = make_dpk(p, c0->ch);
    const int stride = c0->stride;
    {
        ProfScope ps("vpath", b->stream, 0);
        if (p->delta_x == 1)
            hipLaunchKernelGGL(k_vpath1<1>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index);
        else if (p->delta_x == 2)
            hipLaunchKernelGGL(k_vpath1<2>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index);
        else if (p->delta_x == 3)
            hipLaunchKernelGGL(k_vpath1<3>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index);
        else if (p->delta_x == 4)
            hipLaunchKernelGGL(k_vpath1<4>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index);
        else
            hipLaunchKernelGGL(k_vpath, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, p->delta_x,
                               log_index);
    }
    const int wnew = w - 1;
    const int move_dp = (wnew > 1 && !full_rebuild) ? 1 : 0;
    bool has_rigmask = false;
    for (auto *c : b->cs) has_rigmask |= (c->rig != nullptr);
    {
        // algorithmic bytes of one carve launch: read + write of one 4-byte
        // plane over the half of each row right of the seam = 8 B * w*h/2 per image
        ProfScope ps("carve", b->stream, 4.0 * (double) w * h * n);
        hipLaunchKernelGGL(k_carve, dim3((h + 3) / 4, n), dim3(256), 0, b->stream, b->d_desc, w, h, stride, p->delta_x, move_dp);
    }
    if (wnew <= 1) {            // liblqr's finish_vsmap case: nothing left to update
        HIPCK(hipGetLastError());
        return 0;
    }
    {
        ProfScope ps("emap_update", b->stream, 0);
        // the energy update walks the seam log back to the frozen frame (O(lag) per sample); compacting the frozen planes
        // costs a pass over them.  Few images: the walk is on the critical path and the pass is cheap -> short lag
        const int lag_max = n <= 4 ? FROZEN_LAG_MAX / 4 : FROZEN_LAG_MAX;
        if (log_index + 1 - c0->frozen_epoch > lag_max && (rc = frozen_catchup(b, log_index + 1, wnew, h))) return rc;
        const int epoch = c0->frozen_epoch;
#define LAUNCH_EUPD_NT(N, NT) hipLaunchKernelGGL((k_emap_update<N, NT>), dim3((h + EU_ROWS - 1) / EU_ROWS, n), dim3(64), 0, b->stream, b->d_desc, k, wnew, h, stride, log_index, epoch)
#define LAUNCH_EUPD(N) do { if (p->delta_x <= 2) LAUNCH_EUPD_NT(N, 12); else if (p->delta_x <= 8) LAUNCH_EUPD_NT(N, 36); else LAUNCH_EUPD_NT(N, 68); } while (0)
        NRG_DISPATCH(p->nrg_func, LAUNCH_EUPD)
#undef LAUNCH_EUPD
#undef LAUNCH_EUPD_NT
    }
    if (full_rebuild) {
        ProfScope ps("dp_sweep", b->stream, 9.0 * wnew * h * n);
        if ((rc = launch_dp<false>(b, k, wnew, h, leftright_next))) return rc;
        HIPCK(hipGetLastError());
        return 0;
    }
    // How E9 (update_mmap) runs.  Small batches: the whole chip recomputing every row (tiled full-width keep-rule
    // sweep) beats the one-workgroup-per-image band walk; for large batches its 14 B/px of traffic would not.
    // "plain": delta_x = 1 and no rigidity mask that matters -- every fast kernel.  delta_x = 2 and rigidity masks run on the
    // tiled full-width update (k_dp_tile_p's general instantiations) whenever its grid fits; only beyond that do they fall
    // to the one-wave-per-image band kernel and the one-workgroup-per-image sweep (measured at 8K: 37x slower)
    const bool rigm = has_rigmask && p->use_rigidity;
    const bool fast_ok = p->delta_x == 1 && !rigm && g_update_mode != 3;
    const bool tiled_update = fast_ok ? ((g_update_mode < 0 ? (size_t) n * (size_t) w * (size_t) h <= (size_t) g_tiled_update_px : g_update_mode == 1) &&
                                         dp_persistent_ok(b, w))
                                      : (p->delta_x >= 1 && p->delta_x <= 4 && g_update_mode != 0 && g_update_mode != 2 && g_update_mode != 3 && dp_persistent_px(b, w, true, p->delta_x) != 0);
    // Batches of 8 to ~40 images: the band spread over several CUs per image (k_band_tiles).  Measured (Mseams*px/s, 4K, tiles /
    // k_band_update_tw or the full-width tiled update): 4 images 80 k / 93 k (the full-width tiled update stays), 8: 145 / 127,
    // 12: 195 / 152, 16: 246 / 193, 24: 294 / 260, 32: 382 / 341, 40: 423 / 398, 48: 434 / 448, 64: 488 / 510-540 -- beyond ~500
    // resident tile workgroups the carves of the sibling streams are starved of registers, so large groups
    // keep k_band_update_tw.
    {
        const int T = (fast_ok && (g_update_mode < 0 || g_update_mode == 4)) ? band_tiles_T(b, h) : 0;
        const size_t group_images = (size_t) n * (size_t) std::max(b->shared_n, 1);
        if (T > 0 && (g_update_mode == 4 || (group_images >= 8 && group_images * (size_t) T <= 480))) {
            {
                ProfScope ps("band_update", b->stream, 0);
                if ((rc = launch_band_tiles(b, k, wnew, h, leftright_next, T))) return rc;
            }
            ProfScope ps("dp_update", b->stream, 0);
            if ((rc = launch_dp<true>(b, k, wnew, h, leftright_next))) return rc;
            HIPCK(hipGetLastError());
            return 0;
        }
    }
    if (tiled_update) {
        ProfScope ps("dp_update_tiled", b->stream, 0);
        if ((rc = launch_dp_persistent<true>(b, k, wnew, h, leftright_next))) return rc;
        HIPCK(hipGetLastError());
        return 0;
    }
    const bool fast_band = fast_ok && (size_t) h * sizeof(int) <= 60 * 1024;
    // the trapezoid-wave band kernel takes rows up to ~4200 px (wider rows: the changes outgrow its 896-column window
    // too often, and an 8-slot build spills registers); beyond that, and in update mode 2, k_band_update_mw
    const bool band_tw = fast_band && g_update_mode != 2 && wnew <= 4200 && (size_t) 2 * h * sizeof(int) <= 64 * 1024;
#ifdef LQR_BAND_EXPERIMENTS
    if (band_tw && g_band_kernel == 3 && ls_lds_bytes(2, 7, h) <= 160 * 1024) {
        ProfScope ps("band_update", b->stream, 0);
        const size_t lds = ls_lds_bytes(2, 7, h);
#define LAUNCH_LS(LRV, RIGV) do { HIPCK(hipFuncSetAttribute((const void *) k_band_update_ls<2, 7, LRV, RIGV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
                                  hipLaunchKernelGGL((k_band_update_ls<2, 7, LRV, RIGV>), dim3(n), dim3(ls_threads(7)), lds, b->stream, b->d_desc, k, wnew, h, stride, g_dev_err, g_band_variant); } while (0)
        if (leftright_next) { if (p->use_rigidity) LAUNCH_LS(true, true); else LAUNCH_LS(true, false); }
        else { if (p->use_rigidity) LAUNCH_LS(false, true); else LAUNCH_LS(false, false); }
#undef LAUNCH_LS
    } else if (band_tw && g_band_kernel != 0) {
        ProfScope ps("band_update", b->stream, 0);
#define LAUNCH_TD(PXV, NWV, LRV, RIGV) hipLaunchKernelGGL((k_band_update_td<PXV, NWV, LRV, RIGV>), dim3(n), dim3(128 * NWV), (size_t) 2 * h * sizeof(int), b->stream, b->d_desc, k, wnew, h, stride, g_dev_err, g_band_variant)
#define LAUNCH_TD_LR(PXV, NWV) do { if (leftright_next) { if (p->use_rigidity) LAUNCH_TD(PXV, NWV, true, true); else LAUNCH_TD(PXV, NWV, true, false); } \
                                    else { if (p->use_rigidity) LAUNCH_TD(PXV, NWV, false, true); else LAUNCH_TD(PXV, NWV, false, false); } } while (0)
        for (int rep = 0; rep < g_band_repeat; rep++) { if (g_band_kernel == 1) LAUNCH_TD_LR(4, 4); else LAUNCH_TD_LR(2, 8); }
#undef LAUNCH_TD_LR
#undef LAUNCH_TD
    } else
#endif
    if (band_tw) {
        ProfScope ps("band_update", b->stream, 0);
#define LAUNCH_TW(LRV, RIGV) hipLaunchKernelGGL((k_band_update_tw<4, LRV, RIGV>), dim3(n), dim3(128 * 4), (size_t) 2 * h * sizeof(int), b->stream, b->d_desc, k, wnew, h, stride, g_dev_err)
        if (leftright_next) { if (p->use_rigidity) LAUNCH_TW(true, true); else LAUNCH_TW(true, false); }
        else { if (p->use_rigidity) LAUNCH_TW(false, true); else LAUNCH_TW(false, false); }
#undef LAUNCH_TW
    } else if (fast_band) {
        ProfScope ps("band_update", b->stream, 0);
#define LAUNCH_MW(NWV, LRV, RIGV) hipLaunchKernelGGL((k_band_update_mw<2, NWV, 8, LRV, RIGV>), dim3(n), dim3(64 * NWV), (size_t) h * sizeof(int), b->stream, b->d_desc, k, wnew, h, stride)
#define LAUNCH_MW_N(LRV, RIGV) do { if (wnew > 4200) LAUNCH_MW(16, LRV, RIGV); /* 8K: dirty regions up to ~900 px */ else LAUNCH_MW(8, LRV, RIGV); } while (0)
        if (leftright_next) { if (p->use_rigidity) LAUNCH_MW_N(true, true); else LAUNCH_MW_N(true, false); }
        else { if (p->use_rigidity) LAUNCH_MW_N(false, true); else LAUNCH_MW_N(false, false); }
#undef LAUNCH_MW_N
#undef LAUNCH_MW
    } else {
        ProfScope ps("band_update", b->stream, 0);
        hipLaunchKernelGGL(k_band_update, dim3(n), dim3(64), 0, b->stream, b->d_desc, k, wnew, h, stride, leftright_next);
    }
    {
        // rows the band kernel handed over (flags[FLAG_OVF_ROW] .. h): the keep rule over the full width
        ProfScope ps("dp_update", b->stream, 0);
        if ((rc = launch_dp<true>(b, k, wnew, h, leftright_next))) return rc;
    }
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int lqrhip_seam_log_reserve(LqrHipBatch *b, int n_seams, int h)
{
    int rc;
    for (auto *c : b->cs) if ((rc = ensure_log(c, n_seams, h))) return rc;
    return 0;
}

extern "C" int lqrhip_vs_commit(LqrHipBatch *b, int w0, int h0, int wc0, int n_seams, int first_level, int finish)
{
    int rc;
    if ((rc = batch_upload(b))) return rc;
    size_t lds = ((size_t) n_seams + wc0) * sizeof(int);
    if (lds > 150 * 1024) { g_err = "vs_commit: session too large for LDS"; return LQRHIP_EARG; }
    if (lds > 64 * 1024)
        HIPCK(hipFuncSetAttribute((const void *) k_vs_commit, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipLaunchKernelGGL(k_vs_commit, dim3(h0, (unsigned) b->cs.size()), dim3(256), lds, b->stream, b->d_desc, w0, h0, wc0, n_seams,
                       first_level, finish);
    HIPCK(hipGetLastError());
    // the session is over: bring the frozen planes to the carved frame, the log restarts at 0
    if ((rc = frozen_catchup(b, n_seams, wc0 - n_seams, h0))) return rc;
    for (auto *c : b->cs) c->frozen_epoch = 0;
    return 0;
}

// E14 / E11: every carver of the batch (roots and their attached carvers) goes through ONE launch (a job table in device
// memory, one grid slice of blocks per job); the new planes replace the old ones after its synchronisation.  Everything
// staged for the pass is owned by a PlaneJobs object until then: any error return gives all of it back to the pool.
struct PlaneJob {
    LqrHipCarver *c;
    uint8_t *nrgb;
    float *nbias, *nrig;
};
struct PlaneJobs {
    std::vector<PlaneJob> jobs;
    std::vector<InflateDev> dev;
    std::vector<int32_t *> new_vs;          // one per root (may be null)
    InflateDev *d_jobs = nullptr;
    bool committed = false;
    ~PlaneJobs()
    {
        dfree(d_jobs);
        if (committed) return;
        for (auto &j : jobs) { dfree(j.nrgb); dfree(j.nbias); dfree(j.nrig); }
        for (auto *&v : new_vs) dfree(v);
    }
    // stage the output planes of carver c: n1 pixels each
    int add(LqrHipCarver *c, const int32_t *vs_old, int32_t *nvs, size_t n1)
    {
        PlaneJob j{c, nullptr, nullptr, nullptr};
        int rc = dmalloc(&j.nrgb, n1 * c->ch);
        if (!rc && c->bias0) rc = dmalloc(&j.nbias, n1);
        if (!rc && c->rig0) rc = dmalloc(&j.nrig, n1);
        jobs.push_back(j);                  // owned from here on, also when rc != 0
        if (rc) return rc;
        dev.push_back(InflateDev{c->rgb0, vs_old, c->bias0, c->rig0, j.nrgb, nvs, j.nbias, j.nrig, c->ch});
        return 0;
    }
    int upload(hipStream_t s)
    {
        int rc = dmalloc(&d_jobs, dev.size());
        if (rc) return rc;
        HIPCK(hipMemcpyAsync(d_jobs, dev.data(), dev.size() * sizeof(InflateDev), hipMemcpyHostToDevice, s));
        return 0;
    }
    // after the pass has completed: the new base planes become the carvers'
    void commit()
    {
        for (auto &j : jobs) {
            dfree(j.c->rgb0); j.c->rgb0 = j.nrgb;
            if (j.nbias) { dfree(j.c->bias0); j.c->bias0 = j.nbias; }
            if (j.nrig) { dfree(j.c->rig0); j.c->rig0 = j.nrig; }
        }
        committed = true;
    }
};

extern "C" int lqrhip_inflate(LqrHipBatch *b, int w0, int h0, int l, int max_level)
{
    int rc;
    const int w1 = w0 + l - max_level + 1;
    PlaneJobs pj;
    for (auto *c : b->cs) {
        int32_t *nvs = nullptr;
        if ((rc = dmalloc(&nvs, (size_t) w1 * h0))) return rc;
        pj.new_vs.push_back(nvs);
        for (auto *a : c->aux)
            if ((rc = pj.add(a, c->vs, nullptr, (size_t) w1 * h0))) return rc;
        if ((rc = pj.add(c, c->vs, nvs, (size_t) w1 * h0))) return rc;
    }
    if ((rc = pj.upload(b->stream))) return rc;
    hipLaunchKernelGGL(k_inflate, dim3(h0, (unsigned) pj.dev.size()), dim3(256), 0, b->stream, pj.d_jobs, w0, w1, l, max_level);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(b->stream));
    pj.commit();
    for (auto &j : pj.jobs) j.c->w0 = w1;
    size_t i = 0;
    for (auto *c : b->cs) {
        dfree(c->vs);
        c->vs = pj.new_vs[i++];
        for (auto *a : c->aux) a->vs = c->vs;
    }
    b->dirty = true;
    return 0;
}

extern "C" int lqrhip_flatten(LqrHipBatch *b, int w0, int h0, int w, int level)
{
    int rc;
    PlaneJobs pj;
    for (auto *c : b->cs) {
        int32_t *nvs = nullptr;                 // the flat carver's visibility map: all zero
        if ((rc = dmalloc(&nvs, (size_t) w * h0))) return rc;
        pj.new_vs.push_back(nvs);
        HIPCK(hipMemsetAsync(nvs, 0, (size_t) w * h0 * sizeof(int32_t), b->stream));
        for (auto *a : c->aux)
            if ((rc = pj.add(a, c->vs, nullptr, (size_t) w * h0))) return rc;
        if ((rc = pj.add(c, c->vs, nullptr, (size_t) w * h0))) return rc;
    }
    if ((rc = pj.upload(b->stream))) return rc;
    hipLaunchKernelGGL(k_compact_jobs, dim3(h0, (unsigned) pj.dev.size()), dim3(256), 0, b->stream, pj.d_jobs, w0, w, level);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(b->stream));
    pj.commit();
    for (auto &j : pj.jobs) j.c->w0 = w;
    size_t i = 0;
    for (auto *c : b->cs) {
        dfree(c->vs);
        c->vs = pj.new_vs[i++];
        for (auto *a : c->aux) a->vs = c->vs;
    }
    b->dirty = true;
    return 0;
}

extern "C" int lqrhip_transpose(LqrHipBatch *b, int w, int h)
{
    int rc;
    PlaneJobs pj;
    for (auto *c : b->cs) {
        for (auto *a : c->aux)
            if ((rc = pj.add(a, nullptr, nullptr, (size_t) w * h))) return rc;
        if ((rc = pj.add(c, nullptr, nullptr, (size_t) w * h))) return rc;
        HIPCK(hipMemsetAsync(c->vs, 0, (size_t) w * h * sizeof(int32_t), b->stream));   // flat carver: all zero already
    }
    if ((rc = pj.upload(b->stream))) return rc;
    hipLaunchKernelGGL(k_transpose, dim3((w + 31) / 32, (h + 31) / 32, (unsigned) pj.dev.size()), dim3(32, 8), 0, b->stream, pj.d_jobs, w, h);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(b->stream));
    pj.commit();
    for (auto &j : pj.jobs) { j.c->w0 = h; j.c->h0 = w; }
    b->dirty = true;
    return 0;
}

// ---- read-back ---------------------------------------------------------------
extern "C" int lqrhip_read_visible(LqrHipCarver *c, int w0, int h0, int w, int level, unsigned char *out)
{
    int rc = batch_sync_of(c);
    if (rc) return rc;
    uint8_t *d = nullptr;
    size_t n = (size_t) w * h0 * c->ch;
    if ((rc = dmalloc(&d, n))) return rc;
    auto run = [&]() -> int {
        hipLaunchKernelGGL(k_compact, dim3(h0), dim3(256), 0, g_stream0, c->rgb0, c->vs, (const float *) nullptr, (const float *) nullptr,
                           d, (float *) nullptr, (float *) nullptr, (int32_t *) nullptr, w0, w, c->ch, level, 0);
        HIPCK(hipGetLastError());
        return d2h_staged(out, d, n);
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);
    dfree(d);
    return rc;
}

extern "C" int lqrhip_read_visible_device(LqrHipCarver *c, int w0, int h0, int w, int level, void *device_out)
{
    int rc = batch_sync_of(c);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compact, dim3(h0), dim3(256), 0, g_stream0, c->rgb0, c->vs, (const float *) nullptr, (const float *) nullptr,
                       (uint8_t *) device_out, (float *) nullptr, (float *) nullptr, (int32_t *) nullptr, w0, w, c->ch, level, 0);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(g_stream0));
    return 0;
}

extern "C" int lqrhip_mask_line_max(const unsigned char *mask, int channels, int width, int height, int a0, int b0, int n_lines,
                                    int line_len, int direction)
{
    if (lqrhip_init() < 0) return LQRHIP_EHIP;
    if (n_lines <= 0 || line_len <= 0) return 0;
    uint8_t *d = nullptr;
    int *dout = nullptr;
    int rc, result = 0;
    size_t bytes = (size_t) width * height * channels;
    if ((rc = dmalloc(&d, bytes)) || (rc = dmalloc(&dout, 1))) { dfree(d); return rc; }
    auto run = [&]() -> int {
        HIPCK(hipMemcpyAsync(d, mask, bytes, hipMemcpyHostToDevice, g_stream0));
        HIPCK(hipMemsetAsync(dout, 0, sizeof(int), g_stream0));
        hipLaunchKernelGGL(k_mask_line_max, dim3(n_lines), dim3(256), 0, g_stream0, d, channels, width, a0, b0, line_len, direction, dout);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(&result, dout, sizeof(int), hipMemcpyDeviceToHost, g_stream0));
        HIPCK(hipStreamSynchronize(g_stream0));
        return 0;
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);
    dfree(d); dfree(dout);
    return rc ? rc : result;
}

extern "C" void lqrhip_pool_trim(void)
{
    (void) hipDeviceSynchronize();
    for (auto &kv : g_pool_free) { (void) hipFree(kv.second); g_pool_size.erase(kv.second); }
    g_pool_free.clear();
    g_pool_cached = 0;
}

extern "C" int lqrhip_device_sync(void)
{
    HIPCK(hipDeviceSynchronize());
    return check_dev_error();
}

extern "C" int lqrhip_read_vmap(LqrHipCarver *c, int w0, int h0, int w, int level, int depth, int *out)
{
    int rc = batch_sync_of(c);
    if (rc) return rc;
    int32_t *d = nullptr;
    size_t n = (size_t) w * h0;
    if ((rc = dmalloc(&d, n))) return rc;
    auto run = [&]() -> int {
        hipLaunchKernelGGL(k_compact, dim3(h0), dim3(256), 0, g_stream0, (const uint8_t *) nullptr, c->vs, (const float *) nullptr,
                           (const float *) nullptr, (uint8_t *) nullptr, (float *) nullptr, (float *) nullptr, d, w0, w, c->ch, level,
                           depth);
        HIPCK(hipGetLastError());
        return d2h_staged(out, d, n * sizeof(int32_t));
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);
    dfree(d);
    return rc;
}

extern "C" int lqrhip_read_working(LqrHipCarver *c, int w, int h, float *en, float *m, int *least_dx)
{
    int rc = batch_sync_of(c);
    if (rc) return rc;
    if (!c->pix) return LQRHIP_EARG;
    int32_t fl[FLAG_COUNT];
    HIPCK(hipMemcpy(fl, c->flags, sizeof fl, hipMemcpyDeviceToHost));
    const int org = fl[FLAG_ORG];                 // the carved planes start `org` elements into each row
    size_t n = (size_t) c->stride * h;
    std::vector<float> t(n);
    std::vector<int8_t> tl(n);
    if (en) {
        HIPCK(hipMemcpy(t.data(), c->en, n * sizeof(float), hipMemcpyDeviceToHost));
        for (int y = 0; y < h; y++) memcpy(en + (size_t) y * w, t.data() + (size_t) y * c->stride + org, (size_t) w * sizeof(float));
    }
    if (m) {
        HIPCK(hipMemcpy(t.data(), c->m, n * sizeof(float), hipMemcpyDeviceToHost));
        for (int y = 0; y < h; y++) memcpy(m + (size_t) y * w, t.data() + (size_t) y * c->stride + org, (size_t) w * sizeof(float));
    }
    if (least_dx) {
        HIPCK(hipMemcpy(tl.data(), c->least, n, hipMemcpyDeviceToHost));
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) least_dx[(size_t) y * w + x] = y == 0 ? 0 : (int) tl[(size_t) y * c->stride + org + x];
    }
    return 0;
}

// ---- start over from a device-resident image ---------------------------------------------------
extern "C" int lqrhip_carver_reset(LqrHipCarver *c, const void *device_rgb, int w, int h)
{
    if (!c || c->root || !c->aux.empty() || w < 1 || h < 1) return LQRHIP_EARG;
    int rc = batch_sync_of(c);
    if (rc) return rc;
    const size_t n = (size_t) w * h;
    dfree(c->rgb0); dfree(c->vs); dfree(c->bias0); dfree(c->rig0);
    // a carver that had masks carries bias / rig working planes: the fresh carver has none
    if (c->bias || c->rig) { free_working(c); c->stride = 0; c->wk_h = 0; }
    c->w0 = w; c->h0 = h;
    c->frozen_epoch = 0;
    if ((rc = dmalloc(&c->rgb0, n * c->ch)) || (rc = dmalloc(&c->vs, n))) return rc;
    HIPCK(hipMemcpyAsync(c->rgb0, device_rgb, n * c->ch, hipMemcpyDeviceToDevice, g_stream0));
    HIPCK(hipMemsetAsync(c->vs, 0, n * sizeof(int32_t), g_stream0));
    if (c->batch) c->batch->dirty = true;
    if (c->active && (rc = ensure_working(c, w, h))) return rc;        // synchronises g_stream0 when it allocates
    return 0;
}

// order everything the shim enqueued on its own stream (resets, mask uploads) before the caller goes on
extern "C" int lqrhip_reset_sync(void)
{
    if (g_stream0) HIPCK(hipStreamSynchronize(g_stream0));
    return 0;
}

extern "C" int lqrhip_mem_info(unsigned long long *free_bytes, unsigned long long *total_bytes, unsigned long long *cached_bytes)
{
    if (lqrhip_init() < 0) return LQRHIP_EHIP;
    size_t f = 0, t = 0;
    HIPCK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    if (cached_bytes) *cached_bytes = g_pool_cached;
    return 0;
}

// ---- measured HBM ceiling: streaming copy, ONE 16-byte element per thread, non-temporal -- the form that measured
// fastest on this device (6.5 TB/s read + write at 2 GiB; grid-stride loops with 1-8 loads in flight per thread and
// 1k-64k workgroups: 4.4-5.8 TB/s; scripts/dbg/t_copy.hip)
__global__ __launch_bounds__(256) void k_copy16(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load((const GLOBAL_AS u32x4 *) src + i), (GLOBAL_AS u32x4 *) dst + i);
}

extern "C" int lqrhip_copy_bandwidth(unsigned long long bytes, int iters, double *gbps)
{
    if (lqrhip_init() < 0) return LQRHIP_EHIP;
    if (iters < 1 || bytes < 4096) return LQRHIP_EARG;
    uint8_t *a = nullptr, *b = nullptr;
    int rc;
    if ((rc = dmalloc(&a, bytes)) || (rc = dmalloc(&b, bytes))) { dfree(a); return rc; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    const size_t n16 = bytes / 16;
    auto run = [&]() -> int {
        HIPCK(hipMemsetAsync(a, 1, bytes, g_stream0));
        HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
        const dim3 grid((unsigned) ((n16 + 255) / 256));
        hipLaunchKernelGGL(k_copy16, grid, dim3(256), 0, g_stream0, (const u32x4 *) a, (u32x4 *) b, n16);     // warm-up
        HIPCK(hipEventRecord(e0, g_stream0));
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_copy16, grid, dim3(256), 0, g_stream0, (const u32x4 *) a, (u32x4 *) b, n16);
        HIPCK(hipEventRecord(e1, g_stream0));
        HIPCK(hipStreamSynchronize(g_stream0));
        HIPCK(hipEventElapsedTime(&ms, e0, e1));
        return 0;
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    dfree(a); dfree(b);
    if (rc) return rc;
    if (gbps) *gbps = 2.0 * (double) (n16 * 16) * iters / (ms * 1e-3) / 1e9;
    return 0;
}

// ---- seam-map colour ramp -----------------------------------------------------
// write_vmap_to_layer's per-pixel arithmetic, src/io_functions.c:249-279, in double with every
// operation individually rounded; (guchar)(255 * x) truncates.  One thread per pixel, streaming.
__global__ __launch_bounds__(256) void k_vmap_ramp(const int32_t *__restrict__ vmap, uint32_t *__restrict__ out, size_t n, int depth,
                                                   double sr, double sg, double sb, double er, double eg, double eb)
{
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int vs = vmap[i];
    uint32_t px = 0;                                                     // vs == 0: all four bytes 0 (:253-259)
    if (vs != 0) {
        const double value = __ddiv_rn((double) (depth + 1 - vs), (double) (depth + 1));        // :263
        const double inv = __dsub_rn(1.0, value);
        const double rd = __dadd_rn(__dmul_rn(value, sr), __dmul_rn(inv, er));                    // :264
        const double gr = __dadd_rn(__dmul_rn(value, sg), __dmul_rn(inv, eg));                    // :265
        const double bl = __dadd_rn(__dmul_rn(value, sb), __dmul_rn(inv, eb));                    // :266
        const double al = __dmul_rn(0.5, __dadd_rn(1.0, value));                                  // :267
        // (guchar) of a double: truncation towards zero, then the low 8 bits (values are in [0, 255])
        const uint32_t r8 = (uint32_t) (int) __dmul_rn(255.0, rd) & 0xffu, g8 = (uint32_t) (int) __dmul_rn(255.0, gr) & 0xffu;
        const uint32_t b8 = (uint32_t) (int) __dmul_rn(255.0, bl) & 0xffu, a8 = (uint32_t) (int) __dmul_rn(255.0, al) & 0xffu;
        px = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
    }
    out[i] = px;
}

extern "C" int lqrhip_vmap_to_rgba(const int *vmap, int w, int h, int depth, const double col_start[3], const double col_end[3],
                                   unsigned char *out_rgba)
{
    if (lqrhip_init() < 0) return LQRHIP_EHIP;
    if (!vmap || !out_rgba || w < 1 || h < 1) return LQRHIP_EARG;
    const size_t n = (size_t) w * h;
    int32_t *dv = nullptr;
    uint32_t *dout = nullptr;
    int rc;
    if ((rc = dmalloc(&dv, n)) || (rc = dmalloc(&dout, n))) { dfree(dv); return rc; }
    auto run = [&]() -> int {
        HIPCK(hipMemcpyAsync(dv, vmap, n * sizeof(int32_t), hipMemcpyHostToDevice, g_stream0));
        hipLaunchKernelGGL(k_vmap_ramp, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, g_stream0, dv, dout, n, depth, col_start[0], col_start[1],
                           col_start[2], col_end[0], col_end[1], col_end[2]);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(out_rgba, dout, n * 4, hipMemcpyDeviceToHost, g_stream0));
        HIPCK(hipStreamSynchronize(g_stream0));
        return 0;
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);      // nothing may still use the blocks when they go back to the pool
    dfree(dv); dfree(dout);
    return rc;
}
