// lqr_common.h -- what every translation unit of the gfx950 engine shares: the device descriptors and the
// origin-advanced plane view, the exactly-rounded arithmetic helpers and the energy function, the DP row (dp_row4 / dp_row /
// dp_row_g) that all the halo kernels run, and the geometry constants of the tiled kernels that their launchers need.
//
// Translation units (one per stage, so that a change to one protocol rebuilds -- and re-register-allocates -- only that one):
//   k_energy.hip     E1/E2/E3/E4/E6  k_wk_init, k_mask_add, k_emap_full, k_emap_update, k_frozen_catchup
//   k_backtrack.hip  E7              k_vpath, k_vpath1 (they also pick the side the carve moves)
//   k_carve.hip      E8              k_carve
//   k_band.hip       E5/E9           k_dp_sweep, k_band_update, k_band_update_mw, k_band_update_tw (one workgroup per image)
//   k_tiles.hip      E5/E9           k_dp_tile, k_dp_tile_p (an image spread over several compute units)
//   k_levels.hip     E9              k_band_levels (the band on several compute units, tiles assigned level by level)
//   k_oneoff.hip     E8(vs)/E11/E12/E14, auto-size  k_vs_commit, k_inflate, k_compact(_jobs), k_transpose, k_mask_line_max
//   lqr_shim.hip     the lqrhip_* C ABI of include/lqr_hip.h: allocation cache, batches, the per-seam launch sequence
// lqr_kernels.h declares every kernel for the shim; each kernel file instantiates the templates the shim launches.
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
#pragma once
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
    int8_t *vp_map;         // parallel backtrack (k_vp_*): per chunk of rows, the displacement of every column across the chunk
    int8_t *vp_path;        // ... and after every step inside it ([chunk][step / 4][column][step % 4])
};

// Pointers fetched from a descriptor in memory are "generic" to the compiler, which then
// emits flat_load/flat_store (slower issue, and every access also ties up lgkmcnt, which
// defeats software prefetching).  Kernels therefore work on a view whose members are typed
// as address-space-1 (global) pointers, so that they become global_load/global_store.
// A flag in LDS that one wave of a workgroup writes while the other reads it.  NOT `volatile int` / `*(volatile int *) &x`: inside a
// lambda (and after inlining) a __shared__ variable is reached through a GENERIC pointer, and the compiler leaves volatile accesses
// through generic pointers as FLAT instructions, each followed by s_waitcnt vmcnt(0): every read of such a flag drained the
// wave's outstanding prefetch loads AND waited for the acknowledgement of its write-through hand-over stores (found in round 5
// in k_band_levels' publish path and k_dp_tile_p's hold-back spin).  With the address space spelt out it is ds_read / ds_write.
typedef __attribute__((address_space(3))) volatile int lds_vint;
#ifdef LDS_FLAG_GENERIC            // (the old form, for A/B measurements: make EXTRA=-DLDS_FLAG_GENERIC)
#define LDS_FLAG(x) (*(volatile int *) &(x))
#else
#define LDS_FLAG(x) (*(lds_vint *) &(x))
#endif
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
    gi8 *vp_map, *vp_path;
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
    g.vp_map = (gi8 *) d.vp_map; g.vp_path = (gi8 *) d.vp_path;
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

// A pointer every lane holds, moved to scalar registers ONCE.  The device descriptors are read with vector loads (other kernels
// write them, so the compiler may not use the scalar cache), which leaves the plane pointers in VGPRs: uniform, but every use as
// the scalar base of a load or store then costs v_readfirstlane x 2 and a hazard nop -- inside k_band_update_tw's row loop 7 of
// 71 instructions per row.
template <class T> __device__ __forceinline__ T *uni_ptr(T *p)
{
    const unsigned long long v = (unsigned long long) p;
    const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) v), hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (v >> 32));
    return (T *) (((unsigned long long) hi << 32) | lo);
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

// Device-side failures (a spin wait that timed out because a persistent grid was not co-resident, a
// prediction that did not hold) never trap: the kernel stores a code in this host-mapped word and
// leaves; the host finds it at its next synchronisation and returns LQRHIP_EHIP (-> LQR_ERROR).
#define DEVERR_TILE_TIMEOUT 1
#define DEVERR_BAND_PREDICTION 2
#define DEVERR_SEAMLOG 3            // k_seam_check: the session's seam log does not describe seams
#define DEVERR_LEVELS 4             // k_inflate: a level of the session missing or twice in a row of the base layout
__device__ __forceinline__ void dev_fail(int *flag, int code)
{
    __hip_atomic_store(flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ int dev_failed(int *flag)
{
    return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

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


#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138

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
// nl[i] / nr[i] for dp_row_g: the row above at this lane's first pixel - 1 - i / last pixel + 1 + i, i < DELTA -- pixel i % PX of the lane
// i / PX + 1 lanes away.  The adjacent lane by a DPP wave shift (bound_ctrl: lane 0's left / lane 63's right neighbour read as 0), the
// second one by two (delta_x <= 4, rounds 3 - 4); lanes further away (delta_x 5 .. 10, round 6) by ds_bpermute, whose lane index wraps
// around the wave.  Either way what the outermost lanes receive is wrong by construction: they are halo, the error moves inwards
// DELTA columns per row, and a block is HALO / DELTA rows.
template <int PX, int DELTA>
__device__ __forceinline__ void lane_reach(const float (&mp)[PX], float (&nl)[DELTA], float (&nr)[DELTA])
{
    const int lane4 = (int) (threadIdx.x & 63) << 2;
#pragma unroll
    for (int i = 0; i < DELTA; i++) {
        const int away = i / PX + 1;
        int a, b2;
        if (away <= 2) {
            a = __builtin_amdgcn_mov_dpp(__float_as_int(mp[PX - 1 - i % PX]), DPP_WAVE_SHR1, 0xf, 0xf, true);
            b2 = __builtin_amdgcn_mov_dpp(__float_as_int(mp[i % PX]), DPP_WAVE_SHL1, 0xf, 0xf, true);
            if (away == 2) { a = __builtin_amdgcn_mov_dpp(a, DPP_WAVE_SHR1, 0xf, 0xf, true); b2 = __builtin_amdgcn_mov_dpp(b2, DPP_WAVE_SHL1, 0xf, 0xf, true); }
        } else {
            a = __builtin_amdgcn_ds_bpermute(lane4 - 4 * away, __float_as_int(mp[PX - 1 - i % PX]));
            b2 = __builtin_amdgcn_ds_bpermute(lane4 + 4 * away, __float_as_int(mp[i % PX]));
        }
        nl[i] = __int_as_float(a);
        nr[i] = __int_as_float(b2);
    }
}
// PX floats / PX back-pointer bytes of one lane, as one load or store
template <int PX> struct LaneVec;
template <> struct LaneVec<2> { typedef float F __attribute__((ext_vector_type(2))); typedef uint16_t L; };
template <> struct LaneVec<4> { typedef f32x4 F; typedef uint32_t L; };

// ---- geometry of the persistent tiled kernels (k_tiles.hip), needed by their launchers too
// PX pixels per lane (4, or 2 when the device has room for twice the tiles: half the instructions per wave and row):
// a tile is 64 * PX columns of which the 16 outer lanes on each side are halo
// `px` below is a geometry code: 2 / 4 = pixels per lane with 16 halo lanes per side; 3 (round 6) = 2 pixels per lane with 24 halo lanes per
// side -- 32 own columns + 48-column halos, blocks of 48 rows: the hand-over through memory (a third of a 32-row level) is paid 45
// times per 4K sweep instead of 68, for twice the tiles; used while every tile still has a compute unit to itself (single images)
constexpr int dppx_px(int px) { return px == 3 ? 2 : px; }
constexpr int dppx_hl(int px) { return px == 3 ? 24 : 16; }
constexpr int dpp_halo(int px) { return dppx_hl(px) * dppx_px(px); }              // halo columns on each side = rows per block
constexpr int dpp_own(int px) { return 64 * dppx_px(px) - 2 * dpp_halo(px); }     // columns a tile owns
constexpr int dpp_ex_tile(int px) { return px == 3 ? 2 * dpp_own(3) : 2 * 2 * dpp_halo(px); }       // granules a tile publishes: [block parity][side: 0 to the left, 1 to the right][column]; px 3: [block parity][own column]
constexpr int dpp_rb(int px, int delta) { return delta >= 5 ? dpp_halo(2) / delta : delta >= 3 ? 8 : dpp_halo(px) / delta; }      // rows per block (delta_x 5 .. 10: 6, 5, 4, 4, 3, 3)
constexpr int DPP_R = 16;                       // rows per batch
constexpr int DPP_W = 2;                        // waves taking turns
static_assert(dpp_halo(2) % (2 * DPP_R) == 0 && dpp_halo(4) % (2 * DPP_R) == 0, "a block (halo / delta_x rows, delta_x <= 2) is a whole number of batches");
constexpr int DPP_BLK_BITS = 12;                // bits of the block index in a granule's tag
#define DPT_ROWS 32
#define DPT_OWN 192
// k_band_levels (k_levels.hip): slots per image at most, tiles per image at most (one 64-bit mask: rows up to 4096 px)
constexpr int LV_PMAX = 16;
constexpr int LV_MAX_TILES = 64;
constexpr int LV_MAX_LEVELS = 1020;       // levels per image at most (10 bits of the tags hold level + 1; 4K rows at delta_x 10: 720 levels of 3 rows)
constexpr int lv_rows(int delta, bool rigm = false) { return delta == 1 ? (rigm ? 16 : 32) : delta == 2 ? 16 : delta <= 4 ? 8 : 32 / delta; }      // rows per level: halo (32 columns) / delta_x (a rigidity mask: 16, for the registers)
constexpr int LQR_FAST_MAX_DELTA = 10;    // delta_x up to which the tiled kernels have instantiations (the plug-in's UI: src/interface.c:47, MAX_DELTA_X 10)

// a job of the one-launch plane passes (inflate, flatten, transpose): one carver (root or attached) of a batch
struct InflateDev {
    const uint8_t *rgb;
    const int32_t *vs;
    const float *bias, *rig;
    uint8_t *nrgb;
    int32_t *nvs;
    float *nbias, *nrig;
    int ch;
};
#define EU_ROWS 62          // k_emap_update: rows per block (+2 halo rows)
#ifndef EU_LOGB
#define EU_LOGB 8           // k_emap_update / k_carve_e: log entries fetched per round of the walk back to the frozen frame
#endif
// parallel backtrack (k_backtrack.hip, k_vp_*): a chunk is VP_REACH / delta_x rows, so that a path moves at most VP_REACH columns
// inside a chunk (the displacement fits a byte); k_vp_solve walks VP_STAGE chunks per LDS-resident stage
constexpr int VP_REACH = 56;
constexpr int VP_STAGE = 20;        // (4K: 39 chunks = 2 stages; the cone of a stage is 2 * 56 * 20 columns wide: 45 KB of LDS)
constexpr int vp_chunk_rows(int delta) { return VP_REACH / delta; }
#define VP_ROWS 62
