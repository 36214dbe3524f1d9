// k_band.hip -- E5 / E9 with one workgroup per image: the full-width row sweep (k_dp_sweep), the generic band walk (k_band_update), the multi-wave band (k_band_update_mw) and the trapezoid-wave band (k_band_update_tw)
// (gfx950 / CDNA4, wave64; see lqr_common.h for the file map and DESIGN.md section 4 for the measurements)
#include "lqr_common.h"
#include "lqr_kernels.h"

// ---------------------------------------------------------------------------
// E5 / E9 (full width): cumulative-min DP row sweep, one persistent workgroup
// per image.  The previous row of m lives in LDS (ping-pong), so the only HBM
// traffic is en in, m + back-pointer out (9 B/px), all coalesced.  One
// s_barrier per row.  UPDATE applies liblqr's update_mmap keep-rule to every
// pixel of rows >= flags[FLAG_OVF_ROW]; applied to a superset of liblqr's band
// it leaves identical memory contents (pixels outside the band have unchanged
// inputs, float ops are deterministic).
// ---------------------------------------------------------------------------
// NTH threads: 1024 for the full sweeps; 256 (round 6) for the launch behind the band kernels, which almost always only LOOKS at
// flags[FLAG_OVF_ROW] and leaves -- a 1024-thread workgroup needs 16 waves' worth of one compute unit free at once, and beside the
// sibling streams' kernels that wait was 43 us per seam round at 64 images
template <int PXT, bool UPDATE, int NTH>
__global__ __launch_bounds__(NTH) void k_dp_sweep(const DevCarver *cs, DpK p, int w, int h, int stride, int lr)
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
        for (int x = tid; x < w; x += NTH) {
            float e = c.en[x];
            c.m[x] = e;
            prev[x] = e;
        }
        y0 = 1;
    } else {
        for (int x = tid; x < w; x += NTH) prev[x] = c.m[(size_t) (y0 - 1) * stride + x];
    }
    __syncthreads();

    float e_nx[PXT], mo_nx[PXT], rf_nx[PXT];
    int lo_nx[PXT];
    auto prefetch = [&](int y) {
#pragma unroll
        for (int k = 0; k < PXT; k++) {
            int x = tid + k * NTH;
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
            int x = tid + k * NTH;
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

#ifdef LQR_TIMING
// per wave of image 0's workgroup: cycles [0] waiting for the prefetched batch (landed), [1] the batch (rows, hand-over, or idling), [2] barrier,
// [3] between barrier and issue, [4] issue, [5] whole kernel
__device__ unsigned long long g_tw_time[8][8];
#define TWT(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); twt[i] += t__ - twprev; twprev = t__; } while (0)
extern "C" int lqrhip_band_tw_timing(unsigned long long *out) { (void) hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tw_time), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1; }
#else
#define TWT(i) do { } while (0)
#endif
template <int NW, bool LR, bool RIG>
__device__ __forceinline__ void band_update_tw_body(const DevCarver &dc, const DpK &p, int w, int h, int stride, int *dev_err)
{
    constexpr int R = TW_R, OWN = TW_OWN, WIN = NW * OWN, NT = 128 * NW;
    GCarver c = gview(dc);
    c.en = uni_ptr(c.en); c.m = uni_ptr(c.m); c.least = uni_ptr(c.least);       // (scalar bases for the row loop: lqr_common.h)
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

#ifdef LQR_TIMING
    unsigned long long twt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long twprev = __builtin_readcyclecounter();
    const unsigned long long twstart = twprev;
#endif
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
            TWT(0);
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
            TWT(0);
        }
        // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. the prefetch
        TWT(1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        TWT(2);
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
            TWT(3);
            issue(y_issue, lane_staged);
            TWT(4);
            loads_full = issue_full;
        }
        if (just_rebased) {
            landed();                        // before anybody stores rows >= y
            __syncthreads();
        }
    }
    if (tid == 0) c.flags[FLAG_OVF_ROW] = ovf;
#ifdef LQR_TIMING
    if (blockIdx.x == 0 && lane == 0) { twt[5] = __builtin_readcyclecounter() - twstart; for (int i = 0; i < 8; i++) g_tw_time[wv][i] = twt[i]; }
#endif
}

template <int NW, bool LR, bool RIG>
__global__ __launch_bounds__(128 * NW) void k_band_update_tw(const DevCarver *cs, DpK p, int w, int h, int stride, int *dev_err)
{
    // Claim the whole register file of the SIMDs this workgroup sits on (2 waves x 256 VGPRs): no wave of a sibling stream's kernel is
    // placed on this CU while the band is walked.  Measured at 64 x 4K on one box, alternating builds: 500.0 / 501.1 k against 489.8 /
    // 493.6 k Mseams*px/s (+1.8 %) -- NOT through this kernel's own time (630 us either way: what stretches it from 466 us alone is the
    // loaded memory system, not its CU's neighbours) but through the kernels that no longer land beside it (k_emap_update 50 -> 32 us,
    // k_dp_tile 2.1 -> 1.5 ms per sweep).
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    band_update_tw_body<NW, LR, RIG>(cs[blockIdx.x], p, w, h, stride, dev_err);
}


// ---- the instantiations the shim launches (lqr_kernels.h declares them)
#define INST_SWEEP(P) template __global__ void k_dp_sweep<P, false, DP_THREADS>(const DevCarver *, DpK, int, int, int, int); template __global__ void k_dp_sweep<P, true, DP_THREADS>(const DevCarver *, DpK, int, int, int, int); \
    template __global__ void k_dp_sweep<P, true, 256>(const DevCarver *, DpK, int, int, int, int);
INST_SWEEP(1) INST_SWEEP(2) INST_SWEEP(4) INST_SWEEP(8) INST_SWEEP(16)
#define INST_BAND(LRV, RIGV) template __global__ void k_band_update_tw<4, LRV, RIGV>(const DevCarver *, DpK, int, int, int, int *); \
    template __global__ void k_band_update_mw<2, 8, 8, LRV, RIGV>(const DevCarver *, DpK, int, int, int); \
    template __global__ void k_band_update_mw<2, 16, 8, LRV, RIGV>(const DevCarver *, DpK, int, int, int);
INST_BAND(false, false) INST_BAND(false, true) INST_BAND(true, false) INST_BAND(true, true)
