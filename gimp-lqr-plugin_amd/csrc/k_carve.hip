// k_carve.hip -- E8 carve: the seam leaves every carved plane, in place; the HBM-bound kernel
// (gfx950 / CDNA4, wave64; see lqr_common.h for the file map and DESIGN.md section 4 for the measurements)
#include "lqr_common.h"
#include "lqr_kernels.h"

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
            __builtin_nontemporal_store(o, (GLOBAL_AS u32x4 *) (row + x));
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
            __builtin_nontemporal_store(o, (GLOBAL_AS u32x4 *) (row + x));
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
            __builtin_nontemporal_store(o, row32 + (x >> 2));
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
            __builtin_nontemporal_store(o, row32 + (x >> 2));
        }
    }
}

// one row of the carve, by one wave (the body of k_carve's row loop).  c = the
// PHYSICAL view, org / side = what k_vpath* published for this seam, w = the width before the carve.
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
            st_left_u32(en, base, pv, pend, lane, E);
            if (move_dp) { st_left_u32(mm, base, pv, pend, lane, M); st_left_8(l32, base, org, start, v, vprev, y, wnew, lane, L); }
        }
        if (rg)
            for (int base = pv & ~3; base < pend; base += CGPX) { G32 R; ld_left_u32(rg, base, pend, lane, R); st_left_u32(rg, base, pv, pend, lane, R); }
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
            st_right_u32(en, gbase, org, pv, lane, E);
            if (move_dp) { st_right_u32(mm, gbase, org, pv, lane, M); st_right_8(l32, gbase, org, max(end_l, 0), v, vprev, y, lane, L); }
        }
        if (rg)
            for (int top = top_u; top > org + 1; top -= CGPX) { G32 R; ld_right_u32(rg, top - CGPX, org, pv, lane, R); st_right_u32(rg, top - CGPX, org, pv, lane, R); }
    }
}

__global__ __launch_bounds__(256) void k_carve(const DevCarver *cs, int w, int h, int stride, int delta, int move_dp)
{
    // blockIdx.x = row block (fastest): consecutive workgroups take consecutive rows of one image (2.5 % faster than
    // image-fastest, which round 1 used so that a concurrent band update could follow all images' top rows)
    const GCarver c = gview_phys(cs[blockIdx.y]);
    const int org = c.flags[FLAG_ORG_PREV], side = c.flags[FLAG_SIDE];      // published by k_vpath* for this seam
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) c.flags[FLAG_OVF_ROW] = h;       // the band kernels lower / overwrite it when they hand rows over to k_dp_sweep<UPDATE>
    for (int y = blockIdx.x * 4 + (threadIdx.x >> 6); y < h; y += gridDim.x * 4) carve_row(c, org, side, y, w, stride, delta, move_dp, lane);
}

// ---------------------------------------------------------------------------
// k_carve_e<NRG>: k_carve and k_emap_update<NRG, 12> in ONE launch, for single images and small groups (round 6).  There a seam round is
// a chain of dependent launches of small kernels (a 4K carve is 9 us, the energy update 9 us, each plus its dispatch), and the
// energy of row y needs nothing but row y's own carve: the wave that has moved a row refreshes the energies next to the seam on
// that row.  What k_emap_update does with a thread per row and LDS rows shared between neighbouring threads, a wave does alone:
// lanes 0 .. 35 stage the 12 brightness samples of rows y - 1, y, y + 1 (each mapped back to the frozen pixel plane through the
// seam log, as there), lanes 0 .. 11 then evaluate the gradient energy of the row's changed interval.  Same arithmetic, same
// order of operations: bit-identical energies.  delta_x <= 2 (12 samples per row suffice); larger delta_x and larger groups keep
// the two kernels (the carve's 96 registers and 5 waves per SIMD are what its bandwidth rests on).
// ---------------------------------------------------------------------------
template <int NRG>
__global__ __launch_bounds__(256) void k_carve_e(const DevCarver *cs, DpK p, int w, int h, int stride, int move_dp, int k, int epoch)
{
    constexpr int NT = 12;
    constexpr bool luma = (NRG >= 3);
    const GCarver c = gview_phys(cs[blockIdx.y]);
    const int org = c.flags[FLAG_ORG_PREV], side = c.flags[FLAG_SIDE];      // published by the backtrack for this seam
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ double s_n255[256];
    __shared__ double s_bt[4][3][NT];
    __shared__ float s_bb[4][3][NT];
    __shared__ int s_lo[4][3];
    fill_norm255(s_n255, threadIdx.x, 256);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) c.flags[FLAG_OVF_ROW] = h;
    const int wn = w - 1;                                        // the frame after the carve
    gf32 *en = c.en + org + side;                                // its logical view (the origin after this seam)
    for (int y = blockIdx.x * 4 + wv; y < h; y += gridDim.x * 4) {
        carve_row(c, org, side, y, w, stride, p.delta, move_dp, lane);
        if (wn <= 1) continue;
        // ---- the energy of row y next to the seam (k_emap_update, one row)
        const int g = lane / NT, i = lane - g * NT, t = y - 1 + g;          // lanes 0 .. 35: sample i of row t
        const bool samp = lane < 3 * NT && t >= 0 && t < h;
        int lo = 0, r = -1, pos = 0;
        if (samp) {
            int xmin, xmax;
            nrg_interval(c.seam_x, t, h, wn, p.radius, xmin, xmax);
            int l = xmin - 1; r = xmax + 1;
            if (t > 0) { int a, b; nrg_interval(c.seam_x, t - 1, h, wn, p.radius, a, b); if (b >= a) { l = min(l, a); r = max(r, b); } }
            if (t < h - 1) { int a, b; nrg_interval(c.seam_x, t + 1, h, wn, p.radius, a, b); if (b >= a) { l = min(l, a); r = max(r, b); } }
            lo = max(l, 0);
            pos = lo + i;
            const gi32 *lg = c.seam_log + t;
            for (int j = k; j >= epoch; j -= EU_LOGB) {
                int v[EU_LOGB];
#pragma unroll
                for (int u = 0; u < EU_LOGB; u++) v[u] = lg[(size_t) max(j - u, epoch) * h];
#pragma unroll
                for (int u = 0; u < EU_LOGB; u++) { const int vu = (j - u >= epoch) ? v[u] : 0x7fffffff; pos += (vu <= pos) ? 1 : 0; }
            }
            const int wf = wn + (k - epoch) + 1;                 // width of the frozen frame
            const bool ok = (lo + i <= min(r, wn - 1)) && pos < wf;
            const size_t o = (size_t) t * stride + (ok ? pos : 0);
            s_bt[wv][g][i] = ok ? px_bright(c.pix[o], p.ch, luma, Norm255Lut{s_n255}) : 0.0;
            s_bb[wv][g][i] = (ok && c.bias) ? c.bias[o] : 0.0f;
            if (i == 0) s_lo[wv][g] = lo;
        }
        __builtin_amdgcn_wave_barrier();                          // (one wave: its LDS operations execute in order)
        int xmin, xmax;
        nrg_interval(c.seam_x, y, h, wn, p.radius, xmin, xmax);
        const int x = xmin + lane;
        if (x <= xmax) {
            float e = grad_energy_f<NRG>([&](int xx, int yy) { const int gg = yy - y + 1; return s_bt[wv][gg][xx - s_lo[wv][gg]]; }, x, y, wn, h);
            if (c.bias) e = __fadd_rn(e, __fdiv_rn(s_bb[wv][1][x - s_lo[wv][1]], (float) p.w_start));
            // (behind the carve's stores of this row in program order; different cache policies: drain them first)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            en[(size_t) y * stride + x] = e;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the instantiations the shim launches (lqr_kernels.h declares them)
// (k_carve is not a template)
#define INST_CE(N) template __global__ void k_carve_e<N>(const DevCarver *, DpK, int, int, int, int, int, int);
INST_CE(0) INST_CE(1) INST_CE(2) INST_CE(3) INST_CE(4) INST_CE(5) INST_CE(6)
