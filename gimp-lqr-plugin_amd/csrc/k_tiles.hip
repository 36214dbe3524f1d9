// k_tiles.hip -- E5 / E9 with an image spread over several compute units: k_dp_tile (a launch per 32 rows), k_dp_tile_p (persistent, halo hand-over through tagged granules), k_band_tiles (the band update on a tile set that grows on demand)
// (gfx950 / CDNA4, wave64; see lqr_common.h for the file map and DESIGN.md section 4 for the measurements)
#include "lqr_common.h"
#include "lqr_kernels.h"

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
// co-residency bound for the spin waits, set from the occupancy query in lqrhip_init (dpp_resident_workgroups)


// DELTA = delta_x (1 or 2: errors move DELTA columns per row, so a block is HALO / DELTA rows); RIGM = a rigidity mask
// scales the rigidity term per pixel (one more 4-byte plane read).  The plain instantiations (1, false) use the
// 3-neighbour row above, the others dp_row_g.
#ifdef LQR_TIMING
__device__ unsigned long long g_tile_dbg[2 * 16];
#define TT(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); tdbg[i] += t__ - ttprev; ttprev = t__; } while (0)
#else
#define TT(i) do { } while (0)
#endif
__device__ int g_dpp_dbg;          // experiment switches (lqrhip_dp_tile_debug): 1 no near copies, 2 tile = workgroup index (neighbours on different XCDs) and no near copies
template <int PX, bool LR, bool RIG, bool UPDATE, int DELTA, bool RIGM>
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
    __shared__ int s_polled;                     // last block whose halo wave 0 has received (LDS_FLAG: lqr_common.h)
    if (threadIdx.x == 0) { s_fail = 0; s_polled = 0; }
    __syncthreads();
    const GCarver c = gview(cs[blockIdx.y]);
    gf32 *m_out = UPDATE ? c.m2 : c.m;
    gi8 *least_out = UPDATE ? c.least2 : c.least;
    // Workgroup b is observed to run on XCD b mod 8 (no promise: speed only).  The tiles are numbered so that the workgroups of one
    // XCD hold CONSECUTIVE tiles (class b & 7 holds tiles [cls a + min(cls, r), ...), ntiles = 8 a + r): all but seven of the
    // neighbour pairs then sit on one XCD, whose L2 serves their hand-over through the near copies below.  (debug switch 2: tile = b)
    const int dbg = __builtin_amdgcn_readfirstlane(g_dpp_dbg);
    const int ntiles = gridDim.x;
    const int cls = (int) blockIdx.x & 7, cls_k = (int) blockIdx.x >> 3, cls_n = (ntiles >> 3) + (cls < (ntiles & 7) ? 1 : 0);
    const int tile = (dbg & 2) ? (int) blockIdx.x : cls * (ntiles >> 3) + min(cls, ntiles & 7) + cls_k;
    // exchange area of this image: per tile EX_TILE granules ({m bits, tag}, 8 bytes, one store each), then one
    // word that counts finished tiles; then the same again: the NEAR copies of the granules (plain stores: they stay in the writer's
    // L2, where a neighbour on the same XCD finds them without the trip to memory and back -- see k_levels.hip).  A lane whose
    // neighbour is of its own class polls the near copy three times out of four, the write-through copy the fourth; the others
    // poll the write-through copy only.  The protocol rests on the write-through copies alone.
    typedef GLOBAL_AS unsigned long long gu64;
    const size_t near_off = (size_t) ntiles * EX_TILE + 8;
    gu64 *ex_img = (gu64 *) exch + (size_t) blockIdx.y * 2 * near_off;
    const bool near_l = !(dbg & 3) && cls_k > 0, near_r = !(dbg & 3) && cls_k + 1 < cls_n;      // the left / right neighbour is on this XCD
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
#ifdef LQR_TIMING
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
                    const bool lane_near = need && (lane < 32 ? near_l : near_r);
                    while (true) {
                        gu64 *s2 = src + ((lane_near && (spins & 3) != 3) ? near_off : 0);
#pragma unroll
                        for (int k = 0; k < PX; k++) g[k] = __hip_atomic_load(s2 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        bool ok = true;
#pragma unroll
                        for (int k = 0; k < PX; k++) ok &= ((unsigned) (g[k] >> 32) == want);
                        if (__all(ok || !need)) break;
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                        if ((spins & 1023) == 0 && dev_failed(dev_err)) { failed = true; break; }
                        if (spins > (1 << 22)) { if (lane == 0) dev_fail(dev_err, DEVERR_TILE_TIMEOUT); failed = true; break; }
                    }
                    if (failed) LDS_FLAG(s_fail) = 1;
                    if (need) {
#pragma unroll
                        for (int k = 0; k < PX; k++) mp[k] = in[k] ? __uint_as_float((unsigned) g[k]) : INF;
                    }
                    if (lane == 0) LDS_FLAG(s_polled) = j;          // the partner's prefetch may start (see the issue site)
                }
                TT(1);
#ifdef LQR_TIMING
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
                        for (int k = 0; k < PX; k++) {
                            if (side ? near_r : near_l) __hip_atomic_store(dst + near_off + k, tag | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store(dst + k, tag | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
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
                    while (LDS_FLAG(s_polled) < j + 1 && !LDS_FLAG(s_fail) && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
                }
                TT(6);
                if constexpr (UPDATE) store_u(yb);        // the batch this wave has just computed, before its registers are reloaded
                TT(7);
                issue(yb + DPP_W * R);
                TT(8);
            }
        }
    }
#ifdef LQR_TIMING
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
// [0] images not covered by their tile set, [1] images aborted at an edge, [2] reserve tiles woken, [3] requests that found
// no reserve left (rare events: one atomic each)
__device__ unsigned long long g_bt_stats[8];
#ifdef LQR_TIMING
// per tile slot of image 0 and wave: cycles in [0] receive, [1] compute, [2] rest before the barrier, [3] barrier, [4] wait for the partner's poll,
// [5] stores, [6] prefetch issue, [7] active blocks, [8] whole kernel
__device__ unsigned long long g_bt_time[16][2][10];
#define BTT(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); btt[i] += t__ - btprev; btprev = t__; } while (0)
#else
#define BTT(i) do { } while (0)
#endif
#define BT_STAT(i) do { if (lane == 0) atomicAdd(&g_bt_stats[i], 1ull); } while (0)

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
    __shared__ int s_polled;                      // last block whose hand-over this workgroup has received (flags: LDS_FLAG, lqr_common.h)
    __shared__ int s_own_chg;                     // an own pixel changed on the last row of the block just finished
    __shared__ int s_nbr_live;                    // the hand-over last received says a neighbour was active or changed at its edge
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

#ifdef LQR_TIMING
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
                if (rcv & 4) LDS_FLAG(s_fail) = 1;
                if (lane == 0) { LDS_FLAG(s_nbr_live) = (rcv & 9) != 0; LDS_FLAG(s_polled) = j; }
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
#ifdef LQR_TIMING
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
                const int fl = LDS_FLAG(s_from[0]), fr = LDS_FLAG(s_from[1]);
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
                while (LDS_FLAG(s_polled) < j + 1 && !LDS_FLAG(s_fail) && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
                if (LDS_FLAG(s_fail)) continue;          // the partner saw an abort or a time-out: it is at the barrier
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
                staged = act || nbr_act || LDS_FLAG(s_nbr_live) != 0 || (fnext & 8u);      // an active neighbour's band may arrive within two blocks
                if (staged) issue_full(j2 * R); else issue_last(j2 * R);
            }
            BTT(6);
        }
    }
#ifdef LQR_TIMING
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
#ifdef LQR_TIMING
extern "C" int lqrhip_band_tiles_timing(unsigned long long *out) { (void) hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bt_time), sizeof(unsigned long long) * 320) == hipSuccess ? 0 : -1; }
#endif
extern "C" void lqrhip_dp_tile_debug(int v) { (void) hipMemcpyToSymbol(HIP_SYMBOL(g_dpp_dbg), &v, sizeof v); }
extern "C" int lqrhip_band_tiles_stats(unsigned long long *out, int reset)
{
    (void) hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bt_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_bt_stats), z, sizeof z); }
    return 0;
}
#ifdef LQR_TIMING
extern "C" int lqrhip_tile_timing(unsigned long long *out) { (void) hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_dbg), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1; }
#endif

// ---- the instantiations the shim launches (lqr_kernels.h declares them)
#define INST_TILE(LRV, RIGV) template __global__ void k_dp_tile<LRV, RIGV>(const DevCarver *, DpK, int, int, int, int); \
    template __global__ void k_band_tiles<LRV, RIGV>(DevCarver *, DpK, int, int, int, unsigned long long *, int, int *, int, int);
INST_TILE(false, false) INST_TILE(false, true) INST_TILE(true, false) INST_TILE(true, true)
#define INST_P(...) template __global__ void k_dp_tile_p<__VA_ARGS__>(DevCarver *, DpK, int, int, int, unsigned long long *, int, int *);
#define INST_P_LR(LRV, UPD) INST_P(4, LRV, false, UPD) INST_P(4, LRV, true, UPD) INST_P(2, LRV, false, UPD) INST_P(2, LRV, true, UPD) \
    INST_P(2, LRV, true, UPD, 1, true) \
    INST_P(2, LRV, false, UPD, 2, false) INST_P(2, LRV, true, UPD, 2, false) INST_P(2, LRV, true, UPD, 2, true) \
    INST_P(2, LRV, false, UPD, 3, false) INST_P(2, LRV, true, UPD, 3, false) INST_P(2, LRV, true, UPD, 3, true) \
    INST_P(2, LRV, false, UPD, 4, false) INST_P(2, LRV, true, UPD, 4, false) INST_P(2, LRV, true, UPD, 4, true)
INST_P_LR(false, false) INST_P_LR(false, true) INST_P_LR(true, false) INST_P_LR(true, true)
