// k_tiles.hip -- E5 / E9 with an image spread over several compute units: k_dp_tile (a launch per 32 rows), k_dp_tile_p (persistent, halo hand-over through tagged granules)
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
__device__ int g_dpp_dbg;          // experiment switches (lqrhip_dp_tile_debug): 1 no near copies, 2 tile = workgroup index (neighbours on different XCDs) and no near copies,
                                   // 4 timing jitter: pseudo-random sleeps (0 .. ~15 x 8 k cycles) at the protocol's hand-over sites -- a race that needs an unusual
                                   // interleaving of tiles and waves gets thousands of them per launch (scripts/jitter_soak.py; round 6's hunt for the one-offs of rounds 4 / 5)
// A build of its own (make EXTRA=-DLQR_JITTER): even a uniform scalar branch per site moved the rigidity-mask instantiation from 256 to
// 294 registers (tests/test_kernel_budgets.py); the product kernels carry no site.
#ifdef LQR_JITTER
#define JIT(site) do { if (dbg & 4) { unsigned h__ = (unsigned) (tile * 131 + j * 17 + (site) * 7 + epoch * 2654435761u + q * 40503u); h__ ^= h__ >> 13; h__ *= 0x5bd1e995u; h__ ^= h__ >> 15; \
    for (unsigned i__ = h__ & 15u; i__ > 0; i__--) __builtin_amdgcn_s_sleep(127); } } while (0)
#else
#define JIT(site) do { } while (0)
#endif
template <int PX, bool LR, bool RIG, bool UPDATE, int DELTA, bool RIGM, int HLN>
__global__ __launch_bounds__(64 * DPP_W) void k_dp_tile_p(DevCarver *cs, DpK p, int w, int h, int stride, unsigned long long *exch, int epoch, int *dev_err)
{
    static_assert(DELTA >= 1 && DELTA <= LQR_FAST_MAX_DELTA && (DELTA <= 4 || (PX == 2 && RIG)) && (RIG || !RIGM), "delta_x 1 .. 10 (5 .. 10: the rigidity form, with a zero table if there is none); a rigidity mask only matters with rigidity");
    typedef typename LaneVec<PX>::F FV;
    typedef typename LaneVec<PX>::L LV;
    typedef GLOBAL_AS FV GFV;
    typedef GLOBAL_AS LV GLV;
    static_assert(HLN == 16 || (HLN == 24 && PX == 2 && DELTA == 1 && !RIGM), "24 halo lanes per side: the plain 2-px instantiations only");
    constexpr int PXC = HLN == 24 ? 3 : PX;         // the geometry code of lqr_common.h
    constexpr int HALO = dpp_halo(PXC), OWN = dpp_own(PXC), EX_TILE = dpp_ex_tile(PXC), TILE = 64 * PX, HL = HLN;      // HL: halo lanes per side
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
    constexpr int R = HLN == 24 ? 24 : (PX == 2 && DELTA == 1 && (!RIGM || DPP_RIGM32)) ? DPP_R2 : DELTA >= 5 ? dpp_rb(PX, DELTA) : DELTA >= 3 ? 8 : DPP_R;      // delta_x 3, 4: 8-row blocks (errors move up to 4 columns per row); 5 .. 10: 6 .. 3 rows
    FV q_e[R], q_mo[R], q_rf[RIGM ? R : 1];
    LV q_lo[R];
    constexpr int RB = dpp_rb(PXC, DELTA), NBB = RB / R;         // rows, batches per block (24 halo lanes: 48-row blocks in two 24-row batches, the waves alternating)
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
                        lane_reach<PX, DELTA>(mp, nl, nr);       // the neighbouring lanes' pixels next to this lane's: DELTA on each side
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
                lane_reach<PX, DELTA>(mp, nl, nr);
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
                    int nb, col;
                    bool src_near;
                    if constexpr (HLN == 16) {
                        nb = (lane < 32) ? tile - 1 : tile + 1;                     // left halo <- left neighbour's right-going granules
                        col = !need ? 0 : (lane < 32) ? PX * lane : PX * (lane - 64 + HL);        // lanes that need nothing poll a dummy
                        col += (((j - 1) & 1) * 2 + (lane < 32 ? 1 : 0)) * HALO;
                        src_near = lane < 32 ? near_l : near_r;
                    } else {
                        // 48-column halos over 32-column tiles: a halo reaches across the adjacent tile into the one behind it; every tile
                        // publishes ALL its own columns once ([parity][column]), a halo lane reads the tile that owns its columns
                        nb = need ? x0 / OWN : tile;                                 // (x0 >= 0 for a lane with a pixel inside the image)
                        col = (need ? x0 - nb * OWN : 0) + ((j - 1) & 1) * OWN;
                        src_near = !(dbg & 3) && nb >= tile - cls_k && nb < tile - cls_k + cls_n;          // the owner sits on this XCD
                    }
                    gu64 *src = ex_img + (size_t) (need ? nb : tile) * EX_TILE + col;
                    const unsigned want = ((unsigned) epoch << DPP_BLK_BITS) | (unsigned) j;
                    unsigned long long g[PX];
                    int spins = 0;
                    bool failed = false;
                    JIT(1);
                    const bool lane_near = need && src_near;
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
                    JIT(2);
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
                    JIT(3);
                    if (own_lane) {
                        const unsigned long long tag = (unsigned long long) (((unsigned) epoch << DPP_BLK_BITS) | (unsigned) (j + 1)) << 32;
                        if constexpr (HLN == 16) {
                            const int side = lane < 32 ? 0 : 1;
                            gu64 *dst = ex_img + (size_t) tile * EX_TILE + (size_t) ((j & 1) * 2 + side) * HALO + PX * (lane - (side ? 32 : HL));
#pragma unroll
                            for (int k = 0; k < PX; k++) {
                                if (side ? near_r : near_l) __hip_atomic_store(dst + near_off + k, tag | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_store(dst + k, tag | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        } else {
                            gu64 *dst = ex_img + (size_t) tile * EX_TILE + (size_t) (j & 1) * OWN + PX * (lane - HL);
#pragma unroll
                            for (int k = 0; k < PX; k++) {
                                if (!(dbg & 3) && cls_n > 1) __hip_atomic_store(dst + near_off + k, tag | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_store(dst + k, tag | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
            }
            TT(4);
            // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. every wave's prefetch
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TT(5);
            if (s_fail) return;                  // uniform: written before the barrier, read by both waves after it
            JIT(4 + (mine ? 1 : 0));
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
                JIT(6);
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
    { const int j = nblk; (void) j; JIT(7); }
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

// lqrhip_dp_tile_debug: experiment switches of k_dp_tile_p (g_dpp_dbg above); ADVICE r5: on the device the library selected
extern "C" void lqrhip_dp_tile_debug(int v) { if (lqrhip_init() < 0) return; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_dpp_dbg), &v, sizeof v); }
#ifdef LQR_TIMING
extern "C" int lqrhip_tile_timing(unsigned long long *out) { (void) hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_dbg), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1; }
#endif

// ---- the instantiations the shim launches (lqr_kernels.h declares them)
#define INST_TILE(LRV, RIGV) template __global__ void k_dp_tile<LRV, RIGV>(const DevCarver *, DpK, int, int, int, int);
INST_TILE(false, false) INST_TILE(false, true) INST_TILE(true, false) INST_TILE(true, true)
#define INST_P(...) template __global__ void k_dp_tile_p<__VA_ARGS__>(DevCarver *, DpK, int, int, int, unsigned long long *, int, int *);
#define INST_P_WIDE(LRV, RIGV, UPD) INST_P(2, LRV, RIGV, UPD, 1, false, 24)
#define INST_P_LR(LRV, UPD) INST_P(4, LRV, false, UPD) INST_P(4, LRV, true, UPD) INST_P(2, LRV, false, UPD) INST_P(2, LRV, true, UPD) \
    INST_P(2, LRV, true, UPD, 1, true) \
    INST_P(2, LRV, false, UPD, 2, false) INST_P(2, LRV, true, UPD, 2, false) INST_P(2, LRV, true, UPD, 2, true) \
    INST_P(2, LRV, false, UPD, 3, false) INST_P(2, LRV, true, UPD, 3, false) INST_P(2, LRV, true, UPD, 3, true) \
    INST_P(2, LRV, false, UPD, 4, false) INST_P(2, LRV, true, UPD, 4, false) INST_P(2, LRV, true, UPD, 4, true) \
    INST_P(2, LRV, true, UPD, 5, false) INST_P(2, LRV, true, UPD, 5, true) INST_P(2, LRV, true, UPD, 6, false) INST_P(2, LRV, true, UPD, 6, true) \
    INST_P(2, LRV, true, UPD, 7, false) INST_P(2, LRV, true, UPD, 7, true) INST_P(2, LRV, true, UPD, 8, false) INST_P(2, LRV, true, UPD, 8, true) \
    INST_P(2, LRV, true, UPD, 9, false) INST_P(2, LRV, true, UPD, 9, true) INST_P(2, LRV, true, UPD, 10, false) INST_P(2, LRV, true, UPD, 10, true)
INST_P_LR(false, false) INST_P_LR(false, true) INST_P_LR(true, false) INST_P_LR(true, true)
INST_P_WIDE(false, false, false) INST_P_WIDE(false, true, false) INST_P_WIDE(true, false, false) INST_P_WIDE(true, true, false)
INST_P_WIDE(false, false, true) INST_P_WIDE(false, true, true) INST_P_WIDE(true, false, true) INST_P_WIDE(true, true, true)
