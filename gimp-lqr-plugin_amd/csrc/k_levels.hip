// k_levels.hip -- E9 update_mmap for batches, one image's band on SEVERAL compute units, level by level (round 5)
// (gfx950 / CDNA4, wave64; see lqr_common.h for the file map and DESIGN.md section 4.16 for the measurements)
#include "lqr_common.h"
#include "lqr_kernels.h"

// ---------------------------------------------------------------------------
// k_band_levels: the keep-rule sweep of k_band_tiles with the tiles assigned to workgroups PER 32-ROW LEVEL.
//
// k_band_tiles (round 4) parks 8 + 4 workgroups per image on fixed 64-column tiles for the whole launch; 62 % of the
// tile-blocks they sit on are inactive (they only forward a row), and at 64 images their 768 x 2 waves x 256 registers
// starve the sibling streams' carves.  Here an image has P "slots" (workgroups of two turn-taking waves, as there), and
// what a slot works on is decided level by level:
//   * level L = rows [32 L, 32 L + 32).  A_L = the ACTIVE tiles of the level = tiles whose own 64 columns can change in
//     it: a carve-touched pixel of the level's rows within 32 columns of them, or a pixel that changed on the last row of
//     level L - 1 inside them or inside the adjacent 32 columns of a neighbour tile (a change moves one column per row).
//     Nothing else can change (any superset of the pixels with changed inputs leaves liblqr's memory),
//     so inactive tiles are not computed, not stored, not resident -- they cost nothing.
//   * tile t is processed by slot t mod P: own 64 columns + 32-column halos recomputed from the stored inputs, 2 px per
//     lane, the 32 rows staged in registers (exactly k_band_tiles' row loop).  The tiles of a level do not depend on each
//     other, so when a slot has TWO active tiles in a level (a band wider than P tiles) its two waves take one each, side by
//     side (the second with a synchronous load: its prefetch was for the next level).  THREE on one slot: every slot sees
//     that in the same A_L and the image stops there -- flags[FLAG_OVF_ROW] = 32 L, k_dp_sweep<UPDATE> redoes the rows
//     below (a band wider than 2 P tiles; counted).
//   * a LEVEL BARRIER through memory replaces the pairwise hand-over AND the in-place rule: after its level a slot
//     publishes the last row of its tile's own columns as data-tagged granules ({m, tag}, one write-through store each)
//     and one word per tile {tag, tile, changed: own / left 32 / right 32}; the wave that takes the slot's next level polls all
//     2 P words (one load, lanes 0 .. 2 P - 1; the granules it expects to need ride along in the same poll) -- when they carry the level's tag every slot has finished the level: that is the
//     barrier, the words give A_{L+1}, and the granules of the neighbouring active tiles give the halo's row above.
//     Columns whose tile was not active in level L come from memory (nothing changed there).  Level L's results are
//     stored only after the partner wave has passed that barrier: every slot has then CONSUMED its inputs of level L, so
//     in place is safe (a slot's halo columns are its neighbours' own columns); inputs of later levels are rows nobody
//     writes before those levels' own barriers.
//   * no reserve tiles, no requests, no edge watches, no inactive forwarding: the only spin is the level barrier.
//   * every word and granule is published TWICE: the write-through copy above (visible from every XCD: the protocol rests on it
//     alone) and a "near" copy at + near_off written with a plain store, which stays in the writer's L2.  An image's slots are
//     observed to sit on ONE XCD (the grid mapping below), whose L2 then hands the near copy over without the trip to memory
//     and back; the poll reads the near copy three times out of four and the write-through copy the fourth, so a placement
//     that puts the slots on different XCDs only polls slower.  A stale near line can never be taken for a fresh one: tags.
//     (+3 % at 8 and 16 images, +4 % at 48: DESIGN.md 4.16; lqrhip_band_levels_debug(4) turns the near copy off.)
// tags = epoch << 10 | (level + 1): nothing is cleared between launches.  Inputs are prefetched two levels ahead for the
// tile the slot is expected to have then (the same one, or the tile of its residue nearest to the seam); a wrong guess
// costs a synchronous load, never a result.
// Grid: 8 * ceil(images / 8) * P workgroups (see the mapping at the top of the kernel), all co-resident (bounded spins, DEVERR_TILE_TIMEOUT as in k_dp_tile_p).
// ---------------------------------------------------------------------------
// [0] images stopped by three active tiles on one slot, [1] synchronous (mispredicted or second-tile) loads, [2] tile-levels
// processed, [3] slot-levels idle, [4] levels in which a slot had two tiles
__device__ unsigned long long g_lv_stats[8];
#ifdef LQR_TIMING
// per slot of image 0 and wave: cycles in [0] barrier poll (wait_level), [1] active set + tile choice, [2] waiting for the partner's poll,
// [3] stores, [4] loads (issue; synchronous ones include the wait), [5] row above, [6] the 32 rows, [7] publish, [8] LDS barrier, [9] whole kernel
__device__ unsigned long long g_lv_time[16][2][10];
#define LTT(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); ltt[i] += t__ - ltprev; ltprev = t__; } while (0)
extern "C" int lqrhip_band_levels_timing(unsigned long long *out) { (void) hipDeviceSynchronize(); return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lv_time), sizeof(unsigned long long) * 320) == hipSuccess ? 0 : -1; }
#else
#define LTT(i) do { } while (0)
#endif
__device__ int g_lv_dbg;               // experiment switches (lqrhip_band_levels_debug): 1 no speculative granule fetch, 2 sleep longer in the poll, 4 no near copy, 8 an image's slots on different XCDs

// timing jitter for the soak build (make EXTRA=-DLQR_JITTER; switch 16 of lqrhip_band_levels_debug): pseudo-random sleeps at the protocol's
// hand-over sites, as in k_dp_tile_p; the product build has no site (any edit of this kernel can flip its row schedule)
#ifdef LQR_JITTER
#define LJIT(site) do { if (dbg & 16) { unsigned h__ = (unsigned) (slot * 131 + L * 17 + (site) * 7 + epoch * 2654435761u + q * 40503u + image * 977u); h__ ^= h__ >> 13; h__ *= 0x5bd1e995u; h__ ^= h__ >> 15; \
    for (unsigned i__ = h__ & 15u; i__ > 0; i__--) __builtin_amdgcn_s_sleep(127); } } while (0)
#else
#define LJIT(site) do { } while (0)
#endif
// a value every lane holds (read from LDS), as a scalar
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
    return ((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (v >> 32)) << 32) | (unsigned) __builtin_amdgcn_readfirstlane((int) v);
}
struct LvMask {                       // a set of tiles (uniform over the wave); LV_MAX_TILES = 64: one word
    unsigned long long lo;
    __device__ __forceinline__ void set(int t) { lo |= 1ull << t; }
    __device__ __forceinline__ bool has(int t) const { return t >= 0 && t < LV_MAX_TILES && ((lo >> t) & 1ull); }
    __device__ __forceinline__ void set_range(int a, int b) { lo |= ((b - a >= 63) ? ~0ull : ((1ull << (b - a + 1)) - 1ull)) << a; }          // [a, b], 0 <= a <= b < 64
    __device__ __forceinline__ int first() const { return lo ? __builtin_ctzll(lo) : -1; }
    __device__ __forceinline__ int last() const { return lo ? 63 - __builtin_clzll(lo) : -1; }
};

// DELTA = delta_x (1 .. 4): a change moves DELTA columns per row, so a level is HALO / DELTA rows (32, 16, 8, 8); RIGM = a rigidity
// mask scales the rigidity term per pixel (one more 4-byte plane read).  The plain instantiations (1, false) use the 3-neighbour
// row dp_row, the others dp_row_g, exactly as k_dp_tile_p's general instantiations do.
template <bool LR, bool RIG, int DELTA, bool RIGM>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_band_levels(DevCarver *cs, DpK p, int w, int h, int stride, unsigned long long *exch, int epoch, int *dev_err, int P, int n_img)
{
    static_assert(DELTA >= 1 && DELTA <= LQR_FAST_MAX_DELTA && (DELTA <= 4 || RIG) && (RIG || !RIGM), "delta_x 1 .. 10 (5 .. 10: the rigidity form, with a zero table if there is none); a rigidity mask only matters with rigidity");
    constexpr int PX = 2, HALO = 32, OWN = 64, HL = 16, R = lv_rows(DELTA, RIGM), TILE = 128;
    static_assert(R * DELTA <= HALO, "a level's errors stay inside the halo");
    typedef LaneVec<2>::F FV;
    typedef LaneVec<2>::L LV;
    typedef GLOBAL_AS FV GFV;
    typedef GLOBAL_AS LV GLV;
    typedef GLOBAL_AS unsigned long long gu64;
    __shared__ int s_tlo[LV_MAX_LEVELS], s_thi[LV_MAX_LEVELS];      // per level: columns the carve touched on its rows
    __shared__ int s_fail;                        // 1: a spin timed out (results invalid), 2: the image stopped (collision): leave at the next barrier
    __shared__ int s_polled;                      // last level whose barrier this workgroup has passed
    __shared__ unsigned long long s_A[2];         // the active set of a level (by parity), from the wave that received it to its partner
        const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    // A one-dimensional grid of 8 * ceil(n_img / 8) * P workgroups.  Workgroup b is observed to run on XCD b mod 8 (no promise: speed
    // only, the protocol is placement-independent): the P slots of an image are the workgroups b = xcd + 8 (P g + slot), image = 8 g +
    // xcd -- all on one XCD, whose L2 then serves the level hand-overs' reads a little sooner (handoff-1to1: cross-XCD +0.1 - 0.3 us)
    const int dbg = __builtin_amdgcn_readfirstlane(g_lv_dbg);
    const int b_xcd = (int) blockIdx.x & 7, b_k = (int) blockIdx.x >> 3;
    // (debug switch 8: consecutive workgroups = the slots of one image, i.e. on DIFFERENT XCDs -- the placement the near copies do
    // not serve; the tests run it to show that nothing but speed depends on where the slots sit)
    const int slot = (dbg & 8) ? (int) blockIdx.x % P : b_k % P, image = (dbg & 8) ? (int) blockIdx.x / P : (b_k / P) * 8 + b_xcd;
    if (image >= n_img) return;
    const int nblk = (h + R - 1) / R;
    const int ntiles = (w + OWN - 1) / OWN;
    GCarver c = gview(cs[image]);
    if constexpr (RIG && DELTA == 1 && !RIGM) { c.en = uni_ptr(c.en); c.m = uni_ptr(c.m); c.least = uni_ptr(c.least); }      // (see LEAN below)
    const size_t near_off = (size_t) 4 * LV_PMAX + (size_t) 2 * ntiles * OWN;      // the near copies (see the header)
    gu64 *ex_img = (gu64 *) exch + (size_t) image * 2 * near_off;
    gu64 *flagw = ex_img;                                    // [2 parities][LV_PMAX slots][2 tiles]
    gu64 *gran = ex_img + 4 * LV_PMAX;                       // [2 parities][ntiles][OWN]
    const float INF = __int_as_float(0x7f800000);
    const float rig_l = p.rigmap[0], rig_r = p.rigmap[2];
    float rg[2 * DELTA + 1];
#pragma unroll
    for (int i = 0; i < 2 * DELTA + 1; i++) rg[i] = p.rigmap[i];
    // which lanes of the barrier poll hold the words of the slot of tile `lane`, of tile `lane + 1`, of tile `lane - 1`
    const int ix_own = 2 * (lane % P), ix_right = 2 * ((lane + 1) % P), ix_left = 2 * ((lane + P - 1) % P);

    // ---- carve-touched columns per level (as k_band_tiles / k_band_update_tw)
    for (int i = tid; i < nblk; i += 128) { s_tlo[i] = 1 << 30; s_thi[i] = -1; }
    if (tid == 0) { s_fail = 0; s_polled = -1; }        // (flags: through LDS_FLAG once the waves run apart)
    __syncthreads();
    for (int y = tid; y < h; y += 128) {
        const int v0 = c.seam_x[y], vm = c.seam_x[max(y - 1, 0)], vp = c.seam_x[min(y + 1, h - 1)];
        // pixels of row y whose inputs the carve changed: the energy next to the seam on rows y - 1 .. y + 1, and the pixels whose
        // parent window (DELTA either way on row y - 1) straddles the seam there: [min - DELTA - 1, max + DELTA]
        const int t0 = max(min(min(v0, vm), vp) - DELTA - 1, 0), t1 = min(max(max(v0, vm), vp) + DELTA, w - 1);
        atomicMin(&s_tlo[y / R], t0); atomicMax(&s_thi[y / R], t1);
    }
    __syncthreads();
    // tiles within reach of a level's touched columns (own_hi + HALO + 2 >= lo  and  own_lo - HALO - 2 <= hi), as one mask per
    // level, worked out once (the level loop reads one LDS word)
    __shared__ unsigned long long s_tm[LV_MAX_LEVELS + 2];
    for (int L = tid; L < nblk + 2; L += 128) {
        unsigned long long m = 0ull;
        if (L < nblk && s_thi[L] >= s_tlo[L]) {
            const int a = max(0, (s_tlo[L] - (OWN - 1 + HALO + 2) + OWN - 1) / OWN), b = min(ntiles - 1, (s_thi[L] + HALO + 2) / OWN);
            m = ((b - a >= 63) ? ~0ull : ((1ull << (b - a + 1)) - 1ull)) << a;
        }
        s_tm[L] = m;
    }
    __syncthreads();
    auto touch_mask = [&](int L) -> LvMask {
        LvMask m;
        m.lo = uni64(s_tm[L]);
        return m;
    };
    // tiles of this slot's residue (t mod P == slot), and of residue 0: no integer division inside the level loop (a lone wave pays
    // ~40 instructions x 4-5 cycles for each: five of them were 2 000 cycles per level)
    unsigned long long res_mine = 0ull;
    for (int t = slot; t < LV_MAX_TILES; t += P) res_mine |= 1ull << t;
    res_mine = uni64(res_mine);
    // the which-th (0, 1) tile of this slot's residue in a set, -1: none
    auto my_tile = [&](const LvMask &A, int which) -> int {
        unsigned long long m = A.lo & res_mine;
        if (which && m) m &= m - 1ull;
        return m ? __builtin_ctzll(m) : -1;
    };
    // does ANY slot have three tiles in the set?  (identical decision in every slot)
    auto any_collision = [&](const LvMask &A) -> bool {
        const int a = A.first(), b = A.last();
        if (a < 0 || b - a < 2 * P) return false;            // a window of 2 P consecutive tiles: at most two per residue
        unsigned long long res_zero = 0ull;                  // (worked out here: the wide window is rare, and two scalar registers held
        for (int t = 0; t < LV_MAX_TILES; t += P) res_zero |= 1ull << t;      //  across the row loop cost it its free compare registers)
        for (int r = 0; r < P; r++)
            if (__popcll(A.lo & (res_zero << r)) > 2) return true;
        return false;
    };
    // the tile of this slot's residue nearest to [a, b] (-1: the slot has no tile in the image)
    auto nearest_tile = [&](int a, int b) -> int {
        const unsigned long long up_m = res_mine & (~0ull << a), dn_m = res_mine & (b >= 63 ? ~0ull : ((1ull << (b + 1)) - 1ull));
        const int up = up_m ? __builtin_ctzll(up_m) : -1, dn = dn_m ? 63 - __builtin_clzll(dn_m) : -1;
        if (up >= 0 && up <= b) return up;                   // inside
        if (up < 0 || up >= ntiles) return dn;
        if (dn < 0) return up;
        return (up - b <= a - dn) ? up : dn;
    };

    // ---- geometry of the tile staged in this wave's registers
    int cur_t = -1, cur_L = -1;
    bool cur_full = false;
    int x0 = 0;
    unsigned lo_off = 0;
    // The per-lane predicates of the staged tile (inside the image? an own column? a tile that reaches over the border?).  As booleans
    // each is a pair of scalar registers held across the whole level loop, and scalar registers are what the row loop runs out of:
    // its compares then go through VCC one after the other (4 hazard nops per row, -8 %).  The rigidity
    // instantiations (LEAN) therefore work them out from x0 where they are needed -- one unsigned compare each, on a copy the
    // compiler cannot see through -- and get the fast row schedule (296 -> 275 us at 16 images); the plain instantiations have the
    // fast schedule WITH the booleans and lose 4 % without them (255 -> 267 us at 8 images: 38 more s_waitcnt in the rows), so
    // they keep them.  tests/test_kernel_budgets.py watches both.
    constexpr bool LEAN = RIG && DELTA == 1 && !RIGM;       // (measured: delta_x 2 with rigidity 440 -> 456 us WITH it, so only here)
    bool in[PX] = {false, false}, own = false, interior = false;
    const bool own_lane = lane >= HL && lane < 64 - HL;
    auto X0 = [&]() -> int { int v = x0; asm volatile("" : "+v"(v)); return v; };
    auto in_k = [&](int k) -> bool { if constexpr (LEAN) return (unsigned) (X0() + k) < (unsigned) w; else return in[k]; };
    auto is_own = [&]() -> bool { if constexpr (LEAN) return own_lane && X0() < w; else return own; };
    auto is_interior = [&]() -> bool {
        if constexpr (LEAN) { const int t0 = __builtin_amdgcn_readfirstlane(X0()); return t0 >= 0 && t0 + TILE <= w; }
        else return interior;
    };
    auto set_tile = [&](int t) {
        x0 = t * OWN - HALO + PX * lane;
        lo_off = (unsigned) min(max(x0, 0), stride - PX);
        if constexpr (!LEAN) {
#pragma unroll
            for (int k = 0; k < PX; k++) in[k] = x0 + k >= 0 && x0 + k < w;
            own = own_lane && x0 < w;
            interior = (x0 - PX * lane >= 0) && (x0 - PX * lane + TILE <= w);
        }
    };
    // the tile that owns this lane's columns when the wave works on tile t, and the lane's first column inside that tile
    auto owner_of = [&](int t) -> int { return lane < HL ? t - 1 : lane >= 64 - HL ? t + 1 : t; };
    const int owner_col = lane < HL ? HALO + PX * lane : lane >= 64 - HL ? PX * (lane - (64 - HL)) : PX * (lane - HL);
    FV q_e[R], q_mo[R], q_ab, q_rf[RIGM ? R : 1];
    LV q_lo[R];
    q_ab[0] = q_ab[1] = INF;
    auto issue_full = [&](int L) {              // inputs of level L for the tile set_tile() chose, and the row above them
        const int ybase = L * R;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const unsigned row = (unsigned) min(ybase + r, h - 1) * (unsigned) stride;
            const unsigned ro = row + lo_off, ro4 = (row << 2) + (lo_off << 2);
            q_e[r] = *(const GFV *) ((const gu8 *) c.en + ro4);
            q_mo[r] = *(const GFV *) ((const gu8 *) c.m + ro4);
            q_lo[r] = *(const GLV *) (c.least + ro);
            if (RIGM) q_rf[RIGM ? r : 0] = *(const GFV *) ((const gu8 *) c.rig + ro4);
        }
        q_ab = *(const GFV *) ((const gu8 *) c.m + ((((unsigned) max(ybase - 1, 0) * (unsigned) stride) + lo_off) << 2));
    };
    float mp[PX] = {INF, INF};
    int acc_l0 = 0, acc_l1 = 0;                 // XOR of old and new m on the level's last row (pixel 0 / 1 of the lane)
    auto batch_u = [&](int ybase) {
        const int nr = min(R, h - ybase);       // (the image's last level computes surplus rows from copies of its last row)
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
            if (r == R - 1) {
                const int msk = (r < nr) ? -1 : 0;
                acc_l0 = (__float_as_int(mc[0]) ^ __float_as_int(mo[0])) & msk;
                acc_l1 = (__float_as_int(mc[1]) ^ __float_as_int(mo[1])) & msk;
            }
#pragma unroll
            for (int k = 0; k < PX; k++) { mp[k] = mc[k]; q_mo[r][k] = mc[k]; }
            q_lo[r] = (LV) lnew;
        }
    };
    auto store_u = [&](int ybase) {
        const bool own_ = is_own();
        const unsigned inc = own_ ? (unsigned) stride : 0u, inc4 = inc * 4u;
        unsigned so = own_ ? (unsigned) ybase * (unsigned) stride + (unsigned) x0 : (unsigned) h * (unsigned) stride + (unsigned) (PX * lane), so4 = so * 4u;
        const int nr = min(R, h - ybase);
#pragma unroll
        for (int r = 0; r < R; r++, so += inc, so4 += inc4) {
            if (r < nr) {
                *(GFV *) ((gu8 *) c.m + so4) = q_mo[r];
                *(GLV *) (c.least + so) = q_lo[r];
            }
        }
    };
    // The barrier that ends level L - 1 (L >= 1): wait until all 2 P words of the slots carry its tag.  With spec_t >= 0 the
    // granules of the row above level L for tile spec_t (the tile this wave expects to have) are fetched in the same poll, so
    // that barrier, active set and halo cost ONE round trip through memory; prev = A_{L-1} says which tiles published.
    // Returns 0, or 1 on a time-out.  fw: this lane's word (lanes < 2 P); g: this lane's granules (valid if spec_t was right).
    auto wait_level = [&](int L, int spec_t, const LvMask &prev, unsigned long long &fw, unsigned long long (&g)[PX]) -> int {
        const unsigned want = ((unsigned) epoch << 10) | (unsigned) L;
        gu64 *fsrc = flagw + (size_t) ((L - 1) & 1) * 2 * LV_PMAX + (lane < 2 * P ? lane : 0);
        const bool need_g = spec_t >= 0 && prev.has(owner_of(spec_t)) && (in_k(0) || in_k(1));
        gu64 *gsrc = gran + ((size_t) ((L - 1) & 1) * ntiles + (need_g ? owner_of(spec_t) : 0)) * OWN + (need_g ? owner_col : 0);
        int sp = 0;
        while (true) {
            const size_t off = (!(dbg & 4) && (sp & 3) != 3) ? near_off : 0;
            fw = __hip_atomic_load(fsrc + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (spec_t >= 0) {
#pragma unroll
                for (int k = 0; k < PX; k++) g[k] = __hip_atomic_load(gsrc + off + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const bool okf = __all(lane >= 2 * P || (unsigned) (fw >> 32) == want);
            bool okg = true;
#pragma unroll
            for (int k = 0; k < PX; k++) okg &= (unsigned) (g[k] >> 32) == want;
            if (okf && __all(!need_g || okg)) return 0;
            if (dbg & 2) __builtin_amdgcn_s_sleep(16); else if (okf || sp < 8) __builtin_amdgcn_s_sleep(1); else if (sp < 64) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(48);
            ++sp;
            if ((sp & 255) == 0 && dev_failed(dev_err)) return 1;
            if (sp > (1 << 18)) { if (lane == 0) dev_fail(dev_err, DEVERR_TILE_TIMEOUT); return 1; }
        }
    };
    // the row above level L (L >= 1) for the tile set_tile() chose: per column from its tile's granules if that tile was active
    // in level L - 1, else from memory (q_ab: nothing changed there).  have_g: g holds this tile's granules already.
    auto row_above = [&](int L, int t, const LvMask &prev, bool have_g, unsigned long long (&g)[PX]) -> int {
        const int u = owner_of(t);
        const bool from_gran = prev.has(u) && (in_k(0) || in_k(1));
        if (!have_g) {
            gu64 *src = gran + ((size_t) ((L - 1) & 1) * ntiles + (from_gran ? u : 0)) * OWN + (from_gran ? owner_col : 0);
            const unsigned want = ((unsigned) epoch << 10) | (unsigned) L;
            int sp = 0;
            while (true) {
                const size_t off = (!(dbg & 4) && (sp & 3) != 3) ? near_off : 0;
#pragma unroll
                for (int k = 0; k < PX; k++) g[k] = __hip_atomic_load(src + off + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int k = 0; k < PX; k++) ok &= (unsigned) (g[k] >> 32) == want;
                if (__all(ok || !from_gran)) break;
                __builtin_amdgcn_s_sleep(1);                 // (the words were there: the granules are on their way)
                if (++sp > (1 << 16)) { if (lane == 0) dev_fail(dev_err, DEVERR_TILE_TIMEOUT); return 1; }
            }
        }
#pragma unroll
        for (int k = 0; k < PX; k++) mp[k] = !in_k(k) ? INF : from_gran ? __uint_as_float((unsigned) g[k]) : q_ab[k];
        return 0;
    };
    // ---- the level loop.  ONE site each for the loads, the 32 rows and the stores of the staging registers (a second site of any of
    // them makes the register allocator keep two sets: 330 spilled VGPRs).  Wave q receives the barrier of the levels of its
    // parity and takes the slot's FIRST tile there; in the other levels it stores what it holds, takes the slot's SECOND tile if
    // there is one, else prefetches its own next level.  Iteration nblk only closes the last level.
    unsigned n_proc = 0, n_idle = 0, n_sync = 0, n_two = 0;             // (32 bits: scalar registers are what the row loop is short of)
    int hold_L = -1;                            // the level whose (unstored) results sit in this wave's registers
#ifdef LQR_TIMING
    unsigned long long ltt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ltprev = __builtin_readcyclecounter();
    const unsigned long long ltstart = ltprev;
#endif
    for (int L = 0; L <= nblk; L++) {
        const bool mine = (L & 1) == q;
        int t = -1;
        bool have_g = false, passed = false;
        unsigned long long g[PX] = {0ull, 0ull};
        LvMask A = {0ull}, A_prev = {0ull};
        if (L > 0) A_prev.lo = uni64(s_A[(L - 1) & 1]);       // (written before the barrier that ended iteration L - 1)
        LTT(8);
        if (mine) {
            // ---- the barrier of level L - 1 and the active set of level L
            unsigned long long fw = 0;
            A = touch_mask(L);
            passed = true;
            if (L > 0) {
                const int spec_t = (hold_L < 0 && cur_full && cur_L == L && L < nblk && !(dbg & 1)) ? cur_t : -1;
                LJIT(1);
                const int wl_rc = wait_level(L, spec_t, A_prev, fw, g);
                LJIT(2);
                LTT(0);
                if (wl_rc) { LDS_FLAG(s_fail) = 1; passed = false; }
                else if (L < nblk) {
                    // tile `lane` is active in level L iff its own slot's words say "own pixels changed", or the tile to its right says
                    // "left 32 changed", or the tile to its left "right 32 changed" (six words, fetched across the lanes; one ballot)
                    const int wl = (int) (unsigned) fw;
                    bool act = false;
                    unsigned wo[2], wr[2], wlf[2];          // (all six fetches issued before the first is waited for: one LDS latency instead of two)
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        wo[i] = (unsigned) __builtin_amdgcn_ds_bpermute((ix_own + i) << 2, wl);
                        wr[i] = (unsigned) __builtin_amdgcn_ds_bpermute((ix_right + i) << 2, wl);
                        wlf[i] = (unsigned) __builtin_amdgcn_ds_bpermute((ix_left + i) << 2, wl);
                    }
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        act |= (wo[i] & 0x900u) == 0x900u && (int) (wo[i] & 0xffu) == lane;
                        act |= (wr[i] & 0xa00u) == 0xa00u && (int) (wr[i] & 0xffu) == lane + 1;
                        act |= (wlf[i] & 0xc00u) == 0xc00u && (int) (wlf[i] & 0xffu) == lane - 1;
                    }
                    A.lo |= __ballot(act && lane < ntiles);
                    have_g = spec_t >= 0;
                }
            }
            if (lane == 0) { s_A[L & 1] = A.lo; if (passed) LDS_FLAG(s_polled) = L; }        // level L - 1 may be stored now (not after a time-out)
        } else {
            int spins = 0;
            while (LDS_FLAG(s_polled) < L && LDS_FLAG(s_fail) != 1 && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
            passed = LDS_FLAG(s_polled) >= L;
            if (passed) A.lo = uni64(s_A[L & 1]);
            else if (LDS_FLAG(s_fail) != 1) { LDS_FLAG(s_fail) = 1; if (lane == 0) dev_fail(dev_err, DEVERR_TILE_TIMEOUT); }     // (never fall through silently)
            LTT(2);
        }
        // ---- STORE: level L - 1 is final and every slot has consumed its inputs
        LTT(1);
        LJIT(3);
        if (hold_L >= 0 && passed) { store_u(hold_L * R); hold_L = -1; }
        LTT(3);
        if (L == nblk) break;
        if (mine && passed && any_collision(A)) {
            // three active tiles on one slot: the image stops here (every slot decides the same); rows from 32 L on are the sweep's
            if (slot == 0 && lane == 0) { __hip_atomic_fetch_min(c.flags + FLAG_OVF_ROW, L * R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(&g_lv_stats[0], 1ull); }
            LDS_FLAG(s_fail) = 2;
            passed = false;
        }
        // (no look at s_fail here: a partner that stops the image or times out in this iteration is not waited for -- what this wave
        // then computes or publishes for level L is never stored nor read, every slot takes the same decision and leaves at the
        // barrier below; an LDS read is ~130 cycles of a level's ~8 700.  The read in front of the word store STAYS: without it
        // the compiler's schedule of the row loop flips to the slow one)
        if (passed) t = my_tile(A, mine ? 0 : 1);
        if (!mine && t < 0 && passed && lane == 0) {       // no second tile: said right away
            const unsigned long long tag = (unsigned long long) (((unsigned) epoch << 10) | (unsigned) (L + 1)) << 32;
            gu64 *fdst = flagw + (size_t) (L & 1) * 2 * LV_PMAX + 2 * slot + 1;
            if (!(dbg & 4)) __hip_atomic_store(fdst + near_off, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(fdst, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- LOAD: the tile of this level if it is not what was prefetched (synchronous), else -- the other wave -- the prefetch for
        // its next level: the tile this slot is expected to have then
        {
            int ld_t = -1, ld_L = L;
            if (t >= 0) {
                if (!(cur_full && cur_t == t && cur_L == L)) { ld_t = t; n_sync++; have_g = false; }
                if (!mine) n_two++;
            } else if (!mine && passed && L + 1 < nblk) {
                LvMask M = touch_mask(L + 1);
                M.lo |= A.lo;
                int pt = my_tile(M, 0);
                if (pt < 0 && M.first() >= 0) { pt = nearest_tile(M.first(), M.last()); if (pt < M.first() - 1 || pt > M.last() + 1 || pt >= ntiles) pt = -1; }
                ld_t = pt; ld_L = L + 1;
                cur_t = -1; cur_L = ld_L; cur_full = false;
            }
            LTT(1);
            LJIT(4);
            if (ld_t >= 0) { set_tile(ld_t); issue_full(ld_L); cur_t = ld_t; cur_L = ld_L; cur_full = true; }
#ifdef LQR_TIMING
            if (t >= 0 && ld_t >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            LTT(4);
        }
        // ---- the 32 rows of the tile, its last row to the granules
        unsigned word = 0;
        if (t >= 0) {
            n_proc++;
            if (L > 0 && row_above(L, t, A_prev, have_g, g)) LDS_FLAG(s_fail) = 1;
#ifdef LQR_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            LTT(5);
            if (!is_interior()) {
                // outside the image the energy AND the old value become +inf (see k_dp_tile_p)
                const bool ik[PX] = {in_k(0), in_k(1)};
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int k = 0; k < PX; k++) { q_e[r][k] = ik[k] ? q_e[r][k] : INF; q_mo[r][k] = ik[k] ? q_mo[r][k] : INF; }
            }
            batch_u(L * R);
            LTT(6);
            hold_L = L;
            cur_full = false;                    // (the registers hold results now)
            LJIT(5);
            if (L + 1 < nblk && own_lane) {
                gu64 *dst = gran + ((size_t) (L & 1) * ntiles + t) * OWN + PX * (lane - HL);
                const unsigned long long tag = (unsigned long long) (((unsigned) epoch << 10) | (unsigned) (L + 1)) << 32;
#pragma unroll
                for (int k = 0; k < PX; k++) {
                    if (!(dbg & 4)) __hip_atomic_store(dst + near_off + k, tag | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(dst + k, tag | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const bool chg = own_lane && (acc_l0 | acc_l1) != 0;
            const bool any_o = __any(chg), any_l = __any(chg && lane < 32), any_r = __any(chg && lane >= 32);
            word = 0x800u | (any_o ? 0x100u : 0u) | (any_l ? 0x200u : 0u) | (any_r ? 0x400u : 0u) | (unsigned) t;
        } else if (mine && passed) n_idle++;
        // the slot's word for this wave's tile of the level (its granules were issued above; a reader checks their tags itself)
        LJIT(6);
        if (lane == 0 && (t >= 0 || mine) && LDS_FLAG(s_fail) == 0) {
            const unsigned long long tag = (unsigned long long) (((unsigned) epoch << 10) | (unsigned) (L + 1)) << 32;
            gu64 *fdst = flagw + (size_t) (L & 1) * 2 * LV_PMAX + 2 * slot + (mine ? 0 : 1);
            if (!(dbg & 4)) __hip_atomic_store(fdst + near_off, tag | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(fdst, tag | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        LTT(7);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        LTT(8);
        if (LDS_FLAG(s_fail)) return;    // 1: a time-out, results are invalid anyway; 2: the image stopped at this level (level L - 1 was stored above)
    }
#ifdef LQR_TIMING
    if (image == 0 && lane == 0 && slot < 16) { ltt[9] = __builtin_readcyclecounter() - ltstart; for (int i = 0; i < 10; i++) g_lv_time[slot][q][i] = ltt[i]; }
#endif
    if (lane == 0) { atomicAdd(&g_lv_stats[2], (unsigned long long) n_proc); atomicAdd(&g_lv_stats[3], (unsigned long long) n_idle); if (n_two) atomicAdd(&g_lv_stats[4], (unsigned long long) n_two); }
    if (lane == 0 && n_sync) atomicAdd(&g_lv_stats[1], (unsigned long long) n_sync);
}

extern "C" void lqrhip_band_levels_debug(int v) { if (lqrhip_init() < 0) return; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_lv_dbg), &v, sizeof v); }
extern "C" int lqrhip_band_levels_stats(unsigned long long *out, int reset)
{
    if (lqrhip_init() < 0) return -1;          // the symbol of the device the library selected (LOCAL_RANK), not device 0's
    (void) hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lv_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_lv_stats), z, sizeof z); }
    return 0;
}

// ---- the instantiations the shim launches (lqr_kernels.h declares them)
#define INST_LV(...) template __global__ void k_band_levels<__VA_ARGS__>(DevCarver *, DpK, int, int, int, unsigned long long *, int, int *, int, int);
#define INST_LV_LR(LRV) INST_LV(LRV, false, 1, false) INST_LV(LRV, true, 1, false) INST_LV(LRV, true, 1, true) \
    INST_LV(LRV, false, 2, false) INST_LV(LRV, true, 2, false) INST_LV(LRV, true, 2, true) \
    INST_LV(LRV, false, 3, false) INST_LV(LRV, true, 3, false) INST_LV(LRV, true, 3, true) \
    INST_LV(LRV, false, 4, false) INST_LV(LRV, true, 4, false) INST_LV(LRV, true, 4, true) \
    INST_LV(LRV, true, 5, false) INST_LV(LRV, true, 5, true) INST_LV(LRV, true, 6, false) INST_LV(LRV, true, 6, true) \
    INST_LV(LRV, true, 7, false) INST_LV(LRV, true, 7, true) INST_LV(LRV, true, 8, false) INST_LV(LRV, true, 8, true) \
    INST_LV(LRV, true, 9, false) INST_LV(LRV, true, 9, true) INST_LV(LRV, true, 10, false) INST_LV(LRV, true, 10, true)
INST_LV_LR(false) INST_LV_LR(true)
