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
//     lane, the 32 rows staged in registers (exactly k_band_tiles' row loop).  Two active tiles on one slot in one level
//     (a band wider than P tiles, or two distant clusters): every slot sees that in the same A_L and the image stops
//     there -- flags[FLAG_OVF_ROW] = 32 L, k_dp_sweep<UPDATE> redoes the rows below (rare; counted).
//   * a LEVEL BARRIER through memory replaces the pairwise hand-over AND the in-place rule: after its level a slot
//     publishes the last row of its tile's own columns as data-tagged granules ({m, tag}, one write-through store each)
//     and ONE word {tag, tile, changed: own / left 32 / right 32}; the wave that takes the slot's next level polls all P
//     words (one load, lanes 0 .. P - 1) -- when they carry the level's tag every slot has finished the level: that is the
//     barrier, the words give A_{L+1}, and the granules of the neighbouring active tiles give the halo's row above.
//     Columns whose tile was not active in level L come from memory (nothing changed there).  Level L's results are
//     stored only after the partner wave has passed that barrier: every slot has then CONSUMED its inputs of level L, so
//     in place is safe (a slot's halo columns are its neighbours' own columns); inputs of later levels are rows nobody
//     writes before those levels' own barriers.
//   * no reserve tiles, no requests, no edge watches, no inactive forwarding: the only spin is the level barrier.
// tags = epoch << 9 | (level + 1): nothing is cleared between launches.  Inputs are prefetched two levels ahead for the
// tile the slot is expected to have then (the same one, or the tile of its residue nearest to the seam); a wrong guess
// costs a synchronous load, never a result.
// Grid (P, images), all co-resident (bounded spins, DEVERR_TILE_TIMEOUT as in k_dp_tile_p).
// ---------------------------------------------------------------------------
// [0] images stopped by two active tiles on one slot, [1] synchronous (mispredicted) loads, [2] tile-levels processed,
// [3] slot-levels idle
__device__ unsigned long long g_lv_stats[8];

struct LvMask {                       // a set of tiles (uniform over the wave)
    unsigned long long lo, hi;
    __device__ __forceinline__ void set(int t) { if (t < 64) lo |= 1ull << t; else hi |= 1ull << (t - 64); }
    __device__ __forceinline__ bool has(int t) const { return t >= 0 && t < LV_MAX_TILES && ((t < 64 ? lo >> t : hi >> (t - 64)) & 1ull); }
    __device__ __forceinline__ void set_range(int a, int b)          // [a, b], 0 <= a, b < LV_MAX_TILES
    {
        for (int t = a; t <= b; t++) set(t);
    }
    __device__ __forceinline__ int first() const { return lo ? __builtin_ctzll(lo) : hi ? 64 + __builtin_ctzll(hi) : -1; }
    __device__ __forceinline__ int last() const { return hi ? 127 - __builtin_clzll(hi) : lo ? 63 - __builtin_clzll(lo) : -1; }
    __device__ __forceinline__ int count() const { return __popcll(lo) + __popcll(hi); }
};

template <bool LR, bool RIG>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_band_levels(DevCarver *cs, DpK p, int w, int h, int stride, unsigned long long *exch, int epoch, int *dev_err)
{
    constexpr int PX = 2, HALO = 32, OWN = 64, HL = 16, R = 32, TILE = 128;
    typedef LaneVec<2>::F FV;
    typedef LaneVec<2>::L LV;
    typedef GLOBAL_AS FV GFV;
    typedef GLOBAL_AS LV GLV;
    typedef GLOBAL_AS unsigned long long gu64;
    __shared__ int s_tlo[BT_MAX_BLK], s_thi[BT_MAX_BLK];      // per level: columns the carve touched on its rows
    __shared__ FV s_mp[64];                       // last row of the slot's tile, handed from wave to wave
    __shared__ int s_fail;                        // 1: a spin timed out (results invalid), 2: the image stopped (collision): leave at the next barrier
    __shared__ volatile int s_polled;             // last level whose barrier this workgroup has passed
    __shared__ unsigned long long s_A[2][2];      // the active set of a level (by parity), from the wave that received it to its partner
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = (int) gridDim.x, slot = (int) blockIdx.x;
    const int nblk = (h + R - 1) / R;
    const int ntiles = (w + OWN - 1) / OWN;
    const GCarver c = gview(cs[blockIdx.y]);
    gu64 *ex_img = (gu64 *) exch + (size_t) blockIdx.y * ((size_t) 2 * LV_PMAX + (size_t) 2 * ntiles * OWN);
    gu64 *flagw = ex_img;                                    // [2][LV_PMAX]
    gu64 *gran = ex_img + 2 * LV_PMAX;                       // [2][ntiles][OWN]
    const float INF = __int_as_float(0x7f800000);
    const float rig_l = p.rigmap[0], rig_r = p.rigmap[2];

    // ---- carve-touched columns per level (as k_band_tiles / k_band_update_tw)
    for (int i = tid; i < nblk; i += 128) { s_tlo[i] = 1 << 30; s_thi[i] = -1; }
    if (tid == 0) { s_fail = 0; s_polled = -1; }
    __syncthreads();
    for (int y = tid; y < h; y += 128) {
        const int v0 = c.seam_x[y], vm = c.seam_x[max(y - 1, 0)], vp = c.seam_x[min(y + 1, h - 1)];
        const int t0 = max(min(min(v0, vm), vp) - 2, 0), t1 = min(max(max(v0, vm), vp) + 1, w - 1);
        atomicMin(&s_tlo[y / R], t0); atomicMax(&s_thi[y / R], t1);
    }
    __syncthreads();
    // tiles within reach of a level's touched columns: own_hi + HALO + 2 >= lo  and  own_lo - HALO - 2 <= hi
    auto touch_first = [&](int L) -> int { const int lo = s_tlo[L]; return max(0, (lo - (OWN - 1 + HALO + 2) + OWN - 1) / OWN); };
    auto touch_last = [&](int L) -> int { const int hi = s_thi[L]; return min(ntiles - 1, (hi + HALO + 2) / OWN); };
    auto touch_mask = [&](int L) -> LvMask {
        LvMask m = {0ull, 0ull};
        if (L < nblk && s_thi[L] >= s_tlo[L]) m.set_range(touch_first(L), touch_last(L));
        return m;
    };
    // the tile of this slot's residue in a set (-1: none; collision = a second one)
    auto my_tile = [&](const LvMask &A, bool &collision) -> int {
        int found = -1;
        collision = false;
        const int a = A.first(), b = A.last();
        if (a < 0) return -1;
        for (int t = a + ((slot - a) % P + P) % P; t <= b; t += P)
            if (A.has(t)) { if (found < 0) found = t; else collision = true; }
        return found;
    };
    // does ANY slot have two tiles in the set?  (identical decision in every slot)
    auto any_collision = [&](const LvMask &A) -> bool {
        const int a = A.first(), b = A.last();
        if (a < 0 || b - a < P) return false;               // a window of P consecutive tiles: all residues distinct
        unsigned seen = 0;
        for (int t = a; t <= b; t++)
            if (A.has(t)) { const unsigned bit = 1u << (t % P); if (seen & bit) return true; seen |= bit; }
        return false;
    };
    // the tile of this slot's residue nearest to [a, b]
    auto nearest_tile = [&](int a, int b) -> int {
        int t1 = a + ((slot - a) % P + P) % P;              // first one >= a
        if (t1 <= b || t1 - P < 0) return t1 < ntiles ? t1 : (t1 - P >= 0 ? t1 - P : -1);
        return (t1 - b <= a - (t1 - P) && t1 < ntiles) ? t1 : t1 - P;
    };

    // ---- geometry of the tile staged in this wave's registers
    int cur_t = -1, cur_L = -1;
    bool cur_full = false;
    int x0 = 0;
    unsigned lo_off = 0;
    bool in[PX] = {false, false}, own = false, interior = false;
    const bool own_lane = lane >= HL && lane < 64 - HL;
    auto set_tile = [&](int t) {
        x0 = t * OWN - HALO + PX * lane;
        lo_off = (unsigned) min(max(x0, 0), stride - PX);
#pragma unroll
        for (int k = 0; k < PX; k++) in[k] = x0 + k >= 0 && x0 + k < w;
        own = own_lane && x0 < w;
        interior = (x0 - PX * lane >= 0) && (x0 - PX * lane + TILE <= w);
    };
    FV q_e[R], q_mo[R], q_ab;
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
            const float left = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[PX - 1]), DPP_WAVE_SHR1, 0xf, 0xf, true));
            const float right = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[0]), DPP_WAVE_SHL1, 0xf, 0xf, true));
            dp_row<PX, LR, RIG, true, false>(mp, left, right, e, mo, (uint32_t) q_lo[r], in, rig_l, rig_r, mc, lnew, ch);
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
    // The barrier that ends level L - 1 (L >= 1): wait until all P slots' words carry its tag; returns 0, or 1 on a time-out.
    // fw: this lane's word (lanes < P).
    auto wait_level = [&](int L, unsigned long long &fw) -> int {
        const unsigned want = ((unsigned) epoch << 9) | (unsigned) L;
        gu64 *src = flagw + ((L - 1) & 1) * LV_PMAX + (lane < P ? lane : 0);
        int sp = 0;
        while (true) {
            fw = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(lane >= P || (unsigned) (fw >> 32) == want)) return 0;
            if (sp < 8) __builtin_amdgcn_s_sleep(1); else if (sp < 64) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(48);
            ++sp;
            if ((sp & 255) == 0 && dev_failed(dev_err)) return 1;
            if (sp > (1 << 18)) { if (lane == 0) dev_fail(dev_err, DEVERR_TILE_TIMEOUT); return 1; }
        }
    };

    unsigned long long n_proc = 0, n_idle = 0, n_sync = 0;
    // ---- first prefetch: wave q takes levels q, q + 2, ...; level 0's active set is its touched tiles, level 1 is guessed
    {
        const int L0 = q;
        if (L0 < nblk) {
            LvMask g = touch_mask(L0);
            if (L0 == 1) { const LvMask g0 = touch_mask(0); g.lo |= g0.lo; g.hi |= g0.hi; }
            bool coll;
            const int t = my_tile(g, coll);
            if (t >= 0) { set_tile(t); issue_full(L0); cur_t = t; cur_L = L0; cur_full = true; }
        }
    }
    LvMask A_prev = {0ull, 0ull};               // A_{L-1}, as this wave knows it
    for (int L = 0; L < nblk; L++) {
        const bool mine = (L & 1) == q;
        int t = -1;
        bool processed = false;
        LvMask A = {0ull, 0ull};
        if (mine) {
            // ---- the barrier of level L - 1 and the active set of level L
            unsigned long long fw = 0;
            A = touch_mask(L);
            if (L > 0) {
                A_prev.lo = s_A[(L - 1) & 1][0]; A_prev.hi = s_A[(L - 1) & 1][1];
                if (wait_level(L, fw)) { s_fail = 1; }
                else {
                    for (int pp = 0; pp < P; pp++) {
                        const unsigned wlo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) fw, pp);
                        if (wlo & 0x800u) {
                            const int tt = (int) (wlo & 0xffu);
                            if (wlo & 0x100u) A.set(tt);
                            if ((wlo & 0x200u) && tt > 0) A.set(tt - 1);
                            if ((wlo & 0x400u) && tt + 1 < ntiles) A.set(tt + 1);
                        }
                    }
                }
            }
            if (lane == 0) { s_A[L & 1][0] = A.lo; s_A[L & 1][1] = A.hi; if (!s_fail) s_polled = L; }        // level L - 1 may be stored now (not after a time-out)
            if (!s_fail && any_collision(A)) {
                // two active tiles on one slot: the image stops here (every slot decides the same); rows from 32 L on are the sweep's
                if (slot == 0 && lane == 0) { __hip_atomic_fetch_min(c.flags + FLAG_OVF_ROW, L * R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(&g_lv_stats[0], 1ull); }
                s_fail = 2;
            }
            bool coll = false;
            if (!s_fail) t = my_tile(A, coll);
            if (t >= 0) {
                processed = true;
                n_proc++;
                if (!(cur_t == t && cur_L == L && cur_full)) { set_tile(t); issue_full(L); cur_t = t; cur_L = L; cur_full = true; n_sync++; }
                // ---- the row above the level: per column from its tile's granules if that tile was active in level L - 1
                // (this slot's own tile: from LDS, its partner wave left it there), else from memory (nothing changed there)
                if (L > 0) {
                    const int u = lane < HL ? t - 1 : lane >= 64 - HL ? t + 1 : t;
                    const bool from_gran = !own_lane && A_prev.has(u) && (in[0] || in[1]);
                    const int col = lane < HL ? HALO + PX * lane : PX * (lane - (64 - HL));             // the neighbour's own column of this lane's first pixel
                    gu64 *src = gran + ((size_t) ((L - 1) & 1) * ntiles + (from_gran ? u : 0)) * OWN + (from_gran ? col : 0);
                    const unsigned want = ((unsigned) epoch << 9) | (unsigned) L;
                    unsigned long long g[PX] = {0ull, 0ull};
                    int sp = 0;
                    while (true) {
#pragma unroll
                        for (int k = 0; k < PX; k++) g[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        bool ok = true;
#pragma unroll
                        for (int k = 0; k < PX; k++) ok &= (unsigned) (g[k] >> 32) == want;
                        if (__all(ok || !from_gran)) break;
                        __builtin_amdgcn_s_sleep(1);                 // (the words were there: the granules are on their way)
                        if (++sp > (1 << 16)) { if (lane == 0) dev_fail(dev_err, DEVERR_TILE_TIMEOUT); s_fail = 1; break; }
                    }
                    const bool own_prev = A_prev.has(t);             // this slot processed the tile in level L - 1: its last row is in LDS
                    const FV v = s_mp[lane];
#pragma unroll
                    for (int k = 0; k < PX; k++)
                        mp[k] = !in[k] ? INF : from_gran ? __uint_as_float((unsigned) g[k]) : (own_lane && own_prev) ? v[k] : q_ab[k];
                }
                if (!interior) {
                    // outside the image the energy AND the old value become +inf (see k_dp_tile_p)
#pragma unroll
                    for (int r = 0; r < R; r++)
#pragma unroll
                        for (int k = 0; k < PX; k++) { q_e[r][k] = in[k] ? q_e[r][k] : INF; q_mo[r][k] = in[k] ? q_mo[r][k] : INF; }
                }
                batch_u(L * R);
                {
                    FV v;
                    v[0] = mp[0]; v[1] = mp[1];
                    s_mp[lane] = v;
                }
                // ---- publish: the last row of the own columns, then the slot's word
                if (L + 1 < nblk && own_lane) {
                    gu64 *dst = gran + ((size_t) (L & 1) * ntiles + t) * OWN + PX * (lane - HL);
                    const unsigned long long tag = (unsigned long long) (((unsigned) epoch << 9) | (unsigned) (L + 1)) << 32;
#pragma unroll
                    for (int k = 0; k < PX; k++) __hip_atomic_store(dst + k, tag | __float_as_uint(mp[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (!s_fail) n_idle++;
            {
                const bool chg = own_lane && processed && (acc_l0 | acc_l1) != 0;
                const bool any_o = __any(chg), any_l = __any(chg && lane < 32), any_r = __any(chg && lane >= 32);
                if (lane == 0) {
                    const unsigned long long word = ((unsigned long long) (((unsigned) epoch << 9) | (unsigned) (L + 1)) << 32) |
                        (processed ? (0x800u | (any_o ? 0x100u : 0u) | (any_l ? 0x200u : 0u) | (any_r ? 0x400u : 0u) | (unsigned) t) : 0u);
                    __hip_atomic_store(flagw + (L & 1) * LV_PMAX + slot, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int fail = *(volatile int *) &s_fail;
        if (fail == 1) return;                   // a time-out: results are invalid anyway
        if (mine) {
            if (fail == 2) return;               // (this wave received the collision itself: nothing of level L was computed)
            // ---- level L is stored once every slot has finished it (the partner's barrier for level L + 1; the last level: ours)
            bool may_store = true;
            if (L + 1 < nblk) {
                int spins = 0;
                while (s_polled < L + 1 && *(volatile int *) &s_fail != 1 && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
                may_store = s_polled >= L + 1;
            } else {
                unsigned long long fw = 0;
                may_store = wait_level(L + 1, fw) == 0;
            }
            if (may_store && processed) store_u(L * R);
            // ---- prefetch for level L + 2 (this wave's next): the tile this slot is expected to have then
            const int L2 = L + 2;
            if (may_store && L2 < nblk && *(volatile int *) &s_fail == 0) {
                int a = A.first(), b = A.last();
                if (s_thi[L2] >= s_tlo[L2]) { const int f2 = touch_first(L2), l2 = touch_last(L2); a = a < 0 ? f2 : min(a, f2); b = b < 0 ? l2 : max(b, l2); }
                if (s_thi[L + 1] >= s_tlo[L + 1]) { const int f1 = touch_first(L + 1), l1 = touch_last(L + 1); a = a < 0 ? f1 : min(a, f1); b = b < 0 ? l1 : max(b, l1); }
                int pt = -1;
                if (a >= 0) {
                    pt = processed ? t : nearest_tile(a, b);
                    if (pt < 0 || pt >= ntiles || pt < a - 1 || pt > b + 1) pt = -1;           // too far from everything that is going on
                }
                cur_t = -1; cur_L = L2; cur_full = false;
                if (pt >= 0) { set_tile(pt); issue_full(L2); cur_t = pt; cur_full = true; }
            }
        } else if (fail == 2) {
            // the partner received a collision at level L: this wave computed level L - 1, whose barrier has been passed -- store it
            // (FLAG_OVF_ROW = 32 L: every row above must be final), then leave
            return;
        }
    }
    if (lane == 0) { atomicAdd(&g_lv_stats[2], n_proc); atomicAdd(&g_lv_stats[3], n_idle); }
    if (lane == 0 && n_sync) atomicAdd(&g_lv_stats[1], n_sync);
}

extern "C" int lqrhip_band_levels_stats(unsigned long long *out, int reset)
{
    (void) hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lv_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_lv_stats), z, sizeof z); }
    return 0;
}

// ---- the instantiations the shim launches (lqr_kernels.h declares them)
#define INST_LV(LRV, RIGV) template __global__ void k_band_levels<LRV, RIGV>(DevCarver *, DpK, int, int, int, unsigned long long *, int, int *);
INST_LV(false, false) INST_LV(false, true) INST_LV(true, false) INST_LV(true, true)
