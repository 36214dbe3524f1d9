// k_backtrack.hip -- E7 build_vpath: seam pick (argmin with liblqr's tie rule) and backtrack; publishes the side the carve moves
// (gfx950 / CDNA4, wave64; see lqr_common.h for the file map and DESIGN.md section 4 for the measurements)
#include "lqr_common.h"
#include "lqr_kernels.h"

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
// moved_unit: bytes the carve that follows moves per pixel of that side, read + write (en, and m + back pointer unless a full DP
// follows, and the rigidity mask if there is one); the sum over all images and seams goes to g_moved_bytes -- what the carve
// could not avoid moving, next to SURVEY 8(d)'s half-row assumption (bench.py: roofline.moved_bytes_per_launch).
__device__ unsigned long long g_moved_bytes;
__device__ __forceinline__ void publish_side(const GCarver &c, int org, int acc, int w, int h, int lane, int moved_unit)
{
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const long long all = (long long) h * (w - 1);
    const int side = (2ll * acc < all) ? 1 : 0;
    if (lane == 0) {
        c.flags[FLAG_ORG_PREV] = org; c.flags[FLAG_SIDE] = side; c.flags[FLAG_ORG] = org + side;
        atomicAdd(&g_moved_bytes, (unsigned long long) (side ? (long long) acc : all - acc) * (unsigned long long) moved_unit);
    }
}
extern "C" int lqrhip_moved_bytes(unsigned long long *out, int reset)
{
    if (lqrhip_init() < 0) return -1;          // the symbol of the device the library selected (LOCAL_RANK), not device 0's
    (void) hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_moved_bytes), sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z = 0; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_moved_bytes), &z, sizeof z); }
    return 0;
}

__global__ __launch_bounds__(VPATH_THREADS) void k_vpath(const DevCarver *cs, int w, int h, int stride, int lr, int delta,
                                                          int log_index, int moved_unit)
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
    publish_side(c, org, acc, w, h, lane, moved_unit);
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
constexpr int vp1_rows(int delta) { return delta == 1 ? 28 : delta == 2 ? 12 : delta == 3 ? 8 : 4; }      // (delta_x 4 .. 7: 4 rows)
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
__global__ __launch_bounds__(VPATH_THREADS) void k_vpath1(const DevCarver *cs, int w, int h, int stride, int lr, int log_index, int moved_unit)
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
    publish_side(c, org, acc, w, h, lane, moved_unit);
}


// ---------------------------------------------------------------------------
// Parallel backtrack (round 6): k_vp_maps -> k_vp_solve, for single images.
//
// k_vpath1 is one wave walking H dependent steps: 57 - 70 us per 4K seam whatever the chip has idle (20 % of a single image's seam
// round).  Two multi-wave rewrites inside the workgroup lost to it (the backtrack on three waves).  The walk is function
// composition -- column at row y - 1 = x + least[y][x] -- and composition is associative, so the chip can do most of it before
// anybody knows where the seam ends:
//   1. k_vp_maps: the rows are cut into chunks of R = VP_REACH / delta_x rows; for EVERY column x the walk across chunk c is done
//      -- a workgroup per 256 columns and chunk stages the chunk's bytes in LDS and every thread walks its R steps there (dependent LDS
//      reads, 4 K independent walks per chunk in flight) -- and written down: the displacement D_c[x] across the chunk (vp_map) and the
//      displacement after every step (vp_path, four steps to a dword, [chunk][step / 4][x]).  No dependence on the argmin.
//   2. k_vp_solve: one workgroup per image finds the argmin of the last row (as k_vpath1), then needs only ONE step per chunk,
//      x <- x + D_c[x]: VP_STAGE chunks at a time, the part of the maps the path can reach from the stage's start (VP_REACH columns
//      per chunk either way) loaded into LDS by all threads at once, then VP_STAGE dependent LDS reads; with every chunk's start
//      known all threads pick the seam's rows out of vp_path (independent loads), record them (seam_x, the session log) and
//      publish the side the carve moves (publish_side).
// Two launches (a third one for the paths, walked again per chunk, cost more than it saved: a dependent launch is ~10 us).
// Same results as k_vpath1 by construction (integer function composition); back pointers marked invalid by the carve do not
// survive to the backtrack (see k_vpath1) -- a stray one is treated as 0, as k_vpath does, so that no walk can leave its window.
// ---------------------------------------------------------------------------
template <int DELTA>
__global__ __launch_bounds__(256) void k_vp_maps(const DevCarver *cs, int w, int h, int stride)
{
    constexpr int R = vp_chunk_rows(DELTA), RD = R * DELTA, WIN = 256 + 2 * RD, R4 = (R + 3) / 4;        // RD <= VP_REACH
    __shared__ __attribute__((aligned(4))) int8_t s_rows[R][WIN + 8];
    const GCarver c = gview(cs[blockIdx.z]);
    const int chunk = blockIdx.y, ytop = h - 1 - chunk * R, nrows = min(R, ytop);      // steps at rows ytop, ytop - 1, .. ytop - nrows + 1 (row 0 has none)
    const int X0 = blockIdx.x * 256, base = X0 - RD, tid = threadIdx.x;
    // stage: bytes [base, base + WIN) of the chunk's rows, as dwords (the plane's origin may make the address unaligned: fine on
    // gfx950); columns outside [0, stride) are not loaded -- no path of the image goes there
    // (all of a thread's loads are issued before the first goes to LDS: one memory round trip per workgroup instead of ~20)
    constexpr int WD = (WIN + 3) / 4, NI = (R * WD + 255) / 256;
    uint32_t v[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) {
        const int i = tid + 256 * k, r = i / WD, d = i - r * WD, col = base + 4 * d;
        v[k] = 0u;
        if (i >= nrows * WD) continue;
        if (col >= 0 && col + 3 < stride) v[k] = *(const gu32 *) (c.least + (size_t) (ytop - r) * stride + col);
        else for (int b = 0; b < 4; b++) if (col + b >= 0 && col + b < stride) v[k] |= (uint32_t) (uint8_t) c.least[(size_t) (ytop - r) * stride + col + b] << (8 * b);
    }
#pragma unroll
    for (int k = 0; k < NI; k++) {
        const int i = tid + 256 * k, r = i / WD, d = i - r * WD;
        if (i < nrows * WD) *(uint32_t *) &s_rows[r][4 * d] = v[k];
    }
    __syncthreads();
    const int x = X0 + tid;
    if (x >= w) return;
    int p = x - base;
    gu32 *path = (gu32 *) c.vp_path + (size_t) chunk * R4 * stride + x;
#pragma unroll
    for (int r4 = 0; r4 < R4; r4++) {
        uint32_t pk = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = 4 * r4 + k;
            pk |= (uint32_t) (uint8_t) (int8_t) (p + base - x) << (8 * k);          // the displacement BEFORE step r: the seam's column on row ytop - r
            if (r < nrows) { const int d = s_rows[r < R ? r : 0][p]; p += (d == LEAST_INVALID) ? 0 : d; }
        }
        path[(size_t) r4 * stride] = pk;
    }
    c.vp_map[(size_t) chunk * stride + x] = (int8_t) (p + base - x);
}

template <int DELTA>
__global__ __launch_bounds__(VPATH_THREADS) void k_vp_solve(const DevCarver *cs, int w, int h, int stride, int lr, int log_index, int moved_unit)
{
    constexpr int R = vp_chunk_rows(DELTA), RD = R * DELTA, S = VP_STAGE, R4 = (R + 3) / 4;
    constexpr int CP = (2 * RD * S + 4 + 15 + 16) & ~15;       // LDS pitch of a stage's rows: the cone [x - RD S, x + RD S] from a base rounded down to 4, in 16-byte units
    const GCarver c = gview(cs[blockIdx.x]);
    const int org = c.flags[FLAG_ORG];           // read before wave 0 publishes the next one
    __shared__ float s_val[VPATH_THREADS / 64];
    __shared__ int s_idx[VPATH_THREADS / 64];
    __shared__ __attribute__((aligned(16))) int8_t s_cone[S][CP];
    extern __shared__ int s_xs[];                // [nchunks + 1]: the seam's column on the first row of every chunk, and on row 0
    __shared__ int s_acc[VPATH_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int xmin = row_argmin(c.m + (size_t) (h - 1) * stride, w, lr, s_val, s_idx);
    const int nchunks = (h - 1 + R - 1) / R;
    int x = max(xmin, 0);                        // uniform
    for (int c0 = 0; c0 < nchunks; c0 += S) {
        const int ns = min(S, nchunks - c0);
        const int base = (x - RD * S) & ~3;      // the path stays inside [x - RD S, x + RD S] during the stage
        // the stage's part of the maps into LDS, 16 bytes per load (4-byte aligned: fine on gfx950); row j is only needed within RD (j + 1)
        // columns of x -- the rest of its cone is not loaded (delta_x 10 at 4K is 22 stages: this loop is what a stage costs)
        // ALL loads of a thread are issued before the first is stored to LDS: as one loop (load, store, load, ...) every iteration waited for
        // its own round trip -- 5 or 6 of them per stage, 9 us per stage, 245 us per 4K seam at delta_x 10
        constexpr int W16 = CP / 16, NI = (S * W16 + VPATH_THREADS - 1) / VPATH_THREADS;
        u32x4 v[NI];
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = tid + k * VPATH_THREADS, j = i / W16, d = i - j * W16, col = base + 16 * d;
            v[k] = (u32x4) {0u, 0u, 0u, 0u};
            if (i >= ns * W16 || col + 15 < x - RD * (j + 1) || col > x + RD * (j + 1)) continue;
            const gi8 *src = c.vp_map + (size_t) (c0 + j) * stride + col;
            if (col >= 0 && col + 15 < stride) v[k] = *(const GLOBAL_AS u32x4 *) src;
            else {
                uint32_t q4[4] = {0u, 0u, 0u, 0u};
                for (int b = 0; b < 16; b++) if (col + b >= 0 && col + b < stride) q4[b >> 2] |= (uint32_t) (uint8_t) src[b] << (8 * (b & 3));
                v[k] = (u32x4) {q4[0], q4[1], q4[2], q4[3]};
            }
        }
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = tid + k * VPATH_THREADS, j = i / W16, d = i - j * W16;
            if (i < ns * W16) *(u32x4 *) &s_cone[j][16 * d] = v[k];
        }
        __syncthreads();
        if (tid == 0) {
            int xx = x;
            for (int j = 0; j < ns; j++) { s_xs[c0 + j] = xx; xx += s_cone[j][xx - base]; }
            s_xs[c0 + ns] = xx;
        }
        __syncthreads();
        x = s_xs[c0 + ns];
    }
    // every row's column: the chunk's start + the recorded displacement before that row's step; row 0 is where the last chunk ends
    gi32 *seam = c.seam_x, *logp = c.seam_log + (size_t) log_index * h;
    int acc = 0;
    for (int y = tid; y < h; y += VPATH_THREADS) {
        int v;
        if (y == 0) v = s_xs[nchunks];
        else {
            const int s = h - 1 - y, chunk = s / R, r = s - chunk * R, xs = s_xs[chunk];
            v = xs + (int) ((const gi8 *) c.vp_path)[(((size_t) chunk * R4 + (r >> 2)) * stride + xs) * 4 + (r & 3)];
        }
        seam[y] = v; logp[y] = v; acc += v;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) s_acc[tid >> 6] = acc;
    __syncthreads();
    if (tid >= 64) return;
    int total = 0;
    if (lane == 0) for (int k = 0; k < VPATH_THREADS / 64; k++) total += s_acc[k];
    publish_side(c, org, total, w, h, lane, moved_unit);
}

// ---- the instantiations the shim launches (lqr_kernels.h declares them)
#define INST_VP(D) template __global__ void k_vp_maps<D>(const DevCarver *, int, int, int); \
    template __global__ void k_vp_solve<D>(const DevCarver *, int, int, int, int, int, int);
INST_VP(1) INST_VP(2) INST_VP(3) INST_VP(4) INST_VP(5) INST_VP(6) INST_VP(7) INST_VP(8) INST_VP(9) INST_VP(10)
template __global__ void k_vpath1<1>(const DevCarver *, int, int, int, int, int, int);
template __global__ void k_vpath1<2>(const DevCarver *, int, int, int, int, int, int);
template __global__ void k_vpath1<3>(const DevCarver *, int, int, int, int, int, int);
template __global__ void k_vpath1<4>(const DevCarver *, int, int, int, int, int, int);
template __global__ void k_vpath1<5>(const DevCarver *, int, int, int, int, int, int);
template __global__ void k_vpath1<6>(const DevCarver *, int, int, int, int, int, int);
template __global__ void k_vpath1<7>(const DevCarver *, int, int, int, int, int, int);
