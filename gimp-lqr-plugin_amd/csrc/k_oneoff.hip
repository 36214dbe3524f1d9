// k_oneoff.hip -- the one-off passes: visibility map from the seam log (E8), inflate (E14), flatten / read-out compaction (E11, E12), transpose (E11), auto-size mask scan
// (gfx950 / CDNA4, wave64; see lqr_common.h for the file map and DESIGN.md section 4 for the measurements)
#include "lqr_common.h"
#include "lqr_kernels.h"

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
// Fused self-check (round 6): the levels this session wrote -- [2 max_level - 1, l + max_level - 1] -- must each occur exactly
// once in every row (k_vs_commit's contract).  The pass reads every level anyway: a bit per level in LDS (atomicOr) finds a level
// that occurs twice, the row's dup count a missing one.  A failure goes to the host-visible error word (DEVERR_LEVELS); the host
// then does not adopt the inflated planes, rolls the session back and redoes it (host/lqr_carver.c, group_build_maps).
__global__ __launch_bounds__(256) void k_inflate(const InflateDev *jobs, int w0, int w1, int l, int max_level, int *dev_err)
{
    __shared__ int s_wave[4];
    extern __shared__ unsigned s_seen[];          // [(n_levels + 31) / 32]
    const int n_levels = l - max_level + 1, lvl0 = 2 * max_level - 1;
    for (int i = threadIdx.x; i < (n_levels + 31) / 32; i += 256) s_seen[i] = 0u;
    __syncthreads();
    bool twice = false;
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
        if (dup) twice |= (atomicOr(&s_seen[(v - lvl0) >> 5], 1u << ((v - lvl0) & 31)) >> ((v - lvl0) & 31)) & 1u;
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
    if (dev_err && (twice || (tid == 0 && carry != n_levels))) dev_fail(dev_err, DEVERR_LEVELS);
}

// Session self-check before the levels are committed (round 6): the session's seam log must describe seams -- entry k of
// every row inside the frame that seam was found in (0 <= x < wc0 - k), and delta_x-connected from row to row.  Anything
// else means a kernel of the seam loop misbehaved; the host rolls the session back and redoes it on the non-spinning kernels.
// One block per ROWS rows of one image; the log is n_seams x h ints (1.7 MB for 200 seams of a 4K image).
__global__ __launch_bounds__(256) void k_seam_check(const DevCarver *cs, int h, int wc0, int n_seams, int delta, int *dev_err)
{
    const GCarver c = gview_phys(cs[blockIdx.y]);
    const int y = blockIdx.x * 256 + threadIdx.x;
    if (y >= h) return;
    bool bad = false;
    for (int k = 0; k < n_seams; k++) {
        const int x = c.seam_log[(size_t) k * h + y];
        bad |= (x < 0) || (x >= wc0 - k);
        if (y > 0) { const int xu = c.seam_log[(size_t) k * h + y - 1]; bad |= (x - xu > delta) || (xu - x > delta); }
    }
    if (bad) dev_fail(dev_err, DEVERR_SEAMLOG);
}

// Undo a session's commit: levels at or above first_level are this session's (every older level is below it: inflate shifts
// the levels of a finished session to [.., 2 depth - 2] and the next session starts at 2 depth - 1), plus finish_vsmap's w0.
__global__ __launch_bounds__(256) void k_vs_rollback(const DevCarver *cs, size_t n, int first_level, int finish_level)
{
    const GCarver c = gview_phys(cs[blockIdx.y]);
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
        const int v = c.vs[i];
        if (v >= first_level || (finish_level > 0 && v == finish_level)) c.vs[i] = 0;
    }
}

// Fault injection for tests/test_faults_gpu.py (lqrhip_debug_inject): damage one entry of the session's seam log (what = 0: out of
// the frame, 1: a jump of 40 columns), or clear / duplicate one committed level in the base layout (2 / 3)
__global__ void k_inject(const DevCarver *cs, int what, int h, int w0, int log_index, int first_level)
{
    const GCarver c = gview_phys(cs[0]);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int y = h / 2;
    if (what == 0) c.seam_log[(size_t) log_index * h + y] = -7;
    else if (what == 1) { gi32 *e = c.seam_log + (size_t) log_index * h + y; *e = (*e >= 40) ? *e - 40 : *e + 40; }
    else {
        gi32 *row = c.vs + (size_t) y * w0;
        int at = -1, other = -1;
        for (int x = 0; x < w0; x++) { if (row[x] == first_level) at = x; else if (row[x] == first_level + 1) other = x; }
        if (what == 2 && at >= 0) row[at] = 0;
        if (what == 3 && at >= 0 && other >= 0) row[other] = first_level;
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


// ---- the instantiations the shim launches (lqr_kernels.h declares them)
// (no templates)
