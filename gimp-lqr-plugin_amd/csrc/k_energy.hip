// k_energy.hip -- E1 working planes, E2 masks, E3/E4 energy map, E6 energy update next to the seam, frozen-plane catch-up
// (gfx950 / CDNA4, wave64; see lqr_common.h for the file map and DESIGN.md section 4 for the measurements)
#include "lqr_common.h"
#include "lqr_kernels.h"

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

// The same from a carver that is NOT flat (round 6: a session redone after a fault, or working planes lost on a multi-size
// image): the carved frame is the pixels of the base layout that have no level yet (vs == 0), in order -- what k_vs_commit
// ranks.  One block per row, ballot-rank compaction; the tail of the row is zero-filled as k_wk_init does.
__global__ __launch_bounds__(256) void k_wk_init_visible(const DevCarver *cs, int w0, int h, int stride, int ch)
{
    __shared__ int s_wave[4];
    const GCarver c = gview_phys(cs[blockIdx.y]);
    const int y = blockIdx.x, tid = threadIdx.x;
    if (y == 0 && tid == 0) { c.flags[FLAG_ORG] = 0; c.flags[FLAG_ORG_PREV] = 0; c.flags[FLAG_SIDE] = 0; }
    const size_t ri = (size_t) y * w0, ro = (size_t) y * stride;
    int carry = 0;
    for (int base = 0; base < w0; base += 256) {
        const int col = base + tid;
        const bool keep = (col < w0) && c.vs[ri + col] == 0;
        int total;
        const int rank = carry + block_rank_256(keep, s_wave, total);
        if (keep && rank < stride) {
            const gu8 *s = c.rgb0 + (ri + col) * ch;
            uint32_t p = 0;
            if (ch == 4) p = *(const gu32 *) s;
            else for (int k = 0; k < ch; k++) p |= (uint32_t) s[k] << (8 * k);
            c.pix[ro + rank] = p;
            if (c.bias) c.bias[ro + rank] = c.bias0 ? c.bias0[ri + col] : 0.0f;
            if (c.rig) c.rig[ro + rank] = c.rig0 ? c.rig0[ri + col] : 0.0f;
        }
        carry += total;
    }
    for (int x = carry + tid; x < stride; x += 256) {
        c.pix[ro + x] = 0u;
        if (c.bias) c.bias[ro + x] = 0.0f;
        if (c.rig) c.rig[ro + x] = 0.0f;
    }
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

// E6 update_emap: recompute en next to the carved seam (w = new width); runs after the carve.
// The packed-pixel (and bias) planes are frozen in the frame they had at seam `epoch`
// of the session; current coordinates are mapped back by undoing seams k..epoch of the
// row (p += (log[j] <= p)), which costs O(k - epoch) per pixel for ~10 pixels per row and
// saves moving 4 (8 with bias) of the 13 bytes per pixel that a carve would otherwise move.
// Brightness samples staged per row: row y is asked for columns [min - 2, max + 1] of the seam over rows
// y-1..y+1 by its own gradient and [min - 1, max] of the seam over rows y-2..y+2 by its neighbours', and the
// seam moves at most delta_x per row: at most max(4*delta_x + 2, 2*delta_x + 4) columns.  EU_NT is a template
// parameter chosen by the launch from delta_x: 12 (delta_x <= 2), 36 (<= 8), 68 (<= 16 = LQRHIP_MAX_DELTA).
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


// ---- the instantiations the shim launches (lqr_kernels.h declares them)
#define INST_EMAP(N) template __global__ void k_emap_full<N>(const DevCarver *, DpK, int, int, int); \
    template __global__ void k_emap_update<N, 12>(const DevCarver *, DpK, int, int, int, int, int); \
    template __global__ void k_emap_update<N, 36>(const DevCarver *, DpK, int, int, int, int, int); \
    template __global__ void k_emap_update<N, 68>(const DevCarver *, DpK, int, int, int, int, int);
INST_EMAP(0) INST_EMAP(1) INST_EMAP(2) INST_EMAP(3) INST_EMAP(4) INST_EMAP(5) INST_EMAP(6)
