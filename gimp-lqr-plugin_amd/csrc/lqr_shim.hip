// lqr_shim.hip -- the host side: the lqrhip_* C ABI of include/lqr_hip.h (plain pointers and sizes) on top of the kernels of
// k_*.hip: device selection, the allocation cache, host <-> device transfers through a pinned ring, batches and their streams,
// lqrhip_seam_step's per-seam launch sequence and the choice of update_mmap form, read-out, reset from device memory,
// the copy ceiling and the seam-map colour ramp.  The three kernels that live here (k_poison_random, k_copy16, k_vmap_ramp)
// are debugging / measuring / one-off aids next to their only callers.
#include "lqr_common.h"
#include "lqr_kernels.h"

static thread_local std::string g_err;
static int g_device = -1;

static int *g_dev_err_host = nullptr;      // hipHostMalloc'ed, mapped
static int *g_dev_err = nullptr;           // its device address

#define HIPCK_VOID(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { g_err = std::string(#expr) + ": " + hipGetErrorString(e__); (void) hipGetLastError(); } } while (0)
#define HIPCK(expr)                                                                   \
    do {                                                                              \
        hipError_t e__ = (expr);                                                      \
        if (e__ != hipSuccess) {                                                      \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e__);               \
            (void) hipGetLastError();                                                 \
            return (e__ == hipErrorOutOfMemory) ? LQRHIP_ENOMEM : LQRHIP_EHIP;        \
        }                                                                             \
    } while (0)

// co-residency bounds for the spin waits of the persistent kernels, set from the occupancy queries in lqrhip_init (dpp_resident_workgroups)
static int g_dpp_max_wgs = 0;
static int g_dpp_max_wgs_plain = 0, g_dpp_max_wgs_general = 0;      // ... of the plain / the delta_x = 2..4, rigidity-mask instantiations
static int g_dpp_max_wgs_px4 = 0;                                   // ... of the plain 4-px instantiations alone (fewer registers than the 2-px ones)
static int g_dpp_max_wgs_levels = 0;                                // ... of k_band_levels
static int g_n_cu = 0;

// ===========================================================================
// host side of the shim
// ===========================================================================
struct LqrHipCarver {
    int ch = 0;
    int w0 = 0, h0 = 0;              // base layout dims
    // base planes
    uint8_t *rgb0 = nullptr;
    int32_t *vs = nullptr;           // owned by roots only
    float *bias0 = nullptr, *rig0 = nullptr;
    // working planes
    int active = 0;
    int stride = 0, wk_h = 0;
    uint32_t *pix = nullptr;
    float *en = nullptr, *m = nullptr, *m2 = nullptr, *bias = nullptr, *rig = nullptr;
    int8_t *least = nullptr, *least2 = nullptr;
    int32_t *seam_x = nullptr, *seam_log = nullptr, *flags = nullptr;
    int8_t *vp_map = nullptr;       // parallel backtrack: chunk displacement maps and paths (allocated on first use)
    int8_t *vp_path = nullptr;
    size_t vp_cap = 0;
    int log_cap = 0, log_h = 0;
    int frozen_epoch = 0;           // pix / bias are in the frame before seam `frozen_epoch` of the session
    LqrHipCarver *root = nullptr;
    std::vector<LqrHipCarver *> aux;
    LqrHipBatch *batch = nullptr;
};

struct LqrHipBatch {
    std::vector<LqrHipCarver *> cs;
    DevCarver *d_desc = nullptr;
    hipStream_t stream = nullptr;
    unsigned long long *exch = nullptr;     // k_dp_tile_p: halo granules per image and tile + finished-tile counters
    size_t exch_elems = 0;
    int exch_ntiles = 0, exch_n = 0, exch_px = 0;      // geometry the exchange area was last laid out for
    int tile_epoch = 0;                     // launches of k_dp_tile_p on this batch (part of the granule tags)
    bool dirty = true;
    int shared_n = 1;                       // ... how many batches of the group there are (lqrhip_batch_set_shared)
    bool shared = false;                    // other batches of the same group run concurrently on their own streams:
                                            // no persistent (spin-waiting, co-residency-dependent) kernels
    bool safe = false;                      // a session is being redone after a fault: kernels without spin waits only
    void *pending_inflate = nullptr;        // the staged planes of lqrhip_inflate / _flatten / _transpose, until lqrhip_planes_commit adopts them (PendingInflate)
};
static void discard_pending(LqrHipBatch *b);
static bool g_no_spin = false;             // set by a spin time-out (check_dev_error): the process stays on the non-spinning kernels
static inline bool no_spin(const LqrHipBatch *b) { return b->safe || g_no_spin; }

struct ProfRec {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double bytes = 0;
};
static int g_prof = 0;                 // 0 off, 1 every kernel, 2 the roofline kernel (k_carve) only
static std::map<std::string, ProfRec> g_profrec;
static hipStream_t g_stream0 = nullptr;

extern "C" const char *lqrhip_last_error(void) { return g_err.c_str(); }

// Workgroups of k_dp_tile_p the device holds at once.  Its tiles spin on their neighbours, so the grid
// must be co-resident: the bound comes from the occupancy query of every instantiation that can be
// launched (the minimum over them), less one workgroup per CU of margin -- the API is known to answer
// one block per CU too many at some SGPR counts (MI355X_MICROARCH.md, residency) -- times the CU count.
// A grid above the bound goes to k_dp_tile (kernel boundaries instead of spin waits).
static int dpp_resident_workgroups(int dev)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 0;
    int per_cu = 1 << 20;
    auto q = [&](auto kern) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 64 * DPP_W, 0) != hipSuccess) { (void) hipGetLastError(); n = 0; }
        per_cu = std::min(per_cu, n);
    };
    q(k_dp_tile_p<4, false, false, false>); q(k_dp_tile_p<4, false, true, false>); q(k_dp_tile_p<4, true, false, false>); q(k_dp_tile_p<4, true, true, false>);
    q(k_dp_tile_p<4, false, false, true>); q(k_dp_tile_p<4, false, true, true>); q(k_dp_tile_p<4, true, false, true>); q(k_dp_tile_p<4, true, true, true>);
    // the 2-px instantiations stage a whole 32-row block (193 VGPRs): their bound is lower, and a grid that is too large for
    // them but fits the 4-px ones must not be sent to k_dp_tile for it
    g_dpp_max_wgs_px4 = std::max(0, per_cu - 1) * prop.multiProcessorCount;
    q(k_dp_tile_p<2, false, false, false>); q(k_dp_tile_p<2, false, true, false>); q(k_dp_tile_p<2, true, false, false>); q(k_dp_tile_p<2, true, true, false>);
    q(k_dp_tile_p<2, false, false, true>); q(k_dp_tile_p<2, false, true, true>); q(k_dp_tile_p<2, true, false, true>); q(k_dp_tile_p<2, true, true, true>);
    // (round 6: the 24-halo-lane geometry of the plain 2-px kernels, px code 3)
    q(k_dp_tile_p<2, false, false, false, 1, false, 24>); q(k_dp_tile_p<2, false, true, false, 1, false, 24>); q(k_dp_tile_p<2, true, false, false, 1, false, 24>); q(k_dp_tile_p<2, true, true, false, 1, false, 24>);
    q(k_dp_tile_p<2, false, false, true, 1, false, 24>); q(k_dp_tile_p<2, false, true, true, 1, false, 24>); q(k_dp_tile_p<2, true, false, true, 1, false, 24>); q(k_dp_tile_p<2, true, true, true, 1, false, 24>);
    g_dpp_max_wgs_plain = std::max(0, per_cu - 1) * prop.multiProcessorCount;
    g_n_cu = prop.multiProcessorCount;
    // the delta_x = 2 / rigidity-mask instantiations (2 px per lane only) hold more registers
    per_cu = 1 << 20;
#define QG(LRV, UPD) q(k_dp_tile_p<2, LRV, true, UPD, 1, true>); q(k_dp_tile_p<2, LRV, false, UPD, 2, false>); q(k_dp_tile_p<2, LRV, true, UPD, 2, false>); q(k_dp_tile_p<2, LRV, true, UPD, 2, true>); \
    q(k_dp_tile_p<2, LRV, false, UPD, 3, false>); q(k_dp_tile_p<2, LRV, true, UPD, 3, false>); q(k_dp_tile_p<2, LRV, true, UPD, 3, true>); \
    q(k_dp_tile_p<2, LRV, false, UPD, 4, false>); q(k_dp_tile_p<2, LRV, true, UPD, 4, false>); q(k_dp_tile_p<2, LRV, true, UPD, 4, true>)
    QG(false, false); QG(false, true); QG(true, false); QG(true, true);
#undef QG
#define QW(DV) q(k_dp_tile_p<2, false, true, false, DV, false>); q(k_dp_tile_p<2, false, true, true, DV, false>); q(k_dp_tile_p<2, true, true, false, DV, true>); q(k_dp_tile_p<2, true, true, true, DV, true>)
    QW(5); QW(6); QW(7); QW(8); QW(9); QW(10);         // delta_x 5 .. 10 (round 6): few staged rows, wide candidate scans
#undef QW
    g_dpp_max_wgs_general = std::max(0, per_cu - 1) * prop.multiProcessorCount;
    {
        int per_cu = 1 << 20;
        auto ql = [&](auto kern) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 128, 0) != hipSuccess) { (void) hipGetLastError(); n = 0; }
            per_cu = std::min(per_cu, n);
        };
#define QL(LRV) ql(k_band_levels<LRV, false, 1, false>); ql(k_band_levels<LRV, true, 1, false>); ql(k_band_levels<LRV, true, 1, true>); \
    ql(k_band_levels<LRV, false, 2, false>); ql(k_band_levels<LRV, true, 2, false>); ql(k_band_levels<LRV, true, 2, true>); \
    ql(k_band_levels<LRV, false, 3, false>); ql(k_band_levels<LRV, true, 3, false>); ql(k_band_levels<LRV, true, 3, true>); \
    ql(k_band_levels<LRV, false, 4, false>); ql(k_band_levels<LRV, true, 4, false>); ql(k_band_levels<LRV, true, 4, true>)
        QL(false); QL(true);
#undef QL
#define QLW(DV) ql(k_band_levels<false, true, DV, false>); ql(k_band_levels<true, true, DV, false>); ql(k_band_levels<false, true, DV, true>); ql(k_band_levels<true, true, DV, true>)
        QLW(5); QLW(6); QLW(7); QLW(8); QLW(9); QLW(10);
#undef QLW
        g_dpp_max_wgs_levels = std::max(0, per_cu - 1) * prop.multiProcessorCount;
    }
    return g_dpp_max_wgs_plain;
}

// Large lock-step groups are carved on 4 streams, and those need hardware queues of their own: the HIP runtime's
// GPU_MAX_HW_QUEUES, default 4 per process, read ONCE when the runtime initialises (lqrhip_sub_batches below).  A host
// that has never heard of the variable (the plug-in) would silently get one stream and 10 % less.  So when this library
// is loaded into a process that has not brought the GPU runtime up yet -- no descriptor of /dev/kfd is open -- and the
// variable is not set, it is set to 8 here, before the library's own first HIP call initialises the runtime.  A host that
// set it (to anything) keeps its value; a host whose runtime is already up keeps one stream.
static bool kfd_is_open(void)
{
    DIR *d = opendir("/proc/self/fd");
    if (!d) return true;                      // cannot tell: leave the environment alone
    bool open_ = false;
    while (struct dirent *e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        char path[64], link[64];
        snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(path, link, sizeof link - 1);
        if (n <= 0) continue;
        link[n] = 0;
        if (strcmp(link, "/dev/kfd") == 0) { open_ = true; break; }
    }
    closedir(d);
    return open_;
}
__attribute__((constructor)) static void lqrhip_on_load(void)
{
    if (getenv("GPU_MAX_HW_QUEUES") || kfd_is_open()) return;
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

extern "C" int lqrhip_init(void)
{
    if (g_device >= 0) return g_device;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_err = "no HIP device visible: the MI355X engine needs a gfx950 GPU (there is no CPU fallback)";
        (void) hipGetLastError();
        return LQRHIP_EHIP;
    }
    int dev = 0;
    const char *lr = getenv("LOCAL_RANK");
    if (lr) dev = atoi(lr) % n;
    HIPCK(hipSetDevice(dev));
    HIPCK(hipStreamCreateWithFlags(&g_stream0, hipStreamNonBlocking));
    HIPCK(hipHostMalloc((void **) &g_dev_err_host, sizeof(int), hipHostMallocMapped));
    *g_dev_err_host = 0;
    HIPCK(hipHostGetDevicePointer((void **) &g_dev_err, g_dev_err_host, 0));
    g_dpp_max_wgs = dpp_resident_workgroups(dev);
    g_device = dev;
    return dev;
}

// A kernel recorded a failure (dev_fail): report it once, as LQRHIP_EFAULT, and clear the word.  The word is one per
// process and whoever synchronises first finds it -- not necessarily the batch whose kernel failed (the host rolls back and
// redoes the session of EVERY sub-batch of the group).  A persistent sweep that gave up half way leaves its batch's exchange
// area (tags, finished-tile counter) and, for an update, the plane pointers in the device descriptors in an unknown state,
// so EVERY live batch is marked for a fresh lay-out of both.
static std::vector<LqrHipBatch *> g_live_batches;
static void invalidate_all_batches(void);
// [0] spin time-outs, [1] failed activity predictions, [2] seam-log self-check failures, [3] level self-check failures,
// [4] sessions rolled back, [5] faults injected (lqrhip_debug_inject), [6] sessions carved on the non-spinning kernels
static unsigned long long g_fault_stats[8];
// After a spin time-out the persistent (co-residency-dependent) kernels are not chosen again in this process: a device that is
// shared or partitioned now will be in a minute, and every further resize would first wait out the time-out (~0.3 s) and then be
// redone.  lqrhip_set_no_spin(0) re-arms them (tests).
extern "C" void lqrhip_set_no_spin(int on) { g_no_spin = on != 0; }
extern "C" int lqrhip_get_no_spin(void) { return g_no_spin ? 1 : 0; }
static int check_dev_error(void)
{
    if (!g_dev_err_host || *g_dev_err_host == 0) return 0;
    const int code = *g_dev_err_host;
    *g_dev_err_host = 0;
    invalidate_all_batches();
    switch (code) {
    case DEVERR_TILE_TIMEOUT:
        g_fault_stats[0]++;
        if (!g_no_spin) fprintf(stderr, "liblqr-hip: a persistent kernel's workgroups were not co-resident in time (GPU shared or partitioned?): "
                                        "this process now uses the non-spinning kernels\n");
        g_no_spin = true;
        g_err = "persistent tiled DP sweep: a neighbour tile never became resident (GPU shared or partitioned?); results of this session are invalid";
        break;
    case DEVERR_BAND_PREDICTION: g_fault_stats[1]++; g_err = "band update: activity prediction failed; results of this session are invalid"; break;
    case DEVERR_SEAMLOG: g_fault_stats[2]++; g_err = "self-check: the session's seam log does not describe delta_x-connected seams inside the frame; results of this session are invalid"; break;
    case DEVERR_LEVELS: g_fault_stats[3]++; g_err = "self-check: a level of the session is missing or occurs twice in a row of the visibility map; results of this session are invalid"; break;
    default: g_err = "device-side failure " + std::to_string(code) + "; results of this session are invalid"; break;
    }
    return LQRHIP_EFAULT;
}
extern "C" int lqrhip_fault_stats(unsigned long long *out8, int reset)
{
    memcpy(out8, g_fault_stats, sizeof g_fault_stats);
    if (reset) memset(g_fault_stats, 0, sizeof g_fault_stats);
    return 0;
}

// Device allocations go through a small size-class cache: the carve path allocates and frees
// image-sized planes for every inflate / flatten / read-out, and hipMalloc / hipFree (which
// synchronises the device) would otherwise cost more than the kernels between them.  A block is
// only returned to the cache after the stream that used it has been synchronised.
static std::multimap<size_t, void *> g_pool_free;
static std::map<void *, size_t> g_pool_size;
static size_t g_pool_cached = 0;
static const size_t POOL_MAX_CACHED = (size_t) 24 << 30;

// LQRHIP_POISON=<byte> in the environment (debugging aid, scripts/fuzz_parity.py): every block handed out is first filled
// with that byte and the device synchronised, so that a kernel that reads memory nothing wrote yet fails the same way every
// time instead of depending on what the block held before
// LQRHIP_POISON=r1 / r2 / r3: pseudo-random words that look like what a recycled block holds -- floats in [0, 100),
// integers in [0, 2048), arbitrary bits
__global__ void k_poison_random(unsigned *p, size_t n, int mode, int stride, int c0, int c1, int r0, int r1)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        if (stride > 0) {       // LQRHIP_POISON_WINDOW=stride:c0:c1:r0:r1 -- only that window of a plane, zero elsewhere
            const int col = (int) (i % stride), row = (int) (i / stride);
            if (col < c0 || col >= c1 || row < r0 || row >= r1) { p[i] = 0; continue; }
        }
        unsigned hsh = (unsigned) i * 2654435761u + 0x9e3779b9u;
        hsh ^= hsh >> 15; hsh *= 0x85ebca6bu; hsh ^= hsh >> 13; hsh *= 0xc2b2ae35u; hsh ^= hsh >> 16;
        p[i] = mode == 1 ? __float_as_uint((float) (hsh >> 8) * (100.0f / 16777216.0f)) : mode == 2 ? (hsh >> 21) : hsh;
    }
}
static const char *g_alloc_name = "";      // what the allocation is for (LQRHIP_POISON_LOG)
static int g_poison = -2;
static unsigned g_poison32 = 0;
static int pool_poison(void *p, size_t sz)
{
    if (g_poison == -2) {
        const char *e = getenv("LQRHIP_POISON");
        g_poison = -1;
        if (e && e[0] == '0' && e[1] == 'x') { g_poison = 256; g_poison32 = (unsigned) strtoul(e, nullptr, 16); }     // a 32-bit word
        else if (e && e[0] == 'r') g_poison = 256 + atoi(e + 1);
        else if (e && *e) g_poison = atoi(e) & 255;
    }
    if (g_poison < 0) return 0;
    {   // LQRHIP_POISON_RANGE=a:b poisons only the allocations numbered a .. b-1 of the process (LQRHIP_POISON_LOG lists them)
        static long seq = 0, lo = 0, hi = -1;
        static int logit = -1;
        if (logit < 0) {
            logit = getenv("LQRHIP_POISON_LOG") != nullptr;
            const char *r = getenv("LQRHIP_POISON_RANGE");
            if (r) sscanf(r, "%ld:%ld", &lo, &hi);
        }
        const long me = seq++;
        if (logit) fprintf(stderr, "alloc %ld %zu %s\n", me, sz, g_alloc_name);
        if (hi >= 0 && (me < lo || me >= hi)) { HIPCK(hipMemset(p, 0, sz)); HIPCK(hipDeviceSynchronize()); return 0; }
    }
    if (g_poison > 256) {
        static int win[5] = {0, 0, 0, 0, 0};
        static bool once = false;
        if (!once) { once = true; const char *wv = getenv("LQRHIP_POISON_WINDOW"); if (wv) sscanf(wv, "%d:%d:%d:%d:%d", win, win + 1, win + 2, win + 3, win + 4); }
        hipLaunchKernelGGL(k_poison_random, dim3(1024), dim3(256), 0, 0, (unsigned *) p, sz / 4, g_poison - 256, win[0], win[1], win[2], win[3], win[4]);
    }
    else if (g_poison == 256) HIPCK(hipMemsetD32((hipDeviceptr_t) p, (int) g_poison32, sz / 4));
    else HIPCK(hipMemset(p, g_poison, sz));
    HIPCK(hipDeviceSynchronize());
    return 0;
}

// fault injection (tests/test_faults_gpu.py): the nth device allocation from now on fails with "out of memory", once (-1: disarmed)
static long g_fail_alloc_in = -1;
extern "C" void lqrhip_debug_fail_alloc(int nth) { g_fail_alloc_in = nth; }
static int pool_alloc(void **p, size_t bytes)
{
    const size_t sz = (bytes + ((size_t) 1 << 20) - 1) & ~(((size_t) 1 << 20) - 1);      // 1 MiB classes
    if (g_fail_alloc_in >= 0 && g_fail_alloc_in-- == 0) { *p = nullptr; g_err = std::string("injected allocation failure (") + (g_alloc_name ? g_alloc_name : "?") + ")"; return LQRHIP_ENOMEM; }
    auto it = g_pool_free.find(sz);
    if (it != g_pool_free.end()) {
        *p = it->second;
        g_pool_free.erase(it);
        g_pool_cached -= sz;
        return pool_poison(*p, sz);
    }
    hipError_t e = hipMalloc(p, sz);
    if (e == hipErrorOutOfMemory && !g_pool_free.empty()) {       // give the cache back and retry once
        (void) hipGetLastError();
        for (auto &kv : g_pool_free) { (void) hipFree(kv.second); g_pool_size.erase(kv.second); }
        g_pool_free.clear();
        g_pool_cached = 0;
        e = hipMalloc(p, sz);
    }
    HIPCK(e);
    g_pool_size[*p] = sz;
    return pool_poison(*p, sz);
}
static void pool_free(void *p)
{
    auto it = g_pool_size.find(p);
    if (it == g_pool_size.end()) { (void) hipFree(p); return; }
    if (g_pool_cached + it->second > POOL_MAX_CACHED) {
        (void) hipFree(p);
        g_pool_size.erase(it);
        return;
    }
    g_pool_free.emplace(it->second, p);
    g_pool_cached += it->second;
}

template <typename T>
static int dmalloc_(T **p, size_t n, const char *name)
{
    *p = nullptr;
    g_alloc_name = name;
    return pool_alloc((void **) p, (n ? n : 1) * sizeof(T));
}
#define dmalloc(p, n) dmalloc_((p), (n), #p)
template <typename T>
static void dfree(T *&p)
{
    if (p) pool_free((void *) p);
    p = nullptr;
}

// zero device memory and wait: hipMemset on the null stream is asynchronous for device memory and the
// engine's streams are non-blocking, so a null-stream memset is not ordered with the kernels after it
static hipError_t dzero(void *p, size_t bytes)
{
    hipError_t e = hipMemsetAsync(p, 0, bytes, g_stream0);
    return e != hipSuccess ? e : hipStreamSynchronize(g_stream0);
}

// Host <-> device copies of whole images.  The plug-in hands over and takes back PAGEABLE memory (a g_malloc'ed buffer at
// lqr_carver_new, render.c:222; the scan-line buffer at read-out, io_functions.c:155-164); hipMemcpy on pageable memory
// runs at ~1-2 GB/s here (it pins and unpins as it goes).  These go through a ring of pinned bounce buffers instead, and
// the CPU side of the bounce -- memcpy between the caller's pageable buffer and the ring, page faults of a freshly
// allocated destination included -- is done by a few helper threads in parallel while the DMA engine moves other chunks
// (round 3: one thread, two 8 MB buffers: 8.6 GB/s up, 4.2 GB/s down; a single core's memcpy and its page faults were
// the limit, not PCIe).  The helpers touch host memory only; every HIP call stays on the caller's thread.
// Both functions return with the transfer complete, also on error (the stream is drained before they return).
static const size_t STAGE_BYTES = (size_t) 4 << 20;
static const int STAGE_SLOTS = 16;
static uint8_t *g_stage[STAGE_SLOTS];
static hipEvent_t g_stage_ev[STAGE_SLOTS];
static bool g_stage_ready = false;

struct CopyPool {
    struct Job { void *dst; const void *src; size_t n; };
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<std::pair<int, Job>> q;      // (ticket, job)
    std::vector<char> done;                 // per ticket of the current transfer
    size_t pending = 0;                     // submitted and not finished
    bool stop = false;
    void start(int n)
    {
        for (int i = 0; i < n; i++) th.emplace_back([this] { run(); });
    }
    void run()
    {
        for (;;) {
            std::pair<int, Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [this] { return stop || !q.empty(); });
                if (stop && q.empty()) return;
                j = q.front(); q.pop_front();
            }
            memcpy(j.second.dst, j.second.src, j.second.n);
            {
                std::lock_guard<std::mutex> lk(mu);
                done[j.first] = 1;
                pending--;
            }
            cv_done.notify_all();
        }
    }
    void begin(size_t tickets) { std::lock_guard<std::mutex> lk(mu); done.assign(tickets, 0); }
    void submit(int ticket, void *dst, const void *src, size_t n)
    {
        { std::lock_guard<std::mutex> lk(mu); q.emplace_back(ticket, Job{dst, src, n}); pending++; }
        cv_job.notify_one();
    }
    void drain()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [this] { return pending == 0; });
    }
    void wait(int ticket)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done[ticket] != 0; });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_job.notify_all();
        for (auto &t : th) t.join();
    }
};
static CopyPool *g_copy_pool = nullptr;

static int stage_init(void)
{
    if (g_stage_ready) return 0;
    // all or nothing: a partial set of buffers / events is released again, so a later call starts over
    int made = 0;
    hipError_t e = hipSuccess;
    for (; made < STAGE_SLOTS; made++) {
        g_stage[made] = nullptr; g_stage_ev[made] = nullptr;
        if ((e = hipHostMalloc((void **) &g_stage[made], STAGE_BYTES, hipHostMallocDefault)) != hipSuccess) break;
        if ((e = hipEventCreateWithFlags(&g_stage_ev[made], hipEventDisableTiming)) != hipSuccess) { (void) hipHostFree(g_stage[made]); break; }
    }
    if (made < STAGE_SLOTS) {
        for (int i = 0; i < made; i++) { (void) hipHostFree(g_stage[i]); (void) hipEventDestroy(g_stage_ev[i]); g_stage[i] = nullptr; }
        g_err = std::string("staging buffers: ") + hipGetErrorString(e);
        (void) hipGetLastError();
        return e == hipErrorOutOfMemory ? LQRHIP_ENOMEM : LQRHIP_EHIP;
    }
    if (!g_copy_pool) {
        unsigned hw = std::thread::hardware_concurrency();
        g_copy_pool = new CopyPool();
        g_copy_pool->start(hw >= 16 ? 8 : hw >= 4 ? (int) hw / 2 : 1);
    }
    g_stage_ready = true;
    return 0;
}
static int h2d_staged(void *dst, const void *src, size_t bytes)
{
    int rc = stage_init();
    if (rc) return rc;
    const size_t nchunk = (bytes + STAGE_BYTES - 1) / STAGE_BYTES;
    CopyPool &cp = *g_copy_pool;
    cp.begin(nchunk);
    auto body = [&]() -> int {
        size_t submitted = 0;
        for (size_t i = 0; i < nchunk; i++) {
            // keep the helpers a ring ahead: chunk j goes into slot j % STAGE_SLOTS once the DMA that last read it is done
            for (; submitted < nchunk && submitted < i + STAGE_SLOTS; submitted++) {
                const int slot = (int) (submitted % STAGE_SLOTS);
                HIPCK(hipEventSynchronize(g_stage_ev[slot]));
                const size_t off = submitted * STAGE_BYTES;
                cp.submit((int) submitted, g_stage[slot], (const uint8_t *) src + off, std::min(STAGE_BYTES, bytes - off));
            }
            const int slot = (int) (i % STAGE_SLOTS);
            const size_t off = i * STAGE_BYTES;
            cp.wait((int) i);
            HIPCK(hipMemcpyAsync((uint8_t *) dst + off, g_stage[slot], std::min(STAGE_BYTES, bytes - off), hipMemcpyHostToDevice, g_stream0));
            HIPCK(hipEventRecord(g_stage_ev[slot], g_stream0));
        }
        return 0;
    };
    rc = body();
    cp.drain();         // whatever happened, nobody touches the caller's buffer or the ring after we return
    hipError_t e = hipStreamSynchronize(g_stream0);
    if (!rc && e != hipSuccess) { g_err = std::string("upload: ") + hipGetErrorString(e); rc = LQRHIP_EHIP; }
    return rc;
}
static int d2h_staged(void *dst, const void *src, size_t bytes)
{
    int rc = stage_init();
    if (rc) return rc;
    const size_t nchunk = (bytes + STAGE_BYTES - 1) / STAGE_BYTES;
    CopyPool &cp = *g_copy_pool;
    cp.begin(nchunk);
    size_t copied_out = 0;      // chunks handed to the helpers
    auto hand_over = [&](size_t j) -> int {
        const int slot = (int) (j % STAGE_SLOTS);
        const size_t off = j * STAGE_BYTES;
        HIPCK(hipEventSynchronize(g_stage_ev[slot]));
        cp.submit((int) j, (uint8_t *) dst + off, g_stage[slot], std::min(STAGE_BYTES, bytes - off));
        return 0;
    };
    auto body = [&]() -> int {
        for (size_t i = 0; i < nchunk; i++) {
            if (i >= (size_t) STAGE_SLOTS) {              // slot reuse: the helper must have emptied it
                for (; copied_out <= i - STAGE_SLOTS; copied_out++) { int r = hand_over(copied_out); if (r) return r; }
                cp.wait((int) (i - STAGE_SLOTS));
            }
            const int slot = (int) (i % STAGE_SLOTS);
            const size_t off = i * STAGE_BYTES;
            HIPCK(hipMemcpyAsync(g_stage[slot], (const uint8_t *) src + off, std::min(STAGE_BYTES, bytes - off), hipMemcpyDeviceToHost, g_stream0));
            HIPCK(hipEventRecord(g_stage_ev[slot], g_stream0));
            // hand over whatever has landed already, without waiting for it
            while (copied_out < i && hipEventQuery(g_stage_ev[copied_out % STAGE_SLOTS]) == hipSuccess) { int r = hand_over(copied_out); if (r) return r; copied_out++; }
        }
        for (; copied_out < nchunk; copied_out++) { int r = hand_over(copied_out); if (r) return r; }
        return 0;
    };
    rc = body();
    cp.drain();         // the helpers are done with the caller's buffer
    if (rc) (void) hipStreamSynchronize(g_stream0);
    return rc;
}

static int batch_sync_of(LqrHipCarver *c)
{
    LqrHipCarver *r = c->root ? c->root : c;
    if (r->batch) HIPCK(hipStreamSynchronize(r->batch->stream));
    return 0;
}

extern "C" LqrHipCarver *lqrhip_carver_create(const unsigned char *rgb, int w, int h, int channels)
{
    if (lqrhip_init() < 0) return nullptr;
    LqrHipCarver *c = new LqrHipCarver();
    c->ch = channels; c->w0 = w; c->h0 = h;
    size_t n = (size_t) w * h;
    if (dmalloc(&c->rgb0, n * channels) || dmalloc(&c->vs, n)) { lqrhip_carver_destroy(c); return nullptr; }
    // the visibility map is cleared on the same stream, under the upload: one synchronisation for both
    if (hipMemsetAsync(c->vs, 0, n * sizeof(int32_t), g_stream0) != hipSuccess || h2d_staged(c->rgb0, rgb, n * channels) != 0) {
        g_err = "upload failed";
        lqrhip_carver_destroy(c);
        return nullptr;
    }
    return c;
}

static void free_working(LqrHipCarver *c)
{
    dfree(c->pix); dfree(c->en); dfree(c->m); dfree(c->least); dfree(c->m2); dfree(c->least2); dfree(c->bias); dfree(c->rig);
    dfree(c->seam_x); dfree(c->seam_log); dfree(c->flags);
    dfree(c->vp_map); dfree(c->vp_path);
    c->log_cap = 0; c->vp_cap = 0;
}

extern "C" void lqrhip_carver_destroy(LqrHipCarver *c)
{
    if (!c) return;
    // its planes may still be in use by kernels on the owning batch's stream or by the shim's own stream (resets, mask
    // uploads, read-outs): wait for those two, not for the device (tearing a batch down was 64 device synchronisations)
    {
        LqrHipCarver *r = c->root ? c->root : c;
        if (r->batch && r->batch->stream) (void) hipStreamSynchronize(r->batch->stream);
        if (c->batch && c->batch != r->batch && c->batch->stream) (void) hipStreamSynchronize(c->batch->stream);
        if (g_stream0) (void) hipStreamSynchronize(g_stream0);
        (void) hipGetLastError();
    }
    dfree(c->rgb0);
    if (!c->root) dfree(c->vs);
    dfree(c->bias0); dfree(c->rig0);
    free_working(c);
    delete c;
}

extern "C" int lqrhip_carver_attach(LqrHipCarver *root, LqrHipCarver *aux)
{
    if (root->w0 != aux->w0 || root->h0 != aux->h0) return LQRHIP_EARG;
    dfree(aux->vs);
    aux->vs = root->vs;
    aux->root = root;
    root->aux.push_back(aux);
    if (root->batch) root->batch->dirty = true;
    return 0;
}

// (re)allocate the working planes for a w x h carved frame
static int ensure_working(LqrHipCarver *c, int w, int h)
{
    int stride = ((w + 16) + 63) & ~63;
    bool need_bias = c->bias0 != nullptr, need_rig = c->rig0 != nullptr;
    if (c->pix && c->stride == stride && c->wk_h == h && (!!c->bias == need_bias) && (!!c->rig == need_rig)) return 0;
    free_working(c);
    c->stride = 0; c->wk_h = 0;
    size_t n = (size_t) stride * (h + 1) + 1024;
    int rc;
    if ((rc = dmalloc(&c->pix, n)) || (rc = dmalloc(&c->en, n)) || (rc = dmalloc(&c->m, n)) || (rc = dmalloc(&c->least, n)) ||
        (rc = dmalloc(&c->seam_x, (size_t) h + 8)) || (rc = dmalloc(&c->flags, (size_t) FLAG_WORDS)) ||
        (need_bias && (rc = dmalloc(&c->bias, n))) || (need_rig && (rc = dmalloc(&c->rig, n)))) {
        free_working(c);            // never leave a half-allocated set behind: a retry must not pass the early-out above
        return rc;
    }
    // all on the shim's stream, one synchronisation
    hipError_t e = hipMemsetAsync(c->least, 0, n, g_stream0);
    if (e == hipSuccess) e = hipMemsetAsync(c->m, 0, n * sizeof(float), g_stream0);
    if (e == hipSuccess) e = hipMemsetAsync(c->en, 0, n * sizeof(float), g_stream0);
    if (e == hipSuccess) e = hipMemsetAsync(c->pix, 0, n * sizeof(uint32_t), g_stream0);
    if (e == hipSuccess) e = hipMemsetAsync(c->flags, 0, (size_t) FLAG_WORDS * sizeof(int32_t), g_stream0);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream0);
    if (e != hipSuccess) { free_working(c); HIPCK(e); }
    c->stride = stride; c->wk_h = h;
    if (c->batch) c->batch->dirty = true;
    return 0;
}

static int ensure_log(LqrHipCarver *c, int n_seams, int h)
{
    if (c->seam_log && c->log_cap >= n_seams && c->log_h == h) return 0;
    dfree(c->seam_log);
    int rc = dmalloc(&c->seam_log, (size_t) n_seams * h);
    if (rc) return rc;
    c->log_cap = n_seams; c->log_h = h;
    if (c->batch) c->batch->dirty = true;
    return 0;
}

extern "C" int lqrhip_carver_activate(LqrHipCarver *c)
{
    c->active = 1;
    // E1 lqr_carver_init allocates the DP maps: do the same here, so that the first resize does
    // not pay for hipMalloc (re-done lazily by lqrhip_wk_init if the geometry changes)
    return ensure_working(c, c->w0, c->h0);
}

extern "C" int lqrhip_mask_add(LqrHipCarver *c, const unsigned char *mask, int channels, int width, int height, int x_off,
                               int y_off, int transposed, int is_rigmask, int bias_factor)
{
    int rc = batch_sync_of(c);
    if (rc) return rc;
    size_t n = (size_t) c->w0 * c->h0;
    float **plane = is_rigmask ? &c->rig0 : &c->bias0;
    if (!*plane) {
        if ((rc = dmalloc(plane, n))) return rc;
        HIPCK(dzero(*plane, n * sizeof(float)));
        if (c->batch) c->batch->dirty = true;
    }
    int wt = transposed ? c->h0 : c->w0, ht = transposed ? c->w0 : c->h0;
    int x0 = x_off < 0 ? x_off : 0, y0 = y_off < 0 ? y_off : 0;
    int x1 = x_off > 0 ? x_off : 0, y1 = y_off > 0 ? y_off : 0;
    int x2 = wt < width + x_off ? wt : width + x_off, y2 = ht < height + y_off ? ht : height + y_off;
    int nx = x2 - x1, ny = y2 - y1;
    if (nx <= 0 || ny <= 0) return 0;
    uint8_t *dmask = nullptr;
    size_t mbytes = (size_t) width * height * channels;
    if ((rc = dmalloc(&dmask, mbytes))) return rc;
    auto run = [&]() -> int {
        int rcu = h2d_staged(dmask, mask, mbytes);
        if (rcu) return rcu;
        dim3 grid((nx + 255) / 256, ny);
        hipLaunchKernelGGL(k_mask_add, grid, dim3(256), 0, g_stream0, *plane, c->w0, dmask, channels, width, x0, y0, x1, y1, nx, ny,
                           transposed, is_rigmask, bias_factor);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(g_stream0));
        return 0;
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);
    dfree(dmask);
    return rc;
}

// ---- batch -----------------------------------------------------------------
// A lock-step group can be split over several HIP streams (sub-batches) that advance seam by seam side by side: a seam
// round is a latency-bound chain (backtrack, energy update, band update: ~0.7 ms at 4K whatever the batch size, on a
// few CUs) followed by the bandwidth-bound carve, so one sub-batch's chain runs under another's carve.  Measured at
// 64 x 4K with 4 streams: +12 % throughput (424k vs 377k Mseams*px/s), but every kernel then shares the chip -- a carve
// launch of 16 images takes 0.20 ms next to the others' kernels (2.6 TB/s algorithmic) instead of 0.13 ms alone -- and
// it needs a hardware queue per stream: with the HIP runtime's default of 4 queues per process (GPU_MAX_HW_QUEUES) the
// streams share queues and the same split is 30 % SLOWER.  lqrhip_set_sub_batches (bench.py --sub-batches) pins the
// number of streams; the default is automatic (lqrhip_sub_batches below).  DESIGN.md 4.11.
static int g_sub_batches = 0;           // 0: automatic (below)
extern "C" void lqrhip_set_sub_batches(int n) { g_sub_batches = n > 0 ? n : 0; }
// Streams a lock-step group of n carvers is split over.  Automatic: 4 for groups of 49 and more, 2 for 32 to 48 (round 5) WHEN the process has the
// hardware queues for them -- the HIP runtime's GPU_MAX_HW_QUEUES (default 4, read when HIP initialises, shared with every
// other stream of the process) must be 8 or more; with fewer, streams share queues and the split is 30 % slower than
// one stream, so it is not made.
extern "C" int lqrhip_sub_batches(int n)
{
    int nb = g_sub_batches;
    if (nb == 0) {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        // 4 streams for the groups that run k_band_update_tw (49 images and more), 2 for the groups of 32 to 48 that run k_band_levels
        // (round 5, one box, Mseams*px/s with 2 / 4 streams: 32 images 360 / 352 k, 48 images 432 / 404 k)
        nb = (q && atoi(q) >= 8) ? (n >= 49 ? 4 : n >= 32 ? 2 : 1) : 1;
    }
    return n >= 2 * nb ? nb : 1;
}

extern "C" void lqrhip_batch_set_shared(LqrHipBatch *b, int shared) { b->shared = shared != 0; b->shared_n = shared > 1 ? shared : 1; }
extern "C" void lqrhip_batch_set_safe(LqrHipBatch *b, int safe) { b->safe = safe != 0; if (safe) g_fault_stats[6]++; }

extern "C" LqrHipBatch *lqrhip_batch_create(LqrHipCarver **carvers, int n)
{
    if (lqrhip_init() < 0 || n <= 0) return nullptr;
    LqrHipBatch *b = new LqrHipBatch();
    for (int i = 0; i < n; i++) {
        b->cs.push_back(carvers[i]);
        carvers[i]->batch = b;
    }
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **) &b->d_desc, sizeof(DevCarver) * n) != hipSuccess) {
        g_err = "batch_create failed";
        if (b->stream) (void) hipStreamDestroy(b->stream);
        for (auto *c : b->cs) if (c->batch == b) c->batch = nullptr;
        delete b;
        return nullptr;
    }
    g_live_batches.push_back(b);
    return b;
}

static void invalidate_all_batches(void)
{
    for (auto *b : g_live_batches) { b->exch_ntiles = 0; b->dirty = true; }
}

// after a failed resize: drain the stream, drop whatever the kernels recorded (it belongs to the failed call, the next
// resize must not report it), lay everything out afresh
extern "C" void lqrhip_batch_abort(LqrHipBatch *b)
{
    if (!b) return;
    (void) hipStreamSynchronize(b->stream);
    (void) hipGetLastError();
    discard_pending(b);
    if (g_dev_err_host) *g_dev_err_host = 0;
    invalidate_all_batches();
}

extern "C" void lqrhip_batch_destroy(LqrHipBatch *b)
{
    if (!b) return;
    g_live_batches.erase(std::remove(g_live_batches.begin(), g_live_batches.end(), b), g_live_batches.end());
    if (b->stream) { (void) hipStreamSynchronize(b->stream); (void) hipStreamDestroy(b->stream); }
    discard_pending(b);
    dfree(b->exch);
    for (auto *c : b->cs) if (c->batch == b) c->batch = nullptr;
    if (b->d_desc) (void) hipFree(b->d_desc);
    delete b;
}

extern "C" int lqrhip_batch_sync(LqrHipBatch *b)
{
    HIPCK(hipStreamSynchronize(b->stream));
    return check_dev_error();
}
extern "C" void *lqrhip_batch_stream(LqrHipBatch *b) { return (void *) b->stream; }

static DevCarver make_desc(const LqrHipCarver *c)
{
    DevCarver d;
    d.rgb0 = c->rgb0; d.vs = c->vs; d.bias0 = c->bias0; d.rig0 = c->rig0;
    d.pix = c->pix; d.en = c->en; d.m = c->m; d.least = c->least; d.m2 = c->m2; d.least2 = c->least2; d.bias = c->bias; d.rig = c->rig;
    d.seam_x = c->seam_x; d.seam_log = c->seam_log; d.flags = c->flags;
    d.vp_map = c->vp_map; d.vp_path = c->vp_path;
    return d;
}

static int batch_upload(LqrHipBatch *b)
{
    if (!b->dirty) return 0;
    std::vector<DevCarver> h;
    for (auto *c : b->cs) h.push_back(make_desc(c));
    HIPCK(hipStreamSynchronize(b->stream));
    HIPCK(hipMemcpy(b->d_desc, h.data(), sizeof(DevCarver) * h.size(), hipMemcpyHostToDevice));
    b->dirty = false;
    return 0;
}

static DpK make_dpk(const LqrHipDpParams *p, int ch)
{
    DpK k;
    k.delta = p->delta_x; k.use_rig = p->use_rigidity;
    memcpy(k.rigmap, p->rigidity_map, sizeof k.rigmap);
    k.nrg = p->nrg_func; k.radius = p->nrg_radius; k.w_start = p->w_start; k.ch = ch;
    return k;
}

struct ProfScope {
    ProfRec *rec = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t s;
    ProfScope(const char *name, hipStream_t stream, double bytes) : s(stream)
    {
        // every timed scope costs ~10 us of queue time (two event packets): mode 2 keeps that to the one
        // kernel whose launch time the bench line needs
        if (!g_prof || (g_prof == 2 && strcmp(name, "carve") != 0)) return;
        rec = &g_profrec[name];
        rec->bytes += bytes;
        (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
        (void) hipEventRecord(e0, s);
    }
    ~ProfScope()
    {
        if (!rec) return;
        (void) hipEventRecord(e1, s);
        rec->ev.emplace_back(e0, e1);
    }
};

extern "C" void lqrhip_prof_enable(int on) { g_prof = on; }
static int g_vpath_mode = -1;            // -1: the parallel backtrack for groups up to g_vpath_par_max images of at least g_vpath_min_rows rows; 0: never; 1: always (delta_x 1 .. 4)
static int g_vpath_par_max = 3, g_vpath_min_rows = 1000;     // (3 x 4K: 58 -> 46 us per seam; 4: equal; 8: slower -- the maps of n images are n times the work)
extern "C" void lqrhip_set_vpath_mode(int mode, int par_max) { g_vpath_mode = mode; if (par_max > 0) g_vpath_par_max = par_max; }
static int g_sweep_threads = 256;        // threads of the k_dp_sweep<UPDATE> launch behind the band kernels (256, or 1024 as in rounds 1 - 5)
extern "C" void lqrhip_set_sweep_threads(int n) { g_sweep_threads = n == 256 ? 256 : DP_THREADS; }
static int g_carve_fused = 4;            // k_carve_e (carve + energy update in one launch) for groups up to this many images (0: the two kernels always)
extern "C" void lqrhip_set_carve_fused(int max_images) { g_carve_fused = max_images == 1 ? 4 : max_images < 0 ? 0 : max_images; }
static int g_update_mode = -1;
static int g_band_levels = -1;           // k_band_levels: -1 automatic; 0 never; n: n slots per image (lqrhip_set_band_levels)
// -1: by batch size (g_tiled_update_px); 0: band kernel (k_band_update_tw); 1: tiled full-width update whenever its
// grid fits; 2: the per-row-barrier band kernel (k_band_update_mw); 3: the generic one-wave band kernel + sweep
// (what delta_x > 2 runs on), whatever the parameters
extern "C" void lqrhip_set_update_mode(int mode) { g_update_mode = mode; }
static int g_dpp_limit_override = -1;
static int g_dpp_px_override = 0;       // test hook: 2 or 4 pins the persistent sweep's pixels per lane (0 = by batch size)
extern "C" void lqrhip_set_dp_persistent_px(int px) { g_dpp_px_override = (px == 2 || px == 3 || px == 4) ? px : 0; }
// -1: the occupancy-derived bound (dpp_resident_workgroups); >= 0: at most that many workgroups for the persistent
// tiled sweep -- 0 sends every full DP to k_dp_tile and every incremental update to a band kernel
extern "C" void lqrhip_set_dp_persistent_limit(int workgroups) { g_dpp_limit_override = workgroups; }
extern "C" void lqrhip_prof_reset(void)
{
    for (auto &kv : g_profrec) for (auto &e : kv.second.ev) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
    g_profrec.clear();
}
extern "C" int lqrhip_prof_get(const char *kernel, double *ms_total, long long *launches, double *bytes_total)
{
    auto it = g_profrec.find(kernel);
    *ms_total = 0; *launches = 0; *bytes_total = 0;
    if (it == g_profrec.end()) return 0;
    (void) hipDeviceSynchronize();
    for (auto &e : it->second.ev) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) *ms_total += ms;
    }
    *launches = (long long) it->second.ev.size();
    *bytes_total = it->second.bytes;
    return 0;
}

// Time during which at least one launch of `kernel` was running: with sub-batch streams launches overlap each other and
// other kernels, and bytes / (sum of launch times) would count the overlapped time twice.  Event times are taken relative
// to the first recorded event of the kernel (the GPU's clock is common to all streams).
extern "C" int lqrhip_prof_get_union(const char *kernel, double *ms_union)
{
    *ms_union = 0;
    auto it = g_profrec.find(kernel);
    if (it == g_profrec.end() || it->second.ev.empty()) return 0;
    (void) hipDeviceSynchronize();
    const hipEvent_t base = it->second.ev[0].first;
    std::vector<std::pair<float, float>> iv;
    for (auto &e : it->second.ev) {
        float a = 0, b = 0;
        // an event recorded before `base` on another stream gives a negative time: both orders are tried
        if (hipEventElapsedTime(&a, base, e.first) != hipSuccess) { (void) hipGetLastError(); float t = 0; if (hipEventElapsedTime(&t, e.first, base) == hipSuccess) a = -t; else (void) hipGetLastError(); }
        if (hipEventElapsedTime(&b, base, e.second) != hipSuccess) { (void) hipGetLastError(); float t = 0; if (hipEventElapsedTime(&t, e.second, base) == hipSuccess) b = -t; else (void) hipGetLastError(); }
        iv.emplace_back(a, b);
    }
    std::sort(iv.begin(), iv.end());
    float end = -1e30f;
    for (auto &p : iv) {
        if (p.first > end) { *ms_union += p.second - p.first; end = p.second; }
        else if (p.second > end) { *ms_union += p.second - end; end = p.second; }
    }
    return 0;
}

extern "C" int lqrhip_wk_init(LqrHipBatch *b, int from_visible)
{
    LqrHipCarver *c0 = b->cs[0];
    int w = c0->w0, h = c0->h0, rc;
    for (auto *c : b->cs) {
        if (c->w0 != w || c->h0 != h || c->ch != c0->ch) return LQRHIP_EARG;
        if ((rc = ensure_working(c, w, h))) return rc;
    }
    if ((rc = batch_upload(b))) return rc;
    for (auto *c : b->cs) c->frozen_epoch = 0;
    if (from_visible) {
        hipLaunchKernelGGL(k_wk_init_visible, dim3(h, (unsigned) b->cs.size()), dim3(256), 0, b->stream, b->d_desc, w, h, c0->stride, c0->ch);
    } else {
        dim3 grid((c0->stride + 255) / 256, h, (unsigned) b->cs.size());
        hipLaunchKernelGGL(k_wk_init, grid, dim3(256), 0, b->stream, b->d_desc, w, h, c0->stride, c0->ch);
    }
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int lqrhip_emap_build(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h)
{
    int rc;
    if ((rc = batch_upload(b))) return rc;
    LqrHipCarver *c0 = b->cs[0];
    dim3 grid((w + 255) / 256, h, (unsigned) b->cs.size());
    DpK k = make_dpk(p, c0->ch);
#define LAUNCH_EMAP(N) hipLaunchKernelGGL((k_emap_full<N>), grid, dim3(256), 0, b->stream, b->d_desc, k, w, h, c0->stride)
    NRG_DISPATCH(p->nrg_func, LAUNCH_EMAP)
#undef LAUNCH_EMAP
    HIPCK(hipGetLastError());
    return 0;
}


// E5 as H/32 dependent launches of one wave per 192-column tile (any batch size)
static int launch_dp_tiled(LqrHipBatch *b, const DpK &k, int w, int h, int lr)
{
    LqrHipCarver *c0 = b->cs[0];
    const dim3 grid((w + DPT_OWN - 1) / DPT_OWN, (unsigned) b->cs.size());
#define LAUNCH_TILE(LRV, RIGV) hipLaunchKernelGGL((k_dp_tile<LRV, RIGV>), grid, dim3(64), 0, b->stream, b->d_desc, k, w, h, c0->stride, y0)
    for (int y0 = 0; y0 < h; y0 += DPT_ROWS) {
        if (lr) { if (k.use_rig) LAUNCH_TILE(true, true); else LAUNCH_TILE(true, false); }
        else { if (k.use_rig) LAUNCH_TILE(false, true); else LAUNCH_TILE(false, false); }
    }
#undef LAUNCH_TILE
    HIPCK(hipGetLastError());
    return 0;
}

// can the persistent tiled sweep (k_dp_tile_p) take this batch?  Its tiles spin on each other, so the
// whole grid has to be resident at once.
// Pixels per lane of the persistent sweep for this batch: 2 while twice the tiles still fit the residency bound (the row
// chain is then ~33 instructions per wave instead of ~58; measured per 4K seam round, 2 vs 4 px per lane:
// 1 image 0.40 / 0.50 ms, 4: 0.45 / 0.55, 8: 0.58 / 0.62, 12: 0.76 / 0.77), else 4, 0 = not at all.
// `general`: delta_x = 2 and / or a rigidity mask (with rigidity): those instantiations exist for 2 px per lane only
static int dp_persistent_px(const LqrHipBatch *b, int w, bool general = false, int delta = 1, int count = -1)
{
    if (b->shared || no_spin(b)) return 0;
    const int bound = general ? g_dpp_max_wgs_general : g_dpp_max_wgs;
    const int limit = g_dpp_limit_override >= 0 ? std::min(g_dpp_limit_override, bound) : bound;
    const size_t n = count < 0 ? b->cs.size() : (size_t) count;
    const int hh = b->cs[0]->wk_h;                            // the block index is DPP_BLK_BITS bits of the granule tag
    const int maxblk = (1 << DPP_BLK_BITS) - 1;
    // px code 3 (round 6): 32-column tiles with 48-column halos, 48-row blocks -- a third fewer hand-overs through memory for twice the
    // tiles; while every tile still gets a compute unit of its own (a single 4K image: 120 tiles; measured: DESIGN.md 4.2)
    if (!general && delta == 1 && (g_dpp_px_override == 3 || g_dpp_px_override == 0) && hh <= maxblk * dpp_rb(3, 1) &&
        (size_t) ((w + dpp_own(3) - 1) / dpp_own(3)) * n <= (size_t) std::min(limit, g_dpp_px_override == 3 ? limit : g_n_cu)) return 3;
    if ((general || (g_dpp_px_override != 4 && g_dpp_px_override != 3)) && hh <= maxblk * dpp_rb(2, delta) && (size_t) ((w + dpp_own(2) - 1) / dpp_own(2)) * n <= (size_t) limit) return 2;
    const int limit4 = g_dpp_limit_override >= 0 ? std::min(g_dpp_limit_override, g_dpp_max_wgs_px4) : g_dpp_max_wgs_px4;
    if (!general && g_dpp_px_override != 2 && g_dpp_px_override != 3 && hh <= maxblk * dpp_halo(4) && (size_t) ((w + dpp_own(4) - 1) / dpp_own(4)) * n <= (size_t) limit4) return 4;
    return 0;
}
static bool dp_persistent_ok(const LqrHipBatch *b, int w) { return dp_persistent_px(b, w) != 0; }
// How many images of frame width `w` (the direction being carved) one lock-step batch may hold and still run delta_x = 2 /
// rigidity-mask carvers on the tiled kernels (k_dp_tile_p's general instantiations: one workgroup per 64 columns per image,
// all co-resident).  Beyond it such a batch would fall to the one-wave-per-image band kernel (~30x slower), so the host
// carves larger batches of such carvers group after group (host/lqr_carver.c, lqrx_carver_resize_batch).  0: no bound known.
extern "C" int lqrhip_general_batch_limit_delta(int w, int delta);
extern "C" int lqrhip_general_batch_limit(int w) { return lqrhip_general_batch_limit_delta(w, 2); }
extern "C" int lqrhip_general_batch_limit_delta(int w, int delta)
{
    if (lqrhip_init() < 0 || w < 1) return 0;
    const int limit = g_dpp_limit_override >= 0 ? std::min(g_dpp_limit_override, g_dpp_max_wgs_general) : g_dpp_max_wgs_general;
    const int tiled = limit / ((w + dpp_own(2) - 1) / dpp_own(2));
    // round 5: groups of 8 and more such carvers run on k_band_levels (7 or more slots per image, rows up to 4096 px), which takes
    // far larger groups than the full-width tiled kernels; its full DPs (3 per resize) then go to k_dp_sweep, one workgroup per image
    // (delta_x 5 .. 10, round 6: the full-width tiled kernels only -- a change moves up to ten columns per row, the band is the whole
    // width after a few hundred rows and the level kernel's images stop at a collision: 16 x 4K at delta_x 8 spent 6.6 of 9.6 ms per seam
    // in the sweep that takes over)
    if (delta <= 4 && g_band_levels != 0 && g_update_mode < 0 && (w + 63) / 64 <= LV_MAX_TILES) {
        const int lim_lv = g_dpp_limit_override >= 0 ? std::min(g_dpp_limit_override, g_dpp_max_wgs_levels) : g_dpp_max_wgs_levels;
        const int lv = lim_lv / 7;
        if (lv >= 8) return std::max(tiled, lv);
    }
    return tiled;
}

// E5 (UPDATE = false) or the full-width form of E9 (UPDATE = true) as one persistent launch
// (first, count): a range of the batch's images (E5 only: a general batch too large for one persistent grid is swept group after group)
template <bool UPDATE>
static int launch_dp_persistent(LqrHipBatch *b, const DpK &k, int w, int h, int lr, int first = 0, int count = -1)
{
    LqrHipCarver *c0 = b->cs[0];
    const size_t n = count < 0 ? b->cs.size() : (size_t) count;
    if (UPDATE && count >= 0) return LQRHIP_EARG;
    bool rigm = false;
    for (auto *c : b->cs) rigm |= (c->rig != nullptr);
    rigm = rigm && k.use_rig;                                  // without rigidity the mask multiplies nothing
    const bool general = k.delta != 1 || rigm;
    if (k.delta < 1 || k.delta > LQR_FAST_MAX_DELTA) return LQRHIP_EARG;
    const int px = dp_persistent_px(b, w, general, k.delta, count);
    if (!px) return LQRHIP_EARG;
    const int ntiles = (w + dpp_own(px) - 1) / dpp_own(px);
    int rc;
    const size_t need_elems = 2 * ((size_t) ntiles * dpp_ex_tile(px) + 8) * n;      // (granules + finished-tile counter, and the near copies: k_tiles.hip)
    if (b->exch_elems < need_elems) {
        HIPCK(hipStreamSynchronize(b->stream));
        dfree(b->exch);
        b->exch_elems = 0;
        if ((rc = dmalloc(&b->exch, need_elems))) return rc;
        b->exch_elems = need_elems;
        b->exch_ntiles = 0;
    }
    if (b->exch_ntiles != ntiles || b->exch_n != (int) n || b->exch_px != px) {
        // (re)lay the exchange area out: tags and finished-tile counters start at 0 (afterwards nothing is ever
        // cleared: tags carry the launch epoch, the last tile re-arms the counter)
        HIPCK(hipMemsetAsync(b->exch, 0, need_elems * sizeof(unsigned long long), b->stream));
        b->exch_ntiles = ntiles; b->exch_n = (int) n; b->exch_px = px;
    }
    if (UPDATE) {
        // second planes, allocated on first use -- per carver: a batch may mix carvers that already went
        // through a tiled update on their own with fresh ones
        bool grew = false;
        for (auto *c : b->cs) {
            if (c->m2 && c->least2) continue;
            if (!grew) { HIPCK(hipStreamSynchronize(b->stream)); grew = true; }
            const size_t pe = (size_t) c->stride * (c->wk_h + 1) + 1024;
            const bool fresh_m2 = !c->m2, fresh_l2 = !c->least2;
            if (!c->m2 && (rc = dmalloc(&c->m2, pe))) return rc;
            if (!c->least2 && (rc = dmalloc(&c->least2, pe))) { if (fresh_m2) dfree(c->m2); return rc; }     // (an m2 that was never zeroed must not pass for an old one)
            // like the first planes (ensure_working): nothing in them depends on what the block held before
            if (fresh_m2) HIPCK(hipMemsetAsync(c->m2, 0, pe * sizeof(float), b->stream));
            if (fresh_l2) HIPCK(hipMemsetAsync(c->least2, 0, pe, b->stream));
        }
        if (grew) {
            b->dirty = true;
            if ((rc = batch_upload(b))) return rc;
        }
    }
    const int epoch = 1 + ((b->tile_epoch++) % ((1 << (31 - DPP_BLK_BITS)) - 2));          // never 0; above the block index in the 32-bit tag
    const dim3 grid(ntiles, (unsigned) n);
#define LAUNCH_TILE(PXV, LRV, RIGV) hipLaunchKernelGGL((k_dp_tile_p<PXV, LRV, RIGV, UPDATE>), grid, dim3(64 * DPP_W), 0, b->stream, b->d_desc + first, k, w, h, c0->stride, b->exch, epoch, g_dev_err)
#define LAUNCH_TILE_PX(PXV)                                                                 \
    do {                                                                                    \
        if (lr) { if (k.use_rig) LAUNCH_TILE(PXV, true, true); else LAUNCH_TILE(PXV, true, false); }     \
        else { if (k.use_rig) LAUNCH_TILE(PXV, false, true); else LAUNCH_TILE(PXV, false, false); }      \
    } while (0)
#define LAUNCH_TILE_G(LRV, RIGV, DV, RMV) hipLaunchKernelGGL((k_dp_tile_p<2, LRV, RIGV, UPDATE, DV, RMV>), grid, dim3(64 * DPP_W), 0, b->stream, b->d_desc + first, k, w, h, c0->stride, b->exch, epoch, g_dev_err)
#define LAUNCH_TILE_G_LR(RIGV, DV, RMV) do { if (lr) LAUNCH_TILE_G(true, RIGV, DV, RMV); else LAUNCH_TILE_G(false, RIGV, DV, RMV); } while (0)
    if (general) {
#define LAUNCH_TILE_G_D(DV) do { if (!k.use_rig) LAUNCH_TILE_G_LR(false, DV, false); else if (!rigm) LAUNCH_TILE_G_LR(true, DV, false); else LAUNCH_TILE_G_LR(true, DV, true); } while (0)
        // delta_x 5 .. 10 (round 6): the rigidity form only -- without rigidity the host's table is all zeros, and x + 0.0f is x for every
        // candidate (no cumulative minimum is -0.0f: energies are sums of non-negative gradients and finite biases)
#define LAUNCH_TILE_G_W(DV) do { if (rigm) LAUNCH_TILE_G_LR(true, DV, true); else LAUNCH_TILE_G_LR(true, DV, false); } while (0)
        switch (k.delta) {
        case 1: LAUNCH_TILE_G_LR(true, 1, true); break;
        case 2: LAUNCH_TILE_G_D(2); break;
        case 3: LAUNCH_TILE_G_D(3); break;
        case 4: LAUNCH_TILE_G_D(4); break;
        case 5: LAUNCH_TILE_G_W(5); break;
        case 6: LAUNCH_TILE_G_W(6); break;
        case 7: LAUNCH_TILE_G_W(7); break;
        case 8: LAUNCH_TILE_G_W(8); break;
        case 9: LAUNCH_TILE_G_W(9); break;
        default: LAUNCH_TILE_G_W(10); break;
        }
#undef LAUNCH_TILE_G_W
#undef LAUNCH_TILE_G_D
    }
    else if (px == 2) LAUNCH_TILE_PX(2);
    else if (px == 3) {
#define LAUNCH_TILE_W(LRV, RIGV) hipLaunchKernelGGL((k_dp_tile_p<2, LRV, RIGV, UPDATE, 1, false, 24>), grid, dim3(64 * DPP_W), 0, b->stream, b->d_desc + first, k, w, h, c0->stride, b->exch, epoch, g_dev_err)
        if (lr) { if (k.use_rig) LAUNCH_TILE_W(true, true); else LAUNCH_TILE_W(true, false); }
        else { if (k.use_rig) LAUNCH_TILE_W(false, true); else LAUNCH_TILE_W(false, false); }
#undef LAUNCH_TILE_W
    }
    else LAUNCH_TILE_PX(4);
#undef LAUNCH_TILE_G_LR
#undef LAUNCH_TILE_G
#undef LAUNCH_TILE_PX
#undef LAUNCH_TILE
    HIPCK(hipGetLastError());
    if (UPDATE)       // the kernel's last tile swapped the pointers in the device descriptors: mirror it
        for (auto *c : b->cs) { std::swap(c->m, c->m2); std::swap(c->least, c->least2); }
    return 0;
}

template <bool UPDATE>
static int launch_dp(LqrHipBatch *b, const DpK &k, int w, int h, int lr)
{
    LqrHipCarver *c0 = b->cs[0];
    if (!UPDATE) {
        bool rigm = false;
        for (auto *c : b->cs) rigm |= (c->rig != nullptr);
        rigm = rigm && k.use_rig;
        if (k.delta == 1 && !rigm) return dp_persistent_ok(b, w) ? launch_dp_persistent<false>(b, k, w, h, lr) : launch_dp_tiled(b, k, w, h, lr);
        if (k.delta >= 1 && k.delta <= LQR_FAST_MAX_DELTA && dp_persistent_px(b, w, true, k.delta)) return launch_dp_persistent<false>(b, k, w, h, lr);
        // round 5: a general batch too large for one persistent grid (it runs its incremental updates on k_band_levels): the full DP
        // group after group of as many images as fit, instead of one 1024-thread workgroup per image (k_dp_sweep: 4 - 8 ms per sweep
        // of 16 x 4K against 2 x 0.5)
        if (k.delta >= 1 && k.delta <= LQR_FAST_MAX_DELTA && !b->shared) {
            const int total = (int) b->cs.size();
            int per = total;
            while (per > 1 && !dp_persistent_px(b, w, true, k.delta, per)) per = (per + 1) / 2;
            if (dp_persistent_px(b, w, true, k.delta, per)) {
                int rc;
                for (int first = 0; first < total; first += per)
                    if ((rc = launch_dp_persistent<false>(b, k, w, h, lr, first, std::min(per, total - first)))) return rc;
                return 0;
            }
        }
    }
    // the launch behind a band kernel (UPDATE) almost always only looks at flags[FLAG_OVF_ROW]: 256 threads for rows up to 4096 px (a
    // workgroup that finds room at once beside the sibling streams' kernels), 1024 for the full sweeps and wider rows
    const int nth = (UPDATE && g_sweep_threads == 256 && w <= 16 * 256) ? 256 : DP_THREADS;
    int pxt = (w + nth - 1) / nth;
    size_t lds = (size_t) 2 * ((w + 3) & ~3) * sizeof(float);
    dim3 grid((unsigned) b->cs.size());
#define LAUNCH_DP_T(P, T)                                                                                             \
    do {                                                                                                              \
        if (lds > 64 * 1024)                                                                                          \
            HIPCK(hipFuncSetAttribute((const void *) k_dp_sweep<P, UPDATE, T>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int) lds));                                                                    \
        hipLaunchKernelGGL((k_dp_sweep<P, UPDATE, T>), grid, dim3(T), lds, b->stream, b->d_desc, k, w, h, c0->stride, lr); \
    } while (0)
#define LAUNCH_DP(P) do { if constexpr (UPDATE) { if (nth == 256) LAUNCH_DP_T(P, 256); else LAUNCH_DP_T(P, DP_THREADS); } else LAUNCH_DP_T(P, DP_THREADS); } while (0)
    if (pxt <= 1) LAUNCH_DP(1);
    else if (pxt <= 2) LAUNCH_DP(2);
    else if (pxt <= 4) LAUNCH_DP(4);
    else if (pxt <= 8) LAUNCH_DP(8);
    else if (pxt <= 16) LAUNCH_DP(16);
    else { g_err = "image wider than 16384 px is not supported"; return LQRHIP_EARG; }
#undef LAUNCH_DP_T
#undef LAUNCH_DP
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int lqrhip_mmap_build(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h, int leftright)
{
    int rc;
    if ((rc = batch_upload(b))) return rc;
    if (p->delta_x > LQRHIP_MAX_DELTA) return LQRHIP_EARG;
    ProfScope ps("dp_sweep", b->stream, 9.0 * w * h * b->cs.size());
    return launch_dp<false>(b, make_dpk(p, b->cs[0]->ch), w, h, leftright);
}

#ifndef FROZEN_LAG_MAX
#define FROZEN_LAG_MAX 128      // seams the frozen planes may lag behind before they are compacted
#endif

// remove seams [epoch, to) from the frozen planes of every carver of the batch
static int frozen_catchup(LqrHipBatch *b, int to, int w_at_to, int h)
{
    LqrHipCarver *c0 = b->cs[0];
    const int from = c0->frozen_epoch;
    if (to <= from) return 0;
    const int w_from = w_at_to + (to - from);
    size_t lds = (size_t) (to - from) * sizeof(int) + (size_t) w_from + 16;
    hipLaunchKernelGGL(k_frozen_catchup, dim3(h, (unsigned) b->cs.size()), dim3(256), lds, b->stream, b->d_desc, from, to, w_from, h,
                       c0->stride);
    HIPCK(hipGetLastError());
    for (auto *c : b->cs) c->frozen_epoch = to;
    return 0;
}

// Slots (workgroups) per image for k_band_levels (0: not usable here): as many as the group's images leave room for within the
// residency bound, at most LV_PMAX; lqrhip_set_band_levels pins it (tests, experiments).  The default for large groups is 6:
// the active tiles of a 4K level are 4.5 on average, a window of 6 consecutive tiles never collides, and 64 x 6 workgroups hold
// half the registers of round 4's 64 x 12.
extern "C" void lqrhip_set_band_levels(int slots) { g_band_levels = slots; }
static int band_levels_P(const LqrHipBatch *b, int w, int h, int delta)
{
    if (no_spin(b) || g_band_levels == 0 || delta < 1 || delta > LQR_FAST_MAX_DELTA || (h + lv_rows(delta, true) - 1) / lv_rows(delta, true) > LV_MAX_LEVELS || (w + 63) / 64 > LV_MAX_TILES) return 0;
    const int limit = g_dpp_limit_override >= 0 ? std::min(g_dpp_limit_override, g_dpp_max_wgs_levels) : g_dpp_max_wgs_levels;
    const int per_batch = limit / std::max(b->shared_n, 1);
    int P = std::min(LV_PMAX, per_batch / (int) std::max<size_t>(b->cs.size(), 1));
    // automatic: 12 slots while the group's workgroups stay below ~384 (beyond that the sibling kernels are starved of registers,
    // 48 images with 10 slots 452 k, with 8 slots 479 k), never fewer than 7 (6 and fewer put second tiles on a slot in 9 % of the tile-levels)
    const size_t group_images = b->cs.size() * (size_t) std::max(b->shared_n, 1);
    const int want = g_band_levels > 0 ? g_band_levels : std::max(7, std::min(12, (int) (384 / std::max<size_t>(group_images, 1))));
    P = std::min(P, want);
    return P >= (g_band_levels > 0 ? 1 : 7) ? P : 0;
}
static int launch_band_levels(LqrHipBatch *b, const DpK &k, int w, int h, int lr, int P, bool rigm)
{
    LqrHipCarver *c0 = b->cs[0];
    const size_t n = b->cs.size();
    int rc;
    const int ntiles = (w + 63) / 64;
    const size_t need_elems = 2 * ((size_t) 4 * LV_PMAX + (size_t) 2 * ntiles * 64) * n;    // (two copies of each word: k_levels.hip, near_off)
    if (b->exch_elems < need_elems) {
        HIPCK(hipStreamSynchronize(b->stream));
        dfree(b->exch);
        b->exch_elems = 0;
        if ((rc = dmalloc(&b->exch, need_elems))) return rc;
        b->exch_elems = need_elems;
        b->exch_ntiles = 0;
    }
    if (b->exch_ntiles != ntiles || b->exch_n != (int) n || b->exch_px != 103) {       // 103: this kernel's layout and tags
        HIPCK(hipMemsetAsync(b->exch, 0, need_elems * sizeof(unsigned long long), b->stream));
        b->exch_ntiles = ntiles; b->exch_n = (int) n; b->exch_px = 103;
    }
    const int epoch = 1 + ((b->tile_epoch++) % ((1 << 22) - 2));           // never 0; 22 bits above the 10 bits of level + 1
    const dim3 grid((unsigned) (8 * ((n + 7) / 8) * P));          // the slots of an image on one XCD (k_levels.hip)
#define LAUNCH_LV(LRV, RIGV, DV, RMV) hipLaunchKernelGGL((k_band_levels<LRV, RIGV, DV, RMV>), grid, dim3(128), 0, b->stream, b->d_desc, k, w, h, c0->stride, b->exch, epoch, g_dev_err, P, (int) n)
#define LAUNCH_LV_LR(RIGV, DV, RMV) do { if (lr) LAUNCH_LV(true, RIGV, DV, RMV); else LAUNCH_LV(false, RIGV, DV, RMV); } while (0)
#define LAUNCH_LV_D(DV) do { if (!k.use_rig) LAUNCH_LV_LR(false, DV, false); else if (!rigm) LAUNCH_LV_LR(true, DV, false); else LAUNCH_LV_LR(true, DV, true); } while (0)
#define LAUNCH_LV_W(DV) do { if (rigm) LAUNCH_LV_LR(true, DV, true); else LAUNCH_LV_LR(true, DV, false); } while (0)      // delta_x 5 .. 10: the rigidity form (zero table without rigidity)
    switch (k.delta) {
    case 1: LAUNCH_LV_D(1); break;
    case 2: LAUNCH_LV_D(2); break;
    case 3: LAUNCH_LV_D(3); break;
    case 4: LAUNCH_LV_D(4); break;
    case 5: LAUNCH_LV_W(5); break;
    case 6: LAUNCH_LV_W(6); break;
    case 7: LAUNCH_LV_W(7); break;
    case 8: LAUNCH_LV_W(8); break;
    case 9: LAUNCH_LV_W(9); break;
    default: LAUNCH_LV_W(10); break;
    }
#undef LAUNCH_LV_W
#undef LAUNCH_LV_D
#undef LAUNCH_LV_LR
#undef LAUNCH_LV
    HIPCK(hipGetLastError());
    return 0;
}
// batches up to this many pixels use the tiled full-width update (measured break-even with the band kernel at 4K, Mseams*px/s
// tiled / band: 7 images 118 k / 97 k, 8: 130 / 109, 9: 119 / 121, 12: 130 / 139+, 16: 158 / 175+)
static const long long g_tiled_update_px = 8LL * 3840 * 2160;

// One seam of a lock-step batch: k_vpath* (pick + backtrack, publishes the side to move) -> k_carve ->
// k_emap_update -> one form of update_mmap (or the full DP after a side switch), all on the batch's stream.
static int seam_step_impl(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h, int log_index, int leftright_pick,
                          int full_rebuild, int leftright_next);
static void inject_after_step(LqrHipBatch *b, int h, int log_index);
static void inject_after_commit(LqrHipBatch *b, int w0, int h0, int first_level);
// LQRHIP_DUMP=<prefix> (debugging aid): after every seam step the first image's seam, flags and DP planes go to
// <prefix>_<call>.bin -- header {w, h, stride, log_index, FLAG_COUNT}, flags, seam_x[h], m[stride * h], least[stride * h]
extern "C" int lqrhip_seam_step(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h, int log_index, int leftright_pick,
                                int full_rebuild, int leftright_next)
{
    const int rc = seam_step_impl(b, p, w, h, log_index, leftright_pick, full_rebuild, leftright_next);
    if (rc == 0) inject_after_step(b, h, log_index);
    static const char *dump = getenv("LQRHIP_DUMP");
    if (rc == 0 && dump) {
        static int call = 0;
        LqrHipCarver *c = b->cs[0];
        HIPCK(hipStreamSynchronize(b->stream));
        const size_t np = (size_t) c->stride * h;
        std::vector<int> hdr = {w, h, c->stride, log_index, FLAG_COUNT}, flags(FLAG_COUNT), seam(h);
        std::vector<float> m(np);
        std::vector<int8_t> least(np);
        HIPCK(hipMemcpy(flags.data(), c->flags, FLAG_COUNT * sizeof(int), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(seam.data(), c->seam_x, (size_t) h * sizeof(int), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(m.data(), c->m, np * sizeof(float), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(least.data(), c->least, np, hipMemcpyDeviceToHost));
        char path[512];
        snprintf(path, sizeof path, "%s_%04d.bin", dump, call++);
        if (FILE *f = fopen(path, "wb")) {
            fwrite(hdr.data(), sizeof(int), hdr.size(), f); fwrite(flags.data(), sizeof(int), flags.size(), f);
            fwrite(seam.data(), sizeof(int), seam.size(), f); fwrite(m.data(), sizeof(float), np, f); fwrite(least.data(), 1, np, f);
            fclose(f);
        }
    }
    return rc;
}

static int seam_step_impl(LqrHipBatch *b, const LqrHipDpParams *p, int w, int h, int log_index, int leftright_pick,
                          int full_rebuild, int leftright_next)
{
    int rc;
    LqrHipCarver *c0 = b->cs[0];
    for (auto *c : b->cs)
        if (log_index >= c->log_cap) return LQRHIP_EARG;
    if ((rc = batch_upload(b))) return rc;
    const unsigned n = (unsigned) b->cs.size();
    DpK k = make_dpk(p, c0->ch);
    const int stride = c0->stride;
    const int wnew = w - 1;
    const int move_dp = (wnew > 1 && !full_rebuild) ? 1 : 0;
    bool has_rigmask = false;
    for (auto *c : b->cs) has_rigmask |= (c->rig != nullptr);
    // bytes the carve moves per pixel of the side it moves, read + write: en 4 (+ m 4 + back pointer 1 unless a full DP follows,
    // + the rigidity mask 4) -- k_vpath* knows how many pixels that is for the seam it finds and keeps the sum (lqrhip_moved_bytes)
    const int moved_unit = 2 * (4 + (move_dp ? 5 : 0) + (has_rigmask ? 4 : 0));
    // Backtrack: for single images the two-kernel parallel form (k_vp_maps / k_vp_solve, k_backtrack.hip: the chip walks every column
    // through every chunk of rows, the serial part is one step per chunk); for groups the one-wave-per-image walk, whose launches
    // keep the chip busy anyway.  Measured on one box, us per seam with every kernel event-timed, k_vpath1 / parallel: 4K 70 / 41 (single4k
    // 20.9 -> 22.7 k Mseams*px/s), 8K 108 / 64 (config 5 55.5 -> 60.8 k), FHD 32 / 30, 2 x 4K 57 / 44 (51.0 -> 53.4 k), 4 x 4K 59 / 61,
    // 8 x 4K 60 / 79: each launch is ~10 us of dependent-dispatch latency, and the maps of n images are n times the work.
    const size_t vp_group = (size_t) n * (size_t) std::max(b->shared_n, 1);
    // delta_x 5 .. 10: always -- the one-wave walks there take 0.2 (k_vpath1<5>) to 1.15 ms (k_vpath, delta_x 10) per 4K seam, the parallel
    // form ~0.06 whatever delta_x is (a chunk is 56 / delta_x rows, the cone of a stage as wide as at delta_x 1)
    const bool use_vp = p->delta_x >= 1 && p->delta_x <= LQR_FAST_MAX_DELTA && h >= 2 && g_vpath_mode != 0 &&
                        (g_vpath_mode == 1 || p->delta_x >= 5 || (vp_group <= (size_t) g_vpath_par_max && h >= g_vpath_min_rows));
    if (use_vp) {
        const int R = vp_chunk_rows(p->delta_x), nchunks = (h - 1 + R - 1) / R, R4 = (R + 3) / 4;
        const size_t need = (size_t) (nchunks + 1) * stride + 64;
        bool grew = false;
        for (auto *c : b->cs) {
            if (c->vp_map && c->vp_path && c->vp_cap >= need) continue;
            if (!grew) { HIPCK(hipStreamSynchronize(b->stream)); grew = true; }
            dfree(c->vp_map); dfree(c->vp_path); c->vp_cap = 0;
            if ((rc = dmalloc(&c->vp_map, need)) || (rc = dmalloc(&c->vp_path, need * 4 * R4))) return rc;
            c->vp_cap = need;
        }
        if (grew) { b->dirty = true; if ((rc = batch_upload(b))) return rc; }
        ProfScope ps("vpath", b->stream, 0);
#define LAUNCH_VP(DV) do { \
        hipLaunchKernelGGL(k_vp_maps<DV>, dim3((w + 255) / 256, nchunks, n), dim3(256), 0, b->stream, b->d_desc, w, h, stride); \
        hipLaunchKernelGGL(k_vp_solve<DV>, dim3(n), dim3(VPATH_THREADS), (size_t) (nchunks + 2) * sizeof(int), b->stream, b->d_desc, w, h, stride, leftright_pick, log_index, moved_unit); } while (0)
        switch (p->delta_x) {
        case 1: LAUNCH_VP(1); break; case 2: LAUNCH_VP(2); break; case 3: LAUNCH_VP(3); break; case 4: LAUNCH_VP(4); break; case 5: LAUNCH_VP(5); break;
        case 6: LAUNCH_VP(6); break; case 7: LAUNCH_VP(7); break; case 8: LAUNCH_VP(8); break; case 9: LAUNCH_VP(9); break; default: LAUNCH_VP(10); break;
        }
#undef LAUNCH_VP
    } else {
        ProfScope ps("vpath", b->stream, 0);
        if (p->delta_x == 1)
            hipLaunchKernelGGL(k_vpath1<1>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index, moved_unit);
        else if (p->delta_x == 2)
            hipLaunchKernelGGL(k_vpath1<2>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index, moved_unit);
        else if (p->delta_x == 3)
            hipLaunchKernelGGL(k_vpath1<3>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index, moved_unit);
        else if (p->delta_x == 4)
            hipLaunchKernelGGL(k_vpath1<4>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index, moved_unit);
        else if (p->delta_x == 5)
            hipLaunchKernelGGL(k_vpath1<5>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index, moved_unit);
        else if (p->delta_x == 6)
            hipLaunchKernelGGL(k_vpath1<6>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index, moved_unit);
        else if (p->delta_x == 7)
            hipLaunchKernelGGL(k_vpath1<7>, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, log_index, moved_unit);
        else
            hipLaunchKernelGGL(k_vpath, dim3(n), dim3(VPATH_THREADS), 0, b->stream, b->d_desc, w, h, stride, leftright_pick, p->delta_x,
                               log_index, moved_unit);
    }
    // Single images and groups up to 4: the carve and the energy update in one launch (k_carve_e, k_carve.hip) -- the wave that has moved
    // a row refreshes that row's energies; one dependent launch less per seam.  delta_x <= 2 (12 brightness samples per row).
    const bool fuse_e = p->delta_x <= 2 && vp_group <= (size_t) g_carve_fused && wnew > 1;
    if (fuse_e) {
        const int lag_max = n <= 4 ? FROZEN_LAG_MAX / 4 : FROZEN_LAG_MAX;
        if (log_index + 1 - c0->frozen_epoch > lag_max && (rc = frozen_catchup(b, log_index + 1, wnew, h))) return rc;      // (needs the seam log only: before the carve)
        const int epoch = c0->frozen_epoch;
        ProfScope ps("carve", b->stream, 4.0 * (double) w * h * n);
#define LAUNCH_CE(N) hipLaunchKernelGGL((k_carve_e<N>), dim3((h + 3) / 4, n), dim3(256), 0, b->stream, b->d_desc, k, w, h, stride, move_dp, log_index, epoch)
        NRG_DISPATCH(p->nrg_func, LAUNCH_CE)
#undef LAUNCH_CE
    } else {
        // algorithmic bytes of one carve launch: read + write of one 4-byte
        // plane over the half of each row right of the seam = 8 B * w*h/2 per image
        ProfScope ps("carve", b->stream, 4.0 * (double) w * h * n);
        // (a grid of a half, a quarter, an eighth of the rows, the kernel striding over the rest, measured at 64 x 4K in round 5: 508 / 504 / 490 k
        // against 488 - 501 k: inside the run-to-run spread; not adopted)
        hipLaunchKernelGGL(k_carve, dim3((h + 3) / 4, n), dim3(256), 0, b->stream, b->d_desc, w, h, stride, p->delta_x, move_dp);
    }
    if (wnew <= 1) {            // liblqr's finish_vsmap case: nothing left to update
        HIPCK(hipGetLastError());
        return 0;
    }
    if (!fuse_e) {
        ProfScope ps("emap_update", b->stream, 0);
        // the energy update walks the seam log back to the frozen frame (O(lag) per sample); compacting the frozen planes
        // costs a pass over them.  Few images: the walk is on the critical path and the pass is cheap -> short lag
        const int lag_max = n <= 4 ? FROZEN_LAG_MAX / 4 : FROZEN_LAG_MAX;
        if (log_index + 1 - c0->frozen_epoch > lag_max && (rc = frozen_catchup(b, log_index + 1, wnew, h))) return rc;
        const int epoch = c0->frozen_epoch;
#define LAUNCH_EUPD_NT(N, NT) hipLaunchKernelGGL((k_emap_update<N, NT>), dim3((h + EU_ROWS - 1) / EU_ROWS, n), dim3(64), 0, b->stream, b->d_desc, k, wnew, h, stride, log_index, epoch)
#define LAUNCH_EUPD(N) do { if (p->delta_x <= 2) LAUNCH_EUPD_NT(N, 12); else if (p->delta_x <= 8) LAUNCH_EUPD_NT(N, 36); else LAUNCH_EUPD_NT(N, 68); } while (0)
        NRG_DISPATCH(p->nrg_func, LAUNCH_EUPD)
#undef LAUNCH_EUPD
#undef LAUNCH_EUPD_NT
    }
    if (full_rebuild) {
        ProfScope ps("dp_sweep", b->stream, 9.0 * wnew * h * n);
        if ((rc = launch_dp<false>(b, k, wnew, h, leftright_next))) return rc;
        HIPCK(hipGetLastError());
        return 0;
    }
    // How E9 (update_mmap) runs.  Small batches: the whole chip recomputing every row (tiled full-width keep-rule
    // sweep) beats the one-workgroup-per-image band walk; for large batches its 14 B/px of traffic would not.
    // "plain": delta_x = 1 and no rigidity mask that matters -- every fast kernel.  delta_x = 2 and rigidity masks run on the
    // tiled full-width update (k_dp_tile_p's general instantiations) whenever its grid fits; only beyond that do they fall
    // to the one-wave-per-image band kernel and the one-workgroup-per-image sweep (measured at 8K: 37x slower)
    const bool rigm = has_rigmask && p->use_rigidity;
    const bool fast_ok = p->delta_x == 1 && !rigm && g_update_mode != 3;
    const bool tiled_update = fast_ok ? ((g_update_mode < 0 ? (size_t) n * (size_t) w * (size_t) h <= (size_t) g_tiled_update_px : g_update_mode == 1) &&
                                         dp_persistent_ok(b, w))
                                      : (p->delta_x >= 1 && p->delta_x <= LQR_FAST_MAX_DELTA && g_update_mode != 0 && g_update_mode != 2 && g_update_mode != 3 && dp_persistent_px(b, w, true, p->delta_x) != 0);
    {
        // round 5: the band on P slots per image, tiles assigned level by level (k_band_levels): the default for groups of 8 to 64
        // images (measured, Mseams*px/s at 4K, levels / k_band_update_tw: 8 images 148 / 127, 16: 256 / 180, 48: 479 / 404).  64 images
        // on four streams with 7 slots: alternating 3-step runs on three boxes 516 / 495-516, 497 / 488, 520 / 504; the driver's
        // 20-step command on two boxes of equal speed (every other figure within 1 %) 567.2 / 538.9 k.  Its 448 workgroups stretch
        // the sibling streams' carves (k_carve 172 -> 203 us per launch) while the whole step's share of the HBM roof rises (0.279
        // -> 0.293).  96 images run 585 / 622: groups above 64 keep k_band_update_tw; update mode 0 / 5 force either
        // round 5: also delta_x 2 .. 4 and rigidity masks (k_band_levels' general instantiations): a batch of such carvers used to be
        // carved in groups of as many as the full-width tiled kernels hold (16 x 4K, delta_x 2: 68 k Mseams*px/s)
        const size_t group_images = (size_t) n * (size_t) std::max(b->shared_n, 1);
        const bool lv_ok = p->delta_x >= 1 && (p->delta_x <= 4 || g_update_mode == 5) && p->delta_x <= LQR_FAST_MAX_DELTA && g_update_mode != 3;        // (delta_x 5 .. 10: on request only, see lqrhip_general_batch_limit_delta)
        const int PL = (lv_ok && (g_update_mode == 5 || (g_update_mode < 0 && group_images >= 8 && (group_images <= 64 || !fast_ok)))) ? band_levels_P(b, wnew, h, p->delta_x) : 0;
        if (PL > 0) {
            {
                ProfScope ps("band_levels", b->stream, 0);
                if ((rc = launch_band_levels(b, k, wnew, h, leftright_next, PL, rigm))) return rc;
            }
#ifndef LQR_EXP_NO_SWEEP        // (experiment builds only: what the almost always empty launch costs; results are wrong when an image stopped)
            ProfScope ps("dp_update", b->stream, 0);
            if ((rc = launch_dp<true>(b, k, wnew, h, leftright_next))) return rc;
#endif
            HIPCK(hipGetLastError());
            return 0;
        }
    }
    if (tiled_update) {
        ProfScope ps("dp_update_tiled", b->stream, 0);
        if ((rc = launch_dp_persistent<true>(b, k, wnew, h, leftright_next))) return rc;
        HIPCK(hipGetLastError());
        return 0;
    }
    const bool fast_band = fast_ok && (size_t) h * sizeof(int) <= 60 * 1024;
    // the trapezoid-wave band kernel takes rows up to ~4200 px (wider rows: the changes outgrow its 896-column window
    // too often, and an 8-slot build spills registers); beyond that, and in update mode 2, k_band_update_mw
    const bool band_tw = fast_band && g_update_mode != 2 && wnew <= 4200 && (size_t) 2 * h * sizeof(int) <= 64 * 1024;
    if (band_tw) {
        ProfScope ps("band_update", b->stream, 0);
#define LAUNCH_TW(LRV, RIGV) hipLaunchKernelGGL((k_band_update_tw<4, LRV, RIGV>), dim3(n), dim3(128 * 4), (size_t) 2 * h * sizeof(int), b->stream, b->d_desc, k, wnew, h, stride, g_dev_err)
        if (leftright_next) { if (p->use_rigidity) LAUNCH_TW(true, true); else LAUNCH_TW(true, false); }
        else { if (p->use_rigidity) LAUNCH_TW(false, true); else LAUNCH_TW(false, false); }
#undef LAUNCH_TW
        // (round 5: the kernel finishing the rows its window cannot hold itself, without the (almost always empty) k_dp_sweep<UPDATE>
        // launch behind it, was built and measured on one box: 527.5 / 523.5 k against 531 / 528 k -- the launch's 22 us reappear in
        // the kernels around it (k_vpath1 74 -> 92 us, k_carve 149 -> 162), the step is not the sum of a chain's kernels; removed)
    } else if (fast_band) {
        ProfScope ps("band_update", b->stream, 0);
#define LAUNCH_MW(NWV, LRV, RIGV) hipLaunchKernelGGL((k_band_update_mw<2, NWV, 8, LRV, RIGV>), dim3(n), dim3(64 * NWV), (size_t) h * sizeof(int), b->stream, b->d_desc, k, wnew, h, stride)
#define LAUNCH_MW_N(LRV, RIGV) do { if (wnew > 4200) LAUNCH_MW(16, LRV, RIGV); /* 8K: dirty regions up to ~900 px */ else LAUNCH_MW(8, LRV, RIGV); } while (0)
        if (leftright_next) { if (p->use_rigidity) LAUNCH_MW_N(true, true); else LAUNCH_MW_N(true, false); }
        else { if (p->use_rigidity) LAUNCH_MW_N(false, true); else LAUNCH_MW_N(false, false); }
#undef LAUNCH_MW_N
#undef LAUNCH_MW
    } else {
        ProfScope ps("band_update", b->stream, 0);
        hipLaunchKernelGGL(k_band_update, dim3(n), dim3(64), 0, b->stream, b->d_desc, k, wnew, h, stride, leftright_next);
    }
    {
        // rows the band kernel handed over (flags[FLAG_OVF_ROW] .. h): the keep rule over the full width
        ProfScope ps("dp_update", b->stream, 0);
        if ((rc = launch_dp<true>(b, k, wnew, h, leftright_next))) return rc;
    }
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int lqrhip_seam_log_reserve(LqrHipBatch *b, int n_seams, int h)
{
    int rc;
    for (auto *c : b->cs) if ((rc = ensure_log(c, n_seams, h))) return rc;
    return 0;
}

extern "C" int lqrhip_vs_commit(LqrHipBatch *b, int w0, int h0, int wc0, int n_seams, int first_level, int finish)
{
    int rc;
    if ((rc = batch_upload(b))) return rc;
    size_t lds = ((size_t) n_seams + wc0) * sizeof(int);
    if (lds > 150 * 1024) { g_err = "vs_commit: session too large for LDS"; return LQRHIP_EARG; }
    if (lds > 64 * 1024)
        HIPCK(hipFuncSetAttribute((const void *) k_vs_commit, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipLaunchKernelGGL(k_vs_commit, dim3(h0, (unsigned) b->cs.size()), dim3(256), lds, b->stream, b->d_desc, w0, h0, wc0, n_seams,
                       first_level, finish);
    HIPCK(hipGetLastError());
    inject_after_commit(b, w0, h0, first_level);
    // the session is over: bring the frozen planes to the carved frame, the log restarts at 0
    if ((rc = frozen_catchup(b, n_seams, wc0 - n_seams, h0))) return rc;
    for (auto *c : b->cs) c->frozen_epoch = 0;
    return 0;
}

// ---- session self-check, roll-back, fault injection (round 6) ---------------------------------------------------------------
// The seam loop's kernels are latency-bound protocols (spin barriers, data-tagged hand-overs); a defect or a device that
// preempts them must not end in LQR_OK with a wrong map.  Two structural checks run inside every session: k_seam_check on the
// seam log BEFORE the levels are committed, and the level count / uniqueness test fused into k_inflate.  Both cost < 1 % of a
// session (1.7 MB of log per 4K image; the inflate pass reads the levels anyway) and are on by default.
static int g_selfcheck = 1, g_recovery = 1;
extern "C" void lqrhip_set_selfcheck(int on) { g_selfcheck = on != 0; }
extern "C" void lqrhip_set_recovery(int on) { g_recovery = on != 0; }
extern "C" int lqrhip_get_recovery(void) { return g_recovery; }
extern "C" int lqrhip_session_check(LqrHipBatch *b, int h, int wc0, int n_seams, int delta_x)
{
    int rc;
    if (!g_selfcheck || n_seams < 1) return 0;
    if ((rc = batch_upload(b))) return rc;
    hipLaunchKernelGGL(k_seam_check, dim3((h + 255) / 256, (unsigned) b->cs.size()), dim3(256), 0, b->stream, b->d_desc, h, wc0, n_seams, delta_x, g_dev_err);
    HIPCK(hipGetLastError());
    return 0;
}
// Undo what a failed session left: drain the stream, drop the error record, clear the session's levels from the base layout
// (a no-op when they were never committed).  The working planes are garbage afterwards: the host re-lays them out from the
// base layout (lqrhip_wk_init(b, 1)) before it redoes the session.
extern "C" int lqrhip_session_rollback(LqrHipBatch *b, int w0, int h0, int first_level, int finish)
{
    (void) hipStreamSynchronize(b->stream);
    discard_pending(b);
    (void) hipGetLastError();
    if (g_dev_err_host) *g_dev_err_host = 0;
    invalidate_all_batches();
    int rc;
    if ((rc = batch_upload(b))) return rc;
    const size_t n = (size_t) w0 * h0;
    hipLaunchKernelGGL(k_vs_rollback, dim3((unsigned) std::min<size_t>((n + 255) / 256, 4096), (unsigned) b->cs.size()), dim3(256), 0, b->stream, b->d_desc, n, first_level, finish ? w0 : 0);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(b->stream));
    for (auto *c : b->cs) c->frozen_epoch = 0;
    g_fault_stats[4]++;
    return 0;
}
// Fault injection (tests/test_faults_gpu.py).  kind 1: a spin time-out, 2: a failed activity prediction -- the error word is
// written from the host while the kernels of seam step `at_step` of the next session run, which is exactly what the kernels see
// when one of them gives up (they all leave their spins); 3 / 4: one entry of the seam log out of the frame / disconnected (after
// that step); 5 / 6: one committed level cleared / duplicated (between the commit and the inflate pass).  `times`: how many
// sessions in a row are hit (1: the redo succeeds; 2: it fails too).  0 disarms.
static int g_inject_kind = 0, g_inject_step = 0, g_inject_times = 0;
extern "C" void lqrhip_debug_inject(int kind, int at_step, int times) { g_inject_kind = kind; g_inject_step = at_step; g_inject_times = kind ? std::max(times, 1) : 0; }
static void inject_after_step(LqrHipBatch *b, int h, int log_index)
{
    if (g_inject_times <= 0 || log_index != g_inject_step) return;
    if (g_inject_kind == 1 || g_inject_kind == 2) { *g_dev_err_host = g_inject_kind == 1 ? DEVERR_TILE_TIMEOUT : DEVERR_BAND_PREDICTION; }
    else if (g_inject_kind == 3 || g_inject_kind == 4) hipLaunchKernelGGL(k_inject, dim3(1), dim3(64), 0, b->stream, b->d_desc, g_inject_kind - 3, h, 0, log_index, 0);
    else return;
    g_inject_times--; g_fault_stats[5]++;
}
static void inject_after_commit(LqrHipBatch *b, int w0, int h0, int first_level)
{
    if (g_inject_times <= 0 || (g_inject_kind != 5 && g_inject_kind != 6)) return;
    if (g_inject_step > 0) { g_inject_step--; return; }          // (kinds 5 / 6: at_step = how many commits to let pass first -- the second sub-batch of a group)
    hipLaunchKernelGGL(k_inject, dim3(1), dim3(64), 0, b->stream, b->d_desc, g_inject_kind - 3, h0, w0, 0, first_level);
    g_inject_times--; g_fault_stats[5]++;
}

// E14 / E11: every carver of the batch (roots and their attached carvers) goes through ONE launch (a job table in device
// memory, one grid slice of blocks per job); the new planes replace the old ones after its synchronisation.  Everything
// staged for the pass is owned by a PlaneJobs object until then: any error return gives all of it back to the pool.
struct PlaneJob {
    LqrHipCarver *c;
    uint8_t *nrgb;
    float *nbias, *nrig;
};
struct PlaneJobs {
    std::vector<PlaneJob> jobs;
    std::vector<InflateDev> dev;
    std::vector<int32_t *> new_vs;          // one per root (may be null)
    InflateDev *d_jobs = nullptr;
    bool committed = false;
    ~PlaneJobs()
    {
        dfree(d_jobs);
        if (committed) return;
        for (auto &j : jobs) { dfree(j.nrgb); dfree(j.nbias); dfree(j.nrig); }
        for (auto *&v : new_vs) dfree(v);
    }
    // stage the output planes of carver c: n1 pixels each
    int add(LqrHipCarver *c, const int32_t *vs_old, int32_t *nvs, size_t n1)
    {
        PlaneJob j{c, nullptr, nullptr, nullptr};
        int rc = dmalloc(&j.nrgb, n1 * c->ch);
        if (!rc && c->bias0) rc = dmalloc(&j.nbias, n1);
        if (!rc && c->rig0) rc = dmalloc(&j.nrig, n1);
        jobs.push_back(j);                  // owned from here on, also when rc != 0
        if (rc) return rc;
        dev.push_back(InflateDev{c->rgb0, vs_old, c->bias0, c->rig0, j.nrgb, nvs, j.nbias, j.nrig, c->ch});
        return 0;
    }
    int upload(hipStream_t s)
    {
        int rc = dmalloc(&d_jobs, dev.size());
        if (rc) return rc;
        HIPCK(hipMemcpyAsync(d_jobs, dev.data(), dev.size() * sizeof(InflateDev), hipMemcpyHostToDevice, s));
        return 0;
    }
    // after the pass has completed: the new base planes become the carvers'
    void commit()
    {
        for (auto &j : jobs) {
            dfree(j.c->rgb0); j.c->rgb0 = j.nrgb;
            if (j.nbias) { dfree(j.c->bias0); j.c->bias0 = j.nbias; }
            if (j.nrig) { dfree(j.c->rig0); j.c->rig0 = j.nrig; }
        }
        committed = true;
    }
};

// Two phases (round 6): lqrhip_inflate / lqrhip_flatten / lqrhip_transpose stage the new planes of the batch, run the pass (the inflate
// pass with its fused self-check) and ADOPT NOTHING; lqrhip_planes_commit adopts what was staged.  The host runs phase one on every
// sub-batch of a group before it commits any: a failed check or a failed allocation in sub-batch k must not find sub-batches 0 .. k - 1
// already living in their new layouts (the roll-back / the error return leaves the whole group where it was).
// lqrhip_session_rollback / lqrhip_batch_abort / lqrhip_batch_destroy discard a staged pass.
struct PendingInflate { PlaneJobs pj; int kind = 0 /* 0 inflate, 1 flatten, 2 transpose */, w1 = 0, h1 = 0; };
static void discard_pending(LqrHipBatch *b) { delete (PendingInflate *) b->pending_inflate; b->pending_inflate = nullptr; }
static PendingInflate *new_pending(LqrHipBatch *b, int kind, int w1, int h1)
{
    discard_pending(b);
    PendingInflate *pi = new PendingInflate();
    b->pending_inflate = pi;                // owned by the batch from here on, whatever happens below
    pi->kind = kind; pi->w1 = w1; pi->h1 = h1;
    return pi;
}
extern "C" int lqrhip_inflate(LqrHipBatch *b, int w0, int h0, int l, int max_level)
{
    int rc;
    const int w1 = w0 + l - max_level + 1;
    PlaneJobs &pj = new_pending(b, 0, w1, h0)->pj;
    auto run = [&]() -> int {
        for (auto *c : b->cs) {
            int32_t *nvs = nullptr;
            if ((rc = dmalloc(&nvs, (size_t) w1 * h0))) return rc;
            pj.new_vs.push_back(nvs);
            for (auto *a : c->aux)
                if ((rc = pj.add(a, c->vs, nullptr, (size_t) w1 * h0))) return rc;
            if ((rc = pj.add(c, c->vs, nvs, (size_t) w1 * h0))) return rc;
        }
        if ((rc = pj.upload(b->stream))) return rc;
        const size_t lds = (size_t) ((l - max_level + 1 + 31) / 32 + 1) * sizeof(unsigned);      // one bit per level of the session (the fused self-check)
        hipLaunchKernelGGL(k_inflate, dim3(h0, (unsigned) pj.dev.size()), dim3(256), lds, b->stream, pj.d_jobs, w0, w1, l, max_level, g_selfcheck ? g_dev_err : (int *) nullptr);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(b->stream));
        return check_dev_error();           // a failed level check: nothing is adopted, the host rolls the session back
    };
    rc = run();
    if (rc) { (void) hipStreamSynchronize(b->stream); discard_pending(b); }
    return rc;
}
extern "C" int lqrhip_planes_commit(LqrHipBatch *b)
{
    PendingInflate *pi = (PendingInflate *) b->pending_inflate;
    if (!pi) return LQRHIP_EARG;
    PlaneJobs &pj = pi->pj;
    pj.commit();
    for (auto &j : pj.jobs) { j.c->w0 = pi->w1; j.c->h0 = pi->h1; }
    if (pi->kind != 2) {                    // (a transpose keeps the carvers' -- all zero -- visibility maps)
        size_t i = 0;
        for (auto *c : b->cs) {
            dfree(c->vs);
            c->vs = pj.new_vs[i++];
            for (auto *a : c->aux) a->vs = c->vs;
        }
    }
    b->dirty = true;
    discard_pending(b);                     // (committed: the destructor frees only the job table)
    return 0;
}
extern "C" int lqrhip_inflate_commit(LqrHipBatch *b) { return lqrhip_planes_commit(b); }

extern "C" int lqrhip_flatten(LqrHipBatch *b, int w0, int h0, int w, int level)
{
    int rc;
    PlaneJobs &pj = new_pending(b, 1, w, h0)->pj;
    auto run = [&]() -> int {
        for (auto *c : b->cs) {
            int32_t *nvs = nullptr;                 // the flat carver's visibility map: all zero
            if ((rc = dmalloc(&nvs, (size_t) w * h0))) return rc;
            pj.new_vs.push_back(nvs);
            HIPCK(hipMemsetAsync(nvs, 0, (size_t) w * h0 * sizeof(int32_t), b->stream));
            for (auto *a : c->aux)
                if ((rc = pj.add(a, c->vs, nullptr, (size_t) w * h0))) return rc;
            if ((rc = pj.add(c, c->vs, nullptr, (size_t) w * h0))) return rc;
        }
        if ((rc = pj.upload(b->stream))) return rc;
        hipLaunchKernelGGL(k_compact_jobs, dim3(h0, (unsigned) pj.dev.size()), dim3(256), 0, b->stream, pj.d_jobs, w0, w, level);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(b->stream));
        return 0;
    };
    rc = run();
    if (rc) { (void) hipStreamSynchronize(b->stream); discard_pending(b); }
    return rc;
}

extern "C" int lqrhip_transpose(LqrHipBatch *b, int w, int h)
{
    int rc;
    PlaneJobs &pj = new_pending(b, 2, h, w)->pj;
    auto run = [&]() -> int {
        for (auto *c : b->cs) {
            for (auto *a : c->aux)
                if ((rc = pj.add(a, nullptr, nullptr, (size_t) w * h))) return rc;
            if ((rc = pj.add(c, nullptr, nullptr, (size_t) w * h))) return rc;
            HIPCK(hipMemsetAsync(c->vs, 0, (size_t) w * h * sizeof(int32_t), b->stream));   // flat carver: all zero already
        }
        if ((rc = pj.upload(b->stream))) return rc;
        hipLaunchKernelGGL(k_transpose, dim3((w + 31) / 32, (h + 31) / 32, (unsigned) pj.dev.size()), dim3(32, 8), 0, b->stream, pj.d_jobs, w, h);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(b->stream));
        return 0;
    };
    rc = run();
    if (rc) { (void) hipStreamSynchronize(b->stream); discard_pending(b); }
    return rc;
}

// ---- read-back ---------------------------------------------------------------
extern "C" int lqrhip_read_visible(LqrHipCarver *c, int w0, int h0, int w, int level, unsigned char *out)
{
    int rc = batch_sync_of(c);
    if (rc) return rc;
    uint8_t *d = nullptr;
    size_t n = (size_t) w * h0 * c->ch;
    if ((rc = dmalloc(&d, n))) return rc;
    auto run = [&]() -> int {
        hipLaunchKernelGGL(k_compact, dim3(h0), dim3(256), 0, g_stream0, c->rgb0, c->vs, (const float *) nullptr, (const float *) nullptr,
                           d, (float *) nullptr, (float *) nullptr, (int32_t *) nullptr, w0, w, c->ch, level, 0);
        HIPCK(hipGetLastError());
        return d2h_staged(out, d, n);
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);
    dfree(d);
    return rc;
}

extern "C" int lqrhip_read_visible_device(LqrHipCarver *c, int w0, int h0, int w, int level, void *device_out)
{
    int rc = batch_sync_of(c);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compact, dim3(h0), dim3(256), 0, g_stream0, c->rgb0, c->vs, (const float *) nullptr, (const float *) nullptr,
                       (uint8_t *) device_out, (float *) nullptr, (float *) nullptr, (int32_t *) nullptr, w0, w, c->ch, level, 0);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(g_stream0));
    return 0;
}

extern "C" int lqrhip_mask_line_max(const unsigned char *mask, int channels, int width, int height, int a0, int b0, int n_lines,
                                    int line_len, int direction)
{
    if (lqrhip_init() < 0) return LQRHIP_EHIP;
    if (n_lines <= 0 || line_len <= 0) return 0;
    uint8_t *d = nullptr;
    int *dout = nullptr;
    int rc, result = 0;
    size_t bytes = (size_t) width * height * channels;
    if ((rc = dmalloc(&d, bytes)) || (rc = dmalloc(&dout, 1))) { dfree(d); return rc; }
    auto run = [&]() -> int {
        HIPCK(hipMemcpyAsync(d, mask, bytes, hipMemcpyHostToDevice, g_stream0));
        HIPCK(hipMemsetAsync(dout, 0, sizeof(int), g_stream0));
        hipLaunchKernelGGL(k_mask_line_max, dim3(n_lines), dim3(256), 0, g_stream0, d, channels, width, a0, b0, line_len, direction, dout);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(&result, dout, sizeof(int), hipMemcpyDeviceToHost, g_stream0));
        HIPCK(hipStreamSynchronize(g_stream0));
        return 0;
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);
    dfree(d); dfree(dout);
    return rc ? rc : result;
}

extern "C" void lqrhip_pool_trim(void)
{
    (void) hipDeviceSynchronize();
    for (auto &kv : g_pool_free) { (void) hipFree(kv.second); g_pool_size.erase(kv.second); }
    g_pool_free.clear();
    g_pool_cached = 0;
}

extern "C" int lqrhip_device_sync(void)
{
    HIPCK(hipDeviceSynchronize());
    return check_dev_error();
}

extern "C" int lqrhip_read_vmap(LqrHipCarver *c, int w0, int h0, int w, int level, int depth, int *out)
{
    int rc = batch_sync_of(c);
    if (rc) return rc;
    int32_t *d = nullptr;
    size_t n = (size_t) w * h0;
    if ((rc = dmalloc(&d, n))) return rc;
    auto run = [&]() -> int {
        hipLaunchKernelGGL(k_compact, dim3(h0), dim3(256), 0, g_stream0, (const uint8_t *) nullptr, c->vs, (const float *) nullptr,
                           (const float *) nullptr, (uint8_t *) nullptr, (float *) nullptr, (float *) nullptr, d, w0, w, c->ch, level,
                           depth);
        HIPCK(hipGetLastError());
        return d2h_staged(out, d, n * sizeof(int32_t));
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);
    dfree(d);
    return rc;
}

extern "C" int lqrhip_read_working(LqrHipCarver *c, int w, int h, float *en, float *m, int *least_dx)
{
    int rc = batch_sync_of(c);
    if (rc) return rc;
    if (!c->pix) return LQRHIP_EARG;
    int32_t fl[FLAG_COUNT];
    HIPCK(hipMemcpy(fl, c->flags, sizeof fl, hipMemcpyDeviceToHost));
    const int org = fl[FLAG_ORG];                 // the carved planes start `org` elements into each row
    size_t n = (size_t) c->stride * h;
    std::vector<float> t(n);
    std::vector<int8_t> tl(n);
    if (en) {
        HIPCK(hipMemcpy(t.data(), c->en, n * sizeof(float), hipMemcpyDeviceToHost));
        for (int y = 0; y < h; y++) memcpy(en + (size_t) y * w, t.data() + (size_t) y * c->stride + org, (size_t) w * sizeof(float));
    }
    if (m) {
        HIPCK(hipMemcpy(t.data(), c->m, n * sizeof(float), hipMemcpyDeviceToHost));
        for (int y = 0; y < h; y++) memcpy(m + (size_t) y * w, t.data() + (size_t) y * c->stride + org, (size_t) w * sizeof(float));
    }
    if (least_dx) {
        HIPCK(hipMemcpy(tl.data(), c->least, n, hipMemcpyDeviceToHost));
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) least_dx[(size_t) y * w + x] = y == 0 ? 0 : (int) tl[(size_t) y * c->stride + org + x];
    }
    return 0;
}

// ---- start over from a device-resident image ---------------------------------------------------
__global__ __launch_bounds__(256) void k_copy16(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16);
extern "C" int lqrhip_carver_reset(LqrHipCarver *c, const void *device_rgb, int w, int h)
{
    if (!c || c->root || !c->aux.empty() || w < 1 || h < 1) return LQRHIP_EARG;
    int rc = batch_sync_of(c);
    if (rc) return rc;
    const size_t n = (size_t) w * h;
    dfree(c->rgb0); dfree(c->vs); dfree(c->bias0); dfree(c->rig0);
    // a carver that had masks carries bias / rig working planes: the fresh carver has none
    if (c->bias || c->rig) { free_working(c); c->stride = 0; c->wk_h = 0; }
    c->w0 = w; c->h0 = h;
    c->frozen_epoch = 0;
    if ((rc = dmalloc(&c->rgb0, n * c->ch)) || (rc = dmalloc(&c->vs, n))) return rc;
    // (round 6: these copies on four more streams side by side -- one 33 MB device-to-device copy runs at ~0.5 TB/s, 64 of them are 4 ms of a
    // 190-ms step -- made the 64-image step 45 % LONGER: with g_stream0 and the four sub-batch streams that is nine streams on the
    // process's eight hardware queues, and sub-batch streams that share a queue run one after the other.  One stream.)
    {
        // the runtime's device-to-device copy kernel moves a 33 MB image in 66 us (0.5 TB/s; 64 of them: 4.2 ms of a 190-ms step, one after
        // the other); the engine's own streaming copy (k_copy16, the one that measures the HBM ceiling) takes ~10
        const size_t bytes = n * c->ch, n16 = bytes / 16;
        if (n16 && !(((uintptr_t) device_rgb | (uintptr_t) c->rgb0) & 15)) {
            hipLaunchKernelGGL(k_copy16, dim3((unsigned) ((n16 + 255) / 256)), dim3(256), 0, g_stream0, (const u32x4 *) device_rgb, (u32x4 *) c->rgb0, n16);
            HIPCK(hipGetLastError());
            if (bytes & 15) HIPCK(hipMemcpyAsync(c->rgb0 + n16 * 16, (const uint8_t *) device_rgb + n16 * 16, bytes & 15, hipMemcpyDeviceToDevice, g_stream0));
        } else {
            HIPCK(hipMemcpyAsync(c->rgb0, device_rgb, bytes, hipMemcpyDeviceToDevice, g_stream0));
        }
    }
    HIPCK(hipMemsetAsync(c->vs, 0, n * sizeof(int32_t), g_stream0));
    if (c->batch) c->batch->dirty = true;
    if (c->active && (rc = ensure_working(c, w, h))) return rc;        // synchronises g_stream0 when it allocates
    return 0;
}

// order everything the shim enqueued on its own stream (resets, mask uploads) before the caller goes on
extern "C" int lqrhip_reset_sync(void)
{
    if (g_stream0) HIPCK(hipStreamSynchronize(g_stream0));
    return 0;
}

extern "C" int lqrhip_mem_info(unsigned long long *free_bytes, unsigned long long *total_bytes, unsigned long long *cached_bytes)
{
    if (lqrhip_init() < 0) return LQRHIP_EHIP;
    size_t f = 0, t = 0;
    HIPCK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    if (cached_bytes) *cached_bytes = g_pool_cached;
    return 0;
}

// ---- measured HBM ceiling: streaming copy, ONE 16-byte element per thread, non-temporal -- the form that measured
// fastest on this device (6.5 TB/s read + write at 2 GiB; grid-stride loops with 1-8 loads in flight per thread and
// 1k-64k workgroups: 4.4-5.8 TB/s; scripts/dbg/t_copy.hip)
__global__ __launch_bounds__(256) void k_copy16(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load((const GLOBAL_AS u32x4 *) src + i), (GLOBAL_AS u32x4 *) dst + i);
}

extern "C" int lqrhip_copy_bandwidth(unsigned long long bytes, int iters, double *gbps)
{
    if (lqrhip_init() < 0) return LQRHIP_EHIP;
    if (iters < 1 || bytes < 4096) return LQRHIP_EARG;
    uint8_t *a = nullptr, *b = nullptr;
    int rc;
    if ((rc = dmalloc(&a, bytes)) || (rc = dmalloc(&b, bytes))) { dfree(a); return rc; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    const size_t n16 = bytes / 16;
    auto run = [&]() -> int {
        HIPCK(hipMemsetAsync(a, 1, bytes, g_stream0));
        HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
        const dim3 grid((unsigned) ((n16 + 255) / 256));
        hipLaunchKernelGGL(k_copy16, grid, dim3(256), 0, g_stream0, (const u32x4 *) a, (u32x4 *) b, n16);     // warm-up
        HIPCK(hipEventRecord(e0, g_stream0));
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_copy16, grid, dim3(256), 0, g_stream0, (const u32x4 *) a, (u32x4 *) b, n16);
        HIPCK(hipEventRecord(e1, g_stream0));
        HIPCK(hipStreamSynchronize(g_stream0));
        HIPCK(hipEventElapsedTime(&ms, e0, e1));
        return 0;
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    dfree(a); dfree(b);
    if (rc) return rc;
    if (gbps) *gbps = 2.0 * (double) (n16 * 16) * iters / (ms * 1e-3) / 1e9;
    return 0;
}

// ---- seam-map colour ramp -----------------------------------------------------
// write_vmap_to_layer's per-pixel arithmetic, src/io_functions.c:249-279, in double with every
// operation individually rounded; (guchar)(255 * x) truncates.  One thread per pixel, streaming.
__global__ __launch_bounds__(256) void k_vmap_ramp(const int32_t *__restrict__ vmap, uint32_t *__restrict__ out, size_t n, int depth,
                                                   double sr, double sg, double sb, double er, double eg, double eb)
{
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int vs = vmap[i];
    uint32_t px = 0;                                                     // vs == 0: all four bytes 0 (:253-259)
    if (vs != 0) {
        const double value = __ddiv_rn((double) (depth + 1 - vs), (double) (depth + 1));        // :263
        const double inv = __dsub_rn(1.0, value);
        const double rd = __dadd_rn(__dmul_rn(value, sr), __dmul_rn(inv, er));                    // :264
        const double gr = __dadd_rn(__dmul_rn(value, sg), __dmul_rn(inv, eg));                    // :265
        const double bl = __dadd_rn(__dmul_rn(value, sb), __dmul_rn(inv, eb));                    // :266
        const double al = __dmul_rn(0.5, __dadd_rn(1.0, value));                                  // :267
        // (guchar) of a double: truncation towards zero, then the low 8 bits (values are in [0, 255])
        const uint32_t r8 = (uint32_t) (int) __dmul_rn(255.0, rd) & 0xffu, g8 = (uint32_t) (int) __dmul_rn(255.0, gr) & 0xffu;
        const uint32_t b8 = (uint32_t) (int) __dmul_rn(255.0, bl) & 0xffu, a8 = (uint32_t) (int) __dmul_rn(255.0, al) & 0xffu;
        px = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
    }
    out[i] = px;
}

extern "C" int lqrhip_vmap_to_rgba(const int *vmap, int w, int h, int depth, const double col_start[3], const double col_end[3],
                                   unsigned char *out_rgba)
{
    if (lqrhip_init() < 0) return LQRHIP_EHIP;
    if (!vmap || !out_rgba || w < 1 || h < 1) return LQRHIP_EARG;
    const size_t n = (size_t) w * h;
    int32_t *dv = nullptr;
    uint32_t *dout = nullptr;
    int rc;
    if ((rc = dmalloc(&dv, n)) || (rc = dmalloc(&dout, n))) { dfree(dv); return rc; }
    auto run = [&]() -> int {
        HIPCK(hipMemcpyAsync(dv, vmap, n * sizeof(int32_t), hipMemcpyHostToDevice, g_stream0));
        hipLaunchKernelGGL(k_vmap_ramp, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, g_stream0, dv, dout, n, depth, col_start[0], col_start[1],
                           col_start[2], col_end[0], col_end[1], col_end[2]);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(out_rgba, dout, n * 4, hipMemcpyDeviceToHost, g_stream0));
        HIPCK(hipStreamSynchronize(g_stream0));
        return 0;
    };
    rc = run();
    if (rc) (void) hipStreamSynchronize(g_stream0);      // nothing may still use the blocks when they go back to the pool
    dfree(dv); dfree(dout);
    return rc;
}
