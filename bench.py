#!/usr/bin/env python3
"""bench.py -- seam-carving throughput on MI355X (metric of BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (lqr_carver_resize: energy -> DP -> seam
pick/backtrack -> carve) over one batch of synthetic
images that are already resident in HBM.  Default workload = config 4 of
BASELINE.json ("batch of 64 independent 4K RGBA images, 200 seams each"): every
rank carves its own batch of 64 images of 3840x2160 RGBA by 200 vertical seams as
one lock-step batch; images are independent, so there is no data-path collective
and scaling is weak.  With N > 1 the same invocation afterwards also runs config 4
AS STATED -- 64 images in total, 64 / N per GPU -- and reports it as `strong` on the
same JSON line (--strong makes that the timed workload itself).
--workload single4k | fhd | 8k run configs 3 / 2 / 5's geometry; --workload config5
is config 5 as stated: 7680x4320 RGBA, preservation ellipse (+1000), discard band
(-1000) at x in [1500, 2100), rigidity 10, 1000 seams (--delta 2, --rigmask for its
variants; src/render.c:224-233,781-792).

Memory: ONE set of carvers per rank, whatever K and W are.  The input batch is
generated on the GPU once and stays in HBM; every step starts by re-loading the
carvers from it (lqrx_carver_reload_device_batch: a device-to-device copy, the
HBM-resident analogue of lqr_carver_new's buffer hand-over, src/render.c:222), adds
the masks where the workload has them (host buffers, as the plug-in hands them over,
src/io_functions.c:94-95,125-126) and resizes; all of that is INSIDE the timed region.

value = Mseams*px/s over ALL ranks = sum over phases (n_seams * W * H) * images
* K / wall time; wall time = max over ranks of the time of exactly K steps,
bracketed by barrier + device synchronize on both sides.

The three phases the reference brackets with __CLOCK_IT__ (src/render.c:214-217
read, :314-316 + :358-362 resize, :437-440 write) are reported under `phases`:
upload_ms (lqr_carver_new + lqr_carver_init of every image from HOST memory),
resize_ms (= ms_per_step), readout_ms (every result back to host memory), and
value_end_to_end is the metric over their sum -- the PCIe-inclusive figure; it is
never `value`.

Run with --gpus N > 1 and no RANK in the environment, the script re-executes
itself under torch.distributed.run with N ranks (one per GPU, RCCL).

Also on the JSON line:
  roofline      the dominant HBM kernel (k_carve), HIP-event timed per launch on its own stream(s) inside the timed
                region.  achieved = algorithmic bytes per launch (8 B x W*H/2 per image) / the average
                launch duration, or -- when smaller -- the bytes on the side of the seam the carve really moves (18 B per
                pixel, counted by k_vpath*: moved_bytes_per_launch); frac = achieved / 8 TB/s.  (frac_while_active divides by
                the union of the overlapping launches' intervals instead.)  traffic = HBM bytes per launch from the committed PMC passes of this
                command (profiles/pmc_kernels.json, scripts/profile_r05.sh).  kernels = EVERY kernel of the step from one
                extra untimed step with all of them timed: launches, average duration, share of the kernel time,
                algorithmic and PMC bytes, fraction of the roof.  end_to_end = the whole step against the roof:
                value x (4 B carve + 9 B x full DPs per phase / seams per phase) / 8 TB/s.
  configs       after the headline: config 4's literal per-GPU shard (batch4k_8img: 8 x 4K), BASELINE configs 2, 3 and 5
                (fhd, single4k, config5: one carver, the plug-in's own call shape) and 96 x 4K (batch4k_96img: what the same
                four streams carry when the group is larger), 3 steps each, with their own
                roofline.kernels, phases and cpu_baseline (--no-configs skips).  summary = every workload's value, last on the line.
  cpu_baseline  the CPU oracle (oracle/, a port of liblqr pinned against the genuine liblqr 0.4.1, oracle/REF_CHECK.md)
                timed on this host: one image on one core, and one image per core on all cores for the batch workload
                (nproc stated).
"""
import argparse
import ctypes as C
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Large batches run on 4 HIP streams (sub-batches: the latency-bound chain kernels of one under the bandwidth-bound carve
# of another, +10 %).  Each stream needs a hardware queue of its own and the HIP runtime's default is 4 per process,
# shared with torch's streams; with fewer queues than streams the split is slower than one stream, so the engine only
# makes it when this variable says there are 8 or more.  It must be in the environment before the first HIP call of the
# process, i.e. before torch is imported.  INTEGRATION.md says the same to a host application.
# (The engine library sets it itself when it is loaded before the runtime is up -- `lqrhip_on_load` -- which is the case here too;
# LQR_BENCH_NO_QUEUE_ENV=1 leaves it to the library, to show that.)
# 16, not 8: under torch.distributed RCCL's communicator brings streams of its own into the process, and with 8 hardware queues the four
# sub-batch streams then share queues with them -- measured at world size 1 under torchrun on one MI355X (round 6,
# profiles/r06/q_queues_under_torchrun.txt): 400 k with 8 queues, 575 - 582 k with 12 / 16 / 24; without RCCL 8 and 16 measure the same
# (578 / 576 k).
if not os.environ.get("LQR_BENCH_NO_QUEUE_ENV"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="batch4k", choices=["batch4k", "single4k", "fhd", "8k", "config5"])
    ap.add_argument("--images-per-gpu", type=int, default=64)
    ap.add_argument("--strong", action="store_true",
                    help="config 4 as stated: 64 images in total, 64 // N per GPU (scaling: strong)")
    ap.add_argument("--seams", type=int, default=None)
    ap.add_argument("--delta", type=int, default=1, help="delta_x (lqr_carver_init, render.c:224); the plug-in's UI offers up to 10")
    ap.add_argument("--rigidity", type=float, default=None, help="rigidity (default: 10 for config5, else 0)")
    ap.add_argument("--rigmask", action="store_true", help="config5 variant: rigidity mask over the top half (rigidity x 3, render.c:784-787)")
    ap.add_argument("--kernel-times", action="store_true",
                    help="HIP-event time every kernel of the seam loop (kernels_ms), not only k_carve; costs ~2.5 %% of the step")
    ap.add_argument("--sub-batches", type=int, default=0,
                    help="HIP streams the batch is split over (chain kernels of one sub-batch under the carve of another); "
                         "0 = the engine's choice: 4 for 32 images and more, given GPU_MAX_HW_QUEUES >= 8 (set above)")
    ap.add_argument("--update-mode", type=int, default=-1,
                    help="update_mmap kernel: -1 the engine's choice, 0 band (k_band_update_tw), 1 tiled full width, 2 band-mw, 3 generic, 5 k_band_levels")
    ap.add_argument("--band-levels", type=int, default=-1,
                    help="slots per image of k_band_levels (update mode 5): -1 the engine's choice, 0 never, n exactly n")
    ap.add_argument("--dp-px", type=int, default=0, help="pin the persistent tiled sweep's pixels per lane (2 or 4; 0: by batch size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-phases", action="store_true", help="skip the upload / read-out phase measurement")
    ap.add_argument("--vpath-mode", type=int, default=-1, help="backtrack: -1 the engine's choice (parallel k_vp_* up to 16 images), 0 k_vpath1 always, 1 parallel always")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--image-seed-offset", type=int, default=None, help="single process only: generate the images rank k of an N-rank run would (offset = k * images per GPU); tests/test_bench_multirank_gpu.py")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the extra legs after the headline: BASELINE configs 2, 3 and 5 (fhd, single4k, config5), 3 steps each")
    ap.add_argument("--no-kernel-breakdown", action="store_true", help="skip the extra untimed step that HIP-event times every kernel")
    ap.add_argument("--switch-freq", type=int, default=2, help="lqr side switch frequency (plug-in: 2, render.c:237)")
    return ap.parse_args()


WORKLOADS = {
    #            W     H    new_w  new_h
    "batch4k": (3840, 2160, 3640, 2160),      # config 4, per-GPU shard
    "single4k": (3840, 2160, 3340, 1660),     # config 3
    "fhd": (1920, 1080, 1720, 1080),          # config 2
    "8k": (7680, 4320, 6680, 4320),           # config 5's geometry, no masks, no rigidity
    "config5": (7680, 4320, 6680, 4320),      # config 5 as stated
}


def work_seam_px(w, h, nw, nh):
    """sum over phases of n_seams * W_start * H_start (HOR order)"""
    return abs(w - nw) * w * h + abs(h - nh) * nw * h


def alg_bytes_per_seam_px(w, h, nw, nh, switch_freq):
    """Algorithmic HBM bytes per seam*px of the whole step with the schedule the engine runs: the carve moves
    one 4-byte plane over half a row, read + write (4 B per seam*px), and a full DP (9 B/px) runs once at the start of a
    phase and once after each of the `switch_freq` side switches."""
    work, by = 0.0, 0.0
    for n, ww, hh in ((abs(w - nw), w, h), (abs(h - nh), nw, h)):
        if n:
            work += n * ww * hh
            by += 4.0 * n * ww * hh + 9.0 * ww * hh * (1 + min(switch_freq, max(n - 1, 0)))
    return by / work if work else 0.0


def make_images(n, w, h, seed, device):
    """photo-like synthetic RGBA batch (low-frequency structure + 1/f octave noise, u8, alpha=255),
    same recipe as tests/datasets.photo_like, generated with torch on `device`; returns u8 [n, h, w, 4]"""
    import torch
    import torch.nn.functional as F
    out_all = torch.empty((n, h, w, 4), dtype=torch.uint8, device=device)
    yy = torch.arange(h, dtype=torch.float32, device=device)[:, None]
    xx = torch.arange(w, dtype=torch.float32, device=device)[None, :]
    for i in range(n):
        g = torch.Generator(device="cpu").manual_seed(int(seed) + i)
        base = torch.zeros(h, w, device=device)
        for _ in range(4):
            f = (torch.rand(2, generator=g) * 3.5 + 0.5) * 6.2831853
            ph = torch.rand(1, generator=g) * 6.2831853
            amp = torch.rand(1, generator=g) * 0.7 + 0.3
            base += float(amp) * torch.sin(float(f[0]) * xx / w + float(f[1]) * yy / h + float(ph))
        out = base[None].repeat(3, 1, 1)
        m, k = 4, 0
        while m < max(w, h):
            gh, gw = max(2, m * h // max(w, h) + 1), max(2, m * w // max(w, h) + 1)
            grid = torch.randn(1, 3, gh, gw, generator=g).to(device)
            out += F.interpolate(grid, size=(h, w), mode="bilinear", align_corners=True)[0] * (0.9 / (k + 1))
            m *= 2
            k += 1
        out -= out.min()
        out *= 255.0 / max(float(out.max()), 1e-6)
        out_all[i, :, :, :3] = out.round().clamp(0, 255).to(torch.uint8).permute(1, 2, 0)
        out_all[i, :, :, 3] = 255
    return out_all


def config5_masks(w, h, rigmask):
    """config 5's layers, RGBA u8 as the plug-in reads them from GIMP layers: a filled
    white ellipse at the centre covering a quarter of the image (preservation), a white band x in [1500, 2100)
    (discard), and -- variant -- the top half (rigidity mask); same shapes as tests/datasets.py"""
    import numpy as np
    yy = (np.arange(h, dtype=np.float32)[:, None] - (h - 1) / 2) / (h / 2)
    xx = (np.arange(w, dtype=np.float32)[None, :] - (w - 1) / 2) / (w / 2)
    pres = np.zeros((h, w, 4), np.uint8)
    pres[(xx * xx + yy * yy) <= 0.25 * 4 / np.pi] = 255
    disc = np.zeros((h, w, 4), np.uint8)
    disc[:, 1500 * w // 7680:2100 * w // 7680] = 255
    rig = None
    if rigmask:
        rig = np.zeros((h, w, 4), np.uint8)
        rig[: h // 2] = 255
    return pres, disc, rig


# the engine's profiling names -> the kernels behind them (large batch, small batch / single image): for the PMC look-up
KERNEL_NAMES = {
    "carve": (["k_carve"], ["k_carve"]),
    "vpath": (["k_vpath1", "k_vpath"], ["k_vp_maps", "k_vpath1", "k_vpath"]),      # (small groups: k_vp_maps + k_vp_solve + k_vp_paths, one scope)
    "band_update": (["k_band_update_tw", "k_band_update_mw", "k_band_update"], ["k_band_update_tw", "k_band_update"]),
    "band_levels": (["k_band_levels"], ["k_band_levels"]),
    "dp_update": (["k_dp_sweep"], ["k_dp_sweep"]),
    "dp_update_tiled": (["k_dp_tile_p"], ["k_dp_tile_p"]),
    "dp_sweep": (["k_dp_tile", "k_dp_sweep"], ["k_dp_tile_p", "k_dp_tile", "k_dp_sweep"]),
    "emap_update": (["k_emap_update"], ["k_emap_update"]),
}
_PMC = None


def pmc_traffic(workload, kernels, nimg, streams):
    """HBM bytes per launch of the first of `kernels` found in profiles/pmc_kernels.json (rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE in separate passes over this command, (2 * FETCH + WRITE) * 1024 per MI355X_MICROARCH.md), scaled from the
    profiled images per launch to this run's; None when there is no committed profile for this workload"""
    global _PMC
    if _PMC is None:
        path = os.path.join(ROOT, "profiles", "pmc_kernels.json")
        _PMC = json.load(open(path)) if os.path.exists(path) else {}
    wl = _PMC.get(workload)
    if not wl:
        return None
    for k in kernels:
        if k in wl["kernels"]:
            return round(wl["kernels"][k]["traffic_bytes_per_launch"] / wl["images_per_launch"] * nimg / max(streams, 1))
    return None


def spawn_command(gpus, argv):
    """the launcher line the driver uses for N > 1: one rank per GPU of one node, rendezvous on 127.0.0.1"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_spawn(args):
    """--gpus N without a launcher: become N ranks under torch.distributed.run (one per GPU)"""
    cmd = spawn_command(args.gpus, sys.argv[1:])
    os.execvp(cmd[0], cmd)


def strong_images_per_gpu(world, total=64):
    """config 4 as stated: `total` images over `world` GPUs"""
    return max(1, total // max(world, 1))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        self_spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; using the launcher's world size" % (args.gpus, world), file=sys.stderr)

    # torch first: its bundled HIP runtime (same soname) must be the one runtime of the process;
    # the engine then binds to it and pins itself to device LOCAL_RANK
    import numpy as np
    import torch
    import __graft_entry__ as ge
    ge._import_package()
    from gimp_lqr_plugin_amd import binding as L
    eng = L.engine_api()
    lib = eng.lib
    lib.lqrhip_last_error.restype = C.c_char_p
    if lib.lqrhip_init() < 0:
        raise SystemExit("bench.py: no usable HIP device: %s" % lib.lqrhip_last_error().decode())
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: torch sees no GPU")
    lib.lqrhip_set_sub_batches.argtypes = [C.c_int]
    lib.lqrhip_set_sub_batches(args.sub_batches)
    lib.lqrhip_sub_batches.argtypes = [C.c_int]
    lib.lqrhip_set_update_mode.argtypes = [C.c_int]
    lib.lqrhip_set_update_mode(args.update_mode)
    lib.lqrhip_set_band_levels.argtypes = [C.c_int]
    lib.lqrhip_set_band_levels(args.band_levels)
    if os.environ.get("LQR_SWEEP_THREADS"):   # A/B switch: 1024 = the k_dp_sweep<UPDATE> launch behind the band kernels as in rounds 1 - 5
        lib.lqrhip_set_sweep_threads.argtypes = [C.c_int]
        lib.lqrhip_set_sweep_threads(int(os.environ["LQR_SWEEP_THREADS"]))
    if os.environ.get("LQR_NO_FUSE") or os.environ.get("LQR_FUSE_MAX"):        # A/B switches: two kernels also for small groups / one launch up to n images
        lib.lqrhip_set_carve_fused.argtypes = [C.c_int]
        lib.lqrhip_set_carve_fused(int(os.environ.get("LQR_FUSE_MAX", "0")))
    lib.lqrhip_set_vpath_mode.argtypes = [C.c_int, C.c_int]
    lib.lqrhip_set_vpath_mode(args.vpath_mode, 0)
    if os.environ.get("LQR_LV_DBG"):
        lib.lqrhip_band_levels_debug.argtypes = [C.c_int]
        lib.lqrhip_band_levels_debug(int(os.environ["LQR_LV_DBG"]))
    if os.environ.get("LQR_DPP_DBG"):
        lib.lqrhip_dp_tile_debug.argtypes = [C.c_int]
        lib.lqrhip_dp_tile_debug(int(os.environ["LQR_DPP_DBG"]))
    if args.dp_px:
        lib.lqrhip_set_dp_persistent_px.argtypes = [C.c_int]
        lib.lqrhip_set_dp_persistent_px(args.dp_px)
    lib.lqrhip_moved_bytes.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    # LQR_BENCH_DIST_BACKEND=gloo: the N > 1 control flow (barriers, max over ranks, the strong leg, the gather) on a box with ONE
    # GPU -- ranks share device LOCAL_RANK mod device count and the collectives run on host tensors.  Not a measurement: it exists so
    # that this code is not executed for the first time on the day an 8-GPU node appears (tests/test_bench_multirank_gpu.py).
    backend = os.environ.get("LQR_BENCH_DIST_BACKEND", "nccl")
    local_dev = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")        # where collective operands live

    dist = None
    if world > 1 or "RANK" in os.environ:       # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    def measure(wl, steps, warmup, headline, n_images=None, delta=None):
        """one workload: set-up, `warmup` untimed + exactly `steps` timed steps, per-kernel breakdown, phases, CPU baseline;
        `headline`: also the gather / strong / all-cores legs; `n_images`: images per GPU of a batch workload other than the
        command line's.  Returns the JSON object of the workload."""
        W, H, NW, NH = WORKLOADS[wl]
        dlt = delta if delta is not None else args.delta
        batch = wl == "batch4k"
        nimg = (n_images or args.images_per_gpu) if batch else 1
        if batch and args.strong and not n_images:
            nimg = strong_images_per_gpu(world)
        if args.seams is not None:
            NW = W - args.seams
        rigidity = args.rigidity if args.rigidity is not None else (10.0 if wl == "config5" else 0.0)
        pres = disc = rigm = None
        if wl == "config5":
            pres, disc, rigm = config5_masks(W, H, args.rigmask)
            if rigm is not None:
                rigidity *= 3           # render.c:784-787

        def sync():
            rc = lib.lqrhip_device_sync()
            assert rc == 0, "device error: %s" % lib.lqrhip_last_error().decode()
            torch.cuda.synchronize()

        def barrier():
            if dist is not None:
                dist.barrier()

        def mem_used_gb():
            f, t, c = C.c_ulonglong(0), C.c_ulonglong(0), C.c_ulonglong(0)
            lib.lqrhip_mem_info(C.byref(f), C.byref(t), C.byref(c))
            return (t.value - f.value) / 1e9, t.value / 1e9

        def max_over_ranks(x):
            if dist is None:
                return x
            t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        # ---- phase "read" (render.c:214-217, 220-224): inputs generated on the GPU and kept in HBM for the timed steps; the
        # carvers are created from HOST copies of them, as the plug-in creates its carver from a host buffer -- that is the
        # measured upload phase (lqr_carver_new uploads, lqr_carver_init allocates the working planes)
        t_setup = time.perf_counter()
        images = make_images(nimg, W, H, 100 + (rank * nimg if args.image_seed_offset is None else args.image_seed_offset), dev)
        torch.cuda.synchronize()
        host_imgs = [images[i].cpu().numpy() for i in range(nimg)]
        ptrs = [images[i].data_ptr() for i in range(nimg)]
        bufs = [L._malloc_copy(im) for im in host_imgs]      # liblqr takes ownership of a malloc'ed buffer (render.c:222)
        sync()
        tu = time.perf_counter()
        carvers = [L.Carver.from_buffer(eng, bufs[i], W, H, 4, delta_x=dlt, rigidity=rigidity) for i in range(nimg)]
        sync()
        upload_ms = (time.perf_counter() - tu) * 1e3
        for c in carvers:
            c.configure(switch_freq=args.switch_freq, enl_step=1.5)                          # plug-in defaults, main.c:62-87
        img0_host = host_imgs[0]
        host_keep = host_imgs      # for the all-cores CPU baseline and the warm upload leg
        sync()
        t_setup = time.perf_counter() - t_setup

        def add_masks(cs):
            # render.c:225-233 (pres_coeff / disc_coeff defaults 1000, main.c:62-87)
            for c in cs:
                if pres is not None:
                    assert c.bias_add(pres, 1000) == L.LQR_OK
                if disc is not None:
                    assert c.bias_add(disc, -1000) == L.LQR_OK
                if rigm is not None:
                    assert c.rigmask_add(rigm) == L.LQR_OK

        def run_step(cs, ps):
            ret = L.reload_device_batch(eng, cs, ps)
            assert ret == L.LQR_OK, "reload failed: %d (%s)" % (ret, lib.lqrhip_last_error().decode())
            add_masks(cs)
            if len(cs) == 1:
                ret = cs[0].resize(NW, NH)
            else:
                ret = L.resize_batch(eng, cs, NW, NH)
            assert ret == L.LQR_OK, "resize failed: %d (%s)" % (ret, lib.lqrhip_last_error().decode())

        def timed(cs, ps, steps, warmup, prof_mode):
            for _ in range(warmup):
                run_step(cs, ps)
            lib.lqrhip_prof_reset()
            lib.lqrhip_prof_enable(prof_mode)
            barrier(); sync()
            lib.lqrhip_moved_bytes(C.byref(C.c_ulonglong(0)), 1)         # what the carves of the timed steps HAVE to move: from zero
            t0 = time.perf_counter()
            for _ in range(steps):
                run_step(cs, ps)
            sync(); barrier()
            t1 = time.perf_counter()
            lib.lqrhip_prof_enable(0)
            mv = C.c_ulonglong(0)
            lib.lqrhip_moved_bytes(C.byref(mv), 0)
            timed.moved_bytes = mv.value
            return max_over_ranks(t1 - t0)

        # ---- phase "resize" (render.c:314-316): W warm-up steps, then exactly K timed steps
        elapsed = timed(carvers, ptrs, steps, warmup, 1 if args.kernel_times else 2)
        moved_total = timed.moved_bytes            # bytes the timed steps' carves had to move (k_vpath*'s count of the side that moves)
        used_gb, total_gb = mem_used_gb()
        streams = lib.lqrhip_sub_batches(nimg) if nimg > 1 else 1

        # ---- per-kernel HIP-event times collected inside the timed region
        def prof(name):
            ms, n, by = C.c_double(0), C.c_longlong(0), C.c_double(0)
            lib.lqrhip_prof_get(name.encode(), C.byref(ms), C.byref(n), C.byref(by))
            return ms.value, n.value, by.value
        kern = {k: prof(k) for k in KERNEL_NAMES}
        c_ms, c_n, c_bytes = kern["carve"]
        work_rank = work_seam_px(W, H, NW, NH) * nimg * steps       # seam*px per rank
        value = work_rank * world / elapsed / 1e6
        roofline = None
        if c_n:
            un = C.c_double(0)
            lib.lqrhip_prof_get_union.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
            lib.lqrhip_prof_get_union(b"carve", C.byref(un))
            active_ms = un.value if un.value > 0 else c_ms
            avg_us = c_ms * 1e3 / c_n
            alg_launch = c_bytes / c_n
            # The numerator of a roofline fraction must be bytes the kernel could not avoid.  SURVEY 8(d) prices the carve at one
            # 4-byte plane over HALF a row, read + write (alg); the engine moves the SHORTER side of the seam, three planes
            # (en, m, back pointer: 18 B per pixel moved, read + write; 8 B on the seams a full DP follows) -- `moved` is that,
            # counted by k_vpath* for the seams it actually found.  frac takes the smaller of the two.
            moved_launch = moved_total / c_n if moved_total else None
            num_launch = min(alg_launch, moved_launch) if moved_launch else alg_launch
            achieved = num_launch / (avg_us * 1e-6) / 1e9                    # bytes per launch / average launch duration
            while_active = num_launch * c_n / (active_ms * 1e-3) / 1e9       # ... / the time during which at least one carve was running
            # HBM traffic per launch from the committed rocprofv3 PMC passes of this command (separate FETCH_SIZE / WRITE_SIZE
            # runs; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note; scripts/profile_r05.sh), null if missing
            traffic = pmc_traffic(wl if not n_images else "%s_%dimg" % (wl, nimg), ["k_carve"], nimg, streams)
            b_alg = alg_bytes_per_seam_px(W, H, NW, NH, args.switch_freq)
            roofline = {"bound": "hbm", "kernel": "k_carve", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                        "avg_launch_us": round(avg_us, 2), "launches": c_n, "streams": streams,
                        "alg_bytes_per_launch": round(alg_launch),
                        "moved_bytes_per_launch": None if moved_launch is None else round(moved_launch),
                        "frac_basis": "moved_bytes" if (moved_launch and moved_launch < alg_launch) else "alg_bytes",
                        "frac_alg": round(alg_launch / (avg_us * 1e-6) / 8e12, 4),
                        "traffic_ratio": None if not (traffic and moved_launch) else round(traffic / moved_launch, 3),
                        "achieved_while_active": round(while_active, 1), "frac_while_active": round(while_active / 8000.0, 4),
                        "carve_active_ms": round(active_ms, 3), "sum_of_launches_ms": round(c_ms, 3),
                        "note": "frac = min(alg_bytes, moved_bytes) per launch / avg_launch_us / peak for k_carve, the HBM-bound kernel: alg = SURVEY 8(d)'s "
                                "8 B x W*H/2 per image, moved = 18 B x the pixels on the side of the seam the carve really moves (k_vpath*'s count; "
                                "8 B on seams a full DP follows); traffic_ratio = PMC traffic / moved; frac_alg = the alg-only figure of earlier "
                                "rounds; launches of the sub-batch streams overlap, frac_while_active divides by the union of their intervals, `alone` is the same kernel on one stream with nothing beside it; "
                                "`kernels` lists every kernel of the step, `end_to_end` is the whole step against the roof",
                        "end_to_end": {"bytes_per_seam_px": round(b_alg, 4), "achieved": round(value * b_alg * 1e-3 / max(world, 1), 1),
                                       "unit": "GB/s per GPU", "frac": round(value * b_alg * 1e-3 / max(world, 1) / 8000.0, 4)}}
            if rank == 0 and headline:
                # this device's own ceiling: 16-B streaming copy of 2 GiB (read + write), outside the timed region
                g = C.c_double(0)
                lib.lqrhip_copy_bandwidth.argtypes = [C.c_ulonglong, C.c_int, C.POINTER(C.c_double)]
                if lib.lqrhip_copy_bandwidth(2 << 30, 10, C.byref(g)) == 0 and g.value > 0:
                    roofline["measured_copy_peak"] = round(g.value, 1)
                    roofline["frac_of_measured"] = round(achieved / g.value, 4)

        # ---- where the step's time goes: ONE extra, untimed step with every kernel of the seam loop HIP-event timed (the
        # events cost queue time, which is why the timed region times k_carve only)
        if not args.no_kernel_breakdown and not args.kernel_times:
            timed(carvers, ptrs, 1, 0, 1)
            kern = {k: prof(k) for k in KERNEL_NAMES}
        tot_ms = sum(v[0] for v in kern.values()) or 1.0
        # ---- the same kernel with the device to itself: ONE more untimed step on ONE stream (every image of the group in one
        # launch, no sibling stream's kernels beside it).  In the timed region the launches of the sub-batch streams overlap each
        # other and the other streams' band kernels, and avg_launch_us is a contended figure; this is the kernel's own.
        if roofline is not None and headline and streams > 1 and not args.no_kernel_breakdown and not args.kernel_times:
            lib.lqrhip_set_sub_batches(1)
            timed(carvers, ptrs, 1, 0, 2)
            lib.lqrhip_set_sub_batches(args.sub_batches)
            a_ms, a_n, a_bytes = prof("carve")
            if a_n:
                a_num = min(a_bytes / a_n, timed.moved_bytes / a_n) if timed.moved_bytes else a_bytes / a_n
                roofline["alone"] = {"streams": 1, "launches": a_n, "avg_launch_us": round(a_ms * 1e3 / a_n, 2), "bytes_per_launch": round(a_num),
                                     "achieved": round(a_num / (a_ms * 1e-3 / a_n) / 1e9, 1), "frac": round(a_num / (a_ms * 1e-3 / a_n) / 8e12, 4),
                                     "note": "k_carve over the whole group in one launch per seam, one stream, nothing beside it (an extra untimed step)"}
        if roofline is not None:
            roofline["kernels"] = []
            for k, (ms, n, by) in sorted(kern.items(), key=lambda kv: -kv[1][0]):
                if not n:
                    continue
                cand = KERNEL_NAMES[k][0 if nimg > 8 else 1]
                pmc = pmc_traffic(wl if not n_images else "%s_%dimg" % (wl, nimg), cand, nimg, streams)
                alg = by / n if by else None
                if k == "carve" and alg and timed.moved_bytes and n:
                    alg = min(alg, timed.moved_bytes / n)            # bytes it could not avoid (see roofline.note)
                basis = alg if alg else pmc
                roofline["kernels"].append({
                    "name": k, "kernel": cand[0], "launches": n, "avg_us": round(ms * 1e3 / n, 2), "share_of_kernel_time": round(ms / tot_ms, 4),
                    "alg_bytes": None if alg is None else round(alg), "pmc_bytes": pmc,
                    "frac": None if not basis else round(basis / (ms * 1e-3 / n) / 8e12, 4),
                    "frac_basis": None if not basis else ("alg_bytes" if alg else "pmc_bytes")})

        # ---- phase "write" (render.c:358-362, io_functions.c:134-182): every result of the last step back to host memory
        # (device compaction of the visible pixels + D2H; the scan-line loop of io_functions.c:155-164 is served from that
        # host copy), then -- N > 1 -- the gather to rank 0 over RCCL
        readout_ms = None
        if not args.no_phases:
            sync()
            tr = time.perf_counter()
            host_out = [c.read_image() for c in carvers]
            readout_ms = max_over_ranks((time.perf_counter() - tr) * 1e3)
            assert host_out[0].shape == (NH, NW, 4)
            del host_out
            upload_ms = max_over_ranks(upload_ms)
        gather_ms = gathered_checksums = None
        outs = torch.empty((nimg, NH, NW, 4), dtype=torch.uint8, device=dev)
        if carvers[0].getters()["orientation"] == 0:
            for i, c in enumerate(carvers):
                assert eng.lqrx_carver_read_image_device(c.p, outs[i].data_ptr()) == L.LQR_OK
        else:       # transposed carver frame: go through the host image-orientation read-out
            for i, c in enumerate(carvers):
                outs[i].copy_(torch.from_numpy(c.read_image()))
        if dist is not None and not args.no_gather:
            sync(); barrier()
            tg = time.perf_counter()
            src = outs if backend == "nccl" else outs.cpu()
            gathered = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
            dist.gather(src, gathered, dst=0)
            sync()
            gather_ms = (time.perf_counter() - tg) * 1e3
            gathered_checksums = [int(g.to(torch.int64).sum().item()) for g in gathered] if rank == 0 else None       # what arrived from each rank
            del gathered, src
        checksum = int(outs.to(torch.int64).sum().item())
        g = carvers[0].getters()
        assert (g["width"], g["height"]) == (NW, NH), g

        ms_per_step = elapsed * 1e3 / steps
        mode = "strong" if (batch and args.strong) else "weak"
        variant = ""
        if wl == "config5":
            variant = ", preservation ellipse +1000, discard band -1000, rigidity %g, delta_x %d%s" % (
                rigidity, dlt, ", rigidity mask (top half)" if rigm is not None else "")
        elif dlt != 1 or rigidity:
            variant = ", rigidity %g, delta_x %d" % (rigidity, dlt)
        result = {
            "metric": "Mseams*pixels/sec on 4K RGBA", "value": round(value, 1), "unit": "Mseams*px/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": mode,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d x %dx%d RGBA per GPU, resize to %dx%d (%d vertical%s seams each), side-switch %d%s" % (
                           wl, nimg, W, H, NW, NH, W - NW, (" + %d horizontal" % (H - NH)) if NH != H else "",
                           args.switch_freq, variant),
                       "images_per_gpu": nimg, "width": W, "height": H, "new_width": NW, "new_height": NH,
                       "parallelism": "images sharded i mod N, no data-path collective", "streams_per_gpu": streams},
            "roofline": roofline,
            "kernels_ms": {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in kern.items() if v[1]},
            "gather_ms": None if gather_ms is None else round(gather_ms, 2),
            "gathered_checksums": gathered_checksums if gather_ms is not None else None,
            "setup_s": round(t_setup, 2),
            "hbm_used_gb": round(used_gb, 1), "hbm_total_gb": round(total_gb, 1),
            "output_checksum": checksum,
        }
        if readout_ms is not None:
            e2e_ms = upload_ms + ms_per_step + readout_ms
            result["phases"] = {
                "upload_ms": round(upload_ms, 2), "resize_ms": round(ms_per_step, 3), "readout_ms": round(readout_ms, 2),
                "value_end_to_end": round(work_rank / steps * world / (e2e_ms * 1e-3) / 1e6, 1),
                "note": "upload = lqr_carver_new + lqr_carver_init of every image from pageable host memory (render.c:222-224), "
                        "readout = every result into host memory (io_functions.c:134-182); value_end_to_end = the metric over "
                        "upload + resize + readout; `value` itself is the HBM-resident rate"}

        # ---- config 4 as stated, on the same line when N > 1: 64 images in total = 64 // N per GPU (strong scaling)
        if headline and batch and world > 1 and not args.strong:
            ns = min(strong_images_per_gpu(world), nimg)
            el = timed(carvers[:ns], ptrs[:ns], steps, 1, 0)
            result["strong"] = {"images_total": ns * world, "images_per_gpu": ns, "scaling": "strong",
                                "ms_per_step": round(el * 1e3 / steps, 3),
                                "value": round(work_seam_px(W, H, NW, NH) * ns * steps * world / el / 1e6, 1),
                                "streams_per_gpu": lib.lqrhip_sub_batches(ns) if ns > 1 else 1}

        # ---- CPU baseline: the oracle (a port of the algorithm; test infrastructure, loaded here only as the
        # reported baseline and the spot checker) on this host's cores
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from concurrent.futures import ThreadPoolExecutor
            orc = L.Api(os.path.join(ROOT, "oracle", "liblqr_oracle.so"), "o")
            ncores = os.cpu_count() or 1
            cw, chh, cnw, cnh = W, H, NW, NH
            sample = "1 image of the workload (%dx%d -> %dx%d), 1 core" % (cw, chh, cnw, cnh)
            if wl in ("single4k", "8k", "config5"):     # bound the sample: 100 (+100) seams instead of 500+500 / 1000
                cnw, cnh = W - 100, (H - 100 if NH != H else H)
                sample = "1 image %dx%d -> %dx%d (first %d seams of the workload), 1 core" % (cw, chh, cnw, cnh, (W - cnw) + (H - cnh))

            def oracle_carver(im):
                o = L.Carver(orc, im, delta_x=dlt, rigidity=rigidity).configure(switch_freq=args.switch_freq, enl_step=1.5)
                add_masks([o])
                return o
            oc = oracle_carver(img0_host)
            tc = time.perf_counter()
            assert oc.resize(cnw, cnh) == L.LQR_OK
            tc = time.perf_counter() - tc
            cpu_val = work_seam_px(cw, chh, cnw, cnh) / tc / 1e6
            result["cpu_baseline"] = {"value": round(cpu_val, 1), "unit": "Mseams*px/s", "cores": 1, "kind": "port",
                                      "sample": sample, "seconds": round(tc, 2), "nproc": ncores, "cpu": cpu_model()}
            # parity spot check of the timed workload's first image against the oracle
            if (cnw, cnh) == (NW, NH):
                ref = oc.read_image()
                got = carvers[0].read_image()
                result["parity_vs_oracle"] = bool(np.array_equal(ref, got))
            oc.destroy()
            if nimg > 1 and headline:
                # for the batch, one image per core over all host cores (liblqr itself is single-threaded;
                # ctypes releases the GIL inside the C call, so threads run the oracle truly in parallel)
                nt = min(ncores, nimg, len(host_keep))

                def one(im):
                    o = oracle_carver(im)
                    r = o.resize(NW, NH)
                    o.destroy()
                    return r
                ta = time.perf_counter()
                with ThreadPoolExecutor(max_workers=nt) as ex:
                    rets = list(ex.map(one, host_keep[:nt]))
                ta = time.perf_counter() - ta
                assert all(r == L.LQR_OK for r in rets)
                result["cpu_baseline"]["all_cores"] = {
                    "value": round(work_seam_px(W, H, NW, NH) * nt / ta / 1e6, 1), "unit": "Mseams*px/s", "cores": nt,
                    "sample": "%d images of the workload, one per core, concurrently" % nt, "seconds": round(ta, 2)}

        for c in carvers:
            c.destroy()
        if "phases" in result:
            # the upload phase again with the device blocks of the carvers just destroyed recycled by the engine's block cache
            # (a host that carves image after image; the first measurement above also pays hipMalloc for 12 GB)
            bufs = [L._malloc_copy(im) for im in host_keep]
            sync()
            tu = time.perf_counter()
            cs2 = [L.Carver.from_buffer(eng, bufs[i], W, H, 4, delta_x=dlt, rigidity=rigidity) for i in range(nimg)]
            sync()
            warm = max_over_ranks((time.perf_counter() - tu) * 1e3)
            for c in cs2:
                c.destroy()
            ph = result["phases"]
            ph["upload_cold_ms"] = ph["upload_ms"]
            ph["upload_ms"] = round(warm, 2)
            ph["value_end_to_end_cold"] = ph["value_end_to_end"]
            ph["value_end_to_end"] = round(work_rank / steps * world / ((warm + ms_per_step + ph["readout_ms"]) * 1e-3) / 1e6, 1)
            ph["note"] += "; upload_cold_ms = the first creation in the process (hipMalloc of every block), upload_ms = the same calls with the engine's block cache warm, as the timed steps are"
        del images, outs
        torch.cuda.empty_cache()
        return result

    result = measure(args.workload, args.steps, args.warmup, True)
    if hasattr(lib, "lqrhip_fault_stats"):        # a bench line made with a redone session says so
        st = (C.c_ulonglong * 8)()
        if lib.lqrhip_fault_stats(st, 0) == 0 and any(st[i] for i in range(7)):
            result["fault_stats"] = {"spin_timeouts": st[0], "failed_predictions": st[1], "seam_log_check_failures": st[2], "level_check_failures": st[3],
                                     "sessions_rolled_back": st[4], "sessions_redone_without_spin_kernels": st[6]}
    if hasattr(lib, "lqrhip_band_levels_stats"):
        st = (C.c_ulonglong * 8)()
        if lib.lqrhip_band_levels_stats(st, 1) == 0 and any(st[i] for i in range(4)):
            result["band_levels_stats"] = {"images_stopped_by_a_collision": st[0], "synchronous_loads": st[1], "tile_levels_processed": st[2], "slot_levels_idle": st[3]}
    # ---- on the same line: config 4's LITERAL per-GPU shard (64 images over 8 GPUs = 8 per GPU: the strong-scaling end nobody
    # can measure without the node), then BASELINE configs 2, 3 and 5 -- the plug-in's own call shape, one carver (render.c:318)
    if args.workload == "batch4k" and world == 1 and not args.no_configs and args.seams is None and not args.strong:
        result["configs"] = {}
        keep = ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "kernels_ms", "phases", "cpu_baseline", "parity_vs_oracle", "hbm_used_gb")
        # (batch4k_96img: the headline's 64 images are four chains of 16 in flight and the chain's latency, not the chip, bounds
        # them -- half as many images again run through the same four streams in a fifth more time: DESIGN.md 4.11 / 9)
        # (round 6: delta_x 10 / 8 -- the upper half of the plug-in's dialog range, src/interface.c:47 -- on the tiled kernels' general instantiations)
        legs = [("batch4k_8img", "batch4k", 8, None, 3), ("fhd", "fhd", None, None, 3), ("single4k", "single4k", None, None, 3), ("config5", "config5", None, None, 3),
                ("batch4k_96img", "batch4k", 96, None, 3), ("single4k_delta10", "single4k", None, 10, 1), ("batch4k_16img_delta8", "batch4k", 16, 8, 1)]
        for name, wl, images, leg_delta, leg_steps in legs:
            if images and images == args.images_per_gpu and leg_delta is None:
                continue
            if leg_delta is not None and args.delta != 1:
                continue
            r = measure(wl, leg_steps, 1, False, n_images=images, delta=leg_delta)
            for rk in (r.get("roofline") or {}, r.get("phases") or {}):
                rk.pop("note", None)                      # said once, on the headline
            result["configs"][name] = {k: r[k] for k in keep if k in r}
        # the numbers a reader (or a log tail) wants first, LAST on the line: Mseams*px/s per workload, k_carve's fraction of the roof
        result["summary"] = dict({"batch4k_%dimg" % args.images_per_gpu: result["value"], "k_carve_frac": (result.get("roofline") or {}).get("frac"),
                                  "end_to_end_frac": ((result.get("roofline") or {}).get("end_to_end") or {}).get("frac")},
                                 **{k: v["value"] for k, v in result["configs"].items()})
    if dist is not None:
        result["rccl_ranks"] = dist.get_world_size()
        result["dist_backend"] = backend
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
