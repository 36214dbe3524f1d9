#!/usr/bin/env python3
"""bench.py -- seam-carving throughput on MI355X (metric of BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (lqr_carver_resize: energy -> DP -> seam
pick/backtrack -> carve) over one batch of synthetic
images that are already resident in HBM.  Default workload = config 4 of
BASELINE.json ("batch of 64 independent 4K RGBA images, 200 seams each"): every
rank carves its own batch of 64 images of 3840x2160 RGBA by 200 vertical seams as
one lock-step batch (--images-per-gpu 8 gives the literal 64/8 shard); images are
independent, so there is no data-path collective and scaling is weak.
--workload single4k | fhd | 8k run configs 3 / 2 / 5's geometry instead.

Memory: ONE set of carvers per rank, whatever K and W are.  The input batch is
generated on the GPU once and stays in HBM; every step starts by re-loading the
carvers from it (lqrx_carver_reload_device_batch: a device-to-device copy, the
HBM-resident analogue of lqr_carver_new's buffer hand-over, src/render.c:222) and
that copy is INSIDE the timed region.  The three phases the reference brackets
with __CLOCK_IT__ (src/render.c:214-217 read, :314-316 resize, :358-362 write)
map to: image generation + first upload (untimed, reported as setup_s), the K
timed steps, and the read-out + gather after the timed region (gather_ms).

value = Mseams*px/s over ALL ranks = sum over phases (n_seams * W * H) * images
* K / wall time; wall time = max over ranks of the time of exactly K steps,
bracketed by barrier + device synchronize on both sides.

Run with --gpus N > 1 and no RANK in the environment, the script re-executes
itself under torch.distributed.run with N ranks (one per GPU, RCCL).

Also on the JSON line:
  roofline      the dominant HBM kernel (k_carve), HIP-event timed on its own
                stream inside the timed region; achieved = algorithmic bytes
                (8 B x W*H/2 per image per launch) / mean launch
                time; peak 8 TB/s nominal, measured_copy_peak = this device's
                own 16-B streaming-copy rate (read+write).
  cpu_baseline  the CPU oracle (oracle/, a port -- real liblqr is not available
                here) timed on this host: one image on one core, and one image
                per core on all cores for the batch workload (nproc stated).
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
# --sub-batches N (opt-in) runs the batch on N HIP streams; each needs a hardware queue of its own and the HIP runtime's
# default is 4 per process, shared with torch's streams.  The variable must be in the environment before the first HIP
# call of the process, i.e. before torch is imported; it changes nothing for the default single-stream run.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="batch4k", choices=["batch4k", "single4k", "fhd", "8k"])
    ap.add_argument("--images-per-gpu", type=int, default=64)
    ap.add_argument("--seams", type=int, default=None)
    ap.add_argument("--kernel-times", action="store_true",
                    help="HIP-event time every kernel of the seam loop (kernels_ms), not only k_carve; costs ~2.5 %% of the step")
    ap.add_argument("--sub-batches", type=int, default=1,
                    help="split the batch over this many HIP streams (chain kernels of one under the carve of another): more "
                         "throughput, but every kernel then shares the chip and the per-launch roofline figure drops")
    ap.add_argument("--update-mode", type=int, default=-1,
                    help="update_mmap kernel: -1 the engine's choice, 0 band, 1 tiled full width, 2 band-mw")
    ap.add_argument("--band-kernel", type=int, default=None,
                    help="A/B hook (experiments build): 0 k_band_update_tw, 1 k_band_update_td<4 px>, 2 k_band_update_td<2 px>, 3 k_band_update_ls")
    ap.add_argument("--band-variant", type=int, default=0, help="A/B hook for band-kernel experiments")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--switch-freq", type=int, default=2, help="lqr side switch frequency (plug-in: 2, render.c:237)")
    return ap.parse_args()


WORKLOADS = {
    #            W     H    new_w  new_h
    "batch4k": (3840, 2160, 3640, 2160),      # config 4, per-GPU shard
    "single4k": (3840, 2160, 3340, 1660),     # config 3
    "fhd": (1920, 1080, 1720, 1080),          # config 2
    "8k": (7680, 4320, 6680, 4320),           # config 5 geometry (no masks)
}


def work_seam_px(w, h, nw, nh):
    """sum over phases of n_seams * W_start * H_start (HOR order)"""
    return abs(w - nw) * w * h + abs(h - nh) * nw * h


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
    lib.lqrhip_set_update_mode.argtypes = [C.c_int]
    lib.lqrhip_set_update_mode(args.update_mode)
    if args.band_kernel is not None or args.band_variant:      # only in a -DLQR_BAND_EXPERIMENTS build of the library
        if not hasattr(lib, "lqrhip_set_band_kernel"):
            raise SystemExit("bench.py: --band-kernel / --band-variant need a library built with -DLQR_BAND_EXPERIMENTS")
        lib.lqrhip_set_band_kernel.argtypes = [C.c_int]
        lib.lqrhip_set_band_kernel(args.band_kernel or 0)
        lib.lqrhip_set_band_variant.argtypes = [C.c_int]
        lib.lqrhip_set_band_variant(args.band_variant)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    dist = None
    if world > 1 or "RANK" in os.environ:       # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)

    W, H, NW, NH = WORKLOADS[args.workload]
    nimg = args.images_per_gpu if args.workload == "batch4k" else 1
    if args.seams is not None:
        NW = W - args.seams

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

    # ---- phase "read" (render.c:214-217): inputs generated on the GPU, resident in HBM from here on;
    # ONE set of carvers, created (working planes allocated) before the timed region
    t_setup = time.perf_counter()
    images = make_images(nimg, W, H, 100 + rank * nimg, dev)
    torch.cuda.synchronize()
    img0_host = images[0].cpu().numpy()
    ptrs = [images[i].data_ptr() for i in range(nimg)]
    carvers = []
    for i in range(nimg):
        c = L.Carver(eng, img0_host if i == 0 else np.zeros((H, W, 4), np.uint8))      # pixels come from `images` at every step
        c.configure(switch_freq=args.switch_freq, enl_step=1.5)                          # plug-in defaults, main.c:62-87
        carvers.append(c)
    sync()
    t_setup = time.perf_counter() - t_setup

    def run_step():
        ret = L.reload_device_batch(eng, carvers, ptrs)
        assert ret == L.LQR_OK, "reload failed: %d (%s)" % (ret, lib.lqrhip_last_error().decode())
        if len(carvers) == 1:
            ret = carvers[0].resize(NW, NH)
        else:
            ret = L.resize_batch(eng, carvers, NW, NH)
        assert ret == L.LQR_OK, "resize failed: %d (%s)" % (ret, lib.lqrhip_last_error().decode())

    # ---- phase "resize" (render.c:314-316): W warm-up steps, then exactly K timed steps
    for _ in range(args.warmup):
        run_step()
    lib.lqrhip_prof_reset()
    lib.lqrhip_prof_enable(1 if args.kernel_times else 2)
    barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    sync(); barrier()
    t1 = time.perf_counter()
    lib.lqrhip_prof_enable(0)
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    used_gb, total_gb = mem_used_gb()

    # ---- per-kernel HIP-event times collected inside the timed region
    def prof(name):
        ms, n, by = C.c_double(0), C.c_longlong(0), C.c_double(0)
        lib.lqrhip_prof_get(name.encode(), C.byref(ms), C.byref(n), C.byref(by))
        return ms.value, n.value, by.value
    kern = {k: prof(k) for k in ("carve", "vpath", "band_update", "dp_update", "dp_update_tiled", "dp_sweep", "emap_update")}
    c_ms, c_n, c_bytes = kern["carve"]
    roofline = None
    if c_n:
        achieved = c_bytes / (c_ms * 1e-3) / 1e9
        # HBM traffic of k_carve per launch from the committed rocprofv3 PMC passes of this command
        # (separate FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950
        # note), scaled from the profiled batch size to this run's; null if the profile is missing
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_k_carve.json")
        if os.path.exists(pmc) and args.workload == "batch4k" and args.seams is None:
            pj = json.load(open(pmc))
            traffic = round((2 * pj["fetch_size_kb_mean"] + pj["write_size_kb_mean"]) * 1024 / pj["images_per_launch"] * nimg)
        roofline = {"bound": "hbm", "kernel": "k_carve", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                    "avg_launch_us": round(c_ms * 1e3 / c_n, 2), "launches": c_n,
                    "alg_bytes_per_launch": round(c_bytes / c_n)}
        if rank == 0:
            # this device's own ceiling: 16-B streaming copy of 2 GiB (read + write), outside the timed region
            g = C.c_double(0)
            lib.lqrhip_copy_bandwidth.argtypes = [C.c_ulonglong, C.c_int, C.POINTER(C.c_double)]
            if lib.lqrhip_copy_bandwidth(2 << 30, 10, C.byref(g)) == 0 and g.value > 0:
                roofline["measured_copy_peak"] = round(g.value, 1)
                roofline["frac_of_measured"] = round(achieved / g.value, 4)

    # ---- phase "write" (render.c:358-362): results of the last step, gathered to rank 0 over RCCL
    gather_ms = None
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
        gathered = [torch.empty_like(outs) for _ in range(world)] if rank == 0 else None
        dist.gather(outs, gathered, dst=0)
        sync()
        gather_ms = (time.perf_counter() - tg) * 1e3
        del gathered
    checksum = int(outs.to(torch.int64).sum().item())
    g = carvers[0].getters()
    assert (g["width"], g["height"]) == (NW, NH), g

    work = work_seam_px(W, H, NW, NH) * nimg * args.steps       # seam*px per rank
    value = work * world / elapsed / 1e6

    result = {
        "metric": "Mseams*pixels/sec on 4K RGBA", "value": round(value, 1), "unit": "Mseams*px/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d x %dx%d RGBA per GPU, resize to %dx%d (%d vertical%s seams each), side-switch %d" % (
                       args.workload, nimg, W, H, NW, NH, W - NW, (" + %d horizontal" % (H - NH)) if NH != H else "",
                       args.switch_freq),
                   "images_per_gpu": nimg, "width": W, "height": H, "new_width": NW, "new_height": NH,
                   "parallelism": "images sharded i mod N, no data-path collective", "streams_per_gpu": args.sub_batches},
        "roofline": roofline,
        "kernels_ms": {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in kern.items() if v[1]},
        "gather_ms": None if gather_ms is None else round(gather_ms, 2),
        "setup_s": round(t_setup, 2),
        "hbm_used_gb": round(used_gb, 1), "hbm_total_gb": round(total_gb, 1),
        "output_checksum": checksum,
    }

    # ---- CPU baseline: the oracle (a port of the algorithm; test infrastructure, loaded here only as the
    # reported baseline and the spot checker) on this host's cores
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from concurrent.futures import ThreadPoolExecutor
        orc = L.Api(os.path.join(ROOT, "oracle", "liblqr_oracle.so"), "o")
        ncores = os.cpu_count() or 1
        cw, chh, cnw, cnh = W, H, NW, NH
        sample = "1 image of the workload (%dx%d -> %dx%d), 1 core" % (cw, chh, cnw, cnh)
        if args.workload in ("single4k", "8k"):     # bound the sample: 100 (+100) seams instead of 500+500 / 1000
            cnw, cnh = W - 100, (H - 100 if NH != H else H)
            sample = "1 image %dx%d -> %dx%d (first %d seams of the workload), 1 core" % (cw, chh, cnw, cnh, (W - cnw) + (H - cnh))
        oc = L.Carver(orc, img0_host).configure(switch_freq=args.switch_freq, enl_step=1.5)
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
        if nimg > 1:
            # for the batch, one image per core over all host cores (liblqr itself is single-threaded;
            # ctypes releases the GIL inside the C call, so threads run the oracle truly in parallel)
            nt = min(ncores, nimg)
            host_imgs = [images[i].cpu().numpy() for i in range(nt)]

            def one(im):
                o = L.Carver(orc, im).configure(switch_freq=args.switch_freq, enl_step=1.5)
                r = o.resize(NW, NH)
                o.destroy()
                return r
            ta = time.perf_counter()
            with ThreadPoolExecutor(max_workers=nt) as ex:
                rets = list(ex.map(one, host_imgs))
            ta = time.perf_counter() - ta
            assert all(r == L.LQR_OK for r in rets)
            result["cpu_baseline"]["all_cores"] = {
                "value": round(work_seam_px(W, H, NW, NH) * nt / ta / 1e6, 1), "unit": "Mseams*px/s", "cores": nt,
                "sample": "%d images of the workload, one per core, concurrently" % nt, "seconds": round(ta, 2)}

    if rank == 0:
        print(json.dumps(result), flush=True)
    for c in carvers:
        c.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
