#!/usr/bin/env python3
"""bench.py -- seam-carving throughput on MI355X (metric of BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (lqr_carver_resize: energy -> DP -> seam
pick/backtrack -> carve) over one batch of synthetic
images that are already resident in HBM.  Default workload = config 4 of
BASELINE.json ("batch of 64 independent 4K RGBA images, 200 seams each"): every
rank carves its own batch of 64 images of 3840x2160 RGBA by 200 vertical seams as
one lock-step batch (--images-per-gpu 8 gives the literal 64/8 shard); images are
independent, so there is no data-path collective and scaling is weak.  --workload single4k runs
config 3 (one 4K image, 500 vertical + 500 horizontal seams) instead.

value = Mseams*px/s over ALL ranks = sum over phases (n_seams * W * H) * images
/ wall time, wall time = max over ranks of the time of exactly K steps,
bracketed by barrier + device synchronize.

Also on the JSON line:
  roofline      the dominant HBM kernel (k_carve), HIP-event timed inside the
                timed region; achieved = algorithmic bytes (8 B x W*H/2 per
                image per launch) / mean launch time; peak 8 TB/s.
  cpu_baseline  the CPU oracle (oracle/, a port -- real liblqr is not
                available) timed on this host on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--switch-freq", type=int, default=2, help="lqr side switch frequency (plug-in: 2, render.c:237)")
    return ap.parse_args()


WORKLOADS = {
    #            W     H    new_w  new_h  images
    "batch4k": (3840, 2160, 3640, 2160),      # config 4, per-GPU shard
    "single4k": (3840, 2160, 3340, 1660),     # config 3
    "fhd": (1920, 1080, 1720, 1080),          # config 2
    "8k": (7680, 4320, 6680, 4320),           # config 5 geometry (no masks)
}


def work_seam_px(w, h, nw, nh):
    """sum over phases of n_seams * W_start * H_start (HOR order)"""
    return abs(w - nw) * w * h + abs(h - nh) * nw * h


def make_image(w, h, seed):
    """photo-like synthetic RGBA (low-frequency structure + 1/f octave noise, u8, alpha=255),
    generated with torch on the CPU (fast at 4K); same recipe as tests/datasets.photo_like"""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(int(seed))
    yy = torch.arange(h, dtype=torch.float32)[:, None]
    xx = torch.arange(w, dtype=torch.float32)[None, :]
    base = torch.zeros(h, w)
    for _ in range(4):
        f = (torch.rand(2, generator=g) * 3.5 + 0.5) * 6.2831853
        ph = torch.rand(1, generator=g) * 6.2831853
        amp = torch.rand(1, generator=g) * 0.7 + 0.3
        base += amp * torch.sin(f[0] * xx / w + f[1] * yy / h + ph)
    out = base[None].repeat(3, 1, 1)
    n, i = 4, 0
    while n < max(w, h):
        gh, gw = max(2, n * h // max(w, h) + 1), max(2, n * w // max(w, h) + 1)
        grid = torch.randn(1, 3, gh, gw, generator=g)
        out += F.interpolate(grid, size=(h, w), mode="bilinear", align_corners=True)[0] * (0.9 / (i + 1))
        n *= 2
        i += 1
    out -= out.min()
    out *= 255.0 / max(float(out.max()), 1e-6)
    rgb = out.round().clamp(0, 255).to(torch.uint8).permute(1, 2, 0)
    a = torch.full((h, w, 1), 255, dtype=torch.uint8)
    return torch.cat([rgb, a], dim=2).contiguous().numpy()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # torch first: its bundled HIP runtime (same soname) must be the one runtime of the process;
    # the engine then binds to it and pins itself to device LOCAL_RANK
    import numpy as np
    import torch
    import lqr_ctypes as L
    eng = L.engine_api()
    lib = eng.lib
    if lib.lqrhip_init() < 0:
        lib.lqrhip_last_error.restype = C.c_char_p
        raise SystemExit("bench.py: no usable HIP device: %s" % lib.lqrhip_last_error().decode())

    dist = None
    if world > 1 or "RANK" in os.environ:       # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    W, H, NW, NH = WORKLOADS[args.workload]
    nimg = args.images_per_gpu if args.workload == "batch4k" else 1
    if args.seams is not None:
        NW = W - args.seams
    total_steps = args.warmup + args.steps

    # ---- inputs: synthetic images, uploaded (carvers created) BEFORE the timed region
    t_gen = time.time()
    images = [make_image(W, H, 100 + rank * nimg + i) for i in range(nimg)]
    t_gen = time.time() - t_gen

    def new_carvers():
        cs = []
        for im in images:
            c = L.Carver(eng, im)
            c.configure(switch_freq=args.switch_freq, enl_step=1.5)      # plug-in defaults, main.c:62-87
            cs.append(c)
        return cs

    steps = [new_carvers() for _ in range(total_steps)]

    def run_step(cs):
        if len(cs) == 1:
            ret = cs[0].resize(NW, NH)
        else:
            ret = L.resize_batch(eng, cs, NW, NH)
        assert ret == L.LQR_OK, "resize failed: %d" % ret

    def sync():
        lib.lqrhip_device_sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    for k in range(args.warmup):
        run_step(steps[k])
    lib.lqrhip_prof_reset()
    lib.lqrhip_prof_enable(1 if args.kernel_times else 2)
    barrier(); sync()
    t0 = time.perf_counter()
    for k in range(args.warmup, total_steps):
        run_step(steps[k])
    sync(); barrier()
    t1 = time.perf_counter()
    lib.lqrhip_prof_enable(0)
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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

    # ---- the roofline kernel in isolation: in the timed region k_carve runs concurrently with the
    # band update (that is what makes the step faster), which stretches its launch time; one extra
    # untimed step with the two kernels back to back gives the kernel's own bandwidth
    # (only with LQRHIP_BAND_TW=0: the default band kernel runs after a plain carve, nothing concurrent)
    if roofline and nimg > 1 and rank == 0 and os.environ.get("LQRHIP_BAND_TW", "1") == "0" and os.environ.get("LQRHIP_OVERLAP", "1") != "0":
        extra = new_carvers()
        lib.lqrhip_set_overlap(0)
        lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
        run_step(extra); sync()
        lib.lqrhip_prof_enable(0); lib.lqrhip_set_overlap(-1)
        i_ms, i_n, i_bytes = prof("carve")
        if i_n:
            ia = i_bytes / (i_ms * 1e-3) / 1e9
            roofline["concurrent_with"] = "k_band_update_mw (second stream)"
            roofline["isolated"] = {"achieved": round(ia, 1), "frac": round(ia / 8000.0, 4), "avg_launch_us": round(i_ms * 1e3 / i_n, 2),
                                    "note": "same kernel, same batch, carve and band update back to back (untimed extra step)"}
        for c in extra:
            c.destroy()

    # ---- results of the last step: gather to rank 0 over RCCL (outside the timed region)
    gather_ms = None
    last = steps[-1]
    checksum = 0
    if torch.cuda.is_available():
        outs = torch.empty((len(last), NH, NW, 4), dtype=torch.uint8, device="cuda")
        if last[0].getters()["orientation"] == 0:
            for i, c in enumerate(last):
                assert eng.lqrx_carver_read_image_device(c.p, outs[i].data_ptr()) == L.LQR_OK
        else:       # transposed carver frame: go through the host image-orientation read-out
            for i, c in enumerate(last):
                outs[i].copy_(torch.from_numpy(c.read_image()))
        if dist is not None and not args.no_gather:
            sync(); barrier()
            tg = time.perf_counter()
            gathered = [torch.empty_like(outs) for _ in range(world)] if rank == 0 else None
            dist.gather(outs, gathered, dst=0)
            sync()
            gather_ms = (time.perf_counter() - tg) * 1e3
        checksum = int(outs.to(torch.int64).sum().item())
    g = last[0].getters()
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
                   "parallelism": "images sharded i mod N, no data-path collective"},
        "roofline": roofline,
        "kernels_ms": {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in kern.items() if v[1]},
        "gather_ms": None if gather_ms is None else round(gather_ms, 2),
        "output_checksum": checksum,
    }

    # ---- CPU baseline: the oracle (a port of the algorithm) on one image of the same workload
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        orc = L.oracle_api()
        cw, chh, cnw, cnh = W, H, NW, NH
        sample = "1 image of the workload (%dx%d -> %dx%d)" % (cw, chh, cnw, cnh)
        if args.workload == "single4k":     # bound the sample: 100 + 100 seams instead of 500 + 500
            cnw, cnh = W - 100, H - 100
            sample = "1 image %dx%d -> %dx%d (100+100 of the 500+500 seams)" % (cw, chh, cnw, cnh)
        oc = L.Carver(orc, images[0]).configure(switch_freq=args.switch_freq, enl_step=1.5)
        tc = time.perf_counter()
        assert oc.resize(cnw, cnh) == L.LQR_OK
        tc = time.perf_counter() - tc
        cpu_val = work_seam_px(cw, chh, cnw, cnh) / tc / 1e6
        result["cpu_baseline"] = {"value": round(cpu_val, 1), "unit": "Mseams*px/s", "cores": 1, "kind": "port",
                                  "sample": sample, "seconds": round(tc, 2)}
        # parity spot check of the timed workload's first image against the oracle
        if (cnw, cnh) == (NW, NH):
            ref = oc.read_image()
            got = last[0].read_image()
            result["parity_vs_oracle"] = bool(np.array_equal(ref, got))
        oc.destroy()

    if rank == 0:
        print(json.dumps(result), flush=True)
    for cs in steps:
        for c in cs:
            c.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
