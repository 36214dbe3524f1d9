# k_band_tiles at 64 and 48 images per GPU with forced tile counts (base + reserve) against the default k_band_update_tw: DESIGN.md 4.15
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["ms_per_step"], d.get("band_tiles_stats"))'
run() { echo -n "$* : "; python bench.py --steps 8 --warmup 3 --no-configs --no-kernel-breakdown "$@" 2>/dev/null | python3 -c "$P"; }
run --images-per-gpu 64
run --images-per-gpu 64 --update-mode 4 --band-tiles 6 --band-tiles-reserve 2
run --images-per-gpu 64 --update-mode 4 --band-tiles 8 --band-tiles-reserve 2
run --images-per-gpu 64 --update-mode 4 --band-tiles 8 --band-tiles-reserve 4
run --images-per-gpu 64 --update-mode 4 --band-tiles 10 --band-tiles-reserve 4
run --images-per-gpu 64 --update-mode 4 --band-tiles 12 --band-tiles-reserve 6
run --images-per-gpu 64 --update-mode 4 --band-tiles 8 --band-tiles-reserve 2 --sub-batches 2
run --images-per-gpu 64 --update-mode 4 --band-tiles 12 --band-tiles-reserve 4 --sub-batches 2
run --images-per-gpu 48
run --images-per-gpu 48 --update-mode 4 --band-tiles 8 --band-tiles-reserve 2
run --images-per-gpu 48 --update-mode 4 --band-tiles 10 --band-tiles-reserve 4
