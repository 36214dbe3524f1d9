#!/bin/bash
# liblqr-hip.so with the experimental band kernels (csrc/band_experiments.inc) and, with "timing", their cycle accounting
cd "$(dirname "$0")/../gimp-lqr-plugin_amd" || exit 1
extra="-DLQR_BAND_EXPERIMENTS"; [ "$1" = timing ] && extra="$extra -DLQR_BAND_TIMING"
touch csrc/lqr_hip.hip
make HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -mllvm -amdgpu-sched-strategy=max-ilp $extra"
