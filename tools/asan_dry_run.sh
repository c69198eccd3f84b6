#!/bin/bash
# Host-side AddressSanitizer sweep of the library WITHOUT a GPU (tools/asan_dry_run.py): an ASan-instrumented HOST build of every
# csrc/*.hip (device code untouched: -fno-gpu-sanitize), then the tiny ControlNet and UNet (+ ControlNet residuals) plans built and "run"
# through the real C entry points on host tensors -- every launch fails at hipLaunchKernel, everything before it (validation, routing,
# planning, argument blocks, XCD map) runs for real -- plus every (variant, split-K) of every tunable op the way the autotuner asks.
# usage: bash tools/asan_dry_run.sh        (a few minutes; prints the number of AddressSanitizer reports: 0 is the expected answer)
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/sfast_asan; mkdir -p $OUT
RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)/libclang_rt.asan-x86_64.so
cd stable-fast_amd/csrc
ls *.hip | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=fast -fsanitize=address -fno-gpu-sanitize -shared-libsan -I../../include \$( [ {} = attention.hip ] && echo '-mllvm -amdgpu-mfma-vgpr-form=1' ) -c {} -o $OUT/{}.o 2>$OUT/{}.err"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -o $OUT/libsfast_hip_asan.so $OUT/*.o
cd ../..
SFAST_ASAN_LIB=$OUT/libsfast_hip_asan.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 LD_PRELOAD=$RT python tools/asan_dry_run.py > $OUT/dry.log 2>&1 || true
echo "AddressSanitizer reports: $(grep -c AddressSanitizer $OUT/dry.log)"
tail -n 5 $OUT/dry.log | cut -c1-400
