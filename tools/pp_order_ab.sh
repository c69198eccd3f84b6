#!/bin/bash
# same-box A/B of library builds that differ in one compile-time knob of the pipe-5 lockstep loop (gpurun_ab/lib_<tag>.so, built with
# SFAST_EXTRA_CFLAGS=-D... python stable-fast_amd/build.py and copied there): conv / GEMM / GEGLU timings of the lockstep variants per build.
# usage: tools/pp_order_ab.sh tag [tag ...]
cp stable-fast_amd/sfast/_lib/libsfast_hip.so /tmp/lib_keep.so
for rep in 1 2; do
  for tag in "$@"; do
    cp gpurun_ab/lib_$tag.so stable-fast_amd/sfast/_lib/libsfast_hip.so
    echo "== build $tag rep $rep"
    timeout -k 10 400 python tools/pp_ab.py --no-check --quick --only conv 2>/dev/null | grep "best old\|v57\|v58" | cut -c1-150
    timeout -k 10 300 python tools/pp_ksweep.py 2>/dev/null | grep "v57\|v58" | cut -c1-230
  done
done
cp /tmp/lib_keep.so stable-fast_amd/sfast/_lib/libsfast_hip.so
