#!/bin/bash
# Round-4 GPU sessions (one gpurun call each): tools/gpu_r4.sh <stage>
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
STAGE=${1:-s1}
export TMPDIR=/tmp
PYT="python -m pytest -q --no-header --tb=short -p no:cacheprovider --timeout=900 --maxfail=30 -m gpu"
run() { # name, timeout, command...
  local name=$1 to=$2; shift 2
  echo "=== $name ===" | tee -a gpurun_out/session.log
  local t0=$(date +%s)
  timeout $to "$@" > gpurun_out/$name.log 2>&1
  echo "exit=$? $(( $(date +%s) - t0 ))s $(tail -n 1 gpurun_out/$name.log | cut -c1-300)" | tee -a gpurun_out/session.log
}
rm -f gpurun_out/parity.jsonl gpurun_out/session.log
rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -2 >> gpurun_out/session.log
case $STAGE in
s1)  # the reference's own Triton kernels on this box (fixture + oracle / HIP comparison + timing); loop-structure probe; baseline bench
  run ref_fixture 900 python tests/golden/make_golden_ref_triton.py --out gpurun_out/ref_triton_small.pt
  run t_ref_triton 1500 $PYT tests/test_ref_triton_gpu.py
  run wdirect 300 tools/micro/wdirect
  run trread 60 tools/micro/trread
  run t_sanity 900 $PYT tests/test_ops_gpu.py -k "attention or split or conv_bias or group_norm" -x
  run bench 900 python bench.py --steps 50 --warmup 10 --no-end-to-end --dump-kernels gpurun_out/kernels.json
  ;;
s2)  # gnconv (fused GroupNorm+SiLU -> conv at the 8x8 level): parity, then the step with and without it; round-4 small items
  run t_gnconv 900 $PYT tests/test_ops_gpu.py -k "gn_conv2d or layer_norm"
  run t_api 900 $PYT tests/test_reference_api_gpu.py -k "addmm or geglu or bmm or grouped"
  run t_unet 1200 $PYT tests/test_unet_gpu.py -k "controlnet or tiny or sd15_unet_parity or compile_drop_in"
  run t_ref_triton 1500 $PYT tests/test_ref_triton_gpu.py
  run smoke 600 python __graft_entry__.py smoke
  run bench_fused 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --dump-kernels gpurun_out/kernels_fused.json
  SFAST_FUSE_GN_CONV=0 run bench_unfused 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --dump-kernels gpurun_out/kernels_unfused.json
  run bench_fused2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  ;;
s3)  # gnconv with the deep weight prefetch: parity, same-process A/B + per-kernel durations, the step with and without it
  run t_gnconv 900 $PYT tests/test_ops_gpu.py -k "gn_conv2d"
  run t_api 900 $PYT tests/test_reference_api_gpu.py -k "geglu"
  run t_ref_triton 1500 $PYT tests/test_ref_triton_gpu.py -k "group_norm"
  run gnconv_ab 600 python tools/gnconv_ab.py
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_gnconv -o gnconv -- python $OLDPWD/tools/gnconv_ab.py > $OLDPWD/gpurun_out/prof_gnconv.log 2>&1)
  find gpurun_out/prof_gnconv -name "*kernel_stats*" -exec cp {} gpurun_out/gnconv_kernel_stats.csv \;
  rm -rf gpurun_out/prof_gnconv
  run bench_fused 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --dump-kernels gpurun_out/kernels_fused.json
  SFAST_FUSE_GN_CONV=0 run bench_unfused 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  run bench_fused2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  ;;
s4)  # where a gnconv workgroup spends its time
  run gnconv_trace 300 python tools/gnconv_trace.py
  run gnconv_ab 300 python tools/gnconv_ab.py
  ;;
s5)  # gnconv v2 (slice first, gamma / beta through LDS, contiguous k-step ranges, addresses per tap): parity, trace, A/B, the step
  run t_gnconv 900 $PYT tests/test_ops_gpu.py -k "gn_conv2d"
  run gnconv_trace 300 python tools/gnconv_trace.py
  run gnconv_ab 300 python tools/gnconv_ab.py
  run bench_fused 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --dump-kernels gpurun_out/kernels_fused.json
  SFAST_FUSE_GN_CONV=0 run bench_unfused 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  run bench_fused2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  ;;
s6)  # the consumer GroupNorm inside the split-K reduce launch: parity, whole-UNet parity, the step with and without the pass
  run t_reduce_gn 900 $PYT tests/test_ops_gpu.py -k "consumer_groupnorm or fused_groupnorm_epilogue"
  run t_unet 1500 $PYT tests/test_unet_gpu.py -k "tiny or sd15_unet_parity or compile_drop_in or controlnet or live_weight" tests/test_vae_gpu.py
  run smoke 600 python __graft_entry__.py smoke
  run bench_on 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --dump-kernels gpurun_out/kernels_on.json
  SFAST_GN_IN_REDUCE=0 run bench_off 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  run bench_on2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  SFAST_GN_IN_REDUCE=0 run bench_off2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  ;;
s7)  # crash hunt (ControlNet chain) with the reduce pass off / on; reduce+GN two-phase loads: parity + step A/B
  SFAST_GN_IN_REDUCE=0 run t_cn_off 600 $PYT tests/test_unet_gpu.py -k "controlnet"
  run t_cn_on 600 $PYT tests/test_unet_gpu.py -k "controlnet"
  run t_reduce_gn 900 $PYT tests/test_ops_gpu.py -k "consumer_groupnorm or fused_groupnorm_epilogue"
  run bench_on 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --dump-kernels gpurun_out/kernels_on.json
  SFAST_GN_IN_REDUCE=0 run bench_off 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  run bench_on2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  SFAST_GN_IN_REDUCE=0 run bench_off2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  ;;
s8)  # GroupNorm apply: input rows requested before the statistics prologue -- parity + step A/B
  run t_gn 900 $PYT tests/test_ops_gpu.py -k "group_norm or gn_apply or statistics"
  run t_unet 900 $PYT tests/test_unet_gpu.py -k "tiny or sd15_unet_parity"
  run bench_on 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --dump-kernels gpurun_out/kernels_on.json
  SFAST_GN_PREFETCH=0 run bench_off 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --dump-kernels gpurun_out/kernels_off.json
  run bench_on2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  SFAST_GN_PREFETCH=0 run bench_off2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  ;;
full)  # the whole GPU suite as the driver runs it, then the probe-build subset, then smoke
  run t_all 1700 $PYT tests --durations=12
  SFAST_HIP_PROBES=1 run t_probes 900 $PYT tests/test_ops_gpu.py -k "patch or join"
  run smoke 600 python __graft_entry__.py smoke
  ;;
final)  # the round's SD1.5 evidence: PMC traffic by symbol, the default bench line (roofline + cpu_baseline + end-to-end + variants), rocprofv3 kernel stats
  run pmc_traffic 900 bash tools/gpu_pmc_bench.sh sd15 6
  cp gpurun_out/pmcb/traffic_by_symbol.json profiles/r04_pmc_traffic_by_symbol.json
  run bench_default 900 python bench.py --dump-kernels gpurun_out/kernels.json
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o bench -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants > $OLDPWD/gpurun_out/rocprof.log 2>&1 )
  echo "rocprof exit=$?" >> gpurun_out/session.log
  for db in $(find gpurun_out/prof -name "*.db"); do python tools/rocpd_summary.py $db --csv gpurun_out/kernel_stats.csv --top 60 --step-marker cfg_ddim --steps 12 > gpurun_out/kernel_stats.txt; done
  rm -rf gpurun_out/prof
  run pmc_attn 700 bash tools/gpu_pmc_attn.sh
  run bench_torchrun 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants
  ;;
s9)  # transposing copies; probe-build subset; crash hunt with the reduce pass on over the whole UNet / VAE files
  run t_copy 600 $PYT tests/test_ops_gpu.py -k "strided_copy or transposes" tests/test_ref_triton_gpu.py -k "copy or strided"
  SFAST_HIP_PROBES=1 run t_probes 900 $PYT tests/test_ops_gpu.py -k "patch or join"
  SFAST_GN_IN_REDUCE=1 run t_unet_gnred 1500 $PYT tests/test_unet_gpu.py tests/test_vae_gpu.py
  ;;
s10)  # IP-Adapter + SDXL-style ControlNet (native plans, through compile); the SDXL and SVD-XT bench lines of the round
  run t_ip 1200 $PYT tests/test_unet_gpu.py -k "ip_adapter or sdxl_style_controlnet or controlnet"
  run bench_sdxl 1200 python bench.py --config sdxl --no-cpu-baseline
  run bench_svd 1200 python bench.py --config svd --no-cpu-baseline
  ;;
s11)  # fragment-load probe, second pass (own rows per wave, L2-resident, strided vs pre-swizzled); the weight-streaming conv without GN
  run wdirect 300 tools/micro/wdirect
  run t_gnconv 900 $PYT tests/test_ops_gpu.py -k "gn_conv2d"
  run gnconv_ab 300 python tools/gnconv_ab.py
  ;;
s12)  # pipe 4 (packed weights): parity
  run t_pk 1500 $PYT tests/test_packed_weights_gpu.py
  ;;
s13)  # pipe 4 in the step: 1x1 conv parity, whole-UNet parity + live weight update, bench with / without packed weights
  run t_pk 900 $PYT tests/test_packed_weights_gpu.py -k "conv_packed or repack"
  run t_unet 1200 $PYT tests/test_unet_gpu.py -k "sd15_unet_parity or tiny or live_weight or compile_drop_in"
  SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_pk.json run bench_pk 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --dump-kernels gpurun_out/kernels_pk.json
  SFAST_PACKED_WEIGHTS=0 run bench_nopk 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_pk.json run bench_pk2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  ;;
s14)  # where pipe 4 stands against the ring kernels per shape, and where its workgroups spend their time
  run t_pk 1500 $PYT tests/test_packed_weights_gpu.py -x
  run pk_ab 900 python tools/pk_ab.py
  run pk_trace 300 python tools/pk_trace.py
  ;;
s15)  # the step with pipe 4 among the tuner's candidates, against the same build without packed copies
  SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_pk.json run bench_pk 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --dump-kernels gpurun_out/kernels_pk.json
  SFAST_PACKED_WEIGHTS=0 run bench_nopk 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_pk.json run bench_pk2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  SFAST_PACKED_WEIGHTS=0 run bench_nopk2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_pk_sdxl.json run bench_sdxl_pk 900 python bench.py --config sdxl --no-cpu-baseline --no-roofline
  SFAST_PACKED_WEIGHTS=0 run bench_sdxl_nopk 900 python bench.py --config sdxl --no-cpu-baseline --no-roofline
  run t_unet 1200 $PYT tests/test_unet_gpu.py -k "sd15_unet_parity or tiny or live_weight or compile_drop_in"
  ;;
s16)  # un-fused LoRA on the native plan (both layouts, scale, in-place switch, SD1.5-size merge cost); op-level tests of the round
  run t_lora 1500 $PYT tests/test_unet_gpu.py -k "lora"
  run t_pk 1500 $PYT tests/test_packed_weights_gpu.py
  run t_gnconv 600 $PYT tests/test_ops_gpu.py -k "gn_conv2d"
  ;;
final2)  # end of round 4: the whole GPU suite on HEAD, smoke, the default bench line, kernel stats, 1-rank torchrun, 8 images per GPU, SVD-XT
  run pytest_full 2400 $PYT tests
  run smoke 600 python __graft_entry__.py smoke
  run bench_default 1200 python bench.py --dump-kernels gpurun_out/kernels.json
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o bench -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants > $OLDPWD/gpurun_out/rocprof.log 2>&1 )
  echo "rocprof exit=$?" >> gpurun_out/session.log
  for db in $(find gpurun_out/prof -name "*.db"); do python tools/rocpd_summary.py $db --csv gpurun_out/kernel_stats.csv --top 60 --step-marker cfg_ddim --steps 12 > gpurun_out/kernel_stats.txt; done
  rm -rf gpurun_out/prof
  run bench_torchrun 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants
  SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_images8.json run bench_images8 1200 python bench.py --images 8 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants
  SFAST_TUNE_CACHE=$PWD/gpurun_out/tune_svd.json run bench_svd 1500 python bench.py --config svd --no-cpu-baseline
  ;;
s17)  # IP-Adapter family after the external-projection mode
  run t_ip 1200 $PYT tests/test_unet_gpu.py -k "ip_adapter or lora"
  ;;
s18)  # engine-level repack test; live-weight and LoRA switch through compile with packed copies on
  run t_repack 900 $PYT tests/test_packed_weights_gpu.py -k "repack" tests/test_unet_gpu.py -k "live_weight or lora or repack"
  ;;
s19)  # XCD map with contiguous runs: parity over the GEMM / conv families, then the step with and without it
  run t_ops 1500 $PYT tests/test_ops_gpu.py -k "linear or conv or geglu or split or statistics" tests/test_packed_weights_gpu.py
  run bench_run 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --dump-kernels gpurun_out/kernels_xrun.json
  SFAST_XCD_MAP=1 run bench_box 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  run bench_run2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  SFAST_XCD_MAP=1 run bench_box2 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-end-to-end --no-variants --no-roofline
  run bench_sdxl_run 900 python bench.py --config sdxl --no-cpu-baseline --no-roofline
  SFAST_XCD_MAP=1 run bench_sdxl_box 900 python bench.py --config sdxl --no-cpu-baseline --no-roofline
  ;;
final3)  # HEAD after the last library change: the whole GPU suite, smoke, the default bench line
  run pytest_full 2400 $PYT tests
  run smoke 600 python __graft_entry__.py smoke
  run bench_default 1200 python bench.py --dump-kernels gpurun_out/kernels.json
  ;;
extra)  # measurement only (no code change behind it): SDXL line with its roofline block, 2 / 4 images per GPU
  run bench_sdxl 900 python bench.py --config sdxl --no-cpu-baseline
  run bench_images2 600 python bench.py --images 2 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants
  run bench_images4 600 python bench.py --images 4 --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants
  ;;
extra2)  # the SDXL and SD1.5 lines again after bench.py learned the pipe-4 symbols (roofline grouping only; no library change)
  run bench_sdxl 900 python bench.py --config sdxl --no-cpu-baseline
  run bench_default 1200 python bench.py --dump-kernels gpurun_out/kernels.json
  ;;
probes)  # the probe-build subset on HEAD (patch pipe, in-kernel join: they share decode_block / the epilogue with the product kernels)
  SFAST_HIP_PROBES=1 run t_probes 1200 $PYT tests/test_ops_gpu.py -k "patch or join"
  ;;
nopk)  # the whole-model suites with packed copies switched off (the strictly-live-weights mode)
  SFAST_PACKED_WEIGHTS=0 run t_models_nopk 1800 $PYT tests/test_unet_gpu.py tests/test_sdxl_gpu.py tests/test_vae_gpu.py tests/test_svd_gpu.py
  ;;
dbg)  # the ControlNet -> UNet chain test that aborted with packed copies off: localise
  SFAST_PACKED_WEIGHTS=0 run d_plain 600 $PYT tests/test_unet_gpu.py -k "controlnet_engine_and_compiled_chain" -x
  SFAST_PACKED_WEIGHTS=0 HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 run d_blocking 600 $PYT tests/test_unet_gpu.py -k "controlnet_engine_and_compiled_chain" -x
  SFAST_PACKED_WEIGHTS=0 SFAST_AUTOTUNE=0 run d_notune 600 $PYT tests/test_unet_gpu.py -k "controlnet_engine_and_compiled_chain" -x
  SFAST_PACKED_WEIGHTS=0 run d_file 900 $PYT tests/test_unet_gpu.py -x
  for f in d_plain d_blocking d_notune d_file; do echo "--- $f"; head -c 1500 gpurun_out/$f.log; echo; done >> gpurun_out/session.log
  ;;
flaky)  # how often does the ControlNet -> UNet chain test die? 24 fresh processes of the first 14 tests of the file (the order of the failing run)
  for i in $(seq 1 12); do
    SFAST_PACKED_WEIGHTS=0 timeout 300 $PYT tests/test_unet_gpu.py -k "tiny or sd15_unet_parity or compile_drop_in or lcm or live_weight or rccl or add_strided or controlnet" > gpurun_out/flaky_$i.log 2>&1
    echo "nopk run $i exit=$? $(tail -n 1 gpurun_out/flaky_$i.log | cut -c1-100)" >> gpurun_out/session.log
  done
  for i in $(seq 13 24); do
    timeout 300 $PYT tests/test_unet_gpu.py -k "tiny or sd15_unet_parity or compile_drop_in or lcm or live_weight or rccl or add_strided or controlnet" > gpurun_out/flaky_$i.log 2>&1
    echo "default run $i exit=$? $(tail -n 1 gpurun_out/flaky_$i.log | cut -c1-100)" >> gpurun_out/session.log
  done
  for f in gpurun_out/flaky_*.log; do grep -q passed $f && rm -f $f; done
  ;;
esac
cat gpurun_out/session.log
