#!/bin/bash
# PMC passes (counters only with --kernel-trace, as the pool requires) over tools/pmc_delivery.py: how busy the matrix pipe, the
# vector-memory issue path and the texture-address / L1 blocks are in the GEMM kernels. Counter names are taken from `rocprofv3 -L`
# on the box (a name this build does not know is skipped, not guessed).
cd "$(dirname "$0")/.."
rm -rf gpurun_out/pmcd; mkdir -p gpurun_out/pmcd
export TMPDIR=/tmp PYTHONPATH=$PWD/stable-fast_amd
R=$PWD
( cd /tmp && rocprofv3 -L > $R/gpurun_out/pmcd/avail.txt 2>&1 )
pick() { for c in "$@"; do grep -qw "$c" $R/gpurun_out/pmcd/avail.txt && printf "%s " "$c"; done; }
pass() { # name, counters...
  local name=$1; shift
  local cs=$(pick "$@")
  echo "pass $name: $cs"
  [ -z "$cs" ] && return
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $cs -d $R/gpurun_out/pmcd -o $name -- python $R/tools/pmc_delivery.py > $R/gpurun_out/pmcd/$name.log 2>&1 )
  echo "pmc $name exit=$? $(tail -n 1 $R/gpurun_out/pmcd/$name.log | cut -c1-120)"
  for db in $(find $R/gpurun_out/pmcd -name "*.db"); do python $R/tools/pmc_extract.py $db $R/gpurun_out/pmcd/$name.json; rm -f $db; done
}
if [ "$1" != "ta" ]; then
pass sq SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
fi
# the TA / TCP blocks take few counters at a time ("Request exceeds the capabilities of the hardware to collect" with eight)
pass ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
pass tcp2 TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum
grep -c . gpurun_out/pmcd/avail.txt
ls gpurun_out/pmcd
