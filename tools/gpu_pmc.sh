#!/bin/bash
# PMC passes (counters only with --kernel-trace, as the pool requires) over tools/pmc_probe.py
cd "$(dirname "$0")/.."
rm -rf gpurun_out/prof gpurun_out/pmc; mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$PWD
pass() { # name, counters...
  local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc -o $name -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc/$name.log 2>&1 )
  echo "pmc $name exit=$? $(tail -n 1 $R/gpurun_out/pmc/$name.log | cut -c1-120)"
  for db in $(find $R/gpurun_out/pmc -name "*.db"); do python $R/tools/pmc_extract.py $db $R/gpurun_out/pmc/$name.json; rm -f $db; done
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcc2 FETCH_SIZE
pass tcc3 WRITE_SIZE
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE
ls gpurun_out/pmc
