#!/bin/bash
# One PMC pass (counters only with --kernel-trace, as the pool requires) over tools/ws_loop_probe.py: LDS bank conflicts / active cycles,
# MFMA busy cycles and LDS wait cycles of the wave-specialised conv kernel and its timing-only instantiations.
cd "$(dirname "$0")/.."
rm -rf gpurun_out/pmc_ws; mkdir -p gpurun_out/pmc_ws
export TMPDIR=/tmp PYTHONPATH=$PWD/stable-fast_amd
R=$PWD
( cd /tmp && timeout 110 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $R/gpurun_out/pmc_ws -o sq -- python $R/tools/ws_loop_probe.py > $R/gpurun_out/pmc_ws/run.log 2>&1 )
echo "pmc exit=$? $(tail -n 1 $R/gpurun_out/pmc_ws/run.log | cut -c1-120)"
for db in $(find $R/gpurun_out/pmc_ws -name "*.db"); do python $R/tools/pmc_extract.py $db $R/gpurun_out/pmc_ws/ws_sq.json --by-symbol; rm -f $db; done
ls gpurun_out/pmc_ws
