#!/bin/bash
# One PMC pass (counters only with --kernel-trace, as the pool requires) over tools/attn_ab.py: MFMA-pipe busy cycles of the
# attention kernels at the UNet's shapes.
cd "$(dirname "$0")/.."
rm -rf gpurun_out/pmc_attn; mkdir -p gpurun_out/pmc_attn
export TMPDIR=/tmp PYTHONPATH=$PWD/stable-fast_amd
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_attn -o sq -- python $R/tools/attn_ab.py --variants 32,64 > $R/gpurun_out/pmc_attn/run.log 2>&1 )
echo "pmc exit=$? $(tail -n 1 $R/gpurun_out/pmc_attn/run.log | cut -c1-120)"
for db in $(find $R/gpurun_out/pmc_attn -name "*.db"); do python $R/tools/pmc_extract.py $db $R/gpurun_out/pmc_attn/attn_sq.json; rm -f $db; done
ls gpurun_out/pmc_attn
