#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ablate_igemm.py > gpurun_out/ablate.txt 2>&1
echo "ablate exit=$?"
( cd /tmp && rocprofv3 -L > /root/repo/gpurun_out/counters_list.txt 2>&1 ) ; grep -c . gpurun_out/counters_list.txt
cat gpurun_out/ablate.txt | grep -v amdgpu.ids | cut -c1-220
