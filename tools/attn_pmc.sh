#!/bin/bash
# PMC passes over the divided-attention micro-benchmark (tools/attn_bench.py): where the attention kernels' wave cycles go.
# usage (GPU box, repo root): bash tools/attn_pmc.sh <tag>  -> gpurun_out/<tag>/attn_pmc.txt
TAG=${1:-attnpmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout -s KILL 300 rocprofv3 "$@" --kernel-trace -f csv -d $O/$name -o p -- python $R/tools/attn_bench.py > $O/$name.log 2>&1 || echo "$name failed"; }
run p1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run p2 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run p3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM
for p in p1 p2 p3; do echo "== $p"; python $R/tools/pmc_csv.py $O/$p; done > $O/attn_pmc.txt 2>&1
find $O -name "*.csv" -size +1M -delete
cat $O/attn_pmc.txt | cut -c1-600
