#!/bin/bash
# PMC passes over the grouped weight-gradient launch at several CU grants (tools/wgrad_group_pmc.py) and the forward GEMM (tools/pp_exp.py)
TAG=${1:-wgpmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { name=$1; wl=$2; shift; shift; timeout -s KILL 300 rocprofv3 "$@" --kernel-trace -f csv -d $O/$name -o p -- python $R/tools/$wl > $O/$name.log 2>&1 || echo "$name failed"; }
for wl in wgrad_group_pmc.py pp_exp.py; do
  t=${wl%%.py}
  run ${t}_p1 $wl --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
  run ${t}_p2 $wl --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  for p in p1 p2; do echo "== $t $p"; python $R/tools/pmc_csv.py $O/${t}_$p | grep -E "gemm_wgrad_group|gemm_pp"; done
done > $O/pmc.txt 2>&1
find $O -name "*.csv" -size +1M -delete
cut -c1-900 $O/pmc.txt
