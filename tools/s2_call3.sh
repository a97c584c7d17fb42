#!/bin/bash
O=gpurun_out/s2c3; mkdir -p $O
python bench.py --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
for r in $(seq 1 10); do for v in 2 3 4 5 prod; do
  if [ $v = prod ]; then L=egovlpv2_amd/libegovlp_hip.so; else L=tools/exp_libs/libegovlp_hip_smd$v.so; fi
  EGV_LIB_PATH=$L RUNS=12 REC=0 timeout 300 python tools/repro_check.py > $O/run.txt 2>&1
  nb=$(grep -c "vs 0" $O/run.txt); nbad=$(grep "vs 0" $O/run.txt | grep -vc " 0 of 213")
  echo "smd$v round $r: comparisons $nb bad $nbad"
  grep "elements differ" $O/run.txt | cut -c1-260
done; done 2>&1 | tee $O/stale_diag3.txt
python tools/bf16_bound_probe.py 2>&1 | tail -4 | tee $O/bound_probe.txt
