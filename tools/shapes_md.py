"""stderr of `EGV_BENCH_SHAPES=1 python bench.py ...` -> markdown table of the in-step per-shape GEMM timings
usage: python tools/shapes_md.py <stderr file> > profiles/roundN_gemm_shapes_instep.md"""
import re
import sys

KIND = {0: 'gemm_kernel bf16 NT', 7: 'gemm_skinny_kernel (<= 16 rows)', 9: 'wgrad_small_m_kernel (<= 16 rows)', 6: 'gemm_kernel fp32 TN', 1: 'gemm_kernel bf16 NN', 4: 'gemm_kernel fp32', 5: 'gemm_kernel fp32 NN', 8: 'gemm_ring_kernel<256x128>',
        12: 'gemm_pp_kernel (persistent, fwd+dgrad)', 13: 'gemm_ring_kernel<128x128>', 14: 'gemm_wgrad_pp_kernel (one gradient per launch)',
        15: 'gemm_wgrad_group_kernel (all gradients of a block call)', 16: 'gemm_pp_kernel MX-fp8',
        2: 'gemm_kernel bf16 TN', 10: 'gemm_wgrad_ring_kernel', 20: 'attention forward (not a GEMM: GFLOP = QK^T + PV)', 21: 'attention dQ', 22: 'attention dK/dV',
        23: 'attention one-pass backward (space)', 30: 'layernorm forward (HBM-bound)', 31: 'layernorm backward (HBM-bound)', 32: 'quant_mx (HBM-bound)'}
rows = {}
for line in open(sys.argv[1]):
    m = re.match(r'shape\[(\w+)\] kind=(\d+) gflop=\s*([\d.]+) n/step=\s*([\d.]+) ms/step=\s*([\d.]+) avg_us=\s*([\d.]+) TF=\s*([\d.]+)', line)
    if m:
        rows.setdefault(m.group(1), []).append((int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6)), float(m.group(7))))
print("# In-step per-shape GEMM timings (HIP events around every GEMM launch inside `bench.py`, configs[2], bf16)\n")
print("Command: `EGV_BENCH_SHAPES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2> shapes.txt; python tools/shapes_md.py shapes.txt`.  in_step = the "
      "training step as it runs (two companion streams); isolated = `EGV_NO_OVERLAP=1` (no co-scheduled kernels).  A row is one (kernel, FLOPs per launch) "
      "class; TF = FLOPs / mean launch time.  Grouped weight-gradient launches are classed by the FLOPs of the whole group.\n")
for tag, rs in rows.items():
    print(f"\n## {tag}\n")
    print("| kernel | GFLOP / launch | launches / step | ms / step | mean us | TFLOP/s | frac of 2.5 PF |")
    print("|---|---|---|---|---|---|---|")
    for kd, gf, n, ms, us, tf in rs:
        print(f"| `{KIND.get(kd, kd)}` | {gf:.2f} | {n:.0f} | {ms:.2f} | {us:.1f} | {tf:.0f} | {tf / 2500:.3f} |")
    tot_ms = sum(r[3] for r in rs if r[0] < 20)
    tot_fl = sum(r[1] * r[2] for r in rs if r[0] < 20)
    print(f"\nGEMM family total: {tot_ms:.1f} ms/step, {tot_fl / tot_ms:.0f} TFLOP/s average")
