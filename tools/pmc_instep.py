"""Summarise in-step rocprofv3 --pmc passes over bench.py into profiles/round2_pmc_instep.{json,md}.

usage: pmc_instep.py <fetch.db> <write.db> <sq.db> <trace.db> <steps_in_each_run> <out_prefix>
  fetch.db : rocprofv3 --pmc FETCH_SIZE --kernel-trace  -- python bench.py ...        (separate passes, as
  write.db : rocprofv3 --pmc WRITE_SIZE --kernel-trace  -- python bench.py ...         MI355X_MICROARCH.md prescribes)
  sq.db    : rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python bench.py ...
  trace.db : rocprofv3 --kernel-trace --stats -- python bench.py ...   (launch counts, durations without counters)
FETCH_SIZE / WRITE_SIZE are in KB per dispatch; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced stream, so
traffic = 2 * FETCH + WRITE (an upper bound of HBM bytes: Infinity-Cache hits are included)."""
import sqlite3, sys, json, re, collections

GROUPS = [('gemm_pp_kernel', r'gemm_pp_kernel'), ('gemm_ring_kernel<256x128>', r'gemm_ring_kernel<egv::Cfg<4, 2, 4, 4>'),
          ('gemm_ring_kernel<128x128>', r'gemm_ring_kernel<egv::Cfg<4, 2, 2, 4>'), ('gemm_wgrad_ring_kernel', r'gemm_wgrad_ring_kernel'), ('gemm_wgrad_pp_kernel', r'gemm_wgrad_pp_kernel'),
          ('gemm_wgrad_group_kernel', r'gemm_wgrad_group_kernel'), ('attn time fwd/dq/dkv (17 keys)', r'attn_(fwd|dq|dkv)_mfma_kernel<2, 1, 0, 0>'),
          ('attn_fwd_mfma (space, 196+1 keys)', r'attn_fwd_mfma_kernel<14, 4, 0, 13>'), ('attn_dq_mfma (space)', r'attn_dq_mfma_kernel<14, 4, 0, 0>'),
          ('attn_dkv_mfma (space)', r'attn_dkv_mfma_kernel<14, 4, 0, 0>'), ('attn_bwd_fused (space: dQ+dK+dV)', r'attn_bwd_fused_kernel'),
          ('attn_time_fwd (round 4: one wave per group, CLS query folded in)', r'attn_time_fwd_kernel'),
          ('attn_time_bwd (round 4: one-launch backward)', r'attn_time_bwd_kernel'),
          ('attn_space_fwd (round 4: row-major LDS images)', r'attn_space_fwd_kernel'),
          ('attn_space_bwd (round 4: two-phase one-launch backward)', r'attn_space_bwd_kernel'),
          ('attn_fewkeys_fwd (round 5: image -> text, 25 096 queries over 32 keys)', r'attn_fewkeys_fwd_kernel'),
          ('attn_fewkeys_bwd (round 5)', r'attn_fewkeys_bwd_kernel'),
          ('attn_fewq_fwd (round 5: text -> image, 32 queries over 25 096 keys)', r'attn_fewq_fwd_kernel'),
          ('attn_fewq_bwd (round 5)', r'attn_fewq_bwd_kernel'),
          ('sum_ln (round 4: fp32 residual sums + LayerNorm of the video tower)', r'sum_ln_kernel'), ('layernorm_fwd', r'layernorm_fwd_kernel'), ('layernorm_bwd', r'layernorm_bwd(_bf16)?_kernel'), ('reduce_slabs', r'reduce_slabs_kernel')]


def group_of(name):
    for g, pat in GROUPS:
        if re.search(pat, name):
            return g
    return None


def counters(path):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(duration) from counters_collection "
                      "group by kernel_name, counter_name, dispatch_id").fetchall()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, c, d, v, du in rows:
        g = group_of(k)
        if g:
            agg[g][c].append(v)
            agg[g]['_dur_ns_' + c].append(du)
    return agg


fetch, write, sq, trace, steps, out = sys.argv[1:7]
steps = float(steps)
F, W, S = counters(fetch), counters(write), counters(sq)
tdb = sqlite3.connect(trace)
cols = [c[1] for c in tdb.execute("pragma table_info('kernels')")]
namecol = 'name' if 'name' in cols else 'kernel_name'
trows = tdb.execute(f"select {namecol}, start, end from kernels").fetchall()
launches = len(trows) / steps
tg = collections.defaultdict(list)
for n, s, e in trows:
    g = group_of(n)
    if g:
        tg[g].append((e - s) / 1e3)
res = {'launches_per_step': round(launches), 'steps_per_run': steps, 'kernels': {}}
md = ["# In-step rocprofv3 PMC passes over `bench.py` (configs[2], B=8, 16x224^2, bf16; one pass per counter group)\n",
      "| kernel | launches/step | avg us (trace) | FETCH KB/launch | WRITE KB/launch | traffic MB/launch (2*FETCH+WRITE) | MFMA busy / (SQ busy) | MFMA busy frac of kernel time |",
      "|---|---|---|---|---|---|---|---|"]
for g, _ in GROUPS:
    f = F[g].get('FETCH_SIZE', [])
    w = W[g].get('WRITE_SIZE', [])
    mb = S[g].get('SQ_VALU_MFMA_BUSY_CYCLES', [])
    sb = S[g].get('SQ_BUSY_CYCLES', [])
    ga = S[g].get('GRBM_GUI_ACTIVE', [])
    if not f and not tg[g]:
        continue
    fk = sum(f) / len(f) if f else None
    wk = sum(w) / len(w) if w else None
    ent = {'launches_per_step': round(len(tg[g]) / steps, 1), 'avg_us': round(sum(tg[g]) / max(1, len(tg[g])), 1)}
    if fk is not None and wk is not None:
        ent.update(fetch_bytes_per_launch=round(2 * fk * 1024), write_bytes_per_launch=round(wk * 1024),
                   traffic_bytes_per_launch=round((2 * fk + wk) * 1024),
                   method='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate in-step passes over bench.py; traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction)')
    frac = None
    if mb and ga:
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs that report it; GRBM_GUI_ACTIVE over the XCDs: normalise to
        # busy cycles per SIMD / active cycles per XCD (1024 SIMDs, 8 XCDs)
        frac = (sum(mb) / len(mb) / 1024.0) / (sum(ga) / len(ga) / 8.0)
        ent['mfma_busy_frac'] = round(frac, 4)
    res['kernels'][g] = ent
    md.append(f"| `{g}` | {ent['launches_per_step']} | {ent['avg_us']} | {fk and round(fk)} | {wk and round(wk)} | "
              f"{ent.get('traffic_bytes_per_launch', 0) / 1e6:.1f} | {(sum(mb) / len(mb)) if mb else 0:.3g} / {(sum(sb) / len(sb)) if sb else 0:.3g} | {frac if frac is None else round(frac, 3)} |")
json.dump(res, open(out + '.json', 'w'), indent=1)
open(out + '.md', 'w').write('\n'.join(md) + f"\n\nTotal kernel launches per step (kernel trace): {launches:.0f}\n")
print('\n'.join(md))
