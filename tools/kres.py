"""kernel resource usage of one translation unit: python tools/kres.py egv_gemm3 [extra hipcc flags] -> name, VGPRs, SGPR/VGPR spills, scratch bytes per lane"""
import subprocess, sys, re, os
unit = sys.argv[1]
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'egovlpv2_amd', 'csrc')
extra = sys.argv[2:]
if unit in ('egv_attn_mfma', 'egv_attn_time', 'egv_attn_space', 'egv_attn_cross'):
    extra += ['-mllvm', '-amdgpu-mfma-vgpr-form']
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I../../include', '-Wno-unused-value', '-Rpass-analysis=kernel-resource-usage',
       '-c', unit + '.hip', '-o', '/tmp/kres_%s.o' % unit] + extra
out = subprocess.run(cmd, cwd=src, capture_output=True, text=True).stderr
cur = {}
for ln in out.splitlines():
    m = re.search(r'remark: +(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass', ln)
    if not m:
        if 'error' in ln: print(ln)
        continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name':
        cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()}
    cur[k] = v
    if k.startswith('LDS'):
        print(f"{cur['name'][:110]:110s} VGPR {cur.get('VGPRs')} AGPR {cur.get('AGPRs')} spill s{cur.get('SGPRs Spill')}/v{cur.get('VGPRs Spill')} scratch {cur.get('ScratchSize [bytes/lane]')}")
