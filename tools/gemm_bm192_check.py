"""192-row-tile variant of the persistent GEMM (csrc/egv_gemm3.hip, IM = 3) against the 256-row one: bitwise equality of the
outputs on the N = 768 hot-path shapes (+ a ragged M) and time per call.  The tile height is chosen per process
(EGV_PP_BM192), so the script re-executes itself once per setting."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [('fc2', 25096, 768, 3072, 'res'), ('fc1_dgrad', 25096, 768, 3072, 'plain'), ('qkv_dgrad', 25096, 768, 2304, 'plain'),
         ('proj_dgrad', 25096, 768, 768, 'plain'), ('ragged', 24999, 768, 1536, 'res'), ('b3', 9411, 768, 3072, 'plain'),
         ('wide', 25096, 2304, 768, 'bias')]


def run(tag):
    import torch
    from egovlpv2_amd import hipops as ops
    dev = 'cuda'
    res = {}
    for ci, (name, M, N, K, kind) in enumerate(CASES):
        g = torch.Generator(device=dev).manual_seed(100 + ci)
        x = torch.randn(M, K, device=dev, generator=g).bfloat16()
        w = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
        b = torch.randn(N, device=dev, generator=g)
        r1 = torch.randn(M, N, device=dev, generator=g).bfloat16()
        y = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
        kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
        f = {'res': lambda: ops.gemm(x, w, y, bias=b, res1=r1, **kw), 'plain': lambda: ops.gemm(x, w, y, **kw),
             'bias': lambda: ops.gemm(x, w, y, bias=b, **kw)}[kind]
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        ref = x[:512].float() @ w.float().t() + (b if kind != 'plain' else 0) + (r1[:512].float() if kind == 'res' else 0)
        err = ((y[:512].float() - ref).abs().max() / ref.abs().max()).item()
        torch.save(y.cpu(), f'/tmp/bm192_{tag}_{name}.pt')
        res[name] = dict(us=round(ms * 1e3, 1), tf=round(2 * M * N * K / ms / 1e9, 1), err=err)
    print(json.dumps(res))


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        import torch
        out = {}
        for tag in ('0', '1'):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), tag], capture_output=True, text=True, env=dict(os.environ, EGV_PP_BM192=tag))
            try:
                out[tag] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:
                print('FAILED', tag, r.stdout[-1000:], r.stderr[-3000:])
                sys.exit(1)
        for name, *_ in CASES:
            same = torch.equal(torch.load(f'/tmp/bm192_0_{name}.pt'), torch.load(f'/tmp/bm192_1_{name}.pt'))
            a, b = out['0'][name], out['1'][name]
            print(f"{name:10s} 256-row {a['us']:7.1f} us {a['tf']:6.1f} TF err {a['err']:.1e} | 192-row {b['us']:7.1f} us {b['tf']:6.1f} TF err {b['err']:.1e} | bitwise equal: {same}")
