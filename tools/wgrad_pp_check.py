"""Correctness + timing of the ping-pong weight-gradient kernel (csrc/egv_gemm4.hip) vs torch fp32 and vs the ring kernel."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from egovlpv2_amd import hipops as ops
    dev = 'cuda'
    torch.manual_seed(0)

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    out = {}
    for name, M, N, K in [('qkv', 25096, 2304, 768), ('proj', 25096, 768, 768), ('fc1', 25096, 3072, 768), ('fc2', 25096, 768, 3072),
                          ('patch', 25088, 768, 768), ('short', 5000, 768, 768)]:
        dy = torch.randn(M, N, device=dev).bfloat16()
        x = torch.randn(M, K, device=dev).bfloat16()
        f = lambda: ops.wgrad(dy, x, M, N, K, bias=True)
        ms = timeit(f)
        dw, db = f()
        ref = dy.float().t() @ x.float()
        rb = dy.float().sum(0)
        out[name] = dict(us=round(ms * 1e3, 1), tf=round(2 * M * N * K / ms / 1e9, 1), err=((dw - ref).abs().max() / ref.abs().max()).item(),
                         berr=((db - rb).abs().max() / rb.abs().max()).item())
    print(json.dumps(out))


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run()
    else:
        res = {}
        for pp in ('0', '1'):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], capture_output=True, text=True, env=dict(os.environ, EGV_WGRAD_PP=pp))
            try:
                res[pp] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:
                print('FAILED', pp, r.stdout[-1500:], r.stderr[-2500:])
                res[pp] = {}
        for n in res['0']:
            a, b = res['0'][n], res['1'].get(n, {})
            print(f"{n:6s} ring {a['us']:7.1f} us {a['tf']:6.1f} TF err {a['err']:.1e}/{a['berr']:.1e} | pp {b.get('us', 0):7.1f} us {b.get('tf', 0):6.1f} TF err {b.get('err', -1):.1e}/{b.get('berr', -1):.1e}")
