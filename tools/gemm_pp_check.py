"""Correctness + timing of the persistent ping-pong GEMM (csrc/egv_gemm3.hip) against torch fp32 matmul on bf16-rounded
inputs and against the 256x128 ring kernel, on the hot-path shapes (run on the GPU box).
The kernel choice is made per process by EGV_GEMM_PP, so this script re-executes itself once per setting."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from egovlpv2_amd import hipops as ops
    from egovlpv2_amd import _lib as L
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
    cases = [  # name, M, N, K, kind
        ('qkv', 25096, 2304, 768, 'bias'), ('proj', 25096, 768, 768, 'res'), ('fc1', 25096, 3072, 768, 'gelu_pre'),
        ('fc2', 25096, 768, 3072, 'res'), ('fc2_dgrad', 25096, 3072, 768, 'dact'), ('fc1_dgrad', 25096, 768, 3072, 'plain'),
        ('i2t', 25096, 768, 768, 'gate_res2'), ('ragged', 4000, 1032, 256, 'res'), ('small', 777, 520, 128, 'bias'),
        ('infer', 100368, 2304, 768, 'bias')]
    for name, M, N, K, kind in cases:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev)
        r1 = torch.randn(M, N, device=dev).bfloat16()
        r2 = torch.randn(M, N, device=dev).bfloat16()
        aux = torch.randn(M, N, device=dev).bfloat16()
        gate = torch.tensor([0.37], device=dev)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        pre = torch.empty_like(y)
        kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
        if kind == 'bias':
            f = lambda: ops.gemm(x, w, y, bias=b, **kw)
        elif kind == 'plain':
            f = lambda: ops.gemm(x, w, y, **kw)
        elif kind == 'res':
            f = lambda: ops.gemm(x, w, y, bias=b, res1=r1, **kw)
        elif kind == 'gelu_pre':
            f = lambda: ops.gemm(x, w, y, bias=b, act=L.ACT_GELU, pre=pre, **kw)
        elif kind == 'dact':
            f = lambda: ops.gemm(x, w, y, aux=aux, dact=L.ACT_GELU, **kw)
        elif kind == 'gate_res2':
            f = lambda: ops.gemm(x, w, y, bias=b, gate=gate, res1=r1, res2=r2, **kw)
        ms = timeit(f)
        # reference on a row sample (full reference for the small ones)
        idx = torch.arange(M, device=dev) if M <= 4000 else torch.cat([torch.arange(0, 300, device=dev), torch.randint(0, M, (1500,), device=dev), torch.arange(M - 300, M, device=dev)])
        acc = x[idx].float() @ w.float().t()
        if kind in ('bias', 'res', 'gelu_pre', 'gate_res2'):
            acc = acc + b
        ref_pre = acc.clone()
        if kind == 'gelu_pre':
            acc = torch.nn.functional.gelu(acc)
        if kind == 'gate_res2':
            acc = acc * gate + r1[idx].float() + r2[idx].float()
        if kind == 'res':
            acc = acc + r1[idx].float()
        if kind == 'dact':
            a = aux[idx].float()
            acc = acc * (0.5 * (1 + torch.erf(a / 2 ** 0.5)) + a * torch.exp(-0.5 * a * a) / (2 * 3.141592653589793) ** 0.5)
        err = ((y[idx].float() - acc).abs().max() / acc.abs().max()).item()
        perr = ((pre[idx].float() - ref_pre).abs().max() / ref_pre.abs().max()).item() if kind == 'gelu_pre' else 0.0
        out[name] = dict(us=round(ms * 1e3, 1), tf=round(2 * M * N * K / ms / 1e9, 1), err=err, pre_err=perr)
    print(json.dumps(out))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        run()
    else:
        res = {}
        for pp in ('0', '1'):
            env = dict(os.environ, EGV_GEMM_PP=pp)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], capture_output=True, text=True, env=env)
            try:
                res[pp] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:
                print('FAILED pp=' + pp, r.stdout[-2000:], r.stderr[-3000:])
                res[pp] = {}
        for name in res['0']:
            a, b = res['0'][name], res['1'].get(name, {})
            print(f"{name:10s} ring {a['us']:8.1f} us {a['tf']:7.1f} TF err {a['err']:.2e} | pp {b.get('us', 0):8.1f} us {b.get('tf', 0):7.1f} TF err {b.get('err', -1):.2e} pre_err {b.get('pre_err', -1):.2e}")
