import os, sys, torch.multiprocessing as mp
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import test_multirank_gpu as T
if __name__ == '__main__':
    overlap = sys.argv[1] == '1'; gsync = sys.argv[2]
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    procs = [ctx.Process(target=T._worker, args=(r, 2, 29811, q, 3, overlap, gsync)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=900) for _ in procs]
    [p.join(timeout=60) for p in procs]
    for rank, status, info in res:
        print('RANK', rank, status, {k: v for k, v in info.items() if k != 'bad'} if isinstance(info, dict) else info)
        if isinstance(info, dict):
            for b in info.get('bad', []): print('   ', b)
