import time, numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
m, n = 50000, 20000
dev = "cuda:0"
X = np.random.RandomState(0).rand(m, n).astype(np.float32)
torch.zeros(1, device=dev); torch.cuda.synchronize()
def t(f, name):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); print("%-60s %.3f s" % (name, time.perf_counter() - t0)); return r
a = t(lambda: torch.from_numpy(X).to(device=dev, dtype=torch.bfloat16), "baseline .to(device, bf16)")
b = t(lambda: torch.from_numpy(X).to(device=dev).to(torch.bfloat16), "pageable f32 copy + device convert")
def chunked(rows):
    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    for r0 in range(0, m, rows):
        out[r0:r0 + rows] = torch.from_numpy(X[r0:r0 + rows]).to(device=dev)
    return out
c = t(lambda: chunked(2048), "pageable chunks of 2048 rows + device convert")
def pipeline(rows, nthreads, nbuf=3):
    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    t0 = time.perf_counter()
    pinned = [torch.empty((rows, n), dtype=torch.float32).pin_memory() for _ in range(nbuf)]
    tp = time.perf_counter() - t0
    evs = [None] * nbuf
    st = torch.cuda.Stream()
    pool = ThreadPoolExecutor(nthreads)
    for i, r0 in enumerate(range(0, m, rows)):
        j = i % nbuf
        if evs[j] is not None: evs[j].synchronize()
        r1 = min(m, r0 + rows)
        pn = pinned[j].numpy()
        cuts = np.linspace(r0, r1, nthreads + 1).astype(int)
        list(pool.map(lambda ab: np.copyto(pn[ab[0] - r0:ab[1] - r0], X[ab[0]:ab[1]]), zip(cuts[:-1], cuts[1:])))
        with torch.cuda.stream(st):
            out[r0:r1].copy_(pinned[j][:r1 - r0].to(dev, non_blocking=True))
            evs[j] = torch.cuda.Event(); evs[j].record(st)
    st.synchronize()
    pool.shutdown()
    return out, tp
for rows, nt in ((1024, 8), (2048, 8), (2048, 16), (4096, 16), (1024, 4)):
    (d, tp) = t(lambda: pipeline(rows, nt), "pinned pipeline rows=%d threads=%d" % (rows, nt)); print("    (pinned alloc %.3f s)" % tp)
    assert torch.equal(d, a)
X64 = X.astype(np.float64)
t(lambda: torch.from_numpy(X64).to(device=dev, dtype=torch.float32), "float64 source: baseline .to(device, f32)")
