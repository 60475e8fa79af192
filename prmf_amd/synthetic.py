"""Synthetic inputs of the benchmark shapes: dense positive
X with a low-rank part plus noise, KEGG-sized random tree pathways.  Used by
bench.py, __graft_entry__.smoke() and the large-shape GPU tests."""
import numpy as np

CONFIGS = {
    # name: (m, n, k, P, x_dtype)   -- BASELINE.json configs[1..4]
    "c2": (5000, 10000, 16, 50, "f32"),
    "c3": (50000, 20000, 64, 300, "bf16"),
    "c3f32": (50000, 20000, 64, 300, "f32"),      # the CLI default storage of X
    "c5": (200000, 25000, 128, 1000, "bf16"),
    "c5shard": (25000, 25000, 128, 1000, "bf16"),     # one rank's share of c5 on 8 GPUs
    "c5shardf32": (25000, 25000, 128, 1000, "f32"),   # the same with the CLI's default storage of X
    "c4share8": (6250, 20000, 64, 300, "bf16"),      # one rank's share of c3 on 8 GPUs (c4)
    "c4share4": (12500, 20000, 64, 300, "bf16"),     # ... on 4 GPUs
    "c4share2": (25000, 20000, 64, 300, "bf16"),     # ... on 2 GPUs
    "c5share4": (50000, 25000, 128, 1000, "bf16"),    # one rank's share of c5 on 4 GPUs
    "c5share2": (100000, 25000, 128, 1000, "bf16"),   # ... on 2 GPUs
    # planner corner cases: an 8-wave (512-row panel) sweep over a matrix with very few rows -- X^T of a tall thin X, X of a wide one
    "tallthin": (40000, 700, 64, 20, "bf16"),       # X^T U: 10 MB factor over 700 rows (2 panels)
    "tallthin128": (40000, 600, 128, 20, "bf16"),   # the same at k = 128 (20 MB factor)
    "wideshort": (300, 70000, 64, 50, "bf16"),      # X V: 18 MB factor over 300 rows (1 panel)
    "cli": (500, 20000, 6, 50, "f32"),                # the reference's typical use: samples << genes, default k = 6
}


def tree_pathways(n, P, order, seed):
    """P random recursive trees on `order` distinct genes drawn without replacement
    (after script/gen_sim.py:17-51), unit weights.  Returns [(edges, weights, nodes)]."""
    rs = np.random.RandomState([seed, 104729])
    out = []
    for _ in range(P):
        nodes = rs.choice(n, size=order, replace=False)
        parent = (rs.rand(order - 1) * np.arange(1, order)).astype(np.int64)
        edges = np.stack([nodes[1:], nodes[parent]], axis=1)
        out.append((edges, np.ones(len(edges)), nodes))
    return out


def x_rows_torch(torch, device, seed, n, k, row0, row1, dtype):
    """Rows [row0,row1) of X = (A @ B) * scale + 0.05 * rand, A ~ Gamma(2) (m x r),
    B ~ Gamma(2) (r x n), r = k/2, generated on the GPU in 1024-row blocks with one
    generator stream per block, so any row sharding yields the same global matrix.
    Values lie in (0, ~1]; mimics the post-quantile_transform range (prmf_runner.py:1020)."""
    r = max(2, k // 2)
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1000003)
    B = torch._standard_gamma(torch.full((r, n), 2.0, device=device), generator=g)
    scale = 1.0 / (4.0 * r * 2.5)
    out = torch.empty((row1 - row0, n), device=device, dtype=dtype)
    blk = 1024
    b0 = (row0 // blk) * blk
    while b0 < row1:
        g.manual_seed(seed * 1000003 + 1 + b0 // blk)
        A = torch._standard_gamma(torch.full((blk, r), 2.0, device=device), generator=g)
        noise = torch.rand((blk, n), device=device, generator=g)
        lo, hi = max(b0, row0), min(b0 + blk, row1)
        out[lo - row0:hi - row0] = ((A[lo - b0:hi - b0] @ B) * scale + 0.05 * noise[lo - b0:hi - b0]).to(dtype)
        b0 += blk
    return out


def init_factor_torch(torch, device, seed, rows, k, row0=0, row1=None):
    """3 * (1 - rand) as the reference initialises U and V (prmf_runner.py:651-653);
    block-wise seeded like x_rows_torch."""
    row1 = rows if row1 is None else row1
    g = torch.Generator(device=device)
    out = torch.empty((row1 - row0, k), device=device, dtype=torch.float32)
    blk = 1024
    b0 = (row0 // blk) * blk
    while b0 < row1:
        g.manual_seed(seed * 1000003 + 500000 + b0 // blk)
        u = 3.0 * (1.0 - torch.rand((blk, k), device=device, generator=g))
        lo, hi = max(b0, row0), min(b0 + blk, row1)
        out[lo - row0:hi - row0] = u[lo - b0:hi - b0]
        b0 += blk
    return out


def shard_rows(m, nranks, rank):
    """Contiguous row shard [lo, hi) of rank `rank` (sizes differ by at most 1)."""
    base, rem = divmod(m, nranks)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
