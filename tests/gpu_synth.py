"""Synthetic inputs of the BASELINE.json configurations, generated in HBM: uniform
i.i.d. symbols, or genomes with their own base composition each (i.i.d. uniform genomes of 3 Mb all
have the same k-mer spectrum to within 1e-3 and are a degenerate input)."""
import numpy as np


def synth_device(nseq, lo, hi, seed, composition=False, device="cuda:0", batch=256):
    """-> (uint8 torch tensor of all symbols (+16 bytes of slack), uint64 offsets[nseq + 1])"""
    import torch

    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, size=nseq, dtype=np.int64) if hi > lo else np.full(nseq, lo, np.int64)
    offsets = np.zeros(nseq + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    total = int(offsets[-1])
    if not composition:
        seqs = torch.randint(0, 4, (total + 16,), dtype=torch.uint8, device=device, generator=g)
    else:
        seqs = torch.zeros(total + 16, dtype=torch.uint8, device=device)
        cuts = np.minimum(255, np.cumsum(rng.dirichlet([4.0] * 4, size=nseq), axis=1)[:, :3] * 256).astype(np.uint8)
        for b0 in range(0, nseq, batch):  # the random bytes of a batch of genomes at a time
            b1 = min(nseq, b0 + batch)
            base = int(offsets[b0])
            r = torch.randint(0, 256, (int(offsets[b1]) - base,), dtype=torch.uint8, device=device, generator=g)
            for i in range(b0, b1):
                a, b = int(offsets[i]), int(offsets[i + 1])
                x = r[a - base:b - base]
                seqs[a:b] = (x >= int(cuts[i, 0])).to(torch.uint8) + (x >= int(cuts[i, 1])).to(torch.uint8) + \
                    (x >= int(cuts[i, 2])).to(torch.uint8)
            del r
    torch.cuda.synchronize()
    return seqs, offsets
