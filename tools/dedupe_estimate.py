"""development tool: what merging identical super-k-mers inside a list would save on C2-like data (30x coverage, 1 % substitutions, k = 31, m = 16):
the share of k-mers that lie in unique (minimizer, bases) records.  HISTORY.md section 6, "What was tried and dropped"."""
import numpy as np, sys
from numpy.lib.stride_tricks import sliding_window_view
rng = np.random.default_rng(1)
G = 200_000; cov = 30; L = 150; k = 31; m = 16; err = 0.01
n = G * cov // L
genome = rng.integers(0, 4, G, dtype=np.uint8)
starts = rng.integers(0, G - L, n)
strand = rng.integers(0, 2, n)
total = 0; uniq = {}; recs = 0
w = k - m + 1   # m-mer offsets per k-mer = 16
pw = 4 ** np.arange(m - 1, -1, -1, dtype=np.uint64)
def mix(x):
    x = (x ^ (x >> np.uint64(33))) * np.uint64(0xff51afd7ed558ccd); x = (x ^ (x >> np.uint64(33))) * np.uint64(0xc4ceb9fe1a85ec53); return x ^ (x >> np.uint64(33))
kept_after_truncation = 0
for r in range(n):
    s = genome[starts[r]:starts[r] + L].copy()
    e = rng.random(L) < err
    s[e] = (s[e] + rng.integers(1, 4, e.sum())) % 4
    if strand[r]: s = (3 - s)[::-1]
    mm = sliding_window_view(s, m).astype(np.uint64)          # L-m+1 m-mers
    f = (mm * pw).sum(1); rc = ((3 - mm[:, ::-1]) * pw).sum(1)
    h = mix(np.minimum(f, rc))
    mins = sliding_window_view(h, w).min(1)                    # per k-mer (L-k+1)
    nk = mins.size; total += nk
    brk = np.flatnonzero(np.diff(mins) != 0) + 1
    segs = np.split(np.arange(nk), brk)
    for sg in segs:
        a, b = sg[0], sg[-1]
        key = (int(mins[a]), bytes(s[a:b + k]))
        recs += 1
        uniq[key] = uniq.get(key, 0) + 1
uk = sum(len(kb[1]) - k + 1 for kb in uniq)
print("reads", n, "k-mers", total, "records", recs, "avg n %.2f" % (total / recs), "unique records", len(uniq), "k-mers in unique records", uk, "ratio %.3f" % (uk / total))
# distribution of multiplicities
import collections
c = collections.Counter(uniq.values()); print(sorted(c.items())[:12])
