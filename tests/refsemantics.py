"""Pure-Python restatement of the read-scoring consumer used to pin k=31 counts
against the reference's FilterReads golden output (test/1000-Filtered.fastq):
ReadSelector::trimReadByMinimumKmerScore (src/ReadSelector.h:949-1014),
scoreReadByMedianKmer (:1123-1147) and setTrimHeaders (:1015-1036).
Test infrastructure only."""


def median_trim_label(counts, k, min_depth=2):
    vals = [float(c) if c >= min_depth else 0.0 for c in counts]   # setKmerValues :1064-1076
    best = (0, 0, 0)
    off = ln = sc = 0
    for v in vals:
        if v >= min_depth:
            ln += 1
            sc += 1
        else:
            if sc > best[2]:
                best = (off, ln, sc)
            sc = 0
            off += ln + 1
            ln = 0
    if sc > best[2]:
        best = (off, ln, sc)
    toff, tlen = best[0], best[1]
    trimmed = tlen < len(vals)
    if tlen > 0:
        run = sorted(vals[toff:toff + tlen])
        score = run[len(run) // 2]
        tl = tlen + k - 1
    else:
        score, toff, tl = -1.0, 0, 0
    label = b""
    if trimmed:
        label += b"Trim:%d+%d " % (toff, tl)
    label += b"MedianScore:%d" % int(score + 0.5)
    return label


def score_and_trim(counts, seq, k, min_score, scoring):
    """ReadSelector::scoreAndTrimReads for one read (src/ReadSelector.h:949-1207), returning
    (trim_offset, trim_length_in_bases, score, was_trimmed).  counts = weak-map count per k-mer position."""
    n = len(counts)
    for i, c in enumerate(seq):                           # firstMarkupNorX + _setNumKmers
        ch = chr(c) if isinstance(c, int) else c
        if ch in "NX.":
            m = i + 1
            n = min(n, m - k) if m > k else 0
            break
    vals = [float(c) for c in counts[:n]]
    best = (0, 0)
    off = ln = 0
    for v in vals:
        if v >= min_score:
            ln += 1
        else:
            if ln > best[1]:
                best = (off, ln)
            off += ln + 1
            ln = 0
    if ln > best[1]:
        best = (off, ln)
    toff, tlen = best
    trimmed = tlen < n
    if tlen == 0:
        return 0, 0, -1.0, trimmed
    run = vals[toff:toff + tlen]
    if scoring == "MEDIAN":
        sc = sorted(run)[len(run) // 2]
    elif scoring == "AVG":
        import numpy as np
        sc = float(np.float32(sum(run) / len(run)))
    elif scoring == "MIN":
        sc = min(run)
    elif scoring == "MAX":
        sc = max(run)
    else:
        sc = 0.0
    return toff, tlen + k - 1, sc, trimmed
