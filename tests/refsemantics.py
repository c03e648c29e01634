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
