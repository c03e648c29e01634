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


def passes_length(length, read_length, minimum_length):
    """ReadSelectorUtil::passesLength (src/ReadSelector.h:209-228): a minimum <= 1 is a fraction of the read, above 1 a length"""
    if length <= 1.0:
        return False
    if minimum_length <= 1.0:
        return read_length * minimum_length <= length
    return minimum_length <= length


def filterreads_output(names, seqs, quals, labels, discarded, trim_off, trim_len, scores, min_depth, min_read_length, both_pass, qual_shift=0, out_base=33):
    """What FilterReads writes for a paired read set (apps/FilterReads.h:159-260 with max-kmer-depth and partition-by-depth off):
    pickAllPassingPairs (src/ReadSelector.h:585-596) over the pairs (2i, 2i+1) -- a pair passes when both (min-passing-in-pair 2)
    or either of its reads isPassingRead (:550-568: score >= min depth and passesLength of the trimmed length against the read as
    the artifact filter left it); BOTH reads of a passing pair are picked (pickIfNew only asks for availability) and written in
    pair order with their own trims (writePicks :1242-1262).  A read whose trim is <= 1 base, or that the artifact filter discarded,
    prints as one 'N' with quality (output base) + 1 (Sequence::getFasta src/Sequence.cpp:305-311, Read::getQuals :729-733); a discarded read
    was never scored (scoreAndTrimReads :1195-1197) and has no label.  Per-read inputs are the reads AFTER the artifact filter."""
    n = len(names)
    assert n % 2 == 0
    passing = []
    for i in range(n):
        if discarded[i]:
            passing.append(False)
        else:
            passing.append(scores[i] >= min_depth and passes_length(float(trim_len[i]), len(seqs[i]), min_read_length))
    out = []
    for p in range(n // 2):
        a, b = 2 * p, 2 * p + 1
        ok = (passing[a] and passing[b]) if both_pass else (passing[a] or passing[b])
        if not ok:
            continue
        for i in (a, b):
            tl = 0 if discarded[i] else int(trim_len[i])
            if discarded[i] or tl <= 1:
                s, q = b"N", bytes([out_base + 1])
            else:
                to = int(trim_off[i])
                s = seqs[i][to:to + tl]
                q = bytes((c + qual_shift) & 0xff for c in quals[i][to:to + tl])
            head = names[i] + ((b" " + labels[i]) if labels[i] else b"")
            out.append(b"@" + head + b"\n" + s + b"\n+\n" + q + b"\n")
    return b"".join(out)
