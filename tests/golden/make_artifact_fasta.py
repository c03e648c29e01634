#!/usr/bin/env python3
"""Writes the artifact / simple-repeat / PhiX sequence tables the reference screens reads against as FASTA
fixtures (tests/golden/artifact_sequences.fa, simple_repeats.fa, phix.fa).

They are the string constants of FilterKnownOddities::getArtifactFasta / getSimpleRepeatFasta / getPhiX
(src/FilterKnownOddities.h:742-805, 811-1306, 1316-1397): sequence data, read out of the reference tree here
because the product takes the sequences as an input (kmr_artifact_filter_create) and the golden FilterReads
outputs were made with exactly these.  Run in the build container: python tests/golden/make_artifact_fasta.py"""
import os
import re
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/FilterKnownOddities.h"
HERE = os.path.dirname(os.path.abspath(__file__))


def table(text, fn):
    start = text.index("static std::string %s()" % fn)
    end = text.index("return ss.str();", start)
    return "".join(m + "\n" for m in re.findall(r'ss << "([^"]*)"', text[start:end]))


text = open(SRC).read()
for fn, out in (("getArtifactFasta", "artifact_sequences.fa"), ("getSimpleRepeatFasta", "simple_repeats.fa"), ("getPhiX", "phix.fa")):
    data = table(text, fn)
    open(os.path.join(HERE, out), "w").write(data)
    print(out, data.count(">"), "sequences", len(data), "bytes")
