"""include/kmernator_amd_shim.hpp -- the binding a Kmernator maintainer adds on the reference's side -- compiled as C++03
against tests/cpp/mock_kmernator.h, a MOCK that declares the reference names the shim uses with their signatures and access
levels (the real headers need Boost 1.53 / sparsehash, absent here).  This catches syntax, access-control and ownership
mistakes in the shim (the reference copies and assigns spectra by value, apps/FilterReads.cpp:126,136); it pins nothing about
the reference.  On a GPU the demo also runs: shim -> C-ABI -> image -> mock map, counts equal to the oracle's."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, KMR_MAP_WEAK, ROOT, OracleSpectrum, default_config, read_fastq

CPP = os.path.join(ROOT, "tests", "cpp")
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "kmernator_amd", "csrc")
DEMO = os.path.join(CPP, "shim_demo")


def build_demo():
    src = os.path.join(CPP, "shim_demo.cpp")
    deps = [src, os.path.join(CPP, "mock_kmernator.h"), os.path.join(INC, "kmernator_amd_shim.hpp"), os.path.join(INC, "kmernator_amd.h")]
    if not os.path.exists(DEMO) or any(os.path.getmtime(d) > os.path.getmtime(DEMO) for d in deps):
        subprocess.check_call(["g++", "-std=c++03", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + INC, "-o", DEMO, src,
                               "-L" + LIBDIR, "-lkmernator_amd", "-Wl,-rpath,$ORIGIN/../../kmernator_amd/csrc"])
    return DEMO


def test_shim_compiles_and_links_as_cxx03_against_the_mock():
    build_demo()


def test_mpi_flavour_of_the_shim_instantiates():
    """GpuDistributedKmerSpectrum (MPI_Alltoallv exchange around the two host-buffer halves of the C-ABI): every line seen by
    the compiler, with the real <mpi.h> when the image has one"""
    mpi_inc = None
    for d in ("/opt/conda/include", "/usr/include/x86_64-linux-gnu/mpi", "/usr/include/mpi", "/usr/lib/x86_64-linux-gnu/openmpi/include"):
        if os.path.exists(os.path.join(d, "mpi.h")):
            mpi_inc = d
            break
    if mpi_inc is None:
        pytest.skip("no mpi.h in this image")
    subprocess.check_call(["g++", "-std=c++03", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + INC, "-I" + mpi_inc,
                           os.path.join(CPP, "shim_mpi_check.cpp")])


def test_c_header_is_plain_c():
    """the boundary is a C ABI: the header must compile as C99"""
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", "-I" + INC, "-"],
                       input='#include "kmernator_amd.h"\nint main(void) { return (int)sizeof(kmr_config) == 0; }\n', text=True, capture_output=True)
    assert p.returncode == 0, p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("k,fq,start", [(31, "1000.std.fastq", 33), (21, "1000.fastq", 64)])
def test_shim_demo_runs_the_filterreads_stanza(k, fq, start):
    """FilterReads.cpp:126-140 with GKS in place of KS: counts restored into the (mock) map equal the oracle's weak map"""
    demo = build_demo()
    path = os.path.join(GOLDEN, fq)
    p = subprocess.run([demo, str(k), path, str(start)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = {}
    for line in p.stdout.splitlines():
        key, cnt = line.split()
        got[key] = int(cnt)
    rb = read_fastq(path)
    cfg = default_config(k, fastq_start_char=start, estimated_raw_kmers=(76 - k + 1) * 1000)
    o = OracleSpectrum(cfg)
    o.add_reads(rb)
    o.finalize(2)
    keys, counts, _, _, _ = o.entries()
    exp = {bytes(kk).hex(): int(c) for kk, c in zip(keys, counts)}
    assert got == exp
    st = o.stats()
    assert "raw %d good %d unique %d singleton %d weak %d" % (st["raw_kmers"], st["raw_good_kmers"], st["unique_kmers"], st["singleton_kmers"], st["weak_entries"]) in p.stderr
    # the size history the reference object now holds (FilterReads.cpp:141-147 writes it out): the oracle's per-k-mer history, the reference's own
    hist = [tuple(int(v) for v in line.split("\t")[1:]) for line in p.stderr.splitlines() if line.startswith("history\t")]
    want = [tuple(int(v) for v in e) for e in o.size_tracker(per_read=False, force_last=True)]
    assert hist == want and len(hist) > 100
