"""The C++ host side above the C-ABI (include/kmernator_amd.hpp) driven the way the reference's apps drive KmerSpectrum
(tests/cpp/host_demo.cpp): MeraculousCounter against the reference's phiX goldens and FilterReads' trim/score labels
against test/1000-Filtered.fastq, FASTQ text in, everything between on the device."""
import os
import subprocess

import pytest

from helpers import GOLDEN, ROOT, read_fastq

DEMO = os.path.join(ROOT, "tests", "cpp", "host_demo")
SRC = os.path.join(ROOT, "tests", "cpp", "host_demo.cpp")
LIBDIR = os.path.join(ROOT, "kmernator_amd", "csrc")


def build_demo():
    deps = [SRC, os.path.join(ROOT, "include", "kmernator_amd.hpp"), os.path.join(ROOT, "include", "kmernator_amd.h")]
    if not os.path.exists(DEMO) or any(os.path.getmtime(d) > os.path.getmtime(DEMO) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", DEMO, SRC,
                               "-L" + LIBDIR, "-lkmernator_amd", "-Wl,-rpath,$ORIGIN/../../kmernator_amd/csrc"])
    return DEMO


def _gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_cpp_host_compiles_links_and_fails_loudly_without_a_device(tmp_path):
    demo = build_demo()
    if _gpu():
        pytest.skip("a GPU is visible: the no-device path cannot be shown")
    p = subprocess.run([demo, "filter", os.path.join(GOLDEN, "1000.fastq"), str(tmp_path / "x")], capture_output=True, text=True)
    assert p.returncode == 3 and "no HIP device" in p.stderr


@pytest.mark.gpu
def test_cpp_meraculous_counter_goldens(tmp_path):
    demo = build_demo()
    out = str(tmp_path / "phix")
    p = subprocess.run([demo, "mercount", os.path.join(GOLDEN, "1000.fastq"), out], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "quality-base 64" in p.stdout and "raw 56000" in p.stdout
    for ext, gold in ((".mercount", "phix.mercount.m21"), (".mergraph", "phix.mergraph.m21.D2")):
        got = sorted(open(out + ext).read().splitlines())
        exp = sorted(open(os.path.join(GOLDEN, gold)).read().splitlines())
        assert got == exp


@pytest.mark.gpu
def test_cpp_filterreads_labels(tmp_path):
    demo = build_demo()
    out = str(tmp_path / "labels")
    p = subprocess.run([demo, "filter", os.path.join(GOLDEN, "1000.fastq"), out], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert p.stdout.startswith("Counts, Weights and Directions")
    gold = read_fastq(os.path.join(GOLDEN, "1000-Filtered.fastq"))
    lines = open(out).read().splitlines()
    assert len(lines) == 1000
    checked = 0
    for i, line in enumerate(lines):
        if b"AFTrim" in gold.names[i]:
            continue
        assert line.split(" ", 1)[1].encode() == gold.names[i].split(b" ", 1)[1], (i, line, gold.names[i])
        checked += 1
    assert checked == 949


@pytest.mark.gpu
def test_cpp_filterreads_with_artifact_filter(tmp_path):
    """the same flow with FilterKnownOddities in front (C++ mirror class): all 1000 labels incl. the 51 AFTrim ones"""
    demo = build_demo()
    out = str(tmp_path / "labels")
    p = subprocess.run([demo, "filter", os.path.join(GOLDEN, "1000.fastq"), out, os.path.join(GOLDEN, "artifact_sequences.fa")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    gold = read_fastq(os.path.join(GOLDEN, "1000-Filtered.fastq"))
    lines = open(out).read().splitlines()
    assert len(lines) == 1000
    for i, line in enumerate(lines):
        assert line.split(" ", 1)[1].encode() == gold.names[i].split(b" ", 1)[1].replace(b"\t", b" "), (i, line, gold.names[i])
    assert sum("AFTrim" in line for line in lines) == 51


@pytest.mark.gpu
def test_cpp_meraculous_counter_through_the_library_exchange(tmp_path):
    """MeraculousCounter as the one-process-per-GPU job a C++ host runs (host_demo mercount-ranks): communicator, counts, all-to-all
    and inserts all happen inside the library over RCCL.  This box has one GPU, so the job has one rank (RCCL refuses two ranks
    on a device; the N-rank logic of the same driver runs over a host transport in tests/test_gpu_exchange_library.py): its dumps
    must be the reference's goldens."""
    demo = build_demo()
    out = str(tmp_path / "phix")
    p = subprocess.run([demo, "mercount-ranks", os.path.join(GOLDEN, "1000.fastq"), out, "0", "1", str(tmp_path / "id")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "rank 0 of 1: reads 1000 raw 56000" in p.stdout and "bytes-to-peers 0" in p.stdout
    for ext, gold in ((".mercount.0", "phix.mercount.m21"), (".mergraph.0", "phix.mergraph.m21.D2")):
        got = sorted(open(out + ext).read().splitlines())
        exp = sorted(open(os.path.join(GOLDEN, gold)).read().splitlines())
        assert got == exp
