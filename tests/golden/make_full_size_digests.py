#!/usr/bin/env python3
"""Writes tests/golden/full_size_digests.json: what the ORACLE (oracle/kmr_oracle.cpp, the CPU restatement of the reference's
serial build, src/KmerSpectrum.h:1914-1931) makes of BASELINE.json's full-size configurations, so that the GPU tests can hold
the HIP build of C2 and C4 to the oracle without a second build on the device and without moving the maps.

The reads come from SURVEY.md 8(d)'s generator (orc_synth_reads == kmr_synth_reads_dev byte for byte), in chunks.  A whole
spectrum does not fit this container's memory beside its reads, so each configuration is built in `parts` passes with the
reference's own partition filter (kmr_config.num_parts / part_idx = getDMPThread(kmer, numParts) == partIdx,
src/KmerSpectrum.h:1680): every pass sees all reads in input order on ONE thread (the reference's serial order, which decides
the first sighting of a k-mer and with it the quantised first weight and the float accumulation order) and keeps its share of
the k-mers.  Statistics add up over the parts; the map digest (include/kmernator_amd.h, kmr_map_digest) is defined so that
part digests add (hash_sum, count_sum, dir_sum, weighted_sum) or xor (hash_xor) to the whole map's.

Run time here (8 cores): C2 about 6 minutes per quality mode, C4 40 minutes, config 3 (100 M reads) about two hours.

    python tests/golden/make_full_size_digests.py [name ...]
"""
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

CONFIGS = {
    # name: k, seed, reads, genome (5 bases of genome per read = 30x), noisy qualities, parts [, min_depth, cfg = kmr_config fields]
    "c2_flat": dict(k=31, seed=1, reads=10_000_000, genome=50_000_000, noisy=False, parts=8),
    "c2_noisy": dict(k=31, seed=1, reads=10_000_000, genome=50_000_000, noisy=True, parts=8),
    "c4_flat": dict(k=51, seed=3, reads=50_000_000, genome=250_000_000, noisy=False, parts=16),
    # what `bench.py --gpus N` builds (weak scaling: C2's batch per GPU of one genome at 30x; N = 8 is BASELINE config 3 exactly: seed 2,
    # 100 M reads, 500 Mbp): the ranks' digests must add up to these -- the bench line of a multi-GPU run checks itself against them
    "scale_n2": dict(k=31, seed=1, reads=20_000_000, genome=100_000_000, noisy=False, parts=8),
    "scale_n4": dict(k=31, seed=1, reads=40_000_000, genome=200_000_000, noisy=False, parts=8),
    "c3_flat": dict(k=31, seed=2, reads=100_000_000, genome=500_000_000, noisy=False, parts=24),
    # small ones: the same code path at sizes any test can rebuild (tests/test_full_size_digests.py does, on the CPU)
    "small_k31_noisy": dict(k=31, seed=11, reads=200_000, genome=1_000_000, noisy=True, parts=2),
    "small_k51_flat": dict(k=51, seed=12, reads=200_000, genome=1_000_000, noisy=False, parts=3),
    # the at-scale cases of tests/test_zz_gpu_at_scale.py: multi-word keys, singleton maps, extension values
    "k51_noisy_3m": dict(k=51, seed=7, reads=3_000_000, genome=15_000_000, noisy=True, parts=8),
    "k64_flat_3m": dict(k=64, seed=7, reads=3_000_000, genome=15_000_000, noisy=False, parts=8),
    "k96_flat_3m": dict(k=96, seed=7, reads=3_000_000, genome=15_000_000, noisy=False, parts=8),
    "k96_noisy_3m": dict(k=96, seed=7, reads=3_000_000, genome=15_000_000, noisy=True, parts=8),
    "k127_flat_3m": dict(k=127, seed=7, reads=3_000_000, genome=15_000_000, noisy=False, parts=8),
    "k127_noisy_3m": dict(k=127, seed=7, reads=3_000_000, genome=15_000_000, noisy=True, parts=8),
    "sing_k31_d1": dict(k=31, seed=5, reads=3_000_000, genome=15_000_000, noisy=False, parts=8, min_depth=1),
    "sing_k31_d3": dict(k=31, seed=5, reads=3_000_000, genome=15_000_000, noisy=False, parts=8, min_depth=3),
    "sing_k31_d1_one_map": dict(k=31, seed=5, reads=3_000_000, genome=15_000_000, noisy=False, parts=8, min_depth=1, cfg=dict(separate_singletons=0)),
    "sing_k51_d1_noisy": dict(k=51, seed=5, reads=3_000_000, genome=15_000_000, noisy=True, parts=8, min_depth=1),
    "ext_k21_5m": dict(k=21, seed=3, reads=5_000_000, genome=25_000_000, noisy=False, parts=8,
                       cfg=dict(value_kind=1, min_weight=0.0, min_quality_score=2)),
    # the jobs the list exchange splits over 2 / 4 ranks on one GPU
    "xchg_k31_4m": dict(k=31, seed=4, reads=4_000_000, genome=20_000_000, noisy=False, parts=8),
    "xchg_k31_8m": dict(k=31, seed=4, reads=8_000_000, genome=40_000_000, noisy=False, parts=8),
    "xchg_k51_4m": dict(k=51, seed=4, reads=4_000_000, genome=20_000_000, noisy=False, parts=8),
    "ext_k21_noisy_2m": dict(k=21, seed=4, reads=2_000_000, genome=10_000_000, noisy=True, parts=8, min_depth=1,
                             cfg=dict(value_kind=1, min_weight=0.0, min_quality_score=2)),
}
READ_LEN = 150
CHUNK = 500_000
MIN_DEPTH = 2          # unless the configuration says otherwise


CACHE = os.environ.get("DIGEST_CACHE", "/tmp/kmr_digest_parts")      # finished parts are kept: a run that lost a worker (memory) is simply started again


def one_part(args):
    name, part = args
    from helpers import KMR_MAP_SINGLETON, KMR_MAP_WEAK, OracleSpectrum, default_config, synth_reads_8d
    c = CONFIGS[name]
    cached = os.path.join(CACHE, "%s_%dof%d.json" % (name, part, c["parts"]))
    if os.path.exists(cached):
        return tuple(json.load(open(cached)))
    per = READ_LEN - c["k"] + 1
    cfg = default_config(c["k"], estimated_raw_kmers=c["reads"] * per, num_parts=c["parts"], part_idx=part, **c.get("cfg", {}))
    o = OracleSpectrum(cfg)
    t0 = time.time()
    for lo in range(0, c["reads"], CHUNK):
        m = min(CHUNK, c["reads"] - lo)
        rb = synth_reads_8d(c["seed"], lo, m, READ_LEN, c["genome"], c["noisy"], threads=1)
        o.add_reads(rb, lo, 1)
    o.finalize(c.get("min_depth", MIN_DEPTH))
    st, dg, ds = o.stats(), o.digest(KMR_MAP_WEAK), o.digest(KMR_MAP_SINGLETON)
    o.close()
    sys.stderr.write("%s part %d/%d: %.0f s, %d weak entries\n" % (name, part, c["parts"], time.time() - t0, dg["entries"]))
    os.makedirs(CACHE, exist_ok=True)
    json.dump([st, dg, ds], open(cached + ".tmp", "w"))
    os.replace(cached + ".tmp", cached)
    return st, dg, ds


def combine(results):
    from helpers import add_digests
    stats, weak, sing = {}, None, None
    for st, dg, ds in results:
        for key, v in st.items():
            stats[key] = stats.get(key, 0) + v
        weak, sing = add_digests(weak, dg), add_digests(sing, ds)
    return stats, weak, sing


def main():
    names = sys.argv[1:] or list(CONFIGS)
    path = os.path.join(HERE, "full_size_digests.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    procs = int(os.environ.get("DIGEST_PROCS", "8"))
    with mp.get_context("spawn").Pool(procs) as pool:
        for name in names:
            c = CONFIGS[name]
            t0 = time.time()
            stats, dig, sing = combine(pool.map(one_part, [(name, p) for p in range(c["parts"])], chunksize=1))
            stats["reads"] //= c["parts"]          # every pass saw every read
            out[name] = {"config": dict(c, read_len=READ_LEN, min_depth=c.get("min_depth", MIN_DEPTH)), "stats": stats, "weak_digest": dig, "singleton_digest": sing,
                         "oracle": "serial build, %d parts" % c["parts"], "seconds": round(time.time() - t0)}
            json.dump(out, open(path, "w"), indent=1, sort_keys=True)
            print(name, json.dumps(out[name]), flush=True)


if __name__ == "__main__":
    main()
