/*
 * ref_lookup3_wrap.cpp -- exports the REFERENCE's own lookup3 (compiled from
 * where it lies: $(REF)/src/lookup3.h) so tests can check the oracle's hash
 * against it on arbitrary inputs.  TEST INFRASTRUCTURE; output goes to
 * oracle/_ref/ only and nothing of the reference is copied into this repo.
 * lookup3.h includes its system headers inside `class Lookup3`
 * (src/lookup3.h:4,44-49), so they are included here first.
 */
#include <stdio.h>
#include <time.h>
#include <stdint.h>
#include <stddef.h>
#include <sys/param.h>
#include <endian.h>
#include "lookup3.h"

extern "C" void ref_hashlittle2(const void *key, uint64_t len, uint32_t *pc, uint32_t *pb) {
	Lookup3::hashlittle2(key, (size_t)len, pc, pb);
}
/* KmerHasher::getHash convention (src/Kmer.h:207-230) */
extern "C" uint64_t ref_get_hash(const void *key, uint64_t len) {
	uint64_t hash = 0xDEADBEEF;
	uint32_t *pc = (uint32_t *)&hash, *pb = pc + 1;
	Lookup3::hashlittle2(key, (size_t)len, pc, pb);
	return hash;
}
