/* f4: the artifact filter of FilterReads on the device (FilterKnownOddities, src/FilterKnownOddities.h).
 *
 * The filter is a set of canonical match_length-mers (<= 28 bases, one 64-bit word, first base most significant in the
 * low 2*length bits) with the index of the artifact sequence each one came from, kept as an open-addressed table in HBM
 * behind a 2 MB presence filter that stays in L2.  Kernels:
 *   artifact_insert / artifact_neighbours / artifact_compact   prepareMaps (:242-287): the substitution neighbours of
 *       every key, first writer in the reference's map order wins (atomicMin on the writer's rank)
 *   artifact_screen     applyFilterToRead (:389-541), one read per thread: best / second-best quality run, every 4th
 *       window of the read looked up, the side of the read to keep
 *   artifact_action     recordAffectedRead (:551-640): discard / trim decision, pair rules
 *   artifact_gather     the read set after applyFilter: trimmed reads in place, remnant reads appended
 */
#ifndef KMR_ARTIFACT_HPP_
#define KMR_ARTIFACT_HPP_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmr {

static const uint64_t ART_EMPTY = ~0ull;

/* bits: a 2^24-bit presence filter in front of the lookup table (2 MB: it stays in L2 while the table of the reference's
 * default filter, 1.2 M keys, does not): nearly every window of a clean read is turned away by one bit test */
static const uint32_t ART_FILTER_LOG2 = 24;
struct ArtifactTable { uint64_t *keys; uint32_t *vals; uint32_t *rank; uint32_t log2cap; uint32_t *bits; };
__host__ __device__ __forceinline__ uint32_t art_filter_bit(uint64_t key) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - ART_FILTER_LOG2)); }
struct ArtifactParams {
	uint32_t length, nSeq, numErrors, srBegin, srEnd, phix, refBegin;
	int32_t minQualChar;              /* (char)(FASTQ_START_CHAR + min quality), compared as signed chars (:410-413) */
	float minReadLength;
};

__host__ __device__ __forceinline__ uint64_t art_slot(uint64_t key, uint32_t log2cap) { return (key * 0x9E3779B97F4A7C15ull) >> (64 - log2cap); }

__device__ __forceinline__ uint64_t art_revcomp(uint64_t v, uint32_t length) {
	uint64_t r = __brevll(~v);
	r = ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
	return r >> (64 - 2 * length);
}
__device__ __forceinline__ uint64_t art_least(uint64_t v, uint32_t length) { const uint64_t r = art_revcomp(v, length); return r < v ? r : v; }

/* find-or-claim; returns the slot */
__device__ __forceinline__ uint64_t art_claim(const ArtifactTable &t, uint64_t key) {
	const uint64_t mask = (1ull << t.log2cap) - 1;
	uint64_t s = art_slot(key, t.log2cap);
	for (;;) {
		unsigned long long cur = __hip_atomic_load((unsigned long long *)&t.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (cur == ART_EMPTY) cur = atomicCAS((unsigned long long *)&t.keys[s], (unsigned long long)ART_EMPTY, (unsigned long long)key);
		if (cur == ART_EMPTY || cur == key) return s;
		s = (s + 1) & mask;
	}
}
__device__ __forceinline__ uint32_t art_find(const ArtifactTable &t, uint64_t key) {          /* sequence index, 0 = absent */
	if (t.bits) { const uint32_t b = art_filter_bit(key); if (!((t.bits[b >> 5] >> (b & 31)) & 1u)) return 0; }
	const uint64_t mask = (1ull << t.log2cap) - 1;
	uint64_t s = art_slot(key, t.log2cap);
	for (;;) {
		const uint64_t cur = t.keys[s];
		if (cur == key) return t.vals[s];
		if (cur == ART_EMPTY) return 0;
		s = (s + 1) & mask;
	}
}

__global__ void artifact_fill(uint64_t *keys, uint32_t *rank, uint64_t cap) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) { keys[i] = ART_EMPTY; if (rank) rank[i] = 0xffffffffu; }
}
/* entries that are already in the filter: rank 0 = settled, no later writer can take the key */
__global__ void artifact_insert(ArtifactTable t, const uint64_t *keys, const uint32_t *vals, uint64_t n) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t s = art_claim(t, keys[i]);
		t.vals[s] = vals[i];
		if (t.rank) t.rank[s] = 0;
		if (t.bits) { const uint32_t b = art_filter_bit(keys[i]); atomicOr(&t.bits[b >> 5], 1u << (b & 31)); }
	}
}
/* KmerArrayPair::permuteBases(key, value, true) (src/Kmer.h:1434-1459) for entry e = the e-th of the map's iteration:
 * one thread per (entry, base): three substitutions, canonical form, the earliest entry keeps the key (getOrSetElement) */
__global__ void artifact_neighbours(ArtifactTable t, const uint64_t *keys, uint64_t n, uint32_t length) {
	const uint64_t total = n * length;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t e = i / length;
		const uint32_t b = (uint32_t)(i - e * length), sh = 2 * (length - 1 - b);
		const uint64_t key = keys[e], cur = (key >> sh) & 3;
		for (uint64_t nb = 0; nb < 4; nb++) {
			if (nb == cur) continue;
			const uint64_t s = art_claim(t, art_least((key & ~(3ull << sh)) | (nb << sh), length));
			atomicMin(&t.rank[s], (uint32_t)e + 1);
		}
	}
}
__global__ void artifact_compact(ArtifactTable t, const uint32_t *snap_vals, uint64_t *out_keys, uint32_t *out_vals, unsigned long long *count) {
	const uint64_t cap = 1ull << t.log2cap;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t k = t.keys[i];
		if (k == ART_EMPTY) continue;
		const uint32_t r = t.rank[i];
		const unsigned long long o = atomicAdd(count, 1ull);
		out_keys[o] = k;
		out_vals[o] = r == 0 ? t.vals[i] : snap_vals[r - 1];
	}
}

/* sequential bytes of a read through aligned 8-byte loads (the arrays are padded and 256-byte aligned) */
struct ArtBytes {
	const uint8_t *p; uint64_t w;
	__device__ __forceinline__ void init(const uint8_t *q) { p = q; w = *(const uint64_t *)((uintptr_t)q & ~(uintptr_t)7); }
	__device__ __forceinline__ uint8_t next() {
		const uint32_t o = (uint32_t)((uintptr_t)p & 7);
		if (o == 0) w = *(const uint64_t *)p;
		p++;
		return (uint8_t)(w >> (8 * o));
	}
};

struct ArtHit {
	uint32_t value, minAffected, maxAffected; bool wasPhiX;
	__device__ __forceinline__ void look(const ArtifactTable &t, const ArtifactParams &P, uint64_t key, uint32_t pos) {
		const uint32_t v = art_find(t, key);
		if (v == 0) return;
		value = v; wasPhiX |= (P.phix != 0 && v == P.phix);
		if (minAffected > pos) minAffected = pos;
		if (maxAffected < pos + P.length) maxAffected = pos + P.length;
	}
};

__device__ __forceinline__ bool art_passes_length(float length, uint32_t readLength, float minimumLength) {      /* src/ReadSelector.h:219-228 */
	if (length <= 1.0f) return false;
	if (minimumLength <= 1.0f) return readLength * minimumLength <= length;
	return minimumLength <= length;
}

__global__ __launch_bounds__(256)
void artifact_screen(const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t n, ArtifactTable t, ArtifactParams P,
                     uint32_t *value_out, uint32_t *min_pass, uint32_t *max_pass, uint32_t *rem_off, uint32_t *rem_len) {
	const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t off = offsets[i];
	const uint32_t seqLen = (uint32_t)(offsets[i + 1] - off);
	/* the two longest runs of qualities >= minQual, with the reference's swap order (:405-436) */
	int64_t b0 = 0, b1 = 0, s0 = 0, s1 = 0, t0 = 0, t1 = 0;
	{
		ArtBytes q; q.init(quals + off);
		for (uint32_t j = 0; j < seqLen; j++) {
			const int32_t c = (int8_t)q.next();
			t1 = j;
			if (c < P.minQualChar) {
				if (t1 - t0 > b1 - b0) { int64_t x = b0; b0 = t0; t0 = x; x = b1; b1 = t1; t1 = x; }
				if (t1 - t0 > s1 - s0) { int64_t x = s0; s0 = t0; t0 = x; x = s1; s1 = t1; t1 = x; }
				t0 = t1 = (int64_t)j + 1;
			}
		}
		t1 = seqLen;
		if (t1 - t0 > b1 - b0) { int64_t x = b0; b0 = t0; t0 = x; x = b1; b1 = t1; t1 = x; }
		if (t1 - t0 > s1 - s0) { int64_t x = s0; s0 = t0; t0 = x; x = s1; s1 = t1; t1 = x; }
	}
	uint32_t minPass = 0, maxPass = 0;
	if (b1 > b0) { minPass = (uint32_t)b0; maxPass = (uint32_t)b1; }

	const int64_t twoBitLength = P.length / 4, bytes = ((int64_t)seqLen + 3) / 4;
	int64_t byteHops = (int64_t)((maxPass + 3) / 4) - twoBitLength - ((seqLen & 3) == 0 ? 0 : 1);
	if (byteHops < 0 || byteHops > bytes) byteHops = 0;
	ArtHit H; H.value = 0; H.wasPhiX = false; H.minAffected = maxPass; H.maxAffected = minPass;
	const int64_t firstHop = minPass / 4;
	if (firstHop <= byteHops) {
		/* the reference's byte pointer starts at the first byte of the read whatever minPass is (:446-452, 487): window w
		 * holds bases [4w, 4w + length) and is booked at position 4 * (firstHop + w) */
		const uint64_t lastBase = (uint64_t)(byteHops - firstHop) * 4 + P.length;      /* exclusive */
		const uint32_t topShift = 2 * (P.length - 1);
		const uint64_t kmask = P.length == 32 ? ~0ull : ((1ull << (2 * P.length)) - 1);
		uint64_t fwd = 0, rc = 0;
		ArtBytes bs; bs.init(bases + off);
		for (uint64_t p = 0; p < lastBase; p++) {
			uint64_t c = 0;
			if (p < seqLen) {                                    /* beyond the read the reference reads whatever follows its buffer; zeros here */
				const uint8_t ch = bs.next();
				c = (ch == 'C' || ch == 'c') ? 1 : (ch == 'G' || ch == 'g') ? 2 : (ch == 'T' || ch == 't') ? 3 : 0;      /* compressBase: markup packs as A */
			}
			fwd = ((fwd << 2) | c) & kmask;
			rc = (rc >> 2) | ((3 - c) << topShift);
			if (p + 1 >= P.length && ((p + 1 - P.length) & 3) == 0) {
				const uint32_t pos = (uint32_t)((firstHop + (int64_t)((p + 1 - P.length) >> 2)) * 4);
				const uint64_t lk = rc < fwd ? rc : fwd;
				H.look(t, P, lk, pos);
				if (P.numErrors > 0) {
					/* Kmer.h:1389-1427: per base the three substitutions, then (edit distance 2) their subtrees; not re-canonicalised */
					for (uint32_t b1i = 0; b1i < P.length; b1i++) {
						const uint32_t sh1 = 2 * (P.length - 1 - b1i);
						const uint64_t cur1 = (lk >> sh1) & 3;
						for (uint64_t nb = 0; nb < 4; nb++) if (nb != cur1) H.look(t, P, (lk & ~(3ull << sh1)) | (nb << sh1), pos);
						if (P.numErrors > 1) {
							for (uint64_t nb = 0; nb < 4; nb++) if (nb != cur1) {
								const uint64_t v1 = (lk & ~(3ull << sh1)) | (nb << sh1);
								for (uint32_t b2i = b1i + 1; b2i < P.length; b2i++) {
									const uint32_t sh2 = 2 * (P.length - 1 - b2i);
									const uint64_t cur2 = (v1 >> sh2) & 3;
									for (uint64_t nb2 = 0; nb2 < 4; nb2++) if (nb2 != cur2) H.look(t, P, (v1 & ~(3ull << sh2)) | (nb2 << sh2), pos);
								}
							}
						}
					}
				}
			}
		}
	}
	uint32_t value = H.value, minAffected = H.minAffected, maxAffected = H.maxAffected;
	if (H.wasPhiX) value = P.phix;
	else if (P.srEnd != 0 && value >= P.srBegin && value < P.srEnd) {
		bool good = true;                                        /* a simple repeat in the middle of a read with good edges is let through */
		if ((int64_t)(uint32_t)(minAffected - minPass) < (int64_t)3 * P.length / 2) good = false;
		if ((int64_t)(uint32_t)(maxPass - maxAffected) < (int64_t)3 * P.length / 2) good = false;
		if (good) { value = 0; minAffected = maxPass; maxAffected = minPass; }
	}
	if (value > 0 && minAffected <= maxAffected) {
		if ((uint32_t)(minAffected - minPass) >= (uint32_t)(maxPass - maxAffected)) maxPass = minAffected;      /* keep the left side */
		else minPass = maxAffected;
	}
	uint32_t ro = 0, rl = 0;
	if (value == 0 && (uint32_t)(maxPass - minPass) != seqLen) {
		value = P.nSeq;
		if (art_passes_length((float)(s1 - s0), seqLen, P.minReadLength)) { ro = (uint32_t)s0; rl = (uint32_t)(s1 - s0); }
	}
	value_out[i] = value; min_pass[i] = minPass; max_pass[i] = maxPass; rem_off[i] = ro; rem_len[i] = rl;
}

/* recordAffectedRead (:551-640) + applyFilterToPair (:355-386); new_len[i] = the read's length afterwards, rem_flag = has a remnant */
__global__ void artifact_action(const uint64_t *offsets, uint64_t n, const int64_t *mate, ArtifactParams P, const uint32_t *value,
                                const uint32_t *min_pass, const uint32_t *max_pass, const uint32_t *rem_len, uint8_t *action, uint32_t *new_len, uint32_t *rem_flag) {
	const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int64_t m = mate ? mate[i] : -1;
	const uint32_t v1 = value[i], v2 = (m >= 0 && (uint64_t)m < n) ? value[m] : 0;
	const uint32_t L = (uint32_t)(offsets[i + 1] - offsets[i]);
	const bool wasPhiX = P.phix != 0 && (v1 == P.phix || v2 == P.phix);
	const bool wasReference = P.refBegin != 0 && ((v1 != P.nSeq && P.refBegin <= v1) || (v2 != P.nSeq && P.refBegin <= v2));
	uint8_t a = 0;
	if (v1 != 0 || v2 != 0) {
		if (wasPhiX) a = 2;
		else if (v1 != 0) {
			const int32_t passLength = (int32_t)(max_pass[i] - min_pass[i]);
			a = (wasReference || passLength <= 0 || !art_passes_length((float)passLength, L, P.minReadLength)) ? 2 : 1;
		}
	}
	action[i] = a;
	new_len[i] = a == 0 ? L : a == 1 ? max_pass[i] - min_pass[i] : 0;
	rem_flag[i] = rem_len[i] > 0 ? 1u : 0u;
}
/* lengths and sources of the remnant reads, appended behind the n reads in read order */
__global__ void artifact_remnants(uint64_t n, const uint32_t *rem_len, const uint64_t *rem_idx, uint32_t *new_len, uint64_t *src) {
	const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	if (i >= n || rem_len[i] == 0) return;
	new_len[n + rem_idx[i]] = rem_len[i];
	src[rem_idx[i]] = i;
}
/* one wavefront per output read */
__global__ __launch_bounds__(256)
void artifact_gather(const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, const uint64_t *name_off, const uint32_t *name_len,
                     uint64_t n, uint64_t n_out, const uint8_t *action, const uint32_t *min_pass, const uint32_t *rem_off, const uint64_t *src,
                     const uint64_t *out_offsets, uint8_t *out_bases, uint8_t *out_quals, uint64_t *out_name_off, uint32_t *out_name_len) {
	const int lane = threadIdx.x & 63;
	const uint64_t wavesPerGrid = (uint64_t)gridDim.x * (blockDim.x >> 6);
	for (uint64_t j = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); j < n_out; j += wavesPerGrid) {
		const uint64_t i = j < n ? j : src[j - n];
		const uint64_t from = offsets[i] + (j < n ? (action[i] == 1 ? min_pass[i] : 0) : rem_off[i]);
		const uint64_t to = out_offsets[j], L = out_offsets[j + 1] - to;
		for (uint64_t x = lane; x < L; x += 64) { out_bases[to + x] = bases[from + x]; out_quals[to + x] = quals[from + x]; }
		if (lane == 0 && out_name_off) { out_name_off[j] = name_off ? name_off[i] : 0; out_name_len[j] = name_len ? name_len[i] : 0; }
	}
}

}  // namespace kmr
#endif
