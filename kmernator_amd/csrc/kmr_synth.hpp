/* The synthetic workload of SURVEY.md section 8(d) on the device, and the order-independent digest of a finished map.
 *
 * Generator (integer only, so that a CPU restatement -- oracle/kmr_oracle.cpp, orc_synth_reads -- gives the same bytes):
 *
 *   next(s):        xorshift64*  s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1D
 *   state(seed, stream, idx): z = (seed ^ stream * 0xD1B54A32D192ED03) + idx * 0x9E3779B97F4A7C15, through the splitmix64
 *                   finalizer; 0 is replaced by 0x9E3779B97F4A7C15 (xorshift has no zero state)
 *   genome base j:  s = state(seed, 1, j >> 5); (next(s) >> 2 * (j & 31)) & 3               -- uniform i.i.d. ACGT, no buffer
 *   read r (GLOBAL index: a rank's slice of a job is a range of r):
 *       s = state(seed, 2, r); start = mulhi64(next(s), G - L + 1); strand = next(s) >> 63
 *       base i: x = next(s); code = strand ? 3 - genome[start + L - 1 - i] : genome[start + i];
 *               err = (x >> 32) < 42949673            (floor(0.01 * 2^32): 1 % substitutions)
 *               if err: code = (code + 1 + (((x & 0xffff) * 3) >> 16)) & 3     (a uniformly chosen different base)
 *               quality flat: 'I' (Q40); noisy: u = (((x >> 16) & 0xffff) * 100) >> 16;
 *                       Q = u < 80 ? 40 : u < 90 ? 30 : u < 95 ? 20 : u < 99 ? 10 : 2; an error is Q10; char = 33 + Q
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef KMR_HD
#define KMR_HD __host__ __device__ __forceinline__
#endif

namespace synth {

KMR_HD uint64_t next(uint64_t &s) {
	s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
	return s * 0x2545F4914F6CDD1Dull;
}
KMR_HD uint64_t state(uint64_t seed, uint64_t stream, uint64_t idx) {
	uint64_t z = (seed ^ (stream * 0xD1B54A32D192ED03ull)) + idx * 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	z ^= z >> 31;
	return z ? z : 0x9E3779B97F4A7C15ull;
}
__device__ __forceinline__ uint64_t genome_word(uint64_t seed, uint64_t block) { uint64_t s = state(seed, 1, block); return next(s); }

/* One thread per read.  The 32-base genome words a read spans are fetched as the read walks them; bases and qualities leave as
 * bytes (a generator, not a hot path: 10 M reads take a few milliseconds). */
__global__ __launch_bounds__(256)
void synth_reads_kernel(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t L, uint64_t G, uint32_t noisy,
                        uint8_t *bases, uint8_t *quals, uint64_t *offsets) {
	const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	if (t > n_reads) return;
	if (offsets) offsets[t] = t * L;
	if (t == n_reads) return;
	uint64_t s = state(seed, 2, first_read + t);
	const uint64_t start = __umul64hi(next(s), G - L + 1);
	const bool strand = (next(s) >> 63) != 0;
	uint8_t *bp = bases + t * L, *qp = quals ? quals + t * L : nullptr;
	uint64_t blk = ~0ull, word = 0;
	for (uint32_t i = 0; i < L; i++) {
		const uint64_t x = next(s);
		const uint64_t j = strand ? start + L - 1 - i : start + i;
		if ((j >> 5) != blk) { blk = j >> 5; word = genome_word(seed, blk); }
		uint32_t code = (uint32_t)(word >> (2 * (j & 31))) & 3u;
		if (strand) code = 3u - code;
		const bool err = (uint32_t)(x >> 32) < 42949673u;
		if (err) code = (code + 1u + (uint32_t)(((x & 0xffffu) * 3u) >> 16)) & 3u;
		bp[i] = (uint8_t)"ACGT"[code];
		if (qp) {
			uint32_t q = 40;
			if (noisy) {
				const uint32_t u = (uint32_t)((((x >> 16) & 0xffffu) * 100u) >> 16);
				q = u < 80 ? 40 : u < 90 ? 30 : u < 95 ? 20 : u < 99 ? 10 : 2;
				if (err) q = 10;
			}
			qp[i] = (uint8_t)(33 + q);
		}
	}
}

/* ---- digest of a map -------------------------------------------------------------------------------------------------
 * Per entry e = mix(...mix(mix(v0 ^ w[0]) ^ w[1])...) over the key's big-endian 8-byte words (zero padded, as the maps keep them)
 * started from the entry's integer value fields; the digest is the sum and the xor of e over all entries, so it depends neither on
 * the order of the entries nor on how they are split over buckets, parts or ranks (partial digests combine by + and ^).
 *   weak map:      v0 = count | directionBias << 16; extension values fold their 12 tallies in pairs behind the key words
 *   singleton map: v0 = 0x100000000 | _weight | packet << 40 (the 1-byte weight; extension singletons add their 4-byte packet)
 * weightedCount is a float sum whose last bits depend on the order of addition (src/KmerTrackingData.h:427-448): it is not hashed
 * but summed in double beside the hash. */
KMR_HD uint64_t mix(uint64_t x) {
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

struct Digest { unsigned long long entries, count_sum, dir_sum, hash_sum, hash_xor; double weighted_sum; };

__global__ __launch_bounds__(256)
void map_digest_kernel(const uint64_t *keys, uint32_t W, const uint32_t *vals, uint32_t vw, const uint8_t *sweight, const uint32_t *spkt,
                       uint64_t n, Digest *out) {
	unsigned long long cs = 0, ds = 0, hs = 0, hx = 0; double ws = 0.0;
	for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t x;
		if (vals) {
			const uint32_t c = vals[e * vw] & 0xffffu, d = vals[e * vw + 2] & 0xffffu;
			x = c | ((uint64_t)d << 16);
			cs += c; ds += d; ws += (double)__uint_as_float(vals[e * vw + 1]);
		} else {
			const uint32_t b = sweight[e];
			x = 0x100000000ull | b | (spkt ? (uint64_t)spkt[e] << 40 : 0ull);
			cs += b ? 1 : 0; ws += b ? (double)(int)(b - 1) / 254.0 : 0.0;
		}
		for (uint32_t j = 0; j < W; j++) x = mix(x ^ keys[e * W + j]);
		if (vals && vw > 3) for (uint32_t j = 3; j + 1 < vw; j += 2) x = mix(x ^ (vals[e * vw + j] | ((uint64_t)vals[e * vw + j + 1] << 32)));
		hs += x; hx ^= x;
	}
	/* wavefront, then block, then one set of atomics per block */
	for (int off = 32; off > 0; off >>= 1) {
		cs += __shfl_down(cs, off); ds += __shfl_down(ds, off); hs += __shfl_down(hs, off); hx ^= __shfl_down(hx, off); ws += __shfl_down(ws, off);
	}
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&out->count_sum, cs); atomicAdd(&out->dir_sum, ds); atomicAdd(&out->hash_sum, hs); atomicXor(&out->hash_xor, hx);
		atomicAdd(&out->weighted_sum, ws);
	}
}

}  // namespace synth
