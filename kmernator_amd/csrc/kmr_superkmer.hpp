/*
 * kmr_superkmer.hpp -- build_mode 3: the streaming build over SUPER-K-MERS instead of k-mer records.
 *
 * The two partition passes of kmr_partition.hpp move every k-mer occurrence as a 16-byte record five times through HBM
 * (140 GB per 1.2e9 k-mers, 3.2 x the algorithmic bytes).  Consecutive k-mers of a read overlap in k-1 bases, so here a
 * record is a RUN of consecutive good k-mers of one read that share a minimizer -- a super-k-mer: one 16-byte header, the
 * run's n + k - 1 bases packed to 2 bits, and one weight (or n of them when the qualities differ inside the run) --
 * 32 bytes for ~8 k-mers.  All occurrences of a k-mer, on either strand, have the same minimizer, so hashing the minimizer
 * into 2^list_bits lists puts them into one list in a SINGLE scatter pass fused into the extraction; the count pass
 * expands each list inside LDS (lane per k-mer: bases -> forward / reverse-complement word -> canonical key) and counts it
 * in the same LDS hash table as count_kernel.  HBM sees the reads once, the super-k-mers once written and once read, and
 * the kept entries.
 *
 *   sk_extract_kernel   reads -> (exact fp64 weight chain, minimizer, runs) -> records appended to chunked lists
 *   sk_close_kernel     fill counts of the lists' open chunks (before the chunk CSR is built)
 *   sk_count_kernel     list -> LDS: records expanded to k-mers -> LDS table -> kept entries + per-bucket counts
 *
 * Replaces the same reference functions as extract_kernel + InsertOp: KmerReadUtils::buildWeightedKmers
 * (src/KmerReadUtils.h:176-248), KmerSpectrum::append + track() (src/KmerSpectrum.h:1578-1668,
 * src/KmerTrackingData.h:427,517,641) and purgeMinDepth (:1805-1815).  The minimizer, the list function and the record
 * format are private: no result depends on them (tests hold the maps byte-identical to the other build modes and to the
 * oracle).  Values: TrackingDataWithDirection / TrackingDataSingleton (KMR_VALUE_COUNT_DIR) and, with two quality bytes per k-mer and
 * the two outer neighbours in a record, ExtensionTrackingData / ...Singleton (KMR_VALUE_EXT: sk_extract_kernel<..., EXT>, sk_count_kernel<..., EXT>).
 *
 * Minimizer of a k-mer: the smallest hash among the canonical m-mers at WIN consecutive offsets placed symmetrically inside
 * the k-mer ([off, off + WIN) with 2 * off + WIN = k - m + 1), so a k-mer and its reverse complement see the same set of
 * canonical m-mers.  WIN is a compile-time 4, 8 or 16: the sliding minimum is kept in registers (block decomposition: suffix
 * minima of the previous block of WIN values, prefix minimum of the current one) with static register indices because the
 * position loop is unrolled by 16.
 */
#ifndef KMR_SUPERKMER_HPP_
#define KMR_SUPERKMER_HPP_

#include "kmr_partition.hpp"
#include "kmr_buckets.hpp"

namespace kmr {

#ifndef KMR_SK_MAX_N
#define KMR_SK_MAX_N 128
#endif
static const uint32_t SK_CHUNK_G = 64;        /* 16-byte granules per chunk: 1 KB, the CH * 16 of PoolView           */
static const uint32_t SK_MAX_N = KMR_SK_MAX_N;         /* k-mers per record                                                    */
static const int SK_WAVES = 3;                /* wavefronts per block of sk_extract_kernel (two blocks per CU by LDS) */
static const int SK_WINDOW = 16;              /* positions between two gathers = the unroll of the position loop      */
static const int SK_RR = 3, SKL_RR = 4;      /* general / lean extraction (general: 2 / 3 / 4 / 6 per round = 19.2 / 19.0 / 19.25 / 19.8 ms per noisy C2 batch; lean: 1 / 2 / 3 / 4 / 6 = 8.1 / 7.7 / 7.8 / 7.7 / 9.6): records a lane books per gather round */

#ifdef KMR_DEBUG_HOOKS
#define SK_DBG(flags, bit) (((flags) & (bit)) != 0)
#else
#define SK_DBG(flags, bit) false
#endif
/* KmerSpectrum::SizeTracker (src/KmerSpectrum.h:812-900).  The reference calls track() before every k-mer it appends; here the
 * same rule is applied at READ boundaries: after a read, if the raw k-mers so far reach nextToTrack, the spectrum's four counters
 * as they stand after that read become an element.  Extraction leaves per read what the rule needs (its raw and good k-mers and
 * the stream ordinal its bases end at); the count pass, which keeps the two smallest stream ordinals of every key, then says for
 * every boundary how many keys had been seen (first < boundary) and how many exactly once (first < boundary <= second). */
struct SkTrackRec { uint32_t raw, good; unsigned long long end_ordinal; };
static const uint32_t SK_TRACK_MAX = 512;       /* boundaries: 128 * 1.05^n reaches 10^12 k-mers at n = 467 */
struct SkTrackView { const unsigned long long *bounds; uint32_t n; unsigned int *d_unique, *d_single; };      /* d_*: [n + 1] difference arrays */

struct SkParams {
	uint32_t dbg;          /* measurement switches of a -DKMR_DEBUG_HOOKS build (they void the result): extract 1 = no list appends, 2 = no gather; count 1 = no table, 2 = no emit, 4 = no insert loop */
	uint32_t m;            /* minimizer length in bases, <= 16                                        */
	uint32_t off;          /* offset inside the k-mer of the first m-mer the minimizer looks at       */
	uint32_t list_bits;    /* code of the list count (sk_list_of): <= 32 = that many bits, above = the count itself */
	uint32_t fast_div;     /* 1: P[a] / P[b] may be formed as fma(fma(-q0, P[b], P[a]), R[b], q0), q0 = P[a] * R[b], R = 1 / P: the host has checked that this gives the correctly rounded quotient for every pair of table entries (kmr_create) */
	const double *Rp;      /* 256 entries: 1 / P[c] (0 where P[c] == 0) */
	uint32_t keep_all_owners;      /* world_size > 1: 1 inside an owner exchange (the list decides the owner), 0 = keep what getDistributedThreadId gives this rank */
	unsigned long long *state;     /* per list: open chunk << 32 | granules used (SK_CHUNK_G and NO_CHUNK: none) */
	struct SkTrackRec *track;      /* size tracker (kmr_config.size_tracker): one record per read of this launch, or null */
	const double *Pk;      /* 256 entries: P[c] multiplied k times in sequence, the weight of a window of k equal qualities */
};

/* Size history at k-mer granularity (SizeTracker::track before every append, src/KmerSpectrum.h:879-894,1574-1581): the element
 * taken when rawKmers reaches a threshold holds the counters after exactly that many raw k-mers, i.e. somewhere inside a read.  For
 * every (read, t) pair -- the threshold falls on the t-th raw k-mer of that read -- one thread walks the read as buildWeightedKmers
 * does (src/KmerReadUtils.h:176-248: the same products, quotients and restarts, in the same order) and reports the stream ordinal
 * behind that k-mer and how many of the read's first t raw k-mers were good. */
struct SkBoundary { uint64_t read; uint32_t t, good; unsigned long long ordinal; };
template <int W>
__global__ __launch_bounds__(64)
void sk_track_boundary_kernel(ReadsView rv, DevParams p, SkBoundary *bd, uint32_t n_bd) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_bd) return;
	const uint64_t r = bd[i].read, b0 = rv.offsets[r], L = rv.offsets[r + 1] - b0;
	const uint32_t k = p.k, target = bd[i].t;
	const uint8_t *bs = rv.bases + b0, *qs = rv.quals ? rv.quals + b0 : nullptr;
	const bool isRef = !qs || (L > 0 && qs[0] == 127);
	const bool filt = p.subsample > 1 || p.num_parts > 1 || p.world > 1;
	Roller<W> roll; roll.init(k);
	double w = 0.0;
	uint32_t raw = 0, good = 0, nN = 0;
	unsigned long long ord = rv.stream_base + b0 + L;
	for (uint64_t j = 0; j < L; j++) {
		const uint32_t code = base_code(bs[j]);
		nN += code >> 2;
		if (j >= k) nN -= base_code(bs[j - k]) >> 2;
		if (filt) roll.push(code & 3u);
		if (j + 1 < k) continue;
		const uint64_t x = j + 1 - k;                                    /* k-mer index */
		if (isRef) w = 1.0;
		else if (x % 1024 == 0 || w == 0.0) { w = 1.0; for (uint32_t jj = 0; jj < k; jj++) w *= p.P[qs[x + jj]]; }
		else w *= p.P[qs[x + k - 1]] / p.P[qs[x - 1]];
		if (nN) w = 0.0;
		bool mine = true;
		if (filt) {
			const Key<W> kf = roll.getFwd(), kr = roll.getRc();
			const Key<W> canon = key_le<W>(kf, kr) ? kf : kr;
			const uint64_t hash = key_hash<W>(canon, p.kb);
			if (p.subsample > 1 && hash % p.subsample != 0) mine = false;
			if (p.world > 1 && distributed_thread_id(hash, p.world) != p.rank) mine = false;
			if (p.num_parts > 1 && distributed_thread_id(hash, p.num_parts) != p.part_idx) mine = false;
		}
		if (!mine) continue;
		raw++;
		if ((float)w > p.min_weight) good++;
		if (raw == target) { ord = rv.stream_base + b0 + x + 1; break; }
	}
	bd[i].good = good; bd[i].ordinal = ord;
}

/* Record, in 16-byte granules:
 *   granule 0   { ordinal low 32 | ordinal bits 32..39, n << 8, uniform << 16, granules << 17 | minimizer hash | weight (f32 bits) }
 *   granules    the run's n + k - 1 bases, 64 per granule, first base in the top two bits of the first dword
 *   granules    n f32 weights, 4 per granule -- only when the run's weights are not all equal (uniform == 0)
 * ordinal = stream ordinal of the run's first k-mer; the others follow by +1. */
__host__ __device__ __forceinline__ uint32_t sk_base_granules(uint32_t n, uint32_t k) { return (n + k - 1 + 63) / 64; }
/* Records of a build with extension values (KMR_VALUE_EXT: ExtensionTrackingData, src/KmerTrackingData.h:1027-1126) end in
 *   granules    ceil(n / 8): two bytes per k-mer, the quality of its LEFT neighbour and of its RIGHT neighbour,
 *               as (quality char - fastq base) & 0xff, Read::REF_QUAL for a read without qualities and the minimum
 *               extension quality where the read ends (Extension('X', extMinQuality), src/KmerReadUtils.h:224-236)
 * and say in their header (bits 24..26, 27..29 of the second word; bit 30 marks such a record) which base lies left of the first
 * k-mer and right of the last one: 0..3, or 5 = 'X' where the read ends.  The neighbours in between are the record's own bases. */
static const uint32_t SK_EXT_X = 5u;
__host__ __device__ __forceinline__ uint32_t sk_ext_granules(uint32_t n) { return (n + 7) / 8; }
__host__ __device__ __forceinline__ uint32_t sk_rec_granules(uint32_t n, uint32_t k, bool uniform, bool ext) { return 1 + sk_base_granules(n, k) + (uniform ? 0u : (n + 3) / 4) + (ext ? sk_ext_granules(n) : 0u); }

#ifndef KMR_INSTANCE_TU
__global__ void sk_state_init_kernel(unsigned long long *state, uint64_t n) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) state[i] = ((unsigned long long)NO_CHUNK << 32) | SK_CHUNK_G;
}
#endif
/* the open chunk of every list gets its fill count (chunks closed by an append already have theirs) */
#ifndef KMR_INSTANCE_TU
__global__ void sk_close_kernel(const unsigned long long *state, uint64_t n, uint32_t *chunk_count, uint32_t cap) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const unsigned long long s = state[i];
		const uint32_t c = (uint32_t)(s >> 32), f = (uint32_t)s;
		if (c != NO_CHUNK && c < cap) chunk_count[c] = f < SK_CHUNK_G ? f : SK_CHUNK_G;
	}
}
#endif

/* lanes of ONE wavefront hand data to each other through LDS: the LDS executes a wavefront's instructions in issue order, so all that
 * is needed is that the compiler keeps the order (a workgroup-scope fence here also drains the vector-memory counter, i.e. waits for
 * every prefetch in flight) */
__device__ __forceinline__ void sk_wave_lds_order() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

/* inclusive prefix sum over the 64 lanes by data-parallel-primitive moves (row shifts inside the rows of 16, then the row totals
 * broadcast into the rows behind): six vector additions, no LDS permutes */
__device__ __forceinline__ uint32_t sk_wave_scan_u32(uint32_t x) {
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);      /* row_shr:1 */
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);      /* row_shr:2 */
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);      /* row_shr:4 */
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);      /* row_shr:8 */
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);     /* row_bcast:15 into rows 1 and 3 */
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);     /* row_bcast:31 into rows 2 and 3 */
	return x;
}

/* Reserve g granules in list `list`: lock-free append to a chain of fixed chunks.  The list's word is open chunk << 32 |
 * fill; an atomic add books [fill, fill + g).  The ONE adder that crosses the end of the chunk closes it (its fill count is
 * the value it saw), takes a new chunk from its wavefront's slab and publishes it with its own g already booked; adders that
 * arrive in between see a fill beyond the end and try again.  Records never straddle chunks.  Returns the granule index
 * in the pool (chunk * SK_CHUNK_G + offset) or ~0 when the pool is exhausted (error flagged). */
/* per wavefront, in LDS: two slabs of 64 chunks, and the wavefront's HOT LIST.  A list's word lets a chunk's worth of appends through
 * per round trip (~10^7 records per second): a homopolymer or a repeat that sends more than that to one list (2 % poly-A reads in
 * a C2 batch: 2 x 10^5 records in 10 ms, every lane of the chip retrying on one word, 700 ms) would stall the whole pass.  A lane
 * whose booking has failed SK_HOT_AFTER times makes that list the hot list of its wavefront: from then on the wavefront appends
 * its records for that list to a chain of chunks of its own through a word in LDS (64 contenders instead of 10^5) and hands the
 * chunks to the list closed -- a list is the set of chunks that name it, only shared appends need its word. */
struct SkSlab { uint32_t base[2]; uint32_t next; uint32_t hot_list; unsigned long long hot_state; };
static const uint32_t SK_NO_LIST = 0xffffffffu, SK_LIST_LOCKED = 0xfffffffeu;
static const int SK_HOT_AFTER = 3;
__device__ __forceinline__ uint32_t sk_alloc_chunk(SkSlab *slab, const PoolView &pool) {
	const uint32_t idx = atomicAdd(&slab->next, 1u);
	uint32_t c;
	if (idx < 128u) c = slab->base[idx >> 6] + (idx & 63u);
	else c = atomicAdd(pool.head, 1u);
	if (c >= pool.cap) { atomicOr(pool.err, (uint32_t)ERR_POOL_FULL); return NO_CHUNK; }
	return c;
}
/* the part of sk_append after the add, for adds issued ahead: a fit or a crossing is settled here; an add that landed behind
 * somebody else's crossing is void and reported through `waits` (the caller books again with sk_append) */
__device__ __forceinline__ uint64_t sk_append_settle(unsigned long long *state, uint32_t list, uint32_t g, unsigned long long old, SkSlab *slab, const PoolView &pool, bool &waits) {
	const uint32_t c = (uint32_t)(old >> 32), f = (uint32_t)old;
	if (f + g <= SK_CHUNK_G) return (uint64_t)c * SK_CHUNK_G + f;
	if (f <= SK_CHUNK_G) {
		if (c != NO_CHUNK) pool.chunk_count[c] = f;
		const uint32_t c2 = sk_alloc_chunk(slab, pool);
		if (c2 == NO_CHUNK) { atomicExch(state + list, ((unsigned long long)NO_CHUNK << 32) | (SK_CHUNK_G + 1)); return ~0ull; }
		pool.chunk_list[c2] = list;
		atomicExch(state + list, ((unsigned long long)c2 << 32) | g);
		return (uint64_t)c2 * SK_CHUNK_G;
	}
	waits = true;
	return ~0ull;
}
/* append to the wavefront's own chain for its hot list (the caller has seen slab->hot_list == list) */
__device__ __forceinline__ uint64_t sk_append_hot(SkSlab *slab, uint32_t list, uint32_t g, const PoolView &pool) {
	for (int spin = 0; spin < 4096; spin++) {
		const unsigned long long old = atomicAdd(&slab->hot_state, (unsigned long long)g);
		const uint32_t c = (uint32_t)(old >> 32), f = (uint32_t)old;
		if (f + g <= SK_CHUNK_G) return (uint64_t)c * SK_CHUNK_G + f;
		if (f <= SK_CHUNK_G) {                     /* crossed the end: close the chunk, open the next */
			pool.chunk_count[c] = f;
			const uint32_t c2 = sk_alloc_chunk(slab, pool);
			if (c2 == NO_CHUNK) return ~0ull;
			pool.chunk_list[c2] = list; pool.chunk_count[c2] = 0;
			atomicExch(&slab->hot_state, ((unsigned long long)c2 << 32) | g);
			return (uint64_t)c2 * SK_CHUNK_G;
		}
		/* a lane of this wavefront is replacing the chunk in this very iteration: again */
	}
	atomicOr(pool.err, (uint32_t)ERR_POOL_FULL);
	return ~0ull;
}
__device__ __forceinline__ uint64_t sk_append(unsigned long long *state, uint32_t list, uint32_t g, SkSlab *slab, const PoolView &pool) {
	unsigned long long *word = state + list;
	/* (every round of a contended list lets a chunk's worth of adders through: with all the chip's lanes on ONE list -- reads that are
	 * one long homopolymer -- a lane may lose some thousand rounds before it is its turn; the bound only has to end a build whose pool
	 * is gone) */
	for (int spin = 0; spin < (1 << 22); spin++) {
		if (spin >= SK_HOT_AFTER) {
			uint32_t hl = slab->hot_list;
			if (hl == SK_NO_LIST && atomicCAS(&slab->hot_list, SK_NO_LIST, SK_LIST_LOCKED) == SK_NO_LIST) {
				const uint32_t c2 = sk_alloc_chunk(slab, pool);
				if (c2 == NO_CHUNK) { slab->hot_list = SK_NO_LIST; return ~0ull; }
				pool.chunk_list[c2] = list; pool.chunk_count[c2] = 0;
				slab->hot_state = ((unsigned long long)c2 << 32) | g;        /* this record first */
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__hip_atomic_store(&slab->hot_list, list, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
				return (uint64_t)c2 * SK_CHUNK_G;
			}
			hl = __hip_atomic_load(&slab->hot_list, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (hl == list) return sk_append_hot(slab, list, g, pool);
		}
		const unsigned long long old = atomicAdd(word, (unsigned long long)g);
		const uint32_t c = (uint32_t)(old >> 32), f = (uint32_t)old;
		if (f + g <= SK_CHUNK_G) return (uint64_t)c * SK_CHUNK_G + f;
		if (f <= SK_CHUNK_G) {                     /* this add crossed the end: replace the chunk */
			if (c != NO_CHUNK) pool.chunk_count[c] = f;
			const uint32_t c2 = sk_alloc_chunk(slab, pool);
			if (c2 == NO_CHUNK) {                  /* pool exhausted: leave the list closed for good, the build is void */
				atomicExch(word, ((unsigned long long)NO_CHUNK << 32) | (SK_CHUNK_G + 1));
				return ~0ull;
			}
			pool.chunk_list[c2] = list;
			atomicExch(word, ((unsigned long long)c2 << 32) | g);
			return (uint64_t)c2 * SK_CHUNK_G;
		}
		/* somebody else is replacing the chunk: wait until the word names another chunk */
		for (int w = 0; w < 256; w++) {
			const unsigned long long now = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((uint32_t)(now >> 32) != c || (uint32_t)now <= SK_CHUNK_G) break;
			__builtin_amdgcn_s_sleep(2);
		}
		if (spin > 64) __builtin_amdgcn_s_sleep(32);      /* a crowd: come back later rather than add to it */
		if (__hip_atomic_load(pool.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ERR_POOL_FULL) return ~0ull;
	}
	atomicOr(pool.err, (uint32_t)ERR_POOL_FULL);
	return ~0ull;
}

/* 16 bases starting at base index x of a 2-bit packed array (16 bases per dword, first base in the top bits) */
__device__ __forceinline__ uint32_t sk_bases16(const uint32_t *pk, uint32_t x) {
	const uint32_t g = x >> 4, s = 2u * (x & 15u);
	const uint64_t two = ((uint64_t)pk[g] << 32) | pk[g + 1];
	return (uint32_t)((two << s) >> 32);
}
/* 16 flags starting at index x of a bit array kept as one u16 per 16 positions (bit b = position 16 g + b) */
__device__ __forceinline__ uint32_t sk_flags16(const uint16_t *a, uint32_t x) {
	const uint32_t g = x >> 4, s = x & 15u;
	return (((uint32_t)a[g] | ((uint32_t)a[g + 1] << 16)) >> s) & 0xffffu;
}

template <int W> struct SkRoll { Roller<W> r; };

/* LDS of one wavefront: quality chars of the tile, bases packed to 2 bits, N flags, (quality below the floor | quality equal
 * to the one before) flags, and per position of the current window the minimizer hash and the f32 weight of its k-mer */
static const int SK_Q_BYTES = TILE_BUF;                       /* 9856 */
static const int SK_GROUPS = TILE_BUF / 16 + 8;               /* 16-base groups incl. padding: 624 */
static const int SK_WAVE_LDS = (SK_Q_BYTES + SK_GROUPS * 4 + SK_GROUPS * 2 + SK_GROUPS * 4 + 2 * SK_WINDOW * 64 * 4 + 64 + 15) & ~15;
static const size_t SK_EXTRACT_SMEM = (size_t)SK_WAVES * SK_WAVE_LDS;

/* eight bytes from byte offset off of an LDS array (whose base is 4-byte aligned), by aligned loads */
__device__ __forceinline__ uint2 sk_lds_bytes8(const uint8_t *base, uint32_t off) {
	const uint32_t *a = (const uint32_t *)(base + (off & ~3u));
	const uint32_t d0 = a[0], d1 = a[1], d2 = a[2], sh = off & 3u;
	return make_uint2(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh));
}
/* (x - c) & 0xff in every byte of x; c4 = (256 - c) & 0xff in every byte */
__device__ __forceinline__ uint32_t sk_bytes_add(uint32_t x, uint32_t c4) { return ((x & 0x7f7f7f7fu) + (c4 & 0x7f7f7f7fu)) ^ ((x ^ c4) & 0x80808080u); }

/* the product a weight chain starts afresh with: P[q] of the k qualities from *q on, multiplied in sequence (buildWeightedKmers,
 * src/KmerReadUtils.h:205-208) */
static const int SK_PB = 8;      /* (4 / 8 / 16 at a time: 18.45 / 18.45 / 18.7 ms per noisy C2 batch; factor by factor 19.2) */
__device__ __forceinline__ double sk_fresh_product(const double *sP, const uint8_t *q, uint32_t k) {
	double w = 1.0;
	for (uint32_t j0 = 0; j0 < k; j0 += SK_PB) {
		double pv[SK_PB];
#pragma unroll
		for (int u = 0; u < SK_PB; u++) pv[u] = sP[q[j0 + u < k ? j0 + u : k - 1]];
#pragma unroll
		for (int u = 0; u < SK_PB; u++) if (j0 + u < k) w *= pv[u];
	}
	return w;
}

template <int W, int WIN, bool FILT, bool EXT = false>
__global__ __launch_bounds__(SK_WAVES * 64, 2)
void sk_extract_kernel(ReadsView rv, DevParams p, SkParams sp, PoolView pool) {
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	__shared__ double sP[256], sR[256];      /* P and 1 / P; the k-fold products Pk are read from global memory (once per read, or where a chain starts afresh) */
	__shared__ SkSlab s_slab[SK_WAVES];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	for (int i = threadIdx.x; i < 256; i += blockDim.x) { sP[i] = p.P[i]; sR[i] = sp.Rp[i]; }
	uint8_t *wb = smem + (size_t)wave * SK_WAVE_LDS;
	uint8_t *tq = wb;                                                   /* quality chars                          */
	uint32_t *pk = (uint32_t *)(wb + SK_Q_BYTES);                       /* [SK_GROUPS] packed bases               */
	uint32_t *qe = pk + SK_GROUPS;                                      /* [SK_GROUPS] low: q < floor, high: q == previous q */
	uint16_t *nm = (uint16_t *)(qe + SK_GROUPS);                        /* [SK_GROUPS] N flags                    */
	uint32_t *mhr = (uint32_t *)(wb + SK_Q_BYTES + SK_GROUPS * 10 + 8);   /* [SK_WINDOW][64] minimizer hash         */
	float *wtr = (float *)(mhr + SK_WINDOW * 64);                       /* [SK_WINDOW][64] weight                 */
	SkSlab *slab = &s_slab[wave];
	if (lane == 0) { slab->base[0] = atomicAdd(pool.head, 64u); slab->base[1] = atomicAdd(pool.head, 64u); slab->next = 0; slab->hot_list = SK_NO_LIST; slab->hot_state = 0; }
	__syncthreads();                       /* the only block-wide barrier; waves are independent below */

	const uint32_t k = p.k, m = sp.m;
	const uint32_t mmask = m >= 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
	const uint32_t mtop = 2 * (m - 1);
	unsigned long long nRaw = 0, nGood = 0, nSub = 0;
#ifdef KMR_DEBUG_HOOKS
	unsigned long long nFlatBlk = 0, nGenBlk = 0;
#endif
	const uint64_t n_items = rv.u_start ? rv.n_units : rv.n_reads;
	const uint64_t n_tiles = (n_items + 63) / 64;
	for (uint64_t tile = (uint64_t)blockIdx.x * SK_WAVES + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * SK_WAVES) {
	const uint64_t r0 = tile * 64;
	const uint32_t nr = (uint32_t)((n_items - r0) < 64 ? (n_items - r0) : 64);
	const bool have = (uint32_t)lane < nr;
	uint64_t myStart = 0, myEnd = 0, myRead = 0;
	bool myDiscard = true, myRefQual = false;
	if (have) {
		if (rv.u_start) {
			myStart = rv.u_start[r0 + lane]; myEnd = rv.u_end[r0 + lane]; myRead = rv.u_read[r0 + lane];
			const uint64_t rs = rv.offsets[myRead];
			myRefQual = rv.quals && rv.offsets[myRead + 1] > rs && rv.quals[rs] == 127;
		} else {
			myRead = r0 + lane;
			myStart = rv.offsets[myRead];
			myEnd = rv.offsets[myRead + 1];
		}
		myDiscard = rv.discarded ? (rv.discarded[myRead] != 0) : false;
	}
	uint32_t tRaw = 0, tGood = 0;

	uint32_t done = 0;
	while (done < nr) {
		const uint64_t B0 = __shfl(myStart, (int)done, 64);
		const bool fits = have && (uint32_t)lane >= done && (myEnd - B0 <= (uint64_t)TILE_SPAN);
		unsigned long long fm = __ballot(fits) >> done;
		uint32_t n = ~fm ? (uint32_t)__builtin_ctzll(~fm) : 64u;          /* run of fitting reads starting at 'done' (ctz of 0 is undefined: it came out as 32 and halved every tile) */
		if (n > nr - done) n = nr - done;
		if (n == 0) {                                        /* read longer than a tile */
			if (lane == 0) atomicOr(p.err, (uint32_t)ERR_READ_TOO_LONG);
			done += 1;
			continue;
		}
		const uint64_t B1 = __shfl(myEnd, (int)(done + n - 1), 64);
#ifdef KMR_DEBUG_HOOKS
		if (lane == 0) nGenBlk += 1000000ull + n;
#endif
		const uintptr_t gb = (uintptr_t)rv.bases + B0, gq = (uintptr_t)rv.quals + B0;
		const uintptr_t ab = gb & ~(uintptr_t)15, aq = gq & ~(uintptr_t)15;
		const uint32_t nb16 = (uint32_t)(((uintptr_t)rv.bases + B1 - ab + 15) >> 4);
		const bool haveQuals = rv.quals != nullptr;
		const uint32_t nq16 = haveQuals ? (uint32_t)(((uintptr_t)rv.quals + B1 - aq + 15) >> 4) : 0u;
		/* stage the tile: 16 bytes per lane and round, coalesced.  Bases leave as 2 bits each plus an N flag, qualities are
		 * kept as they are (the weight chain multiplies by table entries of them) plus two flags per position */
		{
			const uint4 *gbp = (const uint4 *)(rv.bases + (ptrdiff_t)(ab - (uintptr_t)rv.bases));
			const uint4 *gqp = (const uint4 *)(rv.quals + (ptrdiff_t)(aq - (uintptr_t)rv.quals));
			constexpr int STG = (TILE_BUF / 16 + 63) / 64;
			uint32_t carryq = 0x100;           /* last quality of the previous round's last lane: no match at the very start */
#pragma unroll 2
			for (int c = 0; c < STG; c++) {
				if (64u * c >= nb16 && 64u * c >= nq16) break;
				const uint32_t idx = (uint32_t)lane + 64u * c;
				uint4 vb = make_uint4(0, 0, 0, 0), vq = make_uint4(0, 0, 0, 0);
				if (idx < nb16) vb = gbp[idx];
				if (idx < nq16) vq = gqp[idx];
				{
					const uint32_t w4[4] = {vb.x, vb.y, vb.z, vb.w};
					uint32_t packed = 0, nflags = 0;
#pragma unroll
					for (int b = 0; b < 16; b++) {
						const uint32_t code = base_code((uint8_t)(w4[b >> 2] >> (8 * (b & 3))));
						packed |= (code & 3u) << (30 - 2 * b);          /* markup packs as A */
						nflags |= (code >> 2) << b;
					}
					if (idx < (uint32_t)SK_GROUPS) { pk[idx] = packed; nm[idx] = (uint16_t)nflags; }
				}
				if (haveQuals) {
					const uint32_t w4[4] = {vq.x, vq.y, vq.z, vq.w};
					uint32_t prevq = (uint32_t)__shfl_up((int)(vq.w >> 24), 1, 64);
					if (lane == 0) prevq = carryq;
					uint32_t low = 0, eq = 0;
#pragma unroll
					for (int b = 0; b < 16; b++) {
						const uint32_t q = (w4[b >> 2] >> (8 * (b & 3))) & 0xffu;
						low |= (q < p.qzero ? 1u : 0u) << b;
						eq |= (q == prevq ? 1u : 0u) << b;
						prevq = q;
					}
					if (idx < (uint32_t)SK_GROUPS) qe[idx] = low | (eq << 16);
					if (idx < nq16) *(uint4 *)(tq + 16u * idx) = vq;
					carryq = (uint32_t)__builtin_amdgcn_readlane((int)(vq.w >> 24), 63);
				}
			}
			/* two groups of padding behind the data so that window reads of the last positions stay defined */
			if (lane < 4) { const uint32_t g = nb16 + (uint32_t)lane; if (g < (uint32_t)SK_GROUPS) { pk[g] = 0; nm[g] = 0; } const uint32_t g2 = nq16 + (uint32_t)lane; if (g2 < (uint32_t)SK_GROUPS) qe[g2] = 0; }
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

		const bool active = have && (uint32_t)lane >= done && (uint32_t)lane < done + n && !myDiscard;
		const uint32_t L = active ? (uint32_t)(myEnd - myStart) : 0;
		const uint32_t rbOff = active ? (uint32_t)(gb - ab) + (uint32_t)(myStart - B0) : 0u;
		const uint32_t rqOff = active ? (uint32_t)(gq - aq) + (uint32_t)(myStart - B0) : 0u;
		const uint8_t *rq = tq + rqOff;
		uint32_t Lmax = L;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { uint32_t x = __shfl_xor(Lmax, o, 64); Lmax = x > Lmax ? x : Lmax; }

		const bool isRef = !haveQuals || (rv.u_start ? myRefQual : (L > 0 && rq[0] == 127));
		const uint64_t ord0 = rv.stream_base + myStart;      /* + k-mer index = stream ordinal of the occurrence */
		/* weight chain */
		double w = 0.0;
		uint32_t zc = 0;
		constexpr int ZN = W == 1 ? 1 : (W == 2 ? 2 : 3);      /* history of zero flags, one bit per position: k + 1 bits */
		uint64_t zbits[3] = {0, 0, 0};
		uint32_t qrun = 0;
		/* minimizer pipeline: runs sp.off positions behind the k-mer pipeline */
		uint32_t mf = 0, mr = 0;
		/* WIN > 16: the minimum over the last WIN hashes is the smaller of the block minimum over the last 16 and the block minimum of
		 * WIN - 16 positions earlier -- a delay line of WIN - 16 registers (mprev), indexed by the position modulo its length, which is a
		 * constant after unrolling as long as that length divides the 16 positions of an iteration: WIN = 17, 18, 20, 24, 32 */
		constexpr int WB = WIN > 16 ? 16 : WIN;
		constexpr int WD = WIN > 16 ? WIN - 16 : 1;
		static_assert(WIN <= 16 || SK_WINDOW % WD == 0, "the delay line of a window above 16 has to divide SK_WINDOW");
		uint32_t hs[WB], mprev[WD];
#pragma unroll
		for (int i = 0; i < WB; i++) hs[i] = 0xffffffffu;
#pragma unroll
		for (int i = 0; i < WD; i++) mprev[i] = 0xffffffffu;
		uint32_t pref = 0xffffffffu;
		/* the run in progress */
		bool runOpen = false, runUniform = true, runInWin = false;
		uint32_t runStart = 0, runN = 0, runMh = 0, runW0 = 0;
		SkRoll<FILT ? W : 1> fr;           /* forward / reverse-complement words: only the filters need the k-mer itself */
		if (FILT) fr.r.init(k);

		/* records booked but not yet written (see the gather below): start | n << 16 | uniform << 24 | window position << 25 | 1 << 31,
		 * minimizer hash, weight bits, and what the booking add returned */
		uint32_t q_info[SK_RR], q_mh[SK_RR], q_w0[SK_RR]; unsigned long long q_booked[SK_RR];
#pragma unroll
		for (int r = 0; r < SK_RR; r++) { q_info[r] = 0; q_mh[r] = 0; q_w0[r] = 0; q_booked[r] = 0; }
		auto flush_pending = [&]() {
			uint64_t at[SK_RR]; bool waits[SK_RR];
#pragma unroll
			for (int r = 0; r < SK_RR; r++) {
				waits[r] = false; at[r] = ~0ull;
				if ((q_info[r] >> 31) && !SK_DBG(sp.dbg, 1)) {
					const uint32_t n = (q_info[r] >> 16) & 0xffu; const bool uni = (q_info[r] >> 24) & 1u;
					if ((q_info[r] >> 30) & 1u) at[r] = q_booked[r];
					else at[r] = sk_append_settle(sp.state, sk_list_of(q_mh[r], sp.list_bits), sk_rec_granules(n, k, uni, EXT), q_booked[r], slab, pool, waits[r]);
				}
			}
#pragma unroll
			for (int r = 0; r < SK_RR; r++) if (waits[r]) {
				const uint32_t n = (q_info[r] >> 16) & 0xffu; const bool uni = (q_info[r] >> 24) & 1u;
				at[r] = sk_append(sp.state, sk_list_of(q_mh[r], sp.list_bits), sk_rec_granules(n, k, uni, EXT), slab, pool);
			}
#pragma unroll
			for (int r = 0; r < SK_RR; r++) {
				if ((q_info[r] >> 31) && at[r] != ~0ull && !SK_DBG(sp.dbg, 4)) {
					const uint32_t start = q_info[r] & 0xffffu, n = (q_info[r] >> 16) & 0xffu, pos = (q_info[r] >> 25) & 15u; const bool uni = (q_info[r] >> 24) & 1u;
					const uint32_t nbg = sk_base_granules(n, k), nwg = uni ? 0u : (n + 3) / 4;
					uint4 *dst = (uint4 *)pool.base + at[r];
					const uint64_t ord = ord0 + start;
					uint32_t hdrExt = 0;
					if constexpr (EXT) {
						/* neighbours outside the run: the base before its first k-mer and the one behind its last (in the tile; for the first / last
						 * k-mer of a unit of a long read in the read itself; 'X' with the minimum extension quality where the read ends), and the
						 * qualities of every k-mer's two neighbours */
						const uint32_t c4 = ((256u - (p.fastq_start & 0xffu)) & 0xffu) * 0x01010101u;
						const uint32_t qref = (127u - p.fastq_start) & 0xffu;
						const uint32_t e = start + n + k - 1;                 /* position behind the last k-mer */
						uint32_t oL = SK_EXT_X, oR = SK_EXT_X, qL = p.ext_min_q & 0xffu, qR = p.ext_min_q & 0xffu;
						if (start > 0) { oL = (sk_bases16(pk, rbOff + start - 1) >> 30) & 3u; qL = isRef ? qref : ((uint32_t)rq[start - 1] - p.fastq_start) & 0xffu; }
						else if (rv.u_start && myStart > rv.offsets[myRead]) { uint32_t c = base_code(rv.bases[myStart - 1]); oL = c == 4 ? 0u : c; qL = isRef ? qref : ((uint32_t)rv.quals[myStart - 1] - p.fastq_start) & 0xffu; }
						if (e < L) { oR = (sk_bases16(pk, rbOff + e) >> 30) & 3u; qR = isRef ? qref : ((uint32_t)rq[e] - p.fastq_start) & 0xffu; }
						else if (rv.u_start && myEnd < rv.offsets[myRead + 1]) { uint32_t c = base_code(rv.bases[myEnd]); oR = c == 4 ? 0u : c; qR = isRef ? qref : ((uint32_t)rv.quals[myEnd] - p.fastq_start) & 0xffu; }
						hdrExt = (oL << 24) | (oR << 27) | (1u << 30);
						const uint32_t neg = sk_ext_granules(n);
						for (uint32_t g = 0; g < neg; g++) {
							uint2 lq = make_uint2(qref * 0x01010101u, qref * 0x01010101u), rqv = lq;
							if (!isRef) {
								const uint32_t offL = rqOff + start + 8 * g;      /* lq[8 g + i] is the quality at start + 8 g + i - 1 */
								if (offL == 0) { const uint2 v = sk_lds_bytes8(tq, 0u); lq = make_uint2(v.x << 8, (v.y << 8) | (v.x >> 24)); }
								else lq = sk_lds_bytes8(tq, offL - 1);
								rqv = sk_lds_bytes8(tq, rqOff + start + k + 8 * g);
								lq.x = sk_bytes_add(lq.x, c4); lq.y = sk_bytes_add(lq.y, c4); rqv.x = sk_bytes_add(rqv.x, c4); rqv.y = sk_bytes_add(rqv.y, c4);
							}
							if (g == 0) lq.x = (lq.x & ~0xffu) | qL;              /* (the same value again when the neighbour lies in the tile) */
							if (g == (n - 1) >> 3) { const uint32_t b = (n - 1) & 7u; if (b < 4) rqv.x = (rqv.x & ~(0xffu << (8 * b))) | (qR << (8 * b)); else rqv.y = (rqv.y & ~(0xffu << (8 * (b - 4)))) | (qR << (8 * (b - 4))); }
							dst[1 + nbg + nwg + g] = make_uint4(__builtin_amdgcn_perm(rqv.x, lq.x, 0x05010400u), __builtin_amdgcn_perm(rqv.x, lq.x, 0x07030602u),
							                                    __builtin_amdgcn_perm(rqv.y, lq.y, 0x05010400u), __builtin_amdgcn_perm(rqv.y, lq.y, 0x07030602u));
						}
					}
					dst[0] = make_uint4((uint32_t)ord, (uint32_t)(ord >> 32) | (n << 8) | ((uni ? 1u : 0u) << 16) | (sk_rec_granules(n, k, uni, EXT) << 17) | hdrExt, q_mh[r], q_w0[r]);
					const uint32_t xb = rbOff + start;
					for (uint32_t b = 0; b < nbg; b++)
						dst[1 + b] = make_uint4(sk_bases16(pk, xb + 64 * b), sk_bases16(pk, xb + 64 * b + 16), sk_bases16(pk, xb + 64 * b + 32), sk_bases16(pk, xb + 64 * b + 48));
					for (uint32_t b = 0; b < nwg; b++) {
						uint32_t v[4];
#pragma unroll
						for (int u = 0; u < 4; u++) { const uint32_t tt = pos + 4 * b + u; v[u] = tt < 16 ? __float_as_uint(wtr[tt * 64 + lane]) : 0u; }
						dst[1 + nbg + b] = make_uint4(v[0], v[1], v[2], v[3]);
					}
				}
				q_info[r] = 0;
			}
		};
		for (uint32_t jb = 0; jb < Lmax || __any(runOpen); jb += SK_WINDOW) {
			/* the window's 16 positions of this lane's read: bases, N flags, quality flags; and the bases sp.off positions back */
			const uint32_t pkw = sk_bases16(pk, rbOff + jb);
			const uint32_t nmw = sk_flags16(nm, rbOff + jb);
			uint32_t qlow = 0, qeq = 0xffffu;
			if (haveQuals) {
				const uint32_t xq = rqOff + jb, g = xq >> 4, s = xq & 15u;
				const uint32_t a = qe[g], b = qe[g + 1];
				qlow = (((a & 0xffffu) | (b << 16)) >> s) & 0xffffu;
				qeq = (((a >> 16) | (b & 0xffff0000u)) >> s) & 0xffffu;
			}
			/* (the very first positions of a tile have nothing sp.off bases back: zeros, they end no m-mer a k-mer looks at) */
			const int32_t xm = (int32_t)(rbOff + jb) - (int32_t)sp.off;
			const uint32_t mpkw = xm >= 0 ? sk_bases16(pk, (uint32_t)xm) : (xm > -16 ? sk_bases16(pk, 0u) >> (2 * (uint32_t)(-xm)) : 0u);
			/* the run carried into this window (always uniform) */
			const bool cinOpen = runOpen; const uint32_t cinStart = runStart, cinN = runN, cinMh = runMh, cinW0 = runW0;
			uint32_t Sm = 0, Vm = 0, Cm = 0;
			runInWin = false;
			/* minimizer of the k-mer that ends at position jb + t: m-mer ending sp.off positions back, canonical, hashed; minimum of
			 * the last WIN of them (t is a constant after unrolling, so hs[] stays in registers) */
			auto minimizer_step = [&](const int t) -> uint32_t {
				const uint32_t mc = (mpkw >> (30 - 2 * t)) & 3u;
				mf = ((mf << 2) | mc) & mmask;
				mr = (mr >> 2) | ((3u - mc) << mtop);
				const uint32_t x = sk_mmer_hash(mf < mr ? mf : mr);
				const int r = t % WB;
				pref = r == 0 ? x : (x < pref ? x : pref);
				hs[r] = x;
				uint32_t M = pref;
				if (r < WB - 1) { const uint32_t sfx = hs[r + 1 < WB ? r + 1 : 0]; M = sfx < M ? sfx : M; }
				else {
#pragma unroll
					for (int u = WB - 2; u >= 0; u--) hs[u] = hs[u] < hs[u + 1] ? hs[u] : hs[u + 1];
				}
				if constexpr (WIN > 16) { const uint32_t m16 = M; const uint32_t before = mprev[t % WD]; M = before < M ? before : M; mprev[t % WD] = m16; }
				return M;
			};
			/* Flat window: no lane sees an N or a quality below the floor (in the window or in the k positions before it), and the
			 * qualities around every k-mer of the window are all equal -- a run of k + 1 equal chars behind each k-mer that continues
			 * the weight chain (x / x == 1.0: the chain does not move), k equal chars where the product starts afresh (then it is
			 * the table entry Pk[q]).  A fresh start is only allowed at the window's first k-mer: the first k-mer of a read, a lane
			 * whose chain is at zero, or -- the same for every lane -- a k-mer index that is a multiple of 1024.  Then the weight is one
			 * number per lane for the whole window and the positions need no case analysis: the same arithmetic as below without
			 * the branches (the general path spends more scalar exec-mask instructions than vector ones).  Windows a read ends in
			 * and the window its first k-mer falls into are covered (positions without a k-mer only move the minimizer on). */
			bool flatBlk = false; uint32_t flatWbits = 0;
			if (!FILT && jb + SK_WINDOW >= k) {
				const bool live = jb < L;
				const uint32_t tk = jb + 1 >= k ? 0u : k - 1 - jb;            /* first position of the window that ends a k-mer */
				const uint32_t i0 = jb + tk + 1 - k;                           /* its k-mer index */
				const uint32_t nin = live ? (L - jb < (uint32_t)SK_WINDOW ? L - jb : (uint32_t)SK_WINDOW) : 0u;
				const uint32_t inmask = (1u << nin) - 1u;
				/* fresh starts other than at tk send the window down the general path */
				const uint32_t ph = i0 & 1023u;
				const bool noLaterRestart = ph == 0 ? (SK_WINDOW - tk <= 1024u) : (ph + (SK_WINDOW - 1 - tk) < 1024u);
				const bool fresh = ph == 0 || w == 0.0;
				const uint32_t eqm = (isRef ? 0xffffu : qeq) | (jb == 0 ? 1u : 0u);       /* position 0 has no predecessor: not looked at */
				const uint32_t qAtTk = jb == 0 ? tk : qrun + tk + 1;           /* qrun as it will stand at position tk if the window's qualities are equal */
				const bool ok = !live || nin <= tk ||
				                ((nmw & inmask) == 0 && zc == 0 && (isRef || ((qlow & inmask) == 0 && (eqm & inmask) == inmask && qAtTk >= (fresh ? k - 1 : k))));
				if (noLaterRestart && __all(ok)) {
					flatBlk = true;
					const bool kmers = live && nin > tk;
					if (kmers) { if (isRef) w = 1.0; else if (fresh) w = sp.Pk[rq[jb + tk]]; }
					const float wf = (float)w;
					flatWbits = __float_as_uint(wf);
					const uint32_t kmask = kmers ? inmask & ~((1u << tk) - 1u) : 0u;      /* positions with a k-mer */
					const bool good = wf > p.min_weight;
					Vm = good ? kmask : 0u;
					tRaw += (uint32_t)__builtin_popcount(kmask); tGood += (uint32_t)__builtin_popcount(Vm);
					if (live) {
						if (ZN > 2) zbits[2] = (zbits[2] << 16) | (zbits[1] >> 48);
						if (ZN > 1) zbits[1] = (zbits[1] << 16) | (zbits[0] >> 48);
						zbits[0] <<= 16;
						qrun = jb == 0 ? SK_WINDOW - 1 : qrun + SK_WINDOW;
					}
#pragma unroll
					for (int t = 0; t < SK_WINDOW; t++) {
						const uint32_t M = minimizer_step(t);
						const bool valid = ((Vm >> t) & 1u) != 0;
						const bool cont = runOpen && M == runMh && runN < SK_MAX_N && flatWbits == runW0;
						const bool nw = valid && !cont;
						Sm |= nw ? (1u << t) : 0u;
						runStart = nw ? jb + (uint32_t)t + 1 - k : runStart;
						runN = nw ? 1u : runN + ((valid && cont) ? 1u : 0u);
						runMh = nw ? M : runMh; runW0 = nw ? flatWbits : runW0;
						runUniform = runUniform || nw; runInWin = runInWin || nw;
						runOpen = valid || (runOpen && (uint32_t)t < tk);
						mhr[t * 64 + lane] = M;
					}
				}
			}
			/* Lead-in window: no position of it ends a k-mer yet (and every lane has all 16): only the histories move */
			if (!flatBlk && !FILT) {
				const bool live = jb < L;
				if (__all(!live || (jb + SK_WINDOW <= L && jb + SK_WINDOW < k))) {
					flatBlk = true;
					if (live) {
						const uint32_t zm = nmw | (isRef ? 0u : qlow);
						zc += (uint32_t)__builtin_popcount(zm);
						if (ZN > 2) zbits[2] = (zbits[2] << 16) | (zbits[1] >> 48);
						if (ZN > 1) zbits[1] = (zbits[1] << 16) | (zbits[0] >> 48);
						zbits[0] = (zbits[0] << 16) | (uint64_t)(__builtin_bitreverse32(zm) >> 16);      /* the newest position in bit 0 */
						const uint32_t eq = (isRef ? 0xffffu : qeq) & (jb == 0 ? 0xfffeu : 0xffffu);       /* position 0 has no predecessor */
						qrun = eq == 0xffffu ? qrun + SK_WINDOW : (uint32_t)__builtin_clz(~(eq << 16));      /* equal qualities counted back from the window's end */
					}
#pragma unroll
					for (int t = 0; t < SK_WINDOW; t++) mhr[t * 64 + lane] = minimizer_step(t);
					if (!live) runOpen = false;
				}
			}
#ifdef KMR_DEBUG_HOOKS
			if (lane == 0) { if (flatBlk) nFlatBlk++; else nGenBlk++; }
#endif
			if (!flatBlk) {
			/* General path, in two sweeps over the window: the minimizers first (unrolled: their block decomposition wants the position in
			 * the window as a constant) into the ring, then the case analysis of the weight chain and the runs as ONE loop body -- unrolled
			 * 16 times it was 70 KB of code and ran out of the instruction cache (24 ms per C2 batch of noisy reads against 7 ms for the walk
			 * of flat ones) */
#pragma unroll
			for (int t = 0; t < SK_WINDOW; t++) mhr[t * 64 + lane] = minimizer_step(t);
			if constexpr (!FILT) {
			/* Without filters the case analysis is taken apart (noisy qualities: 215 vector + 147 scalar instructions per position
			 * through the nested branches below, most of the scalar ones exec-mask bookkeeping, and a k-step product loop whenever ANY
			 * lane's chain starts afresh -- with zero-probability bases sprinkled over the reads that is nearly every position):
			 *   sweep A  flags only: zero history, equal-quality run, and where in the window this lane's chain will start afresh
			 *            (the first k-mer, a multiple of 1024, or the k-mer behind a zero run: all known from the flags);
			 *   once per window, all lanes together: the fresh product at that position (table entry or the k-step loop);
			 *   sweep B  weights and runs with selects instead of branches: the chain's multiply / divide is applied where the quality
			 *            that enters differs from the one that leaves (x / x == 1.0 otherwise: the same arithmetic as skipping it),
			 *            the division is skipped only when no lane of the wavefront needs it.
			 * Every decision still looks at the actual state (w == 0.0, not its prediction): a start the flags did not foresee -- a second
			 * one in the same window -- takes the loop in place. */
			uint32_t zcm = 0, tR = 16, qrunR = 0;
			bool wz = w == 0.0;
			if (k >= (uint32_t)SK_WINDOW) {
				/* sweep A on whole 16-bit masks (k >= 16: what leaves the k-window during these 16 positions lies wholly in the history):
				 * zero flags of the window, the flags that leave it, and from the two the positions where the window count is above
				 * zero; k-mer positions; where the chain starts afresh */
				const uint32_t nin = jb < L ? (L - jb < (uint32_t)SK_WINDOW ? L - jb : (uint32_t)SK_WINDOW) : 0u;
				const uint32_t inmask = (1u << nin) - 1u;
				const uint32_t tk = jb + 1 >= k ? 0u : k - 1 - jb;                     /* < 16 here: jb + 16 >= k was checked by the caller's paths */
				const uint32_t km = tk < 16u ? (inmask & ~((1u << tk) - 1u)) : 0u;
				const uint32_t zm = (nmw | (isRef ? 0u : qlow)) & inmask;
				uint32_t field;                                                          /* history bits k-16 .. k-1 (bit b = position jb - 1 - b) */
				{
					const uint32_t lo = k - 16u, wi = lo >> 6, sh = lo & 63u;
					uint64_t a = zbits[wi < (uint32_t)ZN ? wi : ZN - 1];
					if (wi >= (uint32_t)ZN) a = 0;
					uint64_t b = (wi + 1 < (uint32_t)ZN) ? zbits[wi + 1 < (uint32_t)ZN ? wi + 1 : ZN - 1] : 0ull;
					field = (uint32_t)((sh ? (a >> sh) | (b << (64u - sh)) : a) & 0xffffu);
				}
				const uint32_t lm = __builtin_bitreverse32(field) >> 16;                 /* bit t: the flag that leaves at position jb + t */
				uint32_t zcw = zc, zcpos = 0;
#pragma unroll
				for (int t = 0; t < SK_WINDOW; t++) { zcw += (zm >> t) & 1u; zcw -= (lm >> t) & 1u; zcpos |= (zcw != 0 ? 1u : 0u) << t; }
				zc = zcw;
				if (ZN > 2) zbits[2] = (zbits[2] << 16) | (zbits[1] >> 48);
				if (ZN > 1) zbits[1] = (zbits[1] << 16) | (zbits[0] >> 48);
				zbits[0] = (zbits[0] << 16) | (uint64_t)(__builtin_bitreverse32(zm) >> 16);
				zcm = zcpos & km;
				const uint32_t t1024 = (k - 1u - jb) & 1023u;                             /* position whose k-mer index is a multiple of 1024 */
				const uint32_t p1024 = t1024 < 16u ? (1u << t1024) : 0u;
				const uint32_t wzprev = ((zcm << 1) & 0xffffu) | ((wz && tk < 16u) ? (1u << tk) : 0u);
				const uint32_t fm = isRef ? 0u : (km & ~zcpos & (p1024 | wzprev));
				const uint32_t em = (isRef ? 0xffffu : qeq) & (jb == 0 ? 0xfffeu : 0xffffu);
				if (fm) {
					tR = (uint32_t)__builtin_ctz(fm);
					const uint32_t x = ~em & ((2u << tR) - 1u);
					qrunR = x ? tR - (31u - (uint32_t)__builtin_clz(x)) : qrun + tR + 1u;
				}
				{ const uint32_t x = ~em & 0xffffu; qrun = x ? 15u - (31u - (uint32_t)__builtin_clz(x)) : qrun + 16u; }
			} else {
#pragma nounroll
			for (uint32_t t = 0; t < (uint32_t)SK_WINDOW; t++) {
				const uint32_t j = jb + t;
				const bool in = j < L;
				const bool z = in && ((((nmw >> t) & 1u) != 0) || (!isRef && ((qlow >> t) & 1u) != 0));
				if (ZN > 2) zbits[2] = (zbits[2] << 1) | (zbits[1] >> 63);
				if (ZN > 1) zbits[1] = (zbits[1] << 1) | (zbits[0] >> 63);
				zbits[0] = (zbits[0] << 1) | (z ? 1ull : 0ull);
				zc += z ? 1u : 0u;
				zc -= (uint32_t)((zbits[ZN == 1 ? 0 : (k >> 6)] >> (k & 63)) & 1ull);
				qrun = (j > 0 && (isRef || ((qeq >> t) & 1u))) ? qrun + 1 : 0;
				const bool hasK = in && j + 1 >= k;
				const bool zero = zc > 0;
				zcm |= (hasK && zero) ? (1u << t) : 0u;
				const bool fresh = hasK && !zero && !isRef && ((((j + 1 - k) & 1023u) == 0) || wz);
				if (fresh && tR == 16) { tR = t; qrunR = qrun; }
				wz = hasK ? zero : wz;
			}
			}
			double wR = 0.0;
			if (__any(tR < 16)) {
				if (tR < 16) {
					const uint32_t jR = jb + tR, iR = jR + 1 - k;
					if (qrunR + 1 >= k) wR = sp.Pk[rq[jR]];                 /* k equal qualities: the table holds the same sequence of products */
					else {
						/* the k probabilities first (SK_PB at a time: the byte reads, then the table reads, each batch ONE LDS round trip), then the
						 * multiplications in the reference's order -- factor by factor it was two dependent LDS round trips each */
						wR = sk_fresh_product(sP, rq + iR, k);
					}
				}
			}
#pragma unroll 4      /* (rolled: 19.9 ms per noisy C2 batch; by 2: 19.6; by 4: 19.4; by 8: 19.7; whole: 19.9 -- the code of 16 bodies) */
			for (uint32_t t = 0; t < (uint32_t)SK_WINDOW; t++) {
				const uint32_t j = jb + t;
				const bool in = j < L;
				const bool hasK = in && j + 1 >= k;
				const uint32_t i = hasK ? j + 1 - k : 0u;
				const bool zero = ((zcm >> t) & 1u) != 0;
				const bool live = hasK && !zero && !isRef;
				const bool fresh = live && ((i & 1023u) == 0 || w == 0.0);
				if (__any(fresh && t != tR)) {
										if (fresh && t != tR) { w = 1.0; for (uint32_t jj = 0; jj < k; jj++) w *= sP[rq[i + jj]]; }      /* (rare, and inside the unrolled sweep: the plain loop) */      /* (the table entry is this very product) */
				}
				w = (fresh && t == tR) ? wR : w;
				const uint32_t q = rq[hasK ? j : 0u], qo = rq[(hasK && i > 0) ? i - 1 : 0u];
				const bool moves = live && !fresh && q != qo;
				if (__any(moves)) {
					double change;
					if (sp.fast_div) { const double a = sP[q], b = sP[qo], r = sR[qo]; const double q0 = a * r; change = fma(fma(-q0, b, a), r, q0); }
					else change = sP[q] / sP[qo];
					w = moves ? w * change : w;
				}
				w = hasK ? (zero ? 0.0 : (isRef ? 1.0 : w)) : w;
				const float wf = hasK ? (float)w : 0.0f;
				const bool valid = hasK && wf > p.min_weight;
				tRaw += hasK ? 1u : 0u; tGood += valid ? 1u : 0u;
				const uint32_t M = mhr[t * 64 + lane];
				const uint32_t wbits = __float_as_uint(wf);
				const bool wsame = wbits == runW0;
				const bool cont = valid && runOpen && M == runMh && runN < SK_MAX_N && (wsame || runInWin);
				const bool nw = valid && !cont;
				const bool uneven = cont && !wsame;
				Cm |= uneven ? (1u << t) : 0u; Sm |= nw ? (1u << t) : 0u; Vm |= valid ? (1u << t) : 0u;
				runUniform = nw || (runUniform && !uneven);
				runStart = nw ? i : runStart;
				runN = nw ? 1u : runN + (cont ? 1u : 0u);
				runMh = nw ? M : runMh; runW0 = nw ? wbits : runW0;
				runInWin = runInWin || nw;
				runOpen = hasK ? valid : (in && runOpen);
				wtr[t * 64 + lane] = wf;
			}
			} else {
#pragma nounroll
			for (uint32_t t = 0; t < (uint32_t)SK_WINDOW; t++) {
				const uint32_t j = jb + t;
				const bool in = j < L;
				const uint32_t code = (pkw >> (30 - 2 * t)) & 3u;
				const bool z = in && ((((nmw >> t) & 1u) != 0) || (!isRef && ((qlow >> t) & 1u) != 0));
				if (ZN > 2) zbits[2] = (zbits[2] << 1) | (zbits[1] >> 63);
				if (ZN > 1) zbits[1] = (zbits[1] << 1) | (zbits[0] >> 63);
				zbits[0] = (zbits[0] << 1) | (z ? 1ull : 0ull);
				zc += z ? 1u : 0u;
				zc -= (uint32_t)((zbits[ZN == 1 ? 0 : (k >> 6)] >> (k & 63)) & 1ull);   /* position j-k leaves the window (0 while j < k) */
				qrun = (j > 0 && (isRef || ((qeq >> t) & 1u))) ? qrun + 1 : 0;
				if (FILT) fr.r.push(code);
				const uint32_t M = mhr[t * 64 + lane];
				bool valid = false;
				float wf = 0.0f;
				if (in && j + 1 >= k) {
					const uint32_t i = j + 1 - k;
					if (zc > 0) w = 0.0;
					else if (isRef) w = 1.0;
					else if ((i & 1023u) == 0 || w == 0.0) {
						if (qrun + 1 >= k) w = sp.Pk[rq[j]];                 /* k equal qualities: the table holds the same sequence of products */
						else w = sk_fresh_product(sP, rq + i, k);
					} else if (qrun < k) {
						/* x / x == 1.0 exactly, so equal qualities leave w unchanged; a run of k+1 equal chars proves that without a load */
						const uint32_t q = rq[j], qo = rq[i - 1];
						if (qo != q) { const double change = sP[q] / sP[qo]; w *= change; }
					}
					wf = (float)w;
					bool mine = true;
					if (FILT) {
						const Key<FILT ? W : 1> kf = fr.r.getFwd(), kr = fr.r.getRc();
						const Key<FILT ? W : 1> canon = key_le<FILT ? W : 1>(kf, kr) ? kf : kr;
						const uint64_t hash = key_hash<FILT ? W : 1>(canon, p.kb);
						if (p.subsample > 1 && hash % p.subsample != 0) mine = false;
						if (p.world > 1 && !sp.keep_all_owners && distributed_thread_id(hash, p.world) != p.rank) mine = false;
						if (p.num_parts > 1 && distributed_thread_id(hash, p.num_parts) != p.part_idx) mine = false;
						if (mine && (p.sub_wnb | p.sub_snb)) {       /* subtractingReference->exists(least): skipped before rawKmers++ */
							MapView<FILT ? W : 1> sw, ss;
							sw.start = p.sub_wstart; sw.keys = p.sub_wkeys; sw.vals = p.sub_wvals; sw.sweight = nullptr; sw.nb = p.sub_wnb; sw.vw = p.sub_vw;
							ss.start = p.sub_sstart; ss.keys = p.sub_skeys; ss.vals = nullptr; ss.sweight = p.sub_sweight; ss.nb = p.sub_snb; ss.vw = 0;
							if (maps_count<FILT ? W : 1>(sw, ss, canon, hash) > 0) { mine = false; nSub++; }
						}
					}
					if (mine) {
						tRaw++;
						valid = wf > p.min_weight;
						if (valid) tGood++;
					}
					/* run logic: the k-mer continues the run in progress if it is good, has the same minimizer and -- for a run that
					 * began before this window, whose earlier weights are no longer at hand -- the same weight */
					const uint32_t wbits = __float_as_uint(wf);
					if (valid) {
						const bool wsame = wbits == runW0;
						const bool cont = runOpen && M == runMh && runN < SK_MAX_N && (wsame || runInWin);
						if (cont) { runN++; if (!wsame) { runUniform = false; Cm |= 1u << t; } }
						else { Sm |= 1u << t; runOpen = true; runStart = i; runN = 1; runMh = M; runW0 = wbits; runUniform = true; runInWin = true; }
						Vm |= 1u << t;
					} else runOpen = false;
				} else if (!in) runOpen = false;
				wtr[t * 64 + lane] = wf;
			}
			}
			}
			/* a run with unequal weights ends with its window (its weights live in this window's ring) */
			const bool openEnd = runOpen && runUniform;
			if (runOpen && !runUniform) runOpen = false;
			/* gather: every run that ended in this window becomes a record of its list */
			const uint32_t brk = (~Vm | Sm) & 0xffffu;
			bool pendC = false; uint32_t lead = 16;
			if (cinOpen) { lead = (uint32_t)__builtin_ctz(brk | 0x10000u); pendC = lead < 16; }
			uint32_t Srem = Sm;
			if (openEnd && runInWin && Sm) Srem &= ~(1u << (31 - __builtin_clz(Sm)));      /* the run still open is the last one begun */
			if (SK_DBG(sp.dbg, 2)) { pendC = false; Srem = 0; }
			/* Four records per lane and round: their list words are bumped back to back, and the bookings of a round are settled
			 * and the records written at the NEXT window's gather (or at the end of the tile): scattered device atomics run at
			 * ~2 x 10^10 per second chip-wide and take tens of microseconds under that load -- time the wavefront now spends
			 * walking the next 16 positions.  Settling goes in two steps: first everything that does not wait (a fit, or the one
			 * add that crossed a chunk's end and replaces it), only then the adds that have to wait for somebody else's new chunk
			 * -- a lane of this wavefront may be that somebody.  Records with a weight per k-mer are settled at once (their
			 * weights live in this window's ring). */
			flush_pending();
			bool firstRound = true;
			const uint32_t hotNow = slab->hot_list;      /* read once per window: a stale value only delays the switch to the wavefront's own chain */
			while (__any(pendC || Srem)) {
				bool anyNow = false;
#pragma unroll
				for (int r = 0; r < SK_RR; r++) {
					q_info[r] = 0;
					if (pendC) { pendC = false; q_info[r] = (1u << 31) | (1u << 24) | ((cinN + lead) << 16) | cinStart; q_mh[r] = cinMh; q_w0[r] = cinW0; }
					else if (Srem) {
						const uint32_t pos = (uint32_t)__builtin_ctz(Srem); Srem &= Srem - 1;
						const uint32_t end = pos + 1 + (uint32_t)__builtin_ctz((brk >> (pos + 1)) | (1u << (15 - pos)));
						const uint32_t n = end - pos;
						const bool uni = ((Cm >> (pos + 1)) & ((1u << (n - 1)) - 1u)) == 0;
						q_info[r] = (1u << 31) | (pos << 25) | ((uni ? 1u : 0u) << 24) | (n << 16) | (jb + pos + 1 - k);
						q_mh[r] = mhr[pos * 64 + lane]; q_w0[r] = flatBlk ? flatWbits : __float_as_uint(wtr[pos * 64 + lane]);
						anyNow = anyNow || !uni;
					}
					if ((q_info[r] >> 31) && !SK_DBG(sp.dbg, 1)) {
						const uint32_t n = (q_info[r] >> 16) & 0xffu; const bool uni = (q_info[r] >> 24) & 1u;
						const uint32_t need = sk_rec_granules(n, k, uni, EXT), myList = sk_list_of(q_mh[r], sp.list_bits);
						if (myList == hotNow) { q_booked[r] = sk_append_hot(slab, myList, need, pool); q_info[r] |= 1u << 30; }      /* an address, not what a booking add returned */
						else q_booked[r] = atomicAdd(sp.state + myList, (unsigned long long)need);
					}
				}
				if (!firstRound || __any(anyNow) || __any(pendC || Srem)) flush_pending();      /* more rounds to come, or weights in the ring: settle now */
				firstRound = false;
			}
			/* refill the slab cache of the wavefront (uniform) */
			if (slab->next >= 64u) {
				__builtin_amdgcn_wave_barrier();
				if (lane == 0) { const uint32_t used = slab->next; slab->base[0] = slab->base[1]; slab->base[1] = atomicAdd(pool.head, 64u); slab->next = used >= 128u ? 64u : used - 64u; }
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
				__builtin_amdgcn_wave_barrier();
			}
		}
		flush_pending();                   /* the last window's records: their bases are read from this tile */
		done += n;
		__builtin_amdgcn_wave_barrier();   /* all lanes are done reading the tile before it is overwritten */
	}
	nRaw += tRaw; nGood += tGood;
	if (sp.track && have && !myDiscard) {      /* the lane's read (or its unit of a long read: the units of a read add up) */
		SkTrackRec *tr = sp.track + myRead;
		if (rv.u_start) { atomicAdd(&tr->raw, tRaw); atomicAdd(&tr->good, tGood); atomicMax(&tr->end_ordinal, (unsigned long long)(rv.stream_base + myEnd)); }
		else { tr->raw = tRaw; tr->good = tGood; tr->end_ordinal = rv.stream_base + myEnd; }
	}
	}
	/* chunks of the slabs nobody took belong to no list */
	__builtin_amdgcn_wave_barrier();
	{
		const uint32_t used = slab->next < 128u ? slab->next : 128u;
		for (uint32_t idx = used + (uint32_t)lane; idx < 128u; idx += 64) { const uint32_t c = slab->base[idx >> 6] + (idx & 63u); if (c < pool.cap) { pool.chunk_list[c] = NO_CHUNK; pool.chunk_count[c] = 0; } }
	if (lane == 0 && slab->hot_list < SK_LIST_LOCKED) {      /* the open chunk of the wavefront's hot chain */
		const uint32_t hc = (uint32_t)(slab->hot_state >> 32), hf = (uint32_t)slab->hot_state;
		if (hc < pool.cap) pool.chunk_count[hc] = hf < SK_CHUNK_G ? hf : SK_CHUNK_G;
	}
	}
	nRaw = wave_sum(nRaw); nGood = wave_sum(nGood);
	if (lane == 0) { atomicAdd(&p.stats->raw, nRaw); atomicAdd(&p.stats->good, nGood); }
#ifdef KMR_DEBUG_HOOKS
	if (lane == 0) { atomicAdd(&p.stats->claimed, nGenBlk); atomicAdd(&p.stats->inserted, nFlatBlk); }      /* windows down the general / the fast paths */
#endif
	if (FILT) { nSub = wave_sum(nSub); if (lane == 0 && nSub) atomicAdd(&p.stats->subtracted, nSub); }
}

/* ------------------------------------------------------------------ extraction when every k-mer without an N weighs the same */
/* A launch whose reads carry no qualities (FASTA input, a reference: every base has probability 1, src/Sequence.h REF_QUAL) or
 * one and the same quality character everywhere (found by sk_qual_range_kernel) needs no weight chain: a k-mer's weight is 0 when
 * its window holds an N and the table's k-fold product of that one probability otherwise (buildWeightedKmers,
 * src/KmerReadUtils.h:176-248: equal qualities leave the chain where it is, every fresh start gives the same product).  Without the
 * quality bytes, their flags and the weights ring a wavefront's tile is 7.8 KB of LDS instead of 24, and without the fp64 chain the
 * walk fits 128 registers: 16 wavefronts per CU instead of 6 -- the general kernel is bound by instruction issue at 1.5
 * wavefronts per SIMD.  The records, the list appends and their settlement one window later are the general kernel's. */
static const int SKL_WAVES = 4, SKL_MIN_BLOCKS = 2;      /* 8 wavefronts per CU at 193 registers: 12 (168 registers, or fewer records booked per round) and 16 (128, spilling) were slower -- the pass runs at the chip's rate of scattered device atomics (1.8 x 10^10 per second, one per record) */      /* 168 registers: at 128 (four blocks) the walk spilled 59 dwords */
static const int SKL_WAVE_LDS = (SK_GROUPS * 4 + SK_GROUPS * 2 + 8 + SK_WINDOW * 64 * 4 + 15) & ~15;
static const size_t SKL_EXTRACT_SMEM = (size_t)SKL_WAVES * SKL_WAVE_LDS;

/* Are the quality characters of the reads offsets[0] .. offsets[n_reads] all the same one?  range[0] gets a lower bound of the smallest
 * and range[1] an upper bound of the largest (the bytes of the AND and of the OR of all dwords: two instructions a dword instead of
 * a minimum and a maximum per byte), equal -- and then exact -- iff one character fills the range; 255 / 0 are left alone when there is none */
#ifndef KMR_INSTANCE_TU
__global__ __launch_bounds__(256)
void sk_qual_range_kernel(const uint8_t *quals, const uint64_t *offsets, uint64_t n_reads, unsigned int *range) {
	unsigned int lo = 255, hi = 0;
	uint32_t all = 0xffffffffu, any = 0u;
	const uint64_t b0 = offsets[0], b1 = offsets[n_reads];
	const uintptr_t p0 = (uintptr_t)(quals + b0), p1 = (uintptr_t)(quals + b1);
	const uintptr_t a0 = (p0 + 15) & ~(uintptr_t)15, a1 = p1 & ~(uintptr_t)15;      /* the 16-byte aligned middle */
	const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
	if (a0 < a1) {
		const uint4 *q4 = (const uint4 *)a0;
		const uint64_t n16 = (a1 - a0) / 16;
		/* four loads in flight per thread (one at a time the pass ran at 3.4 TB/s on a C2 batch, 1.5 TB/s on a quarter of it) */
		for (uint64_t i0 = tid; i0 < n16; i0 += 4 * nthreads) {
			uint4 v[4];
#pragma unroll
			for (int u = 0; u < 4; u++) { const uint64_t i = i0 + (uint64_t)u * nthreads; v[u] = q4[i < n16 ? i : i0]; }
#pragma unroll
			for (int u = 0; u < 4; u++) { all &= v[u].x & v[u].y & v[u].z & v[u].w; any |= v[u].x | v[u].y | v[u].z | v[u].w; }
		}
		if (tid < 16) { const uintptr_t q = p0 + tid; if (q < a0) { const unsigned int c = *(const uint8_t *)q; lo = c < lo ? c : lo; hi = c > hi ? c : hi; } }
		if (tid >= 16 && tid < 32) { const uintptr_t q = a1 + (tid - 16); if (q < p1) { const unsigned int c = *(const uint8_t *)q; lo = c < lo ? c : lo; hi = c > hi ? c : hi; } }
	} else {
		for (uintptr_t q = p0 + tid; q < p1; q += nthreads) { const unsigned int c = *(const uint8_t *)q; lo = c < lo ? c : lo; hi = c > hi ? c : hi; }
	}
#pragma unroll
	for (int b = 0; b < 4; b++) { const unsigned int c0 = (all >> (8 * b)) & 0xffu, c1 = (any >> (8 * b)) & 0xffu; lo = c0 < lo ? c0 : lo; hi = c1 > hi ? c1 : hi; }
	for (int o = 32; o > 0; o >>= 1) { const unsigned int a = (unsigned int)__shfl_xor((int)lo, o, 64), b = (unsigned int)__shfl_xor((int)hi, o, 64); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
	if ((threadIdx.x & 63) == 0) { atomicMin(&range[0], lo); atomicMax(&range[1], hi); }
}
#endif

/* PACKED: the bases arrive as the reference's Read keeps them (TwoBitSequence, src/TwoBitSequence.cpp:242-269: four bases per byte, the
 * first one in bits 7-6, every read on bytes of its own) plus markups -- kmr_add_reads_twobit* with one quality character for the
 * batch.  Four packed bytes read as a big-endian dword ARE a group of the tile's LDS form, so the tile is staged with a byte swap
 * per dword instead of sixteen base_code()s per group, a quarter of the bytes come in, and no unpacked copy of the batch is
 * written and read back; a read's place in the tile is a base offset as before (4 * its byte offset, + the bases a unit of a
 * long read starts behind the read's first one).  Markups set the N flag of their position (a markup that names a base rewrites it). */
struct SkPacked {
	const uint8_t *bytes;          /* packed bases */
	const uint64_t *off;           /* per read: byte offset of its first base */
	const uint64_t *mk_off;        /* per read: its markups are entries [mk_off[i], mk_off[i + 1]); may be null */
	const uint32_t *mk_pos;        /* position inside the read */
	const uint8_t *mk_char;
};
static const uint32_t SK_PACKED_SPAN = TILE_SPAN - 64;      /* bases of a tile: the staging starts on a 16-byte boundary, up to 60 bases in front of the first read */

template <int W, int WIN, bool PACKED = false>
__global__ __launch_bounds__(SKL_WAVES * 64, SKL_MIN_BLOCKS)
void sk_extract_lean_kernel(ReadsView rv, DevParams p, SkParams sp, PoolView pool, float wK, SkPacked pkd) {
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	__shared__ SkSlab s_slab[SKL_WAVES];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	uint8_t *wb = smem + (size_t)wave * SKL_WAVE_LDS;
	uint32_t *pk = (uint32_t *)wb;                                      /* [SK_GROUPS] packed bases */
	uint16_t *nm = (uint16_t *)(pk + SK_GROUPS);                        /* [SK_GROUPS] N flags      */
	uint32_t *mhr = (uint32_t *)(wb + SK_GROUPS * 6 + 8);               /* [SK_WINDOW][64] minimizer hash */
	SkSlab *slab = &s_slab[wave];
	if (lane == 0) { slab->base[0] = atomicAdd(pool.head, 64u); slab->base[1] = atomicAdd(pool.head, 64u); slab->next = 0; slab->hot_list = SK_NO_LIST; slab->hot_state = 0; }
	__syncthreads();                       /* the only block-wide barrier; waves are independent below */

	const uint32_t k = p.k, m = sp.m;
	const uint32_t mmask = m >= 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
	const uint32_t mtop = 2 * (m - 1);
	const uint32_t wbits = __float_as_uint(wK);
	const bool good = wK > p.min_weight;           /* the same for every k-mer without an N */
	unsigned long long nRaw = 0, nGood = 0;
	const uint64_t n_items = rv.u_start ? rv.n_units : rv.n_reads;
	const uint64_t n_tiles = (n_items + 63) / 64;
	for (uint64_t tile = (uint64_t)blockIdx.x * SKL_WAVES + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * SKL_WAVES) {
	const uint64_t r0 = tile * 64;
	const uint32_t nr = (uint32_t)((n_items - r0) < 64 ? (n_items - r0) : 64);
	const bool have = (uint32_t)lane < nr;
	uint64_t myStart = 0, myEnd = 0, myRead = 0;
	bool myDiscard = true;
	if (have) {
		if (rv.u_start) { myStart = rv.u_start[r0 + lane]; myEnd = rv.u_end[r0 + lane]; myRead = rv.u_read[r0 + lane]; }
		else { myRead = r0 + lane; myStart = rv.offsets[myRead]; myEnd = rv.offsets[myRead + 1]; }
		myDiscard = rv.discarded ? (rv.discarded[myRead] != 0) : false;
	}
	/* PACKED: where this lane's bases are, counted in bases from pkd.bytes */
	uint64_t myPb = 0, myPe = 0;
	if (PACKED && have) { myPb = 4 * pkd.off[myRead] + (myStart - rv.offsets[myRead]); myPe = myPb + (myEnd - myStart); }
	uint32_t tRaw = 0, tGood = 0;
	uint32_t done = 0;
	while (done < nr) {
		const uint64_t B0 = PACKED ? __shfl(myPb, (int)done, 64) : __shfl(myStart, (int)done, 64);
		const bool fits = have && (uint32_t)lane >= done && (PACKED ? (myPb >= B0 && myPe - B0 <= (uint64_t)SK_PACKED_SPAN) : (myEnd - B0 <= (uint64_t)TILE_SPAN));
		unsigned long long fm = __ballot(fits) >> done;
		uint32_t n = ~fm ? (uint32_t)__builtin_ctzll(~fm) : 64u;
		if (n > nr - done) n = nr - done;
		if (n == 0) {                                        /* read longer than a tile */
			if (lane == 0) atomicOr(p.err, (uint32_t)ERR_READ_TOO_LONG);
			done += 1;
			continue;
		}
		uint64_t B1 = __shfl(PACKED ? myPe : myEnd, (int)(done + n - 1), 64);
		if (PACKED) {      /* (nothing says that the reads' bytes lie in the order of the reads: the tile ends where the last of its reads does) */
			uint64_t e = ((uint32_t)lane >= done && (uint32_t)lane < done + n) ? myPe : 0;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) { const uint64_t x = __shfl_xor(e, o, 64); e = x > e ? x : e; }
			B1 = e;
		}
		const uintptr_t gb = PACKED ? (uintptr_t)pkd.bytes + (uintptr_t)(B0 >> 2) : (uintptr_t)rv.bases + B0;
		const uintptr_t ab = gb & ~(uintptr_t)15;
		const uint32_t nb16 = PACKED ? (uint32_t)(((uintptr_t)pkd.bytes + (uintptr_t)((B1 + 3) >> 2) - ab + 15) >> 4) : (uint32_t)(((uintptr_t)rv.bases + B1 - ab + 15) >> 4);
		/* stage the tile: 16 bytes per lane and round, coalesced; bases leave as 2 bits each plus an N flag.  All rounds' loads are
		 * issued before the first one is packed. */
		if (PACKED) {
			const uint4 *gbp = (const uint4 *)ab;
			constexpr int STGP = (SK_GROUPS / 4 + 63) / 64;
			static_assert(SK_GROUPS % 4 == 0, "the packed staging writes four groups at a time");
			uint4 vb[STGP];
#pragma unroll
			for (int c = 0; c < STGP; c++) { const uint32_t idx = (uint32_t)lane + 64u * c; vb[c] = make_uint4(0, 0, 0, 0); if (idx < nb16 && idx < (uint32_t)(SK_GROUPS / 4)) vb[c] = gbp[idx]; }
#pragma unroll
			for (int c = 0; c < STGP; c++) {
				const uint32_t idx = (uint32_t)lane + 64u * c;
				if (idx < (uint32_t)(SK_GROUPS / 4)) {      /* (zeros behind the data) */
					((uint4 *)pk)[idx] = make_uint4(__builtin_bswap32(vb[c].x), __builtin_bswap32(vb[c].y), __builtin_bswap32(vb[c].z), __builtin_bswap32(vb[c].w));
					((uint2 *)nm)[idx] = make_uint2(0, 0);
				}
			}
		} else {
			const uint4 *gbp = (const uint4 *)(rv.bases + (ptrdiff_t)(ab - (uintptr_t)rv.bases));
			constexpr int STG = (TILE_BUF / 16 + 63) / 64;
			uint4 vb[STG];
#pragma unroll
			for (int c = 0; c < STG; c++) { const uint32_t idx = (uint32_t)lane + 64u * c; vb[c] = make_uint4(0, 0, 0, 0); if (idx < nb16) vb[c] = gbp[idx]; }
#pragma unroll
			for (int c = 0; c < STG; c++) {
				const uint32_t idx = (uint32_t)lane + 64u * c;
				const uint32_t w4[4] = {vb[c].x, vb[c].y, vb[c].z, vb[c].w};
				uint32_t packed = 0, nflags = 0;
#pragma unroll
				for (int b = 0; b < 16; b++) {
					const uint32_t code = base_code((uint8_t)(w4[b >> 2] >> (8 * (b & 3))));
					packed |= (code & 3u) << (30 - 2 * b);          /* markup packs as A */
					nflags |= (code >> 2) << b;
				}
				if (idx < nb16 + 4u && idx < (uint32_t)SK_GROUPS) { pk[idx] = idx < nb16 ? packed : 0u; nm[idx] = idx < nb16 ? (uint16_t)nflags : (uint16_t)0; }      /* (four groups of padding behind the data) */
			}
		}
		sk_wave_lds_order();

		const bool active = have && (uint32_t)lane >= done && (uint32_t)lane < done + n && !myDiscard;
		const uint32_t L = active ? (uint32_t)(myEnd - myStart) : 0;
		const uint32_t rbOff = active ? (PACKED ? (uint32_t)(myPb - 4 * (uint64_t)(ab - (uintptr_t)pkd.bytes)) : (uint32_t)(gb - ab) + (uint32_t)(myStart - B0)) : 0u;
		if (PACKED && pkd.mk_off) {      /* applyMarkup (src/TwoBitSequence.cpp:314-340) on the tile */
			if (active) {
				const uint64_t inRead = myStart - rv.offsets[myRead];      /* a unit of a long read starts that far inside it */
				for (uint64_t e = pkd.mk_off[myRead]; e < pkd.mk_off[myRead + 1]; e++) {
					const uint64_t mp = pkd.mk_pos[e];
					if (mp < inRead || mp - inRead >= L) continue;
					const uint32_t x = rbOff + (uint32_t)(mp - inRead), code = base_code(pkd.mk_char[e]);
					if (code >> 2) atomicOr((uint32_t *)nm + (x >> 5), 1u << (x & 31u));
					else { const uint32_t sh = 30u - 2u * (x & 15u); atomicAnd(pk + (x >> 4), ~(3u << sh)); atomicOr(pk + (x >> 4), code << sh); }
				}
			}
			sk_wave_lds_order();
		}
		uint32_t Lmax = L;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { uint32_t x = __shfl_xor(Lmax, o, 64); Lmax = x > Lmax ? x : Lmax; }
		const uint64_t ord0 = rv.stream_base + myStart;      /* + k-mer index = stream ordinal of the occurrence */
		uint32_t zc = 0;
		constexpr int ZN = W == 1 ? 1 : (W == 2 ? 2 : 3);      /* history of N flags, one bit per position: k + 1 bits */
		uint64_t zbits[3] = {0, 0, 0};
		uint32_t mf = 0, mr = 0;
		/* WIN > 16: the minimum over the last WIN hashes is the smaller of the block minimum over the last 16 and the block minimum of
		 * WIN - 16 positions earlier -- a delay line of WIN - 16 registers (mprev), indexed by the position modulo its length, which is a
		 * constant after unrolling as long as that length divides the 16 positions of an iteration: WIN = 17, 18, 20, 24, 32 */
		constexpr int WB = WIN > 16 ? 16 : WIN;
		constexpr int WD = WIN > 16 ? WIN - 16 : 1;
		static_assert(WIN <= 16 || SK_WINDOW % WD == 0, "the delay line of a window above 16 has to divide SK_WINDOW");
		uint32_t hs[WB], mprev[WD];
#pragma unroll
		for (int i = 0; i < WB; i++) hs[i] = 0xffffffffu;
#pragma unroll
		for (int i = 0; i < WD; i++) mprev[i] = 0xffffffffu;
		uint32_t pref = 0xffffffffu;
		bool runOpen = false, runInWin = false;
		uint32_t runStart = 0, runN = 0, runMh = 0;
		/* records booked but not yet written: start | n << 16 | 1 << 24 | 1 << 30 (an address, not a booking) | 1 << 31, minimizer hash, what the booking add returned */
		uint32_t q_info[SKL_RR], q_mh[SKL_RR]; unsigned long long q_booked[SKL_RR];
#pragma unroll
		for (int r = 0; r < SKL_RR; r++) { q_info[r] = 0; q_mh[r] = 0; q_booked[r] = 0; }
		auto flush_pending = [&]() {
			uint64_t at[SKL_RR]; bool waits[SKL_RR];
#pragma unroll
			for (int r = 0; r < SKL_RR; r++) {
				waits[r] = false; at[r] = ~0ull;
				if (q_info[r] >> 31) {
					const uint32_t nn = (q_info[r] >> 16) & 0xffu;
					if ((q_info[r] >> 30) & 1u) at[r] = q_booked[r];
					else at[r] = sk_append_settle(sp.state, sk_list_of(q_mh[r], sp.list_bits), 1 + sk_base_granules(nn, k), q_booked[r], slab, pool, waits[r]);
				}
			}
#pragma unroll
			for (int r = 0; r < SKL_RR; r++) if (waits[r]) {
				const uint32_t nn = (q_info[r] >> 16) & 0xffu;
				at[r] = sk_append(sp.state, sk_list_of(q_mh[r], sp.list_bits), 1 + sk_base_granules(nn, k), slab, pool);
			}
#pragma unroll
			for (int r = 0; r < SKL_RR; r++) {
				if ((q_info[r] >> 31) && at[r] != ~0ull) {
					const uint32_t start = q_info[r] & 0xffffu, nn = (q_info[r] >> 16) & 0xffu;
					const uint32_t nbg = sk_base_granules(nn, k);
					uint4 *dst = (uint4 *)pool.base + at[r];
					const uint64_t ord = ord0 + start;
					dst[0] = make_uint4((uint32_t)ord, (uint32_t)(ord >> 32) | (nn << 8) | (1u << 16) | ((1 + nbg) << 17), q_mh[r], wbits);
					const uint32_t xb = rbOff + start;
					for (uint32_t b = 0; b < nbg; b++)
						dst[1 + b] = make_uint4(sk_bases16(pk, xb + 64 * b), sk_bases16(pk, xb + 64 * b + 16), sk_bases16(pk, xb + 64 * b + 32), sk_bases16(pk, xb + 64 * b + 48));
				}
				q_info[r] = 0;
			}
		};
		for (uint32_t jb = 0; jb < Lmax || __any(runOpen); jb += SK_WINDOW) {
			const uint32_t nmw = sk_flags16(nm, rbOff + jb);
			const int32_t xm = (int32_t)(rbOff + jb) - (int32_t)sp.off;
			const uint32_t mpkw = xm >= 0 ? sk_bases16(pk, (uint32_t)xm) : (xm > -16 ? sk_bases16(pk, 0u) >> (2 * (uint32_t)(-xm)) : 0u);
			const bool cinOpen = runOpen; const uint32_t cinStart = runStart, cinN = runN, cinMh = runMh;
			uint32_t Sm = 0, Vm = 0;
			runInWin = false;
			/* the window's positions: which of them end a k-mer (km), which of those hold no N in their k bases (Vm) */
			const bool live = jb < L;
			const uint32_t nin = live ? (L - jb < (uint32_t)SK_WINDOW ? L - jb : (uint32_t)SK_WINDOW) : 0u;
			const uint32_t inmask = (1u << nin) - 1u;
			const uint32_t tk = jb + 1 >= k ? 0u : k - 1 - jb;            /* first position of the window that ends a k-mer (may be >= 16: none) */
			const uint32_t km = tk < 16u ? (inmask & ~((1u << tk) - 1u)) : 0u;
			const uint32_t zm = nmw & inmask;
			uint32_t zcpos = 0;                                              /* positions at which the k bases behind hold an N */
			if (__any(zm != 0 || zc != 0)) {
				if (k >= (uint32_t)SK_WINDOW) {
					uint32_t field;                                          /* history bits k-16 .. k-1 (bit b = position jb - 1 - b) */
					{
						const uint32_t lo = k - 16u, wi = lo >> 6, sh = lo & 63u;
						uint64_t a = zbits[wi < (uint32_t)ZN ? wi : ZN - 1];
						if (wi >= (uint32_t)ZN) a = 0;
						uint64_t b = (wi + 1 < (uint32_t)ZN) ? zbits[wi + 1 < (uint32_t)ZN ? wi + 1 : ZN - 1] : 0ull;
						field = (uint32_t)((sh ? (a >> sh) | (b << (64u - sh)) : a) & 0xffffu);
					}
					const uint32_t lm = __builtin_bitreverse32(field) >> 16;     /* bit t: the flag that leaves at position jb + t */
					uint32_t zcw = zc;
#pragma unroll
					for (int t = 0; t < SK_WINDOW; t++) { zcw += (zm >> t) & 1u; zcw -= (lm >> t) & 1u; zcpos |= (zcw != 0 ? 1u : 0u) << t; }
					zc = zcw;
					if (ZN > 2) zbits[2] = (zbits[2] << 16) | (zbits[1] >> 48);
					if (ZN > 1) zbits[1] = (zbits[1] << 16) | (zbits[0] >> 48);
					zbits[0] = (zbits[0] << 16) | (uint64_t)(__builtin_bitreverse32(zm) >> 16);
				} else {
#pragma nounroll
					for (uint32_t t = 0; t < (uint32_t)SK_WINDOW; t++) {
						const bool z = ((zm >> t) & 1u) != 0;
						if (ZN > 2) zbits[2] = (zbits[2] << 1) | (zbits[1] >> 63);
						if (ZN > 1) zbits[1] = (zbits[1] << 1) | (zbits[0] >> 63);
						zbits[0] = (zbits[0] << 1) | (z ? 1ull : 0ull);
						zc += z ? 1u : 0u;
						zc -= (uint32_t)((zbits[ZN == 1 ? 0 : (k >> 6)] >> (k & 63)) & 1ull);
						zcpos |= (zc != 0 ? 1u : 0u) << t;
					}
				}
			} else if (live) {      /* no N anywhere near: the histories move on by sixteen clean positions */
				if (ZN > 2) zbits[2] = (zbits[2] << 16) | (zbits[1] >> 48);
				if (ZN > 1) zbits[1] = (zbits[1] << 16) | (zbits[0] >> 48);
				zbits[0] <<= 16;
			}
			Vm = good ? (km & ~zcpos) : 0u;
			tRaw += (uint32_t)__builtin_popcount(km); tGood += (uint32_t)__builtin_popcount(Vm);
#pragma unroll
			for (int t = 0; t < SK_WINDOW; t++) {
				/* minimizer of the k-mer that ends at position jb + t: m-mer ending sp.off positions back, canonical, hashed; minimum of the
				 * last WIN of them by block decomposition (prefix minimum of the current block, suffix minima of the one before) */
				const uint32_t mc = (mpkw >> (30 - 2 * t)) & 3u;
				mf = ((mf << 2) | mc) & mmask;
				mr = (mr >> 2) | ((3u - mc) << mtop);
				const uint32_t x = sk_mmer_hash(mf < mr ? mf : mr);
				const int r = t % WB;
				pref = r == 0 ? x : (x < pref ? x : pref);
				hs[r] = x;
				uint32_t M = pref;
				if (r < WB - 1) { const uint32_t sfx = hs[r + 1 < WB ? r + 1 : 0]; M = sfx < M ? sfx : M; }
				else {
#pragma unroll
					for (int u = WB - 2; u >= 0; u--) hs[u] = hs[u] < hs[u + 1] ? hs[u] : hs[u + 1];
				}
				if constexpr (WIN > 16) { const uint32_t m16 = M; const uint32_t before = mprev[t % WD]; M = before < M ? before : M; mprev[t % WD] = m16; }
				const bool valid = ((Vm >> t) & 1u) != 0;
				const bool cont = runOpen && M == runMh && runN < SK_MAX_N;
				const bool nw = valid && !cont;
				Sm |= nw ? (1u << t) : 0u;
				runStart = nw ? jb + (uint32_t)t + 1 - k : runStart;
				runN = nw ? 1u : runN + ((valid && cont) ? 1u : 0u);
				runMh = nw ? M : runMh;
				runInWin = runInWin || nw;
				runOpen = valid || (runOpen && (uint32_t)t < tk);
				mhr[t * 64 + lane] = M;
			}
			if (!live) runOpen = false;
			/* gather: every run that ended in this window becomes a record of its list (as in sk_extract_kernel; every record is uniform) */
			const uint32_t brk = (~Vm | Sm) & 0xffffu;
			bool pendC = false; uint32_t lead = 16;
			if (cinOpen) { lead = (uint32_t)__builtin_ctz(brk | 0x10000u); pendC = lead < 16; }
			uint32_t Srem = Sm;
			if (runOpen && runInWin && Sm) Srem &= ~(1u << (31 - __builtin_clz(Sm)));      /* the run still open is the last one begun */
			flush_pending();
			bool firstRound = true;
			const uint32_t hotNow = slab->hot_list;
			while (__any(pendC || Srem)) {
#pragma unroll
				for (int r = 0; r < SKL_RR; r++) {
					q_info[r] = 0;
					if (pendC) { pendC = false; q_info[r] = (1u << 31) | (1u << 24) | ((cinN + lead) << 16) | cinStart; q_mh[r] = cinMh; }
					else if (Srem) {
						const uint32_t pos = (uint32_t)__builtin_ctz(Srem); Srem &= Srem - 1;
						const uint32_t end = pos + 1 + (uint32_t)__builtin_ctz((brk >> (pos + 1)) | (1u << (15 - pos)));
						q_info[r] = (1u << 31) | (1u << 24) | ((end - pos) << 16) | (jb + pos + 1 - k);
						q_mh[r] = mhr[pos * 64 + lane];
					}
					if (q_info[r] >> 31) {
						const uint32_t nn = (q_info[r] >> 16) & 0xffu;
						const uint32_t need = 1 + sk_base_granules(nn, k), myList = sk_list_of(q_mh[r], sp.list_bits);
						if (myList == hotNow) { q_booked[r] = sk_append_hot(slab, myList, need, pool); q_info[r] |= 1u << 30; }
						else q_booked[r] = atomicAdd(sp.state + myList, (unsigned long long)need);
					}
				}
				if (!firstRound || __any(pendC || Srem)) flush_pending();      /* more rounds to come: settle now */
				firstRound = false;
			}
			if (slab->next >= 64u) {
				__builtin_amdgcn_wave_barrier();
				if (lane == 0) { const uint32_t used = slab->next; slab->base[0] = slab->base[1]; slab->base[1] = atomicAdd(pool.head, 64u); slab->next = used >= 128u ? 64u : used - 64u; }
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
				__builtin_amdgcn_wave_barrier();
			}
		}
		flush_pending();                   /* the last window's records: their bases are read from this tile */
		done += n;
		sk_wave_lds_order();               /* all lanes are done reading the tile before it is overwritten */
	}
	nRaw += tRaw; nGood += tGood;
	if (sp.track && have && !myDiscard) {
		SkTrackRec *tr = sp.track + myRead;
		if (rv.u_start) { atomicAdd(&tr->raw, tRaw); atomicAdd(&tr->good, tGood); atomicMax(&tr->end_ordinal, (unsigned long long)(rv.stream_base + myEnd)); }
		else { tr->raw = tRaw; tr->good = tGood; tr->end_ordinal = rv.stream_base + myEnd; }
	}
	}
	/* chunks of the slabs nobody took belong to no list */
	__builtin_amdgcn_wave_barrier();
	{
		const uint32_t used = slab->next < 128u ? slab->next : 128u;
		for (uint32_t idx = used + (uint32_t)lane; idx < 128u; idx += 64) { const uint32_t c = slab->base[idx >> 6] + (idx & 63u); if (c < pool.cap) { pool.chunk_list[c] = NO_CHUNK; pool.chunk_count[c] = 0; } }
		if (lane == 0 && slab->hot_list < SK_LIST_LOCKED) {      /* the open chunk of the wavefront's hot chain */
			const uint32_t hc = (uint32_t)(slab->hot_state >> 32), hf = (uint32_t)slab->hot_state;
			if (hc < pool.cap) pool.chunk_count[hc] = hf < SK_CHUNK_G ? hf : SK_CHUNK_G;
		}
	}
	nRaw = wave_sum(nRaw); nGood = wave_sum(nGood);
	if (lane == 0) { atomicAdd(&p.stats->raw, nRaw); atomicAdd(&p.stats->good, nGood); }
}

/* reverse complement of a left-justified k-mer of W words, left-justified again */
template <int W> __device__ __forceinline__ Key<W> key_revcomp(const Key<W> &f, uint32_t k) {
	Key<W> r;
#pragma unroll
	for (int i = 0; i < W; i++) {
		uint64_t x = f.w[W - 1 - i];
		x = ((uint64_t)__builtin_bitreverse32((uint32_t)x) << 32) | __builtin_bitreverse32((uint32_t)(x >> 32));      /* bit reversal of 64 bits */
		x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);                                  /* bits of a base back in order */
		r.w[i] = ~x;
	}
	/* the string now ends with the complement of the pad: shift it out */
	const uint32_t sh = 64u * W - 2u * k;
	const uint32_t ws = sh >> 6, bs = sh & 63u;
	Key<W> o;
#pragma unroll
	for (int i = 0; i < W; i++) {
		/* words i + ws and i + ws + 1 of r (zero beyond the last one), by selects over the W words: r.w[] indexed with the run-time word
		 * shift lived in scratch memory -- a store, a load and a wait for the vector-memory counter per k-mer of a multi-word key */
		const int a = i + (int)ws;
		uint64_t hi = 0ull, lo = 0ull;
#pragma unroll
		for (int j = 0; j < W; j++) { if (j == a) hi = r.w[j]; if (j == a + 1) lo = r.w[j]; }
		o.w[i] = bs ? (hi << bs) | (lo >> (64 - bs)) : hi;
	}
	return o;
}

/* ------------------------------------------------------------------ count over super-k-mer lists */
/* One block per list, lists taken SK_LBATCH at a time.  The list's chunks are staged SK_STAGE_CHUNKS at a time (the next group
 * is held in registers meanwhile), every wavefront finds the record starts of one chunk by following the granule counts in
 * the headers, the records' k-mer counts are scanned, and then every thread expands and inserts k-mers t, t + 256, ... of the
 * group: the record's bases give the forward word, key_revcomp the other strand, the smaller one is the key.  Table, sub-pass
 * splitting and the emission of kept entries are those of count_kernel (COUNT_DIR values). */
static const int SK_STAGE_CHUNKS = 4;                          /* = wavefronts of the block */
static const int SK_STAGE_G = SK_STAGE_CHUNKS * SK_CHUNK_G;    /* 256 granules = one per thread */
static const uint32_t SK_LBATCH = 24;
static const uint32_t SK_DESC_CAP = 160;                       /* chunk descriptors of a batch of lists kept in LDS */

/* Long lists.  A list is counted by one block, and a block gets through ~4 x 10^5 k-mers per millisecond: a minimizer that draws
 * 10^7 k-mers (a homopolymer, a repeat family) would keep one block busy for tens of milliseconds after the rest of the chip is
 * done.  Lists of more than long_threshold chunks are therefore left out of the main launch and counted in a second one whose work
 * items are chunk RANGES of them: every block counts its range in LDS as usual, but instead of emitting entries it adds what its
 * table holds into a device hash table (the open-addressed table of build_mode 1: count | forward, f64 weight sum, first-sighting
 * word -- add, add, min), and sk_merge_emit_kernel turns that table into entries.  Ranges of one list share keys; the merge makes
 * the result the same as if one block had seen the whole list (the f64 sum is formed in another order: within the tolerance). */
template <int W> struct SkLong {
	const uint64_t *item_c0, *item_c1;      /* item mode: chunk range of work item i (device); null = the lists of list_start */
	uint64_t n_items;
	uint64_t long_threshold;                /* list mode: lists of more chunks than this are skipped (0 = none) */
	uint32_t list_first, list_stride;       /* list mode: the lists looked at are list_first, list_first + list_stride, ... -- after an owner exchange a rank only holds
	                                           the lists it owns (rank, rank + world, ...), and a job's list space is world times a rank's share */
	Table<W> merge;                         /* item mode: where the tables go */
	unsigned long long *merge_used;         /* slots of it claimed so far: beyond 5/8 of the table the launch gives up (ERR_TABLE_FULL) and the host comes back with a larger one */
};

/* EXT (extension values): six more words per slot -- the twelve tallies as 16-bit halves, exact while a block's share of a list stays
 * below 65 536 k-mers (the host cuts longer lists into pieces, SK_EXT_LONG_CHUNKS) -- and the packet of one occurrence, read only
 * when the key ends as a singleton: 60 bytes per slot, two blocks per CU */
/* (the one-weight pass over multi-word keys keeps no weight sums and no state words: 32 instead of 44 bytes per slot at W = 2) */
template <int W, int LOG2S, bool TRACK = false, bool EXT = false, bool UNI = false>
__host__ __device__ constexpr size_t sk_count_smem_bytes() { return (size_t)(1 << LOG2S) * (8 * W + 24 + (W > 1 ? 4 : 0) - (UNI && W > 1 ? 12 : 0) + (TRACK ? 8 : 0) + (EXT ? 28 : 0)) + (size_t)SK_STAGE_G * 16 + 256 + 64 + (TRACK ? 8 * (SK_TRACK_MAX + 1) : 0); }
/* a chunk of extension records holds at most ~6.4 k-mers per granule (n = 128: 21 granules): 128 chunks stay below 65 536 k-mers */
static const uint64_t SK_EXT_LONG_CHUNKS = 128;
static const int SK_EXT_BLOCKS = 3;      /* blocks per CU of the count pass with extension values (166 registers; four -- 128 registers, 34 dwords spilled -- ran 11 % slower) */
__device__ __forceinline__ uint32_t sk_ext_char(uint32_t code) { return (uint32_t)((0x584e54474341ull >> (8 * code)) & 0xffu); }      /* "ACGTNX" */

/* Life of a list in the block (round 3: two block barriers per list instead of eight):
 *   insert   the four wavefronts take the list's chunks on their own (chunk c0 + wave, + 4, ...), the table is shared through
 *            LDS atomics                                                                                        -- barrier A
 *   emit     every wavefront looks after a quarter of the table's slots: what a slot holds is classified (append() /
 *            purgeMinDepth semantics), kept entries go to the WAVEFRONT's own output slab (positions by ballot, a slab of
 *            SK_OSLAB entries is taken with one device atomic when the last one is full), and the slot is cleared on the
 *            spot -- the table is empty again when the next list starts, there is no clearing pass            -- barrier B
 * A list whose distinct keys overflow the table (rare: the lists are sized for ~40 % load) is redone in sub-passes that split
 * it by further hash bits, with barriers around every step (the cold path below).  The flags the insert phase raises
 * (claimed slots, overflow) exist twice and alternate from list to list, so nobody has to wait for their reset. */
#ifndef KMR_SKC_WAVES
#define KMR_SKC_WAVES 4
#endif
static const int SKC_WAVES = KMR_SKC_WAVES, SKC_THREADS = SKC_WAVES * 64;      /* wavefronts of a count block (they share one table) */
/* Multi-word keys whose last word ends in pad bits (k not a multiple of 32) are claimed like one-word keys: a compare-and-swap on the
 * FIRST word; the winner then writes the other words, the last one last, and whoever finds its own first word in a slot reads the last
 * word -- SK_KEY_PENDING (all ones: no padded word looks like that) until the winner has written it -- and compares.  One LDS round
 * trip per probe instead of two (state word, then the key). */
static const unsigned long long SK_KEY_PENDING = ~0ull;
/* A k-mer seen this often gets its weightedCount (and directionBias) from its sightings IN INPUT ORDER, added one after the other into a
 * float as TrackingData::track does (weightedCount += weight, src/KmerTrackingData.h:427-448) -- sat_*_kernel below.  Below it the count
 * pass's exact f64 sum rounded once is within 1e-5 * count of that whatever the order: each of the reference's n float additions is off by
 * at most 2^-24 of its partial sum (<= n, weights are <= 1), n^2 * 2^-25 in all, <= 1e-5 * n for n <= 335. */
static const uint32_t SK_ORDERED_FROM = 256;
static const unsigned long long SK_OSLAB = 2048;      /* entries a wavefront reserves at a time in the count pass's output */

/* UNI: every record of every list is uniform and carries ONE weight, f.uni_wbits (a build whose calls all went through the lean
 * extraction with the same quality character, or with none): the weight sum of a slot is its count times that weight -- exact in
 * f64, count * w needs 24 + 20 bits -- so the table takes no ds_add_f64, no weight is read or converted per k-mer, and the weight
 * part of the first-sighting word is a constant of the launch (a sixth of the inner loop's vector instructions). */
template <int W, int LOG2S, bool TRACK = false, bool EXT = false, bool UNI = false>
__global__ __launch_bounds__(SKC_THREADS, EXT ? (W == 1 ? (LOG2S <= 9 ? SK_EXT_BLOCKS : 2) : 1) : ((W == 1 && LOG2S <= 10 && !TRACK) ? 4 : (W == 2 && LOG2S <= 10 && !TRACK ? (UNI ? 4 : 3) : 1)))
void sk_count_kernel(PoolView pool, const uint64_t *list_start, const uint64_t *list_chunks, uint64_t n_lists, uint32_t k,
                     CountOut out, FinalizeParams f, unsigned int *work_counter, uint32_t dbgFlags, SkTrackView tv, SkLong<W> lg) {
	static_assert(!UNI || (!TRACK && !EXT), "the one-weight count pass is the plain one's");
	constexpr int S = 1 << LOG2S;
	/* LEAN: the one-weight pass over multi-word keys keeps neither weight sums (count x weight at emit) nor state words (the host sends only
	 * k with pad bits in the last key word here: the claim protocol of SK_KEY_PENDING needs none) -- W = 2: 37 KB a block, four to a CU */
	constexpr bool LEAN = UNI && W > 1;
#ifndef KMR_SKC_LIMIT_PCT
#define KMR_SKC_LIMIT_PCT 80      /* share of the table a list may fill before it is redone in sub-passes (C2 count pass at 70 / 80 / 85 / 90: 9.84 / 9.54 / 9.57 / 9.55 ms: overflowing lists are not what the uneven lists cost) */
#endif
	constexpr uint32_t LIMIT = (uint32_t)(S * (KMR_SKC_LIMIT_PCT / 100.0));
	constexpr int WSLOTS = S / SKC_WAVES;        /* slots a wavefront looks after in the emit phase */
	extern __shared__ __attribute__((aligned(16))) uint8_t csm[];
	uint64_t *tkeys = (uint64_t *)csm;                                 /* [S][W] */
	unsigned long long *tcnt = (unsigned long long *)(tkeys + (size_t)S * W);
	double *twsum = (double *)(tcnt + S);
	unsigned long long *tfirst = (unsigned long long *)(twsum + (LEAN ? 0 : S));
	uint32_t *tstate = (uint32_t *)(tfirst + S);                       /* W > 1 only, and not LEAN */
	uint32_t *ttally = tstate + (W > 1 && !LEAN ? S : 0);                       /* EXT only: [S][6] tallies (left A C | G T | N X, right A C | G T | N X as 16-bit halves), [S] packets */
	uint32_t *tpkt = ttally + (EXT ? 6 * S : 0);
	uint4 *stage = (uint4 *)(tpkt + (EXT ? S : 0));                    /* 16-byte aligned: every table array is a multiple of 16 bytes */
	uint8_t *recOf = (uint8_t *)(stage + SK_STAGE_G);                  /* [4][64] per wavefront: header lane of the record a lane's first k-mer lies in */
	/* size tracker: the second-smallest first-sighting word of every slot, and this block's share of the two difference arrays */
	unsigned long long *tsecond = (unsigned long long *)(csm + (size_t)S * (8 * W + 24 + (W > 1 ? 4 : 0) + (EXT ? 28 : 0)) + (size_t)SK_STAGE_G * 16 + 256 + 64);
	unsigned int *trkU = (unsigned int *)(tsecond + (TRACK ? S : 0)), *trkS = trkU + (SK_TRACK_MAX + 1);
	__shared__ uint32_t s_list, s_sp;
	__shared__ uint32_t s_claimed[2], s_overflow[2];
	__shared__ uint32_t s_stackBits[40], s_stackVal[40];
	__shared__ unsigned long long s_c0[SK_LBATCH + 1], s_c1[SK_LBATCH + 1];      /* chunk range of every list (work item) of the batch */
	__shared__ uint32_t s_dchunk[SK_DESC_CAP];
	__shared__ uint8_t s_dcount[SK_DESC_CAP];
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	if (TRACK) for (int i = t; i < 2 * (int)(SK_TRACK_MAX + 1); i += SKC_THREADS) trkU[i] = 0;
	const uint32_t vw = EXT ? 15 : 3;
	unsigned long long uniq = 0, single = 0, keptW = 0, keptS = 0, satK = 0, satS = 0;
	unsigned long long uniqW = 0, singleW = 0;      /* wave-uniform: used slots and slots with a count of one, from ballots */
	/* this wavefront's output slabs: [wpos, wend) of the weak entries, [spos, send) of the singletons */
	unsigned long long wpos = 0, wend = 0, spos = 0, send = 0;
	bool outFull = false;
	if (t == 0) { s_claimed[0] = s_claimed[1] = 0; s_overflow[0] = s_overflow[1] = 0; s_sp = 0; }
	for (int i = t; i < S; i += SKC_THREADS) { tkeys[(size_t)i * W] = EMPTY_KEY; if (W > 1) tkeys[(size_t)i * W + W - 1] = SK_KEY_PENDING; tcnt[i] = 0; if constexpr (!LEAN) twsum[i] = 0.0; tfirst[i] = NO_FIRST; if constexpr (W > 1 && !LEAN) tstate[i] = 0; if (TRACK) tsecond[i] = NO_FIRST; }
	if (EXT) for (int i = t; i < 6 * S; i += SKC_THREADS) ttally[i] = 0;
	lds_barrier();
	/* classify() of kmr_kernels.hpp folded into launch-wide scalars: a count of one goes to class singC when singletons are separate,
	 * any other count below weakMin is dropped */
	const uint32_t singC = f.has_singletons ? (f.min_depth > 1 ? 0u : 2u) : 3u;
	const uint32_t weakMin = ((!f.has_singletons || f.min_depth > 2) && f.min_depth != 1) ? f.min_depth : 0u;
	const uint32_t uniFirst = UNI ? (first_weight_bits(__uint_as_float(f.uni_wbits)) & 0x7fffffu) : 0u;      /* the weight bits of every first-sighting word */
	const uint4 *poolg = (const uint4 *)pool.base;
	uint4 pre = make_uint4(0, 0, 0, 0); uint32_t preCount = 0; uint64_t preList = ~0ull;      /* this wavefront's first chunk of the list named */
	uint32_t gen = 0;                                  /* lists this block has counted: the flag pair in use is gen & 1 */

	const bool itemMode = lg.item_c0 != nullptr;
	const uint32_t grab = itemMode ? 1u : SK_LBATCH;      /* work items are large: one at a time */
	for (;;) {
		if (t == 0) s_list = atomicAdd(work_counter, grab);
		lds_barrier();
		const uint64_t lfirst = s_list;
		const uint64_t stride = lg.list_stride ? lg.list_stride : 1u;
		const uint64_t n_work = itemMode ? lg.n_items : (n_lists > lg.list_first ? (n_lists - lg.list_first + stride - 1) / stride : 0);
		if (lfirst >= n_work) break;
		const uint32_t nl = (uint32_t)(n_work - lfirst < (uint64_t)grab ? n_work - lfirst : (uint64_t)grab);
		if ((uint32_t)t < nl) {
			uint64_t a, b;
			if (itemMode) { a = lg.item_c0[lfirst + t]; b = lg.item_c1[lfirst + t]; }
			else { const uint64_t l = lg.list_first + (lfirst + t) * stride; a = list_start[l]; b = list_start[l + 1]; if (lg.long_threshold && b - a > lg.long_threshold) b = a; }      /* a long list: the second launch's */
			s_c0[t] = a; s_c1[t] = b;
		}
		lds_barrier();
		/* the batch's chunk descriptors are contiguous in list_chunks: the first SK_DESC_CAP of them wait in LDS, so that a chunk can be
		 * requested a whole list ahead without a descriptor load in front of it */
		/* (with a stride the lists in between hold no chunks -- they went to their owners -- so the batch's descriptors are still one run) */
		const uint64_t dbase = itemMode ? s_c0[0] : list_start[lg.list_first + lfirst * stride], dend = itemMode ? s_c1[0] : list_start[lg.list_first + (lfirst + nl - 1) * stride + 1];
		const uint64_t dcached = dend - dbase < (uint64_t)SK_DESC_CAP ? dend - dbase : (uint64_t)SK_DESC_CAP;      /* descriptors [dbase, dbase + dcached) are in LDS (item mode: the first item's) */
		for (uint64_t i = (uint64_t)t; i < dcached; i += SKC_THREADS) { const uint64_t d = list_chunks[dbase + i]; s_dchunk[i] = (uint32_t)d; s_dcount[i] = (uint8_t)(d >> 32); }
		lds_barrier();
		/* this wavefront's chunk ci: granule `lane` of it (zeros past its fill count) and the fill count */
		auto describe = [&](uint64_t ci, uint32_t &chunk, uint32_t &count) {
			if (ci - dbase < dcached) { chunk = s_dchunk[ci - dbase]; count = s_dcount[ci - dbase]; }
			else { const uint64_t d = list_chunks[ci]; chunk = (uint32_t)d; count = (uint32_t)(d >> 32); }
		};
		auto fetch = [&](uint64_t ci, uint4 &v, uint32_t &count) {
			uint32_t chunk;
			describe(ci, chunk, count);
			v = make_uint4(0, 0, 0, 0);
			if ((uint32_t)lane < count) v = poolg[(size_t)chunk * SK_CHUNK_G + lane];
		};
		for (uint32_t lj = 0; lj < nl; lj++) {
			const uint64_t c0 = s_c0[lj], c1 = s_c1[lj];
			if (c0 == c1) continue;
			if (itemMode && (__hip_atomic_load(out.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ERR_TABLE_FULL)) continue;      /* uniform enough: the launch is void anyway */
			uint32_t fl = gen & 1u;                    /* this list's flags */
			gen++;
			/* ---- insert: wavefront w takes chunks c0 + w, c0 + w + 4, ... of the list, each one on its own: it stages the chunk in its
			 * quarter of the staging area, finds the record starts, maps k-mer slots to records and inserts (only the k-mers whose hash
			 * bits `bits` equal val: sub-passes of a split list) */
			auto insert_pass = [&](const uint32_t bits, const uint32_t val, const bool firstPass) {
				const uint32_t subMask = bits ? ((1u << bits) - 1) : 0;
				const bool padded = W > 1 && (LEAN || (k & 31u) != 0);      /* the last key word ends in pad bits: the claim protocol of SK_KEY_PENDING */
				uint4 *wstage = stage + wv * SK_CHUNK_G;
				uint8_t *wrecOf = recOf + wv * 64;
				uint4 cur = make_uint4(0, 0, 0, 0); uint32_t curCount = 0;
				if (firstPass && preList == lfirst + lj) { cur = pre; curCount = preCount; }        /* requested while the list before was counted */
				else if (c0 + wv < c1) fetch(c0 + wv, cur, curCount);
				bool wantPre = firstPass;      /* this wavefront's first chunk of the next list is requested below, once the current chunk is out of its registers */
				if (firstPass) preList = ~0ull;
				for (uint64_t ci = c0 + wv; ci < c1 && !s_overflow[fl] && s_claimed[fl] <= LIMIT; ci += SKC_WAVES) {
					wstage[lane] = cur;
					/* record starts: follow the granule counts from granule 0 */
					const uint32_t glen = (cur.y >> 17) & 0x7fu;
					uint32_t claimedHere = 0;
					unsigned long long starts = 0;
					/* the usual chunk holds two-granule records only (a header and up to 64 bases): if every even granule says "2" where a
					 * header keeps its granule count, every even granule is a header (granule 0 is one, and each one vouches for the next) */
					if (__all((lane & 1) != 0 || (uint32_t)lane >= curCount || glen == 2u)) starts = 0x5555555555555555ull & (curCount >= 64u ? ~0ull : ((1ull << curCount) - 1ull));
					/* (the usual chunk of a build with extension values: header, bases, qualities -- three granules a record) */
					else if (EXT && __all(((uint32_t)lane * 43u >> 7) * 3u != (uint32_t)lane || (uint32_t)lane >= curCount || glen == 3u)) starts = 0x9249249249249249ull & (curCount >= 64u ? ~0ull : ((1ull << curCount) - 1ull));
					else {
						/* Records of mixed sizes (a weight per k-mer, extension values of long runs).  Every granule that READS like a header -- a k-mer
						 * count and a granule count that agree -- is a candidate; the candidates are exactly the headers iff following the granule
						 * counts maps them one-to-one onto themselves without the first (each but granule 0 is pointed at by one, the pointers only
						 * go forward): eight ballots by distance instead of a scalar walk with a v_readlane per record.  A data granule that
						 * passes for a header breaks the equality, and the walk decides. */
						const uint32_t hn = (cur.y >> 8) & 0xffu;
						const bool cand = (uint32_t)lane < curCount && hn >= 1u && hn <= SK_MAX_N && glen == sk_rec_granules(hn, k, ((cur.y >> 16) & 1u) != 0, EXT) && ((cur.y >> 30) & 1u) == (EXT ? 1u : 0u);
						const unsigned long long C = __ballot(cand), inRange = curCount >= 64u ? ~0ull : ((1ull << curCount) - 1ull);
						unsigned long long N = 0;
#pragma unroll
						for (uint32_t d = 2; d <= 9; d++) N |= __ballot(cand && glen == d) << d;
						if ((C & 1ull) && !__any(cand && glen > 9u) && (N & inRange) == (C & ~1ull)) starts = C;
						else for (uint32_t pos = 0; pos < curCount; ) {
						starts |= 1ull << pos;
						const uint32_t step = (uint32_t)__builtin_amdgcn_readlane((int)glen, (int)pos);
						pos += step ? step : SK_CHUNK_G;        /* a zero would never end: a corrupt chunk is dropped */
					}
					}
					/* The chunk's k-mers are dealt out evenly: with T of them, lane l takes slots [l Lk, (l + 1) Lk), Lk = ceil(T / 64),
					 * in the order the records lie in the chunk.  The record a lane starts in is told to it by that record's header
					 * lane (which knows the slots it covers from a prefix sum of the n's); the lane cuts its first k-mer out of the
					 * record's bases and reverse-complements it once, the others follow by rolling one base in, and when a record
					 * runs out the lane starts the next one the same way.  The table slot of the next k-mer is read while the
					 * current one is being added. */
					const bool isStart = (starts >> lane) & 1ull;
					const uint32_t myN = isStart ? (cur.y >> 8) & 0xffu : 0u;
					const uint32_t incl = sk_wave_scan_u32(myN);      /* (six LDS permutes with their address arithmetic cost the pass 0.85 ms) */
					const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
					const uint32_t myOff = incl - myN;
					const uint32_t Lk = (T + 63u) >> 6;
					if (myN) {
						/* floor(x / Lk) as (x + 0.5) * (1 / Lk): x < 2^14 and Lk <= 128 leave the product at least 0.5 / Lk away from an integer,
						 * a thousand times the rounding error of the reciprocal and the product (two f32 divisions are 24 instructions) */
						const float inv = __builtin_amdgcn_rcpf((float)Lk);
						const uint32_t l0 = (uint32_t)(((float)(myOff + Lk - 1) + 0.5f) * inv), l1 = (uint32_t)(((float)(myOff + myN - 1) + 0.5f) * inv);
						for (uint32_t l = l0; l <= l1 && l < 64u; l++) wrecOf[l] = (uint8_t)lane;
					}
					sk_wave_lds_order();
					const uint32_t e0 = (uint32_t)lane * Lk;
					uint32_t left = e0 < T ? (T - e0 < Lk ? T - e0 : Lk) : 0u;      /* k-mers this lane still has to do */
					uint32_t rs = left ? wrecOf[lane] : 0u;
					uint32_t j = e0 - (uint32_t)__shfl((int)myOff, (int)rs, 64);
					uint32_t hx = (uint32_t)__shfl((int)cur.x, (int)rs, 64), hy = (uint32_t)__shfl((int)cur.y, (int)rs, 64), hw = (uint32_t)__shfl((int)cur.w, (int)rs, 64);
					if (SK_DBG(dbgFlags, 4)) left = 0;
					/* The chunk is in LDS and its headers have been read: only NOW are the next chunks requested (this list's c + 4 and, once
					 * per list, this wavefront's first chunk of the next list).  Requested at the top of the iteration -- before the current
					 * chunk's registers were consumed -- the compiler had to drain the vector-memory counter to zero to be sure of the older
					 * load, i.e. it waited for the request it had just made: every chunk paid a full HBM latency and nothing was prefetched. */
					uint4 nxt = make_uint4(0, 0, 0, 0); uint32_t nxtCount = 0, nxtChunk = 0, preChunk = 0;
					/* (descriptors first -- the rare one that is not cached in LDS is a load of its own that has to be waited for -- then the
					 * two chunk requests back to back with nothing to wait for in between) */
					const bool wantNxt = ci + SKC_WAVES < c1;
					bool havePre = false;
					if (wantNxt) describe(ci + SKC_WAVES, nxtChunk, nxtCount);
					if (wantPre) {
						wantPre = false;
						if (lj + 1 < nl) { const uint64_t n0 = s_c0[lj + 1], n1 = s_c1[lj + 1]; if (n0 + wv < n1) { describe(n0 + wv, preChunk, preCount); havePre = true; preList = lfirst + lj + 1; } }
					}
					if (wantNxt && (uint32_t)lane < nxtCount) nxt = poolg[(size_t)nxtChunk * SK_CHUNK_G + lane];
					if (havePre) { pre = make_uint4(0, 0, 0, 0); if ((uint32_t)lane < preCount) pre = poolg[(size_t)preChunk * SK_CHUNK_G + lane]; }
					/* the record the lane is in: its k-mer count, weights, first ordinal */
					uint32_t n = 0; const uint32_t *ww = nullptr; bool uniformW = true; uint64_t ord0 = 0;
					const uint8_t *xq = nullptr;      /* EXT: the record's neighbour qualities */
					uint32_t leftCarry = 0;           /* EXT: the base left of the next k-mer to be made, when it is a base of the same record */
					auto enter_record = [&]() {      /* header in hx, hy, hw */
						n = (hy >> 8) & 0xffu;
						ww = (const uint32_t *)(wstage + rs + 1) + 4 * sk_base_granules(n, k);
						uniformW = ((hy >> 16) & 1u) != 0;
						ord0 = (uint64_t)hx | ((uint64_t)(hy & 0xffu) << 32);
						if (EXT) xq = (const uint8_t *)(ww + (uniformW ? 0u : 4u * ((n + 3) / 4)));
					};
					/* The bases of the k-mers a lane expands lie in REGISTERS: a window of 2 W + 2 dwords of its record's packed bases, the
					 * current k-mer `sb` bits into it (0 <= sb <= 62); the next k-mer of the same record is the window two bits further on
					 * (no LDS read, and nothing to wait for, per k-mer). */
					constexpr int NWD = 2 * W + 2;
					uint32_t win[NWD], sb = 0;
#pragma unroll
					for (int i = 0; i < NWD; i++) win[i] = 0;
					auto seed_window = [&]() {       /* the window of k-mer j of the record at granule rs (a record entered in its middle, or a window run out) */
						const uint32_t *bw = (const uint32_t *)(wstage + rs + 1) + (j >> 4);
#pragma unroll
						for (int i = 0; i < NWD; i++) win[i] = bw[i];
						sb = 2u * (j & 15u);
					};
					/* a k-mer on its way to the table: canonical key, strand, home slot and probe step, weight, stream ordinal.  Two of them
					 * exist per lane -- the one being inserted and the one being made -- and the loop below alternates their roles instead of
					 * copying one into the other every step (the copies were a quarter of the loop's vector instructions) */
					struct KState { Key<W> key; bool fwd, mine; uint32_t slot, step, w; uint64_t ord; uint32_t x; };      /* x (EXT): left code | right code << 8 | left quality << 16 | right quality << 24, as the canonical strand sees them */
					KState kA, kB;
#pragma unroll
					for (int wi = 0; wi < W; wi++) { kA.key.w[wi] = 0; kB.key.w[wi] = 0; }
					kA.fwd = kB.fwd = true; kA.mine = kB.mine = false; kA.slot = kB.slot = 0; kA.step = kB.step = 1; kA.w = kB.w = 0; kA.ord = kB.ord = 0; kA.x = kB.x = 0;
					auto prepare = [&](KState &st) {      /* the k-mer at (record rs, index j): window bit offset sb */
						Key<W> kf;
						const bool up = sb >= 32u; const uint32_t r = sb & 31u;
#pragma unroll
						for (int wi = 0; wi < W; wi++) {
							const uint32_t d0 = up ? win[2 * wi + 1] : win[2 * wi], d1 = up ? win[2 * wi + 2] : win[2 * wi + 1], d2 = up ? win[2 * wi + 3] : win[2 * wi + 2];
							const uint32_t hi = (uint32_t)(((((uint64_t)d0) << 32 | d1) << r) >> 32), lo = (uint32_t)(((((uint64_t)d1) << 32 | d2) << r) >> 32);
							kf.w[wi] = ((uint64_t)hi << 32) | lo;
						}
						const uint32_t kbits = 2u * k;
						uint32_t rcIn = 0; uint64_t kfFirst = 0;
						if constexpr (EXT) {
							/* the base behind the k-mer: base k of the 32 W bases just cut out of the window (k = 32 W: the next dword of the record) */
							kfFirst = kf.w[0];
							if ((k >> 5) < (uint32_t)W) { uint64_t wsel = kf.w[0];
#pragma unroll
								for (int wi = 1; wi < W; wi++) if ((k >> 5) == (uint32_t)wi) wsel = kf.w[wi];
								rcIn = (uint32_t)(wsel >> (62u - 2u * (k & 31u))) & 3u; }
							else { const uint32_t b = j + k; rcIn = (((const uint32_t *)(wstage + rs + 1))[b >> 4] >> (30 - 2 * (b & 15u))) & 3u; }
						}
#pragma unroll
						for (int wi = 0; wi < W; wi++) {
							const uint32_t lo = 64u * wi;
							if (kbits <= lo) kf.w[wi] = 0;
							else if (kbits < lo + 64u) kf.w[wi] &= ~0ull << (lo + 64u - kbits);
						}
						const Key<W> kr = key_revcomp<W>(kf, k);
						st.fwd = key_le<W>(kf, kr);
						st.key = st.fwd ? kf : kr;
						const uint64_t h = slot_hash<W>(st.key.w);
						st.slot = (uint32_t)(h >> (64 - LOG2S));
						/* a key that does not find its home slot free (or its own) goes on in steps of an odd number taken from other bits of its
						 * hash: with steps of one the occupied slots grow into runs, and the wavefront waits for the longest probe sequence
						 * among its 64 lanes every time (one claim attempt and one LDS round trip per step) */
						st.step = ((uint32_t)(h >> (64 - 2 * LOG2S)) & (uint32_t)(S - 1)) | 1u;
						st.mine = ((uint32_t)(h >> 20) & subMask) == val;
						if constexpr (!UNI) st.w = uniformW ? hw : ww[j];
						st.ord = ord0 + j;
						if constexpr (EXT) {
							/* the two neighbours of k-mer j of the record (Extension left / right of buildWeightedKmers, src/KmerReadUtils.h:224-236):
							 * what the header says lies outside the run for its first / last k-mer; in between the left one is the first base of the
							 * k-mer made before this one (leftCarry) and the right one the base behind the k-mer in the window it was cut from; seen
							 * from the other strand they swap and are complemented */
							uint32_t lc = j > 0 ? leftCarry : (hy >> 24) & 7u, rc = (hy >> 27) & 7u;
							if (j + 1 < n) rc = rcIn;
							leftCarry = (uint32_t)(kfFirst >> 62);
							const uint32_t q2 = ((const uint16_t *)xq)[j];
							uint32_t lq = q2 & 0xffu, rq = q2 >> 8;
							if (!st.fwd) {
								const uint32_t tl = rc < 4u ? 3u - rc : rc, tr = lc < 4u ? 3u - lc : lc, tq = rq;
								lc = tl; rc = tr; rq = lq; lq = tq;
							}
							st.x = lc | (rc << 8) | (lq << 16) | (rq << 24);
						}
					};
					if (left) {
						enter_record(); seed_window();
						if (EXT && j > 0) { const uint32_t b = j - 1; leftCarry = (((const uint32_t *)(wstage + rs + 1))[b >> 4] >> (30 - 2 * (b & 15u))) & 3u; }      /* (a lane that starts inside a record) */
						prepare(kA);
					}
					uint32_t dbgSink = 0;
					/* one step: the current k-mer's claim is sent off, the lane's next k-mer is made (two bits further in the window, or the first
					 * one of the next record), then the claim is looked at and the three sums go to the slot */
					auto step_one = [&](KState &cur, KState &nxt) {
						if (left) {
							uint32_t s = cur.slot;
							const float wa = __uint_as_float(UNI ? f.uni_wbits : cur.w);
							/* (one-word keys: a compare-and-swap against "empty" returns what the slot holds whether it wins or not, so the slot is
							 * not read first) */
							unsigned long long old = EMPTY_KEY;
							const bool tableOn = !SK_DBG(dbgFlags, 1) && cur.mine;
							if (W == 1 || padded) { if (tableOn) old = atomicCAS((unsigned long long *)&tkeys[(size_t)s * W], (unsigned long long)EMPTY_KEY, (unsigned long long)cur.key.w[0]); }
							j++; left--; sb += 2;
							if (left) {
								if (j >= n) {
									/* the next record is fetched when the lane gets there (two LDS reads, waited for on the spot).  Its header and first window
									 * used to be asked for a record ahead and held in registers: eight of the 128 -- without them nothing spills and the pass
									 * is 0.45 ms faster, the exposed LDS latency notwithstanding */
									rs = rs + ((hy >> 17) & 0x7fu);
									{ const uint32_t g0 = rs < SK_CHUNK_G - 1 ? rs : SK_CHUNK_G - 1; const uint4 hh = wstage[g0]; hx = hh.x; hy = hh.y; hw = hh.w; }
									j = 0;
									seed_window();
									enter_record();
								} else if (sb > 62u) seed_window();
								prepare(nxt);
							}
							if (SK_DBG(dbgFlags, 1)) { dbgSink ^= (uint32_t)cur.key.w[0] ^ s; }
							else if (cur.mine) {
								bool placed = false;
								const uint32_t claimedBefore = claimedHere;
								if constexpr (W == 1) {
									if (old == EMPTY_KEY) { claimedHere++; placed = true; }
									else if (old == cur.key.w[0]) placed = true;
									else s = (s + cur.step) & (S - 1);
								}
								if (W > 1 && padded) {
									/* Every round: claim attempt, then ALL of this wavefront's winners publish their remaining words, then the lanes that met
									 * their own first word look at the last one.  Nobody leaves the loop alone (the exit is wave-uniform) and nobody waits
									 * inside a round: a lane may only ever poll for a winner in ANOTHER wavefront -- lanes of one wavefront that waited on each
									 * other inside a divergent loop would depend on where the compiler places the winner's store. */
									bool poll = false;
									for (int probe = 0; probe < 8 * S; probe++) {
										const bool active = !placed;
										if (probe > 0 && active && !poll) old = atomicCAS((unsigned long long *)&tkeys[(size_t)s * W], (unsigned long long)EMPTY_KEY, (unsigned long long)cur.key.w[0]);
										const bool won = active && !poll && old == EMPTY_KEY;
										if (won) {
#pragma unroll
											for (int qq = 1; qq < W - 1; qq++) tkeys[(size_t)s * W + qq] = cur.key.w[qq];
											__hip_atomic_store(&tkeys[(size_t)s * W + W - 1], cur.key.w[W - 1], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      /* the last word is the one waited for */
											claimedHere++; placed = true;
										}
										sk_wave_lds_order();
										if (active && !won) {
											bool moveOn = true;
											if (old == cur.key.w[0]) {
												const unsigned long long last = __hip_atomic_load(&tkeys[(size_t)s * W + W - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
												if (last == SK_KEY_PENDING) { poll = true; moveOn = false; }      /* a winner in another wavefront is between its two writes */
												else {
													bool eq = last == cur.key.w[W - 1];
#pragma unroll
													for (int qq = 1; qq < W - 1; qq++) eq = eq && (tkeys[(size_t)s * W + qq] == cur.key.w[qq]);
													if (eq) { placed = true; moveOn = false; }
												}
											}
											if (moveOn) { poll = false; s = (s + cur.step) & (S - 1); }
										}
										if (!__any(!placed)) break;
									}
								}
								for (int probe = 0; probe < S && !placed && !(W > 1 && padded); probe++) {
									if constexpr (W == 1) {
										const unsigned long long o2 = atomicCAS((unsigned long long *)&tkeys[s], (unsigned long long)EMPTY_KEY, (unsigned long long)cur.key.w[0]);
										if (o2 == EMPTY_KEY) { claimedHere++; placed = true; break; }
										if (o2 == cur.key.w[0]) { placed = true; break; }
									} else if constexpr (!LEAN) {
										uint32_t st = __hip_atomic_load(&tstate[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
										if (st == 0) {
											const uint32_t old2 = atomicCAS(&tstate[s], 0u, 1u);
											if (old2 == 0) {
#pragma unroll
												for (int qq = 0; qq < W; qq++) tkeys[(size_t)s * W + qq] = cur.key.w[qq];
												__hip_atomic_store(&tstate[s], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
												claimedHere++;
												placed = true; break;
											}
											st = old2;
										}
										if (st == 1) { probe--; continue; }      /* writer publishes unconditionally: re-poll the same slot */
										bool eq = true;
#pragma unroll
										for (int qq = 0; qq < W; qq++) eq = eq && (tkeys[(size_t)s * W + qq] == cur.key.w[qq]);
										if (eq) { placed = true; break; }
									}
									s = (s + cur.step) & (S - 1);
								}
								if (!placed) s_overflow[fl] = 1;      /* table full */
								else {
									atomicAdd(&tcnt[s], 1ull | ((unsigned long long)(cur.fwd ? 1 : 0) << 32));
									if constexpr (!UNI) atomicAdd(&twsum[s], (double)wa);
									const unsigned long long fp = UNI ? (((unsigned long long)cur.ord << FIRST_ORD_SHIFT) | ((unsigned long long)(cur.fwd ? 1u : 0u) << 23) | (unsigned long long)uniFirst) : first_pack(cur.ord, cur.fwd, wa);
									if (TRACK) {      /* the two smallest: whichever of (old first, this one) is larger is a candidate for second */
										const unsigned long long was = atomicMin(&tfirst[s], fp);
										const unsigned long long cand = was > fp ? was : fp;
										if (cand != NO_FIRST) atomicMin(&tsecond[s], cand);
									} else atomicMin(&tfirst[s], fp);
									if constexpr (EXT) {      /* ExtensionTracking::trackExtension (src/KmerTrackingData.h:195-201) */
										const uint32_t lc = cur.x & 0xffu, rc = (cur.x >> 8) & 0xffu, lq = (cur.x >> 16) & 0xffu, rq = cur.x >> 24;
										if (lq >= f.ext_min_q || lc > 3u) atomicAdd(&ttally[(size_t)s * 6 + (lc >> 1)], 1u << (16 * (lc & 1u)));
										if (rq >= f.ext_min_q || rc > 3u) atomicAdd(&ttally[(size_t)s * 6 + 3 + (rc >> 1)], 1u << (16 * (rc & 1u)));
										if (claimedHere != claimedBefore) tpkt[s] = sk_ext_char(lc) | (sk_ext_char(rc) << 8) | (cur.x & 0xffff0000u);      /* the packet of the sighting that claimed the slot: read only when it stays the only one */
									}
								}
							}
						}
					};
					for (uint32_t it = 0; it < Lk; it += 2) { step_one(kA, kB); step_one(kB, kA); }
					if (SK_DBG(dbgFlags, 1) && dbgSink == 0x12345u) s_overflow[fl] = 2;
					claimedHere = (uint32_t)__builtin_amdgcn_readlane((int)sk_wave_scan_u32(claimedHere), 63);
					if (lane == 0 && claimedHere) atomicAdd(&s_claimed[fl], claimedHere);
					sk_wave_lds_order();
					cur = nxt; curCount = nxtCount;
				}
				if (wantPre && lj + 1 < nl) { const uint64_t n0 = s_c0[lj + 1], n1 = s_c1[lj + 1]; if (n0 + wv < n1) { fetch(n0 + wv, pre, preCount); preList = lfirst + lj + 1; } }      /* (no chunk of this list was this wavefront's) */
			};
			/* ---- emit: this wavefront's quarter of the table -> entries of its output slabs (or the merge table / the size tracker's
			 * difference arrays), every slot cleared behind it */
			auto emit_pass = [&]() {
				/* Of a list's used slots only a few need more than a look at their count (C2: ~350 used, ~55 kept): the slots that do --
				 * kept entries; every used slot for a piece of a long list or under the size tracker -- are collected first (their indices
				 * go to this wavefront's quarter of the staging area, idle now), and the heavy part runs over them densely. */
				uint16_t *wsel = (uint16_t *)(stage + wv * SK_CHUNK_G);
				static_assert(WSLOTS * 2 <= (int)SK_CHUNK_G * 16, "the selected slots' indices live in one staging quarter");
				uint32_t nsel = 0;
				unsigned long long usedMask[WSLOTS / 64];
				const unsigned long long below = (1ull << lane) - 1;
#pragma unroll
				for (int i = 0; i < WSLOTS / 64; i++) {
					const int s = wv * WSLOTS + i * 64 + lane;
					const uint32_t count = (uint32_t)tcnt[s];
					const bool used = count != 0;      /* every claim is followed by its own count */
					const unsigned long long um = __ballot(used);
					usedMask[i] = um;
					if (!itemMode) { uniqW += (uint32_t)__builtin_popcountll(um); singleW += (uint32_t)__builtin_popcountll(__ballot(count == 1)); }      /* (item mode: a key may lie in several items' tables; sk_merge_emit_kernel counts) */
					uint32_t cls = !used ? 0u : ((count == 1 && singC != 3u) ? singC : (count < weakMin ? 0u : 1u));
					if (SK_DBG(dbgFlags, 2)) cls = 0;
					const bool need = (itemMode || TRACK) ? used : cls != 0;
					const unsigned long long nm = __ballot(need);
					if (need) wsel[nsel + (uint32_t)__builtin_popcountll(nm & below)] = (uint16_t)s;
					nsel += (uint32_t)__builtin_popcountll(nm);
				}
				sk_wave_lds_order();
#pragma unroll 1
				for (uint32_t sb0 = 0; sb0 < nsel; sb0 += 64) {
					const bool used = sb0 + (uint32_t)lane < nsel;
					const int s = used ? (int)wsel[sb0 + lane] : wv * WSLOTS;
					const unsigned long long cf = used ? tcnt[s] : 0ull;
					const uint32_t count = (uint32_t)cf;
					uint32_t cls = !used ? 0u : ((count == 1 && singC != 3u) ? singC : (count < weakMin ? 0u : 1u));
					if (SK_DBG(dbgFlags, 2)) cls = 0;
					Key<W> key;
#pragma unroll
					for (int q = 0; q < W; q++) key.w[q] = used ? tkeys[(size_t)s * W + q] : 0ull;
					const unsigned long long fst = used ? tfirst[s] : NO_FIRST;
					const double wsum = UNI ? (double)count * (double)__uint_as_float(f.uni_wbits) : (used ? twsum[s] : 0.0);
					if (TRACK && used) {
						/* the key counts as seen from the first boundary behind its first sighting on, and as a singleton until the
						 * first boundary behind its second one: index = number of boundaries <= the ordinal */
						auto behind = [&](unsigned long long ord) -> uint32_t {
							uint32_t lo = 0, hi = tv.n;
							while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (tv.bounds[mid] <= ord) lo = mid + 1; else hi = mid; }
							return lo;
						};
						const uint32_t iu = behind(fst >> 24);
						atomicAdd(&trkU[iu], 1u);
						atomicAdd(&trkS[iu], 1u);
						const unsigned long long sec = tsecond[s];
						if (sec != NO_FIRST) atomicAdd(&trkS[behind(sec >> 24)], 0xffffffffu);      /* -1 */
					}
					if (itemMode) {      /* a range of a long list: the slot goes into the merge table, entries come from there */
						cls = 0;
						/* multi-word keys are claimed through a state word that a loser polls until the winner has written the key: two
						 * lanes of one wavefront after the same empty slot would wait on each other forever, so there the lanes of a
						 * wavefront go one after the other (a handful of keys per item; other wavefronts are no problem) */
						for (int turn = 0; turn < (W == 1 ? 1 : 64); turn++) {
							if (used && (W == 1 || lane == turn) && !(__hip_atomic_load(out.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ERR_TABLE_FULL)) {
								bool claimed;
								const uint64_t ms = table_find_or_insert<W>(lg.merge, key, part_hash<W>(key.w), claimed);
								if (claimed && atomicAdd(lg.merge_used, 1ull) > (5ull << lg.merge.log2cap) / 8) atomicOr(out.err, (uint32_t)ERR_TABLE_FULL);
								if (ms == ~0ull) atomicOr(out.err, (uint32_t)ERR_TABLE_FULL);
								else {
									Slot<W> *sl = &lg.merge.slots[ms];
									atomicAdd(&sl->cntfwd, cf);
									atomicAdd(&sl->wsum, wsum);
									atomicMin(&sl->first, fst);
									if constexpr (EXT) {
										ExtSlot *es = &lg.merge.ext[ms];
#pragma unroll
										for (int j = 0; j < 12; j++) { const uint32_t v = (ttally[(size_t)s * 6 + (j >> 1)] >> (16 * (j & 1))) & 0xffffu; if (v) atomicAdd(&es->tally[j], v); }
										es->pkt = tpkt[s];
									}
								}
							}
						}
					}
					const unsigned long long mw = __ballot(cls == 1), ms = __ballot(cls == 2);
					if (mw) {
						const uint32_t cnt = (uint32_t)__builtin_popcountll(mw);
						if (wpos + cnt > wend) {           /* the slab is full: its tail becomes a hole, a new one is taken */
							for (unsigned long long e = wpos + lane; e < wend && e < out.wcap; e += 64) { if (out.wentries) out.wentries[e * (W + 1) + W] = 0; else out.wvals[e * vw] = 0; }
							unsigned long long g = 0;
							if (lane == 0) g = atomicAdd(out.wcursor, SK_OSLAB);
							wpos = ((unsigned long long)(uint32_t)__shfl((int)(g >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)g, 0, 64);
							wend = wpos + SK_OSLAB;
							if (wend > out.wcap) { if (lane == 0) atomicOr(out.err, (uint32_t)ERR_ENTRIES_FULL); outFull = true; wend = wpos; }
						}
						if (!outFull) {
							if (cls == 1) {
								const uint64_t pos = wpos + (uint32_t)__builtin_popcountll(mw & below);
								uint32_t fwdc = (uint32_t)(cf >> 32), cnt16 = count;
								if (f.has_singletons && first_forward(fst)) fwdc -= 1;
								if (cnt16 >= SK_ORDERED_FROM) { satK++; satS += count; if (cnt16 > 65535u) { cnt16 = 65535u; if (fwdc > 65534u) fwdc = 65534u; } }      /* (weight and direction of such a key are redone from its first 65 535 sightings in input order, sat_*_kernel) */
								if (fwdc > 65535u) fwdc = 65535u;
								const uint32_t wbits = __float_as_uint((float)(f.has_singletons ? wsum + first_weight_shift(fst) : wsum));
								if (out.wentries) {
									uint64_t ew[W + 1];
#pragma unroll
									for (int q = 0; q < W; q++) ew[q] = key.w[q];
									ew[W] = bb_pack_value(cnt16, wbits, fwdc);
									bb_store_entry<W>(out.wentries, pos, ew);
								} else {
#pragma unroll
									for (int q = 0; q < W; q++) out.wkeys[pos * W + q] = key.w[q];
									uint32_t *v = out.wvals + pos * vw;
									v[0] = cnt16; v[1] = wbits; v[2] = fwdc;
									if constexpr (EXT) {
#pragma unroll
										for (int j = 0; j < 12; j++) v[3 + j] = (ttally[(size_t)s * 6 + (j >> 1)] >> (16 * (j & 1))) & 0xffffu;
									}
								}
							}
							wpos += cnt; keptW += lane == 0 ? cnt : 0u;
						}
					}
					if (out.weakCount) {      /* (null in build_mode 3's own finalize: the map is bucketed by kmr_buckets.hpp, which counts for itself) */
						const uint64_t bucket = cls == 1 ? key_hash<W>(key, f.kb) & (f.nb_weak - 1) : 0;
						bucket_count_add(out.weakCount, bucket, cls == 1 && !outFull);
					}
					if (ms) {
						const uint32_t cnt = (uint32_t)__builtin_popcountll(ms);
						if (spos + cnt > send) {
							for (unsigned long long e = spos + lane; e < send && e < out.scap; e += 64) out.sweight[e] = 0;
							unsigned long long g = 0;
							if (lane == 0) g = atomicAdd(out.scursor, SK_OSLAB);
							spos = ((unsigned long long)(uint32_t)__shfl((int)(g >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)g, 0, 64);
							send = spos + SK_OSLAB;
							if (send > out.scap) { if (lane == 0) atomicOr(out.err, (uint32_t)ERR_ENTRIES_FULL); outFull = true; send = spos; }
						}
						uint64_t bucket = 0;
						if (!outFull && cls == 2) {
							const uint64_t pos = spos + (uint32_t)__builtin_popcountll(ms & below);
							bucket = key_hash<W>(key, f.kb) & (f.nb_sing - 1);
#pragma unroll
							for (int q = 0; q < W; q++) out.skeys[pos * W + q] = key.w[q];
							const float wf = (float)wsum;
							out.sweight[pos] = (uint8_t)((unsigned char)(((double)wf * 254.0)) + 1);
							if constexpr (EXT) out.spkt[pos] = tpkt[s];
						}
						bucket_count_add(out.singCount, bucket, cls == 2 && !outFull);
						if (!outFull) { spos += cnt; keptS += lane == 0 ? cnt : 0u; }
					}
				}
				sk_wave_lds_order();
#pragma unroll
				for (int i = 0; i < WSLOTS / 64; i++) {
					const int s = wv * WSLOTS + i * 64 + lane;
					if ((usedMask[i] >> lane) & 1ull) {
						tkeys[(size_t)s * W] = EMPTY_KEY; if (W > 1) tkeys[(size_t)s * W + W - 1] = SK_KEY_PENDING; tcnt[s] = 0; if constexpr (!LEAN) twsum[s] = 0.0; tfirst[s] = NO_FIRST; if constexpr (W > 1 && !LEAN) tstate[s] = 0; if (TRACK) tsecond[s] = NO_FIRST;
						if constexpr (EXT) {
#pragma unroll
							for (int q = 0; q < 6; q++) ttally[(size_t)s * 6 + q] = 0;
						}
					}
				}
			};
			auto clear_table = [&]() {
				for (int i = t; i < S; i += SKC_THREADS) { tkeys[(size_t)i * W] = EMPTY_KEY; if (W > 1) tkeys[(size_t)i * W + W - 1] = SK_KEY_PENDING; tcnt[i] = 0; if constexpr (!LEAN) twsum[i] = 0.0; tfirst[i] = NO_FIRST; if constexpr (W > 1 && !LEAN) tstate[i] = 0; if (TRACK) tsecond[i] = NO_FIRST; }
				if (EXT) for (int i = t; i < 6 * S; i += SKC_THREADS) ttally[i] = 0;
			};

			insert_pass(0, 0, true);
			lds_barrier();                                                   /* A: every insert of the list is in the table */
			const bool ovf = s_overflow[fl] != 0 || s_claimed[fl] > LIMIT;
			if (t == 0) { s_claimed[fl ^ 1u] = 0; s_overflow[fl ^ 1u] = 0; }      /* the other pair: last read before the barrier B of the list before, next raised after this list's */
			if (!ovf) {
				emit_pass();
				lds_barrier();                                               /* B: the table is empty again */
				continue;
			}
			/* ---- cold path: the list's distinct keys do not fit the table: sub-passes by further hash bits, a small stack of (bits, value) */
			clear_table();
			if (t == 0) { s_sp = 2; s_stackBits[0] = 1; s_stackVal[0] = 0; s_stackBits[1] = 1; s_stackVal[1] = 1; }
			lds_barrier();
			while (s_sp > 0) {
				const uint32_t bits = s_stackBits[s_sp - 1], val = s_stackVal[s_sp - 1];
				lds_barrier();
				if (t == 0) { s_sp--; s_claimed[fl] = 0; s_overflow[fl] = 0; }
				lds_barrier();
				insert_pass(bits, val, false);
				lds_barrier();
				if (s_overflow[fl] || s_claimed[fl] > LIMIT) {       /* split this sub-pass in two by one more hash bit */
					lds_barrier();
					clear_table();
					if (t == 0) {
						if (bits >= 20) atomicOr(out.err, (uint32_t)ERR_TABLE_FULL);
						else {
							s_stackBits[s_sp] = bits + 1; s_stackVal[s_sp] = val; s_sp++;
							s_stackBits[s_sp] = bits + 1; s_stackVal[s_sp] = val | (1u << bits); s_sp++;
						}
					}
					lds_barrier();
					continue;
				}
				emit_pass();
				lds_barrier();
			}
			/* both flag pairs are clean for the lists to come */
			if (t == 0) { s_claimed[0] = s_claimed[1] = 0; s_overflow[0] = s_overflow[1] = 0; }
			lds_barrier();
		}
	}
	/* the unused tails of this wavefront's slabs are holes */
	for (unsigned long long e = wpos + lane; e < wend && e < out.wcap; e += 64) { if (out.wentries) out.wentries[e * (W + 1) + W] = 0; else out.wvals[e * vw] = 0; }
	for (unsigned long long e = spos + lane; e < send && e < out.scap; e += 64) out.sweight[e] = 0;
	uniq = wave_sum(uniq) + uniqW; single = wave_sum(single) + singleW; satK = wave_sum(satK); satS = wave_sum(satS);
	if (lane == 0) {
		if (uniq) atomicAdd(&out.fc->unique, uniq); if (single) atomicAdd(&out.fc->singletons, single);
		if (keptW) atomicAdd(&out.fc->weak_kept, keptW); if (keptS) atomicAdd(&out.fc->sing_kept, keptS);
		if (satK) { atomicAdd(&out.fc->saturated, satK); atomicAdd(&out.fc->sat_sightings, satS); }
	}
	if (TRACK) {
		lds_barrier();
		for (uint32_t i = (uint32_t)t; i <= tv.n && i <= SK_TRACK_MAX; i += SKC_THREADS) { if (trkU[i]) atomicAdd(&tv.d_unique[i], trkU[i]); if (trkS[i]) atomicAdd(&tv.d_single[i], trkS[i]); }
	}
}


/* the merge table of the long lists -> entries, with count_kernel's rules (classify folded into scalars as in sk_count_kernel) */
template <int W>
__global__ __launch_bounds__(256)
void sk_merge_emit_kernel(Table<W> tbl, CountOut out, FinalizeParams f) {
	const uint32_t vw = tbl.ext ? 15 : 3;      /* a table with extension tallies beside its slots: entries of 15 value words */
	const uint32_t singC = f.has_singletons ? (f.min_depth > 1 ? 0u : 2u) : 3u;
	const uint32_t weakMin = ((!f.has_singletons || f.min_depth > 2) && f.min_depth != 1) ? f.min_depth : 0u;
	const uint64_t cap = 1ull << tbl.log2cap;
	unsigned long long uniq = 0, single = 0, keptW = 0, keptS = 0;
	for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x; i0 < cap; i0 += (uint64_t)gridDim.x * blockDim.x) {      /* whole wavefronts stay together (bucket_count_add) */
		const uint64_t i = i0 + threadIdx.x;
		const bool used = i < cap && slot_used<W>(tbl.slots[i]);
		uint32_t cls = 0; uint64_t bucket = 0;
		if (used) {
			const Slot<W> sl = tbl.slots[i];
			const Key<W> key = slot_key<W>(sl);
			uint32_t cnt = (uint32_t)sl.cntfwd, fwdc = (uint32_t)(sl.cntfwd >> 32);
			uniq++; if (cnt == 1) single++;
			cls = (cnt == 1 && singC != 3u) ? singC : (cnt < weakMin ? 0u : 1u);
			if (cls == 1) {
				const unsigned long long pos = atomicAdd(out.wcursor, 1ull);
				if (pos >= out.wcap) { atomicOr(out.err, (uint32_t)ERR_ENTRIES_FULL); cls = 0; }
				else {
					if (out.weakCount) bucket = key_hash<W>(key, f.kb) & (f.nb_weak - 1);
					if (f.has_singletons && first_forward(sl.first)) fwdc -= 1;
					if (cnt >= SK_ORDERED_FROM) { atomicAdd(&out.fc->saturated, 1ull); atomicAdd(&out.fc->sat_sightings, (unsigned long long)cnt); if (cnt > 65535u) { cnt = 65535u; if (fwdc > 65534u) fwdc = 65534u; } }
					if (fwdc > 65535u) fwdc = 65535u;
					const uint32_t wbits = __float_as_uint((float)(f.has_singletons ? sl.wsum + first_weight_shift(sl.first) : sl.wsum));
					if (out.wentries) {
						uint64_t ew[W + 1];
#pragma unroll
						for (int q = 0; q < W; q++) ew[q] = key.w[q];
						ew[W] = bb_pack_value(cnt, wbits, fwdc);
						bb_store_entry<W>(out.wentries, pos, ew);
					} else {
#pragma unroll
						for (int q = 0; q < W; q++) out.wkeys[pos * W + q] = key.w[q];
						uint32_t *v = out.wvals + pos * vw;
						v[0] = cnt; v[1] = wbits; v[2] = fwdc;
						if (tbl.ext) for (int j = 0; j < 12; j++) v[3 + j] = tbl.ext[i].tally[j];
					}
					keptW++;
				}
			} else if (cls == 2) {
				const unsigned long long pos = atomicAdd(out.scursor, 1ull);
				if (pos >= out.scap) { atomicOr(out.err, (uint32_t)ERR_ENTRIES_FULL); cls = 0; }
				else {
					bucket = key_hash<W>(key, f.kb) & (f.nb_sing - 1);
#pragma unroll
					for (int q = 0; q < W; q++) out.skeys[pos * W + q] = key.w[q];
					const float wf = (float)sl.wsum;
					out.sweight[pos] = (uint8_t)((unsigned char)(((double)wf * 254.0)) + 1);
					if (tbl.ext && out.spkt) out.spkt[pos] = tbl.ext[i].pkt;
					keptS++;
				}
			}
		}
		if (out.weakCount) bucket_count_add(out.weakCount, bucket, cls == 1);
		bucket_count_add(out.singCount, bucket, cls == 2);
	}
	uniq = wave_sum(uniq); single = wave_sum(single); keptW = wave_sum(keptW); keptS = wave_sum(keptS);
	if ((threadIdx.x & 63) == 0) {
		if (uniq) atomicAdd(&out.fc->unique, uniq); if (single) atomicAdd(&out.fc->singletons, single);
		if (keptW) atomicAdd(&out.fc->weak_kept, keptW); if (keptS) atomicAdd(&out.fc->sing_kept, keptS);
	}
}
/* lists of more than `threshold` chunks -> work items of `piece` chunks each (their number through *n_items; nothing is written past cap) */
#ifndef KMR_INSTANCE_TU
__global__ void sk_long_items_kernel(const uint64_t *list_start, uint64_t n_lists, uint64_t threshold, uint64_t piece, uint64_t *item_c0, uint64_t *item_c1, uint64_t cap, unsigned long long *n_items, uint32_t *item_list = nullptr) {
	for (uint64_t l = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; l < n_lists; l += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t a = list_start[l], b = list_start[l + 1];
		if (b - a <= threshold) continue;
		const uint64_t n = (b - a + piece - 1) / piece;
		const unsigned long long at = atomicAdd(n_items, (unsigned long long)n);
		for (uint64_t i = 0; i < n && at + i < cap; i++) { item_c0[at + i] = a + i * piece; item_c1[at + i] = a + (i + 1) * piece < b ? a + (i + 1) * piece : b; if (item_list) item_list[at + i] = (uint32_t)l; }
	}
}
#endif

/* ------------------------------------------------------------------ lookups as a streaming pass (f1) */
/* ReadSelector::scoreAndTrimReads asks the weak map for the count of every k-mer of every read (getValue, src/ReadSelector.h:924-931;
 * setKmerValues :1060-1090): 1.2 x 10^9 independent lookups per C2 batch, which as random probes of a 2 GB table run at the
 * chip's random-access rate (72 ms).  The same lists that make the build stream make the lookups stream: the weak map's entries
 * are grouped ONCE per finalized map by the list their minimizer selects (sk_index_*: key + count, 12 bytes per entry at k <= 32),
 * the reads are cut into super-k-mers exactly as for the build (no qualities: every k-mer without an N is asked for), and one block
 * per list puts the list's entries into an LDS table, expands the list's records and answers each k-mer from LDS, writing the
 * count to the k-mer's position (its stream ordinal).  A k-mer that is not in the map, or holds an N, keeps the zero the output was
 * cleared to. */
template <int W>
__global__ __launch_bounds__(256)
void sk_index_hist_kernel(const uint64_t *keys, uint64_t n, uint32_t m, uint32_t off, uint32_t win, uint32_t list_bits, uint32_t *elist, uint32_t *hist) {
	for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t l = sk_list_of(sk_key_minimizer<W>(keys + e * W, m, off, win), list_bits);
		elist[e] = l;
		atomicAdd(&hist[l], 1u);
	}
}
template <int W>
__global__ __launch_bounds__(256)
void sk_index_scatter_kernel(const uint64_t *keys, const uint32_t *vals, uint32_t vw, uint64_t n, const uint32_t *elist, const uint64_t *start, uint32_t *cursor,
                             uint64_t *ikeys, uint32_t *icounts) {
	for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t l = elist[e];
		const uint64_t pos = start[l] + atomicAdd(&cursor[l], 1u);
#pragma unroll
		for (int q = 0; q < W; q++) ikeys[pos * W + q] = keys[e * W + q];
		icounts[pos] = vals[e * vw] & 0xffffu;       /* TrackingData::getCount: the u16 at the head of the value */
	}
}

static const int SKL_LOG2S = 10;                       /* LDS table of a list's entries: 1024 slots, at most SKL_FILL per pass */
static const uint32_t SKL_FILL = 704;
template <int W> __host__ __device__ constexpr size_t sk_lookup_smem_bytes() { return (size_t)(1 << SKL_LOG2S) * (8 * W + 4 + 4) + (size_t)SK_STAGE_G * 16 + 256 + 64; }

template <int W>
__global__ __launch_bounds__(COUNT_THREADS, W == 1 ? 4 : 1)
void sk_lookup_kernel(PoolView pool, const uint64_t *list_start, const uint64_t *list_chunks, uint64_t n_lists, uint32_t k,
                      const uint64_t *ix_start, const uint64_t *ix_keys, const uint32_t *ix_counts, uint32_t *out, uint64_t out_n, unsigned int *work_counter,
                      const uint64_t *item_c0, const uint64_t *item_c1, const uint32_t *item_list, uint64_t n_items, uint64_t long_threshold) {
	constexpr int S = 1 << SKL_LOG2S;
	extern __shared__ __attribute__((aligned(16))) uint8_t csm[];
	uint64_t *tkeys = (uint64_t *)csm;                                 /* [S][W] */
	uint32_t *tval = (uint32_t *)(tkeys + (size_t)S * W);
	uint32_t *tstate = tval + S;                                       /* 0 empty, 1 filled (entries are put in before anybody looks) */
	uint4 *stage = (uint4 *)(tstate + S);
	uint8_t *recOf = (uint8_t *)(stage + SK_STAGE_G);
	__shared__ uint32_t s_list;
	__shared__ unsigned long long s_c0[SK_LBATCH], s_c1[SK_LBATCH];
	__shared__ uint32_t s_lid[SK_LBATCH];
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	const uint4 *poolg = (const uint4 *)pool.base;
	/* a list much longer than the others (a hot minimizer) is answered in pieces, each a work item of a second launch: lookups of
	 * different records of a list do not depend on each other */
	const bool itemMode = item_c0 != nullptr;
	const uint32_t grab = itemMode ? 1u : SK_LBATCH;
	const uint64_t n_work = itemMode ? n_items : n_lists;
	for (;;) {
		lds_barrier();
		if (t == 0) s_list = atomicAdd(work_counter, grab);
		lds_barrier();
		const uint64_t lfirst = s_list;
		if (lfirst >= n_work) break;
		const uint32_t nl = (uint32_t)(n_work - lfirst < (uint64_t)grab ? n_work - lfirst : (uint64_t)grab);
		if ((uint32_t)t < nl) {
			uint64_t a, b; uint32_t l;
			if (itemMode) { a = item_c0[lfirst + t]; b = item_c1[lfirst + t]; l = item_list[lfirst + t]; }
			else { a = list_start[lfirst + t]; b = list_start[lfirst + t + 1]; l = (uint32_t)(lfirst + t); if (long_threshold && b - a > long_threshold) b = a; }
			s_c0[t] = a; s_c1[t] = b; s_lid[t] = l;
		}
		lds_barrier();
		for (uint32_t lj = 0; lj < nl; lj++) {
			const uint64_t c0 = s_c0[lj], c1 = s_c1[lj];
			if (c0 == c1) continue;
			const uint64_t e0 = ix_start[s_lid[lj]], e1 = ix_start[s_lid[lj] + 1];
			/* a list with more entries than the table takes is answered in passes over its records, one table fill each (a key is in
			 * exactly one of them) */
			for (uint64_t eb = e0; eb < e1; eb += SKL_FILL) {
				lds_barrier();
				for (int i = t; i < S; i += COUNT_THREADS) tstate[i] = 0;
				lds_barrier();
				const uint64_t ee = eb + SKL_FILL < e1 ? eb + SKL_FILL : e1;
				for (uint64_t e = eb + (uint64_t)t; e < ee; e += COUNT_THREADS) {
					uint64_t kw[W];
#pragma unroll
					for (int q = 0; q < W; q++) kw[q] = ix_keys[e * W + q];
					uint32_t sl = (uint32_t)(slot_hash<W>(kw) >> (64 - SKL_LOG2S));
					while (atomicCAS(&tstate[sl], 0u, 1u) != 0u) sl = (sl + 1) & (S - 1);      /* keys are distinct: a taken slot is somebody else's */
#pragma unroll
					for (int q = 0; q < W; q++) tkeys[(size_t)sl * W + q] = kw[q];
					tval[sl] = ix_counts[e];
				}
				lds_barrier();
				uint4 *wstage = stage + wv * SK_CHUNK_G;
				uint8_t *wrecOf = recOf + wv * 64;
				for (uint64_t ci = c0 + wv; ci < c1; ci += SK_STAGE_CHUNKS) {
					const uint64_t d = list_chunks[ci];
					const uint32_t chunk = (uint32_t)d, curCount = (uint32_t)(d >> 32);
					uint4 cur = make_uint4(0, 0, 0, 0);
					if ((uint32_t)lane < curCount) cur = poolg[(size_t)chunk * SK_CHUNK_G + lane];
					__builtin_amdgcn_wave_barrier();      /* the lanes are done with the chunk before */
					wstage[lane] = cur;
					const uint32_t glen = (cur.y >> 17) & 0x7fu;
					unsigned long long starts = 0;
					if (__all((lane & 1) != 0 || (uint32_t)lane >= curCount || glen == 2u)) starts = 0x5555555555555555ull & (curCount >= 64u ? ~0ull : ((1ull << curCount) - 1ull));
					else for (uint32_t pos = 0; pos < curCount; ) {
						starts |= 1ull << pos;
						const uint32_t step = (uint32_t)__builtin_amdgcn_readlane((int)glen, (int)pos);
						pos += step ? step : SK_CHUNK_G;
					}
					/* the chunk's k-mers dealt evenly over the lanes, as in sk_count_kernel */
					const bool isStart = (starts >> lane) & 1ull;
					const uint32_t myN = isStart ? (cur.y >> 8) & 0xffu : 0u;
					const uint32_t incl = sk_wave_scan_u32(myN);
					const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
					const uint32_t myOff = incl - myN;
					const uint32_t Lk = (T + 63u) >> 6;
					if (myN) {
						const float inv = __builtin_amdgcn_rcpf((float)Lk);      /* as in sk_count_kernel */
						const uint32_t l0 = (uint32_t)(((float)(myOff + Lk - 1) + 0.5f) * inv), l1 = (uint32_t)(((float)(myOff + myN - 1) + 0.5f) * inv);
						for (uint32_t l = l0; l <= l1 && l < 64u; l++) wrecOf[l] = (uint8_t)lane;
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					const uint32_t k0 = (uint32_t)lane * Lk;
					uint32_t left = k0 < T ? (T - k0 < Lk ? T - k0 : Lk) : 0u;
					uint32_t rs = left ? wrecOf[lane] : 0u;
					uint32_t j = k0 - (uint32_t)__shfl((int)myOff, (int)rs, 64);
					uint32_t hx = (uint32_t)__shfl((int)cur.x, (int)rs, 64), hy = (uint32_t)__shfl((int)cur.y, (int)rs, 64);
					uint32_t n = 0; const uint32_t *bw = nullptr; uint64_t ord0 = 0;
					for (uint32_t it = 0; it < Lk; it++) {
						if (left) {
							if (it == 0 || j >= n) {
								if (it != 0) { rs += (hy >> 17) & 0x7fu; const uint4 hd = wstage[rs]; hx = hd.x; hy = hd.y; j = 0; }
								n = (hy >> 8) & 0xffu;
								bw = (const uint32_t *)(wstage + rs + 1);
								ord0 = (uint64_t)hx | ((uint64_t)(hy & 0xffu) << 32);
							}
							Key<W> kf;
							const uint32_t d0 = j >> 4, sft = 2u * (j & 15u);
#pragma unroll
							for (int wi = 0; wi < W; wi++) {
								const uint32_t a = bw[d0 + 2 * wi], b = bw[d0 + 2 * wi + 1], c = bw[d0 + 2 * wi + 2];
								const uint64_t hi = ((uint64_t)a << 32) | b;
								kf.w[wi] = sft ? (hi << sft) | ((uint64_t)c >> (32 - sft)) : hi;
							}
							const uint32_t kbits = 2u * k;
#pragma unroll
							for (int wi = 0; wi < W; wi++) {
								const uint32_t lo = 64u * wi;
								if (kbits <= lo) kf.w[wi] = 0;
								else if (kbits < lo + 64u) kf.w[wi] &= ~0ull << (lo + 64u - kbits);
							}
							const Key<W> kr = key_revcomp<W>(kf, k);
							const Key<W> key = key_le<W>(kf, kr) ? kf : kr;
							uint32_t sl = (uint32_t)(slot_hash<W>(key.w) >> (64 - SKL_LOG2S));
							uint32_t found = 0;
							for (int probe = 0; probe < S; probe++) {
								if (tstate[sl] == 0) break;
								bool eq = true;
#pragma unroll
								for (int q = 0; q < W; q++) eq = eq && tkeys[(size_t)sl * W + q] == key.w[q];
								if (eq) { found = tval[sl]; break; }
								sl = (sl + 1) & (S - 1);
							}
							const uint64_t at = ord0 + j;
							if (found && at < out_n) out[at] = found;
							j++; left--;
						}
					}
				}
			}
		}
	}
}

/* ------------------------------------------------------------------ k-mers seen more than 65 535 times */
/* TrackingData::track stops at MAX_COUNT (src/KmerTrackingData.h:427-448): the 65 535 sightings a saturated k-mer's weightedCount and
 * directionBias (:517-529) are made of are its FIRST 65 535 in input order.  The count pass adds up all sightings (it has no order);
 * for the handful of keys it clamps, this pass goes back to their lists, writes down every sighting as (key index, stream ordinal,
 * strand; weight), sorts them (kmr_sort.hip) and sums the first 65 535 of every key -- with the first one of all treated as the
 * promotion from the singleton map treats it (direction lost, weight as the singleton byte kept it, :641-658). */
/* entries of the finished weak map whose count is 65 535: their index in the map and the list their minimizer selects */
template <int W>
__global__ __launch_bounds__(256)
void sat_find_kernel(const uint64_t *keys, const uint32_t *vals, uint64_t n, uint32_t m, uint32_t off, uint32_t win, uint32_t list_bits,
                     unsigned long long *found, uint64_t cap, uint64_t *sat_entry, uint32_t *sat_list, uint32_t vw = 3) {
	for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
		if ((vals[e * vw] & 0xffffu) < SK_ORDERED_FROM) continue;
		const unsigned long long at = atomicAdd(found, 1ull);
		if (at < cap) { sat_entry[at] = e; sat_list[at] = sk_list_of(sk_key_minimizer<W>(keys + e * W, m, off, win), list_bits); }
	}
}
#ifndef KMR_INSTANCE_TU
__global__ void sat_gather_kernel(const uint64_t *list_start, const uint32_t *lists, uint64_t n, uint64_t *c0, uint64_t *c1) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { c0[i] = list_start[lists[i]]; c1[i] = list_start[lists[i] + 1]; }
}
#endif
/* work item i: chunks [item_c0[i], item_c1[i]) of a list, saturated keys [item_e0[i], item_e1[i]) (indices into sat_entry, which is
 * ordered by list).  Every sighting of one of those keys becomes a pair: key = index << 41 | ordinal << 1 | forward, value = weight. */
static const uint32_t SAT_FILL = 192;
template <int W>
__global__ __launch_bounds__(256)
void sat_collect_kernel(PoolView pool, const uint64_t *list_chunks, uint32_t k, const uint64_t *map_keys, const uint64_t *sat_entry,
                        const uint64_t *item_c0, const uint64_t *item_c1, const uint64_t *item_e0, const uint64_t *item_e1, uint64_t n_items,
                        unsigned long long *cursor, uint64_t cap, unsigned long long *out_keys, uint32_t *out_vals, unsigned int *work_counter) {
	constexpr int S = 256;
	__shared__ uint64_t tkeys[S * W];
	__shared__ uint32_t tval[S], tstate[S];
	__shared__ __attribute__((aligned(16))) uint4 stage[4 * SK_CHUNK_G];
	__shared__ uint8_t recOf[4 * 64];
	__shared__ uint32_t s_item;
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	const uint4 *poolg = (const uint4 *)pool.base;
	for (;;) {
		__syncthreads();
		if (t == 0) s_item = atomicAdd(work_counter, 1u);
		__syncthreads();
		const uint64_t it = s_item;
		if (it >= n_items) break;
		const uint64_t c0 = item_c0[it], c1 = item_c1[it], e0 = item_e0[it], e1 = item_e1[it];
		for (uint64_t eb = e0; eb < e1; eb += SAT_FILL) {
			__syncthreads();
			for (int i = t; i < S; i += 256) tstate[i] = 0;
			__syncthreads();
			const uint64_t ee = eb + SAT_FILL < e1 ? eb + SAT_FILL : e1;
			for (uint64_t e = eb + (uint64_t)t; e < ee; e += 256) {
				uint64_t kw[W];
#pragma unroll
				for (int q = 0; q < W; q++) kw[q] = map_keys[sat_entry[e] * W + q];
				uint32_t sl = (uint32_t)(slot_hash<W>(kw) >> 56);
				while (atomicCAS(&tstate[sl], 0u, 1u) != 0u) sl = (sl + 1) & (S - 1);
#pragma unroll
				for (int q = 0; q < W; q++) tkeys[(size_t)sl * W + q] = kw[q];
				tval[sl] = (uint32_t)e;
			}
			__syncthreads();
			uint4 *wstage = stage + wv * SK_CHUNK_G;
			uint8_t *wrecOf = recOf + wv * 64;
			for (uint64_t ci = c0 + wv; ci < c1; ci += 4) {
				const uint64_t d = list_chunks[ci];
				const uint32_t chunk = (uint32_t)d, curCount = (uint32_t)(d >> 32);
				uint4 cur = make_uint4(0, 0, 0, 0);
				if ((uint32_t)lane < curCount) cur = poolg[(size_t)chunk * SK_CHUNK_G + lane];
				sk_wave_lds_order();      /* the lanes are done with the chunk before */
				wstage[lane] = cur;
				const uint32_t glen = (cur.y >> 17) & 0x7fu;
				unsigned long long starts = 0;
				if (__all((lane & 1) != 0 || (uint32_t)lane >= curCount || glen == 2u)) starts = 0x5555555555555555ull & (curCount >= 64u ? ~0ull : ((1ull << curCount) - 1ull));
				else for (uint32_t pos = 0; pos < curCount; ) {
					starts |= 1ull << pos;
					const uint32_t step = (uint32_t)__builtin_amdgcn_readlane((int)glen, (int)pos);
					pos += step ? step : SK_CHUNK_G;
				}
				/* the chunk's k-mers dealt evenly over the lanes, as in sk_lookup_kernel */
				const bool isStart = (starts >> lane) & 1ull;
				const uint32_t myN = isStart ? (cur.y >> 8) & 0xffu : 0u;
				const uint32_t incl = sk_wave_scan_u32(myN);
				const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
				const uint32_t myOff = incl - myN;
				const uint32_t Lk = (T + 63u) >> 6;
				if (myN) {
					const float inv = __builtin_amdgcn_rcpf((float)Lk);      /* as in sk_count_kernel */
					const uint32_t l0 = (uint32_t)(((float)(myOff + Lk - 1) + 0.5f) * inv), l1 = (uint32_t)(((float)(myOff + myN - 1) + 0.5f) * inv);
					for (uint32_t l = l0; l <= l1 && l < 64u; l++) wrecOf[l] = (uint8_t)lane;
				}
				sk_wave_lds_order();
				const uint32_t k0 = (uint32_t)lane * Lk;
				uint32_t left = k0 < T ? (T - k0 < Lk ? T - k0 : Lk) : 0u;
				uint32_t rs = left ? wrecOf[lane] : 0u;
				uint32_t j = k0 - (uint32_t)__shfl((int)myOff, (int)rs, 64);
				uint32_t hx = (uint32_t)__shfl((int)cur.x, (int)rs, 64), hy = (uint32_t)__shfl((int)cur.y, (int)rs, 64), hw = (uint32_t)__shfl((int)cur.w, (int)rs, 64);
				uint32_t n = 0; const uint32_t *bw = nullptr, *ww = nullptr; uint64_t ord0 = 0; bool uniformW = true;
				for (uint32_t itr = 0; itr < Lk; itr++) {
					uint32_t found = 0xffffffffu, wbits = 0; bool fwd = true; uint64_t ord = 0;
					if (left) {
						if (itr == 0 || j >= n) {
							if (itr != 0) { rs += (hy >> 17) & 0x7fu; const uint4 hd = wstage[rs]; hx = hd.x; hy = hd.y; hw = hd.w; j = 0; }
							n = (hy >> 8) & 0xffu;
							bw = (const uint32_t *)(wstage + rs + 1);
							ww = bw + 4 * sk_base_granules(n, k);
							uniformW = ((hy >> 16) & 1u) != 0;
							ord0 = (uint64_t)hx | ((uint64_t)(hy & 0xffu) << 32);
						}
						Key<W> kf;
						const uint32_t d0 = j >> 4, sft = 2u * (j & 15u);
#pragma unroll
						for (int wi = 0; wi < W; wi++) {
							const uint32_t a = bw[d0 + 2 * wi], b = bw[d0 + 2 * wi + 1], c = bw[d0 + 2 * wi + 2];
							const uint64_t hi = ((uint64_t)a << 32) | b;
							kf.w[wi] = sft ? (hi << sft) | ((uint64_t)c >> (32 - sft)) : hi;
						}
						const uint32_t kbits = 2u * k;
#pragma unroll
						for (int wi = 0; wi < W; wi++) {
							const uint32_t lo = 64u * wi;
							if (kbits <= lo) kf.w[wi] = 0;
							else if (kbits < lo + 64u) kf.w[wi] &= ~0ull << (lo + 64u - kbits);
						}
						const Key<W> kr = key_revcomp<W>(kf, k);
						fwd = key_le<W>(kf, kr);
						const Key<W> key = fwd ? kf : kr;
						uint32_t sl = (uint32_t)(slot_hash<W>(key.w) >> 56);
						for (int probe = 0; probe < S; probe++) {
							if (tstate[sl] == 0) break;
							bool eq = true;
#pragma unroll
							for (int q = 0; q < W; q++) eq = eq && tkeys[(size_t)sl * W + q] == key.w[q];
							if (eq) { found = tval[sl]; break; }
							sl = (sl + 1) & (S - 1);
						}
						wbits = uniformW ? hw : ww[j];
						ord = ord0 + j;
						j++; left--;
					}
					/* the lanes that hit a saturated key take consecutive places: one device atomic per wavefront and step */
					const unsigned long long hit = __ballot(found != 0xffffffffu);
					if (hit) {
						unsigned long long base = 0;
						if (lane == 0) base = atomicAdd(cursor, (unsigned long long)__builtin_popcountll(hit));
						base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0, 64);
						if (found != 0xffffffffu) {
							const unsigned long long at = base + (unsigned long long)__builtin_popcountll(hit & ((1ull << lane) - 1));
							if (at < cap) { out_keys[at] = ((unsigned long long)found << 41) | ((unsigned long long)ord << 1) | (fwd ? 1ull : 0ull); out_vals[at] = wbits; }
						}
					}
				}
			}
		}
	}
}
/* one block per such key (index b into sat_entry): its sightings are the sorted pairs with key >> 41 == b; the first 65 535 of
 * them make weightedCount and directionBias of map entry sat_entry[b].
 * The weights are added as the reference adds them -- one after the other in input order into a float (weightedCount += weight,
 * src/KmerTrackingData.h:440: at 6 x 10^4 a float moves in steps of 2^-8, so 65 534 additions of ~1 drift by tens against the exact sum)
 * -- by one thread out of LDS tiles the block loads together; the direction count is a plain sum. */
#ifndef KMR_INSTANCE_TU
__global__ __launch_bounds__(256)
void sat_reduce_kernel(const unsigned long long *keys, const uint32_t *weights, uint64_t n_pairs, const uint64_t *sat_entry, uint64_t n_sat, uint32_t has_singletons, uint32_t *map_vals, uint32_t vw = 3) {
	constexpr int TILE = 4096;
	__shared__ float s_w[TILE]; __shared__ unsigned int s_f[256];
	const int t = threadIdx.x;
	for (uint64_t b = blockIdx.x; b < n_sat; b += gridDim.x) {
		auto lower = [&](unsigned long long v) { uint64_t lo = 0, hi = n_pairs; while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (keys[mid] < v) lo = mid + 1; else hi = mid; } return lo; };
		const uint64_t s = lower((unsigned long long)b << 41), e = lower((unsigned long long)(b + 1) << 41);
		__syncthreads();
		if (e <= s) continue;
		const uint64_t lim = e - s < 65535 ? e - s : 65535;      /* the reference stops tracking at the 65 535th sighting */
		/* sighting 1: without a singleton map it is tracked like the others; with one it comes back from there without its direction and
		 * with the weight the singleton byte kept (src/KmerTrackingData.h:641-658) */
		const float w1 = __uint_as_float(weights[s]);
		float acc = has_singletons ? (float)((double)(first_weight_bits(w1) >> 15) / 254.0) : (float)(0.0 + (double)w1);
		unsigned int f = 0;
		for (uint64_t i = s + 1 + t; i < s + lim; i += 256) f += (unsigned int)(keys[i] & 1ull);      /* sightings 2 .. 65 535 */
		s_f[t] = f;
		for (uint64_t base = s + 1; base < s + lim; base += TILE) {
			const uint64_t nt = s + lim - base < (uint64_t)TILE ? s + lim - base : (uint64_t)TILE;
			__syncthreads();
			for (uint64_t i = t; i < nt; i += 256) s_w[i] = __uint_as_float(weights[base + i]);
			__syncthreads();
			if (t == 0) for (uint64_t i = 0; i < nt; i++) acc = (float)((double)acc + (double)s_w[i]);
		}
		__syncthreads();
		for (int o = 128; o > 0; o >>= 1) { if (t < o) s_f[t] += s_f[t + o]; __syncthreads(); }
		if (t == 0) {
			unsigned int fwd = s_f[0];
			if (!has_singletons) fwd += (unsigned int)(keys[s] & 1ull);
			uint32_t *v = map_vals + sat_entry[b] * vw;
			v[1] = __float_as_uint(acc); v[2] = fwd;
		}
	}
}
#endif

/* ------------------------------------------------------------------ owner exchange of super-k-mer lists */
/* One process per GPU: every rank scatters the super-k-mers of ITS reads into all 2^list_bits lists; list l belongs to rank
 * l % world.  What a rank holds of other ranks' lists travels as it lies -- the used granules of every such chunk, plus
 * (list, granules) per chunk -- and the owner appends the records to its own chains (sk_adopt_kernel), so its lists end up as
 * dense as if it had extracted everything itself and the count pass runs unchanged.  This replaces the k-mer exchange of
 * _buildKmerSpectrumMPI (src/DistributedFunctions.h:340-458; MPI_Alltoallv, src/MPIBuffer.h:588-600) with ~4 bytes per k-mer on
 * the wire instead of 24 + kb; the owner function is the build's own (a k-mer's minimizer decides), not getDistributedThreadId. */
static const uint32_t SK_OWNER_MAX = 64;
#ifndef KMR_INSTANCE_TU
__global__ __launch_bounds__(256)
void sk_owner_count_kernel(const uint32_t *chunk_list, const uint32_t *chunk_count, uint32_t n_chunks, uint32_t world, unsigned long long *chunks, unsigned long long *granules,
                           uint64_t list_lo = 0, uint64_t list_hi = ~0ull) {      /* [list_lo, list_hi): the part of the list space this exchange step is about (kmr_sk_exchange_range) */
	__shared__ unsigned long long lc[SK_OWNER_MAX], lg[SK_OWNER_MAX];
	if (threadIdx.x < SK_OWNER_MAX) { lc[threadIdx.x] = 0; lg[threadIdx.x] = 0; }
	__syncthreads();
	for (uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; c < n_chunks; c += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t l = chunk_list[c];
		if (l == NO_CHUNK || chunk_count[c] == 0 || l < list_lo || l >= list_hi) continue;
		atomicAdd(&lc[l % world], 1ull); atomicAdd(&lg[l % world], (unsigned long long)chunk_count[c]);
	}
	__syncthreads();
	if (threadIdx.x < world) { if (lc[threadIdx.x]) { atomicAdd(&chunks[threadIdx.x], lc[threadIdx.x]); atomicAdd(&granules[threadIdx.x], lg[threadIdx.x]); } }
}
#endif
/* the chunks of other owners -> send buffers (owner after owner); the chunks leave the pool.  A block takes SK_PACK_TILE chunks: every
 * thread looks at its chunks and books them in LDS -- ONE word per owner, chunks << 40 | granules, so that a chunk's place among
 * the (list, granules) pairs and the place of its granules are booked together and the data lies in the order of the pairs -- one
 * thread per owner then books the tile's totals in the owner's device cursor (a handful of atomics per tile: per chunk they
 * were 2.6 x 10^6 atomics on `world` addresses, 31 ms for half a C2 batch), and the wavefronts copy chunk after chunk, a granule per
 * lane. */
static const int SK_PACK_TILE = 1024;
#ifndef KMR_INSTANCE_TU
__global__ __launch_bounds__(256)
void sk_pack_kernel(PoolView pool, uint32_t n_chunks, uint32_t world, uint32_t rank, const unsigned long long *granule_base, const unsigned long long *chunk_base,
                    unsigned long long *granule_cursor, unsigned long long *chunk_cursor, uint4 *out_data, uint2 *out_meta, uint64_t list_lo = 0, uint64_t list_hi = ~0ull) {
	__shared__ unsigned long long s_book[SK_OWNER_MAX], s_base[SK_OWNER_MAX];
	__shared__ unsigned long long s_at[SK_PACK_TILE];          /* per chunk of the tile: place in the tile's share of its owner (chunks << 40 | granules), ~0 = stays */
	__shared__ uint32_t s_list[SK_PACK_TILE];
	__shared__ uint8_t s_cnt[SK_PACK_TILE], s_own[SK_PACK_TILE];
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	(void)chunk_cursor;
	for (uint64_t tile = (uint64_t)blockIdx.x * SK_PACK_TILE; tile < n_chunks; tile += (uint64_t)gridDim.x * SK_PACK_TILE) {
		__syncthreads();
		if (t < (int)SK_OWNER_MAX) s_book[t] = 0;
		__syncthreads();
		for (int i = t; i < SK_PACK_TILE; i += 256) {
			const uint64_t c = tile + i;
			unsigned long long at = ~0ull; uint32_t l = NO_CHUNK, cnt = 0, o = 0;
			if (c < n_chunks) {
				l = pool.chunk_list[c]; cnt = pool.chunk_count[c];
				if (l != NO_CHUNK && cnt != 0 && l % world != rank && l >= list_lo && l < list_hi) { o = l % world; if (cnt > SK_CHUNK_G) cnt = SK_CHUNK_G; at = atomicAdd(&s_book[o], (1ull << 40) | (unsigned long long)cnt); }
			}
			s_at[i] = at; s_list[i] = l; s_cnt[i] = (uint8_t)cnt; s_own[i] = (uint8_t)o;
		}
		__syncthreads();
		if ((uint32_t)t < world) s_base[t] = s_book[t] ? atomicAdd(&granule_cursor[t], s_book[t]) : 0ull;
		__syncthreads();
		for (int i = wv; i < SK_PACK_TILE; i += 4) {
			const unsigned long long at = s_at[i];
			if (at == ~0ull) continue;
			const uint64_t c = tile + i;
			const uint32_t o = s_own[i], cnt = s_cnt[i];
			const unsigned long long both = s_base[o] + at;          /* chunk and granule fields add up separately (neither overflows its field) */
			const unsigned long long gp = both & ((1ull << 40) - 1), cp = both >> 40;
			if ((uint32_t)lane < cnt) out_data[granule_base[o] + gp + lane] = ((const uint4 *)pool.base)[c * SK_CHUNK_G + lane];
			if (lane == 0) { out_meta[chunk_base[o] + cp] = make_uint2(s_list[i], cnt); pool.chunk_list[c] = NO_CHUNK; pool.chunk_count[c] = 0; }
		}
	}
}
#endif
/* lists of other owners start afresh (their chunks are gone) */
#ifndef KMR_INSTANCE_TU
__global__ void sk_state_drop_kernel(unsigned long long *state, uint64_t n, uint32_t world, uint32_t rank, uint64_t list_lo = 0, uint64_t list_hi = ~0ull) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		if (i % world != rank && i >= list_lo && i < list_hi) state[i] = ((unsigned long long)NO_CHUNK << 32) | SK_CHUNK_G;
}
#endif
/* Are the records a rank receives all uniform with one weight (the count pass's UNI form)?  One thread per received chunk follows its
 * headers; uni[0] collects the one weight (0xffffffff: none seen yet), uni[1] is raised by a record with weights of its own or by a
 * second weight. */
static const uint32_t SK_UNI_NONE = 0xffffffffu;
#ifndef KMR_INSTANCE_TU
__global__ void sk_uniform_check_kernel(const uint4 *data, const uint64_t *start, const uint32_t *cnt, uint64_t n_chunks, uint32_t *uni) {
	for (uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; c < n_chunks; c += (uint64_t)gridDim.x * blockDim.x) {
		const uint4 *g = data + start[c];
		const uint32_t n = cnt[c];
		for (uint32_t pos = 0; pos < n; ) {
			const uint4 hd = g[pos];
			const uint32_t glen = (hd.y >> 17) & 0x7fu;
			if (!((hd.y >> 16) & 1u)) { uni[1] = 1; break; }
			/* (a plain look first: after the first record of a launch every thread finds the weight there -- a compare-and-swap per record
			 * put 2 x 10^8 atomics on ONE word, 1.5 s for an eighth of a C2 batch) */
			uint32_t was = __hip_atomic_load(&uni[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (was == SK_UNI_NONE) was = atomicCAS(&uni[0], SK_UNI_NONE, hd.w);
			if (was != SK_UNI_NONE && was != hd.w) { uni[1] = 1; break; }
			if (glen == 0) break;
			pos += glen;
		}
	}
}
#endif
#ifndef KMR_INSTANCE_TU
__global__ void sk_meta_counts_kernel(const uint2 *meta, uint64_t n, uint32_t *counts) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) counts[i] = meta[i].y;
}
#endif
/* received chunks -> this rank's own lists: one wavefront per chunk.  A chunk's records all belong to one list and lie back to
 * back, so the chunk's used granules are appended as ONE piece (one booking of the list's word; they land behind what the list's
 * open chunk holds if they fit, else at the head of a fresh chunk) and copied a granule per lane -- booking record by record put 32
 * lanes of a wavefront on the same word at once (65 ms for half a C2 batch). */
static const int SK_ADOPT_WAVES = 4, SK_ADOPT_GROUP = 32;
#ifndef KMR_INSTANCE_TU
__global__ __launch_bounds__(SK_ADOPT_WAVES * 64)
void sk_adopt_kernel(const uint4 *in_data, const uint2 *in_meta, const uint64_t *in_start, uint64_t n_in, SkParams sp, PoolView pool) {
	__shared__ SkSlab s_slab[SK_ADOPT_WAVES];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	SkSlab *slab = &s_slab[wave];
	if (lane == 0) { slab->base[0] = atomicAdd(pool.head, 64u); slab->base[1] = atomicAdd(pool.head, 64u); slab->next = 0; slab->hot_list = SK_NO_LIST; slab->hot_state = 0; }
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
	const uint64_t wavesPerGrid = (uint64_t)gridDim.x * SK_ADOPT_WAVES;
	/* SK_ADOPT_GROUP chunks per wavefront and round: their bookings are made by as many lanes at once (a booking is a round trip to
	 * a word somewhere in HBM), then the chunks are copied one after the other */
	for (uint64_t c0 = ((uint64_t)blockIdx.x * SK_ADOPT_WAVES + wave) * SK_ADOPT_GROUP; c0 < n_in; c0 += wavesPerGrid * SK_ADOPT_GROUP) {
		uint32_t myList = 0, myCnt = 0; unsigned long long myStart = 0, myAt = ~0ull;
		if (lane < SK_ADOPT_GROUP && c0 + lane < n_in) {
			const uint2 m = in_meta[c0 + lane];
			myList = m.x; myCnt = m.y < SK_CHUNK_G ? m.y : SK_CHUNK_G; myStart = in_start[c0 + lane];
			if (myCnt) myAt = sk_append(sp.state, myList, myCnt, slab, pool);
		}
		/* the group's granules are asked for together (the sources' positions are known before the bookings come back), then stored where the
		 * bookings say; the per-chunk scalars come by readlane (j is a constant after unrolling), not by LDS permutes */
		uint4 v[SK_ADOPT_GROUP];
#pragma unroll
		for (int j = 0; j < SK_ADOPT_GROUP; j++) {
			const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)myCnt, j);
			const unsigned long long st = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(myStart >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)myStart, j);
			v[j] = make_uint4(0, 0, 0, 0);
			if ((uint32_t)lane < cnt) v[j] = in_data[st + lane];
		}
#pragma unroll
		for (int j = 0; j < SK_ADOPT_GROUP; j++) {
			const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)myCnt, j);
			const unsigned long long at = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(myAt >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)myAt, j);
			if (at != ~0ull && (uint32_t)lane < cnt) ((uint4 *)pool.base)[at + lane] = v[j];
		}
		if (slab->next >= 64u) {
			__builtin_amdgcn_wave_barrier();
			if (lane == 0) { const uint32_t used = slab->next; slab->base[0] = slab->base[1]; slab->base[1] = atomicAdd(pool.head, 64u); slab->next = used >= 128u ? 64u : used - 64u; }
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
			__builtin_amdgcn_wave_barrier();
		}
	}
	__builtin_amdgcn_wave_barrier();
	const uint32_t used = slab->next < 128u ? slab->next : 128u;
	for (uint32_t idx = used + (uint32_t)lane; idx < 128u; idx += 64) { const uint32_t c = slab->base[idx >> 6] + (idx & 63u); if (c < pool.cap) { pool.chunk_list[c] = NO_CHUNK; pool.chunk_count[c] = 0; } }
	if (lane == 0 && slab->hot_list < SK_LIST_LOCKED) {      /* the open chunk of the wavefront's hot chain */
		const uint32_t hc = (uint32_t)(slab->hot_state >> 32), hf = (uint32_t)slab->hot_state;
		if (hc < pool.cap) pool.chunk_count[hc] = hf < SK_CHUNK_G ? hf : SK_CHUNK_G;
	}
}
#endif

/* ------------------------------------------------------------------ coarse lists on the wire, fine lists for the count pass */
/* In a job of `world` ranks a rank used to scatter its reads into the JOB's fine lists (2^23 of them at 8 ranks x 10 M reads: half a
 * chunk per list and rank -- slower extraction, half-filled chunks to pack, send and adopt, and no way to send a batch in pieces
 * without multiplying them).  Now the lists a rank scatters into, exchanges and adopts are COARSE: 2^shift fine lists each (shift =
 * ceil(log2 world)), as many and as full as on one GPU; the owner of a coarse list is coarse % world, i.e. all its fine lists have
 * one owner.  Before the count pass the owner splits what it holds: one wavefront per chunk, the records of a chunk grouped by the
 * fine list their minimizer selects (the low `shift` bits of the fine id vary within a chunk), ONE booking per group -- all groups'
 * bookings are made at once by their first lanes -- and every record copied to its place in the group's piece. */
static const int SK_REFINE_WAVES = 4;
#ifndef KMR_INSTANCE_TU
__global__ __launch_bounds__(SK_REFINE_WAVES * 64)
void sk_refine_kernel(PoolView pool, uint32_t n_before, uint32_t fine_bits, unsigned long long *fine_state) {
	__shared__ SkSlab s_slab[SK_REFINE_WAVES];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	SkSlab *slab = &s_slab[wave];
	if (lane == 0) { slab->base[0] = atomicAdd(pool.head, 64u); slab->base[1] = atomicAdd(pool.head, 64u); slab->next = 0; slab->hot_list = SK_NO_LIST; slab->hot_state = 0; }
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
	const uint64_t wavesPerGrid = (uint64_t)gridDim.x * SK_REFINE_WAVES;
	const uint4 *poolg = (const uint4 *)pool.base;
	for (uint64_t c = (uint64_t)blockIdx.x * SK_REFINE_WAVES + wave; c < n_before; c += wavesPerGrid) {
		const uint32_t l = pool.chunk_list[c];
		uint32_t cnt = pool.chunk_count[c];
		if (l == NO_CHUNK || cnt == 0) continue;
		if (cnt > SK_CHUNK_G) cnt = SK_CHUNK_G;
		uint4 cur = make_uint4(0, 0, 0, 0);
		if ((uint32_t)lane < cnt) cur = poolg[c * SK_CHUNK_G + lane];
		const uint32_t glen = (cur.y >> 17) & 0x7fu;
		unsigned long long starts = 0;
		if (__all((lane & 1) != 0 || (uint32_t)lane >= cnt || glen == 2u)) starts = 0x5555555555555555ull & (cnt >= 64u ? ~0ull : ((1ull << cnt) - 1ull));
		else for (uint32_t pos = 0; pos < cnt; ) { starts |= 1ull << pos; const uint32_t step = (uint32_t)__builtin_amdgcn_readlane((int)glen, (int)pos); pos += step ? step : SK_CHUNK_G; }
		const bool isRec = ((starts >> lane) & 1ull) && glen && (uint32_t)lane + glen <= cnt;
		const uint32_t fine = isRec ? sk_list_of(cur.z, fine_bits) : 0u;      /* cur.z: the record's minimizer hash */
		/* groups: lanes with the same fine list; a lane learns its offset inside the group's piece, the group's total and its first lane */
		uint32_t goff = 0, gtot = 0; int gfirst = -1;
		bool done = !isRec;
		for (int round = 0; round < 64; round++) {
			const unsigned long long pending = __ballot(!done);
			if (!pending) break;
			const int leader = __ffsll((long long)pending) - 1;
			const uint32_t lf = (uint32_t)__shfl((int)fine, leader, 64);
			const bool mine = !done && fine == lf;
			uint32_t incl = mine ? glen : 0u;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += x; }
			const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
			if (mine) { goff = incl - glen; gtot = tot; gfirst = leader; done = true; }
		}
		unsigned long long at = ~0ull;
		if (isRec && gfirst == lane) at = sk_append(fine_state, fine, gtot, slab, pool);      /* every group's first lane at once */
		const int src = gfirst < 0 ? 0 : gfirst;
		const unsigned long long gat = ((unsigned long long)(uint32_t)__shfl((int)(at >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)at, src, 64);
		if (isRec && gat != ~0ull) {
			uint4 *dst = (uint4 *)pool.base + gat + goff;
			dst[0] = cur;
			for (uint32_t g = 1; g < glen; g++) dst[g] = poolg[c * SK_CHUNK_G + lane + g];
		}
		__builtin_amdgcn_wave_barrier();
		if (lane == 0) { pool.chunk_list[c] = NO_CHUNK; pool.chunk_count[c] = 0; }
		if (slab->next >= 64u) {
			__builtin_amdgcn_wave_barrier();
			if (lane == 0) { const uint32_t used = slab->next; slab->base[0] = slab->base[1]; slab->base[1] = atomicAdd(pool.head, 64u); slab->next = used >= 128u ? 64u : used - 64u; }
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
			__builtin_amdgcn_wave_barrier();
		}
	}
	__builtin_amdgcn_wave_barrier();
	const uint32_t used = slab->next < 128u ? slab->next : 128u;
	for (uint32_t idx = used + (uint32_t)lane; idx < 128u; idx += 64) { const uint32_t cc = slab->base[idx >> 6] + (idx & 63u); if (cc < pool.cap) { pool.chunk_list[cc] = NO_CHUNK; pool.chunk_count[cc] = 0; } }
	if (lane == 0 && slab->hot_list < SK_LIST_LOCKED) {
		const uint32_t hc = (uint32_t)(slab->hot_state >> 32), hf = (uint32_t)slab->hot_state;
		if (hc < pool.cap) pool.chunk_count[hc] = hf < SK_CHUNK_G ? hf : SK_CHUNK_G;
	}
}
#endif

}  // namespace kmr
#endif
