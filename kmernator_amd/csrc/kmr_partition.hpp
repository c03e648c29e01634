/*
 * kmr_partition.hpp -- streaming build path: hash-partition the k-mer records (twice, or as often as
 * the input needs), then count every final partition inside LDS.
 *
 * Why: the open-addressed table of kmr_kernels.hpp touches HBM randomly (one 32-byte
 * sector and three device-scope atomics per k-mer occurrence) and runs at the chip's
 * scattered-atomic rate, ~3 % of the HBM roofline.  Here every HBM access is a
 * streaming one and all atomics are LDS atomics:
 *
 *   extract_kernel<LinearOp>        reads -> compacted records {key, signed weight, ordinal | packet + ordinal}
 *   partition_direct_kernel<..,1>  records (linear buffer, or wire records of the owner exchange) -> chunked lists
 *                                   by the top bits of the rotated lookup3 hash; its per-block state survives launches
 *   partition_direct_kernel<..,2>  each list -> sub-lists by the next bits, into the same pool (fresh chunks, or
 *                                   the chunks it has just read); repeated while lists are too long to count
 *   count_kernel                    each final list -> LDS hash table -> (key,count,fwd,weight,first[,tallies])
 *                                   -> kept entries + per-bucket counts
 *   scan / entry_scatter / sort     entries -> bucketed sorted maps (same layout as the table path)
 *   owner_scatter_kernel            sender side of the owner exchange: linear records -> owner segments
 *
 * It replaces the same reference functions as InsertOp (KmerSpectrum::append + track(),
 * src/KmerSpectrum.h:1578-1668, src/KmerTrackingData.h:427,517,641) and purgeMinDepth
 * (:1805-1815); the partition function is private (results do not depend on it).
 *
 * Write combining: one 1024-thread block per compute unit holds a batch of 8192 records in
 * registers, counts it per destination list in LDS, lets one thread per list plan where the
 * list's run goes, and then every thread stores its own records at (run base + rank).  A
 * list only ever sends whole groups of 4 records (64 bytes for 16-byte records) to HBM; the
 * remainder waits in a per-list LDS line for the next batch.  Lists are linked from fixed
 * 64-record chunks which a block takes from the pool 64 at a time.
 */
#ifndef KMR_PARTITION_HPP_
#define KMR_PARTITION_HPP_

#include "kmr_kernels.hpp"

namespace kmr {

#ifndef KMR_CH
#define KMR_CH 64
#endif
static const int CH = KMR_CH;                /* records per chunk */
static const uint32_t NO_CHUNK = 0xffffffffu;
enum { ERR_POOL_FULL = 8, ERR_ENTRIES_FULL = 16 };

struct PoolView {
	uint8_t *base;             /* chunk c occupies [c*CH*sizeof(Record), +CH*sizeof(Record)) */
	uint32_t *chunk_list;      /* list id of each chunk */
	uint32_t *chunk_count;     /* valid records in each chunk */
	unsigned int *head;        /* next free chunk */
	uint32_t cap;
	uint32_t *err;
};

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
	x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
	return x;
}
/* table-slot / sub-pass hash of the count pass (nothing user visible depends on it) */
template <int W> __host__ __device__ __forceinline__ uint64_t part_hash(const uint64_t *key) {
	uint64_t h = mix64(key[0]);
#pragma unroll
	for (int i = 1; i < W; i++) h = mix64(h ^ (key[i] * 0x9E3779B97F4A7C15ull));
	return h;
}
/* slot hash of the count pass: one multiply per key word (Fibonacci hashing); the table slot comes from the top bits, the
 * sub-pass selector from bits 20..39 (the low bits of a product only see the low bits of the key, and those are pad) */
template <int W> __host__ __device__ __forceinline__ uint64_t slot_hash(const uint64_t *key) {
	uint64_t h = key[0] * 0x9E3779B97F4A7C15ull;
#pragma unroll
	for (int i = 1; i < W; i++) h = (h ^ (h >> 31)) * 0xD6E8FEB86659FD93ull + key[i] * 0x9E3779B97F4A7C15ull;
	return h;
}
/* Partition order of a k-mer: the reference's bucket hash (KmerHasher, lookup3) rotated so that the bucket index
 * of the weak map, h & (NB - 1) with NB = 2^rot, becomes the most significant bits.  Lists are cut from the top
 * bits of this value, so a final list is a contiguous range of buckets (or, with more lists than buckets, a bucket
 * is a contiguous range of lists) and the entries a list produces land next to each other in the bucketed map:
 * entry_scatter_kernel then writes ~1 KB regions instead of one random 20-byte entry at a time. */
template <int W> __host__ __device__ __forceinline__ uint64_t part_order(const uint64_t *key, uint32_t kb, uint32_t rot) {
	Key<W> k;
#pragma unroll
	for (int i = 0; i < W; i++) k.w[i] = key[i];
	const uint64_t h = key_hash<W>(k, kb);
	return rot ? (h >> rot) | (h << (64 - rot)) : h;
}

/* murmur3 finaliser: a bijection of 32-bit words */
__host__ __device__ __forceinline__ uint32_t sk_fmix(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
/* order of the canonical m-mers (the minimizer is the smallest) */
__host__ __device__ __forceinline__ uint32_t sk_mmer_hash(uint32_t canon) { return sk_fmix(canon ^ 0x9e3779b9u); }
/* list of a minimizer: minima crowd near zero, so the list is cut from a second scramble of the hash, not from its top bits */
/* `list_bits` is a CODE: up to 32 it is the number of bits of a power-of-two list count (the top bits of the scramble); above 32 it is
 * the list count itself, any number (multiply-shift: floor(scramble * count / 2^32)) -- a single GPU's build sizes its lists for the
 * count pass (~1450 k-mers each, sk_list_code_for in kmr_api.hip) instead of taking the next power of two.  Exchanges between ranks
 * (owner = list % ranks, coarse and fine lists) keep powers of two. */
__host__ __device__ __forceinline__ uint32_t sk_list_of(uint32_t mh, uint32_t list_bits) {
	const uint32_t x = sk_fmix(mh * 0x2545f491u + 0x7f4a7c15u);
	return list_bits > 32u ? (uint32_t)(((uint64_t)x * list_bits) >> 32) : (list_bits ? x >> (32 - list_bits) : 0u);
}
__host__ __device__ __forceinline__ uint64_t sk_list_count(uint32_t list_bits) { return list_bits > 32u ? (uint64_t)list_bits : 1ull << list_bits; }

/* Owner of a k-mer in a job built on super-k-mer lists (build_mode 3 + exchange): the list of its canonical minimizer, modulo the
 * ranks.  OwnerFn.m == 0: the reference's getDistributedThreadId (lookup3).  sk_key_minimizer recomputes, from the packed k-mer,
 * what sk_extract_kernel found while walking the read: the smallest hash among the canonical m-mers at offsets off .. off + win - 1
 * (the window is symmetric, so both strands of a k-mer give the same value). */
struct OwnerFn { uint32_t m, off, win, list_bits; };
template <int W> __host__ __device__ __forceinline__ uint32_t sk_key_minimizer(const uint64_t *key, uint32_t m, uint32_t off, uint32_t win) {
	const uint32_t mmask = m >= 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u), mtop = 2 * (m - 1);
	uint32_t mf = 0, mr = 0, best = 0xffffffffu;
	for (uint32_t i = off; i < off + m + win - 1; i++) {
		const uint32_t c = (uint32_t)(key[(i >> 5) < (uint32_t)W ? (i >> 5) : W - 1] >> (62 - 2 * (i & 31u))) & 3u;
		mf = ((mf << 2) | c) & mmask;
		mr = (mr >> 2) | ((3u - c) << mtop);
		if (i + 1 >= off + m) { const uint32_t x = sk_mmer_hash(mf < mr ? mf : mr); best = x < best ? x : best; }
	}
	return best;
}

/* ------------------------------------------------------------------ LinearOp */
/* extract -> compacted linear records.  Each wavefront owns the region
 * [koff[r0], koff[r0+nr)) of the record buffer (koff = exclusive scan of the per-read
 * k-mer capacities) and appends its good records there, wave-compacted; tile_count
 * says how many it wrote. */
template <int W, bool EXT, bool STATS = true> struct LinearOp {
	typedef typename PoolRec<W, EXT>::type Rec;
	Rec *records;
	const uint64_t *koff;          /* [n_reads+1] */
	uint32_t *tile_count;          /* [n_tiles] */
	uint64_t first_read_idx;
	static const bool NEEDS_WEIGHT = true;
	static const bool COUNTS_STATS = STATS;      /* false on the sender side of the exchange: the owner counts what it receives */
	static const bool NEEDS_HASH = false;
	struct State { uint64_t base; uint32_t n; };
	__device__ __forceinline__ void wave_begin(State &, int) const {}
	__device__ __forceinline__ void wave_end(State &, int) const {}
	__device__ __forceinline__ void tile_begin(State &st, uint32_t *, uint64_t r0, int) const {
		/* the same for every lane: made a scalar so that the record address is (scalar base + 32-bit lane offset) */
		const uint64_t b = koff[r0];
		st.base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
		st.n = 0;
	}
	__device__ __forceinline__ void tile_end(State &st, uint64_t tile, int lane) const { if (lane == 0) tile_count[tile] = st.n; }
	/* called by all lanes under uniform control flow: compact the valid lanes behind the running count */
	__device__ __forceinline__ void emit(State &st, bool valid, const DevParams &, const Key<W> &key, uint64_t, const Occurrence &o,
	                                     uint64_t, uint32_t, unsigned &, bool &) const {
		const unsigned long long mask = __ballot(valid);
		const int lane = (int)(threadIdx.x & 63);
		if (valid) {
			const uint32_t rank = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1));
			Rec r;
#pragma unroll
			for (int i = 0; i < W; i++) r.key[i] = key.w[i];
			r.w = o.forward ? o.w : -o.w;
			rec_set<W>(r, o.pkt, o.ordinal);
			(records + (st.base + st.n))[rank] = r;
		}
		st.n += (uint32_t)__builtin_popcountll(mask);
	}
};
/* STATS == false is the exchange sender: it keeps the k-mers of every owner */
template <int W, bool EXT, bool STATS> __device__ __forceinline__ bool op_keeps_all_owners(const LinearOp<W, EXT, STATS> &) { return !STATS; }
template <int W, bool EXT, bool STATS> __device__ __forceinline__ uint32_t op_fail_code(const LinearOp<W, EXT, STATS> &) { return 0; }

/* Sender side of the owner exchange, second half (the first is extract_kernel<LinearOp<.., false>>): the linear records
 * of a read batch -> `world` contiguous owner segments, owner = getDistributedThreadId(lookup3(key), world)
 * (src/Kmer.h:2284-2295; the routing of _buildKmerSpectrumMPI, src/DistributedFunctions.h:418-438).  One wavefront per
 * tile of the linear buffer, in pieces of OSEG records: pass 1 hashes every record, keeps its owner in LDS and counts per
 * owner; one device atomic per owner reserves the piece's run in each segment; pass 2 reads the records again (L2 / MALL)
 * and stores them behind the runs in order, in the wire format.  No holes: seg_counts are exact record counts. */
static const int OSEG = 2048, OWNER_THREADS = 256, OWNER_MAX = 8;
/* REQ: the same binning for lookup requests (the distributed form of scoreAndTrimReads, src/DistributedFunctions.h:876-900):
 * the owner segment gets the key words only, the position of the k-mer in the requester's read batch stays behind in pos_out
 * at the same index (the answers come back in request order, so no request id travels) */
template <int W, bool EXT, bool REQ = false>
__global__ __launch_bounds__(OWNER_THREADS)
void owner_scatter_kernel(const typename PoolRec<W, EXT>::type *linear, const uint64_t *koff, const uint32_t *tile_count, uint64_t n_tiles, uint32_t kb, uint32_t world,
                          uint32_t *out, uint64_t seg_capacity, unsigned long long *seg_counts, unsigned int *work_counter, uint32_t *err, uint32_t *pos_out, OwnerFn of) {
	constexpr uint32_t RW = 2 * W + (EXT ? 2 : 1);      /* dwords of a wire record (KMR_RECORD_BYTES) */
	/* One block per tile of the linear buffer, in pieces of OSEG records that are read from HBM once and held in
	 * registers: every thread hashes its records and takes a rank per owner from an LDS counter, one thread per owner
	 * reserves the piece's run in that owner's segment with ONE device atomic, and every thread stores its records at
	 * (run base + rank): the stores of a piece fill one contiguous run per owner. */
	typedef typename PoolRec<W, EXT>::type Rec;
	__shared__ uint32_t s_cnt[OWNER_MAX];
	__shared__ unsigned long long s_base[OWNER_MAX];
	__shared__ uint32_t s_tile;
	const int t = threadIdx.x;
	constexpr int PER = OSEG / OWNER_THREADS;
	for (;;) {
		__syncthreads();
		if (t == 0) s_tile = atomicAdd(work_counter, 1u);
		__syncthreads();
		const uint32_t tile = s_tile;
		if (tile >= n_tiles) break;
		const uint64_t start = koff[(uint64_t)tile * 64];
		const uint32_t n = tile_count[tile];
		for (uint32_t s0 = 0; s0 < n; s0 += OSEG) {
			const uint32_t m = n - s0 < (uint32_t)OSEG ? n - s0 : (uint32_t)OSEG;
			const Rec *src = linear + start + s0;
			if (t < OWNER_MAX) s_cnt[t] = 0;
			__syncthreads();
			Rec r[PER];
			uint32_t ow[PER], rank[PER];
#pragma unroll
			for (int u = 0; u < PER; u++) { const uint32_t i = (uint32_t)u * OWNER_THREADS + t; r[u] = src[i < m ? i : m - 1]; }
#pragma unroll
			for (int u = 0; u < PER; u++) {
				const uint32_t i = (uint32_t)u * OWNER_THREADS + t;
				ow[u] = 0xffu; rank[u] = 0;
				if (i < m) {
					Key<W> key;
#pragma unroll
					for (int j = 0; j < W; j++) key.w[j] = r[u].key[j];
					ow[u] = of.m ? sk_list_of(sk_key_minimizer<W>(key.w, of.m, of.off, of.win), of.list_bits) % world : distributed_thread_id(key_hash<W>(key, kb), world);
					rank[u] = atomicAdd(&s_cnt[ow[u]], 1u);
				}
			}
			__syncthreads();
			if ((uint32_t)t < world) {
				const uint32_t c = s_cnt[t];
				unsigned long long b = c ? atomicAdd(&seg_counts[t], (unsigned long long)c) : 0ull;
				if (b + c > seg_capacity) { atomicOr(err, (uint32_t)ERR_SEGMENT_OVERFLOW); b = ~0ull; }
				s_base[t] = b;
			}
			__syncthreads();
#pragma unroll
			for (int u = 0; u < PER; u++) {
				const uint32_t i = (uint32_t)u * OWNER_THREADS + t;
				if (i < m) {
					const unsigned long long b = s_base[ow[u]];
					if (b != ~0ull) {
						const uint64_t at = (uint64_t)ow[u] * seg_capacity + b + rank[u];
						if (REQ) {
							uint64_t *dst = (uint64_t *)out + at * W;
#pragma unroll
							for (int j = 0; j < W; j++) dst[j] = r[u].key[j];
							pos_out[at] = (uint32_t)rec_ordinal<W>(r[u]);
						} else {
							uint32_t *dst = out + at * RW;
#pragma unroll
							for (int j = 0; j < W; j++) { dst[2 * j] = (uint32_t)r[u].key[j]; dst[2 * j + 1] = (uint32_t)(r[u].key[j] >> 32); }
							dst[2 * W] = __float_as_uint(r[u].w);
							if (EXT) dst[2 * W + 1] = r[u].pkt;
						}
					}
				}
			}
		}
	}
}

/* measurement aid: point every tile at the records of one of the first `distinct` tiles (KMR_DEBUG_SAME_TILE = distinct;
 * equal-length reads only: a tile's region is tile_records long) */
#ifndef KMR_INSTANCE_TU
__global__ void same_tile_kernel(uint64_t *koff, uint64_t n_tiles, uint64_t distinct, uint64_t tile_records) {
	for (uint64_t t = distinct + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n_tiles; t += (uint64_t)gridDim.x * blockDim.x) koff[t * 64] = (t % distinct) * tile_records;
}
#endif
/* k-mer capacity of every work unit (a read, or a segment of a long read) */
#ifndef KMR_INSTANCE_TU
__global__ void kmer_capacity_kernel(ReadsView rv, uint32_t k, uint32_t *cap) {
	const uint64_t n = rv.u_start ? rv.n_units : rv.n_reads;
	for (uint64_t u = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; u < n; u += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t r = rv.u_start ? rv.u_read[u] : u;
		const uint64_t L = rv.u_start ? rv.u_end[u] - rv.u_start[u] : rv.offsets[u + 1] - rv.offsets[u];
		cap[u] = (rv.discarded && rv.discarded[r]) ? 0u : (L >= k ? (uint32_t)(L - k + 1) : 0u);
	}
}
#endif

/* Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every global load and store the
 * wavefront has in flight (s_waitcnt vmcnt(0)), which would expose the HBM latency of the prefetched records and
 * of the fire-and-forget stores at every barrier.  The kernels below never read, inside a launch, global memory
 * that another thread of the block wrote, so LDS ordering is all their barriers need. */
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* ------------------------------------------------------------------ partition */
template <int W> struct PartSource {
	/* LEVEL 1: extents of a linear record buffer */
	const void *linear;            /* PoolRec<W, EXT>::type records, or wire records (packed_words) */
	const uint64_t *ext_start;     /* per extent: first record (NULL: uniform extents of ext_len) */
	const uint32_t *ext_count;     /* per extent: valid records   (NULL with uniform extents)     */
	uint64_t n_ext, ext_len, total;
	uint32_t ext_stride;           /* ext_start index = extent * ext_stride (64 reads per tile)   */
	unsigned long long *valid_counter;  /* optional: += records with weight != 0 (exchange input)    */
	uint32_t kb, rot;              /* key bytes and log2(buckets of the weak map): see part_order()  */
	/* LEVEL 2: work items over the chunk CSR of the level-1 pool */
	PoolView src;
	const uint64_t *list_chunks;   /* (records << 32 | chunk id), grouped by list */
	const uint64_t *item_begin;    /* per work item: range in list_chunks ... */
	const uint64_t *item_end;
	const uint32_t *item_list;     /* ... and the level-1 list it belongs to */
	uint64_t n_items;
	/* LEVEL 1: per-block state (open chunk, fill and write-combining line of every list) kept in device memory between
	 * launches, so that a build fed in many sub-batches does not end every launch with ~lists half-empty chunks per
	 * block; state_final: flush instead of saving (the last launch, with no input) */
	uint8_t *state;
	uint32_t state_final;
	/* LEVEL 1 input in the exchange wire format (KMR_RECORD_BYTES: 2W key dwords, weight, [extension packet]) instead of
	 * Record<W>: dwords per record, 0 = Record<W>.  Records without a packet get their arrival ordinal. */
	uint32_t packed_words;
	uint64_t ordinal_base;
	/* LEVEL 2: the output pool is the input pool.  A block hands the chunks of a batch it has read back to itself as
	 * free chunks for the lists it writes (it reads ~128 chunks per batch and fills ~128), so level 2 needs fresh
	 * chunks only for the first batch of an item and for partly filled ones: one pool of ~1.15 x the records
	 * instead of two pools. */
	uint32_t recycle;
};

template <int W, bool EXT, int G> __host__ __device__ inline size_t partition_state_bytes(int bits) {
	return ((((size_t)3 << bits) * 4 + 15) & ~(size_t)15) + ((size_t)G << bits) * sizeof(typename PoolRec<W, EXT>::type);
}
/* empty state: no open chunk, nothing waiting */
#ifndef KMR_INSTANCE_TU
__global__ void partition_state_init_kernel(uint8_t *state, size_t stride, int bits, uint32_t n_blocks) {
	const uint32_t P = 1u << bits;
	for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
		uint32_t *gs = (uint32_t *)(state + (size_t)b * stride);
		for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) { gs[p] = NO_CHUNK; gs[P + p] = 0; gs[2 * P + p] = 0; }
	}
}
#endif

/* ------------------------------------------------------------------ partition kernel */
/* Per batch (THREADS * RPT records held in registers, RPT per thread):
 *   1. hist[list]++ for every record (LDS atomics);
 *   2. one thread per list: n = hist[list] new records + the sn records waiting in the list's write-combining
 *      line; the largest multiple of G among them leaves the block now.  plan_run books their positions in the
 *      list's open chunk and in freshly opened chunks, and the planning thread itself copies the waiting records
 *      out of the LDS line;
 *   3. every thread takes a rank per record from an LDS cursor (hist again) and stores the record straight from
 *      its registers to chunk position (run base + rank), or into the LDS line if it belongs to the remainder.
 * Nothing is sorted or staged besides the G-record lines, so LDS holds ~8 words + one line per list and the
 * block is 1024 threads wide: the number of (block, list) write streams, which is what HBM efficiency of this
 * scatter depends on, stays at CUs * lists.  Measured on MI355X at 1024 lists and 16-byte records: 2.0 ms per
 * 2e8 records against 3.1 ms for an LDS-sorted 4096-record batch with two blocks per CU (tools/part_bench.hip). */
template <int W, bool EXT, int THREADS, int RPT, int G>
__host__ __device__ inline size_t partition_direct_smem_bytes(int bits) {
	return ((size_t)1 << bits) * 4 * (G ? 8 : 6)                               /* hist, cur, cnt, pos0, c0, xoff (, stage_n, stage_plan) */
	       + ((size_t)THREADS * RPT / CH + ((size_t)1 << bits) + 8) * 4        /* chunks opened by one batch beyond the first */
	       + ((size_t)G << bits) * sizeof(typename PoolRec<W, EXT>::type)       /* write-combining line of every list */
	       + 64;
}

template <int W, bool EXT, int LEVEL, int THREADS, int RPT, int G>
__global__ __launch_bounds__(THREADS)
void partition_direct_kernel(PartSource<W> S, PoolView out, unsigned int *work_counter, const int LOG2P, const int SHIFT) {
	const int P = 1 << LOG2P;
	constexpr int BATCH = THREADS * RPT;
	typedef typename PoolRec<W, EXT>::type Rec;
	extern __shared__ __attribute__((aligned(16))) uint8_t psm[];
	uint32_t *hist = (uint32_t *)psm;
	uint32_t *cur = hist + P;         /* open chunk of each list (NO_CHUNK if none)          */
	uint32_t *cnt = cur + P;          /* records already in it                                */
	uint32_t *pos0 = cnt + P;         /* per batch: position of the run's first record        */
	uint32_t *c0 = pos0 + P;          /* per batch: chunk that position falls into            */
	uint32_t *xoff = c0 + P;          /* per batch: index of the run's further chunks in extra[] */
	uint32_t *stage_n = xoff + P;                    /* G: records waiting in the list's write-combining line  */
	uint32_t *stage_plan = stage_n + (G ? P : 0);    /* G: per batch, (records that go out << 8) | waiting before */
	uint32_t *extra = stage_plan + (G ? P : 0);
	/* 16-byte aligned behind the words above; the alignment is done by pointer arithmetic, not through an integer, so that the
	 * compiler keeps seeing an LDS pointer (an inttoptr makes it a generic one and every line access a flat_* instruction) */
	uint8_t *stage_raw = (uint8_t *)(extra + BATCH / CH + P + 8);
	stage_raw += (16u - ((uint32_t)(uintptr_t)stage_raw & 15u)) & 15u;
	Rec *stage = (Rec *)stage_raw;
	__shared__ uint32_t s_item;
	__shared__ uint32_t s_nextra;
	constexpr uint32_t SLAB = 64, RING = 128;
	__shared__ uint32_t s_ring[RING];
	__shared__ uint32_t s_alloc, s_filled;
	constexpr int FREE_CAP = LEVEL == 2 ? 2048 : 1;
	__shared__ uint32_t s_free[FREE_CAP];      /* recycled input chunks (LEVEL 2, S.recycle) */
	__shared__ int s_free_n;
	const int t = threadIdx.x;

	uint32_t *gstate = nullptr; Rec *glines = nullptr;
	if (LEVEL == 1 && G && S.state) {
		gstate = (uint32_t *)(S.state + (size_t)blockIdx.x * partition_state_bytes<W, EXT, G>(LOG2P));
		glines = (Rec *)((uint8_t *)gstate + ((((size_t)3 << LOG2P) * 4 + 15) & ~(size_t)15));
	}
	if (gstate) {
		for (int p = t; p < P; p += THREADS) { cur[p] = gstate[p]; cnt[p] = gstate[P + p]; hist[p] = 0; stage_n[p] = gstate[2 * P + p]; }
		for (int i = t; i < P * G; i += THREADS) stage[i] = glines[i];
	} else {
		for (int p = t; p < P; p += THREADS) { cur[p] = NO_CHUNK; cnt[p] = 0; hist[p] = 0; if (G) stage_n[p] = 0; }
	}
	if (t == 0) { s_alloc = 0; s_filled = 0; s_nextra = 0; s_free_n = 0; }
	lds_barrier();

	auto top_up = [&]() {
		const uint32_t want = 2 * (BATCH / CH) + SLAB;
		if (s_alloc > s_filled) s_alloc = s_filled;
		const uint32_t have = s_filled - s_alloc;
		if (have < want) {
			const uint32_t nslab = (want - have + SLAB - 1) / SLAB;
			const uint32_t base = atomicAdd(out.head, nslab * SLAB);
			for (uint32_t i = 0; i < nslab; i++) s_ring[((s_filled / SLAB) + i) % RING] = base + i * SLAB;
			s_filled += nslab * SLAB;
		}
	};
	auto alloc_chunk = [&](uint32_t lid) -> uint32_t {
		if (LEVEL == 2 && S.recycle) {          /* pops race only with pops: pushes happen in another phase of the batch */
			const int old = atomicSub(&s_free_n, 1);
			if (old > 0) { const uint32_t c = s_free[old - 1]; out.chunk_list[c] = lid; return c; }
			atomicAdd(&s_free_n, 1);
		}
		const uint32_t idx = atomicAdd(&s_alloc, 1u);
		const uint32_t c = idx < s_filled ? s_ring[(idx / SLAB) % RING] + (idx % SLAB) : atomicAdd(out.head, 1u);
		if (c >= out.cap) { atomicOr(out.err, (uint32_t)ERR_POOL_FULL); return NO_CHUNK; }
		out.chunk_list[c] = lid;
		return c;
	};
	auto chunk_ptr = [&](uint32_t c) -> Rec * { return (Rec *)(out.base + (size_t)c * CH * sizeof(Rec)); };
	auto plan_run = [&](int p, uint32_t lid, uint32_t n) {
		uint32_t c = cur[p], filled = cnt[p];
		if (c == NO_CHUNK) { c = alloc_chunk(lid); filled = 0; }
		pos0[p] = filled; c0[p] = c;
		const uint32_t endpos = filled + n;
		const uint32_t nextra = endpos > (uint32_t)CH ? (endpos - 1) / CH : 0;
		uint32_t last = c;
		if (nextra) {
			const uint32_t xo = atomicAdd(&s_nextra, nextra);
			xoff[p] = xo;
			if (c != NO_CHUNK) out.chunk_count[c] = CH;
			for (uint32_t q = 0; q < nextra; q++) {
				last = alloc_chunk(lid);
				extra[xo + q] = last;
				if (q + 1 < nextra && last != NO_CHUNK) out.chunk_count[last] = CH;
			}
		}
		uint32_t rem = endpos - nextra * CH;
		if (rem == (uint32_t)CH) { if (last != NO_CHUNK) out.chunk_count[last] = CH; last = NO_CHUNK; rem = 0; }
		cur[p] = last; cnt[p] = rem;
	};
	auto flush_all = [&](uint32_t lid_base) {
		if (G) {
			/* the waiting records go out as a last (short) run */
			for (int p = t; p < P; p += THREADS) {
				const uint32_t sn = stage_n[p];
				if (sn) {
					plan_run(p, lid_base + p, sn);
					const uint32_t c = c0[p];
					if (c != NO_CHUNK) for (uint32_t j = 0; j < sn; j++) chunk_ptr(c)[pos0[p] + j] = stage[(size_t)p * G + j];
					stage_n[p] = 0;
				}
			}
			if (t == 0) s_nextra = 0;
		}
		for (int p = t; p < P; p += THREADS) {
			const uint32_t c = cur[p];
			if (c != NO_CHUNK) out.chunk_count[c] = cnt[p];
			cur[p] = NO_CHUNK; cnt[p] = 0;
		}
	};
	/* hist[] is all zero on entry and on exit */
	auto scatter_batch = [&](Rec (&r)[RPT], uint32_t (&pid)[RPT], uint32_t lid_base) {
		if (t == 0) top_up();
#pragma unroll
		for (int i = 0; i < RPT; i++) if (pid[i] != NO_CHUNK) atomicAdd(&hist[pid[i]], 1u);
		lds_barrier();
		for (int p = t; p < P; p += THREADS) {
			const uint32_t n = hist[p];
			if (G) {
				/* only whole groups of G records (aligned 16*G bytes: chunk positions stay multiples of G) leave
				 * the block; the remainder waits in the list's LDS line for the next batch */
				if (n) {
					const uint32_t sn = stage_n[p], m = sn + n, nout = m / G * G;
					stage_plan[p] = nout << 8 | sn;
					stage_n[p] = m - nout;
					if (nout) {
						plan_run(p, lid_base + p, nout);
						const uint32_t c = c0[p];
						if (c != NO_CHUNK) for (uint32_t j = 0; j < sn; j++) chunk_ptr(c)[pos0[p] + j] = stage[(size_t)p * G + j];
					}
					hist[p] = 0;
				}
			} else if (n) { plan_run(p, lid_base + p, n); hist[p] = 0; }
		}
		lds_barrier();
#pragma unroll
		for (int i = 0; i < RPT; i++) if (pid[i] != NO_CHUNK) {
			const uint32_t p = pid[i];
			uint32_t rank = atomicAdd(&hist[p], 1u);
			if (G) {
				const uint32_t sp = stage_plan[p], nout = sp >> 8;
				rank += sp & 0xff;
				if (rank >= nout) { stage[(size_t)p * G + (rank - nout)] = r[i]; continue; }
			}
			const uint32_t pos = pos0[p] + rank;
			const uint32_t q = pos / CH;
			const uint32_t c = q == 0 ? c0[p] : extra[xoff[p] + q - 1];
			if (c != NO_CHUNK) chunk_ptr(c)[pos % CH] = r[i];
		}
		lds_barrier();
		for (int p = t; p < P; p += THREADS) hist[p] = 0;
		if (t == 0) s_nextra = 0;
		lds_barrier();
	};

	Rec r[RPT];
	uint32_t pid[RPT];
	auto pid_of = [&](const Rec &x) -> uint32_t { return LOG2P ? (uint32_t)(part_order<W>(x.key, S.kb, S.rot) >> (64 - SHIFT - LOG2P)) & (P - 1) : 0u; };

	if (LEVEL == 1) {
		/* extents are handed out dynamically (EBATCH at a time) so ragged tiles balance.  Thread 0 cuts them into
		 * pieces of at most BATCH records and packs consecutive pieces into groups of at most BATCH records: one
		 * group = one batch (a tile of short reads is about half a batch). */
		constexpr uint32_t EBATCH = 8, MAXB = 64;
		__shared__ unsigned long long s_bstart[MAXB];
		__shared__ uint32_t s_bcount[MAXB], s_bslot[MAXB], s_gfirst[MAXB + 1];
		__shared__ uint32_t s_ng;
		__shared__ unsigned long long s_ecur, s_eoff;
		__shared__ unsigned long long s_estart[EBATCH], s_ecount[EBATCH];      /* the grabbed extents, fetched by EBATCH threads at once */
		unsigned long long nvalid = 0;
		auto load1 = [&](uint32_t g, auto &rr, auto &pp) {
#pragma unroll
			for (int i = 0; i < RPT; i++) pp[i] = NO_CHUNK;
			const uint32_t b1 = s_gfirst[g + 1];
			for (uint32_t bj = s_gfirst[g]; bj < b1; bj++) {
				const uint64_t start = s_bstart[bj]; const uint32_t n = s_bcount[bj], first = s_bslot[bj];
#pragma unroll
				for (int i = 0; i < RPT; i++) {
					const uint32_t slot = (uint32_t)i * THREADS + t;
					if (slot >= first && slot < first + n) {
						const uint64_t ri = start + (slot - first);
						if (S.packed_words) {
							const uint32_t *p32 = (const uint32_t *)S.linear + ri * S.packed_words;
#pragma unroll
							for (int j = 0; j < W; j++) rr[i].key[j] = (uint64_t)p32[2 * j] | ((uint64_t)p32[2 * j + 1] << 32);
							rr[i].w = __uint_as_float(p32[2 * W]);
							rec_set<W>(rr[i], S.packed_words > 2u * W + 1u ? p32[2 * W + 1] : 0u, S.ordinal_base + ri);      /* arrival ordinal */
						} else rr[i] = ((const Rec *)S.linear)[ri];
						if (rr[i].w != 0.0f) { pp[i] = pid_of(rr[i]); nvalid++; }
					}
				}
			}
		};
		for (;;) {
			if (t == 0) s_item = atomicAdd(work_counter, EBATCH);
			lds_barrier();
			const uint64_t efirst = s_item;
			if (efirst >= S.n_ext) break;
			if (t == 0) { s_ecur = efirst; s_eoff = 0; }
			if ((uint32_t)t < EBATCH && efirst + t < S.n_ext) {
				const uint64_t e = efirst + t;
				if (S.ext_start) { s_estart[t] = S.ext_start[e * S.ext_stride]; s_ecount[t] = S.ext_count[e]; }
				else { const uint64_t st = e * S.ext_len; s_estart[t] = st; s_ecount[t] = S.total - st < S.ext_len ? S.total - st : S.ext_len; }
			}
			for (;;) {
				lds_barrier();
				if (t == 0) {
					uint32_t nb = 0, ng = 0, filled = BATCH;       /* filled == BATCH: no group open */
					uint64_t e = s_ecur, off = s_eoff;
					while (e < efirst + EBATCH && e < S.n_ext && nb < MAXB) {
						const uint64_t start = s_estart[e - efirst], n = s_ecount[e - efirst];
						while (off < n && nb < MAXB) {
							const uint32_t c = (uint32_t)(n - off < (uint64_t)BATCH ? n - off : (uint64_t)BATCH);
							if (filled + c > (uint32_t)BATCH) { s_gfirst[ng++] = nb; filled = 0; }
							s_bstart[nb] = start + off; s_bcount[nb] = c; s_bslot[nb] = filled;
							filled += c; nb++; off += c;
						}
						if (off >= n) { e++; off = 0; }
					}
					s_gfirst[ng] = nb;
					s_ecur = e; s_eoff = off; s_ng = ng;
				}
				lds_barrier();
				const uint32_t ng = s_ng;
				if (ng == 0) break;
				for (uint32_t g = 0; g < ng; g++) { load1(g, r, pid); scatter_batch(r, pid, 0); }
			}
			lds_barrier();
		}
		lds_barrier();
		if (gstate && !S.state_final) {
			for (int p = t; p < P; p += THREADS) { gstate[p] = cur[p]; gstate[P + p] = cnt[p]; gstate[2 * P + p] = stage_n[p]; }
			for (int i = t; i < P * G; i += THREADS) glines[i] = stage[i];
		} else {
			flush_all(0);
			if (gstate) for (int p = t; p < P; p += THREADS) { gstate[p] = NO_CHUNK; gstate[P + p] = 0; gstate[2 * P + p] = 0; }
		}
		if (S.valid_counter) { nvalid = wave_sum(nvalid); if ((t & 63) == 0 && nvalid) atomicAdd(S.valid_counter, nvalid); }
	} else {
		uint32_t cid[RPT];       /* chunks this wavefront read in the current batch */
		uint64_t dsc[RPT];       /* their descriptors, requested one batch ahead (a descriptor load in front of every
		                            record load would put two HBM latencies in series) */
		auto load_desc = [&](uint64_t cb, uint64_t cb1) {
#pragma unroll
			for (int i = 0; i < RPT; i++) { const uint64_t ci = cb + (uint64_t)i * (THREADS / CH) + (t >> 6); dsc[i] = ci < cb1 ? S.list_chunks[ci] : ~0ull; }
		};
		auto load2 = [&](uint64_t cb, uint64_t cb1, auto &rr, auto &pp) {
			/* a batch = BATCH/CH chunks; wave w reads chunk (i * waves + w), lane = record */
#pragma unroll
			for (int i = 0; i < RPT; i++) {
				const uint64_t ci = cb + (uint64_t)i * (THREADS / CH) + (t >> 6);
				pp[i] = NO_CHUNK; cid[i] = NO_CHUNK;
				if (ci < cb1) {
					const uint64_t d = dsc[i];
					const uint32_t c = (uint32_t)d;
					cid[i] = c;
					if ((uint32_t)(t & 63) < (uint32_t)(d >> 32)) {
						rr[i] = ((const Rec *)(S.src.base + (size_t)c * CH * sizeof(Rec)))[t & 63];
						pp[i] = pid_of(rr[i]);
					}
				}
			}
		};
		for (;;) {
			if (t == 0) s_item = atomicAdd(work_counter, 1u);
			lds_barrier();
			const uint64_t it = s_item;
			lds_barrier();
			if (it >= S.n_items) break;
			const uint64_t cb0 = S.item_begin[it], cb1 = S.item_end[it];
			const uint32_t lid_base = S.item_list[it] << LOG2P;
			constexpr uint64_t STEP = BATCH / CH;
			load_desc(cb0, cb1);
			for (uint64_t cb = cb0; cb < cb1; cb += STEP) {
				load2(cb, cb1, r, pid);
				load_desc(cb + STEP, cb1);
				scatter_batch(r, pid, lid_base);
				/* every record of the batch has left its registers: its chunks are free (scatter_batch ends with a
				 * barrier, and the next one has a barrier between this push and the first pop) */
				if (S.recycle && (t & 63) == 0) {
#pragma unroll
					for (int i = 0; i < RPT; i++) if (cid[i] != NO_CHUNK) {
						const int pos = atomicAdd(&s_free_n, 1);
						if (pos < FREE_CAP) s_free[pos] = cid[i];
						else { atomicSub(&s_free_n, 1); out.chunk_list[cid[i]] = NO_CHUNK; out.chunk_count[cid[i]] = 0; }      /* belongs to no list of the new level */
					}
				}
			}
			lds_barrier();
			flush_all(lid_base);
			lds_barrier();
		}
	}
	lds_barrier();
	if (LEVEL == 2 && S.recycle)       /* consumed chunks nobody took: not part of any list */
		for (int i = t; i < s_free_n && i < FREE_CAP; i += THREADS) { out.chunk_list[s_free[i]] = NO_CHUNK; out.chunk_count[s_free[i]] = 0; }
	for (uint32_t idx = s_alloc + t; idx < s_filled; idx += THREADS) {
		const uint32_t c = s_ring[(idx / SLAB) % RING] + (idx % SLAB);
		if (c < out.cap) { out.chunk_list[c] = NO_CHUNK; out.chunk_count[c] = 0; }
	}
}

/* chunk CSR: chunks grouped by list.  With few lists (level 1: <= 1024) millions of chunks would hammer the same
 * few counters, so a block first ranks its chunks per list in LDS and touches the device counters once per
 * (block, list). */
static const int CSR_LDS_LISTS = 4096;
static const int CSR_THREADS = 256, CSR_ITEMS = 16;       /* chunks per thread */
#ifndef KMR_INSTANCE_TU
__global__ __launch_bounds__(CSR_THREADS)
void chunk_hist_kernel(const uint32_t *chunk_list, uint32_t n_chunks, uint32_t *list_nchunks, uint32_t nl) {
	__shared__ uint32_t lh[CSR_LDS_LISTS];
	const bool priv = nl <= (uint32_t)CSR_LDS_LISTS;
	if (priv) { for (uint32_t i = threadIdx.x; i < nl; i += CSR_THREADS) lh[i] = 0; __syncthreads(); }
	const uint64_t base = (uint64_t)blockIdx.x * CSR_THREADS * CSR_ITEMS;
	for (int j = 0; j < CSR_ITEMS; j++) {
		const uint64_t c = base + (uint64_t)j * CSR_THREADS + threadIdx.x;
		if (c >= n_chunks) break;
		const uint32_t l = chunk_list[c];
		if (l == NO_CHUNK) continue;
		if (priv) atomicAdd(&lh[l], 1u); else atomicAdd(&list_nchunks[l], 1u);
	}
	if (priv) { __syncthreads(); for (uint32_t i = threadIdx.x; i < nl; i += CSR_THREADS) if (lh[i]) atomicAdd(&list_nchunks[i], lh[i]); }
}
#endif
#ifndef KMR_INSTANCE_TU
__global__ __launch_bounds__(CSR_THREADS)
void chunk_scatter_kernel(const uint32_t *chunk_list, const uint32_t *chunk_count, uint32_t n_chunks, uint32_t first, const uint64_t *list_start,
                          uint32_t *cursor, uint64_t *list_chunks, uint32_t nl) {
	__shared__ uint32_t lh[CSR_LDS_LISTS];
	const bool priv = nl <= (uint32_t)CSR_LDS_LISTS;
	const uint64_t base = (uint64_t)blockIdx.x * CSR_THREADS * CSR_ITEMS;
	uint32_t myl[CSR_ITEMS], myr[CSR_ITEMS];
	if (priv) { for (uint32_t i = threadIdx.x; i < nl; i += CSR_THREADS) lh[i] = 0; __syncthreads(); }
#pragma unroll
	for (int j = 0; j < CSR_ITEMS; j++) {
		const uint64_t c = base + (uint64_t)j * CSR_THREADS + threadIdx.x;
		myl[j] = c < n_chunks ? chunk_list[c] : NO_CHUNK;
		if (myl[j] != NO_CHUNK && priv) myr[j] = atomicAdd(&lh[myl[j]], 1u);      /* rank inside the block */
	}
	if (priv) {
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < nl; i += CSR_THREADS) { const uint32_t n = lh[i]; lh[i] = n ? atomicAdd(&cursor[i], n) : 0u; }   /* block's base in the list */
		__syncthreads();
	}
#pragma unroll
	for (int j = 0; j < CSR_ITEMS; j++) {
		const uint32_t l = myl[j];
		if (l == NO_CHUNK) continue;
		const uint64_t c = base + (uint64_t)j * CSR_THREADS + threadIdx.x;
		const uint32_t pos = priv ? lh[l] + myr[j] : atomicAdd(&cursor[l], 1u);
		list_chunks[list_start[l] + pos] = ((uint64_t)chunk_count[c] << 32) | (c + first);      /* chunk_list/chunk_count point at chunk `first` */
	}
}
#endif

/* debugging aid (KMR_DEBUG): records held by a pool */
#ifndef KMR_INSTANCE_TU
__global__ void pool_records_kernel(const uint32_t *chunk_list, const uint32_t *chunk_count, uint32_t n_chunks, unsigned long long *total, unsigned long long *nvalid) {
	unsigned long long s = 0, v = 0;
	for (uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; c < n_chunks; c += (uint64_t)gridDim.x * blockDim.x)
		if (chunk_list[c] != NO_CHUNK) { s += chunk_count[c]; v++; }
	s = wave_sum(s); v = wave_sum(v);
	if ((threadIdx.x & 63) == 0) { atomicAdd(total, s); atomicAdd(nvalid, v); }
}
#endif

/* Share of distinct keys among the records, measured on a sample, so the count pass can size its lists: a k-mer's
 * copies all have the same partition order, so the PROBE_SPLIT blocks of probe p scan ONE level-1 list and keep the
 * records whose next `sel` order bits are zero -- the records one would-be final list would receive, all copies
 * included -- and count them and their distinct 64-bit fingerprints in a small table in device memory (few records
 * pass the filter, so its atomics are few).  out[0] += records, out[1] += distinct, out[2] += distinct keys seen more than once. */
static const int PROBE_SLOTS = 8192, PROBE_SPLIT = 16, PROBE_LISTS = 64;
template <int W, bool EXT>
__global__ __launch_bounds__(256)
void distinct_probe_kernel(PoolView pool, const uint64_t *list_start, const uint64_t *list_chunks, uint64_t n_lists, uint32_t kb, uint32_t rot,
                           int bits1, uint32_t n_probes, unsigned long long *tables /* [n_probes][PROBE_SLOTS], zero */, unsigned long long *out) {
	typedef typename PoolRec<W, EXT>::type Rec;
	const uint32_t p = blockIdx.x / PROBE_SPLIT, part = blockIdx.x % PROBE_SPLIT;
	const uint64_t l = (uint64_t)p * n_lists / n_probes;
	const uint64_t c0 = list_start[l], c1 = list_start[l + 1];
	unsigned long long *fp = tables + (size_t)p * PROBE_SLOTS;
	/* about 2000 records of the list pass the filter */
	int sel = 0; while (sel < 40 - bits1 && (((c1 - c0) * CH) >> sel) > 2000) sel++;
	unsigned long long mine = 0, fresh = 0, again = 0;
	constexpr int UNR = 4;
	const uint64_t nw = (uint64_t)(blockDim.x >> 6) * PROBE_SPLIT;
	for (uint64_t cb = c0 + (uint64_t)part * (blockDim.x >> 6) + (threadIdx.x >> 6); cb < c1; cb += nw * UNR) {
		uint64_t d[UNR]; Rec r[UNR];
#pragma unroll
		for (int u = 0; u < UNR; u++) { const uint64_t ci = cb + (uint64_t)u * nw; d[u] = ci < c1 ? list_chunks[ci] : 0ull; }
#pragma unroll
		for (int u = 0; u < UNR; u++) {
			r[u].w = 0.0f;
			if ((uint32_t)(threadIdx.x & 63) < (uint32_t)(d[u] >> 32)) r[u] = ((const Rec *)(pool.base + (size_t)(uint32_t)d[u] * CH * sizeof(Rec)))[threadIdx.x & 63];
		}
#pragma unroll
		for (int u = 0; u < UNR; u++) {
			if (r[u].w == 0.0f) continue;
			const uint64_t g = part_order<W>(r[u].key, kb, rot);
			if (sel && ((g << bits1) >> (64 - sel)) != 0) continue;
			mine++;
			/* bit 0: slot taken, bit 1: the fingerprint has been seen again (counted once: the key will be a weak entry) */
			const unsigned long long f = (part_hash<W>(r[u].key) & ~3ull) | 1ull;
			uint32_t s = (uint32_t)(f >> 40) & (PROBE_SLOTS - 1);
			for (int probe = 0; probe < PROBE_SLOTS; probe++) {
				const unsigned long long old = atomicCAS(&fp[s], 0ull, f);
				if (old == 0ull) { fresh++; break; }
				if ((old & ~2ull) == f) { if (!(old & 2ull) && !(atomicOr(&fp[s], 2ull) & 2ull)) again++; break; }
				s = (s + 1) & (PROBE_SLOTS - 1);
			}
		}
	}
	mine = wave_sum(mine); fresh = wave_sum(fresh); again = wave_sum(again);
	if ((threadIdx.x & 63) == 0) { if (mine) atomicAdd(&out[0], mine); if (fresh) atomicAdd(&out[1], fresh); if (again) atomicAdd(&out[2], again); }
}

/* debugging aid (KMR_DEBUG): walk a pool through its chunk CSR; count the records and those whose partition hash
 * does not match the list they are filed under */
template <int W, bool EXT>
__global__ void verify_lists_kernel(PoolView pool, const uint64_t *list_start, const uint64_t *list_chunks, uint64_t n_lists, int bits, uint32_t kb, uint32_t rot,
                                    unsigned long long *total, unsigned long long *misfiled, unsigned long long *zero_w) {
	typedef typename PoolRec<W, EXT>::type Rec;
	unsigned long long n = 0, bad = 0, zw = 0;
	for (uint64_t l = blockIdx.x; l < n_lists; l += gridDim.x) {
		for (uint64_t ci = list_start[l] + (threadIdx.x >> 6); ci < list_start[l + 1]; ci += blockDim.x >> 6) {
			const uint64_t d = list_chunks[ci];
			if ((uint32_t)(threadIdx.x & 63) < (uint32_t)(d >> 32)) {
				const Rec r = ((const Rec *)(pool.base + (size_t)(uint32_t)d * CH * sizeof(Rec)))[threadIdx.x & 63];
				n++;
				if (bits && (part_order<W>(r.key, kb, rot) >> (64 - bits)) != l) bad++;
				if (r.w == 0.0f) zw++;
			}
		}
	}
	n = wave_sum(n); bad = wave_sum(bad); zw = wave_sum(zw);
	if ((threadIdx.x & 63) == 0) { atomicAdd(total, n); atomicAdd(misfiled, bad); atomicAdd(zero_w, zw); }
}

/* counts[bucket] += 1 for every live lane of the wavefront (all 64 lanes call it).  The entries of a final list
 * share a few buckets, so lanes with equal buckets elect one of them to add their number; lanes still unserved
 * after a few rounds add for themselves. */
__device__ __forceinline__ void bucket_count_add(uint32_t *counts, uint64_t bucket, bool live) {
	const int lane = (int)(threadIdx.x & 63);
	bool done = !live;
	for (int round = 0; round < 6; round++) {
		const unsigned long long pending = __ballot(!done);
		if (!pending) break;
		const int leader = __ffsll((long long)pending) - 1;
		const uint64_t lb = ((uint64_t)(uint32_t)__shfl((int)(bucket >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)bucket, leader);
		const bool mine = !done && bucket == lb;
		const unsigned long long same = __ballot(mine);
		if (lane == leader) atomicAdd(&counts[lb], (uint32_t)__builtin_popcountll(same));
		if (mine) done = true;
	}
	if (!done) atomicAdd(&counts[bucket], 1u);
}

/* ------------------------------------------------------------------ count */
static const int COUNT_THREADS = 256;

struct CountOut {
	/* unsorted kept entries */
	uint64_t *wkeys; uint32_t *wvals; unsigned long long *wcursor; uint64_t wcap;
	uint64_t *wentries;      /* build_mode 3: the weak entries packed, W key words + one value word each (kmr_buckets.hpp), instead of wkeys / wvals */
	uint64_t *skeys; uint8_t *sweight; uint32_t *spkt; unsigned long long *scursor; uint64_t scap;
	uint32_t *weakCount, *singCount;        /* per bucket */
	FinalizeCounters *fc;
	uint32_t *err;
};

/* NARROW (extension values only): the 12 tallies of a slot are kept as 16-bit halves of six words, which is exact for a
 * list of at most 65 535 records -- 60 instead of 84 bytes per slot, two blocks per CU instead of one.  COUNT_NARROW_CHUNKS
 * is the longest list (in chunks of CH records) the narrow instantiation takes; the few longer ones (a k-mer that repeats
 * 10^5 times) go through the wide one in a second launch (list_filter). */
static const uint64_t COUNT_NARROW_CHUNKS = 65535 / CH;
#ifndef KMR_INSTANCE_TU
__global__ void max_list_chunks_kernel(const uint64_t *list_start, uint64_t n_lists, unsigned int *out) {
	unsigned int m = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_lists; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t n = list_start[i + 1] - list_start[i];
		const unsigned int c = n > 0xffffffffull ? 0xffffffffu : (unsigned int)n;
		m = c > m ? c : m;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { const unsigned int x = (unsigned int)__shfl_xor((int)m, off, 64); m = x > m ? x : m; }
	if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
#endif
template <int W, bool EXT, int LOG2S, bool NARROW = false>
__host__ __device__ constexpr size_t count_smem_bytes() { return (size_t)(1 << LOG2S) * (8 * W + 24 + (W > 1 ? 4 : 0) + (EXT ? (NARROW ? 28 : 52) : 0)); }

/* One block per final list.  LDS table: keys, count|fwd<<32, f64 weight sum, first
 * (ordinal<<1|fwd).  If the table would overflow the list is split by further hash bits
 * and done in sub-passes (a tiny LDS stack), so any input is handled. */
template <int W, bool EXT, int LOG2S, bool NARROW = false>
__global__ __launch_bounds__(COUNT_THREADS, (W == 1 && !EXT && LOG2S <= 10) ? 4 : (NARROW ? 2 : 1))      /* W == 1: four blocks per CU (<= 128 VGPRs, < 40 KB LDS each) */
void count_kernel(PoolView pool, const uint64_t *list_start, const uint64_t *list_chunks, uint64_t n_lists,
                  CountOut out, FinalizeParams f, unsigned int *work_counter, int list_filter = 0) {
	constexpr int TW = NARROW ? 6 : 12;          /* tally words per slot */
	/* list_filter 1: only lists of at most COUNT_NARROW_CHUNKS chunks, 2: only the longer ones, 0: all */
	auto skipped = [&](uint64_t a, uint64_t b) -> bool { return list_filter == 1 ? b - a > COUNT_NARROW_CHUNKS : (list_filter == 2 ? b - a <= COUNT_NARROW_CHUNKS : false); };
	constexpr int S = 1 << LOG2S;
	constexpr uint32_t LIMIT = (uint32_t)(S * 0.80);
	typedef typename PoolRec<W, EXT>::type Rec;
	extern __shared__ __attribute__((aligned(16))) uint8_t csm[];
	uint64_t *tkeys = (uint64_t *)csm;                                 /* [S][W] */
	unsigned long long *tcnt = (unsigned long long *)(tkeys + (size_t)S * W);
	double *twsum = (double *)(tcnt + S);
	unsigned long long *tfirst = (unsigned long long *)(twsum + S);
	uint32_t *tstate = (uint32_t *)(tfirst + S);                       /* W > 1 only */
	uint32_t *ttally = tstate + (W > 1 ? S : 0);                       /* EXT: [S][12] extension tallies, then [S] one packet */
	uint32_t *tpkt = ttally + (EXT ? TW * S : 0);
	__shared__ uint32_t s_list, s_claimed, s_overflow, s_sp, s_nw, s_ns;
	__shared__ unsigned long long s_wbase, s_sbase;
	/* output space is taken from the global cursors one slab at a time (a per-list atomic on one word would
	 * serialise ~10^6 lists); the unused tail of a slab is marked as holes (count 0 / weight 0) */
	__shared__ unsigned long long s_wpos, s_wend, s_spos, s_send;
	__shared__ unsigned long long s_holeW0, s_holeW1, s_holeS0, s_holeS1;
	__shared__ uint16_t s_kept[S];
	__shared__ uint32_t s_stackBits[40], s_stackVal[40];
	const int t = threadIdx.x;
	const uint32_t vw = EXT ? 15 : 3;
	unsigned long long uniq = 0, single = 0, keptW = 0, keptS = 0;   /* kept*: thread 0 only */
	constexpr unsigned long long OSLAB = 8192;
	if (t == 0) { s_wpos = s_wend = 0; s_spos = s_send = 0; }
	lds_barrier();

	/* Lists are taken LBATCH at a time (one word of device memory serves ~90 M atomics/s).  A batch's chunk
	 * descriptors are contiguous in list_chunks and are copied to LDS once, so the records of list j+1 can be
	 * requested (one chunk per wavefront per register slot, HEAD chunks in all) before list j is counted: the
	 * HBM latency of a list is hidden behind the LDS work of the one before it. */
	constexpr uint32_t LBATCH = 24, DESC_CAP = 512;        /* ~20 chunk descriptors per list at the usual list size */
	constexpr int UNR = W == 1 ? 5 : (W == 2 ? 4 : 3);
	constexpr int NWAVE = COUNT_THREADS / CH;
	constexpr uint64_t HEAD = (uint64_t)NWAVE * UNR;
	__shared__ unsigned long long s_ls[LBATCH + 1];
	__shared__ unsigned long long s_desc[DESC_CAP];
	Rec rr[UNR], rn[UNR];
	for (;;) {
		if (t == 0) s_list = atomicAdd(work_counter, LBATCH);
		lds_barrier();
		const uint64_t lfirst = s_list;
		if (lfirst >= n_lists) break;
		const uint32_t nl = (uint32_t)(n_lists - lfirst < (uint64_t)LBATCH ? n_lists - lfirst : (uint64_t)LBATCH);
		if ((uint32_t)t <= nl) s_ls[t] = list_start[lfirst + t];
		lds_barrier();
		const uint64_t dbase = s_ls[0], dend = s_ls[nl];
		for (uint64_t i = dbase + t; i < dend && i - dbase < DESC_CAP; i += COUNT_THREADS) s_desc[i - dbase] = list_chunks[i];
		lds_barrier();
		/* descriptors first (all from LDS, or for a list beyond the LDS window all from global memory: the two
		 * are never mixed in one access, which would turn it into a flat load that waits for everything in
		 * flight), then the record loads back to back */
		auto load_round = [&](uint64_t cb, uint64_t c1, Rec (&dst)[UNR]) {
			uint64_t d[UNR];
			const bool in_lds = c1 - dbase <= (uint64_t)DESC_CAP;      /* uniform */
			if (in_lds) {
#pragma unroll
				for (int u = 0; u < UNR; u++) { const uint64_t ci = cb + (t >> 6) + (uint64_t)u * NWAVE; d[u] = ci < c1 ? s_desc[ci - dbase] : 0ull; }
			} else {
#pragma unroll
				for (int u = 0; u < UNR; u++) { const uint64_t ci = cb + (t >> 6) + (uint64_t)u * NWAVE; d[u] = ci < c1 ? list_chunks[ci] : 0ull; }
			}
			/* a register slot without a record is marked by weight 0 (no record in a pool has it) */
#pragma unroll
			for (int u = 0; u < UNR; u++) {
				dst[u].w = 0.0f;
				if ((uint32_t)(t & 63) < (uint32_t)(d[u] >> 32)) dst[u] = ((const Rec *)(pool.base + (size_t)(uint32_t)d[u] * CH * sizeof(Rec)))[t & 63];
			}
		};
		if (!skipped(s_ls[0], s_ls[1])) load_round(s_ls[0], s_ls[1], rn);
		for (uint32_t j = 0; j < nl; j++) {
		const uint64_t c0 = s_ls[j], c1 = s_ls[j + 1];
#pragma unroll
		for (int u = 0; u < UNR; u++) rr[u] = rn[u];
		if (j + 1 < nl && !skipped(s_ls[j + 1], s_ls[j + 2])) load_round(s_ls[j + 1], s_ls[j + 2], rn);
		if (c0 == c1 || skipped(c0, c1)) continue;
		bool head_in_regs = true;       /* rr holds chunks [c0, c0 + HEAD) until a later round or sub-pass reloads it */
		lds_barrier();       /* every thread has left the previous list's while (s_sp > 0) before s_sp is re-armed */
		if (t == 0) { s_sp = 1; s_stackBits[0] = 0; s_stackVal[0] = 0; }
		lds_barrier();
		while (s_sp > 0) {
			const uint32_t bits = s_stackBits[s_sp - 1], val = s_stackVal[s_sp - 1];
			lds_barrier();
			if (t == 0) { s_sp--; s_claimed = 0; s_overflow = 0; s_nw = 0; s_ns = 0; }
			for (int i = t; i < S; i += COUNT_THREADS) { tkeys[(size_t)i * W] = EMPTY_KEY; tcnt[i] = 0; twsum[i] = 0.0; tfirst[i] = NO_FIRST; if (W > 1) tstate[i] = 0; }
			if (EXT) for (int i = t; i < TW * S; i += COUNT_THREADS) ttally[i] = 0;
			lds_barrier();
			const uint32_t subMask = bits ? ((1u << bits) - 1) : 0;
			/* insert: wave w takes chunks cb + w, + waves, ...; lane = record */
			for (uint64_t cb = c0; cb < c1 && !s_overflow && s_claimed <= LIMIT; cb += HEAD) {
				if (!(head_in_regs && cb == c0)) { load_round(cb, c1, rr); head_in_regs = false; }
				/* The probe of a record is a chain of dependent LDS round trips, so the first probe of the UNR
				 * records a thread holds is issued as one group (most records repeat a key that sits in its home
				 * slot); only what is left goes through the one-record-at-a-time loop.  New keys are counted per
				 * wavefront and added to s_claimed once a round, without waiting for the sum. */
				constexpr uint32_t NO_SLOT = 0xffffffffu;
				uint32_t slot[UNR], claimedHere = 0;
				uint64_t seen[UNR];          /* W == 1: key read from the home slot */
#pragma unroll
				for (int u = 0; u < UNR; u++) {
					const uint64_t h = slot_hash<W>(rr[u].key);
					const bool mine = rr[u].w != 0.0f && ((uint32_t)(h >> 20) & subMask) == val;
					slot[u] = mine ? (uint32_t)(h >> (64 - LOG2S)) : NO_SLOT;
					seen[u] = 0;
					if constexpr (W == 1) if (mine) seen[u] = tkeys[slot[u]];
				}
#pragma unroll
				for (int u = 0; u < UNR; u++) {
				if (slot[u] == NO_SLOT) continue;
				const Rec r = rr[u];
				uint32_t s = slot[u];
				bool placed = false;
				int probe = 0;
				if constexpr (W == 1) {
					/* home slot, from the grouped read */
					uint64_t curk = seen[u];
					if (curk == EMPTY_KEY) {
						const unsigned long long old = atomicCAS((unsigned long long *)&tkeys[s], (unsigned long long)EMPTY_KEY, (unsigned long long)r.key[0]);
						if (old == EMPTY_KEY) { claimedHere++; placed = true; }
						curk = old;
					}
					if (!placed && curk == r.key[0]) placed = true;
					if (!placed) { s = (s + 1) & (S - 1); probe = 1; }
				}
				for (; probe < S && !placed; ) {
					if constexpr (W == 1) {
						/* two slots per round trip: in a wavefront the longest probe sequence of 64 lanes sets the pace (about
						 * four slots at this load), and every step of it is an LDS latency */
						const uint32_t s2 = (s + 1) & (S - 1);
						uint64_t curk = tkeys[s], nextk = tkeys[s2];
						if (curk == EMPTY_KEY) {
							const unsigned long long old = atomicCAS((unsigned long long *)&tkeys[s], (unsigned long long)EMPTY_KEY, (unsigned long long)r.key[0]);
							if (old == EMPTY_KEY) { claimedHere++; placed = true; break; }
							curk = old;
						}
						if (curk == r.key[0]) { placed = true; break; }
						/* slot s holds another key for good: slot s2 is next in this key's probe order */
						s = s2; probe++;
						if (nextk == EMPTY_KEY) {
							const unsigned long long old = atomicCAS((unsigned long long *)&tkeys[s], (unsigned long long)EMPTY_KEY, (unsigned long long)r.key[0]);
							if (old == EMPTY_KEY) { claimedHere++; placed = true; break; }
							nextk = old;
						}
						if (nextk == r.key[0]) { placed = true; break; }
					} else {
						/* state word: 0 empty, 1 being written, 2 ready (same protocol as the global table) */
						uint32_t st = __hip_atomic_load(&tstate[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
						if (st == 0) {
							const uint32_t old = atomicCAS(&tstate[s], 0u, 1u);
							if (old == 0) {
#pragma unroll
								for (int j = 0; j < W; j++) tkeys[(size_t)s * W + j] = r.key[j];
								__hip_atomic_store(&tstate[s], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
								claimedHere++;
								placed = true; break;
							}
							st = old;
						}
						if (st == 1) continue;      /* writer publishes unconditionally: re-poll the same slot */
						bool eq = true;
#pragma unroll
						for (int j = 0; j < W; j++) eq = eq && (tkeys[(size_t)s * W + j] == r.key[j]);
						if (eq) { placed = true; break; }
					}
					s = (s + 1) & (S - 1);
					probe++;
				}
				if (!placed) { s_overflow = 1; continue; }      /* table full */
				const bool fwd = !(r.w < 0.0f);
				const float wa = fwd ? r.w : -r.w;
				atomicAdd(&tcnt[s], 1ull | ((unsigned long long)(fwd ? 1 : 0) << 32));
				atomicAdd(&twsum[s], (double)wa);
				atomicMin(&tfirst[s], first_pack(rec_ordinal<W>(r), fwd, wa));
				if constexpr (EXT) {      /* ExtensionTracking::trackExtension (src/KmerTrackingData.h:195-201) */
					const int lc = ext_tally_index(r.pkt & 0xff), rc_ = ext_tally_index((r.pkt >> 8) & 0xff);
					const uint32_t lq = (r.pkt >> 16) & 0xff, rq_ = r.pkt >> 24;
					if (NARROW) {
						if (lq >= f.ext_min_q || lc > 3) atomicAdd(&ttally[(size_t)s * TW + (lc >> 1)], 1u << (16 * (lc & 1)));
						if (rq_ >= f.ext_min_q || rc_ > 3) atomicAdd(&ttally[(size_t)s * TW + ((6 + rc_) >> 1)], 1u << (16 * ((6 + rc_) & 1)));
					} else {
						if (lq >= f.ext_min_q || lc > 3) atomicAdd(&ttally[(size_t)s * 12 + lc], 1u);
						if (rq_ >= f.ext_min_q || rc_ > 3) atomicAdd(&ttally[(size_t)s * 12 + 6 + rc_], 1u);
					}
					tpkt[s] = r.pkt;      /* exact whenever the key ends with one occurrence, the only case it is read */
				}
				}
				claimedHere = (uint32_t)wave_sum((unsigned long long)claimedHere);
				if ((t & 63) == 0 && claimedHere) atomicAdd(&s_claimed, claimedHere);
			}
			lds_barrier();
			if (s_overflow || s_claimed > LIMIT) {       /* split this sub-pass in two by one more hash bit */
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
			/* emit: the slots of the kept entries are first collected in LDS (weak entries from the front of
			 * s_kept, singletons from its back), so that the bucket hash and the stores below run on a dense
			 * range of threads instead of once per table slot */
			/* classify() of kmr_kernels.hpp with its case analysis folded into two launch-wide scalars: a count of one goes
			 * to class singC when singletons are separate, any other count below weakMin is dropped */
			const uint32_t singC = f.has_singletons ? (f.min_depth > 1 ? 0u : 2u) : 3u;       /* 3: no special case */
			const uint32_t weakMin = ((!f.has_singletons || f.min_depth > 2) && f.min_depth != 1) ? f.min_depth : 0u;
			/* all four slots of a thread are read first, then ranked: one LDS atomic per wavefront and class */
			uint32_t cls[S / COUNT_THREADS];
#pragma unroll
			for (int i = 0; i < S / COUNT_THREADS; i++) {
				const int s = i * COUNT_THREADS + t;
				const bool used = W == 1 ? tkeys[s] != EMPTY_KEY : tstate[s] == 2;
				const uint32_t count = (uint32_t)tcnt[s];            /* 0 in a free slot */
				uniq += used ? 1u : 0u;
				single += (used && count == 1) ? 1u : 0u;
				cls[i] = !used ? 0u : ((count == 1 && singC != 3u) ? singC : (count < weakMin ? 0u : 1u));
			}
			const unsigned long long below = (1ull << (t & 63)) - 1;
#pragma unroll
			for (int i = 0; i < S / COUNT_THREADS; i++) {
				const int s = i * COUNT_THREADS + t;
				const uint32_t c = cls[i];
				const unsigned long long mw = __ballot(c == 1), ms = __ballot(c == 2);
				if ((mw | ms) == 0) continue;                          /* wave-uniform: most slots are free or dropped */
				uint32_t bw = 0, bs = 0;
				if ((t & 63) == 0) { if (mw) bw = atomicAdd(&s_nw, (uint32_t)__builtin_popcountll(mw)); if (ms) bs = atomicAdd(&s_ns, (uint32_t)__builtin_popcountll(ms)); }
				bw = (uint32_t)__shfl((int)bw, 0, 64); bs = (uint32_t)__shfl((int)bs, 0, 64);
				if (c == 1) s_kept[bw + (uint32_t)__builtin_popcountll(mw & below)] = (uint16_t)s;
				else if (c == 2) s_kept[S - 1 - (bs + (uint32_t)__builtin_popcountll(ms & below))] = (uint16_t)s;
			}
			lds_barrier();
			/* thread 0 alone does the slab book-keeping between these two barriers (the other threads must not look
			 * at s_wpos & co. while it moves them); a slab that cannot take this list is retired and its unused
			 * tail, shorter than the list's entry count, is handed to the block to be marked as holes */
			if (t == 0) {
				s_holeW0 = s_holeW1 = 0; s_holeS0 = s_holeS1 = 0;
				if (s_nw && s_wpos + s_nw > s_wend) {
					s_holeW0 = s_wpos; s_holeW1 = s_wend < out.wcap ? s_wend : out.wcap;
					const unsigned long long g = s_nw > OSLAB ? s_nw : OSLAB; s_wpos = atomicAdd(out.wcursor, g); s_wend = s_wpos + g;
				}
				if (s_ns && s_spos + s_ns > s_send) {
					s_holeS0 = s_spos; s_holeS1 = s_send < out.scap ? s_send : out.scap;
					const unsigned long long g = s_ns > OSLAB ? s_ns : OSLAB; s_spos = atomicAdd(out.scursor, g); s_send = s_spos + g;
				}
				s_wbase = s_wpos; s_wpos += s_nw; keptW += s_nw;
				s_sbase = s_spos; s_spos += s_ns; keptS += s_ns;
				if (s_wend > out.wcap || s_send > out.scap) { atomicOr(out.err, (uint32_t)ERR_ENTRIES_FULL); s_nw = 0xffffffffu; }
			}
			lds_barrier();
			for (unsigned long long e = s_holeW0 + t; e < s_holeW1; e += COUNT_THREADS) out.wvals[e * vw] = 0;
			for (unsigned long long e = s_holeS0 + t; e < s_holeS1; e += COUNT_THREADS) out.sweight[e] = 0;
			if (s_nw != 0xffffffffu) {
				for (uint32_t e0 = (uint32_t)t & ~63u; e0 < s_nw; e0 += COUNT_THREADS) {
					const uint32_t e = e0 + (uint32_t)(t & 63);
					const bool live = e < s_nw;
					uint64_t bucket = 0;
					if (live) {
						const uint32_t s = s_kept[e];
						Key<W> key;
#pragma unroll
						for (int j = 0; j < W; j++) key.w[j] = tkeys[(size_t)s * W + j];
						bucket = key_hash<W>(key, f.kb) & (f.nb_weak - 1);
						const unsigned long long cf = tcnt[s];
						const uint64_t pos = s_wbase + e;
#pragma unroll
						for (int j = 0; j < W; j++) out.wkeys[pos * W + j] = key.w[j];
						uint32_t fwd = (uint32_t)(cf >> 32), cnt = (uint32_t)cf;
						const unsigned long long fst = tfirst[s];
						if (f.has_singletons && first_forward(fst)) fwd -= 1;
						if (cnt > 65535u) { cnt = 65535u; if (fwd > 65534u) fwd = 65534u; }
						if (fwd > 65535u) fwd = 65535u;
						uint32_t *v = out.wvals + pos * vw;
						v[0] = cnt; v[1] = __float_as_uint((float)(f.has_singletons ? twsum[s] + first_weight_shift(fst) : twsum[s])); v[2] = fwd;
						if (EXT) {
#pragma unroll
							for (int j = 0; j < 12; j++) v[3 + j] = NARROW ? ((ttally[(size_t)s * TW + (j >> 1)] >> (16 * (j & 1))) & 0xffffu) : ttally[(size_t)s * 12 + j];
						}
					}
					bucket_count_add(out.weakCount, bucket, live);
				}
				for (uint32_t e0 = (uint32_t)t & ~63u; e0 < s_ns; e0 += COUNT_THREADS) {
					const uint32_t e = e0 + (uint32_t)(t & 63);
					const bool live = e < s_ns;
					uint64_t bucket = 0;
					if (live) {
						const uint32_t s = s_kept[S - 1 - e];
						Key<W> key;
#pragma unroll
						for (int j = 0; j < W; j++) key.w[j] = tkeys[(size_t)s * W + j];
						bucket = key_hash<W>(key, f.kb) & (f.nb_sing - 1);
						const uint64_t pos = s_sbase + e;
#pragma unroll
						for (int j = 0; j < W; j++) out.skeys[pos * W + j] = key.w[j];
						const float wf = (float)twsum[s];
						out.sweight[pos] = (uint8_t)((unsigned char)(((double)wf * 254.0)) + 1);
						if (EXT) out.spkt[pos] = tpkt[s];
					}
					bucket_count_add(out.singCount, bucket, live);
				}
			}
			lds_barrier();
		}
		}
	}
	lds_barrier();
	for (unsigned long long e = s_wpos + t; e < s_wend && e < out.wcap; e += COUNT_THREADS) out.wvals[e * vw] = 0;
	for (unsigned long long e = s_spos + t; e < s_send && e < out.scap; e += COUNT_THREADS) out.sweight[e] = 0;
	uniq = wave_sum(uniq); single = wave_sum(single);
	if ((t & 63) == 0) { if (uniq) atomicAdd(&out.fc->unique, uniq); if (single) atomicAdd(&out.fc->singletons, single); }
	if (t == 0) { if (keptW) atomicAdd(&out.fc->weak_kept, keptW); if (keptS) atomicAdd(&out.fc->sing_kept, keptS); }
}

/* unsorted entries -> their bucket segments (then sort_buckets_kernel).  The entries one wavefront sees come
 * from one or two final lists and with the partition cut along the bucket index (part_order) they fall into a
 * handful of neighbouring buckets: lanes with the same bucket share one cursor update (a few rounds of
 * ballot-and-elect) instead of serialising on it; what is left after the rounds takes its own atomic. */
template <int W, bool WIDE = false>
__global__ void entry_scatter_kernel(const uint64_t *ukeys, const uint32_t *uvals, const uint8_t *ub8, const uint32_t *upkt, uint64_t n,
                                     uint32_t vw, uint32_t kb, uint64_t nb, const uint64_t *start, uint32_t *cursor,
                                     uint64_t *keys, uint32_t *vals, uint8_t *b8, uint32_t *pkt, unsigned long long *cursor64 = nullptr, int rounds = 6) {
	const int lane = (int)(threadIdx.x & 63);
	for (uint64_t e0 = blockIdx.x * (uint64_t)blockDim.x + (threadIdx.x & ~63u); e0 < n; e0 += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t e = e0 + lane;
		const bool live = e < n && (uvals ? uvals[e * vw] != 0 : ub8[e] != 0);      /* holes end an output slab */
		Key<W> key;
		uint64_t b = 0;
		if (live) {
#pragma unroll
			for (int j = 0; j < W; j++) key.w[j] = ukeys[e * W + j];
			b = key_hash<W>(key, kb) & (nb - 1);
		}
		bool done = !live;
		uint32_t rank = 0;
		/* cursor64 (the entries of a wavefront fall into unrelated buckets: lists cut by minimizer): the cursor of a bucket starts at
		 * the bucket's first entry, so one returning add gives the position -- no election rounds, no load of start[] */
		if (cursor64) {
			if (!WIDE) {
				if (!live) continue;
				const uint64_t pos = atomicAdd(&cursor64[b], 1ull);
#pragma unroll
				for (int j = 0; j < W; j++) keys[pos * W + j] = key.w[j];
				if (uvals) for (uint32_t j = 0; j < vw; j++) vals[pos * vw + j] = uvals[e * vw + j];
				if (ub8) b8[pos] = ub8[e];
				if (upkt) pkt[pos] = upkt[e];
				continue;
			}
		}
		for (int round = 0; round < rounds; round++) {
			const unsigned long long pending = __ballot(!done);
			if (!pending) break;
			const int leader = __ffsll((long long)pending) - 1;
			const uint64_t lb = ((uint64_t)(uint32_t)__shfl((int)(b >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)b, leader);
			const bool mine = !done && b == lb;
			const unsigned long long same = __ballot(mine);
			uint32_t base = 0;
			if (lane == leader) base = atomicAdd(&cursor[lb], (uint32_t)__builtin_popcountll(same));
			base = (uint32_t)__shfl((int)base, leader);
			if (mine) { rank = base + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1)); done = true; }
		}
		if (live && !done) rank = atomicAdd(&cursor[b], 1u);
		if (!WIDE) {
			if (!live) continue;
			const uint64_t pos = start[b] + rank;
#pragma unroll
			for (int j = 0; j < W; j++) keys[pos * W + j] = key.w[j];
			if (uvals) for (uint32_t j = 0; j < vw; j++) vals[pos * vw + j] = uvals[e * vw + j];
			if (ub8) b8[pos] = ub8[e];
			if (upkt) pkt[pos] = upkt[e];
			continue;
		}
		const uint64_t pos = live ? start[b] + rank : 0;
		if (live) {
#pragma unroll
			for (int j = 0; j < W; j++) keys[pos * W + j] = key.w[j];
			if (ub8) b8[pos] = ub8[e];
			if (upkt) pkt[pos] = upkt[e];
		}
		if (uvals) {
			/* wide values (15 words with extension tallies): a lane copying its own entry word by word touches 64 lines per
			 * instruction; instead 16 lanes share an entry, four entries per step, so a step reads 4 x 60 contiguous bytes */
			for (int i0 = 0; i0 < 64; i0 += 4) {
				const int src = i0 + (lane >> 4);
				const uint32_t word = (uint32_t)(lane & 15);
				const int sl = __shfl(live ? 1 : 0, src);
				const uint64_t sp = ((uint64_t)(uint32_t)__shfl((int)(pos >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)pos, src);
				if (sl && word < vw) vals[sp * vw + word] = uvals[(e0 + src) * vw + word];
			}
		}
	}
}

}  // namespace kmr
#endif
