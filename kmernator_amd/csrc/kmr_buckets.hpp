/*
 * kmr_buckets.hpp -- the bucketed, sorted map out of unsorted kept entries, written in whole lines.
 *
 * The count pass of build_mode 3 emits the entries of a minimizer list; the bucket of an entry (lookup3 of the key,
 * src/Kmer.h:2329-2333) has nothing to do with the list it came from, so scattering the entries straight to their bucket
 * segments wrote one random 20-byte entry at a time (9 GB written for 1.16 GB of entries at C2).  Here the entries reach their
 * buckets by an MSD radix partition over the bucket index instead:
 *
 *   bb_hist_kernel      histogram of the level's bins (top bits of the bucket index below the bits already resolved)
 *   bb_scatter_kernel   a tile of 4096 entries is ranked inside LDS, one device atomic per (tile, bin) reserves the run, the
 *                       entries of a bin leave the tile as one contiguous run
 *   (one or two such levels, <= 10 bits each, until a GROUP of 2^g neighbouring buckets holds ~512-1024 entries)
 *   bb_group_kernel     one block per group: the group's entries in LDS, counting sort by bucket, rank sort by key inside
 *                       every bucket (KmerMapByKmerArrayPair::resort order, src/Kmer.h:3079-3088), coalesced store of keys
 *                       and values at their final places and of the buckets' start offsets
 *
 * The last level's bins are the groups and their exclusive scan is the final position of every group, so the group kernel
 * sorts in place.  Values: COUNT_DIR (three words) only; anything this geometry does not fit (a group that overflows the LDS
 * arrays, more than two levels) takes the scatter + per-bucket sort it replaces (entry_scatter_kernel, sort_buckets_kernel).
 */
#ifndef KMR_BUCKETS_HPP_
#define KMR_BUCKETS_HPP_

#include "kmr_partition.hpp"

namespace kmr {

/* kmr_sort.hip: (u64, u32) pairs sorted by key on the device (rocPRIM); tmp == nullptr returns the scratch size in *tmp_bytes */
int sort_pairs_u64_u32(void *tmp, size_t *tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, const unsigned int *vals_in, unsigned int *vals_out, size_t n, hipStream_t stream);

static const int BB_TILE = 8192;            /* entries per tile: a tile never spans two segments (8192 over 256 bins: runs of 32 entries = 512 bytes) */
static const int BB_THREADS = 512;
static const int BB_GROUP_THREADS = 256;    /* threads of a bb_group_kernel block */
static const int BB_PER_THREAD = BB_TILE / BB_THREADS;
static const int BB_MAX_BITS = 10;          /* bins per level <= 1024                                                   */
static const int BB_MAX_GROUP_BITS = 8;     /* buckets per group <= 256                                                 */
static const uint32_t BB_GROUP_CAP = 1536;  /* entries of a group the LDS arrays take (one-word keys: 30 KB, five blocks per CU) */
/* multi-word keys: 28 bytes an entry at W = 2 -- 1344 of them (37 KB) let four blocks share a CU where 1536 allowed three; the groups
 * are sized for 512-1023 entries on average, and one that is larger than the arrays sends the build down the other path */
template <int W> __host__ __device__ constexpr uint32_t bb_group_cap() { return W == 1 ? BB_GROUP_CAP : 1344u; }

/* the entries a level reads: segments (one at the first level: the count pass's slots with their holes; the bins of the level
 * before afterwards) whose starts are multiples of BB_TILE */
/* An entry on its way: W key words and ONE value word -- count | direction count << 16 | f32 weighted count << 32 (the three
 * fields of TrackingDataWithDirection, src/KmerTrackingData.h:505-535, both counts already clamped to 16 bits) -- so that an entry of
 * a one-word key moves as a single 16-byte store (the three separate value words and the key of the map's final layout were four
 * scattered write transactions per entry and level).  A zero value word is a hole. */
__host__ __device__ __forceinline__ uint64_t bb_pack_value(uint32_t count, uint32_t weighted_bits, uint32_t dir) { return (uint64_t)(count | (dir << 16)) | ((uint64_t)weighted_bits << 32); }

struct BbInput {
	const uint64_t *entries;                         /* [slots][W + 1] */
	const uint64_t *seg_start;                       /* [n_seg] first slot of a segment (multiple of BB_TILE); null: one segment at 0 */
	const uint32_t *seg_count;                       /* [n_seg] entries of a segment; null: n_slots                                     */
	uint32_t n_seg;
	uint64_t n_slots;                                /* slots in all (tiles = ceil(n_slots / BB_TILE))                                  */
	uint32_t holes;                                  /* 1: a slot whose value word is zero is a hole                                     */
};

template <int W> __device__ __forceinline__ void bb_load_entry(const uint64_t *entries, uint64_t e, uint64_t (&w)[W + 1]) {
	if constexpr (W == 1) { const uint4 v = ((const uint4 *)entries)[e]; w[0] = (uint64_t)v.x | ((uint64_t)v.y << 32); w[1] = (uint64_t)v.z | ((uint64_t)v.w << 32); }
	else {
#pragma unroll
		for (int q = 0; q <= W; q++) w[q] = entries[e * (W + 1) + q];
	}
}
template <int W> __device__ __forceinline__ void bb_store_entry(uint64_t *entries, uint64_t e, const uint64_t (&w)[W + 1]) {
	if constexpr (W == 1) ((uint4 *)entries)[e] = make_uint4((uint32_t)w[0], (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32));
	else {
#pragma unroll
		for (int q = 0; q <= W; q++) entries[e * (W + 1) + q] = w[q];
	}
}

/* segment of the tile that starts at slot s0: the last one whose start is <= s0 */
__device__ __forceinline__ uint32_t bb_segment_of(const BbInput &in, uint64_t s0) {
	if (!in.seg_start) return 0;
	uint32_t lo = 0, hi = in.n_seg;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (in.seg_start[mid] <= s0) lo = mid; else hi = mid; }
	return lo;
}

/* hist[(segment << bits) + bin] += entries of the segment whose bucket index has `bin` in bits [shift, shift + bits) */
template <int W>
__global__ __launch_bounds__(BB_THREADS)
void bb_hist_kernel(BbInput in, uint32_t shift, uint32_t bits, uint32_t kb, uint64_t nb, uint32_t tiles_per_block, uint32_t *hist) {
	__shared__ uint32_t lh[1 << BB_MAX_BITS];
	__shared__ uint32_t s_seg;
	const int t = threadIdx.x;
	const uint32_t bins = 1u << bits, mask = bins - 1;
	const uint64_t n_tiles = (in.n_slots + BB_TILE - 1) / BB_TILE;
	const uint64_t tile0 = (uint64_t)blockIdx.x * tiles_per_block;
	uint32_t curSeg = 0xffffffffu;
	for (uint32_t i = t; i < bins; i += BB_THREADS) lh[i] = 0;
	__syncthreads();
	for (uint64_t tile = tile0; tile < tile0 + tiles_per_block && tile < n_tiles; tile++) {
		const uint64_t s0 = tile * BB_TILE;
		if (t == 0) s_seg = bb_segment_of(in, s0);
		__syncthreads();
		const uint32_t seg = s_seg;
		if (seg != curSeg) {          /* a block's tiles are consecutive: the counts of a segment are flushed when it changes */
			if (curSeg != 0xffffffffu) {
				for (uint32_t i = t; i < bins; i += BB_THREADS) { const uint32_t c = lh[i]; if (c) { atomicAdd(&hist[((uint64_t)curSeg << bits) + i], c); lh[i] = 0; } }
				__syncthreads();
			}
			curSeg = seg;
		}
		const uint64_t segEnd = in.seg_start ? in.seg_start[seg] + in.seg_count[seg] : in.n_slots;
#pragma unroll 4
		for (int i = 0; i < BB_PER_THREAD; i++) {
			const uint64_t e = s0 + (uint64_t)i * BB_THREADS + t;
			if (e >= segEnd) continue;
			uint64_t ew[W + 1];
			bb_load_entry<W>(in.entries, e, ew);
			if (in.holes && ew[W] == 0) continue;
			Key<W> key;
#pragma unroll
			for (int q = 0; q < W; q++) key.w[q] = ew[q];
			const uint64_t b = key_hash<W>(key, kb) & (nb - 1);
			atomicAdd(&lh[(uint32_t)(b >> shift) & mask], 1u);
		}
		__syncthreads();
	}
	if (curSeg != 0xffffffffu) for (uint32_t i = t; i < bins; i += BB_THREADS) { const uint32_t c = lh[i]; if (c) atomicAdd(&hist[((uint64_t)curSeg << bits) + i], c); }
}

/* counts -> counts rounded up to whole tiles (the bins of a level that another level reads start at tile boundaries) */
#ifndef KMR_INSTANCE_TU
__global__ void bb_pad_kernel(const uint32_t *hist, uint64_t n, uint32_t *padded, unsigned int *max_count) {
	unsigned int m = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t c = hist[i];
		if (padded) padded[i] = (c + BB_TILE - 1) / BB_TILE * BB_TILE;
		m = c > m ? c : m;
	}
	for (int o = 32; o > 0; o >>= 1) { const unsigned int x = (unsigned int)__shfl_xor((int)m, o, 64); m = x > m ? x : m; }
	if ((threadIdx.x & 63) == 0 && m) atomicMax(max_count, m);
}
__global__ void bb_cursor_init_kernel(const uint64_t *start, uint64_t n, unsigned long long *cursor) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) cursor[i] = start[i];
}
#endif

/* the entries of every tile to their bins: cursor[(segment << bits) + bin] starts at the bin's first output slot */
template <int W>
__global__ __launch_bounds__(BB_THREADS)
void bb_scatter_kernel(BbInput in, uint32_t shift, uint32_t bits, uint32_t kb, uint64_t nb, unsigned long long *cursor, uint64_t *out_entries) {
	__shared__ uint32_t lh[1 << BB_MAX_BITS];
	__shared__ unsigned long long lbase[1 << BB_MAX_BITS];
	__shared__ uint32_t s_seg;
	const int t = threadIdx.x;
	const uint32_t bins = 1u << bits, mask = bins - 1;
	const uint64_t n_tiles = (in.n_slots + BB_TILE - 1) / BB_TILE;
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint64_t s0 = tile * BB_TILE;
		__syncthreads();
		if (t == 0) s_seg = bb_segment_of(in, s0);
		for (uint32_t i = t; i < bins; i += BB_THREADS) lh[i] = 0;
		__syncthreads();
		const uint32_t seg = s_seg;
		const uint64_t segEnd = in.seg_start ? in.seg_start[seg] + in.seg_count[seg] : in.n_slots;
		uint32_t where[BB_PER_THREAD];          /* bin << 16 | rank inside the tile's share of the bin; ~0: no entry */
#pragma unroll
		for (int i = 0; i < BB_PER_THREAD; i++) {
			const uint64_t e = s0 + (uint64_t)i * BB_THREADS + t;
			where[i] = 0xffffffffu;
			if (e >= segEnd) continue;
			uint64_t ew[W + 1];
			bb_load_entry<W>(in.entries, e, ew);
			if (in.holes && ew[W] == 0) continue;
			Key<W> key;
#pragma unroll
			for (int q = 0; q < W; q++) key.w[q] = ew[q];
			const uint32_t bin = (uint32_t)((key_hash<W>(key, kb) & (nb - 1)) >> shift) & mask;
			where[i] = (bin << 16) | atomicAdd(&lh[bin], 1u);
		}
		__syncthreads();
		for (uint32_t i = t; i < bins; i += BB_THREADS) { const uint32_t c = lh[i]; if (c) lbase[i] = atomicAdd(&cursor[((uint64_t)seg << bits) + i], (unsigned long long)c); }
		__syncthreads();
		/* second sweep over the tile (it is in L2) */
#pragma unroll
		for (int i = 0; i < BB_PER_THREAD; i++) {
			if (where[i] == 0xffffffffu) continue;
			const uint64_t e = s0 + (uint64_t)i * BB_THREADS + t;
			uint64_t ew[W + 1];
			bb_load_entry<W>(in.entries, e, ew);
			bb_store_entry<W>(out_entries, lbase[where[i] >> 16] + (where[i] & 0xffffu), ew);
		}
	}
}

static const int BB_GROUP_PER_THREAD = (BB_GROUP_CAP + BB_GROUP_THREADS - 1) / BB_GROUP_THREADS;
template <int W> __host__ __device__ constexpr size_t bb_group_smem_bytes() { return (size_t)bb_group_cap<W>() * (8 * W + 8 + 2 + 2); }

/* One block per group of 2^gbits neighbouring buckets: its entries lie at [gstart[G], gstart[G] + gcount[G]) of `entries` and go
 * to the same range of the map's key and value arrays, in (bucket, key) order.  A thread keeps its (up to six) entries in
 * registers while the buckets are counted and scanned, then files them into LDS bucket by bucket, so that the rank loop of an
 * entry walks the consecutive keys of its bucket (independent LDS reads, no index in between). */
template <int W>
__global__ __launch_bounds__(BB_GROUP_THREADS, W <= 2 ? 4 : 1)      /* (W = 2: 130 registers without the bound, three blocks per CU) */
void bb_group_kernel(const uint64_t *entries, uint64_t *keys, uint32_t *vals, const uint64_t *gstart, const uint32_t *gcount, uint64_t n_groups, uint32_t gbits, uint32_t kb, uint64_t nb,
                     uint64_t *start, uint64_t n_total, uint32_t *err) {
	extern __shared__ __attribute__((aligned(16))) uint8_t gsm[];
	uint64_t *skeys = (uint64_t *)gsm;                                   /* [CAP][W] keys, grouped by bucket */
	constexpr uint32_t CAP = bb_group_cap<W>();
	uint64_t *svalw = skeys + (size_t)CAP * W;                  /* [CAP] value words, same order   */
	uint16_t *sbucket = (uint16_t *)(svalw + CAP);              /* bucket (inside the group) of the entry at a position */
	uint16_t *final_ = sbucket + CAP;                           /* position (in bucket order) of the entry that ends up at a place */
	__shared__ uint32_t bcnt[1 << BB_MAX_GROUP_BITS], bstart[(1 << BB_MAX_GROUP_BITS) + 1], bscan[BB_GROUP_THREADS / 64];
	const int t = threadIdx.x;
	const uint32_t nbk = 1u << gbits;
	for (uint64_t G = blockIdx.x; G < n_groups; G += gridDim.x) {
		const uint64_t base = gstart[G];
		const uint32_t n = gcount[G];
		__syncthreads();
		if ((uint32_t)t < nbk) bcnt[t] = 0;
		if (n > CAP) { if (t == 0) atomicOr(err, (uint32_t)ERR_ENTRIES_FULL); continue; }      /* (the host looked at the largest group before the launch) */
		__syncthreads();
		uint64_t ew[BB_GROUP_PER_THREAD][W + 1]; uint32_t lb[BB_GROUP_PER_THREAD];
#pragma unroll
		for (int u = 0; u < BB_GROUP_PER_THREAD; u++) {
			const uint32_t i = (uint32_t)u * BB_GROUP_THREADS + t;
			lb[u] = 0xffffffffu;
			if (i < n) bb_load_entry<W>(entries, base + i, ew[u]);
		}
#pragma unroll
		for (int u = 0; u < BB_GROUP_PER_THREAD; u++) {
			const uint32_t i = (uint32_t)u * BB_GROUP_THREADS + t;
			if (i >= n) continue;
			Key<W> key;
#pragma unroll
			for (int q = 0; q < W; q++) key.w[q] = ew[u][q];
			lb[u] = (uint32_t)(key_hash<W>(key, kb) & (nb - 1)) & (nbk - 1);
			atomicAdd(&bcnt[lb[u]], 1u);
		}
		__syncthreads();
		/* exclusive scan of the bucket counts (<= 256 of them, one per thread): inside every wavefront, then the wavefronts' totals */
		{
			const uint32_t c = (uint32_t)t < nbk ? bcnt[t] : 0u;
			uint32_t incl = c;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o, 64); if ((t & 63) >= o) incl += x; }
			if ((t & 63) == 63) bscan[t >> 6] = incl;
			__syncthreads();
			uint32_t before = 0;
			for (int w = 0; w < (t >> 6); w++) before += bscan[w];
			if ((uint32_t)t < nbk) { bstart[t] = before + incl - c; start[G * nbk + t] = base + before + incl - c; bcnt[t] = 0; }
			if (t == 0) { bstart[nbk] = n; if (G == n_groups - 1) start[nb] = n_total; }
		}
		__syncthreads();
#pragma unroll
		for (int u = 0; u < BB_GROUP_PER_THREAD; u++) {
			if (lb[u] == 0xffffffffu) continue;
			const uint32_t pos = bstart[lb[u]] + atomicAdd(&bcnt[lb[u]], 1u);
#pragma unroll
			for (int q = 0; q < W; q++) skeys[(size_t)pos * W + q] = ew[u][q];
			svalw[pos] = ew[u][W];
			sbucket[pos] = (uint16_t)lb[u];
		}
		__syncthreads();
		for (uint32_t p = t; p < n; p += BB_GROUP_THREADS) {
			const uint32_t b = sbucket[p];
			const uint32_t s = bstart[b], e = bstart[b + 1];
			Key<W> mine;
#pragma unroll
			for (int q = 0; q < W; q++) mine.w[q] = skeys[(size_t)p * W + q];
			uint32_t rank = 0;
#pragma unroll 4
			for (uint32_t qx = s; qx < e; qx++) {
				Key<W> other;
#pragma unroll
				for (int q = 0; q < W; q++) other.w[q] = skeys[(size_t)qx * W + q];
				rank += (key_lt<W>(other, mine) || (key_eq<W>(other, mine) && qx < p)) ? 1u : 0u;
			}
			final_[s + rank] = (uint16_t)p;
		}
		__syncthreads();
		for (uint32_t p = t; p < n; p += BB_GROUP_THREADS) {
			const uint32_t i = final_[p];
#pragma unroll
			for (int q = 0; q < W; q++) keys[(base + p) * W + q] = skeys[(size_t)i * W + q];
			const uint64_t vw_ = svalw[i];
			vals[(base + p) * 3] = (uint32_t)vw_ & 0xffffu; vals[(base + p) * 3 + 1] = (uint32_t)(vw_ >> 32); vals[(base + p) * 3 + 2] = ((uint32_t)vw_ >> 16) & 0xffffu;
		}
	}
}

/* the fallback path (entry_scatter_kernel + sort_buckets_kernel over separate key and value arrays with per-bucket counts): the packed
 * entries taken apart again and counted per bucket */
template <int W>
__global__ __launch_bounds__(256)
void bb_unpack_kernel(const uint64_t *entries, uint64_t n_slots, uint32_t kb, uint64_t nb, uint64_t *keys, uint32_t *vals, uint32_t *counts) {
	for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n_slots; e += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t ew[W + 1];
		bb_load_entry<W>(entries, e, ew);
		Key<W> key;
#pragma unroll
		for (int q = 0; q < W; q++) { key.w[q] = ew[q]; keys[e * W + q] = ew[q]; }
		vals[e * 3] = (uint32_t)ew[W] & 0xffffu; vals[e * 3 + 1] = (uint32_t)(ew[W] >> 32); vals[e * 3 + 2] = ((uint32_t)ew[W] >> 16) & 0xffffu;
		if (ew[W] != 0) atomicAdd(&counts[key_hash<W>(key, kb) & (nb - 1)], 1u);
	}
}

}  // namespace kmr
#endif
