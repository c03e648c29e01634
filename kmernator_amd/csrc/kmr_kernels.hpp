/*
 * kmr_kernels.hpp -- HIP kernels (gfx950) of the k-mer spectrum build.
 *
 * Kernels, in pipeline order (reference function each one replaces in brackets):
 *
 *  extract_kernel<W,EXT,Op>   reads -> canonical k-mers + weights + hash -> Op
 *        [TwoBitSequence::compressSequence src/TwoBitSequence.cpp:242,
 *         KmerArrayPair::build src/Kmer.h:1323, KmerReadUtils::buildWeightedKmers
 *         src/KmerReadUtils.h:176, KmerHasher::getHash src/Kmer.h:207]
 *     Op = InsertOp   open-addressed insert/increment  [KmerSpectrum::append
 *                     src/KmerSpectrum.h:1578 + track() src/KmerTrackingData.h:427,517,641]
 *     Op = LinearOp   compacted linear records (kmr_partition.hpp); with owner_scatter_kernel the sender side of
 *                     the exchange [_buildKmerSpectrumMPI src/DistributedFunctions.h:418-438]
 *     Op = LookupOp   per-position count lookup [ReadSelector::setKmerValues
 *                     src/ReadSelector.h:1064-1076]
 *  insert_records_kernel      records -> table  [StoreKmerMessageHeaderProcessor::process :323]
 *  rehash_kernel              table growth
 *  classify/scatter/sort/image kernels: table -> bucketed sorted maps in the
 *        reference's store() layout [purgeMinDepth src/KmerSpectrum.h:1805,
 *        KmerMapByKmerArrayPair::store src/Kmer.h:3143, resort :3079]
 *  lookup_keys_kernel         packed keys -> counts [getElementIfExists src/Kmer.h:2617]
 *
 * Mapping to CDNA4: one read per lane (64 reads per wavefront) so the fp64
 * weight recurrence -- sequential per read, and it decides which k-mers are
 * counted -- runs in registers exactly as the reference computes it; each
 * wavefront stages the contiguous byte range of its 64 reads with coalesced
 * 16-byte loads into a private LDS tile and then walks it.  Rolling forward /
 * reverse-complement words make canonicalisation O(1) per base.  All global
 * traffic besides the staged input is the hash-table probe.
 */
#ifndef KMR_KERNELS_HPP_
#define KMR_KERNELS_HPP_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "kmr_key.hpp"

namespace kmr {

static const uint64_t EMPTY_KEY = ~0ull;
static const uint64_t NO_FIRST = ~0ull;

/* "First sighting" word of a table slot: the minimum, over the occurrences of a key, of
 *     ordinal (40 bits) << 24 | forward << 23 | q8 << 15 | r15
 * so the occurrence with the smallest stream ordinal wins and brings along what the reference knows about a k-mer's
 * first sighting, which lives in the singleton map until the second one arrives: no direction
 * (TrackingDataSingleton::getDirectionBias() == 0, src/KmerTrackingData.h:654) and its weight as the byte
 * q8 = (unsigned char)(w * 254) (:646); r15 = 15 more bits of w * 254, enough to take the exact weight out of the f64 sum
 * again (error <= 6e-8).  weightedCount of a promoted key is then (float)(q8 / 254.0) + the other weights (:658, :540-547). */
static const int FIRST_ORD_SHIFT = 24;
static const uint64_t MAX_STREAM_ORDINAL = (1ull << 40) - 1;
/* q8 << 15 | r15 = floor(w * 254 * 2^15) for 0 < w <= 1, in integers: w * 254 is exact in 32 bits (24-bit mantissa x 8 bits),
 * so this is the (unsigned char)(w * 254.0) of the reference's double arithmetic bit for bit, without fp64 instructions */
__host__ __device__ __forceinline__ uint32_t first_weight_bits(float w) {
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t bits = __float_as_uint(w);
#else
	uint32_t bits; memcpy(&bits, &w, 4);
#endif
	const uint32_t ex = (bits >> 23) & 0xffu;
	if (ex == 0 || ex > 127u) return ex > 127u ? (254u << 15) : 0u;      /* 0 and denormals; nothing above 1.0 reaches a spectrum */
	const uint32_t prod = ((bits & 0x7fffffu) | 0x800000u) * 254u;        /* mantissa * 254 < 2^32 */
	const uint32_t sh = 8u + (127u - ex);                                  /* w * 254 * 2^15 = prod * 2^(ex - 127 - 23 + 15) */
	return sh < 32u ? prod >> sh : 0u;
}
__host__ __device__ __forceinline__ unsigned long long first_pack(uint64_t ordinal, bool forward, float w) {
	return ((unsigned long long)ordinal << FIRST_ORD_SHIFT) | ((unsigned long long)(forward ? 1u : 0u) << 23) | (unsigned long long)(first_weight_bits(w) & 0x7fffffu);
}
__host__ __device__ __forceinline__ bool first_forward(unsigned long long f) { return ((f >> 23) & 1ull) != 0; }
/* what promotion from the singleton map does to the weight sum: (float)(q8 / 254.0) - w of the first sighting */
__host__ __device__ __forceinline__ double first_weight_shift(unsigned long long f) {
	const uint32_t q8 = (uint32_t)(f >> 15) & 0xffu, r15 = (uint32_t)f & 0x7fffu;
	const double w1 = ((double)q8 + ((double)r15 + 0.5) / 32768.0) / 254.0;
	return (double)(float)((double)q8 / 254.0) - w1;
}
static const int WAVES_PER_BLOCK = 4;
/* LDS bytes per wave for bases (same again for quals).  Two 4-wave blocks must fit the 160 KiB of a CU:
 * 2 * (8 * TILE_BUF + 2064 static) <= 163840, i.e. TILE_BUF <= 9982 -- 9984 left room for ONE block per CU (one wave
 * per SIMD, nothing to hide LDS latency behind).  9856 = 64 reads of up to 153 bases in one pass. */
#ifndef KMR_TILE_BUF
#define KMR_TILE_BUF 9856
#endif
static const int TILE_BUF = KMR_TILE_BUF;
static const int TILE_SPAN = TILE_BUF - 32;  /* max staged byte span of one pass */

enum { ERR_READ_TOO_LONG = 1, ERR_TABLE_FULL = 2, ERR_SEGMENT_OVERFLOW = 4 };

struct DevStats {
	unsigned long long raw, good, claimed;   /* claimed = new keys inserted */
	unsigned long long inserted;             /* records received through the owner exchange (holes excluded) */
	unsigned long long subtracted;           /* occurrences found in the subtracting reference spectrum */
	unsigned long long sender_bad;           /* sender side of the k-mer record exchange: k-mers of this rank's reads that were not good enough to send */
};

struct DevParams {
	uint32_t k, kb;
	float min_weight;
	uint32_t fastq_start, ext_min_q;
	uint32_t qzero;            /* raw quality chars below this have probability 0 (fastq_start + min_quality_score) */
	uint32_t subsample, rank, world, num_parts, part_idx;
	uint32_t count_sender_bad; /* the launch is the sender side of the k-mer record exchange (the owner counts what it receives, the sender what it drops) */
	const double *P;       /* 256 entries, device */
	DevStats *stats;
	uint32_t *err;
	/* KmerSpectrum::subtractReference (src/KmerSpectrum.h:472-474,1582-1588): finalized maps of another spectrum whose
	 * k-mers are skipped before they are counted; sub_wnb == 0 and sub_snb == 0 when there is none */
	const uint64_t *sub_wstart, *sub_wkeys; const uint32_t *sub_wvals; uint64_t sub_wnb; uint32_t sub_vw;
	const uint64_t *sub_sstart, *sub_skeys; const uint8_t *sub_sweight; uint64_t sub_snb;
};

struct ReadsView {
	const uint8_t *bases;
	const uint8_t *quals;          /* may be NULL */
	const uint64_t *offsets;       /* n_reads + 1 */
	const uint8_t *discarded;      /* may be NULL */
	uint64_t n_reads;
	uint64_t stream_base;          /* ordinal of byte 0 of this batch in the whole input */
	uint64_t first_read_idx;
	/* optional work units (reads longer than one LDS tile are cut into segments whose first k-mer index is a
	 * multiple of 1024, where buildWeightedKmers restarts its weight product anyway, src/KmerReadUtils.h:204):
	 * unit u covers bases [u_start[u], u_end[u]) of read u_read[u]; NULL = one unit per read */
	const uint64_t *u_start, *u_end, *u_read;
	uint64_t n_units;
};

static const uint32_t UNIT_KMERS = 9216;     /* k-mers per segment: 9 * 1024, and 9216 + 127 bases fit a tile */
#ifndef KMR_INSTANCE_TU
__global__ void unit_count_kernel(const uint64_t *offsets, uint64_t n, uint32_t k, uint32_t span, uint32_t *counts, unsigned int *maxLen) {
	unsigned int mx = 0;
	for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t L = offsets[r + 1] - offsets[r];
		mx = L > mx ? (unsigned int)(L > 0xffffffffull ? 0xffffffffu : L) : mx;
		counts[r] = L <= span ? 1u : (uint32_t)((L - k + 1 + UNIT_KMERS - 1) / UNIT_KMERS);
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { unsigned int o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
	/* one word serves ~90 M atomics/s: only a wavefront that would raise the maximum it can see touches it */
	if ((threadIdx.x & 63) == 0 && mx > __hip_atomic_load(maxLen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxLen, mx);
}
#endif
#ifndef KMR_INSTANCE_TU
__global__ void unit_fill_kernel(const uint64_t *offsets, uint64_t n, uint32_t k, uint32_t span, const uint64_t *ufirst,
                                 uint64_t *u_start, uint64_t *u_end, uint64_t *u_read) {
	for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t b = offsets[r], e = offsets[r + 1], L = e - b;
		const uint64_t u0 = ufirst[r], nu = ufirst[r + 1] - u0;
		if (L <= span) { u_start[u0] = b; u_end[u0] = e; u_read[u0] = r; continue; }
		for (uint64_t g = 0; g < nu; g++) {
			const uint64_t s0 = b + g * UNIT_KMERS, e0 = s0 + UNIT_KMERS + k - 1;
			u_start[u0 + g] = s0; u_end[u0 + g] = e0 < e ? e0 : e; u_read[u0 + g] = r;
		}
	}
}
#endif

/* ----------------------------------------------------------------------- */
/* device hash table: AoS slots so one probe touches one 32-byte sector      */
template <int W> struct Slot;
template <> struct Slot<1> {
	uint64_t key;              /* EMPTY_KEY when free                                 */
	unsigned long long cntfwd; /* low 32: occurrences, high 32: forward occurrences   */
	double wsum;               /* sum of weights                                      */
	unsigned long long first;  /* min over occurrences of first_pack(ordinal, forward, w) */
};
template <int W> struct Slot {
	uint64_t key[W];
	unsigned long long cntfwd;
	double wsum;
	unsigned long long first;
	uint32_t state;            /* 0 empty, 1 being written, 2 ready */
	uint32_t pad;
};
struct ExtSlot {               /* EXT mode side array, 64 B */
	uint32_t tally[12];        /* [Left,Right][A,C,G,T,N,X] */
	uint32_t pkt;              /* leftB | rightB<<8 | leftQ<<16 | rightQ<<24 of an occurrence */
	uint32_t pad[3];
};

template <int W> struct Table {
	Slot<W> *slots;
	ExtSlot *ext;              /* NULL unless EXT */
	uint32_t log2cap;
};

__device__ __forceinline__ uint64_t ld_relaxed(const uint64_t *p) {
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* find-or-claim the slot of 'key'; returns slot index or ~0 when the table is full.
 * 'claimed' is set when this lane inserted the key. */
template <int W>
__device__ __forceinline__ uint64_t table_find_or_insert(const Table<W> &t, const Key<W> &key, uint64_t hash, bool &claimed) {
	const uint64_t mask = (1ull << t.log2cap) - 1;
	uint64_t s = table_slot(hash, t.log2cap);
	claimed = false;
	uint64_t probes = 0;
	if constexpr (W == 1) {
		for (;;) {
			uint64_t cur = t.slots[s].key;
			if (cur == key.w[0]) return s;
			if (cur == EMPTY_KEY) {
				unsigned long long old = atomicCAS((unsigned long long *)&t.slots[s].key, (unsigned long long)EMPTY_KEY, (unsigned long long)key.w[0]);
				if (old == EMPTY_KEY) { claimed = true; return s; }
				if (old == key.w[0]) return s;
			}
			s = (s + 1) & mask;
			if (++probes > mask) return ~0ull;
		}
	} else {
		for (;;) {
			uint32_t st = __hip_atomic_load(&t.slots[s].state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
			if (st == 0) {
				uint32_t old = atomicCAS(&t.slots[s].state, 0u, 1u);
				if (old == 0) {
#pragma unroll
					for (int i = 0; i < W; i++) __hip_atomic_store(&t.slots[s].key[i], key.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(&t.slots[s].state, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
					claimed = true;
					return s;
				}
				st = old;
			}
			if (st == 1) continue;   /* the writer publishes without waiting on anyone: re-poll */
			bool eq = true;
#pragma unroll
			for (int i = 0; i < W; i++) eq = eq && (ld_relaxed(&t.slots[s].key[i]) == key.w[i]);
			if (eq) return s;
			s = (s + 1) & mask;
			if (++probes > mask) return ~0ull;
		}
	}
}
/* read-only probe; returns slot or ~0 */
template <int W>
__device__ __forceinline__ uint64_t table_find(const Table<W> &t, const Key<W> &key, uint64_t hash) {
	const uint64_t mask = (1ull << t.log2cap) - 1;
	uint64_t s = table_slot(hash, t.log2cap);
	for (uint64_t probes = 0; probes <= mask; probes++) {
		if constexpr (W == 1) {
			uint64_t cur = t.slots[s].key;
			if (cur == key.w[0]) return s;
			if (cur == EMPTY_KEY) return ~0ull;
		} else {
			if (t.slots[s].state == 0) return ~0ull;
			bool eq = true;
#pragma unroll
			for (int i = 0; i < W; i++) eq = eq && (t.slots[s].key[i] == key.w[i]);
			if (eq) return s;
		}
		s = (s + 1) & mask;
	}
	return ~0ull;
}
template <int W> __device__ __forceinline__ bool slot_used(const Slot<W> &s) {
	if constexpr (W == 1) return s.key != EMPTY_KEY; else return s.state == 2;
}
template <int W> __device__ __forceinline__ Key<W> slot_key(const Slot<W> &s) {
	Key<W> k;
	if constexpr (W == 1) k.w[0] = s.key; else {
#pragma unroll
		for (int i = 0; i < W; i++) k.w[i] = s.key[i];
	}
	return k;
}

template <int W> __global__ void table_clear_kernel(Slot<W> *slots, ExtSlot *ext, uint64_t cap) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
		Slot<W> s;
		if constexpr (W == 1) s.key = EMPTY_KEY; else {
#pragma unroll
			for (int j = 0; j < W; j++) s.key[j] = 0;
			s.state = 0; s.pad = 0;
		}
		s.cntfwd = 0; s.wsum = 0.0; s.first = NO_FIRST;
		slots[i] = s;
		if (ext) { ExtSlot e; for (int j = 0; j < 12; j++) e.tally[j] = 0; e.pkt = 0; e.pad[0] = e.pad[1] = e.pad[2] = 0; ext[i] = e; }
	}
}

/* one occurrence -> table (KmerSpectrum::append, cascade folded into counters) */
struct Occurrence {
	float w;            /* |weight| (> min_weight) */
	bool forward;       /* observed strand is the canonical one */
	uint64_t ordinal;   /* position of the occurrence in the input stream */
	uint32_t pkt;       /* extension packet (EXT) */
	int ltally, rtally; /* tally index 0..11 or -1 (EXT) */
};

template <int W, bool EXT>
__device__ __forceinline__ bool table_add(const Table<W> &t, const Key<W> &key, uint64_t hash, const Occurrence &o, unsigned &claimedCount) {
	bool claimed;
	uint64_t s = table_find_or_insert<W>(t, key, hash, claimed);
	if (s == ~0ull) return false;
	if (claimed) claimedCount++;
	Slot<W> *sl = &t.slots[s];
	atomicAdd(&sl->cntfwd, 1ull | ((unsigned long long)(o.forward ? 1 : 0) << 32));
	atomicAdd(&sl->wsum, (double)o.w);
	atomicMin(&sl->first, first_pack(o.ordinal, o.forward, o.w));
	if constexpr (EXT) {
		ExtSlot *e = &t.ext[s];
		if (o.ltally >= 0) atomicAdd(&e->tally[o.ltally], 1u);
		if (o.rtally >= 0) atomicAdd(&e->tally[o.rtally], 1u);
		e->pkt = o.pkt;      /* exact whenever the key ends with one occurrence, the only case it is read */
	}
	return true;
}

/* ----------------------------------------------------------------------- */
/* Ops                                                                       */
template <int W, bool EXT> struct InsertOp {
	Table<W> table;
	static const bool NEEDS_WEIGHT = true;
	static const bool COUNTS_STATS = true;
	static const bool NEEDS_HASH = true;
	struct State {};
	__device__ __forceinline__ void wave_begin(State &, int) const {}
	__device__ __forceinline__ void wave_end(State &, int) const {}
	__device__ __forceinline__ void tile_begin(State &, uint32_t *, uint64_t, int) const {}
	__device__ __forceinline__ void tile_end(State &, uint64_t, int) const {}
	__device__ __forceinline__ void emit(State &, bool valid, const DevParams &p, const Key<W> &key, uint64_t hash, const Occurrence &o,
	                                     uint64_t, uint32_t, unsigned &claimed, bool &fail) const {
		if (valid && !table_add<W, EXT>(table, key, hash, o, claimed)) fail = true;
	}
};

/* record layout: W key words, then f32 signed weight, u32 ext packet */
template <int W> struct Record {
	uint64_t key[W];
	float w;
	uint32_t pkt;
};
/* record of the streaming path with extension values: the packet AND the stream ordinal travel (24 bytes at k <= 32) */
template <int W> struct RecordX {
	uint64_t key[W];
	float w;
	uint32_t pkt;
	uint32_t ord;
	uint32_t pad;
};
template <int W, bool EXT> struct PoolRec { typedef Record<W> type; };
template <int W> struct PoolRec<W, true> { typedef RecordX<W> type; };
/* Record<W> carries the low 32 bits of the stream ordinal (the host refuses more than 2^32 input bases on that path),
 * RecordX<W> 40 bits (the high byte in its pad word) */
template <int W> __device__ __forceinline__ uint64_t rec_ordinal(const Record<W> &r) { return r.pkt; }
template <int W> __device__ __forceinline__ uint64_t rec_ordinal(const RecordX<W> &r) { return (uint64_t)r.ord | ((uint64_t)r.pad << 32); }
template <int W> __device__ __forceinline__ void rec_set(Record<W> &r, uint32_t pkt, uint64_t ord) { r.pkt = (uint32_t)ord; (void)pkt; }
template <int W> __device__ __forceinline__ void rec_set(RecordX<W> &r, uint32_t pkt, uint64_t ord) { r.pkt = pkt; r.ord = (uint32_t)ord; r.pad = (uint32_t)(ord >> 32); }
/* ExtensionTracking::trackExtension on a packet (src/KmerTrackingData.h:195-201): tally index 0..11 or -1 */
__device__ __forceinline__ int ext_tally_index(uint32_t ch) { switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'X': return 5; default: return 4; } }

/* ----------------------------------------------------------------------- */
/* finalized map on the device: bucketed, keys sorted inside each bucket      */
template <int W> struct MapView {
	const uint64_t *start;     /* [nb+1] entry index of each bucket          */
	const uint64_t *keys;      /* [n][W]                                     */
	const uint32_t *vals;      /* weak: [n][VW] words in the value layout    */
	const uint8_t *sweight;    /* singleton: [n] _weight byte                */
	uint64_t nb;               /* power of two, 0 = map absent               */
	uint32_t vw;               /* value words per weak entry (3 or 15)       */
};

template <int W> __device__ __forceinline__ int64_t map_find(const MapView<W> &m, const Key<W> &key, uint64_t hash) {
	if (m.nb == 0) return -1;
	const uint64_t b = hash & (m.nb - 1);
	uint64_t lo = m.start[b], hi = m.start[b + 1];
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		Key<W> km;
#pragma unroll
		for (int i = 0; i < W; i++) km.w[i] = m.keys[mid * W + i];
		if (key_eq<W>(km, key)) return (int64_t)mid;
		if (key_lt<W>(km, key)) lo = mid + 1; else hi = mid;
	}
	return -1;
}
/* DataPointers::getCount(false), src/KmerSpectrum.h:642-695 */
template <int W> __device__ __forceinline__ uint32_t maps_count(const MapView<W> &weak, const MapView<W> &sing, const Key<W> &key, uint64_t hash) {
	int64_t i = map_find<W>(weak, key, hash);
	if (i >= 0) return weak.vals[(uint64_t)i * weak.vw] & 0xffffu;
	i = map_find<W>(sing, key, hash);
	if (i >= 0) return sing.sweight[i] == 0 ? 0u : 1u;
	return 0u;
}

/* ----------------------------------------------------------------------- */
/* compressBase (src/TwoBitSequence.cpp:124-147) without branches: 0..3 for ACGT/acgt, 4 = markup.
 * (c>>1)&3 maps A,C,G,T to 0,1,3,2; x^(x>>1) turns that into 0,1,2,3. */
__device__ __forceinline__ uint32_t base_code(uint8_t c) {
	const uint32_t u = (uint32_t)c & 0xDFu;                  /* upper case */
	const uint32_t idx = u - 'A';
	const uint32_t valid = idx < 32u ? ((0x00080045u >> idx) & 1u) : 0u;   /* bits A=0, C=2, G=6, T=19 */
	const uint32_t x = (u >> 1) & 3u;
	return valid ? (x ^ (x >> 1)) : 4u;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v;
}

template <int W, bool EXT, class Op, bool SUB = false>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, 2)
void extract_kernel(ReadsView rv, DevParams p, Op op) {
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	__shared__ double sP[256];
	__shared__ uint32_t s_wcount[WAVES_PER_BLOCK];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	for (int i = threadIdx.x; i < 256; i += blockDim.x) sP[i] = p.P[i];
	__syncthreads();                       /* the only block-wide barrier; waves are independent below */

	uint8_t *tb = smem + (size_t)wave * (2 * TILE_BUF);
	uint8_t *tq = tb + TILE_BUF;
	const uint32_t k = p.k;
	unsigned long long nRaw = 0, nGood = 0;
	unsigned nClaimed = 0;
	bool fail = false;
	typename Op::State opst;
	op.wave_begin(opst, lane);
	constexpr bool haveSub = SUB;       /* a subtracting reference spectrum is set: its own instantiation, the usual build pays nothing */
	const bool needHash = Op::NEEDS_HASH || p.subsample > 1 || p.world > 1 || p.num_parts > 1 || haveSub;
	unsigned long long nSub = 0;
	/* a wavefront walks tiles tile0, tile0 + stride, ... (stride = all wavefronts of the grid): with a full grid
	 * that is one tile each */
	const uint64_t n_items = rv.u_start ? rv.n_units : rv.n_reads;
	const uint64_t n_tiles = (n_items + 63) / 64;
	for (uint64_t tile = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * WAVES_PER_BLOCK) {
	const uint64_t r0 = tile * 64;
	const uint32_t nr = (uint32_t)((n_items - r0) < 64 ? (n_items - r0) : 64);
	const bool have = (uint32_t)lane < nr;
	uint64_t myStart = 0, myEnd = 0, myRead = 0;
	uint64_t myReadEnd = 0;            /* units only: end of the whole read (a segment's outer extensions lie outside it) */
	uint32_t kfirst = 0;               /* index in the read of the unit's first k-mer */
	bool myDiscard = true, myRefQual = false;
	if (have) {
		if (rv.u_start) {
			myStart = rv.u_start[r0 + lane]; myEnd = rv.u_end[r0 + lane]; myRead = rv.u_read[r0 + lane];
			const uint64_t rs = rv.offsets[myRead];
			kfirst = (uint32_t)(myStart - rs);
			myReadEnd = rv.offsets[myRead + 1];
			myRefQual = rv.quals && rv.offsets[myRead + 1] > rs && rv.quals[rs] == 127;
		} else {
			myRead = r0 + lane;
			myStart = rv.offsets[myRead];
			myEnd = rv.offsets[myRead + 1];
		}
		myDiscard = rv.discarded ? (rv.discarded[myRead] != 0) : false;
	}
	op.tile_begin(opst, &s_wcount[wave], r0, lane);
	uint32_t tRaw = 0, tGood = 0;      /* per lane and tile: at most 64 units of < 2^14 positions */

	uint32_t done = 0;
	while (done < nr) {
		const uint64_t B0 = __shfl(myStart, (int)done, 64);
		const bool fits = have && (uint32_t)lane >= done && (myEnd - B0 <= (uint64_t)TILE_SPAN);
		unsigned long long m = __ballot(fits) >> done;
		uint32_t n = ~m ? (uint32_t)__builtin_ctzll(~m) : 64u;          /* run of fitting reads starting at 'done' (ctz of 0 is undefined) */
		if (n > nr - done) n = nr - done;
		if (n == 0) {                                        /* read longer than a tile */
			if (lane == 0) atomicOr(p.err, (uint32_t)ERR_READ_TOO_LONG);
			done += 1;
			continue;
		}
		const uint64_t B1 = __shfl(myEnd, (int)(done + n - 1), 64);
		/* stage [B0,B1) with 16-byte loads from the enclosing aligned range */
		const uintptr_t gb = (uintptr_t)rv.bases + B0, gq = (uintptr_t)rv.quals + B0;
		const uintptr_t ab = gb & ~(uintptr_t)15, aq = gq & ~(uintptr_t)15;
		const uint32_t nb16 = (uint32_t)(((uintptr_t)rv.bases + B1 - ab + 15) >> 4);
		/* LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land at a wave-uniform LDS base + lane * 16): every KB of the
		 * tile is requested before the first one is waited for, and no registers are held meanwhile (a load-wait-store
		 * loop pays the HBM latency once per KB) */
		{
			typedef __attribute__((address_space(1))) const void *gptr_t;
			typedef __attribute__((address_space(3))) void *lptr_t;
			constexpr int STG = (TILE_BUF / 16 + 63) / 64;
			const uint4 *gbp = (const uint4 *)(rv.bases + (ptrdiff_t)(ab - (uintptr_t)rv.bases));
			const uint4 *gqp = (const uint4 *)(rv.quals + (ptrdiff_t)(aq - (uintptr_t)rv.quals));
			const uint32_t nq16 = rv.quals ? (uint32_t)(((uintptr_t)rv.quals + B1 - aq + 15) >> 4) : 0u;
#pragma unroll
			for (int c = 0; c < STG; c++) {
				const uint32_t idx = (uint32_t)lane + 64u * c;
				if (64u * c < nb16 && idx < nb16) __builtin_amdgcn_global_load_lds((gptr_t)(gbp + idx), (lptr_t)(tb + 1024 * c), 16, 0, 0);
			}
#pragma unroll
			for (int c = 0; c < STG; c++) {
				const uint32_t idx = (uint32_t)lane + 64u * c;
				if (64u * c < nq16 && idx < nq16) __builtin_amdgcn_global_load_lds((gptr_t)(gqp + idx), (lptr_t)(tq + 1024 * c), 16, 0, 0);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

		const bool active = have && (uint32_t)lane >= done && (uint32_t)lane < done + n && !myDiscard;
		const uint32_t L = active ? (uint32_t)(myEnd - myStart) : 0;
		const uint8_t *rb = tb + (uint32_t)(gb - ab) + (uint32_t)(myStart - B0);
		const uint8_t *rq = tq + (uint32_t)(gq - aq) + (uint32_t)(myStart - B0);
		uint32_t Lmax = L;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(Lmax, off, 64); Lmax = o > Lmax ? o : Lmax; }

		/* Read::REF_QUAL in the read's first quality char (the unit's own first char when units are whole reads) */
		const bool isRef = (rv.quals == nullptr) || (rv.u_start ? myRefQual : (L > 0 && rq[0] == 127));
		Roller<W> roll;
		roll.init(k);
		double w = 0.0;
		uint32_t zc = 0;                 /* positions in the window that force weight 0 */
		uint64_t zbits[3] = {0, 0, 0};   /* zero flag of the last positions (bit i = position j-i), so the flag that
		                                    leaves the window is read from a register instead of the tile */
		uint32_t qprev = 0x100, qrun = 0; /* length of the current run of equal quality chars */
		uint32_t leftCode = 5, leftQ = p.ext_min_q;   /* Extension('X', minQuality) */
		if (EXT && active && kfirst > 0) {            /* a later segment of a long read: its left neighbour is in global memory */
			leftCode = base_code(rv.bases[myStart - 1]); if (leftCode == 4) leftCode = 0;
			leftQ = ((isRef ? 127u : (uint32_t)rv.quals[myStart - 1]) - p.fastq_start) & 0xffu;
		}
		const uint32_t rbOff = (uint32_t)(rb - tb), rqOff = (uint32_t)(rq - tq);
		const bool haveQuals = rv.quals != nullptr;
		/* wave-uniform: the steady-state group below serves builds (not lookups) without extension packets and without a
		 * subsample / part / owner / subtraction filter */
		const bool fastOK = Op::NEEDS_WEIGHT && !EXT && !SUB && haveQuals && p.subsample <= 1 && p.num_parts <= 1 && (p.world <= 1 || op_keeps_all_owners(op));
		for (uint32_t jb = 0; jb < Lmax; jb += 4) {
		/* bases and quals of the next four positions: one aligned 8-byte LDS window each per four iterations
		 * (the tile buffers are 16-byte aligned and 32 bytes longer than the staged span) */
		uint64_t bwin = 0, qwin = 0;
		if (jb < L) {
			const uint32_t a = rbOff + jb;
			const uint32_t *pw = (const uint32_t *)(tb + (a & ~3u));
			bwin = ((uint64_t)pw[0] | ((uint64_t)pw[1] << 32)) >> (8 * (a & 3u));
			if (haveQuals) {
				const uint32_t aq_ = rqOff + jb;
				const uint32_t *pq = (const uint32_t *)(tq + (aq_ & ~3u));
				qwin = ((uint64_t)pq[0] | ((uint64_t)pq[1] << 32)) >> (8 * (aq_ & 3u));
			}
		}
		/* Steady-state group: every lane that still has bases is past its first k-mer, has four more positions, an
		 * all-clear window (no N, no zero-probability quality: zc == 0 and none among the four new positions), a live
		 * weight chain (w != 0) and no restart of the product inside the group.  Then the four positions need none of
		 * the per-position case analysis below -- the same arithmetic in the same order, without the branches (the
		 * general path spends about as many scalar exec-mask instructions as vector ones). */
		if (fastOK && jb >= k && ((jb + 1 - k) & 1023u) != 0 && ((jb + 1 - k) & 1023u) <= 1020u) {
			const bool live = jb < L;
			uint32_t fc[4], fq[4];
			bool ok = !live || (jb + 4 <= L && !isRef && zc == 0 && w != 0.0);
#pragma unroll
			for (int ju = 0; ju < 4; ju++) {
				fc[ju] = base_code((uint8_t)(bwin >> (8 * ju)));
				fq[ju] = (uint32_t)((qwin >> (8 * ju)) & 0xffu);
				ok = ok && !(live && (fc[ju] == 4 || fq[ju] < p.qzero));
			}
			if (__all(ok)) {
				zbits[2] = (zbits[2] << 4) | (zbits[1] >> 60); zbits[1] = (zbits[1] << 4) | (zbits[0] >> 60); zbits[0] <<= 4;
				if (live) tRaw += 4;
				float wf = (float)w;          /* narrowed again only when the chain moves */
#pragma unroll
				for (uint32_t ju = 0; ju < 4; ju++) {
					const uint32_t i = jb + ju + 1 - k;
					const uint32_t q = fq[ju];
					qrun = (q == qprev) ? qrun + 1 : 0; qprev = q;
					roll.push(fc[ju] & 3u);
					if (live && qrun < k) {                   /* same update as the general path below */
						const uint32_t qo = rq[i - 1];
						if (qo != q) { const double change = sP[q] / sP[qo]; w *= change; wf = (float)w; }
					}
					const Key<W> kf = roll.getFwd(), kr = roll.getRc();
					const bool isLeast = key_le<W>(kf, kr);
					const Key<W> canon = isLeast ? kf : kr;
					const bool valid = live && wf > p.min_weight;
					if (valid) tGood++;
					Occurrence o;
					o.w = wf; o.forward = isLeast; o.ordinal = rv.stream_base + myStart + i; o.pkt = 0; o.ltally = -1; o.rtally = -1;
					const uint64_t hash = Op::NEEDS_HASH ? key_hash<W>(canon, p.kb) : 0ull;
					op.emit(opst, valid, p, canon, hash, o, rv.first_read_idx + myRead, kfirst + i, nClaimed, fail);
				}
				continue;
			}
		}
#pragma unroll
		for (uint32_t ju = 0; ju < 4; ju++) {
			const uint32_t j = jb + ju;
			if (j >= Lmax) break;
			/* the occurrence of this iteration is handed to the Op after the divergent part, so that
			 * Ops can use wave-wide primitives (ballot compaction) under uniform control flow */
			bool valid = false;
			Key<W> canon;
			uint64_t hash = 0;
			uint32_t kpos = 0;
			Occurrence o;
			o.w = 0.0f; o.forward = true; o.ordinal = 0; o.pkt = 0; o.ltally = -1; o.rtally = -1;
#pragma unroll
			for (int wi = 0; wi < W; wi++) canon.w[wi] = 0;
			if (j < L) {
				const uint8_t c = (uint8_t)(bwin >> (8 * ju));
				uint32_t code = base_code(c);
				const uint32_t q = isRef ? 127u : (uint32_t)((qwin >> (8 * ju)) & 0xffu);
				bool z = (code == 4) || (!isRef && q < p.qzero);
				if (code == 4) code = 0;                      /* markup packs as A */
				zbits[2] = (zbits[2] << 1) | (zbits[1] >> 63); zbits[1] = (zbits[1] << 1) | (zbits[0] >> 63); zbits[0] = (zbits[0] << 1) | (z ? 1ull : 0ull);
				zc += z ? 1u : 0u;
				zc -= (uint32_t)((zbits[k >> 6] >> (k & 63)) & 1ull);   /* position j-k leaves the window (0 while j < k) */
				qrun = (q == qprev) ? qrun + 1 : 0; qprev = q;
				roll.push(code);
				if (j + 1 >= k) {
					const uint32_t i = j + 1 - k;
					if (Op::NEEDS_WEIGHT) {
						if (zc > 0) w = 0.0;
						else if (isRef) w = 1.0;
						else if ((i & 1023u) == 0 || w == 0.0) {
							w = 1.0;
							for (uint32_t jj = 0; jj < k; jj++) w *= sP[rq[i + jj]];
						} else {
							/* x / x == 1.0 exactly, so equal qualities leave w unchanged; a run of k+1 equal
							 * chars proves that without touching the tile */
							if (qrun < k) {
								const uint32_t qo = rq[i - 1];
								if (qo != q) {
									const double change = sP[q] / sP[qo];
									w *= change;
								}
							}
						}
					}
					const Key<W> kf = roll.getFwd(), kr = roll.getRc();
					const bool isLeast = key_le<W>(kf, kr);
					canon = isLeast ? kf : kr;
					hash = needHash ? key_hash<W>(canon, p.kb) : 0ull;
					kpos = kfirst + i;
					bool mine = true;
					if (Op::NEEDS_WEIGHT) {
						if (p.subsample > 1 && hash % p.subsample != 0) mine = false;   /* owner / part filters apply to the build, not to lookups */
						if (p.world > 1 && !op_keeps_all_owners(op) && distributed_thread_id(hash, p.world) != p.rank) mine = false;
						if (p.num_parts > 1 && distributed_thread_id(hash, p.num_parts) != p.part_idx) mine = false;
						if (haveSub && mine) {          /* subtractingReference->exists(least): skipped before rawKmers++ */
							MapView<W> sw, ss;
							sw.start = p.sub_wstart; sw.keys = p.sub_wkeys; sw.vals = p.sub_wvals; sw.sweight = nullptr; sw.nb = p.sub_wnb; sw.vw = p.sub_vw;
							ss.start = p.sub_sstart; ss.keys = p.sub_skeys; ss.vals = nullptr; ss.sweight = p.sub_sweight; ss.nb = p.sub_snb; ss.vw = 0;
							if (maps_count<W>(sw, ss, canon, hash) > 0) { mine = false; nSub++; }
						}
					}
					if (mine) {
						const float wf = (float)w;
						tRaw++;
						if (!Op::NEEDS_WEIGHT || wf > p.min_weight) {
							tGood++;
							valid = true;
							o.w = wf; o.forward = isLeast; o.ordinal = rv.stream_base + myStart + i;
							if (EXT) {
								uint32_t rc_, rq_;
								if (j + 1 < L) { rc_ = base_code(rb[j + 1]); if (rc_ == 4) rc_ = 0; rq_ = ((isRef ? 127u : (uint32_t)rq[j + 1]) - p.fastq_start) & 0xffu; }
								else if (rv.u_start && myEnd < myReadEnd) { rc_ = base_code(rv.bases[myEnd]); if (rc_ == 4) rc_ = 0; rq_ = ((isRef ? 127u : (uint32_t)rv.quals[myEnd]) - p.fastq_start) & 0xffu; }
								else { rc_ = 5; rq_ = p.ext_min_q; }
								uint32_t lc = leftCode, lq = leftQ;
								if (!isLeast) {          /* swap and complement (KmerReadUtils.h:232-235) */
									const uint32_t tl = rc_ < 4 ? 3 - rc_ : rc_, tr = lc < 4 ? 3 - lc : lc;
									const uint32_t tq_ = rq_; rq_ = lq; lq = tq_;
									lc = tl; rc_ = tr;
								}
								const char dec[6] = {'A', 'C', 'G', 'T', 'N', 'X'};
								o.pkt = (uint32_t)(uint8_t)dec[lc] | ((uint32_t)(uint8_t)dec[rc_] << 8) | (lq << 16) | (rq_ << 24);
								if (lq >= p.ext_min_q || lc > 3) o.ltally = (int)lc;
								if (rq_ >= p.ext_min_q || rc_ > 3) o.rtally = 6 + (int)rc_;
							}
						}
					}
					if (EXT) { leftCode = base_code(rb[i]); if (leftCode == 4) leftCode = 0; leftQ = ((isRef ? 127u : (uint32_t)rq[i]) - p.fastq_start) & 0xffu; }
				}
			}
			op.emit(opst, valid, p, canon, hash, o, rv.first_read_idx + myRead, kpos, nClaimed, fail);
		}
		}
		done += n;
		__builtin_amdgcn_wave_barrier();   /* all lanes are done reading the tile before it is overwritten */
	}
	op.tile_end(opst, tile, lane);
	nRaw += tRaw; nGood += tGood;
	}
	op.wave_end(opst, lane);
	nRaw = wave_sum(nRaw); nGood = wave_sum(nGood);
	unsigned long long nc = wave_sum((unsigned long long)nClaimed);
	if (lane == 0 && Op::COUNTS_STATS) {   /* the owner counts records when they are inserted */
		atomicAdd(&p.stats->raw, nRaw);
		atomicAdd(&p.stats->good, nGood);
		if (nc) atomicAdd(&p.stats->claimed, nc);
	}
	if (lane == 0 && !Op::COUNTS_STATS && p.count_sender_bad && nRaw > nGood) atomicAdd(&p.stats->sender_bad, nRaw - nGood);
	if (haveSub) { nSub = wave_sum(nSub); if (lane == 0 && nSub) atomicAdd(&p.stats->subtracted, nSub); }
	if (__any(fail) && lane == 0) atomicOr(p.err, op_fail_code(op));
}

template <int W, bool EXT> __device__ __forceinline__ bool op_keeps_all_owners(const InsertOp<W, EXT> &) { return false; }
template <int W, bool EXT> __device__ __forceinline__ uint32_t op_fail_code(const InsertOp<W, EXT> &) { return ERR_TABLE_FULL; }

/* Lookup accelerator for read scoring: the weak map once more as an open-addressed table, slot = {key words, count}
 * ((W + 1) u64), at most half full, slot index from the same lookup3 hash.  A sorted bucket costs ~7 dependent loads per
 * k-mer (two bucket bounds, the binary search, the value); a slot of this table is one 16-byte load at k <= 32.  Built on the
 * first scoring call after a finalize (lut_build_kernel), dropped when the map changes. */
template <int W> struct LutView { const uint64_t *slots; uint64_t mask; uint32_t shift; };
__host__ __device__ __forceinline__ uint64_t lut_slot(uint64_t hash, uint32_t shift) { return (hash * 0x9E3779B97F4A7C15ull) >> shift; }
/* The table marks a free slot by EMPTY_KEY in its first word.  A one-word canonical key is never all ones (all-T is not
 * canonical), but the first word of a longer key can be (k >= 64: T^32...A^32 is its own reverse complement): such keys
 * are not put into the table, lookups of them search the sorted buckets. */
template <int W> __device__ __forceinline__ bool lut_holds(const Key<W> &key) { return W == 1 || key.w[0] != EMPTY_KEY; }
template <int W> __device__ __forceinline__ uint32_t lut_count(const LutView<W> &t, const Key<W> &key, uint64_t hash) {
	uint64_t s = lut_slot(hash, t.shift);
	for (;;) {
		const uint64_t *p = t.slots + s * (W + 1);
		if (W == 1) {
			const ulonglong2 v = *(const ulonglong2 *)p;
			if (v.x == key.w[0]) return (uint32_t)v.y;
			if (v.x == EMPTY_KEY) return 0u;
		} else {
			const uint64_t k0 = p[0];
			if (k0 == EMPTY_KEY) return 0u;
			bool eq = k0 == key.w[0];
#pragma unroll
			for (int i = 1; i < W; i++) eq = eq && p[i] == key.w[i];
			if (eq) return (uint32_t)p[W];
		}
		s = (s + 1) & t.mask;
	}
}
template <int W>
__global__ void lut_build_kernel(MapView<W> weak, uint64_t n, uint64_t *slots, uint64_t mask, uint32_t shift, uint32_t kb) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		Key<W> key;
#pragma unroll
		for (int j = 0; j < W; j++) key.w[j] = weak.keys[i * W + j];
		if (!lut_holds<W>(key)) continue;
		uint64_t s = lut_slot(key_hash<W>(key, kb), shift);
		for (;;) {           /* the keys of a map are distinct: a slot is taken by whoever swaps its first word in */
			unsigned long long *p = (unsigned long long *)(slots + s * (W + 1));
			if (atomicCAS(p, (unsigned long long)EMPTY_KEY, (unsigned long long)key.w[0]) == (unsigned long long)EMPTY_KEY) {
#pragma unroll
				for (int j = 1; j < W; j++) p[j] = key.w[j];
				p[W] = weak.vals[i * weak.vw] & 0xffffu;
				break;
			}
			s = (s + 1) & mask;
		}
	}
}
#ifndef KMR_INSTANCE_TU
__global__ void lut_clear_kernel(uint64_t *slots, uint64_t n_slots, uint32_t stride) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * blockDim.x) slots[i * stride] = EMPTY_KEY;
}
#endif

template <int W> struct LookupOp {
	LutView<W> lut;                /* slots == nullptr: search the sorted buckets */
	MapView<W> weak, sing;
	uint32_t *out;
	const uint64_t *out_offsets;   /* per read, indexed by global read index - first_read_idx */
	uint64_t first_read_idx;
	bool weak_only;                /* ReadSelector looks at spectrum.weak only (src/ReadSelector.h:924-931) */
	static const bool NEEDS_WEIGHT = false;
	static const bool COUNTS_STATS = false;
	static const bool NEEDS_HASH = true;
	struct State {};
	__device__ __forceinline__ void wave_begin(State &, int) const {}
	__device__ __forceinline__ void wave_end(State &, int) const {}
	__device__ __forceinline__ void tile_begin(State &, uint32_t *, uint64_t, int) const {}
	__device__ __forceinline__ void tile_end(State &, uint64_t, int) const {}
	__device__ __forceinline__ void emit(State &, bool valid, const DevParams &, const Key<W> &key, uint64_t hash, const Occurrence &,
	                                     uint64_t readIdx, uint32_t pos, unsigned &, bool &) const {
		if (valid) {
			uint32_t c;
			if (weak_only && lut.slots && lut_holds<W>(key)) c = lut_count<W>(lut, key, hash);
			else if (weak_only) { const int64_t i = map_find<W>(weak, key, hash); c = i >= 0 ? (weak.vals[(uint64_t)i * weak.vw] & 0xffffu) : 0u; }
			else c = maps_count<W>(weak, sing, key, hash);
			out[out_offsets[readIdx - first_read_idx] + pos] = c;
		}
	}
};
template <int W> __device__ __forceinline__ bool op_keeps_all_owners(const LookupOp<W> &) { return true; }
template <int W> __device__ __forceinline__ uint32_t op_fail_code(const LookupOp<W> &) { return 0; }

template <int W>
__global__ void lookup_keys_kernel(MapView<W> weak, MapView<W> sing, const uint8_t *packed, uint64_t n, uint32_t kb, uint32_t *out) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		Key<W> key;
		key_from_bytes<W>(key, packed + i * (kb & 0xffffu), kb & 0xffffu);      /* (kb carries the hash kind above bit 16) */
		out[i] = maps_count<W>(weak, sing, key, key_hash<W>(key, kb));
	}
}

/* owner side of a lookup request: key words as they travel (Key<W>::w), weak map only as ReadSelector::getValue
 * (src/ReadSelector.h:924-931; processRequest, src/DistributedFunctions.h:857-866) */
template <int W>
__global__ void lookup_words_kernel(MapView<W> weak, LutView<W> lut, const uint64_t *keys, uint64_t n, uint32_t kb, uint32_t *out) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		Key<W> key;
#pragma unroll
		for (int j = 0; j < W; j++) key.w[j] = keys[i * W + j];
		const uint64_t hash = key_hash<W>(key, kb);
		if (lut.slots && lut_holds<W>(key)) { out[i] = lut_count<W>(lut, key, hash); continue; }
		const int64_t e = map_find<W>(weak, key, hash);
		out[i] = e >= 0 ? (weak.vals[(uint64_t)e * weak.vw] & 0xffffu) : 0u;
	}
}
/* processRespond (:848-854): the answer to request i belongs to the k-mer at position pos[i] of the read batch */
#ifndef KMR_INSTANCE_TU
__global__ void scatter_counts_kernel(const uint32_t *counts, const uint32_t *pos, uint64_t n, uint32_t *position_counts) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) position_counts[pos[i]] = counts[i];
}
#endif

/* ReadSelector::scoreAndTrimReads for one read per lane (src/ReadSelector.h:1182-1207): numKmers is cut at the
 * first N/X markup (_setNumKmers :1037-1047), the read is trimmed to the first longest run of k-mers whose count
 * is >= minScore (trimReadByMinimumKmerScore :949-1014, bimodal detection off), the run is scored
 * (scoreReadByScoringType :1094-1180) and setTrimHeaders (:1015-1036) converts the run to bases. */
enum { SCORE_SUM = 0, SCORE_MEDIAN = 1, SCORE_MIN = 2, SCORE_MAX = 3, SCORE_AVG = 4 };
/* one read: numKmers is already cut at the first markup; cv = its counts (LDS copy as u16, or global u32) */
template <typename T>
__device__ __forceinline__ void score_one_read(const T *cv, uint32_t numKmers, uint32_t k, float minScore, int scoring,
                                               uint32_t &tOff, uint32_t &tLen, float &scOut, bool &trimmedOut) {
	uint32_t bestOff = 0, bestLen = 0, off = 0, len = 0;
	for (uint32_t i = 0; i < numKmers; i++) {
		const float v = (float)(uint32_t)cv[i];
		if (v >= minScore) len++;
		else { if (len > bestLen) { bestLen = len; bestOff = off; } off += len + 1; len = 0; }
	}
	if (len > bestLen) { bestLen = len; bestOff = off; }
	trimmedOut = bestLen < numKmers;
	float sc = -1.0f;
	if (bestLen > 0) {
		const T *run = cv + bestOff;
		if (scoring == SCORE_MEDIAN) {
			/* sorted[n/2]: the smallest v with #(x <= v) > n/2, found by bisection on the 16-bit count */
			const uint32_t t = bestLen / 2;
			uint32_t lo = 65535, hi = 0;          /* the bisection starts from the run's own range, not from 16 bits */
			for (uint32_t i = 0; i < bestLen; i++) { const uint32_t v = run[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
			while (lo < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				uint32_t le = 0;
				for (uint32_t i = 0; i < bestLen; i++) le += (uint32_t)run[i] <= mid ? 1u : 0u;
				if (le > t) hi = mid; else lo = mid + 1;
			}
			sc = (float)lo;
		} else if (scoring == SCORE_AVG) {
			double sum = 0.0; for (uint32_t i = 0; i < bestLen; i++) sum += (float)(uint32_t)run[i];
			sc = (float)(sum / (double)bestLen);
		} else if (scoring == SCORE_MIN || scoring == SCORE_MAX) {
			uint32_t m = run[0];
			for (uint32_t i = 1; i < bestLen; i++) { const uint32_t v = run[i]; m = scoring == SCORE_MAX ? (v > m ? v : m) : (v < m ? v : m); }
			sc = (float)m;
		} else sc = 0.0f;     /* KS_SUM: scoreReadBySumKmer only assigns the score when byAvg (:1149-1162) */
	}
	tOff = bestLen > 0 ? bestOff : 0u;
	tLen = bestLen > 0 ? bestLen + k - 1 : 0u;
	scOut = sc;
}

/* A wavefront takes 64 consecutive reads: their bases are one contiguous range, searched for markups with coalesced 16-byte
 * loads (a hit finds its read by bisection over the 65 offsets in LDS -- hits are rare), their counts are one contiguous
 * range too (count_off is an exclusive scan, or the base offsets themselves), copied into LDS as u16; then every lane walks
 * its own read in LDS.  Groups whose counts do not fit the LDS window (long reads) walk them in global memory. */
static const int SC_WAVES = 3, SC_CAP = 10240;
#ifndef KMR_INSTANCE_TU
__global__ __launch_bounds__(SC_WAVES * 64)
void score_reads_kernel(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, uint32_t k, const uint32_t *counts,
                        const uint64_t *count_off, float minScore, int scoring, uint32_t *trimOffset, uint32_t *trimLength,
                        float *score, uint8_t *wasTrimmed) {
	__shared__ uint16_t s_cnt[SC_WAVES][SC_CAP];
	__shared__ uint64_t s_off[SC_WAVES][65], s_coff[SC_WAVES][65];
	__shared__ uint32_t s_first[SC_WAVES][64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint64_t groups = (n_reads + 63) / 64;
	for (uint64_t g = (uint64_t)blockIdx.x * SC_WAVES + wave; g < groups; g += (uint64_t)gridDim.x * SC_WAVES) {
		const uint64_t r0 = g * 64;
		const uint32_t nr = (uint32_t)(n_reads - r0 < 64 ? n_reads - r0 : 64);
		__builtin_amdgcn_wave_barrier();
		for (uint32_t i = lane; i <= nr; i += 64) { s_off[wave][i] = offsets[r0 + i]; s_coff[wave][i] = count_off[r0 + i]; }
		s_first[wave][lane] = 0xffffffffu;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
		const uint64_t b0 = s_off[wave][0], b1 = s_off[wave][nr];
		for (uint64_t q = (b0 & ~15ull) + (uint64_t)lane * 16; q < b1; q += 1024) {        /* TwoBitSequence::firstMarkupNorX */
			const uint4 v = *(const uint4 *)(bases + q);
			const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int j = 0; j < 16; j++) {
				const uint8_t c = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
				const uint64_t pp = q + j;
				if ((c == 'N' || c == 'X' || c == '.') && pp >= b0 && pp < b1) {
					uint32_t lo = 0, hi = nr;                     /* the read with off[l] <= pp < off[l+1] */
					while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_off[wave][mid] <= pp) lo = mid; else hi = mid; }
					atomicMin(&s_first[wave][lo], (uint32_t)(pp - s_off[wave][lo]));
				}
			}
		}
		const uint64_t c0 = s_coff[wave][0], cn = s_coff[wave][nr] - c0;
		const bool staged = cn <= (uint64_t)SC_CAP;
		if (staged) for (uint64_t i = lane; i < cn; i += 64) s_cnt[wave][i] = (uint16_t)counts[c0 + i];
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
		if ((uint32_t)lane < nr) {
			const uint64_t L = s_off[wave][lane + 1] - s_off[wave][lane];
			uint32_t numKmers = L >= k ? (uint32_t)(L - k + 1) : 0;
			const uint32_t first = s_first[wave][lane];
			if (first != 0xffffffffu) { const uint32_t m = first + 1; numKmers = m > k ? (m - k < numKmers ? m - k : numKmers) : 0; }
			uint32_t tOff, tLen; float sc; bool trimmed;
			if (staged) score_one_read<uint16_t>(&s_cnt[wave][s_coff[wave][lane] - c0], numKmers, k, minScore, scoring, tOff, tLen, sc, trimmed);
			else score_one_read<uint32_t>(counts + s_coff[wave][lane], numKmers, k, minScore, scoring, tOff, tLen, sc, trimmed);
			const uint64_t r = r0 + lane;
			trimOffset[r] = tOff; trimLength[r] = tLen; score[r] = sc; wasTrimmed[r] = trimmed ? 1 : 0;
		}
	}
}
#endif

/* ----------------------------------------------------------------------- */
/* records -> table (receiver side of the exchange)                           */
template <int W, bool EXT>
__global__ void insert_records_kernel(Table<W> table, const uint32_t *recs, uint64_t n, DevParams p, uint64_t ordinal_base) {
	constexpr uint32_t RW = 2 * W + (EXT ? 2 : 1);      /* dwords of a wire record (KMR_RECORD_BYTES) */
	unsigned long long nGood = 0; unsigned nClaimed = 0; bool fail = false;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		Record<W> r;
		const uint32_t *p32 = recs + i * RW;
#pragma unroll
		for (int j = 0; j < W; j++) r.key[j] = (uint64_t)p32[2 * j] | ((uint64_t)p32[2 * j + 1] << 32);
		r.w = __uint_as_float(p32[2 * W]);
		r.pkt = EXT ? p32[2 * W + (EXT ? 1 : 0)] : 0u;
		if (r.w == 0.0f) continue;                   /* hole left by the sender's slab allocation */
		Key<W> key;
#pragma unroll
		for (int j = 0; j < W; j++) key.w[j] = r.key[j];
		Occurrence o;
		o.forward = !(r.w < 0.0f); o.w = o.forward ? r.w : -r.w; o.ordinal = ordinal_base + i; o.pkt = r.pkt; o.ltally = -1; o.rtally = -1;
		if (EXT) {   /* ExtensionTracking::trackExtension on the packet (src/KmerTrackingData.h:195-201) */
			auto idx = [](uint32_t ch) -> int { switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'X': return 5; default: return 4; } };
			const int lc = idx(r.pkt & 0xff), rc_ = idx((r.pkt >> 8) & 0xff);
			const uint32_t lq = (r.pkt >> 16) & 0xff, rq_ = r.pkt >> 24;
			if (lq >= p.ext_min_q || lc > 3) o.ltally = lc;
			if (rq_ >= p.ext_min_q || rc_ > 3) o.rtally = 6 + rc_;
		}
		nGood++;
		if (!table_add<W, EXT>(table, key, key_hash<W>(key, p.kb), o, nClaimed)) fail = true;
	}
	nGood = wave_sum(nGood);
	unsigned long long nc = wave_sum((unsigned long long)nClaimed);
	if ((threadIdx.x & 63) == 0) { atomicAdd(&p.stats->raw, nGood); atomicAdd(&p.stats->good, nGood); if (nc) atomicAdd(&p.stats->claimed, nc); }
	if (__any(fail) && (threadIdx.x & 63) == 0) atomicOr(p.err, (uint32_t)ERR_TABLE_FULL);
}

/* table growth: move every entry of 'src' into the (cleared) 'dst' */
template <int W, bool EXT>
__global__ void rehash_kernel(Table<W> src, Table<W> dst, uint32_t kb, uint32_t *err) {
	const uint64_t cap = 1ull << src.log2cap;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
		const Slot<W> s = src.slots[i];
		if (!slot_used<W>(s)) continue;
		const Key<W> key = slot_key<W>(s);
		bool claimed;
		const uint64_t d = table_find_or_insert<W>(dst, key, key_hash<W>(key, kb), claimed);
		if (d == ~0ull) { atomicOr(err, (uint32_t)ERR_TABLE_FULL); continue; }
		dst.slots[d].cntfwd = s.cntfwd; dst.slots[d].wsum = s.wsum; dst.slots[d].first = s.first;
		if (EXT) dst.ext[d] = src.ext[i];
	}
}

/* ----------------------------------------------------------------------- */
/* finalize: table -> bucketed maps                                           */
struct FinalizeParams {
	uint32_t kb;
	uint32_t ext_min_q;           /* ExtensionTracking::getMinQuality (streaming path with extension values) */
	uint32_t min_depth;
	uint32_t has_singletons;      /* cfg.separate_singletons */
	uint64_t nb_weak, nb_sing;
	uint32_t uni_wbits;           /* sk_count_kernel<.., UNI>: the one weight of every k-mer of the build */
};
struct FinalizeCounters { unsigned long long unique, singletons, weak_kept, sing_kept, saturated, sat_sightings; };      /* saturated: keys seen more than 65 535 times (build_mode 3 counts them and their sightings) */

/* where does an entry with 'count' occurrences end up? 1 = weak, 2 = singleton, 0 = dropped.
 * append() keeps count==1 keys in the singleton map when hasSingletons (src/KmerSpectrum.h:1646-1655);
 * purgeMinDepth (:1805-1815) drops weak entries below min_depth only if !hasSingletons || min_depth > 2,
 * and clears the singleton map when min_depth > 1. */
__device__ __forceinline__ int classify(uint32_t count, const FinalizeParams &f) {
	if (f.has_singletons && count == 1) return f.min_depth > 1 ? 0 : 2;
	if ((!f.has_singletons || f.min_depth > 2) && f.min_depth != 1 && count < f.min_depth) return 0;
	return 1;
}

template <int W>
__global__ void classify_kernel(Table<W> t, FinalizeParams f, uint32_t *weakCount, uint32_t *singCount, FinalizeCounters *fc) {
	const uint64_t cap = 1ull << t.log2cap;
	unsigned long long u = 0, s1 = 0, wk = 0, sk = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
		const Slot<W> &s = t.slots[i];
		if (!slot_used<W>(s)) continue;
		const uint32_t count = (uint32_t)s.cntfwd;
		if (count == 0) continue;
		u++;
		if (count == 1) s1++;
		const int c = classify(count, f);
		if (c == 0) continue;
		const uint64_t hash = key_hash<W>(slot_key<W>(s), f.kb);
		if (c == 1) { atomicAdd(&weakCount[hash & (f.nb_weak - 1)], 1u); wk++; }
		else { atomicAdd(&singCount[hash & (f.nb_sing - 1)], 1u); sk++; }
	}
	u = wave_sum(u); s1 = wave_sum(s1); wk = wave_sum(wk); sk = wave_sum(sk);
	if ((threadIdx.x & 63) == 0) {
		if (u) atomicAdd(&fc->unique, u);
		if (s1) atomicAdd(&fc->singletons, s1);
		if (wk) atomicAdd(&fc->weak_kept, wk);
		if (sk) atomicAdd(&fc->sing_kept, sk);
	}
}

/* weak value words (TrackingDataWithDirection / ExtensionTrackingData byte layout):
 * word0 = u16 count, word1 = f32 weightedCount, word2 = u16 directionBias, words 3..14 = ext tallies */
template <int W, bool EXT>
__global__ void scatter_kernel(Table<W> t, FinalizeParams f, const uint64_t *weakStart, uint32_t *weakCursor,
                               uint64_t *weakKeys, uint32_t *weakVals, const uint64_t *singStart, uint32_t *singCursor,
                               uint64_t *singKeys, uint8_t *singWeight, uint32_t *singPkt) {
	const uint64_t cap = 1ull << t.log2cap;
	const uint32_t vw = EXT ? 15 : 3;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
		const Slot<W> &s = t.slots[i];
		if (!slot_used<W>(s)) continue;
		const uint32_t count = (uint32_t)s.cntfwd;
		if (count == 0) continue;
		const int c = classify(count, f);
		if (c == 0) continue;
		const Key<W> key = slot_key<W>(s);
		const uint64_t hash = key_hash<W>(key, f.kb);
		if (c == 1) {
			const uint64_t b = hash & (f.nb_weak - 1);
			const uint64_t pos = weakStart[b] + atomicAdd(&weakCursor[b], 1u);
#pragma unroll
			for (int j = 0; j < W; j++) weakKeys[pos * W + j] = key.w[j];
			uint32_t fwd = (uint32_t)(s.cntfwd >> 32);
			uint32_t cnt = count;
			/* the first sighting lived in the singleton map, which keeps no direction
			 * (TrackingDataSingleton::getDirectionBias, src/KmerTrackingData.h:654) */
			if (f.has_singletons && first_forward(s.first)) fwd -= 1;
			if (cnt > 65535u) { cnt = 65535u; if (fwd > 65534u) fwd = 65534u; }   /* MAX_COUNT, :434 */
			if (fwd > 65535u) fwd = 65535u;
			uint32_t *v = weakVals + pos * vw;
			v[0] = cnt;
			v[1] = __float_as_uint((float)(f.has_singletons ? s.wsum + first_weight_shift(s.first) : s.wsum));
			v[2] = fwd;
			if (EXT) {
#pragma unroll
				for (int j = 0; j < 12; j++) v[3 + j] = t.ext[i].tally[j];
			}
		} else {
			const uint64_t b = hash & (f.nb_sing - 1);
			const uint64_t pos = singStart[b] + atomicAdd(&singCursor[b], 1u);
#pragma unroll
			for (int j = 0; j < W; j++) singKeys[pos * W + j] = key.w[j];
			const float wf = (float)s.wsum;      /* one occurrence: exactly its weight */
			singWeight[pos] = (uint8_t)((unsigned char)(((double)wf * 254.0)) + 1);   /* TrackingDataSingleton::track :646 */
			if (EXT) singPkt[pos] = t.ext[i].pkt;
		}
	}
}

/* sort the entries of each bucket by key (KmerMapByKmerArrayPair::resort, src/Kmer.h:3079-3088).
 * One lane per bucket: insertion sort for the usual <= 64 entries, heap sort above. */
template <int W> struct SortView {
	uint64_t *keys; uint32_t *vals; uint8_t *b8; uint32_t *pkt; uint32_t vw;
	__device__ __forceinline__ Key<W> key(uint64_t i) const { Key<W> k; for (int j = 0; j < W; j++) k.w[j] = keys[i * W + j]; return k; }
	__device__ __forceinline__ void swap(uint64_t a, uint64_t b) const {
		for (int j = 0; j < W; j++) { uint64_t t = keys[a * W + j]; keys[a * W + j] = keys[b * W + j]; keys[b * W + j] = t; }
		if (vals) for (uint32_t j = 0; j < vw; j++) { uint32_t t = vals[a * vw + j]; vals[a * vw + j] = vals[b * vw + j]; vals[b * vw + j] = t; }
		if (b8) { uint8_t t = b8[a]; b8[a] = b8[b]; b8[b] = t; }
		if (pkt) { uint32_t t = pkt[a]; pkt[a] = pkt[b]; pkt[b] = t; }
	}
};
/* One wavefront per bucket.  Up to 64 entries (the normal case: kmers-per-bucket is 32): every lane loads one
 * entry, its rank is the number of smaller keys in the bucket (keys broadcast lane by lane), and it stores the
 * entry at that rank -- all loads happen before any store, so this is in place.  Larger buckets: lane 0 heap-sorts. */
template <int W, int VW>
__global__ __launch_bounds__(256)
void sort_buckets_kernel(SortView<W> v, const uint64_t *start, uint64_t nb) {
	const int lane = threadIdx.x & 63;
	const uint64_t wavesPerGrid = (uint64_t)gridDim.x * (blockDim.x >> 6);
	for (uint64_t b = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); b < nb; b += wavesPerGrid) {
		const uint64_t lo = start[b], n = start[b + 1] - lo;
		if (n < 2) continue;
		if (n <= 64) {
			const bool have = (uint64_t)lane < n;
			Key<W> key;
			uint32_t vals[VW > 0 ? VW : 1];
			uint8_t b8 = 0; uint32_t pkt = 0;
#pragma unroll
			for (int j = 0; j < W; j++) key.w[j] = have ? v.keys[(lo + lane) * W + j] : ~0ull;
			if (have) {
#pragma unroll
				for (int j = 0; j < VW; j++) vals[j] = v.vals[(lo + lane) * VW + j];
				if (v.b8) b8 = v.b8[lo + lane];
				if (v.pkt) pkt = v.pkt[lo + lane];
			}
			uint32_t rank = 0;
			for (uint32_t i = 0; i < (uint32_t)n; i++) {
				Key<W> other;
#pragma unroll
				for (int j = 0; j < W; j++) other.w[j] = __shfl(key.w[j], (int)i, 64);
				/* ties are broken by lane, so the output is a permutation even when a bucket holds a key twice (a merge of
				 * overlapping maps, a loaded image with a repeated key): duplicate_keys_kernel then finds the pair side by side */
				rank += (key_lt<W>(other, key) || (key_eq<W>(other, key) && i < (uint32_t)lane)) ? 1u : 0u;
			}
			if (have) {
#pragma unroll
				for (int j = 0; j < W; j++) v.keys[(lo + rank) * W + j] = key.w[j];
#pragma unroll
				for (int j = 0; j < VW; j++) v.vals[(lo + rank) * VW + j] = vals[j];
				if (v.b8) v.b8[lo + rank] = b8;
				if (v.pkt) v.pkt[lo + rank] = pkt;
			}
		} else if (n <= 256) {
			/* up to four entries per lane (buckets fuller than the 32 per bucket they were sized for: an underestimated k-mer count, or
			 * -- seen with 80 entries per bucket -- every error k-mer kept): the same rank sort, every key of the bucket read by all
			 * lanes at once from the (cached) array.  One lane heap-sorting such buckets took 200 ms for 2 x 10^6 of them. */
			Key<W> key[4];
			uint32_t vals[4][VW > 0 ? VW : 1];
			uint8_t b8[4] = {0, 0, 0, 0}; uint32_t pkt[4] = {0, 0, 0, 0}, rank[4] = {0, 0, 0, 0};
#pragma unroll
			for (int e = 0; e < 4; e++) {
				const uint64_t idx = (uint64_t)lane + 64u * e;
				const bool have = idx < n;
#pragma unroll
				for (int j = 0; j < W; j++) key[e].w[j] = have ? v.keys[(lo + idx) * W + j] : ~0ull;
				if (have) {
#pragma unroll
					for (int j = 0; j < VW; j++) vals[e][j] = v.vals[(lo + idx) * VW + j];
					if (v.b8) b8[e] = v.b8[lo + idx];
					if (v.pkt) pkt[e] = v.pkt[lo + idx];
				}
			}
			for (uint32_t i = 0; i < (uint32_t)n; i++) {
				Key<W> other;
#pragma unroll
				for (int j = 0; j < W; j++) other.w[j] = v.keys[(lo + i) * W + j];
#pragma unroll
				for (int e = 0; e < 4; e++) rank[e] += (key_lt<W>(other, key[e]) || (key_eq<W>(other, key[e]) && i < (uint32_t)lane + 64u * e)) ? 1u : 0u;
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      /* every lane has read every key before any is overwritten */
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int e = 0; e < 4; e++) {
				if ((uint64_t)lane + 64u * e < n) {
#pragma unroll
					for (int j = 0; j < W; j++) v.keys[(lo + rank[e]) * W + j] = key[e].w[j];
#pragma unroll
					for (int j = 0; j < VW; j++) v.vals[(lo + rank[e]) * VW + j] = vals[e][j];
					if (v.b8) v.b8[lo + rank[e]] = b8[e];
					if (v.pkt) v.pkt[lo + rank[e]] = pkt[e];
				}
			}
		} else if (lane == 0) {
			auto sift = [&](uint64_t root, uint64_t end) {
				for (;;) {
					uint64_t child = 2 * root + 1;
					if (child >= end) break;
					if (child + 1 < end && key_lt<W>(v.key(lo + child), v.key(lo + child + 1))) child++;
					if (key_lt<W>(v.key(lo + root), v.key(lo + child))) { v.swap(lo + root, lo + child); root = child; } else break;
				}
			};
			for (uint64_t i = n / 2; i-- > 0;) sift(i, n);
			for (uint64_t end = n - 1; end > 0; end--) { v.swap(lo, lo + end); sift(0, end); }
		}
	}
}

/* on-disk image (src/Kmer.h:3143-3159): header + offsets + per bucket {u32 n; keys; values} */
#ifndef KMR_INSTANCE_TU
__global__ void image_header_kernel(uint8_t *img, const uint64_t *start, uint64_t nb, uint32_t kb, uint32_t vbytes) {
	uint64_t *numbers = (uint64_t *)img;
	for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) {
		if (b == 0) { numbers[0] = nb; numbers[1] = nb - 1; }
		const uint64_t off = 8 * (2 + nb) + 4 * b + start[b] * (kb + vbytes);
		numbers[2 + b] = off;
		const uint32_t n = (uint32_t)(start[b + 1] - start[b]);
		for (int j = 0; j < 4; j++) img[off + j] = (uint8_t)(n >> (8 * j));
	}
}
#endif
template <int W>
__global__ void image_entries_kernel(uint8_t *img, const uint64_t *start, uint64_t nb, uint32_t kb, uint32_t vbytes,
                                     const uint64_t *keys, const uint32_t *vals, uint32_t vw, const uint8_t *b8, const uint32_t *pkt, uint64_t n) {
	for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
		/* bucket of entry e: last b with start[b] <= e */
		uint64_t lo = 0, hi = nb;
		while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (start[mid] <= e) lo = mid; else hi = mid; }
		const uint64_t b = lo, s0 = start[b], nbk = start[b + 1] - s0, i = e - s0;
		const uint64_t off = 8 * (2 + nb) + 4 * b + s0 * (kb + vbytes) + 4;
		Key<W> key;
		for (int j = 0; j < W; j++) key.w[j] = keys[e * W + j];
		uint8_t *kp = img + off + i * kb;
		for (uint32_t j = 0; j < kb; j++) kp[j] = key_byte<W>(key, j);
		uint8_t *vp = img + off + nbk * kb + i * vbytes;
		if (vals) { for (uint32_t j = 0; j < vbytes; j++) vp[j] = (uint8_t)(vals[e * vw + (j >> 2)] >> (8 * (j & 3))); }
		else { vp[0] = b8[e]; if (pkt) for (int j = 0; j < 4; j++) vp[1 + j] = (uint8_t)(pkt[e] >> (8 * j)); }
	}
}
/* inverse: image -> per-bucket counts, then entries */
#ifndef KMR_INSTANCE_TU
__global__ void image_counts_kernel(const uint8_t *img, uint64_t nb, uint32_t *counts) {
	const uint64_t *numbers = (const uint64_t *)img;
	for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t off = numbers[2 + b];
		uint32_t n = 0;
		for (int j = 0; j < 4; j++) n |= (uint32_t)img[off + j] << (8 * j);
		counts[b] = n;
	}
}
#endif
template <int W>
__global__ void image_unpack_kernel(const uint8_t *img, const uint64_t *start, uint64_t nb, uint32_t kb, uint32_t vbytes,
                                    uint64_t *keys, uint32_t *vals, uint32_t vw, uint8_t *b8, uint32_t *pkt, uint64_t n) {
	const uint64_t *numbers = (const uint64_t *)img;
	for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t lo = 0, hi = nb;
		while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (start[mid] <= e) lo = mid; else hi = mid; }
		const uint64_t b = lo, s0 = start[b], nbk = start[b + 1] - s0, i = e - s0;
		const uint64_t off = numbers[2 + b] + 4;
		Key<W> key;
		key_from_bytes<W>(key, img + off + i * kb, kb);
		for (int j = 0; j < W; j++) keys[e * W + j] = key.w[j];
		const uint8_t *vp = img + off + nbk * kb + i * vbytes;
		if (vals) {
			for (uint32_t j = 0; j < vw; j++) vals[e * vw + j] = 0;
			for (uint32_t j = 0; j < vbytes; j++) vals[e * vw + (j >> 2)] |= (uint32_t)vp[j] << (8 * (j & 3));
			vals[e * vw] &= 0xffffu; vals[e * vw + 2] &= 0xffffu;      /* struct padding bytes */
		} else { b8[e] = vp[0]; if (pkt) { uint32_t x = 0; for (int j = 0; j < 4; j++) x |= (uint32_t)vp[1 + j] << (8 * j); pkt[e] = x; } }
	}
}

/* merge of two bucketed maps with the same bucket count (KmerMapByKmerArrayPair::mergeAdd src/Kmer.h:3209-3261 for
 * disjoint key sets, which is what buildKmerSpectrumInParts merges, src/KmerSpectrum.h:1871-1884): entry e of the source
 * map goes to dst_start[b] + shift[b] + (e - src_start[b]); the buckets are sorted afterwards and checked for duplicates. */
#ifndef KMR_INSTANCE_TU
__global__ void merge_counts_kernel(const uint64_t *a_start, const uint64_t *b_start, uint64_t nb, uint32_t *counts) {
	for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x)
		counts[b] = (uint32_t)((a_start[b + 1] - a_start[b]) + (b_start[b + 1] - b_start[b]));
}
#endif
template <int W>
__global__ void merge_copy_kernel(const uint64_t *src_start, const uint64_t *other_start, bool after_other, uint64_t nb, uint64_t n,
                                  const uint64_t *skeys, const uint32_t *svals, uint32_t vw, const uint8_t *sb8, const uint32_t *spkt,
                                  const uint64_t *dst_start, uint64_t *dkeys, uint32_t *dvals, uint8_t *db8, uint32_t *dpkt) {
	for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t lo = 0, hi = nb;
		while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (src_start[mid] <= e) lo = mid; else hi = mid; }
		const uint64_t b = lo;
		const uint64_t pos = dst_start[b] + (after_other ? other_start[b + 1] - other_start[b] : 0) + (e - src_start[b]);
		for (int j = 0; j < W; j++) dkeys[pos * W + j] = skeys[e * W + j];
		if (svals) for (uint32_t j = 0; j < vw; j++) dvals[pos * vw + j] = svals[e * vw + j];
		if (sb8) db8[pos] = sb8[e];
		if (spkt) dpkt[pos] = spkt[e];
	}
}
/* KmerMapByKmerArrayPair::mergeAdd's cmp == 0 branch (src/Kmer.h:3238-3241) over the sorted union of two weak maps: a key both maps
 * held stands twice in its bucket, next to itself.  First pass: distinct keys per bucket; second pass (after a scan): every run of
 * equal keys becomes one entry whose value is a.add(b) -- TrackingData::add / TrackingDataWithDirection::add / ExtensionTrackingData::add
 * (src/KmerTrackingData.h:489-493,538-542,1059-1064): += on the u16 count and directionBias (they wrap, nothing saturates), on the
 * float weightedCount, on the u32 extension tallies. */
template <int W>
__global__ void merge_distinct_kernel(const uint64_t *start, uint64_t nb, const uint64_t *keys, uint32_t *counts) {
	for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t c = 0;
		for (uint64_t e = start[b]; e < start[b + 1]; e++) {
			bool eq = e > start[b];
			for (int j = 0; j < W && eq; j++) eq = keys[e * W + j] == keys[(e - 1) * W + j];
			c += eq ? 0u : 1u;
		}
		counts[b] = c;
	}
}
template <int W>
__global__ void merge_add_kernel(const uint64_t *start, uint64_t nb, const uint64_t *keys, const uint32_t *vals, uint32_t vw,
                                 const uint64_t *dst_start, uint64_t *dkeys, uint32_t *dvals) {
	for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t o = dst_start[b];
		for (uint64_t e = start[b]; e < start[b + 1]; e++) {
			bool eq = e > start[b];
			for (int j = 0; j < W && eq; j++) eq = keys[e * W + j] == keys[(e - 1) * W + j];
			if (!eq) {
				for (int j = 0; j < W; j++) dkeys[o * W + j] = keys[e * W + j];
				for (uint32_t j = 0; j < vw; j++) dvals[o * vw + j] = vals[e * vw + j];
				o++;
			} else {
				uint32_t *d = dvals + (o - 1) * vw;
				d[0] = (d[0] + vals[e * vw]) & 0xffffu;
				d[1] = __float_as_uint(__uint_as_float(d[1]) + __uint_as_float(vals[e * vw + 1]));
				d[2] = (d[2] + vals[e * vw + 2]) & 0xffffu;
				for (uint32_t j = 3; j < vw; j++) d[j] += vals[e * vw + j];
			}
		}
	}
}
template <int W>
__global__ void duplicate_keys_kernel(const uint64_t *start, uint64_t nb, const uint64_t *keys, uint32_t *found) {
	for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x)
		for (uint64_t e = start[b] + 1; e < start[b + 1]; e++) {
			bool eq = true;
			for (int j = 0; j < W; j++) eq = eq && keys[e * W + j] == keys[(e - 1) * W + j];
			if (eq) { atomicOr(found, 1u); break; }
		}
}

#ifndef KMR_INSTANCE_TU
__global__ void histogram_kernel(const uint32_t *vals, uint32_t vw, uint64_t n, uint32_t nbins, unsigned long long *counts, double *weights) {
	for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t c = vals[e * vw] & 0xffffu;
		if (c >= nbins) c = nbins - 1;
		atomicAdd(&counts[c], 1ull);
		if (weights) atomicAdd(&weights[c], (double)__uint_as_float(vals[e * vw + 1]));
	}
}
#endif

/* KmerSpectrum::Histogram::set/addRecord (src/KmerSpectrum.h:964-974,1036-1056) over the weak map (vals != NULL) or the
 * singleton map (sweight != NULL).  getIdx (:936-938) goes through a 65536-entry table the host fills with its own libm, so
 * the log() that decides a bucket is the one the reference would have used.  Buckets below HIST_LDS (where nearly all
 * entries fall) are accumulated per block in LDS. */
static const int HIST_LDS = 512;
#ifndef KMR_INSTANCE_TU
__global__ __launch_bounds__(256)
void ref_histogram_kernel(const uint32_t *vals, uint32_t vw, const uint8_t *sweight, uint64_t n, const uint32_t *idx_of_count,
                          unsigned long long *visits, unsigned long long *visitedCount, double *visitedWeight) {
	__shared__ uint32_t lv[HIST_LDS];
	__shared__ unsigned long long lc[HIST_LDS];
	__shared__ double lw[HIST_LDS];
	for (int i = threadIdx.x; i < HIST_LDS; i += blockDim.x) { lv[i] = 0; lc[i] = 0; lw[i] = 0.0; }
	__syncthreads();
	for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t c; double w;
		if (vals) { c = vals[e * vw] & 0xffffu; w = (double)__uint_as_float(vals[e * vw + 1]); }
		else { const uint32_t b = sweight[e]; c = b ? 1u : 0u; w = b ? (double)(int)(b - 1) / 254.0 : 0.0; }
		if (c == 0) continue;
		const uint32_t idx = idx_of_count[c];
		if (idx < (uint32_t)HIST_LDS) { atomicAdd(&lv[idx], 1u); atomicAdd(&lc[idx], (unsigned long long)c); atomicAdd(&lw[idx], w); }
		else { atomicAdd(&visits[idx], 1ull); atomicAdd(&visitedCount[idx], (unsigned long long)c); atomicAdd(&visitedWeight[idx], w); }
	}
	__syncthreads();
	for (int i = threadIdx.x; i < HIST_LDS; i += blockDim.x) if (lv[i]) { atomicAdd(&visits[i], (unsigned long long)lv[i]); atomicAdd(&visitedCount[i], lc[i]); atomicAdd(&visitedWeight[i], lw[i]); }
}
#endif

/* ----------------------------------------------------------------------- */
/* exclusive scan u32 -> u64 (bucket sizes -> bucket starts), three launches */
static const int SCAN_ITEMS = 2048;   /* per block of 256 threads */
#ifndef KMR_INSTANCE_TU
__global__ void scan_block_sums_kernel(const uint32_t *in, uint64_t n, unsigned long long *blockSums) {
	__shared__ unsigned long long red[256];
	const uint64_t base = (uint64_t)blockIdx.x * SCAN_ITEMS;
	unsigned long long s = 0;
	for (int j = 0; j < SCAN_ITEMS / 256; j++) { const uint64_t i = base + (uint64_t)j * 256 + threadIdx.x; if (i < n) s += in[i]; }
	red[threadIdx.x] = s;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
	if (threadIdx.x == 0) blockSums[blockIdx.x] = red[0];
}
#endif
#ifndef KMR_INSTANCE_TU
__global__ void scan_sums_kernel(unsigned long long *blockSums, uint64_t nblocks, unsigned long long *total) {
	/* single block; sequential over chunks of 256 */
	__shared__ unsigned long long buf[256];
	__shared__ unsigned long long carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (uint64_t base = 0; base < nblocks; base += 256) {
		const uint64_t i = base + threadIdx.x;
		const unsigned long long v = i < nblocks ? blockSums[i] : 0;
		buf[threadIdx.x] = v;
		__syncthreads();
		for (int o = 1; o < 256; o <<= 1) {
			unsigned long long t = (int)threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
			__syncthreads();
			buf[threadIdx.x] += t;
			__syncthreads();
		}
		if (i < nblocks) blockSums[i] = carry + buf[threadIdx.x] - v;
		__syncthreads();
		if (threadIdx.x == 255) carry += buf[255];
		__syncthreads();
	}
	if (threadIdx.x == 0) *total = carry;
}
#endif
#ifndef KMR_INSTANCE_TU
__global__ void scan_apply_kernel(const uint32_t *in, uint64_t n, const unsigned long long *blockSums, uint64_t *out) {
	__shared__ unsigned long long buf[256];
	const uint64_t base = (uint64_t)blockIdx.x * SCAN_ITEMS;
	/* each thread owns SCAN_ITEMS/256 consecutive items */
	const int per = SCAN_ITEMS / 256;
	unsigned long long local[per];
	unsigned long long s = 0;
	for (int j = 0; j < per; j++) { const uint64_t i = base + (uint64_t)threadIdx.x * per + j; local[j] = i < n ? in[i] : 0; s += local[j]; }
	buf[threadIdx.x] = s;
	__syncthreads();
	for (int o = 1; o < 256; o <<= 1) {
		unsigned long long t = (int)threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
		__syncthreads();
		buf[threadIdx.x] += t;
		__syncthreads();
	}
	unsigned long long run = blockSums[blockIdx.x] + buf[threadIdx.x] - s;
	for (int j = 0; j < per; j++) { const uint64_t i = base + (uint64_t)threadIdx.x * per + j; if (i < n) out[i] = run; run += local[j]; }
	if (base + SCAN_ITEMS >= n && threadIdx.x == 255) out[n] = run;   /* total at out[n] */
}
#endif

}  // namespace kmr
#endif
