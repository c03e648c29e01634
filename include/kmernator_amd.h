/*
 * kmernator_amd.h -- C-ABI of the MI355X-native k-mer spectrum builder.
 *
 * This is the drop-in boundary for ONE path of JGI-Bioinformatics/Kmernator:
 * the FilterReads / KmerSpectrum k-mer-spectrum build and lookup.  Every entry
 * point below names the reference interface it replaces (path:line under the
 * reference checkout).  The reference has no FFI of its own -- the boundary is
 * a compile-time C++ template API -- so the C++ shim a maintainer adds on the
 * Kmernator side (a KmerSpectrum<> subclass that calls these functions and
 * restore()s the returned images into the reference's own map types) is shown
 * in INTEGRATION.md and shipped as include/kmernator_amd_shim.hpp.
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary.
 *   - every function returns 0 on success or a negative kmr_status; the text
 *     of the last error of a handle is kmr_last_error(h) (thread-unsafe, like
 *     the reference's maps, src/Kmer.h:2269 "bucket ownership" model).
 *   - all in/out buffers are caller owned.  "host" pointers are ordinary
 *     memory; "dev" pointers are HIP device memory of the handle's device.
 *   - k is per handle (the reference keeps it in the process-global KmerSizer,
 *     src/Kmer.h:83-127).
 *   - packed k-mers use the reference byte layout (TwoBitSequence,
 *     src/TwoBitSequence.cpp:242-269): 4 bases per byte, first base in bits
 *     7..6, A=0 C=1 G=2 T=3, kb = ceil(k/4) bytes, pad bits zero.
 *   - there is NO CPU fallback: if the HIP device or the code object is
 *     missing every compute entry point fails with KMR_ERR_NO_DEVICE.
 */
#ifndef KMERNATOR_AMD_H_
#define KMERNATOR_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMR_ABI_VERSION 1

typedef enum kmr_status {
	KMR_OK = 0,
	KMR_ERR_INVALID_ARG = -1,
	KMR_ERR_NO_DEVICE = -2,   /* no HIP device / kernel image not loadable */
	KMR_ERR_HIP = -3,         /* a HIP runtime call failed                  */
	KMR_ERR_OOM = -4,         /* device or host allocation failed           */
	KMR_ERR_STATE = -5,       /* call not valid in the handle's state       */
	KMR_ERR_CAPACITY = -6,    /* a caller supplied buffer is too small      */
	KMR_ERR_UNSUPPORTED = -7  /* option not built (e.g. solid map)          */
} kmr_status;

/* Value type families of the reference's maps (src/KmerTrackingData.h).
 * COUNT_DIR: weak = TrackingDataWithDirection (12 B: u16 count @0, f32
 *            weightedCount @4, u16 directionBias @8; :491-551), singleton =
 *            TrackingDataSingleton (1 B; :613-686).   [FilterReads(-P)]
 * EXT:       weak = ExtensionTrackingData (60 B = the 12 B above + u32
 *            [Left,Right][A,C,G,T,N,X]; :1027-1074), singleton =
 *            ExtensionTrackingDataSingleton (5 B; :1078-1126). [MeraculousCounter] */
typedef enum kmr_value_kind { KMR_VALUE_COUNT_DIR = 0, KMR_VALUE_EXT = 1 } kmr_value_kind;

/* Which of the spectrum's maps (src/KmerSpectrum.h:396-403). */
typedef enum kmr_map { KMR_MAP_WEAK = 0, KMR_MAP_SINGLETON = 1, KMR_MAP_SOLID = 2 } kmr_map;

/* Replaces the option singletons read on this path:
 * KmerSizer::set (src/Kmer.h:107), KmerSpectrumOptions (src/KmerSpectrum.h:92-139),
 * GeneralOptions min-quality-score / fastq-base-quality (src/Options.h:327-331),
 * ExtensionTracking::setMinQuality (src/KmerTrackingData.h:157-163) and the bucket
 * sizing of the KmerSpectrum constructor (src/KmerSpectrum.h:414-421, src/Kmer.h:2837,2224). */
/* Two fields SURVEY section 8(b) sketches are deliberately absent.  num_devices: one handle drives ONE device, as one MPI rank
 * drives one share of the spectrum in the reference (DistributedKmerSpectrum, src/DistributedFunctions.h:126); a node's N GPUs are
 * N handles with rank / world_size set (one process or thread each) joined by kmr_exchange_*.  deterministic: results do not
 * depend on scheduling -- counts, direction biases (global stream ordinals decide the first sighting), singleton bytes and
 * extension tallies are exact and repeatable; only weightedCount's low bits depend on the order of an f64 atomic sum, inside the
 * 1e-5 * count the reference's own OpenMP / MPI builds vary by. */
typedef struct kmr_config {
	uint32_t struct_size;            /* = sizeof(kmr_config), ABI guard                       */
	uint32_t k;                      /* k-mer length in bases, 1..128                          */
	uint64_t num_buckets_weak;       /* 0 => derive from estimated_raw_kmers like the ctor     */
	uint64_t num_buckets_singleton;  /* 0 => derive; both rounded up to a power of two <= 2^26 */
	uint64_t estimated_raw_kmers;    /* KmerSpectrum::estimateRawKmers() of the whole job; with world_size > 1 a
	                                  * rank sizes its maps for estimated_raw_kmers / world_size, as
	                                  * DistributedKmerSpectrum::estimateRawKmers does (src/DistributedFunctions.h:144-162) */
	uint32_t value_kind;             /* kmr_value_kind                                         */
	float    min_weight;             /* --min-kmer-quality (TrackingData::minimumWeight), 0.10 */
	uint32_t min_quality_score;      /* --min-quality-score, 3 (MeraculousCounter: 2)          */
	uint32_t fastq_start_char;       /* Phred base of the quals handed in: 33 or 64            */
	uint32_t ext_min_quality;        /* ExtensionTracking min quality, 20                      */
	uint32_t separate_singletons;    /* 1 = reference default (hasSingletons)                  */
	uint32_t kmer_subsample;         /* --kmer-subsample; keep k-mer iff hash % n == 0; 1=all  */
	int32_t  device;                 /* HIP device ordinal; -1 = current                       */
	uint32_t rank;                   /* owner partition of this handle ...                     */
	uint32_t world_size;             /* ... out of world_size (1 = single partition)           */
	double   estimated_depth;        /* --estimated-depth, 20 (used only to derive buckets)    */
	double   estimated_error_rate;   /* --estimated-error-rate, 0.35 (ditto)                   */
	uint32_t kmers_per_bucket;       /* --kmers-per-bucket, 32 (ditto)                         */
	uint32_t num_parts;              /* --build-partitions: keep k-mers whose                  */
	uint32_t part_idx;               /*   getDMPThread(kmer,num_parts)==part_idx; 0/1 = all    */
	uint32_t build_mode;             /* 0 = auto (3 where it applies, else 2), 1 = open-addressed device table, 2 = two-level
	                                    k-mer partition + LDS counting, 3 = super-k-mer lists: one scatter pass by minimizer,
	                                    expansion + counting in LDS (k >= 13; both value kinds in all three modes) */
	uint64_t max_table_entries;      /* 0 = size from estimated_raw_kmers; else distinct-key
	                                    capacity of the device table                           */
	uint32_t hash_kind;              /* kmr_hash_kind: which hash places a k-mer in its bucket / owner / part / subsample.
	                                    0 = KmerHasher::getHash as it is today: lookup3 hashlittle2 (src/Kmer.h:207-230);
	                                    1 = lookup8 hash() with level 0xDEADBEEF (src/lookup8.h:90-160; what getHash used
	                                    before, src/Kmer.h:210-212 -- the reference no longer calls it).  Images written with
	                                    one kind cannot be loaded into a handle of the other (the bucket of a key differs) */
	uint32_t size_tracker;           /* 1 = keep KmerSpectrum::SizeTracker snapshots (src/KmerSpectrum.h:812-900), see
	                                    kmr_size_tracker below                                                          */
} kmr_config;
enum kmr_hash_kind { KMR_HASH_LOOKUP3 = 0, KMR_HASH_LOOKUP8 = 1 };

typedef struct kmr_handle kmr_handle;

/* Counters of KmerSpectrum (src/KmerSpectrum.h:405-409,455-459) and of
 * TrackingData::discarded (src/KmerTrackingData.h:354-364). */
typedef struct kmr_stats {
	uint64_t raw_kmers;        /* every k-mer occurrence offered to append()                  */
	uint64_t raw_good_kmers;   /* occurrences that passed the weight test                     */
	uint64_t unique_kmers;     /* distinct k-mers ever seen (weak + singleton)                */
	uint64_t singleton_kmers;  /* distinct k-mers seen exactly once                           */
	uint64_t discarded;        /* occurrences with weight <= min_weight                       */
	uint64_t weak_entries;     /* entries now in the weak map (after finalize: after purge)   */
	uint64_t singleton_entries;/* entries now in the singleton map                            */
	uint64_t reads;            /* reads consumed (discarded reads included)                   */
} kmr_stats;

/* Fill *cfg with the reference defaults listed above (k must still be set). */
int kmr_config_init(kmr_config *cfg);

/* Create / destroy.  Replaces KmerSpectrum(estimatedRawKmers, separateSingletons)
 * (src/KmerSpectrum.h:414-421) and DistributedKmerSpectrum(world, ...)
 * (src/DistributedFunctions.h:126-131); the handle owns all device memory. */
int  kmr_create(const kmr_config *cfg, kmr_handle **out);
void kmr_destroy(kmr_handle *h);
const char *kmr_last_error(const kmr_handle *h);   /* h may be NULL: creation errors */

/* Effective bucket counts after power-of-two rounding (BucketExposedMapLogic::
 * resizeBuckets, src/Kmer.h:2224-2236). */
int kmr_num_buckets(const kmr_handle *h, int which_map, uint64_t *out);

/* Feed one batch of reads.  Replaces the per-read body of
 * KmerSpectrum::_buildKmerSpectrumSerial/_Parallel (src/KmerSpectrum.h:1914-2074)
 * = KmerReadUtils::buildWeightedKmers (src/KmerReadUtils.h:176-248) + append()
 * (src/KmerSpectrum.h:1578-1668).  Host buffers:
 *   bases  : ASCII sequence characters of all reads back to back
 *   quals  : ASCII qualities, same layout; NULL = reads without quals
 *            (reference reads, weight 1.0; src/KmerReadUtils.h:195-199)
 *   offsets: n_reads+1 byte offsets into bases/quals (read r = [off[r],off[r+1]))
 *   discarded: optional, 1 = Read::isDiscarded() (skipped)
 * May be called any number of times before kmr_finalize. */
int kmr_add_reads(kmr_handle *h, const char *bases, const char *quals,
                  const uint64_t *offsets, uint64_t n_reads,
                  uint64_t first_global_read_idx, const uint8_t *discarded);

/* Same, but the buffers are already in the handle's device memory (this is
 * the form bench.py times: inputs resident in HBM).  Asynchronous on the
 * handle's stream; kmr_sync() waits. */
int kmr_add_reads_dev(kmr_handle *h, const void *dev_bases, const void *dev_quals,
                      const void *dev_offsets, uint64_t n_reads, uint64_t total_bases,
                      uint64_t first_global_read_idx, const void *dev_discarded);
int kmr_sync(kmr_handle *h);

/* Post-build steps of buildKmerSpectrumInParts / DistributedKmerSpectrum::
 * buildKmerSpectrum: purgeMinDepth(min_depth) (src/KmerSpectrum.h:1805-1815,
 * 1825; src/DistributedFunctions.h:560-569) and optimize() (sorted buckets,
 * src/Kmer.h:3079-3088).  After this call the maps are immutable and
 * queryable / exportable. */
int kmr_finalize(kmr_handle *h, uint32_t min_depth);

/* Position in the whole input of the next base handed to kmr_add_reads* (default: 0 after kmr_create / kmr_reset, then the bases
 * added so far).  The order of occurrences in the input decides which sighting of a k-mer was its first -- the one
 * TrackingDataSingleton keeps without a direction and with a quantised weight (src/KmerTrackingData.h:641-658).  Ranks that each
 * read a slice of one input set their slice's offset here, and the owner-partitioned spectra (kmr_sk_exchange_*) come out as the
 * serial build of the whole input would have them, whatever the ranks' timing. */
int kmr_set_stream_origin(kmr_handle *h, uint64_t ordinal);

/* Empty the maps and counters but keep the device allocations, as
 * KmerSpectrum::buildKmerSpectrum does on entry (weak.reset(false);
 * singleton.reset(false), src/KmerSpectrum.h:2091-2096).  Asynchronous. */
int kmr_reset(kmr_handle *h);
/* Give the build-time hash table back to the device once finalized (the
 * finalized maps stay queryable); kmr_reset() re-allocates it. */
int kmr_release_table(kmr_handle *h);

int kmr_get_stats(kmr_handle *h, kmr_stats *out);

/* KmerSpectrum::subtractReference (src/KmerSpectrum.h:472-474; apps/FilterReads-P.cpp:117): k-mers present in the
 * finalized spectrum `reference` (same k, same device) are skipped by later kmr_add_reads* calls on h before they
 * count as raw k-mers (append(), :1582-1588).  kmr_finalize(h) drops the link as optimize() does; NULL drops it now.
 * kmr_subtracted = the `subtracted` counter. */
int kmr_subtract_reference(kmr_handle *h, kmr_handle *reference);
int kmr_subtracted(kmr_handle *h, uint64_t *out);

/* Lookup.  Replaces KmerMap::getElementIfExists(kmer).value().getCount()
 * (src/Kmer.h:2617-2624; consumer src/ReadSelector.h:924-931) and
 * KmerSpectrum::getCount(kmer,false) (src/KmerSpectrum.h:701-725): weak count
 * if present, else 1 if in the singleton map, else 0.  Keys are packed
 * canonical k-mers, kb bytes each. */
int kmr_lookup(kmr_handle *h, const uint8_t *packed_kmers, uint64_t n, uint32_t *counts);

/* Fused extract + canonicalise + lookup for whole reads: counts_out gets one
 * u32 per k-mer position of each read (read r writes out_offsets[r]..+L-k+1).
 * This is ReadSelector::scoreReadByKmers' inner loop (src/ReadSelector.h:1048-1076). */
int kmr_lookup_reads(kmr_handle *h, const char *bases, const uint64_t *offsets,
                     uint64_t n_reads, uint32_t *counts_out, const uint64_t *out_offsets);

/* Trim and score whole reads against the weak map: ReadSelector::scoreAndTrimReads
 * (src/ReadSelector.h:1182-1207) = per-position counts (getValue :924-931, weak map only), cut at the first
 * N/X markup (_setNumKmers :1037-1047), first longest run of k-mers with count >= minimum_kmer_score
 * (trimReadByMinimumKmerScore :949-1014, bimodal detection off), score of that run (scoreReadByScoringType
 * :1094-1180) and setTrimHeaders (:1015-1036).  Per read: trim_offset (bases), trim_length (bases, run + k - 1,
 * 0 = nothing left), score (-1 if nothing left; KS_SUM leaves 0 as the reference does), was_trimmed. */
typedef enum kmr_scoring { KMR_SCORE_SUM = 0, KMR_SCORE_MEDIAN = 1, KMR_SCORE_MIN = 2, KMR_SCORE_MAX = 3, KMR_SCORE_AVG = 4 } kmr_scoring;
int kmr_score_reads(kmr_handle *h, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                    double minimum_kmer_score, int scoring_type,
                    uint32_t *trim_offset, uint32_t *trim_length, float *score, uint8_t *was_trimmed);

/* Export in the reference's on-disk / mmap format so the caller can
 * WeakMapType::restore(dst) it.  Replaces KmerMapByKmerArrayPair::getSizeToStore /
 * store(void*) (src/Kmer.h:3143-3159,3181-3191) and KmerSpectrum::storeMmap
 * (src/KmerSpectrum.h:476-488).  Layout: u64 numBuckets; u64 bucketMask;
 * u64 offset[numBuckets]; then per bucket {u32 n; u8 keys[n][kb]; V values[n]}
 * with keys sorted by memcmp. */
int kmr_image_size(kmr_handle *h, int which_map, uint64_t *bytes);
int kmr_write_image(kmr_handle *h, int which_map, void *dst, uint64_t capacity);

/* Restore: replaces KmerSpectrum::restoreMmap / KmerMapByKmerArrayPair(const void*)
 * (src/KmerSpectrum.h:489-518, src/Kmer.h:3124-3135).  The handle must be fresh
 * (no reads added); it becomes finalized. */
int kmr_load_image(kmr_handle *h, int which_map, const void *src, uint64_t len);
/* KmerMapByKmerArrayPair::mergeAdd (src/Kmer.h:3209-3261) of a stored map into the handle's finalized map of the same kind and
 * bucket count: the restore-and-merge loop of KmerSpectrum::buildKmerSpectrumInParts (src/KmerSpectrum.h:1871-1884; parts built
 * with kmr_config.num_parts / part_idx hold disjoint k-mers) and the weak-map step of KmerSpectrum::mergeVector (:2572-2584): a
 * k-mer both weak maps hold gets a.add(b) -- += on the u16 count and directionBias (as in the reference they wrap, nothing
 * saturates), on the float weightedCount and on the u32 extension tallies (src/KmerTrackingData.h:489-493,538-542,1059-1064).
 * Singleton maps that share a k-mer are refused with KMR_ERR_UNSUPPORTED: that k-mer would have to be promoted into the weak map,
 * KmerMap::mergePromote, which the reference itself throws on (src/Kmer.h:2675-2677). */
int kmr_merge_image(kmr_handle *h, int which_map, const void *src, uint64_t len);

/* Order-independent digest of a finalized map: a spectrum of 10^9 entries is compared, and the rank / part maps of a partitioned
 * build are added up, without moving the maps (the reference compares maps entry by entry on the host, e.g. the store / restore
 * check of test/KmerTest.cpp:545-594; nothing of the kind exists there for a whole spectrum).  Per entry
 *   e = mix(..mix(mix(v0 ^ w[0]) ^ w[1])..)  over the key's 8-byte big-endian words (the packed canonical k-mer, zero padded),
 *   mix(x): x += 0x9E3779B97F4A7C15; x = (x ^ x >> 30) * 0xBF58476D1CE4E5B9; x = (x ^ x >> 27) * 0x94D049BB133111EB; x ^ x >> 31
 *   weak map:      v0 = count | directionBias << 16; extension values then fold their 12 u32 tallies in pairs (lo | hi << 32)
 *   singleton map: v0 = 1 << 32 | _weight | packet << 40
 * hash_sum / hash_xor are the sum (mod 2^64) and the xor of e over all entries, count_sum / dir_sum the plain sums;
 * weighted_sum adds weightedCount (singleton map: (_weight - 1) / 254) in double -- its last bits depend on the order of addition,
 * as the reference's own float accumulation does (src/KmerTrackingData.h:427-448). */
typedef struct kmr_digest {
	uint64_t entries, count_sum, dir_sum, hash_sum, hash_xor;
	double weighted_sum;
} kmr_digest;
int kmr_map_digest(kmr_handle *h, int which_map, kmr_digest *out);

/* The synthetic reads of SURVEY.md section 8(d) (what bench.py times and the full-size parity tests build), generated on the
 * current device into caller-owned device buffers: reads first_read .. first_read + n_reads (global indices: a rank's share of a
 * job is a range) of read_len bases over a uniform random genome of genome_len bases, random strand, 1 % substitutions, no N;
 * Phred-33 qualities, all 'I' or (noisy_quals) Q 40/30/20/10/2 with probabilities .80/.10/.05/.04/.01 and Q10 on errors.
 * Integer arithmetic only (xorshift64*, defined in kmernator_amd/csrc/kmr_synth.hpp), so the CPU restatement of the tests gives
 * the same bytes.  dev_bases / dev_quals: n_reads * read_len bytes (dev_quals may be NULL); dev_offsets: n_reads + 1 entries or
 * NULL.  Runs on the null stream and returns when the bytes are there.  The reference has no generator; its inputs are files. */
int kmr_synth_reads_dev(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t genome_len, uint32_t noisy_quals,
                        void *dev_bases, void *dev_quals, uint64_t *dev_offsets);

/* Histogram of weak counts (KmerSpectrum::Histogram, src/KmerSpectrum.h:909-1057):
 * counts[c] = number of weak entries with count == c for c < n_bins-1, last bin
 * collects the rest; weights[c] = sum of weightedCount (may be NULL). */
int kmr_count_histogram(kmr_handle *h, uint64_t *counts, double *weights, uint32_t n_bins);
/* KmerSpectrum::Histogram(zoom_max, log_base).set(spectrum) (src/KmerSpectrum.h:909-1057; getHistogram uses
 * Histogram(256), :1066-1071): per bucket visits, visitedCount and visitedWeight over the weak map and, when it
 * is kept, the singleton map.  Bucket of a count: count <= zoom_max ? count : log(count)/log(log_base) -
 * zoomLogSkip + zoom_max (:936-938).  The arrays need kmr_histogram_bins(zoom_max) = 65538 + zoom_max entries. */
uint32_t kmr_histogram_bins(uint32_t zoom_max);
int kmr_histogram(kmr_handle *h, uint32_t zoom_max, double log_base, uint64_t *visits,
                  uint64_t *visited_count, double *visited_weight, uint32_t n_bins);

/* MeraculousDistributedKmerSpectrum::dumpCounts / dumpGraphs
 * (src/Meraculous.h:107-134): text lines for every weak entry with
 * count >= min_depth, both orientations.  Appends to the file at 'path'. */
int kmr_dump_mercount(kmr_handle *h, const char *path, uint32_t min_depth);
int kmr_dump_mergraph(kmr_handle *h, const char *path, uint32_t min_depth);

/* ---- stateless helpers (bit-identical to the reference functions) -------- */

/* KmerHasher::getHash (src/Kmer.h:207-230) = Lookup3::hashlittle2
 * (src/lookup3.h:470-641) with pc=0xDEADBEEF, pb=0, result c | b<<32. */
uint64_t kmr_hash(const uint8_t *key, uint32_t len);
uint64_t kmr_hash_of_kind(const uint8_t *key, uint32_t len, uint32_t hash_kind);      /* the same for either hash kind, len <= 32 */
/* BucketExposedMapLogic::getBucketIdx / getLocalThreadId / getDistributedThreadId
 * (src/Kmer.h:2329-2333, 2269-2280, 2284-2295). */
uint64_t kmr_bucket_idx(uint64_t hash, uint64_t num_buckets_pow2);
uint32_t kmr_local_thread_id(uint64_t hash, uint64_t num_buckets_pow2, uint32_t num_threads);
uint32_t kmr_distributed_thread_id(uint64_t hash, uint32_t world_size);
/* TwoBitSequence::compressSequence (src/TwoBitSequence.cpp:242-269); returns
 * the number of markups (non-ACGT bases), written to markup_pos/markup_char
 * up to markup_cap. */
int64_t kmr_compress_sequence(const char *bases, uint64_t len, uint8_t *out,
                              uint32_t *markup_pos, char *markup_char, uint64_t markup_cap);
/* Kmer::buildLeastComplement (src/Kmer.h:356-364): writes the canonical form
 * of a packed k-mer, returns 1 if the input was already the least. */
int kmr_least_complement(const uint8_t *packed, uint32_t k, uint8_t *out);

/* ---- device-level pieces for one-process-per-GPU owner partitioning ------
 * These replace the body of DistributedKmerSpectrum::_buildKmerSpectrumMPI
 * (src/DistributedFunctions.h:340-458): sender side = buildWeightedKmers +
 * isDiscard + getThreadIds (:418-433) + bufferMessage (:438); the
 * MPI_Alltoallv (src/MPIBuffer.h:588-600) is done by the caller (RCCL
 * all-to-all over xGMI via torch.distributed); receiver side =
 * StoreKmerMessageHeaderProcessor::process -> append (:323-328).
 *
 * A record is KMR_RECORD_BYTES(k, value_kind) bytes, 4-byte aligned: the u64 key words (big-endian packed
 * k-mer, zero padded; each word as two little-endian u32, low half first), an f32 signed weight (negative =
 * observed strand was the reverse complement, as StoreKmerMessageHeader::weight :279) and, for
 * KMR_VALUE_EXT only, the u32 extension packet (leftBase,rightBase chars, leftQ,rightQ;
 * ExtensionMessagePacket, src/KmerTrackingData.h:232-288).  12 bytes per k-mer at k <= 32 against the
 * reference's 24 + kb byte message (src/DistributedFunctions.h:274-303). */
#define KMR_KEY_WORDS(k) ((((k) + 3u) / 4u + 7u) / 8u)
#define KMR_RECORD_BYTES(k, value_kind) (8u * KMR_KEY_WORDS(k) + ((value_kind) == KMR_VALUE_EXT ? 8u : 4u))

/* Extract all good k-mers of a device-resident read batch and bin them by
 * owner = kmr_distributed_thread_id(hash, world_size) into world_size
 * contiguous segments of dev_records; segment s starts at record seg_capacity * s.
 * dev_seg_counts[world_size] (u64, device) receives the number of records per owner.
 * Returns KMR_ERR_CAPACITY (after sync) if a segment overflowed.  Asynchronous on the
 * handle's stream otherwise. */
int kmr_extract_by_owner_dev(kmr_handle *h, const void *dev_bases, const void *dev_quals,
                             const void *dev_offsets, uint64_t n_reads, uint64_t total_bases,
                             uint64_t first_global_read_idx, const void *dev_discarded,
                             void *dev_records, uint64_t seg_capacity, void *dev_seg_counts);
/* Insert n records (any owner mix that belongs to this handle) into the table. */
int kmr_insert_records_dev(kmr_handle *h, const void *dev_records, uint64_t n_records);

/* The same exchange for build_mode 3 (super-k-mer lists), whose unit is not the k-mer but the list: every rank scatters the
 * super-k-mers of ITS reads (kmr_add_reads* after kmr_sk_exchange_begin: nothing is filtered by owner; without that call a
 * handle with world_size > 1 keeps what getDistributedThreadId gives its rank, as in the other build modes, and has nothing to
 * exchange) into the job's 2^list_bits lists, list l belongs to
 * rank l % world_size, and what a rank holds of other ranks' lists travels as it lies: ~4 bytes per k-mer on the wire instead of
 * the reference's 24 + kb (src/DistributedFunctions.h:274-303).  The owner of a k-mer is decided by its minimizer (private to
 * the build), not by getDistributedThreadId: per-rank spectra are a different partition of the same k-mers, their union is
 * the same spectrum.
 *   kmr_sk_exchange_begin     the handle's reads are meant for the exchange (sticky; before the first kmr_add_reads*)
 *   kmr_sk_exchange_counts    closes the lists; chunks[r], granules[r] (host, [world_size]) = what this rank holds for owner r
 *                             (16-byte granules; r == rank: what stays)
 *   kmr_sk_exchange_pack_dev  the chunks of the other owners -> dev_data (owner r's granules from granule_offset[r] on, 16 bytes
 *                             each) and dev_meta (its (list, granules) pairs, u32 x 2, from chunk_offset[r] on); they leave the pool
 *   (the caller moves data and meta to their owners: all-to-all over RCCL / MPI)
 *   kmr_sk_exchange_adopt_dev the received chunks (any order) are appended to this rank's own lists
 * then kmr_finalize as usual.  All three synchronise. */
int kmr_sk_exchange_begin(kmr_handle *h);      /* before the first reads of the handle */
int kmr_sk_exchange_counts(kmr_handle *h, uint64_t *chunks, uint64_t *granules);
int kmr_sk_exchange_pack_dev(kmr_handle *h, void *dev_data, void *dev_meta, const uint64_t *granule_offset, const uint64_t *chunk_offset);
int kmr_sk_exchange_adopt_dev(kmr_handle *h, const void *dev_data, const void *dev_meta, uint64_t n_chunks, uint64_t n_granules);
/* Do all records of this rank's lists carry ONE weight (every call so far took the bases-only extraction with the same quality
 * character)?  *state = kind << 32 | weight bits; kind 0: no record yet, 1: one weight, 2: several.  A sender hands its state to the
 * owners along with its chunk counts, an owner folds it in with kmr_sk_exchange_peer_uniform BEFORE adopting that sender's chunks: if all
 * agree, kmr_finalize counts with the one-weight form of the count pass.  An owner that is told nothing checks the received records
 * itself (a pass over every received header).  Nothing in the reference corresponds (its wire records carry a weight per k-mer,
 * src/DistributedFunctions.h:274-303). */
int kmr_sk_exchange_uniform(kmr_handle *h, uint64_t *state);
int kmr_sk_exchange_peer_uniform(kmr_handle *h, uint64_t state);
/* An exchange in steps over the list space, so that an owner can count what has arrived while the rest is on the wire:
 * kmr_sk_exchange_range restricts the next kmr_sk_exchange_counts / kmr_sk_exchange_pack_dev to the lists in [list_lo, list_hi)
 * (0, ~0: all of them, the default; kmr_reset restores it); kmr_count_lists_prefix runs the count pass over this handle's lists below
 * list_hi NOW -- asynchronously on the handle's stream, into entry buffers of their own -- once everything those lists will ever get
 * has been adopted; kmr_finalize (same min_depth) then counts the lists from list_hi on and takes the early entries over.  The
 * result is that of kmr_finalize alone.  What cannot be counted early (extension values, a kept singleton map, the size tracker,
 * coarse lists, early buffers that turn out too small) is quietly left to kmr_finalize.  The reference's MPI build has no such
 * phase: its owners insert k-mers as messages arrive (src/DistributedFunctions.h:323-328) and purge at the end. */
int kmr_sk_exchange_range(kmr_handle *h, uint64_t list_lo, uint64_t list_hi);
int kmr_count_lists_prefix(kmr_handle *h, uint32_t min_depth, uint64_t list_hi);

/* KmerSpectrum::SizeTracker (src/KmerSpectrum.h:812-900): the history of (rawKmers, rawGoodKmers, uniqueKmers, singletonKmers) the
 * apps write as --size-history-file (apps/FilterReads.cpp:141-147) and EstimateSize fits.  The reference calls track() before every
 * k-mer it appends (trackSpectrum, :1574-1581: thread 0 only, so its parallel builds record an arbitrary subset); here the same
 * rule -- an element whenever rawKmers has reached nextToTrack (128, then x 1.05 truncated to long) -- is applied after every READ,
 * in input order: element i holds the four counters as they stand after the first read that brings rawKmers to the i-th threshold
 * (counters of a serial build of exactly the reads up to there; what differs from the reference's serial history is only where
 * inside a read the sample is taken).  Needs kmr_config.size_tracker = 1 (kept by the super-k-mer build of a single partition;
 * kmr_create refuses other combinations) and a finalized handle.  elements: [capacity][4] u64, may be NULL to ask for the count;
 * force_last = 1 appends the element trackSpectrum(true) adds after the build.  With kmer_subsample > 1 the stored values are
 * scaled as track() does (:882-887). */
int kmr_size_tracker(kmr_handle *h, int force_last, uint64_t *elements, uint64_t capacity, uint64_t *n_elements);

/* The whole exchange inside the library, RCCL called directly (librccl is dlopen'ed on first use: no link-time dependency): what a
 * C / C++ host -- one process or thread per GPU of a node, no MPI, no Python -- calls instead of
 * DistributedKmerSpectrum::_buildKmerSpectrumMPI (src/DistributedFunctions.h:340-458; the MPI_Alltoallv of src/MPIBuffer.h:588-600
 * becomes grouped ncclSend / ncclRecv over xGMI, one message of <= 1 GiB per peer and slice; the local share never moves).
 *   kmr_exchange_unique_id      rank 0 makes the job's id; the host hands the KMR_EXCHANGE_ID_BYTES to every rank (file, pipe, socket)
 *   kmr_exchange_init           collective: ncclCommInitRank(cfg.world_size, id, cfg.rank) on the handle's device (or
 *                               kmr_exchange_init_transport below: the host's own collectives).  A build_mode 0
 *                               handle that can build super-k-mer lists moves to them here (build_mode 3 semantics, see above)
 *   kmr_exchange_add_reads_dev  collective, one batch of THIS rank's reads (device pointers as kmr_add_reads_dev; n_reads may be 0):
 *                               build_mode 3: global stream ordinals (an all-gather of the batch sizes), extract into the job's
 *                               lists, counts, chunks of other owners out, received chunks appended; other modes: the batch's
 *                               k-mer records binned by getDistributedThreadId (segments grown and re-extracted if one owner takes
 *                               more than its share), counts, records out, kmr_insert_records_dev of what arrived
 *   kmr_exchange_stats          bytes sent to other ranks and the time of the all-to-alls (HIP events on the handle's stream)
 * then kmr_finalize on every rank.  kmr_destroy frees the communicator.  Every rank must make the same sequence of calls. */
#define KMR_EXCHANGE_ID_BYTES 128
struct kmr_reads;      /* a device-resident read batch, see "FASTQ ingest" below */
int kmr_exchange_unique_id(void *id);
int kmr_exchange_init(kmr_handle *h, const void *id);
/* The same driver over the host's own collectives instead of RCCL (an MPI job: MPI_Allgather and a device-aware MPI_Alltoallv, or one
 * staged through host memory; the tests plug in gloo).  Both are collective and return 0 or a KMR_ERR_* code.
 *   allgather_u64   every rank contributes n values (host memory); all[r * n + j] = value j of rank r
 *   alltoallv_dev   DEVICE buffers: send_bytes[r] bytes at send + send_off[r] go to rank r, recv_bytes[r] bytes from rank r land at
 *                   recv + recv_off[r]; the entries of the calling rank are 0; hip_stream = the handle's stream (the buffers were
 *                   written on it; the call returns when the received bytes are visible to it).  No message exceeds 1 GiB. */
typedef struct kmr_transport {
	void *user;
	int (*allgather_u64)(void *user, const uint64_t *mine, uint64_t n, uint64_t *all);
	int (*alltoallv_dev)(void *user, const void *send, const uint64_t *send_off, const uint64_t *send_bytes,
	                     void *recv, const uint64_t *recv_off, const uint64_t *recv_bytes, void *hip_stream);
} kmr_transport;
int kmr_exchange_init_transport(kmr_handle *h, const kmr_transport *t);
int kmr_exchange_add_reads_dev(kmr_handle *h, const void *dev_bases, const void *dev_quals, const void *dev_offsets, uint64_t n_reads,
                               uint64_t total_bases, uint64_t first_global_read_idx, const void *dev_discarded);
int kmr_exchange_add_read_batch(kmr_handle *h, const struct kmr_reads *batch, uint64_t first_global_read_idx);      /* the same for a device-resident read batch ("FASTQ ingest" below); NULL = no reads this round */
int kmr_exchange_stats(kmr_handle *h, uint64_t *bytes_to_peers, double *alltoall_ms);

/* For a kmr_transport that stages the device segments through host memory (an MPI without device pointers; the -P shim's
 * TRANSPORT_MPI): copies between host memory and the device buffers alltoallv_dev is handed, ordered behind the handle's stream. */
int kmr_copy_to_host(kmr_handle *h, void *host_dst, const void *dev_src, uint64_t bytes);
int kmr_copy_to_device(kmr_handle *h, void *dev_dst, const void *host_src, uint64_t bytes);

/* Host-buffer forms of the two halves for a host whose exchange is MPI_Alltoallv over host memory (the reference's own,
 * src/MPIBuffer.h:588-600; include/kmernator_amd_shim.hpp, GpuDistributedKmerSpectrum).  kmr_extract_by_owner_host: the records
 * of a device-resident batch, owner after owner without gaps; seg_counts[world_size] always receives the counts, and with
 * records == NULL that is all the call does (the segments wait on the device for the call that fetches them). */
struct kmr_reads;      /* a device-resident read batch, see "FASTQ ingest" below */
int kmr_extract_by_owner_host(kmr_handle *h, const struct kmr_reads *batch, uint64_t first_global_read_idx,
                              uint64_t *seg_counts, void *records, uint64_t capacity_bytes);
int kmr_insert_records(kmr_handle *h, const void *host_records, uint64_t n_records);

/* ---- f1, distributed form: scoreAndTrimReads when the spectrum is partitioned by owner --------
 * DistributedReadSelector::scoreAndTrimReads / _batchKmerLookup (src/DistributedFunctions.h:876-1045): every k-mer of a
 * rank's reads is looked up at its owner (request = requestId + k-mer, response = requestId + score over
 * MPI_Alltoallv).  Here a request is the key alone (8 * KMR_KEY_WORDS(k) bytes: the u64 key words, most significant
 * first) and a response a u32 count: the answers come back in request order, so the requester keeps the position of
 * each request instead of sending an id.
 *
 * kmr_lookup_requests_dev: every k-mer without markup of a device-resident read batch, binned by owner into world_size
 * segments of dev_keys (segment s starts at key seg_capacity * s); dev_pos[seg_capacity * s + j] (u32) = position of
 * request j's first base in dev_bases (offsets[r] + i; < 2^32), dev_seg_counts[world_size] (u64) the exact counts.
 * kmr_lookup_keys_dev: owner side, weak-map count of n received keys (ReadSelector::getValue, src/ReadSelector.h:924-931).
 * kmr_scatter_counts_dev: position_counts[pos[j]] = counts[j] for the answers of one owner segment.
 * kmr_score_counts_dev: trimReadByMinimumKmerScore + scoring + setTrimHeaders (as kmr_score_reads) from counts indexed
 * by base position (u32 position_counts[total_bases], zero where no answer was written); dev_bases 16-byte aligned and
 * padded by 64 bytes as for kmr_add_reads_dev.
 * All asynchronous on the handle's stream except kmr_score_counts_dev, which returns host arrays. */
int kmr_lookup_requests_dev(kmr_handle *h, const void *dev_bases, const void *dev_offsets, uint64_t n_reads, uint64_t total_bases,
                            void *dev_keys, void *dev_pos, uint64_t seg_capacity, void *dev_seg_counts);
int kmr_lookup_keys_dev(kmr_handle *h, const void *dev_keys, uint64_t n, void *dev_counts);
int kmr_scatter_counts_dev(kmr_handle *h, const void *dev_counts, const void *dev_pos, uint64_t n, void *dev_position_counts);
int kmr_score_counts_dev(kmr_handle *h, const void *dev_bases, const void *dev_offsets, uint64_t n_reads, const void *dev_position_counts,
                         double minimum_kmer_score, int scoring_type, uint32_t *trim_offset, uint32_t *trim_length,
                         float *score, uint8_t *was_trimmed);

/* ---- f2: FASTQ ingest on the device -------------------------------------
 * Parses a whole in-memory FASTQ block into a device-resident read batch:
 * FastqStreamParser::readRecord (src/ReadFileReader.h:768-835) + ReadFileReader::nextRead
 * (:296-329: Casava-1.8 failed-filter reads are dropped, bases upper-cased, #bases == #quals)
 * + ReadSet::appendFasta/addRead/validateFastqStart (src/ReadSet.cpp:136-141,311-345,
 * src/ReadSet.h:171-209): qualities are rescaled from input_quality_base (33, 64, or 0 = the
 * handle's fastq_start_char; --fastq-base-quality) to the handle's fastq_start_char, and a read
 * among the first 19 999 whose minimum quality lies outside [start, start+40] flips the input base
 * once for the whole batch, as __setFastqStart does.  store_comment = GlobalOptions::isCommentStored()
 * (the reference's default is 1).  Malformed input (what makes the reference throw, plus non-'@' junk
 * between records, which the reference skips) returns KMR_ERR_INVALID_ARG. */
typedef struct kmr_reads kmr_reads;
int kmr_ingest_fastq(kmr_handle *h, const char *text, uint64_t len, uint32_t input_quality_base,
                     int store_comment, kmr_reads **out);
int kmr_ingest_fastq_dev(kmr_handle *h, const void *dev_text, uint64_t len, uint32_t input_quality_base,
                         int store_comment, kmr_reads **out);
/* n_filtered = records dropped by the Casava filter; input_quality_base = the base after detection */
int kmr_reads_info(const kmr_reads *r, uint64_t *n_reads, uint64_t *total_bases,
                   uint32_t *input_quality_base, uint64_t *n_filtered);
/* the same kind of batch from reads the host has already parsed (arrays as for kmr_add_reads, qualities scaled to the
 * handle's fastq_start_char); names are empty */
int kmr_reads_from_host(kmr_handle *h, const char *bases, const char *quals, const uint64_t *offsets, uint64_t n_reads,
                        kmr_reads **out);
/* ... and from reads the host keeps as the reference's Read does (arrays as for kmr_add_reads_twobit below: 2-bit packed bases, markups,
 * qualities as characters / one character for all / none = Read::REF_QUAL): a quarter of the base bytes on the bus */
int kmr_reads_from_twobit(kmr_handle *h, const uint8_t *twobit, const uint64_t *twobit_offsets, const uint64_t *offsets,
                          const uint64_t *markup_offsets, const uint32_t *markup_pos, const char *markup_char,
                          const char *quals, int uniform_quality, uint64_t n_reads, kmr_reads **out);
/* device arrays in the layout kmr_add_reads_dev takes: bases[total], quals[total], u64 offsets[n+1] */
int kmr_reads_device_ptrs(const kmr_reads *r, void **dev_bases, void **dev_quals, void **dev_offsets);
/* copies to host; any pointer may be NULL.  name_off/name_len: span of each read's name line
 * (after '@') in the input text, for the host-side Read names */
int kmr_reads_copy(const kmr_reads *r, char *bases, char *quals, uint64_t *offsets,
                   uint64_t *name_off, uint32_t *name_len);
/* The batch in the form the reference's Read keeps its bases (src/Sequence.h: 2-bit packed + markups):
 * TwoBitSequence::compressSequence (src/TwoBitSequence.cpp:242-269) over every read on the device.  Read i's packed bases are
 * twobit[twobit_offsets[i] .. twobit_offsets[i+1]) (ceil(L/4) bytes, first base in bits 7-6, last byte zero padded), its markups
 * (any character but ACGTacgt; '.' recorded as 'N') are entries markup_offsets[i] .. markup_offsets[i+1] of markup_pos /
 * markup_char, in ascending offset.  *twobit_bytes and *n_markups always receive the sizes; with every array NULL that is all
 * the call does, and KMR_ERR_CAPACITY means an array was too small for them.  All arrays are host memory. */
int kmr_reads_twobit(kmr_handle *h, const kmr_reads *r, uint8_t *twobit, uint64_t twobit_capacity, uint64_t *twobit_offsets,
                     uint32_t *markup_pos, char *markup_char, uint64_t markup_capacity, uint64_t *markup_offsets,
                     uint64_t *twobit_bytes, uint64_t *n_markups);
/* Feed a batch in the form the reference's Read keeps it (src/Sequence.h:166-171,287-289: bases as a TwoBitSequence -- 2 bits per base, every
 * read on bytes of its own, first base in bits 7-6 -- plus markups for everything that is not ACGT, qualities as characters or
 * none): what a ReadSet hands over without a string per read, and a quarter of the bytes of kmr_add_reads on the way to the
 * device.  Replaces Read::getFasta / getQuals in front of KmerReadUtils::buildWeightedKmers (src/KmerReadUtils.h:176-190) and
 * TwoBitSequence::uncompressSequence + applyMarkup (src/TwoBitSequence.cpp:286-340), which run on the device.
 *   twobit / twobit_offsets   packed bases, read i = bytes [twobit_offsets[i], twobit_offsets[i+1]) (>= ceil(L_i / 4) of them; the layout
 *                             kmr_reads_twobit writes); the device form takes NULL offsets when every read starts on the byte
 *                             behind the one before it
 *   offsets                   n_reads + 1 base offsets (read i has offsets[i+1] - offsets[i] bases), as for kmr_add_reads
 *   markup_offsets / _pos / _char   optional: read i's markups are entries [markup_offsets[i], markup_offsets[i+1]), position and character
 *   quals                     qualities indexed by offsets (host form; the device form: dev_quals[0] is the quality of the call's first
 *                             base), or NULL and uniform_quality = the ONE quality character every base of the batch has (0: reads
 *                             without qualities, weight 1.0 as kmr_add_reads with quals == NULL)
 * Results are those of kmr_add_reads on the same reads.  The device form is asynchronous on the handle's stream like kmr_add_reads_dev.
 * A batch without a quality array (one character, or none) that is built on the super-k-mer lists with direction-counting values is
 * extracted from the packed bytes as they are; every other batch is unpacked to text in a scratch of the handle first.  kmr_tune
 * "packed_direct" = 0 forces the unpack (tests, A/B runs).
 * Device form: the packed bytes are fetched 16 bytes at a time from 16-byte aligned addresses, so up to 15 bytes in front of the
 * first read's first byte and behind the last read's last byte are READ (never used): dev_twobit must lie in an allocation that
 * holds them -- as hipMalloc'ed buffers do in front (256-byte aligned), and give the buffer 64 spare bytes behind the last read, as
 * for kmr_add_reads_dev. */
int kmr_add_reads_twobit(kmr_handle *h, const uint8_t *twobit, const uint64_t *twobit_offsets, const uint64_t *offsets,
                         const uint64_t *markup_offsets, const uint32_t *markup_pos, const char *markup_char,
                         const char *quals, int uniform_quality, uint64_t n_reads, uint64_t first_global_read_idx, const uint8_t *discarded);
int kmr_add_reads_twobit_dev(kmr_handle *h, const void *dev_twobit, const void *dev_twobit_offsets, const void *dev_offsets,
                             const void *dev_markup_offsets, const void *dev_markup_pos, const void *dev_markup_char,
                             const void *dev_quals, int uniform_quality, uint64_t n_reads, uint64_t total_bases,
                             uint64_t first_global_read_idx, const void *dev_discarded);
/* kmr_add_reads_dev on the batch, then kmr_sync */
int kmr_add_read_batch(kmr_handle *h, const kmr_reads *r, uint64_t first_global_read_idx);
void kmr_reads_free(kmr_reads *r);
/* kmr_score_reads on a device-resident read batch: FASTQ text -> reads -> spectrum -> trim / score without the host
 * staging the reads (outputs are host arrays of kmr_reads_info's n_reads entries) */
int kmr_score_read_batch(kmr_handle *h, const kmr_reads *r, double minimum_kmer_score, int scoring_type,
                         uint32_t *trim_offset, uint32_t *trim_length, float *score, uint8_t *was_trimmed);

/* ---- f4: artifact filter (FilterKnownOddities, src/FilterKnownOddities.h) ------------------
 * The screen FilterReads runs over every read before the spectrum build (apps/FilterReads.cpp:107-118):
 * (1) the longest run of bases with quality >= start + min_quality ("best"; the runner-up may be rescued as a
 * remnant read), (2) every 4th match_length-mer inside it looked up (canonical form) in a set made of all
 * match_length-mers of the artifact sequences, circularised, plus their substitution neighbours, (3) keep the
 * longer side of the read next to the hits, then discard or trim (recordAffectedRead :551-640).
 *
 * The artifact sequences are handed in as FASTA text (multi-line records allowed; the reference's are
 * FilterKnownOddities::getArtifactFasta/getSimpleRepeatFasta/getPhiX and --artifact-reference-file). Sequence
 * indices are 1-based in file order (0 is the reference's empty "no match" read); value n_sequences marks a
 * read that only lost bases to the quality screen ("MinQualityTrim"). */
typedef struct kmr_artifact_config {
	uint32_t match_length;          /* --artifact-match-length (24): multiple of 4, <= 28                             */
	uint32_t edit_distance;         /* --artifact-edit-distance (2)                                                   */
	uint32_t build_edits;           /* --build-artifact-edits-in-filter (2): 0 never, 1 always, 2 while < 750 000 keys */
	uint32_t simple_repeat_begin;   /* [begin, end) = the simple-repeat sequences (--mask-simple-repeats); 0,0 = none */
	uint32_t simple_repeat_end;
	uint32_t phix_idx;              /* the PhiX sequence (--phix-output); 0 = none                                    */
	uint32_t reference_begin;       /* first --artifact-reference-file sequence (not circularised); 0 = none          */
	uint32_t min_quality;           /* --min-quality-score                                                            */
	uint32_t fastq_start_char;      /* Read::FASTQ_START_CHAR the reads are scaled to                                 */
	float    min_read_length;       /* --min-read-length (fraction of the read if <= 1, bases otherwise)              */
} kmr_artifact_config;
void kmr_artifact_config_init(kmr_artifact_config *c);      /* the reference's defaults, start char 33, min quality 3, min length 0.40 */

typedef struct kmr_artifact_filter kmr_artifact_filter;
/* constructor + prepareMaps (:205-287); the key set lives in device memory */
int kmr_artifact_filter_create(kmr_handle *h, const kmr_artifact_config *cfg, const char *fasta, uint64_t len,
                               kmr_artifact_filter **out);
/* sequences.getSize(), filter.size(), and the edits left for query time (numErrors after prepareMaps) */
int kmr_artifact_filter_info(const kmr_artifact_filter *f, uint64_t *n_sequences, uint64_t *n_filter_kmers,
                             uint32_t *remaining_edits);
/* copies the (key, sequence index) pairs out in ascending key order (key = 2-bit packed match_length-mer,
 * first base most significant); returns KMR_ERR_CAPACITY if cap is too small */
int kmr_artifact_filter_entries(const kmr_artifact_filter *f, uint64_t *keys, uint32_t *values, uint64_t cap);
void kmr_artifact_filter_free(kmr_artifact_filter *f);
/* applyFilter (:663-733) over a device-resident read batch.  mate (host, n entries, may be NULL) = index of the
 * paired read or -1 (applyFilterToPair :355-386).  Per read (host arrays, each may be NULL):
 * value (matched sequence, n_sequences = quality trim only, 0 = clean), [min_pass, max_pass) the part to keep,
 * action 0 = untouched / 1 = trimmed ("AFTrim:<min_pass>+<max_pass-min_pass>") / 2 = discarded, and
 * [remnant_off, +remnant_len) the second-best quality run rescued as an extra read (len 0 = none).
 * *out (may be NULL) receives a new batch: read i trimmed or emptied in place, the remnants appended in read
 * order (the reference appends them per OpenMP thread). */
int kmr_artifact_filter_apply(kmr_handle *h, const kmr_artifact_filter *f, const kmr_reads *in, const int64_t *mate,
                              uint32_t *value, uint32_t *min_pass, uint32_t *max_pass, uint8_t *action,
                              uint32_t *remnant_off, uint32_t *remnant_len, kmr_reads **out);

/* Raw HIP stream of the handle (hipStream_t) so callers can order their own
 * work (torch.cuda.ExternalStream) against it. */
void *kmr_stream(kmr_handle *h);

/* Implementation knobs of one handle; none of them changes a result.  They exist so that tests reach the multi-level,
 * retry and sub-batch code with small inputs and measurement tools can sweep a parameter; the library reads no
 * environment variable.  Knobs (value): "target_list_records" (records per final list the partition bits aim for, 2048),
 * "sub_batch_bases" (bases per extract launch, 0 = default), "recycle_chunks" (-1 auto, 0, 1), "partition_blocks"
 * (0 = one per CU), "entry_share" (initial entry-buffer share of the count pass, < 0 = from the probe), "lookup_table",
 * "narrow_tallies", "keep_level1_state" (1 / 0), "superkmer_minimizer" (minimizer length of build_mode 3, 0 = default),
 * "stream_lookups" (1: kmr_score_* gets its k-mer counts from a streaming pass over minimizer lists where k allows it; 0: per-k-mer
 * probes of the lookup table), "long_list_chunks" (super-k-mer lists of more 1 KB chunks are counted / looked up in pieces, 1024), "coarse_lists" (owner exchange:
 * 1 = scatter into and exchange coarse lists that the owner splits before the count pass, 0 = the job's fine lists; default 0).
 *   "pow2_lists" (1: the list count of build_mode 3 is always a power of two; default 0: a single GPU's build takes est / list_aim lists),
 *   "list_aim" (k-mers per list that count aims for; default 1450 when every k-mer weighs the same, 1700 raw k-mers otherwise, 800 with
 *   extension values, just below the table's bound for keys of more than one word), "packed_direct" (0: kmr_add_reads_twobit* always
 *   unpack to text first).
 *   "uniform_count" (0: never the one-weight form of the count pass), "lean_extract" (0: never the bases-only extraction),
 *   "superkmer_window" (the widest minimizer window build_mode 3 may take: 32 (k >= 45) / 16 / 8 / 4).
 * Call before the first kmr_add_reads* of a build.  KMR_ERR_INVALID_ARG for an unknown knob. */
int kmr_tune(kmr_handle *h, const char *knob, double value);
/* What the current build decided, for tests and measurement tools (the reference logs such figures, LOG_VERBOSE): "lists" = super-k-mer
 * lists of the build (0 before the first reads, or in another build mode), "uniform_count" = 1 if the last kmr_finalize ran the
 * count pass's one-weight form, "chunk_pool_chunks" = 1 KB chunks the pool holds, "superkmer_window" = the minimizer window in use, "early_lists" / "early_entries" = the bound below which the last
 * kmr_finalize took its lists' entries from kmr_count_lists_prefix (0: it counted everything itself) and how many entries those were.  KMR_ERR_INVALID_ARG for an unknown name. */
int kmr_build_info(kmr_handle *h, const char *what, double *value);

/* Timing of the hot path measured with HIP events on the handle's stream
 * (used by bench.py for the roofline object).  Returns the accumulated
 * milliseconds and the number of timed launch groups since the last reset.
 * Groups 0 and 1 span whole phases, the others single kernels of the
 * streaming build path (they nest inside 0 and 1). */
enum kmr_time_group {
	KMR_TIME_BUILD = 0,       /* kmr_add_reads*: extract + insert / level-1 partition, per sub-batch */
	KMR_TIME_FINALIZE = 1,    /* kmr_finalize: everything                                           */
	KMR_TIME_EXTRACT = 2,     /* extract_kernel launches                                             */
	KMR_TIME_PARTITION1 = 3,  /* level-1 partition launches                                          */
	KMR_TIME_PARTITION2 = 4,  /* level-2 partition launch                                            */
	KMR_TIME_COUNT = 5,       /* count pass                                                          */
	KMR_TIME_BUCKETS = 6,     /* bucket scan + entry scatter + bucket sort                           */
	KMR_TIME_EXCHANGE = 7,    /* kmr_exchange_add_reads_dev: the all-to-all of list chunks / k-mer records (RCCL)   */
	KMR_TIME_GROUPS = 8
};
int kmr_kernel_time(kmr_handle *h, int which, double *ms, uint64_t *launches);
int kmr_kernel_time_reset(kmr_handle *h);

uint32_t kmr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KMERNATOR_AMD_H_ */
