/*
 * kmr_api.hip -- C-ABI of include/kmernator_amd.h on top of the HIP kernels.
 *
 * Host logic only: handle life cycle, device memory, launches on the handle's
 * stream, growth of the device table, finalize (bucket histogram -> scan ->
 * scatter -> sort -> image) and the stateless helpers.  There is no CPU
 * fallback: every compute entry point needs a HIP device and fails with
 * KMR_ERR_NO_DEVICE otherwise.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <memory>
#include <string>
#include <vector>

#include "../../include/kmernator_amd.h"
#include "kmr_kernels.hpp"
#include "kmr_partition.hpp"
#include "kmr_ingest.hpp"
#include "kmr_artifact.hpp"
#include "kmr_superkmer.hpp"
#include "kmr_buckets.hpp"
#include "kmr_synth.hpp"
#define KMR_INSTANCES_EXTERN
#include "kmr_instances.hpp"      /* the heavy kernels are compiled in kmr_inst_*.hip */

using namespace kmr;

namespace {

std::string g_create_error;

struct DevMap {                      /* a finalized map resident in HBM */
	uint64_t nb = 0, n = 0;
	uint64_t *start = nullptr;       /* [nb+1] */
	uint64_t *keys = nullptr;        /* [n][W] */
	uint32_t *vals = nullptr;        /* weak: [n][vw] */
	uint8_t *sweight = nullptr;      /* singleton */
	uint32_t *spkt = nullptr;        /* singleton, EXT */
	uint8_t *image = nullptr;        /* reference layout, built lazily */
	uint64_t image_bytes = 0;
	bool present = false;
	/* bytes allocated behind start / keys / vals / sweight: kmr_reset() keeps the buffers of the streaming path for the next build */
	size_t c_start = 0, c_keys = 0, c_vals = 0, c_sw = 0, c_pkt = 0;
};

struct HostPool {                    /* owner of one chunk pool */
	uint8_t *base = nullptr; uint32_t *chunk_list = nullptr, *chunk_count = nullptr; unsigned int *head = nullptr;
	uint32_t cap = 0; size_t chunk_bytes = 0;
	uint64_t used_ub = 0;            /* host-side upper bound of chunks handed out */
	uint64_t presize = 0;            /* chunks the next allocation takes beyond what is asked for (a job fed in many calls, see sk_add_reads) */
};

}  // namespace

/* per-handle knobs of kmr_tune(): sizes the tests shrink to reach the multi-level / retry / sub-batch code with small inputs, and
 * switches the measurement tools flip.  None of them changes a result. */
struct Tuning {
	uint64_t target_list = 2048;      /* records per final list the partition bits aim for */
	uint64_t sub_batch_bases = 0;     /* 0 = SUB_BATCH_BASES */
	int recycle = -1;                 /* -1 auto, 0 fresh chunks, 1 recycle the chunks a pass has just read */
	int part_blocks = 0;              /* 0 = one partition block per CU */
	double entry_share = -1.0;        /* >= 0: initial size of the count pass's entry buffers as a share of the records */
	bool no_lut = false, no_narrow = false, no_l1_state = false, no_stream_lookups = false;
	uint64_t long_list_chunks = 0;     /* lists of more chunks are counted in pieces (0: 1024) */
	uint64_t binned_min = 1ull << 18;  /* weak maps of at least this many entries are bucketed by the radix partition of kmr_buckets.hpp (build_mode 3) */
	uint64_t twobit_piece_bases = 0;   /* kmr_add_reads_twobit: bases per piece of the host-to-device pipeline (0 = 2^26) */
	uint64_t list_aim = 0;             /* k-mers per list the list count of a single GPU's build aims for (0: the defaults of add_reads_superkmer_t) */
	bool pow2_lists = false;           /* the list count of build_mode 3 always a power of two (A/B runs, tests of both list functions) */
	bool no_packed_direct = false;     /* kmr_add_reads_twobit* always unpack to text first (A/B runs, tests of the unpack path) */
	bool no_uniform_count = false;     /* never take sk_count_kernel<.., UNI> (A/B runs, tests of the general count pass on one-weight builds) */
	bool no_lean_extract = false;      /* never take sk_extract_lean_kernel (A/B runs, tests of the general kernel on uniform qualities) */
	bool exchange_fail_once = false;   /* tests: the next kmr_exchange_add_reads_dev of this rank fails locally (the other ranks must come back with an error, not hang) */
	bool no_coarse_lists = true;       /* exchange: scatter into the job's fine lists (default) or, kmr_tune("coarse_lists", 1), into coarse ones that the owner splits before the count pass (sk_refine_kernel: not yet fast enough to pay) */
};

struct kmr_handle {
	kmr_config cfg;
	Tuning tune;
	uint32_t k = 0, kb = 0, hkb = 0, W = 0;
	bool ext = false;
	int device = 0, ncu = 0;
	hipStream_t stream = nullptr;
	std::string err;
	/* device table */
	void *slots = nullptr;
	ExtSlot *extslots = nullptr;
	uint32_t log2cap = 0;
	uint64_t occupied = 0;           /* exact as of the last sync */
	uint64_t pending_kmers = 0;      /* upper bound of keys added since */
	double *dP = nullptr;
	DevStats *dstats = nullptr;
	uint32_t *derr = nullptr;
	uint64_t stream_base = 0, reads = 0;
	uint64_t nb_weak = 0, nb_sing = 0;
	bool finalized = false, has_singletons = true;
	kmr_handle *subtract = nullptr;    /* finalized spectrum whose k-mers are skipped (kmr_subtract_reference) */
	uint64_t subtracted = 0;
	DevMap weak, sing;
	kmr_stats stats;
	/* streaming (partition) build path */
	bool partition_mode = false;
	bool superkmer_mode = false;       /* build_mode 3: super-k-mer lists (kmr_superkmer.hpp); implies partition_mode */
	/* streaming lookups (sk_index_* / sk_lookup_kernel): the weak map's entries grouped by minimizer list, of map generation ix_gen */
	uint64_t *ix_start = nullptr, *ix_keys = nullptr; uint32_t *ix_counts = nullptr; uint64_t ix_cap = 0, ix_lists = 0, ix_gen = ~0ull;
	DevStats *scratch_stats = nullptr;
	uint8_t *adopt_buf = nullptr; size_t adopt_cap = 0;      /* kmr_sk_exchange_adopt_dev's scan */
	/* size tracker (kmr_config.size_tracker): one record per read fed so far, and the elements made of them at kmr_finalize */
	SkTrackRec *trk = nullptr; uint64_t trk_cap = 0, trk_n = 0; std::vector<uint64_t> trk_elems;
	/* the thresholds passed so far (SizeTracker::nextToTrack and the elements' first two counters), found call by call while the
	 * reads are still at hand: the stream ordinal behind the k-mer at which rawKmers reached the threshold, rawKmers, rawGoodKmers */
	long trk_next = 128; uint64_t trk_raw = 0, trk_good = 0; std::vector<unsigned long long> trk_bounds; std::vector<uint64_t> trk_snap_raw, trk_snap_good;
	bool sk_fast_div = false;          /* see kmr_create: the chain's divide as multiply-and-correct */
	bool sender_launch = false;        /* extract_by_owner_t, build (not request) mode: dev_params tells the kernel to count what it does not send */
	bool sk_exchange = false;          /* kmr_sk_exchange_begin: the lists are the whole job's, every owner's k-mers are kept until the exchange */
	bool auto_mode = false;            /* build_mode 0: a handle that is fed k-mer records (the owner exchange) before any reads falls back to mode 2 */
	HostPool l1;                       /* the record pool of every partition level */
	int bits1 = 0;
	uint64_t inserted_records = 0;     /* records fed through kmr_insert_records_dev (counted on the host) */
	uint64_t call_bases_hint = 0;      /* a host batch goes to the device in pieces: the bases of the WHOLE call, for what the first piece sizes (lists, chunk pool) */
	unsigned int *work_counter = nullptr;
	uint8_t *l1_state = nullptr; size_t l1_state_bytes = 0; bool l1_state_dirty = false;   /* see PartSource::state */
	/* temporaries of kmr_finalize (chunk CSRs, work items, counters): one grow-only block handed out by bumping a
	 * cursor, so a finalize neither allocates nor frees device memory once the handle has seen one build */
	uint8_t *arena = nullptr; size_t arena_cap = 0, arena_used = 0, arena_want = 0; std::vector<void *> arena_overflow;
	unsigned long long *scan_sums = nullptr; uint64_t scan_sums_n = 0;
	uint8_t *score_buf = nullptr; size_t score_buf_bytes = 0;        /* temporaries of kmr_score_reads*, grow-only */
	uint64_t *lut = nullptr; size_t lut_bytes = 0; uint32_t lut_log2 = 0;      /* lookup accelerator over the weak map (LutView) */
	uint64_t lut_gen = ~0ull, map_gen = 0;                           /* the table belongs to the maps of generation lut_gen */
	void *linear = nullptr; uint64_t linear_cap = 0;         /* records */
	uint32_t *tile_count = nullptr; uint64_t tile_cap = 0;
	uint32_t *kcap = nullptr; uint64_t *koff = nullptr; uint64_t kcap_n = 0, koff_n = 0;
	/* work units of batches that contain reads longer than one tile */
	uint32_t *ucnt = nullptr; uint64_t *ufirst = nullptr, *u_start = nullptr, *u_end = nullptr, *u_read = nullptr;
	uint64_t ucnt_n = 0, ufirst_n = 0, units_n = 0; unsigned int *umax = nullptr;
	/* kmr_add_reads_twobit*: the unpacked batch (ASCII bases, one quality character throughout, offsets counted from the call's first read) */
	uint8_t *tb_bases = nullptr, *tb_quals = nullptr; uint64_t *tb_rel = nullptr, *tb_off = nullptr; uint32_t *tb_len = nullptr;
	uint64_t tb_bases_cap = 0, tb_quals_cap = 0, tb_quals_filled = 0, tb_n = 0; int tb_quals_char = -1;
	hipStream_t tb_copy_stream = nullptr; hipEvent_t tb_ready[2] = {nullptr, nullptr}, tb_consumed[2] = {nullptr, nullptr}; bool tb_set_used[2] = {false, false};
	uint8_t *tb_stage[2][8] = {{nullptr}}; size_t tb_stage_cap[2][8] = {{0}};      /* kmr_add_reads_twobit: two sets of staging buffers for the pieces on the bus */
	const SkPacked *packed_direct = nullptr;      /* set while kmr_add_reads_twobit_dev feeds a batch that sk_extract_lean_kernel<.., PACKED> takes as it is */
	int uniform_q_hint = -1;           /* >= 0 while kmr_add_reads_twobit_dev feeds a batch whose qualities are this one character */
	void *uw_keys = nullptr, *uw_vals = nullptr, *us_keys = nullptr, *us_b8 = nullptr, *us_pkt = nullptr; uint64_t uw_cap = 0, us_cap = 0;
	/* build_mode 3: the count pass's weak entries packed (kmr_buckets.hpp: W key words + one value word), and the radix partition's scratch of the same layout */
	uint64_t *ue = nullptr, *ue2 = nullptr; uint64_t ue_cap = 0, ue2_cap = 0;
	/* build_mode 3 (kmr_superkmer.hpp): list words, minimizer geometry, table of k-fold quality products */
	unsigned long long *sk_state = nullptr; uint32_t sk_bits = 0, sk_m = 0, sk_off = 0, sk_win = 0; double *dPk = nullptr;
	double hP[256], hPk[256];              /* host copies of the probability table and of its k-fold products */
	/* does every record of the lists carry ONE weight (all calls went through the lean extraction with the same quality character)?  The
	 * host knows for its own calls (sk_uni_w: SK_UNI_NONE before the first; sk_uni_mixed), a device pair collects it for adopted records */
	uint32_t sk_uni_w = 0xffffffffu; bool sk_uni_mixed = false; uint32_t *d_uni = nullptr; bool last_count_uniform = false;
	/* ... or the senders say so themselves (kmr_sk_exchange_peer_uniform): then nothing is looked at on arrival */
	uint32_t peer_uni_w = 0xffffffffu; bool peer_uni_mixed = false, peers_declare = false;
	/* an exchange in steps over the list space (kmr_sk_exchange_range) and the lists below `hi` counted early (kmr_count_lists_prefix):
	 * their entries wait in buffers of their own until kmr_finalize has counted the rest */
	uint64_t xr_lo = 0, xr_hi = ~0ull;
	struct Early { bool active = false; uint64_t hi = 0; uint32_t min_depth = 0; uint64_t *ue = nullptr; uint64_t cap = 0; unsigned long long *cursor = nullptr; void *fc = nullptr; } early;
	uint64_t last_early_hi = 0, last_early_entries = 0;      /* what the last kmr_finalize took over from an early count (kmr_build_info) */
	unsigned int *qrange = nullptr; bool qual_mixed = false;      /* sk_qual_range_kernel's answer; a build that has seen two different quality characters stops asking */
	/* exchange with world_size > 1: sk_bits are the COARSE lists reads are scattered into and that travel; each holds 2^sk_fine_shift
	 * fine lists, made by sk_refine_kernel before the count pass (fine state: sk_fine_state, 2^(sk_bits + sk_fine_shift) words) */
	uint32_t sk_fine_shift = 0; unsigned long long *sk_fine_state = nullptr; uint64_t sk_fine_cap = 0;
	uint32_t sk_min_override = 0;
	/* kmr_extract_by_owner_host: owner segments of one batch kept on the device between the sizing call and the copy-out */
	/* kmr_exchange_* (kmr_exchange_rccl.hpp): communicator, gather scratch, grow-only send / receive buffers, what the job was fed so far */
	void *xc_comm = nullptr; unsigned long long *xc_small = nullptr, *xc_dcounts = nullptr; kmr_transport xc_tr = {nullptr, nullptr, nullptr};
	void *xc_send = nullptr, *xc_send2 = nullptr, *xc_recv = nullptr, *xc_recv2 = nullptr;
	uint64_t xc_send_cap = 0, xc_send2_cap = 0, xc_recv_cap = 0, xc_recv2_cap = 0, xc_job_bases = 0, xc_bytes_to_peers = 0;
	void *xo_dev = nullptr; uint64_t xo_segcap = 0; std::vector<uint64_t> xo_counts; const void *xo_batch = nullptr; uint64_t xo_first = 0;
	/* timing */
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	double ms[KMR_TIME_GROUPS] = {0};
	uint64_t launches[KMR_TIME_GROUPS] = {0};
	std::vector<std::pair<hipEvent_t, hipEvent_t> > pending_events[KMR_TIME_GROUPS];
};

/* device-resident read batch produced by kmr_ingest_fastq* */
struct kmr_reads {
	int device = 0;
	uint8_t *bases = nullptr, *quals = nullptr;   /* 64 bytes of padding behind the data: extract_kernel stages 16-byte blocks */
	uint64_t *offsets = nullptr;                  /* [n + 1] */
	uint64_t *name_off = nullptr; uint32_t *name_len = nullptr;
	uint64_t n = 0, total = 0, filtered = 0;
	uint32_t input_base = 0;
};

namespace {

/* Every device allocation of the library goes through dev_malloc.  An allocation that fails for lack of memory although the card
 * as a whole could hold it is tried again for a bounded time (memory another handle or torch has just freed is handed back by the
 * driver with a delay, and work still running on other streams may hold what it is about to free); what was asked for and what the
 * device had is kept for the error text (oom_note), so that a KMR_ERR_OOM says how far off it was. */
thread_local char g_oom_note[160] = "";
hipError_t dev_malloc(void **p, size_t bytes) {
	hipError_t e = hipMalloc(p, bytes);
	if (e != hipErrorOutOfMemory) return e;
	size_t fr = 0, tot = 0;
	for (int attempt = 0; attempt < 6; attempt++) {
		(void)hipGetLastError();
		hipDeviceSynchronize();
		if (hipMemGetInfo(&fr, &tot) != hipSuccess || bytes > tot) break;
		usleep(20000u << attempt);      /* 20 ms ... 640 ms: 1.3 s at most */
		e = hipMalloc(p, bytes);
		if (e != hipErrorOutOfMemory) return e;
	}
	(void)hipGetLastError();
	hipMemGetInfo(&fr, &tot);
	snprintf(g_oom_note, sizeof(g_oom_note), " [requested %.3f GB; device has %.3f GB free of %.3f GB]", bytes / 1e9, fr / 1e9, tot / 1e9);
	*p = nullptr;
	return hipErrorOutOfMemory;
}
std::string oom_note() { std::string s(g_oom_note); g_oom_note[0] = 0; return s; }

std::string hip_err_text(hipError_t e) { return std::string(hipGetErrorString(e)) + (e == hipErrorOutOfMemory ? oom_note() : std::string()); }

#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
	(h)->err = std::string(#call) + ": " + hip_err_text(e_); \
	return e_ == hipErrorOutOfMemory ? KMR_ERR_OOM : KMR_ERR_HIP; } } while (0)

/* diagnostics on stderr and the measurement-only hooks exist in a -DKMR_DEBUG_HOOKS build alone: the shipped library reads no
 * environment variable */
#ifdef KMR_DEBUG_HOOKS
bool dbg() { static const bool d = getenv("KMR_DEBUG") != nullptr; return d; }
#else
constexpr bool dbg() { return false; }
#endif

int fail(kmr_handle *h, int code, const std::string &msg) { const std::string m = code == KMR_ERR_OOM ? msg + oom_note() : msg; if (h) h->err = m; else g_create_error = m; return code; }

uint64_t min_pow2(uint64_t n) {   /* BucketExposedMapLogic::getMinPowerOf2, src/Kmer.h:2199-2212 */
	uint64_t p = n;
	if (p == 0) p = 1;
	else if ((p & (p - 1)) != 0) { p--; for (size_t i = 1; i < 64; i <<= 1) p |= p >> i; p++; }
	return p;
}
uint64_t resize_buckets(uint64_t n) { if (n > 67108864ull) n = 67108864ull; return min_pow2(n); }   /* :2224-2229 */

/* Read::initializeQualityToProbability (src/Sequence.cpp:522-540) as a function of the Phred value;
 * the reference rescales reads to base 33 first (ReadSet.cpp:324-337) */
void quality_table(double P[256], unsigned minQ, unsigned startChar) {
	for (int raw = 0; raw < 256; raw++) {
		int i = raw - (int)startChar + 33;
		if (i < 33 + (int)minQ) P[raw] = 0.0;
		else if (i < 103) P[raw] = 1.0 - pow(10.0, ((33 - i) / 10.0));
		else P[raw] = 1.0;
	}
}

size_t slot_bytes(uint32_t W) {
	switch (W) { case 1: return sizeof(Slot<1>); case 2: return sizeof(Slot<2>); case 3: return sizeof(Slot<3>); default: return sizeof(Slot<4>); }
}

DevParams dev_params(kmr_handle *h) {
	DevParams p;
	p.k = h->k; p.kb = h->hkb; p.min_weight = h->cfg.min_weight; p.fastq_start = h->cfg.fastq_start_char; p.ext_min_q = h->cfg.ext_min_quality;
	p.qzero = h->cfg.fastq_start_char + std::max<uint32_t>(1u, h->cfg.min_quality_score);   /* Q0 has probability 0 too */
	p.subsample = h->cfg.kmer_subsample; p.rank = h->cfg.rank; p.world = h->cfg.world_size; p.num_parts = h->cfg.num_parts; p.part_idx = h->cfg.part_idx;
	p.count_sender_bad = h->sender_launch ? 1u : 0u;
	p.P = h->dP; p.stats = h->dstats; p.err = h->derr;
	p.sub_wstart = p.sub_wkeys = p.sub_sstart = p.sub_skeys = nullptr; p.sub_wvals = nullptr; p.sub_sweight = nullptr; p.sub_wnb = p.sub_snb = 0; p.sub_vw = 0;
	if (h->subtract) {
		const kmr_handle *o = h->subtract;
		if (o->weak.present && o->weak.n) { p.sub_wstart = o->weak.start; p.sub_wkeys = o->weak.keys; p.sub_wvals = o->weak.vals; p.sub_wnb = o->weak.nb; p.sub_vw = o->ext ? 15 : 3; }
		if (o->sing.present && o->sing.n) { p.sub_sstart = o->sing.start; p.sub_skeys = o->sing.keys; p.sub_sweight = o->sing.sweight; p.sub_snb = o->sing.nb; }
	}
	return p;
}

int grid_for(uint64_t n, int block = 256, int maxBlocks = 256 * 16) {
	uint64_t g = (n + block - 1) / block;
	if (g < 1) g = 1;
	if (g > (uint64_t)maxBlocks) g = maxBlocks;
	return (int)g;
}

template <int W> Table<W> table_of(kmr_handle *h) { Table<W> t; t.slots = (Slot<W> *)h->slots; t.ext = h->extslots; t.log2cap = h->log2cap; return t; }

template <int W> int clear_table(kmr_handle *h, void *slots, ExtSlot *ext, uint32_t log2cap) {
	hipLaunchKernelGGL(table_clear_kernel<W>, dim3(grid_for(1ull << log2cap)), dim3(256), 0, h->stream, (Slot<W> *)slots, ext, 1ull << log2cap);
	HIPCHK(h, hipGetLastError());
	return 0;
}
int clear_table_any(kmr_handle *h, void *slots, ExtSlot *ext, uint32_t log2cap) {
	switch (h->W) { case 1: return clear_table<1>(h, slots, ext, log2cap); case 2: return clear_table<2>(h, slots, ext, log2cap);
	case 3: return clear_table<3>(h, slots, ext, log2cap); default: return clear_table<4>(h, slots, ext, log2cap); }
}

int alloc_table(kmr_handle *h, uint32_t log2cap, void **slots, ExtSlot **ext) {
	*slots = nullptr; *ext = nullptr;
	HIPCHK(h, dev_malloc(slots, slot_bytes(h->W) << log2cap));
	if (h->ext) { hipError_t e = dev_malloc((void **)ext, sizeof(ExtSlot) << log2cap); if (e != hipSuccess) { hipFree(*slots); *slots = nullptr; h->err = "dev_malloc(ext slots)"; return KMR_ERR_OOM; } }
	return clear_table_any(h, *slots, *ext, log2cap);
}

/* read the device error word and counters; synchronises the stream */
int sync_state(kmr_handle *h) {
	HIPCHK(h, hipStreamSynchronize(h->stream));
	for (int which = 0; which < KMR_TIME_GROUPS; which++) {
		for (auto &pr : h->pending_events[which]) {
			float ms = 0; hipEventElapsedTime(&ms, pr.first, pr.second);
			h->ms[which] += ms; h->launches[which]++;
			hipEventDestroy(pr.first); hipEventDestroy(pr.second);
		}
		h->pending_events[which].clear();
	}
	uint32_t e = 0; DevStats s;
	HIPCHK(h, hipMemcpy(&e, h->derr, sizeof(e), hipMemcpyDeviceToHost));
	HIPCHK(h, hipMemcpy(&s, h->dstats, sizeof(s), hipMemcpyDeviceToHost));
#ifdef KMR_DEBUG_HOOKS
	if (h->superkmer_mode) { if (dbg()) fprintf(stderr, "sk_extract windows: %llu general, %llu fast\n", s.claimed, s.inserted); s.claimed = 0; s.inserted = 0; }
#endif
	/* (through the k-mer record exchange: good k-mers are counted where they arrive, bad ones where they were read) */
	h->stats.raw_kmers = s.raw + s.inserted + s.sender_bad; h->stats.raw_good_kmers = s.good + s.inserted; h->stats.discarded = s.raw - s.good + s.sender_bad;
	h->occupied = s.claimed; h->pending_kmers = 0; h->subtracted = s.subtracted;
	if (e & ERR_READ_TOO_LONG) return fail(h, KMR_ERR_UNSUPPORTED, "a read is longer than the per-wavefront LDS tile (" + std::to_string(TILE_SPAN) + " bases)");
	if (e & ERR_TABLE_FULL) return fail(h, KMR_ERR_CAPACITY, "device k-mer table is full; raise kmr_config.max_table_entries / estimated_raw_kmers");
	if (e & ERR_SEGMENT_OVERFLOW) return fail(h, KMR_ERR_CAPACITY, "an owner segment overflowed seg_capacity");
	if (e & ERR_POOL_FULL) return fail(h, KMR_ERR_CAPACITY, "record pool overflow (internal sizing error)");
	if (e & ERR_ENTRIES_FULL) return fail(h, KMR_ERR_CAPACITY, "entry buffer overflow (internal sizing error)");
	return 0;
}

template <int W, bool EXT> int grow_table_t(kmr_handle *h, uint32_t newlog) {
	void *ns; ExtSlot *ne;
	int rc = alloc_table(h, newlog, &ns, &ne);
	if (rc) return rc;
	Table<W> src = table_of<W>(h), dst; dst.slots = (Slot<W> *)ns; dst.ext = ne; dst.log2cap = newlog;
	hipLaunchKernelGGL((rehash_kernel<W, EXT>), dim3(grid_for(1ull << h->log2cap)), dim3(256), 0, h->stream, src, dst, h->hkb, h->derr);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipStreamSynchronize(h->stream));
	hipFree(h->slots); if (h->extslots) hipFree(h->extslots);
	h->slots = ns; h->extslots = ne; h->log2cap = newlog;
	return 0;
}
int grow_table(kmr_handle *h, uint32_t newlog) {
#define GROW(Wv) (h->ext ? grow_table_t<Wv, true>(h, newlog) : grow_table_t<Wv, false>(h, newlog))
	switch (h->W) { case 1: return GROW(1); case 2: return GROW(2); case 3: return GROW(3); default: return GROW(4); }
#undef GROW
}

/* make room for up to 'incoming' new keys: keep the load factor below 0.85 even if all are new */
int ensure_capacity(kmr_handle *h, uint64_t incoming) {
	const uint64_t cap = 1ull << h->log2cap;
	if ((double)(h->occupied + h->pending_kmers + incoming) <= 0.85 * (double)cap) { h->pending_kmers += incoming; return 0; }
	int rc = sync_state(h);          /* learn the true occupancy */
	if (rc) return rc;
	if ((double)(h->occupied + incoming) > 0.85 * (double)cap) {
		uint32_t nl = h->log2cap;
		while ((double)(h->occupied + incoming) > 0.6 * (double)(1ull << nl)) nl++;
		rc = grow_table(h, nl);
		if (rc) return rc;
	}
	h->pending_kmers = incoming;
	return 0;
}

void time_begin(kmr_handle *h, int which, hipEvent_t *a, hipEvent_t *b) {
	hipEventCreate(a); hipEventCreate(b);
	hipEventRecord(*a, h->stream);
	(void)which;
}
void time_end(kmr_handle *h, int which, hipEvent_t a, hipEvent_t b) {
	hipEventRecord(b, h->stream);
	h->pending_events[which].push_back(std::make_pair(a, b));
}

const size_t EXTRACT_SMEM = (size_t)WAVES_PER_BLOCK * 2 * TILE_BUF;

int exclusive_scan(kmr_handle *h, const uint32_t *in, uint64_t n, uint64_t *out);

/* If the batch holds reads longer than one LDS tile, cut them into work units (see ReadsView) and point rv at them. */
int prepare_units(kmr_handle *h, ReadsView &rv, uint32_t span = (uint32_t)TILE_SPAN) {
	rv.u_start = rv.u_end = rv.u_read = nullptr; rv.n_units = 0;
	const uint64_t n = rv.n_reads;
	if (n == 0) return 0;
	if (!h->umax) HIPCHK(h, dev_malloc((void **)&h->umax, 4));
	HIPCHK(h, hipMemsetAsync(h->umax, 0, 4, h->stream));
	if (!h->ucnt || h->ucnt_n < n + 1) { if (h->ucnt) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree(h->ucnt); } h->ucnt = nullptr; HIPCHK(h, dev_malloc((void **)&h->ucnt, 4 * (n + 1))); h->ucnt_n = n + 1; }
	hipLaunchKernelGGL(unit_count_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, rv.offsets, n, h->k, span, h->ucnt, h->umax);
	HIPCHK(h, hipGetLastError());
	unsigned int mx = 0;
	HIPCHK(h, hipMemcpyAsync(&mx, h->umax, 4, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	if (mx <= span) return 0;                   /* the usual case: every read is one unit */
	if (!h->ufirst || h->ufirst_n < n + 1) { if (h->ufirst) hipFree(h->ufirst); h->ufirst = nullptr; HIPCHK(h, dev_malloc((void **)&h->ufirst, 8 * (n + 1))); h->ufirst_n = n + 1; }
	int rc = exclusive_scan(h, h->ucnt, n, h->ufirst); if (rc) return rc;
	uint64_t U = 0;
	HIPCHK(h, hipMemcpy(&U, h->ufirst + n, 8, hipMemcpyDeviceToHost));
	if (h->units_n < U) {
		if (h->u_start) { hipFree(h->u_start); hipFree(h->u_end); hipFree(h->u_read); }
		h->u_start = h->u_end = h->u_read = nullptr; h->units_n = 0;
		HIPCHK(h, dev_malloc((void **)&h->u_start, 8 * U)); HIPCHK(h, dev_malloc((void **)&h->u_end, 8 * U)); HIPCHK(h, dev_malloc((void **)&h->u_read, 8 * U));
		h->units_n = U;
	}
	hipLaunchKernelGGL(unit_fill_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, rv.offsets, n, h->k, span, h->ufirst, h->u_start, h->u_end, h->u_read);
	HIPCHK(h, hipGetLastError());
	rv.u_start = h->u_start; rv.u_end = h->u_end; rv.u_read = h->u_read; rv.n_units = U;
	return 0;
}

template <int W, bool EXT, class Op> int launch_extract(kmr_handle *h, const ReadsView &rv, const Op &op, uint64_t max_blocks = 0) {
	const DevParams dp = dev_params(h);
	const bool sub = Op::NEEDS_WEIGHT && (dp.sub_wnb | dp.sub_snb) != 0;      /* lookups ignore the subtracting reference */
	auto kern = sub ? extract_kernel<W, EXT, Op, true> : extract_kernel<W, EXT, Op, false>;
	HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EXTRACT_SMEM));
	if (dbg()) { int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, WAVES_PER_BLOCK * 64, EXTRACT_SMEM); fprintf(stderr, "extract: %d blocks of %d waves per CU (dynamic LDS %zu)\n", nb, WAVES_PER_BLOCK, EXTRACT_SMEM);
		for (size_t tr : {(size_t)79872, (size_t)65536, (size_t)52000, (size_t)38000}) { hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, WAVES_PER_BLOCK * 64, tr); fprintf(stderr, "   with %zu bytes: %d blocks\n", tr, nb); } }
	const uint64_t tiles = ((rv.u_start ? rv.n_units : rv.n_reads) + 63) / 64;
	uint64_t blocks = (tiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
	if (blocks == 0) return 0;
	if (max_blocks && blocks > max_blocks) blocks = max_blocks;      /* wavefronts then walk several tiles */
	if (blocks > 0x7fffffffull) return fail(h, KMR_ERR_INVALID_ARG, "too many reads in one batch");
	hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES_PER_BLOCK * 64), EXTRACT_SMEM, h->stream, rv, dp, op);
	HIPCHK(h, hipGetLastError());
	return 0;
}

template <int W, bool EXT> int add_reads_dev_t(kmr_handle *h, const ReadsView &rvAll, uint64_t total_bases) {
	/* chunks of reads bounded so that a chunk cannot add more than ~2^27 keys between capacity checks */
	const uint64_t n = rvAll.n_reads;
	const uint64_t avg = n ? std::max<uint64_t>(1, total_bases / n) : 1;
	uint64_t chunk = std::max<uint64_t>(64, ((1ull << 27) / avg) & ~63ull);
	std::vector<uint64_t> off2(2);
	for (uint64_t r = 0; r < n; r += chunk) {
		const uint64_t m = std::min(chunk, n - r);
		/* bases in this chunk: read the two boundary offsets */
		HIPCHK(h, hipMemcpyAsync(&off2[0], rvAll.offsets + r, 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(h, hipMemcpyAsync(&off2[1], rvAll.offsets + r + m, 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(h, hipStreamSynchronize(h->stream));
		const uint64_t bases = off2[1] - off2[0];
		int rc = ensure_capacity(h, bases);     /* #k-mers <= #bases */
		if (rc) return rc;
		ReadsView rv = rvAll;
		rv.offsets = rvAll.offsets + r; rv.n_reads = m;
		rv.discarded = rvAll.discarded ? rvAll.discarded + r : nullptr;
		rv.first_read_idx = rvAll.first_read_idx + r;
		rc = prepare_units(h, rv); if (rc) return rc;
		InsertOp<W, EXT> op; op.table = table_of<W>(h);
		hipEvent_t a, b; time_begin(h, 0, &a, &b);
		rc = launch_extract<W, EXT>(h, rv, op);
		time_end(h, 0, a, b);
		if (rc) return rc;
	}
	return 0;
}

int add_reads_dev_any(kmr_handle *h, const ReadsView &rv, uint64_t total_bases) {
#define ADD(Wv) (h->ext ? add_reads_dev_t<Wv, true>(h, rv, total_bases) : add_reads_dev_t<Wv, false>(h, rv, total_bases))
	switch (h->W) { case 1: return ADD(1); case 2: return ADD(2); case 3: return ADD(3); default: return ADD(4); }
#undef ADD
}

int exclusive_scan(kmr_handle *h, const uint32_t *in, uint64_t n, uint64_t *out /* n+1 */) {
	const uint64_t nblocks = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
	unsigned long long *sums, *total;
	if (h->scan_sums_n < nblocks + 1) {
		if (h->scan_sums) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree(h->scan_sums); h->scan_sums = nullptr; h->scan_sums_n = 0; }
		const uint64_t want = std::max<uint64_t>(nblocks + 1, 4096);
		HIPCHK(h, dev_malloc((void **)&h->scan_sums, sizeof(unsigned long long) * want)); h->scan_sums_n = want;
	}
	sums = h->scan_sums;
	total = sums + nblocks;
	hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nblocks), dim3(256), 0, h->stream, in, n, sums);
	hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, h->stream, sums, nblocks, total);
	hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nblocks), dim3(256), 0, h->stream, in, n, sums, out);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipStreamSynchronize(h->stream));
	return 0;
}

/* bump allocation out of the finalize arena (256-byte aligned); what does not fit is allocated on the side, freed by
 * arena_reset() and added to the size the arena gets next time */
int arena_alloc(kmr_handle *h, void **out, size_t bytes) {
	const size_t need = (bytes + 255) & ~(size_t)255;
	if (h->arena && h->arena_used + need <= h->arena_cap) { *out = h->arena + h->arena_used; h->arena_used += need; h->arena_want += need; return 0; }
	h->arena_want += need;
	HIPCHK(h, dev_malloc(out, std::max<size_t>(need, 256)));
	h->arena_overflow.push_back(*out);
	return 0;
}
/* start of a finalize: everything handed out before is dead (the stream is idle) */
int arena_reset(kmr_handle *h) {
	HIPCHK(h, hipStreamSynchronize(h->stream));
	for (void *p : h->arena_overflow) hipFree(p);
	h->arena_overflow.clear();
	if (h->arena_want > h->arena_cap) {
		if (h->arena) hipFree(h->arena);
		h->arena = nullptr; h->arena_cap = 0;
		const size_t want = h->arena_want + h->arena_want / 8 + (1 << 20);
		if (dev_malloc((void **)&h->arena, want) == hipSuccess) h->arena_cap = want; else { h->arena = nullptr; (void)hipGetLastError(); }
	}
	h->arena_used = 0; h->arena_want = 0;
	return 0;
}
template <class T> int arena_get(kmr_handle *h, T **out, size_t count) { return arena_alloc(h, (void **)out, count * sizeof(T)); }

void free_map(DevMap &m) {
	if (m.start) hipFree(m.start); if (m.keys) hipFree(m.keys); if (m.vals) hipFree(m.vals);
	if (m.sweight) hipFree(m.sweight); if (m.spkt) hipFree(m.spkt); if (m.image) hipFree(m.image);
	m = DevMap();
}

/* empty the map but keep its buffers */
void clear_map(DevMap &m) {
	if (m.image) hipFree(m.image);
	m.image = nullptr; m.image_bytes = 0; m.n = 0; m.present = false;
}
int reserve_bytes(kmr_handle *h, void **ptr, size_t &cap, size_t need) {
	need = std::max<size_t>(need, 8);
	if (*ptr && cap >= need) return 0;
	if (*ptr) hipFree(*ptr);
	*ptr = nullptr; cap = 0;
	HIPCHK(h, dev_malloc(ptr, need));
	cap = need;
	return 0;
}

template <int W> MapView<W> view_of(const DevMap &m, uint32_t vw) {
	MapView<W> v; v.start = m.start; v.keys = m.keys; v.vals = m.vals; v.sweight = m.sweight; v.nb = m.present ? m.nb : 0; v.vw = vw;
	return v;
}

template <int W, bool EXT> int finalize_t(kmr_handle *h, uint32_t min_depth) {
	int rc = sync_state(h);
	if (rc) return rc;
	hipEvent_t ea, eb; time_begin(h, 1, &ea, &eb);
	FinalizeParams f; f.kb = h->hkb; f.min_depth = min_depth; f.has_singletons = h->cfg.separate_singletons ? 1 : 0; f.nb_weak = h->nb_weak; f.nb_sing = h->nb_sing; f.uni_wbits = 0;
	const bool keepSing = f.has_singletons && min_depth <= 1;
	uint32_t *wc = nullptr, *sc = nullptr; FinalizeCounters *fc = nullptr;
	HIPCHK(h, dev_malloc((void **)&wc, 4 * h->nb_weak)); HIPCHK(h, dev_malloc((void **)&sc, 4 * h->nb_sing)); HIPCHK(h, dev_malloc((void **)&fc, sizeof(FinalizeCounters)));
	HIPCHK(h, hipMemsetAsync(wc, 0, 4 * h->nb_weak, h->stream)); HIPCHK(h, hipMemsetAsync(sc, 0, 4 * h->nb_sing, h->stream)); HIPCHK(h, hipMemsetAsync(fc, 0, sizeof(FinalizeCounters), h->stream));
	Table<W> t = table_of<W>(h);
	const int g = grid_for(1ull << h->log2cap);
	hipLaunchKernelGGL(classify_kernel<W>, dim3(g), dim3(256), 0, h->stream, t, f, wc, sc, fc);
	HIPCHK(h, hipGetLastError());
	FinalizeCounters c;
	HIPCHK(h, hipMemcpyAsync(&c, fc, sizeof(c), hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	h->stats.unique_kmers = c.unique;
	/* singletonKmers only moves inside the hasSingletons branches of append() (src/KmerSpectrum.h:1625,1649) */
	h->stats.singleton_kmers = f.has_singletons ? c.singletons : 0;
	DevMap &wm = h->weak, &sm = h->sing;
	free_map(wm); free_map(sm);
	wm.nb = h->nb_weak; wm.n = c.weak_kept; wm.present = true;
	sm.nb = h->nb_sing; sm.n = c.sing_kept; sm.present = keepSing;
	const uint32_t vw = EXT ? 15 : 3;
	HIPCHK(h, dev_malloc((void **)&wm.start, 8 * (wm.nb + 1))); HIPCHK(h, dev_malloc((void **)&sm.start, 8 * (sm.nb + 1)));
	rc = exclusive_scan(h, wc, wm.nb, wm.start); if (rc) return rc;
	rc = exclusive_scan(h, sc, sm.nb, sm.start); if (rc) return rc;
	HIPCHK(h, dev_malloc((void **)&wm.keys, std::max<uint64_t>(8, 8ull * W * wm.n))); HIPCHK(h, dev_malloc((void **)&wm.vals, std::max<uint64_t>(8, 4ull * vw * wm.n)));
	HIPCHK(h, dev_malloc((void **)&sm.keys, std::max<uint64_t>(8, 8ull * W * sm.n))); HIPCHK(h, dev_malloc((void **)&sm.sweight, std::max<uint64_t>(8, sm.n)));
	if (EXT) HIPCHK(h, dev_malloc((void **)&sm.spkt, std::max<uint64_t>(8, 4ull * sm.n)));
	HIPCHK(h, hipMemsetAsync(wc, 0, 4 * h->nb_weak, h->stream)); HIPCHK(h, hipMemsetAsync(sc, 0, 4 * h->nb_sing, h->stream));
	hipLaunchKernelGGL((scatter_kernel<W, EXT>), dim3(g), dim3(256), 0, h->stream, t, f, wm.start, wc, wm.keys, wm.vals, sm.start, sc, sm.keys, sm.sweight, sm.spkt);
	HIPCHK(h, hipGetLastError());
	SortView<W> sv; sv.keys = wm.keys; sv.vals = wm.vals; sv.b8 = nullptr; sv.pkt = nullptr; sv.vw = vw;
	hipLaunchKernelGGL((sort_buckets_kernel<W, EXT ? 15 : 3>), dim3(grid_for(wm.nb, 4, 1 << 20)), dim3(256), 0, h->stream, sv, wm.start, wm.nb);
	if (sm.n) {
		SortView<W> ss; ss.keys = sm.keys; ss.vals = nullptr; ss.b8 = sm.sweight; ss.pkt = sm.spkt; ss.vw = 0;
		hipLaunchKernelGGL((sort_buckets_kernel<W, 0>), dim3(grid_for(sm.nb, 4, 1 << 20)), dim3(256), 0, h->stream, ss, sm.start, sm.nb);
	}
	HIPCHK(h, hipGetLastError());
	time_end(h, 1, ea, eb);
	HIPCHK(h, hipStreamSynchronize(h->stream));
	hipFree(wc); hipFree(sc); hipFree(fc);
	/* the table allocation is kept for kmr_reset(); kmr_release_table() frees it */
	h->has_singletons = keepSing;
	if (!keepSing) { sm.n = 0; }
	h->stats.weak_entries = wm.n; h->stats.singleton_entries = keepSing ? sm.n : 0;
	h->finalized = true; h->map_gen++;
	return sync_state(h);
}

template <int W> int build_image_t(kmr_handle *h, DevMap &m, bool weakMap) {
	if (m.image) return 0;
	const uint32_t vw = h->ext ? 15 : 3;
	const uint32_t vbytes = weakMap ? (h->ext ? 60 : 12) : (h->ext ? 5 : 1);
	m.image_bytes = 8 * (2 + m.nb) + 4 * m.nb + m.n * (h->kb + vbytes);
	HIPCHK(h, dev_malloc((void **)&m.image, m.image_bytes));
	hipLaunchKernelGGL(image_header_kernel, dim3(grid_for(m.nb)), dim3(256), 0, h->stream, m.image, m.start, m.nb, h->kb, vbytes);
	if (m.n) hipLaunchKernelGGL(image_entries_kernel<W>, dim3(grid_for(m.n)), dim3(256), 0, h->stream, m.image, m.start, m.nb, h->kb, vbytes,
	                           m.keys, weakMap ? m.vals : nullptr, vw, m.sweight, m.spkt, m.n);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipStreamSynchronize(h->stream));
	return 0;
}
int build_image(kmr_handle *h, DevMap &m, bool weakMap) {
	switch (h->W) { case 1: return build_image_t<1>(h, m, weakMap); case 2: return build_image_t<2>(h, m, weakMap);
	case 3: return build_image_t<3>(h, m, weakMap); default: return build_image_t<4>(h, m, weakMap); }
}

template <int W> int load_image_t(kmr_handle *h, DevMap &m, bool weakMap, const uint8_t *src, uint64_t len) {
	if (len < 16) return fail(h, KMR_ERR_INVALID_ARG, "image too short");
	uint64_t nb, mask; memcpy(&nb, src, 8); memcpy(&mask, src + 8, 8);
	if (nb == 0 || (nb & (nb - 1)) || mask != nb - 1 || len < 8 * (2 + nb) + 4 * nb) return fail(h, KMR_ERR_INVALID_ARG, "bad image header");
	const uint32_t vw = h->ext ? 15 : 3;
	const uint32_t vbytes = weakMap ? (h->ext ? 60 : 12) : (h->ext ? 5 : 1);
	if ((len - 8 * (2 + nb) - 4 * nb) % (h->kb + vbytes) != 0) return fail(h, KMR_ERR_INVALID_ARG, "image size does not match k / value type");
	free_map(m);
	m.nb = nb; m.n = (len - 8 * (2 + nb) - 4 * nb) / (h->kb + vbytes); m.present = true;
	/* validate offsets on the host before any kernel dereferences them */
	const uint64_t *offs = (const uint64_t *)(src + 16);
	uint64_t expect = 8 * (2 + nb);
	for (uint64_t b = 0; b < nb; b++) {
		if (offs[b] != expect || expect + 4 > len) return fail(h, KMR_ERR_INVALID_ARG, "image offsets are not the packed store() layout");
		uint32_t cnt; memcpy(&cnt, src + expect, 4);
		expect += 4 + (uint64_t)cnt * (h->kb + vbytes);
		if (expect > len) return fail(h, KMR_ERR_INVALID_ARG, "image bucket runs past the end");
	}
	if (expect != len) return fail(h, KMR_ERR_INVALID_ARG, "image length mismatch");
	HIPCHK(h, dev_malloc((void **)&m.image, len)); m.image_bytes = len;
	HIPCHK(h, hipMemcpy(m.image, src, len, hipMemcpyHostToDevice));
	uint32_t *counts; HIPCHK(h, dev_malloc((void **)&counts, 4 * nb));
	hipLaunchKernelGGL(image_counts_kernel, dim3(grid_for(nb)), dim3(256), 0, h->stream, m.image, nb, counts);
	HIPCHK(h, dev_malloc((void **)&m.start, 8 * (nb + 1)));
	int rc = exclusive_scan(h, counts, nb, m.start); hipFree(counts); if (rc) return rc;
	HIPCHK(h, dev_malloc((void **)&m.keys, std::max<uint64_t>(8, 8ull * W * m.n)));
	if (weakMap) HIPCHK(h, dev_malloc((void **)&m.vals, std::max<uint64_t>(8, 4ull * vw * m.n)));
	else { HIPCHK(h, dev_malloc((void **)&m.sweight, std::max<uint64_t>(8, m.n))); if (h->ext) HIPCHK(h, dev_malloc((void **)&m.spkt, std::max<uint64_t>(8, 4 * m.n))); }
	if (m.n) hipLaunchKernelGGL(image_unpack_kernel<W>, dim3(grid_for(m.n)), dim3(256), 0, h->stream, m.image, m.start, nb, h->kb, vbytes, m.keys, m.vals, vw, m.sweight, m.spkt, m.n);
	/* restore() accepts unsorted buckets (setLastSorted); lookups here need them sorted */
	SortView<W> sv; sv.keys = m.keys; sv.vals = m.vals; sv.b8 = m.sweight; sv.pkt = m.spkt; sv.vw = weakMap ? vw : 0;
	if (!weakMap) hipLaunchKernelGGL((sort_buckets_kernel<W, 0>), dim3(grid_for(nb, 4, 1 << 20)), dim3(256), 0, h->stream, sv, m.start, nb);
	else if (h->ext) hipLaunchKernelGGL((sort_buckets_kernel<W, 15>), dim3(grid_for(nb, 4, 1 << 20)), dim3(256), 0, h->stream, sv, m.start, nb);
	else hipLaunchKernelGGL((sort_buckets_kernel<W, 3>), dim3(grid_for(nb, 4, 1 << 20)), dim3(256), 0, h->stream, sv, m.start, nb);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipStreamSynchronize(h->stream));
	hipFree(m.image); m.image = nullptr; m.image_bytes = 0;    /* rebuilt (sorted) on demand */
	return 0;
}

template <int W, int VW> void launch_sort(kmr_handle *h, DevMap &m, bool weakMap) {
	SortView<W> sv; sv.keys = m.keys; sv.vals = weakMap ? m.vals : nullptr; sv.b8 = m.sweight; sv.pkt = m.spkt; sv.vw = weakMap ? VW : 0;
	hipLaunchKernelGGL((sort_buckets_kernel<W, VW>), dim3(grid_for(m.nb, 4, 1 << 20)), dim3(256), 0, h->stream, sv, m.start, m.nb);
}
/* union of the handle's map with a stored map of the same shape (see merge_copy_kernel) */
template <int W> int merge_image_t(kmr_handle *h, DevMap &m, bool weakMap, const uint8_t *src, uint64_t len) {
	DevMap t;
	int rc = load_image_t<W>(h, t, weakMap, src, len);
	if (rc) { free_map(t); return rc; }
	if (t.nb != m.nb) { free_map(t); return fail(h, KMR_ERR_INVALID_ARG, "Can not merge two maps of differing sizes (src/Kmer.h:3210)"); }
	const uint32_t vw = h->ext ? 15 : 3;
	DevMap d; d.nb = m.nb; d.n = m.n + t.n; d.present = true;
	uint32_t *counts = nullptr, *dup = nullptr;
	auto bail = [&](int code) { free_map(t); free_map(d); if (counts) hipFree(counts); if (dup) hipFree(dup); return code; };
#define MCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hip_err_text(e_); return bail(e_ == hipErrorOutOfMemory ? KMR_ERR_OOM : KMR_ERR_HIP); } } while (0)
	MCHK(dev_malloc((void **)&counts, 4 * m.nb)); MCHK(dev_malloc((void **)&dup, 4)); MCHK(hipMemsetAsync(dup, 0, 4, h->stream));
	MCHK(dev_malloc((void **)&d.start, 8 * (m.nb + 1)));
	hipLaunchKernelGGL(merge_counts_kernel, dim3(grid_for(m.nb)), dim3(256), 0, h->stream, m.start, t.start, m.nb, counts);
	rc = exclusive_scan(h, counts, m.nb, d.start); if (rc) return bail(rc);
	MCHK(dev_malloc((void **)&d.keys, std::max<uint64_t>(8, 8ull * W * d.n)));
	if (weakMap) MCHK(dev_malloc((void **)&d.vals, std::max<uint64_t>(8, 4ull * vw * d.n)));
	else { MCHK(dev_malloc((void **)&d.sweight, std::max<uint64_t>(8, d.n))); if (h->ext) MCHK(dev_malloc((void **)&d.spkt, std::max<uint64_t>(8, 4 * d.n))); }
	if (m.n) hipLaunchKernelGGL(merge_copy_kernel<W>, dim3(grid_for(m.n)), dim3(256), 0, h->stream, m.start, t.start, false, m.nb, m.n, m.keys, weakMap ? m.vals : nullptr, vw, m.sweight, m.spkt, d.start, d.keys, d.vals, d.sweight, d.spkt);
	if (t.n) hipLaunchKernelGGL(merge_copy_kernel<W>, dim3(grid_for(t.n)), dim3(256), 0, h->stream, t.start, m.start, true, m.nb, t.n, t.keys, weakMap ? t.vals : nullptr, vw, t.sweight, t.spkt, d.start, d.keys, d.vals, d.sweight, d.spkt);
	if (!weakMap) launch_sort<W, 0>(h, d, false); else if (h->ext) launch_sort<W, 15>(h, d, true); else launch_sort<W, 3>(h, d, true);
	hipLaunchKernelGGL(duplicate_keys_kernel<W>, dim3(grid_for(d.nb)), dim3(256), 0, h->stream, d.start, d.nb, d.keys, dup);
	MCHK(hipGetLastError());
	uint32_t hdup = 0;
	MCHK(hipMemcpyAsync(&hdup, dup, 4, hipMemcpyDeviceToHost, h->stream)); MCHK(hipStreamSynchronize(h->stream));
	if (hdup && !weakMap) {
#undef MCHK
		/* (a k-mer in both singleton maps would have to be promoted into the weak map: KmerMap::mergePromote, which the reference
		 * itself refuses -- "This method is broken", src/Kmer.h:2675-2677) */
		bail(0); return fail(h, KMR_ERR_UNSUPPORTED, "the two singleton maps share k-mers: merging them means promoting those into the weak map (mergePromote, which the reference refuses too)");
	}
	if (hdup) {      /* mergeAdd proper: the k-mers both maps hold add their values */
#define MCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hip_err_text(e_); free_map(d2); return bail(e_ == hipErrorOutOfMemory ? KMR_ERR_OOM : KMR_ERR_HIP); } } while (0)
		DevMap d2; d2.nb = d.nb; d2.present = true;
		hipLaunchKernelGGL(merge_distinct_kernel<W>, dim3(grid_for(d.nb)), dim3(256), 0, h->stream, d.start, d.nb, d.keys, counts);
		MCHK(dev_malloc((void **)&d2.start, 8 * (d.nb + 1)));
		rc = exclusive_scan(h, counts, d.nb, d2.start); if (rc) { free_map(d2); return bail(rc); }
		MCHK(hipMemcpyAsync(&d2.n, d2.start + d.nb, 8, hipMemcpyDeviceToHost, h->stream)); MCHK(hipStreamSynchronize(h->stream));
		MCHK(dev_malloc((void **)&d2.keys, std::max<uint64_t>(8, 8ull * W * d2.n)));
		MCHK(dev_malloc((void **)&d2.vals, std::max<uint64_t>(8, 4ull * vw * d2.n)));
		hipLaunchKernelGGL(merge_add_kernel<W>, dim3(grid_for(d.nb)), dim3(256), 0, h->stream, d.start, d.nb, d.keys, d.vals, vw, d2.start, d2.keys, d2.vals);
		MCHK(hipGetLastError()); MCHK(hipStreamSynchronize(h->stream));
#undef MCHK
		free_map(d);
		d = d2;
	}
	hipFree(counts); hipFree(dup); free_map(t);
	free_map(m);
	m = d;
	return 0;
}

template <int W> int lookup_t(kmr_handle *h, const uint8_t *packed, uint64_t n, uint32_t *counts) {
	uint8_t *dk; uint32_t *dc;
	HIPCHK(h, dev_malloc((void **)&dk, std::max<uint64_t>(8, n * h->kb))); HIPCHK(h, dev_malloc((void **)&dc, std::max<uint64_t>(8, 4 * n)));
	HIPCHK(h, hipMemcpyAsync(dk, packed, n * h->kb, hipMemcpyHostToDevice, h->stream));
	const uint32_t vw = h->ext ? 15 : 3;
	hipLaunchKernelGGL(lookup_keys_kernel<W>, dim3(grid_for(n)), dim3(256), 0, h->stream, view_of<W>(h->weak, vw), view_of<W>(h->sing, vw), dk, n, h->hkb, dc);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipMemcpyAsync(counts, dc, 4 * n, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	hipFree(dk); hipFree(dc);
	return 0;
}

/* the lookup accelerator of the current weak map (built on first use after the map changed); slots == nullptr when it cannot
 * be had (no memory): the callers then search the buckets */
template <int W> LutView<W> lut_of(kmr_handle *h) {
	LutView<W> v; v.slots = nullptr; v.mask = 0; v.shift = 0;
	if (h->tune.no_lut || !h->weak.present || h->weak.n == 0) return v;
	if (!(h->lut && h->lut_gen == h->map_gen)) {
		uint32_t l2 = 10; while ((1ull << l2) < 2 * h->weak.n) l2++;
		const size_t bytes = (size_t)(W + 1) * 8 << l2;
		if (h->lut_bytes < bytes) {
			if (h->lut) { hipStreamSynchronize(h->stream); hipFree(h->lut); h->lut = nullptr; h->lut_bytes = 0; }
			size_t fr = 0, tot = 0;
			if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < bytes + (1ull << 30) || dev_malloc((void **)&h->lut, bytes) != hipSuccess) { h->lut = nullptr; (void)hipGetLastError(); return v; }
			h->lut_bytes = bytes;
		}
		const uint32_t vw = h->ext ? 15 : 3;
		hipLaunchKernelGGL(lut_clear_kernel, dim3(4096), dim3(256), 0, h->stream, h->lut, 1ull << l2, (uint32_t)(W + 1));
		hipLaunchKernelGGL(lut_build_kernel<W>, dim3(grid_for(h->weak.n)), dim3(256), 0, h->stream, view_of<W>(h->weak, vw), h->weak.n, h->lut, (1ull << l2) - 1, 64 - l2, h->hkb);
		if (hipGetLastError() != hipSuccess) return v;
		h->lut_log2 = l2; h->lut_gen = h->map_gen;
	}
	v.slots = h->lut; v.mask = (1ull << h->lut_log2) - 1; v.shift = 64 - h->lut_log2;
	return v;
}

template <int W> int lookup_reads_t(kmr_handle *h, const ReadsView &rv, uint32_t *dout, const uint64_t *dout_off, bool weak_only = false) {
	LookupOp<W> op; const uint32_t vw = h->ext ? 15 : 3; op.weak_only = weak_only;
	op.lut = weak_only ? lut_of<W>(h) : LutView<W>{nullptr, 0, 0};
	op.weak = view_of<W>(h->weak, vw); op.sing = view_of<W>(h->sing, vw); op.out = dout; op.out_offsets = dout_off; op.first_read_idx = rv.first_read_idx;
	return launch_extract<W, false>(h, rv, op);
}

/* stage host read arrays on the device (padded so 16-byte tile loads stay inside the allocation) */
struct StagedReads { uint8_t *b = nullptr, *q = nullptr, *d = nullptr; uint64_t *o = nullptr; void release() { if (b) hipFree(b); if (q) hipFree(q); if (d) hipFree(d); if (o) hipFree(o); b = q = d = nullptr; o = nullptr; } };
int stage_reads(kmr_handle *h, const char *bases, const char *quals, const uint64_t *offsets, uint64_t n, const uint8_t *disc, StagedReads &s, uint64_t &total) {
	total = n ? offsets[n] - offsets[0] : 0;
	HIPCHK(h, dev_malloc((void **)&s.b, total + 64)); HIPCHK(h, dev_malloc((void **)&s.o, 8 * (n + 1)));
	HIPCHK(h, hipMemcpyAsync(s.b, bases + (n ? offsets[0] : 0), total, hipMemcpyHostToDevice, h->stream));
	std::vector<uint64_t> rel(n + 1);
	for (uint64_t i = 0; i <= n; i++) rel[i] = n ? offsets[i] - offsets[0] : 0;
	HIPCHK(h, hipMemcpyAsync(s.o, rel.data(), 8 * (n + 1), hipMemcpyHostToDevice, h->stream));
	if (quals) { HIPCHK(h, dev_malloc((void **)&s.q, total + 64)); HIPCHK(h, hipMemcpyAsync(s.q, quals + (n ? offsets[0] : 0), total, hipMemcpyHostToDevice, h->stream)); }
	if (disc) { HIPCHK(h, dev_malloc((void **)&s.d, n + 8)); HIPCHK(h, hipMemcpyAsync(s.d, disc, n, hipMemcpyHostToDevice, h->stream)); }
	HIPCHK(h, hipStreamSynchronize(h->stream));   /* rel[] goes out of scope */
	return 0;
}


/* ---------------------------------------------------------------------- */
/* streaming build path (kmr_partition.hpp)                                  */
#define TARGET_LIST_RECORDS (h->tune.target_list)      /* records per final list the partition bits aim for (kmr_tune "target_list_records") */
const double MAX_LIST_DISTINCT = 600.0;       /* distinct keys per final list the 1024-slot table takes comfortably (limit 819) */
const uint64_t L2_ITEM_CHUNKS = 16384;       /* level-2 work item = up to 1M records of one level-1 list */
const uint64_t SUB_BATCH_BASES = 1ull << 30;      /* linear records of one sub-batch: <= 17 GB at 16 bytes; 2^28 cost 2 ms per C2 step in launch tails */

size_t rec_bytes(kmr_handle *h) { return h->superkmer_mode ? 16 : 8 * h->W + (h->ext ? 16 : 8); }      /* Record<W> / RecordX<W>; a 16-byte granule of a super-k-mer record */
/* partition kernel shape: one 1024-thread block per compute unit, 8 records per thread per batch, a
 * 4-record write-combining line per list in LDS (see partition_direct_kernel) */
/* (PD_THREADS, PD_RPT, PD_LINE and COUNT_LOG2S: kmr_instances.hpp) */
/* partition bits per level that keep the per-list book-keeping and lines inside the 160 KB of LDS */
int max_part_bits(kmr_handle *h) { return rec_bytes(h) <= 24 ? 10 : 9; }
PoolView pool_view(kmr_handle *h, HostPool &p) { PoolView v; v.base = p.base; v.chunk_list = p.chunk_list; v.chunk_count = p.chunk_count; v.head = p.head; v.cap = p.cap; v.err = h->derr; return v; }
/* log2 of the weak map's bucket count: the partition is cut along the bucket index (part_order) */
uint32_t part_rot(kmr_handle *h) { uint32_t r = 0; while ((1ull << (r + 1)) <= h->nb_weak) r++; return r; }
template <int W, bool EXT, int LEVEL> int launch_partition(kmr_handle *h, const PartSource<W> &S, HostPool &pool, int grid, int bits, int shift) {
	auto kern = partition_direct_kernel<W, EXT, LEVEL, PD_THREADS, PD_RPT, PD_LINE>;
	const size_t smem = partition_direct_smem_bytes<W, EXT, PD_THREADS, PD_RPT, PD_LINE>(bits);
	if (dbg()) fprintf(stderr, "partition level %d W=%d bits=%d shift=%d smem=%zu grid=%d\n", LEVEL, W, bits, shift, smem, grid);
	HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
	                              (int)partition_direct_smem_bytes<W, EXT, PD_THREADS, PD_RPT, PD_LINE>(max_part_bits(h))));
	hipLaunchKernelGGL(kern, dim3(grid), dim3(PD_THREADS), smem, h->stream, S, pool_view(h, pool), h->work_counter, bits, shift);
	HIPCHK(h, hipGetLastError());
	return 0;
}

void pool_free(HostPool &p) {
	if (p.base) hipFree(p.base); if (p.chunk_list) hipFree(p.chunk_list); if (p.chunk_count) hipFree(p.chunk_count); if (p.head) hipFree(p.head);
	p = HostPool();
}
/* make sure the pool can take 'extra' more chunks (keeps the used prefix when it has to move) */
int pool_reserve(kmr_handle *h, HostPool &p, uint64_t extra, bool keep) {
	unsigned int used = 0;
	if (!keep) p.used_ub = 0;
	if (p.head && keep) {
		/* the host keeps an upper bound of the chunks handed out so far (every launch adds what it reserved); only
		 * when that bound no longer fits is the stream drained and the real allocator head read back */
		if (p.base && p.used_ub + extra + 64 <= p.cap) { p.used_ub += extra; return 0; }
		HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, hipMemcpy(&used, p.head, 4, hipMemcpyDeviceToHost)); if (used > p.cap) used = p.cap;
	}
	p.used_ub = (uint64_t)used + extra;
	const uint64_t need = (uint64_t)used + extra + 64;
	if (need >= 0xffffffffull) return fail(h, KMR_ERR_CAPACITY, "record pool would exceed 2^32 chunks");
	if (!p.head) { HIPCHK(h, dev_malloc((void **)&p.head, 4)); HIPCHK(h, hipMemset(p.head, 0, 4)); }
	if (!keep) HIPCHK(h, hipMemsetAsync(p.head, 0, 4, h->stream));
	if (need <= p.cap) return 0;
	uint64_t ncap = keep && used ? need + need / 4 : need;
	if (p.presize && need + p.presize < 0xffffffffull) ncap = std::max(ncap, need + p.presize);
	p.presize = 0;
	p.chunk_bytes = (size_t)CH * rec_bytes(h);
	uint8_t *nb; uint32_t *nl, *nc;
	HIPCHK(h, dev_malloc((void **)&nb, ncap * p.chunk_bytes));
	HIPCHK(h, dev_malloc((void **)&nl, 4 * ncap)); HIPCHK(h, dev_malloc((void **)&nc, 4 * ncap));
	if (used) {
		HIPCHK(h, hipMemcpy(nb, p.base, (size_t)used * p.chunk_bytes, hipMemcpyDeviceToDevice));
		HIPCHK(h, hipMemcpy(nl, p.chunk_list, 4ull * used, hipMemcpyDeviceToDevice));
		HIPCHK(h, hipMemcpy(nc, p.chunk_count, 4ull * used, hipMemcpyDeviceToDevice));
		/* device-to-device copies may return before they have run: the sources are freed next */
		HIPCHK(h, hipDeviceSynchronize());
	}
	if (p.base) hipFree(p.base); if (p.chunk_list) hipFree(p.chunk_list); if (p.chunk_count) hipFree(p.chunk_count);
	p.base = nb; p.chunk_list = nl; p.chunk_count = nc; p.cap = (uint32_t)ncap;
	return 0;
}

template <class T> int ensure_buf(kmr_handle *h, T *&ptr, uint64_t &cap, uint64_t need, size_t elem) {
	if (need <= cap && ptr) return 0;
	if (ptr) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree((void *)ptr); }   /* kernels in flight may still read it */
	ptr = nullptr; cap = 0;
	const uint64_t n = std::max<uint64_t>(need, 16);
	HIPCHK(h, dev_malloc((void **)&ptr, n * elem));
	cap = n;
	return 0;
}

int zero_work_counter(kmr_handle *h) {
	if (!h->work_counter) HIPCHK(h, dev_malloc((void **)&h->work_counter, 4));
	HIPCHK(h, hipMemsetAsync(h->work_counter, 0, 4, h->stream));
	return 0;
}

int num_cus(kmr_handle *h) {
	if (h->ncu <= 0) {
		hipDeviceProp_t pr; h->ncu = 256;
		if (hipGetDeviceProperties(&pr, h->device) == hipSuccess) h->ncu = pr.multiProcessorCount;
	}
	return h->ncu;
}
int part_grid(kmr_handle *h) { return num_cus(h) * 2; }
/* the partition kernel wants a compute unit to itself: every (block, list) pair is a write stream, and the fewer of
 * those there are the longer the runs each batch appends */
int partition_blocks(kmr_handle *h) { return h->tune.part_blocks > 0 ? h->tune.part_blocks : num_cus(h); }

/* per-block level-1 state, allocated (and emptied) on first use */
template <int W, bool EXT> int ensure_l1_state(kmr_handle *h) {
	const size_t stride = partition_state_bytes<W, EXT, PD_LINE>(h->bits1), need = stride * (size_t)partition_blocks(h);
	if (h->l1_state && h->l1_state_bytes == need) return 0;
	if (h->l1_state) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree(h->l1_state); h->l1_state = nullptr; }
	HIPCHK(h, dev_malloc((void **)&h->l1_state, need)); h->l1_state_bytes = need;
	hipLaunchKernelGGL(partition_state_init_kernel, dim3(partition_blocks(h)), dim3(256), 0, h->stream, h->l1_state, stride, h->bits1, (uint32_t)partition_blocks(h));
	HIPCHK(h, hipGetLastError());
	h->l1_state_dirty = false;
	return 0;
}
/* last level-1 launch of a build: no input, every block flushes what it kept back */
template <int W, bool EXT> int flush_l1_state(kmr_handle *h) {
	if (!h->l1_state || !h->l1_state_dirty) return 0;
	int rc = pool_reserve(h, h->l1, (uint64_t)partition_blocks(h) * ((1ull << h->bits1) + 512) + 64, true); if (rc) return rc;
	rc = zero_work_counter(h); if (rc) return rc;
	PartSource<W> S; memset(&S, 0, sizeof(S));
	S.kb = h->hkb; S.rot = part_rot(h); S.state = h->l1_state; S.state_final = 1;
	rc = launch_partition<W, EXT, 1>(h, S, h->l1, partition_blocks(h), h->bits1, 0);
	h->l1_state_dirty = false;
	return rc;
}

/* level-1 partition of a linear record buffer into h->l1 */
template <int W, bool EXT> int partition_level1(kmr_handle *h, const void *linear, const uint64_t *ext_start, const uint32_t *ext_count,
                                      uint64_t n_ext, uint32_t ext_stride, uint64_t ext_len, uint64_t total, uint64_t max_records,
                                      unsigned long long *valid_counter = nullptr, uint32_t packed_words = 0, uint64_t ordinal_base = 0) {
	if (n_ext == 0) return 0;
	const int grid = (int)std::min<uint64_t>(partition_blocks(h), n_ext);
	if (!h->l1.base) {
		const uint64_t est = std::max<uint64_t>(max_records, h->cfg.estimated_raw_kmers / std::max<uint32_t>(1, h->cfg.world_size));
		const uint64_t launches = est / std::max<uint64_t>(1, max_records) + 2;
		/* level 2 writes into the same pool (it recycles the chunks it reads): room for its partly filled chunks */
		const uint64_t l2_allowance = (est / CH / L2_ITEM_CHUNKS + (1ull << h->bits1) + 1) * (1ull << max_part_bits(h)) + (uint64_t)part_grid(h) * 512;
		{	/* what only kmr_finalize uses (entry buffers) is dead during a build: released if the pool would not fit beside it */
			const uint64_t chunks = est / CH + launches * (uint64_t)grid * ((1ull << h->bits1) + 512) + 64 + l2_allowance;
			size_t mfree = 0, mtotal = 0;
			if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess && (double)mfree < (double)chunks * CH * rec_bytes(h) * 1.02 + (double)(2ull << 30)) {
				HIPCHK(h, hipStreamSynchronize(h->stream));
				if (h->uw_keys) hipFree(h->uw_keys); if (h->uw_vals) hipFree(h->uw_vals); if (h->us_keys) hipFree(h->us_keys); if (h->us_b8) hipFree(h->us_b8); if (h->us_pkt) hipFree(h->us_pkt);
	if (h->ue) hipFree(h->ue); if (h->ue2) hipFree(h->ue2); h->ue = h->ue2 = nullptr; h->ue_cap = h->ue2_cap = 0;
				h->uw_keys = h->uw_vals = h->us_keys = h->us_b8 = h->us_pkt = nullptr; h->uw_cap = h->us_cap = 0;
			}
		}
		uint64_t want = est / CH + launches * (uint64_t)grid * ((1ull << h->bits1) + 512) + 64 + l2_allowance;
		{	/* with room to spare the pool takes a second copy of the records, so that level 2 can append fresh chunks */
			size_t mfree = 0, mtotal = 0;
			const double twice = (double)(want + est / CH) * CH * rec_bytes(h) * 1.02;
			if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess && twice + (double)(8ull << 30) < (double)mfree * 0.5) want += est / CH;
		}
		int rc0 = pool_reserve(h, h->l1, want, true);
		if (rc0) return rc0;
	}
	int rc = pool_reserve(h, h->l1, max_records / CH + (uint64_t)grid * ((1ull << h->bits1) + 512) + 64, true);
	if (rc) return rc;
	rc = zero_work_counter(h); if (rc) return rc;
	PartSource<W> S; memset(&S, 0, sizeof(S));
	S.linear = linear; S.ext_start = ext_start; S.ext_count = ext_count; S.n_ext = n_ext; S.ext_stride = ext_stride; S.ext_len = ext_len; S.total = total;
	S.valid_counter = valid_counter; S.kb = h->hkb; S.rot = part_rot(h); S.packed_words = packed_words; S.ordinal_base = ordinal_base;
	if (!h->tune.no_l1_state) {
		rc = ensure_l1_state<W, EXT>(h); if (rc) return rc;
		S.state = h->l1_state; S.state_final = 0; h->l1_state_dirty = true;
	}
	hipEvent_t ta, tb; time_begin(h, KMR_TIME_PARTITION1, &ta, &tb);
	rc = launch_partition<W, EXT, 1>(h, S, h->l1, grid, h->bits1, 0);
	time_end(h, KMR_TIME_PARTITION1, ta, tb);
	return rc;
}

void choose_bits1(kmr_handle *h, uint64_t records_hint) {
	/* total bits aim at TARGET_LIST_RECORDS per final list; level 2 takes up to max_part_bits(h) of them.  Two of
	 * those are held back: if most k-mers turn out to be distinct the lists have to be up to 4x smaller (see the
	 * distinct probe in finalize_partition_t). */
	uint64_t est = std::max<uint64_t>(records_hint, h->cfg.estimated_raw_kmers / std::max<uint32_t>(1, h->cfg.world_size));
	const int mb = max_part_bits(h);
	int T = 0; while (T < 2 * mb && (est >> T) > TARGET_LIST_RECORDS) T++;
	h->bits1 = std::max(0, std::min(mb, T + 2 - mb));
}

/* sender side of the exchange: reads -> linear records (every owner's) -> owner segments */
template <int W, bool EXT> int extract_by_owner_t(kmr_handle *h, const ReadsView &rvAll, uint64_t total_bases, void *dev_records, uint64_t seg_capacity, void *dev_seg_counts, uint32_t *dev_pos = nullptr) {
	const uint64_t n = rvAll.n_reads;
	const uint64_t avg = n ? std::max<uint64_t>(1, total_bases / n) : 1;
	const uint64_t chunk = std::max<uint64_t>(64, (SUB_BATCH_BASES / avg) & ~63ull);
	for (uint64_t r = 0; r < n; r += chunk) {
		const uint64_t m = std::min(chunk, n - r);
		ReadsView rv = rvAll;
		rv.offsets = rvAll.offsets + r; rv.n_reads = m;
		rv.discarded = rvAll.discarded ? rvAll.discarded + r : nullptr;
		rv.first_read_idx = rvAll.first_read_idx + r;
		int rc = prepare_units(h, rv); if (rc) return rc;
		const uint64_t nu = rv.u_start ? rv.n_units : m;
		rc = ensure_buf(h, h->kcap, h->kcap_n, nu + 1, 4); if (rc) return rc;
		rc = ensure_buf(h, h->koff, h->koff_n, nu + 1, 8); if (rc) return rc;
		hipLaunchKernelGGL(kmer_capacity_kernel, dim3(grid_for(nu)), dim3(256), 0, h->stream, rv, h->k, h->kcap);
		HIPCHK(h, hipGetLastError());
		rc = exclusive_scan(h, h->kcap, nu, h->koff); if (rc) return rc;
		uint64_t total_cap = 0;
		HIPCHK(h, hipMemcpy(&total_cap, h->koff + nu, 8, hipMemcpyDeviceToHost));
		const uint64_t tiles = (nu + 63) / 64;
		rc = ensure_buf(h, h->linear, h->linear_cap, total_cap, sizeof(typename PoolRec<W, EXT>::type)); if (rc) return rc;      /* (rec_bytes() is the granule size on a super-k-mer handle) */
		rc = ensure_buf(h, h->tile_count, h->tile_cap, tiles, 4); if (rc) return rc;
		LinearOp<W, EXT, false> op; op.records = (typename PoolRec<W, EXT>::type *)h->linear; op.koff = h->koff; op.tile_count = h->tile_count; op.first_read_idx = rv.first_read_idx;
		h->sender_launch = dev_pos == nullptr;
		rc = launch_extract<W, EXT>(h, rv, op);
		h->sender_launch = false;
		if (rc) return rc;
		rc = zero_work_counter(h); if (rc) return rc;
		const int grid = (int)std::min<uint64_t>((uint64_t)num_cus(h) * 8, tiles);
		/* who owns a k-mer: getDistributedThreadId, or -- a spectrum that was built through the list exchange -- the list of its minimizer */
		OwnerFn of; of.m = 0; of.off = of.win = of.list_bits = 0;
		if (h->superkmer_mode && h->sk_exchange) { of.m = h->sk_m; of.off = h->sk_off; of.win = h->sk_win; of.list_bits = h->sk_bits; }
		if (dev_pos)
			hipLaunchKernelGGL((owner_scatter_kernel<W, EXT, true>), dim3(grid), dim3(OWNER_THREADS), 0, h->stream, (const typename PoolRec<W, EXT>::type *)h->linear, h->koff, h->tile_count, tiles, h->hkb,
			                   h->cfg.world_size, (uint32_t *)dev_records, seg_capacity, (unsigned long long *)dev_seg_counts, h->work_counter, h->derr, dev_pos, of);
		else
			hipLaunchKernelGGL((owner_scatter_kernel<W, EXT>), dim3(grid), dim3(OWNER_THREADS), 0, h->stream, (const typename PoolRec<W, EXT>::type *)h->linear, h->koff, h->tile_count, tiles, h->hkb,
			                   h->cfg.world_size, (uint32_t *)dev_records, seg_capacity, (unsigned long long *)dev_seg_counts, h->work_counter, h->derr, (uint32_t *)nullptr, of);
		HIPCHK(h, hipGetLastError());
	}
	return 0;
}

template <int W, bool EXT> int add_reads_partition_t(kmr_handle *h, const ReadsView &rvAll, uint64_t total_bases) {
	const uint64_t n = rvAll.n_reads;
	if (!h->l1.head) choose_bits1(h, total_bases);
	const uint64_t avg = n ? std::max<uint64_t>(1, total_bases / n) : 1;
	const uint64_t sub_bases = h->tune.sub_batch_bases ? h->tune.sub_batch_bases : SUB_BATCH_BASES;
	const uint64_t chunk = std::max<uint64_t>(64, (sub_bases / avg) & ~63ull);
	for (uint64_t r = 0; r < n; r += chunk) {
		const uint64_t m = std::min(chunk, n - r);
		ReadsView rv = rvAll;
		rv.offsets = rvAll.offsets + r; rv.n_reads = m;
		rv.discarded = rvAll.discarded ? rvAll.discarded + r : nullptr;
		rv.first_read_idx = rvAll.first_read_idx + r;
		/* k-mer capacity of every work unit -> region of each 64-unit tile in the linear buffer */
		int rc = prepare_units(h, rv); if (rc) return rc;
		const uint64_t nu = rv.u_start ? rv.n_units : m;
		rc = ensure_buf(h, h->kcap, h->kcap_n, nu + 1, 4); if (rc) return rc;
		rc = ensure_buf(h, h->koff, h->koff_n, nu + 1, 8); if (rc) return rc;
		hipLaunchKernelGGL(kmer_capacity_kernel, dim3(grid_for(nu)), dim3(256), 0, h->stream, rv, h->k, h->kcap);
		HIPCHK(h, hipGetLastError());
		rc = exclusive_scan(h, h->kcap, nu, h->koff); if (rc) return rc;
		uint64_t total_cap = 0;
		HIPCHK(h, hipMemcpy(&total_cap, h->koff + nu, 8, hipMemcpyDeviceToHost));
		const uint64_t tiles = (nu + 63) / 64;
		rc = ensure_buf(h, h->linear, h->linear_cap, total_cap, rec_bytes(h)); if (rc) return rc;
		rc = ensure_buf(h, h->tile_count, h->tile_cap, tiles, 4); if (rc) return rc;
		LinearOp<W, EXT> op; op.records = (typename PoolRec<W, EXT>::type *)h->linear; op.koff = h->koff; op.tile_count = h->tile_count; op.first_read_idx = rv.first_read_idx;
		hipEvent_t a, b, a2, b2; time_begin(h, KMR_TIME_BUILD, &a, &b);
		time_begin(h, KMR_TIME_EXTRACT, &a2, &b2);
		rc = launch_extract<W, EXT>(h, rv, op);
		time_end(h, KMR_TIME_EXTRACT, a2, b2);
#ifdef KMR_DEBUG_HOOKS
		if (!rc && getenv("KMR_DEBUG_SAME_TILE")) {
			/* measurement aid (tools/l1_write_side.py): every tile of the level-1 pass reads the records of one of the first N tiles again, i.e. its
			 * input comes out of L2 and only the scatter writes go to HBM -- what the pass would cost if extract fed it from
			 * registers.  The result is not a spectrum. */
			const uint64_t distinct = std::max<uint64_t>(1, strtoull(getenv("KMR_DEBUG_SAME_TILE"), nullptr, 10));
			hipLaunchKernelGGL(same_tile_kernel, dim3(grid_for(tiles)), dim3(256), 0, h->stream, h->koff, tiles, distinct, total_cap / std::max<uint64_t>(tiles, 1));
		}
#endif
		if (!rc) rc = partition_level1<W, EXT>(h, h->linear, h->koff, h->tile_count, tiles, 64, 0, 0, total_cap);
		time_end(h, KMR_TIME_BUILD, a, b);
		if (rc) return rc;
	}
	return 0;
}
int add_reads_partition(kmr_handle *h, const ReadsView &rv, uint64_t total_bases) {
#define ARP(Wv) (h->ext ? add_reads_partition_t<Wv, true>(h, rv, total_bases) : add_reads_partition_t<Wv, false>(h, rv, total_bases))
	switch (h->W) { case 1: return ARP(1); case 2: return ARP(2); case 3: return ARP(3); default: return ARP(4); }
#undef ARP
}

/* chunk CSR of a pool: list_start[nl+1] (device) and list_chunks[n_chunks] (device) */
int build_csr(kmr_handle *h, HostPool &p, uint64_t nl, uint32_t first, uint64_t **list_start, uint64_t **list_chunks, uint32_t *n_chunks_out) {
	unsigned int used = 0;
	HIPCHK(h, hipStreamSynchronize(h->stream));
	HIPCHK(h, hipMemcpy(&used, p.head, 4, hipMemcpyDeviceToHost));
	if (used > p.cap) used = p.cap;
	used = used > first ? used - first : 0;            /* only chunks [first, head) are looked at */
	uint32_t *cnt;
	{ int arc = arena_get(h, &cnt, nl); if (arc) return arc; arc = arena_get(h, list_start, nl + 1); if (arc) return arc;
	  arc = arena_get(h, list_chunks, std::max<unsigned>(used, 1)); if (arc) return arc; }
	HIPCHK(h, hipMemsetAsync(cnt, 0, 4 * nl, h->stream));
	const unsigned csr_grid = (unsigned)(((uint64_t)used + CSR_THREADS * CSR_ITEMS - 1) / (CSR_THREADS * CSR_ITEMS));
	if (used) hipLaunchKernelGGL(chunk_hist_kernel, dim3(csr_grid), dim3(CSR_THREADS), 0, h->stream, p.chunk_list + first, used, cnt, (uint32_t)nl);
	int rc = exclusive_scan(h, cnt, nl, *list_start); if (rc) return rc;
	HIPCHK(h, hipMemsetAsync(cnt, 0, 4 * nl, h->stream));
	if (used) hipLaunchKernelGGL(chunk_scatter_kernel, dim3(csr_grid), dim3(CSR_THREADS), 0, h->stream, p.chunk_list + first, p.chunk_count + first, used, first, *list_start, cnt, *list_chunks, (uint32_t)nl);
	HIPCHK(h, hipGetLastError());
	*n_chunks_out = used;
	if (dbg()) {
		unsigned long long *d, hv[2] = {0, 0};
		HIPCHK(h, dev_malloc((void **)&d, 16)); HIPCHK(h, hipMemset(d, 0, 16));
		if (used) hipLaunchKernelGGL(pool_records_kernel, dim3(grid_for(used)), dim3(256), 0, h->stream, p.chunk_list, p.chunk_count, used, d, d + 1);
		HIPCHK(h, hipStreamSynchronize(h->stream));
		HIPCHK(h, hipMemcpy(hv, d, 16, hipMemcpyDeviceToHost)); hipFree(d);
		fprintf(stderr, "build_csr: lists %llu chunks %u valid %llu records %llu (expected %llu)\n", (unsigned long long)nl, used, hv[1], hv[0], (unsigned long long)h->stats.raw_good_kmers);
		unsigned long long *v, vv[3] = {0, 0, 0};
		int bits = 0; while ((1ull << bits) < nl) bits++;
		HIPCHK(h, dev_malloc((void **)&v, 24)); HIPCHK(h, hipMemset(v, 0, 24));
		PoolView pvw = pool_view(h, p);
#define VLK(Wv, E) hipLaunchKernelGGL((verify_lists_kernel<Wv, E>), dim3(4096), dim3(256), 0, h->stream, pvw, *list_start, *list_chunks, nl, bits, h->hkb, part_rot(h), v, v + 1, v + 2)
		switch (h->W) {
		case 1: if (h->ext) VLK(1, true); else VLK(1, false); break;
		case 2: if (h->ext) VLK(2, true); else VLK(2, false); break;
		case 3: if (h->ext) VLK(3, true); else VLK(3, false); break;
		default: if (h->ext) VLK(4, true); else VLK(4, false); break;
		}
#undef VLK
		HIPCHK(h, hipStreamSynchronize(h->stream));
		HIPCHK(h, hipMemcpy(vv, v, 24, hipMemcpyDeviceToHost)); hipFree(v);
		fprintf(stderr, "verify_lists: records via CSR %llu misfiled %llu zero-weight %llu\n", vv[0], vv[1], vv[2]);
	}
	return 0;
}

int finish_maps_from_entries(kmr_handle *h, uint32_t *wc, uint32_t *sc, uint64_t wslots, uint64_t sslots, uint64_t wn, uint64_t sn, bool keepSing, bool weak_uncounted = false);

template <int W, bool EXT> int finalize_partition_t(kmr_handle *h, uint32_t min_depth) {
	int rc = sync_state(h);
	if (rc) return rc;
	hipEvent_t ea, eb; time_begin(h, 1, &ea, &eb);
	const uint64_t G = h->stats.raw_good_kmers;     /* records in the level-1 pool */
	FinalizeParams f; f.kb = h->hkb; f.ext_min_q = h->cfg.ext_min_quality; f.min_depth = min_depth; f.has_singletons = h->cfg.separate_singletons ? 1 : 0; f.nb_weak = h->nb_weak; f.nb_sing = h->nb_sing; f.uni_wbits = 0;
	const bool keepSing = f.has_singletons && min_depth <= 1;
	if (!h->l1.head) { rc = pool_reserve(h, h->l1, 0, false); if (rc) return rc; }
	rc = arena_reset(h); if (rc) return rc;
	rc = flush_l1_state<W, EXT>(h); if (rc) return rc;
	/* level-1 CSR */
	const uint64_t nl1 = 1ull << h->bits1;
	uint64_t *ls1 = nullptr; uint64_t *lc1 = nullptr; uint32_t nch1 = 0;
	rc = build_csr(h, h->l1, nl1, 0, &ls1, &lc1, &nch1); if (rc) return rc;
	/* Final lists are sized by what the count pass can hold in its LDS table: measure the share of distinct keys
	 * on a sample of level-1 lists, then take enough further bits for ~MAX_LIST_DISTINCT distinct keys per list (and
	 * at most TARGET_LIST_RECORDS records), up to max_part_bits per pass and as many passes as that takes (C2: one,
	 * 10 + 10 bits; C4 with 5 x 10^9 two-word records: 10 + 10 + 2). */
	double distinct_share = 1.0, repeated_share = 0.5;      /* distinct keys, and distinct keys seen more than once, per record */
	if (G) {
		unsigned long long *dpr, hpr[4] = {0, 0, 0, 0};
		const uint32_t n_probes = (uint32_t)std::min<uint64_t>(PROBE_LISTS, nl1);
		const size_t tbytes = 8 * ((size_t)n_probes * PROBE_SLOTS + 4);
		rc = arena_alloc(h, (void **)&dpr, tbytes); if (rc) return rc;
		HIPCHK(h, hipMemsetAsync(dpr, 0, tbytes, h->stream));
		hipLaunchKernelGGL((distinct_probe_kernel<W, EXT>), dim3(n_probes * PROBE_SPLIT), dim3(256), 0, h->stream, pool_view(h, h->l1), ls1, lc1, nl1, h->hkb, part_rot(h),
		                   (int)h->bits1, n_probes, dpr + 4, dpr);
		HIPCHK(h, hipGetLastError());
		HIPCHK(h, hipMemcpyAsync(hpr, dpr, 32, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream));
		if (hpr[0] >= 256) { distinct_share = std::min(1.0, std::max(0.01, (double)hpr[1] / (double)hpr[0])); repeated_share = std::min(0.5, (double)hpr[2] / (double)hpr[0]); }
		if (dbg()) fprintf(stderr, "distinct probe: %llu records, %llu distinct (%llu repeated) -> shares %.3f %.3f\n", hpr[0], hpr[1], hpr[2], distinct_share, repeated_share);
	}
	const int mb = max_part_bits(h);
	int T = 0; while (T < 40 && ((G >> T) > TARGET_LIST_RECORDS || (double)(G >> T) * distinct_share > MAX_LIST_DISTINCT)) T++;
	T = std::min(T, 28);                              /* list ids are 32-bit with room to spare */
	int cur_bits = h->bits1;
	uint64_t *ls2 = ls1, *lc2 = lc1;                  /* CSR of the current (finally: the last) level */
	uint32_t valid_from = 0;                          /* chunks below belong to levels that were left behind by a fresh-chunk pass */
	for (int level = 2; cur_bits < T; level++) {
		const int nbits = std::min(mb, T - cur_bits);
		const uint64_t nl_prev = 1ull << cur_bits;
		/* work items: the lists of the previous level, long ones cut into equal items (every item ends with a flush of
		 * partly filled chunks, and a short leftover item would cost as many of those as a full one) */
		std::vector<uint64_t> hs(nl_prev + 1);
		HIPCHK(h, hipMemcpy(hs.data(), ls2, 8 * (nl_prev + 1), hipMemcpyDeviceToHost));
		std::vector<uint64_t> ib, ie; std::vector<uint32_t> il;
		for (uint64_t l = 0; l < nl_prev; l++) {
			const uint64_t nch = hs[l + 1] - hs[l];
			if (!nch) continue;
			const uint64_t nit = (nch + L2_ITEM_CHUNKS - 1) / L2_ITEM_CHUNKS, per = (nch + nit - 1) / nit;
			for (uint64_t c = hs[l]; c < hs[l + 1]; c += per) { ib.push_back(c); ie.push_back(std::min(hs[l + 1], c + per)); il.push_back((uint32_t)l); }
		}
		if (ib.empty()) { cur_bits += nbits; rc = build_csr(h, h->l1, 1ull << cur_bits, valid_from, &ls2, &lc2, &nch1); if (rc) return rc; continue; }
		/* The pass writes into the pool it reads.  With room for a second copy of the records it appends fresh chunks
		 * (compact, slab by slab: the faster writes); without, a block recycles the chunks it has just read. */
		const uint64_t partials = ib.size() * (1ull << nbits) + (uint64_t)part_grid(h) * 512 + 64;
		unsigned int head_before = 0;
		HIPCHK(h, hipMemcpy(&head_before, h->l1.head, 4, hipMemcpyDeviceToHost));
		const bool recycle = h->tune.recycle >= 0 ? h->tune.recycle != 0 : (uint64_t)head_before + G / CH + partials + 64 > h->l1.cap;
		h->l1.used_ub = head_before;
		rc = pool_reserve(h, h->l1, (recycle ? 0 : G / CH) + partials, true); if (rc) return rc;
		uint64_t *dib, *die; uint32_t *dil;
		rc = arena_get(h, &dib, ib.size()); if (rc) return rc; rc = arena_get(h, &die, ie.size()); if (rc) return rc; rc = arena_get(h, &dil, il.size()); if (rc) return rc;
		HIPCHK(h, hipMemcpyAsync(dib, ib.data(), 8 * ib.size(), hipMemcpyHostToDevice, h->stream)); HIPCHK(h, hipMemcpyAsync(die, ie.data(), 8 * ie.size(), hipMemcpyHostToDevice, h->stream));
		HIPCHK(h, hipMemcpyAsync(dil, il.data(), 4 * il.size(), hipMemcpyHostToDevice, h->stream));
		rc = zero_work_counter(h); if (rc) return rc;
		PartSource<W> S; memset(&S, 0, sizeof(S));
		S.src = pool_view(h, h->l1); S.list_chunks = lc2; S.item_begin = dib; S.item_end = die; S.item_list = dil; S.n_items = ib.size(); S.kb = h->hkb; S.rot = part_rot(h);
		S.recycle = recycle ? 1 : 0;
		const int grid = (int)std::min<uint64_t>(partition_blocks(h), ib.size());
		if (dbg()) fprintf(stderr, "level %d: %d bits after %d, %zu items, %s\n", level, nbits, cur_bits, ib.size(), recycle ? "recycling chunks" : "fresh chunks");
		hipEvent_t ta, tb; time_begin(h, KMR_TIME_PARTITION2, &ta, &tb);
		rc = launch_partition<W, EXT, 2>(h, S, h->l1, grid, nbits, cur_bits);
		time_end(h, KMR_TIME_PARTITION2, ta, tb);
		if (rc) return rc;
		HIPCHK(h, hipStreamSynchronize(h->stream));      /* the host vectors behind the item copies go out of scope */
		cur_bits += nbits;
		/* CSR of the new level: fresh chunks lie behind the old head (what is below keeps the list ids of the level
		 * left behind and is never looked at again); recycled ones anywhere in the range that was valid before */
		if (!recycle) valid_from = head_before;
		rc = build_csr(h, h->l1, 1ull << cur_bits, valid_from, &ls2, &lc2, &nch1); if (rc) return rc;
	}
	const uint64_t nl2 = 1ull << cur_bits;
	const int count_log2s = (!EXT && (double)(G >> cur_bits) * distinct_share > MAX_LIST_DISTINCT) ? 11 : COUNT_LOG2S;      /* with extension tallies 2048 slots do not fit LDS */
	if (dbg()) fprintf(stderr, "count pass: %llu lists of ~%llu records, table 2^%d\n", (unsigned long long)nl2, (unsigned long long)(G >> cur_bits), count_log2s);
	const uint32_t vw = EXT ? 15 : 3;
	const uint64_t slack = (uint64_t)part_grid(h) * 8 * 8192 + 16;     /* one partly used output slab per block */
	/* entry buffers: the worst case (every second record a weak entry, or every record a singleton) is 5-10 x what
	 * sequencing data produces, and at C4 size it is 70 GB; they are sized from the probe's shares with 50 % headroom and
	 * the count pass is simply run again with larger ones if that was not enough */
	/* upper bounds of the kept entries; the count pass retires an output slab that cannot take a list's entries whole, so up to
	 * (entries of one list - 1) of every 8192-slot slab stay unused: an eighth on top */
	const uint64_t wbound = f.has_singletons ? G / 2 : G, sbound = keepSing ? G : 0;
	const uint64_t wmax = wbound + wbound / 8 + slack, smax = keepSing ? sbound + sbound / 8 + slack : 16;
	uint64_t wcap = std::min<uint64_t>(wmax, (uint64_t)((double)G * (f.has_singletons ? repeated_share : distinct_share) * 1.5) + G / 64 + slack);
	uint64_t scap = keepSing ? std::min<uint64_t>(smax, (uint64_t)((double)G * std::max(0.0, distinct_share - repeated_share) * 1.5) + G / 64 + slack) : 16;
	if (h->tune.entry_share >= 0) {       /* kmr_tune "entry_share": start from a given (e.g. hopeless) estimate so that the retry below has to run */
		const double sh = h->tune.entry_share;
		wcap = std::min<uint64_t>(wmax, (uint64_t)((double)G * sh) + 16384); if (keepSing) scap = std::min<uint64_t>(smax, (uint64_t)((double)G * sh) + 16384);
		if (h->uw_keys) { hipFree(h->uw_keys); hipFree(h->uw_vals); h->uw_keys = h->uw_vals = nullptr; h->uw_cap = 0; }
		if (h->us_keys) { hipFree(h->us_keys); hipFree(h->us_b8); if (h->us_pkt) hipFree(h->us_pkt); h->us_keys = h->us_b8 = h->us_pkt = nullptr; h->us_cap = 0; }
	}
	if (h->uw_keys && h->uw_cap >= wcap) wcap = h->uw_cap;
	if (h->us_keys && h->us_cap >= scap) scap = h->us_cap;
	uint32_t *wc = nullptr, *sc = nullptr; FinalizeCounters *fc = nullptr; unsigned long long *cursors = nullptr;
	rc = arena_get(h, &wc, h->nb_weak); if (rc) return rc; rc = arena_get(h, &sc, h->nb_sing); if (rc) return rc;
	rc = arena_get(h, &fc, 1); if (rc) return rc; rc = arena_get(h, &cursors, 2); if (rc) return rc;
	FinalizeCounters c; unsigned long long cur[2];
	hipEvent_t tca, tcb; time_begin(h, KMR_TIME_COUNT, &tca, &tcb);
	for (int attempt = 0; ; attempt++) {
	{	/* the linear record buffer is dead during finalize: given back when the entry buffers would not fit beside it */
		const double need = (!h->uw_keys || h->uw_cap < wcap ? (8.0 * W + 4.0 * vw) * (double)wcap : 0.0) + (!h->us_keys || h->us_cap < scap ? (8.0 * W + 1.0) * (double)scap : 0.0);
		size_t mfree = 0, mtotal = 0;
		if (need > 0 && h->linear && hipMemGetInfo(&mfree, &mtotal) == hipSuccess && (double)mfree < need + (double)G * 0.3 * (8.0 * W + 12.0) + (double)(2ull << 30)) {
			HIPCHK(h, hipStreamSynchronize(h->stream));
			hipFree(h->linear); h->linear = nullptr; h->linear_cap = 0;
		}
	}
	if (!h->uw_keys || h->uw_cap < wcap) {
		if (h->uw_keys) hipFree(h->uw_keys); if (h->uw_vals) hipFree(h->uw_vals); h->uw_keys = h->uw_vals = nullptr; h->uw_cap = 0;
		HIPCHK(h, dev_malloc(&h->uw_keys, 8ull * W * wcap)); HIPCHK(h, dev_malloc(&h->uw_vals, 4ull * vw * wcap)); h->uw_cap = wcap;
	}
	if (!h->us_keys || h->us_cap < scap) {
		if (h->us_keys) hipFree(h->us_keys); if (h->us_b8) hipFree(h->us_b8); if (h->us_pkt) hipFree(h->us_pkt); h->us_keys = h->us_b8 = h->us_pkt = nullptr; h->us_cap = 0;
		HIPCHK(h, dev_malloc(&h->us_keys, 8ull * W * scap)); HIPCHK(h, dev_malloc(&h->us_b8, scap)); if (EXT) HIPCHK(h, dev_malloc(&h->us_pkt, 4 * scap)); h->us_cap = scap;
	}
	HIPCHK(h, hipMemsetAsync(wc, 0, 4 * h->nb_weak, h->stream)); HIPCHK(h, hipMemsetAsync(sc, 0, 4 * h->nb_sing, h->stream));
	HIPCHK(h, hipMemsetAsync(fc, 0, sizeof(FinalizeCounters), h->stream)); HIPCHK(h, hipMemsetAsync(cursors, 0, 16, h->stream));
	CountOut out; out.wkeys = (uint64_t *)h->uw_keys; out.wvals = (uint32_t *)h->uw_vals; out.wentries = nullptr; out.wcursor = cursors; out.wcap = h->uw_cap;
	out.skeys = (uint64_t *)h->us_keys; out.sweight = (uint8_t *)h->us_b8; out.spkt = (uint32_t *)h->us_pkt; out.scursor = cursors + 1; out.scap = h->us_cap;
	out.weakCount = wc; out.singCount = sc; out.fc = fc; out.err = h->derr;
	rc = zero_work_counter(h); if (rc) return rc;
	#ifdef KMR_DEBUG_HOOKS
	const int count_reps = getenv("KMR_COUNT_CHECK") ? atoi(getenv("KMR_COUNT_CHECK")) : 0;
#else
	const int count_reps = 0;
#endif
	for (int cr = 0; cr <= count_reps; cr++) {
		if (cr) {      /* debugging aid: the count pass is repeated on the same input and must report the same numbers */
			FinalizeCounters c0; unsigned long long cur0[2];
			HIPCHK(h, hipStreamSynchronize(h->stream));
			HIPCHK(h, hipMemcpy(&c0, fc, sizeof(c0), hipMemcpyDeviceToHost)); HIPCHK(h, hipMemcpy(cur0, cursors, 16, hipMemcpyDeviceToHost));
			fprintf(stderr, "count pass %d: unique %llu singletons %llu weak_kept %llu sing_kept %llu slots %llu/%llu\n", cr - 1, (unsigned long long)c0.unique,
			        (unsigned long long)c0.singletons, (unsigned long long)c0.weak_kept, (unsigned long long)c0.sing_kept, cur0[0], cur0[1]);
			HIPCHK(h, hipMemsetAsync(wc, 0, 4 * h->nb_weak, h->stream)); HIPCHK(h, hipMemsetAsync(sc, 0, 4 * h->nb_sing, h->stream));
			HIPCHK(h, hipMemsetAsync(fc, 0, sizeof(FinalizeCounters), h->stream)); HIPCHK(h, hipMemsetAsync(cursors, 0, 16, h->stream));
			rc = zero_work_counter(h); if (rc) return rc;
		}
		const int grid = (int)std::min<uint64_t>((uint64_t)part_grid(h) * 2, nl2);
		if (count_log2s == 11) {
			auto kern = count_kernel<W, EXT, 11>;
			const size_t smem = count_smem_bytes<W, EXT, 11>();
			HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
			hipLaunchKernelGGL(kern, dim3(grid), dim3(COUNT_THREADS), smem, h->stream, pool_view(h, h->l1), ls2, lc2, nl2, out, f, h->work_counter, 0);
		} else if (EXT && W == 1 && !h->tune.no_narrow) {
			/* extension values at k <= 32: 16-bit tallies for every list of at most 65 535 records (two blocks per CU), then
			 * the wide table for whatever is longer */
			auto kn = count_kernel<W, EXT, COUNT_LOG2S, true>;
			const size_t sn = count_smem_bytes<W, EXT, COUNT_LOG2S, true>();
			HIPCHK(h, hipFuncSetAttribute((const void *)kn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sn));
			hipLaunchKernelGGL(kn, dim3(grid), dim3(COUNT_THREADS), sn, h->stream, pool_view(h, h->l1), ls2, lc2, nl2, out, f, h->work_counter, 1);
			HIPCHK(h, hipGetLastError());
			/* is any list longer than the narrow tallies can take?  (the work counter word doubles as the maximum) */
			rc = zero_work_counter(h); if (rc) return rc;
			hipLaunchKernelGGL(max_list_chunks_kernel, dim3(grid_for(nl2)), dim3(256), 0, h->stream, ls2, nl2, h->work_counter);
			unsigned int longest = 0;
			HIPCHK(h, hipMemcpyAsync(&longest, h->work_counter, 4, hipMemcpyDeviceToHost, h->stream));
			HIPCHK(h, hipStreamSynchronize(h->stream));
			rc = zero_work_counter(h); if (rc) return rc;
			if (longest > COUNT_NARROW_CHUNKS) {
				auto kern = count_kernel<W, EXT, COUNT_LOG2S>;
				const size_t smem = count_smem_bytes<W, EXT, COUNT_LOG2S>();
				HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
				hipLaunchKernelGGL(kern, dim3(std::min(grid, part_grid(h))), dim3(COUNT_THREADS), smem, h->stream, pool_view(h, h->l1), ls2, lc2, nl2, out, f, h->work_counter, 2);
			}
		} else {
			auto kern = count_kernel<W, EXT, COUNT_LOG2S>;
			const size_t smem = count_smem_bytes<W, EXT, COUNT_LOG2S>();
			HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
			hipLaunchKernelGGL(kern, dim3(grid), dim3(COUNT_THREADS), smem, h->stream, pool_view(h, h->l1), ls2, lc2, nl2, out, f, h->work_counter, 0);
		}
		HIPCHK(h, hipGetLastError());
	}
	uint32_t cerr = 0;
	HIPCHK(h, hipMemcpyAsync(&c, fc, sizeof(c), hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipMemcpyAsync(cur, cursors, 16, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipMemcpyAsync(&cerr, h->derr, 4, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	if (!(cerr & ERR_ENTRIES_FULL) || (wcap >= wmax && scap >= smax) || attempt >= 8) break;
	/* more kept entries than the probe promised: larger buffers, same pass again */
	cerr &= ~(uint32_t)ERR_ENTRIES_FULL;
	HIPCHK(h, hipMemcpy(h->derr, &cerr, 4, hipMemcpyHostToDevice));
	wcap = std::min<uint64_t>(wmax, wcap * 2); if (keepSing) scap = std::min<uint64_t>(smax, scap * 2);
	if (dbg()) fprintf(stderr, "count pass: entry buffers too small, retrying with %llu / %llu\n", (unsigned long long)wcap, (unsigned long long)scap);
	}
	time_end(h, KMR_TIME_COUNT, tca, tcb);
	{	/* still full with the buffers at their bounds (or after the last retry): the cursors point past the buffers, nothing
		 * downstream may use them */
		uint32_t cerr2 = 0;
		HIPCHK(h, hipMemcpy(&cerr2, h->derr, 4, hipMemcpyDeviceToHost));
		if (cerr2 & ERR_ENTRIES_FULL) { time_end(h, 1, ea, eb); return fail(h, KMR_ERR_CAPACITY, "entry buffers of the count pass overflowed at their upper bound (internal sizing error)"); }
	}

	h->stats.unique_kmers = c.unique;
	h->stats.singleton_kmers = f.has_singletons ? c.singletons : 0;
	hipEvent_t tma, tmb; time_begin(h, KMR_TIME_BUCKETS, &tma, &tmb);
	rc = finish_maps_from_entries(h, wc, sc, cur[0], cur[1], c.weak_kept, c.sing_kept, keepSing);
	time_end(h, KMR_TIME_BUCKETS, tma, tmb);
	if (rc) return rc;
	time_end(h, 1, ea, eb);
	h->has_singletons = keepSing;
	h->stats.weak_entries = h->weak.n; h->stats.singleton_entries = keepSing ? h->sing.n : 0;
	h->finalized = true; h->map_gen++;
	rc = sync_state(h);
	/* the temporaries are dead: if some of them had to be allocated on the side, the arena is brought to size now,
	 * so that it is this build (a handle's first) that pays for it and not the next one */
	if (!rc && !h->arena_overflow.empty()) rc = arena_reset(h);
	return rc;
}

/* The weak map out of the count pass's packed entries (h->ue) by the radix partition of kmr_buckets.hpp (COUNT_DIR values).
 * done = false and nothing changed when the geometry does not fit: the caller takes the scatter + per-bucket sort. */
template <int W> int binned_buckets_t(kmr_handle *h, uint64_t wslots, uint64_t wn, bool &done) {
	done = false;
	DevMap &wm = h->weak;
	const uint64_t nb = wm.nb;
	uint32_t B = 0; while ((1ull << (B + 1)) <= nb) B++;
	if (h->ext || wn < h->tune.binned_min || wn == 0 || (1ull << B) != nb || wslots >= (1ull << 40)) return 0;
	uint32_t g = 0; while (g < std::min<uint32_t>(B, BB_MAX_GROUP_BITS) && (wn >> (B - g)) < 512) g++;
	const uint32_t R = B - g;
	if (R < 1 || R > 2 * BB_MAX_BITS || (wn >> R) > 1024) return 0;
	const uint32_t bits1 = R <= (uint32_t)BB_MAX_BITS ? R : (R + 1) / 2, bits2 = R - bits1;
	const uint64_t bins1 = 1ull << bits1, groups = 1ull << R;
	uint32_t *hist1 = nullptr, *hist2 = nullptr, *pad1 = nullptr; uint64_t *start1 = nullptr, *gstart = nullptr; unsigned long long *cursor = nullptr; unsigned int *dmax = nullptr;
	int rc = arena_get(h, &hist1, bins1); if (rc) return rc;
	rc = arena_get(h, &start1, bins1 + 1); if (rc) return rc;
	rc = arena_get(h, &cursor, groups); if (rc) return rc;
	rc = arena_get(h, &dmax, 1); if (rc) return rc;
	if (bits2) { rc = arena_get(h, &hist2, groups); if (rc) return rc; rc = arena_get(h, &pad1, bins1); if (rc) return rc; rc = arena_get(h, &gstart, groups + 1); if (rc) return rc; }
	HIPCHK(h, hipMemsetAsync(hist1, 0, 4 * bins1, h->stream)); HIPCHK(h, hipMemsetAsync(dmax, 0, 4, h->stream));
	BbInput in1; in1.entries = h->ue; in1.seg_start = nullptr; in1.seg_count = nullptr; in1.n_seg = 1; in1.n_slots = wslots; in1.holes = 1;
	auto hist_grid = [&](uint64_t n_slots, uint32_t &tpb) { const uint64_t tiles = (n_slots + BB_TILE - 1) / BB_TILE; tpb = (uint32_t)std::max<uint64_t>(1, (tiles + 2047) / 2048); return (unsigned)((tiles + tpb - 1) / tpb); };
	auto scatter_grid = [&](uint64_t n_slots) { const uint64_t tiles = (n_slots + BB_TILE - 1) / BB_TILE; return (unsigned)std::min<uint64_t>(tiles, (uint64_t)num_cus(h) * 8); };
	auto scratch = [&](uint64_t entries) -> int {      /* h->ue2: the other side of the partition's ping-pong */
		if (h->ue2 && h->ue2_cap >= entries) return 0;
		if (h->ue2) hipFree(h->ue2); h->ue2 = nullptr; h->ue2_cap = 0;
		HIPCHK(h, dev_malloc((void **)&h->ue2, 8ull * (W + 1) * entries)); h->ue2_cap = entries;
		return 0;
	};
	auto group_launch = [&](const uint64_t *entries, const uint64_t *gs, const uint32_t *gc) -> int {
		int rc2 = reserve_bytes(h, (void **)&wm.keys, wm.c_keys, 8ull * W * wn); if (rc2) return rc2;
		rc2 = reserve_bytes(h, (void **)&wm.vals, wm.c_vals, 12ull * wn); if (rc2) return rc2;
		auto gk = bb_group_kernel<W>;
		HIPCHK(h, hipFuncSetAttribute((const void *)gk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bb_group_smem_bytes<W>()));
		/* the blocks stride over the groups: as many of them as fit the chip at once, so that they all get the same number of groups */
		int per_cu = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)gk, BB_GROUP_THREADS, bb_group_smem_bytes<W>()) != hipSuccess || per_cu < 1) per_cu = 4;
		if (dbg()) fprintf(stderr, "bb_group<W=%d>: %d blocks per CU, %llu groups\n", W, per_cu, (unsigned long long)groups);
		hipLaunchKernelGGL(gk, dim3((unsigned)std::min<uint64_t>(groups, (uint64_t)num_cus(h) * per_cu)), dim3(BB_GROUP_THREADS), bb_group_smem_bytes<W>(), h->stream, entries, wm.keys, wm.vals, gs, gc, groups, g, h->hkb, nb, wm.start, wn, h->derr);
		HIPCHK(h, hipGetLastError());
		return 0;
	};
	const uint32_t shift1 = B - bits1;
	uint32_t tpb = 1; unsigned grid = hist_grid(wslots, tpb);
	hipLaunchKernelGGL(bb_hist_kernel<W>, dim3(grid), dim3(BB_THREADS), 0, h->stream, in1, shift1, bits1, h->hkb, nb, tpb, hist1);
	unsigned int mx = 0;
	if (!bits2) {
		/* one level: its bins are the groups, their scan is where the groups lie in the map */
		hipLaunchKernelGGL(bb_pad_kernel, dim3(grid_for(bins1)), dim3(256), 0, h->stream, (const uint32_t *)hist1, bins1, (uint32_t *)nullptr, dmax);
		HIPCHK(h, hipMemcpyAsync(&mx, dmax, 4, hipMemcpyDeviceToHost, h->stream));
		rc = exclusive_scan(h, hist1, bins1, start1); if (rc) return rc;      /* (synchronises) */
		if (mx > bb_group_cap<W>()) return 0;
		rc = scratch(wn); if (rc) return rc;
		hipLaunchKernelGGL(bb_cursor_init_kernel, dim3(grid_for(bins1)), dim3(256), 0, h->stream, (const uint64_t *)start1, bins1, cursor);
		hipLaunchKernelGGL(bb_scatter_kernel<W>, dim3(scatter_grid(wslots)), dim3(BB_THREADS), 0, h->stream, in1, shift1, bits1, h->hkb, nb, cursor, h->ue2);
		rc = group_launch(h->ue2, start1, hist1); if (rc) return rc;
		done = true;
		return 0;
	}
	/* two levels: the first one's bins start at tile boundaries (the second level reads them tile by tile) */
	hipLaunchKernelGGL(bb_pad_kernel, dim3(grid_for(bins1)), dim3(256), 0, h->stream, (const uint32_t *)hist1, bins1, pad1, dmax);
	rc = exclusive_scan(h, pad1, bins1, start1); if (rc) return rc;
	uint64_t padded_total = 0;
	HIPCHK(h, hipMemcpy(&padded_total, start1 + bins1, 8, hipMemcpyDeviceToHost));
	rc = scratch(std::max(padded_total, wn)); if (rc) return rc;
	hipLaunchKernelGGL(bb_cursor_init_kernel, dim3(grid_for(bins1)), dim3(256), 0, h->stream, (const uint64_t *)start1, bins1, cursor);
	hipLaunchKernelGGL(bb_scatter_kernel<W>, dim3(scatter_grid(wslots)), dim3(BB_THREADS), 0, h->stream, in1, shift1, bits1, h->hkb, nb, cursor, h->ue2);
	BbInput in2; in2.entries = h->ue2; in2.seg_start = start1; in2.seg_count = hist1; in2.n_seg = (uint32_t)bins1; in2.n_slots = padded_total; in2.holes = 0;
	const uint32_t shift2 = g;
	HIPCHK(h, hipMemsetAsync(hist2, 0, 4 * groups, h->stream)); HIPCHK(h, hipMemsetAsync(dmax, 0, 4, h->stream));
	grid = hist_grid(padded_total, tpb);
	hipLaunchKernelGGL(bb_hist_kernel<W>, dim3(grid), dim3(BB_THREADS), 0, h->stream, in2, shift2, bits2, h->hkb, nb, tpb, hist2);
	hipLaunchKernelGGL(bb_pad_kernel, dim3(grid_for(groups)), dim3(256), 0, h->stream, (const uint32_t *)hist2, groups, (uint32_t *)nullptr, dmax);
	HIPCHK(h, hipMemcpyAsync(&mx, dmax, 4, hipMemcpyDeviceToHost, h->stream));
	rc = exclusive_scan(h, hist2, groups, gstart); if (rc) return rc;
	/* the packed entries the count pass wrote are about to be overwritten (the second level writes where the first one read): a
	 * group too large for the LDS arrays sends the build down the other path BEFORE that */
	if (mx > bb_group_cap<W>() || h->ue_cap < wn) return 0;
	hipLaunchKernelGGL(bb_cursor_init_kernel, dim3(grid_for(groups)), dim3(256), 0, h->stream, (const uint64_t *)gstart, groups, cursor);
	hipLaunchKernelGGL(bb_scatter_kernel<W>, dim3(scatter_grid(padded_total)), dim3(BB_THREADS), 0, h->stream, in2, shift2, bits2, h->hkb, nb, cursor, h->ue);
	rc = group_launch(h->ue, gstart, hist2); if (rc) return rc;
	done = true;
	return 0;
}

/* weak_uncounted: the count pass kept no per-bucket counts of the weak entries (wc is scratch): the radix partition of
 * kmr_buckets.hpp needs none, and if it declines they are counted here */
template <int W> int finish_maps_t(kmr_handle *h, uint32_t *wc, uint32_t *sc, uint64_t wslots, uint64_t sslots, uint64_t wn, uint64_t sn, bool keepSing, bool weak_uncounted) {
	const uint32_t vw = h->ext ? 15 : 3;
	DevMap &wm = h->weak, &sm = h->sing;
	clear_map(wm); clear_map(sm);          /* the buffers of the previous build are reused when they are large enough */
	wm.nb = h->nb_weak; wm.n = wn; wm.present = true;
	sm.nb = h->nb_sing; sm.n = keepSing ? sn : 0; sm.present = keepSing;
	int rc = reserve_bytes(h, (void **)&wm.start, wm.c_start, 8 * (wm.nb + 1)); if (rc) return rc;
	rc = reserve_bytes(h, (void **)&sm.start, sm.c_start, 8 * (sm.nb + 1)); if (rc) return rc;
	bool weakDone = false;
	if (weak_uncounted) {
		rc = binned_buckets_t<W>(h, wslots, wn, weakDone); if (rc) return rc;
		if (!weakDone) {      /* the other path wants keys and values apart and a count per bucket */
			if (!h->uw_keys || h->uw_cap < wslots) {
				if (h->uw_keys) hipFree(h->uw_keys); if (h->uw_vals) hipFree(h->uw_vals); h->uw_keys = h->uw_vals = nullptr; h->uw_cap = 0;
				const uint64_t cap = std::max<uint64_t>(wslots, 16);
				HIPCHK(h, dev_malloc(&h->uw_keys, 8ull * W * cap)); HIPCHK(h, dev_malloc(&h->uw_vals, 12ull * cap)); h->uw_cap = cap;
			}
			HIPCHK(h, hipMemsetAsync(wc, 0, 4 * wm.nb, h->stream));
			if (wslots) hipLaunchKernelGGL(bb_unpack_kernel<W>, dim3(grid_for(wslots)), dim3(256), 0, h->stream, (const uint64_t *)h->ue, wslots, h->hkb, wm.nb, (uint64_t *)h->uw_keys, (uint32_t *)h->uw_vals, wc);
			HIPCHK(h, hipGetLastError());
		}
	}
	if (!weakDone) { rc = exclusive_scan(h, wc, wm.nb, wm.start); if (rc) return rc; }
	if (keepSing) { rc = exclusive_scan(h, sc, sm.nb, sm.start); if (rc) return rc; }
	else HIPCHK(h, hipMemsetAsync(sm.start, 0, 8 * (sm.nb + 1), h->stream));      /* no singleton map is kept: every bucket starts (and ends) at 0 */
	if (!weakDone) {
		rc = reserve_bytes(h, (void **)&wm.keys, wm.c_keys, 8ull * W * wm.n); if (rc) return rc;
		rc = reserve_bytes(h, (void **)&wm.vals, wm.c_vals, 4ull * vw * wm.n); if (rc) return rc;
	}
	rc = reserve_bytes(h, (void **)&sm.keys, sm.c_keys, 8ull * W * sm.n); if (rc) return rc;
	rc = reserve_bytes(h, (void **)&sm.sweight, sm.c_sw, sm.n); if (rc) return rc;
	if (h->ext) { rc = reserve_bytes(h, (void **)&sm.spkt, sm.c_pkt, 4 * sm.n); if (rc) return rc; }
	HIPCHK(h, hipMemsetAsync(wc, 0, 4 * wm.nb, h->stream)); HIPCHK(h, hipMemsetAsync(sc, 0, 4 * sm.nb, h->stream));
	if (weakDone) { /* keys, values and bucket starts are in place */ }
	else if (wm.n && vw > 4) hipLaunchKernelGGL((entry_scatter_kernel<W, true>), dim3(grid_for(wslots)), dim3(256), 0, h->stream, (const uint64_t *)h->uw_keys, (const uint32_t *)h->uw_vals,
	                            (const uint8_t *)nullptr, (const uint32_t *)nullptr, wslots, vw, h->hkb, wm.nb, wm.start, wc, wm.keys, wm.vals, (uint8_t *)nullptr, (uint32_t *)nullptr);
	else if (wm.n) {
		/* lists cut by minimizer (build_mode 3) scatter their entries over unrelated buckets: 64-bit cursors that start at the buckets'
		 * first entries, one returning add per entry */
		unsigned long long *cur64 = nullptr;
		if (h->superkmer_mode && arena_get(h, &cur64, wm.nb) == 0) HIPCHK(h, hipMemcpyAsync(cur64, wm.start, 8 * wm.nb, hipMemcpyDeviceToDevice, h->stream));
		else cur64 = nullptr;
		hipLaunchKernelGGL(entry_scatter_kernel<W>, dim3(grid_for(wslots)), dim3(256), 0, h->stream, (const uint64_t *)h->uw_keys, (const uint32_t *)h->uw_vals,
		                            (const uint8_t *)nullptr, (const uint32_t *)nullptr, wslots, vw, h->hkb, wm.nb, wm.start, wc, wm.keys, wm.vals, (uint8_t *)nullptr, (uint32_t *)nullptr, cur64, 6);
	}
	if (sm.n) hipLaunchKernelGGL(entry_scatter_kernel<W>, dim3(grid_for(sslots)), dim3(256), 0, h->stream, (const uint64_t *)h->us_keys, (const uint32_t *)nullptr,
	                            (const uint8_t *)h->us_b8, (const uint32_t *)h->us_pkt, sslots, 0u, h->hkb, sm.nb, sm.start, sc, sm.keys, (uint32_t *)nullptr, sm.sweight, h->ext ? sm.spkt : (uint32_t *)nullptr);
	HIPCHK(h, hipGetLastError());
	SortView<W> sv; sv.keys = wm.keys; sv.vals = wm.vals; sv.b8 = nullptr; sv.pkt = nullptr; sv.vw = vw;
	if (weakDone) { /* sorted by bb_group_kernel */ }
	else if (h->ext) hipLaunchKernelGGL((sort_buckets_kernel<W, 15>), dim3(grid_for(wm.nb, 4, 1 << 20)), dim3(256), 0, h->stream, sv, wm.start, wm.nb);
	else hipLaunchKernelGGL((sort_buckets_kernel<W, 3>), dim3(grid_for(wm.nb, 4, 1 << 20)), dim3(256), 0, h->stream, sv, wm.start, wm.nb);
	if (sm.n) {
		SortView<W> ss; ss.keys = sm.keys; ss.vals = nullptr; ss.b8 = sm.sweight; ss.pkt = h->ext ? sm.spkt : nullptr; ss.vw = 0;
		hipLaunchKernelGGL((sort_buckets_kernel<W, 0>), dim3(grid_for(sm.nb, 4, 1 << 20)), dim3(256), 0, h->stream, ss, sm.start, sm.nb);
	}
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipStreamSynchronize(h->stream));
	return 0;
}
int finish_maps_from_entries(kmr_handle *h, uint32_t *wc, uint32_t *sc, uint64_t wslots, uint64_t sslots, uint64_t wn, uint64_t sn, bool keepSing, bool weak_uncounted) {
	switch (h->W) { case 1: return finish_maps_t<1>(h, wc, sc, wslots, sslots, wn, sn, keepSing, weak_uncounted); case 2: return finish_maps_t<2>(h, wc, sc, wslots, sslots, wn, sn, keepSing, weak_uncounted);
	case 3: return finish_maps_t<3>(h, wc, sc, wslots, sslots, wn, sn, keepSing, weak_uncounted); default: return finish_maps_t<4>(h, wc, sc, wslots, sslots, wn, sn, keepSing, weak_uncounted); }
}
int finalize_partition(kmr_handle *h, uint32_t min_depth) {
#define FPT(Wv) (h->ext ? finalize_partition_t<Wv, true>(h, min_depth) : finalize_partition_t<Wv, false>(h, min_depth))
	switch (h->W) { case 1: return FPT(1); case 2: return FPT(2); case 3: return FPT(3); default: return FPT(4); }
#undef FPT
}
template <int W, bool EXT> int insert_records_partition_t(kmr_handle *h, const void *recs, uint64_t n) {
	if (!h->l1.head) choose_bits1(h, n);
	hipEvent_t a, b; time_begin(h, 0, &a, &b);
	/* received segments contain holes (weight 0): the device counts the real records into stats.raw/good */
	int rc = partition_level1<W, EXT>(h, recs, nullptr, nullptr, (n + 8191) / 8192, 0, 8192, n, n, &h->dstats->inserted,
	                             2 * W + (h->ext ? 2 : 1), h->stream_base);
	time_end(h, 0, a, b);
	return rc;
}
int insert_records_partition(kmr_handle *h, const void *recs, uint64_t n) {
#define IRP(Wv) (h->ext ? insert_records_partition_t<Wv, true>(h, recs, n) : insert_records_partition_t<Wv, false>(h, recs, n))
	switch (h->W) { case 1: return IRP(1); case 2: return IRP(2); case 3: return IRP(3); default: return IRP(4); }
#undef IRP
}
/* ---------------------------------------------------------------------- */
/* build_mode 3: super-k-mer lists (kmr_superkmer.hpp)                        */
/* Minimizer geometry for k: the window of WIN m-mer offsets sits in the middle of the k-mer (2 * off + WIN = k - m + 1), m is
 * the largest length <= 16 (one 32-bit word) of the right parity, WIN the largest of 16 / 8 / 4 that leaves m >= 10. */
bool sk_geometry(uint32_t k, uint32_t m_wish, uint32_t &win, uint32_t &m, uint32_t &off, uint32_t win_max = 16) {
	/* (a window of 32 offsets -- runs of 16.5 k-mers, half the records and list appends of a window of 16 -- exists where it leaves
	 * m >= 14 (k >= 45, keys of two words and more) but is NOT the default: on C4 it takes the extraction from 40.7 to 28.4 ms and the count
	 * pass from 115 to 257 ms -- one place of the genome then puts 33 x coverage k-mers into ONE list, lists of 700 k-mers hold one or
	 * three such places, and the long ones overflow the LDS table into sub-passes; kmr_tune "superkmer_window" = 32 asks for it) */
	/* (a window of 18 offsets at k = 31 -- minimizers of 14 bases, 10 % fewer records -- was built and measured too: extraction 7.57 -> 6.98 ms,
	 * count pass 9.56 -> 10.92 ms on C2; its instances are no longer compiled, the kernels still take any window 16 + {1, 2, 4, 8, 16}) */
	for (uint32_t w : {32u, 16u, 8u, 4u}) {
		if (w > win_max) continue;
		if (w == 32u && (k < 45u || (m_wish && m_wish < 14u))) continue;
		if (k < w + 9) continue;
		uint32_t mm = std::min<uint32_t>(m_wish ? m_wish : 16u, k - w + 1);
		if (mm > 16) mm = 16;
		if (((k - mm + 1 - w) & 1u) != 0) mm--;                 /* k - m + 1 - WIN must be even */
		if (mm < 10 && !m_wish) continue;
		if (mm < 4) continue;
		win = w; m = mm; off = (k - mm + 1 - w) / 2;
		return true;
	}
	return false;
}
const uint64_t SK_EXTRACT_WAVES_PER_CU = std::max(2 * SK_WAVES, SKL_MIN_BLOCKS * SKL_WAVES);      /* wavefronts an extraction launch keeps per CU (each holds two slabs of 64 chunks) */
template <int W, int WIN, bool FILT, bool EXT = false> int launch_sk_extract(kmr_handle *h, const ReadsView &rv, const SkParams &sp, const DevParams *override_params = nullptr) {
	auto kern = sk_extract_kernel<W, WIN, FILT, EXT>;
	HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_EXTRACT_SMEM));
	const uint64_t tiles = ((rv.u_start ? rv.n_units : rv.n_reads) + 63) / 64;
	uint64_t blocks = (tiles + SK_WAVES - 1) / SK_WAVES;
	if (blocks == 0) return 0;
	blocks = std::min<uint64_t>(blocks, (uint64_t)num_cus(h) * 2);      /* resident grid: a wavefront walks tiles tile0, tile0 + stride, ... */
	if (dbg()) { int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, SK_WAVES * 64, SK_EXTRACT_SMEM); fprintf(stderr, "sk_extract<W=%d,WIN=%d>: %d blocks of %d waves per CU (LDS %zu), grid %llu, m=%u off=%u bits=%u\n", W, WIN, nb, SK_WAVES, SK_EXTRACT_SMEM, (unsigned long long)blocks, sp.m, sp.off, sp.list_bits); }
	hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SK_WAVES * 64), SK_EXTRACT_SMEM, h->stream, rv, override_params ? *override_params : dev_params(h), sp, pool_view(h, h->l1));
	HIPCHK(h, hipGetLastError());
	return 0;
}
uint32_t sk_dbg_flags(const char *name);
bool sp_debug_extract(kmr_handle *h) { (void)h; return sk_dbg_flags("KMR_SK_EXTRACT_DBG") != 0; }      /* the ablation switches live in the general kernel */
template <int W, int WIN, bool PACKED = false> int launch_sk_extract_lean(kmr_handle *h, const ReadsView &rv, const SkParams &sp, float wK, const DevParams *override_params = nullptr, const SkPacked *packed = nullptr) {
	auto kern = sk_extract_lean_kernel<W, WIN, PACKED>;
	SkPacked pkd; pkd.bytes = nullptr; pkd.off = nullptr; pkd.mk_off = nullptr; pkd.mk_pos = nullptr; pkd.mk_char = nullptr;
	if (PACKED) pkd = *packed;
	HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SKL_EXTRACT_SMEM));
	const uint64_t tiles = ((rv.u_start ? rv.n_units : rv.n_reads) + 63) / 64;
	uint64_t blocks = (tiles + SKL_WAVES - 1) / SKL_WAVES;
	if (blocks == 0) return 0;
	blocks = std::min<uint64_t>(blocks, (uint64_t)num_cus(h) * SKL_MIN_BLOCKS);      /* resident grid: a wavefront walks tiles tile0, tile0 + stride, ... */
	hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SKL_WAVES * 64), SKL_EXTRACT_SMEM, h->stream, rv, override_params ? *override_params : dev_params(h), sp, pool_view(h, h->l1), wK, pkd);
	HIPCHK(h, hipGetLastError());
	return 0;
}
/* Do all k-mers without an N of these reads weigh the same (no qualities, or one quality character throughout)?  Then wK is that
 * weight as the general kernel would form it -- (float) of the k-fold product of the character's probability, 1 for REF_QUAL --
 * and sk_extract_lean_kernel may take the launch. */
int sk_uniform_weight(kmr_handle *h, const ReadsView &rv, bool &lean, float &wK) {
	lean = false; wK = 1.0f;
	if (h->tune.no_lean_extract) return 0;
	if (!rv.quals && h->uniform_q_hint < 0) { lean = true; return 0; }
	if (h->qual_mixed || rv.n_reads == 0) return 0;
	if (h->uniform_q_hint >= 0) {      /* the caller said so (kmr_add_reads_twobit*): no pass over the quality bytes, no round trip */
		const unsigned int q0 = (unsigned int)h->uniform_q_hint;
		if (q0 == 127) { lean = true; wK = 1.0f; return 0; }
		if (!(h->hP[q0] > 0.0)) return 0;
		lean = true; wK = (float)h->hPk[q0];
		return 0;
	}
	if (!h->qrange) HIPCHK(h, dev_malloc((void **)&h->qrange, 8));
	const unsigned int init[2] = {255u, 0u};
	HIPCHK(h, hipMemcpyAsync(h->qrange, init, 8, hipMemcpyHostToDevice, h->stream));
	hipLaunchKernelGGL(sk_qual_range_kernel, dim3(num_cus(h) * 8), dim3(256), 0, h->stream, rv.quals, rv.offsets, rv.n_reads, h->qrange);
	HIPCHK(h, hipGetLastError());
	unsigned int got[2] = {0, 0};
	HIPCHK(h, hipMemcpyAsync(got, h->qrange, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	if (got[0] != got[1]) { h->qual_mixed = got[0] < got[1]; return 0; }      /* (255 > 0: no quality byte at all) */
	const unsigned int q0 = got[0];
	if (q0 == 127) { lean = true; wK = 1.0f; return 0; }                       /* Read::REF_QUAL */
	if (!(h->hP[q0] > 0.0)) return 0;                                          /* below the floor: the general kernel's zero handling */
	lean = true; wK = (float)h->hPk[q0];
	return 0;
}
uint32_t sk_dbg_flags(const char *name) {
#ifdef KMR_DEBUG_HOOKS
	const char *e = getenv(name); return e ? (uint32_t)atoi(e) : 0u;
#else
	(void)name; return 0u;
#endif
}
SkParams sk_params(kmr_handle *h) { SkParams sp; sp.dbg = sk_dbg_flags("KMR_SK_EXTRACT_DBG"); sp.keep_all_owners = h->sk_exchange ? 1u : 0u; sp.track = nullptr; sp.m = h->sk_m; sp.off = h->sk_off; sp.list_bits = h->sk_bits; sp.state = h->sk_state; sp.Pk = h->dPk; sp.Rp = h->dPk + 256; sp.fast_div = h->sk_fast_div ? 1u : 0u; return sp; }
template <int W> int add_reads_superkmer_t(kmr_handle *h, const ReadsView &rvAll, uint64_t total_bases) {
	const uint64_t n = rvAll.n_reads;
	const DevParams dp = dev_params(h);
	/* world_size > 1: without the exchange a rank keeps the k-mers the reference's owner function gives it (getDistributedThreadId,
	 * as the other build modes do); inside an exchange (kmr_sk_exchange_begin) every k-mer is kept, the lists decide the owner */
	const bool filt = dp.subsample > 1 || dp.num_parts > 1 || (dp.sub_wnb | dp.sub_snb) != 0 || (dp.world > 1 && !h->sk_exchange);
	bool lean = false; float wK = 1.0f;
	if (!filt && !h->ext && !sp_debug_extract(h)) { int rcq = sk_uniform_weight(h, rvAll, lean, wK); if (rcq) return rcq; }      /* (extension values want every neighbour's quality: the general kernel) */
	if (h->packed_direct && (!lean || h->cfg.size_tracker)) return fail(h, KMR_ERR_STATE, "internal: a packed batch handed to an extraction that wants text");      /* (sk_packed_direct_ok said otherwise) */
	if (rvAll.n_reads) {      /* one weight for the whole build so far? (the count pass's UNI form) */
		uint32_t wb; memcpy(&wb, &wK, 4);
		if (!lean) h->sk_uni_mixed = true;
		else if (h->sk_uni_w == SK_UNI_NONE) h->sk_uni_w = wb;
		else if (h->sk_uni_w != wb) h->sk_uni_mixed = true;
	}
	if (!h->sk_state) {
		/* lists: about 1100 k-mers each, as the final lists of the two-level partition */
		/* the list space is the whole job's (with world_size > 1 a rank owns every world_size-th list): sized from the caller's estimate
		 * of all the k-mers (estimateRawKmers), else from this call's bases times the ranks */
		const uint64_t est = h->cfg.estimated_raw_kmers ? h->cfg.estimated_raw_kmers : std::max<uint64_t>(total_bases, h->call_bases_hint) * std::max<uint32_t>(1, h->cfg.world_size);
		/* k-mers per list: the list's distinct keys have to fit the 1024-slot table of the count pass (80 %), or the list is redone in
		 * sub-passes.  At k <= 32 a list of ~1200 k-mers holds ~350 distinct ones in sequencing data; longer k-mers are hit by read
		 * errors more often (k = 51, 1 % errors: 40 % of the k-mers hold one and are nearly all distinct), so their lists are cut
		 * half as long (C4: count pass 133 -> 93 ms; another halving costs more in per-list work than it saves) */
		/* (extension values: a 512-slot table, COUNT_LOG2S_EXT) */
		/* (one-word keys: the count pass costs 8.8 ms per 10^9 k-mers of C2-like reads with lists of 1300-1600 k-mers, 9.25 at 1144 and at
		 * 1830, 10.2 at 2000 -- overflowing tables --, 11.2 at 715 and ~13 at 570 -- twice the per-list work: between two powers of two
		 * the longer lists win up to ~1900 k-mers; 1536 leaves the table room for inputs with more distinct k-mers than C2's 30 %.  Until
		 * round 3 the bound was 1224: a 12.5 M-read batch went to 2^21 lists of 715 and took 32.2 ms instead of 27.8) */
		const uint64_t per_list = h->ext ? (W > 1 ? 400 : 600) : ((h->tune.target_list == 2048 && W > 1) ? 700 : h->tune.target_list * 3 / 4);
		uint32_t bits = 6; while (bits < 24 && (est >> bits) > per_list) bits++;
		h->sk_fine_shift = 0;
		if (h->sk_exchange && h->cfg.world_size > 1 && !h->tune.no_coarse_lists) {
			uint32_t sh = 0; while ((1u << sh) < h->cfg.world_size) sh++;
			if (bits >= 6 + sh) { h->sk_fine_shift = sh; bits -= sh; }      /* the lists of the wire: as many and as full as one GPU's */
		}
		h->sk_bits = bits;
		/* one GPU, one-word keys, direction-counting values: the list count the count pass likes best instead of a power of two -- lists
		 * of ~1450 k-mers (8.8 ms per 10^9 k-mers; C2's 2^20 lists of 1144: 9.25).  The code of such a count is the count (sk_list_of). */
		if (h->cfg.world_size <= 1 && !h->sk_exchange && h->tune.target_list == 2048 && !h->tune.pow2_lists) {
			/* (two-word keys: lists just below their bound instead of anywhere between half of it and the bound -- C4 at 665 / 850 / 1100
			 * k-mers per list: 195.7 / 201 / 275 ms.  Extension values, 10 M reads at k = 21: 2^22 lists of 310 69.4 ms; 570 / 800 / 1000 /
			 * 1300 per list: 58.6 / 55.8 / 55.2 / 62.4 ms) */
			/* (est counts raw k-mers: reads with qualities of their own lose some of them to the weight floor -- the noisy C2 batch at 1300 /
			 * 1450 / 1600 / 1750 raw k-mers per list: count pass 12.7 / 11.4 / 11.1 / 11.0 ms; flat qualities at 1450 / 1550 / 1650: 10.6 /
			 * 10.8 / 10.8.  The first call's qualities decide) */
			const uint64_t aim = h->tune.list_aim ? h->tune.list_aim : (W == 1 ? (h->ext ? 800 : (lean ? 1450 : 1700)) : per_list * 19 / 20);
			const uint64_t nlists = (est / aim + 63) & ~63ull;
			if (nlists > 64 && (nlists < (1ull << bits) || h->tune.list_aim) && nlists < (1ull << 31)) h->sk_bits = (uint32_t)nlists;
		}
		const uint64_t nl0 = sk_list_count(h->sk_bits);
		HIPCHK(h, dev_malloc((void **)&h->sk_state, 8 * nl0));
		hipLaunchKernelGGL(sk_state_init_kernel, dim3(grid_for(nl0)), dim3(256), 0, h->stream, h->sk_state, nl0);
		HIPCHK(h, hipGetLastError());
	}
	const uint64_t avg = n ? std::max<uint64_t>(1, total_bases / n) : 1;
	const uint64_t sub_bases = h->tune.sub_batch_bases ? h->tune.sub_batch_bases : (1ull << 31);
	const uint64_t chunk = std::max<uint64_t>(64, (sub_bases / avg) & ~63ull);
	if (h->cfg.size_tracker && n) {      /* per-read records of this call (raw and good k-mers, end ordinal) */
		h->trk_n = 0;
		if (n > h->trk_cap) {
			if (h->trk) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree(h->trk); h->trk = nullptr; h->trk_cap = 0; }
			HIPCHK(h, dev_malloc((void **)&h->trk, n * sizeof(SkTrackRec))); h->trk_cap = n;
		}
		HIPCHK(h, hipMemsetAsync(h->trk, 0, n * sizeof(SkTrackRec), h->stream));
	}
	for (uint64_t r = 0; r < n; r += chunk) {
		const uint64_t m = std::min(chunk, n - r);
		ReadsView rv = rvAll;
		rv.offsets = rvAll.offsets + r; rv.n_reads = m;
		rv.discarded = rvAll.discarded ? rvAll.discarded + r : nullptr;
		rv.first_read_idx = rvAll.first_read_idx + r;
		int rc = prepare_units(h, rv, h->packed_direct ? SK_PACKED_SPAN : (uint32_t)TILE_SPAN); if (rc) return rc;
		/* room for this launch: a granule per k-mer is more than any input takes (flat qualities: a quarter of that), one open
		 * chunk per list and two slabs of 64 chunks per wavefront */
		const uint64_t bases = m * avg + avg;
		/* A job fed in many calls (estimated_raw_kmers says how much is to come) gets its pool in ONE allocation: this call's worst case
		 * plus what the rest of the job typically takes (flat qualities ~0.25 granules per base, a weight per k-mer ~0.55), at most half
		 * of the free memory.  Growing call by call frees the old pool every time, and an allocation right after tens of GB were freed
		 * waits seconds for the driver to clear them (config 3's whole input in eight calls: 5.1 s for the first build, HISTORY.md section 6);
		 * denser input than that still grows the pool as before */
		if (!h->l1.base && r == 0 && h->cfg.world_size <= 1 && !h->sk_exchange && (h->cfg.estimated_raw_kmers || h->call_bases_hint > total_bases)) {
			const double job_bases = h->cfg.estimated_raw_kmers ? (double)h->cfg.estimated_raw_kmers * (double)avg / (double)(avg > h->k ? avg - h->k + 1 : 1) : (double)h->call_bases_hint;
			const double rest = job_bases - (double)total_bases;
			size_t fr = 0, tot = 0;
			if (rest > 0 && hipMemGetInfo(&fr, &tot) == hipSuccess) {
				const uint64_t want = (uint64_t)(rest * (h->ext ? 2.0 : 1.0) * (lean ? 0.3 : 0.65) / SK_CHUNK_G);
				h->l1.presize = std::min<uint64_t>(want, (uint64_t)(fr / 4) / ((size_t)CH * rec_bytes(h)));      /* (an estimate: a quarter of what is free at most, kmr_finalize needs room of its own) */
			}
		}
		/* (inside an exchange the lists of other owners start afresh after every pack: an open chunk per list for every call) */
		rc = pool_reserve(h, h->l1, (h->ext ? 2 : 1) * bases / SK_CHUNK_G + ((h->l1.base && !h->sk_exchange) ? 0 : sk_list_count(h->sk_bits)) + (uint64_t)num_cus(h) * SK_EXTRACT_WAVES_PER_CU * 130 + 64, true); if (rc) return rc;
		hipEvent_t a, b, a2, b2; time_begin(h, KMR_TIME_BUILD, &a, &b); time_begin(h, KMR_TIME_EXTRACT, &a2, &b2);
		SkParams sp = sk_params(h);
		if (h->cfg.size_tracker) sp.track = h->trk + r;
#define SKX(WINv) (h->ext ? (filt ? launch_sk_extract<W, WINv, true, true>(h, rv, sp) : launch_sk_extract<W, WINv, false, true>(h, rv, sp)) : \
                   (filt ? launch_sk_extract<W, WINv, true>(h, rv, sp) : (lean ? launch_sk_extract_lean<W, WINv>(h, rv, sp, wK) : launch_sk_extract<W, WINv, false>(h, rv, sp))))
		if (h->packed_direct) {
			SkPacked pkd = *h->packed_direct; pkd.off += r; if (pkd.mk_off) pkd.mk_off += r;
			if (W > 1 && h->sk_win == 32) rc = launch_sk_extract_lean<W, (W > 1 ? 32 : 16), true>(h, rv, sp, wK, nullptr, &pkd);
			else rc = h->sk_win == 16 ? launch_sk_extract_lean<W, 16, true>(h, rv, sp, wK, nullptr, &pkd) : (h->sk_win == 8 ? launch_sk_extract_lean<W, 8, true>(h, rv, sp, wK, nullptr, &pkd) : launch_sk_extract_lean<W, 4, true>(h, rv, sp, wK, nullptr, &pkd));
		} else
		rc = (W > 1 && h->sk_win == 32) ? SKX((W > 1 ? 32 : 16)) : (h->sk_win == 16 ? SKX(16) : (h->sk_win == 8 ? SKX(8) : SKX(4)));
#undef SKX
		time_end(h, KMR_TIME_EXTRACT, a2, b2); time_end(h, KMR_TIME_BUILD, a, b);
		if (rc) return rc;
	}
	if (h->cfg.size_tracker && n) {
		/* SizeTracker::track (src/KmerSpectrum.h:879-894) is called before every k-mer: the thresholds this call's reads pass, each at
		 * the t-th raw k-mer of some read; the walk of those reads (sk_track_boundary_kernel) gives the ordinal and the good k-mers */
		std::vector<SkTrackRec> recs(n);
		HIPCHK(h, hipMemcpyAsync(recs.data(), h->trk, n * sizeof(SkTrackRec), hipMemcpyDeviceToHost, h->stream));
		HIPCHK(h, hipStreamSynchronize(h->stream));
		std::vector<SkBoundary> bd; std::vector<uint64_t> good_before;
		const bool walkable = (dp.sub_wnb | dp.sub_snb) == 0;      /* (a subtracting reference decides per k-mer what is raw: read ends there) */
		for (uint64_t r = 0; r < n; r++) {
			while ((uint64_t)h->trk_next <= h->trk_raw + recs[r].raw) {
				SkBoundary b; b.read = r; b.t = (uint32_t)((uint64_t)h->trk_next - h->trk_raw); b.good = 0; b.ordinal = 0;
				if (!walkable) { b.t = recs[r].raw; b.good = recs[r].good; b.ordinal = recs[r].end_ordinal; }
				bd.push_back(b); good_before.push_back(h->trk_good);
				h->trk_snap_raw.push_back(walkable ? (uint64_t)h->trk_next : h->trk_raw + recs[r].raw);
				h->trk_next = (long)((double)h->trk_next * 1.05);
				if (!walkable) break;      /* one element per read end */
			}
			if (!walkable) while ((uint64_t)h->trk_next <= h->trk_raw + recs[r].raw) h->trk_next = (long)((double)h->trk_next * 1.05);
			h->trk_raw += recs[r].raw; h->trk_good += recs[r].good;
		}
		if (h->trk_bounds.size() + bd.size() > SK_TRACK_MAX) return fail(h, KMR_ERR_UNSUPPORTED, "size tracker: more than 512 elements");
		if (!bd.empty() && walkable) {
			SkBoundary *dbd = nullptr;
			HIPCHK(h, dev_malloc((void **)&dbd, bd.size() * sizeof(SkBoundary)));
			hipError_t e = hipMemcpyAsync(dbd, bd.data(), bd.size() * sizeof(SkBoundary), hipMemcpyHostToDevice, h->stream);
			if (e == hipSuccess) {
				ReadsView rv = rvAll; rv.u_start = rv.u_end = rv.u_read = nullptr; rv.n_units = 0;
				hipLaunchKernelGGL(sk_track_boundary_kernel<W>, dim3((unsigned)((bd.size() + 63) / 64)), dim3(64), 0, h->stream, rv, dp, dbd, (uint32_t)bd.size());
				e = hipGetLastError();
			}
			if (e == hipSuccess) e = hipMemcpyAsync(bd.data(), dbd, bd.size() * sizeof(SkBoundary), hipMemcpyDeviceToHost, h->stream);
			if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
			hipFree(dbd);
			if (e != hipSuccess) return fail(h, KMR_ERR_HIP, std::string("size tracker boundaries: ") + hipGetErrorString(e));
		}
		for (size_t i = 0; i < bd.size(); i++) { h->trk_bounds.push_back(bd[i].ordinal); h->trk_snap_good.push_back(good_before[i] + bd[i].good); }
	}
	return 0;
}
int add_reads_superkmer(kmr_handle *h, const ReadsView &rv, uint64_t total_bases) {
	switch (h->W) { case 1: return add_reads_superkmer_t<1>(h, rv, total_bases); case 2: return add_reads_superkmer_t<2>(h, rv, total_bases);
	case 3: return add_reads_superkmer_t<3>(h, rv, total_bases); default: return add_reads_superkmer_t<4>(h, rv, total_bases); }
}
/* k-mers seen SK_ORDERED_FROM times or more (sat_*_kernel in kmr_superkmer.hpp): weightedCount and directionBias of their entries in the
 * finished weak map from their first 65 535 sightings in input order, added into a float one after the other as the serial reference
 * does.  n_clamped: how many such keys the count pass kept, n_sightings: their sightings. */
template <int W> int saturated_fix_t(kmr_handle *h, const uint64_t *ls, const uint64_t *lc, uint64_t nl, unsigned long long n_clamped, unsigned long long n_sightings, uint32_t has_singletons) {
	DevMap &wm = h->weak;
	if (!n_clamped || !wm.n) return 0;
	uint32_t list_bits = 0; while ((1ull << list_bits) < nl) list_bits++;
	if ((1ull << list_bits) != nl) list_bits = (uint32_t)nl;      /* a list count that is not a power of two is its own code (sk_list_of) */
	unsigned long long *dfound = nullptr; uint64_t *d_entry = nullptr; uint32_t *d_list = nullptr;
	std::vector<void *> owned;
	auto release = [&]() { for (void *p : owned) hipFree(p); owned.clear(); };
#define SATCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { release(); h->err = std::string(#call) + ": " + hip_err_text(e_); return e_ == hipErrorOutOfMemory ? KMR_ERR_OOM : KMR_ERR_HIP; } } while (0)
	auto dalloc = [&](void **p, size_t bytes) -> hipError_t { hipError_t e = dev_malloc(p, std::max<size_t>(bytes, 256)); if (e == hipSuccess) owned.push_back(*p); return e; };
	SATCHK(dalloc((void **)&dfound, 8));
	uint64_t cap = std::min<uint64_t>(wm.n, 4 * n_clamped + 1024);
	unsigned long long found = 0;
	for (;;) {
		SATCHK(dalloc((void **)&d_entry, 8 * cap)); SATCHK(dalloc((void **)&d_list, 4 * cap));
		SATCHK(hipMemsetAsync(dfound, 0, 8, h->stream));
		hipLaunchKernelGGL(sat_find_kernel<W>, dim3(grid_for(wm.n)), dim3(256), 0, h->stream, (const uint64_t *)wm.keys, (const uint32_t *)wm.vals, wm.n, h->sk_m, h->sk_off, h->sk_win, list_bits, dfound, cap, d_entry, d_list, h->ext ? 15u : 3u);
		SATCHK(hipGetLastError());
		SATCHK(hipMemcpyAsync(&found, dfound, 8, hipMemcpyDeviceToHost, h->stream)); SATCHK(hipStreamSynchronize(h->stream));
		if (found <= cap) break;
		cap = found;      /* more entries at exactly 65 535 than expected: again with room for all */
	}
	if (found >= (1ull << 23)) { release(); return 0; }      /* (more than 8 x 10^6 saturated keys: left as the count pass made them) */
	std::vector<uint64_t> entry(found); std::vector<uint32_t> lst(found);
	SATCHK(hipMemcpy(entry.data(), d_entry, 8 * found, hipMemcpyDeviceToHost)); SATCHK(hipMemcpy(lst.data(), d_list, 4 * found, hipMemcpyDeviceToHost));
	std::vector<uint32_t> order(found);
	for (uint32_t i = 0; i < found; i++) order[i] = i;
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lst[a] != lst[b] ? lst[a] < lst[b] : entry[a] < entry[b]; });
	std::vector<uint64_t> sentry(found); std::vector<uint32_t> dl; std::vector<uint64_t> le0, le1;
	for (uint32_t i = 0; i < found; i++) {
		sentry[i] = entry[order[i]];
		const uint32_t l = lst[order[i]];
		if (dl.empty() || dl.back() != l) { dl.push_back(l); le0.push_back(i); le1.push_back(i + 1); } else le1.back() = i + 1;
	}
	SATCHK(hipMemcpy(d_entry, sentry.data(), 8 * found, hipMemcpyHostToDevice));
	SATCHK(hipMemcpy(d_list, dl.data(), 4 * dl.size(), hipMemcpyHostToDevice));
	uint64_t *d_c0 = nullptr, *d_c1 = nullptr;
	SATCHK(dalloc((void **)&d_c0, 8 * dl.size())); SATCHK(dalloc((void **)&d_c1, 8 * dl.size()));
	hipLaunchKernelGGL(sat_gather_kernel, dim3(grid_for(dl.size())), dim3(256), 0, h->stream, ls, (const uint32_t *)d_list, (uint64_t)dl.size(), d_c0, d_c1);
	SATCHK(hipGetLastError());
	std::vector<uint64_t> c0(dl.size()), c1(dl.size());
	SATCHK(hipMemcpy(c0.data(), d_c0, 8 * dl.size(), hipMemcpyDeviceToHost)); SATCHK(hipMemcpy(c1.data(), d_c1, 8 * dl.size(), hipMemcpyDeviceToHost));
	/* work items: pieces of 256 chunks of those lists */
	std::vector<uint64_t> ic0, ic1, ie0, ie1;
	for (size_t i = 0; i < dl.size(); i++)
		for (uint64_t a = c0[i]; a < c1[i]; a += 256) { ic0.push_back(a); ic1.push_back(std::min(c1[i], a + 256)); ie0.push_back(le0[i]); ie1.push_back(le1[i]); }
	if (ic0.empty()) { release(); return 0; }
	uint64_t *d_items = nullptr;
	const size_t ni = ic0.size();
	SATCHK(dalloc((void **)&d_items, 32 * ni));
	SATCHK(hipMemcpy(d_items, ic0.data(), 8 * ni, hipMemcpyHostToDevice)); SATCHK(hipMemcpy(d_items + ni, ic1.data(), 8 * ni, hipMemcpyHostToDevice));
	SATCHK(hipMemcpy(d_items + 2 * ni, ie0.data(), 8 * ni, hipMemcpyHostToDevice)); SATCHK(hipMemcpy(d_items + 3 * ni, ie1.data(), 8 * ni, hipMemcpyHostToDevice));
	/* every sighting of those keys (the count pass has added up their true counts; entries of a map merged in later add theirs) */
	const uint64_t pcap = n_sightings + (found - std::min<unsigned long long>(found, n_clamped)) * 65535ull + 64;
	unsigned long long *pk_in = nullptr, *pk_out = nullptr; uint32_t *pv_in = nullptr, *pv_out = nullptr;
	SATCHK(dalloc((void **)&pk_in, 8 * pcap)); SATCHK(dalloc((void **)&pk_out, 8 * pcap)); SATCHK(dalloc((void **)&pv_in, 4 * pcap)); SATCHK(dalloc((void **)&pv_out, 4 * pcap));
	SATCHK(hipMemsetAsync(dfound, 0, 8, h->stream));
	int rc = zero_work_counter(h); if (rc) { release(); return rc; }
	hipLaunchKernelGGL(sat_collect_kernel<W>, dim3((unsigned)std::min<uint64_t>(ni, (uint64_t)num_cus(h) * 4)), dim3(256), 0, h->stream, pool_view(h, h->l1), lc, h->k, (const uint64_t *)wm.keys, (const uint64_t *)d_entry,
	                   (const uint64_t *)d_items, (const uint64_t *)(d_items + ni), (const uint64_t *)(d_items + 2 * ni), (const uint64_t *)(d_items + 3 * ni), (uint64_t)ni, dfound, pcap, pk_in, pv_in, h->work_counter);
	SATCHK(hipGetLastError());
	unsigned long long n_pairs = 0;
	SATCHK(hipMemcpyAsync(&n_pairs, dfound, 8, hipMemcpyDeviceToHost, h->stream)); SATCHK(hipStreamSynchronize(h->stream));
	if (n_pairs > pcap) { release(); return fail(h, KMR_ERR_CAPACITY, "sightings of saturated k-mers (internal sizing error)"); }
	size_t tmp_bytes = 0; void *tmp = nullptr;
	if (kmr::sort_pairs_u64_u32(nullptr, &tmp_bytes, pk_in, pk_out, pv_in, pv_out, n_pairs, h->stream) != 0) { release(); return fail(h, KMR_ERR_HIP, "radix sort (size query)"); }
	SATCHK(dalloc(&tmp, tmp_bytes));
	if (kmr::sort_pairs_u64_u32(tmp, &tmp_bytes, pk_in, pk_out, pv_in, pv_out, n_pairs, h->stream) != 0) { release(); return fail(h, KMR_ERR_HIP, "radix sort"); }
	hipLaunchKernelGGL(sat_reduce_kernel, dim3((unsigned)std::min<uint64_t>(found, 4096)), dim3(256), 0, h->stream, (const unsigned long long *)pk_out, (const uint32_t *)pv_out, (uint64_t)n_pairs, (const uint64_t *)d_entry, (uint64_t)found, has_singletons, wm.vals, h->ext ? 15u : 3u);
	SATCHK(hipGetLastError());
	SATCHK(hipStreamSynchronize(h->stream));
	release();
#undef SATCHK
	return 0;
}

template <int W> int finalize_superkmer_t(kmr_handle *h, uint32_t min_depth) {
	int rc = sync_state(h);
	if (rc) return rc;
	hipEvent_t ea, eb; time_begin(h, 1, &ea, &eb);
	const uint64_t G = h->stats.raw_good_kmers;
	FinalizeParams f; f.kb = h->hkb; f.ext_min_q = h->cfg.ext_min_quality; f.min_depth = min_depth; f.has_singletons = h->cfg.separate_singletons ? 1 : 0; f.nb_weak = h->nb_weak; f.nb_sing = h->nb_sing; f.uni_wbits = 0;
	const bool keepSing = f.has_singletons && min_depth <= 1;
	if (!h->l1.head) { rc = pool_reserve(h, h->l1, 0, false); if (rc) return rc; }
	rc = arena_reset(h); if (rc) return rc;
	uint64_t nl = h->sk_state ? sk_list_count(h->sk_bits) : 1;
	if (h->sk_state) {
		hipLaunchKernelGGL(sk_close_kernel, dim3(grid_for(nl)), dim3(256), 0, h->stream, h->sk_state, nl, h->l1.chunk_count, h->l1.cap);
		HIPCHK(h, hipGetLastError());
	}
	const bool refined = h->sk_state && h->sk_fine_shift > 0;
	if (refined) {
		/* the coarse lists this rank owns (its own share and what it adopted) -> fine lists (sk_refine_kernel) */
		const uint32_t fine_bits = h->sk_bits + h->sk_fine_shift;
		const uint64_t nlf = 1ull << fine_bits;
		unsigned int head = 0;
		HIPCHK(h, hipStreamSynchronize(h->stream));
		HIPCHK(h, hipMemcpy(&head, h->l1.head, 4, hipMemcpyDeviceToHost));
		if (head > h->l1.cap) head = h->l1.cap;
		if (h->sk_fine_cap < nlf) {
			if (h->sk_fine_state) hipFree(h->sk_fine_state);
			h->sk_fine_state = nullptr; h->sk_fine_cap = 0;
			HIPCHK(h, dev_malloc((void **)&h->sk_fine_state, 8 * nlf)); h->sk_fine_cap = nlf;
		}
		hipLaunchKernelGGL(sk_state_init_kernel, dim3(grid_for(nlf)), dim3(256), 0, h->stream, h->sk_fine_state, nlf);
		const int rgrid = (int)std::min<uint64_t>(((uint64_t)head + SK_REFINE_WAVES - 1) / SK_REFINE_WAVES + 1, (uint64_t)num_cus(h) * 8);
		/* every old chunk's records again, cut into at most 2^shift pieces per chunk (a piece may open a chunk), an open chunk per owned fine list */
		rc = pool_reserve(h, h->l1, (uint64_t)head * 2 + nlf / h->cfg.world_size + (uint64_t)rgrid * SK_REFINE_WAVES * 130 + 64, true); if (rc) return rc;
		if (head) hipLaunchKernelGGL(sk_refine_kernel, dim3(rgrid), dim3(SK_REFINE_WAVES * 64), 0, h->stream, pool_view(h, h->l1), head, fine_bits, h->sk_fine_state);
		hipLaunchKernelGGL(sk_close_kernel, dim3(grid_for(nlf)), dim3(256), 0, h->stream, h->sk_fine_state, nlf, h->l1.chunk_count, h->l1.cap);
		HIPCHK(h, hipGetLastError());
		nl = nlf;
	}
	uint64_t *ls = nullptr, *lc = nullptr; uint32_t nch = 0;
	rc = build_csr(h, h->l1, nl, 0, &ls, &lc, &nch); if (rc) return rc;
	const uint32_t vw = h->ext ? 15 : 3;
	const bool ext = h->ext;      /* extension values: entries of 15 value words, keys and values apart, bucketed by the scatter + per-bucket sort */
	const uint64_t slack = (uint64_t)num_cus(h) * 4 * 8192 + 16;
	const uint64_t wbound = f.has_singletons ? G / 2 : G, sbound = keepSing ? G : 0;
	/* (after an owner exchange the lists this rank counts hold other ranks' k-mers too -- G only knows this rank's own reads: no upper
	 * bound then, the pass is repeated with doubled buffers until the entries fit) */
	const bool adopted = h->sk_exchange && h->cfg.world_size > 1;
	const uint64_t wmax = adopted ? (1ull << 40) : wbound + wbound / 8 + slack, smax = keepSing ? (adopted ? (1ull << 40) : sbound + sbound / 8 + slack) : 16;
	/* entry buffers: sequencing data keeps a few per cent of its k-mers as weak entries; the pass is run again with larger
	 * buffers when that was not enough */
	uint64_t wcap = std::min<uint64_t>(wmax, G / (f.has_singletons ? 8 : 3) + slack), scap = keepSing ? std::min<uint64_t>(smax, G / 3 + slack) : 16;
	if (h->tune.entry_share >= 0) { wcap = std::min<uint64_t>(wmax, (uint64_t)((double)G * h->tune.entry_share) + 16384); if (keepSing) scap = std::min<uint64_t>(smax, (uint64_t)((double)G * h->tune.entry_share) + 16384);
		if (h->ue) { hipFree(h->ue); h->ue = nullptr; h->ue_cap = 0; }
		if (h->us_keys) { hipFree(h->us_keys); hipFree(h->us_b8); if (h->us_pkt) hipFree(h->us_pkt); h->us_keys = h->us_b8 = h->us_pkt = nullptr; h->us_cap = 0; } }
	if (!ext && h->ue && h->ue_cap >= wcap) wcap = h->ue_cap;
	if (ext && h->uw_keys && h->uw_cap >= wcap) wcap = h->uw_cap;
	if (h->us_keys && h->us_cap >= scap) scap = h->us_cap;
	uint32_t *wc = nullptr, *sc = nullptr; FinalizeCounters *fc = nullptr; unsigned long long *cursors = nullptr;
	rc = arena_get(h, &wc, h->nb_weak); if (rc) return rc; rc = arena_get(h, &sc, h->nb_sing); if (rc) return rc;
	rc = arena_get(h, &fc, 1); if (rc) return rc; rc = arena_get(h, &cursors, 2); if (rc) return rc;
	FinalizeCounters c; unsigned long long cur[2];
	/* size tracker: SizeTracker::track (src/KmerSpectrum.h:879-894) applied after every read, in stream order; what it pushes is known
	 * from the per-read records alone except the unique / singleton counters, which the count pass fills in per boundary */
	std::vector<unsigned long long> bounds; std::vector<uint64_t> snap_raw, snap_good;
	SkTrackView tv; tv.bounds = nullptr; tv.n = 0; tv.d_unique = tv.d_single = nullptr;
	const bool tracking = h->cfg.size_tracker != 0;
	/* one weight for every record of the lists (own calls: host state; adopted records: the device pair)?  Then the count pass's UNI form */
	bool uni = false;
	f.uni_wbits = 0;
	if (!tracking && !ext && !h->sk_uni_mixed && !h->tune.no_uniform_count && (W == 1 || (h->k & 31u) != 0)) {      /* (multi-word keys: the one-weight pass has no state words, it needs pad bits in the last key word) */
		uint32_t w = h->sk_uni_w; bool mixed = false;
		if (h->peer_uni_mixed) mixed = true;
		else if (h->peer_uni_w != SK_UNI_NONE) { if (w == SK_UNI_NONE) w = h->peer_uni_w; else if (w != h->peer_uni_w) mixed = true; }
		if (h->d_uni) {
			uint32_t dv[2] = {SK_UNI_NONE, 0u};
			HIPCHK(h, hipMemcpy(dv, h->d_uni, 8, hipMemcpyDeviceToHost));
			if (dv[1]) mixed = true;
			else if (dv[0] != SK_UNI_NONE) { if (w == SK_UNI_NONE) w = dv[0]; else if (w != dv[0]) mixed = true; }
		}
		if (!mixed && w != SK_UNI_NONE) { uni = true; f.uni_wbits = w; }
	}
	h->last_count_uniform = uni;
	if (tracking) {
		bounds = h->trk_bounds; snap_raw = h->trk_snap_raw; snap_good = h->trk_snap_good;
		if (bounds.size() > SK_TRACK_MAX) return fail(h, KMR_ERR_UNSUPPORTED, "size tracker: more than 512 elements");
		unsigned long long *db = nullptr; unsigned int *dd = nullptr;
		rc = arena_get(h, &db, bounds.size() + 1); if (rc) return rc;
		rc = arena_get(h, &dd, 2 * (bounds.size() + 1)); if (rc) return rc;
		if (!bounds.empty()) HIPCHK(h, hipMemcpyAsync(db, bounds.data(), 8 * bounds.size(), hipMemcpyHostToDevice, h->stream));
		tv.bounds = db; tv.n = (uint32_t)bounds.size(); tv.d_unique = dd; tv.d_single = dd + bounds.size() + 1;
	}
	/* long lists (SkLong in kmr_superkmer.hpp): found from the CSR, cut into work items, counted by a second launch into a merge table */
	SkLong<W> lgMain; lgMain.item_c0 = lgMain.item_c1 = nullptr; lgMain.n_items = 0; lgMain.long_threshold = 0; lgMain.merge.slots = nullptr; lgMain.merge.ext = nullptr; lgMain.merge.log2cap = 0; lgMain.merge_used = nullptr;
	lgMain.list_first = 0; lgMain.list_stride = 1;
	if (h->sk_exchange && h->cfg.world_size > 1 && !refined) { lgMain.list_first = h->cfg.rank; lgMain.list_stride = h->cfg.world_size; }      /* the other lists went to their owners */
	/* lists below early.hi were counted by kmr_count_lists_prefix: this pass starts behind them and their entries are taken over below.
	 * An early count that overflowed its buffers, or was made for another min-depth or map layout, is void: everything is counted here */
	uint64_t early_slots = 0; FinalizeCounters early_c; memset(&early_c, 0, sizeof(early_c));
	h->last_early_hi = 0; h->last_early_entries = 0;
	if (h->early.active) {
		uint32_t cerr0 = 0;
		HIPCHK(h, hipMemcpy(&cerr0, h->derr, 4, hipMemcpyDeviceToHost));
		unsigned long long ecur = 0;
		HIPCHK(h, hipMemcpy(&ecur, h->early.cursor, 8, hipMemcpyDeviceToHost)); HIPCHK(h, hipMemcpy(&early_c, h->early.fc, sizeof(early_c), hipMemcpyDeviceToHost));
		const bool ok = !(cerr0 & ERR_ENTRIES_FULL) && ecur <= h->early.cap && h->early.min_depth == min_depth && !tracking && !ext && !keepSing && !refined;
		if (cerr0 & ERR_ENTRIES_FULL) { cerr0 &= ~(uint32_t)ERR_ENTRIES_FULL; HIPCHK(h, hipMemcpy(h->derr, &cerr0, 4, hipMemcpyHostToDevice)); }
		if (ok) {
			early_slots = ecur;
			h->last_early_hi = h->early.hi; h->last_early_entries = early_c.weak_kept;
			const uint64_t stride0 = lgMain.list_stride, first0 = lgMain.list_first;
			uint64_t first = h->early.hi;
			if (stride0 > 1) first += (first0 + stride0 - first % stride0) % stride0;      /* the first list at or behind hi that is this rank's */
			lgMain.list_first = first;
		} else { h->early.active = false; memset(&early_c, 0, sizeof(early_c)); }
	}
	SkLong<W> lgItems = lgMain;
	uint64_t n_items = 0, long_chunks = 0;
	/* (extension values: the 16-bit tallies of a block's table are exact below 65 536 k-mers, SK_EXT_LONG_CHUNKS) */
	const uint64_t LONG_CHUNKS = ext ? std::min<uint64_t>(h->tune.long_list_chunks ? h->tune.long_list_chunks : SK_EXT_LONG_CHUNKS, SK_EXT_LONG_CHUNKS) : (h->tune.long_list_chunks ? h->tune.long_list_chunks : 1024), PIECE = std::max<uint64_t>(1, LONG_CHUNKS / 2);
	if (!tracking && nch > LONG_CHUNKS) {
		const uint64_t cap = (uint64_t)nch / PIECE + 2 * 1024 + 16;
		uint64_t *ic0 = nullptr, *ic1 = nullptr; unsigned long long *dn = nullptr;
		rc = arena_get(h, &ic0, cap); if (rc) return rc; rc = arena_get(h, &ic1, cap); if (rc) return rc; rc = arena_get(h, &dn, 1); if (rc) return rc;
		HIPCHK(h, hipMemsetAsync(dn, 0, 8, h->stream));
		hipLaunchKernelGGL(sk_long_items_kernel, dim3(grid_for(nl)), dim3(256), 0, h->stream, ls, nl, LONG_CHUNKS, PIECE, ic0, ic1, cap, dn);
		HIPCHK(h, hipGetLastError());
		unsigned long long hn = 0;
		HIPCHK(h, hipMemcpyAsync(&hn, dn, 8, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream));
		if (hn > cap) return fail(h, KMR_ERR_CAPACITY, "long-list work items (internal sizing error)");
		n_items = hn; long_chunks = n_items * PIECE;
		if (n_items) {
			lgMain.long_threshold = LONG_CHUNKS;
			lgItems.item_c0 = ic0; lgItems.item_c1 = ic1; lgItems.n_items = n_items; lgItems.merge_used = dn;
		}
	}
	uint32_t merge_log2 = 16;
	hipEvent_t tca, tcb; time_begin(h, KMR_TIME_COUNT, &tca, &tcb);
	for (int attempt = 0; ; attempt++) {
		if (!ext && (!h->ue || h->ue_cap < wcap)) {
			if (h->ue) hipFree(h->ue); h->ue = nullptr; h->ue_cap = 0;
			HIPCHK(h, dev_malloc((void **)&h->ue, 8ull * (W + 1) * wcap)); h->ue_cap = wcap;
		}
		if (ext && (!h->uw_keys || h->uw_cap < wcap)) {
			if (h->uw_keys) hipFree(h->uw_keys); if (h->uw_vals) hipFree(h->uw_vals); h->uw_keys = h->uw_vals = nullptr; h->uw_cap = 0;
			HIPCHK(h, dev_malloc(&h->uw_keys, 8ull * W * wcap)); HIPCHK(h, dev_malloc(&h->uw_vals, 4ull * vw * wcap)); h->uw_cap = wcap;
		}
		if (!h->us_keys || h->us_cap < scap || (ext && !h->us_pkt)) {
			if (h->us_keys) hipFree(h->us_keys); if (h->us_b8) hipFree(h->us_b8); if (h->us_pkt) hipFree(h->us_pkt); h->us_keys = h->us_b8 = h->us_pkt = nullptr; h->us_cap = 0;
			HIPCHK(h, dev_malloc(&h->us_keys, 8ull * W * scap)); HIPCHK(h, dev_malloc(&h->us_b8, scap)); if (ext) HIPCHK(h, dev_malloc(&h->us_pkt, 4 * scap)); h->us_cap = scap;
		}
		HIPCHK(h, hipMemsetAsync(wc, 0, 4 * h->nb_weak, h->stream)); HIPCHK(h, hipMemsetAsync(sc, 0, 4 * h->nb_sing, h->stream));
		HIPCHK(h, hipMemsetAsync(fc, 0, sizeof(FinalizeCounters), h->stream)); HIPCHK(h, hipMemsetAsync(cursors, 0, 16, h->stream));
		CountOut out; out.wkeys = nullptr; out.wvals = nullptr; out.wentries = h->ue; out.wcursor = cursors; out.wcap = h->ue_cap;
		out.skeys = (uint64_t *)h->us_keys; out.sweight = (uint8_t *)h->us_b8; out.spkt = nullptr; out.scursor = cursors + 1; out.scap = h->us_cap;
		out.weakCount = nullptr; out.singCount = sc; out.fc = fc; out.err = h->derr;      /* weak entries are bucketed without a per-bucket histogram (kmr_buckets.hpp) */
		if (ext) { out.wkeys = (uint64_t *)h->uw_keys; out.wvals = (uint32_t *)h->uw_vals; out.wentries = nullptr; out.wcap = h->uw_cap; out.spkt = (uint32_t *)h->us_pkt; out.weakCount = wc; }
		rc = zero_work_counter(h); if (rc) return rc;
		const int grid = (int)std::min<uint64_t>((uint64_t)num_cus(h) * 4, (nl / lgMain.list_stride + SK_LBATCH) / SK_LBATCH);
		auto kern = ext ? sk_count_kernel<W, COUNT_LOG2S_EXT, false, true> : (tracking ? sk_count_kernel<W, COUNT_LOG2S, true> : (uni ? sk_count_kernel<W, COUNT_LOG2S, false, false, true> : sk_count_kernel<W, COUNT_LOG2S, false>));
		const size_t smem = ext ? sk_count_smem_bytes<W, COUNT_LOG2S_EXT, false, true>() : (tracking ? sk_count_smem_bytes<W, COUNT_LOG2S, true>() : (uni ? sk_count_smem_bytes<W, COUNT_LOG2S, false, false, true>() : sk_count_smem_bytes<W, COUNT_LOG2S, false>()));
		if (tracking) HIPCHK(h, hipMemsetAsync(tv.d_unique, 0, 8 * (tv.n + 1), h->stream));
		HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
		if (dbg()) { int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, SKC_THREADS, smem); fprintf(stderr, "sk_count<W=%d>: %d blocks per CU (LDS %zu), %llu lists, %u chunks\n", W, nb, smem, (unsigned long long)nl, nch); }
		hipLaunchKernelGGL(kern, dim3(grid), dim3(SKC_THREADS), smem, h->stream, pool_view(h, h->l1), ls, lc, nl, h->k, out, f, h->work_counter, sk_dbg_flags("KMR_SK_COUNT_DBG"), tv, lgMain);
		HIPCHK(h, hipGetLastError());
		Slot<W> *mslots = nullptr; ExtSlot *mext = nullptr;
		if (n_items) {
			/* the merge table holds the distinct keys of the long lists: few when a list is long because a k-mer repeats, at most the
			 * k-mers of those lists; it starts small and the attempt is repeated with a larger one if it fills */
			if (dev_malloc((void **)&mslots, sizeof(Slot<W>) << merge_log2) != hipSuccess) return fail(h, KMR_ERR_OOM, "merge table of the long lists");
			if (ext && dev_malloc((void **)&mext, sizeof(ExtSlot) << merge_log2) != hipSuccess) { hipFree(mslots); return fail(h, KMR_ERR_OOM, "merge table of the long lists"); }
			hipLaunchKernelGGL(table_clear_kernel<W>, dim3(grid_for(1ull << merge_log2)), dim3(256), 0, h->stream, mslots, mext, 1ull << merge_log2);
			lgItems.merge.slots = mslots; lgItems.merge.ext = mext; lgItems.merge.log2cap = merge_log2;
			HIPCHK(h, hipMemsetAsync(lgItems.merge_used, 0, 8, h->stream));
			rc = zero_work_counter(h); if (rc) { hipFree(mslots); if (mext) hipFree(mext); return rc; }
			const int grid2 = (int)std::min<uint64_t>((uint64_t)num_cus(h) * 4, n_items);
			hipLaunchKernelGGL(kern, dim3(grid2), dim3(SKC_THREADS), smem, h->stream, pool_view(h, h->l1), ls, lc, nl, h->k, out, f, h->work_counter, sk_dbg_flags("KMR_SK_COUNT_DBG"), tv, lgItems);
			hipLaunchKernelGGL(sk_merge_emit_kernel<W>, dim3(grid_for(1ull << merge_log2)), dim3(256), 0, h->stream, lgItems.merge, out, f);
			if (hipGetLastError() != hipSuccess) { hipFree(mslots); if (mext) hipFree(mext); return fail(h, KMR_ERR_HIP, "long-list launches"); }
		}
		uint32_t cerr = 0;
		HIPCHK(h, hipMemcpyAsync(&c, fc, sizeof(c), hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipMemcpyAsync(cur, cursors, 16, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(h, hipMemcpyAsync(&cerr, h->derr, 4, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(h, hipStreamSynchronize(h->stream));
		if (mslots) hipFree(mslots);
		if (mext) hipFree(mext);
		if (n_items && (cerr & ERR_TABLE_FULL) && merge_log2 < 30) {      /* the merge table filled: again with a larger one */
			cerr &= ~(uint32_t)(ERR_TABLE_FULL | ERR_ENTRIES_FULL);
			HIPCHK(h, hipMemcpy(h->derr, &cerr, 4, hipMemcpyHostToDevice));
			merge_log2 += 3;
			continue;
		}
		if (!(cerr & ERR_ENTRIES_FULL)) break;
		if ((wcap >= wmax && scap >= smax) || attempt >= 8) { time_end(h, KMR_TIME_COUNT, tca, tcb); time_end(h, 1, ea, eb); return fail(h, KMR_ERR_CAPACITY, "entry buffers of the count pass overflowed at their upper bound (internal sizing error)"); }
		cerr &= ~(uint32_t)ERR_ENTRIES_FULL;
		HIPCHK(h, hipMemcpy(h->derr, &cerr, 4, hipMemcpyHostToDevice));
		wcap = std::min<uint64_t>(wmax, wcap * 2); if (keepSing) scap = std::min<uint64_t>(smax, scap * 2);
		if (dbg()) fprintf(stderr, "sk count pass: entry buffers too small, retrying with %llu / %llu\n", (unsigned long long)wcap, (unsigned long long)scap);
	}
	time_end(h, KMR_TIME_COUNT, tca, tcb);
	h->stats.unique_kmers = c.unique;
	h->stats.singleton_kmers = f.has_singletons ? c.singletons : 0;
	if (tracking) {
		std::vector<unsigned int> dd(2 * (tv.n + 1), 0);
		HIPCHK(h, hipMemcpy(dd.data(), tv.d_unique, 8 * (tv.n + 1), hipMemcpyDeviceToHost));
		h->trk_elems.clear();
		uint64_t uniq = 0; int64_t single = 0;
		const uint64_t sub = h->cfg.kmer_subsample > 1 ? h->cfg.kmer_subsample : 1;      /* track() scales what it stores, :882-887 */
		for (uint32_t i = 0; i < tv.n; i++) {
			uniq += dd[i]; single += (int32_t)dd[tv.n + 1 + i];
			h->trk_elems.push_back(snap_raw[i] * sub); h->trk_elems.push_back(snap_good[i] * sub);
			h->trk_elems.push_back(uniq * sub); h->trk_elems.push_back(f.has_singletons ? (uint64_t)single * sub : 0);
		}
	}
	if (early_slots) {      /* the early count's entries behind this pass's (the slabs' unused tails are holes in both) */
		const size_t eb = 8ull * (W + 1);
		if (cur[0] + early_slots > h->ue_cap) {
			uint64_t *bigger = nullptr; const uint64_t ncap = cur[0] + early_slots + 4096;
			HIPCHK(h, dev_malloc((void **)&bigger, eb * ncap));
			HIPCHK(h, hipMemcpyAsync(bigger, h->ue, eb * cur[0], hipMemcpyDeviceToDevice, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream));
			hipFree(h->ue); h->ue = bigger; h->ue_cap = ncap;
		}
		HIPCHK(h, hipMemcpyAsync((uint8_t *)h->ue + eb * cur[0], h->early.ue, eb * early_slots, hipMemcpyDeviceToDevice, h->stream));
		cur[0] += early_slots;
		c.unique += early_c.unique; c.singletons += early_c.singletons; c.weak_kept += early_c.weak_kept; c.sing_kept += early_c.sing_kept;
		c.saturated += early_c.saturated; c.sat_sightings += early_c.sat_sightings;
		h->stats.unique_kmers = c.unique; h->stats.singleton_kmers = f.has_singletons ? c.singletons : 0;
	}
	h->early.active = false;
	hipEvent_t tma, tmb; time_begin(h, KMR_TIME_BUCKETS, &tma, &tmb);
	rc = finish_maps_from_entries(h, wc, sc, cur[0], cur[1], c.weak_kept, c.sing_kept, keepSing, !ext);
	time_end(h, KMR_TIME_BUCKETS, tma, tmb);
	if (rc) return rc;
	if (c.saturated) { rc = saturated_fix_t<W>(h, ls, lc, nl, c.saturated, c.sat_sightings, f.has_singletons); if (rc) return rc; }
	time_end(h, 1, ea, eb);
	h->has_singletons = keepSing;
	h->stats.weak_entries = h->weak.n; h->stats.singleton_entries = keepSing ? h->sing.n : 0;
	h->finalized = true; h->map_gen++;
	rc = sync_state(h);
	if (!rc && !h->arena_overflow.empty()) rc = arena_reset(h);
	return rc;
}
/* kmr_count_lists_prefix: the count pass over this handle's lists below `hi`, now, into entry buffers of their own -- the lower part of
 * the list space is counted while the upper part is still on the wire; kmr_finalize counts what is left and takes these entries
 * over.  Direction-counting values without a singleton map in the result (min_depth >= 2 or no separate singletons), fine lists only;
 * anything else, or buffers that turn out too small, leaves the lists to kmr_finalize. */
template <int W> int count_prefix_superkmer_t(kmr_handle *h, uint32_t min_depth, uint64_t hi) {
	int rc = sync_state(h); if (rc) return rc;
	FinalizeParams f; f.kb = h->hkb; f.ext_min_q = h->cfg.ext_min_quality; f.min_depth = min_depth; f.has_singletons = h->cfg.separate_singletons ? 1 : 0; f.nb_weak = h->nb_weak; f.nb_sing = h->nb_sing; f.uni_wbits = 0;
	const bool keepSing = f.has_singletons && min_depth <= 1;
	h->early.active = false;      /* an earlier early count is replaced */
	if (h->ext || h->cfg.size_tracker || keepSing || (h->sk_state && h->sk_fine_shift > 0) || !h->sk_state) return KMR_OK;      /* nothing counted early: kmr_finalize does it all */
	const uint64_t nl = sk_list_count(h->sk_bits);
	if (hi > nl) hi = nl;
	if (hi == 0) return KMR_OK;
	rc = arena_reset(h); if (rc) return rc;
	hipLaunchKernelGGL(sk_close_kernel, dim3(grid_for(nl)), dim3(256), 0, h->stream, h->sk_state, nl, h->l1.chunk_count, h->l1.cap);
	HIPCHK(h, hipGetLastError());
	uint64_t *ls = nullptr, *lc = nullptr; uint32_t nch = 0;
	rc = build_csr(h, h->l1, nl, 0, &ls, &lc, &nch); if (rc) return rc;
	/* room: the share of the good k-mers that lies below hi, at the rate kmr_finalize starts with; too little is found out by
	 * kmr_finalize (ERR_ENTRIES_FULL), which then counts everything itself */
	const uint64_t G = h->stats.raw_good_kmers;      /* (an owner's lists hold about as many k-mers as its own reads gave: the job's share of one rank) */
	const uint64_t slack = (uint64_t)num_cus(h) * 4 * 8192 + 16;
	const uint64_t want = (uint64_t)((double)G * ((double)hi / (double)nl) / (f.has_singletons ? 6.0 : 2.5)) + slack;
	if (!h->early.ue || h->early.cap < want) {
		if (h->early.ue) hipFree(h->early.ue); h->early.ue = nullptr; h->early.cap = 0;
		HIPCHK(h, dev_malloc((void **)&h->early.ue, 8ull * (W + 1) * want)); h->early.cap = want;
	}
	if (!h->early.cursor) HIPCHK(h, dev_malloc((void **)&h->early.cursor, 16));
	if (!h->early.fc) HIPCHK(h, dev_malloc((void **)&h->early.fc, sizeof(FinalizeCounters)));
	HIPCHK(h, hipMemsetAsync(h->early.cursor, 0, 16, h->stream)); HIPCHK(h, hipMemsetAsync(h->early.fc, 0, sizeof(FinalizeCounters), h->stream));
	bool uni = false;
	if (!h->sk_uni_mixed && !h->tune.no_uniform_count && !h->peer_uni_mixed && (W == 1 || (h->k & 31u) != 0)) {
		uint32_t w = h->sk_uni_w; bool mixed = false;
		if (h->peer_uni_w != SK_UNI_NONE) { if (w == SK_UNI_NONE) w = h->peer_uni_w; else if (w != h->peer_uni_w) mixed = true; }
		if (h->d_uni) {
			uint32_t dv[2] = {SK_UNI_NONE, 0u};
			HIPCHK(h, hipMemcpy(dv, h->d_uni, 8, hipMemcpyDeviceToHost));
			if (dv[1]) mixed = true;
			else if (dv[0] != SK_UNI_NONE) { if (w == SK_UNI_NONE) w = dv[0]; else if (w != dv[0]) mixed = true; }
		}
		if (!mixed && w != SK_UNI_NONE) { uni = true; f.uni_wbits = w; }
	}
	SkTrackView tv; tv.bounds = nullptr; tv.n = 0; tv.d_unique = tv.d_single = nullptr;
	SkLong<W> lg; lg.item_c0 = lg.item_c1 = nullptr; lg.n_items = 0; lg.merge.slots = nullptr; lg.merge.ext = nullptr; lg.merge.log2cap = 0; lg.merge_used = nullptr;
	lg.list_first = 0; lg.list_stride = 1;
	if (h->sk_exchange && h->cfg.world_size > 1) { lg.list_first = h->cfg.rank; lg.list_stride = h->cfg.world_size; }
	/* (lists too long for one block are left to kmr_finalize's pass over work items, which covers the whole list space) */
	lg.long_threshold = h->tune.long_list_chunks ? h->tune.long_list_chunks : 1024;
	uint32_t *sc = nullptr;
	rc = arena_get(h, &sc, h->nb_sing); if (rc) return rc;
	HIPCHK(h, hipMemsetAsync(sc, 0, 4 * h->nb_sing, h->stream));
	CountOut out; out.wkeys = nullptr; out.wvals = nullptr; out.wentries = h->early.ue; out.wcursor = h->early.cursor; out.wcap = h->early.cap;
	out.skeys = nullptr; out.sweight = nullptr; out.spkt = nullptr; out.scursor = h->early.cursor + 1; out.scap = 0;
	out.weakCount = nullptr; out.singCount = sc; out.fc = (FinalizeCounters *)h->early.fc; out.err = h->derr;
	rc = zero_work_counter(h); if (rc) return rc;
	const uint64_t n_work = hi > lg.list_first ? (hi - lg.list_first + lg.list_stride - 1) / lg.list_stride : 0;
	if (n_work) {
		const int grid = (int)std::min<uint64_t>((uint64_t)num_cus(h) * 4, (n_work + SK_LBATCH) / SK_LBATCH);
		auto kern = uni ? sk_count_kernel<W, COUNT_LOG2S, false, false, true> : sk_count_kernel<W, COUNT_LOG2S, false>;
		const size_t smem = uni ? sk_count_smem_bytes<W, COUNT_LOG2S, false, false, true>() : sk_count_smem_bytes<W, COUNT_LOG2S, false>();
		HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
		hipEvent_t a, b; time_begin(h, KMR_TIME_COUNT, &a, &b);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(SKC_THREADS), smem, h->stream, pool_view(h, h->l1), ls, lc, hi, h->k, out, f, h->work_counter, sk_dbg_flags("KMR_SK_COUNT_DBG"), tv, lg);
		time_end(h, KMR_TIME_COUNT, a, b);
		HIPCHK(h, hipGetLastError());
	}
	h->early.active = true; h->early.hi = hi; h->early.min_depth = min_depth;
	return KMR_OK;      /* asynchronous: the handle's stream carries the pass; kmr_finalize (or kmr_sync) waits for it */
}
int count_prefix_superkmer(kmr_handle *h, uint32_t min_depth, uint64_t hi) {
	switch (h->W) { case 1: return count_prefix_superkmer_t<1>(h, min_depth, hi); case 2: return count_prefix_superkmer_t<2>(h, min_depth, hi);
	case 3: return count_prefix_superkmer_t<3>(h, min_depth, hi); default: return count_prefix_superkmer_t<4>(h, min_depth, hi); }
}

int finalize_superkmer(kmr_handle *h, uint32_t min_depth) {
	switch (h->W) { case 1: return finalize_superkmer_t<1>(h, min_depth); case 2: return finalize_superkmer_t<2>(h, min_depth);
	case 3: return finalize_superkmer_t<3>(h, min_depth); default: return finalize_superkmer_t<4>(h, min_depth); }
}

void free_partition_state(kmr_handle *h) {
	if (h->sk_state) hipFree(h->sk_state); h->sk_state = nullptr;
	pool_free(h->l1);
	if (h->l1_state) hipFree(h->l1_state); h->l1_state = nullptr; h->l1_state_bytes = 0; h->l1_state_dirty = false;
	for (void *p : h->arena_overflow) hipFree(p);
	h->arena_overflow.clear();
	if (h->arena) hipFree(h->arena); h->arena = nullptr; h->arena_cap = h->arena_used = h->arena_want = 0;
	if (h->work_counter) hipFree(h->work_counter); if (h->linear) hipFree(h->linear); if (h->tile_count) hipFree(h->tile_count);
	if (h->kcap) hipFree(h->kcap); if (h->koff) hipFree(h->koff);
	if (h->ucnt) hipFree(h->ucnt); if (h->ufirst) hipFree(h->ufirst); if (h->u_start) { hipFree(h->u_start); hipFree(h->u_end); hipFree(h->u_read); } if (h->umax) hipFree(h->umax);
	for (int a = 0; a < 2; a++) for (int b = 0; b < 8; b++) { if (h->tb_stage[a][b]) hipFree(h->tb_stage[a][b]); h->tb_stage[a][b] = nullptr; h->tb_stage_cap[a][b] = 0; }
	if (h->tb_copy_stream) { hipStreamDestroy(h->tb_copy_stream); h->tb_copy_stream = nullptr; for (int a = 0; a < 2; a++) { hipEventDestroy(h->tb_ready[a]); hipEventDestroy(h->tb_consumed[a]); h->tb_set_used[a] = false; } }
	if (h->tb_bases) hipFree(h->tb_bases); if (h->tb_quals) hipFree(h->tb_quals); if (h->tb_rel) hipFree(h->tb_rel); if (h->tb_off) hipFree(h->tb_off); if (h->tb_len) hipFree(h->tb_len);
	h->tb_bases = h->tb_quals = nullptr; h->tb_rel = h->tb_off = nullptr; h->tb_len = nullptr; h->tb_bases_cap = h->tb_quals_cap = h->tb_quals_filled = h->tb_n = 0; h->tb_quals_char = -1;
	h->ucnt = nullptr; h->ufirst = nullptr; h->u_start = h->u_end = h->u_read = nullptr; h->umax = nullptr; h->ucnt_n = h->ufirst_n = h->units_n = 0;
	if (h->uw_keys) hipFree(h->uw_keys); if (h->uw_vals) hipFree(h->uw_vals); if (h->us_keys) hipFree(h->us_keys); if (h->us_b8) hipFree(h->us_b8); if (h->us_pkt) hipFree(h->us_pkt);
	if (h->ue) hipFree(h->ue); if (h->ue2) hipFree(h->ue2); h->ue = h->ue2 = nullptr; h->ue_cap = h->ue2_cap = 0;
	h->us_pkt = nullptr; h->work_counter = nullptr; h->linear = nullptr; h->tile_count = nullptr; h->kcap = nullptr; h->koff = nullptr;
	h->uw_keys = h->uw_vals = h->us_keys = h->us_b8 = nullptr; h->linear_cap = h->tile_cap = h->kcap_n = h->koff_n = h->uw_cap = h->us_cap = 0;
}

}  // namespace

/* ====================================================================== */
extern "C" {

uint32_t kmr_abi_version(void) { return KMR_ABI_VERSION; }

int kmr_config_init(kmr_config *c) {
	if (!c) return KMR_ERR_INVALID_ARG;
	memset(c, 0, sizeof(*c));
	c->struct_size = sizeof(*c);
	c->value_kind = KMR_VALUE_COUNT_DIR; c->min_weight = 0.10f; c->min_quality_score = 3; c->fastq_start_char = 33;
	c->ext_min_quality = 20; c->separate_singletons = 1; c->kmer_subsample = 1; c->device = -1; c->rank = 0; c->world_size = 1;
	c->estimated_depth = 20.0; c->estimated_error_rate = 0.35; c->kmers_per_bucket = 32; c->num_parts = 1; c->part_idx = 0;
	return KMR_OK;
}

const char *kmr_last_error(const kmr_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int kmr_create(const kmr_config *cfg, kmr_handle **out) {
	if (!cfg || !out) return fail(nullptr, KMR_ERR_INVALID_ARG, "null argument");
	*out = nullptr;
	if (cfg->struct_size != sizeof(kmr_config)) return fail(nullptr, KMR_ERR_INVALID_ARG, "kmr_config.struct_size mismatch (ABI)");
	if (cfg->k < 1 || cfg->k > 128) return fail(nullptr, KMR_ERR_INVALID_ARG, "k must be in 1..128");
	if (cfg->world_size < 1 || cfg->rank >= cfg->world_size) return fail(nullptr, KMR_ERR_INVALID_ARG, "bad rank/world_size");
	if (cfg->value_kind > KMR_VALUE_EXT) return fail(nullptr, KMR_ERR_INVALID_ARG, "bad value_kind");
	if (cfg->hash_kind > KMR_HASH_LOOKUP8) return fail(nullptr, KMR_ERR_INVALID_ARG, "bad hash_kind");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, KMR_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
	kmr_handle *h = new kmr_handle();
	h->cfg = *cfg;
	if (h->cfg.kmer_subsample == 0) h->cfg.kmer_subsample = 1;
	if (h->cfg.num_parts == 0) h->cfg.num_parts = 1;
	h->k = cfg->k; h->kb = (cfg->k + 3) / 4; h->W = (h->kb + 7) / 8; h->ext = cfg->value_kind == KMR_VALUE_EXT;
	h->hkb = h->kb | (cfg->hash_kind << 16);      /* what the kernels' hash sees: key bytes and the hash kind (kmr_key.hpp, key_hash) */
	memset(&h->stats, 0, sizeof(h->stats));
	int rc = 0;
	do {
		if (cfg->device >= 0) { if (hipSetDevice(cfg->device) != hipSuccess) { rc = fail(nullptr, KMR_ERR_NO_DEVICE, "hipSetDevice failed"); break; } }
		if (hipGetDevice(&h->device) != hipSuccess) { rc = fail(nullptr, KMR_ERR_NO_DEVICE, "hipGetDevice failed"); break; }
		if (hipStreamCreate(&h->stream) != hipSuccess) { rc = fail(nullptr, KMR_ERR_HIP, "hipStreamCreate failed"); break; }
		/* bucket sizing of the KmerSpectrum ctor (src/KmerSpectrum.h:414-416, src/Kmer.h:2837) */
		uint64_t w = cfg->num_buckets_weak, s = cfg->num_buckets_singleton;
		const double depth = cfg->estimated_depth > 0 ? cfg->estimated_depth : 20.0;
		const uint32_t kpb = cfg->kmers_per_bucket ? cfg->kmers_per_bucket : 32;
		/* estimated_raw_kmers is the whole job's; a rank's maps are built for its share, the figure
		 * DistributedKmerSpectrum::estimateRawKmers (src/DistributedFunctions.h:144-162) hands the constructor */
		const uint64_t raw = cfg->estimated_raw_kmers / (cfg->world_size > 1 ? cfg->world_size : 1);
		if (w == 0) { unsigned long est = (unsigned long)(int)(raw / depth); w = est / kpb + 1; }
		if (s == 0) { unsigned long est = cfg->separate_singletons ? (unsigned long)(raw * cfg->estimated_error_rate) : 1; s = est / kpb + 1; }
		h->nb_weak = resize_buckets(w); h->nb_sing = resize_buckets(s);
		h->has_singletons = cfg->separate_singletons != 0;
		/* table capacity */
		uint64_t want = cfg->max_table_entries ? (uint64_t)(cfg->max_table_entries / 0.7) : (uint64_t)(cfg->estimated_raw_kmers * 0.45 / 0.6);
		if (cfg->world_size > 1 && !cfg->max_table_entries) want /= cfg->world_size;
		uint32_t lg = 16; while ((1ull << lg) < want && lg < 40) lg++;
		h->log2cap = lg;
		double P[256]; quality_table(P, cfg->min_quality_score, cfg->fastq_start_char);
		if (dev_malloc((void **)&h->dP, sizeof(P)) != hipSuccess || dev_malloc((void **)&h->dstats, sizeof(DevStats)) != hipSuccess || dev_malloc((void **)&h->derr, 4) != hipSuccess) { rc = fail(nullptr, KMR_ERR_OOM, "hipMalloc failed"); break; }
		hipMemcpy(h->dP, P, sizeof(P), hipMemcpyHostToDevice); hipMemset(h->dstats, 0, sizeof(DevStats)); hipMemset(h->derr, 0, 4);
		/* build_mode: 0 auto (streaming partition path unless EXT values), 1 table, 2 partition */
		if (cfg->build_mode > 3) { rc = fail(nullptr, KMR_ERR_INVALID_ARG, "bad build_mode"); break; }
		h->partition_mode = cfg->build_mode != 1;
		/* auto: super-k-mer lists where they apply (count / direction values, one partition, k >= 13), else the two-level k-mer partition */
		uint32_t wish_w = 0, wish_m = 0, wish_o = 0;
		const bool sk_auto = cfg->build_mode == 0 && cfg->world_size <= 1 && sk_geometry(h->k, 0, wish_w, wish_m, wish_o);
		h->auto_mode = cfg->build_mode == 0;
		/* (an auto handle of a multi-rank job starts on the k-mer partition -- plain kmr_add_reads* there means the getDistributedThreadId
		 * filter -- but is made ready for the lists: kmr_exchange_init moves it over) */
		/* (and any handle whose k has a minimizer geometry can answer lookups as a streaming pass over the same lists) */
		const bool sk_ready = sk_geometry(h->k, 0, wish_w, wish_m, wish_o);
		if (cfg->build_mode == 3 || sk_auto || sk_ready) {
			if (!sk_geometry(h->k, 0, h->sk_win, h->sk_m, h->sk_off)) { rc = fail(nullptr, KMR_ERR_UNSUPPORTED, "build_mode 3 (super-k-mer lists) needs k >= 13"); break; }
			h->superkmer_mode = cfg->build_mode == 3 || sk_auto;
			double Pk[256];
			for (int cidx = 0; cidx < 256; cidx++) { double wv = 1.0; for (uint32_t jj = 0; jj < h->k; jj++) wv *= P[cidx]; Pk[cidx] = wv; }      /* the loop of buildWeightedKmers, src/KmerReadUtils.h:205-208 */
			memcpy(h->hPk, Pk, sizeof(Pk)); memcpy(h->hP, P, sizeof(h->hP));
			if (dev_malloc((void **)&h->dPk, 2 * sizeof(Pk)) != hipSuccess) { rc = fail(nullptr, KMR_ERR_OOM, "hipMalloc failed"); break; }
			hipMemcpy(h->dPk, Pk, sizeof(Pk), hipMemcpyHostToDevice);
			/* reciprocals for the chain's divide, usable only if multiply-and-correct reproduces the correctly rounded quotient of every
			 * pair of table entries (all 256 x 256 are tried; the kernel divides otherwise) */
			double Rp[256]; bool fast = true;
			for (int cidx = 0; cidx < 256; cidx++) Rp[cidx] = P[cidx] != 0.0 ? 1.0 / P[cidx] : 0.0;
			for (int a = 0; a < 256 && fast; a++) for (int b = 0; b < 256; b++) {
				if (P[a] == 0.0 || P[b] == 0.0) continue;
				const double q0 = P[a] * Rp[b];
				if (std::fma(std::fma(-q0, P[b], P[a]), Rp[b], q0) != P[a] / P[b]) { fast = false; break; }
			}
			h->sk_fast_div = fast;
			hipMemcpy(h->dPk + 256, Rp, sizeof(Rp), hipMemcpyHostToDevice);
		}
		if (cfg->size_tracker && (!h->superkmer_mode || h->ext || cfg->world_size > 1)) { rc = fail(nullptr, KMR_ERR_UNSUPPORTED, "size_tracker: kept by the super-k-mer build (build_mode 0 / 3, direction-counting values, k >= 13) of a single partition"); break; }
		if (!h->partition_mode) {
			rc = alloc_table(h, h->log2cap, &h->slots, &h->extslots);
			if (rc) { g_create_error = h->err; break; }
		}
		if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = fail(nullptr, KMR_ERR_HIP, std::string("table clear failed: ") + hipGetErrorString(hipGetLastError())); break; }
	} while (0);
	if (rc) { kmr_destroy(h); return rc; }
	*out = h;
	return KMR_OK;
}

static void exchange_free(kmr_handle *h);
void kmr_destroy(kmr_handle *h) {
	if (!h) return;
	if (h->stream) hipStreamSynchronize(h->stream);
	for (int which = 0; which < KMR_TIME_GROUPS; which++) for (auto &pr : h->pending_events[which]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
	if (h->slots) hipFree(h->slots); if (h->extslots) hipFree(h->extslots);
	if (h->qrange) hipFree(h->qrange);
	if (h->dP) hipFree(h->dP); if (h->dPk) hipFree(h->dPk); if (h->dstats) hipFree(h->dstats); if (h->derr) hipFree(h->derr);
	free_map(h->weak); free_map(h->sing);
	free_partition_state(h);
	if (h->scan_sums) hipFree(h->scan_sums);
	if (h->score_buf) hipFree(h->score_buf);
	if (h->lut) hipFree(h->lut);
	if (h->xo_dev) hipFree(h->xo_dev);
	if (h->trk) hipFree(h->trk);
	if (h->adopt_buf) hipFree(h->adopt_buf);
	if (h->sk_fine_state) hipFree(h->sk_fine_state);
	if (h->d_uni) hipFree(h->d_uni);
	if (h->early.ue) hipFree(h->early.ue); if (h->early.cursor) hipFree(h->early.cursor); if (h->early.fc) hipFree(h->early.fc);
	if (h->ix_start) hipFree(h->ix_start); if (h->ix_keys) hipFree(h->ix_keys); if (h->ix_counts) hipFree(h->ix_counts); if (h->scratch_stats) hipFree(h->scratch_stats);
	exchange_free(h);
	if (h->stream) hipStreamDestroy(h->stream);
	delete h;
}

int kmr_num_buckets(const kmr_handle *h, int which, uint64_t *out) {
	if (!h || !out) return KMR_ERR_INVALID_ARG;
	if (which == KMR_MAP_WEAK) *out = h->finalized && h->weak.present ? h->weak.nb : h->nb_weak;
	else if (which == KMR_MAP_SINGLETON) *out = h->finalized && h->sing.present ? h->sing.nb : h->nb_sing;
	else return KMR_ERR_UNSUPPORTED;
	return KMR_OK;
}

/* KmerSpectrum::buildKmerSpectrum starts with weak.reset(false); singleton.reset(false)
 * (src/KmerSpectrum.h:2091-2096): empty maps, allocations kept. */
int kmr_reset(kmr_handle *h) {
	if (!h) return KMR_ERR_INVALID_ARG;
	hipSetDevice(h->device);
	HIPCHK(h, hipStreamSynchronize(h->stream));
	clear_map(h->weak); clear_map(h->sing);      /* empty maps, allocations kept (reset(false)) */
	int rc = 0;
	if (h->partition_mode) {
		if (h->l1.head) HIPCHK(h, hipMemsetAsync(h->l1.head, 0, 4, h->stream));
		h->l1.used_ub = 0;
		h->inserted_records = 0;
		h->qual_mixed = false;
		h->sk_uni_w = SK_UNI_NONE; h->sk_uni_mixed = false; h->peer_uni_w = SK_UNI_NONE; h->peer_uni_mixed = false;
		h->xr_lo = 0; h->xr_hi = ~0ull; h->early.active = false;
		if (h->d_uni) { const uint32_t init[2] = {SK_UNI_NONE, 0u}; HIPCHK(h, hipMemcpyAsync(h->d_uni, init, 8, hipMemcpyHostToDevice, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream)); }
		if (h->sk_state) hipLaunchKernelGGL(sk_state_init_kernel, dim3(grid_for(sk_list_count(h->sk_bits))), dim3(256), 0, h->stream, h->sk_state, sk_list_count(h->sk_bits));
		if (h->l1_state) {      /* what an unfinished build kept back is dropped with its pool */
			hipLaunchKernelGGL(partition_state_init_kernel, dim3(partition_blocks(h)), dim3(256), 0, h->stream, h->l1_state,
			                   h->l1_state_bytes / (size_t)partition_blocks(h), h->bits1, (uint32_t)partition_blocks(h));
			h->l1_state_dirty = false;
		}
	} else {
		if (!h->slots) rc = alloc_table(h, h->log2cap, &h->slots, &h->extslots);
		else rc = clear_table_any(h, h->slots, h->extslots, h->log2cap);
		if (rc) return rc;
	}
	HIPCHK(h, hipMemsetAsync(h->dstats, 0, sizeof(DevStats), h->stream));
	HIPCHK(h, hipMemsetAsync(h->derr, 0, 4, h->stream));
	memset(&h->stats, 0, sizeof(h->stats));
	h->occupied = h->pending_kmers = 0; h->stream_base = 0; h->reads = 0; h->subtracted = 0; h->xc_job_bases = 0; h->xc_bytes_to_peers = 0;
	h->trk_n = 0; h->trk_elems.clear();
	h->trk_next = 128; h->trk_raw = h->trk_good = 0; h->trk_bounds.clear(); h->trk_snap_raw.clear(); h->trk_snap_good.clear();
	h->finalized = false; h->map_gen++; h->has_singletons = h->cfg.separate_singletons != 0;
	return KMR_OK;
}
int kmr_release_table(kmr_handle *h) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_release_table before kmr_finalize");
	hipSetDevice(h->device);
	if (h->slots) { hipFree(h->slots); h->slots = nullptr; }
	if (h->extslots) { hipFree(h->extslots); h->extslots = nullptr; }
	if (h->partition_mode) free_partition_state(h);
	return KMR_OK;
}

void *kmr_stream(kmr_handle *h) { return h ? (void *)h->stream : nullptr; }

/* the stream ordinal the next kmr_add_reads* starts from (see the header) */
int kmr_set_stream_origin(kmr_handle *h, uint64_t ordinal) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (h->finalized) return fail(h, KMR_ERR_STATE, "kmr_set_stream_origin after kmr_finalize");
	if (ordinal > MAX_STREAM_ORDINAL) return fail(h, KMR_ERR_CAPACITY, "stream ordinals have 40 bits");
	h->stream_base = ordinal;
	return KMR_OK;
}

/* knobs of one handle (see struct Tuning); set before the first kmr_add_reads* of a build */
int kmr_tune(kmr_handle *h, const char *knob, double value) {
	if (!h || !knob) return KMR_ERR_INVALID_ARG;
	const std::string k(knob);
	if (k == "target_list_records") h->tune.target_list = value >= 1 ? (uint64_t)value : 2048;
	else if (k == "sub_batch_bases") h->tune.sub_batch_bases = value >= 1 ? (uint64_t)value : 0;
	else if (k == "recycle_chunks") h->tune.recycle = value < 0 ? -1 : (value != 0 ? 1 : 0);
	else if (k == "partition_blocks") h->tune.part_blocks = value >= 1 ? (int)value : 0;
	else if (k == "entry_share") h->tune.entry_share = value;
	else if (k == "lookup_table") h->tune.no_lut = value == 0;
	else if (k == "stream_lookups") h->tune.no_stream_lookups = value == 0;
	else if (k == "long_list_chunks") h->tune.long_list_chunks = value < 2 ? 2 : (uint64_t)value;
	else if (k == "lean_extract") h->tune.no_lean_extract = value == 0;
	else if (k == "uniform_count") h->tune.no_uniform_count = value == 0;
	else if (k == "packed_direct") h->tune.no_packed_direct = value == 0;
	else if (k == "pow2_lists") h->tune.pow2_lists = value != 0;
	else if (k == "list_aim") h->tune.list_aim = value >= 1 ? (uint64_t)value : 0;
	else if (k == "twobit_piece_bases") h->tune.twobit_piece_bases = (uint64_t)value;
	else if (k == "exchange_fail_once") h->tune.exchange_fail_once = value != 0;
	else if (k == "binned_buckets_min") h->tune.binned_min = value >= 0 ? (uint64_t)value : ~0ull;        /* < 0: never */
	else if (k == "coarse_lists") h->tune.no_coarse_lists = value == 0;
	else if (k == "narrow_tallies") h->tune.no_narrow = value == 0;
	else if (k == "keep_level1_state") h->tune.no_l1_state = value == 0;
	else if (k == "superkmer_window") {      /* largest minimizer window the geometry may take (32 / 16 / 8 / 4): A/B runs, tests of the narrower windows at large k */
		if (h->superkmer_mode && !h->sk_state) { uint32_t w, m, o; if (!sk_geometry(h->k, 0, w, m, o, (uint32_t)value)) return fail(h, KMR_ERR_INVALID_ARG, "no minimizer geometry under that window"); h->sk_win = w; h->sk_m = m; h->sk_off = o; }
	}
	else if (k == "superkmer_minimizer") {
		if (h->superkmer_mode && !h->sk_state) { uint32_t w, m, o; if (!sk_geometry(h->k, (uint32_t)value, w, m, o)) return fail(h, KMR_ERR_INVALID_ARG, "no minimizer geometry for that length"); h->sk_win = w; h->sk_m = m; h->sk_off = o; }
	}
	else return fail(h, KMR_ERR_INVALID_ARG, "unknown tuning knob '" + k + "'");
	return KMR_OK;
}

int kmr_build_info(kmr_handle *h, const char *what, double *value) {
	if (!h || !what || !value) return KMR_ERR_INVALID_ARG;
	const std::string k(what);
	if (k == "lists") *value = h->sk_state ? (double)sk_list_count(h->sk_bits) : 0.0;
	else if (k == "uniform_count") *value = h->last_count_uniform ? 1.0 : 0.0;
	else if (k == "chunk_pool_chunks") *value = (double)h->l1.cap;
	else if (k == "superkmer_window") *value = (double)h->sk_win;
	else if (k == "early_lists") *value = (double)h->last_early_hi;
	else if (k == "early_entries") *value = (double)h->last_early_entries;
	else return fail(h, KMR_ERR_INVALID_ARG, "unknown build figure '" + k + "'");
	return KMR_OK;
}

int kmr_sync(kmr_handle *h) { if (!h) return KMR_ERR_INVALID_ARG; hipSetDevice(h->device); return sync_state(h); }

int kmr_add_reads_dev(kmr_handle *h, const void *dev_bases, const void *dev_quals, const void *dev_offsets, uint64_t n_reads,
                      uint64_t total_bases, uint64_t first_global_read_idx, const void *dev_discarded) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (h->finalized) return fail(h, KMR_ERR_STATE, "kmr_add_reads after kmr_finalize");
	if (n_reads == 0) return KMR_OK;
	if (!dev_bases || !dev_offsets) return fail(h, KMR_ERR_INVALID_ARG, "null device buffer");
	hipSetDevice(h->device);
	ReadsView rv; rv.bases = (const uint8_t *)dev_bases; rv.quals = (const uint8_t *)dev_quals; rv.offsets = (const uint64_t *)dev_offsets;
	rv.discarded = (const uint8_t *)dev_discarded; rv.n_reads = n_reads; rv.stream_base = h->stream_base; rv.first_read_idx = first_global_read_idx;
	rv.u_start = rv.u_end = rv.u_read = nullptr; rv.n_units = 0;
	/* the stream ordinal of an occurrence decides which sighting of a k-mer was its first (directionBias, the quantised first
	 * weight): 40 bits in the table slots, and the 16-byte records of build_mode 2 without extension values carry 32 of them */
	if (h->stream_base + total_bases > MAX_STREAM_ORDINAL) return fail(h, KMR_ERR_CAPACITY, "more than 2^40 input bases on one handle");
	if (h->partition_mode && !h->superkmer_mode && !h->ext && h->stream_base + total_bases > (1ull << 32))
		return fail(h, KMR_ERR_CAPACITY, "build_mode 2 orders occurrences by a 32-bit stream ordinal: at most 2^32 input bases per handle without extension values (use build_mode 0 / 3 or 1)");
	int rc = h->superkmer_mode ? add_reads_superkmer(h, rv, total_bases) : (h->partition_mode ? add_reads_partition(h, rv, total_bases) : add_reads_dev_any(h, rv, total_bases));
	h->stream_base += total_bases; h->reads += n_reads; h->stats.reads = h->reads;
	return rc;
}

/* one piece of a host batch onto the device: staging set `set` of the handle (grow-only), the copy on the handle's copy stream */
static hipError_t tb_stage_copy(kmr_handle *h, int set, int which, const void *src, size_t bytes, void **dst) {
	uint8_t *&buf = h->tb_stage[set][which]; size_t &cap = h->tb_stage_cap[set][which];
	if (cap < bytes + 64) {
		if (buf) { hipError_t e0 = hipStreamSynchronize(h->stream); if (e0 != hipSuccess) return e0; hipFree(buf); buf = nullptr; cap = 0; }
		hipError_t e = dev_malloc((void **)&buf, bytes + bytes / 8 + 4096); if (e != hipSuccess) return e;
		cap = bytes + bytes / 8 + 4096 - 64;
	}
	*dst = buf;
	return bytes ? hipMemcpyAsync(buf, src, bytes, hipMemcpyHostToDevice, h->tb_copy_stream) : hipSuccess;
}
static int tb_pipeline_ready(kmr_handle *h) {
	if (h->tb_copy_stream) return 0;
	HIPCHK(h, hipStreamCreateWithFlags(&h->tb_copy_stream, hipStreamNonBlocking));
	for (int i = 0; i < 2; i++) { HIPCHK(h, hipEventCreateWithFlags(&h->tb_ready[i], hipEventDisableTiming)); HIPCHK(h, hipEventCreateWithFlags(&h->tb_consumed[i], hipEventDisableTiming)); }
	return 0;
}

int kmr_add_reads(kmr_handle *h, const char *bases, const char *quals, const uint64_t *offsets, uint64_t n_reads,
                  uint64_t first_global_read_idx, const uint8_t *discarded) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (h->finalized) return fail(h, KMR_ERR_STATE, "kmr_add_reads after kmr_finalize");
	if (n_reads == 0) return KMR_OK;
	if (!bases || !offsets) return fail(h, KMR_ERR_INVALID_ARG, "null buffer");
	hipSetDevice(h->device);
	/* pieces of about 2^28 bases through two sets of staging buffers and the handle's copy stream: piece i + 1 is on the bus while the
	 * device extracts piece i; the offsets go over as they are (the device arrays are addressed through pointers moved back by the
	 * piece's first offset) */
	{ int rcp = tb_pipeline_ready(h); if (rcp) return rcp; }
#define TBCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { hipStreamSynchronize(h->tb_copy_stream); hipStreamSynchronize(h->stream); h->err = std::string(#call) + ": " + hip_err_text(e_); return e_ == hipErrorOutOfMemory ? KMR_ERR_OOM : KMR_ERR_HIP; } } while (0)
	const uint64_t piece_bases = h->tune.twobit_piece_bases ? h->tune.twobit_piece_bases : (1ull << 28);
	int rc = KMR_OK; int set = 0;
	h->call_bases_hint = n_reads ? offsets[n_reads] - offsets[0] : 0;      /* what the first piece sizes is the whole call's */
	struct HintReset { kmr_handle *h; ~HintReset() { h->call_bases_hint = 0; } } hint_reset{h};
	for (uint64_t r0 = 0; r0 < n_reads && rc == KMR_OK; set ^= 1) {
		uint64_t r1 = r0 + 1;
		while (r1 < n_reads && offsets[r1 + 1] - offsets[r0] <= piece_bases) r1++;
		const uint64_t m = r1 - r0, total = offsets[r1] - offsets[r0];
		void *db = nullptr, *dq = nullptr, *doff = nullptr, *dd = nullptr;
		if (h->tb_set_used[set]) TBCHK(hipStreamWaitEvent(h->tb_copy_stream, h->tb_consumed[set], 0));
		TBCHK(tb_stage_copy(h, set, 0, bases + offsets[r0], total, &db)); TBCHK(tb_stage_copy(h, set, 2, offsets + r0, 8 * (m + 1), &doff));
		if (quals) TBCHK(tb_stage_copy(h, set, 6, quals + offsets[r0], total, &dq));
		if (discarded) TBCHK(tb_stage_copy(h, set, 7, discarded + r0, m, &dd));
		TBCHK(hipMemsetAsync((uint8_t *)db + total, 0, 64, h->tb_copy_stream)); if (dq) TBCHK(hipMemsetAsync((uint8_t *)dq + total, 0, 64, h->tb_copy_stream));
		TBCHK(hipEventRecord(h->tb_ready[set], h->tb_copy_stream));
		TBCHK(hipStreamWaitEvent(h->stream, h->tb_ready[set], 0));
		rc = kmr_add_reads_dev(h, (uint8_t *)db - offsets[r0], dq ? (uint8_t *)dq - offsets[r0] : nullptr, doff, m, total, first_global_read_idx + r0, dd);
		TBCHK(hipEventRecord(h->tb_consumed[set], h->stream)); h->tb_set_used[set] = true;
		r0 = r1;
	}
#undef TBCHK
	hipStreamSynchronize(h->tb_copy_stream);
	if (!rc) rc = sync_state(h);
	else hipStreamSynchronize(h->stream);
	return rc;
}

/* May sk_extract_lean_kernel<.., PACKED> take a packed batch as it is?  Whenever add_reads_superkmer_t would hand the unpacked
 * batch to the lean extraction: lists, direction-counting values, nothing that filters k-mers, and one weight for every k-mer. */
static bool sk_packed_direct_ok(kmr_handle *h, bool has_quals, int uniform_quality) {
	if (!h->superkmer_mode || h->ext || h->cfg.size_tracker || h->tune.no_lean_extract || h->tune.no_packed_direct || sp_debug_extract(h) || has_quals || h->qual_mixed) return false;
	const DevParams dp = dev_params(h);
	if (dp.subsample > 1 || dp.num_parts > 1 || (dp.sub_wnb | dp.sub_snb) != 0 || (dp.world > 1 && !h->sk_exchange)) return false;
	return uniform_quality == 0 || uniform_quality == 127 || h->hP[uniform_quality] > 0.0;
}
int kmr_add_reads_twobit_dev(kmr_handle *h, const void *dev_twobit, const void *dev_twobit_offsets, const void *dev_offsets,
                             const void *dev_markup_offsets, const void *dev_markup_pos, const void *dev_markup_char,
                             const void *dev_quals, int uniform_quality, uint64_t n_reads, uint64_t total_bases,
                             uint64_t first_global_read_idx, const void *dev_discarded) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (h->finalized) return fail(h, KMR_ERR_STATE, "kmr_add_reads after kmr_finalize");
	if (n_reads == 0) return KMR_OK;
	if (!dev_twobit || !dev_offsets) return fail(h, KMR_ERR_INVALID_ARG, "null device buffer");
	if (dev_markup_offsets && (!dev_markup_pos || !dev_markup_char)) return fail(h, KMR_ERR_INVALID_ARG, "markup offsets without markups");
	if (uniform_quality < 0 || uniform_quality > 255 || (dev_quals && uniform_quality)) return fail(h, KMR_ERR_INVALID_ARG, "uniform_quality: 0, or the one quality character of a batch without a quality array");
	hipSetDevice(h->device);
	/* grow-only scratch of the handle; everything below is ordered on the handle's stream, so the next call's unpack waits for
	 * this call's extraction */
	const bool direct = sk_packed_direct_ok(h, dev_quals != nullptr, uniform_quality);
	if (!direct && h->tb_bases_cap < total_bases + 64) {
		if (h->tb_bases) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree(h->tb_bases); h->tb_bases = nullptr; h->tb_bases_cap = 0; }
		HIPCHK(h, dev_malloc((void **)&h->tb_bases, total_bases + 64)); h->tb_bases_cap = total_bases + 64;
		HIPCHK(h, hipMemsetAsync(h->tb_bases, 0, total_bases + 64, h->stream));
	}
	if (h->tb_n < n_reads + 1) {
		if (h->tb_rel) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree(h->tb_rel); hipFree(h->tb_off); hipFree(h->tb_len); h->tb_rel = h->tb_off = nullptr; h->tb_len = nullptr; h->tb_n = 0; }
		HIPCHK(h, dev_malloc((void **)&h->tb_rel, 8 * (n_reads + 1))); HIPCHK(h, dev_malloc((void **)&h->tb_off, 8 * (n_reads + 1))); HIPCHK(h, dev_malloc((void **)&h->tb_len, 4 * (n_reads + 1)));
		h->tb_n = n_reads + 1;
	}
	const uint64_t *tboff = (const uint64_t *)dev_twobit_offsets;
	if (!tboff) {      /* every read on the byte behind the one before it: ceil(L / 4) bytes each */
		hipLaunchKernelGGL(twobit_bytes_kernel, dim3(grid_for(n_reads)), dim3(256), 0, h->stream, (const uint64_t *)dev_offsets, n_reads, h->tb_len);
		HIPCHK(h, hipGetLastError());
		int rc = exclusive_scan(h, h->tb_len, n_reads, h->tb_off); if (rc) return rc;
		tboff = h->tb_off;
	}
	if (direct) {
		/* the lean extraction stages the packed bytes as they are: no unpacked copy of the batch, no quality bytes */
		hipLaunchKernelGGL(offsets_rel_kernel, dim3(grid_for(n_reads + 1)), dim3(256), 0, h->stream, (const uint64_t *)dev_offsets, n_reads, h->tb_rel);
		HIPCHK(h, hipGetLastError());
		if (h->stream_base + total_bases > MAX_STREAM_ORDINAL) return fail(h, KMR_ERR_CAPACITY, "more than 2^40 input bases on one handle");
		ReadsView rv; rv.bases = nullptr; rv.quals = nullptr; rv.offsets = h->tb_rel; rv.discarded = (const uint8_t *)dev_discarded; rv.n_reads = n_reads;
		rv.stream_base = h->stream_base; rv.first_read_idx = first_global_read_idx; rv.u_start = rv.u_end = rv.u_read = nullptr; rv.n_units = 0;
		SkPacked pkd; pkd.bytes = (const uint8_t *)dev_twobit; pkd.off = tboff; pkd.mk_off = (const uint64_t *)dev_markup_offsets; pkd.mk_pos = (const uint32_t *)dev_markup_pos; pkd.mk_char = (const uint8_t *)dev_markup_char;
		h->packed_direct = &pkd; h->uniform_q_hint = uniform_quality ? uniform_quality : -1;
		const int rc = add_reads_superkmer(h, rv, total_bases);
		h->packed_direct = nullptr; h->uniform_q_hint = -1;
		h->stream_base += total_bases; h->reads += n_reads; h->stats.reads = h->reads;
		return rc;
	}
	hipLaunchKernelGGL(twobit_unpack_kernel, dim3((unsigned)std::min<uint64_t>((n_reads + 255) / 256, (uint64_t)num_cus(h) * 32)), dim3(256), 0, h->stream,
	                   (const uint8_t *)dev_twobit, tboff, (const uint64_t *)dev_offsets, n_reads, h->tb_bases, h->tb_rel);
	HIPCHK(h, hipGetLastError());
	if (dev_markup_offsets) {
		hipLaunchKernelGGL(twobit_markup_kernel, dim3(grid_for(n_reads)), dim3(256), 0, h->stream, (const uint64_t *)dev_markup_offsets, (const uint32_t *)dev_markup_pos,
		                   (const uint8_t *)dev_markup_char, (const uint64_t *)h->tb_rel, n_reads, h->tb_bases);
		HIPCHK(h, hipGetLastError());
	}
	const void *q = dev_quals;
	if (!dev_quals && uniform_quality) {
		if (h->tb_quals_cap < total_bases + 64) {
			if (h->tb_quals) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree(h->tb_quals); h->tb_quals = nullptr; h->tb_quals_cap = 0; }
			HIPCHK(h, dev_malloc((void **)&h->tb_quals, total_bases + 64)); h->tb_quals_cap = total_bases + 64; h->tb_quals_filled = 0; h->tb_quals_char = -1;
		}
		if (h->tb_quals_char != uniform_quality || h->tb_quals_filled < total_bases) {      /* (a buffer the last call filled with the same character stands) */
			HIPCHK(h, hipMemsetAsync(h->tb_quals, uniform_quality, h->tb_quals_cap, h->stream));
			h->tb_quals_char = uniform_quality; h->tb_quals_filled = h->tb_quals_cap;
		}
		q = h->tb_quals;
	}
	/* (dev_quals[0] is the quality of the call's first base: the unpacked batch and its offsets start there too) */
	h->uniform_q_hint = (!dev_quals && uniform_quality) ? uniform_quality : -1;
	const int rc = kmr_add_reads_dev(h, h->tb_bases, q, h->tb_rel, n_reads, total_bases, first_global_read_idx, dev_discarded);
	h->uniform_q_hint = -1;
	return rc;
}

int kmr_add_reads_twobit(kmr_handle *h, const uint8_t *twobit, const uint64_t *twobit_offsets, const uint64_t *offsets,
                         const uint64_t *markup_offsets, const uint32_t *markup_pos, const char *markup_char,
                         const char *quals, int uniform_quality, uint64_t n_reads, uint64_t first_global_read_idx, const uint8_t *discarded) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (h->finalized) return fail(h, KMR_ERR_STATE, "kmr_add_reads after kmr_finalize");
	if (n_reads == 0) return KMR_OK;
	if (!twobit || !twobit_offsets || !offsets) return fail(h, KMR_ERR_INVALID_ARG, "null buffer");
	hipSetDevice(h->device);
	/* The batch goes over in pieces of about 2^28 bases through two sets of staging buffers and a copy stream of the handle's own:
	 * while the device unpacks and extracts piece i, piece i + 1 is on the bus (the host's pageable memory: the calling thread feeds
	 * the copies, the device does not wait for it). */
	{ int rcp = tb_pipeline_ready(h); if (rcp) return rcp; }
	auto stage = [&](int set, int which, const void *src, size_t bytes, void **dst) -> hipError_t { return tb_stage_copy(h, set, which, src, bytes, dst); };
#define TBCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { hipStreamSynchronize(h->tb_copy_stream); hipStreamSynchronize(h->stream); h->err = std::string(#call) + ": " + hip_err_text(e_); return e_ == hipErrorOutOfMemory ? KMR_ERR_OOM : KMR_ERR_HIP; } } while (0)
	const uint64_t piece_bases = h->tune.twobit_piece_bases ? h->tune.twobit_piece_bases : (1ull << 28);
	int rc = KMR_OK; int set = 0;
	h->call_bases_hint = n_reads ? offsets[n_reads] - offsets[0] : 0;      /* what the first piece sizes is the whole call's */
	struct HintReset { kmr_handle *h; ~HintReset() { h->call_bases_hint = 0; } } hint_reset{h};
	for (uint64_t r0 = 0; r0 < n_reads && rc == KMR_OK; set ^= 1) {
		uint64_t r1 = r0 + 1;
		while (r1 < n_reads && offsets[r1 + 1] - offsets[r0] <= piece_bases) r1++;
		const uint64_t m = r1 - r0, total = offsets[r1] - offsets[r0], tbytes = twobit_offsets[r1] - twobit_offsets[r0];
		const uint64_t nm = markup_offsets ? markup_offsets[r1] - markup_offsets[r0] : 0;
		void *dtb = nullptr, *dto = nullptr, *doff = nullptr, *dmo = nullptr, *dmp = nullptr, *dmc = nullptr, *dq = nullptr, *dd = nullptr;
		/* (the offset arrays go over as they are: the unpack kernel counts base offsets from the piece's first one itself, and the packed
		 * bytes / markups are addressed through pointers moved back by the piece's first offset -- no pass over the reads on the host) */
		if (h->tb_set_used[set]) TBCHK(hipStreamWaitEvent(h->tb_copy_stream, h->tb_consumed[set], 0));      /* the piece that used this set last has been unpacked and extracted */
		TBCHK(stage(set, 0, twobit + twobit_offsets[r0], tbytes, &dtb)); TBCHK(stage(set, 1, twobit_offsets + r0, 8 * (m + 1), &dto)); TBCHK(stage(set, 2, offsets + r0, 8 * (m + 1), &doff));
		dtb = (uint8_t *)dtb - twobit_offsets[r0];
		if (nm) {
			TBCHK(stage(set, 3, markup_offsets + r0, 8 * (m + 1), &dmo)); TBCHK(stage(set, 4, markup_pos + markup_offsets[r0], 4 * nm, &dmp)); TBCHK(stage(set, 5, markup_char + markup_offsets[r0], nm, &dmc));
			dmp = (uint32_t *)dmp - markup_offsets[r0]; dmc = (uint8_t *)dmc - markup_offsets[r0];
		}
		if (quals) TBCHK(stage(set, 6, quals + offsets[r0], total, &dq));
		if (discarded) TBCHK(stage(set, 7, discarded + r0, m, &dd));
		TBCHK(hipEventRecord(h->tb_ready[set], h->tb_copy_stream));
		TBCHK(hipStreamWaitEvent(h->stream, h->tb_ready[set], 0));
		rc = kmr_add_reads_twobit_dev(h, dtb, dto, doff, dmo, dmp, dmc, dq, quals ? 0 : uniform_quality, m, total, first_global_read_idx + r0, dd);
		TBCHK(hipEventRecord(h->tb_consumed[set], h->stream)); h->tb_set_used[set] = true;
		r0 = r1;
	}
#undef TBCHK
	hipStreamSynchronize(h->tb_copy_stream);
	if (!rc) rc = sync_state(h);
	else hipStreamSynchronize(h->stream);
	return rc;
}

int kmr_finalize(kmr_handle *h, uint32_t min_depth) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (h->finalized) return fail(h, KMR_ERR_STATE, "already finalized");
	hipSetDevice(h->device);
	{ int src_ = sync_state(h); if (src_) return src_; }
	h->subtract = nullptr;                             /* optimize(): subtractingReference.reset() */
	if (h->superkmer_mode) return finalize_superkmer(h, min_depth);
	if (h->partition_mode) return finalize_partition(h, min_depth);
#define FIN(Wv) (h->ext ? finalize_t<Wv, true>(h, min_depth) : finalize_t<Wv, false>(h, min_depth))
	switch (h->W) { case 1: return FIN(1); case 2: return FIN(2); case 3: return FIN(3); default: return FIN(4); }
#undef FIN
}

int kmr_get_stats(kmr_handle *h, kmr_stats *out) {
	if (!h || !out) return KMR_ERR_INVALID_ARG;
	hipSetDevice(h->device);
	if (!h->finalized) { int rc = sync_state(h); if (rc) return rc; h->stats.unique_kmers = h->partition_mode ? 0 : h->occupied; }
	h->stats.reads = h->reads;
	*out = h->stats;
	return KMR_OK;
}

int kmr_lookup(kmr_handle *h, const uint8_t *packed, uint64_t n, uint32_t *counts) {
	if (!h || (n && (!packed || !counts))) return KMR_ERR_INVALID_ARG;
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_lookup before kmr_finalize");
	if (n == 0) return KMR_OK;
	hipSetDevice(h->device);
	switch (h->W) { case 1: return lookup_t<1>(h, packed, n, counts); case 2: return lookup_t<2>(h, packed, n, counts);
	case 3: return lookup_t<3>(h, packed, n, counts); default: return lookup_t<4>(h, packed, n, counts); }
}

int kmr_lookup_reads(kmr_handle *h, const char *bases, const uint64_t *offsets, uint64_t n_reads, uint32_t *counts_out, const uint64_t *out_offsets) {
	if (!h || !bases || !offsets || !counts_out || !out_offsets) return KMR_ERR_INVALID_ARG;
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_lookup_reads before kmr_finalize");
	if (n_reads == 0) return KMR_OK;
	hipSetDevice(h->device);
	StagedReads s; uint64_t total = 0;
	int rc = stage_reads(h, bases, nullptr, offsets, n_reads, nullptr, s, total);
	if (rc) { s.release(); return rc; }
	/* size of the output = last offset + k-mers of the last read */
	uint64_t outN = 0;
	for (uint64_t r = 0; r < n_reads; r++) { uint64_t L = offsets[r + 1] - offsets[r]; uint64_t nk = L >= h->k ? L - h->k + 1 : 0; outN = std::max(outN, out_offsets[r] + nk); }
	uint32_t *dout; uint64_t *doff;
	HIPCHK(h, dev_malloc((void **)&dout, std::max<uint64_t>(8, 4 * outN))); HIPCHK(h, dev_malloc((void **)&doff, 8 * n_reads));
	HIPCHK(h, hipMemsetAsync(dout, 0, 4 * outN, h->stream));
	HIPCHK(h, hipMemcpyAsync(doff, out_offsets, 8 * n_reads, hipMemcpyHostToDevice, h->stream));
	ReadsView rv; rv.bases = s.b; rv.quals = nullptr; rv.offsets = s.o; rv.discarded = nullptr; rv.n_reads = n_reads; rv.stream_base = 0; rv.first_read_idx = 0; rv.u_start = rv.u_end = rv.u_read = nullptr; rv.n_units = 0;
	{ int urc = prepare_units(h, rv); if (urc) { s.release(); return urc; } }
	switch (h->W) { case 1: rc = lookup_reads_t<1>(h, rv, dout, doff); break; case 2: rc = lookup_reads_t<2>(h, rv, dout, doff); break;
	case 3: rc = lookup_reads_t<3>(h, rv, dout, doff); break; default: rc = lookup_reads_t<4>(h, rv, dout, doff); }
	if (!rc) { HIPCHK(h, hipMemcpyAsync(counts_out, dout, 4 * outN, hipMemcpyDeviceToHost, h->stream)); rc = sync_state(h); }
	else hipStreamSynchronize(h->stream);
	hipFree(dout); hipFree(doff); s.release();
	return rc;
}

}  // extern "C"
/* ---- lookups as a streaming pass (kmr_superkmer.hpp, "lookups as a streaming pass"): position_counts[offsets[r] + i] = weak-map
 * count of k-mer i of read r, zero where the k-mer is absent or holds an N.  The handle's list pool and list state are reused
 * (after kmr_finalize the build's lists are dead). */
bool stream_lookups_possible(kmr_handle *h) {
	return h->dPk && !h->tune.no_stream_lookups && h->weak.present && h->weak.n > 0 && h->cfg.size_tracker == 0 && !h->sk_exchange;
}
template <int W> int sk_index_t(kmr_handle *h) {
	if (h->ix_gen == h->map_gen && h->ix_start) return 0;
	const uint64_t nl = sk_list_count(h->sk_bits), n = h->weak.n;
	const uint32_t vw = h->ext ? 15 : 3;
	if (h->ix_lists != nl) { if (h->ix_start) hipFree(h->ix_start); h->ix_start = nullptr; HIPCHK(h, dev_malloc((void **)&h->ix_start, 8 * (nl + 1))); h->ix_lists = nl; }
	if (h->ix_cap < n) {
		if (h->ix_keys) hipFree(h->ix_keys); if (h->ix_counts) hipFree(h->ix_counts); h->ix_keys = nullptr; h->ix_counts = nullptr; h->ix_cap = 0;
		HIPCHK(h, dev_malloc((void **)&h->ix_keys, 8ull * W * n)); HIPCHK(h, dev_malloc((void **)&h->ix_counts, 4 * n)); h->ix_cap = n;
	}
	uint32_t *elist = nullptr, *hist = nullptr;
	int rc = arena_get(h, &elist, n); if (rc) return rc;
	rc = arena_get(h, &hist, nl); if (rc) return rc;
	HIPCHK(h, hipMemsetAsync(hist, 0, 4 * nl, h->stream));
	hipLaunchKernelGGL(sk_index_hist_kernel<W>, dim3(grid_for(n)), dim3(256), 0, h->stream, (const uint64_t *)h->weak.keys, n, h->sk_m, h->sk_off, h->sk_win, h->sk_bits, elist, hist);
	HIPCHK(h, hipGetLastError());
	rc = exclusive_scan(h, hist, nl, h->ix_start); if (rc) return rc;
	HIPCHK(h, hipMemsetAsync(hist, 0, 4 * nl, h->stream));
	hipLaunchKernelGGL(sk_index_scatter_kernel<W>, dim3(grid_for(n)), dim3(256), 0, h->stream, (const uint64_t *)h->weak.keys, (const uint32_t *)h->weak.vals, vw, n, elist, h->ix_start, hist, h->ix_keys, h->ix_counts);
	HIPCHK(h, hipGetLastError());
	h->ix_gen = h->map_gen;
	return 0;
}
template <int W> int lookup_stream_t(kmr_handle *h, const ReadsView &rvAll, uint64_t total_bases, uint32_t *position_counts, uint64_t out_n) {
	int rc = arena_reset(h); if (rc) return rc;
	if (!h->sk_state) {      /* a handle that was not built on the lists (loaded image, other build mode): lists sized for this batch */
		uint32_t bits = 6; while (bits < 24 && (total_bases >> bits) > h->tune.target_list / 2 + 200) bits++;
		h->sk_bits = bits;
		HIPCHK(h, dev_malloc((void **)&h->sk_state, 8ull << bits));
	}
	if (!h->scratch_stats) HIPCHK(h, dev_malloc((void **)&h->scratch_stats, sizeof(DevStats)));
	rc = sk_index_t<W>(h); if (rc) return rc;
	const uint64_t nl = sk_list_count(h->sk_bits);
	hipLaunchKernelGGL(sk_state_init_kernel, dim3(grid_for(nl)), dim3(256), 0, h->stream, h->sk_state, nl);
	if (h->l1.head) HIPCHK(h, hipMemsetAsync(h->l1.head, 0, 4, h->stream));
	h->l1.used_ub = 0;
	ReadsView rv = rvAll;
	rc = prepare_units(h, rv); if (rc) return rc;
	rc = pool_reserve(h, h->l1, total_bases / SK_CHUNK_G + nl + (uint64_t)num_cus(h) * SK_EXTRACT_WAVES_PER_CU * 130 + 64, true); if (rc) return rc;
	/* every k-mer without an N is asked for: no qualities (weight 1, or 0 with an N), no filters, nothing added to the handle's counters */
	DevParams dp = dev_params(h);
	dp.min_weight = 0.5f; dp.subsample = 1; dp.world = 1; dp.num_parts = 1; dp.sub_wnb = 0; dp.sub_snb = 0; dp.stats = h->scratch_stats;
	SkParams sp = sk_params(h); sp.keep_all_owners = 1; sp.track = nullptr;
	if (W > 1 && h->sk_win == 32) rc = (!rv.quals && !h->tune.no_lean_extract) ? launch_sk_extract_lean<W, (W > 1 ? 32 : 16)>(h, rv, sp, 1.0f, &dp) : launch_sk_extract<W, (W > 1 ? 32 : 16), false>(h, rv, sp, &dp);
	else if (!rv.quals && !h->tune.no_lean_extract) rc = h->sk_win == 16 ? launch_sk_extract_lean<W, 16>(h, rv, sp, 1.0f, &dp) : (h->sk_win == 8 ? launch_sk_extract_lean<W, 8>(h, rv, sp, 1.0f, &dp) : launch_sk_extract_lean<W, 4>(h, rv, sp, 1.0f, &dp));
	else rc = h->sk_win == 16 ? launch_sk_extract<W, 16, false>(h, rv, sp, &dp) : (h->sk_win == 8 ? launch_sk_extract<W, 8, false>(h, rv, sp, &dp) : launch_sk_extract<W, 4, false>(h, rv, sp, &dp));
	if (rc) return rc;
	hipLaunchKernelGGL(sk_close_kernel, dim3(grid_for(nl)), dim3(256), 0, h->stream, h->sk_state, nl, h->l1.chunk_count, h->l1.cap);
	HIPCHK(h, hipGetLastError());
	uint64_t *ls = nullptr, *lc = nullptr; uint32_t nch = 0;
	rc = build_csr(h, h->l1, nl, 0, &ls, &lc, &nch); if (rc) return rc;
	rc = zero_work_counter(h); if (rc) return rc;
	const int grid = (int)std::min<uint64_t>((uint64_t)num_cus(h) * 4, (nl + SK_LBATCH - 1) / SK_LBATCH);
	auto kern = sk_lookup_kernel<W>;
	const size_t smem = sk_lookup_smem_bytes<W>();
	HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
	/* long lists in pieces (as in the count pass; here the pieces need no merge) */
	const uint64_t LONG_CHUNKS = h->tune.long_list_chunks ? h->tune.long_list_chunks : 1024, PIECE = std::max<uint64_t>(1, LONG_CHUNKS / 2);
	uint64_t n_items = 0; uint64_t *ic0 = nullptr, *ic1 = nullptr; uint32_t *il = nullptr;
	if (nch > LONG_CHUNKS) {
		const uint64_t cap = (uint64_t)nch / PIECE + 2 * 1024 + 16;
		unsigned long long *dn = nullptr;
		rc = arena_get(h, &ic0, cap); if (rc) return rc; rc = arena_get(h, &ic1, cap); if (rc) return rc; rc = arena_get(h, &il, cap); if (rc) return rc; rc = arena_get(h, &dn, 1); if (rc) return rc;
		HIPCHK(h, hipMemsetAsync(dn, 0, 8, h->stream));
		hipLaunchKernelGGL(sk_long_items_kernel, dim3(grid_for(nl)), dim3(256), 0, h->stream, ls, nl, LONG_CHUNKS, PIECE, ic0, ic1, cap, dn, il);
		HIPCHK(h, hipGetLastError());
		unsigned long long hn = 0;
		HIPCHK(h, hipMemcpyAsync(&hn, dn, 8, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream));
		if (hn > cap) return fail(h, KMR_ERR_CAPACITY, "long-list work items (internal sizing error)");
		n_items = hn;
	}
	hipLaunchKernelGGL(kern, dim3(grid), dim3(COUNT_THREADS), smem, h->stream, pool_view(h, h->l1), ls, lc, nl, h->k, h->ix_start, h->ix_keys, h->ix_counts, position_counts, out_n, h->work_counter,
	                   (const uint64_t *)nullptr, (const uint64_t *)nullptr, (const uint32_t *)nullptr, (uint64_t)0, n_items ? LONG_CHUNKS : (uint64_t)0);
	HIPCHK(h, hipGetLastError());
	if (n_items) {
		rc = zero_work_counter(h); if (rc) return rc;
		hipLaunchKernelGGL(kern, dim3((unsigned)std::min<uint64_t>((uint64_t)num_cus(h) * 4, n_items)), dim3(COUNT_THREADS), smem, h->stream, pool_view(h, h->l1), ls, lc, nl, h->k, h->ix_start, h->ix_keys, h->ix_counts,
		                   position_counts, out_n, h->work_counter, (const uint64_t *)ic0, (const uint64_t *)ic1, (const uint32_t *)il, n_items, (uint64_t)0);
		HIPCHK(h, hipGetLastError());
	}
	return 0;
}
int lookup_stream(kmr_handle *h, const ReadsView &rv, uint64_t total_bases, uint32_t *position_counts, uint64_t out_n) {
	switch (h->W) { case 1: return lookup_stream_t<1>(h, rv, total_bases, position_counts, out_n); case 2: return lookup_stream_t<2>(h, rv, total_bases, position_counts, out_n);
	case 3: return lookup_stream_t<3>(h, rv, total_bases, position_counts, out_n); default: return lookup_stream_t<4>(h, rv, total_bases, position_counts, out_n); }
}

extern "C" {
/* ReadSelector::scoreAndTrimReads (src/ReadSelector.h:1182-1207) on the weak map; s_b / s_o: device bases and offsets,
 * offsets: the same offsets on the host */
static int score_reads_core(kmr_handle *h, const uint8_t *s_b, const uint64_t *s_o, uint64_t n_reads, double minimum_kmer_score, int scoring_type,
                            uint32_t *trim_offset, uint32_t *trim_length, float *score, uint8_t *was_trimmed) {
	int rc = 0;
	ReadsView rv; rv.bases = s_b; rv.quals = nullptr; rv.offsets = s_o; rv.discarded = nullptr; rv.n_reads = n_reads; rv.stream_base = 0; rv.first_read_idx = 0; rv.u_start = rv.u_end = rv.u_read = nullptr; rv.n_units = 0;
	/* per-read k-mer counts and their exclusive scan on the device (no host pass over the reads); one grow-only block for
	 * every temporary (a hipFree of the ~GB count array costs more than the scoring) */
	auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t fixed = al(4 * (n_reads + 1)) + al(8 * (n_reads + 1)) + 3 * al(4 * n_reads) + al(n_reads);
	if (h->score_buf_bytes < fixed) {
		if (h->score_buf) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree(h->score_buf); h->score_buf = nullptr; h->score_buf_bytes = 0; }
		HIPCHK(h, dev_malloc((void **)&h->score_buf, fixed + fixed / 4)); h->score_buf_bytes = fixed + fixed / 4;
	}
	uint8_t *p = h->score_buf;
	uint32_t *dkc = (uint32_t *)p; p += al(4 * (n_reads + 1));
	uint64_t *dcoff = (uint64_t *)p;
	/* counts per k-mer: as a streaming pass over minimizer lists (indexed by the k-mer's base position: the reads' own offsets are the
	 * count offsets), or -- where the handle has no list geometry, or on request -- by probing the lookup table k-mer by k-mer */
	const bool stream = stream_lookups_possible(h);
	uint64_t outN = 0, first_off = 0;
	if (stream) {
		HIPCHK(h, hipMemcpy(&outN, s_o + n_reads, 8, hipMemcpyDeviceToHost)); HIPCHK(h, hipMemcpy(&first_off, s_o, 8, hipMemcpyDeviceToHost));
	} else {
		hipLaunchKernelGGL(kmer_capacity_kernel, dim3(grid_for(n_reads)), dim3(256), 0, h->stream, rv, h->k, dkc);
		HIPCHK(h, hipGetLastError());
		rc = exclusive_scan(h, dkc, n_reads, dcoff); if (rc) return rc;
		HIPCHK(h, hipMemcpy(&outN, dcoff + n_reads, 8, hipMemcpyDeviceToHost));
	}
	const size_t need = fixed + al(std::max<uint64_t>(8, 4 * outN));
	if (h->score_buf_bytes < need) {         /* grow, keeping the scan */
		uint8_t *nbuf; HIPCHK(h, dev_malloc((void **)&nbuf, need + need / 8));
		HIPCHK(h, hipMemcpy(nbuf, h->score_buf, al(4 * (n_reads + 1)) + al(8 * (n_reads + 1)), hipMemcpyDeviceToDevice));
		HIPCHK(h, hipDeviceSynchronize());
		hipFree(h->score_buf); h->score_buf = nbuf; h->score_buf_bytes = need + need / 8;
	}
	p = h->score_buf + al(4 * (n_reads + 1));
	dcoff = (uint64_t *)p; p += al(8 * (n_reads + 1));
	uint32_t *dto = (uint32_t *)p; p += al(4 * n_reads);
	uint32_t *dtl = (uint32_t *)p; p += al(4 * n_reads);
	float *dsc = (float *)p; p += al(4 * n_reads);
	uint8_t *dwt = p; p += al(n_reads);
	uint32_t *dcounts = (uint32_t *)p;
	HIPCHK(h, hipMemsetAsync(dcounts, 0, 4 * outN, h->stream));
	if (stream) {
		rc = lookup_stream(h, rv, outN - first_off, dcounts, outN);
		dcoff = (uint64_t *)s_o;
	} else {
		{ int urc = prepare_units(h, rv); if (urc) return urc; }
		switch (h->W) { case 1: rc = lookup_reads_t<1>(h, rv, dcounts, dcoff, true); break; case 2: rc = lookup_reads_t<2>(h, rv, dcounts, dcoff, true); break;
		case 3: rc = lookup_reads_t<3>(h, rv, dcounts, dcoff, true); break; default: rc = lookup_reads_t<4>(h, rv, dcounts, dcoff, true); }
	}
	if (!rc) {
		hipLaunchKernelGGL(score_reads_kernel, dim3((unsigned)std::min<uint64_t>(((n_reads + 63) / 64 + SC_WAVES - 1) / SC_WAVES, 1u << 16)), dim3(SC_WAVES * 64), 0, h->stream, s_b, s_o, n_reads, h->k, dcounts, dcoff,
		                   (float)minimum_kmer_score, scoring_type, dto, dtl, dsc, dwt);
		HIPCHK(h, hipGetLastError());
		HIPCHK(h, hipMemcpyAsync(trim_offset, dto, 4 * n_reads, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipMemcpyAsync(trim_length, dtl, 4 * n_reads, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(h, hipMemcpyAsync(score, dsc, 4 * n_reads, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipMemcpyAsync(was_trimmed, dwt, n_reads, hipMemcpyDeviceToHost, h->stream));
		rc = sync_state(h);
	} else hipStreamSynchronize(h->stream);
	return rc;
}
int kmr_score_reads(kmr_handle *h, const char *bases, const uint64_t *offsets, uint64_t n_reads, double minimum_kmer_score, int scoring_type,
                    uint32_t *trim_offset, uint32_t *trim_length, float *score, uint8_t *was_trimmed) {
	if (!h || !bases || !offsets || !trim_offset || !trim_length || !score || !was_trimmed) return KMR_ERR_INVALID_ARG;
	if (scoring_type < 0 || scoring_type > 4) return fail(h, KMR_ERR_INVALID_ARG, "bad scoring_type");
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_score_reads before kmr_finalize");
	if (n_reads == 0) return KMR_OK;
	hipSetDevice(h->device);
	StagedReads s; uint64_t total = 0;
	int rc = stage_reads(h, bases, nullptr, offsets, n_reads, nullptr, s, total);
	if (!rc) {
		rc = score_reads_core(h, s.b, s.o, n_reads, minimum_kmer_score, scoring_type, trim_offset, trim_length, score, was_trimmed);
	}
	s.release();
	return rc;
}
/* the same on a device-resident read batch (kmr_ingest_fastq): FASTQ text -> reads -> spectrum -> trim/score without the
 * reads ever being staged by the host */
int kmr_score_read_batch(kmr_handle *h, const kmr_reads *r, double minimum_kmer_score, int scoring_type,
                         uint32_t *trim_offset, uint32_t *trim_length, float *score, uint8_t *was_trimmed) {
	if (!h || !r || !trim_offset || !trim_length || !score || !was_trimmed) return KMR_ERR_INVALID_ARG;
	if (scoring_type < 0 || scoring_type > 4) return fail(h, KMR_ERR_INVALID_ARG, "bad scoring_type");
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_score_read_batch before kmr_finalize");
	if (r->device != h->device) return fail(h, KMR_ERR_INVALID_ARG, "read batch lives on another device");
	if (r->n == 0) return KMR_OK;
	hipSetDevice(h->device);
	return score_reads_core(h, r->bases, r->offsets, r->n, minimum_kmer_score, scoring_type, trim_offset, trim_length, score, was_trimmed);
}

static DevMap *map_of(kmr_handle *h, int which) {
	if (which == KMR_MAP_WEAK) return &h->weak;
	if (which == KMR_MAP_SINGLETON) return &h->sing;
	return nullptr;
}

int kmr_image_size(kmr_handle *h, int which, uint64_t *bytes) {
	if (!h || !bytes) return KMR_ERR_INVALID_ARG;
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_image_size before kmr_finalize");
	DevMap *m = map_of(h, which);
	if (!m) return fail(h, KMR_ERR_UNSUPPORTED, "solid map is not built on this path");
	const uint32_t vbytes = which == KMR_MAP_WEAK ? (h->ext ? 60 : 12) : (h->ext ? 5 : 1);
	const uint64_t n = (which == KMR_MAP_SINGLETON && !m->present) ? 0 : m->n;
	*bytes = 8 * (2 + m->nb) + 4 * m->nb + n * (h->kb + vbytes);
	return KMR_OK;
}

int kmr_write_image(kmr_handle *h, int which, void *dst, uint64_t capacity) {
	if (!h || !dst) return KMR_ERR_INVALID_ARG;
	uint64_t need; int rc = kmr_image_size(h, which, &need); if (rc) return rc;
	if (capacity < need) return fail(h, KMR_ERR_CAPACITY, "image buffer too small");
	hipSetDevice(h->device);
	DevMap *m = map_of(h, which);
	if (which == KMR_MAP_SINGLETON && !m->present) {   /* cleared singleton map: numBuckets empty buckets */
		uint8_t *p = (uint8_t *)dst; uint64_t nb = m->nb, mask = nb - 1; memcpy(p, &nb, 8); memcpy(p + 8, &mask, 8);
		for (uint64_t b = 0; b < nb; b++) { uint64_t off = 8 * (2 + nb) + 4 * b; memcpy(p + 16 + 8 * b, &off, 8); uint32_t z = 0; memcpy(p + off, &z, 4); }
		return KMR_OK;
	}
	rc = build_image(h, *m, which == KMR_MAP_WEAK); if (rc) return rc;
	HIPCHK(h, hipMemcpy(dst, m->image, need, hipMemcpyDeviceToHost));
	return KMR_OK;
}

int kmr_load_image(kmr_handle *h, int which, const void *src, uint64_t len) {
	if (!h || !src) return KMR_ERR_INVALID_ARG;
	if (h->reads != 0) return fail(h, KMR_ERR_STATE, "kmr_load_image needs a fresh handle");
	DevMap *m = map_of(h, which);
	if (!m) return fail(h, KMR_ERR_UNSUPPORTED, "solid map is not built on this path");
	hipSetDevice(h->device);
	int rc;
	switch (h->W) { case 1: rc = load_image_t<1>(h, *m, which == KMR_MAP_WEAK, (const uint8_t *)src, len); break;
	case 2: rc = load_image_t<2>(h, *m, which == KMR_MAP_WEAK, (const uint8_t *)src, len); break;
	case 3: rc = load_image_t<3>(h, *m, which == KMR_MAP_WEAK, (const uint8_t *)src, len); break;
	default: rc = load_image_t<4>(h, *m, which == KMR_MAP_WEAK, (const uint8_t *)src, len); }
	if (rc) return rc;
	if (h->slots) { hipFree(h->slots); h->slots = nullptr; if (h->extslots) { hipFree(h->extslots); h->extslots = nullptr; } }
	if (which == KMR_MAP_WEAK) { h->nb_weak = m->nb; h->stats.weak_entries = m->n; if (!h->sing.present) { h->has_singletons = false; h->sing.nb = h->nb_sing; } }
	else { h->nb_sing = m->nb; h->stats.singleton_entries = m->n; h->has_singletons = true; if (!h->weak.present) h->weak.nb = h->nb_weak; }
	h->finalized = true; h->map_gen++;
	return KMR_OK;
}

int kmr_count_histogram(kmr_handle *h, uint64_t *counts, double *weights, uint32_t n_bins) {
	if (!h || !counts || n_bins < 2) return KMR_ERR_INVALID_ARG;
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_count_histogram before kmr_finalize");
	hipSetDevice(h->device);
	unsigned long long *dc; double *dw = nullptr;
	HIPCHK(h, dev_malloc((void **)&dc, 8 * n_bins)); HIPCHK(h, hipMemsetAsync(dc, 0, 8 * n_bins, h->stream));
	if (weights) { HIPCHK(h, dev_malloc((void **)&dw, 8 * n_bins)); HIPCHK(h, hipMemsetAsync(dw, 0, 8 * n_bins, h->stream)); }
	if (h->weak.present && h->weak.n)
		hipLaunchKernelGGL(histogram_kernel, dim3(grid_for(h->weak.n)), dim3(256), 0, h->stream, h->weak.vals, h->ext ? 15u : 3u, h->weak.n, n_bins, dc, dw);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipMemcpyAsync(counts, dc, 8 * n_bins, hipMemcpyDeviceToHost, h->stream));
	if (weights) HIPCHK(h, hipMemcpyAsync(weights, dw, 8 * n_bins, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	hipFree(dc); if (dw) hipFree(dw);
	return KMR_OK;
}

/* Order-independent digest of a finalized map (see kmr_synth.hpp): what a full-size build is compared by, and what the ranks or
 * parts of a partitioned build add up to. */
int kmr_map_digest(kmr_handle *h, int which_map, kmr_digest *out) {
	if (!h || !out) return KMR_ERR_INVALID_ARG;
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_map_digest before kmr_finalize");
	if (which_map != KMR_MAP_WEAK && which_map != KMR_MAP_SINGLETON) return fail(h, KMR_ERR_UNSUPPORTED, "no such map");
	static_assert(sizeof(kmr_digest) == sizeof(synth::Digest), "kmr_digest layout");
	hipSetDevice(h->device);
	memset(out, 0, sizeof(*out));
	const DevMap &m = which_map == KMR_MAP_WEAK ? h->weak : h->sing;
	if (which_map == KMR_MAP_SINGLETON && !h->has_singletons) return KMR_OK;
	if (!m.present || !m.n) return KMR_OK;
	synth::Digest *d;
	HIPCHK(h, dev_malloc((void **)&d, sizeof(synth::Digest))); HIPCHK(h, hipMemsetAsync(d, 0, sizeof(synth::Digest), h->stream));
	if (which_map == KMR_MAP_WEAK)
		hipLaunchKernelGGL(synth::map_digest_kernel, dim3(grid_for(m.n, 256, 4096)), dim3(256), 0, h->stream, m.keys, (uint32_t)h->W, m.vals, h->ext ? 15u : 3u, (const uint8_t *)nullptr, (const uint32_t *)nullptr, m.n, d);
	else
		hipLaunchKernelGGL(synth::map_digest_kernel, dim3(grid_for(m.n, 256, 4096)), dim3(256), 0, h->stream, m.keys, (uint32_t)h->W, (const uint32_t *)nullptr, 0u, m.sweight, h->ext ? m.spkt : (const uint32_t *)nullptr, m.n, d);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipMemcpyAsync(out, d, sizeof(synth::Digest), hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	hipFree(d);
	out->entries = m.n;
	return KMR_OK;
}

/* SURVEY.md section 8(d)'s synthetic reads, written into caller-owned device memory on the current device (kmr_synth.hpp holds
 * the definition of the generator): reads first_read .. first_read + n_reads of the job `seed`, read_len bases each. */
int kmr_synth_reads_dev(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t genome_len, uint32_t noisy_quals,
                        void *dev_bases, void *dev_quals, uint64_t *dev_offsets) {
	if (!dev_bases || read_len == 0 || genome_len < read_len) return KMR_ERR_INVALID_ARG;
	if (n_reads == 0 && !dev_offsets) return KMR_OK;
	const uint64_t blocks = (n_reads + 1 + 255) / 256;
	if (blocks > 0x7fffffffull) return KMR_ERR_CAPACITY;
	hipLaunchKernelGGL(synth::synth_reads_kernel, dim3((uint32_t)blocks), dim3(256), 0, 0, seed, first_read, n_reads, read_len, genome_len, noisy_quals,
	                   (uint8_t *)dev_bases, (uint8_t *)dev_quals, dev_offsets);
	if (hipGetLastError() != hipSuccess) return KMR_ERR_HIP;
	return hipStreamSynchronize(0) == hipSuccess ? KMR_OK : KMR_ERR_HIP;
}

/* KmerSpectrum::subtractReference (src/KmerSpectrum.h:472-474; apps/FilterReads-P.cpp:117): k-mers that exist in the finalized
 * spectrum `reference` are skipped by every later kmr_add_reads* of h (append(), :1582-1588).  kmr_finalize(h) drops the link,
 * as optimize() does (:463-464); reference == NULL drops it at once.  `reference` must stay alive and unchanged meanwhile. */
int kmr_subtract_reference(kmr_handle *h, kmr_handle *reference) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (reference) {
		if (reference == h) return fail(h, KMR_ERR_INVALID_ARG, "a spectrum cannot subtract itself");
		if (!reference->finalized) return fail(h, KMR_ERR_STATE, "the subtracting spectrum must be finalized");
		if (reference->k != h->k || reference->device != h->device) return fail(h, KMR_ERR_INVALID_ARG, "the subtracting spectrum must have the same k and live on the same device");
		if (h->finalized) return fail(h, KMR_ERR_STATE, "kmr_subtract_reference after kmr_finalize");
	}
	hipSetDevice(h->device);
	int rc = sync_state(h); if (rc) return rc;         /* launches in flight carry the old link */
	h->subtract = reference;
	return KMR_OK;
}
int kmr_subtracted(kmr_handle *h, uint64_t *out) {
	if (!h || !out) return KMR_ERR_INVALID_ARG;
	hipSetDevice(h->device);
	if (!h->finalized) { int rc = sync_state(h); if (rc) return rc; }
	*out = h->subtracted;
	return KMR_OK;
}

/* merge a stored part into the finalized maps (buildKmerSpectrumInParts' restore-and-merge loop, src/KmerSpectrum.h:1871-1884) */
int kmr_merge_image(kmr_handle *h, int which, const void *src, uint64_t len) {
	if (!h || !src) return KMR_ERR_INVALID_ARG;
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_merge_image before kmr_finalize / kmr_load_image");
	DevMap *m = map_of(h, which);
	if (!m) return fail(h, KMR_ERR_UNSUPPORTED, "solid map is not built on this path");
	if (!m->present) return fail(h, KMR_ERR_STATE, "this map is not present in the handle (singletons purged?)");
	hipSetDevice(h->device);
	const bool weakMap = which == KMR_MAP_WEAK;
	int rc;
	switch (h->W) { case 1: rc = merge_image_t<1>(h, *m, weakMap, (const uint8_t *)src, len); break; case 2: rc = merge_image_t<2>(h, *m, weakMap, (const uint8_t *)src, len); break;
	case 3: rc = merge_image_t<3>(h, *m, weakMap, (const uint8_t *)src, len); break; default: rc = merge_image_t<4>(h, *m, weakMap, (const uint8_t *)src, len); }
	h->map_gen++;
	if (rc) return rc;
	if (weakMap) h->stats.weak_entries = m->n; else h->stats.singleton_entries = m->n;
	return KMR_OK;
}

/* KmerSpectrum::Histogram (src/KmerSpectrum.h:909-1057) of the finalized spectrum: Histogram(zoom_max, log_base).set(ks) */
uint32_t kmr_histogram_bins(uint32_t zoom_max) { return (1u << 16) + 1u + zoom_max + 1u; }       /* ctor :948-950 */
int kmr_histogram(kmr_handle *h, uint32_t zoom_max, double log_base, uint64_t *visits, uint64_t *visited_count, double *visited_weight, uint32_t n_bins) {
	if (!h || !visits || !visited_count || !visited_weight) return KMR_ERR_INVALID_ARG;
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_histogram before kmr_finalize");
	if (zoom_max > 65535 || !(log_base > 1.0)) return fail(h, KMR_ERR_INVALID_ARG, "bad zoom_max / log_base");
	const uint32_t nb = kmr_histogram_bins(zoom_max);
	if (n_bins < nb) return fail(h, KMR_ERR_CAPACITY, "histogram arrays need kmr_histogram_bins(zoom_max) entries");
	hipSetDevice(h->device);
	/* getIdx (:936-938) for every 16-bit count, with the host's libm */
	std::vector<uint32_t> lut(65536, 0);
	const double logFactor = log(log_base);
	const unsigned int zoomLogSkip = (unsigned int)(log((double)zoom_max + 1.0) / logFactor - 1.0);
	for (uint32_t c = 1; c < 65536; c++) lut[c] = c <= zoom_max ? c : (unsigned int)(log((double)c) / logFactor - zoomLogSkip + zoom_max);
	uint32_t *dl; unsigned long long *dv, *dc; double *dw;
	HIPCHK(h, dev_malloc((void **)&dl, 4 * 65536)); HIPCHK(h, dev_malloc((void **)&dv, 8ull * nb)); HIPCHK(h, dev_malloc((void **)&dc, 8ull * nb)); HIPCHK(h, dev_malloc((void **)&dw, 8ull * nb));
	HIPCHK(h, hipMemcpyAsync(dl, lut.data(), 4 * 65536, hipMemcpyHostToDevice, h->stream));
	HIPCHK(h, hipMemsetAsync(dv, 0, 8ull * nb, h->stream)); HIPCHK(h, hipMemsetAsync(dc, 0, 8ull * nb, h->stream)); HIPCHK(h, hipMemsetAsync(dw, 0, 8ull * nb, h->stream));
	if (h->weak.present && h->weak.n)
		hipLaunchKernelGGL(ref_histogram_kernel, dim3(grid_for(h->weak.n, 256, 2048)), dim3(256), 0, h->stream, h->weak.vals, h->ext ? 15u : 3u, (const uint8_t *)nullptr, h->weak.n, dl, dv, dc, dw);
	if (h->has_singletons && h->sing.present && h->sing.n)
		hipLaunchKernelGGL(ref_histogram_kernel, dim3(grid_for(h->sing.n, 256, 2048)), dim3(256), 0, h->stream, (const uint32_t *)nullptr, 0u, h->sing.sweight, h->sing.n, dl, dv, dc, dw);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipMemcpyAsync(visits, dv, 8ull * nb, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipMemcpyAsync(visited_count, dc, 8ull * nb, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipMemcpyAsync(visited_weight, dw, 8ull * nb, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	hipFree(dl); hipFree(dv); hipFree(dc); hipFree(dw);
	return KMR_OK;
}

/* text dumps (src/Meraculous.h:107-134): formatting is host work on the downloaded weak map */
static int dump_text(kmr_handle *h, const char *path, uint32_t min_depth, bool graph) {
	if (!h || !path) return KMR_ERR_INVALID_ARG;
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "dump before kmr_finalize");
	if (graph && !h->ext) return fail(h, KMR_ERR_STATE, "mergraph needs value_kind = KMR_VALUE_EXT");
	hipSetDevice(h->device);
	const uint32_t vw = h->ext ? 15 : 3, W = h->W, k = h->k;
	const uint64_t n = h->weak.n;
	std::vector<uint64_t> keys(n * W); std::vector<uint32_t> vals(n * vw);
	if (n) { HIPCHK(h, hipMemcpy(keys.data(), h->weak.keys, 8 * n * W, hipMemcpyDeviceToHost)); HIPCHK(h, hipMemcpy(vals.data(), h->weak.vals, 4 * n * vw, hipMemcpyDeviceToHost)); }
	FILE *f = fopen(path, "a");
	if (!f) return fail(h, KMR_ERR_INVALID_ARG, std::string("cannot open ") + path);
	std::string fa(k, 'A'), rfa(k, 'A');
	static const char dec[] = "ACGT";
	static const int rcIdx[6] = {3, 2, 1, 0, 4, 5};
	for (uint64_t e = 0; e < n; e++) {
		const uint32_t *v = &vals[e * vw];
		const uint32_t count = v[0] & 0xffff;
		if ((int)count < (int)min_depth) continue;
		for (uint32_t p = 0; p < k; p++) {
			const uint32_t code = (uint32_t)(keys[e * W + (p >> 5)] >> (62 - 2 * (p & 31))) & 3;
			fa[p] = dec[code]; rfa[k - 1 - p] = dec[3 - code];
		}
		if (!graph) fprintf(f, "%s\t%u\n%s\t%u\n", fa.c_str(), count, rfa.c_str(), count);
		else {
			const uint32_t *t = v + 3;
			fprintf(f, "%s\t%u %u %u %u %u %u %u %u %u %u %u %u 0\n", fa.c_str(), t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11]);
			uint32_t r[12];   /* ExtensionTracking::getReverseComplement, src/KmerTrackingData.h:219-226 */
			for (int i = 0; i < 6; i++) { r[rcIdx[i]] = t[6 + i]; r[6 + rcIdx[i]] = t[i]; }
			fprintf(f, "%s\t%u %u %u %u %u %u %u %u %u %u %u %u 0\n", rfa.c_str(), r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]);
		}
	}
	fclose(f);
	return KMR_OK;
}
int kmr_dump_mercount(kmr_handle *h, const char *path, uint32_t min_depth) { return dump_text(h, path, min_depth, false); }
int kmr_dump_mergraph(kmr_handle *h, const char *path, uint32_t min_depth) { return dump_text(h, path, min_depth, true); }

/* ---- f2: FASTQ ingest on the device (kmr_ingest.hpp) ------------------------ */
static int ingest_dev(kmr_handle *h, const uint8_t *text, uint64_t len, uint32_t input_base, int store_comment, kmr_reads **out) {
	const uint32_t start = h->cfg.fastq_start_char;
	if (input_base == 0) input_base = start;
	if ((input_base != 33 && input_base != 64) || (start != 33 && start != 64))
		return fail(h, KMR_ERR_INVALID_ARG, "fastq quality base must be 33 or 64 (src/Options.h:490)");
	kmr_reads *R = new kmr_reads(); R->device = h->device; R->input_base = input_base;
	uint32_t *blk = nullptr, *derr = nullptr, *llen = nullptr, *keep = nullptr, *klen = nullptr;
	uint64_t *bbase = nullptr, *lstart = nullptr, *kidx = nullptr, *boff = nullptr;
	auto cleanup = [&]() { hipFree(blk); hipFree(derr); hipFree(llen); hipFree(keep); hipFree(klen); hipFree(bbase); hipFree(lstart); hipFree(kidx); hipFree(boff); };
#define ING(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hip_err_text(e_); cleanup(); kmr_reads_free(R); \
	return e_ == hipErrorOutOfMemory ? KMR_ERR_OOM : KMR_ERR_HIP; } } while (0)
#define INGRC(expr) do { int rc_ = (expr); if (rc_) { cleanup(); kmr_reads_free(R); return rc_; } } while (0)
	const uint64_t nblk = (len + (uint64_t)ING_THREADS * ING_BYTES - 1) / ((uint64_t)ING_THREADS * ING_BYTES);
	uint64_t n_lines = 0;
	ING(dev_malloc((void **)&derr, 8)); ING(hipMemsetAsync(derr, 0, 8, h->stream));
	if (nblk) {
		if (nblk > 0x7fffffffull) { cleanup(); kmr_reads_free(R); return fail(h, KMR_ERR_INVALID_ARG, "FASTQ block too large for one call"); }
		ING(dev_malloc((void **)&blk, 4 * nblk)); ING(dev_malloc((void **)&bbase, 8 * (nblk + 1)));
		hipLaunchKernelGGL(ingest_count_lines, dim3((unsigned)nblk), dim3(ING_THREADS), 0, h->stream, text, len, blk);
		ING(hipGetLastError());
		INGRC(exclusive_scan(h, blk, nblk, bbase));
		ING(hipMemcpy(&n_lines, bbase + nblk, 8, hipMemcpyDeviceToHost));
	}
	if (n_lines % 4 != 0) { cleanup(); kmr_reads_free(R); return fail(h, KMR_ERR_INVALID_ARG, "malformed FASTQ: " + std::to_string(n_lines) + " non-empty lines is not a multiple of 4 (truncated record)"); }
	const uint64_t nrec = n_lines / 4;
	uint64_t n_kept = 0, total = 0;
	if (nrec) {
		ING(dev_malloc((void **)&lstart, 8 * n_lines)); ING(dev_malloc((void **)&llen, 4 * n_lines));
		hipLaunchKernelGGL(ingest_index_lines, dim3((unsigned)nblk), dim3(ING_THREADS), 0, h->stream, text, len, bbase, lstart);
		hipLaunchKernelGGL(ingest_line_lengths, dim3(grid_for(n_lines)), dim3(256), 0, h->stream, text, len, lstart, n_lines, llen, derr);
		ING(dev_malloc((void **)&keep, 4 * nrec)); ING(dev_malloc((void **)&klen, 4 * nrec));
		ING(dev_malloc((void **)&kidx, 8 * (nrec + 1))); ING(dev_malloc((void **)&boff, 8 * (nrec + 1)));
		hipLaunchKernelGGL(ingest_records, dim3(grid_for(nrec)), dim3(256), 0, h->stream, text, lstart, llen, nrec, store_comment, keep, klen, derr);
		ING(hipGetLastError());
		INGRC(exclusive_scan(h, keep, nrec, kidx));
		INGRC(exclusive_scan(h, klen, nrec, boff));
		uint32_t e = 0;
		ING(hipMemcpy(&e, derr, 4, hipMemcpyDeviceToHost));
		if (e) {
			std::string why;
			if (e & ING_ERR_NAME) why += " a record does not start with '@' or has an empty name;";
			if (e & ING_ERR_BLANK) why += " an empty line inside a record;";
			if (e & ING_ERR_PLUS) why += " missing '+' line;";
			if (e & ING_ERR_LEN) why += " number of bases and quals not equal;";
			cleanup(); kmr_reads_free(R);
			return fail(h, KMR_ERR_INVALID_ARG, "malformed FASTQ:" + why);
		}
		ING(hipMemcpy(&n_kept, kidx + nrec, 8, hipMemcpyDeviceToHost)); ING(hipMemcpy(&total, boff + nrec, 8, hipMemcpyDeviceToHost));
	}
	R->n = n_kept; R->total = total; R->filtered = nrec - n_kept;
	ING(dev_malloc((void **)&R->bases, total + 64)); ING(dev_malloc((void **)&R->quals, total + 64)); ING(dev_malloc((void **)&R->offsets, 8 * (n_kept + 1)));
	ING(dev_malloc((void **)&R->name_off, 8 * std::max<uint64_t>(1, n_kept))); ING(dev_malloc((void **)&R->name_len, 4 * std::max<uint64_t>(1, n_kept)));
	ING(hipMemsetAsync(R->bases + total, 0, 64, h->stream)); ING(hipMemsetAsync(R->quals + total, 0, 64, h->stream));
	ING(hipMemcpyAsync(R->offsets + n_kept, &total, 8, hipMemcpyHostToDevice, h->stream));
	if (nrec) {
		/* appendFasta rescales every read from the input base to Read::FASTQ_START_CHAR as it is read (src/ReadSet.cpp:324,336) */
		hipLaunchKernelGGL(ingest_copy, dim3(grid_for(nrec, 4, 1 << 16)), dim3(256), 0, h->stream, text, len, lstart, llen, nrec, keep, kidx, boff,
		                   (int)start - (int)input_base, start, R->bases, R->quals, R->offsets, R->name_off, R->name_len, derr + 1);
		ING(hipGetLastError());
		uint32_t flip = 0;
		ING(hipMemcpyAsync(&flip, derr + 1, 4, hipMemcpyDeviceToHost, h->stream)); ING(hipStreamSynchronize(h->stream));
		const uint32_t want = start == 33 ? 64u : 33u;        /* __setFastqStart(the other base), src/ReadSet.h:174-186 */
		if (flip && want != input_base) {
			if (total) hipLaunchKernelGGL(ingest_shift_quals, dim3(grid_for(total)), dim3(256), 0, h->stream, R->quals, total, (int)input_base - (int)want);
			ING(hipGetLastError());
			R->input_base = want;
		}
	}
	ING(hipStreamSynchronize(h->stream));
#undef ING
#undef INGRC
	cleanup();
	*out = R;
	return KMR_OK;
}

int kmr_ingest_fastq_dev(kmr_handle *h, const void *dev_text, uint64_t len, uint32_t input_quality_base, int store_comment, kmr_reads **out) {
	if (!h || !out || (len && !dev_text)) return KMR_ERR_INVALID_ARG;
	*out = nullptr;
	hipSetDevice(h->device);
	return ingest_dev(h, (const uint8_t *)dev_text, len, input_quality_base, store_comment, out);
}
int kmr_ingest_fastq(kmr_handle *h, const char *text, uint64_t len, uint32_t input_quality_base, int store_comment, kmr_reads **out) {
	if (!h || !out || (len && !text)) return KMR_ERR_INVALID_ARG;
	*out = nullptr;
	hipSetDevice(h->device);
	uint8_t *d = nullptr;
	HIPCHK(h, dev_malloc((void **)&d, len + 16));
	hipError_t e = hipMemcpy(d, text, len, hipMemcpyHostToDevice);
	if (e != hipSuccess) { hipFree(d); h->err = std::string("hipMemcpy(FASTQ text): ") + hipGetErrorString(e); return KMR_ERR_HIP; }
	const int rc = ingest_dev(h, d, len, input_quality_base, store_comment, out);
	hipFree(d);
	return rc;
}
/* a device-resident batch from reads the host already parsed (the reference's ReadSet flattened as for kmr_add_reads) */
int kmr_reads_from_host(kmr_handle *h, const char *bases, const char *quals, const uint64_t *offsets, uint64_t n_reads, kmr_reads **out) {
	if (!h || !out || !offsets || (n_reads && (!bases || !quals))) return KMR_ERR_INVALID_ARG;
	*out = nullptr;
	hipSetDevice(h->device);
	const uint64_t first = offsets[0], total = offsets[n_reads] - first;
	std::unique_ptr<kmr_reads, void (*)(kmr_reads *)> r(new kmr_reads, kmr_reads_free);
	r->device = h->device; r->n = n_reads; r->total = total; r->input_base = h->cfg.fastq_start_char;
	HIPCHK(h, dev_malloc((void **)&r->bases, total + 64)); HIPCHK(h, dev_malloc((void **)&r->quals, total + 64));
	HIPCHK(h, dev_malloc((void **)&r->offsets, 8 * (n_reads + 1)));
	HIPCHK(h, dev_malloc((void **)&r->name_off, 8 * std::max<uint64_t>(n_reads, 1))); HIPCHK(h, dev_malloc((void **)&r->name_len, 4 * std::max<uint64_t>(n_reads, 1)));
	HIPCHK(h, hipMemset(r->bases + total, 0, 64)); HIPCHK(h, hipMemset(r->quals + total, 0, 64));
	HIPCHK(h, hipMemset(r->name_off, 0, 8 * std::max<uint64_t>(n_reads, 1))); HIPCHK(h, hipMemset(r->name_len, 0, 4 * std::max<uint64_t>(n_reads, 1)));
	if (total) { HIPCHK(h, hipMemcpy(r->bases, bases + first, total, hipMemcpyHostToDevice)); HIPCHK(h, hipMemcpy(r->quals, quals + first, total, hipMemcpyHostToDevice)); }
	std::vector<uint64_t> rel(n_reads + 1);
	for (uint64_t i = 0; i <= n_reads; i++) rel[i] = offsets[i] - first;
	HIPCHK(h, hipMemcpy(r->offsets, rel.data(), 8 * (n_reads + 1), hipMemcpyHostToDevice));
	*out = r.release();
	return KMR_OK;
}
void kmr_reads_free(kmr_reads *r) {
	if (!r) return;
	hipSetDevice(r->device);
	if (r->bases) hipFree(r->bases); if (r->quals) hipFree(r->quals); if (r->offsets) hipFree(r->offsets);
	if (r->name_off) hipFree(r->name_off); if (r->name_len) hipFree(r->name_len);
	delete r;
}
int kmr_reads_info(const kmr_reads *r, uint64_t *n_reads, uint64_t *total_bases, uint32_t *input_quality_base, uint64_t *n_filtered) {
	if (!r) return KMR_ERR_INVALID_ARG;
	if (n_reads) *n_reads = r->n; if (total_bases) *total_bases = r->total;
	if (input_quality_base) *input_quality_base = r->input_base; if (n_filtered) *n_filtered = r->filtered;
	return KMR_OK;
}
int kmr_reads_device_ptrs(const kmr_reads *r, void **bases, void **quals, void **offsets) {
	if (!r) return KMR_ERR_INVALID_ARG;
	if (bases) *bases = r->bases; if (quals) *quals = r->quals; if (offsets) *offsets = r->offsets;
	return KMR_OK;
}
int kmr_reads_copy(const kmr_reads *r, char *bases, char *quals, uint64_t *offsets, uint64_t *name_off, uint32_t *name_len) {
	if (!r) return KMR_ERR_INVALID_ARG;
	hipSetDevice(r->device);
	hipError_t e = hipSuccess;
	if (bases && r->total && e == hipSuccess) e = hipMemcpy(bases, r->bases, r->total, hipMemcpyDeviceToHost);
	if (quals && r->total && e == hipSuccess) e = hipMemcpy(quals, r->quals, r->total, hipMemcpyDeviceToHost);
	if (offsets && e == hipSuccess) e = hipMemcpy(offsets, r->offsets, 8 * (r->n + 1), hipMemcpyDeviceToHost);
	if (name_off && r->n && e == hipSuccess) e = hipMemcpy(name_off, r->name_off, 8 * r->n, hipMemcpyDeviceToHost);
	if (name_len && r->n && e == hipSuccess) e = hipMemcpy(name_len, r->name_len, 4 * r->n, hipMemcpyDeviceToHost);
	return e == hipSuccess ? KMR_OK : KMR_ERR_HIP;
}
int kmr_add_read_batch(kmr_handle *h, const kmr_reads *r, uint64_t first_global_read_idx) {
	if (!h || !r) return KMR_ERR_INVALID_ARG;
	if (r->device != h->device) return fail(h, KMR_ERR_INVALID_ARG, "read batch lives on another device");
	int rc = kmr_add_reads_dev(h, r->bases, r->quals, r->offsets, r->n, r->total, first_global_read_idx, nullptr);
	if (!rc) rc = kmr_sync(h);
	return rc;
}

/* ---- f4: artifact filter (FilterKnownOddities) --------------------------- */
}  // extern "C"

struct kmr_artifact_filter {
	int device = 0;
	kmr_artifact_config cfg;
	uint32_t n_seq = 0, remaining_edits = 0, log2cap = 0;
	uint64_t n_keys = 0;
	uint64_t *d_keys = nullptr; uint32_t *d_vals = nullptr;      /* open-addressed lookup table */
	uint32_t *d_bits = nullptr;                                    /* presence filter in front of it (ART_FILTER_LOG2 bits) */
	std::vector<uint64_t> keys; std::vector<uint32_t> vals;        /* the same entries on the host, ascending keys */
};

namespace {

uint32_t art_log2cap(uint64_t n) { uint32_t l = 10; while ((1ull << l) < 2 * n + 16) l++; return l; }

struct ArtBuf {          /* device scratch of one call, released on every exit path */
	std::vector<void *> p;
	~ArtBuf() { for (void *q : p) if (q) hipFree(q); }
	template <typename T> hipError_t get(T **out, size_t n) { void *q = nullptr; hipError_t e = dev_malloc(&q, std::max<size_t>(sizeof(T) * n, 256)); if (e == hipSuccess) p.push_back(q); *out = (T *)q; return e; }
};

uint64_t art_pack(const char *s, uint32_t len) {      /* TwoBitSequence::compressSequence: anything but ACGT packs as A */
	uint64_t v = 0;
	for (uint32_t i = 0; i < len; i++) { const char c = s[i]; v = (v << 2) | (uint64_t)((c == 'C') ? 1 : (c == 'G') ? 2 : (c == 'T') ? 3 : 0); }
	return v;
}
uint64_t art_revcomp_host(uint64_t v, uint32_t len) { uint64_t r = 0; for (uint32_t i = 0; i < len; i++) { r = (r << 2) | (3 - (v & 3)); v >>= 2; } return r; }

/* (re)build the lookup table of the filter from its host entries */
int art_upload(kmr_handle *h, kmr_artifact_filter *f) {
	if (f->d_keys) { hipFree(f->d_keys); f->d_keys = nullptr; } if (f->d_vals) { hipFree(f->d_vals); f->d_vals = nullptr; }
	f->n_keys = f->keys.size();
	f->log2cap = art_log2cap(f->n_keys);
	const uint64_t cap = 1ull << f->log2cap;
	HIPCHK(h, dev_malloc((void **)&f->d_keys, 8 * cap)); HIPCHK(h, dev_malloc((void **)&f->d_vals, 4 * cap));
	if (!f->d_bits) HIPCHK(h, dev_malloc((void **)&f->d_bits, (1u << ART_FILTER_LOG2) / 8));
	HIPCHK(h, hipMemsetAsync(f->d_bits, 0, (1u << ART_FILTER_LOG2) / 8, h->stream));
	ArtBuf tmp; uint64_t *dk; uint32_t *dv;
	HIPCHK(h, tmp.get(&dk, f->n_keys)); HIPCHK(h, tmp.get(&dv, f->n_keys));
	if (f->n_keys) { HIPCHK(h, hipMemcpyAsync(dk, f->keys.data(), 8 * f->n_keys, hipMemcpyHostToDevice, h->stream)); HIPCHK(h, hipMemcpyAsync(dv, f->vals.data(), 4 * f->n_keys, hipMemcpyHostToDevice, h->stream)); }
	ArtifactTable t{f->d_keys, f->d_vals, nullptr, f->log2cap, f->d_bits};
	hipLaunchKernelGGL(artifact_fill, dim3(1024), dim3(256), 0, h->stream, f->d_keys, (uint32_t *)nullptr, cap);
	if (f->n_keys) hipLaunchKernelGGL(artifact_insert, dim3((unsigned)std::min<uint64_t>((f->n_keys + 255) / 256, 4096)), dim3(256), 0, h->stream, t, dk, dv, f->n_keys);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipStreamSynchronize(h->stream));
	return KMR_OK;
}

/* one round of prepareMaps' edit loop (src/FilterKnownOddities.h:264-282) on the device */
int art_build_round(kmr_handle *h, kmr_artifact_filter *f) {
	const uint32_t L = f->cfg.match_length, kb = L / 4;
	const uint64_t n = f->keys.size();
	/* the map's iteration order: bucket by bucket (KmerMap(512*1024): 512*1024/32+1 buckets rounded up to 2^15), sorted inside */
	const uint64_t mask = resize_buckets(512 * 1024 / 32 + 1) - 1;
	std::vector<std::pair<uint64_t, uint64_t>> order(n);
	for (uint64_t i = 0; i < n; i++) {
		uint8_t b[8];
		for (uint32_t j = 0; j < kb; j++) b[j] = (uint8_t)(f->keys[i] >> (8 * (kb - 1 - j)));
		order[i] = std::make_pair(kmr_hash(b, kb) & mask, i);
	}
	std::sort(order.begin(), order.end());       /* ties inside a bucket: ascending index = ascending key */
	std::vector<uint64_t> sk(n); std::vector<uint32_t> sv(n);
	for (uint64_t i = 0; i < n; i++) { sk[i] = f->keys[order[i].second]; sv[i] = f->vals[order[i].second]; }
	const uint64_t worst = n * (3ull * L + 1);
	if (worst > (1ull << 32)) return fail(h, KMR_ERR_UNSUPPORTED, "artifact filter: an edit round over " + std::to_string(n) + " keys does not fit the build table");
	const uint32_t log2cap = art_log2cap(worst);
	const uint64_t cap = 1ull << log2cap;
	ArtBuf tmp; uint64_t *tk, *dk, *ok; uint32_t *tv, *tr, *dv, *ov; unsigned long long *cnt;
	HIPCHK(h, tmp.get(&tk, cap)); HIPCHK(h, tmp.get(&tv, cap)); HIPCHK(h, tmp.get(&tr, cap));
	HIPCHK(h, tmp.get(&dk, n)); HIPCHK(h, tmp.get(&dv, n)); HIPCHK(h, tmp.get(&cnt, 1));
	HIPCHK(h, hipMemcpyAsync(dk, sk.data(), 8 * n, hipMemcpyHostToDevice, h->stream));
	HIPCHK(h, hipMemcpyAsync(dv, sv.data(), 4 * n, hipMemcpyHostToDevice, h->stream));
	HIPCHK(h, hipMemsetAsync(cnt, 0, 8, h->stream));
	ArtifactTable t{tk, tv, tr, log2cap, nullptr};
	hipLaunchKernelGGL(artifact_fill, dim3(2048), dim3(256), 0, h->stream, tk, tr, cap);
	hipLaunchKernelGGL(artifact_insert, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, h->stream, t, dk, dv, n);
	hipLaunchKernelGGL(artifact_neighbours, dim3((unsigned)std::min<uint64_t>((n * L + 255) / 256, 1u << 20)), dim3(256), 0, h->stream, t, dk, n, L);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipStreamSynchronize(h->stream));
	/* the table is sparse (<= 50 % by construction, a few % in practice): count first, then compact */
	HIPCHK(h, tmp.get(&ok, worst)); HIPCHK(h, tmp.get(&ov, worst));
	hipLaunchKernelGGL(artifact_compact, dim3(2048), dim3(256), 0, h->stream, t, dv, ok, ov, cnt);
	HIPCHK(h, hipGetLastError());
	unsigned long long m = 0;
	HIPCHK(h, hipMemcpyAsync(&m, cnt, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	std::vector<uint64_t> nk(m); std::vector<uint32_t> nv(m);
	HIPCHK(h, hipMemcpy(nk.data(), ok, 8 * m, hipMemcpyDeviceToHost)); HIPCHK(h, hipMemcpy(nv.data(), ov, 4 * m, hipMemcpyDeviceToHost));
	std::vector<uint64_t> idx(m);
	for (uint64_t i = 0; i < m; i++) idx[i] = i;
	std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return nk[a] < nk[b]; });
	f->keys.resize(m); f->vals.resize(m);
	for (uint64_t i = 0; i < m; i++) { f->keys[i] = nk[idx[i]]; f->vals[i] = nv[idx[i]]; }
	return KMR_OK;
}

}  // namespace

extern "C" {

void kmr_artifact_config_init(kmr_artifact_config *c) {
	if (!c) return;
	memset(c, 0, sizeof(*c));
	c->match_length = 24; c->edit_distance = 2; c->build_edits = 2;      /* _FilterKnownOdditiesOptions(), src/FilterKnownOddities.h:72-75 */
	c->min_quality = 3; c->fastq_start_char = 33; c->min_read_length = 0.40f;
}

int kmr_artifact_filter_create(kmr_handle *h, const kmr_artifact_config *cfg, const char *fasta, uint64_t len, kmr_artifact_filter **out) {
	if (!h || !cfg || !out || (len && !fasta)) return KMR_ERR_INVALID_ARG;
	*out = nullptr;
	if (cfg->match_length == 0 || cfg->match_length > 28 || (cfg->match_length & 3))
		return fail(h, KMR_ERR_INVALID_ARG, "artifact match length must be a multiple of 4 and <= 28 (src/FilterKnownOddities.h:207-209,244)");
	hipSetDevice(h->device);
	std::unique_ptr<kmr_artifact_filter, void (*)(kmr_artifact_filter *)> f(new kmr_artifact_filter, kmr_artifact_filter_free);
	f->device = h->device; f->cfg = *cfg;
	const uint32_t L = cfg->match_length;
	/* sequences: read 0 is the empty "no match" read, then the FASTA records in file order (:213-231) */
	std::vector<std::string> seqs(1);
	for (uint64_t i = 0; i < len;) {
		uint64_t e = i; while (e < len && fasta[e] != '\n') e++;
		uint64_t le = e; if (le > i && fasta[le - 1] == '\r') le--;
		if (le > i) {
			if (fasta[i] == '>') seqs.push_back(std::string());
			else if (seqs.size() > 1) for (uint64_t j = i; j < le; j++) seqs.back().push_back((char)toupper((unsigned char)fasta[j]));
		}
		i = e + 1;
	}
	f->n_seq = (uint32_t)seqs.size();
	std::vector<std::pair<uint64_t, uint32_t>> kv;
	for (uint32_t s = 1; s < f->n_seq; s++) {
		std::string q = seqs[s];
		if (cfg->reference_begin == 0 || s < cfg->reference_begin) q += seqs[s].substr(0, L);      /* ReadSet::circularize, src/ReadSet.cpp:120-130 */
		for (size_t j = 0; j + L <= q.size(); j++) {
			const uint64_t v = art_pack(q.data() + j, L), r = art_revcomp_host(v, L);
			kv.push_back(std::make_pair(r < v ? r : v, s));
		}
	}
	std::sort(kv.begin(), kv.end());                  /* getOrSetElement in sequence order: the lowest sequence index keeps a key */
	for (size_t i = 0; i < kv.size(); i++) if (i == 0 || kv[i].first != kv[i - 1].first) { f->keys.push_back(kv[i].first); f->vals.push_back(kv[i].second); }
	int edits = (int)cfg->edit_distance;
	const int maxErrors = edits;
	for (int error = 0; error < maxErrors; error++) {
		if (cfg->build_edits == 1 || (cfg->build_edits == 2 && f->keys.size() < 750000)) {
			edits--;
			if (!f->keys.empty()) { const int rc = art_build_round(h, f.get()); if (rc) return rc; }
		}
	}
	if (edits > 2) return fail(h, KMR_ERR_UNSUPPORTED, "artifact filter: more than two edits left for query time");
	f->remaining_edits = (uint32_t)edits;
	const int rc = art_upload(h, f.get());
	if (rc) return rc;
	*out = f.release();
	return KMR_OK;
}
int kmr_artifact_filter_info(const kmr_artifact_filter *f, uint64_t *n_sequences, uint64_t *n_filter_kmers, uint32_t *remaining_edits) {
	if (!f) return KMR_ERR_INVALID_ARG;
	if (n_sequences) *n_sequences = f->n_seq; if (n_filter_kmers) *n_filter_kmers = f->n_keys; if (remaining_edits) *remaining_edits = f->remaining_edits;
	return KMR_OK;
}
int kmr_artifact_filter_entries(const kmr_artifact_filter *f, uint64_t *keys, uint32_t *values, uint64_t cap) {
	if (!f) return KMR_ERR_INVALID_ARG;
	if (cap < f->keys.size()) return KMR_ERR_CAPACITY;
	if (keys) memcpy(keys, f->keys.data(), 8 * f->keys.size());
	if (values) memcpy(values, f->vals.data(), 4 * f->vals.size());
	return KMR_OK;
}
void kmr_artifact_filter_free(kmr_artifact_filter *f) {
	if (!f) return;
	hipSetDevice(f->device);
	if (f->d_keys) hipFree(f->d_keys); if (f->d_vals) hipFree(f->d_vals); if (f->d_bits) hipFree(f->d_bits);
	delete f;
}

int kmr_artifact_filter_apply(kmr_handle *h, const kmr_artifact_filter *f, const kmr_reads *in, const int64_t *mate,
                              uint32_t *value, uint32_t *min_pass, uint32_t *max_pass, uint8_t *action,
                              uint32_t *remnant_off, uint32_t *remnant_len, kmr_reads **out) {
	if (!h || !f || !in) return KMR_ERR_INVALID_ARG;
	if (out) *out = nullptr;
	if (f->device != h->device || in->device != h->device) return fail(h, KMR_ERR_INVALID_ARG, "filter, reads and handle must live on one device");
	hipSetDevice(h->device);
	const uint64_t n = in->n;
	ArtifactTable t{f->d_keys, f->d_vals, nullptr, f->log2cap, f->d_bits};
	ArtifactParams P;
	P.length = f->cfg.match_length; P.nSeq = f->n_seq; P.numErrors = f->remaining_edits;
	P.srBegin = f->cfg.simple_repeat_begin; P.srEnd = f->cfg.simple_repeat_end; P.phix = f->cfg.phix_idx; P.refBegin = f->cfg.reference_begin;
	P.minQualChar = (int32_t)(int8_t)(uint8_t)(f->cfg.fastq_start_char + f->cfg.min_quality);
	P.minReadLength = f->cfg.min_read_length;
	ArtBuf tmp;
	uint32_t *dval, *dmin, *dmax, *dro, *drl, *dlen, *dflag; uint8_t *dact; int64_t *dmate = nullptr; uint64_t *dridx, *dsrc;
	HIPCHK(h, tmp.get(&dval, n)); HIPCHK(h, tmp.get(&dmin, n)); HIPCHK(h, tmp.get(&dmax, n)); HIPCHK(h, tmp.get(&dro, n)); HIPCHK(h, tmp.get(&drl, n));
	HIPCHK(h, tmp.get(&dact, n)); HIPCHK(h, tmp.get(&dflag, n)); HIPCHK(h, tmp.get(&dridx, n + 1));
	if (mate && n) { HIPCHK(h, tmp.get(&dmate, n)); HIPCHK(h, hipMemcpyAsync(dmate, mate, 8 * n, hipMemcpyHostToDevice, h->stream)); }
	uint64_t n_rem = 0;
	if (n) {
		const unsigned blocks = (unsigned)((n + 255) / 256);
		hipLaunchKernelGGL(artifact_screen, dim3(blocks), dim3(256), 0, h->stream, in->bases, in->quals, in->offsets, n, t, P, dval, dmin, dmax, dro, drl);
		HIPCHK(h, hipGetLastError());
	}
	/* lengths after the filter: n reads, then the remnants */
	HIPCHK(h, tmp.get(&dlen, 2 * n + 1));
	if (n) {
		const unsigned blocks = (unsigned)((n + 255) / 256);
		hipLaunchKernelGGL(artifact_action, dim3(blocks), dim3(256), 0, h->stream, in->offsets, n, dmate, P, dval, dmin, dmax, drl, dact, dlen, dflag);
		HIPCHK(h, hipGetLastError());
		int rc = exclusive_scan(h, dflag, n, dridx); if (rc) return rc;
		HIPCHK(h, hipMemcpy(&n_rem, dridx + n, 8, hipMemcpyDeviceToHost));
	}
	HIPCHK(h, tmp.get(&dsrc, n_rem));
	if (n_rem) {
		hipLaunchKernelGGL(artifact_remnants, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, n, drl, dridx, dlen, dsrc);
		HIPCHK(h, hipGetLastError());
	}
	HIPCHK(h, hipStreamSynchronize(h->stream));
	hipError_t e = hipSuccess;
	if (n) {
		if (value && e == hipSuccess) e = hipMemcpy(value, dval, 4 * n, hipMemcpyDeviceToHost);
		if (min_pass && e == hipSuccess) e = hipMemcpy(min_pass, dmin, 4 * n, hipMemcpyDeviceToHost);
		if (max_pass && e == hipSuccess) e = hipMemcpy(max_pass, dmax, 4 * n, hipMemcpyDeviceToHost);
		if (action && e == hipSuccess) e = hipMemcpy(action, dact, n, hipMemcpyDeviceToHost);
		if (remnant_off && e == hipSuccess) e = hipMemcpy(remnant_off, dro, 4 * n, hipMemcpyDeviceToHost);
		if (remnant_len && e == hipSuccess) e = hipMemcpy(remnant_len, drl, 4 * n, hipMemcpyDeviceToHost);
	}
	HIPCHK(h, e);
	if (!out) return KMR_OK;
	const uint64_t n_out = n + n_rem;
	std::unique_ptr<kmr_reads, void (*)(kmr_reads *)> r(new kmr_reads, kmr_reads_free);
	r->device = h->device; r->n = n_out; r->input_base = in->input_base; r->filtered = in->filtered;
	HIPCHK(h, dev_malloc((void **)&r->offsets, 8 * (n_out + 1)));
	if (n_out) { int rc = exclusive_scan(h, dlen, n_out, r->offsets); if (rc) return rc; HIPCHK(h, hipMemcpy(&r->total, r->offsets + n_out, 8, hipMemcpyDeviceToHost)); }
	else HIPCHK(h, hipMemset(r->offsets, 0, 8));
	HIPCHK(h, dev_malloc((void **)&r->bases, r->total + 64)); HIPCHK(h, dev_malloc((void **)&r->quals, r->total + 64));
	HIPCHK(h, hipMemsetAsync(r->bases + r->total, 0, 64, h->stream)); HIPCHK(h, hipMemsetAsync(r->quals + r->total, 0, 64, h->stream));
	HIPCHK(h, dev_malloc((void **)&r->name_off, 8 * std::max<uint64_t>(n_out, 1))); HIPCHK(h, dev_malloc((void **)&r->name_len, 4 * std::max<uint64_t>(n_out, 1)));
	if (n_out) {
		hipLaunchKernelGGL(artifact_gather, dim3((unsigned)std::min<uint64_t>((n_out + 3) / 4, 1u << 16)), dim3(256), 0, h->stream,
		                   in->bases, in->quals, in->offsets, in->name_off, in->name_len, n, n_out, dact, dmin, dro, dsrc, r->offsets, r->bases, r->quals, r->name_off, r->name_len);
		HIPCHK(h, hipGetLastError());
	}
	HIPCHK(h, hipStreamSynchronize(h->stream));
	*out = r.release();
	return KMR_OK;
}

/* ---- f2: the batch as 2-bit packed reads + markups -------------------------- */
int kmr_reads_twobit(kmr_handle *h, const kmr_reads *r, uint8_t *twobit, uint64_t twobit_capacity, uint64_t *twobit_offsets,
                     uint32_t *markup_pos, char *markup_char, uint64_t markup_capacity, uint64_t *markup_offsets,
                     uint64_t *twobit_bytes, uint64_t *n_markups) {
	if (!h || !r) return KMR_ERR_INVALID_ARG;
	if (r->device != h->device) return fail(h, KMR_ERR_INVALID_ARG, "read batch lives on another device");
	hipSetDevice(h->device);
	const uint64_t n = r->n;
	ArtBuf tmp; uint32_t *dlen, *dcnt; uint64_t *dtb, *dmk;
	HIPCHK(h, tmp.get(&dlen, n + 1)); HIPCHK(h, tmp.get(&dcnt, n + 1)); HIPCHK(h, tmp.get(&dtb, n + 1)); HIPCHK(h, tmp.get(&dmk, n + 1));
	uint64_t tb_total = 0, mk_total = 0;
	if (n) {
		hipLaunchKernelGGL(twobit_count_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, r->bases, r->offsets, n, dlen, dcnt);
		HIPCHK(h, hipGetLastError());
		int rc = exclusive_scan(h, dlen, n, dtb); if (rc) return rc;
		rc = exclusive_scan(h, dcnt, n, dmk); if (rc) return rc;
		HIPCHK(h, hipMemcpy(&tb_total, dtb + n, 8, hipMemcpyDeviceToHost)); HIPCHK(h, hipMemcpy(&mk_total, dmk + n, 8, hipMemcpyDeviceToHost));
	} else { HIPCHK(h, hipMemset(dtb, 0, 8)); HIPCHK(h, hipMemset(dmk, 0, 8)); }
	if (twobit_bytes) *twobit_bytes = tb_total; if (n_markups) *n_markups = mk_total;
	if (!twobit && !markup_pos && !markup_char && !twobit_offsets && !markup_offsets) return KMR_OK;      /* sizes only */
	if ((twobit && twobit_capacity < tb_total) || ((markup_pos || markup_char) && markup_capacity < mk_total)) return KMR_ERR_CAPACITY;
	uint8_t *dtw, *dmc; uint32_t *dmp;
	HIPCHK(h, tmp.get(&dtw, tb_total)); HIPCHK(h, tmp.get(&dmp, mk_total)); HIPCHK(h, tmp.get(&dmc, mk_total));
	if (n) {
		hipLaunchKernelGGL(twobit_pack_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, r->bases, r->offsets, n, dtb, dmk, dtw, dmp, dmc);
		HIPCHK(h, hipGetLastError());
	}
	HIPCHK(h, hipStreamSynchronize(h->stream));
	hipError_t e = hipSuccess;
	if (twobit && tb_total) e = hipMemcpy(twobit, dtw, tb_total, hipMemcpyDeviceToHost);
	if (e == hipSuccess && twobit_offsets) e = hipMemcpy(twobit_offsets, dtb, 8 * (n + 1), hipMemcpyDeviceToHost);
	if (e == hipSuccess && markup_pos && mk_total) e = hipMemcpy(markup_pos, dmp, 4 * mk_total, hipMemcpyDeviceToHost);
	if (e == hipSuccess && markup_char && mk_total) e = hipMemcpy(markup_char, dmc, mk_total, hipMemcpyDeviceToHost);
	if (e == hipSuccess && markup_offsets) e = hipMemcpy(markup_offsets, dmk, 8 * (n + 1), hipMemcpyDeviceToHost);
	HIPCHK(h, e);
	return KMR_OK;
}

int kmr_reads_from_twobit(kmr_handle *h, const uint8_t *twobit, const uint64_t *twobit_offsets, const uint64_t *offsets,
                          const uint64_t *markup_offsets, const uint32_t *markup_pos, const char *markup_char,
                          const char *quals, int uniform_quality, uint64_t n_reads, kmr_reads **out) {
	if (!h || !out || !offsets || !twobit_offsets || (n_reads && !twobit)) return KMR_ERR_INVALID_ARG;
	if (uniform_quality < 0 || uniform_quality > 255 || (quals && uniform_quality)) return fail(h, KMR_ERR_INVALID_ARG, "uniform_quality: 0, or the one quality character of a batch without a quality array");
	*out = nullptr;
	hipSetDevice(h->device);
	const uint64_t first = offsets[0], total = offsets[n_reads] - first, tbytes = twobit_offsets[n_reads] - twobit_offsets[0];
	const uint64_t nm = markup_offsets ? markup_offsets[n_reads] - markup_offsets[0] : 0;
	std::unique_ptr<kmr_reads, void (*)(kmr_reads *)> r(new kmr_reads, kmr_reads_free);
	r->device = h->device; r->n = n_reads; r->total = total; r->input_base = h->cfg.fastq_start_char;
	HIPCHK(h, dev_malloc((void **)&r->bases, total + 64)); HIPCHK(h, dev_malloc((void **)&r->quals, total + 64));
	HIPCHK(h, dev_malloc((void **)&r->offsets, 8 * (n_reads + 1)));
	HIPCHK(h, dev_malloc((void **)&r->name_off, 8 * std::max<uint64_t>(n_reads, 1))); HIPCHK(h, dev_malloc((void **)&r->name_len, 4 * std::max<uint64_t>(n_reads, 1)));
	HIPCHK(h, hipMemset(r->bases + total, 0, 64)); HIPCHK(h, hipMemset(r->quals + total, 0, 64));
	HIPCHK(h, hipMemset(r->name_off, 0, 8 * std::max<uint64_t>(n_reads, 1))); HIPCHK(h, hipMemset(r->name_len, 0, 4 * std::max<uint64_t>(n_reads, 1)));
	/* qualities: the array, the one character, or Read::REF_QUAL (a batch always has a quality array; REF_QUAL reads weigh 1) */
	if (quals) { if (total) HIPCHK(h, hipMemcpy(r->quals, quals + first, total, hipMemcpyHostToDevice)); }
	else HIPCHK(h, hipMemset(r->quals, uniform_quality ? uniform_quality : 127, total));
	if (!n_reads) { HIPCHK(h, hipMemset(r->offsets, 0, 8)); *out = r.release(); return KMR_OK; }
	ArtBuf tmp; uint8_t *dtb, *dmc = nullptr; uint64_t *dto, *doff, *dmo = nullptr; uint32_t *dmp = nullptr;
	HIPCHK(h, tmp.get(&dtb, tbytes + 64)); HIPCHK(h, tmp.get(&dto, n_reads + 1)); HIPCHK(h, tmp.get(&doff, n_reads + 1));
	std::vector<uint64_t> rel(n_reads + 1), trel(n_reads + 1), mrel(markup_offsets ? n_reads + 1 : 0);
	for (uint64_t i = 0; i <= n_reads; i++) { rel[i] = offsets[i] - first; trel[i] = twobit_offsets[i] - twobit_offsets[0]; if (markup_offsets) mrel[i] = markup_offsets[i] - markup_offsets[0]; }
	if (tbytes) HIPCHK(h, hipMemcpy(dtb, twobit + twobit_offsets[0], tbytes, hipMemcpyHostToDevice));
	HIPCHK(h, hipMemcpy(dto, trel.data(), 8 * (n_reads + 1), hipMemcpyHostToDevice)); HIPCHK(h, hipMemcpy(doff, rel.data(), 8 * (n_reads + 1), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(twobit_unpack_kernel, dim3((unsigned)std::min<uint64_t>((n_reads + 255) / 256, (uint64_t)num_cus(h) * 32)), dim3(256), 0, h->stream, (const uint8_t *)dtb, (const uint64_t *)dto, (const uint64_t *)doff, n_reads, r->bases, r->offsets);
	HIPCHK(h, hipGetLastError());
	if (nm) {
		HIPCHK(h, tmp.get(&dmo, n_reads + 1)); HIPCHK(h, tmp.get(&dmp, nm)); HIPCHK(h, tmp.get(&dmc, nm));
		HIPCHK(h, hipMemcpy(dmo, mrel.data(), 8 * (n_reads + 1), hipMemcpyHostToDevice));
		HIPCHK(h, hipMemcpy(dmp, markup_pos + markup_offsets[0], 4 * nm, hipMemcpyHostToDevice)); HIPCHK(h, hipMemcpy(dmc, markup_char + markup_offsets[0], nm, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(twobit_markup_kernel, dim3(grid_for(n_reads)), dim3(256), 0, h->stream, (const uint64_t *)dmo, (const uint32_t *)dmp, (const uint8_t *)dmc, (const uint64_t *)r->offsets, n_reads, r->bases);
		HIPCHK(h, hipGetLastError());
	}
	HIPCHK(h, hipStreamSynchronize(h->stream));
	*out = r.release();
	return KMR_OK;
}

/* ---- stateless helpers ------------------------------------------------- */
uint64_t kmr_hash(const uint8_t *key, uint32_t len) {
	if (!key || len == 0 || len > 32) return 0;
	Key<4> k; key_from_bytes<4>(k, key, len);
	return key_hash<4>(k, len);
}
uint64_t kmr_hash_of_kind(const uint8_t *key, uint32_t len, uint32_t hash_kind) {
	if (!key || len == 0 || len > 32 || hash_kind > KMR_HASH_LOOKUP8) return 0;
	Key<4> k; key_from_bytes<4>(k, key, len);
	return key_hash<4>(k, len | (hash_kind << 16));
}
uint64_t kmr_bucket_idx(uint64_t hash, uint64_t nb) { return hash & (nb - 1); }
uint32_t kmr_local_thread_id(uint64_t hash, uint64_t nb, uint32_t t) { return (nb > 1 && t > 1) ? (uint32_t)((hash & (nb - 1)) % t) : 0; }
uint32_t kmr_distributed_thread_id(uint64_t hash, uint32_t n) { return distributed_thread_id(hash, n); }

int64_t kmr_compress_sequence(const char *bases, uint64_t len, uint8_t *out, uint32_t *mpos, char *mchar, uint64_t mcap) {
	if (!bases) return KMR_ERR_INVALID_ARG;
	int64_t nm = 0;
	uint64_t offset = 0;
	while (offset < len) {
		uint8_t c = 0;
		for (int i = 6; i >= 0 && offset < len; i -= 2) {
			char b = bases[offset]; uint8_t code;
			switch (b) { case 'A': case 'a': code = 0; break; case 'C': case 'c': code = 1; break; case 'G': case 'g': code = 2; break; case 'T': case 't': code = 3; break;
			case '\0': len = offset; code = 255; break;
			default: if (b == '.') b = 'N'; if ((uint64_t)nm < mcap) { if (mpos) mpos[nm] = (uint32_t)offset; if (mchar) mchar[nm] = b; } nm++; code = 0; }
			if (code == 255) break;
			offset++;
			c |= code << i;
		}
		if (out) *out++ = c;
	}
	return nm;
}

int kmr_least_complement(const uint8_t *packed, uint32_t k, uint8_t *out) {
	if (!packed || !out || k < 1 || k > 128) return KMR_ERR_INVALID_ARG;
	const uint32_t kb = (k + 3) / 4;
	Roller<4> r; r.init(k);
	for (uint32_t p = 0; p < k; p++) r.push((packed[p >> 2] >> (6 - 2 * (p & 3))) & 3);
	const bool least = key_le<4>(r.fwd, r.rc);
	const Key<4> &c = least ? r.fwd : r.rc;
	for (uint32_t j = 0; j < kb; j++) out[j] = key_byte<4>(c, j);
	return least ? 1 : 0;
}

int kmr_extract_by_owner_dev(kmr_handle *h, const void *dev_bases, const void *dev_quals, const void *dev_offsets, uint64_t n_reads,
                             uint64_t total_bases, uint64_t first_global_read_idx, const void *dev_discarded,
                             void *dev_records, uint64_t seg_capacity, void *dev_seg_counts) {
	if (!h || !dev_bases || !dev_offsets || !dev_records || !dev_seg_counts) return KMR_ERR_INVALID_ARG;
	if (h->cfg.world_size > (uint32_t)OWNER_MAX) return fail(h, KMR_ERR_UNSUPPORTED, "owner exchange supports up to 8 ranks per node");
	if (h->superkmer_mode && h->auto_mode && !h->sk_state) h->superkmer_mode = false;      /* auto: the k-mer record exchange runs on the two-level partition */
	if (h->superkmer_mode) return fail(h, KMR_ERR_UNSUPPORTED, "k-mer records by owner are a build_mode 1 / 2 path");
	hipSetDevice(h->device);
	ReadsView rv; rv.bases = (const uint8_t *)dev_bases; rv.quals = (const uint8_t *)dev_quals; rv.offsets = (const uint64_t *)dev_offsets;
	rv.discarded = (const uint8_t *)dev_discarded; rv.n_reads = n_reads; rv.stream_base = h->stream_base; rv.first_read_idx = first_global_read_idx;
	rv.u_start = rv.u_end = rv.u_read = nullptr; rv.n_units = 0;
	HIPCHK(h, hipMemsetAsync(dev_seg_counts, 0, 8 * h->cfg.world_size, h->stream));
	int rc = 0;
#define REC(Wv, E) rc = extract_by_owner_t<Wv, E>(h, rv, total_bases, dev_records, seg_capacity, dev_seg_counts);
	switch (h->W) {
	case 1: if (h->ext) REC(1, true) else REC(1, false) break;
	case 2: if (h->ext) REC(2, true) else REC(2, false) break;
	case 3: if (h->ext) REC(3, true) else REC(3, false) break;
	default: if (h->ext) REC(4, true) else REC(4, false)
	}
#undef REC
	h->stream_base += total_bases; h->reads += n_reads;
	return rc;
}

/* ---- f1, distributed form: lookups of k-mers other ranks own (DistributedReadSelector, src/DistributedFunctions.h:809-1045) */
int kmr_lookup_requests_dev(kmr_handle *h, const void *dev_bases, const void *dev_offsets, uint64_t n_reads, uint64_t total_bases,
                            void *dev_keys, void *dev_pos, uint64_t seg_capacity, void *dev_seg_counts) {
	if (!h || !dev_bases || !dev_offsets || !dev_keys || !dev_pos || !dev_seg_counts) return KMR_ERR_INVALID_ARG;
	if (h->cfg.world_size > (uint32_t)OWNER_MAX) return fail(h, KMR_ERR_UNSUPPORTED, "owner exchange supports up to 8 ranks per node");
	if (total_bases >= (1ull << 32)) return fail(h, KMR_ERR_UNSUPPORTED, "kmr_lookup_requests_dev: positions are 32-bit, pass at most 2^32 - 1 bases per call");
	hipSetDevice(h->device);
	ReadsView rv; rv.bases = (const uint8_t *)dev_bases; rv.quals = nullptr; rv.offsets = (const uint64_t *)dev_offsets;
	rv.discarded = nullptr; rv.n_reads = n_reads; rv.stream_base = 0; rv.first_read_idx = 0;
	rv.u_start = rv.u_end = rv.u_read = nullptr; rv.n_units = 0;
	HIPCHK(h, hipMemsetAsync(dev_seg_counts, 0, 8 * h->cfg.world_size, h->stream));
	if (n_reads == 0) return KMR_OK;
	int rc = 0;
#define REC(Wv, E) rc = extract_by_owner_t<Wv, E>(h, rv, total_bases, dev_keys, seg_capacity, dev_seg_counts, (uint32_t *)dev_pos);
	switch (h->W) {
	case 1: if (h->ext) REC(1, true) else REC(1, false) break;
	case 2: if (h->ext) REC(2, true) else REC(2, false) break;
	case 3: if (h->ext) REC(3, true) else REC(3, false) break;
	default: if (h->ext) REC(4, true) else REC(4, false)
	}
#undef REC
	return rc;
}
int kmr_lookup_keys_dev(kmr_handle *h, const void *dev_keys, uint64_t n, void *dev_counts) {
	if (!h || (n && (!dev_keys || !dev_counts))) return KMR_ERR_INVALID_ARG;
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_lookup_keys_dev before kmr_finalize");
	if (n == 0) return KMR_OK;
	hipSetDevice(h->device);
	const uint32_t vw = h->ext ? 15 : 3;
#define LK(Wv) hipLaunchKernelGGL(lookup_words_kernel<Wv>, dim3(grid_for(n)), dim3(256), 0, h->stream, view_of<Wv>(h->weak, vw), lut_of<Wv>(h), (const uint64_t *)dev_keys, n, h->hkb, (uint32_t *)dev_counts)
	switch (h->W) { case 1: LK(1); break; case 2: LK(2); break; case 3: LK(3); break; default: LK(4); }
#undef LK
	HIPCHK(h, hipGetLastError());
	return KMR_OK;
}
int kmr_scatter_counts_dev(kmr_handle *h, const void *dev_counts, const void *dev_pos, uint64_t n, void *dev_position_counts) {
	if (!h || (n && (!dev_counts || !dev_pos || !dev_position_counts))) return KMR_ERR_INVALID_ARG;
	if (n == 0) return KMR_OK;
	hipSetDevice(h->device);
	hipLaunchKernelGGL(scatter_counts_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, (const uint32_t *)dev_counts, (const uint32_t *)dev_pos, n, (uint32_t *)dev_position_counts);
	HIPCHK(h, hipGetLastError());
	return KMR_OK;
}
int kmr_score_counts_dev(kmr_handle *h, const void *dev_bases, const void *dev_offsets, uint64_t n_reads, const void *dev_position_counts,
                         double minimum_kmer_score, int scoring_type, uint32_t *trim_offset, uint32_t *trim_length, float *score, uint8_t *was_trimmed) {
	if (!h || !dev_bases || !dev_offsets || !dev_position_counts || !trim_offset || !trim_length || !score || !was_trimmed) return KMR_ERR_INVALID_ARG;
	if (scoring_type < 0 || scoring_type > 4) return fail(h, KMR_ERR_INVALID_ARG, "bad scoring_type");
	if (n_reads == 0) return KMR_OK;
	hipSetDevice(h->device);
	uint32_t *dto, *dtl; float *dsc; uint8_t *dwt;
	HIPCHK(h, dev_malloc((void **)&dto, 4 * n_reads)); HIPCHK(h, dev_malloc((void **)&dtl, 4 * n_reads)); HIPCHK(h, dev_malloc((void **)&dsc, 4 * n_reads)); HIPCHK(h, dev_malloc((void **)&dwt, n_reads));
	/* k-mer i of read r sits at position offsets[r] + i: the offsets are their own count offsets */
	hipLaunchKernelGGL(score_reads_kernel, dim3((unsigned)std::min<uint64_t>(((n_reads + 63) / 64 + SC_WAVES - 1) / SC_WAVES, 1u << 16)), dim3(SC_WAVES * 64), 0, h->stream, (const uint8_t *)dev_bases, (const uint64_t *)dev_offsets, n_reads, h->k,
	                   (const uint32_t *)dev_position_counts, (const uint64_t *)dev_offsets, (float)minimum_kmer_score, scoring_type, dto, dtl, dsc, dwt);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(trim_offset, dto, 4 * n_reads, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(trim_length, dtl, 4 * n_reads, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(score, dsc, 4 * n_reads, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(was_trimmed, dwt, n_reads, hipMemcpyDeviceToHost, h->stream);
	hipStreamSynchronize(h->stream);
	hipFree(dto); hipFree(dtl); hipFree(dsc); hipFree(dwt);
	HIPCHK(h, e);
	return KMR_OK;
}

int kmr_insert_records_dev(kmr_handle *h, const void *dev_records, uint64_t n) {
	if (!h || (n && !dev_records)) return KMR_ERR_INVALID_ARG;
	if (h->finalized) return fail(h, KMR_ERR_STATE, "kmr_insert_records_dev after kmr_finalize");
	if (n == 0) return KMR_OK;
	hipSetDevice(h->device);
	if (h->superkmer_mode && h->auto_mode && !h->sk_state) h->superkmer_mode = false;
	if (h->superkmer_mode) return fail(h, KMR_ERR_UNSUPPORTED, "k-mer records are inserted by build_mode 1 / 2");
	if (h->partition_mode) { int prc = insert_records_partition(h, dev_records, n); h->stream_base += n; return prc; }
	int rc = ensure_capacity(h, n); if (rc) return rc;
	hipEvent_t a, b; time_begin(h, 0, &a, &b);
#define INS(Wv, E) hipLaunchKernelGGL((insert_records_kernel<Wv, E>), dim3(grid_for(n)), dim3(256), 0, h->stream, table_of<Wv>(h), (const uint32_t *)dev_records, n, dev_params(h), h->stream_base)
	switch (h->W) {
	case 1: if (h->ext) INS(1, true); else INS(1, false); break;
	case 2: if (h->ext) INS(2, true); else INS(2, false); break;
	case 3: if (h->ext) INS(3, true); else INS(3, false); break;
	default: if (h->ext) INS(4, true); else INS(4, false);
	}
#undef INS
	time_end(h, 0, a, b);
	HIPCHK(h, hipGetLastError());
	h->stream_base += n;
	return KMR_OK;
}

/* Host-buffer forms of the two halves, for a host whose exchange is MPI_Alltoallv over host memory (the shim's
 * GpuDistributedKmerSpectrum): records of a device-resident batch binned by owner and copied out owner after owner, and records
 * received from the other ranks staged in and inserted. */
int kmr_extract_by_owner_host(kmr_handle *h, const kmr_reads *batch, uint64_t first_global_read_idx, uint64_t *seg_counts, void *records, uint64_t capacity_bytes) {
	if (!h || !batch || !seg_counts) return KMR_ERR_INVALID_ARG;
	if (batch->device != h->device) return fail(h, KMR_ERR_INVALID_ARG, "read batch lives on another device");
	hipSetDevice(h->device);
	const uint32_t world = h->cfg.world_size, rb = KMR_RECORD_BYTES(h->k, h->cfg.value_kind);
	if (!(h->xo_dev && h->xo_batch == (const void *)batch && h->xo_first == first_global_read_idx)) {
		if (h->xo_dev) { hipFree(h->xo_dev); h->xo_dev = nullptr; }
		h->xo_counts.assign(world, 0); h->xo_batch = nullptr;
		unsigned long long *dcounts = nullptr;
		HIPCHK(h, dev_malloc((void **)&dcounts, 8 * world));
		const uint64_t upper = batch->total + 64;          /* k-mers <= bases */
		uint64_t segcap = std::min<uint64_t>(upper, upper / world + upper / (4 * world) + 4096);
		const uint64_t sb = h->stream_base, rd = h->reads;
		unsigned long long bad0 = 0; hipMemcpy(&bad0, &h->dstats->sender_bad, 8, hipMemcpyDeviceToHost);      /* a repeated attempt must not count the dropped k-mers twice */
		for (;;) {
			if (dev_malloc(&h->xo_dev, (size_t)world * segcap * rb) != hipSuccess) { hipFree(dcounts); h->xo_dev = nullptr; return fail(h, KMR_ERR_OOM, "owner segments"); }
			h->stream_base = sb; h->reads = rd;             /* a repeated attempt stamps the same ordinals */
			int rc = kmr_extract_by_owner_dev(h, batch->bases, batch->quals, batch->offsets, batch->n, batch->total, first_global_read_idx, nullptr, h->xo_dev, segcap, dcounts);
			if (!rc) rc = sync_state(h);
			if (rc == KMR_ERR_CAPACITY && segcap < upper) {      /* a skewed batch: one owner takes more than its share */
				uint32_t e = 0; hipMemcpy(&e, h->derr, 4, hipMemcpyDeviceToHost); e &= ~(uint32_t)ERR_SEGMENT_OVERFLOW; hipMemcpy(h->derr, &e, 4, hipMemcpyHostToDevice); hipMemcpy(&h->dstats->sender_bad, &bad0, 8, hipMemcpyHostToDevice);
				hipFree(h->xo_dev); h->xo_dev = nullptr;
				segcap = std::min<uint64_t>(upper, segcap * 2);
				continue;
			}
			if (rc) { hipFree(dcounts); hipFree(h->xo_dev); h->xo_dev = nullptr; return rc; }
			break;
		}
		hipError_t e = hipMemcpy(h->xo_counts.data(), dcounts, 8 * world, hipMemcpyDeviceToHost);
		hipFree(dcounts);
		HIPCHK(h, e);
		h->xo_segcap = segcap; h->xo_batch = batch; h->xo_first = first_global_read_idx;
	}
	uint64_t total = 0;
	for (uint32_t r = 0; r < world; r++) { seg_counts[r] = h->xo_counts[r]; total += h->xo_counts[r]; }
	if (!records) return KMR_OK;                           /* sizing call: the segments wait on the device */
	if (capacity_bytes < total * rb) return fail(h, KMR_ERR_CAPACITY, "record buffer too small");
	uint8_t *dst = (uint8_t *)records;
	for (uint32_t r = 0; r < world; r++) {
		if (h->xo_counts[r]) HIPCHK(h, hipMemcpy(dst, (const uint8_t *)h->xo_dev + (size_t)r * h->xo_segcap * rb, (size_t)h->xo_counts[r] * rb, hipMemcpyDeviceToHost));
		dst += (size_t)h->xo_counts[r] * rb;
	}
	hipFree(h->xo_dev); h->xo_dev = nullptr; h->xo_batch = nullptr;
	return KMR_OK;
}
int kmr_insert_records(kmr_handle *h, const void *host_records, uint64_t n) {
	if (!h || (n && !host_records)) return KMR_ERR_INVALID_ARG;
	if (n == 0) return KMR_OK;
	hipSetDevice(h->device);
	const size_t bytes = (size_t)n * KMR_RECORD_BYTES(h->k, h->cfg.value_kind);
	void *d = nullptr;
	HIPCHK(h, dev_malloc(&d, bytes));
	hipError_t e = hipMemcpy(d, host_records, bytes, hipMemcpyHostToDevice);
	int rc = e == hipSuccess ? kmr_insert_records_dev(h, d, n) : KMR_ERR_HIP;
	if (!rc) rc = sync_state(h); else hipStreamSynchronize(h->stream);
	hipFree(d);
	return rc;
}

/* ---- owner exchange of super-k-mer lists (build_mode 3, world_size > 1): see kmr_superkmer.hpp */
static int sk_exchange_ready(kmr_handle *h, const char *who) {
	if (!h->superkmer_mode) return fail(h, KMR_ERR_STATE, std::string(who) + ": the handle does not build super-k-mer lists (build_mode 3)");
	if (h->finalized) return fail(h, KMR_ERR_STATE, std::string(who) + " after kmr_finalize");
	if (h->cfg.world_size > SK_OWNER_MAX) return fail(h, KMR_ERR_UNSUPPORTED, "at most 64 ranks");
	if (!h->sk_exchange) return fail(h, KMR_ERR_STATE, std::string(who) + " without kmr_sk_exchange_begin (the reads of this handle were filtered by getDistributedThreadId)");
	return 0;
}
int kmr_sk_exchange_begin(kmr_handle *h) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (!h->superkmer_mode) return fail(h, KMR_ERR_STATE, "kmr_sk_exchange_begin: the handle does not build super-k-mer lists (build_mode 3)");
	if (h->cfg.world_size > SK_OWNER_MAX) return fail(h, KMR_ERR_UNSUPPORTED, "at most 64 ranks");
	if (h->sk_state && !h->sk_exchange && h->reads) return fail(h, KMR_ERR_STATE, "kmr_sk_exchange_begin after reads were added");
	h->sk_exchange = true;
	return KMR_OK;
}
static int sk_ensure_state(kmr_handle *h) {      /* a rank without reads still owns lists */
	if (h->sk_state) return 0;
	ReadsView rv; memset(&rv, 0, sizeof(rv));
	return add_reads_superkmer(h, rv, 0);
}
int kmr_sk_exchange_counts(kmr_handle *h, uint64_t *chunks, uint64_t *granules) {
	if (!h || !chunks || !granules) return KMR_ERR_INVALID_ARG;
	int rc = sk_exchange_ready(h, "kmr_sk_exchange_counts"); if (rc) return rc;
	hipSetDevice(h->device);
	rc = sk_ensure_state(h); if (rc) return rc;
	rc = sync_state(h); if (rc) return rc;
	const uint32_t world = h->cfg.world_size;
	const uint64_t nl = sk_list_count(h->sk_bits);
	unsigned int head = 0;
	HIPCHK(h, hipMemcpy(&head, h->l1.head, 4, hipMemcpyDeviceToHost));
	if (head > h->l1.cap) head = h->l1.cap;
	unsigned long long *d = nullptr;
	HIPCHK(h, dev_malloc((void **)&d, 16 * SK_OWNER_MAX)); HIPCHK(h, hipMemsetAsync(d, 0, 16 * SK_OWNER_MAX, h->stream));
	hipLaunchKernelGGL(sk_close_kernel, dim3(grid_for(nl)), dim3(256), 0, h->stream, h->sk_state, nl, h->l1.chunk_count, h->l1.cap);
	if (head) hipLaunchKernelGGL(sk_owner_count_kernel, dim3(grid_for(head)), dim3(256), 0, h->stream, h->l1.chunk_list, h->l1.chunk_count, head, world, d, d + SK_OWNER_MAX, h->xr_lo, h->xr_hi);
	hipError_t e = hipGetLastError();
	std::vector<unsigned long long> hv(2 * SK_OWNER_MAX, 0);
	if (e == hipSuccess) e = hipMemcpyAsync(hv.data(), d, 16 * SK_OWNER_MAX, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
	hipFree(d);
	HIPCHK(h, e);
	for (uint32_t r = 0; r < world; r++) { chunks[r] = hv[r]; granules[r] = hv[SK_OWNER_MAX + r]; }
	return KMR_OK;
}
int kmr_sk_exchange_pack_dev(kmr_handle *h, void *dev_data, void *dev_meta, const uint64_t *granule_offset, const uint64_t *chunk_offset) {
	if (!h || !dev_data || !dev_meta || !granule_offset || !chunk_offset) return KMR_ERR_INVALID_ARG;
	int rc = sk_exchange_ready(h, "kmr_sk_exchange_pack_dev"); if (rc) return rc;
	hipSetDevice(h->device);
	const uint32_t world = h->cfg.world_size;
	unsigned int head = 0;
	HIPCHK(h, hipMemcpy(&head, h->l1.head, 4, hipMemcpyDeviceToHost));
	if (head > h->l1.cap) head = h->l1.cap;
	unsigned long long *d = nullptr;
	HIPCHK(h, dev_malloc((void **)&d, 32 * SK_OWNER_MAX));
	std::vector<unsigned long long> hv(4 * SK_OWNER_MAX, 0);
	for (uint32_t r = 0; r < world; r++) { hv[r] = granule_offset[r]; hv[SK_OWNER_MAX + r] = chunk_offset[r]; }
	hipError_t e = hipMemcpyAsync(d, hv.data(), 32 * SK_OWNER_MAX, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess && head) {
		hipLaunchKernelGGL(sk_pack_kernel, dim3(grid_for((uint64_t)head * 64, 256, num_cus(h) * 8)), dim3(256), 0, h->stream, pool_view(h, h->l1), head, world, h->cfg.rank,
		                   d, d + SK_OWNER_MAX, d + 2 * SK_OWNER_MAX, d + 3 * SK_OWNER_MAX, (uint4 *)dev_data, (uint2 *)dev_meta, h->xr_lo, h->xr_hi);
		hipLaunchKernelGGL(sk_state_drop_kernel, dim3(grid_for(sk_list_count(h->sk_bits))), dim3(256), 0, h->stream, h->sk_state, sk_list_count(h->sk_bits), world, h->cfg.rank, h->xr_lo, h->xr_hi);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->stream);      /* hv and d go out of scope */
	hipFree(d);
	HIPCHK(h, e);
	return KMR_OK;
}
/* the part of the list space the next kmr_sk_exchange_counts / kmr_sk_exchange_pack_dev are about ([0, ~0) = all of it) */
int kmr_sk_exchange_range(kmr_handle *h, uint64_t list_lo, uint64_t list_hi) {
	if (!h || list_lo > list_hi) return KMR_ERR_INVALID_ARG;
	h->xr_lo = list_lo; h->xr_hi = list_hi;
	return KMR_OK;
}
int kmr_count_lists_prefix(kmr_handle *h, uint32_t min_depth, uint64_t list_hi) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (h->finalized) return fail(h, KMR_ERR_STATE, "kmr_count_lists_prefix after kmr_finalize");
	if (!h->superkmer_mode) return KMR_OK;      /* the other build modes have no lists: kmr_finalize counts */
	hipSetDevice(h->device);
	return count_prefix_superkmer(h, min_depth, list_hi);
}
/* One weight for every record of this rank's lists so far?  state = kind << 32 | weight bits, kind 0: no record yet, 1: one weight, 2: several
 * (records with weights of their own).  A sender's state travels with its chunk counts; the owner folds it in with
 * kmr_sk_exchange_peer_uniform and then takes the count pass's one-weight form when all agree, without looking at the records. */
int kmr_sk_exchange_uniform(kmr_handle *h, uint64_t *state) {
	if (!h || !state) return KMR_ERR_INVALID_ARG;
	*state = h->sk_uni_mixed ? (2ull << 32) : (h->sk_uni_w == SK_UNI_NONE ? 0ull : ((1ull << 32) | h->sk_uni_w));
	return KMR_OK;
}
int kmr_sk_exchange_peer_uniform(kmr_handle *h, uint64_t state) {
	if (!h) return KMR_ERR_INVALID_ARG;
	const uint32_t kind = (uint32_t)(state >> 32), w = (uint32_t)state;
	if (kind > 2) return fail(h, KMR_ERR_INVALID_ARG, "bad uniform-weight state");
	h->peers_declare = true;
	if (kind == 2) h->peer_uni_mixed = true;
	else if (kind == 1) { if (h->peer_uni_w == SK_UNI_NONE) h->peer_uni_w = w; else if (h->peer_uni_w != w) h->peer_uni_mixed = true; }
	return KMR_OK;
}
int kmr_sk_exchange_adopt_dev(kmr_handle *h, const void *dev_data, const void *dev_meta, uint64_t n_chunks, uint64_t n_granules) {
	if (!h || (n_chunks && (!dev_data || !dev_meta))) return KMR_ERR_INVALID_ARG;
	int rc = sk_exchange_ready(h, "kmr_sk_exchange_adopt_dev"); if (rc) return rc;
	if (n_chunks == 0) return KMR_OK;
	hipSetDevice(h->device);
	rc = sk_ensure_state(h); if (rc) return rc;
	const int grid = (int)std::min<uint64_t>((n_chunks + SK_ADOPT_WAVES * SK_ADOPT_GROUP - 1) / (SK_ADOPT_WAVES * SK_ADOPT_GROUP), (uint64_t)num_cus(h) * 8);
	/* (a received chunk is appended as one piece: at worst every one of them opens a chunk of its own) */
	rc = pool_reserve(h, h->l1, n_chunks + n_granules / SK_CHUNK_G + (sk_list_count(h->sk_bits) / h->cfg.world_size) + (uint64_t)grid * SK_ADOPT_WAVES * 130 + 64, true); if (rc) return rc;
	/* per-chunk counts and their scan: a grow-only buffer of the handle (a job adopts once per piece and batch) */
	const size_t need = 8 * (n_chunks + 1) + 4 * (n_chunks + 1) + 256;
	if (h->adopt_cap < need) {
		if (h->adopt_buf) { hipStreamSynchronize(h->stream); hipFree(h->adopt_buf); h->adopt_buf = nullptr; h->adopt_cap = 0; }
		HIPCHK(h, dev_malloc((void **)&h->adopt_buf, need + need / 4)); h->adopt_cap = need + need / 4;
	}
	uint64_t *start = (uint64_t *)h->adopt_buf; uint32_t *cnt = (uint32_t *)(h->adopt_buf + 8 * (n_chunks + 1));
	hipLaunchKernelGGL(sk_meta_counts_kernel, dim3(grid_for(n_chunks)), dim3(256), 0, h->stream, (const uint2 *)dev_meta, n_chunks, cnt);
	rc = exclusive_scan(h, cnt, n_chunks, start);
	if (!rc && !h->d_uni && !h->peers_declare) {
		if (dev_malloc((void **)&h->d_uni, 8) != hipSuccess) rc = fail(h, KMR_ERR_OOM, "uniform-weight flags");
		else { const uint32_t init[2] = {SK_UNI_NONE, 0u}; if (hipMemcpyAsync(h->d_uni, init, 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = fail(h, KMR_ERR_HIP, "uniform-weight flags"); else hipStreamSynchronize(h->stream); }
	}
	/* (senders that declare their weights -- kmr_sk_exchange_peer_uniform, what both drivers do -- spare the owner this look at every
	 * received header: 3.5 ms for 4.2 GB at 8 ranks) */
	if (!rc && !h->peers_declare) hipLaunchKernelGGL(sk_uniform_check_kernel, dim3(grid_for(n_chunks)), dim3(256), 0, h->stream, (const uint4 *)dev_data, start, cnt, n_chunks, h->d_uni);
	if (!rc) {
		hipLaunchKernelGGL(sk_adopt_kernel, dim3(grid), dim3(SK_ADOPT_WAVES * 64), 0, h->stream, (const uint4 *)dev_data, (const uint2 *)dev_meta, start, n_chunks, sk_params(h), pool_view(h, h->l1));
		if (hipGetLastError() != hipSuccess) rc = fail(h, KMR_ERR_HIP, "sk_adopt_kernel launch");
	}
	hipStreamSynchronize(h->stream);
	return rc ? rc : sync_state(h);
}

/* ---- f3: KmerSpectrum::SizeTracker (src/KmerSpectrum.h:812-900) */
int kmr_size_tracker(kmr_handle *h, int force_last, uint64_t *elements, uint64_t capacity, uint64_t *n_elements) {
	if (!h || !n_elements) return KMR_ERR_INVALID_ARG;
	if (!h->cfg.size_tracker) return fail(h, KMR_ERR_STATE, "kmr_size_tracker: kmr_config.size_tracker was not set");
	if (!h->finalized) return fail(h, KMR_ERR_STATE, "kmr_size_tracker before kmr_finalize");
	const uint64_t n = h->trk_elems.size() / 4 + (force_last ? 1 : 0);
	*n_elements = n;
	if (!elements) return KMR_OK;
	if (capacity < n) return fail(h, KMR_ERR_CAPACITY, "kmr_size_tracker: element buffer too small");
	if (!h->trk_elems.empty()) memcpy(elements, h->trk_elems.data(), 8 * h->trk_elems.size());
	if (force_last) {      /* trackSpectrum(true), as the apps call it after the build (apps/FilterReads.cpp:141) */
		const uint64_t sub = h->cfg.kmer_subsample > 1 ? h->cfg.kmer_subsample : 1;
		uint64_t *e = elements + h->trk_elems.size();
		e[0] = h->stats.raw_kmers * sub; e[1] = h->stats.raw_good_kmers * sub; e[2] = h->stats.unique_kmers * sub; e[3] = h->stats.singleton_kmers * sub;
	}
	return KMR_OK;
}

#include "kmr_exchange_rccl.hpp"

int kmr_kernel_time(kmr_handle *h, int which, double *ms, uint64_t *launches) {
	if (!h || which < 0 || which >= KMR_TIME_GROUPS) return KMR_ERR_INVALID_ARG;
	hipSetDevice(h->device);
	int rc = sync_state(h); if (rc) return rc;
	if (ms) *ms = h->ms[which];
	if (launches) *launches = h->launches[which];
	return KMR_OK;
}
int kmr_kernel_time_reset(kmr_handle *h) {
	if (!h) return KMR_ERR_INVALID_ARG;
	int rc = sync_state(h); if (rc) return rc;
	for (int i = 0; i < KMR_TIME_GROUPS; i++) { h->ms[i] = 0; h->launches[i] = 0; }
	return KMR_OK;
}

}  // extern "C"
