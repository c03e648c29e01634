/* The owner exchange inside the library: RCCL called directly.  Included by kmr_api.hip inside its extern "C" block.
 *
 * Replaces DistributedKmerSpectrum::_buildKmerSpectrumMPI (src/DistributedFunctions.h:340-458) and the MPI_Alltoallv under it
 * (src/MPIBuffer.h:588-600) for a host that runs one process (or thread) per GPU of a node without MPI and without Python:
 *
 *   rank 0:      kmr_exchange_unique_id(id)            -> hand the 128 bytes to every rank (a file, a pipe, MPI_Bcast ...)
 *   every rank:  kmr_exchange_init(h, id)              collective: ncclCommInitRank(world_size, id, rank) on the handle's device
 *                (or kmr_exchange_init_transport(h, t): the host's own all-gather / all-to-all, e.g. MPI, instead of RCCL)
 *   per batch:   kmr_exchange_add_reads_dev(h, ...)    collective: extract -> counts -> all-to-all -> insert at the owner
 *   then:        kmr_finalize(h, ...) as on one GPU
 *
 * build_mode 3 moves whole chunks of super-k-mer lists (kmr_sk_exchange_*), the other modes 12-byte k-mer records
 * (kmr_extract_by_owner_dev / kmr_insert_records_dev).  librccl is dlopen'ed at kmr_exchange_unique_id / kmr_exchange_init: the
 * library has no link-time dependency on it, and a process that already holds one (PyTorch's) shares that copy.  All-to-alls are
 * grouped ncclSend / ncclRecv pairs, one message per peer and slice of <= 1 GiB; the local share never moves. */
/* (<dlfcn.h> and <rccl/rccl.h> -- for its types; nothing of it is linked -- are included at the top of kmr_api.hip) */

struct RcclApi {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static const char *rccl_load() {
	if (g_rccl.lib) return nullptr;
	void *lib = nullptr;
	for (const char *name : {"librccl.so", "librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD); if (lib) break; }      /* one already in the process */
	if (!lib) for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
	if (!lib) return "librccl.so not found (dlopen)";
#define RSYM(field, sym) g_rccl.field = (decltype(g_rccl.field))dlsym(lib, sym); if (!g_rccl.field) return "librccl.so lacks " sym;
	RSYM(GetUniqueId, "ncclGetUniqueId") RSYM(CommInitRank, "ncclCommInitRank") RSYM(CommDestroy, "ncclCommDestroy")
	RSYM(GroupStart, "ncclGroupStart") RSYM(GroupEnd, "ncclGroupEnd") RSYM(Send, "ncclSend") RSYM(Recv, "ncclRecv")
	RSYM(AllGather, "ncclAllGather") RSYM(GetErrorString, "ncclGetErrorString")
#undef RSYM
	g_rccl.lib = lib;
	return nullptr;
}
#define RCCLCHK(h, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail(h, KMR_ERR_HIP, std::string(#call ": ") + g_rccl.GetErrorString(r_)); } while (0)

static const uint64_t XC_MAX_MESSAGE = 1ull << 30;      /* bytes per ncclSend (tools/a2a_check.py: larger all-to-all payloads came back damaged on this stack) */

int kmr_exchange_unique_id(void *id) {
	if (!id) return KMR_ERR_INVALID_ARG;
	if (const char *e = rccl_load()) return fail(nullptr, KMR_ERR_UNSUPPORTED, e);
	static_assert(sizeof(ncclUniqueId) == KMR_EXCHANGE_ID_BYTES, "KMR_EXCHANGE_ID_BYTES");
	ncclUniqueId u;
	RCCLCHK(nullptr, g_rccl.GetUniqueId(&u));
	memcpy(id, &u, sizeof(u));
	return KMR_OK;
}
static int exchange_ready(kmr_handle *h) {      /* common tail of the two inits */
	if (h->cfg.world_size > SK_OWNER_MAX) return fail(h, KMR_ERR_UNSUPPORTED, "at most 64 ranks");
	/* build_mode 0: a job that exchanges through the library runs on the lists when the handle can (direction-counting values, a
	 * minimizer geometry for k) and nothing has been fed to it yet */
	if (h->auto_mode && !h->superkmer_mode && !h->ext && h->dPk && !h->reads && !h->inserted_records && !h->sk_state) h->superkmer_mode = true;
	if (h->superkmer_mode) return kmr_sk_exchange_begin(h);
	return KMR_OK;
}
static int rccl_allgather_u64(void *user, const uint64_t *mine, uint64_t n, uint64_t *all);
static int rccl_alltoallv_dev(void *user, const void *send, const uint64_t *send_off, const uint64_t *send_bytes, void *recv, const uint64_t *recv_off, const uint64_t *recv_bytes, void *stream);
int kmr_exchange_init(kmr_handle *h, const void *id) {
	if (!h || !id) return KMR_ERR_INVALID_ARG;
	if (h->xc_tr.allgather_u64) return fail(h, KMR_ERR_STATE, "kmr_exchange_init: the handle already has a transport");
	if (const char *e = rccl_load()) return fail(h, KMR_ERR_UNSUPPORTED, e);
	if (h->cfg.world_size > SK_OWNER_MAX) return fail(h, KMR_ERR_UNSUPPORTED, "at most 64 ranks");
	hipSetDevice(h->device);
	ncclUniqueId u; memcpy(&u, id, sizeof(u));
	ncclComm_t comm = nullptr;
	RCCLCHK(h, g_rccl.CommInitRank(&comm, (int)h->cfg.world_size, u, (int)h->cfg.rank));
	h->xc_comm = comm;
	HIPCHK(h, dev_malloc((void **)&h->xc_small, 8ull * (2 * SK_OWNER_MAX + 2) * (SK_OWNER_MAX + 1)));
	h->xc_tr.user = h; h->xc_tr.allgather_u64 = rccl_allgather_u64; h->xc_tr.alltoallv_dev = rccl_alltoallv_dev;
	return exchange_ready(h);
}
int kmr_exchange_init_transport(kmr_handle *h, const kmr_transport *t) {
	if (!h || !t || !t->allgather_u64 || !t->alltoallv_dev) return KMR_ERR_INVALID_ARG;
	if (h->xc_tr.allgather_u64) return fail(h, KMR_ERR_STATE, "kmr_exchange_init_transport: the handle already has a transport");
	h->xc_tr = *t;
	return exchange_ready(h);
}
static void exchange_free(kmr_handle *h) {      /* kmr_destroy */
	if (h->xc_comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)h->xc_comm);
	h->xc_comm = nullptr;
	for (void **p : {&h->xc_send, &h->xc_send2, &h->xc_recv, &h->xc_recv2}) { if (*p) hipFree(*p); *p = nullptr; }
	if (h->xc_small) hipFree(h->xc_small);
	if (h->xc_dcounts) hipFree(h->xc_dcounts);
	h->xc_small = nullptr; h->xc_dcounts = nullptr;
}
static int xc_reserve(kmr_handle *h, void **p, uint64_t &cap, uint64_t bytes) {
	if (bytes <= cap && *p) return 0;
	if (*p) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree(*p); *p = nullptr; cap = 0; }
	bytes = std::max<uint64_t>(bytes + bytes / 8, 4096);
	if (dev_malloc(p, bytes) != hipSuccess) { *p = nullptr; return fail(h, KMR_ERR_OOM, "exchange buffers"); }
	cap = bytes;
	return 0;
}
/* ---- the built-in transport: RCCL on the handle's stream */
static int rccl_allgather_u64(void *user, const uint64_t *mine, uint64_t n, uint64_t *all) {
	kmr_handle *h = (kmr_handle *)user;
	const uint32_t world = h->cfg.world_size;
	if (n > 2 * SK_OWNER_MAX + 2) return fail(h, KMR_ERR_INVALID_ARG, "allgather row too long");
	unsigned long long *d = h->xc_small;      /* [n] mine, then [world][n] */
	HIPCHK(h, hipMemcpyAsync(d, mine, 8 * n, hipMemcpyHostToDevice, h->stream));
	RCCLCHK(h, g_rccl.AllGather(d, d + n, n, ncclUint64, (ncclComm_t)h->xc_comm, h->stream));
	HIPCHK(h, hipMemcpyAsync(all, d + n, 8 * n * world, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	return 0;
}
static int rccl_alltoallv_dev(void *user, const void *send, const uint64_t *send_off, const uint64_t *send_bytes,
                              void *recv, const uint64_t *recv_off, const uint64_t *recv_bytes, void *stream) {
	kmr_handle *h = (kmr_handle *)user;
	const uint32_t world = h->cfg.world_size, rank = h->cfg.rank;
	RCCLCHK(h, g_rccl.GroupStart());
	for (uint32_t r = 0; r < world; r++) {
		if (r == rank) continue;
		if (send_bytes[r]) RCCLCHK(h, g_rccl.Send((const uint8_t *)send + send_off[r], send_bytes[r], ncclUint8, (int)r, (ncclComm_t)h->xc_comm, (hipStream_t)stream));
		if (recv_bytes[r]) RCCLCHK(h, g_rccl.Recv((uint8_t *)recv + recv_off[r], recv_bytes[r], ncclUint8, (int)r, (ncclComm_t)h->xc_comm, (hipStream_t)stream));
	}
	RCCLCHK(h, g_rccl.GroupEnd());
	return 0;
}

/* every rank's row of numbers -> all rows, on the host: all[r * row + j] */
static int xc_allgather_rows(kmr_handle *h, const std::vector<uint64_t> &mine, std::vector<uint64_t> &all) {
	const uint32_t world = h->cfg.world_size; const size_t row = mine.size();
	all.assign(row * world, 0);
	const int rc = h->xc_tr.allgather_u64(h->xc_tr.user, mine.data(), row, all.data());
	if (rc) return h->xc_tr.user == (void *)h ? rc : fail(h, rc, "transport: allgather_u64 failed");
	return 0;
}
/* all-to-all of byte segments: send[r] bytes from sbuf + soff[r] to rank r, recv[r] bytes from rank r to rbuf + roff[r], in `slices`
 * rounds so that no message exceeds XC_MAX_MESSAGE (the number of rounds comes from the gathered matrix: the same on every rank) */
static int xc_alltoallv(kmr_handle *h, const uint8_t *sbuf, const std::vector<uint64_t> &soff, const std::vector<uint64_t> &send,
                        uint8_t *rbuf, const std::vector<uint64_t> &roff, const std::vector<uint64_t> &recv, uint64_t slices, uint64_t unit) {
	const uint32_t world = h->cfg.world_size, rank = h->cfg.rank;
	auto part = [&](uint64_t total, uint64_t s, uint64_t &lo, uint64_t &n) {      /* slice s of `total` bytes, cut at multiples of `unit` */
		const uint64_t units = total / unit, per = (units + slices - 1) / slices;
		const uint64_t a = std::min(units, s * per), b = std::min(units, (s + 1) * per);
		lo = a * unit; n = (b - a) * unit;
	};
	std::vector<uint64_t> so(world), sn(world), ro(world), rn(world);
	for (uint64_t s = 0; s < slices; s++) {
		for (uint32_t r = 0; r < world; r++) {
			uint64_t lo, n;
			part(r == rank ? 0 : send[r], s, lo, n); so[r] = soff[r] + lo; sn[r] = n;
			part(r == rank ? 0 : recv[r], s, lo, n); ro[r] = roff[r] + lo; rn[r] = n;
		}
		const int rc = h->xc_tr.alltoallv_dev(h->xc_tr.user, sbuf, so.data(), sn.data(), rbuf, ro.data(), rn.data(), (void *)h->stream);
		if (rc) return h->xc_tr.user == (void *)h ? rc : fail(h, rc, "transport: alltoallv_dev failed");
	}
	return 0;
}

/* A collective step must not be left by one rank alone: what a rank does between two collectives (extraction, packing, growing its
 * buffers) can fail on that rank only, and the others would wait in the next collective for ever.  Every gathered row therefore
 * ends in a STATUS word -- the rank's error code so far -- and after each gather all ranks look at all status words: if any is
 * set, every rank leaves with an error before anything is sent (the failing rank with its own code and message, the others with
 * KMR_ERR_STATE naming the rank). */
static int xc_agree(kmr_handle *h, int local_rc, const std::vector<uint64_t> &all, size_t row) {
	const uint32_t world = h->cfg.world_size;
	for (uint32_t r = 0; r < world; r++) {
		const int code = (int)(int64_t)all[(size_t)r * row + row - 1];
		if (code == 0) continue;
		if (local_rc) return local_rc;      /* this rank's own error (its message is set) */
		return fail(h, KMR_ERR_STATE, "exchange: rank " + std::to_string(r) + " failed in this batch (code " + std::to_string(code) + "); no rank sent anything");
	}
	return local_rc;
}

int kmr_exchange_add_reads_dev(kmr_handle *h, const void *dev_bases, const void *dev_quals, const void *dev_offsets, uint64_t n_reads, uint64_t total_bases,
                               uint64_t first_global_read_idx, const void *dev_discarded) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (!h->xc_tr.allgather_u64) return fail(h, KMR_ERR_STATE, "kmr_exchange_add_reads_dev before kmr_exchange_init / kmr_exchange_init_transport");
	if (h->finalized) return fail(h, KMR_ERR_STATE, "kmr_exchange_add_reads_dev after kmr_finalize (kmr_reset first)");
	hipSetDevice(h->device);
	const uint32_t world = h->cfg.world_size, rank = h->cfg.rank;
	const size_t row = 2 * (size_t)world + 2;      /* numbers of the step, the sender's uniform-weight state (lists), then the status word */
	int rc = 0, lrc = 0;                            /* lrc: what went wrong on THIS rank since the last agreement */
	if (n_reads && (!dev_bases || !dev_offsets)) { lrc = fail(h, KMR_ERR_INVALID_ARG, "null device buffer"); n_reads = 0; total_bases = 0; }
	if (h->tune.exchange_fail_once) { h->tune.exchange_fail_once = false; lrc = fail(h, KMR_ERR_OOM, "injected failure (kmr_tune exchange_fail_once: the tests' way to fail one rank of a collective step)"); n_reads = 0; total_bases = 0; }
	std::vector<uint64_t> mine(row, 0), all;
	hipEvent_t ea = nullptr, eb = nullptr;
	auto status = [&](int code) { mine[row - 1] = (uint64_t)(int64_t)code; };
	if (h->superkmer_mode) {
		/* global ordinals: this rank's batch begins behind the batches of the lower ranks, and behind everything the job was fed before */
		mine[0] = total_bases;
		rc = xc_allgather_rows(h, mine, all); if (rc) return rc;      /* (a rank that failed already still takes part: it reports at the next gather) */
		uint64_t before = 0, job = 0;
		for (uint32_t r = 0; r < world; r++) { if (r < rank) before += all[(size_t)r * row]; job += all[(size_t)r * row]; }
		std::vector<uint64_t> chunks(world, 0), granules(world, 0), goff(world, 0), coff(world, 0);
		uint64_t ag = 0, ac = 0;
		if (!lrc) lrc = kmr_set_stream_origin(h, h->xc_job_bases + before);
		h->xc_job_bases += job;
		if (!lrc && n_reads) lrc = kmr_add_reads_dev(h, dev_bases, dev_quals, dev_offsets, n_reads, total_bases, first_global_read_idx, dev_discarded);
		if (!lrc) lrc = kmr_sk_exchange_counts(h, chunks.data(), granules.data());
		if (!lrc) {
			for (uint32_t r = 0; r < world; r++) {
				if (r == rank) chunks[r] = granules[r] = 0;
				/* sk_pack_kernel books chunks << 40 | granules in one word per owner */
				if (chunks[r] >= (1ull << 24) || granules[r] >= (1ull << 40)) lrc = fail(h, KMR_ERR_CAPACITY, "exchange: more than 2^24 chunks for one owner in one batch; feed smaller batches");
				goff[r] = ag; coff[r] = ac; ag += granules[r]; ac += chunks[r];
			}
		}
		if (!lrc) lrc = xc_reserve(h, &h->xc_send, h->xc_send_cap, 16 * std::max<uint64_t>(ag, 1));
		if (!lrc) lrc = xc_reserve(h, &h->xc_send2, h->xc_send2_cap, 8 * std::max<uint64_t>(ac, 1));
		if (!lrc) lrc = kmr_sk_exchange_pack_dev(h, h->xc_send, h->xc_send2, goff.data(), coff.data());
		std::fill(mine.begin(), mine.end(), 0);
		if (!lrc) for (uint32_t r = 0; r < world; r++) { mine[r] = chunks[r]; mine[world + r] = granules[r]; }
		if (!lrc) { uint64_t ust = 0; kmr_sk_exchange_uniform(h, &ust); mine[2 * world] = ust; }
		status(lrc);
		rc = xc_allgather_rows(h, mine, all); if (rc) return rc;
		rc = xc_agree(h, lrc, all, row); if (rc) return rc;
		for (uint32_t r = 0; r < world; r++) if (r != rank && all[(size_t)r * row + rank]) { rc = kmr_sk_exchange_peer_uniform(h, all[(size_t)r * row + 2 * world]); if (rc) return rc; }
		std::vector<uint64_t> rc_c(world, 0), rc_g(world, 0), rgo(world, 0), rco(world, 0), sgb(world), scb(world), sgo(world), sco(world);
		uint64_t rg = 0, rcn = 0, biggest = 0;
		for (uint32_t r = 0; r < world; r++) {
			rc_c[r] = 8 * all[(size_t)r * row + rank]; rc_g[r] = 16 * all[(size_t)r * row + world + rank];
			rco[r] = rcn; rgo[r] = rg; rcn += rc_c[r]; rg += rc_g[r];
			scb[r] = 8 * chunks[r]; sgb[r] = 16 * granules[r]; sco[r] = 8 * coff[r]; sgo[r] = 16 * goff[r];
			for (uint32_t q = 0; q < world; q++) biggest = std::max<uint64_t>(biggest, 16 * all[(size_t)r * row + world + q]);
		}
		const uint64_t slices = std::max<uint64_t>(1, (biggest + XC_MAX_MESSAGE - 1) / XC_MAX_MESSAGE);
		/* the receive buffers are the last thing that can fail on one rank alone: one more (status-only) agreement before anything moves */
		lrc = xc_reserve(h, &h->xc_recv, h->xc_recv_cap, std::max<uint64_t>(rg, 16));
		if (!lrc) lrc = xc_reserve(h, &h->xc_recv2, h->xc_recv2_cap, std::max<uint64_t>(rcn, 8));
		std::fill(mine.begin(), mine.end(), 0); status(lrc);
		rc = xc_allgather_rows(h, mine, all); if (rc) return rc;
		rc = xc_agree(h, lrc, all, row); if (rc) return rc;
		time_begin(h, KMR_TIME_EXCHANGE, &ea, &eb);
		rc = xc_alltoallv(h, (const uint8_t *)h->xc_send2, sco, scb, (uint8_t *)h->xc_recv2, rco, rc_c, 1, 8);
		if (!rc) rc = xc_alltoallv(h, (const uint8_t *)h->xc_send, sgo, sgb, (uint8_t *)h->xc_recv, rgo, rc_g, slices, 16);
		time_end(h, KMR_TIME_EXCHANGE, ea, eb);
		if (rc) return rc;
		HIPCHK(h, hipStreamSynchronize(h->stream));
		h->xc_bytes_to_peers += 16 * ag + 8 * ac;
		if (rcn) { rc = kmr_sk_exchange_adopt_dev(h, h->xc_recv, h->xc_recv2, rcn / 8, rg / 16); if (rc) return rc; }
		return KMR_OK;
	}
	/* k-mer records: every owner's segment of this batch, the counts, the records, the insert */
	const uint64_t rb = KMR_RECORD_BYTES(h->k, h->cfg.value_kind);
	const uint64_t upper = total_bases + 64;
	uint64_t segcap = std::min<uint64_t>(upper, upper / world + upper / (4 * world) + 4096);
	const uint64_t sb = h->stream_base, rd = h->reads;
	unsigned long long bad0 = 0; hipMemcpy(&bad0, &h->dstats->sender_bad, 8, hipMemcpyDeviceToHost);      /* a repeated attempt must not count the dropped k-mers twice */
	std::vector<uint64_t> counts(world, 0);
	if (!lrc && !h->xc_dcounts && dev_malloc((void **)&h->xc_dcounts, 8 * SK_OWNER_MAX) != hipSuccess) { h->xc_dcounts = nullptr; lrc = fail(h, KMR_ERR_OOM, "exchange counters"); }
	unsigned long long *dcounts = h->xc_dcounts;
	while (!lrc) {      /* a skewed batch (one owner takes more than its share) is extracted again into larger segments */
		lrc = xc_reserve(h, &h->xc_send, h->xc_send_cap, (uint64_t)world * segcap * rb); if (lrc) break;
		h->stream_base = sb; h->reads = rd;
		if (n_reads) lrc = kmr_extract_by_owner_dev(h, dev_bases, dev_quals, dev_offsets, n_reads, total_bases, first_global_read_idx, dev_discarded, h->xc_send, segcap, dcounts);
		else if (hipMemsetAsync(dcounts, 0, 8 * world, h->stream) != hipSuccess) lrc = fail(h, KMR_ERR_HIP, "hipMemsetAsync(exchange counters)");
		if (!lrc) lrc = sync_state(h);
		if (lrc == KMR_ERR_CAPACITY && segcap < upper) {
			uint32_t e = 0; hipMemcpy(&e, h->derr, 4, hipMemcpyDeviceToHost); e &= ~(uint32_t)ERR_SEGMENT_OVERFLOW; hipMemcpy(h->derr, &e, 4, hipMemcpyHostToDevice); hipMemcpy(&h->dstats->sender_bad, &bad0, 8, hipMemcpyHostToDevice);
			segcap = std::min<uint64_t>(upper, segcap * 2);
			lrc = 0;
			continue;
		}
		break;
	}
	if (!lrc && hipMemcpy(counts.data(), dcounts, 8 * world, hipMemcpyDeviceToHost) != hipSuccess) lrc = fail(h, KMR_ERR_HIP, "hipMemcpy(exchange counters)");
	if (!lrc) for (uint32_t r = 0; r < world; r++) mine[r] = counts[r];
	status(lrc);
	rc = xc_allgather_rows(h, mine, all); if (rc) return rc;
	rc = xc_agree(h, lrc, all, row); if (rc) return rc;
	std::vector<uint64_t> sbytes(world), soff(world), rbytes(world, 0), roff(world, 0);
	uint64_t got = 0, biggest = 0, sent = 0;
	for (uint32_t r = 0; r < world; r++) {
		sbytes[r] = r == rank ? 0 : counts[r] * rb; soff[r] = (uint64_t)r * segcap * rb; sent += sbytes[r];
		rbytes[r] = r == rank ? 0 : all[(size_t)r * row + rank] * rb; roff[r] = got; got += rbytes[r];
		for (uint32_t q = 0; q < world; q++) if (q != r) biggest = std::max<uint64_t>(biggest, all[(size_t)r * row + q] * rb);
	}
	const uint64_t slices = std::max<uint64_t>(1, (biggest + XC_MAX_MESSAGE - 1) / XC_MAX_MESSAGE);
	lrc = xc_reserve(h, &h->xc_recv, h->xc_recv_cap, std::max<uint64_t>(got, 16));
	std::fill(mine.begin(), mine.end(), 0); status(lrc);
	rc = xc_allgather_rows(h, mine, all); if (rc) return rc;
	rc = xc_agree(h, lrc, all, row); if (rc) return rc;
	time_begin(h, KMR_TIME_EXCHANGE, &ea, &eb);
	rc = xc_alltoallv(h, (const uint8_t *)h->xc_send, soff, sbytes, (uint8_t *)h->xc_recv, roff, rbytes, slices, rb);
	time_end(h, KMR_TIME_EXCHANGE, ea, eb);
	if (rc) return rc;
	HIPCHK(h, hipStreamSynchronize(h->stream));
	h->xc_bytes_to_peers += sent;
	if (counts[rank]) { rc = kmr_insert_records_dev(h, (const uint8_t *)h->xc_send + (uint64_t)rank * segcap * rb, counts[rank]); if (rc) return rc; }
	if (got) { rc = kmr_insert_records_dev(h, h->xc_recv, got / rb); if (rc) return rc; }
	return sync_state(h);
}
int kmr_copy_to_host(kmr_handle *h, void *host_dst, const void *dev_src, uint64_t bytes) {
	if (!h || (bytes && (!host_dst || !dev_src))) return KMR_ERR_INVALID_ARG;
	hipSetDevice(h->device);
	HIPCHK(h, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	return KMR_OK;
}
int kmr_copy_to_device(kmr_handle *h, void *dev_dst, const void *host_src, uint64_t bytes) {
	if (!h || (bytes && (!dev_dst || !host_src))) return KMR_ERR_INVALID_ARG;
	hipSetDevice(h->device);
	HIPCHK(h, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	return KMR_OK;
}
int kmr_exchange_add_read_batch(kmr_handle *h, const kmr_reads *r, uint64_t first_global_read_idx) {
	if (!h) return KMR_ERR_INVALID_ARG;
	if (r && r->device != h->device) return fail(h, KMR_ERR_INVALID_ARG, "read batch lives on another device");
	int rc = r ? kmr_exchange_add_reads_dev(h, r->bases, r->quals, r->offsets, r->n, r->total, first_global_read_idx, nullptr)
	           : kmr_exchange_add_reads_dev(h, nullptr, nullptr, nullptr, 0, 0, first_global_read_idx, nullptr);
	if (!rc) rc = kmr_sync(h);
	return rc;
}
int kmr_exchange_stats(kmr_handle *h, uint64_t *bytes_to_peers, double *alltoall_ms) {
	if (!h) return KMR_ERR_INVALID_ARG;
	hipSetDevice(h->device);
	int rc = sync_state(h); if (rc) return rc;
	if (bytes_to_peers) *bytes_to_peers = h->xc_bytes_to_peers;
	if (alltoall_ms) *alltoall_ms = h->ms[KMR_TIME_EXCHANGE];
	return KMR_OK;
}
