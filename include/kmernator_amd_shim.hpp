/*
 * kmernator_amd_shim.hpp -- the reference-side binding a Kmernator maintainer adds.
 *
 * Include AFTER Kmernator's own "KmerSpectrum.h" (and, for the -P tools, "DistributedFunctions.h"): it needs ReadSet, Read,
 * KmerSizer, the map types and the option singletons.  Written in the reference's dialect (C++03, no Boost needed by
 * the shim itself).  The reference's headers cannot be compiled in this repository (Boost 1.53 / sparsehash are absent),
 * so tests/test_shim_compile.py compiles this header against tests/cpp/mock_kmernator.h -- a mock that DECLARES the handful
 * of reference names used here with the signatures they have in the reference (each cites its file:line).  That test catches
 * syntax, access and ownership mistakes; it pins no behaviour of the reference.
 *
 * The whole patch, against the reference as it is:
 *
 *   src/KmerSpectrum.h:404     -private:                       (rawKmers ... subtracted, :405-409)
 *                              +protected:                      so that a derived spectrum can set the counters it computed
 *   apps/FilterReads.h:246     (after `typedef KmerSpectrum<DataType, DataType, SingletonDataType> KS;`)
 *                              +#include "kmernator_amd_shim.hpp"
 *                              +typedef GpuKmerSpectrum<KS> GKS;
 *   apps/FilterReads.cpp:126   -KS spectrum(0);                 +GKS spectrum(0);            (no device handle yet: it is made by the first build)
 *   apps/FilterReads.cpp:136   -spectrum = KS(rawKmers);        +spectrum = GKS(rawKmers);   (value semantics: the handle is shared and counted)
 *   apps/FilterReads.cpp:139   unchanged: buildKmerSpectrumInParts(reads, parts, ...) calls the virtual buildKmerSpectrum(store, isSolid)
 *                              (src/KmerSpectrum.h:1822), which GpuKmerSpectrum overrides; with --build-partitions > 1 the
 *                              reference's own loop (:1831-1902) runs and calls the 4-argument non-virtual overload, i.e. the CPU path.
 *   apps/MeraculousCounter.cpp:126  KS -> GpuDistributedKmerSpectrum<KS> with KMR_VALUE_EXT (maps are ExtensionTrackingData, src/Meraculous.h:79-80)
 *   apps/FilterReads-P.cpp:110      KS spectrum(world, rawKmers) -> GpuDistributedKmerSpectrum<KS> spectrum(world, rawKmers)
 *   apps/CMakeLists.txt        target_link_libraries(<tool> kmernator_amd), -I<this repo>/include, -DKMERNATOR_AMD_SHIM_MPI for the -P tools
 *
 * buildKmerSpectrum() hands the ReadSet over as it keeps its reads (2-bit packed bases + markups, kmr_add_reads_twobit), runs the build on the MI355X and fills weak
 * (and singleton) -- the reference's OWN map objects -- from the images the library writes in the store() layout
 * (src/Kmer.h:3143-3159) through the copying constructor KmerMapByKmerArrayPair(const void *) (src/Kmer.h:3124-3135), exactly
 * as KmerSpectrum::restoreMmap does (src/KmerSpectrum.h:489-518).  Everything downstream (optimize, trackSpectrum, histogram,
 * ReadSelector, storeMmap) then runs unchanged on the reference types.
 *
 * One process drives ONE device (kmr_config.device).  The reference's unit of distribution is the MPI rank
 * (apps/FilterReads-P.cpp:263-325), so "N GPUs" is N ranks with one handle each -- GpuDistributedKmerSpectrum below, or
 * kmr_exchange_* of the C-ABI over RCCL for hosts without MPI; that is why kmr_config has no num_devices.
 */
#ifndef KMERNATOR_AMD_SHIM_HPP_
#define KMERNATOR_AMD_SHIM_HPP_

#include <stdexcept>
#include <string>
#include <sstream>
#include <vector>

#include "kmernator_amd.h"

/* counted owner of a kmr_handle: the reference copies and assigns spectra by value (apps/FilterReads.cpp:126,136) */
class KmrSharedHandle {
public:
	KmrSharedHandle() : _p(NULL) {}
	KmrSharedHandle(const KmrSharedHandle &o) : _p(o._p) { if (_p) _p->refs++; }
	KmrSharedHandle &operator=(const KmrSharedHandle &o) {
		if (o._p) o._p->refs++;
		release();
		_p = o._p;
		return *this;
	}
	~KmrSharedHandle() { release(); }
	kmr_handle *get() const { return _p ? _p->h : NULL; }
	void adopt(kmr_handle *h) { release(); _p = new Box(); _p->h = h; _p->refs = 1; }
private:
	struct Box { kmr_handle *h; long refs; };
	void release() { if (_p && --_p->refs == 0) { kmr_destroy(_p->h); delete _p; } _p = NULL; }
	Box *_p;
};

template <typename KS>
class GpuKmerSpectrum : public KS {
public:
	typedef typename KS::WeakMapType WeakMapType;
	typedef typename KS::SingletonMapType SingletonMapType;

	/* KmerSpectrum(estimatedRawKmers, separateSingletons), src/KmerSpectrum.h:414-421.  No device work happens here: the
	 * reference constructs `KS spectrum(0)` first and assigns the real one later */
	GpuKmerSpectrum(unsigned long estimatedRawKmers = 0, bool separateSingletons = true, int valueKind = KMR_VALUE_COUNT_DIR, int device = -1)
	    : KS(estimatedRawKmers, separateSingletons), _estimatedRawKmers(estimatedRawKmers), _separateSingletons(separateSingletons),
	      _valueKind(valueKind), _device(device), _rank(0), _worldSize(1), _sizeTracking(false), _wantSizeHistory(false) {}
	virtual ~GpuKmerSpectrum() {}
	/* --size-history-file (apps/FilterReads.cpp:141-147): call before the first build of this object.  Off by default: with it on the
	 * count pass keeps two first sightings per key and cannot cut hot minimizer lists (homopolymers, satellites) into pieces */
	void setSizeTracking(bool on) { _wantSizeHistory = on; }
	/* copy construction and assignment: KS copies its maps (src/KmerSpectrum.h:423-440), the device handle is shared */

	/* replaces KmerSpectrum::buildKmerSpectrum(const ReadSet &[, bool]) (src/KmerSpectrum.h:2081-2086) */
	virtual void buildKmerSpectrum(const ReadSet &store) { buildKmerSpectrum(store, false); }
	virtual void buildKmerSpectrum(const ReadSet &store, bool isSolid) {
		if (isSolid) { KS::buildKmerSpectrum(store, isSolid); return; }       /* solid map: not on this path */
		/* the reads go over as the ReadSet keeps them -- 2-bit packed bases and markups (src/Sequence.h:166-171), no string per read -- and
		 * are unpacked on the device (kmr_add_reads_twobit); qualities as characters, or Read::REF_QUAL for all when no read has any */
		PackedReads pr;
		flattenTwoBit(store, pr, 0, store.getSize());
		ensureHandle();
		check(kmr_reset(_handle.get()), "kmr_reset");
		check(kmr_add_reads_twobit(_handle.get(), pr.twobit.empty() ? (const uint8_t *)"" : &pr.twobit[0], &pr.twobitOffsets[0], &pr.offsets[0],
		                           pr.markupPos.empty() ? NULL : &pr.markupOffsets[0], pr.markupPos.empty() ? NULL : &pr.markupPos[0], pr.markupPos.empty() ? NULL : &pr.markupChar[0],
		                           pr.anyQuals ? pr.quals.data() : NULL, 0, store.getSize(), 0, &pr.discarded[0]), "kmr_add_reads_twobit");
		pull(KmerSpectrumOptions::getOptions().getMinDepth());
	}

	/* per-k-mer lookups stay on the restored reference maps; batch form for ReadSelector-style consumers */
	void getCounts(const std::vector<uint8_t> &packedCanonicalKmers, std::vector<uint32_t> &counts) {
		const size_t kb = KmerSizer::getByteSize();
		counts.resize(packedCanonicalKmers.size() / kb);
		if (counts.empty()) return;
		requireHandle();
		check(kmr_lookup(_handle.get(), &packedCanonicalKmers[0], counts.size(), &counts[0]), "kmr_lookup");
	}

	/* The artifact filter on the device, in place of FilterKnownOddities::applyFilter(reads) at apps/FilterReads.cpp:110-114:
	 *     spectrum.applyArtifactFilter(reads, FilterKnownOddities::getArtifactFasta() [+ repeat / PhiX tables], cfg);
	 * The screen runs on the GPU; trims, discards and remnant reads are applied to the reference's ReadSet exactly as
	 * Recorder::recordTrim / recordDiscard (src/FilterKnownOddities.h:310-334) and applyFilterToRead :519-528 do. */
	unsigned long applyArtifactFilter(ReadSet &reads, const std::string &artifactFasta, const kmr_artifact_config &cfg) {
		FlatReads fr;
		flatten(reads, fr, false);
		std::vector<int64_t> mate;
		if (reads.hasPairs()) {
			mate.assign(reads.getSize(), -1);
			for (ReadSet::ReadSetSizeType p = 0; p < reads.getPairSize(); p++) {
				const ReadSet::Pair &pair = reads.getPair(p);
				if (reads.isValidRead(pair.read1) && reads.isValidRead(pair.read2)) { mate[pair.read1] = pair.read2; mate[pair.read2] = pair.read1; }
			}
		}
		ensureHandle();
		kmr_handle *h = _handle.get();
		kmr_reads *batch = NULL; kmr_artifact_filter *filter = NULL;
		check(kmr_reads_from_host(h, fr.bases.data(), fr.quals.data(), &fr.offsets[0], reads.getSize(), &batch), "kmr_reads_from_host");
		int rc = kmr_artifact_filter_create(h, &cfg, artifactFasta.data(), artifactFasta.size(), &filter);
		const size_t n = reads.getSize();
		std::vector<uint32_t> value(n + 1), minPass(n + 1), maxPass(n + 1), remOff(n + 1), remLen(n + 1); std::vector<uint8_t> action(n + 1);
		if (rc == KMR_OK) rc = kmr_artifact_filter_apply(h, filter, batch, mate.empty() ? NULL : &mate[0], &value[0], &minPass[0], &maxPass[0],
		                                                 &action[0], &remOff[0], &remLen[0], NULL);
		kmr_artifact_filter_free(filter); kmr_reads_free(batch);
		check(rc, "kmr_artifact_filter");
		unsigned long affected = 0;
		ReadSet remnants;
		for (size_t i = 0; i < n; i++) {
			Read &read = reads.getRead(i);
			if (remLen[i]) remnants.append(read.getTrimRead(remOff[i], remLen[i], trimLabel(remOff[i], remLen[i]), "-qtrim"));
			if (action[i] == 1) { read = read.getTrimRead(minPass[i], maxPass[i] - minPass[i], trimLabel(minPass[i], maxPass[i] - minPass[i])); affected++; }
			else if (action[i] == 2) read.discard();
		}
		if (remnants.getSize() > 0) reads.append(remnants);
		return affected;
	}

	kmr_handle *handle() { return _handle.get(); }

protected:
	/* for spectra whose constructor takes the communicator first (DistributedKmerSpectrum, src/DistributedFunctions.h:124);
	 * the tag keeps this template from ever being chosen for a copy */
	struct WorldTag {};
	template <typename World>
	GpuKmerSpectrum(WorldTag, World &world, unsigned long estimatedRawKmers, bool separateSingletons, int valueKind, int device)
	    : KS(world, estimatedRawKmers, separateSingletons), _estimatedRawKmers(estimatedRawKmers), _separateSingletons(separateSingletons),
	      _valueKind(valueKind), _device(device), _rank(0), _worldSize(1), _sizeTracking(false), _wantSizeHistory(false) {}

	struct FlatReads {
		std::string bases, quals;
		std::vector<uint64_t> offsets;
		std::vector<uint8_t> discarded;
		bool anyQuals;
		FlatReads() : offsets(1, 0), anyQuals(false) {}
	};
	/* the arrays kmr_add_reads takes: sequence characters with markups applied (N / X), quality characters scaled to
	 * Read::FASTQ_START_CHAR (REF_QUAL strings for reads without quals), byte offsets per read */
	static void flatten(const ReadSet &store, FlatReads &fr, bool skipDiscarded) { flatten(store, fr, skipDiscarded, 0, store.getSize()); }
	static void flatten(const ReadSet &store, FlatReads &fr, bool skipDiscarded, ReadSet::ReadSetSizeType lo, ReadSet::ReadSetSizeType hi) {
		for (ReadSet::ReadSetSizeType i = lo; i < hi; i++) {
			const Read &read = store.getRead(i);
			const bool dis = read.isDiscarded();
			if (skipDiscarded) fr.discarded.push_back(dis ? 1 : 0);
			if (!(skipDiscarded && dis)) { fr.bases += read.getFasta(); fr.quals += read.getQuals(); fr.anyQuals = true; }
			fr.offsets.push_back(fr.bases.size());
		}
	}
	struct PackedReads {
		std::vector<uint8_t> twobit, discarded; std::string quals; std::vector<char> markupChar; std::vector<uint32_t> markupPos;
		std::vector<uint64_t> twobitOffsets, offsets, markupOffsets;
		bool anyQuals;
		PackedReads() : twobitOffsets(1, 0), offsets(1, 0), markupOffsets(1, 0), anyQuals(false) {}
	};
	/* the arrays kmr_add_reads_twobit takes, straight from the reads' own storage: Sequence::getTwoBitSequence / getMarkups
	 * (src/Sequence.h:280-289); a discarded read is handed over empty */
	static void flattenTwoBit(const ReadSet &store, PackedReads &pr, ReadSet::ReadSetSizeType lo, ReadSet::ReadSetSizeType hi) {
		bool someQuals = false, someWithout = false;
		for (ReadSet::ReadSetSizeType i = lo; i < hi; i++) { const Read &read = store.getRead(i); if (read.isDiscarded()) continue; if (read.hasQuals()) someQuals = true; else someWithout = true; }
		pr.anyQuals = someQuals;      /* (a mix: the reads without get their REF_QUAL string, as getQuals() gives it) */
		(void)someWithout;
		for (ReadSet::ReadSetSizeType i = lo; i < hi; i++) {
			const Read &read = store.getRead(i);
			const bool dis = read.isDiscarded();
			pr.discarded.push_back(dis ? 1 : 0);
			if (!dis) {
				const uint32_t nb = read.getTwoBitEncodingSequenceLength();
				const uint8_t *tb = (const uint8_t *)read.getTwoBitSequence();
				pr.twobit.insert(pr.twobit.end(), tb, tb + nb);
				const Read::BaseLocationVectorType mk = read.getMarkups();
				for (size_t m = 0; m < mk.size(); m++) { pr.markupChar.push_back(mk[m].first); pr.markupPos.push_back((uint32_t)mk[m].second); }
				if (pr.anyQuals) { const std::string q = read.getQuals(); pr.quals += q.size() == read.getLength() ? q : std::string(read.getLength(), (char)127); }
			}
			pr.twobitOffsets.push_back(pr.twobit.size());
			pr.offsets.push_back(pr.offsets.back() + (dis ? 0 : read.getLength()));
			pr.markupOffsets.push_back(pr.markupPos.size());
		}
	}
	static std::string trimLabel(uint32_t off, uint32_t len) { std::ostringstream ss; ss << "AFTrim:" << off << "+" << len; return ss.str(); }

	kmr_config makeConfig() {
		kmr_config c;
		kmr_config_init(&c);
		c.k = KmerSizer::getSequenceLength();                                   /* src/Kmer.h:115 */
		c.num_buckets_weak = this->weak.getNumBuckets();                        /* the KS ctor already sized them */
		c.num_buckets_singleton = this->singleton.getNumBuckets();
		c.estimated_raw_kmers = _estimatedRawKmers;
		c.value_kind = (uint32_t)_valueKind;
		c.min_weight = TrackingData::getMinimumWeight();                        /* src/KmerTrackingData.h:377 */
		c.min_quality_score = GeneralOptions::getOptions().getMinQuality();     /* src/Options.h:327 */
		c.fastq_start_char = Read::FASTQ_START_CHAR;                            /* reads are already rescaled to it */
		c.ext_min_quality = ExtensionTracking::getMinQuality();                 /* src/KmerTrackingData.h:157-163 */
		c.separate_singletons = _separateSingletons ? 1 : 0;
		c.kmer_subsample = (uint32_t)KS::getKmerSubsample();                    /* src/KmerSpectrum.h:461 */
		c.device = _device;
		c.rank = (uint32_t)_rank; c.world_size = (uint32_t)_worldSize;
		/* the size history (--size-history-file, apps/FilterReads.cpp:141-147) is kept where the library keeps it: kmr_size_tracker
		 * -- only when the application asks for the history (setSizeTracking): with it on the count pass keeps two first sightings per
		 * key and cannot cut hot minimizer lists into pieces */
		c.size_tracker = (_wantSizeHistory && _valueKind == KMR_VALUE_COUNT_DIR && _worldSize == 1 && c.k >= 16) ? 1 : 0;
		_sizeTracking = c.size_tracker != 0;
		return c;
	}
	/* the device handle is made by the first build (or by whoever asks for it first) and shared by copies of this object */
	void ensureHandle() {
		if (_handle.get()) return;
		kmr_config c = makeConfig();
		kmr_handle *h = NULL;
		const int rc = kmr_create(&c, &h);
		if (rc != KMR_OK) throw std::runtime_error(std::string("kmr_create: ") + kmr_last_error(NULL));
		_handle.adopt(h);
	}
	void requireHandle() { if (!_handle.get()) throw std::runtime_error("GpuKmerSpectrum: no spectrum has been built on the device yet"); }

	/* purgeMinDepth + materialise.  The maps copy out of the images (KmerMapByKmerArrayPair(const void *), src/Kmer.h:3124),
	 * so the image buffers die here. */
	void pull(unsigned int minDepth) {
		kmr_handle *h = _handle.get();
		check(kmr_finalize(h, minDepth), "kmr_finalize");
		std::vector<char> image;
		uint64_t n = 0;
		check(kmr_image_size(h, KMR_MAP_WEAK, &n), "kmr_image_size");
		image.resize(n);
		check(kmr_write_image(h, KMR_MAP_WEAK, &image[0], n), "kmr_write_image");
		{ WeakMapType tmp(&image[0]); this->weak.swap(tmp); }
		if (minDepth <= 1 && this->hasSingletons) {
			check(kmr_image_size(h, KMR_MAP_SINGLETON, &n), "kmr_image_size");
			image.resize(n);
			check(kmr_write_image(h, KMR_MAP_SINGLETON, &image[0], n), "kmr_write_image");
			SingletonMapType tmp(&image[0]); this->singleton.swap(tmp);
		} else {
			this->singleton.clear(false);         /* purgeMinDepth, src/KmerSpectrum.h:1805-1815 */
			this->hasSingletons = false;
		}
		kmr_stats st;
		check(kmr_get_stats(h, &st), "kmr_get_stats");
		/* private in the reference (src/KmerSpectrum.h:404-409): the patch at the top of this file makes them protected */
		this->rawKmers = (long)st.raw_kmers; this->rawGoodKmers = (long)st.raw_good_kmers;
		this->uniqueKmers = (long)st.unique_kmers; this->singletonKmers = (long)st.singleton_kmers;
		if (_sizeTracking) {      /* KmerSpectrum::sizeTracker as the build would have left it (the app's trackSpectrum(true) adds the last element) */
			uint64_t ne = 0;
			check(kmr_size_tracker(h, 0, NULL, 0, &ne), "kmr_size_tracker");
			std::vector<uint64_t> el(4 * ne + 4);
			if (ne) check(kmr_size_tracker(h, 0, &el[0], ne, &ne), "kmr_size_tracker");
			typename KS::SizeTracker tracker;
			tracker.elements.clear();
			for (uint64_t i = 0; i < ne; i++)
				tracker.elements.push_back(typename KS::SizeTracker::SizeTrackerElement((long)el[4 * i], (long)el[4 * i + 1], (long)el[4 * i + 2], (long)el[4 * i + 3]));
			this->setSizeTracker(tracker);
		}
	}

	void check(int rc, const char *what) {
		if (rc != KMR_OK) throw std::runtime_error(std::string(what) + ": " + kmr_last_error(_handle.get()));
	}

	KmrSharedHandle _handle;
	unsigned long _estimatedRawKmers;
	bool _separateSingletons;
	int _valueKind, _device, _rank, _worldSize;
	bool _sizeTracking, _wantSizeHistory;
};

#ifdef KMERNATOR_AMD_SHIM_MPI
#include <mpi.h>
/*
 * The -P tools: DKS = DistributedKmerSpectrum<...> or MeraculousDistributedKmerSpectrum (src/DistributedFunctions.h:100-131,
 * src/Meraculous.h:82-105), one MPI rank per GPU.  buildKmerSpectrum(store) replaces _buildKmerSpectrumMPI
 * (src/DistributedFunctions.h:340-458) with the library's own owner exchange (kmr_exchange_*): the rank's ReadSet goes to the
 * device in bounded batches, every batch is one collective step -- extract, all-to-all, insert at the owner -- and the steps of
 * all ranks are counted out beforehand (MPI_Allreduce MAX) so that every rank makes the same calls.  Two transports:
 *   TRANSPORT_RCCL (default)  the exchange runs over RCCL / xGMI between the GPUs: rank 0 makes the id (kmr_exchange_unique_id),
 *                             MPI_Bcast carries it, kmr_exchange_init joins the communicator.  Nothing but the id and the
 *                             batch count ever goes through MPI -- what replaces MPI_Alltoallv of src/MPIBuffer.h:588-600.
 *   TRANSPORT_MPI             the same driver over MPI (kmr_exchange_init_transport): MPI_Allgather for the counts; the segments
 *                             are staged through host memory and travel as point-to-point messages whose sizes and offsets are
 *                             64-bit (the library never asks for more than 1 GiB per message, and no `int` displacement of an
 *                             MPI_Alltoallv is involved).  For hosts without RCCL between their GPUs.
 * The owner of a k-mer is the build's own (getDistributedThreadId for k-mer records, the minimizer list for the default build):
 * the union of the ranks' maps is the serial spectrum either way.
 */
template <typename DKS>
class GpuDistributedKmerSpectrum : public GpuKmerSpectrum<DKS> {
public:
	typedef GpuKmerSpectrum<DKS> Base;
	enum Transport { TRANSPORT_RCCL = 0, TRANSPORT_MPI = 1 };
	/* DistributedKmerSpectrum(mpi::communicator &, estimatedRawKmers, separateSingletons), src/DistributedFunctions.h:124-131
	 * (mpi = boost::mpi there; the communicator converts to MPI_Comm) */
	GpuDistributedKmerSpectrum(mpi::communicator &world, unsigned long estimatedRawKmers = 0, bool separateSingletons = true, int valueKind = KMR_VALUE_COUNT_DIR, int device = -1,
	                           Transport transport = TRANSPORT_RCCL, uint64_t batchBases = (uint64_t)1 << 28)
	    : Base(typename Base::WorldTag(), world, estimatedRawKmers, separateSingletons, valueKind, device), _comm((MPI_Comm)world), _transport(transport), _batchBases(batchBases), _exchangeReady(false) {
		MPI_Comm_rank(_comm, &this->_rank); MPI_Comm_size(_comm, &this->_worldSize);
		this->_estimatedRawKmers = estimatedRawKmers * (unsigned long)this->_worldSize;      /* kmr_config takes the whole job's figure */
	}
	virtual void buildKmerSpectrum(const ReadSet &store) { buildKmerSpectrum(store, false); }
	virtual void buildKmerSpectrum(const ReadSet &store, bool isSolid) {
		if (isSolid) throw std::runtime_error("GpuDistributedKmerSpectrum: the solid map is not built on this path");
		this->ensureHandle();
		kmr_handle *h = this->handle();
		this->check(kmr_reset(h), "kmr_reset");
		joinExchange(h);
		/* batches of about _batchBases bases: [lo, hi) read ranges of this rank; every rank makes as many steps as the busiest one */
		std::vector<ReadSet::ReadSetSizeType> cuts(1, 0);
		uint64_t acc = 0;
		for (ReadSet::ReadSetSizeType i = 0; i < store.getSize(); i++) {
			acc += store.getRead(i).getLength();
			if (acc >= _batchBases) { cuts.push_back(i + 1); acc = 0; }
		}
		if (cuts.back() != store.getSize()) cuts.push_back(store.getSize());
		unsigned long long mySteps = cuts.size() - 1, steps = 0;
		MPI_Allreduce(&mySteps, &steps, 1, MPI_UNSIGNED_LONG_LONG, MPI_MAX, _comm);
		const uint64_t first = store.getGlobalOffset(this->_rank);
		int failed = KMR_OK; std::string why;
		for (unsigned long long sidx = 0; sidx < steps; sidx++) {
			kmr_reads *batch = NULL;
			int rc = KMR_OK;
			if (sidx < mySteps && failed == KMR_OK) {
				/* (the batch goes over as the ReadSet keeps it: packed bases + markups, kmr_reads_from_twobit; a discarded read is empty) */
				typename Base::PackedReads pr;
				Base::flattenTwoBit(store, pr, cuts[sidx], cuts[sidx + 1]);
				rc = kmr_reads_from_twobit(h, pr.twobit.empty() ? (const uint8_t *)"" : &pr.twobit[0], &pr.twobitOffsets[0], &pr.offsets[0],
				                           pr.markupPos.empty() ? NULL : &pr.markupOffsets[0], pr.markupPos.empty() ? NULL : &pr.markupPos[0], pr.markupPos.empty() ? NULL : &pr.markupChar[0],
				                           pr.anyQuals ? pr.quals.data() : NULL, 0, cuts[sidx + 1] - cuts[sidx], &batch);
				if (rc != KMR_OK) { failed = rc; why = kmr_last_error(h); batch = NULL; }
			}
			/* (a rank whose copy to the device failed still takes part in the step, with nothing: the library's steps are collective.  A
			 * failure INSIDE a step comes back on every rank -- the status word of kmr_exchange_add_reads_dev -- so all ranks stop together) */
			rc = kmr_exchange_add_read_batch(h, batch, first + (sidx < mySteps ? cuts[sidx] : 0));
			if (batch) kmr_reads_free(batch);
			/* (no break: an error that is this rank's alone -- a state check in front of the step's first gather -- must not leave the
			 * others waiting in a collective; the rank keeps taking part with empty batches until all steps are done) */
			if (rc != KMR_OK && failed == KMR_OK) { failed = rc; why = kmr_last_error(h); }
		}
		int anyFailed = failed != KMR_OK ? 1 : 0, jobFailed = 0;
		MPI_Allreduce(&anyFailed, &jobFailed, 1, MPI_INT, MPI_MAX, _comm);
		if (jobFailed) throw std::runtime_error("GpuDistributedKmerSpectrum::buildKmerSpectrum: " + (failed != KMR_OK ? why : std::string("another rank failed")));
		this->pull(KmerSpectrumOptions::getOptions().getMinDepth());
	}
private:
	/* once per device handle: the communicator of the exchange */
	void joinExchange(kmr_handle *h) {
		if (_exchangeReady) return;
		if (_transport == TRANSPORT_RCCL) {
			char id[KMR_EXCHANGE_ID_BYTES];
			int rc = KMR_OK;
			if (this->_rank == 0) rc = kmr_exchange_unique_id(id);
			MPI_Bcast(&rc, 1, MPI_INT, 0, _comm);
			if (rc != KMR_OK) throw std::runtime_error("kmr_exchange_unique_id failed on rank 0 (no RCCL?): construct with TRANSPORT_MPI");
			MPI_Bcast(id, KMR_EXCHANGE_ID_BYTES, MPI_BYTE, 0, _comm);
			this->check(kmr_exchange_init(h, id), "kmr_exchange_init");
		} else {
			kmr_transport t;
			t.user = this; t.allgather_u64 = &mpiAllgather; t.alltoallv_dev = &mpiAlltoallv;
			this->check(kmr_exchange_init_transport(h, &t), "kmr_exchange_init_transport");
		}
		_exchangeReady = true;
	}
	static int mpiAllgather(void *user, const uint64_t *mine, uint64_t n, uint64_t *all) {
		GpuDistributedKmerSpectrum *self = (GpuDistributedKmerSpectrum *)user;
		return MPI_Allgather(const_cast<uint64_t *>(mine), (int)n, MPI_UINT64_T, all, (int)n, MPI_UINT64_T, self->_comm) == MPI_SUCCESS ? KMR_OK : KMR_ERR_HIP;
	}
	/* device segments through host memory: 64-bit sizes and offsets throughout, one message per peer and call (<= 1 GiB each by the
	 * library's contract, checked), receives posted before the sends */
	static int mpiAlltoallv(void *user, const void *send, const uint64_t *sendOff, const uint64_t *sendBytes, void *recv, const uint64_t *recvOff, const uint64_t *recvBytes, void *) {
		GpuDistributedKmerSpectrum *self = (GpuDistributedKmerSpectrum *)user;
		kmr_handle *h = self->handle();
		const int R = self->_worldSize;
		std::vector<std::vector<char> > out(R), in(R);
		std::vector<MPI_Request> reqs;
		/* every size is looked at before anything is posted, and nothing returns while a request is outstanding (the buffers below
		 * would go out of scope under it, and the peers would wait for ever): a copy that fails still sends its -- then meaningless --
		 * bytes, the error is reported once everything has completed */
		for (int r = 0; r < R; r++) if (sendBytes[r] > ((uint64_t)1 << 30) || recvBytes[r] > ((uint64_t)1 << 30)) return KMR_ERR_CAPACITY;
		int failed = KMR_OK;
		for (int r = 0; r < R; r++)
			if (recvBytes[r]) { in[r].resize((size_t)recvBytes[r]); MPI_Request q; MPI_Irecv(&in[r][0], (int)recvBytes[r], MPI_BYTE, r, 7101, self->_comm, &q); reqs.push_back(q); }
		for (int r = 0; r < R; r++) {
			if (!sendBytes[r]) continue;
			out[r].resize((size_t)sendBytes[r]);
			const int rc = kmr_copy_to_host(h, &out[r][0], (const char *)send + sendOff[r], sendBytes[r]);
			if (rc != KMR_OK && failed == KMR_OK) failed = rc;
			MPI_Request q; MPI_Isend(&out[r][0], (int)sendBytes[r], MPI_BYTE, r, 7101, self->_comm, &q); reqs.push_back(q);
		}
		if (!reqs.empty() && MPI_Waitall((int)reqs.size(), &reqs[0], MPI_STATUSES_IGNORE) != MPI_SUCCESS && failed == KMR_OK) failed = KMR_ERR_HIP;
		if (failed != KMR_OK) return failed;
		for (int r = 0; r < R; r++) {
			if (!recvBytes[r]) continue;
			const int rc = kmr_copy_to_device(h, (char *)recv + recvOff[r], &in[r][0], recvBytes[r]);
			if (rc != KMR_OK) return rc;
		}
		return KMR_OK;
	}
	MPI_Comm _comm;
	Transport _transport;
	uint64_t _batchBases;
	bool _exchangeReady;
};
#endif /* KMERNATOR_AMD_SHIM_MPI */

#endif
