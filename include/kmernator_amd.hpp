/*
 * kmernator_amd.hpp -- C++ host side above the C-ABI (include/kmernator_amd.h), for callers that do not
 * pull in Kmernator's own headers (those need Boost; the binding for a Kmernator checkout is
 * kmernator_amd_shim.hpp).  Method names, argument meaning and error behaviour follow the reference:
 *
 *   kmernator::KmerSpectrum   KmerSpectrum<So,We,Si>                       src/KmerSpectrum.h
 *     buildKmerSpectrum       buildKmerSpectrum(const ReadSet&)           :2081-2115 (flat arrays or a device ReadSet)
 *     purgeMinDepth           purgeMinDepth + optimize                     :1805-1815, :460-466
 *     getCount                getCount(kmer, false)                        :701-725
 *     getRawKmers ...         :455-459
 *     subtractReference       :472-474
 *     getHistogram            Histogram(256).set(*this) + toString         :909-1071
 *     storeMmap / restoreMmap :476-518 (file naming: <name>, <name>-singleton)
 *     dumpCounts / dumpGraphs MeraculousDistributedKmerSpectrum            src/Meraculous.h:107-134
 *     scoreAndTrimReads       ReadSelector::scoreAndTrimReads              src/ReadSelector.h:1182-1207
 *   kmernator::ReadSet        ReadSet::appendFastq... on the device        src/ReadSet.cpp:311-345
 *
 * Errors surface as kmernator::KmerSpectrumError (the reference throws LoggedException, src/Log.h:442-484).
 * Header only; link with -lkmernator_amd.
 */
#ifndef KMERNATOR_AMD_HPP_
#define KMERNATOR_AMD_HPP_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <memory>
#include <string>
#include <vector>

#include "kmernator_amd.h"

namespace kmernator {

struct KmerSpectrumError : std::runtime_error {
	int code;
	KmerSpectrumError(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

class KmerSpectrum;

/* KmerSpectrum::Histogram: buckets filled on the device, finish()/toString() as :986-1035 */
class Histogram {
public:
	unsigned int zoomMax; double logBase;
	std::vector<uint64_t> visits, visitedCount, cumulativeVisits; std::vector<double> visitedWeight;
	uint64_t count = 0; double totalCount = 0, totalWeightedCount = 0; unsigned int lastBucket = 0;
	Histogram(unsigned int z, double b) : zoomMax(z), logBase(b) {}
	unsigned int getBucketValue(unsigned int idx) const {
		const unsigned int skip = (unsigned int)(std::log((double)zoomMax + 1.0) / std::log(logBase) - 1.0);
		return idx <= zoomMax ? idx : (unsigned int)std::pow(logBase, (double)(idx + skip - zoomMax));
	}
	void finish() {
		count = 0; totalCount = totalWeightedCount = 0; lastBucket = 0;
		cumulativeVisits.assign(visits.size(), 0);
		for (size_t i = visits.size(); i-- > 0;) {
			cumulativeVisits[i] = count += visits[i];
			if (visits[i]) { totalCount += (double)visitedCount[i]; totalWeightedCount += visitedWeight[i]; if (i > lastBucket) lastBucket = (unsigned int)i; }
		}
	}
	std::string toString() {
		finish();
		std::ostringstream ss; ss.setf(std::ios::fixed); ss.precision(3);
		ss << "Counts, Weights and Directions\n";
		ss << "Counts:\t" << count << "\t" << totalCount << "\t" << (totalCount / count) << "\t\n";
		ss << "Weights:\t" << count << "\t" << totalWeightedCount << "\t" << (totalWeightedCount / count) << "\t" << (totalWeightedCount / totalCount) << "\n\n";
		ss << "Bucket\tCumulative\tUnique\t%Unique\tCount\t%Count\tWeight\tQualProb\t%Weight\n";
		for (unsigned int i = 1; i < lastBucket + 1; i++)
			ss << getBucketValue(i) << "\t" << cumulativeVisits[i] << "\t" << visits[i] << "\t" << 100.0 * visits[i] / count << "\t" << visitedCount[i] << "\t"
			   << 100.0 * visitedCount[i] / totalCount << "\t\t" << visitedWeight[i] << "\t" << visitedWeight[i] / visitedCount[i] << "\t" << 100.0 * visitedWeight[i] / totalWeightedCount << "\t\n";
		return ss.str();
	}
};

/* reads parsed from FASTQ text on the device, bound to the device of the spectrum that parsed them */
class ReadSet {
public:
	ReadSet(KmerSpectrum &sp, const std::string &fastqText, uint32_t inputQualityBase = 0, bool storeComment = true);
	~ReadSet() { kmr_reads_free(_r); }
	ReadSet(const ReadSet &) = delete; ReadSet &operator=(const ReadSet &) = delete;
	uint64_t getSize() const { return _n; }
	uint64_t getBaseCount() const { return _bases; }
	uint32_t getInputQualityBase() const { return _qbase; }
	uint64_t getFiltered() const { return _filtered; }
	std::string getName(uint64_t i) const { return _text.substr(_nameOff[i], _nameLen[i]); }
	const kmr_reads *raw() const { return _r; }
private:
	friend class FilterKnownOddities;
	ReadSet(const std::string &text, kmr_reads *r) : _text(text), _r(r) { load(); }      /* a batch the library derived from another one */
	void load();
	std::string _text; kmr_reads *_r = nullptr; uint64_t _n = 0, _bases = 0, _filtered = 0; uint32_t _qbase = 0;
	std::vector<uint64_t> _nameOff; std::vector<uint32_t> _nameLen;
};

class KmerSpectrum {
public:
	enum ScoringType { KS_SUM = 0, KS_MEDIAN = 1, KS_MIN = 2, KS_MAX = 3, KS_AVG = 4 };
	static kmr_config defaults(uint32_t k, uint64_t estimatedRawKmers) { kmr_config c; kmr_config_init(&c); c.k = k; c.estimated_raw_kmers = estimatedRawKmers; return c; }
	explicit KmerSpectrum(const kmr_config &cfg) : _cfg(cfg) {
		const int rc = kmr_create(&cfg, &_h);
		if (rc != KMR_OK) throw KmerSpectrumError(rc, std::string("kmr_create: ") + kmr_last_error(nullptr));
	}
	~KmerSpectrum() { kmr_destroy(_h); }
	KmerSpectrum(const KmerSpectrum &) = delete; KmerSpectrum &operator=(const KmerSpectrum &) = delete;

	uint32_t k() const { return _cfg.k; }
	uint32_t keyBytes() const { return (_cfg.k + 3) / 4; }
	kmr_handle *raw() { return _h; }
	const kmr_config &config() const { return _cfg; }

	void reset() { check(kmr_reset(_h), "kmr_reset"); }
	void buildKmerSpectrum(const char *bases, const char *quals, const uint64_t *offsets, uint64_t nReads, uint64_t firstReadIdx = 0, const uint8_t *discarded = nullptr) {
		check(kmr_add_reads(_h, bases, quals, offsets, nReads, firstReadIdx, discarded), "kmr_add_reads");
	}
	void buildKmerSpectrum(const ReadSet &reads, uint64_t firstReadIdx = 0) { check(kmr_add_read_batch(_h, reads.raw(), firstReadIdx), "kmr_add_read_batch"); }
	/* N GPUs, one process (or thread) per GPU, no MPI: DistributedKmerSpectrum::buildKmerSpectrum (src/DistributedFunctions.h:340-458)
	 * over RCCL inside the library.  Rank 0 makes the id and the host hands it to the others; config.rank / world_size say who is who;
	 * every rank calls buildKmerSpectrumExchange the same number of times (nullptr: no reads this round), then purgeMinDepth. */
	static std::vector<uint8_t> exchangeUniqueId() {
		std::vector<uint8_t> id(KMR_EXCHANGE_ID_BYTES);
		const int rc = kmr_exchange_unique_id(id.data());
		if (rc != KMR_OK) throw KmerSpectrumError(rc, std::string("kmr_exchange_unique_id: ") + kmr_last_error(nullptr));
		return id;
	}
	void exchangeInit(const std::vector<uint8_t> &id) {
		if (id.size() != KMR_EXCHANGE_ID_BYTES) throw KmerSpectrumError(KMR_ERR_INVALID_ARG, "exchangeInit: the id has KMR_EXCHANGE_ID_BYTES bytes");
		check(kmr_exchange_init(_h, id.data()), "kmr_exchange_init");
	}
	void buildKmerSpectrumExchange(const ReadSet *reads, uint64_t firstReadIdx = 0) { check(kmr_exchange_add_read_batch(_h, reads ? reads->raw() : nullptr, firstReadIdx), "kmr_exchange_add_read_batch"); }
	void subtractReference(KmerSpectrum *other) { check(kmr_subtract_reference(_h, other ? other->_h : nullptr), "kmr_subtract_reference"); }
	void purgeMinDepth(uint32_t minDepth) { check(kmr_finalize(_h, minDepth), "kmr_finalize"); }

	kmr_stats stats() { kmr_stats s; check(kmr_get_stats(_h, &s), "kmr_get_stats"); return s; }
	uint64_t getRawKmers() { return stats().raw_kmers; }
	uint64_t getRawGoodKmers() { return stats().raw_good_kmers; }
	uint64_t getUniqueKmers() { return stats().unique_kmers; }
	uint64_t getSingletonKmers() { return stats().singleton_kmers; }

	/* packed canonical k-mers, keyBytes() each */
	std::vector<uint32_t> getCount(const std::vector<uint8_t> &packedKmers) {
		std::vector<uint32_t> out(packedKmers.size() / keyBytes());
		if (!out.empty()) check(kmr_lookup(_h, packedKmers.data(), out.size(), out.data()), "kmr_lookup");
		return out;
	}
	Histogram getHistogram(unsigned int zoomMax = 256, double logBase = 2.0) {
		Histogram h(zoomMax, logBase);
		const uint32_t nb = kmr_histogram_bins(zoomMax);
		h.visits.resize(nb); h.visitedCount.resize(nb); h.visitedWeight.resize(nb);
		check(kmr_histogram(_h, zoomMax, logBase, h.visits.data(), h.visitedCount.data(), h.visitedWeight.data(), nb), "kmr_histogram");
		h.finish();
		return h;
	}
	struct TrimResult { std::vector<uint32_t> trimOffset, trimLength; std::vector<float> score; std::vector<uint8_t> wasTrimmed; };
	TrimResult scoreAndTrimReads(const ReadSet &reads, double minimumKmerScore, ScoringType t = KS_MEDIAN) {
		TrimResult r; const uint64_t n = reads.getSize();
		r.trimOffset.resize(n); r.trimLength.resize(n); r.score.resize(n); r.wasTrimmed.resize(n);
		if (n) check(kmr_score_read_batch(_h, reads.raw(), minimumKmerScore, (int)t, r.trimOffset.data(), r.trimLength.data(), r.score.data(), r.wasTrimmed.data()), "kmr_score_read_batch");
		return r;
	}

	std::vector<uint8_t> image(int whichMap) {
		uint64_t n = 0; check(kmr_image_size(_h, whichMap, &n), "kmr_image_size");
		std::vector<uint8_t> buf(n);
		check(kmr_write_image(_h, whichMap, buf.data(), n), "kmr_write_image");
		return buf;
	}
	void storeMmap(const std::string &filename, uint32_t minDepth) {
		write(filename, image(KMR_MAP_WEAK));
		if (minDepth <= 1) write(filename + "-singleton", image(KMR_MAP_SINGLETON));
	}
	void restoreMmap(const std::string &filename) {
		bool loaded = false;
		std::vector<uint8_t> b;
		if (read(filename, b)) { check(kmr_load_image(_h, KMR_MAP_WEAK, b.data(), b.size()), "kmr_load_image"); loaded = true; }
		if (read(filename + "-singleton", b)) { check(kmr_load_image(_h, KMR_MAP_SINGLETON, b.data(), b.size()), "kmr_load_image"); loaded = true; }
		if (!loaded) throw KmerSpectrumError(KMR_ERR_INVALID_ARG, "Terribly sorry but there were no kmer spectrum mmap files at: " + filename + "*");
	}
	void dumpCounts(const std::string &filename, uint32_t minDepth) { check(kmr_dump_mercount(_h, filename.c_str(), minDepth), "kmr_dump_mercount"); }
	void dumpGraphs(const std::string &filename, uint32_t minDepth) { check(kmr_dump_mergraph(_h, filename.c_str(), minDepth), "kmr_dump_mergraph"); }

	void check(int rc, const char *what) const { if (rc != KMR_OK) throw KmerSpectrumError(rc, std::string(what) + ": " + kmr_last_error(_h)); }
private:
	static void write(const std::string &f, const std::vector<uint8_t> &b) { std::ofstream o(f, std::ios::binary); o.write((const char *)b.data(), (std::streamsize)b.size()); if (!o) throw KmerSpectrumError(KMR_ERR_INVALID_ARG, "cannot write " + f); }
	static bool read(const std::string &f, std::vector<uint8_t> &b) {
		std::ifstream i(f, std::ios::binary | std::ios::ate); if (!i) return false;
		const std::streamsize n = i.tellg(); if (n <= 0) return false;
		b.resize((size_t)n); i.seekg(0); i.read((char *)b.data(), n); return (bool)i;
	}
	kmr_config _cfg; kmr_handle *_h = nullptr;
	friend class ReadSet;
	friend class FilterKnownOddities;
};

inline ReadSet::ReadSet(KmerSpectrum &sp, const std::string &fastqText, uint32_t inputQualityBase, bool storeComment) : _text(fastqText) {
	sp.check(kmr_ingest_fastq(sp._h, _text.data(), _text.size(), inputQualityBase, storeComment ? 1 : 0, &_r), "kmr_ingest_fastq");
	load();
}
inline void ReadSet::load() {
	kmr_reads_info(_r, &_n, &_bases, &_qbase, &_filtered);
	_nameOff.resize(_n ? _n : 1); _nameLen.resize(_n ? _n : 1);
	if (kmr_reads_copy(_r, nullptr, nullptr, nullptr, _nameOff.data(), _nameLen.data()) != KMR_OK) throw KmerSpectrumError(KMR_ERR_HIP, "kmr_reads_copy");
}

/* FilterKnownOddities (src/FilterKnownOddities.h): the artifact screen FilterReads runs before the spectrum build */
class FilterKnownOddities {
public:
	struct Results {                      /* per read, FilterResults (:289-296) + what recordAffectedRead did with it */
		std::vector<uint32_t> value, minPass, maxPass, remnantOffset, remnantLength;
		std::vector<uint8_t> action;      /* 0 untouched, 1 trimmed ("AFTrim:<minPass>+<maxPass-minPass>"), 2 discarded */
	};
	static kmr_artifact_config defaults(const kmr_config &sp) { kmr_artifact_config c; kmr_artifact_config_init(&c); c.fastq_start_char = sp.fastq_start_char; c.min_quality = sp.min_quality_score; return c; }
	FilterKnownOddities(KmerSpectrum &sp, const std::string &artifactFasta, const kmr_artifact_config &cfg) : _sp(sp) {
		sp.check(kmr_artifact_filter_create(sp.raw(), &cfg, artifactFasta.data(), artifactFasta.size(), &_f), "kmr_artifact_filter_create");
	}
	~FilterKnownOddities() { kmr_artifact_filter_free(_f); }
	FilterKnownOddities(const FilterKnownOddities &) = delete; FilterKnownOddities &operator=(const FilterKnownOddities &) = delete;
	uint64_t getFilterSize() const { uint64_t n = 0; kmr_artifact_filter_info(_f, nullptr, &n, nullptr); return n; }
	/* applyFilter(ReadSet&) (:663-733): the reads afterwards (trimmed in place, remnants appended); mate = paired read or -1 */
	std::unique_ptr<ReadSet> applyFilter(const ReadSet &reads, Results &res, const int64_t *mate = nullptr) {
		const uint64_t n = reads.getSize(), m = n ? n : 1;
		res.value.assign(m, 0); res.minPass.assign(m, 0); res.maxPass.assign(m, 0); res.remnantOffset.assign(m, 0); res.remnantLength.assign(m, 0); res.action.assign(m, 0);
		kmr_reads *out = nullptr;
		_sp.check(kmr_artifact_filter_apply(_sp.raw(), _f, reads.raw(), mate, res.value.data(), res.minPass.data(), res.maxPass.data(), res.action.data(),
		                                    res.remnantOffset.data(), res.remnantLength.data(), &out), "kmr_artifact_filter_apply");
		return std::unique_ptr<ReadSet>(new ReadSet(reads._text, out));
	}
private:
	KmerSpectrum &_sp; kmr_artifact_filter *_f = nullptr;
};

}  // namespace kmernator
#endif
