/*
 * kmernator_amd_shim.hpp -- the reference-side binding a Kmernator maintainer adds.
 *
 * Include AFTER Kmernator's own "KmerSpectrum.h" (it needs ReadSet, KmerSizer, the map
 * types and the option singletons; it is therefore not compiled inside this repository,
 * where Boost and the reference headers are absent -- see INTEGRATION.md).
 *
 *   typedef KmerSpectrum<DataType, DataType, SingletonDataType> KS;       // apps/FilterReads.h:246
 *   typedef GpuKmerSpectrum<KS> GKS;
 *   GKS spectrum(rawKmers);                                               // apps/FilterReads.cpp:136
 *   spectrum.buildKmerSpectrum(reads);                                    // apps/FilterReads.cpp:139
 *
 * buildKmerSpectrum() flattens the ReadSet into the arrays kmr_add_reads takes, runs the
 * build on the MI355X, and materialises weak (and singleton) as the reference's OWN map
 * objects by KmerMapByKmerArrayPair::restore() (src/Kmer.h:3160-3173) over the images the
 * library writes in the store() layout (src/Kmer.h:3143-3159).  Everything downstream
 * (ReadSelector, histogram, storeMmap) then runs unchanged on the reference types.
 */
#ifndef KMERNATOR_AMD_SHIM_HPP_
#define KMERNATOR_AMD_SHIM_HPP_

#include <stdexcept>
#include <string>
#include <vector>

#include "kmernator_amd.h"

template <typename KS>
class GpuKmerSpectrum : public KS {
public:
	typedef typename KS::WeakMapType WeakMapType;
	typedef typename KS::SingletonMapType SingletonMapType;

	GpuKmerSpectrum(unsigned long estimatedRawKmers, bool separateSingletons = true, int valueKind = KMR_VALUE_COUNT_DIR)
	    : KS(estimatedRawKmers, separateSingletons), _h(NULL) {
		kmr_config c;
		kmr_config_init(&c);
		c.k = KmerSizer::getSequenceLength();                                   // src/Kmer.h:115
		c.num_buckets_weak = this->weak.getNumBuckets();                        // the ctor already sized them
		c.num_buckets_singleton = this->singleton.getNumBuckets();
		c.estimated_raw_kmers = estimatedRawKmers;
		c.value_kind = valueKind;
		c.min_weight = TrackingData::getMinimumWeight();                        // src/KmerTrackingData.h:377
		c.min_quality_score = GeneralOptions::getOptions().getMinQuality();     // src/Options.h
		c.fastq_start_char = Read::FASTQ_START_CHAR;                            // reads are already rescaled to it
		c.ext_min_quality = ExtensionTracking::getMinQuality();
		c.separate_singletons = separateSingletons ? 1 : 0;
		c.kmer_subsample = KS::getKmerSubsample();
		check(kmr_create(&c, &_h), "kmr_create");
	}
	virtual ~GpuKmerSpectrum() { kmr_destroy(_h); }

	// replaces KmerSpectrum::buildKmerSpectrum(const ReadSet&, bool) (src/KmerSpectrum.h:2085-2115)
	virtual void buildKmerSpectrum(const ReadSet &store) { buildKmerSpectrum(store, false); }
	virtual void buildKmerSpectrum(const ReadSet &store, bool isSolid) {
		if (isSolid) { KS::buildKmerSpectrum(store, isSolid); return; }       // solid map: not on this path
		std::string bases, quals;
		std::vector<uint64_t> offsets(1, 0);
		std::vector<uint8_t> discarded;
		bool anyQuals = false;
		for (ReadSet::ReadSetSizeType i = 0; i < store.getSize(); i++) {
			const Read &read = store.getRead(i);
			discarded.push_back(read.isDiscarded() ? 1 : 0);
			if (!read.isDiscarded()) {
				std::string f = read.getFasta();                                    // markups applied: N/X become non-ACGT chars
				std::string q = read.getQuals();                                    // REF_QUAL string for reads without quals
				bases += f; quals += q; anyQuals = true;
			}
			offsets.push_back(bases.size());
		}
		check(kmr_reset(_h), "kmr_reset");
		check(kmr_add_reads(_h, bases.data(), anyQuals ? quals.data() : NULL, offsets.data(), store.getSize(), 0, discarded.data()), "kmr_add_reads");
		pull(KmerSpectrumOptions::getOptions().getMinDepth());
	}

	// per-k-mer lookups stay on the restored reference maps; batch form for ReadSelector-style consumers:
	void getCounts(const std::vector<uint8_t> &packedCanonicalKmers, std::vector<uint32_t> &counts) {
		size_t kb = KmerSizer::getByteSize();
		counts.resize(packedCanonicalKmers.size() / kb);
		check(kmr_lookup(_h, packedCanonicalKmers.data(), counts.size(), counts.data()), "kmr_lookup");
	}

private:
	// purgeMinDepth + materialise: the images must outlive the maps that alias them (src/KmerSpectrum.h:1817)
	void pull(unsigned int minDepth) {
		check(kmr_finalize(_h, minDepth), "kmr_finalize");
		uint64_t n = 0;
		check(kmr_image_size(_h, KMR_MAP_WEAK, &n), "kmr_image_size");
		_weakImage.resize(n);
		check(kmr_write_image(_h, KMR_MAP_WEAK, _weakImage.data(), n), "kmr_write_image");
		WeakMapType w = WeakMapType::restore(_weakImage.data());
		this->weak.swap(w);
		if (minDepth <= 1 && this->hasSingletons) {
			check(kmr_image_size(_h, KMR_MAP_SINGLETON, &n), "kmr_image_size");
			_singletonImage.resize(n);
			check(kmr_write_image(_h, KMR_MAP_SINGLETON, _singletonImage.data(), n), "kmr_write_image");
			SingletonMapType s = SingletonMapType::restore(_singletonImage.data());
			this->singleton.swap(s);
		} else {
			this->singleton.clear(false);
			this->hasSingletons = false;
		}
		kmr_stats st;
		check(kmr_get_stats(_h, &st), "kmr_get_stats");
		this->rawKmers = st.raw_kmers; this->rawGoodKmers = st.raw_good_kmers;
		this->uniqueKmers = st.unique_kmers; this->singletonKmers = st.singleton_kmers;
	}
	// The artifact filter on the device, in place of FilterKnownOddities::applyFilter(reads) at apps/FilterReads.cpp:110-114:
	//     spectrum.applyArtifactFilter(reads, FilterKnownOddities::getArtifactFasta() [+ repeat / PhiX tables], cfg);
	// The screen runs on the GPU; trims, discards and remnant reads are applied to the reference's ReadSet exactly as
	// Recorder::recordTrim / recordDiscard (src/FilterKnownOddities.h:310-334) and applyFilterToRead :519-528 do.
	unsigned long applyArtifactFilter(ReadSet &reads, const std::string &artifactFasta, const kmr_artifact_config &cfg) {
		std::string bases, quals;
		std::vector<uint64_t> offsets(1, 0);
		for (ReadSet::ReadSetSizeType i = 0; i < reads.getSize(); i++) {
			const Read &read = reads.getRead(i);
			bases += read.getFasta(); quals += read.getQuals();
			offsets.push_back(bases.size());
		}
		std::vector<int64_t> mate;
		if (reads.hasPairs()) {
			mate.assign(reads.getSize(), -1);
			for (ReadSet::ReadSetSizeType p = 0; p < reads.getPairSize(); p++) {
				const ReadSet::Pair &pair = reads.getPair(p);
				if (reads.isValidRead(pair.read1) && reads.isValidRead(pair.read2)) { mate[pair.read1] = pair.read2; mate[pair.read2] = pair.read1; }
			}
		}
		kmr_reads *batch = NULL; kmr_artifact_filter *filter = NULL;
		check(kmr_reads_from_host(_h, bases.data(), quals.data(), offsets.data(), reads.getSize(), &batch), "kmr_reads_from_host");
		int rc = kmr_artifact_filter_create(_h, &cfg, artifactFasta.data(), artifactFasta.size(), &filter);
		const size_t n = reads.getSize();
		std::vector<uint32_t> value(n), minPass(n), maxPass(n), remOff(n), remLen(n); std::vector<uint8_t> action(n);
		if (rc == KMR_OK) rc = kmr_artifact_filter_apply(_h, filter, batch, mate.empty() ? NULL : mate.data(), value.data(), minPass.data(), maxPass.data(),
		                                                 action.data(), remOff.data(), remLen.data(), NULL);
		kmr_artifact_filter_free(filter); kmr_reads_free(batch);
		check(rc, "kmr_artifact_filter");
		unsigned long affected = 0;
		ReadSet remnants;
		for (size_t i = 0; i < n; i++) {
			Read &read = reads.getRead(i);
			if (remLen[i]) remnants.append(read.getTrimRead(remOff[i], remLen[i], "AFTrim:" + boost::lexical_cast<std::string>(remOff[i]) + "+" + boost::lexical_cast<std::string>(remLen[i]), "-qtrim"));
			if (action[i] == 1) { read = read.getTrimRead(minPass[i], maxPass[i] - minPass[i], "AFTrim:" + boost::lexical_cast<std::string>(minPass[i]) + "+" + boost::lexical_cast<std::string>(maxPass[i] - minPass[i])); affected++; }
			else if (action[i] == 2) read.discard();
		}
		if (remnants.getSize() > 0) reads.append(remnants);
		return affected;
	}

	void check(int rc, const char *what) {
		if (rc != KMR_OK) throw std::runtime_error(std::string(what) + ": " + kmr_last_error(_h));
	}
	kmr_handle *_h;
	std::vector<char> _weakImage, _singletonImage;
};

#endif
