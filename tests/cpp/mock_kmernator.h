/*
 * mock_kmernator.h -- A MOCK.  Test infrastructure for include/kmernator_amd_shim.hpp only.
 *
 * The reference's headers need Boost 1.53 and sparsehash, which this image lacks, so the shim cannot be compiled against
 * them here.  This file DECLARES the few reference names the shim touches, with the signatures and access levels they have in
 * the reference (file:line under the reference checkout, src/KmerSpectrum.h:404 patched to `protected:` as the shim's header
 * says), over toy bodies: Read / ReadSet hold plain strings, the map types parse the store() image (src/Kmer.h:3143-3159)
 * into a std::map.  Compiling the shim against it catches syntax, access-control and ownership mistakes (the reference
 * assigns spectra by value, apps/FilterReads.cpp:126,136).  It pins NOTHING about the reference's behaviour.
 */
#ifndef MOCK_KMERNATOR_H_
#define MOCK_KMERNATOR_H_

#include <stdint.h>
#include <string.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

namespace Kmernator { typedef uint32_t ReadSetSizeType; }

struct KmerSizer {                                        /* src/Kmer.h:83-127 */
	static uint32_t &k() { static uint32_t v = 21; return v; }
	static inline uint32_t getSequenceLength() { return k(); }                   /* :117 */
	static inline uint32_t getByteSize() { return (k() + 3) / 4; }               /* :123 */
	static void set(uint32_t v) { k() = v; }                                     /* :107 */
};
class ExtensionTracking {                                 /* src/KmerTrackingData.h:153 */
public:
	static unsigned char &getMinQuality() { static unsigned char q = 20; return q; }   /* :157 */
};
class TrackingData {                                      /* src/KmerTrackingData.h:299 */
public:
	typedef float WeightType;
	static WeightType &minimumWeight() { static WeightType w = 0.10f; return w; }
	static inline WeightType getMinimumWeight() { return minimumWeight(); }     /* :377 */
};
class _GeneralOptions {                                   /* src/Options.h:325 */
public:
	unsigned int &getMinQuality() { static unsigned int q = 3; return q; }       /* :622 */
};
struct GeneralOptions { static _GeneralOptions &getOptions() { static _GeneralOptions o; return o; } };   /* OptionsBaseTemplate, src/Options.h:199-225 */
class _KmerSpectrumOptions {                              /* src/KmerSpectrum.h:92-240 */
public:
	unsigned int &getMinDepth() { static unsigned int d = 2; return d; }         /* :187 */
	long &getKmerSubsample() { static long s = 1; return s; }
};
struct KmerSpectrumOptions { static _KmerSpectrumOptions &getOptions() { static _KmerSpectrumOptions o; return o; } };

class Read {                                              /* src/Sequence.h */
public:
	static uint8_t FASTQ_START_CHAR;                                              /* :68 */
	Read() : _discarded(false) {}
	Read(const std::string &name, const std::string &fasta, const std::string &quals) : _name(name), _fasta(fasta), _quals(quals), _discarded(false) {}
	inline bool isDiscarded() const { return _discarded; }                        /* :243 */
	inline void discard() const { _discarded = true; }                            /* :254: "permitted even on a constant" */
	uint32_t getLength() const { return (uint32_t)_fasta.size(); }                /* :258 (SequenceLengthType) */
	inline bool hasQuals() const { return !_quals.empty(); }                      /* :241 */
	/* the form the reference keeps a sequence in (:166-171): TwoBitSequence::compressSequence of the text (src/TwoBitSequence.cpp:242-269) */
	typedef unsigned char TwoBitEncoding;                                         /* src/TwoBitSequence.h:66 */
	typedef std::pair<char, uint32_t> BaseLocationType;                           /* src/TwoBitSequence.h:80-81 */
	typedef std::vector<BaseLocationType> BaseLocationVectorType;
	uint32_t getTwoBitEncodingSequenceLength() const { return (uint32_t)((_fasta.size() + 3) / 4); }   /* :287 */
	const TwoBitEncoding *getTwoBitSequence() const { pack(); return _twobit.empty() ? NULL : &_twobit[0]; }   /* :289 */
	BaseLocationVectorType getMarkups() const { pack(); return _markups; }       /* :280 */
	std::string getFasta(uint32_t trimOffset = 0, uint32_t trimLength = 0xffffffffu) const { return trimOffset < _fasta.size() ? _fasta.substr(trimOffset, trimLength) : std::string(); }   /* :273 */
	std::string getQuals(uint32_t trimOffset = 0, uint32_t trimLength = 0xffffffffu) const { return trimOffset < _quals.size() ? _quals.substr(trimOffset, trimLength) : std::string(); }   /* :498 */
	Read getTrimRead(uint32_t trimOffset, uint32_t trimLength, std::string label = "", std::string nameSuffix = "", bool unmasked = false) {   /* :485 */
		(void)unmasked;
		return Read(_name + nameSuffix + (label.empty() ? "" : " " + label), getFasta(trimOffset, trimLength), getQuals(trimOffset, trimLength));
	}
	const std::string &getName() const { return _name; }
private:
	void pack() const {
		if (_twobit.size() == (_fasta.size() + 3) / 4 && (!_twobit.empty() || _fasta.empty())) return;
		_twobit.assign((_fasta.size() + 3) / 4, 0); _markups.clear();
		for (size_t i = 0; i < _fasta.size(); i++) {
			unsigned code = 0;
			switch (_fasta[i]) { case 'A': case 'a': code = 0; break; case 'C': case 'c': code = 1; break; case 'G': case 'g': code = 2; break; case 'T': case 't': code = 3; break;
			default: _markups.push_back(BaseLocationType(_fasta[i] == '.' ? 'N' : _fasta[i], (uint32_t)i)); }
			_twobit[i >> 2] = (TwoBitEncoding)(_twobit[i >> 2] | (code << (6 - 2 * (i & 3))));
		}
	}
	std::string _name, _fasta, _quals;
	mutable std::vector<TwoBitEncoding> _twobit; mutable BaseLocationVectorType _markups;
	mutable bool _discarded;
};

class ReadSet {                                           /* src/ReadSet.h */
public:
	typedef Kmernator::ReadSetSizeType ReadSetSizeType;                            /* :66 */
	class Pair { public: ReadSetSizeType read1, read2; Pair() : read1(0xffffffffu), read2(0xffffffffu) {} Pair(ReadSetSizeType a, ReadSetSizeType b) : read1(a), read2(b) {} };   /* :95-131 */
	void append(const ReadSet &reads) { for (size_t i = 0; i < reads._reads.size(); i++) _reads.push_back(reads._reads[i]); }   /* :380 */
	void append(const Read &read) { _reads.push_back(read); }                     /* :381 */
	inline ReadSetSizeType getSize() const { return (ReadSetSizeType)_reads.size(); }   /* :383 */
	inline ReadSetSizeType getPairSize() const { return (ReadSetSizeType)_pairs.size(); }   /* :390 */
	inline ReadSetSizeType getGlobalOffset(int) const { return 0; }               /* :433 */
	inline bool isValidRead(ReadSetSizeType index) const { return index < getSize(); }   /* :502 */
	inline const Read &getRead(ReadSetSizeType index) const { return _reads[index]; }    /* :507 */
	inline Read &getRead(ReadSetSizeType index) { return _reads[index]; }         /* :512 */
	inline bool hasPairs() const { return getPairSize() != 0 && getPairSize() < getSize(); }   /* :526 */
	inline Pair &getPair(ReadSetSizeType pairIndex) { return _pairs[pairIndex]; } /* :537 */
	inline const Pair &getPair(ReadSetSizeType pairIndex) const { return _pairs[pairIndex]; }   /* :540 */
	std::vector<Pair> &pairs() { return _pairs; }
private:
	std::vector<Read> _reads;
	std::vector<Pair> _pairs;
};

/* KmerMapByKmerArrayPair<V> (src/Kmer.h:2799-3279) as far as the shim uses it: bucket count, swap, clear, and the copying
 * constructor from a store() image (:3124-3135).  V is the value struct's size in bytes. */
template <int VBYTES>
class MockKmerMap {
public:
	MockKmerMap() : _numBuckets(0) {}
	MockKmerMap(unsigned long bucketCount) : _numBuckets(1) { while (_numBuckets < bucketCount) _numBuckets <<= 1; }   /* :2837 + resizeBuckets :2224 */
	MockKmerMap(const void *src) : _numBuckets(0) {                              /* :3124 */
		const uint8_t *p = (const uint8_t *)src;
		uint64_t nb, mask; memcpy(&nb, p, 8); memcpy(&mask, p + 8, 8);
		_numBuckets = nb;
		const uint32_t kb = KmerSizer::getByteSize();
		for (uint64_t b = 0; b < nb; b++) {
			uint64_t off; memcpy(&off, p + 16 + 8 * b, 8);
			uint32_t n; memcpy(&n, p + off, 4);
			const uint8_t *keys = p + off + 4, *vals = keys + (size_t)n * kb;
			for (uint32_t i = 0; i < n; i++) _entries[std::string((const char *)keys + (size_t)i * kb, kb)] = std::string((const char *)vals + (size_t)i * VBYTES, VBYTES);
		}
	}
	uint64_t getNumBuckets() const { return _numBuckets; }                        /* :2264 */
	void swap(MockKmerMap &other) { std::swap(_numBuckets, other._numBuckets); _entries.swap(other._entries); }   /* :2880 */
	void clear(bool releaseMemory = true) { (void)releaseMemory; _entries.clear(); }   /* :2237 */
	size_t size() const { return _entries.size(); }
	const std::map<std::string, std::string> &entries() const { return _entries; }
private:
	uint64_t _numBuckets;
	std::map<std::string, std::string> _entries;
};

template <typename So, typename We, typename Si>
class KmerSpectrum {                                      /* src/KmerSpectrum.h:345- */
public:
	typedef So SolidMapType; typedef We WeakMapType; typedef Si SingletonMapType;
	SolidMapType solid; WeakMapType weak; SingletonMapType singleton;             /* :396-398 */
	bool hasSolids, hasSingletons;                                                /* :399-400 */
protected:                                                /* `private:` at src/KmerSpectrum.h:404; the shim's patch makes it protected */
	long rawKmers, rawGoodKmers, uniqueKmers, singletonKmers, subtracted;         /* :405-409 */
public:
	KmerSpectrum() : hasSolids(false), hasSingletons(false), rawKmers(0), rawGoodKmers(0), uniqueKmers(0), singletonKmers(0), subtracted(0) {}   /* :413 */
	KmerSpectrum(unsigned long estimatedRawKmers, bool separateSingletons = true)   /* :414-421 */
	    : solid(), weak((unsigned long)(int)(estimatedRawKmers / 20.0) / 32 + 1), singleton((unsigned long)(separateSingletons ? estimatedRawKmers * 0.35 : 1) / 32 + 1),
	      hasSolids(false), hasSingletons(separateSingletons), rawKmers(0), rawGoodKmers(0), uniqueKmers(0), singletonKmers(0), subtracted(0) {}
	virtual ~KmerSpectrum() {}
	KmerSpectrum(const KmerSpectrum &copy) { *this = copy; }                      /* :423 */
	KmerSpectrum &operator=(const KmerSpectrum &other) {                          /* :426-440 */
		weak = other.weak; solid = other.solid; singleton = other.singleton; hasSolids = other.hasSolids; hasSingletons = other.hasSingletons;
		rawKmers = other.rawKmers; rawGoodKmers = other.rawGoodKmers; uniqueKmers = other.uniqueKmers; singletonKmers = other.singletonKmers; subtracted = other.subtracted;
		return *this;
	}
	inline long getRawKmers() const { return rawKmers; }                          /* :455-459 */
	inline long getRawGoodKmers() const { return rawGoodKmers; }
	inline long getUniqueKmers() const { return uniqueKmers; }
	inline long getSingletonKmers() const { return singletonKmers; }
	static long &getKmerSubsample() { return KmerSpectrumOptions::getOptions().getKmerSubsample(); }   /* :461 */
	class SizeTracker {                                   /* :812-900 (track() and reset() as there) */
	public:
		class SizeTrackerElement {
		public:
			long rawKmers, rawGoodKmers, uniqueKmers, singletonKmers;
			SizeTrackerElement(long raw = 0, long rawGood = 0, long unique = 0, long single = 0) : rawKmers(raw), rawGoodKmers(rawGood), uniqueKmers(unique), singletonKmers(single) {}
		};
		typedef std::vector<SizeTrackerElement> Elements;
		long nextToTrack;
		Elements elements;
		SizeTracker() { reset(); }
		void track(long raw, long rawGood, long unique, long single, bool force = false) {
			if (raw < nextToTrack && !force) return;
			elements.push_back(SizeTrackerElement(raw, rawGood, unique, single));
			if (raw >= nextToTrack) nextToTrack *= 1.05;
		}
		void reset() { nextToTrack = 128; elements.clear(); track(0, 0, 0, 0); }
	};
	SizeTracker sizeTracker;                              /* :901 */
	SizeTracker getSizeTracker() const { return sizeTracker; }                    /* :902-907 */
	void setSizeTracker(SizeTracker &st) { sizeTracker = st; }
	void trackSpectrum(bool force = false) { sizeTracker.track(rawKmers, rawGoodKmers, uniqueKmers, singletonKmers, force); }   /* :1574-1576 */
	static long estimateRawKmers(const ReadSet &store) {                          /* :573-584 */
		long n = 0; for (ReadSet::ReadSetSizeType i = 0; i < store.getSize(); i++) { const long L = (long)store.getRead(i).getFasta().size(); if (L >= (long)KmerSizer::getSequenceLength()) n += L - KmerSizer::getSequenceLength() + 1; }
		return n;
	}
	/* :1818-1830, numParts <= 1 */
	void buildKmerSpectrumInParts(const ReadSet &store, unsigned long numParts, std::string mmapFileNamePrefix = "") { (void)numParts; (void)mmapFileNamePrefix; buildKmerSpectrum(store, false); }
	virtual void buildKmerSpectrum(const ReadSet &store) { this->buildKmerSpectrum(store, hasSolids); }   /* :2081 */
	virtual void buildKmerSpectrum(const ReadSet &store, bool isSolid) { (void)store; (void)isSolid; }    /* :2084: the CPU build, not mocked */
	void optimize(bool singletonsToo = false) { (void)singletonsToo; }            /* :463 */
};

#endif
