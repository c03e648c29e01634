/* Drives include/kmernator_amd_shim.hpp the way apps/FilterReads.cpp:126-140 drives KmerSpectrum, against the MOCK of the
 * reference's headers (tests/cpp/mock_kmernator.h -- it pins nothing about the reference).  Prints "<kmer bytes hex> <count>"
 * for every weak entry so the test can compare with the oracle. */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "mock_kmernator.h"
uint8_t Read::FASTQ_START_CHAR = 33;
#include "kmernator_amd_shim.hpp"

typedef MockKmerMap<12> DataMap;
typedef MockKmerMap<1> SingletonMap;
typedef KmerSpectrum<DataMap, DataMap, SingletonMap> KS;      /* apps/FilterReads.h:246 */
typedef GpuKmerSpectrum<KS> GKS;

int main(int argc, char **argv) {
	if (argc < 4) { fprintf(stderr, "usage: shim_demo <k> <fastq> <start char>\n"); return 2; }
	KmerSizer::set((uint32_t)atoi(argv[1]));
	Read::FASTQ_START_CHAR = (uint8_t)atoi(argv[3]);
	ReadSet reads;
	std::ifstream in(argv[2]);
	std::string name, seq, plus, qual;
	while (std::getline(in, name) && std::getline(in, seq) && std::getline(in, plus) && std::getline(in, qual)) reads.append(Read(name.substr(1), seq, qual));
	try {
		GKS spectrum(0);                                            /* apps/FilterReads.cpp:126 */
		long rawKmers = KS::estimateRawKmers(reads);                /* :133 */
		spectrum = GKS(rawKmers);                                   /* :136 */
		spectrum.setSizeTracking(true);                             /* --size-history-file given (:141-147): the history is kept by the device build */
		{ GKS copy(spectrum); GKS other(7); other = copy; }         /* copies share the (not yet made) handle and die quietly */
		spectrum.buildKmerSpectrumInParts(reads, 0, "");           /* :139 -> virtual buildKmerSpectrum(store, false) */
		spectrum.optimize();                                        /* :140 */
		spectrum.trackSpectrum(true);                               /* :141; the history itself came with the build (kmr_size_tracker) */
		{
			const KS::SizeTracker hist = spectrum.getSizeTracker();
			for (size_t i = 0; i < hist.elements.size(); i++)
				fprintf(stderr, "history\t%ld\t%ld\t%ld\t%ld\n", hist.elements[i].rawKmers, hist.elements[i].rawGoodKmers, hist.elements[i].uniqueKmers, hist.elements[i].singletonKmers);
		}
		GKS again(spectrum);                                        /* the handle now exists and is shared */
		fprintf(stderr, "raw %ld good %ld unique %ld singleton %ld weak %zu\n", spectrum.getRawKmers(), spectrum.getRawGoodKmers(), spectrum.getUniqueKmers(),
		        spectrum.getSingletonKmers(), spectrum.weak.size());
		for (std::map<std::string, std::string>::const_iterator it = again.weak.entries().begin(); it != again.weak.entries().end(); ++it) {
			for (size_t i = 0; i < it->first.size(); i++) printf("%02x", (unsigned char)it->first[i]);
			uint16_t count; memcpy(&count, it->second.data(), 2);
			printf(" %u\n", count);
		}
	} catch (std::exception &e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
	return 0;
}
