// C++ host side above the C-ABI, driven the way the reference's apps drive KmerSpectrum:
//   host_demo mercount <fastq> <out-prefix>      MeraculousCounter (apps/MeraculousCounter.cpp:110-151): k = 21,
//                                                --min-kmer-quality 0 --min-quality-score 2, dumpCounts + dumpGraphs
//   host_demo filter   <fastq> <out-file> [artifacts.fa]        FilterReads (apps/FilterReads.cpp:83-215) up to scoreAndTrimReads:
//                                                k = 31, one line "<name> [Trim:o+l ]MedianScore:s" per read
// Exit code 3 = no HIP device (the library has no CPU path).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include "kmernator_amd.hpp"

using namespace kmernator;

static std::string slurp(const char *path) { std::ifstream f(path, std::ios::binary); return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()); }

int main(int argc, char **argv) {
	if (argc < 4) { std::fprintf(stderr, "usage: host_demo mercount|filter <fastq> <out> [artifacts.fa]\n"); return 2; }
	const std::string mode = argv[1], text = slurp(argv[2]), out = argv[3];
	try {
		if (mode == "mercount") {
			kmr_config c = KmerSpectrum::defaults(21, 56000);
			c.value_kind = KMR_VALUE_EXT; c.min_weight = 0.0f; c.min_quality_score = 2;
			KmerSpectrum sp(c);
			ReadSet reads(sp, text);                       // Phred-64 detected as validateFastqStart does
			sp.buildKmerSpectrum(reads);
			sp.purgeMinDepth(2);
			std::remove((out + ".mercount").c_str()); std::remove((out + ".mergraph").c_str());
			sp.dumpCounts(out + ".mercount", 2);
			sp.dumpGraphs(out + ".mergraph", 2);
			std::printf("reads %llu quality-base %u raw %llu unique %llu\n", (unsigned long long)reads.getSize(), reads.getInputQualityBase(),
			            (unsigned long long)sp.getRawKmers(), (unsigned long long)sp.getUniqueKmers());
		} else if (mode == "filter") {
			KmerSpectrum sp(KmerSpectrum::defaults(31, 46000));
			ReadSet input(sp, text);
			/* with a 4th argument (FASTA of artifact sequences) the artifact filter runs first, as in FilterReads.cpp:107-118
			 * with the settings of test/runFilterTests.sh (--artifact-edit-distance 1 --min-read-length 25) */
			std::unique_ptr<ReadSet> filtered;
			FilterKnownOddities::Results fr;
			if (argc > 4) {
				kmr_artifact_config ac = FilterKnownOddities::defaults(sp.config());
				ac.edit_distance = 1; ac.min_read_length = 25.0f;
				FilterKnownOddities filter(sp, slurp(argv[4]), ac);
				filtered = filter.applyFilter(input, fr);
			}
			const ReadSet &reads = filtered ? *filtered : input;
			sp.buildKmerSpectrum(reads);
			sp.purgeMinDepth(2);
			KmerSpectrum::TrimResult r = sp.scoreAndTrimReads(reads, 2, KmerSpectrum::KS_MEDIAN);
			std::ofstream o(out);
			for (uint64_t i = 0; i < reads.getSize(); i++) {
				std::string name = reads.getName(i);
				name = name.substr(0, name.find_first_of(" \t"));
				o << name << " ";
				if (filtered && i < input.getSize() && fr.action[i] == 1) o << "AFTrim:" << fr.minPass[i] << "+" << (fr.maxPass[i] - fr.minPass[i]) << " ";
				if (r.wasTrimmed[i]) o << "Trim:" << r.trimOffset[i] << "+" << r.trimLength[i] << " ";
				o << "MedianScore:" << (long)(r.score[i] + 0.5) << "\n";
			}
			std::printf("%s", sp.getHistogram().toString().substr(0, 31).c_str());
		} else return 2;
	} catch (const KmerSpectrumError &e) {
		std::fprintf(stderr, "%s\n", e.what());
		return e.code == KMR_ERR_NO_DEVICE ? 3 : 1;
	}
	return 0;
}
