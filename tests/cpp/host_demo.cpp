// C++ host side above the C-ABI, driven the way the reference's apps drive KmerSpectrum:
//   host_demo mercount <fastq> <out-prefix>      MeraculousCounter (apps/MeraculousCounter.cpp:110-151): k = 21,
//                                                --min-kmer-quality 0 --min-quality-score 2, dumpCounts + dumpGraphs
//   host_demo filter   <fastq> <out-file> [artifacts.fa]        FilterReads (apps/FilterReads.cpp:83-215) up to scoreAndTrimReads:
//                                                k = 31, one line "<name> [Trim:o+l ]MedianScore:s" per read
//   host_demo mercount-ranks <fastq> <out-prefix> <rank> <world> <id-file>
//                                                MeraculousCounter as one process per GPU (device = rank): the rank takes every
//                                                world-th block of 64 reads, rank 0 writes the exchange id to <id-file>, the
//                                                others wait for it; the owner exchange runs inside the library over RCCL;
//                                                every rank dumps <out-prefix>.mercount.<rank> / .mergraph.<rank>
// Exit code 3 = no HIP device (the library has no CPU path).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include "kmernator_amd.hpp"

using namespace kmernator;

static std::string slurp(const char *path) { std::ifstream f(path, std::ios::binary); return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()); }

int main(int argc, char **argv) {
	if (argc < 4) { std::fprintf(stderr, "usage: host_demo mercount|filter|mercount-ranks <fastq> <out> [artifacts.fa | rank world id-file]\n"); return 2; }
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
		} else if (mode == "mercount-ranks") {
			if (argc < 7) return 2;
			const uint32_t rank = (uint32_t)std::atoi(argv[4]), world = (uint32_t)std::atoi(argv[5]);
			const std::string idFile = argv[6];
			kmr_config c = KmerSpectrum::defaults(21, 56000);
			c.value_kind = KMR_VALUE_EXT; c.min_weight = 0.0f; c.min_quality_score = 2;
			c.rank = rank; c.world_size = world; c.device = (int)rank;
			KmerSpectrum sp(c);
			std::vector<uint8_t> id;
			if (rank == 0) {
				id = KmerSpectrum::exchangeUniqueId();
				{ std::ofstream f(idFile + ".tmp", std::ios::binary); f.write((const char *)id.data(), (std::streamsize)id.size()); }
				std::rename((idFile + ".tmp").c_str(), idFile.c_str());
			} else {
				for (int tries = 0; tries < 600 && id.size() != KMR_EXCHANGE_ID_BYTES; tries++) {
					const std::string got = slurp(idFile.c_str());
					if (got.size() == KMR_EXCHANGE_ID_BYTES) id.assign(got.begin(), got.end()); else std::this_thread::sleep_for(std::chrono::milliseconds(100));
				}
			}
			sp.exchangeInit(id);
			/* this rank's share of the file: every world-th block of 64 records (a FASTQ record is four lines here).  The quality
			 * base is detected on each share, as every rank of the reference does on the file it reads */
			std::string mine; uint64_t line = 0, first = ~0ull; size_t at = 0;
			while (at < text.size()) {
				size_t e = text.find('\n', at); if (e == std::string::npos) e = text.size() - 1;
				const uint64_t rec = line / 4;
				if ((rec / 64) % world == rank) { if (first == ~0ull) first = rec; mine.append(text, at, e + 1 - at); }
				at = e + 1; line++;
			}
			ReadSet reads(sp, mine);
			sp.buildKmerSpectrumExchange(&reads, first == ~0ull ? 0 : first);
			sp.purgeMinDepth(2);
			const std::string sfx = "." + std::to_string(rank);
			std::remove((out + ".mercount" + sfx).c_str()); std::remove((out + ".mergraph" + sfx).c_str());
			sp.dumpCounts(out + ".mercount" + sfx, 2);
			sp.dumpGraphs(out + ".mergraph" + sfx, 2);
			uint64_t sent = 0; double ms = 0; kmr_exchange_stats(sp.raw(), &sent, &ms);
			std::printf("rank %u of %u: reads %llu raw %llu unique %llu bytes-to-peers %llu\n", rank, world, (unsigned long long)reads.getSize(),
			            (unsigned long long)sp.getRawKmers(), (unsigned long long)sp.getUniqueKmers(), (unsigned long long)sent);
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
