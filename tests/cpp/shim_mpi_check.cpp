/* Instantiates the MPI flavour of the shim (GpuDistributedKmerSpectrum) against the MOCK of the reference's headers so that the
 * compiler sees every line of it (syntax / access check only: nothing here runs, and nothing about the reference is pinned). */
#define KMERNATOR_AMD_SHIM_MPI
#include <mpi.h>

#include "mock_kmernator.h"
uint8_t Read::FASTQ_START_CHAR = 33;

/* boost::mpi::communicator converts to MPI_Comm (boost/mpi/communicator.hpp:  operator MPI_Comm() const); the reference says
 * `namespace mpi = boost::mpi` (src/MPIBase.h).  DistributedKmerSpectrum(mpi::communicator &, unsigned long estimatedRawKmers = 0,
 * bool separateSingletons = true), src/DistributedFunctions.h:124-131 */
namespace mpi { class communicator { public: MPI_Comm c; operator MPI_Comm() const { return c; } int rank() const { return 0; } int size() const { return 1; } }; }
#include "kmernator_amd_shim.hpp"

typedef MockKmerMap<60> ExtMap;
typedef MockKmerMap<5> ExtSingletonMap;
typedef KmerSpectrum<ExtMap, ExtMap, ExtSingletonMap> KS;
class MockDistributedKmerSpectrum : public KS {
public:
	MockDistributedKmerSpectrum(mpi::communicator &_world, unsigned long estimatedRawKmers = 0, bool separateSingletons = true) : KS(estimatedRawKmers, separateSingletons), world(_world) {}
protected:
	mpi::communicator world;
};

int check(mpi::communicator &world, const ReadSet &reads) {
	GpuDistributedKmerSpectrum<MockDistributedKmerSpectrum> spectrum(world, 1000, true, KMR_VALUE_EXT);
	spectrum.buildKmerSpectrum(reads);
	GpuDistributedKmerSpectrum<MockDistributedKmerSpectrum> copy(spectrum);
	copy = spectrum;
	return (int)copy.weak.size();
}
