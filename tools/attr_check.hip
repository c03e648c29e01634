#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../kmernator_amd/csrc/kmr_partition.hpp"
using namespace kmr;
template <int W> void chk(int bits) {
	auto kern = partition_kernel<W, 1, 0>;
	hipFuncAttributes a; hipError_t e = hipFuncGetAttributes(&a, (const void *)kern);
	size_t want = partition_smem_bytes<W>(bits);
	hipError_t e2 = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
	printf("W=%d get=%s static=%zu maxdyn=%d maxthreads=%d regs=%d  want=%zu set=%s\n", W, hipGetErrorString(e), a.sharedSizeBytes, a.maxDynamicSharedSizeBytes, a.maxThreadsPerBlock, a.numRegs, want, hipGetErrorString(e2));
	for (size_t tryb : {65536ul, 98304ul, 131072ul, 147456ul, 155648ul, 160000ul}) { hipError_t e3 = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tryb); printf("   %zu -> %s\n", tryb, hipGetErrorString(e3)); }
}
int main() { chk<1>(10); chk<2>(9); chk<3>(9); chk<4>(8); return 0; }
