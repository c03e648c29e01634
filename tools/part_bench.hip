// micro-benchmark of partition_direct_kernel shapes (development tool, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../kmernator_amd/csrc/kmr_partition.hpp"
using namespace kmr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(Record<1> *r, uint64_t n) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { r[i].key[0] = mix64(i * 7 + 1) >> 2 << 2; r[i].w = 0.99f; r[i].pkt = (uint32_t)i; }
}
template <int THREADS, int RPT, int G> float run_direct(const Record<1> *lin, uint64_t n, int bits, PoolView pv, unsigned int *wc, int grid) {
	(void)hipMemset(pv.head, 0, 4); (void)hipMemset(wc, 0, 4);
	PartSource<1> S; memset(&S, 0, sizeof(S)); S.kb = 8; S.rot = 21; S.linear = lin; S.n_ext = (n + 8191) / 8192; S.ext_len = 8192; S.total = n;
	auto kern = partition_direct_kernel<1, 1, THREADS, RPT, G>;
	(void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	(void)hipEventRecord(a);
	const size_t smem = partition_direct_smem_bytes<1, THREADS, RPT, G>(bits);
	hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), smem, 0, S, pv, wc, bits, 0);
	(void)hipEventRecord(b); (void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b);
	unsigned int head = 0; (void)hipMemcpy(&head, pv.head, 4, hipMemcpyDeviceToHost);
	printf("  <%d thr, %d rec/thr, line %d> bits %d grid %d: %.2f ms  chunks %u err=%s\n", THREADS, RPT, G, bits, grid, ms, head, hipGetErrorString(hipGetLastError()));
	return ms;
}
__global__ void fill_genome(Record<1> *r, uint64_t n, const uint8_t *g, uint64_t glen) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t pos = i % (glen - 31);
		uint64_t f = 0, rc = 0;
		for (int j = 0; j < 31; j++) { uint64_t b = g[pos + j] & 3; f = (f << 2) | b; rc = (rc >> 2) | ((3 - b) << 60); }
		uint64_t c = f < rc ? f : rc;
		r[i].key[0] = c << 2; r[i].w = 0.99f; r[i].pkt = (uint32_t)i;
	}
}
int main(int argc, char **argv) {
	const uint64_t n = 200000000ull;
	Record<1> *lin; CK(hipMalloc(&lin, n * 16));
	if (argc > 1) {
		const uint64_t glen = 8000000; std::vector<uint8_t> g(glen); uint64_t x = 88172645463325252ull;
		for (auto &b : g) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; b = (uint8_t)(x >> 33) & 3; }
		uint8_t *dg; CK(hipMalloc(&dg, glen)); CK(hipMemcpy(dg, g.data(), glen, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(fill_genome, dim3(4096), dim3(256), 0, 0, lin, n, dg, glen);
		printf("genome-derived canonical 31-mers\n");
	} else
	hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, lin, n);
	PoolView pv; uint32_t cap = (uint32_t)(n / CH + 2000000);
	CK(hipMalloc(&pv.base, (size_t)cap * CH * 16)); CK(hipMalloc(&pv.chunk_list, 4ull * cap)); CK(hipMalloc(&pv.chunk_count, 4ull * cap)); CK(hipMalloc(&pv.head, 4)); CK(hipMalloc(&pv.err, 4));
	pv.cap = cap; (void)hipMemset(pv.err, 0, 4);
	unsigned int *wc; CK(hipMalloc(&wc, 4));
	CK(hipDeviceSynchronize());
	for (int rep = 0; rep < 2; rep++) {
		run_direct<1024, 8, 4>(lin, n, 10, pv, wc, 256);
		run_direct<1024, 8, 4>(lin, n, 9, pv, wc, 256);
		run_direct<1024, 8, 8>(lin, n, 9, pv, wc, 256);
		run_direct<1024, 16, 8>(lin, n, 9, pv, wc, 256);
		run_direct<1024, 8, 8>(lin, n, 8, pv, wc, 256);
		run_direct<1024, 8, 16>(lin, n, 8, pv, wc, 256);
		run_direct<1024, 8, 4>(lin, n, 7, pv, wc, 256);
		run_direct<1024, 8, 16>(lin, n, 7, pv, wc, 256);
	}
	return 0;
}
