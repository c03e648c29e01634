/*
 * kmr_instances.hpp -- the heavy kernel templates are compiled in translation units of their own (kmr_inst_*.hip), in
 * parallel; kmr_api.hip only sees `extern template` declarations of them.  One list, two readings:
 *   KMR_INSTANCES_EXTERN defined   every line is an explicit instantiation DECLARATION (kmr_api.hip)
 *   otherwise                      the lines of the groups selected with KMR_INST_<GROUP> (and KMR_INST_W = key words)
 *                                  are explicit instantiation DEFINITIONS
 * A kernel launched from kmr_api.hip with template arguments that are not listed here fails at link time
 * (-Wl,-z,defs in the Makefile), not at load time.
 */
#ifndef KMR_INSTANCES_HPP_
#define KMR_INSTANCES_HPP_

#include "kmr_superkmer.hpp"

namespace kmr {

static const int COUNT_LOG2S = 10;
static const int COUNT_LOG2S_EXT = 9;                        /* build_mode 3 with extension values: 60 bytes per slot; 512 slots let three blocks share a CU (lists half as long) */                            /* 1024-slot LDS table per final list (expected ~350 distinct keys) */
/* partition kernel shape: one 1024-thread block per compute unit, 8 records per thread per batch, a 4-record write-combining
 * line per list in LDS (see partition_direct_kernel) */
static const int PD_THREADS = 1024, PD_RPT = 8, PD_LINE = 4;

#ifdef KMR_INSTANCES_EXTERN
#define KMR_T extern template
#else
#define KMR_T template
#endif

/* build_mode 3: extraction into super-k-mer lists */
#define KMR_SKX(W, WIN, FILT) KMR_T __global__ void sk_extract_kernel<W, WIN, FILT, false>(ReadsView, DevParams, SkParams, PoolView); KMR_T __global__ void sk_extract_kernel<W, WIN, FILT, true>(ReadsView, DevParams, SkParams, PoolView);
#define KMR_SKXL(W, WIN) KMR_T __global__ void sk_extract_lean_kernel<W, WIN, false>(ReadsView, DevParams, SkParams, PoolView, float, SkPacked); KMR_T __global__ void sk_extract_lean_kernel<W, WIN, true>(ReadsView, DevParams, SkParams, PoolView, float, SkPacked);
#define KMR_SKX_W(W) KMR_SKX(W, 16, false) KMR_SKX(W, 16, true) KMR_SKX(W, 8, false) KMR_SKX(W, 8, true) KMR_SKX(W, 4, false) KMR_SKX(W, 4, true) KMR_SKXL(W, 16) KMR_SKXL(W, 8) KMR_SKXL(W, 4)

#define KMR_SKX_W32(W) KMR_SKX(W, 32, false) KMR_SKX(W, 32, true) KMR_SKXL(W, 32)      /* k >= 45: keys of two words and more */

/* build_mode 3: count pass and streaming lookups */
#define KMR_SKC(W, TRACK) KMR_SKCX(W, TRACK, false)
#define KMR_SKCX(W, TRACK, EXT) KMR_T __global__ void sk_count_kernel<W, EXT ? COUNT_LOG2S_EXT : COUNT_LOG2S, TRACK, EXT>(PoolView, const uint64_t *, const uint64_t *, uint64_t, uint32_t, CountOut, FinalizeParams, unsigned int *, uint32_t, SkTrackView, SkLong<W>);
#define KMR_SKL(W) KMR_T __global__ void sk_lookup_kernel<W>(PoolView, const uint64_t *, const uint64_t *, uint64_t, uint32_t, const uint64_t *, const uint64_t *, const uint32_t *, uint32_t *, uint64_t, unsigned int *, const uint64_t *, const uint64_t *, const uint32_t *, uint64_t, uint64_t);
#define KMR_SKCU(W) KMR_T __global__ void sk_count_kernel<W, COUNT_LOG2S, false, false, true>(PoolView, const uint64_t *, const uint64_t *, uint64_t, uint32_t, CountOut, FinalizeParams, unsigned int *, uint32_t, SkTrackView, SkLong<W>);
#define KMR_SKC_W(W) KMR_SKC(W, false) KMR_SKCU(W) KMR_SKC(W, true) KMR_SKCX(W, false, true) KMR_SKL(W)

/* build modes 1 and 2, lookups, owner requests: the k-mer extraction with its Ops */
#define KMR_EX(W, EXT, OP, SUB) KMR_T __global__ void extract_kernel<W, EXT, OP, SUB>(ReadsView, DevParams, OP);
#define KMR_EX_OP(W, EXT, ...) KMR_T __global__ void extract_kernel<W, EXT, __VA_ARGS__, false>(ReadsView, DevParams, __VA_ARGS__); KMR_T __global__ void extract_kernel<W, EXT, __VA_ARGS__, true>(ReadsView, DevParams, __VA_ARGS__);
#define KMR_EX_W(W) \
	KMR_EX_OP(W, false, InsertOp<W, false>) KMR_EX_OP(W, true, InsertOp<W, true>) \
	KMR_EX_OP(W, false, LinearOp<W, false, false>) KMR_EX_OP(W, false, LinearOp<W, false, true>) \
	KMR_EX_OP(W, true, LinearOp<W, true, false>) KMR_EX_OP(W, true, LinearOp<W, true, true>) \
	KMR_T __global__ void extract_kernel<W, false, LookupOp<W>, false>(ReadsView, DevParams, LookupOp<W>);

/* build_mode 2: partition passes, owner scatter, count pass over k-mer records */
#define KMR_PD(W, EXT, LEVEL) KMR_T __global__ void partition_direct_kernel<W, EXT, LEVEL, PD_THREADS, PD_RPT, PD_LINE>(PartSource<W>, PoolView, unsigned int *, const int, const int);
#define KMR_OS(W, EXT, REQ) KMR_T __global__ void owner_scatter_kernel<W, EXT, REQ>(const typename PoolRec<W, EXT>::type *, const uint64_t *, const uint32_t *, uint64_t, uint32_t, uint32_t, uint32_t *, uint64_t, unsigned long long *, unsigned int *, uint32_t *, uint32_t *, OwnerFn);
#define KMR_CK(W, EXT, LOG2S, NARROW) KMR_T __global__ void count_kernel<W, EXT, LOG2S, NARROW>(PoolView, const uint64_t *, const uint64_t *, uint64_t, CountOut, FinalizeParams, unsigned int *, int);
#define KMR_PART_W(W) \
	KMR_PD(W, false, 1) KMR_PD(W, false, 2) KMR_PD(W, true, 1) KMR_PD(W, true, 2) \
	KMR_OS(W, false, false) KMR_OS(W, false, true) KMR_OS(W, true, false) KMR_OS(W, true, true) \
	KMR_CK(W, false, COUNT_LOG2S, false) KMR_CK(W, false, 11, false) KMR_CK(W, true, COUNT_LOG2S, false)

#if defined(KMR_INSTANCES_EXTERN)
KMR_SKX_W(1) KMR_SKX_W(2) KMR_SKX_W(3) KMR_SKX_W(4)
KMR_SKX_W32(2) KMR_SKX_W32(3) KMR_SKX_W32(4)
KMR_SKC_W(1) KMR_SKC_W(2) KMR_SKC_W(3) KMR_SKC_W(4)
KMR_EX_W(1) KMR_EX_W(2) KMR_EX_W(3) KMR_EX_W(4)
KMR_PART_W(1) KMR_PART_W(2) KMR_PART_W(3) KMR_PART_W(4)
KMR_CK(1, true, COUNT_LOG2S, true)
#else
#ifdef KMR_INST_SKX
KMR_SKX_W(KMR_INST_W)
#if KMR_INST_W > 1
KMR_SKX_W32(KMR_INST_W)
#endif
#endif
#ifdef KMR_INST_SKC
KMR_SKC_W(KMR_INST_W)
#endif
#ifdef KMR_INST_EX
KMR_EX_W(KMR_INST_W)
#endif
#ifdef KMR_INST_PART
KMR_PART_W(KMR_INST_W)
#if KMR_INST_W == 1
KMR_CK(1, true, COUNT_LOG2S, true)
#endif
#endif
#endif

}  // namespace kmr
#endif
