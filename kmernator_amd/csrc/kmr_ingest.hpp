/*
 * kmr_ingest.hpp -- FASTQ text -> device-resident read batch.
 *
 * Replaces, for a whole in-memory FASTQ block, FastqStreamParser::readRecord
 * (src/ReadFileReader.h:768-835), ReadFileReader::nextRead (:296-329: Casava-1.8 failed-filter
 * reads are skipped, bases upper-cased, #bases == #quals), SequenceRecordParser::trimName /
 * isCommentCasava18 (src/Utils.h:561-598,678-685) and the quality handling of
 * ReadSet::appendFasta/addRead/validateFastqStart/__setFastqStart (src/ReadSet.cpp:136-141,311-345,
 * src/ReadSet.h:171-209, src/Sequence.h:456-479).
 *
 * The reference walks the text line by line; here every byte is looked at once, in parallel:
 *   ingest_count_lines   per 4 KB block: number of non-empty line starts
 *   (exclusive scan)
 *   ingest_index_lines   line_start[i] for the i-th non-empty line
 *   ingest_line_lengths  line_len[i], from the start of the next line
 *   ingest_records       record r = lines 4r..4r+3: markers, lengths, Casava filter -> keep[r], len[r]
 *   (two exclusive scans: kept index, base offset)
 *   ingest_copy          one wavefront per record: upper-cased bases, rescaled quals, offsets, name
 *                        spans, and the quality-base check of the first 19 999 kept reads
 *   ingest_shift_quals   the one-time 33 <-> 64 flip of __setFastqStart, applied to the whole batch
 *
 * Strictness: the reference silently skips up to 100 000 lines that do not start with '@' where a
 * record should begin (readName); this parser accepts blank lines between records only and reports
 * anything else as malformed.  Every input the parser accepts is parsed exactly as the reference does.
 */
#ifndef KMR_INGEST_HPP_
#define KMR_INGEST_HPP_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmr {

enum { ING_ERR_NAME = 1, ING_ERR_PLUS = 2, ING_ERR_LEN = 4, ING_ERR_BLANK = 8, ING_ERR_TRUNC = 16, ING_ERR_BASE = 32 };
static const int ING_THREADS = 256, ING_BYTES = 16;      /* bytes per thread: one block looks at 4 KB */
static const uint32_t ING_VALIDATE_READS = 20000;        /* validateFastqStart: getSize() < 20000, src/ReadSet.h:172 */

__device__ __forceinline__ bool ing_is_start(const uint8_t *text, uint64_t p) { return (p == 0 || text[p - 1] == '\n') && text[p] != '\n'; }

/* The ING_BYTES bytes of a thread and the byte in front of them: one 16-byte load when the text is 16-byte aligned and the
 * chunk lies inside it, single bytes otherwise.  Bytes past the end read as '\n'. */
struct IngChunk { uint8_t b[ING_BYTES]; uint8_t prev; };
__device__ __forceinline__ IngChunk ing_load_chunk(const uint8_t *text, uint64_t len, uint64_t p0, bool aligned) {
	IngChunk c;
	if (aligned && p0 + ING_BYTES <= len) {
		const uint4 v = *(const uint4 *)(text + p0);
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int j = 0; j < ING_BYTES; j++) c.b[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
	} else {
#pragma unroll
		for (int j = 0; j < ING_BYTES; j++) c.b[j] = p0 + j < len ? text[p0 + j] : (uint8_t)'\n';
	}
	c.prev = (p0 == 0 || p0 > len) ? (uint8_t)'\n' : text[p0 - 1];
	return c;
}
__device__ __forceinline__ uint32_t ing_start_mask(const IngChunk &c, uint64_t p0, uint64_t len) {
	uint32_t m = 0;
#pragma unroll
	for (int j = 0; j < ING_BYTES; j++) {
		const uint8_t pv = j ? c.b[j - 1] : c.prev;
		if (p0 + j < len && pv == '\n' && c.b[j] != '\n') m |= 1u << j;
	}
	return m;
}
/* first '\n' at or after p (or len): eight bytes per load once p is 8-byte aligned */
__device__ __forceinline__ uint64_t ing_line_end(const uint8_t *text, uint64_t len, uint64_t p, bool aligned) {
	uint64_t e = p;
	if (aligned) {
		while (e < len && (e & 7)) { if (text[e] == '\n') return e; e++; }
		while (e + 8 <= len) {
			const uint64_t x = *(const uint64_t *)(text + e) ^ 0x0a0a0a0a0a0a0a0aull;
			const uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
			if (z) return e + (__builtin_ctzll(z) >> 3);
			e += 8;
		}
	}
	while (e < len && text[e] != '\n') e++;
	return e;
}

/* exclusive prefix of v over the block (256 threads); total in *sum */
__device__ __forceinline__ uint32_t ing_block_scan(uint32_t v, uint32_t *sum) {
	__shared__ uint32_t wsum[ING_THREADS / 64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
	if (lane == 63) wsum[wave] = inc;
	__syncthreads();
	uint32_t base = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < ING_THREADS / 64; w++) { if (w < wave) base += wsum[w]; tot += wsum[w]; }
	*sum = tot;
	__syncthreads();
	return base + inc - v;
}

__global__ __launch_bounds__(ING_THREADS)
void ingest_count_lines(const uint8_t *text, uint64_t len, uint32_t *block_lines) {
	const uint64_t p0 = ((uint64_t)blockIdx.x * ING_THREADS + threadIdx.x) * ING_BYTES;
	const bool aligned = ((uintptr_t)text & 15) == 0;
	const uint32_t c = p0 < len ? (uint32_t)__builtin_popcount(ing_start_mask(ing_load_chunk(text, len, p0, aligned), p0, len)) : 0u;
	uint32_t tot;
	ing_block_scan(c, &tot);
	if (threadIdx.x == 0) block_lines[blockIdx.x] = tot;
}

__global__ __launch_bounds__(ING_THREADS)
void ingest_index_lines(const uint8_t *text, uint64_t len, const uint64_t *block_base, uint64_t *line_start) {
	const uint64_t p0 = ((uint64_t)blockIdx.x * ING_THREADS + threadIdx.x) * ING_BYTES;
	const bool aligned = ((uintptr_t)text & 15) == 0;
	const uint32_t mask = p0 < len ? ing_start_mask(ing_load_chunk(text, len, p0, aligned), p0, len) : 0u;
	const uint32_t c = (uint32_t)__builtin_popcount(mask);
	uint32_t tot;
	uint64_t idx = block_base[blockIdx.x] + ing_block_scan(c, &tot);
	if (!c) return;
	for (int j = 0; j < ING_BYTES; j++) {
		if (mask & (1u << j)) line_start[idx++] = p0 + j;
	}
}
/* one thread per line: the line ends at the first '\n' at or after its start, and everything between that and the next
 * non-empty line's start is '\n' -- so the end is found from the next start (one byte looked at when no blank lines
 * intervene) instead of by reading the line; only the last line is read to its end */
__global__ void ingest_line_lengths(const uint8_t *text, uint64_t len, const uint64_t *line_start, uint64_t n_lines, uint32_t *line_len, uint32_t *err) {
	const bool aligned = ((uintptr_t)text & 15) == 0;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_lines; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t p = line_start[i];
		uint64_t e;
		if (i + 1 < n_lines) { e = line_start[i + 1] - 1; while (e > p && text[e - 1] == '\n') e--; }
		else e = ing_line_end(text, len, p, aligned);
		if (e - p > 0xffffffffull) { atomicOr(err, (uint32_t)ING_ERR_LEN); e = p; }
		line_len[i] = (uint32_t)(e - p);
	}
}

/* one thread per record */
__global__ void ingest_records(const uint8_t *text, const uint64_t *line_start, const uint32_t *line_len, uint64_t n_records, int store_comment,
                               uint32_t *keep, uint32_t *kept_len, uint32_t *err) {
	for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_records; r += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t s0 = line_start[4 * r], s1 = line_start[4 * r + 1], s2 = line_start[4 * r + 2], s3 = line_start[4 * r + 3];
		const uint32_t l0 = line_len[4 * r], l1 = line_len[4 * r + 1], l2 = line_len[4 * r + 2], l3 = line_len[4 * r + 3];
		uint32_t e = 0;
		if (text[s0] != '@' || l0 < 2) e |= ING_ERR_NAME;
		if (s1 != s0 + l0 + 1 || s2 != s1 + l1 + 1 || s3 != s2 + l2 + 1) e |= ING_ERR_BLANK;   /* an empty line inside a record */
		if (text[s2] != '+') e |= ING_ERR_PLUS;
		if (l3 != l1) e |= ING_ERR_LEN;
		if (e) { atomicOr(err, e); keep[r] = 0; kept_len[r] = 0; continue; }
		/* trimName on the name line without its marker (src/Utils.h:561-598) */
		const uint8_t *nm = text + s0 + 1; const uint32_t nl = l0 - 1;
		uint32_t ws = nl;
		for (uint32_t i = 0; i < nl; i++) { const uint8_t c = nm[i]; if (c == ' ' || c == '\t' || c == '\r' || c == '\n') { ws = i; break; } }
		if (ws == 0) { atomicOr(err, (uint32_t)ING_ERR_NAME); keep[r] = 0; kept_len[r] = 0; continue; }   /* an empty name ends the reference's stream */
		bool good = true;
		if (ws < nl && nl >= ws + 2) {
			const uint8_t *c = nm + ws + 1; const uint32_t cl = nl - ws - 1;
			const bool casava = cl >= 6 && c[1] == ':' && c[3] == ':' && c[5] == ':' && (c[0] == '1' || c[0] == '2') && (c[2] == 'Y' || c[2] == 'N');
			if (casava && (ws <= 2 || nm[ws - 2] != '/')) {
				const uint32_t p2 = store_comment ? ws : ws + 2;     /* the reference moves pos when it rewrites "name 1:Y" to "name/1" */
				if (nm[p2 + 3] == 'Y') good = false;
			}
		}
		keep[r] = good ? 1u : 0u;
		kept_len[r] = good ? l1 : 0u;
	}
}

/* four output bytes [a, a + 4) of a record's bases or quals (a 4-byte aligned in the output array): the source bytes sit at
 * text + sa .. sa + 3 at any alignment -- two aligned dword loads and a byte funnel shift when the text is 4-byte aligned and
 * the second dword is inside it, single bytes otherwise */
__device__ __forceinline__ uint32_t ing_src_dword(const uint8_t *text, uint64_t len, int64_t sa, bool aligned) {
	if (aligned && sa >= 0 && (uint64_t)sa + 8 <= len) {
		const uint32_t *w = (const uint32_t *)(text + ((uint64_t)sa & ~3ull));
		const uint64_t both = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
		return (uint32_t)(both >> (8 * ((uint64_t)sa & 3)));
	}
	uint32_t v = 0;
#pragma unroll
	for (int b = 0; b < 4; b++) { const int64_t q = sa + b; if (q >= 0 && (uint64_t)q < len) v |= (uint32_t)text[q] << (8 * b); }
	return v;
}

/* one wavefront per record, one lane per aligned output dword */
__global__ __launch_bounds__(256)
void ingest_copy(const uint8_t *text, uint64_t len, const uint64_t *line_start, const uint32_t *line_len, uint64_t n_records, const uint32_t *keep,
                 const uint64_t *kept_idx, const uint64_t *base_off, int qdelta, uint32_t start_char,
                 uint8_t *bases, uint8_t *quals, uint64_t *offsets, uint64_t *name_off, uint32_t *name_len, uint32_t *flip) {
	const int lane = threadIdx.x & 63;
	const bool aligned = ((uintptr_t)text & 3) == 0;
	const uint64_t wavesPerGrid = (uint64_t)gridDim.x * (blockDim.x >> 6);
	for (uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < n_records; r += wavesPerGrid) {
		if (!keep[r]) continue;
		const uint64_t j = kept_idx[r], o = base_off[r];
		const uint64_t sb = line_start[4 * r + 1], sq = line_start[4 * r + 3];
		const uint32_t L = line_len[4 * r + 1];
		uint32_t mn = 255;
		const uint64_t a0 = o & ~3ull, nd = (((o + L + 3) & ~3ull) - a0) >> 2;
		const int64_t shift = (int64_t)(o - a0);
		for (uint64_t d = lane; d < nd; d += 64) {
			const uint64_t a = a0 + 4 * d;                                   /* output bytes [a, a + 4) */
			const uint32_t vb = ing_src_dword(text, len, (int64_t)sb - shift + (int64_t)(4 * d), aligned);
			const uint32_t vq = ing_src_dword(text, len, (int64_t)sq - shift + (int64_t)(4 * d), aligned);
			uint32_t ob = 0, oq = 0, valid = 0;
#pragma unroll
			for (int b = 0; b < 4; b++) {
				const uint64_t pos = a + b;
				if (pos < o || pos >= o + L) continue;
				valid |= 1u << b;
				uint8_t c = (uint8_t)(vb >> (8 * b));
				if (c >= 'a' && c <= 'z') c -= 32;                           /* std::toupper, src/ReadFileReader.h:311 */
				const uint8_t q = (uint8_t)((uint8_t)(vq >> (8 * b)) + qdelta);  /* Read::rescaleQuality, src/Sequence.h:443-447 */
				ob |= (uint32_t)c << (8 * b); oq |= (uint32_t)q << (8 * b);
				mn = q < mn ? q : mn;
			}
			if (valid == 0xfu) { *(uint32_t *)(bases + a) = ob; *(uint32_t *)(quals + a) = oq; }
			else {
#pragma unroll
				for (int b = 0; b < 4; b++) if (valid & (1u << b)) { bases[a + b] = (uint8_t)(ob >> (8 * b)); quals[a + b] = (uint8_t)(oq >> (8 * b)); }
			}
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) { const uint32_t x = __shfl_xor(mn, off, 64); mn = x < mn ? x : mn; }
		if (lane == 0) {
			offsets[j] = o;
			name_off[j] = line_start[4 * r] + 1; name_len[j] = line_len[4 * r] - 1;
			/* validateFastqStart (src/ReadSet.h:171-191, src/Sequence.h:456-479: min_element for both bounds) */
			if (j + 1 < ING_VALIDATE_READS && L > 0 && (uint8_t)(text[sq] + qdelta) != 127 && (mn < start_char || mn > start_char + 40)) atomicOr(flip, 1u);
		}
	}
}

/* ---- 2-bit pack of a read batch: TwoBitSequence::compressSequence (src/TwoBitSequence.cpp:242-269, compressBase :114-147) ----
 * four bases per byte, first base in bits 7-6, A/a = 0 C/c = 1 G/g = 2 T/t = 3; anything else packs as 0 and is recorded as a
 * markup (char, offset), '.' recorded as 'N'.  One read per thread, bases through aligned 8-byte loads. */
struct IngBytes {
	const uint8_t *p; uint64_t w;
	__device__ __forceinline__ void init(const uint8_t *q) { p = q; w = *(const uint64_t *)((uintptr_t)q & ~(uintptr_t)7); }
	__device__ __forceinline__ uint8_t next() { const uint32_t o = (uint32_t)((uintptr_t)p & 7); if (o == 0) w = *(const uint64_t *)p; p++; return (uint8_t)(w >> (8 * o)); }
};
__device__ __forceinline__ uint32_t ing_base_code(uint8_t c) {      /* 4 = markup */
	switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
__global__ void twobit_count_kernel(const uint8_t *bases, const uint64_t *offsets, uint64_t n, uint32_t *packed_len, uint32_t *markups) {
	for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t o = offsets[r], L = offsets[r + 1] - o;
		IngBytes b; b.init(bases + o);
		uint32_t m = 0;
		for (uint64_t i = 0; i < L; i++) m += ing_base_code(b.next()) == 4 ? 1u : 0u;
		packed_len[r] = (uint32_t)((L + 3) / 4); markups[r] = m;
	}
}
__global__ void twobit_pack_kernel(const uint8_t *bases, const uint64_t *offsets, uint64_t n, const uint64_t *tb_off, const uint64_t *mk_off,
                                   uint8_t *twobit, uint32_t *mk_pos, uint8_t *mk_char) {
	for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t o = offsets[r], L = offsets[r + 1] - o;
		IngBytes b; b.init(bases + o);
		uint8_t *out = twobit + tb_off[r];
		uint64_t m = mk_off[r];
		uint32_t acc = 0;
		for (uint64_t i = 0; i < L; i++) {
			const uint8_t c = b.next();
			uint32_t code = ing_base_code(c);
			if (code == 4) { mk_pos[m] = (uint32_t)i; mk_char[m] = c == '.' ? (uint8_t)'N' : c; m++; code = 0; }
			acc |= code << (6 - 2 * (uint32_t)(i & 3));
			if ((i & 3) == 3) { *out++ = (uint8_t)acc; acc = 0; }
		}
		if (L & 3) *out = (uint8_t)acc;              /* the partial last byte is zero padded */
	}
}

/* ---- the way back: reads as the reference's Read keeps them (2-bit packed, every read on a byte of its own, + markups) -> the ASCII
 * batch the extraction kernels take.  Replaces TwoBitSequence::uncompressSequence + applyMarkup (src/TwoBitSequence.cpp:286-340)
 * over a whole batch. */
__global__ void twobit_bytes_kernel(const uint64_t *offsets, uint64_t n, uint32_t *packed_len) {
	for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) packed_len[r] = (uint32_t)((offsets[r + 1] - offsets[r] + 3) / 4);
}
/* a wavefront takes reads one after the other, a lane per ALIGNED output dword (four bases from two packed bytes); the bytes before the first and behind
 * the last aligned dword of a read share their dword with its neighbours and are stored one by one.  rel[] = the reads' base
 * offsets counted from the first read of the call (what the unpacked batch is indexed by). */
__device__ __forceinline__ uint32_t twobit_char(uint32_t code) { return (0x54474341u >> (8 * code)) & 0xffu; }      /* "ACGT" */
__global__ __launch_bounds__(256)
void twobit_unpack_kernel(const uint8_t *__restrict__ twobit, const uint64_t *__restrict__ tb_off, const uint64_t *__restrict__ offsets, uint64_t n, uint8_t *__restrict__ out, uint64_t *__restrict__ rel) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
	const uint64_t o0 = offsets[0];
	if (wave == 0 && lane == 0) rel[n] = offsets[n] - o0;
	/* 64 reads per wavefront and round: their offsets arrive with one coalesced load each and are handed round by readlane, and the
	 * packed bytes of four reads are asked for before the first of them is written out */
	auto bcast = [&](uint64_t v, uint32_t q) -> uint64_t { return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), (int)q) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)q); };
	for (uint64_t r0 = wave * 64; r0 < n; r0 += nwaves * 64) {
		const uint64_t rr = r0 + lane < n ? r0 + lane : n - 1;
		const uint64_t myA = offsets[rr] - o0, myL = offsets[rr + 1] - offsets[rr], myT = tb_off[rr];
		if (r0 + lane < n) rel[r0 + lane] = myA;
		const uint32_t cnt = (uint32_t)(n - r0 < 64 ? n - r0 : 64);
		for (uint32_t q0 = 0; q0 < cnt; q0 += 4) {
			uint64_t a[4], L[4], h[4], nd[4]; const uint8_t *src[4]; uint32_t v[4], hb[4], tb[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const uint32_t q = q0 + u < cnt ? q0 + u : cnt - 1;
				a[u] = bcast(myA, q); L[u] = q0 + u < cnt ? bcast(myL, q) : 0; src[u] = twobit + bcast(myT, q);
				const uint64_t head = (4 - (a[u] & 3)) & 3;
				h[u] = head < L[u] ? head : L[u]; nd[u] = (L[u] - h[u]) / 4;
				v[u] = 0; hb[u] = 0; tb[u] = 0;
				/* the first 64 aligned dwords of the read (a read of up to 259 bases: all of it), its head and tail bytes */
				if (lane < nd[u]) { const uint64_t j = h[u] + 4 * lane; v[u] = (uint32_t)src[u][j >> 2] << 8; if (j & 3) v[u] |= src[u][(j >> 2) + 1]; }
				if (lane < h[u]) hb[u] = src[u][0];
				const uint64_t t0 = h[u] + 4 * nd[u];
				if (t0 + lane < L[u]) tb[u] = src[u][(t0 + lane) >> 2];
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				if (lane < nd[u]) {
					const uint64_t j = h[u] + 4 * lane; const uint32_t s2 = 2 * (uint32_t)(j & 3);
					const uint32_t four = (v[u] >> (8 - s2)) & 0xffu;           /* first base in bits 7-6 */
					*(uint32_t *)(out + a[u] + j) = twobit_char(four >> 6) | (twobit_char((four >> 4) & 3u) << 8) | (twobit_char((four >> 2) & 3u) << 16) | (twobit_char(four & 3u) << 24);
				}
				if (lane < h[u]) out[a[u] + lane] = (uint8_t)twobit_char((hb[u] >> (6 - 2 * lane)) & 3u);
				const uint64_t t0 = h[u] + 4 * nd[u];
				if (t0 + lane < L[u]) out[a[u] + t0 + lane] = (uint8_t)twobit_char((tb[u] >> (6 - 2 * (uint32_t)((t0 + lane) & 3))) & 3u);
				for (uint64_t d = 64 + lane; d < nd[u]; d += 64) {      /* longer reads: the rest, dword by dword */
					const uint64_t j = h[u] + 4 * d; const uint32_t s2 = 2 * (uint32_t)(j & 3);
					uint32_t w = (uint32_t)src[u][j >> 2] << 8; if (s2) w |= src[u][(j >> 2) + 1];
					const uint32_t four = (w >> (8 - s2)) & 0xffu;
					*(uint32_t *)(out + a[u] + j) = twobit_char(four >> 6) | (twobit_char((four >> 4) & 3u) << 8) | (twobit_char((four >> 2) & 3u) << 16) | (twobit_char(four & 3u) << 24);
				}
			}
		}
	}
}
__global__ void twobit_markup_kernel(const uint64_t *mk_off, const uint32_t *mk_pos, const uint8_t *mk_char, const uint64_t *rel, uint64_t n, uint8_t *out) {
	for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t L = rel[r + 1] - rel[r];
		for (uint64_t m = mk_off[r]; m < mk_off[r + 1]; m++) if (mk_pos[m] < L) out[rel[r] + mk_pos[m]] = mk_char[m];
	}
}

/* rel[i] = offsets[i] - offsets[0], i = 0 .. n: a call's reads addressed from its first base */
__global__ void offsets_rel_kernel(const uint64_t *offsets, uint64_t n, uint64_t *rel) {
	const uint64_t o0 = offsets[0];
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x) rel[i] = offsets[i] - o0;
}

__global__ void ingest_shift_quals(uint8_t *quals, uint64_t n, int delta) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) quals[i] = (uint8_t)(quals[i] + delta);
}

}  // namespace kmr
#endif
