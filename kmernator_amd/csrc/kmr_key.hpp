/*
 * kmr_key.hpp -- packed k-mer arithmetic shared by host and device code.
 *
 * A k-mer is held as W 64-bit words, word 0 most significant, bases
 * left-justified (first base in the top two bits), pad bits zero.  That is the
 * reference's TwoBitEncoding byte string (src/TwoBitSequence.cpp:242-269) read
 * big-endian, so unsigned word-wise comparison equals the reference's memcmp
 * order (Kmer::compare, src/Kmer.h:311-313) and canonical selection is a
 * numeric min (Kmer::buildLeastComplement, src/Kmer.h:356-364).
 *
 * The hash is Bob Jenkins' lookup3 hashlittle2 exactly as KmerHasher::getHash
 * calls it (src/Kmer.h:207-230, src/lookup3.h:470-641): over the kb =
 * ceil(k/4) key bytes, pc = 0xDEADBEEF, pb = 0, result c | b << 32.
 */
#ifndef KMR_KEY_HPP_
#define KMR_KEY_HPP_

#include <stdint.h>

#if defined(__HIPCC__)
#define KMR_HD __host__ __device__ __forceinline__
#else
#define KMR_HD inline
#endif

namespace kmr {

template <int W> struct Key {
	uint64_t w[W];
};

template <int W> KMR_HD bool key_eq(const Key<W> &a, const Key<W> &b) {
	bool e = true;
#pragma unroll
	for (int i = 0; i < W; i++) e = e && (a.w[i] == b.w[i]);
	return e;
}
/* a < b in memcmp order */
template <int W> KMR_HD bool key_lt(const Key<W> &a, const Key<W> &b) {
#pragma unroll
	for (int i = 0; i < W; i++) {
		if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
	}
	return false;
}
template <int W> KMR_HD bool key_le(const Key<W> &a, const Key<W> &b) { return !key_lt<W>(b, a); }

/* Rolling window state for one read: forward k-mer and its reverse complement,
 * both left-justified.  push(base) advances the window by one base. */
template <int W> struct Roller {
	Key<W> fwd, rc;
	/* geometry, identical for every lane: position of base k-1 */
	int lastWord;       /* word holding base k-1                      */
	int lastShift;      /* bit shift of base k-1 inside that word     */
	uint64_t lastMask;  /* bits of lastWord that belong to the k-mer  */

	KMR_HD void init(uint32_t k) {
#pragma unroll
		for (int i = 0; i < W; i++) { fwd.w[i] = 0; rc.w[i] = 0; }
		uint32_t bit = 2 * (k - 1);          /* bits before base k-1 */
		lastWord = (int)(bit >> 6);
		lastShift = 62 - (int)(bit & 63);
		lastMask = ~0ull << lastShift;
	}
	KMR_HD void push(uint32_t b) {
		/* fwd <<= 2 bases-wise, insert b at base position k-1 */
#pragma unroll
		for (int i = 0; i < W; i++) {
			uint64_t nxt = (i + 1 < W) ? fwd.w[i + 1] : 0ull;
			fwd.w[i] = (fwd.w[i] << 2) | (nxt >> 62);
		}
#pragma unroll
		for (int i = 0; i < W; i++)
			if (i == lastWord) fwd.w[i] |= ((uint64_t)b << lastShift);
		/* rc >>= 2, insert complement at base position 0, drop what leaves the window */
#pragma unroll
		for (int i = W - 1; i >= 0; i--) {
			uint64_t prv = (i > 0) ? rc.w[i - 1] : 0ull;
			rc.w[i] = (rc.w[i] >> 2) | (prv << 62);
		}
		rc.w[0] |= ((uint64_t)(3u - b) << 62);
#pragma unroll
		for (int i = 0; i < W; i++) {
			if (i == lastWord) rc.w[i] &= lastMask;
			else if (i > lastWord) rc.w[i] = 0;
		}
	}
	KMR_HD Key<W> getFwd() const { return fwd; }
	KMR_HD Key<W> getRc() const { return rc; }
};

/* One-word k-mers (k <= 32) roll in 32-bit halves: 64-bit shifts run at a fraction of the 32-bit rate on CDNA, and the
 * two funnel shifts below are single v_alignbit_b32 instructions. */
template <> struct Roller<1> {
	uint32_t fh, fl, rh, rl;       /* forward and reverse-complement word, high and low half */
	int lastShift;                 /* bit shift of base k-1 in the 64-bit word */
	uint32_t maskH, maskL;         /* bits of the word that belong to the k-mer */
	KMR_HD void init(uint32_t k) {
		fh = fl = rh = rl = 0;
		lastShift = 62 - (int)(2 * (k - 1));
		const uint64_t m = ~0ull << lastShift;
		maskH = (uint32_t)(m >> 32); maskL = (uint32_t)m;
	}
	KMR_HD void push(uint32_t b) {
		fh = (fh << 2) | (fl >> 30); fl <<= 2;
		if (lastShift >= 32) fh |= b << (lastShift - 32); else fl |= b << lastShift;
		rl = (rl >> 2) | (rh << 30); rh = (rh >> 2) | ((3u - b) << 30);
		rh &= maskH; rl &= maskL;
	}
	KMR_HD Key<1> getFwd() const { Key<1> k; k.w[0] = ((uint64_t)fh << 32) | fl; return k; }
	KMR_HD Key<1> getRc() const { Key<1> k; k.w[0] = ((uint64_t)rh << 32) | rl; return k; }
};

KMR_HD uint32_t rot32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
KMR_HD uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

#define KMR_MIX(a, b, c) { \
	a -= c; a ^= rot32(c, 4);  c += b; \
	b -= a; b ^= rot32(a, 6);  a += c; \
	c -= b; c ^= rot32(b, 8);  b += a; \
	a -= c; a ^= rot32(c, 16); c += b; \
	b -= a; b ^= rot32(a, 19); a += c; \
	c -= b; c ^= rot32(b, 4);  b += a; }
#define KMR_FINAL(a, b, c) { \
	c ^= b; c -= rot32(b, 14); \
	a ^= c; a -= rot32(c, 11); \
	b ^= a; b -= rot32(a, 25); \
	c ^= b; c -= rot32(b, 16); \
	a ^= c; a -= rot32(c, 4);  \
	b ^= a; b -= rot32(a, 14); \
	c ^= b; c -= rot32(b, 24); }

/* little-endian 32-bit word m of the key's byte string (bytes past 8*W read 0) */
template <int W> KMR_HD uint32_t key_le32(const Key<W> &key, int m) {
	if (m >= 2 * W) return 0u;
	uint64_t w = key.w[m >> 1];
	uint32_t be = (m & 1) ? (uint32_t)w : (uint32_t)(w >> 32);
	return bswap32(be);
}

/* KmerHasher::getHash over kb bytes.  Pad bytes inside the last 12-byte block
 * are zero in our representation, so adding whole words equals the reference's
 * masked tail reads (src/lookup3.h:527-541). */
/* lookup8 (Bob Jenkins, lookup8.c 1997, public domain; src/lookup8.h:90-160 hash(), = hash3() on a little-endian machine) over the
 * kb key bytes with level 0xDEADBEEF, the initial value KmerHasher::getHash used with it before it moved to lookup3
 * (src/Kmer.h:210-212).  The reference no longer calls it: selectable (kmr_config.hash_kind), parity unpinned.
 * Little-endian 8-byte words of the byte string = byte-swapped words of the key; pad bytes are zero, so whole words can be added
 * where the reference adds the remaining bytes one by one. */
#define KMR_MIX64(a, b, c) { \
	a -= b; a -= c; a ^= (c >> 43); b -= c; b -= a; b ^= (a << 9);  c -= a; c -= b; c ^= (b >> 8); \
	a -= b; a -= c; a ^= (c >> 38); b -= c; b -= a; b ^= (a << 23); c -= a; c -= b; c ^= (b >> 5); \
	a -= b; a -= c; a ^= (c >> 35); b -= c; b -= a; b ^= (a << 49); c -= a; c -= b; c ^= (b >> 11); \
	a -= b; a -= c; a ^= (c >> 12); b -= c; b -= a; b ^= (a << 18); c -= a; c -= b; c ^= (b >> 22); }
KMR_HD uint64_t kmr_bswap64(uint64_t x) {
	x = ((x & 0x00ff00ff00ff00ffull) << 8) | ((x >> 8) & 0x00ff00ff00ff00ffull);
	x = ((x & 0x0000ffff0000ffffull) << 16) | ((x >> 16) & 0x0000ffff0000ffffull);
	return (x << 32) | (x >> 32);
}
template <int W> KMR_HD uint64_t key_hash8(const Key<W> &key, uint32_t kb) {
	uint64_t a, b, c;
	a = b = 0xDEADBEEFull;
	c = 0x9e3779b97f4a7c13ull;
	uint32_t len = kb; int m = 0;
	if constexpr (W >= 3) {
		if (len >= 24) {      /* one full block of 24 bytes (kb <= 32) */
			a += kmr_bswap64(key.w[0]); b += kmr_bswap64(key.w[1]); c += kmr_bswap64(key.w[2]);
			KMR_MIX64(a, b, c);
			len -= 24; m = 3;
		}
	}
	c += kb;
	/* the last 0..23 bytes: bytes 0-7 into a, 8-15 into b, 16-22 into c above its first byte (reserved for the length) */
	if (len > 0) a += kmr_bswap64(key.w[m < W ? m : W - 1]);
	if (len > 8) b += kmr_bswap64(key.w[m + 1 < W ? m + 1 : W - 1]);
	if (len > 16) c += kmr_bswap64(key.w[m + 2 < W ? m + 2 : W - 1]) << 8;
	KMR_MIX64(a, b, c);
	return c;
}

/* kb: the key bytes in the low 16 bits, the hash kind (kmr_config.hash_kind) above them -- one argument, so that every kernel
 * that already hands "kb" to the hash serves both kinds */
template <int W> KMR_HD uint64_t key_hash(const Key<W> &key, uint32_t kbAndKind) {
	const uint32_t kb = kbAndKind & 0xffffu;
	if (kbAndKind >> 16) return key_hash8<W>(key, kb);
	uint32_t a, b, c;
	a = b = c = 0xdeadbeefu + kb + 0xDEADBEEFu;
	int m = 0;
	uint32_t length = kb;
#pragma unroll
	for (int blk = 0; blk < (8 * W + 11) / 12; blk++) {
		if (length > 12) {
			a += key_le32<W>(key, m); b += key_le32<W>(key, m + 1); c += key_le32<W>(key, m + 2);
			KMR_MIX(a, b, c);
			length -= 12; m += 3;
		}
	}
	/* last block: 1..12 bytes (kb >= 1 always) */
	a += key_le32<W>(key, m);
	if (length > 4) b += key_le32<W>(key, m + 1);
	if (length > 8) c += key_le32<W>(key, m + 2);
	KMR_FINAL(a, b, c);
	return (uint64_t)c | ((uint64_t)b << 32);
}

/* byte j of the reference byte string */
template <int W> KMR_HD uint8_t key_byte(const Key<W> &key, uint32_t j) {
	return (uint8_t)(key.w[j >> 3] >> (56 - 8 * (j & 7)));
}
template <int W> KMR_HD void key_from_bytes(Key<W> &key, const uint8_t *p, uint32_t kb) {
#pragma unroll
	for (int i = 0; i < W; i++) key.w[i] = 0;
	for (uint32_t j = 0; j < kb; j++) key.w[j >> 3] |= (uint64_t)p[j] << (56 - 8 * (j & 7));
}

/* BucketExposedMapLogic (src/Kmer.h:2269-2295,2329-2333) */
static const int DMP_HASH_SHIFT = 24;
static const uint64_t DMP_HASH_MASK = 0x7ffff;
KMR_HD uint32_t distributed_thread_id(uint64_t hash, uint32_t n) {
	return n > 1 ? (uint32_t)(((hash >> DMP_HASH_SHIFT) & DMP_HASH_MASK) % n) : 0u;
}

/* position in the device hash table: a multiplicative re-mix of the lookup3
 * value, because its low 26 bits are the bucket index and bits 24..42 select
 * the owner, so neither range is uniformly populated inside one handle */
KMR_HD uint64_t table_slot(uint64_t hash, uint32_t log2cap) {
	return (hash * 0x9E3779B97F4A7C15ull) >> (64 - log2cap);
}

}  // namespace kmr
#endif
