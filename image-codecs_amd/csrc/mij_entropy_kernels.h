/*
 * mij_entropy_kernels.h -- the baseline Huffman walk on the GPU ("self-synchronising
 * GPU Huffman"), for the layout that makes up batch work: a single interleaved baseline scan (what the
 * reference's own writer emits, codec/jpeg_write.c:283-352), with or without restart intervals -- every interval
 * is one independent DevScan.
 *
 * A JPEG entropy segment has no entry points, but Huffman codes re-synchronise: a decoder started at a
 * wrong bit position falls into step with the true symbol sequence after a few symbols.  The unstuffed
 * bitstream of an image is cut into subsequences of MIJ_ES_BITS bits, one thread each:
 *   1. k_es_cold     every thread decodes its subsequence from a guessed state (block start, at its first bit)
 *                    and records the state it is in when it crosses into the next subsequence
 *   2. k_es_sync     rounds: thread i restarts from the end state of thread i-1 if that differs from what it
 *                    started from last time; the true chain from subsequence 0 wins; stops changing after a
 *                    few rounds because wrong starts re-synchronise inside one subsequence
 *   3. k_es_offsets  prefix sum of the blocks completed per subsequence -> the block ordinal each one starts at
 *   4. k_es_write    decode once more, now knowing where every coefficient goes: straight into the (cleared)
 *                    planes the IDCT kernels read -- compact planes by default (low byte into the tile, and for a
 *                    value outside -128..127 its escape byte and the block's flag: nothing is handed back for
 *                    size), int16 tile layout on request -- DC differences aside;
 *                    k_es_tails adds the rest of blocks that began in the previous subsequence
 *   5. k_es_dc       per component running sum of the DC differences (codec/jpeg.c:323-325), per-block L1
 *                    bound (MIJ_FLAG_WIDE_IDCT), completion checks, the left-over-0xff rule
 * Symbol decoding is the reference's (codec/jpeg.c:193-265: 9-bit fast table, maxcode/delta slow path,
 * extend_receive), so a well-formed stream yields exactly the host walk's coefficients.  Anything else --
 * an invalid code, a run past coefficient 63, a DC category above 11, a stream that ends early, unread data
 * that still holds a stuffed 0xff (the reference then fails with "unknown marker"), no convergence -- raises the image's anomaly flag and the caller re-does that image on the host, whose
 * behaviour on malformed input is the reference's.
 */
#ifndef MIJ_ENTROPY_KERNELS_H
#define MIJ_ENTROPY_KERNELS_H

#include "mij_kernels.h"

namespace mij {

/* Bits per subsequence.  A wrong start has to fall into step not only with the bit position but with the place in
 * the MCU as well (the luma and chroma tables differ), which takes a few MCUs: 4096 bits hold about six 4:2:0 MCUs
 * at 1.8 bit/px, so most wrong starts are right again before their subsequence ends. */
#ifndef MIJ_ES_BITS
#define MIJ_ES_BITS 4096u
#endif
#define MIJ_ES_DEAD 127u  /* z of a state whose decode hit an invalid code */

struct DevHuff { /* stbi__huffman without the code[] array (codec/jpeg.c:21-32) */
	uint8_t fast[512];
	uint8_t size[256];
	uint8_t values[256];
	uint32_t maxcode[18];
	int32_t delta[18];
};

struct DevScan {
	uint64_t stream_off; /* unstuffed entropy bytes in the stream arena (16 zero bytes follow) */
	uint32_t nbits;      /* 8 * bytes */
	uint32_t nsub, sub_off; /* subsequences and where their state slots start */
	uint32_t img;        /* DevImage index */
	uint32_t nblocks;    /* blocks the scan must produce: mcu_x * mcu_y * bpm */
	uint32_t blk_off;    /* where this image's per-block arrays (DC differences, L1) start */
	uint32_t bpm, mcu_x;
	uint32_t first_mcu;  /* restart intervals: the MCU this segment starts at (0 without restart markers) */
	uint32_t last_seg;   /* the segment that ends at EOI */
	uint32_t fmt;        /* 1: compact planes (MIJ_DEV_COEF_BYTES): AC low bytes + escape bytes, DC in its own int16 array */
	uint8_t blk_comp[12], blk_dx[12], blk_dy[12]; /* block-in-MCU -> component and position inside the MCU */
	uint8_t dc_tab[4], ac_tab[4];                 /* component -> table index (0..3 DC, 4..7 AC) of this scan's eight tables */
	uint32_t tab_off;    /* first of the eight DevHuff of this scan */
	uint16_t qz[4][64];  /* quantisation tables, zigzag order, per component (L1 bound only) */
};

struct EsState {
	uint32_t p; /* bit position */
	uint32_t z; /* next coefficient index 0..63, or MIJ_ES_DEAD */
	uint32_t c; /* block inside the MCU */
};
__device__ __forceinline__ uint64_t es_pack(const EsState &s) { return (uint64_t)s.p | ((uint64_t)s.z << 32) | ((uint64_t)s.c << 40); }
__device__ __forceinline__ EsState es_unpack(uint64_t v)
{
	EsState s;
	s.p = (uint32_t)v;
	s.z = (uint32_t)(v >> 32) & 255u;
	s.c = (uint32_t)(v >> 40) & 255u;
	return s;
}

/* the bit window (big-endian bit order) kept in registers: at least 32 valid bits at every symbol start, one
 * aligned dword fetched whenever fewer are left (a symbol takes at most 16 + 16 bits) */
struct EsBits {
	const uint32_t *w;
	uint64_t win;
	uint32_t avail, idx, nxt; /* nxt: the dword after the window, fetched one refill ahead so that its latency overlaps the symbols in between */
	__device__ __forceinline__ void start(const uint8_t *__restrict__ stream, uint32_t p)
	{
		w = reinterpret_cast<const uint32_t *>(stream);
		idx = p >> 5;
		const uint32_t sh = p & 31u;
		win = (((uint64_t)__builtin_bswap32(w[idx]) << 32) | __builtin_bswap32(w[idx + 1])) << sh;
		avail = 64u - sh;
		nxt = w[idx + 2];
		idx += 3;
	}
	__device__ __forceinline__ void take(uint32_t n)
	{
		win <<= n;
		avail -= n;
		if (avail < 32u) {
			win |= (uint64_t)__builtin_bswap32(nxt) << (32u - avail);
			avail += 32u;
			nxt = w[idx++];
		}
	}
};

/* A table as the decode loop wants it in LDS: the reference's two-level fast path (fast[] -> size[], values[],
 * codec/jpeg.c:201-210) folded into one 16-bit entry per 9-bit prefix, because every LDS round trip is on the
 * serial chain from one symbol to the next. */
struct EsTab {
	uint16_t fast16[512]; /* code length << 8 | symbol; 0xffff = longer than 9 bits (or no such code) */
	uint8_t values[256];
	/* lengths 10..17 of stbi__huffman.maxcode / .delta (codec/jpeg.c:21-32), 16-byte aligned: the slow path fetches
	 * all of them at once instead of walking them one LDS round trip at a time */
	__attribute__((aligned(16))) uint32_t maxcode[8];
	__attribute__((aligned(16))) int32_t delta[8];
};

/* codec/jpeg.c:193-243: returns the symbol and its code length, or -1 */
__device__ __forceinline__ int es_symbol(const EsTab &h, uint64_t win, uint32_t &len)
{
	const uint32_t top16 = (uint32_t)(win >> 48);
	const uint32_t e = h.fast16[top16 >> 7];
	if (e != 0xffffu) {
		len = e >> 8;
		return (int)(e & 255u);
	}
	/* :219-221 "for (k = FAST_BITS+1;; ++k) if (temp < maxcode[k]) break": maxcode never decreases with the length
	 * (each is (code + count) << 1 of the one before, left-aligned), so the first length that holds the prefix is
	 * 10 + the number of shorter limits at or below it; maxcode[17] = 0xffffffff ends the count in a defined table */
	const uint4 m0 = *reinterpret_cast<const uint4 *>(&h.maxcode[0]), m1 = *reinterpret_cast<const uint4 *>(&h.maxcode[4]);
	const uint4 d0 = *reinterpret_cast<const uint4 *>(&h.delta[0]), d1 = *reinterpret_cast<const uint4 *>(&h.delta[4]);
	const uint32_t ge[7] = {top16 >= m0.x, top16 >= m0.y, top16 >= m0.z, top16 >= m0.w, top16 >= m1.x, top16 >= m1.y, top16 >= m1.z};
	const uint32_t l = 10u + ge[0] + ge[1] + ge[2] + ge[3] + ge[4] + ge[5] + ge[6];
	if (l >= 17u)
		return -1;
	uint32_t dl = d0.x;
	dl = l == 11u ? d0.y : dl;
	dl = l == 12u ? d0.z : dl;
	dl = l == 13u ? d0.w : dl;
	dl = l == 14u ? d1.x : dl;
	dl = l == 15u ? d1.y : dl;
	dl = l == 16u ? d1.z : dl;
	const int c = (int)((top16 >> (16u - l)) & ((1u << l) - 1u)) + (int)dl;
	if (c < 0 || c > 255)
		return -1;
	len = l;
	return h.values[c];
}

/* codec/jpeg.c:250-265 on the n bits that follow the code */
__device__ __forceinline__ int es_extend(uint64_t win, uint32_t len, uint32_t n)
{
	const uint32_t bits = (uint32_t)((win << len) >> (64u - n));
	const int neg = !(bits >> (n - 1u));
	return neg ? (int)bits - (int)((1u << n) - 1u) : (int)bits;
}

/* What the decode loop looks up per symbol or per block, copied into LDS once per workgroup: indexed by a per-lane
 * block-in-MCU number these would otherwise be dependent global loads inside a divergent, serial loop. */
struct EsLocal {
	uint32_t tabs[12];  /* block-in-MCU -> component | DC table << 8 | AC table << 16 */
	uint32_t geo[12];   /* h | v << 8 | dx << 16 | dy << 24 (write passes only) */
	uint32_t bw[12];    /* the component's plane width in blocks */
	uint64_t plane[12]; /* byte offset of the component's plane in the coefficient arena */
	uint64_t hi[12];    /* compact planes: byte offset of the component's escape bytes */
	uint16_t qz[4][64]; /* DevScan.qz */
};

struct EsWriter { /* where the blocks of the write pass go */
	const DevScan *sc;
	const EsLocal *loc;
	int16_t *coef;       /* coefficient arena */
	int16_t *dcdiff;     /* per block */
	uint32_t *l1;        /* per block */
	const uint16_t *toff; /* zigzag index -> element offset inside the block's tile slot */
	const uint8_t *zpos;  /* zigzag index -> in-block position P (the order of a block's escape bytes) */
	bool skip;            /* the block in progress was begun by the previous subsequence: k_es_tails stores its rest */
	bool stop_after_block;
	bool owner;          /* k_es_write: this thread stores the L1 word of the blocks it begins (k_es_tails adds) */
	uint32_t ord;        /* ordinal of the current block */
	uint32_t mx, my;     /* its MCU */
	int16_t *blk;        /* its tile slot (int16 planes) */
	uint8_t *blk8;       /* its tile slot (compact planes) */
	uint8_t *hi8;        /* its 64 escape bytes (compact planes) */
	uint32_t L;          /* its index in the component's block grid */
	uint32_t acc;        /* L1 of the AC coefficients written by this thread into it */
	uint32_t *pfinal;    /* where the bit position after the scan's last block is recorded */
	__device__ __forceinline__ void locate(uint32_t c)
	{
		const uint32_t g = loc->geo[c];
		const uint32_t bx = mx * (g & 255u) + ((g >> 16) & 255u), by = my * ((g >> 8) & 255u) + (g >> 24);
		L = bx + by * loc->bw[c];
		uint8_t *plane = reinterpret_cast<uint8_t *>(coef) + loc->plane[c];
		blk = reinterpret_cast<int16_t *>(plane) + ((size_t)(L >> 6) << 12) + ((L & 63u) << 3);
		blk8 = plane + ((size_t)(L >> 6) << 12) + ((L & 63u) << 3);
		hi8 = reinterpret_cast<uint8_t *>(coef) + loc->hi[c] + ((size_t)L << 6);
	}
	__device__ __forceinline__ void put(uint32_t k, int v)
	{
		if (sc->fmt) {
			blk8[toff[k]] = (uint8_t)v; /* low byte */
			if ((uint32_t)(v + 128) > 255u) { /* escape: v == sext8(low) + 256 * h (mij_kernels.h, load_block_b8) */
				hi8[zpos[k]] = (uint8_t)((v + 128) >> 8);
				blk8[0] = 1; /* the block's flags byte sits in the DC's place; every writer stores the same value */
			}
		} else
			blk[toff[k]] = (int16_t)v;
	}
};

/*
 * Decode from state s until the bit position reaches p_end (a symbol that starts before p_end is finished).
 * WRITE = false: only the state and the number of completed blocks.  WRITE = true: coefficients are stored,
 * decoding stops at block ordinal sc.nblocks, malformed input sets *anom.
 */
template <bool WRITE>
__device__ __forceinline__ uint32_t es_decode(const DevScan &sc, const EsLocal &loc, const EsTab *__restrict__ tabs, const uint8_t *__restrict__ stream, EsState &s,
															 uint32_t p_end, EsWriter *wr, uint32_t *anom)
{
	uint32_t done = 0, guard = 0;
	const uint32_t limit = sc.nbits + 64u; /* the arena is zero padded: never read far past the data */
	EsBits br;
	br.start(stream, s.p);
	uint32_t tb = loc.tabs[s.c]; /* the current block's component and tables; changes with s.c only */
	while (s.p < p_end && s.z != MIJ_ES_DEAD) {
		if (++guard > MIJ_ES_BITS + 64u) { /* every symbol takes at least one bit: cannot happen, but a wave must always end */
			s.z = MIJ_ES_DEAD;
			break;
		}
		if (WRITE && wr->ord >= sc.nblocks)
			break;
		const uint64_t win = br.win;
		const uint32_t ci = tb & 255u;
		uint32_t len = 0;
		if (s.z == 0) {
			const int t = es_symbol(tabs[(tb >> 8) & 255u], win, len);
			if (t < 0 || t > 11 || len == 0) { /* the reference takes categories up to 16; nothing a conforming stream uses */
				if (!WRITE) { /* a guessed start ran into a non-code: slip one bit and keep looking for the true sequence */
					s.p += 1;
					br.take(1);
					continue;
				}
				atomicOr(anom, 1u);
				s.z = MIJ_ES_DEAD;
				break;
			}
			const int diff = t ? es_extend(win, len, (uint32_t)t) : 0;
			if (WRITE)
				wr->dcdiff[wr->ord] = (int16_t)diff;
			s.p += len + (uint32_t)t;
			br.take(len + (uint32_t)t);
			s.z = 1;
		} else {
			const int rs = es_symbol(tabs[tb >> 16], win, len);
			if (rs < 0 || len == 0) {
				if (!WRITE) {
					s.p += 1;
					br.take(1);
					continue;
				}
				atomicOr(anom, 1u);
				s.z = MIJ_ES_DEAD;
				break;
			}
			const uint32_t r = (uint32_t)rs >> 4, n = (uint32_t)rs & 15u;
			if (n == 0) {
				s.p += len;
				br.take(len);
				if (r == 15u)
					s.z += 16; /* ZRL */
				else if (r == 0u)
					s.z = 64; /* EOB */
				else { /* EOBn only exists in progressive scans; the reference treats it like EOB here (:355) */
					s.z = 64;
				}
			} else {
				const uint32_t k = s.z + r;
				if (k > 63u) { /* the reference would write through its padded de-zigzag table: leave that to the host */
					if (WRITE)
						atomicOr(anom, 1u);
					s.p += len + n;
					br.take(len + n);
					s.z = 64;
				} else {
					const int v = es_extend(win, len, n);
					if (WRITE && !wr->skip) {
						wr->put(k, v);
						const int dq = (int)(int16_t)((uint32_t)v * loc.qz[ci][k]);
						wr->acc += (uint32_t)(dq < 0 ? -dq : dq);
					}
					s.p += len + n;
					br.take(len + n);
					s.z = k + 1;
				}
			}
		}
		if (s.z >= 64u) { /* block complete (ZRL past the end ends it too: same as the host loop's k < 64 test) */
			s.z = 0;
			++done;
			if (WRITE) {
				if (s.p > sc.nbits)
					atomicOr(anom, 16u); /* the data ran out inside this block: the reference decodes on with zero bits */
				/* per-block L1 of the AC coefficients without atomics: whoever holds the block's start stores, the
				 * thread that finishes a block begun elsewhere (k_es_tails, a later launch) adds */
				if (wr->owner) {
					if (!wr->skip)
						wr->l1[wr->ord] = wr->acc;
				} else
					wr->l1[wr->ord] += wr->acc;
				wr->acc = 0;
				if (wr->stop_after_block) { /* k_es_tails: only the rest of the block the subsequence started in */
					s.z = 0;
					break;
				}
				wr->skip = false;
				if (++wr->ord == sc.nblocks)
					*wr->pfinal = s.p;
			}
			if (++s.c == sc.bpm) {
				s.c = 0;
				if (WRITE) {
					if (++wr->mx == sc.mcu_x) {
						wr->mx = 0;
						++wr->my;
					}
				}
			}
			tb = loc.tabs[s.c];
			if (WRITE && wr->ord < sc.nblocks)
				wr->locate(s.c);
		}
		if (s.p > limit) {
			s.z = MIJ_ES_DEAD;
			break;
		}
	}
	if (WRITE && wr->owner && wr->ord < sc.nblocks && !wr->skip && s.z != 0 && s.z != MIJ_ES_DEAD)
		wr->l1[wr->ord] = wr->acc; /* a block that continues in the next subsequence: the L1 of its head */
	return done;
}

/* every kernel below: grid.x = blocks of 256 subsequences over a (scan, first subsequence) work list */
struct EsWork {
	uint32_t scan, first;
};

/* the scan's eight Huffman tables and its EsLocal into LDS (im == nullptr: no block placement needed); ends in a barrier */
__device__ __forceinline__ void es_load_tables(const DevScan &sc, const DevImage *im, const DevHuff *__restrict__ g, EsTab *l, EsLocal *loc)
{
	for (uint32_t i = threadIdx.x; i < 8u * 512u; i += blockDim.x) {
		const DevHuff &h = g[i >> 9];
		const uint32_t k = h.fast[i & 511u]; /* 255 = not in the fast table; entry 255 itself is never fast (:203) */
		l[i >> 9].fast16[i & 511u] = (uint16_t)(k < 255u ? (uint32_t)h.size[k] << 8 | h.values[k] : 0xffffu);
	}
	for (uint32_t i = threadIdx.x; i < 8u * 256u; i += blockDim.x)
		l[i >> 8].values[i & 255u] = g[i >> 8].values[i & 255u];
	if (threadIdx.x < 64u) {
		const uint32_t t = threadIdx.x >> 3, j = threadIdx.x & 7u;
		l[t].maxcode[j] = g[t].maxcode[10u + j];
		l[t].delta[j] = g[t].delta[10u + j];
	}
	if (threadIdx.x < 12u) {
		const uint32_t c = threadIdx.x, ci = sc.blk_comp[c] & 3u;
		loc->tabs[c] = ci | (uint32_t)(sc.dc_tab[ci] & 7u) << 8 | (uint32_t)(sc.ac_tab[ci] & 7u) << 16;
		if (im) {
			const DevComp &cp = im->comp[ci];
			loc->geo[c] = ((uint32_t)cp.h & 255u) | ((uint32_t)cp.v & 255u) << 8 | (uint32_t)sc.blk_dx[c] << 16 | (uint32_t)sc.blk_dy[c] << 24;
			loc->bw[c] = (uint32_t)cp.bw;
			loc->plane[c] = cp.coef_off;
			loc->hi[c] = cp.hi_off;
		}
	}
	loc->qz[threadIdx.x >> 6][threadIdx.x & 63u] = sc.qz[threadIdx.x >> 6][threadIdx.x & 63u];
	__syncthreads();
}

__global__ __launch_bounds__(256) void k_es_cold(const DevScan *__restrict__ scans, const EsWork *__restrict__ work, const DevHuff *__restrict__ huff,
																 const uint8_t *__restrict__ streams, uint64_t *__restrict__ start, uint64_t *__restrict__ end, uint32_t *__restrict__ cnt)
{
	__shared__ EsTab tabs[8];
	__shared__ EsLocal loc;
	const EsWork wk = work[blockIdx.x];
	const DevScan &sc = scans[wk.scan];
	es_load_tables(sc, nullptr, huff + sc.tab_off, tabs, &loc);
	const uint32_t i = wk.first + threadIdx.x;
	if (i >= sc.nsub)
		return;
	EsState s;
	s.p = i * MIJ_ES_BITS;
	s.z = 0;
	s.c = 0;
	start[sc.sub_off + i] = es_pack(s);
	const uint32_t pe = min((i + 1u) * MIJ_ES_BITS, sc.nbits);
	cnt[sc.sub_off + i] = es_decode<false>(sc, loc, tabs, streams + sc.stream_off, s, pe, nullptr, nullptr);
	end[sc.sub_off + i] = es_pack(s);
}

/* one synchronisation round: end_in is the previous round's result, end_out this round's */
__global__ __launch_bounds__(256) void k_es_sync(const DevScan *__restrict__ scans, const EsWork *__restrict__ work, const DevHuff *__restrict__ huff,
																 const uint8_t *__restrict__ streams, uint64_t *__restrict__ start, const uint64_t *__restrict__ end_in,
																 uint64_t *__restrict__ end_out, uint32_t *__restrict__ cnt, uint32_t *__restrict__ changed)
{
	__shared__ EsTab tabs[8];
	__shared__ EsLocal loc;
	const EsWork wk = work[blockIdx.x];
	const DevScan &sc = scans[wk.scan];
	es_load_tables(sc, nullptr, huff + sc.tab_off, tabs, &loc);
	const uint32_t i = wk.first + threadIdx.x;
	if (i >= sc.nsub)
		return;
	const uint32_t slot = sc.sub_off + i;
	if (i == 0) {
		end_out[slot] = end_in[slot];
		return;
	}
	const uint64_t want = end_in[slot - 1];
	if (want == start[slot]) {
		end_out[slot] = end_in[slot];
		return;
	}
	start[slot] = want;
	EsState s = es_unpack(want);
	const uint32_t pe = min((i + 1u) * MIJ_ES_BITS, sc.nbits);
	cnt[slot] = es_decode<false>(sc, loc, tabs, streams + sc.stream_off, s, pe, nullptr, nullptr);
	end_out[slot] = es_pack(s);
	atomicAdd(&changed[wk.scan], 1u);
}

/* exclusive prefix sum of cnt over a scan's subsequences (one workgroup per scan) */
__global__ __launch_bounds__(256) void k_es_offsets(const DevScan *__restrict__ scans, const uint32_t *__restrict__ cnt, uint32_t *__restrict__ base,
																	 uint32_t *__restrict__ total)
{
	__shared__ uint32_t part[256];
	const DevScan &sc = scans[blockIdx.x];
	const uint32_t per = (sc.nsub + 255u) / 256u;
	const uint32_t lo = min(threadIdx.x * per, sc.nsub), hi = min(lo + per, sc.nsub);
	uint32_t sum = 0;
	for (uint32_t i = lo; i < hi; ++i)
		sum += cnt[sc.sub_off + i];
	part[threadIdx.x] = sum;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t run = 0;
		for (int t = 0; t < 256; ++t) {
			const uint32_t v = part[t];
			part[t] = run;
			run += v;
		}
		total[blockIdx.x] = run;
	}
	__syncthreads();
	uint32_t run = part[threadIdx.x];
	for (uint32_t i = lo; i < hi; ++i) {
		base[sc.sub_off + i] = run;
		run += cnt[sc.sub_off + i];
	}
}

/* Every coefficient is stored where it belongs in planes cleared beforehand (hipMemsetAsync): no staging, so the
 * pass runs at the occupancy of the cold pass.  (Round 1 also had a variant that staged each block in LDS and stored
 * it whole: 37 KiB of LDS per workgroup, 3 waves per SIMD, 5.4 ms against 3.4 ms per 256 images -- removed.) */
__global__ __launch_bounds__(256) void k_es_write(const DevScan *__restrict__ scans, const EsWork *__restrict__ work, const DevHuff *__restrict__ huff,
																  const uint8_t *__restrict__ streams, const DevImage *__restrict__ imgs, const uint64_t *__restrict__ start,
																  const uint32_t *__restrict__ base, int16_t *__restrict__ coef, int16_t *__restrict__ dcdiff, uint32_t *__restrict__ l1,
																  uint32_t *__restrict__ anom, uint32_t *__restrict__ pfinal)
{
	__shared__ EsTab tabs[8];
	__shared__ EsLocal loc;
	__shared__ uint8_t zpos[64];
	__shared__ uint16_t toff[64];
	const EsWork wk = work[blockIdx.x];
	const DevScan &sc = scans[wk.scan];
	if (threadIdx.x < 64) {
		const uint32_t P = mij_zigzag_pos[threadIdx.x];
		zpos[threadIdx.x] = (uint8_t)P;
		toff[threadIdx.x] = (uint16_t)(((P >> 3) << 9) + (P & 7u));
	}
	es_load_tables(sc, &imgs[sc.img], huff + sc.tab_off, tabs, &loc);
	const uint32_t i = wk.first + threadIdx.x;
	if (i >= sc.nsub)
		return;
	const uint32_t slot = sc.sub_off + i;
	EsState s = es_unpack(start[slot]);
	EsWriter wr;
	wr.sc = &sc;
	wr.loc = &loc;
	wr.coef = coef;
	wr.dcdiff = dcdiff + sc.blk_off;
	wr.l1 = l1 + sc.blk_off;
	wr.toff = toff;
	wr.zpos = zpos;
	wr.owner = true;
	wr.skip = s.z != 0; /* begun by the previous subsequence */
	wr.stop_after_block = false;
	wr.ord = base[slot];
	wr.acc = 0;
	wr.pfinal = &pfinal[wk.scan];
	if (wr.ord >= sc.nblocks)
		return;
	const uint32_t ml = wr.ord / sc.bpm, m = sc.first_mcu + ml;
	if (wr.ord - ml * sc.bpm != s.c) { /* the chain is inconsistent: cannot happen after convergence */
		atomicOr(&anom[wk.scan], 2u);
		return;
	}
	wr.my = m / sc.mcu_x;
	wr.mx = m - wr.my * sc.mcu_x;
	wr.locate(s.c);
	const uint32_t pe = min((i + 1u) * MIJ_ES_BITS, sc.nbits);
	es_decode<true>(sc, loc, tabs, streams + sc.stream_off, s, pe, &wr, &anom[wk.scan]);
}

/* the rest of every block that began in the previous subsequence: single coefficients into the block that the
 * previous thread's k_es_write stored whole (stream order makes this the later write) */
__global__ __launch_bounds__(256) void k_es_tails(const DevScan *__restrict__ scans, const EsWork *__restrict__ work, const DevHuff *__restrict__ huff,
																  const uint8_t *__restrict__ streams, const DevImage *__restrict__ imgs, const uint64_t *__restrict__ start,
																  const uint32_t *__restrict__ base, int16_t *__restrict__ coef, int16_t *__restrict__ dcdiff, uint32_t *__restrict__ l1,
																  uint32_t *__restrict__ scratch)
{
	__shared__ EsTab tabs[8];
	__shared__ EsLocal loc;
	__shared__ uint8_t zpos[64];
	__shared__ uint16_t toff[64];
	const EsWork wk = work[blockIdx.x];
	const DevScan &sc = scans[wk.scan];
	if (threadIdx.x < 64) {
		const uint32_t P = mij_zigzag_pos[threadIdx.x];
		zpos[threadIdx.x] = (uint8_t)P;
		toff[threadIdx.x] = (uint16_t)(((P >> 3) << 9) + (P & 7u));
	}
	es_load_tables(sc, &imgs[sc.img], huff + sc.tab_off, tabs, &loc);
	const uint32_t i = wk.first + threadIdx.x;
	if (i >= sc.nsub)
		return;
	const uint32_t slot = sc.sub_off + i;
	EsState s = es_unpack(start[slot]);
	if (s.z == 0 || s.z == MIJ_ES_DEAD)
		return;
	EsWriter wr;
	wr.sc = &sc;
	wr.loc = &loc;
	wr.coef = coef;
	wr.dcdiff = dcdiff + sc.blk_off;
	wr.l1 = l1 + sc.blk_off;
	wr.toff = toff;
	wr.zpos = zpos;
	wr.owner = false;
	wr.skip = false;
	wr.stop_after_block = true;
	wr.ord = base[slot];
	wr.acc = 0;
	wr.pfinal = scratch; /* never reached: the walk stops at the end of this block */
	if (wr.ord >= sc.nblocks)
		return;
	const uint32_t ml = wr.ord / sc.bpm, m = sc.first_mcu + ml;
	if (wr.ord - ml * sc.bpm != s.c)
		return;
	wr.my = m / sc.mcu_x;
	wr.mx = m - wr.my * sc.mcu_x;
	wr.locate(s.c);
	/* k_es_write walked the same symbols and reported what there was to report: verdict bits go to a scratch word */
	es_decode<true>(sc, loc, tabs, streams + sc.stream_off, s, sc.nbits, &wr, scratch + 1);
}

/* DC prediction (codec/jpeg.c:323-325), L1 bound and the completion checks; one workgroup per scan */
__global__ __launch_bounds__(256) void k_es_dc(const DevScan *__restrict__ scans, const DevImage *__restrict__ imgs, const uint32_t *__restrict__ total, const uint32_t *__restrict__ changed, int16_t *__restrict__ coef,
															  const int16_t *__restrict__ dcdiff, const uint32_t *__restrict__ l1, uint32_t *__restrict__ anom,
															  uint32_t *__restrict__ l1max, const uint32_t *__restrict__ pfinal, const uint8_t *__restrict__ streams)
{
	__shared__ int part[256][4];
	__shared__ uint32_t wmax[256];
	const DevScan &sc = scans[blockIdx.x];
	const DevImage &im = imgs[sc.img];
	const uint32_t nmcu = sc.nblocks / sc.bpm;
	const uint32_t per = (nmcu + 255u) / 256u;
	const uint32_t lo = min(threadIdx.x * per, nmcu), hi = min(lo + per, nmcu);
	const int16_t *dd = dcdiff + sc.blk_off;
	const uint32_t *bl1 = l1 + sc.blk_off;
	if (threadIdx.x == 0) {
		if (total[blockIdx.x] < sc.nblocks)
			atomicOr(&anom[blockIdx.x], 4u); /* the stream ends before the last block */
		if (changed[blockIdx.x])
			atomicOr(&anom[blockIdx.x], 8u); /* the last synchronisation round still moved something */
	}
	/* After its last block the reference skips ahead to the next 0xff and takes the byte behind it for a marker
	 * (codec/jpeg.c:1727-1737): a stuffed 0xff00 left over in unread data makes it fail with "unknown marker".
	 * How many bytes its 32-bit look-ahead had already taken is not tracked here, so any 0xff data byte behind
	 * the byte of the final bit position sends the image to the host walk.
	 * Before a restart marker the rule is different: the reference only sees the marker if its refill (to 24 bits,
	 * :1181) reaches it, otherwise it quietly stops decoding (:1183); with less than a byte of padding behind the
	 * final bit position it certainly does, anything more goes to the host walk. */
	if (sc.last_seg) {
		const uint8_t *st = streams + sc.stream_off;
		const uint32_t nbytes = sc.nbits >> 3;
		uint32_t found = 0;
		for (uint32_t q = ((pfinal[blockIdx.x] + 7u) >> 3) + threadIdx.x; q < nbytes; q += 256) /* the byte holding the last bit was certainly read */
			found |= st[q] == 0xffu;
		if (found)
			atomicOr(&anom[blockIdx.x], 32u);
	} else if (threadIdx.x == 0 && (pfinal[blockIdx.x] > sc.nbits || sc.nbits - pfinal[blockIdx.x] >= 8u))
		atomicOr(&anom[blockIdx.x], 64u);
	int sum[4] = {0, 0, 0, 0};
	for (uint32_t m = lo; m < hi; ++m)
		for (uint32_t c = 0; c < sc.bpm; ++c)
			sum[sc.blk_comp[c]] += dd[m * sc.bpm + c];
	for (int k = 0; k < 4; ++k)
		part[threadIdx.x][k] = sum[k];
	__syncthreads();
	if (threadIdx.x < 4) {
		int run = 0;
		for (int t = 0; t < 256; ++t) {
			const int v = part[t][threadIdx.x];
			part[t][threadIdx.x] = run;
			run = (int)((unsigned)run + (unsigned)v);
		}
	}
	__syncthreads();
	int pred[4];
	for (int k = 0; k < 4; ++k)
		pred[k] = part[threadIdx.x][k];
	uint32_t mymax = 0;
	uint32_t my = (sc.first_mcu + lo) / sc.mcu_x, mx = (sc.first_mcu + lo) - my * sc.mcu_x;
	for (uint32_t m = lo; m < hi; ++m) {
		for (uint32_t c = 0; c < sc.bpm; ++c) {
			const uint32_t ci = sc.blk_comp[c];
			const DevComp &cp = im.comp[ci];
			pred[ci] = (int)((unsigned)pred[ci] + (unsigned)(int)dd[m * sc.bpm + c]);
			const uint32_t bx = mx * (uint32_t)cp.h + sc.blk_dx[c], by = my * (uint32_t)cp.v + sc.blk_dy[c];
			const uint32_t L = bx + by * (uint32_t)cp.bw;
			if (sc.fmt)
				reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(coef) + cp.dc_off)[L] = (int16_t)pred[ci];
			else
				(reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(coef) + cp.coef_off) + ((size_t)(L >> 6) << 12) + ((L & 63u) << 3))[0] = (int16_t)pred[ci];
			const int dq = (int)(int16_t)((uint32_t)pred[ci] * sc.qz[ci][0]);
			const uint32_t tot = bl1[m * sc.bpm + c] + (uint32_t)(dq < 0 ? -dq : dq);
			mymax = tot > mymax ? tot : mymax;
		}
		if (++mx == sc.mcu_x) {
			mx = 0;
			++my;
		}
	}
	wmax[threadIdx.x] = mymax;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t mxv = 0;
		for (int t = 0; t < 256; ++t)
			mxv = wmax[t] > mxv ? wmax[t] : mxv;
		l1max[blockIdx.x] = mxv;
	}
}

} /* namespace mij */

#endif
