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
 *                    few rounds because wrong starts re-synchronise inside one subsequence.  From the second
 *                    round on (k_es_syncq) the subsequences that have to run again come off a queue the round before
 *                    filled, packed into full wavefronts
 *   3. k_es_offsets  prefix sum of the blocks completed per subsequence -> the block ordinal each one starts at
 *   4. k_es_write    decode once more, now knowing where every coefficient goes, DC differences aside.  Compact
 *                    planes (the default): into a cleared intermediate image of 64 bytes per block in ZIGZAG order,
 *                    block after block in scan order -- consecutive coefficients are neighbouring bytes, so a lane
 *                    gathers them in a 64-bit register and stores eight at a time (2-3 stores per block instead of
 *                    one scattered store per coefficient); a value outside -128..127 also gets its escape byte
 *                    (nothing is handed back for size).  int16 tile layout (on request): every coefficient straight
 *                    to its place in the cleared planes.
 *                    k_es_tails adds the rest of blocks that began in the previous subsequence (single bytes);
 *      k_es_pack     intermediate image -> the tile order of the compact planes (one block per lane, coalesced)
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
/* A batch of one picture (stbi_load_from_memory) is a latency problem, not a throughput problem: every pass lasts as long as ONE lane
 * needs for its subsequence (0.4 ms for 4096 bits) and a 1080p stream is only 900 of them.  Such batches cut the stream into
 * MIJ_ES_BITS_SINGLE bits per lane and pay with more (cheap) synchronisation rounds; the length is a field of DevScan. */
#ifndef MIJ_ES_BITS_SINGLE
#define MIJ_ES_BITS_SINGLE 1024u
#endif
/* Threads per workgroup of the passes that walk the stream (cold, sync, write, tails): one subsequence per thread.  The Huffman and pair
 * tables take 28 KiB of LDS per workgroup whatever its size, so the workgroup size sets how many wavefronts a CU can hold (160 KiB: five
 * workgroups): 256 threads = 5 waves per SIMD, 512 = 10 (then the registers decide). */
#ifndef MIJ_ES_WG
#define MIJ_ES_WG 256u
#endif
#ifndef MIJ_ES_WAVES /* A/B: waves per SIMD the compiler must leave room for (its register budget); 0 = its own choice */
#define MIJ_ES_WAVES 0
#endif
#if MIJ_ES_WAVES
#define MIJ_ES_KERNEL __global__ __launch_bounds__(MIJ_ES_WG) __attribute__((amdgpu_waves_per_eu(MIJ_ES_WAVES, MIJ_ES_WAVES)))
#else
#define MIJ_ES_KERNEL __global__ __launch_bounds__(MIJ_ES_WG)
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
	uint32_t sub_bits;   /* bits per subsequence (MIJ_ES_BITS, or less for a batch that cannot fill the GPU otherwise) */
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

/* Pair tables: a table indexed by the next MIJ_ES_PAIR_BITS bits of the stream folds TWO consecutive AC symbols (code + extra bits of the
 * first, code + extra bits of the second) into one 16-bit entry whenever both lie inside the window -- at the benchmark's 1.8 bit/px that
 * is most pairs -- so the serial chain lookup -> shift -> lookup runs half as often.  The state-only passes have their form of it in
 * EsUni below (round 2 kept it here).  The write pass needs the symbols themselves: EsPair holds its AC tables in the fast table's format
 * under a twelve-bit index, and an iteration looks up a second symbol behind the first in the same table.  A second symbol is only taken
 * when the first leaves the block open and the second starts before the subsequence ends, i.e. exactly when the one-symbol loop would
 * decode both: the hand-over states between subsequences do not change. */
#define MIJ_ES_PAIR_BITS 12u
#ifndef MIJ_ES_PAIR /* A/B switch: 0 = the write pass decodes one symbol per iteration */
#define MIJ_ES_PAIR 1
#endif
struct EsPair {
	uint16_t t2[2][1u << MIJ_ES_PAIR_BITS];
	uint8_t slot[8]; /* table index 0..7 -> 0 / 1 (its pair table) or 255 (none: a DC table, or a third AC table) */
};

/* Round 3, second session: the state-only passes look EVERY symbol up in one table chosen by data -- the block's AC table by the next
 * twelve bits of the stream, its DC table by the next nine -- with one entry format, so that a loop iteration is one LDS read, a few
 * field extractions and one window update whatever the 64 lanes of the wavefront are standing at.  Until then an iteration carried the
 * pair lookup, the nine-bit lookup of the ordinary path (the only one DC terms and long AC symbols had) and, whenever one lane of the
 * wave met a code longer than nine bits, the reference's fifty-instruction search.
 *   0                 not here (a code longer than the index, a non-code, a pair that may not be taken): es_state_slow
 *   bit 15 clear      one symbol:  bits [0:5) = code + extra bits (up to 27), [5:10) = positions it moves on, bit 10 = it ends the block
 *                     (EOB); a DC symbol moves one position
 *   bit 15 set        two AC symbols (EsPair's format): bits [0:4) = bits of both, [4:9) = positions of the first, [9:14) of the second,
 *                     bit 14 = the second is an EOB
 * (the first symbol of a pair is never an EOB, and a ZRL moves sixteen positions)
 * A single AC symbol only needs its CODE inside the twelve bits: what the extra bits are does not matter to the state. */
#define MIJ_ES_DC_BITS 9u
struct __attribute__((aligned(16))) EsUni { /* a multiple of 16 bytes: copied as uint4 (k_es_tables -> global -> LDS) */
	uint16_t ac[2][1u << MIJ_ES_PAIR_BITS];
	uint16_t dc[2][1u << MIJ_ES_DC_BITS];
	uint16_t zero[2];  /* entry 0: where the index of a table without entries is clamped to (MIJ_ES_UNI_ZERO) */
	uint32_t base[12]; /* block-in-MCU -> index (in uint16 units from ac[0][0]) of its AC table's first entry | its DC table's << 16; 0xffff: no entries */
};
#define MIJ_ES_UNI_ZERO ((2u << MIJ_ES_PAIR_BITS) + (2u << MIJ_ES_DC_BITS))

/* codec/jpeg.c:193-243: returns the symbol and its code length, or -1 */
__device__ __forceinline__ int es_symbol_e(const EsTab &h, uint64_t win, uint32_t e, uint32_t &len); /* the same with the fast-table entry already fetched */
__device__ __forceinline__ int es_symbol(const EsTab &h, uint64_t win, uint32_t &len)
{
	return es_symbol_e(h, win, h.fast16[(uint32_t)(win >> 55)], len);
}
__device__ __forceinline__ int es_symbol_e(const EsTab &h, uint64_t win, uint32_t e, uint32_t &len)
{
	const uint32_t top16 = (uint32_t)(win >> 48);
	if (e != 0xffffu) {
		len = e >> 8;
		return (int)(e & 255u);
	}
	if (MIJ_VARIANT & 2048) { /* ablation: what the search below costs (wrong symbols: timing only) */
		len = 10;
		return 0x11;
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
	/* a code is at most 16 bits long and brings at most 15 more: everything lies in the window's high dword */
	const uint32_t bits = ((uint32_t)(win >> 32) << len) >> (32u - n);
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
	uint64_t *meta;      /* per block of the scan: L1 of its de-quantised AC coefficients (low dword) | DC difference << 32 */
	const uint16_t *toff; /* zigzag index -> element offset inside the block's tile slot (int16 planes) */
	const uint8_t *zpos;  /* zigzag index -> in-block position P (the order of a block's escape bytes) */
	bool skip;            /* the block in progress was begun by the previous subsequence: k_es_tails stores its rest */
	bool stop_after_block;
	bool owner;          /* k_es_write: this thread stores the meta word of the blocks it begins (k_es_tails adds to the L1) */
	uint32_t ord;        /* ordinal of the current block */
	uint32_t mx, my;     /* its MCU */
	int16_t *blk;        /* its tile slot (int16 planes) */
	uint8_t *zz;         /* compact planes: its 64 bytes, zigzag order, in the intermediate image (blocks in scan order) */
	uint64_t grp;        /* the eight bytes of group curq gathered so far */
	uint32_t curq;
	bool esc;            /* the current block has an escaped coefficient (written by this thread) */
	int dcd;             /* DC difference of the current block */
	uint32_t acc;        /* L1 of the AC coefficients written by this thread into it */
	uint32_t *pfinal;    /* where the bit position after the scan's last block is recorded */
	/* int16 planes: the tile slot of block-in-MCU c of MCU (mx, my) */
	__device__ __forceinline__ void locate(uint32_t c)
	{
		const uint32_t g = loc->geo[c];
		const uint32_t bx = mx * (g & 255u) + ((g >> 16) & 255u), by = my * ((g >> 8) & 255u) + (g >> 24);
		const uint32_t L = bx + by * loc->bw[c];
		blk = reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(coef) + loc->plane[c]) + ((size_t)(L >> 6) << 12) + ((L & 63u) << 3);
	}
	/* compact planes: the 64 escape bytes of the current block (block-in-MCU c of MCU (mx, my)).  Until round 3 this derived the MCU from
	 * the block's ordinal with two integer divisions (ninety instructions, "rare") -- but a wavefront runs the path whenever ONE of its 64
	 * lanes meets an escaped coefficient, a quarter of the loop's iterations on the benchmark's pictures and nearly all on high-quality ones;
	 * counting the MCU along costs five instructions per block end */
	__device__ __forceinline__ uint8_t *escape_bytes(uint32_t c) const
	{
		const uint32_t g = loc->geo[c];
		const uint32_t bx = mx * (g & 255u) + ((g >> 16) & 255u), by = my * ((g >> 8) & 255u) + (g >> 24);
		return reinterpret_cast<uint8_t *>(coef) + loc->hi[c] + ((size_t)(bx + by * loc->bw[c]) << 6);
	}
	__device__ __forceinline__ void flush_group()
	{
		if (grp && !((MIJ_VARIANT & 256) && grp != 0x123456789abcdefull)) /* all-zero groups are the cleared image already; ablation bit 256: no group stores */
			*reinterpret_cast<uint64_t *>(zz + 8u * curq) = grp;
		grp = 0;
	}
	template <bool CB, bool BW>
	__device__ __forceinline__ void put(uint32_t k, int v, uint32_t c)
	{
		if (CB) {
			if (BW)
				zz[k] = (uint8_t)v;
			else {
				const uint32_t q = k >> 3;
				if (q != curq) {
					flush_group();
					curq = q;
				}
				grp |= (uint64_t)(uint8_t)v << (8u * (k & 7u)); /* low byte */
			}
			if ((uint32_t)(v + 128) > 255u) { /* escape: v == sext8(low) + 256 * h (mij_kernels.h, load_block_b8) */
				uint8_t *hi8 = escape_bytes(c);
				/* the block's first escape clears its 64 escape bytes (nothing else does); a tail looks at the flag
				 * the head may have left (an earlier kernel) */
				if (!esc && !(BW && zz[0] != 0)) {
					uint4 *h = reinterpret_cast<uint4 *>(hi8);
					h[0] = h[1] = h[2] = h[3] = make_uint4(0, 0, 0, 0);
				}
				esc = true;
				hi8[zpos[k]] = (uint8_t)((v + 128) >> 8);
			}
		} else
			blk[toff[k]] = (int16_t)v;
	}
	/* the block (or the head of one that continues in the next subsequence) is done: its bytes and its meta word.  Per-block
	 * L1 without atomics: whoever holds the block's start stores the word, the thread that finishes a block begun elsewhere
	 * (k_es_tails, a later launch) adds to its L1 half */
	template <bool CB, bool BW>
	__device__ __forceinline__ void end_block()
	{
		if (CB) {
			if (!BW)
				flush_group();
			if (esc)
				zz[0] = 1; /* the flags byte sits in the DC's place; behind this lane's own store of group 0 */
		}
		if (owner) {
			if (!((MIJ_VARIANT & 512) && acc != 0x12345678u)) /* ablation bit 512: no meta stores */
				meta[ord] = (uint64_t)acc | ((uint64_t)(uint16_t)(int16_t)dcd << 32);
		} else if (!CB)
			reinterpret_cast<uint32_t *>(meta + ord)[0] += acc;
		acc = 0;
		esc = false;
		curq = 0;
	}
};

/*
 * Decode from state s until the bit position reaches p_end (a symbol that starts before p_end is finished).
 * WRITE = false: only the state and the number of completed blocks.  WRITE = true: coefficients are stored,
 * decoding stops at block ordinal sc.nblocks, malformed input sets *anom.
 */
/* CB: compact planes (DevScan.fmt), BW: the tails pass's byte stores -- template parameters since round 3's second session: the two plane
 * formats and the two store forms were run-time branches inside put / end_block, dead code in every launch but in the way of the compiler */
template <bool WRITE, bool PAIRW = false, bool CB = true, bool BW = false>
__device__ __forceinline__ uint32_t es_decode(const DevScan &sc, const EsLocal &loc, const EsTab *__restrict__ tabs, const uint8_t *__restrict__ stream, EsState &s,
															 uint32_t p_end, EsWriter *wr, uint32_t *anom, const EsPair *__restrict__ pr = nullptr)
{
	static_assert(!(PAIRW && !WRITE), "PAIRW: the write pass's pair table");
	uint32_t done = 0;
	EsBits br;
	br.start(stream, s.p);
	uint32_t tb = loc.tabs[s.c]; /* the current block's component and tables; changes with s.c only */
	const uint16_t *t2cur = nullptr; /* the pair table of the block's AC table (state-only passes), off the per-symbol chain */
	if (PAIRW) {
		const uint32_t slot = pr->slot[tb >> 16];
		t2cur = slot < 2u ? pr->t2[slot] : nullptr;
	}
	/* ONE symbol per iteration whatever it is: the lanes of a wave sit at DC terms, AC runs and block ends all the time,
	 * so a loop with a DC branch and an AC branch executes both on almost every iteration (each at a fraction of the
	 * lanes).  Here the table is chosen by data and the state update is a handful of selects. */
	/* every iteration takes at least one bit (a code has a length, anything else ends the walk), so the position alone bounds the loop:
	 * p_end <= nbits, and the arena is zero padded behind the data for the window's look-ahead */
	while (s.p < p_end && s.z != MIJ_ES_DEAD) {
		if (WRITE && wr->ord >= sc.nblocks)
			break;
		const uint64_t win = br.win;
		const bool isdc = s.z == 0;
		/* both lookups leave together (one LDS round trip on the serial chain, not two): the ordinary fast-table entry of the
		 * table this symbol uses, and -- state-only passes -- the pair entry of the block's AC table */
		const EsTab &htab = tabs[isdc ? (tb >> 8) & 255u : tb >> 16];
		/* the write pass looks AC symbols up by twelve bits (EsPair: the fast table's format under a longer index), so that the search for
		 * long codes -- which a wavefront runs whenever one of its lanes needs it -- only remains for codes beyond twelve bits */
		const uint32_t e9 = (PAIRW && t2cur && !isdc) ? t2cur[(uint32_t)(win >> (64u - MIJ_ES_PAIR_BITS))] : htab.fast16[(uint32_t)(win >> 55)];
		{
		uint32_t len = 0;
		const int sym = es_symbol_e(htab, win, e9, len);
		/* DC: the reference takes categories up to 16; nothing a conforming stream uses beyond 11 */
		if (sym < 0 || len == 0 || (isdc && sym > 11)) {
			if (!WRITE) { /* a guessed start ran into a non-code: slip one bit and keep looking for the true sequence */
				s.p += 1;
				br.take(1);
				continue;
			}
			atomicOr(anom, 1u);
			s.z = MIJ_ES_DEAD;
			break;
		}
		const uint32_t n = isdc ? (uint32_t)sym : ((uint32_t)sym & 15u), r = isdc ? 0u : ((uint32_t)sym >> 4);
		const int ext = es_extend(win, len, n ? n : 1u);
		const int v = n ? ext : 0;
		s.p += len + n;
		/* the write pass shifts the window once per iteration: the window holds 32 valid bits at every symbol start, and a second symbol is
		 * only taken when it ends inside them */
		uint32_t used = len + n;
		if (!PAIRW)
			br.take(used);
		if (isdc) {
			if (WRITE)
				wr->dcd = v;
			s.z = 1;
		} else if (n == 0) {
			/* ZRL, or EOB (EOBn only exists in progressive scans; the reference treats it like EOB here, :355) */
			s.z = r == 15u ? s.z + 16u : 64u;
		} else {
			const uint32_t k = s.z + r;
			if (k > 63u) { /* the reference would write through its padded de-zigzag table: leave that to the host */
				if (WRITE)
					atomicOr(anom, 1u);
				s.z = 64;
			} else {
				if (WRITE && !wr->skip) {
					wr->template put<CB, BW>(k, v, s.c);
					if (!CB) { /* compact planes: k_es_pack has the whole block in registers and sums its L1 there */
						const int dq = (int)(int16_t)((uint32_t)v * loc.qz[tb & 255u][k]);
						wr->acc += (uint32_t)(dq < 0 ? -dq : dq);
					}
				}
				s.z = k + 1;
			}
		}
		/* Write pass: a second AC symbol in the same iteration, when the first symbol (a DC term or an AC symbol) left the block open and
		 * the second starts inside this subsequence -- the same symbol the next iteration would decode, from the same table, without its
		 * checks and its trip round the loop.  (Until round 3's second session the second symbol came out of a pair table and had to lie
		 * inside twelve bits together with the first.) */
		uint32_t e2 = 0xffffu;
		const uint64_t win2 = win << (used & 31u);
		if (PAIRW && t2cur && s.z < 64u && s.p < p_end && used <= 32u - MIJ_ES_PAIR_BITS)
			e2 = t2cur[(uint32_t)(win2 >> (64u - MIJ_ES_PAIR_BITS))];
		if (PAIRW && e2 != 0xffffu && used + (e2 >> 8) + (e2 & 15u) <= 32u) {
			const uint32_t len2 = e2 >> 8, n2 = e2 & 15u, r2 = (e2 >> 4) & 15u;
			const int v2 = n2 ? es_extend(win2, len2, n2) : 0;
			s.p += len2 + n2;
			used += len2 + n2;
			if (n2 == 0) {
				s.z = r2 == 15u ? s.z + 16u : 64u;
			} else {
				const uint32_t k2 = s.z + r2;
				if (k2 > 63u) {
					atomicOr(anom, 1u);
					s.z = 64;
				} else {
					if (!wr->skip) {
						wr->template put<CB, BW>(k2, v2, s.c);
						if (!CB) {
							const int dq = (int)(int16_t)((uint32_t)v2 * loc.qz[tb & 255u][k2]);
							wr->acc += (uint32_t)(dq < 0 ? -dq : dq);
						}
					}
					s.z = k2 + 1;
				}
			}
		}
		if (PAIRW)
			br.take(used);
		}
		if (s.z >= 64u) { /* block complete (ZRL past the end ends it too: same as the host loop's k < 64 test) */
			s.z = 0;
			++done;
			if (WRITE) {
				if (s.p > sc.nbits)
					atomicOr(anom, 16u); /* the data ran out inside this block: the reference decodes on with zero bits */
				if (!wr->skip)
					wr->template end_block<CB, BW>();
				wr->acc = 0;
				if (wr->stop_after_block) { /* k_es_tails: only the rest of the block the subsequence started in */
					s.z = 0;
					break;
				}
				wr->skip = false;
				if (++wr->ord == sc.nblocks)
					*wr->pfinal = s.p;
				wr->zz += 64;
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
			if (PAIRW) {
				const uint32_t slot = pr->slot[tb >> 16];
				t2cur = slot < 2u ? pr->t2[slot] : nullptr;
			}
			if (WRITE && !CB && wr->ord < sc.nblocks)
				wr->locate(s.c);
		}
	}
	if (WRITE && wr->owner && wr->ord < sc.nblocks && !wr->skip && s.z != 0 && s.z != MIJ_ES_DEAD)
		wr->template end_block<CB, BW>(); /* a block that continues in the next subsequence: its bytes so far and the meta word of its head */
	return done;
}

/* The write pass's tables (record form).  Entries in the fast table's format (code length << 8 | symbol, 0 = not here) under a twelve-bit
 * index for the scan's (at most two) AC tables and a nine-bit index for its DC tables; an iteration looks the symbol its window starts with up
 * and a second AC symbol behind it in the same table.  The search for codes beyond the index runs on EsLong.  Built once per picture by
 * k_es_tables, copied by every workgroup of k_es_writer: 21 KiB, which leaves room for the record buffers (below) at four workgroups a CU. */
struct __attribute__((aligned(16))) EsLong { /* what the search for a code of ten bits or more needs of a table (EsTab without its fast table) */
	uint8_t values[256];
	uint32_t maxcode[8];
	int32_t delta[8];
	uint32_t maxlo[12]; /* lengths 1..9 of stbi__huffman.maxcode / .delta: only a table WITHOUT window entries (a third pair) is searched from length 1 */
	int32_t dello[12];
};
struct __attribute__((aligned(16))) EsW {
	uint16_t ac[2][1u << MIJ_ES_PAIR_BITS];
	uint16_t dc[2][1u << MIJ_ES_DC_BITS];
	uint16_t zero[8];  /* right behind dc[][]: entry 0 = where the index of a table without entries is clamped to (MIJ_ES_W_ZERO) */
	EsLong lng[8];     /* the scan's eight tables, for the codes beyond the windows (in LDS: a wavefront meets one in a third of its iterations,
	                    * and a search over tables in global memory made the pass wait 79 % of its time) */
	uint32_t base[12]; /* block-in-MCU -> index of its AC table's first entry | its DC table's << 16; 0xffff: no entries */
	uint32_t tabs[12]; /* EsLocal.tabs: component | DC table << 8 | AC table << 16 (the search) */
};
#define MIJ_ES_W_ZERO ((2u << MIJ_ES_PAIR_BITS) + (2u << MIJ_ES_DC_BITS))

/* codec/jpeg.c:219-243 for a code of ten bits or more (es_symbol_e's search on EsLong): returns the symbol and its code length, or -1 */
__device__ __forceinline__ int es_symbol_long(const EsLong &h, uint64_t win, uint32_t &len)
{
	const uint32_t top16 = (uint32_t)(win >> 48);
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

/* a table without window entries: every length (the reference's loop from 1: its fast table is a shortcut of the same search) */
__device__ __forceinline__ int es_symbol_full(const EsLong &h, uint64_t win, uint32_t &len)
{
	const uint32_t top16 = (uint32_t)(win >> 48);
	for (uint32_t l = 1; l < 10u; ++l)
		if (top16 < h.maxlo[l]) {
			const int c = (int)(top16 >> (16u - l)) + h.dello[l];
			if (c < 0 || c > 255)
				return -1;
			len = l;
			return h.values[c];
		}
	return es_symbol_long(h, win, len);
}

__device__ __forceinline__ void es_build_w(const DevScan &sc, const EsTab *l, const DevHuff *__restrict__ g, EsW *w)
{
	uint32_t ua[2] = {255u, 255u}, ud[2] = {255u, 255u}, na = 0, nd = 0;
	for (uint32_t c = 0; c < 4u; ++c) {
		const uint32_t ta = sc.ac_tab[c] & 7u, td = sc.dc_tab[c] & 7u;
		if (ta != ua[0] && ta != ua[1] && na < 2u)
			ua[na++] = ta;
		if (td != ud[0] && td != ud[1] && nd < 2u)
			ud[nd++] = td;
	}
	if (threadIdx.x < 12u) {
		const uint32_t ci = sc.blk_comp[threadIdx.x] & 3u, ta = sc.ac_tab[ci] & 7u, td = sc.dc_tab[ci] & 7u;
		const uint32_t sa = ta == ua[0] ? 0u : (ta == ua[1] ? 1u : 2u), sd = td == ud[0] ? 0u : (td == ud[1] ? 1u : 2u);
		w->base[threadIdx.x] = (sa < 2u ? sa << MIJ_ES_PAIR_BITS : 0xffffu) | (sd < 2u ? (2u << MIJ_ES_PAIR_BITS) + (sd << MIJ_ES_DC_BITS) : 0xffffu) << 16;
		w->tabs[threadIdx.x] = ci | td << 8 | ta << 16;
	}
	if (threadIdx.x < 8u)
		w->zero[threadIdx.x] = 0;
	for (uint32_t i = threadIdx.x; i < 8u * 256u; i += blockDim.x)
		w->lng[i >> 8].values[i & 255u] = l[i >> 8].values[i & 255u];
	if (threadIdx.x < 64u) {
		w->lng[threadIdx.x >> 3].maxcode[threadIdx.x & 7u] = l[threadIdx.x >> 3].maxcode[threadIdx.x & 7u];
		w->lng[threadIdx.x >> 3].delta[threadIdx.x & 7u] = l[threadIdx.x >> 3].delta[threadIdx.x & 7u];
	}
	if (threadIdx.x < 96u) {
		const uint32_t t = threadIdx.x / 12u, j = threadIdx.x % 12u;
		w->lng[t].maxlo[j] = j < 10u ? g[t].maxcode[j] : 0xffffffffu;
		w->lng[t].dello[j] = j < 10u ? g[t].delta[j] : 0;
	}
	for (uint32_t i = threadIdx.x; i < 2u << MIJ_ES_PAIR_BITS; i += blockDim.x) {
		const uint32_t k = i >> MIJ_ES_PAIR_BITS, wd = i & ((1u << MIJ_ES_PAIR_BITS) - 1u);
		uint32_t e = 0;
		if (ua[k] != 255u) {
			uint32_t len1 = 0;
			const int s1 = es_symbol(l[ua[k]], (uint64_t)wd << (64u - MIJ_ES_PAIR_BITS), len1);
			if (s1 >= 0 && len1 && len1 <= MIJ_ES_PAIR_BITS)
				e = len1 << 8 | ((uint32_t)s1 & 255u);
		}
		w->ac[k][wd] = (uint16_t)e;
	}
	for (uint32_t i = threadIdx.x; i < 2u << MIJ_ES_DC_BITS; i += blockDim.x) {
		const uint32_t k = i >> MIJ_ES_DC_BITS, wd = i & ((1u << MIJ_ES_DC_BITS) - 1u);
		uint32_t e = 0;
		if (ud[k] != 255u) {
			uint32_t len = 0;
			const int sy = es_symbol(l[ud[k]], (uint64_t)wd << (64u - MIJ_ES_DC_BITS), len);
			if (sy >= 0 && len && len <= MIJ_ES_DC_BITS && sy <= 11)
				e = len << 8 | (uint32_t)sy;
		}
		w->dc[k][wd] = (uint16_t)e;
	}
	__syncthreads();
}

/* ------------------------------------------------------------------ the write pass as a record stream (compact planes; round 3, second session)
 *
 * The write pass above places every coefficient where it belongs in a 64-byte image of its block: group bookkeeping, escape bytes, a
 * partial first block left to k_es_tails, block-end stores -- an iteration of its loop is 750 instructions, a third of them scalar mask
 * bookkeeping of nested divergent branches.  This form only says WHAT it decoded: one 16-bit record per DC term and per non-zero AC
 * coefficient, appended to the subsequence's own region of a record arena (four records per 8-byte store); k_es_pack2 -- one lane per
 * block, dense -- follows a block's records and places the bytes.  A block's records start where the lane that decoded its DC term says
 * (meta[ordinal] = record index) and end at the next DC record; a subsequence always ends its region with a jump to the next one (three
 * records) or, behind the scan's last block or after an anomaly, an end record -- so a block that straddles subsequences needs no second
 * pass (k_es_tails) and the intermediate image no clearing.
 *   0 e kkkkkk llllllll   AC coefficient: zigzag index k, low byte l; e: an escape record with the high byte follows
 *   1000 dddddddddddd     DC difference (12 bits, two's complement): the first record of a block
 *   1001 ........         end of the scan's records         1010: jump, the next two records = record index (low, high 16 bits)
 *   1011 ....hhhhhhhh     escape: high byte h of the coefficient before        1111: padding
 * Records per bit of stream: at most one per two bits (every record is a Huffman symbol of at least one code bit and, for a
 * coefficient, at least one more; a DC term shares its block's two bits with the EOB that ends it), so a region of
 * sub_bits + MIJ_ES_REC_SLACK bytes holds whatever a subsequence decodes. */
#ifndef MIJ_ES_RECORDS
#define MIJ_ES_RECORDS 1
#endif
#define MIJ_ES_REC_SLACK 64u
#define MIJ_ES_REC_DC 0x8000u
#define MIJ_ES_REC_END 0x9000u
#define MIJ_ES_REC_JUMP 0xa000u
#define MIJ_ES_REC_ESC 0xb000u
#define MIJ_ES_REC_PAD 0xf000u

/* Why the records do not go out as they come.  gfx950 counts loads and stores in ONE counter (vmcnt), so the wait for the stream word that
 * refills a lane's window also waits for every store issued before it -- and with 64 lanes emitting at their own pace there is a store in
 * flight in every iteration: without its stores this pass takes 0.70 ms per 256 pictures, with them 1.31.  So a lane gathers its records in
 * LDS (dword j of all lanes side by side: every lane its own bank), and every MIJ_ES_REC_EVERY iterations the whole wavefront stores its
 * complete 8-byte words together: one iteration in MIJ_ES_REC_EVERY has stores to wait behind. */
#define MIJ_ES_REC_BUF 32u  /* records of LDS per lane */
#define MIJ_ES_REC_EVERY 7u /* an iteration emits at most four records: 7 x 4 + 3 left over < 32 */
struct EsRecOut {
	uint64_t *slot;  /* the next 8 bytes of the region */
	uint32_t *buf;   /* this lane's column of the workgroup's record buffer: dword j at buf[j * MIJ_ES_WG] */
	uint32_t cnt;    /* records in it */
	uint32_t index;  /* record index (in the arena) of the record at position 0 of the buffer */
	__device__ __forceinline__ void emit(uint32_t rec)
	{
		reinterpret_cast<uint16_t *>(buf + (cnt >> 1) * MIJ_ES_WG)[cnt & 1u] = (uint16_t)rec;
		++cnt;
	}
	/* the complete 8-byte words go out, the (at most three) records behind them move to the front */
	__device__ __forceinline__ void flush()
	{
		const uint32_t words = cnt >> 2;
#pragma unroll
		for (uint32_t j = 0; j < MIJ_ES_REC_BUF / 4u; ++j)
			if (j < words) {
				const uint64_t v = (uint64_t)buf[(2u * j) * MIJ_ES_WG] | (uint64_t)buf[(2u * j + 1u) * MIJ_ES_WG] << 32;
				if (!((MIJ_VARIANT & 4096) && v != 0x123456789abcdefull)) /* ablation bit 4096: no record stores */
					slot[j] = v;
			}
		const uint32_t r0 = buf[(2u * words) * MIJ_ES_WG], r1 = buf[(2u * words + 1u) * MIJ_ES_WG]; /* words <= 7: inside the buffer */
		buf[0] = r0;
		buf[MIJ_ES_WG] = r1;
		slot += words;
		index += 4u * words;
		cnt &= 3u;
	}
	__device__ __forceinline__ void finish()
	{
		while (cnt & 3u)
			emit(MIJ_ES_REC_PAD);
		flush();
	}
};

/* es_decode<true> for compact planes in record form: same symbols, same anomalies, same hand-over rule */
__device__ __forceinline__ void es_write_records(const DevScan &sc, const EsW &w, const uint8_t *__restrict__ stream, EsState &s, uint32_t p_end, uint32_t &ord,
																 EsRecOut &out, uint32_t *__restrict__ meta_idx, uint32_t *__restrict__ anom, uint32_t *__restrict__ pfinal)
{
	EsBits br;
	br.start(stream, s.p);
	const uint16_t *__restrict__ w16 = &w.ac[0][0]; /* ac[2][4096], dc[2][512], zero[] */
	const uint32_t bpm = sc.bpm, nblocks = sc.nblocks, nbits = sc.nbits;
	uint32_t base = w.base[s.c], it = 0;
	while (s.p < p_end && s.z != MIJ_ES_DEAD && ord < nblocks) {
		const uint64_t win = br.win;
		const uint32_t hi = (uint32_t)(win >> 32);
		const bool isdc = s.z == 0;
		const uint32_t idx = min(isdc ? (base >> 16) + (hi >> (32u - MIJ_ES_DC_BITS)) : (base & 0xffffu) + (hi >> (32u - MIJ_ES_PAIR_BITS)), MIJ_ES_W_ZERO);
		uint32_t e = w16[idx];
		if (e == 0u) { /* a code beyond the index, a third table, or no code at all: the reference's search */
			const uint32_t tb = w.tabs[s.c];
			const EsLong &hl = w.lng[isdc ? (tb >> 8) & 255u : tb >> 16];
			uint32_t len = 0;
			const int sym = (isdc ? base >> 16 : base & 0xffffu) != 0xffffu ? es_symbol_long(hl, win, len) : es_symbol_full(hl, win, len);
			if (sym < 0 || len == 0 || (isdc && sym > 11)) {
				atomicOr(anom, 1u);
				s.z = MIJ_ES_DEAD;
				break;
			}
			e = len << 8 | ((uint32_t)sym & 255u);
		}
		const uint32_t len1 = e >> 8, n1 = isdc ? (e & 255u) : (e & 15u), r1 = isdc ? 0u : ((e >> 4) & 15u);
		const int v1 = n1 ? es_extend(win, len1, n1) : 0;
		uint32_t used = len1 + n1;
		s.p += used;
		if (isdc) {
			meta_idx[2u * ord] = out.index + out.cnt; /* low dword of the block's meta word: where its records start */
			out.emit(MIJ_ES_REC_DC | ((uint32_t)v1 & 0xfffu));
			s.z = 1;
		} else if (n1 == 0u) {
			s.z = r1 == 15u ? s.z + 16u : 64u; /* ZRL, or EOB */
		} else {
			const uint32_t k = s.z + r1;
			if (k > 63u) { /* the reference would write through its padded de-zigzag table: leave that to the host */
				atomicOr(anom, 1u);
				s.z = 64;
			} else {
				const bool esc = (uint32_t)(v1 + 128) > 255u;
				out.emit(k << 8 | ((uint32_t)v1 & 255u) | (esc ? 0x4000u : 0u));
				if (esc)
					out.emit(MIJ_ES_REC_ESC | (((uint32_t)(v1 + 128) >> 8) & 255u));
				s.z = k + 1;
			}
		}
		/* a second AC symbol in the same iteration, from the same table, when the first symbol left the block open, the second starts inside
		 * this subsequence and ends inside the window's 32 valid bits (es_decode's rule: the hand-over states do not change) */
		uint32_t e2 = 0;
		const uint64_t win2 = win << (used & 31u);
		if ((base & 0xffffu) != 0xffffu && s.z < 64u && s.p < p_end && used <= 32u - MIJ_ES_PAIR_BITS)
			e2 = w16[(base & 0xffffu) + (uint32_t)(win2 >> (64u - MIJ_ES_PAIR_BITS))];
		if (e2 != 0u && used + (e2 >> 8) + (e2 & 15u) <= 32u) {
			const uint32_t len2 = e2 >> 8, n2 = e2 & 15u, r2 = (e2 >> 4) & 15u;
			const int v2 = n2 ? es_extend(win2, len2, n2) : 0;
			s.p += len2 + n2;
			used += len2 + n2;
			if (n2 == 0u) {
				s.z = r2 == 15u ? s.z + 16u : 64u;
			} else {
				const uint32_t k2 = s.z + r2;
				if (k2 > 63u) {
					atomicOr(anom, 1u);
					s.z = 64;
				} else {
					const bool esc2 = (uint32_t)(v2 + 128) > 255u;
					out.emit(k2 << 8 | ((uint32_t)v2 & 255u) | (esc2 ? 0x4000u : 0u));
					if (esc2)
						out.emit(MIJ_ES_REC_ESC | (((uint32_t)(v2 + 128) >> 8) & 255u));
					s.z = k2 + 1;
				}
			}
		}
		br.take(used);
		if (s.z >= 64u) { /* block complete */
			s.z = 0;
			if (s.p > nbits)
				atomicOr(anom, 16u); /* the data ran out inside this block: the reference decodes on with zero bits */
			if (++ord == nblocks)
				*pfinal = s.p;
			if (++s.c == bpm)
				s.c = 0;
			base = w.base[s.c];
		}
		if (++it == MIJ_ES_REC_EVERY) { /* the same iteration for every lane still in the loop: they entered it together */
			out.flush();
			it = 0;
		}
	}
	out.flush();
}

/* The state-only passes' walk (EsUni): from state s until the bit position reaches p_end (a symbol that starts before p_end is finished, a pair
 * is only taken when both of its symbols END by p_end -- otherwise the first goes alone and the loop decides about the second: what
 * es_decode<true> does with its pairs, so the hand-over states agree).  Returns the number of blocks completed.  On a true state of a
 * stream the write pass accepts, every symbol moves the state exactly as es_decode<true> moves it; on guessed states any deterministic
 * walk will do, and this one slips a bit at a non-code like the ordinary path. */
__device__ __forceinline__ uint32_t es_state_walk(const DevScan &sc, const EsLocal &loc, const EsTab *__restrict__ tabs, const EsUni &un, const uint8_t *__restrict__ stream,
																  EsState &s, uint32_t p_end)
{
	uint32_t done = 0;
	const uint32_t bpm = sc.bpm;
	const uint16_t *__restrict__ u16 = &un.ac[0][0]; /* ac[2][4096], then dc[2][512] */
	EsBits br;
	br.start(stream, s.p);
	uint32_t base = un.base[s.c];
	/* every iteration takes at least one bit (an entry holds a whole symbol, the search a code or a slipped bit), so the position alone
	 * ends the loop: no iteration guard, and no state here is ever MIJ_ES_DEAD (only the write pass makes that one) */
	while (s.p < p_end) {
		const uint32_t hi = (uint32_t)(br.win >> 32);
		const bool isdc = s.z == 0;
		/* a table without entries has base 0xffff: the index lands beyond the tables and is clamped to the zero entry */
		const uint32_t idx = min(isdc ? (base >> 16) + (hi >> (32u - MIJ_ES_DC_BITS)) : (base & 0xffffu) + (hi >> (32u - MIJ_ES_PAIR_BITS)), MIJ_ES_UNI_ZERO);
		const uint32_t e = u16[idx];
		/* everything by selects (bitwise on the comparison results: no short-circuit, no divergent branches): the lanes of a wave sit in
		 * both entry formats all the time.  one symbol: bits [0:5), positions [5:10), EOB bit 10; two: bits [0:4), positions [4:9) + [9:14), EOB bit 14 */
		const uint32_t pair = e >> 15;
		const uint32_t a1 = (e >> 4) & 31u;
		const uint32_t bits_e = e & (pair ? 15u : 31u);
		const uint32_t adv = pair ? a1 + ((e >> 9) & 31u) : (e >> 5) & 31u;
		const uint32_t ends = e & (pair ? 0x4000u : 0x400u);
		const uint32_t ok = pair ? (uint32_t)(s.z + a1 < 64u) & (uint32_t)(s.p + bits_e <= p_end) : (uint32_t)(e != 0u);
		uint32_t bits = bits_e, znew = ends ? 64u : s.z + adv;
		if (!ok) { /* a code longer than the index, a table without entries, a pair that may not be taken, a non-code: one symbol by the reference's search */
			const uint32_t tb = loc.tabs[s.c];
			const EsTab &htab = tabs[isdc ? (tb >> 8) & 255u : tb >> 16];
			uint32_t len = 0;
			const int sym = es_symbol(htab, br.win, len);
			if (sym < 0 || len == 0 || (isdc && sym > 11)) { /* a guessed start ran into a non-code: slip one bit and keep looking for the true sequence */
				bits = 1;
				znew = s.z;
			} else {
				const uint32_t n = isdc ? (uint32_t)sym : ((uint32_t)sym & 15u), r = (uint32_t)sym >> 4;
				bits = len + n;
				znew = isdc ? 1u : (n ? s.z + r + 1u : (r == 15u ? s.z + 16u : 64u)); /* a run past coefficient 63 ends the block like the ordinary path's k > 63 */
			}
		}
		s.p += bits;
		br.take(bits);
		s.z = znew;
		if (s.z >= 64u) { /* block complete */
			s.z = 0;
			++done;
			if (++s.c == bpm)
				s.c = 0;
			base = un.base[s.c];
		}
	}
	return done;
}

/* every kernel below: grid.x = blocks of MIJ_ES_WG subsequences over a (scan, first subsequence) work list */
struct EsWork {
	uint32_t scan, first;
};

/* the scan's eight Huffman tables and its EsLocal into LDS (im == nullptr: no block placement needed); ends in a barrier */
/* pr != nullptr: also the write pass's twelve-bit tables of its (at most two) AC tables (EsPair) */
__device__ __forceinline__ void es_build_uni(const DevScan &sc, const EsTab *l, EsUni *un);
/* un: the state-only passes' entry tables, copied from uni_src (what k_es_tables built for the picture's table set) or built here */
__device__ __forceinline__ void es_load_tables(const DevScan &sc, const DevImage *im, const DevHuff *__restrict__ g, EsTab *l, EsLocal *loc, EsPair *pr = nullptr,
																bool for_write = false, EsUni *un = nullptr, const uint4 *__restrict__ uni_src = nullptr)
{
	if (un && uni_src) {
		uint4 *dst = reinterpret_cast<uint4 *>(un);
		for (uint32_t i = threadIdx.x; i < sizeof(EsUni) / 16u; i += blockDim.x)
			dst[i] = uni_src[i];
	}
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
	if (threadIdx.x < 256u)
		loc->qz[threadIdx.x >> 6][threadIdx.x & 63u] = sc.qz[threadIdx.x >> 6][threadIdx.x & 63u];
	__syncthreads();
	if (pr) {
		/* the (at most two) AC tables of the scan's components, in order of first use; a third one keeps the ordinary path */
		uint32_t used[2] = {255u, 255u}, nused = 0;
		for (uint32_t c = 0; c < 4u; ++c) {
			const uint32_t t = sc.ac_tab[c] & 7u;
			if (t != used[0] && t != used[1] && nused < 2u)
				used[nused++] = t;
		}
		if (threadIdx.x < 8u)
			pr->slot[threadIdx.x] = threadIdx.x == used[0] ? 0 : (threadIdx.x == used[1] ? 1 : 255);
		for (uint32_t i = threadIdx.x; i < 2u << MIJ_ES_PAIR_BITS; i += blockDim.x) {
			const uint32_t k = i >> MIJ_ES_PAIR_BITS, w = i & ((1u << MIJ_ES_PAIR_BITS) - 1u);
			/* the fast table's entry format (code length << 8 | symbol, 0xffff = not here) under a twelve-bit index: full decode (long codes
			 * included) of the window padded with zeros; a symbol counts only if its code lies inside the bits the window really holds */
			uint32_t e = 0xffffu;
			if (used[k] != 255u) {
				const EsTab &h = l[used[k]];
				uint32_t len1 = 0;
				const int s1 = es_symbol(h, (uint64_t)w << (64u - MIJ_ES_PAIR_BITS), len1);
				if (s1 >= 0 && len1 && len1 <= MIJ_ES_PAIR_BITS)
					e = len1 << 8 | ((uint32_t)s1 & 255u);
			}
			pr->t2[k][w] = (uint16_t)e;
		}
		__syncthreads();
	}
	if (un && !uni_src)
		es_build_uni(sc, l, un);
}

/* EsUni from the tables already in LDS (after es_load_tables' barrier); ends in a barrier */
__device__ __forceinline__ void es_build_uni(const DevScan &sc, const EsTab *l, EsUni *un)
{
	/* the (at most two) AC and DC tables of the scan's components, in order of first use; a third one has no entry table (es_state_slow) */
	uint32_t ua[2] = {255u, 255u}, ud[2] = {255u, 255u}, na = 0, nd = 0;
	for (uint32_t c = 0; c < 4u; ++c) {
		const uint32_t ta = sc.ac_tab[c] & 7u, td = sc.dc_tab[c] & 7u;
		if (ta != ua[0] && ta != ua[1] && na < 2u)
			ua[na++] = ta;
		if (td != ud[0] && td != ud[1] && nd < 2u)
			ud[nd++] = td;
	}
	if (threadIdx.x < 12u) {
		const uint32_t ci = sc.blk_comp[threadIdx.x] & 3u, ta = sc.ac_tab[ci] & 7u, td = sc.dc_tab[ci] & 7u;
		const uint32_t sa = ta == ua[0] ? 0u : (ta == ua[1] ? 1u : 2u), sd = td == ud[0] ? 0u : (td == ud[1] ? 1u : 2u);
		un->base[threadIdx.x] = (sa < 2u ? sa << MIJ_ES_PAIR_BITS : 0xffffu) | (sd < 2u ? (2u << MIJ_ES_PAIR_BITS) + (sd << MIJ_ES_DC_BITS) : 0xffffu) << 16;
	}
	if (threadIdx.x < 2u)
		un->zero[threadIdx.x] = 0;
	for (uint32_t i = threadIdx.x; i < 2u << MIJ_ES_PAIR_BITS; i += blockDim.x) {
		const uint32_t k = i >> MIJ_ES_PAIR_BITS, w = i & ((1u << MIJ_ES_PAIR_BITS) - 1u);
		uint32_t e = 0;
		if (ua[k] != 255u) {
			const EsTab &h = l[ua[k]];
			uint32_t len1 = 0;
			const int s1 = es_symbol(h, (uint64_t)w << (64u - MIJ_ES_PAIR_BITS), len1);
			const uint32_t n1 = (uint32_t)s1 & 15u, r1 = ((uint32_t)s1 >> 4) & 15u, bits1 = len1 + n1;
			if (s1 >= 0 && len1 && len1 <= MIJ_ES_PAIR_BITS) { /* the code lies inside the index: its extra bits may reach past it */
				const uint32_t eob1 = (!n1 && r1 != 15u) ? 1u : 0u, adv1 = eob1 ? 0u : (n1 ? r1 + 1u : 16u);
				e = bits1 | adv1 << 5 | eob1 << 10;
				if (!eob1 && bits1 < MIJ_ES_PAIR_BITS) {
					const uint32_t rem = MIJ_ES_PAIR_BITS - bits1;
					uint32_t len2 = 0;
					const int s2 = es_symbol(h, (uint64_t)w << (64u - MIJ_ES_PAIR_BITS + bits1), len2);
					const uint32_t n2 = (uint32_t)s2 & 15u, r2 = ((uint32_t)s2 >> 4) & 15u;
					if (s2 >= 0 && len2 && len2 + n2 <= rem) {
						const uint32_t eob2 = (!n2 && r2 != 15u) ? 1u : 0u, adv2 = eob2 ? 0u : (n2 ? r2 + 1u : 16u);
						e = 0x8000u | (bits1 + len2 + n2) | adv1 << 4 | adv2 << 9 | eob2 << 14;
					}
				}
			}
		}
		un->ac[k][w] = (uint16_t)e;
	}
	for (uint32_t i = threadIdx.x; i < 2u << MIJ_ES_DC_BITS; i += blockDim.x) {
		const uint32_t k = i >> MIJ_ES_DC_BITS, w = i & ((1u << MIJ_ES_DC_BITS) - 1u);
		uint32_t e = 0;
		if (ud[k] != 255u) {
			const EsTab &h = l[ud[k]];
			uint32_t len = 0;
			const int sy = es_symbol(h, (uint64_t)w << (64u - MIJ_ES_DC_BITS), len);
			/* DC: the reference takes categories up to 16; nothing a conforming stream uses beyond 11 (es_state_slow slips a bit there, like the ordinary path) */
			if (sy >= 0 && len && len <= MIJ_ES_DC_BITS && sy <= 11)
				e = (len + (uint32_t)sy) | 1u << 5;
		}
		un->dc[k][w] = (uint16_t)e;
	}
	__syncthreads();
}

/* The entry tables of the state-only passes, once per picture (= per set of eight Huffman tables; tabscan[t] = a scan that uses set t):
 * every workgroup of those passes used to build them for itself -- sixteen thousand table decodes per workgroup, four workgroups per 1080p
 * picture, three passes -- and now copies 18 KiB.  One workgroup per table set. */
__global__ __launch_bounds__(256) void k_es_tables(const DevScan *__restrict__ scans, const uint32_t *__restrict__ tabscan, const DevHuff *__restrict__ huff,
																	uint4 *__restrict__ unis, uint4 *__restrict__ wtabs)
{
	__shared__ EsTab tabs[8];
	__shared__ EsLocal loc;
	__shared__ EsUni uni;
	__shared__ EsW w;
	const DevScan &sc = scans[tabscan[blockIdx.x]];
	es_load_tables(sc, nullptr, huff + sc.tab_off, tabs, &loc, nullptr, false, &uni);
	es_build_w(sc, tabs, huff + sc.tab_off, &w);
	uint4 *dst = unis + (size_t)(sc.tab_off >> 3) * (sizeof(EsUni) / 16u);
	const uint4 *src = reinterpret_cast<const uint4 *>(&uni);
	for (uint32_t i = threadIdx.x; i < sizeof(EsUni) / 16u; i += blockDim.x)
		dst[i] = src[i];
	dst = wtabs + (size_t)(sc.tab_off >> 3) * (sizeof(EsW) / 16u);
	src = reinterpret_cast<const uint4 *>(&w);
	for (uint32_t i = threadIdx.x; i < sizeof(EsW) / 16u; i += blockDim.x)
		dst[i] = src[i];
}

MIJ_ES_KERNEL void k_es_cold(const DevScan *__restrict__ scans, const EsWork *__restrict__ work, const DevHuff *__restrict__ huff,
																 const uint8_t *__restrict__ streams, uint64_t *__restrict__ start, uint64_t *__restrict__ end, uint32_t *__restrict__ cnt,
																 const uint4 *__restrict__ unis)
{
	__shared__ EsTab tabs[8];
	__shared__ EsLocal loc;
	__shared__ EsUni uni;
	const EsWork wk = work[blockIdx.x];
	const DevScan &sc = scans[wk.scan];
	es_load_tables(sc, nullptr, huff + sc.tab_off, tabs, &loc, nullptr, false, &uni, unis + (size_t)(sc.tab_off >> 3) * (sizeof(EsUni) / 16u));
	const uint32_t i = wk.first + threadIdx.x;
	if (i >= sc.nsub)
		return;
	EsState s;
	s.p = i * sc.sub_bits;
	s.z = 0;
	s.c = 0;
	start[sc.sub_off + i] = es_pack(s);
	const uint32_t pe = min((i + 1u) * sc.sub_bits, sc.nbits);
	cnt[sc.sub_off + i] = es_state_walk(sc, loc, tabs, uni, streams + sc.stream_off, s, pe);
	end[sc.sub_off + i] = es_pack(s);
}

/* A subsequence whose end state moved hands that state to its successor for the NEXT round: entries (subsequence, start state) appended to
 * the scan's queue, one atomic per wavefront.  Only the lanes still running call this (the others have returned). */
__device__ __forceinline__ void es_push(bool push, uint32_t j, uint64_t state, uint32_t sub_off, uint32_t *__restrict__ qcount, uint32_t *__restrict__ qidx,
													 uint64_t *__restrict__ qstate)
{
	const uint64_t m = __ballot(push);
	if (!m)
		return;
	const uint32_t lane = __lane_id(), leader = (uint32_t)__builtin_ctzll(m);
	uint32_t base = 0;
	if (lane == leader)
		base = atomicAdd(qcount, (uint32_t)__builtin_popcountll(m));
	base = __shfl(base, (int)leader);
	if (push) {
		const uint32_t pos = sub_off + base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
		qidx[pos] = j;
		qstate[pos] = state;
	}
}

/* The first synchronisation round: every subsequence behind the first restarts from the end state the cold pass found for its predecessor
 * (end_in; practically all of them: a guessed start is a block start at the first bit).  end_out = this round's end states; a subsequence
 * whose end state is no longer the cold pass's puts its successor on the queue of the second round.  pending[scan] counts those. */
MIJ_ES_KERNEL void k_es_sync(const DevScan *__restrict__ scans, const EsWork *__restrict__ work, const DevHuff *__restrict__ huff,
																 const uint8_t *__restrict__ streams, uint64_t *__restrict__ start, const uint64_t *__restrict__ end_in,
																 uint64_t *__restrict__ end_out, uint32_t *__restrict__ cnt, uint32_t *__restrict__ pending, uint32_t *__restrict__ qidx,
																 uint64_t *__restrict__ qstate, const uint4 *__restrict__ unis)
{
	__shared__ EsTab tabs[8];
	__shared__ EsLocal loc;
	__shared__ EsUni uni;
	const EsWork wk = work[blockIdx.x];
	const DevScan &sc = scans[wk.scan];
	const uint32_t i = wk.first + threadIdx.x;
	const uint32_t slot = sc.sub_off + i;
	uint64_t want = 0, was = 0;
	bool redo = false;
	if (i < sc.nsub) {
		was = end_in[slot];
		if (i > 0) {
			want = end_in[slot - 1];
			redo = want != start[slot];
		}
		if (!redo)
			end_out[slot] = was;
	}
	if (!__syncthreads_or(redo ? 1 : 0))
		return;
	es_load_tables(sc, nullptr, huff + sc.tab_off, tabs, &loc, nullptr, false, &uni, unis + (size_t)(sc.tab_off >> 3) * (sizeof(EsUni) / 16u));
	if (!redo)
		return;
	start[slot] = want;
	EsState s = es_unpack(want);
	const uint32_t pe = min((i + 1u) * sc.sub_bits, sc.nbits);
	cnt[slot] = es_state_walk(sc, loc, tabs, uni, streams + sc.stream_off, s, pe);
	const uint64_t now = es_pack(s);
	end_out[slot] = now;
	es_push(i + 1u < sc.nsub && now != was, i + 1u, now, sc.sub_off, &pending[wk.scan], qidx, qstate);
}

/* Every later round: only the subsequences on the scan's queue run -- gathered into full wavefronts, where the first round's form ran a whole
 * wavefront for one lane that moved (round 2 of the benchmark's pictures: 55 % of a full round's instructions and 84 % of its time for a
 * small share of the subsequences).  n_in[scan] entries (subsequence, start state) written by the round before; end states are updated in
 * place (nobody else reads them: a successor gets its state through the queue), the successors of those that moved go on the next queue. */
MIJ_ES_KERNEL void k_es_syncq(const DevScan *__restrict__ scans, const EsWork *__restrict__ work, const DevHuff *__restrict__ huff,
																  const uint8_t *__restrict__ streams, uint64_t *__restrict__ start, uint64_t *__restrict__ end, uint32_t *__restrict__ cnt,
																  const uint32_t *__restrict__ n_in, const uint32_t *__restrict__ qidx_in, const uint64_t *__restrict__ qstate_in,
																  uint32_t *__restrict__ pending, uint32_t *__restrict__ qidx, uint64_t *__restrict__ qstate, const uint4 *__restrict__ unis)
{
	__shared__ EsTab tabs[8];
	__shared__ EsLocal loc;
	__shared__ EsUni uni;
	const EsWork wk = work[blockIdx.x];
	const uint32_t n = n_in[wk.scan];
	if (wk.first >= n) /* most workgroups, from the second round on: gone before the tables are copied */
		return;
	const DevScan &sc = scans[wk.scan];
	es_load_tables(sc, nullptr, huff + sc.tab_off, tabs, &loc, nullptr, false, &uni, unis + (size_t)(sc.tab_off >> 3) * (sizeof(EsUni) / 16u));
	const uint32_t t = wk.first + threadIdx.x;
	if (t >= n)
		return;
	const uint32_t i = qidx_in[sc.sub_off + t];
	const uint64_t want = qstate_in[sc.sub_off + t];
	if (i >= sc.nsub)
		return;
	const uint32_t slot = sc.sub_off + i;
	start[slot] = want;
	EsState s = es_unpack(want);
	const uint32_t pe = min((i + 1u) * sc.sub_bits, sc.nbits);
	cnt[slot] = es_state_walk(sc, loc, tabs, uni, streams + sc.stream_off, s, pe);
	const uint64_t now = es_pack(s), was = end[slot];
	end[slot] = now;
	es_push(i + 1u < sc.nsub && now != was, i + 1u, now, sc.sub_off, &pending[wk.scan], qidx, qstate);
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
/* CB: the scans with compact planes (true) or with int16 planes (false); a workgroup of the other kind leaves at once */
template <bool CB>
MIJ_ES_KERNEL void k_es_write(const DevScan *__restrict__ scans, const EsWork *__restrict__ work, const DevHuff *__restrict__ huff,
																  const uint8_t *__restrict__ streams, const DevImage *__restrict__ imgs, const uint64_t *__restrict__ start,
																  const uint32_t *__restrict__ base, int16_t *__restrict__ coef, uint64_t *__restrict__ meta,
																  uint32_t *__restrict__ anom, uint32_t *__restrict__ pfinal, uint8_t *__restrict__ zz)
{
	__shared__ EsTab tabs[8];
	__shared__ EsLocal loc;
	__shared__ EsPair pair;
	__shared__ uint8_t zpos[64];
	__shared__ uint16_t toff[64];
	const EsWork wk = work[blockIdx.x];
	const DevScan &sc = scans[wk.scan];
	if ((sc.fmt != 0u) != CB)
		return;
	if (threadIdx.x < 64) {
		const uint32_t P = mij_zigzag_pos[threadIdx.x];
		zpos[threadIdx.x] = (uint8_t)P;
		toff[threadIdx.x] = (uint16_t)(((P >> 3) << 9) + (P & 7u));
	}
	es_load_tables(sc, &imgs[sc.img], huff + sc.tab_off, tabs, &loc, MIJ_ES_PAIR ? &pair : nullptr, true);
	const uint32_t i = wk.first + threadIdx.x;
	if (i >= sc.nsub)
		return;
	const uint32_t slot = sc.sub_off + i;
	EsState s = es_unpack(start[slot]);
	EsWriter wr;
	wr.sc = &sc;
	wr.loc = &loc;
	wr.coef = coef;
	wr.meta = meta + sc.blk_off;
	wr.toff = toff;
	wr.zpos = zpos;
	wr.owner = true;
	wr.grp = 0;
	wr.curq = 0;
	wr.esc = false;
	wr.dcd = 0;
	wr.blk = nullptr;
	wr.mx = wr.my = 0;
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
	wr.zz = zz + ((size_t)(sc.blk_off + wr.ord) << 6);
	wr.my = m / sc.mcu_x;
	wr.mx = m - wr.my * sc.mcu_x;
	if (!CB)
		wr.locate(s.c);
	const uint32_t pe = min((i + 1u) * sc.sub_bits, sc.nbits);
	es_decode<true, MIJ_ES_PAIR != 0, CB, false>(sc, loc, tabs, streams + sc.stream_off, s, pe, &wr, &anom[wk.scan], &pair);
}

/* the rest of every block that began in the previous subsequence: single coefficients into the block that the
 * previous thread's k_es_write stored whole (stream order makes this the later write) */
template <bool CB>
MIJ_ES_KERNEL void k_es_tails(const DevScan *__restrict__ scans, const EsWork *__restrict__ work, const DevHuff *__restrict__ huff,
																  const uint8_t *__restrict__ streams, const DevImage *__restrict__ imgs, const uint64_t *__restrict__ start,
																  const uint32_t *__restrict__ base, int16_t *__restrict__ coef, uint64_t *__restrict__ meta,
																  uint32_t *__restrict__ scratch, uint8_t *__restrict__ zz)
{
	__shared__ EsTab tabs[8];
	__shared__ EsLocal loc;
	__shared__ uint8_t zpos[64];
	__shared__ uint16_t toff[64];
	const EsWork wk = work[blockIdx.x];
	const DevScan &sc = scans[wk.scan];
	if ((sc.fmt != 0u) != CB)
		return;
	if (threadIdx.x < 64) {
		const uint32_t P = mij_zigzag_pos[threadIdx.x];
		zpos[threadIdx.x] = (uint8_t)P;
		toff[threadIdx.x] = (uint16_t)(((P >> 3) << 9) + (P & 7u));
	}
	const uint32_t i = wk.first + threadIdx.x;
	const uint32_t slot = sc.sub_off + i;
	EsState s;
	s.p = s.z = s.c = 0;
	if (i < sc.nsub)
		s = es_unpack(start[slot]);
	const bool mine = i < sc.nsub && s.z != 0 && s.z != MIJ_ES_DEAD; /* a block begun by the previous subsequence */
	if (!__syncthreads_or(mine ? 1 : 0))
		return;
	es_load_tables(sc, &imgs[sc.img], huff + sc.tab_off, tabs, &loc);
	if (!mine)
		return;
	EsWriter wr;
	wr.sc = &sc;
	wr.loc = &loc;
	wr.coef = coef;
	wr.meta = meta + sc.blk_off;
	wr.toff = toff;
	wr.zpos = zpos;
	wr.owner = false;
	wr.grp = 0;
	wr.curq = 0;
	wr.esc = false;
	wr.dcd = 0;
	wr.blk = nullptr;
	wr.mx = wr.my = 0;
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
	wr.zz = zz + ((size_t)(sc.blk_off + wr.ord) << 6);
	wr.my = m / sc.mcu_x;
	wr.mx = m - wr.my * sc.mcu_x;
	if (!CB)
		wr.locate(s.c);
	/* k_es_write walked the same symbols and reported what there was to report: verdict bits go to a scratch word */
	es_decode<true, false, CB, true>(sc, loc, tabs, streams + sc.stream_off, s, sc.nbits, &wr, scratch + 1);
}

/* The write pass in record form (es_write_records): compact-plane scans only.  rec: the record arena, region64 eight-byte words per
 * subsequence (slot order); meta[block] low dword <- the record index of the block's DC record (k_es_pack2 replaces the word). */
MIJ_ES_KERNEL void k_es_writer(const DevScan *__restrict__ scans, const EsWork *__restrict__ work, const DevHuff *__restrict__ huff,
																	const uint8_t *__restrict__ streams, const uint64_t *__restrict__ start, const uint32_t *__restrict__ base,
																	uint64_t *__restrict__ meta, uint32_t *__restrict__ anom, uint32_t *__restrict__ pfinal, uint64_t *__restrict__ rec,
																	uint32_t region64, const uint4 *__restrict__ wtabs)
{
	__shared__ EsW w;
	__shared__ uint32_t rbuf[MIJ_ES_REC_BUF / 2u][MIJ_ES_WG]; /* EsRecOut: dword j of every lane's records side by side */
	const EsWork wk = work[blockIdx.x];
	const DevScan &sc = scans[wk.scan];
	if (!sc.fmt)
		return;
	{
		const uint4 *src = wtabs + (size_t)(sc.tab_off >> 3) * (sizeof(EsW) / 16u);
		uint4 *dst = reinterpret_cast<uint4 *>(&w);
		for (uint32_t q = threadIdx.x; q < sizeof(EsW) / 16u; q += blockDim.x)
			dst[q] = src[q];
		__syncthreads();
	}
	const uint32_t i = wk.first + threadIdx.x;
	if (i >= sc.nsub)
		return;
	const uint32_t slot = sc.sub_off + i;
	EsState s = es_unpack(start[slot]);
	uint32_t ord = base[slot];
	if (ord >= sc.nblocks) /* nothing left for this subsequence; no reader comes here (the one before ended the scan's records) */
		return;
	EsRecOut out;
	out.slot = rec + (size_t)slot * region64;
	out.buf = &rbuf[0][threadIdx.x];
	out.cnt = 0;
	out.index = slot * region64 * 4u;
	const uint32_t ml = ord / sc.bpm;
	if (ord - ml * sc.bpm != s.c) { /* the chain is inconsistent: cannot happen after convergence */
		atomicOr(&anom[wk.scan], 2u);
		s.z = MIJ_ES_DEAD;
	} else {
		const uint32_t pe = min((i + 1u) * sc.sub_bits, sc.nbits);
		es_write_records(sc, w, streams + sc.stream_off, s, pe, ord, out, reinterpret_cast<uint32_t *>(meta + sc.blk_off), &anom[wk.scan], &pfinal[wk.scan]);
	}
	/* every region ends in a way out: on to the next subsequence's records, or the end of the scan's */
	if (ord >= sc.nblocks || s.z == MIJ_ES_DEAD || i + 1u >= sc.nsub)
		out.emit(MIJ_ES_REC_END);
	else {
		const uint32_t next = (slot + 1u) * region64 * 4u;
		if ((out.cnt & 3u) >= 2u) /* a jump and its two index records stay inside one 8-byte word (k_es_pack2 reads them from the word in hand) */
			out.finish();
		out.emit(MIJ_ES_REC_JUMP);
		out.emit(next & 0xffffu);
		out.emit(next >> 16);
	}
	out.finish();
}

/* The intermediate image of the write pass (64 bytes per block in zigzag order, blocks in scan order) -> the tiles of
 * the compact planes: one block per lane, its bytes permuted in registers into in-block position order P (mij.h), chunk
 * rows stored coalesced.  Byte 0 (the DC's place) carries the block's flags through.  Blocks of a tile beyond the
 * component's grid come out as zeros.  work.comp = component, work.first = first block. */
__global__ __launch_bounds__(256) void k_es_pack(const DevImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ zz,
																 uint8_t *__restrict__ coef, uint64_t *__restrict__ meta)
{
	const WorkIdct wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const DevComp &cp = im.comp[wk.comp];
	const uint32_t nblk = (uint32_t)(cp.bw * cp.bh), ntile = (nblk + 63u) >> 6;
	const uint32_t L = wk.first + threadIdx.x;
	if (L >= ntile * 64u)
		return;
	uint32_t in[16];
#pragma unroll
	for (int i = 0; i < 16; ++i)
		in[i] = 0;
	uint32_t ord = 0;
	if (L < nblk) {
		const uint32_t by = L / (uint32_t)cp.bw, bx = L - by * (uint32_t)cp.bw;
		const uint32_t mx = bx / (uint32_t)cp.h, dx = bx - mx * (uint32_t)cp.h, my = by / (uint32_t)cp.v, dy = by - my * (uint32_t)cp.v;
		ord = (my * (uint32_t)im.mcu_x + mx) * im.es_bpm + im.es_j0[wk.comp] + dy * (uint32_t)cp.h + dx;
		const uint4 *src = reinterpret_cast<const uint4 *>(zz + ((size_t)(im.es_blk_off + ord) << 6));
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(src + i));
			in[4 * i] = v.x, in[4 * i + 1] = v.y, in[4 * i + 2] = v.z, in[4 * i + 3] = v.w;
		}
	}
	uint32_t out[16];
#pragma unroll
	for (int i = 0; i < 16; ++i)
		out[i] = 0;
#pragma unroll
	for (int k = 0; k < 64; ++k) { /* constant indices: the compiler folds this into byte permutes */
		const int P = mij_zigzag_pos[k];
		out[P >> 2] |= ((in[k >> 2] >> (8 * (k & 3))) & 255u) << (8 * (P & 3));
	}
	uint8_t *dst = coef + cp.coef_off + ((size_t)(L >> 6) << 12) + ((size_t)(L & 63u) << 3);
#pragma unroll
	for (int c = 0; c < 8; ++c)
		*reinterpret_cast<uint2 *>(dst + (c << 9)) = make_uint2(out[2 * c], out[2 * c + 1]);
	if (L >= nblk)
		return;
	/* the block's L1 of de-quantised AC coefficients (the bound behind MIJ_FLAG_WIDE_IDCT; k_es_dc adds the DC term and takes
	 * the maximum): sum over the positions of |(short)(coef * q)|, escape bytes included where the block has them */
	const uint32_t *dq = im.dq[wk.comp];
	const bool esc = (out[0] & 1u) != 0;
	uint32_t hi[16];
#pragma unroll
	for (int i = 0; i < 16; ++i)
		hi[i] = 0;
	if (esc) {
		const uint4 *hp = reinterpret_cast<const uint4 *>(coef + cp.hi_off + ((size_t)L << 6));
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const uint4 v = hp[i];
			hi[4 * i] = v.x, hi[4 * i + 1] = v.y, hi[4 * i + 2] = v.z, hi[4 * i + 3] = v.w;
		}
	}
	uint32_t l1 = 0;
	/* quantisers that fit a byte (every 8-bit DQT) and no escaped block among the wavefront's: |(short)(c * q)| = |c| * q exactly (|c| <= 128,
	 * q <= 255: the product cannot wrap), so the sum is sixteen v_dot4_u32_u8 of per-byte absolute values -- |c| = (c ^ m) + s with s the sign
	 * bits and m = 255 * s -- instead of sixty-three sign-extend / multiply / abs / add steps (round 3: 0.54 -> 0.?? ms per 256 pictures) */
	uint32_t qhigh = 0;
#pragma unroll
	for (int i = 0; i < 32; ++i)
		qhigh |= dq[i];
	if ((qhigh & 0xff00ff00u) == 0u && __builtin_amdgcn_ballot_w64(esc) == 0ull) { /* wave-uniform */
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const uint32_t w = i == 0 ? (out[0] & 0xffffff00u) : out[i]; /* position 0 is the DC's place (the flags byte) */
			const uint32_t sg = (w >> 7) & 0x01010101u, m = (sg << 8) - sg;
			const uint32_t a = (w ^ m) + sg;
			const uint32_t q4 = (dq[2 * i] & 0xffu) | ((dq[2 * i] >> 8) & 0xff00u) | ((dq[2 * i + 1] & 0xffu) << 16) | ((dq[2 * i + 1] >> 16) << 24);
			l1 = __builtin_amdgcn_udot4(a, q4, l1, false);
		}
	} else {
#pragma unroll
		for (int P = 1; P < 64; ++P) { /* position 0 is the DC's place */
			const int lo8 = (int)(int8_t)((out[P >> 2] >> (8 * (P & 3))) & 255u), hi8 = (int)(int8_t)((hi[P >> 2] >> (8 * (P & 3))) & 255u);
			const uint32_t q = (dq[P >> 1] >> (16 * (P & 1))) & 0xffffu;
			const int v = (int)(int16_t)((uint32_t)(lo8 + 256 * hi8) * q);
			l1 += (uint32_t)(v < 0 ? -v : v);
		}
	}
	reinterpret_cast<uint32_t *>(meta + im.es_blk_off + ord)[0] = l1;
}

/* The same from the record stream of k_es_writer: every lane follows its block's records (meta[block] = index of its DC record) into a
 * 64-byte image held in LDS, then permutes and stores it exactly like k_es_pack; escape bytes go straight to their plane. */
__global__ __launch_bounds__(256) void k_es_pack2(const DevImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint64_t *__restrict__ rec,
																  uint32_t rec_words, uint8_t *__restrict__ coef, uint64_t *__restrict__ meta)
{
	const WorkIdct wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const DevComp &cp = im.comp[wk.comp];
	const uint32_t nblk = (uint32_t)(cp.bw * cp.bh), ntile = (nblk + 63u) >> 6;
	const uint32_t L = wk.first + threadIdx.x;
	if (L >= ntile * 64u)
		return;
	uint32_t in[16];
#pragma unroll
	for (int i = 0; i < 16; ++i)
		in[i] = 0;
	/* the block's 64 bytes in zigzag order, gathered in LDS: dword j of all 64 lanes of a wavefront side by side (every lane its own bank) */
	__shared__ uint32_t stage[4][16][64];
	const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
#pragma unroll
	for (int i = 0; i < 16; ++i)
		stage[wv][i][ln] = 0;
	uint32_t ord = 0;
	int dcd = 0;
	bool escaped = false;
	if (L < nblk) {
		const uint32_t by = L / (uint32_t)cp.bw, bx = L - by * (uint32_t)cp.bw;
		const uint32_t mx = bx / (uint32_t)cp.h, dx = bx - mx * (uint32_t)cp.h, my = by / (uint32_t)cp.v, dy = by - my * (uint32_t)cp.v;
		ord = (my * (uint32_t)im.mcu_x + mx) * im.es_bpm + im.es_j0[wk.comp] + dy * (uint32_t)cp.h + dx;
		const uint32_t idx = reinterpret_cast<const uint32_t *>(meta + im.es_blk_off + ord)[0];
		uint8_t *const lane_bytes = reinterpret_cast<uint8_t *>(&stage[wv][0][ln]); /* byte k of the block: lane_bytes[(k >> 2) * 256 + (k & 3)] */
		uint8_t *const hi8 = coef + cp.hi_off + ((size_t)L << 6);
		uint32_t a = idx >> 2, prevk = 0;
		bool going = idx != 0xffffffffu && a < rec_words; /* 0xffffffff: no subsequence ever began this block (the image is handed back) */
		uint64_t cur = going ? rec[a] : 0ull;
		if (idx & 3u) { /* the records in front of this block's in its first word: padding to the reader */
			const uint64_t m = (1ull << (16u * (idx & 3u))) - 1ull;
			cur = (cur & ~m) | (0xf000f000f000f000ull & m);
		}
		bool first = true;
		/* a block has at most 1 + 63 records, as many escapes and a jump per subsequence it crosses: the bound only guards against a chain
		 * that was never written (an image with an anomaly).  One 8-byte word = four records per step, the next word already on its way. */
		for (uint32_t step = 0; going && step < 160u; ++step) {
			uint64_t nxt = rec[min(a + 1u, rec_words - 1u)];
			bool jumped = false;
#pragma unroll
			for (uint32_t q = 0; q < 4u; ++q) {
				const uint32_t r = (uint32_t)(cur >> (16u * q)) & 0xffffu;
				if (!going || jumped)
					continue;
				if (!(r & 0x8000u)) { /* AC coefficient */
					prevk = (r >> 8) & 63u;
					lane_bytes[(prevk >> 2) * 256u + (prevk & 3u)] = (uint8_t)r;
					continue;
				}
				const uint32_t kind = r & 0xf000u;
				if (kind == MIJ_ES_REC_DC) {
					if (!first)
						going = false; /* the next block's: done */
					else
						dcd = ((int)(r << 20)) >> 20;
					first = false;
				} else if (kind == MIJ_ES_REC_ESC) {
					if (!escaped) { /* the block's first escape clears its 64 escape bytes (nothing else does) */
						uint4 *h = reinterpret_cast<uint4 *>(hi8);
						h[0] = h[1] = h[2] = h[3] = make_uint4(0, 0, 0, 0);
						escaped = true;
					}
					hi8[mij_zigzag_pos[prevk]] = (uint8_t)r;
				} else if (kind == MIJ_ES_REC_JUMP) { /* q <= 1 (k_es_writer): its index is in this word */
					const uint32_t nidx = ((uint32_t)(cur >> (16u * ((q + 1u) & 3u))) & 0xffffu) | ((uint32_t)(cur >> (16u * ((q + 2u) & 3u))) & 0xffffu) << 16;
					a = (nidx >> 2) - 1u; /* the first record of a region: a multiple of four */
					if (q > 1u || a + 1u >= rec_words)
						going = false;
					else
						nxt = rec[a + 1u];
					jumped = true;
				} else if (kind == MIJ_ES_REC_END)
					going = false;
				/* padding: nothing */
			}
			cur = nxt;
			++a;
			if (a >= rec_words)
				going = false;
		}
		if (escaped)
			lane_bytes[0] = 1; /* the flags byte sits in the DC's place */
	}
#pragma unroll
	for (int i = 0; i < 16; ++i)
		in[i] = stage[wv][i][ln];
	uint32_t out[16];
#pragma unroll
	for (int i = 0; i < 16; ++i)
		out[i] = 0;
#pragma unroll
	for (int k = 0; k < 64; ++k) { /* constant indices: the compiler folds this into byte permutes */
		const int P = mij_zigzag_pos[k];
		out[P >> 2] |= ((in[k >> 2] >> (8 * (k & 3))) & 255u) << (8 * (P & 3));
	}
	uint8_t *dst = coef + cp.coef_off + ((size_t)(L >> 6) << 12) + ((size_t)(L & 63u) << 3);
#pragma unroll
	for (int c = 0; c < 8; ++c)
		*reinterpret_cast<uint2 *>(dst + (c << 9)) = make_uint2(out[2 * c], out[2 * c + 1]);
	if (L >= nblk)
		return;
	/* the block's L1 of de-quantised AC coefficients (the bound behind MIJ_FLAG_WIDE_IDCT; k_es_dc adds the DC term and takes
	 * the maximum): sum over the positions of |(short)(coef * q)|, escape bytes included where the block has them */
	const uint32_t *dq = im.dq[wk.comp];
	const bool esc = (out[0] & 1u) != 0;
	uint32_t hi[16];
#pragma unroll
	for (int i = 0; i < 16; ++i)
		hi[i] = 0;
	if (esc) {
		const uint4 *hp = reinterpret_cast<const uint4 *>(coef + cp.hi_off + ((size_t)L << 6));
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const uint4 v = hp[i];
			hi[4 * i] = v.x, hi[4 * i + 1] = v.y, hi[4 * i + 2] = v.z, hi[4 * i + 3] = v.w;
		}
	}
	uint32_t l1 = 0;
	/* quantisers that fit a byte (every 8-bit DQT) and no escaped block among the wavefront's: |(short)(c * q)| = |c| * q exactly (|c| <= 128,
	 * q <= 255: the product cannot wrap), so the sum is sixteen v_dot4_u32_u8 of per-byte absolute values -- |c| = (c ^ m) + s with s the sign
	 * bits and m = 255 * s -- instead of sixty-three sign-extend / multiply / abs / add steps (round 3: 0.54 -> 0.?? ms per 256 pictures) */
	uint32_t qhigh = 0;
#pragma unroll
	for (int i = 0; i < 32; ++i)
		qhigh |= dq[i];
	if ((qhigh & 0xff00ff00u) == 0u && __builtin_amdgcn_ballot_w64(esc) == 0ull) { /* wave-uniform */
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const uint32_t w = i == 0 ? (out[0] & 0xffffff00u) : out[i]; /* position 0 is the DC's place (the flags byte) */
			const uint32_t sg = (w >> 7) & 0x01010101u, m = (sg << 8) - sg;
			const uint32_t a = (w ^ m) + sg;
			const uint32_t q4 = (dq[2 * i] & 0xffu) | ((dq[2 * i] >> 8) & 0xff00u) | ((dq[2 * i + 1] & 0xffu) << 16) | ((dq[2 * i + 1] >> 16) << 24);
			l1 = __builtin_amdgcn_udot4(a, q4, l1, false);
		}
	} else {
#pragma unroll
		for (int P = 1; P < 64; ++P) { /* position 0 is the DC's place */
			const int lo8 = (int)(int8_t)((out[P >> 2] >> (8 * (P & 3))) & 255u), hi8 = (int)(int8_t)((hi[P >> 2] >> (8 * (P & 3))) & 255u);
			const uint32_t q = (dq[P >> 1] >> (16 * (P & 1))) & 0xffffu;
			const int v = (int)(int16_t)((uint32_t)(lo8 + 256 * hi8) * q);
			l1 += (uint32_t)(v < 0 ? -v : v);
		}
	}
	meta[im.es_blk_off + ord] = (uint64_t)l1 | ((uint64_t)(uint16_t)(int16_t)dcd << 32); /* what k_es_dc reads: L1 of the AC coefficients | DC difference */
}

/* DC prediction (codec/jpeg.c:323-325), L1 bound and the completion checks; one workgroup per scan */
__global__ __launch_bounds__(256) void k_es_dc(const DevScan *__restrict__ scans, const DevImage *__restrict__ imgs, const uint32_t *__restrict__ total, const uint32_t *__restrict__ changed, int16_t *__restrict__ coef,
															  const uint64_t *__restrict__ meta, uint32_t *__restrict__ anom,
															  uint32_t *__restrict__ l1max, const uint32_t *__restrict__ pfinal, const uint8_t *__restrict__ streams)
{
	__shared__ int part[256][4];
	__shared__ uint32_t wmax[256];
	const DevScan &sc = scans[blockIdx.x];
	const DevImage &im = imgs[sc.img];
	const uint32_t nmcu = sc.nblocks / sc.bpm;
	const uint32_t per = (nmcu + 255u) / 256u;
	const uint32_t lo = min(threadIdx.x * per, nmcu), hi = min(lo + per, nmcu);
	const uint64_t *mt = meta + sc.blk_off; /* L1 of the AC coefficients | DC difference << 32 (EsWriter::end_block) */
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
			sum[sc.blk_comp[c]] += (int)(int16_t)(mt[m * sc.bpm + c] >> 32);
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
			const uint64_t w = mt[m * sc.bpm + c];
			pred[ci] = (int)((unsigned)pred[ci] + (unsigned)(int)(int16_t)(w >> 32));
			const uint32_t bx = mx * (uint32_t)cp.h + sc.blk_dx[c], by = my * (uint32_t)cp.v + sc.blk_dy[c];
			const uint32_t L = bx + by * (uint32_t)cp.bw;
			if (sc.fmt)
				reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(coef) + cp.dc_off)[L] = (int16_t)pred[ci];
			else
				(reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(coef) + cp.coef_off) + ((size_t)(L >> 6) << 12) + ((L & 63u) << 3))[0] = (int16_t)pred[ci];
			const int dq = (int)(int16_t)((uint32_t)pred[ci] * sc.qz[ci][0]);
			const uint32_t tot = (uint32_t)w + (uint32_t)(dq < 0 ? -dq : dq);
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
