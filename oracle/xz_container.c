/*
 * xz_container.c -- TEST INFRASTRUCTURE ONLY (oracle).
 *
 * CPU restatement of the .xz container primitives the multi-threaded stream
 * encoder emits around the LZMA2 payload.  Each function names the reference
 * code it follows (paths relative to /root/reference) and the prose spec
 * (doc/xz-file-format.txt).  Pinned byte-for-byte against the real library in
 * tests/test_oracle_container.py.
 */
#include "oracle.h"
#include <string.h>

/* ---- CRC32 / CRC64 ---------------------------------------------------
 * check/crc32_fast.c:44-92 and check/crc64_fast.c:49-112 compute the same
 * reflected CRCs with slice-by-N tables; a bytewise table is the definition
 * (doc/xz-file-format.txt:1049-1143).  Polynomials: 0xEDB88320 (IEEE),
 * 0xC96C5795D7870F42 (ECMA-182). */
static uint32_t t32[256];
static uint64_t t64[256];
static int tables_ready;

static void make_tables(void)
{
	for (uint32_t b = 0; b < 256; ++b) {
		uint32_t r = b;
		uint64_t q = b;
		for (int k = 0; k < 8; ++k) {
			r = (r >> 1) ^ ((r & 1) ? 0xEDB88320u : 0);
			q = (q >> 1) ^ ((q & 1) ? 0xC96C5795D7870F42ull : 0);
		}
		t32[b] = r;
		t64[b] = q;
	}
	tables_ready = 1;
}

uint32_t orc_crc32(const uint8_t *buf, size_t n, uint32_t crc)
{
	if (!tables_ready) make_tables();
	crc = ~crc;
	for (size_t i = 0; i < n; ++i)
		crc = t32[(crc ^ buf[i]) & 0xFF] ^ (crc >> 8);
	return ~crc;
}

uint64_t orc_crc64(const uint8_t *buf, size_t n, uint64_t crc)
{
	if (!tables_ready) make_tables();
	crc = ~crc;
	for (size_t i = 0; i < n; ++i)
		crc = t64[(crc ^ buf[i]) & 0xFF] ^ (crc >> 8);
	return ~crc;
}

/* GF(2) helpers for the reflected CRC64: multiply two residues mod P and
 * x^(8*nbytes) mod P.  In the reflected representation bit 63 is x^0. */
static uint64_t gf_mul(uint64_t a, uint64_t b)
{
	uint64_t r = 0;
	for (int i = 0; i < 64; ++i) {
		if (a & 0x8000000000000000ull)
			r ^= b;
		a <<= 1;
		b = (b >> 1) ^ ((b & 1) ? 0xC96C5795D7870F42ull : 0);
	}
	return r;
}

static uint64_t gf_xpow8(uint64_t nbytes)
{
	/* x^8 in reflected form: bit (63-8). */
	uint64_t base = 0x8000000000000000ull >> 8;
	uint64_t acc = 0x8000000000000000ull; /* 1 */
	while (nbytes) {
		if (nbytes & 1)
			acc = gf_mul(acc, base);
		base = gf_mul(base, base);
		nbytes >>= 1;
	}
	return acc;
}

uint64_t orc_crc64_combine(uint64_t crc_a, uint64_t crc_b, uint64_t len_b)
{
	/* crc(A||B) = crc(A) * x^(8 len_b)  xor  crc(B)  (with the all-ones
	 * init/final xor cancelling as in zlib's crc32_combine). */
	return gf_mul(crc_a, gf_xpow8(len_b)) ^ crc_b;
}

/* ---- VLI: common/vli_encoder.c:16-68, vli_size.c:16-28; spec :184-250 -- */
uint32_t orc_vli_size(uint64_t v)
{
	uint32_t n = 1;
	while (v >>= 7) ++n;
	return n;
}

uint32_t orc_vli_encode(uint64_t v, uint8_t *out)
{
	uint32_t n = 0;
	while (v >= 0x80) {
		out[n++] = (uint8_t)(v | 0x80);
		v >>= 7;
	}
	out[n++] = (uint8_t)v;
	return n;
}

uint32_t orc_check_size(int check)
{
	/* check/check.c:16-40 (table of sizes by Check ID) */
	static const uint8_t sz[16] = { 0, 4, 4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64 };
	return sz[check & 15];
}

static void put32le(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

/* ---- Stream Header / Footer: common/stream_flags_encoder.c:29-85,
 * magic bytes common/stream_flags_common.c:15-16; spec :306-453 ---------- */
void orc_stream_header(uint8_t out[12], int check)
{
	static const uint8_t magic[6] = { 0xFD, '7', 'z', 'X', 'Z', 0x00 };
	memcpy(out, magic, 6);
	out[6] = 0x00;
	out[7] = (uint8_t)check;
	put32le(out + 8, orc_crc32(out + 6, 2, 0));
}

void orc_stream_footer(uint8_t out[12], uint64_t index_size, int check)
{
	put32le(out + 4, (uint32_t)(index_size / 4 - 1));
	out[8] = 0x00;
	out[9] = (uint8_t)check;
	put32le(out, orc_crc32(out + 4, 6, 0));
	out[10] = 'Y';
	out[11] = 'Z';
}

/* ---- LZMA2 dictionary-size property: lzma/lzma2_encoder.c:376-400 ------- */
uint8_t orc_lzma2_dict_byte(uint32_t d)
{
	if (d < 4096) d = 4096;
	/* round up to 2^n or 2^n + 2^(n-1) */
	--d;
	d |= d >> 2; d |= d >> 3; d |= d >> 4; d |= d >> 8; d |= d >> 16;
	if (d == 0xFFFFFFFFu)
		return 40;
	++d;
	/* get_dist_slot(d) - 24 where slot = 2*bsr + next bit */
	uint32_t bsr = 31;
	while (!(d >> bsr)) --bsr;
	uint32_t slot = 2 * bsr + ((d >> (bsr - 1)) & 1);
	return (uint8_t)(slot - 24);
}

/* ---- Block sizes: common/block_buffer_encoder.c:20-69 -------------------- */
uint64_t orc_block_bound(uint64_t u)
{
	const uint64_t headers_bound = (1 + 1 + 2 * 9 + 3 + 4 + 64 + 3) & ~3ull;
	uint64_t overhead = ((u + 65535) / 65536) * 3 + 1;
	uint64_t l2 = (u + overhead + 3) & ~3ull;
	return headers_bound + l2;
}

/* ---- Block Header: common/block_header_encoder.c:17-131,
 * filter flags common/filter_flags_encoder.c:30-55; spec :483-616 ---------- */
uint32_t orc_block_header_size(uint64_t csize, uint64_t usize, int with_x86)
{
	uint32_t size = 1 + 1 + 4;
	size += orc_vli_size(csize);
	size += orc_vli_size(usize);
	if (with_x86)
		size += 2;          /* id 0x04, props size 0 */
	size += 3;                  /* id 0x21, props size 1, dict byte */
	return (size + 3) & ~3u;
}

void orc_block_header_encode(uint8_t *out, uint32_t header_size,
		uint64_t csize, uint64_t usize, uint8_t dict_byte, int with_x86)
{
	uint32_t body = header_size - 4;
	memset(out, 0, body);
	out[0] = (uint8_t)(body / 4);
	out[1] = (uint8_t)(0x40 | 0x80 | (with_x86 ? 1 : 0));
	uint32_t pos = 2;
	pos += orc_vli_encode(csize, out + pos);
	pos += orc_vli_encode(usize, out + pos);
	if (with_x86) {
		out[pos++] = 0x04;
		out[pos++] = 0x00;
	}
	out[pos++] = 0x21;
	out[pos++] = 0x01;
	out[pos++] = dict_byte;
	put32le(out + body, orc_crc32(out, body, 0));
}

/* ---- Index: common/index_encoder.c:43-164; spec :647-755 ---------------- */
uint64_t orc_index_encode(uint8_t *out, uint64_t nblocks,
		const uint64_t *unpadded, const uint64_t *uncompressed)
{
	uint64_t size = 1 + orc_vli_size(nblocks);
	for (uint64_t i = 0; i < nblocks; ++i)
		size += orc_vli_size(unpadded[i]) + orc_vli_size(uncompressed[i]);
	uint64_t padded = (size + 3) & ~3ull;
	if (out) {
		uint64_t pos = 0;
		out[pos++] = 0x00;
		pos += orc_vli_encode(nblocks, out + pos);
		for (uint64_t i = 0; i < nblocks; ++i) {
			pos += orc_vli_encode(unpadded[i], out + pos);
			pos += orc_vli_encode(uncompressed[i], out + pos);
		}
		while (pos < padded)
			out[pos++] = 0;
		put32le(out + pos, orc_crc32(out, pos, 0));
	}
	return padded + 4;
}

static void put_check(uint8_t *out, int check, const uint8_t *in, uint64_t n)
{
	if (check == ORC_CHECK_CRC32) {
		put32le(out, orc_crc32(in, n, 0));
	} else if (check == ORC_CHECK_CRC64) {
		uint64_t c = orc_crc64(in, n, 0);
		put32le(out, (uint32_t)c);
		put32le(out + 4, (uint32_t)(c >> 32));
	}
}

/* ---- Whole-Block uncompressed fallback: block_buffer_encoder.c:88-162,
 * caller side stream_encoder_mt.c:316-344 ------------------------------- */
uint64_t orc_block_uncomp_encode(const uint8_t *in, uint64_t n, int check,
		uint8_t *out, uint64_t *unpadded)
{
	uint64_t csize = n + ((n + 65535) / 65536) * 3 + 1;
	uint32_t hs = orc_block_header_size(csize, n, 0);
	orc_block_header_encode(out, hs, csize, n, 0x00 /* dict 4 KiB */, 0);
	uint64_t pos = hs;
	uint8_t control = 0x01;
	for (uint64_t ip = 0; ip < n; ) {
		uint64_t c = n - ip < 65536 ? n - ip : 65536;
		out[pos++] = control;
		control = 0x02;
		out[pos++] = (uint8_t)((c - 1) >> 8);
		out[pos++] = (uint8_t)(c - 1);
		memcpy(out + pos, in + ip, c);
		pos += c;
		ip += c;
	}
	out[pos++] = 0x00;
	while ((pos - hs) & 3)
		out[pos++] = 0;
	put_check(out + pos, check, in, n);
	pos += orc_check_size(check);
	*unpadded = hs + csize + orc_check_size(check);
	return pos;
}

/* ---- Whole Stream as stream_encoder_mt.c emits it ------------------------
 * Header size is fixed from the MAXIMUM sizes before encoding
 * (stream_encoder_mt.c:225-237: compressed_size = outbuf->allocated =
 * lzma_block_buffer_bound64(block_size) (:995), uncompressed_size =
 * block_size); real sizes are filled in afterwards (:307).  A Block whose
 * payload does not fit the buffer takes the uncompressed fallback
 * (:298,:316-344).  Index/footer sequencing :842-883. */
uint64_t orc_xz_frame(const uint8_t *const *payloads, const uint64_t *payload_sizes,
		const uint8_t *const *inputs, const uint64_t *input_sizes,
		uint64_t nblocks, uint64_t block_size, uint32_t dict_size,
		int check, uint8_t *out, uint64_t out_cap)
{
	(void)out_cap;
	uint64_t pos = 0;
	orc_stream_header(out, check);
	pos += 12;
	const uint64_t allocated = orc_block_bound(block_size);
	const uint32_t hs = orc_block_header_size(allocated, block_size, 0);
	const uint32_t cs = orc_check_size(check);
	uint64_t *unp = (uint64_t *)__builtin_malloc(sizeof(uint64_t) * (nblocks ? nblocks : 1));
	for (uint64_t b = 0; b < nblocks; ++b) {
		uint64_t csize = payload_sizes[b];
		uint64_t padded = (csize + 3) & ~3ull;
		if (hs + padded + cs > allocated) {
			pos += orc_block_uncomp_encode(inputs[b], input_sizes[b], check,
					out + pos, &unp[b]);
			continue;
		}
		orc_block_header_encode(out + pos, hs, csize, input_sizes[b],
				orc_lzma2_dict_byte(dict_size), 0);
		pos += hs;
		memcpy(out + pos, payloads[b], csize);
		pos += csize;
		while (pos & 3)     /* Stream Header (12) + headers keep 4-alignment */
			out[pos++] = 0;
		put_check(out + pos, check, inputs[b], input_sizes[b]);
		pos += cs;
		unp[b] = hs + csize + cs;
	}
	uint64_t isz = orc_index_encode(out + pos, nblocks, unp, input_sizes);
	pos += isz;
	orc_stream_footer(out + pos, isz, check);
	pos += 12;
	__builtin_free(unp);
	return pos;
}


/* ------------------------------------------------------------------ */
/* x86 BCJ encoder (simple/x86.c:26-118) for one whole Block            */
/* ------------------------------------------------------------------ */
/* Fresh state (x86.c:121-136: prev_mask = 0, prev_pos = -5), start offset 0, the whole Block as one
 * buffer; the last <= 4 bytes pass through (simple_coder.c:178-181).  In place. */
void orc_x86_encode(uint8_t *buf, uint64_t size)
{
	static const uint32_t mask_to_bit[5] = { 0, 1, 2, 2, 3 };
	uint32_t prev_mask = 0;
	uint32_t prev_pos = (uint32_t)(-5);
	if (size < 5)
		return;
	const uint64_t limit = size - 5;
	uint64_t pos = 0;
	while (pos <= limit) {
		uint8_t b = buf[pos];
		if (b != 0xE8 && b != 0xE9) {
			++pos;
			continue;
		}
		const uint32_t offset = (uint32_t)pos - prev_pos;
		prev_pos = (uint32_t)pos;
		if (offset > 5) {
			prev_mask = 0;
		} else {
			for (uint32_t i = 0; i < offset; ++i) {
				prev_mask &= 0x77;
				prev_mask <<= 1;
			}
		}
		b = buf[pos + 4];
		if ((b == 0 || b == 0xFF) && (prev_mask >> 1) <= 4 && (prev_mask >> 1) != 3) {
			uint32_t src = ((uint32_t)b << 24) | ((uint32_t)buf[pos + 3] << 16)
					| ((uint32_t)buf[pos + 2] << 8) | buf[pos + 1];
			uint32_t dest;
			for (;;) {
				dest = src + ((uint32_t)pos + 5);
				if (prev_mask == 0)
					break;
				const uint32_t i = mask_to_bit[prev_mask >> 1];
				b = (uint8_t)(dest >> (24 - i * 8));
				if (!(b == 0 || b == 0xFF))
					break;
				src = dest ^ ((1u << (32 - i * 8)) - 1);
			}
			buf[pos + 4] = (uint8_t)(~(((dest >> 24) & 1) - 1));
			buf[pos + 3] = (uint8_t)(dest >> 16);
			buf[pos + 2] = (uint8_t)(dest >> 8);
			buf[pos + 1] = (uint8_t)dest;
			pos += 5;
			prev_mask = 0;
		} else {
			++pos;
			prev_mask |= 1;
			if (b == 0 || b == 0xFF)
				prev_mask |= 0x10;
		}
	}
}
