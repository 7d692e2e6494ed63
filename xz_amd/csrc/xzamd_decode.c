/*
 * xzamd_decode.c -- host side of the device .xz Stream decoder: container parsing and
 * launch of the Block-parallel LZMA2 decode kernels (lzma_decode.hip).
 *
 * Restates, for a whole single-Stream .xz file resident in HBM, the checks of the reference's decoder chain:
 *   common/stream_flags_decoder.c:30-82     Stream Header / Footer (magic, flags, CRC32, Backward Size)
 *   common/index_decoder.c / index_hash.c   Index (indicator, record count, records, padding, CRC32;
 *                                           sizes must match the Blocks)
 *   common/block_header_decoder.c:17-125    Block Header (size, flags, VLIs, filter flags, padding, CRC32)
 *   common/block_decoder.c:47-230           Compressed / Uncompressed Size vs the header, Block Padding, Check
 *   lzma/lzma2_decoder.c, lzma_decoder.c    -> k_dec_scan / k_dec_units
 *   check/crc32_fast.c, crc64_fast.c        -> the encoder's k_crc_strips / k_crc_fold over the decoded bytes
 * Scheduling model of common/stream_decoder_mt.c: independent Blocks in parallel; with the original data at
 * hand (verification) every state-resetting chunk chain is its own unit.
 * Supported: one Stream, filter chain {LZMA2}, checks none / CRC32 / CRC64 (SHA-256 is skipped, not verified).
 */
#include "xzamd_internal.h"
#include "kernels_api.h"

#include <stdlib.h>
#include <string.h>

#define FORMAT_ERROR 7      /* LZMA_FORMAT_ERROR */

static int vli_get(const uint8_t *p, size_t n, size_t *pos, uint64_t *v)
{
	/* common/vli_decoder.c:16-86 (single call form) */
	uint64_t r = 0;
	for (unsigned i = 0; i < 9; ++i) {
		if (*pos >= n) return -1;
		const uint8_t b = p[(*pos)++];
		r |= (uint64_t)(b & 0x7F) << (7 * i);
		if (!(b & 0x80)) {
			if (b == 0 && i != 0) return -1;        /* non-minimal encoding */
			*v = r;
			return 0;
		}
	}
	return -1;
}

static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

#define FAILD(code, msg) do { rc = xzamd_ctx_fail_(c, (code), (msg)); goto done; } while (0)
#define HIPD(call, msg) do { if (call) FAILD(XZAMD_DEVICE_ERROR, msg); } while (0)

int xzamd_stream_decode_device(xzamd_ctx *c, const void *d_xz_, uint64_t xz_size, void *d_out_, uint64_t out_cap,
		uint64_t *out_size, const void *d_expected_, uint64_t expected_size, uint64_t *mismatches, uint64_t *nblocks_out,
		void *stream)
{
	if (!c || !d_xz_ || !out_size || (!d_out_ && out_cap))
		return XZAMD_PROG_ERROR;
	const uint8_t *d_xz = (const uint8_t *)d_xz_;
	uint8_t *d_out = (uint8_t *)d_out_;
	const uint8_t *d_expected = (const uint8_t *)d_expected_;
	void *st = stream ? stream : xzamd_ctx_stream_(c);
	int rc = XZAMD_OK;
	uint8_t *index = NULL;
	xzamd_dec_block *hb = NULL;
	uint64_t *stored = NULL, *h_crc = NULL;
	uint8_t *stored32 = NULL;          /* SHA-256: the 32 stored bytes of every Block */
	void *d_sha = NULL;
	uint32_t *unit_first = NULL, *h_err = NULL;
	void *d_blocks = NULL, *d_units = NULL, *d_first = NULL, *d_lit = NULL, *d_misc = NULL, *d_strip = NULL, *d_crc = NULL;
	*out_size = 0;
	if (mismatches) *mismatches = 0;
	if (nblocks_out) *nblocks_out = 0;
	if (xzk_set_device(xzamd_ctx_device(c)))
		return xzamd_ctx_fail_(c, XZAMD_DEVICE_ERROR, "hipSetDevice");

	/* Stream Header + Footer */
	static const uint8_t magic[6] = { 0xFD, '7', 'z', 'X', 'Z', 0x00 };
	uint8_t hf[24];
	if (xz_size < 32 || (xz_size & 3))
		FAILD(FORMAT_ERROR, "not an .xz Stream (size)");
	HIPD(xzk_d2h(hf, d_xz, 12, st) || xzk_d2h(hf + 12, d_xz + xz_size - 12, 12, st) || xzk_sync(st), "d2h stream flags");
	if (memcmp(hf, magic, 6) != 0 || hf[22] != 'Y' || hf[23] != 'Z')
		FAILD(FORMAT_ERROR, "bad magic bytes");
	if (rd32(hf + 8) != xzamd_crc32_host_(hf + 6, 2) || rd32(hf + 12) != xzamd_crc32_host_(hf + 16, 6))
		FAILD(XZAMD_DATA_ERROR, "Stream Header / Footer CRC32");
	if (hf[6] != 0 || (hf[7] & 0xF0) || hf[6] != hf[20] || hf[7] != hf[21])
		FAILD(XZAMD_OPTIONS_ERROR, "unsupported or inconsistent Stream Flags");
	const int check = hf[7] & 0x0F;
	/* a Check that cannot be verified is reported, not skipped (the reference: LZMA_UNSUPPORTED_CHECK) */
	if (check != XZAMD_CHECK_NONE && check != XZAMD_CHECK_CRC32 && check != XZAMD_CHECK_CRC64 && check != XZAMD_CHECK_SHA256)
		FAILD(XZAMD_UNSUPPORTED_CHECK, "Check id the device decoder cannot verify");
	static const uint8_t check_sizes[16] = { 0, 4, 4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64 };
	const uint32_t csz = check_sizes[check];
	const uint64_t index_size = ((uint64_t)rd32(hf + 16) + 1) * 4;
	if (index_size + 24 > xz_size)
		FAILD(XZAMD_DATA_ERROR, "Backward Size");

	/* Index */
	index = (uint8_t *)malloc(index_size);
	if (!index) FAILD(XZAMD_MEM_ERROR, "malloc");
	HIPD(xzk_d2h(index, d_xz + xz_size - 12 - index_size, index_size, st) || xzk_sync(st), "d2h index");
	if (index[0] != 0x00 || rd32(index + index_size - 4) != xzamd_crc32_host_(index, index_size - 4))
		FAILD(XZAMD_DATA_ERROR, "Index indicator / CRC32");
	size_t ip = 1;
	uint64_t nb = 0;
	if (vli_get(index, index_size - 4, &ip, &nb) || nb > (index_size / 2))
		FAILD(XZAMD_DATA_ERROR, "Index record count");
	hb = (xzamd_dec_block *)calloc(nb ? nb : 1, sizeof(*hb));
	stored = (uint64_t *)calloc(nb ? nb : 1, 8);
	h_crc = (uint64_t *)calloc(nb ? nb : 1, 8);
	h_err = (uint32_t *)calloc(nb ? nb : 1, 4);
	stored32 = (uint8_t *)calloc(nb ? nb : 1, 32);
	if (!stored32) FAILD(XZAMD_MEM_ERROR, "malloc");
	unit_first = (uint32_t *)calloc(nb + 1, 4);
	if (!hb || !stored || !h_crc || !h_err || !unit_first) FAILD(XZAMD_MEM_ERROR, "malloc");
	uint64_t pos = 12, utotal = 0, max_usize = 0;
	for (uint64_t b = 0; b < nb; ++b) {
		uint64_t unpadded = 0, usize = 0;
		if (vli_get(index, index_size - 4, &ip, &unpadded) || vli_get(index, index_size - 4, &ip, &usize)
				|| unpadded < 5 + csz || unpadded > (1ull << 62))
			FAILD(XZAMD_DATA_ERROR, "Index record");
		/* Block Header */
		uint8_t bh[1024];
		if (pos + 8 > xz_size - 12 - index_size)
			FAILD(XZAMD_DATA_ERROR, "Block beyond the Index");
		const uint64_t avail_h = xz_size - 12 - index_size - pos;
		HIPD(xzk_d2h(bh, d_xz + pos, avail_h < 64 ? avail_h : 64, st) || xzk_sync(st), "d2h block header");
		if (bh[0] == 0)
			FAILD(XZAMD_DATA_ERROR, "Index indicator where a Block Header was expected");
		const uint32_t hs = ((uint32_t)bh[0] + 1) * 4;
		if (hs > avail_h)
			FAILD(XZAMD_DATA_ERROR, "Block Header beyond the Index");
		if (hs > 64)
			HIPD(xzk_d2h(bh, d_xz + pos, hs, st) || xzk_sync(st), "d2h block header");
		if (rd32(bh + hs - 4) != xzamd_crc32_host_(bh, hs - 4))
			FAILD(XZAMD_DATA_ERROR, "Block Header CRC32");
		if (bh[1] & 0x3C)
			FAILD(XZAMD_OPTIONS_ERROR, "reserved Block Flags");
		if ((bh[1] & 3) != 0)
			FAILD(XZAMD_OPTIONS_ERROR, "only the {LZMA2} filter chain is decoded on the device");
		size_t hp = 2;
		uint64_t h_csize = UINT64_MAX, h_usize = UINT64_MAX, fid = 0, fps = 0;
		if ((bh[1] & 0x40) && vli_get(bh, hs - 4, &hp, &h_csize)) FAILD(XZAMD_DATA_ERROR, "Block Header Compressed Size");
		if ((bh[1] & 0x80) && vli_get(bh, hs - 4, &hp, &h_usize)) FAILD(XZAMD_DATA_ERROR, "Block Header Uncompressed Size");
		if (vli_get(bh, hs - 4, &hp, &fid) || vli_get(bh, hs - 4, &hp, &fps))
			FAILD(XZAMD_DATA_ERROR, "Filter Flags");
		if (fid != 0x21 || fps != 1 || hp >= hs - 4)
			FAILD(XZAMD_OPTIONS_ERROR, "only the {LZMA2} filter chain is decoded on the device");
		const uint32_t db = bh[hp++];
		if (db > 40) FAILD(XZAMD_OPTIONS_ERROR, "LZMA2 dictionary size byte");
		const uint32_t dict = db == 40 ? 0xFFFFFFFFu : ((2u | (db & 1u)) << (db / 2 + 11));
		for (; hp < hs - 4; ++hp)
			if (bh[hp] != 0) FAILD(XZAMD_OPTIONS_ERROR, "Block Header padding");
		if (unpadded < hs + csz)
			FAILD(XZAMD_DATA_ERROR, "Unpadded Size smaller than its header");
		const uint64_t csize = unpadded - hs - csz;
		if ((h_csize != UINT64_MAX && h_csize != csize) || (h_usize != UINT64_MAX && h_usize != usize) || csize == 0)
			FAILD(XZAMD_DATA_ERROR, "Block Header sizes differ from the Index");
		const uint64_t padded = (unpadded + 3) & ~3ull;
		if (pos + padded > xz_size - 12 - index_size)
			FAILD(XZAMD_DATA_ERROR, "Block beyond the Index");
		hb[b].cpos = pos + hs;
		hb[b].csize = csize;
		hb[b].upos = utotal;
		hb[b].usize = usize;
		hb[b].dict_size = dict;
		/* Block Padding must be zero; the stored Check */
		uint8_t tail[3 + 64];
		const uint32_t padn = (uint32_t)(padded - unpadded);
		HIPD(xzk_d2h(tail, d_xz + pos + hs + csize, padn + csz, st) || xzk_sync(st), "d2h check");
		for (uint32_t i = 0; i < padn; ++i)
			if (tail[i] != 0) FAILD(XZAMD_DATA_ERROR, "Block Padding");
		uint64_t sv = 0;
		for (uint32_t i = 0; i < csz && i < 8; ++i) sv |= (uint64_t)tail[padn + i] << (8 * i);
		stored[b] = sv;
		if (csz == 32) memcpy(stored32 + 32 * b, tail + padn, 32);
		pos += padded;
		utotal += usize;
		if (usize > max_usize) max_usize = usize;
		if (usize >= (1ull << 31) || csize >= (1ull << 32))
			FAILD(XZAMD_OPTIONS_ERROR, "Block too large for the device decoder");
	}
	for (; ip < index_size - 4; ++ip)
		if (index[ip] != 0) FAILD(XZAMD_DATA_ERROR, "Index Padding");
	if (pos != xz_size - 12 - index_size)
		FAILD(XZAMD_DATA_ERROR, "Blocks do not end where the Index starts");
	if (nblocks_out) *nblocks_out = nb;
	*out_size = utotal;
	if (utotal > out_cap)
		FAILD(XZAMD_BUF_ERROR, "output buffer too small");
	if (d_expected && expected_size != utotal)
		FAILD(XZAMD_DATA_ERROR, "the Stream's uncompressed size differs from the size of the original given for verification");
	if (nb == 0 || utotal == 0) {
		if (nb != 0) {
			/* Blocks of zero bytes still carry a chunk chain: decode it below */
		} else goto done;
	}

	/* unit scan + decode */
	{
		const uint32_t nbk = (uint32_t)nb;
		int split = d_expected != NULL;
		uint32_t units_cap = split ? (uint32_t)(max_usize / 4096 + 8) : 1;
		if ((uint64_t)units_cap * nb > (1ull << 27)) { split = 0; units_cap = 1; }
		uint32_t waves = xzamd_ctx_wave_slots_(c);
		HIPD(xzk_malloc(&d_blocks, nb * sizeof(xzamd_dec_block)), "hipMalloc");
		HIPD(xzk_malloc(&d_misc, 4096 + 4 * nb), "hipMalloc");
		HIPD(xzk_malloc(&d_first, 4 * (nb + 1)), "hipMalloc");
	rescan:
		HIPD(xzk_malloc(&d_units, (uint64_t)units_cap * nb * sizeof(xzamd_dec_unit)), "hipMalloc");
		HIPD(xzk_h2d(d_blocks, hb, nb * sizeof(xzamd_dec_block), st), "h2d blocks");
		HIPD(xzk_dec_scan(d_xz, (xzamd_dec_block *)d_blocks, nbk, (xzamd_dec_unit *)d_units, units_cap, split, st), "scan launch");
		HIPD(xzk_d2h(hb, d_blocks, nb * sizeof(xzamd_dec_block), st) || xzk_sync(st), "d2h blocks");
		uint32_t total_units = 0;
		int overflow = 0;
		for (uint64_t b = 0; b < nb; ++b) {
			if (hb[b].error == 11) overflow = 1;
			else if (hb[b].error) FAILD(XZAMD_DATA_ERROR, "LZMA2 chunk grammar");
			unit_first[b] = total_units;
			total_units += hb[b].nunits;
		}
		unit_first[nb] = total_units;
		if (overflow) {
			/* more state resets than anticipated: one unit per Block, compared afterwards all the same */
			xzk_free(d_units); d_units = NULL;
			for (uint64_t b = 0; b < nb; ++b) { hb[b].error = 0; hb[b].nunits = 0; }
			split = 0; units_cap = 1;
			goto rescan;
		}
		const uint8_t *hist = split ? d_expected : NULL;
		const uint32_t work = hist ? total_units : nbk;
		if (waves > work) waves = work ? work : 1;
		HIPD(xzk_malloc(&d_lit, (uint64_t)waves * (0x300ull << 4) * 2), "hipMalloc");
		HIPD(xzk_memset(d_misc, 0, 4096 + 4 * nb, st), "memset");
		HIPD(xzk_h2d(d_first, unit_first, 4 * (nb + 1), st), "h2d");
		uint32_t *d_counter = (uint32_t *)d_misc;
		unsigned long long *d_mism = (unsigned long long *)((uint8_t *)d_misc + 64);
		uint32_t *d_berr = (uint32_t *)((uint8_t *)d_misc + 4096);
		HIPD(xzk_dec_units(d_xz, (const xzamd_dec_block *)d_blocks, nbk, (const xzamd_dec_unit *)d_units, units_cap,
				(const uint32_t *)d_first, total_units, d_out, hist, (uint16_t *)d_lit, waves, d_counter, d_berr, st), "decode launch");
		HIPD(xzk_d2h(h_err, d_berr, 4 * nb, st) || xzk_sync(st), "decode");
		for (uint64_t b = 0; b < nb; ++b)
			if (h_err[b]) FAILD(XZAMD_DATA_ERROR, "LZMA2 data (range coder / distances / chunk sizes)");
		/* Block checks over the decoded bytes */
		if (check == XZAMD_CHECK_CRC32 || check == XZAMD_CHECK_CRC64) {
			const uint32_t strip = 4096;
			int uniform = 1;
			for (uint64_t b = 0; b + 1 < nb; ++b)
				if (hb[b].usize != hb[0].usize) uniform = 0;
			if (nb > 1 && hb[nb - 1].usize > hb[0].usize) uniform = 0;
			const uint64_t bs0 = hb[0].usize ? hb[0].usize : 1;
			const uint64_t spb = (max_usize + strip - 1) / strip + 1;
			HIPD(xzk_malloc(&d_strip, 8 * spb * nb + 64), "hipMalloc");
			HIPD(xzk_malloc(&d_crc, 8 * nb + 64), "hipMalloc");
			if (uniform && utotal < (1ull << 31)) {
				HIPD(xzk_crc_blocks(d_out, (uint32_t)utotal, (uint32_t)bs0, nbk, strip, check == XZAMD_CHECK_CRC32,
						(uint64_t *)d_strip, (uint64_t *)d_crc, st), "crc launch");
			} else {
				for (uint64_t b = 0; b < nb; ++b) {
					if (hb[b].usize == 0) { HIPD(xzk_memset((uint8_t *)d_crc + 8 * b, 0, 8, st), "memset"); continue; }
					HIPD(xzk_crc_blocks(d_out + hb[b].upos, (uint32_t)hb[b].usize, (uint32_t)hb[b].usize, 1, strip,
							check == XZAMD_CHECK_CRC32, (uint64_t *)d_strip, (uint64_t *)d_crc + b, st), "crc launch");
				}
			}
			HIPD(xzk_d2h(h_crc, d_crc, 8 * nb, st) || xzk_sync(st), "d2h crc");
			for (uint64_t b = 0; b < nb; ++b)
				if (h_crc[b] != stored[b]) FAILD(XZAMD_DATA_ERROR, "Block Check mismatch");
		}
		if (check == XZAMD_CHECK_SHA256) {
			/* check/sha256.c over the decoded bytes: one hash per Block (serial by nature); equally long Blocks in one
			 * launch, else one launch each */
			int uniform = 1;
			for (uint64_t b = 0; b + 1 < nb; ++b)
				if (hb[b].usize != hb[0].usize) uniform = 0;
			if (nb > 1 && hb[nb - 1].usize > hb[0].usize) uniform = 0;
			HIPD(xzk_malloc(&d_sha, 32 * nb + 64), "hipMalloc");
			if (uniform && utotal < (1ull << 31) && hb[0].usize) {
				HIPD(xzk_sha256_blocks(d_out, (uint32_t)utotal, (uint32_t)hb[0].usize, nbk, (uint8_t *)d_sha, st), "sha256 launch");
			} else {
				for (uint64_t b = 0; b < nb; ++b)
					HIPD(xzk_sha256_blocks(d_out + hb[b].upos, (uint32_t)hb[b].usize, hb[b].usize ? (uint32_t)hb[b].usize : 1u, 1,
							(uint8_t *)d_sha + 32 * b, st), "sha256 launch");
			}
			uint8_t *h_sha = (uint8_t *)malloc(32 * nb);
			if (!h_sha) FAILD(XZAMD_MEM_ERROR, "malloc");
			int bad = xzk_d2h(h_sha, d_sha, 32 * nb, st) || xzk_sync(st) ? -1 : memcmp(h_sha, stored32, 32 * nb) != 0;
			free(h_sha);
			if (bad < 0) FAILD(XZAMD_DEVICE_ERROR, "d2h sha256");
			if (bad) FAILD(XZAMD_DATA_ERROR, "Block Check mismatch (SHA-256)");
		}
		if (d_expected) {
			unsigned long long mm = 0;
			HIPD(xzk_dec_compare(d_out, d_expected, utotal, d_mism, st), "compare launch");
			HIPD(xzk_d2h(&mm, d_mism, 8, st) || xzk_sync(st), "d2h compare");
			if (mismatches) *mismatches = mm;
			if (mm) FAILD(XZAMD_DATA_ERROR, "decoded bytes differ from the original");
		}
	}
done:
	xzk_sync(st);
	if (d_blocks) xzk_free(d_blocks);
	if (d_units) xzk_free(d_units);
	if (d_first) xzk_free(d_first);
	if (d_lit) xzk_free(d_lit);
	if (d_misc) xzk_free(d_misc);
	if (d_strip) xzk_free(d_strip);
	if (d_crc) xzk_free(d_crc);
	if (d_sha) xzk_free(d_sha);
	free(stored32);
	free(index); free(hb); free(stored); free(h_crc); free(h_err); free(unit_first);
	return rc;
}
