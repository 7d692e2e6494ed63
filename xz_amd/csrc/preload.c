/*
 * preload.c -- LD_PRELOAD interposer: lets an unmodified liblzma client (the `xz` binary itself,
 * src/xz/coder.c:834-837) run its multi-threaded .xz encoder on the GPU.
 *
 *     LD_PRELOAD=/path/to/libxz_amd_preload.so xz -T4 -6 file
 *
 * Defines the liblzma entry points the MT encoder shares with every other coder and routes per
 * stream (SURVEY.md 8b "What calls it", option 2):
 *   lzma_stream_encoder_mt  -> libxz_amd.so (falls back to the real liblzma if there is no GPU or the
 *                              options are outside the device path)
 *   lzma_code / lzma_end / lzma_get_progress / lzma_filters_update
 *                           -> libxz_amd.so for streams it created (tagged lzma_internal), the real
 *                              liblzma (dlsym RTLD_NEXT) for all others (decoders, single-threaded
 *                              encoder, ...)
 * libxz_amd.so is opened RTLD_LOCAL so its own lzma_* exports never enter the global scope.
 * Environment: XZ_AMD_DISABLE=1 bypasses the GPU, XZ_AMD_VERBOSE=1 reports the routing on stderr.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/xz_amd_lzma.h"

#define XZAMD_MAGIC 0x585A414D44474655ull

typedef lzma_ret (*enc_mt_fn)(lzma_stream *, const lzma_mt *);
typedef lzma_ret (*code_fn)(lzma_stream *, lzma_action);
typedef void (*end_fn)(lzma_stream *);
typedef void (*progress_fn)(lzma_stream *, uint64_t *, uint64_t *);
typedef lzma_ret (*fupd_fn)(lzma_stream *, const lzma_filter *);

static struct {
	int tried;
	void *h;
	enc_mt_fn enc_mt;
	code_fn code;
	end_fn end;
	progress_fn progress;
	fupd_fn fupd;
} gpu;

static int verbose(void) { const char *v = getenv("XZ_AMD_VERBOSE"); return v && *v && *v != '0'; }

static void gpu_load(void)
{
	if (gpu.tried)
		return;
	gpu.tried = 1;
	const char *dis = getenv("XZ_AMD_DISABLE");
	if (dis && *dis && *dis != '0')
		return;
	/* libxz_amd.so sits next to this file's shared object unless XZ_AMD_LIB says otherwise */
	char path[4096];
	const char *lib = getenv("XZ_AMD_LIB");
	if (!lib) {
		Dl_info di;
		if (!dladdr((void *)&gpu_load, &di) || !di.dli_fname)
			return;
		snprintf(path, sizeof(path), "%s", di.dli_fname);
		char *slash = strrchr(path, '/');
		if (!slash)
			return;
		snprintf(slash + 1, sizeof(path) - (size_t)(slash + 1 - path), "libxz_amd.so");
		lib = path;
	}
	void *h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
	if (!h) {
		if (verbose()) fprintf(stderr, "xz_amd preload: cannot open %s: %s\n", lib, dlerror());
		return;
	}
	gpu.enc_mt = (enc_mt_fn)dlsym(h, "lzma_stream_encoder_mt");
	gpu.code = (code_fn)dlsym(h, "lzma_code");
	gpu.end = (end_fn)dlsym(h, "lzma_end");
	gpu.progress = (progress_fn)dlsym(h, "lzma_get_progress");
	gpu.fupd = (fupd_fn)dlsym(h, "lzma_filters_update");
	if (gpu.enc_mt && gpu.code && gpu.end && gpu.progress)
		gpu.h = h;
	else
		dlclose(h);
}

/* the first 8 bytes of a foreign lzma_internal are a pointer (lzma_next_coder.coder), never the tag */
static int is_ours(const lzma_stream *strm)
{
	if (!strm || !strm->internal)
		return 0;
	uint64_t tag;
	memcpy(&tag, strm->internal, sizeof(tag));
	return tag == XZAMD_MAGIC;
}

lzma_ret lzma_stream_encoder_mt(lzma_stream *strm, const lzma_mt *options)
{
	static enc_mt_fn real;
	if (!real)
		real = (enc_mt_fn)dlsym(RTLD_NEXT, "lzma_stream_encoder_mt");
	gpu_load();
	if (gpu.h) {
		if (strm && strm->internal && !is_ours(strm)) {
			/* re-initialising a stream that belongs to the real library: let it free its coder */
			static end_fn real_end;
			if (!real_end) real_end = (end_fn)dlsym(RTLD_NEXT, "lzma_end");
			if (real_end) real_end(strm);
		}
		const lzma_ret r = gpu.enc_mt(strm, options);
		if (r == LZMA_OK) {
			if (verbose()) fprintf(stderr, "xz_amd preload: lzma_stream_encoder_mt -> GPU\n");
			return r;
		}
		if (verbose()) fprintf(stderr, "xz_amd preload: GPU encoder declined (lzma_ret %d), using liblzma\n", (int)r);
	}
	return real ? real(strm, options) : LZMA_PROG_ERROR;
}

lzma_ret lzma_code(lzma_stream *strm, lzma_action action)
{
	static code_fn real;
	if (is_ours(strm) && gpu.h)
		return gpu.code(strm, action);
	if (!real)
		real = (code_fn)dlsym(RTLD_NEXT, "lzma_code");
	return real ? real(strm, action) : LZMA_PROG_ERROR;
}

void lzma_end(lzma_stream *strm)
{
	static end_fn real;
	if (is_ours(strm) && gpu.h) {
		gpu.end(strm);
		return;
	}
	if (!real)
		real = (end_fn)dlsym(RTLD_NEXT, "lzma_end");
	if (real)
		real(strm);
}

void lzma_get_progress(lzma_stream *strm, uint64_t *progress_in, uint64_t *progress_out)
{
	static progress_fn real;
	if (is_ours(strm) && gpu.h) {
		gpu.progress(strm, progress_in, progress_out);
		return;
	}
	if (!real)
		real = (progress_fn)dlsym(RTLD_NEXT, "lzma_get_progress");
	if (real)
		real(strm, progress_in, progress_out);
}

lzma_ret lzma_filters_update(lzma_stream *strm, const lzma_filter *filters)
{
	static fupd_fn real;
	if (is_ours(strm) && gpu.h && gpu.fupd)
		return gpu.fupd(strm, filters);
	if (!real)
		real = (fupd_fn)dlsym(RTLD_NEXT, "lzma_filters_update");
	return real ? real(strm, filters) : LZMA_PROG_ERROR;
}
