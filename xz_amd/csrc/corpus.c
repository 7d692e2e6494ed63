/*
 * corpus.c -- seeded synthetic inputs for bench.py and the tests (host memory).
 * There is no network on the GPU box, so enwik/Linux tarballs cannot be used
 * (SURVEY.md 8d); these generators are deterministic functions of (n, seed).
 */
#include "../../include/xz_amd.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- lorem-LCG: tests/create_compress_files.c:110-152 (write_text) run on ---- */
static const char *const lorem[69] = {
	"Lorem", "ipsum", "dolor", "sit", "amet,", "consectetur", "adipisicing", "elit,", "sed", "do",
	"eiusmod", "tempor", "incididunt", "ut", "labore", "et", "dolore", "magna", "aliqua.", "Ut",
	"enim", "ad", "minim", "veniam,", "quis", "nostrud", "exercitation", "ullamco", "laboris", "nisi",
	"ut", "aliquip", "ex", "ea", "commodo", "consequat.", "Duis", "aute", "irure", "dolor", "in",
	"reprehenderit", "in", "voluptate", "velit", "esse", "cillum", "dolore", "eu", "fugiat", "nulla",
	"pariatur.", "Excepteur", "sint", "occaecat", "cupidatat", "non", "proident,", "sunt", "in",
	"culpa", "qui", "officia", "deserunt", "mollit", "anim", "id", "est", "laborum."
};

typedef struct { uint8_t *out; uint64_t n, pos; } sink;

static int put(sink *s, const char *str)
{
	while (*str) {
		if (s->pos >= s->n)
			return 1;
		s->out[s->pos++] = (uint8_t)*str++;
	}
	return s->pos >= s->n;
}

void xzamd_corpus_lorem(uint8_t *out, uint64_t n)
{
	sink s = { out, n, 0 };
	for (int w = 0; w < 69; ++w) {
		if (put(&s, lorem[w]) || put(&s, " ")) return;
		if (w % 7 == 6 && put(&s, "\n")) return;
	}
	uint32_t x = 29;
	for (;;) {
		if (put(&s, "\n\n")) return;
		for (int w = 0; w < 69; ++w) {
			x = 101771u * x + 71777u;
			if (put(&s, lorem[x % 69]) || put(&s, " ")) return;
			if (w % 7 == 6 && put(&s, "\n")) return;
		}
	}
}

/* ---- "enwik-style" text: Zipf vocabulary + first-order phrase structure + wiki markup ----
 * Generated in independent 1 MiB segments (seeded by segment index) so any
 * number of host threads produces identical bytes. */
static inline uint64_t rng_next(uint64_t *s)
{
	/* splitmix64 */
	uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

#define VOCAB 32768
typedef struct {
	char words[VOCAB][14];
	uint8_t wlen[VOCAB];
	uint32_t zipf_cdf[VOCAB];   /* scaled to 2^32 */
} vocab;

static vocab *g_vocab;
static pthread_once_t vocab_once = PTHREAD_ONCE_INIT;

static void vocab_init(void)
{
	vocab *v = (vocab *)malloc(sizeof(vocab));
	uint64_t s = 0x5EEDC0DEull;
	static const char cons[] = "bcdfghjklmnprstvwz";
	static const char vow[] = "aeiou";
	for (int i = 0; i < VOCAB; ++i) {
		/* frequent words are short */
		int len = 2 + (int)(rng_next(&s) % 4);
		if (i > 64) len += (int)(rng_next(&s) % 4);
		if (i > 4096) len += (int)(rng_next(&s) % 3);
		for (int k = 0; k < len; ++k)
			v->words[i][k] = (k & 1) ? vow[rng_next(&s) % 5] : cons[rng_next(&s) % 18];
		v->words[i][len] = 0;
		v->wlen[i] = (uint8_t)len;
	}
	double tot = 0, acc = 0;
	for (int i = 0; i < VOCAB; ++i) tot += 1.0 / (double)(i + 2);
	for (int i = 0; i < VOCAB; ++i) {
		acc += (1.0 / (double)(i + 2)) / tot;
		double x = acc * 4294967296.0;
		v->zipf_cdf[i] = x >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)x;
	}
	v->zipf_cdf[VOCAB - 1] = 0xFFFFFFFFu;
	g_vocab = v;
}

static uint32_t zipf_draw(const vocab *v, uint64_t *s)
{
	const uint32_t r = (uint32_t)(rng_next(s) >> 32);
	uint32_t lo = 0, hi = VOCAB - 1;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) / 2;
		if (v->zipf_cdf[mid] < r) lo = mid + 1; else hi = mid;
	}
	return lo;
}

static void gen_segment(uint8_t *out, uint64_t len, uint64_t seed, uint64_t seg)
{
	const vocab *v = g_vocab;
	uint64_t s = seed * 0x9E3779B97F4A7C15ull + seg * 0xD1B54A32D192ED03ull + 1;
	sink k = { out, len, 0 };
	uint32_t prev = zipf_draw(v, &s);
	uint32_t article = 0;
	char tmp[64];
	while (k.pos < k.n) {
		/* page header now and then */
		if ((rng_next(&s) & 255) == 0 || k.pos == 0) {
			article = (uint32_t)(rng_next(&s) % 1000000);
			if (put(&k, "  <page>\n    <title>")) break;
			if (put(&k, v->words[zipf_draw(v, &s)]) || put(&k, " ") || put(&k, v->words[zipf_draw(v, &s)])) break;
			if (put(&k, "</title>\n    <id>")) break;
			int t = 0; uint32_t a = article; char d[12]; do { d[t++] = (char)('0' + a % 10); a /= 10; } while (a);
			for (int i = 0; i < t; ++i) tmp[i] = d[t - 1 - i];
			tmp[t] = 0;
			if (put(&k, tmp) || put(&k, "</id>\n    <text xml:space=\"preserve\">")) break;
		}
		/* a paragraph of sentences */
		const int sentences = 1 + (int)(rng_next(&s) % 6);
		if ((rng_next(&s) & 15) == 0) {
			if (put(&k, "\n== ") || put(&k, v->words[zipf_draw(v, &s)]) || put(&k, " ==\n")) break;
		}
		int stop = 0;
		for (int sn = 0; sn < sentences && !stop; ++sn) {
			const int words = 4 + (int)(rng_next(&s) % 18);
			for (int w = 0; w < words && !stop; ++w) {
				/* first-order structure: with p=0.55 the next word is one of 4
				 * fixed successors of the previous word (gives repeated phrases) */
				uint32_t id;
				const uint64_t r = rng_next(&s);
				if ((r & 127) < 70)
					id = (uint32_t)((prev * 2654435761u + ((r >> 8) & 3) * 40503u) >> 7) % (VOCAB / 8)
						+ ((r >> 12) & 1) * (prev % 97);
				else
					id = zipf_draw(v, &s);
				id %= VOCAB;
				const uint64_t deco = rng_next(&s) & 63;
				if (deco == 0) stop |= put(&k, "[[");
				else if (deco == 1) stop |= put(&k, "''");
				if (w == 0 && v->words[id][0] >= 'a') {
					memcpy(tmp, v->words[id], v->wlen[id] + 1);
					tmp[0] = (char)(tmp[0] - 32);
					stop |= put(&k, tmp);
				} else {
					stop |= put(&k, v->words[id]);
				}
				if (deco == 0) stop |= put(&k, "]]");
				else if (deco == 1) stop |= put(&k, "''");
				else if (deco == 2) {
					int t = 0; uint32_t a = (uint32_t)(rng_next(&s) % 2100); char d[12];
					do { d[t++] = (char)('0' + a % 10); a /= 10; } while (a);
					tmp[0] = ' ';
					for (int i = 0; i < t; ++i) tmp[1 + i] = d[t - 1 - i];
					tmp[1 + t] = 0;
					stop |= put(&k, tmp);
				}
				stop |= put(&k, w + 1 == words ? ". " : ((rng_next(&s) & 15) == 0 ? ", " : " "));
				prev = id;
			}
		}
		if (put(&k, "\n\n")) break;
		if ((rng_next(&s) & 127) == 0)
			if (put(&k, "</text>\n  </page>\n")) break;
	}
}

typedef struct { uint8_t *out; uint64_t n, seed; uint64_t first, step; } job;

static void *worker(void *arg)
{
	job *j = (job *)arg;
	const uint64_t seg_size = 1u << 20;
	const uint64_t nseg = (j->n + seg_size - 1) / seg_size;
	for (uint64_t sgi = j->first; sgi < nseg; sgi += j->step) {
		const uint64_t off = sgi * seg_size;
		gen_segment(j->out + off, j->n - off < seg_size ? j->n - off : seg_size, j->seed, sgi);
	}
	return NULL;
}

void xzamd_corpus_text(uint8_t *out, uint64_t n, uint64_t seed, int threads)
{
	pthread_once(&vocab_once, vocab_init);
	if (threads < 1) threads = 1;
	if (threads > 256) threads = 256;
	pthread_t th[256];
	job jobs[256];
	for (int t = 0; t < threads; ++t) {
		jobs[t].out = out; jobs[t].n = n; jobs[t].seed = seed;
		jobs[t].first = (uint64_t)t; jobs[t].step = (uint64_t)threads;
		pthread_create(&th[t], NULL, worker, &jobs[t]);
	}
	for (int t = 0; t < threads; ++t)
		pthread_join(th[t], NULL);
}

/* ---- "concatenated tarballs" (SURVEY.md 8d, config C4): a ustar stream of the source trees present on
 * the box, cycled until n bytes are there.  roots = ':'-separated directories, walked depth first in
 * strcmp order of the entry names (deterministic for a given image); regular files only, symlinks are
 * not followed.  Cycle c >= 1 perturbs the file contents at the byte level (one alphanumeric byte per
 * 4 KiB replaced by another, seeded by (seed, c)) so that the cycles are not identical.
 * Returns the number of files of the first cycle, 0 if the roots hold no readable regular file. */
#include <dirent.h>
#include <stdio.h>
#include <sys/stat.h>

typedef struct { char **v; size_t n, cap; } strlist;

static int cmp_str(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

/* Appends to `files`; an allocation failure ends the walk early (the corpus is then built from what was listed). */
static int walk(const char *dir, strlist *files)
{
	DIR *d = opendir(dir);
	if (!d) return 0;
	strlist names = { NULL, 0, 0 };
	struct dirent *de;
	int oom = 0;
	while (!oom && (de = readdir(d)) != NULL) {
		if (!strcmp(de->d_name, ".") || !strcmp(de->d_name, "..")) continue;
		if (names.n == names.cap) {
			const size_t cap = names.cap ? names.cap * 2 : 64;
			char **v = (char **)realloc(names.v, cap * sizeof(char *));
			if (!v) { oom = 1; break; }
			names.v = v; names.cap = cap;
		}
		char *nm = strdup(de->d_name);
		if (!nm) { oom = 1; break; }
		names.v[names.n++] = nm;
	}
	closedir(d);
	if (names.n) qsort(names.v, names.n, sizeof(char *), cmp_str);
	for (size_t i = 0; i < names.n; ++i) {
		const size_t len = strlen(dir) + strlen(names.v[i]) + 2;
		char *path = oom ? NULL : (char *)malloc(len);
		if (!path) { oom = 1; free(names.v[i]); continue; }
		snprintf(path, len, "%s/%s", dir, names.v[i]);
		struct stat st;
		if (lstat(path, &st) == 0 && S_ISDIR(st.st_mode)) {
			oom |= walk(path, files);
			free(path);
		} else if (lstat(path, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
			if (files->n == files->cap) {
				const size_t cap = files->cap ? files->cap * 2 : 1024;
				char **v = (char **)realloc(files->v, cap * sizeof(char *));
				if (!v) { oom = 1; free(path); free(names.v[i]); continue; }
				files->v = v; files->cap = cap;
			}
			files->v[files->n++] = path;
		} else {
			free(path);
		}
		free(names.v[i]);
	}
	free(names.v);
	return oom;
}

static void tar_octal(uint8_t *dst, int width, uint64_t v)
{
	for (int i = width - 2; i >= 0; --i) { dst[i] = (uint8_t)('0' + (v & 7)); v >>= 3; }
	dst[width - 1] = 0;
}

static void tar_header(uint8_t h[512], const char *path, uint64_t size)
{
	memset(h, 0, 512);
	while (*path == '/') ++path;
	const size_t len = strlen(path);
	if (len <= 100) {
		memcpy(h, path, len);
	} else {
		/* ustar: split at a '/' into prefix (<= 155) and name (<= 100); else keep the tail */
		const char *cut = NULL;
		for (const char *s = path + (len > 255 ? len - 255 : 0); *s; ++s)
			if (*s == '/' && (size_t)(s - path) <= 155 && len - (size_t)(s - path) - 1 <= 100) { cut = s; break; }
		if (cut) {
			memcpy(h + 345, path, (size_t)(cut - path));
			memcpy(h, cut + 1, len - (size_t)(cut - path) - 1);
		} else {
			memcpy(h, path + len - 100, 100);
		}
	}
	tar_octal(h + 100, 8, 0644);
	tar_octal(h + 108, 8, 0);
	tar_octal(h + 116, 8, 0);
	tar_octal(h + 124, 12, size);
	tar_octal(h + 136, 12, 0);
	memset(h + 148, ' ', 8);
	h[156] = '0';
	memcpy(h + 257, "ustar", 6);
	h[263] = '0'; h[264] = '0';
	uint32_t sum = 0;
	for (int i = 0; i < 512; ++i) sum += h[i];
	tar_octal(h + 148, 7, sum);
	h[155] = ' ';
}

uint64_t xzamd_corpus_tar(uint8_t *out, uint64_t n, const char *roots, uint64_t seed)
{
	strlist files = { NULL, 0, 0 };
	char *r = strdup(roots ? roots : "");
	if (!r) return 0;
	for (char *tok = r, *next; tok && *tok; tok = next) {
		next = strchr(tok, ':');
		if (next) *next++ = 0;
		size_t l = strlen(tok);
		while (l > 1 && tok[l - 1] == '/') tok[--l] = 0;
		if (walk(tok, &files)) break;
	}
	free(r);
	const uint64_t nfiles = files.n;
	uint64_t pos = 0, made = 0;
	for (uint64_t cycle = 0; pos < n && nfiles; ++cycle) {
		const uint64_t before = pos;
		for (size_t f = 0; f < files.n && pos < n; ++f) {
			FILE *fp = fopen(files.v[f], "rb");
			if (!fp) continue;
			struct stat st;
			if (fstat(fileno(fp), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) { fclose(fp); continue; }
			const uint64_t size = (uint64_t)st.st_size;
			uint8_t h[512];
			tar_header(h, files.v[f], size);
			const uint64_t hn = n - pos < 512 ? n - pos : 512;
			memcpy(out + pos, h, hn);
			pos += hn;
			const uint64_t body = n - pos < size ? n - pos : size;
			const uint64_t got = body ? fread(out + pos, 1, body, fp) : 0;
			fclose(fp);
			if (got < body) memset(out + pos + got, 0, body - got);
			if (cycle) {
				uint64_t s = seed * 0x9E3779B97F4A7C15ull + cycle * 0xD1B54A32D192ED03ull + f;
				for (uint64_t o = 0; o + 4096 <= body; o += 4096) {
					const uint64_t rr = rng_next(&s);
					uint8_t *b = out + pos + o + (rr & 4095);
					if ((*b >= 'a' && *b <= 'z') || (*b >= 'A' && *b <= 'Z'))
						*b = (uint8_t)((*b & 0xE0) | (1 + ((rr >> 16) % 26)));
					else if (*b >= '0' && *b <= '9')
						*b = (uint8_t)('0' + (rr >> 16) % 10);
				}
			}
			pos += body;
			const uint64_t padn = (512 - (size & 511)) & 511;
			const uint64_t pn = n - pos < padn ? n - pos : padn;
			memset(out + pos, 0, pn);
			pos += pn;
			++made;
		}
		if (pos == before) break;       /* nothing readable: avoid spinning */
	}
	if (pos < n) memset(out + pos, 0, n - pos);
	for (size_t i = 0; i < files.n; ++i) free(files.v[i]);
	free(files.v);
	return made ? nfiles : 0;
}
