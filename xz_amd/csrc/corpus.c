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
