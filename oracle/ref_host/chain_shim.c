/* chain_shim.c -- test infrastructure: gives the reference host objects an mm_chain_dp (mmpriv.h:65) that is computed
 * by the repo's CPU oracle, and optionally dumps every anchor list that reaches it (MM2O_DUMP=<file>): per call a
 * header {int64 n; int32 max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, is_cdna, n_segs; float gap_scale}
 * followed by n 16-byte anchors. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "minimap.h"
#include "mmpriv.h"
#include "kalloc.h"
#include "chain_oracle.h"
#include <time.h>

/* MM2O_TIME=1: wall time inside mm_chain_dp summed over the threads, by size class of the call (what a per-read GPU call has to beat) */
static long long g_tm_ns[8], g_tm_calls[8], g_tm_anchors[8];
static int g_tm_on = -1;
static void tm_report(void)
{
	static const char *cls[8] = { "<256", "<512", "<1024", "<2048", "<4096", "<8192", "<16384", ">=16384" };
	long long ns = 0, c = 0, a = 0;
	int k;
	for (k = 0; k < 8; ++k) {
		ns += g_tm_ns[k]; c += g_tm_calls[k]; a += g_tm_anchors[k];
		if (g_tm_calls[k]) fprintf(stderr, "[chain_shim] n %s: %lld calls, %lld anchors, %.3f s, %.1f us per call\n", cls[k], g_tm_calls[k], g_tm_anchors[k], g_tm_ns[k] * 1e-9, g_tm_ns[k] * 1e-3 / g_tm_calls[k]);
	}
	fprintf(stderr, "[chain_shim] CPU chaining: %lld calls, %lld anchors, %.3f s inside mm_chain_dp (summed over threads), %.1f us per call\n", c, a, ns * 1e-9, c ? ns * 1e-3 / c : 0.0);
}
static int tm_class(int64_t n) { int k = 0; while (k < 7 && n >= (256LL << k)) ++k; return k; }

mm128_t *mm_chain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc, float gap_scale,
                     int is_cdna, int n_segs, int64_t n, mm128_t *a, int *n_u_, uint64_t **_u, void *km, int tid)
{
	mm2o_params_t par = { max_dist_x, max_dist_y, bw, max_skip, max_iter, gap_scale, is_cdna, n_segs };
	uint64_t *u = 0;
	mm2o_anchor_t *b = 0;
	int64_t n_b = 0;
	int32_t n_u;
	mm128_t *ret = 0;
	const char *dump = getenv("MM2O_DUMP");
	struct timespec ts0, ts1;
	(void)tid;
	if (g_tm_on < 0) { g_tm_on = getenv("MM2O_TIME") != 0; if (g_tm_on) atexit(tm_report); }
	if (g_tm_on) clock_gettime(CLOCK_MONOTONIC, &ts0);
	if (_u) *_u = 0, *n_u_ = 0;
	if ((n == 0 || a == 0) && dump && getenv("MM2O_DUMP_ALL")) {      /* a record for calls without anchors too (all-vs-all fixtures) */
		FILE *fp = fopen(dump, "ab");
		int64_t zero = 0;
		int32_t h[9] = { max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, is_cdna, n_segs };
		fwrite(&zero, 8, 1, fp); fwrite(h, 4, 9, fp); fwrite(&gap_scale, 4, 1, fp);
		fclose(fp);
	}
	if (n == 0 || a == 0) { kfree(km, a); return 0; }
	if (dump) {
		FILE *fp = fopen(dump, "ab");
		int32_t h[9] = { max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, is_cdna, n_segs };
		fwrite(&n, 8, 1, fp); fwrite(h, 4, 9, fp); fwrite(&gap_scale, 4, 1, fp); fwrite(a, 16, (size_t)n, fp);
		fclose(fp);
	}
	n_u = mm2o_mm_chain_dp(&par, min_cnt, min_sc, n, (const mm2o_anchor_t *)a, &u, &b, &n_b);
	kfree(km, a);                                         /* chain.c:421: the callee owns a */
	if (n_u > 0) {
		uint64_t *uk = (uint64_t *)kmalloc(km, (size_t)n_u * 8);
		ret = (mm128_t *)kmalloc(km, (size_t)n_b * sizeof(mm128_t));
		memcpy(uk, u, (size_t)n_u * 8);
		memcpy(ret, b, (size_t)n_b * sizeof(mm128_t));
		*n_u_ = n_u, *_u = uk;
	}
	free(u); free(b);
	if (g_tm_on) {
		const int k = tm_class(n);
		clock_gettime(CLOCK_MONOTONIC, &ts1);
		__sync_fetch_and_add(&g_tm_ns[k], (ts1.tv_sec - ts0.tv_sec) * 1000000000LL + (ts1.tv_nsec - ts0.tv_nsec));
		__sync_fetch_and_add(&g_tm_calls[k], 1); __sync_fetch_and_add(&g_tm_anchors[k], n);
	}
	return ret;
}
