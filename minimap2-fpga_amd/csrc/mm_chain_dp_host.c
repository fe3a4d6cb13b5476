/*
 * mm_chain_dp_host.c -- host mirror of mm_chain_dp (kisarur/minimap2-fpga chain.c:29-423, prototype mmpriv.h:65).
 *
 * Same signature, same ownership (frees `a`, returns km-owned b[] and u[]), so a minimap2 host links this
 * object in place of chain.o.  The f[]/p[] DP -- the hot loop, chain.c:184-238 -- ALWAYS runs on the GPU via
 * mm2c_chain_task_host (stock CPU semantics incl. max_skip); there is no software DP in this file and no
 * HW/SW time model (chain.c:53-81 is an FPGA artefact).  The O(n) epilogue (v[], chain ends, backtrack, chain
 * order; chain.c:106-111, 348-422) runs on the calling thread.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mm2chain.h"

/* the host program's arena allocator (kalloc.h:14,17).  Weak: when the library is loaded without a minimap2
 * host (tests, bench) only km == NULL is accepted, which kalloc.c itself maps to malloc/free. */
extern void *kmalloc(void *km, size_t size) __attribute__((weak));
extern void kfree(void *km, void *ptr) __attribute__((weak));

static void *xalloc(void *km, size_t sz)
{
	if (kmalloc) return kmalloc(km, sz);
	if (km) { fprintf(stderr, "[mm2chain] km != NULL but the host's kmalloc is not linked\n"); exit(EXIT_FAILURE); }
	return malloc(sz ? sz : 1);
}
static void xfree(void *km, void *p)
{
	if (kfree) kfree(km, p);
	else free(p);
}

/* ---- in-place MSD byte radix sort with the klib pass structure (ksort.h:101-151; misc.c:155-159) ----
 * Order among equal keys is part of mm_chain_dp's observable output (chain.c:411), so the passes follow the
 * same rules: n <= 64 insertion sort; else cycle-leader distribution on the current byte, top byte first;
 * sub-buckets > 64 recurse on the next byte, 2..64 are insertion sorted. */
typedef struct { mm2c_anchor_t *lo, *hi; } span128_t;

static void ins128(mm2c_anchor_t *lo, mm2c_anchor_t *hi)
{
	mm2c_anchor_t *q, *r, key;
	for (q = lo + 1; q < hi; ++q) {
		if (q->x >= (q - 1)->x) continue;
		key = *q;
		for (r = q; r > lo && key.x < (r - 1)->x; --r) *r = *(r - 1);
		*r = key;
	}
}

static void msd128(mm2c_anchor_t *lo, mm2c_anchor_t *hi, int shift)
{
	span128_t bk[256];
	size_t hist[256] = {0};
	mm2c_anchor_t *q;
	int d;
	for (q = lo; q != hi; ++q) ++hist[(q->x >> shift) & 255];
	for (d = 0, q = lo; d < 256; ++d) { bk[d].lo = q; q += hist[d]; bk[d].hi = q; }
	for (d = 0; d < 256; ) {
		int dst;
		if (bk[d].lo == bk[d].hi) { ++d; continue; }
		dst = (int)((bk[d].lo->x >> shift) & 255);
		if (dst == d) { ++bk[d].lo; continue; }
		{
			mm2c_anchor_t hand = *bk[d].lo, next;
			do {
				next = *bk[dst].lo; *bk[dst].lo++ = hand; hand = next;
				dst = (int)((hand.x >> shift) & 255);
			} while (dst != d);
			*bk[d].lo++ = hand;
		}
	}
	if (shift == 0) return;
	shift = shift > 8 ? shift - 8 : 0;
	for (d = 0, q = lo; d < 256; ++d) {
		mm2c_anchor_t *e = bk[d].hi;
		if (e - q > 64) msd128(q, e, shift);
		else if (e - q > 1) ins128(q, e);
		q = e;
	}
}

static void sort128x(mm2c_anchor_t *a, size_t n) { if (n <= 64) ins128(a, a + n); else msd128(a, a + n, 56); }

/* keys of u[] are distinct (low word = anchor index), so any correct ascending sort equals radix_sort_64 (chain.c:368).  A read has a few hundred chain ends: runs of
 * 8 by insertion, then bottom-up merges through `buf` (room for n keys) -- a quarter of what qsort() with its compare callback took on the per-read path (round 5). */
static void sort_u64(uint64_t *u, int64_t n, uint64_t *buf)
{
	int64_t i, j, w;
	uint64_t *src = u, *dst = buf;
	for (i = 0; i < n; i += 8) {
		const int64_t e = i + 8 < n ? i + 8 : n;
		for (j = i + 1; j < e; ++j) {
			const uint64_t key = u[j];
			int64_t k = j;
			while (k > i && u[k - 1] > key) { u[k] = u[k - 1]; --k; }
			u[k] = key;
		}
	}
	for (w = 8; w < n; w <<= 1) {
		for (i = 0; i < n; i += 2 * w) {
			int64_t a = i, am = i + w < n ? i + w : n, b = am, bm = i + 2 * w < n ? i + 2 * w : n, o = i;
			while (a < am && b < bm) dst[o++] = src[b] < src[a] ? src[b++] : src[a++];
			while (a < am) dst[o++] = src[a++];
			while (b < bm) dst[o++] = src[b++];
		}
		{ uint64_t *t = src; src = dst; dst = t; }
	}
	if (src != u) memcpy(u, src, (size_t)n * 8);
}

/*
 * The O(n) epilogue of mm_chain_dp (chain.c:106-111, 348-422) on caller-provided storage: v[] from f[]/p[], chain ends, peak
 * search, sort by score, backtrack without re-using anchors, emission, chains ordered by the x of their first anchor.
 * f, p: the DP result (read only).  vt: scratch of 2n ints, 8-byte aligned (v[] then t[]).  u_out / b_out: room for n entries
 * each (a chain has >= 1 anchor).  tmp: scratch for 2n anchors.  Returns the number of chains; *n_b_out = anchors in b_out.
 */
static int32_t chain_epilogue(int min_cnt, int min_sc, int64_t n, const mm2c_anchor_t *a, const int32_t *f, const int32_t *p,
                              int32_t *vt, uint64_t *u_out, mm2c_anchor_t *b_out, mm2c_anchor_t *tmp, int64_t *n_b_out)
{
	int32_t *v = vt, *t = vt + n, n_u, n_v, k;
	int64_t i, j;
	uint64_t *u = u_out;
	*n_b_out = 0;
	memset(t, 0, (size_t)n * 4);
	for (i = 0; i < n; ++i) {                                                /* chain.c:106-111, and in the same sweep the child marks of chain.c:350 */
		const int32_t pi = p[i];
		if (pi >= 0) { const int32_t vp = v[pi]; v[i] = vp > f[i] ? vp : f[i]; t[pi] = 1; }
		else v[i] = f[i];
	}
	/* chain ends (chain.c:349-367) */
	for (i = 0, n_u = 0; i < n; ++i) {
		if (t[i] != 0 || v[i] < min_sc) continue;
		for (j = i; j >= 0 && f[j] < v[j]; ) j = p[j];
		if (j < 0) j = i;
		u[n_u++] = (uint64_t)f[j] << 32 | (uint64_t)j;
	}
	if (n_u == 0) return 0;
	sort_u64(u, n_u, (uint64_t *)tmp);                                       /* chain.c:368 (tmp: 2 n anchors of scratch, free until the emission below) */
	for (i = 0; i < n_u >> 1; ++i) { uint64_t s = u[i]; u[i] = u[n_u - i - 1]; u[n_u - i - 1] = s; }
	/* backtrack (chain.c:375-390); v[] is reused as the list of chained anchors */
	memset(t, 0, (size_t)n * 4);
	for (i = 0, n_v = 0, k = 0; i < n_u; ++i) {
		const int32_t n_v0 = n_v, k0 = k, peak = (int32_t)(u[i] >> 32);
		int32_t len;
		j = (int32_t)u[i];
		do { v[n_v++] = (int32_t)j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
		len = n_v - n_v0;
		if (j < 0) { if (len >= min_cnt) u[k++] = (uint64_t)peak << 32 | (uint32_t)len; }
		else if (peak - f[j] >= min_sc) { if (len >= min_cnt) u[k++] = (uint64_t)(peak - f[j]) << 32 | (uint32_t)len; }
		if (k0 == k) n_v = n_v0;
	}
	n_u = k;
	if (n_u == 0) return 0;
	/* order the chains by the x of their first anchor (chain.c:406-420), then emit each straight into its final place (chain.c:397-402): v[] holds the chained anchors of
	 * chain i in backtrack order (last anchor first), so its first anchor is the last entry of its stretch.  (Round 5: one copy of the chained anchors instead of two.) */
	{
		mm2c_anchor_t *w = tmp + n;                                          /* second half of tmp: one sort record per chain */
		uint64_t *u2 = (uint64_t *)tmp;                                      /* first half: the chains' (score, count) in their final order */
		int64_t n_b = 0;
		for (i = 0, k = 0; i < n_u; ++i) { const int32_t ni = (int32_t)u[i]; w[i].x = a[v[k + ni - 1]].x; w[i].y = (uint64_t)k << 32 | (uint64_t)i; k += ni; }
		sort128x(w, (size_t)n_u);
		for (i = 0; i < n_u; ++i) {
			const int32_t src = (int32_t)w[i].y, cnt = (int32_t)u[src];
			const int32_t *vi = v + (w[i].y >> 32) + cnt - 1;
			u2[i] = u[src];
			for (j = 0; j < cnt; ++j) b_out[n_b + j] = a[vi[-j]];
			n_b += cnt;
		}
		memcpy(u, u2, (size_t)n_u * 8);
		*n_b_out = n_b;
	}
	return n_u;
}

mm2c_anchor_t *mm_chain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                           float gap_scale, int is_cdna, int n_segs, int64_t n, mm2c_anchor_t *a, int *n_u_, uint64_t **_u,
                           void *km, int tid)
{
	int32_t *f, *p, *vt, n_u;
	int64_t i, n_b = 0;
	uint64_t *u, sum_qspan = 0;
	float avg_qspan_scaled;
	mm2c_anchor_t *b, *tmp;
	mm2c_params_t par;

	if (_u) *_u = 0, *n_u_ = 0;
	if (n == 0 || a == 0) { xfree(km, a); return 0; }                       /* chain.c:37-41 */
	f = (int32_t *)xalloc(km, (size_t)n * 4); p = (int32_t *)xalloc(km, (size_t)n * 4);
	vt = (int32_t *)xalloc(km, (size_t)n * 8);

	for (i = 0; i < n; ++i) sum_qspan += a[i].y >> 32 & 0xff;                /* chain.c:48-49 */
	avg_qspan_scaled = (float)(.01 * (float)sum_qspan / n);

	par.max_dist_x = max_dist_x; par.max_dist_y = max_dist_y; par.bw = bw;
	par.max_skip = max_skip; par.max_iter = max_iter; par.gap_scale = gap_scale;
	par.is_cdna = is_cdna; par.n_segs = n_segs; par.q_span_override = -1; par.flags = 0;
	if (mm2c_chain_task_host(&par, n, a, avg_qspan_scaled, f, p, tid) != 0) { /* chain.c:103; errors as chain_hardware.cpp:208-235 */
		fprintf(stderr, "Error: GPU chaining failed (n = %ld): %s\n", (long)n, mm2c_last_error());
		exit(EXIT_FAILURE);
	}
	u = (uint64_t *)xalloc(km, (size_t)n * 8);
	b = (mm2c_anchor_t *)xalloc(km, (size_t)n * sizeof(mm2c_anchor_t));
	tmp = (mm2c_anchor_t *)xalloc(km, (size_t)n * 2 * sizeof(mm2c_anchor_t));
	n_u = chain_epilogue(min_cnt, min_sc, n, a, f, p, vt, u, b, tmp, &n_b);
	xfree(km, f); xfree(km, p); xfree(km, vt); xfree(km, tmp); xfree(km, a);   /* chain.c:394,421: the callee owns a */
	if (n_u == 0) { xfree(km, u); xfree(km, b); return 0; }
	*n_u_ = n_u, *_u = u;
	return b;
}

/* ---- the epilogue for a CSR batch on host threads (tasks handed out through an atomic counter, in the style of kt_for,
 * kthread.c:30-52); compact outputs as mm2c_plan_chains_device ---- */
#include <pthread.h>
typedef struct {
	int min_cnt, min_sc; int64_t n_tasks; const int64_t *off; const mm2c_anchor_t *a; const int32_t *f, *p;
	int64_t *n_u, *n_b; uint64_t *u; mm2c_anchor_t *b; int64_t next; int64_t max_n;
} epi_shared_t;

static void *epi_worker(void *vp)
{
	epi_shared_t *s = (epi_shared_t *)vp;
	int32_t *vt = (int32_t *)malloc((size_t)(s->max_n ? s->max_n : 1) * 8);
	mm2c_anchor_t *tmp = (mm2c_anchor_t *)malloc((size_t)(s->max_n ? s->max_n : 1) * 2 * sizeof(mm2c_anchor_t));
	for (;;) {
		const int64_t k = __sync_fetch_and_add(&s->next, 1);
		int64_t o, n;
		if (k >= s->n_tasks) break;
		o = s->off[k] - s->off[0]; n = s->off[k + 1] - s->off[k];
		s->n_u[k] = 0; s->n_b[k] = 0;
		if (n) s->n_u[k] = chain_epilogue(s->min_cnt, s->min_sc, n, s->a + s->off[k], s->f + s->off[k], s->p + s->off[k], vt, s->u + o, s->b + o, tmp, &s->n_b[k]);
	}
	free(vt); free(tmp);
	return 0;
}

int mm2c_chain_epilogue_host(int min_cnt, int min_sc, int64_t n_tasks, const int64_t *h_offsets, const mm2c_anchor_t *h_anchors,
                             const int32_t *h_f, const int32_t *h_p, int n_threads, int64_t *u_off, uint64_t *u, int64_t *b_off,
                             mm2c_anchor_t *b)
{
	epi_shared_t s;
	pthread_t *th;
	int64_t k, total, au = 0, ab = 0;
	int i;
	if (n_tasks < 0 || !u_off || !b_off) return MM2C_E_ARG;
	u_off[0] = b_off[0] = 0;
	if (n_tasks == 0) return 0;
	if (!h_offsets || !u || !b) return MM2C_E_ARG;
	total = h_offsets[n_tasks] - h_offsets[0];
	s.min_cnt = min_cnt; s.min_sc = min_sc; s.n_tasks = n_tasks; s.off = h_offsets; s.a = h_anchors; s.f = h_f; s.p = h_p;
	s.n_u = u_off + 1; s.n_b = b_off + 1; s.next = 0; s.max_n = 0;                 /* counts first, turned into offsets below */
	s.u = (uint64_t *)malloc((size_t)(total ? total : 1) * 8);
	s.b = (mm2c_anchor_t *)malloc((size_t)(total ? total : 1) * sizeof(mm2c_anchor_t));
	for (k = 0; k < n_tasks; ++k) if (h_offsets[k + 1] - h_offsets[k] > s.max_n) s.max_n = h_offsets[k + 1] - h_offsets[k];
	if (n_threads < 1) n_threads = 1;
	th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
	for (i = 0; i < n_threads; ++i) pthread_create(&th[i], 0, epi_worker, &s);
	for (i = 0; i < n_threads; ++i) pthread_join(th[i], 0);
	for (k = 0; k < n_tasks; ++k) {                                               /* compact */
		const int64_t o = h_offsets[k] - h_offsets[0], nu = u_off[k + 1], nb = b_off[k + 1];
		memcpy(u + au, s.u + o, (size_t)nu * 8);
		memcpy(b + ab, s.b + o, (size_t)nb * sizeof(mm2c_anchor_t));
		au += nu; ab += nb;
		u_off[k + 1] = au; b_off[k + 1] = ab;
	}
	free(th); free(s.u); free(s.b);
	return 0;
}
