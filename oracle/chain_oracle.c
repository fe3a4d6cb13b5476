/*
 * chain_oracle.c -- CPU ORACLE (test infrastructure, never shipped / never called by the product path).
 * Plain-C restatement of the reference chaining DP; see chain_oracle.h for scope and pinning status.
 * Build: oracle/Makefile (gcc -O2 -fwrapv -ffp-contract=off).  Reference citations are file:line under
 * /root/reference (kisarur/minimap2-fpga).
 */
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <pthread.h>
#include <time.h>
#include "chain_oracle.h"

#define SEG_SHIFT 48                     /* mmpriv.h:22 MM_SEED_SEG_SHIFT */
#define SEG_OF(y) ((int32_t)(((y) >> SEG_SHIFT) & 0xff)) /* mmpriv.h:23 */
#define SPAN_OF(y) ((int32_t)(((y) >> 32) & 0xff))       /* chain.c:189: only 8 bits of span */

/* floor(log2(v)) for v>0; chain.c:15-27 uses a byte LUT, which equals 31-clz(v). */
static inline int32_t floor_log2_u32(uint32_t v) { return 31 - __builtin_clz(v); }

/* chain.c:48-49: u64 sum -> float, times double .01, divided by n (int64 -> double), rounded to float */
float mm2o_avg_qspan_scaled(int64_t n, const mm2o_anchor_t *a)
{
	uint64_t sum = 0;
	int64_t i;
	for (i = 0; i < n; ++i) sum += (uint64_t)SPAN_OF(a[i].y);
	return (float)(.01 * (float)sum / n);
}

/*
 * Score of extending predecessor j to anchor i (chain.c:199-220), WITHOUT adding f[j].
 * Returns 0 when the pair is filtered out (the `continue`s at chain.c:202-206), 1 otherwise.
 */
static inline int pair_score(const mm2o_params_t *par, float avg, uint64_t xi, int32_t qi, int32_t span_i, int32_t seg_i,
                             const mm2o_anchor_t *aj, int32_t *sc_out)
{
	int64_t dr = (int64_t)(xi - aj->x);
	int32_t dq = qi - (int32_t)aj->y;
	int32_t seg_j = SEG_OF(aj->y);
	int same = (seg_i == seg_j);
	int32_t dd, lg, sc, gap, lin;
	if ((same && dr == 0) || dq <= 0) return 0;                               /* chain.c:202 */
	if ((same && dq > par->max_dist_y) || dq > par->max_dist_x) return 0;     /* chain.c:203 */
	dd = (int32_t)(dr > dq ? dr - dq : dq - dr);                              /* chain.c:204 */
	if (same && dd > par->bw) return 0;                                       /* chain.c:205 */
	if (par->n_segs > 1 && !par->is_cdna && same && dr > par->max_dist_y) return 0; /* chain.c:206 */
	sc = (int32_t)(dq < dr ? dq : dr);                                        /* chain.c:207 */
	if (sc > span_i) sc = span_i;                                             /* chain.c:208 */
	lg = dd ? floor_log2_u32((uint32_t)dd) : 0;                               /* chain.c:209 */
	lin = (int32_t)(dd * avg);                                                /* float multiply, truncation */
	if (par->is_cdna || !same) {                                              /* chain.c:211-217 */
		if (!same && dr == 0) { ++sc; gap = 0; }
		else if (dr > dq || !same) gap = lin < lg ? lin : lg;
		else gap = lin + (lg >> 1);
	} else gap = lin + (lg >> 1);                                             /* chain.c:218 */
	sc -= (int32_t)((double)gap * par->gap_scale + .499);                    /* chain.c:219: double mul, then add */
	*sc_out = sc;
	return 1;
}

void mm2o_chain_fpv(const mm2o_params_t *par, int64_t n, const mm2o_anchor_t *a, float avg,
                    int32_t *f, int32_t *p, int32_t *v, int32_t *t)
{
	int64_t i, j, st = 0;
	if (n <= 0) return;
	memset(t, 0, (size_t)n * 4);                                              /* chain.c:46 */
	for (i = 0; i < n; ++i) {
		const uint64_t xi = a[i].x;
		const int32_t qi = (int32_t)a[i].y, span_i = SPAN_OF(a[i].y), seg_i = SEG_OF(a[i].y);
		int32_t best = span_i, n_skip = 0;
		int64_t best_j = -1;
		while (st < i && xi > a[st].x + (uint64_t)(int64_t)par->max_dist_x) ++st; /* chain.c:192 */
		if (i - st > par->max_iter) st = i - par->max_iter;                   /* chain.c:193 (persists) */
		for (j = i - 1; j >= st; --j) {                                       /* chain.c:197 */
			int32_t sc;
			if (!pair_score(par, avg, xi, qi, span_i, seg_i, &a[j], &sc)) continue;
			sc += f[j];                                                       /* chain.c:220 */
			if (sc > best) {                                                  /* chain.c:226-228 */
				best = sc; best_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == (int32_t)i) {                                  /* chain.c:229-232 */
				if (++n_skip > par->max_skip) break;
			}
			if (p[j] >= 0) t[p[j]] = (int32_t)i;                              /* chain.c:233 */
		}
		f[i] = best; p[i] = (int32_t)best_j;                                  /* chain.c:236 */
		if (v) v[i] = (best_j >= 0 && v[best_j] > best) ? v[best_j] : best;   /* chain.c:237 */
	}
}

int64_t mm2o_predict(int64_t n, const mm2o_anchor_t *a, int32_t max_dist_x, uint8_t *num_subparts, int64_t *total_trip)
{
	int64_t i, st = 0, tot_sub = 0, tot_trip = 0;
	for (i = 0; i < n; ++i) {                                                 /* chain.c:62-78 */
		int64_t trip, sub;
		while (st < i && a[i].x > a[st].x + (uint64_t)(int64_t)max_dist_x) ++st;
		trip = i - st;
		if (trip > 1024) trip = 1024;                                         /* MAX_TRIPCOUNT chain_hardware.h:60 */
		tot_trip += trip;
		sub = trip / 128;                                                     /* TRIPCOUNT_PER_SUBPART chain_hardware.h:58 */
		if (trip == 0 || trip % 128 > 0) ++sub;
		if (num_subparts) num_subparts[i] = (uint8_t)sub;
		tot_sub += sub;
	}
	if (total_trip) *total_trip = tot_trip;
	return tot_sub;
}

void mm2o_fill_v(int64_t n, const int32_t *f, const int32_t *p, int32_t *v)
{
	int64_t i;
	for (i = 0; i < n; ++i)                                                   /* chain.c:106-111 */
		v[i] = (p[i] >= 0 && v[p[i]] > f[i]) ? v[p[i]] : f[i];
}

/*
 * Literal emulation of device/minimap2_opencl.cl:24-172.  The FPGA keeps the last 1024 anchors in shift
 * registers (slot r = anchor i-r, zero-initialised), walks `num_subparts[i]` groups of 128 slots per
 * anchor, scores every slot of the group (.cl:116-127), takes the group's best scanning far->near with
 * `>=` and the `sc != q_span` guard (.cl:135-148), commits with a strict compare against the running
 * best of this anchor (.cl:150-154) and shifts after the last group (.cl:158-170).
 */
void mm2o_chain_hw_literal(int64_t n, int32_t max_dist_x, int32_t max_dist_y, int32_t bw, int32_t q_span,
                           float avg, const mm2o_anchor_t *a, const uint8_t *num_subparts,
                           int32_t *f, int32_t *p)
{
	enum { G = 128, DEPTH = 1024 };
	uint64_t *rx = (uint64_t*)calloc(DEPTH + 1, 8);
	int32_t *ry = (int32_t*)calloc(DEPTH + 1, 4), *rf = (int32_t*)calloc(DEPTH + 1, 4);
	int64_t i;
	for (i = 0; i < n; ++i) {
		int s, r, nsub = num_subparts[i];
		rx[0] = a[i].x; ry[0] = (int32_t)a[i].y; rf[0] = 0;
		for (s = 0; s < nsub; ++s) {
			int32_t grp_best = q_span;
			int64_t grp_j = -1;
			for (r = G; r > 0; --r) {                                         /* far -> near inside the group */
				int slot = s * G + r;
				int64_t dr = (int64_t)(rx[0] - rx[slot]);
				int32_t dq, dd, lg, sc, md;
				if (dr > max_dist_x || dr <= 0) continue;                     /* .cl:117 */
				dq = ry[0] - ry[slot];
				if (dq <= 0) continue;                                        /* .cl:119 */
				if (dq > max_dist_y || dq > max_dist_x) continue;             /* .cl:120 */
				dd = (int32_t)(dr > dq ? dr - dq : dq - dr);
				if (dd > bw) continue;                                        /* .cl:122 */
				md = (int32_t)(dq < dr ? dq : dr);
				sc = md > q_span ? q_span : md;
				lg = dd ? floor_log2_u32((uint32_t)dd) : 0;
				sc -= (int32_t)(dd * avg) + (lg >> 1);                        /* .cl:126 */
				sc += rf[slot];
				/* a filtered slot leaves sc_a[] = 0 (.cl:69); 0 >= q_span only if q_span <= 0: not modelled */
				if (sc >= grp_best && sc != q_span) { grp_best = sc; grp_j = i - slot; } /* .cl:137-147 */
			}
			if (grp_best > rf[0]) { f[i] = grp_best; p[i] = (int32_t)grp_j; rf[0] = grp_best; } /* .cl:150-154 */
		}
		memmove(rx + 1, rx, DEPTH * 8); memmove(ry + 1, ry, DEPTH * 4); memmove(rf + 1, rf, DEPTH * 4); /* .cl:158-164 */
	}
	free(rx); free(ry); free(rf);
}

/* ---------- radix sorts used by the backtrack stage (ksort.h:101-151, instantiated misc.c:155-159) ---------- */
/* The sort on 128-bit records keyed by .x is NOT stable (in-place American-flag passes), and chain emission
 * order depends on its permutation for equal keys, so the restatement follows the same pass structure:
 * <=64 elements: insertion sort; otherwise MSD byte passes from the top byte, buckets >64 recurse, 2..64
 * insertion-sorted. */
#define RS_SMALL 64

static void isort_u64(uint64_t *b, uint64_t *e)
{
	uint64_t *i, *j;
	for (i = b + 1; i < e; ++i)
		if (*i < *(i - 1)) {
			uint64_t tmp = *i;
			for (j = i; j > b && tmp < *(j - 1); --j) *j = *(j - 1);
			*j = tmp;
		}
}
static void isort_128x(mm2o_anchor_t *b, mm2o_anchor_t *e)
{
	mm2o_anchor_t *i, *j;
	for (i = b + 1; i < e; ++i)
		if (i->x < (i - 1)->x) {
			mm2o_anchor_t tmp = *i;
			for (j = i; j > b && tmp.x < (j - 1)->x; --j) *j = *(j - 1);
			*j = tmp;
		}
}

#define DEFINE_FLAG_SORT(NAME, T, KEY, ISORT) \
static void NAME(T *beg, T *end, int shift) \
{ \
	T *head[256], *tail[256]; \
	size_t cnt[256]; \
	int k; T *i; \
	memset(cnt, 0, sizeof(cnt)); \
	for (i = beg; i != end; ++i) ++cnt[(KEY(*i) >> shift) & 0xff]; \
	head[0] = beg; tail[0] = beg + cnt[0]; \
	for (k = 1; k < 256; ++k) { head[k] = tail[k-1]; tail[k] = head[k] + cnt[k]; } \
	for (k = 0; k < 256;) { \
		if (head[k] != tail[k]) { \
			int l = (int)((KEY(*head[k]) >> shift) & 0xff); \
			if (l != k) { \
				T carry = *head[k], sw; \
				do { sw = carry; carry = *head[l]; *head[l]++ = sw; l = (int)((KEY(carry) >> shift) & 0xff); } while (l != k); \
				*head[k]++ = carry; \
			} else ++head[k]; \
		} else ++k; \
	} \
	if (shift) { \
		int ns = shift > 8 ? shift - 8 : 0; \
		T *lo = beg; \
		for (k = 0; k < 256; ++k) { \
			T *hi = tail[k]; \
			if (hi - lo > RS_SMALL) NAME(lo, hi, ns); \
			else if (hi - lo > 1) ISORT(lo, hi); \
			lo = hi; \
		} \
	} \
}
#define KEY_U64(v) (v)
#define KEY_128X(v) ((v).x)
DEFINE_FLAG_SORT(flag_sort_u64, uint64_t, KEY_U64, isort_u64)
DEFINE_FLAG_SORT(flag_sort_128x, mm2o_anchor_t, KEY_128X, isort_128x)

static void sort_u64(uint64_t *b, uint64_t *e) { if (e - b <= RS_SMALL) isort_u64(b, e); else flag_sort_u64(b, e, 56); }
static void sort_128x(mm2o_anchor_t *b, mm2o_anchor_t *e) { if (e - b <= RS_SMALL) isort_128x(b, e); else flag_sort_128x(b, e, 56); }

/* exported for tests */
void mm2o_radix_sort_64(uint64_t *b, int64_t n) { sort_u64(b, b + n); }
void mm2o_radix_sort_128x(mm2o_anchor_t *b, int64_t n) { sort_128x(b, b + n); }

int32_t mm2o_backtrack(int64_t n, const mm2o_anchor_t *a, int32_t min_cnt, int32_t min_sc,
                       const int32_t *f, const int32_t *p, int32_t *v, int32_t *t,
                       uint64_t **u_out, mm2o_anchor_t **b_out, int64_t *n_b_out)
{
	int64_t i, j;
	int32_t n_u = 0, n_v = 0, k = 0;
	uint64_t *u, *u2;
	mm2o_anchor_t *b, *w, *tmp;
	*u_out = 0; *b_out = 0; *n_b_out = 0;
	/* chain ends: anchors nobody points to, whose peak score passes min_sc (chain.c:349-354) */
	memset(t, 0, (size_t)n * 4);
	for (i = 0; i < n; ++i) if (p[i] >= 0) t[p[i]] = 1;
	for (i = 0; i < n; ++i) if (t[i] == 0 && v[i] >= min_sc) ++n_u;
	if (n_u == 0) return 0;
	u = (uint64_t*)malloc((size_t)n_u * 8);
	for (i = 0, n_u = 0; i < n; ++i)
		if (t[i] == 0 && v[i] >= min_sc) {                                    /* chain.c:361-367 */
			j = i;
			while (j >= 0 && f[j] < v[j]) j = p[j];
			if (j < 0) j = i;
			u[n_u++] = (uint64_t)f[j] << 32 | (uint64_t)j;
		}
	sort_u64(u, u + n_u);                                                     /* chain.c:368 */
	for (i = 0; i < n_u >> 1; ++i) { uint64_t s = u[i]; u[i] = u[n_u - i - 1]; u[n_u - i - 1] = s; }
	/* backtrack from the best end, never re-using an anchor (chain.c:375-390); v[] is reused as the list */
	memset(t, 0, (size_t)n * 4);
	for (i = 0; i < n_u; ++i) {
		int32_t n_v0 = n_v, k0 = k;
		j = (int32_t)u[i];
		do { v[n_v++] = (int32_t)j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
		if (j < 0) {
			if (n_v - n_v0 >= min_cnt) u[k++] = u[i] >> 32 << 32 | (uint64_t)(n_v - n_v0);
		} else if ((int32_t)(u[i] >> 32) - f[j] >= min_sc) {
			if (n_v - n_v0 >= min_cnt) u[k++] = ((u[i] >> 32) - (uint64_t)f[j]) << 32 | (uint64_t)(n_v - n_v0);
		}
		if (k0 == k) n_v = n_v0;
	}
	n_u = k;
	/* emit anchors chain by chain in ascending order (chain.c:397-402) */
	b = (mm2o_anchor_t*)malloc((size_t)(n_v > 0 ? n_v : 1) * sizeof(*b));
	for (i = 0, k = 0; i < n_u; ++i) {
		int32_t k0 = k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	/* order chains by the x of their first anchor (chain.c:406-420) */
	w = (mm2o_anchor_t*)malloc((size_t)(n_u > 0 ? n_u : 1) * sizeof(*w));
	for (i = 0, k = 0; i < n_u; ++i) { w[i].x = b[k].x; w[i].y = (uint64_t)k << 32 | (uint64_t)i; k += (int32_t)u[i]; }
	sort_128x(w, w + n_u);
	u2 = (uint64_t*)malloc((size_t)(n_u > 0 ? n_u : 1) * 8);
	tmp = (mm2o_anchor_t*)malloc((size_t)(n_v > 0 ? n_v : 1) * sizeof(*tmp));
	for (i = 0, k = 0; i < n_u; ++i) {
		int32_t src = (int32_t)w[i].y, cnt = (int32_t)u[src];
		u2[i] = u[src];
		memcpy(&tmp[k], &b[w[i].y >> 32], (size_t)cnt * sizeof(*b));
		k += cnt;
	}
	free(b); free(w); free(u);
	if (n_u == 0) { free(u2); free(tmp); return 0; }
	*u_out = u2; *b_out = tmp; *n_b_out = k;
	return n_u;
}

int32_t mm2o_mm_chain_dp(const mm2o_params_t *par, int32_t min_cnt, int32_t min_sc, int64_t n,
                         const mm2o_anchor_t *a, uint64_t **u_out, mm2o_anchor_t **b_out, int64_t *n_b_out)
{
	int32_t *f, *p, *t, *v, n_u;
	*u_out = 0; *b_out = 0; *n_b_out = 0;
	if (n == 0 || a == 0) return 0;                                           /* chain.c:37-41 */
	f = (int32_t*)malloc((size_t)n * 4); p = (int32_t*)malloc((size_t)n * 4);
	t = (int32_t*)malloc((size_t)n * 4); v = (int32_t*)malloc((size_t)n * 4);
	mm2o_chain_fpv(par, n, a, mm2o_avg_qspan_scaled(n, a), f, p, v, t);
	n_u = mm2o_backtrack(n, a, min_cnt, min_sc, f, p, v, t, u_out, b_out, n_b_out);
	free(f); free(p); free(t); free(v);
	return n_u;
}

/* ---------- multi-thread timing helper (bench.py cpu_baseline) ---------- */
typedef struct {
	const mm2o_params_t *par; int64_t n_tasks; const int64_t *off; const mm2o_anchor_t *a;
	int32_t *f, *p; int tid, n_threads;
} bench_arg_t;

static void *bench_worker(void *vp)
{
	bench_arg_t *w = (bench_arg_t*)vp;
	int64_t k, max_n = 0;
	int32_t *t;
	for (k = w->tid; k < w->n_tasks; k += w->n_threads)
		if (w->off[k + 1] - w->off[k] > max_n) max_n = w->off[k + 1] - w->off[k];
	t = (int32_t*)malloc((size_t)(max_n > 0 ? max_n : 1) * 4);
	for (k = w->tid; k < w->n_tasks; k += w->n_threads) {
		int64_t o = w->off[k], n = w->off[k + 1] - o;
		mm2o_chain_fpv(w->par, n, w->a + o, mm2o_avg_qspan_scaled(n, w->a + o), w->f + o, w->p + o, 0, t);
	}
	free(t);
	return 0;
}

double mm2o_bench_batch(const mm2o_params_t *par, int64_t n_tasks, const int64_t *offsets,
                        const mm2o_anchor_t *a, int32_t *f, int32_t *p, int n_threads)
{
	struct timespec t0, t1;
	pthread_t *th;
	bench_arg_t *args;
	int i;
	if (n_threads < 1) n_threads = 1;
	th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
	args = (bench_arg_t*)malloc(sizeof(bench_arg_t) * (size_t)n_threads);
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (i = 0; i < n_threads; ++i) {
		bench_arg_t x = { par, n_tasks, offsets, a, f, p, i, n_threads };
		args[i] = x;
		pthread_create(&th[i], 0, bench_worker, &args[i]);
	}
	for (i = 0; i < n_threads; ++i) pthread_join(th[i], 0);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(th); free(args);
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* skip_seed, map.c:122-147.  The name comparison strcmp(qname, name[rid]) is carried by ranks: ref_rank[rid] = rank of the reference
 * sequence's name among the distinct reference names (strcmp order), q_lo = number of those names below the read's, q_eq = the read's
 * name is one of them -- so cmp > 0 <=> ref_rank[rid] < q_lo and cmp == 0 <=> q_eq && ref_rank[rid] == q_lo.  ref_rank == NULL stands
 * for qname == NULL (no name-dependent skipping, map.c:125). */
static int skip_seed(int32_t flag, uint64_t r, const mm2o_match_t *q, int32_t qlen, const int32_t *ref_rank, const int32_t *ref_len,
                     int32_t q_lo, int32_t q_eq, int *is_self)
{
	*is_self = 0;
	if (ref_rank && (flag & (MM2O_F_NO_DIAG | MM2O_F_NO_DUAL))) {
		const int32_t rr = ref_rank[r >> 32];
		const int cmp = rr < q_lo ? 1 : (q_eq && rr == q_lo) ? 0 : -1;
		if ((flag & MM2O_F_NO_DIAG) && cmp == 0 && ref_len[r >> 32] == qlen) {
			if ((uint32_t)r >> 1 == (q->q_pos >> 1)) return 1;                    /* the diagonal, map.c:131 */
			if ((r & 1) == (q->q_pos & 1)) *is_self = 1;                           /* map.c:132 */
		}
		if ((flag & MM2O_F_NO_DUAL) && cmp > 0) return 1;                        /* all-vs-all: map once, map.c:134-135 */
	}
	if (flag & (MM2O_F_FOR_ONLY | MM2O_F_REV_ONLY)) {                            /* map.c:137-143 */
		if ((r & 1) == (q->q_pos & 1)) { if (flag & MM2O_F_REV_ONLY) return 1; }
		else if (flag & MM2O_F_FOR_ONLY) return 1;
	}
	return 0;
}

int64_t mm2o_collect_seed_hits(int64_t n_m, const mm2o_match_t *m, const uint64_t *hits, int32_t qlen, mm2o_anchor_t *a)
{
	return mm2o_collect_seed_hits_flags(n_m, m, hits, qlen, 0, 0, 0, 0, 0, a);
}

/* map.c:215-247 (collect_seed_hits) */
int64_t mm2o_collect_seed_hits_flags(int64_t n_m, const mm2o_match_t *m, const uint64_t *hits, int32_t qlen, int32_t flag,
                                     const int32_t *ref_rank, const int32_t *ref_len, int32_t q_lo, int32_t q_eq, mm2o_anchor_t *a)
{
	int64_t i, n_a = 0;
	for (i = 0; i < n_m; ++i) {
		const mm2o_match_t *q = &m[i];
		const uint64_t *r = hits + q->cr_off;
		uint32_t k;
		for (k = 0; k < q->n; ++k) {
			const int32_t rpos = (uint32_t)r[k] >> 1;
			int is_self;
			mm2o_anchor_t *p;
			if (skip_seed(flag, r[k], q, qlen, ref_rank, ref_len, q_lo, q_eq, &is_self)) continue;
			p = &a[n_a++];
			if ((r[k] & 1) == (q->q_pos & 1)) {                                  /* forward strand, map.c:232-234 */
				p->x = (r[k] & 0xffffffff00000000ULL) | (uint32_t)rpos;
				p->y = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
			} else {                                                             /* reverse strand, map.c:235-238 */
				p->x = 1ULL << 63 | (r[k] & 0xffffffff00000000ULL) | (uint32_t)rpos;
				p->y = (uint64_t)q->q_span << 32 | (uint32_t)(qlen - (int32_t)((q->q_pos >> 1) + 1 - q->q_span) - 1);
			}
			p->y |= (uint64_t)(q->seg_tandem >> 1) << 48;                        /* MM_SEED_SEG_SHIFT, map.c:239 */
			if (q->seg_tandem & 1) p->y |= 1ULL << 42;                           /* MM_SEED_TANDEM, map.c:240 */
			if (is_self) p->y |= 1ULL << 43;                                     /* MM_SEED_SELF, map.c:241, mmpriv.h:20 */
		}
	}
	sort_128x(a, a + n_a);                                                       /* map.c:245 */
	return n_a;
}

/* ---- collect_seed_hits_heap (map.c:149-213; selected by MM_F_HEAP_SORT: `--heap-sort`, main.c:245, and -x sr, options.c:125) ----
 * The hit lists of the matches (each ascending) are merged through a binary heap keyed on the hit alone (heap_lt, map.c:80: a.x > b.x,
 * a min-heap without a tie rule; ks_heapmake / ks_heapdown of ksort.h:43-60), forward-strand anchors are written from the front, reverse-strand
 * ones from the back and turned around afterwards (map.c:201-211).  The result is ascending in x like the radix-sorted list; the two differ in
 * the order among anchors with EQUAL x (one reference position hit by several query minimizers), which here is whatever order the heap pops
 * equal keys in -- so the heap itself is restated, operation by operation. */
typedef struct { uint64_t x, y; } heap_ent_t;                                  /* x = the hit, y = match index << 32 | position in its list */

static void heap_down(size_t i, size_t n, heap_ent_t *l)                       /* ksort.h:43-53 with heap_lt */
{
	size_t k = i;
	heap_ent_t tmp = l[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && l[k].x > l[k + 1].x) ++k;
		if (l[k].x > tmp.x) break;
		l[i] = l[k]; i = k;
	}
	l[i] = tmp;
}

int64_t mm2o_collect_seed_hits_heap(int64_t n_m, const mm2o_match_t *m, const uint64_t *hits, int32_t qlen, int32_t flag,
                                    const int32_t *ref_rank, const int32_t *ref_len, int32_t q_lo, int32_t q_eq, mm2o_anchor_t *a)
{
	int64_t i, j, n_a = 0, n_for = 0, n_rev = 0;
	size_t heap_size = 0;
	heap_ent_t *heap = (heap_ent_t *)malloc((size_t)(n_m > 0 ? n_m : 1) * sizeof(heap_ent_t));
	for (i = 0; i < n_m; ++i) n_a += m[i].n;
	for (i = 0; i < n_m; ++i)                                                   /* map.c:162-168 */
		if (m[i].n > 0) { heap[heap_size].x = hits[m[i].cr_off]; heap[heap_size].y = (uint64_t)i << 32; ++heap_size; }
	if (heap_size > 1) { size_t k; for (k = (heap_size >> 1) - 1; k != (size_t)(-1); --k) heap_down(k, heap_size, heap); }   /* ks_heapmake, ksort.h:54-59 */
	while (heap_size > 0) {                                                     /* map.c:170-198 */
		const mm2o_match_t *q = &m[heap->y >> 32];
		const uint64_t r = heap->x;
		const int32_t rpos = (uint32_t)r >> 1;
		int is_self;
		if (!skip_seed(flag, r, q, qlen, ref_rank, ref_len, q_lo, q_eq, &is_self)) {
			mm2o_anchor_t *p;
			if ((r & 1) == (q->q_pos & 1)) {
				p = &a[n_for++];
				p->x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
				p->y = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
			} else {
				p = &a[n_a - (++n_rev)];
				p->x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
				p->y = (uint64_t)q->q_span << 32 | (uint32_t)(qlen - (int32_t)((q->q_pos >> 1) + 1 - q->q_span) - 1);
			}
			p->y |= (uint64_t)(q->seg_tandem >> 1) << 48;
			if (q->seg_tandem & 1) p->y |= 1ULL << 42;
			if (is_self) p->y |= 1ULL << 43;
		}
		if ((uint32_t)heap->y < q->n - 1) {                                     /* the match's next hit takes the root */
			++heap[0].y;
			heap[0].x = hits[m[heap[0].y >> 32].cr_off + (uint32_t)heap[0].y];
		} else {
			heap[0] = heap[heap_size - 1];
			--heap_size;
		}
		if (heap_size > 0) heap_down(0, heap_size, heap);
	}
	free(heap);
	for (j = 0; j < n_rev >> 1; ++j) {                                          /* map.c:201-206 */
		const mm2o_anchor_t t = a[n_a - 1 - j];
		a[n_a - 1 - j] = a[n_a - (n_rev - j)];
		a[n_a - (n_rev - j)] = t;
	}
	if (n_a > n_for + n_rev) {                                                  /* map.c:207-210 */
		memmove(a + n_for, a + n_a - n_rev, (size_t)n_rev * sizeof(mm2o_anchor_t));
		n_a = n_for + n_rev;
	}
	return n_a;
}
