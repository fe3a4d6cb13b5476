/* batch_driver.c -- what worker_for / mm_map_frag (map.c:272-392,427-467) look like once they are restructured around the batch API
 * (SURVEY.md section 8 f2): seed ALL reads of a mini-batch -> ONE call that does collect_seed_hits + mm_chain_dp for all of them on the
 * GPU (mm2c_seed_chain_batch_host: matches in, chains out) -> post-process ALL reads -> print.
 * An example host of the product (INTEGRATION.md path C) and the caller of the end-to-end checks: it links the reference's own objects (sketch, index, hit, esterr, format, ...) and the
 * product library; the control flow around the one GPU call restates mm_map_frag for the `-x map-ont`, PAF-without-CIGAR case:
 *   before: hash (map.c:285-287), mm_sketch (collect_minimizers map.c:61-74, sdust_thres = 0), collect_matches (map.c:84-120)
 *   after : mm_gen_regs (map.c:345), chain_post (map.c:249-259), mm_est_err (map.c:360), mm_set_mapq (map.c:364), output (map.c:584-596)
 * The PAF it prints must be byte-identical to the reference host's (oracle/_ref/mm2_refhost), which tools/e2e_batch.sh checks.
 * usage: mm2_batchhost [-t threads] <ref.fa> <query.fa> */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "minimap.h"
#include "mmpriv.h"
#include "bseq.h"
#include "kalloc.h"
#include "kthread.h"
#include "khash.h"
#define MM2C_NO_MM_CHAIN_DP_DECL
#include "mm2chain.h"

void mm_mapopt_update(mm_mapopt_t *opt, const mm_idx_t *mi)          /* options.c:59-69 */
{
	if ((opt->flag & MM_F_SPLICE_FOR) || (opt->flag & MM_F_SPLICE_REV)) opt->flag |= MM_F_SPLICE;
	if (opt->mid_occ <= 0) opt->mid_occ = mm_idx_cal_max_occ(mi, opt->mid_occ_frac);
	if (opt->mid_occ < opt->min_mid_occ) opt->mid_occ = opt->min_mid_occ;
}

void mm_idxopt_init(mm_idxopt_t *io)                                    /* options.c:8-15, imported by index.c:569 */
{
	memset(io, 0, sizeof(*io));
	io->k = 15; io->w = 10; io->flag = 0; io->bucket_bits = 14;
	io->mini_batch_size = 50000000; io->batch_size = 4000000000ULL;
}

static void defaults(mm_idxopt_t *io, mm_mapopt_t *mo)               /* as driver.c: options.c:8-57 + map-ont */
{
	mm_idxopt_init(io); memset(mo, 0, sizeof(*mo));
	mo->seed = 11; mo->mid_occ_frac = 2e-4f; mo->sdust_thres = 0;
	mo->min_cnt = 3; mo->min_chain_score = 40; mo->bw = 500; mo->max_gap = 5000; mo->max_gap_ref = -1;
	mo->max_chain_skip = 25; mo->max_chain_iter = 5000; mo->chain_gap_scale = 1.0f;
	mo->mask_level = 0.5f; mo->mask_len = INT_MAX; mo->pri_ratio = 0.8f; mo->best_n = 5;
	mo->max_join_long = 20000; mo->max_join_short = 2000; mo->min_join_flank_sc = 1000; mo->min_join_flank_ratio = 0.5f;
	mo->alt_drop = 0.15f;
	mo->a = 2; mo->b = 4; mo->q = 4; mo->e = 2; mo->q2 = 24; mo->e2 = 1; mo->sc_ambi = 1;
	mo->zdrop = 400; mo->zdrop_inv = 200; mo->end_bonus = -1; mo->min_dp_max = mo->min_chain_score * mo->a;
	mo->min_ksw_len = 200; mo->anchor_ext_len = 20; mo->anchor_ext_shift = 6; mo->max_clip_ratio = 1.0f;
	mo->mini_batch_size = 500000000; mo->pe_ori = 0; mo->pe_bonus = 33;
}

/* The index's position arrays as ONE pool that lives on the GPU (mm2c_hitpool_create): per bucket its p[] (index.c:24) followed by the value
 * array of its hash table (a minimizer that occurs once keeps its position in the table itself, index.c:91-94).  mm_idx_get hands out a pointer
 * into one of the two; pool_offset() turns it into an offset into the pool, which is what a match then carries instead of a copy of the hits.
 * Both layouts are the reference's own: struct mm_idx_bucket_s is private to index.c (index.c:26-31, restated here field for field), the
 * table type comes from the reference's khash.h through the macro index.c itself uses (index.c:20). */
__KHASH_TYPE(idx, uint64_t, uint64_t)
typedef struct { mm128_v a; int32_t n; uint64_t *p; void *h; } idx_bucket_t;
typedef struct { int64_t *base_p, *base_v, n; mm2c_hitpool_t *dev; } pool_t;

static int pool_build(const mm_idx_t *mi, pool_t *pl)
{
	const idx_bucket_t *B = (const idx_bucket_t *)mi->B;
	const int nb = 1 << mi->b;
	uint64_t *host;
	int i;
	pl->base_p = (int64_t *)malloc((size_t)nb * 8); pl->base_v = (int64_t *)malloc((size_t)nb * 8); pl->n = 0;
	for (i = 0; i < nb; ++i) {
		const kh_idx_t *h = (const kh_idx_t *)B[i].h;
		pl->base_p[i] = pl->n; pl->n += B[i].n;
		pl->base_v[i] = pl->n; pl->n += h ? h->n_buckets : 0;
	}
	host = (uint64_t *)calloc((size_t)pl->n + 1, 8);
	for (i = 0; i < nb; ++i) {
		const kh_idx_t *h = (const kh_idx_t *)B[i].h;
		if (B[i].n) memcpy(host + pl->base_p[i], B[i].p, (size_t)B[i].n * 8);
		if (h && h->n_buckets) memcpy(host + pl->base_v[i], h->vals, (size_t)h->n_buckets * 8);
	}
	pl->dev = mm2c_hitpool_create(host, pl->n);
	free(host);
	return pl->dev ? 0 : -1;
}

static inline int64_t pool_offset(const mm_idx_t *mi, const pool_t *pl, uint64_t minier, const uint64_t *cr)
{
	const int i = (int)(minier & ((1u << mi->b) - 1));
	const idx_bucket_t *b = &((const idx_bucket_t *)mi->B)[i];
	if (cr >= b->p && cr < b->p + b->n) return pl->base_p[i] + (cr - b->p);
	return pl->base_v[i] + (cr - ((const kh_idx_t *)b->h)->vals);
}

static void pool_free(pool_t *pl) { mm2c_hitpool_destroy(pl->dev); free(pl->base_p); free(pl->base_v); memset(pl, 0, sizeof(*pl)); }

typedef struct {
	uint32_t hash;
	int32_t rep_len, n_mini_pos, n_m, n_reg;
	int64_t n_a;
	uint64_t *mini_pos;
	mm2c_match_t *m;              /* cr_off holds nothing yet; crs[] has the pointers into the index */
	const uint64_t **crs;
	mm_reg1_t *reg;
} read_t;

/* the big arrays of a mini-batch live in page-locked memory (mm2c_pinned_alloc) and are reused by later mini-batches: no page faults
 * on fresh pages, copies at PCIe rate */
typedef struct {
	int64_t cap_reads, cap_matches, cap_hits;
	int64_t *match_off, *hit_off, *anchor_off, *u_off, *b_off;
	mm2c_match_t *matches; uint64_t *hits, *u; mm2c_anchor_t *b; int32_t *qlen;
	int busy;
} bufs_t;

typedef struct {
	const mm_idx_t *mi; const mm_mapopt_t *opt; mm_bseq1_t *seq; read_t *rd; int n;
	const pool_t *pool;           /* resident position arrays, or 0: the hits are copied into the batch (MM2_BATCH_HOSTPOOL=1) */
	void **km;                    /* per-thread kalloc arenas for mm_sketch, as mm_tbuf_t::km (map.c:22) */
	bufs_t *bf;
	/* batch arrays (in bf) */
	int64_t *match_off, *hit_off, *anchor_off, *u_off, *b_off;
	mm2c_match_t *matches; uint64_t *hits, *u; mm2c_anchor_t *b; int32_t *qlen;
} batch_t;

/* stage A, per read: map.c:281-295 up to the point where the anchors would be made */
static void seed_one(void *data, long i, int tid)
{
	batch_t *bt = (batch_t *)data;
	const mm_idx_t *mi = bt->mi; const mm_mapopt_t *opt = bt->opt;
	mm_bseq1_t *t = &bt->seq[i];
	read_t *r = &bt->rd[i];
	mm128_v mv = {0, 0, 0};
	int rep_st = 0, rep_en = 0, max_occ = opt->mid_occ;
	size_t j;
	void *km = bt->km[tid];
	memset(r, 0, sizeof(*r));
	if (t->l_seq == 0 || (opt->max_qlen > 0 && t->l_seq > opt->max_qlen)) return;                /* map.c:282-283 */
	r->hash = t->name ? __ac_X31_hash_string(t->name) : 0;                                         /* map.c:285-287 */
	r->hash ^= __ac_Wang_hash(t->l_seq) + __ac_Wang_hash(opt->seed);
	r->hash = __ac_Wang_hash(r->hash);
	mm_sketch(km, t->seq, t->l_seq, mi->w, mi->k, 0, mi->flag & MM_I_HPC, &mv);                  /* map.c:69 */
	r->mini_pos = (uint64_t *)malloc((mv.n + 1) * 8);
	r->m = (mm2c_match_t *)malloc((mv.n + 1) * sizeof(mm2c_match_t));
	r->crs = bt->pool ? 0 : (const uint64_t **)malloc((mv.n + 1) * sizeof(*r->crs));
	for (j = 0; j < mv.n; ++j) {                                                                   /* map.c:95-118 */
		const mm128_t *p = &mv.a[j];
		uint32_t q_pos = (uint32_t)p->y, q_span = p->x & 0xff;
		int n;
		const uint64_t *cr = mm_idx_get(mi, p->x >> 8, &n);
		if (n >= max_occ) {
			int en = (q_pos >> 1) + 1, st = en - q_span;
			if (st > rep_en) { r->rep_len += rep_en - rep_st; rep_st = st, rep_en = en; }
			else rep_en = en;
		} else {
			mm2c_match_t *q = &r->m[r->n_m];
			uint32_t is_tandem = 0;
			if (j > 0 && p->x >> 8 == mv.a[j - 1].x >> 8) is_tandem = 1;
			if (j < mv.n - 1 && p->x >> 8 == mv.a[j + 1].x >> 8) is_tandem = 1;
			if (n > 0) {                                              /* a match without hits makes no anchor: not sent to the GPU */
				q->cr_off = bt->pool ? pool_offset(mi, bt->pool, p->x >> 8, cr) : 0;
				q->n = (uint32_t)n; q->q_pos = q_pos; q->q_span = q_span; q->seg_tandem = (uint32_t)(p->y >> 32) << 1 | is_tandem;
				if (!bt->pool) r->crs[r->n_m] = cr;
				++r->n_m;
				r->n_a += n;
			}
			r->mini_pos[r->n_mini_pos++] = (uint64_t)q_span << 32 | q_pos >> 1;
		}
	}
	r->rep_len += rep_en - rep_st;
	kfree(km, mv.a);
}

/* stage A', per read: its matches and hits into the batch arrays (a host that kept the index's position arrays on the GPU would
 * pass offsets into them instead of copying the hits) */
static void pack_one(void *data, long i, int tid)
{
	batch_t *bt = (batch_t *)data;
	read_t *r = &bt->rd[i];
	int64_t h = bt->hit_off[i];
	int k;
	(void)tid;
	if (bt->pool) {                                                /* the matches already point into the resident pool */
		if (r->n_m) memcpy(&bt->matches[bt->match_off[i]], r->m, (size_t)r->n_m * sizeof(mm2c_match_t));
		bt->qlen[i] = bt->seq[i].l_seq;
		free(r->m); r->m = 0;
		return;
	}
	for (k = 0; k < r->n_m; ++k) {
		mm2c_match_t *q = &bt->matches[bt->match_off[i] + k];
		*q = r->m[k];
		q->cr_off = h;
		memcpy(bt->hits + h, r->crs[k], (size_t)q->n * 8);
		h += q->n;
	}
	bt->qlen[i] = bt->seq[i].l_seq;
	free(r->m); free(r->crs); r->m = 0; r->crs = 0;
}

/* stage C, per read: map.c:345-365 for one segment without CIGAR */
static void post_one(void *data, long i, int tid)
{
	batch_t *bt = (batch_t *)data;
	const mm_idx_t *mi = bt->mi; const mm_mapopt_t *opt = bt->opt;
	read_t *r = &bt->rd[i];
	const int qlen = bt->seq[i].l_seq;
	int n_regs0 = (int)(bt->u_off[i + 1] - bt->u_off[i]);
	uint64_t *u = bt->u + bt->u_off[i];
	mm128_t *a = (mm128_t *)(bt->b + bt->b_off[i]);
	mm_reg1_t *regs0;
	(void)tid;
	if (qlen == 0 || (opt->max_qlen > 0 && qlen > opt->max_qlen)) { r->n_reg = 0; r->reg = 0; return; }
	regs0 = mm_gen_regs(0, r->hash, qlen, n_regs0, u, a);                                          /* map.c:345 */
	if (!(opt->flag & MM_F_ALL_CHAINS)) {                                                          /* chain_post, map.c:251-258 */
		mm_set_parent(0, opt->mask_level, opt->mask_len, n_regs0, regs0, opt->a * 2 + opt->b, opt->flag & MM_F_HARD_MLEVEL, opt->alt_drop);
		mm_select_sub(0, opt->pri_ratio, mi->k * 2, opt->best_n, &n_regs0, regs0);
		if (!(opt->flag & (MM_F_SPLICE | MM_F_SR | MM_F_NO_LJOIN))) mm_join_long(0, opt, qlen, &n_regs0, regs0, a);
	}
	mm_est_err(mi, qlen, n_regs0, regs0, a, r->n_mini_pos, r->mini_pos);                           /* map.c:360 */
	mm_set_mapq(0, n_regs0, regs0, opt->min_chain_score, opt->a, r->rep_len, 0);                   /* map.c:364 */
	r->n_reg = n_regs0; r->reg = regs0;
	free(r->mini_pos); r->mini_pos = 0;
}

/* the mini-batch pipeline (cf. worker_pipeline, map.c:529-620): read | seed all | pack | chain the batch on the GPU + post all | print;
 * the steps of consecutive mini-batches overlap, so the GPU call of one batch hides behind the seeding of the next */
typedef struct {
	const mm_idx_t *mi; const mm_mapopt_t *opt; mm_bseq_file_t *fp; mm2c_params_t par; int n_threads;
	pool_t ipool; int use_pool, pageable_out;
	kstring_t str;
	void **km;
	bufs_t pool[6];
	double t_gpu, t_seed, t_pack, t_post, t_out, t_read;
	int64_t tot_anchors, tot_reads;
} shared_t;

static void *pipeline_step(void *shared, int step, void *in)
{
	shared_t *sh = (shared_t *)shared;
	const mm_mapopt_t *mo = sh->opt;
	double tt = realtime();
	if (step == 0) {                                                                               /* read a mini-batch, map.c:530-536 */
		batch_t *bt = (batch_t *)calloc(1, sizeof(batch_t));
		bt->mi = sh->mi; bt->opt = mo;
		bt->seq = mm_bseq_read3(sh->fp, mo->mini_batch_size, 0, 0, 0, &bt->n);
		sh->t_read += realtime() - tt;
		if (bt->seq) return bt;
		free(bt);
		return 0;
	} else if (step == 1) {                                                                        /* seed all */
		batch_t *bt = (batch_t *)in;
		bt->rd = (read_t *)calloc((size_t)bt->n, sizeof(read_t));
		bt->km = sh->km; bt->pool = sh->use_pool ? &sh->ipool : 0;
		kt_for(sh->n_threads, seed_one, bt, bt->n);
		sh->t_seed += realtime() - tt;
		return bt;
	} else if (step == 2) {                                                                        /* pack: the matches of all reads into one page-locked array */
		batch_t *bt = (batch_t *)in;
		int i;
		int64_t n_m = 0, n_h = 0;
		for (i = 0; i < bt->n; ++i) { n_m += bt->rd[i].n_m; n_h += bt->rd[i].n_a; }
		{	/* a free set of page-locked buffers, grown if this mini-batch is bigger than the ones it served before */
			bufs_t *bf = 0;
			int k;
			for (k = 0; k < 6 && !bf; ++k) if (!sh->pool[k].busy) bf = &sh->pool[k];   /* at most 5 mini-batches (one per pipeline thread) hold a set */
			if (!bf) { fprintf(stderr, "ERROR: no free buffer set\n"); exit(1); }
			bf->busy = 1; bt->bf = bf;
			if (bt->n + 1 > bf->cap_reads) {
				mm2c_pinned_free(bf->match_off); mm2c_pinned_free(bf->hit_off); mm2c_pinned_free(bf->anchor_off); mm2c_pinned_free(bf->u_off);
				mm2c_pinned_free(bf->b_off); mm2c_pinned_free(bf->qlen);
				bf->cap_reads = (bt->n + 1) * 5 / 4;
				bf->match_off = (int64_t *)mm2c_pinned_alloc((size_t)bf->cap_reads * 8); bf->hit_off = (int64_t *)mm2c_pinned_alloc((size_t)bf->cap_reads * 8);
				bf->anchor_off = (int64_t *)mm2c_pinned_alloc((size_t)bf->cap_reads * 8); bf->u_off = (int64_t *)mm2c_pinned_alloc((size_t)bf->cap_reads * 8);
				bf->b_off = (int64_t *)mm2c_pinned_alloc((size_t)bf->cap_reads * 8); bf->qlen = (int32_t *)mm2c_pinned_alloc((size_t)bf->cap_reads * 4);
			}
			if (n_m + 1 > bf->cap_matches) {
				mm2c_pinned_free(bf->matches);
				bf->cap_matches = (n_m + 1) * 5 / 4;
				bf->matches = (mm2c_match_t *)mm2c_pinned_alloc((size_t)bf->cap_matches * sizeof(mm2c_match_t));
			}
			if (n_h + 1 > bf->cap_hits) {
				mm2c_pinned_free(bf->hits);
				if (sh->pageable_out) { free(bf->u); free(bf->b); } else { mm2c_pinned_free(bf->u); mm2c_pinned_free(bf->b); }
				bf->cap_hits = (n_h + 1) * 5 / 4;
				bf->hits = sh->use_pool ? 0 : (uint64_t *)mm2c_pinned_alloc((size_t)bf->cap_hits * 8);
				/* the chains come back into these; the interface wants room for every anchor although about half of them end up in chains.  Page-locking
				 * 24 bytes per anchor costs more than it saves when a buffer set serves one mini-batch only (measured: 120 000 reads, -K 500M: 6.2 s page-locked, 5.1 s plain; -K 100M: 4.1 / 3.5 s).  MM2_BATCH_PINNED_OUT=1: page-locked */
				bf->u = (uint64_t *)(sh->pageable_out ? malloc((size_t)bf->cap_hits * 8) : mm2c_pinned_alloc((size_t)bf->cap_hits * 8));
				bf->b = (mm2c_anchor_t *)(sh->pageable_out ? malloc((size_t)bf->cap_hits * 16) : mm2c_pinned_alloc((size_t)bf->cap_hits * 16));
			}
			bt->match_off = bf->match_off; bt->hit_off = bf->hit_off; bt->anchor_off = bf->anchor_off; bt->u_off = bf->u_off; bt->b_off = bf->b_off;
			bt->qlen = bf->qlen; bt->matches = bf->matches; bt->hits = bf->hits; bt->u = bf->u; bt->b = bf->b;
		}
		bt->match_off[0] = bt->hit_off[0] = 0;
		n_m = n_h = 0;
		for (i = 0; i < bt->n; ++i) { n_m += bt->rd[i].n_m; n_h += bt->rd[i].n_a; bt->match_off[i + 1] = n_m; bt->hit_off[i + 1] = n_h; }
		kt_for(sh->n_threads < 4 ? sh->n_threads : 4, pack_one, bt, bt->n);                        /* copies: a few threads saturate memory, the cores seed the next mini-batch */
		sh->t_pack += realtime() - tt;
		return bt;
	} else if (step == 3) {                                                                        /* chain the batch, post all */
		batch_t *bt = (batch_t *)in;
		const int64_t n_h = bt->hit_off[bt->n];
		if ((sh->use_pool ? mm2c_seed_chain_batch_pool(&sh->par, mo->min_cnt, mo->min_chain_score, bt->n, bt->match_off, bt->matches, sh->ipool.dev, bt->qlen,
		                                               bt->anchor_off, bt->u_off, bt->u, bt->b_off, bt->b)
		                  : mm2c_seed_chain_batch_host(&sh->par, mo->min_cnt, mo->min_chain_score, bt->n, bt->match_off, bt->matches, bt->hits, n_h, bt->qlen,
		                                               bt->anchor_off, bt->u_off, bt->u, bt->b_off, bt->b)) != 0) {
			fprintf(stderr, "ERROR: %s\n", mm2c_last_error()); exit(1);
		}
		sh->t_gpu += realtime() - tt; sh->tot_anchors += n_h; sh->tot_reads += bt->n;
		tt = realtime();
		kt_for(sh->n_threads, post_one, bt, bt->n);
		sh->t_post += realtime() - tt;
		return bt;
	} else {                                                                                       /* output, map.c:584-596 */
		batch_t *bt = (batch_t *)in;
		int i, j;
		for (i = 0; i < bt->n; ++i) {
			read_t *r = &bt->rd[i];
			for (j = 0; j < r->n_reg; ++j) {
				mm_reg1_t *reg = &r->reg[j];
				if ((mo->flag & MM_F_NO_PRINT_2ND) && reg->id != reg->parent) continue;
				mm_write_paf3(&sh->str, sh->mi, &bt->seq[i], reg, 0, mo->flag, r->rep_len);
				puts(sh->str.s);
			}
			for (j = 0; j < r->n_reg; ++j) free(r->reg[j].p);
			free(r->reg);
			free(bt->seq[i].seq); free(bt->seq[i].name);
			if (bt->seq[i].qual) free(bt->seq[i].qual);
			if (bt->seq[i].comment) free(bt->seq[i].comment);
		}
		free(bt->seq); free(bt->rd);
		bt->bf->busy = 0;
		free(bt);
		sh->t_out += realtime() - tt;
	}
	return 0;
}

int main(int argc, char *argv[])
{
	mm_idxopt_t io;
	mm_mapopt_t mo;
	mm_idx_reader_t *rd;
	mm_idx_t *mi;
	shared_t sh;
	int n_threads = 1;
	double t_idx = 0, t_pool = 0, t_init = 0, tt;
	if (argc >= 5 && strcmp(argv[1], "-t") == 0) { n_threads = atoi(argv[2]); argv += 2; argc -= 2; }
	if (argc < 3) { fprintf(stderr, "usage: %s [-t threads] <ref.fa> <query.fa>\n", argv[0]); return 1; }
	mm_verbose = 1;
	mm_realtime0 = realtime();
	setenv("GPU_MAX_HW_QUEUES", "16", 0);   /* the host's own environment, before its first HIP call: the library's pipelines want a hardware queue per stream */
	/* hardware_init's place (main.c:367); the runtime start-up runs beside the index loading below, the first library call waits for it (MM2_SYNC_INIT: the old blocking form) */
	if ((getenv("MM2_SYNC_INIT") ? mm2c_init(-1) : mm2c_init_async(-1)) != 0) { fprintf(stderr, "ERROR: %s\n", mm2c_last_error()); return 1; }
	t_init = realtime() - mm_realtime0;
	defaults(&io, &mo);
	if (getenv("MM2_MINI_BATCH")) mo.mini_batch_size = atoll(getenv("MM2_MINI_BATCH"));              /* main.c -K */
	io.flag |= MM_I_NO_SEQ;
	rd = mm_idx_reader_open(argv[1], &io, 0);
	if (!rd) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
	memset(&sh, 0, sizeof(sh));
	tt = realtime();
	while ((mi = mm_idx_reader_read(rd, n_threads)) != 0) {
		t_idx += realtime() - tt;
		mm_mapopt_update(&mo, mi);
		/* mm_chain_dp arguments as mm_map_frag passes them (map.c:305-316): max_gap for both distances when max_gap_ref <= 0 */
		sh.par.max_dist_x = mo.max_gap_ref > 0 ? mo.max_gap_ref : mo.max_gap; sh.par.max_dist_y = mo.max_gap; sh.par.bw = mo.bw;
		sh.par.max_skip = mo.max_chain_skip; sh.par.max_iter = mo.max_chain_iter; sh.par.gap_scale = mo.chain_gap_scale;
		sh.par.is_cdna = 0; sh.par.n_segs = 1; sh.par.q_span_override = -1; sh.par.flags = 0;
		sh.mi = mi; sh.opt = &mo; sh.n_threads = n_threads;
		sh.use_pool = !(getenv("MM2_BATCH_HOSTPOOL") && atoi(getenv("MM2_BATCH_HOSTPOOL")));
		sh.pageable_out = !(getenv("MM2_BATCH_PINNED_OUT") && atoi(getenv("MM2_BATCH_PINNED_OUT")));   /* default: plain memory for the chains */
		if (sh.use_pool) {
			double tp = realtime();
			if (pool_build(mi, &sh.ipool) != 0) { fprintf(stderr, "ERROR: %s\n", mm2c_last_error()); return 1; }
			t_pool += realtime() - tp;
			fprintf(stderr, "[mm2_batchhost] position arrays of the index resident on the GPU: %lld entries (%.1f MB), %.2f s\n", (long long)sh.ipool.n, sh.ipool.n * 8e-6, realtime() - tp);
		}
		if (!sh.km) { int k; sh.km = (void **)calloc((size_t)n_threads, sizeof(void *)); for (k = 0; k < n_threads; ++k) sh.km[k] = km_init(); }
		sh.fp = mm_bseq_open(argv[2]);
		if (!sh.fp) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
		kt_pipeline(5, pipeline_step, &sh, 5);
		mm_bseq_close(sh.fp);
		if (sh.use_pool) pool_free(&sh.ipool);
		mm_idx_destroy(mi);
		tt = realtime();
	}
	free(sh.str.s);
	mm_idx_reader_close(rd);
	fprintf(stderr, "[mm2_batchhost] stages (summed over mini-batches, they overlap): index %.2f s, read %.2f, seed all %.2f, pack %.2f, GPU call %.2f, post all %.2f, output %.2f\n",
	        t_idx, sh.t_read, sh.t_seed, sh.t_pack, sh.t_gpu, sh.t_post, sh.t_out);
	fprintf(stderr, "[mm2_batchhost] %lld reads, %lld anchors; %.3f s in the batched GPU calls (matches in, chains out, PCIe included) = %.1f M anchors/s\n",
	        (long long)sh.tot_reads, (long long)sh.tot_anchors, sh.t_gpu, sh.t_gpu > 0 ? sh.tot_anchors / sh.t_gpu / 1e6 : 0.0);
	{
		mm2c_stage_stats_t st;
		mm2c_get_stage_stats(&st);
		fprintf(stderr, "[mm2_batchhost] inside the library (mm2c_get_stage_stats): %llu calls in %llu chunks, %.3f s wall; host: setup %.3f, alloc %.3f (%llu), free %.3f (%llu), "
		        "blocked %.3f; device (chunks overlap): upload %.3f, seed hits %.3f, DP %.3f, epilogue %.3f, offsets down %.3f; index pool upload %.2f s\n",
		        (unsigned long long)st.calls, (unsigned long long)st.chunks, st.total_ns * 1e-9, st.setup_ns * 1e-9, st.alloc_ns * 1e-9, (unsigned long long)st.n_alloc,
		        st.free_ns * 1e-9, (unsigned long long)st.n_free, st.wait_ns * 1e-9, st.h2d_ns * 1e-9, st.seed_ns * 1e-9, st.dp_ns * 1e-9, st.epi_ns * 1e-9, st.d2h_ns * 1e-9, t_pool);
	}
	tt = realtime();
	mm2c_shutdown();
	fprintf(stderr, "[mm2_batchhost] HIP start-up (mm2c_init) %.2f s, shut-down %.2f s, whole process %.2f s\n", t_init, realtime() - tt, realtime() - mm_realtime0);
	return fflush(stdout) == EOF;
}
