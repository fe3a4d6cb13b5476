/* seed_dump.c -- test infrastructure: writes, for every read of <query.fa>, the seed matches that collect_seed_hits
 * (map.c:215-247) expands into anchors, computed with the reference's own sketch / index objects:
 *   minimizers  = mm_sketch (sketch.c) as collect_minimizers calls it for one segment (map.c:65-71; sdust_thres = 0 for map-ont)
 *   matches     = collect_matches restated (map.c:84-120): mm_idx_get per minimizer, occurrences >= mid_occ dropped,
 *                 is_tandem from equal neighbouring minimizers
 * Together with the anchor lists the real map.o hands to mm_chain_dp for the same files (MM2O_DUMP of mm2_refhost) this pins the
 * seed-hit path: matches in, reference anchors out.
 * Output per read with at least one hit: int32 qlen, int32 n_m, n_m x {uint32 n, q_pos, q_span, seg_tandem}, then the hits
 * (uint64 each) of all matches in order.
 * With `-x ava-ont` (k = 15, w = 5, options.c:83) the file starts with what skip_seed (map.c:122-147) needs to know about names, in a form
 * that does not need the names: int32 n_ref, then per reference sequence {int32 rank of its name among the distinct reference names in
 * strcmp order, int32 length}; and every read record carries two more int32 after n_m: q_lo = number of distinct reference names that
 * are < the read's name (so strcmp(qname, name[rid]) > 0 <=> rank[rid] < q_lo), q_eq = 1 when the read's name is a reference name
 * (then strcmp == 0 <=> rank[rid] == q_lo).
 * usage: seed_dump [-x ava-ont] <ref.fa> <query.fa> <out.bin> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "minimap.h"
#include "mmpriv.h"
#include "bseq.h"
#include "kalloc.h"

void mm_idxopt_init(mm_idxopt_t *io)                                    /* options.c:8-15, imported by index.c:569 */
{
	memset(io, 0, sizeof(*io));
	io->k = 15; io->w = 10; io->flag = 0; io->bucket_bits = 14;
	io->mini_batch_size = 50000000; io->batch_size = 4000000000ULL;
}

int main(int argc, char *argv[])
{
	mm_idxopt_t io;
	mm_idx_reader_t *r;
	mm_idx_t *mi;
	FILE *out;
	int ava = 0;
	if (argc >= 6 && strcmp(argv[1], "-x") == 0 && strcmp(argv[2], "ava-ont") == 0) { ava = 1; argv += 2; argc -= 2; }
	if (argc < 4) { fprintf(stderr, "usage: %s [-x ava-ont] <ref.fa> <query.fa> <out.bin>\n", argv[0]); return 1; }
	mm_verbose = 1;
	mm_idxopt_init(&io);
	if (ava) io.w = 5;                                                    /* options.c:83 */
	io.flag |= MM_I_NO_SEQ;
	r = mm_idx_reader_open(argv[1], &io, 0);
	if (!r) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
	out = fopen(argv[3], "wb");
	while ((mi = mm_idx_reader_read(r, 1)) != 0) {
		int mid_occ = mm_idx_cal_max_occ(mi, 2e-4f);                      /* options.c:21,62-63: mid_occ_frac */
		int32_t *rank = 0, n_dist = 0;
		const char **dist = 0;
		if (ava) {
			/* distinct reference names in strcmp order; rank of every sequence's name in that list */
			int32_t n_ref = (int32_t)mi->n_seq, a, b;
			int32_t *ord = (int32_t *)malloc(n_ref * sizeof(int32_t));
			rank = (int32_t *)malloc(n_ref * sizeof(int32_t));
			dist = (const char **)malloc(n_ref * sizeof(char *));
			for (a = 0; a < n_ref; ++a) ord[a] = a;
			for (a = 1; a < n_ref; ++a) {                                 /* insertion sort: the fixture sets are small */
				int32_t v = ord[a];
				for (b = a; b > 0 && strcmp(mi->seq[ord[b - 1]].name, mi->seq[v].name) > 0; --b) ord[b] = ord[b - 1];
				ord[b] = v;
			}
			for (a = 0; a < n_ref; ++a) {
				if (a == 0 || strcmp(mi->seq[ord[a]].name, mi->seq[ord[a - 1]].name) != 0) dist[n_dist++] = mi->seq[ord[a]].name;
				rank[ord[a]] = n_dist - 1;
			}
			fwrite(&n_ref, 4, 1, out);
			for (a = 0; a < n_ref; ++a) { int32_t len = (int32_t)mi->seq[a].len; fwrite(&rank[a], 4, 1, out); fwrite(&len, 4, 1, out); }
			free(ord);
		}
		mm_bseq_file_t *fp = mm_bseq_open(argv[2]);
		int n_seq, i;
		mm_bseq1_t *seqs;
		while ((seqs = mm_bseq_read(fp, 500000000, 0, &n_seq)) != 0) {
			for (i = 0; i < n_seq; ++i) {
				mm128_v mv = {0, 0, 0};
				size_t j;
				int32_t n_m = 0, qlen = seqs[i].l_seq;
				int64_t n_a = 0;
				uint32_t *rec;
				const uint64_t **crs;
				mm_sketch(0, seqs[i].seq, qlen, mi->w, mi->k, 0, mi->flag & MM_I_HPC, &mv);   /* map.c:69 */
				rec = (uint32_t *)malloc((mv.n + 1) * 16);
				crs = (const uint64_t **)malloc((mv.n + 1) * sizeof(*crs));
				for (j = 0; j < mv.n; ++j) {                              /* map.c:95-118 */
					const mm128_t *p = &mv.a[j];
					int t;
					const uint64_t *cr = mm_idx_get(mi, p->x >> 8, &t);
					uint32_t is_tandem = 0;
					if (t >= mid_occ) continue;                           /* map.c:104-110 (only rep_len is updated there) */
					if (j > 0 && p->x >> 8 == mv.a[j - 1].x >> 8) is_tandem = 1;
					if (j < mv.n - 1 && p->x >> 8 == mv.a[j + 1].x >> 8) is_tandem = 1;
					rec[4 * n_m] = (uint32_t)t; rec[4 * n_m + 1] = (uint32_t)p->y; rec[4 * n_m + 2] = p->x & 0xff;
					rec[4 * n_m + 3] = (uint32_t)(p->y >> 32) << 1 | is_tandem;
					crs[n_m++] = cr;
					n_a += t;
				}
				if (n_a > 0 || ava) {                                       /* all-vs-all fixtures keep every read (it may lose all its hits to skip_seed) */
					int32_t k;
					fwrite(&qlen, 4, 1, out); fwrite(&n_m, 4, 1, out);
					if (ava) {
						int32_t q_lo = 0, q_eq = 0;
						while (q_lo < n_dist && strcmp(dist[q_lo], seqs[i].name) < 0) ++q_lo;
						q_eq = q_lo < n_dist && strcmp(dist[q_lo], seqs[i].name) == 0;
						fwrite(&q_lo, 4, 1, out); fwrite(&q_eq, 4, 1, out);
					}
					fwrite(rec, 16, (size_t)n_m, out);
					for (k = 0; k < n_m; ++k) fwrite(crs[k], 8, rec[4 * k], out);
				}
				free(rec); free(crs); kfree(0, mv.a);
				free(seqs[i].seq); free(seqs[i].name);
				if (seqs[i].qual) free(seqs[i].qual);
				if (seqs[i].comment) free(seqs[i].comment);
			}
			free(seqs);
		}
		mm_bseq_close(fp);
		free(rank); free(dist);
		mm_idx_destroy(mi);
	}
	mm_idx_reader_close(r);
	fclose(out);
	return 0;
}
