/* driver.c -- test infrastructure: `minimap2 -x map-ont -t 1 <ref.fa> <query.fa>` without main.c / options.c (which need
 * the Xilinx header).  Option values restated from options.c:8-57 (defaults) and :93-94 (map-ont: k=15, flag 0); flow as
 * main.c:286-287,371-410; `-x ava-ont` adds options.c:82-86.  Links the reference's own index / sketch / map / hit / format objects. */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "minimap.h"
#include "mmpriv.h"
#ifdef MM2_GPU_CHAINING
#define MM2C_NO_MM_CHAIN_DP_DECL
#include "mm2chain.h"
#endif

/* symbols of options.c that the library objects import */
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

static void defaults(mm_idxopt_t *io, mm_mapopt_t *mo)
{
	mm_idxopt_init(io); memset(mo, 0, sizeof(*mo));
	mo->seed = 11; mo->mid_occ_frac = 2e-4f; mo->sdust_thres = 0;          /* options.c:20-22 */
	mo->min_cnt = 3; mo->min_chain_score = 40; mo->bw = 500; mo->max_gap = 5000; mo->max_gap_ref = -1;   /* :24-28 */
	mo->max_chain_skip = 25; mo->max_chain_iter = 5000; mo->chain_gap_scale = 1.0f;                       /* :29-31 */
	mo->mask_level = 0.5f; mo->mask_len = INT_MAX; mo->pri_ratio = 0.8f; mo->best_n = 5;                  /* :33-36 */
	mo->max_join_long = 20000; mo->max_join_short = 2000; mo->min_join_flank_sc = 1000; mo->min_join_flank_ratio = 0.5f;
	mo->alt_drop = 0.15f;
	mo->a = 2; mo->b = 4; mo->q = 4; mo->e = 2; mo->q2 = 24; mo->e2 = 1; mo->sc_ambi = 1;                 /* :45-46 */
	mo->zdrop = 400; mo->zdrop_inv = 200; mo->end_bonus = -1; mo->min_dp_max = mo->min_chain_score * mo->a;
	mo->min_ksw_len = 200; mo->anchor_ext_len = 20; mo->anchor_ext_shift = 6; mo->max_clip_ratio = 1.0f;
	mo->mini_batch_size = 500000000; mo->pe_ori = 0; mo->pe_bonus = 33;                                    /* :53-56 */
}

int main(int argc, char *argv[])
{
	mm_idxopt_t io;
	mm_mapopt_t mo;
	mm_idx_reader_t *r;
	mm_idx_t *mi;
	int n_threads = 1, ava = 0;
	if (argc >= 5 && strcmp(argv[1], "-x") == 0 && strcmp(argv[2], "ava-ont") == 0) { ava = 1; argv += 2; argc -= 2; }   /* main.c -x */
	if (argc >= 5 && strcmp(argv[1], "-t") == 0) { n_threads = atoi(argv[2]); argv += 2; argc -= 2; }   /* main.c:153 -t */
	if (argc < 3) { fprintf(stderr, "usage: %s [-x ava-ont] [-t threads] <ref.fa> <query.fa>\n", argv[0]); return 1; }
	mm_verbose = 1;
#ifdef MM2_GPU_CHAINING
	setenv("GPU_MAX_HW_QUEUES", "16", 0);   /* the host's own environment, before its first HIP call: the library's pipelines want a hardware queue per stream */
	{
		double t0 = realtime();
		/* hardware_init's place (main.c:367); the runtime start-up runs beside the index loading below, the first chaining call waits for it (MM2_SYNC_INIT: the blocking form) */
		if ((getenv("MM2_SYNC_INIT") ? mm2c_init(-1) : mm2c_init_async(-1)) != 0) { fprintf(stderr, "ERROR: %s\n", mm2c_last_error()); return 1; }
		if (getenv("MM2_TIMING")) fprintf(stderr, "[mm2_gpuhost] mm2c_init: %.3f s\n", realtime() - t0);
	}
#endif
	if (getenv("MM2_PRINT_SEEDS")) mm_dbg_flag |= MM_DBG_PRINT_SEED;   /* main.c:193 --print-seeds */
	mm_realtime0 = realtime();
	defaults(&io, &mo);
	if (getenv("MM2_MINI_BATCH")) mo.mini_batch_size = atoll(getenv("MM2_MINI_BATCH"));   /* main.c -K */
	io.flag = 0; io.k = 15;                              /* -x map-ont, options.c:93-94 */
	if (ava) {                                           /* -x ava-ont, options.c:82-86 */
		io.w = 5;
		mo.flag |= MM_F_ALL_CHAINS | MM_F_NO_DIAG | MM_F_NO_DUAL | MM_F_NO_LJOIN;
		mo.min_chain_score = 100; mo.pri_ratio = 0.0f; mo.max_gap = 10000; mo.max_chain_skip = 25; mo.bw = 2000;
	}
	if (getenv("MM2_HEAP_SORT")) mo.flag |= MM_F_HEAP_SORT;   /* main.c:245 --heap-sort=yes (collect_seed_hits_heap, map.c:149-213) */
	io.flag |= MM_I_NO_SEQ;                              /* main.c:286-287: no -d, no CIGAR */
	r = mm_idx_reader_open(argv[1], &io, 0);
	if (!r) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
	while ((mi = mm_idx_reader_read(r, n_threads)) != 0) {
		mm_mapopt_update(&mo, mi);                       /* main.c:399 */
		if (mm_map_file(mi, argv[2], &mo, n_threads) < 0) { fprintf(stderr, "mapping failed\n"); return 1; }   /* main.c:406 */
		mm_idx_destroy(mi);
	}
	mm_idx_reader_close(r);
#ifdef MM2_GPU_CHAINING
	{ mm2c_stats_t st; mm2c_get_stats(&st); fprintf(stderr, "[mm2_gpuhost] GPU chaining: %llu tasks, %llu anchors, %llu pieces, %llu passes, %llu launches, %.3f s inside the chaining calls (summed over threads), %.1f us per call, %.2f calls per pass\n",
	                                                 (unsigned long long)st.tasks, (unsigned long long)st.anchors, (unsigned long long)st.segments, (unsigned long long)st.passes, (unsigned long long)st.launches,
	                                                 st.host_call_ns * 1e-9, st.tasks ? st.host_call_ns * 1e-3 / st.tasks : 0.0, st.passes ? (double)st.tasks / st.passes : 0.0); }
	{
		int s, nd = mm2c_device_count();
		fprintf(stderr, "[mm2_gpuhost] per device slot:");
		for (s = 0; s < nd; ++s) {
			mm2c_slot_stats_t ss;
			if (mm2c_get_slot_stats(s, &ss) == 0) fprintf(stderr, " slot %d (device %d) %llu passes %llu calls %llu anchors;", s, ss.device, (unsigned long long)ss.passes, (unsigned long long)ss.calls, (unsigned long long)ss.anchors);
		}
		fprintf(stderr, "\n");
	}
	{
		double t0 = realtime();
		mm2c_shutdown();                                 /* cleanup, main.c:430 */
		if (getenv("MM2_TIMING")) fprintf(stderr, "[mm2_gpuhost] mm2c_shutdown: %.3f s; %.3f s since start\n", realtime() - t0, realtime() - mm_realtime0);
	}
#endif
	return fflush(stdout) == EOF;
}
