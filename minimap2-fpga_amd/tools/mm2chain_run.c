/*
 * mm2chain_run -- batched caller for anchor streams: what worker_for (map.c:427) looks like once
 * seeding and chaining are decoupled.  Reads an MM2ANCH1 file (or a `minimap2 --print-seeds` text dump with -t), runs
 * every task through the GPU DP in mini-batches (cf. mini_batch_size, map.c:530) and writes f[] / p[] or the chains.
 *
 *   mm2chain_run [-t] [-b max_anchors_per_batch] [-c] [-e threads] [-o out.bin] <stream>
 *     -t   the input is a --print-seeds text dump (RS / SD lines), chained with map-ont parameters
 *     -c   run the whole mm_chain_dp per mini-batch (mm2c_mm_chain_dp_batch_host: DP and backtrack on the GPU) and print the
 *          chains as "CH\t<task>\t<score>\t<n_anchors>" lines
 *     -e   with -c: run the backtrack on that many host threads instead (the reference's arrangement)
 *     -o   write int32 f[total] then int32 p[total]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mm2chain.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
	int i, text = 0, chains = 0, epi_threads = 0, rc;
	int64_t batch = 64 << 20, k0, done = 0;
	const char *out_path = 0, *in_path = 0;
	mm2c_stream_t s;
	int32_t *f, *p;
	double t0, t_gpu = 0;
	for (i = 1; i < argc; ++i) {
		if (strcmp(argv[i], "-t") == 0) text = 1;
		else if (strcmp(argv[i], "-c") == 0) chains = 1;
		else if (strcmp(argv[i], "-b") == 0 && i + 1 < argc) batch = atoll(argv[++i]);
		else if (strcmp(argv[i], "-e") == 0 && i + 1 < argc) epi_threads = atoi(argv[++i]);
		else if (strcmp(argv[i], "-o") == 0 && i + 1 < argc) out_path = argv[++i];
		else in_path = argv[i];
	}
	if (!in_path) { fprintf(stderr, "usage: mm2chain_run [-t] [-b max_anchors_per_batch] [-c] [-e threads] [-o out.bin] <stream>\n"); return 2; }
	if (text) { mm2c_params_t par; mm2c_params_map_ont(&par); rc = mm2c_stream_from_seed_dump(in_path, &par, 3, 40, &s); }
	else rc = mm2c_stream_read(in_path, &s);
	if (rc != 0) { fprintf(stderr, "cannot read %s\n", in_path); return 1; }
	if (mm2c_init(-1) != 0) { fprintf(stderr, "ERROR: %s\n", mm2c_last_error()); return 1; }   /* no GPU -> fail, never a CPU path */
	f = (int32_t *)malloc((size_t)(s.total ? s.total : 1) * 4);
	p = (int32_t *)malloc((size_t)(s.total ? s.total : 1) * 4);
	t0 = now();
	for (k0 = 0; k0 < s.n_tasks; ) {                       /* mini-batches of whole tasks */
		int64_t k1 = k0 + 1;
		double t1;
		while (k1 < s.n_tasks && s.offsets[k1 + 1] - s.offsets[k0] <= batch) ++k1;
		t1 = now();
		rc = mm2c_chain_batch_host(&s.par, k1 - k0, s.offsets + k0, s.anchors, 0, f, p);
		t_gpu += now() - t1;
		if (rc != 0) { fprintf(stderr, "ERROR: %s\n", mm2c_last_error()); return 1; }
		done += s.offsets[k1] - s.offsets[k0];
		k0 = k1;
	}
	{
		uint64_t h = 1469598103934665603ULL;              /* FNV-1a over f then p: a checksum to compare runs */
		int64_t j;
		for (j = 0; j < s.total; ++j) { h = (h ^ (uint32_t)f[j]) * 1099511628211ULL; h = (h ^ (uint32_t)p[j]) * 1099511628211ULL; }
		fprintf(stderr, "[mm2chain_run] %lld tasks, %lld anchors, %.3f s in the chaining calls (%.1f M anchors/s incl. PCIe), checksum %016llx\n",
		        (long long)s.n_tasks, (long long)done, t_gpu, t_gpu > 0 ? done / t_gpu / 1e6 : 0.0, (unsigned long long)h);
	}
	if (out_path) {
		FILE *fp = fopen(out_path, "wb");
		if (!fp) { fprintf(stderr, "cannot write %s\n", out_path); return 1; }
		fwrite(f, 4, (size_t)s.total, fp); fwrite(p, 4, (size_t)s.total, fp);
		fclose(fp);
	}
	if (chains) {
		uint64_t *u = (uint64_t *)malloc((size_t)(s.total ? s.total : 1) * 8);
		mm2c_anchor_t *b = (mm2c_anchor_t *)malloc((size_t)(s.total ? s.total : 1) * 16);
		int64_t *u_off = (int64_t *)malloc((size_t)(s.n_tasks + 1) * 8), *b_off = (int64_t *)malloc((size_t)(s.n_tasks + 1) * 8);
		int64_t n_chains = 0, n_chained = 0;
		double t_ch = 0;
		for (k0 = 0; k0 < s.n_tasks; ) {
			int64_t k1 = k0 + 1, k, c;
			double t1;
			while (k1 < s.n_tasks && s.offsets[k1 + 1] - s.offsets[k0] <= batch) ++k1;
			t1 = now();
			rc = mm2c_mm_chain_dp_batch_host(&s.par, s.min_cnt, s.min_sc, k1 - k0, s.offsets + k0, s.anchors, epi_threads, u_off, u, b_off, b);
			t_ch += now() - t1;
			if (rc != 0) { fprintf(stderr, "ERROR: %s\n", mm2c_last_error()); return 1; }
			for (k = k0; k < k1; ++k)
				for (c = u_off[k - k0]; c < u_off[k - k0 + 1]; ++c) printf("CH\t%lld\t%d\t%d\n", (long long)k, (int)(u[c] >> 32), (int)(uint32_t)u[c]);
			n_chains += u_off[k1 - k0]; n_chained += b_off[k1 - k0];
			k0 = k1;
		}
		fprintf(stderr, "[mm2chain_run] whole mm_chain_dp (%s): %lld chains, %lld chained anchors, %.3f s (%.1f M anchors/s incl. PCIe)\n",
		        epi_threads > 0 ? "backtrack on host threads" : "backtrack on the GPU", (long long)n_chains, (long long)n_chained, t_ch,
		        t_ch > 0 ? s.total / t_ch / 1e6 : 0.0);
		free(u); free(b); free(u_off); free(b_off);
	}
	fprintf(stderr, "[mm2chain_run] total %.3f s\n", now() - t0);
	free(f); free(p);
	mm2c_stream_free(&s);
	mm2c_shutdown();
	return 0;
}
