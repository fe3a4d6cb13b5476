/*
 * anchor_stream.c -- on-disk anchor streams: the unit of work of the batched caller.
 *
 * A stream is a CSR batch of chaining tasks exactly as mm2c_chain_batch_host takes it, plus the scalars of the
 * mm_chain_dp calls it came from.  Layout (little endian):
 *     char     magic[8]  = "MM2ANCH1"
 *     uint32   version   = 1,  uint32 header_bytes = 128
 *     int64    n_tasks, total_anchors
 *     int32    max_dist_x, max_dist_y, bw, max_skip, max_iter, is_cdna, n_segs, min_cnt, min_sc;  float gap_scale
 *     (zero padding up to header_bytes)
 *     int64    offsets[n_tasks + 1]      offsets[0] = 0
 *     mm128_t  anchors[total_anchors]    16 B each, every task sorted by x (map.c:245)
 * The text importer reads what `minimap2 --print-seeds` writes before chaining (map.c:298-303: one "RS" line per read, then
 * one "SD\t<rname>\t<rpos>\t<strand>\t<qpos>\t<span>\t<gap>" line per anchor, already sorted), which is the reference's own
 * tap for golden anchors from real data.  Reference ids are assigned in order of first appearance within the file.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mm2chain.h"

#define HDR_BYTES 128
static const char MAGIC[8] = { 'M', 'M', '2', 'A', 'N', 'C', 'H', '1' };

typedef struct {
	char magic[8];
	uint32_t version, header_bytes;
	int64_t n_tasks, total;
	int32_t max_dist_x, max_dist_y, bw, max_skip, max_iter, is_cdna, n_segs, min_cnt, min_sc;
	float gap_scale;
} hdr_t;

int mm2c_stream_write(const char *path, const mm2c_params_t *par, int min_cnt, int min_sc, int64_t n_tasks,
                      const int64_t *offsets, const mm2c_anchor_t *anchors)
{
	unsigned char buf[HDR_BYTES];
	hdr_t h;
	FILE *fp;
	int64_t k, base, total;
	if (!path || !par || n_tasks < 0 || (n_tasks > 0 && (!offsets || !anchors))) return MM2C_E_ARG;
	base = n_tasks ? offsets[0] : 0;
	total = n_tasks ? offsets[n_tasks] - base : 0;
	memset(&h, 0, sizeof(h)); memset(buf, 0, sizeof(buf));
	memcpy(h.magic, MAGIC, 8); h.version = 1; h.header_bytes = HDR_BYTES; h.n_tasks = n_tasks; h.total = total;
	h.max_dist_x = par->max_dist_x; h.max_dist_y = par->max_dist_y; h.bw = par->bw; h.max_skip = par->max_skip;
	h.max_iter = par->max_iter; h.is_cdna = par->is_cdna; h.n_segs = par->n_segs; h.min_cnt = min_cnt; h.min_sc = min_sc;
	h.gap_scale = par->gap_scale;
	memcpy(buf, &h, sizeof(h));
	if ((fp = fopen(path, "wb")) == 0) return MM2C_E_ARG;
	fwrite(buf, 1, HDR_BYTES, fp);
	for (k = 0; k <= n_tasks; ++k) { int64_t o = (n_tasks ? offsets[k] : 0) - base; fwrite(&o, 8, 1, fp); }
	if (total) fwrite(anchors + base, 16, (size_t)total, fp);
	return fclose(fp) == 0 ? 0 : MM2C_E_ARG;
}

void mm2c_stream_free(mm2c_stream_t *s)
{
	if (!s) return;
	free(s->offsets); free(s->anchors);
	memset(s, 0, sizeof(*s));
}

int mm2c_stream_read(const char *path, mm2c_stream_t *out)
{
	unsigned char buf[HDR_BYTES];
	hdr_t h;
	FILE *fp;
	int64_t k;
	if (!path || !out) return MM2C_E_ARG;
	memset(out, 0, sizeof(*out));
	if ((fp = fopen(path, "rb")) == 0) return MM2C_E_ARG;
	if (fread(buf, 1, HDR_BYTES, fp) != HDR_BYTES) { fclose(fp); return MM2C_E_ARG; }
	memcpy(&h, buf, sizeof(h));
	if (memcmp(h.magic, MAGIC, 8) != 0 || h.version != 1 || h.header_bytes < HDR_BYTES || h.n_tasks < 0 || h.total < 0) { fclose(fp); return MM2C_E_ARG; }
	{
		/* the counts come from the file: they must fit the file (and size_t) before anything is allocated */
		long file_bytes;
		if (fseek(fp, 0, SEEK_END) != 0 || (file_bytes = ftell(fp)) < 0) { fclose(fp); return MM2C_E_ARG; }
		if (h.header_bytes > file_bytes || h.n_tasks > ((int64_t)file_bytes - h.header_bytes) / 8 - 1 ||
		    h.total > ((int64_t)file_bytes - h.header_bytes - (h.n_tasks + 1) * 8) / 16) { fclose(fp); return MM2C_E_ARG; }
		if (fseek(fp, (long)h.header_bytes, SEEK_SET) != 0) { fclose(fp); return MM2C_E_ARG; }
	}
	out->n_tasks = h.n_tasks; out->total = h.total; out->min_cnt = h.min_cnt; out->min_sc = h.min_sc;
	out->par.max_dist_x = h.max_dist_x; out->par.max_dist_y = h.max_dist_y; out->par.bw = h.bw; out->par.max_skip = h.max_skip;
	out->par.max_iter = h.max_iter; out->par.gap_scale = h.gap_scale; out->par.is_cdna = h.is_cdna; out->par.n_segs = h.n_segs;
	out->par.q_span_override = -1; out->par.flags = 0;
	out->offsets = (int64_t *)malloc(((size_t)h.n_tasks + 1) * 8);
	out->anchors = (mm2c_anchor_t *)malloc((size_t)(h.total ? h.total : 1) * 16);
	if (!out->offsets || !out->anchors) { fclose(fp); mm2c_stream_free(out); return MM2C_E_ARG; }
	if (fread(out->offsets, 8, (size_t)h.n_tasks + 1, fp) != (size_t)h.n_tasks + 1 ||
	    fread(out->anchors, 16, (size_t)h.total, fp) != (size_t)h.total) { fclose(fp); mm2c_stream_free(out); return MM2C_E_ARG; }
	fclose(fp);
	if (out->offsets[0] != 0 || out->offsets[h.n_tasks] != h.total) { mm2c_stream_free(out); return MM2C_E_ARG; }
	for (k = 0; k < h.n_tasks; ++k) if (out->offsets[k + 1] < out->offsets[k]) { mm2c_stream_free(out); return MM2C_E_ARG; }
	return 0;
}

typedef struct { mm2c_anchor_t a; int64_t i; } sort_rec_t;
static int cmp_sort_rec(const void *pa, const void *pb)
{
	const sort_rec_t *a = (const sort_rec_t *)pa, *b = (const sort_rec_t *)pb;
	if (a->a.x != b->a.x) return a->a.x < b->a.x ? -1 : 1;
	return a->i < b->i ? -1 : a->i > b->i;
}

int mm2c_stream_from_seed_dump(const char *text_path, const mm2c_params_t *par, int min_cnt, int min_sc, mm2c_stream_t *out)
{
	FILE *fp;
	char line[1024], name[512], strand;
	char **names = 0;
	int n_names = 0, m_names = 0, rpos, qpos, span, gap, started = 0, oom = 0;
	int64_t m_a = 1 << 16, m_t = 1 << 10;
	if (!text_path || !par || !out) return MM2C_E_ARG;
	memset(out, 0, sizeof(*out));
	if ((fp = fopen(text_path, "r")) == 0) return MM2C_E_ARG;
	out->par = *par; out->min_cnt = min_cnt; out->min_sc = min_sc;
	out->anchors = (mm2c_anchor_t *)malloc((size_t)m_a * 16);
	out->offsets = (int64_t *)malloc((size_t)(m_t + 1) * 8);
	if (!out->anchors || !out->offsets) { fclose(fp); mm2c_stream_free(out); return MM2C_E_ARG; }
	out->offsets[0] = 0;
	while (fgets(line, sizeof(line), fp)) {
		if (line[0] == 'R' && line[1] == 'S' && line[2] == '\t') {              /* map.c:299: a new read starts */
			if (started) {
				if (out->n_tasks + 1 >= m_t) {
					void *q = realloc(out->offsets, (size_t)(2 * m_t + 1) * 8);
					if (!q) { oom = 1; break; }
					m_t <<= 1; out->offsets = (int64_t *)q;
				}
				out->offsets[++out->n_tasks] = out->total;
			}
			started = 1;
		} else if (line[0] == 'S' && line[1] == 'D' && line[2] == '\t') {       /* map.c:301 */
			int rid;
			if (sscanf(line + 3, "%511s\t%d\t%c\t%d\t%d\t%d", name, &rpos, &strand, &qpos, &span, &gap) != 6) continue;
			for (rid = 0; rid < n_names; ++rid) if (strcmp(names[rid], name) == 0) break;
			if (rid == n_names) {
				if (n_names == m_names) {
					void *q = realloc(names, (size_t)(m_names ? m_names << 1 : 16) * sizeof(char *));
					if (!q) { oom = 1; break; }
					m_names = m_names ? m_names << 1 : 16; names = (char **)q;
				}
				if ((names[n_names] = strdup(name)) == 0) { oom = 1; break; }
				++n_names;
			}
			if (out->total == m_a) {
				void *q = realloc(out->anchors, (size_t)m_a * 32);
				if (!q) { oom = 1; break; }
				m_a <<= 1; out->anchors = (mm2c_anchor_t *)q;
			}
			out->anchors[out->total].x = (uint64_t)(strand == '-') << 63 | (uint64_t)rid << 32 | (uint32_t)rpos;   /* map.c:232-241 */
			out->anchors[out->total].y = (uint64_t)(span & 0xff) << 32 | (uint32_t)qpos;
			++out->total;
			started = 1;
		}
	}
	fclose(fp);
	if (started && !oom) out->offsets[++out->n_tasks] = out->total;
	while (n_names) free(names[--n_names]);
	free(names);
	if (oom) { mm2c_stream_free(out); return MM2C_E_ARG; }
	/* reference ids are numbered here by first appearance in the dump, not by the index's order: a read that hits several references
	 * may come out with descending ids.  Every chaining entry needs x ascending inside a task (map.c:245), so each task is sorted by x
	 * (stable: the dump's order among equal x is kept).  Segment ids and flag bits are not part of the SD lines and are 0 here. */
	{
		int64_t k;
		for (k = 0; k < out->n_tasks; ++k) {
			mm2c_anchor_t *a = out->anchors + out->offsets[k];
			const int64_t n = out->offsets[k + 1] - out->offsets[k];
			int64_t i, j;
			int sorted = 1;
			for (i = 1; i < n && sorted; ++i) sorted = a[i - 1].x <= a[i].x;
			if (sorted) continue;
			{
				sort_rec_t *r = (sort_rec_t *)malloc((size_t)n * sizeof(sort_rec_t));
				if (!r) { mm2c_stream_free(out); return MM2C_E_ARG; }
				for (i = 0; i < n; ++i) { r[i].a = a[i]; r[i].i = i; }
				qsort(r, (size_t)n, sizeof(sort_rec_t), cmp_sort_rec);
				for (j = 0; j < n; ++j) a[j] = r[j].a;
				free(r);
			}
		}
	}
	return 0;
}
