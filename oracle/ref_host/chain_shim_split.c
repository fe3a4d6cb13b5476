/* chain_shim_split.c -- test infrastructure: the CALLER's side of INTEGRATION.md path A, restated.  The reference's mm_chain_dp (chain.c:29-423) runs a prediction
 * pass (chain.c:53-81), sends the read to the device when hw_time_pred < sw_time_pred (chain.c:101,103) and chains it on the calling thread when the model says so
 * or when the device's answer is 1 = "busy, declined" (chain.c:106,112-164; chain_hardware.cpp:54-75).  chain.c itself cannot be compiled in this image (it includes
 * chain_hardware.h -> xcl2.hpp -> <CL/cl_ext_xilinx.h>), so this file restates that control flow over
 *   - the repo's CPU oracle for everything chain.c does on the host: the prediction pass (mm2o_predict), the software loop (mm2o_chain_fpv), v[] after a device run
 *     (mm2o_fill_v) and the backtrack (mm2o_backtrack);
 *   - the PRODUCT library for the device branch: mm2c_chain_task_host_pred, the extended form of run_chaining_on_hw (stock-CPU semantics + the two predictions the busy
 *     protocol needs), with the MI355X constants of include/mm2chain_split.h (mm2c_split_model).
 * It is what a minimap2 host looks like that keeps the reference's HW/SW split on an MI355X: small reads stay on the CPU threads, big ones go to the GPU, a busy device
 * hands reads back.  Both branches compute the V1 recurrence, so the PAF must equal the CPU-chaining host's byte for byte.  The product itself never falls back to a CPU:
 * the software loop here is the caller's, as in the reference. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "minimap.h"
#include "mmpriv.h"
#include "kalloc.h"
#include "chain_oracle.h"
#define MM2C_NO_MM_CHAIN_DP_DECL
#include "mm2chain.h"

static float K1_HW, K2_HW, C_HW, K_SW, C_SW;                 /* options.c:6; set from the library's constants at the first call */
static int g_have_model;
static long long g_hw, g_declined, g_sw_model, g_anchors_hw, g_anchors_sw;

static void report(void)
{
	fprintf(stderr, "[mm2_splithost] split model K1_HW %.3g K2_HW %.3g C_HW %.3g K_SW %.3g C_SW %.3g: %lld reads on the device (%lld anchors), %lld kept on the CPU by the model + %lld declined by a busy device (%lld anchors)\n",
	        K1_HW, K2_HW, C_HW, K_SW, C_SW, g_hw, g_anchors_hw, g_sw_model, g_declined, g_anchors_sw);
}

mm128_t *mm_chain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc, float gap_scale,
                     int is_cdna, int n_segs, int64_t n, mm128_t *a, int *n_u_, uint64_t **_u, void *km, int tid)
{
	mm2o_params_t opar = { max_dist_x, max_dist_y, bw, max_skip, max_iter, gap_scale, is_cdna, n_segs };
	mm2c_params_t par;
	int32_t *f, *p, *v, *t, n_u;
	uint8_t *ns;
	uint64_t *u = 0;
	mm2o_anchor_t *b = 0;
	int64_t n_b = 0, total_trip = 0, total_sub;
	float avg, hw_pred, sw_pred;
	int rc = 1;
	mm128_t *ret = 0;
	if (_u) *_u = 0, *n_u_ = 0;
	if (n == 0 || a == 0) { kfree(km, a); return 0; }                  /* chain.c:37-41 */
	if (!g_have_model) {                                                /* (racy first calls all write the same values) */
		if (mm2c_split_model("map-ont", &K1_HW, &K2_HW, &C_HW, &K_SW, &C_SW) != 0) { fprintf(stderr, "ERROR: %s\n", mm2c_last_error()); exit(1); }
		if (getenv("MM2_SPLIT_ALL_HW")) C_HW = -1e30f;                 /* everything to the device */
		g_have_model = 1; atexit(report);
	}
	f = (int32_t *)kmalloc(km, (size_t)n * 4); p = (int32_t *)kmalloc(km, (size_t)n * 4); v = (int32_t *)kmalloc(km, (size_t)n * 4);
	t = (int32_t *)kmalloc(km, (size_t)n * 4); ns = (uint8_t *)kmalloc(km, (size_t)n);
	avg = mm2o_avg_qspan_scaled(n, (const mm2o_anchor_t *)a);           /* chain.c:48-49 */
	total_sub = mm2o_predict(n, (const mm2o_anchor_t *)a, max_dist_x, ns, &total_trip);   /* chain.c:53-78 */
	hw_pred = K1_HW * n + K2_HW * total_sub + C_HW;                     /* chain.c:80 */
	sw_pred = K_SW * total_trip + C_SW;                                 /* chain.c:81 */
	if (hw_pred < sw_pred) {                                            /* chain.c:101 */
		par.max_dist_x = max_dist_x; par.max_dist_y = max_dist_y; par.bw = bw; par.max_skip = max_skip; par.max_iter = max_iter; par.gap_scale = gap_scale;
		par.is_cdna = is_cdna; par.n_segs = n_segs; par.q_span_override = -1; par.flags = 0;
		rc = mm2c_chain_task_host_pred(&par, n, (const mm2c_anchor_t *)a, avg, f, p, tid, hw_pred, sw_pred);   /* chain.c:103 */
		if (rc < 0) { fprintf(stderr, "Error: GPU chaining failed (n = %ld): %s\n", (long)n, mm2c_last_error()); exit(EXIT_FAILURE); }
		if (rc == 0) { mm2o_fill_v(n, f, p, v); __sync_fetch_and_add(&g_hw, 1); __sync_fetch_and_add(&g_anchors_hw, n); }   /* chain.c:106-111 */
		else __sync_fetch_and_add(&g_declined, 1);
	} else __sync_fetch_and_add(&g_sw_model, 1);
	if (rc != 0) { mm2o_chain_fpv(&opar, n, (const mm2o_anchor_t *)a, avg, f, p, v, t); __sync_fetch_and_add(&g_anchors_sw, n); }   /* chain.c:112-164 */
	n_u = mm2o_backtrack(n, (const mm2o_anchor_t *)a, min_cnt, min_sc, f, p, v, t, &u, &b, &n_b);   /* chain.c:348-422 */
	kfree(km, f); kfree(km, p); kfree(km, v); kfree(km, t); kfree(km, ns);
	kfree(km, a);                                                        /* chain.c:421: the callee owns a */
	if (n_u > 0) {
		uint64_t *uk = (uint64_t *)kmalloc(km, (size_t)n_u * 8);
		ret = (mm128_t *)kmalloc(km, (size_t)n_b * sizeof(mm128_t));
		memcpy(uk, u, (size_t)n_u * 8);
		memcpy(ret, b, (size_t)n_b * sizeof(mm128_t));
		*n_u_ = n_u, *_u = uk;
	}
	free(u); free(b);
	return ret;
}
