// chain_kernel.h -- internal interface between the C-ABI shim (mm2chain_api.cpp, mm2chain_host.cpp, mm2chain_seeds.cpp) and the HIP kernels.
#ifndef MM2C_CHAIN_KERNEL_H
#define MM2C_CHAIN_KERNEL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mm2c {

enum { KF_IGNORE_SEG = 0x1, KF_FORCE_GENERAL = 0x2 };

// scalars of one mm_chain_dp call (mmpriv.h:65), passed by value to the kernel
struct KParams {
	int32_t max_dist_x, max_dist_y, bw;
	int32_t max_skip, max_iter;
	int32_t is_cdna, n_segs;
	int32_t span_override;   // >= 0: one q_span for every anchor (chain_hardware.h:68 `q_span`)
	int32_t max_dq;          // min(max_dist_y, max_dist_x)
	int32_t flags;           // KF_*
	float gap_scale;
};

// pieces cut on the device (chain_cut): every array has room for max_pieces = n_tasks + total / seg_min entries; d_count and d_status must
// be zero on entry.  max_pieces == 0: tasks are run as they are.
struct CutArgs {
	int64_t max_pieces = 0;
	int32_t seg_min = 256;
	int32_t min_anchors = 8192;     // only tasks at least this long are cut: the others do not make the tail
	int64_t *d_start = nullptr, *d_end = nullptr;
	int32_t *d_pbase = nullptr, *d_status = nullptr, *d_count = nullptr;
	int32_t *d_has_cut = nullptr;   // per task (n_tasks entries), zero on entry: set by the prepass where a window is empty
	float *d_avg = nullptr;
};

struct LaunchArgs {
	KParams P;
	int64_t n_tasks;
	const int64_t *d_offsets;   // n_tasks+1, CSR
	const int32_t *d_order;     // launch order (longest task first) or nullptr
	const void *d_anchors;      // 16 B per anchor
	const float *d_avg;         // per task or nullptr (computed on the device, chain.c:48-49)
	float *d_avg_ws = nullptr;  // n_tasks floats of workspace: when d_avg is nullptr the prepass computes avg_qspan_scaled into it
	const int32_t *d_pbase;     // per task or nullptr: added to every p >= 0 on output (tasks that are pieces of a caller's task)
	int32_t *d_f, *d_p;
	int32_t *d_t;               // stamp scratch for look-back beyond the LDS ring, one int per anchor
	int32_t *d_st;              // window start per anchor (chain.c:192-193), filled by the prepass kernel
	int32_t *d_status;          // per task, must be zero on entry
	int ring_class;             // 3: tile kernel (4 register tiles + 3 LDS tiles); 0 / 1 / 2: first-generation kernel with 256 / 512 / 1024 anchors of LDS ring
	CutArgs cut;                // plans: cut the tasks into independent pieces on the device first
};

int chain_ring_anchors(int ring_class);
// chain.c:53-78 for a CSR batch: per-anchor sub-part counts, per-task totals (any output may be nullptr)
hipError_t launch_predict(int32_t max_dist_x, int64_t n_tasks, const int64_t *d_offsets, const int32_t *d_order, const void *d_anchors,
                          uint8_t *d_num_subparts, int64_t *d_total_subparts, int64_t *d_total_trip, hipStream_t st);
// ev_dp_begin (optional) is recorded between the window-start prepass and the DP kernel
hipError_t launch_chain_dp(const LaunchArgs &L, hipStream_t st, int *n_launches, hipEvent_t ev_dp_begin);

// ---- device epilogue of mm_chain_dp (chain.c:106-111,348-422), chain_epilogue.hip ----
struct EpiArgs {
	int64_t n_tasks, total;
	const int64_t *d_off;       // n_tasks+1, CSR (task 0 at 0)
	const int32_t *d_order;     // longest task first, or nullptr
	const ulonglong2 *d_a;      // anchors
	const int32_t *d_f, *d_p;   // DP result
	int32_t min_cnt, min_sc;
	int32_t debug_phases;       // development aid (MM2C_EPI_PHASES): kernels return after that many of their phases; 0 = run everything
	// scratch, `total` entries each unless noted
	int32_t *v;                 // v[] (chain.c:106-111), later the depth of an anchor inside its chain
	int32_t *own;               // child marks, later the rank of the chain that takes the anchor
	int32_t *ctop, *rk2kk, *dest, *val0, *val1;
	uint64_t *key0, *key1;      // chain-end keys (f[peak]<<32 | peak) unsorted / sorted; key0 is reused for the first-x keys
	uint64_t *u2, *rkey1;       // score<<32|count per kept chain; first-x keys sorted
	uint32_t *seg_begin, *seg_end1, *seg_end2;   // per task: segment bounds for the two sorts
	int32_t *cnt_u, *cnt_b;     // per task: kept chains, their anchors
	void *sort_tmp; size_t sort_tmp_bytes;
	// outputs (compact): chains of task k are u_out[u_off[k] .. u_off[k+1]), their anchors b_out[b_off[k] .. b_off[k+1])
	int64_t *u_off, *b_off;     // n_tasks+1 each
	uint64_t *u_out;
	ulonglong2 *b_out;
};
size_t epilogue_sort_temp_bytes(int64_t total, int64_t n_tasks);
hipError_t launch_chain_epilogue(const EpiArgs &A, hipStream_t st, int *n_launches);

// ---- seed hits -> sorted anchors (collect_seed_hits, map.c:215-247), seed_hits.hip ----
struct Match { int64_t cr_off; uint32_t n, q_pos, q_span, seg_tandem; };   // = mm2c_match_t
struct SeedArgs {
	int64_t n_reads;
	const int64_t *d_match_off, *d_anchor_off;   // n_reads+1 each
	const int32_t *d_order;                      // reads by anchor count, biggest first, or nullptr
	const Match *d_matches;
	const uint64_t *d_hits;                      // the hit pool the matches point into
	int64_t n_hits = 0;                          // its length when the caller declared it (0: matches are trusted)
	// skip_seed (map.c:122-147): flag = MM_F_NO_DIAG 0x1 | MM_F_NO_DUAL 0x2 | MM_F_FOR_ONLY 0x100000 | MM_F_REV_ONLY 0x200000 (0: every hit is kept);
	// names as ranks: per reference sequence the rank of its name and its length, per read q_lo / q_eq (include/mm2chain.h)
	int32_t skip_flag = 0;
	const int32_t *d_ref_rank = nullptr, *d_ref_len = nullptr, *d_q_lo = nullptr, *d_q_eq = nullptr;
	int32_t *d_count = nullptr;                  // per read: anchors kept (nullptr: all, the plan's offsets are exact)
	int64_t *d_out_off = nullptr;                // n_reads + 1: where the packed result of each read starts (written by seed_offsets)
	const int32_t *d_qlen;
	ulonglong2 *unsorted, *scratch;              // anchors in expansion order; second buffer of the sorts
	ulonglong2 *d_anchors;                       // out
	int32_t *status, *has_ties;                  // per read; status must be zero on entry (1: hit counts and anchor offsets disagree)
	uint64_t *xdiff;                             // per read: the bits of x that differ among its anchors (written by seed_expand)
	int32_t *tiecnt;                             // per anchor: equal-x neighbours before this position of the sorted read
	int64_t biggest;                             // anchors of the longest read
	int32_t *stack;                              // pending buckets of the tie replay: 4 * (total / 64 + 2 * n_reads + 2) ints
	uint32_t *big_id; uint8_t *big_dg;           // replay arrays for reads too long for the LDS (nullptr when there is none)
};
int seed_tie_lds_max();
// aux: three helper streams (or nullptr: everything on st), ev: four events without timing
hipError_t launch_seed_hits(const SeedArgs &A, hipStream_t st, int *n_launches, hipStream_t *aux, hipEvent_t *ev);

} // namespace mm2c
#endif
