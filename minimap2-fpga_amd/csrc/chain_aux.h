// chain_aux.h -- argument blocks and launchers of the kernels around the DP: the device epilogue (chain_epilogue.hip) and seed hits -> anchors
// (seed_hits.hip).  Included by chain_kernel.h.
#ifndef MM2C_CHAIN_AUX_H
#define MM2C_CHAIN_AUX_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mm2c {

// ---- device epilogue of mm_chain_dp (chain.c:106-111,348-422), chain_epilogue.hip ----
struct EpiArgs {
	int64_t n_tasks, total;
	const int64_t *d_off;       // n_tasks+1, CSR (task 0 at 0)
	const int32_t *d_order;     // longest task first, or nullptr
	const ulonglong2 *d_a;      // anchors
	const int32_t *d_f, *d_p;   // DP result
	int32_t min_cnt, min_sc;
	int32_t debug_phases;       // development aid (MM2C_EPI_PHASES): kernels return after that many of their phases; 0 = run everything
	int32_t fused = 1;          // tasks that fit the LDS run the passes of kernels A and B in one kernel with their per-anchor state in LDS
	int64_t max_task = -1;      // upper bound of the task sizes when the host knows one (spares the launches of unused size classes)
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
	int32_t heap_order = 0;                      // 1: the order among equal x that collect_seed_hits_heap leaves (map.c:149-213, MM_F_HEAP_SORT) instead of radix_sort_128x's
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
	uint32_t *tie_id;                            // per anchor: the arrangement of the tie replay (position -> anchor of the unsorted array)
	uint8_t *big_dg;                             // digits of the replay for reads too long for the LDS: total + n_reads bytes (nullptr when there is none)
	int32_t lds_sort = 1;                        // reads of up to seed_sort_lds_cap() anchors whose differing x bits fit 32 are sorted in LDS (seed_sort_lds); 0: seed_sort for every read
	int64_t n_sort_big = 0, n_sort_huge = 0;     // reads whose capacity exceeds seed_sort_lds_cap0() / seed_sort_lds_cap(): the first workgroups of the launch order
	int32_t debug_cut = 0;                       // development aid (MM2C_TIE_CUT): seed_ties returns early / skips parts (results are then wrong)
	int64_t n_above[6] = {0, 0, 0, 0, 0, 0};     // reads whose capacity exceeds the lower bound of each class of seed_ties (64, then the class sizes): the grid of that class
	int32_t tie_global_waves = 0;                // experiments / tests: 1, 2, 4 or 8 waves per read of that class whatever their number (0: by their number)
	int32_t tie_global_mw_below = 600;           // the tie replay with the digits in memory: up to this many reads in that class run it on eight waves each (independent buckets of a level side by side), up to 1 280 on four, up to 3 000 on two, more on one wave each
	int32_t mw_sort = 1;                         // reads beyond seed_sort_lds_cap() anchors are sorted by sixteen waves (seed_sort_mw); 0: by one (seed_sort)
	int64_t tie_global_above = 131072;           // the tie replay of reads longer than this keeps its digits in memory, one wave per read (many reads in flight per CU); up to it: LDS classes
};
int seed_tie_lds_max();
int seed_tie_mid_lower();
int seed_sort_lds_cap();
int seed_sort_lds_cap0();
const int64_t *seed_tie_class_lower();          // six lower bounds (exclusive) of the size classes of seed_ties, shortest class first
// aux: three helper streams (or nullptr: everything on st), ev: four events without timing
hipError_t launch_seed_hits(const SeedArgs &A, hipStream_t st, int *n_launches, hipStream_t *aux, hipEvent_t *ev);

// ---- the copies of a small per-read pass as kernels (host_stage.hip): page-locked host staging buffer -> device arena, device results -> page-locked host buffer
// + a flag word the host polls.  bytes are rounded up to 16: both buffers must have that room.  d_done: one unsigned of device scratch.
hipError_t launch_stage_in(const void *h_src, void *d_dst, size_t bytes, unsigned *d_done, hipStream_t st);
hipError_t launch_stage_out(const void *d_src, void *h_dst, size_t bytes, unsigned *d_done, unsigned *h_flag, unsigned seq, hipStream_t st);

// force the runtime to load each translation unit's code object onto the current device now (mm2c_warm_up)
hipError_t warm_chain_kernels();
hipError_t warm_epilogue_kernels();
hipError_t warm_seed_kernels();
hipError_t warm_stage_kernels();

} // namespace mm2c
#endif
