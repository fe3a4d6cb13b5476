// chain_kernel.h -- internal interface between the C-ABI shim (mm2chain_api.cpp, mm2chain_host.cpp, mm2chain_seeds.cpp) and the HIP kernels.
#ifndef MM2C_CHAIN_KERNEL_H
#define MM2C_CHAIN_KERNEL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mm2c {

enum { KF_IGNORE_SEG = 0x1, KF_FORCE_GENERAL = 0x2 };
constexpr int CLS_STAT_SLOTS = 64;   // sets of class counters the tasks of a batch spread their atomic additions over (chain_window_start -> chain_cls_settle)

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
	int32_t *d_pbase = nullptr, *d_status = nullptr, *d_count = nullptr;   // d_count: four words -- [0] pieces cut, [1] / [2] / [3] the pieces chain_route gives to the one-wave kernels / the cooperative kernel of sixteen / of eight waves
	int32_t w8_above = 256;         // chain_route: more pieces than this -> eight waves per piece instead of sixteen (LaunchArgs::coop_w8_above)
	int32_t *d_live = nullptr;      // the count the one-wave kernels go by (nullptr: d_count; under the device-side route: d_count + 1)
	int32_t *d_has_cut = nullptr;   // per task (n_tasks entries), zero on entry: set by the prepass where a window is empty
	float *d_avg = nullptr;
	uint8_t *d_cls = nullptr;       // per piece: ring-size class of its task (nullptr: no classes)
};

struct LaunchArgs {
	KParams P;
	int64_t n_tasks;
	const int64_t *d_offsets;   // n_tasks+1, CSR
	const int32_t *d_order;     // launch order (longest task first) or nullptr
	const void *d_anchors;      // 16 B per anchor
	const float *d_avg;         // per task or nullptr (computed on the device, chain.c:48-49)
	float *d_avg_ws = nullptr;  // n_tasks floats of workspace: when d_avg is nullptr the prepass computes avg_qspan_scaled into it
	uint8_t *d_cls = nullptr;   // n_tasks bytes of workspace, or nullptr: class per task, written by the prepass (tile kernel only): bit 0 long ring, bit 1 the
	                            // task needs the 32-bit x / q ring (its q values span more than the compact ring can tell apart, chain_dp_tile.h Lds<>)
	hipStream_t side = nullptr; // with ev_fork / ev_join: when the batch is split over the 32-bit and the compact instantiations, the 32-bit ones run on this stream
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // beside the compact one (two launches one after the other each end with the GPU part empty)
	int noskip_loop = 1;        // calls whose max-skip exit cannot fire (max_skip >= max_iter) run with max_skip = max_iter - 1 through the hand-written loop (0: the instantiations
	                            // without the max-skip machinery, C++ loop; mm2c_tune("noskip_loop"), the parity tests run both)
	int compact = 1;            // 0: never the compact x / q ring (mm2c_tune("compact_ring", 0); the parity tests run both)
	int q24 = 1;                // 0: the long ring of class-1 tasks keeps its 32-bit slots (mm2c_tune("q24_ring", 0); the parity tests run both); 1: the q24 ring (16 + 24 bits, 7 KB)
	unsigned long long *d_cls_stat = nullptr;   // CLS_STAT_SLOTS sets of four counters, zero on entry (chain_cls_settle), or nullptr
	int wide_pct = 40;          // when the tasks that need the 32-bit x / q ring hold more than this share of the batch's anchors, every task takes it
	int far_thr10 = 7;          // far_ring 1: a task takes the long ring when it expects more than far_thr10 / 10 tiles beyond the short ring per anchor
	int far_ring = 0;           // 0: one ring size; 1: tasks whose scans are expected to leave the 448-anchor ring get the long ring; 2: every task gets it
	const int32_t *d_pbase;     // per task or nullptr: added to every p >= 0 on output (tasks that are pieces of a caller's task)
	int32_t *d_f, *d_p;
	int32_t *d_t;               // stamp scratch for look-back beyond the LDS ring, one int per anchor
	int32_t *d_st;              // window start per anchor (chain.c:192-193), filled by the prepass kernel
	int32_t *d_status;          // per task, must be zero on entry
	int64_t max_task_anchors = 0;   // length of the longest task when the caller knows it (0: unknown): lets a small pass size a grid of one block per 256 anchors (chain_window_start_wide)
	int64_t longest_task = 0;       // the same for a batch of any size, with d_seg_ws: long tasks are given a prepass block per segment (chain_window_start_t<true>) -- 0: a block per task
	unsigned long long *d_seg_ws = nullptr;   // 4 words of 64 bits per task, zero before the first run (the prepass leaves them zero)
	int coop_waves = 0;         // > 1: a pass of few tasks -- each task gets a workgroup of several waves that share its LDS rings (chain_dp_coop.h; the variants of the hand-written loop, no device-side cut)
	int dry_run = 0;            // 1: launch_chain_dp fills `info` (which kernel a pass would take, whether it makes its window starts, whether it writes the caller's buffer) and launches nothing
	const int64_t *hm_off = nullptr; const float *hm_avg = nullptr; const int32_t *hm_pbase = nullptr;   // ... and the HOST's view of that metadata (plain host pointers): a pass of few pieces hands it over in the kernel's arguments
	const void *h_anchors = nullptr;   // a single-launch per-read pass: d_anchors has not been uploaded -- the cooperative kernel copies every task from here (same layout, pinned and mapped) itself;
	                                   // d_offsets / d_order / d_avg / d_pbase then point into the pinned arena as well (the kernel reads them once per workgroup)
	int fuse_st = 1;            // a pass of few short tasks (<= COOP_ST_MAX anchors each, max_task_anchors known) for the sixteen-wave kernel: it makes the window starts itself, no prepass launch
	int coop_w8_above = 256;    // the cooperative kernel takes eight waves per piece (two workgroups per CU) beyond this many pieces, sixteen up to it (chain_kernel.hip: launch_coop)
	                            // < 0 (with a device-side cut): decided on the device once the pieces are known (chain_route): few long pieces -> the cooperative kernel
	// a small per-read pass that ends in the cooperative kernel (round 6): f / p also go straight to the caller's page-locked buffer and the last workgroup raises the
	// flag the caller polls (what stage_out did in a launch of its own, host_stage.hip); taken only when that kernel is the pass's one DP launch (LaunchInfo::host_out)
	int32_t *h_f = nullptr, *h_p = nullptr; unsigned *d_done = nullptr, *h_flag = nullptr; unsigned seq = 0;
	int st_ready = 0;           // 1: d_st already holds the window starts (a per-read pass computes them on the host while it stages the anchors): no prepass for the cooperative kernel
	int force_tab = 0;          // 1: the gap-cost table of the tile kernel also for gap_scale == 1 (mm2c_tune("force_tab"); slower, kept for the parity tests)
	int ring_class;             // 3: tile kernel (general variant: first-generation kernel); 4: tile kernel for every variant; 0 / 1 / 2: first-generation kernel with 256 / 512 / 1024 anchors of LDS ring
	CutArgs cut;                // plans: cut the tasks into independent pieces on the device first
};

// which instantiation launch_chain_dp chose for the tasks' first pass (mm2c_plan_last_variant; the parity tests assert it, so that a vector set
// is known to have gone through the hand-written loop and not through the C++ restatement beside it)
struct LaunchInfo {
	int tile;        // 1: chain_dp_tile (second generation), 0: chain_dp_wave
	int nx, nf, r;   // tile kernel: tiles in the x / q and f / p rings of class 0; wave kernel: anchors in the LDS ring
	int skip, gen, gs1, far_, tab;
	int asm_loop;    // the hand-written per-tile loop (scan_tile_asm_*) runs, not scan_anchor<>
	int classes;     // ring-size classes: class-1 tasks run the instantiation with 2 * nx tiles
	int c16;         // tasks whose q values allow it run the instantiations with the compact x / q ring (per task: bit 1 of its class clear)
	int q24;         // class-1 tasks (long ring) run the instantiation with the q24 ring (chain_dp_tile.h, Lds<> RING 2) instead of 32-bit slots
	int cut;         // tasks are cut into pieces on the device first
	int coop;        // waves per task of the cooperative kernel (chain_dp_coop), 0: one wave per task
	int host_out;    // 1: the cooperative kernel wrote f / p to the caller's buffer and raised the flag itself (LaunchArgs::h_f ...)
	int single_ok;   // 1: a pass like this one can run as ONE launch (the cooperative kernel makes the window starts, writes the caller's buffer and can read the pinned arena itself)
	int fused_st;    // 1: the cooperative kernel made the window starts itself (no prepass launch)
	int route_auto;  // 1: which of the two ran is decided on the device after the cut (chain_route; CutArgs::d_count[1], [2] say which)
};
constexpr int COOP_ROUTE_MAX_PIECES = 2048;   // the cooperative kernel is only considered for batches of at most this many pieces
// few long pieces -> several waves per piece.  One wave per piece is bound by its longest piece (about 0.7 us per anchor); the cooperative kernel by the anchors a CU
// is dealt (about 0.1 us per anchor: total / 256 + the longest piece at worst): 7 Lmax > 1.05 (total / 256 + Lmax)  <=>  1450 Lmax > total (measured on long ava-ont
// reads, round 6: 2 048 x 100 000 anchors 80 ms / 94 ms, 1 020 x 300 000 210 / 127, 255 x 10^6 618 / 102 for one wave / sixteen).  With more pieces than `w8_above` the
// kernel runs eight waves per piece, two workgroups per CU (launch_coop): 0.38 ns per anchor of the batch against 0.47 -- for LONG pieces (a tile of 64 anchors costs the
// workgroup its barriers however few predecessors the anchors have) that moves the line to 2 048 equal pieces: 2 048 x 100 000 80.1 / 77.4, 2 048 x 10 000 11.9 / 10.9,
// 1 533 x 150 000 117.8 / 85.1, but 2 048 x 3 000 3.2 / 5.8 ms (gpurun_out/coop_w_sweep.txt, coop_w_small.txt -> profiles/r6_long_reads.md).
__host__ __device__ inline bool coop_pays(long long n_pieces, long long longest, long long total, long long w8_above = 256)
{
	if (n_pieces <= 0 || n_pieces > COOP_ROUTE_MAX_PIECES) return false;
	return 1450ll * longest > total || (n_pieces > w8_above && longest >= 8192 && 2300ll * longest > total);
}


int chain_ring_anchors(int ring_class);
// label counters of the hand-written loop (builds with -DMM2C_LABEL_COUNT only; hipErrorNotSupported otherwise): 8 rows (compact << 2 | table << 1 | far) x 32 labels
hipError_t label_hits_read(unsigned long long *out /* 256 */, bool reset);
// chain.c:53-78 for a CSR batch: per-anchor sub-part counts, per-task totals (any output may be nullptr)
hipError_t launch_predict(int32_t max_dist_x, int64_t n_tasks, const int64_t *d_offsets, const int32_t *d_order, const void *d_anchors,
                          uint8_t *d_num_subparts, int64_t *d_total_subparts, int64_t *d_total_trip, hipStream_t st);
// ev_dp_begin (optional) is recorded between the window-start prepass and the DP kernel
hipError_t launch_chain_dp(const LaunchArgs &L, hipStream_t st, int *n_launches, hipEvent_t ev_dp_begin, LaunchInfo *info = nullptr);

} // namespace mm2c

// argument blocks of the kernels around the DP (device epilogue, seed hits); kept in a file of their own because this header and the DP kernel
// sources are what profiles/traffic.json is tied to by hash
#include "chain_aux.h"
#endif
