/*
 * mm2chain.h -- C ABI of the MI355X (gfx950) chaining library, libmm2chain_hip.so.
 *
 * Drop-in boundary for ONE hot path of kisarur/minimap2-fpga: the predecessor-scan DP that fills f[] / p[]
 * inside mm_chain_dp (chain.c:184-238), which the reference offloads through run_chaining_on_hw
 * (chain_hardware.h:68, chain_hardware.cpp:27-197, kernel device/minimap2_opencl.cl:24-182) -- and, for batched callers, the
 * steps either side of it inside mm_map_frag: collect_seed_hits before (map.c:215-247, "seed hits -> sorted anchors") and the rest
 * of mm_chain_dp after (chain.c:348-422, "f/p -> chains").
 *
 * Plain C: pointers and sizes only, no torch / C++ types.  All entry points return 0 on success and a
 * negative MM2C_E_* code on failure unless stated otherwise; mm2c_last_error() gives the message.  There is
 * NO CPU fallback anywhere behind this header: if no HIP device is usable every compute entry fails.
 *
 * Preconditions shared by every chaining entry (they are the reference caller's guarantees):
 *   - anchors of one task are sorted ascending by x (map.c:245); x = strand<<63 | rid<<32 | rpos (map.c:228-241) is compared as a
 *     64-bit value, no layout inside it is assumed
 *   - 0 <= max_dist_x, and every task has n < 2^31 - 64 anchors (chain.c:30 "TODO" holds here too)
 */
#ifndef MM2CHAIN_H
#define MM2CHAIN_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same layout as mm128_t (minimap.h:53): 16 bytes, x then y */
typedef struct { uint64_t x, y; } mm2c_anchor_t;

enum {
	MM2C_OK = 0,
	MM2C_E_NODEVICE = -1,  /* no usable HIP device / mm2c_init not called (cf. hardware_init false, main.c:367) */
	MM2C_E_ARG = -2,       /* bad argument (negative n, NULL pointer, negative max_dist_x, ...) */
	MM2C_E_TOOBIG = -3,    /* task larger than the supported maximum (cf. chain_hardware.cpp:34-37): see MM2C_MAX_TASK_ANCHORS */
	MM2C_E_HIP = -4        /* HIP runtime error (cf. checkError, chain_hardware.cpp:208-235) */
};

/* flags for mm2c_params_t.flags */
#define MM2C_F_IGNORE_SEG  0x1  /* treat all anchors as one segment (what the FPGA kernel does, .cl:116-127) */
#define MM2C_F_FORCE_GENERAL 0x2 /* always run the general (segment / cDNA aware) kernel variant */

/*
 * Scalars of one mm_chain_dp call (mmpriv.h:65 arguments 1-5, 8-10).  q_span_override >= 0 replaces the
 * per-anchor span (a[i].y>>32 & 0xff, chain.c:189) by one value, which is what the reference's device
 * interface carries (chain_hardware.h:68 argument q_span, chain.c:91-96).
 */
typedef struct {
	int32_t max_dist_x, max_dist_y, bw;
	int32_t max_skip, max_iter;
	float   gap_scale;
	int32_t is_cdna, n_segs;
	int32_t q_span_override;   /* -1: per-anchor span */
	int32_t flags;             /* MM2C_F_* */
} mm2c_params_t;

/* ---- lifecycle: replaces hardware_init(BUFFER_N, XCLBIN_FILE) / cleanup() (chain_hardware.h:70-71, main.c:367,430) ---- */
int  mm2c_init(int device_ordinal);           /* -1: current device -- or, when the environment variable MM2C_DEVICES is set ("all" or a
                                               * comma-separated list of ordinals), those devices as with mm2c_init_devices: the route for a host
                                               * whose init hook carries no ordinals (hardware_init, chain_hardware.h:69).  Idempotent. */
/* mm2c_init on a thread of its own: returns at once, the runtime start-up (about 0.2 s) and the loading of the kernels' code objects run while the host
 * does its own start-up work -- a minimap2 host reads or builds its index between hardware_init (main.c:367) and the first chaining call (main.c:371-406).
 * Every entry that needs the device waits for that thread first; a failure shows there (and in mm2c_init_wait) as MM2C_E_NODEVICE with the reason, the way
 * a runtime error of the reference surfaces at the call (chain_hardware.cpp:208-235).  mm2c_init_wait: 0 once the library is ready.  mm2c_warm_up: loads
 * the code objects onto every configured device now instead of at each translation unit's first launch (what the asynchronous form does after its init). */
int  mm2c_init_async(int device_ordinal);
int  mm2c_init_wait(void);
int  mm2c_warm_up(void);
/* Several devices in one process (the reference scaffolds NUM_HW_KERNELS command queues / buffer sets / locks, chain_hardware.cpp:9-23,
 * chain_hardware.h:57): mm2c_init_devices instead of mm2c_init.  ordinals[0] is the primary device (plans run there); the per-read entries (run_chaining_on_hw, mm_chain_dp,
 * mm2c_chain_task_host) go to the device slot with the least work inside it, every slot with a call combiner of its own (mm2c_get_slot_stats); the host-batch entries (mm2c_chain_batch_host, mm2c_mm_chain_dp_batch_host, mm2c_seed_chain_batch_host)
 * split a batch of at least "multi_min_anchors" anchors (mm2c_tune, default 2^20) into one contiguous range of tasks per device with
 * about equal anchor counts (mm2c_split_tasks) and run the ranges side by side, each on its own stream set and arenas.  Chaining tasks
 * are independent (chain.c:42-45), so there is no exchange between devices.  An ordinal may be listed more than once. */
int  mm2c_init_devices(int n, const int *ordinals);
int  mm2c_device_count(void);
/* Host feed of several devices from one process: the worker thread of every device slot is persistent and pinned to the CPUs of the NUMA node its device hangs
 * off ("pin_workers", default 1), so that the page-locked staging buffers it allocates and the copies it drives stay on that socket.  mm2c_numa_cpulist reads
 * the mapping the way the workers do -- <sysfs_root>/bus/pci/devices/<pci bus id>/numa_node, then <sysfs_root>/devices/system/node/node<N>/cpulist -- and returns
 * the node (cpulist text in buf) or -1 when there is no NUMA information; it touches no device (sysfs_root "/sys"; tests hand it a made-up tree).
 * mm2c_slot_worker_node: the node slot's worker was pinned to, -1 = not started or not pinned. */
int  mm2c_numa_cpulist(const char *sysfs_root, const char *pci_bus_id, char *buf, size_t len);
int  mm2c_slot_worker_node(int slot);
int  mm2c_split_tasks(int64_t n_tasks, const int64_t *offsets, int n_parts, int64_t *bounds /* n_parts + 1 */);
void mm2c_shutdown(void);                     /* not while another thread is inside a compute entry */
const char *mm2c_last_error(void);            /* thread-local message of the last failing call */
int  mm2c_device_info(char *name, size_t name_len, int *cu_count, size_t *hbm_bytes);
/* which physical device the calling thread's library device is: its HIP ordinal, PCI bus id ("0000:c1:00.0") and architecture name.  A multi-GPU
 * launcher prints one per rank and refuses to report a scaling figure when two ranks sit on the same card (bench.py); the reference's counterpart is
 * the device list of hardware_init (chain_hardware.cpp:278-330: platform / device enumeration before the queues are made). */
int  mm2c_device_identity(int *ordinal, char *pci_bus_id, size_t bus_len, char *arch, size_t arch_len);
/* tuning knobs (key, value): "ring_class" 3 = the tile-aligned DP kernel (default; env MM2C_RING_CLASS; the segment / cDNA variant runs in the
 * first-generation kernel, which is faster for it), 4 = the tile kernel for every variant, 0/1/2 = the first-generation kernel with
 * 256/512/1024 anchors of LDS ring per task; "far_ring" 1 = plans give tasks with the 32-bit LDS ring (see "compact_ring") whose scans are expected to go far beyond the 448-anchor
 * LDS ring of the tile kernel an instantiation with a ring twice as long (chosen per task by the prepass; default; env MM2C_FAR_RING), 0 = never,
 * 2 = every task; "epi_fused" 1 = the device epilogue keeps the per-anchor state of tasks of up to
 * 7 680 anchors in LDS (default; env MM2C_EPI_FUSED), 0 = in HBM for every task; "compact_ring" 1 = tasks whose query positions span at most 65535 - min(max_dist_x, max_dist_y) (reads of up to about 55-60 kb) run
 * an instantiation of the tile kernel whose LDS ring keeps the low 16 bits of x and q only (16 tiles of look-back in the LDS of 8; default; env MM2C_COMPACT_RING),
 * 0 = never; "split_streams" 1 = a plan whose tasks differ in size runs the instantiations its batch is split over side by side on two streams (default;
 * env MM2C_SPLIT_STREAMS), 2 = every plan, 0 = never; "wide_share_threshold" = when the tasks that need the 32-bit ring hold more than this percentage of the
 * batch's anchors every task takes it (default 40); "force_tab" 1 = the tile kernel reads the gap
 * cost from its LDS table also when gap_scale is 1 (default 0: computed; tests); "seg_min" = shortest piece (anchors) a task is cut into at
 * empty-window positions (default 256, 0 = never cut); "plan_cut" 0/1 = plans cut their tasks of at least "plan_cut_min" anchors (default
 * 8192) into such pieces on the device before the DP (default 1); "pipeline_chunk_anchors" = chunk size of the two-stream pipeline used for
 * host batches of at least twice that size (default 20 Mi anchors); "multi_min_anchors" = smallest host batch that is split across the
 * devices of mm2c_init_devices (default 2^20); "trim" = give the cached device memory back to the runtime.  Results never depend on a knob.  "coop_waves" > 1 (default 16, the only width built: any value above 1 means 16) = a host-buffer pass of at most "coop_max_tasks" (1024) pieces gives every piece a workgroup of
 * 16 waves that share its LDS rings (csrc/chain_dp_coop.h: candidates counted and the older tiles reduced in parallel, the exact scan only where the early exit of chain.c:231 can fire;
 * env MM2C_COOP_WAVES), 0 = one wave per piece always; "coop_plans" 1 = plans of few tasks take that kernel too (tests); "combiner_lanes" 1..16 = passes the call combiner of the per-read
 * entries may have in flight at once on each device (default 4; env MM2C_COMBINER_LANES), "combine_max_anchors" = a call of more anchors than this runs alone (env MM2C_COMBINE_MAX). */
int  mm2c_tune(const char *key, int value);

/* HW/SW split model of the reference for this hardware (chain.c:80-81,101; constants in the form of chain_hardware.h:19-30 live in
 * include/mm2chain_split.h): hw_ms = k1_hw * n + k2_hw * total_subparts + c_hw for one synchronous per-read call into this library,
 * sw_ms = k_sw * total_trip_count + c_sw for chain.c's loop on one host core.  preset: "map-ont" or "asm20".  The library itself never
 * declines a task; a host that keeps the reference's predictor loads K1_HW .. C_SW (options.c:6,95-99) from here. */
int  mm2c_split_model(const char *preset, float *k1_hw, float *k2_hw, float *c_hw, float *k_sw, float *c_sw);

/* defaults of `minimap2 -x map-ont` (options.c:24-31,93-99; map.c:305-316) */
void mm2c_params_map_ont(mm2c_params_t *p);
/* V2 = what the FPGA kernel computes for a run_chaining_on_hw call */
void mm2c_params_fpga_v2(mm2c_params_t *p, int32_t max_dist_x, int32_t max_dist_y, int32_t bw, int32_t q_span);

/* ---- batched, HBM-resident path ------------------------------------------------------------------------------ */
/*
 * A plan describes one batch of independent chaining tasks in CSR form: task k owns anchors
 * [offsets[k], offsets[k+1]) of the concatenated arrays.  Creating a plan uploads offsets and the
 * longest-first launch order and reserves the device workspace; it can be run many times.
 */
typedef struct mm2c_plan mm2c_plan_t;
mm2c_plan_t *mm2c_plan_create(const mm2c_params_t *par, int64_t n_tasks, const int64_t *h_offsets);
void mm2c_plan_destroy(mm2c_plan_t *plan);
int64_t mm2c_plan_total_anchors(const mm2c_plan_t *plan);

/* stream arguments: a hipStream_t passed as void*.  NULL is the HIP null stream (what a default-stream caller such as PyTorch works on,
 * so the call is ordered with the caller's own work); MM2C_STREAM_LIBRARY asks for the library's private non-blocking stream. */
#define MM2C_STREAM_LIBRARY ((void *)(intptr_t)-1)

/*
 * Enqueue the DP for every task of the plan on `stream`.  All pointers are DEVICE pointers: d_anchors[total] (16 B each), d_f[total], d_p[total].
 * d_avg_qspan is either NULL (the kernel computes avg_qspan_scaled per task exactly as chain.c:48-49) or
 * n_tasks floats.  p[] is task-relative, -1 = no predecessor, exactly as chain.c:236.  Asynchronous.
 * A plan owns one workspace: do not run the same plan concurrently with itself (different plans and different streams are fine).
 */
int mm2c_plan_run_device(mm2c_plan_t *plan, const void *d_anchors, const float *d_avg_qspan,
                         int32_t *d_f, int32_t *d_p, void *stream);

/* Task sizes that only the device knows (anchors made by mm2c_seedplan_run_device_skip): from now on the plan's kernels take the CSR
 * offsets from d_offsets (n_tasks + 1 entries, device memory, offsets[0] = 0, every task no longer than the plan was created for);
 * NULL goes back to the plan's own.  The buffers handed to the run / chains entries keep the plan's (capacity) extents. */
int mm2c_plan_set_device_offsets(mm2c_plan_t *plan, const int64_t *d_offsets);

/* The same with the extent of every buffer stated (elements, not bytes): MM2C_E_TOOBIG when one is shorter than the plan needs, the way
 * the reference refuses n > BUFFER_N (chain_hardware.cpp:34-37).  The entries without _n trust the caller. */
int mm2c_plan_run_device_n(mm2c_plan_t *plan, const void *d_anchors, int64_t n_anchors, const float *d_avg_qspan, int64_t n_avg,
                           int32_t *d_f, int64_t n_f, int32_t *d_p, int64_t n_p, void *stream);

/* milliseconds between HIP events recorded (on the run's stream) around the DP kernel (chain_dp_wave) of the most
 * recent mm2c_plan_run_device; synchronises on the end event.  _prepass_ms: the same for the window-start kernel
 * (chain_window_start) that runs just before it. */
int mm2c_plan_last_kernel_ms(mm2c_plan_t *plan, float *ms);
/* Round 6: which of the two DP kernels took the pieces of the most recent run.  A batch of few long pieces -- fewer pieces than the GPU has wave slots, BASELINE config 5's
 * long reads -- runs with sixteen (or eight) waves per piece (chain_dp_coop, the analogue of the reference's one deep pipeline per task, device/minimap2_opencl.cl:49,71), anything
 * else with one wave per piece; with long tasks cut into pieces on the device the choice is made there (chain_route), so this call waits for the run and reads it back.
 * pieces = the tasks, or the pieces they were cut into; one_wave_pieces + coop_pieces = pieces.  mm2c_tune("coop_plans", 0 | 1 | 2): never / every small plan / per run. */
int mm2c_plan_last_route(mm2c_plan_t *plan, int64_t *pieces, int64_t *one_wave_pieces, int64_t *coop_pieces);
/* The rule itself (pure arithmetic, no device): waves per piece -- 16, 8 or 1 -- that a batch of `pieces` pieces, the longest of `longest` anchors, `total` anchors in all, is
 * given under "coop_plans" 2.  One wave per piece is bound by its longest piece (about 0.7 us per anchor), several waves per piece by the anchors a CU is dealt (about
 * 0.1 us per anchor: total / 256 + the longest piece at worst): several when pieces <= 2048 and 1450 * longest > total; of these sixteen (one workgroup per CU) up to
 * mm2c_tune("coop_w8_above", 256) pieces, eight (two workgroups per CU, each filling the other's waits) beyond -- where pieces of at least 8 192 anchors also take
 * them when 2300 * longest > total (profiles/r6_long_reads.md). */
int mm2c_route_pieces(int64_t pieces, int64_t longest, int64_t total);
/* Which kernel instantiation the most recent mm2c_plan_run_device launched for the tasks' first pass, as text, e.g.
 * "chain_dp_tile<NX=8,NF=2,SKIP=1,GEN=0,GS1=1,FAR=1,TAB=0> loop=asm classes=1 cut=0" (loop=asm: the hand-written per-tile loop, loop=c++: its
 * C++ restatement; chain_dp_wave<...>: the first-generation kernel).  For tests and logs: results never depend on the instantiation. */
int mm2c_plan_last_variant(mm2c_plan_t *plan, char *buf, size_t len);
/* the same text for the last DP launch of a host-buffer entry (mm2c_chain_task_host, mm2c_chain_batch_host, mm2c_mm_chain_dp_batch_host), process-wide:
 * those entries give their passes the prepass classes too, so tasks whose q values allow it take the compact ring there as in plans */
int  mm2c_last_host_variant(char *buf, size_t len);
/* test instrumentation: how often the hand-written anchor loop of the DP kernel (csrc/chain_dp_tile.h, MM2C_SCAN_TILE_ASM) passed each of its labels, 8 rows
 * (compact ring << 2 | gap-cost table << 1 | far instantiation) x 32 label bits (MM2C_LB_*), since the last reset.  Only the build with -DMM2C_LABEL_COUNT
 * (minimap2-fpga_amd/variants/labelcount.so, loaded through MM2C_LIB_PATH by tests/test_gpu_labels.py) counts; the shipped library returns MM2C_E_ARG. */
int  mm2c_debug_label_hits(unsigned long long *hits /* 256 */, int reset);
int mm2c_plan_last_prepass_ms(mm2c_plan_t *plan, float *ms);

/*
 * The reference's HW/SW prediction pass (chain.c:53-78) for every task of the plan, on the GPU: num_subparts[total]
 * (uint8, what chain.c:76 stores and run_chaining_on_hw receives), and per task total_subparts (chain.c:77) and
 * total_trip_count (chain.c:69), the inputs of the two linear time models of chain.c:80-81.  Any output may be NULL.
 */
int mm2c_plan_predict_device(mm2c_plan_t *plan, const void *d_anchors, uint8_t *d_num_subparts,
                             int64_t *d_total_subparts, int64_t *d_total_trip_count, void *stream);

/* ---- host-buffer paths (PCIe included) ---------------------------------------------------------------------- */
/* whole batch from pageable host memory: staging, H2D, DP, D2H, sync */
int mm2c_chain_batch_host(const mm2c_params_t *par, int64_t n_tasks, const int64_t *h_offsets,
                          const mm2c_anchor_t *h_anchors, const float *h_avg_qspan /* or NULL */,
                          int32_t *h_f, int32_t *h_p);

/* page-locked host memory for callers that want full PCIe rate on the host-buffer paths (any pointer works there; pageable
 * memory is staged by the runtime at roughly a quarter of the rate).  NULL on failure. */
void *mm2c_pinned_alloc(size_t bytes);
void  mm2c_pinned_free(void *ptr);

/*
 * THE LARGEST TASK (round 6).  The reference sizes its device buffers for BUFFER_N = 332 000 000 / 2 / 32 = 5 187 500 anchors per call (chain_hardware.h:62-64) and ends
 * the process on a longer one (chain_hardware.cpp:34-37).  Here buffers grow on demand and indices, stamps and p bases are 32 bits wide:
 *   - every entry that takes tasks (plans, batch entries, mm2c_chain_task_host[_pred], mm_chain_dp) accepts a task of up to MM2C_MAX_TASK_ANCHORS anchors and answers
 *     MM2C_E_TOOBIG for a longer one -- nothing wraps;
 *   - the device epilogue (mm2c_plan_chains_device, mm2c_mm_chain_dp_batch_host) takes batches (or pipeline chunks) of fewer than 2^31 anchors in all;
 *   - run_chaining_on_hw (the reference's symbol) additionally keeps the reference's own contract for the size its host states in hardware_init(buf_size, ...):
 *     n > buf_size -> "Error: The size of the call ..." on stderr and exit(1).
 * Tested at the reference's limit: one task of 5 187 500 anchors through run_chaining_on_hw, one of 2 000 000 through mm2c_chain_task_host, element-wise against the
 * oracle (tests/test_gpu_long_reads.py).
 */
#define MM2C_MAX_TASK_ANCHORS ((int64_t)2147483582)   /* 2^31 - 66 */

/*
 * One task, synchronous, V1 (stock CPU) semantics: the EXTENDED form of run_chaining_on_hw
 * (chain_hardware.h:68) that also carries max_skip / max_iter / gap_scale / is_cdna / n_segs, which the
 * reference interface cannot express.  Re-entrant from many host threads (map.c:561); `tid` as chain.c:103.
 * Always returns 0 (= "computed on device", chain_hardware.cpp:195) or a negative error; never 1.
 */
int mm2c_chain_task_host(const mm2c_params_t *par, int64_t n, const mm2c_anchor_t *a, float avg_qspan_scaled,
                         int32_t *f, int32_t *p, int tid);

/*
 * The same call with the reference's busy protocol (chain_hardware.cpp:54-75, PROCESS_ON_SW_IF_HW_BUSY, chain_hardware.h:50): the caller hands in the two
 * predictions chain.c:80-81 computes (milliseconds).  OFF BY DEFAULT since round 6 ("decline_when_busy" 0: every call is accepted) -- a chain.o built without
 * PROCESS_ON_SW_IF_HW_BUSY ignores the answer 1 (chain.c:105,163-169) and would chain from uninitialised f / p, and on the measured end-to-end run no decline rule
 * beat never declining (profiles/r6_per_read.md).  A host that HAS the software loop opts in (environment MM2C_DECLINE_WHEN_BUSY or mm2c_tune):
 *   1: declined when (calls inside the device slot / passes it runs at once + 1) x (what a pass of that slot has taken lately) > sw_time_pred -- the measured thing;
 *   2: round 5's rule: (predicted device time booked on the slot) / (passes at once) + hw_time_pred >= sw_time_pred;
 * the slots are tried from tid % n, the call RUNS on the slot that accepted it (chain_hardware.cpp:58-72), and when none will it returns 1 = "declined": nothing was
 * computed, f / p are untouched, the caller runs its own loop (chain.c:106,112-164).  This library's own mm_chain_dp (path B) never declines -- it has no software DP.
 * Predictions that are not both positive never decline.
 */
int mm2c_chain_task_host_pred(const mm2c_params_t *par, int64_t n, const mm2c_anchor_t *a, float avg_qspan_scaled,
                              int32_t *f, int32_t *p, int tid, float hw_time_pred, float sw_time_pred);

/* Per device slot (the order of mm2c_init_devices / MM2C_DEVICES; slot 0 = the primary device): what its call combiner has served since mm2c_init --
 * the reference keeps a queue, a lock and a buffer set per kernel (chain_hardware.cpp:9-23); here every device has its own, and a per-read call goes to
 * the slot with the least anchors outstanding.  declined: calls mm2c_chain_task_host_pred turned away (counted on slot 0). */
typedef struct { int32_t device; int32_t reserved; uint64_t passes, calls, anchors, declined; } mm2c_slot_stats_t;
int mm2c_get_slot_stats(int slot, mm2c_slot_stats_t *out);
/* the routing rule on its own (pure; no device is touched): given the anchors outstanding on each of n_slots device slots, the slot a call from worker `tid` goes to
 * -- the least loaded one, ties to the first such slot at or after tid % n_slots (so idle devices are taken in turn, chain_hardware.cpp:58-72 scans its kernels
 * from 0 instead).  A negative tid counts as 0. */
int mm2c_route_slot(int n_slots, const int64_t *outstanding, int tid);

/* ---- host mirror of the reference function around the path ------------------------------------------------------ */
/*
 * mm_chain_dp with the reference's exact signature and ownership rules (mmpriv.h:65, chain.c:29-423): frees `a`
 * with kfree(km, a); returns b[] and *_u allocated with kmalloc(km, ...).  The DP (f[], p[]) runs on the GPU
 * through mm2c_chain_task_host; v[] and the backtrack run on the calling thread.  Needs the host program's
 * kmalloc/kfree (kalloc.h:14,17) at link/load time.  Declared here with mm2c_anchor_t == mm128_t.
 */
#ifndef MM2C_NO_MM_CHAIN_DP_DECL   /* define it in a host that already includes mmpriv.h (same function, mm128_t spelling) */
mm2c_anchor_t *mm_chain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt,
                           int min_sc, float gap_scale, int is_cdna, int n_segs, int64_t n, mm2c_anchor_t *a,
                           int *n_u_, uint64_t **_u, void *km, int tid);
#endif

/* ---- whole mm_chain_dp for a batch: DP + epilogue (chain.c:106-111,348-422) ----
 * Outputs are compact: the chains of task k are u[u_off[k] .. u_off[k+1]) (score<<32 | count, in the order mm_chain_dp returns
 * them, chain.c:385-388,406-420) and their anchors b[b_off[k] .. b_off[k+1]); u_off / b_off have n_tasks+1 entries, u and b need
 * room for every anchor of the batch.  Results equal mm_chain_dp called task by task. */

/* the epilogue on the GPU, after mm2c_plan_run_device on the same stream; all pointers are device memory; needs
 * mm2c_plan_total_anchors < 2^31 */
int mm2c_plan_chains_device(mm2c_plan_t *plan, const void *d_anchors, const int32_t *d_f, const int32_t *d_p, int min_cnt, int min_sc,
                            int64_t *d_u_off, uint64_t *d_u, int64_t *d_b_off, void *d_b, void *stream);
int mm2c_plan_chains_device_n(mm2c_plan_t *plan, const void *d_anchors, int64_t n_anchors, const int32_t *d_f, int64_t n_f, const int32_t *d_p, int64_t n_p,
                              int min_cnt, int min_sc, int64_t *d_u_off, int64_t n_u_off, uint64_t *d_u, int64_t n_u, int64_t *d_b_off, int64_t n_b_off,
                              void *d_b, int64_t n_b, void *stream);
int mm2c_plan_last_epilogue_ms(mm2c_plan_t *plan, float *ms);

/* the epilogue on n_threads host threads from f[] / p[] in host memory (what the reference does on the calling thread) */
int mm2c_chain_epilogue_host(int min_cnt, int min_sc, int64_t n_tasks, const int64_t *h_offsets, const mm2c_anchor_t *h_anchors,
                             const int32_t *h_f, const int32_t *h_p, int n_threads, int64_t *u_off, uint64_t *u, int64_t *b_off,
                             mm2c_anchor_t *b);

/* host buffers in, chains out.  epilogue_threads == 0: DP and epilogue on the GPU, only chains come back over PCIe;
 * epilogue_threads > 0: DP on the GPU, f[] / p[] come back and the epilogue runs on that many host threads. */
int mm2c_mm_chain_dp_batch_host(const mm2c_params_t *par, int min_cnt, int min_sc, int64_t n_tasks, const int64_t *h_offsets,
                                const mm2c_anchor_t *h_anchors, int epilogue_threads, int64_t *u_off, uint64_t *u, int64_t *b_off,
                                mm2c_anchor_t *b);

/* ---- seed hits -> sorted anchors on the GPU (collect_seed_hits, map.c:215-247) ----
 * One match = one query minimizer found in the index (mm_match_t, map.c:76-81, as collect_matches map.c:84-120 fills it); its hits
 * are hits[cr_off .. cr_off + n) of a hit pool (the arrays mm_idx_get returns: rid<<32 | pos<<1 | strand).  The anchors of read r are
 * written to anchors[anchor_off[r] .. anchor_off[r+1]) exactly as collect_seed_hits leaves them for mm_chain_dp (encoding map.c:232-241,
 * order of radix_sort_128x including its order among equal x), ready for mm2c_plan_run_device with the same offsets.
 * mm2c_seedplan_run_device(_n) covers the flag-free case (no MM_F_NO_DIAG / NO_DUAL / FOR_ONLY / REV_ONLY), which is map-ont;
 * mm2c_seedplan_run_device_skip adds skip_seed (map.c:122-147) for ava-ont and the strand-restricted modes. */
typedef struct {
	int64_t cr_off;
	uint32_t n;          /* mm_match_t.n */
	uint32_t q_pos;      /* query position << 1 | strand (mm128_t.y low word of the minimizer) */
	uint32_t q_span;     /* mm128_t.x & 0xff */
	uint32_t seg_tandem; /* seg_id << 1 | is_tandem (map.c:112-115) */
} mm2c_match_t;
typedef struct mm2c_seedplan mm2c_seedplan_t;
/* h_anchor_off[r+1] - h_anchor_off[r] must equal the sum of n over the matches of read r (checked on the device: mm2c_seedplan_check) */
mm2c_seedplan_t *mm2c_seedplan_create(int64_t n_reads, const int64_t *h_match_off, const int64_t *h_anchor_off);
void mm2c_seedplan_destroy(mm2c_seedplan_t *plan);
/* asynchronous on `stream`; all pointers are device memory; d_anchors needs room for h_anchor_off[n_reads] anchors */
int mm2c_seedplan_run_device(mm2c_seedplan_t *plan, const mm2c_match_t *d_matches, const uint64_t *d_hits, const int32_t *d_qlen,
                             void *d_anchors, void *stream);
/* the same with the extents of the buffers stated; a match that points outside hits[0 .. n_hits) is detected on the device and reported by
 * mm2c_seedplan_check (its read is not expanded) */
int mm2c_seedplan_run_device_n(mm2c_seedplan_t *plan, const mm2c_match_t *d_matches, int64_t n_matches, const uint64_t *d_hits, int64_t n_hits,
                               const int32_t *d_qlen, int64_t n_qlen, void *d_anchors, int64_t n_anchors, void *stream);
/* With skip_seed (map.c:122-147): all-vs-all and strand-restricted modes drop hits, so a read keeps FEWER anchors than its matches have hits.
 * flag: MM_F_NO_DIAG 0x001 | MM_F_NO_DUAL 0x002 | MM_F_FOR_ONLY 0x100000 | MM_F_REV_ONLY 0x200000 (minimap.h:8-9,28-29; -x ava-ont sets the
 * first two, options.c:84).  The name comparison strcmp(qname, name[rid]) of map.c:128 travels as ranks: d_ref_rank[rid] = rank of the
 * reference sequence's name among the DISTINCT reference names in strcmp order, d_ref_len[rid] = its length (mm_idx_seq_t.len), and per read
 * d_q_lo = number of those names below the read's name, d_q_eq = 1 when the read's name is one of them (cmp > 0 <=> rank < q_lo,
 * cmp == 0 <=> q_eq && rank == q_lo); d_ref_rank == NULL stands for qname == NULL (map.c:125).  All arrays are device memory.
 * The plan's anchor offsets are then capacities; the anchors of read r are written packed to d_anchors[off[r] .. off[r+1]) with
 * off = d_anchor_off_out (n_reads + 1 entries, device) -- hand it to mm2c_plan_set_device_offsets to chain them.  MM_SEED_SELF (map.c:241) is set. */
typedef struct {
	int32_t flag;
	const int32_t *d_ref_rank, *d_ref_len;   /* per reference sequence */
	const int32_t *d_q_lo, *d_q_eq;          /* per read */
} mm2c_seed_skip_t;
int mm2c_seedplan_run_device_skip(mm2c_seedplan_t *plan, const mm2c_match_t *d_matches, int64_t n_matches, const uint64_t *d_hits, int64_t n_hits,
                                  const int32_t *d_qlen, int64_t n_qlen, const mm2c_seed_skip_t *skip, void *d_anchors, int64_t n_anchors,
                                  int64_t *d_anchor_off_out, void *stream);
/* MM_F_HEAP_SORT (minimap.h:30; --heap-sort, main.c:245; set by -x sr, options.c:125): the following runs of the plan leave the anchors as
 * collect_seed_hits_heap does (map.c:149-213) -- the same anchors, ascending in x, but anchors with EQUAL x in the order the reference's binary heap pops
 * them instead of the order radix_sort_128x leaves.  Works with and without skip_seed. */
int mm2c_seedplan_set_heap_sort(mm2c_seedplan_t *plan, int on);   /* mm2c_tune("heap_sort", 1): the default of the seed plans created afterwards, those of
                                                                    * mm2c_seed_hits_batch_host / mm2c_seed_chain_batch_host / _pool included (a host that maps with MM_F_HEAP_SORT) */
int mm2c_seedplan_check(mm2c_seedplan_t *plan, int64_t *n_reads_with_ties);   /* waits; MM2C_E_ARG if a read's counts disagreed */
int mm2c_seedplan_last_ms(mm2c_seedplan_t *plan, float *ms);
/* host buffers in, anchors out (computes the anchor offsets itself): anchor_off[n_reads+1], anchors with room for the sum of all n */
int mm2c_seed_hits_batch_host(int64_t n_reads, const int64_t *h_match_off, const mm2c_match_t *h_matches, const uint64_t *h_hits,
                              int64_t n_hits, const int32_t *h_qlen, int64_t *anchor_off, mm2c_anchor_t *anchors);

/* The index's position arrays resident in HBM (the arrays mm_idx_get, index.c, hands out pointers into: rid<<32 | pos<<1 | strand): uploaded
 * once per index, shared by every later call -- a 288 GB device holds the positions of a human genome index (about 10 GB) beside the batches.
 * With a resident pool a match's cr_off is an offset into THAT pool and no hit crosses PCIe per read any more. */
typedef struct mm2c_hitpool mm2c_hitpool_t;
mm2c_hitpool_t *mm2c_hitpool_create(const uint64_t *h_hits, int64_t n_hits);   /* NULL on failure */
int64_t mm2c_hitpool_size(const mm2c_hitpool_t *pool);
void mm2c_hitpool_destroy(mm2c_hitpool_t *pool);                                 /* not while a call that uses it is running */

/* matches in, chains out: collect_seed_hits + mm_chain_dp for a batch of reads (map.c:295-316) with the anchors staying on the GPU.
 * Big batches run in chunks of whole reads ("pipeline_chunk_anchors") on two streams: the upload of one chunk, the kernels of the
 * one before and the download of the chains of the one before that overlap; every buffer the chunks use is kept for the next call.
 * anchor_off[n_reads+1] is filled with the anchor counts' prefix sums (= the sum of n per read); u / b as mm2c_mm_chain_dp_batch_host
 * (room for one entry per anchor of the batch). */
int mm2c_seed_chain_batch_host(const mm2c_params_t *par, int min_cnt, int min_sc, int64_t n_reads, const int64_t *h_match_off,
                               const mm2c_match_t *h_matches, const uint64_t *h_hits, int64_t n_hits, const int32_t *h_qlen,
                               int64_t *anchor_off, int64_t *u_off, uint64_t *u, int64_t *b_off, mm2c_anchor_t *b);
/* the same with the hits taken from a resident pool: h_matches[i].cr_off points into `pool` */
int mm2c_seed_chain_batch_pool(const mm2c_params_t *par, int min_cnt, int min_sc, int64_t n_reads, const int64_t *h_match_off,
                               const mm2c_match_t *h_matches, const mm2c_hitpool_t *pool, const int32_t *h_qlen,
                               int64_t *anchor_off, int64_t *u_off, uint64_t *u, int64_t *b_off, mm2c_anchor_t *b);

/* ---- anchor streams on disk (csrc/anchor_stream.c documents the layout) ------------------------ */
typedef struct {
	mm2c_params_t par;            /* scalars of the mm_chain_dp calls the tasks came from */
	int32_t min_cnt, min_sc;      /* mm_chain_dp arguments 6-7, used by the backtrack only */
	int64_t n_tasks, total;
	int64_t *offsets;             /* n_tasks + 1, offsets[0] = 0; malloc'ed by the readers */
	mm2c_anchor_t *anchors;       /* total */
} mm2c_stream_t;
int  mm2c_stream_write(const char *path, const mm2c_params_t *par, int min_cnt, int min_sc, int64_t n_tasks,
                       const int64_t *offsets, const mm2c_anchor_t *anchors);
int  mm2c_stream_read(const char *path, mm2c_stream_t *out);
/* import what `minimap2 --print-seeds` prints before chaining (RS / SD lines, map.c:298-303) */
int  mm2c_stream_from_seed_dump(const char *text_path, const mm2c_params_t *par, int min_cnt, int min_sc, mm2c_stream_t *out);
void mm2c_stream_free(mm2c_stream_t *s);

/* statistics since mm2c_init: tasks, anchors, and kernel launches issued through any entry point */
typedef struct { uint64_t tasks, anchors, launches, segments, host_call_ns, passes; } mm2c_stats_t;   /* segments: pieces the host
   paths cut tasks into; host_call_ns: wall time inside the host-buffer entry points, summed over calling threads; passes: GPU
   passes they issued (concurrent small calls are combined into one pass) */
void mm2c_get_stats(mm2c_stats_t *out);

/* Where the time of the host-batch entries goes (mm2c_seed_chain_batch_host / _pool, mm2c_mm_chain_dp_batch_host, mm2c_chain_batch_host), summed
 * since mm2c_init or the last mm2c_reset_stage_stats.  Host stages are wall time of the calling thread(s); device stages are HIP-event time on
 * the stream of each chunk (chunks of one call overlap on two streams, so the device stages may add up to more than total_ns). */
typedef struct {
	uint64_t calls, chunks;      /* entries served, and the chunks their batches were pipelined in */
	uint64_t total_ns;           /* wall time inside the entries */
	uint64_t alloc_ns, n_alloc;  /* hipMalloc / hipHostMalloc: device-cache misses, arena growth, pinned staging (whole library) */
	uint64_t free_ns, n_free;    /* hipFree / hipHostFree and the device waits in front of them (whole library) */
	uint64_t setup_ns;           /* host: checking the offsets, launch orders, plan objects and their small synchronous uploads */
	uint64_t h2d_ns;             /* device: uploads (matches, hits or anchors, per-read metadata) */
	uint64_t seed_ns;            /* device: seed hits -> sorted anchors */
	uint64_t dp_ns;              /* device: window prepass + chaining DP */
	uint64_t epi_ns;             /* device: v[], backtrack, chain order */
	uint64_t d2h_ns;             /* device: downloads of offsets and chains (or f / p) */
	uint64_t wait_ns;            /* host: blocked on a stream or an event */
} mm2c_stage_stats_t;
void mm2c_get_stage_stats(mm2c_stage_stats_t *out);
void mm2c_reset_stage_stats(void);

#ifdef __cplusplus
}
#endif

/*
 * C++-linkage symbols with the reference's exact prototypes are defined in csrc/mm2chain_dropin.cpp
 * (chain_hardware.h:68-71; the reference compiles every .c with $(CXX), Makefile:184-185, so its objects import
 * _Z18run_chaining_on_hwliiiifP7mm128_tPiS1_Phliff, _Z13hardware_initlPc, _Z7cleanupv):
 *   int  run_chaining_on_hw(long n, int max_dist_x, int max_dist_y, int bw, int q_span, float avg_qspan, mm128_t *a,
 *                           int *f, int *p, unsigned char *num_subparts, long total_subparts, int tid,
 *                           float hw_time_pred, float sw_time_pred);   // computes V2 = what the FPGA kernel computes
 *   bool hardware_init(long, char *);
 *   void cleanup();
 */

#endif /* MM2CHAIN_H */
