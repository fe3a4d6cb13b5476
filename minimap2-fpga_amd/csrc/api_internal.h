// api_internal.h -- shared between the translation units of the C-ABI shim (mm2chain_api.cpp: init, plans; mm2chain_host.cpp: host-buffer
// paths; mm2chain_seeds.cpp: seed-hit entries).  Not installed; include/mm2chain.h is the public interface.
#ifndef MM2C_API_INTERNAL_H
#define MM2C_API_INTERNAL_H
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <mutex>
#include <numeric>
#include <vector>
#include "mm2chain.h"
#include "chain_kernel.h"

namespace mm2c_api {

extern thread_local char g_err[512];
int fail(int code, const char *fmt, ...);      // records the message mm2c_last_error() returns, hands back `code`

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	return ::mm2c_api::fail(MM2C_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

struct ThreadCtx;

struct Global {
	std::mutex mu;
	std::atomic<bool> ready{false};         // written under `mu`; read without it by lib_ready()
	int device = -1;                        // primary device (= devices[0])
	std::vector<int> devices;               // every device the library drives (mm2c_init_devices); entries may repeat
	std::atomic<int64_t> multi_min_anchors{1 << 20};   // host batches of at least this many anchors are split across the devices
	// tuning knobs: written by mm2c_tune / mm2c_init under `mu`, read by compute entries on other threads (atomics: no torn or stale-forever reads)
	std::atomic<int> ring_class{3};
	std::atomic<int> far_thr10{7};                      // ... from this many tenths of an expected far tile per anchor
	std::atomic<int> noskip_loop{1};                    // max_skip >= max_iter: through the hand-written loop with max_skip = max_iter - 1 (same results)
	std::atomic<int> heap_sort{0};                      // seed plans created from now on leave collect_seed_hits_heap's order among equal x (MM_F_HEAP_SORT)
	std::atomic<int> wide_pct{40};                      // plans: tasks with the 32-bit ring hold more than this % of the anchors -> no task takes the compact ring
	std::atomic<int> split_streams{1};                  // plans: the instantiations a batch is split over (32-bit / compact ring) run side by side on two streams
	std::atomic<int> compact_ring{1};                   // tile kernel: the compact x / q ring for the tasks whose q values allow it (0: never; the parity tests run both)
	std::atomic<int> q24_ring{1};                       // tile kernel: the long ring of class-1 tasks in the q24 form (0: 32-bit slots; the parity tests run both)
	std::atomic<int> force_tab{0};                      // tile kernel: gap cost from the LDS table also when gap_scale == 1 (tests; slower)
	std::atomic<int> far_ring{1};                       // plans: tasks whose scans are expected to leave the short LDS ring run with a ring twice as long (0: never, 2: all)
	std::atomic<int> epi_fused{1};                      // device epilogue: tasks that fit the LDS take the fused kernel (0: kernels A, B, C for every task)
	std::atomic<size_t> combine_max_anchors{1u << 17};   // host paths: calls up to this many anchors are combined with concurrent callers' calls
	std::atomic<size_t> stage_max_anchors{1u << 21};     // host paths: calls up to this many anchors go through pinned staging buffers
	std::atomic<int64_t> pipeline_chunk_anchors{20 << 20};  // host paths: batches of at least twice this size are pipelined in chunks of this size
	std::atomic<int64_t> pipeline_pieces{8}, pipeline_min_chunk{4 << 20};   // host paths: a batch of a few chunks' worth is cut into about `pieces` chunks of at least `min_chunk` anchors
	std::atomic<int64_t> pipeline_taper{0};             // host paths: the last chunks of a pipelined batch halve in size this many times (0, the default: equal chunks -- the sweep of
	                                                    // round 6, profiles/r6_hoststream.md, shows nothing to gain: 16.4-16.7 ms for 0 .. 4 halvings)
	std::atomic<int64_t> cut_below_tasks{4096};         // host paths: passes with at least this many tasks are not cut (they fill the GPU anyway)
	std::atomic<int> seg_min{256};                      // shortest piece a task is cut into at empty-window positions (0 = never cut)
	std::atomic<int> coop_waves{16};                    // passes of at most coop_max_tasks tasks: several waves per task (chain_dp_coop; 0 or 1: never; the value only switches, the width -- 16 or 8 -- goes by coop_w8_above)
	std::atomic<int64_t> coop_max_tasks{1024};
	std::atomic<int> single_launch{1};                  // per-read passes of short tasks: ONE launch -- the cooperative kernel reads the pass from the pinned arena itself (no stage_in)
	std::atomic<int> fuse_st{1};                        // per-read passes: the window starts inside the cooperative kernel (no prepass launch)
	std::atomic<int> seg_prepass{1};                    // plans with long tasks: the window-start prepass with a block per segment of a task instead of a block per task
	std::atomic<int> coop_w8_above{256};                // the cooperative kernel: eight waves per piece (two workgroups per CU) in passes of more pieces than this, sixteen up to it
	std::atomic<int> pin_workers{1};                    // the worker thread of a device slot is pinned to the CPUs of the device's NUMA node (sysfs; 0: left to the scheduler)
	std::atomic<int> decline_when_busy{0};              // mm2c_chain_task_host_pred / run_chaining_on_hw: the reference's busy protocol (chain_hardware.cpp:54-75).  0 (default since round 6):
	                                                    // always accept -- a chain.o built without PROCESS_ON_SW_IF_HW_BUSY ignores the answer 1 (chain.c:105,163-169) and would chain from
	                                                    // uninitialised f / p, and no decline rule beat "never" on the measured run (profiles/r6_per_read.md); 1: decline by the measured
	                                                    // service time of the slot; 2: round 5's rule (booked predictions).  MM2C_DECLINE_WHEN_BUSY / mm2c_tune opt in.
	std::atomic<int> direct_pass{1};                    // small staged passes: the two copies are kernels and the host polls a flag word (host_stage.hip; 0: copy commands + stream wait)
	std::atomic<size_t> direct_max_anchors{1u << 18};   // ... passes of up to this many anchors
	std::atomic<int> host_st{0};                        // ... bring their window starts along, computed by the host while it stages the anchors (no prepass launch).  Off: measured on the
	                                                    // end-to-end run it trades 4.3 us of submission for 8 us of host sweep per pass (profiles/r6_per_read.md); mm2c_tune("host_st", 1) turns it on
	std::atomic<int> fused_out{1};                      // ... that end in the cooperative kernel: the kernel writes f / p to the result buffer and raises the flag itself (no stage_out launch)
	std::atomic<int> pipe_coop_chunks{1};               // pipelined host batches: this many of the LAST chunks run with several waves per piece when they have few enough pieces
	std::atomic<int> combiner_lanes{4};                 // passes of the call combiner in flight at once (1 .. 16; round 5, with direct passes: 4.60 / 4.55 / 4.65 / 4.67 / 4.78 / 4.83 s for 3 / 4 / 6 / 8 / 12 / 16 on the 120 000-read run)
	std::atomic<int> coop_plans{2};                     // plans and the cooperative kernel: 2 (default) per run, by coop_pays (few long pieces); 1: every plan of few tasks (tests); 0: never
	std::atomic<int> plan_cut{1};                       // plans: cut long tasks into pieces on the device (chain_cut) before the DP
	std::atomic<int> plan_cut_min{8192};                // ... tasks of at least this many anchors (the ones that make the tail of a batch)
	hipStream_t stream = nullptr;           // library stream for plan runs with stream == NULL
	std::vector<ThreadCtx *> thread_ctxs;   // owned; released in mm2c_shutdown
	std::atomic<uint64_t> tasks{0}, anchors{0}, launches{0}, segments{0}, host_call_ns{0}, passes{0};
	uint64_t epoch = 0;                     // bumped by shutdown so stale thread-local pointers are dropped
};
extern Global G;
// mm2c_init_async: the initialisation runs on a thread of its own while the host does something else (a minimap2 host loads its index, main.c:371-399, before the
// first chaining call); every entry that needs the device joins that thread first.  async_init_join is a no-op when none is pending and on the thread itself.
void async_init_join();
inline bool lib_ready() { async_init_join(); return G.ready; }
int fail_not_ready();                          // MM2C_E_NODEVICE with the reason: never initialised, or the asynchronous initialisation's own error

// mm2c_stage_stats_t, as atomics (the entries run on many host threads)
struct StageStats {
	std::atomic<uint64_t> calls{0}, chunks{0}, total_ns{0}, alloc_ns{0}, n_alloc{0}, free_ns{0}, n_free{0}, setup_ns{0}, h2d_ns{0}, seed_ns{0}, dp_ns{0},
	                      epi_ns{0}, d2h_ns{0}, wait_ns{0};
};
extern StageStats SS;
struct ScopedNs {                        // adds the wall time of its scope to a counter
	std::atomic<uint64_t> &acc; std::chrono::steady_clock::time_point t0;
	explicit ScopedNs(std::atomic<uint64_t> &a) : acc(a), t0(std::chrono::steady_clock::now()) {}
	~ScopedNs() { acc += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

// one chunk in flight of mm2c_mm_chain_dp_batch_host (anchors up, DP, epilogue, chains down); grow-only arenas
struct WholeSlot {
	hipStream_t st = nullptr;
	char *d_in = nullptr, *d_work = nullptr, *d_res = nullptr, *h_meta = nullptr;
	size_t cap_in = 0, cap_work = 0, cap_res = 0, cap_hmeta = 0;
	int64_t k0 = 0, k1 = 0;                    // tasks of the chunk in flight
	size_t o_res_u = 0, o_res_b = 0, o_hres = 0;
	bool busy = false;
	void release()
	{
		if (d_in) (void)hipFree(d_in); if (d_work) (void)hipFree(d_work); if (d_res) (void)hipFree(d_res);
		if (h_meta) (void)hipHostFree(h_meta); if (st) (void)hipStreamDestroy(st);
		*this = WholeSlot();
	}
};

// one chunk in flight of mm2c_seed_chain_batch_host / _pool (matches up, seed hits -> anchors, DP, epilogue, chains down)
struct SeedSlot {
	hipStream_t st = nullptr;
	hipEvent_t ev[6] = {};                    // begin, uploaded, anchors made, DP done, chains made, offsets downloaded
	char *d_buf = nullptr, *h_meta = nullptr; // grow-only arena [matches | hits | qlen | anchors | f | p | u_off | b_off | u | b]; pinned [u_off | b_off]
	size_t cap_buf = 0, cap_hmeta = 0;
	void *seedplan = nullptr, *plan = nullptr;   // the chunk's plans (their workspace comes from the device cache)
	int64_t k0 = 0, k1 = 0;
	size_t o_uo = 0, o_bo = 0, o_u = 0, o_b = 0;
	bool busy = false, timed = false;
	void release()
	{
		if (d_buf) (void)hipFree(d_buf); if (h_meta) (void)hipHostFree(h_meta);
		for (auto &e : ev) if (e) (void)hipEventDestroy(e);
		if (st) (void)hipStreamDestroy(st);
		*this = SeedSlot();
	}
};

// per host thread: stream + grow-only buffers (the reference keeps one buffer set per FPGA kernel,
// chain_hardware.cpp:13-16,379-397, and serialises callers on a mutex).  One device arena for everything that is uploaded
// ([anchors | piece offsets | launch order | p base | avg | status]) and one for everything that is downloaded ([f | p]), each
// mirrored by a pinned host staging buffer, so that a call is one H2D copy, the kernels, one D2H copy and one sync.
struct ThreadCtx {
	int device = -1;                           // >= 0: the context belongs to this device whoever runs on it (a lane of a device slot's call combiner)
	WholeSlot whole[2];
	SeedSlot seed[2];
	hipStream_t st = nullptr, st2 = nullptr;   // st2: second compute stream of the pipelined big-batch path
	hipStream_t st3 = nullptr, st_up = nullptr;   // ... its third compute stream and its upload stream
	std::vector<hipEvent_t> evs;               // ... and its events (one per chunk in flight), kept for the next call
	hipEvent_t ev = nullptr;
	mm2c::LaunchInfo last_info = {};           // which instantiation the context's last DP launch chose (copied to the process-wide record, mm2c_last_host_variant)
	char *d_in = nullptr, *d_out = nullptr, *d_scratch = nullptr;   // device
	char *h_in = nullptr, *h_out = nullptr;                          // pinned host
	unsigned *h_flag = nullptr;                                      // pinned host: the word stage_out raises when a direct pass is done (host_stage.hip)
	unsigned seq = 0;                                                // number of the context's last direct pass (the value the flag takes)
	unsigned *d_cnt = nullptr;                                       // device: the counter of finished workgroups of a direct pass (zero between passes: whoever counts last puts it back)
	size_t cap_in = 0, cap_out = 0, cap_scratch = 0, cap_hin = 0, cap_hout = 0;
	void release()
	{
		if (d_in) (void)hipFree(d_in); if (d_out) (void)hipFree(d_out); if (d_scratch) (void)hipFree(d_scratch);
		if (h_in) (void)hipHostFree(h_in); if (h_out) (void)hipHostFree(h_out); if (h_flag) (void)hipHostFree(h_flag); if (d_cnt) (void)hipFree(d_cnt);
		if (st) (void)hipStreamDestroy(st); if (st2) (void)hipStreamDestroy(st2); if (ev) (void)hipEventDestroy(ev);
		if (st3) (void)hipStreamDestroy(st3); if (st_up) (void)hipStreamDestroy(st_up);
		for (hipEvent_t e : evs) (void)hipEventDestroy(e);
		whole[0].release(); whole[1].release();
		seed[0].release(); seed[1].release();
		*this = ThreadCtx();
	}
};

// "chain_dp_tile<...> loop=asm ... compact=1": the text of mm2c_plan_last_variant / mm2c_last_host_variant
inline void format_variant(const mm2c::LaunchInfo &I, char *buf, size_t len)
{
	if (I.tile && I.coop) snprintf(buf, len, "chain_dp_coop<W=%d,NX=%d,NF=%d,GS1=%d,FAR=%d,TAB=%d> loop=%s classes=0 cut=0 compact=0 coop=%d st=%s", I.coop, I.nx, I.nf, I.gs1,
	                               I.far_, I.tab, I.asm_loop ? "asm" : "c++", I.coop, I.fused_st ? "kernel" : "prepass");
	else if (I.tile) snprintf(buf, len, "chain_dp_tile<NX=%d,NF=%d,SKIP=%d,GEN=%d,GS1=%d,FAR=%d,TAB=%d> loop=%s classes=%d cut=%d compact=%d q24=%d", I.nx, I.nf, I.skip, I.gen, I.gs1,
	                     I.far_, I.tab, I.asm_loop ? "asm" : "c++", I.classes, I.cut, I.c16, I.q24);
	else snprintf(buf, len, "chain_dp_wave<R=%d,SKIP=%d,GEN=%d,GS1=%d,FAR=%d> loop=c++ classes=0 cut=%d", I.r, I.skip, I.gen, I.gs1, I.far_, I.cut);
}
void note_host_variant(const mm2c::LaunchInfo &I);   // mm2chain_host.cpp

int get_thread_ctx(ThreadCtx **out);
// context of the big-batch entries: the worker of a split batch gets the context of its device slot (as get_thread_ctx); every other caller gets
// ONE shared context, locked for the duration of the call -- a host whose mini-batches arrive on different pipeline threads (kt_pipeline,
// map.c:529-620) then reuses the same arenas instead of growing a set per thread
int get_batch_ctx(ThreadCtx **out, std::unique_lock<std::mutex> &hold);
int cur_device();                                    // device of the calling thread: a worker of a split batch drives its own, everyone else the primary
int n_devices();
bool in_split_worker();
// runs fn(part, k0, k1) for the contiguous task ranges mm2c_split_tasks gives, each on its own host thread bound to its own device context;
// returns the first non-zero code.  Only used when more than one device is configured and the caller is not itself such a worker.
bool should_split(int64_t total_anchors);
int run_split(int64_t n_tasks, const int64_t *h_offsets, const std::function<int(int, int64_t, int64_t)> &fn);
hipError_t create_partner_stream(hipStream_t *st);
int grow_device(char **p, size_t *cap, size_t need);
int grow_pinned(char **p, size_t *cap, size_t need, bool gpu_addressed = false);
inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
int check_params(const mm2c_params_t *p);
mm2c::KParams to_kparams(const mm2c_params_t *p);
int check_offsets(int64_t n_tasks, const int64_t *off);
int build_order(int64_t n_tasks, const int64_t *off, std::vector<int32_t> &order);
size_t layout_epilogue(mm2c::EpiArgs &E, char *base, size_t tot, size_t nt, size_t sort_tmp);
int epilogue_debug_phases();
hipError_t dev_alloc(void **out, size_t bytes);   // cached device memory for plans and one-shot calls (of the calling thread's current device)
void dev_free(void *p);                           // waits for the block's device first (as hipFree would), then parks the block in that device's cache
void dev_free_synced(void *p);                    // the caller has already made sure nothing is in flight on the block (one wait for many blocks)

// makes `dev` the calling thread's current device for the lifetime of the object and puts the caller's device back afterwards: an entry point
// must not change the current device of a host thread that also drives other GPUs (a torch caller, a worker of another library)
struct DeviceScope {
	int prev = -1; bool changed = false; hipError_t err = hipSuccess;
	explicit DeviceScope(int dev)
	{
		if (hipGetDevice(&prev) != hipSuccess) prev = -1;
		if (prev != dev) { err = hipSetDevice(dev); changed = err == hipSuccess; }
	}
	~DeviceScope() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
	DeviceScope(const DeviceScope &) = delete;
	DeviceScope &operator=(const DeviceScope &) = delete;
};
// the stream a stream argument of the C ABI stands for; MM2C_STREAM_LIBRARY is the library's stream, which lives on the primary device
inline int resolve_stream(void *stream, int device, hipStream_t *out)
{
	if (stream == MM2C_STREAM_LIBRARY) {
		if (device != G.device) return fail(MM2C_E_ARG, "MM2C_STREAM_LIBRARY belongs to the primary device %d; this plan lives on device %d: pass a stream of that device", G.device, device);
		*out = G.stream;
	} else *out = (hipStream_t)stream;             // NULL = the HIP null stream (what a default-stream caller such as PyTorch works on)
	return 0;
}
void dev_cache_release();
void release_combiner();                            // mm2chain_host.cpp
int get_slot_stats(int slot, uint64_t *passes, uint64_t *calls, uint64_t *anchors);   // per-device combiner counters (mm2chain_host.cpp)
int book_pred(int tid, float hw_ms, float sw_ms);   // path A's decline: the slot that books the call, or -1 (mm2chain_host.cpp)
void release_pred(int slot, float hw_ms);
void release_seed_aux();                            // mm2chain_seeds.cpp
// a pooled set of helper streams (distinct priorities = hardware queues of their own) and fork / join events, kept between plans (mm2chain_seeds.cpp)
struct AuxSet { hipStream_t aux[3] = {}; hipEvent_t fork[4] = {}; int device = -1; uint64_t epoch = 0; };   // epoch: G.epoch when the set was made (a set of an earlier one is destroyed, not pooled)
hipError_t aux_acquire(int device, AuxSet *out);
void aux_release(const AuxSet &a);
// for callers that have already waited for the stream(s) the plan ran on: no device-wide wait (chunks of a pipelined batch overlap)
void plan_destroy_synced(mm2c_plan_t *pl);          // mm2chain_api.cpp
void seedplan_destroy_synced(mm2c_seedplan_t *pl);  // mm2chain_seeds.cpp

} // namespace mm2c_api
#endif
