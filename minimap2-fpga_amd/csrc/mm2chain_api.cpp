// mm2chain_api.cpp -- the C-ABI shim over HIP (include/mm2chain.h).
//
// Replaces the reference's XRT/OpenCL enqueue path: hardware_init (chain_hardware.cpp:278-400: platform, xclbin,
// kernel object, four cl_mem buffers), the body of run_chaining_on_hw (chain_hardware.cpp:104-189: two
// clEnqueueWriteBuffer, clEnqueueTask, two clEnqueueReadBuffer, clFinish) and cleanup (chain_hardware.cpp:403-441).
// Differences by design: no busy/queue time model and no "declined, do it on the CPU" return (chain_hardware.cpp:54-93)
// -- every accepted call is computed on the GPU; many tasks are in flight at once (one wave each) instead of one
// task per kernel; callers on different host threads get their own stream and staging buffers instead of a mutex.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <vector>
#include "mm2chain.h"
#include "chain_kernel.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	return fail(MM2C_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

struct ThreadCtx;
void release_combiner();

struct Global {
	std::mutex mu;
	bool ready = false;
	int device = -1;
	int ring_class = 0;
	size_t combine_max_anchors = 1u << 17;   // host paths: calls up to this many anchors are combined with concurrent callers' calls
	size_t stage_max_anchors = 1u << 21;     // host paths: calls up to this many anchors go through pinned staging buffers
	int64_t pipeline_chunk_anchors = 20 << 20;  // host paths: batches of at least twice this size are pipelined in chunks of this size
	int64_t cut_below_tasks = 4096;         // host paths: passes with at least this many tasks are not cut (they fill the GPU anyway)
	int seg_min = 256;                      // host paths: shortest piece a task is cut into at empty-window positions (0 = never cut)
	hipStream_t stream = nullptr;           // library stream for plan runs with stream == NULL
	std::vector<ThreadCtx *> thread_ctxs;   // owned; released in mm2c_shutdown
	std::atomic<uint64_t> tasks{0}, anchors{0}, launches{0}, segments{0}, host_call_ns{0}, passes{0};
	uint64_t epoch = 0;                     // bumped by shutdown so stale thread-local pointers are dropped
} G;

// per host thread: stream + grow-only buffers (the reference keeps one buffer set per FPGA kernel,
// chain_hardware.cpp:13-16,379-397, and serialises callers on a mutex).  One device arena for everything that is uploaded
// ([anchors | piece offsets | launch order | p base | avg | status]) and one for everything that is downloaded ([f | p]), each
// mirrored by a pinned host staging buffer, so that a call is one H2D copy, the kernels, one D2H copy and one sync.
// Device memory of plans and one-shot calls goes through a small cache: a batched caller creates and destroys plans of similar size
// for every mini-batch, and hipMalloc / hipFree of gigabytes cost milliseconds each (hipFree also waits for the device).  A freed
// block is kept and handed to the next request it fits (size within 1x..1.5x); the cache is bounded and emptied by mm2c_shutdown.
struct DevCache {
	std::mutex mu;
	struct Block { void *p; size_t size; };
	std::vector<Block> free_blocks;                 // cached, not in use
	std::vector<Block> live;                        // handed out (to know their size on free)
	size_t cached_bytes = 0;
	static constexpr size_t MAX_CACHED = (size_t)64 << 30;
} DC;

hipError_t dev_alloc(void **out, size_t bytes)
{
	if (bytes == 0) bytes = 1;
	{
		std::lock_guard<std::mutex> lk(DC.mu);
		size_t best = (size_t)-1;
		for (size_t i = 0; i < DC.free_blocks.size(); ++i) {
			const size_t sz = DC.free_blocks[i].size;
			if (sz >= bytes && sz <= bytes + bytes / 2 + 4096 && (best == (size_t)-1 || sz < DC.free_blocks[best].size)) best = i;
		}
		if (best != (size_t)-1) {
			DevCache::Block b = DC.free_blocks[best];
			DC.free_blocks.erase(DC.free_blocks.begin() + (long)best);
			DC.cached_bytes -= b.size;
			DC.live.push_back(b);
			*out = b.p;
			return hipSuccess;
		}
	}
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, bytes);
	if (e != hipSuccess) {                                     // out of memory with blocks parked in the cache: release them and retry
		std::vector<DevCache::Block> drop;
		{ std::lock_guard<std::mutex> lk(DC.mu); drop.swap(DC.free_blocks); DC.cached_bytes = 0; }
		for (auto &b : drop) (void)hipFree(b.p);
		(void)hipGetLastError();
		e = hipMalloc(&p, bytes);
		if (e != hipSuccess) return e;
	}
	std::lock_guard<std::mutex> lk(DC.mu);
	DC.live.push_back({p, bytes});
	*out = p;
	return hipSuccess;
}

void dev_free(void *p)
{
	if (!p) return;
	DevCache::Block b{p, 0};
	bool park = false;
	{
		std::lock_guard<std::mutex> lk(DC.mu);
		for (size_t i = 0; i < DC.live.size(); ++i)
			if (DC.live[i].p == p) { b = DC.live[i]; DC.live.erase(DC.live.begin() + (long)i); break; }
		if (b.size != 0 && DC.cached_bytes + b.size <= DevCache::MAX_CACHED && DC.free_blocks.size() < 64) {
			DC.free_blocks.push_back(b); DC.cached_bytes += b.size; park = true;
		}
	}
	if (!park) (void)hipFree(p);
}

void dev_cache_release()
{
	std::vector<DevCache::Block> drop;
	{ std::lock_guard<std::mutex> lk(DC.mu); drop.swap(DC.free_blocks); DC.cached_bytes = 0; DC.live.clear(); }
	for (auto &b : drop) (void)hipFree(b.p);
}

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

struct ThreadCtx {
	WholeSlot whole[2];
	hipStream_t st = nullptr, st2 = nullptr;   // st2: second stream of the pipelined big-batch path
	hipEvent_t ev = nullptr;
	char *d_in = nullptr, *d_out = nullptr, *d_scratch = nullptr;   // device
	char *h_in = nullptr, *h_out = nullptr;                          // pinned host
	size_t cap_in = 0, cap_out = 0, cap_scratch = 0, cap_hin = 0, cap_hout = 0;
	void release()
	{
		if (d_in) (void)hipFree(d_in); if (d_out) (void)hipFree(d_out); if (d_scratch) (void)hipFree(d_scratch);
		if (h_in) (void)hipHostFree(h_in); if (h_out) (void)hipHostFree(h_out);
		if (st) (void)hipStreamDestroy(st); if (st2) (void)hipStreamDestroy(st2); if (ev) (void)hipEventDestroy(ev);
		whole[0].release(); whole[1].release();
		*this = ThreadCtx();
	}
};

thread_local ThreadCtx *tl_ctx = nullptr;
thread_local uint64_t tl_epoch = 0;

int get_thread_ctx(ThreadCtx **out)
{
	std::lock_guard<std::mutex> lk(G.mu);
	if (!G.ready) return fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device");
	if (tl_ctx && tl_epoch == G.epoch) { *out = tl_ctx; return 0; }
	HIP_TRY(hipSetDevice(G.device));
	ThreadCtx *c = new ThreadCtx();
	hipError_t e = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking);
	if (e != hipSuccess) { delete c; return fail(MM2C_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
	G.thread_ctxs.push_back(c);
	tl_ctx = c; tl_epoch = G.epoch;
	*out = c;
	return 0;
}

// Second stream of a two-stream pipeline.  The runtime spreads streams of equal priority over a few hardware queues (4 by default)
// in creation order, so two streams of one pipeline can end up on the same queue and then run strictly one after the other
// (measured: no overlap at all with the default GPU_MAX_HW_QUEUES=4).  Streams of different priority never share a queue.
hipError_t create_partner_stream(hipStream_t *st)
{
	int least = 0, greatest = 0;
	hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
	if (e != hipSuccess || greatest == least) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
	return hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest);
}

int grow_device(char **p, size_t *cap, size_t need)
{
	if (need <= *cap) return 0;
	const size_t want = std::max(need, *cap * 2);
	if (*p) (void)hipFree(*p);
	*p = nullptr; *cap = 0;
	HIP_TRY(hipMalloc((void **)p, want));
	*cap = want;
	return 0;
}

int grow_pinned(char **p, size_t *cap, size_t need)
{
	if (need <= *cap) return 0;
	const size_t want = std::max(need, *cap * 2);
	if (*p) (void)hipHostFree(*p);
	*p = nullptr; *cap = 0;
	HIP_TRY(hipHostMalloc((void **)p, want, hipHostMallocDefault));
	*cap = want;
	return 0;
}

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

int check_params(const mm2c_params_t *p)
{
	if (!p) return fail(MM2C_E_ARG, "params is NULL");
	if (p->max_dist_x < 0) return fail(MM2C_E_ARG, "max_dist_x must be >= 0 (got %d)", p->max_dist_x);
	return 0;
}

mm2c::KParams to_kparams(const mm2c_params_t *p)
{
	mm2c::KParams k;
	k.max_dist_x = p->max_dist_x; k.max_dist_y = p->max_dist_y; k.bw = p->bw;
	k.max_skip = p->max_skip;
	k.max_iter = std::max(p->max_iter, 0);      // a negative max_iter leaves no predecessor at all (chain.c:193), same as 0
	k.is_cdna = p->is_cdna; k.n_segs = p->n_segs;
	k.span_override = p->q_span_override;
	k.max_dq = std::max(std::min(p->max_dist_y, p->max_dist_x), 0);
	k.flags = 0;
	if (p->flags & MM2C_F_IGNORE_SEG) k.flags |= mm2c::KF_IGNORE_SEG;
	if (p->flags & MM2C_F_FORCE_GENERAL) k.flags |= mm2c::KF_FORCE_GENERAL;
	k.gap_scale = p->gap_scale;
	return k;
}

// offsets sanity + longest-first launch order (so the tail of the grid is made of short tasks)
int build_order(int64_t n_tasks, const int64_t *off, std::vector<int32_t> &order)
{
	if (n_tasks < 0 || n_tasks > INT32_MAX) return fail(MM2C_E_ARG, "n_tasks out of range");
	if (n_tasks > 0 && !off) return fail(MM2C_E_ARG, "offsets is NULL");
	for (int64_t k = 0; k < n_tasks; ++k) {
		const int64_t n = off[k + 1] - off[k];
		if (n < 0) return fail(MM2C_E_ARG, "offsets not monotone at task %lld", (long long)k);
		if (n >= (int64_t)INT32_MAX - 64)
			return fail(MM2C_E_TOOBIG, "task %lld has %lld anchors; the limit is 2^31-65 (cf. chain_hardware.cpp:34)", (long long)k, (long long)n);
	}
	order.resize((size_t)n_tasks);
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return off[x + 1] - off[x] > off[y + 1] - off[y]; });
	return 0;
}


// carves the scratch of the device epilogue (chain_epilogue.hip) out of one allocation; returns the bytes needed (base may be NULL)
size_t layout_epilogue(mm2c::EpiArgs &E, char *base, size_t tot, size_t nt, size_t sort_tmp)
{
	size_t at = 0;
	auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return base + o; };
	E.key0 = (uint64_t *)take(tot * 8); E.key1 = (uint64_t *)take(tot * 8); E.u2 = (uint64_t *)take(tot * 8); E.rkey1 = (uint64_t *)take(tot * 8);
	E.v = (int32_t *)take(tot * 4); E.own = (int32_t *)take(tot * 4); E.ctop = (int32_t *)take(tot * 4); E.rk2kk = (int32_t *)take(tot * 4);
	E.dest = (int32_t *)take(tot * 4); E.val0 = (int32_t *)take(tot * 4); E.val1 = (int32_t *)take(tot * 4);
	E.seg_begin = (uint32_t *)take(nt * 4); E.seg_end1 = (uint32_t *)take(nt * 4); E.seg_end2 = (uint32_t *)take(nt * 4);
	E.cnt_u = (int32_t *)take(nt * 4); E.cnt_b = (int32_t *)take(nt * 4);
	E.sort_tmp = take(sort_tmp ? sort_tmp : 1); E.sort_tmp_bytes = sort_tmp;
	return at;
}

int epilogue_debug_phases() { const char *dbg = getenv("MM2C_EPI_PHASES"); return dbg ? atoi(dbg) : 0; }

} // namespace

struct mm2c_plan {
	mm2c_params_t par;
	int64_t n_tasks = 0, total = 0;
	int64_t *d_off = nullptr; int32_t *d_order = nullptr, *d_status = nullptr, *d_t = nullptr, *d_st = nullptr;
	hipEvent_t ev_pre = nullptr, ev0 = nullptr, ev1 = nullptr, ev_e0 = nullptr, ev_e1 = nullptr;
	bool ran = false, epi_ran = false;
	char *d_epi = nullptr;                  // scratch of the device epilogue, allocated by the first mm2c_plan_chains_device
	mm2c::EpiArgs E;
};

extern "C" {

const char *mm2c_last_error(void) { return g_err; }

int mm2c_init(int device_ordinal)
{
	std::lock_guard<std::mutex> lk(G.mu);
	if (G.ready) return 0;
	int n_dev = 0;
	hipError_t e = hipGetDeviceCount(&n_dev);
	if (e != hipSuccess || n_dev <= 0)
		return fail(MM2C_E_NODEVICE, "no HIP device: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
	int dev = device_ordinal;
	if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
	if (dev >= n_dev) return fail(MM2C_E_NODEVICE, "device ordinal %d out of range (%d devices)", dev, n_dev);
	HIP_TRY(hipSetDevice(dev));
	HIP_TRY(hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking));
	G.device = dev;
	const char *rc = getenv("MM2C_RING_CLASS");
	G.ring_class = rc ? std::max(0, std::min(2, atoi(rc))) : 0;
	G.ready = true;
	return 0;
}

void mm2c_shutdown(void)
{
	std::lock_guard<std::mutex> lk(G.mu);
	if (!G.ready) return;
	(void)hipSetDevice(G.device);
	(void)hipDeviceSynchronize();
	for (ThreadCtx *c : G.thread_ctxs) { c->release(); delete c; }
	release_combiner();
	dev_cache_release();
	G.thread_ctxs.clear();
	if (G.stream) (void)hipStreamDestroy(G.stream);
	G.stream = nullptr;
	G.ready = false;
	++G.epoch;
}

int mm2c_device_info(char *name, size_t name_len, int *cu_count, size_t *hbm_bytes)
{
	if (!G.ready) return fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device");
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, G.device));
	if (name && name_len) { snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName); }
	if (cu_count) *cu_count = prop.multiProcessorCount;
	if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
	return 0;
}

int mm2c_tune(const char *key, int value)
{
	if (!key) return fail(MM2C_E_ARG, "key is NULL");
	if (strcmp(key, "ring_class") == 0) {
		if (value < 0 || value > 2) return fail(MM2C_E_ARG, "ring_class must be 0, 1 or 2");
		G.ring_class = value;
		return 0;
	}
	if (strcmp(key, "pipeline_chunk_anchors") == 0) {
		if (value < 1024) return fail(MM2C_E_ARG, "pipeline_chunk_anchors must be >= 1024");
		G.pipeline_chunk_anchors = value;
		return 0;
	}
	if (strcmp(key, "seg_min") == 0) {
		if (value < 0) return fail(MM2C_E_ARG, "seg_min must be >= 0");
		G.seg_min = value;
		return 0;
	}
	return fail(MM2C_E_ARG, "unknown tuning key '%s'", key);
}

void mm2c_params_map_ont(mm2c_params_t *p)
{
	// options.c:24-31 (mm_mapopt_init) + :93-99 (map-ont keeps them); map.c:305-316 passes max_gap as both max_dist
	p->max_dist_x = 5000; p->max_dist_y = 5000; p->bw = 500;
	p->max_skip = 25; p->max_iter = 5000; p->gap_scale = 1.0f;
	p->is_cdna = 0; p->n_segs = 1; p->q_span_override = -1; p->flags = 0;
}

void mm2c_params_fpga_v2(mm2c_params_t *p, int32_t max_dist_x, int32_t max_dist_y, int32_t bw, int32_t q_span)
{
	// device/minimap2_opencl.cl:116-127: no max_skip, look-back <= 128*8 (chain_hardware.h:58-60), one q_span,
	// segment ids ignored, gap_scale 1
	p->max_dist_x = max_dist_x; p->max_dist_y = max_dist_y; p->bw = bw;
	p->max_skip = INT32_MAX; p->max_iter = 1024; p->gap_scale = 1.0f;
	p->is_cdna = 0; p->n_segs = 1; p->q_span_override = q_span; p->flags = MM2C_F_IGNORE_SEG;
}

void mm2c_get_stats(mm2c_stats_t *out)
{
	if (!out) return;
	out->tasks = G.tasks.load(); out->anchors = G.anchors.load(); out->launches = G.launches.load(); out->segments = G.segments.load(); out->host_call_ns = G.host_call_ns.load(); out->passes = G.passes.load();
}

// ------------------------------------------------------------------------------------------------ plans
mm2c_plan_t *mm2c_plan_create(const mm2c_params_t *par, int64_t n_tasks, const int64_t *h_offsets)
{
	if (check_params(par)) return nullptr;
	if (!G.ready) { fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device"); return nullptr; }
	std::vector<int32_t> order;
	if (build_order(n_tasks, h_offsets, order)) return nullptr;
	mm2c_plan *pl = new mm2c_plan();
	pl->par = *par; pl->n_tasks = n_tasks;
	pl->total = n_tasks > 0 ? h_offsets[n_tasks] - h_offsets[0] : 0;
	hipError_t e = hipSetDevice(G.device);
	const size_t nt = (size_t)std::max<int64_t>(n_tasks, 1), tot = (size_t)std::max<int64_t>(pl->total, 1);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_off, (nt + 1) * 8);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_order, nt * 4);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_status, nt * 4);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_t, tot * 4);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_st, tot * 4);
	if (e == hipSuccess && n_tasks > 0) {
		// rebase offsets so that task 0 starts at 0 of the arrays handed to mm2c_plan_run_device
		std::vector<int64_t> off((size_t)n_tasks + 1);
		for (int64_t k = 0; k <= n_tasks; ++k) off[(size_t)k] = h_offsets[k] - h_offsets[0];
		e = hipMemcpy(pl->d_off, off.data(), ((size_t)n_tasks + 1) * 8, hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMemcpy(pl->d_order, order.data(), (size_t)n_tasks * 4, hipMemcpyHostToDevice);
	}
	if (e == hipSuccess) e = hipEventCreate(&pl->ev_pre);
	if (e == hipSuccess) e = hipEventCreate(&pl->ev0);
	if (e == hipSuccess) e = hipEventCreate(&pl->ev1);
	if (e != hipSuccess) {
		fail(MM2C_E_HIP, "mm2c_plan_create: %s", hipGetErrorString(e));
		mm2c_plan_destroy(pl);
		return nullptr;
	}
	return pl;
}

void mm2c_plan_destroy(mm2c_plan_t *pl)
{
	if (!pl) return;
	if (pl->ran || pl->epi_ran) (void)hipDeviceSynchronize();   // as hipFree would: the blocks go back to the cache and may be reused at once
	dev_free(pl->d_off); dev_free(pl->d_order); dev_free(pl->d_status); dev_free(pl->d_t); dev_free(pl->d_st);
	if (pl->ev_pre) (void)hipEventDestroy(pl->ev_pre);
	if (pl->ev0) (void)hipEventDestroy(pl->ev0); if (pl->ev1) (void)hipEventDestroy(pl->ev1);
	if (pl->ev_e0) (void)hipEventDestroy(pl->ev_e0); if (pl->ev_e1) (void)hipEventDestroy(pl->ev_e1);
	dev_free(pl->d_epi);
	delete pl;
}

int64_t mm2c_plan_total_anchors(const mm2c_plan_t *pl) { return pl ? pl->total : 0; }

int mm2c_plan_run_device(mm2c_plan_t *pl, const void *d_anchors, const float *d_avg_qspan, int32_t *d_f, int32_t *d_p, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (!G.ready) return fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device");
	if (pl->n_tasks == 0 || pl->total == 0) return 0;
	if (!d_anchors || !d_f || !d_p) return fail(MM2C_E_ARG, "device pointer is NULL");
	hipStream_t st = stream ? (hipStream_t)stream : G.stream;
	mm2c::LaunchArgs L;
	L.P = to_kparams(&pl->par);
	L.n_tasks = pl->n_tasks; L.d_offsets = pl->d_off; L.d_order = pl->d_order;
	L.d_anchors = d_anchors; L.d_avg = d_avg_qspan; L.d_pbase = nullptr; L.d_f = d_f; L.d_p = d_p; L.d_t = pl->d_t; L.d_st = pl->d_st; L.d_status = pl->d_status;
	L.ring_class = G.ring_class;
	HIP_TRY(hipMemsetAsync(pl->d_status, 0, (size_t)pl->n_tasks * 4, st));
	HIP_TRY(hipEventRecord(pl->ev_pre, st));
	int nl = 0;
	HIP_TRY(mm2c::launch_chain_dp(L, st, &nl, pl->ev0));
	HIP_TRY(hipEventRecord(pl->ev1, st));
	pl->ran = true;
	G.tasks += (uint64_t)pl->n_tasks; G.anchors += (uint64_t)pl->total; G.launches += (uint64_t)nl;
	return 0;
}

int mm2c_plan_predict_device(mm2c_plan_t *pl, const void *d_anchors, uint8_t *d_num_subparts, int64_t *d_total_subparts,
                             int64_t *d_total_trip_count, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (!G.ready) return fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device");
	if (pl->n_tasks == 0) return 0;
	if (!d_anchors && pl->total > 0) return fail(MM2C_E_ARG, "device pointer is NULL");
	hipStream_t st = stream ? (hipStream_t)stream : G.stream;
	HIP_TRY(mm2c::launch_predict(pl->par.max_dist_x, pl->n_tasks, pl->d_off, pl->d_order, d_anchors, d_num_subparts,
	                             d_total_subparts, d_total_trip_count, st));
	G.launches += 1;
	return 0;
}

int mm2c_plan_last_prepass_ms(mm2c_plan_t *pl, float *ms)
{
	if (!pl || !ms) return fail(MM2C_E_ARG, "NULL argument");
	if (!pl->ran) return fail(MM2C_E_ARG, "plan has not been run");
	HIP_TRY(hipEventSynchronize(pl->ev0));
	HIP_TRY(hipEventElapsedTime(ms, pl->ev_pre, pl->ev0));
	return 0;
}

int mm2c_plan_last_kernel_ms(mm2c_plan_t *pl, float *ms)
{
	if (!pl || !ms) return fail(MM2C_E_ARG, "NULL argument");
	if (!pl->ran) return fail(MM2C_E_ARG, "plan has not been run");
	HIP_TRY(hipEventSynchronize(pl->ev1));
	HIP_TRY(hipEventElapsedTime(ms, pl->ev0, pl->ev1));
	return 0;
}

// the epilogue of mm_chain_dp (chain.c:106-111,348-422) for every task of the plan, on the GPU (chain_epilogue.hip)
int mm2c_plan_chains_device(mm2c_plan_t *pl, const void *d_anchors, const int32_t *d_f, const int32_t *d_p, int min_cnt, int min_sc,
                            int64_t *d_u_off, uint64_t *d_u, int64_t *d_b_off, void *d_b, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (!G.ready) return fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device");
	if (!d_u_off || !d_b_off) return fail(MM2C_E_ARG, "device pointer is NULL");
	hipStream_t st = stream ? (hipStream_t)stream : G.stream;
	if (pl->n_tasks == 0 || pl->total == 0) {
		HIP_TRY(hipMemsetAsync(d_u_off, 0, ((size_t)pl->n_tasks + 1) * 8, st));
		HIP_TRY(hipMemsetAsync(d_b_off, 0, ((size_t)pl->n_tasks + 1) * 8, st));
		return 0;
	}
	if (!d_anchors || !d_f || !d_p || !d_u || !d_b) return fail(MM2C_E_ARG, "device pointer is NULL");
	if (pl->total >= (int64_t)INT32_MAX) return fail(MM2C_E_TOOBIG, "the device epilogue takes batches of fewer than 2^31 anchors (got %lld)", (long long)pl->total);
	mm2c::EpiArgs &E = pl->E;
	if (!pl->d_epi) {
		HIP_TRY(hipSetDevice(G.device));
		const size_t tmp = mm2c::epilogue_sort_temp_bytes(pl->total, pl->n_tasks);
		const size_t bytes = layout_epilogue(E, nullptr, (size_t)pl->total, (size_t)pl->n_tasks, tmp);
		HIP_TRY(dev_alloc((void **)&pl->d_epi, bytes));
		layout_epilogue(E, pl->d_epi, (size_t)pl->total, (size_t)pl->n_tasks, tmp);
		HIP_TRY(hipEventCreate(&pl->ev_e0));
		HIP_TRY(hipEventCreate(&pl->ev_e1));
	}
	E.n_tasks = pl->n_tasks; E.total = pl->total; E.d_off = pl->d_off; E.d_order = pl->d_order;
	E.d_a = (const ulonglong2 *)d_anchors; E.d_f = d_f; E.d_p = d_p; E.min_cnt = min_cnt; E.min_sc = min_sc;
	E.debug_phases = epilogue_debug_phases();
	E.u_off = d_u_off; E.b_off = d_b_off; E.u_out = d_u; E.b_out = (ulonglong2 *)d_b;
	int nl = 0;
	HIP_TRY(hipEventRecord(pl->ev_e0, st));
	HIP_TRY(mm2c::launch_chain_epilogue(E, st, &nl));
	HIP_TRY(hipEventRecord(pl->ev_e1, st));
	pl->epi_ran = true;
	G.launches += (uint64_t)nl;
	return 0;
}

int mm2c_plan_last_epilogue_ms(mm2c_plan_t *pl, float *ms)
{
	if (!pl || !ms) return fail(MM2C_E_ARG, "NULL argument");
	if (!pl->epi_ran) return fail(MM2C_E_ARG, "mm2c_plan_chains_device has not been run");
	HIP_TRY(hipEventSynchronize(pl->ev_e1));
	HIP_TRY(hipEventElapsedTime(ms, pl->ev_e0, pl->ev_e1));
	return 0;
}

// ------------------------------------------------------------------------------------------------ seed hits -> anchors
} // extern "C"

struct mm2c_seedplan {
	int64_t n_reads = 0, total = 0, n_matches = 0;
	char *d_mem = nullptr;                 // [match_off | anchor_off | order | status | has_ties | stack | unsorted | scratch | big_id | big_dg]
	mm2c::SeedArgs S;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	hipStream_t aux[3] = {};               // helper streams: the size classes of the tie replay run side by side
	hipEvent_t fork[4] = {};
	bool ran = false;
};

extern "C" {

mm2c_seedplan_t *mm2c_seedplan_create(int64_t n_reads, const int64_t *h_match_off, const int64_t *h_anchor_off)
{
	if (!G.ready) { fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device"); return nullptr; }
	if (n_reads < 0 || n_reads > INT32_MAX || (n_reads > 0 && (!h_match_off || !h_anchor_off))) { fail(MM2C_E_ARG, "bad argument"); return nullptr; }
	std::vector<int32_t> order;
	if (build_order(n_reads, h_anchor_off, order)) return nullptr;            // validates the anchor offsets; biggest read first
	int64_t biggest = 0;
	for (int64_t r = 0; r < n_reads; ++r) {
		if (h_match_off[r + 1] < h_match_off[r]) { fail(MM2C_E_ARG, "match offsets not monotone at read %lld", (long long)r); return nullptr; }
		biggest = std::max(biggest, h_anchor_off[r + 1] - h_anchor_off[r]);
	}
	mm2c_seedplan *pl = new mm2c_seedplan();
	pl->n_reads = n_reads;
	pl->total = n_reads ? h_anchor_off[n_reads] - h_anchor_off[0] : 0;
	pl->n_matches = n_reads ? h_match_off[n_reads] - h_match_off[0] : 0;
	const size_t nr = (size_t)std::max<int64_t>(n_reads, 1), tot = (size_t)std::max<int64_t>(pl->total, 1);
	const bool big = biggest > mm2c::seed_tie_lds_max();
	size_t at = 0;
	auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return o; };
	const size_t o_moff = take((nr + 1) * 8), o_aoff = take((nr + 1) * 8), o_ord = take(nr * 4), o_stat = take(nr * 4), o_ties = take(nr * 4),
	             o_stack = take(2 * (tot / 64 + 2 * nr + 2) * 4), o_un = take(tot * 16), o_scr = take(tot * 16), o_tc = take(tot * 4), o_xd = take(nr * 8),
	             o_bid = take(big ? tot * 4 : 1), o_bdg = take(big ? tot : 1);
	hipError_t e = hipSetDevice(G.device);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_mem, at);
	if (e == hipSuccess && n_reads > 0) {
		std::vector<int64_t> off((size_t)n_reads + 1);
		for (int64_t k = 0; k <= n_reads; ++k) off[(size_t)k] = h_match_off[k] - h_match_off[0];
		e = hipMemcpy(pl->d_mem + o_moff, off.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice);
		for (int64_t k = 0; k <= n_reads; ++k) off[(size_t)k] = h_anchor_off[k] - h_anchor_off[0];
		if (e == hipSuccess) e = hipMemcpy(pl->d_mem + o_aoff, off.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMemcpy(pl->d_mem + o_ord, order.data(), (size_t)n_reads * 4, hipMemcpyHostToDevice);
	}
	if (e == hipSuccess) e = hipEventCreate(&pl->ev0);
	if (e == hipSuccess) e = hipEventCreate(&pl->ev1);
	{	// helper streams on hardware queues of their own: different priorities never share a queue (see create_partner_stream)
		int least = 0, greatest = 0;
		if (e == hipSuccess && hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
		const int prio[3] = { greatest, least, (least + greatest) / 2 };
		for (int i = 0; i < 3 && e == hipSuccess; ++i)
			e = least != greatest ? hipStreamCreateWithPriority(&pl->aux[i], hipStreamNonBlocking, prio[i]) : hipStreamCreateWithFlags(&pl->aux[i], hipStreamNonBlocking);
	}
	for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&pl->fork[i], hipEventDisableTiming);
	if (e != hipSuccess) { fail(MM2C_E_HIP, "mm2c_seedplan_create: %s", hipGetErrorString(e)); mm2c_seedplan_destroy(pl); return nullptr; }
	mm2c::SeedArgs &S = pl->S;
	char *b = pl->d_mem;
	S.n_reads = n_reads; S.d_match_off = (const int64_t *)(b + o_moff); S.d_anchor_off = (const int64_t *)(b + o_aoff);
	S.d_order = (const int32_t *)(b + o_ord); S.status = (int32_t *)(b + o_stat); S.has_ties = (int32_t *)(b + o_ties);
	S.tiecnt = (int32_t *)(b + o_tc); S.xdiff = (uint64_t *)(b + o_xd); S.biggest = biggest;
	S.stack = (int32_t *)(b + o_stack); S.unsorted = (ulonglong2 *)(b + o_un); S.scratch = (ulonglong2 *)(b + o_scr);
	S.big_id = big ? (uint32_t *)(b + o_bid) : nullptr; S.big_dg = big ? (uint8_t *)(b + o_bdg) : nullptr;
	return pl;
}

void mm2c_seedplan_destroy(mm2c_seedplan_t *pl)
{
	if (!pl) return;
	if (pl->ran) (void)hipDeviceSynchronize();
	dev_free(pl->d_mem);
	if (pl->ev0) (void)hipEventDestroy(pl->ev0); if (pl->ev1) (void)hipEventDestroy(pl->ev1);
	for (int i = 0; i < 3; ++i) if (pl->aux[i]) (void)hipStreamDestroy(pl->aux[i]);
	for (int i = 0; i < 4; ++i) if (pl->fork[i]) (void)hipEventDestroy(pl->fork[i]);
	delete pl;
}

int mm2c_seedplan_run_device(mm2c_seedplan_t *pl, const mm2c_match_t *d_matches, const uint64_t *d_hits, const int32_t *d_qlen,
                             void *d_anchors, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (!G.ready) return fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device");
	if (pl->n_reads == 0) return 0;
	if (!d_qlen || (pl->n_matches > 0 && !d_matches) || (pl->total > 0 && (!d_hits || !d_anchors))) return fail(MM2C_E_ARG, "device pointer is NULL");
	static_assert(sizeof(mm2c_match_t) == sizeof(mm2c::Match), "mm2c_match_t layout");
	hipStream_t st = stream ? (hipStream_t)stream : G.stream;
	mm2c::SeedArgs &S = pl->S;
	S.d_matches = (const mm2c::Match *)d_matches; S.d_hits = d_hits; S.d_qlen = d_qlen; S.d_anchors = (ulonglong2 *)d_anchors;
	HIP_TRY(hipMemsetAsync(S.status, 0, (size_t)pl->n_reads * 4, st));
	HIP_TRY(hipMemsetAsync(S.has_ties, 0, (size_t)pl->n_reads * 4, st));
	HIP_TRY(hipEventRecord(pl->ev0, st));
	int nl = 0;
	HIP_TRY(mm2c::launch_seed_hits(S, st, &nl, pl->aux, pl->fork));
	HIP_TRY(hipEventRecord(pl->ev1, st));
	pl->ran = true;
	G.launches += (uint64_t)nl;
	return 0;
}

int mm2c_seedplan_check(mm2c_seedplan_t *pl, int64_t *n_reads_with_ties)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (n_reads_with_ties) *n_reads_with_ties = 0;
	if (!pl->ran || pl->n_reads == 0) return 0;
	HIP_TRY(hipEventSynchronize(pl->ev1));
	std::vector<int32_t> st((size_t)pl->n_reads), ti((size_t)pl->n_reads);
	HIP_TRY(hipMemcpy(st.data(), pl->S.status, st.size() * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(ti.data(), pl->S.has_ties, ti.size() * 4, hipMemcpyDeviceToHost));
	int64_t nt = 0;
	for (size_t r = 0; r < st.size(); ++r) {
		if (st[r] != 0) return fail(MM2C_E_ARG, "read %zu: the hit counts of its matches do not add up to its anchor range", r);
		nt += ti[r] != 0;
	}
	if (n_reads_with_ties) *n_reads_with_ties = nt;
	return 0;
}

int mm2c_seedplan_last_ms(mm2c_seedplan_t *pl, float *ms)
{
	if (!pl || !ms) return fail(MM2C_E_ARG, "NULL argument");
	if (!pl->ran) return fail(MM2C_E_ARG, "plan has not been run");
	HIP_TRY(hipEventSynchronize(pl->ev1));
	HIP_TRY(hipEventElapsedTime(ms, pl->ev0, pl->ev1));
	return 0;
}

int mm2c_seed_hits_batch_host(int64_t n_reads, const int64_t *h_match_off, const mm2c_match_t *h_matches, const uint64_t *h_hits,
                              int64_t n_hits, const int32_t *h_qlen, int64_t *anchor_off, mm2c_anchor_t *anchors)
{
	if (n_reads < 0 || !anchor_off) return fail(MM2C_E_ARG, "bad argument");
	anchor_off[0] = 0;
	if (n_reads == 0) return 0;
	if (!h_match_off || !h_qlen) return fail(MM2C_E_ARG, "host pointer is NULL");
	const int64_t mb = h_match_off[0], n_m = h_match_off[n_reads] - mb;
	if (n_m > 0 && !h_matches) return fail(MM2C_E_ARG, "matches is NULL");
	for (int64_t r = 0; r < n_reads; ++r) {
		int64_t sum = 0;
		if (h_match_off[r + 1] < h_match_off[r]) return fail(MM2C_E_ARG, "match offsets not monotone at read %lld", (long long)r);
		for (int64_t i = h_match_off[r]; i < h_match_off[r + 1]; ++i) {
			if (h_matches[i].cr_off < 0 || h_matches[i].cr_off + (int64_t)h_matches[i].n > n_hits)
				return fail(MM2C_E_ARG, "match %lld reaches beyond the hit pool", (long long)i);
			sum += h_matches[i].n;
		}
		anchor_off[r + 1] = anchor_off[r] + sum;
	}
	const int64_t total = anchor_off[n_reads];
	if (total == 0) return 0;
	if (!h_hits || !anchors) return fail(MM2C_E_ARG, "host pointer is NULL");
	mm2c_seedplan_t *pl = mm2c_seedplan_create(n_reads, h_match_off, anchor_off);
	if (!pl) return MM2C_E_HIP;
	char *d = nullptr;
	size_t at = 0;
	auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return o; };
	const size_t o_m = take((size_t)n_m * sizeof(mm2c_match_t)), o_h = take((size_t)n_hits * 8), o_q = take((size_t)n_reads * 4), o_a = take((size_t)total * 16);
	auto body = [&]() -> int {
		int r;
		HIP_TRY(dev_alloc((void **)&d, at));
		HIP_TRY(hipMemcpyAsync(d + o_m, h_matches + mb, (size_t)n_m * sizeof(mm2c_match_t), hipMemcpyHostToDevice, G.stream));
		HIP_TRY(hipMemcpyAsync(d + o_h, h_hits, (size_t)n_hits * 8, hipMemcpyHostToDevice, G.stream));
		HIP_TRY(hipMemcpyAsync(d + o_q, h_qlen, (size_t)n_reads * 4, hipMemcpyHostToDevice, G.stream));
		if ((r = mm2c_seedplan_run_device(pl, (const mm2c_match_t *)(d + o_m), (const uint64_t *)(d + o_h), (const int32_t *)(d + o_q), d + o_a, G.stream))) return r;
		HIP_TRY(hipMemcpyAsync(anchors, d + o_a, (size_t)total * 16, hipMemcpyDeviceToHost, G.stream));
		HIP_TRY(hipStreamSynchronize(G.stream));
		return mm2c_seedplan_check(pl, nullptr);
	};
	const int rc = body();
	dev_free(d);
	mm2c_seedplan_destroy(pl);
	return rc;
}

// matches in, chains out: collect_seed_hits + mm_chain_dp for a batch of reads (map.c:295-316) without the anchors leaving the GPU
int mm2c_seed_chain_batch_host(const mm2c_params_t *par, int min_cnt, int min_sc, int64_t n_reads, const int64_t *h_match_off,
                               const mm2c_match_t *h_matches, const uint64_t *h_hits, int64_t n_hits, const int32_t *h_qlen,
                               int64_t *anchor_off, int64_t *u_off, uint64_t *u, int64_t *b_off, mm2c_anchor_t *b)
{
	int rc;
	if ((rc = check_params(par))) return rc;
	if (n_reads < 0 || !anchor_off || !u_off || !b_off) return fail(MM2C_E_ARG, "bad argument");
	anchor_off[0] = 0; u_off[0] = b_off[0] = 0;
	if (n_reads == 0) return 0;
	if (!h_match_off || !h_qlen) return fail(MM2C_E_ARG, "host pointer is NULL");
	const int64_t mb = h_match_off[0], n_m = h_match_off[n_reads] - mb;
	if (n_m > 0 && !h_matches) return fail(MM2C_E_ARG, "matches is NULL");
	for (int64_t r = 0; r < n_reads; ++r) {
		int64_t sum = 0;
		if (h_match_off[r + 1] < h_match_off[r]) return fail(MM2C_E_ARG, "match offsets not monotone at read %lld", (long long)r);
		for (int64_t i = h_match_off[r]; i < h_match_off[r + 1]; ++i) {
			if (h_matches[i].cr_off < 0 || h_matches[i].cr_off + (int64_t)h_matches[i].n > n_hits)
				return fail(MM2C_E_ARG, "match %lld reaches beyond the hit pool", (long long)i);
			sum += h_matches[i].n;
		}
		anchor_off[r + 1] = anchor_off[r] + sum;
	}
	const int64_t total = anchor_off[n_reads];
	if (total == 0) { for (int64_t r = 1; r <= n_reads; ++r) u_off[r] = b_off[r] = 0; return 0; }
	if (!h_hits || !u || !b) return fail(MM2C_E_ARG, "host pointer is NULL");
	if (total >= (int64_t)INT32_MAX) return fail(MM2C_E_TOOBIG, "batch of %lld anchors; the limit of one call is 2^31-1", (long long)total);
	mm2c_seedplan_t *sp = mm2c_seedplan_create(n_reads, h_match_off, anchor_off);
	if (!sp) return MM2C_E_HIP;
	mm2c_plan_t *pl = mm2c_plan_create(par, n_reads, anchor_off);
	if (!pl) { mm2c_seedplan_destroy(sp); return MM2C_E_HIP; }
	char *d = nullptr;
	size_t at = 0;
	auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return o; };
	const size_t nr = (size_t)n_reads, tot = (size_t)total;
	const size_t o_m = take((size_t)n_m * sizeof(mm2c_match_t)), o_h = take((size_t)n_hits * 8), o_q = take(nr * 4), o_a = take(tot * 16),
	             o_f = take(tot * 4), o_p = take(tot * 4), o_uo = take((nr + 1) * 8), o_bo = take((nr + 1) * 8), o_u = take(tot * 8), o_b = take(tot * 16);
	hipStream_t st = G.stream;
	auto body = [&]() -> int {
		int r;
		HIP_TRY(dev_alloc((void **)&d, at));
		HIP_TRY(hipMemcpyAsync(d + o_m, h_matches + mb, (size_t)n_m * sizeof(mm2c_match_t), hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(d + o_h, h_hits, (size_t)n_hits * 8, hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(d + o_q, h_qlen, nr * 4, hipMemcpyHostToDevice, st));
		if ((r = mm2c_seedplan_run_device(sp, (const mm2c_match_t *)(d + o_m), (const uint64_t *)(d + o_h), (const int32_t *)(d + o_q), d + o_a, st))) return r;
		if ((r = mm2c_plan_run_device(pl, d + o_a, nullptr, (int32_t *)(d + o_f), (int32_t *)(d + o_p), st))) return r;
		if ((r = mm2c_plan_chains_device(pl, d + o_a, (int32_t *)(d + o_f), (int32_t *)(d + o_p), min_cnt, min_sc, (int64_t *)(d + o_uo), (uint64_t *)(d + o_u),
		                                 (int64_t *)(d + o_bo), d + o_b, st))) return r;
		HIP_TRY(hipMemcpyAsync(u_off, d + o_uo, (nr + 1) * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(b_off, d + o_bo, (nr + 1) * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		if (u_off[nr] > 0) HIP_TRY(hipMemcpyAsync(u, d + o_u, (size_t)u_off[nr] * 8, hipMemcpyDeviceToHost, st));
		if (b_off[nr] > 0) HIP_TRY(hipMemcpyAsync(b, d + o_b, (size_t)b_off[nr] * 16, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		return mm2c_seedplan_check(sp, nullptr);
	};
	rc = body();
	dev_free(d);
	mm2c_plan_destroy(pl);
	mm2c_seedplan_destroy(sp);
	G.passes += 1;
	return rc;
}

// ------------------------------------------------------------------------------------------------ host-buffer paths
} // extern "C"

namespace {

// one caller's batch: CSR tasks in pageable host memory
struct HostReq {
	const mm2c_params_t *par; int64_t n_tasks; const int64_t *off; const mm2c_anchor_t *a; const float *avg;
	int32_t *f, *p;
	int rc = 0; bool done = false; char err[256];
};

// Runs one GPU pass over the union of the requests (all with the same scalars).
//  * every task is split at empty-window cut points: where x_i > x_{i-1} + max_dist_x no anchor at or
//    after i can chain to, stamp or be stamped by an anchor before i (chain.c:192 pushes st to i), so the pieces are
//    independent tasks for f[]/p[] and run as parallel waves.  Real reads hit many loci: this is what gives one mm_chain_dp
//    call more than one wave of work.  Pieces shorter than seg_min anchors are merged with their successor.
//  * avg_qspan_scaled is a whole-task quantity (chain.c:48-49): computed here per task unless handed in.
//  * one upload arena [anchors | piece offsets | launch order | p base | avg | status(0)] and one download arena [f | p],
//    mirrored in pinned memory for small passes: one H2D copy, the kernels, one D2H copy, one sync.
int run_requests(ThreadCtx *c, HostReq **reqs, int n_req)
{
	int rc;
	const mm2c_params_t *par = reqs[0]->par;
	const uint64_t D = (uint64_t)(int64_t)par->max_dist_x;
	int64_t total = 0, n_tasks_all = 0;
	for (int r = 0; r < n_req; ++r) { total += reqs[r]->off[reqs[r]->n_tasks] - reqs[r]->off[0]; n_tasks_all += reqs[r]->n_tasks; }
	if (total == 0) return 0;
	// cutting costs a host pass over the anchors: only worth it when the pass has too few tasks to fill the GPU on its own
	const int64_t seg_min = n_tasks_all < G.cut_below_tasks ? G.seg_min : 0;
	// uncut tasks without a caller-supplied avg_qspan_scaled: the kernel sums the spans itself (chain.c:48-49), no host pass
	bool kernel_avg = seg_min == 0;
	for (int r = 0; r < n_req; ++r) if (reqs[r]->avg) kernel_avg = false;
	std::vector<int64_t> seg_off; std::vector<int32_t> pbase, order; std::vector<float> seg_avg;
	seg_off.reserve((size_t)n_tasks_all + 16); pbase.reserve((size_t)n_tasks_all + 16); seg_avg.reserve((size_t)n_tasks_all + 16);
	int64_t g0 = 0;                                                // where this request's anchors start in the arena
	for (int r = 0; r < n_req; ++r) {
		const HostReq &q = *reqs[r];
		const int64_t base = q.off[0];
		const mm2c_anchor_t *a = q.a + base;
		for (int64_t k = 0; k < q.n_tasks; ++k) {
			const int64_t t0 = q.off[k] - base, t1 = q.off[k + 1] - base;
			if (t1 == t0) continue;
			float avg = 0.f;
			if (q.avg) avg = q.avg[k];
			else if (!kernel_avg) {
				uint64_t sum = 0;
				for (int64_t i = t0; i < t1; ++i) sum += a[i].y >> 32 & 0xff;
				avg = (float)(.01 * (float)sum / (t1 - t0));
			}
			int64_t s0 = t0;
			for (int64_t i = t0 + 1; i < t1; ++i)
				if (seg_min > 0 && i - s0 >= seg_min && a[i].x > a[i - 1].x + D) {
					seg_off.push_back(g0 + s0); pbase.push_back((int32_t)(s0 - t0)); seg_avg.push_back(avg);
					s0 = i;
				}
			seg_off.push_back(g0 + s0); pbase.push_back((int32_t)(s0 - t0)); seg_avg.push_back(avg);
		}
		g0 += q.off[q.n_tasks] - base;
	}
	const int64_t n_seg = (int64_t)seg_off.size();
	seg_off.push_back(total);
	if ((rc = build_order(n_seg, seg_off.data(), order))) return rc;

	HIP_TRY(hipSetDevice(G.device));
	const size_t o_a = 0, o_off = align16((size_t)total * 16), o_ord = align16(o_off + ((size_t)n_seg + 1) * 8),
	             o_pb = align16(o_ord + (size_t)n_seg * 4), o_avg = align16(o_pb + (size_t)n_seg * 4),
	             o_stat = align16(o_avg + (size_t)n_seg * 4), in_bytes = align16(o_stat + (size_t)n_seg * 4);
	const size_t meta_bytes = in_bytes - o_off;
	const bool staged = (size_t)total <= G.stage_max_anchors;      // small passes go through pinned staging, big ones copy in place
	if ((rc = grow_device(&c->d_in, &c->cap_in, in_bytes))) return rc;
	if ((rc = grow_device(&c->d_out, &c->cap_out, (size_t)total * 8))) return rc;
	if ((rc = grow_device(&c->d_scratch, &c->cap_scratch, (size_t)total * 8))) return rc;
	if ((rc = grow_pinned(&c->h_in, &c->cap_hin, staged ? in_bytes : meta_bytes))) return rc;
	char *hm = staged ? c->h_in + o_off : c->h_in;                 // where the metadata block starts in the staging buffer
	memcpy(hm, seg_off.data(), ((size_t)n_seg + 1) * 8);
	memcpy(hm + (o_ord - o_off), order.data(), (size_t)n_seg * 4);
	memcpy(hm + (o_pb - o_off), pbase.data(), (size_t)n_seg * 4);
	memcpy(hm + (o_avg - o_off), seg_avg.data(), (size_t)n_seg * 4);
	memset(hm + (o_stat - o_off), 0, in_bytes - o_stat);
	if (staged) {
		size_t at = o_a;
		for (int r = 0; r < n_req; ++r) {
			const size_t nb = (size_t)(reqs[r]->off[reqs[r]->n_tasks] - reqs[r]->off[0]) * 16;
			memcpy(c->h_in + at, reqs[r]->a + reqs[r]->off[0], nb);
			at += nb;
		}
		HIP_TRY(hipMemcpyAsync(c->d_in, c->h_in, in_bytes, hipMemcpyHostToDevice, c->st));                  // cf. chain_hardware.cpp:110,114
	} else if (n_req == 1 && total >= 2 * G.pipeline_chunk_anchors) {
		// big batch: pipeline it in chunks of whole pieces on two streams, so that the upload of chunk k+1, the kernels of chunk k
		// and the download of chunk k-1 overlap (PCIe is full duplex); with page-locked caller buffers this runs at PCIe rate
		if (!c->st2) HIP_TRY(create_partner_stream(&c->st2));
		if (!c->ev) HIP_TRY(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
		HIP_TRY(hipMemcpyAsync(c->d_in + o_off, hm, meta_bytes, hipMemcpyHostToDevice, c->st));
		HIP_TRY(hipEventRecord(c->ev, c->st));
		HIP_TRY(hipStreamWaitEvent(c->st2, c->ev, 0));
		const HostReq &q = *reqs[0];
		const mm2c_anchor_t *src = q.a + q.off[0];
		int32_t *dst_f = q.f + q.off[0], *dst_p = q.p + q.off[0];
		const int64_t chunk_anchors = G.pipeline_chunk_anchors;      // big enough for one chunk to fill the GPU on its own
		int nl = 0, k = 0;
		for (int64_t s0 = 0; s0 < n_seg; ++k) {
			int64_t s1 = s0 + 1;
			while (s1 < n_seg && seg_off[(size_t)s1 + 1] - seg_off[(size_t)s0] <= chunk_anchors) ++s1;
			const int64_t a0 = seg_off[(size_t)s0], a1 = seg_off[(size_t)s1];
			hipStream_t st = (k & 1) ? c->st2 : c->st;
			HIP_TRY(hipMemcpyAsync(c->d_in + o_a + (size_t)a0 * 16, src + a0, (size_t)(a1 - a0) * 16, hipMemcpyHostToDevice, st));
			mm2c::LaunchArgs L;
			L.P = to_kparams(par);
			L.n_tasks = s1 - s0; L.d_offsets = (const int64_t *)(c->d_in + o_off) + s0; L.d_order = nullptr;
			L.d_anchors = c->d_in + o_a; L.d_avg = kernel_avg ? nullptr : (const float *)(c->d_in + o_avg) + s0;
			L.d_pbase = (const int32_t *)(c->d_in + o_pb) + s0; L.d_status = (int32_t *)(c->d_in + o_stat) + s0;
			L.d_f = (int32_t *)c->d_out; L.d_p = (int32_t *)(c->d_out + (size_t)total * 4);
			L.d_t = (int32_t *)c->d_scratch; L.d_st = (int32_t *)(c->d_scratch + (size_t)total * 4);
			L.ring_class = G.ring_class;
			HIP_TRY(mm2c::launch_chain_dp(L, st, &nl, nullptr));
			HIP_TRY(hipMemcpyAsync(dst_f + a0, L.d_f + a0, (size_t)(a1 - a0) * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(dst_p + a0, L.d_p + a0, (size_t)(a1 - a0) * 4, hipMemcpyDeviceToHost, st));
			s0 = s1;
		}
		HIP_TRY(hipStreamSynchronize(c->st));
		HIP_TRY(hipStreamSynchronize(c->st2));
		G.tasks += (uint64_t)n_tasks_all; G.anchors += (uint64_t)total; G.launches += (uint64_t)nl; G.segments += (uint64_t)n_seg;
		G.passes += 1;
		return 0;
	} else {
		HIP_TRY(hipMemcpyAsync(c->d_in + o_off, hm, meta_bytes, hipMemcpyHostToDevice, c->st));
		size_t at = o_a;
		for (int r = 0; r < n_req; ++r) {
			const size_t nb = (size_t)(reqs[r]->off[reqs[r]->n_tasks] - reqs[r]->off[0]) * 16;
			HIP_TRY(hipMemcpyAsync(c->d_in + at, reqs[r]->a + reqs[r]->off[0], nb, hipMemcpyHostToDevice, c->st));
			at += nb;
		}
	}
	mm2c::LaunchArgs L;
	L.P = to_kparams(par);
	L.n_tasks = n_seg; L.d_offsets = (const int64_t *)(c->d_in + o_off); L.d_order = (const int32_t *)(c->d_in + o_ord);
	L.d_anchors = c->d_in + o_a; L.d_avg = kernel_avg ? nullptr : (const float *)(c->d_in + o_avg); L.d_pbase = (const int32_t *)(c->d_in + o_pb);
	L.d_status = (int32_t *)(c->d_in + o_stat);
	L.d_f = (int32_t *)c->d_out; L.d_p = (int32_t *)(c->d_out + (size_t)total * 4);
	L.d_t = (int32_t *)c->d_scratch; L.d_st = (int32_t *)(c->d_scratch + (size_t)total * 4);
	L.ring_class = G.ring_class;
	int nl = 0;
	HIP_TRY(mm2c::launch_chain_dp(L, c->st, &nl, nullptr));                                                          // cf. chain_hardware.cpp:156
	if (staged) {
		if ((rc = grow_pinned(&c->h_out, &c->cap_hout, (size_t)total * 8))) return rc;
		HIP_TRY(hipMemcpyAsync(c->h_out, c->d_out, (size_t)total * 8, hipMemcpyDeviceToHost, c->st));           // cf. chain_hardware.cpp:167,170
		HIP_TRY(hipStreamSynchronize(c->st));                                                               // cf. chain_hardware.cpp:175
		size_t at = 0;
		for (int r = 0; r < n_req; ++r) {
			const size_t n = (size_t)(reqs[r]->off[reqs[r]->n_tasks] - reqs[r]->off[0]);
			memcpy(reqs[r]->f + reqs[r]->off[0], c->h_out + at * 4, n * 4);
			memcpy(reqs[r]->p + reqs[r]->off[0], c->h_out + (size_t)total * 4 + at * 4, n * 4);
			at += n;
		}
	} else {
		size_t at = 0;
		for (int r = 0; r < n_req; ++r) {
			const size_t n = (size_t)(reqs[r]->off[reqs[r]->n_tasks] - reqs[r]->off[0]);
			HIP_TRY(hipMemcpyAsync(reqs[r]->f + reqs[r]->off[0], L.d_f + at, n * 4, hipMemcpyDeviceToHost, c->st));
			HIP_TRY(hipMemcpyAsync(reqs[r]->p + reqs[r]->off[0], L.d_p + at, n * 4, hipMemcpyDeviceToHost, c->st));
			at += n;
		}
		HIP_TRY(hipStreamSynchronize(c->st));
	}
	G.tasks += (uint64_t)n_tasks_all; G.anchors += (uint64_t)total; G.launches += (uint64_t)nl; G.segments += (uint64_t)n_seg;
	G.passes += 1;
	return 0;
}

// Combiner for small synchronous calls (the reference's call pattern: up to n_threads host threads, each blocking in
// run_chaining_on_hw / mm_chain_dp, map.c:561).  A caller that finds no pass in flight becomes the leader: it takes every
// pending request with the same scalars, runs ONE GPU pass for all of them and wakes their owners; callers that arrive
// meanwhile queue up and are served by the next leader.  (The reference instead serialises callers on a mutex and a FIFO,
// chain_hardware.cpp:54-93.)  Big requests skip the combiner and run on the caller's own stream.
struct Combiner {
	std::mutex mu;
	std::condition_variable cv;
	std::vector<HostReq *> pending;
	bool leader_active = false;
	ThreadCtx ctx;                      // stream + arenas of the pass in flight (leader-exclusive)
	uint64_t epoch = ~0ull;
} CB;

void release_combiner()
{
	std::lock_guard<std::mutex> lk(CB.mu);
	CB.ctx.release();
	CB.epoch = ~0ull;
}

int submit_combined(HostReq *me)
{
	std::unique_lock<std::mutex> lk(CB.mu);
	CB.pending.push_back(me);
	for (;;) {
		if (me->done) return me->rc;
		if (!CB.leader_active) break;
		CB.cv.wait(lk);
	}
	// leader: collect the pending requests that share my scalars, up to the staging size
	CB.leader_active = true;
	std::vector<HostReq *> batch, rest;
	size_t tot = 0;
	for (HostReq *q : CB.pending) {
		const size_t n = (size_t)(q->off[q->n_tasks] - q->off[0]);
		if ((q == me || (memcmp(q->par, me->par, sizeof(mm2c_params_t)) == 0 && tot + n <= G.stage_max_anchors)) ) { batch.push_back(q); tot += n; }
		else rest.push_back(q);
	}
	CB.pending.swap(rest);
	lk.unlock();
	int rc = 0;
	{
		std::lock_guard<std::mutex> gl(G.mu);
		if (!G.ready) rc = fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device");
		else if (CB.epoch != G.epoch) {                           // first pass after (re)initialisation: fresh stream and arenas
			CB.ctx = ThreadCtx();
			hipError_t e = hipSetDevice(G.device);
			if (e == hipSuccess) e = hipStreamCreateWithFlags(&CB.ctx.st, hipStreamNonBlocking);
			if (e != hipSuccess) rc = fail(MM2C_E_HIP, "combiner stream: %s", hipGetErrorString(e));
			else CB.epoch = G.epoch;
		}
	}
	if (rc == 0) rc = run_requests(&CB.ctx, batch.data(), (int)batch.size());
	lk.lock();
	for (HostReq *q : batch) {
		q->rc = rc; q->done = true;
		if (rc != 0) { strncpy(q->err, g_err, sizeof(q->err) - 1); q->err[sizeof(q->err) - 1] = 0; }
	}
	CB.leader_active = false;
	CB.cv.notify_all();
	return me->rc;
}

} // namespace

extern "C" {

int mm2c_chain_batch_host(const mm2c_params_t *par, int64_t n_tasks, const int64_t *h_offsets, const mm2c_anchor_t *h_anchors,
                          const float *h_avg_qspan, int32_t *h_f, int32_t *h_p)
{
	int rc;
	const auto t_begin = std::chrono::steady_clock::now();
	if ((rc = check_params(par))) return rc;
	std::vector<int32_t> order;
	if ((rc = build_order(n_tasks, h_offsets, order))) return rc;      // validates the offsets
	if (n_tasks == 0) return 0;
	const int64_t total = h_offsets[n_tasks] - h_offsets[0];
	if (total == 0) return 0;
	if (!h_anchors || !h_f || !h_p) return fail(MM2C_E_ARG, "host pointer is NULL");
	HostReq req;
	req.par = par; req.n_tasks = n_tasks; req.off = h_offsets; req.a = h_anchors; req.avg = h_avg_qspan; req.f = h_f; req.p = h_p;
	req.err[0] = 0;
	if ((size_t)total <= G.combine_max_anchors) {
		rc = submit_combined(&req);
		if (rc != 0 && req.err[0]) fail(rc, "%s", req.err);
	} else {
		ThreadCtx *c;
		if ((rc = get_thread_ctx(&c))) return rc;
		HostReq *one = &req;
		rc = run_requests(c, &one, 1);
	}
	G.host_call_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
	return rc;
}

int mm2c_mm_chain_dp_batch_host(const mm2c_params_t *par, int min_cnt, int min_sc, int64_t n_tasks, const int64_t *h_offsets,
                                const mm2c_anchor_t *h_anchors, int epilogue_threads, int64_t *u_off, uint64_t *u, int64_t *b_off,
                                mm2c_anchor_t *b)
{
	int rc;
	if ((rc = check_params(par))) return rc;
	if (n_tasks < 0 || !u_off || !b_off) return fail(MM2C_E_ARG, "bad argument");
	u_off[0] = b_off[0] = 0;
	if (n_tasks == 0) return 0;
	if (!h_offsets) return fail(MM2C_E_ARG, "offsets is NULL");
	const int64_t total = h_offsets[n_tasks] - h_offsets[0];
	if (total > 0 && (!h_anchors || !u || !b)) return fail(MM2C_E_ARG, "host pointer is NULL");
	if (epilogue_threads > 0) {
		std::vector<int32_t> f((size_t)std::max<int64_t>(total, 1)), p((size_t)std::max<int64_t>(total, 1));
		if ((rc = mm2c_chain_batch_host(par, n_tasks, h_offsets, h_anchors, nullptr, f.data() - h_offsets[0], p.data() - h_offsets[0]))) return rc;
		rc = mm2c_chain_epilogue_host(min_cnt, min_sc, n_tasks, h_offsets, h_anchors, f.data() - h_offsets[0], p.data() - h_offsets[0],
		                              epilogue_threads, u_off, u, b_off, b);
		return rc ? fail(rc, "mm2c_chain_epilogue_host failed") : 0;
	}
	if (total == 0) { for (int64_t k = 1; k <= n_tasks; ++k) u_off[k] = b_off[k] = 0; return 0; }
	// everything on the GPU: anchors up, DP, epilogue, chains down; big batches in chunks of whole tasks on two streams, so that the
	// upload of chunk k+1, the kernels of chunk k and the download of chunk k-1 overlap
	if (total >= (int64_t)INT32_MAX && G.pipeline_chunk_anchors >= (int64_t)INT32_MAX) return fail(MM2C_E_TOOBIG, "batch too big for one chunk");
	ThreadCtx *c;
	if ((rc = get_thread_ctx(&c))) return rc;
	HIP_TRY(hipSetDevice(G.device));
	const int64_t chunk_anchors = total >= 2 * G.pipeline_chunk_anchors ? G.pipeline_chunk_anchors : total;
	const mm2c_anchor_t *a0 = h_anchors + h_offsets[0];
	int64_t base_u = 0, base_b = 0;
	int nl = 0;

	auto enqueue = [&](WholeSlot &w, int64_t k0, int64_t k1) -> int {
		int r;
		const size_t nt = (size_t)(k1 - k0), tot = (size_t)(h_offsets[k1] - h_offsets[k0]);
		if (!w.st) HIP_TRY(&w == &c->whole[1] ? create_partner_stream(&w.st) : hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking));
		w.k0 = k0; w.k1 = k1; w.busy = true;
		// upload arena: [anchors | offsets | order | status]; pinned mirror of the metadata + room for the offsets that come back
		const size_t o_off = align16(tot * 16), o_ord = align16(o_off + (nt + 1) * 8), o_stat = align16(o_ord + nt * 4), in_bytes = align16(o_stat + nt * 4);
		const size_t meta_bytes = in_bytes - o_off;
		w.o_hres = align16(meta_bytes);
		if ((r = grow_device(&w.d_in, &w.cap_in, in_bytes))) return r;
		if ((r = grow_pinned(&w.h_meta, &w.cap_hmeta, w.o_hres + 2 * (nt + 1) * 8))) return r;
		std::vector<int32_t> order;
		if ((r = build_order((int64_t)nt, h_offsets + k0, order))) return r;
		int64_t *m_off = (int64_t *)w.h_meta;
		for (size_t k = 0; k <= nt; ++k) m_off[k] = h_offsets[k0 + (int64_t)k] - h_offsets[k0];
		memcpy(w.h_meta + (o_ord - o_off), order.data(), nt * 4);
		memset(w.h_meta + (o_stat - o_off), 0, in_bytes - o_stat);
		// work arena: [f | p | t | st | epilogue scratch]; result arena: [u_off | b_off | u | b]
		mm2c::EpiArgs E;
		const size_t tmp = mm2c::epilogue_sort_temp_bytes((int64_t)tot, (int64_t)nt);
		const size_t o_epi = align16(tot * 16), work_bytes = o_epi + layout_epilogue(E, nullptr, tot, nt, tmp);
		if ((r = grow_device(&w.d_work, &w.cap_work, work_bytes))) return r;
		layout_epilogue(E, w.d_work + o_epi, tot, nt, tmp);
		const size_t o_boff = align16((nt + 1) * 8);
		w.o_res_u = align16(o_boff + (nt + 1) * 8); w.o_res_b = align16(w.o_res_u + tot * 8);
		if ((r = grow_device(&w.d_res, &w.cap_res, w.o_res_b + tot * 16))) return r;
		if (tot == 0) { memset(w.h_meta + w.o_hres, 0, 2 * (nt + 1) * 8); return 0; }
		HIP_TRY(hipMemcpyAsync(w.d_in + o_off, w.h_meta, meta_bytes, hipMemcpyHostToDevice, w.st));
		HIP_TRY(hipMemcpyAsync(w.d_in, a0 + (h_offsets[k0] - h_offsets[0]), tot * 16, hipMemcpyHostToDevice, w.st));
		int32_t *d_f = (int32_t *)w.d_work, *d_p = d_f + tot;
		mm2c::LaunchArgs L;
		L.P = to_kparams(par);
		L.n_tasks = (int64_t)nt; L.d_offsets = (const int64_t *)(w.d_in + o_off); L.d_order = (const int32_t *)(w.d_in + o_ord);
		L.d_anchors = w.d_in; L.d_avg = nullptr; L.d_pbase = nullptr; L.d_status = (int32_t *)(w.d_in + o_stat);
		L.d_f = d_f; L.d_p = d_p; L.d_t = d_p + tot; L.d_st = d_p + 2 * tot;
		L.ring_class = G.ring_class;
		HIP_TRY(mm2c::launch_chain_dp(L, w.st, &nl, nullptr));
		E.n_tasks = (int64_t)nt; E.total = (int64_t)tot; E.d_off = L.d_offsets; E.d_order = L.d_order;
		E.d_a = (const ulonglong2 *)w.d_in; E.d_f = d_f; E.d_p = d_p; E.min_cnt = min_cnt; E.min_sc = min_sc;
		E.debug_phases = epilogue_debug_phases();
		E.u_off = (int64_t *)w.d_res; E.b_off = (int64_t *)(w.d_res + o_boff);
		E.u_out = (uint64_t *)(w.d_res + w.o_res_u); E.b_out = (ulonglong2 *)(w.d_res + w.o_res_b);
		HIP_TRY(mm2c::launch_chain_epilogue(E, w.st, &nl));
		HIP_TRY(hipMemcpyAsync(w.h_meta + w.o_hres, E.u_off, (nt + 1) * 8, hipMemcpyDeviceToHost, w.st));
		HIP_TRY(hipMemcpyAsync(w.h_meta + w.o_hres + (nt + 1) * 8, E.b_off, (nt + 1) * 8, hipMemcpyDeviceToHost, w.st));
		return 0;
	};
	// waits for the chunk's offsets, places them behind the chunks before it and starts the download of its chains
	auto finalize = [&](WholeSlot &w) -> int {
		if (!w.busy) return 0;
		w.busy = false;
		const size_t nt = (size_t)(w.k1 - w.k0);
		HIP_TRY(hipStreamSynchronize(w.st));
		const int64_t *cu = (const int64_t *)(w.h_meta + w.o_hres), *cb = cu + nt + 1;
		for (size_t k = 1; k <= nt; ++k) { u_off[w.k0 + (int64_t)k] = base_u + cu[k]; b_off[w.k0 + (int64_t)k] = base_b + cb[k]; }
		if (cu[nt] > 0) HIP_TRY(hipMemcpyAsync(u + base_u, w.d_res + w.o_res_u, (size_t)cu[nt] * 8, hipMemcpyDeviceToHost, w.st));
		if (cb[nt] > 0) HIP_TRY(hipMemcpyAsync(b + base_b, w.d_res + w.o_res_b, (size_t)cb[nt] * 16, hipMemcpyDeviceToHost, w.st));
		base_u += cu[nt]; base_b += cb[nt];
		return 0;
	};
	int n_chunks = 0;
	for (int64_t k0 = 0; k0 < n_tasks && rc == 0; ++n_chunks) {
		int64_t k1 = k0 + 1;
		while (k1 < n_tasks && h_offsets[k1 + 1] - h_offsets[k0] <= chunk_anchors) ++k1;
		if (h_offsets[k1] - h_offsets[k0] >= (int64_t)INT32_MAX) { rc = fail(MM2C_E_TOOBIG, "a chunk of the batch has 2^31 anchors or more"); break; }
		WholeSlot &w = c->whole[n_chunks & 1];
		if ((rc = finalize(w))) break;                              // the chunk before the previous one (same slot)
		rc = enqueue(w, k0, k1);
		k0 = k1;
	}
	if (rc == 0) rc = finalize(c->whole[n_chunks & 1]);             // in chunk order: the older slot first
	if (rc == 0) rc = finalize(c->whole[(n_chunks + 1) & 1]);
	for (int i = 0; i < 2; ++i) {
		if (c->whole[i].st) { hipError_t e = hipStreamSynchronize(c->whole[i].st); if (e != hipSuccess && rc == 0) rc = fail(MM2C_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e)); }
		c->whole[i].busy = false;
	}
	G.tasks += (uint64_t)n_tasks; G.anchors += (uint64_t)total; G.launches += (uint64_t)nl; G.passes += (uint64_t)n_chunks;
	return rc;
}

void *mm2c_pinned_alloc(size_t bytes)
{
	void *p = nullptr;
	if (!G.ready) { fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device"); return nullptr; }
	if (hipSetDevice(G.device) != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
		fail(MM2C_E_HIP, "hipHostMalloc(%zu) failed", bytes);
		return nullptr;
	}
	return p;
}

void mm2c_pinned_free(void *ptr) { if (ptr) (void)hipHostFree(ptr); }

int mm2c_chain_task_host(const mm2c_params_t *par, int64_t n, const mm2c_anchor_t *a, float avg_qspan_scaled,
                         int32_t *f, int32_t *p, int tid)
{
	(void)tid;  // the reference uses tid for its FIFO (chain_hardware.cpp:65,83); each host thread owns a stream here
	if (n == 0) return 0;                                                                                   // chain_hardware.cpp:30-32
	if (n < 0) return fail(MM2C_E_ARG, "n < 0");
	const int64_t off[2] = { 0, n };
	return mm2c_chain_batch_host(par, 1, off, a, &avg_qspan_scaled, f, p);
}

} // extern "C"
