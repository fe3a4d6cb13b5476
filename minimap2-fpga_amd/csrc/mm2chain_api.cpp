// mm2chain_api.cpp -- the C-ABI shim over HIP (include/mm2chain.h).
//
// Replaces the reference's XRT/OpenCL enqueue path: hardware_init (chain_hardware.cpp:278-400: platform, xclbin,
// kernel object, four cl_mem buffers), the body of run_chaining_on_hw (chain_hardware.cpp:104-189: two
// clEnqueueWriteBuffer, clEnqueueTask, two clEnqueueReadBuffer, clFinish) and cleanup (chain_hardware.cpp:403-441).
// Differences by design: no busy/queue time model and no "declined, do it on the CPU" return (chain_hardware.cpp:54-93)
// -- every accepted call is computed on the GPU; many tasks are in flight at once (one wave each) instead of one
// task per kernel; callers on different host threads get their own stream and staging buffers instead of a mutex.
#include "api_internal.h"
#include "mm2chain_split.h"
#include <string>
#include <cctype>
#include <pthread.h>
#include <sched.h>

namespace mm2c_api {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

Global G;
StageStats SS;

// ---- mm2c_init_async: initialisation beside the host's own start-up work
static std::mutex g_async_mu;
static std::thread *g_async_th = nullptr;         // (never destroyed: a host that exits without mm2c_shutdown must not meet std::terminate in a static destructor)
static std::atomic<bool> g_async_pending{false};
static char g_async_err[512] = "";
static thread_local bool tl_is_init_thread = false;   // set on the initialisation thread itself: its own calls into the library must not wait for it

static void async_join_locked()                    // g_async_mu held
{
	if (g_async_th && g_async_th->joinable()) g_async_th->join();
	g_async_pending.store(false, std::memory_order_release);
}

void async_init_join()
{
	if (tl_is_init_thread || !g_async_pending.load(std::memory_order_acquire)) return;
	std::lock_guard<std::mutex> lk(g_async_mu);
	async_join_locked();
}

int fail_not_ready()
{
	if (g_async_err[0]) return fail(MM2C_E_NODEVICE, "the asynchronous initialisation (mm2c_init_async) failed: %s", g_async_err);
	return fail(MM2C_E_NODEVICE, "mm2c_init() has not been called or found no HIP device");
}

// Device memory of plans and one-shot calls goes through a small cache: a batched caller creates and destroys plans of similar size
// for every mini-batch, and hipMalloc / hipFree of gigabytes cost milliseconds each (hipFree also waits for the device).  A freed
// block is kept and handed to the next request it fits (the smallest cached block of 1x .. 3x the size: the last chunk of a pipelined batch is
// smaller than the others and must not cost an allocation); the cache is bounded and emptied by mm2c_shutdown.
struct DevCache {
	std::mutex mu;
	struct Block { void *p; size_t size; };
	std::vector<Block> free_blocks;                 // cached, not in use
	std::vector<Block> live;                        // handed out (to know their size on free)
	size_t cached_bytes = 0;
	static constexpr size_t MAX_CACHED = (size_t)16 << 30;   // HBM the cache may keep out of sight of other allocators (mm2c_tune("trim", 0) returns it)
};
static DevCache g_caches[64];                      // one per device ordinal: a block is only ever handed back to the device it came from
static DevCache &dev_cache()
{
	int d = 0;
	if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
	return g_caches[d];
}
#define DC dev_cache()

hipError_t dev_alloc(void **out, size_t bytes)
{
	if (bytes == 0) bytes = 1;
	{
		std::lock_guard<std::mutex> lk(DC.mu);
		size_t best = (size_t)-1;
		for (size_t i = 0; i < DC.free_blocks.size(); ++i) {
			const size_t sz = DC.free_blocks[i].size;
			if (sz >= bytes && sz <= 3 * bytes + (1u << 20) && (best == (size_t)-1 || sz < DC.free_blocks[best].size)) best = i;
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
	ScopedNs timed(SS.alloc_ns); ++SS.n_alloc;
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

// the cache that handed out p (blocks are parked in the cache of the device they were allocated on, whatever device is current now)
static int owner_of(void *p, DevCache::Block *b)
{
	for (int d = 0; d < 64; ++d) {
		DevCache &c = g_caches[d];
		std::lock_guard<std::mutex> lk(c.mu);
		for (size_t i = 0; i < c.live.size(); ++i)
			if (c.live[i].p == p) { *b = c.live[i]; c.live.erase(c.live.begin() + (long)i); return d; }
	}
	return -1;
}

static void park_or_release(int d, DevCache::Block b)
{
	bool park = false;
	if (d >= 0) {
		DevCache &c = g_caches[d];
		std::lock_guard<std::mutex> lk(c.mu);
		if (b.size != 0 && c.cached_bytes + b.size <= DevCache::MAX_CACHED && c.free_blocks.size() < 64) {
			c.free_blocks.push_back(b); c.cached_bytes += b.size; park = true;
		}
	}
	if (!park) { ScopedNs timed(SS.free_ns); ++SS.n_free; (void)hipFree(b.p); }
}

void dev_free_synced(void *p)
{
	if (!p) return;
	DevCache::Block b{p, 0};
	park_or_release(owner_of(p, &b), b);
}

void dev_free(void *p)
{
	if (!p) return;
	DevCache::Block b{p, 0};
	const int d = owner_of(p, &b);
	{
		// like hipFree, this waits for the block's device: a parked block is handed to the next caller at once, so nothing may still be in flight on it
		DeviceScope on(d >= 0 ? d : cur_device());
		ScopedNs timed(SS.free_ns);
		(void)hipDeviceSynchronize();
	}
	park_or_release(d, b);
}

void dev_cache_release()
{
	for (DevCache &c : g_caches) {
		std::vector<DevCache::Block> drop;
		{ std::lock_guard<std::mutex> lk(c.mu); drop.swap(c.free_blocks); c.cached_bytes = 0; }   // blocks still handed out stay known (their size is needed when they come back)
		for (auto &b : drop) (void)hipFree(b.p);
	}
}

thread_local ThreadCtx *tl_ctx = nullptr;
thread_local uint64_t tl_epoch = 0;
thread_local int tl_slot = -1;                     // >= 0: this thread is the worker of a split batch and drives G.devices[tl_slot]

// per device slot: the context the worker of a split batch uses (persistent, so that its arenas are allocated once); concurrent split batches
// take turns on it through the slot's one worker thread
static ThreadCtx g_slot_ctx[64];
static uint64_t g_slot_epoch[64];

int n_devices() { return (int)G.devices.size(); }
bool in_split_worker() { return tl_slot >= 0; }
int cur_device() { return tl_slot >= 0 && tl_slot < (int)G.devices.size() ? G.devices[(size_t)tl_slot] : G.device; }
bool should_split(int64_t total_anchors) { return tl_slot < 0 && G.devices.size() > 1 && total_anchors >= G.multi_min_anchors; }

// ---- the workers of split batches: ONE PERSISTENT THREAD PER DEVICE SLOT (round 5; rounds 2-4 started a std::thread per call), pinned to the CPUs of the NUMA node its
// device hangs off -- the page-locked staging buffers of the slot's context are allocated by this thread (first touch on that node), and the copies it drives cross no
// socket link.  The node comes from sysfs: /sys/bus/pci/devices/<pci bus id>/numa_node, its CPUs from /sys/devices/system/node/node<N>/cpulist, intersected with the
// CPUs the process may use; no NUMA information (numa_node -1, a container without the files) leaves the thread where the scheduler puts it.
struct SplitJob { std::function<int()> fn; int rc = 0; std::string err; bool done = false; };
struct SlotWorker {
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::vector<SplitJob *> queue;           // jobs of concurrent split batches take turns (what g_slot_mu did for the per-call threads)
	bool quit = false, started = false;
	int pinned_node = -1;                    // NUMA node the thread was pinned to, -1: not pinned
};
// Never destroyed (advisor, round 5): a host that leaves through exit() -- the drop-in's own exit(EXIT_FAILURE), minimap2's exit(1), a Python interpreter that ends
// without mm2c_shutdown -- would otherwise run ~thread() on a joinable thread (std::terminate -> SIGABRT instead of its exit code) and destroy a mutex and a
// condition variable that a worker still waits on.  g_async_th above is kept the same way.
static SlotWorker *const g_workers = new SlotWorker[64];

static bool parse_cpulist(const char *txt, cpu_set_t *out)
{
	CPU_ZERO(out);
	bool any = false;
	for (const char *c = txt; *c && *c != '\n';) {
		char *end = nullptr;
		const long lo = strtol(c, &end, 10);
		if (end == c || lo < 0) return false;
		long hi = lo;
		if (*end == '-') { const char *h = end + 1; hi = strtol(h, &end, 10); if (end == h || hi < lo) return false; }
		for (long k = lo; k <= hi && k < CPU_SETSIZE; ++k) { CPU_SET((int)k, out); any = true; }
		c = (*end == ',') ? end + 1 : end;
		if (*end != ',' && *end != 0 && *end != '\n') return false;
	}
	return any;
}

static void pin_to_device_node(int device, SlotWorker &w)
{
	if (!G.pin_workers.load()) return;
	char bus[64] = "";
	if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) return;
	for (char *c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
	const char *root = getenv("MM2C_SYSFS_ROOT");
	char list[4096];
	const int node = mm2c_numa_cpulist(root && *root ? root : "/sys", bus, list, sizeof(list));
	if (node < 0) return;
	cpu_set_t want, have, both;
	if (!parse_cpulist(list, &want) || sched_getaffinity(0, sizeof(have), &have) != 0) return;
	CPU_AND(&both, &want, &have);
	if (CPU_COUNT(&both) == 0) return;
	if (pthread_setaffinity_np(pthread_self(), sizeof(both), &both) == 0) w.pinned_node = node;
}

static void slot_worker_main(int s)
{
	SlotWorker &w = g_workers[s];
	tl_slot = s;
	{
		int dev = -1;
		{ std::lock_guard<std::mutex> lk(G.mu); if (s < (int)G.devices.size()) dev = G.devices[(size_t)s]; }
		if (dev >= 0) pin_to_device_node(dev, w);
	}
	std::unique_lock<std::mutex> lk(w.mu);
	for (;;) {
		w.cv.wait(lk, [&] { return w.quit || !w.queue.empty(); });
		if (w.queue.empty()) { if (w.quit) return; continue; }
		SplitJob *j = w.queue.front();
		w.queue.erase(w.queue.begin());
		lk.unlock();
		try { j->rc = j->fn(); }
		catch (...) { j->rc = fail(MM2C_E_ARG, "out of host memory in the worker of a split batch"); }
		if (j->rc != 0) j->err = g_err;
		lk.lock();
		j->done = true;
		w.cv.notify_all();
	}
}

// ends the workers (mm2c_shutdown, before it takes the library's lock: a worker's job may need it)
static void stop_slot_workers()
{
	for (int s = 0; s < 64; ++s) {
		SlotWorker &w = g_workers[s];
		{
			std::lock_guard<std::mutex> lk(w.mu);
			if (!w.started) continue;
			w.quit = true;
		}
		w.cv.notify_all();
		if (w.th.joinable()) w.th.join();
		std::lock_guard<std::mutex> lk(w.mu);
		w.started = false; w.quit = false; w.pinned_node = -1;
	}
}

int run_split(int64_t n_tasks, const int64_t *h_offsets, const std::function<int(int, int64_t, int64_t)> &fn)
{
	const int nd = (int)std::min<size_t>(G.devices.size(), 64);
	std::vector<int64_t> bounds((size_t)nd + 1);
	if (mm2c_split_tasks(n_tasks, h_offsets, nd, bounds.data()) != 0) return MM2C_E_ARG;
	std::vector<SplitJob> jobs((size_t)nd);
	std::vector<int> posted;
	bool spawn_failed = false;
	for (int s = 0; s < nd && !spawn_failed; ++s) {
		if (bounds[(size_t)s] == bounds[(size_t)s + 1]) continue;
		SlotWorker &w = g_workers[s];
		const int64_t k0 = bounds[(size_t)s], k1 = bounds[(size_t)s + 1];
		jobs[(size_t)s].fn = [&fn, s, k0, k1]() { return fn(s, k0, k1); };
		std::lock_guard<std::mutex> lk(w.mu);
		if (!w.started) {
			try { w.th = std::thread(slot_worker_main, s); w.started = true; }
			catch (...) { spawn_failed = true; break; }          // std::system_error from thread creation: the jobs already posted are waited for below
		}
		w.queue.push_back(&jobs[(size_t)s]);
		posted.push_back(s);
		w.cv.notify_all();
	}
	for (int s : posted) {
		SlotWorker &w = g_workers[s];
		std::unique_lock<std::mutex> lk(w.mu);
		w.cv.wait(lk, [&] { return jobs[(size_t)s].done; });
	}
	if (spawn_failed) return fail(MM2C_E_ARG, "could not start the worker thread of a device slot");
	for (int s : posted) if (jobs[(size_t)s].rc != 0) return fail(jobs[(size_t)s].rc, "device %d: %s", G.devices[(size_t)s], jobs[(size_t)s].err.c_str());
	return 0;
}

int get_thread_ctx(ThreadCtx **out)
{
	async_init_join();
	std::lock_guard<std::mutex> lk(G.mu);
	if (!G.ready) return fail_not_ready();           // (the join is above, outside the lock: joining under G.mu would deadlock with an initialisation thread that wants it)
	if (tl_slot >= 0) {
		// the worker of a split batch: the persistent context of its device slot (the slot's one worker thread)
		ThreadCtx *c = &g_slot_ctx[tl_slot];
		HIP_TRY(hipSetDevice(cur_device()));
		if (!c->st || g_slot_epoch[tl_slot] != G.epoch) {
			*c = ThreadCtx();
			hipError_t e = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking);
			if (e != hipSuccess) return fail(MM2C_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
			g_slot_epoch[tl_slot] = G.epoch;
		}
		*out = c;
		return 0;
	}
	if (tl_ctx && tl_epoch == G.epoch) { *out = tl_ctx; return 0; }
	HIP_TRY(hipSetDevice(cur_device()));
	ThreadCtx *c = new ThreadCtx();
	hipError_t e = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking);
	if (e != hipSuccess) { delete c; return fail(MM2C_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
	G.thread_ctxs.push_back(c);
	tl_ctx = c; tl_epoch = G.epoch;
	*out = c;
	return 0;
}

// Two batch contexts (stream set + grow-only arenas each).  A host whose mini-batches arrive one at a time (kt_pipeline runs a step for one mini-batch at a time,
// map.c:529-620) only ever uses the first, so its arenas are reused whichever pipeline thread calls; a second caller that arrives while the first is busy takes the
// other set instead of waiting for the whole call (upload, kernels, download), and only a third one waits.
static constexpr int N_BATCH_CTX = 2;
static ThreadCtx g_batch_ctx[N_BATCH_CTX];
static std::mutex g_batch_mu[N_BATCH_CTX];
static uint64_t g_batch_epoch[N_BATCH_CTX] = { ~0ull, ~0ull };

int get_batch_ctx(ThreadCtx **out, std::unique_lock<std::mutex> &hold)
{
	if (tl_slot >= 0) return get_thread_ctx(out);                   // this IS the slot's worker thread
	int k = 0;
	for (;;) {
		for (k = 0; k < N_BATCH_CTX; ++k) {
			hold = std::unique_lock<std::mutex>(g_batch_mu[k], std::try_to_lock);
			if (hold.owns_lock()) break;
		}
		if (k < N_BATCH_CTX) break;
		// both busy: a third caller takes whichever set is given back first (it used to queue on the first one even when the second came free earlier); batch calls last
		// milliseconds, so looking every 100 us costs nothing.  (Each set keeps its own grow-only arenas: once two pipeline threads have overlapped, the resident
		// footprint of big mini-batches is doubled; mm2c_tune("trim", 0) gives the device cache back, the arenas go with mm2c_shutdown.)
		std::this_thread::sleep_for(std::chrono::microseconds(100));
	}
	async_init_join();
	std::lock_guard<std::mutex> lk(G.mu);
	if (!G.ready) return fail_not_ready();
	if (!g_batch_ctx[k].st || g_batch_epoch[k] != G.epoch) {
		g_batch_ctx[k] = ThreadCtx();
		DeviceScope on(G.device);
		hipError_t e = on.err;
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&g_batch_ctx[k].st, hipStreamNonBlocking);
		if (e != hipSuccess) return fail(MM2C_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
		g_batch_epoch[k] = G.epoch;
	}
	*out = &g_batch_ctx[k];
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
	if (*p) { ScopedNs timed(SS.free_ns); ++SS.n_free; (void)hipFree(*p); }
	*p = nullptr; *cap = 0;
	ScopedNs timed(SS.alloc_ns); ++SS.n_alloc;
	HIP_TRY(hipMalloc((void **)p, want));
	*cap = want;
	return 0;
}

int grow_pinned(char **p, size_t *cap, size_t need, bool gpu_addressed)
{
	if (need <= *cap) return 0;
	const size_t want = std::max(need, *cap * 2);
	if (*p) { ScopedNs timed(SS.free_ns); ++SS.n_free; (void)hipHostFree(*p); }
	*p = nullptr; *cap = 0;
	ScopedNs timed(SS.alloc_ns); ++SS.n_alloc;
	// gpu_addressed: kernels read / write the buffer themselves (host_stage.hip) and the host reads results behind a flag word, not behind a stream wait: mapped
	// into the device's address space and coherent (fine-grained) by request rather than by the runtime's default
	HIP_TRY(hipHostMalloc((void **)p, want, gpu_addressed ? (hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) : hipHostMallocDefault));
	*cap = want;
	return 0;
}

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

// offsets sanity alone (what a per-read call needs: no allocation, no sort)
int check_offsets(int64_t n_tasks, const int64_t *off)
{
	if (n_tasks < 0 || n_tasks > INT32_MAX) return fail(MM2C_E_ARG, "n_tasks out of range");
	if (n_tasks > 0 && !off) return fail(MM2C_E_ARG, "offsets is NULL");
	for (int64_t k = 0; k < n_tasks; ++k) {
		const int64_t n = off[k + 1] - off[k];
		if (n < 0) return fail(MM2C_E_ARG, "offsets not monotone at task %lld", (long long)k);
		if (n >= (int64_t)INT32_MAX - 64)
			return fail(MM2C_E_TOOBIG, "task %lld has %lld anchors; the limit is 2^31-65 (cf. chain_hardware.cpp:34)", (long long)k, (long long)n);
	}
	return 0;
}

// offsets sanity + longest-first launch order (so the tail of the grid is made of short tasks)
int build_order(int64_t n_tasks, const int64_t *off, std::vector<int32_t> &order)
{
	if (const int rc = check_offsets(n_tasks, off)) return rc;
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

} // namespace mm2c_api

using namespace mm2c_api;

struct mm2c_plan {
	mm2c_params_t par;
	int device = 0;                         // the device the plan's workspace lives on
	const int64_t *d_off_user = nullptr;    // mm2c_plan_set_device_offsets: task sizes that only the device knows
	int64_t n_tasks = 0, total = 0;
	int64_t *d_off = nullptr; int32_t *d_order = nullptr, *d_status = nullptr, *d_t = nullptr, *d_st = nullptr; float *d_avg_ws = nullptr; uint8_t *d_cls = nullptr;
	unsigned long long *d_seg_ws = nullptr; // plans with long tasks: the words in which the segments of a task add up the prepass's sums (chain_window_start_t<true>; zero between runs)
	hipEvent_t ev_pre = nullptr, ev0 = nullptr, ev1 = nullptr, ev_e0 = nullptr, ev_e1 = nullptr;
	mm2c_api::AuxSet aux;                   // helper stream + fork / join events (pooled): taken by the first run that may split its tasks over two instantiations
	bool ran = false, epi_ran = false;
	mm2c::LaunchInfo info = {};             // what the last run launched (mm2c_plan_last_variant)
	std::vector<int32_t> sizes_desc;        // task sizes, longest first (host copy: bounds the number of pieces of the device-side cut)
	char *d_cut = nullptr;                  // piece arrays of the device-side cut (chain_cut), allocated by the first run that cuts
	mm2c::CutArgs cut;
	char *d_epi = nullptr;                  // scratch of the device epilogue, allocated by the first mm2c_plan_chains_device
	mm2c::EpiArgs E;
};

extern "C" {

const char *mm2c_last_error(void) { return g_err; }

int mm2c_init(int device_ordinal)
{
	async_init_join();                               // (a no-op on the initialisation thread itself)
	std::lock_guard<std::mutex> lk(G.mu);
	if (G.ready) return 0;
	// the pipelines of the host-buffer entries run an upload stream and three compute streams side by side; with the runtime's default of four
	// hardware queues per process two of them can land on one queue and then take turns (profiles/r3_e2e.md).  The runtime reads GPU_MAX_HW_QUEUES
	// when it starts, and the environment is the host's, not the library's, to change (setenv is not safe against getenv in the host's other
	// threads and would alter every HIP user of the process): the host exports GPU_MAX_HW_QUEUES=16 before its first HIP call (INTEGRATION.md C;
	// bench.py and oracle/ref_host's drivers do).  Without it the pipelines still work -- their streams differ in priority where the device
	// offers that (create_partner_stream) -- and the library says so once.
	{
		const char *q = getenv("GPU_MAX_HW_QUEUES");
		static bool told = false;
		if ((!q || atoi(q) < 8) && !told && getenv("MM2C_QUIET") == nullptr) {
			told = true;
			fprintf(stderr, "[mm2chain] GPU_MAX_HW_QUEUES is %s: the upload / compute streams of the host-batch pipelines may share a hardware queue "
			                "(export GPU_MAX_HW_QUEUES=16 before the process's first HIP call)\n", q ? q : "unset (runtime default 4)");
		}
	}
	int n_dev = 0;
	hipError_t e = hipGetDeviceCount(&n_dev);
	if (e != hipSuccess || n_dev <= 0)
		return fail(MM2C_E_NODEVICE, "no HIP device: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
	int dev = device_ordinal;
	std::vector<int> env_devices;
	if (dev < 0 && G.devices.empty()) {
		// a host that cannot pass ordinals (hardware_init(long, char *), chain_hardware.h:69) names its devices in the environment:
		// MM2C_DEVICES=all, or a comma-separated list of ordinals; batches are then split across them as after mm2c_init_devices
		const char *env = getenv("MM2C_DEVICES");
		if (env && *env) {
			std::vector<int> ds;
			if (strcmp(env, "all") == 0) { for (int k = 0; k < n_dev && k < 64; ++k) ds.push_back(k); }
			else {
				for (const char *c = env; *c;) {
					char *end = nullptr;
					const long v = strtol(c, &end, 10);
					if (end == c || v < 0 || v >= n_dev || ds.size() >= 64) return fail(MM2C_E_NODEVICE, "MM2C_DEVICES=%s: bad ordinal (%d devices)", env, n_dev);
					ds.push_back((int)v);
					c = (*end == ',') ? end + 1 : end;
					if (*end != ',' && *end != 0) return fail(MM2C_E_ARG, "MM2C_DEVICES=%s: comma-separated ordinals or `all` expected", env);
				}
			}
			if (!ds.empty()) { env_devices = ds; dev = ds[0]; }
		}
	}
	if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
	if (dev >= n_dev) return fail(MM2C_E_NODEVICE, "device ordinal %d out of range (%d devices)", dev, n_dev);
	HIP_TRY(hipSetDevice(dev));
	HIP_TRY(hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking));
	G.device = dev;
	if (!env_devices.empty()) G.devices = env_devices;
	if (G.devices.empty()) G.devices.assign(1, dev);
	const char *rc = getenv("MM2C_RING_CLASS");
	G.ring_class = rc ? std::max(0, std::min(4, atoi(rc))) : 3;
	const char *ef = getenv("MM2C_EPI_FUSED");           // 0: the device epilogue works in HBM for every task (kernels A, B, C)
	if (ef) G.epi_fused = atoi(ef) != 0;
	const char *fr = getenv("MM2C_FAR_RING");            // 0: one LDS ring size for every task; 2: the long ring for every task (tests)
	if (fr) G.far_ring = std::max(0, std::min(2, atoi(fr)));
	const char *ss = getenv("MM2C_SPLIT_STREAMS");       // 0: the instantiations of a split batch run one after the other on the caller's stream
	if (ss) G.split_streams = std::max(0, std::min(2, atoi(ss)));
	const char *wp = getenv("MM2C_WIDE_SHARE_THRESHOLD"); // % of the anchors in tasks that need the 32-bit ring from which every task takes it
	if (wp) G.wide_pct = std::max(0, std::min(100, atoi(wp)));
	const char *cm = getenv("MM2C_COMBINE_MAX");         // experiments: 0 = no call combiner
	if (cm) G.combine_max_anchors = (size_t)std::max(0, atoi(cm));
	const char *cl = getenv("MM2C_COMBINER_LANES");      // experiments: passes of the call combiner in flight
	if (cl) G.combiner_lanes = std::max(1, std::min(16, atoi(cl)));
	const char *dp = getenv("MM2C_DIRECT_PASS");         // 0: small per-read passes use copy commands and a stream wait instead of the staging kernels and the polled flag (experiments)
	if (dp) G.direct_pass = atoi(dp) != 0;
	const char *cw = getenv("MM2C_COOP_WAVES");          // 0: the host-buffer entries never use several waves per task (experiments; the tests use mm2c_tune)
	if (cw) G.coop_waves = std::max(0, atoi(cw));
	const char *pc = getenv("MM2C_PIPE_COOP_CHUNKS");    // experiments: the last chunks of a pipelined host batch with several waves per piece
	if (pc) G.pipe_coop_chunks = std::max(0, atoi(pc));
	const char *q4 = getenv("MM2C_Q24_RING");            // 0: the long ring of class-1 tasks keeps its 32-bit slots (experiments; the tests use mm2c_tune)
	if (q4) G.q24_ring = atoi(q4) != 0;
	const char *cr = getenv("MM2C_COMPACT_RING");        // 0: never the compact x / q ring of the tile kernel (experiments; the tests use mm2c_tune)
	if (cr) G.compact_ring = atoi(cr) != 0;
	const char *sl = getenv("MM2C_SINGLE_LAUNCH");       // 0: per-read passes keep stage_in (experiments)
	if (sl) G.single_launch = atoi(sl) != 0;
	const char *fs = getenv("MM2C_FUSE_ST");             // 0: per-read passes keep the prepass launch for the window starts (experiments)
	if (fs) G.fuse_st = atoi(fs) != 0;
	const char *dwb = getenv("MM2C_DECLINE_WHEN_BUSY");  // a path-A host (no mm2c_tune call site) opts into the busy protocol here: 1 / 2 = the rules of mm2chain_host.cpp, book_pred
	if (dwb) G.decline_when_busy = std::max(0, std::min(2, atoi(dwb)));
	const char *ft = getenv("MM2C_FAR_RING_THRESHOLD");  // tenths of an expected far tile per anchor from which a task takes the long ring
	if (ft) G.far_thr10 = std::max(0, atoi(ft));
	G.ready = true;
	return 0;
}

/* /sys/bus/pci/devices/<bus id>/numa_node and /sys/devices/system/node/node<N>/cpulist under `sysfs_root`: pure file reading, no device is touched */
int mm2c_numa_cpulist(const char *sysfs_root, const char *pci_bus_id, char *buf, size_t len)
{
	if (!sysfs_root || !pci_bus_id || !buf || len == 0) return -1;
	buf[0] = 0;
	char path[512];
	snprintf(path, sizeof(path), "%s/bus/pci/devices/%s/numa_node", sysfs_root, pci_bus_id);
	FILE *fp = fopen(path, "r");
	if (!fp) return -1;
	int node = -1;
	if (fscanf(fp, "%d", &node) != 1) node = -1;
	fclose(fp);
	if (node < 0) return -1;
	snprintf(path, sizeof(path), "%s/devices/system/node/node%d/cpulist", sysfs_root, node);
	fp = fopen(path, "r");
	if (!fp) return -1;
	const bool ok = fgets(buf, (int)len, fp) != nullptr;
	fclose(fp);
	if (!ok) { buf[0] = 0; return -1; }
	for (char *c = buf; *c; ++c) if (*c == '\n') *c = 0;
	return buf[0] ? node : -1;
}

int mm2c_slot_worker_node(int slot)
{
	if (slot < 0 || slot >= 64) return -1;
	std::lock_guard<std::mutex> lk(g_workers[slot].mu);
	return g_workers[slot].started ? g_workers[slot].pinned_node : -1;
}

int mm2c_init_async(int device_ordinal)
{
	if (tl_is_init_thread) return 0;
	std::lock_guard<std::mutex> lk(g_async_mu);      // held across the join of the previous start AND the creation of the next: two concurrent callers take turns,
	async_join_locked();                             // and the second one finds the first one's thread either finished (joined here) or the library ready
	if (G.ready) return 0;
	g_async_err[0] = 0;
	try {
		g_async_pending.store(true, std::memory_order_release);
		delete g_async_th; g_async_th = nullptr;     // (joined above: never a joinable thread)
		g_async_th = new std::thread([device_ordinal]() {
			tl_is_init_thread = true;
			int rc = mm2c_init(device_ordinal);
			if (rc == 0) rc = mm2c_warm_up();
			if (rc != 0) { strncpy(g_async_err, g_err, sizeof(g_async_err) - 1); g_async_err[sizeof(g_async_err) - 1] = 0; }
		});
	} catch (...) {
		g_async_pending.store(false, std::memory_order_release);
		return fail(MM2C_E_ARG, "could not start the initialisation thread");
	}
	return 0;
}

int mm2c_init_wait(void)
{
	async_init_join();
	return G.ready ? 0 : fail_not_ready();
}

// loads the code objects of the path's kernels onto every configured device now (the runtime otherwise does it at the first launch of each translation unit: the
// first chaining calls of a run would pay for it)
int mm2c_warm_up(void)
{
	if (!lib_ready()) return fail_not_ready();
	std::vector<int> devs;
	{ std::lock_guard<std::mutex> lk(G.mu); devs = G.devices; }
	int prev = -1;
	(void)hipGetDevice(&prev);
	hipError_t e = hipSuccess;
	for (size_t k = 0; k < devs.size() && e == hipSuccess; ++k) {
		bool seen = false;
		for (size_t j = 0; j < k; ++j) seen = seen || devs[j] == devs[k];
		if (seen) continue;
		e = hipSetDevice(devs[k]);
		if (e == hipSuccess) e = mm2c::warm_chain_kernels();
		if (e == hipSuccess) e = mm2c::warm_epilogue_kernels();
		if (e == hipSuccess) e = mm2c::warm_seed_kernels();
		if (e == hipSuccess) e = mm2c::warm_stage_kernels();
	}
	if (prev >= 0) (void)hipSetDevice(prev);
	return e == hipSuccess ? 0 : fail(MM2C_E_HIP, "warm-up: %s", hipGetErrorString(e));
}

int mm2c_init_devices(int n, const int *ordinals)
{
	if (n < 1 || n > 64 || !ordinals) return fail(MM2C_E_ARG, "1 to 64 device ordinals expected");
	async_init_join();
	{
		std::lock_guard<std::mutex> lk(G.mu);
		if (G.ready) return fail(MM2C_E_ARG, "mm2c_init_devices after initialisation: call mm2c_shutdown first");
		int n_dev = 0;
		hipError_t e = hipGetDeviceCount(&n_dev);
		if (e != hipSuccess || n_dev <= 0)
			return fail(MM2C_E_NODEVICE, "no HIP device: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
		for (int k = 0; k < n; ++k)
			if (ordinals[k] < 0 || ordinals[k] >= n_dev) return fail(MM2C_E_NODEVICE, "device ordinal %d out of range (%d devices)", ordinals[k], n_dev);
		G.devices.assign(ordinals, ordinals + n);
	}
	const int rc = mm2c_init(ordinals[0]);
	if (rc != 0) { std::lock_guard<std::mutex> lk(G.mu); G.devices.clear(); }
	return rc;
}

int mm2c_device_count(void) { return lib_ready() ? (int)G.devices.size() : 0; }

/* Contiguous ranges of tasks with about equal anchor counts: range s = tasks [bounds[s], bounds[s+1]).  A range ends at the first task
 * boundary at or beyond s+1 parts of the total (so no range exceeds its share by more than one task); ranges may be empty. */
int mm2c_split_tasks(int64_t n_tasks, const int64_t *offsets, int n_parts, int64_t *bounds)
{
	if (n_tasks < 0 || n_parts < 1 || !bounds || (n_tasks > 0 && !offsets)) return fail(MM2C_E_ARG, "bad argument");
	bounds[0] = 0;
	const int64_t total = n_tasks > 0 ? offsets[n_tasks] - offsets[0] : 0;
	int64_t k = 0;
	for (int s = 1; s <= n_parts; ++s) {
		if (s == n_parts) { bounds[s] = n_tasks; break; }
		const int64_t goal = offsets ? offsets[0] + (int64_t)((__int128)total * s / n_parts) : 0;
		while (k < n_tasks && offsets[k] < goal) ++k;
		bounds[s] = k;
	}
	return 0;
}

void mm2c_shutdown(void)
{
	async_init_join();
	stop_slot_workers();
	std::lock_guard<std::mutex> bl0(g_batch_mu[0]);   // same order as get_batch_ctx: a batch context's lock (a caller holds at most one), then the library's
	std::lock_guard<std::mutex> bl1(g_batch_mu[1]);
	std::lock_guard<std::mutex> lk(G.mu);
	if (!G.ready) return;
	(void)hipSetDevice(cur_device());
	(void)hipDeviceSynchronize();
	for (ThreadCtx *c : G.thread_ctxs) { c->release(); delete c; }
	for (size_t k = 0; k < G.devices.size() && k < 64; ++k) if (g_slot_ctx[k].st) { (void)hipSetDevice(G.devices[k]); (void)hipDeviceSynchronize(); g_slot_ctx[k].release(); }
	(void)hipSetDevice(G.device);
	for (int k = 0; k < N_BATCH_CTX; ++k) { if (g_batch_ctx[k].st) g_batch_ctx[k].release(); g_batch_epoch[k] = ~0ull; }
	release_combiner();
	release_seed_aux();
	dev_cache_release();
	G.thread_ctxs.clear();
	if (G.stream) (void)hipStreamDestroy(G.stream);
	G.stream = nullptr;
	G.devices.clear();
	G.ready = false;
	++G.epoch;
}

int mm2c_device_info(char *name, size_t name_len, int *cu_count, size_t *hbm_bytes)
{
	if (!lib_ready()) return fail_not_ready();
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, cur_device()));
	if (name && name_len) { snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName); }
	if (cu_count) *cu_count = prop.multiProcessorCount;
	if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
	return 0;
}

int mm2c_debug_label_hits(unsigned long long *hits, int reset)
{
	if (!lib_ready()) return fail_not_ready();
	if (!hits) return fail(MM2C_E_ARG, "hits is NULL");
	const hipError_t e = mm2c::label_hits_read(hits, reset != 0);
	if (e == hipErrorNotSupported) return fail(MM2C_E_ARG, "this library was not built with -DMM2C_LABEL_COUNT (minimap2-fpga_amd/variants/labelcount.so is)");
	HIP_TRY(e);
	return 0;
}

int mm2c_device_identity(int *ordinal, char *pci_bus_id, size_t bus_len, char *arch, size_t arch_len)
{
	if (!lib_ready()) return fail_not_ready();
	const int dev = cur_device();
	if (ordinal) *ordinal = dev;
	if (pci_bus_id && bus_len) {
		pci_bus_id[0] = 0;
		HIP_TRY(hipDeviceGetPCIBusId(pci_bus_id, (int)std::min<size_t>(bus_len, 64), dev));
	}
	if (arch && arch_len) {
		hipDeviceProp_t prop;
		HIP_TRY(hipGetDeviceProperties(&prop, dev));
		snprintf(arch, arch_len, "%s", prop.gcnArchName);
	}
	return 0;
}

int mm2c_tune(const char *key, int value)
{
	std::lock_guard<std::mutex> lk(G.mu);
	if (key && strcmp(key, "trim") == 0) { dev_cache_release(); return 0; }   // give the cached device memory back to the runtime
	if (!key) return fail(MM2C_E_ARG, "key is NULL");
	if (strcmp(key, "ring_class") == 0) {
		if (value < 0 || value > 4) return fail(MM2C_E_ARG, "ring_class must be 0 .. 4");
		G.ring_class = value;
		return 0;
	}
	if (strcmp(key, "far_ring") == 0) {
		if (value < 0 || value > 2) return fail(MM2C_E_ARG, "far_ring must be 0, 1 or 2");
		G.far_ring = value;
		return 0;
	}
	if (strcmp(key, "far_ring_threshold") == 0) {
		if (value < 0) return fail(MM2C_E_ARG, "far_ring_threshold (tenths of a far tile per anchor) must be >= 0");
		G.far_thr10 = value;
		return 0;
	}
	if (strcmp(key, "noskip_loop") == 0) {
		if (value < 0 || value > 1) return fail(MM2C_E_ARG, "noskip_loop must be 0 or 1");
		G.noskip_loop = value;
		return 0;
	}
	if (strcmp(key, "heap_sort") == 0) {
		if (value < 0 || value > 1) return fail(MM2C_E_ARG, "heap_sort must be 0 or 1");
		G.heap_sort = value;
		return 0;
	}
	if (strcmp(key, "wide_share_threshold") == 0) {
		if (value < 0 || value > 100) return fail(MM2C_E_ARG, "wide_share_threshold is a percentage");
		G.wide_pct = value;
		return 0;
	}
	if (strcmp(key, "split_streams") == 0) {
		if (value < 0 || value > 2) return fail(MM2C_E_ARG, "split_streams must be 0, 1 (batches of mixed task sizes) or 2 (every batch)");
		G.split_streams = value;
		return 0;
	}
	if (strcmp(key, "compact_ring") == 0) {
		if (value < 0 || value > 1) return fail(MM2C_E_ARG, "compact_ring must be 0 or 1");
		G.compact_ring = value;
		return 0;
	}
	if (strcmp(key, "force_tab") == 0) {
		if (value < 0 || value > 1) return fail(MM2C_E_ARG, "force_tab must be 0 or 1");
		G.force_tab = value;
		return 0;
	}
	if (strcmp(key, "epi_fused") == 0) {
		if (value < 0 || value > 1) return fail(MM2C_E_ARG, "epi_fused must be 0 or 1");
		G.epi_fused = value;
		return 0;
	}
	if (strcmp(key, "pipeline_chunk_anchors") == 0) {
		if (value < 1024) return fail(MM2C_E_ARG, "pipeline_chunk_anchors must be >= 1024");
		G.pipeline_chunk_anchors = value;
		return 0;
	}
	if (strcmp(key, "pipeline_pieces") == 0) {
		if (value < 1) return fail(MM2C_E_ARG, "pipeline_pieces must be >= 1");
		G.pipeline_pieces = value;
		return 0;
	}
	if (strcmp(key, "pipeline_taper") == 0) {
		if (value < 0 || value > 6) return fail(MM2C_E_ARG, "pipeline_taper must be 0 .. 6");
		G.pipeline_taper = value;
		return 0;
	}
	if (strcmp(key, "pipeline_min_chunk") == 0) {
		if (value < 1024) return fail(MM2C_E_ARG, "pipeline_min_chunk must be >= 1024");
		G.pipeline_min_chunk = value;
		return 0;
	}
	if (strcmp(key, "coop_waves") == 0) {
		if (value < 0 || value > 64) return fail(MM2C_E_ARG, "coop_waves must be 0 .. 64 (0 / 1: one wave per task always; any other value: the cooperative kernel, which is built for 16 waves per task)");
		G.coop_waves = value;
		return 0;
	}
	if (strcmp(key, "combine_max_anchors") == 0) {            // calls of up to this many anchors go through the call combiner (0: every call runs a pass of its own on its thread's stream)
		if (value < 0) return fail(MM2C_E_ARG, "combine_max_anchors must be >= 0");
		G.combine_max_anchors = (size_t)value;
		return 0;
	}
	if (strcmp(key, "combiner_lanes") == 0) {
		if (value < 1 || value > 16) return fail(MM2C_E_ARG, "combiner_lanes must be 1 .. 16");
		G.combiner_lanes = value;
		return 0;
	}
	if (strcmp(key, "pipe_coop_chunks") == 0) {
		if (value < 0) return fail(MM2C_E_ARG, "pipe_coop_chunks must be >= 0");
		G.pipe_coop_chunks = value;
		return 0;
	}
	if (strcmp(key, "q24_ring") == 0) {
		G.q24_ring = value != 0;
		return 0;
	}
	if (strcmp(key, "pin_workers") == 0) {
		G.pin_workers = value != 0;
		return 0;
	}
	if (strcmp(key, "decline_when_busy") == 0) {
		if (value < 0 || value > 2) return fail(MM2C_E_ARG, "decline_when_busy must be 0 (never decline), 1 (by the slot's measured service time) or 2 (round 5's rule: booked predictions)");
		G.decline_when_busy = value;
		return 0;
	}
	if (strcmp(key, "direct_pass") == 0) {
		G.direct_pass = value != 0;
		return 0;
	}
	if (strcmp(key, "coop_plans") == 0) {
		if (value < 0 || value > 2) return fail(MM2C_E_ARG, "coop_plans must be 0 (never), 1 (every plan of at most coop_max_tasks tasks) or 2 (per run: few long pieces)");
		G.coop_plans = value;
		return 0;
	}
	if (strcmp(key, "host_st") == 0) {
		G.host_st = value != 0;
		return 0;
	}
	if (strcmp(key, "fused_out") == 0) {
		G.fused_out = value != 0;
		return 0;
	}
	if (strcmp(key, "single_launch") == 0) { G.single_launch = value != 0; return MM2C_OK; }   // per-read passes of short tasks: the cooperative kernel reads the pinned arena itself (1) or stage_in uploads it first (0)
	if (strcmp(key, "fuse_st") == 0) { G.fuse_st = value != 0; return MM2C_OK; }   // passes of few tasks of at most 7 168 anchors: the sixteen-wave kernel makes the window starts itself (1) or a prepass launch does (0)
	if (strcmp(key, "seg_prepass") == 0) { G.seg_prepass = value != 0; return MM2C_OK; }   // plans with tasks of 65 536 anchors or more: a prepass block per 32 768 anchors (1) or per task (0)
	if (strcmp(key, "coop_w8_above") == 0) {
		if (value < 0) return fail(MM2C_E_ARG, "coop_w8_above must be >= 0 (pieces beyond which the cooperative kernel takes eight waves per piece instead of sixteen)");
		G.coop_w8_above = (int)std::min<int64_t>(value, 1 << 30);
		return MM2C_OK;
	}
	if (strcmp(key, "coop_max_tasks") == 0) {
		if (value < 0) return fail(MM2C_E_ARG, "coop_max_tasks must be >= 0");
		G.coop_max_tasks = value;
		return 0;
	}
	if (strcmp(key, "plan_cut") == 0) {
		if (value < 0 || value > 1) return fail(MM2C_E_ARG, "plan_cut must be 0 or 1");
		G.plan_cut = value;
		return 0;
	}
	if (strcmp(key, "plan_cut_min") == 0) {
		if (value < 1) return fail(MM2C_E_ARG, "plan_cut_min must be >= 1");
		G.plan_cut_min = value;
		return 0;
	}
	if (strcmp(key, "multi_min_anchors") == 0) {
		if (value < 0) return fail(MM2C_E_ARG, "multi_min_anchors must be >= 0");
		G.multi_min_anchors = value;
		return 0;
	}
	if (strcmp(key, "seg_min") == 0) {
		if (value < 0) return fail(MM2C_E_ARG, "seg_min must be >= 0");
		G.seg_min = value;
		return 0;
	}
	return fail(MM2C_E_ARG, "unknown tuning key '%s'", key);
}

int mm2c_split_model(const char *preset, float *k1_hw, float *k2_hw, float *c_hw, float *k_sw, float *c_sw)
{
	// the constants of include/mm2chain_split.h (fitted on the MI355X box by tools/fit_split_model.py) for a host that keeps the reference's
	// predictor chain.c:80-81,101; presets as options.c:93-99 ("map-ont" and the other ONT / default presets) and :113-122 ("asm20", PacBio CCS)
	if (!preset || !k1_hw || !k2_hw || !c_hw || !k_sw || !c_sw) return fail(MM2C_E_ARG, "NULL argument");
	if (strcmp(preset, "map-ont") == 0 || strcmp(preset, "ont") == 0 || strcmp(preset, "ava-ont") == 0) {
		*k1_hw = (float)MI355X_ONT_K1_HW; *k2_hw = (float)MI355X_ONT_K2_HW; *c_hw = (float)MI355X_ONT_C_HW; *k_sw = (float)MI355X_ONT_K_SW; *c_sw = (float)MI355X_ONT_C_SW;
		return 0;
	}
	if (strcmp(preset, "asm20") == 0 || strcmp(preset, "pbccs") == 0 || strcmp(preset, "map-hifi") == 0) {
		*k1_hw = (float)MI355X_PBCCS_K1_HW; *k2_hw = (float)MI355X_PBCCS_K2_HW; *c_hw = (float)MI355X_PBCCS_C_HW; *k_sw = (float)MI355X_PBCCS_K_SW; *c_sw = (float)MI355X_PBCCS_C_SW;
		return 0;
	}
	return fail(MM2C_E_ARG, "unknown preset '%s' (map-ont, asm20)", preset);
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

void mm2c_get_stage_stats(mm2c_stage_stats_t *o)
{
	if (!o) return;
	o->calls = SS.calls; o->chunks = SS.chunks; o->total_ns = SS.total_ns; o->alloc_ns = SS.alloc_ns; o->n_alloc = SS.n_alloc; o->free_ns = SS.free_ns;
	o->n_free = SS.n_free; o->setup_ns = SS.setup_ns; o->h2d_ns = SS.h2d_ns; o->seed_ns = SS.seed_ns; o->dp_ns = SS.dp_ns; o->epi_ns = SS.epi_ns;
	o->d2h_ns = SS.d2h_ns; o->wait_ns = SS.wait_ns;
}

void mm2c_reset_stage_stats(void)
{
	SS.calls = 0; SS.chunks = 0; SS.total_ns = 0; SS.alloc_ns = 0; SS.n_alloc = 0; SS.free_ns = 0; SS.n_free = 0; SS.setup_ns = 0; SS.h2d_ns = 0;
	SS.seed_ns = 0; SS.dp_ns = 0; SS.epi_ns = 0; SS.d2h_ns = 0; SS.wait_ns = 0;
}

// ------------------------------------------------------------------------------------------------ plans
mm2c_plan_t *mm2c_plan_create(const mm2c_params_t *par, int64_t n_tasks, const int64_t *h_offsets)
{
	if (check_params(par)) return nullptr;
	if (!lib_ready()) { fail_not_ready(); return nullptr; }
	std::vector<int32_t> order;
	if (build_order(n_tasks, h_offsets, order)) return nullptr;
	mm2c_plan *pl = new mm2c_plan();
	pl->par = *par; pl->n_tasks = n_tasks;
	pl->total = n_tasks > 0 ? h_offsets[n_tasks] - h_offsets[0] : 0;
	pl->sizes_desc.resize((size_t)n_tasks);
	for (int64_t k = 0; k < n_tasks; ++k) pl->sizes_desc[(size_t)k] = (int32_t)(h_offsets[order[(size_t)k] + 1] - h_offsets[order[(size_t)k]]);
	pl->device = cur_device();
	DeviceScope on(pl->device);
	hipError_t e = on.err;
	const size_t nt = (size_t)std::max<int64_t>(n_tasks, 1), tot = (size_t)std::max<int64_t>(pl->total, 1);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_off, (nt + 1) * 8);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_order, nt * 4);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_status, nt * 4);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_t, tot * 4);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_st, tot * 4);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_avg_ws, nt * 4);
	if (e == hipSuccess) e = dev_alloc((void **)&pl->d_cls, ((nt + 15) & ~(size_t)15) + 32 * mm2c::CLS_STAT_SLOTS);   // class per task + the counter sets of chain_cls_settle
	if (e == hipSuccess && n_tasks > 0 && pl->sizes_desc[0] >= 65536) {                                              // long tasks: a prepass block per segment of a task
		e = dev_alloc((void **)&pl->d_seg_ws, nt * 32);
		if (e == hipSuccess) e = hipMemset(pl->d_seg_ws, 0, nt * 32);
	}
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

static void plan_destroy_impl(mm2c_plan_t *pl, bool wait)
{
	if (!pl) return;
	{
		DeviceScope on(pl->device);
		if (wait && (pl->ran || pl->epi_ran)) { ScopedNs timed(SS.free_ns); (void)hipDeviceSynchronize(); }   // ONE wait, as hipFree would: the blocks go back to the cache and may be reused at once
		dev_free_synced(pl->d_off); dev_free_synced(pl->d_order); dev_free_synced(pl->d_status); dev_free_synced(pl->d_t); dev_free_synced(pl->d_st);
		dev_free_synced(pl->d_avg_ws); dev_free_synced(pl->d_seg_ws); dev_free_synced(pl->d_cls); dev_free_synced(pl->d_epi); dev_free_synced(pl->d_cut);
		if (pl->ev_pre) (void)hipEventDestroy(pl->ev_pre);
		if (pl->ev0) (void)hipEventDestroy(pl->ev0); if (pl->ev1) (void)hipEventDestroy(pl->ev1);
		if (pl->ev_e0) (void)hipEventDestroy(pl->ev_e0); if (pl->ev_e1) (void)hipEventDestroy(pl->ev_e1);
		aux_release(pl->aux);                 // the helper stream joins the plan's stream at the end of every run: idle once that stream has been waited for
	}
	delete pl;
}

void mm2c_plan_destroy(mm2c_plan_t *pl) { plan_destroy_impl(pl, true); }

int64_t mm2c_plan_total_anchors(const mm2c_plan_t *pl) { return pl ? pl->total : 0; }

int mm2c_plan_set_device_offsets(mm2c_plan_t *pl, const int64_t *d_offsets)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	pl->d_off_user = d_offsets;
	return 0;
}

int mm2c_plan_run_device(mm2c_plan_t *pl, const void *d_anchors, const float *d_avg_qspan, int32_t *d_f, int32_t *d_p, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (!lib_ready()) return fail_not_ready();
	if (pl->n_tasks == 0 || pl->total == 0) return 0;
	if (!d_anchors || !d_f || !d_p) return fail(MM2C_E_ARG, "device pointer is NULL");
	DeviceScope on(pl->device);
	HIP_TRY(on.err);
	hipStream_t st;
	if (const int rc = resolve_stream(stream, pl->device, &st)) return rc;
	mm2c::LaunchArgs L;
	L.P = to_kparams(&pl->par);
	L.n_tasks = pl->n_tasks; L.d_offsets = pl->d_off_user ? pl->d_off_user : pl->d_off; L.d_order = pl->d_order;
	L.d_anchors = d_anchors; L.d_avg = d_avg_qspan; L.d_pbase = nullptr; L.d_f = d_f; L.d_p = d_p; L.d_t = pl->d_t; L.d_st = pl->d_st; L.d_status = pl->d_status;
	L.d_avg_ws = pl->d_avg_ws;
	L.d_cls = pl->d_cls; L.far_ring = G.far_ring; L.far_thr10 = G.far_thr10;
	L.d_cls_stat = (unsigned long long *)(pl->d_cls + (((size_t)std::max<int64_t>(pl->n_tasks, 1) + 15) & ~(size_t)15));
	HIP_TRY(hipMemsetAsync(L.d_cls_stat, 0, 32 * mm2c::CLS_STAT_SLOTS, st));
	L.ring_class = G.ring_class; L.force_tab = G.force_tab; L.compact = G.compact_ring; L.q24 = G.q24_ring; L.wide_pct = G.wide_pct; L.noskip_loop = G.noskip_loop;
	HIP_TRY(hipMemsetAsync(pl->d_status, 0, (size_t)pl->n_tasks * 4, st));
	// Few long pieces: several waves per piece (chain_dp_coop.h; launch_chain_dp takes it for the variants of the hand-written loop).  "coop_plans" 2 (default, round 6):
	// decided per run by coop_pays (chain_kernel.h) -- here when the tasks run as they are, on the device (chain_route) when long tasks are cut into pieces first;
	// 1: every plan of at most coop_max_tasks tasks, uncut (the parity tests' way to the kernel); 0: never.
	const int coop_mode = G.coop_waves.load() > 1 ? G.coop_plans.load() : 0;
	const int64_t longest = pl->sizes_desc.empty() ? 0 : (int64_t)pl->sizes_desc[0];
	const bool will_cut = G.plan_cut && G.seg_min > 0 && longest >= G.plan_cut_min;
	if (pl->d_seg_ws && !pl->d_off_user && G.seg_prepass.load()) { L.d_seg_ws = pl->d_seg_ws; L.longest_task = longest; }
	L.coop_waves = 0; L.coop_w8_above = G.coop_w8_above.load(); L.fuse_st = G.fuse_st.load();
	if (coop_mode == 1 && pl->n_tasks <= G.coop_max_tasks) L.coop_waves = G.coop_waves.load();
	else if (coop_mode == 2 && !will_cut && !pl->d_off_user && mm2c::coop_pays(pl->n_tasks, longest, pl->total, G.coop_w8_above.load())) L.coop_waves = G.coop_waves.load();
	else if (coop_mode == 2 && will_cut) L.coop_waves = -1;
	if (L.coop_waves > 1) L.max_task_anchors = longest;
	if (L.coop_waves <= 1 && G.plan_cut && G.seg_min > 0) {
		// long reads are chains of loci: cut them at empty windows into independent pieces (one wave each) on the device.  Only tasks of
		// plan_cut_min anchors or more (they make the tail of the batch); a batch without any runs exactly as before.
		// (a plan of few very long tasks -- the regime of the cooperative kernel -- cuts no piece shorter than 1/64 of its longest task: every kernel of the run is launched
		// over the pieces the cut MAY make, and 255 tasks of 10^6 anchors at 256 anchors a piece were 10^6 workgroups per launch, four of five launches with nothing to do)
		const int seg_min = (pl->n_tasks <= mm2c::COOP_ROUTE_MAX_PIECES && longest >= 65536) ? (int)std::max<int64_t>(G.seg_min, longest / 64) : (int)G.seg_min;
		int64_t extra = 0;
		for (size_t k = 0; k < pl->sizes_desc.size() && pl->sizes_desc[k] >= G.plan_cut_min; ++k) extra += pl->sizes_desc[k] / seg_min;
		const int64_t max_pieces = pl->n_tasks + extra;
		if (extra > 0 && max_pieces <= (int64_t)INT32_MAX) {
			if (!pl->d_cut || pl->cut.max_pieces != max_pieces || pl->cut.seg_min != seg_min) {
				dev_free(pl->d_cut); pl->d_cut = nullptr;
				const size_t mp = (size_t)max_pieces;
				size_t at = 0;
				auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return o; };
				const size_t o_cnt = take(4), o_stat = take(mp * 4), o_hc = take((size_t)pl->n_tasks * 4), o_start = take(mp * 8), o_end = take(mp * 8),
				             o_pb = take(mp * 4), o_avg = take(mp * 4), o_cls = take(mp);
				HIP_TRY(dev_alloc((void **)&pl->d_cut, at));
				char *b = pl->d_cut;
				pl->cut.max_pieces = max_pieces; pl->cut.seg_min = seg_min;
				pl->cut.d_count = (int32_t *)(b + o_cnt); pl->cut.d_status = (int32_t *)(b + o_stat); pl->cut.d_has_cut = (int32_t *)(b + o_hc);
				pl->cut.d_start = (int64_t *)(b + o_start); pl->cut.d_end = (int64_t *)(b + o_end);
				pl->cut.d_pbase = (int32_t *)(b + o_pb); pl->cut.d_avg = (float *)(b + o_avg); pl->cut.d_cls = (uint8_t *)(b + o_cls);
			}
			pl->cut.min_anchors = G.plan_cut_min;
			HIP_TRY(hipMemsetAsync(pl->d_cut, 0, 256 + (((size_t)max_pieces * 4 + 255) & ~(size_t)255) + (((size_t)pl->n_tasks * 4 + 255) & ~(size_t)255), st));   // count + status + has_cut
			L.cut = pl->cut;
		}
	}
	// Two streams for a batch that is split between the compact and the 32-bit instantiations (chain_kernel.hip, launch_tile_one).  Which tasks take which is
	// decided on the device (the span of their q values), so the host goes by what it knows, the task sizes: a batch of equal-sized tasks is taken to be of one
	// kind and runs on the caller's stream alone -- beside a kernel that has all the tasks, the other one's workgroups (one per task, each returning at once) only
	// take slots away from it: 25.8 -> 26.5 ms on the colinear stream, 35.1 -> 36.8 on ava-ont colinear, whose device-side cut sizes the grid for 1.3 million pieces.
	const bool mixed_sizes = !pl->sizes_desc.empty() && (int64_t)pl->sizes_desc[0] * 4 > (int64_t)pl->sizes_desc[pl->sizes_desc.size() / 2] * 5;
	if (G.split_streams && G.compact_ring && G.ring_class >= 3 && (mixed_sizes || G.split_streams > 1)) {
		if (pl->aux.device < 0) HIP_TRY(aux_acquire(pl->device, &pl->aux));
		L.side = pl->aux.aux[0]; L.ev_fork = pl->aux.fork[0]; L.ev_join = pl->aux.fork[1];   // (aux[0]: the helper stream of highest priority -- the 32-bit instantiations hold the longest tasks)
	}
	HIP_TRY(hipEventRecord(pl->ev_pre, st));
	int nl = 0;
	HIP_TRY(mm2c::launch_chain_dp(L, st, &nl, pl->ev0, &pl->info));
	HIP_TRY(hipEventRecord(pl->ev1, st));
	pl->ran = true;
	G.tasks += (uint64_t)pl->n_tasks; G.anchors += (uint64_t)pl->total; G.launches += (uint64_t)nl;
	return 0;
}

int mm2c_plan_predict_device(mm2c_plan_t *pl, const void *d_anchors, uint8_t *d_num_subparts, int64_t *d_total_subparts,
                             int64_t *d_total_trip_count, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (!lib_ready()) return fail_not_ready();
	if (pl->n_tasks == 0) return 0;
	if (!d_anchors && pl->total > 0) return fail(MM2C_E_ARG, "device pointer is NULL");
	DeviceScope on(pl->device);
	HIP_TRY(on.err);
	hipStream_t st;
	if (const int rc = resolve_stream(stream, pl->device, &st)) return rc;
	HIP_TRY(mm2c::launch_predict(pl->par.max_dist_x, pl->n_tasks, pl->d_off_user ? pl->d_off_user : pl->d_off, pl->d_order, d_anchors, d_num_subparts,
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

int mm2c_plan_run_device_n(mm2c_plan_t *pl, const void *d_anchors, int64_t n_anchors, const float *d_avg_qspan, int64_t n_avg,
                           int32_t *d_f, int64_t n_f, int32_t *d_p, int64_t n_p, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	// the reference refuses a call that does not fit its device buffers (chain_hardware.cpp:34-37); here the caller states what its buffers hold
	if (n_anchors < pl->total || n_f < pl->total || n_p < pl->total || (d_avg_qspan && n_avg < pl->n_tasks))
		return fail(MM2C_E_TOOBIG, "a buffer is shorter than the plan's %lld anchors / %lld tasks (anchors %lld, f %lld, p %lld, avg %lld)",
		            (long long)pl->total, (long long)pl->n_tasks, (long long)n_anchors, (long long)n_f, (long long)n_p, (long long)n_avg);
	return mm2c_plan_run_device(pl, d_anchors, d_avg_qspan, d_f, d_p, stream);
}

int mm2c_plan_last_variant(mm2c_plan_t *pl, char *buf, size_t len)
{
	if (!pl || !buf || len == 0) return fail(MM2C_E_ARG, "NULL argument");
	if (!pl->ran) return fail(MM2C_E_ARG, "plan has not been run");
	format_variant(pl->info, buf, len);
	return 0;
}

int mm2c_plan_last_route(mm2c_plan_t *pl, int64_t *pieces, int64_t *one_wave_pieces, int64_t *coop_pieces)
{
	if (!pl || !pieces || !one_wave_pieces || !coop_pieces) return fail(MM2C_E_ARG, "NULL argument");
	if (!pl->ran) return fail(MM2C_E_ARG, "plan has not been run");
	if (pl->info.route_auto && pl->d_cut) {
		// decided on the device (chain_route): the three count words of the cut arena
		DeviceScope on(pl->device);
		HIP_TRY(on.err);
		HIP_TRY(hipEventSynchronize(pl->ev1));
		int32_t w[4] = {0, 0, 0, 0};
		HIP_TRY(hipMemcpy(w, pl->d_cut, sizeof(w), hipMemcpyDeviceToHost));
		*pieces = w[0]; *one_wave_pieces = w[1]; *coop_pieces = w[2] + w[3];              // (sixteen waves per piece or eight: only one of the two is set)
		return 0;
	}
	*pieces = pl->n_tasks;
	*one_wave_pieces = pl->info.coop ? 0 : pl->n_tasks; *coop_pieces = pl->info.coop ? pl->n_tasks : 0;
	if (pl->info.cut && pl->d_cut) {
		DeviceScope on(pl->device);
		HIP_TRY(on.err);
		HIP_TRY(hipEventSynchronize(pl->ev1));
		int32_t w = 0;
		HIP_TRY(hipMemcpy(&w, pl->d_cut, sizeof(w), hipMemcpyDeviceToHost));
		*pieces = w; *one_wave_pieces = w;
	}
	return 0;
}

int mm2c_route_pieces(int64_t pieces, int64_t longest, int64_t total) { const int w8 = G.coop_w8_above.load(); return mm2c::coop_pays(pieces, longest, total, w8) ? (pieces > w8 ? 8 : 16) : 1; }

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
	if (!lib_ready()) return fail_not_ready();
	if (!d_u_off || !d_b_off) return fail(MM2C_E_ARG, "device pointer is NULL");
	DeviceScope on(pl->device);
	HIP_TRY(on.err);
	hipStream_t st;
	if (const int rc = resolve_stream(stream, pl->device, &st)) return rc;
	if (pl->n_tasks == 0 || pl->total == 0) {
		HIP_TRY(hipMemsetAsync(d_u_off, 0, ((size_t)pl->n_tasks + 1) * 8, st));
		HIP_TRY(hipMemsetAsync(d_b_off, 0, ((size_t)pl->n_tasks + 1) * 8, st));
		return 0;
	}
	if (!d_anchors || !d_f || !d_p || !d_u || !d_b) return fail(MM2C_E_ARG, "device pointer is NULL");
	if (pl->total >= (int64_t)INT32_MAX) return fail(MM2C_E_TOOBIG, "the device epilogue takes batches of fewer than 2^31 anchors (got %lld)", (long long)pl->total);
	mm2c::EpiArgs &E = pl->E;
	if (!pl->d_epi) {
		const size_t tmp = mm2c::epilogue_sort_temp_bytes(pl->total, pl->n_tasks);
		const size_t bytes = layout_epilogue(E, nullptr, (size_t)pl->total, (size_t)pl->n_tasks, tmp);
		HIP_TRY(dev_alloc((void **)&pl->d_epi, bytes));
		layout_epilogue(E, pl->d_epi, (size_t)pl->total, (size_t)pl->n_tasks, tmp);
		HIP_TRY(hipEventCreate(&pl->ev_e0));
		HIP_TRY(hipEventCreate(&pl->ev_e1));
	}
	E.n_tasks = pl->n_tasks; E.total = pl->total; E.d_off = pl->d_off_user ? pl->d_off_user : pl->d_off; E.d_order = pl->d_order;
	E.d_a = (const ulonglong2 *)d_anchors; E.d_f = d_f; E.d_p = d_p; E.min_cnt = min_cnt; E.min_sc = min_sc;
	E.debug_phases = epilogue_debug_phases();
	E.fused = G.epi_fused.load(); E.max_task = pl->sizes_desc.empty() ? -1 : (int64_t)pl->sizes_desc[0];
	E.u_off = d_u_off; E.b_off = d_b_off; E.u_out = d_u; E.b_out = (ulonglong2 *)d_b;
	int nl = 0;
	HIP_TRY(hipEventRecord(pl->ev_e0, st));
	HIP_TRY(mm2c::launch_chain_epilogue(E, st, &nl));
	HIP_TRY(hipEventRecord(pl->ev_e1, st));
	pl->epi_ran = true;
	G.launches += (uint64_t)nl;
	return 0;
}

int mm2c_plan_chains_device_n(mm2c_plan_t *pl, const void *d_anchors, int64_t n_anchors, const int32_t *d_f, int64_t n_f, const int32_t *d_p, int64_t n_p,
                              int min_cnt, int min_sc, int64_t *d_u_off, int64_t n_u_off, uint64_t *d_u, int64_t n_u, int64_t *d_b_off, int64_t n_b_off,
                              void *d_b, int64_t n_b, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (n_anchors < pl->total || n_f < pl->total || n_p < pl->total || n_u < pl->total || n_b < pl->total || n_u_off < pl->n_tasks + 1 || n_b_off < pl->n_tasks + 1)
		return fail(MM2C_E_TOOBIG, "a buffer is shorter than the plan's %lld anchors / %lld tasks", (long long)pl->total, (long long)pl->n_tasks);
	return mm2c_plan_chains_device(pl, d_anchors, d_f, d_p, min_cnt, min_sc, d_u_off, d_u, d_b_off, d_b, stream);
}

int mm2c_plan_last_epilogue_ms(mm2c_plan_t *pl, float *ms)
{
	if (!pl || !ms) return fail(MM2C_E_ARG, "NULL argument");
	if (!pl->epi_ran) return fail(MM2C_E_ARG, "mm2c_plan_chains_device has not been run");
	HIP_TRY(hipEventSynchronize(pl->ev_e1));
	HIP_TRY(hipEventElapsedTime(ms, pl->ev_e0, pl->ev_e1));
	return 0;
}

void *mm2c_pinned_alloc(size_t bytes)
{
	void *p = nullptr;
	if (!lib_ready()) { fail_not_ready(); return nullptr; }
	DeviceScope on(cur_device());
	ScopedNs timed(SS.alloc_ns); ++SS.n_alloc;
	if (on.err != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
		fail(MM2C_E_HIP, "hipHostMalloc(%zu) failed", bytes);
		return nullptr;
	}
	return p;
}

void mm2c_pinned_free(void *ptr) { if (ptr) { ScopedNs timed(SS.free_ns); ++SS.n_free; (void)hipHostFree(ptr); } }

} // extern "C"

namespace mm2c_api { void plan_destroy_synced(mm2c_plan_t *pl) { plan_destroy_impl(pl, false); } }
