// mm2chain_seeds.cpp -- C-ABI entries of the seed-hit path (include/mm2chain.h): seed plans (matches -> sorted anchors on the device,
// collect_seed_hits map.c:215-247) and the host-buffer entries built on them.
#include "api_internal.h"

using namespace mm2c_api;

struct mm2c_seedplan {
	int64_t n_reads = 0, total = 0, n_matches = 0;
	int device = 0;
	int32_t *d_cnt = nullptr; int64_t *d_oo = nullptr;   // per-read anchor counts / packed offsets of runs with skip_seed
	const mm2c_seed_skip_t *skip = nullptr;                // set by mm2c_seedplan_run_device_skip for the run it starts
	int64_t n_hits_declared = 0;           // set by mm2c_seedplan_run_device_n for the run it starts
	char *d_mem = nullptr;                 // [match_off | anchor_off | order | status | has_ties | stack | unsorted | scratch | tie_id | big_dg]
	mm2c::SeedArgs S;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	hipStream_t aux[3] = {};               // helper streams: the size classes of the tie replay run side by side
	hipEvent_t fork[4] = {};
	bool has_aux = false;                  // aux / fork come from (and go back to) the pool of helper stream sets
	bool ran = false;
};

// helper streams and fork / join events of the seed plans, kept between plans: a pipelined batch makes a plan per chunk, and creating three
// streams of distinct priority per plan cost milliseconds each
namespace {
std::mutex g_aux_mu;
std::vector<AuxSet> g_aux_free;
}

namespace mm2c_api {
hipError_t aux_acquire(int device, AuxSet *out)
{
	uint64_t epoch_now;
	{
		// the library's epoch is written by mm2c_init / mm2c_shutdown under G.mu: read it under that lock, then the pool's (the order shutdown takes them in)
		std::lock_guard<std::mutex> gl(G.mu);
		epoch_now = G.epoch;
		std::lock_guard<std::mutex> lk(g_aux_mu);
		for (size_t i = 0; i < g_aux_free.size(); ++i)
			if (g_aux_free[i].device == device && g_aux_free[i].epoch == epoch_now) { *out = g_aux_free[i]; g_aux_free.erase(g_aux_free.begin() + (long)i); return hipSuccess; }
	}
	AuxSet a; a.device = device; a.epoch = epoch_now;
	hipError_t e = hipSuccess;
	// helper streams on hardware queues of their own: different priorities never share a queue (see create_partner_stream)
	int least = 0, greatest = 0;
	if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
	const int prio[3] = { greatest, least, (least + greatest) / 2 };
	for (int i = 0; i < 3 && e == hipSuccess; ++i)
		e = least != greatest ? hipStreamCreateWithPriority(&a.aux[i], hipStreamNonBlocking, prio[i]) : hipStreamCreateWithFlags(&a.aux[i], hipStreamNonBlocking);
	for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&a.fork[i], hipEventDisableTiming);
	if (e != hipSuccess) {
		for (auto &st : a.aux) if (st) (void)hipStreamDestroy(st);
		for (auto &ev : a.fork) if (ev) (void)hipEventDestroy(ev);
		return e;
	}
	*out = a;
	return hipSuccess;
}

static void aux_destroy(const AuxSet &a)
{
	DeviceScope on(a.device);
	for (auto &st : a.aux) if (st) (void)hipStreamDestroy(st);
	for (auto &ev : a.fork) if (ev) (void)hipEventDestroy(ev);
}

void aux_release(const AuxSet &a)
{
	if (a.device < 0) return;
	// a plan that outlived mm2c_shutdown() brings back handles of the library's previous life: they are destroyed here, never handed to a plan of the next one
	// (decided under the library's lock: a shutdown or re-initialisation running beside this release either sees the set in the pool or comes after it was destroyed)
	std::lock_guard<std::mutex> gl(G.mu);
	if (a.epoch != G.epoch || !G.ready) { aux_destroy(a); return; }
	std::lock_guard<std::mutex> lk(g_aux_mu);
	g_aux_free.push_back(a);
}

void release_seed_aux()                       // mm2c_shutdown
{
	std::lock_guard<std::mutex> lk(g_aux_mu);
	for (AuxSet &a : g_aux_free) aux_destroy(a);
	g_aux_free.clear();
}
}

extern "C" {

mm2c_seedplan_t *mm2c_seedplan_create(int64_t n_reads, const int64_t *h_match_off, const int64_t *h_anchor_off)
{
	if (!lib_ready()) { fail_not_ready(); return nullptr; }
	if (n_reads < 0 || n_reads > INT32_MAX || (n_reads > 0 && (!h_match_off || !h_anchor_off))) { fail(MM2C_E_ARG, "bad argument"); return nullptr; }
	std::vector<int32_t> order;
	if (build_order(n_reads, h_anchor_off, order)) return nullptr;            // validates the anchor offsets; biggest read first
	int64_t biggest = 0;
	for (int64_t r = 0; r < n_reads; ++r) {
		if (h_match_off[r + 1] < h_match_off[r]) { fail(MM2C_E_ARG, "match offsets not monotone at read %lld", (long long)r); return nullptr; }
		biggest = std::max(biggest, h_anchor_off[r + 1] - h_anchor_off[r]);
	}
	mm2c_seedplan *pl = new mm2c_seedplan();
	pl->S.heap_order = G.heap_sort ? 1 : 0;              // the process-wide default (mm2c_tune("heap_sort")): the host-batch entries make their seed plans themselves
	pl->n_reads = n_reads;
	pl->total = n_reads ? h_anchor_off[n_reads] - h_anchor_off[0] : 0;
	pl->n_matches = n_reads ? h_match_off[n_reads] - h_match_off[0] : 0;
	const size_t nr = (size_t)std::max<int64_t>(n_reads, 1), tot = (size_t)std::max<int64_t>(pl->total, 1);
	// The tie replay of a read of 12 289 .. 131 072 anchors runs on eight waves with its digits in LDS -- one read per CU (up to 141 KB) --, which is the way for the few
	// long reads of a mapping batch; a batch that is MADE of such reads (an all-vs-all chunk: thousands of them) is better off with one wave per read and the digits in
	// memory, eight and more reads in flight per CU: 2 048 reads of 10^5 anchors 97 -> 44 ms (round 6, profiles/r6_long_reads.md).
	int64_t n_mid = 0;
	for (int64_t r = 0; r < n_reads; ++r) { const int64_t cap = h_anchor_off[r + 1] - h_anchor_off[r]; n_mid += cap > mm2c::seed_tie_mid_lower() && cap <= mm2c::seed_tie_lds_max(); }
	int64_t tie_global_above = mm2c::seed_tie_lds_max();
	{ const char *tg = getenv("MM2C_TIE_GLOBAL_ABOVE"); if (tg) tie_global_above = std::max(64, atoi(tg)); else if (n_mid > 512) tie_global_above = mm2c::seed_tie_mid_lower(); }
	const bool big = biggest > tie_global_above;
	size_t at = 0;
	auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return o; };
	const size_t o_moff = take((nr + 1) * 8), o_aoff = take((nr + 1) * 8), o_ord = take(nr * 4), o_stat = take(nr * 4), o_ties = take(nr * 4),
	             o_stack = take(4 * (tot / 64 + 2 * nr + 2) * 4), o_un = take(tot * 16), o_scr = take(tot * 16), o_tc = take(tot * 4), o_xd = take(nr * 8),
	             o_bid = take(tot * 4), o_bdg = take(big ? tot + nr + 64 : 1), o_cnt = take(nr * 4), o_oo = take((nr + 1) * 8);
	pl->device = cur_device();
	DeviceScope on(pl->device);
	hipError_t e = on.err;
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
	if (e == hipSuccess) {
		AuxSet a;
		e = aux_acquire(pl->device, &a);
		if (e == hipSuccess) { for (int i = 0; i < 3; ++i) pl->aux[i] = a.aux[i]; for (int i = 0; i < 4; ++i) pl->fork[i] = a.fork[i]; pl->has_aux = true; }
	}
	if (e != hipSuccess) { fail(MM2C_E_HIP, "mm2c_seedplan_create: %s", hipGetErrorString(e)); mm2c_seedplan_destroy(pl); return nullptr; }
	mm2c::SeedArgs &S = pl->S;
	char *b = pl->d_mem;
	S.n_reads = n_reads; S.d_match_off = (const int64_t *)(b + o_moff); S.d_anchor_off = (const int64_t *)(b + o_aoff);
	S.d_order = (const int32_t *)(b + o_ord); S.status = (int32_t *)(b + o_stat); S.has_ties = (int32_t *)(b + o_ties);
	S.tiecnt = (int32_t *)(b + o_tc); S.xdiff = (uint64_t *)(b + o_xd); S.biggest = biggest;
	S.stack = (int32_t *)(b + o_stack); S.unsorted = (ulonglong2 *)(b + o_un); S.scratch = (ulonglong2 *)(b + o_scr);
	{ const char *cut = getenv("MM2C_TIE_CUT"); S.debug_cut = cut ? atoi(cut) : 0; }
	S.tie_id = (uint32_t *)(b + o_bid); S.big_dg = big ? (uint8_t *)(b + o_bdg) : nullptr;
	{
		const int64_t *lower = mm2c::seed_tie_class_lower();                     // a read keeps at most its capacity: these bound the grids of the classes
		const int64_t sort_cap = mm2c::seed_sort_lds_cap0(), sort_cap2 = mm2c::seed_sort_lds_cap();
		S.tie_global_above = tie_global_above;
		for (int64_t r = 0; r < n_reads; ++r) { const int64_t cap = h_anchor_off[r + 1] - h_anchor_off[r]; for (int k = 0; k < 5; ++k) S.n_above[k] += cap > lower[k]; S.n_above[5] += cap > tie_global_above; S.n_sort_big += cap > sort_cap; S.n_sort_huge += cap > sort_cap2; }
		{ const char *ls = getenv("MM2C_LDS_SORT"); if (ls) S.lds_sort = atoi(ls) != 0; }
		{ const char *ms = getenv("MM2C_MW_SORT"); if (ms) S.mw_sort = atoi(ms) != 0; }
		{ const char *tm = getenv("MM2C_TIE_GLOBAL_MW_BELOW"); if (tm) S.tie_global_mw_below = atoi(tm); }
		{ const char *tw = getenv("MM2C_TIE_GLOBAL_WAVES"); if (tw) S.tie_global_waves = atoi(tw); }
	}
	pl->d_cnt = (int32_t *)(b + o_cnt); pl->d_oo = (int64_t *)(b + o_oo);
	return pl;
}

static void seedplan_destroy_impl(mm2c_seedplan_t *pl, bool wait)
{
	if (!pl) return;
	{
		DeviceScope on(pl->device);
		if (wait && pl->ran) { ScopedNs timed(SS.free_ns); (void)hipDeviceSynchronize(); }
		dev_free_synced(pl->d_mem);
		if (pl->ev0) (void)hipEventDestroy(pl->ev0); if (pl->ev1) (void)hipEventDestroy(pl->ev1);
		if (pl->has_aux) {
			// the helper streams join the plan's stream at the end of every run, so once the caller's stream (or the device) has been waited for they are idle
			AuxSet a; a.device = pl->device;
			for (int i = 0; i < 3; ++i) a.aux[i] = pl->aux[i];
			for (int i = 0; i < 4; ++i) a.fork[i] = pl->fork[i];
			aux_release(a);
		}
	}
	delete pl;
}

void mm2c_seedplan_destroy(mm2c_seedplan_t *pl) { seedplan_destroy_impl(pl, true); }

int mm2c_seedplan_run_device(mm2c_seedplan_t *pl, const mm2c_match_t *d_matches, const uint64_t *d_hits, const int32_t *d_qlen,
                             void *d_anchors, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (!lib_ready()) return fail_not_ready();
	if (pl->n_reads == 0) return 0;
	if (!d_qlen || (pl->n_matches > 0 && !d_matches) || (pl->total > 0 && (!d_hits || !d_anchors))) return fail(MM2C_E_ARG, "device pointer is NULL");
	static_assert(sizeof(mm2c_match_t) == sizeof(mm2c::Match), "mm2c_match_t layout");
	DeviceScope on(pl->device);
	HIP_TRY(on.err);
	hipStream_t st;
	if (const int rc = resolve_stream(stream, pl->device, &st)) return rc;
	mm2c::SeedArgs &S = pl->S;
	S.d_matches = (const mm2c::Match *)d_matches; S.d_hits = d_hits; S.d_qlen = d_qlen; S.d_anchors = (ulonglong2 *)d_anchors;
	S.n_hits = pl->n_hits_declared; pl->n_hits_declared = 0;
	if (pl->skip) {
		S.skip_flag = pl->skip->flag; S.d_ref_rank = pl->skip->d_ref_rank; S.d_ref_len = pl->skip->d_ref_len; S.d_q_lo = pl->skip->d_q_lo; S.d_q_eq = pl->skip->d_q_eq;
		S.d_count = pl->d_cnt; S.d_out_off = pl->d_oo;
		pl->skip = nullptr;
	} else { S.skip_flag = 0; S.d_ref_rank = S.d_ref_len = S.d_q_lo = S.d_q_eq = nullptr; S.d_count = nullptr; S.d_out_off = nullptr; }
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

int mm2c_seedplan_set_heap_sort(mm2c_seedplan_t *pl, int on)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	pl->S.heap_order = on ? 1 : 0;
	return 0;
}

int mm2c_seedplan_run_device_n(mm2c_seedplan_t *pl, const mm2c_match_t *d_matches, int64_t n_matches, const uint64_t *d_hits, int64_t n_hits,
                               const int32_t *d_qlen, int64_t n_qlen, void *d_anchors, int64_t n_anchors, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (n_matches < pl->n_matches || n_qlen < pl->n_reads || n_anchors < pl->total || n_hits < 0)
		return fail(MM2C_E_TOOBIG, "a buffer is shorter than the plan needs (matches %lld of %lld, qlen %lld of %lld, anchors %lld of %lld)",
		            (long long)n_matches, (long long)pl->n_matches, (long long)n_qlen, (long long)pl->n_reads, (long long)n_anchors, (long long)pl->total);
	if (pl->total > 0 && n_hits == 0) return fail(MM2C_E_TOOBIG, "the hit pool is empty but the plan expands %lld hits", (long long)pl->total);
	pl->n_hits_declared = n_hits;                  // checked per match on the device (mm2c_seedplan_check reports it)
	return mm2c_seedplan_run_device(pl, d_matches, d_hits, d_qlen, d_anchors, stream);
}

int mm2c_seedplan_run_device_skip(mm2c_seedplan_t *pl, const mm2c_match_t *d_matches, int64_t n_matches, const uint64_t *d_hits, int64_t n_hits,
                                  const int32_t *d_qlen, int64_t n_qlen, const mm2c_seed_skip_t *skip, void *d_anchors, int64_t n_anchors,
                                  int64_t *d_anchor_off_out, void *stream)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (!skip || !d_anchor_off_out) return fail(MM2C_E_ARG, "skip description / offset output is NULL");
	if ((skip->flag & (0x001 | 0x002)) && skip->d_ref_rank && (!skip->d_ref_len || !skip->d_q_lo || !skip->d_q_eq))
		return fail(MM2C_E_ARG, "NO_DIAG / NO_DUAL need ref_rank, ref_len, q_lo and q_eq");
	pl->skip = skip;
	int rc = mm2c_seedplan_run_device_n(pl, d_matches, n_matches, d_hits, n_hits, d_qlen, n_qlen, d_anchors, n_anchors, stream);
	pl->skip = nullptr;
	if (rc != 0 || pl->n_reads == 0) return rc;
	DeviceScope on(pl->device);
	HIP_TRY(on.err);
	hipStream_t st;
	if (const int rc2 = resolve_stream(stream, pl->device, &st)) return rc2;
	HIP_TRY(hipMemcpyAsync(d_anchor_off_out, pl->d_oo, ((size_t)pl->n_reads + 1) * 8, hipMemcpyDeviceToDevice, st));
	return 0;
}

int mm2c_seedplan_check(mm2c_seedplan_t *pl, int64_t *n_reads_with_ties)
{
	if (!pl) return fail(MM2C_E_ARG, "plan is NULL");
	if (n_reads_with_ties) *n_reads_with_ties = 0;
	if (!pl->ran || pl->n_reads == 0) return 0;
	DeviceScope on(pl->device);
	HIP_TRY(on.err);
	HIP_TRY(hipEventSynchronize(pl->ev1));
	std::vector<int32_t> st((size_t)pl->n_reads), ti((size_t)pl->n_reads);
	HIP_TRY(hipMemcpy(st.data(), pl->S.status, st.size() * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(ti.data(), pl->S.has_ties, ti.size() * 4, hipMemcpyDeviceToHost));
	int64_t nt = 0;
	for (size_t r = 0; r < st.size(); ++r) {
		if (st[r] == 2) return fail(MM2C_E_ARG, "read %zu: a match points outside the declared hit pool", r);
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

// ---- the index's position arrays resident on the device(s)
struct mm2c_hitpool {
	int64_t n = 0;
	int n_dev = 0;
	int dev[64] = {};
	uint64_t *d[64] = {};                       // one copy per device the library drives (a split batch reads the copy of its own device)
};

mm2c_hitpool_t *mm2c_hitpool_create(const uint64_t *h_hits, int64_t n_hits)
{
	if (!lib_ready()) { fail_not_ready(); return nullptr; }
	if (n_hits < 0 || (n_hits > 0 && !h_hits)) { fail(MM2C_E_ARG, "bad argument"); return nullptr; }
	mm2c_hitpool *hp = new mm2c_hitpool();
	hp->n = n_hits;
	for (size_t k = 0; k < G.devices.size() && hp->n_dev < 64; ++k) {
		const int dv = G.devices[k];
		bool seen = false;
		for (int j = 0; j < hp->n_dev; ++j) seen = seen || hp->dev[j] == dv;
		if (seen) continue;
		DeviceScope on(dv);
		void *p = nullptr;
		hipError_t e = on.err;
		{ ScopedNs timed(SS.alloc_ns); ++SS.n_alloc; if (e == hipSuccess) e = hipMalloc(&p, (size_t)std::max<int64_t>(n_hits, 1) * 8); }
		if (e == hipSuccess && n_hits > 0) e = hipMemcpy(p, h_hits, (size_t)n_hits * 8, hipMemcpyHostToDevice);
		if (e != hipSuccess) { if (p) (void)hipFree(p); fail(MM2C_E_HIP, "mm2c_hitpool_create: %s", hipGetErrorString(e)); mm2c_hitpool_destroy(hp); return nullptr; }
		hp->dev[hp->n_dev] = dv; hp->d[hp->n_dev] = (uint64_t *)p; ++hp->n_dev;
	}
	return hp;
}

int64_t mm2c_hitpool_size(const mm2c_hitpool_t *hp) { return hp ? hp->n : 0; }

void mm2c_hitpool_destroy(mm2c_hitpool_t *hp)
{
	if (!hp) return;
	for (int j = 0; j < hp->n_dev; ++j) { DeviceScope on(hp->dev[j]); ScopedNs timed(SS.free_ns); ++SS.n_free; (void)hipDeviceSynchronize(); (void)hipFree(hp->d[j]); }
	delete hp;
}

static const uint64_t *pool_on(const mm2c_hitpool_t *hp, int device)
{
	for (int j = 0; j < hp->n_dev; ++j) if (hp->dev[j] == device) return hp->d[j];
	return nullptr;
}

// matches in, chains out: collect_seed_hits + mm_chain_dp for a batch of reads (map.c:295-316) without the anchors leaving the GPU.
// Big batches run in chunks of whole reads on two streams of different priority (= two hardware queues): while chunk k computes, chunk k+1
// uploads and the chains of chunk k-1 download.  Per chunk: a seed plan and a chain plan (workspace from the device cache: chunks are of
// similar size, so after the first ones nothing is allocated) and one grow-only arena per slot.  hits come from the caller's host pool
// (only the range the chunk's matches point into is uploaded; a pool that is shared by all chunks is uploaded once) or from a resident pool.
static int seed_chain_impl(const mm2c_params_t *par, int min_cnt, int min_sc, int64_t n_reads, const int64_t *h_match_off,
                           const mm2c_match_t *h_matches, const uint64_t *h_hits, int64_t n_hits, const mm2c_hitpool_t *pool, const int32_t *h_qlen,
                           int64_t *anchor_off, int64_t *u_off, uint64_t *u, int64_t *b_off, mm2c_anchor_t *b)
{
	int rc;
	ScopedNs timed_total(SS.total_ns);
	if ((rc = check_params(par))) return rc;
	if (n_reads < 0 || !anchor_off || !u_off || !b_off) return fail(MM2C_E_ARG, "bad argument");
	anchor_off[0] = 0; u_off[0] = b_off[0] = 0;
	if (n_reads == 0) return 0;
	if (!h_match_off || !h_qlen) return fail(MM2C_E_ARG, "host pointer is NULL");
	if (pool) n_hits = pool->n;
	const int64_t mb = h_match_off[0], n_m = h_match_off[n_reads] - mb;
	if (n_m > 0 && !h_matches) return fail(MM2C_E_ARG, "matches is NULL");
	std::vector<int64_t> cr_lo, cr_hi;                      // per read: the range of the hit pool its matches point into
	{
		ScopedNs timed(SS.setup_ns);
		cr_lo.resize((size_t)n_reads); cr_hi.resize((size_t)n_reads);
		for (int64_t r = 0; r < n_reads; ++r) {
			int64_t sum = 0, lo = INT64_MAX, hi = 0;
			if (h_match_off[r + 1] < h_match_off[r]) return fail(MM2C_E_ARG, "match offsets not monotone at read %lld", (long long)r);
			for (int64_t i = h_match_off[r]; i < h_match_off[r + 1]; ++i) {
				const int64_t c0 = h_matches[i].cr_off, c1 = c0 + (int64_t)h_matches[i].n;
				if (c0 < 0 || c1 > n_hits) return fail(MM2C_E_ARG, "match %lld reaches beyond the hit pool", (long long)i);
				sum += h_matches[i].n;
				if (h_matches[i].n) { lo = std::min(lo, c0); hi = std::max(hi, c1); }
			}
			cr_lo[(size_t)r] = lo; cr_hi[(size_t)r] = hi;
			anchor_off[r + 1] = anchor_off[r] + sum;
		}
	}
	const int64_t total = anchor_off[n_reads];
	if (total == 0) { for (int64_t r = 1; r <= n_reads; ++r) u_off[r] = b_off[r] = 0; return 0; }
	if ((!pool && !h_hits) || !u || !b) return fail(MM2C_E_ARG, "host pointer is NULL");
	if (should_split(total)) {
		// several devices: a contiguous range of reads each (about equal anchor counts), results closed up afterwards as in mm2c_mm_chain_dp_batch_host
		const int nd = n_devices();
		std::vector<std::vector<int64_t>> ao((size_t)nd), uo((size_t)nd), bo((size_t)nd);
		std::vector<int64_t> r0((size_t)nd, 0), r1((size_t)nd, 0);
		rc = run_split(n_reads, anchor_off, [&](int part, int64_t k0, int64_t k1) {
			const size_t m = (size_t)(k1 - k0) + 1;
			ao[(size_t)part].assign(m, 0); uo[(size_t)part].assign(m, 0); bo[(size_t)part].assign(m, 0);
			r0[(size_t)part] = k0; r1[(size_t)part] = k1;
			const int64_t at = anchor_off[k0];
			return seed_chain_impl(par, min_cnt, min_sc, k1 - k0, h_match_off + k0, h_matches, h_hits, n_hits, pool, h_qlen + k0,
			                       ao[(size_t)part].data(), uo[(size_t)part].data(), u + at, bo[(size_t)part].data(), b + at);
		});
		if (rc != 0) return rc;
		int64_t U = 0, B = 0;
		for (int part = 0; part < nd; ++part) {
			const int64_t k0 = r0[(size_t)part], k1 = r1[(size_t)part];
			if (k1 == k0) continue;
			const int64_t at = anchor_off[k0], nu = uo[(size_t)part].back(), nb = bo[(size_t)part].back();
			if (U != at) memmove(u + U, u + at, (size_t)nu * 8);
			if (B != at) memmove(b + B, b + at, (size_t)nb * 16);
			for (int64_t k = k0; k < k1; ++k) { u_off[k + 1] = U + uo[(size_t)part][(size_t)(k - k0) + 1]; b_off[k + 1] = B + bo[(size_t)part][(size_t)(k - k0) + 1]; }
			U += nu; B += nb;
		}
		return 0;
	}
	++SS.calls;
	ThreadCtx *c;
	std::unique_lock<std::mutex> hold;                              // batch calls from different host threads take turns on one set of arenas
	if ((rc = get_batch_ctx(&c, hold))) return rc;
	const int device = cur_device();
	DeviceScope on(device);
	HIP_TRY(on.err);
	const uint64_t *d_pool = nullptr;
	if (pool && !(d_pool = pool_on(pool, device))) return fail(MM2C_E_ARG, "the hit pool has no copy on device %d (created before mm2c_init_devices?)", device);
	// chunks of whole reads
	const int64_t chunk_anchors = total >= 2 * G.pipeline_chunk_anchors ? G.pipeline_chunk_anchors.load() : std::min<int64_t>(total, (int64_t)INT32_MAX - 1);
	std::vector<int64_t> cuts(1, 0);
	for (int64_t k0 = 0; k0 < n_reads;) {
		int64_t k1 = k0 + 1;
		while (k1 < n_reads && anchor_off[k1 + 1] - anchor_off[k0] <= chunk_anchors) ++k1;
		if (anchor_off[k1] - anchor_off[k0] >= (int64_t)INT32_MAX) return fail(MM2C_E_TOOBIG, "a read with 2^31 anchors or more");
		cuts.push_back(k1); k0 = k1;
	}
	const int n_chunks = (int)cuts.size() - 1;
	// hits of a chunk = the range of the host pool its matches point into; when those ranges add up to much more than the pool (matches that
	// point all over an index-wide pool) the whole pool goes up once instead
	std::vector<int64_t> h_lo((size_t)n_chunks, 0), h_hi((size_t)n_chunks, 0);
	bool whole_pool = false;
	char *d_whole = nullptr;
	if (!pool) {
		int64_t span = 0;
		for (int ck = 0; ck < n_chunks; ++ck) {
			int64_t lo = INT64_MAX, hi = 0;
			for (int64_t r = cuts[(size_t)ck]; r < cuts[(size_t)ck + 1]; ++r) { lo = std::min(lo, cr_lo[(size_t)r]); hi = std::max(hi, cr_hi[(size_t)r]); }
			if (hi <= lo) lo = hi = 0;
			h_lo[(size_t)ck] = lo; h_hi[(size_t)ck] = hi; span += hi - lo;
		}
		whole_pool = n_chunks > 1 && span > n_hits + n_hits / 2;
	}
	int64_t base_u = 0, base_b = 0;
	int nl = 0;

	auto drop_plans = [&](SeedSlot &w, bool waited) {           // the chunk's kernels are done (waited) or the call failed (wait inside)
		if (w.seedplan) { if (waited) seedplan_destroy_synced((mm2c_seedplan_t *)w.seedplan); else mm2c_seedplan_destroy((mm2c_seedplan_t *)w.seedplan); w.seedplan = nullptr; }
		if (w.plan) { if (waited) plan_destroy_synced((mm2c_plan_t *)w.plan); else mm2c_plan_destroy((mm2c_plan_t *)w.plan); w.plan = nullptr; }
	};
	auto enqueue = [&](SeedSlot &w, int ck) -> int {
		int r;
		const int64_t k0 = cuts[(size_t)ck], k1 = cuts[(size_t)ck + 1];
		const size_t nr = (size_t)(k1 - k0), tot = (size_t)(anchor_off[k1] - anchor_off[k0]);
		const int64_t m0 = h_match_off[k0], m1 = h_match_off[k1];
		const size_t nm = (size_t)(m1 - m0), nh = pool || whole_pool ? 0 : (size_t)(h_hi[(size_t)ck] - h_lo[(size_t)ck]);
		if (!w.st) {
			HIP_TRY(&w == &c->seed[1] ? create_partner_stream(&w.st) : hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking));
			for (auto &e : w.ev) HIP_TRY(hipEventCreate(&e));
		}
		w.k0 = k0; w.k1 = k1; w.busy = true; w.timed = tot > 0;
		if (tot == 0) return 0;
		{
			ScopedNs timed(SS.setup_ns);
			w.seedplan = mm2c_seedplan_create((int64_t)nr, h_match_off + k0, anchor_off + k0);
			if (!w.seedplan) return MM2C_E_HIP;
			w.plan = mm2c_plan_create(par, (int64_t)nr, anchor_off + k0);
			if (!w.plan) return MM2C_E_HIP;
		}
		size_t at = 0;
		auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return o; };
		const size_t o_m = take(nm * sizeof(mm2c_match_t)), o_h = take(nh * 8), o_q = take(nr * 4), o_a = take(tot * 16), o_f = take(tot * 4), o_p = take(tot * 4);
		w.o_uo = take((nr + 1) * 8); w.o_bo = take((nr + 1) * 8); w.o_u = take(tot * 8); w.o_b = take(tot * 16);
		if (at > w.cap_buf || 2 * (nr + 1) * 8 > w.cap_hmeta) {      // the arena moves: the chains of the slot's last chunk may still be on their way out of it
			ScopedNs timed(SS.wait_ns);
			HIP_TRY(hipStreamSynchronize(w.st));
		}
		if ((r = grow_device(&w.d_buf, &w.cap_buf, at + at / 8))) return r;   // (some headroom: chunks differ by a read or two)
		if ((r = grow_pinned(&w.h_meta, &w.cap_hmeta, 2 * (nr + 1) * 8 + 4096))) return r;
		char *d = w.d_buf;
		hipStream_t st = w.st;
		HIP_TRY(hipEventRecord(w.ev[0], st));
		HIP_TRY(hipMemcpyAsync(d + o_m, h_matches + m0, nm * sizeof(mm2c_match_t), hipMemcpyHostToDevice, st));
		if (nh) HIP_TRY(hipMemcpyAsync(d + o_h, h_hits + h_lo[(size_t)ck], nh * 8, hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(d + o_q, h_qlen + k0, nr * 4, hipMemcpyHostToDevice, st));
		HIP_TRY(hipEventRecord(w.ev[1], st));
		// the matches keep their pool-relative cr_off: the kernels get the pool's base, i.e. the chunk's range shifted back by its start
		const uint64_t *d_hits = pool ? d_pool : whole_pool ? (const uint64_t *)d_whole : (const uint64_t *)(d + o_h) - h_lo[(size_t)ck];
		if ((r = mm2c_seedplan_run_device((mm2c_seedplan_t *)w.seedplan, (const mm2c_match_t *)(d + o_m), d_hits, (const int32_t *)(d + o_q), d + o_a, st))) return r;
		HIP_TRY(hipEventRecord(w.ev[2], st));
		if ((r = mm2c_plan_run_device((mm2c_plan_t *)w.plan, d + o_a, nullptr, (int32_t *)(d + o_f), (int32_t *)(d + o_p), st))) return r;
		HIP_TRY(hipEventRecord(w.ev[3], st));
		if ((r = mm2c_plan_chains_device((mm2c_plan_t *)w.plan, d + o_a, (int32_t *)(d + o_f), (int32_t *)(d + o_p), min_cnt, min_sc, (int64_t *)(d + w.o_uo),
		                                 (uint64_t *)(d + w.o_u), (int64_t *)(d + w.o_bo), d + w.o_b, st))) return r;
		HIP_TRY(hipEventRecord(w.ev[4], st));
		HIP_TRY(hipMemcpyAsync(w.h_meta, d + w.o_uo, (nr + 1) * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(w.h_meta + (nr + 1) * 8, d + w.o_bo, (nr + 1) * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipEventRecord(w.ev[5], st));
		++SS.chunks;
		return 0;
	};
	// waits for the chunk's offsets, places them behind the chunks before it and starts the download of its chains
	auto finalize = [&](SeedSlot &w) -> int {
		if (!w.busy) return 0;
		w.busy = false;
		const size_t nr = (size_t)(w.k1 - w.k0);
		if (!w.timed) { for (size_t k = 1; k <= nr; ++k) { u_off[w.k0 + (int64_t)k] = base_u; b_off[w.k0 + (int64_t)k] = base_b; } return 0; }
		{ ScopedNs timed(SS.wait_ns); HIP_TRY(hipEventSynchronize(w.ev[5])); }
		float ms[5] = {};
		for (int k = 0; k < 5; ++k) (void)hipEventElapsedTime(&ms[k], w.ev[k], w.ev[k + 1]);
		SS.h2d_ns += (uint64_t)(ms[0] * 1e6f); SS.seed_ns += (uint64_t)(ms[1] * 1e6f); SS.dp_ns += (uint64_t)(ms[2] * 1e6f); SS.epi_ns += (uint64_t)(ms[3] * 1e6f);
		SS.d2h_ns += (uint64_t)(ms[4] * 1e6f);
		{	// every kernel of the chunk is done: its plans go back to the device cache now (only the chains, in the slot's arena, are still to leave)
			const int r = mm2c_seedplan_check((mm2c_seedplan_t *)w.seedplan, nullptr);
			drop_plans(w, true);
			if (r) return r;
		}
		const int64_t *cu = (const int64_t *)w.h_meta, *cb = cu + nr + 1;
		for (size_t k = 1; k <= nr; ++k) { u_off[w.k0 + (int64_t)k] = base_u + cu[k]; b_off[w.k0 + (int64_t)k] = base_b + cb[k]; }
		if (cu[nr] > 0) HIP_TRY(hipMemcpyAsync(u + base_u, w.d_buf + w.o_u, (size_t)cu[nr] * 8, hipMemcpyDeviceToHost, w.st));
		if (cb[nr] > 0) HIP_TRY(hipMemcpyAsync(b + base_b, w.d_buf + w.o_b, (size_t)cb[nr] * 16, hipMemcpyDeviceToHost, w.st));
		base_u += cu[nr]; base_b += cb[nr];
		return 0;
	};
	rc = 0;
	if (whole_pool) {
		hipError_t e = dev_alloc((void **)&d_whole, (size_t)n_hits * 8);
		if (e == hipSuccess) e = hipMemcpy(d_whole, h_hits, (size_t)n_hits * 8, hipMemcpyHostToDevice);   // before either stream starts
		if (e != hipSuccess) rc = fail(MM2C_E_HIP, "uploading the hit pool: %s", hipGetErrorString(e));
	}
	for (int ck = 0; ck < n_chunks && rc == 0; ++ck) {
		SeedSlot &w = c->seed[ck & 1];
		if ((rc = finalize(w))) break;                              // the chunk before the previous one (same slot): in chunk order
		rc = enqueue(w, ck);
	}
	if (rc == 0) rc = finalize(c->seed[n_chunks & 1]);              // in chunk order: the older slot first
	if (rc == 0) rc = finalize(c->seed[(n_chunks + 1) & 1]);
	for (int i = 0; i < 2; ++i) {                                   // until the chains have landed in the caller's arrays
		SeedSlot &w = c->seed[i];
		if (w.st) { ScopedNs timed(SS.wait_ns); hipError_t e = hipStreamSynchronize(w.st); if (e != hipSuccess && rc == 0) rc = fail(MM2C_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e)); }
		drop_plans(w, false);                                       // (only after a failure is anything left here)
		w.busy = false;
	}
	if (d_whole) dev_free(d_whole);
	G.tasks += (uint64_t)n_reads; G.anchors += (uint64_t)total; G.launches += (uint64_t)nl; G.passes += (uint64_t)n_chunks;
	return rc;
}

int mm2c_seed_chain_batch_host(const mm2c_params_t *par, int min_cnt, int min_sc, int64_t n_reads, const int64_t *h_match_off,
                               const mm2c_match_t *h_matches, const uint64_t *h_hits, int64_t n_hits, const int32_t *h_qlen,
                               int64_t *anchor_off, int64_t *u_off, uint64_t *u, int64_t *b_off, mm2c_anchor_t *b)
{
	return seed_chain_impl(par, min_cnt, min_sc, n_reads, h_match_off, h_matches, h_hits, n_hits, nullptr, h_qlen, anchor_off, u_off, u, b_off, b);
}

int mm2c_seed_chain_batch_pool(const mm2c_params_t *par, int min_cnt, int min_sc, int64_t n_reads, const int64_t *h_match_off,
                               const mm2c_match_t *h_matches, const mm2c_hitpool_t *pool, const int32_t *h_qlen,
                               int64_t *anchor_off, int64_t *u_off, uint64_t *u, int64_t *b_off, mm2c_anchor_t *b)
{
	if (!pool) return fail(MM2C_E_ARG, "pool is NULL");
	return seed_chain_impl(par, min_cnt, min_sc, n_reads, h_match_off, h_matches, nullptr, 0, pool, h_qlen, anchor_off, u_off, u, b_off, b);
}

} // extern "C"

namespace mm2c_api { void seedplan_destroy_synced(mm2c_seedplan_t *pl) { seedplan_destroy_impl(pl, false); } }
