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
	             o_stack = take(4 * (tot / 64 + 2 * nr + 2) * 4), o_un = take(tot * 16), o_scr = take(tot * 16), o_tc = take(tot * 4), o_xd = take(nr * 8),
	             o_bid = take(big ? tot * 4 : 1), o_bdg = take(big ? tot : 1), o_cnt = take(nr * 4), o_oo = take((nr + 1) * 8);
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
	pl->d_cnt = (int32_t *)(b + o_cnt); pl->d_oo = (int64_t *)(b + o_oo);
	return pl;
}

void mm2c_seedplan_destroy(mm2c_seedplan_t *pl)
{
	if (!pl) return;
	{
		DeviceScope on(pl->device);
		if (pl->ran) (void)hipDeviceSynchronize();
		dev_free_synced(pl->d_mem);
		if (pl->ev0) (void)hipEventDestroy(pl->ev0); if (pl->ev1) (void)hipEventDestroy(pl->ev1);
		for (int i = 0; i < 3; ++i) if (pl->aux[i]) (void)hipStreamDestroy(pl->aux[i]);
		for (int i = 0; i < 4; ++i) if (pl->fork[i]) (void)hipEventDestroy(pl->fork[i]);
	}
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
			return mm2c_seed_chain_batch_host(par, min_cnt, min_sc, k1 - k0, h_match_off + k0, h_matches, h_hits, n_hits, h_qlen + k0,
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
	if (in_split_worker()) {                       // the worker of a split batch: the stream of its own device
		ThreadCtx *c;
		if ((rc = get_thread_ctx(&c))) { mm2c_plan_destroy(pl); mm2c_seedplan_destroy(sp); return rc; }
		st = c->st;
	}
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

} // extern "C"
