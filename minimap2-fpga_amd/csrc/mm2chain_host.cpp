// mm2chain_host.cpp -- C-ABI entries that take host buffers (include/mm2chain.h): the reference's call pattern (one blocking call per
// read from many threads, chain_hardware.cpp:27-197) and the batched ones; passes over staged / pipelined copies, the combiner of small
// concurrent calls, the whole-function batch entry.
#include "api_internal.h"
#include <sched.h>

using namespace mm2c_api;


namespace mm2c_api {

// one caller's batch: CSR tasks in pageable host memory
struct HostReq {
	const mm2c_params_t *par; int64_t n_tasks; const int64_t *off; const mm2c_anchor_t *a; const float *avg;
	int32_t *f, *p;
	int rc = 0; bool done = false; char err[256];
	bool taken = false;               // a leader of the call combiner has put the request into its pass (its owner then only waits for `done`)
};

// where the wall time of the small staged passes goes (MM2C_PASS_TIMING=1: printed when the combiner is released, mm2c_shutdown): assembling the pass on the host,
// the runtime calls that put it on the stream, the wait for it, handing the results back
static std::atomic<uint64_t> g_pt_build{0}, g_pt_submit{0}, g_pt_wait{0}, g_pt_out{0}, g_pt_n{0};
static std::atomic<uint64_t> g_pt_cls_ns[8], g_pt_cls_calls[8], g_pt_cls_reqs{0};   // ... and the callers' wall time by size of the call (< 256, < 512, ... anchors)
static const bool g_pt_on = getenv("MM2C_PASS_TIMING") != nullptr;
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
	__builtin_ia32_pause();
#elif defined(__aarch64__)
	asm volatile("yield" ::: "memory");
#else
	asm volatile("" ::: "memory");
#endif
}
static inline uint64_t pt_now() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
	const uint64_t pt0 = pt_now();
	const mm2c_params_t *par = reqs[0]->par;
	const uint64_t D = (uint64_t)(int64_t)par->max_dist_x;
	int64_t total = 0, n_tasks_all = 0;
	for (int r = 0; r < n_req; ++r) { total += reqs[r]->off[reqs[r]->n_tasks] - reqs[r]->off[0]; n_tasks_all += reqs[r]->n_tasks; }
	if (total == 0) return 0;
	// cutting costs a host pass over the anchors: only worth it when the pass has too few tasks to fill the GPU on its own
	const int64_t seg_min = n_tasks_all < G.cut_below_tasks ? (int)G.seg_min.load() : 0;
	// uncut tasks without a caller-supplied avg_qspan_scaled: the kernel sums the spans itself (chain.c:48-49), no host pass
	bool kernel_avg = seg_min == 0;
	for (int r = 0; r < n_req; ++r) if (reqs[r]->avg) kernel_avg = false;
	std::vector<int64_t> seg_off; std::vector<int32_t> pbase, order; std::vector<float> seg_avg;
	seg_off.reserve((size_t)n_tasks_all + 16); pbase.reserve((size_t)n_tasks_all + 16); seg_avg.reserve((size_t)n_tasks_all + 16);
	int64_t g0 = 0;                                                // where this request's anchors start in the arena
	// while the cutting pass looks at every anchor anyway: does any task hold more than one segment id (mmpriv.h:22-23)?  If none does, the pass runs with the ids
	// ignored -- the same f / p, chain.c:202-206 only looks at whether two ids are EQUAL -- and needs neither the kernel's per-tile check nor the launch that redoes
	// flagged tasks with the general variant
	bool one_seg_each = seg_min > 0;
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
			if (seg_min > 0) {
				const uint64_t seg_first = a[t0].y >> 48 & 0xff;
				uint64_t seg_diff = 0;
				for (int64_t i = t0 + 1; i < t1; ++i) {
					seg_diff |= (a[i].y >> 48 & 0xff) ^ seg_first;
					if (i - s0 >= seg_min && a[i].x > a[i - 1].x + D) {
						seg_off.push_back(g0 + s0); pbase.push_back((int32_t)(s0 - t0)); seg_avg.push_back(avg);
						s0 = i;
					}
				}
				if (seg_diff) one_seg_each = false;
			}
			seg_off.push_back(g0 + s0); pbase.push_back((int32_t)(s0 - t0)); seg_avg.push_back(avg);
		}
		g0 += q.off[q.n_tasks] - base;
	}
	const int64_t n_seg = (int64_t)seg_off.size();
	seg_off.push_back(total);
	if ((rc = build_order(n_seg, seg_off.data(), order))) return rc;

	HIP_TRY(hipSetDevice(c->device >= 0 ? c->device : cur_device()));   // the context's stream and arenas belong to that device (a combiner context: its slot's)
	// chunk size of the two-stream pipeline: "pipeline_chunk_anchors" (a chunk that fills the GPU on its own) for batches many times that size;
	// a batch of a few chunks' worth is cut into about eight pieces of at least 4 Mi anchors instead, whose kernels overlap on the two streams --
	// the upload of a piece then hides behind the kernels of the one before (one pass over 2 * 10^7 anchors: 1.44 G anchors/s, PCIe and kernels in series)
	const int64_t pipe_chunk = std::max<int64_t>(std::min<int64_t>(G.pipeline_chunk_anchors, total / std::max<int64_t>(G.pipeline_pieces, 1)),
	                                             std::min<int64_t>(G.pipeline_chunk_anchors, G.pipeline_min_chunk));
	// the prepass classes (which LDS ring a piece takes: chain_window_start -> chain_cls_settle -> the instantiations of chain_dp_tile), as plans have them:
	// one byte per piece + one zeroed set of counter blocks per launch (a pipelined batch settles its classes chunk by chunk)
	const size_t cstat_bytes = 32 * (size_t)mm2c::CLS_STAT_SLOTS, max_launches = (size_t)(total / std::max<int64_t>(pipe_chunk, 1)) + 10;   // (+ the small chunks of the tapered tail)
	const size_t o_a = 0, o_off = align16((size_t)total * 16), o_ord = align16(o_off + ((size_t)n_seg + 1) * 8),
	             o_pb = align16(o_ord + (size_t)n_seg * 4), o_avg = align16(o_pb + (size_t)n_seg * 4),
	             o_stat = align16(o_avg + (size_t)n_seg * 4), o_cls = align16(o_stat + (size_t)n_seg * 4), o_cstat = align16(o_cls + (size_t)n_seg),
	             o_hst = align16(o_cstat + cstat_bytes * max_launches);
	// round 6: a small pass that the kernels stage themselves brings its window starts along (chain.c:192-193, computed below in one sweep of the host: the reference's own
	// persistent pointer) -- 4 more bytes per anchor on PCIe instead of a launch of its own between the upload and the DP (17 -> 14.5 -> 10 us of submissions per pass)
	const bool host_st = (size_t)total <= G.stage_max_anchors && G.direct_pass.load() != 0 && (size_t)total <= G.direct_max_anchors && G.host_st.load() != 0;
	const size_t in_bytes = align16(o_hst + (host_st ? (size_t)total * 4 : 0));
	const size_t meta_bytes = in_bytes - o_off;
	const bool staged = (size_t)total <= G.stage_max_anchors;      // small passes go through pinned staging, big ones copy in place
	if ((rc = grow_device(&c->d_in, &c->cap_in, in_bytes))) return rc;
	if ((rc = grow_device(&c->d_out, &c->cap_out, (size_t)total * 8 + 16))) return rc;   // (+ 16: stage_out moves whole 16-byte pieces)
	if ((rc = grow_device(&c->d_scratch, &c->cap_scratch, (size_t)total * 8 + 64))) return rc;   // [t | st | counter of the direct pass]
	if ((rc = grow_pinned(&c->h_in, &c->cap_hin, staged ? in_bytes : meta_bytes, true))) return rc;
	char *hm = staged ? c->h_in + o_off : c->h_in;                 // where the metadata block starts in the staging buffer
	memcpy(hm, seg_off.data(), ((size_t)n_seg + 1) * 8);
	memcpy(hm + (o_ord - o_off), order.data(), (size_t)n_seg * 4);
	memcpy(hm + (o_pb - o_off), pbase.data(), (size_t)n_seg * 4);
	memcpy(hm + (o_avg - o_off), seg_avg.data(), (size_t)n_seg * 4);
	memset(hm + (o_stat - o_off), 0, in_bytes - o_stat);           // status, classes, class counters
	uint64_t pt1 = 0;
	// a small staged pass: the copies are kernels on the pass's own stream and the host polls a flag word (host_stage.hip)
	const bool direct = staged && G.direct_pass.load() != 0 && (size_t)total <= G.direct_max_anchors;
	unsigned *d_done = (unsigned *)(c->d_scratch + align16((size_t)total * 8));
	if (direct && !c->h_flag) {
		HIP_TRY(hipHostMalloc((void **)&c->h_flag, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
		*c->h_flag = 0; c->seq = 0;
	}
	if (direct && !c->d_cnt) {
		HIP_TRY(hipMalloc((void **)&c->d_cnt, 64));
		HIP_TRY(hipMemsetAsync(c->d_cnt, 0, 64, c->st));             // on the pass's own stream: in front of its kernels (hipMemset may return before the words are zero, and c->st does not wait for the null stream)
	}
	if (direct) d_done = c->d_cnt;                                 // zero between passes: the workgroup counted last puts it back (stage_out, chain_dp_coop)
	if (staged) {
		size_t at = o_a;
		for (int r = 0; r < n_req; ++r) {
			const size_t nb = (size_t)(reqs[r]->off[reqs[r]->n_tasks] - reqs[r]->off[0]) * 16;
			memcpy(c->h_in + at, reqs[r]->a + reqs[r]->off[0], nb);
			at += nb;
		}
		if (host_st) {
			// st[i] of every piece, relative to the piece: the first j with x_i <= x_j + max_dist_x, clamped to i - max_iter (chain.c:192-193; what chain_window_start finds by search)
			const mm2c_anchor_t *ha = (const mm2c_anchor_t *)(c->h_in + o_a);
			int32_t *hst = (int32_t *)(c->h_in + o_hst);
			const int64_t max_iter = std::max(par->max_iter, 0);
			for (int64_t k = 0; k < n_seg; ++k) {
				const int64_t b = seg_off[(size_t)k], n = seg_off[(size_t)k + 1] - b;
				int64_t st = 0;
				for (int64_t i = 0; i < n; ++i) {
					const uint64_t xi = ha[b + i].x;
					while (st < i && xi > ha[b + st].x + D) ++st;
					if (i - st > max_iter) st = i - max_iter;
					hst[b + i] = (int32_t)st;
				}
			}
		}
		pt1 = pt_now();
		if (!direct) HIP_TRY(hipMemcpyAsync(c->d_in, c->h_in, in_bytes, hipMemcpyHostToDevice, c->st));           // cf. chain_hardware.cpp:110,114
		// (direct: stage_in is launched below, once it is known whether the pass needs it -- a pass of short tasks that ends in the cooperative kernel does not)
	} else if (n_req == 1 && total >= 2 * pipe_chunk) {
		// big batch: a three-stage pipeline over chunks of whole pieces.  One stream uploads the chunks back to back (PCIe never idles), the
		// kernels of chunk k start when its upload has landed -- on one of three compute streams in turn, so that the kernels of consecutive
		// chunks overlap (a chunk does not fill the GPU: its kernel lasts as long as its longest task, whatever the chunk's size) -- and each
		// compute stream downloads f / p of its chunk behind its kernels (PCIe is full duplex).  With page-locked caller buffers the batch takes
		// about (upload of everything) + (kernel + download of the last chunk).  [Round 2: two streams that each ran upload, kernels, download for
		// every other chunk moved in lockstep: both uploaded, then both computed -- measured 15.9 ms for 2 * 10^7 anchors against 14.2 unpipelined.]
		if (!c->st2) HIP_TRY(create_partner_stream(&c->st2));
		if (!c->st3) {
			int least = 0, greatest = 0;
			if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
			HIP_TRY(least != greatest ? hipStreamCreateWithPriority(&c->st3, hipStreamNonBlocking, least) : hipStreamCreateWithFlags(&c->st3, hipStreamNonBlocking));
		}
		if (!c->st_up) HIP_TRY(hipStreamCreateWithFlags(&c->st_up, hipStreamNonBlocking));
		const HostReq &q = *reqs[0];
		const mm2c_anchor_t *src = q.a + q.off[0];
		int32_t *dst_f = q.f + q.off[0], *dst_p = q.p + q.off[0];
		const int64_t chunk_anchors = pipe_chunk;
		// Round 6: a tapered schedule.  The batch takes (upload of everything) + (kernel and download of the LAST chunk), so the last chunks are made small: 1/2, 1/4, 1/8 of
		// the chunk size ("pipeline_taper": how many halvings, 0 = equal chunks) -- what stays unhidden at the end is the kernel and download of an eighth of a chunk.
		const int taper = (int)std::max<int64_t>(0, std::min<int64_t>(6, G.pipeline_taper.load()));
		int64_t tail_total = 0;
		for (int h = 1; h <= taper; ++h) tail_total += chunk_anchors >> h;
		std::vector<int64_t> cuts(1, 0);
		for (int64_t s0 = 0; s0 < n_seg;) {
			// anchors left: the tail starts where what is left fits the tapered sizes; inside it the target halves from chunk to chunk
			const int64_t left = total - seg_off[(size_t)s0];
			int64_t target = chunk_anchors;
			if (taper > 0 && left <= tail_total + chunk_anchors / 2) {
				target = chunk_anchors >> 1;
				int64_t rest = tail_total;
				for (int h = 1; h <= taper && left < rest; ++h) { rest -= chunk_anchors >> h; target = chunk_anchors >> std::min(h + 1, taper); }
			}
			int64_t s1 = s0 + 1;
			while (s1 < n_seg && seg_off[(size_t)s1 + 1] - seg_off[(size_t)s0] <= target) ++s1;
			cuts.push_back(s1); s0 = s1;
		}
		const int n_chunks = (int)cuts.size() - 1;
		while (c->evs.size() < (size_t)n_chunks) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->evs.push_back(e); }
		HIP_TRY(hipMemcpyAsync(c->d_in + o_off, hm, meta_bytes, hipMemcpyHostToDevice, c->st_up));
		hipStream_t comp[3] = { c->st, c->st2, c->st3 };
		int nl = 0;
		for (int k = 0; k < n_chunks; ++k) {
			const int64_t s0 = cuts[(size_t)k], s1 = cuts[(size_t)k + 1];
			const int64_t a0 = seg_off[(size_t)s0], a1 = seg_off[(size_t)s1];
			hipStream_t st = comp[k % 3];
			hipEvent_t ev_up = c->evs[(size_t)k];
			HIP_TRY(hipMemcpyAsync(c->d_in + o_a + (size_t)a0 * 16, src + a0, (size_t)(a1 - a0) * 16, hipMemcpyHostToDevice, c->st_up));
			HIP_TRY(hipEventRecord(ev_up, c->st_up));
			HIP_TRY(hipStreamWaitEvent(st, ev_up, 0));
			mm2c::LaunchArgs L; L.coop_w8_above = G.coop_w8_above.load(); L.fuse_st = G.fuse_st.load();
			L.P = to_kparams(par);
			L.n_tasks = s1 - s0; L.d_offsets = (const int64_t *)(c->d_in + o_off) + s0; L.d_order = nullptr;
			L.d_anchors = c->d_in + o_a; L.d_avg = kernel_avg ? nullptr : (const float *)(c->d_in + o_avg) + s0;
			L.d_pbase = (const int32_t *)(c->d_in + o_pb) + s0; L.d_status = (int32_t *)(c->d_in + o_stat) + s0;
			L.d_f = (int32_t *)c->d_out; L.d_p = (int32_t *)(c->d_out + (size_t)total * 4);
			L.d_t = (int32_t *)c->d_scratch; L.d_st = (int32_t *)(c->d_scratch + (size_t)total * 4);
			L.ring_class = G.ring_class; L.force_tab = G.force_tab; L.compact = G.compact_ring; L.q24 = G.q24_ring; L.noskip_loop = G.noskip_loop;
			if ((size_t)k < max_launches) {
				L.d_cls = (uint8_t *)(c->d_in + o_cls) + s0; L.d_cls_stat = (unsigned long long *)(c->d_in + o_cstat + cstat_bytes * (size_t)k);
				L.far_ring = G.far_ring; L.far_thr10 = G.far_thr10; L.wide_pct = G.wide_pct;
			}
			// the LAST chunks of the pipeline: nothing follows them that could hide the length of their longest task (one wave per task: 3.5 ms for 5 000 anchors on an
			// otherwise empty GPU, the tail of the whole batch) -- chunks of few enough pieces take several waves per piece instead ("pipe_coop_chunks": how many, 0 = none)
			// Round 6: ANY chunk of few long pieces does (coop_pays: a chunk of long reads has fewer pieces than the GPU has wave slots).
			{
				int64_t longest = 0;
				for (int64_t q2 = s0; q2 < s1; ++q2) longest = std::max<int64_t>(longest, seg_off[(size_t)q2 + 1] - seg_off[(size_t)q2]);
				// (inside the pipeline three chunks' kernels share the GPU and what counts is their joint rate: one wave per piece keeps up with the upload while a piece
				// lasts no longer than about three chunk uploads -- 0.7 us per anchor against 16 bytes per anchor at 56 GB/s: pieces up to 1/800 of the chunk -- and the
				// cooperative kernel, alone at 3 G anchors/s, would not: measured 2.6 -> 2.1 G anchors/s on the bench's batch when every chunk took it)
				if ((k >= n_chunks - G.pipe_coop_chunks.load() && s1 - s0 <= G.coop_max_tasks) ||
				    (G.coop_plans.load() == 2 && mm2c::coop_pays(s1 - s0, longest, a1 - a0, G.coop_w8_above.load()) && longest * 800 > a1 - a0)) {
					L.coop_waves = G.coop_waves.load();
					L.max_task_anchors = longest;
				}
			}
			HIP_TRY(mm2c::launch_chain_dp(L, st, &nl, nullptr, k == 0 ? &c->last_info : nullptr));
			if (k == 0) note_host_variant(c->last_info);
			// each compute stream downloads its own chunk (a separate download stream behind an event turned the copies into blit kernels that
			// held up the next upload: profiles/r3_e2e.md)
			HIP_TRY(hipMemcpyAsync(dst_f + a0, L.d_f + a0, (size_t)(a1 - a0) * 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipMemcpyAsync(dst_p + a0, L.d_p + a0, (size_t)(a1 - a0) * 4, hipMemcpyDeviceToHost, st));
		}
		{	// every chunk's kernels precede its download; the uploads precede the kernels
			ScopedNs timed(SS.wait_ns);
			for (int k = 0; k < 3; ++k) HIP_TRY(hipStreamSynchronize(comp[k]));
		}
		SS.chunks += (uint64_t)n_chunks;
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
	mm2c::LaunchArgs L; L.coop_w8_above = G.coop_w8_above.load(); L.fuse_st = G.fuse_st.load();
	L.P = to_kparams(par);
	L.n_tasks = n_seg; L.d_offsets = (const int64_t *)(c->d_in + o_off); L.d_order = (const int32_t *)(c->d_in + o_ord);
	L.d_anchors = c->d_in + o_a; L.d_avg = kernel_avg ? nullptr : (const float *)(c->d_in + o_avg); L.d_pbase = (const int32_t *)(c->d_in + o_pb);
	L.d_status = (int32_t *)(c->d_in + o_stat);
	L.d_f = (int32_t *)c->d_out; L.d_p = (int32_t *)(c->d_out + (size_t)total * 4);
	L.d_t = (int32_t *)c->d_scratch; L.d_st = (int32_t *)(c->d_scratch + (size_t)total * 4);
	if (staged && host_st) { L.d_st = (int32_t *)(c->d_in + o_hst); L.st_ready = 1; }   // (came up with the arena; a pass that runs one wave per piece after all computes them again, with its classes)
	L.ring_class = G.ring_class; L.force_tab = G.force_tab; L.compact = G.compact_ring; L.q24 = G.q24_ring; L.noskip_loop = G.noskip_loop;
	L.d_cls = (uint8_t *)(c->d_in + o_cls); L.d_cls_stat = (unsigned long long *)(c->d_in + o_cstat);
	L.far_ring = G.far_ring; L.far_thr10 = G.far_thr10; L.wide_pct = G.wide_pct;
	// a pass of few pieces (a lone call, a handful of combined calls) cannot fill the GPU with one wave per piece: several waves per piece (chain_dp_coop.h)
	// (round 6: and a bigger pass of few LONG pieces, coop_pays)
	{
		int64_t longest = 0;
		for (int64_t k = 0; k < n_seg; ++k) longest = std::max<int64_t>(longest, seg_off[(size_t)k + 1] - seg_off[(size_t)k]);
		L.coop_waves = (n_seg <= G.coop_max_tasks || (G.coop_plans.load() == 2 && mm2c::coop_pays(n_seg, longest, total, G.coop_w8_above.load()))) ? G.coop_waves.load() : 0;
		if (L.coop_waves > 1) L.max_task_anchors = longest;
	}
	if (one_seg_each && par->n_segs <= 1 && !par->is_cdna) L.P.flags |= mm2c::KF_IGNORE_SEG;
	int nl = 0;
	if (staged) { if ((rc = grow_pinned(&c->h_out, &c->cap_hout, (size_t)total * 8 + 16, true))) return rc; }
	if (direct) {
		// round 6: a pass that ends in the cooperative kernel needs no fourth launch -- the kernel stores f / p to the result buffer as it goes and raises the flag
		if (++c->seq == 0) c->seq = 1;
		if (G.fused_out.load()) { L.h_f = (int32_t *)c->h_out; L.h_p = (int32_t *)(c->h_out + (size_t)total * 4); L.d_done = d_done; L.h_flag = c->h_flag; L.seq = c->seq; }
	}
	if (direct) {
		// round 6: ONE launch for a pass of short tasks.  When the pass ends in the sixteen-wave kernel, which makes its own window starts and writes the caller's buffer, the
		// kernel can read the pass from the pinned arena as well: the metadata where it lies (a few words per workgroup), the anchors copied to the device by the workgroup that
		// chains them.  Asked of the launcher itself (dry run), so that the conditions live in one place.
		bool single = false;
		if (G.single_launch.load() && !host_st) {
			mm2c::LaunchArgs Ld = L; Ld.dry_run = 1;
			mm2c::LaunchInfo inf = {};
			HIP_TRY(mm2c::launch_chain_dp(Ld, c->st, nullptr, nullptr, &inf));
			single = inf.single_ok != 0;
		}
		if (single) {
			L.h_anchors = c->h_in + o_a;
			L.d_offsets = (const int64_t *)(c->h_in + o_off); L.d_order = nullptr;   // (no launch order: a handful of workgroups start together anyway, and the look-up would be one more trip to host memory in front of the others)
			L.d_avg = (const float *)(c->h_in + o_avg); L.d_pbase = (const int32_t *)(c->h_in + o_pb);
			L.hm_off = (const int64_t *)(c->h_in + o_off); L.hm_avg = (const float *)(c->h_in + o_avg); L.hm_pbase = (const int32_t *)(c->h_in + o_pb);   // (few pieces: these travel in the kernel's arguments)
		} else HIP_TRY(mm2c::launch_stage_in(c->h_in, c->d_in, in_bytes, d_done, c->st));
	}
	HIP_TRY(mm2c::launch_chain_dp(L, c->st, &nl, nullptr, &c->last_info));                                           // cf. chain_hardware.cpp:156
	note_host_variant(c->last_info);
	if (staged) {
		uint64_t pt2;
		if (direct) {
			if (!c->last_info.host_out) HIP_TRY(mm2c::launch_stage_out(c->d_out, c->h_out, (size_t)total * 8, d_done, c->h_flag, c->seq, c->st));
			pt2 = pt_now();
			// the flag is the last thing the pass writes (after a system-scope fence behind every store of f / p).  The calling thread has nothing else to do (the call is
			// synchronous, chain_hardware.cpp:175 clFinish), so it polls; a pass that does not report within 50 ms is waited for through the runtime, which also surfaces
			// an error of the stream
			bool seen = false;
			for (uint64_t spins = 0; ; ++spins) {
				if (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) == c->seq) { seen = true; break; }
				cpu_relax();
				// a host that runs more threads than it has cores (a path-B host may: a thread inside the call only waits) must not lose a core to this loop: after about the
				// length of a short pass the thread offers its core between looks (returns at once when nobody else is runnable)
				if (spins > 1024 && (spins & 31) == 31) sched_yield();
				if ((spins & 1023) == 1023 && pt_now() - pt2 > 50000000ull) break;
			}
			if (!seen) {
				HIP_TRY(hipStreamSynchronize(c->st));
				if (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) != c->seq) return fail(MM2C_E_HIP, "a direct pass ended without raising its flag (pass %u)", c->seq);
			} else if ((c->seq & 31) == 0) (void)hipStreamQuery(c->st);   // lets the runtime retire the finished commands of a stream nobody ever waits on
		} else {
			HIP_TRY(hipMemcpyAsync(c->h_out, c->d_out, (size_t)total * 8, hipMemcpyDeviceToHost, c->st));           // cf. chain_hardware.cpp:167,170
			pt2 = pt_now();
			HIP_TRY(hipStreamSynchronize(c->st));                                                               // cf. chain_hardware.cpp:175
		}
		const uint64_t pt3 = pt_now();
		size_t at = 0;
		for (int r = 0; r < n_req; ++r) {
			const size_t n = (size_t)(reqs[r]->off[reqs[r]->n_tasks] - reqs[r]->off[0]);
			memcpy(reqs[r]->f + reqs[r]->off[0], c->h_out + at * 4, n * 4);
			memcpy(reqs[r]->p + reqs[r]->off[0], c->h_out + (size_t)total * 4 + at * 4, n * 4);
			at += n;
		}
		g_pt_build += pt1 - pt0; g_pt_submit += pt2 - pt1; g_pt_wait += pt3 - pt2; g_pt_out += pt_now() - pt3; ++g_pt_n; g_pt_cls_reqs += (uint64_t)n_req;
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

// which instantiation the last DP launch of a host-buffer entry chose, process-wide (the parity tests assert that these entries reach the compact ring)
static std::mutex g_variant_mu;
static mm2c::LaunchInfo g_last_host_info = {};
static bool g_have_host_info = false;
void note_host_variant(const mm2c::LaunchInfo &I)
{
	std::lock_guard<std::mutex> lk(g_variant_mu);
	g_last_host_info = I; g_have_host_info = true;
}

// Combiner for small synchronous calls (the reference's call pattern: up to n_threads host threads, each blocking in
// run_chaining_on_hw / mm_chain_dp, map.c:561).  A caller that finds no pass in flight becomes the leader: it takes every
// pending request with the same scalars, runs ONE GPU pass for all of them and wakes their owners; callers that arrive
// meanwhile queue up and are served by the next leader.  (The reference instead serialises callers on a mutex and a FIFO,
// chain_hardware.cpp:54-93.)  Big requests skip the combiner and run on the caller's own stream.
// Up to N_LANES passes in flight, each on a context (stream + arenas) of its own: with the cooperative kernel a pass is short on the GPU (tens of microseconds), and the
// host side of the next one -- collecting the requests, staging their anchors, the runtime calls -- can run beside it ("combiner_lanes", default 3: measured 5.53-5.67 / 5.40 / 5.26 / 5.53 s for 1 / 2 / 3 / 4 on the 120 000-read run).
// ONE COMBINER PER DEVICE SLOT (round 5): the reference keeps a queue, a lock, a buffer set and an expected end time per kernel and picks one per call
// (chain_hardware.cpp:9-23,54-72); here every device of mm2c_init_devices / MM2C_DEVICES has its own combiner -- lanes, streams, arenas on that device -- and a
// per-read call goes to the slot with the least work outstanding (anchors of the calls inside it), the scan starting at tid % n so that idle devices are taken in turn.
constexpr int N_LANES = 16;
constexpr int MAX_SLOTS = 64;
struct Combiner {
	std::mutex mu;
	std::condition_variable cv;
	std::vector<HostReq *> pending;
	int leaders = 0;                    // passes being put together or in flight
	bool busy[N_LANES] = {};
	ThreadCtx ctx[N_LANES];             // stream + arenas of a pass in flight (exclusive to its leader)
	uint64_t epoch[N_LANES];
	Combiner() { for (uint64_t &e : epoch) e = ~0ull; }
	std::atomic<int64_t> outstanding{0};            // anchors of the calls that have entered this slot and not yet left it (what the routing balances)
	std::atomic<uint64_t> passes{0}, calls{0}, anchors{0};   // served since mm2c_init (mm2c_get_slot_stats)
	std::atomic<double> pred_ms{0.0};               // path A, rule 2: predicted device time (hw_time_pred) of the calls inside the slot (run_chaining_on_hw's decline, chain_hardware.cpp:54-75)
	std::atomic<int> inside{0};                     // path A, rule 1: calls that have entered the slot and not yet left it ...
	std::atomic<float> svc_ms{0.f};                 // ... and what a pass of this slot has taken lately (EWMA of the wall time of run_requests, ms; 0: nothing measured yet)
};
static Combiner CB[MAX_SLOTS];
std::atomic<uint64_t> g_declined{0};

void release_combiner()
{
	if (g_pt_on && g_pt_n.load()) {
		static const char *cls[8] = { "<256", "<512", "<1024", "<2048", "<4096", "<8192", "<16384", ">=16384" };
		for (int k = 0; k < 8; ++k)
			if (g_pt_cls_calls[k].load()) fprintf(stderr, "[mm2chain] calls of %s anchors: %llu, %.1f us each\n", cls[k], (unsigned long long)g_pt_cls_calls[k].load(), g_pt_cls_ns[k].load() / 1e3 / g_pt_cls_calls[k].load());
		fprintf(stderr, "[mm2chain] %.2f requests per staged pass\n", (double)g_pt_cls_reqs.load() / g_pt_n.load());
		fprintf(stderr, "[mm2chain] %llu staged passes: assembled on the host %.1f us each, put on the stream %.1f us, waited for %.1f us, results handed back %.1f us\n",
		        (unsigned long long)g_pt_n.load(), g_pt_build.load() / 1e3 / g_pt_n.load(), g_pt_submit.load() / 1e3 / g_pt_n.load(), g_pt_wait.load() / 1e3 / g_pt_n.load(), g_pt_out.load() / 1e3 / g_pt_n.load());
	}
	for (int s = 0; s < MAX_SLOTS; ++s) {
		std::lock_guard<std::mutex> lk(CB[s].mu);
		for (int k = 0; k < N_LANES; ++k) {
			if (CB[s].ctx[k].st && CB[s].ctx[k].device >= 0) (void)hipSetDevice(CB[s].ctx[k].device);
			CB[s].ctx[k].release(); CB[s].epoch[k] = ~0ull;
		}
		CB[s].passes = 0; CB[s].calls = 0; CB[s].anchors = 0; CB[s].outstanding = 0; CB[s].pred_ms = 0.0; CB[s].inside = 0; CB[s].svc_ms = 0.f;
	}
	g_declined = 0;
}

// The slot a per-read call goes to: the one with the least anchors outstanding, idle slots in turn from tid % n (negative tid: as 0).  Adds the call's anchors to the
// slot; the caller hands them back with leave_slot().
int enter_slot(int tid, int64_t n_anchors, int forced)
{
	const int nd = std::max(1, std::min(n_devices(), MAX_SLOTS));
	int best = 0;
	if (forced >= 0 && forced < nd) best = forced;             // path A: the slot that accepted the call under the busy protocol runs it (chain_hardware.cpp:58-72: the kernel that accepts is the kernel that runs)
	else if (nd > 1) {
		int64_t out[MAX_SLOTS];
		for (int s = 0; s < nd; ++s) out[s] = CB[s].outstanding.load(std::memory_order_relaxed);
		best = mm2c_route_slot(nd, out, tid);
	}
	CB[best].outstanding.fetch_add(n_anchors, std::memory_order_relaxed);
	CB[best].inside.fetch_add(1, std::memory_order_relaxed);
	return best;
}
void leave_slot(int slot, int64_t n_anchors) { CB[slot].outstanding.fetch_sub(n_anchors, std::memory_order_relaxed); CB[slot].inside.fetch_sub(1, std::memory_order_relaxed); }

int get_slot_stats(int slot, uint64_t *passes, uint64_t *calls, uint64_t *anchors)
{
	if (slot < 0 || slot >= MAX_SLOTS) return -1;
	*passes = CB[slot].passes.load(); *calls = CB[slot].calls.load(); *anchors = CB[slot].anchors.load();
	return 0;
}

// Path A's decline (chain_hardware.cpp:54-75): the reference accepts a call on a kernel when (the time until that kernel is free) + hw_time_pred < sw_time_pred and
// otherwise tries the next kernel, returning 1 when none will do.  Here a device serves several calls at once (combined passes, `combiner_lanes` of them in
// flight), so "the time until it is free" is the predicted device time of the calls inside the slot divided by the lanes.  Returns the slot that took the call (its
// hw_time_pred is then booked until release_pred) or -1 = declined.  Predictions that are not positive (a caller that has no model) never decline.
// Round 6, rule 1 ("decline_when_busy" 1): the measured thing instead of a sum of predictions -- per slot the number of calls inside it and an EWMA of what a pass of
// that slot has taken lately; a call is turned away only when (calls ahead / lanes + 1) x that service time exceeds sw_time_pred, i.e. when waiting its turn and being
// served is expected to last longer than the caller's own loop.  Rule 2 ("decline_when_busy" 2) is round 5's: booked hw_time_pred / lanes + hw_time_pred < sw_time_pred.
// Returns the slot that took the call (under rule 2 its hw_time_pred is booked until release_pred) or -1 = declined.
int book_pred(int tid, float hw_ms, float sw_ms)
{
	const int nd = std::max(1, std::min(n_devices(), MAX_SLOTS));
	const int first = (int)((unsigned)(tid < 0 ? 0 : tid) % (unsigned)nd);
	const double lanes = (double)std::max(1, std::min<int>(N_LANES, G.combiner_lanes));
	if (G.decline_when_busy.load() == 1) {
		for (int k = 0; k < nd; ++k) {
			const int s = (first + k) % nd;
			const double svc = (double)CB[s].svc_ms.load(std::memory_order_relaxed);
			const double ahead = (double)std::max(0, CB[s].inside.load(std::memory_order_relaxed));
			if (!(hw_ms > 0.f && sw_ms > 0.f) || svc <= 0.0 || (ahead / lanes + 1.0) * svc <= (double)sw_ms) return s;
		}
		// turned away everywhere.  The estimate only moves when passes run, so a slot that looks slow (its first passes pay for code loading and arena growth) would
		// be avoided for good: every decline lets it forget a little, and the slot is tried again
		for (int k = 0; k < nd; ++k) { const float v = CB[k].svc_ms.load(std::memory_order_relaxed); CB[k].svc_ms.store(v * 0.98f, std::memory_order_relaxed); }
		return -1;
	}
	for (int k = 0; k < nd; ++k) {
		const int s = (first + k) % nd;
		double cur = CB[s].pred_ms.load(std::memory_order_relaxed);
		for (;;) {
			const double total = cur / lanes + (double)hw_ms;
			if (!(hw_ms > 0.f && sw_ms > 0.f) || total < (double)sw_ms) {
				if (CB[s].pred_ms.compare_exchange_weak(cur, cur + (double)hw_ms, std::memory_order_relaxed)) return s;
				continue;                                            // another caller booked meanwhile: look again
			}
			break;
		}
	}
	return -1;
}
void release_pred(int slot, float hw_ms)
{
	if (G.decline_when_busy.load() != 2) return;              // (only rule 2 books predictions)
	double cur = CB[slot].pred_ms.load(std::memory_order_relaxed);
	while (!CB[slot].pred_ms.compare_exchange_weak(cur, std::max(0.0, cur - (double)hw_ms), std::memory_order_relaxed)) {}
}

int submit_combined(HostReq *me, int slot)
{
	Combiner &cb = CB[slot];
	std::unique_lock<std::mutex> lk(cb.mu);
	cb.pending.push_back(me);
	for (;;) {
		if (me->done) return me->rc;
		if (!me->taken && cb.leaders < std::max(1, std::min<int>(N_LANES, G.combiner_lanes))) break;   // (a request that sits in another leader's pass waits for that pass)
		cb.cv.wait(lk);
	}
	int lane_k = 0;
	while (lane_k < N_LANES - 1 && cb.busy[lane_k]) ++lane_k;
	cb.busy[lane_k] = true;
	// leader: collect the pending requests that share my scalars, up to the staging size.  Nothing in here may leave the followers waiting
	// or `leader_active` set: allocation failures (std::bad_alloc from the vectors here and inside run_requests) become an error code for
	// every request of the batch, and no exception crosses the extern "C" boundary.
	++cb.leaders;
	std::vector<HostReq *> batch, rest;
	int rc = 0;
	try {
		size_t tot = 0;
		for (HostReq *q : cb.pending) {
			const size_t n = (size_t)(q->off[q->n_tasks] - q->off[0]);
			if ((q == me || (memcmp(q->par, me->par, sizeof(mm2c_params_t)) == 0 && tot + n <= G.stage_max_anchors)) ) { batch.push_back(q); tot += n; q->taken = true; }
			else rest.push_back(q);
		}
		cb.pending.swap(rest);
	} catch (...) {
		// could not even form the batch: serve only myself (the others stay pending for the next leader)
		for (HostReq *q : cb.pending) if (q != me) q->taken = false;
		batch.clear();
		for (size_t k = 0; k < cb.pending.size(); ++k) if (cb.pending[k] == me) { cb.pending.erase(cb.pending.begin() + (long)k); break; }
		me->rc = fail(MM2C_E_ARG, "out of host memory in the call combiner"); me->done = true;
		strncpy(me->err, g_err, sizeof(me->err) - 1); me->err[sizeof(me->err) - 1] = 0;
		--cb.leaders; cb.busy[lane_k] = false;
		cb.cv.notify_all();
		return me->rc;
	}
	lk.unlock();
	try {
		async_init_join();
		{
			std::lock_guard<std::mutex> gl(G.mu);
			if (!G.ready) rc = fail_not_ready();                      // (joined above, outside the lock)
			else if (slot >= (int)G.devices.size()) rc = fail(MM2C_E_ARG, "device slot %d of %d", slot, (int)G.devices.size());
			else if (cb.epoch[lane_k] != G.epoch) {                   // first pass after (re)initialisation: fresh stream and arenas ON THE SLOT'S DEVICE
				cb.ctx[lane_k] = ThreadCtx();
				cb.ctx[lane_k].device = G.devices[(size_t)slot];
				hipError_t e = hipSetDevice(cb.ctx[lane_k].device);
				if (e == hipSuccess) e = hipStreamCreateWithFlags(&cb.ctx[lane_k].st, hipStreamNonBlocking);
				if (e != hipSuccess) rc = fail(MM2C_E_HIP, "combiner stream: %s", hipGetErrorString(e));
				else cb.epoch[lane_k] = G.epoch;
			}
		}
		if (rc == 0) {
			const uint64_t t_pass = pt_now();
			rc = run_requests(&cb.ctx[lane_k], batch.data(), (int)batch.size());
			if (rc == 0 && cb.passes.load(std::memory_order_relaxed) >= 8) {   // what a pass of this slot takes, for the busy protocol (racy read-modify-write: an estimate; not its first passes)
				const float ms = (float)((pt_now() - t_pass) * 1e-6), old = cb.svc_ms.load(std::memory_order_relaxed);
				cb.svc_ms.store(old > 0.f ? 0.875f * old + 0.125f * ms : ms, std::memory_order_relaxed);
			}
		}
	} catch (...) {
		rc = fail(MM2C_E_ARG, "out of host memory in a combined chaining pass");
	}
	if (rc == 0) {
		uint64_t tot = 0;
		for (HostReq *q : batch) tot += (uint64_t)(q->off[q->n_tasks] - q->off[0]);
		cb.passes += 1; cb.calls += (uint64_t)batch.size(); cb.anchors += tot;
	}
	lk.lock();
	for (HostReq *q : batch) {
		q->rc = rc; q->done = true;
		if (rc != 0) { strncpy(q->err, g_err, sizeof(q->err) - 1); q->err[sizeof(q->err) - 1] = 0; }
	}
	--cb.leaders; cb.busy[lane_k] = false;
	cb.cv.notify_all();
	return me->rc;
}

} // namespace mm2c_api

extern "C" {

static int chain_batch_host_tid(const mm2c_params_t *par, int64_t n_tasks, const int64_t *h_offsets, const mm2c_anchor_t *h_anchors,
                                const float *h_avg_qspan, int32_t *h_f, int32_t *h_p, int tid, int forced_slot = -1)
{
	int rc;
	const auto t_begin = std::chrono::steady_clock::now();
	if ((rc = check_params(par))) return rc;
	if ((rc = check_offsets(n_tasks, h_offsets))) return rc;            // the launch order is built where the pass is put together (run_requests)
	if (n_tasks == 0) return 0;
	const int64_t total = h_offsets[n_tasks] - h_offsets[0];
	if (total == 0) return 0;
	if (!h_anchors || !h_f || !h_p) return fail(MM2C_E_ARG, "host pointer is NULL");
	if (should_split(total))      // several devices: one contiguous range of tasks per device, side by side (f / p are indexed by the caller's offsets)
		return run_split(n_tasks, h_offsets, [&](int, int64_t k0, int64_t k1) {
			return chain_batch_host_tid(par, k1 - k0, h_offsets + k0, h_anchors, h_avg_qspan ? h_avg_qspan + k0 : nullptr, h_f, h_p, tid);
		});
	HostReq req;
	req.par = par; req.n_tasks = n_tasks; req.off = h_offsets; req.a = h_anchors; req.avg = h_avg_qspan; req.f = h_f; req.p = h_p;
	req.err[0] = 0;
	// the combiner's context (stream, arenas) lives on the primary device: the worker of a split batch drives another one and runs on the
	// context of its own device slot, however small its range is
	if (!in_split_worker() && (size_t)total <= G.combine_max_anchors) {
		// the device slot with the least work inside it takes the call (one combiner per device)
		const int slot = enter_slot(tid, total, forced_slot);
		rc = submit_combined(&req, slot);
		leave_slot(slot, total);
		if (rc != 0 && req.err[0]) fail(rc, "%s", req.err);
	} else {
		ScopedNs timed_total(SS.total_ns); ++SS.calls;
		ThreadCtx *c;
		if ((rc = get_thread_ctx(&c))) return rc;
		HostReq *one = &req;
		rc = run_requests(c, &one, 1);
	}
	const uint64_t call_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
	G.host_call_ns += call_ns;
	if (g_pt_on) { int k = 0; while (k < 7 && total >= (256LL << k)) ++k; g_pt_cls_ns[k] += call_ns; ++g_pt_cls_calls[k]; }
	return rc;
}

int mm2c_chain_batch_host(const mm2c_params_t *par, int64_t n_tasks, const int64_t *h_offsets, const mm2c_anchor_t *h_anchors,
                          const float *h_avg_qspan, int32_t *h_f, int32_t *h_p)
{
	return chain_batch_host_tid(par, n_tasks, h_offsets, h_anchors, h_avg_qspan, h_f, h_p, -1);
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
	if (should_split(total)) {
		// several devices: each takes a contiguous range of tasks and leaves its chains compact at the place in u / b where its anchors
		// would start (there is room: a range has no more chains or chained anchors than anchors); then the ranges are closed up
		const int nd = n_devices();
		std::vector<std::vector<int64_t>> uo((size_t)nd), bo((size_t)nd);
		std::vector<int64_t> r0((size_t)nd, 0), r1((size_t)nd, 0);
		rc = run_split(n_tasks, h_offsets, [&](int part, int64_t k0, int64_t k1) {
			uo[(size_t)part].assign((size_t)(k1 - k0) + 1, 0); bo[(size_t)part].assign((size_t)(k1 - k0) + 1, 0);
			r0[(size_t)part] = k0; r1[(size_t)part] = k1;
			const int64_t at = h_offsets[k0] - h_offsets[0];
			return mm2c_mm_chain_dp_batch_host(par, min_cnt, min_sc, k1 - k0, h_offsets + k0, h_anchors, 0, uo[(size_t)part].data(), u + at,
			                                   bo[(size_t)part].data(), b + at);
		});
		if (rc != 0) return rc;
		int64_t U = 0, B = 0;
		for (int part = 0; part < nd; ++part) {
			const int64_t k0 = r0[(size_t)part], k1 = r1[(size_t)part];
			if (k1 == k0) continue;
			const int64_t at = h_offsets[k0] - h_offsets[0], nu = uo[(size_t)part].back(), nb = bo[(size_t)part].back();
			if (U != at) memmove(u + U, u + at, (size_t)nu * 8);
			if (B != at) memmove(b + B, b + at, (size_t)nb * 16);
			for (int64_t k = k0; k < k1; ++k) { u_off[k + 1] = U + uo[(size_t)part][(size_t)(k - k0) + 1]; b_off[k + 1] = B + bo[(size_t)part][(size_t)(k - k0) + 1]; }
			U += nu; B += nb;
		}
		return 0;
	}
	// everything on the GPU: anchors up, DP, epilogue, chains down; big batches in chunks of whole tasks on two streams, so that the
	// upload of chunk k+1, the kernels of chunk k and the download of chunk k-1 overlap
	if (total >= (int64_t)INT32_MAX && G.pipeline_chunk_anchors >= (int64_t)INT32_MAX) return fail(MM2C_E_TOOBIG, "batch too big for one chunk");
	ScopedNs timed_total(SS.total_ns); ++SS.calls;
	ThreadCtx *c;
	std::unique_lock<std::mutex> hold;                              // batch calls from different host threads take turns on one set of arenas
	if ((rc = get_batch_ctx(&c, hold))) return rc;
	DeviceScope on(cur_device());
	HIP_TRY(on.err);
	const int64_t chunk_anchors = total >= 2 * G.pipeline_chunk_anchors ? G.pipeline_chunk_anchors.load() : total;
	const mm2c_anchor_t *a0 = h_anchors + h_offsets[0];
	int64_t base_u = 0, base_b = 0;
	int nl = 0;

	auto enqueue = [&](WholeSlot &w, int64_t k0, int64_t k1) -> int {
		int r;
		const size_t nt = (size_t)(k1 - k0), tot = (size_t)(h_offsets[k1] - h_offsets[k0]);
		if (!w.st) HIP_TRY(&w == &c->whole[1] ? create_partner_stream(&w.st) : hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking));
		w.k0 = k0; w.k1 = k1; w.busy = true;
		// upload arena: [anchors | offsets | order | status]; pinned mirror of the metadata + room for the offsets that come back
		const size_t o_off = align16(tot * 16), o_ord = align16(o_off + (nt + 1) * 8), o_stat = align16(o_ord + nt * 4), o_cls = align16(o_stat + nt * 4),
		             o_clstat = align16(o_cls + nt), in_bytes = align16(o_clstat + 32 * (size_t)mm2c::CLS_STAT_SLOTS);   // ... | status | class per task | class counters (all zero on entry)
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
		// long tasks are cut into independent pieces on the device, as plans do (chain_cut)
		int64_t extra = 0;
		if (G.plan_cut && G.seg_min > 0)
			for (size_t k = 0; k < nt; ++k) { const int64_t len = m_off[k + 1] - m_off[k]; if (len >= G.plan_cut_min) extra += len / G.seg_min; }
		const size_t mp = extra > 0 ? nt + (size_t)extra : 0;
		const size_t o_epi = align16(tot * 16), epi_bytes = layout_epilogue(E, nullptr, tot, nt, tmp);
		const size_t o_cut = align16(o_epi + epi_bytes), o_cstat = o_cut + 256, o_chc = align16(o_cstat + mp * 4), o_cstart = align16(o_chc + nt * 4),
		             o_cend = align16(o_cstart + mp * 8), o_cpb = align16(o_cend + mp * 8), o_cavg = align16(o_cpb + mp * 4), o_ccls = align16(o_cavg + mp * 4),
		             work_bytes = align16(o_ccls + mp);
		if ((r = grow_device(&w.d_work, &w.cap_work, work_bytes))) return r;
		layout_epilogue(E, w.d_work + o_epi, tot, nt, tmp);
		const size_t o_boff = align16((nt + 1) * 8);
		w.o_res_u = align16(o_boff + (nt + 1) * 8); w.o_res_b = align16(w.o_res_u + tot * 8);
		if ((r = grow_device(&w.d_res, &w.cap_res, w.o_res_b + tot * 16))) return r;
		if (tot == 0) { memset(w.h_meta + w.o_hres, 0, 2 * (nt + 1) * 8); return 0; }
		HIP_TRY(hipMemcpyAsync(w.d_in + o_off, w.h_meta, meta_bytes, hipMemcpyHostToDevice, w.st));
		HIP_TRY(hipMemcpyAsync(w.d_in, a0 + (h_offsets[k0] - h_offsets[0]), tot * 16, hipMemcpyHostToDevice, w.st));
		int32_t *d_f = (int32_t *)w.d_work, *d_p = d_f + tot;
		mm2c::LaunchArgs L; L.coop_w8_above = G.coop_w8_above.load(); L.fuse_st = G.fuse_st.load();
		L.P = to_kparams(par);
		L.n_tasks = (int64_t)nt; L.d_offsets = (const int64_t *)(w.d_in + o_off); L.d_order = (const int32_t *)(w.d_in + o_ord);
		L.d_anchors = w.d_in; L.d_avg = nullptr; L.d_pbase = nullptr; L.d_status = (int32_t *)(w.d_in + o_stat);
		L.d_f = d_f; L.d_p = d_p; L.d_t = d_p + tot; L.d_st = d_p + 2 * tot;
		L.ring_class = G.ring_class; L.force_tab = G.force_tab; L.compact = G.compact_ring; L.q24 = G.q24_ring; L.noskip_loop = G.noskip_loop;
		L.d_cls = (uint8_t *)(w.d_in + o_cls); L.d_cls_stat = (unsigned long long *)(w.d_in + o_clstat);       // the prepass classes, as plans have them
		L.far_ring = G.far_ring; L.far_thr10 = G.far_thr10; L.wide_pct = G.wide_pct;
		if (mp > 0 && mp <= (size_t)INT32_MAX) {
			char *b = w.d_work;
			L.cut.max_pieces = (int64_t)mp; L.cut.seg_min = G.seg_min; L.cut.min_anchors = G.plan_cut_min;
			L.cut.d_count = (int32_t *)(b + o_cut); L.cut.d_status = (int32_t *)(b + o_cstat); L.cut.d_has_cut = (int32_t *)(b + o_chc);
			L.cut.d_start = (int64_t *)(b + o_cstart); L.cut.d_end = (int64_t *)(b + o_cend); L.cut.d_pbase = (int32_t *)(b + o_cpb); L.cut.d_avg = (float *)(b + o_cavg);
			L.cut.d_cls = (uint8_t *)(b + o_ccls);
			HIP_TRY(hipMemsetAsync(b + o_cut, 0, o_cstart - o_cut, w.st));          // count, status, has_cut
			if (G.coop_plans.load() == 2 && G.coop_waves.load() > 1) L.coop_waves = -1;   // few long pieces -> several waves per piece, decided on the device (chain_route)
		} else if (G.coop_plans.load() == 2 && G.coop_waves.load() > 1) {
			int64_t longest = 0;
			for (size_t k = 0; k < nt; ++k) longest = std::max<int64_t>(longest, m_off[k + 1] - m_off[k]);
			if (mm2c::coop_pays((int64_t)nt, longest, (int64_t)tot, G.coop_w8_above.load())) { L.coop_waves = G.coop_waves.load(); L.max_task_anchors = longest; }
		}
		HIP_TRY(mm2c::launch_chain_dp(L, w.st, &nl, nullptr, &c->last_info));
		note_host_variant(c->last_info);
		E.n_tasks = (int64_t)nt; E.total = (int64_t)tot; E.d_off = L.d_offsets; E.d_order = L.d_order;
		E.d_a = (const ulonglong2 *)w.d_in; E.d_f = d_f; E.d_p = d_p; E.min_cnt = min_cnt; E.min_sc = min_sc;
		E.debug_phases = epilogue_debug_phases();
		E.fused = G.epi_fused.load(); E.max_task = -1;
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

int mm2c_last_host_variant(char *buf, size_t len)
{
	if (!buf || len == 0) return fail(MM2C_E_ARG, "NULL argument");
	std::lock_guard<std::mutex> lk(g_variant_mu);
	if (!g_have_host_info) return fail(MM2C_E_ARG, "no host-buffer entry has launched the DP yet");
	format_variant(g_last_host_info, buf, len);
	return 0;
}

int mm2c_chain_task_host_pred(const mm2c_params_t *par, int64_t n, const mm2c_anchor_t *a, float avg_qspan_scaled,
                              int32_t *f, int32_t *p, int tid, float hw_time_pred, float sw_time_pred)
{
	if (n == 0) return 0;                                                                                   // chain_hardware.cpp:30-32
	if (!lib_ready()) return fail_not_ready();          // (before the protocol: "declined" must never be the answer of a library that has no device -- that would be a silent CPU route)
	if (n < 0) return fail(MM2C_E_ARG, "n < 0");
	if (!G.decline_when_busy.load() || !(hw_time_pred > 0.f && sw_time_pred > 0.f)) return mm2c_chain_task_host(par, n, a, avg_qspan_scaled, f, p, tid);
	const int slot = book_pred(tid, hw_time_pred, sw_time_pred);
	if (slot < 0) { ++g_declined; return 1; }                                                                // chain_hardware.cpp:75
	const int64_t off[2] = { 0, n };
	const int rc = chain_batch_host_tid(par, 1, off, a, &avg_qspan_scaled, f, p, tid, slot);                // the slot that accepted the call is the slot that runs it
	release_pred(slot, hw_time_pred);
	return rc;
}

int mm2c_route_slot(int n_slots, const int64_t *outstanding, int tid)
{
	if (n_slots <= 1 || !outstanding) return 0;
	const int first = (int)((unsigned)(tid < 0 ? 0 : tid) % (unsigned)n_slots);
	int best = first;
	int64_t least = INT64_MAX;
	for (int k = 0; k < n_slots; ++k) {
		const int s = (first + k) % n_slots;
		if (outstanding[s] < least) { least = outstanding[s]; best = s; }
	}
	return best;
}

int mm2c_get_slot_stats(int slot, mm2c_slot_stats_t *out)
{
	if (!out) return fail(MM2C_E_ARG, "NULL argument");
	if (!lib_ready()) return fail_not_ready();
	if (slot < 0 || slot >= n_devices()) return fail(MM2C_E_ARG, "device slot %d of %d", slot, n_devices());
	memset(out, 0, sizeof(*out));
	out->device = G.devices[(size_t)slot];
	get_slot_stats(slot, &out->passes, &out->calls, &out->anchors);
	out->declined = slot == 0 ? g_declined.load() : 0;
	return 0;
}

int mm2c_chain_task_host(const mm2c_params_t *par, int64_t n, const mm2c_anchor_t *a, float avg_qspan_scaled,
                         int32_t *f, int32_t *p, int tid)
{
	// tid: the kt_for worker id (map.c:427,449).  The reference uses it for its FIFO (chain_hardware.cpp:65,83); here it is where the scan over the device slots
	// starts (idle devices are taken in turn, thread k first looks at device k % n)
	if (n == 0) return 0;                                                                                   // chain_hardware.cpp:30-32
	if (n < 0) return fail(MM2C_E_ARG, "n < 0");
	const int64_t off[2] = { 0, n };
	return chain_batch_host_tid(par, 1, off, a, &avg_qspan_scaled, f, p, tid);
}

} // extern "C"
