// chain_kernel.hip -- the chaining DP for MI355X (gfx950 / CDNA4), hand-written HIP.
//
// Computes, for every anchor i of every task, f[i] (best chain score ending at i) and p[i] (its
// predecessor) exactly as the reference's CPU loop does (kisarur/minimap2-fpga chain.c:184-238, twin copy
// at :113-163), including the max_skip early exit (chain.c:226-233).  The same kernel, run with
// max_skip = INT_MAX, max_iter = 1024, one q_span and segments ignored, reproduces what the reference's
// FPGA kernel computes (device/minimap2_opencl.cl:24-172).
//
// Mapping (MI355X-first, not a translation of the FPGA's 1025-deep shift registers or of the CPU loop):
//   * one 64-lane wavefront = one workgroup = one chaining task; thousands of tasks are resident at once
//     (up to 32 waves per CU x 256 CUs), which is where the throughput comes from: the recurrence is
//     sequential in i, so a task can only use the parallelism across its look-back window.
//   * the look-back window of anchor i is scanned nearest-first in chunks of 64 predecessors, lane L of
//     chunk c holding j = i-1-64c-L, i.e. ascending lane = the reference's scan order.
//   * chunk 0 (the 64 nearest predecessors) lives in VGPRs and is shifted one lane per anchor with a
//     DPP wave_shr:1, so the i -> i+1 dependency never goes through memory.
//   * chunks 1.. read an LDS ring (x, q | f, p as two ds_read_b64 per lane, conflict-free); anchors enter
//     the ring in coalesced 1 KiB tiles (one global_load_dwordx4 per lane per 64 anchors, next tile
//     prefetched while the current one is processed).  Only look-back beyond the ring goes to L2/HBM.
//   * the max_skip rule is order dependent.  Per chunk it is evaluated with two DPP prefix scans:
//     a prefix max (which lanes raise the running best) and a max-plus scan of the skip counter
//     (n -> max(n-1,0) on a new best, n -> n+1 on a "predecessor already on a visited chain" event;
//     both are of the form n -> max(n+a, b) and compose), then the first lane whose counter exceeds
//     max_skip is the reference's `break`.
//   * f[] and p[] leave in coalesced 256 B stores per 64 anchors.
//
// Floating point: (int)(dd * avg_qspan_scaled) is an f32 multiply then truncation (chain.c:213,218) and
// (int)((double)gap * gap_scale + .499) is an f64 multiply THEN add (chain.c:219).  This file is compiled
// with -ffp-contract=off and uses __dmul_rn/__dadd_rn so no FMA is ever formed.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "chain_kernel.h"

namespace mm2c {

// ---------------------------------------------------------------- wave64 primitives (DPP, gfx9 encodings)
// dpp_ctrl: row_shr:n = 0x110+n, wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143
__device__ __forceinline__ int wave_shr1(int lane0_value, int v)
{
	return __builtin_amdgcn_update_dpp(lane0_value, v, 0x138, 0xf, 0xf, false);
}

// inclusive prefix max over ascending lanes (6 v_max_i32_dpp)
__device__ __forceinline__ int prefix_max_incl(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x111, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x112, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x114, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x118, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x142, 0xa, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x143, 0xc, 0xf, false));
	return v;
}

__device__ __forceinline__ uint64_t ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ int lanes_below(uint64_t m)   // number of set bits of m in lanes below this one
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

struct Carry { int best, best_j, n_skip; };

// ---------------------------------------------------------------- score of one (i, j) pair, chain.c:199-219
// dr = x_i - x_j (in-window, so 0 <= dr <= max_dist_x < 2^31), dq = q_i - q_j.  Returns validity; sc gets the
// score WITHOUT f[j].
template <bool GEN, bool GS1>
__device__ __forceinline__ bool pair_score(const KParams &P, float avg, int dr, int dq, bool same, int span_i, int &sc)
{
	bool ok;
	const int dd = dr > dq ? dr - dq : dq - dr;                       // chain.c:204
	if (GEN) {
		ok = !((same && dr == 0) || dq <= 0);                         // chain.c:202
		ok = ok && !((same && dq > P.max_dist_y) || dq > P.max_dist_x); // chain.c:203
		ok = ok && !(same && dd > P.bw);                              // chain.c:205
		ok = ok && !(P.n_segs > 1 && !P.is_cdna && same && dr > P.max_dist_y); // chain.c:206
	} else {
		ok = dr != 0 && dq > 0 && dq <= P.max_dq && dd <= P.bw;       // same segment, not cDNA
	}
	int s = min(min(dq, dr), span_i);                                 // chain.c:207-208
	const int lg = dd ? 31 - __builtin_clz((unsigned)dd) : 0;         // chain.c:209 (ilog2_32 == 31-clz, chain.c:15-27)
	const int lin = (int)((float)dd * avg);                           // f32 multiply, truncate
	int gap;
	if (GEN) {
		if (P.is_cdna || !same) {                                     // chain.c:211-217
			if (!same && dr == 0) { ++s; gap = 0; }
			else if (dr > dq || !same) gap = min(lin, lg);
			else gap = lin + (lg >> 1);
		} else gap = lin + (lg >> 1);
	} else gap = lin + (lg >> 1);                                     // chain.c:218
	if (GS1) s -= gap;                                                // (int)((double)g*1.0+.499) == g for g >= 0
	else s -= (int)__dadd_rn(__dmul_rn((double)gap, (double)P.gap_scale), .499); // chain.c:219
	sc = s;
	return ok;
}

// ---------------------------------------------------------------- one chunk of 64 predecessors
// Lane L holds predecessor j = jtop - L.  `inwin` says the lane is inside the look-back window.
// Returns true when the reference loop would have executed `break` inside this chunk.
template <int R, bool SKIP, bool GEN, bool GS1, bool FAR>
__device__ __forceinline__ bool eval_chunk(const KParams &P, float avg, int lane, int i, int jtop, int lo, int lds_lo,
                                           bool inwin, int dr, int dq, bool same, int span_i, int fj, int pj,
                                           int *s_t, int32_t *t_glob, Carry &c)
{
	int sc;
	bool valid = pair_score<GEN, GS1>(P, avg, dr, dq, same, span_i, sc) && inwin;
	sc += fj;                                                         // chain.c:220
	const int scv = valid ? sc : INT_MIN;
	const int incl = prefix_max_incl(scv);
	int last = 63;                                                    // last lane the reference visits in this chunk
	bool broke = false;

	if (SKIP) {
		const int j = jtop - lane;
		const int stamp = i + 1;                                      // t[] holds i+1, 0 = never stamped (chain.c:46 memset)
		// chain.c:233: every visited, unfiltered j stamps its own predecessor.  Stamps for targets outside
		// the window are never read for this i, so they are dropped (keeps ring slots unaliased).
		const bool do_mark = valid && pj >= lo;
		bool far_mark = false;
		if (do_mark) {
			if (!FAR || pj >= lds_lo) s_t[pj & (R - 1)] = stamp;
			else { __hip_atomic_store(&t_glob[pj], stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); far_mark = true; }
		}
		if (FAR && ballot64(far_mark)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		int tj;
		if (!FAR || j >= lds_lo) tj = s_t[j & (R - 1)];
		else tj = inwin ? __hip_atomic_load(&t_glob[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
		const bool marked = (tj == stamp);                            // chain.c:229 `t[j] == i`

		const int run = max(c.best, wave_shr1(INT_MIN, incl));        // best before this lane, in scan order
		const bool nm = valid && sc > run;                            // chain.c:226 takes the branch
		const bool se = valid && !nm && marked;                       // chain.c:229-230 `++n_skip`
		const uint64_t nmm = ballot64(nm), sem = ballot64(se);
		if (sem != 0) {
			// skip counter after each lane: Lindley recursion n <- max(n + d, 0), d = +1 (se) / -1 (nm)
			const int S = lanes_below(sem) - lanes_below(nmm) + (se ? 1 : 0) - (nm ? 1 : 0);
			const int nl = S + max(c.n_skip, prefix_max_incl(-S));
			const uint64_t brk = ballot64(se && nl > P.max_skip);     // chain.c:230-231
			if (brk != 0) { last = (int)__builtin_ctzll(brk) - 1; broke = true; }
			else c.n_skip = rdlane(nl, 63);
		} else {
			c.n_skip = max(c.n_skip - (int)__builtin_popcountll(nmm), 0);
		}
	}
	if (last >= 0) {
		const int mc = rdlane(incl, last);                            // best over the visited lanes of this chunk
		if (mc > c.best) {                                            // strict: nearest j wins ties (chain.c:226)
			const uint64_t eq = ballot64(valid && sc == mc);
			c.best = mc;
			c.best_j = jtop - (int)__builtin_ctzll(eq);
		}
	}
	return broke;
}

// ---------------------------------------------------------------- the kernel: one wave per task
template <int R, bool SKIP, bool GEN, bool GS1, bool FAR>
__global__ void __launch_bounds__(64)
chain_dp_wave(KParams P, int64_t n_tasks, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
              const uint4 *__restrict__ a_all, const float *__restrict__ avg_in,
              int32_t *__restrict__ f_all, int32_t *__restrict__ p_all, int32_t *__restrict__ t_all,
              int32_t *__restrict__ status, int only_flagged)
{
	static_assert(R >= 128 && (R & (R - 1)) == 0, "ring must be a power of two >= 128");
	__shared__ uint2 s_xq[R];     // x low word, query position
	__shared__ int2 s_fp[R];      // f, p
	__shared__ int s_t[R];        // stamps (chain.c t[]), i+1
	__shared__ uint8_t s_g[GEN ? R : 64];

	const int lane = threadIdx.x;
	const int64_t task = order ? (int64_t)order[blockIdx.x] : (int64_t)blockIdx.x;
	if (task >= n_tasks) return;
	if (only_flagged && status[task] == 0) return;
	const int64_t base = offsets[task];
	const int n = (int)(offsets[task + 1] - base);
	if (n <= 0) return;
	const uint4 *a = a_all + base;         // {x lo, x hi, y lo (= query pos), y hi (span | flags | seg)}
	int32_t *f = f_all + base, *p = p_all + base, *t = FAR ? t_all + base : nullptr;

	for (int s = lane; s < R; s += 64) s_t[s] = 0;

	// avg_qspan_scaled, chain.c:48-49
	float avg;
	if (avg_in) avg = avg_in[task];
	else {
		uint64_t sum = 0;
		for (int k = lane; k < n; k += 64) sum += (a[k].w & 0xffu);
		for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
		avg = (float)(__dmul_rn(.01, (double)(float)sum) / (double)n);
	}

	const uint32_t D = (uint32_t)P.max_dist_x;
	uint32_t wx = 0; int wq = 0, wf = 0, wp = -1, wg = 0;   // chunk-0 window: lane L = anchor i-1-L
	uint32_t run_hi = 0; int hs = 0;                          // start of the run of anchors sharing x's high word
	int seg0 = 0;

	uint4 cur = (lane < n) ? a[lane] : make_uint4(0, 0, 0, 0);
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int idx = i0 + lane;
		const int cnt = min(64, n - i0);
		uint4 nxt = (idx + 64 < n) ? a[idx + 64] : make_uint4(0, 0, 0, 0);   // prefetch the next tile
		const int g_l = (cur.w >> 16) & 0xff;                                 // MM_SEED_SEG_MASK mmpriv.h:22-23
		if (!GEN && !(P.flags & KF_IGNORE_SEG)) {
			// the simple variant assumes one segment id per task; anything else is redone by the general one
			if (i0 == 0) seg0 = rdlane(g_l, 0);
			if (ballot64(lane < cnt && g_l != seg0)) { if (lane == 0) status[task] = 1; return; }
		}
		// the tile enters the ring (slots of anchors idx-R are recycled)
		s_xq[idx & (R - 1)] = make_uint2(cur.x, cur.z);
		if (GEN) s_g[idx & (R - 1)] = (uint8_t)g_l;
		if (FAR && idx < n) t[idx] = 0;
		const int lds_lo = i0 + 64 - R;       // oldest anchor index still in the ring while this tile is processed
		int tf = 0, tp = -1;

		for (int k = 0; k < cnt; ++k) {
			const int i = i0 + k;
			const uint32_t xi = (uint32_t)rdlane((int)cur.x, k), xhi = (uint32_t)rdlane((int)cur.y, k);
			const int qi = rdlane((int)cur.z, k);
			const uint32_t yhi = (uint32_t)rdlane((int)cur.w, k);
			const int span_i = P.span_override >= 0 ? P.span_override : (int)(yhi & 0xff);   // chain.c:189
			const int seg_i = (yhi >> 16) & 0xff;                                               // chain.c:191
			if (i == 0 || xhi != run_hi) { run_hi = xhi; hs = i; }
			// chain.c:192-193: st = max(first j with x_i <= x_j + max_dist_x, i - max_iter); the x bound is
			// applied per lane below (dr <= D), hs keeps the 32-bit difference exact.
			const int lo = max(hs, (int)max((int64_t)i - (int64_t)P.max_iter, (int64_t)0));
			Carry c = { span_i, -1, 0 };                                                         // chain.c:188-190
			int jtop = i - 1;
			if (jtop >= lo) {
				// ---- chunk 0 from registers
				const int j0 = jtop - lane;
				uint32_t dr = xi - wx;
				bool inwin = j0 >= lo && dr <= D;
				bool more = ballot64(inwin) == ~0ull;
				bool broke = eval_chunk<R, SKIP, GEN, GS1, FAR>(P, avg, lane, i, jtop, lo, lds_lo, inwin, (int)dr, qi - wq,
				                                                 GEN ? (wg == seg_i) : true, span_i, wf, wp, s_t, t, c);
				jtop -= 64;
				// ---- older chunks from the LDS ring (and from L2/HBM beyond it)
				while (more && !broke && jtop >= lo) {
					const int j = jtop - lane;
					uint2 xq = make_uint2(0, 0); int2 fp = make_int2(0, -1); int gj = 0;
					const bool near_all = !FAR || max(jtop - 63, lo) >= lds_lo;
					if (near_all || j >= lds_lo) {
						xq = s_xq[j & (R - 1)]; fp = s_fp[j & (R - 1)];
						if (GEN) gj = s_g[j & (R - 1)];
					} else if (j >= lo) {
						const uint4 aj = a[j];
						xq = make_uint2(aj.x, aj.z);
						if (GEN) gj = (aj.w >> 16) & 0xff;
						fp.x = __hip_atomic_load(&f[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						fp.y = __hip_atomic_load(&p[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
					dr = xi - xq.x;
					inwin = j >= lo && dr <= D;
					more = ballot64(inwin) == ~0ull;
					broke = eval_chunk<R, SKIP, GEN, GS1, FAR>(P, avg, lane, i, jtop, lo, lds_lo, inwin, (int)dr, qi - (int)xq.y,
					                                            GEN ? (gj == seg_i) : true, span_i, fp.x, fp.y, s_t, t, c);
					jtop -= 64;
				}
			}
			// ---- commit anchor i (chain.c:236): tile registers, LDS ring, chunk-0 window
			if (lane == k) { tf = c.best; tp = c.best_j; }
			if (lane == 0) s_fp[i & (R - 1)] = make_int2(c.best, c.best_j);
			wx = (uint32_t)wave_shr1((int)xi, (int)wx);
			wq = wave_shr1(qi, wq);
			wf = wave_shr1(c.best, wf);
			wp = wave_shr1(c.best_j, wp);
			if (GEN) wg = wave_shr1(seg_i, wg);
		}
		if (idx < n) { f[idx] = tf; p[idx] = tp; }     // coalesced 256 B stores
		cur = nxt;
	}
}

// ---------------------------------------------------------------- host-side launcher
template <int R, bool SKIP, bool GEN, bool GS1, bool FAR>
static hipError_t launch_one(const LaunchArgs &L, hipStream_t st, int only_flagged)
{
	hipLaunchKernelGGL((chain_dp_wave<R, SKIP, GEN, GS1, FAR>), dim3((unsigned)L.n_tasks), dim3(64), 0, st,
	                   L.P, L.n_tasks, L.d_offsets, L.d_order, (const uint4 *)L.d_anchors, L.d_avg, L.d_f, L.d_p, L.d_t,
	                   L.d_status, only_flagged);
	return hipGetLastError();
}

template <int R, bool GEN, bool FAR>
static hipError_t launch_sg(const LaunchArgs &L, hipStream_t st, bool skip, bool gs1, int only_flagged)
{
	if (skip) return gs1 ? launch_one<R, true, GEN, true, FAR>(L, st, only_flagged) : launch_one<R, true, GEN, false, FAR>(L, st, only_flagged);
	return gs1 ? launch_one<R, false, GEN, true, FAR>(L, st, only_flagged) : launch_one<R, false, GEN, false, FAR>(L, st, only_flagged);
}

template <int R>
static hipError_t launch_r(const LaunchArgs &L, hipStream_t st, bool skip, bool gen, bool gs1, bool far_, int only_flagged)
{
	if (gen) return far_ ? launch_sg<R, true, true>(L, st, skip, gs1, only_flagged) : launch_sg<R, true, false>(L, st, skip, gs1, only_flagged);
	return far_ ? launch_sg<R, false, true>(L, st, skip, gs1, only_flagged) : launch_sg<R, false, false>(L, st, skip, gs1, only_flagged);
}

int chain_ring_anchors(int ring_class) { return ring_class == 0 ? 256 : ring_class == 1 ? 512 : 1024; }

hipError_t launch_chain_dp(const LaunchArgs &L, hipStream_t st, int *n_launches)
{
	const KParams &P = L.P;
	if (L.n_tasks <= 0) return hipSuccess;
	// the early exit can never fire when the skip counter cannot exceed max_skip inside one window
	const bool skip = (int64_t)P.max_skip < (int64_t)P.max_iter;
	const bool gs1 = P.gap_scale == 1.0f;
	const bool want_gen = P.is_cdna || P.n_segs > 1 || (P.flags & KF_FORCE_GENERAL);
	const int R = chain_ring_anchors(L.ring_class);
	const bool far_ = (int64_t)P.max_iter + 64 > (int64_t)R;
	hipError_t e = hipSuccess;
	for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
		// pass 0: the variant the parameters ask for; pass 1 (simple variant only, segments not ignored): redo the
		// tasks that turned out to carry more than one segment id with the general variant.
		const bool gen = want_gen || pass == 1;
		if (pass == 1 && (want_gen || (P.flags & KF_IGNORE_SEG))) break;
		const int flagged = pass;
		switch (R) {
		case 256: e = launch_r<256>(L, st, skip, gen, gs1, far_, flagged); break;
		case 512: e = launch_r<512>(L, st, skip, gen, gs1, far_, flagged); break;
		default:  e = launch_r<1024>(L, st, skip, gen, gs1, far_, flagged); break;
		}
		if (n_launches) ++*n_launches;
	}
	return e;
}

} // namespace mm2c
