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
//   * the window start of every anchor (chain.c:192-193) comes from a prepass kernel (chain_window_start, 64-bit
//     binary searches over x values staged in LDS), so the DP works on low words of x and knows each window exactly.
//   * plans first cut long tasks at empty windows into independent pieces (chain_cut): many waves instead of one long
//     dependent chain for a read made of many loci.
//   * chunks 1.. read an LDS ring of R anchors (x, q | f, p as two ds_read_b64 per lane, conflict-free); anchors
//     arrive in coalesced 1 KiB tiles (one global_load_dwordx4 per lane per 64 anchors, next tile prefetched while
//     the current one is processed) and a finished tile enters the ring in one shot, so the ring always holds the R
//     anchors before the current tile.  The NV tiles that left the ring last stay in VGPRs ("victim" tiles) and are
//     read across lanes with ds_bpermute.  Only look-back beyond that goes to L2/HBM, and there f/p/stamps are
//     fetched only for lanes that passed the filters.
//   * chain.c's t[] (stamps "predecessor already on a visited chain") is a 16-bit stamp ring in LDS covering 2R
//     anchors (scatter by p[j], gather by j, cleared as anchors enter), 32-bit stamps in a global scratch beyond it.
//   * the kernel is integer-issue bound (measured: SQ_ACTIVE_INST_VALU and SQ_ACTIVE_INST_SCA each ~80 % of all SIMD cycles,
//     both 4 cycles per wave64 instruction), so the instruction stream is kept lean: lane predicates live as
//     64-bit masks in SGPRs (one v_cmp each, combined on the scalar unit), the window bound j >= lo is a
//     scalar-built lane mask, a chunk with no lane passing the filters (chain.c:202-206) skips scoring, an older
//     chunk with no score above the running best skips the scan.
//   * the max_skip rule is order dependent.  Per chunk it is evaluated with a DPP prefix max (which lanes
//     raise the running best) and, only when a skip event interleaves with a new best, a max-plus scan of
//     the skip counter (n -> max(n-1,0) on a new best, n -> n+1 on a "predecessor already on a visited
//     chain" event; both are of the form n -> max(n+a, b) and compose); the first lane whose counter
//     exceeds max_skip is the reference's `break`.
//   * f[] and p[] leave in coalesced 256 B stores per 64 anchors.
//
// Floating point: (int)(dd * avg_qspan_scaled) is an f32 multiply then truncation (chain.c:213,218) and
// (int)((double)gap * gap_scale + .499) is an f64 multiply THEN add (chain.c:219).  This file is compiled
// with -ffp-contract=off and uses __dmul_rn/__dadd_rn so no FMA is ever formed.

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <algorithm>
#include <stdint.h>
#include <limits.h>
#include <stdlib.h>
#include "chain_kernel.h"
#include "chain_wave.h"
#ifndef MM2C_NX
#define MM2C_NX 8        // tiles of x / q in the LDS ring of the tile kernel: 384 anchors before the own tile pass the filters without global memory
#define MM2C_NF 2        // tiles of f / p beside them (deeper ones are fetched from L2 for the lanes that passed)
#endif
#ifndef MM2C_NF1
#define MM2C_NF1 2       // ... of the instantiation with the long x / q ring (ring-size class 1)
#endif
// the instantiation with the compact x / q ring (4 bytes per ring anchor instead of 8): 16 tiles cost the LDS that 8 cost in the 32-bit form, so there is one
// ring size for every task that takes it (measured, ms headline / dense / asm20 mixed: 8 tiles + 2 of f / p 46.6 / 59.6 / 94.7, 16 + 2 44.0 / 59.9 / 80.0,
// 8 + 4 44.3 / 62.2 / 87.1, 16 + 4 46.3 / 62.3 / 82.7; the 32-bit rings with ring-size classes 45.7 / 75.2 / 93.9)
#ifndef MM2C_CNX
#define MM2C_CNX 16
#endif
#ifndef MM2C_CNF
#define MM2C_CNF 2
#endif
#include "chain_dp_tile.h"
#include "chain_dp_coop.h"

namespace mm2c {

// ---------------------------------------------------------------- chain.c:233 + :229 for a chunk whose lanes are all ring-resident
// Every visited, unfiltered j stamps its predecessor p[j] (stamps for targets outside the window are never read for
// this i and are dropped); then each lane fetches its own stamp.  LDS stamps are 16 bit, s16 = 1 + i % 16384 (0 = never
// stamped, chain.c:46), in a ring of 2R slots that is cleared as anchors enter it, so a value identifies its anchor.
// Lanes that do not stamp write to the slot of anchor i itself (see below).  The stamp ring holds the 2R anchors before the end of the current
// tile (stamp_lo = i0 + 64 - 2R), twice the data ring; a stamp for an older target (the full i+1) goes to the global scratch t[].
template <int R, bool FAR>
__device__ __forceinline__ int stamp_and_fetch(mask_t valid, int pj, int lo, int stamp_lo, int stamp, int s16_v, char *t_bytes,
                                               int32_t *t_glob, int lane, int own_off2)
{
	const mask_t mk = valid & BALLOT(pj >= lo);
	// lanes that do not stamp write into the slot of anchor i itself (stamp = i + 1): it is not read during i's own scan, and what lands
	// there (s16 of i) can never equal the s16 of a later anchor within the ring's reach -- so the ring needs no extra sink slot and the
	// kernel's LDS is exactly 5 KB at R = 256 (32 waves per CU instead of 29)
	const int sink = ((stamp - 1) & (2 * R - 1)) << 1;
	int tgt = sel(mk, sink, (pj & (2 * R - 1)) << 1);
	if (FAR) {
		const mask_t fm = mk & BALLOT(pj < stamp_lo);
		if (fm != 0) {
			int pj2 = pj;
			asm volatile("" : "+v"(pj2));                             // keep the far addressing out of the hot loop
			if (fm >> lane & 1) __hip_atomic_store(&t_glob[pj2], stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			tgt = sel(fm, tgt, sink);
		}
	}
	*(uint16_t *)(t_bytes + tgt) = (uint16_t)s16_v;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	return *(const uint16_t *)(t_bytes + own_off2);                   // the caller tests it against s16 (chain.c:229 `t[j] == i`)
}

// ---------------------------------------------------------------- the look-back scan of one anchor, chain.c:197-235
struct Win { int x, q, f, p, g; };      // chunk-0 window registers: lane L = anchor i-1-L
// the NV tiles that most recently left the LDS ring, kept in registers ("victim" tiles): tile k covers anchors
// [base - 64k, base - 64k + 64), lane L = anchor base - 64k + 63 - L
#define NV 3
struct Victim { int x[NV], q[NV], f[NV], p[NV], g[NV], base; };

// ---------------------------------------------------------------- one older chunk (j <= i-65): LDS ring, victim tiles, L2/HBM
// Returns true when the reference loop executes `break` inside the chunk.  FULL: all 64 lanes are inside the window.
template <int R, bool SKIP, bool GEN, bool GS1, bool FAR, bool FULL>
__device__ __forceinline__ bool older_chunk(const KParams &P, float avg, int lane, int i, int lo, int lds_lo, int stamp_lo, int stamp, int s16,
                                            int s16_v, int xi, int qi, int span_i, int seg_i, int jtop, int rem, const Victim &vt,
                                            const char *xq_bytes, const char *fp_bytes, const uint8_t *s_g, char *t_bytes, uint16_t *s_t,
                                            const uint4 *a, const int32_t *f, const int32_t *p, int32_t *t, int pbase, Carry &c)
{
	const int nl8 = -8 * lane, nl2 = -2 * lane;
	const int off8 = ((jtop << 3) + nl8) & ((R - 1) << 3);
	const uint2 xq = *(const uint2 *)(xq_bytes + off8);
	const int2 fp = *(const int2 *)(fp_bytes + off8);
	int xj = (int)xq.x, qj = (int)xq.y, fj = fp.x, pj = fp.y, gj = 0;
	if (GEN) gj = s_g[off8 >> 3];
	const int own_off2 = ((jtop << 1) + nl2) & ((2 * R - 1) << 1);
	const mask_t in_w = FULL ? ~0ull : first_lanes(rem);   // only the last chunk of a window is partial
	mask_t far_l = 0;                                     // lanes whose predecessor left the ring
	if (FAR && jtop - 63 < lds_lo && lo < lds_lo) {
		int j = jtop - lane;
		asm volatile("" : "+v"(j));                       // keep the far addressing out of the hot loop
		const mask_t out_l = BALLOT(j < lds_lo) & in_w;       // lanes beyond the ring
		// the 64*NV anchors just beyond the ring are still in registers: fetch them across lanes (no memory access)
		far_l = out_l;
#pragma unroll
		for (int k = 0; k < NV; ++k) {
			const int rel = j - (vt.base - 64 * k);            // position inside victim tile k
			const mask_t vic = out_l & BALLOT((unsigned)rel < 64u);
			if (vic != 0) {
				const int src4 = (63 - rel) << 2;
				xj = sel(vic, xj, __builtin_amdgcn_ds_bpermute(src4, vt.x[k]));
				qj = sel(vic, qj, __builtin_amdgcn_ds_bpermute(src4, vt.q[k]));
				fj = sel(vic, fj, __builtin_amdgcn_ds_bpermute(src4, vt.f[k]));
				pj = sel(vic, pj, __builtin_amdgcn_ds_bpermute(src4, vt.p[k]));
				if (GEN) gj = sel(vic, gj, __builtin_amdgcn_ds_bpermute(src4, vt.g[k]));
				far_l &= ~vic;                                  // what is left goes to L2/HBM
			}
		}
		if (far_l >> lane & 1) {
			const uint4 aj = a[j];
			xj = (int)aj.x; qj = (int)aj.z;
			if (GEN) gj = (aj.w >> 16) & 0xff;
		}
	}
	const int dr = xi - xj, dq = qi - qj;
	const int dd = absdiff(dr, dq);
	const mask_t same = GEN ? BALLOT(gj == seg_i) : ~0ull;
	const mask_t valid = pair_filter<GEN>(P, in_w, dr, dq, dd, same);
	if (valid == 0) return false;
	{
		mask_t marked = 0;
		if (FAR && far_l != 0) {
			// look-back beyond the ring: f, p and stamps from L2/HBM, only for lanes that passed the filters
			int j = jtop - lane;
			asm volatile("" : "+v"(j));
			const bool fl = (far_l & valid) >> lane & 1;
			if (fl) {
				fj = __hip_atomic_load(&f[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				pj = __hip_atomic_load(&p[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				if (pj >= 0) pj -= pbase;             // p[] in memory is relative to the caller's task, the scan works piece-relative
			}
			if (SKIP) {
				const bool mkv = (valid >> lane & 1) && pj >= lo;
				if (mkv) {
					if (pj >= stamp_lo) s_t[pj & (2 * R - 1)] = (uint16_t)s16;
					else __hip_atomic_store(&t[pj], stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // far stamps of this and earlier chunks have landed
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				int tj = 0;
				if (fl && j < stamp_lo) tj = __hip_atomic_load(&t[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				else tj = s_t[j & (2 * R - 1)] == s16 ? stamp : 0;
				marked = BALLOT(tj == stamp);
			}
		}
		int tj = 0;
		const bool near_stamps = SKIP && !(FAR && far_l != 0);
		if (near_stamps) tj = stamp_and_fetch<R, FAR>(valid, pj, lo, stamp_lo, stamp, s16_v, t_bytes, t, lane, own_off2);
		const int sc = pair_score<GEN, GS1>(P, avg, dr, dq, dd, same, span_i) + fj;
		const int scv = sel(valid, SENT, sc);
		if (near_stamps) marked = BALLOT(tj == s16);
		return fold_chunk<SKIP, true>(P, jtop, valid, marked, scv, c);
	}
	return false;
}

template <int R, bool SKIP, bool GEN, bool GS1, bool FAR>
__device__ __forceinline__ void scan_window(const KParams &P, float avg, int lane, int i, int lo, int lds_lo, int xi, int qi, int span_i,
                                            int seg_i, const Win &w, const Victim &vt, const char *xq_bytes, const char *fp_bytes, const uint8_t *s_g,
                                            char *t_bytes, uint16_t *s_t, const uint4 *a, const int32_t *f, const int32_t *p,
                                            int32_t *t, int pbase, Carry &c)
{
	const int nl2 = -2 * lane;                                // ring byte offsets go down with the lane
	const int wx = w.x, wq = w.q, wf = w.f, wp = w.p, wg = w.g;
	int rem = i - lo;                  // predecessors still to visit (> 0)
	{
		const int stamp = i + 1;              // stamp in the global scratch t[] (look-back beyond the ring)
		const int s16 = 1 + (i & 0x3fff);     // stamp in the LDS ring: unique over the < 2R+64 anchors that can stamp a slot between two clears
		const int stamp_lo = lds_lo + 64 - R;  // = i0 + 64 - 2R: oldest anchor whose stamp slot is in the LDS ring
		const int s16_v = s16;                 // (one v_mov: the ds_write data operand)
		int jtop = i - 1;
		asm volatile("" : "+s"(jtop));      // opaque: keeps per-lane ring offsets out of the per-anchor loop (computed where used)
		bool broke = false;
		// ---------------- chunk 0 from registers
		{
			const int dr = xi - wx, dq = qi - wq;
			const int dd = absdiff(dr, dq);
			const mask_t same = GEN ? BALLOT(wg == seg_i) : ~0ull;
			const mask_t valid = pair_filter<GEN>(P, first_lanes(rem), dr, dq, dd, same);
			if (valid != 0) {
				int tj = 0;                    // stamp round trip through LDS overlaps the scoring below
				if (SKIP) tj = stamp_and_fetch<R, FAR>(valid, wp, lo, stamp_lo, stamp, s16_v, t_bytes, t, lane,
				                                        (((jtop << 1) + nl2) & ((2 * R - 1) << 1)));
				const int sc = pair_score<GEN, GS1>(P, avg, dr, dq, dd, same, span_i) + wf;   // chain.c:220
				const int scv = sel(valid, SENT, sc);
				const mask_t marked = SKIP ? BALLOT(tj == s16) : 0;
				broke = fold_chunk<SKIP, false>(P, jtop, valid, marked, scv, c);
			}
			jtop -= 64; rem -= 64;
		}
		// ---------------- older chunks from the LDS ring (and from registers / L2 / HBM beyond it): full chunks, then the partial one
		if (broke) return;
		for (; rem >= 64; jtop -= 64, rem -= 64)
			if (older_chunk<R, SKIP, GEN, GS1, FAR, true>(P, avg, lane, i, lo, lds_lo, stamp_lo, stamp, s16, s16_v, xi, qi, span_i, seg_i, jtop, rem,
			                                              vt, xq_bytes, fp_bytes, s_g, t_bytes, s_t, a, f, p, t, pbase, c)) return;
		if (rem > 0)
			older_chunk<R, SKIP, GEN, GS1, FAR, false>(P, avg, lane, i, lo, lds_lo, stamp_lo, stamp, s16, s16_v, xi, qi, span_i, seg_i, jtop, rem,
			                                           vt, xq_bytes, fp_bytes, s_g, t_bytes, s_t, a, f, p, t, pbase, c);
	}
}

constexpr int PREPASS_SEG = 32768;   // anchors of a long task per block of the prepass (chain_window_start_t<true>)
// ---------------------------------------------------------------- prepass: window start of every anchor
// st[i] = max(first j of the task with x_i <= x_j + max_dist_x, i - max_iter), chain.c:192-193 (the `st` pointer of the
// reference is monotone, so its value at i is exactly this).  Full 64-bit compares, so inside
// [st[i], i) every x difference fits 31 bits and the DP kernel works on low words.  One 256-thread block per task,
// each lane a binary search over the task's sorted x (L2-resident); O(n log max_iter) and ~1 % of the DP.
// SEG (round 6, long reads): a block walks its task's tiles one after the other, so a batch of 255 tasks of 10^6 anchors had 255 blocks at work for 3 900 steps each (2.6 ms
// of a 4.7 ms prepass, on a GPU that reads the anchors in 0.5).  With SEG a task is `seg_anchors` (a multiple of 256: the tiles, and with them the lanes' neighbours in the
// class count, stay the same) per block, blockIdx.y the segment: the first tiles of a segment search in memory until the window start of the tile before lies inside the
// segment (two tiles, typically), the sums and extremes of a task are added up in `seg_ws` (4 words of 64 bits per task, zero between runs) and the block that arrives
// last writes avg / class and puts the words back to zero.  The answers are the same by construction: same bounds, same condition, integer sums.
template <bool SEG>
__global__ void __launch_bounds__(256)
chain_window_start_t(KParams P, int64_t n_tasks, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
                   const ulonglong2 *__restrict__ a_all, int32_t *__restrict__ st_all, int32_t *__restrict__ has_cut /* per task, or nullptr */,
                   float *__restrict__ avg_out /* per task, or nullptr */, uint8_t *__restrict__ cls_out /* per task, or nullptr */, int far_ring, int far_thr10,
                   unsigned long long *__restrict__ cls_stat /* CLS_STAT_SLOTS sets of [anchors of class-1 tasks, of all tasks, -, of tasks with the 32-bit ring], or nullptr */,
                   unsigned q_span_max /* compact x / q ring: the widest span of q values a task may have (0: no task takes it) */,
                   int q24 /* the long ring is the q24 ring: a task with a q value of 2^24 or more stays out of class 1 (and carries bit 2) */,
                   int seg_anchors, unsigned long long *__restrict__ seg_ws)
{
	const int lane = threadIdx.x;
	const int64_t task = order ? (int64_t)order[blockIdx.x] : (int64_t)blockIdx.x;
	if (task >= n_tasks) return;
	const int64_t base = offsets[task];
	const int n = (int)(offsets[task + 1] - base);
	const int s0 = SEG ? (int)min((long long)blockIdx.y * seg_anchors, (long long)INT_MAX) : 0;
	if (SEG && s0 >= n) return;
	const int s1 = SEG ? (int)min((long long)s0 + seg_anchors, (long long)n) : n;
	const ulonglong2 *a = a_all + base;
	int32_t *st = st_all + base;
	const uint64_t D = (uint64_t)(int64_t)P.max_dist_x;
	// st[] is monotone: the answers of a tile of 256 anchors start at the answer of the tile before.  The x values live in an LDS ring indexed
	// by the anchor (RING of them): a tile adds its own 256 with one coalesced 16-byte load per lane -- requested while the tile before is
	// being searched, the tiles depend on each other only through that one answer -- and every search probes LDS.  (Staging tile + window
	// afresh for every tile read each x two to three times and waited for the loads tile by tile: 2.03 ms on the headline batch.)
	// A window longer than the ring is searched in global memory.
	constexpr int RING = 2048;
	__shared__ uint64_t s_x[RING];
	__shared__ int s_prev;
	__shared__ unsigned long long s_sum;
	__shared__ unsigned long long s_far;
	__shared__ int s_qmin, s_qmax;
	if (lane == 0) { s_prev = 0; s_sum = 0; s_far = 0; s_qmin = INT_MAX; s_qmax = INT_MIN; }
	int q_min = INT_MAX, q_max = INT_MIN;                             // of this lane's anchors (compact ring: bit 1 of the class)
	const int max_dq = min(P.max_dist_x, P.max_dist_y);
	uint64_t far_sum = 0;                                             // expected tiles beyond the short ring, see below
	uint64_t span_sum = 0;                                            // chain.c:48: spans of this lane's anchors
	ulonglong2 nxt = s0 + lane < n ? a[s0 + lane] : make_ulonglong2(0, 0);
	for (int i0 = s0; i0 < s1; i0 += 256) {                        // 4 waves per task (or segment)
		const int i = i0 + lane, cnt = min(256, n - i0);
		const ulonglong2 cur = nxt;
		if (i < n) { s_x[i & (RING - 1)] = cur.x; span_sum += (cur.y >> 32) & 0xff; q_min = min(q_min, (int)(uint32_t)cur.y); q_max = max(q_max, (int)(uint32_t)cur.y); }
		__syncthreads();                                            // the tile is in the ring; s_prev of the tile before is visible
		if (i + 256 < n) nxt = a[i + 256];
		const int range_lo = max(s_prev, max(i0 - P.max_iter, 0)), len = i0 + cnt - range_lo;
		int lo = 0;
		if (i < n) {
			const uint64_t xi = cur.x;
			int hi = i;
			lo = max(i - P.max_iter, range_lo);                       // answer in [lo, hi]; x_i <= x_i + D always holds
			if (len <= RING && (!SEG || range_lo >= s0)) {           // (SEG: the ring holds the segment's own anchors only)
				while (lo < hi) {
					const int mid = (lo + hi) >> 1;
					if (xi > s_x[mid & (RING - 1)] + D) lo = mid + 1; else hi = mid;   // chain.c:192 condition for "++st"
				}
			} else {
				while (lo < hi) {
					const int mid = (lo + hi) >> 1;
					if (xi > a[mid].x + D) lo = mid + 1; else hi = mid;
				}
			}
			st[i] = lo;
			if (has_cut && lo == i && i > 0) has_cut[task] = 1;      // an empty window: the task can be cut here (chain_cut)
		}
		if (cls_out && far_ring == 1) {
			// Ring-size class of the task.  A scan leaves the 64 (NX - 1) anchors of the short ring when the window is longer AND the `break` of
			// chain.c:231 does not come first; the break needs a chain among the nearest predecessors, so an anchor counts with the tiles of its
			// window beyond the ring unless one of its three nearest predecessors passes the filters (chain.c:202-205).  Neighbours in other
			// waves of the block are not looked at (lanes 0-2 of a wave count as "no chain").
			const int beyond = i < n ? max((i & ~63) - 64 * (MM2C_NX - 1) - lo, 0) : 0;
			if (__ballot(beyond > 0) != 0) {                          // (a wave whose windows all fit the short ring has nothing to count)
				bool chain = false;
				const int xi = (int)(uint32_t)cur.x, qi = (int)(uint32_t)cur.y;
#pragma unroll
				for (int d = 1; d <= 3; ++d) {
					const int xj = __shfl_up(xi, d), qj = __shfl_up(qi, d);
					const int dr = xi - xj, dq = qi - qj, dd = dr > dq ? dr - dq : dq - dr;
					chain |= (lane & 63) >= d && i - d >= lo && dr > 0 && dq > 0 && dq <= max_dq && dd <= P.bw;
				}
				if (beyond > 0 && !chain) far_sum += (uint64_t)((beyond + 63) >> 6);
			}
		}
		__syncthreads();                                            // everybody has read s_prev and is done with the ring slots the next tile overwrites
		if (i < n && lane == cnt - 1) s_prev = lo;
	}
	if constexpr (SEG) {
		// the segment's share of the task's sums and extremes -> seg_ws[4 * task ..]: [0] spans, [1] far tiles, [2] low word max of ~(q_min biased), high word max of
		// (q_max biased) (unsigned maxima: zero is "nothing yet"), [3] segments done; the last segment to arrive goes on with the totals
		for (int o = 32; o > 0; o >>= 1) { q_min = min(q_min, __shfl_xor(q_min, o)); q_max = max(q_max, __shfl_xor(q_max, o)); far_sum += __shfl_xor(far_sum, o); span_sum += __shfl_xor(span_sum, o); }
		if ((lane & 63) == 0) { atomicMin(&s_qmin, q_min); atomicMax(&s_qmax, q_max); if (far_sum) atomicAdd(&s_far, (unsigned long long)far_sum); atomicAdd(&s_sum, (unsigned long long)span_sum); }
		__syncthreads();
		__shared__ int s_last;
		unsigned long long *ws = seg_ws + 4 * task;
		if (lane == 0) {
			atomicAdd(&ws[0], s_sum); if (s_far) atomicAdd(&ws[1], s_far);
			atomicMax((unsigned *)&ws[2], ~((unsigned)s_qmin ^ 0x80000000u)); atomicMax((unsigned *)&ws[2] + 1, (unsigned)s_qmax ^ 0x80000000u);
			__threadfence();
			const unsigned n_seg = (unsigned)(((long long)n + seg_anchors - 1) / seg_anchors);
			const bool last = atomicAdd((unsigned *)&ws[3], 1u) == n_seg - 1;
			s_last = last;
			if (last) {
				__threadfence();
				s_sum = atomicAdd(&ws[0], 0ull); s_far = atomicAdd(&ws[1], 0ull);
				s_qmin = (int)(~atomicMax((unsigned *)&ws[2], 0u) ^ 0x80000000u); s_qmax = (int)(atomicMax((unsigned *)&ws[2] + 1, 0u) ^ 0x80000000u);
				ws[0] = 0; ws[1] = 0; ws[2] = 0; ws[3] = 0;                      // zero again for the next run
			}
		}
		__syncthreads();
		if (!s_last) return;
		far_sum = 0; span_sum = 0; q_min = s_qmin; q_max = s_qmax;              // (the tail below adds the lanes' shares to the LDS totals: nothing left to add)
	}
	if (cls_out && n > 0) {
		// bit 1: the task's q values span more than the compact x / q ring tells apart (chain_dp_tile.h, Lds<>): it runs with the 32-bit ring
		for (int o = 32; o > 0; o >>= 1) { q_min = min(q_min, __shfl_xor(q_min, o)); q_max = max(q_max, __shfl_xor(q_max, o)); }
		if ((lane & 63) == 0) { atomicMin(&s_qmin, q_min); atomicMax(&s_qmax, q_max); }
		__syncthreads();
		const int wide = (q_span_max == 0 || (unsigned)s_qmax - (unsigned)s_qmin > q_span_max) ? 2 : 0;
		const int huge = (q24 && (s_qmin < 0 || s_qmax >= (1 << 24))) ? 4 : 0;    // (q as a signed word: a position of 2^31 or more shows as negative)
		if (far_ring == 1) {
			for (int o = 32; o > 0; o >>= 1) far_sum += __shfl_xor(far_sum, o);
			if ((lane & 63) == 0 && far_sum) atomicAdd(&s_far, (unsigned long long)far_sum);
			__syncthreads();
			// measured with every task in one class (short / long ring, ms per 3.28e8 anchors): ava-ont mixed (3.3 tiles per anchor by this
			// count) 118 / 96, dense (0.85) 83.6 / 82.3, asm20 mixed (0.5) 99.7 / 110.6, headline (0.05) 50.1 / 61.5: the long ring, at half the
			// occupancy, pays only where scans go far beyond the short one
			// round 3 (faster ring path): dense 80.4 / 76.0, asm20 mixed 93.2 / 102.3, ava-ont mixed 113.5 / 91.6, headline 45.5 / 56.2 -> the bar sits between
			// asm20 mixed (0.5) and dense (0.85)
			if (lane == 0) {
				const int c = (!huge && n >= 1024 && 10 * s_far > (unsigned long long)far_thr10 * (unsigned long long)n) ? 1 : 0;
				cls_out[task] = (uint8_t)(c | wide | huge);
				if (cls_stat) {
					unsigned long long *cs = cls_stat + 4 * (task & (CLS_STAT_SLOTS - 1));   // 64 sets of counters: 65 536 tasks adding to ONE set cost the dense stream's prepass 0.6 ms
					atomicAdd(&cs[1], (unsigned long long)n); if (c) atomicAdd(&cs[0], (unsigned long long)n);
					if (wide) atomicAdd(&cs[3], (unsigned long long)n);
				}
			}
		} else if (lane == 0) {
			cls_out[task] = (uint8_t)((far_ring == 2 && !huge ? 1 : 0) | wide | huge);
			if (cls_stat) { unsigned long long *cs = cls_stat + 4 * (task & (CLS_STAT_SLOTS - 1)); atomicAdd(&cs[1], (unsigned long long)n); if (wide) atomicAdd(&cs[3], (unsigned long long)n); }
		}
	}
	if (avg_out && n > 0) {
		// avg_qspan_scaled of the task (chain.c:48-49), so that the DP kernel does not sweep the anchors a second time
		for (int o = 32; o > 0; o >>= 1) span_sum += __shfl_xor(span_sum, o);
		if ((lane & 63) == 0) atomicAdd(&s_sum, (unsigned long long)span_sum);
		__syncthreads();
		if (lane == 0) avg_out[task] = (float)(__dmul_rn(.01, (double)(float)s_sum) / (double)n);
	}
}

// The same window starts for a pass of few tasks (a lone call: one block above would walk the task's tiles one after the other, 25 us for 5 000 anchors, while the
// GPU is empty): one block per 256 anchors of a task, blockIdx.y = the tile, every lane a binary search over the task's sorted x in memory -- the tiles do not wait for
// one another, the answers are the same by construction (the same bounds [max(i - max_iter, 0), i], the same condition).  Only st[]: no classes, no avg, no cut flags.
// span_sum (optional, one zeroed word per task): the block adds the spans of its anchors (chain.c:48: a[i].y >> 32 & 0xff) -- chain_avg_finish turns the sums into
// avg_qspan_scaled in place.  (A task of up to 2^22 anchors: the sum fits 32 bits.)
__global__ void __launch_bounds__(256)
chain_window_start_wide(KParams P, int64_t n_tasks, const int64_t *__restrict__ offsets, const ulonglong2 *__restrict__ a_all, int32_t *__restrict__ st_all,
                        unsigned *__restrict__ span_sum)
{
	const int64_t task = (int64_t)blockIdx.x;
	if (task >= n_tasks) return;
	const int64_t base = offsets[task];
	const int n = (int)(offsets[task + 1] - base);
	if ((int)blockIdx.y * 256 >= n) return;
	const int i = (int)blockIdx.y * 256 + (int)threadIdx.x;
	const ulonglong2 *a = a_all + base;
	if (span_sum) {
		unsigned v = i < n ? (unsigned)(a[i].y >> 32) & 0xffu : 0u;
		for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
		if ((threadIdx.x & 63) == 0 && v) atomicAdd(&span_sum[task], v);
	}
	if (i >= n) return;
	const uint64_t D = (uint64_t)(int64_t)P.max_dist_x;
	const uint64_t xi = a[i].x;
	int hi = i, lo = max(i - P.max_iter, 0);                          // answer in [lo, hi]; x_i <= x_i + D always holds
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (xi > a[mid].x + D) lo = mid + 1; else hi = mid;           // chain.c:192 condition for "++st"
	}
	st_all[base + i] = lo;
}

// Which kernel takes the pieces of a batch, decided where the pieces are known: chain_cut has just counted them.  Few long pieces (coop_pays, chain_kernel.h) go to the
// cooperative kernel, anything else to one wave per piece; both kernels are launched, each goes by its own count word and the one that was not chosen finds 0 there
// (the cooperative kernel in two widths, see launch_coop).
__global__ void __launch_bounds__(256)
chain_route(CutArgs C)
{
	__shared__ unsigned long long s_tot;
	__shared__ int s_max;
	const int n = *C.d_count;
	if (n > COOP_ROUTE_MAX_PIECES) { if (threadIdx.x == 0) { C.d_count[1] = n; C.d_count[2] = 0; C.d_count[3] = 0; } return; }
	if (threadIdx.x == 0) { s_tot = 0; s_max = 0; }
	__syncthreads();
	unsigned long long tot = 0; int mx = 0;
	for (int p = (int)threadIdx.x; p < n; p += 256) { const int len = (int)(C.d_end[p] - C.d_start[p]); tot += (unsigned long long)len; mx = max(mx, len); }
	atomicAdd(&s_tot, tot); atomicMax(&s_max, mx);
	__syncthreads();
	if (threadIdx.x == 0) {
		const bool coop = coop_pays(n, s_max, (long long)s_tot, C.w8_above);
		const bool eight = n > C.w8_above;                                  // more pieces than CUs: two workgroups of eight waves per CU
		C.d_count[1] = coop ? 0 : n; C.d_count[2] = coop && !eight ? n : 0; C.d_count[3] = coop && eight ? n : 0;
	}
}

// avg_qspan_scaled (chain.c:48-49: .01 * (float)sum / n, the product and the quotient in double, rounded to float) from the span sums of chain_window_start_wide, in place
__global__ void __launch_bounds__(256)
chain_avg_finish(int64_t n_tasks, const int64_t *__restrict__ offsets, unsigned *__restrict__ sum_avg)
{
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (t >= n_tasks) return;
	const int64_t n = offsets[t + 1] - offsets[t];
	const unsigned sum = sum_avg[t];
	((float *)sum_avg)[t] = n > 0 ? (float)(__dmul_rn(.01, (double)(float)sum) / (double)n) : 0.f;
}

// ---------------------------------------------------------------- ring-size classes: one class for a batch that is nearly of one kind
// The two classes are two launches that run one after the other, and each ends with the GPU part empty while its last tasks finish.  A handful
// of long tasks in a launch of their own costs the batch the whole length of one of them (ragged mixed stream: 53.1 -> 59.6 ms for a few reads
// of 8 800 anchors), and a stream whose tasks sit around the bar is cut in two halves (dense stream at a bar of 0.9: 89 ms against 76 / 80 for
// one class).  So: when the class-1 tasks hold less than a quarter of the batch's anchors every task runs in class 0, when they hold more than
// three quarters every task runs in class 1 (any task is correct in either); in between the split stands.
// The same goes for the split between the compact and the 32-bit x / q ring (they run side by side on two streams where the caller has a second one, which
// takes the edge off: ragged mixed stream, 18 % of the anchors in tasks with the 32-bit ring, 54.6 -> 48.0 ms; but the longest tasks are the ones with the
// 32-bit ring, and with neighbours on their CUs they last longer): when the tasks that need the 32-bit ring hold more than wide_pct % of the anchors, every
// task takes it (ragged dense stream, 65 %: 86.2 ms split, 83.8 not; ragged asm20, 60 %: 108.2 / 106.6).
__global__ void __launch_bounds__(256)
chain_cls_settle(int64_t n_tasks, uint8_t *__restrict__ cls, const unsigned long long *__restrict__ cls_stat, int settle_ring, int wide_pct)
{
	__shared__ unsigned long long s_tot[4];
	if (threadIdx.x < 4) s_tot[threadIdx.x] = 0;
	__syncthreads();
	if (threadIdx.x < 4 * CLS_STAT_SLOTS) atomicAdd(&s_tot[threadIdx.x & 3], cls_stat[threadIdx.x]);
	__syncthreads();
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (t >= n_tasks) return;
	const bool all_wide = 100 * s_tot[3] > (unsigned long long)wide_pct * s_tot[1];
	const unsigned long long far = s_tot[0], all = s_tot[1];   // (over every task: counted among the few tasks with the 32-bit ring alone the split often stands, 47.6 -> 54.1 ms on the ragged mixed stream)
	int c = cls[t];
	if (all_wide) c |= 2;
	if (settle_ring) {
		if (4 * far < all) c &= 6;                            // (bit 1, the 32-bit ring, and bit 2, a q value beyond the q24 ring, are the task's own)
		else if (4 * far > 3 * all && !(c & 4)) c |= 1;       // (a task the q24 long ring cannot hold stays in class 0, whatever the batch does)
	}
	cls[t] = (uint8_t)c;
}

// ---------------------------------------------------------------- the reference's HW/SW prediction pass, chain.c:53-78
// Per anchor: inner-loop trip count min(i - st, 1024) with the UNclamped st of chain.c:64 (no max_iter here), the
// number of 128-wide sub-parts the FPGA pipeline would spend on it (chain.c:74-76, chain_hardware.h:58-60); per task
// their sums (total_trip_count chain.c:69, total_subparts chain.c:77), which feed the two linear time models
// (chain.c:80-81).  One wave per task.
__global__ void __launch_bounds__(64)
chain_predict(int32_t max_dist_x, int64_t n_tasks, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
              const ulonglong2 *__restrict__ a_all, uint8_t *__restrict__ nsub_all, int64_t *__restrict__ total_sub,
              int64_t *__restrict__ total_trip)
{
	const int lane = threadIdx.x;
	const int64_t task = order ? (int64_t)order[blockIdx.x] : (int64_t)blockIdx.x;
	if (task >= n_tasks) return;
	const int64_t base = offsets[task];
	const int n = (int)(offsets[task + 1] - base);
	const ulonglong2 *a = a_all + base;
	const uint64_t D = (uint64_t)(int64_t)max_dist_x;
	int64_t s_sub = 0, s_trip = 0;
	for (int i = lane; i < n; i += 64) {
		const uint64_t xi = a[i].x;
		int lo = 0, hi = i;
		while (lo < hi) {
			const int mid = (lo + hi) >> 1;
			if (xi > a[mid].x + D) lo = mid + 1; else hi = mid;
		}
		const int trip = min(i - lo, 1024);
		const int sub = trip / 128 + ((trip == 0 || trip % 128 > 0) ? 1 : 0);
		if (nsub_all) nsub_all[base + i] = (uint8_t)sub;
		s_sub += sub; s_trip += trip;
	}
	for (int o = 32; o > 0; o >>= 1) { s_sub += __shfl_xor(s_sub, o); s_trip += __shfl_xor(s_trip, o); }
	if (lane == 0) { if (total_sub) total_sub[task] = s_sub; if (total_trip) total_trip[task] = s_trip; }
}

// ---------------------------------------------------------------- pieces of a task, cut on the device
// Where st[i] == i (empty window, chain.c:192: x_i > x_{i-1} + max_dist_x) no anchor at or after i can chain to, stamp or be stamped by an
// anchor before i, so [.., i) and [i, ..) are independent for f[] / p[].  A long read made of many loci becomes many
// short pieces = many waves instead of one long dependent chain.  Pieces shorter than seg_min are merged with their successor.  One wave
// per task: count the pieces, reserve a contiguous range of piece slots, write them in order.  avg_qspan_scaled is a whole-task quantity
// (chain.c:48-49): computed here (or taken from the caller) and copied to every piece.
__global__ void __launch_bounds__(64)
chain_cut(int seg_min, int64_t n_tasks, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order, const uint4 *__restrict__ a_all,
          const float *__restrict__ avg_in, const int32_t *__restrict__ st_all, CutArgs C, const uint8_t *__restrict__ cls)
{
	const int lane = threadIdx.x;
	const int64_t task = order ? (int64_t)order[blockIdx.x] : (int64_t)blockIdx.x;
	if (task >= n_tasks) return;
	const int64_t base = offsets[task];
	const int n = (int)(offsets[task + 1] - base);
	if (n <= 0) return;
	const int32_t *st = st_all + base;
	if (n < C.min_anchors || C.d_has_cut[task] == 0) {
		// a short task, or no empty window inside it (the prepass saw none): one piece; its avg is left to the DP kernel (negative = "not computed")
		if (lane == 0) {
			const int k = atomicAdd(C.d_count, 1);
			C.d_start[k] = base; C.d_end[k] = base + n; C.d_pbase[k] = 0; C.d_avg[k] = avg_in ? avg_in[task] : -1.0f; if (C.d_cls) C.d_cls[k] = cls ? cls[task] : 0;
		}
		return;
	}
	float avg;
	if (avg_in) avg = avg_in[task];
	else {
		uint64_t sum = 0;
		for (int k = lane; k < n; k += 64) sum += (a_all[base + k].w & 0xffu);
		for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
		avg = (float)(__dmul_rn(.01, (double)(float)sum) / (double)n);
	}
	int slot0 = 0;
	for (int pass = 0; pass < 2; ++pass) {                      // pass 0 counts, pass 1 writes
		int s0 = 0, cnt = 0;
		for (int i0 = 0; i0 < n; i0 += 64) {
			const int i = i0 + lane;
			uint64_t m = __ballot(i > 0 && i < n && st[i] == i);
			for (;;) {                                            // the first candidate that leaves >= seg_min anchors behind, then again from there
				const int from = s0 + seg_min - i0;
				if (from >= 64) break;
				if (from > 0) m &= ~0ull << from;
				if (m == 0) break;
				const int c = i0 + (int)__builtin_ctzll(m);
				if (pass == 1 && lane == 0) {
					const int k = slot0 + cnt;
					C.d_start[k] = base + s0; C.d_end[k] = base + c; C.d_pbase[k] = s0; C.d_avg[k] = avg; if (C.d_cls) C.d_cls[k] = cls ? cls[task] : 0;
				}
				++cnt; s0 = c;
				m &= m - 1;
			}
		}
		if (pass == 1 && lane == 0) {
			const int k = slot0 + cnt;
			C.d_start[k] = base + s0; C.d_end[k] = base + n; C.d_pbase[k] = s0; C.d_avg[k] = avg; if (C.d_cls) C.d_cls[k] = cls ? cls[task] : 0;
		}
		++cnt;                                                    // the last piece
		if (pass == 0) {
			if (lane == 0) slot0 = atomicAdd(C.d_count, cnt);
			slot0 = __shfl(slot0, 0);
		}
	}
}

// ---------------------------------------------------------------- the kernel: one wave per task
template <int R, bool SKIP, bool GEN, bool GS1, bool FAR>
__global__ void __launch_bounds__(64)
chain_dp_wave(KParams P, int64_t n_tasks, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
              const uint4 *__restrict__ a_all, const float *__restrict__ avg_in, const int32_t *__restrict__ pbase_in,
              const int32_t *__restrict__ st_all, int32_t *__restrict__ f_all, int32_t *__restrict__ p_all, int32_t *__restrict__ t_all,
              int32_t *__restrict__ status, int only_flagged, const int64_t *__restrict__ ends, const int32_t *__restrict__ n_live)
{
	static_assert(R >= 128 && (R & (R - 1)) == 0, "ring must be a power of two >= 128");
	__shared__ uint2 s_xq[R];        // x low word, query position
	__shared__ int2 s_fp[R];         // f, p
	__shared__ uint16_t s_t[2 * R];     // 16-bit stamps (chain.c t[]), ring of 2R anchors
	__shared__ uint8_t s_g[GEN ? R : 64];

	const int lane = threadIdx.x;
	const int64_t task = order ? (int64_t)__builtin_amdgcn_readfirstlane(order[blockIdx.x]) : (int64_t)blockIdx.x;
	if (task >= n_tasks) return;
	if (n_live && task >= (int64_t)*n_live) return;           // pieces cut on the device (chain_cut): the grid is sized for the worst case
	if (only_flagged && status[task] == 0) return;
	const int64_t base = offsets[task];
	const int n = __builtin_amdgcn_readfirstlane((int)((ends ? ends[task] : offsets[task + 1]) - base));   // wave-uniform: loop bounds stay on the scalar unit
	if (n <= 0) return;
	const uint4 *a = a_all + base;         // {x lo, x hi, y lo (= query pos), y hi (span | flags | seg)}
	const int32_t *st = st_all + base;
	int32_t *f = f_all + base, *p = p_all + base, *t = FAR ? t_all + base : nullptr;

	for (int s = lane; s < 2 * R; s += 64) s_t[s] = 0;
	const int pbase = pbase_in ? pbase_in[task] : 0;
	const int st_sub = ends ? pbase : 0;                      // device-cut pieces: st[] was computed for the whole task (task-relative)

	// avg_qspan_scaled, chain.c:48-49
	float avg = avg_in ? avg_in[task] : -1.0f;
	if (avg < 0.f) {                                          // not handed in (chain_cut leaves uncut tasks to this kernel)
		uint64_t sum = 0;
		for (int k = lane; k < n; k += 64) sum += (a[k].w & 0xffu);
		for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
		avg = (float)(__dmul_rn(.01, (double)(float)sum) / (double)n);
	}

	int wx = 0, wq = 0, wf = 0, wp = -1, wg = 0;              // chunk-0 window: lane L = anchor i-1-L
	Victim vt;                                                // tiles that left the ring last (FAR variants only)
#pragma unroll
	for (int k = 0; k < NV; ++k) { vt.x[k] = vt.q[k] = vt.f[k] = vt.g[k] = 0; vt.p[k] = -1; }
	vt.base = INT_MIN / 2;
	int seg0 = 0;
	bool t_ready = false;                                     // t[0 .. i0) has been zeroed (wave-uniform)
	char *const t_bytes = (char *)s_t;
	const char *const xq_bytes = (const char *)s_xq, *const fp_bytes = (const char *)s_fp;

	uint4 cur = (lane < n) ? a[lane] : make_uint4(0, 0, 0, 0);
	int cur_st = (lane < n) ? st[lane] - st_sub : 0;
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int idx = i0 + lane;
		const int cnt = __builtin_amdgcn_readfirstlane(min(64, n - i0));   // keep the anchor loop bound on the scalar unit
		uint4 nxt = make_uint4(0, 0, 0, 0); int nxt_st = 0;
		if (idx + 64 < n) { nxt = a[idx + 64]; nxt_st = st[idx + 64] - st_sub; }   // prefetch the next tile
		const int g_l = (cur.w >> 16) & 0xff;                                 // MM_SEED_SEG_MASK mmpriv.h:22-23
		if (!GEN && !(P.flags & KF_IGNORE_SEG)) {
			// the simple variant assumes one segment id per task; anything else is redone by the general one
			if (i0 == 0) seg0 = rdlane(g_l, 0);
			if (BALLOT(lane < cnt && g_l != seg0)) { if (lane == 0) status[task] = 1; return; }
		}
		// while this tile is processed the ring holds anchors [i0-R, i0): the tile itself lives in `cur` and in the
		// chunk-0 window and enters the ring when it is finished (older chunks never reach into the current tile)
		s_t[idx & (2 * R - 1)] = 0;          // stamp slots of the entering anchors (recycled from idx-2R)
		// global stamp scratch t[]: zeroed lazily, only once this task's windows can reach beyond the LDS stamp ring
		if (FAR) {
			const int reach = rdlane(cur_st, 0);                  // window start of the first anchor of the tile (st[] is monotone)
			if (!t_ready && reach < i0 + 64 - 2 * R) {
				for (int z = lane; z < i0; z += 64) t[z] = 0;
				t_ready = true;
			}
			if (t_ready && idx < n) t[idx] = 0;
		}
		const int span_l = P.span_override >= 0 ? P.span_override : (int)(cur.w & 0xff);   // chain.c:189
		const int lds_lo = i0 - R;            // oldest anchor index still in the ring while this tile is processed

		for (int k = 0; k < cnt; ++k) {
			const int i = i0 + k;
			const int xi = rdlane((int)cur.x, k), qi = rdlane((int)cur.z, k);
			const int span_i = rdlane(span_l, k);
			const int seg_i = GEN ? rdlane(g_l, k) : 0;                                         // chain.c:191
			const int lo = rdlane(cur_st, k);                                                   // chain.c:192-193
			Carry c = { span_i, -1, 0 };                                                         // chain.c:188-190
			if (i - lo > 0) {
				Win w = { wx, wq, wf, wp, wg };
				// the common case (window entirely inside the LDS ring) runs a loop with no look-back-beyond-the-ring tests
				if (!FAR || lo >= lds_lo)
					scan_window<R, SKIP, GEN, GS1, false>(P, avg, lane, i, lo, lds_lo, xi, qi, span_i, seg_i, w, vt, xq_bytes, fp_bytes, s_g,
					                                      t_bytes, s_t, a, f, p, t, pbase, c);
				else
					scan_window<R, SKIP, GEN, GS1, FAR>(P, avg, lane, i, lo, lds_lo, xi, qi, span_i, seg_i, w, vt, xq_bytes, fp_bytes, s_g,
					                                    t_bytes, s_t, a, f, p, t, pbase, c);
			}
			// ---- commit anchor i (chain.c:236) into the chunk-0 window
			wx = window_push(wx, xi);
			wq = window_push(wq, qi);
			wf = window_push(wf, c.best);
			wp = window_push(wp, c.best_j);
			if (GEN) wg = window_push(wg, seg_i);
		}
		// after the tile, window lane L holds anchor i0+cnt-1-L: f/p of the tile enter the ring (older chunks never
		// reach into the current tile, so once per tile is enough) and leave in coalesced 256 B stores
		if (FAR) {
			// the tile that the finished one pushes out of the ring stays reachable in registers for NV more tiles
			const int o = i0 + 63 - lane;                         // its slots are the ones written just below
			const uint2 oxq = s_xq[o & (R - 1)]; const int2 ofp = s_fp[o & (R - 1)];
#pragma unroll
			for (int k = NV - 1; k > 0; --k) { vt.x[k] = vt.x[k - 1]; vt.q[k] = vt.q[k - 1]; vt.f[k] = vt.f[k - 1]; vt.p[k] = vt.p[k - 1]; if (GEN) vt.g[k] = vt.g[k - 1]; }
			vt.x[0] = (int)oxq.x; vt.q[0] = (int)oxq.y; vt.f[0] = ofp.x; vt.p[0] = ofp.y; vt.base = i0 - R;
			if (GEN) vt.g[0] = s_g[o & (R - 1)];
		}
		if (lane < cnt) {
			const int o = i0 + cnt - 1 - lane;
			s_xq[o & (R - 1)] = make_uint2((uint32_t)wx, (uint32_t)wq);
			s_fp[o & (R - 1)] = make_int2(wf, wp);
			if (GEN) s_g[o & (R - 1)] = (uint8_t)wg;
			f[o] = wf; p[o] = wp < 0 ? wp : wp + pbase;
		}
		cur = nxt; cur_st = nxt_st;
	}
}

// ---------------------------------------------------------------- host-side launcher
template <int R, bool SKIP, bool GEN, bool GS1, bool FAR>
static hipError_t launch_one(const LaunchArgs &L, hipStream_t st, int only_flagged)
{
	if (L.cut.max_pieces > 0) {
		// pieces cut on the device: starts / ends / p base / avg per piece, st[] relative to the task
		hipLaunchKernelGGL((chain_dp_wave<R, SKIP, GEN, GS1, FAR>), dim3((unsigned)L.cut.max_pieces), dim3(64), 0, st,
		                   L.P, L.cut.max_pieces, L.cut.d_start, (const int32_t *)nullptr, (const uint4 *)L.d_anchors, L.cut.d_avg, L.cut.d_pbase, L.d_st, L.d_f, L.d_p,
		                   L.d_t, L.cut.d_status, only_flagged, L.cut.d_end, L.cut.d_count);
		return hipGetLastError();
	}
	hipLaunchKernelGGL((chain_dp_wave<R, SKIP, GEN, GS1, FAR>), dim3((unsigned)L.n_tasks), dim3(64), 0, st,
	                   L.P, L.n_tasks, L.d_offsets, L.d_order, (const uint4 *)L.d_anchors, L.d_avg, L.d_pbase, L.d_st, L.d_f, L.d_p, L.d_t,
	                   L.d_status, only_flagged, (const int64_t *)nullptr, (const int32_t *)nullptr);
	return hipGetLastError();
}

// ---- second-generation kernel (chain_dp_tile.h): x / q rings of NX tiles, f / p rings of NF tiles
// with_cls: the kernel takes the tasks whose class (prepass) masked with cls_mask equals my_cls
template <int NX, int NF, bool SKIP, bool GEN, bool GS1, bool FAR, bool TAB, int C16>
static hipError_t launch_tile_nx(const LaunchArgs &L, const float *d_avg, hipStream_t st, int only_flagged, bool with_cls, int my_cls, int cls_mask)
{
	if (L.cut.max_pieces > 0) {
		hipLaunchKernelGGL((chain_dp_tile<NX, NF, SKIP, GEN, GS1, FAR, TAB, C16>), dim3((unsigned)L.cut.max_pieces), dim3(64), 0, st,
		                   L.P, L.cut.max_pieces, L.cut.d_start, (const int32_t *)nullptr, (const uint4 *)L.d_anchors, L.cut.d_avg, L.cut.d_pbase, L.d_st, L.d_f, L.d_p,
		                   L.d_t, L.cut.d_status, only_flagged, L.cut.d_end, L.cut.d_live ? L.cut.d_live : L.cut.d_count, with_cls ? (const uint8_t *)L.cut.d_cls : (const uint8_t *)nullptr, my_cls, cls_mask);
		return hipGetLastError();
	}
	hipLaunchKernelGGL((chain_dp_tile<NX, NF, SKIP, GEN, GS1, FAR, TAB, C16>), dim3((unsigned)L.n_tasks), dim3(64), 0, st,
	                   L.P, L.n_tasks, L.d_offsets, L.d_order, (const uint4 *)L.d_anchors, d_avg, L.d_pbase, L.d_st, L.d_f, L.d_p, L.d_t,
	                   L.d_status, only_flagged, (const int64_t *)nullptr, (const int32_t *)nullptr, with_cls ? (const uint8_t *)L.d_cls : (const uint8_t *)nullptr, my_cls, cls_mask);
	return hipGetLastError();
}

// the class array of the prepass reaches the kernels (plans have it; the piece arrays of a device-side cut must carry it too)
static bool have_cls(const LaunchArgs &L) { return L.d_cls != nullptr && (L.cut.max_pieces == 0 || L.cut.d_cls != nullptr); }
// ring-size classes: class-1 tasks run the instantiation with the long ring (variants with the hand-written loop only)
static bool use_classes(const LaunchArgs &L, bool skip, bool gen) { return !gen && skip && L.far_ring != 0 && have_cls(L); }
// the compact x / q ring (chain_dp_tile.h, Lds<>): differences of the low halves are exact when max_dist_x < 2^16 and, per task, the q values span
// at most 65535 - max_dq; returns that bound for the prepass (0: not used)
static unsigned compact_q_span(const LaunchArgs &L, bool asm_loop)
{
	const KParams &P = L.P;
	if (!L.compact || !asm_loop || !have_cls(L) || P.max_dist_x < 0 || P.max_dist_x > 65535 || P.max_dq < 1 || P.max_dq > 32768) return 0;
	return 65535u - (unsigned)P.max_dq;
}

// the q24 ring (chain_dp_tile.h, Lds<> RING 2) is the form of the LONG ring (ring-size class 1): dr from the low halves of x needs max_dist_x < 2^16, q is exact for
// tasks whose q values are below 2^24 -- the prepass keeps every other task out of class 1 (bit 2 of the class byte)
static bool q24_ring(const LaunchArgs &L, bool skip, bool gen, bool gs1, bool tab)
{
	return L.q24 != 0 && use_classes(L, skip, gen) && (gs1 || tab) && L.P.max_dist_x >= 0 && L.P.max_dist_x <= 65535;
}

// The 32-bit rings: one ring size for every task, or -- ring-size classes, variants with the hand-written loop only -- the short ring for class 0 and a ring of
// twice the length for class 1.  Where the compact x / q ring applies, the tasks whose q values allow it (bit 1 of the class clear) take the one instantiation
// with it instead (a ring of MM2C_CNX tiles, whatever their ring-size class).  Every instantiation is launched over all tasks and returns at once for the tasks
// of another one (0.02 ms per launch); the 32-bit ones go first: they hold the longest tasks (a wide span of q values comes with a long read).
template <bool SKIP, bool GEN, bool GS1, bool FAR, bool TAB>
static hipError_t launch_tile_one(const LaunchArgs &L, const float *d_avg, hipStream_t st, int only_flagged, int *n_launches)
{
	const bool classes = use_classes(L, SKIP, GEN);
	bool c16 = false;
	if constexpr (SKIP && !GEN && (GS1 || TAB)) c16 = compact_q_span(L, L.P.bw >= 0 && L.P.max_dq - 1 >= L.P.bw) != 0;
	const int wide = c16 ? 2 : 0, mask = (c16 ? 2 : 0) | (classes ? 1 : 0);
	// the 32-bit instantiations beside the compact one: on the side stream, between a fork and a join event
	const bool fork = c16 && L.side != nullptr && L.ev_fork != nullptr && L.ev_join != nullptr;
	hipStream_t sw = fork ? L.side : st;
	hipError_t e = hipSuccess;
	bool forked = false;                                   // the side stream has been made to wait for `st`: it is joined again whatever fails in between
	if (fork) { e = hipEventRecord(L.ev_fork, st); if (e == hipSuccess) { e = hipStreamWaitEvent(sw, L.ev_fork, 0); forked = e == hipSuccess; } }
	if (e == hipSuccess) e = launch_tile_nx<MM2C_NX, MM2C_NF, SKIP, GEN, GS1, FAR, TAB, 0>(L, d_avg, sw, only_flagged, mask != 0, wide, mask);
	if constexpr (!GEN && SKIP)
		if (e == hipSuccess && classes) {
			bool done = false;
			if constexpr (GS1 || TAB)
				if (q24_ring(L, SKIP, GEN, GS1, TAB)) { e = launch_tile_nx<2 * MM2C_NX, MM2C_NF1, SKIP, GEN, GS1, FAR, TAB, 2>(L, d_avg, sw, only_flagged, true, wide | 1, mask); done = true; }
			if (!done) e = launch_tile_nx<2 * MM2C_NX, MM2C_NF1, SKIP, GEN, GS1, FAR, TAB, 0>(L, d_avg, sw, only_flagged, true, wide | 1, mask);
			if (n_launches) ++*n_launches;
		}
	bool joined = false;
	if (forked) joined = hipEventRecord(L.ev_join, sw) == hipSuccess;
	if constexpr (SKIP && !GEN && (GS1 || TAB))
		if (e == hipSuccess && c16) { e = launch_tile_nx<MM2C_CNX, MM2C_CNF, SKIP, GEN, GS1, FAR, TAB, 1>(L, d_avg, st, only_flagged, true, 0, 2); if (n_launches) ++*n_launches; }
	if (forked) {
		const hipError_t ej = joined ? hipStreamWaitEvent(st, L.ev_join, 0) : hipStreamSynchronize(sw);   // no event to wait on: the host waits for the side stream instead
		if (e == hipSuccess) e = joined ? ej : hipErrorUnknown;
	}
	return e;
}

// ---- several waves per task (chain_dp_coop.h): passes too small to fill the GPU with one wave per task; the variants of the hand-written loop only
// Waves per piece: sixteen (one workgroup per CU: the pieces of a pass of at most one per CU get a CU each) or eight (two workgroups per CU: with more pieces than CUs a CU
// that holds two fills the waits of one -- its barriers, 52 % of the wave cycles at sixteen -- with the rows of the other: 1 020 reads of 300 000 anchors 129.3 -> 111.3 ms,
// 2 048 of 100 000 96.1 -> 77.7, while 255 of 10^6 take 153.9 instead of 101.5: profiles/r6_long_reads.md).  `w8_above`: pieces beyond which eight are taken.
template <int W>
static hipError_t launch_coop_w(const LaunchArgs &L, const float *d_avg, hipStream_t st, bool tab, int only_flagged, const int32_t *n_live, int32_t *st_out = nullptr, float *avg_out = nullptr, const uint4 *a_src = nullptr)
{
	const bool far_ = (int64_t)L.P.max_iter > 64 * (COOP_NX - 1);
	const dim3 block(64 * W);
	if (L.cut.max_pieces > 0) {
		// the pieces chain_route gave to this kernel (*n_live of them: at most COOP_ROUTE_MAX_PIECES, or none)
		const dim3 grid((unsigned)std::min<int64_t>(L.cut.max_pieces, COOP_ROUTE_MAX_PIECES));
#define MM2C_COOP(GS1, FAR, TAB) hipLaunchKernelGGL((chain_dp_coop<W, GS1, FAR, TAB>), grid, block, 0, st, L.P, L.cut.max_pieces, L.cut.d_start, (const int32_t *)nullptr, \
	                                            (const uint4 *)L.d_anchors, (const float *)L.cut.d_avg, L.cut.d_pbase, L.d_st, L.d_f, L.d_p, L.d_t, L.cut.d_status, only_flagged, \
	                                            (const int64_t *)L.cut.d_end, n_live, CoopHostOut(), (int32_t *)nullptr, (float *)nullptr, (const uint4 *)nullptr, CoopMeta())
		if (tab) { if (far_) MM2C_COOP(true, true, true); else MM2C_COOP(true, false, true); }
		else { if (far_) MM2C_COOP(true, true, false); else MM2C_COOP(true, false, false); }
#undef MM2C_COOP
		return hipGetLastError();
	}
	const dim3 grid((unsigned)L.n_tasks);
	CoopHostOut H;
	if (L.h_flag && L.h_f && L.h_p && L.d_done && (L.P.flags & KF_IGNORE_SEG) && !only_flagged) { H.f = L.h_f; H.p = L.h_p; H.d_done = L.d_done; H.h_flag = L.h_flag; H.seq = L.seq; }
	CoopMeta MT;
	if (a_src && H.h_flag && L.hm_off && L.hm_avg && L.hm_pbase && L.n_tasks <= COOP_META_MAX && !L.d_order) {
		MT.n = (int32_t)L.n_tasks;
		for (int64_t k = 0; k < L.n_tasks; ++k) { MT.off[k] = L.hm_off[k]; MT.avg[k] = L.hm_avg[k]; MT.pbase[k] = L.hm_pbase[k]; }
		MT.off[L.n_tasks] = L.hm_off[L.n_tasks];
	}
#define MM2C_COOP(GS1, FAR, TAB) hipLaunchKernelGGL((chain_dp_coop<W, GS1, FAR, TAB>), grid, block, 0, st, L.P, L.n_tasks, L.d_offsets, L.d_order, (const uint4 *)L.d_anchors, \
	                                            d_avg, L.d_pbase, L.d_st, L.d_f, L.d_p, L.d_t, L.d_status, only_flagged, (const int64_t *)nullptr, (const int32_t *)nullptr, H, st_out, avg_out, H.h_flag ? a_src : (const uint4 *)nullptr, MT)
	if (tab) { if (far_) MM2C_COOP(true, true, true); else MM2C_COOP(true, false, true); }
	else { if (far_) MM2C_COOP(true, true, false); else MM2C_COOP(true, false, false); }
#undef MM2C_COOP
	return hipGetLastError();
}

static int coop_width(const LaunchArgs &L) { return L.n_tasks > (int64_t)L.coop_w8_above ? 8 : 16; }   // (pieces = tasks: no cut on the device)

// a pass of few SHORT tasks whose window starts nobody has made: the sixteen-wave kernel makes them itself (chain_dp_coop.h, st_out) -- no prepass launch
static bool coop_makes_st(const LaunchArgs &L)
{
	return L.fuse_st && !L.st_ready && L.cut.max_pieces == 0 && coop_width(L) == 16 && L.max_task_anchors > 0 && L.max_task_anchors <= COOP_ST_MAX && (L.d_avg != nullptr || L.d_avg_ws != nullptr);
}

static hipError_t launch_coop(const LaunchArgs &L, const float *d_avg, hipStream_t st, bool tab, int only_flagged, int *n_launches)
{
	if (L.cut.max_pieces > 0) {
		// cut on the device: chain_route put the count under the width it chose (d_count[2]: sixteen waves, [3]: eight); the launch that was not chosen finds 0
		hipError_t e = launch_coop_w<16>(L, d_avg, st, tab, only_flagged, L.cut.d_count + 2);
		if (n_launches) ++*n_launches;
		if (e == hipSuccess && L.cut.max_pieces > (int64_t)L.coop_w8_above) { e = launch_coop_w<8>(L, d_avg, st, tab, only_flagged, L.cut.d_count + 3); if (n_launches) ++*n_launches; }
		return e;
	}
	if (n_launches) ++*n_launches;
	if (coop_width(L) == 8) return launch_coop_w<8>(L, d_avg, st, tab, only_flagged, nullptr);
	const bool makes = coop_makes_st(L) && !only_flagged;
	// (no avg handed in: the kernel sweeps the task itself and leaves the value in the workspace)
	return launch_coop_w<16>(L, makes && !L.d_avg ? (const float *)nullptr : d_avg, st, tab, only_flagged, nullptr, makes ? L.d_st : (int32_t *)nullptr, makes && !L.d_avg ? L.d_avg_ws : (float *)nullptr,
	                         makes && L.d_avg ? (const uint4 *)L.h_anchors : (const uint4 *)nullptr);
}

template <bool SKIP, bool FAR>
static hipError_t launch_tile_sf(const LaunchArgs &L, const float *d_avg, hipStream_t st, bool gen, bool gs1, bool tab, int only_flagged, int *nl)
{
	if (tab && !gen) return launch_tile_one<SKIP, false, true, FAR, true>(L, d_avg, st, only_flagged, nl);   // the table absorbs gap_scale
	if (gen) return gs1 ? launch_tile_one<SKIP, true, true, FAR, false>(L, d_avg, st, only_flagged, nl) : launch_tile_one<SKIP, true, false, FAR, false>(L, d_avg, st, only_flagged, nl);
	return gs1 ? launch_tile_one<SKIP, false, true, FAR, false>(L, d_avg, st, only_flagged, nl) : launch_tile_one<SKIP, false, false, FAR, false>(L, d_avg, st, only_flagged, nl);
}

static hipError_t launch_tile(const LaunchArgs &L, const float *d_avg, hipStream_t st, bool skip, bool gen, bool gs1, bool far_, bool tab, int only_flagged, int *nl)
{
	if (skip) return far_ ? launch_tile_sf<true, true>(L, d_avg, st, gen, gs1, tab, only_flagged, nl) : launch_tile_sf<true, false>(L, d_avg, st, gen, gs1, tab, only_flagged, nl);
	return far_ ? launch_tile_sf<false, true>(L, d_avg, st, gen, gs1, tab, only_flagged, nl) : launch_tile_sf<false, false>(L, d_avg, st, gen, gs1, tab, only_flagged, nl);
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

hipError_t launch_predict(int32_t max_dist_x, int64_t n_tasks, const int64_t *d_offsets, const int32_t *d_order, const void *d_anchors,
                          uint8_t *d_num_subparts, int64_t *d_total_subparts, int64_t *d_total_trip, hipStream_t st)
{
	if (n_tasks <= 0) return hipSuccess;
	hipLaunchKernelGGL(chain_predict, dim3((unsigned)n_tasks), dim3(64), 0, st, max_dist_x, n_tasks, d_offsets, d_order,
	                   (const ulonglong2 *)d_anchors, d_num_subparts, d_total_subparts, d_total_trip);
	return hipGetLastError();
}

hipError_t label_hits_read(unsigned long long *out, bool reset)
{
#ifdef MM2C_LABEL_COUNT
	hipError_t e = hipDeviceSynchronize();
	if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_label_hits), sizeof(unsigned long long) * 256);
	if (e == hipSuccess && reset) {
		static const unsigned long long zero[256] = {};
		e = hipMemcpyToSymbol(HIP_SYMBOL(g_label_hits), zero, sizeof(zero));
	}
	return e;
#else
	(void)out; (void)reset;
	return hipErrorNotSupported;
#endif
}

int chain_ring_anchors(int ring_class) { return ring_class == 0 ? 256 : ring_class == 1 ? 512 : ring_class == 2 ? 1024 : 64 * (MM2C_NX - 1); }   // 3, 4: the tile kernel

hipError_t launch_chain_dp(const LaunchArgs &L_in, hipStream_t st, int *n_launches, hipEvent_t ev_dp_begin, LaunchInfo *info)
{
	if (L_in.n_tasks <= 0) return hipSuccess;
	LaunchArgs L = L_in;
	// The early exit of chain.c:231 can never fire when the skip counter cannot exceed max_skip inside one window: a window holds at most max_iter candidates and the
	// nearest one is never stamped, so the counter stays below max_iter.  Such calls (the V2 scalars of run_chaining_on_hw: max_skip = INT_MAX, max_iter = 1024) used to
	// take the instantiations without the max-skip machinery, which have no hand-written loop; with max_skip = max_iter - 1 the machinery is compiled in and runs, still
	// cannot fire, and the hand-written loop serves them (same f / p, V2 scalars: mixed 47.3 -> 44.3 ms per 1.6e8 anchors, dense 85.4 -> 67.5).
	const bool want_gen = L.P.is_cdna || L.P.n_segs > 1 || (L.P.flags & KF_FORCE_GENERAL);
	const bool loop_ok = L.P.gap_scale == 1.0f || L_in.force_tab || (L.P.bw <= 511 && L.P.gap_scale > -20.f && L.P.gap_scale < 20.f);   // gap cost computed or from the table
	if (L.noskip_loop && (int64_t)L.P.max_skip >= (int64_t)L.P.max_iter && L.P.max_iter >= 1 && L.ring_class >= 3 && !want_gen && L.P.bw >= 0 && L.P.max_dq - 1 >= L.P.bw && loop_ok)
		L.P.max_skip = L.P.max_iter - 1;
	const KParams &P = L.P;
	const bool skip = (int64_t)P.max_skip < (int64_t)P.max_iter;
	const bool gs1 = P.gap_scale == 1.0f;
	const int R = chain_ring_anchors(L.ring_class);
	const bool tile = L.ring_class >= 3;                   // second-generation kernel: 448 anchors before the current tile without global memory
	// the general variant (segment ids / cDNA) has no hand-written loop in the tile kernel and is faster in the first-generation one (headline
	// stream with --general: 92.0 vs 108.7 ms, dense 151.8 vs 163.8): ring_class 3 sends it there, ring_class 4 keeps it in the tile kernel
	const bool tile_gen = L.ring_class >= 4;
	const bool far_ = (int64_t)P.max_iter > (int64_t)R;   // the ring always holds the R anchors before the current tile
	const bool far_old = (int64_t)P.max_iter > 256;       // ... of the first-generation kernel when it stands in (R = 256)
	// the gap-cost table of the tile kernel: dd <= bw <= 511 entries of int16 (cost <= 2.55 * 511 + 4, times gap_scale)
	// used when gap_scale != 1 (it takes the f64 path of chain.c:219 out of the loop); with gap_scale 1 computing the cost is as fast and the
	// kernel's LDS stays at 6 KB (measured: 62.8 vs 66.3 ms on the headline batch)
	static const bool force_tab_env = getenv("MM2C_FORCE_TAB") != nullptr;   // experiment switch: the table also for gap_scale == 1
	const bool force_tab = force_tab_env || L.force_tab != 0;
	const bool tab = tile && (!gs1 || force_tab) && P.bw >= 0 && P.bw <= 511 && P.gap_scale > -20.f && P.gap_scale < 20.f;
	// several waves per task: asked for by the caller for a pass of few tasks; the variants with the hand-written loop, tasks not cut on the device
	const bool coop_cfg = tile && !want_gen && skip && (gs1 || tab) && P.bw >= 0 && P.max_dq - 1 >= P.bw;
	const bool coop = L.coop_waves > 1 && coop_cfg && L.cut.max_pieces == 0;
	// ... or left to the device: with a cut, how many pieces there are and how long is only known there (chain_route, after chain_cut)
	const bool coop_auto = L.coop_waves < 0 && coop_cfg && L.cut.max_pieces > 0 && L.cut.d_count != nullptr;
	if (coop_auto) { L.cut.d_live = L.cut.d_count + 1; L.cut.w8_above = L.coop_w8_above; }
	if (info) {
		info->route_auto = coop_auto ? 1 : 0;
		info->fused_st = coop && coop_makes_st(L) ? 1 : 0;
		info->host_out = (coop && L.h_flag && L.h_f && L.h_p && L.d_done && (L.P.flags & KF_IGNORE_SEG)) ? 1 : 0;
		info->single_ok = info->fused_st && info->host_out && L.d_avg != nullptr;   // such a pass can do without stage_in (LaunchArgs::h_anchors)
		const bool t0 = tile && (!want_gen || tile_gen);          // pass 0 runs in the tile kernel
		info->coop = coop ? coop_width(L) : 0;
		info->tile = t0; info->nx = t0 ? MM2C_NX : 0; info->nf = t0 ? MM2C_NF : 0; info->r = t0 ? 64 * (MM2C_NX - 1) : (tile ? 256 : R);
		info->skip = skip; info->gen = want_gen; info->gs1 = gs1; info->far_ = t0 ? far_ : (tile ? far_old : far_); info->tab = t0 && tab && !want_gen;
		info->asm_loop = t0 && skip && !want_gen && (gs1 || tab) && P.bw >= 0 && P.max_dq - 1 >= P.bw;   // = ASM of chain_dp_tile
		info->classes = t0 && use_classes(L, skip, want_gen);
		info->c16 = t0 && compact_q_span(L, info->asm_loop != 0) != 0;
		info->q24 = t0 && info->classes && q24_ring(L, skip, want_gen, gs1, tab);
		info->cut = L.cut.max_pieces > 0;
		if (coop) { info->nx = COOP_NX; info->nf = COOP_NF; info->r = 64 * (COOP_NX - 1); info->far_ = (int64_t)P.max_iter > 64 * (COOP_NX - 1); info->classes = 0; info->c16 = 0; info->q24 = 0; }
	}
	if (L.dry_run) return hipSuccess;                                  // the caller only wanted to know (info)
	// avg_qspan_scaled per task: the caller's, or computed by the prepass into the workspace (else the DP kernel sweeps the task itself)
	float *avg_out = L.d_avg ? nullptr : L.d_avg_ws;
	const float *d_avg = L.d_avg ? L.d_avg : L.d_avg_ws;
	const unsigned c16_bound = tile && (!want_gen || tile_gen) ? compact_q_span(L, skip && !want_gen && (gs1 || tab) && P.bw >= 0 && P.max_dq - 1 >= P.bw) : 0u;
	// a pass of few tasks that wants nothing but st[] from the prepass (the cooperative kernel: no classes; avg handed in; no cut on the device): one block per tile
	// (avg not handed in: the blocks add up the spans into the avg workspace and a small kernel finishes them -- a plan of few long tasks: 256 reads of 10^6 anchors
	// 3.9 -> about 1 ms, one block per task walks its tiles one after the other)
	const bool wide_prepass = coop && L.max_task_anchors > 0 && L.max_task_anchors <= (1 << 22) && (L.d_avg != nullptr || L.d_avg_ws != nullptr) && L.cut.max_pieces == 0
	                          && (L.max_task_anchors + 255) / 256 <= 65535;
	const bool no_prepass = coop && ((L.st_ready && L.d_avg != nullptr) || coop_makes_st(L));
	if (no_prepass) {
		// nothing to launch: st[] came with the pass (mm2chain_host.cpp) and avg was handed in, or the cooperative kernel makes both itself (short tasks); it has no classes
	} else if (wide_prepass) {
		unsigned *sums = L.d_avg ? nullptr : (unsigned *)L.d_avg_ws;
		if (sums && hipMemsetAsync(sums, 0, (size_t)L.n_tasks * 4, st) != hipSuccess) return hipGetLastError();
		hipLaunchKernelGGL(chain_window_start_wide, dim3((unsigned)L.n_tasks, (unsigned)((L.max_task_anchors + 255) / 256)), dim3(256), 0, st, P, L.n_tasks, L.d_offsets,
		                   (const ulonglong2 *)L.d_anchors, L.d_st, sums);
		if (sums) hipLaunchKernelGGL(chain_avg_finish, dim3((unsigned)((L.n_tasks + 255) / 256)), dim3(256), 0, st, L.n_tasks, L.d_offsets, sums);
	} else {
		// long tasks (the caller knows the longest and lends the words for the sums): a block per segment of PREPASS_SEG anchors instead of a block per task
		// (few tasks only: 2 048 blocks fill the GPU as they are -- 2 048 tasks of 100 000 anchors 1.12 ms by task, 1.46 by segment; 255 of 10^6: 4.7 -> 1.6 ms)
		const bool seg = L.d_seg_ws != nullptr && L.n_tasks <= 512 && L.longest_task >= 2 * PREPASS_SEG && (L.longest_task + PREPASS_SEG - 1) / PREPASS_SEG <= 65535;
#define MM2C_WS(SEG, GRID, SEGN, WS) hipLaunchKernelGGL(chain_window_start_t<SEG>, GRID, dim3(256), 0, st, P, L.n_tasks, L.d_offsets, L.d_order, \
	                   (const ulonglong2 *)L.d_anchors, L.d_st, L.cut.max_pieces > 0 ? L.cut.d_has_cut : (int32_t *)nullptr, avg_out, \
	                   tile && !coop ? L.d_cls : (uint8_t *)nullptr, L.far_ring, L.far_thr10, tile && !coop ? L.d_cls_stat : (unsigned long long *)nullptr, \
	                   coop ? 0u : c16_bound,                     /* (the cooperative kernel has one ring form: no classes to find) */ \
	                   (!coop && tile && q24_ring(L, skip, want_gen, gs1, tab)) ? 1 : 0, SEGN, WS)
		if (seg) MM2C_WS(true, dim3((unsigned)L.n_tasks, (unsigned)((L.longest_task + PREPASS_SEG - 1) / PREPASS_SEG)), PREPASS_SEG, L.d_seg_ws);
		else MM2C_WS(false, dim3((unsigned)L.n_tasks), 0, (unsigned long long *)nullptr);
#undef MM2C_WS
	}
	hipError_t e = hipGetLastError();
	if (n_launches && !no_prepass) ++*n_launches;
	if (e == hipSuccess && tile && !coop && L.d_cls && L.d_cls_stat && (L.far_ring == 1 || c16_bound != 0)) {
		hipLaunchKernelGGL(chain_cls_settle, dim3((unsigned)((L.n_tasks + 255) / 256)), dim3(256), 0, st, L.n_tasks, L.d_cls, L.d_cls_stat, L.far_ring == 1 ? 1 : 0,
		                   c16_bound != 0 ? L.wide_pct : 100);
		e = hipGetLastError();
		if (n_launches) ++*n_launches;
	}
	if (e == hipSuccess && L.cut.max_pieces > 0) {
		hipLaunchKernelGGL(chain_cut, dim3((unsigned)L.n_tasks), dim3(64), 0, st, L.cut.seg_min, L.n_tasks, L.d_offsets, L.d_order,
		                   (const uint4 *)L.d_anchors, d_avg, L.d_st, L.cut, tile ? (const uint8_t *)L.d_cls : (const uint8_t *)nullptr);   // (far_ring 0: bit 1, the 32-bit ring, still counts)
		e = hipGetLastError();
		if (n_launches) ++*n_launches;
	}
	if (e == hipSuccess && coop_auto) {
		hipLaunchKernelGGL(chain_route, dim3(1), dim3(256), 0, st, L.cut);
		e = hipGetLastError();
		if (n_launches) ++*n_launches;
	}
	if (e == hipSuccess && ev_dp_begin) e = hipEventRecord(ev_dp_begin, st);
	LaunchArgs L1 = L; L1.d_avg = d_avg;
	for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
		// pass 0: the variant the parameters ask for; pass 1 (simple variant only, segments not ignored): redo the
		// tasks that turned out to carry more than one segment id with the general variant.
		const bool gen = want_gen || pass == 1;
		if (pass == 1 && (want_gen || (P.flags & KF_IGNORE_SEG))) break;
		const int flagged = pass;
		if (pass == 1) { L.cut.d_live = nullptr; L1.cut.d_live = nullptr; }   // the pieces flagged for the general variant come from either kernel: every piece is looked at
		if (coop && !gen) { e = launch_coop(L, d_avg, st, tab, flagged, n_launches); continue; }
		if (coop_auto && !gen) e = launch_coop(L, d_avg, st, tab, flagged, n_launches);   // (and the one-wave kernels below: each goes by its own count)
		if (tile && (!gen || tile_gen)) { e = launch_tile(L, d_avg, st, skip, gen, gs1, far_, tab, flagged, n_launches); if (n_launches) ++*n_launches; continue; }
		if (tile) { e = launch_r<256>(L1, st, skip, gen, gs1, far_old, flagged); if (n_launches) ++*n_launches; continue; }
		switch (R) {
		case 256: e = launch_r<256>(L1, st, skip, gen, gs1, far_, flagged); break;
		case 512: e = launch_r<512>(L1, st, skip, gen, gs1, far_, flagged); break;
		default:  e = launch_r<1024>(L1, st, skip, gen, gs1, far_, flagged); break;
		}
		if (n_launches) ++*n_launches;
	}
	return e;
}

hipError_t warm_chain_kernels()
{
	hipFuncAttributes at;
	return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&chain_window_start_wide));
}

} // namespace mm2c
