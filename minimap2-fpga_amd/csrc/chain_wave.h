// chain_wave.h -- wave64 primitives and the order-dependent part of the chaining DP, shared by the DP kernels (chain_kernel.hip).
// Everything here is per wave: lane predicates are 64-bit masks in SGPR pairs, scores live one per lane.
#ifndef MM2C_CHAIN_WAVE_H
#define MM2C_CHAIN_WAVE_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "chain_kernel.h"

namespace mm2c {

typedef unsigned long long mask_t;       // one bit per lane, lives in an SGPR pair
#define SENT INT_MIN                     // score of a lane that is not a candidate

// ---------------------------------------------------------------- wave64 primitives (DPP, gfx9 encodings)
// dpp_ctrl: row_shr:n = 0x110+n, wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143
__device__ __forceinline__ int wave_shr1(int lane0_value, int v)
{
	return __builtin_amdgcn_update_dpp(lane0_value, v, 0x138, 0xf, 0xf, false);
}

// shift the chunk-0 window one lane up and put a wave-uniform value into lane 0 (2 VALU)
__device__ __forceinline__ int window_push(int w, int lane0_value)
{
	w = __builtin_amdgcn_update_dpp(w, w, 0x138, 0xf, 0xf, false);
	asm("v_writelane_b32 %0, %1, 0" : "+v"(w) : "s"(lane0_value));
	return w;
}
// inclusive prefix max over ascending lanes (6 v_max_i32_dpp)
__device__ __forceinline__ int prefix_max_incl(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp(SENT, v, 0x111, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(SENT, v, 0x112, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(SENT, v, 0x114, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(SENT, v, 0x118, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(SENT, v, 0x142, 0xa, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(SENT, v, 0x143, 0xc, 0xf, false));
	return v;
}

#define BALLOT(c) ((mask_t)__builtin_amdgcn_ballot_w64(c))
__device__ __forceinline__ int lanes_below(mask_t m)   // number of set bits of m in lanes below this one
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
// per-lane select by a scalar lane mask: bit set -> b, clear -> a
__device__ __forceinline__ int sel(mask_t m, int a, int b)
{
	int r;
	asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
	return r;
}
// |a - b| for unsigned operands (one VALU)
__device__ __forceinline__ int absdiff(int a, int b)
{
	int r;
	asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
// lanes 0..n-1 (the lanes whose predecessor index is still >= the window start)
__device__ __forceinline__ mask_t first_lanes(int n)   // n >= 1
{
	int sh = 64 - n;
	sh = sh < 0 ? 0 : sh;
	return ~0ull >> sh;
}

struct Carry { int best, best_j, n_skip; };

// ---------------------------------------------------------------- filters of chain.c:202-206 as a lane mask
// dr = x_i - x_j (low words; exact inside the window), dq = q_i - q_j.  `ok` = lanes inside the window.
// For a lane inside the window 0 <= dr <= max_dist_x.
template <bool GEN>
__device__ __forceinline__ mask_t pair_filter(const KParams &P, mask_t ok, int dr, int dq, int dd, mask_t same)
{
	if (!GEN) {
		// same segment, genomic: dr != 0, 0 < dq <= min(max_dist_y, max_dist_x), dd <= bw
		ok &= BALLOT(dr != 0);
		ok &= BALLOT((unsigned)(dq - 1) < (unsigned)P.max_dq);
		ok &= BALLOT(dd <= P.bw);
		return ok;
	}
	const mask_t dr0 = BALLOT(dr == 0);
	ok &= ~(same & dr0) & BALLOT(dq > 0);                                       // chain.c:202
	ok &= ~(same & BALLOT(dq > P.max_dist_y)) & BALLOT(dq <= P.max_dist_x);     // chain.c:203
	ok &= ~(same & BALLOT(dd > P.bw));                                          // chain.c:205
	if (P.n_segs > 1 && !P.is_cdna) ok &= ~(same & BALLOT(dr > P.max_dist_y));  // chain.c:206
	return ok;
}

// ---------------------------------------------------------------- score of a pair, chain.c:207-219, WITHOUT f[j]
template <bool GEN, bool GS1>
__device__ __forceinline__ int pair_score(const KParams &P, float avg, int dr, int dq, int dd, mask_t same, int span_i)
{
	int s = min(min(dq, dr), span_i);                                 // chain.c:207-208
	const int c = __builtin_clz((unsigned)dd | 1u);                   // chain.c:209: log_dd = dd ? ilog2_32(dd) : 0 = 31 - c
	const int lin = (int)((float)dd * avg);                           // f32 multiply, truncate
	int gap;
	if (GEN) {
		const int lg = 31 - c;
		const int g_same = lin + (lg >> 1);                           // chain.c:216,218
		if (P.is_cdna) {                                              // chain.c:211-217 with is_cdna
			const int g_cdna = dr > dq ? min(lin, lg) : g_same;
			const int g_diff = dr == 0 ? 0 : min(lin, lg);
			gap = sel(same, g_diff, g_cdna);
		} else {
			const int g_diff = dr == 0 ? 0 : min(lin, lg);            // sidi != sidj
			gap = sel(same, g_diff, g_same);
		}
		s += sel(same, dr == 0 ? 1 : 0, 0);                           // chain.c:214 `++sc`
	} else gap = lin + 15 - (c >> 1);                                 // (31 - c) >> 1 == 15 - (c >> 1) for c in 0..31
	if (GS1) s -= gap;                                                // (int)((double)g*1.0+.499) == g for g >= 0
	else s -= (int)__dadd_rn(__dmul_rn((double)gap, (double)P.gap_scale), .499); // chain.c:219
	return s;
}

// ---------------------------------------------------------------- the order-dependent part of one chunk
// scv: score per lane (SENT where the lane is not a candidate), marked: lanes with t[j] == i.
// Updates the carry exactly as chain.c:226-232 would after walking the lanes in ascending order.
// Returns true when the reference loop executes `break` inside this chunk.
// a chunk in which no lane raises the best (chain.c:226 never taken): every marked lane is a skip event (chain.c:229-231)
template <bool SKIP>
__device__ __forceinline__ bool skips_only(const KParams &P, mask_t se, Carry &c)
{
	if (SKIP && se != 0) {
		const int64_t need = (int64_t)P.max_skip - c.n_skip;               // the event of this 0-based rank breaks
		if (need < (int64_t)__builtin_popcountll(se)) return true;
		c.n_skip += (int)__builtin_popcountll(se);
	}
	return false;
}

template <bool SKIP, bool PRETEST>
__device__ __forceinline__ bool fold_chunk(const KParams &P, int jtop, mask_t valid, mask_t marked, int scv, Carry &c)
{
	// Most older chunks hold no score above the running best (the scan is nearest-first and chains grow from near predecessors): then
	// no lane takes chain.c:226, every marked lane is a skip event, and the counter needs no scan at all.  (Not worth a test in
	// chunk 0, which usually does raise the best.)
	if (PRETEST && BALLOT(scv > c.best) == 0) return skips_only<SKIP>(P, marked & valid, c);
	const int incl = prefix_max_incl(scv);
	int last = 63;                                                    // last lane the reference visits in this chunk
	bool broke = false;
	if (SKIP) {
		const mask_t cand = marked & valid;
		if (cand != 0 || c.n_skip > 0) {
			const int run = max(c.best, wave_shr1(SENT, incl));       // best before this lane, in scan order
			const mask_t nm = BALLOT(scv > run);                      // chain.c:226 takes the branch
			const mask_t se = cand & ~nm;                             // chain.c:229-230 `++n_skip`
			if (se == 0) {
				c.n_skip = max(c.n_skip - (int)__builtin_popcountll(nm), 0);
			} else if (nm == 0 || (63 - (int)__builtin_clzll(nm)) < (int)__builtin_ctzll(se)) {
				// every new best precedes every skip event: counter = max(n - #nm, 0) + rank of the event
				const int n1 = max(c.n_skip - (int)__builtin_popcountll(nm), 0);
				const int64_t need = (int64_t)P.max_skip - n1;         // the event of this 0-based rank breaks
				if (need < (int64_t)__builtin_popcountll(se)) {
					const int r = need < 0 ? 0 : (int)need;
					const mask_t hit = se & BALLOT(lanes_below(se) == r);
					last = (int)__builtin_ctzll(hit) - 1; broke = true;
				} else c.n_skip = n1 + (int)__builtin_popcountll(se);
			} else {
				// general interleaving: Lindley recursion n <- max(n + d, 0), d = +1 (se) / -1 (nm)
				const int S = lanes_below(se) - lanes_below(nm) + sel(se, 0, 1) - sel(nm, 0, 1);
				const int nl = S + max(c.n_skip, prefix_max_incl(-S));
				const mask_t brk = se & BALLOT(nl > P.max_skip);      // chain.c:230-231
				if (brk != 0) { last = (int)__builtin_ctzll(brk) - 1; broke = true; }
				else c.n_skip = rdlane(nl, 63);
			}
		}
	}
	if (last >= 0) {
		const int mc = rdlane(incl, last);                            // best over the visited lanes of this chunk
		if (mc > c.best) {                                            // strict: nearest j wins ties (chain.c:226)
			c.best = mc;
			c.best_j = jtop - (int)__builtin_ctzll(BALLOT(scv == mc));
		}
	}
	return broke;
}


// ---------------------------------------------------------------- fold_lean: the same order-dependent step as fold_chunk, arranged for the
// scalar unit (the DP is bound by SALU issue, tools/ubench/issue_rate.hip): three paths, no flags carried between them.
//   A  no lane beats the running best (chain.c:226 never taken): every marked lane is a skip event (chain.c:229-231)
//   B1 some lane does, nothing is marked and the skip counter is 0: plain max / first argmax
//   B2 the general case: prefix max -> lanes that raise the best (nm), skip events (se), the counter by a max-plus scan over the lanes
template <bool SKIP>
__device__ __forceinline__ bool fold_lean(const KParams &P, int jtop, mask_t marked, int scv, Carry &c)
{
	if (BALLOT(scv > c.best) == 0) {
		if (SKIP && marked != 0) {
			c.n_skip += (int)__builtin_popcountll(marked);
			return c.n_skip > P.max_skip;                              // the `break` of chain.c:231 (n_skip < 2^31 - 64, no overflow)
		}
		return false;
	}
	const int incl = prefix_max_incl(scv);
	if (!SKIP || (marked == 0 && c.n_skip == 0)) {
		const int mc = rdlane(incl, 63);
		c.best = mc;
		c.best_j = jtop - (int)__builtin_ctzll(BALLOT(scv == mc));     // strict: the nearest j wins ties (chain.c:226)
		return false;
	}
	const int run = max(c.best, wave_shr1(SENT, incl));                // best before this lane, in scan order
	const mask_t nm = BALLOT(scv > run);                               // chain.c:226 takes the branch
	const mask_t se = marked & ~nm;                                    // chain.c:229-230 `++n_skip`
	int last = 63;
	if (se == 0) c.n_skip = max(c.n_skip - (int)__builtin_popcountll(nm), 0);
	else {
		// Lindley recursion n <- max(n + d, 0), d = +1 (se) / -1 (nm): n after lane L = S_L + max(n0, max_{l<=L} -S_l), S = prefix sum of d
		const int S = lanes_below(se) - lanes_below(nm) + sel(se, 0, 1) - sel(nm, 0, 1);
		const int nl = S + max(c.n_skip, prefix_max_incl(-S));
		const mask_t brk = se & BALLOT(nl > P.max_skip);               // chain.c:230-231
		if (brk != 0) last = (int)__builtin_ctzll(brk) - 1;
		else c.n_skip = rdlane(nl, 63);
	}
	if (last >= 0) {
		const int mc = rdlane(incl, last);                              // best over the visited lanes of this chunk
		if (mc > c.best) {
			c.best = mc;
			c.best_j = jtop - (int)__builtin_ctzll(BALLOT(scv == mc));
		}
	}
	return last != 63;
}

} // namespace mm2c
#endif
