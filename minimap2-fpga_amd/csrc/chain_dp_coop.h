// chain_dp_coop.h -- several waves per task: the chaining DP for passes that cannot fill the GPU with tasks (a lone run_chaining_on_hw / mm_chain_dp call,
// chain.c:103; the few long pieces of a small batch).  Included by chain_kernel.hip after chain_dp_tile.h, whose rings, filters, scan and hand-written loop it uses.
//
// The reference's device kernel scores 128 predecessors of ONE task per pipeline step (device/minimap2_opencl.cl:71-148); chain_dp_tile gives a task one wave,
// so a lone task is bound by that wave's latency (0.3-0.6 us per anchor).  What can run in parallel inside one task, exactly:
//   * Whether a predecessor j passes the filters of anchor i (chain.c:202-205) depends on x and q only, and the score of the pair (chain.c:207-220) on f[j],
//     which is final for every j in a tile before i's own.
//   * The scan order only matters through the early exit (chain.c:226-233): the `break` needs more than max_skip skip events, and a skip event is a candidate
//     that passed the filters.  An anchor with at most max_skip candidates in its whole window can therefore never take the `break`, and without it the loop
//     computes the maximum of f[j] + score over the candidates, the nearest j winning ties (strict `>`, chain.c:226) -- an order-independent reduction.  That
//     is every anchor of a V2 call (max_skip = INT_MAX, what run_chaining_on_hw computes) and the noise anchors of a V1 call, the ones whose scans run through
//     their whole window; the anchors on a chain have many candidates, but their scans end after a tile or two.
// So a workgroup of W waves shares the task's LDS rings, and per tile of 64 anchors:
//   phase A, all waves, anchors dealt round robin: count the candidates of the anchor in its own tile and in the older tiles of its window (stopping as soon
//            as the count passes max_skip) and reduce the older tiles' candidates to (best score, nearest index) -- summary per anchor in LDS;
//   phase B, wave 0, anchor by anchor as before: an anchor whose count allows it folds its own-tile candidates (their f only becomes final here) as a plain
//            maximum and merges the summary (the older tiles are farther: strict `>`); every other anchor -- too many candidates, a window that reaches
//            beyond the ring, an equal-x run that reaches into the tile before -- takes the exact scan: the hand-written loop of chain_dp_tile (the anchors
//            that take the short cut carry bit 31 of their tw word, "not for this loop") or its C++ restatement.
// Two workgroup barriers per tile; results bit-identical to chain_dp_tile (and so to chain.c:184-238): tests/test_gpu_parity.py runs the reference-kernel
// vectors and the parity inputs through this kernel as another route.
#ifndef MM2C_CHAIN_DP_COOP_H
#define MM2C_CHAIN_DP_COOP_H
#include "chain_dp_tile.h"

namespace mm2c {

constexpr int COOP_NX = 16, COOP_NF = 8;    // rings of the cooperative kernel: 960 anchors of look-back in LDS, f / p of the 8 nearest tiles beside them
constexpr int COOP_NEVER = 0x7fffffff;      // candidate count of an anchor that must take the exact scan

// wave-wide maximum of (score, index) with the larger index winning ties; every lane gets the result
__device__ __forceinline__ void wave_max_pair(int &sc, int &j)
{
	long long key = ((long long)sc << 32) | (unsigned)j;             // lanes without a candidate: sc = SENT, the smallest key there is
	for (int o = 32; o > 0; o >>= 1) {
		const long long other = __shfl_xor(key, o);
		key = other > key ? other : key;
	}
	sc = (int)(key >> 32); j = (int)(unsigned)key;
}

template <int W, bool GS1, bool FAR, bool TAB>
__global__ void __launch_bounds__(64 * W)
chain_dp_coop(KParams P, int64_t n_tasks, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
              const uint4 *__restrict__ a_all, const float *__restrict__ avg_in, const int32_t *__restrict__ pbase_in,
              const int32_t *__restrict__ st_all, int32_t *__restrict__ f_all, int32_t *__restrict__ p_all, int32_t *__restrict__ t_all,
              int32_t *__restrict__ status, int only_flagged)
{
	constexpr int NX = COOP_NX, NF = COOP_NF;
	constexpr bool SKIP = true, GEN = false;
	typedef Lds<NX, NF, GEN, TAB, false> LY;
	constexpr int SN = LY::SN;
	constexpr int SUM = LY::BYTES;                           // per anchor of the tile in progress: candidate count, best of the older tiles, its index
	const bool ASM = P.bw >= 0 && P.max_dq - 1 >= P.bw;     // the hand-written loop's three-instruction filter applies (every preset)
	__shared__ __attribute__((aligned(16))) char lds[LY::BYTES + 3 * 64 * 4];   // the kernel's only LDS object: the assembly addresses the rings from 0

	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int64_t task = order ? (int64_t)__builtin_amdgcn_readfirstlane(order[blockIdx.x]) : (int64_t)blockIdx.x;
	// every exit below is taken by all waves of the workgroup or by none: the conditions are the same values in every wave
	if (task >= n_tasks) return;
	if (only_flagged && status[task] == 0) return;
	const int64_t base0 = offsets[task];
	const int n = __builtin_amdgcn_readfirstlane((int)(offsets[task + 1] - base0));
	if (n <= 0) return;
	const uint4 *a = a_all + base0;
	const int32_t *st = st_all + base0;
	int32_t *f = f_all + base0, *p = p_all + base0, *t = FAR ? t_all + base0 : nullptr;
	if ((uint32_t)(uintptr_t)(void *)lds != 0) { if (threadIdx.x == 0) status[task] = 3; return; }   // cannot happen: one LDS object per kernel

	const int pbase = pbase_in ? pbase_in[task] : 0;
	float avg = avg_in ? avg_in[task] : -1.0f;
	if (avg < 0.f) {
		uint64_t sum = 0;
		for (int k = lane; k < n; k += 64) sum += (a[k].w & 0xffu);
		for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
		avg = (float)(__dmul_rn(.01, (double)(float)sum) / (double)n);
	}
	avg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, avg)));
	if (TAB && wv == 0) {
		int16_t *const s_gap = (int16_t *)(lds + LY::GAP);
		for (int dd = lane; dd <= P.bw && dd < 512; dd += 64) {
			const int lg = dd ? 31 - __builtin_clz((unsigned)dd) : 0;
			int g = (int)((float)dd * avg) + (lg >> 1);
			if (P.gap_scale != 1.0f) g = (int)__dadd_rn(__dmul_rn((double)g, (double)P.gap_scale), .499);   // chain.c:219
			s_gap[dd] = (int16_t)(1 - g);
		}
	}

	const int rl = 63 - lane;
	AnchorCtx X;
	X.avg = avg; X.rl = rl; X.seg_i = 0; X.far_mode = 0;
	X.mdq1_v = P.max_dq - 1; X.bw_v = P.bw;
	int sent_v = SENT, mdqbw_v = P.max_dq - 1 - P.bw;
	asm volatile("" : "+v"(X.mdq1_v), "+v"(X.bw_v), "+v"(sent_v), "+v"(mdqbw_v));
	TileMem M;
	M.lds = lds; M.a = a; M.f = f; M.p = p; M.t = t; M.pbase = pbase;
	int *const s_cnt = (int *)(lds + SUM), *const s_best = s_cnt + 64, *const s_j = s_cnt + 128;

	int own_x = 0, own_q = 0, own_g = 0, own_f = 0, own_p = -1;
	int seg0 = 0;
	bool t_ready = false;
	const bool no_pairs = P.max_dq <= 0 || P.bw < 0;

	uint4 cur = (rl < n) ? a[rl] : make_uint4(0, 0, 0, 0);
	int cur_st = (rl < n) ? st[rl] : 0;
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int idx = i0 + rl;
		const int cnt = __builtin_amdgcn_readfirstlane(min(64, n - i0));
		uint4 nxt = make_uint4(0, 0, 0, 0); int nxt_st = 0;
		if (idx + 64 < n) { nxt = a[idx + 64]; nxt_st = st[idx + 64]; }
		int prev_last = rdlane(own_x, 0);
		own_x = (int)cur.x; own_q = (int)cur.z;
		own_g = (cur.w >> 16) & 0xff;
		if (!(P.flags & KF_IGNORE_SEG)) {
			if (i0 == 0) seg0 = rdlane(own_g, 63);
			if (BALLOT(rl < cnt && own_g != seg0)) { if (threadIdx.x == 0) status[task] = 1; return; }   // (every wave sees the same tile: all leave together)
		}
		const int stamp_lo = i0 - 64 * (NX - 1);
		if (wv == 0) {
			for (int s = lane; s < SN / 4; s += 64) ((int *)(lds + LY::ST))[s] = 0;
			const int o = (idx & (SN - 1)) * LY::XS;
			*(int2 *)(lds + LY::XQ + o) = make_int2(own_x, own_q);
			if (FAR) {
				const int reach = rdlane(cur_st, 63);
				if (!t_ready && reach < stamp_lo) {
					for (int z = lane; z < i0; z += 64) t[z] = 0;
					t_ready = true;
				}
				if (t_ready && idx < n) t[idx] = 0;
			}
		}
		const int span_l = P.span_override >= 0 ? P.span_override : (int)(cur.w & 0xff);
		mask_t eq_prev = 0;
		{
			asm volatile("" : "+v"(prev_last));
			const int px = __builtin_amdgcn_update_dpp(prev_last, own_x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
			eq_prev = BALLOT(px == own_x);
			if (i0 == 0) eq_prev &= ~(1ull << 63);
		}
		X.stamp_lo = stamp_lo;
		const int addr0 = ((idx - 64) & (SN - 1)) * LY::XS;
		const int addr0b = ((idx - 128) & (SN - 1)) * LY::XS;
		int lomc_v = stamp_lo - 1;
		asm volatile("" : "+v"(lomc_v));
		const int lo_l = no_pairs ? idx : min(cur_st, idx);
		const mask_t above = ~(eq_prev >> lane);
		const int e_l = above ? (int)__builtin_ctzll(above) : 64;
		const int w_l = min(rl, idx - lo_l);
		const int lo_c = max(lo_l, stamp_lo), bef_l = max(i0 - lo_c, 0);
		int tw_l = max(w_l - e_l, 0) | (min(lane + 1 + e_l, 64) << 8) | (bef_l << 15);
		if (lo_l >= idx) tw_l |= (int)0xa0000000;
		if (FAR && lo_l < stamp_lo) tw_l |= 1 << 30;
		if (e_l > rl) tw_l |= (int)0x80000000;
		const int ownst = idx & (SN - 1);
		const int tx1_l = own_x - 1, tq1_l = own_q - 1;

		__syncthreads();      // the rings hold x / q of this tile and f / p of the tiles before it (wave 0 wrote them); the summaries of the tile before have been read

		// ---------------------------------------------------------------- phase A: candidate counts and the older tiles' best, anchors dealt to the waves
		for (int k = wv; k < cnt; k += W) {
			const int L = 63 - k, i = i0 + k;
			const int lo = rdlane(lo_l, L), e = rdlane(e_l, L);
			int c_tot = 0, b_old = SENT, j_old = -1;
			if (i - lo > 0) {
				if ((FAR && lo < stamp_lo) || e > k || no_pairs) c_tot = COOP_NEVER;     // window beyond the ring / equal-x run into the tile before: the exact scan
				else {
					const int xi1 = rdlane(own_x, L) - 1, qi1 = rdlane(own_q, L) - 1, span_i = rdlane(span_l, L);
					X.xi1 = xi1; X.qi1 = qi1; X.span_i = span_i; X.span1_v = span_i - 1;
					// own tile: anchors i-1-e .. max(lo, i0) sit in lanes L+1+e .. L+w (their f is not final yet: counted only)
					const int w = min(k, i - lo);
					if (w - e > 0) {
						const mask_t m = (w - e >= 64 ? ~0ull : ((1ull << (w - e)) - 1)) << (L + 1 + e);
						const int dr1 = xi1 - own_x, dq1 = qi1 - own_q;
						mask_t same = ~0ull;
						c_tot += (int)__builtin_popcountll(chunk_filter<GEN, false, false>(P, X, m, dr1, dq1, absdiff(dr1, dq1), 0, same));
					}
					// older tiles, nearest first; a count beyond max_skip settles it (the exact scan will run): stop there
					const int before = i0 - lo;
					int best_l = SENT, j_l = -1;
					if (before > 0) {
						const int n_full = before >> 6, part = before & 63;
						int base = i0 - 64, depth = 1, addr = addr0;
						for (int c = n_full + (part ? 1 : 0); c > 0 && (int64_t)c_tot <= (int64_t)P.max_skip; --c) {
							int dr1, dq1;
							ring_dr_dq<LY>(X, M, addr, dr1, dq1);
							const int dd = absdiff(dr1, dq1);
							mask_t same = ~0ull;
							const mask_t in_w = (c == 1 && part) ? first_lanes(part) : ~0ull;
							const mask_t valid = chunk_filter<GEN, false, false>(P, X, in_w, dr1, dq1, dd, 0, same);
							if (valid != 0) {
								c_tot += (int)__builtin_popcountll(valid);
								int fj, pj;
								ring_fp<LY, NF>(M, addr, depth, base, rl, fj, pj);
								int sc;
								if (TAB) {
									const int g = *(const int16_t *)(lds + LY::GAP + (min((unsigned)dd, 511u) << 1));
									sc = add3i(min3i(dq1, dr1, X.span1_v), fj, g);
								} else sc = pair_score<GEN, GS1>(P, avg, dr1 + 1, dq1 + 1, dd, same, span_i) + fj;
								if ((valid >> lane & 1) && sc > best_l) { best_l = sc; j_l = base + rl; }   // strict: a nearer tile keeps a tie (chain.c:226)
							}
							addr = (addr - LY::TILE) & (LY::RB - 1);
							base -= 64; ++depth;
						}
					}
					if ((int64_t)c_tot <= (int64_t)P.max_skip) { wave_max_pair(best_l, j_l); b_old = best_l; j_old = j_l; }
					else c_tot = COOP_NEVER;
				}
			}
			if (lane == 0) { s_cnt[k] = c_tot; s_best[k] = b_old; s_j[k] = j_old; }
		}
		__syncthreads();

		// ---------------------------------------------------------------- phase B: wave 0 walks the tile's anchors; the other waves go on to the next barrier
		if (wv == 0) {
			const int c_l = rl < cnt ? s_cnt[rl] : 0, bo_l = rl < cnt ? s_best[rl] : SENT, jo_l = rl < cnt ? s_j[rl] : -1;   // lane L: the summary of anchor i0 + 63 - L
			// the anchors that take the short cut: at most max_skip candidates in the whole window -> chain.c:231 cannot fire (max_skip < 0: only anchors without any)
			const bool short_l = (int64_t)c_l <= (int64_t)P.max_skip && lo_l < idx && rl < cnt;
			const mask_t shorts = BALLOT(short_l);
			if (short_l) tw_l |= (int)0x80000000;               // "not for the hand-written loop": it hands these anchors back
			const bool tile_far = FAR && BALLOT((tw_l >> 30) & 1) != 0;
			for (int k = 0; k < cnt; ++k) {
				const int L = 63 - k;
				if (!(shorts >> L & 1) && ASM) {
#define MM2C_CALL(FN, LO0) FN<NX, NF>(i0, __builtin_amdgcn_readfirstlane(k), cnt, P.max_skip, avg, f, p, pbase, a, t, own_x, tx1_l, own_q, tq1_l, span_l, lo_c, \
                                 LO0, tw_l, own_f, own_p, addr0, addr0b, lomc_v, ownst, rl, mdqbw_v, X.bw_v, sent_v MM2C_LC_ARG)
#ifdef MM2C_LABEL_COUNT
					int lc_v = 0;
#endif
					if (FAR && tile_far) k = TAB ? MM2C_CALL(scan_tile_asm_tab_far, lo_l) : MM2C_CALL(scan_tile_asm_cmp_far, lo_l);
					else k = TAB ? MM2C_CALL(scan_tile_asm_tab, lo_l) : MM2C_CALL(scan_tile_asm_cmp, lo_l);
#undef MM2C_CALL
					k = __builtin_amdgcn_readfirstlane(k);
					if (k >= cnt) break;
				}
				const int Lk = 63 - k;
				const int i = i0 + k;
				const int xi = rdlane(own_x, Lk), qi = rdlane(own_q, Lk);
				const int span_i = rdlane(span_l, Lk);
				const int lo = rdlane(lo_l, Lk);
				Carry c = { span_i, -1, 0 };                                                         // chain.c:188-190
				if (i - lo > 0) {
					X.xi1 = xi - 1; X.qi1 = qi - 1; X.span_i = span_i; X.span1_v = span_i - 1;
					if (shorts >> Lk & 1) {
						// ---- the short cut: maximum over the own-tile candidates (scan order = ascending lane, the nearest wins ties), then the older tiles' best
						const int e = rdlane(e_l, Lk);
						const int w = min(k, i - lo);
						if (w - e > 0) {
							const mask_t m = (w - e >= 64 ? ~0ull : ((1ull << (w - e)) - 1)) << (Lk + 1 + e);
							const int dr1 = X.xi1 - own_x, dq1 = X.qi1 - own_q;
							const int dd = absdiff(dr1, dq1);
							mask_t same = ~0ull;
							const mask_t valid = chunk_filter<GEN, false, false>(P, X, m, dr1, dq1, dd, 0, same);
							if (valid != 0) {
								int sc;
								if (TAB) {
									const int g = *(const int16_t *)(lds + LY::GAP + (min((unsigned)dd, 511u) << 1));
									sc = add3i(min3i(dq1, dr1, X.span1_v), own_f, g);
								} else sc = pair_score<GEN, GS1>(P, avg, dr1 + 1, dq1 + 1, dd, same, span_i) + own_f;
								const int scv = sel(valid, SENT, sc);
								const int mc = rdlane(prefix_max_incl(scv), 63);
								if (mc > c.best) { c.best = mc; c.best_j = i0 + 63 - (int)__builtin_ctzll(BALLOT(scv == mc)); }
							}
						}
						const int bo = rdlane(bo_l, Lk);
						if (bo > c.best) { c.best = bo; c.best_j = rdlane(jo_l, Lk); }               // the older tiles are farther: strict
					} else {
						mask_t eq_run = 0; bool dr0 = false;
						if (eq_prev != 0) {
							const mask_t r = eq_prev >> Lk;
							const int e = (int)__builtin_ctzll(~r);
							if (e > k) dr0 = true;
							else if (e > 0) eq_run = ((1ull << e) - 1) << (Lk + 1);
						}
						X.lo = lo; X.stamp = i + 1; X.s16 = 1 + k; X.s16_v = X.s16;
						X.far_mode = FAR && lo < stamp_lo;
						if (!dr0) scan_anchor<NX, NF, SKIP, GEN, GS1, FAR, TAB, false, false>(P, X, M, lane, i0, k, eq_run, own_x, own_q, own_g, own_f, own_p, addr0, c);
						else scan_anchor<NX, NF, SKIP, GEN, GS1, FAR, TAB, true, false>(P, X, M, lane, i0, k, 0, own_x, own_q, own_g, own_f, own_p, addr0, c);
					}
				}
				write_lane2(own_f, own_p, __builtin_amdgcn_readfirstlane(c.best), __builtin_amdgcn_readfirstlane(c.best_j), Lk);
			}
			// ---- the finished tile: results leave in coalesced stores and enter the f / p ring
			if (rl < cnt) { f[idx] = own_f; p[idx] = own_p < 0 ? own_p : own_p + pbase; }
			const int o = (idx << 3) & LY::FMASK;
			*(int2 *)(lds + LY::FP + o) = make_int2(own_f - FBIAS, own_p);
		}
		cur = nxt; cur_st = nxt_st;
	}
}

} // namespace mm2c
#endif
