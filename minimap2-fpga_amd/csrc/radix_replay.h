// radix_replay.h -- replay of the passes of klib's radix sort (ksort.h:101-151: rs_sort, in-place MSD byte radix sort with a cycle-leader
// distribution, NOT stable) on an index array, shared by the seed-hit path (radix_sort_128x of the anchors, map.c:245) and the device
// epilogue (radix_sort_128x of the chains' first anchors, chain.c:411).
//
// Only the order among EQUAL keys depends on that algorithm (a sorted order of distinct keys is unique), so the caller first sorts with any
// stable sort and calls this only for arrays that contain equal keys.  The arrangement is tracked as an index array id[] (position -> record
// of the unsorted array) with the current digit dg[] beside it.  A bucket of the reference's sort is a range of positions before and after
// the sort, hence: the sorted array tells in O(1) whether a bucket holds equal keys (prefix count `tiecnt`) and which byte is the highest in
// which its keys differ (clz of smallest ^ largest); buckets without equal keys, passes in which all keys share the digit, and buckets of
// <= 64 records (insertion sort in the reference = stable; the caller's final stable sort of the replayed arrangement does the same) need
// no replay.  A pass over two buckets has a closed form (all lanes); a pass over more buckets is the reference's loop as a walk on one lane
// that reads digits only (replay_walk).
#ifndef MM2C_RADIX_REPLAY_H
#define MM2C_RADIX_REPLAY_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mm2c {

__device__ __forceinline__ int rp_lanes_before(uint64_t m)
{
	return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ int rp_incl_scan(int x, int lane)
{
	for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
	return x;
}

// un_x / sorted_x: the keys of the unsorted and of the (stably) sorted array, `stride` 64-bit words apart; tiecnt[i] = number of positions
// j < i of the sorted array with key[j] == key[j+1]; id: position -> record (any memory); dg: one byte per position, with at least one
// readable byte behind the last position; moved: one int per position; fa, fb: one int per position each (only with TWO_BUCKET);
// s_cur: 576 ints (8-byte aligned; 256 cells and, with the digits in LDS, 256 bytes of links), s_lo: 257 ints of LDS, private to the calling wave.  All synchronisation inside is wave-local, so several waves of a
// workgroup may replay different buckets at the same time.

// LDS and global memory written by some lanes of the wave, read by others
__device__ __forceinline__ void rp_wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_wave_barrier();
}

// ---- the cycle-leader distribution of ksort.h:117-131 as a walk over the buckets -------------------------------------------------------
// What the reference's loop does to a bucket array, stated without the swaps: every bucket d is a queue of its original occupants in
// position order, with ONE cursor that is both where the next occupant is taken from and where the next arrival is put (`l->b++`; the slots
// at and behind a cursor are always untouched originals).  Standing at bucket c: take the occupant at c's cursor (digit d), advance the
// cursor, the record goes to bucket d and lands at d's cursor -- which is the slot whose occupant is taken next, standing at d.  The head
// bucket k (ksort.h:118, filled first, then k + 1 ...) differs in one thing: it gives up its occupant when a cycle starts and receives the
// record that closes the cycle into that same slot (`*k->b++ = tmp`), i.e. one place before its cursor; a record of k that is in place
// (`++k->b`) is a cycle of length one.  When the head's queue is exhausted (only the head's can be: every other bucket still expects as many
// arrivals as it has occupants left) the next bucket that is not exhausted becomes the head.  So the order inside a bucket after the pass is
// the order in which the walk sent records there, and the walk needs the digits only -- n steps of (cursor of c, digit at it) -> next c on
// one lane -- and moves nothing: it writes moved[destination] = source, which all lanes apply afterwards.  The records (16 bytes, or an
// index) are thus never touched by the sequential part, and the only LDS the replay needs is one byte per position.
// s_cur[d] = {cursor of d, the digit at the cursor}: the cursor of the next bucket and the digit behind the current cursor are fetched side
// by side, one LDS round trip per step.
template <bool LDS_DG>
__device__ __forceinline__ void replay_walk(const uint8_t *dg, int lo, int hi, int32_t *moved, int lane, int *s_cnt, const int *s_lo, int *s_buf = nullptr)
{
	int2 *s_cur = (int2 *)s_cnt;                                             // {cursor, digit at the cursor}: read and written in one piece
	if constexpr (LDS_DG) {
		// The loop below written by hand (one lane; the compiler's version of it is twice as long): cells hold the LDS ADDRESS of the cursor's
		// digit, s_nx[d] (bytes behind the 256 cells) links every bucket to the next one that is not empty (0: none) for the change of head,
		// and the destination is stored as moved_base + 4 * address with the address of dg[0] folded into the base.
		uint8_t *s_nx = (uint8_t *)(s_cur + 256);
		const uint32_t dgA = (uint32_t)(uintptr_t)(const void *)dg;
		uint64_t mask[4];
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int d = 64 * r + lane, b = s_lo[d];
			s_cur[d] = int2{(int)dgA + b, (int)dg[b]};                           // (an empty last bucket reads the byte behind the array)
			mask[r] = __ballot(s_lo[d + 1] > b);
		}
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			int nx = 0;
#pragma unroll
			for (int rr = 3; rr >= r; --rr) {
				const uint64_t m = rr == r ? (lane == 63 ? 0 : mask[rr] & (~0ull << (lane + 1))) : mask[rr];
				if (m) nx = 64 * rr + (int)__builtin_ctzll(m);
			}
			s_nx[64 * r + lane] = (uint8_t)nx;
		}
		int head = 0;
#pragma unroll
		for (int r = 3; r >= 0; --r) if (mask[r]) head = 64 * r + (int)__builtin_ctzll(mask[r]);
		rp_wave_sync();
		if (lane == 0) {
			const uint64_t mb0 = (uint64_t)(uintptr_t)moved - 4ull * dgA;
			// (a scalar operand: the pointer is the same in every lane, but the compiler does not always see that and would hand the block a VGPR pair)
			const uint64_t mb = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mb0 >> 32)) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mb0);
			asm volatile(
				"v_mov_b32 v56, %1\n\tv_mov_b32 v57, %2\n\tv_mov_b32 v58, %3\n\tv_mov_b32 v59, %4\n\tv_mov_b32 v60, %0\n\t"
				"s_branch 5f\n"
				"1:\n\t"                                                          // step: state {v40 cell address of c, v41 cursor, v42 digit} -> {v44, v45, v46}
				"v_cmp_eq_u64 vcc, v[40:41], v[52:53]\n\t"                        // standing at the head with its queue exhausted?
				"s_cbranch_vccnz 3f\n\t"
				"v_lshl_add_u32 v44, v42, 3, v56\n\t"                             // the cell of bucket d
				"ds_read_u8 v51, v41 offset:1\n\t"                                // the digit behind the one taken
				"ds_read_b64 v[48:49], v44\n\t"
				"v_add_u32 v50, 1, v41\n\t"
				"v_cmp_eq_u32 vcc, v44, v40\n\t"
				"s_waitcnt lgkmcnt(0)\n\t"
				"ds_write_b64 v40, v[50:51]\n\t"
				"v_cndmask_b32 v45, v48, v50, vcc\n\t"                            // d == c: the cell just written
				"v_cndmask_b32 v46, v49, v51, vcc\n\t"
				"v_cmp_eq_u32 vcc, v44, v52\n\t"                                  // d is the head: one place before its cursor
				"v_sub_u32 v55, v41, v57\n\t"                                     // source position -- and with the s_nop the two wait states between a VALU write of VCC and
				"s_nop 0\n\t"                                                    // a VALU that reads it (gfx940 / gfx950; tools/check_isa_hazards.py, SGPR_VALU)
				"v_subb_co_u32 v54, vcc, v45, 0, vcc\n\t"
				"v_lshlrev_b32 v54, 2, v54\n\t"
				"global_store_dword v54, v55, %5\n\t"
				"v_cmp_eq_u64 vcc, v[44:45], v[52:53]\n\t"                        // the same with the two states exchanged
				"s_cbranch_vccnz 3f\n\t"
				"v_lshl_add_u32 v40, v46, 3, v56\n\t"
				"ds_read_u8 v51, v45 offset:1\n\t"
				"ds_read_b64 v[48:49], v40\n\t"
				"v_add_u32 v50, 1, v45\n\t"
				"v_cmp_eq_u32 vcc, v40, v44\n\t"
				"s_waitcnt lgkmcnt(0)\n\t"
				"ds_write_b64 v44, v[50:51]\n\t"
				"v_cndmask_b32 v41, v48, v50, vcc\n\t"
				"v_cndmask_b32 v42, v49, v51, vcc\n\t"
				"v_cmp_eq_u32 vcc, v40, v52\n\t"
				"v_sub_u32 v55, v45, v57\n\t"
				"s_nop 0\n\t"
				"v_subb_co_u32 v54, vcc, v41, 0, vcc\n\t"
				"v_lshlrev_b32 v54, 2, v54\n\t"
				"global_store_dword v54, v55, %5\n\t"
				"s_branch 1b\n"
				"3:\n\t"                                                          // the next bucket that is not exhausted becomes the head
				"v_add_u32 v61, v59, v60\n\t"
				"ds_read_u8 v60, v61\n\t"
				"s_waitcnt lgkmcnt(0)\n\t"
				"v_cmp_eq_u32 vcc, 0, v60\n\t"
				"s_cbranch_vccnz 9f\n"
				"5:\n\t"
				"v_lshl_add_u32 v52, v60, 3, v56\n\t"
				"v_lshl_add_u32 v61, v60, 2, v58\n\t"
				"ds_read_b64 v[48:49], v52\n\t"
				"ds_read_b32 v53, v61 offset:4\n\t"
				"v_mov_b32 v40, v52\n\t"
				"s_waitcnt lgkmcnt(0)\n\t"
				"v_mov_b32 v41, v48\n\t"
				"v_mov_b32 v42, v49\n\t"
				"v_add_u32 v53, v53, v57\n\t"
				"v_cmp_eq_u32 vcc, v41, v53\n\t"
				"s_cbranch_vccnz 3b\n\t"
				"s_branch 1b\n"
				"9:\n\t"
				"s_waitcnt vmcnt(0) lgkmcnt(0)"
				: : "v"(head), "v"((uint32_t)(uintptr_t)(void *)s_cur), "v"(dgA), "v"((uint32_t)(uintptr_t)(const void *)s_lo), "v"((uint32_t)(uintptr_t)(void *)s_nx), "s"(mb)
				: "memory", "vcc", "v40", "v41", "v42", "v44", "v45", "v46", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61");
		}
		return;
	}
	for (int d = lane; d < 256; d += 64) { const int b = s_lo[d]; s_cur[d] = int2{b - lo, (int)dg[b]}; }   // (an empty last bucket reads the byte behind the array)
	rp_wave_sync();
	if (s_buf) {
		// Long reads (round 6): the walk with its results BUFFERED.  As written below, every step stores moved[destination] = source to memory and loads the digit behind
		// its cursor; vmcnt counts loads and stores in one order, so the wait for the load was a wait for the store of the step before as well.  Here lane 0 walks RP_BUF
		// steps at a time, leaving (destination, source) in LDS, and the 64 lanes store the pairs together; inside the walk only the digit loads are in flight
		// (1 020 reads of 3e5 anchors 147 -> 131 ms).  The digit loads K steps deep on top of this (cursor and digit of a cell written separately, the loads in flight
		// kept in registers): 140 ms -- the step is its own dependent instructions on one lane, about 0.18 us; not kept.
		constexpr int RP_BUF = 256;
		const int n = hi - lo;
		int head = 0;
		while (s_lo[head] == s_lo[head + 1]) ++head;
		int c = head, head_end = s_lo[head + 1] - lo;
		int2 pk = s_cur[c];
		for (int sb = 0; sb < n; sb += RP_BUF) {
			const int m = min(RP_BUF, n - sb);
			if (lane == 0) {
				for (int t = 0; t < m; ++t) {
					if (c == head && pk.x == head_end) {                         // the head's queue is exhausted: the next bucket that is not takes over
						do ++head; while (s_cur[head].x == s_lo[head + 1] - lo);
						c = head; pk = s_cur[c]; head_end = s_lo[head + 1] - lo;
					}
					const int p = pk.x, d = pk.y;
					const int nx = dg[lo + p + 1];
					const int2 pkd = s_cur[d];
					const int2 npk = int2{p + 1, nx};
					s_cur[c] = npk;
					pk = d == c ? npk : pkd;
					*(int2 *)(s_buf + 2 * t) = int2{lo + pk.x - (d == head ? 1 : 0), lo + p};
					c = d;
				}
			}
			rp_wave_sync();
			for (int t = lane; t < m; t += 64) { const int2 e = *(const int2 *)(s_buf + 2 * t); moved[e.x] = e.y; }
			rp_wave_sync();
		}
		return;
	}
	if (lane != 0) return;
	const int n = hi - lo;
	int head = 0;
	while (s_lo[head] == s_lo[head + 1]) ++head;
	int c = head, head_end = s_lo[head + 1] - lo;
	int2 pk = s_cur[c];
	for (int s = 0; s < n; ++s) {
		if (c == head && pk.x == head_end) {                                 // the head's queue is exhausted: the next bucket that is not takes over
			do ++head; while (s_cur[head].x == s_lo[head + 1] - lo);
			c = head; pk = s_cur[c]; head_end = s_lo[head + 1] - lo;
		}
		const int p = pk.x, d = pk.y;
		const int nx = dg[lo + p + 1];                                       // the digit behind it and the cursor of the next bucket: in flight together
		const int2 pkd = s_cur[d];
		const int2 npk = int2{p + 1, nx};
		s_cur[c] = npk;
		pk = d == c ? npk : pkd;
		moved[lo + pk.x - (d == head ? 1 : 0)] = lo + p;
		c = d;
	}
}

// One pass of the reference's sort over the bucket [lo, hi) (which must hold equal keys).  Sub-buckets that need the next pass are appended
// to out_list (two ints each) through the counter *out_count.
template <typename IdT, bool TWO_BUCKET, bool LDS_DG>
__device__ __forceinline__ void replay_bucket(const uint64_t *un_x, int un_stride, const uint64_t *sorted_x, int sorted_stride, const int32_t *tiecnt, int lo, int hi,
                              IdT *id, uint8_t *dg, int32_t *moved, int32_t *fa, int32_t *fb, int lane, int *s_cur, int *s_lo,
                              int32_t *out_list, int *out_count, int *s_buf = nullptr)
{
	// the keys of a bucket are the keys of the same positions of the sorted array: smallest and largest differ first in the
	// highest byte in which any two differ; the passes above that byte move nothing (one bucket each, ksort.h:117-131)
	const uint64_t diff = sorted_x[(int64_t)lo * sorted_stride] ^ sorted_x[(int64_t)(hi - 1) * sorted_stride];
	if (diff == 0) return;                                                 // all equal: every pass is a no-op
	const int shift = (63 - __clzll(diff)) & ~7;
	for (int d = lane; d < 256; d += 64) s_cur[d] = 0;
	rp_wave_sync();
	for (int q0 = lo; q0 < hi; q0 += 512) {                                  // eight gathers in flight per lane
		int r[8]; uint64_t x[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) { const int q = q0 + 64 * k + lane; r[k] = (int)id[q < hi ? q : lo]; }
#pragma unroll
		for (int k = 0; k < 8; ++k) x[k] = un_x[(int64_t)r[k] * un_stride];
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const int q = q0 + 64 * k + lane;
			if (q < hi) { const int d = (int)(x[k] >> shift) & 255; dg[q] = (uint8_t)d; atomicAdd(&s_cur[d], 1); }
		}
	}
	rp_wave_sync();
	int n_buckets = 0;
	{
		int h[4], sum = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) { h[k] = s_cur[4 * lane + k]; sum += h[k]; n_buckets += h[k] > 0; }
		int at = lo + rp_incl_scan(sum, lane) - sum;
#pragma unroll
		for (int k = 0; k < 4; ++k) { s_lo[4 * lane + k] = at; at += h[k]; }
		if (lane == 63) s_lo[256] = at;                                      // = hi
	}
	rp_wave_sync();
	for (int o = 32; o > 0; o >>= 1) n_buckets += __shfl_xor(n_buckets, o);
	if (TWO_BUCKET && n_buckets == 2) {
		// Two buckets A | B (the strand byte, often the top position byte): the distribution has a closed form, no walk.
		// Bucket A is filled first (ksort.h:118).  Its t-th misplaced record starts a cycle: it is dropped at B's cursor, the records
		// of B that follow are pushed one place on until B's t-th misplaced record falls out, and that one comes back to the slot the
		// cycle started from.  So A's misplaced slot t gets B's t-th misplaced record; in B the t-th record from A lands right after
		// B's misplaced slot t-1 (at B's start for t = 0) and the B-records before misplaced slot t move one place up.
		const int da = (int)(sorted_x[(int64_t)lo * sorted_stride] >> shift) & 255, db = (int)(sorted_x[(int64_t)(hi - 1) * sorted_stride] >> shift) & 255;
		const int mid = s_lo[da + 1];
		int32_t *fposA = fa + lo, *fposB = fb + lo;                           // at most min(|A|, |B|) entries each
		int F = 0;
		for (int q0 = lo; q0 < mid; q0 += 64) {
			const int q = q0 + lane;
			const bool in = q < mid, foreign = in && dg[q] != da;
			const uint64_t m = __ballot(foreign);
			if (foreign) fposA[F + rp_lanes_before(m)] = q;
			else if (in) moved[q] = q;
			F += __popcll(m);
		}
		int FB = 0;
		for (int q0 = mid; q0 < hi; q0 += 64) {
			const int q = q0 + lane;
			const bool in = q < hi, foreign = in && dg[q] != db;
			const uint64_t m = __ballot(foreign);
			const int t = FB + rp_lanes_before(m);                              // misplaced slots of B before q
			if (foreign) fposB[t] = q;
			else if (in) moved[q + (t < F ? 1 : 0)] = q;
			FB += __popcll(m);
		}
		rp_wave_sync();
		for (int t = lane; t < F; t += 64) {
			moved[fposA[t]] = fposB[t];
			moved[t == 0 ? mid : fposB[t - 1] + 1] = fposA[t];
		}
	} else replay_walk<LDS_DG>(dg, lo, hi, moved, lane, s_cur, s_lo, LDS_DG ? nullptr : s_buf);      // ksort.h:117-131
	rp_wave_sync();
	// the new arrangement: position q holds the record that stood at moved[q]
	for (int q0 = lo; q0 < hi; q0 += 512) {
		int v[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) { const int q = q0 + 64 * k + lane; v[k] = moved[q < hi ? q : lo]; }
#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] = (int)id[v[k]];
#pragma unroll
		for (int k = 0; k < 8; ++k) { const int q = q0 + 64 * k + lane; if (q < hi) moved[q] = v[k]; }
	}
	rp_wave_sync();
	for (int q = lo + lane; q < hi; q += 64) id[q] = (IdT)moved[q];
	rp_wave_sync();
	if (shift == 0) return;                                                // ksort.h:132
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int d = 4 * lane + k, bl = s_lo[d], bh = s_lo[d + 1];
		// ksort.h:143: buckets of more than 64 records get the next pass (smaller ones an insertion sort = the final stable sort);
		// those without equal keys end up in their one sorted order whatever happens inside
		if (bh - bl > 64 && tiecnt[bh - 1] - tiecnt[bl] > 0) {
			const int slot = atomicAdd(out_count, 1);
			out_list[2 * slot] = bl; out_list[2 * slot + 1] = bh;
		}
	}
}

// The whole replay on one wave: buckets on a stack (2 * (n / 64 + 2) ints), s_sp one int of LDS.
template <typename IdT, bool TWO_BUCKET, bool LDS_DG>
__device__ __forceinline__ void replay_passes(const uint64_t *un_x, int un_stride, const uint64_t *sorted_x, int sorted_stride, const int32_t *tiecnt, int n, IdT *id,
                              uint8_t *dg, int32_t *stack, int32_t *moved, int32_t *fa, int32_t *fb, int lane, int *s_cur, int *s_lo, int *s_sp)
{
	for (int i = lane; i < n; i += 64) id[i] = (IdT)i;
	if (lane == 0) { stack[0] = 0; stack[1] = n; *s_sp = 1; }                  // only buckets that hold equal keys are ever pushed
	for (;;) {
		rp_wave_sync();
		const int sp = *s_sp;
		if (sp == 0) break;
		const int lo = stack[2 * sp - 2], hi = stack[2 * sp - 1];
		rp_wave_sync();
		if (lane == 0) *s_sp = sp - 1;
		rp_wave_sync();
		replay_bucket<IdT, TWO_BUCKET, LDS_DG>(un_x, un_stride, sorted_x, sorted_stride, tiecnt, lo, hi, id, dg, moved, fa, fb, lane, s_cur, s_lo, stack, s_sp);
	}
}

// The whole replay on the NW waves of a workgroup, level by level: the buckets of one level are independent, wave w takes every NW-th of
// them; between levels one workgroup barrier (every wave reaches it: the loop is bounded by the eight byte positions of a key).
// list_a / list_b: n / 64 + 2 buckets (two ints) each; s_n: two ints of LDS; s_cur: NW * 576, s_lo: NW * 257 ints of LDS.
template <typename IdT, bool TWO_BUCKET, bool LDS_DG, int NW>
__device__ __forceinline__ void replay_levels(const uint64_t *un_x, int un_stride, const uint64_t *sorted_x, int sorted_stride, const int32_t *tiecnt, int n, IdT *id,
                              uint8_t *dg, int32_t *list_a, int32_t *list_b, int32_t *moved, int32_t *fa, int32_t *fb, int tid, int *s_cur, int *s_lo, int *s_n, int *s_buf = nullptr /* 512 ints per wave, or none */)
{
	const int lane = tid & 63, wave = tid >> 6;
	for (int i = tid; i < n; i += 64 * NW) id[i] = (IdT)i;
	if (tid == 0) { list_a[0] = 0; list_a[1] = n; s_n[0] = 1; s_n[1] = 0; }
	for (int level = 0; level < 9; ++level) {
		__syncthreads();
		const int cur = level & 1, n_seg = s_n[cur];
		__syncthreads();
		if (n_seg == 0) break;                                                   // uniform: every wave read the same count
		if (tid == 0) s_n[cur ^ 1] = 0;
		__syncthreads();
		int32_t *in = cur ? list_b : list_a, *out = cur ? list_a : list_b;
#ifdef MM2C_REPLAY_PROBE
		if (tid == 0 && blockIdx.x == 0) { int big = 0; long long tot = 0; for (int k = 0; k < n_seg; ++k) { big = max(big, in[2 * k + 1] - in[2 * k]); tot += in[2 * k + 1] - in[2 * k]; }
			printf("replay level %d: %d buckets, largest %d, total %lld, clock %lld\n", level, n_seg, big, tot, (long long)wall_clock64()); }
#endif
		for (int k = wave; k < n_seg; k += NW)
			replay_bucket<IdT, TWO_BUCKET, LDS_DG>(un_x, un_stride, sorted_x, sorted_stride, tiecnt, in[2 * k], in[2 * k + 1], id, dg, moved, fa, fb, lane,
			                               s_cur + 576 * wave, s_lo + 257 * wave, out, &s_n[cur ^ 1], s_buf ? s_buf + 512 * wave : nullptr);
	}
	__syncthreads();
#ifdef MM2C_REPLAY_PROBE
	if (tid == 0 && blockIdx.x == 0) printf("replay end: clock %lld\n", (long long)wall_clock64());
#endif
}

} // namespace mm2c
#endif
