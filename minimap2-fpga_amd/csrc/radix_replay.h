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
// no replay.  A pass over two buckets has a closed form (all lanes); a pass over more buckets is the reference's loop on one lane.
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
// j < i of the sorted array with key[j] == key[j+1]; work: 4 ints per record (only with TWO_BUCKET); s_cur, s_lo, s_hi: 256 ints of LDS
// each, private to the calling wave.  All synchronisation inside is wave-local, so several waves of a workgroup may replay different
// buckets at the same time.

// LDS and global memory written by some lanes of the wave, read by others
__device__ __forceinline__ void rp_wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_wave_barrier();
}

// One pass of the reference's sort over the bucket [lo, hi) (which must hold equal keys).  Sub-buckets that need the next pass are appended
// to out_list (two ints each) through the counter *out_count.
template <typename IdT, bool TWO_BUCKET>
__device__ void replay_bucket(const uint64_t *un_x, int un_stride, const uint64_t *sorted_x, int sorted_stride, const int32_t *tiecnt, int lo, int hi,
                              IdT *id, uint8_t *dg, int32_t *work, int lane, int *s_cur, int *s_lo, int *s_hi, int32_t *out_list, int *out_count)
{
	// the keys of a bucket are the keys of the same positions of the sorted array: smallest and largest differ first in the
	// highest byte in which any two differ; the passes above that byte move nothing (one bucket each, ksort.h:117-131)
	const uint64_t diff = sorted_x[(int64_t)lo * sorted_stride] ^ sorted_x[(int64_t)(hi - 1) * sorted_stride];
	if (diff == 0) return;                                                 // all equal: every pass is a no-op
	const int shift = (63 - __clzll(diff)) & ~7;
	for (int d = lane; d < 256; d += 64) s_cur[d] = 0;
	rp_wave_sync();
	for (int q0 = lo; q0 < hi; q0 += 256) {                                  // four gathers in flight per lane
		uint64_t x[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { const int q = q0 + 64 * k + lane; x[k] = q < hi ? un_x[(int64_t)id[q] * un_stride] : 0; }
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int q = q0 + 64 * k + lane;
			if (q < hi) { const int d = (int)(x[k] >> shift) & 255; dg[q] = (uint8_t)d; atomicAdd(&s_cur[d], 1); }
		}
	}
	rp_wave_sync();
	{
		int h[4], sum = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) { h[k] = s_cur[4 * lane + k]; sum += h[k]; }
		int at = lo + rp_incl_scan(sum, lane) - sum;
		rp_wave_sync();
#pragma unroll
		for (int k = 0; k < 4; ++k) { s_lo[4 * lane + k] = at; s_cur[4 * lane + k] = at; at += h[k]; s_hi[4 * lane + k] = at; }
	}
	rp_wave_sync();
	int n_buckets = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) n_buckets += s_hi[4 * lane + k] > s_lo[4 * lane + k];
	for (int o = 32; o > 0; o >>= 1) n_buckets += __shfl_xor(n_buckets, o);
	if (TWO_BUCKET && n_buckets == 2) {
		// Two buckets A | B (the strand byte, often the top position byte): the distribution has a closed form, no lane has to walk.
		// Bucket A is filled first (ksort.h:118).  Its t-th misplaced record starts a cycle: it is dropped at B's cursor, the records
		// of B that follow are pushed one place on until B's t-th misplaced record falls out, and that one comes back to the slot the
		// cycle started from.  So A's misplaced slot t gets B's t-th misplaced record; in B the t-th record from A lands right after
		// B's misplaced slot t-1 (at B's start for t = 0) and the B-records before misplaced slot t move one place up.
		const int da = (int)(sorted_x[(int64_t)lo * sorted_stride] >> shift) & 255, db = (int)(sorted_x[(int64_t)(hi - 1) * sorted_stride] >> shift) & 255;
		const int mid = s_hi[da], sz = hi - lo, half = (sz + 1) / 2 + 1;
		int32_t *g = work + 4 * (int64_t)lo;                                // 4 ints of scratch per position of the bucket
		int32_t *fposA = g, *fidA = g + half, *fposB = g + 2 * half, *fidB = g + 3 * half, *newB = g + 4 * half;
		int F = 0;
		for (int q0 = lo; q0 < mid; q0 += 64) {
			const int q = q0 + lane;
			const bool foreign = q < mid && dg[q] != da;
			const uint64_t m = __ballot(foreign);
			if (foreign) { const int t = F + rp_lanes_before(m); fposA[t] = q; fidA[t] = (int32_t)id[q]; }
			F += __popcll(m);
		}
		int FB = 0;
		for (int q0 = mid; q0 < hi; q0 += 64) {
			const int q = q0 + lane;
			const bool in = q < hi, foreign = in && dg[q] != db;
			const uint64_t m = __ballot(foreign);
			const int t = FB + rp_lanes_before(m);                              // misplaced slots of B before q
			if (foreign) { fposB[t] = q; fidB[t] = (int32_t)id[q]; }
			else if (in) newB[q + (t < F ? 1 : 0) - mid] = (int32_t)id[q];
			FB += __popcll(m);
		}
		rp_wave_sync();
		for (int t = lane; t < F; t += 64) {
			id[fposA[t]] = (IdT)fidB[t];
			newB[(t == 0 ? mid : fposB[t - 1] + 1) - mid] = fidA[t];
		}
		rp_wave_sync();
		for (int q = mid + lane; q < hi; q += 64) id[q] = (IdT)newB[q - mid];
		rp_wave_sync();
	} else if (lane == 0) {                                                  // ksort.h:117-131
		for (int d = 0; d < 256; ) {
			const int bl = s_cur[d];
			if (bl == s_hi[d]) { ++d; continue; }
			int dst = dg[bl];
			if (dst == d) { s_cur[d] = bl + 1; continue; }
			IdT hid = id[bl]; uint8_t hd = (uint8_t)dst;
			do {
				const int at = s_cur[dst]++;
				const IdT nid = id[at]; const uint8_t nd = dg[at];
				id[at] = hid; dg[at] = hd; hid = nid; hd = nd;
				dst = hd;
			} while (dst != d);
			id[s_cur[d]] = hid; dg[s_cur[d]] = hd; ++s_cur[d];
		}
	}
	rp_wave_sync();
	if (shift == 0) return;                                                // ksort.h:132
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int d = 4 * lane + k, bl = s_lo[d], bh = s_hi[d];
		// ksort.h:143: buckets of more than 64 records get the next pass (smaller ones an insertion sort = the final stable sort);
		// those without equal keys end up in their one sorted order whatever happens inside
		if (bh - bl > 64 && tiecnt[bh - 1] - tiecnt[bl] > 0) {
			const int slot = atomicAdd(out_count, 1);
			out_list[2 * slot] = bl; out_list[2 * slot + 1] = bh;
		}
	}
}

// The whole replay on one wave: buckets on a stack (2 * (n / 64 + 2) ints), s_sp one int of LDS.
template <typename IdT, bool TWO_BUCKET>
__device__ void replay_passes(const uint64_t *un_x, int un_stride, const uint64_t *sorted_x, int sorted_stride, const int32_t *tiecnt, int n, IdT *id,
                              uint8_t *dg, int32_t *stack, int32_t *work, int lane, int *s_cur, int *s_lo, int *s_hi, int *s_sp)
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
		replay_bucket<IdT, TWO_BUCKET>(un_x, un_stride, sorted_x, sorted_stride, tiecnt, lo, hi, id, dg, work, lane, s_cur, s_lo, s_hi, stack, s_sp);
	}
}

// The whole replay on the NW waves of a workgroup, level by level: the buckets of one level are independent, wave w takes every NW-th of
// them; between levels one workgroup barrier (every wave reaches it: the loop is bounded by the eight byte positions of a key).
// list_a / list_b: n / 64 + 2 buckets (two ints) each; s_n: two ints of LDS; s_cur / s_lo / s_hi: NW * 256 ints of LDS each.
template <typename IdT, bool TWO_BUCKET, int NW>
__device__ void replay_levels(const uint64_t *un_x, int un_stride, const uint64_t *sorted_x, int sorted_stride, const int32_t *tiecnt, int n, IdT *id,
                              uint8_t *dg, int32_t *list_a, int32_t *list_b, int32_t *work, int tid, int *s_cur, int *s_lo, int *s_hi, int *s_n)
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
		for (int k = wave; k < n_seg; k += NW)
			replay_bucket<IdT, TWO_BUCKET>(un_x, un_stride, sorted_x, sorted_stride, tiecnt, in[2 * k], in[2 * k + 1], id, dg, work, lane,
			                               s_cur + 256 * wave, s_lo + 256 * wave, s_hi + 256 * wave, out, &s_n[cur ^ 1]);
	}
	__syncthreads();
}

} // namespace mm2c
#endif
