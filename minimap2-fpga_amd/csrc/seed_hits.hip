// seed_hits.hip -- seed hits -> sorted anchors on the GPU: anchors are born on the device.
//
// Reference: collect_seed_hits (map.c:215-247): every match (a query minimizer found in the index, mm_match_t map.c:76-81) is
// expanded into one anchor per reference hit (encoding map.c:232-241), then the read's anchors are sorted by x with
// radix_sort_128x (misc.c:155-156 / ksort.h:101-151).  One 64-lane wave per read in every kernel.
//
//   seed_expand : prefix sums of the hit counts, then 64 matches at a time: their hits are enumerated by all lanes (owner match by a
//                 binary search over the 64 starts in LDS), encoded and written in match order = the order the reference fills a[].
//   seed_sort   : stable LSD radix sort of the 16-byte anchors on x (bytes in which all keys agree are skipped; ping-pong between
//                 two global buffers).  Where all x of a read differ this IS the reference's result: a sorted order is unique.
//   seed_ties   : radix_sort_128x is not stable, so a read with equal x values gets that sort's passes replayed: the arrangement is
//                 tracked as (digit, index) pairs in LDS, the cycle-leader distribution of ksort.h:117-131 runs on one lane per
//                 pass (its outcome depends on the order of the swaps), only for buckets that hold equal keys (a bucket is a
//                 position range, so the sorted output tells which ones do); buckets of <= 64 records are insertion-sorted by the
//                 reference (stable), so a final stable sort of the replayed arrangement gives the reference's array.

#include <hip/hip_runtime.h>
#include <climits>
#include "chain_kernel.h"

namespace mm2c {

namespace {

__device__ __forceinline__ int lanes_before(uint64_t m)
{
	return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ int wave_incl_scan(int x, int lane)
{
	for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
	return x;
}
__device__ __forceinline__ uint64_t wave_or(uint64_t v)
{
	for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
	return v;
}

// ---- kernel 1: expansion (map.c:222-243) ----------------------------------------------------------------------------
__global__ __launch_bounds__(64) void seed_expand(SeedArgs A)
{
	__shared__ int s_start[65];
	__shared__ uint32_t s_qpos[64], s_span[64], s_segt[64];
	__shared__ int64_t s_cr[64];
	const int read = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int lane = (int)threadIdx.x;
	const int64_t m0 = A.d_match_off[read], a0 = A.d_anchor_off[read];
	const int nm = (int)(A.d_match_off[read + 1] - m0), na = (int)(A.d_anchor_off[read + 1] - a0);
	const int qlen = A.d_qlen[read];
	const Match *m = A.d_matches + m0;
	ulonglong2 *out = A.unsorted + a0;
	int run = 0;                                                                // anchors written so far (wave uniform)
	for (int c0 = 0; c0 < nm; c0 += 64) {
		const int i = c0 + lane;
		Match q = {};
		if (i < nm) q = m[i];
		const int incl = wave_incl_scan((int)q.n, lane);
		const int total = __shfl(incl, 63);
		__syncthreads();
		s_start[lane] = incl - (int)q.n; s_cr[lane] = q.cr_off; s_qpos[lane] = q.q_pos; s_span[lane] = q.q_span; s_segt[lane] = q.seg_tandem;
		if (lane == 63) s_start[64] = total;
		__syncthreads();
		if (run + total > na) { if (lane == 0) A.status[read] = 1; return; }    // the caller's anchor offsets do not match the hit counts
		for (int t = lane; t < total; t += 64) {
			int lo = 0, hi = 63;                                                // last match of the chunk with start <= t
			while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_start[mid] <= t) lo = mid; else hi = mid - 1; }
			const uint64_t r = A.d_hits[s_cr[lo] + (t - s_start[lo])];
			const uint32_t q_pos = s_qpos[lo], q_span = s_span[lo], segt = s_segt[lo];
			const uint32_t rpos = (uint32_t)r >> 1;
			ulonglong2 a;
			if ((r & 1) == (q_pos & 1)) {                                       // forward strand, map.c:232-234
				a.x = (r & 0xffffffff00000000ULL) | rpos;
				a.y = (uint64_t)q_span << 32 | q_pos >> 1;
			} else {                                                            // reverse strand, map.c:235-238
				a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | rpos;
				a.y = (uint64_t)q_span << 32 | (uint32_t)((uint32_t)qlen - ((q_pos >> 1) + 1 - q_span) - 1);
			}
			a.y |= (uint64_t)(segt >> 1) << 48;                                 // MM_SEED_SEG_SHIFT, map.c:239
			if (segt & 1) a.y |= 1ULL << 42;                                    // MM_SEED_TANDEM, map.c:240
			out[run + t] = a;
		}
		run += total;
	}
	if (run != na && lane == 0) A.status[read] = 1;
}

// ---- stable LSD radix sort of one read's anchors on x by one wave; the result ends in `dst` ------------------------
__device__ void wave_sort_anchors(ulonglong2 *src, ulonglong2 *dst, int n, int lane, int *s_cnt /* 256 ints of LDS */)
{
	if (n <= 0) return;
	uint64_t diff = 0;
	const uint64_t first = src[0].x;
	for (int i = lane; i < n; i += 64) diff |= src[i].x ^ first;
	diff = wave_or(diff);
	ulonglong2 *from = src, *to = dst;
	for (int shift = 0; shift < 64; shift += 8) {
		if (((diff >> shift) & 255) == 0) continue;
		for (int d = lane; d < 256; d += 64) s_cnt[d] = 0;
		__syncthreads();
		for (int i = lane; i < n; i += 64) atomicAdd(&s_cnt[(int)(from[i].x >> shift) & 255], 1);
		__syncthreads();
		{
			int h[4], sum = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) { h[k] = s_cnt[4 * lane + k]; sum += h[k]; }
			int at = wave_incl_scan(sum, lane) - sum;
			__syncthreads();
#pragma unroll
			for (int k = 0; k < 4; ++k) { s_cnt[4 * lane + k] = at; at += h[k]; }
		}
		__syncthreads();
		for (int i0 = 0; i0 < n; i0 += 64) {
			const int i = i0 + lane;
			const bool valid = i < n;
			ulonglong2 rec = {};
			if (valid) rec = from[i];
			const int d = (int)(rec.x >> shift) & 255;
			uint64_t peers = __ballot(valid);
#pragma unroll
			for (int b = 0; b < 8; ++b) {
				const uint64_t bal = __ballot((d >> b) & 1);
				peers &= ((d >> b) & 1) ? bal : ~bal;
			}
			const int rank = lanes_before(peers);
			if (valid) to[s_cnt[d] + rank] = rec;
			__syncthreads();
			if (valid && rank == 0) s_cnt[d] += __popcll(peers);
			__syncthreads();
		}
		{ ulonglong2 *t = from; from = to; to = t; }
	}
	if (from != dst) for (int i = lane; i < n; i += 64) dst[i] = from[i];
	__syncthreads();
}

// ---- kernel 2: sort + tie detection ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void seed_sort(SeedArgs A)
{
	__shared__ int s_cnt[256];
	const int read = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int lane = (int)threadIdx.x;
	const int64_t a0 = A.d_anchor_off[read];
	const int na = (int)(A.d_anchor_off[read + 1] - a0);
	if (A.status[read] != 0) return;
	// the unsorted array must survive (the replay of kernel 3 starts from it): sort a copy
	ulonglong2 *un = A.unsorted + a0, *tmp = A.scratch + a0, *out = A.d_anchors + a0;
	for (int i = lane; i < na; i += 64) tmp[i] = un[i];
	__syncthreads();
	wave_sort_anchors(tmp, out, na, lane, s_cnt);
	bool tie = false;
	for (int i = lane; i + 1 < na; i += 64) tie |= out[i].x == out[i + 1].x;
	if (lane == 0) A.has_ties[read] = 0;
	if (__ballot(tie) && lane == 0) A.has_ties[read] = 1;
}

// ---- kernel 3: replay of radix_sort_128x for reads with equal x ----------------------------------------------------
// The arrangement is an index array id[] (position -> anchor of the unsorted array) with the current digit dg[] beside it.
template <typename IdT>
__device__ void replay_passes(const ulonglong2 *un, const ulonglong2 *sorted, int n, IdT *id, uint8_t *dg, int32_t *stack, int lane,
                              int *s_cur, int *s_lo, int *s_hi, int *s_sp)
{
	for (int i = lane; i < n; i += 64) id[i] = (IdT)i;
	if (lane == 0) { stack[0] = 0; stack[1] = n; stack[2] = 56; *s_sp = 1; }
	for (;;) {
		__syncthreads();
		const int sp = *s_sp;
		if (sp == 0) break;
		const int lo = stack[3 * sp - 3], hi = stack[3 * sp - 2];
		int shift = stack[3 * sp - 1];
		__syncthreads();
		if (lane == 0) *s_sp = sp - 1;
		// a bucket is a range of positions, before and after the sort: equal keys inside it show in the sorted output
		bool tie = false;
		for (int q = lo + lane; q + 1 < hi; q += 64) tie |= sorted[q].x == sorted[q + 1].x;
		if (!__ballot(tie)) continue;                                            // all keys differ: the order inside is the sorted one
		const uint64_t x0 = un[id[lo]].x;
		uint64_t diff = 0;
		for (int q = lo + lane; q < hi; q += 64) diff |= un[id[q]].x ^ x0;
		diff = wave_or(diff);
		if (shift < 56) diff &= (1ull << (shift + 8)) - 1;
		if (diff == 0) continue;                                                 // equal from this byte down: every later pass is a no-op
		shift = (63 - __clzll(diff)) & ~7;                                       // passes above it move nothing (one bucket each)
		for (int d = lane; d < 256; d += 64) s_cur[d] = 0;
		__syncthreads();
		for (int q = lo + lane; q < hi; q += 64) {
			const int d = (int)(un[id[q]].x >> shift) & 255;
			dg[q] = (uint8_t)d;
			atomicAdd(&s_cur[d], 1);
		}
		__syncthreads();
		{
			int h[4], sum = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) { h[k] = s_cur[4 * lane + k]; sum += h[k]; }
			int at = lo + wave_incl_scan(sum, lane) - sum;
			__syncthreads();
#pragma unroll
			for (int k = 0; k < 4; ++k) { s_lo[4 * lane + k] = at; s_cur[4 * lane + k] = at; at += h[k]; s_hi[4 * lane + k] = at; }
		}
		__syncthreads();
		if (lane == 0) {                                                         // ksort.h:117-131
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
		__syncthreads();
		if (shift == 0) continue;                                                // ksort.h:132
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int d = 4 * lane + k;
			if (s_hi[d] - s_lo[d] > 64) {                                       // ksort.h:143; smaller ones: insertion sort = the final stable sort
				const int slot = atomicAdd(s_sp, 1);
				stack[3 * slot] = s_lo[d]; stack[3 * slot + 1] = s_hi[d]; stack[3 * slot + 2] = shift - 8;
			}
		}
	}
}

constexpr int TIE_LDS_MAX = 12288;   // anchors of a read whose replay runs in LDS (3 bytes each)

__global__ __launch_bounds__(64) void seed_ties(SeedArgs A)
{
	__shared__ uint16_t s_id[TIE_LDS_MAX];
	__shared__ uint8_t s_dg[TIE_LDS_MAX];
	__shared__ int s_cur[256], s_lo[256], s_hi[256], s_sp;
	const int read = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	if (A.status[read] != 0 || A.has_ties[read] == 0) return;
	const int lane = (int)threadIdx.x;
	const int64_t a0 = A.d_anchor_off[read];
	const int na = (int)(A.d_anchor_off[read + 1] - a0);
	if (na <= 64) return;                                                        // insertion sort only: stable (ksort.h:149)
	ulonglong2 *un = A.unsorted + a0, *tmp = A.scratch + a0, *out = A.d_anchors + a0;
	int32_t *stack = A.stack + 3 * (a0 / 64 + 2 * (int64_t)read);               // > 64 anchors per pending bucket: na/64 + 2 entries suffice
	if (na <= TIE_LDS_MAX) {
		replay_passes<uint16_t>(un, out, na, s_id, s_dg, stack, lane, s_cur, s_lo, s_hi, &s_sp);
		for (int i = lane; i < na; i += 64) tmp[i] = un[s_id[i]];
	} else {                                                                     // does not fit the LDS: same replay through global memory
		uint32_t *g_id = A.big_id + a0; uint8_t *g_dg = A.big_dg + a0;
		replay_passes<uint32_t>(un, out, na, g_id, g_dg, stack, lane, s_cur, s_lo, s_hi, &s_sp);
		for (int i = lane; i < na; i += 64) tmp[i] = un[g_id[i]];
	}
	__syncthreads();
	wave_sort_anchors(tmp, out, na, lane, s_cur);                                // stable: keeps the replayed order among equal x
}

} // namespace

int seed_tie_lds_max() { return TIE_LDS_MAX; }

hipError_t launch_seed_hits(const SeedArgs &A, hipStream_t st, int *n_launches)
{
	if (A.n_reads <= 0) return hipSuccess;
	const unsigned nr = (unsigned)A.n_reads;
	hipError_t e;
	hipLaunchKernelGGL(seed_expand, dim3(nr), dim3(64), 0, st, A);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	hipLaunchKernelGGL(seed_sort, dim3(nr), dim3(64), 0, st, A);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	hipLaunchKernelGGL(seed_ties, dim3(nr), dim3(64), 0, st, A);
	if (n_launches) *n_launches += 3;
	return hipGetLastError();
}

} // namespace mm2c
