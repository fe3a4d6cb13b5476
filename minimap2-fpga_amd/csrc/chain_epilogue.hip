// chain_epilogue.hip -- the part of mm_chain_dp after the DP, on the GPU.
//
// Reference: chain.c:106-111 (v[] from f[]/p[]) and chain.c:348-422 (chain ends, peak search, sort by score, backtrack with
// used-marks, score/length filter, emission, chains ordered by the x of their first anchor).  The reference walks the chains
// one after the other on one thread; every step below is a data-parallel restatement with the same result:
//
//   v[i]       = max of f over i and its ancestors in the p[] forest          -> pointer jumping inside 64-anchor chunks
//   chain ends = anchors without a child whose v >= min_sc; each walks back to its peak (f[j] >= v[j])   -> one lane per end
//   sort       = descending on (f[peak] << 32 | peak); the keys are compared in full, so any correct sort gives the
//                reference's order (wave_sort64: LSD radix sort by the task's own wave)
//   backtrack  : chain r (rank in that order) takes its peak's ancestors up to the first anchor an earlier chain took.
//                Equivalent closed form: owner(x) = min rank over the peaks in the subtree of x.  (If m is that minimum, chain m
//                cannot have been stopped below x: a stop needs an earlier chain with a peak in a sub-subtree, which would have a
//                smaller rank.)  owner() is a min-reduction towards the roots: LDS atomics + pointer jumping inside a chunk,
//                global atomics across chunks, chunks in descending order.  A chain whose own peak is taken keeps just that
//                peak (the reference's do-while, chain.c:381-383).
//   length     = depth of the peak inside its owner path + 1; the stop anchor is the parent of the path's top
//   filter     = chain.c:385-388; survivors keep their rank order
//   final order= ascending x of the first anchor.  radix_sort_128x (ksort.h:101-151) is not stable for more than 64 records,
//                so a task with more than 64 chains AND two equal first-x values replays that sort's passes on one lane;
//                everywhere else the order is unique (stable wave_sort64 of (x, chain)).
//
// Two forms.  Kernels A / B / C (epi_ends, epi_claim, epi_emit): one 64-lane wave per task, per-anchor state (marks, v, peaks, owners, depths)
// in HBM scratch arrays, chunks of 256 anchors with pointer jumping inside a chunk -- any task size.  epi_fused + epi_emit_cd (round 2, the
// default for tasks of up to 7 680 anchors): one workgroup of 512 threads per task with that state in LDS, pointer jumping over the whole task
// (see epi_fused).  Tasks are independent.  Outputs are compact: chains of task k are u[u_off[k] .. u_off[k+1]), their anchors
// b[b_off[k] .. b_off[k+1]).

#include <hip/hip_runtime.h>
#include <climits>
#include "chain_kernel.h"
#include "radix_replay.h"

namespace mm2c {

namespace {

constexpr int NONE = INT_MAX;

__device__ __forceinline__ int lanes_before(uint64_t m)
{
	return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// A task is handled by one workgroup (one wave), so every atomic and every ordering below is workgroup scope: the atomics run in the
// XCD's own L2 and nothing is written back or invalidated.  (Agent scope, the HIP default, sends them past the L2 -- the L2s of the
// eight XCDs are not coherent with each other -- and made these kernels 3x slower.)
__device__ __forceinline__ void wg_min(int *addr, int val)
{
	(void)__hip_atomic_fetch_min(addr, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int wg_load(const int *addr)   // a load that sees the workgroup's L2 atomics
{
	return __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ int wave_sum(int x)
{
	for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
	return x;
}

__device__ __forceinline__ int wave_incl_scan(int x, int lane)
{
	for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
	return x;
}

// ---- stable LSD radix sort of one task's records by one wave (8-bit digits, ping-pong between two global buffers) ----
// Bytes in which all keys agree are skipped, so (score << 32 | index) keys cost about four passes.  Per 64 records: the lanes with
// the same digit find each other with eight ballots; rank among them = position in lane order (stable).  `vals` may be NULL.
// The result ends in (k1, v1).  No host involvement -- rocPRIM's segmented sort synchronises the stream on the host.
// SOLO: the wave runs alone inside a bigger block (the other waves wait at the next __syncthreads): its own LDS and memory operations are
// ordered by a workgroup fence, without the barrier.
template <bool SOLO>
__device__ __forceinline__ void sort_sync()
{
	if (SOLO) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
	else __syncthreads();
}
#define __syncthreads_sort() sort_sync<SOLO>()
template <bool DESC, bool SOLO = false>
__device__ void wave_sort64(uint64_t *k0, uint64_t *k1, int32_t *v0, int32_t *v1, int m, int lane, int *s_cnt /* 256 ints of LDS */)
{
	if (m <= 0) return;
	uint64_t diff = 0;
	const uint64_t first = k0[0];
	for (int i = lane; i < m; i += 64) diff |= k0[i] ^ first;
	for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o);
	uint64_t *src = k0, *dst = k1;
	int32_t *vsrc = v0, *vdst = v1;
	for (int shift = 0; shift < 64; shift += 8) {
		if (((diff >> shift) & 255) == 0) continue;
		for (int d = lane; d < 256; d += 64) s_cnt[d] = 0;
		__syncthreads_sort();
		for (int i = lane; i < m; i += 64) {
			const int d = (int)(src[i] >> shift) & 255;
			atomicAdd(&s_cnt[DESC ? 255 - d : d], 1);
		}
		__syncthreads_sort();
		{
			int h[4], sum = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) { h[k] = s_cnt[4 * lane + k]; sum += h[k]; }
			int at = wave_incl_scan(sum, lane) - sum;
			__syncthreads_sort();
#pragma unroll
			for (int k = 0; k < 4; ++k) { s_cnt[4 * lane + k] = at; at += h[k]; }
		}
		__syncthreads_sort();
		for (int i0 = 0; i0 < m; i0 += 64) {
			const int i = i0 + lane;
			const bool valid = i < m;
			const uint64_t key = valid ? src[i] : 0;
			const int32_t val = (valid && vsrc) ? vsrc[i] : 0;
			int d = (int)(key >> shift) & 255;
			if (DESC) d = 255 - d;
			uint64_t peers = __ballot(valid);
#pragma unroll
			for (int b = 0; b < 8; ++b) {
				const uint64_t bal = __ballot((d >> b) & 1);
				peers &= ((d >> b) & 1) ? bal : ~bal;
			}
			if (valid) {
				const int rank = lanes_before(peers);
				const int pos = s_cnt[d] + rank;
				dst[pos] = key;
				if (vdst) vdst[pos] = val;
			}
			__syncthreads_sort();
			if (valid && lanes_before(peers) == 0) s_cnt[d] += __popcll(peers);
			__syncthreads_sort();
		}
		{ uint64_t *t = src; src = dst; dst = t; }
		{ int32_t *t = vsrc; vsrc = vdst; vdst = t; }
	}
	if (src != k1) {                                                             // even number of passes: the result is in (k0, v0)
		for (int i = lane; i < m; i += 64) { k1[i] = src[i]; if (v1) v1[i] = vsrc[i]; }
	}
	__syncthreads_sort();
}

#undef __syncthreads_sort

// The three chunked passes below walk a task in chunks of W = 64*K anchors (K per lane), because the passes are sequential from chunk
// to chunk (a chunk needs the finished values of the chunks before it) and every step costs a global-memory round trip: wide
// chunks mean few steps.  Links that stay inside a chunk are resolved by pointer jumping through LDS (log2 W rounds at most).
constexpr int K = 4, W = 64 * K;
// tasks of at most FUSE_L anchors can keep their per-anchor state in LDS (epi_fused below): two size classes
constexpr int FUSE_S = 5120, FUSE_L = 7680;        // 52.9 KB of LDS -> three tasks per CU; 78.8 KB -> two
constexpr int NONE16 = 0xffff;
constexpr int FNT = 512;                           // epi_fused: threads per task
constexpr int RANK_MAX = 768;                       // epi_fused: up to this many keys are ordered by counting (quadratic, but barrier-free and on all threads)
constexpr int NOT_MINE = 1 << 30;                   // rk2kk flag: the chain kept only its peak, which belongs to an earlier chain (chain.c:381-383)

// ---- kernel A: v[], child marks, chain ends -> unsorted keys (chain.c:106-111, 349-367) -------------------------------
__global__ __launch_bounds__(64) void epi_ends(EpiArgs A)
{
	__shared__ int s_cur[W], s_ptr[W];
	const int task = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int64_t base = A.d_off[task];
	const int n = (int)(A.d_off[task + 1] - base);
	if (A.fused && n <= FUSE_L) return;                                          // epi_fused has it
	const int lane = (int)threadIdx.x;
	const int32_t *__restrict__ f = A.d_f + base, *__restrict__ p = A.d_p + base;
	int32_t *v = A.v + base, *mark = A.own + base, *peak = A.ctop + base;
	if (lane == 0) A.seg_begin[task] = (uint32_t)base;
	for (int i = lane; i < n; i += 64) mark[i] = 0;
	__syncthreads();
	if (A.debug_phases == 1) { if (lane == 0) A.seg_end1[task] = (uint32_t)base; return; }
	for (int c0 = 0; c0 < n; c0 += W) {
		int cur[K], ptr[K], fi[K], pi[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = c0 + lane + 64 * k;
			pi[k] = ptr[k] = i < n ? p[i] : -1;
			fi[k] = cur[k] = i < n ? f[i] : INT_MIN;
		}
#pragma unroll
		for (int k = 0; k < K; ++k) {
			if (ptr[k] >= 0) mark[ptr[k]] = 1;                                     // chain.c:350
			if (ptr[k] >= 0 && ptr[k] < c0) { cur[k] = max(cur[k], v[ptr[k]]); ptr[k] = -1; }   // parent in an earlier chunk: final already
		}
		for (;;) {                                                               // parents inside the chunk: pointer jumping
			bool open = false;
#pragma unroll
			for (int k = 0; k < K; ++k) open |= ptr[k] >= c0;
			if (!__ballot(open)) break;
#pragma unroll
			for (int k = 0; k < K; ++k) { s_cur[lane + 64 * k] = cur[k]; s_ptr[lane + 64 * k] = ptr[k]; }
			__syncthreads();
#pragma unroll
			for (int k = 0; k < K; ++k)
				if (ptr[k] >= c0) { const int q = ptr[k] - c0; cur[k] = max(cur[k], s_cur[q]); ptr[k] = s_ptr[q]; }
			__syncthreads();
		}
		// peak[i] = the nearest anchor j on the path from i towards the root with f[j] >= v[j] (what the walk of chain.c:360-361
		// finds; v > f implies a parent, so the walk never falls off the root): same recurrence, same jumping
		int pk[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = c0 + lane + 64 * k;
			if (fi[k] >= cur[k]) { pk[k] = i; ptr[k] = -1; }
			else if (pi[k] < c0) { pk[k] = peak[pi[k]]; ptr[k] = -1; }
			else { pk[k] = -1; ptr[k] = pi[k]; }
		}
		for (;;) {
			bool open = false;
#pragma unroll
			for (int k = 0; k < K; ++k) open |= ptr[k] >= c0;
			if (!__ballot(open)) break;
#pragma unroll
			for (int k = 0; k < K; ++k) { s_cur[lane + 64 * k] = pk[k]; s_ptr[lane + 64 * k] = ptr[k]; }
			__syncthreads();
#pragma unroll
			for (int k = 0; k < K; ++k)
				if (ptr[k] >= c0) { const int q = ptr[k] - c0; pk[k] = s_cur[q]; ptr[k] = s_ptr[q]; }
			__syncthreads();
		}
#pragma unroll
		for (int k = 0; k < K; ++k) { const int i = c0 + lane + 64 * k; if (i < n) { v[i] = cur[k]; peak[i] = pk[k]; } }
		__syncthreads();
	}
	if (A.debug_phases == 2) { if (lane == 0) A.seg_end1[task] = (uint32_t)base; return; }
	uint64_t *__restrict__ keys = A.key0 + base;
	int cnt = 0;
	for (int c0 = 0; c0 < n; c0 += W) {
		bool is_end[K];
		int pk[K], fpk[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = c0 + lane + 64 * k;
			is_end[k] = i < n && mark[i] == 0 && v[i] >= A.min_sc;              // chain.c:352
			pk[k] = is_end[k] ? peak[i] : 0;
		}
#pragma unroll
		for (int k = 0; k < K; ++k) fpk[k] = is_end[k] ? f[pk[k]] : 0;
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = c0 + lane + 64 * k;
			if (i < n) mark[i] = NONE;                                         // from here on: the chain that takes the anchor (kernel B)
			const uint64_t m = __ballot(is_end[k]);
			if (is_end[k]) keys[cnt + lanes_before(m)] = (uint64_t)(uint32_t)fpk[k] << 32 | (uint32_t)pk[k];
			cnt += __popcll(m);
		}
	}
	if (lane == 0) A.seg_end1[task] = (uint32_t)(base + cnt);
}

// ---- kernel B: owners, depths, per-chain length / score / filter (chain.c:375-390) ----------------------------------
__global__ __launch_bounds__(64) void epi_claim(EpiArgs A)
{
	__shared__ int s_val[W], s_ptr[W];
	const int task = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int64_t base = A.d_off[task];
	const int n = (int)(A.d_off[task + 1] - base);
	if (A.fused && n <= FUSE_L) return;
	const int lane = (int)threadIdx.x;
	const int nu = (int)(A.seg_end1[task] - (uint32_t)base);
	const int32_t *__restrict__ f = A.d_f + base, *__restrict__ p = A.d_p + base;
	int32_t *own = A.own + base, *dep = A.v + base, *ctop = A.ctop + base, *rk2kk = A.rk2kk + base, *val0 = A.val0 + base;
	const uint64_t *us = A.key1 + base;
	uint64_t *u2 = A.u2 + base, *rkey = A.key0 + base;

	wave_sort64<true>(A.key0 + base, A.key1 + base, nullptr, nullptr, nu, lane, s_val);   // chain.c:368-372: best peak first

	for (int r = lane; r < nu; r += 64) wg_min(&own[(int32_t)us[r]], r);       // own[] is NONE on entry; a peak listed twice belongs to the first listing
	__syncthreads();
	if (A.debug_phases == 1) { if (lane == 0) { A.seg_end2[task] = (uint32_t)base; A.cnt_u[task] = 0; A.cnt_b[task] = 0; } return; }
	// owner(x) = min over the subtree of x: children have larger indices, so chunks go from the end to the front
	for (int c0 = n > 0 ? (n - 1) / W * W : -W; c0 >= 0; c0 -= W) {
		int pi[K], up[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = c0 + lane + 64 * k;
			pi[k] = i < n ? p[i] : -1;
			s_val[lane + 64 * k] = i < n ? wg_load(&own[i]) : NONE;   // sees the atomics of later chunks
			up[k] = pi[k] >= c0 ? pi[k] - c0 : -1;                                  // 2^t-th ancestor, while it is inside the chunk
		}
		__syncthreads();
		for (;;) {
			bool open = false;
#pragma unroll
			for (int k = 0; k < K; ++k) open |= up[k] >= 0;
			if (!__ballot(open)) break;
			int acc[K];
#pragma unroll
			for (int k = 0; k < K; ++k) { acc[k] = s_val[lane + 64 * k]; s_ptr[lane + 64 * k] = up[k]; }
			__syncthreads();
#pragma unroll
			for (int k = 0; k < K; ++k)
				if (up[k] >= 0) { if (acc[k] != NONE) atomicMin(&s_val[up[k]], acc[k]); up[k] = s_ptr[up[k]]; }
			__syncthreads();
		}
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = c0 + lane + 64 * k;
			if (i < n) {
				const int fin = s_val[lane + 64 * k];
				own[i] = fin;
				if (pi[k] >= 0 && pi[k] < c0 && fin != NONE) wg_min(&own[pi[k]], fin);
			}
		}
		__syncthreads();
	}
	if (A.debug_phases == 2) { if (lane == 0) { A.seg_end2[task] = (uint32_t)base; A.cnt_u[task] = 0; A.cnt_b[task] = 0; } return; }
	// depth inside the owner path (0 = top) and the top of every path
	for (int c0 = 0; c0 < n; c0 += W) {
		int o[K], pi[K], d[K], ptr[K];
		bool claimed[K], link[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = c0 + lane + 64 * k;
			o[k] = i < n ? wg_load(&own[i]) : NONE; pi[k] = i < n ? p[i] : -1;
		}
#pragma unroll
		for (int k = 0; k < K; ++k) {
			claimed[k] = o[k] != NONE;
			link[k] = claimed[k] && pi[k] >= 0 && wg_load(&own[pi[k]]) == o[k];
		}
#pragma unroll
		for (int k = 0; k < K; ++k) {
			d[k] = 0; ptr[k] = -1;
			if (link[k]) { if (pi[k] < c0) d[k] = dep[pi[k]] + 1; else { d[k] = 1; ptr[k] = pi[k]; } }
		}
		for (;;) {
			bool open = false;
#pragma unroll
			for (int k = 0; k < K; ++k) open |= ptr[k] >= c0;
			if (!__ballot(open)) break;
#pragma unroll
			for (int k = 0; k < K; ++k) { s_val[lane + 64 * k] = d[k]; s_ptr[lane + 64 * k] = ptr[k]; }
			__syncthreads();
#pragma unroll
			for (int k = 0; k < K; ++k)
				if (ptr[k] >= c0) { const int q = ptr[k] - c0; d[k] += s_val[q]; ptr[k] = s_ptr[q]; }
			__syncthreads();
		}
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = c0 + lane + 64 * k;
			if (i < n) dep[i] = d[k];
			if (claimed[k] && !link[k]) ctop[o[k]] = i;
		}
		__syncthreads();
	}
	if (A.debug_phases == 3) { if (lane == 0) { A.seg_end2[task] = (uint32_t)base; A.cnt_u[task] = 0; A.cnt_b[task] = 0; } return; }
	// one lane per chain, in rank order (chain.c:377-389)
	int kept = 0, n_b = 0;
	for (int r0 = 0; r0 < nu; r0 += 64) {
		const int r = r0 + lane;
		const bool valid = r < nu;
		bool keep = false;
		int len = 0, sc = 0, top = 0;
		if (valid) {
			const uint64_t key = us[r];
			const int j = (int32_t)key, peak = (int32_t)(key >> 32);
			const bool mine = wg_load(&own[j]) == r;
			len = mine ? dep[j] + 1 : 1;
			top = mine ? ctop[r] : j;
			const int stop = p[top];
			sc = stop < 0 ? peak : peak - f[stop];
			keep = (stop < 0 || sc >= A.min_sc) && len >= A.min_cnt;
		}
		const uint64_t m = __ballot(keep);
		const int kk = kept + lanes_before(m);
		if (keep) {
			u2[kk] = (uint64_t)(uint32_t)sc << 32 | (uint32_t)len;
			rkey[kk] = A.d_a[base + top].x;
			val0[kk] = kk;
		}
		if (valid) rk2kk[r] = keep ? kk : -1;
		kept += __popcll(m);
		n_b += wave_sum(keep ? len : 0);
	}
	if (lane == 0) {
		A.seg_end2[task] = (uint32_t)(base + kept);
		A.cnt_u[task] = kept;
		A.cnt_b[task] = n_b;
	}
	__syncthreads();
	// chain.c:406-411: chains by the x of their first anchor (stable here; kernel T replays the reference's sort where that matters)
	wave_sort64<false>(rkey, A.rkey1 + base, val0, A.val1 + base, kept, lane, s_val);
}


// ---- the passes of kernels A and B for a task that fits the LDS -------------------------------------------------------------------------
// Kernels A and B move every per-anchor quantity (marks, v, peaks, owners, depths) through global scratch arrays several times: 156 bytes
// of HBM traffic per anchor against 8 bytes of input (profiles/r1_seed_hits.md).  For a task of at most CAP anchors the same passes run
// with those arrays in LDS (indices fit 16 bits): parent 2 B, v / owner 4 B, peak / depth 2 B, child mark 1 bit per anchor.  Global traffic
// is then f and p once, the chain records (per chain, not per anchor) and one word per anchor for kernel C: (chain << 16 | depth), or -1.

template <int CAP>
__global__ __launch_bounds__(FNT, CAP <= 5120 ? 6 : 4)   // waves per SIMD: three (two) workgroups per CU, as many as the LDS takes
void epi_fused(EpiArgs A, int n_above)
{
	static_assert(CAP % 64 == 0 && CAP < NONE16, "indices and NONE16 in 16 bits");
	// one 8-byte cell per anchor, read and written whole (so that another wave sees a consistent pair), used three times:
	//   v / peaks:  low = running maximum of f over the path walked so far, high = position of that maximum << 16 | next ancestor to visit
	//   owners:     low = owner (rank of the chain that takes the anchor),  high = jump pointer of the doubling rounds
	//   depths:     low = owner,                                            high = links counted so far << 16 | next ancestor to visit
	// and in between as the two buffers of the sorts
	__shared__ uint64_t s_a[CAP];
	__shared__ uint16_t s_p[CAP];                   // parent (NONE16: none)
	__shared__ uint32_t s_mark[CAP / 32];           // has a child (chain.c:350)
	__shared__ int s_cnt[256];
	__shared__ int s_n;
	int *const s_w = (int *)s_a;                    // word 2i = low, 2i + 1 = high
	const int task = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int64_t base = A.d_off[task];
	const int n = (int)(A.d_off[task + 1] - base);
	if (n > CAP || n <= n_above) return;            // another class, or kernels A / B
	const int tid = (int)threadIdx.x, lane = tid & 63;
	const bool wave0 = tid < 64;
	const int32_t *__restrict__ f = A.d_f + base, *__restrict__ p = A.d_p + base;
	if (tid == 0) { A.seg_begin[task] = (uint32_t)base; s_n = 0; }
	if (A.debug_phases && tid == 0) { A.seg_end1[task] = (uint32_t)base; A.seg_end2[task] = (uint32_t)base; A.cnt_u[task] = 0; A.cnt_b[task] = 0; }   // development aid: cut after phase N
	for (int w = tid; w < (n + 31) / 32; w += FNT) s_mark[w] = 0;
	__syncthreads();
	constexpr int KE = CAP / FNT;                   // anchors per thread: anchor tid + FNT k for k < KE.  Every loop over them is written so that
	                                                // the KE memory accesses of a step are independent and in flight together
	uint64_t e[KE];                                 // the thread's own cells (only their owner writes them: they stay in registers over the rounds)
	{
		int pv[KE], fv[KE];
#pragma unroll
		for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; pv[k] = i < n ? p[i] : -1; fv[k] = i < n ? f[i] : 0; }
#pragma unroll
		for (int k = 0; k < KE; ++k) {
			const int i = tid + FNT * k, q = pv[k] < 0 ? NONE16 : pv[k];
			e[k] = (uint64_t)(uint32_t)(i << 16 | q) << 32 | (uint32_t)fv[k];
			if (i < n) {
				s_p[i] = (uint16_t)q;
				if (pv[k] >= 0) atomicOr(&s_mark[pv[k] >> 5], 1u << (pv[k] & 31));
				s_a[i] = e[k];
			}
		}
	}
	__syncthreads();
	if (A.debug_phases == 1) return;
	// ---- v[] and peaks by pointer jumping over the whole task, every thread at its own pace.  peak[i], the first anchor on the way up with
	// f >= v (chain.c:360-361), is the NEAREST ancestor-or-self whose f equals the maximum over the path (everything nearer has f < v[i] = its
	// own v; there f = v): a running (maximum, nearest position of it), which composes over path segments.  A cell always describes the path
	// from its anchor up to (excluding) `next`; cells are read and written in one piece, so whatever state another cell is in, appending it is valid.
	for (;;) {
		uint64_t q[KE];
#pragma unroll
		for (int k = 0; k < KE; ++k) {
			const int nxt = (int)(e[k] >> 32) & 0xffff;
			q[k] = nxt != NONE16 ? __hip_atomic_load(&s_a[nxt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;   // another thread's cell, in one piece
		}
		bool again = false;
#pragma unroll
		for (int k = 0; k < KE; ++k) {
			if (((int)(e[k] >> 32) & 0xffff) == NONE16) continue;
			int v = (int32_t)e[k], pk = (int)(e[k] >> 48);
			if ((int32_t)q[k] > v) { v = (int32_t)q[k]; pk = (int)(q[k] >> 48); }
			const int nn = (int)(q[k] >> 32) & 0xffff;
			e[k] = (uint64_t)(uint32_t)(pk << 16 | nn) << 32 | (uint32_t)v;
			__hip_atomic_store(&s_a[tid + FNT * k], e[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			again |= nn != NONE16;
		}
		if (!again) break;
	}
	__syncthreads();
	if (A.debug_phases == 2) return;
	// ---- chain ends -> keys f[peak] << 32 | peak with f[peak] = v (any order: they are sorted next), collected in the cells' place
	bool wide = false;                                // a score that does not fit 19 bits beside a 13-bit peak, or is not positive
	{
		uint64_t key[KE]; bool is_end[KE];
#pragma unroll
		for (int k = 0; k < KE; ++k) {
			const int i = tid + FNT * k;
			is_end[k] = i < n && !((s_mark[i >> 5] >> (i & 31)) & 1) && (int32_t)e[k] >= A.min_sc;   // chain.c:352 (the cell is final: v, peak)
			key[k] = is_end[k] ? (e[k] << 32 | e[k] >> 48) : 0;
			wide |= is_end[k] && ((int32_t)e[k] < 1 || (int32_t)e[k] >= (1 << 19));
		}
		wide = __syncthreads_or(wide);
#pragma unroll
		for (int k = 0; k < KE; ++k) {
			const uint64_t m = __ballot(is_end[k]);
			int at = 0;
			if (m != 0 && lane == 0) at = atomicAdd(&s_n, __popcll(m));
			at = __shfl(at, 0);
			if (is_end[k]) s_a[at + lanes_before(m)] = key[k];
		}
	}
	__syncthreads();
	const int nu = s_n;
	if (A.debug_phases == 3) return;
	int32_t *ctop = A.ctop + base, *rk2kk = A.rk2kk + base, *val0 = A.val0 + base;
	uint64_t *us = A.key1 + base;
	uint64_t *u2 = A.u2 + base, *rkey = A.key0 + base;
	if (tid == 0) A.seg_end1[task] = (uint32_t)(base + nu);
	const bool by_key = !wide && nu <= 2 * FNT && CAP <= 8192 && A.debug_phases == 0;
	if (tid == 0) A.seg_begin[task] = by_key ? 1u : 0u;   // tells kernel C which of the two forms of its input this task has
	if (by_key) {
		// ---- Owners WITHOUT sorting the chain ends first.  The owner of an anchor is the BEST peak in its subtree; "best" is the order of the
		// keys f[peak] << 32 | peak, and with scores below 2^19 and tasks of at most 8 192 anchors the key fits one word, f[peak] << 13 | peak, that
		// an LDS atomic maximum can carry.  Only the chains that survive the filter (a sixth of the chain ends on the headline stream) need their
		// rank, and a handful of keys is ranked by counting.  A chain is then known by its peak: `ctop`, `rk2kk` are indexed by peaks here.
		int32_t *end2kk = A.v + base;                  // per chain end: its number among the kept chains (| NOT_MINE), or -1
		uint16_t *const s_map = (uint16_t *)(s_a + CAP / 2 + 512);   // peak -> number of its chain among the kept ones (behind the kept keys and their flags)
		static_assert(CAP / 2 + 512 + CAP / 4 <= CAP && 2 * FNT <= CAP / 2, "kept keys, their flags and the peak -> chain map share the cells' space");
		uint64_t mk[2];
#pragma unroll
		for (int m = 0; m < 2; ++m) { const int en = tid + FNT * m; mk[m] = en < nu ? s_a[en] : 0; if (en < nu) us[en] = mk[m]; }
		for (int w = tid; w < (n + 31) / 32; w += FNT) s_mark[w] = 0;   // from here on: this peak has been listed
		__syncthreads();
		for (int i = tid; i < n; i += FNT) s_a[i] = (uint64_t)(uint32_t)(s_p[i] == NONE16 ? -1 : (int)s_p[i]) << 32;   // owner 0: none (keys are >= 8 192)
		__syncthreads();
#pragma unroll
		for (int m = 0; m < 2; ++m)
			if (tid + FNT * m < nu) atomicMax((unsigned *)&s_w[2 * (int32_t)(mk[m] & 0xffff)], (uint32_t)(mk[m] >> 32) << 13 | (uint32_t)(mk[m] & 0x1fff));
		__syncthreads();
		{
			int up[KE];
#pragma unroll
			for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; up[k] = i < n ? s_w[2 * i + 1] : -1; }
			for (;;) {
				bool open = false;
#pragma unroll
				for (int k = 0; k < KE; ++k) open |= up[k] >= 0;
				if (!__syncthreads_or(open)) break;
#pragma unroll
				for (int k = 0; k < KE; ++k)
					if (up[k] >= 0) {
						const int acc = __hip_atomic_load(&s_w[2 * (tid + FNT * k)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						if (acc != 0) atomicMax((unsigned *)&s_w[2 * up[k]], (unsigned)acc);
						up[k] = s_w[2 * up[k] + 1];
					}
				__syncthreads();
#pragma unroll
				for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; if (i < n) s_w[2 * i + 1] = up[k]; }
			}
		}
		__syncthreads();
		// depths and tops, as below
		int dn[KE], ow[KE];
		{
			int pi[KE], op[KE];
#pragma unroll
			for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; ow[k] = i < n ? s_w[2 * i] : 0; pi[k] = i < n ? (int)s_p[i] : NONE16; }
#pragma unroll
			for (int k = 0; k < KE; ++k) op[k] = (ow[k] != 0 && pi[k] != NONE16) ? s_w[2 * pi[k]] : 0;
#pragma unroll
			for (int k = 0; k < KE; ++k) {
				const bool claimed = ow[k] != 0, link = claimed && pi[k] != NONE16 && op[k] == ow[k];
				if (claimed && !link) ctop[ow[k] & 0x1fff] = tid + FNT * k;
				dn[k] = link ? (1 << 16 | pi[k]) : NONE16;
			}
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; if (i < n) s_w[2 * i + 1] = dn[k]; }
		__syncthreads();
		for (;;) {
			int q[KE];
#pragma unroll
			for (int k = 0; k < KE; ++k) {
				const int nxt = dn[k] & 0xffff;
				q[k] = nxt != NONE16 ? __hip_atomic_load(&s_w[2 * nxt + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : NONE16;
			}
			bool again = false;
#pragma unroll
			for (int k = 0; k < KE; ++k) {
				if ((dn[k] & 0xffff) == NONE16) continue;
				const int nn = q[k] & 0xffff;
				dn[k] = (int)(((unsigned)dn[k] >> 16) + ((unsigned)q[k] >> 16)) << 16 | nn;
				__hip_atomic_store(&s_w[2 * (tid + FNT * k) + 1], dn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				again |= nn != NONE16;
			}
			if (!again) break;
		}
		__syncthreads();
		// ---- one thread per chain end (chain.c:377-389)
		bool keep[2], mine[2]; int len[2], sc[2], top[2];
#pragma unroll
		for (int m = 0; m < 2; ++m) {
			keep[m] = mine[m] = false; len[m] = sc[m] = top[m] = 0;
			if (tid + FNT * m < nu) {
				const int j = (int)(mk[m] & 0xffff), peak = (int32_t)(mk[m] >> 32);
				const bool first = !((atomicOr(&s_mark[j >> 5], 1u << (j & 31)) >> (j & 31)) & 1);   // a peak listed twice belongs to one listing
				mine[m] = first && s_w[2 * j] == (int)((uint32_t)peak << 13 | (uint32_t)j);
				len[m] = mine[m] ? (int)((unsigned)s_w[2 * j + 1] >> 16) + 1 : 1;
			}
		}
		__syncthreads();                               // the tops are in memory (ctop), the cells have been read
#pragma unroll
		for (int m = 0; m < 2; ++m)
			if (tid + FNT * m < nu) {
				const int j = (int)(mk[m] & 0xffff), peak = (int32_t)(mk[m] >> 32);
				top[m] = mine[m] ? ctop[j] : j;
				const int stop = s_p[top[m]] == NONE16 ? -1 : (int)s_p[top[m]];
				sc[m] = stop < 0 ? peak : peak - f[stop];
				keep[m] = (stop < 0 || sc[m] >= A.min_sc) && len[m] >= A.min_cnt;
			}
		// the kept chains in the order of their keys (best first; of two listings of one peak the one that took the anchors first)
		if (tid == 0) s_n = 0;
		__syncthreads();
		int at[2];
#pragma unroll
		for (int m = 0; m < 2; ++m) {
			at[m] = -1;
			if (keep[m]) { at[m] = atomicAdd(&s_n, 1); s_a[at[m]] = mk[m]; s_w[2 * (CAP / 2) + at[m]] = mine[m]; }
		}
		__syncthreads();
		const int nk = s_n;
		int n_b = 0;
#pragma unroll
		for (int m = 0; m < 2; ++m) {
			const int en = tid + FNT * m;
			if (en >= nu) continue;
			const int j = (int)(mk[m] & 0xffff);
			int kk = -1;
			if (keep[m]) {
				kk = 0;
				for (int t = 0; t < nk; ++t) {
					const uint64_t kt = s_a[t];
					const int mt = s_w[2 * (CAP / 2) + t];
					kk += (kt > mk[m]) | ((kt == mk[m]) & ((mt > (int)mine[m]) | ((mt == (int)mine[m]) & (t < at[m]))));
				}
				u2[kk] = (uint64_t)(uint32_t)sc[m] << 32 | (uint32_t)len[m];
				rkey[kk] = A.d_a[base + top[m]].x;
				n_b += len[m];
			}
			end2kk[en] = kk < 0 ? -1 : (mine[m] ? kk : kk | NOT_MINE);
			if (mine[m]) s_map[j] = (uint16_t)kk;        // anchors of a dropped chain find 0xffff here
		}
		n_b = wave_sum(n_b);
		__syncthreads();                               // the keys have been read, the chain numbers are in place
		// what kernel C needs per anchor: the kept chain that takes it and its position inside that chain
		{
			int32_t *cd = A.own + base;
#pragma unroll
			for (int k = 0; k < KE; ++k) {
				const int i = tid + FNT * k;
				const int kk = ow[k] != 0 ? (int)s_map[ow[k] & 0x1fff] : 0xffff;
				if (i < n) cd[i] = kk != 0xffff ? (kk << 16 | (int)((unsigned)dn[k] >> 16)) : -1;
			}
		}
		__syncthreads();                               // ... and read: the first-x keys may take their place
		if (tid == 0) s_n = 0;
		__syncthreads();
		if (lane == 0 && n_b) atomicAdd(&s_n, n_b);
		__syncthreads();
		if (tid == 0) { A.seg_end2[task] = (uint32_t)(base + nk); A.cnt_u[task] = nk; A.cnt_b[task] = s_n; }
		// chain.c:406-411: chains by the x of their first anchor, stable (kernel T replays the reference's sort where that matters)
		if (nk <= RANK_MAX) {
			for (int i = tid; i < nk; i += FNT) s_a[i] = rkey[i];
			__syncthreads();
			for (int i = tid; i < nk; i += FNT) {
				const uint64_t key = s_a[i];
				int before = 0;
				for (int j = 0; j < nk; ++j) { const uint64_t kj = s_a[j]; before += (kj < key) | ((kj == key) & (j < i)); }
				A.rkey1[base + before] = key; A.val1[base + before] = i;
			}
		} else {
			for (int i = tid; i < nk; i += FNT) val0[i] = i;
			__syncthreads();
			if (wave0) wave_sort64<false, true>(rkey, A.rkey1 + base, val0, A.val1 + base, nk, lane, s_cnt);
		}
		return;
	}
	// ---- the general form: the chain ends are sorted first and an owner is the RANK of its chain
	// chain.c:368-372: best peak first.  Up to RANK_MAX keys: every thread counts the keys that come before its own (all threads read the same
	// key at a time: a broadcast) -- no passes, no barriers; more keys: the radix sort of the first wave, in LDS while both buffers fit
	if (nu <= RANK_MAX) {
		for (int r = tid; r < nu; r += FNT) {
			const uint64_t key = s_a[r];
			int before = 0;
			for (int j = 0; j < nu; ++j) { const uint64_t kj = s_a[j]; before += (kj > key) | ((kj == key) & (j < r)); }
			s_a[CAP / 2 + before] = key;
		}
	} else if (wave0) {
		if (nu <= CAP / 2) wave_sort64<true, true>(s_a, s_a + CAP / 2, nullptr, nullptr, nu, lane, s_cnt);
		else {
			for (int r = lane; r < nu; r += 64) rkey[r] = s_a[r];
			wave_sort64<true, true>(rkey, us, nullptr, nullptr, nu, lane, s_cnt);
		}
	}
	__syncthreads();
	if (nu <= CAP / 2) for (int r = tid; r < nu; r += FNT) us[r] = s_a[CAP / 2 + r];
	__syncthreads();
	if (A.debug_phases == 4) return;
	// ---- owners: owner(x) = min rank over the peaks in the subtree of x, pushed towards the roots in doubling rounds
	for (int i = tid; i < n; i += FNT) s_a[i] = (uint64_t)(uint32_t)(s_p[i] == NONE16 ? -1 : (int)s_p[i]) << 32 | (uint32_t)NONE;
	__syncthreads();
	for (int r = tid; r < nu; r += FNT) atomicMin(&s_w[2 * (int32_t)us[r]], r);   // a peak listed twice belongs to the first listing
	__syncthreads();
	{
		// round t: every anchor pushes what has reached it to its 2^t-th ancestor, then its jump pointer doubles.  A value that arrives early
		// (a push of the same round that lands before the anchor reads its own cell) is still the rank of a peak below it: harmless
		int up[KE];
#pragma unroll
		for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; up[k] = i < n ? s_w[2 * i + 1] : -1; }
		for (;;) {
			bool open = false;
#pragma unroll
			for (int k = 0; k < KE; ++k) open |= up[k] >= 0;
			if (!__syncthreads_or(open)) break;         // also: the jump pointers written at the end of the round before are in place
#pragma unroll
			for (int k = 0; k < KE; ++k)
				if (up[k] >= 0) {
					const int acc = __hip_atomic_load(&s_w[2 * (tid + FNT * k)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					if (acc != NONE) atomicMin(&s_w[2 * up[k]], acc);
					up[k] = s_w[2 * up[k] + 1];
				}
			__syncthreads();                              // all jump pointers read before any is replaced
#pragma unroll
			for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; if (i < n) s_w[2 * i + 1] = up[k]; }
		}
	}
	__syncthreads();
	if (A.debug_phases == 5) return;
	// ---- depth inside the owner path (links to the path's top) and the top of every path: pulled, like v
	int dn[KE];                                       // links counted so far << 16 | next ancestor to visit
	{
		int o[KE], pi[KE], op[KE];
#pragma unroll
		for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; o[k] = i < n ? s_w[2 * i] : NONE; pi[k] = i < n ? (int)s_p[i] : NONE16; }
#pragma unroll
		for (int k = 0; k < KE; ++k) op[k] = (o[k] != NONE && pi[k] != NONE16) ? s_w[2 * pi[k]] : NONE;
#pragma unroll
		for (int k = 0; k < KE; ++k) {
			const int i = tid + FNT * k;
			const bool claimed = o[k] != NONE, link = claimed && pi[k] != NONE16 && op[k] == o[k];
			if (claimed && !link) ctop[o[k]] = i;
			dn[k] = link ? (1 << 16 | pi[k]) : NONE16;
		}
	}
	__syncthreads();                                  // every owner has been read: the high words can be overwritten
#pragma unroll
	for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; if (i < n) s_w[2 * i + 1] = dn[k]; }
	__syncthreads();
	for (;;) {
		int q[KE];
#pragma unroll
		for (int k = 0; k < KE; ++k) {
			const int nxt = dn[k] & 0xffff;
			q[k] = nxt != NONE16 ? __hip_atomic_load(&s_w[2 * nxt + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : NONE16;
		}
		bool again = false;
#pragma unroll
		for (int k = 0; k < KE; ++k) {
			if ((dn[k] & 0xffff) == NONE16) continue;
			const int nn = q[k] & 0xffff;
			dn[k] = (int)(((unsigned)dn[k] >> 16) + ((unsigned)q[k] >> 16)) << 16 | nn;
			__hip_atomic_store(&s_w[2 * (tid + FNT * k) + 1], dn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			again |= nn != NONE16;
		}
		if (!again) break;
	}
	__syncthreads();
	if (A.debug_phases == 6) return;
	// ---- one lane per chain, in rank order (chain.c:377-389): the first wave alone
	if (wave0) {
		int kept = 0, n_b = 0;
		for (int r0 = 0; r0 < nu; r0 += 64) {
			const int r = r0 + lane;
			const bool valid = r < nu;
			bool keep = false, mine = false;
			int len = 0, sc = 0, top = 0;
			if (valid) {
				const uint64_t key = us[r];
				const int j = (int32_t)key, peak = (int32_t)(key >> 32);
				mine = s_w[2 * j] == r;
				len = mine ? (int)((unsigned)s_w[2 * j + 1] >> 16) + 1 : 1;
				top = mine ? ctop[r] : j;
				const int stop = s_p[top] == NONE16 ? -1 : (int)s_p[top];
				sc = stop < 0 ? peak : peak - f[stop];
				keep = (stop < 0 || sc >= A.min_sc) && len >= A.min_cnt;
			}
			const uint64_t m = __ballot(keep);
			const int kk = kept + lanes_before(m);
			if (keep) {
				u2[kk] = (uint64_t)(uint32_t)sc << 32 | (uint32_t)len;
				rkey[kk] = A.d_a[base + top].x;
				val0[kk] = kk;
			}
			if (valid) rk2kk[r] = keep ? (mine ? kk : kk | NOT_MINE) : -1;
			kept += __popcll(m);
			n_b += wave_sum(keep ? len : 0);
		}
		if (lane == 0) {
			A.seg_end2[task] = (uint32_t)(base + kept);
			A.cnt_u[task] = kept;
			A.cnt_b[task] = n_b;
			s_n = kept;
		}
	}
	__syncthreads();
	if (A.debug_phases == 7) return;
	// what kernel C needs per anchor: the kept chain that takes it and its position inside that chain
	int32_t *cd = A.own + base;
	{
		int o[KE], kk[KE];
#pragma unroll
		for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; o[k] = i < n ? s_w[2 * i] : NONE; }
#pragma unroll
		for (int k = 0; k < KE; ++k) kk[k] = o[k] != NONE ? rk2kk[o[k]] : -1;   // the owner of an anchor is always `mine` for it: no flag
#pragma unroll
		for (int k = 0; k < KE; ++k) { const int i = tid + FNT * k; if (i < n) cd[i] = kk[k] >= 0 ? (kk[k] << 16 | (int)((unsigned)dn[k] >> 16)) : -1; }
	}
	// chain.c:406-411: chains by the x of their first anchor (stable here; kernel T replays the reference's sort where that matters)
	__syncthreads();                                  // the cells are free again
	const int nk = s_n;
	if (nk <= RANK_MAX) {                             // stable: equal first x keep rank order
		for (int i = tid; i < nk; i += FNT) s_a[i] = rkey[i];
		__syncthreads();
		for (int i = tid; i < nk; i += FNT) {
			const uint64_t key = s_a[i];
			int before = 0;
			for (int j = 0; j < nk; ++j) { const uint64_t kj = s_a[j]; before += (kj < key) | ((kj == key) & (j < i)); }
			A.rkey1[base + before] = key; A.val1[base + before] = i;
		}
	} else if (wave0) {
		if (nk <= CAP / 3) {                            // keys and values of both buffers fit the cells' space
			uint64_t *k0 = s_a, *k1 = s_a + CAP / 3;
			int32_t *v0 = (int32_t *)(s_a + 2 * (CAP / 3)), *v1 = v0 + CAP / 3;
			for (int i = lane; i < nk; i += 64) { k0[i] = rkey[i]; v0[i] = i; }
			wave_sort64<false, true>(k0, k1, v0, v1, nk, lane, s_cnt);
			for (int i = lane; i < nk; i += 64) { A.rkey1[base + i] = k1[i]; A.val1[base + i] = v1[i]; }
		} else wave_sort64<false, true>(rkey, A.rkey1 + base, val0, A.val1 + base, nk, lane, s_cnt);
	}
}

// ---- exclusive scans of the per-task chain / anchor counts -> compact output offsets ---------------------------------------
__global__ __launch_bounds__(1024) void epi_offsets(EpiArgs A)
{
	// one workgroup per 1 024 tasks: it first adds up the counts of all the tasks before its own (coalesced, the loads of a step in flight together; the last
	// workgroup reads 64 K counts), then scans its own inside the waves by shuffles and across them through LDS.  (One workgroup walking over all the tasks
	// with a running carry took 0.19 ms for 65 536 tasks, whatever it did per step: 64 dependent steps.)
	__shared__ int64_t s_u[16], s_b[16];
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int64_t nt = A.n_tasks, t0 = (int64_t)blockIdx.x * 1024;
	int64_t pre_u = 0, pre_b = 0;
	for (int64_t t = tid; t < t0; t += 1024) { pre_u += A.cnt_u[t]; pre_b += A.cnt_b[t]; }
	for (int o = 32; o > 0; o >>= 1) { pre_u += __shfl_xor(pre_u, o); pre_b += __shfl_xor(pre_b, o); }
	if (lane == 0) { s_u[wave] = pre_u; s_b[wave] = pre_b; }
	__syncthreads();
	int64_t carry_u = 0, carry_b = 0;
	for (int w = 0; w < 16; ++w) { carry_u += s_u[w]; carry_b += s_b[w]; }
	__syncthreads();
	const int64_t t = t0 + tid;
	const int64_t cu = t < nt ? A.cnt_u[t] : 0, cb = t < nt ? A.cnt_b[t] : 0;
	int64_t iu = cu, ib = cb;                                                    // inclusive scan inside the wave
	for (int o = 1; o < 64; o <<= 1) {
		const int64_t yu = __shfl_up(iu, o), yb = __shfl_up(ib, o);
		if (lane >= o) { iu += yu; ib += yb; }
	}
	if (lane == 63) { s_u[wave] = iu; s_b[wave] = ib; }
	__syncthreads();
	int64_t wu = 0, wb = 0, tot_u = 0, tot_b = 0;
	for (int w = 0; w < 16; ++w) { const int64_t xu = s_u[w], xb = s_b[w]; if (w < wave) { wu += xu; wb += xb; } tot_u += xu; tot_b += xb; }
	if (t < nt) { A.u_off[t] = carry_u + wu + iu - cu; A.b_off[t] = carry_b + wb + ib - cb; }
	if (tid == 0 && t0 + 1024 >= nt) { A.u_off[nt] = carry_u + tot_u; A.b_off[nt] = carry_b + tot_b; }
}

// ---- kernel T: tasks with more than 64 chains and equal first-x values: the order radix_sort_128x leaves (chain.c:411) ----
// radix_replay.h on the chain records (x of the first anchor, chain) in rank order; only buckets that hold equal keys are replayed, then a
// stable sort of the replayed arrangement gives the reference's array.  The index array lives in LDS for up to TS_MAX chains.
constexpr int TS_MAX = 4096;

__global__ __launch_bounds__(64) void epi_tiesort(EpiArgs A)
{
	__shared__ uint16_t s_id[TS_MAX];
	__shared__ uint8_t s_dg[TS_MAX + 8];                                         // (the walk of radix_replay.h looks one byte beyond the digit it takes)
	__shared__ __attribute__((aligned(8))) int s_cur[576];
	__shared__ int s_lo[257], s_sp;
	const int task = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int64_t base = A.d_off[task];
	const int lane = (int)threadIdx.x;
	const int nk = (int)(A.seg_end2[task] - (uint32_t)base);
	if (nk <= 64) return;                                                        // insertion sort only: stable (ksort.h:141-143)
	uint64_t *sx = A.rkey1 + base, *rank_x = A.key0 + base;                      // first-x keys: sorted / in rank order (chain k at index k)
	int32_t *ord = A.val1 + base, *ids = A.val0 + base, *tiecnt = A.dest + base, *stack = A.ctop + base;
	int32_t *moved = (int32_t *)A.sort_tmp + base;                               // source position per position of a replayed pass (radix_replay.h)
	// tiecnt[i] = equal neighbours before position i of the sorted keys
	int run = 0;
	for (int i0 = 0; i0 < nk; i0 += 64) {
		const int i = i0 + lane;
		const int flag = (i + 1 < nk && sx[i] == sx[i + 1]) ? 1 : 0;
		const int incl = wave_incl_scan(flag, lane);
		if (i < nk) tiecnt[i] = run + incl - flag;
		run += __shfl(incl, 63);
	}
	if (run == 0) return;                                                        // distinct keys: the order is unique
	if (A.debug_phases == 21) return;
	__syncthreads();
	for (int i = lane; i < nk; i += 64) rank_x[ord[i]] = sx[i];                  // back to rank order, as chain.c:407-410 fills w[]
	__syncthreads();
	if (nk <= TS_MAX) {
		replay_passes<uint16_t, false, true>(rank_x, 1, sx, 1, tiecnt, nk, s_id, s_dg, stack, moved, nullptr, nullptr, lane, s_cur, s_lo, &s_sp);
		for (int i = lane; i < nk; i += 64) ids[i] = (int32_t)s_id[i];
	} else {                                                                     // does not fit the LDS: same replay through global memory
		uint8_t *g_dg = (uint8_t *)(stack + 2 * (nk / 64 + 2));
		replay_passes<uint32_t, false, false>(rank_x, 1, sx, 1, tiecnt, nk, (uint32_t *)ids, g_dg, stack, moved, nullptr, nullptr, lane, s_cur, s_lo, &s_sp);
	}
	if (A.debug_phases == 22) return;
	__syncthreads();
	for (int i = lane; i < nk; i += 64) sx[i] = rank_x[ids[i]];                  // keys of the replayed arrangement
	__syncthreads();
	wave_sort64<false>(sx, rank_x, ids, ord, nk, lane, s_cur);                   // stable: = the insertion sorts of ksort.h:144, sorted order elsewhere
}

// ---- kernel C: final chain order, u[] and b[] (chain.c:397-420) -----------------------------------------------------
__global__ __launch_bounds__(64) void epi_emit(EpiArgs A)
{
	const int task = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int64_t base = A.d_off[task];
	const int n = (int)(A.d_off[task + 1] - base);
	const int lane = (int)threadIdx.x;
	if (A.fused && n <= FUSE_L) return;                                          // epi_emit_cd has it
	const int nu = (int)(A.seg_end1[task] - (uint32_t)base), nk = (int)(A.seg_end2[task] - (uint32_t)base);
	if (nk == 0) return;
	const int32_t *own = A.own + base, *dep = A.v + base, *rk2kk = A.rk2kk + base;
	int32_t *dest = A.dest + base;
	const int32_t *ord = A.val1 + base;
	const uint64_t *u2 = A.u2 + base, *us = A.key1 + base;
	uint64_t *u_out = A.u_out + A.u_off[task];
	ulonglong2 *b_out = A.b_out + A.b_off[task];
	const int n_b = (int)(A.b_off[task + 1] - A.b_off[task]);

	int run = 0;
	for (int i0 = 0; i0 < nk; i0 += 64) {
		const int i = i0 + lane;
		const bool valid = i < nk;
		const int kk = valid ? ord[i] : 0;
		const uint64_t uu = valid ? u2[kk] : 0;
		const int len = valid ? (int32_t)uu : 0;
		const int incl = wave_incl_scan(len, lane);
		if (valid) { dest[kk] = run + incl - len; u_out[i] = uu; }
		run += __shfl(incl, 63);
	}
	__syncthreads();
	if (A.debug_phases == 11) return;
	for (int i0 = 0; i0 < n; i0 += W) {
		int at[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = i0 + lane + 64 * k;
			const int o = i < n ? own[i] : NONE;
			const int kk = o != NONE ? rk2kk[o] : -1;
			at[k] = kk >= 0 ? dest[kk] + dep[i] : -1;                            // ascending along the chain (chain.c:399-400)
		}
#pragma unroll
		for (int k = 0; k < K; ++k)
			if (at[k] >= 0 && at[k] < n_b) b_out[at[k]] = A.d_a[base + i0 + lane + 64 * k];
	}
	for (int r = lane; r < nu; r += 64) {                                        // chains that kept only their (already taken) peak
		const int kk = rk2kk[r];
		if (kk < 0) continue;
		const int j = (int32_t)us[r];
		if (own[j] != r && dest[kk] >= 0 && dest[kk] < n_b) b_out[dest[kk]] = A.d_a[base + j];
	}
}


// ---- kernel C for the tasks of epi_fused: b[] from one word per anchor ------------------------------------------------------------------
__global__ __launch_bounds__(64) void epi_emit_cd(EpiArgs A)
{
	const int task = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int64_t base = A.d_off[task];
	const int n = (int)(A.d_off[task + 1] - base);
	if (n > FUSE_L) return;                                                      // kernel C proper
	const int lane = (int)threadIdx.x;
	const int nu = (int)(A.seg_end1[task] - (uint32_t)base), nk = (int)(A.seg_end2[task] - (uint32_t)base);
	if (nk == 0) return;
	const int32_t *cd = A.own + base, *rk2kk = A.rk2kk + base, *end2kk = A.v + base;
	const bool by_key = A.seg_begin[task] == 1;                                  // the chain ends are listed unsorted, with their chain numbers in end2kk
	int32_t *dest = A.dest + base;
	const int32_t *ord = A.val1 + base;
	const uint64_t *u2 = A.u2 + base, *us = A.key1 + base;
	uint64_t *u_out = A.u_out + A.u_off[task];
	ulonglong2 *b_out = A.b_out + A.b_off[task];
	const int n_b = (int)(A.b_off[task + 1] - A.b_off[task]);
	int run = 0;
	for (int i0 = 0; i0 < nk; i0 += 64) {
		const int i = i0 + lane;
		const bool valid = i < nk;
		const int kk = valid ? ord[i] : 0;
		const uint64_t uu = valid ? u2[kk] : 0;
		const int len = valid ? (int32_t)uu : 0;
		const int incl = wave_incl_scan(len, lane);
		if (valid) { dest[kk] = run + incl - len; u_out[i] = uu; }
		run += __shfl(incl, 63);
	}
	__syncthreads();
	for (int i0 = 0; i0 < n; i0 += W) {
		int at[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = i0 + lane + 64 * k;
			const int c = i < n ? cd[i] : -1;
			at[k] = c >= 0 ? dest[c >> 16] + (c & 0xffff) : -1;                    // ascending along the chain (chain.c:399-400)
		}
#pragma unroll
		for (int k = 0; k < K; ++k)
			if (at[k] >= 0 && at[k] < n_b) b_out[at[k]] = A.d_a[base + i0 + lane + 64 * k];
	}
	for (int r = lane; r < nu; r += 64) {                                        // chains that kept only their (already taken) peak
		const int kk = by_key ? end2kk[r] : rk2kk[r];
		if (kk < 0 || !(kk & NOT_MINE)) continue;
		const int at = dest[kk & ~NOT_MINE];
		if (at >= 0 && at < n_b) b_out[at] = A.d_a[base + (int32_t)us[r]];
	}
}

} // namespace

size_t epilogue_sort_temp_bytes(int64_t total, int64_t) { return (size_t)total * 4; }   // one int per anchor: the permutation of a replayed pass (epi_tiesort); the sorts themselves need no library scratch

hipError_t launch_chain_epilogue(const EpiArgs &A, hipStream_t st, int *n_launches)
{
	if (A.n_tasks <= 0) return hipSuccess;
	const unsigned nt = (unsigned)A.n_tasks;
	hipError_t e;
	// tasks that fit the LDS (two size classes) take the fused kernel; the others kernels A, B, C.  max_task < 0: sizes known to the device only
	const bool fused = A.fused != 0, small = fused, large = fused && (A.max_task < 0 || A.max_task > FUSE_S),
	           huge = !fused || A.max_task < 0 || A.max_task > FUSE_L;
	int nl = 2;
	if (small) { hipLaunchKernelGGL(epi_fused<FUSE_S>, dim3(nt), dim3(FNT), 0, st, A, -1); ++nl; if ((e = hipGetLastError()) != hipSuccess) return e; }
	if (large) { hipLaunchKernelGGL(epi_fused<FUSE_L>, dim3(nt), dim3(FNT), 0, st, A, FUSE_S); ++nl; if ((e = hipGetLastError()) != hipSuccess) return e; }
	if (huge) {
		hipLaunchKernelGGL(epi_ends, dim3(nt), dim3(64), 0, st, A);
		if ((e = hipGetLastError()) != hipSuccess) return e;
		hipLaunchKernelGGL(epi_claim, dim3(nt), dim3(64), 0, st, A);
		if ((e = hipGetLastError()) != hipSuccess) return e;
		nl += 2;
	}
	hipLaunchKernelGGL(epi_offsets, dim3((nt + 1023) / 1024), dim3(1024), 0, st, A);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	hipLaunchKernelGGL(epi_tiesort, dim3(nt), dim3(64), 0, st, A);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	if (fused) { hipLaunchKernelGGL(epi_emit_cd, dim3(nt), dim3(64), 0, st, A); ++nl; if ((e = hipGetLastError()) != hipSuccess) return e; }
	if (huge) { hipLaunchKernelGGL(epi_emit, dim3(nt), dim3(64), 0, st, A); ++nl; }
	if (n_launches) *n_launches += nl;
	return hipGetLastError();
}

hipError_t warm_epilogue_kernels()
{
	hipFuncAttributes at;
	return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&epi_offsets));
}

} // namespace mm2c
