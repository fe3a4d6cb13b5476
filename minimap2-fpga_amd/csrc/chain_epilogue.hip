// chain_epilogue.hip -- the part of mm_chain_dp after the DP, on the GPU.
//
// Reference: chain.c:106-111 (v[] from f[]/p[]) and chain.c:348-422 (chain ends, peak search, sort by score, backtrack with
// used-marks, score/length filter, emission, chains ordered by the x of their first anchor).  The reference walks the chains
// one after the other on one thread; every step below is a data-parallel restatement with the same result:
//
//   v[i]       = max of f over i and its ancestors in the p[] forest          -> pointer jumping inside 64-anchor chunks
//   chain ends = anchors without a child whose v >= min_sc; each walks back to its peak (f[j] >= v[j])   -> one lane per end
//   sort       = descending on (f[peak] << 32 | peak); the keys are compared in full, so any correct sort gives the
//                reference's order (rocPRIM segmented radix sort, one segment per task)
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
//                everywhere else the order is unique (rocPRIM segmented sort of (x, chain)).
//
// One 64-lane wave per task in each kernel; tasks are independent.  Outputs are compact: chains of task k are
// u[u_off[k] .. u_off[k+1]), their anchors b[b_off[k] .. b_off[k+1]).

#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <climits>
#include "chain_kernel.h"

namespace mm2c {

namespace {

constexpr int NONE = INT_MAX;

__device__ __forceinline__ int lanes_before(uint64_t m)
{
	return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
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

// ---- kernel A: v[], child marks, chain ends -> unsorted keys (chain.c:106-111, 349-367) -------------------------------
__global__ __launch_bounds__(64) void epi_ends(EpiArgs A)
{
	const int task = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int64_t base = A.d_off[task];
	const int n = (int)(A.d_off[task + 1] - base);
	const int lane = (int)threadIdx.x;
	const int32_t *f = A.d_f + base, *p = A.d_p + base;
	int32_t *v = A.v + base, *mark = A.own + base;
	if (lane == 0) A.seg_begin[task] = (uint32_t)base;
	for (int i = lane; i < n; i += 64) mark[i] = 0;
	__syncthreads();
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int i = c0 + lane;
		const bool valid = i < n;
		const int pi = valid ? p[i] : -1;
		int cur = valid ? f[i] : INT_MIN, ptr = pi;
		if (pi >= 0) mark[pi] = 1;                                            // chain.c:350
		if (ptr >= 0 && ptr < c0) { cur = max(cur, v[ptr]); ptr = -1; }       // parent in an earlier chunk: final already
		while (__ballot(ptr >= c0)) {                                         // parents inside the chunk: pointer jumping
			const int src = ptr >= c0 ? ptr - c0 : lane;
			const int oc = __shfl(cur, src), op = __shfl(ptr, src);
			if (ptr >= c0) { cur = max(cur, oc); ptr = op; }
		}
		if (valid) v[i] = cur;
		__syncthreads();
	}
	uint64_t *keys = A.key0 + base;
	int cnt = 0;
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int i = c0 + lane;
		const bool is_end = i < n && mark[i] == 0 && v[i] >= A.min_sc;        // chain.c:352
		uint64_t key = 0;
		if (is_end) {
			int j = i;
			while (j >= 0 && f[j] < v[j]) j = p[j];                           // chain.c:360-361
			if (j < 0) j = i;
			key = (uint64_t)(uint32_t)f[j] << 32 | (uint32_t)j;
		}
		const uint64_t m = __ballot(is_end);
		if (is_end) keys[cnt + lanes_before(m)] = key;
		cnt += __popcll(m);
	}
	if (lane == 0) A.seg_end1[task] = (uint32_t)(base + cnt);
}

// ---- kernel B: owners, depths, per-chain length / score / filter (chain.c:375-390) ----------------------------------
__global__ __launch_bounds__(64) void epi_claim(EpiArgs A)
{
	__shared__ int s_own[64];
	const int task = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int64_t base = A.d_off[task];
	const int n = (int)(A.d_off[task + 1] - base);
	const int lane = (int)threadIdx.x;
	const int nu = (int)(A.seg_end1[task] - (uint32_t)base);
	const int32_t *f = A.d_f + base, *p = A.d_p + base;
	int32_t *own = A.own + base, *dep = A.v + base, *ctop = A.ctop + base, *rk2kk = A.rk2kk + base, *val0 = A.val0 + base;
	const uint64_t *us = A.key1 + base;
	uint64_t *u2 = A.u2 + base, *rkey = A.key0 + base;

	for (int i = lane; i < n; i += 64) own[i] = NONE;
	__syncthreads();
	for (int r = lane; r < nu; r += 64) atomicMin(&own[(int32_t)us[r]], r);       // a peak listed twice belongs to the first listing
	__threadfence();
	__syncthreads();
	// owner(x) = min over the subtree of x: children have larger indices, so chunks go from the end to the front
	for (int c0 = n > 0 ? (n - 1) & ~63 : -64; c0 >= 0; c0 -= 64) {
		const int i = c0 + lane;
		const bool valid = i < n;
		const int pi = valid ? p[i] : -1;
		s_own[lane] = valid ? __hip_atomic_load(&own[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : NONE;   // sees the atomics of later chunks
		int up = pi >= c0 ? pi - c0 : -1;                                         // 2^t-th ancestor, while it is inside the chunk
		__syncthreads();
		while (__ballot(up >= 0)) {
			const int acc = s_own[lane];
			if (up >= 0 && acc != NONE) atomicMin(&s_own[up], acc);
			const int nup = __shfl(up, up >= 0 ? up : lane);
			up = up >= 0 ? nup : -1;
			__syncthreads();
		}
		const int fin = s_own[lane];
		if (valid) {
			own[i] = fin;
			if (pi >= 0 && pi < c0 && fin != NONE) atomicMin(&own[pi], fin);
		}
		__syncthreads();
	}
	__threadfence();
	// depth inside the owner path (0 = top) and the top of every path
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int i = c0 + lane;
		const bool valid = i < n;
		const int o = valid ? own[i] : NONE, pi = valid ? p[i] : -1;
		const bool claimed = o != NONE;
		const bool link = claimed && pi >= 0 && own[pi] == o;
		int d = 0, ptr = -1;
		if (link) { if (pi < c0) d = dep[pi] + 1; else { d = 1; ptr = pi; } }
		while (__ballot(ptr >= c0)) {
			const int src = ptr >= c0 ? ptr - c0 : lane;
			const int od = __shfl(d, src), op = __shfl(ptr, src);
			if (ptr >= c0) { d += od; ptr = op; }
		}
		if (valid) dep[i] = d;
		if (claimed && !link) ctop[o] = i;
		__syncthreads();
	}
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
			const bool mine = own[j] == r;
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
}

// ---- exclusive scans of the per-task chain / anchor counts -> compact output offsets ---------------------------------------
__global__ __launch_bounds__(1024) void epi_offsets(EpiArgs A)
{
	__shared__ int64_t s_u[1024], s_b[1024];
	const int64_t nt = A.n_tasks, per = (nt + 1023) / 1024;
	const int64_t t0 = min(nt, (int64_t)threadIdx.x * per), t1 = min(nt, t0 + per);
	int64_t su = 0, sb = 0;
	for (int64_t t = t0; t < t1; ++t) { su += A.cnt_u[t]; sb += A.cnt_b[t]; }
	s_u[threadIdx.x] = su; s_b[threadIdx.x] = sb;
	__syncthreads();
	if (threadIdx.x == 0) {
		int64_t au = 0, ab = 0;
		for (int k = 0; k < 1024; ++k) { const int64_t xu = s_u[k], xb = s_b[k]; s_u[k] = au; s_b[k] = ab; au += xu; ab += xb; }
		A.u_off[nt] = au; A.b_off[nt] = ab;
	}
	__syncthreads();
	su = s_u[threadIdx.x]; sb = s_b[threadIdx.x];
	for (int64_t t = t0; t < t1; ++t) { A.u_off[t] = su; A.b_off[t] = sb; su += A.cnt_u[t]; sb += A.cnt_b[t]; }
}

// ---- the passes of radix_sort_128x (ksort.h:101-151) on (x, chain) records, one lane -------------------------------
__device__ void insertion_pass(uint64_t *x, int32_t *c, int lo, int hi)
{
	for (int q = lo + 1; q < hi; ++q) {
		if (x[q] >= x[q - 1]) continue;
		const uint64_t kx = x[q]; const int32_t kc = c[q];
		int r = q;
		for (; r > lo && kx < x[r - 1]; --r) { x[r] = x[r - 1]; c[r] = c[r - 1]; }
		x[r] = kx; c[r] = kc;
	}
}

__device__ void flag_sort_one_lane(uint64_t *x, int32_t *c, int n, int32_t *stack, int *hist, int *blo, int *bhi)
{
	if (n <= 64) { insertion_pass(x, c, 0, n); return; }
	int sp = 0;
	stack[0] = 0; stack[1] = n; stack[2] = 56; sp = 3;
	while (sp > 0) {
		sp -= 3;
		const int lo = stack[sp], hi = stack[sp + 1], shift = stack[sp + 2];
		for (int d = 0; d < 256; ++d) hist[d] = 0;
		for (int q = lo; q < hi; ++q) ++hist[(int)(x[q] >> shift) & 255];
		for (int d = 0, q = lo; d < 256; ++d) { blo[d] = q; q += hist[d]; bhi[d] = q; }
		for (int d = 0; d < 256; ) {
			if (blo[d] == bhi[d]) { ++d; continue; }
			int dst = (int)(x[blo[d]] >> shift) & 255;
			if (dst == d) { ++blo[d]; continue; }
			uint64_t hx = x[blo[d]]; int32_t hc = c[blo[d]];
			do {
				const int at = blo[dst]++;
				const uint64_t nx = x[at]; const int32_t nc = c[at];
				x[at] = hx; c[at] = hc; hx = nx; hc = nc;
				dst = (int)(hx >> shift) & 255;
			} while (dst != d);
			x[blo[d]] = hx; c[blo[d]] = hc; ++blo[d];
		}
		if (shift == 0) continue;
		const int ns = shift > 8 ? shift - 8 : 0;
		for (int d = 0, q = lo; d < 256; ++d) {
			const int e = bhi[d];
			if (e - q > 64) { stack[sp] = q; stack[sp + 1] = e; stack[sp + 2] = ns; sp += 3; }
			else if (e - q > 1) insertion_pass(x, c, q, e);
			q = e;
		}
	}
}

// ---- kernel C: final chain order, u[] and b[] (chain.c:397-420) -----------------------------------------------------
__global__ __launch_bounds__(64) void epi_emit(EpiArgs A)
{
	__shared__ int s_hist[256], s_lo[256], s_hi[256];
	const int task = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int64_t base = A.d_off[task];
	const int n = (int)(A.d_off[task + 1] - base);
	const int lane = (int)threadIdx.x;
	const int nu = (int)(A.seg_end1[task] - (uint32_t)base), nk = (int)(A.seg_end2[task] - (uint32_t)base);
	if (nk == 0) return;
	const int32_t *own = A.own + base, *dep = A.v + base, *rk2kk = A.rk2kk + base;
	int32_t *dest = A.dest + base, *ord = A.val1 + base;
	uint64_t *sx = A.rkey1 + base;
	const uint64_t *u2 = A.u2 + base, *us = A.key1 + base;
	uint64_t *u_out = A.u_out + A.u_off[task];
	ulonglong2 *b_out = A.b_out + A.b_off[task];
	const int n_b = (int)(A.b_off[task + 1] - A.b_off[task]);

	if (nk > 64) {                                                               // <= 64 records: insertion sort, stable (ksort.h:141-143)
		bool tie = false;
		for (int i = lane; i + 1 < nk; i += 64) tie |= sx[i] == sx[i + 1];
		if (__ballot(tie)) {
			__syncthreads();
			for (int i = lane; i < nk; i += 64) { sx[i] = A.key0[base + i]; ord[i] = i; }   // back to rank order
			__syncthreads();
			if (lane == 0) flag_sort_one_lane(sx, ord, nk, dest, s_hist, s_lo, s_hi);
			__syncthreads();
		}
	}
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
	for (int i = lane; i < n; i += 64) {
		const int o = own[i];
		if (o == NONE) continue;
		const int kk = rk2kk[o];
		if (kk < 0) continue;
		const int at = dest[kk] + dep[i];                                       // ascending along the chain (chain.c:399-400)
		if (at >= 0 && at < n_b) b_out[at] = A.d_a[base + i];
	}
	for (int r = lane; r < nu; r += 64) {                                        // chains that kept only their (already taken) peak
		const int kk = rk2kk[r];
		if (kk < 0) continue;
		const int j = (int32_t)us[r];
		if (own[j] != r && dest[kk] >= 0 && dest[kk] < n_b) b_out[dest[kk]] = A.d_a[base + j];
	}
}

} // namespace

size_t epilogue_sort_temp_bytes(int64_t total, int64_t n_tasks)
{
	size_t s1 = 0, s2 = 0;
	uint64_t *k = nullptr; int32_t *v = nullptr; uint32_t *o = nullptr;
	(void)rocprim::segmented_radix_sort_keys_desc(nullptr, s1, k, k, (unsigned)total, (unsigned)n_tasks, o, o, 0, 64, (hipStream_t)0);
	(void)rocprim::segmented_radix_sort_pairs(nullptr, s2, k, k, v, v, (unsigned)total, (unsigned)n_tasks, o, o, 0, 64, (hipStream_t)0);
	return s1 > s2 ? s1 : s2;
}

hipError_t launch_chain_epilogue(const EpiArgs &A, hipStream_t st, int *n_launches)
{
	if (A.n_tasks <= 0) return hipSuccess;
	const unsigned nt = (unsigned)A.n_tasks, tot = (unsigned)A.total;
	hipError_t e;
	size_t tmp = A.sort_tmp_bytes;
	hipLaunchKernelGGL(epi_ends, dim3(nt), dim3(64), 0, st, A);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	e = rocprim::segmented_radix_sort_keys_desc(A.sort_tmp, tmp, A.key0, A.key1, tot, nt, A.seg_begin, A.seg_end1, 0, 64, st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(epi_claim, dim3(nt), dim3(64), 0, st, A);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	hipLaunchKernelGGL(epi_offsets, dim3(1), dim3(1024), 0, st, A);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	tmp = A.sort_tmp_bytes;
	e = rocprim::segmented_radix_sort_pairs(A.sort_tmp, tmp, A.key0, A.rkey1, A.val0, A.val1, tot, nt, A.seg_begin, A.seg_end2, 0, 64, st);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(epi_emit, dim3(nt), dim3(64), 0, st, A);
	if (n_launches) *n_launches += 6;
	return hipGetLastError();
}

} // namespace mm2c
