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
//                 tracked as an index array, the cycle-leader distribution of ksort.h:117-131 runs as a walk over the digits (in LDS) on one lane per
//                 pass (its outcome depends on the order of the swaps), only for buckets that hold equal keys (a bucket is a
//                 position range, so the sorted output tells which ones do); buckets of <= 64 records are insertion-sorted by the
//                 reference (stable), so a final stable sort of the replayed arrangement gives the reference's array.

#include <hip/hip_runtime.h>
#include <climits>
#include "chain_kernel.h"
#include "radix_replay.h"

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

// where a read's anchors are: the plan's offsets give the CAPACITY of a read (every hit kept); with skip_seed (map.c:122-147) a read keeps
// cnt[read] <= capacity anchors, the work arrays (unsorted, scratch, tiecnt) stay laid out by capacity and the result is packed by out_off
struct ReadGeom { int64_t a0, o0; int na; };
__device__ __forceinline__ ReadGeom read_geom(const SeedArgs &A, int read)
{
	ReadGeom g;
	g.a0 = A.d_anchor_off[read];
	g.na = A.d_count ? A.d_count[read] : (int)(A.d_anchor_off[read + 1] - g.a0);
	g.o0 = A.d_out_off ? A.d_out_off[read] : g.a0;
	return g;
}

// one hit of a match as an anchor (map.c:222-243 / :175-188): false when skip_seed (map.c:122-147; names as ranks, see mm2chain.h) drops it
__device__ __forceinline__ bool hit_anchor(const SeedArgs &A, uint64_t r, uint32_t q_pos, uint32_t q_span, uint32_t segt, int qlen, int q_lo, int q_eq, ulonglong2 &a)
{
	const uint32_t rpos = (uint32_t)r >> 1;
	const bool fwd = (r & 1) == (q_pos & 1);
	bool keep = true, is_self = false;
	if (A.skip_flag) {
		if (A.d_ref_rank && (A.skip_flag & (0x001 | 0x002))) {
			const int rid = (int)(r >> 32);
			const int rr = A.d_ref_rank[rid];
			const int cmp = rr < q_lo ? 1 : (q_eq && rr == q_lo) ? 0 : -1;
			if ((A.skip_flag & 0x001) && cmp == 0 && A.d_ref_len[rid] == qlen) {   // MM_F_NO_DIAG
				if (rpos == (q_pos >> 1)) keep = false;                 // the diagonal
				if (fwd) is_self = true;
			}
			if ((A.skip_flag & 0x002) && cmp > 0) keep = false;         // MM_F_NO_DUAL: every pair once
		}
		if (fwd ? (A.skip_flag & 0x200000) : (A.skip_flag & 0x100000)) keep = false;   // MM_F_REV_ONLY / MM_F_FOR_ONLY
	}
	if (fwd) {                                                      // forward strand, map.c:232-234
		a.x = (r & 0xffffffff00000000ULL) | rpos;
		a.y = (uint64_t)q_span << 32 | q_pos >> 1;
	} else {                                                        // reverse strand, map.c:235-238
		a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | rpos;
		a.y = (uint64_t)q_span << 32 | (uint32_t)((uint32_t)qlen - ((q_pos >> 1) + 1 - q_span) - 1);
	}
	a.y |= (uint64_t)(segt >> 1) << 48;                             // MM_SEED_SEG_SHIFT, map.c:239
	if (segt & 1) a.y |= 1ULL << 42;                                // MM_SEED_TANDEM, map.c:240
	if (is_self) a.y |= 1ULL << 43;                                 // MM_SEED_SELF, map.c:241
	return keep;
}

// Long reads (round 6): which reads the sixteen-wave expansion takes (seed_expand_mw below) -- more anchors than the LDS sorts hold, every hit kept (with skip_seed the
// place of an anchor depends on what the hits before it decided: one wave), no more chunks of 64 matches than its table of chunk totals holds
constexpr int EXPAND_MW_WAVES = 16, EXPAND_MW_CHUNKS = 16384, EXPAND_MW_ABOVE = 16384;   // (EXPAND_MW_ABOVE = SORT_LDS_CAP below: the reads the host counts in n_sort_huge)
__device__ __forceinline__ bool expand_mw_takes(const SeedArgs &A, int read)
{
	return A.mw_sort && !A.d_count && A.d_anchor_off[read + 1] - A.d_anchor_off[read] > EXPAND_MW_ABOVE && (A.d_match_off[read + 1] - A.d_match_off[read] + 63) / 64 <= EXPAND_MW_CHUNKS;
}

// ---- kernel 1: expansion (map.c:222-243) ----------------------------------------------------------------------------
__global__ __launch_bounds__(64) void seed_expand(SeedArgs A)
{
	__shared__ int s_start[65];
	__shared__ uint32_t s_qpos[64], s_span[64], s_segt[64];
	__shared__ int64_t s_cr[64];
	const int read = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	if (expand_mw_takes(A, read)) return;
	const int lane = (int)threadIdx.x;
	const int64_t m0 = A.d_match_off[read], a0 = A.d_anchor_off[read];
	const int nm = (int)(A.d_match_off[read + 1] - m0), na = (int)(A.d_anchor_off[read + 1] - a0);
	const int qlen = A.d_qlen[read];
	const Match *m = A.d_matches + m0;
	ulonglong2 *out = A.unsorted + a0;
	int run = 0, hits_seen = 0;                                                 // anchors written / hits looked at so far (wave uniform)
	const int q_lo = A.d_q_lo ? A.d_q_lo[read] : 0, q_eq = A.d_q_eq ? A.d_q_eq[read] : 0;
	uint64_t x_or = 0, x_and = ~0ull;                                           // -> the bits of x that are not the same in every anchor of the read
	for (int c0 = 0; c0 < nm; c0 += 64) {
		const int i = c0 + lane;
		Match q = {};
		if (i < nm) q = m[i];
		if (A.n_hits > 0 && __ballot(i < nm && (q.cr_off < 0 || q.cr_off + (int64_t)q.n > A.n_hits))) {
			if (lane == 0) { A.status[read] = 2; if (A.d_count) A.d_count[read] = 0; }   // a match reaches beyond the hit pool the caller declared
			return;
		}
		const int incl = wave_incl_scan((int)q.n, lane);
		const int total = __shfl(incl, 63);
		__syncthreads();
		s_start[lane] = incl - (int)q.n; s_cr[lane] = q.cr_off; s_qpos[lane] = q.q_pos; s_span[lane] = q.q_span; s_segt[lane] = q.seg_tandem;
		if (lane == 63) s_start[64] = total;
		__syncthreads();
		if (hits_seen + total > na) { if (lane == 0) { A.status[read] = 1; if (A.d_count) A.d_count[read] = 0; } return; }   // the caller's anchor offsets do not match the hit counts
		for (int t0 = 0; t0 < total; t0 += 64) {
			const int t = t0 + lane;
			bool keep = t < total;
			ulonglong2 a = {0, 0};
			if (keep) {
				int lo = 0, hi = 63;                                            // last match of the chunk with start <= t
				while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_start[mid] <= t) lo = mid; else hi = mid - 1; }
				const uint64_t r = A.d_hits[s_cr[lo] + (t - s_start[lo])];
				keep = hit_anchor(A, r, s_qpos[lo], s_span[lo], s_segt[lo], qlen, q_lo, q_eq, a);
			}
			const uint64_t km = __ballot(keep);                                 // kept hits stay in hit order (the order collect_seed_hits fills a[])
			if (keep) {
				out[run + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0))] = a;
				x_or |= a.x; x_and &= a.x;
			}
			run += (int)__builtin_popcountll(km);
		}
		hits_seen += total;
	}
	// the hits of the read's matches must add up to its anchor range in both modes (with skip_seed the range is a capacity and `run` may be smaller)
	if (hits_seen != na) { if (lane == 0) { A.status[read] = 1; if (A.d_count) A.d_count[read] = 0; } return; }
	if (A.d_count) { if (lane == 0) A.d_count[read] = run; }
	else if (run != na && lane == 0) A.status[read] = 1;
	x_or = wave_or(x_or); x_and = ~wave_or(~x_and);
	if (lane == 0) A.xdiff[read] = run > 0 ? x_or ^ x_and : 0;
}

// ---- kernel 1b (round 6): the expansion of a LONG read on sixteen waves.  seed_expand walks a read's matches 64 at a time on one wave -- 34 ms for a read of 10^6
// anchors, one read per CU and the other waves of the CU idle.  Where an anchor lands is the prefix sum of the hit counts before its match, so: (1) every wave adds up the hit
// counts of its chunks of 64 matches, (2) one block-wide exclusive scan over the chunk totals, (3) every wave expands its chunks at their places -- the same lanes, the same
// owner search, the same encoding (hit_anchor) as seed_expand, chunk by chunk.  Every hit is kept here (no skip_seed), so the totals are the places.
template <int NW>
__global__ __launch_bounds__(64 * NW) void seed_expand_mw(SeedArgs A)
{
	constexpr int NT = 64 * NW;
	__shared__ int s_tot[EXPAND_MW_CHUNKS];
	__shared__ int s_start[NW][65];
	__shared__ uint32_t s_qpos[NW][64], s_span[NW][64], s_segt[NW][64];
	__shared__ int64_t s_cr[NW][64];
	__shared__ int s_part[NT], s_bad;
	__shared__ unsigned long long s_or, s_and;
	const int read = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	if (!expand_mw_takes(A, read)) return;
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int64_t m0 = A.d_match_off[read], a0 = A.d_anchor_off[read];
	const int nm = (int)(A.d_match_off[read + 1] - m0), na = (int)(A.d_anchor_off[read + 1] - a0);
	const int qlen = A.d_qlen[read];
	const Match *m = A.d_matches + m0;
	ulonglong2 *out = A.unsorted + a0;
	const int n_chunks = (nm + 63) / 64;
	if (tid == 0) { s_bad = 0; s_or = 0; s_and = ~0ull; }
	__syncthreads();
	// (1) totals of the chunks (and the check of seed_expand: no match may reach beyond the hit pool the caller declared)
	for (int c = wave; c < n_chunks; c += NW) {
		const int i = 64 * c + lane;
		Match q = {};
		if (i < nm) q = m[i];
		if (A.n_hits > 0 && i < nm && (q.cr_off < 0 || q.cr_off + (int64_t)q.n > A.n_hits)) s_bad = 2;
		unsigned long long n64 = q.n;                                               // the chunk's hits in 64 bits: a chunk with more than the read's anchors is an error (and keeps the int sums exact)
		for (int d = 1; d < 64; d <<= 1) n64 += (unsigned long long)__shfl_xor((long long)n64, d);
		if (n64 > (unsigned long long)na) { if (!s_bad) s_bad = 1; n64 = 0; }
		if (lane == 63) s_tot[c] = (int)n64;
	}
	__syncthreads();
	if (s_bad) { if (tid == 0) A.status[read] = s_bad; return; }
	// (2) exclusive scan over the chunk totals: every thread a run of consecutive chunks
	const int per = (n_chunks + NT - 1) / NT, c_lo = min(tid * per, n_chunks), c_hi = min(c_lo + per, n_chunks);
	long long mine = 0;
	for (int c = c_lo; c < c_hi; ++c) mine += s_tot[c];
	{
		// (an int suffices for the places: the grand total must equal na < 2^31, checked below; a partial sum that overflowed shows up as a mismatch there)
		int v = (int)min(mine, (long long)INT_MAX);
		const int incl = wave_incl_scan(v, lane);
		if (lane == 63) s_part[wave] = incl;
		__syncthreads();
		int before = 0;
		for (int w = 0; w < wave; ++w) before += s_part[w];
		int at = before + incl - v;
		int grand = 0;
		for (int w = 0; w < NW; ++w) grand += s_part[w];
		__syncthreads();
		for (int c = c_lo; c < c_hi; ++c) { const int t = s_tot[c]; s_tot[c] = at; at += t; }
		if (grand != na || mine > INT_MAX) { if (tid == 0) A.status[read] = 1; return; }        // the caller's anchor offsets do not match the hit counts (uniform: every thread sees the same sums)
	}
	__syncthreads();
	// (3) expansion, chunk by chunk as seed_expand does it
	uint64_t x_or = 0, x_and = ~0ull;
	for (int c = wave; c < n_chunks; c += NW) {
		const int i = 64 * c + lane;
		Match q = {};
		if (i < nm) q = m[i];
		const int incl = wave_incl_scan((int)q.n, lane);
		const int total = __shfl(incl, 63);
		s_start[wave][lane] = incl - (int)q.n; s_cr[wave][lane] = q.cr_off; s_qpos[wave][lane] = q.q_pos; s_span[wave][lane] = q.q_span; s_segt[wave][lane] = q.seg_tandem;
		if (lane == 63) s_start[wave][64] = total;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (a wave's own rows: no workgroup barrier)
		const int run = s_tot[c];
		for (int t0 = 0; t0 < total; t0 += 64) {
			const int t = t0 + lane;
			if (t < total) {
				int lo = 0, hi = 63;                                            // last match of the chunk with start <= t
				while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_start[wave][mid] <= t) lo = mid; else hi = mid - 1; }
				const uint64_t r = A.d_hits[s_cr[wave][lo] + (t - s_start[wave][lo])];
				ulonglong2 a = {0, 0};
				hit_anchor(A, r, s_qpos[wave][lo], s_span[wave][lo], s_segt[wave][lo], qlen, 0, 0, a);   // (skip_flag is 0 here: every hit is kept)
				out[run + t] = a;
				x_or |= a.x; x_and &= a.x;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the rows are rewritten for the wave's next chunk
	}
	x_or = wave_or(x_or); x_and = ~wave_or(~x_and);
	if (lane == 0) { atomicOr(&s_or, x_or); atomicAnd(&s_and, x_and); }
	__syncthreads();
	if (tid == 0) A.xdiff[read] = na > 0 ? s_or ^ s_and : 0;
}

// ---- packed offsets of the result: exclusive prefix sums of the per-read counts (one block of 1024 threads) ----------------
__global__ __launch_bounds__(1024) void seed_offsets(SeedArgs A)
{
	__shared__ int64_t s_part[1024];
	const int tid = (int)threadIdx.x;
	const int64_t per = (A.n_reads + 1023) / 1024, r0 = tid * per, r1 = r0 + per < A.n_reads ? r0 + per : A.n_reads;
	int64_t sum = 0;
	for (int64_t r = r0; r < r1; ++r) sum += A.d_count[r];
	s_part[tid] = sum;
	__syncthreads();
	if (tid == 0) { int64_t at = 0; for (int k = 0; k < 1024; ++k) { const int64_t v = s_part[k]; s_part[k] = at; at += v; } A.d_out_off[A.n_reads] = at; }
	__syncthreads();
	int64_t at = s_part[tid];
	for (int64_t r = r0; r < r1; ++r) { A.d_out_off[r] = at; at += A.d_count[r]; }
}

// ---- stable LSD radix sort of one read's anchors on x by one wave; the result ends in `dst` ------------------------
// `src` is only read; the passes write alternately to `alt` and `dst`, arranged so that the last one writes `dst` (alt may be src)
__device__ void wave_sort_anchors(const ulonglong2 *src, ulonglong2 *alt, ulonglong2 *dst, int n, int lane, int *s_cnt /* 256 ints of LDS */)
{
	if (n <= 0) return;
	uint64_t diff = 0;
	const uint64_t first = src[0].x;
	for (int i = lane; i < n; i += 64) diff |= src[i].x ^ first;
	diff = wave_or(diff);
	int passes = 0;
	for (int shift = 0; shift < 64; shift += 8) passes += ((diff >> shift) & 255) != 0;
	if (passes == 0) { if (src != dst) for (int i = lane; i < n; i += 64) dst[i] = src[i]; __syncthreads(); return; }
	const ulonglong2 *from = src;
	ulonglong2 *to = (passes & 1) ? dst : alt;                                   // an odd number of passes: dst, alt, dst, ...
	for (int shift = 0; shift < 64; shift += 8) {
		if (((diff >> shift) & 255) == 0) continue;
		for (int d = lane; d < 256; d += 64) s_cnt[d] = 0;
		__syncthreads();
		for (int i0 = 0; i0 < n; i0 += 256) {                                    // four loads in flight per lane
			uint64_t x[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) { const int i = i0 + 64 * k + lane; x[k] = i < n ? from[i].x : 0; }
#pragma unroll
			for (int k = 0; k < 4; ++k) if (i0 + 64 * k + lane < n) atomicAdd(&s_cnt[(int)(x[k] >> shift) & 255], 1);
		}
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
		for (int i0 = 0; i0 < n; i0 += 256) {                                    // 256 records per step: one round trip to memory for four chunks
			ulonglong2 rec[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) { const int i = i0 + 64 * k + lane; rec[k] = ulonglong2{0, 0}; if (i < n) rec[k] = from[i]; }
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const bool valid = i0 + 64 * k + lane < n;
				const int d = (int)(rec[k].x >> shift) & 255;
				uint64_t peers = __ballot(valid);
#pragma unroll
				for (int b = 0; b < 8; ++b) {
					const uint64_t bal = __ballot((d >> b) & 1);
					peers &= ((d >> b) & 1) ? bal : ~bal;
				}
				const int rank = lanes_before(peers);
				if (valid) to[s_cnt[d] + rank] = rec[k];
				__syncthreads();
				if (valid && rank == 0) s_cnt[d] += __popcll(peers);
				__syncthreads();
			}
		}
		from = to;
		to = to == dst ? alt : dst;
	}
	__syncthreads();
}

// ---- stable LSD radix sort of 8-byte keys on the bits [bit_lo, bit_lo + n_bits), ping-pong between a and b; returns the buffer that holds
// the result ----
// s_hist (optional): the digit histograms of all passes, 256 ints per pass, counted by the caller while it made the keys (saves reading
// the keys once more per pass)
__device__ uint64_t *wave_sort_keys(uint64_t *a, uint64_t *b, int n, int bit_lo, int n_bits, int lane, int *s_cnt /* 256 ints of LDS */,
                                    const int *s_hist = nullptr)
{
	uint64_t *from = a, *to = b;
	for (int shift = bit_lo, pass = 0; shift < bit_lo + n_bits; shift += 8, ++pass) {
		const int mask = bit_lo + n_bits - shift >= 8 ? 255 : (1 << (bit_lo + n_bits - shift)) - 1;
		if (s_hist) {
			for (int d = lane; d < 256; d += 64) s_cnt[d] = s_hist[256 * pass + d];
		} else {
			for (int d = lane; d < 256; d += 64) s_cnt[d] = 0;
			__syncthreads();
			for (int i0 = 0; i0 < n; i0 += 256) {
				uint64_t k[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u + lane; k[u] = i < n ? from[i] : 0; }
#pragma unroll
				for (int u = 0; u < 4; ++u) if (i0 + 64 * u + lane < n) atomicAdd(&s_cnt[(int)(k[u] >> shift) & mask], 1);
			}
		}
		__syncthreads();
		{
			int h[4], sum = 0;
#pragma unroll
			for (int u = 0; u < 4; ++u) { h[u] = s_cnt[4 * lane + u]; sum += h[u]; }
			int at = wave_incl_scan(sum, lane) - sum;
			__syncthreads();
#pragma unroll
			for (int u = 0; u < 4; ++u) { s_cnt[4 * lane + u] = at; at += h[u]; }
		}
		__syncthreads();
		for (int i0 = 0; i0 < n; i0 += 256) {
			uint64_t k[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u + lane; k[u] = i < n ? from[i] : 0; }
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const bool valid = i0 + 64 * u + lane < n;
				const int d = (int)(k[u] >> shift) & mask;
				uint64_t peers = __ballot(valid);
#pragma unroll
				for (int bb = 0; bb < 8; ++bb) {
					const uint64_t bal = __ballot((d >> bb) & 1);
					peers &= ((d >> bb) & 1) ? bal : ~bal;
				}
				const int rank = lanes_before(peers);
				if (valid) to[s_cnt[d] + rank] = k[u];
				__syncthreads();
				if (valid && rank == 0) s_cnt[d] += __popcll(peers);
				__syncthreads();
			}
		}
		{ uint64_t *t = from; from = to; to = t; }
	}
	__syncthreads();
	return from;
}

// ---- kernel 2a: the same sort for reads of up to SORT_LDS_CAP anchors whose differing bits of x fit 32, entirely in LDS ------------------------
// seed_sort moves 8-byte keys through global memory, four scattered passes for every read of a batch at once: the lines are evicted half written
// (thousands of reads in flight, two 40 KB buffers each) and the passes run at the speed of that traffic.  Here a workgroup of four or eight waves keeps
// the read's squeezed keys (4 bytes) and two index buffers (2 bytes each) in LDS -- 8 bytes per anchor -- and only the final gather
// of the 16-byte anchors touches memory.  Stable LSD radix sort of the indices on the key's bytes: per step of 64 * NW indices every wave ranks its own 64
// (peers by ballots) and publishes its per-digit counts; an index lands at the digit's cursor + the counts of the waves before its own + its rank.
constexpr int SORT_LDS_CAP0 = 5120, SORT_LDS_CAP = 16384;                    // two size classes: four waves and 49 KB (three reads per CU), eight waves and 141 KB (one)

template <int CAP, int PREV, int NW>
__global__ __launch_bounds__(64 * NW) void seed_sort_lds(SeedArgs A, int blk0)
{
	constexpr int NT = 64 * NW;
	__shared__ uint32_t s_key[CAP];
	__shared__ uint16_t s_ia[CAP], s_ib[CAP];
	__shared__ int s_cnt[256], s_w[256 * NW], s_carry[NW + 1], s_hist[4 * 256];
	const int read = A.d_order ? A.d_order[blk0 + blockIdx.x] : blk0 + (int)blockIdx.x;
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const ReadGeom g = read_geom(A, read);
	const int na = g.na;
	{ const int64_t cap = A.d_anchor_off[read + 1] - g.a0; if (cap > CAP || cap <= PREV) return; }   // by capacity, as the launch order and seed_sort's test
	if (A.status[read] != 0 || na == 0) { if (tid == 0) A.has_ties[read] = 0; return; }
	const uint64_t diff = A.xdiff[read];
	const uint32_t dlo = (uint32_t)diff, dhi = (uint32_t)(diff >> 32) & 0x7fffffffu;
	const int b0 = dlo ? 32 - __clz((int)dlo) : 0, b1 = dhi ? 32 - __clz((int)dhi) : 0, bs = (int)(diff >> 63);   // position, target id, strand
	const int kb = b0 + b1 + bs;
	if (kb > 32) return;                                                       // seed_sort takes it
	const ulonglong2 *un = A.unsorted + g.a0;
	ulonglong2 *out = A.d_anchors + g.o0;
	int32_t *tiecnt = A.tiecnt + g.a0;
	uint32_t *srt = A.tie_id + g.a0;
	const uint64_t m0 = b0 >= 32 ? 0xffffffffull : (1ull << b0) - 1, m1 = (1ull << b1) - 1;
	for (int d = tid; d < 4 * 256; d += NT) s_hist[d] = 0;
	__syncthreads();
	for (int i = tid; i < na; i += NT) {
		const uint64_t x = un[i].x;
		const uint32_t sq = (uint32_t)((x & m0) | (((x >> 32) & m1) << b0) | (bs ? (x >> 63) << (b0 + b1) : 0));   // order preserving: the dropped bits are constant
		s_key[i] = sq;
		s_ia[i] = (uint16_t)i;
		for (int pass = 0; 8 * pass < kb; ++pass) atomicAdd(&s_hist[256 * pass + ((sq >> (8 * pass)) & 255)], 1);  // the histograms of all passes, counted while the keys are made
	}
	uint16_t *from = s_ia, *to = s_ib;
	for (int shift = 0; shift < kb; shift += 8) {
		for (int d = tid; d < 256 * NW; d += NT) s_w[d] = 0;
		__syncthreads();                                                       // (also: the keys, histograms and the indices of the pass before are in place)
		if (wave == 0) {
			int h[4], sum = 0;
#pragma unroll
			for (int u = 0; u < 4; ++u) { h[u] = s_hist[32 * shift + 4 * lane + u]; sum += h[u]; }
			int at = wave_incl_scan(sum, lane) - sum;
#pragma unroll
			for (int u = 0; u < 4; ++u) { s_cnt[4 * lane + u] = at; at += h[u]; }
		}
		__syncthreads();
		for (int i0 = 0; i0 < na; i0 += NT) {
			const int i = i0 + tid;
			const bool valid = i < na;
			const int id = valid ? (int)from[i] : 0;
			const int d = valid ? (int)((s_key[id] >> shift) & 255) : 0;
			uint64_t peers = __ballot(valid);
#pragma unroll
			for (int bb = 0; bb < 8; ++bb) {
				const uint64_t bal = __ballot((d >> bb) & 1);
				peers &= ((d >> bb) & 1) ? bal : ~bal;
			}
			const int rank = lanes_before(peers);
			if (valid && rank == 0) s_w[256 * wave + d] = __popcll(peers);
			__syncthreads();
			if (valid) {
				int before = 0;
#pragma unroll
				for (int w = 0; w < NW - 1; ++w) before += w < wave ? s_w[256 * w + d] : 0;
				to[s_cnt[d] + before + rank] = (uint16_t)id;
			}
			__syncthreads();
			if (valid && rank == 0) { atomicAdd(&s_cnt[d], __popcll(peers)); s_w[256 * wave + d] = 0; }
			__syncthreads();
		}
		{ uint16_t *t = from; from = to; to = t; }
	}
	__syncthreads();
	// gather, which anchor stands where, and tiecnt[i] = number of positions j < i with x[j] == x[j + 1] (as seed_sort leaves them)
	int run = 0;
	for (int i0 = 0; i0 < na; i0 += NT) {
		const int i = i0 + tid;
		const int id = i < na ? (int)from[i] : 0, idn = i + 1 < na ? (int)from[i + 1] : 0;
		if (i < na) { out[i] = un[id]; srt[i] = (uint32_t)id; }
		const int flag = (i + 1 < na && s_key[id] == s_key[idn]) ? 1 : 0;
		const int incl = wave_incl_scan(flag, lane);
		if (lane == 63) s_carry[wave + 1] = incl;
		__syncthreads();
		int before = run;
#pragma unroll
		for (int w = 0; w < NW; ++w) { const int c = s_carry[w + 1]; before += w < wave ? c : 0; run += c; }
		if (i < na) tiecnt[i] = before + incl - flag;
		__syncthreads();
	}
	if (tid == 0) A.has_ties[read] = run > 0 ? 1 : 0;
}

// ---- kernel 2: sort + tie detection ---------------------------------------------------------------------------------
// The anchors are not moved pass by pass: each becomes one 8-byte key = (the bits of x that differ inside the read, squeezed
// together) << id_bits | position in the unsorted array.  Sorting the keys on the x bits with a stable sort and gathering the anchors
// once at the end moves 8 bytes per pass instead of 16 and needs ceil(differing bits / 8) passes (4 for one chromosome-sized target,
// against 6 byte positions of the full x).  Falls back to sorting the anchors themselves if the squeezed bits do not fit.
__global__ __launch_bounds__(64) void seed_sort(SeedArgs A)
{
	__shared__ int s_cnt[256], s_hist[4 * 256];
	const int read = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int lane = (int)threadIdx.x;
	const ReadGeom g = read_geom(A, read);
	const int64_t a0 = g.a0;
	const int na = g.na;
	if (A.status[read] != 0 || na == 0) { if (lane == 0) A.has_ties[read] = 0; return; }
	// the unsorted array must survive (the replay of kernel 3 starts from it): it is only read
	const ulonglong2 *un = A.unsorted + a0;
	ulonglong2 *tmp = A.scratch + a0, *out = A.d_anchors + g.o0;
	int32_t *tiecnt = A.tiecnt + a0;
	uint32_t *srt = A.tie_id + a0;
	const uint64_t diff = A.xdiff[read];
	const uint32_t dlo = (uint32_t)diff, dhi = (uint32_t)(diff >> 32) & 0x7fffffffu;
	const int b0 = dlo ? 32 - __clz((int)dlo) : 0, b1 = dhi ? 32 - __clz((int)dhi) : 0, bs = (int)(diff >> 63);   // position, target id, strand
	const int kb = b0 + b1 + bs, idb = na > 1 ? 32 - __clz(na - 1) : 1;
	if (A.lds_sort && A.d_anchor_off[read + 1] - a0 <= SORT_LDS_CAP && kb <= 32) return;   // seed_sort_lds has sorted this read (it goes by the read's capacity: the launch order does)
	if (A.mw_sort && A.d_anchor_off[read + 1] - a0 > SORT_LDS_CAP && kb + idb <= 64) return;   // seed_sort_mw takes it (long reads: sixteen waves)
	int run = 0;
	if (kb + idb <= 64) {
		uint64_t *ka = (uint64_t *)tmp, *kbuf = ka + na;                        // the two halves of the 16-byte-per-anchor scratch
		const uint64_t m0 = b0 >= 32 ? 0xffffffffull : (1ull << b0) - 1, m1 = (1ull << b1) - 1;
		const bool fused = kb <= 32;                                            // up to four passes: their histograms are counted right here
		if (fused) { for (int d = lane; d < 4 * 256; d += 64) s_hist[d] = 0; __syncthreads(); }
		for (int i = lane; i < na; i += 64) {
			const uint64_t x = un[i].x;
			const uint64_t sq = (x & m0) | (((x >> 32) & m1) << b0) | (bs ? (x >> 63) << (b0 + b1) : 0);   // order preserving: the dropped bits are constant
			ka[i] = sq << idb | (uint64_t)i;
			if (fused)
				for (int pass = 0; 8 * pass < kb; ++pass) atomicAdd(&s_hist[256 * pass + ((int)(sq >> (8 * pass)) & 255)], 1);   // the top digit may be partial: its high bits are zero
		}
		__syncthreads();
		const uint64_t *ks = wave_sort_keys(ka, kbuf, na, idb, kb, lane, s_cnt, fused ? s_hist : nullptr);
		const uint64_t idm = (1ull << idb) - 1;
		// gather, and tiecnt[i] = number of positions j < i with x[j] == x[j+1]: tells in O(1) whether a range of positions (= a bucket
		// of the reference's sort, before and after it) holds equal keys
		for (int i0 = 0; i0 < na; i0 += 64) {
			const int i = i0 + lane;
			const uint64_t k = i < na ? ks[i] : 0, kn = i + 1 < na ? ks[i + 1] : ~0ull;
			if (i < na) { out[i] = un[(int)(k & idm)]; srt[i] = (uint32_t)(k & idm); }   // srt: which anchor of the unsorted array stands here (seed_ties reorders equal x by it)
			const int flag = (i + 1 < na && (k >> idb) == (kn >> idb)) ? 1 : 0;
			const int incl = wave_incl_scan(flag, lane);
			if (i < na) tiecnt[i] = run + incl - flag;
			run += __shfl(incl, 63);
		}
	} else {
		wave_sort_anchors(un, tmp, out, na, lane, s_cnt);
		for (int i0 = 0; i0 < na; i0 += 64) {
			const int i = i0 + lane;
			const int flag = (i + 1 < na && out[i].x == out[i + 1].x) ? 1 : 0;
			const int incl = wave_incl_scan(flag, lane);
			if (i < na) tiecnt[i] = run + incl - flag;
			run += __shfl(incl, 63);
		}
	}
	if (lane == 0) A.has_ties[read] = run > 0 ? (kb + idb <= 64 ? 1 : 2) : 0;       // 2: sorted as whole anchors, no srt[]
}

// ---- the key sort of wave_sort_keys on the NW waves of a workgroup (for ONE long read whose sort a batch would otherwise wait for): per step
// of 64 * NW records every wave ranks its own 64 (peers by ballots) and publishes its per-digit counts; a record's place is the digit's
// cursor + the counts of the waves before its own + its rank inside the wave (stable).  s_cnt: 256 ints, s_w: NW * 256 ints of LDS.
template <int NW>
__device__ uint64_t *block_sort_keys(uint64_t *a, uint64_t *b, int n, int bit_lo, int n_bits, int tid, int *s_cnt, int *s_w)
{
	const int lane = tid & 63, wave = tid >> 6;
	uint64_t *from = a, *to = b;
	for (int shift = bit_lo; shift < bit_lo + n_bits; shift += 8) {
		const int mask = bit_lo + n_bits - shift >= 8 ? 255 : (1 << (bit_lo + n_bits - shift)) - 1;
		for (int d = tid; d < 256; d += 64 * NW) s_cnt[d] = 0;
		for (int d = tid; d < 256 * NW; d += 64 * NW) s_w[d] = 0;
		__syncthreads();
		for (int i = tid; i < n; i += 64 * NW) atomicAdd(&s_cnt[(int)(from[i] >> shift) & mask], 1);
		__syncthreads();
		if (wave == 0) {
			int h[4], sum = 0;
#pragma unroll
			for (int u = 0; u < 4; ++u) { h[u] = s_cnt[4 * lane + u]; sum += h[u]; }
			int at = wave_incl_scan(sum, lane) - sum;
#pragma unroll
			for (int u = 0; u < 4; ++u) { s_cnt[4 * lane + u] = at; at += h[u]; }
		}
		__syncthreads();
		for (int i0 = 0; i0 < n; i0 += 64 * NW) {
			const int i = i0 + tid;
			const bool valid = i < n;
			const uint64_t k = valid ? from[i] : 0;
			const int d = (int)(k >> shift) & mask;
			uint64_t peers = __ballot(valid);
#pragma unroll
			for (int bb = 0; bb < 8; ++bb) {
				const uint64_t bal = __ballot((d >> bb) & 1);
				peers &= ((d >> bb) & 1) ? bal : ~bal;
			}
			const int rank = lanes_before(peers);
			if (valid && rank == 0) s_w[256 * wave + d] = __popcll(peers);
			__syncthreads();
			if (valid) {
				int before = 0;
				for (int w = 0; w < wave; ++w) before += s_w[256 * w + d];
				to[s_cnt[d] + before + rank] = k;
			}
			__syncthreads();
			if (valid && rank == 0) { atomicAdd(&s_cnt[d], __popcll(peers)); s_w[256 * wave + d] = 0; }
			__syncthreads();
		}
		{ uint64_t *t = from; from = to; to = t; }
	}
	__syncthreads();
	return from;
}

// ---- kernel 2b (round 6): seed_sort for LONG reads (capacity beyond SORT_LDS_CAP) on the sixteen waves of a workgroup --------------------------------------
// One wave sorting a read of 10^5 .. 10^6 anchors through global memory sets the time of a whole batch of long reads (BASELINE config 5: 256 reads of 10^6 anchors
// 55 ms, one read per CU and fifteen sixteenths of every CU idle).  The same keys ((differing bits of x, squeezed) << id bits | position), the same stable LSD
// passes through the read's 16 bytes of scratch per anchor -- block_sort_keys: per step of 1 024 records every wave ranks its own 64 and publishes its per-digit
// counts --, the same gather, srt[] and prefix count of equal neighbours as seed_sort leaves them (what the tie replay reads): a stable sort of distinct
// (key, position) pairs has one result, whoever computes it.  Reads whose keys do not fit 64 bits keep seed_sort's whole-anchor fall-back.
constexpr int SORT_MW_WAVES = 16;
template <int NW>
__global__ __launch_bounds__(64 * NW) void seed_sort_mw(SeedArgs A)
{
	constexpr int NT = 64 * NW;
	__shared__ int s_cnt[256], s_w[256 * NW], s_carry[NW + 1];
	const int read = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const ReadGeom g = read_geom(A, read);
	const int64_t a0 = g.a0;
	const int na = g.na;
	if (A.d_anchor_off[read + 1] - a0 <= SORT_LDS_CAP) return;                   // by capacity, as the launch order and seed_sort's test
	if (A.status[read] != 0 || na == 0) { if (tid == 0) A.has_ties[read] = 0; return; }
	const uint64_t diff = A.xdiff[read];
	const uint32_t dlo = (uint32_t)diff, dhi = (uint32_t)(diff >> 32) & 0x7fffffffu;
	const int b0 = dlo ? 32 - __clz((int)dlo) : 0, b1 = dhi ? 32 - __clz((int)dhi) : 0, bs = (int)(diff >> 63);   // position, target id, strand
	const int kb = b0 + b1 + bs, idb = na > 1 ? 32 - __clz(na - 1) : 1;
	if (kb + idb > 64) return;                                                   // seed_sort sorts the anchors themselves
	const ulonglong2 *un = A.unsorted + a0;
	ulonglong2 *tmp = A.scratch + a0, *out = A.d_anchors + g.o0;
	int32_t *tiecnt = A.tiecnt + a0;
	uint32_t *srt = A.tie_id + a0;
	uint64_t *ka = (uint64_t *)tmp, *kbuf = ka + na;
	const uint64_t m0 = b0 >= 32 ? 0xffffffffull : (1ull << b0) - 1, m1 = (1ull << b1) - 1;
	for (int i = tid; i < na; i += NT) {
		const uint64_t x = un[i].x;
		ka[i] = ((x & m0) | (((x >> 32) & m1) << b0) | (bs ? (x >> 63) << (b0 + b1) : 0)) << idb | (uint64_t)i;   // order preserving: the dropped bits are constant
	}
	__syncthreads();
	const uint64_t *ks = block_sort_keys<NW>(ka, kbuf, na, idb, kb, tid, s_cnt, s_w);
	const uint64_t idm = (1ull << idb) - 1;
	int run = 0;
	for (int i0 = 0; i0 < na; i0 += NT) {
		const int i = i0 + tid;
		const uint64_t k = i < na ? ks[i] : 0, kn = i + 1 < na ? ks[i + 1] : ~0ull;
		if (i < na) { out[i] = un[(int)(k & idm)]; srt[i] = (uint32_t)(k & idm); }
		const int flag = (i + 1 < na && (k >> idb) == (kn >> idb)) ? 1 : 0;
		const int incl = wave_incl_scan(flag, lane);
		if (lane == 63) s_carry[wave + 1] = incl;
		__syncthreads();
		int before = run;
#pragma unroll
		for (int w = 0; w < NW; ++w) { const int c = s_carry[w + 1]; before += w < wave ? c : 0; run += c; }
		if (i < na) tiecnt[i] = before + incl - flag;
		__syncthreads();
	}
	if (tid == 0) A.has_ties[read] = run > 0 ? 1 : 0;
}

// ---- kernel 3: replay of radix_sort_128x for reads with equal x (radix_replay.h) -----------------------------------
// The sequential part of the replay (the walk of replay_walk) reads digits only, so a read needs ONE byte of LDS per anchor (the index
// array, the permutation of a pass and the lists of the closed form live in global memory and are touched by all lanes); size classes keep
// the occupancy of the common (short) reads high: up to 5 120 anchors one wave replays a read, longer reads -- whose replay a batch waits
// for -- run the independent buckets of each level, the sweeps and the final order on two (up to 12 288 anchors) or eight waves of a
// workgroup, with up to 128 K digits in LDS (gfx950: 160 KB per workgroup).  CAP = 0: reads beyond that, digits in global memory.
constexpr int TIE_CAP0 = 2560, TIE_CAP1 = 5120, TIE_CAP2 = 12288, TIE_CAP3 = 65536, TIE_CAP4 = 131072;
constexpr int TIE_MW_WAVES = 8;
#ifndef MM2C_TIE_SW
#define MM2C_TIE_SW 2
#endif
constexpr int TIE_SW = MM2C_TIE_SW;            // waves per read of the 5 121 .. 12 288 class (its reads are few and long enough to set the time of a chunk; the classes below it are many reads: one wave each)

template <int CAP, int PREV, int NW>
__global__ __launch_bounds__(64 * NW, NW > 2 ? 1 : CAP > TIE_CAP1 ? 2 : CAP > TIE_CAP0 ? 4 : 5) void seed_ties(SeedArgs A)   // one-wave classes: five waves per SIMD (102 VGPRs), as many reads in flight as their LDS allows
{
	__shared__ __attribute__((aligned(8))) uint8_t s_dg[CAP + 8];            // (the walk looks one byte beyond the digit it takes)
	constexpr int SC = 576;                                                      // ints of replay tables per wave
	__shared__ __attribute__((aligned(8))) int s_cur[SC * NW];
	__shared__ int s_lo[257 * NW], s_n[2];
	__shared__ __attribute__((aligned(8))) int s_wbuf[CAP ? 2 : 512 * NW];          // digits in memory: the walk's results of 256 steps, stored by all lanes together (radix_replay.h)
	const int read = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	if (A.status[read] != 0 || A.has_ties[read] == 0) return;
	const int tid = (int)threadIdx.x;
	const ReadGeom g = read_geom(A, read);
	const int64_t a0 = g.a0;
	const int na = g.na;
	if (na <= 64 || (CAP ? na <= PREV || na > CAP || na > A.tie_global_above : na <= A.tie_global_above)) return;   // <= 64: insertion sort only, stable (ksort.h:149); the bound between
	                                                                             // the LDS classes and the one with the digits in memory moves with the batch (launch_seed_hits' caller)
	if (A.debug_cut == 1) return;
	ulonglong2 *un = A.unsorted + a0, *tmp = A.scratch + a0, *out = A.d_anchors + g.o0;
	// The replay leaves an arrangement id[] (position -> anchor of the unsorted array).  The reference's array is the stable sort of that
	// arrangement by x; seed_sort already left the anchors sorted by x with equal x in the order of the unsorted array and srt[] = which anchor
	// stands where, so only the runs of equal x have to be put in the order of the arrangement (has_ties == 1; == 2: no srt[], full sort below).
	const bool fix = A.has_ties[read] == 1;
	int32_t *w = (int32_t *)tmp;                                                 // the read's 16 bytes of scratch per anchor: four int arrays
	uint32_t *id = fix ? (uint32_t *)(w + 3 * (int64_t)na) : A.tie_id + a0;
	uint8_t *dg = CAP ? s_dg : A.big_dg + a0 + read;                             // one spare byte per read
	int32_t *lists = A.stack + 4 * (a0 / 64 + 2 * (int64_t)read);               // > 64 anchors per pending bucket: na / 64 + 2 entries per list suffice
	int32_t *moved = w, *fa = w + na, *fb = fa + na;
	replay_levels<uint32_t, true, CAP != 0, NW>((const uint64_t *)un, 2, (const uint64_t *)out, 2, A.tiecnt + a0, na, id, dg, lists, lists + 2 * (na / 64 + 2),
	                                             moved, fa, fb, tid, s_cur, s_lo, s_n, CAP ? (int *)nullptr : s_wbuf);
	if (A.debug_cut == 2) return;
	if (fix) {
		// place[i] = where the anchor at sorted position i stands in the arrangement; inside a run of equal x the anchor with the k-th smallest
		// place goes to the k-th position of the run.  Runs are short (one reference position hit by a few query minimizers): every position
		// of a run counts the smaller places of its run by itself, from a window of place[] and of the prefix counts of equal neighbours that
		// its wave keeps in LDS (STEP positions and RUN_MAX - 1 on either side; the digits and tables of the replay have served: their space
		// takes the windows).  A run of RUN_MAX positions or more: the full sort below.
		constexpr int RUN_MAX = 224, MARGIN = RUN_MAX - 1;
		constexpr int ROOM_P = CAP ? (CAP / NW) / 4 : 576, ROOM_T = CAP ? 2 * 576 : 2 * 257;   // entries the two windows may take per wave
		constexpr int STEP_P = (ROOM_P - 2 * MARGIN) / 64 * 64, STEP_T = (ROOM_T - 2 * MARGIN) / 64 * 64;
		constexpr int STEP = STEP_P < STEP_T ? (STEP_P < 512 ? STEP_P : 512) : (STEP_T < 512 ? STEP_T : 512), KP = STEP / 64, WIN = STEP + 2 * MARGIN;
		static_assert(STEP >= 64 && WIN <= ROOM_P && WIN <= ROOM_T, "the windows fit the space of the replay");
		const uint32_t *srt = A.tie_id + a0;
		const int32_t *tiecnt = A.tiecnt + a0;
		int32_t *inv = w, *place = w + na;
		const int lane = tid & 63, wave = tid >> 6;
		int *winp = CAP ? (int *)(s_dg + (CAP / NW) * wave) : s_cur + SC * wave;
		uint16_t *wint = CAP ? (uint16_t *)(s_cur + SC * wave) : (uint16_t *)(s_lo + 257 * wave);
		for (int q = tid; q < na; q += 64 * NW) inv[id[q]] = q;
		__syncthreads();
		for (int i = tid; i < na; i += 64 * NW) place[i] = inv[srt[i]];
		__syncthreads();
		if (A.debug_cut == 3) return;
		bool over = false;
		for (int c0 = STEP * wave; c0 < na; c0 += STEP * NW) {
			bool tied[KP], any = false;
#pragma unroll
			for (int k = 0; k < KP; ++k) {
				const int i = c0 + 64 * k + lane, ic = i < na ? i : na - 1;
				const int t0 = tiecnt[ic], t1 = tiecnt[ic + 1 < na ? ic + 1 : ic], tm = tiecnt[ic > 0 ? ic - 1 : 0];
				tied[k] = i < na && (t1 != t0 || t0 != tm);                        // tiecnt[j + 1] - tiecnt[j] = 1: x[j] == x[j + 1]
				any |= tied[k];
			}
			if (!__ballot(any)) continue;
			const int base = c0 - MARGIN, tb = tiecnt[base > 0 ? base : 0];
			rp_wave_sync();                                                        // the windows of the step before have been read
			{
				constexpr int NWIN = (WIN + 63) / 64;
				int wp[NWIN], wt[NWIN];                                              // all loads of the two windows in flight together
#pragma unroll
				for (int k = 0; k < NWIN; ++k) {
					const int idx = base + 64 * k + lane, ci = idx < 0 ? 0 : idx >= na ? na - 1 : idx;
					wp[k] = place[ci]; wt[k] = tiecnt[ci];
				}
#pragma unroll
				for (int k = 0; k < NWIN; ++k)
					if (64 * k + lane < WIN) { winp[64 * k + lane] = wp[k]; wint[64 * k + lane] = (uint16_t)(wt[k] - tb); }
			}
			rp_wave_sync();
			int dst[KP], rec[KP];
#pragma unroll
			for (int k = 0; k < KP; ++k) {
				dst[k] = -1; rec[k] = 0;
				if (!tied[k]) continue;
				// the run [s, e] of i, in window positions: every neighbour pair between two positions is equal when their prefix counts differ by their distance
				const int iw = 64 * k + lane + MARGIN, ti = wint[iw];
				int sw = iw, ew = iw;
				for (int step = 128; step > 0; step >>= 1) {
					const int a = sw - step, b = ew + step;
					if (a >= iw - MARGIN && base + a >= 0 && ti - (int)wint[a] == iw - a) sw = a;
					if (b <= iw + MARGIN && base + b < na && (int)wint[b] - ti == b - iw) ew = b;
				}
				if (ew - sw >= MARGIN) { over = true; continue; }                    // (the first position of a run that long sees it)
				const int mine = winp[iw];
				int rank = 0;
#pragma unroll 4
				for (int j = sw; j <= ew; ++j) rank += winp[j] < mine;
				dst[k] = base + sw + rank; rec[k] = (int)srt[c0 + 64 * k + lane];
			}
			ulonglong2 an[KP];
#pragma unroll
			for (int k = 0; k < KP; ++k) { an[k] = ulonglong2{0, 0}; if (dst[k] >= 0) an[k] = un[rec[k]]; }
#pragma unroll
			for (int k = 0; k < KP; ++k) if (dst[k] >= 0) out[dst[k]] = an[k];
		}
		if (!__syncthreads_or(over)) return;
		for (int q = tid; q < na; q += 64 * NW) A.tie_id[a0 + q] = id[q];        // srt[] has served: the arrangement moves there, the scratch is needed for the keys
		__syncthreads();
		id = A.tie_id + a0;
	}
	// the final stable sort of the replayed arrangement: keys = (differing bits of x, squeezed) << id bits | position in the arrangement, as in
	// seed_sort (the scratch of the replay is free now)
	const uint64_t diff = A.xdiff[read];
	const uint32_t dlo = (uint32_t)diff, dhi = (uint32_t)(diff >> 32) & 0x7fffffffu;
	const int b0 = dlo ? 32 - __clz((int)dlo) : 0, b1 = dhi ? 32 - __clz((int)dhi) : 0, bs = (int)(diff >> 63);
	const int kb = b0 + b1 + bs, idb = 32 - __clz(na - 1);
	if (kb + idb <= 64) {
		uint64_t *ka = (uint64_t *)tmp, *kbuf = ka + na;
		const uint64_t m0 = b0 >= 32 ? 0xffffffffull : (1ull << b0) - 1, m1 = (1ull << b1) - 1;
		for (int q = tid; q < na; q += 64 * NW) {
			const uint64_t x = un[id[q]].x;
			ka[q] = ((x & m0) | (((x >> 32) & m1) << b0) | (bs ? (x >> 63) << (b0 + b1) : 0)) << idb | (uint64_t)q;
		}
		__syncthreads();
		const uint64_t *ks = NW == 1 ? wave_sort_keys(ka, kbuf, na, idb, kb, tid, s_cur)
		                             : block_sort_keys<NW>(ka, kbuf, na, idb, kb, tid, s_cur, s_cur + 256);   // the tables of the replay are free now
		const uint64_t idm = (1ull << idb) - 1;
		for (int i = tid; i < na; i += 64 * NW) out[i] = un[id[(int)(ks[i] & idm)]];
		return;
	}
	for (int i = tid; i < na; i += 64 * NW) tmp[i] = un[id[i]];
	__syncthreads();
	if (tid >= 64) return;                                                       // the anchors themselves: a one-wave routine
	wave_sort_anchors(tmp, un, out, na, tid, s_cur);
}

// ---- kernel 3': the order collect_seed_hits_heap leaves (map.c:149-213; MM_F_HEAP_SORT) -----------------------------------------------------
// That function merges the matches' hit lists through a binary heap keyed on the hit alone: the same anchors in the same ascending order of x as the
// radix-sorted list, except among anchors with EQUAL x, which come out in whatever order the heap pops equal keys -- a function of the heap's whole
// history, so for the reads that have such anchors (has_ties, from seed_sort) the heap is replayed operation by operation (ksort.h:43-60 with heap_lt,
// map.c:80) by one lane, heap in LDS for up to HEAP_CAP matches (else in the read's scratch).  Forward-strand anchors are written in pop order, reverse-strand
// ones collected in pop order and appended (what map.c:201-210 leaves).  Every other read already has the order the heap gives: it is unique.
constexpr int HEAP_CAP = 2048;

__global__ __launch_bounds__(64) void seed_heap(SeedArgs A)
{
	__shared__ uint64_t s_hx[HEAP_CAP];
	__shared__ uint32_t s_hm[HEAP_CAP], s_hk[HEAP_CAP];
	__shared__ int s_nfor, s_nrev;
	const int read = A.d_order ? A.d_order[blockIdx.x] : (int)blockIdx.x;
	if (A.status[read] != 0 || A.has_ties[read] == 0) return;
	const int lane = (int)threadIdx.x;
	const ReadGeom g = read_geom(A, read);
	const int64_t m0 = A.d_match_off[read];
	const int nm = (int)(A.d_match_off[read + 1] - m0);
	const Match *m = A.d_matches + m0;
	const int qlen = A.d_qlen[read];
	const int q_lo = A.d_q_lo ? A.d_q_lo[read] : 0, q_eq = A.d_q_eq ? A.d_q_eq[read] : 0;
	ulonglong2 *out = A.d_anchors + g.o0, *rev = A.unsorted + g.a0;              // (the expansion-order copy is not needed any more)
	// heap entries: the hit, the match, the position in the match's list
	uint64_t *hx = s_hx; uint32_t *hm = s_hm, *hk = s_hk;
	if (nm > HEAP_CAP) { hx = (uint64_t *)(A.scratch + g.a0); hm = (uint32_t *)(hx + nm); hk = hm + nm; }   // 16 bytes per match <= 16 bytes per anchor: matches without a hit take no room ...
	if (lane == 0) {
		int hs = 0;
		if (nm > HEAP_CAP) { int live = 0; for (int i = 0; i < nm; ++i) live += m[i].n > 0; hm = (uint32_t *)(hx + live); hk = hm + live; }   // ... because the arrays are sized by the live ones
		for (int i = 0; i < nm; ++i)                                            // map.c:162-168
			if (m[i].n > 0) { hx[hs] = A.d_hits[m[i].cr_off]; hm[hs] = (uint32_t)i; hk[hs] = 0; ++hs; }
		auto down = [&](int i, int n) {                                         // ks_heapdown, ksort.h:43-53, heap_lt(a, b) = a.x > b.x
			const uint64_t tx = hx[i]; const uint32_t tm = hm[i], tk = hk[i];
			int k = i;
			while ((k = (k << 1) + 1) < n) {
				if (k != n - 1 && hx[k] > hx[k + 1]) ++k;
				if (hx[k] > tx) break;
				hx[i] = hx[k]; hm[i] = hm[k]; hk[i] = hk[k]; i = k;
			}
			hx[i] = tx; hm[i] = tm; hk[i] = tk;
		};
		for (int i = (hs >> 1) - 1; i >= 0; --i) down(i, hs);                   // ks_heapmake, ksort.h:54-59
		int n_for = 0, n_rev = 0;
		while (hs > 0) {                                                        // map.c:170-198
			const Match q = m[hm[0]];
			ulonglong2 a;
			if (hit_anchor(A, hx[0], q.q_pos, q.q_span, q.seg_tandem, qlen, q_lo, q_eq, a)) {
				if ((a.x >> 63) == 0) out[n_for++] = a; else rev[n_rev++] = a;
			}
			if (hk[0] + 1 < q.n) { ++hk[0]; hx[0] = A.d_hits[q.cr_off + hk[0]]; }
			else { --hs; hx[0] = hx[hs]; hm[0] = hm[hs]; hk[0] = hk[hs]; }
			if (hs > 0) down(0, hs);
		}
		s_nfor = n_for; s_nrev = n_rev;
	}
	__syncthreads();
	const int n_for = s_nfor, n_rev = s_nrev;
	for (int i = lane; i < n_rev; i += 64) out[n_for + i] = rev[i];
	if (lane == 0 && n_for + n_rev != g.na) A.status[read] = 1;                  // cannot happen: the expansion kept exactly these hits
}

} // namespace

int seed_tie_lds_max() { return TIE_CAP4; }
int seed_tie_mid_lower() { return TIE_CAP2; }
static int64_t lower_of_class(int c) { static const int64_t lower[6] = { 64, TIE_CAP0, TIE_CAP1, TIE_CAP2, TIE_CAP3, TIE_CAP4 }; return lower[c]; }
int seed_sort_lds_cap() { return SORT_LDS_CAP; }
int seed_sort_lds_cap0() { return SORT_LDS_CAP0; }
const int64_t *seed_tie_class_lower()
{
	static const int64_t lower[6] = { 64, TIE_CAP0, TIE_CAP1, TIE_CAP2, TIE_CAP3, TIE_CAP4 };
	return lower;
}

// The tie replay of one read is a long dependent chain on one wave, so each size class ends in a tail of a few long reads; the classes
// handle disjoint reads and run side by side on the caller's stream and three helper streams (fork after the sort, join at the end).
hipError_t launch_seed_hits(const SeedArgs &A, hipStream_t st, int *n_launches, hipStream_t *aux, hipEvent_t *ev)
{
	if (A.n_reads <= 0) return hipSuccess;
	const unsigned nr = (unsigned)A.n_reads;
	hipError_t e;
	hipLaunchKernelGGL(seed_expand, dim3(nr), dim3(64), 0, st, A);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	if (A.mw_sort && !A.d_count && (A.d_order ? A.n_sort_huge : (int64_t)nr) > 0) {   // long reads: sixteen waves each (the first workgroups of the launch order)
		hipLaunchKernelGGL((seed_expand_mw<EXPAND_MW_WAVES>), dim3(A.d_order ? (unsigned)A.n_sort_huge : nr), dim3(64 * EXPAND_MW_WAVES), 0, st, A);
		if ((e = hipGetLastError()) != hipSuccess) return e;
		if (n_launches) ++*n_launches;
	}
	if (A.d_count) {                                                             // skip_seed in force: the reads keep fewer anchors than they have hits
		hipLaunchKernelGGL(seed_offsets, dim3(1), dim3(1024), 0, st, A);
		if ((e = hipGetLastError()) != hipSuccess) return e;
		if (n_launches) ++*n_launches;
	}
	if (A.lds_sort) {                                                            // the launch order is by capacity, descending: each class is a range of it
		const unsigned huge = A.d_order ? (unsigned)A.n_sort_huge : 0u, big = A.d_order ? (unsigned)A.n_sort_big : 0u;
		if (big < nr) {
			hipLaunchKernelGGL((seed_sort_lds<SORT_LDS_CAP0, 0, 4>), dim3(nr - big), dim3(256), 0, st, A, (int)big);
			if ((e = hipGetLastError()) != hipSuccess) return e;
			if (n_launches) ++*n_launches;
		}
		const unsigned end1 = A.d_order ? big : nr;
		if (huge < end1) {
			hipLaunchKernelGGL((seed_sort_lds<SORT_LDS_CAP, SORT_LDS_CAP0, 8>), dim3(end1 - huge), dim3(512), 0, st, A, (int)huge);
			if ((e = hipGetLastError()) != hipSuccess) return e;
			if (n_launches) ++*n_launches;
		}
	}
	if (A.mw_sort && (A.d_order ? A.n_sort_huge : (int64_t)nr) > 0) {            // long reads: sixteen waves each (the first workgroups of the launch order)
		hipLaunchKernelGGL((seed_sort_mw<SORT_MW_WAVES>), dim3(A.d_order ? (unsigned)A.n_sort_huge : nr), dim3(64 * SORT_MW_WAVES), 0, st, A);
		if ((e = hipGetLastError()) != hipSuccess) return e;
		if (n_launches) ++*n_launches;
	}
	hipLaunchKernelGGL(seed_sort, dim3(nr), dim3(64), 0, st, A);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	if (n_launches) *n_launches += 2;
	if (A.heap_order) {                                                          // collect_seed_hits_heap: its own order among equal x instead of the radix sort's
		hipLaunchKernelGGL(seed_heap, dim3(nr), dim3(64), 0, st, A);
		if (n_launches) ++*n_launches;
		return hipGetLastError();
	}
	// The grid is in order of decreasing read capacity: class c holds reads of more than its PREV anchors, i.e. the first n_above[c] workgroups
	// at most (a read keeps at most its capacity); those that belong to a longer class exit at once.  The classes of the longest reads go first, each on
	// a helper stream; the shortest class stays on the caller's stream.
	const unsigned grid[6] = { A.d_order ? (unsigned)A.n_above[0] : nr, A.d_order ? (unsigned)A.n_above[1] : nr, A.d_order ? (unsigned)A.n_above[2] : nr,
	                           A.d_order ? (unsigned)A.n_above[3] : nr, A.d_order ? (unsigned)A.n_above[4] : nr, A.d_order ? (unsigned)A.n_above[5] : nr };
	if ((e = hipEventRecord(ev[0], st)) != hipSuccess) return e;                 // fork
	bool used[3] = { false, false, false };
	int helper = 0;
	for (int c = 5; c >= 0; --c) {
		if (grid[c] == 0 || (c == 5 && !A.big_dg)) continue;
		if (c < 5 && lower_of_class(c) >= A.tie_global_above) continue;             // the whole class has gone to the kernel with the digits in memory
		hipStream_t s = st;
		if (c != 0 && aux) {
			const int h = helper++ % 3;
			s = aux[h];
			if (!used[h] && (e = hipStreamWaitEvent(s, ev[0], 0)) != hipSuccess) return e;
			used[h] = true;
		}
		switch (c) {
		case 5:
			// digits in memory.  The buckets of a level are independent walks (only a level of one bucket stays on one wave), so a read can use several waves -- at the price of
			// fewer reads in flight per CU: few such reads (a batch of reads of 10^6 anchors: fewer than the GPU has CUs) eight waves each, many one wave each, eight reads per CU
			{
				// waves per read by the number of reads in the class (ms for 1, 2, 4, 8 waves, profiles/r6_long_reads.md): 4 096 reads of 5e4 anchors 61 / 66 / 84 / 91; 2 048 of 1e5
				// 88 / 73 / 87 / 99; 1 533 of 1.5e5 125 / 102 / 113 / 120; 1 020 of 3e5 224 / 195 / 149 / 176; 510 of 5e5 243 / 209 / 175 / 168; 255 of 1e6 414 / 387 / 249 / 226-263
				const int w = A.tie_global_waves ? A.tie_global_waves : grid[c] <= (unsigned)A.tie_global_mw_below ? TIE_MW_WAVES : grid[c] <= 1280u ? 4 : grid[c] <= 3000u ? 2 : 1;
				if (w >= 8) hipLaunchKernelGGL((seed_ties<0, TIE_CAP4, TIE_MW_WAVES>), dim3(grid[c]), dim3(64 * TIE_MW_WAVES), 0, s, A);
				else if (w == 4) hipLaunchKernelGGL((seed_ties<0, TIE_CAP4, 4>), dim3(grid[c]), dim3(256), 0, s, A);
				else if (w == 2) hipLaunchKernelGGL((seed_ties<0, TIE_CAP4, 2>), dim3(grid[c]), dim3(128), 0, s, A);
				else hipLaunchKernelGGL((seed_ties<0, TIE_CAP4, 1>), dim3(grid[c]), dim3(64), 0, s, A);
			}
			break;
		case 4: hipLaunchKernelGGL((seed_ties<TIE_CAP4, TIE_CAP3, TIE_MW_WAVES>), dim3(grid[c]), dim3(64 * TIE_MW_WAVES), 0, s, A); break;
		case 3: hipLaunchKernelGGL((seed_ties<TIE_CAP3, TIE_CAP2, TIE_MW_WAVES>), dim3(grid[c]), dim3(64 * TIE_MW_WAVES), 0, s, A); break;
		case 2: hipLaunchKernelGGL((seed_ties<TIE_CAP2, TIE_CAP1, TIE_SW>), dim3(grid[c]), dim3(64 * TIE_SW), 0, s, A); break;
		case 1: hipLaunchKernelGGL((seed_ties<TIE_CAP1, TIE_CAP0, 1>), dim3(grid[c]), dim3(64), 0, s, A); break;
		default: hipLaunchKernelGGL((seed_ties<TIE_CAP0, 0, 1>), dim3(grid[c]), dim3(64), 0, s, A); break;
		}
		if ((e = hipGetLastError()) != hipSuccess) return e;
		if (n_launches) ++*n_launches;
	}
	for (int h = 0; h < 3; ++h) {                                               // join
		if (!used[h]) continue;
		if ((e = hipEventRecord(ev[1 + h], aux[h])) != hipSuccess) return e;
		if ((e = hipStreamWaitEvent(st, ev[1 + h], 0)) != hipSuccess) return e;
	}
	return hipSuccess;
}

hipError_t warm_seed_kernels()
{
	hipFuncAttributes at;
	return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&seed_offsets));
}

} // namespace mm2c
