// chain_dp_tile.h -- the chaining DP kernel of the library (second generation), included by chain_kernel.hip.
//
// Same contract as before: f[i] / p[i] of chain.c:184-238 for every anchor of every task, bit for bit, incl. the max_skip early exit
// (chain.c:226-233); with max_skip = INT_MAX, max_iter = 1024, one q_span it is what device/minimap2_opencl.cl:24-172 computes.
//
// What the measurements of round 2 say about gfx950 (tools/ubench/issue_rate.hip, profiles/r2_issue_rate.md): per SIMD and ns a wave64
// stream gets ~0.84 plain VALU instructions, but only ~0.55 of anything that touches the scalar side -- SALU ops (ONE scalar unit per CU),
// v_cmp into an SGPR pair, v_readlane / v_writelane, v_cndmask with an SGPR mask, DPP ops -- 0.29 ds_read_b64 and 0.10 ds_bpermute.  The
// first kernel spent 117 VALU + 113 SALU instructions per anchor and was bound by the SCALAR unit (its time is 0.62 ms per SALU
// instruction per anchor on the headline batch, whatever else changes).  Hence this layout:
//   * one 64-lane wave = one workgroup = one task, as before (the recurrence is sequential in i).
//   * the look-back window of anchor i is scanned nearest-first in TILE-ALIGNED chunks: tile T = anchors [64T, 64T+64), lane L holds
//     anchor 64T + 63 - L, so ascending lane = descending j = the reference's scan order inside every chunk, and the chunks of one
//     anchor are: the part of its own tile before it (from registers: x, q, and f / p that v_writelane puts into the anchor's lane),
//     then tiles T-1, T-2, ... from an LDS ring that is written once per 64 anchors: x / q of NX tiles (all the filters need), f / p of
//     the NF nearest; deeper f / p and anything beyond NX tiles come from L2 (the task's own earlier stores), for nonempty chunks only.
//     No per-lane ring arithmetic, no window shifting, no cross-lane data movement per anchor.
//   * the filters chain.c:202-205 of a chunk cost 5 plain VALU + one v_cmp: dr-1, dq-1, |dr-dq| (v_sad_u32), one saturating subtraction
//     against max_dq-1-bw, v_max_u32, compare with bw (valid when max_dq-1 >= bw >= 0: every preset); dr == 0 (equal x, chain.c:202) never
//     reaches the vector unit: equal x are neighbours in the sorted array, so the lanes to drop are a run that one ballot per TILE locates
//     and that only moves the start of the own-tile lane mask.
//   * chain.c's t[] stamps: one byte per anchor the ring reaches, value 1 + position in the tile (they only mean something during the scan of
//     the anchor that wrote them: the ring is wiped per tile), ds_write_b8 by p, ds_read_u8 by j; lanes that must not stamp write the slot of
//     anchor lo - 1, so the store needs no exec mask.
//   * the whole scan of the anchors of a tile -- per-anchor scalars, chunk loop, f / p fetch, stamps, score, the order-dependent fold (running
//     max; max_skip counter in closed form where the first surviving lane is the only new maximum, else prefix max by DPP and, if needed, a
//     max-plus prefix scan) -- is ONE hand-written instruction sequence (MM2C_SCAN_TILE_ASM below) for the variants that matter (max_skip
//     on, one segment, gap_scale 1 or the gap-cost table): the compiler's code for wave-uniform control flow (64-bit boolean masks, s_mov
//     phi chains) needs 2.5x the scalar instructions.  Two instantiations: `lean` for tiles in which no window reaches beyond the ring,
//     `far` for the others (stamps beyond the ring in global scratch, x / q requests one tile ahead).  Everything else (segments / cDNA,
//     gap_scale != 1 without the table, max_skip off, an equal-x run that reaches back into the tile before) goes through the C++
//     restatement of the same scan (scan_anchor); the segment / cDNA variant is by default run by the first-generation kernel
//     (chain_kernel.hip), which is faster for it.
#ifndef MM2C_CHAIN_DP_TILE_H
#define MM2C_CHAIN_DP_TILE_H
#ifndef MM2C_DEEP_PREFETCH
#define MM2C_DEEP_PREFETCH 0     // 1: the 32-bit ring forms of the hand-written loop request the deep f / p of the next tile ahead (MM2C_FG_W).  Measured in round 4 and
                                 // left off: slower on every stream (ava-ont mixed 90.3 -> 92.2 ms, dense ragged 85.2 -> 87.8, colinear 25.7 -> 26.1)
#endif
#include "chain_wave.h"

namespace mm2c {

// a - b, saturating at 0 (unsigned); both operands in VGPRs so that the instruction issues at the full VALU rate
__device__ __forceinline__ int usat_sub(int a, int b)
{
	int r;
	asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ int min3i(int a, int b, int c)
{
	int r;
	asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
__device__ __forceinline__ int add3i(int a, int b, int c)
{
	int r;
	asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
// 16-bit LDS store for the lanes of mask m only (exec is all ones everywhere in this kernel: control flow is wave-uniform)
__device__ __forceinline__ void lds_store_b16_masked(mask_t m, int byte_addr, int value)
{
	asm volatile("s_mov_b64 exec, %0\n\tds_write_b8 %1, %2\n\ts_mov_b64 exec, -1" : : "s"(m), "v"(byte_addr), "v"(value) : "memory");
}
// put two wave-uniform values into lane `l` (wave-uniform) of two registers; the lane select goes through M0 because a VALU
// instruction of gfx9 reads at most one SGPR
__device__ __forceinline__ void write_lane2(int &v0, int &v1, int a0, int a1, int l)
{
	asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0"
	    : "+v"(v0), "+v"(v1) : "s"(l), "s"(a0), "s"(a1));   // M0 is scratch for the compiler too: it loads it right before each of its own uses
}

// ---------------------------------------------------------------- LDS layout of one wave (byte offsets into the kernel's only LDS object)
// rings are indexed by the anchor itself: anchor j of the task lives in the 8-byte slot (j mod 64 NX) of the x / q ring as the pair {x, q} (NX tiles,
// a power of two, the tile in progress included: 64 (NX - 1) anchors before it are reachable), in the 8-byte slot (j mod 64 NF) of the f / p ring as
// the pair {f - FBIAS, p} (the NF tiles before the one in progress: its own f / p are in registers until it is finished) and at byte (j mod SN) of
// the stamp ring.  Pairs: one ds_read_b64 per tile instead of two ds_read_b32 (an LDS instruction costs the loop about as much as a VALU one).
// C16 (compact x / q ring, round 3): the slot holds the LOW 16 BITS of x and of q in one dword.  Inside the window of anchor i every
// x_i - x_j lies in [0, max_dist_x] (that is what the window is, chain.c:192), so with max_dist_x < 2^16 the difference of the low halves
// mod 2^16 IS dr; q_i - q_j is not bounded by the window, but a pair only passes with 0 < dq <= max_dq (chain.c:202-203), and when the
// task's q values span at most 65535 - max_dq no other difference can alias into that range mod 2^16 -- the prepass checks that per task
// (chain_window_start: bit 1 of the task's class sends it to the 32-bit instantiation).  Half the LDS per ring anchor: the ring of 16 tiles
// costs what the ring of 8 did.
// RING 2 (q24 ring, round 5): the same 4-byte slot {x & 0xffff, q << 16} plus ONE MORE BYTE per ring anchor, bits 16-23 of q, in a byte ring of its own (QH): 5 bytes per
// ring anchor instead of 8.  dr from the low halves as above; q of the ring anchor is put together from the slot's high half and its byte (one v_perm_b32) and subtracted
// in 32 bits: exact for every task whose q values are below 2^24 (reads of up to 16.7 Mb: the prepass clears the long-ring class of any other task).  It is the form of
// the LONG ring (ring-size class 1: tasks whose scans leave the short ring -- long noisy reads, all-vs-all overlaps): 16 tiles in 7 KB instead of 10 KB, 22 waves per CU
// instead of 16, which is what bounded those streams (16 384 reads of 20 000 anchors were exactly four rounds of 16 x 256 wave slots).
template <int NX, int NF, bool GEN, bool TAB, int RING = 0>
struct Lds {
	static constexpr int SN = 64 * NX;           // anchors with a stamp slot = anchors in the x / q ring
	static constexpr int XS = RING ? 4 : 8;      // bytes of an x / q slot
	static constexpr int TILE = 64 * XS;         // ... of a tile in the x / q ring
	static constexpr int RB = SN * XS;           // ... of the x / q ring
	static constexpr int XQ = 0, FP = RB, ST = FP + NF * 512, QH = ST + SN, GAP = QH + (RING == 2 ? SN : 0),
	                     G = GAP + (TAB ? 1024 : 0), BYTES = G + (GEN ? NX * 64 : 0);
	static constexpr int FMASK = NF * 512 - 1;   // slot of a tile in the f / p ring = its x / q slot mod NF (NF a power of two dividing NX)
	static constexpr int SBITS = __builtin_ctz(SN);
};

constexpr int FBIAS = 14;   // min(dq, dr, span) - gap cost = min3(dq - 1, dr - 1, span - 1) - linear part + (clz(dd | 1) >> 1) - 14 (chain.c:207-209,218)

// what the chunks of one anchor share (wave-uniform unless noted)
struct AnchorCtx {
	int xi1, qi1;            // x_i - 1, q_i - 1 (so that dr - 1 and dq - 1 come out of one subtraction each)
	int span_i, seg_i;
	int lo;                  // start of the window (chain.c:192-193)
	int stamp, s16;          // i + 1 (global scratch t[]), 1 + i % 1024 (LDS stamp ring)
	int stamp_lo;            // oldest anchor whose stamp slot is in the LDS ring
	int far_mode;            // the window reaches beyond the LDS ring (FAR variants)
	float avg;
	// per-lane values
	int mdq1_v, bw_v, span1_v, s16_v, rl;   // max_dq - 1, bw, span_i - 1, s16, 63 - lane
};

struct TileMem {
	char *lds;               // the wave's LDS (layout: Lds<>)
	const uint4 *a; const int32_t *f, *p; int32_t *t;   // global arrays of the task
	int pbase;
};

// ---------------------------------------------------------------- the filters chain.c:202-205 of one chunk as a lane mask
template <bool GEN, bool FULL, bool DR0>
__device__ __forceinline__ mask_t chunk_filter(const KParams &P, const AnchorCtx &X, mask_t in_w, int dr1, int dq1, int dd, int gj, mask_t &same)
{
	mask_t valid;
	if (!GEN) {
		// same segment, genomic: dr != 0 (the caller's masks), 0 < dq <= min(max_dist_y, max_dist_x), dd <= bw
		const int viol = usat_sub(dq1, X.mdq1_v) | usat_sub(dd, X.bw_v);
		valid = BALLOT(viol == 0);
		if (!FULL) valid &= in_w;
		if (DR0) valid &= BALLOT(dr1 != -1);            // an equal-x run that reaches beyond the own tile (rare)
	} else {
		same = BALLOT(gj == X.seg_i);
		valid = pair_filter<true>(P, FULL ? ~0ull : in_w, dr1 + 1, dq1 + 1, dd, same);
	}
	return valid;
}

// ---------------------------------------------------------------- what follows the filters for a chunk with at least one surviving lane
// stamps (chain.c:229,233), score (chain.c:207-220), then the order-dependent fold.  base = first anchor of the tile the chunk belongs to.
// FAR: stamp targets may lie beyond the LDS stamp ring (X.far_mode, wave-uniform).
template <class LY, bool SKIP, bool GEN, bool GS1, bool FAR, bool TAB>
__device__ __forceinline__ bool chunk_finish(const KParams &P, const AnchorCtx &X, const TileMem &M, mask_t valid, mask_t same,
                                             int dr1, int dq1, int dd, int fj, int pj, int base, int lane, Carry &c)
{
	constexpr int SN = LY::SN;
	mask_t marked = 0;
	if (SKIP) {
		// every visited, unfiltered j stamps its predecessor; stamps whose target lies before the window are never read for this i and are
		// dropped (their slots may belong to other anchors by now)
		mask_t mk = valid & BALLOT(pj >= X.lo);
		if (FAR && X.far_mode) {
			const mask_t fm = mk & BALLOT(pj < X.stamp_lo);
			if (fm != 0) {
				int pj2 = pj;
				asm volatile("" : "+v"(pj2));                 // keep the far addressing out of the hot path
				if (fm >> lane & 1) __hip_atomic_store(&M.t[pj2], X.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				mk &= ~fm;
			}
		}
		lds_store_b16_masked(mk, LY::ST + (pj & (SN - 1)), X.s16_v);
		const int tj = *(const uint8_t *)(M.lds + LY::ST + (X.rl + (base & (SN - 1))));   // same wave, LDS is in order: the stores above (asm volatile, "memory") have landed
		marked = valid & BALLOT(tj == X.s16);
	}
	int sc;
	if (!GEN && TAB) {
		const int g = *(const int16_t *)(M.lds + LY::GAP + (min((unsigned)dd, 511u) << 1));   // 1 - gap cost (lanes that failed the filters hold any dd)
		sc = add3i(min3i(dq1, dr1, X.span1_v), fj, g);          // min(dq, dr, span) - cost + f[j], chain.c:207-208,219-220
	} else sc = pair_score<GEN, GS1>(P, X.avg, dr1 + 1, dq1 + 1, dd, same, X.span_i) + fj;
	const int scv = sel(valid, SENT, sc);
	return fold_lean<SKIP>(P, base + 63, marked, scv, c);
}

// f / p of the tile `depth` tiles before the own one (first anchor `base`, x / q ring address `addr`): from the LDS rings of the NF nearest
// tiles, else from L2 / HBM (the task's own earlier stores; relative to the caller's task there, piece-relative here)
template <class LY, int NF>
__device__ __forceinline__ void ring_fp(const TileMem &M, int addr, int depth, int base, int rl, int &fj, int &pj)
{
	if (depth <= NF) {
		const int o = (LY::XS == 4 ? addr << 1 : addr) & LY::FMASK;   // (j mod 64 NF) * 8
		const int2 fp = *(const int2 *)(M.lds + LY::FP + o);      // the ring holds f - FBIAS and p (piece-relative), see the end of the tile loop
		fj = fp.x + FBIAS; pj = fp.y;
	} else {
		const int j = max(base + rl, 0);                          // lanes before the window of a partly covered tile may point before the task
		fj = __hip_atomic_load(&M.f[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		pj = __hip_atomic_load(&M.p[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		pj = max(pj - M.pbase, -1);
	}
}

// ---------------------------------------------------------------- one chunk beyond the LDS ring: anchors, f, p and stamps from L2 / HBM
template <class LY, bool SKIP, bool GEN, bool GS1, bool TAB, bool DR0>
__device__ __forceinline__ bool far_chunk(const KParams &P, const AnchorCtx &X, const TileMem &M, mask_t in_w, int base, int lane, Carry &c)
{
	const int j = base + X.rl;
	int xj = 0, qj = 0, gj = 0;
	if (in_w >> lane & 1) {
		const uint4 aj = M.a[j];
		xj = (int)aj.x; qj = (int)aj.z;
		if (GEN) gj = (aj.w >> 16) & 0xff;
	}
	const int dr1 = X.xi1 - xj, dq1 = X.qi1 - qj;
	const int dd = absdiff(dr1, dq1);
	mask_t same = ~0ull;
	const mask_t valid = chunk_filter<GEN, false, DR0>(P, X, in_w, dr1, dq1, dd, gj, same);
	if (valid == 0) return false;
	int fj = 0, pj = -1;
	const bool vl = valid >> lane & 1;
	if (vl) {
		fj = __hip_atomic_load(&M.f[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		pj = __hip_atomic_load(&M.p[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (pj >= 0) pj -= M.pbase;               // p[] in memory is relative to the caller's task, the scan works piece-relative
	}
	mask_t marked = 0;
	if (SKIP) {
		if (vl && pj >= X.lo) __hip_atomic_store(&M.t[pj], X.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // pj < j < stamp_lo: always the global scratch
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // far stamps of this and earlier chunks have landed
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		int tj = 0;
		if (vl) tj = __hip_atomic_load(&M.t[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		marked = valid & BALLOT(tj == X.stamp);
	}
	int sc;
	if (!GEN && TAB) {
		const int g = *(const int16_t *)(M.lds + LY::GAP + (min((unsigned)dd, 511u) << 1));
		sc = add3i(min3i(dq1, dr1, X.span1_v), fj, g);
	} else sc = pair_score<GEN, GS1>(P, X.avg, dr1 + 1, dq1 + 1, dd, same, X.span_i) + fj;
	const int scv = sel(valid, SENT, sc);
	return fold_lean<SKIP>(P, base + 63, marked, scv, c);
}

// ---------------------------------------------------------------- the look-back scan of one anchor, chain.c:197-235 (C++ path: every variant)
// k = position of the anchor inside its tile (first anchor i0); own_* = the tile itself in registers (lane L = anchor i0 + 63 - L; f / p of its
// finished anchors); addr0 = per-lane byte offset of this lane's anchor of the tile before in the x / q rings.  DR0: test dr != 0 per lane.
// x / q of one ring tile for this lane (byte offset addr) as dr - 1, dq - 1.  C16: from the low halves, mod 2^16 (see Lds<>); the value -1 (dr == 0 /
// dq == 0) is kept as -1 so that the tests on it read as in the 32-bit form
template <class LY>
__device__ __forceinline__ void ring_dr_dq(const AnchorCtx &X, const TileMem &M, int addr, int &dr1, int &dq1)
{
	if (LY::XS == 8) {
		const int2 xq = *(const int2 *)(M.lds + LY::XQ + addr);
		dr1 = X.xi1 - xq.x; dq1 = X.qi1 - xq.y;
	} else if (LY::QH != LY::GAP) {              // q24 ring: the low halves of x, all 24 bits of q (the byte ring is indexed by the anchor: slot = addr / 4)
		const unsigned xq = *(const unsigned *)(M.lds + LY::XQ + addr);
		const unsigned qh = *(const uint8_t *)(M.lds + LY::QH + (addr >> 2));
		dr1 = (int)((unsigned)(X.xi1 + 1 - (int)(xq & 0xffffu)) & 0xffffu) - 1;
		dq1 = X.qi1 - (int)((xq >> 16) | (qh << 16));
	} else {
		const unsigned xq = *(const unsigned *)(M.lds + LY::XQ + addr);
		dr1 = (int)((unsigned)(X.xi1 + 1 - (int)(xq & 0xffffu)) & 0xffffu) - 1;
		dq1 = (int)((unsigned)(X.qi1 + 1 - (int)(xq >> 16)) & 0xffffu) - 1;
	}
}

template <int NX, int NF, bool SKIP, bool GEN, bool GS1, bool FAR, bool TAB, bool DR0, int C16 = 0>
__device__ __forceinline__ void scan_anchor(const KParams &P, const AnchorCtx &X, const TileMem &M, int lane, int i0, int k, mask_t eq_run,
                                            int own_x, int own_q, int own_g, int own_f, int own_p, int addr0, Carry &c)
{
	typedef Lds<NX, NF, GEN, TAB, C16> LY;
	// ---- the own tile: anchors i-1 .. i0 sit in lanes 64-k .. 63
	if (k > 0) {
		mask_t m = ~0ull << (64 - k);
		const int nin = i0 + 64 - X.lo;                       // lanes below nin hold j >= lo
		if (nin < 64) m &= first_lanes(nin);
		if (!DR0) m &= ~eq_run;
		const int dr1 = X.xi1 - own_x, dq1 = X.qi1 - own_q;
		const int dd = absdiff(dr1, dq1);
		mask_t same = ~0ull;
		const mask_t valid = chunk_filter<GEN, false, DR0>(P, X, m, dr1, dq1, dd, own_g, same);
		if (valid != 0 && chunk_finish<LY, SKIP, GEN, GS1, FAR, TAB>(P, X, M, valid, same, dr1, dq1, dd, own_f, own_p, i0, lane, c)) return;
	}
	const int before = i0 - X.lo;                             // anchors of older tiles inside the window
	if (before <= 0) return;
	const int n_full = before >> 6, part = before & 63;      // whole tiles, and the lanes of the last one that are inside
	int base = i0 - 64, depth = 1;
	int addr = addr0;                                         // this lane's anchor of the tile the scan has reached, in the x / q rings
	// ---- whole tiles from the LDS ring
#pragma nounroll
	for (int cfull = FAR ? min(n_full, NX - 1) : n_full; cfull > 0; --cfull) {
		int dr1, dq1;
		ring_dr_dq<LY>(X, M, addr, dr1, dq1);
		const int dd = absdiff(dr1, dq1);
		mask_t same = ~0ull;
		const int gj = GEN ? *(const uint8_t *)(M.lds + LY::G + (addr >> 3)) : 0;
		const mask_t valid = chunk_filter<GEN, true, DR0>(P, X, ~0ull, dr1, dq1, dd, gj, same);
		if (valid != 0) {
			int fj, pj;
			ring_fp<LY, NF>(M, addr, depth, base, X.rl, fj, pj);
			if (chunk_finish<LY, SKIP, GEN, GS1, FAR, TAB>(P, X, M, valid, same, dr1, dq1, dd, fj, pj, base, lane, c)) return;
		}
		addr = (addr - LY::TILE) & (LY::RB - 1);              // one tile back
		base -= 64; ++depth;
	}
	if (!FAR || n_full < NX - 1) {
		// ---- the last, partly covered tile, from the ring
		if (part == 0) return;
		int dr1, dq1;
		ring_dr_dq<LY>(X, M, addr, dr1, dq1);
		const int dd = absdiff(dr1, dq1);
		mask_t same = ~0ull;
		const int gj = GEN ? *(const uint8_t *)(M.lds + LY::G + (addr >> 3)) : 0;
		const mask_t valid = chunk_filter<GEN, false, DR0>(P, X, first_lanes(part), dr1, dq1, dd, gj, same);
		if (valid != 0) {
			int fj, pj;
			ring_fp<LY, NF>(M, addr, depth, base, X.rl, fj, pj);
			chunk_finish<LY, SKIP, GEN, GS1, FAR, TAB>(P, X, M, valid, same, dr1, dq1, dd, fj, pj, base, lane, c);
		}
		return;
	}
	// ---- beyond the ring (FAR): whole tiles, then the partly covered one
#pragma nounroll
	for (int cfar = n_full - (NX - 1); cfar >= 0; --cfar, base -= 64) {
		const mask_t m = cfar > 0 ? ~0ull : (part ? first_lanes(part) : 0);
		if (m != 0 && far_chunk<LY, SKIP, GEN, GS1, TAB, DR0>(P, X, M, m, base, lane, c)) return;
	}
}

// ---------------------------------------------------------------- the same scan, hand-written, for the variant that carries the throughput
// max_skip on, one segment, gap_scale 1 (chain.c:218 without :219's double path), window inside the LDS ring, no equal-x run beyond the
// own tile.  One instruction sequence per anchor; registers: see the operand list.  Chunk loop: tile `d` tiles back (d = 0: the own tile)
// is filtered with 7 VALU + 1 v_cmp + 1 s_and; only a chunk with a surviving lane enters Lne (f / p fetch, stamps, score, fold).
// The fold has three paths (as fold_lean): A no lane beats the running best, B1 some does and neither marks nor skips exist,
// B2 the general case (prefix max by DPP, skip counter as a max-plus scan n <- max(n + d, 0) over the lanes).
// Wait states of gfx940 that the assembler does not insert for inline asm: VALU write -> DPP read of the same VGPR: 2 (s_nop 1);
// VALU write -> v_readlane of it: s_nop 0 kept for safety; LDS results: s_waitcnt lgkmcnt(0); global results: s_waitcnt vmcnt(0).
// ---------------------------------------------------------------- the same scan, hand-written, for the variant that carries the throughput
// max_skip on, one segment, max_dq - 1 >= bw; gap cost computed (gap_scale 1) or read from the per-task table.  One instruction sequence
// per TILE: anchors k_start .. of the tile with first anchor i0, one after the other; lane L = 63 - k holds anchor i0 + k in the per-tile
// registers: tx / tq = x, q; tx1 / tq1 = x - 1, q - 1; tspan = span; tlo = window start, tbef = anchors of older tiles inside the window
// (both clamped to what the ring holds; tbef travels in bits 15-24 of tw); 64 - L = LDS stamp; tw = number of own-tile predecessors inside the window, bit 29: no window at
// all, bit 30: window clamped, bit 31: anchor not handled here.  Results go into the anchor's lane of own_f / own_p.  Returns the position
// of the first anchor it did not process (cnt when the tile is done, else a bit-31 anchor).  A clamped window that the ring part of the scan
// does not end (no `break` of chain.c:231) goes on tile by tile from L2 / HBM: x, q from the anchor array, f / p from the task's own
// earlier stores, stamps in the 32-bit global scratch t[] (value i + 1) -- also for the far predecessors of ring lanes.
// Per anchor: the own tile (lanes L+1 .. L+w), then `nfull` whole older tiles in a count-down loop that branches on VCC (7 VALU + 2 LDS
// + 1 SALU + 2 branches per tile; x / q of the next tile are requested while this one is filtered), then the partly covered tile.
// A chunk with a surviving lane goes through Lhf: f / p of its tile (registers, LDS, or L2 beyond NF tiles), stamps, score, fold.
// The fold: A no lane beats the running best; B0 the first surviving lane does and no other lane beats IT (the usual case on a real chain: the
// nearest predecessor is the best) -- it is the only new maximum, every marked lane behind it is a skip event, so counter and `break` have a
// closed form (the `break` needs a skip event: without one the counter is not even compared, max_skip may be negative); B1 some lane does and neither marks nor skips exist (one candidate: no reduction at all);
// B2 general: prefix max by DPP -> lanes that raise the best (nm), skip events (se); closed form when every nm precedes every se,
// else the max-plus scan n <- max(n + d, 0) over the lanes.
// Wait states the assembler does not insert for inline asm (gfx940): VALU write -> DPP read of that VGPR: 2 (s_nop 1); VALU write ->
// v_readlane of it: s_nop 0 kept for safety; SDWA write with dst_sel != DWORD (the compact ring's 16-bit subtractions) -> VALU read of that
// VGPR: 1 (met by instruction order, see MM2C_FILTER2); LDS results: s_waitcnt lgkmcnt(0); global results: s_waitcnt vmcnt(0).
// Registers the block touches beyond its operands: VCC and SCC (declared clobbers); M0 (the lane select of v_writelane) and EXEC (the masked
// stamp stores of the `far` instantiation).  clang refuses both on a clobber list ("reserved registers"): M0 is safe because the compiler loads it
// right before each of its own uses and never keeps a value in it across a statement; EXEC is saved on entry and that value -- not a literal
// -1 -- is what the block puts back after every masked store, so the block leaves EXEC exactly as it found it.
// LDS: the block addresses the kernel's LDS from 0 with immediate offsets (Lds<>): the kernel must own exactly ONE LDS object.  Checked three
// times: when the library is built (Makefile: the group segment size of every chain_dp_tile kernel in the code object equals Lds<>::BYTES of its
// template arguments -- a second object would add to it), by tests/test_cpu_abi.py on the shipped library, and at run time (status 3).
#define MM2C_DPP_STEP(R, CTRL) "s_nop 1\n\t" "v_max_i32_dpp " R ", " R ", " R " " CTRL "\n\t"
#define MM2C_DPP_PREFIX_MAX(R) MM2C_DPP_STEP(R, "row_shr:1 row_mask:0xf bank_mask:0xf") MM2C_DPP_STEP(R, "row_shr:2 row_mask:0xf bank_mask:0xf") \
	MM2C_DPP_STEP(R, "row_shr:4 row_mask:0xf bank_mask:0xf") MM2C_DPP_STEP(R, "row_shr:8 row_mask:0xf bank_mask:0xf") \
	MM2C_DPP_STEP(R, "row_bcast:15 row_mask:0xa bank_mask:0xf") MM2C_DPP_STEP(R, "row_bcast:31 row_mask:0xc bank_mask:0xf")
#define MM2C_FILTER(X, Q) "v_sub_u32 %[dr], %[xi1], " X "\n\t" "v_sub_u32 %[dq], %[qi1], " Q "\n\t"
// (order: the saturating subtraction on dq first, then |dr - dq| -- with the compact ring dq and dr come out of SDWA instructions with dst_sel:WORD_0, and gfx940 / gfx950
// need one wait state between such a write and a VALU read of the register; the compact filters write dq, then dr, so that each is read two instructions after its write)
#define MM2C_FILTER2 "v_sub_u32_e64 %[u1], %[dq], %[mdqbw] clamp\n\t" "v_sad_u32 %[dd], %[dr], %[dq], 0\n\t" "v_max_u32 %[u1], %[u1], %[dd]\n\t" \
	"v_cmp_ge_u32 vcc, %[bw], %[u1]\n\t"
// x / q of an older tile and f - FBIAS / p of a scored one arrive as pairs (one ds_read_b64 each).  Their halves are used one by one, and the
// operand syntax of inline assembly cannot name half of a 64-bit operand: the four values live in FIXED registers (MM2C_R_X .. MM2C_R_P) that the
// block lists as clobbers, so the compiler keeps them free across it.
#define MM2C_R_XQ "v[62:63]"
#define MM2C_R_X "v62"
#define MM2C_R_Q "v63"
#define MM2C_R_FP "v[60:61]"
#define MM2C_R_F "v60"
#define MM2C_R_P "v61"
// ---- what depends on the slot size of the x / q ring (Lds<>::XS): the 32-bit form `W` and the compact form `C` (low halves of x and q in one dword)
// first request of an anchor; request of the next tile + the running address one tile back; the filter's two subtractions on a ring tile; the f / p and
// stamp slots of a scored ring tile from the running address (which is two tiles further on); the running address one tile back (partly covered tile)
#define MM2C_XQ1_W "ds_read_b64 " MM2C_R_XQ ", %[addr1] offset:%[XQOFF]\n\t"
#define MM2C_XQ1_C "ds_read_b32 " MM2C_R_X ", %[addr1] offset:%[XQOFF]\n\t"
#define MM2C_BACK_W "v_add_u32 %[addr], 0xfffffe00, %[addr]\n\t" "v_and_b32 %[addr], %[RBM1], %[addr]\n\t"
#define MM2C_BACK_C "v_add_u32 %[addr], 0xffffff00, %[addr]\n\t" "v_and_b32 %[addr], %[RBM1], %[addr]\n\t"
#define MM2C_NEXT_XQ_W "ds_read_b64 " MM2C_R_XQ ", %[addr] offset:%[XQOFF]\n\t" MM2C_BACK_W
#define MM2C_NEXT_XQ_C "ds_read_b32 " MM2C_R_X ", %[addr] offset:%[XQOFF]\n\t" MM2C_BACK_C
#define MM2C_RFILTER_W MM2C_FILTER("" MM2C_R_X "", "" MM2C_R_Q "")
#define MM2C_OWNFILTER_W MM2C_FILTER("%[tx]", "%[tq]")
#define MM2C_FARFILTER_W MM2C_FILTER("%[fx]", "%[fq]")
#define MM2C_RDXQ_W "v_readfirstlane_b32 %[xi1], %[tx1]\n\t" "v_readfirstlane_b32 %[qi1], %[tq1]\n\t"
// 16-bit subtractions: (x_i - 1 - x_j) mod 2^16 and (q_i - 1 - q_j) mod 2^16, zero-extended (SDWA: the result goes to the low word, the rest is padded with
// zeros).  The anchor's own x - 1 and q - 1 travel as ONE packed word too (xi1: low halves of x - 1 | q - 1 << 16; one v_readfirstlane per anchor less), the own
// tile is filtered from its packed word (the operand tx: what the tile wrote into the ring), and a tile from memory (fx, fq: 32-bit loads) by its low halves
#define MM2C_SUB16(D, S0SEL, V, S1SEL) "v_sub_u16_sdwa " D ", %[xi1], " V " dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:" S0SEL " src1_sel:" S1SEL "\n\t"
#define MM2C_RFILTER_C MM2C_SUB16("%[dq]", "WORD_1", MM2C_R_X, "WORD_1") MM2C_SUB16("%[dr]", "WORD_0", MM2C_R_X, "WORD_0")
#define MM2C_OWNFILTER_C MM2C_SUB16("%[dq]", "WORD_1", "%[tx]", "WORD_1") MM2C_SUB16("%[dr]", "WORD_0", "%[tx]", "WORD_0")
#define MM2C_FARFILTER_C MM2C_SUB16("%[dq]", "WORD_1", "%[fq]", "WORD_0") MM2C_SUB16("%[dr]", "WORD_0", "%[fx]", "WORD_0")
#define MM2C_RDXQ_C "v_readfirstlane_b32 %[xi1], %[tx1]\n\t"
// q24 ring: the slot as in the compact form + the byte ring QH (bits 16-23 of q), addressed by the anchor's slot number = addr / 4.  x - 1 and q - 1 of the anchor travel
// as two full scalars like in the 32-bit form (the own tile and tiles from memory are filtered exactly as there); only a RING tile differs: dr from the low halves,
// q of the ring anchor = {slot.b2, slot.b3, byte, 0} by v_perm_b32 (selector 0x0c040302 in a VGPR: bytes 0-3 of the selector's source are S1's, 4-7 S0's, 0x0c = 0)
#define MM2C_XQ1_Q "ds_read_b32 " MM2C_R_X ", %[addr1] offset:%[XQOFF]\n\t" "ds_read_u8 " MM2C_R_Q ", %[addr1q] offset:%[QHOFF]\n\t"
#define MM2C_NEXT_XQ_Q "v_lshrrev_b32 %[u2], 2, %[addr]\n\t" "ds_read_b32 " MM2C_R_X ", %[addr] offset:%[XQOFF]\n\t" "ds_read_u8 " MM2C_R_Q ", %[u2] offset:%[QHOFF]\n\t" MM2C_BACK_C
#define MM2C_RFILTER_Q "v_sub_u16_sdwa %[dr], %[xi1], " MM2C_R_X " dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n\t" \
	"v_perm_b32 %[u1], " MM2C_R_Q ", " MM2C_R_X ", %[selq]\n\t" "v_sub_u32 %[dq], %[qi1], %[u1]\n\t"
#define MM2C_OLDADDR_W "v_add_u32 %[vb], 0x400, %[addr]\n\t"
#define MM2C_OLDADDR_C "v_add_lshl_u32 %[vb], %[addr], %[c200], 1\n\t"
// x / q of the tile with first anchor fb from memory (anchors are 16 bytes: x low word at 0, q at 8), fb one tile back afterwards
#define MM2C_FAR_REQ "v_add_u32 %[u2], %[fb], %[rl]\n\t" "v_max_i32 %[u2], 0, %[u2]\n\t" "v_lshlrev_b32 %[u2], 4, %[u2]\n\t" \
	"global_load_dword %[fx], %[u2], %[aptr]\n\t" "global_load_dword %[fq], %[u2], %[aptr] offset:8\n\t" "s_sub_i32 %[fb], %[fb], 64\n\t"
#define MM2C_SCORE_CMP "v_or_b32 %[va], 1, %[dd]\n\t" "v_ffbh_u32 %[va], %[va]\n\t" "v_lshrrev_b32 %[va], 1, %[va]\n\t" "v_cvt_f32_u32 %[vc], %[dd]\n\t" \
	"v_mul_f32 %[vc], %[avg], %[vc]\n\t" "v_cvt_i32_f32 %[vc], %[vc]\n\t" "v_min3_i32 %[sc], %[dq], %[dr], %[span1]\n\t" "v_sub_u32 %[sc], %[sc], %[vc]\n\t"
#define MM2C_ADDF_CMP "v_add3_u32 %[sc], %[sc], %[va], " MM2C_R_F "\n\t"   /* vf = f[j] - 14 */
#define MM2C_SCORE_TAB "v_min_u32 %[va], 0x1ff, %[dd]\n\t" "v_lshlrev_b32 %[va], 1, %[va]\n\t" "ds_read_i16 %[va], %[va] offset:%[GAPOFF]\n\t" \
	"v_min3_i32 %[sc], %[dq], %[dr], %[span1]\n\t"
#define MM2C_ADDF_TAB "v_add3_u32 %[sc], %[sc], %[va], " MM2C_R_F "\n\t" "v_add_u32 %[sc], 14, %[sc]\n\t"   /* vf = f[j] - 14; the table holds 1 - cost */

// ---- the segments in which the two instantiations of the loop differ.  `far`: the tile holds an anchor whose window reaches beyond the LDS ring
// (bit 30 of its tw word): stamps with a target before the ring go to the global scratch, and a scan that runs through the whole ring without
// the `break` goes on from memory (bit 28 set at run time).  `lean`: no such anchor in the tile, so nothing of that is tested; the stamp store
// needs no exec mask either: lanes that must not stamp (filtered out, or p before the window) write the slot of anchor lo - 1 instead, which
// no scan of THIS anchor reads and whose content no other anchor can mistake for its own stamp (stamps are unique within a tile and the ring is wiped when a tile starts).
// Requests for tiles beyond the ring: the one for the first such tile goes out when an anchor with a clamped window starts (it is needed unless
// the scan of the ring ends with the `break`), the one for each further tile while the tile before it is filtered.  A request that the `break`
// has made useless is not awaited when the anchor is committed (that wait cost 5 % on colinear streams): loads return in order, so a later
// request into the same registers lands later, and everything is awaited where the block ends (Lexit) -- nothing may be in flight there, the
// compiler does not know about these loads.  Measured (dense / ava-ont colinear, ms): no prefetch 90.0 / 37.7; first request only once the ring is
// exhausted 87.6 / 37.9; at the anchor's start 85.5 / 39.4 (kept); at the start only if the last such anchor went beyond the ring 87.0 / 38.8.
#define MM2C_DONE_FAR ""
#define MM2C_RD_FAR \
	"v_readfirstlane_b32 %[lo], %[tlo]\n\t" \
	"v_readfirstlane_b32 %[lo0], %[tlo0]\n\t"
#define MM2C_RD_LEAN ""
#define MM2C_LK_FAR \
	"s_sub_i32 %[fb], %[i0], %[REACH]\n\t" \
	"s_bitcmp1_b32 %[pk], 30\n\t" \
	"s_cbranch_scc0 Lnf_%=\n\t" \
	MM2C_FAR_REQ \
	"Lnf_%=:\n\t"
#define MM2C_LK_LEAN ""
#define MM2C_HF_FAR \
	"v_cmp_le_i32 vcc, %[lo], " MM2C_R_P "\n\t" \
	"s_and_b64 %[mk], vcc, %[valid]\n\t" \
	"v_and_b32 %[u2], %[SNM1], " MM2C_R_P "\n\t" \
	"s_mov_b64 exec, %[mk]\n\t" \
	"ds_write_b8 %[u2], %[s16v] offset:%[STOFF]\n\t" \
	"s_mov_b64 exec, %[valid]\n\t" \
	"ds_read_i8 %[vb], %[vb] offset:%[STOFF]\n\t" \
	"s_bitcmp1_b32 %[pk], 30\n\t" \
	"s_cbranch_scc0 Lmk_%=\n\t" \
	"v_cmp_le_i32 vcc, %[lo0], " MM2C_R_P "\n\t" \
	"s_andn2_b64 %[mask], vcc, %[mk]\n\t"      /* (exec = valid: vcc is already confined to the lanes that passed) */ \
	"s_cbranch_scc0 Lmk_%=\n\t" \
	MM2C_LC(MM2C_LB_FARSTAMP) \
	"s_add_i32 %[t0], %[icnt1], %[c]\n\t" \
	"v_mov_b32 %[u1], %[t0]\n\t" \
	"v_lshlrev_b32 %[u2], 2, " MM2C_R_P "\n\t" \
	"s_mov_b64 exec, %[mask]\n\t" \
	"global_store_dword %[u2], %[u1], %[tptr] sc0\n\t" \
	"s_mov_b64 exec, %[valid]\n"
#define MM2C_HF_LEAN \
	"s_mov_b64 exec, %[valid]\n\t" \
	"v_max_i32 %[u2], " MM2C_R_P ", %[lomc]\n\t" \
	"v_and_b32 %[u2], %[SNM1], %[u2]\n\t" \
	"ds_write_b8 %[u2], %[s16v] offset:%[STOFF]\n\t" \
	"ds_read_i8 %[vb], %[vb] offset:%[STOFF]\n"
#define MM2C_TAIL_FAR \
	"s_cbranch_scc0 Lret_%=\n\t" \
	"s_bcnt1_i32_b64 %[t0], %[marked]\n\t" \
	"s_add_u32 %[nskip], %[nskip], %[t0]\n\t" \
	"s_cmp_gt_i32 %[nskip], %[maxskip]\n\t" \
	MM2C_LC_BR_SCC1("Ldone_%=", MM2C_LB_BRKA) \
	"\n" \
	"Lret_%=:\n\t" \
	"s_bitcmp1_b32 %[pk], 28\n\t" \
	"s_cbranch_scc0 Lloop_%=\n\t" \
	"s_branch Lfloop_%=\n"
#define MM2C_TAIL_LEAN \
	"s_cbranch_scc0 Lloop_%=\n\t" \
	"s_bcnt1_i32_b64 %[t0], %[marked]\n\t" \
	"s_add_u32 %[nskip], %[nskip], %[t0]\n\t" \
	"s_cmp_gt_i32 %[nskip], %[maxskip]\n\t" \
	MM2C_LC_BR_SCC1("Ldone_%=", MM2C_LB_BRKA) \
	"\n" \
	"Lret_%=:\n\t" \
	"s_branch Lloop_%=\n"
#define MM2C_END_FAR(SCORE, FARFILTER) \
	"Lend_%=:\n\t" \
	MM2C_LC(MM2C_LB_END) \
	"s_bitcmp1_b32 %[pk], 30\n\t" \
	"s_cbranch_scc0 Ldone_%=\n\t" \
	"s_bitset1_b32 %[pk], 28\n\t" \
	"s_sub_i32 %[t0], %[i0], %[lo0]\n\t" \
	"s_and_b32 %[part], %[t0], 63\n\t" \
	"s_lshr_b32 %[nfull], %[t0], 6\n\t" \
	"s_sub_i32 %[n], %[nfull], %[NXM1]\n\t" \
	"s_mov_b32 %[d], %[NXM1]\n\t" \
	"s_mov_b32 %[lo], %[lo0]\n\t" \
	"s_add_i32 %[s16], %[icnt1], %[c]\n\t" \
	"v_mov_b32 %[s16v], %[s16]\n" \
	"Lfloop_%=:\n\t" \
	MM2C_LC(MM2C_LB_FLOOP) \
	"s_add_u32 %[d], %[d], 1\n\t" \
	"s_sub_u32 %[n], %[n], 1\n\t" \
	"s_cbranch_scc1 Lfpart_%=\n\t" \
	"s_waitcnt vmcnt(0)\n\t" \
	FARFILTER MM2C_FAR_REQ MM2C_FILTER2 \
	"s_cbranch_vccz Lfloop_%=\n\t" \
	"s_mov_b64 %[valid], vcc\n\t" \
	"s_branch Lfold_%=\n" \
	"Lfpart_%=:\n\t" \
	MM2C_LC(MM2C_LB_FPART) \
	"s_mov_b32 %[n], 0\n\t" \
	"s_cmp_eq_u32 %[part], 0\n\t" \
	"s_cbranch_scc1 Ldone_%=\n\t" \
	"s_waitcnt vmcnt(0)\n\t" \
	FARFILTER MM2C_FILTER2 \
	"s_sub_i32 %[t0], 64, %[part]\n\t" \
	"s_lshr_b64 %[mask], -1, %[t0]\n\t" \
	"s_mov_b32 %[part], 0\n\t" \
	"s_and_b64 %[valid], vcc, %[mask]\n\t" \
	"s_cbranch_scc0 Ldone_%=\n" \
	"Lfold_%=:\n\t" \
	MM2C_LC(MM2C_LB_FOLD) \
	"s_lshl_b32 %[t0], %[d], 6\n\t" \
	"s_sub_i32 %[base], %[i0], %[t0]\n\t" \
	"v_add_u32 %[u2], %[base], %[rl]\n\t" \
	"v_max_i32 %[u2], 0, %[u2]\n\t" \
	"v_lshlrev_b32 %[u2], 2, %[u2]\n\t" \
	"global_load_dword " MM2C_R_P ", %[u2], %[pptr] sc0\n\t" \
	"global_load_dword " MM2C_R_F ", %[u2], %[fptr] sc0\n\t" \
	SCORE \
	"s_waitcnt vmcnt(0)\n\t" \
	"v_add_u32 " MM2C_R_F ", -14, " MM2C_R_F "\n\t" \
	"v_subrev_u32 " MM2C_R_P ", %[pbase], " MM2C_R_P "\n\t" \
	"v_max_i32 " MM2C_R_P ", -1, " MM2C_R_P "\n\t" \
	"v_cmp_le_i32 vcc, %[lo], " MM2C_R_P "\n\t" \
	"s_and_b64 %[mk], vcc, %[valid]\n\t" \
	"v_lshlrev_b32 %[u1], 2, " MM2C_R_P "\n\t" \
	"s_mov_b64 exec, %[mk]\n\t" \
	"global_store_dword %[u1], %[s16v], %[tptr] sc0\n\t" \
	"s_mov_b64 exec, %[valid]\n\t" \
	"s_waitcnt vmcnt(0)\n\t" \
	"global_load_dword %[vb], %[u2], %[tptr] sc0\n\t" \
	"s_waitcnt vmcnt(0)\n\t" \
	"v_cmp_eq_u32 vcc, %[s16], %[vb]\n\t" \
	"s_branch Lmk2_%=\n"
#define MM2C_END_LEAN(SCORE, FARFILTER) \
	"Lend_%=:\n\t" \
	MM2C_LC(MM2C_LB_END)

#define MM2C_READ_ANCHOR(SEG_RD, RDXQ) \
	"v_readfirstlane_b32 %[pk], %[tw]\n\t" \
	"v_readfirstlane_b32 %[best], %[tspan]\n\t" \
	SEG_RD \
	RDXQ
// price-list probes (tools/probe_prices.sh): extra instructions of one class per anchor (MM2C_PROBE_LK) or per older tile (MM2C_PROBE_LOOP) that
// change no result -- u1 / t1 are dead at both places -- so that the time per added instruction of each class can be measured on the real kernel
// (-DMM2C_PROBE=1..6: four plain VALU / four SALU / four v_readlane per anchor, two plain VALU / two SALU / two v_cmp per older tile)
#define MM2C_P_V "v_add_u32 %[u1], 1, %[u1]\n\t"
#define MM2C_P_S "s_add_u32 %[t1], %[t1], 1\n\t"
#define MM2C_P_R "v_readlane_b32 %[t1], %[tw], %[c]\n\t"
#define MM2C_P_C "v_cmp_eq_u32 vcc, %[u1], %[u1]\n\t"
#if MM2C_PROBE == 1
#define MM2C_PROBE_LK MM2C_P_V MM2C_P_V MM2C_P_V MM2C_P_V
#elif MM2C_PROBE == 2
#define MM2C_PROBE_LK MM2C_P_S MM2C_P_S MM2C_P_S MM2C_P_S
#elif MM2C_PROBE == 3
#define MM2C_PROBE_LK MM2C_P_R MM2C_P_R MM2C_P_R MM2C_P_R
#elif MM2C_PROBE == 4
#define MM2C_PROBE_LOOP MM2C_P_V MM2C_P_V
#elif MM2C_PROBE == 5
#define MM2C_PROBE_LOOP MM2C_P_S MM2C_P_S
#elif MM2C_PROBE == 6
#define MM2C_PROBE_LOOP MM2C_P_C MM2C_P_C
#elif MM2C_PROBE == 7     /* v_readlane with a constant lane */
#define MM2C_P_X "v_readlane_b32 %[t1], %[tw], 5\n\t"
#define MM2C_PROBE_LK MM2C_P_X MM2C_P_X MM2C_P_X MM2C_P_X
#elif MM2C_PROBE == 8     /* v_readfirstlane */
#define MM2C_P_X "v_readfirstlane_b32 %[t1], %[tw]\n\t"
#define MM2C_PROBE_LK MM2C_P_X MM2C_P_X MM2C_P_X MM2C_P_X
#elif MM2C_PROBE == 9     /* one scalar load per anchor (always the same line) */
#define MM2C_PROBE_LK "s_load_dwordx2 %[pr], %[aptr], 0x0\n\t"
#define MM2C_PROBE_OPERAND , [pr] "=&s"(pr)
#elif MM2C_PROBE == 10    /* two LDS reads per anchor */
#define MM2C_P_X "ds_read_b32 %[u1], %[addr1]\n\t"
#define MM2C_PROBE_LK MM2C_P_X MM2C_P_X
#elif MM2C_PROBE == 11    /* v_readlane of four DIFFERENT registers into four different SGPRs */
#define MM2C_PROBE_LK "v_readlane_b32 %[t1], %[tw], %[c]\n\tv_readlane_b32 %[t0], %[tx], %[c]\n\tv_readlane_b32 %[last], %[tq], %[c]\n\tv_readlane_b32 %[base], %[tlo0], %[c]\n\t"
#elif MM2C_PROBE == 12    /* two v_writelane through m0 into a dead register */
#define MM2C_PROBE_LK "s_mov_b32 m0, %[c]\n\ts_nop 0\n\tv_writelane_b32 %[u1], %[c], m0\n\tv_writelane_b32 %[u1], %[c], m0\n\t"
#elif MM2C_PROBE == 13    /* four taken branches per anchor */
#define MM2C_P_X "s_branch 1f\n\ts_nop 0\n1:\n\t"
#define MM2C_PROBE_LK MM2C_P_X MM2C_P_X MM2C_P_X MM2C_P_X
#endif
#ifndef MM2C_PROBE_LK
#define MM2C_PROBE_LK ""
#endif
#ifndef MM2C_PROBE_OPERAND
#define MM2C_PROBE_OPERAND
#endif
#ifndef MM2C_PROBE_LOOP
#define MM2C_PROBE_LOOP ""
#endif
// ---- f / p of a scored ring tile deeper than the f / p ring: from L2 (the task's own earlier stores), the score computed while they are on their way.
// MM2C_FG_PLAIN: request, score, wait.  MM2C_FG_W (the 32-bit ring forms; the compact form has no registers to spare under its launch bound): the values of the
// NEXT tile of the window are requested as soon as this tile's are in, into a second register pair, keyed by the tile's first anchor (bpre) -- a deep tile
// that follows a scored one (the usual case on a noisy long read: ava-ont streams, 1.5 deep tiles per anchor) then finds its f / p already there, or on their
// way since the fold of the tile before began.  The values belong to a finished tile, so they stay good for the following anchors of the own tile as well.
// Every wait in the block is vmcnt(0), so the extra loads in flight change no other wait; nothing is in flight at Lexit.
#define MM2C_R_F2 "v58"
#define MM2C_R_P2 "v59"
#define MM2C_FG_CLOB_PF0
#define MM2C_FG_CLOB_PF2 , MM2C_R_F2, MM2C_R_P2
#define MM2C_FG_CLOB(SEL) MM2C_FG_CLOB_##SEL      /* (a name, pasted here: a clobber list as a macro argument would fall apart at its commas) */
#define MM2C_FG_PLAIN(SCORE) \
	"s_lshl_b32 %[t0], %[d], 6\n\t" \
	"s_sub_i32 %[base], %[i0], %[t0]\n\t" \
	"v_add_u32 %[u2], %[base], %[rl]\n\t" \
	"v_max_i32 %[u2], 0, %[u2]\n\t" \
	"v_lshlrev_b32 %[u2], 2, %[u2]\n\t" \
	"global_load_dword " MM2C_R_P ", %[u2], %[pptr] sc0\n\t" \
	"global_load_dword " MM2C_R_F ", %[u2], %[fptr] sc0\n\t" \
	SCORE \
	"s_waitcnt vmcnt(0)\n\t"
#if MM2C_DEEP_PREFETCH
#define MM2C_FG_W(SCORE) \
	"s_lshl_b32 %[t0], %[d], 6\n\t" \
	"s_sub_i32 %[base], %[i0], %[t0]\n\t" \
	"s_cmp_eq_u32 %[base], %[bpre]\n\t" \
	"s_cbranch_scc1 Lfgh_%=\n\t" \
	"v_add_u32 %[u2], %[base], %[rl]\n\t" \
	"v_max_i32 %[u2], 0, %[u2]\n\t" \
	"v_lshlrev_b32 %[u2], 2, %[u2]\n\t" \
	"global_load_dword " MM2C_R_P ", %[u2], %[pptr] sc0\n\t" \
	"global_load_dword " MM2C_R_F ", %[u2], %[fptr] sc0\n\t" \
	SCORE \
	"s_waitcnt vmcnt(0)\n\t" \
	"s_branch Lfgp_%=\n" \
	"Lfgh_%=:\n\t" \
	SCORE \
	"s_waitcnt vmcnt(0)\n\t" \
	"v_mov_b32 " MM2C_R_P ", " MM2C_R_P2 "\n\t" \
	"v_mov_b32 " MM2C_R_F ", " MM2C_R_F2 "\n" \
	"Lfgp_%=:\n\t" \
	"s_or_b32 %[t0], %[n], %[part]\n\t"            /* whole tiles left + lanes of the partly covered one: anything further back in this window? */ \
	"s_cbranch_scc0 Lfgx_%=\n\t" \
	"s_sub_i32 %[bpre], %[base], 64\n\t" \
	"v_add_u32 %[u2], %[bpre], %[rl]\n\t" \
	"v_max_i32 %[u2], 0, %[u2]\n\t" \
	"v_lshlrev_b32 %[u2], 2, %[u2]\n\t" \
	"global_load_dword " MM2C_R_P2 ", %[u2], %[pptr] sc0\n\t" \
	"global_load_dword " MM2C_R_F2 ", %[u2], %[fptr] sc0\n" \
	"Lfgx_%=:\n\t"
#else
#define MM2C_FG_W(SCORE) MM2C_FG_PLAIN(SCORE)
#endif

// Label counters (-DMM2C_LABEL_COUNT: minimap2-fpga_amd/variants/labelcount.so, never the shipped library): MM2C_LC(bit) adds one to lane `bit` of the
// per-wave register `lc` wherever the block passes -- at the labels of the assembly and on the fall-through side of the branches that pick a fold -- so that the
// GPU test tests/test_gpu_labels.py can show that the parity inputs drive every path of the REAL instruction sequence (the kernel adds `lc` to
// g_label_hits after every call; mm2c_debug_label_hits reads the table).  It touches no register the loop uses: exec is parked in `pr` and put back,
// VCC and SCC are left alone (v_add_u32 has no carry out on gfx9), and the s_nop keeps a DPP instruction that follows away from the exec write.
#ifdef MM2C_LABEL_COUNT
#define MM2C_LC(BIT) "s_mov_b64 %[pr], exec\n\t" "s_mov_b64 exec, " BIT "\n\t" "v_add_u32 %[lc], 1, %[lc]\n\t" "s_mov_b64 exec, %[pr]\n\t" "s_nop 4\n\t"
#define MM2C_LC_BR_SCC1(LABEL, BIT) "s_cbranch_scc0 8f\n\t" MM2C_LC(BIT) "s_branch " LABEL "\n" "8:\n\t"
#define MM2C_LC_PARAM , int &lc
#define MM2C_LC_ARG , lc_v
#define MM2C_LC_OPERAND , [lc] "+v"(lc), [pr] "=&s"(pr)
__device__ unsigned long long g_label_hits[8 * 32];   // row = compact << 2 | table << 1 | far, column = label bit
#else
#define MM2C_LC(BIT) ""
#define MM2C_LC_BR_SCC1(LABEL, BIT) "s_cbranch_scc1 " LABEL "\n\t"
#define MM2C_LC_PARAM
#define MM2C_LC_ARG
#define MM2C_LC_OPERAND
#endif
#define MM2C_LB_K "0x1"
#define MM2C_LB_OWN "0x2"
#define MM2C_LB_LOOP "0x4"
#define MM2C_LB_OLD "0x8"
#define MM2C_LB_HF "0x10"
#define MM2C_LB_FARSTAMP "0x20"
#define MM2C_LB_BRKA "0x40"
#define MM2C_LB_FG "0x80"
#define MM2C_LB_PART "0x100"
#define MM2C_LB_PARTPASS "0x200"
#define MM2C_LB_IMP "0x400"
#define MM2C_LB_B0 "0x800"
#define MM2C_LB_SLOW2 "0x1000"
#define MM2C_LB_SLOW "0x2000"
#define MM2C_LB_B1 "0x4000"
#define MM2C_LB_B1M "0x8000"
#define MM2C_LB_B2 "0x10000"
#define MM2C_LB_B2CLOSED "0x20000"
#define MM2C_LB_LI "0x40000"
#define MM2C_LB_LICLOSED "0x80000"
#define MM2C_LB_CFB "0x100000"
#define MM2C_LB_GEN "0x200000"
#define MM2C_LB_BK "0x400000"
#define MM2C_LB_TK "0x800000"
#define MM2C_LB_AF "0x1000000"
#define MM2C_LB_END "0x2000000"
#define MM2C_LB_FLOOP "0x4000000"
#define MM2C_LB_FPART "0x8000000"
#define MM2C_LB_FOLD "0x10000000"
#define MM2C_LB_DONE "0x20000000"
#define MM2C_LB_SPEC "0x40000000"
#define MM2C_SCAN_TILE_ASM(NAME, TABV, C16V, XQ1, NEXT_XQ, RFILTER, OLDADDR, BACK, OWNFILTER, FARFILTER, RDXQ, SEG_FG, CLOB, SCORE, ADDF, SEG_RD, SEG_LK, SEG_HF, SEG_TAIL, SEG_END, SEG_DONE, LNEXT) \
template <int NX, int NF> \
__device__ __forceinline__ int NAME(int i0, int k_start, int cnt, int max_skip, float avg, const int32_t *f, const int32_t *p, int pbase, const uint4 *a, \
                                    int32_t *tg, int tx, int tx1, int tq, int tq1, int tspan, int tlo, int tlo0, int tw, int &own_f, int &own_p, \
                                    int addr1, int addr2, int lomc, int ownst, int rl, int mdqbw_v, int bw_v, int sent_v, int addr1q, int selq_v MM2C_LC_PARAM) \
{ \
	typedef Lds<NX, NF, false, TABV, C16V> LY; \
	typedef Lds<NX, NF, false, true, C16V> LYT; \
	int best, bestj, nskip, n, nfull, part, base, t0, t1, last, c, pk, lo, lo0, xi1, qi1, span1, s16, d, fb, bpre; \
	mask_t mask, valid, mk, marked, nm, se, ex, oh, pr; (void)pr; \
	int dr, dq, dd, u1, u2, sc, va, vb, vc, addr, s16v, lom1v, fx, fq; \
	asm volatile( \
		"s_mov_b64 %[ex], exec\n\t" \
		"s_mov_b32 %[bpre], 0x7fffffff\n\t"       /* no tile's deep f / p has been requested ahead (MM2C_FG_W) */ \
		"s_sub_i32 %[c], %[kstart], %[cnt]\n\t"    /* the anchor counter: position in the tile - anchors of the tile, in [-64, -1]; its carry ends the loop and its low byte is the LDS stamp */ \
		"s_sub_i32 %[t0], 63, %[kstart]\n\t" \
		"s_lshl_b64 %[mask], 1, %[t0]\n\t" \
		"s_lshr_b64 %[oh], %[mask], 1\n\t" \
		"s_or_b64 %[oh], %[oh], %[mask]\n\t"      /* lanes L and L - 1: the anchor in progress and the next one */ \
		"s_mov_b64 exec, %[mask]\n\t" \
		MM2C_READ_ANCHOR(SEG_RD, RDXQ) \
		"s_mov_b64 exec, %[ex]\n" \
		"Lk_%=:\n\t" \
		MM2C_LC(MM2C_LB_K) \
		MM2C_PROBE_LK \
		"s_mov_b32 %[bestj], -1\n\t" \
		"s_cmp_lt_i32 %[pk], 0\n\t" \
		"s_cbranch_scc1 Lspec_%=\n\t"              /* bit 31: not for this loop, or (bit 29 too) no window at all */ \
		SEG_LK \
		"s_bfe_u32 %[n], %[pk], 0x40015\n\t"       /* whole older tiles inside the window */ \
		XQ1 \
		"v_mov_b32 %[addr], %[addr2]\n\t" \
		"s_add_i32 %[span1], %[best], -1\n\t" \
		"s_mov_b32 %[nskip], 0\n\t" \
		"s_bfe_u32 %[part], %[pk], 0x6000f\n\t"    /* lanes of the partly covered tile behind them */ \
		"s_mov_b32 %[nfull], %[n]\n\t" \
		"v_mov_b32 %[s16v], %[c]\n\t" \
		"s_and_b32 %[t0], %[pk], 63\n\t" \
		"s_cbranch_scc0 Lloop_%=\n\t" \
		"s_bfe_u32 %[t1], %[pk], 0x70008\n\t" \
		"s_bfm_b64 %[mask], %[t0], %[t1]\n\t" \
		OWNFILTER MM2C_FILTER2 \
		"s_and_b64 %[valid], vcc, %[mask]\n\t" \
		"s_cbranch_scc0 Lloop_%=\n\t" \
		MM2C_LC(MM2C_LB_OWN) \
		"s_mov_b32 %[d], 0\n\t" \
		"v_add_u32 " MM2C_R_F ", -14, %[own_f]\n\t" \
		"v_mov_b32 " MM2C_R_P ", %[own_p]\n\t" \
		"v_mov_b32 %[vb], %[ownst]\n\t" \
		SCORE \
		"s_branch Lhf_%=\n" \
		"Lloop_%=:\n\t" \
		MM2C_LC(MM2C_LB_LOOP) \
		MM2C_PROBE_LOOP \
		"s_sub_u32 %[n], %[n], 1\n\t" \
		"s_cbranch_scc1 Lpart_%=\n\t" \
		"s_waitcnt lgkmcnt(0)\n\t" \
		RFILTER NEXT_XQ MM2C_FILTER2 \
		"s_cbranch_vccz Lloop_%=\n\t" \
		"s_mov_b64 %[valid], vcc\n\t" \
		"s_sub_i32 %[d], %[nfull], %[n]\n" \
		"Lold_%=:\n\t" \
		MM2C_LC(MM2C_LB_OLD) \
		OLDADDR                                        /* the running address is two tiles further on: back to this tile's (in units of 8-byte slots) */ \
		"v_and_b32 %[u2], %[FMASK], %[vb]\n\t"          /* its slot in the f / p ring */ \
		"v_bfe_u32 %[vb], %[vb], 3, %[SBITS]\n\t"       /* its slot in the stamp ring */ \
		"s_cmp_gt_u32 %[d], %[NFI]\n\t" \
		"s_cbranch_scc1 Lfg_%=\n\t" \
		"ds_read_b64 " MM2C_R_FP ", %[u2] offset:%[FPOFF]\n\t" \
		SCORE \
		"s_waitcnt lgkmcnt(0)\n" \
		"Lhf_%=:\n\t" \
		MM2C_LC(MM2C_LB_HF) \
		SEG_HF \
		"Lmk_%=:\n\t"                                  /* exec = the lanes that passed the filters, until the fold has looked at the scores */ \
		"s_waitcnt lgkmcnt(0)\n\t" \
		"v_cmp_eq_u32 vcc, %[c], %[vb]\n" \
		"Lmk2_%=:\n\t" \
		ADDF \
		"s_and_b64 %[marked], vcc, exec\n\t" \
		"v_cmp_lt_i32 vcc, %[best], %[sc]\n\t" \
		"s_cbranch_vccnz Limp_%=\n\t" \
		"s_mov_b64 exec, %[ex]\n\t" \
		SEG_TAIL \
		"Lfg_%=:\n\t" \
		MM2C_LC(MM2C_LB_FG) \
		SEG_FG(SCORE) \
		"v_subrev_u32 " MM2C_R_P ", %[pbase], " MM2C_R_P "\n\t" \
		"v_max_i32 " MM2C_R_P ", -1, " MM2C_R_P "\n\t" \
		"v_add_u32 " MM2C_R_F ", -14, " MM2C_R_F "\n\t" \
		"s_branch Lhf_%=\n" \
		"Lpart_%=:\n\t" \
		MM2C_LC(MM2C_LB_PART) \
		"s_mov_b32 %[n], 0\n\t" \
		"s_cmp_eq_u32 %[part], 0\n\t" \
		"s_cbranch_scc1 Lend_%=\n\t" \
		"s_waitcnt lgkmcnt(0)\n\t" \
		RFILTER MM2C_FILTER2 \
		"s_sub_i32 %[t0], 64, %[part]\n\t" \
		"s_lshr_b64 %[mask], -1, %[t0]\n\t" \
		"s_mov_b32 %[part], 0\n\t" \
		"s_and_b64 %[valid], vcc, %[mask]\n\t" \
		"s_cbranch_scc0 Lend_%=\n\t" \
		MM2C_LC(MM2C_LB_PARTPASS) \
		"s_add_i32 %[d], %[nfull], 1\n\t" \
		BACK \
		"s_branch Lold_%=\n" \
		"Limp_%=:\n\t" \
		MM2C_LC(MM2C_LB_IMP) \
		"s_ff1_i32_b64 %[t0], %[valid]\n\t" \
		"v_readfirstlane_b32 %[t1], %[sc]\n\t"          /* the first lane that passed = the lowest active lane */ \
		"s_cmp_gt_i32 %[t1], %[best]\n\t" \
		"s_cbranch_scc0 Lslow_%=\n\t" \
		"v_cmp_lt_i32 vcc, %[t1], %[sc]\n\t" \
		"s_cbranch_vccnz Lslow2_%=\n\t" \
		"s_mov_b64 exec, %[ex]\n\t" \
		MM2C_LC(MM2C_LB_B0) \
		"s_mov_b32 %[best], %[t1]\n\t" \
		"s_lshl_b32 %[t1], %[d], 6\n\t" \
		"s_sub_i32 %[bestj], %[i063], %[t1]\n\t"   /* i0 + 63 - 64 d - lane */ \
		"s_sub_i32 %[bestj], %[bestj], %[t0]\n\t" \
		"s_sub_i32 %[nskip], %[nskip], 1\n\t" \
		"s_max_i32 %[nskip], %[nskip], 0\n\t" \
		"s_bitset0_b64 %[marked], %[t0]\n\t" \
		"s_bcnt1_i32_b64 %[t1], %[marked]\n\t" \
		"s_cbranch_scc0 " LNEXT "\n\t" \
		"s_add_i32 %[nskip], %[nskip], %[t1]\n\t" \
		"s_cmp_gt_i32 %[nskip], %[maxskip]\n\t" \
		"s_cbranch_scc1 Ldone_%=\n\t" \
		"s_branch " LNEXT "\n" \
		"Lslow2_%=:\n\t" \
		MM2C_LC(MM2C_LB_SLOW2) \
		"v_cmp_lt_i32 vcc, %[best], %[sc]\n" \
		"Lslow_%=:\n\t" \
		MM2C_LC(MM2C_LB_SLOW) \
		"s_mov_b64 exec, %[ex]\n\t"                     /* the general folds work on all lanes: the lanes that did not pass get the sentinel score */ \
		"v_cndmask_b32_e64 %[sc], %[sent], %[sc], %[valid]\n\t" \
		"s_lshl_b32 %[t0], %[d], 6\n\t" \
		"s_sub_i32 %[base], %[i0], %[t0]\n\t" \
		"s_cmp_lg_u64 %[marked], 0\n\t" \
		"s_cbranch_scc1 Lb2_%=\n\t" \
		"s_cmp_lg_u32 %[nskip], 0\n\t" \
		"s_cbranch_scc1 Lb2_%=\n\t" \
		"s_bcnt1_i32_b64 %[t0], vcc\n\t" \
		"s_cmp_eq_u32 %[t0], 1\n\t" \
		"s_cbranch_scc0 Lb1m_%=\n\t" \
		MM2C_LC(MM2C_LB_B1) \
		"s_ff1_i32_b64 %[t0], vcc\n\t" \
		"v_readlane_b32 %[best], %[sc], %[t0]\n\t" \
		"s_add_i32 %[t1], %[base], 63\n\t" \
		"s_sub_i32 %[bestj], %[t1], %[t0]\n\t" \
		"s_branch Lret_%=\n" \
		"Lb1m_%=:\n\t" \
		MM2C_LC(MM2C_LB_B1M) \
		"v_mov_b32 %[va], %[sc]\n\t" \
		MM2C_DPP_PREFIX_MAX("%[va]") \
		"s_nop 0\n\t" \
		"v_readlane_b32 %[best], %[va], 63\n\t" \
		"s_nop 1\n\t"                                   /* a VALU-written SGPR read by a VALU: two wait states on gfx940 / gfx950 (tools/check_isa_hazards.py, SGPR_VALU) */ \
		"v_cmp_eq_u32 vcc, %[best], %[sc]\n\t" \
		"s_ff1_i32_b64 %[t0], vcc\n\t" \
		"s_add_i32 %[t1], %[base], 63\n\t" \
		"s_sub_i32 %[bestj], %[t1], %[t0]\n\t" \
		"s_branch Lret_%=\n" \
		"Lb2_%=:\n\t" \
		MM2C_LC(MM2C_LB_B2) \
		"v_mov_b32 %[va], %[sc]\n\t" \
		MM2C_DPP_PREFIX_MAX("%[va]") \
		"v_bfrev_b32 %[vb], 1\n\t" \
		"s_nop 1\n\t" \
		"v_mov_b32_dpp %[vb], %[va] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_max_i32 %[vb], %[best], %[vb]\n\t" \
		"v_cmp_gt_i32_e64 %[nm], %[sc], %[vb]\n\t" \
		"s_andn2_b64 %[se], %[marked], %[nm]\n\t" \
		"s_cbranch_scc1 Lli_%=\n\t" \
		MM2C_LC(MM2C_LB_B2CLOSED) \
		"s_bcnt1_i32_b64 %[t0], %[nm]\n\t" \
		"s_sub_i32 %[nskip], %[nskip], %[t0]\n\t" \
		"s_max_i32 %[nskip], %[nskip], 0\n\t" \
		"s_mov_b32 %[last], 63\n\t" \
		"s_branch Ltk_%=\n" \
		"Lli_%=:\n\t" \
		MM2C_LC(MM2C_LB_LI) \
		"s_flbit_i32_b64 %[t0], %[nm]\n\t" \
		"s_sub_i32 %[t0], 63, %[t0]\n\t" \
		"s_ff1_i32_b64 %[t1], %[se]\n\t" \
		"s_cmp_lt_i32 %[t0], %[t1]\n\t" \
		"s_cbranch_scc0 Lgen_%=\n\t" \
		"s_bcnt1_i32_b64 %[t0], %[nm]\n\t" \
		"s_sub_i32 %[nskip], %[nskip], %[t0]\n\t" \
		"s_max_i32 %[nskip], %[nskip], 0\n\t" \
		"s_bcnt1_i32_b64 %[t0], %[se]\n\t" \
		"s_add_i32 %[t1], %[nskip], %[t0]\n\t" \
		"s_cmp_le_i32 %[t1], %[maxskip]\n\t" \
		"s_cbranch_scc0 Lcfb_%=\n\t" \
		MM2C_LC(MM2C_LB_LICLOSED) \
		"s_mov_b32 %[nskip], %[t1]\n\t" \
		"s_mov_b32 %[last], 63\n\t" \
		"s_branch Ltk_%=\n" \
		"Lcfb_%=:\n\t" \
		MM2C_LC(MM2C_LB_CFB) \
		"s_sub_i32 %[t0], %[maxskip], %[nskip]\n\t" \
		"s_max_i32 %[t0], %[t0], 0\n\t" \
		"s_mov_b64 vcc, %[se]\n\t" \
		"v_mbcnt_lo_u32_b32 %[vc], vcc_lo, 0\n\t" \
		"v_mbcnt_hi_u32_b32 %[vc], vcc_hi, %[vc]\n\t" \
		"v_cmp_eq_u32 vcc, %[t0], %[vc]\n\t" \
		"s_and_b64 %[nm], vcc, %[se]\n\t" \
		"s_ff1_i32_b64 %[last], %[nm]\n\t" \
		"s_sub_i32 %[last], %[last], 1\n\t" \
		"s_branch Ltk_%=\n" \
		"Lgen_%=:\n\t" \
		MM2C_LC(MM2C_LB_GEN) \
		"s_mov_b64 vcc, %[se]\n\t" \
		"v_mbcnt_lo_u32_b32 %[vc], vcc_lo, 0\n\t" \
		"v_mbcnt_hi_u32_b32 %[vc], vcc_hi, %[vc]\n\t" \
		"s_mov_b64 vcc, %[nm]\n\t" \
		"v_mbcnt_lo_u32_b32 %[u1], vcc_lo, 0\n\t" \
		"v_mbcnt_hi_u32_b32 %[u1], vcc_hi, %[u1]\n\t" \
		"v_sub_u32 %[vc], %[vc], %[u1]\n\t" \
		"v_cndmask_b32_e64 %[u1], 0, 1, %[se]\n\t" \
		"v_add_u32 %[vc], %[vc], %[u1]\n\t" \
		"v_cndmask_b32_e64 %[u1], 0, 1, %[nm]\n\t" \
		"v_sub_u32 %[vc], %[vc], %[u1]\n\t" \
		"v_sub_u32 %[vb], 0, %[vc]\n\t" \
		MM2C_DPP_PREFIX_MAX("%[vb]") \
		"v_max_i32 %[vb], %[nskip], %[vb]\n\t" \
		"v_add_u32 %[vb], %[vc], %[vb]\n\t" \
		"v_cmp_lt_i32 vcc, %[maxskip], %[vb]\n\t" \
		"s_and_b64 %[nm], vcc, %[se]\n\t" \
		"s_cbranch_scc1 Lbk_%=\n\t" \
		"s_nop 0\n\t" \
		"v_readlane_b32 %[nskip], %[vb], 63\n\t" \
		"s_mov_b32 %[last], 63\n\t" \
		"s_branch Ltk_%=\n" \
		"Lbk_%=:\n\t" \
		MM2C_LC(MM2C_LB_BK) \
		"s_ff1_i32_b64 %[last], %[nm]\n\t" \
		"s_sub_i32 %[last], %[last], 1\n" \
		"Ltk_%=:\n\t" \
		MM2C_LC(MM2C_LB_TK) \
		"s_cmp_lt_i32 %[last], 0\n\t" \
		"s_cbranch_scc1 Ldone_%=\n\t" \
		"v_readlane_b32 %[t0], %[va], %[last]\n\t" \
		"s_cmp_le_i32 %[t0], %[best]\n\t" \
		"s_cbranch_scc1 Laf_%=\n\t" \
		"s_mov_b32 %[best], %[t0]\n\t" \
		"v_cmp_eq_u32 vcc, %[t0], %[sc]\n\t" \
		"s_ff1_i32_b64 %[t1], vcc\n\t" \
		"s_add_i32 %[t0], %[base], 63\n\t" \
		"s_sub_i32 %[bestj], %[t0], %[t1]\n" \
		"Laf_%=:\n\t" \
		MM2C_LC(MM2C_LB_AF) \
		"s_cmp_eq_u32 %[last], 63\n\t" \
		"s_cbranch_scc1 Lret_%=\n\t" \
		"s_branch Ldone_%=\n" \
		SEG_END(SCORE, FARFILTER) \
		"Ldone_%=:\n\t" \
		MM2C_LC(MM2C_LB_DONE) \
		SEG_DONE \
		"s_mov_b64 exec, %[oh]\n\t"               /* commit into lane L (lane L - 1, the next anchor's, is written too: its own commit follows) ... */ \
		"v_mov_b32 %[own_f], %[best]\n\t" \
		"v_mov_b32 %[own_p], %[bestj]\n\t" \
		MM2C_READ_ANCHOR(SEG_RD, RDXQ)                     /* ... and the scalars of the next anchor from the lowest active lane, L - 1 */ \
		"s_mov_b64 exec, %[ex]\n\t" \
		"s_lshr_b64 %[oh], %[oh], 1\n\t" \
		"s_add_u32 %[c], %[c], 1\n\t"              /* carry out: that was the tile's last anchor */ \
		"s_cbranch_scc0 Lk_%=\n\t" \
		"s_branch Lexit_%=\n" \
		"Lspec_%=:\n\t" \
		MM2C_LC(MM2C_LB_SPEC) \
		"s_bitcmp1_b32 %[pk], 29\n\t" \
		"s_cbranch_scc1 Ldone_%=\n" \
		"Lexit_%=:\n\t" \
		"s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
		: [best] "=&s"(best), [bestj] "=&s"(bestj), [nskip] "=&s"(nskip), [n] "=&s"(n), [nfull] "=&s"(nfull), [part] "=&s"(part), [base] "=&s"(base), \
		  [t0] "=&s"(t0), [t1] "=&s"(t1), [last] "=&s"(last), [c] "=&s"(c), [pk] "=&s"(pk), [lo] "=&s"(lo), [xi1] "=&s"(xi1), [qi1] "=&s"(qi1), \
		  [span1] "=&s"(span1), [s16] "=&s"(s16), [d] "=&s"(d), [lo0] "=&s"(lo0), [fb] "=&s"(fb), [bpre] "=&s"(bpre), \
		  [mask] "=&s"(mask), [valid] "=&s"(valid), [mk] "=&s"(mk), [marked] "=&s"(marked), [nm] "=&s"(nm), [se] "=&s"(se), [ex] "=&s"(ex), [oh] "=&s"(oh), \
		  [dr] "=&v"(dr), [dq] "=&v"(dq), [dd] "=&v"(dd), [u1] "=&v"(u1), [u2] "=&v"(u2), \
		  [sc] "=&v"(sc), [va] "=&v"(va), [vb] "=&v"(vb), [vc] "=&v"(vc), [addr] "=&v"(addr), [s16v] "=&v"(s16v), [lom1v] "=&v"(lom1v), [fx] "=&v"(fx), [fq] "=&v"(fq), \
		  [own_f] "+v"(own_f), [own_p] "+v"(own_p) MM2C_PROBE_OPERAND MM2C_LC_OPERAND \
		: [i0] "s"(i0), [kstart] "s"(k_start), [cnt] "s"(cnt), [icnt1] "s"(i0 + cnt + 1), [i063] "s"(i0 + 63), [c200] "s"(0x200), [maxskip] "s"(max_skip), [avg] "s"(avg), [fptr] "s"(f), [pptr] "s"(p), [pbase] "s"(pbase), [aptr] "s"(a), [tptr] "s"(tg), \
		  [tx] "v"(tx), [tx1] "v"(tx1), [tq] "v"(tq), [tq1] "v"(tq1), [tspan] "v"(tspan), [tlo] "v"(tlo), [tlo0] "v"(tlo0), [tw] "v"(tw), \
		  [addr1] "v"(addr1), [addr2] "v"(addr2), [lomc] "v"(lomc), [ownst] "v"(ownst), [rl] "v"(rl), [mdqbw] "v"(mdqbw_v), [bw] "v"(bw_v), [sent] "v"(sent_v), \
		  [addr1q] "v"(addr1q), [selq] "v"(selq_v), [QHOFF] "n"(LY::QH), \
		  [XQOFF] "n"(LY::XQ), [FPOFF] "n"(LY::FP), [STOFF] "n"(LY::ST), [RBM1] "n"(LY::RB - 1), [FMASK] "n"(LY::FMASK), \
		  [SNM1] "n"(LY::SN - 1), [RMASK] "n"(64 * NX - 1), [SBITS] "n"(LY::SBITS), [NFI] "n"(NF), [GAPOFF] "n"(LYT::GAP), [NXM1] "n"(NX - 1), [REACH] "n"(64 * NX) \
		: "memory", "vcc", "scc", MM2C_R_X, MM2C_R_Q, MM2C_R_F, MM2C_R_P MM2C_FG_CLOB(CLOB)); \
	return cnt + c; \
}

// two instantiations of each: `lean` for tiles in which no window reaches beyond the LDS ring (no test for it anywhere in the loop, stamps written
// without touching exec), `far` for the others
#define MM2C_RING_W MM2C_XQ1_W, MM2C_NEXT_XQ_W, MM2C_RFILTER_W, MM2C_OLDADDR_W, MM2C_BACK_W, MM2C_OWNFILTER_W, MM2C_FARFILTER_W, MM2C_RDXQ_W, MM2C_FG_W, PF2
#define MM2C_RING_C MM2C_XQ1_C, MM2C_NEXT_XQ_C, MM2C_RFILTER_C, MM2C_OLDADDR_C, MM2C_BACK_C, MM2C_OWNFILTER_C, MM2C_FARFILTER_C, MM2C_RDXQ_C, MM2C_FG_PLAIN, PF0
#define MM2C_SCAN_TILE_ASM_(...) MM2C_SCAN_TILE_ASM(__VA_ARGS__)
#define MM2C_LEAN MM2C_RD_LEAN, MM2C_LK_LEAN, MM2C_HF_LEAN, MM2C_TAIL_LEAN, MM2C_END_LEAN, "", "Lloop_%="
#define MM2C_FARS MM2C_RD_FAR, MM2C_LK_FAR, MM2C_HF_FAR, MM2C_TAIL_FAR, MM2C_END_FAR, MM2C_DONE_FAR, "Lret_%="
MM2C_SCAN_TILE_ASM_(scan_tile_asm_cmp, false, false, MM2C_RING_W, MM2C_SCORE_CMP, MM2C_ADDF_CMP, MM2C_LEAN)
MM2C_SCAN_TILE_ASM_(scan_tile_asm_tab, true, false, MM2C_RING_W, MM2C_SCORE_TAB, MM2C_ADDF_TAB, MM2C_LEAN)
MM2C_SCAN_TILE_ASM_(scan_tile_asm_cmp_far, false, false, MM2C_RING_W, MM2C_SCORE_CMP, MM2C_ADDF_CMP, MM2C_FARS)
MM2C_SCAN_TILE_ASM_(scan_tile_asm_tab_far, true, false, MM2C_RING_W, MM2C_SCORE_TAB, MM2C_ADDF_TAB, MM2C_FARS)
// the same four over the compact x / q ring
MM2C_SCAN_TILE_ASM_(scan_tile_asm_cmp_c, false, true, MM2C_RING_C, MM2C_SCORE_CMP, MM2C_ADDF_CMP, MM2C_LEAN)
MM2C_SCAN_TILE_ASM_(scan_tile_asm_tab_c, true, true, MM2C_RING_C, MM2C_SCORE_TAB, MM2C_ADDF_TAB, MM2C_LEAN)
MM2C_SCAN_TILE_ASM_(scan_tile_asm_cmp_far_c, false, true, MM2C_RING_C, MM2C_SCORE_CMP, MM2C_ADDF_CMP, MM2C_FARS)
MM2C_SCAN_TILE_ASM_(scan_tile_asm_tab_far_c, true, true, MM2C_RING_C, MM2C_SCORE_TAB, MM2C_ADDF_TAB, MM2C_FARS)
// ... and over the q24 ring (the long ring of class-1 tasks): everything but a ring tile's request and filter is the 32-bit form's or the compact form's
#define MM2C_RING_Q MM2C_XQ1_Q, MM2C_NEXT_XQ_Q, MM2C_RFILTER_Q, MM2C_OLDADDR_C, MM2C_BACK_C, MM2C_OWNFILTER_W, MM2C_FARFILTER_W, MM2C_RDXQ_W, MM2C_FG_PLAIN, PF0
MM2C_SCAN_TILE_ASM_(scan_tile_asm_cmp_q, false, 2, MM2C_RING_Q, MM2C_SCORE_CMP, MM2C_ADDF_CMP, MM2C_LEAN)
MM2C_SCAN_TILE_ASM_(scan_tile_asm_tab_q, true, 2, MM2C_RING_Q, MM2C_SCORE_TAB, MM2C_ADDF_TAB, MM2C_LEAN)
MM2C_SCAN_TILE_ASM_(scan_tile_asm_cmp_far_q, false, 2, MM2C_RING_Q, MM2C_SCORE_CMP, MM2C_ADDF_CMP, MM2C_FARS)
MM2C_SCAN_TILE_ASM_(scan_tile_asm_tab_far_q, true, 2, MM2C_RING_Q, MM2C_SCORE_TAB, MM2C_ADDF_TAB, MM2C_FARS)

// ---------------------------------------------------------------- the kernel: one wave per task
// LDS rings before the own tile: x / q of NX tiles, f / p of the NF nearest (NF a power of two dividing NX).
// C16: the compact x / q ring (Lds<>), for the variants with the hand-written loop; the launcher picks it per task (cls bit 1 clear)
template <int NX, int NF, bool SKIP, bool GEN, bool GS1, bool FAR, bool TAB, int C16 /* the ring form: 0 32-bit slots, 1 compact (16 + 16 bits), 2 q24 (16 + 24 bits) */>
#ifdef MM2C_LABEL_COUNT
#define MM2C_WAVES_PER_SIMD(C16V, BYTES) 1
#else
#define MM2C_WAVES_PER_SIMD(C16V, BYTES) ((C16V) == 1 && (BYTES) <= 6144 ? 7 : (C16V) == 2 && (BYTES) <= 7424 ? 6 : 1)   /* q24 ring: 22 waves per CU by its LDS, at most 80 VGPRs then */
#endif
__global__ void __launch_bounds__(64, MM2C_WAVES_PER_SIMD(C16, (Lds<NX, NF, GEN, TAB, C16>::BYTES)))   // the compact ring leaves room for 7 waves per SIMD: at most 72 VGPRs then (it came out at 73)
chain_dp_tile(KParams P, int64_t n_tasks, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
              const uint4 *__restrict__ a_all, const float *__restrict__ avg_in, const int32_t *__restrict__ pbase_in,
              const int32_t *__restrict__ st_all, int32_t *__restrict__ f_all, int32_t *__restrict__ p_all, int32_t *__restrict__ t_all,
              int32_t *__restrict__ status, int only_flagged, const int64_t *__restrict__ ends, const int32_t *__restrict__ n_live,
              const uint8_t *__restrict__ cls, int my_cls, int cls_mask)
{
	static_assert(!C16 || (SKIP && !GEN && (GS1 || TAB)), "the compact and the q24 ring belong to the variants of the hand-written loop");
	static_assert(NF >= 1 && NF < NX && (NF & (NF - 1)) == 0 && (NX & (NX - 1)) == 0, "rings of a power of two of tiles, addressed with masks");
	static_assert(NF <= NX, "the f / p ring holds a prefix of the tiles of the x / q ring");
	static_assert(64 * (NX - 1) < 1024, "bef (anchors of older tiles inside the ring window) travels in the 10-bit field bits 15-24 of the per-anchor word");
	typedef Lds<NX, NF, GEN, TAB, C16> LY;
	constexpr int SN = LY::SN;
	constexpr bool ASMV = SKIP && !GEN && (GS1 || TAB);        // the hand-written scan covers this variant ...
	const bool ASM = ASMV && P.bw >= 0 && P.max_dq - 1 >= P.bw;   // ... when its three-instruction filter applies (max_dq - 1 >= bw: every preset)
	__shared__ __attribute__((aligned(16))) char lds[LY::BYTES];   // the kernel's only LDS object: the assembly addresses it from 0

	const int lane = threadIdx.x;
	const int64_t task = order ? (int64_t)__builtin_amdgcn_readfirstlane(order[blockIdx.x]) : (int64_t)blockIdx.x;
	if (task >= n_tasks) return;
	if (n_live && task >= (int64_t)*n_live) return;           // pieces cut on the device (chain_cut): the grid is sized for the worst case
	if (only_flagged && status[task] == 0) return;
	if (cls && (cls[task] & cls_mask) != my_cls) return;      // classes (chain_window_start: bit 0 ring size, bit 1 32-bit ring): this task belongs to another instantiation
	const int64_t base0 = offsets[task];
	const int n = __builtin_amdgcn_readfirstlane((int)((ends ? ends[task] : offsets[task + 1]) - base0));
	if (n <= 0) return;
	const uint4 *a = a_all + base0;        // {x lo, x hi, y lo (= query pos), y hi (span | flags | seg)}
	const int32_t *st = st_all + base0;
	int32_t *f = f_all + base0, *p = p_all + base0, *t = FAR ? t_all + base0 : nullptr;
	if (ASMV && (uint32_t)(uintptr_t)(void *)lds != 0) { if (lane == 0) status[task] = 3; return; }   // cannot happen: one LDS object per kernel

	const int pbase = pbase_in ? pbase_in[task] : 0;
	const int st_sub = ends ? pbase : 0;                      // device-cut pieces: st[] was computed for the whole task (task-relative)

	// avg_qspan_scaled, chain.c:48-49: from the prepass (or the caller); computed here only when neither supplied it
	float avg = avg_in ? avg_in[task] : -1.0f;
	if (avg < 0.f) {
		uint64_t sum = 0;
		for (int k = lane; k < n; k += 64) sum += (a[k].w & 0xffu);
		for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
		avg = (float)(__dmul_rn(.01, (double)(float)sum) / (double)n);
	}
	avg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, avg)));
	if (TAB) {
		// gap cost of chain.c:209,218-219 for every dd the filter lets through (dd <= bw <= 511), stored as 1 - cost
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
	int sent_v = SENT, mdqbw_v = P.max_dq - 1 - P.bw;        // the score of a dead lane; the bound of the one-compare filter
	int selq_v = 0x0c040302;                                 // q24 ring: the byte selector that puts {slot.b2, slot.b3, byte ring, 0} together (v_perm_b32)
	asm volatile("" : "+v"(X.mdq1_v), "+v"(X.bw_v), "+v"(sent_v), "+v"(mdqbw_v));   // per-lane copies: VALU operands from VGPRs issue at the full rate
	if (C16 == 2) asm volatile("" : "+v"(selq_v));           // (a register of its own only in the instantiation that uses it)
	TileMem M;
	M.lds = lds; M.a = a; M.f = f; M.p = p; M.t = t; M.pbase = pbase;

	int own_x = 0, own_q = 0, own_g = 0, own_f = 0, own_p = -1;   // the own tile: lane L = anchor i0 + 63 - L
	int seg0 = 0;
	bool t_ready = false;                                     // t[0 .. i0) has been zeroed (wave-uniform)
	const bool no_pairs = !GEN && (P.max_dq <= 0 || P.bw < 0);   // chain.c:203 / chain.c:205 (dd >= 0 > bw) let nothing through

#ifdef MM2C_LABEL_COUNT
	int lc_v = 0;                                             // lane b: how often the hand-written loop passed label b since the last flush
#endif
	uint4 cur = (rl < n) ? a[rl] : make_uint4(0, 0, 0, 0);
	int cur_st = (rl < n) ? st[rl] - st_sub : 0;
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int idx = i0 + rl;
		const int cnt = __builtin_amdgcn_readfirstlane(min(64, n - i0));
		uint4 nxt = make_uint4(0, 0, 0, 0); int nxt_st = 0;
		if (idx + 64 < n) { nxt = a[idx + 64]; nxt_st = st[idx + 64] - st_sub; }   // prefetch the next tile
		// x of anchor i0 - 1 (lane 0 of the tile before) for the equal-x test, before own_x is replaced
		int prev_last = rdlane(own_x, 0);
		own_x = (int)cur.x; own_q = (int)cur.z;
		own_g = (cur.w >> 16) & 0xff;                                         // MM_SEED_SEG_MASK mmpriv.h:22-23
		const int own_xq = (int)(((unsigned)own_x & 0xffffu) | ((unsigned)own_q << 16));               // compact ring: the low halves of x and q ...
		const int own_xq1 = (int)(((unsigned)(own_x - 1) & 0xffffu) | ((unsigned)(own_q - 1) << 16));   // ... and of x - 1 and q - 1
		(void)own_xq; (void)own_xq1;
		if (!GEN && !(P.flags & KF_IGNORE_SEG)) {
			// the simple variant assumes one segment id per task; anything else is redone by the general one
			if (i0 == 0) seg0 = rdlane(own_g, 63);
			if (BALLOT(rl < cnt && own_g != seg0)) { if (lane == 0) status[task] = 1; return; }
		}
		// stamps are one byte: 1 + the anchor's position in its tile.  They only mean something during the scan of the anchor that wrote them, so
		// the whole ring is wiped when a tile starts (two dword stores per lane) and a value identifies its anchor within the tile
		for (int s = lane; s < SN / 4; s += 64) ((int *)(lds + LY::ST))[s] = 0;
		const int stamp_lo = i0 - 64 * (NX - 1);   // oldest anchor reachable without global memory while this tile is processed
		{
			const int o = (idx & (SN - 1)) * LY::XS;   // the tile enters the x / q ring (its slot held the tile NX tiles back)
			if (C16) *(int *)(lds + LY::XQ + o) = own_xq;
			else *(int2 *)(lds + LY::XQ + o) = make_int2(own_x, own_q);
			if (C16 == 2) *(uint8_t *)(lds + LY::QH + (idx & (SN - 1))) = (uint8_t)((unsigned)own_q >> 16);   // q24 ring: bits 16-23 of q (q < 2^24 for every task of this instantiation)
			if (GEN) *(uint8_t *)(lds + LY::G + (o >> 3)) = (uint8_t)own_g;
		}
		if (FAR) {
			// global stamp scratch t[]: zeroed lazily, only once this task's windows can reach beyond the LDS ring
			const int reach = rdlane(cur_st, 63);                 // window start of the first anchor of the tile (st[] is monotone)
			if (!t_ready && reach < stamp_lo) {
				for (int z = lane; z < i0; z += 64) t[z] = 0;
				t_ready = true;
			}
			if (t_ready && idx < n) t[idx] = 0;
		}
		const int span_l = P.span_override >= 0 ? P.span_override : (int)(cur.w & 0xff);   // chain.c:189
		// anchors with the x of their predecessor (chain.c:202 `dr == 0`): bit L set <=> the anchor of lane L has the x of the anchor of lane L+1
		mask_t eq_prev = 0;
		if (!GEN) {
			asm volatile("" : "+v"(prev_last));
			const int px = __builtin_amdgcn_update_dpp(prev_last, own_x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
			eq_prev = BALLOT(px == own_x);
			if (i0 == 0) eq_prev &= ~(1ull << 63);
		}
		X.stamp_lo = stamp_lo;
		const int addr0 = ((idx - 64) & (SN - 1)) * LY::XS;   // per lane: byte offset of its anchor of the tile before in the x / q ring
		const int addr0b = ((idx - 128) & (SN - 1)) * LY::XS; // ... and of the tile before that
		// lean tiles (every window inside the ring): a stamp may go to any slot the ring holds, also one before the anchor's own window -- nothing
		// reads it during this anchor's scan and its value is this anchor's alone -- so the threshold below which a target is diverted is a constant of
		// the tile, the ring's oldest anchor; the sink, stamp_lo - 1 = i0 + 63 (mod SN), is the slot of the tile's last anchor, which no scan of this tile reads
		int lomc_v = stamp_lo - 1;
		asm volatile("" : "+v"(lomc_v));

		// per-anchor scalars of the tile, kept per lane (the hand-written loop fetches them with v_readlane): window start, LDS stamp, number
		// of own-tile predecessors inside the window; bit 31 of the latter marks the anchors that take the C++ path (x equal to the
		// predecessor's).  An anchor whose window reaches beyond the ring is scanned as far as the ring goes; only if that scan runs out without
		// the `break` of chain.c:231 does the hand-written loop hand the anchor back to the C++ path
		const int lo_l = no_pairs ? idx : min(cur_st, idx);
		// An anchor at the end of a run of e anchors with its x: the e lanes above it are not predecessors (dr == 0); a run that reaches the
		// tile's first anchor may go on in the tile before, and such an anchor takes the C++ path (per-lane test in every chunk)
		const mask_t above = ~(eq_prev >> lane);                  // bit 0: this anchor's x differs from its predecessor's, bit 1: the predecessor's from ...
		const int e_l = above ? (int)__builtin_ctzll(above) : 64;
		const int w_l = min(rl, idx - lo_l);                      // own-tile predecessors inside the window: lanes lane + 1 .. lane + w
		const int lo_c = max(lo_l, stamp_lo), bef_l = max(i0 - lo_c, 0);   // the window clamped to what the ring holds; its anchors in older tiles (<= 64 (NX - 1))
		int tw_l = max(w_l - e_l, 0) | (min(lane + 1 + e_l, 64) << 8) | (bef_l << 15);   // bits 0-5: lanes to scan, bits 8-14: the first of them, bits 15-24: bef
		if (lo_l >= idx) tw_l |= (int)0xa0000000;             // no window at all: bit 29, and bit 31 so that the loop needs one test for both rare cases
		if (FAR && lo_l < stamp_lo) tw_l |= 1 << 30;
		if (e_l > rl) tw_l |= (int)0x80000000;
		const bool tile_far = FAR && BALLOT((tw_l >> 30) & 1) != 0;
		const int ownst = idx & (SN - 1);                     // byte offset of the anchor's slot in the stamp ring
		const int tx1_l = own_x - 1, tq1_l = own_q - 1;

		for (int k = 0; k < cnt; ++k) {
			if (ASM) {
#define MM2C_CALL(FN, LO0) FN<NX, NF>(i0, __builtin_amdgcn_readfirstlane(k), cnt, P.max_skip, avg, f, p, pbase, a, t, own_x, tx1_l, own_q, tq1_l, span_l, lo_c, \
                                 LO0, tw_l, own_f, own_p, addr0, addr0b, lomc_v, ownst, rl, mdqbw_v, X.bw_v, sent_v, addr0 >> 2, selq_v MM2C_LC_ARG)
				if (C16 == 2) {
					// the q24 forms take what the 32-bit ones take (x, q, x - 1, q - 1 in full) + the byte ring's address of the tile before and the byte selector
					if (FAR && tile_far) k = TAB ? MM2C_CALL(scan_tile_asm_tab_far_q, lo_l) : MM2C_CALL(scan_tile_asm_cmp_far_q, lo_l);
					else k = TAB ? MM2C_CALL(scan_tile_asm_tab_q, lo_l) : MM2C_CALL(scan_tile_asm_cmp_q, lo_l);
				} else if (C16) {
					// the compact forms take packed words where the 32-bit ones take x and q: the tile's own {x, q} halves and the anchors' {x - 1, q - 1} halves
#define MM2C_CALLC(FN, LO0) FN<NX, NF>(i0, __builtin_amdgcn_readfirstlane(k), cnt, P.max_skip, avg, f, p, pbase, a, t, own_xq, own_xq1, own_xq, own_xq1, span_l, lo_c, \
                                 LO0, tw_l, own_f, own_p, addr0, addr0b, lomc_v, ownst, rl, mdqbw_v, X.bw_v, sent_v, 0, 0 MM2C_LC_ARG)
					if (FAR && tile_far) k = TAB ? MM2C_CALLC(scan_tile_asm_tab_far_c, lo_l) : MM2C_CALLC(scan_tile_asm_cmp_far_c, lo_l);
					else k = TAB ? MM2C_CALLC(scan_tile_asm_tab_c, lo_l) : MM2C_CALLC(scan_tile_asm_cmp_c, lo_l);
#undef MM2C_CALLC
				} else if (FAR && tile_far) k = TAB ? MM2C_CALL(scan_tile_asm_tab_far, lo_l) : MM2C_CALL(scan_tile_asm_cmp_far, lo_l);
				else k = TAB ? MM2C_CALL(scan_tile_asm_tab, lo_l) : MM2C_CALL(scan_tile_asm_cmp, lo_l);
#undef MM2C_CALL
#ifdef MM2C_LABEL_COUNT
				{	// this call's label hits go to the row of the instantiation that ran: compact << 2 | table << 1 | far
					const int row = (C16 == 1 ? 4 : 0) | (TAB ? 2 : 0) | ((FAR && tile_far) ? 1 : 0);   // (the q24 forms count into the 32-bit rows: the same labels, the same paths)
					if (lane < 32 && lc_v != 0) atomicAdd(&g_label_hits[row * 32 + lane], (unsigned long long)(unsigned)lc_v);
					lc_v = 0;
				}
#endif
				k = __builtin_amdgcn_readfirstlane(k);
				if (k >= cnt) break;
			}
			const int L = 63 - k;
			const int i = i0 + k;
			const int xi = rdlane(own_x, L), qi = rdlane(own_q, L);
			const int span_i = rdlane(span_l, L);
			const int lo = rdlane(lo_l, L);                                                      // chain.c:192-193
			Carry c = { span_i, -1, 0 };                                                         // chain.c:188-190
			if (i - lo > 0) {
				mask_t eq_run = 0; bool dr0 = false;
				if (!GEN && eq_prev != 0) {
					const mask_t r = eq_prev >> L;                    // bit 0: i has the x of i-1, bit 1: i-1 has the x of i-2, ... (k+1 bits)
					const int e = (int)__builtin_ctzll(~r);            // length of the run of equal x that ends at i
					if (e > k) dr0 = true;                              // it reaches beyond the tile: per-lane test in every chunk
					else if (e > 0) eq_run = ((1ull << e) - 1) << (L + 1);
				}
				X.xi1 = xi - 1; X.qi1 = qi - 1; X.span_i = span_i; X.span1_v = span_i - 1;
				if (GEN) X.seg_i = rdlane(own_g, L);                                                 // chain.c:191
				X.lo = lo; X.stamp = i + 1; X.s16 = 1 + k; X.s16_v = X.s16;             // LDS stamps: unique within the tile (the ring is wiped per tile)
				X.far_mode = FAR && lo < stamp_lo;
				if (!dr0) scan_anchor<NX, NF, SKIP, GEN, GS1, FAR, TAB, false, C16>(P, X, M, lane, i0, k, eq_run, own_x, own_q, own_g, own_f, own_p, addr0, c);
				else scan_anchor<NX, NF, SKIP, GEN, GS1, FAR, TAB, true, C16>(P, X, M, lane, i0, k, 0, own_x, own_q, own_g, own_f, own_p, addr0, c);
			}
			// ---- commit anchor i (chain.c:236) into its lane of the own tile
			write_lane2(own_f, own_p, __builtin_amdgcn_readfirstlane(c.best), __builtin_amdgcn_readfirstlane(c.best_j), L);
		}
		// ---- the finished tile: results leave in coalesced stores ...
		if (rl < cnt) { f[idx] = own_f; p[idx] = own_p < 0 ? own_p : own_p + pbase; }
		{
			const int o = (idx << 3) & LY::FMASK;    // ... and enters the f / p ring: what the hand-written score adds (f[j] and its constant term in one), and p
			*(int2 *)(lds + LY::FP + o) = make_int2(own_f - FBIAS, own_p);
		}
		cur = nxt; cur_st = nxt_st;
	}
}

} // namespace mm2c
#endif
