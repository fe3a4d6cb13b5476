// chain_dp_coop.h -- several waves per task: the chaining DP for passes that cannot fill the GPU with tasks (a lone run_chaining_on_hw / mm_chain_dp call,
// chain.c:103; the few long pieces of a small batch).  Included by chain_kernel.hip after chain_dp_tile.h, whose rings, filters, scan and hand-written loop it uses.
//
// The reference's device kernel scores 128 predecessors of ONE task per pipeline step (device/minimap2_opencl.cl:71-148); chain_dp_tile gives a task one wave,
// so a lone task is bound by that wave's latency (0.3-0.6 us per anchor).  What can run in parallel inside one task, exactly:
//   * Whether a predecessor j passes the filters of anchor i (chain.c:202-205) depends on x and q only, and the score of the pair (chain.c:207-220) on f[j],
//     which is final for every j in a tile before i's own.
//   * The scan order only matters through the early exit (chain.c:226-233): the `break` needs more than max_skip skip events, and a skip event is a candidate
//     that passed the filters -- so the scan always visits an anchor's first max_skip + 1 candidates (nearest first).  Whenever the BEST candidate of the whole
//     window (the maximum of f[j] + score, nearest j among equal scores) is among those first max_skip + 1, it is what the scan returns: it is visited, nothing
//     visited beats it, and the strict `>` of chain.c:226 keeps the nearest of equal scores.  The maximum over all candidates is an order-independent
//     reduction, and the rank of the best one -- how many candidates are nearer -- is a popcount over candidate masks.  That covers every anchor of a V2 call
//     (max_skip = INT_MAX, what run_chaining_on_hw computes), the noise anchors of a V1 call (few candidates) AND the anchors on a chain (many candidates, but
//     the best predecessor is one of the nearest): on the bench's streams every anchor.  The rare anchor whose best predecessor lies farther takes the exact scan.
// So a workgroup of W waves shares the task's LDS rings, and per tile of 64 anchors (lane L of every wave stands for the anchor i0 + 63 - L, as in chain_dp_tile):
//   phase A, all waves, PAIRS dealt by candidate: a candidate j (its x, q, f as scalars) is scored against the 64 anchors of the tile at once, one anchor per
//            lane -- filters, score, "inside this anchor's window" -- and each lane keeps the count of its candidates and the best (score, nearest index) among
//            them.  The candidates of the older tiles (final f) go to the waves in units of 16; the own tile's 64 candidates have no final f yet, so for them
//            the score WITHOUT f is stored as a 64 x 64 table in LDS.  Partial results per wave in LDS, merged by wave 0;
//   phase B, wave 0, anchor by anchor as before: an anchor whose count allows the short cut starts from the older tiles' best (or its span) and gets its own-tile
//            candidates PUSHED to it -- when anchor k becomes final, one table row + f[k] updates every later lane: a dependent chain of a few instructions per
//            anchor instead of a chunk scan; every other anchor -- too many candidates, a window that reaches beyond the ring, an equal-x run that reaches into the
//            tile before -- takes the exact scan: the hand-written loop of chain_dp_tile (the anchors that take the short cut carry bit 31 of their tw word,
//            "not for this loop") or its C++ restatement.
// Two workgroup barriers per tile; results bit-identical to chain_dp_tile (and so to chain.c:184-238): tests/test_gpu_parity.py runs the reference-kernel
// vectors and the parity inputs through this kernel as another route.
#ifndef MM2C_CHAIN_DP_COOP_H
#define MM2C_CHAIN_DP_COOP_H
#include "chain_dp_tile.h"

#ifndef MM2C_COOP_PROBE
#define MM2C_COOP_PROBE 0      // timing experiments (tools/probe_build.sh): 1 / 2 / 3 / 4 switch parts of the work off -- results are wrong then, never in the shipped library
#endif

namespace mm2c {

#ifndef MM2C_COOP_FAR_TILES
#define MM2C_COOP_FAR_TILES 4
#endif
constexpr int COOP_FAR_TILES = MM2C_COOP_FAR_TILES;   // tiles beyond the x / q ring whose candidates phase A still deals (from memory)
constexpr int COOP_NX = 16, COOP_NF = 8;    // rings of the cooperative kernel: 960 anchors of look-back in LDS, f / p of the 8 nearest tiles beside them
constexpr int COOP_NEVER = 0x7fffffff;      // candidate count of an anchor that must take the exact scan

// LDS behind the rings: per anchor of the tile (= per lane) the best of the older tiles' candidates as one 64-bit key (score << 32 | index: the waves merge their
// partial results with an LDS atomic maximum, equal scores -> the nearer index) and the candidate count (atomic add), then the own tile's table of pair scores
// without f (64 x 64 ints, row = candidate, column = lane of the anchor)
// Round 6: the candidate rings.  A candidate's x, q, f used to reach the 64 lanes of a pair row through v_readlane into SGPRs (eight candidates fetched by lanes 0 .. 7
// per unit): three scalar-side instructions per row plus the wait states behind them, in a kernel that the counters show to be bound by scalar-side issue (437 instructions
// per anchor on the long ava-ont reads, 60 % of them scalar-side: v_readlane, compares into SGPR pairs, exec juggling, branches).  Now every anchor of the last COOP_NC tiles has
// {x, q} (8 bytes) and f (4 bytes) in two rings of their own, written by wave 0 when the anchor's tile is final; a row reads them with ONE LDS address for all lanes (a
// broadcast read: no bank conflict, no scalar side), f only when some lane passed the filters.  32 tiles cover everything phase A deals (NX - 1 + COOP_FAR_TILES = 19 tiles back), so
// no candidate of phase A comes from memory any more.
#ifndef MM2C_COOP_PUSH2
#define MM2C_COOP_PUSH2 1
#endif
constexpr int COOP_NC = 32;
constexpr int COOP_ST_MAX = (2 * 64 * 64 * 4 + COOP_NC * 64 * 12) / 8;   // = 7 168: anchors of a task whose window starts the kernel makes itself (their x in the LDS of the tables and candidate rings)
template <int W> struct CoopLds { static constexpr int KEYS = 0, CNTS = 2 * 64 * 8, MASKS = CNTS + 2 * 64 * 4, PAIRS = MASKS + 2 * 2 * 64 * 8, CXQ = PAIRS + 2 * 64 * 64 * 4,
                                                       CF = CXQ + COOP_NC * 64 * 8, QCNT = CF + COOP_NC * 64 * 4, BYTES = QCNT + 16; };   // (two sets of summaries and two tables: see the schedule; QCNT: the group counters of phase A1, by tile parity)
static_assert(COOP_NC >= COOP_NX + COOP_FAR_TILES, "the candidate rings hold every tile phase A deals");

// ---- the rows of phase A that lie inside every anchor's window, by hand (round 6).  `n_rows` candidates, the first at LDS addresses lds_xq ({x, q}, 8 bytes) / lds_f
// (f, 4 bytes), the following ones at descending addresses (a segment of the candidate rings that does not wrap), candidate indices j_first, j_first - 1, ...;
// per lane: the anchor's x - 1, q - 1, span - 1, the bound's addend, and its running (best, index of best, candidates counted).  Per row:
//   the filters of chain.c:202-205 as in the hand-written loop of chain_dp_tile (two subtractions, saturating subtraction, v_sad, v_max, one compare)      7 instructions, then,
//   when some lane passed: count it, and test whether any such lane can still beat its best (f + span > best, see older_pairs)                            + 5,
//   when one can: the score of chain.c:207-209,218 with gap_scale 1 for every lane, SENT where the filter failed, strict maximum (nearest first)         + 15.
// The compiler's code for the same C++ (the `row` lambda of older_pairs, which still serves the rows with the window test, the tile before, and the gap-cost table)
// is 10 / + 7 / + 19 with two or three s_nop and a boolean round trip per row, and its own address arithmetic on the scalar unit -- in a kernel bound by
// scalar-side issue.  Four rows' reads are in flight; reads and waits are all inside the block (nothing in flight when it ends).  Temporaries: v100 .. v116.
#define MM2C_CROW(X, Q, F, JOFF, WAITF, LBL, FILT, D1) \
	"v_sub_u32 v112, %[tx1], " X "\n\t" \
	"v_sub_u32 v113, %[tq1], " Q "\n\t" \
	"v_sub_u32_e64 v114, v113, %[mdqbw] clamp\n\t" \
	"v_sad_u32 v115, v112, v113, 0\n\t" \
	FILT \
	"v_cmp_ge_u32 vcc, %[bw], v114\n\t" \
	"s_cbranch_vccz " LBL "\n\t" \
	WAITF \
	"v_add_u32 v114, " F ", %[spb]\n\t" \
	"v_cmp_ge_i32 %[st], v114, %[best]\n\t" \
	"v_addc_co_u32 %[cnt], %[sc2], 0, %[cnt], vcc\n\t"       /* (two instructions behind the branch: the wait states between a VALU write of VCC and a VALU that reads it) */ \
	D1 \
	"s_and_b64 %[st], %[st], vcc\n\t" \
	"s_cbranch_scc0 " LBL "\n\t" \
	"v_cvt_f32_i32 v116, v115\n\t" \
	"v_or_b32 v115, 1, v115\n\t" \
	"v_ffbh_u32 v115, v115\n\t" \
	"v_min3_i32 v112, v113, v112, %[sp1]\n\t" \
	"v_mul_f32 v116, %[avg], v116\n\t" \
	"v_cvt_i32_f32 v116, v116\n\t" \
	"v_lshrrev_b32 v115, 1, v115\n\t" \
	"v_add3_u32 v112, v112, v115, " F "\n\t" \
	"v_sub_u32 v112, v112, v116\n\t" \
	"v_add_u32 v112, -14, v112\n\t" \
	"v_cndmask_b32 v112, %[sent], v112, vcc\n\t" \
	"s_sub_i32 %[sj2], %[sj], " JOFF "\n\t" \
	"v_cmp_gt_i32 vcc, v112, %[best]\n\t" \
	"v_max_i32 %[best], v112, %[best]\n\t" \
	"v_mov_b32 v113, %[sj2]\n\t" \
	"v_cndmask_b32 %[jb], %[jb], v113, vcc\n" \
	LBL ":\n\t"
// the filter's last step.  Inside every window: the maximum of the two violations.  EDGE rows (before the window start of some anchor of the tile) add a third: how far the
// candidate lies before the lane's window start, (lov + r) - (j of the group's first row) saturated at 0, shifted beyond any bw the block is used with (bw < 2^20)
#define MM2C_CFILT_INNER "v_max_u32 v114, v114, v115\n\t"
#define MM2C_CFILT_EDGE(LOV) "v_sub_u32_e64 v117, " LOV ", %[sj] clamp\n\t" "v_lshlrev_b32 v117, 20, v117\n\t" "v_max3_u32 v114, v114, v115, v117\n\t"
// the tile before (phase A2, a group of exactly four rows per call): which of its anchors are a lane's candidates, as bit r of a word for row r of the group
#define MM2C_CD1(R) "v_cndmask_b32_e64 v116, 0, 1, vcc\n\t" "v_lshl_or_b32 %[ml], v116, " R ", %[ml]\n\t"
#define MM2C_CROWS_BODY(F0, F1, F2, F3, D0, D1, D2, D3) \
		"Lcr_grp_%=:\n\t" \
		"s_cmp_lt_i32 %[sn], 4\n\t" \
		"s_cbranch_scc1 Lcr_tail_%=\n\t" \
		"ds_read2_b64 v[100:103], %[pxl] offset0:3 offset1:2\n\t" \
		"ds_read2_b32 v[108:109], %[pfl] offset0:3 offset1:2\n\t" \
		"ds_read2_b64 v[104:107], %[pxl] offset0:1 offset1:0\n\t" \
		"ds_read2_b32 v[110:111], %[pfl] offset0:1 offset1:0\n\t" \
		"v_subrev_u32 %[pxl], 32, %[pxl]\n\t" \
		"v_subrev_u32 %[pfl], 16, %[pfl]\n\t" \
		"s_waitcnt lgkmcnt(3)\n\t" \
		MM2C_CROW("v100", "v101", "v108", "0", "s_waitcnt lgkmcnt(2)\n\t", "Lcr_a_%=", F0, D0) \
		MM2C_CROW("v102", "v103", "v109", "1", "s_waitcnt lgkmcnt(2)\n\t", "Lcr_b_%=", F1, D1) \
		"s_waitcnt lgkmcnt(1)\n\t" \
		MM2C_CROW("v104", "v105", "v110", "2", "s_waitcnt lgkmcnt(0)\n\t", "Lcr_c_%=", F2, D2) \
		MM2C_CROW("v106", "v107", "v111", "3", "s_waitcnt lgkmcnt(0)\n\t", "Lcr_d_%=", F3, D3) \
		"s_sub_i32 %[sj], %[sj], 4\n\t" \
		"s_sub_i32 %[sn], %[sn], 4\n\t" \
		"s_branch Lcr_grp_%=\n" \
		"Lcr_tail_%=:\n\t" \
		"s_cmp_lt_i32 %[sn], 1\n\t" \
		"s_cbranch_scc1 Lcr_end_%=\n\t" \
		"ds_read_b64 v[100:101], %[pxl] offset:24\n\t" \
		"ds_read_b32 v108, %[pfl] offset:12\n\t" \
		"v_subrev_u32 %[pxl], 8, %[pxl]\n\t" \
		"v_subrev_u32 %[pfl], 4, %[pfl]\n\t" \
		"s_waitcnt lgkmcnt(0)\n\t" \
		MM2C_CROW("v100", "v101", "v108", "0", "", "Lcr_e_%=", F0, D0) \
		"s_sub_i32 %[sj], %[sj], 1\n\t" \
		"s_sub_i32 %[sn], %[sn], 1\n\t" \
		"s_branch Lcr_tail_%=\n" \
		"Lcr_end_%=:\n\t" \
		"s_waitcnt lgkmcnt(0)\n\t"
#define MM2C_CROWS_CLOBBERS "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117"

__device__ __forceinline__ void coop_rows_inner(int lds_xq, int lds_f, int n_rows, int j_first, int tx1, int tq1, int sp1, int spb, int mdqbw, int bw, int sent,
                                                float avg, int &best, int &jb, int &cnt)
{
	int pxl = lds_xq - 24, pfl = lds_f - 12;                   // the lowest address of a group of four rows
	int sn = __builtin_amdgcn_readfirstlane(n_rows), sj = __builtin_amdgcn_readfirstlane(j_first), sj2;
	unsigned long long st, sc2;
	avg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, avg)));
	asm volatile(MM2C_CROWS_BODY(MM2C_CFILT_INNER, MM2C_CFILT_INNER, MM2C_CFILT_INNER, MM2C_CFILT_INNER, "", "", "", "")
		: [best] "+v"(best), [jb] "+v"(jb), [cnt] "+v"(cnt), [pxl] "+v"(pxl), [pfl] "+v"(pfl), [sn] "+s"(sn), [sj] "+s"(sj), [sj2] "=&s"(sj2), [st] "=&s"(st), [sc2] "=&s"(sc2)
		: [tx1] "v"(tx1), [tq1] "v"(tq1), [sp1] "v"(sp1), [spb] "v"(spb), [mdqbw] "v"(mdqbw), [bw] "v"(bw), [sent] "v"(sent), [avg] "s"(avg)
		: MM2C_CROWS_CLOBBERS);
}
// the same for the rows that need the window test (lov: the lane's window start; needs bw < 2^20 and windows of fewer than 2^11 anchors before the tile, which the
// tiles phase A deals guarantee)
__device__ __forceinline__ void coop_rows_edge(int lds_xq, int lds_f, int n_rows, int j_first, int tx1, int tq1, int sp1, int spb, int mdqbw, int bw, int sent,
                                               float avg, int lov, int &best, int &jb, int &cnt)
{
	int pxl = lds_xq - 24, pfl = lds_f - 12;
	int sn = __builtin_amdgcn_readfirstlane(n_rows), sj = __builtin_amdgcn_readfirstlane(j_first), sj2;
	unsigned long long st, sc2;
	const int lov1 = lov + 1, lov2 = lov + 2, lov3 = lov + 3;
	avg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, avg)));
	asm volatile(MM2C_CROWS_BODY(MM2C_CFILT_EDGE("%[lov0]"), MM2C_CFILT_EDGE("%[lov1]"), MM2C_CFILT_EDGE("%[lov2]"), MM2C_CFILT_EDGE("%[lov3]"), "", "", "", "")
		: [best] "+v"(best), [jb] "+v"(jb), [cnt] "+v"(cnt), [pxl] "+v"(pxl), [pfl] "+v"(pfl), [sn] "+s"(sn), [sj] "+s"(sj), [sj2] "=&s"(sj2), [st] "=&s"(st), [sc2] "=&s"(sc2)
		: [tx1] "v"(tx1), [tq1] "v"(tq1), [sp1] "v"(sp1), [spb] "v"(spb), [mdqbw] "v"(mdqbw), [bw] "v"(bw), [sent] "v"(sent), [avg] "s"(avg),
		  [lov0] "v"(lov), [lov1] "v"(lov1), [lov2] "v"(lov2), [lov3] "v"(lov3)
		: MM2C_CROWS_CLOBBERS);
}
// a group of exactly four rows of the tile before (n_rows == 4: the single-row loop behind the group, which has no row number, must not run); ml: bit r = row r counted
__device__ __forceinline__ void coop_rows4_d1(bool edge, int lds_xq, int lds_f, int j_first, int tx1, int tq1, int sp1, int spb, int mdqbw, int bw, int sent,
                                              float avg, int lov, int &best, int &jb, int &cnt, int &ml)
{
	int pxl = lds_xq - 24, pfl = lds_f - 12;
	int sn = 4, sj = __builtin_amdgcn_readfirstlane(j_first), sj2;
	unsigned long long st, sc2;
	const int lov1 = lov + 1, lov2 = lov + 2, lov3 = lov + 3;
	avg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, avg)));
	asm volatile("" : "+s"(sn));
	if (!edge)
		asm volatile(MM2C_CROWS_BODY(MM2C_CFILT_INNER, MM2C_CFILT_INNER, MM2C_CFILT_INNER, MM2C_CFILT_INNER, MM2C_CD1("0"), MM2C_CD1("1"), MM2C_CD1("2"), MM2C_CD1("3"))
			: [best] "+v"(best), [jb] "+v"(jb), [cnt] "+v"(cnt), [ml] "+v"(ml), [pxl] "+v"(pxl), [pfl] "+v"(pfl), [sn] "+s"(sn), [sj] "+s"(sj), [sj2] "=&s"(sj2), [st] "=&s"(st), [sc2] "=&s"(sc2)
			: [tx1] "v"(tx1), [tq1] "v"(tq1), [sp1] "v"(sp1), [spb] "v"(spb), [mdqbw] "v"(mdqbw), [bw] "v"(bw), [sent] "v"(sent), [avg] "s"(avg)
			: MM2C_CROWS_CLOBBERS);
	else
		asm volatile(MM2C_CROWS_BODY(MM2C_CFILT_EDGE("%[lov0]"), MM2C_CFILT_EDGE("%[lov1]"), MM2C_CFILT_EDGE("%[lov2]"), MM2C_CFILT_EDGE("%[lov3]"), MM2C_CD1("0"), MM2C_CD1("1"), MM2C_CD1("2"), MM2C_CD1("3"))
			: [best] "+v"(best), [jb] "+v"(jb), [cnt] "+v"(cnt), [ml] "+v"(ml), [pxl] "+v"(pxl), [pfl] "+v"(pfl), [sn] "+s"(sn), [sj] "+s"(sj), [sj2] "=&s"(sj2), [st] "=&s"(st), [sc2] "=&s"(sc2)
			: [tx1] "v"(tx1), [tq1] "v"(tq1), [sp1] "v"(sp1), [spb] "v"(spb), [mdqbw] "v"(mdqbw), [bw] "v"(bw), [sent] "v"(sent), [avg] "s"(avg),
			  [lov0] "v"(lov), [lov1] "v"(lov1), [lov2] "v"(lov2), [lov3] "v"(lov3)
			: MM2C_CROWS_CLOBBERS);
}
#undef MM2C_CROW
#undef MM2C_CROWS_BODY

// Round 6: a small per-read pass (mm2chain_host.cpp, the reference's call pattern chain_hardware.cpp:104-189) ends in this kernel -- it used to end in a fourth launch,
// stage_out, that copied f / p to the caller's page-locked buffer and raised the flag the caller polls.  The walker stores every finished tile to the host buffer as
// well as to the device arrays (the device copy is what the exact scans re-read), and the last workgroup to finish raises the flag behind a system-scope fence,
// exactly as stage_out did (host_stage.hip).  Only for a pass whose tasks cannot be flagged for the general variant (segment ids ignored): nothing rewrites f / p then.
struct CoopHostOut {
	int32_t *f = nullptr, *p = nullptr;       // the caller's result buffer (page-locked, mapped), indexed like f_all / p_all; nullptr: device arrays only
	unsigned *d_done = nullptr;               // workgroups finished (zero between passes: the last to be counted puts it back)
	unsigned *h_flag = nullptr; unsigned seq = 0;
};
// A single-launch pass of at most COOP_META_MAX pieces brings its metadata in the kernel's arguments (piece offsets, avg_qspan_scaled, p base): read through the scalar cache
// like any argument, instead of a trip to the pinned arena in front of the trip for the anchors.  n = 0: read offsets / avg_in / pbase_in as usual.
constexpr int COOP_META_MAX = 32;
struct CoopMeta { int32_t n = 0; int32_t pbase[COOP_META_MAX]; float avg[COOP_META_MAX]; int64_t off[COOP_META_MAX + 1]; };
__device__ __forceinline__ void coop_host_done(const CoopHostOut &H)
{
	if (!H.h_flag) return;
	__threadfence_system();                                  // this wave's stores to the host have left before the workgroup is counted (the walker is wave 0)
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned done = __hip_atomic_fetch_add(H.d_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		if (done == gridDim.x - 1) {
			__hip_atomic_store(H.d_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next pass (a single-launch pass has no stage_in to do it)
			__threadfence_system();
			__hip_atomic_store(H.h_flag, H.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

template <int W, bool GS1, bool FAR, bool TAB>
__global__ void __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(4, 4)))   // at most 128 VGPRs: sixteen waves per CU whichever the width (one workgroup of sixteen waves, two of eight)
chain_dp_coop(KParams P, int64_t n_tasks, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
              const uint4 *__restrict__ a_all, const float *__restrict__ avg_in, const int32_t *__restrict__ pbase_in,
              const int32_t *__restrict__ st_all, int32_t *__restrict__ f_all, int32_t *__restrict__ p_all, int32_t *__restrict__ t_all,
              int32_t *__restrict__ status, int only_flagged, const int64_t *__restrict__ ends, const int32_t *__restrict__ n_live, CoopHostOut H, int32_t *st_out, float *avg_out, const uint4 *a_src, CoopMeta MT)
{
	constexpr int NX = COOP_NX, NF = COOP_NF;
	constexpr bool SKIP = true, GEN = false;
	typedef Lds<NX, NF, GEN, TAB, false> LY;
	constexpr int SN = LY::SN;
	typedef CoopLds<W> CL;
	const bool ASM = P.bw >= 0 && P.max_dq - 1 >= P.bw;     // the hand-written loop's three-instruction filter applies (every preset)
	__shared__ __attribute__((aligned(16))) char lds[LY::BYTES + CL::BYTES];   // the kernel's only LDS object: the assembly addresses the rings from 0

	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int64_t task = order ? (int64_t)__builtin_amdgcn_readfirstlane(order[blockIdx.x]) : (int64_t)blockIdx.x;
	// every exit below is taken by all waves of the workgroup or by none: the conditions are the same values in every wave
	if (task >= n_tasks) { coop_host_done(H); return; }
	if (n_live && task >= (int64_t)*n_live) return;           // pieces cut on the device (chain_cut) that chain_route gave to this kernel: the grid is sized for the most it may give
	if (only_flagged && status[task] == 0) return;
	const bool mt = MT.n > 0;                                   // (then task < MT.n <= COOP_META_MAX: the grid is the pass's pieces)
	const int64_t base0_v = mt ? MT.off[task] : offsets[task];
	const int64_t base0 = (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base0_v >> 32)) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((int)base0_v));   // (the same in every lane; said so)
	const int n = __builtin_amdgcn_readfirstlane((int)((ends ? ends[task] : mt ? MT.off[task + 1] : offsets[task + 1]) - base0));
	// the one-word keys of the straight-line pushes (score << 7 | origin) need |score| < 2^23: at most 255 gained per link and a gap cost that cannot overflow
	// the word either (<= gap_scale * (2.55 * bw + 17) before the shift)
	const bool key32_ok = n < (1 << 15) && P.span_override <= 255 && P.gap_scale >= 0.f && P.gap_scale <= 4.f && P.bw <= (1 << 17);
	if (n <= 0) { coop_host_done(H); return; }
	const uint4 *a = a_all + base0;
	const int32_t *st = (st_out ? (const int32_t *)st_out : st_all) + base0;   // (st_out: the window starts are made below, by this workgroup)
	int32_t *f = f_all + base0, *p = p_all + base0, *t = FAR ? t_all + base0 : nullptr;
	if ((uint32_t)(uintptr_t)(void *)lds != 0) { if (threadIdx.x == 0) status[task] = 3; coop_host_done(H); return; }   // cannot happen: one LDS object per kernel
	int32_t *const hf = H.f ? H.f + base0 : nullptr, *const hp = H.p ? H.p + base0 : nullptr;

	const int pbase = mt ? MT.pbase[task] : pbase_in ? pbase_in[task] : 0;
	const int st_sub = ends ? pbase : 0;                      // device-cut pieces: st[] was computed for the whole task (task-relative), as in chain_dp_tile
	float avg = mt ? MT.avg[task] : avg_in ? avg_in[task] : -1.0f;
	if (avg < 0.f) {
		uint64_t sum = 0;
		for (int k = lane; k < n; k += 64) sum += (a[k].w & 0xffu);
		for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
		avg = (float)(__dmul_rn(.01, (double)(float)sum) / (double)n);
		if (avg_out && threadIdx.x == 0) avg_out[task] = avg;         // (for the second pass of tasks with several segment ids, which expects it in the workspace)
	}
	avg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, avg)));
	if (st_out) {
		// ---- the window starts of a SHORT task made here instead of by a prepass launch (a per-read pass: one launch less to submit and to wait for; round 6).  st[i] = max(first j
		// with x_i <= x_j + max_dist_x, i - max_iter), chain.c:192-193, the bounds and the condition of chain_window_start: the task's x (64 bits) go into LDS -- the
		// space of the pair tables and the candidate rings, not in use before the first tile: COOP_ST_MAX anchors -- and every thread searches there.
		// a_src (a single-launch pass, round 6): the anchors are still in the caller's pinned arena -- this sweep IS their upload (every workgroup its own task: what
		// stage_in did for the whole pass in a launch of its own), the copy in device memory is what the tiles below read.
		uint64_t *const s_xs = (uint64_t *)(lds + LY::BYTES + CL::PAIRS);
		int32_t *const so = st_out + base0;
		const uint4 *const src = a_src ? a_src + base0 : a;
		uint4 *const aw = a_src ? const_cast<uint4 *>(a_all) + base0 : nullptr;
		for (int i = (int)threadIdx.x; i < n; i += 64 * W) { const uint4 v = src[i]; if (aw) aw[i] = v; s_xs[i] = (uint64_t)v.y << 32 | v.x; }
		__syncthreads();
		const uint64_t D = (uint64_t)(int64_t)P.max_dist_x;
		for (int i = (int)threadIdx.x; i < n; i += 64 * W) {
			const uint64_t xi = s_xs[i];
			int hi = i, lo = max(i - P.max_iter, 0);
			while (lo < hi) {
				const int mid = (lo + hi) >> 1;
				if (xi > s_xs[mid] + D) lo = mid + 1; else hi = mid;               // chain.c:192 condition for "++st"
			}
			so[i] = lo;
		}
		__threadfence_block();
		__syncthreads();                                                         // st[] is read below by other threads than wrote it; the LDS space goes back to its owners
		if (a_src) asm volatile("" : "+s"(a) :: "memory");                       // (the anchors were WRITTEN through another name a moment ago: what is read through `a` from here on is not known to the compiler)
	}
	if (TAB && wv == 0) {
		int16_t *const s_gap = (int16_t *)(lds + LY::GAP);
		for (int dd = lane; dd <= P.bw && dd < 512; dd += 64) {
			const int lg = dd ? 31 - __builtin_clz((unsigned)dd) : 0;
			int g = (int)((float)dd * avg) + (lg >> 1);
			if (P.gap_scale != 1.0f) g = (int)__dadd_rn(__dmul_rn((double)g, (double)P.gap_scale), .499);   // chain.c:219
			s_gap[dd] = (int16_t)(1 - g);
		}
	}

#ifdef MM2C_COOP_PRIO
	if (wv == 0) __builtin_amdgcn_s_setprio(MM2C_COOP_PRIO);   // the walker is the wave every tile waits for: its instructions go first on its SIMD
#endif
	const int rl = 63 - lane;
	AnchorCtx X;
	X.avg = avg; X.rl = rl; X.seg_i = 0; X.far_mode = 0;
	X.mdq1_v = P.max_dq - 1; X.bw_v = P.bw;
	int sent_v = SENT, mdqbw_v = P.max_dq - 1 - P.bw;
	asm volatile("" : "+v"(X.mdq1_v), "+v"(X.bw_v), "+v"(sent_v), "+v"(mdqbw_v));
	TileMem M;
	M.lds = lds; M.a = a; M.f = f; M.p = p; M.t = t; M.pbase = pbase;
	long long *const s_key2 = (long long *)(lds + LY::BYTES + CL::KEYS);  // [tile parity][lane] best (score, index) over the older tiles' candidates
	int *const s_cnt2 = (int *)(lds + LY::BYTES + CL::CNTS);              // [tile parity][lane] candidates in the whole window
	unsigned long long *const s_own2 = (unsigned long long *)(lds + LY::BYTES + CL::MASKS);   // [tile parity][lane] bit k: candidate k of the anchor's own tile is one of its candidates
	unsigned long long *const s_d12 = s_own2 + 2 * 64;                                          // [tile parity][lane] bit c: anchor (tile start - 1 - c), in the tile before, is one of its candidates
	int *const s_pair2 = (int *)(lds + LY::BYTES + CL::PAIRS);            // [tile parity][candidate k of that tile][lane]
	int *const s_q2 = (int *)(lds + LY::BYTES + CL::QCNT);                // [tile parity] next group of candidates of phase A1

	int own_x = 0, own_q = 0, own_g = 0, own_f = 0, own_p = -1;
	int seg0 = 0;
	bool t_ready = false;
	const bool no_pairs = P.max_dq <= 0 || P.bw < 0;

#if MM2C_COOP_PROBE == 9
	long long tp[6] = {0, 0, 0, 0, 0, 0}, tq = wall_clock64();   // wave 0: 100 MHz ticks spent up to barrier 1 / in phase A2 / summary + flags / in the pushes / rest of phase B / tile end
	long long th[6] = {0, 0, 0, 0, 0, 0}, hq = wall_clock64();   // a helper wave: tile prologue / waiting at barrier 1 / its share of phase A2 / waiting at barrier 2 / phase A1 / the next tile's table
#define MM2C_HTICK(K) do { const long long tn_ = wall_clock64(); th[K] += tn_ - hq; hq = tn_; } while (0)
#else
#define MM2C_HTICK(K) do {} while (0)
#endif
	uint4 cur = (rl < n) ? a[rl] : make_uint4(0, 0, 0, 0);
	int cur_st = (rl < n) ? st[rl] - st_sub : 0;
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int idx = i0 + rl;
		const int cnt = __builtin_amdgcn_readfirstlane(min(64, n - i0));
		// The next tile's anchors and window starts, requested a tile ahead -- and NOT looked at before phase A1 (the helpers) or the end of the walk (wave 0): as
		// `if (in range) load, else 0` the compiler merged the loaded registers with the zeros under the lanes' mask right here, i.e. waited for the memory it had just
		// asked for, in every wave, at the top of every tile (0.6 us of a tile's 6.5).  So: an unconditional load from a clamped index, the values of lanes beyond the
		// task's end (copies of its last anchor: phase A1 computes rows for them that nobody reads) replaced by zeros only where `cur` is made from them, a tile later.
		const bool nxt_in = idx + 64 < n;
		const int nxt_i = min(idx + 64, n - 1);
		uint4 nxt = a[nxt_i]; int nxt_st = st[nxt_i];                     // (nxt_st: still relative to the task, st_sub comes off at the uses)
		int prev_last = rdlane(own_x, 0);
		own_x = (int)cur.x; own_q = (int)cur.z;
		own_g = (cur.w >> 16) & 0xff;
		if (!(P.flags & KF_IGNORE_SEG)) {
			if (i0 == 0) seg0 = rdlane(own_g, 63);
			if (BALLOT(rl < cnt && own_g != seg0)) { if (threadIdx.x == 0) status[task] = 1; coop_host_done(H); return; }   // (every wave sees the same tile: all leave together)
		}
		const int stamp_lo = i0 - 64 * (NX - 1);
		if (wv == 0) {
			{	// the summaries of the NEXT tile start empty (its older-tile pairs are dealt while this tile is walked); tile 0 has no older tiles, its set is emptied here too
				const int nb = ((i0 >> 6) + 1) & 1;
				s_key2[nb * 64 + lane] = (long long)((unsigned long long)(unsigned)SENT << 32); s_cnt2[nb * 64 + lane] = 0; s_own2[nb * 64 + lane] = 0; s_d12[nb * 64 + lane] = 0;
				if (lane == 0) s_q2[nb] = 0;
				if (i0 == 0) { s_key2[lane] = (long long)((unsigned long long)(unsigned)SENT << 32); s_cnt2[lane] = 0; s_own2[lane] = 0; s_d12[lane] = 0; }
			}
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
		X.stamp_lo = stamp_lo;
		const int addr0 = ((idx - 64) & (SN - 1)) * LY::XS;
		const int addr0b = ((idx - 128) & (SN - 1)) * LY::XS;
		int lomc_v = stamp_lo - 1;
		asm volatile("" : "+v"(lomc_v));
		const int lo_l = no_pairs ? idx : min(cur_st, idx);
		const int lo_c = max(lo_l, stamp_lo);
		int tw_l = 0;
		if (wv == 0) {                                               // what only the walker's exact scans read (the other waves deal pairs: x, q, span, window start)
			asm volatile("" : "+v"(prev_last));
			const int px = __builtin_amdgcn_update_dpp(prev_last, own_x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
			eq_prev = BALLOT(px == own_x);
			if (i0 == 0) eq_prev &= ~(1ull << 63);
			const mask_t above = ~(eq_prev >> lane);
			const int e_l = above ? (int)__builtin_ctzll(above) : 64;
			const int w_l = min(rl, idx - lo_l);
			const int bef_l = max(i0 - lo_c, 0);
			tw_l = max(w_l - e_l, 0) | (min(lane + 1 + e_l, 64) << 8) | (bef_l << 15);
			if (lo_l >= idx) tw_l |= (int)0xa0000000;
			if (FAR && lo_l < stamp_lo) tw_l |= 1 << 30;
			if (e_l > rl) tw_l |= (int)0x80000000;
		}
		const int ownst = idx & (SN - 1);
		const int tx1_l = own_x - 1, tq1_l = own_q - 1;

		MM2C_HTICK(0);
		__syncthreads();      // the rings hold x / q of this tile and f / p of the tiles before it (wave 0 wrote them); the summaries of the tile before have been read
		MM2C_HTICK(1);
#if MM2C_COOP_PROBE == 9
		if (wv == 0) { const long long tn = wall_clock64(); tp[1] += tn - tq; tq = tn; }
#endif

		// ---------------------------------------------------------------- phase A: every pair (candidate, anchor of a tile) once, a candidate per step, an anchor per lane
		// Schedule: the pairs of tile T with the candidates of tiles <= T - 2 are dealt while wave 0 walks tile T - 1 (phase A1, waves 1 .. W - 1, below); what is left
		// for this point is the tile before this one, whose f became final a moment ago (8 units of 8 candidates), and the own tile's table of scores without f.
		const int span1_l = span_l - 1;
		// One pair (candidate with low word of x = xj, q = qj -> a lane's anchor): the filters chain.c:202-205 as the hand-written loop has them -- saturating
		// subtraction against max_dq - 1 - bw, maximum with |dr - dq|, one compare with bw; dr == 0 (equal x) fails it too: dr - 1 is then 0xffffffff and either
		// |dr - dq| or the subtraction is huge -- and the score without f[j] (chain.c:207-209,218; gap_scale 1 or the table).
		auto pair_ok = [&](int dr1, int dq1, int dd) -> bool {
			int m;
			asm("v_max_u32 %0, %1, %2" : "=v"(m) : "v"(usat_sub(dq1, mdqbw_v)), "v"(dd));
			return (unsigned)m <= (unsigned)X.bw_v;
		};
		auto pair_score0 = [&](int dr1, int dq1, int dd, int sp1) -> int {
			if (TAB) {
				const int g = *(const int16_t *)(lds + LY::GAP + (min((unsigned)dd, 511u) << 1));
				return min3i(dq1, dr1, sp1) + g;
			}
			const int cz = __builtin_clz((unsigned)dd | 1u);
			return min3i(dq1, dr1, sp1) - 14 - (int)((float)dd * avg) + (cz >> 1);       // min(dq, dr, span) - (lin + (ilog2(dd) >> 1)): 1 - 15 = -14
		};
		// candidates j1 - 1 down to j0 of older tiles against the 64 anchors of the tile that starts at anchor t0 (per lane: x - 1, q - 1, span - 1, window start), in units
		// of 8 dealt to `nw` waves of which this is number `me`; results into that tile's set of summaries.  (A predecessor with equal x, dr == 0, is rejected by pair_ok: dr - 1 is
		// 0xffffffff and the unsigned |dr - dq| huge.)
		auto older_pairs = [&](int t0, int j0, int j1, int me, int nw, int tx1v, int tq1v, int sp1v, int lov, bool d1, int lo_max) {
			// best so far starts at the anchor's own span (chain.c:188): a candidate only matters to the maximum when it beats that, and whether a candidate counts
			// (cnt_l, the masks of the rank test) does not depend on its score
			const int spv = sp1v + 1;
			int best_l = spv, jb_l = -1, cnt_l = 0;
			unsigned long long m_l = 0;                          // d1: which anchors of the tile before t0 are candidates (bit c: anchor t0 - 1 - c)
			const char *const cxq = lds + LY::BYTES + CL::CXQ, *const cf = lds + LY::BYTES + CL::CF;
			constexpr int CM = 64 * COOP_NC - 1;
			// A pair scores at most f[j] + span (chain.c:207-220: min(dq, dr, span) minus a gap cost that is never negative while gap_scale >= 0), so a row in which
			// no lane that passed the filters can BEAT its best so far (strictly: the wave meets its candidates nearest first and chain.c:226 keeps the first of equal
			// scores) is only counted.  On a chain the nearest candidate of a wave's block has the highest f; the ones behind it are counted, not scored.
			const int spb = P.gap_scale >= 0.f ? sp1v : 0x3fffffff;   // (a negative gap_scale turns the gap cost into a gain: no bound -- every counted row is scored)
			const bool asm_rows = P.bw < (1 << 20);              // the hand-written rows (the edge form shifts its window violation beyond bw)
			// one row: candidate j (x, q, f: broadcast reads, one LDS address for all lanes) against the 64 anchors of the tile.  EDGE rows lie before the window start of
			// some anchor of the tile (j < lo_max = the window start of its last anchor: st[] is monotone) and carry the per-lane window test; the others are inside every window.
			auto row = [&](int j, int2 xq, int fj, bool edge) {
				const int dr1 = tx1v - xq.x, dq1 = tq1v - xq.y;
				const int dd = absdiff(dr1, dq1);
				bool ok = pair_ok(dr1, dq1, dd);
				if (edge) ok = ok & (j >= lov);                  // (`&`, and masks below: around `&&` of divergent values the compiler builds exec regions)
				const mask_t okm = BALLOT(ok);
				if (okm != 0) {                                  // (most candidates are candidates of none of the 64 anchors)
					cnt_l += ok ? 1 : 0;
					if (d1) m_l |= ok ? 1ull << (t0 - 1 - j) : 0ull;
					if ((okm & BALLOT(fj + spb >= best_l)) != 0) {   // f + span > best, as f + (span - 1) >= best
						int s0 = pair_score0(dr1, dq1, dd, sp1v);
						asm volatile("" : "+v"(s0));             // for every lane, whatever its filter said: nine instructions are cheaper than the exec mask around them
						const int sc = ok ? s0 + fj : SENT;
						const bool take = sc > best_l;           // a wave meets its candidates nearest first: strict, as chain.c:226 (the waves' results are merged by (score, index))
						best_l = max(sc, best_l); jb_l = take ? j : jb_l;
					}
				}
			};
			// `cnt` rows jt, jt - 1, ..., contiguous in the rings: one base address, the rows at constant offsets from it, eight rows' reads in flight
			auto seg = [&](int jt, int cnt_rows, bool edge) {
#ifndef MM2C_COOP_ROWS_CXX
				if constexpr (!TAB) {
					if (!d1 && asm_rows) {                       // the rows dealt a tile ahead: the hand-written blocks
						if (!edge) coop_rows_inner(LY::BYTES + CL::CXQ + ((jt & CM) << 3), LY::BYTES + CL::CF + ((jt & CM) << 2), cnt_rows, jt, tx1v, tq1v, sp1v, spb, mdqbw_v, X.bw_v,
						                           sent_v, avg, best_l, jb_l, cnt_l);
						else coop_rows_edge(LY::BYTES + CL::CXQ + ((jt & CM) << 3), LY::BYTES + CL::CF + ((jt & CM) << 2), cnt_rows, jt, tx1v, tq1v, sp1v, spb, mdqbw_v, X.bw_v,
						                    sent_v, avg, lov, best_l, jb_l, cnt_l);
						return;
					} else if (d1 && asm_rows && cnt_rows == 4) {   // the tile before: a whole group, with the candidate masks of the rank test
						int ml = 0;
						coop_rows4_d1(edge, LY::BYTES + CL::CXQ + ((jt & CM) << 3), LY::BYTES + CL::CF + ((jt & CM) << 2), jt, tx1v, tq1v, sp1v, spb, mdqbw_v, X.bw_v,
						              sent_v, avg, lov, best_l, jb_l, cnt_l, ml);
						m_l |= (unsigned long long)(unsigned)ml << (t0 - 1 - jt);   // row r of the group is anchor jt - r: bit t0 - 1 - jt + r
						return;
					}
				}
#endif
				const int2 *const px = (const int2 *)(cxq + ((jt & CM) << 3));
				const int *const pf = (const int *)(cf + ((jt & CM) << 2));
				int k = 0;
				for (; k + 8 <= cnt_rows; k += 8) {
					int2 c[8]; int fv[8];
#pragma unroll
					for (int u = 0; u < 8; ++u) { c[u] = px[-(k + u)]; fv[u] = pf[-(k + u)]; }
#pragma unroll
					for (int u = 0; u < 8; ++u) row(jt - k - u, c[u], fv[u], edge);
				}
				for (; k < cnt_rows; ++k) row(jt - k, px[-k], pf[-k], edge);
			};
			// rows [jz, ja) of one kind; a range that crosses the rings' wrap-around is two segments
			auto rows = [&](int ja, int jz, bool edge) {
				if (ja <= jz) return;
				const int jw = (ja - 1) & ~CM;                                  // first anchor of the ring revolution that holds ja - 1
				if (jw > jz) { seg(ja - 1, ja - jw, edge); seg(jw - 1, jw - jz, edge); }
				else seg(ja - 1, ja - jz, edge);
			};
			// the candidates of [j0, j1) in groups of four consecutive ones, dealt to the nw waves in turn, nearest group first (every wave meets near and far candidates,
			// chain and noise alike: contiguous blocks per wave left the waves with the far blocks 25 % behind); the part of a group at or behind jm needs no window test
			const int jm = min(max(lo_max, j0), j1);
#if MM2C_COOP_PROBE == 1 || MM2C_COOP_PROBE == 4
			if (0)
#endif
			if (d1) {
				// the tile before (every wave, between two barriers): one group of four each
				const int n_grp = (j1 - j0 + 3) >> 2;
				for (int g = me; g < n_grp; g += nw) {
					const int ja = j1 - 4 * g, jz = max(ja - 4, j0);
					rows(ja, max(jz, jm), false);
					rows(min(ja, jm), jz, true);
				}
			} else {
				// the tiles before that, dealt a tile ahead beside the walk.  Equal shares do not end together: the CU serves its waves oldest first, and with equal contiguous
				// blocks the youngest wave finished a quarter behind the oldest (measured per wave, round 6); taking every group of candidates from a counter costs a
				// third more than it balances (per group: the atomic, two window-start splits, a call of the hand-written block).  So: three quarters of the candidates in
				// equal contiguous blocks, the nearest to wave 0 -- and the farthest quarter in groups of eight TAKEN FROM A COUNTER in LDS by whoever is done with its
				// block (every wave still meets its candidates nearest first).
				const int n_all = j1 - j0;
				const int per = ((n_all * 3 / 4) / nw) & ~3;                       // rows of a static block (whole groups of four)
				const int jq = j1 - per * nw;                                       // [j0, jq): the queue's part
				int *const q = s_q2 + ((t0 >> 6) & 1);
				int gv = 0;
				if (lane == 0) gv = __hip_atomic_fetch_add(q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (requested before the block is worked on)
				if (per > 0) {
					const int ja = j1 - me * per, jz = ja - per;
					rows(ja, max(jz, jm), false);
					rows(min(ja, jm), jz, true);
				}
				const int n_grp = (jq - j0 + 7) >> 3;
				int g = __builtin_amdgcn_readfirstlane(gv);
				while (g < n_grp) {
					if (lane == 0) gv = __hip_atomic_fetch_add(q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					const int ja = jq - 8 * g, jz = max(ja - 8, j0);
					rows(ja, max(jz, jm), false);
					rows(min(ja, jm), jz, true);
					g = __builtin_amdgcn_readfirstlane(gv);
				}
			}
			const int sb = ((t0 >> 6) & 1) * 64;
			if (BALLOT(cnt_l != 0) != 0) {
				if (best_l > spv) __hip_atomic_fetch_max(&s_key2[sb + lane], (long long)(((unsigned long long)(unsigned)best_l << 32) | (unsigned)jb_l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				if (cnt_l != 0) __hip_atomic_fetch_add(&s_cnt2[sb + lane], cnt_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				if (d1 && m_l != 0) __hip_atomic_fetch_or(&s_d12[sb + lane], m_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		};
		// the table of a tile (first anchor t0, `tn` anchors; x, q per lane in xv, qv): for candidate k and the lane of a later anchor, the pair's score without f, or SENT
		// when k is not one of that anchor's candidates (the candidates' f is not final when the table is made); the candidates are counted on the way
		auto own_table = [&](int t0, int tn, int me, int nw, int xv, int qv, int sp1v, int lov) {
			int cnt_l = 0;
			unsigned long long m_l = 0;
			int *const tab = s_pair2 + ((t0 >> 6) & 1) * (64 * 64);
#if MM2C_COOP_PROBE == 2 || MM2C_COOP_PROBE == 4
			if (0)
#endif
			for (int k = me; k < tn; k += nw) {
				const int Lk = 63 - k;
				const int dr1 = xv - 1 - rdlane(xv, Lk), dq1 = qv - 1 - rdlane(qv, Lk);
				const int dd = absdiff(dr1, dq1);
				const bool ok = pair_ok(dr1, dq1, dd) & (dr1 != -1) & (t0 + k >= lov) & (lane < Lk);
				int s0 = SENT;
				if (BALLOT(ok) != 0) {                           // (a noise candidate is nobody's: its row is SENT throughout)
					cnt_l += ok ? 1 : 0;
					m_l |= ok ? 1ull << k : 0ull;
					int sc = pair_score0(dr1, dq1, dd, sp1v);
					asm volatile("" : "+v"(sc));
					s0 = ok ? sc : SENT;
				}
				tab[k * 64 + lane] = s0;
			}
			if (cnt_l != 0) {
				__hip_atomic_fetch_add(&s_cnt2[((t0 >> 6) & 1) * 64 + lane], cnt_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__hip_atomic_fetch_or(&s_own2[((t0 >> 6) & 1) * 64 + lane], m_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		};
		{
			// ---- the tile before this one (the part of it inside the first anchor's window and the ring); the first tile's table (the later ones are made a tile ahead, below)
			const int lo_first = rdlane(lo_l, 63);
			const int jmin = max(max(lo_first, stamp_lo), 0);
			if (i0 > 0) older_pairs(i0, max(jmin, i0 - 64), i0, wv, W, tx1_l, tq1_l, span1_l, lo_l, true, rdlane(lo_l, 64 - cnt));   // (lo_max: the window start of the tile's last anchor)
			else own_table(0, cnt, wv, W, own_x, own_q, span1_l, lo_l);
		}
		int *const s_pair = s_pair2 + ((i0 >> 6) & 1) * (64 * 64);          // this tile's table
		MM2C_HTICK(2);
		__syncthreads();
		MM2C_HTICK(3);
#if MM2C_COOP_PROBE == 9
		if (wv == 0) { const long long tn = wall_clock64(); tp[2] += tn - tq; tq = tn; }
#endif

		// ---------------------------------------------------------------- phase B: wave 0 walks the tile's anchors; the other waves go on to the next barrier
		// An anchor that takes the short cut only needs the maximum over its candidates, so its own-tile candidates are PUSHED to it: when anchor k is final, every
		// later short-cut anchor of the tile (one per lane) scores the pair (k -> itself) and keeps the better of the two -- the dependent chain per anchor is
		// read f[k], add, compare, instead of a whole chunk scan.  Candidates arrive in ascending j, the reference scans in descending j and keeps the first of
		// equal scores (strict `>`, chain.c:226): so an equal score from a later (nearer) candidate replaces an earlier one, but never the anchor's own span
		// (p = -1), which only a strictly better score beats -- one 64-bit comparison, see `acc` below.
		if (wv == 0) {
			const int sbuf = ((i0 >> 6) & 1) * 64;
			const long long key = s_key2[sbuf + lane];          // this lane's anchor: best of the older tiles' candidates (score << 32 | index), candidates in the whole window
			const int bo_l = (int)(key >> 32), jo_l = (int)(unsigned)key, c_l = s_cnt2[sbuf + lane];
			const unsigned long long own_m = s_own2[sbuf + lane], d1_m = s_d12[sbuf + lane];   // which anchors of the own tile / of the tile before are its candidates
			// WHICH ANCHORS TAKE THE SHORT CUT.  The `break` of chain.c:231 needs more than max_skip skip events and every skip event is a candidate, so the scan always
			// visits an anchor's first max_skip + 1 candidates (in scan order: nearest first).  The maximum over ALL candidates (nearest index among equal scores) is
			// therefore the scan's result whenever that best candidate is among the first max_skip + 1: it is visited, nothing visited beats it, and the nearest of equal
			// scores is the one the strict `>` of chain.c:226 keeps.  Its rank = the number of candidates nearer than it, counted from the candidate masks of the own
			// tile and of the tile before (deeper: bounded by the candidate count).  No candidate better than the span (p = -1): true of every visited subset as well.
			// The rank is known only once the anchor's maximum is, so the anchors are walked as if all of them qualified and each is checked when it becomes final;
			// the few that fail (a best predecessor more than max_skip candidates away) take the exact scan.  Not eligible at all: a window that reaches further back
			// than the tiles phase A deals.  (An equal-x run that reaches into the tile before is no obstacle: a predecessor with equal x fails the pair filter -- dr - 1
			// is 0xffffffff, the unsigned |dr - dq| huge -- in every tile; only the hand-written loop leaves such anchors to the C++ scan.)
			const bool tent_l = rl < cnt && !(FAR && lo_l < stamp_lo - 64 * COOP_FAR_TILES && lo_l < idx) && !no_pairs;
			mask_t tents = BALLOT(tent_l);
			auto rank_ok = [&](int j) -> bool {                 // per lane: is candidate j (this lane's best) among the first max_skip + 1 of its scan?
				int r;
				if (j >= i0) { const int kb = j - i0 + 1; r = kb >= 64 ? 0 : (int)__builtin_popcountll(own_m >> kb); }
				else if (j >= i0 - 64) r = (int)__builtin_popcountll(own_m) + (int)__builtin_popcountll(d1_m & ((1ull << (i0 - 1 - j)) - 1ull));
				else r = c_l - 1;
				return j == -1 || (int64_t)r <= (int64_t)P.max_skip;
			};
			if (tent_l && !(lo_l >= idx)) tw_l |= (int)0x80000000;   // "not for the hand-written loop" (it commits the anchors without a window itself, bit 29): it hands these anchors back
			const bool tile_far = FAR && BALLOT((tw_l >> 30) & 1) != 0;
			// best so far = the older tiles' best or the span (chain.c:188).  Score and index travel as ONE signed 64-bit key, score << 32 | index: a maximum over keys
			// prefers the higher score and, among equal scores, the higher = nearer index; the span's key carries index 0xffffffff (p = -1), so an equal score never
			// replaces it (chain.c:226 is strict)
			auto mk_key = [](int sc, int j) -> long long { return (long long)(((unsigned long long)(unsigned)sc << 32) | (unsigned)j); };
			const long long acc0 = bo_l > span_l ? mk_key(bo_l, jo_l) : mk_key(span_l, -1);
			long long acc = acc0;
			// A short-cut anchor's result IS its accumulator once every earlier anchor of the tile has been pushed and its rank has been checked; it is copied into the
			// own-tile registers (what the exact scans read, and what leaves the tile) for all such lanes at once, before an exact scan runs and at the end of the tile.
			mask_t copied = 0;
			auto flush = [&](int k_done) {                       // anchors 0 .. k_done - 1 are final
				const mask_t fin = tents & (k_done >= 64 ? ~0ull : ~(~0ull >> k_done)) & ~copied;   // lanes 63 .. 64 - k_done
				if (fin == 0) return;
				own_f = sel(fin, own_f, (int)(acc >> 32)); own_p = sel(fin, own_p, (int)(unsigned)acc);
				copied |= fin;
			};
#if MM2C_COOP_PROBE == 9
			{ const long long tn = wall_clock64(); tp[3] += tn - tq; tq = tn; }
#endif
			// A whole tile of eligible anchors: the 63 pushes as a loop with nothing else in it -- the lane that holds f[k] is a constant of the instruction, the
			// table rows come in ahead of their use -- and the ranks checked for all lanes at once afterwards; one failing lane sends the tile through the walk below.
			bool all_done = false;
#if MM2C_COOP_PROBE != 3 && MM2C_COOP_PROBE != 4
			if (cnt == 64 && tents == ~0ull) {
				if constexpr (MM2C_COOP_PUSH2 != 0 && W == 16) {
					// Two sweeps over the table instead of one that carries (score, origin) together.  With sixteen waves the tile waits for THIS chain -- anchor k + 1 cannot be
					// pushed before anchor k is final -- so the chain carries the score alone: read a lane, add, maximum (three dependent instructions a push; with the origin
					// packed into the same word it is six: mask, add the code, select against SENT).  A row of SENT needs no select: SENT + f stays below every score (f >= 0).
					// The origins come from a second sweep with every f final, whose rows do not wait for one another: the nearest candidate whose score + f EQUALS the
					// maximum (chain.c:226 is strict and the reference scans nearest first: of equal scores the nearest wins), none if the anchor's own span holds the maximum
					// (p = -1 only loses to a HIGHER score), the older tiles' best if no candidate of this tile reaches it.  No bound on the scores (the packed form needs
					// |score| < 2^23).  255 reads of 10^6 anchors 107.1 -> 102.7 ms; with eight waves (two workgroups per CU) the walk is not what a tile waits for and the
					// second sweep only adds instructions (113.9 against 113.1 ms): they keep the one sweep below.
					const int S0 = (int)(acc >> 32);
					const bool from_span = (int)(unsigned)acc == -1;
					int F = S0;
#pragma unroll 4
					for (int k = 0; k < 63; ++k) {               // (anchor 63 has nobody after it)
						const int row_k = s_pair[k * 64 + lane];
						F = max(F, row_k + rdlane(F, 63 - k));
					}
					int org = -1;
					for (int k0 = 0; k0 < 63; k0 += 7) {         // seven rows' reads in flight (63 = 9 x 7)
						int r[7];
#pragma unroll
						for (int u = 0; u < 7; ++u) r[u] = s_pair[(k0 + u) * 64 + lane];
#pragma unroll
						for (int u = 0; u < 7; ++u) org = r[u] + rdlane(F, 63 - k0 - u) == F ? k0 + u : org;
					}
					acc = mk_key(F, (from_span && F == S0) ? -1 : org >= 0 ? i0 + org : from_span ? -1 : jo_l);
				} else if (key32_ok) {
					// Scores of a task of fewer than 2^15 anchors stay below 2^23 in size when a chain gains at most 255 per anchor (the 8-bit span of chain.c:189; key32_ok
					// rules out a larger q_span_override and a negative gap_scale, under which a link can ADD its gap cost), so score and origin fit ONE word:
					// score << 7 | code, code 0 = the older tiles' best, 1 + k = candidate k of this tile, 127 = the span itself (p = -1).  A signed maximum then is the whole
					// rule: the higher score, among equal scores the nearer origin, and never an equal score over the span.  A push is: read a lane, mask, add, maximum.
					int a32 = (int)(acc >> 32) * 128 + ((int)(unsigned)acc == -1 ? 127 : 0);
#pragma unroll 4
					for (int k = 0; k < 63; ++k) {               // (anchor 63 has nobody after it)
						const int row_k = s_pair[k * 64 + lane];
						const int fk7 = (rdlane(a32, 63 - k) & ~127) + (k + 1);             // f[k] << 7 | code of candidate k
						const int key32 = row_k != SENT ? row_k * 128 + fk7 : SENT;
						a32 = max(a32, key32);
					}
					const int code = a32 & 127;
					acc = mk_key(a32 >> 7, code == 127 ? -1 : code == 0 ? jo_l : i0 + code - 1);
				} else {
#pragma unroll 4
					for (int k = 0; k < 63; ++k) {
						const int row_k = s_pair[k * 64 + lane];
						const long long key64 = mk_key(row_k + rdlane((int)(acc >> 32), 63 - k), i0 + k);
						acc = (row_k != SENT && key64 > acc) ? key64 : acc;
					}
				}
				if (BALLOT(!rank_ok((int)(unsigned)acc)) == 0) { all_done = true; own_f = (int)(acc >> 32); own_p = (int)(unsigned)acc; }
				else acc = acc0;                                 // some anchor's best predecessor lies beyond its first max_skip + 1 candidates: the walk below sorts it out
			}
#endif
			int row = cnt > 0 ? s_pair[lane] : SENT;             // row k of the pair table, requested one anchor ahead
			for (int k = all_done ? cnt : 0; k < cnt;) {
				const int L = 63 - k;
				if (tents >> L & 1) {
					// ---- everything that can reach this anchor has been pushed: its maximum is final.  Does the short cut hold for it?
					const int row_k = row;
					if (k + 1 < cnt) row = s_pair[(k + 1) * 64 + lane];
					if (BALLOT(rank_ok((int)(unsigned)acc)) >> L & 1) {
#if MM2C_COOP_PROBE != 3 && MM2C_COOP_PROBE != 4
						if ((tents & (L > 0 ? ~0ull >> (64 - L) : 0ull)) != 0) {           // push it on: candidate k -> the eligible anchors after it
							const long long key64 = mk_key(row_k + rdlane((int)(acc >> 32), L), i0 + k);
							acc = (row_k != SENT && key64 > acc) ? key64 : acc;
						}
#endif
						++k;
						continue;
					}
					tents &= ~(1ull << L);                         // no: the exact scan for this anchor (the hand-written loop takes it once bit 31 of its word is cleared)
					tw_l = lane == L ? tw_l & 0x7fffffff : tw_l;
				}
				flush(k);
				int k2 = k;
				if (ASM) {
#define MM2C_CALL(FN, LO0) FN<NX, NF>(i0, __builtin_amdgcn_readfirstlane(k), cnt, P.max_skip, avg, f, p, pbase, a, t, own_x, tx1_l, own_q, tq1_l, span_l, lo_c, \
                                 LO0, tw_l, own_f, own_p, addr0, addr0b, lomc_v, ownst, rl, mdqbw_v, X.bw_v, sent_v, 0, 0 MM2C_LC_ARG)
#ifdef MM2C_LABEL_COUNT
					int lc_v = 0;
#endif
					if (FAR && tile_far) k2 = TAB ? MM2C_CALL(scan_tile_asm_tab_far, lo_l) : MM2C_CALL(scan_tile_asm_cmp_far, lo_l);
					else k2 = TAB ? MM2C_CALL(scan_tile_asm_tab, lo_l) : MM2C_CALL(scan_tile_asm_cmp, lo_l);
#undef MM2C_CALL
					k2 = __builtin_amdgcn_readfirstlane(k2);
				}
				if (k2 == k) {
					// ---- an anchor the hand-written loop does not take (or no hand-written loop for these scalars): the C++ restatement of the exact scan
					const int i = i0 + k;
					const int xi = rdlane(own_x, L), qi = rdlane(own_q, L);
					const int span_i = rdlane(span_l, L);
					const int lo = rdlane(lo_l, L);
					Carry c = { span_i, -1, 0 };                                                     // chain.c:188-190
					if (i - lo > 0) {
						X.xi1 = xi - 1; X.qi1 = qi - 1; X.span_i = span_i; X.span1_v = span_i - 1;
						mask_t eq_run = 0; bool dr0 = false;
						if (eq_prev != 0) {
							const mask_t r = eq_prev >> L;
							const int e = (int)__builtin_ctzll(~r);
							if (e > k) dr0 = true;
							else if (e > 0) eq_run = ((1ull << e) - 1) << (L + 1);
						}
						X.lo = lo; X.stamp = i + 1; X.s16 = 1 + k; X.s16_v = X.s16;
						X.far_mode = FAR && lo < stamp_lo;
						if (!dr0) scan_anchor<NX, NF, SKIP, GEN, GS1, FAR, TAB, false, false>(P, X, M, lane, i0, k, eq_run, own_x, own_q, own_g, own_f, own_p, addr0, c);
						else scan_anchor<NX, NF, SKIP, GEN, GS1, FAR, TAB, true, false>(P, X, M, lane, i0, k, 0, own_x, own_q, own_g, own_f, own_p, addr0, c);
					}
					write_lane2(own_f, own_p, __builtin_amdgcn_readfirstlane(c.best), __builtin_amdgcn_readfirstlane(c.best_j), L);
					k2 = k + 1;
				}
				// the anchors just made final by an exact scan, pushed to the eligible anchors after them
#if MM2C_COOP_PROBE != 3 && MM2C_COOP_PROBE != 4
				for (int kk = k; kk < k2 && kk < cnt; ++kk) {
					const int Lk = 63 - kk;
					if ((tents & (Lk > 0 ? ~0ull >> (64 - Lk) : 0ull)) == 0) break;
					const int s0 = s_pair[kk * 64 + lane];
					const long long key64 = mk_key(s0 + rdlane(own_f, Lk), i0 + kk);
					acc = (s0 != SENT && key64 > acc) ? key64 : acc;
				}
#endif
				k = k2;
				if (k < cnt) row = s_pair[k * 64 + lane];
			}
#if MM2C_COOP_PROBE == 9
			{ const long long tn = wall_clock64(); tp[4] += tn - tq; tq = tn; }
#endif
			if (!all_done) flush(64);
			// (The next tile's anchors, requested at the top of this iteration and not looked at by this wave since: they have long arrived, and saying so HERE -- before
			// the stores below, which vmcnt would count in front of them -- keeps the wait for `cur = nxt` at the end of the iteration from waiting for the stores as well.)
			__builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
			// ---- the finished tile: results leave in coalesced stores and enter the f / p ring
			if (rl < cnt) {
				const int pv = own_p < 0 ? own_p : own_p + pbase;
				f[idx] = own_f; p[idx] = pv;
				if (hf) { hf[idx] = own_f; hp[idx] = pv; }          // a per-read pass: straight to the caller's buffer as well
			}
			const int o = (idx << 3) & LY::FMASK;
			*(int2 *)(lds + LY::FP + o) = make_int2(own_f - FBIAS, own_p);
			// ... and the candidate rings of phase A (the slots of the tile COOP_NC back: the last rows that read them were dealt a tile ago, for this very tile)
			*(int2 *)(lds + LY::BYTES + CL::CXQ + ((idx & (64 * COOP_NC - 1)) << 3)) = make_int2(own_x, own_q);
			*(int *)(lds + LY::BYTES + CL::CF + ((idx & (64 * COOP_NC - 1)) << 2)) = own_f;
		} else if (i0 + 64 < n) {
			// ---------------------------------------------------------------- phase A1 for the NEXT tile, beside wave 0's walk of this one: its anchors (already in `nxt`) against
			// the candidates of the tiles before this one -- final f, and ring slots nobody writes during the walk (wave 0 puts this tile's f / p into the slot of the tile
			// NF + 1 back from the next one, which the next tile reads from memory; the x / q of the next tile enter the ring after the barrier)
			const int t0 = i0 + 64, idn = t0 + rl;
			const int lo_n = no_pairs ? idn : min(nxt_st - st_sub, idn);
			const int lo_first_n = rdlane(lo_n, 63);
			// (FAR: up to COOP_FAR_TILES tiles beyond the ring are dealt as well, their x / q from memory, so that an anchor whose window reaches a little further back
			// than the ring -- the 1 024 anchors of a V2 scan are up to 17 tiles -- keeps the short cut)
			const int jmin_n = max(max(lo_first_n, t0 - 64 * (NX - 1 + (FAR ? COOP_FAR_TILES : 0))), 0);
			const int sp_n = (P.span_override >= 0 ? P.span_override : (int)(nxt.w & 0xff)) - 1;
			if (jmin_n < i0) older_pairs(t0, jmin_n, i0, wv - 1, W - 1, (int)nxt.x - 1, (int)nxt.z - 1, sp_n, lo_n, false, rdlane(lo_n, 64 - min(64, n - t0)));   // (candidates from the candidate rings: their slots are final a tile before the walker reuses them)
			MM2C_HTICK(4);
			own_table(t0, min(64, n - t0), wv - 1, W - 1, (int)nxt.x, (int)nxt.z, sp_n, lo_n);   // x and q only: nothing of it waits for this tile's walk
			MM2C_HTICK(5);
		}
#if MM2C_COOP_PROBE == 9
		if (wv == 0) { const long long tn = wall_clock64(); tp[5] += tn - tq; tq = tn; }
#endif
		cur = nxt_in ? nxt : make_uint4(0, 0, 0, 0); cur_st = nxt_in ? nxt_st - st_sub : 0;
	}
	coop_host_done(H);
#if MM2C_COOP_PROBE == 9
	if (lane == 0 && wv >= 1 && task == 0) printf("coop helper wave %d ticks: prologue %lld, at barrier 1 %lld, A2 %lld, at barrier 2 %lld, A1 %lld, table %lld\n", wv, th[0], th[1], th[2], th[3], th[4], th[5]);
	if (threadIdx.x == 0 && task == 0) printf("coop ticks (100 MHz) n=%d: to barrier1 %lld, A2 %lld, summary %lld, pushes %lld, rest of B %lld, tile end %lld\n", n, tp[1], tp[2], tp[3], tp[4], tp[5], tp[0]);
#endif
}

} // namespace mm2c
#endif
