// issue_rate.hip -- development probe: wave64 instructions one gfx950 SIMD issues per ns, by instruction kind and waves per SIMD.
// (the chaining DP is instruction-issue bound; this pins the denominators used in DESIGN.md)   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define R4(s) s s s s
#define R16(s) R4(s) R4(s) R4(s) R4(s)
#define R32(s) R16(s) R16(s)

template <int MODE>
__global__ void __launch_bounds__(256) k_issue(int iters, int *out)
{
	__shared__ int lds[4096];
	int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
	int s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
	unsigned long long m0 = 0x123456789abcdefull + blockIdx.x, m1 = ~m0;
	int addr = (threadIdx.x * 8) & 4095 * 4;
	lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1;
	__syncthreads();
	for (int it = 0; it < iters; ++it) {
		if (MODE == 0) asm volatile(R16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(it));
		if (MODE == 1) asm volatile(R16("s_add_u32 %0, %0, %2\n s_add_u32 %1, %1, %2\n") : "+s"(s0), "+s"(s1) : "s"(it));
		if (MODE == 2) asm volatile(R16("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n") : "+s"(m0), "+s"(m1) :: "scc");
		if (MODE == 3) asm volatile(R32("s_nop 0\n"));
		if (MODE == 4) asm volatile(R16("s_cmp_lg_u32 %0, 0\n s_cbranch_scc0 1f\n 1:\n s_cmp_lg_u32 %0, 1\n s_cbranch_scc0 2f\n 2:\n") :: "s"(s0 | 4) : "scc");   // not taken: 64 instr
		if (MODE == 5) asm volatile(R32("v_readlane_b32 %0, %1, 3\n") : "=s"(s0) : "v"(a0));
		if (MODE == 6) asm volatile(R32("s_waitcnt lgkmcnt(0)\n"));
		if (MODE == 7) asm volatile(R16("v_cmp_lt_u32 %0, %2, %3\n v_cmp_gt_u32 %1, %2, %3\n") : "=s"(m0), "=s"(m1) : "v"(a0), "v"(a1));
		if (MODE == 8) asm volatile(R16("v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %1, %1, %0, %2\n") : "+v"(a0), "+v"(a1) : "s"(m0));
		if (MODE == 9) asm volatile(R32("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a0));
		if (MODE == 10) asm volatile(R16("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(*(long long *)&m0), "=v"(*(long long *)&m1) : "v"(addr) : "memory");
		if (MODE == 11) asm volatile(R32("ds_bpermute_b32 %0, %1, %0\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(a0) : "v"(addr) : "memory");
		if (MODE == 12) asm volatile(R16("v_cmp_lt_u32 %0, %2, %3\n s_and_b64 %1, %1, %0\n") : "=s"(m0), "+s"(m1) : "v"(a0), "v"(a1) : "scc");   // 16 VALU + 16 SALU
		if (MODE == 13) asm volatile(R16("v_cmp_lt_u32 vcc, %0, %1\n s_cbranch_vccz 1f\n 1:\n") :: "v"(a0), "v"(a1) : "vcc");   // 16 + 16
		if (MODE == 14) asm volatile(R16("v_writelane_b32 %0, %1, 5\n v_writelane_b32 %0, %1, 6\n") : "+v"(a0) : "s"(s0));
		if (MODE == 15) asm volatile(R16("s_bcnt1_i32_b64 %0, %1\n s_ff1_i32_b64 %0, %1\n") : "=s"(s0) : "s"(m0) : "scc");
		if (MODE == 16) asm volatile(R16("v_mbcnt_lo_u32_b32 %0, %1, 0\n v_mbcnt_hi_u32_b32 %0, %2, %0\n") : "=v"(a0) : "s"((int)m0), "s"((int)(m0 >> 32)));
		if (MODE == 17) asm volatile(R16("ds_write_b16 %0, %1\n ds_read_u16 %1, %0 offset:2\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(a0) : "memory");
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + s0 + s1 + s2 + s3 + (int)m0 + (int)m1;
}

template <int MODE> static void run(const char *name, int per_iter, int cus, int *d_out)
{
	const int iters = 4000;
	for (int wps = 1; wps <= 8; wps *= 2) {              // blocks of 256 threads = 4 waves = one per SIMD; wps blocks per CU
		hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
		float ms = 0;
		for (int rep = 0; rep < 2; ++rep) {
			CK(hipEventRecord(e0));
			hipLaunchKernelGGL(k_issue<MODE>, dim3(cus * wps), dim3(256), 0, 0, iters, d_out);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			CK(hipEventElapsedTime(&ms, e0, e1));
		}
		printf("%-34s waves/SIMD %d: %7.3f ms -> %.3f wave-instr per SIMD per ns\n", name, wps, ms, (double)per_iter * iters * wps / (ms * 1e6));
	}
}

int main()
{
	int *d_out; CK(hipMalloc(&d_out, 1 << 24));
	hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
	const int cus = pr.multiProcessorCount;
	printf("%d CUs, clock %d kHz\n", cus, pr.clockRate);
	run<0>("v_add_u32", 32, cus, d_out);
	run<1>("s_add_u32", 32, cus, d_out);
	run<2>("s_and_b64 / s_or_b64", 32, cus, d_out);
	run<3>("s_nop 0", 32, cus, d_out);
	run<4>("s_cmp + s_cbranch (not taken)", 64, cus, d_out);
	run<5>("v_readlane_b32", 32, cus, d_out);
	run<6>("s_waitcnt (nothing pending)", 32, cus, d_out);
	run<7>("v_cmp -> sgpr pair", 32, cus, d_out);
	run<8>("v_cndmask sgpr mask", 32, cus, d_out);
	run<9>("v_max_i32_dpp row_shr", 32, cus, d_out);
	run<10>("ds_read_b64", 32, cus, d_out);
	run<11>("ds_bpermute_b32", 32, cus, d_out);
	run<12>("v_cmp + s_and_b64 (16+16)", 32, cus, d_out);
	run<13>("v_cmp vcc + s_cbranch_vccz (16+16)", 32, cus, d_out);
	run<14>("v_writelane_b32", 32, cus, d_out);
	run<15>("s_bcnt1 / s_ff1 b64", 32, cus, d_out);
	run<16>("v_mbcnt lo+hi", 32, cus, d_out);
	run<17>("ds_write_b16 + ds_read_u16 (16+16)", 32, cus, d_out);
	return 0;
}
