// host_stage.hip -- the two copies of a small per-read pass done by kernels instead of copy commands (round 5).
//
// A pass of the call combiner (mm2chain_host.cpp: the reference's pattern, one blocking run_chaining_on_hw / mm_chain_dp call per read, chain_hardware.cpp:104-189
// = two clEnqueueWriteBuffer, the kernel, two clEnqueueReadBuffer, clFinish) used to be: hipMemcpyAsync up, the kernels, hipMemcpyAsync down, hipStreamSynchronize.
// On the GPU's time line that is copy 7.5 us + 8-15 us until the first kernel starts (the copy runs on another engine: cross-queue signals) + kernels + 7-9 us
// until the copy back starts + 5.3 us, and the host learns about the end through the runtime's signal wait (gpurun_out/r5_trace, profiles/r5_per_read.md).  Both
// buffers are page-locked host memory that the GPU can address, so:
//   stage_in  : the upload arena [anchors | piece offsets | order | p base | avg | status ...] is read from the host's staging buffer by a kernel (coalesced 16-byte
//               loads over PCIe) and written to the device arena the DP kernels work on; the same stream, no engine change, no cross-queue wait.
//   stage_out : f / p are written from the device arena into the host's result buffer, and the LAST workgroup to finish writes the pass number into a flag word in
//               host memory after a system-scope fence; the host thread polls that word.
// The DP kernels themselves never touch host memory (their look-back re-reads f / p and x / q: that must stay in HBM / L2).
#include "chain_kernel.h"
#include <algorithm>

namespace mm2c {

__global__ void __launch_bounds__(256)
stage_in(const uint4 *__restrict__ h_src, uint4 *__restrict__ d_dst, int64_t n16, unsigned *__restrict__ d_done)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) *d_done = 0;         // stage_out's counter of finished workgroups (ordered before it by the stream)
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
		d_dst[i] = h_src[i];
}

__global__ void __launch_bounds__(256)
stage_out(const uint4 *__restrict__ d_src, uint4 *__restrict__ h_dst, int64_t n16, unsigned *__restrict__ d_done, unsigned *__restrict__ h_flag, unsigned seq)
{
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
		h_dst[i] = d_src[i];
	// every store of this workgroup has left for the host before the workgroup is counted; the last one raises the flag
	__threadfence_system();
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned done = __hip_atomic_fetch_add(d_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		if (done == gridDim.x - 1) {
			__hip_atomic_store(d_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next pass (which may have no stage_in)
			__threadfence_system();
			__hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

static unsigned stage_grid(int64_t n16) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n16 + 255) / 256, 512)); }

hipError_t launch_stage_in(const void *h_src, void *d_dst, size_t bytes, unsigned *d_done, hipStream_t st)
{
	const int64_t n16 = (int64_t)((bytes + 15) / 16);
	hipLaunchKernelGGL(stage_in, dim3(stage_grid(n16)), dim3(256), 0, st, (const uint4 *)h_src, (uint4 *)d_dst, n16, d_done);
	return hipGetLastError();
}

hipError_t launch_stage_out(const void *d_src, void *h_dst, size_t bytes, unsigned *d_done, unsigned *h_flag, unsigned seq, hipStream_t st)
{
	const int64_t n16 = (int64_t)((bytes + 15) / 16);
	hipLaunchKernelGGL(stage_out, dim3(stage_grid(n16)), dim3(256), 0, st, (const uint4 *)d_src, (uint4 *)h_dst, n16, d_done, h_flag, seq);
	return hipGetLastError();
}

hipError_t warm_stage_kernels()
{
	hipFuncAttributes at;
	return hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&stage_in));
}

} // namespace mm2c
