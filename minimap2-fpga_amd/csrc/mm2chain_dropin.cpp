// mm2chain_dropin.cpp -- the reference's dispatch surface, symbol for symbol (chain_hardware.h:68-71).
//
// The reference builds every .c with $(CXX) (Makefile:184-185), so chain.o / main.o import C++-mangled
//   _Z18run_chaining_on_hwliiiifP7mm128_tPiS1_Phliff   _Z13hardware_initlPc   _Z7cleanupv
// This TU defines exactly those, over the C ABI of include/mm2chain.h, so an unmodified reference chain.o /
// main.o links against libmm2chain_hip.so instead of chain_hardware.o + XRT.  run_chaining_on_hw keeps the
// contract of chain_hardware.cpp:27-197 (fills f[0..n), p[0..n) before returning 0) and computes what the
// FPGA kernel computes (V2, device/minimap2_opencl.cl): its signature cannot carry max_skip / max_iter /
// gap_scale / n_segs, see mm2c_chain_task_host for the extended entry used by our own mm_chain_dp.
#include <cstdio>
#include <cstdlib>
#include <string>
#include "mm2chain.h"

typedef struct { uint64_t x, y; } mm128_t;   // minimap.h:53

// hardware_init's first argument: the host's BUFFER_N (main.c:367 passes chain_hardware.h:64's 5 187 500).  The reference sizes its device buffers by it and refuses a
// longer call (chain_hardware.cpp:34-37); this library's buffers grow on demand, but a host that states a size keeps the reference's contract for it.  0: no size stated.
static long g_buffer_n = 0;

int run_chaining_on_hw(long n, int max_dist_x, int max_dist_y, int bw, int q_span, float avg_qspan,
                       mm128_t *a, int *f, int *p, unsigned char *num_subparts, long total_subparts, int tid,
                       float hw_time_pred, float sw_time_pred)
{
	(void)num_subparts; (void)total_subparts;   // FPGA pipeline bookkeeping (chain.c:62-78), not needed on a GPU
	if (n == 0) return 0;                       // chain_hardware.cpp:30-32
	if (g_buffer_n > 0 && n > g_buffer_n) {     // chain_hardware.cpp:34-37: same message, same exit code
		fprintf(stderr, "Error: The size of the call (n = %ld) exceeds buffer size (%ld). Process this read on SW?\n", n, g_buffer_n);
		exit(1);
	}
	mm2c_params_t par;
	mm2c_params_fpga_v2(&par, max_dist_x, max_dist_y, bw, q_span);
	// the busy protocol of chain_hardware.cpp:54-75 (PROCESS_ON_SW_IF_HW_BUSY): 1 = declined, the caller's own loop runs (chain.c:106,112-164).  The caller of THIS
	// symbol is the reference's chain.o, which has that loop; predictions that are not positive (a caller without the model) never decline.
	int rc = mm2c_chain_task_host_pred(&par, n, (const mm2c_anchor_t *)a, avg_qspan, f, p, tid, hw_time_pred, sw_time_pred);
	if (rc == 1) return 1;                      // chain_hardware.cpp:75
	if (rc != 0) {                              // chain_hardware.cpp:34-37, 208-235: message + exit
		fprintf(stderr, "Error: GPU chaining failed (n = %ld): %s\n", n, mm2c_last_error());
		exit(EXIT_FAILURE);
	}
	return 0;                                   // chain_hardware.cpp:195
}

bool hardware_init(long buf_size, char *binary_name)
{
	(void)binary_name;                          // xclbin path (main.c:367): no bitstream
	g_buffer_n = buf_size > 0 ? buf_size : 0;   // BUFFER_N: the longest call run_chaining_on_hw accepts (the buffers themselves grow on demand)
	// MM2C_ASYNC_INIT=1: the runtime start-up (0.2 s) runs on a thread of its own while main.c:371-399 reads the index; a device that then turns out to be missing
	// ends the process at the first chaining call (message + exit, as chain_hardware.cpp:208-235) instead of making this function return false
	const char *as = getenv("MM2C_ASYNC_INIT");
	if (as && atoi(as) != 0) return mm2c_init_async(-1) == 0;
	if (mm2c_init(-1) != 0) {
		fprintf(stderr, "ERROR: %s\n", mm2c_last_error());
		return false;                           // main.c:367-369 returns -1 on false
	}
	return true;
}

void cleanup() { mm2c_shutdown(); }             // main.c:430

// chain_hardware.h:72 (defined chain_hardware.cpp:208-235): a status other than success prints the message and ends the process.  No reference object imports it
// (only chain_hardware.cpp itself calls it); exported so that the header's four prototypes all resolve against this library.  The status is an OpenCL code in the
// reference; here any non-zero value is a failure and the library's own last error is printed beside the caller's text.
void checkError(int err, const std::string message)
{
	if (err == 0) return;                       // CL_SUCCESS
	fprintf(stderr, "%s\n", message.c_str());
	const char *last = mm2c_last_error();
	if (last && *last) fprintf(stderr, "%s\n", last);
	exit(EXIT_FAILURE);
}
