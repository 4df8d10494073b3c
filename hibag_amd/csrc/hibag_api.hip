// hibag_api.hip -- the small shared part of libhibag_hip.so's host side: the calling thread's error state and device
// selection, the kernel-target switch (hlaSetKernelTarget), and the TypeGPUExtProc-compatible plugin table.  The model is
// hibag_model.hip, the batch driver hibag_predict.hip (hibag_internal.h lists what they share).
//
// There is no CPU fallback here: every compute entry runs the HIP kernels or fails with an error code.

#include "hibag_internal.h"

namespace {

thread_local std::string g_last_error;
thread_local int g_device = 0;

} // namespace

// for the other translation units of the library
int hibag_selected_device() { return g_device; }        // the calling thread's hibag_hip_set_device() choice

int hibag_fail(int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_last_error = buf;
	return code;
}

// ===========================================================================
// C ABI: library-wide entries

extern "C" {

int hibag_hip_abi_version(void) { return HIBAG_HIP_ABI_VERSION; }

const char *hibag_hip_last_error(void) { return g_last_error.c_str(); }

int hibag_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return n;
}

int hibag_hip_set_device(int device)
{
	const int n = hibag_hip_device_count();
	if (device < 0 || device >= n)
		return hibag_fail(HIBAG_HIP_ENODEV, "HIP device %d not available (%d visible)", device, n);
	g_device = device;
	return 0;
}

int hibag_hip_get_device(void) { return g_device; }

int hibag_hip_set_kernel_target(const char *target, char *info, size_t info_len)
{
	if (!target || strcmp(target, "hip") != 0)
		return hibag_fail(HIBAG_HIP_EINVAL, "this library implements the kernel target \"hip\" only (got \"%s\")",
			target ? target : "(null)");
	const int n = hibag_hip_device_count();
	if (n <= 0 || g_device >= n) return hibag_fail(HIBAG_HIP_ENODEV, "no HIP device available");
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, g_device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return hibag_fail(HIBAG_HIP_ENODEV, "device %d is %s; the kernels are built for gfx950 only", g_device, prop.gcnArchName);
	if (info && info_len)
		snprintf(info, info_len, "HIP, %s, %s, %d CUs", prop.gcnArchName, prop.name, prop.multiProcessorCount);
	return 0;
}

} // extern "C"

// ===========================================================================
// HIBAG plugin table: layout-compatible with HLA_LIB::TypeGPUExtProc
// (inst/include/LibHLA_ext.h:358-388).  The host calls predict_init once per
// PredictHLA (src/LibHLA.cpp:2498-2523), predict_avg_prob once per sample with
// nthread = 1 (:2433-2441) and predict_done from a destructor (:2525-2531).
// A failure cannot be returned through these void signatures; like a C++
// plugin would, it throws `const char *`, which the host's CORE_CATCH turns
// into an R error (src/HIBAG.cpp:41-60).

namespace {

// The predict entries live in hibag_sample.hip (a per-sample form of the path: lane = allele-pair cell), the build entries in
// hibag_build.hip.  One model and one training state per process at a time, like the reference's single staging buffer.
const PluginTable g_plugin_table = {
	hibag_build_init, hibag_build_done, hibag_build_set_bootstrap, hibag_build_haplomatch,
	hibag_build_set_haplo_geno, hibag_build_acc_oob, hibag_build_acc_ib,
	hibag_sample_init, hibag_sample_done, hibag_sample_avg_prob,
};

} // namespace

extern "C" const void *hibag_hip_gpu_ext_proc(void) { return &g_plugin_table; }

// predict_avg_prob calls since the last predict_init whose full-width launch could not get all its workgroups resident (the
// device is shared) and that were therefore repeated on a single workgroup; 0 on a device the process has to itself.
extern "C" long long hibag_hip_plugin_degraded_calls(void) { return hibag_sample_degraded_calls(); }

// The loop an unmodified HIBAG runs around predict_avg_prob (src/LibHLA.cpp:2362-2411 with :2433-2441), in C like the host's: for
// every sample one call through the table with its packed genotypes and weights, then BestGuessEnsemble's scan of the posterior
// (:1549-1566: first strict maximum in cell order, -1 when nothing is positive).  Measurement and tests: the time of the n_samp
// calls as a compiled host sees them (no interpreter between the calls), and the per-sample best cell to check them with.
extern "C" int hibag_hip_test_time_avg_prob(const void *geno, const double *weight, int n_samp, int n_classifier, int n_cell,
	int32_t *best_cell, double *matching, double *seconds)
{
	if (!geno || !weight || n_samp < 0 || n_classifier < 0 || n_cell <= 0) return hibag_fail(HIBAG_HIP_EINVAL, "bad arguments");
	std::vector<double> prob;
	try { prob.assign((size_t)n_cell, 0.0); } catch (...) { return hibag_fail(HIBAG_HIP_ENOMEM, "out of host memory"); }
	const PluginGenotype *g = (const PluginGenotype *)geno;
	double match = 0;
	const auto t0 = std::chrono::steady_clock::now();
	try {
		for (int i = 0; i < n_samp; i++) {
			g_plugin_table.predict_avg_prob(g + (size_t)i * n_classifier, weight + (size_t)i * n_classifier, prob.data(), &match);
			double best = 0;
			int cell = -1;
			for (int p = 0; p < n_cell; p++)
				if (best < prob[p]) { best = prob[p]; cell = p; }
			if (best_cell) best_cell[i] = cell;
			if (matching) matching[i] = match;
		}
	} catch (const char *msg) {
		return hibag_fail(HIBAG_HIP_ENODEV, "%s", msg);
	}
	if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return 0;
}
