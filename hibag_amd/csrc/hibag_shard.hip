// hibag_shard.hip -- classifier-sharded prediction from ONE process over the devices of a node, merged by one RCCL
// all-reduce per batch (gfx950 / MI355X, xGMI).
//
// What is split: the ensemble sum  S[p] = sum_c (cell_c[p] * (1 / total_c)) * w_c  of CAttrBag_Model::_PredictHLA
// (src/LibHLA.cpp:2448-2476 with the accumulators of :1497-1518) over the classifiers c.  Shard r holds a contiguous
// run of the classifiers (hibag_hip_model_shard: built with the FULL model's per-SNP classifier counts, which the
// weights w_c depend on, src/LibHLA.cpp:2418-2431, :2484-2496) and writes its un-normalised partial sums
// [P + 3][n_pad] (hibag_hip_predict_partial_device); ncclAllReduce(ncclDouble, ncclSum) adds them over the devices;
// hibag_hip_finish_device turns the merged sums into the PredictHLA outputs.  The reference's own multi-worker branch,
// hlaPredict(cl = ) (R/HIBAG.R:764-808), splits the SAMPLES (that is hibag_hip_predict_multi); splitting the
// classifiers is for models too large to replicate and for small, latency-bound batches.  The order in which the
// classifiers' terms are added changes with the split, so the result is held to identical calls and 1e-10 relative on
// the posteriors against the unsharded run, not to bit equality (tests/test_hip_shard.py).
//
// One RCCL rank per DISTINCT device (ncclCommInitAll): shards that share a device are added up on it first, in shard
// order.  The whole call is driven by the calling thread -- per batch: upload + partial pass on every shard's stream,
// one ncclGroupStart / ncclAllReduce per rank / ncclGroupEnd, finish on rank 0 -- and nothing but the all-reduce
// crosses devices.  A shard whose launch failed a hand-over poisons its three scalar rows with NaN, the sum carries
// that to every rank, and the host entry runs the batch again without hand-overs (include/hibag_hip.h, "launch status").
//
// RCCL is loaded when the first group is created (dlopen of librccl.so.1, the ROCm 7 build is a 570 MB object: a host
// that never shards never maps it), through the prototypes of <rccl/rccl.h>.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <new>
#include <vector>
#include "../../include/hibag_hip.h"

int hibag_fail(int code, const char *fmt, ...);        // hibag_api.hip: records the thread's last error, returns code

namespace {

struct Rccl {
	void *lib = nullptr;
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclCommAbort) CommAbort = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	decltype(&ncclGetVersion) GetVersion = nullptr;
};

// librccl.so.1 of the process (PyTorch ships its own under the same soname: whichever is mapped already is reused)
const Rccl *rccl()
{
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
			r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (r.lib) break;
		}
		if (!r.lib) return;
		r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
		r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
		r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
		r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
		r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
		r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
		r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
		r.GetVersion = (decltype(r.GetVersion))dlsym(r.lib, "ncclGetVersion");
		if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) r.lib = nullptr;
	});
	return r.lib ? &r : nullptr;
}

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hibag_fail(HIBAG_HIP_ENODEV, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
#define NCCL_OK(expr) do { ncclResult_t e_ = (expr); if (e_ != ncclSuccess) return hibag_fail(HIBAG_HIP_ENODEV, "RCCL: %s: %s", #expr, rccl()->GetErrorString(e_)); } while (0)

// dst[i] += src[i]: the partial sums of shards that share a device, added in shard order before the all-reduce
__global__ void k_add_partial(double *__restrict__ dst, const double *__restrict__ src, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

struct Rank {                       // one per distinct device
	int device = 0;
	ncclComm_t comm = nullptr;
	hipStream_t st = nullptr;
	int32_t *d_geno = nullptr;      // the batch's genotypes [batch][n_snp]
	double *d_part = nullptr;       // merged partial sums [P + 3][batch rounded up to 64]
	double *d_tmp = nullptr;        // a further shard's sums on this device, before they are added to d_part
	std::vector<int> shards;        // indices into hibag_hip_shard_group::shards, ascending
};

} // namespace

struct hibag_hip_shard_group {
	std::vector<hibag_hip_model *> shards;
	std::vector<Rank> ranks;
	int n_hla = 0, n_snp = 0, batch = 0;
	size_t part_doubles = 0;        // (P + 3) * batch, batch a multiple of 64
	// rank 0's outputs of one batch
	int32_t *d_h1 = nullptr, *d_h2 = nullptr;
	double *d_prob = nullptr, *d_match = nullptr, *d_dos = nullptr, *d_pp = nullptr;
	int64_t allreduces = 0, retried = 0;
	bool broken = false;            // a collective failed half-way: the communicators were aborted, the group serves no further call
	std::mutex lock;

	~hibag_hip_shard_group()
	{
		const Rccl *R = rccl();
		for (Rank &r : ranks) {
			(void)hipSetDevice(r.device);
			if (r.st && !broken) (void)hipStreamSynchronize(r.st);
			if (r.comm && R) (void)R->CommDestroy(r.comm);          // (aborted communicators were set to null)
			for (void *p : {(void *)r.d_geno, (void *)r.d_part, (void *)r.d_tmp}) if (p) (void)hipFree(p);
			if (r.st) (void)hipStreamDestroy(r.st);
		}
		if (!ranks.empty()) (void)hipSetDevice(ranks[0].device);
		for (void *p : {(void *)d_h1, (void *)d_h2, (void *)d_prob, (void *)d_match, (void *)d_dos, (void *)d_pp}) if (p) (void)hipFree(p);
	}
};

namespace {

// one batch: genotypes up, partial pass per shard, merge on and across devices, finish on rank 0, outputs down
int run_batch(hibag_hip_shard_group *g, const int32_t *geno, int n, int32_t *H1, int32_t *H2, double *max_prob, double *matching,
	double *dosage, double *postprob)
{
	const Rccl *R = rccl();
	const size_t P = (size_t)g->n_hla * (g->n_hla + 1) / 2, n_pad = ((size_t)n + 63) / 64 * 64, cnt = (P + 3) * n_pad;
	// Every shard's partial pass is enqueued BEFORE any rank joins the collective: a pass that cannot be enqueued (an error
	// code from the entry) ends the batch while no stream holds an all-reduce that its peers would never join.
	for (Rank &r : g->ranks) {
		HIP_OK(hipSetDevice(r.device));
		HIP_OK(hipMemcpyAsync(r.d_geno, geno, (size_t)n * g->n_snp * sizeof(int32_t), hipMemcpyHostToDevice, r.st));
		bool first = true;
		for (int s : r.shards) {
			double *out = first ? r.d_part : r.d_tmp;
			if (int rc = hibag_hip_predict_partial_device(g->shards[s], r.d_geno, n, out, r.st)) return rc;
			if (!first) hipLaunchKernelGGL(k_add_partial, dim3(1024), dim3(256), 0, r.st, r.d_part, (const double *)r.d_tmp, cnt);
			first = false;
		}
	}
	// the posterior merge: ONE all-reduce of [P + 3][n_pad] doubles over the devices (xGMI).  A failure in here may leave
	// some ranks' streams waiting in a collective the others never joined: the communicators are aborted (which releases
	// those streams) and the group is marked unusable -- the caller must not synchronise on it and gets the error code.
	auto collective_failed = [&](const char *what, ncclResult_t e) {
		const int rc = hibag_fail(HIBAG_HIP_ENODEV, "RCCL: %s: %s (the shard group is unusable from here on)", what, R->GetErrorString(e));
		g->broken = true;
		for (Rank &r : g->ranks) {
			if (r.comm && R->CommAbort) { (void)R->CommAbort(r.comm); r.comm = nullptr; }
		}
		return rc;
	};
	ncclResult_t e = R->GroupStart();
	if (e != ncclSuccess) return collective_failed("ncclGroupStart", e);
	for (Rank &r : g->ranks) {
		e = R->AllReduce(r.d_part, r.d_part, cnt, ncclDouble, ncclSum, r.comm, r.st);
		if (e != ncclSuccess) { (void)R->GroupEnd(); return collective_failed("ncclAllReduce", e); }
	}
	e = R->GroupEnd();
	if (e != ncclSuccess) return collective_failed("ncclGroupEnd", e);
	g->allreduces++;
	Rank &r0 = g->ranks[0];
	HIP_OK(hipSetDevice(r0.device));
	if (int rc = hibag_hip_finish_device(g->shards[r0.shards[0]], r0.d_part, n, H1 ? g->d_h1 : nullptr, H1 ? g->d_h2 : nullptr,
			max_prob ? g->d_prob : nullptr, matching ? g->d_match : nullptr, dosage ? g->d_dos : nullptr, postprob ? g->d_pp : nullptr, r0.st))
		return rc;
	if (H1) {
		HIP_OK(hipMemcpyAsync(H1, g->d_h1, (size_t)n * 4, hipMemcpyDeviceToHost, r0.st));
		HIP_OK(hipMemcpyAsync(H2, g->d_h2, (size_t)n * 4, hipMemcpyDeviceToHost, r0.st));
	}
	if (max_prob) HIP_OK(hipMemcpyAsync(max_prob, g->d_prob, (size_t)n * 8, hipMemcpyDeviceToHost, r0.st));
	if (matching) HIP_OK(hipMemcpyAsync(matching, g->d_match, (size_t)n * 8, hipMemcpyDeviceToHost, r0.st));
	if (dosage) HIP_OK(hipMemcpyAsync(dosage, g->d_dos, (size_t)n * g->n_hla * 8, hipMemcpyDeviceToHost, r0.st));
	if (postprob) HIP_OK(hipMemcpyAsync(postprob, g->d_pp, (size_t)n * P * 8, hipMemcpyDeviceToHost, r0.st));
	for (Rank &r : g->ranks) {
		HIP_OK(hipSetDevice(r.device));
		HIP_OK(hipStreamSynchronize(r.st));
	}
	return 0;
}

} // namespace

extern "C" {

int hibag_hip_shard_bounds(int n_classifier, int n_shards, int shard, int *first, int *count)
{
	if (n_classifier < 0 || n_shards <= 0 || shard < 0 || shard >= n_shards)
		return hibag_fail(HIBAG_HIP_EINVAL, "bad shard query (n_classifier=%d, n_shards=%d, shard=%d)", n_classifier, n_shards, shard);
	const int base = n_classifier / n_shards, rem = n_classifier % n_shards;
	if (first) *first = shard * base + std::min(shard, rem);
	if (count) *count = base + (shard < rem ? 1 : 0);
	return 0;
}

static hibag_hip_shard_group *group_create(hibag_hip_model *const *shards, int n_shards, int *code)
{
	*code = 0;
	if (!shards || n_shards <= 0) { *code = hibag_fail(HIBAG_HIP_EINVAL, "no shards given"); return nullptr; }
	const Rccl *R = rccl();
	if (!R) { *code = hibag_fail(HIBAG_HIP_ENODEV, "librccl.so.1 could not be loaded (or lacks the collective entry points)"); return nullptr; }
	hibag_hip_shard_group *g = new (std::nothrow) hibag_hip_shard_group;
	if (!g) { *code = hibag_fail(HIBAG_HIP_ENOMEM, "out of host memory"); return nullptr; }
	auto bail = [&](int c) { *code = c; delete g; return (hibag_hip_shard_group *)nullptr; };
	try {
		g->n_hla = hibag_hip_model_n_hla(shards[0]); g->n_snp = hibag_hip_model_n_snp(shards[0]);
		g->batch = 1 << 30;
		for (int i = 0; i < n_shards; i++) {
			hibag_hip_model *m = shards[i];
			if (!m || hibag_hip_model_n_hla(m) != g->n_hla || hibag_hip_model_n_snp(m) != g->n_snp)
				return bail(hibag_fail(HIBAG_HIP_EINVAL, "shard %d is not a shard of the model of shard 0 (alleles / SNPs differ)", i));
			const int lim = hibag_hip_model_batch_limit(m);
			if (lim <= 0) return bail(hibag_fail(HIBAG_HIP_ESTATE, "shard %d is not finalized", i));
			g->batch = std::min(g->batch, lim);
			g->shards.push_back(m);
			const int dev = hibag_hip_model_device(m);
			auto it = std::find_if(g->ranks.begin(), g->ranks.end(), [&](const Rank &r) { return r.device == dev; });
			if (it == g->ranks.end()) { g->ranks.emplace_back(); g->ranks.back().device = dev; it = g->ranks.end() - 1; }
			it->shards.push_back(i);
		}
		g->batch = std::min(g->batch, 32768) / 64 * 64;      // (merged sums of 32k samples of a 50-allele locus: 335 MB per all-reduce)
		const size_t P = (size_t)g->n_hla * (g->n_hla + 1) / 2;
		g->part_doubles = (P + 3) * (size_t)g->batch;
		std::vector<int> devs;
		std::vector<ncclComm_t> comms(g->ranks.size(), nullptr);
		for (const Rank &r : g->ranks) devs.push_back(r.device);
		ncclResult_t e = R->CommInitAll(comms.data(), (int)devs.size(), devs.data());
		if (e != ncclSuccess) return bail(hibag_fail(HIBAG_HIP_ENODEV, "RCCL: ncclCommInitAll over %d device(s): %s", (int)devs.size(), R->GetErrorString(e)));
		for (size_t i = 0; i < g->ranks.size(); i++) g->ranks[i].comm = comms[i];
		for (Rank &r : g->ranks) {
			if (hipSetDevice(r.device) != hipSuccess) return bail(hibag_fail(HIBAG_HIP_ENODEV, "hipSetDevice(%d) failed", r.device));
			if (hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking) != hipSuccess ||
			    hipMalloc((void **)&r.d_geno, (size_t)g->batch * std::max(g->n_snp, 1) * sizeof(int32_t)) != hipSuccess ||
			    hipMalloc((void **)&r.d_part, g->part_doubles * sizeof(double)) != hipSuccess ||
			    (r.shards.size() > 1 && hipMalloc((void **)&r.d_tmp, g->part_doubles * sizeof(double)) != hipSuccess))
				return bail(hibag_fail(HIBAG_HIP_ENOMEM, "out of device memory on device %d", r.device));
		}
		if (hipSetDevice(g->ranks[0].device) != hipSuccess ||
		    hipMalloc((void **)&g->d_h1, (size_t)g->batch * 4) != hipSuccess || hipMalloc((void **)&g->d_h2, (size_t)g->batch * 4) != hipSuccess ||
		    hipMalloc((void **)&g->d_prob, (size_t)g->batch * 8) != hipSuccess || hipMalloc((void **)&g->d_match, (size_t)g->batch * 8) != hipSuccess ||
		    hipMalloc((void **)&g->d_dos, (size_t)g->batch * g->n_hla * 8) != hipSuccess || hipMalloc((void **)&g->d_pp, (size_t)g->batch * P * 8) != hipSuccess)
			return bail(hibag_fail(HIBAG_HIP_ENOMEM, "out of device memory on device %d", g->ranks[0].device));
	} catch (...) {
		return bail(hibag_fail(HIBAG_HIP_ENOMEM, "out of host memory"));
	}
	return g;
}

hibag_hip_shard_group *hibag_hip_shard_group_new(hibag_hip_model *const *shards, int n_shards)
{
	int code = 0;
	return group_create(shards, n_shards, &code);
}

void hibag_hip_shard_group_free(hibag_hip_shard_group *g) { delete g; }

int hibag_hip_shard_group_ranks(const hibag_hip_shard_group *g) { return g ? (int)g->ranks.size() : 0; }

int64_t hibag_hip_shard_group_allreduces(const hibag_hip_shard_group *g) { return g ? g->allreduces : 0; }

int hibag_hip_rccl_version(void)
{
	const Rccl *R = rccl();
	int v = 0;
	if (!R || !R->GetVersion || R->GetVersion(&v) != ncclSuccess) return 0;
	return v;
}

int hibag_hip_shard_group_predict(hibag_hip_shard_group *g, const int32_t *geno, int n_samp, int32_t *H1, int32_t *H2,
	double *max_prob, double *matching, double *dosage, double *postprob)
{
	if (!g) return hibag_fail(HIBAG_HIP_EINVAL, "shard group is NULL");
	if (n_samp < 0) return hibag_fail(HIBAG_HIP_EINVAL, "n_samp < 0");
	if (n_samp > 0 && !geno) return hibag_fail(HIBAG_HIP_EINVAL, "geno is NULL");
	if ((H1 == nullptr) != (H2 == nullptr)) return hibag_fail(HIBAG_HIP_EINVAL, "H1 and H2 must be given together");
	std::lock_guard<std::mutex> lk(g->lock);
	if (g->broken) return hibag_fail(HIBAG_HIP_ESTATE, "the shard group's communicators were aborted after a failed collective: create a new group");
	int dev0 = 0;
	(void)hipGetDevice(&dev0);
	const size_t P = (size_t)g->n_hla * (g->n_hla + 1) / 2;
	int rc = 0;
	for (int s0 = 0; s0 < n_samp && !rc; s0 += g->batch) {
		const int n = std::min(g->batch, n_samp - s0);
		for (int attempt = 0; attempt < 2; attempt++) {
			rc = run_batch(g, geno + (size_t)s0 * g->n_snp, n, H1 ? H1 + s0 : nullptr, H2 ? H2 + s0 : nullptr, max_prob ? max_prob + s0 : nullptr,
				matching ? matching + s0 : nullptr, dosage ? dosage + (size_t)s0 * g->n_hla : nullptr, postprob ? postprob + (size_t)s0 * P : nullptr);
			if (rc) {                                                  // nothing of a failed batch may still be running when the caller's buffers go away
				// (not after an aborted collective: the abort released the streams, and what they held is void)
				if (!g->broken) for (Rank &r : g->ranks) { (void)hipSetDevice(r.device); (void)hipStreamSynchronize(r.st); }
				break;
			}
			// a failed hand-over on any shard poisoned the merged sums of every rank: the shard now launches without
			// hand-overs (sticky status cleared here), and the batch is run once more -- never returned as numbers
			bool fault = false;
			for (hibag_hip_model *m : g->shards)
				if (hibag_hip_model_status(m) == HIBAG_HIP_EHANDOVER) { fault = true; (void)hibag_hip_model_clear_status(m); }
			if (!fault) break;
			g->retried++;
			if (attempt == 1) rc = hibag_fail(HIBAG_HIP_EHANDOVER, "a hand-over between workgroups failed in a launch without hand-overs");
		}
	}
	(void)hipSetDevice(dev0);
	return rc;
}

int hibag_hip_predict_multi_sharded(hibag_hip_model *const *shards, int n_shards, const int32_t *geno, int n_samp,
	int32_t *H1, int32_t *H2, double *max_prob, double *matching, double *dosage, double *postprob)
{
	int code = 0;
	hibag_hip_shard_group *g = group_create(shards, n_shards, &code);
	if (!g) return code ? code : HIBAG_HIP_EINVAL;
	const int rc = hibag_hip_shard_group_predict(g, geno, n_samp, H1, H2, max_prob, matching, dosage, postprob);
	hibag_hip_shard_group_free(g);
	return rc;
}

} // extern "C"
