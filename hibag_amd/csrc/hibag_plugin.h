// hibag_plugin.h -- layout mirrors of the types HIBAG shares with a GPU plugin
// (inst/include/LibHLA_ext.h) and the entry points behind the plugin table.
#ifndef HIBAG_PLUGIN_H_
#define HIBAG_PLUGIN_H_

#include <stddef.h>
#include <stdint.h>

struct PluginHaplotype {            // THaplotype, LibHLA_ext.h:261-299 (32 bytes)
	int64_t packed[2];
	double freq;
	struct { float freq_f32; int hla_allele; } aux;   // filled by SetHaploAux_GPU, src/LibHLA.cpp:565-578
};
struct PluginGenotype {             // TGenotype, LibHLA_ext.h:311-352 (48 bytes)
	int64_t snp1[2], snp2[2];
	int bootstrap_count, hla1, hla2, pad;
};
static_assert(sizeof(PluginHaplotype) == 32, "THaplotype must be 32 bytes");
static_assert(sizeof(PluginGenotype) == 48, "TGenotype must be 48 bytes");

struct PluginTable {                // TypeGPUExtProc, LibHLA_ext.h:358-388
	void (*build_init)(int, int);
	void (*build_done)();
	void (*build_set_bootstrap)(const int[]);
	uint32_t *(*build_haplomatch)(const PluginHaplotype[], const size_t[], int, const PluginGenotype[], size_t &);
	void (*build_set_haplo_geno)(const PluginHaplotype[], int, const PluginGenotype[], int);
	int (*build_acc_oob)();
	double (*build_acc_ib)();
	void (*predict_init)(int, int, const PluginHaplotype *const[], const int[], const int[]);
	void (*predict_done)();
	void (*predict_avg_prob)(const PluginGenotype[], const double[], double[], double[]);
};

// per-sample prediction entries (hibag_sample.hip): predict_init / predict_avg_prob / predict_done
void hibag_sample_init(int n_hla, int n_classifier, const PluginHaplotype *const p_haplo[], const int n_haplo[], const int n_snp[]);
void hibag_sample_avg_prob(const PluginGenotype geno[], const double weight[], double out_prob[], double out_match[]);
void hibag_sample_done();
long long hibag_sample_degraded_calls();   // calls of the current predict_init that had to be repeated on one workgroup (shared device)

// training-side entries (hibag_build.hip); failures throw `const char *` like the
// predict entries (the host's CORE_CATCH turns that into an R error, src/HIBAG.cpp:41-60)
void hibag_build_init(int n_hla, int n_sample);
void hibag_build_done();
void hibag_build_set_bootstrap(const int oob_cnt[]);
uint32_t *hibag_build_haplomatch(const PluginHaplotype haplo[], const size_t n_haplo[], int n_snp,
	const PluginGenotype geno[], size_t &out_n);
void hibag_build_set_haplo_geno(const PluginHaplotype haplo[], int n_haplo, const PluginGenotype geno[], int n_snp);
int hibag_build_acc_oob();
double hibag_build_acc_ib();


// batched scoring of the candidate SNPs of one growth step, for the library's own driver
// (hibag_train.hip); same results as set_haplo_geno + acc_oob + acc_ib per candidate
struct HibagBuildCandidate {
	const PluginHaplotype *haplo;   // aux.hla_allele filled (SetHaploAux_GPU)
	int n_haplo;
	const int32_t *column;          // [n_sample] raw genotype of the candidate SNP
	int snp = -1;                   // its row in the matrix given to hibag_build_set_genotypes (-1: none was given; `column` is used)
};
// the cohort's genotypes, SNP-major int32 [n_snp][n_sample], kept on the device for the training call (after hibag_build_init)
void hibag_build_set_genotypes(const int32_t *geno_snp_major, int n_snp);
// acc_floor: the search's best out-of-bag count so far.  The reference looks at a candidate's in-bag loss only when its
// out-of-bag count reaches the running maximum of the comparison (src/LibHLA.cpp:2033-2034); the same rule decides here
// which candidates' losses are computed at all (the others get 0 -- they are never read).
void hibag_build_eval_batch(const PluginGenotype base_geno[], int n_snp, const HibagBuildCandidate cand[], int n_cand,
	int acc_floor, int acc_oob[], double loss_ib[]);

void hibag_build_eval_launch(int slot, const PluginGenotype base_geno[], int n_snp, const HibagBuildCandidate cand[], int n_cand);
void hibag_build_eval_collect(int slot, int *acc_floor, int acc_oob[], double loss_ib[]);

#endif
