/*
 * hibag_oracle_train.c -- CPU restatement of HIBAG's training driver: bootstrap,
 * greedy SNP selection with an EM haplotype fit per candidate, out-of-bag /
 * in-bag scoring (hlaAttrBagging -> HIBAG_NewClassifiers ->
 * CAttrBag_Model::BuildClassifiers, src/LibHLA.cpp:2268-2305).
 * TEST INFRASTRUCTURE ONLY (see hibag_oracle.c): used by tests/ as the checker
 * of the product's GPU-scored training driver.
 *
 * PARITY PIN.  Training consumes R's random stream (unif_rand).  R is not part of
 * /root/reference; its default generator is restated here from its published
 * algorithm (R >= 1.7: Mersenne-Twister MT19937, set.seed() scrambling of
 * src/main/RNG.c: 50 + 625 steps of seed = 69069*seed + 1, mti forced to 624,
 * output y * 2.3283064365386963e-10 clamped into (0,1)).  The pin is the
 * reference's own fixture inst/extdata/OutOfBag.RData, which the vignette builds
 * with `set.seed(100); hlaAttrBagging(hlatab$training, train.geno, nclassifier=100)`
 * (vignettes/HIBAG.Rmd:218-220): tests/test_oracle_train.py re-runs that call with
 * this file and compares bootstrap counts, selected SNPs, haplotypes, frequencies
 * and out-of-bag accuracies of the stored classifiers.
 *
 * Every function cites the reference lines it follows; the scoring kernels are
 * the ones of hibag_oracle.c.
 */

#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NA_INT (-2147483647 - 1)
#define MAX_SNP 128                           /* inst/include/LibHLA_ext.h:223 */

/* src/LibHLA.cpp:98-116 */
static const int EM_MAX_ITER = 500;
static const double EM_INIT_VAL_FRAC = 0.001;
#define EM_FUNC_RELTOL sqrt(DBL_EPSILON)
static const double MIN_RARE_FREQ = 1e-5;     /* LibHLA_ext.h:230 */
static const double FRACTION_HAPLO = 1.0 / 10;
static const double STOP_RELTOL_LOGLIK_ADDSNP = 0.001;
static const double PRUNE_RELTOL_LOGLIK = 0.1;

/* scoring kernels (hibag_oracle.c) */
void oracle_best_guess(int n_hla, int n_snp, const int *len_per_hla, const uint64_t *bits, const double *freq,
	const uint64_t s1[2], const uint64_t s2[2], int out_hla[2]);
double oracle_post_prob(int n_hla, int n_snp, const int *len_per_hla, const uint64_t *bits, const double *freq,
	const uint64_t s1[2], const uint64_t s2[2], int a1, int a2);
int oracle_prep_haplo_match(int n_snp, const uint64_t *bits, int st1, int n1, int st2, int n2,
	const uint64_t s1[2], const uint64_t s2[2], int *out_pairs);
int oracle_compare_hla(int p1, int p2, int t1, int t2);

/* ---- R's default RNG ----------------------------------------------------- */

typedef struct { uint32_t mt[624]; int mti; } RRng;

void oracle_rng_set_seed(RRng *r, uint32_t seed)        /* set.seed(seed): RNG.c Randomize() + FixupSeeds() */
{
	for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
	for (int j = 0; j < 625; j++) {
		seed = 69069u * seed + 1u;
		if (j > 0) r->mt[j - 1] = seed;                  /* word 0 is the position, overwritten below */
	}
	r->mti = 624;
}

double oracle_rng_unif(RRng *r)                         /* MT_genrand() + fixup() */
{
	enum { N = 624, M = 397 };
	uint32_t *mt = r->mt, y;
	if (r->mti >= N) {
		int kk;
		for (kk = 0; kk < N - M; kk++) {
			y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
			mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
		}
		for (; kk < N - 1; kk++) {
			y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
			mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
		}
		y = (mt[N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
		mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
		r->mti = 0;
	}
	y = mt[r->mti++];
	y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
	const double v = (double)y * 2.3283064365386963e-10;
	const double i2_32m1 = 2.328306437080797e-10;
	if (v <= 0.0) return 0.5 * i2_32m1;
	if (1.0 - v <= 0.0) return 1.0 - 0.5 * i2_32m1;
	return v;
}

static int random_num(RRng *r, int n)                   /* src/LibHLA.cpp:120-126 */
{
	int v = (int)(n * oracle_rng_unif(r));
	if (v >= n) v = n - 1;
	return v;
}

/* ---- containers ---------------------------------------------------------- */

typedef struct { uint64_t bits[2]; double freq, old_freq; } Hap;       /* THaplotype */
typedef struct { Hap *list; size_t n, cap; int n_snp; int *len; } HapList;  /* CHaplotypeList; len[n_hla] */
typedef struct { uint64_t s1[2], s2[2]; int boot, a1, a2; } Geno;      /* TGenotype */
typedef struct { int flag, h1, h2; double gfreq; } Pair;               /* CAlg_EM::THaploPair, indices into NextHaplo */
typedef struct { int boot, samp; Pair *p; int n; } PairList;           /* CAlg_EM::THaploPairList */

typedef struct {
	int n_snp, n_samp, n_hla;
	const int *geno;          /* [n_samp][n_snp] */
	const int *h1, *h2;
} TrainData;

typedef struct {
	int *snpidx; int n_snpidx;
	int *samp_num;
	Hap *hap; int *hap_hla; int n_hap;
	double acc;
} OutClassifier;

typedef struct {
	TrainData d;
	RRng rng;
	Geno *g; int g_nsnp;
	int *inbag, n_in, *outbag, n_out;
	PairList *pl; int n_pl;
	double *log_buf;
	OutClassifier *out; int n_outc;
} Trainer;

static void hl_reserve(HapList *h, size_t n)
{
	if (n > h->cap) { h->cap = n * 2 + 16; h->list = (Hap *)realloc(h->list, sizeof(Hap) * h->cap); }
}
static void hl_init(HapList *h, int n_hla) { memset(h, 0, sizeof(*h)); h->len = (int *)calloc((size_t)n_hla, sizeof(int)); }
static void hl_free(HapList *h) { free(h->list); free(h->len); }
static void hl_copy(HapList *dst, const HapList *src, int n_hla)        /* CHaplotypeList::operator=, :397-404 */
{
	hl_reserve(dst, src->n);
	dst->n = src->n; dst->n_snp = src->n_snp;
	memcpy(dst->len, src->len, sizeof(int) * (size_t)n_hla);
	memcpy(dst->list, src->list, sizeof(Hap) * src->n);
}

static void set_allele(Hap *h, int idx, int v)                          /* THaplotype::_SetAllele, :339-345 */
{
	const uint64_t bit = (uint64_t)1 << (idx & 63);
	if (v) h->bits[idx >> 6] |= bit; else h->bits[idx >> 6] &= ~bit;
}
static int get_allele(const Hap *h, int idx) { return (int)((h->bits[idx >> 6] >> (idx & 63)) & 1); }

static void geno_set(Geno *g, int idx, int val)                         /* TGenotype::_SetSNP, :609-622 */
{
	const uint64_t bit = (uint64_t)1 << (idx & 63);
	const int w = idx >> 6;
	int b1, b2;
	switch (val) {
	case 0: b1 = 0; b2 = 0; break;
	case 1: b1 = 1; b2 = 0; break;
	case 2: b1 = 1; b2 = 1; break;
	default: b1 = 0; b2 = 1;
	}
	if (b1) g->s1[w] |= bit; else g->s1[w] &= ~bit;
	if (b2) g->s2[w] |= bit; else g->s2[w] &= ~bit;
}

/* CHaplotypeList::DoubleHaplos, :416-442 */
static void double_haplos(const HapList *cur, HapList *out, int n_hla)
{
	hl_reserve(out, cur->n * 2);
	out->n = cur->n * 2; out->n_snp = cur->n_snp + 1;
	for (size_t i = 0; i < cur->n; i++) {
		out->list[2 * i] = cur->list[i]; set_allele(&out->list[2 * i], cur->n_snp, 0);
		out->list[2 * i + 1] = cur->list[i]; set_allele(&out->list[2 * i + 1], cur->n_snp, 1);
	}
	for (int h = 0; h < n_hla; h++) out->len[h] = cur->len[h] * 2;
}

/* CHaplotypeList::DoubleHaplosInitFreq, :444-459 */
static void double_haplos_init_freq(const HapList *cur, HapList *out, double afreq)
{
	const double p0 = 1 - afreq, p1 = afreq;
	for (size_t i = 0; i < cur->n; i++) {
		out->list[2 * i].freq = p0 * cur->list[i].freq + EM_INIT_VAL_FRAC;
		out->list[2 * i + 1].freq = p1 * cur->list[i].freq + EM_INIT_VAL_FRAC;
	}
}

/* CHaplotypeList::EraseDoubleHaplos, :461-515 */
static void erase_double_haplos(const HapList *in, double rare_prob, HapList *out, int n_hla)
{
	hl_reserve(out, in->n);
	out->n_snp = in->n_snp;
	const Hap *p = in->list;
	Hap *po = out->list;
	double sum = 0;
	for (int h = 0; h < n_hla; h++) {
		int num = 0;
		for (int n = in->len[h]; n > 0; n -= 2, p += 2) {
			const double sumfreq = p[0].freq + p[1].freq;
			if (p[0].freq < rare_prob || p[1].freq < rare_prob) {
				if (sumfreq >= MIN_RARE_FREQ) {
					*po = (p[0].freq >= p[1].freq) ? p[0] : p[1];
					po->freq = sumfreq; po++;
					sum += sumfreq;
					num++;
				}
			} else {
				*po++ = p[0]; *po++ = p[1];
				sum += sumfreq;
				num += 2;
			}
		}
		out->len[h] = num;
	}
	out->n = (size_t)(po - out->list);
	const double scale = 1 / sum;                                        /* ScaleFrequency, :527-532 */
	for (size_t i = 0; i < out->n; i++) out->list[i].freq *= scale;
}

/* ---- CSamplingWithoutReplace, :930-993 ----------------------------------- */

typedef struct { int *a; int n, m_try; } Sampling;

static void smp_init(Sampling *s, int m_total)
{
	s->a = (int *)realloc(s->a, sizeof(int) * (size_t)(m_total > 0 ? m_total : 1));
	s->n = m_total; s->m_try = 0;
	for (int i = 0; i < m_total; i++) s->a[i] = i;
}
static void smp_select(Sampling *s, RRng *r, int m_try)
{
	const int n = s->n;
	if (m_try > n) m_try = n;
	if (m_try < n)
		for (int i = 0; i < m_try; i++) {
			const int I = random_num(r, n - i);
			const int t = s->a[I]; s->a[I] = s->a[n - i - 1]; s->a[n - i - 1] = t;
		}
	s->m_try = m_try;
}
static int *smp_at(Sampling *s, int idx) { return &s->a[s->n - s->m_try + idx]; }
static void smp_erase(Sampling *s, int pos) { memmove(s->a + pos, s->a + pos + 1, sizeof(int) * (size_t)(s->n - pos - 1)); s->n--; }
static void smp_remove(Sampling *s, int idx) { smp_erase(s, s->n - s->m_try + idx); }
static void smp_remove_selection(Sampling *s) { s->n -= s->m_try; }
static void smp_remove_flag(Sampling *s)
{
	const int n = s->n;
	for (int i = n - 1; i >= n - s->m_try; i--)
		if (s->a[i] < 0) smp_erase(s, i);
}

/* ---- CVariableSelection / CAlg_EM ---------------------------------------- */

/* CVariableSelection::InitSelection, :1843-1878 */
static void init_selection(Trainer *t, const int *boot)
{
	const int n = t->d.n_samp;
	t->n_in = t->n_out = 0;
	for (int i = 0; i < n; i++) {
		Geno *g = &t->g[i];
		g->boot = boot[i];
		g->a1 = t->d.h1[i]; g->a2 = t->d.h2[i];
		if (g->a2 < g->a1) { const int w = g->a2; g->a2 = g->a1; g->a1 = w; }
		if (boot[i] > 0) t->inbag[t->n_in++] = i; else t->outbag[t->n_out++] = i;
		g->s1[0] = g->s1[1] = 0;                                         /* SetAllMissing, :883-891 */
		g->s2[0] = g->s2[1] = ~(uint64_t)0;
	}
	t->g_nsnp = 0;
}

/* CVariableSelection::_InitHaplotype, :1880-1911 (unused bits are zero here; the
 * reference leaves them uninitialised, they are masked by the genotypes) */
static void init_haplotype(Trainer *t, HapList *h)
{
	const int nh = t->d.n_hla;
	int *tmp = (int *)calloc((size_t)nh, sizeof(int));
	int sum = 0;
	for (int k = 0; k < t->n_in; k++) {
		const Geno *g = &t->g[t->inbag[k]];
		tmp[g->a1] += g->boot; tmp[g->a2] += g->boot;
		sum += g->boot;
	}
	hl_reserve(h, (size_t)nh);
	h->n_snp = 0; h->n = 0;
	const double scale = 0.5 / sum;
	for (int i = 0; i < nh; i++) {
		h->len[i] = tmp[i] > 0 ? 1 : 0;
		if (tmp[i] > 0) {
			Hap *p = &h->list[h->n++];
			memset(p, 0, sizeof(*p));
			p->freq = tmp[i] * scale;
		}
	}
	free(tmp);
}

static void add_snp(Trainer *t, int snp)                                /* CGenotypeList::AddSNP, :860-874 */
{
	for (int i = 0; i < t->d.n_samp; i++)
		geno_set(&t->g[i], t->g_nsnp, t->d.geno[(size_t)i * t->d.n_snp + snp]);
	t->g_nsnp++;
}
static void set_missing(Trainer *t, int idx)                            /* CGenotypeList::SetMissing, :893-903 */
{
	for (int i = 0; i < t->d.n_samp; i++) geno_set(&t->g[i], idx, -1);
}

/* CAlg_EM::PrepareHaplotypes, CPU branch, :1002-1125 */
static void prepare_haplotypes(Trainer *t, const HapList *cur, HapList *next)
{
	const int nh = t->d.n_hla;
	double_haplos(cur, next, nh);
	int *start = (int *)malloc(sizeof(int) * (size_t)nh);
	for (int i = 0, st = 0; i < nh; i++) { start[i] = st; st += next->len[i]; }
	uint64_t *bits = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (next->n ? next->n : 1));
	for (size_t i = 0; i < next->n; i++) { bits[2 * i] = next->list[i].bits[0]; bits[2 * i + 1] = next->list[i].bits[1]; }
	for (int i = 0; i < t->n_pl; i++) { free(t->pl[i].p); t->pl[i].p = NULL; }
	t->n_pl = t->n_in;
	const int n_snp = next->n_snp - 1;
	for (int i = 0; i < t->n_in; i++) {
		const int k = t->inbag[i];
		const Geno *g = &t->g[k];
		PairList *pl = &t->pl[i];
		pl->boot = g->boot; pl->samp = k;
		const int n = oracle_prep_haplo_match(n_snp, bits, start[g->a1], next->len[g->a1], start[g->a2], next->len[g->a2],
			g->s1, g->s2, NULL);
		int *idx = (int *)malloc(sizeof(int) * 2 * (size_t)(n ? n : 1));
		oracle_prep_haplo_match(n_snp, bits, start[g->a1], next->len[g->a1], start[g->a2], next->len[g->a2], g->s1, g->s2, idx);
		pl->p = (Pair *)malloc(sizeof(Pair) * (size_t)(n ? n : 1));
		pl->n = n;
		for (int j = 0; j < n; j++) { pl->p[j].flag = 0; pl->p[j].h1 = idx[2 * j]; pl->p[j].h2 = idx[2 * j + 1]; pl->p[j].gfreq = 0; }
		free(idx);
	}
	free(bits); free(start);
}

/* CAlg_EM::PrepareNewSNP, :1127-1183 */
static int prepare_new_snp(Trainer *t, int snp, const HapList *cur, HapList *next)
{
	int allele_cnt = 0, valid_cnt = 0;
	for (int k = 0; k < t->n_in; k++) {
		const int i = t->inbag[k];
		const int dup = t->g[i].boot;
		const int g = t->d.geno[(size_t)i * t->d.n_snp + snp];
		if (0 <= g && g <= 2) { allele_cnt += g * dup; valid_cnt += 2 * dup; }
	}
	if (allele_cnt == 0 || allele_cnt == valid_cnt) return 0;
	double_haplos_init_freq(cur, next, (double)allele_cnt / valid_cnt);
	const int idx_new = next->n_snp - 1;
	for (int i = 0; i < t->n_pl; i++) {
		PairList *pl = &t->pl[i];
		const int geno = t->d.geno[(size_t)pl->samp * t->d.n_snp + snp];
		for (int j = 0; j < pl->n; j++) {
			Pair *p = &pl->p[j];
			p->flag = (0 <= geno && geno <= 2) ?
				(get_allele(&next->list[p->h1], idx_new) + get_allele(&next->list[p->h2], idx_new) == geno) : 1;
		}
	}
	return 1;
}

/* CAlg_EM::ExpectationMaximization, :1185-1255 */
static void expectation_maximization(Trainer *t, HapList *next)
{
	const int total = t->d.n_samp;
	double conv_tol = 0, loglik = -1e+30;
	for (int iter = 0; iter <= EM_MAX_ITER; iter++) {
		const double old_loglik = loglik;
		for (size_t i = 0; i < next->n; i++) { next->list[i].old_freq = next->list[i].freq; next->list[i].freq = 0; }
		for (int i = 0; i < t->n_pl; i++) {
			PairList *pl = &t->pl[i];
			double psum = 0;
			for (int j = 0; j < pl->n; j++) {
				Pair *p = &pl->p[j];
				if (p->flag) {
					p->gfreq = (p->h1 != p->h2) ?
						(2 * next->list[p->h1].old_freq * next->list[p->h2].old_freq) :
						(next->list[p->h1].old_freq * next->list[p->h2].old_freq);
					psum += p->gfreq;
				}
			}
			t->log_buf[i] = pl->boot * log(psum);
			psum = pl->boot / psum;
			for (int j = 0; j < pl->n; j++) if (pl->p[j].flag) pl->p[j].gfreq *= psum;
		}
		loglik = 0;
		for (int i = 0; i < t->n_pl; i++) {
			loglik += t->log_buf[i];
			const PairList *pl = &t->pl[i];
			for (int j = 0; j < pl->n; j++)
				if (pl->p[j].flag) {
					const double r = pl->p[j].gfreq;
					next->list[pl->p[j].h1].freq += r; next->list[pl->p[j].h2].freq += r;
				}
		}
		const double scale = 0.5 / total;
		for (size_t i = 0; i < next->n; i++) next->list[i].freq *= scale;
		if (iter > 0) {
			if (fabs(loglik - old_loglik) <= conv_tol) break;
		} else {
			conv_tol = EM_FUNC_RELTOL * (fabs(loglik) + EM_FUNC_RELTOL);
			if (conv_tol < 0) conv_tol = 0;
		}
	}
}

typedef struct { int *len; uint64_t *bits; double *freq; } Soa;

static void to_soa(const HapList *h, Soa *s, int n_hla)
{
	s->len = h->len; (void)n_hla;
	s->bits = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (h->n ? h->n : 1));
	s->freq = (double *)malloc(sizeof(double) * (h->n ? h->n : 1));
	for (size_t i = 0; i < h->n; i++) {
		s->bits[2 * i] = h->list[i].bits[0]; s->bits[2 * i + 1] = h->list[i].bits[1];
		s->freq[i] = h->list[i].freq;
	}
}

/* CVariableSelection::_OutOfBagAccuracy, :1934-1955 (CPU branch) */
static int out_of_bag_accuracy(Trainer *t, const HapList *h, const Soa *s)
{
	int correct = 0;
	for (int i = 0; i < t->n_out; i++) {
		const Geno *g = &t->g[t->outbag[i]];
		int guess[2];
		oracle_best_guess(t->d.n_hla, h->n_snp, s->len, s->bits, s->freq, g->s1, g->s2, guess);
		correct += oracle_compare_hla(guess[0], guess[1], g->a1, g->a2);
	}
	return correct;
}

/* CVariableSelection::_InBagLogLik, :1957-1979 (CPU branch) */
static double in_bag_loglik(Trainer *t, const HapList *h, const Soa *s)
{
	double loglik = 0;
	for (int i = 0; i < t->n_in; i++) {
		const Geno *g = &t->g[t->inbag[i]];
		loglik += g->boot * log(oracle_post_prob(t->d.n_hla, h->n_snp, s->len, s->bits, s->freq, g->s1, g->s2, g->a1, g->a2));
	}
	return loglik * -2;
}

/* CVariableSelection::Search, :1981-2122 */
static void search(Trainer *t, Sampling *vs, HapList *out_haplo, int *out_snp, int *n_out_snp, double *out_acc,
	int mtry, int prune)
{
	const int nh = t->d.n_hla;
	const double rare_prob = fmax(FRACTION_HAPLO / (2 * t->d.n_samp), MIN_RARE_FREQ);
	init_haplotype(t, out_haplo);
	*n_out_snp = 0;
	const int num_oob = t->n_out;
	int global_max_acc = 0;
	double global_min_loss = 1e+30;
	HapList next, reduced, minh;
	hl_init(&next, nh); hl_init(&reduced, nh); hl_init(&minh, nh);

	while (vs->n > 0 && *n_out_snp < MAX_SNP) {
		prepare_haplotypes(t, out_haplo, &next);
		int max_acc = global_max_acc;
		double min_loss = global_min_loss;
		int min_i = -1;
		smp_select(vs, &t->rng, mtry);
		for (int i = 0; i < vs->m_try; i++) {
			if (prepare_new_snp(t, *smp_at(vs, i), out_haplo, &next)) {
				expectation_maximization(t, &next);
				erase_double_haplos(&next, rare_prob, &reduced, nh);
				add_snp(t, *smp_at(vs, i));
				Soa s;
				to_soa(&reduced, &s, nh);
				double loss = 0;
				const int acc = out_of_bag_accuracy(t, &reduced, &s);
				if (acc >= max_acc) loss = in_bag_loglik(t, &reduced, &s);
				free(s.bits); free(s.freq);
				t->g_nsnp--;                                             /* ReduceSNP, :876-881 */
				if (acc > max_acc) {
					min_i = i; min_loss = loss; max_acc = acc;
					hl_copy(&minh, &reduced, nh);
				} else if (acc == max_acc) {
					if (loss < min_loss) { min_i = i; min_loss = loss; hl_copy(&minh, &reduced, nh); }
				}
				if (prune) {
					if (acc < global_max_acc) *smp_at(vs, i) = -1;
					else if (acc == global_max_acc) {
						if (loss > global_min_loss * (1 + PRUNE_RELTOL_LOGLIK) && min_i != i) *smp_at(vs, i) = -1;
					}
				}
			}
		}
		int sign = 0;
		if (max_acc > global_max_acc) sign = 1;
		else if (max_acc == global_max_acc) {
			if (min_i >= 0)
				sign = (min_loss >= STOP_RELTOL_LOGLIK_ADDSNP) && (min_loss < global_min_loss * (1 - STOP_RELTOL_LOGLIK_ADDSNP));
		}
		if (sign) {
			global_max_acc = max_acc;
			global_min_loss = min_loss;
			hl_copy(out_haplo, &minh, nh);
			out_snp[(*n_out_snp)++] = *smp_at(vs, min_i);
			add_snp(t, *smp_at(vs, min_i));
			if (prune) { *smp_at(vs, min_i) = -1; smp_remove_flag(vs); }
			else smp_remove(vs, min_i);
		} else {
			smp_remove_selection(vs);
			set_missing(t, t->g_nsnp);
		}
	}
	*out_acc = 0.5 * global_max_acc / num_oob;
	hl_free(&next); hl_free(&reduced); hl_free(&minh);
}

/* ---- public: CAttrBag_Model::BuildClassifiers, :2268-2305 ----------------- */

void *oracle_train_new(int n_snp, int n_samp, const int *geno, int n_hla, const int *h1, const int *h2, unsigned seed)
{
	Trainer *t = (Trainer *)calloc(1, sizeof(Trainer));
	t->d.n_snp = n_snp; t->d.n_samp = n_samp; t->d.n_hla = n_hla;
	t->d.geno = geno; t->d.h1 = h1; t->d.h2 = h2;                        /* caller keeps the arrays alive */
	oracle_rng_set_seed(&t->rng, seed);
	t->g = (Geno *)calloc((size_t)n_samp, sizeof(Geno));
	t->inbag = (int *)malloc(sizeof(int) * (size_t)n_samp);
	t->outbag = (int *)malloc(sizeof(int) * (size_t)n_samp);
	t->pl = (PairList *)calloc((size_t)n_samp, sizeof(PairList));
	t->log_buf = (double *)malloc(sizeof(double) * (size_t)n_samp);
	return t;
}

int oracle_train_run(void *handle, int nclassifier, int mtry, int prune)
{
	Trainer *t = (Trainer *)handle;
	const int n = t->d.n_samp;
	Sampling vs = {0};
	for (int k = 0; k < nclassifier; k++) {
		smp_init(&vs, t->d.n_snp);
		/* NewClassifierBootstrap, :2220-2245 */
		int *S = (int *)malloc(sizeof(int) * (size_t)n);
		int n_unique;
		do {
			memset(S, 0, sizeof(int) * (size_t)n);
			n_unique = 0;
			for (int i = 0; i < n; i++) {
				const int j = random_num(&t->rng, n);
				if (S[j] == 0) n_unique++;
				S[j]++;
			}
		} while (n_unique >= n);
		/* Grow, :2167-2174 */
		init_selection(t, S);
		HapList h;
		hl_init(&h, t->d.n_hla);
		t->out = (OutClassifier *)realloc(t->out, sizeof(OutClassifier) * (size_t)(t->n_outc + 1));
		OutClassifier *o = &t->out[t->n_outc++];
		o->snpidx = (int *)malloc(sizeof(int) * MAX_SNP);
		o->samp_num = S;
		search(t, &vs, &h, o->snpidx, &o->n_snpidx, &o->acc, mtry, prune);
		o->n_hap = (int)h.n;
		o->hap = (Hap *)malloc(sizeof(Hap) * (h.n ? h.n : 1));
		o->hap_hla = (int *)malloc(sizeof(int) * (h.n ? h.n : 1));
		memcpy(o->hap, h.list, sizeof(Hap) * h.n);
		for (int a = 0, i = 0; a < t->d.n_hla; a++)
			for (int m = 0; m < h.len[a]; m++) o->hap_hla[i++] = a;
		hl_free(&h);
	}
	free(vs.a);
	return t->n_outc;
}

int oracle_train_info(void *handle, int idx, int *n_snp, int *n_haplo, double *acc)
{
	Trainer *t = (Trainer *)handle;
	if (idx < 0 || idx >= t->n_outc) return -1;
	*n_snp = t->out[idx].n_snpidx; *n_haplo = t->out[idx].n_hap; *acc = t->out[idx].acc;
	return 0;
}

int oracle_train_get(void *handle, int idx, int *snpidx, int *samp_num, double *freq, int *hla, uint64_t *bits)
{
	Trainer *t = (Trainer *)handle;
	if (idx < 0 || idx >= t->n_outc) return -1;
	const OutClassifier *o = &t->out[idx];
	memcpy(snpidx, o->snpidx, sizeof(int) * (size_t)o->n_snpidx);
	memcpy(samp_num, o->samp_num, sizeof(int) * (size_t)t->d.n_samp);
	for (int i = 0; i < o->n_hap; i++) {
		freq[i] = o->hap[i].freq; hla[i] = o->hap_hla[i];
		bits[2 * i] = o->hap[i].bits[0]; bits[2 * i + 1] = o->hap[i].bits[1];
	}
	return 0;
}

void oracle_train_free(void *handle)
{
	Trainer *t = (Trainer *)handle;
	if (!t) return;
	for (int i = 0; i < t->n_outc; i++) { free(t->out[i].snpidx); free(t->out[i].samp_num); free(t->out[i].hap); free(t->out[i].hap_hla); }
	for (int i = 0; i < t->d.n_samp; i++) free(t->pl[i].p);
	free(t->out); free(t->g); free(t->inbag); free(t->outbag); free(t->pl); free(t->log_buf);
	free(t);
}
