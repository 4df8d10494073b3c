// hibag_ablation.h -- the timing ablations and diagnostics of the hot kernels, kept out of their loops.
//
// A build with one of the HIBAG_ABL* switches produces WRONG results or slower code on purpose: it exists to attribute time
// (tools/build_variant.sh NAME -DHIBAG_ABL...; tools/run_var2.sh times the variants of one gpurun call side by side).  In
// the shipped library every constant below is false and every helper is the plain operation, so the loops in
// hibag_kernels.hip read -- and compile -- without them.
//
//   HIBAG_ABL_NOMFMA        the matrix instructions replaced by one logic operation (their share of a block)
//   HIBAG_ABL_NOSWAP        no lane swaps behind them
//   HIBAG_ABL_NOTAB         no table look-up in LDS: the value is made from the distance's bits
//   HIBAG_ABL_NOFAC         every group of records multiplies by the block's first factors (no further scalar loads)
//   HIBAG_ABL_NOEND         no cell ever closes; HIBAG_ABL_NOSTORE: cells close, none is stored
//   HIBAG_ABL_STX4 / _STHALF   pass 1's stores 16 bytes per lane instead of 8 (overlapping: the same lines) / only every second stored cell written
//   HIBAG_ABL_WIDE_NOSTORE  k_total_wide stores no cell sums
//   HIBAG_ABL2_NOLOOP       pass 2 without its block loop; _NOEVAL: without the pairs' evaluation; _NOSV: without stored sums;
//   HIBAG_ABL2_SVHOT        stored sums read from eight cache-hot rows instead of HBM; _NOWINV: weight and 1/total constants
//   HIBAG_ABL2_SVNOLOAD / _SVNOADD   pass 2's stored sums added but never loaded / loaded but never added
//   HIBAG_ABL2_SVX4 / _SVHALF   the stored sums' loads 16 bytes per lane instead of 8 (overlapping: the same lines) / only every second one issued
//   HIBAG_ABL2_NOADD        pass 2's closing cells make their product but do not add it to the LDS sums
//   HIBAG_ABL2_LDSPAD=bytes pass 2 with that much more LDS per workgroup: fewer resident wavefronts, same code
//   HIBAG_ACCUM_STAMPS      (diagnostic, right results) pass 2 reads the clock at the phase boundaries of every block and sums
//                           the differences per launch (hibag_hip_test_read_diag, tools/accum_stamps.py)
#ifndef HIBAG_ABLATION_H_
#define HIBAG_ABLATION_H_

#ifdef HIBAG_ABL_NOMFMA
constexpr bool ABL_NOMFMA = true;
#else
constexpr bool ABL_NOMFMA = false;
#endif
#ifdef HIBAG_ABL_NOSWAP
constexpr bool ABL_NOSWAP = true;
#else
constexpr bool ABL_NOSWAP = false;
#endif
#ifdef HIBAG_ABL_NOFAC
constexpr bool ABL_NOFAC = true;
#else
constexpr bool ABL_NOFAC = false;
#endif
#ifdef HIBAG_ABL_WIDE_NOSTORE
constexpr bool ABL_WIDE_NOSTORE = true;
#else
constexpr bool ABL_WIDE_NOSTORE = false;
#endif
#ifdef HIBAG_ABL2_NOLOOP
constexpr bool ABL2_NOLOOP = true;
#else
constexpr bool ABL2_NOLOOP = false;
#endif
#ifdef HIBAG_ABL2_NOEVAL
constexpr bool ABL2_NOEVAL = true;
#else
constexpr bool ABL2_NOEVAL = false;
#endif
#ifdef HIBAG_ABL2_NOWINV
constexpr bool ABL2_NOWINV = true;
#else
constexpr bool ABL2_NOWINV = false;
#endif

// TAB[d] for the byte offset 8 d the matrix instructions leave in the lane's register
__device__ __forceinline__ double table_value(const double *tab_s, int off)
{
#ifdef HIBAG_ABL_NOTAB
	return __hiloint2double(0x3ff00000, off);
#else
	return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(tab_s) + off);
#endif
}

// a block header's end-of-cell and stored-cell masks
__device__ __forceinline__ uint32_t abl_endmask(uint32_t m)
{
#ifdef HIBAG_ABL_NOEND
	return 0u;
#else
	return m;
#endif
}
__device__ __forceinline__ uint32_t abl_storemask(uint32_t m)
{
#if defined(HIBAG_ABL_NOEND) || defined(HIBAG_ABL_NOSTORE)
	return 0u;
#else
	return m;
#endif
}

// word 1 of a header of pass 2's block stream: stored sums of the block, and their first row
__device__ __forceinline__ int abl2_stored(uint32_t w1)
{
#ifdef HIBAG_ABL2_NOSV
	return 0;
#else
	return (int)(w1 >> 25) & 15;
#endif
}
__device__ __forceinline__ uint32_t abl2_stored_row(uint32_t w1)
{
#ifdef HIBAG_ABL2_SVHOT
	return w1 & 7u;
#elif defined(HIBAG_ABL2_SVL2)                // (64 rows per sample group: 5 MB in all -- L2 hits, not L1 hits)
	return w1 & 63u;
#else
	return w1 & 0x1FFFFFFu;
#endif
}

// what stands in for the two matrix instructions of a block in the NOMFMA build
template <class V8, class V16>
__device__ __forceinline__ void abl_fake_distances(const V8 &a8, const V8 &b0, const V8 &b1, int sb, V16 &d0, V16 &d1)
{
	d0[0] = __builtin_bit_cast(float, (a8[0] ^ b0[0]) & 0xF8);
	d1[0] = __builtin_bit_cast(float, (a8[1] ^ b1[1] ^ sb) & 0xF8);
}

// pass 2's clock stamps
#ifdef HIBAG_ACCUM_STAMPS
#define ACCUM_STAMP_N 8
#define ACCUM_STAMP(p) do { const unsigned long long now_ = __builtin_readcyclecounter();                                   \
	if (lane == 0) atomicAdd(&stamp_s[p], now_ - stamp_t); stamp_t = now_; } while (0)
#else
#define ACCUM_STAMP(p) do { } while (0)
#endif

#endif
