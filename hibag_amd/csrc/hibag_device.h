// hibag_device.h -- device-side views shared by the kernels and the host code
// of libhibag_hip.so.  gfx950 only.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//
//  MODEL (read-only).
//    The loop nest of _PostProb2 (src/LibHLA.cpp:1776-1821) depends only on the
//    model: all 64 lanes (samples) of a wavefront walk it together, in exactly the
//    reference's visiting order (allele pair h1<=h2, then haplotype i1, then i2).
//    For pass 2 the posterior cells are grouped in TILES of up to HIBAG_TILE consecutive
//    cells with about equal work.
//
//    VALU-engine classifiers (more than 112 SNPs)
//    get the nest flattened into a PAIR STREAM, one record per haplotype pair, fetched with scalar loads:
//        W[nwp]  the 3k-bit string  H1 | H2 << k | ~(H1^H2) << 2k   (k = #SNPs)
//        prod    the frequency factor, rounded as the reference rounds it:
//                f1*f1 for the leading diagonal term, (2*f1)*f2 otherwise
//    Records are grouped in CHUNKS of HIBAG_CHUNK, SoA inside a chunk
//    (W[nwp][CHUNK] then prod[CHUNK]); every allele-pair cell owns whole chunks,
//    padded with {W=0, prod=+0.0} records (adding +0.0*TAB[d] is exact).
//    Cells without haplotype pairs own no chunk.  cls_cnt lists, per classifier,
//    the chunk counts of its non-empty cells in posterior order (pass 1 walks
//    it); tile_meta[c][t] = { #non-empty cells, first
//    chunk of the tile, the row numbers j of the non-empty cells packed 4 bits each
//    (2 dwords), then one entry (j << 24 | chunks) per cell of the tile: the
//    non-empty ones in order, then the empty ones }.
//
//    The matrix-core engine (classifiers with at most 112 SNPs) does not read pair
//    records at all: it GENERATES them from an O(H) haplotype table
//        entry i of classifier c = { image of the haplotype's bits (bytes or nibbles, by engine), ff (double), f (double) }   48 / 32 bytes
//    where f is the haplotype's frequency and ff = 2 f the factor it contributes as the FIRST haplotype
//    of a pair (the reference's `ff = 2 * f1`, then `ff * f2`).  Entries H+1 .. 2H repeat the haplotypes
//    with ff = f: the leading diagonal pair (i, i) of a cell (h, h), whose factor is f * f, is listed as
//    (H + 1 + i, i).  Entry H is all zero and serves the padding slots.  The records are 4-byte
//    words  i1 | i2 << 16 | store << 30 | end << 31  (end = this slot closes a cell, store = pass 1 keeps the cell's
//    sum for pass 2; i2 < 2^14), in BLOCKS of 32 slots
//    (128 bytes); unused trailing slots of a segment's last block point at the zero entry.
//    Each lane builds its record's A-operand rows from the two images (element-wise sum and AND)
//    and the frequency factor ff[i1] * f[i2] -- one multiplication, rounded like the reference's
//    (src/LibHLA.cpp:1786-1813) -- itself.  Cells are
//    padded to an even slot count.  Two lists: all cells of a classifier back to back
//    (pass 1), and per (tile, classifier) segments stored tile-major (pass 2: a wavefront
//    walks one tile's segments classifier after classifier through contiguous memory); the pass-2
//    segments leave out the cells whose sums pass 1 stores (those with many pairs).
//
//  BATCH ("lane = sample": consecutive samples are consecutive addresses, so
//  every per-lane access is one coalesced row segment of a wavefront):
//    masks : uint32 [mask_rows][n_pad]; classifier c owns rows mask_row[c] +
//            {0..nwp-1: XOR mask T', nwp..2nwp-1: AND mask M'} so that the
//            distance of src/LibHLA.cpp:747-819 is  sum_w popc((W[w]^T'[w])&M'[w])
//    cw, tot, inv : double [C][n_pad]  classifier weight, in-order posterior
//            total, 1/total
//    part  : double [P+3][n_pad]  ensemble sums per allele pair + 3 scalars
#ifndef HIBAG_DEVICE_H_
#define HIBAG_DEVICE_H_

#include <stdint.h>
#include <hip/hip_vector_types.h>

#define HIBAG_WAVE 64
#define HIBAG_TAB_N 257          // 2*128 + 1 distances (src/LibHLA.cpp:167)
#ifndef HIBAG_TILE
#define HIBAG_TILE 15            // allele-pair cells per tile (one LDS accumulator row each; at most 16).  Fifteen, not sixteen: pass 2's
                                 // workgroup then needs 4 x 15 x 64 x 8 = 30,720 bytes of accumulators (+ 512 of table), and FIVE of
                                 // them fit a CU's 160 KB instead of four (measured: k_accum -4 %, k_accum_cells -19 %)
#endif
#ifndef HIBAG_CHUNK
#define HIBAG_CHUNK 4            // pair records per chunk
#endif
// wavefronts per workgroup of pass 1 (each on its own work item: four sample groups on one classifier) and of pass 2 (the
// sample groups that share a tile's blocks in L1); the host sizes the hand-over flags by them
#ifndef HIBAG_BLOCK_WAVES
#define HIBAG_BLOCK_WAVES 4
#endif
#ifndef HIBAG_ACCUM_WAVES
#define HIBAG_ACCUM_WAVES 4
#endif
#define HIBAG_TILE_META (4 + HIBAG_TILE)   // dwords of one tile_meta entry
static_assert(HIBAG_TILE <= 16, "cell rows are packed 4 bits each");
#define HIBAG_MAX_NWP 12         // ceil(3*128/32) words of the packed pair string

// dwords of one chunk for a classifier with nwp words per record
#define HIBAG_CHUNK_DWORDS(nwp) (HIBAG_CHUNK * ((nwp) + 2))
// dwords of one 32-slot block of a pair list (matrix-core engine)
#define HIBAG_PLIST_DWORDS 32
// slot flags (the kernels take them from the block headers, HibagModelView::phdr; the host builds the headers from the words)
#define HIBAG_PLIST_END 0x80000000u          // the slot closes a cell
#define HIBAG_PLIST_STORE 0x40000000u        // with END: pass 1 stores this cell's sum for pass 2 to read back
#ifndef HIBAG_STORED_PER_VISIT
#define HIBAG_STORED_PER_VISIT 7             // mode 2: stored cells per block of pass 2 (and per (classifier, tile) visit of a classifier it evaluates) -- what k_accum
                                             // keeps in registers; at most 8.  Seven: with eight k_accum needs 97 vector registers, one more than five
                                             // wavefronts per SIMD leave it (96) -- four spilled; with seven it needs 95
#endif
// Matrix-engine variants (HibagModelView::engine[c]; 0 = VALU engine) and their haplotype-table entries:
//   FP4  (up to 30 SNPs; 33 .. 112 in several K steps, below)  v_mfma_scale_f32_32x32x64_f8f6f4 with e2m1 operands: one instruction per sample half covers all
//                         K = 64 positions.  Entry = { N[16], W[16], ff, f }: 12 dwords.  N ("sum" image, fetched by lanes 0..31) has nibble
//                         s = 2 (the code of 1.0) where bit s is set and the codes 1, 3 (0.5, 1.5) at nibbles k, k+1 -- two of them add up
//                         to the A-row constants 1 and 4 of the lower K half; W ("pair" image, lanes 32..63) has the code 3 (1.5) where
//                         bit s is set and at nibbles k, k+1 -- two of them add up to the code 6 (4.0).  The A row of EVERY lane is the
//                         plain sum of its two images: neither the AND nor the offset digits' constants cost an instruction.
//                         (Several K steps: { N[16], ff, f, N'[16] ... }, 8 + 4 (steps - 1) dwords; AND and constants made by the kernel.)
//   I8   (31 SNPs), I8S (32 SNPs)  v_mfma_i32_32x32x32_i8, two K blocks.  Entry = { E[32] (byte s = bit s), ff, f }: 12 dwords
#define HIBAG_ENGINE_VALU 0
#define HIBAG_ENGINE_FP4 1
#define HIBAG_ENGINE_I8 2
#define HIBAG_ENGINE_I8S 3
// FP4 in several K steps (33 .. 112 SNPs): the SNPs are cut into groups of 28, each group a complete FP4 operand pair of its
// own (its SNPs, its share of the offset), and the group's products are chained through the accumulator operand -- a
// denormal accumulator survives the instruction exactly (tools/mfma_fp4_probe, trial 3).  Entry = the FP4 entry followed by
// one more 16-byte image per further step; B operands: two rows per step.
#define HIBAG_FP4_STEP_SNPS 28
#define HIBAG_FP4_MAX_STEPS 4
#define HIBAG_FP4_STEPS(k) (((k) + HIBAG_FP4_STEP_SNPS - 1) / HIBAG_FP4_STEP_SNPS)
#define HIBAG_FP4_MAX_SNPS 30     // one K step: 2 k + 4 positions (the SNPs' sums and ANDs, four offset digits) within K = 64
#define HIBAG_ENGINE_OF(k, fp4) ((fp4) && (k) <= HIBAG_FP4_MAX_SNPS ? HIBAG_ENGINE_FP4 : (k) < 32 ? HIBAG_ENGINE_I8 : (k) == 32 ? HIBAG_ENGINE_I8S : \
	(fp4) && (k) <= HIBAG_FP4_STEP_SNPS * HIBAG_FP4_MAX_STEPS ? HIBAG_ENGINE_FP4 : HIBAG_ENGINE_VALU)
#define HIBAG_ENGINE_STEPS(e, k) ((e) == HIBAG_ENGINE_FP4 ? (k) <= HIBAG_FP4_MAX_SNPS ? 1 : HIBAG_FP4_STEPS(k) : 1)
#define HIBAG_ENGINE_ROWS(e, k) ((e) == HIBAG_ENGINE_FP4 ? 2 * HIBAG_ENGINE_STEPS(e, k) : ((e) == HIBAG_ENGINE_VALU ? 0 : 4))   // B-operand rows (16 B per lane each)
#define HIBAG_ENGINE_HAP_DWORDS(e) 12                                                             // (FP4: of a one-step entry)
#define HIBAG_FP4_ENTRY_DWORDS(steps) (8 + 4 * ((steps) - 1))
// K layout of the distance dot product for a classifier with k SNPs.  With the genotype g of the sample at SNP s:
//   g = 0: h1 + h2      g = 2: 2 - h1 - h2      g = 1: [h1 == h2] = 1 - h1 - h2 + 2 h1 h2        (src/LibHLA.cpp:747-819)
//   8 d = sum_s (h1_s + h2_s) * 8 t_s  +  sum_s (h1_s & h2_s) * 16 [g_s == 1]  +  8 * offset,
//   t_s = +1 / -1 / -1 / 0 for g = 0 / 1 / 2 / missing,   offset = 2 #[g == 2] + #[g == 1]
// I8 / I8S (K positions = bytes of the int8 operands):
//   [0, k)          A: h1_s + h2_s (0/1/2)   B: +8 / -8 / -8 / 0
//   [32, 32 + k)    A: h1_s & h2_s (0/1)     B: 16 [g == 1]
//   31              A: 8                     B: offset            (32 SNPs: position 31 is taken, the offset
//                                                                   starts the accumulators instead: HibagBatchView::bias)
// FP4 (K positions = nibbles; the lower K half [0, 32) of B is scaled by 2^-73, the upper one by 2^-72, A by 2^-73, so
// that the f32 result is the DENORMAL number 8 d * 2^-149, whose bit pattern is the integer 8 d):
//   [0, k)          A: S = h1_s + h2_s       B: +1 / -4 / -1 / 0  for g = 0 / 1 / 2 / missing
//   [32, 32 + k)    A: w(S) = 0 / 1.5 / 4    B: [g == 1]          (counts twice through the scale)
//                   -- h1 & h2 = w - 1.5 S, so 2 (h1 & h2) [g == 1] - S [g == 1] = 2 w [g == 1] - 4 S [g == 1]: the sum of two
//                   "pair" images (code 3 + 3 = 6) stands in for the AND, and g = 1 costs -4 instead of -1 in the lower half
//                   (classifiers of several K steps keep A: h1 & h2 against B: -1)
//   k, k + 1        A: 1, 4                  B: offset & 3, (offset >> 2) & 3      (the values 0 .. 3: all e2m1 numbers)
//   32 + k, + 1     A: 4, 4                  B: offset bit 4 * 2, bit 5 * 4        (count twice; offset <= 2 k <= 60)
// so one K step holds up to 30 SNPs (round 2 spent six positions on the offset: 28).
#define HIBAG_FP4_SCALE_A 54
#define HIBAG_FP4_SCALE_B_LO 54
#define HIBAG_FP4_SCALE_B_HI 55

struct HibagModelView {
	int n_hla;
	int n_classifier;
	int n_snp;          // SNPs in the model (row length of the raw genotype matrix)
	int n_cell;         // n_hla*(n_hla+1)/2
	int mask_rows;      // sum_c 2*nwp_c
	int n_tile;

	const int *n_snp_c;          // [C]
	const int *nwp;              // [C] 32-bit words of the packed pair string = ceil(3*n_snp_c/32)
	const int *snp_off;          // [C]
	const int *snp_index;        // concatenated 0-based SNP indices
	const int *snp_weight;       // [n_snp] #classifiers using the SNP (src/LibHLA.cpp:2484-2496)
	const int *mask_row;         // [C] first mask row of the classifier
	const int *c_order;          // [C] classifiers sorted by pair count, heaviest first
	const int *tile_p0;          // [n_tile] first cell (posterior index) of the tile
	const int *tile_n;           // [n_tile] cells in the tile (1..HIBAG_TILE)
	const uint32_t *tile_meta;   // [C][n_tile][HIBAG_TILE_META]
	const uint32_t *cls_cnt;     // per classifier: chunks of each non-empty cell (+1 pad)
	const uint32_t *cls_cell;    // same layout: posterior index of each non-empty cell
	const int *cls_off;          // [C] offset of the classifier's list in cls_cnt / cls_cell
	const int *cls_n;            // [C] number of non-empty cells
	// pass-1 work items (heaviest first): {classifier, first cell, end cell, first chunk} in the classifier's
	// non-empty cell list.  A VALU-engine classifier with far more work than the others is cut into
	// several items that store their cell sums (split_row[c] >= 0: the classifier is split,
	// -1 = not split); k_total_scan then adds them in order.
	int n_item, n_split;
	int all_fp4;                 // every pass-1 work item is a one-step FP4 classifier (k_total's denser build carries that loop only)
	const int *item;             // [n_item][4]: the launcher points this at the split or the whole list
	int n_item_whole, n_item_split;
	const int *item_whole, *item_split;
	double split_heavy_ns, split_rest_ns;   // cost model: split when heavy > rest * groups / resident wavefronts
	const int *split_row;        // [C]
	const int *split_cls;        // [n_split] the split classifiers
	const uint64_t *stream_off;  // [C] dword offset of the classifier's stream
	const uint32_t *stream;      // the pair records of the VALU-engine classifiers
	const double *tab;           // [257] exp(d*log(1e-5))

	// matrix-core engine (classifiers with at most 112 SNPs; hibag_kernels.hip "MFMA engine")
	const int *engine;           // [C] HIBAG_ENGINE_*
	const int *n_step;           // [C] K steps of the FP4 engine (1 up to 28 SNPs; HIBAG_FP4_STEPS), 1 for the others
	int n_valu;                  // classifiers on the VALU engine (more than 112 SNPs, or HIBAG_ENGINE=valu)
	int n_wide;                  // classifiers with n_step > 1: pass 1 in k_total_wide, not among the work items; all their cells stored
	const int *wide_cls;         // [n_wide]
	int n_wide_scan;             // ... of them the ones whose in-order total k_total_scan forms from the stored sums (cut into several segments)
	const int *wide_scan;        // [n_wide_scan]
	int n_wide_seg;              // their pair lists in segments of whole cells (one workgroup each per group quad):
	const int *wide_seg;         // [n_wide_seg][4] {classifier, first stored row, blocks, 1 = the segment is the whole classifier: the walk forms the total}
	const uint64_t *wide_seg_off; // [n_wide_seg] dword offset of the segment in plist
	const int *bt_row;           // [C] first operand row of the classifier in HibagBatchView::bt
	const uint32_t *hap;         // haplotype table (entry size by engine, see below)
	const uint32_t *hap_off;     // [C] dword offset of the classifier's first entry
	uint32_t hap_dwords;         // size of the table
	const uint64_t *blk_off;     // [C] dword offset of the classifier's pass-1 pair list (all cells back to back)
	const int *cls_nblk;         // [C] blocks in that list
	const uint32_t *plist;       // pair lists: blocks of HIBAG_PLIST_DWORDS dwords
	uint64_t plist_dwords;       // total size (a raw buffer is rebased per classifier / tile segment: no 4 GB limit)
	// per slot of plist its frequency factor ff[i1] * f[i2] (the rounded product of src/LibHLA.cpp:1786-1813), and per block
	// {end-of-cell mask, stored-cell mask, slots worth evaluating, 0}: both wave-uniform, read through the scalar cache
	const double *pfac;          // [plist_dwords]
	const uint32_t *phdr;        // [plist_dwords / 32][4]
	// PREBUILT A-operand rows of the one-step FP4 engine: per block of plist 64 x 16 bytes in lane order -- what lane l would
	// make of its slot's two haplotype images (their element-wise sum: lanes 0..31 the "sum" images, lanes 32..63 the "pair"
	// images).  One coalesced 16-byte load per lane replaces the slot-word load, two 16-byte gathers from the haplotype table,
	// four address instructions and four additions per block -- at 1 KB per block instead of 128 bytes.  Pass 2's E-stream
	// always has them (its blocks come first in plist); the pass-1 lists of the one-step FP4 classifiers while all of it stays
	// below HIBAG_PREBUILT_MB (default 128: the DRB1 shape would take 400 MB and every wavefront would stream its
	// classifier's 4 MB from HBM; those models keep generating their rows from the O(H) table).
	const uint4 *parow;          // [parow_blocks][64]
	uint64_t parow_blocks;       // blocks of plist, from its first on, that have rows
	int p1_prebuilt;             // 1: the pass-1 lists (k_total's FP4 walk) have rows too
	const uint32_t *ctile;       // [C][n_tile][8]: everything pass 2 needs per (classifier, tile) in one s_load_dwordx8:
	                             // {engine | k << 2 | #listed cells << 8 | (K steps - 1) << 13 | bt_row << 16    (k: SNPs of the LAST K step), dword offset of the first haplotype-table entry,
	                             //  pair list dword offset lo/hi, #blocks, first stored row | #stored cells << 27, row list lo/hi}
	                             // row list: 4 bits per cell -- the listed (evaluated) cells in closing order, then the stored ones

	// stored cells: rows of HibagBatchView::cells.  Evaluating a haplotype pair again in pass 2 costs ~0.25 ps per sample,
	// writing a cell sum in pass 1 and reading it back ~2.2 ps, so a cell with more than ~8 pairs is better stored.
	//   store_cells = 1: every cell of every classifier (models with many pairs per cell; pass 2 = k_accum_cells);
	//   store_cells = 2: the cells with many pairs of the matrix-engine classifiers (pass 2 = k_accum: the small cells from
	//                    their tile-major lists, the stored ones from memory);
	//   store_cells = 0: none (only a split VALU-engine classifier has rows, for k_total_scan).
	int store_cells;
	const int *cell_row;         // [C + 1] first row of the classifier (one row per stored cell, in cell order); cells[group][row][64]
	const uint32_t *blk_close;   // per pass-1 block: cells closed in the classifier's earlier blocks (where a chunk resumes)
	uint64_t p1_base;            // dword offset of the first pass-1 list in plist (block number = (offset - p1_base) / 32)
	long long p1_blocks;         // blocks of all pass-1 lists together

	// E-stream: what pass 2 (k_accum) reads when it evaluates pairs again (store_cells != 1).  Per tile the blocks of
	// classifier 0, 1, 2 ... that have anything for the tile, back to back -- pair slots in `plist` (32 per block, as in the
	// pass-1 lists; the kernel reads their prebuilt rows, `parow`), one 8-dword header per block in `ehdr`:
	//   [0] the block's end-of-cell mask (bit i: slot i closes a cell)
	//   [1] first stored-sum row (model-wide numbering, HibagBatchView::cells) | stored sums of this block (0..HIBAG_STORED_PER_VISIT) << 25
	//       | (the record before the block's first one closed a cell: the first product starts a sum) << 29
	//   [2], [3] the request words of the NEXT block of the stream -- its classifier | first B-operand row << 16, and its word 1:
	//            what the kernel needs to REQUEST a block's per-lane data (weight, 1/total, B operand, stored sums) it finds in
	//            the header of the block before, so that only two headers are alive at a time -- the one in use and the one in
	//            flight -- and the loop, unrolled twice, rotates nothing
	//   [4], [5] tile rows of the cells that CLOSE in this block, 4 bits each: field i / 2 for the cell that closes at slot i
	//            (cells are padded to an even number of slots: only odd slots close one)
	//   [6] tile rows of the stored sums, 4 bits each (28 bits) | groups of four slots worth evaluating (0..8) << 28
	//   [7] this block's own classifier | first B-operand row << 16 (a walk requests its FIRST block from words 7 and 1)
	// The header is all the scalar data of a block besides its slots' factors (`pfac`): one s_load_dwordx8.
	// Only one-step FP4 classifiers have pair slots here; the others' blocks carry stored sums only (all slots padding).
	const uint32_t *ehdr;        // [estream_blocks][8]
	uint64_t estream_blocks;     // blocks of all tiles together, incl. the look-ahead slack behind the last
	const uint64_t *etile_blk0;  // [n_tile] first block of the tile
	const uint32_t *etile_cstart;   // [n_tile][C + 1] block, counted from the tile's first, where classifier c's blocks begin:
	                             // where a chunked work item is cut (hibag_kernels.hip "hand-overs"), and the item's length

	// resident workgroups of k_total<STORE, occupancy> ([2 * STORE + (six per CU)]) and of k_accum on the model's device
	// (0 = unknown), queried when the model is finalized (hibag_query_slots)
	int slots_total[4], slots_accum;
};

struct HibagBatchView {
	int n_samp;         // samples in this batch
	int n_pad;          // rounded up to a multiple of 64
	uint32_t *masks;    // [mask_rows][n_pad]
	double *cw;         // [C][n_pad]
	double *tot;        // [C][n_pad]
	double *inv;        // [C][n_pad]
	double *winv;       // [C][n_pad][2] {classifier weight (k_pack), 1/total -- 0 where the weight is not positive -- (pass 1)}: what pass 2 reads per block, in ONE 16-byte load
	// Majority vote (vote_method = 2): the cell a classifier votes for is the FIRST strict maximum of cell * (1/total) in cell
	// order (src/LibHLA.cpp:2465-2475 -> :1549-1566), and 1/total is only known when pass 1 ends.  Multiplying by a positive
	// constant is monotone, so the winner is among the RECORDS of the raw cell sums -- the cells larger than every cell before
	// them -- namely the earliest record whose product equals the last record's: only records within a few ulps of the
	// maximum can tie with it (at most five doubles map to one product).  Pass 1 therefore logs its records per (classifier,
	// sample) -- entry 0 {maximum, number of records}, entry 1 the first record (what wins where 1/total is infinite: the
	// first positive cell), entries 2..7 a ring of the latest six {value, position in the classifier's cell list} -- and
	// k_vote_pick settles the ties with the products themselves.  No second walk over the pairs.
	uint4 *vrec;        // [C][8][n_pad] (16-byte entries; null unless the call votes)
	double *cells;      // [n_pad / 64][cell_row[C]][64] the cell sums pass 1 stores
	double *part;       // [P+3][n_pad]
	// matrix-core engine: per classifier and sample group the B operand tiles
	// (int8, MFMA lane layout; K layout above) and, for classifiers with 32 SNPs, the distance offsets
	uint4 *bt;          // [(bt_row[c] + n * (rows / 2) + kb)][n_pad/64][64]
	int bt_rows;        // rows of `bt` that exist (the model's + 2 spare): pass 2's buffer descriptor ends there
	int *bias;          // [(2c + n)][n_pad/64][64]  8 * (2*#(g=2) + #(g=1)), written for 32-SNP classifiers only
	// hand-over flags of chunked work items: (epoch << 32 | progress) per item; a new epoch per batch, so the
	// flags are never cleared
	unsigned long long *sync;         // pass 2: [8 XCDs][group quads][tiles]
	unsigned long long *sync_total;   // pass 1: [items x group quads]
	uint32_t epoch;
	// A hand-over that never arrives (or crosses XCDs) must not go unnoticed: the waiting workgroup sets the host-mapped
	// word `err` (what the host entries and hibag_hip_model_status() look at) AND stores the batch's epoch in the device
	// word `err_dev`; k_scalars, which runs behind both passes, then turns the sample's three ensemble scalars into NaN, and
	// every k_finish_* kernel writes NA / NaN outputs for a sample whose weight sum is NaN -- sums the kernels cannot vouch for
	// never reach the caller as numbers.
	int *err;           // host-mapped: 1 = a hand-over timed out, 2 = its flag was written on another XCD
	uint32_t *err_dev;  // device memory: [0] epoch of the last batch with a failed hand-over; [2] count and, from byte 16 on, list of
	                    // the (sample, classifier) pairs whose 1/total is not finite (pass 1 -> k_nan_cells; the host resets the count)
	uint32_t spin_limit;   // polls before a waiting workgroup gives up (scaled with the longest work item of the model)
	int tail_k;         // chunks per item of the last rounds: 0 = the launcher's choice, 1 = none (no hand-overs at all)
	int drop_post;      // fault injection (tests): 1 / 2 = the first chunked item of pass 1 / 2 never posts its first hand-over
};

#endif
