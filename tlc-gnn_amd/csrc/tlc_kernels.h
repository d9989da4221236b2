// tlc_kernels.h -- kernel parameter blocks and tier constants shared by the .hip files and the host API.
#pragma once
#include <stdint.h>

// per-pair status beyond the public ones: vicinity does not fit the packed 16-bit local ids
#ifndef TLC_ST_TOO_LARGE
#define TLC_ST_TOO_LARGE 5
#endif

// Size tiers of the PD kernel (nodes / undirected edges of the vicinity subgraph).
//   SMALL : one wavefront per subgraph, all state in ~6.5 KB of LDS   (>= 16 waves per CU)
//   MEDIUM: 256 threads per subgraph, ~36 KB of LDS, 128 VGPRs         (4 workgroups per CU)
//   LARGE : 512 threads per subgraph, ~140 KB of LDS, weights stay in HBM/L2 (1 workgroup per CU)
//   HUGE  : 256 threads per subgraph, state in a per-workgroup HBM scratch slot (any size < 65536 nodes)
#define TLC_TIER_SMALL 0
#define TLC_TIER_MEDIUM 1
#define TLC_TIER_LARGE 2
#define TLC_TIER_HUGE 3
#define TLC_TIER_MID 4      /* the lower end of MEDIUM (reported with it): 128 threads, ~10 KB of LDS, 8 workgroups per CU */
#define TLC_TIER_MEDHI 6    /* a chunk on its own only: the MEDIUM-sized vicinities with many Pos edges or beyond the compact configuration
                               (reported with MEDIUM): the wide kernels, launched FIRST and on the critical stream, so that their long
                               serial cycle swaps overlap the rest of the MEDIUM tier */
#ifndef TLC_MH_MIN_POS             /* (overridable for the threshold sweep: tools/gpu_threshold_sweep.sh) */
#define TLC_MH_MIN_POS 120
#endif
#define TLC_TIER_TINY 5     /* the lower end of SMALL (reported with it): ONE LANE per subgraph, 64 subgraphs per wavefront (pd_tiny.hip) */
#define TLC_TIER_MEDWIDE 7  /* pipelined chunks only (no MEDHI list there): the upper end of MEDIUM (reported with it), more than TLC_C_NMAX
                               nodes or TLC_C_MMAX edges.  MEDIUM runs the kernels sized for TLC_C_NMAX / TLC_C_MMAX (26 KB of LDS, six
                               workgroups per CU; cycle swap 14 KB, eleven wavefronts per CU), MEDHI and this tier the ones sized for
                               TLC_M_NMAX / TLC_M_MMAX (37 KB, four; 20 KB, eight) */
// hard limits of one subgraph: local node ids are packed in 16 bits, edge ranks + 1 in 24
#define TLC_MAX_SUBGRAPH_NODES 65535
#define TLC_MAX_SUBGRAPH_EDGES ((1 << 24) - 2)
#define TLC_N_TIERS 8
/* pairs per block (= threads) of tlc_scan_bin; a chunk has at most 2^20 pairs: TLC_SCAN_MAX_BLOCKS block flags / sums per workspace
 * (round 6: 512 / 256 -- smaller workgroups are easier to place on a full machine -- measured: pipelined batch equal, one batch alone +1 / +2 %) */
#ifndef TLC_SCAN_BLOCK
#define TLC_SCAN_BLOCK 1024
#endif
#define TLC_SCAN_MAX_BLOCKS ((1 << 20) / TLC_SCAN_BLOCK)
/* the TINY list once more, by size class: bin b holds the vicinities with (n + m) / TLC_TINY_BIN_W == TLC_TINY_BINS - 1 - b (largest
   first); the lane-per-subgraph kernel takes 64 consecutive entries of ONE bin per wavefront (it waits for its slowest lane) */
/* Pos edges from which a vicinity of a 256-thread tier takes the divide and conquer (ext1_dc.h has the measurements); the scan counts them */
#ifndef TLC_DC_MIN_POS_SHARED
#define TLC_DC_MIN_POS_SHARED 320
#endif
#ifndef TLC_TINY_BINS            /* (overridable for an A/B of the bin width) */
#define TLC_TINY_BINS 8
#define TLC_TINY_BIN_W 5
#endif

#ifndef TLC_T_NMAX               /* (overridable together: make EXTRA="-DTLC_T_NMAX=20 -DTLC_T_MMAX=32 -DTLC_TINY_BINS=8 -DTLC_TINY_BIN_W=7") */
#define TLC_T_NMAX 16
#define TLC_T_MMAX 24
#endif
/* which SMALL-tier vicinities the scan sends there (<= the kernel's capacity above; overridable for the sweep) */
#ifndef TLC_T_NCUT
#define TLC_T_NCUT TLC_T_NMAX
#endif
#ifndef TLC_T_MCUT
#define TLC_T_MCUT TLC_T_MMAX
#endif
#define TLC_S_NMAX 64
#define TLC_S_MMAX 128
#define TLC_D_NMAX 128
#define TLC_D_MMAX 256
#define TLC_D_THREADS 128
#define TLC_M_NMAX 512
#define TLC_M_MMAX 1024
/* the compact configuration of the MEDIUM-sized tiers (99.6 % of the MEDIUM-sized vicinities of the PubMed-shaped graph's non-edge
   list fit, 86 % of its edge batch's: tools/strong_stats.py) */
#ifndef TLC_C_NMAX                 /* (overridable: -DTLC_C_NMAX=512 -DTLC_C_MMAX=1024 is the build without the compact configuration) */
#define TLC_C_NMAX 384
#define TLC_C_MMAX 512
#endif
#define TLC_L_NMAX 2048
#define TLC_L_MMAX 4096
#define TLC_L_THREADS 512   /* 1024 measured slower (0.99 vs 0.94 ms): barriers over 16 wavefronts, 128-VGPR cap */
#define TLC_HUGE_MIN_TABLE 16384  /* bytes reserved for the image table in a HUGE scratch slot */

// Node record of tlc_extract_kernel (extract.hip): what a sweep needs of a member node in ONE 64-byte line -- where its CSR row
// starts, its degree, its index in the heavy set (or -1) and its first four entries (n_in = min(degree, 4) of them)
struct __attribute__((aligned(64))) TlcNodeRec {
    int row_start, deg, hidx, n_in;
    int col[4];
    double w[4];
};

struct TlcVicParams {
    // graph (device CSR)
    int n_nodes;
    int nw;  // bitmap words = ceil(n_nodes / 32)
    const int* rowptr;
    const int* col;
    const double* w;
    // batch
    const int* pairs;
    int n_pairs;
    int hop;
    unsigned flags;
    int res;
    // per-workgroup scratch slot: 4 * n_nodes + 1 ints (two frontiers, id list, row offsets)
    int* scratch;
    long long scratch_stride;
    // per-pair header
    int* hdr_n;
    int* hdr_m2;
    int* hdr_lu;
    int* hdr_lv;
    // finished-in-COUNT outputs
    double* out_pi;
    unsigned char* out_status;
    int* out_n;  // optional (tlc_vicinity_filtration)
    int* out_m;  // optional (tlc_vicinity_filtration)
    // FILL
    long long* edge_off;
    unsigned* A_dir;
    double* A_lw;
    // COUNT writing the MID / MEDIUM vicinities itself (rows still warm, no second pass over the pair): the arena offset comes
    // from a bump counter; a vicinity that does not fit below bump_cap is counted in *bump_overflow and the whole chunk then
    // takes the scan + FILL path.  Null: COUNT writes only the SMALL tier's fixed slots.
    unsigned long long* bump_top;
    long long bump_cap;
    int* bump_overflow;
    const long long* ids_off;  // optional id output (tlc_vicinity_filtration)
    int* out_ids;
    // FILL scheduling: 0 = every pair, 1 = only the pairs in fill_list[0..fill_count) (the heavy tiers go first so that
    // their long serial tails start early), 2 = every pair that is NOT in a heavy tier
    int fill_mode;
    const int* fill_list;
    int fill_count;
    const int* work_count_dev;   // fill_mode 1: the list length lives on the device (clamped to fill_count); null: fill_count
    int scratch_base_slot;       // first scratch slot of this launch (concurrent launches use disjoint slot ranges)
    // work distribution: null = pairs statically strided over the workgroups; else chunks of work_chunk (strided) pairs are
    // taken from this counter (zeroed per launch), which evens out the heavy-tailed per-pair cost
    int* work_counter;
    int work_chunk;
    // early pass (COUNT over the predicted-heavy pairs, ahead of and concurrent with the main COUNT): a vicinity that
    // turns out to be LARGE-tier takes the next fixed-size slot of the early arena (2*TLC_L_MMAX entries), is written
    // there at once and appended to early_list, so that its tier kernel starts without waiting for the scan of the batch
    int* early_list;             // [early_cap] pair indices (null: not an early pass)
    int* early_count;
    int early_cap;
    unsigned* early_dir;
    double* early_lw;
    int* started;                // early pass: workgroups with work count themselves here once resident (null: no count)
    // fixed-size slots (2*TLC_S_MMAX entries per pair) for the vicinities of the SMALL tier, written by the COUNT pass
    unsigned* small_dir;
    double* small_lw;
    unsigned long long* dbg;   // PHASE_DEBUG builds: per-phase cycle sums of the COUNT pass (null otherwise)
    unsigned long long* dbg_pair_t;   // PAIR_TIMES builds: [n_pairs][16] wall-clock stamps (100 MHz) along a pair's way through tlc_extract_kernel
    // ---- tlc_extract_kernel (extract.hip): precomputed structure of the graph ---------------------------------------
    const int* bptr;            // [n_nodes + 1] ball lists: ball_hop(x) = bcol[bptr[x] .. bptr[x+1]), ascending ids, x included
    const int* bcol;
    const TlcNodeRec* nrec;     // [n_nodes] node records; hidx = index of a HEAVY node (one of the hh_k highest degrees >= 32) or -1
    const double* hh_w;         // [hh_k][hh_k] weight of the entry a -> b between heavy nodes, 0 = not adjacent
    int hh_k;
    // main pass beside an early pass: a pair whose smaller ball has at least skip_threshold nodes belongs to the early pass as
    // long as the candidate list held every such pair (*skip_count <= skip_cap); null: no early pass
    int skip_threshold;
    const int* skip_count;
    int skip_cap;
    // pairs binned by tlc_classify_kernel (smaller ball >= 256 / 128 / 64 nodes): extracted first, in that order
    const int* big_count;       // [3] on the device; null: no bins
    const int* big_list;        // [4][n_pairs]
    // arena regions: workgroup b writes its vicinities from (region_base_wg + b) * region_entries on, then into blocks taken from
    // bump_top (relative to bump_base)
    int x_fill;                 // 1: FILL pass of tlc_extract_kernel (headers exist, entries go to edge_off[i])
    int region_base_wg;
    long long region_entries;
    long long region_base_entries;   // arena entries in front of this launch's regions
    long long bump_base;
    // ---- ball subgraphs (round 5): for every node x whose ball has at most TLC_BE_CAP nodes, the directed entries of the graph
    // with BOTH ends in ball(x), as (position of the source in the ball list << 16 | position of the target) and the weight, sources
    // ascending, a source's entries in CSR order.  A vicinity is a subset of the smaller ball of its pair, so its induced subgraph
    // is a filter over that list (extract.hip, x_sweep_ball).  Null: not built.
    const int* be_ptr;          // [n_nodes + 1]; be_ptr[x] == be_ptr[x + 1] for a node whose ball is larger
    const unsigned* be_pos;
    const double* be_w;
    // ---- ball bitmaps (round 6): row x = the members of ball(x) as bb_nw words of 32 bits (n_nodes * bb_nw words: 48 MB for PubMed,
    // resident in L2 / Infinity Cache like the lists).  The subgraph-list launch tests the smaller ball's members against the LARGER
    // ball's row with one gather instead of marking that ball in an LDS bitmap per pair and clearing it again (half of a pair's time
    // in that launch), and needs no N-bit bitmap in LDS: its launch runs with nw = 0.  Null: not built (a graph whose N^2 / 8 bytes
    // are beyond the budget of api.hip: the LDS bitmap serves as before).
    const unsigned* bbits;
    int bb_nw;
    int fast_split;             // 1: a launch of tlc_extract_kernel<64, true> takes the pairs the subgraph lists serve; this one leaves them alone
    int early_min_ball;   // > 0: the early pass owns the pairs whose smaller ball has at least this many nodes (tlc_classify_kernel's candidates):
                          // the subgraph-list launch leaves them alone
};
#define TLC_BE_CAP 512
#define TLC_BE_REG_CAP 128       /* up to here the member masks of a pair stay in registers (two ballots); above: a table in LDS */

struct TlcScanParams {
    int n_pairs;
    const int* hdr_n;
    const int* hdr_m2;
    long long* block_agg;   // [n_blocks] block sums (chained scan)
    int* block_flag;        // [n_blocks] zeroed per launch: 1 once block_agg[b] is valid
    int* sync;              // [4] zeroed per launch, 8-byte aligned: block-index ticket, completion count, u64 arena total
    long long* totals;      // [1] arena entries (device copy)
    long long* edge_off;
    int* tier_count;  // [TLC_N_TIERS]
    int* tier_list;   // [TLC_N_TIERS][n_pairs]
    int small_arena;
    int mh_min_pos;         // MEDIUM-sized vicinities with at least this many Pos edges go to the MEDHI list (TLC_MH_MIN_POS; INT_MAX: none)
    int mh_compact_only;    // 1: ... only those within the compact configuration (the wide ones keep the MEDWIDE list): the MEDHI list is then the
                            // front part of the compact MEDIUM launch (TlcPdParams::tier_list_hi), pipelined chunks
    int tiny_ok;            // the SMALL-tier vicinities of at most TLC_T_NMAX nodes / TLC_T_MMAX edges get a list of their own
    int* dcm_count;         // device counter (zeroed per chunk): MEDHI / MEDWIDE vicinities with enough Pos edges for the divide and conquer
    int* h_dcm;             // mapped host memory: that count
    int* tiny_bin_count;    // [TLC_TINY_BINS] device counters (zeroed per chunk), null: no size bins
    int* tiny_bin_list;     // [TLC_TINY_BINS][n_pairs]
    int* h_tiny_bins;       // mapped host memory: the bin counts
    // COUNT wrote the MID / MEDIUM vicinities at bump-allocated offsets (TlcVicParams::bump_top): unless *bump_overflow, only
    // the heavy tiers still need arena space, handed out above *bump_top; null: every vicinity outside the SMALL tier does
    const unsigned long long* bump_top;
    long long bump_base;    // arena entries in front of the bump area (the extraction's per-workgroup regions)
    const int* bump_overflow;
    int* h_overflow;        // mapped host memory: *bump_overflow
    // pairs the early pass has already written (they are left out of the arena and of the tier lists); null: none
    const int* early_list;
    const int* early_count;
    int early_cap;
    int* h_early;           // mapped host memory: number of early pairs (statistics)
    // mapped host memory the last block publishes into (api.hip, HostSync)
    long long* h_total;
    int* h_tier;
    unsigned* h_seq;
    unsigned seq;
};

struct TlcPdParams {
    // batch mode inputs (arena written by the FILL pass)
    const int* tier_list;  // pair indices of this tier
    int tier_count;
    // (round 6) list positions [0, n_hi) come from tier_list_hi, the rest from tier_list[wi - n_hi]: the compact MEDIUM list of a pipelined
    // chunk with its many-Pos vicinities in front (the scan's MEDHI list), so that the longest walks of the swap kernel start first
    const int* tier_list_hi;
    int n_hi;
    const int* hdr_n;
    const int* hdr_m2;
    const int* hdr_lu;
    const int* hdr_lv;
    const long long* edge_off;
    const unsigned* A_dir;
    const double* A_lw;
    const unsigned* small_dir;   // SMALL tier: fixed-size slots written by the COUNT pass (null: use the arena)
    const double* small_lw;
    unsigned flags;
    int res;
    // outputs
    double* out_pi;             // [n_pairs, res*res]
    unsigned char* out_status;  // may be null
    // optional filtration output (tlc_vicinity_filtration)
    const long long* ids_off;
    double* out_f;
    int* out_n;
    const long long* edges_off;  // optional induced-edge output (tlc_vicinity_filtration)
    int* out_edges;
    int* out_m;
    int pi_enabled;
    // HUGE tier: per-workgroup scratch in HBM
    unsigned char* huge_scratch;
    long long huge_stride;
    int huge_nmax;
    int huge_mmax;
    int huge_slots;
    int huge_lds;           // HUGE tier: dynamic LDS bytes of the launch (tables of the serial cycle swap), 0 = none
    // lane-per-subgraph kernel: the TINY list by size class (null: the plain tier list)
    const int* tiny_bin_list;   // [TLC_TINY_BINS][tiny_bin_stride]
    int tiny_bin_stride;
    int tiny_bin_cnt[TLC_TINY_BINS];
    // statistics: [0] sources that took the exact tie fallback
    unsigned long long* stats;
    // diagnostics (null in production): per tier 16 accumulated cycle counts of thread 0, see pd_pipeline.hip
    unsigned long long* phase_cycles;
    int* started;      // LARGE tier: workgroups that have begun (the launcher holds the small tiers back until then)
    // hand-off slots of this tier ([tier_count] x handoff_stride bytes) between the tier kernel and tlc_pd_swap_kernel;
    // null: the tier kernel runs the cycle swap and the image itself
    unsigned char* handoff;
    long long handoff_stride;
    // early LARGE launch: the list length lives on the device (grid = capacity), subgraph wi sits in slot wi of the early
    // arena (A_dir / A_lw point at it), and the workgroups that have work count themselves in `started`
    const int* tier_count_dev;
    long long slot_entries;      // != 0: entry offset of list position wi is wi * slot_entries instead of edge_off[i]
    // speculative launch (submitted behind the scan without waiting for the host to see the sizes): grid > 0 overrides the
    // grid (the workgroups stride over the list, whose length is tier_count_dev), only the first handoff_cap list positions
    // have a hand-off slot (the rest run the cycle swap in the tier kernel), and a set *abort_flag (COUNT overflowed the
    // arena: the chunk is redone by the scan + FILL path) makes every workgroup return at once
    // subgraphs a tier kernel hands to tlc_pd_dc_kernel (many Pos edges: the cycle swap as a divide and conquer): their list
    // positions are appended to dc_list (*dc_count entries, zeroed per chunk) so that the kernel runs with a small grid and costs
    // nothing when there are none; what it cannot finish (ranks that are no minimum-spanning-tree order) it finishes itself with
    // the serial walk
    int* dc_count;
    int* dc_list;
    int dc_inplace;              // LARGE tier: the tier kernel's own workgroup runs the divide and conquer from the record (no tlc_pd_dc_kernel launch)
    int dc_force_fail;           // tests: tlc_pd_dc_kernel treats every solve as failed (the give-back path to the serial walk)
    int no_plain;                // 1: tlc_launch_pd_tier takes the general kernel instances even for a plain image batch (option plain_kernels = 0)
    int grid;
    int wi_base;                   // LDS tiers: list position of workgroup 0 (a launch that completes a shorter one)
    int phase;                   // 0 = tier kernel + its swap kernel, 1 = tier kernel only, 2 = swap kernel only
    int handoff_cap;
    const int* abort_flag;
};


// PD from a caller-supplied filtration (tlc_pd_from_filtration)
struct TlcPdfParams {
    const int* list;  // graph indices of this tier
    int count;
    const long long* node_offs;
    const long long* edge_offs;
    const int* edges;  // int32[sum m, 2]
    const double* f;
    unsigned flags;
    double* pd_up;
    double* pd_down;
    double* pd_one;
    double* ext0;
    int* counts;
    int* edge_rank;
    unsigned char* huge_scratch;
    long long huge_stride;
    int huge_nmax;
    int huge_mmax;
    int huge_slots;
};

#ifdef __HIPCC__
template <bool FILL, int BW>
__global__ void tlc_vicinity_kernel(TlcVicParams p);
__global__ void tlc_scan_bin(TlcScanParams p);
#endif

// host-side launchers implemented next to their kernels
int tlc_launch_pd_tier(int tier, const TlcPdParams& p, void* stream);
int tlc_launch_pd_tiny(const TlcPdParams& p, void* stream);
int tlc_launch_pdf_tier(int tier, const TlcPdfParams& p, void* stream);
int tlc_launch_pdf_bin(int n_graphs, const long long* node_offs, const long long* edge_offs, int* counts, int* tier_count,
                       int* tier_list, void* stream);
size_t tlc_extract_lds_bytes(int nw, int bw, bool fast = false);
int tlc_launch_extract(int bw, int grid, size_t lds, const TlcVicParams& p, void* stream, bool fast = false);
int tlc_launch_classify(int n_pairs, const int* pairs, int n_nodes, const int* bptr, int cand_threshold, int cand_cap,
                        int* cand_count, int* cand_list, int* big_count, int* big_list, void* stream);
int tlc_launch_tiny_sort(int count, const int* list, const int* hdr_n, const int* hdr_m2, int* out, void* stream, int shift = 0);
int tlc_launch_ball_edges(bool fill, int n_nodes, int nw, const int* rowptr, const int* col, const double* w, const int* bptr, const int* bcol,
                          int* esize, const int* be_ptr, unsigned* be_pos, double* be_w, int grid, void* stream);
int tlc_launch_ball_bits(int n_nodes, int nw, const int* bptr, const int* bcol, unsigned* bbits, void* stream);   // rows zeroed by the caller
int tlc_launch_ball_list(bool fill, int n_nodes, int nw, const int* rowptr, const int* col, int hop, int* bsize, const int* bptr,
                         int* bcol, int grid, void* stream);
int tlc_launch_ball_bound(int n_nodes, const int* rowptr, const int* col, const int* prev, int* out, void* stream);
int tlc_launch_select_heavy(int n_pairs, const int* pairs, int n_nodes, const int* ub, int threshold, int cap, int* count,
                            int* list, void* stream);
size_t tlc_huge_slot_bytes(int nmax, int mmax);
size_t tlc_handoff_slot_bytes(int tier);   // 0: the tier has no hand-off (its kernel runs the cycle swap itself)
int tlc_launch_copy_sizes(int n_pairs, const int* hdr_n, const int* hdr_m2, int* out_n, int* out_m, void* stream);
int tlc_launch_pi_raster_wgrad(int n_dgms, long long n_pts, const long long* offs, const double* pts, int res, const double* grad_img,
                               double* grad_pts, void* stream);
int tlc_launch_pi_raster(int n_dgms, const long long* offs, const double* pts, int res, double* out, void* stream);
