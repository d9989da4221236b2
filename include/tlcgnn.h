/*
 * tlcgnn.h -- C ABI of libtlcgnn_hip.so: the MI355X (gfx950) implementation of TLC-GNN / PDGNN's
 * per-edge topological-feature hot path (vicinity subgraph -> extended persistence diagram ->
 * persistence image -> link-prediction forward).
 *
 * The reference (pkuyzy/TLC-GNN) has no FFI layer of its own: its boundary is a set of Python call
 * signatures.  Each entry point below names the reference interface (file:line under /root/reference)
 * that it replaces; INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in signatures (`stream` is a hipStream_t
 *     passed as void*, NULL = the default stream);
 *   - pointers prefixed d_ are DEVICE pointers, h_ are HOST pointers;
 *   - the caller allocates every buffer; the library keeps nothing beyond the call except the
 *     `tlc_graph` handle (which owns a device copy of the CSR and the kernels' scratch);
 *   - every call is stream-ordered: its work is ordered after what `stream` holds and later work on `stream` sees its
 *     outputs.  The calls RETURN before that work is done, with one documented exception: tlc_pd_pi_batch,
 *     tlc_pd_pi_batch_async and tlc_vicinity_filtration hold the calling thread, once per chunk of 2^20 pairs, until a
 *     chunk's extraction and size scan have run on the device (~0.3 ms for 37 676 pairs; the sizes come back through
 *     mapped host memory and size the tier launches that follow) -- nothing is synchronised and `stream` is not drained,
 *     but the call is not free of host waiting.  A pipelined chunk (tlc_pd_pi_batch_async, or a list of several chunks) waits
 *     for the PREVIOUS chunk's sizes, after it has submitted its own first half: the tier launches of a chunk go in one call
 *     later, those of the last chunk with the join.  The statistics / timing / sizes getters synchronise `stream`.  Outputs are fully
 *     overwritten (zero rows are written explicitly, mirroring `pi_sg = np.zeros(...)`,
 *     sg2dgm/riccidist2dgm.py:363);
 *   - a tlc_graph handle owns mutable scratch: calls on the SAME handle must not overlap (use one handle per
 *     host thread / per concurrent stream); different handles are independent;
 *   - return value: TLC_OK or a TLC_ERR_* code for API misuse / runtime failure.  Per-pair conditions
 *     that the reference swallows into a zero row (`except BaseException`, riccidist2dgm.py:352-357)
 *     are reported in the per-pair status byte, never as a return code.
 */
#ifndef TLCGNN_H
#define TLCGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes ------------------------------------------------------------------------------ */
#define TLC_OK                 0
#define TLC_ERR_INVALID_ARG    1
#define TLC_ERR_HIP            2   /* a HIP runtime call failed; see tlc_last_error() */
#define TLC_ERR_NO_DEVICE      3
#define TLC_ERR_UNSUPPORTED    4   /* e.g. graph too large for the LDS-resident vicinity bitmaps */
#define TLC_ERR_OUT_OF_MEMORY  5

/* ---- per-pair status byte (the reference's swallowed exception classes) ---------- */
#define TLC_ST_OK              0   /* row computed (may still be all zero: d(u,v) > hop)            */
#define TLC_ST_MISSING_NODE    1   /* KeyError: endpoint has no edge (graph is built from edges)    */
#define TLC_ST_DISCONNECTED    2   /* AssertionError: vicinity empty or not connected (:318)        */
#define TLC_ST_ZERO_RANGE      3   /* ZeroDivisionError: all filtration values 0 (:54)              */
#define TLC_ST_NO_TREE_EDGE    4   /* IndexError: single-node vicinity (accelerated_PD.py:122)      */
#define TLC_ST_TOO_LARGE       5   /* not a reference class: the vicinity has more than 65 535 nodes or
                                      2^24-2 edges (packed local ids / edge ranks); zero row, never silent */

/* ---- variant flags (one kernel family serves the TLC-GNN and the PDGNN forks) ---- */
#define TLC_KEEP_ZERO_PERS   0x01u /* Knowledge_Distillation/accelerated_PD.py:68-69,108-109,169-170 */
#define TLC_INCLUDE_ROOTS    0x02u /* Knowledge_Distillation/data_utils_LP.py:111                    */
#define TLC_NORM_EPS         0x04u /* divide by (max + 1e-10): data_utils_LP.py:64                   */
#define TLC_PI_ORD0_EXT1     0x08u /* image over Ord0 ++ Ext1 only: data_utils_GC.py:155-163         */
#define TLC_NO_EXT1          0x10u /* extended_flag=False: riccidist2dgm.py:323-326                  */
#define TLC_UNREACHABLE_100  0x20u /* no connectivity assert; an unreachable root costs the sentinel 100:
                                      Knowledge_Distillation/data_utils_LP.py:41-49 (filtration only)  */
/* which of the three node values of filtration.build_fv (riccidist2dgm.py:47-49) is the filtration: 'sum' = d1 + d2 (the
 * pipeline's choice, loaddatas.py:101; flag value 0), 'min', 'max' (get_pimg_for_all_edges' own default is 'min', :362) --
 * 'min' and 'max' are both divided by max_S max(d1, d2), 'sum' by max_S (d1 + d2) (:51-56) -- or the distance to the FIRST
 * root alone, the single-root filtration of the node-centred PDGNN vicinities (Knowledge_Distillation/data_utils_NC.py:27-50;
 * hand the node in as the pair (u, u): ball(u) & ball(u) is its ball) */
#define TLC_DESC_MIN         0x40u
#define TLC_DESC_MAX         0x80u
#define TLC_DESC_ROOT1       0xC0u
#define TLC_DESC_MASK        0xC0u
/* norm=False of build_fv (:50): raw distances, no division, no ZeroDivisionError.  tlc_vicinity_filtration only: the image
 * stage of tlc_pd_pi_batch assumes values in [0, 1] (TLC_ERR_UNSUPPORTED there); unnormalised images are
 * tlc_vicinity_filtration -> tlc_pd_from_filtration -> tlc_pi_raster (what the sg2dgm_accelerate drop-in does) */
#define TLC_NO_NORM          0x100u

typedef struct tlc_graph tlc_graph;   /* opaque: device CSR + scratch, bound to one device */

/* ---- library ------------------------------------------------------------------------------------ */
const char* tlc_version(void);
const char* tlc_last_error(void);          /* thread-local text of the last TLC_ERR_* */
int  tlc_device_count(void);

/* ---- P1: graph2pi.__init__ (sg2dgm/riccidist2dgm.py:216-226) ------------------------------------
 * Symmetric CSR of the weighted graph: node ids 0..n_nodes-1, h_w[e] = kappa_e + 1 (must be > 0),
 * both directions present.  Nodes with an empty row are "missing" (TLC_ST_MISSING_NODE), exactly as
 * the reference's graph is built from edges only (loaddatas.py:88-92).  Synchronous (uploads). */
int tlc_graph_create(int32_t n_nodes, const int32_t* h_rowptr, const int32_t* h_col, const double* h_w,
                     int device, tlc_graph** out);
int tlc_graph_destroy(tlc_graph* g);

/* ---- P2-P9: graph2pi.get_pimg_for_all_edges (sg2dgm/riccidist2dgm.py:348-370) ------------------
 * For every pair (u,v): S = ball_hop(u) & ball_hop(v) on the unweighted graph (:311-316), induced
 * subgraph, filtration f = (d(x,u)+d(x,v)) / max with node-sourced weighted shortest paths (:20-61),
 * perturbed keys + two union-find passes + spanning-tree cycle swap (accelerated_PD.py:6-178),
 * res x res Gaussian persistence image (PersistenceImager.pyx:352-388).
 *   d_pairs      int32[n_pairs,2]
 *   d_out_pi     float64[n_pairs, res*res], row-major [birth_bin*res + pers_bin]
 *   d_out_status uint8[n_pairs] (may be NULL)
 * hop >= 1; res in 1..8. */
int tlc_pd_pi_batch(tlc_graph* g, const int32_t* d_pairs, int64_t n_pairs, int hop, uint32_t flags,
                    int res, double* d_out_pi, uint8_t* d_out_status, void* stream);

/* Same pipeline, stopping after the filtration (used by tests and by the PDGNN 'filtration' mode,
 * Knowledge_Distillation/data_utils_LP.py:105-200 mode='filtration'):
 *   d_node_offs  int64[n_pairs+1]  (caller-provided capacity layout: slot i holds up to
 *                                   node_offs[i+1]-node_offs[i] nodes)
 *   d_out_ids    int32[cap]  ascending node ids of S;  d_out_f float64[cap];  d_out_n int32[n_pairs]
 * A vicinity larger than its slot sets d_out_n[i] = -(size).
 * Optional (all three or none): d_edge_offs int64[n_pairs+1] capacity layout, d_out_edges int32[cap,2] = the induced
 * undirected edges as LOCAL ids (position in the pair's id list, lower id first), d_out_m int32[n_pairs] (-(m) if the slot is
 * too small): the (filtration_val, edge_index) pair that data_utils_LP.compute_persistence_image(mode='filtration') returns. */
int tlc_vicinity_filtration(tlc_graph* g, const int32_t* d_pairs, int64_t n_pairs, int hop, uint32_t flags,
                            const int64_t* d_node_offs, int32_t* d_out_ids, double* d_out_f,
                            int32_t* d_out_n, uint8_t* d_out_status,
                            const int64_t* d_edge_offs, int32_t* d_out_edges, int32_t* d_out_m, void* stream);

/* The same batch WITHOUT the final join -- for a caller that submits batch after batch (the reference's sweep over its pair
 * lists, sg2dgm/riccidist2dgm.py:362-370 over loaddatas.py:44-53) and wants them to overlap: the batch is ordered after what
 * `stream` holds at the time of the call (its inputs may be produced there), runs on streams of the handle, and `stream` does
 * NOT wait for it.  The outputs are complete for work that follows a tlc_pd_pi_batch_join() on its stream -- and only then:
 * the second half of the most recent batch (the launches sized by its vicinity counts) is submitted by the next
 * tlc_pd_pi_batch_async, by the join, or by any other call on the handle, whichever comes first.  A handle keeps three
 * batches in flight; submitting a fourth waits ON THE HOST for the first.  tlc_pd_pi_batch == async + join.
 * Buffers (pairs, out_pi, out_status) of a batch in flight must stay valid and must not be written until joined. */
int tlc_pd_pi_batch_async(tlc_graph* g, const int32_t* d_pairs, int64_t n_pairs, int hop, uint32_t flags,
                          int res, double* d_out_pi, uint8_t* d_out_status, void* stream);
int tlc_pd_pi_batch_join(tlc_graph* g, void* stream);

/* Counters of the last tlc_pd_pi_batch on this handle (synchronises the stream):
 * h_out[0..3] = pairs in tier small / medium / large / huge, [4] = induced directed entries (arena size),
 * [5] = sources that needed the exact tie fallback, [6] = chunks, [7] = pairs in tier mid (between small and medium),
 * [8] = pairs of tier small that took the lane-per-subgraph kernel (at most 16 nodes / 24 edges), [9] = pairs of tier medium with many Pos edges (launched first).  h_out: 10 entries. */
int tlc_pd_pi_batch_stats(tlc_graph* g, int64_t* h_out, void* stream);

/* Measurement helpers used by bench.py (no reference counterpart: the reference only prints time.time() deltas,
 * riccidist2dgm.py:349-350).  set_timing(1) makes every later tlc_pd_pi_batch bracket each of its kernels with HIP
 * events on the stream that kernel runs on (set_timing(1 << (k+1)), or a sum of such bits: only the kernels of slots k -- the
 * event records themselves cost time, 48 us of the 1.06 ms PubMed batch for all eight); timings() returns the last chunk's
 * durations in ms:
 * h_ms[0..7] = COUNT, scan+binning, FILL, PD tier SMALL, MEDIUM, LARGE, HUGE, MID (-1 = not launched).  Synchronises.
 * sizes(): per pair of the last chunk, |S| and the induced directed entry count.
 * algorithmic_bytes(): SURVEY.md 8(d) per-pair byte model, evaluated on the HOST CSR (pure accounting). */
int tlc_pd_pi_batch_set_timing(tlc_graph* g, int enable);
int tlc_pd_pi_batch_timings(tlc_graph* g, double* h_ms, void* stream);
/* The same for a caller that enqueues chunk after chunk WITHOUT synchronising (the shape of the reference's sweep over
 * total_edges, riccidist2dgm.py:362-370): the events of the last 64 chunks are kept; history() synchronises once, afterwards,
 * and returns timing slot `slot` (index into h_ms above) of the most recent min(cap, 64, chunks so far) chunks, oldest
 * first, -1 where the kernel did not run; *n_out = entries written. */
int tlc_pd_pi_batch_timing_history(tlc_graph* g, int slot, double* h_ms, int32_t cap, int32_t* n_out, void* stream);
int tlc_pd_pi_batch_sizes(tlc_graph* g, int32_t* h_n, int32_t* h_m2, int64_t cap, void* stream);
int tlc_pd_pi_algorithmic_bytes(int32_t n_nodes, const int32_t* h_rowptr, const int32_t* h_col, const int32_t* h_pairs,
                                int64_t n_pairs, int hop, int res, double* h_out_bytes);

/* Diagnostics (development tools and the tests of the divide-and-conquer cycle swap; no reference counterpart).
 * dc_stats(): of the chunks since the last tlc_pd_pi_batch call began, h_out[0] = subgraphs whose cycle swap ran as the divide
 * and conquer of tlc_pd_dc_kernel, h_out[1] = subgraphs it gave back to the serial walk (ranks that are no minimum-spanning-tree
 * order).  Synchronises the stream.
 * phase_profile(): per-phase cycle counters of the tier kernels in a library built with `make PHASE_DEBUG=1` (all zero
 * otherwise): rows of 32 u64, one per tier and one for the early pass; at most cap_u64 values are written to h_out (may be
 * null), *n_rows (may be null) = rows kept.  enable != 0 starts counting, 0 stops and frees the counters.
 * set_option(): the fourteen switches of one handle, each exercised by a test that checks that results do not depend on it
 * (tests/test_gpu_extract.py, tests/test_gpu_tiers.py, tests/test_gpu_pd_parity.py); 1 = on is the default of the first eight:
 *   "extract"      ball-list extraction of the vicinities (any hop since round 5; 0: the breadth-first kernels; TLC_EXTRACT=0 at creation)
 *   "heavy"        its hub-row skipping (TLC_HEAVY=0)
 *   "tiny"         lane-per-subgraph kernel for vicinities of at most 16 nodes / 24 edges (TLC_TINY=0)
 *   "ball_edges"   vicinities of pairs whose smaller ball has <= 128 nodes from that ball's subgraph list (TLC_BALL_EDGES=0)
 *   "fast_split"   ... in a launch of their own beside the classification and the early pass (TLC_FAST_SPLIT=0)
 *   "ball_bits"    ... which tests membership in the larger ball against per-node ball bitmaps (N^2 / 8 bytes, built up to 1 GiB)
 *                  instead of marking that ball in an LDS bitmap per pair (round 6; TLC_BALL_BITS=0)
 *   "plain_kernels" the tier / swap kernel instances that have a plain image batch's parameters (flags 0, resolution 5, no filtration
 *                  outputs) as compile-time constants (round 6; 0: the general instances; TLC_PLAIN_KERNELS=0)
 *   "dc_inplace"   the LARGE tier's divide and conquer by the tier kernel's own workgroup (0: tlc_pd_dc_kernel; TLC_DC_INPLACE=0)
 *   "dc_force_fail" every divide-and-conquer solve given back to the serial walk (test hook)
 *   "spec_cap"     slots reserved for the speculative tier launches (test hook: beyond them the second-launch / in-kernel paths)
 *   "x_arena"      arena entries per extraction workgroup and of the bump area (test hook: reach the overflow paths; 0 = defaults)
 *   "tier_mask"    bit t: tier t's kernels are launched at all (cost tables under profiles/)
 *   "n_ws"         workspaces taken in turn by pipelined chunks (2..4, default 3)
 *   "timing_every" kernel events on every n-th chunk only
 * An unknown name is TLC_ERR_INVALID_ARG. */
int tlc_debug_set_option(tlc_graph* g, const char* name, int value);
int tlc_debug_dc_stats(tlc_graph* g, long long* h_out, void* stream);
/* The tier lists of the last tlc_pd_pi_batch call as the device cut them, h_out[8]: small, medium (the compact kernel configuration,
 * <= 384 nodes / 512 edges), large, huge, mid, tiny, medium with many Pos edges, medium beyond the compact configuration (<= 512 /
 * 1024).  tlc_pd_pi_batch_stats reports tiny with small and the three medium lists as one. */
int tlc_debug_tier_counts(tlc_graph* g, long long* h_out, void* stream);
int tlc_debug_phase_profile(tlc_graph* g, int enable, unsigned long long* h_out, int64_t cap_u64, int32_t* n_rows);
/* (builds with PAIR_TIMES=1 only: wall-clock stamps of the extraction, h_out[n_pairs][16] ticks of 10 ns; zeros otherwise) */
int tlc_debug_pair_times(tlc_graph* g, unsigned long long* h_out, int64_t n_pairs);

/* ---- P6-P8: perturb_filter_function / Union_find / Accelerate_PD ---------------------------------
 * (sg2dgm/accelerated_PD.py:6-178 and the Knowledge_Distillation fork, selected by TLC_KEEP_ZERO_PERS)
 * Batch of n_graphs independent graphs with caller-supplied filtration values.
 *   d_node_offs int64[n_graphs+1], d_edge_offs int64[n_graphs+1]
 *   d_edges     int32[sum m, 2] local node ids;  d_f float64[sum n]
 * Outputs (slot of graph g starts at node_offs[g] resp. edge_offs[g]):
 *   d_pd_up    float64[sum n, 2]  Ord0 points   (ascending pass, :46-68)
 *   d_pd_down  float64[sum n, 2]  Rel1 points   (descending pass, :83-109)
 *   d_pd_one   float64[sum m, 2]  Ext1 points   (:115-178)
 *   d_ext0     float64[n_graphs, 2]  [min f, max f] (:110)
 *   d_counts   int32[n_graphs, 4] = {#up, #down, #one, #connected components}; all four -1 for a graph with more than
 *              65 535 nodes or 2^24-2 edges (not computed)
 *   d_edge_rank int32[sum m] (may be NULL): >= 0: position among the Pos edges, in descending-pass
 *               order (:109); < 0: -(position among the Neg edges)-1 (:99). */
int tlc_pd_from_filtration(int32_t n_graphs, const int64_t* d_node_offs, const int64_t* d_edge_offs,
                           const int32_t* d_edges, const double* d_f, uint32_t flags,
                           double* d_pd_up, double* d_pd_down, double* d_pd_one, double* d_ext0,
                           int32_t* d_counts, int32_t* d_edge_rank, void* stream);

/* ---- P9: PersistenceImager(resolution=res).transform (sg2dgm/PersistenceImager.pyx:352-388) -------
 * isotropic sigma=1 Gaussian, ranges [0,1]^2, linear-ramp weight (:9-30), birth-death input (skew=True).
 *   d_offs int64[n_dgms+1];  d_pts float64[sum k, 2];  d_out float64[n_dgms, res*res] */
int tlc_pi_raster(int32_t n_dgms, const int64_t* d_offs, const double* d_pts, int res, double* d_out,
                  void* stream);
/* Backward of the DIFFERENTIABLE imager, Knowledge_Distillation/pimg.py:354-400 (Teacher_Model.forward(grad_PI=True),
 * Teacher_model.py:80-81).  The reference computes a point's two normal-CDF factors from detached coordinates (:392,395), so a
 * point reaches the image through its weight alone, linear_ramp(death - birth) (:11-30, :371): d_grad_pts[i] = (-g, +g) with
 * g = sum_pixels d_grad_img[diagram of i] * (dPhi_b * dPhi_p of point i) for 0 <= death - birth <= 1, else 0.
 *   d_grad_img float64[n_dgms, res*res];  d_grad_pts float64[n_pts, 2] (every row written) */
int tlc_pi_raster_wgrad(int32_t n_dgms, int64_t n_pts, const int64_t* d_offs, const double* d_pts, int res,
                        const double* d_grad_img, double* d_grad_pts, void* stream);

/* ---- M1-M3: TLCGNN forward (baselines/TLCGNN.py:19-62) ------------------------------------------- */

/* gcn_norm of GCNConv(cached=True) (in-tree spec: Knowledge_Distillation/PD_conv.py:35-70): add the
 * remaining self loops (w=1), deg = scatter_add(w, target), norm = d^-1/2[src] * w * d^-1/2[dst];
 * result as CSR by target row, sources ascending inside a row.
 *   d_edge_index int64[2,n_edges] (row 0 = source, row 1 = target), existing self loops keep w=1 once
 *   d_rowptr int32[n_nodes+1]; d_col int32[n_edges+n_nodes]; d_val float32[n_edges+n_nodes]
 *   d_nnz int32[1]: entries actually written (<= n_edges+n_nodes) */
int tlc_gcn_norm_csr(int32_t n_nodes, int64_t n_edges, const int64_t* d_edge_index,
                     int32_t* d_rowptr, int32_t* d_col, float* d_val, int32_t* d_nnz, void* stream);

/* C[M,N] = A[M,K] @ B[K,N] (+bias[N]) (optional ReLU), fp32 in / fp32 accumulate on the f32 MFMA.
 * The dense projection x @ W of GCNConv (PD_conv.py:179-181). Row-major, leading dims = N/K. */
int tlc_gemm_f32(int32_t M, int32_t N, int32_t K, const float* d_A, const float* d_B, const float* d_bias,
                 int relu, float* d_C, void* stream);

/* The same product for a SPARSE A given as CSR (rowptr int32[M+1], col int32[nnz] < K, val f32[nnz]): the feature projection
 * x @ W of GCNConv (PD_conv.py:179-181) on bag-of-words / TF-IDF features (PubMed: 10 % non-zeros): the zeros contribute
 * nothing, a tenth of the flops.  B is staged in LDS in column slices of at most 64 (N = 100: 52 + 48): K * slice * 4 B + 1 KiB
 * <= 160 KiB, i.e. K <= 636 for a 64-column slice (TLC_ERR_UNSUPPORTED beyond).  Sums in a fixed order of its own (within 1e-5
 * relative of the dense product).  nnz < 2^29 (byte offsets of the entries are 32-bit). */
int tlc_spgemm_csr_dense_f32(int32_t M, int32_t K, int32_t N, const int32_t* d_rowptr, const int32_t* d_col, const float* d_val,
                             const float* d_B, const float* d_bias, int relu, float* d_C, void* stream);

/* Y[n,k] = act( CSR(rowptr,col,val) @ X[n,k] + bias[k] ): the propagate/scatter-add of GCNConv
 * (PD_conv.py:183-188; message_passing.py:275-293 aggr='add').  relu: bit 0 = ReLU; bit 1 = afterwards renormalise
 * every row like tlc_renorm_rows_f32 (the emb.renorm_ of TLCGNN.py:48 fused into the last layer's aggregation). */
int tlc_spmm_csr_f32(int32_t n_rows, const int32_t* d_rowptr, const int32_t* d_col, const float* d_val,
                     const float* d_X, int32_t k, const float* d_bias, int relu, float* d_Y, void* stream);

/* emb.renorm_(2, 0, 1) (TLCGNN.py:48): rows with L2 norm > 1 are scaled by 1/(norm + 1e-7), in place. */
int tlc_renorm_rows_f32(int32_t n_rows, int32_t k, float* d_emb, void* stream);

/* Net.encode in eval mode (baselines/TLCGNN.py:19-26: conv1 -> ReLU -> conv2 on the normalised adjacency of tlc_gcn_norm_csr)
 * as ONE call: tlc_gemm_f32, tlc_spmm_csr_f32 (+ b1, ReLU), tlc_gemm_f32, tlc_spmm_csr_f32 (+ b2; `flags` bit 0 = ReLU, bit 1 =
 * the emb.renorm_(2, 0, 1) of TLCGNN.py:48) submitted back to back.  d_ws: scratch of (2 * hidden + out_dim) * n_nodes + 12
 * floats; d_emb: [n_nodes, out_dim].  hidden, out_dim <= 128. */
int tlc_gcn2_encode_f32(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, const float* d_val,
                        const float* d_x, int32_t f_in, const float* d_w1, const float* d_b1, int32_t hidden,
                        const float* d_w2, const float* d_b2, int32_t out_dim, int flags, float* d_ws, float* d_emb, void* stream);

/* The same with the node features given as CSR (d_xs_rowptr int32[n_nodes+1], d_xs_col int32[nnz] < f_in, d_xs_val f32[nnz]): the
 * first projection is tlc_spgemm_csr_dense_f32 (its limit on f_in applies).  What Net.encode runs on PubMed's TF-IDF features. */
int tlc_gcn2_encode_csr_f32(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, const float* d_val,
                            const int32_t* d_xs_rowptr, const int32_t* d_xs_col, const float* d_xs_val, int32_t f_in,
                            const float* d_w1, const float* d_b1, int32_t hidden, const float* d_w2, const float* d_b2,
                            int32_t out_dim, int flags, float* d_ws, float* d_emb, void* stream);

/* Net.decode after the renorm (TLCGNN.py:52-61), one fused pass per pair:
 *   h = LeakyReLU_0.2( W1 @ [ (emb[u]-emb[v])^2 || PI ] + b1 );  d = clamp(|W2 @ h + b2|, 0, 40);
 *   prob = 1 / (exp(d - 2) + 1)
 *   d_pairs int32[n_pairs,2]; d_emb float32[n_nodes,emb_dim]; d_pi float64[n_pairs,pi_dim] (cast to
 *   float32 on load, as torch.Tensor(PI) does); d_W1 float32[pi_dim, emb_dim+pi_dim] (torch Linear
 *   layout [out,in]); d_b1[pi_dim]; d_W2 float32[pi_dim]; d_b2 float32[1]; d_prob float32[n_pairs] */
int tlc_lp_decode_fused(int64_t n_pairs, const int32_t* d_pairs, const float* d_emb, int32_t emb_dim,
                        const double* d_pi, int32_t pi_dim, const float* d_W1, const float* d_b1,
                        const float* d_W2, const float* d_b2, float* d_prob, void* stream);
/* The same pass over a float32 image table [n_pairs, pi_dim]: the reference casts the images to float32 before the layer on
 * every decode (`PI = torch.Tensor(self.PI[...])`, TLCGNN.py:35-36,52-53); a caller that casts its table ONCE at set-up hands
 * it in here -- same values (the cast is the same rounding), 100 instead of 200 bytes per pair.  Both forms run the hidden
 * layer on the f32 MFMA when emb_dim == 16 and pi_dim == 25 (one lane per pair, K = 1 MFMAs in the reference's summation order,
 * weights in registers, no LDS). */
int tlc_lp_decode_fused_f32(int64_t n_pairs, const int32_t* d_pairs, const float* d_emb, int32_t emb_dim,
                            const float* d_pi, int32_t pi_dim, const float* d_W1, const float* d_b1,
                            const float* d_W2, const float* d_b2, float* d_prob, void* stream);

/* ---- M4-M6: PDGNN layer forward (Knowledge_Distillation/gat_conv.py:113-216) ----------------------
 * One GATConv(heads=1, new_node_feat, use_edge_attn) layer on a block-diagonal batch of graphs whose
 * edges are given as CSR BY TARGET (self loops already added, gat_conv.py:146-160):
 *   x_l = X @ Wl^T (no bias);  alpha = x_l . att;  per edge j->i: a = softmax_i(leaky_relu(alpha_j+alpha_i)),
 *   m = leaky_relu(Wij @ [x_i || x_j]) * a;  out_i = [ sum m || (min m + max m) ] + bias  (:166-172,202-216)
 *   d_X float32[n, c_in]; d_Wl float32[c_out, c_in]; d_att float32[c_out]; d_Wij float32[c_out, 2*c_out];
 *   d_bias float32[2*c_out]; d_out float32[n, 2*c_out]; prelu_slope < 0 disables the fused PReLU.
 *   d_work float32[n * (3*c_out + 4) + c_in*c_out + c_out*(2*c_out + 4)]: caller-provided scratch (per node: x_l, the two
 *   lin_ij half-projections, alpha; then the layer's weights packed for the MFMA GEMM).
 *   c_out in {8,16,32,64}. */
int tlc_gat_layer_fwd(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_src,
                      const float* d_X, int32_t c_in, int32_t c_out,
                      const float* d_Wl, const float* d_att, const float* d_Wij, const float* d_bias,
                      float prelu_slope, float* d_work, float* d_out, void* stream);
/* The same layer on a BLOCK-DIAGONAL batch cut into self-contained tiles (round 5; gat_forward.hip, gat_tile_kernel): d_tile_ptr
 * int32[n_tiles + 1] = node offsets of tiles of at most 192 consecutive nodes such that every in-edge of a node has its source in the
 * node's own tile (a batch of small graphs cut at positions no edge crosses: Knowledge_Distillation/gat_conv.py, GraphBatch).  The node
 * rows [P | Q | alpha] are computed on the f32 MFMA into LDS and aggregated from there: they never reach HBM.  c_in = 1 or 64 with
 * c_out = 32 or 16 (the PDGNN layers of Teacher_model.py:182-189); other shapes: TLC_ERR_UNSUPPORTED (take tlc_gat_layer_fwd).
 * d_work: float32[c_in*c_out + c_out*(2*c_out + 4) + c_in*(2*c_out + 4) + 2*c_out + 8].  Results equal tlc_gat_layer_fwd's up to the
 * rounding of fp32 sums. */
int tlc_gat_layer_tiled_fwd(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_src, int32_t n_tiles,
                            const int32_t* d_tile_ptr, const float* d_X, int32_t c_in, int32_t c_out,
                            const float* d_Wl, const float* d_att, const float* d_Wij, const float* d_bias,
                            float prelu_slope, float* d_work, float* d_out, void* stream);

/* The tile cut tlc_gat_layer_tiled_fwd takes (Knowledge_Distillation/gat_conv.py GraphBatch -> ops.gat_tiles), made on the device:
 * d_tile_ptr int32[*n_tiles + 1] = node offsets of tiles of at most tile_nodes consecutive nodes, cut only at positions no edge of
 * the CSR (rows = targets, d_col = sources) crosses; with `gap` the largest distance between two such positions next to each other,
 * the first of them at or behind every multiple of tile_nodes - gap starts a tile.  *n_tiles (HOST int; the call waits for the
 * stream) = 0 when 2 gap > tile_nodes -- one big graph: the two-kernel layer (tlc_gat_layer_fwd) serves it.
 * d_work: int32[n_nodes / 32 + 6]; d_tile_ptr: room for 2 n_nodes / tile_nodes + 3 entries. */
int tlc_gat_tile_cut(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, int32_t tile_nodes, int32_t* d_work,
                     int32_t* d_tile_ptr, int32_t* n_tiles, void* stream);


/* remove_self_loops + add_self_loops + grouping by target (gat_conv.py:146-152), the structure alone and per BATCH: what
 * tlc_gcn_norm_csr builds without its values, with the caller's temporaries -- nothing is allocated, nothing waits for the stream.
 *   d_rowptr int32[n_nodes+1]; d_col int32[n_edges+n_nodes] (sources ascending inside a row); d_nnz int32[1];
 *   d_work int32[3 n_nodes + n_edges]. */
int tlc_csr_by_target(int32_t n_nodes, int64_t n_edges, const int64_t* d_edge_index, int32_t* d_rowptr, int32_t* d_col,
                      int32_t* d_nnz, int32_t* d_work, void* stream);

/* Teacher_Model.forward(compute_loss=False) without gradients (Knowledge_Distillation/Teacher_model.py:49-88) as ONE call: CSR by
 * target + tile cut of the batch (skipped when the caller hands in d_rowptr / d_col / d_tile_ptr of a batch it holds; d_tile_ptr NULL
 * with n_tiles 0: the two-kernel layers), conv1 -> conv2 -> conv4 -> conv3 with PReLU(0.1) behind the first three
 * (Base_Model.forward :218-227), the edge head (:54-59) and one res x res image per graph (:84).  Same kernels, same results as the
 * entry points above called one by one; the steps are submitted from native code out of one workspace.
 *   d_edge_index int64[2][n_edges], the n_nodes self loops LAST (train_Teacher_Model.py:43-44); d_x float32[n_nodes] (in_dim 1);
 *   hidden = 32; params: HOST array of 20 device pointers (float32) = for conv1, conv2, conv4, conv3 in this order
 *   {lin_l.weight, att_l, lin_ij.weight, bias}, then lin5.weight, lin5.bias, lin6.weight, lin6.bias;
 *   d_edge_ptr int64[n_graphs+1]: offsets into the n_edges - n_nodes edges; d_points float32[n_edges - n_nodes][2] (the predicted
 *   diagram points); d_img float64[n_graphs][res*res]; d_work: tlc_pdgnn_forward_work_bytes(n_nodes, n_edges, hidden) bytes.
 * The call waits for the stream once (the tile count) unless the structure is handed in.
 * Both entry points take n_nodes >= 1, n_edges >= n_nodes, hidden > 0 (work_bytes returns -1 otherwise, the forward TLC_ERR_INVALID_ARG). */
int64_t tlc_pdgnn_forward_work_bytes(int32_t n_nodes, int64_t n_edges, int32_t hidden);
int tlc_pdgnn_forward(int32_t n_nodes, int64_t n_edges, const int64_t* d_edge_index, const float* d_x, int32_t hidden,
                      const float* const* params, int64_t n_graphs, const int64_t* d_edge_ptr, int32_t res,
                      const int32_t* d_rowptr, const int32_t* d_col, const int32_t* d_tile_ptr, int32_t n_tiles,
                      void* d_work, int64_t work_bytes, float* d_points, double* d_img, void* stream);


/* ---- the diagram loss of PDGNN training ------------------------------------------------------------
 * `wasserstein_distance(X, Y, order=p, internal_p=inf, enable_autodiff=True, num_models=1)` of
 * Knowledge_Distillation/wasserstein.py:198-379, as called by Teacher_model.py:131 (compute_PD_loss, kernel='wasserstein'),
 * for a batch of diagram pairs: problem b has predicted points X[xoff[b] .. xoff[b+1]) and target points Y[yoff[b] .. yoff[b+1]),
 * (birth, death) float64 pairs.  Every predicted point goes to a target point (each target takes exactly one) or to the diagonal;
 * cost = ||X_i - Y_j||_inf ^ p, resp. ((death - birth) / 2) ^ p (:45-67); the optimum is found by the Hungarian method on the
 * device, one wavefront per problem (the reference calls POT's ot.emd: third-party, parity unpinned -- the optimal cost is
 * unique, the choice among tied optima is not reproduced).
 *   d_loss[b] = (sum over the matched distances d_k of |d_k|^p)^(1/p)   (:303-372; order p = 1 or 2)
 *   d_wxy / d_wxd[b] = the same norm over the point-point pairs / the points sent to the diagonal (what the reference logs)
 *   d_assign[i] = problem-local target index of predicted point i, -1 = diagonal
 *   d_gradX[i] (may be null) = d loss[b] / d X_i: what `loss.backward()` leaves on the predicted diagram
 *   d_status[b]: 0 ok; 1 = fewer predicted than target points (the reference's transport has negative diagonal mass: no
 *   result, loss 0); 2 = more than 4 096 predicted points (not supported: loss 0); 3 = a NaN / Inf coordinate (loss NaN, zero
 *   gradient: the augmenting search cannot run on it).  Problems of up to 512 predicted points take one
 *   wavefront each, larger ones a 512-thread workgroup each (a second launch when max_points > 512).  max_points: an upper bound of the predicted
 *   points of one problem (selects the kernel variant). */
int tlc_w2_partial_matching(int32_t n_problems, const int64_t* d_xoff, const double* d_X, const int64_t* d_yoff,
                            const double* d_Y, int order, int32_t max_points, double* d_loss, double* d_wxy, double* d_wxd,
                            int32_t* d_assign, double* d_gradX, uint8_t* d_status, void* stream);

/* The evaluation form: `wasserstein_distance_inference(X, Y, order=p, internal_p=inf, enable_autodiff=True)` of
 * Knowledge_Distillation/wasserstein.py:93-195, reached with `pair_diagonal=True` (train_Teacher_Model.py:99 ->
 * Teacher_model.py:66 -> compute_PD_loss(type='inference') :134-136).  The classic transport in which BOTH diagrams may use the
 * diagonal (masses [1]*n + [m] against [1]*m + [n], (n+1) x (m+1) costs, C[n, j] = ((Y_j.death - Y_j.birth) / 2) ^ p,
 * C[n, m] = 0 :127-131) = the assignment of n + m rows onto n + m columns; same kernels as tlc_w2_partial_matching.
 *   d_loss[b] = (sum |d_k|^p)^(1/p) over point-point pairs, predicted points sent to the diagonal and target points sent to the
 *   diagonal (:140-181); d_wxy / d_wxd / d_wyd[b]: the same norm over each of the three groups (the reference's five return
 *   values are (loss, None, wxy, wxd, wyd)).  An empty diagram on either side: loss = total persistence of the other one and
 *   the three parts 0 (:98-113).
 *   d_assign_x[i] = target index of predicted point i or -1 = diagonal; d_assign_y[j] = predicted index of target j or -1.
 *   d_gradX (may be null): d loss[b] / d X_i.  d_status[b]: 0 ok; 2 = n + m > 4 096; 3 = non-finite coordinates (loss NaN).
 *   max_points: an upper bound of n + m of one problem.  Parity unpinned like tlc_w2_partial_matching (POT absent). */
int tlc_w2_inference_matching(int32_t n_problems, const int64_t* d_xoff, const double* d_X, const int64_t* d_yoff,
                              const double* d_Y, int order, int32_t max_points, double* d_loss, double* d_wxy, double* d_wxd,
                              double* d_wyd, int32_t* d_assign_x, int32_t* d_assign_y, double* d_gradX, uint8_t* d_status,
                              void* stream);

/* ---- what `loss.backward()` runs through the PDGNN layer and the edge head ---------------------------
 * (Knowledge_Distillation/train_Teacher_Model.py:55-62 through gat_conv.py:113-216 and Teacher_model.py:53-59.)
 * tlc_gat_layer_bwd: gradients of one layer, same operands as tlc_gat_layer_fwd.  Nothing of the forward is kept: the node rows,
 * the row softmax and the channel-wise minima / maxima are recomputed (one wavefront per target row).  A tied minimum / maximum
 * sends its gradient to the first edge of the row that attains it.
 *   d_out  float32[n, 2*c_out]: the forward's output, read only when prelu_slope >= 0 (sign of the pre-activation); else may be NULL
 *   d_gout float32[n, 2*c_out]: d loss / d out
 *   d_gX   float32[n, c_in] or NULL (first layer);  d_gWl [c_out, c_in], d_gatt [c_out], d_gWij [c_out, 2*c_out], d_gbias [2*c_out]:
 *   OVERWRITTEN.   d_work float32[n * (8*c_out + 5) + 2*c_out*c_out + c_in*c_out + 4 + c_out*(2*c_out + 4)] scratch.   c_out in {8,16,32,64}, c_in <= 64.
 * Float atomics towards the sources of the edges and in the weight reductions: the summation order is not fixed. */
int tlc_gat_layer_bwd(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_src, const float* d_X, int32_t c_in,
                      int32_t c_out, const float* d_Wl, const float* d_att, const float* d_Wij, float prelu_slope,
                      const float* d_out, const float* d_gout, float* d_gX, float* d_gWl, float* d_gatt, float* d_gWij,
                      float* d_gbias, float* d_work, void* stream);

/* tlc_edge_head_bwd: gradients of tlc_edge_head_fwd.  d_gpd float32[n_edges, 2] = d loss / d pd.
 *   d_gX float32[n_nodes, c] is ADDED to (the caller zeroes it); d_gW5 [hidden, 2c], d_gb5 [hidden], d_gW6 [2, hidden], d_gb6 [2]:
 *   OVERWRITTEN.   d_work float32[n_edges * (2*c + 2*hidden)] scratch.   hidden in {16,32,64}. */
int tlc_edge_head_bwd(int64_t n_edges, const int32_t* d_src, const int32_t* d_dst, const float* d_X, int32_t c,
                      const float* d_W5, const float* d_b5, int32_t hidden, float prelu_slope, const float* d_W6,
                      const float* d_gpd, float* d_gX, float* d_gW5, float* d_gb5, float* d_gW6, float* d_gb6,
                      float* d_work, void* stream);

/* MessagePassing.aggregate (Knowledge_Distillation/message_passing.py:275-293): torch_scatter.scatter(inputs, index, dim=0,
 * dim_size=n_out, reduce) with reduce 0 = sum, 1 = mean, 2 = min, 3 = max; empty segments give 0, as torch_scatter does.
 *   d_index int64[n_src]; d_src float32[n_src,k]; d_out float32[n_out,k]; d_count_work int32[n_out] (not needed for sum).
 * Float atomics: the summation order is not fixed (same as the reference's scatter kernels). */
int tlc_scatter_f32(int64_t n_src, const int64_t* d_index, const float* d_src, int32_t k, int reduce, int32_t n_out,
                    float* d_out, int32_t* d_count_work, void* stream);

/* Edge head of Teacher_Model.forward (Teacher_model.py:54-59): for every non-self-loop edge e=(s,t):
 *   pd[e] = W6 @ prelu(W5 @ [x[s] || x[t]] + b5) + b6   -> float32[n_edges,2]
 * d_work (optional, float32[(n_nodes + c) * 2 * hidden]): with it the first layer is computed per node on the MFMA GEMM
 * (W5[:, :c] x_s + W5[:, c:] x_t) and only gathered per edge; NULL: one 2c x hidden product per edge. */
int tlc_edge_head_fwd(int64_t n_edges, const int32_t* d_src, const int32_t* d_dst, const float* d_X,
                      int32_t c, const float* d_W5, const float* d_b5, int32_t hidden, float prelu_slope,
                      const float* d_W6, const float* d_b6, float* d_pd, int32_t n_nodes, float* d_work, void* stream);

/* ---- the callers' side of the path: negative enumeration and the image cache ----------
 *
 * loaddatas.py:44-45 lists the non-edges as `sp.triu(sp.csr_matrix(1. - adj.toarray())).nonzero()` (a dense N x N float64
 * matrix), shuffles the [n_neg, 2] array and slices it (:46,:51-53).  Pair number r of that list -- row-major over x <= y with
 * adj[x,y] == 0, the diagonal included -- is a function of the CSR alone:
 *
 * tlc_complement_rows: d_row_start int64[n_nodes+1] = exclusive prefix of the per-row non-edge counts (d_row_start[n_nodes] =
 *   n_neg).  The CSR must be the symmetric adjacency with columns ascending and unique inside a row (a stored entry is an
 *   edge, whatever its weight -- the reference's matrices are 0/1).
 * tlc_complement_pairs: d_pairs int32[count,2] = the pairs with list numbers d_ranks[0..count) (int64; e.g. a slice of the
 *   shuffled index list), or first, first+1, ... when d_ranks is NULL.  A number outside [0, n_neg) yields (-1, -1). */
int tlc_complement_rows(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, int64_t* d_row_start, void* stream);
int tlc_complement_pairs(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, const int64_t* d_row_start,
                         const int64_t* d_ranks, int64_t first, int64_t count, int32_t* d_pairs, void* stream);

/* The distance <= hop pre-filter of the sweep (PI-C): an image row can be non-zero only when d(u,v) <= hop
 * (SURVEY.md A.6, Z0), so only those non-edges need tlc_pd_pi_batch.  Appends every non-adjacent pair u <= v with d(u,v) <= hop
 * (the diagonal included unless u has a self loop): d_out_pairs int32[k,2] and d_out_rank int64[k] = the pair's number in the
 * list of tlc_complement_pairs.  *d_count (uint64, device, zeroed by the caller) is advanced even beyond `cap` (pairs past the
 * capacity are not written).  Append order is not fixed.  Graphs up to ~400 000 nodes (three LDS bitmaps per wavefront);
 * beyond that TLC_ERR_UNSUPPORTED (the full sweep does not have the limit). */
int tlc_near_pairs(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, const int64_t* d_row_start, int hop, int64_t cap,
                   uint64_t* d_count, int64_t* d_out_rank, int32_t* d_out_pairs, void* stream);

/* The reference caches the dense float64[n_pairs, res^2] image array (loaddatas.py:62-64,102: 39 GB for PubMed's sweep) although
 * every pair with d(u,v) > hop has a zero row, and keeps of the exceptions it swallows only their number (`cnt_compute`,
 * riccidist2dgm.py:355).  tlc_select_rows appends the rows of one image block that have a non-zero entry -- with
 * TLC_SELECT_KEEP_FAILED also the zero rows whose status is not TLC_ST_OK -- to a sparse store: d_out_idx[k] = index_base + row
 * number, d_out_status[k] (may be NULL), d_out_rows[k, width]; and adds the block's status bytes to the histogram
 * d_status_hist uint64[8] (may be NULL; the caller zeroes it once per store).  *d_count (uint64, device; the caller zeroes it
 * per call) is advanced by the number of kept rows even beyond `cap` (rows past the capacity are not written: re-run the block
 * with a larger store, and a fresh histogram).  Append order is not fixed. */
/* |S| and the number of induced edges of every pair's vicinity, nothing else (the extraction of tlc_vicinity_filtration without the
 * filtration): d_n / d_m int32[n_pairs], n = 0 for a pair without a vicinity; same hop / flags as the tlc_vicinity_filtration call
 * that follows.  With tlc_pack_offsets a caller gets exact offsets from them: no per-pair capacity to guess (replaces the sizing
 * the reference does implicitly by building Python lists, data_utils_LP.py:107-125). */
int tlc_vicinity_sizes(tlc_graph* g, const int32_t* d_pairs, int64_t n_pairs, int hop, uint32_t flags, int32_t* d_n, int32_t* d_m,
                       void* stream);
/* The offsets of that packed batch from the per-pair counts of tlc_vicinity_filtration, one launch: d_node_ptr / d_edge_ptr int64[n_pairs+1]
 * = exclusive prefix sums of (m > 0 ? n : 0) and max(m, 0) -- a vicinity without an edge is left out, the reference returns (None, None)
 * for it (data_utils_LP.py:117-118) -- and d_totals int64[4] = {min n, min m, sum n, sum m} (a negative minimum: some vicinity did not
 * fit the capacity it was given). */
int tlc_pack_offsets(int64_t n_pairs, const int32_t* d_n, const int32_t* d_m, int64_t* d_node_ptr, int64_t* d_edge_ptr,
                     int64_t* d_totals, void* stream);
/* The caller side of the PDGNN fork's vicinity extraction (Knowledge_Distillation/data_utils_LP.py:105-200 returns one
 * (filtration values, edge_index) per candidate edge; gcn_LP_GIN.Net.compute_PI :43-64 feeds them to the model one by one): the
 * per-pair capacity slots tlc_vicinity_filtration wrote -> ONE packed block-diagonal batch.  d_node_ptr / d_edge_ptr int64[n+1]:
 * the packed offsets (exclusive prefix sums of the node and edge counts the caller wants kept -- a pair's slice may be empty);
 * d_out_ids int64 = d_label[id] (NULL: the id itself), d_out_f, d_out_edges int32[.,2] (local ids, as written), and the owner
 * pair of every packed node / edge (either may be NULL).  One wavefront per pair, coalesced both ways. */
int tlc_pack_vicinities(int64_t n_pairs, const int64_t* d_node_offs, const int32_t* d_ids, const double* d_f,
                        const int64_t* d_edge_offs, const int32_t* d_edges, const int64_t* d_node_ptr, const int64_t* d_edge_ptr,
                        const int64_t* d_label, int64_t* d_out_ids, double* d_out_f, int32_t* d_out_edges,
                        int64_t* d_pair_of_node, int64_t* d_pair_of_edge, void* stream);

/* The packed batch -> the operands Teacher_Model.forward takes, one launch: what gcn_LP_GIN.py:43-64 builds per vicinity (its
 * edge_index plus self loops, the filtration as a float32 column), for the block-diagonal batch as a whole.  d_edge_index
 * int64[2][tot_m + tot_n]: global ids = local id + d_node_ptr[owner], the tot_n self loops LAST (train_Teacher_Model.py:43-44);
 * d_x float32[tot_n] = (float)d_f (both may be NULL).  d_node_ptr / d_edge_ptr int64[n_graphs + 1] with totals tot_n / tot_m. */
int tlc_stack_batch(int64_t n_graphs, const int64_t* d_node_ptr, const int64_t* d_edge_ptr, const int32_t* d_edges, const double* d_f,
                    int64_t tot_n, int64_t tot_m, int64_t* d_edge_index, float* d_x, void* stream);

#define TLC_SELECT_KEEP_FAILED 0x1u
int tlc_select_rows(int64_t n_rows, int32_t width, const double* d_pi, const uint8_t* d_status, int64_t index_base, int64_t cap,
                    uint32_t flags, uint64_t* d_count, uint64_t* d_status_hist, int64_t* d_out_idx, uint8_t* d_out_status,
                    double* d_out_rows, void* stream);

/* ---- the producer of the path's edge weights -------------------------------------------------------
 * compute_ricci_curvature (loaddatas.py:105-123) = third-party GraphRicciCurvature `OllivierRicci(G, alpha=0.5,
 * method="Sinkhorn")` (not in the reference tree, version not pinned; restated from the published algorithm -- parity unpinned):
 * per edge (s,t), m_s = alpha at s + (1-alpha)/deg on the neighbours, cost = hop distance, W = <P, d> of POT's
 * `sinkhorn2(x, y, d, reg)` (sinkhorn_knopp: stop when the marginal violation <= stop_thr, tested every 10th iteration, or after
 * max_iter iterations), kappa = 1 - W.  The library's call is (alpha 0.5, reg 0.1, max_iter 1000, stop_thr 1e-9).
 *   CSR: symmetric, no self loops, columns ascending and unique inside a row, unit weights.  d_edges int32[n_edges,2]: adjacent
 *   pairs (a self pair gets curvature 0).  d_kappa double[n_edges]; d_iters int32[n_edges] (may be NULL): iterations used.
 *   d_work/work_bytes: >= 16 + 4*n_edges (rounded up to 16) + k * max_product/4 bytes (rounded up to 16), k >= 1 slots for the
 *   hub edges (2-bit hop codes of supports beyond the LDS); max_support >= max over edges of deg(s)+deg(t)+2 and max_product >= max of (deg(s)+1)*(deg(t)+1) over the edges
 *   with more than 8 192 entries or deg(s)+deg(t)+2 > 256 (those the one-wavefront kernel leaves to the workgroup kernel).  An edge that exceeds them gets NaN (and iters -1), never a silent value. */
int tlc_ollivier_ricci_sinkhorn(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, int64_t n_edges,
                                const int32_t* d_edges, double alpha, double reg, int32_t max_iter, double stop_thr,
                                double* d_kappa, int32_t* d_iters, void* d_work, int64_t work_bytes, int32_t max_support,
                                int64_t max_product, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TLCGNN_H */
