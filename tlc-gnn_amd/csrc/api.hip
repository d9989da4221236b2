// api.hip -- host side of the C ABI (include/tlcgnn.h): handle, workspaces, launch orchestration.
//
// tlc_pd_pi_batch on a chunk of pairs:
//   COUNT (vicinity sizes) -> exclusive scan + tier binning -> [one 48-byte read-back: arena size and tier
//   counts] -> FILL (induced subgraphs into the arena) -> one PD kernel per size tier, the tiers running
//   concurrently on side streams so that the few large subgraphs overlap the many small ones.
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <new>
#include <vector>

#include "tlc_common.h"
#include <atomic>
#include <chrono>
#include "tlc_kernels.h"

// ---- error text ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void tlc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* tlc_last_error(void) { return g_err; }
extern "C" const char* tlc_version(void) { return "tlcgnn-hip 0.1.0 (gfx950)"; }
extern "C" int tlc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

#define TLC_CHUNK_PAIRS (1 << 20)
#define TLC_N_SIDE 8
// early pass (run_chunk): at most this many predicted-heavy pairs are counted ahead of the batch, at most this many LARGE-tier
// vicinities among them get a slot of the early arena (12 B x 2*TLC_L_MMAX entries each)
#define TLC_EARLY_CAND 512
#define TLC_EARLY_SLOTS 256
#define TLC_EARLY_WG 256          /* workgroups (and scratch slots) of the early COUNT */
#ifndef TLC_EARLY_MIN_PAIRS
#define TLC_EARLY_MIN_PAIRS 4096  /* smaller batches gain nothing from a second COUNT launch */
#endif
#define TLC_TIMING_RING 64        /* chunks whose kernel events are kept */
#define TLC_X_REGION 4096         /* arena entries of the region each workgroup of the extraction starts with (extract.hip) */

// a workspace's control block: [0, 64) control words | [64, 64 + TLC_SCAN_MAX_BLOCKS) the scan's per-block flags | 8 ints of statistics | 8 work counters of the
// general extraction launch, 64 ints apart
#define TLC_CTL_INTS (64 + TLC_SCAN_MAX_BLOCKS + 8 + 8 * 64)
struct HostSync {
    long long total_entries;
    int tier_count[TLC_N_TIERS];
    int pad[2];
    unsigned long long stats[2];
    // written by tlc_scan_bin straight into this (pinned, device-mapped) block; seq last, after a system-scope fence
    volatile long long pub_total;
    volatile int pub_tier[TLC_N_TIERS];
    volatile int pub_tiny[TLC_TINY_BINS];     // the TINY list's size classes
    volatile int pub_dcm;                     // MEDHI / MEDWIDE vicinities with Pos edges enough for the divide and conquer
    volatile int pub_early;
    volatile int pub_overflow;
    volatile unsigned pub_seq;
};

#define TLC_N_WS 4                /* workspaces of a handle: chunks (and asynchronous batches) take them in turn */

// Everything one chunk of pairs writes while it is in flight.  A handle has TLC_N_WS of them, taken in turn, each with its own
// streams: the chunks of one call, and the batches of tlc_pd_pi_batch_async, overlap -- the lead-in of one (selection, early
// extraction, the main extraction, all latency-bound) runs under the tail of the tier kernels of the one before.
// What the second half of a chunk (everything behind the size publication: run_chunk_back) needs from the first.  A pipelined
// chunk's second half is submitted one call later (see run_batch), so this lives in the workspace.
struct ChunkCtx {
    TlcVicParams vp;
    TlcPdParams pp;
    hipStream_t s;
    int n_pairs, hop, pi_enabled;
    bool bump, use_x, early, spec, count_only, tiny_bins, pipelined;
    bool mh_front;                 // the scan put the compact list's many-Pos vicinities on the MEDHI list: the front part of the MEDIUM launch
    bool plain, fsplit;            // a plain image batch (no filtration outputs); the subgraph-list pairs have a launch of their own
    int xfgrid;                    // ... of this many workgroups
    long long bump_base;
    int xgrid, vgrid, tmask;
    unsigned seq;
    size_t spec_base[TLC_N_TIERS];
    int spec_cap[TLC_N_TIERS];
    bool used[TLC_N_SIDE];
    hipEvent_t* ev_t;              // the chunk's set of timing events (tmask != 0)
    unsigned char* ev_used;
    std::chrono::steady_clock::time_point ht0;
    double ht_front;
    unsigned long long call_seq;   // the call the chunk belongs to (its statistics count only while that call is the last one)
};

struct Workspace {
    size_t cap_pairs;
    int *hdr_n, *hdr_m2, *hdr_lu, *hdr_lv, *tier_list;
    int* dc_lists;             // [3][cap_pairs + TLC_EARLY_SLOTS]: list positions a tier kernel hands to tlc_pd_dc_kernel (MEDIUM / LARGE / early LARGE)
    int* tiny_bins;            // [TLC_TINY_BINS][cap_pairs]: the TINY list by size class (tlc_scan_bin)
    int* big_lists;            // [4][cap_pairs]: the bins of tlc_classify_kernel (extract.hip; three in use)
    hipEvent_t ev_cls;         // the classification is done (recorded on the early stream)
    int prev_dcm;              // the previous chunk's count of MEDHI vicinities for the divide and conquer (decides the speculative chain)
    long long* edge_off;
    // small device block: [0..6] tier counts, [10..13] scan, [16..19] early pass, [20..22] bump allocator, [24] work counter,
    // [26..31] divide-and-conquer lists, [32..35] bins, [36..37] entry sum
    int* d_ctl;
    long long* d_block_sums;   // 1024
    long long* d_totals;       // 1
    unsigned long long* d_stats;  // [0] tie-fallback sources, [1] divide-and-conquer swaps, [2] (as int) LARGE workgroups started, [3] swaps given back
    HostSync* h_sync;          // pinned
    HostSync* h_sync_dev;      // the same block as the device sees it
    unsigned pub_seq;          // sequence number of the last size publication
    // arena
    size_t cap_entries;
    unsigned* A_dir;
    double* A_lw;
    // fixed-size slots of the SMALL tier (breadth-first COUNT pass only)
    size_t cap_small;
    unsigned* S_dir;
    double* S_lw;
    // vicinity scratch
    int vic_hop_cap;           // frontiers allocated for hop >= 3 ?
    int* vic_scratch;
    long long vic_stride;
    // HUGE tier scratch
    int huge_slots;
    unsigned char* huge_scratch;
    size_t huge_stride;
    // early pass: candidate / early lists, the early arena
    int* d_cand_list;
    int* d_early_list;
    unsigned* E_dir;
    double* E_lw;
    unsigned char* handoff;        // hand-off slots between the tier kernels and tlc_pd_swap_kernel
    size_t cap_handoff;
    unsigned char* handoff_large;  // the LARGE tier's own slots (only subgraphs meant for tlc_pd_dc_kernel use theirs)
    size_t cap_handoff_large;
    hipStream_t main;              // the chunk's own "caller's stream": everything the caller's stream used to carry
    hipStream_t side[TLC_N_SIDE];
    hipEvent_t ev_in, ev_done;     // fork from / join into the caller's stream
    hipEvent_t ev_fork, ev_join[TLC_N_SIDE], ev_early, ev_scan;
    int prev_tc[TLC_N_TIERS];      // tier counts of the previous chunk (sizes of the speculative launches)
    size_t x_entries_hint;         // induced entries of the largest chunk seen (sizes the bump area of the next one)
    int busy;                      // a chunk was submitted and ev_done has not been waited for on the host
    int in_call;                   // ... by the call in progress (its statistics are still to be collected)
    int n_pairs;                   // pairs of that chunk
    ChunkCtx ctx;                  // the chunk in flight
    int own_early;                 // side[4] is this workspace's own stream
    int back_pending;              // its second half has not been submitted yet
};

struct tlc_graph {
    int device;
    int n_nodes;
    long long nnz;
    int nw;
    int *d_rowptr, *d_col;
    double* d_w;
    Workspace ws[TLC_N_WS];
    unsigned long long next_ws;    // chunks submitted so far
    Workspace* last_ws;            // workspace of the most recent chunk (sizes / divide-and-conquer statistics)
    unsigned long long call_seq;   // calls of run_batch so far
    Workspace* pending;            // the chunk whose second half is still to be submitted (deferred, see run_batch)
    // vicinity kernels
    int vic_slots;
    size_t vic_lds;
    int lds_attr_set;          // hipFuncAttributeMaxDynamicSharedMemorySize raised for this handle's device and vic_lds
    // early pass of the breadth-first kernels: ball-size bounds per node for one hop value
    int ball_hop;
    int* d_ball_ub[2];
    long long last_stats[10];
    long long last_tc[TLC_N_TIERS];    // the scan's own tier counts of the last call (tlc_debug_tier_counts)
    long long acc_tie, acc_entries; // tie-fallback sources / induced entries of the chunks whose workspace was taken again within the call
    unsigned long long* d_phase;   // diagnostics: [TLC_N_TIERS][32] cycle counters, null unless enabled
    unsigned long long* d_pair_t;  // diagnostics (PAIR_TIMES builds): [cap][4] wall-clock stamps per pair of the extraction
    size_t cap_pair_t;
    // optional per-kernel timing (tlc_pd_pi_batch_set_timing): events bracket each launch on its own stream
    int timing;
    // A ring of event sets, one per chunk: a caller that enqueues batch after batch without synchronising reads the
    // durations of the last TLC_TIMING_RING chunks afterwards (tlc_pd_pi_batch_timing_history).  Created on first use.
    hipEvent_t ev_ring[TLC_TIMING_RING][16];   // per set: 0/1 count, 2/3 scan, 4/5 fill, 6+2t / 7+2t tier t
    unsigned char ev_ring_used[TLC_TIMING_RING][8];
    int ev_ring_ready;
    unsigned long long ring_pos;   // chunks timed so far; the current set is (ring_pos - 1) % TLC_TIMING_RING
    hipEvent_t* ev_t;              // the current set
    unsigned char* ev_used;
    int last_n_pairs;
    // tlc_extract_kernel (extract.hip): ball lists of one hop value (built on first use), the heavy set (fixed)
    int ball_list_hop;             // 0: none yet; -1: lists do not fit (the breadth-first kernels are used)
    int* d_bptr;
    int* d_bcol;
    unsigned* d_bbits;             // ball bitmaps (TlcVicParams::bbits: n_nodes rows of nw words); null: not built
    int* d_be_ptr;                 // ball subgraphs (TlcVicParams::be_ptr ...); null: not built
    unsigned* d_be_pos;
    double* d_be_w;
    long long be_entries;
    long long ball_entries;
    TlcNodeRec* d_nrec;            // node records (extract.hip)
    double* d_hh_w;
    int hh_k;
    int hh_diag;                   // some heavy node has a self loop
    size_t x_lds64, x_lds512, x_lds64f, x_lds64fb;     // (x_lds64fb: the subgraph-list launch without an N-bit bitmap in LDS: ball bitmaps)
    // development / test switches (tlc_debug_set_option; initial values from the environment: TLC_EXTRACT, TLC_HEAVY, TLC_TINY)
    int opt_extract, opt_heavy, opt_tiny;
    int opt_fast_split;                // the subgraph-list pairs in a launch of their own (run_chunk_front)
    int opt_dc_inplace;            // LARGE tier: divide and conquer by the tier kernel's own workgroup (TlcPdParams::dc_inplace)
    int opt_ball_edges;            // the extraction filters the smaller ball's subgraph list where there is one (extract.hip, x_sweep_ball)
    int opt_ball_bits;             // the subgraph-list launch tests membership in the larger ball against the per-node ball bitmaps (TlcVicParams::bbits)
    int opt_dc_force_fail;              // tests: see TlcPdParams::dc_force_fail
    int opt_plain_kernels;              // the PLAIN instances of the tier / swap kernels for plain image batches (tlc_launch_pd_tier); 0: the general ones
    int count_only;                     // set by tlc_vicinity_sizes around its run_batch: chunks stop after the scan, their sizes are copied out
    int opt_n_ws;                       // workspaces taken in turn (2..TLC_N_WS, default 3)
    int opt_spec_cap;                   // tests: upper bound of the slots reserved for the speculative launches (0 = none)
    int opt_timing_every;               // measurement: kernel events on every n-th chunk only
    unsigned timing_seq;
    int opt_tier_mask;                  // development: which tier kernels are launched at all (timing a tier alone; rows of the others are garbage)
    int opt_x_region, opt_x_bump_min;   // arena entries per workgroup region / minimum bump area of the extraction (tests shrink them)
};

static int finish_pending(tlc_graph* g);
// host-side wait for every chunk in flight on this handle (before shared, read-only-while-running structures change).
// A deferred chunk's second half goes in first: its ev_done is only recorded there, and its saved ChunkCtx holds pointers
// into the structures the caller is about to rebuild (ball lists of another hop, the ball-size bounds).
static int quiesce(tlc_graph* g) {
    { const int rp = finish_pending(g); if (rp != TLC_OK) return rp; }
    for (int k = 0; k < TLC_N_WS; ++k)
        if (g->ws[k].busy) { TLC_HIP_CHECK(hipEventSynchronize(g->ws[k].ev_done)); g->ws[k].busy = 0; }
    return TLC_OK;
}

static int ensure_pairs(tlc_graph* g, Workspace* ws, size_t n) {
    if (n <= ws->cap_pairs) return TLC_OK;
    hipFree(ws->hdr_n); hipFree(ws->hdr_m2); hipFree(ws->hdr_lu); hipFree(ws->hdr_lv); hipFree(ws->tier_list); hipFree(ws->edge_off);
    hipFree(ws->dc_lists); hipFree(ws->big_lists); hipFree(ws->tiny_bins);
    ws->hdr_n = ws->hdr_m2 = ws->hdr_lu = ws->hdr_lv = ws->tier_list = ws->dc_lists = ws->big_lists = ws->tiny_bins = nullptr;
    ws->edge_off = nullptr;
    ws->cap_pairs = 0;
    TLC_HIP_CHECK(hipMalloc(&ws->hdr_n, n * sizeof(int)));
    TLC_HIP_CHECK(hipMalloc(&ws->hdr_m2, n * sizeof(int)));
    TLC_HIP_CHECK(hipMalloc(&ws->hdr_lu, n * sizeof(int)));
    TLC_HIP_CHECK(hipMalloc(&ws->hdr_lv, n * sizeof(int)));
    TLC_HIP_CHECK(hipMalloc(&ws->tier_list, n * TLC_N_TIERS * sizeof(int)));
    TLC_HIP_CHECK(hipMalloc(&ws->edge_off, (n + 1) * sizeof(long long)));
    TLC_HIP_CHECK(hipMalloc(&ws->dc_lists, 3 * (n + TLC_EARLY_SLOTS) * sizeof(int)));
    TLC_HIP_CHECK(hipMalloc(&ws->big_lists, 4 * n * sizeof(int)));
    TLC_HIP_CHECK(hipMalloc(&ws->tiny_bins, (size_t)TLC_TINY_BINS * n * sizeof(int)));
    ws->cap_pairs = n;
    return TLC_OK;
}

// `keep` > 0: the first `keep` entries hold vicinities already written on stream s and survive the growth
static int ensure_arena(tlc_graph* g, Workspace* ws, size_t entries, size_t keep = 0, hipStream_t s = nullptr) {
    if (entries <= ws->cap_entries) return TLC_OK;
    size_t want = std::max(entries + entries / 4, (size_t)1 << 16);
    unsigned* old_dir = ws->A_dir;
    double* old_lw = ws->A_lw;
    ws->A_dir = nullptr; ws->A_lw = nullptr; ws->cap_entries = 0;
    if (keep == 0) { hipFree(old_dir); hipFree(old_lw); old_dir = nullptr; old_lw = nullptr; }
    TLC_HIP_CHECK(hipMalloc(&ws->A_dir, want * sizeof(unsigned)));
    TLC_HIP_CHECK(hipMalloc(&ws->A_lw, want * sizeof(double)));
    if (keep > 0) {
        TLC_HIP_CHECK(hipMemcpyAsync(ws->A_dir, old_dir, keep * sizeof(unsigned), hipMemcpyDeviceToDevice, s));
        TLC_HIP_CHECK(hipMemcpyAsync(ws->A_lw, old_lw, keep * sizeof(double), hipMemcpyDeviceToDevice, s));
        TLC_HIP_CHECK(hipDeviceSynchronize());          // (speculatively launched tier kernels may still read the old arena)
        hipFree(old_dir); hipFree(old_lw);
    }
    ws->cap_entries = want;
    return TLC_OK;
}


static int ensure_handoff(tlc_graph* g, Workspace* ws, size_t bytes) {
    if (bytes <= ws->cap_handoff) return TLC_OK;
    const size_t want = std::max(bytes + bytes / 4, (size_t)1 << 20);
    hipFree(ws->handoff);
    ws->handoff = nullptr; ws->cap_handoff = 0;
    TLC_HIP_CHECK(hipMalloc(&ws->handoff, want));
    ws->cap_handoff = want;
    return TLC_OK;
}

static int ensure_handoff_large(tlc_graph* g, Workspace* ws, size_t slots) {
    if (slots * tlc_handoff_slot_bytes(TLC_TIER_LARGE) <= ws->cap_handoff_large) return TLC_OK;
    // (a quarter more than asked for: how many LARGE vicinities the early pass leaves to the main launch differs from call to call
    // on the same list, and hipFree waits for everything on the device -- see the caller)
    const size_t bytes = (slots + slots / 4) * tlc_handoff_slot_bytes(TLC_TIER_LARGE);
    hipFree(ws->handoff_large);
    ws->handoff_large = nullptr; ws->cap_handoff_large = 0;
    TLC_HIP_CHECK(hipMalloc(&ws->handoff_large, bytes));
    ws->cap_handoff_large = bytes;
    return TLC_OK;
}

static int ensure_small(tlc_graph* g, Workspace* ws, size_t n_pairs) {
    if (n_pairs <= ws->cap_small) return TLC_OK;
    hipFree(ws->S_dir); hipFree(ws->S_lw);
    ws->S_dir = nullptr; ws->S_lw = nullptr; ws->cap_small = 0;
    TLC_HIP_CHECK(hipMalloc(&ws->S_dir, n_pairs * (2 * TLC_S_MMAX) * sizeof(unsigned)));
    TLC_HIP_CHECK(hipMalloc(&ws->S_lw, n_pairs * (2 * TLC_S_MMAX) * sizeof(double)));
    ws->cap_small = n_pairs;
    return TLC_OK;
}

static int ensure_vic_scratch(tlc_graph* g, Workspace* ws, int hop) {
    const int need_front = hop >= 3 ? 1 : 0;
    if (ws->vic_scratch && ws->vic_hop_cap >= need_front) return TLC_OK;
    hipFree(ws->vic_scratch);
    ws->vic_scratch = nullptr;
    // slot = [ids | lrow | frontA | frontB]: the id list and the row offsets of one vicinity (at most 65 535 nodes, whatever
    // the graph's size), and for hop >= 3 the two BFS frontiers (up to n_nodes each)
    const long long cap = std::min<long long>(g->n_nodes, TLC_MAX_SUBGRAPH_NODES + 1);
    ws->vic_stride = 2 * cap + (need_front ? 2ll * g->n_nodes : 0) + 16;
    TLC_HIP_CHECK(hipMalloc(&ws->vic_scratch, (size_t)(g->vic_slots + TLC_EARLY_WG) * ws->vic_stride * sizeof(int)));
    ws->vic_hop_cap = need_front;
    return TLC_OK;
}

// per-node upper bound of |ball_hop(x)| (vicinity.hip, tlc_ball_bound_kernel) and the early pass's buffers; one-off per
// (graph, hop)
static int ensure_early(tlc_graph* g, Workspace* ws, int hop, hipStream_t s) {
    if (!g->d_ball_ub[0]) {
        TLC_HIP_CHECK(hipMalloc(&g->d_ball_ub[0], (size_t)g->n_nodes * sizeof(int)));
        TLC_HIP_CHECK(hipMalloc(&g->d_ball_ub[1], (size_t)g->n_nodes * sizeof(int)));
    }
    if (!ws->d_cand_list) {
        TLC_HIP_CHECK(hipMalloc(&ws->d_early_list, TLC_EARLY_SLOTS * sizeof(int)));
        TLC_HIP_CHECK(hipMalloc(&ws->E_dir, (size_t)TLC_EARLY_SLOTS * 2 * TLC_L_MMAX * sizeof(unsigned)));
        TLC_HIP_CHECK(hipMalloc(&ws->E_lw, (size_t)TLC_EARLY_SLOTS * 2 * TLC_L_MMAX * sizeof(double)));
        TLC_HIP_CHECK(hipMalloc(&ws->d_cand_list, TLC_EARLY_CAND * sizeof(int)));
    }
    if (g->ball_hop != hop) {
        int rc;
        if ((rc = quiesce(g)) != TLC_OK) return rc;          // (a chunk in flight on another workspace may be reading the bounds)
        for (int h = 1; h <= hop; ++h) {
            // the result of round h lands in buffer (h - 1) & 1 ... the caller reads buffer (hop - 1) & 1
            if ((rc = tlc_launch_ball_bound(g->n_nodes, g->d_rowptr, g->d_col, h == 1 ? nullptr : g->d_ball_ub[h & 1],
                                            g->d_ball_ub[(h - 1) & 1], s)) != TLC_OK) return rc;
        }
        g->ball_hop = hop;
    }
    return TLC_OK;
}

static int ensure_huge(tlc_graph* g, Workspace* ws) {
    if (ws->huge_scratch) return TLC_OK;
    const int nmax = std::min(g->n_nodes, 65535);
    const long long mmax = g->nnz / 2 + 1;
    ws->huge_stride = tlc_huge_slot_bytes(nmax, (int)mmax);
    ws->huge_slots = 64;
    TLC_HIP_CHECK(hipMalloc(&ws->huge_scratch, ws->huge_stride * (size_t)ws->huge_slots));
    return TLC_OK;
}


// ---- the heavy set of tlc_extract_kernel (extract.hip) -------------------------------------------------------------------
// The hh_k <= 256 nodes of the highest degrees >= 32.  Their rows are never read by the extraction sweep: an entry h -> x
// is emitted as the mirror of x -> h found in row x, which needs the CSR to be symmetric AT THE HEAVY ROWS (checked here,
// entry for entry, weights included) and entries between two heavy nodes come from a dense table (so a repeated entry
// between two heavy nodes cannot be represented).  Whatever fails switches the heavy set off; the sweep then reads every row.
#define TLC_HEAVY_MIN_DEG 32
#define TLC_HEAVY_MAX 256
static int build_heavy_set(tlc_graph* g, const int32_t* rp, const int32_t* col, const double* w, std::vector<int>* hidx_out) {
    const int n = g->n_nodes;
    std::vector<int> cand;
    for (int x = 0; x < n; ++x) if (rp[x + 1] - rp[x] >= TLC_HEAVY_MIN_DEG) cand.push_back(x);
    if (cand.empty()) return TLC_OK;
    if ((int)cand.size() > TLC_HEAVY_MAX) {
        std::nth_element(cand.begin(), cand.begin() + TLC_HEAVY_MAX, cand.end(), [&](int a, int b) {
            const int da = rp[a + 1] - rp[a], db = rp[b + 1] - rp[b];
            return da != db ? da > db : a < b;
        });
        cand.resize(TLC_HEAVY_MAX);
    }
    std::sort(cand.begin(), cand.end());
    const int K = (int)cand.size();
    std::vector<int> hidx((size_t)n, -1);
    for (int k = 0; k < K; ++k) hidx[cand[k]] = k;
    std::vector<double> hh((size_t)K * K, 0.0);
    bool ok = true;
    std::vector<int> seen((size_t)n, -1);                     // last heavy row that listed this column
    for (int k = 0; k < K && ok; ++k) {
        const int h = cand[k];
        for (int j = rp[h]; j < rp[h + 1] && ok; ++j) {
            const int y = col[j];
            if (seen[y] == k) { ok = false; break; }          // a column twice in one heavy row
            seen[y] = k;
            if (hidx[y] >= 0) {
                hh[(size_t)k * K + hidx[y]] = w[j];
            } else {
                // the mirror entry y -> h with the same weight, exactly once
                int found = 0;
                for (int t = rp[y]; t < rp[y + 1]; ++t) if (col[t] == h) found += (w[t] == w[j]) ? 1 : 2;
                if (found != 1) ok = false;
            }
        }
    }
    for (int a = 0; a < K && ok; ++a)
        for (int b = 0; b < K && ok; ++b) if ((hh[(size_t)a * K + b] != 0.0) != (hh[(size_t)b * K + a] != 0.0) || hh[(size_t)a * K + b] != hh[(size_t)b * K + a]) ok = false;
    if (ok) {
        TLC_HIP_CHECK(hipMalloc(&g->d_hh_w, (size_t)K * K * sizeof(double)));
        TLC_HIP_CHECK(hipMemcpy(g->d_hh_w, hh.data(), (size_t)K * K * sizeof(double), hipMemcpyHostToDevice));
        g->hh_k = K;
        for (int a = 0; a < K; ++a) if (hh[(size_t)a * K + a] != 0.0) g->hh_diag = 1;
        *hidx_out = hidx;
    }
    return TLC_OK;                                            // (not ok: no heavy set, the sweep reads every row)
}

// the node records of the extraction (TlcNodeRec, tlc_kernels.h)
static int build_node_records(tlc_graph* g, const int32_t* rp, const int32_t* col, const double* w, const std::vector<int>& hidx) {
    const int n = g->n_nodes;
    std::vector<TlcNodeRec> rec((size_t)n);
    for (int x = 0; x < n; ++x) {
        TlcNodeRec& r = rec[x];
        memset(&r, 0, sizeof(r));
        r.row_start = rp[x]; r.deg = rp[x + 1] - rp[x];
        r.hidx = hidx.empty() ? -1 : hidx[x];
        r.n_in = r.deg < 4 ? r.deg : 4;
        for (int q = 0; q < r.n_in; ++q) { r.col[q] = col[rp[x] + q]; r.w[q] = w[rp[x] + q]; }
    }
    TLC_HIP_CHECK(hipMalloc(&g->d_nrec, (size_t)n * sizeof(TlcNodeRec)));
    TLC_HIP_CHECK(hipMemcpy(g->d_nrec, rec.data(), (size_t)n * sizeof(TlcNodeRec), hipMemcpyHostToDevice));
    return TLC_OK;
}

// ball_hop(x) of every node as sorted id lists, for tlc_extract_kernel; one-off per (graph, hop), synchronous (the sizes come
// back to the host for the offsets).  Lists beyond 2^31 - 1 entries or a quarter of the free memory are not built:
// ball_list_hop = -1 and the caller uses the breadth-first kernels.
static int ensure_ball_lists(tlc_graph* g, int hop, hipStream_t s) {
    if (g->ball_list_hop == hop) return TLC_OK;
    {
        const int rq = quiesce(g);                            // (a chunk in flight on another workspace may be reading the lists)
        if (rq != TLC_OK) return rq;
    }
    hipFree(g->d_bptr); hipFree(g->d_bcol);
    hipFree(g->d_be_ptr); hipFree(g->d_be_pos); hipFree(g->d_be_w); hipFree(g->d_bbits);
    g->d_bptr = g->d_bcol = nullptr; g->ball_list_hop = 0; g->ball_entries = 0;
    g->d_be_ptr = nullptr; g->d_be_pos = nullptr; g->d_be_w = nullptr; g->be_entries = 0; g->d_bbits = nullptr;
    const int n = g->n_nodes;
    int* d_size = nullptr;
    TLC_HIP_CHECK(hipMalloc(&d_size, (size_t)n * sizeof(int)));
    const int grid = std::min(n, 4096);
    int rc = tlc_launch_ball_list(false, n, g->nw, g->d_rowptr, g->d_col, hop, d_size, nullptr, nullptr, grid, s);
    std::vector<int> sz((size_t)n + 1, 0);
    if (rc == TLC_OK && hipMemcpyAsync(sz.data(), d_size, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) rc = TLC_ERR_HIP;
    if (rc == TLC_OK && hipStreamSynchronize(s) != hipSuccess) rc = TLC_ERR_HIP;
    hipFree(d_size);
    if (rc != TLC_OK) { tlc_set_error("ball lists: %s", hipGetErrorString(hipGetLastError())); return rc; }
    long long tot = 0;
    for (int x = 0; x < n; ++x) { const int c = sz[x]; sz[x] = (int)tot; tot += c; if (tot > 0x7fffffffll) break; }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;       // (unknown: the lists are not built, the breadth-first kernels serve)
    if (tot > 0x7fffffffll || (size_t)tot * sizeof(int) > free_b / 4) { g->ball_list_hop = -1; return TLC_OK; }
    sz[n] = (int)tot;
    TLC_HIP_CHECK(hipMalloc(&g->d_bptr, ((size_t)n + 1) * sizeof(int)));
    TLC_HIP_CHECK(hipMalloc(&g->d_bcol, ((size_t)tot + 8) * sizeof(int)));
    TLC_HIP_CHECK(hipMemcpyAsync(g->d_bptr, sz.data(), ((size_t)n + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    if ((rc = tlc_launch_ball_list(true, n, g->nw, g->d_rowptr, g->d_col, hop, nullptr, g->d_bptr, g->d_bcol, grid, s)) != TLC_OK) return rc;
    TLC_HIP_CHECK(hipStreamSynchronize(s));                  // (sz is host memory of this frame)
    g->ball_list_hop = hop; g->ball_entries = tot;
    // the ball subgraphs (nodes whose ball has at most TLC_BE_CAP members): count, prefix on the host, fill.  Built into locals and
    // published on success only; without them the extraction sweeps the members' rows as before.
    {
        int* d_es = nullptr;
        int* d_bp = nullptr;
        unsigned* d_pos = nullptr;
        double* d_bw = nullptr;
        std::vector<int> es((size_t)n + 1, 0);
        bool ok = hipMalloc(&d_es, (size_t)n * sizeof(int)) == hipSuccess;
        ok = ok && tlc_launch_ball_edges(false, n, g->nw, g->d_rowptr, g->d_col, g->d_w, g->d_bptr, g->d_bcol, d_es, nullptr, nullptr, nullptr, grid, s) == TLC_OK;
        ok = ok && hipMemcpyAsync(es.data(), d_es, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess;
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
        long long et = 0;
        if (ok) for (int x = 0; x < n; ++x) { const int c = es[x]; es[x] = (int)et; et += c; if (et > 0x7fffffffll) { ok = false; break; } }
        if (ok) {
            es[n] = (int)et;
            size_t fb = 0, tb = 0;
            ok = hipMemGetInfo(&fb, &tb) == hipSuccess && (size_t)et * 12 <= fb / 8;
        }
        ok = ok && hipMalloc(&d_bp, ((size_t)n + 1) * sizeof(int)) == hipSuccess;
        ok = ok && hipMalloc(&d_pos, ((size_t)et + 64) * sizeof(unsigned)) == hipSuccess;
        ok = ok && hipMalloc(&d_bw, ((size_t)et + 64) * sizeof(double)) == hipSuccess;
        ok = ok && hipMemcpyAsync(d_bp, es.data(), ((size_t)n + 1) * sizeof(int), hipMemcpyHostToDevice, s) == hipSuccess;
        ok = ok && tlc_launch_ball_edges(true, n, g->nw, g->d_rowptr, g->d_col, g->d_w, g->d_bptr, g->d_bcol, nullptr, d_bp, d_pos, d_bw, grid, s) == TLC_OK;
        ok = ok && hipStreamSynchronize(s) == hipSuccess;                 // (es is host memory of this frame)
        hipFree(d_es);
        if (ok) { g->d_be_ptr = d_bp; g->d_be_pos = d_pos; g->d_be_w = d_bw; g->be_entries = et; }
        else { hipFree(d_bp); hipFree(d_pos); hipFree(d_bw); (void)hipGetLastError(); }
    }
    // the ball bitmaps (round 6): N rows of nw words -- N^2 / 8 bytes: 48 MB for PubMed's 19 717 nodes.  Built for the subgraph-list
    // launch when they stay within 1 GiB (N <= ~92 000) and an eighth of the free memory; else that launch marks the larger ball in
    // its LDS bitmap as before.  Published on success only.
    if (g->d_be_ptr) {
        const size_t bytes = (size_t)n * (size_t)g->nw * sizeof(unsigned);
        size_t fb = 0, tb = 0;
        unsigned* d_bb = nullptr;
        bool ok = bytes <= ((size_t)1 << 30) && hipMemGetInfo(&fb, &tb) == hipSuccess && bytes <= fb / 8;
        ok = ok && hipMalloc(&d_bb, bytes) == hipSuccess;
        ok = ok && hipMemsetAsync(d_bb, 0, bytes, s) == hipSuccess;
        ok = ok && tlc_launch_ball_bits(n, g->nw, g->d_bptr, g->d_bcol, d_bb, s) == TLC_OK;
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
        if (ok) g->d_bbits = d_bb;
        else { hipFree(d_bb); (void)hipGetLastError(); }
    }
    return TLC_OK;
}

extern "C" int tlc_graph_create(int32_t n_nodes, const int32_t* h_rowptr, const int32_t* h_col, const double* h_w,
                                int device, tlc_graph** out) {
    TLC_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    TLC_REQUIRE(n_nodes > 0 && h_rowptr && h_col && h_w, "null graph arrays or n_nodes <= 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        tlc_set_error("no HIP device visible");
        return TLC_ERR_NO_DEVICE;
    }
    TLC_REQUIRE(device >= 0 && device < ndev, "device index out of range");
    const long long nnz = h_rowptr[n_nodes];
    TLC_REQUIRE(h_rowptr[0] == 0 && nnz >= 0, "rowptr must start at 0");
    for (int i = 0; i < n_nodes; ++i) TLC_REQUIRE(h_rowptr[i + 1] >= h_rowptr[i], "rowptr not monotone");
    for (long long j = 0; j < nnz; ++j) {
        TLC_REQUIRE(h_col[j] >= 0 && h_col[j] < n_nodes, "column index out of range");
        TLC_REQUIRE(h_w[j] > 0.0, "edge weights (kappa+1) must be > 0");
    }
    const int nw = (n_nodes + 31) / 32;
    const size_t lds = ((size_t)5 * ((nw + 3) & ~3) / 2 + 4 + 16) * 4;   // two bitmaps + a 16-bit prefix per bitmap word
    if (lds > 160 * 1024) {
        tlc_set_error("graph has %d nodes: the vicinity bitmaps (%zu B) exceed the 160 KiB LDS of a CU", n_nodes, lds);
        return TLC_ERR_UNSUPPORTED;
    }
    TLC_ON_DEVICE(device);
    tlc_graph* g = new (std::nothrow) tlc_graph();
    if (!g) return TLC_ERR_OUT_OF_MEMORY;
    memset(g, 0, sizeof(*g));
    g->device = device; g->n_nodes = n_nodes; g->nnz = nnz; g->nw = nw; g->vic_lds = lds;
    auto env_on = [](const char* name) { const char* v = getenv(name); return !(v && v[0] == '0'); };
    g->opt_extract = env_on("TLC_EXTRACT"); g->opt_heavy = env_on("TLC_HEAVY"); g->opt_tiny = env_on("TLC_TINY");
    g->opt_ball_edges = env_on("TLC_BALL_EDGES"); g->opt_dc_inplace = env_on("TLC_DC_INPLACE") ? 1 : 0;
    g->opt_fast_split = env_on("TLC_FAST_SPLIT") ? 1 : 0;
    g->opt_ball_bits = env_on("TLC_BALL_BITS") ? 1 : 0;
    g->opt_plain_kernels = env_on("TLC_PLAIN_KERNELS") ? 1 : 0;
    g->opt_n_ws = 3;
    g->opt_x_region = TLC_X_REGION; g->opt_x_bump_min = 1 << 20; g->opt_tier_mask = (1 << TLC_N_TIERS) - 1; g->opt_timing_every = 1;
    int rc = TLC_OK;
    auto fail = [&](int code) { tlc_graph_destroy(g); return code; };
#define CK(e) do { if ((e) != hipSuccess) { tlc_set_error("%s failed: %s", #e, hipGetErrorString(hipGetLastError())); return fail(TLC_ERR_HIP); } } while (0)
    CK(hipMalloc(&g->d_rowptr, (size_t)(n_nodes + 1) * sizeof(int)));
    // 8 entries of zero padding: the vicinity kernels read short rows with 16-byte loads that may run past a row's end
    CK(hipMalloc(&g->d_col, ((size_t)nnz + 8) * sizeof(int)));
    CK(hipMalloc(&g->d_w, ((size_t)nnz + 8) * sizeof(double)));
    CK(hipMemset(g->d_col + nnz, 0, 8 * sizeof(int)));
    CK(hipMemset(g->d_w + nnz, 0, 8 * sizeof(double)));
    CK(hipMemcpy(g->d_rowptr, h_rowptr, (size_t)(n_nodes + 1) * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(g->d_col, h_col, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(g->d_w, h_w, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice));
    // Streams.  The runtime maps streams onto a pool of 4 hardware queues PER PRIORITY, least-shared first, and a stream that
    // is blocked on an event stalls whatever shares its queue; so the handle keeps within 4 streams per priority, the caller's
    // own stream included (normal priority):
    //   per workspace: main (normal; carries what the caller's stream carries for a single stream-ordered chunk) and
    //                  side[4], the early chain (high) -- the lead-in of the next chunk must not queue behind this chunk's tiers;
    //   shared by the workspaces (a chunk's tier kernels queue behind the previous chunk's, which is the order they finish in
    //                  anyway): side[0] SMALL, side[5] TINY, side[3] MID (low), side[6] MEDIUM and its rarely used twin
    //                  side[2], side[7] MEDWIDE (normal), side[1] the heavy tiers the early pass did not take (high).
    // (MEDWIDE, pipelined chunks only: on MEDIUM's stream the few hundred largest MEDIUM-sized vicinities of a batch, 0.38 ms of
    // tier + swap kernel, ran in front of the MEDIUM chain instead of beside it.  A fifth stream of normal priority shares a
    // queue with another one; in-process A/B, tools/gpu_env_ab3.sh TLC_MEDWIDE_PRIO=0|1|2: 0.683 / 0.651 / 0.703 ms per pipelined
    // batch -- as the fourth stream of the low pool it starts too late, in the high pool it delays the early chains.
    // TINY or SMALL in the normal pool (TLC_TINY_PRIO / TLC_SMALL_PRIO = 1): 0.81 / 0.75 ms.  More hardware queues for the runtime
    // (GPU_MAX_HW_QUEUES=6 / 8, its environment variable; 4 is the default this layout was tuned for): 0.69 ms; 2: 0.95 ms.)
    // Normal priority then holds the caller's stream, the two main streams and MEDIUM: four.  With MID there as well (round 2)
    // the second workspace's main stream shared a hardware queue with the MID chain, and the lead-in of every other pipelined
    // batch sat behind ~0.3 ms of tier + swap kernel: 45.8 -> 48.4 M images/s pipelined, 0.833 -> 0.840 ms for one batch alone
    // (profiles/r03_async_timeline.txt shows the queue ids).
    // (Dedicated queues through hipExtStreamCreateWithCUMask -- with a full mask, or with CUs reserved for the whole-CU
    // workgroups of the LARGE tier -- were measured: 1.2 - 3.3 ms per batch instead of 0.9.  tools/probes/cumask_probe.hip.)
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const int prio_mid = (prio_lo + prio_hi) / 2;
    // The streams of the three workspaces in use exist from the start: the runtime deals hardware queues to streams as they are
    // created, and streams that come into being later -- a framework's capture streams, a communicator's -- would otherwise take
    // the queues the second and third workspace get with their first chunk (measured: a HIP graph captured between the handle's
    // creation and its first pipelined batch cost 6 % images/s; none with the streams created here).  TLC_LAZY_STREAMS=1: as before.
    const bool eager_streams = getenv("TLC_LAZY_STREAMS") == nullptr;
    // (Four workspaces within the same eleven streams -- early streams shared by workspaces i and i + 2, MEDIUM's side stream in
    // the low-priority pool -- were measured: 0.655 vs 0.643 ms per pipelined batch; the LARGE chains of two chunks on one stream
    // cost more than the fourth chunk in flight gains.  With three workspaces the submitting thread does wait 0.1 - 0.4 ms for a
    // workspace in most calls (TLC_HOST_TRACE), but a fourth with streams of its own -- thirteen in all -- is no faster either:
    // the wait is the machine's queue, not a missing slot.)
    const int n_eager = 3;
    for (int i = 0; i < TLC_N_WS; ++i) {
        Workspace* ws = &g->ws[i];
        CK(hipMalloc(&ws->d_ctl, TLC_CTL_INTS * sizeof(int)));   // counters, the scan's per-block flags, 4 x u64 statistics, work counters
        ws->d_stats = reinterpret_cast<unsigned long long*>(ws->d_ctl + 64 + TLC_SCAN_MAX_BLOCKS);      // (8-byte aligned: hipMalloc is 256-byte aligned)
        CK(hipMalloc(&ws->d_block_sums, TLC_SCAN_MAX_BLOCKS * sizeof(long long)));
        CK(hipMalloc(&ws->d_totals, 2 * sizeof(long long)));
        CK(hipHostMalloc((void**)&ws->h_sync, sizeof(HostSync), hipHostMallocMapped | hipHostMallocCoherent));
        memset(ws->h_sync, 0, sizeof(HostSync));
        CK(hipHostGetDevicePointer((void**)&ws->h_sync_dev, ws->h_sync, 0));
        // (the fourth workspace's own two streams are created when the first chunk lands on it: option n_ws = 4 only)
        if (i == 0 || (eager_streams && i < n_eager)) CK(hipStreamCreateWithPriority(&ws->main, hipStreamNonBlocking, prio_mid));
        for (int k = 0; k < TLC_N_SIDE; ++k) {
            if (k == 4) {
                if (i == 0 || (eager_streams && i < n_eager)) { CK(hipStreamCreateWithPriority(&ws->side[k], hipStreamNonBlocking, prio_hi)); ws->own_early = 1; }
            }
            else if (i > 0) {
                ws->side[k] = g->ws[0].side[k];
                // the third workspace's MEDWIDE list goes to the LARGE list's stream, which is idle while the early pass carries the
                // LARGE list (always, in pipelined chunks): the MEDWIDE stream's queue was the busiest of a pipelined region (0.74: tier
                // kernel + the longest serial swaps of three chunks in order, and a main stream on the same hardware queue), every
                // chunk's LAST kernel was its MEDWIDE swap, started behind the previous chunk's.  0.576 -> 0.551 ms per pipelined batch;
                // the other assignments (second workspace's, the MEDIUM list's, the swaps on a stream of their own) were worse
                // (profiles/r05_stream_assignment.txt)
                if (i == 2 && k == 7) ws->side[k] = g->ws[0].side[1];
            }
            else if (k == 2) ws->side[k] = nullptr;                      // (= side[6], set below)
            else {
                int pr = k == 1 ? prio_hi : ((k == 6 || k == 7) ? prio_mid : prio_lo);
                // (development A/B, tools/gpu_prio_ab.sh: 0 low, 1 normal, 2 high)
                const char* ev = k == 3 ? getenv("TLC_MID_PRIO") : (k == 6 ? getenv("TLC_MEDIUM_PRIO") : (k == 7 ? getenv("TLC_MEDWIDE_PRIO") : (k == 5 ? getenv("TLC_TINY_PRIO") : (k == 0 ? getenv("TLC_SMALL_PRIO") : nullptr))));
                if (ev) pr = atoi(ev) == 0 ? prio_lo : (atoi(ev) == 1 ? prio_mid : prio_hi);
                CK(hipStreamCreateWithPriority(&ws->side[k], hipStreamNonBlocking, pr));
            }
            CK(hipEventCreateWithFlags(&ws->ev_join[k], hipEventDisableTiming));
        }
        if (i == 0) ws->side[2] = ws->side[6];
        CK(hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ws->ev_early, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ws->ev_scan, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ws->ev_cls, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ws->ev_in, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ws->ev_done, hipEventDisableTiming));
    }
#undef CK
    // concurrent vicinity workgroups worth launching: LDS-bound per CU, 256 CUs
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    int per_cu = (int)std::min<size_t>(28, (160 * 1024) / std::max<size_t>(lds + 64, 1));
    if (per_cu < 1) per_cu = 1;
    g->x_lds64 = tlc_extract_lds_bytes(nw, 64);
    g->x_lds512 = tlc_extract_lds_bytes(nw, 512);
    g->x_lds64f = tlc_extract_lds_bytes(nw, 64, true);
    g->x_lds64fb = tlc_extract_lds_bytes(0, 64, true);
    // (the extraction kernel's workgroups are smaller: the scratch slots cover whichever kernel runs more of them)
    per_cu = std::max(per_cu, (int)std::min<size_t>(32, (160 * 1024) / std::max<size_t>(g->x_lds64 + 64, 1)));
    g->vic_slots = cus * per_cu;
    {
        std::vector<int> hidx;
        if ((rc = build_heavy_set(g, h_rowptr, h_col, h_w, &hidx)) != TLC_OK) return fail(rc);
        if ((rc = build_node_records(g, h_rowptr, h_col, h_w, hidx)) != TLC_OK) return fail(rc);
    }
    *out = g;
    return TLC_OK;
}

extern "C" int tlc_graph_destroy(tlc_graph* g) {
    if (!g) return TLC_OK;
    TlcDeviceScope scope(g->device);
    hipDeviceSynchronize();
    hipFree(g->d_rowptr); hipFree(g->d_col); hipFree(g->d_w);
    for (int i = 0; i < TLC_N_WS; ++i) {
        Workspace* ws = &g->ws[i];
        hipFree(ws->hdr_n); hipFree(ws->hdr_m2); hipFree(ws->hdr_lu); hipFree(ws->hdr_lv); hipFree(ws->tier_list); hipFree(ws->edge_off);
        hipFree(ws->dc_lists); hipFree(ws->big_lists); hipFree(ws->tiny_bins);
        hipFree(ws->d_ctl); hipFree(ws->d_block_sums); hipFree(ws->d_totals);
        if (ws->h_sync) hipHostFree(ws->h_sync);
        hipFree(ws->A_dir); hipFree(ws->A_lw); hipFree(ws->S_dir); hipFree(ws->S_lw); hipFree(ws->vic_scratch); hipFree(ws->huge_scratch);
        hipFree(ws->handoff); hipFree(ws->handoff_large);
        hipFree(ws->d_cand_list); hipFree(ws->d_early_list); hipFree(ws->E_dir); hipFree(ws->E_lw);
        if (ws->main) hipStreamDestroy(ws->main);
        for (int k = 0; k < TLC_N_SIDE; ++k) {
            const bool own = (k == 4 && ws->own_early) || (i == 0 && k != 2);   // (the rest are workspace 0's, see tlc_graph_create)
            if (own && ws->side[k]) hipStreamDestroy(ws->side[k]);
            if (ws->ev_join[k]) hipEventDestroy(ws->ev_join[k]);
        }
        for (hipEvent_t e : {ws->ev_fork, ws->ev_early, ws->ev_scan, ws->ev_in, ws->ev_done, ws->ev_cls}) if (e) hipEventDestroy(e);
    }
    hipFree(g->d_phase); hipFree(g->d_pair_t);
    hipFree(g->d_bptr); hipFree(g->d_bcol); hipFree(g->d_nrec); hipFree(g->d_hh_w);
    hipFree(g->d_be_ptr); hipFree(g->d_be_pos); hipFree(g->d_be_w); hipFree(g->d_bbits);
    hipFree(g->d_ball_ub[0]); hipFree(g->d_ball_ub[1]);
    if (g->ev_ring_ready)
        for (int r = 0; r < TLC_TIMING_RING; ++r)
            for (int k = 0; k < 16; ++k) if (g->ev_ring[r][k]) hipEventDestroy(g->ev_ring[r][k]);
    delete g;
    return TLC_OK;
}

// one chunk (<= TLC_CHUNK_PAIRS pairs)
// one wavefront that returns when *counter >= target or after max_ticks of the 100 MHz wall clock
__global__ void tlc_wait_started(const int* counter, int target, long long max_ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && wall_clock64() - t0 < max_ticks)
        __builtin_amdgcn_s_sleep(8);
}

// the same with the target on the device (the early pass's LARGE count), clamped to `cap`
__global__ void tlc_wait_started_dev(const int* counter, const int* target, int cap, long long max_ticks) {
    if (threadIdx.x != 0) return;
    int tg = *target;
    tg = tg < cap ? tg : cap;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tg && wall_clock64() - t0 < max_ticks)
        __builtin_amdgcn_s_sleep(8);
}

// The first half of a chunk, by stage (round 6: one 350-line function before).  The stages share the chunk's context (ChunkCtx: the
// parameter blocks of the extraction and of the tier kernels, what the preparation decided) and are submitted in this order:
//   front_prepare      buffers, parameter blocks, which extraction serves the chunk, arena layout
//   front_fast         the fork of the early chain, then the subgraph-list launch (needs nothing from the classification)
//   front_early_chain  classification -> early pass -> LARGE tier kernel on the early stream
//   front_join_early   what the chunk's own stream waits for before the general launch (classification / early pass + residency gate)
//   front_main_scan    the general extraction launch and the scan that publishes the chunk's sizes
//   front_speculative  (a chunk on its own) the many-Pos MEDIUM list behind the scan, sized from the previous chunk
#define T0(k, st) do { if ((c.tmask >> (k)) & 1) { TLC_HIP_CHECK(hipEventRecord(c.ev_t[2 * (k)], st)); } } while (0)
#define T1(k, st) do { if ((c.tmask >> (k)) & 1) { TLC_HIP_CHECK(hipEventRecord(c.ev_t[2 * (k) + 1], st)); c.ev_used[k] = 1; } } while (0)
// lists for tlc_pd_dc_kernel: counters in the control block (zeroed with it), [d_ctl + 26 + 2k];
// k = 0 MEDIUM, 1 LARGE (regular launch), 2 LARGE (early launch)
static inline void dc_lists_for(Workspace* ws, TlcPdParams& q, int k) {
    const size_t cap = ws->cap_pairs + TLC_EARLY_SLOTS;
    q.dc_count = ws->d_ctl + 26 + 2 * k;
    q.dc_list = ws->dc_lists + (size_t)k * cap;
}
// timing slot of each tier kernel (TINY is reported with SMALL, MEDHI / MEDWIDE as MEDIUM)
static const int tslot[TLC_N_TIERS] = {3, 4, 5, 6, 7, 3, 4, 4};

static int front_prepare(tlc_graph* g, Workspace* ws, const int32_t* d_pairs, int n_pairs, int hop, uint32_t flags, int res,
                         double* d_out_pi, uint8_t* d_out_status, const int64_t* d_ids_off, int32_t* d_out_ids,
                         double* d_out_f, int32_t* d_out_n, const int64_t* d_edge_offs, int32_t* d_out_edges, int32_t* d_out_m,
                         int pi_enabled, hipStream_t s, bool pipelined) {
    int rc;
    ChunkCtx& c = ws->ctx;
    c.ht0 = std::chrono::steady_clock::now();
    c.s = s; c.n_pairs = n_pairs; c.hop = hop; c.pi_enabled = pi_enabled; c.call_seq = g->call_seq;
    if ((rc = ensure_pairs(g, ws, (size_t)n_pairs)) != TLC_OK) return rc;
    if ((rc = ensure_vic_scratch(g, ws, hop)) != TLC_OK) return rc;
    TLC_HIP_CHECK(hipMemsetAsync(ws->d_ctl, 0, TLC_CTL_INTS * sizeof(int), s));       // control words, scan flags, statistics, work counters

    TlcVicParams& vp = c.vp;
    memset(&vp, 0, sizeof(vp));
    vp.n_nodes = g->n_nodes; vp.nw = g->nw; vp.rowptr = g->d_rowptr; vp.col = g->d_col; vp.w = g->d_w;
    vp.dbg = g->d_phase ? g->d_phase + 32 * TLC_TIER_HUGE : nullptr;   // (diagnostics share the HUGE tier's counter row)
    if (g->d_phase) {
        if (g->cap_pair_t < (size_t)n_pairs) {
            hipFree(g->d_pair_t); g->d_pair_t = nullptr; g->cap_pair_t = 0;
            TLC_HIP_CHECK(hipMalloc(&g->d_pair_t, (size_t)n_pairs * 16 * sizeof(unsigned long long)));
            g->cap_pair_t = (size_t)n_pairs;
        }
        TLC_HIP_CHECK(hipMemsetAsync(g->d_pair_t, 0, (size_t)n_pairs * 16 * sizeof(unsigned long long), s));
        vp.dbg_pair_t = g->d_pair_t;
    }
    vp.pairs = d_pairs; vp.n_pairs = n_pairs; vp.hop = hop; vp.flags = flags; vp.res = res;
    vp.scratch = ws->vic_scratch; vp.scratch_stride = ws->vic_stride;
    vp.hdr_n = ws->hdr_n; vp.hdr_m2 = ws->hdr_m2; vp.hdr_lu = ws->hdr_lu; vp.hdr_lv = ws->hdr_lv;
    vp.out_pi = d_out_pi; vp.out_status = d_out_status; vp.out_n = d_out_n; vp.out_m = d_out_m;
    vp.edge_off = ws->edge_off; vp.A_dir = nullptr; vp.A_lw = nullptr;
    vp.ids_off = (const long long*)d_ids_off; vp.out_ids = d_out_ids;

    // (the attribute is per device and per size: tracked in the handle, which is bound to one device and one graph size)
    if (!g->lds_attr_set && g->vic_lds > 64 * 1024) {
        TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_vicinity_kernel<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->vic_lds));
        TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_vicinity_kernel<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->vic_lds));
        TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_vicinity_kernel<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->vic_lds));
        TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_vicinity_kernel<false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->vic_lds));
        g->lds_attr_set = 1;
    }
    const int vgrid = std::min(n_pairs, g->vic_slots);
    // (opt_timing_every = n: only every n-th chunk carries the event records -- each costs the stream it is recorded on a few us)
    const int tmask = (g->timing && (g->timing_seq++ % (unsigned)std::max(g->opt_timing_every, 1)) == 0) ? g->timing : 0;
    if (tmask) {
        if (!g->ev_ring_ready) {
            for (int r = 0; r < TLC_TIMING_RING; ++r)
                for (int k = 0; k < 16; ++k) TLC_HIP_CHECK(hipEventCreate(&g->ev_ring[r][k]));
            g->ev_ring_ready = 1;
        }
        const int set = (int)(g->ring_pos++ % TLC_TIMING_RING);
        g->ev_t = g->ev_ring[set];
        g->ev_used = g->ev_ring_used[set];
        memset(g->ev_used, 0, 8);
        c.ev_t = g->ev_t; c.ev_used = g->ev_used;
    }
    c.tmask = tmask; c.vgrid = vgrid;
    g->last_n_pairs = n_pairs;
    // ---- early pass --------------------------------------------------------------------------------------------------------
    // The batch waits for its largest vicinity: 0.9 ms of mostly serial work that used to start only after COUNT, the scan,
    // the size publication and the heavy FILL (0.31 ms into the batch).  The pairs that can be that large are known up
    // front -- a vicinity is no larger than the smaller of its endpoints' balls, and a per-node bound on the ball size
    // is a one-off pass over the CSR -- so they are counted ahead of and beside the main COUNT by 512-thread workgroups
    // on the high-priority stream; those that do come out LARGE-tier are written to fixed slots of a separate arena at once
    // and their tier kernel follows on the same stream, no host round trip in between.  The main COUNT still counts them
    // (same header values); the scan leaves them out of the arena and of the tier lists.  Whatever the prediction misses,
    // or the slots cannot hold, takes the ordinary path below.
    // COUNT writes the MID / MEDIUM vicinities itself at bump-allocated arena offsets (images only): the arena has to exist
    // before the sizes are known, so it starts from a guess and grows when a chunk overflows it (that chunk falls back to the
    // scan + FILL path below)
    // (round 4: tlc_vicinity_filtration as well -- its id / f / edge outputs have caller-given offsets, so COUNT can finish the
    // MID / MEDIUM vicinities too, ids included; the 64-thread FILL pass over a handful of MEDIUM pairs was 0.15 - 0.22 ms of a
    // 0.45 ms call on 4 096 Amazon-shaped pairs)
    const bool plain = pi_enabled && !d_out_ids && !d_out_f && !d_out_edges;
    const bool bump = true;
    unsigned long long* d_bump_top = reinterpret_cast<unsigned long long*>(ws->d_ctl + 20);
    int* d_bump_overflow = ws->d_ctl + 22;
    // The extraction runs from the ball lists (extract.hip); rounds 3 - 4: hop <= 2 only.  (Round 4: also with the
    // id / f / edge outputs of tlc_vicinity_filtration and with TLC_INCLUDE_ROOTS, the PDGNN fork's vicinities -- the breadth-first
    // COUNT pays two bitmaps of N bits per pair whatever the vicinity's size: 0.18 - 0.25 ms of a 0.37 ms call on 4 096 Amazon-shaped
    // pairs at hop 1.)
    // (its member bitmap of N bits lives in LDS: a graph beyond ~1 M nodes keeps the breadth-first kernels, whose bitmaps are in HBM)
    // (round 5: any hop -- the ball lists of hop >= 3 come from breadth-first levels; three bitmaps of N bits must fit the LDS then)
    bool use_x = g->opt_extract && g->x_lds512 <= (size_t)160 * 1024 && g->x_lds64 <= (size_t)160 * 1024 &&
                 (hop <= 2 || (size_t)g->nw * 12 + 16 <= (size_t)160 * 1024);
    if (use_x) {
        if ((rc = ensure_ball_lists(g, hop, s)) != TLC_OK) return rc;
        use_x = g->ball_list_hop == hop;
    }
    const bool early = pi_enabled && !d_out_ids && !d_out_f && !d_out_edges && (hop <= 2 || use_x) && n_pairs >= TLC_EARLY_MIN_PAIRS;
    // (2 048 .. 6 400 extraction workgroups measured within 2 %: one per scratch slot)
    const int xgrid = std::min(n_pairs, g->vic_slots);
    // The pairs whose vicinity is a filter over the smaller ball's subgraph list (extract.hip, x_sweep_ball: smaller ball <= 128
    // nodes, 93 % of a PubMed batch) get a launch of their own, tlc_extract_kernel<64, true>: no row sweep in it, so half the
    // registers, twice the wavefronts, 3.6 KB of LDS -- and it needs nothing from the classification, so it is the FIRST thing on the
    // chunk's main stream and runs beside the classification and the early pass, in front of the residency gate.  The general launch
    // behind the gate leaves those pairs alone (TlcVicParams::fast_split).
    const bool fsplit = use_x && g->opt_fast_split && g->opt_ball_edges && g->d_be_ptr != nullptr && !(flags & TLC_INCLUDE_ROOTS);
    const int xfgrid = fsplit ? std::min(n_pairs, 8192) : 0;        // (2 048 / 4 096 / 8 192 measured: the last by 1 - 2 %; round 6: 12 288 / 16 384 / one per pair: equal)
    long long bump_base = 0;
    if (use_x) {
        // arena = one region per workgroup of the extraction (main pass, then the early pass), then the bump area
        // (main pass, early pass, the launch of the subgraph-list pairs)
        const long long regions = (long long)xgrid + TLC_EARLY_WG + xfgrid;
        bump_base = regions * g->opt_x_region;
        const size_t want = (size_t)bump_base + std::max<size_t>(std::max<size_t>((size_t)n_pairs * 32, (size_t)g->opt_x_bump_min), ws->x_entries_hint + ws->x_entries_hint / 2);
        if (ws->cap_entries < want && (rc = ensure_arena(g, ws, want)) != TLC_OK) return rc;
        vp.small_dir = nullptr; vp.small_lw = nullptr;
        vp.bptr = g->d_bptr; vp.bcol = g->d_bcol;
        if (g->opt_ball_edges) { vp.be_ptr = g->d_be_ptr; vp.be_pos = g->d_be_pos; vp.be_w = g->d_be_w; }
        vp.nrec = g->d_nrec;
        if (g->hh_k > 0 && g->opt_heavy) { vp.hh_w = g->d_hh_w; vp.hh_k = g->hh_k; }
        vp.region_entries = g->opt_x_region; vp.bump_base = bump_base; vp.region_base_wg = 0;
    } else {
        if ((rc = ensure_small(g, ws, (size_t)n_pairs)) != TLC_OK) return rc;
        vp.small_dir = ws->S_dir; vp.small_lw = ws->S_lw;
    }
    if (bump) {
        if (ws->cap_entries == 0 &&
            (rc = ensure_arena(g, ws, std::min<size_t>((size_t)n_pairs * 128, (size_t)1 << 23))) != TLC_OK) return rc;
        vp.A_dir = ws->A_dir; vp.A_lw = ws->A_lw;
        vp.bump_top = d_bump_top; vp.bump_cap = (long long)ws->cap_entries; vp.bump_overflow = d_bump_overflow;
    }
    TlcPdParams& pp = c.pp;
    memset(&pp, 0, sizeof(pp));
    pp.hdr_n = ws->hdr_n; pp.hdr_m2 = ws->hdr_m2; pp.hdr_lu = ws->hdr_lu; pp.hdr_lv = ws->hdr_lv;
    pp.edge_off = ws->edge_off;
    pp.small_dir = use_x ? nullptr : ws->S_dir; pp.small_lw = use_x ? nullptr : ws->S_lw;
    pp.flags = flags; pp.res = res; pp.out_pi = d_out_pi; pp.out_status = d_out_status;
    pp.ids_off = (const long long*)d_ids_off; pp.out_f = d_out_f; pp.out_n = d_out_n; pp.pi_enabled = pi_enabled;
    pp.edges_off = (const long long*)d_edge_offs; pp.out_edges = d_out_edges; pp.out_m = d_out_m;
    pp.stats = ws->d_stats;
    pp.dc_force_fail = g->opt_dc_force_fail;
    pp.no_plain = g->opt_plain_kernels ? 0 : 1;
    c.bump = bump; c.use_x = use_x; c.early = early; c.bump_base = bump_base; c.xgrid = xgrid;
    c.plain = plain; c.fsplit = fsplit; c.xfgrid = xfgrid; c.pipelined = pipelined; c.spec = false;
    c.count_only = g->count_only != 0;
    return TLC_OK;
}

// the fork of the early chain and the launch of the subgraph-list pairs
static int front_fast(tlc_graph* g, Workspace* ws) {
    int rc;
    ChunkCtx& c = ws->ctx;
    TlcVicParams& vp = c.vp;
    hipStream_t s = c.s;
    const int hop = c.hop, xgrid = c.xgrid, xfgrid = c.xfgrid;
    const bool early = c.early, use_x = c.use_x, fsplit = c.fsplit;
    if (early) {
        // (the fork of the early chain: ahead of the FAST launch, which runs beside it; the one-off ball bounds go in front of it)
        if ((rc = ensure_early(g, ws, hop, s)) != TLC_OK) return rc;
        TLC_HIP_CHECK(hipEventRecord(ws->ev_fork, s));                  // after the memsets (and the one-off bounds)
        TLC_HIP_CHECK(hipStreamWaitEvent(ws->side[4], ws->ev_fork, 0));
    }
    // (the early pass's candidates -- smaller ball >= 511 nodes -- are not the subgraph-list launch's, in either extraction launch)
    vp.early_min_ball = (early && use_x) ? TLC_M_NMAX - 1 : 0;
    if (fsplit) {
        TlcVicParams fp = vp;                                 // (no bins, no early list, no work counter: every pair by index, statically strided)
        fp.region_base_wg = xgrid + TLC_EARLY_WG;
        fp.scratch_base_slot = 0;                             // (never used: its member lists fit the LDS)
        fp.work_counter = nullptr;
        // (round 6: its pairs dealt from work counters of its own instead of statically strided -- per-pair times spread from 4 to 21 us and a
        // workgroup takes 4.6 pairs, so the launch lasts 2.1 x its mean workgroup -- measured: chunks of 1 / 2 / 4 pairs +14 / +6.5 / +3.8 % per
        // pipelined batch: eight counters do not serve 37 676 dequeues in the 50 us the launch lasts.  What does help is the ORDER of the
        // pairs: tools/order_probe.py, the batch sorted by vicinity size descending: -3.2 %; the caller's order is the caller's.)
        size_t flds = g->x_lds64f;
        if (g->opt_ball_bits && g->d_bbits) {                 // (no bitmap of N bits in that launch's LDS: nw = 0 in its layout)
            fp.bbits = g->d_bbits; fp.bb_nw = g->nw; fp.nw = 0;
            flds = g->x_lds64fb;
        }
        if ((rc = tlc_launch_extract(64, xfgrid, flds, fp, s, true)) != TLC_OK) return rc;
        vp.fast_split = 1;
    }
    return TLC_OK;
}

// the early chain (classification, early pass, LARGE tier kernel) and the residency gate
static int front_early_chain(tlc_graph* g, Workspace* ws) {
    int rc;
    ChunkCtx& c = ws->ctx;
    TlcVicParams& vp = c.vp;
    TlcPdParams& pp = c.pp;
    hipStream_t s = c.s;
    const int n_pairs = c.n_pairs, hop = c.hop, xgrid = c.xgrid;
    const bool early = c.early, use_x = c.use_x;
    const int32_t* d_pairs = vp.pairs;
    // control words of the early pass (zeroed with the control block)
    int* d_cand_count = ws->d_ctl + 16;
    int* d_early_count = ws->d_ctl + 17;
    int* d_early_started = ws->d_ctl + 18;
    int* d_cand_started = ws->d_ctl + 19;
    if (early) {
        if ((rc = ensure_early(g, ws, hop, s)) != TLC_OK) return rc;
        hipStream_t es = ws->side[4];
        // (TLC_INCLUDE_ROOTS adds at most the two roots to a vicinity)
        if (use_x) {
            // exact ball sizes: the candidates of the early pass and the bins the main pass takes first
            if ((rc = tlc_launch_classify(n_pairs, d_pairs, g->n_nodes, g->d_bptr, TLC_M_NMAX - 1, TLC_EARLY_CAND, d_cand_count,
                                          ws->d_cand_list, ws->d_ctl + 32, ws->big_lists, es)) != TLC_OK) return rc;
            TLC_HIP_CHECK(hipEventRecord(ws->ev_cls, es));
        } else if ((rc = tlc_launch_select_heavy(n_pairs, d_pairs, g->n_nodes, g->d_ball_ub[(hop - 1) & 1], TLC_M_NMAX - 1, TLC_EARLY_CAND,
                                                 d_cand_count, ws->d_cand_list, es)) != TLC_OK) return rc;
        TlcVicParams ep = vp;
        ep.fill_mode = 1; ep.fill_list = ws->d_cand_list; ep.fill_count = TLC_EARLY_CAND; ep.work_count_dev = d_cand_count;
        ep.scratch_base_slot = g->vic_slots;
        ep.dbg = g->d_phase ? g->d_phase + 32 * TLC_N_TIERS : nullptr;        // (diagnostics: the early pass has its own row)
        ep.early_list = ws->d_early_list; ep.early_count = d_early_count; ep.early_cap = TLC_EARLY_SLOTS;
        ep.early_dir = ws->E_dir; ep.early_lw = ws->E_lw;
        ep.started = d_cand_started;
        if (use_x) {
            // the early pass OWNS its candidates: headers, status bytes, zero rows, slots and bump-allocated vicinities are all
            // its own, and the main COUNT leaves those pairs out (same predicate as the selection, see TlcVicParams::skip_ub)
            ep.region_base_wg = xgrid;
            if ((rc = tlc_launch_extract(512, TLC_EARLY_WG, g->x_lds512, ep, es)) != TLC_OK) return rc;
            vp.skip_threshold = TLC_M_NMAX - 1; vp.skip_count = d_cand_count; vp.skip_cap = TLC_EARLY_CAND;
            vp.big_count = ws->d_ctl + 32; vp.big_list = ws->big_lists;
        } else {
            ep.bump_top = nullptr;
            ep.out_pi = nullptr; ep.out_status = nullptr; ep.out_n = nullptr; ep.out_m = nullptr;   // the main COUNT reports
            ep.small_dir = nullptr; ep.small_lw = nullptr;
            hipLaunchKernelGGL((tlc_vicinity_kernel<false, 512>), dim3(TLC_EARLY_WG), dim3(512), g->vic_lds, es, ep);
            TLC_HIP_CHECK(hipGetLastError());
        }
        TLC_HIP_CHECK(hipEventRecord(ws->ev_early, es));
        TlcPdParams lp = pp;
        lp.tier_list = ws->d_early_list; lp.tier_count = TLC_EARLY_SLOTS; lp.tier_count_dev = d_early_count;
        lp.slot_entries = 2 * TLC_L_MMAX; lp.A_dir = ws->E_dir; lp.A_lw = ws->E_lw;
        lp.started = d_early_started;
        if (tlc_handoff_slot_bytes(TLC_TIER_LARGE) != 0) {
            if ((rc = ensure_handoff_large(g, ws, TLC_EARLY_SLOTS)) != TLC_OK) return rc;
            lp.handoff = ws->handoff_large; lp.handoff_stride = (long long)tlc_handoff_slot_bytes(TLC_TIER_LARGE);
            lp.handoff_cap = TLC_EARLY_SLOTS;
            dc_lists_for(ws, lp, 2);
            lp.dc_inplace = g->opt_dc_inplace;
        }
        lp.phase_cycles = g->d_phase ? g->d_phase + 32 * TLC_TIER_LARGE : nullptr;
        T0(5, es);
        if (((g->opt_tier_mask >> TLC_TIER_LARGE) & 1) && (rc = tlc_launch_pd_tier(TLC_TIER_LARGE, lp, es)) != TLC_OK) return rc;
        T1(5, es);
        TLC_HIP_CHECK(hipEventRecord(ws->ev_join[4], es));
    }
    return TLC_OK;
}

// ... and what the chunk's own stream waits for before its general launch: the classification (pipelined chunks) or the whole early
// pass and the LARGE workgroups' residency (a chunk on its own)
static int front_join_early(tlc_graph* g, Workspace* ws) {
    ChunkCtx& c = ws->ctx;
    hipStream_t s = c.s;
    const bool use_x = c.use_x;
    int* d_early_count = ws->d_ctl + 17;
    int* d_early_started = ws->d_ctl + 18;
    if (c.early) {
        // The workgroups of the main COUNT are persistent (each strides over its share of the pairs) and fill every wavefront
        // slot and most of the LDS of the machine: once they are running, a 512-thread workgroup of the early pass -- let alone
        // one of its tier kernel, which needs a whole CU's LDS -- is placed only as they drain (measured: the early tier kernel
        // then runs 1.05 instead of 0.88 ms because its last workgroups start ~0.15 ms late).  So the main COUNT is held until
        // the early COUNT is done and the early tier kernel's workgroups are resident (bounded: 50 us after the former).
        // (Measured and dropped, pipelined chunks: the main COUNT beside the early pass instead of behind it, and bounds of the gate
        // from none to 200 us: all within noise -- the machine is full of the previous chunk's tier kernels either way.)
#ifndef TLC_MAIN_BESIDE_EARLY
#define TLC_MAIN_BESIDE_EARLY 1
#endif
        // (Round 6, pipelined chunks: the general launch waits for the classification only -- its bins and the candidates' count are all it
        // needs from the early stream; a pair has one owner, so the two launches write disjoint headers, regions and slots -- and runs
        // BESIDE the early pass; the scan still waits for the early list.  Two libraries in turn, three rounds: 0.4957 -> 0.4847 ms per
        // pipelined batch (-2.2 %), rotating batches -2.9 %.  A chunk on its own keeps the order that gets its LARGE workgroups placed first.)
        if (TLC_MAIN_BESIDE_EARLY && c.pipelined && use_x) { TLC_HIP_CHECK(hipStreamWaitEvent(s, ws->ev_cls, 0)); }
        else TLC_HIP_CHECK(hipStreamWaitEvent(s, ws->ev_early, 0));
        // (Round 6: no gate for a pipelined chunk.  With other chunks' tier kernels on every CU the LARGE workgroups are never resident
        // within the bound, so the gate was a 50 us wait -- and a kernel of its own -- in the middle of every first half: in-region timeline
        // profiles/r06_queue_occupancy.txt.  Two libraries in turn, three rounds: 0.5048 -> 0.4930 ms per pipelined batch (-2.3 %).
        // -DTLC_GATE_PIPELINED=5000: as before.)
#ifndef TLC_GATE_PIPELINED
#define TLC_GATE_PIPELINED 0
#endif
        const long long gate = c.pipelined ? (long long)TLC_GATE_PIPELINED : 5000ll;   // 10 ns ticks
        if (gate > 0)
            hipLaunchKernelGGL(tlc_wait_started_dev, dim3(1), dim3(TLC_WAVE), 0, s, (const int*)d_early_started, (const int*)d_early_count,
                               192, gate);
        TLC_HIP_CHECK(hipGetLastError());
    }
    return TLC_OK;
}

// the general extraction launch and the scan (publishes the sizes; records the fork point of the tier launches)
static int front_main_scan(tlc_graph* g, Workspace* ws) {
    int rc;
    ChunkCtx& c = ws->ctx;
    TlcVicParams& vp = c.vp;
    hipStream_t s = c.s;
    const int n_pairs = c.n_pairs, xgrid = c.xgrid, vgrid = c.vgrid;
    const bool early = c.early, use_x = c.use_x, fsplit = c.fsplit, plain = c.plain, pipelined = c.pipelined, bump = c.bump;
    const long long bump_base = c.bump_base;
    int* d_early_count = ws->d_ctl + 17;
    unsigned long long* d_bump_top = reinterpret_cast<unsigned long long*>(ws->d_ctl + 20);
    int* d_bump_overflow = ws->d_ctl + 22;
    vp.work_counter = ws->d_ctl + 64 + TLC_SCAN_MAX_BLOCKS + 8;         // TLC_X_COUNTERS (8) counters, 64 ints apart, behind the statistics
    // (about one chunk per RESIDENT extraction wavefront -- 16 per CU, 4 096 -- when batches are pipelined: the machine is full of
    // other chunks' kernels then and the extraction's own tail costs nothing; twice as many for a lone batch.  In-process A/B,
    // tools/gpu_chunk_ab.sh, x_chunk_div 4096 against 8192 / 16384 / 32768: pipelined batch +0.4 / +2.2 / +2.7 %, latency of one
    // batch -1.9 / -1.1 / -1.7 %.  Half as many leaves half of the machine without a first chunk: +6 %)
    vp.work_chunk = std::max(2, n_pairs / (pipelined ? 4096 : 8192));
    // (behind a FAST launch the work list is the two top bins only -- a few thousand pairs of 20 - 80 us: one pair per chunk)
    if (fsplit && early) vp.work_chunk = 1;
    T0(0, s);
    if (use_x) {
        // (behind a FAST launch this one is left with the pairs whose smaller ball has more than 128 nodes -- a few thousand, 20 - 80 us
        // each on one wavefront.  512-thread workgroups for them instead -- 256 / 512 / 1024 of them -- measured: pipelined batch
        // 0.576 -> 0.62 - 0.63 ms, one batch alone 0.68 -> 0.74 ms; the count-prefix-write form of the sweep costs more than it spreads)
        // (round 6: 1 024 / 2 048 workgroups for it behind a FAST launch instead of one per scratch slot -- its work list is a few thousand
        // pairs, and in a pipelined region the launch takes 150 us for 12 us of work -- measured: 0.4962 / 0.4991 vs 0.4994 ms, within noise)
        if ((rc = tlc_launch_extract(64, xgrid, g->x_lds64, vp, s)) != TLC_OK) return rc;
    } else {
        hipLaunchKernelGGL((tlc_vicinity_kernel<false, 64>), dim3(vgrid), dim3(TLC_WAVE), g->vic_lds, s, vp);
    }
    T1(0, s);
    vp.work_counter = nullptr;
    vp.skip_count = nullptr; vp.big_count = nullptr;      // (the FILL launches below are list-driven)
    TLC_HIP_CHECK(hipGetLastError());
    if (early) TLC_HIP_CHECK(hipStreamWaitEvent(s, ws->ev_early, 0));   // the scan reads the early list

    // exclusive scan of the induced entry counts + tier binning
    const int nb = (n_pairs + TLC_SCAN_BLOCK - 1) / TLC_SCAN_BLOCK;
    T0(1, s);
    const unsigned seq = ++ws->pub_seq;
    TlcScanParams sp;
    memset(&sp, 0, sizeof(sp));                              // (every optional pointer null unless set below)
    sp.n_pairs = n_pairs; sp.hdr_n = ws->hdr_n; sp.hdr_m2 = ws->hdr_m2;
    sp.block_agg = ws->d_block_sums; sp.block_flag = ws->d_ctl + 64; sp.sync = ws->d_ctl + 10; sp.totals = ws->d_totals;
    sp.edge_off = ws->edge_off; sp.tier_count = ws->d_ctl; sp.tier_list = ws->tier_list; sp.small_arena = use_x ? 0 : 1;
    // the plain TLC-GNN image batch at resolution 5: the smallest vicinities go to the lane-per-subgraph kernel (pd_tiny.hip)
    // The MEDIUM tier's split by Pos-edge count exists for the LATENCY of one chunk: the vicinities with the longest serial swaps
    // are submitted behind the scan at once, so that chain starts 50 us earlier (0.79 -> 0.76 ms).  With chunks in flight on both
    // workspaces the machine is full either way and the split only costs: a second pair of kernels per chunk, and the MEDIUM chain's
    // hardest part ahead of the main stream's join (tools/ab_option.py mh_always 1 0, one process: rotating batches 0.718 -> 0.700 ms,
    // the fixed batch 0.729 vs 0.734; profiles/r03_threshold_sweep.txt has the curve over the cut).  So a pipelined chunk does not split.
    const bool mh_split = !pipelined;
    sp.mh_min_pos = mh_split ? TLC_MH_MIN_POS : 0x7fffffff;
    // (Round 6, pipelined plain chunks: the compact list's many-Pos vicinities still get a list of their own -- not for kernels of their own,
    // but to stand in FRONT of the compact MEDIUM launch (TlcPdParams::tier_list_hi): the swap kernel ends most batches and lasts as long as
    // its longest walk plus the time that walk's wavefront waited to be placed.  tools/order_probe.py: the MEDIUM list with its most-Pos
    // vicinities first, -2 % per pipelined batch.)
#ifndef TLC_MH_FRONT_POS
#define TLC_MH_FRONT_POS 64      /* (96: -0.9 %, 64: -1.2 ... -1.9 %, 48: about the same, 32 and 128: less; two libraries or four in turn) */
#endif
    c.mh_front = !mh_split && plain && TLC_MH_FRONT_POS > 0;
    if (c.mh_front) { sp.mh_min_pos = TLC_MH_FRONT_POS; sp.mh_compact_only = 1; }
    sp.tiny_ok = (g->opt_tiny && plain && vp.flags == 0u && vp.res == 5) ? 1 : 0;      // (plain: images and none of the filtration outputs)
    sp.dcm_count = ws->d_ctl + 44; sp.h_dcm = const_cast<int*>(&ws->h_sync_dev->pub_dcm);
    // (the TINY list by size class as well: d_ctl[48..63] count, zeroed with the control block; the scan's flags start at 64)
    static_assert(TLC_TINY_BINS <= 16, "the size-class counters of the TINY list live in d_ctl[48..63]");
    if (sp.tiny_ok) { sp.tiny_bin_count = ws->d_ctl + 48; sp.tiny_bin_list = ws->tiny_bins; sp.h_tiny_bins = const_cast<int*>(ws->h_sync_dev->pub_tiny); }
    sp.early_list = early ? ws->d_early_list : nullptr; sp.early_count = d_early_count; sp.early_cap = TLC_EARLY_SLOTS;
    sp.h_early = const_cast<int*>(&ws->h_sync_dev->pub_early);
    sp.bump_top = bump ? d_bump_top : nullptr; sp.bump_overflow = d_bump_overflow; sp.bump_base = bump_base;
    sp.h_overflow = const_cast<int*>(&ws->h_sync_dev->pub_overflow);
    // The arena size and the tier counts come back through mapped host memory: the last block of the scan stores them,
    // fences at system scope and bumps a sequence number the host polls -- no copy kernels, no stream synchronisation on
    // the critical path.  The poll gives up after 200 us and falls back to synchronising the stream (which also surfaces
    // a kernel fault).
    sp.h_total = const_cast<long long*>(&ws->h_sync_dev->pub_total);
    sp.h_tier = const_cast<int*>(ws->h_sync_dev->pub_tier);
    sp.h_seq = const_cast<unsigned*>(&ws->h_sync_dev->pub_seq);
    sp.seq = seq;
    hipLaunchKernelGGL(tlc_scan_bin, dim3(nb), dim3(TLC_SCAN_BLOCK), 0, s, sp);
    T1(1, s);
    TLC_HIP_CHECK(hipGetLastError());
    c.seq = seq;
    c.tiny_bins = sp.tiny_bin_count != nullptr;
    return TLC_OK;
}

// the speculative launch behind the scan
static int front_speculative(tlc_graph* g, Workspace* ws) {
    int rc;
    ChunkCtx& c = ws->ctx;
    TlcPdParams& pp = c.pp;
    hipStream_t s = c.s;
    const int n_pairs = c.n_pairs;
    const bool early = c.early, plain = c.plain, pipelined = c.pipelined;
    int* d_bump_overflow = ws->d_ctl + 22;
    // ---- speculative submission of the MID / MEDIUM tiers ---------------------------------------------------------------
    // Their inputs are complete once the scan has run (COUNT wrote the vicinities, the scan the tier lists), so they are
    // submitted behind it right away, with the list lengths on the device and grids / hand-off buffers sized from the
    // previous chunk: the ~50 us the host needs to see the published sizes and issue a dozen launch calls are no longer
    // between the scan and the batch's second-longest chain.  If COUNT overflowed the arena the kernels return at once
    // (abort flag) and the chunk is redone below.
    bool (&used)[TLC_N_SIDE] = c.used;
    for (int k = 0; k < TLC_N_SIDE; ++k) used[k] = (k == 4) && early;
    // The fork point of the side-stream launches that need nothing but the scan.  Recorded here, it is long complete when the
    // host has seen the sizes and submits them, and a wait on a complete event is no command at all -- an event recorded at
    // submission time costs every side stream a barrier packet on a signal that is still in flight: 60 us per chunk (measured:
    // 0.796 -> 0.733 ms per pipelined batch, tools/ab_option.py mh_always 0 3 before this was unconditional).
    TLC_HIP_CHECK(hipEventRecord(ws->ev_scan, s));
    const bool mh_split = !pipelined;                               // (front_main_scan: a pipelined chunk does not split the MEDIUM tier)
    const bool spec = plain && mh_split;
    size_t (&spec_base)[TLC_N_TIERS] = c.spec_base;
    int (&spec_cap)[TLC_N_TIERS] = c.spec_cap;
    for (int t = 0; t < TLC_N_TIERS; ++t) { spec_base[t] = 0; spec_cap[t] = 0; }
    if (spec) {
        spec_cap[TLC_TIER_MID] = std::min(n_pairs, std::max(4096, ws->prev_tc[TLC_TIER_MID] + ws->prev_tc[TLC_TIER_MID] / 4));
        spec_cap[TLC_TIER_MEDIUM] = std::min(n_pairs, std::max(2048, ws->prev_tc[TLC_TIER_MEDIUM] + ws->prev_tc[TLC_TIER_MEDIUM] / 4));
        spec_cap[TLC_TIER_MEDHI] = std::min(n_pairs, std::max(1024, ws->prev_tc[TLC_TIER_MEDHI] + ws->prev_tc[TLC_TIER_MEDHI] / 4));
        spec_cap[TLC_TIER_MEDWIDE] = 0;       // (the scan fills that list only when the split by Pos edges is off, i.e. never beside a speculative launch)
        if (g->opt_spec_cap > 0)                                        // (tests: reach the paths beyond the reserved slots)
            for (int t : {TLC_TIER_MID, TLC_TIER_MEDIUM, TLC_TIER_MEDHI}) spec_cap[t] = std::min(spec_cap[t], g->opt_spec_cap);
        // hand-off buffer: [MID | MEDHI (speculative launch) | MEDIUM | MEDWIDE]
        spec_base[TLC_TIER_MEDHI] = (size_t)spec_cap[TLC_TIER_MID] * tlc_handoff_slot_bytes(TLC_TIER_MID);
        spec_base[TLC_TIER_MEDIUM] = spec_base[TLC_TIER_MEDHI] + (size_t)spec_cap[TLC_TIER_MEDHI] * tlc_handoff_slot_bytes(TLC_TIER_MEDHI);
        spec_base[TLC_TIER_MEDWIDE] = spec_base[TLC_TIER_MEDIUM] + (size_t)spec_cap[TLC_TIER_MEDIUM] * tlc_handoff_slot_bytes(TLC_TIER_MEDIUM);
        if ((rc = ensure_handoff(g, ws, spec_base[TLC_TIER_MEDWIDE] +
                                        (size_t)spec_cap[TLC_TIER_MEDWIDE] * tlc_handoff_slot_bytes(TLC_TIER_MEDWIDE))) != TLC_OK) return rc;
        pp.A_dir = ws->A_dir; pp.A_lw = ws->A_lw;
        // On the caller's stream itself, tier kernels first, then their swap kernels.  (On side streams they would sit behind
        // event waits until the scan is done, and a blocked stream stalls whatever shares its hardware queue -- ROCm maps all
        // streams onto 4 by default: measured, the scan then started 0.1 ms late and took 55 instead of 11 us.)  The fork
        // point of the side-stream launches below is the event recorded here, ahead of these kernels.
        // (only the MEDIUM-sized vicinities with many Pos edges, whose tier kernel + long serial swaps are the longest chain of
        // the small tiers: kernels on one stream do not overlap, and another tier's pair of kernels between that tier kernel and
        // its swap kernel costs more than the host round trip saves -- measured)
        {
            const int t = TLC_TIER_MEDHI;
            pp.tier_list = ws->tier_list + (size_t)t * n_pairs; pp.tier_count = n_pairs; pp.tier_count_dev = ws->d_ctl + t;
            pp.grid = spec_cap[t]; pp.handoff_cap = spec_cap[t]; pp.phase = 0;
            pp.handoff = ws->handoff + spec_base[t]; pp.handoff_stride = (long long)tlc_handoff_slot_bytes(t);
            pp.abort_flag = d_bump_overflow;
            pp.phase_cycles = g->d_phase ? g->d_phase + 32 * t : nullptr;
            // (the divide and conquer in this chain only if the previous chunk had vicinities for it: the count is not known yet)
            pp.dc_count = nullptr; pp.dc_list = nullptr;
            if (ws->prev_dcm > 0) dc_lists_for(ws, pp, 0);
            T0(tslot[t], s);
            if (((g->opt_tier_mask >> t) & 1) && (rc = tlc_launch_pd_tier(t, pp, s)) != TLC_OK) return rc;
            T1(tslot[t], s);
            pp.dc_count = nullptr; pp.dc_list = nullptr;
        }
        pp.phase = 0;
        pp.grid = 0; pp.tier_count_dev = nullptr; pp.abort_flag = nullptr; pp.handoff = nullptr; pp.handoff_cap = 0;
    }
    c.spec = spec;
    return TLC_OK;
}

static int run_chunk_front(tlc_graph* g, Workspace* ws, const int32_t* d_pairs, int n_pairs, int hop, uint32_t flags, int res,
                     double* d_out_pi, uint8_t* d_out_status, const int64_t* d_ids_off, int32_t* d_out_ids,
                     double* d_out_f, int32_t* d_out_n, const int64_t* d_edge_offs, int32_t* d_out_edges, int32_t* d_out_m,
                     int pi_enabled, hipStream_t s, bool pipelined) {
    int rc;
    ChunkCtx& c = ws->ctx;
    if ((rc = front_prepare(g, ws, d_pairs, n_pairs, hop, flags, res, d_out_pi, d_out_status, d_ids_off, d_out_ids, d_out_f, d_out_n,
                            d_edge_offs, d_out_edges, d_out_m, pi_enabled, s, pipelined)) != TLC_OK) return rc;
    // (round 6: the early stream's launches submitted AHEAD of the subgraph-list launch, so that the classification is placed before that
    // launch's 8 192 workgroups ask for every wavefront slot: 0.4872 vs 0.4861 ms per pipelined batch, no difference)
    if ((rc = front_fast(g, ws)) != TLC_OK) return rc;
    if ((rc = front_early_chain(g, ws)) != TLC_OK) return rc;
    if ((rc = front_join_early(g, ws)) != TLC_OK) return rc;
    if ((rc = front_main_scan(g, ws)) != TLC_OK) return rc;
    if ((rc = front_speculative(g, ws)) != TLC_OK) return rc;
    c.ht_front = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - c.ht0).count() * 1e-3;
    ws->back_pending = 1;
    return TLC_OK;
}

// has the scan of the workspace's chunk published its sizes?
static inline bool sizes_published(const Workspace* ws) { return ws->h_sync->pub_seq == ws->ctx.seq; }

// a chunk's tier counts into the statistics of the call it belongs to (tlc_pd_pi_batch_stats, tlc_debug_tier_counts)
static void account_chunk(tlc_graph* g, const int (&tc)[TLC_N_TIERS], int n_early) {
    for (int t = 0; t <= TLC_TIER_HUGE; ++t) g->last_stats[t] += tc[t];
    for (int t = 0; t < TLC_N_TIERS; ++t) g->last_tc[t] += tc[t];
    g->last_stats[TLC_TIER_MEDIUM] += tc[TLC_TIER_MEDWIDE];            // (reported with MEDIUM)
    g->last_stats[TLC_TIER_MEDIUM] += tc[TLC_TIER_MEDHI];              // (reported with MEDIUM; on its own in [9])
    g->last_stats[9] += tc[TLC_TIER_MEDHI];
    g->last_stats[TLC_TIER_SMALL] += tc[TLC_TIER_TINY];                 // (reported with SMALL; on its own in [8])
    g->last_stats[8] += tc[TLC_TIER_TINY];
    g->last_stats[TLC_TIER_LARGE] += n_early;
    g->last_stats[7] += tc[TLC_TIER_MID];
    g->last_stats[6] += 1;
}

// The host's wait for the sizes a chunk's scan publishes (mapped memory, sequence number).  hipStreamSynchronize would also wait for
// the kernels submitted behind the scan; the stream is only queried, now and then, so that a fault surfaces instead of a spin.
static int wait_for_sizes(const Workspace* ws, hipStream_t s, unsigned seq) {
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (unsigned it = 1; !(seen = (ws->h_sync->pub_seq == seq)); ++it) {
        if ((it & 0x3ff) != 0) continue;
        const auto el = std::chrono::steady_clock::now() - t0;
        if (el < std::chrono::microseconds(300)) continue;
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) { seen = (ws->h_sync->pub_seq == seq); break; }
        if (q != hipErrorNotReady) { tlc_set_error(hipGetErrorString(q)); return TLC_ERR_HIP; }
        if (el > std::chrono::seconds(20)) break;
    }
    TLC_REQUIRE(seen, "size publication did not arrive");
    return TLC_OK;
}

// The second half of a chunk: waits (on the host) for the sizes the scan publishes, then submits every launch whose grid or
// buffers depend on them, and joins the side streams into the chunk's stream.
static int run_chunk_back(tlc_graph* g, Workspace* ws) {
    int rc;
    ChunkCtx& c = ws->ctx;
    // (development: TLC_HOST_TRACE=1 prints where the submitting thread spends a chunk -- front submitted, sizes seen, tiers submitted)
    static const bool host_trace = getenv("TLC_HOST_TRACE") != nullptr;
    const auto ht0 = c.ht0;
    auto ht_us = [&]() { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - ht0).count() * 1e-3; };
    const double ht_front = c.ht_front;
    double ht_seen = 0;
    TlcVicParams& vp = c.vp;
    TlcPdParams& pp = c.pp;
    hipStream_t s = c.s;
    const int n_pairs = c.n_pairs, hop = c.hop, pi_enabled = c.pi_enabled, xgrid = c.xgrid, vgrid = c.vgrid;
    const bool bump = c.bump, use_x = c.use_x, early = c.early, spec = c.spec;
    const long long bump_base = c.bump_base;
    const unsigned seq = c.seq;
    size_t (&spec_base)[TLC_N_TIERS] = c.spec_base;
    int (&spec_cap)[TLC_N_TIERS] = c.spec_cap;
    bool (&used)[TLC_N_SIDE] = c.used;
    ws->back_pending = 0;
    if ((rc = wait_for_sizes(ws, s, seq)) != TLC_OK) return rc;
    ht_seen = ht_us();
    std::atomic_thread_fence(std::memory_order_acquire);
    const long long total = ws->h_sync->pub_total;
    int tc[TLC_N_TIERS];
    for (int t = 0; t < TLC_N_TIERS; ++t) tc[t] = ws->h_sync->pub_tier[t];
    // (the many-Pos part of a pipelined chunk's compact MEDIUM list: one launch with the rest, in front of it)
    int n_hi = 0;
    if (c.mh_front) { n_hi = tc[TLC_TIER_MEDHI]; tc[TLC_TIER_MEDIUM] += n_hi; tc[TLC_TIER_MEDHI] = 0; }
    // (COUNT's writes stand unless the chunk overflowed the arena: then everything is laid out by the scan and written by FILL)
    const bool bumped = bump && ws->h_sync->pub_overflow == 0;
    if (use_x) ws->x_entries_hint = std::max(ws->x_entries_hint, (size_t)std::max<long long>(total - (bumped ? bump_base : 0), 0));
    if ((rc = ensure_arena(g, ws, (size_t)total, bumped ? (size_t)std::min<long long>(total, (long long)ws->cap_entries) : 0, s)) != TLC_OK) return rc;
    if (tc[TLC_TIER_HUGE] > 0 && (rc = ensure_huge(g, ws)) != TLC_OK) return rc;

    const int n_early = early ? ws->h_sync->pub_early : 0;
    const int n_dcm = ws->h_sync->pub_dcm;
    ws->prev_dcm = n_dcm;
    int todo = tc[0] + tc[1] + tc[2] + tc[3] + tc[4] + tc[5] + tc[6] + tc[7];
    if (c.count_only) {
        // tlc_vicinity_sizes: the headers are all it asks for (pp.out_n / out_m: the caller's arrays at this chunk's offset)
        if ((rc = tlc_launch_copy_sizes(n_pairs, ws->hdr_n, ws->hdr_m2, pp.out_n, pp.out_m, s)) != TLC_OK) return rc;
        todo = 0;
    }
    const bool spec_done = spec && bumped;          // the MID / MEDIUM tiers are already running
    ws->prev_tc[TLC_TIER_MID] = tc[TLC_TIER_MID]; ws->prev_tc[TLC_TIER_MEDIUM] = tc[TLC_TIER_MEDIUM]; ws->prev_tc[TLC_TIER_MEDHI] = tc[TLC_TIER_MEDHI];
    ws->prev_tc[TLC_TIER_MEDWIDE] = tc[TLC_TIER_MEDWIDE];
    if (todo > 0) {
        vp.A_dir = ws->A_dir; vp.A_lw = ws->A_lw;
        pp.A_dir = ws->A_dir; pp.A_lw = ws->A_lw;
        pp.huge_scratch = ws->huge_scratch; pp.huge_stride = (long long)ws->huge_stride;
        pp.huge_nmax = std::min(g->n_nodes, TLC_MAX_SUBGRAPH_NODES); pp.huge_mmax = (int)std::min<long long>(g->nnz / 2 + 1, TLC_MAX_SUBGRAPH_EDGES); pp.huge_slots = ws->huge_slots;
        pp.started = (int*)(ws->d_stats + 2);
        // The extraction ran out of arena: every vicinity below the heavy tiers is laid out by the scan and written by the
        // breadth-first FILL -- before any tier kernel may read it (that includes the SMALL tier, which has no slots of its own here)
        bool filled_all = false;
        if (use_x && !bumped && tc[0] + tc[1] + tc[4] + tc[5] + tc[6] + tc[7] > 0) {
            vp.fill_mode = (tc[TLC_TIER_LARGE] + tc[TLC_TIER_HUGE] > 0 || n_early > 0) ? 2 : 0; vp.fill_list = nullptr; vp.fill_count = 0;
            vp.x_fill = 1; vp.bump_top = nullptr; vp.work_count_dev = nullptr;
            if ((rc = tlc_launch_extract(64, xgrid, g->x_lds64, vp, s)) != TLC_OK) return rc;
            filled_all = true;
        }
        // hand-off slots (images only): the tiers with long serial tails run their cycle swap in a second, one-wavefront kernel
        size_t hand_base[TLC_N_TIERS] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (pi_enabled && !spec_done) {
            size_t hand_total = 0;
            for (int t = 0; t < TLC_N_TIERS; ++t) {
                if (t == TLC_TIER_LARGE) continue;                   // (its own buffer)
                hand_base[t] = hand_total;
                hand_total += (size_t)tc[t] * tlc_handoff_slot_bytes(t);
            }
            if ((rc = ensure_handoff(g, ws, hand_total)) != TLC_OK) return rc;
        } else if (pi_enabled) {
            // the speculative launch in flight owns its part of the buffer, which must not move: the MID tier's slots were
            // reserved in front of it and the MEDIUM tier's behind it, for spec_cap[] subgraphs each; beyond that a tier runs
            // its cycle swap itself
            hand_base[TLC_TIER_MID] = 0;
            hand_base[TLC_TIER_MEDIUM] = spec_base[TLC_TIER_MEDIUM];
            hand_base[TLC_TIER_MEDWIDE] = spec_base[TLC_TIER_MEDWIDE];
        }
        // the speculative launch had one workgroup per reserved slot: list positions beyond them get a launch of their own,
        // behind it on s (tier kernel only: without a slot a subgraph's cycle swap runs in the tier kernel itself)
        if (spec_done && tc[TLC_TIER_MEDHI] > spec_cap[TLC_TIER_MEDHI]) {
            const int t = TLC_TIER_MEDHI;
            pp.tier_list = ws->tier_list + (size_t)t * n_pairs; pp.tier_count = tc[t]; pp.tier_count_dev = nullptr;
            pp.wi_base = spec_cap[t]; pp.grid = tc[t] - spec_cap[t]; pp.handoff_cap = spec_cap[t]; pp.phase = 1;
            pp.handoff = ws->handoff + spec_base[t]; pp.handoff_stride = (long long)tlc_handoff_slot_bytes(t);
            pp.abort_flag = nullptr; pp.dc_count = nullptr; pp.dc_list = nullptr;
            pp.phase_cycles = g->d_phase ? g->d_phase + 32 * t : nullptr;
            if (((g->opt_tier_mask >> t) & 1) && (rc = tlc_launch_pd_tier(t, pp, s)) != TLC_OK) return rc;
            pp.wi_base = 0; pp.grid = 0; pp.phase = 0; pp.handoff = nullptr; pp.handoff_cap = 0;
        }
        // The tier kernels of all lists first, their second kernels (cycle swap / divide and conquer, same stream, behind a tier kernel
        // that runs > 100 us) afterwards: a launch costs the host 3 - 4 us, and with both kernels of a tier submitted together the last
        // list's tier kernel started 84 us after the scan had ended (profiles/r04_bench_timeline.txt).  Option split_launch, default on.
        // (per tier, not per stream: two lists may share a stream)
        struct PendingSwap { bool on; int k; bool timed; TlcPdParams pp; };
        PendingSwap pend[TLC_N_TIERS];
        for (int t = 0; t < TLC_N_TIERS; ++t) pend[t].on = false;
        auto finish_pending_swaps = [&]() -> int {
            for (int t = 0; t < TLC_N_TIERS; ++t) {
                if (!pend[t].on) continue;
                pend[t].on = false;
                const int k = pend[t].k;
                hipStream_t ss = ws->side[k];
                int r = ((g->opt_tier_mask >> t) & 1) ? tlc_launch_pd_tier(t, pend[t].pp, ss) : TLC_OK;
                if (r != TLC_OK) return r;
                if (pend[t].timed) T1(tslot[t], ss);
                TLC_HIP_CHECK(hipEventRecord(ws->ev_join[k], ss));
            }
            return TLC_OK;
        };
        // (`behind_s`: the launch depends on what was just submitted to s, e.g. a FILL; else only on the scan)
        auto launch_side = [&](int k, int t, bool behind_s = true) -> int {
            if (bumped && !behind_s) {
                TLC_HIP_CHECK(hipStreamWaitEvent(ws->side[k], ws->ev_scan, 0));    // (not behind the speculative kernels on s)
            } else {
                TLC_HIP_CHECK(hipEventRecord(ws->ev_fork, s));
                TLC_HIP_CHECK(hipStreamWaitEvent(ws->side[k], ws->ev_fork, 0));
            }
            pp.tier_list = ws->tier_list + (size_t)t * n_pairs; pp.tier_count = tc[t];
            pp.tier_list_hi = nullptr; pp.n_hi = 0;
            if (t == TLC_TIER_MEDIUM && n_hi > 0) { pp.tier_list_hi = ws->tier_list + (size_t)TLC_TIER_MEDHI * n_pairs; pp.n_hi = n_hi; }
            const size_t hs = pi_enabled ? tlc_handoff_slot_bytes(t) : 0;       // (0: the tier kernel runs the cycle swap itself -- SMALL, MID)
            pp.handoff = hs ? ws->handoff + hand_base[t] : nullptr;
            pp.handoff_stride = (long long)hs;
            pp.handoff_cap = (spec_done && (t == TLC_TIER_MID || t == TLC_TIER_MEDIUM || t == TLC_TIER_MEDWIDE)) ? std::min(tc[t], spec_cap[t]) : tc[t];
            pp.dc_count = nullptr; pp.dc_list = nullptr; pp.dc_inplace = 0;
            if (t == TLC_TIER_LARGE) { dc_lists_for(ws, pp, 1); pp.dc_inplace = g->opt_dc_inplace; }
            if ((t == TLC_TIER_MEDHI || t == TLC_TIER_MEDWIDE) && n_dcm > 0) dc_lists_for(ws, pp, 0);
            if (hs && t == TLC_TIER_LARGE) {
                // (the early launch may still be using the first TLC_EARLY_SLOTS slots: this launch takes the ones behind them)
                int r2 = ensure_handoff_large(g, ws, (size_t)TLC_EARLY_SLOTS + (size_t)tc[t]);
                if (r2 != TLC_OK) return r2;
                pp.handoff = ws->handoff_large + (size_t)TLC_EARLY_SLOTS * hs;
            }
            pp.grid = 0; pp.phase = 0; pp.tier_count_dev = nullptr; pp.abort_flag = nullptr;
            pp.phase_cycles = g->d_phase ? g->d_phase + 32 * t : nullptr;
            const bool timed = !(early && t == TLC_TIER_LARGE) && !(t == TLC_TIER_MEDIUM && spec) && t != TLC_TIER_MEDWIDE;   // (those slots time the early launch / MEDHI / MEDIUM)
            if (timed) T0(tslot[t], ws->side[k]);
            const bool two = hs && !(pp.flags & TLC_NO_EXT1) &&
                             (t == TLC_TIER_MEDIUM || t == TLC_TIER_MID || t == TLC_TIER_MEDHI || t == TLC_TIER_MEDWIDE);
            if (two) pp.phase = 1;                                                                      // (the tier kernel only)
            int r = ((g->opt_tier_mask >> t) & 1) ? tlc_launch_pd_tier(t, pp, ws->side[k]) : TLC_OK;   // (development: tiers timed alone)
            pp.phase = 0;
            if (r != TLC_OK) return r;
            used[k] = true;
            if (two) {
                pend[t].on = true; pend[t].k = k; pend[t].timed = timed; pend[t].pp = pp; pend[t].pp.phase = 2;
                return TLC_OK;
            }
            if (timed) T1(tslot[t], ws->side[k]);
            TLC_HIP_CHECK(hipEventRecord(ws->ev_join[k], ws->side[k]));
            return TLC_OK;
        };
        // 1. the heavy tiers first: their subgraphs are filled by a small early pass (8 wavefronts per pair) so that the
        //    long serial tails of the largest vicinities start as soon as possible and overlap everything else
        const int heavy = tc[TLC_TIER_LARGE] + tc[TLC_TIER_HUGE];
        // (the LARGE tier's hand-off slots BEFORE anything heavy is submitted: growing them frees the old buffer, and hipFree waits
        // for the device -- behind the HUGE tier kernel that was 6.3 ms in which the host submitted nothing: the long list of the
        // strong-scaling leg took 33 instead of 26 ms in every call that needed a few slots more than the one before)
        if (pi_enabled && tc[TLC_TIER_LARGE] > 0 && (rc = ensure_handoff_large(g, ws, (size_t)TLC_EARLY_SLOTS + (size_t)tc[TLC_TIER_LARGE])) != TLC_OK) return rc;
        T0(2, s);
        for (int t = TLC_TIER_HUGE; t >= TLC_TIER_LARGE; --t) {
            if (tc[t] <= 0) continue;
            vp.fill_mode = 1; vp.fill_list = ws->tier_list + (size_t)t * n_pairs; vp.fill_count = tc[t];
            // the heavy vicinities get 8 wavefronts each (hop <= 2), so that their many long CSR rows are in flight together
            if (use_x) {
                // (same entry order as the early pass's slots: rows do not depend on which way a vicinity took)
                vp.x_fill = 1; vp.bump_top = nullptr; vp.work_count_dev = nullptr;
                if ((rc = tlc_launch_extract(512, std::min(tc[t], TLC_EARLY_WG), g->x_lds512, vp, s)) != TLC_OK) return rc;
            } else if (hop <= 2)
                hipLaunchKernelGGL((tlc_vicinity_kernel<true, 512>), dim3(std::min(tc[t], g->vic_slots)), dim3(512), g->vic_lds, s, vp);
            else
                hipLaunchKernelGGL((tlc_vicinity_kernel<true, 64>), dim3(std::min(tc[t], g->vic_slots)), dim3(TLC_WAVE), g->vic_lds, s, vp);
            TLC_HIP_CHECK(hipGetLastError());
            if ((rc = launch_side(1, t)) != TLC_OK) return rc;
        }
        // A LARGE workgroup needs nearly all of a CU's LDS.  If the many small workgroups of the other tiers reach the CUs
        // first, the dispatcher cannot place it until those kernels drain (measured: the whole LARGE tier, the critical
        // path of the batch, starts ~0.35 ms late).  So the main stream -- and with it the SMALL fork and the MEDIUM fill --
        // is held until the LARGE workgroups report themselves resident; the wait is bounded (50 us).
        if (tc[TLC_TIER_LARGE] > 0) {
            hipLaunchKernelGGL(tlc_wait_started, dim3(1), dim3(TLC_WAVE), 0, s, (const int*)pp.started,
                               std::min(tc[TLC_TIER_LARGE], 192), 5000ll);
            TLC_HIP_CHECK(hipGetLastError());
        }
        // 0. the SMALL tier needs nothing more (its subgraphs were written by the COUNT pass); it is submitted after the
        //    heavy chain so that its many workgroups do not delay that chain's start
        // the lower end of the SMALL tier first: one lane per subgraph, a few hundred latency-bound wavefronts that need 36 KB of
        // LDS each -- they must find room before the other tiers' workgroups take it
        auto launch_tiny_small = [&]() -> int {
            if (tc[TLC_TIER_TINY] > 0) {
                if (bumped) { TLC_HIP_CHECK(hipStreamWaitEvent(ws->side[5], ws->ev_scan, 0)); }
                else { TLC_HIP_CHECK(hipEventRecord(ws->ev_fork, s)); TLC_HIP_CHECK(hipStreamWaitEvent(ws->side[5], ws->ev_fork, 0)); }
                pp.tier_list = ws->tier_list + (size_t)TLC_TIER_TINY * n_pairs; pp.tier_count = tc[TLC_TIER_TINY];
                // (by size class, largest first, as the scan binned it: a wavefront of that kernel waits for its slowest lane)
                pp.tiny_bin_list = nullptr;
                if (c.tiny_bins) {
                    pp.tiny_bin_list = ws->tiny_bins; pp.tiny_bin_stride = n_pairs;
                    for (int b = 0; b < TLC_TINY_BINS; ++b) pp.tiny_bin_cnt[b] = ws->h_sync->pub_tiny[b];
                }
                pp.handoff = nullptr; pp.handoff_stride = 0; pp.handoff_cap = 0; pp.grid = 0; pp.phase = 0;
                pp.tier_count_dev = nullptr; pp.abort_flag = nullptr;
                pp.dc_count = nullptr; pp.dc_list = nullptr;
                pp.phase_cycles = g->d_phase ? g->d_phase + 32 * TLC_TIER_TINY : nullptr;
                int r = ((g->opt_tier_mask >> TLC_TIER_TINY) & 1) ? tlc_launch_pd_tiny(pp, ws->side[5]) : TLC_OK;
                if (r != TLC_OK) return r;
                TLC_HIP_CHECK(hipEventRecord(ws->ev_join[5], ws->side[5]));
                used[5] = true;
            }
            if (tc[TLC_TIER_SMALL] > 0) return launch_side(0, TLC_TIER_SMALL, false);
            return TLC_OK;
        };
        // 2. the MEDIUM-sized tiers
        auto launch_medium_mid = [&]() -> int {
            int r;
            if (tc[TLC_TIER_MEDIUM] + tc[TLC_TIER_MEDHI] + tc[TLC_TIER_MEDWIDE] + tc[TLC_TIER_MID] > 0) {
                if (!bumped && !filled_all) {
                    vp.fill_mode = (heavy > 0 || n_early > 0) ? 2 : 0; vp.fill_list = nullptr; vp.fill_count = 0;
                    hipLaunchKernelGGL((tlc_vicinity_kernel<true, 64>), dim3(vgrid), dim3(TLC_WAVE), g->vic_lds, s, vp);
                    TLC_HIP_CHECK(hipGetLastError());
                }
                T1(2, s);
                // (the few vicinities beyond the compact configuration first: the largest of the MEDIUM-sized ones, the longest swaps)
                if (tc[TLC_TIER_MEDWIDE] > 0 && (r = launch_side(7, TLC_TIER_MEDWIDE, !bumped)) != TLC_OK) return r;
                if (!spec_done && tc[TLC_TIER_MEDHI] > 0 && (r = launch_side(2, TLC_TIER_MEDHI)) != TLC_OK) return r;
                // (round 6, again after the first half got shorter and the MEDIUM stream's queue reached 0.82 busy: a workspace's MEDIUM list on
                // the heavy tiers' stream or on the MEDWIDE stream: +0.7 ... +5 % per pipelined batch in every assignment tried)
                // (round 6: every other pipelined chunk with the MEDIUM list on the MID stream and the MID list on the MEDIUM stream -- the two
                // queues would then be 0.57 busy each instead of 0.82 / 0.42 -- measured: +3 % per pipelined batch)
                if (tc[TLC_TIER_MEDIUM] > 0 && (r = launch_side(6, TLC_TIER_MEDIUM, !bumped)) != TLC_OK) return r;
                if (tc[TLC_TIER_MID] > 0 && (r = launch_side(3, TLC_TIER_MID, !bumped)) != TLC_OK) return r;
            } else {
                T1(2, s);
            }
            return TLC_OK;
        };
        // Submission order.  A chunk on its own with the TINY list in size classes: MEDIUM / MID first -- their tier kernels have a
        // serial second kernel behind them and the lane-per-subgraph kernel, no longer the last to finish, only takes LDS from them
        // when it starts alongside (tools/ab_option.py medium_first 0 1: one batch alone 0.7605 -> 0.7325 ms, pipelined batches equal;
        // with the TINY kernel held back 110 us by an explicit sort it was 0.696).
        // (round 6: MEDIUM / MID first for pipelined chunks too: 0.4884 vs 0.4866 ms, no gain)
        if (!c.pipelined && c.tiny_bins) {
            if ((rc = launch_medium_mid()) != TLC_OK) return rc;
            if ((rc = launch_tiny_small()) != TLC_OK) return rc;
        } else {
            if ((rc = launch_tiny_small()) != TLC_OK) return rc;
            if ((rc = launch_medium_mid()) != TLC_OK) return rc;
        }
        if ((rc = finish_pending_swaps()) != TLC_OK) return rc;
    }
    for (int k = 0; k < TLC_N_SIDE; ++k)
        if (used[k]) TLC_HIP_CHECK(hipStreamWaitEvent(s, ws->ev_join[k], 0));
#undef T0
#undef T1
    if (c.call_seq == g->call_seq) account_chunk(g, tc, n_early);      // (a deferred second half submitted by a LATER call does not count into that call's statistics)
    if (host_trace) {
        static std::chrono::steady_clock::time_point last_end;
        const double gap = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(ht0 - last_end).count() * 1e-3;
        fprintf(stderr, "[tlc host] ws%d since-last-chunk %.0f us | front submitted %.0f | sizes seen %.0f | tiers submitted %.0f\n",
                (int)(ws - g->ws), gap, ht_front, ht_seen, ht_us());
        last_end = std::chrono::steady_clock::now();
    }
    return TLC_OK;
}

// statistics: induced directed entries of the chunk a (finished) workspace holds -- summed on the host when somebody asks
static int chunk_entries(Workspace* ws, long long* out) {
    std::vector<int> n((size_t)ws->n_pairs), m2((size_t)ws->n_pairs);
    *out = 0;
    if (ws->n_pairs <= 0) return TLC_OK;
    TLC_HIP_CHECK(hipMemcpy(n.data(), ws->hdr_n, n.size() * sizeof(int), hipMemcpyDeviceToHost));
    TLC_HIP_CHECK(hipMemcpy(m2.data(), ws->hdr_m2, m2.size() * sizeof(int), hipMemcpyDeviceToHost));
    long long e = 0;
    for (size_t k = 0; k < n.size(); ++k) if (n[k] > 0) e += m2[k];
    *out = e;
    return TLC_OK;
}

// submits the second half of the chunk whose first half is in flight (tlc_pd_pi_batch_async defers it, see run_batch)
static int finish_pending(tlc_graph* g) {
    Workspace* ws = g->pending;
    if (!ws) return TLC_OK;
    g->pending = nullptr;
    int rc = run_chunk_back(g, ws);
    if (rc != TLC_OK) return rc;
    TLC_HIP_CHECK(hipEventRecord(ws->ev_done, ws->ctx.s));
    return TLC_OK;
}

// the next workspace in turn; if a chunk is still in flight on it, the host waits for that chunk (two chunks ahead of the GPU
// is as far as a caller can run)
static int acquire_workspace(tlc_graph* g, Workspace** out, bool same) {
    // (`same`: a stream-ordered single chunk -- consecutive calls cannot overlap anyway, and staying on one workspace keeps its
    // arena, headers and lists warm in the Infinity Cache: alternating cost the back-to-back batch 0.09 ms)
    Workspace* ws = same ? &g->ws[0] : &g->ws[g->next_ws % (unsigned)g->opt_n_ws];
    if (!same) ++g->next_ws;
    if (!ws->main) {
        int prio_lo = 0, prio_hi = 0;
        hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        TLC_HIP_CHECK(hipStreamCreateWithPriority(&ws->main, hipStreamNonBlocking, (prio_lo + prio_hi) / 2));
        TLC_HIP_CHECK(hipStreamCreateWithPriority(&ws->side[4], hipStreamNonBlocking, prio_hi));
        ws->own_early = 1;
    }
    // (round 6, measured and not kept: an asynchronous batch that does NOT wait here -- everything it submits is ordered on the GPU behind the
    // previous chunk's join on the workspace's own main stream anyway, buffers that grow are freed by hipFree, which waits for the device,
    // and nothing of a finished call is read on the host -- was 1.5 - 2.5 % SLOWER per pipelined batch: the first half's packets then sit in
    // a hardware queue behind the join's waits, and that queue is shared with a tier stream)
    if (ws->busy) {
        // (development: TLC_HOST_TRACE=1 also prints how long the submitting thread waits here for the workspace's previous chunk)
        static const bool host_trace = getenv("TLC_HOST_TRACE") != nullptr;
        const auto w0 = std::chrono::steady_clock::now();
        TLC_HIP_CHECK(hipEventSynchronize(ws->ev_done));
        if (host_trace)
            fprintf(stderr, "[tlc host] ws%d waited %.0f us for its previous chunk\n", (int)(ws - g->ws),
                    (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count() * 1e-3);
        ws->busy = 0;
        if (ws->in_call) {                                    // statistics of a chunk of the call in progress
            unsigned long long tie = 0;
            TLC_HIP_CHECK(hipMemcpy(&tie, ws->d_stats, sizeof(tie), hipMemcpyDeviceToHost));
            g->acc_tie += (long long)tie;
            long long e = 0;
            int rc = chunk_entries(ws, &e);
            if (rc != TLC_OK) return rc;
            g->acc_entries += e;
            ws->in_call = 0;
        }
    }
    *out = ws;
    return TLC_OK;
}

// `join`: the caller's stream waits for the chunks before the call returns (the stream-ordered contract of tlc_pd_pi_batch);
// else they are left in flight (tlc_pd_pi_batch_async) until tlc_pd_pi_batch_join.
// An error in the middle of a batch (an allocation that failed, a launch the runtime refused): the chunks in flight are waited for
// as far as the device still answers, and the handle forgets them -- the next call starts from a clean state instead of finishing a
// chunk whose first half never ran (review of round 4).  Returns rc.
static int fail_batch(tlc_graph* g, int rc) {
    g->pending = nullptr;
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    for (int k = 0; k < TLC_N_WS; ++k) { g->ws[k].busy = 0; g->ws[k].in_call = 0; g->ws[k].back_pending = 0; }
    return rc;
}

static int run_batch(tlc_graph* g, const int32_t* d_pairs, int64_t n_pairs, int hop, uint32_t flags, int res,
                     double* d_out_pi, uint8_t* d_out_status, const int64_t* d_ids_off, int32_t* d_out_ids,
                     double* d_out_f, int32_t* d_out_n, const int64_t* d_edge_offs, int32_t* d_out_edges, int32_t* d_out_m,
                     int pi_enabled, void* stream, bool join) {
    TLC_REQUIRE(g != nullptr, "graph handle is null");
    TLC_REQUIRE(n_pairs >= 0, "n_pairs < 0");
    TLC_REQUIRE(hop >= 1 && hop <= 64, "hop must be in 1..64");
    TLC_REQUIRE(res >= 1 && res <= 8, "res must be in 1..8");
    TLC_REQUIRE(n_pairs == 0 || d_pairs != nullptr, "pairs is null");
    TLC_ON_DEVICE(g->device);
    memset(g->last_stats, 0, sizeof(g->last_stats));
    memset(g->last_tc, 0, sizeof(g->last_tc));
    g->acc_tie = 0; g->acc_entries = 0;
    ++g->call_seq;
    for (int k = 0; k < TLC_N_WS; ++k) g->ws[k].in_call = 0;
    hipStream_t s = (hipStream_t)stream;
    // a single chunk of a stream-ordered call runs on the caller's stream itself (no cross-stream hops in its latency);
    // otherwise every chunk runs on its workspace's own stream so that consecutive chunks overlap
    // (test hook: TLC_CHUNK_PAIRS_TEST in the environment cuts a list into smaller chunks so that small inputs reach the multi-chunk path)
    static const int64_t chunk_env = getenv("TLC_CHUNK_PAIRS_TEST") ? atoll(getenv("TLC_CHUNK_PAIRS_TEST")) : 0;
    const int64_t chunk_pairs = chunk_env > 0 ? std::min<int64_t>(chunk_env, TLC_CHUNK_PAIRS) : TLC_CHUNK_PAIRS;
    const bool inline_main = join && n_pairs <= chunk_pairs;
    // Deferred second halves (asynchronous batches and the chunks of a long list): a chunk's tier launches need the sizes its scan
    // publishes, ~0.3 ms after the chunk was submitted.  A host that waits for them before it submits the NEXT chunk's first half
    // (selection, early pass, extraction, scan: a 0.5 ms chain of its own) serialises the two chains: chunk period = first half +
    // host round trip.  So the next chunk's first half goes in first, and then the host waits for this chunk's sizes: the first
    // half of chunk i+1 runs under the tier kernels of chunk i.  The last chunk's second half is submitted by the join
    // (tlc_pd_pi_batch_join, or the end of this call).
    const bool defer = !inline_main;
    int rc;
    if (!defer && (rc = finish_pending(g)) != TLC_OK) return fail_batch(g, rc);
    for (int64_t off = 0; off < n_pairs; off += chunk_pairs) {
        const int cnt = (int)std::min<int64_t>(chunk_pairs, n_pairs - off);
        Workspace* ws = nullptr;
        // (sizes already there: nothing to wait for, and the tier kernels should not queue behind another extraction)
        if (g->pending && sizes_published(g->pending) && (rc = finish_pending(g)) != TLC_OK) return fail_batch(g, rc);
        rc = acquire_workspace(g, &ws, inline_main);
        if (rc != TLC_OK) return fail_batch(g, rc);
        hipStream_t m = inline_main ? s : ws->main;
        if (!inline_main) {
            if (hipEventRecord(ws->ev_in, s) != hipSuccess || hipStreamWaitEvent(m, ws->ev_in, 0) != hipSuccess) {
                tlc_set_error("run_batch: %s", hipGetErrorString(hipGetLastError()));
                return fail_batch(g, TLC_ERR_HIP);
            }
        }
        // NOTE: ids_off is indexed by the global pair index, the kernels index by chunk-local index
        rc = run_chunk_front(g, ws, d_pairs + 2 * off, cnt, hop, flags, res,
                       d_out_pi ? d_out_pi + (size_t)off * res * res : nullptr,
                       d_out_status ? d_out_status + off : nullptr,
                       d_ids_off ? d_ids_off + off : nullptr, d_out_ids, d_out_f,
                       d_out_n ? d_out_n + off : nullptr, d_edge_offs ? d_edge_offs + off : nullptr, d_out_edges,
                       d_out_m ? d_out_m + off : nullptr, pi_enabled, m, !inline_main);
        if (rc != TLC_OK) return fail_batch(g, rc);
        ws->busy = 1; ws->in_call = 1; ws->n_pairs = cnt;
        g->last_ws = ws;
        if ((rc = finish_pending(g)) != TLC_OK) return fail_batch(g, rc);      // the previous chunk's second half (none unless deferred)
        g->pending = ws;
        if (!defer && (rc = finish_pending(g)) != TLC_OK) return fail_batch(g, rc);
    }
    if (join && (rc = finish_pending(g)) != TLC_OK) return fail_batch(g, rc);
    // tlc_pd_pi_batch == async + join: the caller's stream also waits for asynchronous batches still in flight on the other
    // workspaces (a single stream-ordered chunk runs on `s` itself; a wait on a complete event is no command at all)
    if (join)
        for (int k = 0; k < TLC_N_WS; ++k)
            if (g->ws[k].busy && !(inline_main && &g->ws[k] == g->last_ws)) TLC_HIP_CHECK(hipStreamWaitEvent(s, g->ws[k].ev_done, 0));
    return TLC_OK;
}

extern "C" int tlc_pd_pi_batch(tlc_graph* g, const int32_t* d_pairs, int64_t n_pairs, int hop, uint32_t flags, int res,
                               double* d_out_pi, uint8_t* d_out_status, void* stream) {
    TLC_REQUIRE(n_pairs == 0 || d_out_pi != nullptr, "out_pi is null");
    if (flags & TLC_NO_NORM) {
        tlc_set_error("tlc_pd_pi_batch: TLC_NO_NORM is not supported by the fused image stage (it assumes filtration values in "
                      "[0, 1]); use tlc_vicinity_filtration + tlc_pd_from_filtration + tlc_pi_raster");
        return TLC_ERR_UNSUPPORTED;
    }
    return run_batch(g, d_pairs, n_pairs, hop, flags, res, d_out_pi, d_out_status, nullptr, nullptr, nullptr, nullptr, nullptr,
                     nullptr, nullptr, 1, stream, true);
}

// tlc_pd_pi_batch without the final join: the batch is ordered AFTER what `stream` holds at the time of the call (its inputs may
// be produced there), but `stream` does not wait for it -- batches submitted back to back overlap, each on one of the handle's
// workspaces.  Outputs are complete for work that follows a tlc_pd_pi_batch_join on its stream.  At most three batches run ahead
// of the GPU: submitting a fourth waits on the host for the first.  (The batch's second half is deferred, see run_batch.)
extern "C" int tlc_pd_pi_batch_async(tlc_graph* g, const int32_t* d_pairs, int64_t n_pairs, int hop, uint32_t flags, int res,
                                     double* d_out_pi, uint8_t* d_out_status, void* stream) {
    TLC_REQUIRE(n_pairs == 0 || d_out_pi != nullptr, "out_pi is null");
    if (flags & TLC_NO_NORM) {
        tlc_set_error("tlc_pd_pi_batch_async: TLC_NO_NORM is not supported by the fused image stage");
        return TLC_ERR_UNSUPPORTED;
    }
    return run_batch(g, d_pairs, n_pairs, hop, flags, res, d_out_pi, d_out_status, nullptr, nullptr, nullptr, nullptr, nullptr,
                     nullptr, nullptr, 1, stream, false);
}

// makes `stream` wait for every batch submitted on this handle so far (asynchronous: nothing is waited for on the host)
extern "C" int tlc_pd_pi_batch_join(tlc_graph* g, void* stream) {
    TLC_REQUIRE(g != nullptr, "graph handle is null");
    TLC_ON_DEVICE(g->device);
    int rc = finish_pending(g);
    if (rc != TLC_OK) return rc;
    for (int k = 0; k < TLC_N_WS; ++k)
        if (g->ws[k].busy) TLC_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, g->ws[k].ev_done, 0));
    return TLC_OK;
}

extern "C" int tlc_vicinity_filtration(tlc_graph* g, const int32_t* d_pairs, int64_t n_pairs, int hop, uint32_t flags,
                                       const int64_t* d_node_offs, int32_t* d_out_ids, double* d_out_f, int32_t* d_out_n,
                                       uint8_t* d_out_status, const int64_t* d_edge_offs, int32_t* d_out_edges,
                                       int32_t* d_out_m, void* stream) {
    TLC_REQUIRE(d_node_offs && d_out_ids && d_out_f && d_out_n, "null output");
    const int ne = (d_edge_offs != nullptr) + (d_out_edges != nullptr) + (d_out_m != nullptr);
    TLC_REQUIRE(ne == 0 || ne == 3, "edge_offs / out_edges / out_m must be given together");
    return run_batch(g, d_pairs, n_pairs, hop, flags, 5, nullptr, d_out_status, d_node_offs, d_out_ids, d_out_f, d_out_n,
                     d_edge_offs, d_out_edges, d_out_m, 0, stream, true);
}

extern "C" int tlc_pd_pi_batch_stats(tlc_graph* g, int64_t* h_out, void* stream) {
    TLC_REQUIRE(g && h_out, "null argument");
    TLC_ON_DEVICE(g->device);
    { int rc = finish_pending(g); if (rc != TLC_OK) return rc; }
    TLC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    long long tie = g->acc_tie, entries = g->acc_entries;
    for (int k = 0; k < TLC_N_WS; ++k) {
        Workspace* ws = &g->ws[k];
        if (!ws->in_call) continue;
        if (ws->busy) { TLC_HIP_CHECK(hipEventSynchronize(ws->ev_done)); ws->busy = 0; }   // (an asynchronous call not joined yet)
        unsigned long long v = 0;
        TLC_HIP_CHECK(hipMemcpy(&v, ws->d_stats, sizeof(v), hipMemcpyDeviceToHost));
        tie += (long long)v;
        long long e = 0;
        int rc = chunk_entries(ws, &e);
        if (rc != TLC_OK) return rc;
        entries += e;
    }
    for (int k = 0; k < 10; ++k) h_out[k] = g->last_stats[k];
    h_out[4] = entries;
    h_out[5] += tie;
    return TLC_OK;
}

// ---- tlc_pd_from_filtration ----------------------------------------------------------------------------------------
extern "C" int tlc_pd_from_filtration(int32_t n_graphs, const int64_t* d_node_offs, const int64_t* d_edge_offs,
                                      const int32_t* d_edges, const double* d_f, uint32_t flags, double* d_pd_up,
                                      double* d_pd_down, double* d_pd_one, double* d_ext0, int32_t* d_counts,
                                      int32_t* d_edge_rank, void* stream) {
    TLC_REQUIRE(n_graphs >= 0, "n_graphs < 0");
    if (n_graphs == 0) return TLC_OK;
    TLC_REQUIRE(d_node_offs && d_edge_offs && d_f && d_pd_up && d_pd_down && d_pd_one && d_ext0 && d_counts, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    int* d_tier = nullptr;   // [4 counts | 4*n lists]
    TLC_HIP_CHECK(hipMalloc(&d_tier, ((size_t)TLC_N_TIERS * n_graphs + 8) * sizeof(int)));
    int rc = TLC_OK;
    int tc[TLC_N_TIERS] = {0, 0, 0, 0, 0};
    long long tail[2] = {0, 0};
    unsigned char* huge = nullptr;
    do {
        if (hipMemsetAsync(d_tier, 0, 8 * sizeof(int), s) != hipSuccess) { rc = TLC_ERR_HIP; break; }
        if ((rc = tlc_launch_pdf_bin(n_graphs, (const long long*)d_node_offs, (const long long*)d_edge_offs, d_counts, d_tier, d_tier + 8, s))) break;
        if (hipMemcpyAsync(tc, d_tier, sizeof(tc), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = TLC_ERR_HIP; break; }
        if (hipStreamSynchronize(s) != hipSuccess) { rc = TLC_ERR_HIP; break; }
        TlcPdfParams p;
        memset(&p, 0, sizeof(p));
        p.node_offs = (const long long*)d_node_offs; p.edge_offs = (const long long*)d_edge_offs;
        p.edges = d_edges; p.f = d_f; p.flags = flags;
        p.pd_up = d_pd_up; p.pd_down = d_pd_down; p.pd_one = d_pd_one; p.ext0 = d_ext0; p.counts = d_counts;
        p.edge_rank = d_edge_rank;
        if (tc[TLC_TIER_HUGE] > 0) {
            // size the scratch by the largest graph: bounded by the totals
            if (hipMemcpy(&tail[0], d_node_offs + n_graphs, sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(&tail[1], d_edge_offs + n_graphs, sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) { rc = TLC_ERR_HIP; break; }
            p.huge_nmax = (int)std::min<long long>(tail[0], 65535);
            p.huge_mmax = (int)std::min<long long>(tail[1] + 1, TLC_MAX_SUBGRAPH_EDGES);
            p.huge_stride = (long long)tlc_huge_slot_bytes(p.huge_nmax, p.huge_mmax);
            p.huge_slots = std::min(tc[TLC_TIER_HUGE], 32);
            if (hipMalloc(&huge, (size_t)p.huge_stride * p.huge_slots) != hipSuccess) { tlc_set_error("HUGE scratch alloc failed"); rc = TLC_ERR_OUT_OF_MEMORY; break; }
            p.huge_scratch = huge;
        }
        for (int t = TLC_N_TIERS - 1; t >= 0 && rc == TLC_OK; --t) {
            p.list = d_tier + 8 + (size_t)t * n_graphs; p.count = tc[t];
            rc = tlc_launch_pdf_tier(t, p, s);
        }
    } while (0);
    if (rc == TLC_ERR_HIP) tlc_set_error("tlc_pd_from_filtration: HIP runtime call failed: %s", hipGetErrorString(hipGetLastError()));
    // the temporaries must outlive the kernels
    hipStreamSynchronize(s);
    hipFree(d_tier);
    if (huge) hipFree(huge);
    return rc;
}

extern "C" int tlc_pi_raster(int32_t n_dgms, const int64_t* d_offs, const double* d_pts, int res, double* d_out,
                             void* stream) {
    TLC_REQUIRE(n_dgms >= 0, "n_dgms < 0");
    TLC_REQUIRE(res >= 1 && res <= 8, "res must be in 1..8");
    if (n_dgms == 0) return TLC_OK;
    TLC_REQUIRE(d_offs && d_out, "null pointer");
    return tlc_launch_pi_raster(n_dgms, (const long long*)d_offs, d_pts, res, d_out, stream);
}

// |S| and the induced edge count of every pair's vicinity, nothing else: the extraction and the scan of tlc_vicinity_filtration without
// its tier kernels.  A caller sizes exact offsets from them (tlc_pack_offsets) and calls tlc_vicinity_filtration with those -- no
// per-pair capacity to guess.  d_n / d_m int32[n_pairs] (n = 0 for a pair without a vicinity; same flags as the call that follows).
extern "C" int tlc_vicinity_sizes(tlc_graph* g, const int32_t* d_pairs, int64_t n_pairs, int hop, uint32_t flags, int32_t* d_n, int32_t* d_m,
                                  void* stream) {
    TLC_REQUIRE(g != nullptr, "graph handle is null");
    TLC_REQUIRE(n_pairs == 0 || (d_n && d_m), "null output");
    g->count_only = 1;
    const int rc = run_batch(g, d_pairs, n_pairs, hop, flags, 5, nullptr, nullptr, nullptr, nullptr, nullptr, d_n, nullptr, nullptr, d_m,
                             0, stream, true);
    g->count_only = 0;
    return rc;
}

// gradient of tlc_pi_raster's images with respect to the points, as the reference's differentiable imager defines it (through the
// weights only: pd_pipeline.hip, tlc_pi_raster_wgrad_kernel)
extern "C" int tlc_pi_raster_wgrad(int32_t n_dgms, int64_t n_pts, const int64_t* d_offs, const double* d_pts, int res,
                                   const double* d_grad_img, double* d_grad_pts, void* stream) {
    TLC_REQUIRE(n_dgms >= 0 && n_pts >= 0, "negative count");
    TLC_REQUIRE(res >= 1 && res <= 8, "res must be in 1..8");
    if (n_pts == 0) return TLC_OK;
    TLC_REQUIRE(n_dgms > 0 && d_offs && d_pts && d_grad_img && d_grad_pts, "null pointer");
    return tlc_launch_pi_raster_wgrad(n_dgms, (long long)n_pts, (const long long*)d_offs, d_pts, res, d_grad_img, d_grad_pts, stream);
}

// ---- diagnostics (declared in include/tlcgnn.h, "diagnostics" section): per-phase cycle counters of the PD tier kernels --
// Rows of 32 u64: one per tier (TLC_N_TIERS; the HUGE tier's row doubles as the main COUNT pass's), then the early pass, then the
// lane-per-pair extraction.
// `cap_u64` = how many u64 the caller allocated at h_out: never more than that is written; *n_rows (may be null) = rows the
// library keeps, so a tool can size its buffer by asking first (h_out null).
extern "C" int tlc_debug_phase_profile(tlc_graph* g, int enable, unsigned long long* h_out, int64_t cap_u64, int32_t* n_rows) {
    TLC_REQUIRE(g != nullptr, "null graph");
    TLC_REQUIRE(h_out == nullptr || cap_u64 >= 0, "cap_u64 < 0");
    TLC_ON_DEVICE(g->device);
    { int rc_p = finish_pending(g); if (rc_p != TLC_OK) return rc_p; }
    const size_t rows = (size_t)TLC_N_TIERS + 2;      // (the last row: the lane-per-pair extraction)
    if (n_rows) *n_rows = (int32_t)rows;
    TLC_HIP_CHECK(hipDeviceSynchronize());
    if (h_out && g->d_phase) {
        const size_t k = std::min<size_t>(rows * 32, (size_t)cap_u64);
        TLC_HIP_CHECK(hipMemcpy(h_out, g->d_phase, k * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (enable && !g->d_phase) TLC_HIP_CHECK(hipMalloc(&g->d_phase, rows * 32 * sizeof(unsigned long long)));
    if (g->d_phase) TLC_HIP_CHECK(hipMemset(g->d_phase, 0, rows * 32 * sizeof(unsigned long long)));
    if (!enable && g->d_phase) { hipFree(g->d_phase); g->d_phase = nullptr; }
    return TLC_OK;
}

// diagnostics (PHASE_DEBUG builds): wall-clock stamps (100 MHz ticks) of the extraction per pair of the last chunk, [n][8]
extern "C" int tlc_debug_pair_times(tlc_graph* g, unsigned long long* h_out, int64_t n_pairs) {
    TLC_REQUIRE(g && h_out, "null argument");
    TLC_ON_DEVICE(g->device);
    { int rc_p = finish_pending(g); if (rc_p != TLC_OK) return rc_p; }
    TLC_HIP_CHECK(hipDeviceSynchronize());
    const size_t k = std::min<size_t>((size_t)std::max<int64_t>(n_pairs, 0), g->d_pair_t ? g->cap_pair_t : 0);
    if (k) TLC_HIP_CHECK(hipMemcpy(h_out, g->d_pair_t, k * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return TLC_OK;
}

// development / test switches of one handle: "extract" (ball-list extraction, extract.hip), "heavy" (its heavy-row skipping),
// "tiny" (lane-per-subgraph kernel, pd_tiny.hip); 1 = on (default), 0 = off.  "x_region" / "x_bump_min": arena entries of a
// workgroup's region / of the bump area behind the regions (tests shrink them to reach the overflow path).  Results must not
// depend on any of them.
extern "C" int tlc_debug_set_option(tlc_graph* g, const char* name, int value) {
    TLC_REQUIRE(g && name, "null argument");
    if (!strcmp(name, "extract")) g->opt_extract = value != 0;
    else if (!strcmp(name, "heavy")) g->opt_heavy = value != 0;
    else if (!strcmp(name, "tiny")) g->opt_tiny = value != 0;
    else if (!strcmp(name, "ball_edges")) g->opt_ball_edges = value != 0;
    else if (!strcmp(name, "dc_inplace")) g->opt_dc_inplace = value != 0;
    else if (!strcmp(name, "fast_split")) g->opt_fast_split = value != 0;
    else if (!strcmp(name, "ball_bits")) g->opt_ball_bits = value != 0;
    else if (!strcmp(name, "plain_kernels")) g->opt_plain_kernels = value != 0;
    else if (!strcmp(name, "tier_mask")) g->opt_tier_mask = value;
    else if (!strcmp(name, "spec_cap")) g->opt_spec_cap = std::max(value, 0);
    else if (!strcmp(name, "n_ws")) { int rc_p = finish_pending(g); if (rc_p != TLC_OK) return rc_p; g->opt_n_ws = std::min(std::max(value, 2), TLC_N_WS); }
    else if (!strcmp(name, "timing_every")) { g->opt_timing_every = std::max(value, 1); g->timing_seq = 0; }
    else if (!strcmp(name, "dc_force_fail")) g->opt_dc_force_fail = value != 0;
    else if (!strcmp(name, "x_arena")) { g->opt_x_region = value > 0 ? value : TLC_X_REGION; g->opt_x_bump_min = value > 0 ? value : (1 << 20); }
    else { tlc_set_error("tlc_debug_set_option: unknown option '%s'", name); return TLC_ERR_INVALID_ARG; }
    return TLC_OK;
}

// diagnostics: subgraphs of the last chunk whose cycle swap ran as a divide and conquer (h_out[0]) / fell back to the serial walk
// after trying (h_out[1])
extern "C" int tlc_debug_dc_stats(tlc_graph* g, long long* h_out, void* stream) {
    TLC_REQUIRE(g && h_out, "null argument");
    TLC_ON_DEVICE(g->device);
    { int rc_p = finish_pending(g); if (rc_p != TLC_OK) return rc_p; }
    TLC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    unsigned long long v[4] = {0, 0, 0, 0};
    if (g->last_ws) {
        if (g->last_ws->busy) { TLC_HIP_CHECK(hipEventSynchronize(g->last_ws->ev_done)); g->last_ws->busy = 0; }
        TLC_HIP_CHECK(hipMemcpy(v, g->last_ws->d_stats, sizeof(v), hipMemcpyDeviceToHost));
    }
    h_out[0] = (long long)v[1];
    h_out[1] = (long long)v[3];
    return TLC_OK;
}

// diagnostics: the tier lists of the last call as the scan cut them (tlc_kernels.h: TLC_TIER_*), h_out[TLC_N_TIERS = 8]:
// small, medium (compact configuration, few Pos edges), large, huge, mid, tiny, medium with many Pos edges, medium beyond the compact
// configuration.  (tlc_pd_pi_batch_stats reports tiny with small and the three medium lists together.)
extern "C" int tlc_debug_tier_counts(tlc_graph* g, long long* h_out, void* stream) {
    TLC_REQUIRE(g && h_out, "null argument");
    TLC_ON_DEVICE(g->device);
    { int rc_p = finish_pending(g); if (rc_p != TLC_OK) return rc_p; }
    TLC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    for (int t = 0; t < TLC_N_TIERS; ++t) h_out[t] = g->last_tc[t];
    return TLC_OK;
}


// ---- measurement helpers (declared in include/tlcgnn.h) ----------------------------------------------------------------
extern "C" int tlc_pd_pi_batch_set_timing(tlc_graph* g, int enable) {
    TLC_REQUIRE(g != nullptr, "null graph");
    // 0: off; 1: every kernel; else bit k+1 selects timing slot k (sixteen timed event records per batch cost the PubMed
    // batch 48 us, 4.5 %: bench.py times everything during warm-up and only the dominant kernel inside the timed region)
    g->timing = enable == 1 ? 0xff : (enable > 1 ? (enable >> 1) & 0xff : 0);
    return TLC_OK;
}

// h_ms[0..7] = COUNT, scan+binning, FILL, tier SMALL, MEDIUM, LARGE, HUGE, MEDIUM's 128-thread sub-tier: kernel durations (ms, -1 = not launched) of the
// LAST chunk of the last tlc_pd_pi_batch, from HIP events recorded on the stream each kernel ran on.  Synchronises.
extern "C" int tlc_pd_pi_batch_timings(tlc_graph* g, double* h_ms, void* stream) {
    TLC_REQUIRE(g && h_ms, "null argument");
    TLC_ON_DEVICE(g->device);
    { int rc_p = finish_pending(g); if (rc_p != TLC_OK) return rc_p; }
    TLC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    for (int k = 0; k < 8; ++k) {
        h_ms[k] = -1.0;
        if (g->timing && g->ev_used && g->ev_used[k]) {
            TLC_HIP_CHECK(hipEventSynchronize(g->ev_t[2 * k + 1]));
            float ms = 0.f;
            TLC_HIP_CHECK(hipEventElapsedTime(&ms, g->ev_t[2 * k], g->ev_t[2 * k + 1]));
            h_ms[k] = (double)ms;
        }
    }
    return TLC_OK;
}

// Durations (ms) of timing slot `slot` (the order of tlc_pd_pi_batch_timings) over the most recent chunks, oldest first:
// at most `cap` and at most TLC_TIMING_RING (64) of them; -1 where the kernel was not launched or not selected.
// *n_out = how many were written.  Synchronises the stream; nothing was synchronised while the chunks ran.
extern "C" int tlc_pd_pi_batch_timing_history(tlc_graph* g, int slot, double* h_ms, int32_t cap, int32_t* n_out, void* stream) {
    TLC_REQUIRE(g && h_ms && n_out, "null argument");
    TLC_REQUIRE(slot >= 0 && slot < 8 && cap >= 0, "slot must be in 0..7");
    TLC_ON_DEVICE(g->device);
    { int rc_p = finish_pending(g); if (rc_p != TLC_OK) return rc_p; }
    TLC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    const unsigned long long have = std::min<unsigned long long>(g->ring_pos, TLC_TIMING_RING);
    const int n = (int)std::min<unsigned long long>(have, (unsigned long long)cap);
    for (int i = 0; i < n; ++i) {
        const int set = (int)((g->ring_pos - n + i) % TLC_TIMING_RING);
        h_ms[i] = -1.0;
        if (g->ev_ring_ready && g->ev_ring_used[set][slot]) {
            TLC_HIP_CHECK(hipEventSynchronize(g->ev_ring[set][2 * slot + 1]));
            float ms = 0.f;
            TLC_HIP_CHECK(hipEventElapsedTime(&ms, g->ev_ring[set][2 * slot], g->ev_ring[set][2 * slot + 1]));
            h_ms[i] = (double)ms;
        }
    }
    *n_out = n;
    return TLC_OK;
}

// vicinity sizes of the last chunk: h_n[i] = |S| (0 for pairs finished early), h_m2[i] = induced directed entries
extern "C" int tlc_pd_pi_batch_sizes(tlc_graph* g, int32_t* h_n, int32_t* h_m2, int64_t cap, void* stream) {
    TLC_REQUIRE(g && h_n && h_m2, "null argument");
    TLC_ON_DEVICE(g->device);
    { int rc_p = finish_pending(g); if (rc_p != TLC_OK) return rc_p; }
    TLC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    const size_t k = (size_t)std::min<int64_t>(cap, g->last_n_pairs);
    if (k && g->last_ws) {
        if (g->last_ws->busy) { TLC_HIP_CHECK(hipEventSynchronize(g->last_ws->ev_done)); g->last_ws->busy = 0; }
        TLC_HIP_CHECK(hipMemcpy(h_n, g->last_ws->hdr_n, k * sizeof(int), hipMemcpyDeviceToHost));
        TLC_HIP_CHECK(hipMemcpy(h_m2, g->last_ws->hdr_m2, k * sizeof(int), hipMemcpyDeviceToHost));
    }
    return TLC_OK;
}

// Algorithmic HBM bytes per pair, host-side accounting on the host CSR:
//   8 [pair] + sum_{x in B_{hop-1}(u) U B_{hop-1}(v)} (8 + 4 deg x) + sum_{x in S} (8 + 12 deg x) + 8 res^2
extern "C" int tlc_pd_pi_algorithmic_bytes(int32_t n_nodes, const int32_t* h_rowptr, const int32_t* h_col, const int32_t* h_pairs,
                                           int64_t n_pairs, int hop, int res, double* h_out_bytes) {
    TLC_REQUIRE(n_nodes > 0 && h_rowptr && h_col && h_out_bytes && (n_pairs == 0 || h_pairs), "null argument");
    std::vector<int> su(n_nodes, 0), sv(n_nodes, 0), sx(n_nodes, 0), fr(n_nodes), nx(n_nodes), ball(n_nodes);
    int ep = 0;
    for (int64_t i = 0; i < n_pairs; ++i) {
        const int u = h_pairs[2 * i], v = h_pairs[2 * i + 1];
        double b = 8.0 + 8.0 * res * res;
        if (u < 0 || v < 0 || u >= n_nodes || v >= n_nodes || h_rowptr[u + 1] == h_rowptr[u] || h_rowptr[v + 1] == h_rowptr[v]) {
            h_out_bytes[i] = b;
            continue;
        }
        ++ep;
        int n_ball = 0;
        for (int side = 0; side < 2; ++side) {
            const int root = side ? v : u;
            std::vector<int>& st = side ? sv : su;
            int nf = 1;
            fr[0] = root; st[root] = ep;
            if (side == 0) ball[n_ball++] = root;
            for (int d = 0; d < hop && nf > 0; ++d) {
                int nn = 0;
                for (int k = 0; k < nf; ++k) {
                    const int a = fr[k];
                    if (sx[a] != ep) { sx[a] = ep; b += 8.0 + 4.0 * (h_rowptr[a + 1] - h_rowptr[a]); }
                    for (int j = h_rowptr[a]; j < h_rowptr[a + 1]; ++j) {
                        const int c = h_col[j];
                        if (st[c] != ep) { st[c] = ep; nx[nn++] = c; if (side == 0) ball[n_ball++] = c; }
                    }
                }
                std::copy(nx.begin(), nx.begin() + nn, fr.begin());
                nf = nn;
            }
        }
        for (int k = 0; k < n_ball; ++k) {
            const int a = ball[k];
            if (sv[a] == ep) b += 8.0 + 12.0 * (h_rowptr[a + 1] - h_rowptr[a]);
        }
        h_out_bytes[i] = b;
    }
    return TLC_OK;
}
