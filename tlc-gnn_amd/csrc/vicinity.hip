// vicinity.hip -- P4 of the hot path: S = ball_hop(u) & ball_hop(v) and the induced weighted subgraph.
//
// Replaces sg2dgm_accelerate's BFS + set intersection + graph.subgraph (sg2dgm/riccidist2dgm.py:311-316).
//
// One wavefront per pair (64-thread workgroups, pairs dequeued from an atomic counter so that the
// heavy-tailed vicinity sizes balance).  The two balls live as N-bit bitmaps in LDS (3 * N/8 bytes per
// wave: ball(u), ball(v) -> S, and the per-word popcount prefix that turns a set bit into a local id),
// so membership tests never leave the CU; CSR rows are read with one lane per short row and with the
// whole wave (coalesced) for rows of >= 32 entries.
//
// Two passes over a batch (no device-side allocation, no overflow path):
//   COUNT  writes |S|, the number of induced directed entries and the local ids of u and v per pair,
//          and finishes the pairs that need no further work (missing endpoint, empty vicinity);
//   FILL   (after an exclusive scan sized the arena) writes the induced subgraph of every remaining
//          pair as packed directed entries (src<<16 | dst, local ids ascending by node id) + fp64 weights.
#include "vicinity_dev.h"

template <bool FILL, int BW>
__global__ __launch_bounds__(BW) void tlc_vicinity_kernel(TlcVicParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    int* slot = p.scratch + (size_t)(p.scratch_base_slot + blockIdx.x) * p.scratch_stride;
    // static striding over the pairs: a single global work counter saturates at ~90 dequeues/us (MI355X_MICROARCH.md,
    // "dequeue"), which would cap this kernel at ~0.4 ms per 37k pairs; consecutive pairs land on different workgroups,
    // so hub-heavy runs of the pair list are spread out anyway
    int n_work = p.n_pairs;
    if (p.fill_mode == 1) {
        n_work = p.fill_count;
        if (p.work_count_dev) { const int c = *p.work_count_dev; n_work = c < n_work ? c : n_work; }
    }
    if (p.started && threadIdx.x == 0 && (int)blockIdx.x < n_work) atomicAdd(p.started, 1);
    if (p.work_counter) {
        // (one dequeue per chunk: a single counter serves ~90 dequeues/us, MI355X_MICROARCH.md, so the host sizes the chunks
        // to keep the total in the low thousands)
        __shared__ int s_chunk;
        const int n_chunks = (n_work + p.work_chunk - 1) / p.work_chunk;
        // the first chunk of a workgroup is its own index (thousands of workgroups hitting the counter at once would wait
        // ~70 us for it); the counter hands out the chunks from gridDim.x on
        for (int c = blockIdx.x;;) {
            if (c >= n_chunks) break;
            // (a chunk is strided, not contiguous: runs of the pair list that share a hub endpoint stay spread out)
            for (int wi = c; wi < n_work; wi += n_chunks) vicinity_pair<FILL, BW>(p, p.fill_mode == 1 ? p.fill_list[wi] : wi, lds, slot);
            if (threadIdx.x == 0) s_chunk = (int)gridDim.x + atomicAdd(p.work_counter, 1);
            __syncthreads();
            c = s_chunk;
            __syncthreads();
        }
        return;
    }
    for (int wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const int i = p.fill_mode == 1 ? p.fill_list[wi] : wi;
        vicinity_pair<FILL, BW>(p, i, lds, slot);
    }
}

template __global__ void tlc_vicinity_kernel<false, 64>(TlcVicParams);
template __global__ void tlc_vicinity_kernel<false, 512>(TlcVicParams);
template __global__ void tlc_vicinity_kernel<true, 64>(TlcVicParams);
template __global__ void tlc_vicinity_kernel<true, 512>(TlcVicParams);

// ---- exclusive scan of the per-pair sizes + tier binning + publication of the sizes, one kernel ----------------------
// n_pairs per chunk is <= 2^20 (<= 1024 blocks).  Chained scan: a block takes its index from a ticket (so that every
// block it has to wait for is already running or done), scans its 1024 sizes with wave shuffles, publishes its sum behind
// a flag and adds up the sums of the blocks before it.  The last block to finish stores the arena size and the tier
// counts into mapped host memory, fences at system scope and bumps the sequence number the host polls.
#define SCAN_BLOCK TLC_SCAN_BLOCK

// `bumped`: COUNT has written the MID / MEDIUM vicinities already; only the heavy tiers need space
__device__ __forceinline__ long long arena_entries(int n, int m2, int small_arena, bool bumped) {
    if (n <= 0) return 0;
    if (small_arena && n <= TLC_S_NMAX && (m2 >> 1) <= TLC_S_MMAX) return 0;
    if (bumped && n <= TLC_M_NMAX && (m2 >> 1) <= TLC_M_MMAX) return 0;
    return (long long)m2;
}

__global__ __launch_bounds__(SCAN_BLOCK) void tlc_scan_bin(TlcScanParams p) {
    __shared__ long long s_wave[SCAN_BLOCK / TLC_WAVE];
    __shared__ long long s_prefix;
    __shared__ int s_bid;
    __shared__ unsigned s_pre[SCAN_BLOCK / 32];       // pairs of this block that the early pass has already written
    __shared__ int s_tc[TLC_N_TIERS][SCAN_BLOCK / TLC_WAVE];
    __shared__ int s_tbase[TLC_N_TIERS];
    __shared__ int s_bc[TLC_TINY_BINS][SCAN_BLOCK / TLC_WAVE];
    __shared__ int s_bbase[TLC_TINY_BINS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_bid = atomicAdd(&p.sync[0], 1);
    if (t < SCAN_BLOCK / 32) s_pre[t] = 0u;
    __syncthreads();
    const int bid = s_bid;
    const int i = bid * SCAN_BLOCK + t;
    int n_early = 0;
    if (p.early_list) {
        n_early = *p.early_count;
        n_early = n_early < p.early_cap ? n_early : p.early_cap;
        for (int k = t; k < n_early; k += SCAN_BLOCK) {
            const int j = p.early_list[k] - bid * SCAN_BLOCK;
            if (j >= 0 && j < SCAN_BLOCK) atomicOr(&s_pre[j >> 5], 1u << (j & 31));
        }
        __syncthreads();
    }
    const bool pre = (s_pre[t >> 5] >> (t & 31)) & 1u;
    int n = i < p.n_pairs ? p.hdr_n[i] : 0;
    const int m2v = i < p.n_pairs ? p.hdr_m2[i] : 0;
    const bool bumped = p.bump_top != nullptr && *p.bump_overflow == 0;
    const long long arena_base = bumped ? p.bump_base + (long long)*p.bump_top : 0;
    const long long own = (i < p.n_pairs && !pre) ? arena_entries(n, m2v, p.small_arena, bumped) : 0;
    // inclusive scan inside the wavefront, then over the 16 wavefront totals
    // (sizes are < 2^25 per pair, so a wavefront's running sum fits 32 bits)
    long long incl = (long long)(unsigned)tlc_wave_iscan_i32((int)own);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    long long wbase = 0, btotal = 0;
#pragma unroll
    for (int k = 0; k < SCAN_BLOCK / TLC_WAVE; ++k) {
        const long long v = s_wave[k];
        if (k < wave) wbase += v;
        btotal += v;
    }
    // publish this block's sum, then add up the blocks before it (wavefront 0, one predecessor per lane)
    if (t == 0) {
        p.block_agg[bid] = btotal;
        __hip_atomic_store(&p.block_flag[bid], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {
        long long acc = 0;
        for (int base = 0; base < bid; base += TLC_WAVE) {
            const int j = base + lane;
            long long v = 0;
            if (j < bid) {
                while (__hip_atomic_load(&p.block_flag[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(&p.block_agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            acc += tlc_wave_sum_i64(v);
        }
        if (lane == 0) s_prefix = acc;
    }
    __syncthreads();
    int tier = -1;
    if (i < p.n_pairs) {
        if (own > 0 || !bumped) p.edge_off[i] = arena_base + s_prefix + wbase + incl - own;   // (else: COUNT's offset stands)
        if (n > 0 && !pre) {
            const int m = m2v >> 1;
            tier = TLC_TIER_HUGE;
            if (p.tiny_ok && n <= TLC_T_NCUT && m <= TLC_T_MCUT) tier = TLC_TIER_TINY;
            else if (n <= TLC_S_NMAX && m <= TLC_S_MMAX) tier = TLC_TIER_SMALL;
            else if (n <= TLC_D_NMAX && m <= TLC_D_MMAX) tier = TLC_TIER_MID;
            else if (n <= TLC_M_NMAX && m <= TLC_M_MMAX) {
                // the MEDIUM-sized vicinities: those within the compact configuration -> MEDIUM; the rest -> MEDWIDE.  With the split
                // by Pos edges on (a chunk on its own: mh_min_pos < INT_MAX) the ones with many Pos edges AND the wide ones are one
                // list, MEDHI: the longest chains of these tiers, launched first (speculatively) with the wide kernels
                // (a vicinity with Pos edges enough for the divide and conquer is "wide" whatever its size: only the wide kernels mark
                // for it, and marking fixes the order of tied descending keys -- a row must not depend on whether its chunk was alone,
                // where the many-Pos list takes it with the wide kernels, or pipelined)
                // (round 6: a lower Pos cut for the wide list, for the sake of the compact list's swap kernel -- the last kernel of most batches,
                // as long as its longest walk -- measured: 160 / 128: -0.5 % per pipelined batch (noise), 96: +2 %.  The cut stays the
                // divide and conquer's.)
                const bool wide = n > TLC_C_NMAX || m > TLC_C_MMAX || m - n + 1 >= TLC_DC_MIN_POS_SHARED;
                const bool split = p.mh_min_pos != 0x7fffffff;
                if (split && !p.mh_compact_only && (wide || m - n + 1 >= p.mh_min_pos)) tier = TLC_TIER_MEDHI;
                else if (split && p.mh_compact_only && !wide && m - n + 1 >= p.mh_min_pos) tier = TLC_TIER_MEDHI;   // (same kernels as MEDIUM: the front of its launch)
                else tier = wide ? TLC_TIER_MEDWIDE : TLC_TIER_MEDIUM;
            }
            else if (n <= TLC_L_NMAX && m <= TLC_L_MMAX) tier = TLC_TIER_LARGE;
        }
    }
    // block-aggregated append: one atomic per block and tier (every wavefront of every block adding to the same five counters
    // cost the scan ~15 us of serialised L2 atomics)
    int my_rank = 0;                                  // position of this pair among the block's pairs of its tier
#pragma unroll
    for (int tt = 0; tt < TLC_N_TIERS; ++tt) {
        const unsigned long long mk = __ballot(tier == tt);
        if (lane == 0) s_tc[tt][wave] = __popcll(mk);
        if (tier == tt) my_rank = __popcll(mk & tlc_lanemask_lt());
    }
    __syncthreads();
    if (t < TLC_N_TIERS) {
        int tot = 0;
        for (int k = 0; k < SCAN_BLOCK / TLC_WAVE; ++k) { const int c = s_tc[t][k]; s_tc[t][k] = tot; tot += c; }
        s_tbase[t] = tot > 0 ? atomicAdd(&p.tier_count[t], tot) : 0;
    }
    __syncthreads();
    if (tier >= 0) p.tier_list[(size_t)tier * p.n_pairs + s_tbase[tier] + s_tc[tier][wave] + my_rank] = i;
    // how many vicinities of the wide MEDIUM configuration have Pos edges enough for the divide and conquer (the host puts
    // tlc_pd_dc_kernel into that tier's chain only when there are any: an idle kernel there costs a full machine 70 - 85 us)
    if (p.dcm_count) {
        const int mm = m2v >> 1;
        const bool big = (tier == TLC_TIER_MEDHI || tier == TLC_TIER_MEDWIDE) && mm - n + 1 >= TLC_DC_MIN_POS_SHARED && mm <= 8 * 256;
        const unsigned long long mk = __ballot(big);
        if (lane == 0 && mk) atomicAdd(p.dcm_count, __popcll(mk));
    }
    // the TINY pairs once more, by size class (same block-aggregated append)
    if (p.tiny_bin_count) {
        int bin = -1;
        if (tier == TLC_TIER_TINY) {
            const int c = (n + (m2v >> 1)) / TLC_TINY_BIN_W;
            bin = TLC_TINY_BINS - 1 - (c < TLC_TINY_BINS - 1 ? c : TLC_TINY_BINS - 1);
        }
        int brank = 0;
#pragma unroll
        for (int b = 0; b < TLC_TINY_BINS; ++b) {
            const unsigned long long mk = __ballot(bin == b);
            if (lane == 0) s_bc[b][wave] = __popcll(mk);
            if (bin == b) brank = __popcll(mk & tlc_lanemask_lt());
        }
        __syncthreads();
        if (t < TLC_TINY_BINS) {
            int tot = 0;
            for (int k = 0; k < SCAN_BLOCK / TLC_WAVE; ++k) { const int c = s_bc[t][k]; s_bc[t][k] = tot; tot += c; }
            s_bbase[t] = tot > 0 ? atomicAdd(&p.tiny_bin_count[t], tot) : 0;
        }
        __syncthreads();
        if (bin >= 0) p.tiny_bin_list[(size_t)bin * p.n_pairs + s_bbase[bin] + s_bc[bin][wave] + brank] = i;
    }
    // the last block to get here publishes the sizes
    __syncthreads();
    if (t == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(p.sync + 2), (unsigned long long)btotal);   // running arena total
        __threadfence();
        if (atomicAdd(&p.sync[1], 1) == (int)gridDim.x - 1) {
            const long long total = arena_base + (long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(p.sync + 2),
                                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.h_overflow) *p.h_overflow = p.bump_overflow ? *p.bump_overflow : 0;
            p.totals[0] = total;
            *p.h_total = total;
            if (p.h_early) {
                int valid = 0;                            // (a slot may be empty: extract.hip)
                for (int k = 0; k < n_early; ++k) valid += p.early_list[k] >= 0 ? 1 : 0;
                *p.h_early = valid;
            }
            for (int tt = 0; tt < TLC_N_TIERS; ++tt)
                p.h_tier[tt] = __hip_atomic_load(&p.tier_count[tt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.dcm_count && p.h_dcm) *p.h_dcm = __hip_atomic_load(p.dcm_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.tiny_bin_count && p.h_tiny_bins)
                for (int b = 0; b < TLC_TINY_BINS; ++b)
                    p.h_tiny_bins[b] = __hip_atomic_load(&p.tiny_bin_count[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(p.h_seq, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- early pass: which pairs are predicted heavy ----------------------------------------------------------------------
// ub[x] >= |ball_hop(x)|: ub_1 = 1 + deg, ub_h(x) = 1 + sum over the neighbours y of ub_{h-1}(y) (saturating).  A vicinity is
// a subset of both endpoints' balls, so min(ub[u], ub[v]) bounds its node count: on the PubMed-shaped batch the bound
// exceeds the MEDIUM tier for 185 of 37 676 pairs, among them all 71 LARGE-tier ones.
__global__ void tlc_ball_bound_kernel(int n_nodes, const int* __restrict__ rowptr, const int* __restrict__ col,
                                      const int* __restrict__ prev, int* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_nodes) return;
    const int b = rowptr[x], e = rowptr[x + 1];
    long long s = 1;
    if (!prev) s += e - b;
    else for (int j = b; j < e; ++j) s += prev[col[j]];
    out[x] = s > (1 << 30) ? (1 << 30) : (int)s;
}
int tlc_launch_ball_bound(int n_nodes, const int* rowptr, const int* col, const int* prev, int* out, void* stream) {
    if (n_nodes <= 0) return TLC_OK;
    hipLaunchKernelGGL(tlc_ball_bound_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_nodes, rowptr, col,
                       prev, out);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// candidates of the early pass: the first `cap` pairs (in no particular order) whose bound exceeds `threshold`
__global__ void tlc_select_heavy_kernel(int n_pairs, const int* __restrict__ pairs, int n_nodes, const int* __restrict__ ub,
                                        int threshold, int cap, int* __restrict__ count, int* __restrict__ list) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool cand = false;
    if (i < n_pairs) {
        const int u = pairs[2 * (size_t)i], v = pairs[2 * (size_t)i + 1];
        if (u >= 0 && v >= 0 && u < n_nodes && v < n_nodes) {
            const int a = ub[u], b = ub[v];
            cand = (a < b ? a : b) >= threshold;
        }
    }
    const unsigned long long mk = __ballot(cand);
    if (mk == 0ull) return;
    int base = 0;
    const int leader = __builtin_ctzll(mk);
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(count, __popcll(mk));
    base = __builtin_amdgcn_readlane(base, leader);
    const int pos = base + __popcll(mk & tlc_lanemask_lt());
    if (cand && pos < cap) list[pos] = i;
}
int tlc_launch_select_heavy(int n_pairs, const int* pairs, int n_nodes, const int* ub, int threshold, int cap, int* count,
                            int* list, void* stream) {
    if (n_pairs <= 0) return TLC_OK;
    hipLaunchKernelGGL(tlc_select_heavy_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_pairs, pairs,
                       n_nodes, ub, threshold, cap, count, list);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}
