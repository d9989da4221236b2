// pdgnn_forward.hip -- the PDGNN inference forward as ONE library call.
//
//   tlc_pdgnn_forward   Teacher_Model.forward(compute_loss=False) of Knowledge_Distillation/Teacher_model.py:49-88 without gradients:
//                       the CSR by target of the batch (gat_conv.py:146-152), its tile cut, conv1 -> conv2 -> conv4 -> conv3 with the
//                       PReLUs between them (Base_Model.forward :218-227), the edge head (:54-59) and one 5 x 5 image per graph (:84).
//
// Every step is a launch of a kernel this library already exports on its own (tlc_csr_by_target, tlc_gat_tile_cut,
// tlc_gat_layer_tiled_fwd / tlc_gat_layer_fwd, tlc_edge_head_fwd, tlc_pi_raster); what this entry adds is the ORDER, submitted from
// native code out of one caller-provided workspace.  On 4 096 hop-1 vicinities of an Amazon-shaped graph (19 000 nodes) the device
// needs 0.1 ms for the forward while the same steps submitted one by one from the host language took 0.5 ms: two dozen launches each
// paid an allocation, an argument conversion and a device guard, and the CSR build allocated, waited and freed on every batch.
#include <algorithm>

#include "tlc_common.h"
#include "gat_internal.h"

namespace {
inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

struct PdgnnWork {
    size_t rowptr, col, nnz, csr_tmp, cut_work, tiles, h0, h1, layer, prep, head, pts, total;
    PdgnnWork(long long n, long long E, int hidden, int tile_nodes) {
        const long long m = E - n, C = hidden, N2 = 2 * C + 4, cin = 2 * C;
        size_t o = 0;
        auto take = [&](size_t bytes) { const size_t at = o; o += up256(bytes); return at; };
        rowptr = take((size_t)(n + 1) * 4);
        col = take((size_t)(E + n) * 4);
        nnz = take(4);
        csr_tmp = take((size_t)(3 * n + E) * 4);
        cut_work = take((size_t)(n / 32 + 6) * 4);
        tiles = take((size_t)(2 * n / tile_nodes + 3) * 4);
        h0 = take((size_t)n * 2 * C * 4);
        h1 = take((size_t)n * 2 * C * 4);
        const size_t tiled = (size_t)(cin * C + C * N2 + cin * N2 + 2 * C + 8), plain = (size_t)n * (3 * C + 4) + cin * C + C * (2 * C + 4);
        layer = take(std::max(tiled, plain) * 4);
        prep = take((size_t)TLC_GAT_PREP_MAX * cin * N2 * 4);
        head = take((size_t)(n + C) * 2 * hidden * 4);            // conv3 gives C = hidden channels per node
        pts = take((size_t)std::max(m, 1ll) * 2 * 8);
        total = o;
    }
};
}  // namespace

extern "C" int64_t tlc_pdgnn_forward_work_bytes(int32_t n_nodes, int64_t n_edges, int32_t hidden) {
    if (n_nodes <= 0 || n_edges < n_nodes || hidden <= 0) return -1;           // (the same sizes tlc_pdgnn_forward takes)
    return (int64_t)PdgnnWork(n_nodes, n_edges, hidden, 192).total;
}

extern "C" int tlc_pdgnn_forward(int32_t n_nodes, int64_t n_edges, const int64_t* d_edge_index, const float* d_x, int32_t hidden,
                                 const float* const* params, int64_t n_graphs, const int64_t* d_edge_ptr, int32_t res,
                                 const int32_t* d_rowptr, const int32_t* d_col, const int32_t* d_tile_ptr, int32_t n_tiles,
                                 void* d_work, int64_t work_bytes, float* d_points, double* d_img, void* stream) {
    TLC_REQUIRE(n_nodes > 0 && n_edges >= n_nodes && n_graphs >= 0 && res > 0, "bad sizes");
    TLC_REQUIRE(d_edge_index && d_x && params && d_work && d_points && (n_graphs == 0 || (d_edge_ptr && d_img)), "null pointer");
    TLC_REQUIRE(hidden == 32, "tlc_pdgnn_forward: hidden_dim 32 (the reference's) is what the layers are built for");
    TLC_REQUIRE((d_rowptr == nullptr) == (d_col == nullptr) && (d_rowptr || !d_tile_ptr) && (d_tile_ptr || n_tiles == 0), "rowptr / col / tiles: all of the batch's structure or none");
    for (int i = 0; i < 20; ++i) TLC_REQUIRE(params[i], "null parameter");
    const PdgnnWork w(n_nodes, n_edges, hidden, 192);
    TLC_REQUIRE(work_bytes >= (int64_t)w.total, "workspace smaller than tlc_pdgnn_forward_work_bytes");
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)d_work;
    const long long m = n_edges - n_nodes;
    int rc;
    // conv1 -> conv2 -> conv4 -> conv3 (Base_Model.forward), PReLU(0.1) fused behind the first three
    const int C = hidden;
    struct { int c_in, c_out, p; float slope; } L[4] = {{1, C, 0, 0.1f}, {2 * C, C, 4, 0.1f}, {2 * C, C, 8, 0.1f}, {2 * C, C / 2, 12, -1.0f}};
    const float* in = d_x;
    float* bufs[2] = {(float*)(base + w.h0), (float*)(base + w.h1)};
    float* prep = (float*)(base + w.prep);
    const size_t prep_stride = (size_t)2 * C * (2 * C + 4);
    // first what needs no structure -- the combined weights of the four (tiled) layers in one launch -- so that the device has work
    // while the host waits for the tile count
    if (!d_rowptr || n_tiles > 0) {
        TlcGatPrepLayer pl[4];
        for (int l = 0; l < 4; ++l) pl[l] = {L[l].c_in, L[l].c_out, params[L[l].p], params[L[l].p + 1], params[L[l].p + 2], prep + l * prep_stride};
        rc = tlc_gat_tiled_prepare(4, pl, s);
        if (rc != TLC_OK) return rc;
    }
    // the batch's structure, unless the caller holds it (a loop over the same batch: train_Teacher_Model.py:124-151)
    const int32_t *rowptr = d_rowptr, *col = d_col, *tiles = d_tile_ptr;
    if (!rowptr) {
        rc = tlc_csr_by_target(n_nodes, n_edges, d_edge_index, (int32_t*)(base + w.rowptr), (int32_t*)(base + w.col), (int32_t*)(base + w.nnz),
                               (int32_t*)(base + w.csr_tmp), stream);
        if (rc != TLC_OK) return rc;
        rowptr = (const int32_t*)(base + w.rowptr);
        col = (const int32_t*)(base + w.col);
        rc = tlc_gat_tile_cut(n_nodes, rowptr, col, 192, (int32_t*)(base + w.cut_work), (int32_t*)(base + w.tiles), &n_tiles, stream);
        if (rc != TLC_OK) return rc;
        tiles = (const int32_t*)(base + w.tiles);
    }
    for (int l = 0; l < 4; ++l) {
        float* out = bufs[l & 1];
        const float *Wl = params[L[l].p], *att = params[L[l].p + 1], *Wij = params[L[l].p + 2], *bias = params[L[l].p + 3];
        rc = n_tiles > 0 ? tlc_gat_tiled_run(n_tiles, tiles, rowptr, col, in, L[l].c_in, L[l].c_out, prep + l * prep_stride, bias, L[l].slope, out, s)
                         : tlc_gat_layer_fwd(n_nodes, rowptr, col, in, L[l].c_in, L[l].c_out, Wl, att, Wij, bias, L[l].slope, (float*)(base + w.layer), out, stream);
        if (rc != TLC_OK) return rc;
        in = out;
    }
    if (m == 0) {
        if (n_graphs) TLC_HIP_CHECK(hipMemsetAsync(d_img, 0, (size_t)n_graphs * res * res * sizeof(double), s));
        return TLC_OK;
    }
    // the edge head over the batch without its self loops (they are the LAST n columns), one image per graph
    double* pts = n_graphs ? (double*)(base + w.pts) : nullptr;
    rc = tlc_edge_head_fwd_i64(m, (const long long*)d_edge_index, (const long long*)d_edge_index + n_edges, in, C, params[16], params[17], hidden, 0.1f,
                               params[18], params[19], d_points, pts, n_nodes, (float*)(base + w.head), s);
    if (rc != TLC_OK) return rc;
    if (n_graphs) {
        rc = tlc_pi_raster((int32_t)n_graphs, d_edge_ptr, pts, res, d_img, stream);
        if (rc != TLC_OK) return rc;
    }
    return TLC_OK;
}
