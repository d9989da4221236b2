// gat_internal.h -- what gat_forward.hip offers pdgnn_forward.hip beside the C ABI: the tiled PDGNN layer in two steps, so that the
// weights of all layers of a forward are prepared by ONE launch.
#pragma once
#include "tlc_common.h"

// one layer's combined weights: c_in == 1: prep[0 .. 2C] = Wl[:, 0]^T [Wij_t^T | Wij_s^T | att]; c_in == 64: prep[c_in][2C + 4] = Wl^T
// [Wij_t^T | Wij_s^T | att | 0 0 0].  prep: float32[c_in * (2 c_out + 4)] of the caller's scratch.
struct TlcGatPrepLayer {
    int c_in, c_out;
    const float *Wl, *att, *Wij;
    float* prep;
};
constexpr int TLC_GAT_PREP_MAX = 4;
int tlc_gat_tiled_prepare(int n_layers, const TlcGatPrepLayer* layers, hipStream_t s);
int tlc_gat_tiled_run(int n_tiles, const int* tile_ptr, const int* rowptr, const int* src, const float* X, int c_in, int c_out, const float* prep,
                      const float* bias, float prelu_slope, float* out, hipStream_t s);
// the edge head of Teacher_Model.forward (tlc_edge_head_fwd) over an int64 edge_index as it lies, the points also as float64 (may be null)
int tlc_edge_head_fwd_i64(int64_t n_edges, const long long* d_src, const long long* d_dst, const float* d_X, int c, const float* d_W5, const float* d_b5,
                          int hidden, float prelu_slope, const float* d_W6, const float* d_b6, float* d_pd, double* d_pd64, int n_nodes, float* d_work,
                          hipStream_t stream);
