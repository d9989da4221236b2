"""ctypes binding of libtlcgnn_hip.so (the C ABI of include/tlcgnn.h).

PyTorch is plumbing here: it owns device memory and streams; every compute call goes through the C ABI
with raw device pointers.  There is NO CPU fallback: if the HIP library is missing or no GPU is visible,
the product path raises.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtlcgnn_hip.so")
CSRC = os.path.join(HERE, "csrc")

OK = 0
ERRORS = {1: "TLC_ERR_INVALID_ARG", 2: "TLC_ERR_HIP", 3: "TLC_ERR_NO_DEVICE", 4: "TLC_ERR_UNSUPPORTED",
          5: "TLC_ERR_OUT_OF_MEMORY"}

ST_OK, ST_MISSING_NODE, ST_DISCONNECTED, ST_ZERO_RANGE, ST_NO_TREE_EDGE, ST_TOO_LARGE = range(6)
KEEP_ZERO_PERS, INCLUDE_ROOTS, NORM_EPS, PI_ORD0_EXT1, NO_EXT1, UNREACHABLE_100 = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
DESC_MIN, DESC_MAX, DESC_ROOT1, NO_NORM = 0x40, 0x80, 0xC0, 0x100
DESCRIPTOR_FLAG = {"sum": 0, "min": DESC_MIN, "max": DESC_MAX}      # the three node values of filtration.build_fv

# every symbol include/tlcgnn.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "tlc_version", "tlc_last_error", "tlc_device_count", "tlc_graph_create", "tlc_graph_destroy",
    "tlc_pd_pi_batch", "tlc_pd_pi_batch_async", "tlc_pd_pi_batch_join", "tlc_vicinity_filtration", "tlc_pd_pi_batch_stats", "tlc_pd_pi_batch_set_timing",
    "tlc_pd_pi_batch_timings", "tlc_pd_pi_batch_timing_history", "tlc_pd_pi_batch_sizes", "tlc_pd_pi_algorithmic_bytes", "tlc_pd_from_filtration",
    "tlc_pi_raster", "tlc_pi_raster_wgrad", "tlc_gcn_norm_csr", "tlc_gemm_f32", "tlc_spgemm_csr_dense_f32", "tlc_spmm_csr_f32", "tlc_renorm_rows_f32", "tlc_gcn2_encode_f32", "tlc_gcn2_encode_csr_f32",
    "tlc_lp_decode_fused", "tlc_lp_decode_fused_f32", "tlc_gat_layer_fwd", "tlc_gat_layer_tiled_fwd", "tlc_gat_tile_cut", "tlc_csr_by_target", "tlc_pdgnn_forward", "tlc_pdgnn_forward_work_bytes", "tlc_scatter_f32", "tlc_edge_head_fwd",
    "tlc_complement_rows", "tlc_complement_pairs", "tlc_select_rows", "tlc_pack_vicinities", "tlc_stack_batch", "tlc_ollivier_ricci_sinkhorn",
    "tlc_near_pairs", "tlc_w2_partial_matching", "tlc_w2_inference_matching", "tlc_gat_layer_bwd", "tlc_edge_head_bwd", "tlc_pack_offsets", "tlc_vicinity_sizes", "tlc_debug_dc_stats", "tlc_debug_tier_counts", "tlc_debug_phase_profile", "tlc_debug_set_option", "tlc_debug_pair_times",
]


class TlcError(RuntimeError):
    pass


def build(verbose=False, force=False):
    """Compile the HIP sources for gfx950 into tlc-gnn_amd/libtlcgnn_hip.so (hipcc cross-compiles without a GPU).
    force=True: `make clean` first, so every object is rebuilt from source (the "does it build" check must not be
    satisfied by a shipped .so)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise TlcError("building libtlcgnn_hip.so failed")
    return LIB_PATH


_lib = None


def lib():
    """The loaded library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TlcError("libtlcgnn_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the product path)")
        # torch first: it ships its own libamdhip64, and a process must end up with ONE HIP runtime -- with this library (linked
        # against /opt/rocm's) loaded before torch, the two copies each initialise and the second sees no device
        # (`build()` followed by `smoke()` in one process: TLC_ERR_NO_DEVICE)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.tlc_version.restype = C.c_char_p
        L.tlc_last_error.restype = C.c_char_p
        L.tlc_graph_create.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.tlc_graph_destroy.argtypes = [C.c_void_p]
        L.tlc_pd_pi_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        L.tlc_pd_pi_batch_async.argtypes = L.tlc_pd_pi_batch.argtypes
        L.tlc_pd_pi_batch_join.argtypes = [C.c_void_p, C.c_void_p]
        L.tlc_vicinity_filtration.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]
        L.tlc_pd_pi_batch_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.tlc_pd_pi_batch_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.tlc_pd_pi_batch_timings.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.tlc_pd_pi_batch_timing_history.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.tlc_pd_pi_batch_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.tlc_pd_pi_algorithmic_bytes.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.tlc_pd_from_filtration.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 7
        L.tlc_pi_raster.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.tlc_pi_raster_wgrad.argtypes = [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(L, "tlc_gcn_norm_csr"):
            L.tlc_gcn_norm_csr.argtypes = [C.c_int32, C.c_int64] + [C.c_void_p] * 6
            L.tlc_gemm_f32.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p]
            L.tlc_spgemm_csr_dense_f32.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p]
            L.tlc_spmm_csr_f32.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_void_p, C.c_int, C.c_void_p,
                                                                            C.c_void_p]
            L.tlc_renorm_rows_f32.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
            L.tlc_gcn2_encode_f32.argtypes = ([C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                              C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
            L.tlc_gcn2_encode_csr_f32.argtypes = ([C.c_int32] + [C.c_void_p] * 6 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                  C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
            L.tlc_lp_decode_fused.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L.tlc_lp_decode_fused_f32.argtypes = L.tlc_lp_decode_fused.argtypes
        if hasattr(L, "tlc_gat_layer_fwd"):
            L.tlc_gat_layer_fwd.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
            L.tlc_csr_by_target.argtypes = [C.c_int32, C.c_int64] + [C.c_void_p] * 6
            L.tlc_pdgnn_forward_work_bytes.argtypes = [C.c_int32, C.c_int64, C.c_int32]
            L.tlc_pdgnn_forward_work_bytes.restype = C.c_int64
            L.tlc_pdgnn_forward.argtypes = [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
            L.tlc_gat_tile_cut.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
            L.tlc_gat_layer_tiled_fwd.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]
            L.tlc_scatter_f32.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
            L.tlc_edge_head_fwd.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int32, C.c_void_p, C.c_void_p]
        L.tlc_complement_rows.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tlc_complement_pairs.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                           C.c_void_p, C.c_void_p]
        L.tlc_select_rows.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_uint32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tlc_pack_vicinities.argtypes = [C.c_int64] + [C.c_void_p] * 14
        L.tlc_pack_offsets.argtypes = [C.c_int64] + [C.c_void_p] * 6
        L.tlc_stack_batch.argtypes = [C.c_int64] + [C.c_void_p] * 4 + [C.c_int64, C.c_int64] + [C.c_void_p] * 3
        L.tlc_vicinity_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tlc_ollivier_ricci_sinkhorn.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_double,
                                                  C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                                  C.c_int64, C.c_void_p]
        L.tlc_near_pairs.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
        L.tlc_w2_partial_matching.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int32] + [C.c_void_p] * 7
        L.tlc_w2_inference_matching.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int32] + [C.c_void_p] * 9
        L.tlc_gat_layer_bwd.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_float] + [C.c_void_p] * 9
        L.tlc_edge_head_bwd.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_float] + [C.c_void_p] * 9
        L.tlc_debug_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.tlc_debug_dc_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.tlc_debug_tier_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.tlc_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != OK:
        msg = lib().tlc_last_error().decode("utf-8", "replace")
        raise TlcError("%s: %s (%s)" % (what or "libtlcgnn_hip", ERRORS.get(rc, rc), msg))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise TlcError("no MI355X/HIP device visible: the TLC-GNN hot path has no CPU fallback")
    return torch


def stream_ptr(device=None):
    """torch's current stream ON `device` (default: torch's current device) -- a stream belongs to one device."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device_of(fn):
    """Decorator for the pointer-only entry points (no handle, so the library cannot know the device): run the call with the
    device of its first CUDA tensor argument current, so the stream handed over and the temporaries belong to the device
    that owns the pointers; the caller's current device is restored on exit."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        import torch
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapped


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())
