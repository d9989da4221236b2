"""development: the bf16x3 projection probe through its direct entry (tlc_probe_gemm_bf16x3), against the f32 kernel"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from tlc_gnn_amd import ops, _lib
L = _lib.lib()
L.tlc_probe_gemm_bf16x3.argtypes = [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3
L.tlc_probe_gemm_bf16x3_work_bytes.restype = C.c_longlong
torch.manual_seed(0)
for M in (4096, 19717, 80000):
    K, N = 500, 100
    a = torch.rand(M, K, device="cuda") * 0.2
    b = torch.randn(K, N, device="cuda") * 0.1
    bias = torch.randn(N, device="cuda") * 0.1
    out = torch.empty(M, N, device="cuda")
    wb = L.tlc_probe_gemm_bf16x3_work_bytes(M, N, K)
    work = torch.empty(wb, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    def run():
        rc = L.tlc_probe_gemm_bf16x3(M, N, K, a.data_ptr(), b.data_ptr(), bias.data_ptr(), 1, out.data_ptr(), work.data_ptr(), s)
        assert rc == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    ref = torch.relu(a.double() @ b.double() + bias.double())
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    out2 = ops.gemm(a, b, bias, relu=True)
    err2 = float((out2.double() - ref).abs().max() / ref.abs().max())
    print("M=%6d  bf16x3 err %.2e   f32 kernel err %.2e" % (M, err, err2))
    for _ in range(20): run()
    for _ in range(20): ops.gemm(a, b, bias, relu=True, out=out2)
torch.cuda.synchronize()
