"""Scratch: slowdown of a weight-gradient GEMM while the FPS kernel runs on a side stream."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_bridge_amd import _lib, ops
L = _lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
xyz = torch.rand(16, 16384, 3, device="cuda"); start = torch.zeros(16, dtype=torch.long, device="cuda")
side = torch.cuda.Stream()
def mk(R, M, N):
    dz = torch.randn(R, M, device="cuda").to(torch.bfloat16); y = torch.randn(R, M, device="cuda").to(torch.bfloat16)
    xx = torch.randn(R, N, device="cuda").to(torch.bfloat16)
    v = [torch.rand(M, device="cuda") for _ in range(4)]; xv = [torch.rand(N, device="cuda") for _ in range(2)]
    ws = torch.empty(L.pcb_gemm_tn_workspace(R, M, N), device="cuda"); dw = torch.empty(M, N, device="cuda")
    return lambda: L.pcb_gemm_tn_bf16(2, dz.data_ptr(), y.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                                   0, 0, 1, 1, 1, xx.data_ptr(), xv[0].data_ptr(), xv[1].data_ptr(), 1, R, M, N, ws.data_ptr(), dw.data_ptr(), 0, 0, st())
for shape in ((262144, 128, 128), (262144, 256, 264), (524288, 64, 64)):
    f = mk(*shape)
    def run(with_fps):
        torch.cuda.synchronize()
        if with_fps:
            with torch.cuda.stream(side):
                ops.furthest_point_sample(xyz, 1024, start)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(8): f()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / 8 * 1e3
    for _ in range(3): run(False)
    print(shape, "alone %.1f us, beside FPS %.1f us" % (run(False), run(True)))
