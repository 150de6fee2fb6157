#!/usr/bin/env python
"""The dominant launch on its own: the batched Winograd GEMM [16] x (M x K) x (K x N) through him_winograd_gemm.
Used for the HIP-event timing check and as the target of the PMC passes (profiles/r01_pmc_dominant_kernel.json):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d out1 -- python tools/gemm_bench.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d out2 -- python tools/gemm_bench.py
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -f csv -d out3 -- python tools/gemm_bench.py
"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from neurips18_hierchical_image_manipulation_amd._cabi import lib


def main():
    M = K = 1024
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    a = torch.randn(16, M, K, device='cuda') * 0.02
    b = torch.randn(16, K, N, device='cuda')
    c = torch.empty(16, M, N, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.him_winograd_gemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, K, N, None, st)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        lib.him_winograd_gemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, K, N, None, st)
    e.record()
    e.synchronize()
    ms = s.elapsed_time(e) / iters
    ref = torch.einsum('zmk,zkn->zmn', a[:1].double(), b[:1].double())
    err = (c[:1].double() - ref).abs().max().item() / ref.abs().max().item()
    print('winograd gemm [16]x(%dx%d)x(%dx%d): %.4f ms  %.1f TFLOP/s executed   max rel err vs fp64 %.2e' % (
        M, K, K, N, ms, 2.0 * 16 * M * K * N / ms / 1e9, err))


if __name__ == '__main__':
    main()
