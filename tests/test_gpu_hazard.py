"""The packed-fp32 hazard of DESIGN.md section 2, measured for code this library does NOT compile.

Round 5 found v_pk_{mul,add,fma}_f32 returning +0 in lanes 48..63 while another wave runs v_mfma_f32_32x32x16_f16 on the same SIMD, and
rebuilt this library without packed fp32.  torch's elementwise kernels and RCCL's reduction kernels are built WITH it, and the
data-parallel exchange runs them on a second stream beside the MFMA kernels of the backward pass.  This file runs torch's fp32
add / mul / addcmul (the shape of an all-reduce's reduction and of an optimizer's update) on a second stream for a few thousand launches
while k_mlp_bwd128 loops on the first, and holds BOTH sides to the bits of a quiet run."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_torch_fp32_elementwise_kernels_are_bit_stable_beside_the_mfma_backward():
    from autolabel_amd import hip as H
    from autolabel_amd.pipeline import ModelLayout, Params
    H.require_gpu()
    dev = torch.device('cuda', 0)
    L = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=2.0)
    P = Params(L, dev); P.init_(seed=0)
    g = torch.Generator().manual_seed(0)
    rows = 1 << 18
    x = (torch.randn(rows, 48, generator=g) * 0.5).half().to(dev)
    d_out = (torch.randn(rows, 16, generator=g) * 0.02).half().to(dev)
    d_in = torch.empty(rows, 48, dtype=torch.float16, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    desc = P.descs['sigma']
    gp = C.c_void_p(P.grad.data_ptr() + 4 * L.offsets['sigma'])
    slabs = P.frags['sigma'][3]

    def mlp_bwd(stream):
        H.call('aln_mlp_bwd', C.byref(desc), H.ptr(x), None, None, H.ptr(d_out), rows, None, None, None, H.ptr(d_in), gp, H.ptr(flag),
               C.c_void_p(stream.cuda_stream))

    n = 1 << 22
    a = torch.randn(n, generator=g).to(dev); b = torch.randn(n, generator=g).to(dev); c_ = torch.randn(n, generator=g).to(dev)
    s_mfma, s_elt = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    # ---- quiet references (nothing else on the GPU)
    with torch.cuda.stream(s_elt):
        ref_add, ref_mul, ref_fma = a + b, a * b, torch.addcmul(c_, a, b)
    with torch.cuda.stream(s_mfma):
        mlp_bwd(s_mfma)
    torch.cuda.synchronize()
    ref_din, ref_slabs = d_in.clone(), slabs.clone()
    used = int(H.lib().aln_mlp_bwd_blocks(C.byref(desc), rows)) * (48 * 128 + 128 * 128 + 128 * 16)
    # ---- both at once: the MFMA kernel loops on one stream, torch's kernels on the other; every result is compared on the device
    bad_elt = torch.zeros(1, dtype=torch.int64, device=dev)
    bad_mlp = torch.zeros(1, dtype=torch.int64, device=dev)
    out = torch.empty_like(a)
    # One round = ~1.2 ms of backward launches on one stream and ~1.2 ms of elementwise launches on the other, issued back to back so that
    # both streams stay busy for the whole run: 150 rounds x 36 x 3 = 16 200 elementwise launches (6.8e10 fp32 results) beside 3 600
    # launches of k_mlp_bwd128 (measured: both loops take about the same time; the test reports the overlap it achieved).
    rounds, per_round, bwd_per_round = 150, 36, 24
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(s_mfma); ev[2].record(s_elt)
    for _ in range(rounds):
        with torch.cuda.stream(s_mfma):
            for _ in range(bwd_per_round):
                mlp_bwd(s_mfma)
            bad_mlp += (d_in != ref_din).sum() + (slabs[:used] != ref_slabs[:used]).sum()
        with torch.cuda.stream(s_elt):
            for _ in range(per_round):
                torch.add(a, b, out=out); bad_elt += (out != ref_add).sum()
                torch.mul(a, b, out=out); bad_elt += (out != ref_mul).sum()
                torch.addcmul(c_, a, b, out=out); bad_elt += (out != ref_fma).sum()
    ev[1].record(s_mfma); ev[3].record(s_elt)
    torch.cuda.synchronize()
    t_mfma, t_elt = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])
    print(f'hazard test: backward stream busy {t_mfma:.0f} ms, elementwise stream busy {t_elt:.0f} ms (concurrent)')
    assert min(t_mfma, t_elt) > 0.5 * max(t_mfma, t_elt), (t_mfma, t_elt)      # the two loops really ran side by side most of the time
    assert int(flag.item()) == 0
    assert int(bad_mlp.item()) == 0, f'{int(bad_mlp.item())} values of k_mlp_bwd128 differ from the quiet run beside torch elementwise kernels'
    assert int(bad_elt.item()) == 0, (f'{int(bad_elt.item())} fp32 results of torch.add / mul / addcmul differ from the quiet run while k_mlp_bwd128 '
                                      'runs on another stream: the packed-fp32 hazard reaches code outside this library')
