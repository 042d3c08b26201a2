#!/usr/bin/env python3
"""GPU box: what the demapper fused into the sweep kernel's frame load costs: 8PSK 3/4 chain from symbols against the LLR-domain chain and the
plain LDPC handle, at iteration caps 1 and 3 (never-converging input: the difference between the entries is the load stage)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dvbs2rx_amd import FecChain, LdpcDecoder, capi
dev = torch.device("cuda", 0); nf, G = 4096, 32
def tm(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
st = torch.cuda.current_stream().cuda_stream
for cap in (1, 3):
    ch = FecChain(rate="C3_4", constellation=capi.MOD_8PSK, group_size=G, max_frames=nf, max_trials=cap, device=0)
    syms = torch.randn((nf, ch.n_syms * 2), device=dev) * 0.7071; n0 = torch.tensor([1.0], device=dev)
    msg = torch.empty((nf, ch.msg_bytes), dtype=torch.uint8, device=dev); r = torch.empty(nf // G, dtype=torch.int32, device=dev); c = torch.empty(nf, dtype=torch.int32, device=dev)
    t_sym = tm(lambda: ch.work_device(syms.data_ptr(), nf, n0.data_ptr(), 1, msg.data_ptr(), r.data_ptr(), c.data_ptr(), st)); ch.close()
    cl = FecChain(rate="C3_4", group_size=G, max_frames=nf, max_trials=cap, device=0, from_llr=True)
    x = torch.clamp(torch.round(torch.randn((nf, 64800), device=dev) * 8), -128, 127).to(torch.int8)
    t_llr = tm(lambda: cl.work_llr_device(x.data_ptr(), nf, msg.data_ptr(), r.data_ptr(), c.data_ptr(), st)); cl.close()
    d = LdpcDecoder(standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C3_4", outputmode=capi.OM_MESSAGE, max_trials=cap, group_size=G, max_frames=nf, device=0)
    b = torch.empty((nf, d.out_bytes), dtype=torch.uint8, device=dev)
    t_ld = tm(lambda: d.work_device(x.data_ptr(), nf, b.data_ptr(), 0, r.data_ptr(), st)); d.close()
    print(f"cap {cap}: chain from symbols {t_sym:.3f} ms, chain from LLRs {t_llr:.3f} ms, LDPC alone {t_ld:.3f} ms per {nf} frames")
