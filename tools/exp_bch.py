import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gr-dvbs2rx_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from dvbs2rx_amd import BchDecoder, Demapper, capi
nf = 4096
for rate in ["C3_4", "C9_10"]:
    dec = BchDecoder(framesize=capi.FECFRAME_NORMAL, rate=rate, max_frames=nf)
    import fec_testlib as T
    mb, prim = T.BCH_FIELDS[capi.FECFRAME_NORMAL]; ob = T.OracleBch(mb, prim, dec.t, dec.n)
    words = torch.from_numpy(np.tile(ob.encode_bytes(np.random.default_rng(1).integers(0, 256, (64, dec.k // 8), dtype=np.uint8)), (nf // 64, 1))).cuda()
    for kind in ["garbage", "zero", "codewords"]:
        cw = torch.randint(0, 256, (nf, dec.n // 8), dtype=torch.uint8, device='cuda') if kind == "garbage" else torch.zeros((nf, dec.n // 8), dtype=torch.uint8, device='cuda') if kind == "zero" else words
        msg = torch.empty((nf, dec.k // 8), dtype=torch.uint8, device='cuda'); corr = torch.empty(nf, dtype=torch.int32, device='cuda')
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3): dec.work_device(cw.data_ptr(), nf, msg.data_ptr(), corr.data_ptr(), st)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): dec.work_device(cw.data_ptr(), nf, msg.data_ptr(), corr.data_ptr(), st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(f"BCH {rate} n={dec.n} t={dec.t} {kind}: {dt*1e3:.2f} ms / {nf} frames = {nf/dt/1e3:.0f} kframes/s, status counts {dict(zip(*[x.tolist() for x in np.unique(corr.cpu().numpy(), return_counts=True)]))}")
    dec.close()
for const, name in [] if os.environ.get('BCH_ONLY') else [(capi.MOD_QPSK, "QPSK"), (capi.MOD_8PSK, "8PSK")]:
    dm = Demapper(framesize=capi.FECFRAME_NORMAL, rate="C3_4", constellation=const, max_frames=nf)
    syms = torch.randn((nf, dm.n_syms * 2), device='cuda'); n0 = torch.tensor([0.3], device='cuda'); llr = torch.empty((nf, dm.n_llr), dtype=torch.int8, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): dm.work_device(syms.data_ptr(), nf, n0.data_ptr(), 1, llr.data_ptr(), st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): dm.work_device(syms.data_ptr(), nf, n0.data_ptr(), 1, llr.data_ptr(), st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    by = nf * (dm.n_syms * 8 + dm.n_llr)
    print(f"demap {name}: {dt*1e3:.3f} ms / {nf} frames, {by/dt/1e9:.0f} GB/s")
