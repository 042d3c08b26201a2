#!/usr/bin/env python3
"""GPU box: does a two-handle pipeline hide the launch tail and the fixed cost per call when ONE stream has priority over the other?
usage: exp_prio.py <table> <esn0> <cap> <nf>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fec_testlib as T
from dvbs2rx_amd import LdpcDecoder, capi
table, esn0, cap, nf = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
N, K, _, _ = T.ldpc_info(table)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(4242)
cw = T.ldpc_encode(table, rng.integers(0, 2, (64, K), dtype=np.uint8))
n0 = 10.0 ** (-esn0 / 10.0)
tx = torch.from_numpy(np.tile((1.0 - 2.0 * cw.astype(np.float32)) * np.float32(0.5 ** 0.5), (nf // 64 + 1, 1))[:nf]).to(dev)
g = torch.Generator(device=dev); g.manual_seed(4242)
x = torch.clamp(torch.round((tx + (n0 / 2.0) ** 0.5 * torch.randn((nf, N), generator=g, device=dev)) * (2.0 * 2.0 ** 0.5 / n0)), -128, 127).to(torch.int8)
G = 32
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range", lo, hi)
def mk(n):
    hs = [LdpcDecoder(table=table, message_bits=K, group_size=G, max_frames=nf, max_trials=cap, outputmode=capi.OM_MESSAGE) for _ in range(n)]
    bs = [torch.empty((nf, K // 8), dtype=torch.uint8, device=dev) for _ in range(n)]
    rs = [torch.empty((nf + G - 1) // G, dtype=torch.int32, device=dev) for _ in range(n)]
    return hs, bs, rs
def pipe(n_handles, prios, ncalls=24, stagger_s=0.0):
    hs, bs, rs = mk(n_handles)
    sts = [torch.cuda.Stream(device=dev, priority=p) for p in prios]
    def enq(i):
        k = i % n_handles
        hs[k].enqueue_device(x.data_ptr(), nf, bs[k].data_ptr(), 0, rs[k].data_ptr(), sts[k].cuda_stream)
    def run():
        for i in range(ncalls):
            if i >= n_handles: hs[i % n_handles].finish()
            elif i and stagger_s:  # the first calls start out of phase
                t_end = time.perf_counter() + stagger_s
                while time.perf_counter() < t_end: pass
            enq(i)
        for h in hs: h.finish()
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    same = all(torch.equal(bs[0], b) for b in bs)
    fb = sum(h.fallback_rounds for h in hs)
    for h in hs: h.close()
    return nf * ncalls / sorted(ts)[1], same, fb
hs, bs, rs = mk(1)
st = torch.cuda.current_stream().cuda_stream
fn = lambda: hs[0].work_device(x.data_ptr(), nf, bs[0].data_ptr(), 0, rs[0].data_ptr(), st)
for _ in range(3): fn()
torch.cuda.synchronize(); ts = []
for _ in range(7):
    t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
sync = nf / sorted(ts)[3]
hs[0].close()
print(f"{table} {esn0} dB nf={nf}: synchronous {sync:.0f} fr/s")
call_s = nf / sync
for name, n, pr, stg in (("two handles, equal priority", 2, (0, 0), 0.0), ("two handles, second starts half a call later", 2, (0, 0), 0.5 * call_s),
                         ("two handles, second starts a quarter call later", 2, (0, 0), 0.25 * call_s),
                         ("three handles, each a third of a call later", 3, (0, 0, 0), call_s / 3)):
    try:
        r, same, fb = pipe(n, pr, 24, stg)
        print(f"  {name}: {r:.0f} fr/s ({r / sync:.3f} of synchronous) same results {same} fallback rounds {fb}", flush=True)
    except Exception as e:
        print(f"  {name}: failed: {e}")
